# Round 3's closing run, second edition (after the X-drop band rework, the GBWT records taken over in place and the banded host changes):
# the whole -m gpu suite, smoke(), the headline line and the lines of the paths that changed (every command under `timeout`)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03_final; mkdir -p $O
timeout -s KILL 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
timeout -s KILL 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log | cut -c1-300
timeout -s KILL 600 python bench.py --steps 20 --warmup 5 > $O/bench_1M.json 2> $O/bench_1M.err; echo "bench rc=$?"
timeout -s KILL 300 python bench.py --workload xband > $O/bench_xband_200k.json 2> $O/xband.err; echo "xband rc=$?"
timeout -s KILL 300 python bench.py --workload banded > $O/bench_banded_100k.json 2> $O/banded.err; echo "banded rc=$?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03_final/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"]), d["unit"], round(d["ms_per_step"], 2), "parity", d.get("parity"), "frac", (d.get("roofline") or {}).get("frac"))
    except Exception as e:
        print(f, "unreadable", e)
PY
