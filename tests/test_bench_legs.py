"""bench.py's secondary legs that keep batches in flight — the paired slice (request table on the device, rescue on the resident graph, two lanes) and the
long-read stage (two ChainStages) — run end to end on the emulated kernels (VGAMD_BENCH_ONE_DEVICE=1: the functional form, no GPU): the JSON line carries the
contract's objects and the leg's own parity against the oracle paths is complete.  (Numbers mean nothing here; the -m gpu box measures.)"""
import json
import os
import subprocess
import sys

import pytest

from util import EMU_LIB, ROOT


@pytest.fixture(scope="module")
def emu_lib():
    subprocess.check_call(["make", "-s", "emu", "host", "oracle"], cwd=ROOT)
    return EMU_LIB


def run_leg(emu_lib, args, env_extra):
    env = dict(os.environ, VGAMD_BENCH_ONE_DEVICE="1", VGAMD_ENGINE_LIB=emu_lib, **env_extra)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_default_run_ends_with_a_short_headline_line(emu_lib, tmp_path):
    """the driver keeps a bounded tail of stdout and parses the LAST line.  The default run's last line is the headline
    alone (< 4 KB, the only line that starts with "{"); every secondary record is an earlier `[secondary]` line and a record of
    bench_secondary.json."""
    side = str(tmp_path / "side.json")
    legs = [["gapless", ["--reads", "300", "--steps", "1", "--warmup", "0"], 600], ["wfa", ["--reads", "300", "--steps", "1", "--warmup", "0"], 600]]
    env = dict(os.environ, VGAMD_BENCH_ONE_DEVICE="1", VGAMD_ENGINE_LIB=emu_lib, VGAMD_BENCH_SECONDARY_LEGS=json.dumps(legs), VGAMD_BENCH_SIDE_FILE=side)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--reads", "400", "--steps", "2", "--warmup", "1", "--cpu-sample", "400"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.strip().splitlines()
    assert [l for l in lines if l.startswith("{")] == [lines[-1]]
    assert len(lines[-1]) < 4096
    d = json.loads(lines[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity"):
        assert k in d, k
    assert "secondary" not in d and d["steps"] == 2 and d["warmup"] == 1 and d["n_gpus"] == 1
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(d["roofline"])
    assert {"value", "unit", "cores", "kind", "sample"} <= set(d["cpu_baseline"])
    assert {"workload", "timed_region", "reads_per_gpu_per_step", "parallelism", "device"} <= set(d["config"])
    assert d["parity"]["checked"] == d["parity"]["identical"] > 0
    sec = [json.loads(l[len("[secondary] "):]) for l in lines if l.startswith("[secondary] ")]
    assert [r["workload"] for r in sec] == ["gapless", "wfa"] and all("error" not in r and r["value"] > 0 for r in sec), sec
    assert any(l.startswith("[detail] ") for l in lines)
    saved = json.load(open(side))
    assert saved["secondary"] == sec and "one_stream" in saved["headline_detail"]


def test_paired_leg_on_the_emulated_kernels(emu_lib):
    d = run_leg(emu_lib, ["--workload", "paired", "--reads", "1200", "--steps", "1", "--warmup", "1", "--cpu-sample", "1200"],
                {"VGAMD_PAIRED_REF_LEN": "300000", "VGAMD_PAIRED_BATCH": "300"})
    assert d["unit"] == "reads/s" and d["n_gpus"] == 1 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert {"bound", "achieved", "peak", "frac", "traffic"} <= set(d["roofline"]) and d["roofline_rescue"]["alg_bytes_per_batch"] > 0
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0
    p = d["parity"]
    assert p["checked"] == p["identical"] == 300 and p["rescued_mates_checked"] == p["rescued_alignments_identical"] > 5
    assert p["rescued_alignment_ops"] == p["rescued_alignment_ops_identical"] > 20
    c = d["config"]
    assert "two lanes" in c["batches"] and "rescue requests (device table + the mates' reads)" in c["stage_ms_per_batch"]
    assert c["rescue_rounds_in_batch_0"]["second_pass"] > 5 and d["problems_failed"] == 0


def test_longread_leg_on_the_emulated_kernels(emu_lib):
    d = run_leg(emu_lib, ["--workload", "longread", "--reads", "12", "--steps", "1", "--warmup", "1"], {"VGAMD_LONGREAD_BATCH": "6", "VGAMD_LONGREAD_REF_LEN": "400000"})
    assert d["unit"] == "reads/s" and {"bound", "achieved", "peak", "frac", "traffic"} <= set(d["roofline"])
    p = d["parity"]
    assert p["checked"] == 12 and p["identical"] == 12 and p["differing_reads"] == 0
    assert p["composed_alignments"]["identical"] == 12 and p["composed_alignments"]["mappings"] > 12 * 300 and p["composed_alignments"]["broken_chains"] == 0 and "every edit run" in p["what"]
    assert "one alignment per read: pieces + vgk_chain_stitch" in d["config"]["stage_ms_per_batch"]
    c = d["config"]
    assert "2 lanes" in c["batches"] and c["one_lane"]["ms_per_batch"] > 0 and c["links"]["failed"] == 0 and d["problems_failed"] == 0
