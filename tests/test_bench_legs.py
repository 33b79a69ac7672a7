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
    d = run_leg(emu_lib, ["--workload", "longread", "--reads", "12", "--steps", "1", "--warmup", "1"], {"VGAMD_LONGREAD_BATCH": "6"})
    assert d["unit"] == "reads/s" and {"bound", "achieved", "peak", "frac", "traffic"} <= set(d["roofline"])
    p = d["parity"]
    assert p["checked"] == 12 and p["identical"] == 12 and p["differing_reads"] == 0
    c = d["config"]
    assert "2 lanes" in c["batches"] and c["one_lane"]["ms_per_batch"] > 0 and c["links"]["failed"] == 0 and d["problems_failed"] == 0
