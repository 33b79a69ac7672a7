"""The N>1 path on CPU: two processes (gloo, world_size 2) each align their shard of one
read stream with the emulated engine; rank 0 gathers and compares against a single-process
run of the whole stream.  Mirrors how bench.py shards reads across GPUs (no data-path collective)."""
import os
import subprocess
import sys
import textwrap

from util import ROOT

WORKER = textwrap.dedent("""
    import os, sys, json
    import numpy as np
    import torch.distributed as dist
    sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
    from gen import random_problem
    from vg_amd import capi, shard
    rank, local_rank, world = shard.env_rank()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(2024)
    problems = [random_problem(rng) for _ in range(301)]          # identical stream on every rank
    eng = capi.Engine(lib=os.path.join(%(root)r, "tests", "emu", "libvgamd_emu.so"))
    begin, res, cigars = shard.align_shard(eng, problems, rank, world)
    mine = {"begin": begin, "scores": [int(x) for x in res["score"]], "cigars": cigars}
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    dist.barrier()
    if rank == 0:
        scores, cig = [], []
        for g in sorted(gathered, key=lambda g: g["begin"]):
            assert g["begin"] == len(scores)
            scores += g["scores"]; cig += g["cigars"]
        _, res1, cig1 = shard.align_shard(eng, problems, 0, 1)
        assert scores == [int(x) for x in res1["score"]]
        assert cig == cig1
        print("SHARD_OK", len(scores))
    dist.destroy_process_group()
""")


def test_two_rank_gloo_sharding_matches_single_process(tmp_path):
    subprocess.check_call(["make", "-s", "emu"], cwd=ROOT)
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29541", str(script)],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "SHARD_OK 301" in out.stdout


def test_shard_ranges_cover_the_stream_exactly():
    sys.path.insert(0, ROOT)
    from vg_amd.shard import shard_range
    for n in (0, 1, 7, 8, 1000001):
        for world in (1, 2, 4, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(e - b for b, e in spans) - min(e - b for b, e in spans) <= 1
            pairs = [shard_range(n, r, world, group=2) for r in range(world)]       # read pairs stay on one rank
            assert pairs[0][0] == 0 and pairs[-1][1] == n
            assert all(pairs[i][1] == pairs[i + 1][0] and (pairs[i][1] % 2 == 0 or pairs[i][1] == n) for i in range(world - 1))
            assert max(e - b for b, e in pairs) - min(e - b for b, e in pairs) <= 3      # one pair, plus the odd read at the end of the stream


def test_host_thread_budget_is_checked_per_rank(monkeypatch):
    """N ranks share the host: a leg that still works on host threads must refuse to print a number when its share is too small
    (VERDICT r02 #9 / weak #12); the resident legs need one submitting thread and always pass."""
    import pytest
    from vg_amd import shard
    monkeypatch.delenv("VGAMD_ALLOW_HOST_STARVED", raising=False)
    assert shard.check_host_thread_budget("linear", 8, cores=16) == 2
    assert shard.check_host_thread_budget("config2", 8, cores=8) == 1
    assert shard.check_host_thread_budget("banded", 2, cores=16) == 8
    with pytest.raises(shard.HostThreadBudgetError) as e:
        shard.check_host_thread_budget("banded", 8, cores=16)
    assert "8 host threads per rank" in str(e.value) and "leave 2" in str(e.value)
    with pytest.raises(shard.HostThreadBudgetError):
        shard.check_host_thread_budget("longread", 4, cores=16)
    monkeypatch.setenv("VGAMD_ALLOW_HOST_STARVED", "1")
    assert shard.check_host_thread_budget("banded", 8, cores=16) == 2
