#!/usr/bin/env python3
"""bench.py — reads/sec aligned (150 bp) on N MI355X, next to the CPU path.

One "step" = one pass of the hot path (batched graph Smith-Waterman with
traceback: fill kernel + traceback kernel) over one batch of synthetic reads
whose packed inputs are already resident in HBM.  Workload = BASELINE.json
configs[1]: linear 1 Mbp synthetic graph (32 bp nodes), 150 bp reads, 384-416 bp
windows, LOCAL alignment with traceback, default vg scoring 1/4/6/1/5.

Multi-GPU: ONE read stream of reads-per-GPU x N reads is cut into contiguous shards
(vg_amd/shard.py, pairs kept together), one process per GPU, no data-path collective
-> weak scaling.  torch is used only for process-group plumbing (barrier, max-reduce
of the times).  Besides the resident-kernel rate (`value`) every rank also streams its
shard from host buffers (pack + kernels + fetch), with the host's packing threads
divided among the ranks; the slowest rank sets `end_to_end_*_per_s`.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
RDEV = "cpu" if os.environ.get("VGAMD_BENCH_ONE_DEVICE") == "1" else "cuda"      # where the max-reduce of the times lives (gloo in the one-device check)


def _device_sync(torch):
    """torch.cuda.synchronize() — a no-op only in the one-device functional check on a box without a GPU (the emulated engine on the CPU:
    VGAMD_BENCH_ONE_DEVICE=1 with VGAMD_ENGINE_LIB naming tests/emu's library), where there is no device to wait for"""
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    elif os.environ.get("VGAMD_BENCH_ONE_DEVICE") != "1":
        raise SystemExit("bench.py: no GPU visible (the measurement needs one; VGAMD_BENCH_ONE_DEVICE=1 is the functional check)")


def _set_device(torch, index):
    if torch.cuda.is_available():
        torch.cuda.set_device(index)
    elif os.environ.get("VGAMD_BENCH_ONE_DEVICE") != "1":
        raise SystemExit("bench.py: no GPU visible (the measurement needs one; VGAMD_BENCH_ONE_DEVICE=1 is the functional check)")


def relaunch_on_ranks(n_gpus):
    """`python bench.py --gpus N` without a launcher around it (no WORLD_SIZE in the environment): run this very command line again as N ranks,
    one per GPU, under torch.distributed.run on 127.0.0.1 — what the driver's own `python -m torch.distributed.run --nproc-per-node N bench.py
    --gpus N ...` does — and hand its exit code back.  Rank 0 of that run prints the JSON line (n_gpus = N)."""
    import socket
    import subprocess
    with socket.socket() as sock:                     # a free port for the rendezvous
        sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


# HBM bytes per unit of work of the dominant kernel: STORED constants from the PMC passes committed under profiles/r02 (rocprofv3 --pmc
# FETCH_SIZE and --pmc WRITE_SIZE in separate runs of the same workload; KiB per launch, FETCH_SIZE doubled as MI355X_MICROARCH.md
# prescribes for gfx950).  bench.py cannot read counters itself, so `roofline.traffic` = this figure x the units of one launch and
# `traffic_source` says so; re-measure with `tools/gpu_r06.sh pmc <workload>` + tools/pmc_constants.py.
def _hbm_in_use(torch, index=0):
    """bytes of the device's HBM in use right now, by anyone (hipMemGetInfo through torch); None where there is no device"""
    try:
        free, total = torch.cuda.mem_get_info(index)
        return int(total - free)
    except Exception:
        return None


def _pmc_constants():
    """profiles/pmc_constants.json (`tools/gpu_r06.sh pmc` + tools/pmc_constants.py): per workload the counter bytes per unit and the commit
    they were measured at; round 2's figures stand in for a workload the file does not hold."""
    r02 = {"linear": (2 * 704590 + 13574060) * 1024 / 400000, "banded": (2 * 406349 + 2245028) * 1024 / 100000,
           "gapless": (2 * 11460323 + 2860944) * 1024 / 1000000, "wfa": (2 * 4094544 + 1914811) * 1024 / 500000}
    values = dict(r02); source = {k: "profiles/r02 (commit not recorded)" for k in r02}
    try:
        for k, v in json.load(open(os.path.join(ROOT, "profiles", "pmc_constants.json"))).items():
            values[k] = v["bytes_per_unit"]; source[k] = "%s at commit %s" % (", ".join(v["files"]), v["commit"])
    except (OSError, ValueError, KeyError):
        pass
    return values, source


PMC_BYTES_PER_UNIT, PMC_SOURCE = _pmc_constants()


def traffic_source(workload):
    return "stored constant from the rocprofv3 PMC passes %s (not measured in this run), scaled by the units of one launch" % PMC_SOURCE[workload]




def bench_gapless(args, eng, rank, world, dist, torch, dev_name, cus):
    """giraffe's gapless-extension stage (secondary line).  One vgk_gapless_extend call packs the batch, moves it to HBM, runs the kernel
    and fetches the extension sets (timed separately as end_to_end_from_host_buffers); the timed region then re-launches the kernel
    K times on the inputs that stayed resident in HBM (vgk_gapless_rerun), which is what `value` reports."""
    import numpy as np
    from vg_amd import capi, shard, workloads
    n = min(args.reads, int(os.environ.get("VGAMD_GAPLESS_MAX_READS", "1000000")))
    wl = workloads.GaplessWorkload(n, seed=123 + rank)
    index = eng.haplo_index(wl.nodes, wl.threads)          # the haplotype index is resident in HBM from here on

    def barrier():
        _device_sync(torch)
        if dist is not None:
            dist.barrier()
        _device_sync(torch)

    eng.gapless_extend(index, wl.gs)                       # warms the cached buffers
    te = time.perf_counter(); out = eng.gapless_extend(index, wl.gs); te = time.perf_counter() - te
    for _ in range(args.warmup):
        eng.gapless_rerun()
    barrier()
    kms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.gapless_rerun()                                # synchronous: the kernel on the resident batch
        kms.append(eng.gapless_last_ms())
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=RDEV)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    res, ext, nodes, mism = out
    cpu = parity = None
    if rank == 0 and world == 1 and not args.no_cpu:      # the CPU leg (checker + baseline) runs at N = 1 only
        ora = capi.Engine(capi.Scoring.simple(1, 4, 6, 1, 5), lib=os.path.join(ROOT, "oracle", "libvgoracle.so"))
        ora.lib.vgo_set_threads(shard.usable_cpus())
        oidx = ora.haplo_index(wl.nodes, wl.threads)
        tc = time.perf_counter(); o = ora.gapless_extend(oidx, wl.gs); tc = time.perf_counter() - tc
        cpu = {"value": n / tc, "unit": "reads/s", "cores": shard.usable_cpus(), "kind": "port", "impl": "scalar checker (oracle/vgo_gapless.c: un-tuned restatement, built -march=x86-64-v2); not a tuned CPU baseline",
               "sample": "the same %d reads, oracle/vgo_gapless.c (scalar best-first extension over the uncompressed haplotype index), OpenMP over reads" % n}
        same = all(len(a) == len(b) and bool((a == b).all()) for a, b in zip(o, out))
        parity = {"checked": n, "identical": n if same else int((o[0] == res).sum())}
    if rank == 0:
        k = (sum(kms) / len(kms)) or 1e-9          # (the emulated engine of the functional check reports no kernel time)
        ns = np.diff(wl.gs.seed_off); rl = np.diff(wl.gs.read_off)
        alg_bytes = float((rl * (1 + ns) + 8 * ns).sum() + 60 * len(ext) + 4 * len(nodes) + 4 * len(mism))
        achieved = alg_bytes / (k * 1e-3) / 1e9
        print(json.dumps({
            "metric": "reads/sec gapless-extended (150 bp, %.1f seeds per read)" % float(ns.mean()), "value": n * world * args.steps / elapsed,
            "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "1 Mbp variation graph, 8 random haplotype threads, %d x 150 bp reads per GPU from either strand with 1 %% "
                                   "substitutions, seeds at true positions; GaplessExtender semantics (max 4 mismatches, overlap 0.8, trim)" % n,
                       "timed_region": "K launches of the three gapless kernels (search, rules, slab retries) on the batch resident in HBM (vgk_gapless_rerun)",
                       "end_to_end_from_host_buffers_reads_per_s": n / te, "parallelism": "read-sharded x%d" % world,
                       "device": dev_name, "compute_units": cus},
            "roofline": {"bound": "hbm", "limiter": "memory latency (a lane owns 128 B of L2: every hop fetches its record, bases and read words from beyond it) and divergent instruction issue, not bandwidth (DESIGN.md §11): `frac` prices the algorithmic bytes against the HBM peak as the contract asks", "kernel": "gapless_search_kernel + gapless_rules_kernel (+ gapless_kernel for reads that outgrow the LDS queue)", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": PMC_BYTES_PER_UNIT["gapless"] * n, "traffic_source": traffic_source("gapless"), "alg_bytes_per_launch": alg_bytes, "avg_launch_ms": k,
                         "kernel_only_reads_per_s": n / (k * 1e-3)},
            "cpu_baseline": cpu, "parity": parity, "problems_failed": int((res["status"] != 0).sum()),
            "full_length_fraction": float(res["full_length"].mean())}))
    if dist is not None:
        dist.destroy_process_group()


def bench_wfa(args, eng, rank, world, dist, torch, dev_name, cus):
    """The long-read chaining stage's WFA problems (configs[5], secondary line).  One vgk_wfa_extend call packs the batch, moves it to
    HBM, runs the kernel and fetches the alignments (timed separately as end_to_end_from_host_buffers); the timed region then
    re-launches the kernel K times on the inputs that stayed resident in HBM (vgk_wfa_rerun), which is what `value` reports."""
    import numpy as np
    from vg_amd import capi, shard, workloads
    n = min(args.reads, int(os.environ.get("VGAMD_WFA_MAX_PROBLEMS", "500000")))
    wl = workloads.WfaWorkload(n, seed=321 + rank)
    index = eng.haplo_index(wl.nodes, wl.threads)

    def barrier():
        _device_sync(torch)
        if dist is not None:
            dist.barrier()
        _device_sync(torch)

    eng.wfa_extend(index, wl.ws)                           # warms the cached buffers
    te = time.perf_counter(); out = eng.wfa_extend(index, wl.ws); te = time.perf_counter() - te
    for _ in range(args.warmup):
        eng.wfa_rerun()
    barrier()
    kms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.wfa_rerun()                                    # synchronous: the kernel on the resident batch
        kms.append(eng.wfa_last_ms())
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=RDEV)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    res, paths, edits = out
    cpu = parity = None
    if rank == 0 and world == 1 and not args.no_cpu:      # the CPU leg (checker + baseline) runs at N = 1 only
        ora = capi.Engine(capi.Scoring.simple(1, 4, 6, 1, 5), lib=os.path.join(ROOT, "oracle", "libvgoracle.so"))
        ora.lib.vgo_set_threads(shard.usable_cpus())
        oidx = ora.haplo_index(wl.nodes, wl.threads)
        tc = time.perf_counter(); o = ora.wfa_extend(oidx, wl.ws); tc = time.perf_counter() - tc
        cpu = {"value": n / tc, "unit": "alignments/s", "cores": shard.usable_cpus(), "kind": "port", "impl": "scalar int32 checker (oracle/vgo_wfa.c: un-tuned restatement); not a tuned CPU baseline",
               "sample": "the same %d problems, oracle/vgo_wfa.c (scalar WFA over the haplotype trie, hash-table wavefronts), OpenMP over problems" % n}
        good = res["status"] == 0
        fields = ("ok", "score", "node_offset", "seq_offset", "length", "path_len", "n_edits")
        same = np.ones(n, dtype=bool)
        for f in fields:
            same &= o[0][f] == res[f]
        # paths / edits element-wise, problem by problem (offsets differ once a problem hit a kernel table limit)
        for i in np.nonzero(good & same)[0][:: max(1, n // 20000)]:
            a, b = o[0][i], res[i]
            same[i] = bool((o[1][a["path_begin"]:a["path_begin"] + a["path_len"]] == paths[b["path_begin"]:b["path_begin"] + b["path_len"]]).all()
                           and (o[2][a["edit_begin"]:a["edit_begin"] + a["n_edits"]] == edits[b["edit_begin"]:b["edit_begin"] + b["n_edits"]]).all())
        parity = {"checked": int(good.sum()), "identical": int((same & good).sum()), "kernel_table_limit": int((~good).sum())}
    if rank == 0:
        k = (sum(kms) / len(kms)) or 1e-9          # (the emulated engine of the functional check reports no kernel time)
        alg_bytes = float(wl.bases + wl.graph_bases + 32 * n + 40 * n + 4 * len(paths) + 4 * len(edits))
        achieved = alg_bytes / (k * 1e-3) / 1e9
        print(json.dumps({
            "metric": "WFA alignments/sec (connect <= 250 bp, tails <= 100 bp)", "value": n * world * args.steps / elapsed,
            "unit": "alignments/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": "1 Mbp variation graph, 8 random haplotype threads, %d WFA problems per GPU: 80 %% connect between anchors "
                                   "50..250 bp apart, 20 %% prefix/suffix tails <= 100 bp, 0.5 %% errors (half substitutions, half 1-bp indels); "
                                   "WFAExtender semantics, default error model" % n,
                       "timed_region": "K launches of the WFA kernels on the batch resident in HBM (vgk_wfa_rerun)",
                       "kernels": (lambda w: {"first_launch_ms": w[0], "wavefront_kernel_behind_it_ms": w[1], "handed_over": int(w[2]),
                                              "form": "hybrid, the thread kernel and the wavefront kernel at once" if w[1] == 0 and w[2] else "hybrid, one kernel after the other" if w[2] else "one kernel"})(eng.wfa_last_wave()),
                       "end_to_end_from_host_buffers_alignments_per_s": n / te, "parallelism": "problem-sharded x%d" % world,
                       "device": dev_name, "compute_units": cus, "sequence_bases": wl.bases},
            "roofline": {"bound": "hbm", "limiter": "memory latency on each thread's dependent chain and lane divergence, not bandwidth (DESIGN.md §12)", "kernel": {"thread": "wfa_kernel", "wave": "wfa_wave_kernel"}.get(os.environ.get("VGAMD_WFA_KERNEL", ""), "wfa_kernel + wfa_wave_kernel (hybrid: a problem is handed to a wavefront at 16 points)"), "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": PMC_BYTES_PER_UNIT["wfa"] * n, "traffic_source": traffic_source("wfa"), "alg_bytes_per_launch": alg_bytes, "avg_launch_ms": k,
                         "kernel_only_alignments_per_s": n / (k * 1e-3)},
            "cpu_baseline": cpu, "parity": parity, "problems_failed": int((res["status"] != 0).sum()),
            "aligned_fraction": float(res["ok"].mean())}))
    if dist is not None:
        dist.destroy_process_group()


def bench_banded(args, eng, rank, world, dist, torch, dev_name, cus):
    """configs[4] stand-in (secondary line, not the headline metric).  One vgk_banded_align call does the host band geometry, moves the
    batch to HBM, runs fill + traceback and fetches the results (timed separately as end_to_end_from_host_buffers); the timed region
    then re-launches the kernels K times on the inputs that stayed resident in HBM (vgk_banded_rerun), which is what `value` reports."""
    import numpy as np
    from vg_amd import capi, shard, workloads
    n = min(args.reads, 100_000)
    wl = workloads.BandedWorkload(n, seed=99 + rank)

    def barrier():
        _device_sync(torch)
        if dist is not None:
            dist.barrier()
        _device_sync(torch)

    # from host buffers: the call as a caller makes it — a large one runs as four sub-batches, two in flight (banded_align_pipelined)
    eng.banded_align(wl.bs); eng.banded_align(wl.bs)       # warm the cached staging and device buffers
    te = time.perf_counter()
    for _ in range(3):
        res_p, ops_p = eng.banded_align(wl.bs)
    te = (time.perf_counter() - te) / 3
    # ... with the geometry made by prepare() on the host threads (the path of round 3 and of graphs with empty nodes)
    os.environ["VGAMD_BANDED_HOST_GEOMETRY"] = "1"
    eng.banded_align(wl.bs)
    th = time.perf_counter()
    for _ in range(3):
        res_h, ops_h = eng.banded_align(wl.bs)
    th = (time.perf_counter() - th) / 3
    del os.environ["VGAMD_BANDED_HOST_GEOMETRY"]
    assert res_p.tobytes() == res_h.tobytes() and ops_p.tobytes() == ops_h.tobytes()
    # ... and once as ONE batch that stays resident in HBM, which the timed region re-launches
    os.environ["VGAMD_BANDED_ONE_BATCH"] = "1"
    eng.banded_align(wl.bs)
    t1 = time.perf_counter(); res, ops = eng.banded_align(wl.bs); te_one = time.perf_counter() - t1
    del os.environ["VGAMD_BANDED_ONE_BATCH"]
    assert res_p.tobytes() == res.tobytes() and ops_p.tobytes() == ops.tobytes()
    cells = eng.banded_last(2); alg_bytes = eng.banded_last(3)
    for _ in range(args.warmup):
        eng.banded_rerun()
    barrier()
    fill_ms, walk_ms = [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.banded_rerun()                                 # synchronous: fill launches + traceback on the resident batch
        fill_ms.append(eng.banded_last(0)); walk_ms.append(eng.banded_last(1))
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=RDEV)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    cpu = parity = None
    if rank == 0 and world == 1 and not args.no_cpu:      # the CPU leg (checker + baseline) runs at N = 1 only
        ora = capi.Engine(capi.Scoring.simple(1, 4, 6, 1, 5), lib=os.path.join(ROOT, "oracle", "libvgoracle.so"))
        ora.lib.vgo_set_threads(shard.usable_cpus())
        tc = time.perf_counter(); ores, oops = ora.banded_align(wl.bs); tc = time.perf_counter() - tc
        cpu = {"value": n / tc, "unit": "alignments/s", "cores": shard.usable_cpus(), "kind": "port", "impl": "scalar int32 checker (oracle/vgo_banded.c: un-tuned restatement; the reference picks int8 / int16 cells where they fit); not a tuned CPU baseline",
               "sample": "the same %d problems, oracle/vgo_banded.c scalar int32 three-matrix DP + traceback, OpenMP over problems" % n}
        hdr = (res["score"] == ores["score"]) & (res["status"] == ores["status"]) & (res["n_ops"] == ores["n_ops"])
        same = int(hdr.sum())
        if hdr.all():
            bad_ops = ops.view(np.uint64) != oops.view(np.uint64)
            if bad_ops.any():
                owner = np.repeat(np.arange(n), ores["n_ops"])
                same = n - len(np.unique(owner[bad_ops]))
        parity = {"checked": n, "identical": same}
    if rank == 0:
        fill = sum(fill_ms) / len(fill_ms); walk = sum(walk_ms) / len(walk_ms)
        achieved = alg_bytes / (fill * 1e-3) / 1e9
        print(json.dumps({
            "metric": "banded global alignments/sec (anchor-to-anchor, 30-500 bp)", "value": n * world * args.steps / elapsed,
            "unit": "alignments/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "i32", "data": "synthetic",
            "config": {"workload": "configs[4] stand-in: 1 Mbp variation graph, %d anchor-to-anchor windows of 30-500 bp per GPU, "
                                   "BandedGlobalAligner semantics, permissive band, padding floor(sqrt(L))+1, scores 1/4/6/1" % n,
                       "timed_region": "K runs of the fill launches + traceback kernel on the batch resident in HBM (vgk_banded_rerun)",
                       "end_to_end_from_host_buffers_alignments_per_s": n / te, "end_to_end_host_geometry_alignments_per_s": n / th, "end_to_end_one_batch_alignments_per_s": n / te_one,
                       "from_host_buffers": "vgk_banded_align on the caller's arrays: four sub-batches, two in flight; the band geometry and the kernels' tables made on the device from the raw graph arrays (banded_geom_device.hpp; *_host_geometry_*: by prepare() on the host threads, as for graphs with empty nodes) — DESIGN.md §27.13",
                       "parallelism": "read-sharded x%d" % world, "device": dev_name, "compute_units": cus},
            "roofline": {"bound": "hbm", "limiter": "VALU issue: 1.74 G VALU wave-instructions per launch = 2.8 ms on 1024 SIMDs; ~41 VALU per wave-column of <= 64 cells inside the read, plus the per-node work (DESIGN.md §10)", "kernel": "banded_fill_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": PMC_BYTES_PER_UNIT["banded"] * n, "traffic_source": traffic_source("banded"), "alg_bytes_per_launch": alg_bytes, "avg_launch_ms": fill,
                         "traceback_ms": walk, "band_cells": cells, "gcups_fill": cells / (fill * 1e-3) / 1e9,
                         "kernel_only_alignments_per_s": n / ((fill + walk) * 1e-3)},
            "cpu_baseline": cpu, "parity": parity, "problems_failed": int((res["status"] != 0).sum())}))
    if dist is not None:
        dist.destroy_process_group()


def bench_wide(args, eng, rank, world, dist, torch, dev_name, cus):
    """The wide route (secondary line; VERDICT r04 weak #15: correct, tested, never timed): reads beyond the packed kernels' 1 024 rows — the long
    tails giraffe's chain alignment hands to the pinned X-drop, like the reference's own 4.4 kbp tail (src/unittest/minimizer_mapper.cpp:682-709), and
    long local alignments — through vgk_gssw_align, which sends them to gssw_wide_kernel (four wavefronts per problem, int32 cells, strips of rows
    through HBM) and gssw_wide_walk_kernel.  One step = one vgk_gssw_align call from host buffers (pack, H2D, kernels, D2H)."""
    import ctypes
    import numpy as np
    from vg_amd import capi, shard
    n = args.reads if args.reads else 2000
    rng = np.random.default_rng(97 + rank)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    problems = []
    for i in range(n):
        L = int(rng.integers(1100, 9000))
        ref = acgt[rng.integers(0, 4, L + 300)]
        read = ref[100:100 + L].copy()
        sub = rng.random(L) < 0.01
        read[sub] = acgt[rng.integers(0, 4, int(sub.sum()))]
        ins = np.nonzero(rng.random(L) < 0.002)[0]
        read = np.delete(read, ins)[:L]
        nodes, preds, at = [], [], 0
        while at < len(ref):                                            # a chain of 32-base nodes with a SNP bubble now and then
            ln = min(32, len(ref) - at)
            nodes.append(ref[at:at + ln].tobytes().decode()); preds.append([len(nodes) - 2] if len(nodes) > 1 else [])
            at += ln
            if rng.random() < 0.05 and at + 1 < len(ref):
                a = len(nodes) - 1
                alt = "ACGT"[(b"ACGT".index(ref[at]) + 1) % 4]
                nodes.append(chr(ref[at])); preds.append([a]); nodes.append(alt); preds.append([a])
                at += 1
                ln = min(32, len(ref) - at)
                if ln:
                    nodes.append(ref[at:at + ln].tobytes().decode()); preds.append([len(nodes) - 3, len(nodes) - 2]); at += ln
        mode = capi.VGK_XDROP_PINNED if i % 2 else capi.VGK_GSSW_LOCAL
        if mode == capi.VGK_XDROP_PINNED:                               # pinned at the graph's first base: the read starts there
            read = np.concatenate([ref[:100], read])[:L]
        problems.append(dict(read=read.tobytes().decode(), nodes=nodes, preds=preds, flags=mode | capi.VGK_GSSW_TRACEBACK, pinning=None, max_gap=40))
    ps = capi.ProblemSet.from_lists(problems)
    eng.lib.vgk_gssw_wide_last.restype = ctypes.c_double; eng.lib.vgk_gssw_wide_last.argtypes = [ctypes.c_void_p, ctypes.c_int]

    def barrier():
        _device_sync(torch)
        if dist is not None:
            dist.barrier()
        _device_sync(torch)

    for _ in range(max(1, args.warmup)):
        res, ops = eng.align_call(ps)
    barrier()
    fill_ms = walk_ms = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res, ops = eng.align_call(ps)
        fill_ms += eng.lib.vgk_gssw_wide_last(eng.h, 0); walk_ms += eng.lib.vgk_gssw_wide_last(eng.h, 1)
    barrier()
    elapsed = time.perf_counter() - t0
    cells = eng.lib.vgk_gssw_wide_last(eng.h, 2); tb_cells = eng.lib.vgk_gssw_wide_last(eng.h, 3); launches = eng.lib.vgk_gssw_wide_last(eng.h, 4)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=RDEV)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    cpu = parity = None
    if rank == 0 and world == 1 and not args.no_cpu:
        ora = capi.Engine(capi.Scoring.simple(1, 4, 6, 1, 5), lib=os.path.join(ROOT, "oracle", "libvgoracle.so"))
        cores = shard.usable_cpus(); ora.lib.vgo_set_threads(cores)
        k = min(n, args.cpu_sample or 400)
        sub = capi.ProblemSet.from_lists(problems[:k])
        t1 = time.perf_counter(); ores, oops = ora.align(sub); tc = time.perf_counter() - t1
        same = 0
        for i in range(k):
            a, b = res[i], ores[i]
            if all(a[f] == b[f] for f in ("status", "score", "end_node", "end_offset", "end_read", "first_offset", "n_ops")) and \
               (ops[a["ops_begin"]:a["ops_begin"] + a["n_ops"]].view(np.uint64) == oops[b["ops_begin"]:b["ops_begin"] + b["n_ops"]].view(np.uint64)).all():
                same += 1
        cpu = {"value": k / tc, "unit": "alignments/s", "cores": cores, "kind": "port", "impl": "scalar int32 checker (oracle/vgo_gssw.c, vgo_xdrop.c: DP + traceback, OpenMP over problems); not a tuned CPU baseline", "sample": "the first %d problems" % k}
        parity = {"checked": k, "identical": same, "what": "score, end cell, first offset and every op"}
    if rank == 0:
        read_b = float(np.diff(ps.read_off).sum()); graph_b = float(np.diff(ps.seq_off).sum())
        alg = read_b + graph_b + tb_cells + 16.0 * n + 2.0 * len(ops)       # inputs + a byte per cell with a traceback + results + ops
        ms = (fill_ms + walk_ms) / args.steps
        print(json.dumps({
            "metric": "alignments/sec on the wide route (reads of 1.1-9 kbp: pinned X-drop tails and local alignments with tracebacks)",
            "value": n * world * args.steps / elapsed, "unit": "alignments/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "i32", "data": "synthetic",
            "config": {"workload": "%d reads of 1 100-9 000 bp (1 %% substitutions, 0.2 %% deletions) against chains of 32-base nodes with SNP bubbles (read + 300 bases), half left-pinned X-drop, "
                                   "half LOCAL, all with tracebacks, scores 1/4/6/1/5" % n,
                       "timed_region": "per step one vgk_gssw_align call from host buffers: serial packing, H2D, gssw_wide_kernel<8|16> (a block of four wavefronts per problem), gssw_wide_walk_kernel, D2H",
                       "kernel_ms_per_step": {"fill": fill_ms / args.steps, "traceback": walk_ms / args.steps}, "launches_per_step": launches, "cells_per_step": cells,
                       "gcups_fill": cells / (fill_ms / args.steps * 1e-3) / 1e9 if fill_ms else None, "read_bases": read_b,
                       "parallelism": "read-sharded x%d" % world, "device": dev_name, "compute_units": cus},
            "roofline": {"bound": "hbm", "kernel": "gssw_wide_kernel + gssw_wide_walk_kernel", "limiter": "one block per problem: 2 000 problems fill 256 CUs once, the longest read's strips set the launch; int32 cells, 8 B per cell of strip carry",
                         "achieved": alg / (ms * 1e-3) / 1e9 if ms else None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS if ms else None,
                         "traffic": PMC_BYTES_PER_UNIT["wide"] * n if "wide" in PMC_BYTES_PER_UNIT else None, "traffic_source": traffic_source("wide") if "wide" in PMC_BYTES_PER_UNIT else None,
                         "alg_bytes_per_launch": alg, "avg_launch_ms": ms},
            "cpu_baseline": cpu, "parity": parity, "problems_failed": int((res["status"] != 0).sum())}))
    if dist is not None:
        dist.destroy_process_group()


def bench_longread(args, eng, rank, world, dist, torch, dev_name, cus):
    """configs[4] as reads (secondary line): 15 kbp HiFi-like reads cut at their anchors; every stretch between anchors through
    WFAExtender (connect / prefix / suffix); what it declines through align_sequence_between_consistently — the local graph between /
    beyond the anchors cut out of the haplotype graph, strands split, dagified, then BandedGlobalAligner (pinned X-drop for a tail)
    (src/minimizer_mapper_from_chains.cpp:2900-3300, :3342-3920).  The whole stage is ONE call into the host shim (vgh_chain_stage,
    vg_amd/host/chain_stage.cpp): one step = that call for the whole batch from host buffers, the extraction of the local graphs included.
    The headline and the parity line run WITHOUT a WFA point budget (the reference's WFAExtender has none); the budgeted rate is reported
    beside it, never instead."""
    import numpy as np
    import threading
    from vg_amd import pipeline, shard, workloads
    # a step = n reads in batches of `per`; two batches in flight (a lane = a ChainStage of its own: engine context, haplotype graph, WFA extender,
    # host threads), because a launch of the WFA kernel ends with ONE heavy link's dependent chain — the second lane's links run in the wavefronts
    # the first lane's launch has already given back, and its host work (local graphs, problem records) runs beside the other's kernels
    # (round 6: four lanes — the stage's host work per batch, descriptors, pieces and copies, is as long as its kernels: 190 k reads/s with four batches of
    # 8 000 in flight against 167 k with two, profiles/r06/sweeps.txt)
    want_lanes = 1 if os.environ.get("VGAMD_LONGREAD_ONE_LANE") else max(1, int(os.environ.get("VGAMD_LONGREAD_LANES", "4")))
    per = int(os.environ.get("VGAMD_LONGREAD_BATCH", "8000"))      # (a launch of the WFA kernel is at least its heaviest link's dependent chain, ~25 ms: 4 000 reads 28.7 ms, 8 000 36.1, 16 000 68.4)
    n = args.reads if args.reads else per * max(2, want_lanes)
    per = min(n, per)
    n_batches = max(1, n // per); n = n_batches * per
    n_lanes = min(want_lanes, n_batches)
    t0 = time.perf_counter()
    # the graph BASELINE.json names for this configuration: the chr22-scale SNP + indel graph of configs[2] / configs[3] (VariationGraph: 50.8 Mbp,
    # ~1.74 M nodes, two haplotypes); VGAMD_LONGREAD_REF_LEN=0: rounds 2-5's 1 Mbp graph with 8 random threads
    ref_len = int(os.environ.get("VGAMD_LONGREAD_REF_LEN", "50818468"))
    if ref_len:
        big = workloads.VariationGraph(ref_len=ref_len)
        t_graph = time.perf_counter() - t0
        wls = [workloads.LongReadWorkload.in_parallel(per, big, seed=515 + rank + 1000 * b, workers=max(1, shard.usable_cpus() // max(world, 1))) for b in range(n_batches)]
    else:
        t_graph = 0.0
        wls = [workloads.LongReadWorkload(per, seed=515 + rank + 1000 * b) for b in range(n_batches)]
    wl = wls[0]
    t_gen = time.perf_counter() - t0
    threads = max(1, shard.usable_cpus() // max(world, 1))
    lane_threads = max(1, threads // n_lanes)
    t0 = time.perf_counter()
    stages = [pipeline.ChainStage(w, device=eng.device) for w in wls]
    t_stages = time.perf_counter() - t0
    for st in stages:
        st.set_point_budgets(0, 0)
    lock = threading.Lock()
    # every step composes ONE alignment per read (find_chain_alignment's composed_path, simplified: vgk_chain_stitch) and brings it to the host;
    # VGAMD_LONGREAD_SCORES_ONLY=1: round 5's form (chain scores only), for comparisons
    compose = not os.environ.get("VGAMD_LONGREAD_SCORES_ONLY")

    def barrier():
        _device_sync(torch)
        if dist is not None:
            dist.barrier()
        _device_sync(torch)

    def keep(o):                                      # (the alignments are views of the stage's own arrays: copied before the stage runs again)
        if compose:
            o["alignments"] = tuple(np.array(x) for x in o["alignments"]); o["broken"] = np.array(o["broken"])
        return o

    def run_lane(which, timing, outs):
        for b in which:
            tm = {} if timing is not None else None
            o = stages[b].run(threads=lane_threads, timing=tm, compose=compose)
            with lock:
                outs[b] = o
                if timing is not None:
                    for k, v in tm.items():
                        timing[k] = timing.get(k, 0.0) + v

    def one_step(timing=None):
        outs = [None] * n_batches
        th = [threading.Thread(target=run_lane, args=(range(k, n_batches, n_lanes), timing, outs)) for k in range(1, n_lanes)]
        for t in th: t.start()
        run_lane(range(0, n_batches, n_lanes), timing, outs)
        for t in th: t.join()
        return outs

    def timed(steps, timing=None):
        barrier()
        t = time.perf_counter()
        for _ in range(steps):
            o = one_step(timing)
        barrier()
        e = time.perf_counter() - t
        if dist is not None:
            tt = torch.tensor([e], dtype=torch.float64, device=RDEV)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            e = float(tt.item())
        return e, o

    for _ in range(max(1, args.warmup)):
        outs = one_step()
    timing = {}
    elapsed, outs = timed(args.steps, timing)
    outs = [keep(o) for o in outs]
    out = outs[0]
    one_lane = None
    if n_lanes > 1:                                  # the same batches one after the other in one lane, for the record
        t1 = time.perf_counter()
        run_lane(range(n_batches), None, [None] * n_batches)
        one_lane = {"ms_per_batch": 1e3 * (time.perf_counter() - t1) / n_batches}
        one_lane["reads_per_s"] = per / (one_lane["ms_per_batch"] * 1e-3)
    # the same with round 2's point budgets (connects give up at 128 stored wavefront points and go to the DP route at once, tails at 512)
    budget = int(os.environ.get("VGAMD_WFA_POINT_BUDGET", "128")); tail_budget = int(os.environ.get("VGAMD_WFA_TAIL_BUDGET", "512"))
    for st in stages:
        st.set_point_budgets(budget, tail_budget)
    one_step()
    b_timing = {}
    b_elapsed, b_outs = timed(args.steps, b_timing)
    b_out = b_outs[0]
    for st in stages:
        st.set_point_budgets(0, 0)
    stats_sum = lambda os_: {k: int(sum(o["stats"][k] for o in os_)) for k in os_[0]["stats"]}
    n_links = sum(w.n for w in wls); read_bases = sum(w.read_bases for w in wls)
    cpu = parity = None
    if rank == 0 and world == 1 and not args.no_cpu:
        import ctypes
        cores = shard.usable_cpus()
        ctypes.CDLL(os.path.join(ROOT, "oracle", "libvgoracle.so")).vgo_set_threads(cores)
        same = b_same = differing = higher = 0; tc = 0.0
        aln_same = aln_other_route = n_mappings = n_edit_runs = broken = 0

        def same_alignments(x, y):
            """per read: is the composed alignment of x the one of y — status, every mapping (node, offset), every edit run"""
            (rx, mx, ex), (ry, my, ey) = x["alignments"], y["alignments"]
            hdr = np.ones(len(rx), dtype=bool)
            for f in ("status", "n_mappings", "n_edits", "from_length", "to_length"):
                hdr &= rx[f] == ry[f]
            if hdr.all() and mx.tobytes() == my.tobytes() and ex.tobytes() == ey.tobytes():      # same sizes -> same offsets: the dense arrays compare as wholes
                return hdr
            for r in np.nonzero(hdr)[0]:
                a = mx[int(rx["mapping_begin"][r]):int(rx["mapping_begin"][r]) + int(rx["n_mappings"][r])].copy(); a["edit_begin"] -= rx["edit_begin"][r]
                c = my[int(ry["mapping_begin"][r]):int(ry["mapping_begin"][r]) + int(ry["n_mappings"][r])].copy(); c["edit_begin"] -= ry["edit_begin"][r]
                hdr[r] = a.tobytes() == c.tobytes() and ex[int(rx["edit_begin"][r]):int(rx["edit_begin"][r]) + int(rx["n_edits"][r])].tobytes() == ey[int(ry["edit_begin"][r]):int(ry["edit_begin"][r]) + int(ry["n_edits"][r])].tobytes()
            return hdr
        for b in range(n_batches):                    # every batch of the step against the same stage over the oracle
            ora = pipeline.ChainStage(wls[b], lib=os.path.join(ROOT, "oracle", "libvgoracle.so"))
            t1 = time.perf_counter(); o = ora.run(threads=cores, compose=compose); tc += time.perf_counter() - t1
            diff = o["chain_score"] != outs[b]["chain_score"]
            if compose:
                ok = same_alignments(outs[b], o)
                # a link that took another route in the engine (its tables declined it: DP over the local graph, not bound to haplotypes) may be aligned differently
                route = np.bincount(wls[b].read_of, weights=(o["link_source"] != outs[b]["link_source"]), minlength=per) > 0
                aln_same += int((ok & ~diff).sum()); aln_other_route += int((~ok & route).sum())
                diff = diff | (~ok & ~route)
                n_mappings += len(outs[b]["alignments"][1]); n_edit_runs += len(outs[b]["alignments"][2]); broken += int(outs[b]["broken"].sum())
            ora.close()
            same += int((~diff).sum()) if not compose else int((ok & ~diff).sum()); differing += int(diff.sum()); higher += int((outs[b]["chain_score"][diff] > o["chain_score"][diff]).sum())
            b_same += int((o["chain_score"] == b_outs[b]["chain_score"]).sum())
        cpu = {"value": n / tc, "unit": "reads/s", "cores": cores, "kind": "port", "impl": "scalar int32 checkers: the same stage (vgh_chain_stage) bound to the oracle — vgo_wfa.c, vgo_banded.c, vgo_xdrop.c, vgo_chain.c (OpenMP over problems / reads); not a tuned CPU baseline",
               "sample": "all %d reads" % n}
        parity = {"checked": n, "identical": same, "wfa_point_budget": "none", "differing_reads": differing,
                  "differing_reads_where_the_engine_scores_higher": higher,
                  "identical_with_point_budgets": b_same,
                  "composed_alignments": {"identical": aln_same, "reads_with_a_link_on_another_route_and_another_alignment": aln_other_route, "mappings": n_mappings, "edit_runs": n_edit_runs, "broken_chains": broken} if compose else None,
                  "what": ("per read: the chain score (anchors + every link) AND the composed alignment — every mapping (node, offset) and every edit run of find_chain_alignment's "
                           "simplified Path, stitched on the device inside the timed region — against the same stage bound to the oracle, every batch of the step.  " if compose else "") +
                          "per-read chain score (anchors + every link), every batch of the step.  No point budget: a link leaves the WFA route only when the engine's tables decline it (VGK_ETOOBIG), "
                          "which the oracle's WFA (no tables) never does; such a link takes align_sequence_between, which is not bound to haplotypes and can only score "
                          "as high or higher"}
    wfa_ms = float(np.mean([o["wfa_kernel_ms"] or 0.0 for o in outs]))
    if rank == 0:
        print(json.dumps({
            "metric": "15 kbp reads/sec through the chain alignment stage (WFA between anchors; align_sequence_between — local graph extraction + banded global / pinned X-drop — for what WFA declines)",
            "value": n * world * args.steps / elapsed, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "i32", "data": "synthetic",
            "config": {"workload": ("configs[4]: %s, %%d reads of 15 000 bp per GPU on either strand, error-free 29-mer anchors every "
                                    "120-400 bp, 0.5 %%%% errors between them (half substitutions, half 1-bp indels), 1 %%%% of the connects with a 25-60 bp insertion; "
                                    "WFAExtender connect / prefix / suffix with the default error model and NO point budget; align_sequence_between_consistently for what it declines"
                                    % (("the chr22-scale SNP + indel graph (%d bp, %d nodes, two haplotypes)" % (ref_len, big.n_nodes)) if ref_len else "1 Mbp variation graph, 8 random haplotype threads")) % n,
                       "hbm_bytes": {"with the lanes' indexes, WFA tables and the step's arenas": _hbm_in_use(torch, eng.device)},
                       "setup_seconds": {"graph": t_graph, "graph + reads generated": t_gen, "stages (host graph, haplotype graph, index in HBM; one per lane)": t_stages},
                       "batches": "%d batches of %d reads per step; %s" % (n_batches, per, ("%d batches in flight: %d lanes (a ChainStage each: engine context, haplotype graph, WFA extender; %d host threads per lane)" % (n_lanes, n_lanes, lane_threads))
                                                                          if n_lanes > 1 else "one after the other in one lane"),
                       "one_lane": one_lane, "ms_per_batch": 1e3 * elapsed / args.steps / n_batches,
                       "timed_region": "per batch, from host buffers, one vgh_chain_stage call: vgk_wfa_extend over every link; for the declined links extract_connecting_graph / "
                                       "extract_extending_graph on the haplotype graph, strand split, dagify_from, tip trimming (host threads); one flush of banded-global / pinned X-drop problems; "
                                       "translation back to the base graph; per-read totals" + ("; the pieces of every read (anchors, WFA results in HBM, the DP route's Paths) composed into one alignment "
                                       "per read on the device (vgk_chain_stitch: to_path, append_path, simplify) and brought to the host" if compose else " (scores only: VGAMD_LONGREAD_SCORES_ONLY)"),
                       "problems": n_links, "problems_per_read": n_links / n, "read_bases": read_bases, "bases_per_s": read_bases * world * args.steps / elapsed,
                       "links": stats_sum(outs), "host_threads": threads,
                       "stage_ms_per_batch": {k: 1e3 * v / args.steps / n_batches for k, v in timing.items()}, "wfa_kernel_ms": wfa_ms, "wfa_launches_of_batch_0": out["wfa_launches"],
                       "stitch_device_ms": float(np.mean([o.get("stitch_kernel_ms", 0.0) for o in outs])) if compose else None,
                       "with_point_budgets": {"connect": budget, "tail": tail_budget, "reads_per_s": n * world * args.steps / b_elapsed, "ms_per_step": 1e3 * b_elapsed / args.steps,
                                              "links": stats_sum(b_outs), "stage_ms_per_batch": {k: 1e3 * v / args.steps / n_batches for k, v in b_timing.items()}, "wfa_kernel_ms": b_out["wfa_kernel_ms"]},
                       "parallelism": "read-sharded x%d" % world, "device": dev_name, "compute_units": cus, "generation_seconds": t_gen},
            "roofline": {"bound": "hbm", "kernel": "wfa_wave_kernel", "limiter": "the mass of easy links (12 wavefronts per CU, one link each) and the critical path of the heaviest one; memory latency, not bandwidth (DESIGN.md §21)",
                         # algorithmic bytes of a batch's WFA launch: every link's read bases and as many haplotype bases once, a problem descriptor and
                         # a result per link; against the launch's own device time (vgk_wfa_last_ms as the stage reports it, mean over the step's batches)
                         **(lambda alg, ms: {"achieved": alg / (ms * 1e-3) / 1e9 if ms else None, "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS if ms else None,
                                             "alg_bytes_per_launch": alg, "avg_launch_ms": ms})(float(2 * read_bases + 72 * n_links) / n_batches, wfa_ms),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "traffic": PMC_BYTES_PER_UNIT["longread"] * per if "longread" in PMC_BYTES_PER_UNIT else None,
                         "traffic_source": traffic_source("longread") if "longread" in PMC_BYTES_PER_UNIT else None},
            "cpu_baseline": cpu, "parity": parity,
            "problems_failed": int(sum(o["stats"]["failed"] + o["stats"]["no_graph"] + o["stats"]["too_big"] for o in outs))}))
    for st in stages:
        st.close()
    if dist is not None:
        dist.destroy_process_group()


def bench_paired(args, eng, rank, world, dist, torch, dev_name, cus):
    """A one-GPU slice of BASELINE.json configs[3] (secondary line): read PAIRS of 2 x 150 bp on a variation graph of the chr22-scale
    construction, both mates through seeding -> gapless extension -> tails (the configs[2] stage), then — giraffe's paired-end shape,
    MinimizerMapper::attempt_rescue (src/minimizer_mapper.cpp:3264-3440) — the mate without a full-length extension rescued from its
    partner's position: the rescue nodes, its best extension inside them as dozeu's seed, both passes of Aligner::align_xdrop as extension
    windows of the RESIDENT graph, fix_dozeu_score, fix_dozeu_end_deletions (vg_amd/host/rescue_resident.cpp).  One step = all batches of
    pairs from host buffers, two batches in flight (two engine contexts over one copy of the indexes, a host thread each) the way configs[2]
    runs.  The pairs of a stream stay together when it is sharded (shard.shard_range(group = 2)): this is what each rank of an 8-GPU run would do."""
    import numpy as np
    import threading
    from vg_amd import capi, pipeline, shard, workloads
    n_total = (args.reads // 2) if args.reads else 1_000_000                 # pairs per step
    n_pairs = min(n_total, int(os.environ.get("VGAMD_PAIRED_BATCH", "250000")))   # pairs per batch
    n_batches = max(1, n_total // n_pairs); n_total = n_batches * n_pairs
    t0 = time.perf_counter()
    wl_all = workloads.PairedWorkload(n_total, ref_len=int(os.environ.get("VGAMD_PAIRED_REF_LEN", "50818468")), seed=41 + rank)
    batches = [wl_all.batch(b * n_pairs, (b + 1) * n_pairs) for b in range(n_batches)]
    wl = batches[0]
    t_gen = time.perf_counter() - t0
    graph = (wl.node_len, wl.seq)
    hbm0 = _hbm_in_use(torch, eng.device)
    t1 = time.perf_counter(); index = eng.haplo_index(graph, wl.threads); t_hindex = time.perf_counter() - t1
    t1 = time.perf_counter(); mindex = eng.minimizer_index(graph, wl.threads); t_mindex = time.perf_counter() - t1
    eng.reuse_outputs = True
    eng.host_register(wl_all.reads)
    threads = int(os.environ.get("VGAMD_HOST_THREADS", "0")) or min(shard.usable_cpus(), 48)
    per_graph = bool(os.environ.get("VGAMD_PAIRED_PER_GRAPH"))
    n_lanes = 1 if (n_batches < 2 or os.environ.get("VGAMD_PAIRED_ONE_CONTEXT") or per_graph) else 2
    lane_threads = max(1, threads // n_lanes)
    # a lane = an engine context for the stage + a host aligner (its own context) with the graph resident in it for the rescue half
    # (vg_amd/host/rescue_resident.cpp); VGAMD_PAIRED_PER_GRAPH=1: round 4's form, one HashGraph per mate on host threads
    # (vg_amd/host/rescue_stage.cpp) — kept as the reference-shaped checker
    lanes = []
    for k in range(n_lanes):
        e = eng if k == 0 else capi.Engine(capi.Scoring.simple(1, 4, 6, 1, 5), device=eng.device, lib=eng.lib)
        e.reuse_outputs = True
        al = pipeline.HostAlignerHandle(os.environ.get("VGAMD_ENGINE_LIB"), device=eng.device)
        lanes.append((e, al, None if per_graph else al.rescue_graph(wl)))
    aligner, rg = lanes[0][1], lanes[0][2]

    def barrier():
        _device_sync(torch)
        if dist is not None:
            dist.barrier()
        _device_sync(torch)

    lock = threading.Lock()
    acc = {"kernel_ms": 0.0, "alg": 0.0, "cells": 0.0, "rescued": 0, "gapless_ms": 0.0}

    def run_lane(lane, which, timing, keep):
        e, al, g_res = lane
        for b in which:
            tm = {} if timing is not None else None
            o = pipeline.paired_stage(e, index, mindex, batches[b], al, timing=tm, host_threads=lane_threads, resident=g_res)
            with lock:
                if timing is not None:
                    for k2, v in tm.items():
                        timing[k2] = timing.get(k2, 0.0) + v
                    c = o["rescue_counts"]
                    acc["kernel_ms"] += c.get("kernel_ms", 0.0); acc["alg"] += c.get("alg_bytes", 0); acc["cells"] += c.get("cells", 0); acc["rescued"] += len(o["rescued"])
                    acc["gapless_ms"] += e.gapless_last_ms()
                if keep is not None and b == 0:
                    keep["out"] = o

    def one_step(timing=None, keep=None):
        th = [threading.Thread(target=run_lane, args=(lanes[k], range(k, n_batches, n_lanes), timing, keep)) for k in range(1, n_lanes)]
        for t in th: t.start()
        run_lane(lanes[0], range(0, n_batches, n_lanes), timing, keep)
        for t in th: t.join()

    for _ in range(max(1, args.warmup)):
        one_step()
    barrier()
    timing = {}; kept = {}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step(timing, kept)
    barrier()
    elapsed = time.perf_counter() - t0
    rescue_kernel_ms, rescue_alg, rescue_cells = acc["kernel_ms"], acc["alg"], acc["cells"]
    gapless_ms = acc["gapless_ms"] / max(1, args.steps * n_batches)
    one_context = None
    if n_lanes > 1:
        run_lane(lanes[0], range(n_batches), None, None)
        barrier(); t1 = time.perf_counter()
        run_lane(lanes[0], range(n_batches), None, None)
        barrier(); t_one = time.perf_counter() - t1
        one_context = {"ms_per_batch": 1e3 * t_one / n_batches, "reads_per_s": 2 * n_total / t_one}
    out = pipeline.paired_stage(eng, index, mindex, wl, aligner, host_threads=threads, resident=rg, want_ops=True)      # (batch 0 once more, untimed, with the rescued alignments' ops for the parity leg)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=RDEV)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    cpu = parity = None
    if rank == 0 and world == 1 and not args.no_cpu:
        ora_lib = os.path.join(ROOT, "oracle", "libvgoracle.so")
        ora = capi.Engine(capi.Scoring.simple(1, 4, 6, 1, 5), lib=ora_lib)
        cores = shard.usable_cpus(); ora.lib.vgo_set_threads(cores)
        k = min(n_pairs, (args.cpu_sample // 2) if args.cpu_sample else 100_000)
        sub = wl.subset(k)
        oidx = ora.haplo_index(graph, wl.threads); omi = ora.minimizer_index(graph, wl.threads)
        oal = pipeline.HostAlignerHandle(ora_lib)
        t1 = time.perf_counter()
        # the checker: the whole stage over the oracle, its rescue half in the REFERENCE-SHAPED form (one HashGraph and one Alignment per mate, Aligner::align_xdrop_many,
        # fix_dozeu_score, fix_dozeu_end_deletions: vg_amd/host/rescue_stage.cpp) — not the flat path the engine's side ran
        o = pipeline.paired_stage(ora, oidx, omi, sub, oal, oriented_len=np.repeat(wl.node_len, 2), device=False, host_threads=cores, want_ops=True)
        tc = time.perf_counter() - t1
        same = int((o["pair_score"] == out["pair_score"][:k]).sum())
        n_resc = len(o["rescued"])
        same_resc = same_ops = 0
        if n_resc and (o["rescued"] == out["rescued"][:n_resc]).all() and (o["requests"] == out["requests"][:n_resc]).all():
            row_same = (o["rescue"] == out["rescue"][:n_resc]).all(axis=1)
            ob, gb = o["rescue_ops_begin"].astype(np.int64), out["rescue_ops_begin"][:n_resc + 1].astype(np.int64)
            if (ob == gb).all():
                tot = int(ob[-1])
                op_bad = o["rescue_ops"][:tot].view(np.uint64) != out["rescue_ops"][:tot].view(np.uint64)
                owner_bad = np.unique(np.searchsorted(ob, np.nonzero(op_bad)[0], side="right") - 1)
                row_same[owner_bad] = False
                same_ops = tot - int(op_bad.sum())
            else:
                row_same &= np.diff(ob) == np.diff(gb)
            same_resc = int(row_same.sum())
        cpu = {"value": 2 * k / tc, "unit": "reads/s", "cores": cores, "kind": "port", "impl": "scalar checkers: the same stage over the oracle (vgo_minimizer.c, vgo_gapless.c, vgo_tail.c, vgo_xdrop.c) and the host shim's reference-shaped rescue path bound to it",
               "sample": "the first %d pairs" % k}
        parity = {"checked": k, "identical": same, "rescued_mates_checked": n_resc, "rescued_alignments_identical": same_resc,
                  "rescued_alignment_ops": int(o["rescue_ops_begin"][-1]), "rescued_alignment_ops_identical": same_ops,
                  "what": "per-pair score (mapped mate + the better of the rescued alignment and the stage's own); per rescued mate the request (node range, dozeu's seed), score, status, first node "
                          "and offset, mappings, aligned bases AND every (node, op, length) run of the final alignment, against the reference-shaped rescue path over the oracle"}
    if rank == 0:
        resc = out["rescue"]
        print(json.dumps({
            "metric": "150 bp paired reads/sec through giraffe's alignment stage with mate rescue (one-GPU slice of configs[3])",
            "value": 2 * n_total * world * args.steps / elapsed, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u16", "data": "synthetic",
            "config": {"workload": "configs[3] slice: variation graph of the chr22-scale construction over a %d bp reference (%d nodes, two haplotypes), %d pairs of 2 x 150 bp per GPU, fragments N(%.0f, %.0f), "
                                   "1 %% substitutions, 8 %% of the second mates with an inserted stretch and 3 %% substitutions; k = 29, w = 11 minimizers; rescue window = fragment mean +- 4 sd by column "
                                   "coordinate (a stand-in for subgraph_in_distance_range: the SnarlDistanceIndex is absent)" % (len(wl.graph.haps[0][0]), len(wl.node_len), n_total, wl.mean, wl.sd),
                       "batches": "%d batches of %d pairs per step; %s" % (n_batches, n_pairs, ("two batches in flight: two lanes (engine context + host aligner context + resident rescue graph each, one host thread each, "
                                                                                                  "%d worker threads per lane) over ONE copy of the haplotype and minimizer indexes" % lane_threads) if n_lanes > 1 else "one after the other in one lane"),
                       "one_context": one_context, "ms_per_batch": 1e3 * elapsed / args.steps / n_batches,
                       "timed_region": ("per step, one batch of pairs from host buffers: vgk_minimizer_seeds -> vgk_gapless_extend_seeded -> vgk_tail_stage for all 2 n reads, then the request table "
                                        "(vgk_rescue_requests: one lane per pair over the sets where the extension kernels left them; the host adds the lost mates' bytes), then vgh_rescue_stage_resident: both X-drop passes of every lost mate as extension windows of the "
                                        "resident graph (vgk_gssw_pack_extensions: sub-DAGs derived on the device), dozeu's scan and the full-DP fallback as plain windows, the fix-ups over flat arrays") if rg is not None else
                                       ("per step, one batch of pairs from host buffers: the stage for all 2 n reads, the rescue requests (numpy), then vgh_rescue_stage: one HashGraph per mate on host threads, "
                                        "Aligner::align_xdrop_many, the fix-ups [VGAMD_PAIRED_PER_GRAPH: round 4's form]"),
                       "rescue_rounds_in_batch_0": {k2: v for k2, v in out["rescue_counts"].items() if k2 in ("first_pass", "scans", "second_pass", "fallbacks")},
                       "pairs_rescued_in_batch_0": int(len(out["rescued"])), "rescued_with_positive_score": int((resc[:, 0] > 0).sum()), "refused_by_cell_budget": int((resc[:, 1] == 1).sum()),
                       "pairs_rescued_per_step": acc["rescued"] / args.steps,
                       "stage_ms_per_batch": {k: 1e3 * v / args.steps / n_batches for k, v in timing.items()}, "host_threads": threads,
                       "index_seconds": {"graph + pairs generated": t_gen, "haplotype index": t_hindex, "minimizer index": t_mindex},
                       "hbm_bytes": {"before the indexes": hbm0, "with indexes, lanes and the step's arenas": _hbm_in_use(torch, eng.device)},
                       "parallelism": "pair-sharded x%d" % world, "device": dev_name, "compute_units": cus, "generation_seconds": t_gen},
            # the leg's longest kernel family is the stage's gapless search (as in configs[2]); priced the same way: the reads once per seed + outputs
            "roofline": (lambda alg, ms: {"bound": "hbm", "kernel": "gapless_search_kernel + gapless_rules_kernel (the stage's longest kernels, as in configs[2])", "limiter": "memory latency and divergent issue, not bandwidth",
                                          "achieved": alg / (ms * 1e-3) / 1e9 if ms else None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS if ms else None,
                                          "traffic": PMC_BYTES_PER_UNIT["config2"] * 2 * n_pairs if "config2" in PMC_BYTES_PER_UNIT else None,      # (per launch = per batch)
                                          "traffic_source": (traffic_source("config2") + " — the same kernels on the same graph construction, per read") if "config2" in PMC_BYTES_PER_UNIT else None,
                                          "alg_bytes_per_launch": alg, "avg_launch_ms": ms})(
                float(wl.read_len * 2 * n_pairs * 5 + 60 * len(out["res"]) * 2), gapless_ms),
            # ... and the rescue half's own kernels: the X-drop fills and tracebacks of the three rounds (extension windows, scans, fallbacks), SURVEY §8(d)'s bytes per problem
            "roofline_rescue": {"bound": "valu", "kernel": "gssw_fill_kernel + walk kernels over the rescue rounds' batches", "achieved": rescue_alg / (rescue_kernel_ms * 1e-3) / 1e9 if rescue_kernel_ms else None,
                                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": rescue_alg / (rescue_kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if rescue_kernel_ms else None,
                                "traffic": PMC_BYTES_PER_UNIT["rescue"] * n_pairs if "rescue" in PMC_BYTES_PER_UNIT else None,      # (per batch of pairs, like the figures beside it)
                                "traffic_source": (traffic_source("rescue") + " — the counters name kernels, not callers: this is every gssw fill and walk of a step, the stage's tail windows "
                                                   "(rows-per-lane 16 / 20 / 24) as well as the rescue rounds (19), so an upper bound for the rescue half") if "rescue" in PMC_BYTES_PER_UNIT else None,
                                "alg_bytes_per_batch": rescue_alg / args.steps / n_batches, "kernel_ms_per_batch": rescue_kernel_ms / args.steps / n_batches,
                                "gcups": rescue_cells / (rescue_kernel_ms * 1e-3) / 1e9 if rescue_kernel_ms else None},
            "cpu_baseline": cpu, "parity": parity, "problems_failed": int((out["res"]["status"] != 0).sum())}))
    for e, al, g_res in lanes:
        if g_res is not None:
            g_res.close()
        al.close()
    if dist is not None:
        dist.destroy_process_group()


def bench_config2(args, eng, rank, world, dist, torch, dev_name, cus):
    """BASELINE.json configs[2] as the whole stage at its stated size (secondary line): the chr22-scale SNP + indel graph (50.8 Mbp, ~1.7 M
    nodes, two haplotypes: SURVEY §8(d)) with its haplotype index and its minimizer index resident in HBM; 10 M bare reads of 150 bp, in
    batches of 1 M, through minimizer seeding -> haplotype-consistent gapless extension -> tail forests -> the tails' X-drop alignments
    (vgk_minimizer_seeds, vgk_gapless_extend_seeded, vgk_tail_stage_aligned).  One step = all batches, from host buffers."""
    import numpy as np
    from vg_amd import capi, pipeline, shard, workloads
    n = args.reads if args.reads else 10_000_000
    batch = min(n, 1_000_000)
    t0 = time.perf_counter()
    ref_len = int(os.environ.get("VGAMD_CONFIG2_REF_LEN", "0"))                  # (a smaller reference: functional checks only)
    wl = workloads.Config2Workload(n, batch=batch, seed=31 + rank, graph=workloads.VariationGraph(ref_len=ref_len) if ref_len else None)
    t_gen = time.perf_counter() - t0
    graph = (wl.node_len, wl.seq)
    t0 = time.perf_counter(); index = eng.haplo_index(graph, wl.threads); t_hindex = time.perf_counter() - t0
    t0 = time.perf_counter(); mindex = eng.minimizer_index(graph, wl.threads); t_mindex = time.perf_counter() - t0
    eng.reuse_outputs = True
    with_policy = bool(os.environ.get("VGAMD_CONFIG2_POLICY"))              # find_seeds' choice of minimizers on the device (vgk_minimizer_set_policy, giraffe's defaults)
    if with_policy:
        mindex.set_policy(10, 500, 0.9)

    class Batch:
        def __init__(self, k): self.n = k

    def barrier():
        _device_sync(torch)
        if dist is not None:
            dist.barrier()
        _device_sync(torch)

    kernel_ms = {"minimizer": 0.0, "gapless": 0.0, "tails derived": 0.0, "tail forest": 0.0, "windows packed": 0.0, "x-drop fill + traceback + totals": 0.0}
    import threading
    acc_lock = threading.Lock()

    def run_batches(lane, which, tot, timing=None, keep=None):
        """the batches `which` of a step on one engine context (lane = (engine, haplotype index, minimizer index))"""
        e, hidx, midx = lane
        for b in which:
            reads, off = wl.batches[b]
            t1 = time.perf_counter()
            seed_off, _, mins = e.minimizer_seeds(midx, hidx, reads, off, keep_on_device=True)
            t2 = time.perf_counter()
            tm = {} if timing is not None else None
            out = pipeline.align_stage_device(e, hidx, Batch(len(off) - 1), seeded=int(seed_off[-1]), aligned=True, timing=tm)
            st = out["stats"]
            with acc_lock:
                if timing is not None:
                    timing["minimizer_seeds"] = timing.get("minimizer_seeds", 0.0) + t2 - t1
                    for k, v in tm.items():
                        timing[k] = timing.get(k, 0.0) + v
                    kernel_ms["minimizer"] += e.minimizer_last_ms(); kernel_ms["gapless"] += e.gapless_last_ms()
                    for k, v in zip(("tails derived", "tail forest", "windows packed", "x-drop fill + traceback + totals"), e.tail_stage_last_ms()):
                        kernel_ms[k] += v
                tot["seeds"] += int(seed_off[-1]); tot["ext"] += len(out["ext"]); tot["tails"] += int(st[0]); tot["trees"] += int(st[1]); tot["tree_nodes"] += int(st[2]); tot["failed"] += int(st[3])
                tot["full_length"] += int((out["res"]["full_length"] != 0).sum()); tot["truncated"] += int(e.minimizers_truncated.sum())
                if mins is not None:      # reads with more than 64 minimizers, seeded without find_seeds' policy (VGK_MINIMIZERS_POLICY_SKIPPED): a caller routes those through the shim's select_minimizers
                    tot["policy_skipped"] += int(((np.asarray(mins, dtype=np.int64) & 0x40000000) != 0).sum())
                if keep is not None and b == 0:
                    keep.update(read_score=out["read_score"].copy(), res=out["res"].copy(), ext=out["ext"].copy(), nodes=out["nodes"].copy(), seed_off=seed_off.copy(),
                                ext_total=out["ext_total"].copy(), tails=out["tails"].copy(), tail_ops=out["tail_ops"].copy(),
                                n_minimizers=int((np.asarray(mins, dtype=np.int64) & 0x7fffffff).sum()) if mins is not None else 19 * (len(off) - 1))

    def new_tot():
        return {"seeds": 0, "ext": 0, "tails": 0, "trees": 0, "tree_nodes": 0, "failed": 0, "full_length": 0, "truncated": 0, "policy_skipped": 0}

    lanes = [(eng, index, mindex)]
    # a streaming caller keeps its read buffers: page-locked once (vgk_host_register), they go up at the link's rate
    pinned = [bool(eng.host_register(reads)) for reads, _ in wl.batches] if not os.environ.get("VGAMD_CONFIG2_PAGEABLE") else []

    def one_step(timing=None, keep=None):
        """one context, one batch after the other"""
        tot = new_tot()
        run_batches(lanes[0], range(len(wl.batches)), tot, timing, keep)
        return tot

    # Two batches in flight, the way vg itself calls an aligner (many threads, one process): a second engine context with its own copy of
    # the two indexes, one host thread per context, alternate batches.  One context's seeding and copies run under the other's extension
    # and tail kernels; every batch's results are what the one-context form gives (checked below).  This is the timed form when a step
    # has two batches or more; `config.one_context` keeps the serial rate beside it.
    pipelined = len(wl.batches) >= 2 and not os.environ.get("VGAMD_CONFIG2_ONE_CONTEXT")
    n_contexts = max(2, min(int(os.environ.get("VGAMD_CONFIG2_CONTEXTS", "2")), len(wl.batches))) if pipelined else 1
    for _ in range(n_contexts - 1):
        eng_b = capi.Engine(capi.Scoring.simple(1, 4, 6, 1, 5), device=eng.device, lib=eng.lib)          # (the same library, the same device)
        eng_b.reuse_outputs = True
        if os.environ.get("VGAMD_CONFIG2_OWN_INDEXES"):          # (round 4's form: every context uploads its own copy of both indexes)
            lanes.append((eng_b, eng_b.haplo_index(graph, wl.threads), eng_b.minimizer_index(graph, wl.threads)))
            if with_policy:
                lanes[-1][2].set_policy(10, 500, 0.9)
        else:                                                     # the indexes are read-only tables of the device: one copy serves every context on it (include/vgk.h "sharing an index")
            lanes.append((eng_b, index, mindex))

    def one_step_pipelined(timing=None, keep=None):
        tot = new_tot()
        th = [threading.Thread(target=run_batches, args=(lanes[k], range(k, len(wl.batches), n_contexts), tot, timing, keep)) for k in range(n_contexts)]
        for t in th: t.start()
        for t in th: t.join()
        return tot

    step_fn = one_step_pipelined if pipelined else one_step
    for _ in range(max(1, args.warmup)):
        step_fn()
    barrier()
    timing = {}; first = {}
    t0 = time.perf_counter()
    for k in range(args.steps):
        tot = step_fn(timing, first if k + 1 == args.steps else None)      # (batch 0's products are copied aside for the parity leg once, in the last step: ~0.3 GB of copies)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=RDEV)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    one_context = None
    if pipelined:
        serial_first = {}
        one_step()
        barrier(); t1 = time.perf_counter()
        for k in range(max(1, args.steps // 2)):
            one_step(None, serial_first if k + 1 == max(1, args.steps // 2) else None)
        barrier(); t_one = (time.perf_counter() - t1) / max(1, args.steps // 2)
        one_context = {"ms_per_batch": 1e3 * t_one / len(wl.batches), "reads_per_s": n / t_one,
                       "read_scores_equal_to_the_two_context_run": bool((serial_first["read_score"] == first["read_score"]).all())}
    cpu = parity = None
    if rank == 0 and world == 1 and not args.no_cpu:
        ora = capi.Engine(capi.Scoring.simple(1, 4, 6, 1, 5), lib=os.path.join(ROOT, "oracle", "libvgoracle.so"))
        cores = shard.usable_cpus(); ora.lib.vgo_set_threads(cores)
        k = min(batch, args.cpu_sample or 200_000)
        reads, off = wl.batches[0]
        oidx = ora.haplo_index(graph, wl.threads); omi = ora.minimizer_index(graph, wl.threads)
        if with_policy:
            omi.set_policy(10, 500, 0.9)
        olen = np.repeat(wl.node_len, 2)
        t1 = time.perf_counter()
        so, sd, _ = ora.minimizer_seeds(omi, oidx, reads[:off[k]], off[:k + 1])
        sub = capi.GaplessSet(reads[:off[k]], off[:k + 1], sd, so, node_cap=len(sd) * 16, mism_cap=len(sd) * 12)
        o = pipeline.align_stage(ora, oidx, olen, sub); tc = time.perf_counter() - t1
        same = int((o["read_score"] == first["read_score"][:k]).sum())
        cpu = {"value": k / tc, "unit": "reads/s", "cores": cores, "kind": "port",
               "impl": "scalar checkers, not a tuned CPU baseline: the same stage over the oracle: vgo_minimizer.c, vgo_gapless.c (OpenMP over reads), vgo_tail.c, vgo_xdrop.c (OpenMP over problems)",
               "sample": "the first %d reads of the first batch" % k}
        # every product of the stage for those reads, not their scores alone: the seeds' counts, every extension of every set (interval, offset,
        # mismatches, score, both search states, every node of its path), every extension's total, and every TAIL ALIGNMENT the engine chose on the
        # device (vgk_tail_stage_aligned: score, read interval, first offset, every op with its oriented node) against the best tree's alignment of
        # the oracle's stage (pipeline.winning_alignment_arrays)
        tc2 = time.perf_counter()
        n_ext_k = int(o["res"]["n_ext"].sum())
        sets_same = pipeline.compare_extension_sets(first["res"], first["ext"], first["nodes"], o["res"], o["ext"], o["nodes"], k)
        totals_same = int((first["ext_total"][:n_ext_k] == o["ext_total"][:n_ext_k]).sum())
        tails_verdict = pipeline.compare_tail_alignments(first["tails"], first["tail_ops"], pipeline.winning_alignment_arrays(o), n_ext=n_ext_k)
        seeds_same = int((np.diff(first["seed_off"][:k + 1].astype(np.int64)) == np.diff(np.asarray(so, dtype=np.int64))).sum())
        parity = {"checked": k, "identical": min(same, sets_same, seeds_same) if tails_verdict["identical"] == tails_verdict["tails"] and totals_same == n_ext_k else min(same, sets_same, seeds_same, tails_verdict["identical"]),
                  "reads_with_identical_best_score": same, "reads_with_identical_seed_count": seeds_same, "reads_with_identical_extension_sets": sets_same,
                  "extensions": n_ext_k, "extensions_with_identical_total": totals_same,
                  "tail_alignments": tails_verdict["tails"], "tail_alignments_identical": tails_verdict["identical"], "tail_alignment_ops": tails_verdict["ops"],
                  "first_differing_tail": tails_verdict["first_bad"], "compare_seconds": time.perf_counter() - tc2,
                  "what": "ALIGNMENTS, not scores: for every one of the checked reads its seed count, every extension of its set (read interval, offset, mismatches, score, both haplotype "
                          "search states, every path node), every extension's total, and every tail alignment chosen on the device — score, read interval, first offset and every "
                          "(oriented node, op, length) — against the oracle stage's best tree per tail (src/minimizer_mapper.cpp:5654-5716); `identical` counts a read only when all of that agrees "
                          "(and is capped by the identical tail alignments when any tail differs)"}
    if rank == 0:
        steps = args.steps
        # the stage's dominant kernel: the gapless search; its algorithmic bytes as in --workload gapless (the read once per seed, outputs)
        ext, nodes, res = first["ext"], first["nodes"], first["res"]
        ns = np.diff(first["seed_off"]).astype(np.int64); rl = wl.read_len
        alg0 = float((rl * (1 + ns) + 8 * ns).sum() + 60 * len(ext) + 4 * len(nodes))          # batch 0
        gap_ms = kernel_ms["gapless"] / steps / len(wl.batches)
        achieved = alg0 / (gap_ms * 1e-3) / 1e9 if gap_ms else None
        print(json.dumps({
            "metric": "150 bp reads/sec through giraffe's alignment stage from bare reads on the chr22-scale graph (minimizer seeding, gapless extension, tail forests, X-drop tails with their alignments)",
            "value": n * world * steps / elapsed, "unit": "reads/s", "n_gpus": world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u16", "data": "synthetic",
            "config": {"workload": "configs[2]: chr22-scale graph (50 818 468 bp reference, 41 %% GC, SNPs 1/1000, indels of 1-20 bp 1/10 000, nodes <= 32 bp: %d nodes), two haplotypes carrying each "
                                   "variant with p = 0.5; %d reads of 150 bp per GPU from the haplotypes on either strand, 1 %% substitutions, 10 %% of them with one inserted base; "
                                   "k = 29, w = 11 minimizers, hit cap 500; max_mismatches 4; tails left-pinned X-drop against their haplotype trees, scores 1/4/6/1/5" % (len(wl.node_len), n),
                       "timed_region": "per step, %d batches of %d reads from host buffers: vgk_minimizer_seeds (clusters stay in HBM) -> vgk_gapless_extend_seeded (sets come down under the "
                                       "tail stage) -> vgk_tail_stage_aligned%s" % (len(wl.batches), batch, "; %d batches in flight: %d engine contexts, one host thread each, batches in turn" % (n_contexts, n_contexts) if pipelined else ""),
                       "one_context": one_context,
                       "indexes": ("one copy of the haplotype and minimizer indexes in HBM, shared by the %d contexts" % n_contexts) if pipelined and not os.environ.get("VGAMD_CONFIG2_OWN_INDEXES") else "one copy per context",
                       "read_buffers": "page-locked by the caller (vgk_host_register)" if pinned and all(pinned) else "pageable",
                       "policies": ("find_seeds' choice on the device (vgk_minimizer_set_policy: hit cap 10, hard cap 500 over a key's run, score fraction 0.9); clusters = all seeds of the chosen minimizers" if with_policy else
                                    "every minimizer of a read looked up, hit cap 500 (hard cap) per minimizer, no score-based selection (vgk_minimizer_set_policy exists: VGAMD_CONFIG2_POLICY=1 turns it on here); clusters = all seeds of a read"),
                       "per_step": {k: v / steps for k, v in tot.items()} if steps == 1 else tot,
                       "ms_per_batch": 1e3 * elapsed / steps / len(wl.batches),
                       "stage_ms_per_batch": {k: 1e3 * v / steps / len(wl.batches) for k, v in timing.items()},
                       "kernel_ms_per_batch": {k: v / steps / len(wl.batches) for k, v in kernel_ms.items()},
                       "index_seconds": {"graph + reads generated": t_gen, "haplotype index": t_hindex, "minimizer index": t_mindex, "minimizer keys": mindex.keys},
                       "parallelism": "read-sharded x%d" % world, "device": dev_name, "compute_units": cus},
            "roofline": {"bound": "hbm", "kernel": "gapless_search_kernel + gapless_rules_kernel (the stage's longest kernels)", "limiter": "memory latency and divergent issue, not bandwidth (DESIGN.md §11)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS if achieved else None,
                         "traffic": PMC_BYTES_PER_UNIT["config2"] * batch if "config2" in PMC_BYTES_PER_UNIT else None,
                         "traffic_source": traffic_source("config2") if "config2" in PMC_BYTES_PER_UNIT else None,
                         "alg_bytes_per_launch": alg0, "avg_launch_ms": gap_ms,
                         "launch_ms_note": "kernel time of the extension call per batch as the engine's own events give it; with two batches in flight it includes what the other context's kernels took of the device"},
            # the stage's second-longest kernel family on its own: the seeding (minimizer_kernel over four slices of the batch + the gather).
            # algorithmic bytes: the read once + a 16-byte slot per minimizer looked up + 8 bytes per seed written
            "roofline_minimizer": (lambda alg_m, ms_m: {"bound": "hbm", "kernel": "minimizer_kernel + minimizer_gather_kernel", "limiter": "64-bit integer issue (two Wang hashes per k-mer position), then the latency of the table lookups",
                                    "achieved": alg_m / (ms_m * 1e-3) / 1e9 if ms_m else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": alg_m / (ms_m * 1e-3) / 1e9 / HBM_PEAK_GBS if ms_m else None,
                                    "traffic": PMC_BYTES_PER_UNIT["minimizer"] * batch if "minimizer" in PMC_BYTES_PER_UNIT else None,
                                    "traffic_source": traffic_source("minimizer") if "minimizer" in PMC_BYTES_PER_UNIT else None,
                                    "alg_bytes_per_launch": alg_m, "avg_launch_ms": ms_m,
                                    "launch_ms_note": "device time of the seeding call per batch (its copies of the reads included)"})(
                float(rl * batch + 16 * first.get("n_minimizers", 19 * batch) + 8 * ns.sum()), kernel_ms["minimizer"] / steps / len(wl.batches)),
            "cpu_baseline": cpu, "parity": parity, "problems_failed": int(tot["failed"])}))
    if dist is not None:
        dist.destroy_process_group()


def bench_giraffe(args, eng, rank, world, dist, torch, dev_name, cus):
    """configs[2]'s alignment stage as giraffe runs it (secondary line): seeds -> haplotype-consistent gapless extension -> for the
    clusters no full-length extension resolves, tail forests -> the trees as left-pinned X-drop windows -> total scores
    (vg_amd/pipeline.py; src/minimizer_mapper.cpp:5480-5535).  One step = the whole stage for the batch FROM HOST BUFFERS: every call
    uploads its inputs and the extension sets come back to the host, where the tails are derived (numpy)."""
    import numpy as np
    from vg_amd import capi, pipeline, shard, workloads
    n = args.reads if args.reads else 1_000_000
    inserted = 0.1
    t0 = time.perf_counter()
    wl = workloads.GaplessWorkload(n, seed=123 + rank, inserted_reads=inserted)
    t_gen = time.perf_counter() - t0
    olen = np.repeat(np.array([len(s) for s in wl.nodes]), 2)
    index = eng.haplo_index(wl.nodes, wl.threads)

    def barrier():
        _device_sync(torch)
        if dist is not None:
            dist.barrier()
        _device_sync(torch)

    stage = pipeline.align_stage if os.environ.get("VGAMD_GIRAFFE_NUMPY_GLUE") else pipeline.align_stage_native      # the glue in the host shim (C++) or in numpy
    device_tails = not os.environ.get("VGAMD_GIRAFFE_HOST_TAILS") and stage is pipeline.align_stage_native          # ... or no glue: vgk_tail_stage on the device
    with_alignments = device_tails and bool(os.environ.get("VGAMD_GIRAFFE_ALIGNED"))      # vgk_tail_stage_aligned: the tails' winning alignments come back too
    from_reads = bool(os.environ.get("VGAMD_GIRAFFE_FROM_READS"))         # start from the bare reads: minimizer seeding on the device makes the clusters
    mindex = None; seeds_per_read = None; t_index = 0.0
    stay = from_reads and not os.environ.get("VGAMD_GIRAFFE_SEEDS_VIA_HOST") and stage is pipeline.align_stage_native      # the clusters stay on the device between seeding and extension
    step_seed_off = [None]
    if from_reads:
        t1 = time.perf_counter(); mindex = eng.minimizer_index(wl.nodes, wl.threads); t_index = time.perf_counter() - t1

    eng.reuse_outputs = True            # the loop consumes a step's outputs before the next step: one set of output arrays, reused (capi.Engine._out)

    def one_step(timing=None):
        gs = wl.gs
        seeded = None
        if from_reads:
            t1 = time.perf_counter()
            seed_off, seeds, mins = eng.minimizer_seeds(mindex, index, wl.gs.reads, wl.gs.read_off, keep_on_device=stay)
            t2 = time.perf_counter()
            if stay:
                seeded = int(seed_off[-1]); step_seed_off[0] = seed_off
            else:
                gs = capi.GaplessSet(wl.gs.reads, wl.gs.read_off, seeds, seed_off, node_cap=len(seeds) * 16, mism_cap=len(seeds) * 12)
                step_seed_off[0] = gs.seed_off
            if timing is not None:
                timing["minimizer_seeds"] = timing.get("minimizer_seeds", 0.0) + t2 - t1
                timing["minimizer_seeds (device)"] = timing.get("minimizer_seeds (device)", 0.0) + eng.minimizer_last_ms() * 1e-3
                timing["clusters assembled (host)"] = timing.get("clusters assembled (host)", 0.0) + time.perf_counter() - t2
        if device_tails:
            out = pipeline.align_stage_device(eng, index, gs, timing=timing, seeded=seeded, aligned=with_alignments)
        else:
            out = stage(eng, index, olen, gs, timing=timing, seeded=seeded) if seeded is not None else stage(eng, index, olen, gs, timing=timing)
        if "forest" in out:
            out["forest"].close()
        out["gs"] = gs
        return out

    for _ in range(max(1, args.warmup)):
        out = one_step()
    barrier()
    t0 = time.perf_counter()
    timing = {}
    for _ in range(args.steps):
        out = one_step(timing)
    barrier()
    elapsed = time.perf_counter() - t0
    if from_reads:
        seeds_per_read = float(np.diff(step_seed_off[0]).mean())
    # Steady state of a caller that keeps two batches in flight (vg's many-threads-one-process pattern): two engine contexts, one host
    # thread each, the same stage on alternate batches — one context's seeding (VALU-bound) runs under the other's extension (issue- and
    # latency-bound) and its copies under the other's kernels.  Reported beside `value`, which stays the one-batch-at-a-time rate.
    two_contexts = None
    if from_reads and stay and device_tails and world == 1 and not os.environ.get("VGAMD_GIRAFFE_ONE_CONTEXT"):
        import threading
        eng_b = capi.Engine(capi.Scoring.simple(1, 4, 6, 1, 5)); eng_b.reuse_outputs = True
        index_b = eng_b.haplo_index(wl.nodes, wl.threads); mindex_b = eng_b.minimizer_index(wl.nodes, wl.threads)
        lanes = [(eng, index, mindex), (eng_b, index_b, mindex_b)]

        def lane_steps(lane, count, sink):
            e, hi, mi = lane
            for _ in range(count):
                so, _, _ = e.minimizer_seeds(mi, hi, wl.gs.reads, wl.gs.read_off, keep_on_device=True)
                sink.append(pipeline.align_stage_device(e, hi, wl.gs, seeded=int(so[-1]), aligned=with_alignments))
        for lane in lanes:
            lane_steps(lane, 1, [])                                       # warm both contexts
        per_lane = max(1, (args.steps + 1) // 2)
        sinks = [[], []]
        barrier()
        t1 = time.perf_counter()
        th = [threading.Thread(target=lane_steps, args=(lanes[k], per_lane, sinks[k])) for k in range(2)]
        for t in th: t.start()
        for t in th: t.join()
        barrier()
        t_two = time.perf_counter() - t1
        same = all((s[-1]["read_score"] == out["read_score"]).all() for s in sinks)
        two_contexts = {"batches": 2 * per_lane, "ms_per_batch": 1e3 * t_two / (2 * per_lane), "reads_per_s": n * 2 * per_lane / t_two, "read_scores_equal_to_the_serial_run": bool(same)}
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=RDEV)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    res = out["res"]; n_tails = out["stats"][0] if "stats" in out else len(out["tails"]["problems"])
    cpu = parity = None
    if rank == 0 and world == 1 and not args.no_cpu:
        ora = capi.Engine(capi.Scoring.simple(1, 4, 6, 1, 5), lib=os.path.join(ROOT, "oracle", "libvgoracle.so"))
        cores = shard.usable_cpus(); ora.lib.vgo_set_threads(cores)
        k = min(n, args.cpu_sample or 200_000)
        oidx = ora.haplo_index(wl.nodes, wl.threads)
        t1 = time.perf_counter()
        if from_reads:
            omi = ora.minimizer_index(wl.nodes, wl.threads); t1 = time.perf_counter()
            so, sd, _ = ora.minimizer_seeds(omi, oidx, wl.gs.reads[:wl.gs.read_off[k]], wl.gs.read_off[:k + 1])
            sub = capi.GaplessSet(wl.gs.reads[:wl.gs.read_off[k]], wl.gs.read_off[:k + 1], sd, so, node_cap=len(sd) * 16, mism_cap=len(sd) * 12)
        else:
            sub = capi.GaplessSet(wl.gs.reads[:wl.gs.read_off[k]], wl.gs.read_off[:k + 1], wl.gs.seeds[:wl.gs.seed_off[k]], wl.gs.seed_off[:k + 1])
        o = pipeline.align_stage(ora, oidx, olen, sub); tc = time.perf_counter() - t1
        same = int((o["read_score"] == out["read_score"][:k]).sum())
        cpu = {"value": k / tc, "unit": "reads/s", "cores": cores, "kind": "port",
               "impl": "scalar checkers, not a tuned CPU baseline: the same pipeline over the oracle: vgo_gapless.c (OpenMP over reads), vgo_tail.c (one thread), vgo_xdrop.c (OpenMP over problems)",
               "sample": "the first %d reads of the batch" % k}
        parity = {"checked": k, "identical": same, "what": "per-read best total score (extension + both tails); tests/test_giraffe_stage.py compares every intermediate product"}
        if with_alignments and not from_reads:
            # the tails of the first k reads are a prefix of the right tails and a prefix of the left tails (both in extension order)
            want = pipeline.winning_alignments(o)
            tl, tops = out["tails"], out["tail_ops"]
            ne_k = int(o["res"]["n_ext"].sum())
            mine = [x for x in tl if x["ext"] < ne_k]
            ok = len(mine) == len(want)
            same_aln = 0
            for x, wrow in zip(mine, want):
                ops = tops[x["ops_begin"]:x["ops_begin"] + x["n_ops"]]
                got = (int(x["ext"]), int(x["left"]), int(x["read_begin"]), int(x["read_end"]), int(x["score"]), int(x["first_offset"]), [(int(q["node"]), int(q["op"]), int(q["len"])) for q in ops])
                same_aln += got == wrow
            parity["tail_alignments_checked"] = len(want); parity["tail_alignments_identical"] = same_aln if ok else 0
    if rank == 0:
        open_reads = int((res["full_length"] == 0).sum())
        print(json.dumps({
            "metric": "reads/sec through giraffe's alignment stage (gapless extension; tail forests + pinned X-drop for unresolved clusters)",
            "value": n * world * args.steps / elapsed, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8 / u16", "data": "synthetic",
            "config": {"workload": "1 Mbp variation graph, 8 random haplotype threads, %d x 150 bp reads per GPU from either strand, 1 %% substitutions, %d %% of the reads with one "
                                   "inserted base, 4.0 seeds per read at true positions; GaplessExtender + get_tail_forest + align_pinned(xdrop) semantics, scores 1/4/6/1/5" % (n, int(100 * inserted)),
                       "timed_region": ("per step, from host buffers: vgk_gapless_extend (the extension sets back on the host), then vgk_tail_stage: tails derived, forest, one window per tree, "
                                        "window packing, fill + traceback, best tree per tail and totals, all on the device from what the extension call left in HBM") if device_tails else
                                       ("per step, from host buffers: vgk_gapless_extend (results back on the host), then the host shim's run_tail_stage (vg_amd/host/tail_stage.cpp): "
                                        "tails derived on host threads, vgk_tail_forest, one window per tree, vgk_gssw_pack_windows + run + fetch, totals"), "reads_without_full_length_extension": open_reads, "tails": n_tails,
                       "clusters_from": ("minimizer seeding on the device (k 29, w 11; %.1f seeds per read; index of %d minimizer k-mers built in %.1f s); %s" % (seeds_per_read, mindex.keys, t_index,
                                          "reads and seeds stay in HBM for the extension (vgk_gapless_extend_seeded)" if stay else "seeds via the host")) if from_reads else "seeds given (true positions)",
                       "stage_ms": {k: 1e3 * v / args.steps for k, v in timing.items()},
                       "two_contexts_in_flight": two_contexts,
                       "tail_alignments": ("returned: per tail the best tree's alignment, chosen on the device (vgk_tail_stage_aligned); %d tails, %d ops per step" % (len(out["tails"]), len(out["tail_ops"]))) if with_alignments else "scores only (vgk_tail_stage)",
                       "parallelism": "read-sharded x%d" % world, "device": dev_name, "compute_units": cus, "generation_seconds": t_gen},
            "roofline": {"bound": "hbm", "kernel": "gapless_search_kernel (the stage's largest kernel)", "limiter": "host glue and PCIe round trips between the stages, then memory latency (DESIGN.md §11, §17)",
                         "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None},
            "cpu_baseline": cpu, "parity": parity, "problems_failed": int((res["status"] != 0).sum())}))
    if dist is not None:
        dist.destroy_process_group()


def bench_forest(args, eng, rank, world, dist, torch, dev_name, cus):
    """giraffe's tail path behind the gapless extension (secondary line): per tail a GBWT search state + cut ->
    the tail forest walked on the device (vgk_tail_forest) and left in HBM as a resident graph -> the trees as window problems, packed on
    the device (vgk_gssw_pack_windows) -> left-pinned X-drop fill + traceback.  One step = all of that for the whole batch, from the
    host's problem arrays (20 B + the tail's bases per tail) to results resident in HBM."""
    import numpy as np
    from vg_amd import capi, shard, workloads
    n = args.reads if args.reads else 1_000_000
    OPS_PER = 32
    t0 = time.perf_counter()
    wl = workloads.TailForestWorkload(n, seed=99 + rank)
    t_gen = time.perf_counter() - t0
    index = eng.haplo_index(wl.nodes, wl.threads)

    def barrier():
        _device_sync(torch)
        if dist is not None:
            dist.barrier()
        _device_sync(torch)

    def step(keep=False):
        t = [time.perf_counter()]
        res, forest = eng.tail_forest(index, wl.problems); t.append(time.perf_counter())
        fms = eng.tail_last_ms()
        ws = wl.windows(res); t.append(time.perf_counter())
        b = eng.pack_windows(forest.graph, ws, OPS_PER); t.append(time.perf_counter())
        b.run(); b.sync(); t.append(time.perf_counter())
        out = (res, forest, b, fms, np.diff(t))
        if not keep:
            b.free(); forest.close()
        return out

    for _ in range(max(1, args.warmup)):
        step()
    barrier()
    parts = []; fms = []; fill_ms = []; walk_ms = []
    t0 = time.perf_counter()
    last = None
    for k in range(args.steps):
        if last is not None:
            last[2].free(); last[1].close()
        last = step(keep=True)
        parts.append(last[4]); fms.append(last[3]); fill_ms.append(last[2].kernel_ms(0)); walk_ms.append(last[2].kernel_ms(1))
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=RDEV)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    res, forest, b, _, _ = last
    r, ops = b.fetch()
    n_bad = int((r["status"] != 0).sum() + (res["status"] != 0).sum())
    tree_nodes = int(res["n_nodes"].sum()); tree_bases = int(res["bases"].sum()); cells = b.cells(); alg_fill = b.alg_bytes()
    cpu = parity = None
    if rank == 0 and world == 1 and not args.no_cpu:
        ora = capi.Engine(capi.Scoring.simple(1, 4, 6, 1, 5), lib=os.path.join(ROOT, "oracle", "libvgoracle.so"))
        cores = shard.usable_cpus(); ora.lib.vgo_set_threads(cores)
        k = min(n, args.cpu_sample or 50_000)
        oidx = ora.haplo_index(wl.nodes, wl.threads)
        t1 = time.perf_counter()
        ores, oforest = ora.tail_forest(oidx, wl.problems[:k]); t_walk = time.perf_counter() - t1
        ows = wl.windows(ores, 0, k)
        with ora.pack_windows(oforest.graph, ows, OPS_PER) as ob:
            t1 = time.perf_counter(); ob.run(); t_align = time.perf_counter() - t1
            orr, oops = ob.fetch()
        same = np.ones(k, dtype=bool)
        for f in ("status", "first_node", "n_nodes", "n_trees", "root_trim", "bases"):
            same &= res[f][:k] == ores[f]
        pa, na, la = forest.fetch(); pb, nb, lb = oforest.fetch()
        m = int(ores["n_nodes"].sum())
        forest_same = bool((pa[:m] == pb).all() and (na[:m] == nb).all() and (la[:m] == lb).all())
        for f in ("score", "status", "end_node", "end_offset", "end_read", "first_offset", "n_ops"):
            same &= r[f][:k] == orr[f]
        tot = int(orr["n_ops"].sum())
        ops_same = bool((ops[:tot].view(np.uint64) == oops[:tot].view(np.uint64)).all()) if same.all() else False
        cpu = {"value": k / (t_walk + t_align), "unit": "tails/s", "cores": cores, "kind": "port",
               "impl": "scalar checkers, not a tuned CPU baseline: oracle/vgo_tail.c (dfs_gbwt restated, one thread) + oracle/vgo_xdrop.c (scalar int32 checker, OpenMP over problems)",
               "sample": "the first %d tails: forests %.2f s on one core, alignments %.2f s on %d" % (k, t_walk, t_align, cores)}
        parity = {"checked": k, "identical": int(same.sum()) if (forest_same and ops_same) else int(same.sum()) - 1, "forests_identical": forest_same, "ops_identical": ops_same}
    if rank == 0:
        parts = np.array(parts).mean(axis=0) * 1e3
        # algorithmic bytes of the forest stage: the problem in, per tree node its record's header + edges read once (48 B), (parent, node, length, cut)
        # written and read once (32 B), its bases read and their column-info bytes written (2 B per base), the graph tables (16 B), the result out
        alg_forest = 20 * n + 24 * n + tree_nodes * (48 + 32 + 16) + 2 * tree_bases
        fm = float(np.mean(fms))
        achieved = alg_forest / (fm * 1e-3) / 1e9
        print(json.dumps({
            "metric": "read tails/sec: tail forest (GBWT walk) + left-pinned X-drop alignment against it", "value": n * world * args.steps / elapsed,
            "unit": "tails/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u16", "data": "synthetic",
            "config": {"workload": "1 Mbp variation graph, 8 random haplotype threads; %d tails per GPU (1-121 bp, 1 %% substitutions) starting inside a node of a thread on either "
                                   "strand with the state of all the node's visits; walk distance = tail + longest detectable gap of a 150 bp read; get_tail_forest + "
                                   "align_pinned(xdrop) semantics, scores 1/4/6/1/5" % n,
                       "timed_region": "per step: vgk_tail_forest (problems up, two walks + graph tables on the device, results down), WindowSet on the host, "
                                       "vgk_gssw_pack_windows, fill + traceback kernels; results stay in HBM",
                       "tree_nodes_per_tail": tree_nodes / n, "tree_bases_per_tail": tree_bases / n, "parallelism": "tail-sharded x%d" % world, "device": dev_name, "compute_units": cus,
                       "generation_seconds": t_gen},
            "stage_ms": {"tail_forest_call": float(parts[0]), "tail_forest_device": fm, "window_set_host": float(parts[1]), "pack_windows": float(parts[2]),
                         "fill_and_traceback": float(parts[3]), "fill_kernel": float(np.mean(fill_ms)), "traceback_kernel": float(np.mean(walk_ms))},
            "roofline": {"bound": "hbm", "kernel": "tail_walk_kernel x 2 + forest_flags_kernel + forest_emit_kernel + 4 rocPRIM scans (the forest stage)",
                         "limiter": "memory latency: one lane per tail chases records through a 20 KB stack slab (DESIGN.md §17)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "alg_bytes_per_launch": alg_forest, "avg_launch_ms": fm, "fill_alg_bytes": alg_fill, "gcups_fill": cells / (float(np.mean(fill_ms)) * 1e-3) / 1e9},
            "cpu_baseline": cpu, "parity": parity, "problems_failed": n_bad}))
    b.free(); forest.close()
    if dist is not None:
        dist.destroy_process_group()


def bench_tails(args, eng, rank, world, dist, torch, dev_name, cus):
    """configs[2] at its stated size (secondary line): a chr22-scale SNP + indel graph (50.8 Mbp, ~1.7 M nodes) resident in HBM on both
    strands, 10 M reads with BOTH tails (1-121 bp) aligned left-pinned X-drop (right tails against the reverse-complement strand, as
    giraffe does) — 20 M window problems, packed on the device in batches of <= 5 M that all stay resident.  One step = the fill +
    traceback kernels of every batch."""
    import numpy as np
    from vg_amd import capi, shard, workloads
    n_reads = args.reads if args.reads else 10_000_000
    OPS_PER = 32
    g = workloads.VariationGraph()
    strands = [(g, 7 + 2 * rank), (g.reverse_complement(), 8 + 2 * rank)]
    t0 = time.perf_counter()
    batches, sets = [], []
    for graph, seed in strands:
        dg = eng.graph(*graph.arrays())
        tails = workloads.GraphTails(graph, n_reads, seed=seed)
        for lo in range(0, n_reads, 5_000_000):
            ws = tails.subset(min(5_000_000, n_reads - lo), lo)
            sets.append((graph, tails, lo, ws, dg))
    t_gen = time.perf_counter() - t0
    t0 = time.perf_counter()
    for graph, tails, lo, ws, dg in sets:
        batches.append(eng.pack_windows(dg, ws, OPS_PER))
    t_pack = time.perf_counter() - t0
    n_tails = sum(b.ps.n for b in batches)

    def barrier():
        _device_sync(torch)
        if dist is not None:
            dist.barrier()
        _device_sync(torch)

    for _ in range(args.warmup):
        for b in batches:
            b.run()
    for b in batches:
        b.sync()
    barrier()
    fill_ms, walk_ms, launches = [], [], 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        for b in batches:
            b.run()
            fill_ms.append(b.kernel_ms(0)); walk_ms.append(b.kernel_ms(1)); launches += max(1, int(round(b.kernel_ms(2))))
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=RDEV)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    tf = time.perf_counter()
    outs = [b.fetch() for b in batches]
    t_fetch = time.perf_counter() - tf
    alg_bytes = sum(b.alg_bytes() for b in batches); cells = sum(b.cells() for b in batches); dev_bytes = sum(b.device_bytes() for b in batches)
    n_bad = int(sum((r["status"] != 0).sum() for r, _ in outs))
    cpu = parity = None
    if rank == 0 and world == 1 and not args.no_cpu:
        ora = capi.Engine(capi.Scoring.simple(1, 4, 6, 1, 5), lib=os.path.join(ROOT, "oracle", "libvgoracle.so"))
        cores = shard.usable_cpus(); ora.lib.vgo_set_threads(cores)
        checked = same = 0; tc = 0.0; k = args.cpu_sample or 100_000
        for (graph, tails, lo, ws, dg), (res, ops) in zip((sets[0], sets[len(sets) // 2]), (outs[0], outs[len(sets) // 2])):      # one batch of either strand
            sub = tails.subset(min(k, ws.n), lo)
            og = ora.graph(*graph.arrays())
            with ora.pack_windows(og, sub, OPS_PER) as ob:
                t1 = time.perf_counter(); ob.run(); tc += time.perf_counter() - t1
                ores, oops = ob.fetch()
            hdr = np.ones(sub.n, dtype=bool)
            for f in ("score", "status", "end_node", "end_offset", "end_read", "first_offset", "n_ops"):
                hdr &= res[f][:sub.n] == ores[f]
            good = int(hdr.sum())
            if hdr.all():
                tot = int(ores["n_ops"].sum())
                bad = ops[:tot].view(np.uint64) != oops[:tot].view(np.uint64)
                if bad.any():
                    good = sub.n - len(np.unique(np.repeat(np.arange(sub.n), ores["n_ops"])[bad]))
            checked += sub.n; same += good
        cpu = {"value": checked / tc, "unit": "alignments/s", "cores": cores, "kind": "port", "impl": "scalar int32 checker (oracle/vgo_xdrop.c), OpenMP over problems",
               "sample": "%d tails of either strand, the induced subgraphs built by the oracle itself" % (checked // 2)}
        parity = {"checked": checked, "identical": same}
    if rank == 0:
        fill_step = sum(fill_ms) / args.steps; fill_avg = sum(fill_ms) / max(launches, 1)
        achieved = (alg_bytes * args.steps / max(launches, 1)) / (fill_avg * 1e-3) / 1e9
        print(json.dumps({
            "metric": "tail alignments/sec (pinned X-drop, 1-121 bp, both tails of every read)", "value": n_tails * world * args.steps / elapsed,
            "unit": "alignments/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u16", "data": "synthetic",
            "config": {"workload": "configs[2]: chr22-scale graph (50 818 468 bp, 41 %% GC, SNPs 1/1000, indels 1-20 bp 1/10 000, <= 32 bp nodes: %d nodes), "
                                   "%d reads per GPU with both tails (1-121 bp, 1 %% substitutions, walks of two haplotypes), left-pinned X-drop (dozeu "
                                   "semantics; right tails on the reverse-complement strand) + traceback, windows of the resident graph, scores 1/4/6/1/5" % (g.n_nodes, n_reads),
                       "tails_per_gpu_per_step": n_tails, "batches_resident": len(batches), "parallelism": "read-sharded x%d" % world,
                       "device": dev_name, "compute_units": cus, "generation_seconds": t_gen},
            "roofline": {"bound": "valu", "kernel": "gssw_fill_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None, "alg_bytes_per_step": alg_bytes, "fill_launches_per_step": launches // args.steps, "avg_launch_ms": fill_avg,
                         "fill_ms_per_step": fill_step, "traceback_ms_per_step": sum(walk_ms) / args.steps, "gcups_fill": cells / (fill_step * 1e-3) / 1e9},
            "cpu_baseline": cpu, "parity": parity, "problems_failed": n_bad, "hbm_footprint_bytes": dev_bytes,
            "packing": "windows of the resident graph, packed on the device (vgk_gssw_pack_windows)", "pack_seconds": t_pack, "fetch_seconds": t_fetch,
            "end_to_end_from_host_buffers_per_s": n_tails / (t_pack + elapsed / args.steps + t_fetch)}))
    for b in batches:
        b.free()
    if dist is not None:
        dist.destroy_process_group()


def bench_xdrop_band(args, eng, rank, world, dist, torch, dev_name, cus):
    """Row a10 (secondary line): the round-1 tails stand-in (giraffe-style pinned tails on a 2 Mbp variation graph, one explicit graph
    per problem) through vgk_xdrop_band_align — dozeu's band restated [PARITY-UNPINNED] — next to the exact extension of the same
    problems.  Reports the cells the band keeps, the kernel time, and how often the banded score equals the exact one."""
    import ctypes
    import numpy as np
    from vg_amd import capi, shard, workloads
    n = min(args.reads or 200_000, 200_000)
    ps = workloads.TailWorkload(n, seed=77 + rank).ps
    eng.lib.vgk_xdrop_band_last_ms.restype = ctypes.c_double; eng.lib.vgk_xdrop_band_last_ms.argtypes = [ctypes.c_void_p]
    for _ in range(max(args.warmup, 2)):                             # warm the context's cached staging and device buffers, as the other workloads' first calls do
        res, ops, st = eng.xdrop_band_align(ps)
    keep = (res, np.zeros(len(ops) + 1024, dtype=ops.dtype))         # the caller's own output buffers, kept between calls
    eng.xdrop_band_align(ps, out=keep)
    steps = max(args.steps, 1)
    t0 = time.perf_counter()
    for _ in range(steps):                                           # a step = one whole call from host buffers
        res, ops, st = eng.xdrop_band_align(ps, out=keep)
    t_band = (time.perf_counter() - t0) / steps
    k_ms = eng.lib.vgk_xdrop_band_last_ms(eng.h)
    eng.lib.vgk_xdrop_band_last_cells.argtypes = [ctypes.c_void_p]
    cell_form = int(eng.lib.vgk_xdrop_band_last_cells(eng.h)); cell_bytes = 4 if cell_form == 4 else 2
    t0 = time.perf_counter(); eres, eops = eng.align(ps, 48); t_exact = time.perf_counter() - t0
    # band mode against exact mode, alignment by alignment: a tail counts as different when any header field
    # or any CIGAR element differs.  Both op arrays are packed per problem (n_ops elements from ops_begin).
    def same_alignment(ra, oa, rb, ob):
        same = np.ones(len(ra), dtype=bool)
        for f in ("score", "status", "end_node", "end_offset", "end_read", "first_offset", "n_ops"):
            same &= ra[f] == rb[f]
        idx = np.flatnonzero(same & (ra["n_ops"] > 0))
        if len(idx):
            cnt = ra["n_ops"][idx].astype(np.int64)
            owner = np.repeat(np.arange(len(idx)), cnt)
            within = np.arange(int(cnt.sum())) - np.repeat(np.cumsum(cnt) - cnt, cnt)
            a = oa.view(np.uint64)[ra["ops_begin"][idx].astype(np.int64)[owner] + within]
            b = ob.view(np.uint64)[rb["ops_begin"][idx].astype(np.int64)[owner] + within]
            bad = np.unique(owner[a != b])
            same[idx[bad]] = False
        return same
    band_same = same_alignment(res, ops, eres, eops)
    parity = cpu = None
    if not args.no_cpu:
        ora = capi.Engine(capi.Scoring.simple(1, 4, 6, 1, 5), lib=os.path.join(ROOT, "oracle", "libvgoracle.so"))
        ora.lib.vgo_set_threads(shard.usable_cpus())
        k = min(n, args.cpu_sample or 50_000)
        sub = ps.subset(k)
        t0 = time.perf_counter(); ores, oops, ost = ora.xdrop_band_align(sub); tc = time.perf_counter() - t0
        good = np.ones(k, dtype=bool)
        for f in ("score", "status", "end_node", "end_offset", "end_read", "first_offset", "n_ops"):
            good &= res[f][:k] == ores[f]
        tot = int(ores["n_ops"].sum())
        if good.all() and (ops[:tot].view(np.uint64) != oops.view(np.uint64)).any():
            good[:] = False
        parity = {"checked": k, "identical": int(good.sum())}
        cpu = {"value": k / tc, "unit": "alignments/s", "cores": shard.usable_cpus(), "kind": "port", "impl": "scalar int32 checker with the band (oracle/vgo_xdrop.c)", "sample": "first %d problems" % k}
    # the kernel keeps H and E of the cells inside the band (16-bit cells when the call's bounds allow: vgk_xdrop_band_last_cells), in whole
    # 8-row vectors, and two bytes per column for the band's extent: that, the inputs and the ops are its algorithmic bytes
    L_of = np.diff(ps.read_off).astype(np.int64); cols_of = np.diff(ps.seq_off).astype(np.int64)
    alg_bytes = float(2 * cell_bytes * st[0] + 3 * cols_of.sum() + L_of.sum() + 8 * int(res["n_ops"].sum()))
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9
    roofline = {"bound": "hbm", "kernel": ("xdrop_band_pk_kernel16 (+ xdrop_band_pk_kernel for reads over 127 bases): two rows to a register, 16-bit cells" if cell_form == 2 else
                                                "xdrop_band_kernel16 (+ xdrop_band_kernel for reads over 127 bases), %d-byte cells" % cell_bytes) + " + xdrop_band_walk_kernel: fill with end cell, then the tracebacks; one timed region",
                "cell_bytes": cell_bytes, "limiter": "VALU issue: ~235 instructions per wavefront-column of four tails (DESIGN.md \u00a727.8), not bandwidth",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": PMC_BYTES_PER_UNIT["xband"] * n if "xband" in PMC_BYTES_PER_UNIT else None,
                "traffic_source": traffic_source("xband") if "xband" in PMC_BYTES_PER_UNIT else None,
                "alg_bytes_per_launch": alg_bytes, "avg_launch_ms": k_ms}
    print(json.dumps({
        "metric": "tail alignments/sec, X-drop with dozeu's band restated (host-inclusive, second call on a warm context: pack + H2D + fill / end cell / traceback kernel + packed ops back)",
        "value": n / t_band, "unit": "alignments/s", "n_gpus": world, "steps": steps, "warmup": max(args.warmup, 2), "ms_per_step": 1e3 * t_band, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "i32" if cell_form == 4 else "u16", "data": "synthetic",
        "config": {"workload": "round-1 tails stand-in: 2 Mbp variation graph, %d tails of 1-121 bp, left-pinned, one explicit graph per problem; vgk_xdrop_band_align [PARITY-UNPINNED]" % n,
                   "device": dev_name, "compute_units": cus},
        "band": {"cells_in_band": st[0], "cells_of_the_rectangles": st[1], "fraction_kept": st[0] / max(st[1], 1), "fill_kernel_ms": k_ms,
                 "banded_score_equals_exact": float((res["score"] == eres["score"]).mean()), "banded_score_never_higher": bool((res["score"] <= eres["score"]).all()),
                 "tails_whose_band_mode_alignment_differs_from_exact_mode": int((~band_same).sum()), "of_them_with_the_same_score": int((~band_same & (res["score"] == eres["score"])).sum()),
                 "exact_path_same_problems_host_inclusive_per_s": n / t_exact},
        "roofline": roofline, "cpu_baseline": cpu, "parity": parity, "problems_failed": int((res["status"] != 0).sum())}))
    if dist is not None:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=0, help="reads per GPU per step (0 = the configuration's own size: configs[1] 1M reads; tails: 10M reads = 20M tails)")
    ap.add_argument("--tails-per-problem-graphs", action="store_true", help="tails workload: round 1's stand-in (200k tails on a 2 Mbp graph, one explicit graph per problem, host packer)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="reads for the CPU baseline leg (0 = auto, ~15 s)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--host-pack", action="store_true", help="linear workload: one explicit graph per problem, packed on host threads (vgk_gssw_pack) instead of windows of the resident graph packed on the device")
    ap.add_argument("--no-e2e", action="store_true", help="skip the legs that overlap launches — the two-lane steady state and the warm / double-buffered end-to-end legs — so that a profiler's per-kernel averages are each kernel's own")
    ap.add_argument("--no-secondary", action="store_true", help="default (linear, one GPU) run: do not append the `secondary` records — the other kernel families' own bench lines, each run "
                         "as `bench.py --workload X` in a process of its own after the headline has been measured")
    ap.add_argument("--sub-rate", type=float, default=None, help="linear workload: substitution rate of the reads (default: the configuration's 1 %%)")
    ap.add_argument("--indel-rate", type=float, default=None, help="linear workload: indel rate of the reads (default: the configuration's 0.1 %%; 0.05 makes nearly every read miss the "
                         "speculative fill's diagonal-run shortcut: the leg that shows the context's feedback turning the speculation off)")
    ap.add_argument("--workload", choices=["linear", "tails", "banded", "gapless", "wfa", "xband", "forest", "giraffe", "longread", "config2", "paired", "wide"], default="linear",
                    help="linear = BASELINE.json configs[1] (the headline metric); tails = configs[2] stand-in: "
                         "giraffe-style pinned X-drop tail alignments on a variation graph; banded = configs[4] stand-in: "
                         "banded global alignments between chained anchors; gapless = giraffe's first stage: "
                         "haplotype-consistent gapless extension of seeds; wfa = the long-read chaining stage's WFA connects and tails")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be at least 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:          # no launcher around this process: become one
        raise SystemExit(relaunch_on_ranks(args.gpus))

    from vg_amd import shard
    rank, local_rank, world = shard.env_rank()
    if world != args.gpus and args.gpus == 1:
        args.gpus = world                                          # (a launcher without --gpus: its rank count is the answer)
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d ranks (WORLD_SIZE): the two must agree" % (args.gpus, world))
    import torch   # first, so its bundled HIP runtime is the one the engine library binds to
    dist = None
    # VGAMD_BENCH_ONE_DEVICE=1: a functional check of the N > 1 code path on a box with ONE GPU (every rank on device 0, gloo for
    # the barrier / max-reduce, since RCCL refuses two ranks on one device).  Not a measurement.
    one_device = world > 1 and os.environ.get("VGAMD_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local_rank = 0
    if world > 1 and one_device:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        _set_device(torch, 0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    elif world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        _set_device(torch, local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        _set_device(torch, local_rank)

    import numpy as np
    from vg_amd import capi, workloads
    if world > 1:      # N ranks share one host: each gets its share of the packing / unpacking threads
        os.environ["VGAMD_HOST_THREADS"] = str(shard.host_threads_per_rank(world))
        try:           # ... and a leg that needs more host threads per rank than that share must not print a number at all
            shard.check_host_thread_budget(args.workload, world)
        except shard.HostThreadBudgetError as e:
            raise SystemExit("bench.py: " + str(e))

    eng_lib = os.environ.get("VGAMD_ENGINE_LIB") or os.path.join(ROOT, "vg_amd", "libvgamd.so")     # (the override is for kernel experiments: another build of the same library)
    if not os.path.exists(eng_lib):
        raise SystemExit("vg_amd/libvgamd.so missing: run `python -c 'import __graft_entry__ as g; g.build()'` (no CPU fallback)")
    eng = capi.Engine(capi.Scoring.simple(1, 4, 6, 1, 5), device=local_rank, lib=eng_lib)
    dev_name, cus, hbm = eng.device_info()
    if os.environ.get("VGAMD_WFA_POINT_BUDGET"):
        eng.wfa_set_point_budget(int(os.environ["VGAMD_WFA_POINT_BUDGET"]))

    if args.workload == "wide":
        return bench_wide(args, eng, rank, world, dist, torch, dev_name, cus)
    if args.workload == "longread":
        return bench_longread(args, eng, rank, world, dist, torch, dev_name, cus)
    if args.workload == "config2":
        return bench_config2(args, eng, rank, world, dist, torch, dev_name, cus)
    if args.workload == "paired":
        return bench_paired(args, eng, rank, world, dist, torch, dev_name, cus)
    if args.workload == "giraffe":
        return bench_giraffe(args, eng, rank, world, dist, torch, dev_name, cus)
    if args.workload == "forest":
        return bench_forest(args, eng, rank, world, dist, torch, dev_name, cus)
    if args.workload == "xband":
        return bench_xdrop_band(args, eng, rank, world, dist, torch, dev_name, cus)
    if args.workload == "tails" and not args.tails_per_problem_graphs:
        return bench_tails(args, eng, rank, world, dist, torch, dev_name, cus)
    if not args.reads:
        args.reads = 1_000_000
    if args.workload == "banded":
        return bench_banded(args, eng, rank, world, dist, torch, dev_name, cus)
    if args.workload == "wfa":
        return bench_wfa(args, eng, rank, world, dist, torch, dev_name, cus)
    if args.workload == "gapless":
        return bench_gapless(args, eng, rank, world, dist, torch, dev_name, cus)

    # ONE read stream of reads-per-GPU x N reads (weak scaling: the work per GPU is fixed), cut into contiguous shards with the two
    # reads of a pair kept together (shard.shard_range); every rank materialises its own shard only
    if args.workload == "tails":
        n_tails = min(args.reads, 200_000)          # per-problem graphs are built in Python: keep generation short
        wl = workloads.TailWorkload(n_tails, seed=77 + rank).ps
        args.reads = n_tails
    else:
        lo, hi = shard.shard_range(args.reads * world, rank, world, group=2)
        rates = {}
        if args.sub_rate is not None:
            rates["sub_rate"] = args.sub_rate
        if args.indel_rate is not None:
            rates["indel_rate"] = args.indel_rate
        wl = workloads.LinearWorkload(hi - lo, stream_begin=lo, **rates)
    OPS_PER = 48
    # configs[1]: the reference graph is resident in HBM once (vgk_graph_create) and every read is a window of it, packed by
    # kernels (vgk_gssw_pack_windows); --host-pack (and the tails workload) hand over one explicit graph per problem instead
    windows = args.workload == "linear" and not args.host_pack
    if windows:
        graph = eng.graph(*wl.graph_arrays()); ws = wl.windows()
        pack = lambda: eng.pack_windows(graph, ws, OPS_PER)
    else:
        pack = lambda: eng.pack(wl, OPS_PER)
    t0 = time.time()
    batch = pack()                         # packing + H2D: inputs resident in HBM from here on
    t_pack = time.time() - t0

    def barrier():
        _device_sync(torch)
        if dist is not None:
            dist.barrier()
        _device_sync(torch)

    for _ in range(args.warmup):
        batch.run()
    batch.sync()
    barrier()
    fill_ms, walk_ms, refill_ms, speculated = [], [], [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        batch.run()                        # fill + traceback kernels, async on the engine stream
        speculated.append(batch.speculated())
        # per-launch kernel durations from HIP events on the launch stream (synchronises this step)
        fill_ms.append(batch.kernel_ms(0)); walk_ms.append(batch.kernel_ms(1)); refill_ms.append(batch.kernel_ms(3))
    n_launch = max(1, int(round(batch.kernel_ms(2))))   # fill launches per step (the batch runs as a chunk pipeline)
    batch.sync()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=RDEV)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    one_stream = {"ms_per_step": 1e3 * elapsed / args.steps, "fill_ms": sum(fill_ms) / len(fill_ms), "traceback_ms": sum(walk_ms) / len(walk_ms),
                  "reads_per_s": args.reads * world * args.steps / elapsed,
                  "second_fill_ms": sum(refill_ms) / len(refill_ms),
                  "speculated_steps": int(sum(speculated)), "speculation": eng.speculation_state(), "step_ms_fill_plus_tail": [round(a + b, 3) for a, b in zip(fill_ms, walk_ms)],
                  "speculative_fill": ("on: the first fill builds no traceback codes; fill_ms is that launch; traceback_ms holds everything behind it — gssw_walk_first_kernel, then (second_fill_ms) "
                                       "the layout and the second fill, with codes, of the reads whose alignment is not one diagonal run, then gssw_walk_missed_kernel (DESIGN.md \u00a727.12)"
                                       if sum(refill_ms) > 0 else "off"),
                  "traceback_kernels": ("one: gssw_walk_kernel (VGAMD_WALK_ONE_PASS)" if os.environ.get("VGAMD_WALK_ONE_PASS") or args.workload == "tails" else
                                        "two: gssw_walk_first_kernel (alignments that are one diagonal run, settled from the read's and the columns' bytes) + gssw_walk_missed_kernel (the rest, by their codes) — DESIGN.md \u00a727.11")}
    # Steady state of a streaming caller: consecutive batches alternate between the context's two launch lanes (streams), so the
    # traceback of one batch — bound by memory latency — runs under the fill of the next — bound by VALU issue.  Two batches are
    # resident (the same reads packed twice); step i runs batch i mod 2; exactly K steps between the barriers.
    two_lane = None
    if (windows or args.workload == "tails") and not args.no_e2e:      # (profiling runs skip the legs that overlap launches)
        batch2 = pack()
        if batch2.lane() != batch.lane():
            pair = (batch, batch2)
            for k in range(max(args.warmup, 2)):
                pair[k & 1].run()
            batch.sync(); batch2.sync()
            barrier()
            t0 = time.perf_counter()
            for k in range(args.steps):
                pair[k & 1].run()
            batch.sync(); batch2.sync()
            barrier()
            e2 = time.perf_counter() - t0
            if dist is not None:
                t = torch.tensor([e2], dtype=torch.float64, device=RDEV)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                e2 = float(t.item())
            # the last launch on either lane, timed by HIP events on its own stream while the other lane was busy
            two_lane = {"ms_per_step": 1e3 * e2 / args.steps, "reads_per_s": args.reads * world * args.steps / e2,
                        "fill_ms_under_overlap": [pair[0].kernel_ms(0), pair[1].kernel_ms(0)], "traceback_ms_under_overlap": [pair[0].kernel_ms(1), pair[1].kernel_ms(1)]}
        batch2.free()

    # results of the last step: parity spot-check against the oracle + exact algorithmic bytes
    tf = time.perf_counter(); res, ops = batch.fetch(); t_fetch = time.perf_counter() - tf
    alg_bytes = batch.alg_bytes()
    cells = batch.cells()
    dev_bytes = batch.device_bytes()
    wave_steps = batch.wave_steps()
    n_bad = int((res["status"] != 0).sum())

    cpu = None
    parity = None
    if rank == 0 and world == 1 and not args.no_cpu:      # the CPU leg (checker + baseline) runs at N = 1 only
        ora_lib = os.path.join(ROOT, "oracle", "libvgoracle.so")
        ora = capi.Engine(capi.Scoring.simple(1, 4, 6, 1, 5), lib=ora_lib)     # the checker / CPU baseline leg
        cores = shard.usable_cpus()          # the affinity mask cut by the cgroup quota: what the CPU leg can really use
        ora.lib.vgo_set_threads(cores)
        k = args.cpu_sample
        if k <= 0:
            probe = wl.subset(min(args.reads, 64 * cores))
            with ora.pack(probe, OPS_PER) as pb:
                tp = time.perf_counter(); pb.run(); tp = time.perf_counter() - tp
            k = int(min(args.reads, max(probe.n, 15.0 * probe.n / max(tp, 1e-6))))
        sample = wl.subset(k)
        with ora.pack(sample, OPS_PER) as ob:
            tc = time.perf_counter(); ob.run(); tc = time.perf_counter() - tc
            ores, oops = ob.fetch()
        cpu = {"value": k / tc, "unit": "reads/s", "cores": cores, "kind": "port", "impl": "scalar int32 checker",
               "sample": "first %d problems of the same batch, oracle/vgo_%s.c scalar int32 DP + traceback, OpenMP over reads" % (k, "xdrop" if args.workload == "tails" else "gssw")}

        def identical(res_a, ops_a, res_b, ops_b, m):
            """vectorised bit-exact comparison (score, status, end cell, first offset, every CIGAR element) of the first m problems"""
            hdr = np.ones(m, dtype=bool)
            for f in ("score", "status", "end_node", "end_offset", "end_read", "first_offset", "n_ops"):
                hdr &= res_a[f][:m] == res_b[f][:m]
            same = int(hdr.sum())
            if hdr.all():
                tot = int(res_b["n_ops"][:m].sum())
                bad_ops = ops_a[:tot].view(np.uint64) != ops_b[:tot].view(np.uint64)
                if bad_ops.any():
                    owner = np.repeat(np.arange(m), res_b["n_ops"][:m])
                    same = m - len(np.unique(owner[bad_ops]))
            return same
        parity = {"checked": k, "identical": identical(res, ops, ores, oops, k)}
        # the CPU baseline proper (gssw modes): the SIMD restatement, timed on the same reads, and only quoted when every one of its
        # results equals the engine's (which the checker has just vouched for on its sample)
        import ctypes
        ora.lib.vgo_gssw_run_fast.argtypes = [ctypes.c_void_p]
        if args.workload == "linear" and ora.lib.vgo_gssw_fast_supported():
            kf = min(args.reads, 1_000_000)
            fs = wl.subset(kf)
            with ora.pack(fs, OPS_PER) as fb:
                ora.lib.vgo_gssw_run_fast(fb.h)                                  # warm: per-thread arenas sized, pages faulted in
                tfast = time.perf_counter(); rcf = ora.lib.vgo_gssw_run_fast(fb.h); tfast = time.perf_counter() - tfast
                fres, fops = fb.fetch()
            same_fast = identical(res, ops, fres, fops, kf) if rcf == 0 else 0
            model = "unknown"
            try:
                model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
            except Exception:
                pass
            if same_fast == kf:
                cpu = {"value": kf / tfast, "unit": "reads/s", "cores": cores, "kind": "port",
                       "impl": "restated CPU, SIMD int16 (oracle/vgo_gssw_fast.c: AVX2 rows, per-thread arenas, 1 B/cell traceback, OpenMP dynamic)",
                       "cpu_model": model, "host_hw_threads": os.cpu_count(), "cores_note": "cores = CPUs this container may use (affinity and cgroup quota), one OpenMP thread each",
                       "gcups": fs.cells() / tfast / 1e9, "gcups_per_core": fs.cells() / tfast / 1e9 / cores,
                       "sample": "%d problems of the same batch (DP + traceback, packing excluded), all %d identical to the engine's results" % (kf, kf),
                       "scalar_checker_reads_per_s": k / tc}
            else:
                cpu["fast_path_mismatches"] = kf - same_fast
    batch.free()
    # steady state from host buffers: three more batches, each packed, run once and fetched in turn on the warm context
    # (page-locked staging and device arenas are reused; nothing overlaps — pack, kernels and fetch are serial here)
    t_warm = t_pipe = None
    if dist is not None:
        barrier()                                  # all ranks start their streaming legs together: they share the host
    if not args.no_e2e:        # every rank streams its shard from host buffers; the slowest rank sets the rate
        def touched(a):                            # a second set of output arrays, pages already faulted in
            z = np.empty_like(a); z.view(np.uint8)[:] = 0
            return z
        out_bufs = [(res, ops), (touched(res), touched(ops.base if ops.base is not None else ops))]   # a streaming caller keeps its output arrays
        for _ in range(2):                         # pools and staging of the context settle
            with pack() as wb:
                wb.run(); wb.fetch(into=out_bufs[0])
        tw = time.perf_counter()
        for _ in range(5):
            with pack() as wb:
                wb.run(); wb.fetch(into=out_bufs[0])
        t_warm = (time.perf_counter() - tw) / 5
        # the same with the next batch packed on a second host thread while this one runs and is fetched (double buffering: what a
        # caller that streams reads does; the C ABI's pack / run / fetch split exists for it)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(1) as ex:
            nxt = ex.submit(pack)
            prev = None
            for k in range(13):                    # the first four fill the pipeline (third set of device arenas, staging buffers)
                if k == 5:
                    tp = time.perf_counter()       # (batch 4 is in flight: 8 batches complete between here and the end)
                pb = None
                if k < 12:
                    pb = nxt.result()
                    if k + 1 < 12:
                        nxt = ex.submit(pack)
                    pb.run()                       # queued behind the previous batch's kernels: the GPU does not idle while that one is fetched
                if prev is not None:
                    prev.fetch(into=out_bufs[k & 1]); prev.free()
                prev = pb
            t_pipe = (time.perf_counter() - tp) / 8

    if dist is not None and t_warm is not None:
        barrier()
        t = torch.tensor([t_warm, t_pipe], dtype=torch.float64, device=RDEV)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        t_warm, t_pipe = float(t[0].item()), float(t[1].item())
    if rank == 0:
        total_reads = args.reads * world * args.steps
        value = total_reads / elapsed
        ms_per_step = 1e3 * elapsed / args.steps
        timed_region = "K runs of one resident batch on one stream: fill kernel, then traceback kernel"
        # `value` stays the one-stream loop: its per-launch durations are each kernel's own (the roofline below needs that).  The
        # two-lane steady state (config.two_lanes) and the streaming legs from host buffers (end_to_end_*) run fills of consecutive
        # batches side by side, where a launch's wall time is no longer its own.
        fill_step = (sum(fill_ms) / len(fill_ms)) or 1e-9          # all fill launches of one step (the emulated engine of the functional check reports no kernel time)
        fill_avg = fill_step / n_launch                  # average duration of one fill launch
        # A speculative batch fills twice: every read without traceback codes, then — inside the traceback tail — the reads that need codes
        # again with them.  The algorithmic bytes (SURVEY \u00a78(d): a byte of traceback per cell among them) are the job's, so they are priced
        # against BOTH fills' time; `avg_launch_ms` stays the first launch's own duration, `second_fill_ms` stands beside it.
        refill_step = sum(refill_ms) / len(refill_ms)
        achieved = (alg_bytes / n_launch) / ((fill_avg + refill_step / n_launch) * 1e-3) / 1e9
        tails = args.workload == "tails"
        # VALU-issue model: a wavefront step of the 19-rows-per-lane x8 build issues 259 half-rate (4 cycles per
        # wave64 instruction per SIMD) + 94 full-rate (2 cycles) instructions in its hot block and ~75 more around it
        valu = None
        if not tails and wave_steps and refill_step > 0:
            valu = {"note": "the issue model below was made for the fill WITH traceback codes (428 instructions per wavefront-step); a speculative batch's first fill runs the recurrence alone — "
                            "its measured rate against that one: fill_variant_*.json in profiles/r04 (12.7 vs 19.1 ms per million reads)", "wave_steps": wave_steps}
        elif not tails and wave_steps:
            cyc = 259 * 4 + 94 * 2 + 75 * 4
            valu = {"wave_steps": wave_steps, "issue_cycles_per_step_model": cyc, "simds": 4 * cus, "clock_ghz": 2.4,
                    "frac_of_issue_peak": wave_steps * cyc / (4 * max(cus, 1) * 2.4e9 * fill_step * 1e-3),
                    "source": "instruction mix from the ISA listing of gssw_fill_kernel<19,true> (DESIGN.md), issue costs from tools/valu_rate.hip"}
        out = {
            "metric": "tail alignments/sec (pinned X-drop, 1-121 bp)" if tails else "reads/sec aligned (150 bp)",
            "value": value, "unit": "alignments/s" if tails else "reads/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u16", "data": "synthetic",
            "config": {"workload": ("configs[2] stand-in: 2 Mbp variation graph (SNP + insertion bubbles), %d tails of 1-121 bp "
                                    "per GPU, left-pinned X-drop (dozeu semantics) + traceback, scores 1/4/6/1/5" % args.reads) if tails else
                                   ("configs[1]: linear 1 Mbp graph (32 bp nodes), %d x 150 bp reads per GPU, "
                                    "384-416 bp windows, gssw LOCAL + traceback, scores 1/4/6/1/5" % args.reads),
                       "read_error_rates": {"substitutions": 0.01 if args.sub_rate is None else args.sub_rate, "indels": 0.001 if args.indel_rate is None else args.indel_rate,
                                            "note": "the configuration's own unless --sub-rate / --indel-rate were given (then this is NOT the headline configuration)"},
                       "timed_region": timed_region, "one_stream": one_stream, "two_lanes": two_lane,
                       "reads_per_gpu_per_step": args.reads, "parallelism": "read-sharded x%d" % world,
                       "device": dev_name, "compute_units": cus},
            # `achieved / peak / frac` price the kernel's ALGORITHMIC bytes against the HBM peak, as the contract asks; the kernel's
            # real limiter is VALU issue (PMC: profiles/), so `bound` says so and `valu` holds the issue-rate model beside it
            "roofline": {"bound": "valu", "kernel": "gssw_fill_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "valu": valu,
                         "traffic": None if tails else PMC_BYTES_PER_UNIT["linear"] * args.reads / n_launch,
                         "traffic_source": None if tails else traffic_source("linear"),
                         "alg_bytes_per_launch": alg_bytes / n_launch, "avg_launch_ms": fill_avg, "second_fill_ms": refill_step,
                         "achieved_over": "both fills of a speculative batch (avg_launch_ms + second_fill_ms)" if refill_step > 0 else "the fill launch",
                         "launches_per_step": n_launch,
                         "traceback_tail_ms": sum(walk_ms) / len(walk_ms),
                         "gcups_fill": cells / (fill_step * 1e-3) / 1e9},
            "cpu_baseline": cpu,
            "parity": parity,
            "problems_failed": n_bad,
            "hbm_footprint_bytes": dev_bytes,
            "packing": "windows of the resident graph, packed on the device (vgk_gssw_pack_windows)" if windows else "one graph per problem, packed on host threads (vgk_gssw_pack)",
            "pack_seconds": t_pack, "fetch_seconds": t_fetch,
            # one batch from host buffers: pack (validate + encode + H2D) + one run + fetch (D2H of results and CIGAR ops)
            "end_to_end_from_host_buffers_per_s": args.reads / (t_pack + elapsed / args.steps + t_fetch),
            # whole-job rates from host buffers (all ranks' reads / the slowest rank's time per batch): pack + kernels + fetch in turn,
            # and with the next batch packed and queued while this one runs and is fetched
            "end_to_end_warm_per_s": args.reads * world / t_warm if t_warm else None,
            "end_to_end_double_buffered_per_s": args.reads * world / t_pipe if t_pipe else None,
            "host_threads_per_rank": int(os.environ.get("VGAMD_HOST_THREADS", "0")) or min(shard.usable_cpus(), 48),
        }
        # The LAST stdout line is the headline alone and short (round 5's driver record lost a 24 KB line to its bounded tail): the
        # step's breakdown and every secondary record go out as earlier `[detail] ...` / `[secondary] ...` lines (not starting with
        # "{", so the headline is the only JSON line of the run) and into bench_secondary.json beside this file.
        detail = {"workload": args.workload, "one_stream": out["config"].pop("one_stream"), "two_lanes": out["config"].pop("two_lanes"),
                  "valu": out["roofline"].pop("valu"), "read_error_rates_note": out["config"]["read_error_rates"].pop("note"),
                  "traffic_source": out["roofline"].pop("traffic_source"), "cpu_baseline_notes": {k: cpu.pop(k) for k in ("cores_note",) if cpu and k in cpu}}
        print("[detail] " + json.dumps(detail), flush=True)
        side = {"headline_detail": detail, "secondary": []}
        if world == 1 and args.workload == "linear" and not args.no_secondary and not args.no_cpu and os.environ.get("VGAMD_BENCH_SECONDARY", "1") != "0":
            if windows:
                graph.close()
            eng.close()                                        # the headline is measured: its HBM goes back before the other legs start
            for rec in secondary_records():
                side["secondary"].append(rec)
                print("[secondary] " + json.dumps(rec), flush=True)
            out["secondary_file"] = "bench_secondary.json (%d records; also the `[secondary]` lines above)" % len(side["secondary"])
        if args.workload == "linear":
            try:
                with open(os.path.join(ROOT, os.environ.get("VGAMD_BENCH_SIDE_FILE", "bench_secondary.json")), "w") as f:
                    json.dump(side, f, indent=1)
            except OSError:
                pass                                           # (a read-only tree: the lines above still carry every record)
        line = json.dumps(out)
        assert len(line) < 4096, "bench.py: the headline line must stay under 4 KB (%d)" % len(line)
        print(line, flush=True)
    if dist is not None:
        dist.destroy_process_group()


# The other kernel families' bench lines, appended to the default run's JSON line as `secondary` (the headline's keys and meaning are
# untouched: it has been measured and its engine closed before these start).  Each is `bench.py --workload X` in a process of its own,
# at a size that keeps the whole default run within a few minutes; a record keeps the line's metric, value, roofline, cpu_baseline and
# parity.  A leg that fails or overruns its time limit leaves {"workload", "error"} — never a missing headline.
SECONDARY = [
    ("config2", ["--reads", "8000000", "--steps", "3", "--warmup", "1", "--cpu-sample", "1000000"], 420),
    ("paired", ["--steps", "5", "--warmup", "2", "--cpu-sample", "200000"], 360),
    ("gapless", ["--steps", "5", "--warmup", "2"], 90),
    ("xband", ["--steps", "5", "--warmup", "2"], 90),
    ("banded", ["--reads", "100000", "--steps", "5", "--warmup", "2"], 90),
    ("wfa", ["--reads", "500000", "--steps", "5", "--warmup", "2"], 90),
    ("wide", ["--steps", "3", "--warmup", "1"], 150),
    # (last: its two contexts hold 31 GB of WFA tables; the leg that followed it in one run of round 5 — paired — measured a third slower than in every run of its own)
    ("longread", ["--steps", "3", "--warmup", "1"], 420),
]


def secondary_records():
    """yields one record per leg as soon as that leg has finished"""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    legs = json.loads(os.environ["VGAMD_BENCH_SECONDARY_LEGS"]) if os.environ.get("VGAMD_BENCH_SECONDARY_LEGS") else SECONDARY   # (the override: the CPU test's small legs)
    for name, extra, limit in legs:
        t0 = time.perf_counter()
        rec = {"workload": name}
        try:
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--gpus", "1", "--workload", name] + extra, env=env, cwd=ROOT,
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=limit, text=True)
            line = [l for l in p.stdout.splitlines() if l.startswith("{")]
            if p.returncode != 0 or not line:
                rec["error"] = "rc %d: %s" % (p.returncode, (p.stderr or "").strip()[-300:])
            else:
                d = json.loads(line[-1])
                cfg = d.get("config") or {}
                rec.update({k: d.get(k) for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype", "roofline", "roofline_minimizer", "cpu_baseline", "parity", "problems_failed") if k in d})
                rec["config"] = {k: cfg.get(k) for k in ("workload", "timed_region", "ms_per_batch", "kernel_ms_per_batch", "stage_ms_per_batch", "one_context", "read_buffers", "policies", "index_seconds", "setup_seconds", "hbm_bytes", "stitch_device_ms", "links") if k in cfg}
        except subprocess.TimeoutExpired:
            rec["error"] = "time limit of %d s" % limit
        except Exception as e:                                # (a leg must never take the headline down)
            rec["error"] = "%s: %s" % (type(e).__name__, e)
        rec["wall_s"] = round(time.perf_counter() - t0, 1)
        yield rec


if __name__ == "__main__":
    main()
