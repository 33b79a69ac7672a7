"""Bounded probe of the WFA kernels on a GPU box: each stage prints as it goes (flushed), so a hang shows where.  Run each stage under `timeout`."""
import sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import util
from vg_amd import capi
import test_wfa as T

stage = sys.argv[1]
t0 = time.time()
def say(*a):
    print("[%6.1fs]" % (time.time() - t0), *a, flush=True)

if stage == "golden":
    eng = capi.Engine()
    say("engine up")
    fx = util.load_golden("ref_wfa_extender.json")
    n = 0
    for case in fx["cases"]:
        nodes, threads, index_of = T.graph_tables(case["graph"])
        idx = eng.haplo_index(nodes, threads)
        res, paths, edits = eng.wfa_extend(idx, [T.problem_of(case, index_of)], case["error_model"])
        n += 1
        if n % 10 == 0:
            say("golden", n, case["name"][:40], int(res["status"][0]), int(res["ok"][0]))
    say("golden done", n)
elif stage == "random":
    ora = capi.Engine(lib=util.ORACLE_LIB); eng = capi.Engine()
    for s in range(400, 400 + int(sys.argv[2])):
        rng = np.random.default_rng(s)
        nodes, threads, problems = T.random_wfa_case(rng, int(sys.argv[3]))
        model = T.MODELS[s % len(T.MODELS)]
        a = ora.wfa_extend(ora.haplo_index(nodes, threads), problems, model)
        say("seed", s, "oracle done; engine ...")
        b = eng.wfa_extend(eng.haplo_index(nodes, threads), problems, model)
        same = int(((a[0]["status"] == b[0]["status"]) & (a[0]["score"] == b[0]["score"]) & (a[0]["ok"] == b[0]["ok"])).sum())
        say("seed", s, "same", same, "of", len(problems), "declined", int((b[0]["status"] == -7).sum()), "retried", T.last_wave(eng, 2), "ms", T.last_wave(eng, 0), T.last_wave(eng, 1))
elif stage == "longread":
    from vg_amd import pipeline, workloads
    wl = workloads.LongReadWorkload(int(sys.argv[2]), seed=515)
    say("workload", wl.n)
    cs = pipeline.ChainStage(wl)
    for k in range(3):
        out = cs.run(threads=8)
        say("run", k, out["stats"], "kernel ms", out["wfa_kernel_ms"])
elif stage == "each":
    ora = capi.Engine(lib=util.ORACLE_LIB); eng = capi.Engine()
    s = int(sys.argv[2]); n = int(sys.argv[3]); lo = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    rng = np.random.default_rng(s)
    nodes, threads, problems = T.random_wfa_case(rng, n)
    model = T.MODELS[s % len(T.MODELS)]
    a = ora.wfa_extend(ora.haplo_index(nodes, threads), problems, model)
    idx = eng.haplo_index(nodes, threads)
    for i in range(lo, n):
        print("problem", i, problems[i]["mode"], len(problems[i]["seq"]), end=" ... ", flush=True)
        b = eng.wfa_extend(idx, problems[i:i + 1], model)
        print(int(b[0]["status"][0]), int(b[0]["score"][0]), "oracle", int(a[0]["status"][i]), int(a[0]["score"][i]), "retried", T.last_wave(eng, 2), flush=True)
elif stage == "launches":
    from vg_amd import workloads
    eng = capi.Engine()
    n = int(sys.argv[2])
    wl = workloads.WfaWorkload(n) if hasattr(workloads, "WfaWorkload") else None
    idx = eng.haplo_index(wl.nodes, wl.threads)
    for k in range(3):
        res, paths, edits = eng.wfa_extend(idx, wl.ws)
        say("call", k, "kernel ms", eng.wfa_last_ms(), "small", T.last_wave(eng, 0), "large", T.last_wave(eng, 1), "retried", T.last_wave(eng, 2), "ok", int((res["ok"] != 0).sum()), "declined", int((res["status"] != 0).sum()))
