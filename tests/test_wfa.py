"""Haplotype-consistent wavefront alignment: the oracle against the reference's known-answer tests
(src/unittest/gbwt_extender.cpp:1531-2640, transcribed into tests/golden/ref_wfa_extender.json by
tests/golden/extract_wfa_tests.py), then the kernel (CPU emulation here, HIP under -m gpu) against the oracle."""
import ctypes

import numpy as np
import pytest

import util
from vg_amd import capi

GOLD = util.load_golden("ref_wfa_extender.json")
SCORES = dict(match=1, mismatch=4, gap_open=6, gap_extend=1, bonus=5)


def graph_tables(name):
    g = GOLD["graphs"][name]
    ids = [i for i, _ in g["nodes"]]
    assert ids == sorted(ids)
    index_of = {nid: k for k, nid in enumerate(ids)}
    nodes = [s for _, s in g["nodes"]]
    threads = [[2 * index_of[i] + int(r) for i, r in p] for p in g["paths"]]
    return nodes, threads, index_of


def problem_of(case, index_of):
    def pos(p):
        if p is None:
            return None
        nid, rev, off = p
        return (2 * index_of[nid] + int(rev), off) if nid in index_of else (0x7fffffff, off)
    return dict(seq=case["sequence"], mode=case["call"], **{"from": pos(case["from"]), "to": pos(case["to"])})


def node_seq(nodes, o):
    s = nodes[o // 2]
    return s if not o & 1 else s[::-1].translate(str.maketrans("ACGT", "TGCA"))


def edges_of(threads):
    e = set()
    for t in threads:
        for a, b in zip(t, t[1:]):
            e.add((a, b)); e.add((b ^ 1, a ^ 1))
    return e


def unpack(res, paths, edits, i=0):
    r = res[i]
    path = [int(x) for x in paths[r["path_begin"]:r["path_begin"] + r["path_len"]]]
    ed = [(int(x) & 3, int(x) >> 2) for x in edits[r["edit_begin"]:r["edit_begin"] + r["n_edits"]]]
    return r, path, ed


def final_offset(r, path, ed, nodes):                                 # WFAAlignment::final_offset (:821-832)
    f = int(r["node_offset"]) + sum(n for t, n in ed if t != capi.WFA_INSERTION)
    return f - sum(len(nodes[o // 2]) for o in path[:-1])


def correct_score(r, ed, seq_len, pinned_left, pinned_right):         # unittest correct_score (:132-167)
    s = 0
    for t, n in ed:
        if t == capi.WFA_MATCH:
            s += n * SCORES["match"]
        elif t == capi.WFA_MISMATCH:
            s -= n * SCORES["mismatch"]
        else:
            s -= SCORES["gap_open"] + (n - 1) * SCORES["gap_extend"]
    if ed and int(r["length"]) == seq_len:
        if not pinned_right and ed[-1][0] in (capi.WFA_MATCH, capi.WFA_MISMATCH):
            s += SCORES["bonus"]
        if not pinned_left and ed[0][0] in (capi.WFA_MATCH, capi.WFA_MISMATCH):
            s += SCORES["bonus"]
    return s


def check_alignment(case, r, path, ed, nodes, threads, index_of):     # unittest check_alignment (:1423-1521)
    seq = case["sequence"]
    frm = case["from"] if case["call"] != "prefix" else None
    to = case["to"] if case["call"] != "suffix" else None
    assert r["ok"]
    assert r["seq_offset"] + r["length"] <= len(seq)
    assert frm is None or r["seq_offset"] == 0
    assert to is None or r["seq_offset"] + r["length"] == len(seq)
    assert sum(n for t, n in ed if t != capi.WFA_DELETION) == r["length"]
    edges = edges_of(threads)
    for a, b in zip(path, path[1:]):
        assert (a, b) in edges
    if path:
        assert r["node_offset"] < len(nodes[path[0] // 2])
        if frm is not None:
            fh = 2 * index_of[frm[0]] + int(frm[1])
            if path[0] == fh and r["node_offset"] > 0:
                assert r["node_offset"] == frm[2] + 1
            else:
                assert frm[2] + 1 == len(nodes[fh // 2]) and (fh, path[0]) in edges and r["node_offset"] == 0
        fo = final_offset(r, path, ed, nodes)
        assert fo > 0
        if to is not None:
            th = 2 * index_of[to[0]] + int(to[1])
            last_len = len(nodes[path[-1] // 2])
            if path[-1] == th and fo < last_len:
                assert fo == to[2]
            else:
                assert to[2] == 0 and (path[-1], th) in edges and fo == last_len
    for (a, _), (b, _) in zip(ed, ed[1:]):
        assert a != b
    assert r["score"] == correct_score(r, ed, len(seq), frm is not None, to is not None)
    so, no, po = int(r["seq_offset"]), int(r["node_offset"]), 0
    for t, n in ed:
        if t == capi.WFA_INSERTION:
            so += n
            continue
        end = no + n
        while end > no:
            assert po < len(path)
            ns = node_seq(nodes, path[po])
            ln = min(end, len(ns)) - no
            if t == capi.WFA_MATCH:
                assert seq[so:so + ln] == ns[no:no + ln]
                so += ln
            elif t == capi.WFA_MISMATCH:
                assert all(seq[so + i] != ns[no + i] for i in range(ln))
                so += ln
            no += ln
            if no >= len(ns):
                no = 0; end -= len(ns); po += 1
    if path:
        assert po == len(path) - 1 or (po == len(path) and no == 0)


def run_golden(eng):
    indexes = {}
    for case in GOLD["cases"]:
        name = case["graph"]
        if name not in indexes:
            nodes, threads, index_of = graph_tables(name)
            indexes[name] = (eng.haplo_index(nodes, threads), nodes, threads, index_of)
        index, nodes, threads, index_of = indexes[name]
        res, paths, edits = eng.wfa_extend(index, [problem_of(case, index_of)], case["error_model"])
        r, path, ed = unpack(res, paths, edits)
        assert r["status"] == 0, case["name"]
        exp = case["expect"]
        try:
            if exp["kind"] == "fail":
                assert not r["ok"]
            elif exp["kind"] == "unlocalized_insertion":            # check_unlocalized_insertion (:1407-1421)
                assert r["ok"] and not path and ed == [(capi.WFA_INSERTION, len(case["sequence"]))]
                assert r["seq_offset"] == 0 and r["length"] == len(case["sequence"])
                assert r["score"] == -SCORES["gap_open"] - (int(r["length"]) - 1) * SCORES["gap_extend"]
            else:
                ext = exp["gap_length"] - exp["gaps"]               # check_score (:1390-1405)
                want = (exp["matches"] * SCORES["match"] - exp["mismatches"] * SCORES["mismatch"] - exp["gaps"] * SCORES["gap_open"]
                        - ext * SCORES["gap_extend"] + exp["full_length_ends"] * SCORES["bonus"])
                assert r["ok"] and r["score"] == want
                if exp["check_alignment"]:
                    check_alignment(case, r, path, ed, nodes, threads, index_of)
        except AssertionError as e:
            raise AssertionError("%s (%s): %s  got %s path %s edits %s" % (case["name"], case["source"], e, r, path, ed)) from e
    return len(GOLD["cases"])


def test_oracle_matches_reference_known_answers():
    eng = capi.Engine(lib=capi.load_library(util.ORACLE_LIB))
    assert run_golden(eng) >= 100


# ---- random haplotype graphs: the engine against the oracle ----------------------------------------------------------

def random_wfa_case(rng, n_problems=60, n_haplotypes=None):
    """The bubble chains of test_gapless with some threads going round a cycle; sequences cut from a thread (either strand)
    between a `from` and a `to` base, with substitutions, insertions and deletions; connect / suffix / prefix problems,
    a few of them between unrelated positions."""
    from test_gapless import random_haplotype_case
    bases = "ACGT"
    nodes, threads, _ = random_haplotype_case(rng, n_reads=0, n_haplotypes=n_haplotypes or int(rng.integers(1, 6)), chain_nodes=int(rng.integers(4, 16)))
    if rng.random() < 0.4:                                   # cycles: repeat a stretch of a thread
        for t in threads:
            if len(t) > 4 and rng.random() < 0.7:
                i = int(rng.integers(0, len(t) - 2)); k = int(rng.integers(i + 1, len(t)))
                t[k:k] = t[i:k] * int(rng.integers(1, 3))

    def comp(s):
        return s[::-1].translate(str.maketrans("ACGT", "TGCA"))

    problems = []
    for _ in range(n_problems):
        t = threads[int(rng.integers(0, len(threads)))]
        if rng.random() < 0.5:
            t = [o ^ 1 for o in reversed(t)]
        seq = "".join(nodes[o // 2] if not o & 1 else comp(nodes[o // 2]) for o in t)
        starts = np.cumsum([0] + [len(nodes[o // 2]) for o in t])

        def pos(g):
            k = int(np.searchsorted(starts, g, side="right") - 1)
            return (int(t[k]), int(g - starts[k]))
        if len(seq) < 3:
            continue
        f = int(rng.integers(0, len(seq) - 1)); n = int(rng.integers(0, min(50, len(seq) - f - 1) + 1))
        read = []
        rate = float(rng.choice([0.0, 0.03, 0.1]))
        for c in seq[f + 1:f + 1 + n]:
            r = rng.random()
            if r < rate:
                read.append(bases[int(rng.integers(0, 4))])
            elif r < 1.4 * rate:
                continue
            elif r < 1.8 * rate:
                read.append(c); read.append(bases[int(rng.integers(0, 4))])
            elif r < 1.8 * rate + 0.004:
                read.append("N")
            else:
                read.append(c)
        read = "".join(read)
        mode = ["connect", "suffix", "prefix"][int(rng.integers(0, 3))]
        g_to = f + n + 1
        if mode == "connect" and g_to >= len(seq):
            mode = "suffix"
        p = dict(seq=read, mode=mode)
        if mode != "prefix":
            p["from"] = pos(f)
        if mode == "connect":
            p["to"] = pos(g_to)
        if mode == "prefix":
            p["to"] = pos(min(g_to, len(seq) - 1))
        if rng.random() < 0.08:                              # unrelated endpoints
            o = int(rng.integers(0, 2 * len(nodes)))
            p["to" if mode != "suffix" else "from"] = (o, int(rng.integers(0, len(nodes[o // 2]))))
        if rng.random() < 0.02:
            p["from" if mode != "prefix" else "to"] = (2 * len(nodes) + 3, 0) if mode != "prefix" else p["to"]
        problems.append(p)
    return nodes, threads, problems


MODELS = [None, ((0.03, 1, 6), (0.05, 1, 10), (0.1, 1, 20), (0.1, 2, 4)), ((0.0, 2, 2), (0.0, 1, 1), (0.0, 3, 3), (0.5, 0, 200)),
          ((0.1, 3, 12), (0.1, 2, 12), (0.2, 4, 30), (0.1, 10, 200))]


def compare_engines(lib, seeds, n_problems=60):
    ora = capi.Engine(lib=util.ORACLE_LIB); eng = capi.Engine(lib=lib) if lib else capi.Engine()
    ok = 0; statuses = {}
    for s in seeds:
        rng = np.random.default_rng(s)
        nodes, threads, problems = random_wfa_case(rng, n_problems)
        model = MODELS[s % len(MODELS)]
        a = ora.wfa_extend(ora.haplo_index(nodes, threads), problems, model)
        b = eng.wfa_extend(eng.haplo_index(nodes, threads), problems, model)
        for st in b[0]["status"]:
            statuses[int(st)] = statuses.get(int(st), 0) + 1
        good = b[0]["status"] != -7                          # the kernel's table limits are its own; everything else must agree
        assert (a[0]["status"][good] == b[0]["status"][good]).all(), (s, np.nonzero(a[0]["status"] != b[0]["status"])[0])
        for i in np.nonzero(good)[0]:
            ra, pa, ea = unpack(*a, i); rb, pb, eb = unpack(*b, i)
            fields = ("status", "ok", "score", "node_offset", "seq_offset", "length", "path_len", "n_edits")
            assert all(ra[f] == rb[f] for f in fields) and pa == pb and ea == eb, (s, i, problems[i], ra, rb, pa, pb, ea, eb)
            ok += int(ra["ok"])
    return ok, statuses


def test_emulated_wfa_kernel_matches_reference_unit_tests_and_oracle():
    import subprocess
    subprocess.check_call(["make", "-s", "emu"], cwd=util.ROOT)
    assert run_golden(capi.Engine(lib=capi.load_library(util.EMU_LIB))) >= 100
    ok, statuses = compare_engines(util.EMU_LIB, range(300, 360))
    assert ok > 1500 and statuses.get(0, 0) > 0.98 * sum(statuses.values()), (ok, statuses)


@pytest.mark.gpu
def test_hip_wfa_matches_reference_unit_tests_and_oracle():
    assert run_golden(capi.Engine()) >= 100
    ok, statuses = compare_engines(None, range(400, 480), n_problems=400)
    assert ok > 15000 and statuses.get(0, 0) > 0.98 * sum(statuses.values()), (ok, statuses)


# ---- the C++ host shim (vg_amd/host/gbwt_extender.hpp: WFAExtender, WFAAlignment), driven like the reference's unit tests ----

def shim_golden(engine_lib):
    """Every known-answer case through WFAExtender::connect / suffix / prefix of the host shim: the reference's own checks
    on the returned WFAAlignment, plus WFAAlignment::to_path consistency."""
    import ctypes, json
    h = util.host()
    h.vgh_wfa_create.restype = ctypes.c_void_p
    h.vgh_wfa_create.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int32), ctypes.c_int,
                                 ctypes.POINTER(ctypes.c_double)]
    h.vgh_wfa_destroy.argtypes = [ctypes.c_void_p]
    h.vgh_wfa_align.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64),
                                ctypes.c_char_p, ctypes.c_size_t]
    al = util.HostAligner(engine_lib)
    checked = 0
    for case in GOLD["cases"]:
        gdef = GOLD["graphs"][case["graph"]]
        nodes, threads, index_of = graph_tables(case["graph"])
        g = h.vgh_graph_create()
        try:
            for nid, s in gdef["nodes"]:
                assert h.vgh_graph_add_node(g, nid, s.encode()) == 0
            flat = [2 * i + int(r) for p in gdef["paths"] for i, r in p]
            off = np.concatenate([[0], np.cumsum([len(p) for p in gdef["paths"]])])
            model = [float(v) for row in case["error_model"] for v in row]
            x = h.vgh_wfa_create(al.ptr, g, (ctypes.c_int64 * len(flat))(*flat), (ctypes.c_int32 * len(off))(*[int(v) for v in off]), len(gdef["paths"]),
                                 (ctypes.c_double * 12)(*model))
            assert x, h.vgh_last_error().decode()
            try:
                def pos(p):
                    return (ctypes.c_int64 * 3)(p[0], int(p[1]), p[2]) if p is not None else None
                buf = ctypes.create_string_buffer(1 << 16)
                kind = {"connect": 0, "suffix": 1, "prefix": 2}[case["call"]]
                rc = h.vgh_wfa_align(x, kind, case["sequence"].encode(), pos(case["from"]), pos(case["to"]), buf, len(buf))
                assert rc == 0, h.vgh_last_error().decode()
                out = json.loads(buf.value.decode())
            finally:
                h.vgh_wfa_destroy(x)
        finally:
            h.vgh_graph_destroy(g)
        exp = case["expect"]
        if exp["kind"] == "fail":
            assert not out["ok"], case["name"]
            continue
        assert out["ok"], case["name"]
        r = dict(ok=1, score=out["score"], node_offset=out["node_offset"], seq_offset=out["seq_offset"], length=out["length"])
        path = [2 * index_of[i] + int(rev) for i, rev in out["path"]]
        ed = [tuple(e) for e in out["edits"]]
        if exp["kind"] == "unlocalized_insertion":
            assert out["unlocalized_insertion"] and not path and ed == [(capi.WFA_INSERTION, len(case["sequence"]))]
        else:
            ext = exp["gap_length"] - exp["gaps"]
            assert out["score"] == (exp["matches"] * SCORES["match"] - exp["mismatches"] * SCORES["mismatch"] - exp["gaps"] * SCORES["gap_open"]
                                    - ext * SCORES["gap_extend"] + exp["full_length_ends"] * SCORES["bonus"]), case["name"]
            if exp["check_alignment"]:
                check_alignment(case, r, path, ed, nodes, threads, index_of)
        # WFAAlignment::to_path: one mapping per path node, edits add up to the aligned interval
        maps = out["alignment"]["path"].get("mapping", [])
        if path:
            assert [m["position"]["node_id"] for m in maps] == [i for i, _ in out["path"]]
            assert maps[0]["position"].get("offset", 0) == out["node_offset"]
        assert sum(e.get("to_length", 0) for m in maps for e in m["edit"]) == out["length"]
        checked += 1
    return checked


def test_host_shim_wfa_extender_on_the_oracle():
    assert shim_golden(util.ORACLE_LIB) > 80


@pytest.mark.gpu
def test_host_shim_wfa_extender_on_hip():
    assert shim_golden(util.ENGINE_LIB) > 80


def test_wfa_over_run_length_encoded_records(monkeypatch):
    """Hundreds of haplotypes: the engine's index stores most visit bodies run-length encoded (gapless_device.hpp); WFA's walk over the
    haplotype trie (w_follow) must give what the oracle's uncompressed index gives, problem by problem."""
    import subprocess
    subprocess.check_call(["make", "-s", "emu"], cwd=util.ROOT)
    n_ok = 0
    for seed in (1, 2, 3):
        nodes, threads, wp = random_wfa_case(np.random.default_rng(seed), 80, n_haplotypes=250)
        eng = capi.Engine(lib=util.EMU_LIB); ora = capi.Engine(lib=util.ORACLE_LIB)
        (ra, pa, ea), (rb, pb, eb) = eng.wfa_extend(eng.haplo_index(nodes, threads), wp), ora.wfa_extend(ora.haplo_index(nodes, threads), wp)
        for i in range(len(ra)):
            if ra["status"][i] != 0:
                assert ra["status"][i] == -7      # a kernel table limit: the caller's DP path takes the problem
                continue
            for f in ("ok", "score", "node_offset", "seq_offset", "length", "path_len", "n_edits"):
                assert ra[f][i] == rb[f][i], (seed, i, f)
            assert (pa[ra["path_begin"][i]:ra["path_begin"][i] + ra["path_len"][i]] == pb[rb["path_begin"][i]:rb["path_begin"][i] + rb["path_len"][i]]).all()
            assert (ea[ra["edit_begin"][i]:ra["edit_begin"][i] + ra["n_edits"][i]] == eb[rb["edit_begin"][i]:rb["edit_begin"][i] + rb["n_edits"][i]]).all()
            n_ok += 1
    assert n_ok > 150


# ---- the kernels: one thread per problem, one wavefront per problem (with its two table sizes), and the default, hybrid: the thread kernel for
# the easy majority, the wavefront kernel for what it hands over ----
def test_thread_form_of_the_kernel_matches_the_oracle(monkeypatch):
    import subprocess
    subprocess.check_call(["make", "-s", "emu"], cwd=util.ROOT)
    monkeypatch.setenv("VGAMD_WFA_KERNEL", "thread")
    assert run_golden(capi.Engine(lib=capi.load_library(util.EMU_LIB))) >= 100
    ok, statuses = compare_engines(util.EMU_LIB, range(300, 330))
    assert ok > 700 and statuses.get(0, 0) > 0.98 * sum(statuses.values()), (ok, statuses)


def last_wave(eng, which):
    eng.lib.vgk_wfa_last_wave.restype = ctypes.c_double
    eng.lib.vgk_wfa_last_wave.argtypes = [ctypes.c_void_p, ctypes.c_int]
    return eng.lib.vgk_wfa_last_wave(eng.h, which)


def retries_with_large_tables(lib, seeds, n_problems):
    """small tables of 16 points: most problems outgrow them and run again with the large ones; every answer still equals the oracle's"""
    ora = capi.Engine(lib=util.ORACLE_LIB); eng = capi.Engine(lib=lib) if lib else capi.Engine()
    retried = answered = 0
    for s in seeds:
        nodes, threads, problems = random_wfa_case(np.random.default_rng(s), n_problems)
        a = ora.wfa_extend(ora.haplo_index(nodes, threads), problems, MODELS[s % len(MODELS)])
        b = eng.wfa_extend(eng.haplo_index(nodes, threads), problems, MODELS[s % len(MODELS)])
        retried += int(last_wave(eng, 2))
        for i in range(len(problems)):
            if b[0]["status"][i] == -7:
                continue
            ra, pa, ea = unpack(*a, i); rb, pb, eb = unpack(*b, i)
            assert all(ra[f] == rb[f] for f in ("status", "ok", "score", "node_offset", "seq_offset", "length", "path_len", "n_edits")) and pa == pb and ea == eb, (s, i)
            answered += 1
    return retried, answered


def test_wave_form_matches_the_oracle(monkeypatch):
    import subprocess
    subprocess.check_call(["make", "-s", "emu"], cwd=util.ROOT)
    monkeypatch.setenv("VGAMD_WFA_KERNEL", "wave")
    assert run_golden(capi.Engine(lib=capi.load_library(util.EMU_LIB))) >= 100
    ok, statuses = compare_engines(util.EMU_LIB, range(300, 340))
    assert ok > 1000 and statuses.get(0, 0) > 0.98 * sum(statuses.values()), (ok, statuses)


def test_hybrid_hands_over_at_a_few_points(monkeypatch):
    """the thread kernel gives up at 12 stored points: most problems are then answered by the wavefront kernel, all as the oracle answers them"""
    import subprocess
    subprocess.check_call(["make", "-s", "emu"], cwd=util.ROOT)
    monkeypatch.setenv("VGAMD_WFA_HAND_OVER_POINTS", "12")
    handed, answered = retries_with_large_tables(util.EMU_LIB, range(540, 560), 60)
    assert handed > 200 and answered > 1150, (handed, answered)


def test_wave_form_runs_what_outgrows_the_small_tables_again(monkeypatch):
    import subprocess
    subprocess.check_call(["make", "-s", "emu"], cwd=util.ROOT)
    monkeypatch.setenv("VGAMD_WFA_KERNEL", "wave")
    monkeypatch.setenv("VGAMD_WFA_SMALL_POINTS", "16")
    retried, answered = retries_with_large_tables(util.EMU_LIB, range(500, 520), 60)
    assert retried > 150 and answered > 1150, (retried, answered)


def budgets_decline_the_same_problems(lib, monkeypatch, seeds=range(600, 612), n_problems=60):
    """a caller's point budget ends the same problems in either form of the kernel, and leaves every other answer alone"""
    for s in seeds:
        nodes, threads, problems = random_wfa_case(np.random.default_rng(s), n_problems)
        out = []
        for form in ("wave", "thread", "hybrid"):
            monkeypatch.setenv("VGAMD_WFA_KERNEL", form)
            eng = capi.Engine(lib=lib) if lib else capi.Engine()
            eng.wfa_set_point_budget(24)
            out.append(eng.wfa_extend(eng.haplo_index(nodes, threads), problems, MODELS[s % len(MODELS)]))
        a, b, c = out
        assert (a[0]["status"] == b[0]["status"]).all() and (a[0]["status"] == c[0]["status"]).all(), s
        assert (a[0]["status"] == -7).any()
        for i in np.nonzero(a[0]["status"] == 0)[0]:
            assert unpack(*a, i) == unpack(*b, i) == unpack(*c, i), (s, i)


def test_point_budget_declines_the_same_problems_in_both_forms(monkeypatch):
    import subprocess
    subprocess.check_call(["make", "-s", "emu"], cwd=util.ROOT)
    budgets_decline_the_same_problems(util.EMU_LIB, monkeypatch)


@pytest.mark.gpu
def test_both_forms_and_both_table_sizes_on_the_gpu(monkeypatch):
    budgets_decline_the_same_problems(None, monkeypatch, seeds=range(700, 720), n_problems=300)
    monkeypatch.setenv("VGAMD_WFA_KERNEL", "thread")
    ok, statuses = compare_engines(None, range(440, 460), n_problems=400)
    assert ok > 3500
    monkeypatch.setenv("VGAMD_WFA_KERNEL", "wave")
    ok, statuses = compare_engines(None, range(460, 480), n_problems=400)
    assert ok > 3500
    monkeypatch.setenv("VGAMD_WFA_SMALL_POINTS", "16")
    retried, answered = retries_with_large_tables(None, range(800, 820), 400)
    assert retried > 1000 and answered > 7800, (retried, answered)
    monkeypatch.delenv("VGAMD_WFA_SMALL_POINTS"); monkeypatch.setenv("VGAMD_WFA_KERNEL", "hybrid"); monkeypatch.setenv("VGAMD_WFA_HAND_OVER_POINTS", "12")
    handed, answered = retries_with_large_tables(None, range(840, 860), 400)
    assert handed > 1200 and answered > 7800, (handed, answered)


def cost_hints_change_the_order_only(lib):
    """vgk_wfa_set_cost_hints: whatever the caller claims about the problems' costs — nothing, nonsense, a list of the wrong length — the
    answers are the same; a hint list is used by one call."""
    ora = capi.Engine(lib=util.ORACLE_LIB); eng = capi.Engine(lib=lib) if lib else capi.Engine()
    rng = np.random.default_rng(4242)
    nodes, threads, problems = random_wfa_case(rng, 300)
    want = ora.wfa_extend(ora.haplo_index(nodes, threads), problems)
    idx = eng.haplo_index(nodes, threads)
    plain = eng.wfa_extend(idx, problems)
    for hints in (rng.integers(0, 60000, len(problems)), np.zeros(len(problems)), rng.integers(0, 5000, len(problems) - 7)):
        eng.wfa_set_cost_hints(hints)
        got = eng.wfa_extend(idx, problems)
        assert all(len(x) == len(y) and (x == y).all() for x, y in zip(got, plain))
    assert (plain[0]["status"] == want[0]["status"]).all() and (plain[0]["score"] == want[0]["score"]).all()


def sequences_masked_on_the_device_or_the_host(lib, monkeypatch):
    """the caller's sequences in one stretch of memory go up as they are and a kernel masks / flips / lays them out (wfa_mask_one); sequences
    scattered over the heap — or VGAMD_WFA_HOST_MASK=1 — are gathered and masked by the host threads: the same answers, N's and PREFIX problems included"""
    eng = capi.Engine(lib=lib) if lib else capi.Engine()
    rng = np.random.default_rng(909)
    nodes, threads, problems = random_wfa_case(rng, 400)
    assert any(p["mode"] == "prefix" for p in problems) and any("N" in p["seq"] for p in problems)
    idx = eng.haplo_index(nodes, threads)
    device = eng.wfa_extend(idx, problems)
    monkeypatch.setenv("VGAMD_WFA_HOST_MASK", "1")
    host = eng.wfa_extend(idx, problems)
    monkeypatch.delenv("VGAMD_WFA_HOST_MASK")
    assert all(len(x) == len(y) and (x == y).all() for x, y in zip(device, host))
    # scattered: every sequence a buffer of its own, megabytes apart
    ws = capi.WfaSet.from_lists(problems)
    keep = [np.frombuffer((p["seq"] or "A").encode(), dtype=np.uint8).copy() for p in problems] + [np.zeros(1 << 22, dtype=np.uint8)]
    far = np.zeros(1 << 23, dtype=np.uint8); far[:len(keep[0])] = keep[0]; keep[0] = far
    ws.array["seq"] = [k.ctypes.data for k in keep[:len(problems)]]
    scattered = eng.wfa_extend(idx, ws)
    assert all(len(x) == len(y) and (x == y).all() for x, y in zip(device, scattered))


def test_sequences_masked_on_the_emulated_device_or_the_host(monkeypatch):
    sequences_masked_on_the_device_or_the_host(util.EMU_LIB, monkeypatch)


@pytest.mark.gpu
def test_sequences_masked_on_the_device_or_the_host(monkeypatch):
    sequences_masked_on_the_device_or_the_host(None, monkeypatch)


def test_emulated_cost_hints_change_the_order_only():
    cost_hints_change_the_order_only(util.EMU_LIB)


@pytest.mark.gpu
def test_hip_cost_hints_change_the_order_only():
    cost_hints_change_the_order_only(None)


# ---- the wavefront kernel over MERGED RUNS (wfa_wave_device.hpp: pieces of runs, cut where the node-by-node walk ends a trie node) ----
def long_run_case(rng, n_problems, n_haplotypes=3):
    """Chains of long non-branching stretches (2-14 nodes of 1-32 bases, node ids in path order: what the index merges into runs) between
    SNP bubbles and the odd deletion; sequences of up to 1 400 bases cut from a thread, few errors: trie nodes that run past WFANode's
    1 024 bases inside a run, targets in the middle of runs, prefixes (the other strand: runs walked backwards)."""
    bases = "ACGT"
    nodes = []; segments = []
    for _ in range(int(rng.integers(6, 16))):
        run = []
        for _ in range(int(rng.integers(2, 15))):
            nodes.append("".join(bases[int(x)] for x in rng.integers(0, 4, int(rng.integers(1, 33))))); run.append(len(nodes) - 1)
        kind = rng.random()
        if kind < 0.6:                                         # a SNP bubble behind the stretch
            a, b = bases[int(rng.integers(0, 4))], bases[int(rng.integers(0, 4))]
            nodes.append(a); nodes.append(b if b != a else bases[(bases.index(a) + 1) % 4])
            segments.append((run, [len(nodes) - 2, len(nodes) - 1]))
        elif kind < 0.8:                                       # a node some threads skip
            nodes.append("".join(bases[int(x)] for x in rng.integers(0, 4, int(rng.integers(1, 9)))))
            segments.append((run, [len(nodes) - 1, None]))
        else:
            segments.append((run, [None]))                     # the next stretch follows at once (a run ends at the 255-base cap or goes on)
    threads = []
    for _ in range(n_haplotypes):
        t = []
        for run, alleles in segments:
            t += [2 * v for v in run]
            a = alleles[int(rng.integers(0, len(alleles)))]
            if a is not None:
                t.append(2 * a)
        threads.append(t)

    def comp(s):
        return s[::-1].translate(str.maketrans("ACGT", "TGCA"))
    problems = []
    for _ in range(n_problems):
        t = threads[int(rng.integers(0, len(threads)))]
        if rng.random() < 0.5:
            t = [o ^ 1 for o in reversed(t)]
        seq = "".join(nodes[o // 2] if not o & 1 else comp(nodes[o // 2]) for o in t)
        starts = np.cumsum([0] + [len(nodes[o // 2]) for o in t])

        def pos(g):
            k = int(np.searchsorted(starts, g, side="right") - 1)
            return (int(t[k]), int(g - starts[k]))
        f = int(rng.integers(0, max(1, len(seq) // 3)))
        n = int(rng.integers(0, min(1400, len(seq) - f - 2) + 1))
        rate = float(rng.choice([0.0, 0.004, 0.01]))
        read = []
        for c in seq[f + 1:f + 1 + n]:
            r = rng.random()
            if r < rate:
                read.append(bases[int(rng.integers(0, 4))])
            elif r < 1.3 * rate:
                continue
            elif r < 1.6 * rate:
                read.append(c); read.append(bases[int(rng.integers(0, 4))])
            else:
                read.append(c)
        mode = ["connect", "suffix", "prefix"][int(rng.integers(0, 3))]
        p = dict(seq="".join(read), mode=mode)
        if mode != "prefix":
            p["from"] = pos(f)
        if mode != "suffix":
            p["to"] = pos(f + n + 1)
        problems.append(p)
    return nodes, threads, problems


def merged_runs_give_the_node_by_node_answers(lib, seeds, monkeypatch, n_problems=40):
    ora = capi.Engine(lib=util.ORACLE_LIB)
    monkeypatch.setenv("VGAMD_WFA_KERNEL", "wave")
    fields = ("status", "ok", "score", "node_offset", "seq_offset", "length", "path_len", "n_edits")
    answered = long_nodes = 0
    for s in seeds:
        rng = np.random.default_rng(s)
        nodes, threads, problems = long_run_case(rng, n_problems, n_haplotypes=1 if s % 3 == 0 else 3)      # (one thread: nothing branches, every long trie node meets the 1 024-base rule)
        eng = capi.Engine(lib=lib)
        index = eng.haplo_index(nodes, threads)
        assert index.run_nodes() < len(nodes) * 0.6, "the case is meant to have runs to merge"
        a = ora.wfa_extend(ora.haplo_index(nodes, threads), problems)
        b = eng.wfa_extend(index, problems)
        monkeypatch.setenv("VGAMD_WFA_NO_MERGE", "1")
        c = eng.wfa_extend(index, problems)                  # the same kernel hopping node by node
        monkeypatch.delenv("VGAMD_WFA_NO_MERGE")
        # the trie is the same node for node: what one walk declines for its tables (points, trie nodes) the other declines as well; only the
        # path pool differs (a run is one entry)
        for i in range(len(problems)):
            rb, pb, eb = unpack(*b, i); rc, pc, ec = unpack(*c, i)
            if rb["status"] == -7 or rc["status"] == -7:
                assert rc["status"] == -7, (s, i, "declined on the runs only", rb, rc)
                continue
            ra, pa, ea = unpack(*a, i)
            assert all(ra[f] == rb[f] == rc[f] for f in fields) and pa == pb == pc and ea == eb == ec, (s, i, problems[i], ra, rb, rc, pa, pb, pc)
            answered += int(ra["ok"]); long_nodes += int(ra["ok"] and len(problems[i]["seq"]) > 1100)
        eng.close()
    return answered, long_nodes


def test_wave_form_over_merged_runs_on_the_emulator(monkeypatch):
    import subprocess
    subprocess.check_call(["make", "-s", "emu"], cwd=util.ROOT)
    answered, long_nodes = merged_runs_give_the_node_by_node_answers(util.EMU_LIB, range(900, 912), monkeypatch)
    assert answered > 250 and long_nodes > 10


@pytest.mark.gpu
def test_wave_form_over_merged_runs_on_hip(monkeypatch):
    answered, long_nodes = merged_runs_give_the_node_by_node_answers(util.ENGINE_LIB, range(900, 940), monkeypatch, n_problems=60)
    assert answered > 1200 and long_nodes > 50
