"""Haplotype-consistent gapless extension: the oracle against the reference's known-answer tests
(src/unittest/gbwt_extender.cpp:868-1158, transcribed by hand below with their line numbers)."""
import subprocess

import numpy as np
import pytest

import util
from vg_amd import capi

# the toy graph GA(T|GGG)TA(C|A)A and its three threads (src/unittest/gbwt_extender.cpp:31-92)
TOY_NODES = ["G", "A", "T", "GGG", "T", "A", "C", "A", "A"]          # ids 1..9
ALT_PATH = [1, 2, 4, 5, 6, 8, 9]
SHORT_PATH = [1, 4, 5, 6, 7, 9]


def oriented(node_id, is_reverse=False):
    return 2 * (node_id - 1) + int(is_reverse)


def toy_index(eng):
    threads = [[oriented(i) for i in t] for t in (SHORT_PATH, ALT_PATH, SHORT_PATH)]
    return eng.haplo_index(TOY_NODES, threads)


def seed(pos, read_offset):
    node_id, is_reverse, offset = pos
    return (oriented(node_id, is_reverse), read_offset - offset)      # GaplessExtender::to_seed (gbwt_extender.hpp:160-163)


def extension_as_mappings(read, e, nodes, mism, node_seqs):
    """GaplessExtension::to_path (src/gbwt_extender.cpp:105-151) in the tests' notation: [((id, rev, offset), edits)] with
    edits like "1A1" = 1 match, mismatch to read base A, 1 match."""
    out = []
    ro = int(e["read_begin"]); no = int(e["offset"])
    mm = list(mism[e["mism_begin"]:e["mism_begin"] + e["n_mismatches"]])
    for k in range(e["path_len"]):
        o = int(nodes[e["path_begin"] + k])
        ln = len(node_seqs[o // 2])
        limit = min(ro + ln - no, int(e["read_end"]))
        edits = ""
        while mm and mm[0] < limit:
            if ro < mm[0]:
                edits += str(mm[0] - ro)
            edits += read[mm[0]]
            ro = mm[0] + 1; mm.pop(0)
        if ro < limit:
            edits += str(limit - ro); ro = limit
        out.append(((o // 2 + 1, bool(o & 1), no), edits))
        no = 0
    return out


def correct_score(e, match=1, mismatch=4, bonus=5):                   # (:124-130)
    n = int(e["read_end"] - e["read_begin"]); m = int(e["n_mismatches"])
    return (n - m) * match - m * mismatch + (int(e["left_full"]) + int(e["right_full"])) * bonus


def run(eng, index, seeds, read, error_bound, overlap_threshold=0.8):
    problem = dict(read=read, seeds=[seed(p, r) for p, r in seeds], max_mismatches=error_bound, overlap_threshold=overlap_threshold)
    res, ext, nodes, mism = eng.gapless_extend(index, [problem])
    assert res["status"][0] == 0
    return [ext[i] for i in range(res["ext_begin"][0], res["ext_begin"][0] + res["n_ext"][0])], nodes, mism


def full_length_match(eng, index, seeds, read, correct, error_bound):            # (:499-527)
    exts, nodes, mism = run(eng, index, seeds, read, error_bound)
    if not correct:
        for e in exts:
            if e["left_full"] and e["right_full"]:
                assert e["n_mismatches"] > error_bound
        return
    assert len(exts) == 1
    e = exts[0]
    assert e["read_end"] > e["read_begin"] and e["left_full"] and e["right_full"] and e["n_mismatches"] <= error_bound
    assert e["score"] == correct_score(e)
    assert extension_as_mappings(read, e, nodes, mism, TOY_NODES) == correct


def full_length_matches(eng, index, seeds, read, corrects, error_bound, overlap_threshold):   # (:529-544)
    exts, nodes, mism = run(eng, index, seeds, read, error_bound, overlap_threshold)
    assert len(exts) == len(corrects)
    for e, c in zip(exts, corrects):
        assert e["left_full"] and e["right_full"] and e["n_mismatches"] <= error_bound
        assert e["score"] == correct_score(e)
        assert extension_as_mappings(read, e, nodes, mism, TOY_NODES) == c


def partial_matches(eng, index, seeds, read, corrects, offsets, error_bound, node_seqs=TOY_NODES):   # (:546-564)
    exts, nodes, mism = run(eng, index, seeds, read, error_bound)
    assert len(exts) == len(corrects)
    for e, c, off in zip(exts, corrects, offsets):
        assert e["read_end"] > e["read_begin"]
        if e["left_full"] and e["right_full"]:
            assert e["n_mismatches"] > error_bound
        assert e["read_begin"] == off
        assert e["score"] == correct_score(e)
        assert extension_as_mappings(read, e, nodes, mism, node_seqs) == c


def reference_gapless_cases(eng):
    index = toy_index(eng)
    F, T = False, True
    # "Full-length alignments" (:868-1001)
    full_length_match(eng, index, [((4, F, 2), 0), ((6, F, 0), 2)], "GTACA",                                         # :880
                      [((4, F, 2), "1"), ((5, F, 0), "1"), ((6, F, 0), "1"), ((7, F, 0), "1"), ((9, F, 0), "1")], 0)
    errors = [((1, F, 0), "1"), ((4, F, 0), "1A1"), ((5, F, 0), "1"), ((6, F, 0), "1"), ((7, F, 0), "1")]
    full_length_match(eng, index, [((5, F, 0), 4), ((4, F, 2), 3)], "GGAGTAC", errors, 1)                            # :897
    full_length_match(eng, index, [((5, F, 0), 4), ((4, F, 2), 3), ((2, F, 0), 0)], "GGAGTAC", errors, 1)            # :914
    full_length_match(eng, index, [((5, T, 0), 2), ((6, T, 0), 1)], "GTACT",                                         # :932
                      [((7, T, 0), "1"), ((6, T, 0), "1"), ((5, T, 0), "1"), ((4, T, 0), "1T")], 1)
    full_length_match(eng, index, [((5, F, 0), 4), ((4, F, 2), 3)], "AGAGTAC", [], 1)                                # :948
    seeds = [((2, F, 0), 1), ((4, F, 0), 2), ((4, F, 0), 1)]
    best = [((1, F, 0), "1"), ((2, F, 0), "1"), ((4, F, 0), "2A")]
    second = [((1, F, 0), "1"), ((4, F, 0), "A2"), ((5, F, 0), "A")]
    full_length_matches(eng, index, seeds, "GAGGA", [best, second], 2, 0.9)                                          # :959
    full_length_matches(eng, index, seeds, "GAGGA", [best], 2, 0.1)                                                  # :983
    # "Local alignments" (:1005-1120)
    partial_matches(eng, index, [((4, F, 0), 1), ((2, F, 0), 7), ((5, F, 0), 11), ((7, F, 0), 15), ((6, F, 0), 20)],  # :1017
                    "AGGGxCGAGxGTAxACAAxTAA",
                    [[((2, F, 0), "1"), ((4, F, 0), "3")],
                     [((1, F, 0), "1"), ((2, F, 0), "1"), ((4, F, 0), "1")],
                     [((4, F, 2), "1"), ((5, F, 0), "1"), ((6, F, 0), "1")],
                     [((6, F, 0), "1"), ((7, F, 0), "1"), ((9, F, 0), "1")],
                     [((5, F, 0), "1"), ((6, F, 0), "1"), ((8, F, 0), "1")]],
                    [0, 6, 10, 14, 19], 0)
    partial_matches(eng, index, [((4, F, 2), 4)], "xAGxGTAx", [[((4, F, 2), "1"), ((5, F, 0), "1"), ((6, F, 0), "1")]], [4], 0)   # :1063
    trimmed = [[((2, F, 0), "1"), ((4, F, 0), "3"), ((5, F, 0), "1")]]
    partial_matches(eng, index, [((4, F, 2), 4)], "xAGGGTxAx", trimmed, [1], 1)                                      # :1082
    partial_matches(eng, index, [((2, F, 0), 1), ((4, F, 2), 4)], "xAGGGTxAx", trimmed, [1], 1)                      # :1101
    # "Non-ACGT characters do not match" (:1124-1156)
    one = eng.haplo_index(["NNNGATTACANNN"], [[0]])
    partial_matches(eng, one, [((1, F, 5), 4)], "NNGATTACANN", [[((1, F, 3), "7")]], [2], 0, node_seqs=["NNNGATTACANNN"])


def test_oracle_matches_reference_gapless_extender_unit_tests():
    reference_gapless_cases(capi.Engine(lib=util.ORACLE_LIB))


# ---- random haplotype graphs: the engine against the oracle ----------------------------------------------------------

def random_haplotype_case(rng, n_reads=40, n_haplotypes=4, chain_nodes=14):
    """A bubble chain with n_haplotypes random threads; reads sampled from a thread (either strand) with substitutions;
    seeds at true positions plus a few false ones."""
    bases = "ACGT"
    nodes, chain, bubbles = [], [], []
    for c in range(chain_nodes):
        nodes.append("".join(bases[i] for i in rng.integers(0, 4, int(rng.integers(3, 12)))))
        chain.append(len(nodes) - 1)
        if c + 1 < chain_nodes and rng.random() < 0.6:
            alts = []
            for _ in range(int(rng.integers(2, 4))):
                nodes.append("".join(bases[i] for i in rng.integers(0, 4, int(rng.integers(1, 4)))))
                alts.append(len(nodes) - 1)
            if rng.random() < 0.3:
                alts.append(None)                      # a deletion allele: skip the bubble
            bubbles.append(alts)
        else:
            bubbles.append(None)
    threads = []
    for _ in range(n_haplotypes):
        t = []
        first = int(rng.integers(0, 3)); last = chain_nodes - int(rng.integers(0, 3))
        for c in range(first, last):
            t.append(2 * chain[c])
            if c + 1 < last and bubbles[c]:
                a = bubbles[c][int(rng.integers(0, len(bubbles[c])))]
                if a is not None:
                    t.append(2 * a)
        threads.append(t)

    def comp(s):
        return s[::-1].translate(str.maketrans("ACGT", "TGCA"))

    problems = []
    for _ in range(n_reads):
        t = threads[int(rng.integers(0, len(threads)))]
        if rng.random() < 0.5:
            t = [o ^ 1 for o in reversed(t)]
        seq = "".join(nodes[o // 2] if not o & 1 else comp(nodes[o // 2]) for o in t)
        starts = np.cumsum([0] + [len(nodes[o // 2]) for o in t])
        L = int(rng.integers(5, min(60, len(seq)) + 1)); a = int(rng.integers(0, len(seq) - L + 1))
        read = list(seq[a:a + L])
        for k in range(L):
            r = rng.random()
            if r < 0.06:
                read[k] = bases[int(rng.integers(0, 4))]
            elif r < 0.07:
                read[k] = "N"
        seeds = []
        for _ in range(int(rng.integers(1, 6))):
            ro = int(rng.integers(0, L)); g = a + ro
            k = int(np.searchsorted(starts, g, side="right") - 1)
            seeds.append((t[k], ro - (g - int(starts[k]))))
        if rng.random() < 0.3:                          # a false seed
            o = int(rng.integers(0, 2 * len(nodes)))
            seeds.append((o, int(rng.integers(0, L)) - int(rng.integers(0, len(nodes[o // 2])))))
        seeds = list(dict.fromkeys(seeds))              # a cluster is a set
        problems.append(dict(read="".join(read), seeds=seeds, max_mismatches=int(rng.integers(0, 5)),
                             overlap_threshold=float(rng.choice([0.1, 0.8, 0.9])), trim=bool(rng.random() < 0.8)))
    return nodes, threads, problems


def compare_engines(lib, seeds, n_reads=40):
    ora = capi.Engine(lib=util.ORACLE_LIB); eng = capi.Engine(lib=lib) if lib else capi.Engine()
    total = 0; full = 0
    for s in seeds:
        rng = np.random.default_rng(s)
        nodes, threads, problems = random_haplotype_case(rng, n_reads=n_reads)
        a = ora.gapless_extend(ora.haplo_index(nodes, threads), problems)
        b = eng.gapless_extend(eng.haplo_index(nodes, threads), problems)
        for x, y in zip(a, b):
            assert x.dtype == y.dtype and len(x) == len(y) and (x == y).all(), (s, x[:3], y[:3])
        assert (a[0]["status"] == 0).all()
        total += int(a[0]["n_ext"].sum()); full += int(a[0]["full_length"].sum())
    return total, full


def many_partial_extensions_case():
    """One read whose 14 seeds each give a distinct partial extension (more winners than the kernel's hot scratch slab holds)."""
    rng = np.random.default_rng(77)
    nodes = ["".join("ACGT"[i] for i in rng.integers(0, 4, 9)) for _ in range(30)]
    threads = [[2 * i for i in range(30)]]
    chunks, seeds, pos = [], [], 0
    for k in range(14):
        chunks.append(nodes[2 * k][1:8]); seeds.append((2 * (2 * k), pos - 1)); pos += 7          # bases 1..7 of every other node
        chunks.append("x"); pos += 1
    return nodes, threads, [dict(read="".join(chunks), seeds=seeds, max_mismatches=0)]


def check_many_partial_extensions(lib):
    nodes, threads, problems = many_partial_extensions_case()
    ora = capi.Engine(lib=util.ORACLE_LIB); eng = capi.Engine(lib=lib) if lib else capi.Engine()
    a = ora.gapless_extend(ora.haplo_index(nodes, threads), problems)
    b = eng.gapless_extend(eng.haplo_index(nodes, threads), problems)
    assert a[0]["n_ext"][0] == 14 and not a[0]["full_length"][0]
    for x, y in zip(a, b):
        assert len(x) == len(y) and (x == y).all()


def test_emulated_gapless_kernel_matches_reference_unit_tests_and_oracle():
    import subprocess
    subprocess.check_call(["make", "-s", "emu"], cwd=util.ROOT)
    reference_gapless_cases(capi.Engine(lib=util.EMU_LIB))
    check_many_partial_extensions(util.EMU_LIB)
    total, full = compare_engines(util.EMU_LIB, range(100, 140))
    assert total > 1500 and full > 300


@pytest.mark.gpu
def test_hip_gapless_matches_reference_unit_tests_and_oracle():
    reference_gapless_cases(capi.Engine())
    check_many_partial_extensions(None)
    total, full = compare_engines(None, range(200, 260), n_reads=400)
    assert total > 20000 and full > 4000


# ---- the C++ host shim (vg_amd/host/gbwt_extender.hpp), driven like src/unittest/gbwt_extender.cpp drives vg's class ------

def shim_extend(engine_lib, read, seeds, error_bound, overlap_threshold=0.8):
    import ctypes, json
    h = util.host()
    h.vgh_gapless_create.restype = ctypes.c_void_p
    h.vgh_gapless_create.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int32), ctypes.c_int]
    h.vgh_gapless_destroy.argtypes = [ctypes.c_void_p]
    h.vgh_gapless_extend.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int64), ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                     ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t]
    al = util.HostAligner(engine_lib)
    g = h.vgh_graph_create()
    try:
        for i, s in enumerate(TOY_NODES):
            assert h.vgh_graph_add_node(g, i + 1, s.encode()) == 0
        for a, b in [(1, 2), (1, 4), (1, 6), (2, 3), (2, 4), (3, 5), (4, 5), (5, 6), (6, 7), (6, 8), (7, 9), (8, 9)]:
            assert h.vgh_graph_add_edge(g, a, b) == 0
        threads = [SHORT_PATH, ALT_PATH, SHORT_PATH]
        flat = [2 * n for t in threads for n in t]; off = np.concatenate([[0], np.cumsum([len(t) for t in threads])])
        x = h.vgh_gapless_create(al.ptr, g, (ctypes.c_int64 * len(flat))(*flat), (ctypes.c_int32 * len(off))(*[int(v) for v in off]), len(threads))
        assert x, h.vgh_last_error().decode()
        try:
            sd = [v for (node_id, rev, offset), ro in seeds for v in (node_id, int(rev), offset, ro)]
            buf = ctypes.create_string_buffer(1 << 16)
            rc = h.vgh_gapless_extend(x, read.encode(), (ctypes.c_int64 * len(sd))(*sd), len(seeds), error_bound, overlap_threshold, 1, buf, len(buf))
            assert rc == 0, h.vgh_last_error().decode()
            return json.loads(buf.value.decode())
        finally:
            h.vgh_gapless_destroy(x)
    finally:
        h.vgh_graph_destroy(g)


def shim_cases(engine_lib):
    F = False
    # "read matches with errors" (:897): one full-length extension with one mismatch that contains both seeds
    out = shim_extend(engine_lib, "GGAGTAC", [((5, F, 0), 4), ((4, F, 2), 3)], 1)
    assert out["full_length"] and len(out["extensions"]) == 1
    e = out["extensions"][0]
    assert e["left_full"] and e["right_full"] and e["mismatches"] == 1 and e["contains_all_seeds"]
    assert e["alignment"]["score"] == 6 * 1 - 4 + 2 * 5
    maps = e["alignment"]["path"]["mapping"]
    assert [m["position"]["node_id"] for m in maps] == [1, 4, 5, 6, 7]
    assert [(x["from_length"], x["to_length"], x["sequence"]) for x in maps[1]["edit"]] == [(1, 1, ""), (1, 1, "A"), (1, 1, "")]
    # "trim right flank" (:1082): a partial extension starting at read offset 1
    out = shim_extend(engine_lib, "xAGGGTxAx", [((4, F, 2), 4)], 1)
    assert not out["full_length"] and len(out["extensions"]) == 1
    e = out["extensions"][0]
    assert e["read_begin"] == 1 and e["read_end"] == 6 and not e["left_full"] and not e["right_full"]
    assert [m["position"]["node_id"] for m in e["alignment"]["path"]["mapping"]] == [2, 4, 5]


def test_host_shim_gapless_extender_on_the_oracle():
    shim_cases(util.ORACLE_LIB)


@pytest.mark.gpu
def test_host_shim_gapless_extender_on_hip():
    shim_cases(util.ENGINE_LIB)


def many_haplotypes(lib, n_haplotypes=300):
    """Hundreds of haplotypes over a small bubble chain: the visits of a node leave through the same edge in long runs, so the
    engine's index stores most record bodies run-length encoded (GBWT's own form; gapless_device.hpp) — the oracle's index stays
    uncompressed, and both must give the same extensions."""
    rng = np.random.default_rng(77)
    tot = 0
    for rep in range(4):
        nodes, threads, problems = random_haplotype_case(rng, n_reads=150, n_haplotypes=n_haplotypes, chain_nodes=18)
        eng = capi.Engine(lib=lib); ora = capi.Engine(lib=util.ORACLE_LIB)
        a = eng.gapless_extend(eng.haplo_index(nodes, threads), problems)
        b = ora.gapless_extend(ora.haplo_index(nodes, threads), problems)
        # problem by problem (with this much diversity a search may outgrow the engine's per-seed limits: VGK_ETOOBIG for that read, none
        # of the others moves)
        (ra, ea, na, ma), (rb, eb, nb, mb) = a, b
        declined = 0
        for i in range(len(ra)):
            if ra["status"][i] != 0:
                assert ra["status"][i] == -7; declined += 1; continue
            assert rb["status"][i] == 0 and ra["n_ext"][i] == rb["n_ext"][i] and ra["full_length"][i] == rb["full_length"][i], (rep, i)
            for k in range(ra["n_ext"][i]):
                x, y = ea[ra["ext_begin"][i] + k], eb[rb["ext_begin"][i] + k]
                for f in ("path_len", "offset", "read_begin", "read_end", "n_mismatches", "score", "left_full", "right_full"):
                    assert x[f] == y[f], (rep, i, k, f)
                assert (x["state"] == y["state"]).all(), (rep, i, k)
                assert (na[x["path_begin"]:x["path_begin"] + x["path_len"]] == nb[y["path_begin"]:y["path_begin"] + y["path_len"]]).all()
                assert (ma[x["mism_begin"]:x["mism_begin"] + x["n_mismatches"]] == mb[y["mism_begin"]:y["mism_begin"] + y["n_mismatches"]]).all()
        assert declined <= len(ra) // 10        # (more haplotypes, more branching: a few more searches outgrow the per-seed limits)
        tot += int(ra["n_ext"].sum())
    assert tot > 300


def test_run_length_encoded_records_with_many_haplotypes(monkeypatch):
    many_haplotypes(util.EMU_LIB)
    monkeypatch.setenv("VGAMD_HAPLO_NO_RLE", "1")        # the byte-per-visit form of the same index gives the same answers
    many_haplotypes(util.EMU_LIB)


@pytest.mark.gpu
def test_run_length_encoded_records_with_many_haplotypes_on_the_gpu():
    many_haplotypes(util.ENGINE_LIB, 2000)


def test_output_arrays_too_small_are_reported_not_overrun():
    """the caller's extension array holds 20 of the ~55 extensions: VGK_EOPS from the engine (emulated) and from the oracle alike"""
    import subprocess
    from util import EMU_LIB, ORACLE_LIB, ROOT
    from vg_amd import workloads
    subprocess.check_call(["make", "-s", "emu"], cwd=ROOT)
    wl = workloads.GaplessWorkload(50, seed=5, graph_bp=30000)
    for lib in (EMU_LIB, ORACLE_LIB):
        eng = capi.Engine(capi.Scoring.simple(1, 4, 6, 1, 5), lib=lib)
        idx = eng.haplo_index(wl.nodes, wl.threads)
        res, ext, _, _ = eng.gapless_extend(idx, wl.gs)
        assert len(ext) > 20
        small = workloads.GaplessWorkload(50, seed=5, graph_bp=30000).gs
        small.ext_cap = 20
        with pytest.raises(capi.VgkError, match="too small"):
            eng.gapless_extend(idx, small)


def test_reads_gathered_or_uploaded_in_a_row_give_the_same_sets(monkeypatch):
    """vgk_gapless_extend uploads reads and seeds straight from the caller's buffers when they lie behind each other in problem order
    (GaplessSet builds them so) and gathers them into staging otherwise (VGAMD_GAPLESS_GATHER forces that path): same results, and a
    batch whose reads are NOT in a row (problem order reversed over the same buffers) takes the gather path by itself"""
    subprocess.check_call(["make", "-s", "emu"], cwd=util.ROOT)
    rng = np.random.default_rng(77)
    nodes, threads, problems = random_haplotype_case(rng, n_reads=300)
    eng = capi.Engine(lib=util.EMU_LIB)
    idx = eng.haplo_index(nodes, threads)
    gs = capi.GaplessSet.from_lists(problems)
    a = eng.gapless_extend(idx, gs)
    monkeypatch.setenv("VGAMD_GAPLESS_GATHER", "1")
    b = eng.gapless_extend(idx, gs)
    monkeypatch.delenv("VGAMD_GAPLESS_GATHER")
    for x, y in zip(a, b):
        assert x.tobytes() == y.tobytes()
    back = capi.GaplessSet.from_lists(problems)
    back.array[:] = back.array[::-1].copy()                            # same buffers, problems in reverse order: no longer in a row
    c = eng.gapless_extend(idx, back)
    ra, rc = a[0], c[0][::-1]
    assert (ra["status"] == rc["status"]).all() and (ra["n_ext"] == rc["n_ext"]).all() and (ra["full_length"] == rc["full_length"]).all()
    for i in range(len(ra)):
        ea = a[1][ra["ext_begin"][i]:ra["ext_begin"][i] + ra["n_ext"][i]]; ec = c[1][rc["ext_begin"][i]:rc["ext_begin"][i] + rc["n_ext"][i]]
        for f in ("offset", "read_begin", "read_end", "score", "path_len", "n_mismatches", "state"):
            assert (ea[f] == ec[f]).all(), (i, f)


# Unary runs merged at index build (gapless_device.hpp GMerge): on a chopped variation graph most nodes sit in runs, the search walks a fraction of
# the records, and every set still equals the oracle's node-by-node search — with the seeds that branched run again on the original index.
def merged_runs_equal_the_node_by_node_search(lib, n_reads, graph_bp, monkeypatch):
    from vg_amd import workloads
    wl = workloads.GaplessWorkload(n_reads, seed=17, graph_bp=graph_bp, inserted_reads=0.2)
    ora = capi.Engine(lib=util.ORACLE_LIB)
    want = ora.gapless_extend(ora.haplo_index(wl.nodes, wl.threads), wl.gs)
    outs = []
    for merge in (True, False):
        if merge:
            monkeypatch.setenv("VGAMD_HAPLO_MERGE", "1")                     # (off by default: DESIGN.md §28.3)
        else:
            monkeypatch.delenv("VGAMD_HAPLO_MERGE")
        eng = capi.Engine(lib=lib)
        hi = eng.haplo_index(wl.nodes, wl.threads)
        got = eng.gapless_extend(hi, wl.gs)
        outs.append((hi.search_nodes(), eng.gapless_last_redone(), got))
    (m_nodes, m_redone, with_runs), (p_nodes, p_redone, plain) = outs
    assert p_nodes == len(wl.nodes) and p_redone == 0
    assert m_nodes < 0.7 * p_nodes, (m_nodes, p_nodes)                      # (a SNP every 100 bases here: runs of two or three nodes; configs[2]'s graph: seven or eight)
    assert 0 < m_redone < 0.3 * len(wl.gs.seeds)                            # ties at the top: the reads with an inserted base (a fifth here), whose extensions run on misaligned past it
    for got in (with_runs, plain):
        for a, b in zip(got, want):
            assert len(a) == len(b) and a.tobytes() == b.tobytes()
    return m_nodes, p_nodes, m_redone


def test_emulated_merged_runs_equal_the_node_by_node_search(monkeypatch):
    subprocess.check_call(["make", "-s", "emu"], cwd=util.ROOT)
    merged_runs_equal_the_node_by_node_search(util.EMU_LIB, 3000, 120_000, monkeypatch)


@pytest.mark.gpu
def test_hip_merged_runs_equal_the_node_by_node_search(monkeypatch):
    merged_runs_equal_the_node_by_node_search(util.ENGINE_LIB, 200_000, 2_000_000, monkeypatch)
