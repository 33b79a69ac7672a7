"""Minimizer seeding (vgk_minimizer_index_create / vgk_minimizer_seeds): the (k, w)-minimizers of reads, looked up in the index of the
haplotype threads' minimizers, as seeds of the extension stage (MinimizerMapper::find_minimizers / find_seeds over gbwtgraph's
MinimizerIndex, src/minimizer_mapper.cpp:3918-3965, :4109-4290).  gbwtgraph is not in the snapshot; the scheme is pinned on the one
MinimizerIndex file the reference keeps (test/primers/y.min -> tests/golden/ref_primers_y.json, last tests of this file); what becomes
of a read's minimizers after the lookup is PARITY-UNPINNED.  Three constructions: this file's brute force (every k-mer hashed, every window scanned, the index a dict), the oracle's
(oracle/vgo_minimizer.c), the engine's (ring + hash table on the device; the emulator here, the MI355X in the gpu test) — and the
property the next stage relies on: a seed of a read sampled from a haplotype lies on the read's true diagonal."""
import json
import os
import subprocess

import numpy as np
import pytest

from util import EMU_LIB, ENGINE_LIB, ORACLE_LIB, ROOT
from vg_amd import capi, workloads

M64 = (1 << 64) - 1


@pytest.fixture(scope="module")
def emu_lib():
    subprocess.check_call(["make", "-s", "emu"], cwd=ROOT)
    return EMU_LIB


def wang(key):
    key = (~key + (key << 21)) & M64; key ^= key >> 24; key = (key + (key << 3) + (key << 8)) & M64; key ^= key >> 14
    key = (key + (key << 2) + (key << 4)) & M64; key ^= key >> 28; key = (key + (key << 31)) & M64
    return key


CODE = {"A": 0, "C": 1, "G": 2, "T": 3}
COMP = str.maketrans("ACGT", "TGCA")


def canonical(kmer):
    if any(c not in CODE for c in kmer):
        return None
    f = 0
    for c in kmer:
        f = (f << 2) | CODE[c]
    r = 0
    for c in kmer.translate(COMP)[::-1]:
        r = (r << 2) | CODE[c]
    hf, hr = wang(f), wang(r)
    return (hr, r, True) if hr < hf else (hf, f, False)


def minimizers(seq, k, w):
    """[(offset, key, reverse)]: leftmost smallest canonical hash of every window of w k-mers, each offset once"""
    n = len(seq) - k + 1
    if len(seq) < k + w - 1:
        return []
    km = [canonical(seq[p:p + k]) for p in range(n)]
    out = []
    for s in range(n - w + 1):
        best = None
        for p in range(s, s + w):
            if km[p] is not None and (best is None or km[p][0] < km[best][0]):
                best = p
        if best is not None and (not out or out[-1][0] != best):
            out.append((best, km[best][1], km[best][2]))
    return out


def oriented_seq(nodes, o):
    return nodes[o >> 1] if not (o & 1) else nodes[o >> 1].translate(COMP)[::-1]


def build_index(nodes, threads, k, w):
    index = {}
    for t in threads:
        seq = "".join(oriented_seq(nodes, o) for o in t)
        where = [(x, b) for x, o in enumerate(t) for b in range(len(nodes[o >> 1]))]
        for p, key, rev in minimizers(seq, k, w):
            if not rev:
                x, b = where[p]; pos = (t[x], b)
            else:
                x, b = where[p + k - 1]; pos = (t[x] ^ 1, len(nodes[t[x] >> 1]) - 1 - b)
            index.setdefault(key, set()).add(pos)
    return {key: sorted(v) for key, v in index.items()}


def seeds_of(read, index, nodes, k, w, hit_cap, truncated=None):
    """truncated: a list that receives True when the read reaches the cap of 64 seeds with hits still unexamined (VGK_MINIMIZERS_TRUNCATED)"""
    out = []; cut = False
    for p, key, rev in minimizers(read, k, w):
        hits = index.get(key, [])
        if not hits or len(hits) > hit_cap:
            continue
        for node, off in hits:
            if len(out) >= 64:
                cut = True
                break
            s = (node, p - off) if not rev else (node ^ 1, (p + k - 1) - (len(nodes[node >> 1]) - 1 - off))
            if s not in out:
                out.append(s)
    if truncated is not None:
        truncated.append(cut)
    return out


def sample_reads(rng, nodes, threads, n, L, error=0.01, with_n=0.02):
    reads, truth = [], []
    for _ in range(n):
        t = threads[int(rng.integers(0, len(threads)))]
        if rng.random() < 0.5:
            t = [o ^ 1 for o in reversed(t)]
        seq = "".join(oriented_seq(nodes, o) for o in t)
        a = int(rng.integers(0, len(seq) - L))
        rd = list(seq[a:a + L])
        for i in range(L):
            if rng.random() < error:
                rd[i] = "ACGT"[int(rng.integers(0, 4))]
        if rng.random() < with_n:
            rd[int(rng.integers(0, L))] = "N"
        reads.append("".join(rd)); truth.append((t, a))
    return reads, truth


def run(lib, seed, k, w, n_reads, hit_cap=500, L=100):
    wl = workloads.GaplessWorkload(4, seed=seed, graph_bp=8000, n_haplotypes=4, snp_every=40, indel_every=300)
    rng = np.random.default_rng(seed)
    reads, truth = sample_reads(rng, wl.nodes, wl.threads, n_reads, L)
    reads += ["ACGT" * 5, "", "A" * (k + w - 2)]                       # too short for a window: no minimizers
    index = build_index(wl.nodes, wl.threads, k, w)
    cut = []
    expected = [seeds_of(r, index, wl.nodes, k, w, hit_cap, cut) for r in reads]
    flat = np.frombuffer("".join(reads).encode(), dtype=np.uint8); off = np.concatenate([[0], np.cumsum([len(r) for r in reads])])
    eng = capi.Engine(lib=lib)
    mi = eng.minimizer_index(wl.nodes, wl.threads, k, w); hi = eng.haplo_index(wl.nodes, wl.threads)
    assert mi.keys == len(index)
    seed_off, seeds, mins = eng.minimizer_seeds(mi, hi, flat, off, hit_cap)
    for i, exp in enumerate(expected):
        got = [(int(s["node"]), int(s["diff"])) for s in seeds[seed_off[i]:seed_off[i + 1]]]
        assert got == exp, "read %d: %s vs %s" % (i, got[:6], exp[:6])
        assert mins[i] == len(minimizers(reads[i], k, w))
        assert bool(eng.minimizers_truncated[i]) == cut[i], "read %d: truncated flag" % i
    # nearly every seed of a sampled read lies on its true diagonal (base j of the read is base a + j of the thread it came from); the rest
    # are hits that are just as true on another haplotype: the other allele of a SNP whose alleles agree with the read, a k-mer that ends on the
    # first base behind an insertion the read carries
    lens = [len(s) for s in wl.nodes]
    on_diagonal = total = with_seeds = 0
    for i, (t, a) in enumerate(truth):
        starts = np.concatenate([[0], np.cumsum([lens[o >> 1] for o in t])])
        want = {(o, int(starts[x]) - a) for x, o in enumerate(t)}       # node o's first base is thread base starts[x] = read offset starts[x] - a; diff = read - node offset
        got = expected[i]
        with_seeds += bool(got)
        for node, diff in got:
            total += 1
            on_diagonal += (node, diff) in want
    assert with_seeds > 0.9 * n_reads and on_diagonal > 0.9 * total, (with_seeds, on_diagonal, total)
    return eng, mi, hi


@pytest.mark.parametrize("lib_name", ["oracle", "emu"])
def test_seeds_equal_the_brute_force_construction(lib_name, emu_lib):
    lib = ORACLE_LIB if lib_name == "oracle" else emu_lib
    run(lib, 1, 15, 6, 120)
    run(lib, 2, 29, 11, 60)
    run(lib, 3, 11, 4, 60, hit_cap=2)
    run(lib, 4, 31, 32, 40)
    run(lib, 6, 25, 64, 20, L=250)
    run(lib, 5, 21, 7, 20, L=1500)


def test_seeds_feed_the_extension_stage(emu_lib):
    """reads -> seeds -> gapless extension: the clusters the seeding makes resolve the reads (error-free reads: full-length extensions)"""
    wl = workloads.GaplessWorkload(4, seed=5, graph_bp=20000, n_haplotypes=4)
    rng = np.random.default_rng(5)
    reads, truth = sample_reads(rng, wl.nodes, wl.threads, 150, 150, error=0.0, with_n=0.0)
    flat = np.frombuffer("".join(reads).encode(), dtype=np.uint8); off = np.concatenate([[0], np.cumsum([len(r) for r in reads])])
    for lib in (emu_lib, ORACLE_LIB):
        eng = capi.Engine(capi.Scoring.simple(1, 4, 6, 1, 5), lib=lib)
        mi = eng.minimizer_index(wl.nodes, wl.threads); hi = eng.haplo_index(wl.nodes, wl.threads)
        seed_off, seeds, mins = eng.minimizer_seeds(mi, hi, flat, off)
        assert (np.diff(seed_off) > 0).all() and (mins >= 150 // 11 - 2).all()
        gs = capi.GaplessSet(flat, off, seeds, seed_off)
        res, ext, nodes, mism = eng.gapless_extend(hi, gs)
        assert (res["status"] == 0).all() and (res["full_length"] == 1).all()
        assert (ext["score"][res["ext_begin"]] == 150 + 10).all()


@pytest.mark.gpu
def test_seeds_on_the_gpu_equal_the_brute_force_construction():
    run(ENGINE_LIB, 7, 29, 11, 400)
    run(ENGINE_LIB, 8, 15, 6, 400)
    run(ENGINE_LIB, 9, 31, 32, 150)
    run(ENGINE_LIB, 10, 21, 7, 60, L=1500)          # many rounds per read, more than 64 distinct seeds


def seeded_equals_via_host(lib, n_reads, seed):
    """vgk_gapless_extend_seeded (clusters left on the device by the seeding) = vgk_gapless_extend on the same seeds brought through the host,
    byte for byte; and the stage built on it gives the oracle pipeline's per-read totals"""
    from vg_amd import pipeline
    wl = workloads.GaplessWorkload(4, seed=seed, graph_bp=30000, n_haplotypes=4)
    rng = np.random.default_rng(seed)
    reads, truth = sample_reads(rng, wl.nodes, wl.threads, n_reads, 150, error=0.01, with_n=0.05)
    reads[3] = reads[3][:70] + "ACG" + reads[3][70:147]                     # an insertion: tails for the stage below
    flat = np.frombuffer("".join(reads).encode(), dtype=np.uint8); off = np.concatenate([[0], np.cumsum([len(r) for r in reads])])
    eng = capi.Engine(capi.Scoring.simple(1, 4, 6, 1, 5), lib=lib)
    mi = eng.minimizer_index(wl.nodes, wl.threads); hi = eng.haplo_index(wl.nodes, wl.threads)
    seed_off, seeds, _ = eng.minimizer_seeds(mi, hi, flat, off)
    via_host = eng.gapless_extend(hi, capi.GaplessSet(flat, off, seeds, seed_off))
    so2, none, _ = eng.minimizer_seeds(mi, hi, flat, off, keep_on_device=True)
    assert (so2 == seed_off).all() and len(none) == 0
    seeded = eng.gapless_extend_seeded(hi, len(reads), int(seed_off[-1]))
    for x, y in zip(seeded, via_host):
        assert x.tobytes() == y.tobytes()
    # without a seeding call before it, or with another index, the seeded form refuses
    eng2 = capi.Engine(capi.Scoring.simple(1, 4, 6, 1, 5), lib=lib)
    with pytest.raises(capi.VgkError):
        eng2.gapless_extend_seeded(eng2.haplo_index(wl.nodes, wl.threads), len(reads), 10)
    olen = np.repeat(np.array([len(s) for s in wl.nodes]), 2)
    gs = capi.GaplessSet(flat, off, seeds, seed_off)
    eng.minimizer_seeds(mi, hi, flat, off, keep_on_device=True)
    a = pipeline.align_stage_native(eng, hi, olen, gs, seeded=int(seed_off[-1]))
    ora = capi.Engine(capi.Scoring.simple(1, 4, 6, 1, 5), lib=ORACLE_LIB)
    oso, oseeds, _ = ora.minimizer_seeds(ora.minimizer_index(wl.nodes, wl.threads), ora.haplo_index(wl.nodes, wl.threads), flat, off)
    b = pipeline.align_stage(ora, ora.haplo_index(wl.nodes, wl.threads), olen, capi.GaplessSet(flat, off, oseeds, oso))
    assert (a["read_score"] == b["read_score"]).all()
    # VGK_GAPLESS_DEFER: the sets come down while the tail stage runs; complete (and the same bytes) when that call returns ...
    eng.minimizer_seeds(mi, hi, flat, off, keep_on_device=True)
    c = pipeline.align_stage_device(eng, hi, gs, seeded=int(seed_off[-1]))
    for name, y in zip(("res", "ext", "nodes"), via_host):
        assert c[name].tobytes() == y.tobytes(), name
    assert (c["read_score"] == b["read_score"]).all()
    # ... or after vgk_gapless_fetch_deferred; a call that is not followed by either is finished by the next extension call
    eng.minimizer_seeds(mi, hi, flat, off, keep_on_device=True)
    d = eng.gapless_extend_seeded(hi, len(reads), int(seed_off[-1]), defer=True)
    eng.gapless_fetch_deferred()
    for x, y in zip(d, via_host):
        assert x.tobytes() == y.tobytes()
    eng.minimizer_seeds(mi, hi, flat, off, keep_on_device=True)
    e = eng.gapless_extend_seeded(hi, len(reads), int(seed_off[-1]), defer=True)
    eng.gapless_extend(hi, capi.GaplessSet(flat, off, seeds, seed_off))
    for x, y in zip(e, via_host):
        assert x.tobytes() == y.tobytes()


def test_clusters_that_stay_on_the_device(emu_lib):
    seeded_equals_via_host(emu_lib, 200, 21)


@pytest.mark.gpu
def test_clusters_that_stay_on_the_device_on_the_gpu():
    seeded_equals_via_host(ENGINE_LIB, 5000, 22)


# ---- pinned on the reference's own index file ------------------------------------------------------------------------------------
def primers_fixture():
    """test/primers/y.{gbwt,gg,min} of the reference, decoded by tests/golden/extract_primers_fixture.py"""
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_primers_y.json")))
    nodes = fx["node_sequences"]
    first = fx["gbwt_header"]["offset"] + 1                                  # GBWT node of (first node id, forward)
    threads = [[x - first for x in t] for t in fx["gbwt_threads"][0::2]]    # the odd sequences are the even ones on the other strand
    mi = fx["minimizer_index"]
    want = sorted((e["key"], 2 * (e["id"] - first // 2) + e["is_reverse"], e["offset"]) for e in mi["entries"])
    return fx, nodes, threads, mi["k"], mi["w"], want


def index_equals_the_reference_file(lib):
    fx, nodes, threads, k, w, want = primers_fixture()
    assert (k, w) == (31, 50) and len(want) == 62
    eng = capi.Engine(lib=lib)
    mi = eng.minimizer_index(nodes, threads, k, w)
    hits = mi.fetch()
    got = [(int(h["key"]), int(h["node"]), int(h["offset"])) for h in hits]
    assert got == want
    assert mi.keys == 62
    # the file's hash table: a key sits in the first free cell from hash & (capacity - 1) on, in insertion order — so every key's
    # cell is at or cyclically behind its home cell, and every cell between the two is taken
    cap = fx["minimizer_index"]["capacity"]
    taken = {e["cell"] for e in fx["minimizer_index"]["entries"]}
    for e in fx["minimizer_index"]["entries"]:
        home = wang(e["key"]) & (cap - 1)
        c = home
        while c != e["cell"]:
            assert c in taken; c = (c + 1) & (cap - 1)
    return eng, mi, nodes, threads, k, w


@pytest.mark.parametrize("lib_name", ["oracle", "emu"])
def test_index_of_the_primers_graph_is_the_reference_minimizer_index(lib_name, emu_lib):
    """y.min (gbwtgraph's MinimizerIndex of the graph in y.gg / y.gbwt, k = 31, w = 50) holds 62 keys with one position each: the
    index built here from the same graph and haplotypes holds exactly those keys at exactly those positions"""
    index_equals_the_reference_file(ORACLE_LIB if lib_name == "oracle" else emu_lib)
    fx, nodes, threads, k, w, want = primers_fixture()
    mine = build_index(nodes, threads, k, w)                               # this file's brute force agrees as well
    assert sorted((key, n, o) for key, v in mine.items() for n, o in v) == want


def wide_window_seeds(lib, n_reads=40):
    """k = 31, w = 50 (the long-read parameters of the file) through the seeding path: engine = brute force"""
    eng, mi, nodes, threads, k, w = index_equals_the_reference_file(lib)
    rng = np.random.default_rng(11)
    reads, truth = sample_reads(rng, nodes, threads, n_reads, 400, error=0.005, with_n=0.05)
    index = build_index(nodes, threads, k, w)
    flat = np.frombuffer("".join(reads).encode(), dtype=np.uint8); off = np.concatenate([[0], np.cumsum([len(r) for r in reads])])
    hi = eng.haplo_index(nodes, threads)
    seed_off, seeds, mins = eng.minimizer_seeds(mi, hi, flat, off, 500)
    some = 0
    for i, r in enumerate(reads):
        got = [(int(s["node"]), int(s["diff"])) for s in seeds[seed_off[i]:seed_off[i + 1]]]
        assert got == seeds_of(r, index, nodes, k, w, 500)
        assert mins[i] == len(minimizers(r, k, w))
        some += bool(got)
    assert some > n_reads // 2


@pytest.mark.parametrize("lib_name", ["oracle", "emu"])
def test_seeds_with_the_wide_window_of_the_reference_file(lib_name, emu_lib):
    wide_window_seeds(ORACLE_LIB if lib_name == "oracle" else emu_lib)


@pytest.mark.gpu
def test_reference_minimizer_index_and_wide_window_on_the_gpu():
    wide_window_seeds(ENGINE_LIB, 200)
    run(ENGINE_LIB, 12, 31, 64, 80, L=300)          # the widest window the kernel takes


def no_seed_batch(lib):
    """A batch whose reads have no seed at all (all N, too short for a window) through the resident path: the seeded extension call gives
    every read an empty set, as vgk_gapless_extend does for a problem without seeds — not an error (round 2's advisor finding: the seed
    buffer did not exist for a batch without seeds and the extension call read that as out of memory)."""
    wl = workloads.GaplessWorkload(4, seed=3, graph_bp=4000, n_haplotypes=4, snp_every=40, indel_every=300)
    eng = capi.Engine(lib=lib)
    mi = eng.minimizer_index(wl.nodes, wl.threads, 21, 7); hi = eng.haplo_index(wl.nodes, wl.threads)
    reads = ["N" * 60, "ACGT" * 3, "N" * 30 + "ACGTACGTAC" + "N" * 30]
    flat = np.frombuffer("".join(reads).encode(), dtype=np.uint8); off = np.concatenate([[0], np.cumsum([len(r) for r in reads])])
    seed_off, seeds, mins = eng.minimizer_seeds(mi, hi, flat, off, keep_on_device=True)
    assert seed_off[-1] == 0 and not eng.minimizers_truncated.any()
    res, ext, nodes, mism = eng.gapless_extend_seeded(hi, len(reads), 0)
    assert len(ext) == 0 and (res["n_ext"] == 0).all() and (res["status"] == 0).all()


def test_a_batch_without_seeds_extends_to_empty_sets(emu_lib):
    no_seed_batch(emu_lib)


@pytest.mark.gpu
def test_a_batch_without_seeds_extends_to_empty_sets_on_the_gpu():
    no_seed_batch(ENGINE_LIB)
