"""vgk_haplo_create_gbwt: the haplotype index read from a GBWT file.  gbwt is an absent submodule; the reference
keeps one GBWT for its own tests, test/primers/y.gbwt (with the graph's sequences in y.gg), which tests/golden/extract_primers_fixture.py
decodes — independently of the engine's reader (vg_amd/csrc/gbwt_file.cpp) — into tests/golden/ref_primers_y.json: the file image,
its records (edges and runs), the sequences followed out of them.  Pinned here:
  * the engine's reader finds the same haplotypes: extensions over the index read from the file = over the index built from the
    extracted threads = the oracle's;
  * the index the engine BUILDS (vgk_haplo_create: visits ordered like GBWT's) is the file's: the bidirectional search states it
    reports for an extension (vgk_extension.state) are the ranges gbwt's find() has over the FILE's records;
  * malformed images are refused."""
import json
import os
import subprocess

import numpy as np
import pytest

from util import EMU_LIB, ENGINE_LIB, ORACLE_LIB, ROOT
from vg_amd import capi

COMP = str.maketrans("ACGT", "TGCA")


@pytest.fixture(scope="module")
def emu_lib():
    subprocess.check_call(["make", "-s", "emu"], cwd=ROOT)
    return EMU_LIB


def fixture():
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_primers_y.json")))
    first = fx["gbwt_header"]["offset"] + 1
    threads = [[x - first for x in t] for t in fx["gbwt_threads"][0::2]]
    return fx, fx["node_sequences"], threads, bytes.fromhex(fx["gbwt_file_hex"]), first


class FileGBWT:
    """find() over the records as the file holds them (gbwt's search: the range of visits of the last node that arrive along the path)"""

    def __init__(self, fx):
        self.records, self.offset = fx["gbwt_records"], fx["gbwt_header"]["offset"]

    def record(self, node):
        return self.records[node - self.offset]

    def size(self, node):
        return sum(length for _, length in self.record(node)["runs"])

    def lf(self, node, i, to):
        """visits of `to` that come from the first i visits of `node`, behind the ones from smaller predecessors"""
        rec = self.record(node)
        ranks = [r for r, (n, _) in enumerate(rec["edges"]) if n == to]
        if not ranks:
            return None
        rank = ranks[0]
        seen, at = rec["edges"][rank][1], 0
        for r, length in rec["runs"]:
            if at >= i:
                break
            take = min(length, i - at)
            if r == rank:
                seen += take
            at += length
        return seen

    def find(self, path):
        node = path[0]
        lo, hi = 0, self.size(node) - 1
        for to in path[1:]:
            a, b = self.lf(node, lo, to), self.lf(node, hi + 1, to)
            if a is None or b - 1 < a:
                return None
            node, lo, hi = to, a, b - 1
        return node, lo, hi


def oriented_seq(nodes, o):
    return nodes[o >> 1] if not (o & 1) else nodes[o >> 1].translate(COMP)[::-1]


def sample(nodes, threads, rng, n, L):
    problems = []
    lens = [len(s) for s in nodes]
    for _ in range(n):
        t = threads[int(rng.integers(0, len(threads)))]
        if rng.random() < 0.5:
            t = [o ^ 1 for o in reversed(t)]
        seq = "".join(oriented_seq(nodes, o) for o in t)
        a = int(rng.integers(0, len(seq) - L))
        rd = list(seq[a:a + L])
        if rng.random() < 0.5:
            j = int(rng.integers(0, L)); rd[j] = "ACGT"[("ACGT".index(rd[j]) + 1) % 4]
        starts = np.concatenate([[0], np.cumsum([lens[o >> 1] for o in t])])
        x = int(np.searchsorted(starts, a + L // 2, side="right")) - 1       # a seed in the middle of the read, on its true diagonal
        problems.append({"read": "".join(rd), "seeds": [(t[x], int(starts[x]) - a)], "trim": True})
    return problems


def extend(eng, index, problems):
    res, ext, nodes, mism = eng.gapless_extend(index, problems)
    out = []
    for r in res:
        row = [int(r["status"]), int(r["full_length"])]
        for e in ext[r["ext_begin"]:r["ext_begin"] + r["n_ext"]]:
            row.append((tuple(int(x) for x in nodes[e["path_begin"]:e["path_begin"] + e["path_len"]]), int(e["offset"]), int(e["read_begin"]), int(e["read_end"]),
                        int(e["score"]), tuple(int(x) for x in e["state"]), tuple(int(x) for x in mism[e["mism_begin"]:e["mism_begin"] + e["n_mismatches"]])))
        out.append(row)
    return out


def check(lib, n=150):
    fx, nodes, threads, image, first = fixture()
    eng = capi.Engine(capi.Scoring.simple(1, 4, 6, 1, 5), lib=lib)
    from_file = eng.haplo_index_from_gbwt(nodes, image)
    from_threads = eng.haplo_index(nodes, threads)
    problems = sample(nodes, threads, np.random.default_rng(3), n, 60)
    a = extend(eng, from_file, problems)
    assert a == extend(eng, from_threads, problems)
    orc = capi.Engine(capi.Scoring.simple(1, 4, 6, 1, 5), lib=ORACLE_LIB)
    assert a == extend(orc, orc.haplo_index(nodes, threads), problems)
    # the states are the file's
    gb = FileGBWT(fx)
    checked = 0
    for row in a:
        assert row[0] == 0
        for path, offset, rb, re_, score, state, mm in row[2:]:
            fwd = gb.find([o + first for o in path])
            bwd = gb.find([(o ^ 1) + first for o in reversed(path)])
            assert fwd is not None and bwd is not None
            assert (state[0] + first, state[1], state[2]) == fwd, (path, state, fwd)
            assert (state[3] + first, state[4], state[5]) == bwd, (path, state, bwd)
            checked += 1
    assert checked >= n
    return eng, nodes, image


def test_find_over_the_file_records_counts_the_haplotypes():
    """the independent decoder itself: every extracted sequence is found in the records, as often as it occurs"""
    fx, nodes, threads, image, first = fixture()
    gb = FileGBWT(fx)
    seqs = fx["gbwt_threads"]
    for t in seqs:
        node, lo, hi = gb.find(t)
        assert hi - lo + 1 == sum(1 for u in seqs if u == t)
        assert gb.find(t[:5])[2] - gb.find(t[:5])[1] + 1 == sum(1 for u in seqs for i in range(len(u) - 4) if u[i:i + 5] == t[:5])


def test_index_read_from_the_reference_gbwt(emu_lib):
    eng, nodes, image = check(emu_lib)
    # malformed images
    lens = np.array([len(s) for s in nodes], dtype=np.uint32)
    for bad, code in ((image[:200], "EINVAL"), (image[:-900], "EINVAL"), (b"\\0" * 64, "EINVAL"), (image[:40] + bytes([3]) + image[41:], "EUNSUPPORTED"),
                      (image[:40] + bytes([6]) + image[41:], "EUNSUPPORTED")):
        with pytest.raises(capi.VgkError):
            eng.haplo_index_from_gbwt(nodes, bad)
    with pytest.raises(capi.VgkError):
        eng.haplo_index_from_gbwt(nodes[:-1], image)                # the node count must be the file's
    broken = bytearray(image); broken[0x198 + 8] = 0x7f              # the endmarker's outdegree
    with pytest.raises(capi.VgkError):
        eng.haplo_index_from_gbwt(nodes, bytes(broken))


def test_records_taken_in_place_equal_the_walked_threads(emu_lib, monkeypatch):
    """the default loader takes the records over as they lie in the file; VGAMD_GBWT_VIA_THREADS walks every sequence out with LF and builds
    the index from those.  Same extensions, same search states; and no corrupted body byte may take the in-place decoder down"""
    fx, nodes, threads, image, first = fixture()
    eng = capi.Engine(capi.Scoring.simple(1, 4, 6, 1, 5), lib=emu_lib)
    problems = sample(nodes, threads, np.random.default_rng(11), 120, 50)
    in_place = extend(eng, eng.haplo_index_from_gbwt(nodes, image), problems)
    monkeypatch.setenv("VGAMD_GBWT_VIA_THREADS", "1")
    assert in_place == extend(eng, eng.haplo_index_from_gbwt(nodes, image), problems)
    monkeypatch.delenv("VGAMD_GBWT_VIA_THREADS")
    rng = np.random.default_rng(5)
    raised = 0
    for _ in range(300):
        broken = bytearray(image)
        for _ in range(int(rng.integers(1, 4))):
            broken[int(rng.integers(0x100, len(image)))] = int(rng.integers(0, 256))
        try:
            eng.haplo_index_from_gbwt(nodes, bytes(broken))
        except capi.VgkError:
            raised += 1
    assert raised > 50


def test_gbz_container_decodes_to_the_graph_and_the_haplotypes(emu_lib):
    """y.giraffe.gbz holds the graph of y.gg and the haplotypes of y.gbwt: the engine's GBZ loader must hand back exactly what the golden
    script decoded from those two OTHER files"""
    fx, nodes, threads, image, first = fixture()
    eng = capi.Engine(lib=emu_lib)
    gbz = bytes.fromhex(fx["gbz_file_hex"])
    got_nodes, got_threads = eng.gbz_load(gbz)
    assert got_nodes == nodes
    assert got_threads == threads
    for bad in (gbz[:100], gbz[:0x840], gbz[:-700], b"GBZ " + b"\0" * 60, gbz[:0xc8] + b"\0" * 8 + gbz[0xd0:]):
        with pytest.raises(capi.VgkError):
            eng.gbz_load(bad)


@pytest.mark.gpu
def test_index_read_from_the_reference_gbwt_on_the_gpu():
    check(ENGINE_LIB, 400)
