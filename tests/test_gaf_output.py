"""GAF records of composed alignments (vg_amd/host/gaf_output.cpp; SURVEY §8(f) N4, the output half) against what the reference's own tests hold about them
(tests/golden/ref_gaf.json, transcribed by tests/golden/extract_gaf_tests.py: two unit tests of src/unittest/alignment.cpp and the records of
test/surject/opposite_strands.gaf), and — for whole batches — against the difference string's own meaning: applied to the path's bases it gives back the read."""
import json
import os
import re

import numpy as np
import pytest

import util
from vg_amd import capi, pipeline

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_gaf.json")
MATCH, MISMATCH, INS, DEL = 0, 1, 2, 3


def flat_case(case):
    """an Alignment of the fixture -> the flat arrays vgk_chain_stitch would have left (node = 2 * index + reverse)"""
    ids = sorted(int(k) for k in case["graph"])
    index = {i: k for k, i in enumerate(ids)}
    node_seq = "".join(case["graph"][str(i)] for i in ids)
    node_off = np.cumsum([0] + [len(case["graph"][str(i)]) for i in ids]).astype(np.uint64)
    maps = np.zeros(len(case["mappings"]), dtype=capi.CHAIN_MAPPING_DT)
    runs = []
    for k, m in enumerate(case["mappings"]):
        maps[k]["node"] = 2 * index[m["node_id"]] + (1 if m["is_reverse"] else 0); maps[k]["offset"] = m["offset"]; maps[k]["edit_begin"] = len(runs)
        for f, t, s in m["edits"]:
            kind = MATCH if f == t and not s else MISMATCH if f == t else INS if f == 0 else DEL
            runs.append(((t if kind == INS else f) << 2) | kind)
        maps[k]["n_edits"] = len(runs) - maps[k]["edit_begin"]
    res = np.zeros(1, dtype=capi.CHAIN_RESULT_DT)
    res[0]["status"] = 0 if case["mappings"] else -1
    res[0]["n_mappings"] = len(maps); res[0]["n_edits"] = len(runs)
    seq = np.frombuffer(case["sequence"].encode(), dtype=np.uint8)
    return seq, np.array([0, len(seq)], dtype=np.uint64), res, maps, np.array(runs, dtype=np.uint32), np.frombuffer(node_seq.encode(), dtype=np.uint8), node_off, np.array(ids, dtype=np.int64)


def test_records_equal_what_the_reference_tests_hold():
    cases = json.load(open(GOLD))["cases"]
    assert len(cases) == 4
    for c in cases:
        seq, seq_off, res, maps, runs, node_seq, node_off, ids = flat_case(c)
        if not len(ids):
            node_off = np.zeros(1, dtype=np.uint64)
        line = pipeline.gaf_lines(seq, seq_off, res, maps, runs, node_seq, node_off, names=[c["name"].encode()], node_ids=ids if len(ids) else None,
                                  mapq=[c.get("mapq", 0)])[0].decode()
        f = line.split("\t")
        if c.get("expect_line"):
            assert line == c["expect_line"], c["name"]
        e = c.get("expect")
        if e:
            assert f[0] == e["query_name"] and int(f[1]) == e["query_length"] and int(f[2]) == e["query_start"] and int(f[3]) == e["query_end"] and f[4] == "+", c["name"]
            assert re.findall(r"[<>]\d+", f[5]) == e["path"] and (f[5] == "*") == (not e["path"]), c["name"]
            if "path_length" in e:
                assert (int(f[6]), int(f[7]), int(f[8])) == (e["path_length"], e["path_start"], e["path_end"]), c["name"]
            assert f[-1] == "cs:Z:" + e["cs"], c["name"]


def apply_cs(cs, path_bases):
    """the read a difference string describes over the bases of its path interval"""
    read, at = [], 0
    for op, arg in re.findall(r"([:*+-])([0-9]+|[A-Z]+)", cs):
        if op == ":":
            n = int(arg); read.append(path_bases[at:at + n]); at += n
        elif op == "*":
            assert path_bases[at] == arg[0]; read.append(arg[1]); at += 1
        elif op == "+":
            read.append(arg)
        else:
            assert path_bases[at:at + len(arg)] == arg; at += len(arg)
    return "".join(read), at


def gaf_of_a_long_read_batch(lib):
    """the chain stage's composed alignments of a small long-read batch as GAF: every record's difference string, applied to its path's interval, is the read"""
    from test_longread_stage import native_stage, read_sequences
    wl, out, _ = native_stage(lib, 8, 2500, 11, 0.02, compose=True)
    res, maps, runs = out["alignments"]
    nodes = wl.nodes
    reads = read_sequences(wl)
    node_seq = np.frombuffer("".join(nodes).encode(), dtype=np.uint8)
    node_off = np.cumsum([0] + [len(s) for s in nodes]).astype(np.uint64)
    seqs = np.frombuffer("".join(reads).encode(), dtype=np.uint8)
    seq_off = np.cumsum([0] + [len(r) for r in reads]).astype(np.uint64)
    lines = pipeline.gaf_lines(seqs, seq_off, res, maps, runs, node_seq, node_off, names=[b"read%d" % r for r in range(len(reads))],
                               score=np.asarray(out["chain_score"], dtype=np.int32), mapq=[60] * len(reads))
    comp = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}
    checked = 0
    for r, line in enumerate(lines):
        f = line.decode().split("\t")
        assert f[0] == "read%d" % r and int(f[1]) == len(reads[r]) and (int(f[2]), int(f[3])) == (0, len(reads[r])) and f[4] == "+" and f[11] == "60"
        assert f[5] != "*"
        bases = "".join(nodes[int(i) - 1] if o == ">" else "".join(comp[c] for c in reversed(nodes[int(i) - 1])) for o, i in re.findall(r"([<>])(\d+)", f[5]))
        assert len(bases) == int(f[6])
        tags = {t.split(":", 2)[0]: t.split(":", 2)[2] for t in f[12:]}
        read, used = apply_cs(tags["cs"], bases[int(f[7]):int(f[8])])
        assert read == reads[r].upper() and used == int(f[8]) - int(f[7]), r
        assert int(tags["AS"]) == int(out["chain_score"][r])
        checked += 1
    return checked


def test_long_read_batch_as_gaf_on_the_emulated_kernels():
    import subprocess
    subprocess.check_call(["make", "-s", "emu"], cwd=util.ROOT)
    assert gaf_of_a_long_read_batch(util.EMU_LIB) == 8


@pytest.mark.gpu
def test_long_read_batch_as_gaf_on_hip():
    assert gaf_of_a_long_read_batch(util.ENGINE_LIB) == 8
