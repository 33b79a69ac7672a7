#!/usr/bin/env python3
"""Transcribe what the reference's own tests hold about GAF records into JSON.

Run in the dev container only (needs /root/reference):

    python tests/golden/extract_gaf_tests.py      ->  tests/golden/ref_gaf.json

Sources (alignment_to_gaf itself is libvgio's — an empty submodule of the snapshot; these are the reference-held vectors for it):
  * /root/reference/src/unittest/alignment.cpp "Conversion to GAF removes an unused final node": the graph's create_node calls, the Alignment
    built by the protobuf setters, and the REQUIRE lines on the record (query interval, path steps, path length / start / end, the cs string);
  * the same file, "Unaligned sequences survive round-trip to GAF": an alignment without a path (no steps, cs = "+" + the read);
  * /root/reference/test/surject/opposite_strands.gfa + .gaf: two records of one read on either strand of a path of eight nodes — a data file of the
    reference's test suite; the whole line is the expectation.  The alignment it records is read back FROM the line (steps, start, a cs of one match run).
The script copies literals and evaluates the two expressions the first test states in terms of its own graph; it never executes reference code.

Every case: {"source", "name", "graph": {node id: sequence}, "sequence", "mappings": [{"node_id", "is_reverse", "offset", "edits": [[from, to, seq]...]}...],
"expect": {field: value ...} and/or "expect_line": the record's text}.
"""
import json
import os
import re

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_gaf.json")


def test_case(text, title):
    at = text.index('TEST_CASE("%s"' % title)
    line = text.count("\n", 0, at) + 1
    rest = text.find("\nTEST_CASE(", at + 1)
    body = text[at:rest if rest >= 0 else len(text)]
    return body, line


def built_alignment(body):
    """the protobuf setters of a test body, in order -> (graph, name, sequence, mappings)"""
    graph = {int(i): s for s, i in re.findall(r'graph\.create_node\("([ACGTN]+)",\s*(\d+)\)', body)}
    seq = re.search(r'aln\.set_sequence\("([ACGTN]*)"\)', body).group(1)
    name = re.search(r'aln\.set_name\("([^"]*)"\)', body)
    mappings = []
    tok = re.compile(r'(path->add_mapping\(\))|set_node_id\((\d+)\)|set_is_reverse\((true|false)\)|set_offset\((\d+)\)|(->add_edit\(\))|'
                     r'set_from_length\((\d+)\)|set_to_length\((\d+)\)|edit->set_sequence\("([ACGTN]*)"\)')
    for m in tok.finditer(body):
        if m.group(1): mappings.append({"node_id": 0, "is_reverse": False, "offset": 0, "edits": []})
        elif m.group(2): mappings[-1]["node_id"] = int(m.group(2))
        elif m.group(3): mappings[-1]["is_reverse"] = m.group(3) == "true"
        elif m.group(4): mappings[-1]["offset"] = int(m.group(4))
        elif m.group(5): mappings[-1]["edits"].append([0, 0, ""])
        elif m.group(6): mappings[-1]["edits"][-1][0] = int(m.group(6))
        elif m.group(7): mappings[-1]["edits"][-1][1] = int(m.group(7))
        elif m.group(8) is not None: mappings[-1]["edits"][-1][2] = m.group(8)
    return graph, (name.group(1) if name else ""), seq, mappings


def main():
    cases = []
    src = os.path.join(REF, "src/unittest/alignment.cpp")
    text = open(src).read()

    body, line = test_case(text, "Conversion to GAF removes an unused final node")
    graph, name, seq, mappings = built_alignment(body)
    steps = int(re.search(r'gaf\.path\.size\(\) == (\d+)', body).group(1))
    names = [re.search(r'gaf\.path\[%d\]\.name == "(\d+)"' % k, body).group(1) for k in range(steps)]
    revs = [re.search(r'gaf\.path\[%d\]\.is_reverse == (true|false)' % k, body).group(1) == "true" for k in range(steps)]
    # "size_t path_length = graph.get_length(graph.get_handle(1)) + graph.get_length(graph.get_handle(2));"  REQUIRE(gaf.path_length == path_length)
    plen_ids = [int(x) for x in re.findall(r'get_handle\((\d+)\)\)', re.search(r'size_t path_length = ([^;]+);', body).group(1))]
    assert "gaf.path_length == path_length" in body and "gaf.path_end == path_length" in body
    assert "gaf.path_start == aln.path().mapping(0).position().offset()" in body
    assert "gaf.query_start == 0" in body and "gaf.query_end == aln.sequence().size()" in body and "gaf.query_length == aln.sequence().size()" in body
    cs = re.search(r'true_difference_string = "([^"]+)"', body).group(1)
    path_length = sum(len(graph[i]) for i in plen_ids)
    cases.append({"source": "src/unittest/alignment.cpp:%d" % line, "name": name, "graph": {str(k): v for k, v in graph.items()}, "sequence": seq, "mappings": mappings,
                  "expect": {"query_name": name, "query_length": len(seq), "query_start": 0, "query_end": len(seq),
                             "path": [("<" if r else ">") + n for n, r in zip(names, revs)], "path_length": path_length,
                             "path_start": mappings[0]["offset"], "path_end": path_length, "cs": cs}})

    body, line = test_case(text, "Unaligned sequences survive round-trip to GAF")
    seq = re.search(r'aln\.set_sequence\("([ACGTN]*)"\)', body).group(1)
    name = re.search(r'aln\.set_name\("([^"]*)"\)', body).group(1)
    assert "gaf.path.size() == 0" in body and 'std::string expected_cs = "+" + aln.sequence();' in body
    cases.append({"source": "src/unittest/alignment.cpp:%d" % line, "name": name, "graph": {}, "sequence": seq, "mappings": [],
                  "expect": {"query_name": name, "query_length": len(seq), "query_start": 0, "query_end": len(seq), "path": [], "cs": "+" + seq}})

    gfa = os.path.join(REF, "test/surject/opposite_strands.gfa")
    graph = {}
    for l in open(gfa):
        f = l.rstrip("\n").split("\t")
        if f[0] == "S":
            graph[int(f[1])] = f[2]
    comp = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}
    for l in open(os.path.join(REF, "test/surject/opposite_strands.gaf")):
        f = l.rstrip("\n").split("\t")
        steps = re.findall(r'([<>])(\d+)', f[5])
        start, end = int(f[7]), int(f[8])
        assert f[12] == "cs:Z::%d" % (end - start)
        # the bases of the path's interval are the read; one match edit per step
        mappings, seq, at = [], "", 0
        for o, i in steps:
            s = graph[int(i)]
            if o == "<": s = "".join(comp[c] for c in reversed(s))
            lo, hi = max(start, at), min(end, at + len(s))
            if lo < hi:
                mappings.append({"node_id": int(i), "is_reverse": o == "<", "offset": lo - at, "edits": [[hi - lo, hi - lo, ""]]})
                seq += s[lo - at:hi - at]
            at += len(s)
        assert len(seq) == int(f[1]) and at == int(f[6])
        used = {str(m["node_id"]) for m in mappings}
        cases.append({"source": "test/surject/opposite_strands.gaf", "name": f[0], "graph": {str(k): v for k, v in graph.items() if str(k) in {i for _, i in steps}},
                      "sequence": seq, "mappings": mappings, "mapq": int(f[11]), "steps_in_record": [o + i for o, i in steps], "steps_the_read_touches": sorted(used),
                      "expect_line": l.rstrip("\n")})
    json.dump({"cases": cases}, open(OUT, "w"), indent=1)
    print(OUT, len(cases), "cases")


if __name__ == "__main__":
    main()
