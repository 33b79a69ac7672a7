#!/usr/bin/env python3
"""Transcribe the align_sequence_between / with_dagified_local_graph known-answer tests of the reference into JSON.

Run in the dev container only (needs /root/reference):

    python tests/golden/extract_minimizer_mapper_tests.py      ->  tests/golden/ref_minimizer_mapper.json

Source: /root/reference/src/unittest/minimizer_mapper.cpp:254-880 — the TEST_CASEs that drive MinimizerMapper's static members
align_sequence_between, align_sequence_between_consistently, longest_detectable_gap_in_range and with_dagified_local_graph on small
literal graphs (HashGraph statements or JSON).  The tests are literal statements; this script evaluates exactly those statements
(graph construction, the read, the two anchor positions, the call's numeric arguments) and keeps every REQUIRE as an expression over
the result.  It never executes reference code.

Every TEST_CASE / SECTION becomes
    {"source": "src/unittest/minimizer_mapper.cpp:LINE", "name": "...",
     "graph": {"nodes": [[id, seq]...], "edges": [[from, from_start, to, to_end]...]},
     "call": "align_sequence_between" | "align_sequence_between_consistently" | "with_dagified_local_graph" | "longest_detectable_gap_in_range",
     "sequence": str | [str...] (the consistency test runs a list of reads),
     "left": [id, is_rev, offset] | null, "right": ... | null, "max_path_length": n | "len+gap", "max_gap_length": n | "gap_in_range",
     "requires": [python expression over `aln` (the alignment as the shim's JSON), ...]}
The connect_consistently SECTION (WFAExtender through the same anchors, :793-821) is transcribed as call "connect_consistently".
"""
import json
import os
import re

SRC = "/root/reference/src/unittest/minimizer_mapper.cpp"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_minimizer_mapper.json")
FIRST, LAST = 254, 880


def block_end(lines, start):
    depth = 0; seen = False
    for i in range(start, len(lines)):
        s = re.sub(r'R"\(.*?\)"', '""', lines[i])
        s = re.sub(r'"(?:[^"\\]|\\.)*"', '""', s)
        depth += s.count("{") - s.count("}")
        seen = seen or "{" in s
        if seen and depth == 0:
            return i
    raise ValueError("unbalanced block at line %d" % (start + 1))


def raw_json(text):
    m = re.search(r'R"\((.*?)\)"', text, re.S)
    return json.loads(m.group(1)) if m else None


def graph_of(body):
    """-> (graph dict, {handle variable: node id})"""
    js = raw_json(body)
    if js is not None:
        nodes = [[int(n["id"]), n["sequence"]] for n in js.get("node", [])]
        edges = [[int(e["from"]), bool(e.get("from_start", False)), int(e["to"]), bool(e.get("to_end", False))] for e in js.get("edge", [])]
        return {"nodes": nodes, "edges": edges}, {}
    nodes, edges, var = [], [], {}
    for m in re.finditer(r'auto (\w+) = graph\.create_handle\("([ACGTN]*)"\);|graph\.create_edge\(([^;]*)\);', body):
        if m.group(1):
            var[m.group(1)] = len(nodes) + 1                                  # HashGraph::create_handle hands out 1, 2, 3, ...
            nodes.append([len(nodes) + 1, m.group(2)])
        else:
            ends = []
            for arg in split_args(m.group(3)):
                f = re.fullmatch(r"graph\.flip\((\w+)\)", arg)
                ends.append((var[f.group(1)], True) if f else (var[arg], False))
            (a, ar), (b, br) = ends
            edges.append([a, ar, b, br])                                      # a handle's reverse leaves through the node's start / arrives at its end
    return {"nodes": nodes, "edges": edges}, var


def split_args(text):
    out, depth, cur = [], 0, ""
    for ch in text:
        if ch in "({": depth += 1
        if ch in ")}": depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip()); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def position(text, name, var):
    """the value a `pos_t name ...;` statement gives -> [id, is_rev, offset] or None (empty)"""
    m = re.search(r"pos_t %s\s*(?:=\s*)?(\{[^;]*\}|empty_pos_t\(\))?;" % name, text)
    if not m:
        return None                                                          # (the caller passes empty_pos_t() in place)
    if m.group(1) is None or m.group(1).startswith("empty_pos_t"):
        return None
    a = split_args(m.group(1)[1:-1])
    g = re.fullmatch(r"graph\.get_id\((\w+)\)", a[0])
    return [var[g.group(1)] if g else int(a[0]), a[1] == "true", int(a[2])]


def to_python(req, var, anchors):
    """one REQUIRE's C++ expression -> a python expression over `aln`"""
    e = req
    e = re.sub(r"graph\.get_id\((\w+)\)", lambda m: str(var[m.group(1)]), e)
    e = re.sub(r"!graph\.get_is_reverse\(\w+\)", "True", e)                  # the tests' handles are all forward
    e = re.sub(r"graph\.get_is_reverse\(\w+\)", "False", e)
    e = re.sub(r"offset\((\w+)\)", lambda m: str(anchors[m.group(1)][2]), e)
    e = e.replace("aln.path().mapping_size()", 'len(aln["path"]["mapping"])')
    e = re.sub(r"aln\.path\(\)\.mapping\((\d+)\)", r'aln["path"]["mapping"][\1]', e)
    e = e.replace(".position().node_id()", '["position"]["node_id"]').replace(".position().is_reverse()", '["position"]["is_reverse"]')
    e = e.replace(".position().offset()", '["position"]["offset"]')
    e = e.replace(".edit_size()", '["n_edits"]')
    e = re.sub(r"\.edit\((\d+)\)", r'["edit"][\1]', e)
    e = e.replace(".from_length()", '["from_length"]').replace(".to_length()", '["to_length"]')
    e = e.replace('.sequence().empty()', '["sequence"] == ""').replace(".sequence()", '["sequence"]')
    e = e.replace("aln.score()", 'aln["score"]')
    e = e.replace("true", "True").replace("false", "False").replace("std::max", "max")
    return e


def main():
    text = open(SRC).read().split("\n")
    cases = []
    i = FIRST - 1
    while i < LAST:
        m = re.match(r'TEST_CASE\("([^"]+)"', text[i])
        if not m:
            i += 1
            continue
        end = block_end(text, i)
        name = m.group(1); body_lines = text[i:end + 1]; body = "\n".join(body_lines)
        src = "src/unittest/minimizer_mapper.cpp:%d" % (i + 1)
        graph, var = graph_of(body)

        if "longest_detectable_gap_in_range(aln, aln.sequence().begin(), aln.sequence().end(), &aligner);\n" in body and "align_sequence_between" not in body:
            seq = re.search(r'aln\.set_sequence\("([ACGTN]*)"\)', body).group(1)
            ranges = {}
            for g in re.finditer(r"size_t (\w+) = TestMinimizerMapper::longest_detectable_gap_in_range\(aln,\s*([^,]+),\s*([^,]+),\s*&aligner\);", body):
                def idx(s):
                    s = s.strip().replace("aln.sequence().begin()", "0").replace("aln.sequence().end()", str(len(seq)))
                    return int(eval(s))
                ranges[g.group(1)] = [idx(g.group(2)), idx(g.group(3))]
            reqs = [r.strip() for r in re.findall(r"REQUIRE\((.*)\);", body)]
            cases.append({"source": src, "name": name, "call": "longest_detectable_gap_in_range", "sequence": seq, "ranges": ranges, "requires": reqs})
            i = end + 1
            continue

        if "with_dagified_local_graph(" in body:
            c = re.search(r"with_dagified_local_graph\(make_pos_t\((\d+), (true|false), (\d+)\), empty_pos_t\(\), (\d+), graph", body)
            cases.append({"source": src, "name": name, "call": "with_dagified_local_graph", "graph": graph,
                          "left": [int(c.group(1)), c.group(2) == "true", int(c.group(3))], "right": None, "max_path_length": int(c.group(4)),
                          # :865-878: every head tip is the anchor's node read forwards; two tips in all; the left anchor handle is a tip, 4 bases long
                          "requires": ["all(base == (60245283, False) for base in head_tip_bases)", "n_tips == 2", "left_anchor_is_tip", "left_anchor_length == 4"]})
            i = end + 1
            continue

        # sections share the TEST_CASE's preamble
        sections = []
        j = 0
        pre = []
        while j < len(body_lines):
            s = re.match(r'\s*SECTION\("([^"]+)"\)', body_lines[j])
            if s:
                e2 = block_end(body_lines, j)
                sections.append((s.group(1), i + j + 1, "\n".join(pre + body_lines[j:e2 + 1])))
                j = e2 + 1
            else:
                pre.append(body_lines[j]); j += 1
        if not sections:
            sections = [(None, i + 1, body)]
        for sname, line, sbody in sections:
            full = name + (" / " + sname if sname else "")
            ssrc = "src/unittest/minimizer_mapper.cpp:%d" % line
            if "test_seqs" in sbody:
                seqs = re.findall(r'"([ACGT]+)"', re.search(r"test_seqs \{([^}]*)\}", sbody).group(1))
                anchors = {n: position(sbody, n, var) for n in ("left_anchor", "right_anchor", "rev_left_anchor", "rev_right_anchor")}
                threads = [[[var[v], False] for v in re.findall(r"gbwt::Node::encode\(graph\.get_id\((\w+)\), false\)", sbody)]]
                cases.append({"source": ssrc, "name": full, "graph": graph, "sequence": seqs, "threads": threads,
                              "call": "connect_consistently" if "connect_consistently(" in sbody.split("SECTION")[-1] else "align_sequence_between_consistently",
                              "left": anchors["left_anchor"], "right": anchors["right_anchor"], "rev_left": anchors["rev_left_anchor"], "rev_right": anchors["rev_right_anchor"],
                              "max_path_length": 100, "max_gap_length": 20,
                              "requires": ["flipped reverse-strand alignment == forward alignment (require_alignments_equal, :712-728)"]})
                continue
            seq = re.search(r'aln\.set_sequence\("([ACGTN]*)"\)', sbody).group(1)
            anchors = {"left_anchor": position(sbody, "left_anchor", var), "right_anchor": position(sbody, "right_anchor", var)}
            c = re.search(r"align_sequence_between\(left_anchor, (?:right_anchor|empty_pos_t\(\)), ([^,]+), ([^,]+), &graph, &aligner, aln\)", sbody)
            if "align_sequence_between(left_anchor, empty_pos_t()" in sbody:
                anchors["right_anchor"] = None
            mp, mg = c.group(1).strip(), c.group(2).strip()
            reqs = []
            for r in re.findall(r"REQUIRE\((.*)\);", sbody):
                r = r.strip()
                if r.startswith("last_"):                                     # :705-708, over the last mapping's last edit
                    r = r.replace("last_mapping.edit_size()", 'aln["path"]["mapping"][-1]["n_edits"]').replace("last_edit.to_length()", 'aln["path"]["mapping"][-1]["edit"][-1]["to_length"]')
                    r = r.replace("last_edit.from_length()", 'aln["path"]["mapping"][-1]["edit"][-1]["from_length"]').replace("std::max", "max")
                    reqs.append(r)
                else:
                    reqs.append(to_python(r, var, {k: (v or [0, False, 0]) for k, v in anchors.items()}))
            cases.append({"source": ssrc, "name": full, "graph": graph, "call": "align_sequence_between", "sequence": seq,
                          "left": anchors["left_anchor"], "right": anchors["right_anchor"],
                          "max_path_length": int(mp) if mp.isdigit() else "len+gap", "max_gap_length": int(mg) if mg.isdigit() else "gap_in_range",
                          "requires": reqs})
        i = end + 1
    with open(OUT, "w") as f:
        json.dump({"source": "src/unittest/minimizer_mapper.cpp:%d-%d" % (FIRST, LAST), "cases": cases}, f, indent=1)
        f.write("\n")
    print("%d cases -> %s" % (len(cases), OUT))
    for c in cases:
        print("  %-40s %-38s %d requires" % (c["source"], c["call"], len(c["requires"])))


if __name__ == "__main__":
    main()
