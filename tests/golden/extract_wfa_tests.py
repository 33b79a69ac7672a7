#!/usr/bin/env python3
"""Transcribe the WFAExtender known-answer tests of the reference into JSON.

Run in the dev container only (needs /root/reference):

    python tests/golden/extract_wfa_tests.py      ->  tests/golden/ref_wfa_extender.json

Source: /root/reference/src/unittest/gbwt_extender.cpp, the `[wfa_extender]` TEST_CASEs (:1531-2620) and the
graph/haplotype builders they use (:1230-1390).  The tests are literal statements (create_node, push_back of encoded
nodes, std::string sequence("..."), pos_t from(...), an optional ErrorModel, one extender call and the checks on its
result); this script evaluates exactly those statements and never executes reference code.

Every SECTION becomes
    {"source": "src/unittest/gbwt_extender.cpp:LINE", "name": "case / section", "graph": name,
     "sequence": str, "call": "connect"|"prefix"|"suffix", "from": [id, is_rev, off]|null, "to": [...]|null,
     "error_model": [[per_base, min, max] * 4],
     "expect": {"kind": "score", "matches": m, "mismatches": x, "gaps": g, "gap_length": l, "full_length_ends": f,
                "check_alignment": bool}
             | {"kind": "fail"} | {"kind": "unlocalized_insertion"}}
and "graphs" holds {name: {"nodes": [[id, seq]...], "paths": [[[id, is_rev]...]...]}}.
"""
import json
import os
import re

SRC = "/root/reference/src/unittest/gbwt_extender.cpp"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_wfa_extender.json")

DEFAULT_MODEL = [[0.03, 1, 6], [0.05, 1, 10], [0.1, 1, 20], [0.1, 10, 200]]   # gbwt_extender.hpp:386-395


def strip_comments(text):
    return re.sub(r"//[^\n]*", "", text)


def block_end(lines, start):
    """lines[start] opens a block with '{'; return the index of the line closing it."""
    depth = 0
    for i in range(start, len(lines)):
        s = re.sub(r'"[^"]*"', '""', lines[i])
        depth += s.count("{") - s.count("}")
        if depth == 0 and i >= start and "{" in re.sub(r'"[^"]*"', '""', "".join(lines[start:i + 1])):
            return i
    raise ValueError("unbalanced block at line %d" % start)


def unroll(lines):
    """Expand `for (size_t i = 0; i < N; i++) { body }` blocks."""
    out, i = [], 0
    while i < len(lines):
        m = re.search(r"for \(size_t i = 0; i < (\d+); i\+\+\) \{", lines[i])
        if m:
            end = block_end(lines, i)
            body = unroll(lines[i + 1:end])
            out.extend(body * int(m.group(1)))
            i = end + 1
        else:
            out.append(lines[i])
            i += 1
    return out


def parse_builders(lines):
    graphs = {}
    i = 0
    while i < len(lines):
        m = re.match(r"\s*gbwt::GBWT (wfa_\w+)_gbwt\(\) \{", lines[i])
        g = re.match(r"\s*gbwtgraph::GBWTGraph (wfa_\w+)_graph\(const gbwt::GBWT& index\) \{", lines[i])
        if m:
            end = block_end(lines, i)
            paths = []
            for ln in unroll(lines[i + 1:end]):
                if "paths.emplace_back()" in ln:
                    paths.append([])
                p = re.search(r"push_back\(gbwt::Node::encode\((\d+), (true|false)\)\)", ln)
                if p:
                    paths[-1].append([int(p.group(1)), p.group(2) == "true"])
            graphs.setdefault(m.group(1), {})["paths"] = paths
            i = end
        elif g:
            end = block_end(lines, i)
            nodes = []
            for ln in lines[i + 1:end]:
                p = re.search(r'create_node\((\d+), "([A-Za-z]*)"\)', ln)
                if p:
                    nodes.append([int(p.group(1)), p.group(2)])
            graphs.setdefault(g.group(1), {})["nodes"] = nodes
            i = end
        i += 1
    return graphs


def eval_len_expr(expr, seq):
    expr = expr.strip().replace("sequence.length()", str(len(seq))).replace("sequence.size()", str(len(seq)))
    if not re.fullmatch(r"[\d\s+\-*]+", expr):
        raise ValueError("unexpected expression: " + expr)
    return int(eval(expr))


def parse_section(name, lineno, lines, graph, pre):
    st = {"sequence": None, "from": None, "to": None, "model": [list(x) for x in DEFAULT_MODEL], "ss": ""}
    call, expect, check_aln = None, None, False
    body = unroll(pre + lines)
    i = 0
    while i < len(body):
        ln = body[i]
        m = re.search(r'std::string sequence\("([A-Za-z]*)"\)', ln)
        if m:
            st["sequence"] = m.group(1)
        elif re.search(r"std::string sequence;", ln):
            st["sequence"] = ""
        m = re.search(r'ss << "([A-Za-z]*)"', ln)
        if m:
            st["ss"] += m.group(1)
        if "sequence = ss.str()" in ln:
            st["sequence"] = st["ss"]
        for key in ("from", "to"):
            m = re.search(r"pos_t %s\((\d+), (true|false), (\d+)\)" % key, ln)
            if m:
                st[key] = [int(m.group(1)), m.group(2) == "true", int(m.group(3))]
        if re.search(r"WFAExtender::ErrorModel errors \{", ln):
            rows = []
            j = i + 1
            while len(rows) < 4:
                r = re.search(r"\{([\d.]+), (\d+), (\d+)\}", body[j])
                if r:
                    rows.append([float(r.group(1)), int(r.group(2)), int(r.group(3))])
                elif "default_distance()" in body[j]:
                    rows.append(list(DEFAULT_MODEL[3]))
                j += 1
            st["model"] = rows
            i = j
            continue
        m = re.search(r"model\.(mismatches|gaps|gap_length|distance)\.(per_base|min|max) = ([\d.]+);", ln)
        if m:
            row = ["mismatches", "gaps", "gap_length", "distance"].index(m.group(1))
            col = ["per_base", "min", "max"].index(m.group(2))
            st["model"][row][col] = float(m.group(3)) if col == 0 else int(m.group(3))
        m = re.search(r"extender\.(connect|prefix|suffix)\(sequence, (\w+)(?:, (\w+))?\)", ln)
        if m:
            call = m.group(1)
        m = re.search(r"check_score\(result, aligner, (.*)\);", ln)
        if m:
            args = [a.strip() for a in m.group(1).split(",")]
            vals = [eval_len_expr(a, st["sequence"]) for a in args]
            while len(vals) < 5:
                vals.append(0)
            expect = {"kind": "score", "matches": vals[0], "mismatches": vals[1], "gaps": vals[2],
                      "gap_length": vals[3], "full_length_ends": vals[4]}
        if "check_alignment(result" in ln:
            check_aln = True
        if "REQUIRE_FALSE(result)" in ln or "REQUIRE(!(bool)(result))" in ln:
            expect = {"kind": "fail"}
        if "check_unlocalized_insertion(result" in ln:
            expect = {"kind": "unlocalized_insertion"}
        i += 1
    if call is None or expect is None or st["sequence"] is None:
        raise ValueError("section %r at line %d not understood" % (name, lineno))
    if expect["kind"] == "score":
        expect["check_alignment"] = check_aln
    return {"source": "src/unittest/gbwt_extender.cpp:%d" % lineno, "name": name, "graph": graph,
            "sequence": st["sequence"], "call": call,
            "from": st["from"] if call != "prefix" else None, "to": st["to"] if call != "suffix" else None,
            "error_model": st["model"], "expect": expect}


def main():
    raw = open(SRC).read().split("\n")
    lines = strip_comments("\n".join(raw)).split("\n")
    graphs = parse_builders(lines)
    cases = []
    i = 0
    while i < len(lines):
        m = re.match(r'TEST_CASE\("([^"]*)", "\[wfa_extender\]"\) \{', lines[i])
        if not m:
            i += 1
            continue
        end = block_end(lines, i)
        case_name, graph, pre = m.group(1), None, []
        j = i + 1
        while j < end:
            g = re.search(r"gbwt::GBWT index = (wfa_\w+)_gbwt\(\);", lines[j])
            if g:
                graph = g.group(1)
            s = re.match(r'\s*SECTION\("([^"]*)"\) \{', lines[j])
            if s:
                send = block_end(lines, j)
                cases.append(parse_section(case_name + " / " + s.group(1), j + 1, lines[j + 1:send], graph, pre))
                j = send + 1
                continue
            pre.append(lines[j])
            j += 1
        i = end + 1
    for c in cases:
        assert c["graph"] in graphs and "nodes" in graphs[c["graph"]] and "paths" in graphs[c["graph"]], c["graph"]
    with open(OUT, "w") as f:
        json.dump({"graphs": graphs, "cases": cases}, f, indent=1)
    kinds = {}
    for c in cases:
        kinds[c["expect"]["kind"]] = kinds.get(c["expect"]["kind"], 0) + 1
    print("wrote %s: %d graphs, %d cases %s" % (OUT, len(graphs), len(cases), kinds))


if __name__ == "__main__":
    main()
