#!/usr/bin/env python3
"""Transcribe the reference's known-answer tests of the graph algorithms behind N1 into JSON.

Run in the dev container only (needs /root/reference):

    python tests/golden/extract_graph_algorithm_tests.py      ->  tests/golden/ref_graph_algorithms.json

Sources (literal statements only are evaluated; no reference code is executed):
  src/unittest/vg_algorithms.cpp:46-1160    extract_connecting_graph  (acyclic, cyclic, reversing, self-loop, doubling-back, pruning cases)
  src/unittest/vg_algorithms.cpp:1720-1933  extract_containing_graph
  src/unittest/vg_algorithms.cpp:1934-2495  extract_extending_graph
  src/unittest/dagify.cpp:20-517            handlealgs::dagify (node counts 6 / 8 / 6, preserved walks) and dagify_from (tips)

Every TEST_CASE / SECTION that calls one of the algorithms becomes
    {"source": "file:LINE", "name": "...", "graph": {"nodes": [[id, seq]...], "edges": [[from, from_start, to, to_end]...]},
     "call": "extract_connecting" | "extract_containing" | "extract_extending" | "dagify" | "dagify_from", "args": [...] (vgh_graph_algorithm's),
     "facts": {...}}
with the REQUIREs the patterns below recognise:
     n_nodes / n_edges                         node_size() / get_node_count() / edge_size() == N on the extracted graph
     empty                                     node_size() == 0 (the "no path under the maximum length" cases)
     sources_within                            set<int64_t> expected_node_ids{...}: every extracted node stands for one of these
     retained                                  retained_node_ids.count(id): these source nodes are present
     no_duplicates                             trans.size() == retained_node_ids.size()
     n_retained / sequences_include            retained_node_ids.size() == N (or .empty()); node_sequences.count("...")
     sequence_of / other_sequence              if (trans[n.id()] == nX->id()) REQUIRE(n.sequence() == "..."), and the final else branch
     acyclic, walks (dagify)                   is_acyclic, and the walks "preserved in the new DAG"
What a section checks beyond these (edge-by-edge orientation flags, mostly) is not transcribed; the section's line is kept so it can be read.
"""
import json
import os
import re

ROOT = "/root/reference/src/unittest/"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_graph_algorithms.json")


def block_end(lines, start):
    depth = 0; seen = False
    for i in range(start, len(lines)):
        s = re.sub(r'"(?:[^"\\]|\\.)*"', '""', lines[i])
        s = re.sub(r"//.*", "", s)
        depth += s.count("{") - s.count("}")
        seen = seen or "{" in s
        if seen and depth == 0:
            return i
    raise ValueError("unbalanced block at line %d" % (start + 1))


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


def graph_of(body):
    """VG (create_node / create_edge(a, b, from_start, to_end)) or HashGraph (create_handle / create_edge(h, flip(h))) statements -> graph, {var: id}"""
    js = re.search(r'R"\((.*?)\)"', body, re.S)
    if js:
        g = json.loads(js.group(1))
        return {"nodes": [[int(n["id"]), n["sequence"]] for n in g.get("node", [])],
                "edges": [[int(e["from"]), bool(e.get("from_start", False)), int(e["to"]), bool(e.get("to_end", False))] for e in g.get("edge", [])]}, {}
    nodes, edges, var = [], [], {}
    for m in re.finditer(r"(?:Node\*|handle_t|auto)\s+(\w+)\s*=\s*\w+(?:\.|->)create_(?:node|handle)\(\"([ACGTN]*)\"\);|\w+(?:\.|->)create_edge\(([^;]*)\);", body):
        if m.group(1):
            var[m.group(1)] = len(nodes) + 1
            nodes.append([len(nodes) + 1, m.group(2)])
            continue
        a = split_args(m.group(3))
        def end(x):
            f = re.fullmatch(r"\w+(?:\.|->)flip\((\w+)\)", x)
            return (var[f.group(1)], True) if f else (var[x], False)
        if len(a) == 2:
            (u, ur), (v, vr) = end(a[0]), end(a[1])
            edges.append([u, ur, v, vr])
        else:                                                        # VG::create_edge(from, to, from_start, to_end)
            edges.append([var[a[0]], a[2] == "true", var[a[1]], a[3] == "true"])
    return {"nodes": nodes, "edges": edges}, var


def node_id(expr, var):
    m = re.fullmatch(r"(\w+)->id\(\)|\w+\.get_id\((\w+)\)|(\d+)", expr.strip())
    if not m:
        raise ValueError("node id expression: " + expr)
    return int(m.group(3)) if m.group(3) else var[m.group(1) or m.group(2)]


def position(body, name, var):
    inline = re.fullmatch(r"make_pos_t\(([^;]*)\)", name.strip())
    if inline:
        a = split_args(inline.group(1))
        return [node_id(a[0], var), 1 if a[1] == "true" else 0, int(a[2])]
    m = None
    for m in re.finditer(r"pos_t %s\s*=\s*make_pos_t\(([^;]*)\);" % name, body):
        pass                                                         # the section's own (last) definition wins
    if not m:
        return None
    a = split_args(m.group(1))
    return [node_id(a[0], var), 1 if a[1] == "true" else 0, int(a[2])]


def int_var(body, name):
    m = None
    for m in re.finditer(r"(?:int64_t|size_t|int|auto)\s+%s\s*=\s*(\d+);" % name, body):
        pass
    return int(m.group(1)) if m else None


def bool_var(body, name):
    if name in ("true", "false"):
        return 1 if name == "true" else 0
    m = None
    for m in re.finditer(r"bool\s+%s\s*=\s*(true|false);" % name, body):
        pass
    return 1 if m.group(1) == "true" else 0


def facts_of(body, var):
    f = {}
    m = re.search(r"(?:g\.node_size\(\)|extractor\.get_node_count\(\)|dagified\.get_node_count\(\))\s*==\s*(\d+)", body)
    if m:
        f["n_nodes"] = int(m.group(1))
    m = re.search(r"g\.edge_size\(\)\s*==\s*(\d+)", body)
    if m:
        f["n_edges"] = int(m.group(1))
    m = re.search(r"set<int64_t> expected_node_ids\s*\{([^}]*)\}", body)
    if m:
        f["sources_within"] = sorted(node_id(x, var) for x in split_args(m.group(1)))
    kept = [node_id(x, var) for x in re.findall(r"REQUIRE\(\s*retained_node_ids\.count\(([^;]*?)\)\s*\);", body)]
    if kept:
        f["retained"] = sorted(set(kept))
    if re.search(r"trans\.size\(\)\s*==\s*retained_node_ids\.size\(\)", body):
        f["no_duplicates"] = True
    seqs = {}
    for m in re.finditer(r"trans\[n\.id\(\)\]\s*==\s*(\w+)->id\(\)\s*\)\s*\{\s*REQUIRE\(\s*n\.sequence\(\)\s*==\s*\"([ACGTN]*)\"", body):
        seqs.setdefault(str(var[m.group(1)]), set()).add(m.group(2))
    if seqs:
        f["sequence_of"] = {k: sorted(v) for k, v in seqs.items()}           # (a node cut twice may appear with two sequences)
    m = re.search(r"else\s*\{\s*REQUIRE\(\s*n\.sequence\(\)\s*==\s*\"([ACGTN]*)\"\s*\);\s*\}", body)
    if m:
        f["other_sequence"] = m.group(1)
    m = re.search(r"g\.node\(0\)\.sequence\(\)\s*==\s*\"([ACGTN]*)\"", body)
    if m:
        f["only_sequence"] = m.group(1)
    m = re.search(r"REQUIRE\(retained_node_ids\.size\(\) == (\d+)\);", body)
    if m:
        f["n_retained"] = int(m.group(1))
    if re.search(r"REQUIRE\(retained_node_ids\.empty\(\)\);", body):
        f["n_retained"] = 0
    inc = re.findall(r"REQUIRE\(node_sequences\.count\(\"([ACGTN]*)\"\)\);", body)
    if inc:
        f["sequences_include"] = inc
    if re.search(r"REQUIRE\(\s*handlealgs::is_acyclic\(&dagified\)\s*\)", body):
        f["acyclic"] = True
    return f


def sections_of(lines, first, last):
    """-> [(name, line number, body with the TEST_CASE's shared preamble)]"""
    out = []
    i = first - 1
    while i < last:
        m = re.match(r'\s*TEST_CASE\(\s*"([^"]+)"', lines[i])
        if not m:
            i += 1
            continue
        end = block_end(lines, i)
        body = lines[i:end + 1]
        pre, j, found = [], 0, []
        while j < len(body):
            s = re.match(r'\s*SECTION\(\s*"([^"]+)"', body[j])
            if s and j > 0:
                e2 = block_end(body, j)
                found.append((m.group(1) + " / " + s.group(1), i + j + 1, "\n".join(pre + body[j:e2 + 1])))
                j = e2 + 1
            else:
                pre.append(body[j]); j += 1
        out.extend(found if found else [(m.group(1), i + 1, "\n".join(body))])
        i = end + 1
    return out


def main():
    cases = []
    lines = open(ROOT + "vg_algorithms.cpp").read().split("\n")
    for name, line, body in sections_of(lines, 46, 2495):
        src = "src/unittest/vg_algorithms.cpp:%d" % line
        graph, var = graph_of(body)
        c = re.search(r"extract_connecting_graph\(&\w+, &\w+,\s*([^;]*)\);", body)
        if c:
            a = split_args(c.group(1))
            max_len = int(a[0]) if a[0].isdigit() else int_var(body, a[0])
            p1, p2 = position(body, a[1], var), position(body, a[2], var)
            strict = 1 if len(a) > 3 and a[3] == "true" else 0
            cases.append({"source": src, "name": name, "graph": graph, "call": "extract_connecting", "args": [max_len] + p1 + p2 + [strict], "facts": facts_of(body, var)})
            continue
        c = re.search(r"extract_extending_graph\(&\w+, &\w+,\s*([^;]*)\);", body)
        if c:
            a = split_args(c.group(1))
            max_len = int(a[0]) if a[0].isdigit() else int_var(body, a[0])
            cases.append({"source": src, "name": name, "graph": graph, "call": "extract_extending",
                          "args": [max_len] + position(body, a[1], var) + [bool_var(body, a[2]), bool_var(body, a[3])], "facts": facts_of(body, var)})
            continue
        c = re.search(r"extract_containing_graph\(&\w+, &\w+,\s*([^;]*)\);", body)
        if c:
            a = split_args(c.group(1))
            listed = None
            for listed in re.finditer(r"vector<pos_t> positions\s*\{([^;]*)\};", body):
                pass                                                 # the section's own list
            pos = [[node_id(x[0], var), 1 if x[1] == "true" else 0, int(x[2])] for x in
                   (split_args(re.fullmatch(r"make_pos_t\((.*)\)", p).group(1)) for p in split_args(listed.group(1)))]
            def lens(v):
                if v.isdigit():
                    return [int(v)] * len(pos)
                one = int_var(body, v)
                if one is not None:
                    return [one] * len(pos)
                return [int(x) for x in split_args(re.search(r"vector<size_t> %s\s*\{([^}]*)\}" % v, body).group(1))]
            fw = lens(a[1]); bw = lens(a[2]) if len(a) > 2 else fw
            args = [0, len(pos)]
            for p, f_, b_ in zip(pos, fw, bw):
                args += p + [f_, b_]
            f = facts_of(body, var)
            found = [node_id(x, var) for x in re.findall(r"bool found_node_\d+ = false;|if \(n\.id\(\) == (n\d+->id\(\))", body) if x]
            if found:
                f["node_ids"] = sorted(set(found))                   # the nodes the section looks for one by one: exactly these
            cases.append({"source": src, "name": name, "graph": graph, "call": "extract_containing", "args": args, "facts": f})

    lines = open(ROOT + "dagify.cpp").read().split("\n")
    for name, line, body in sections_of(lines, 20, 517):
        src = "src/unittest/dagify.cpp:%d" % line
        graph, var = graph_of(body)
        if "random_graph(" in body:
            continue                                                 # (1000 random graphs: tests/test_graph_algorithms.py makes its own)
        f = facts_of(body, var)
        c = re.search(r"dagify_from\(&graph, \{([^}]*)\}, &dagified, (\d+)\)", body)
        if c:
            s = c.group(1).strip()
            fl = re.fullmatch(r"graph\.flip\((\w+)\)", s)
            start = [var[fl.group(1)], 1] if fl else [var[s], 0]
            # :466-482 / :497-513, in words: one tip on the start's side, it stands for the start node and is the embedded start (flipped when
            # the start was a reverse handle); tips on the other side stand for the named nodes, forward
            if not fl:
                f.update({"heads": {"count": 1, "sources": [var["n1"]]}, "tails": {"sources": [var["n3"], var["n4"]]}, "start_is": "head"})
            else:
                f.update({"tails": {"count": 1, "sources": [var["n3"]]}, "heads": {"sources": [var["n1"]]}, "start_is": "flipped tail"})
            cases.append({"source": src, "name": name, "graph": graph, "call": "dagify_from", "args": [int(c.group(2))] + start, "facts": f})
            continue
        if "handlealgs::dagify(&graph" in body:
            walks, cur = [], None
            for m in re.finditer(r"\bwalks\.emplace_back\(\);|\bwalks\.back\(\)\.push_back\(([^;]*)\);", body):
                if m.group(1) is None:
                    cur = []; walks.append(cur)
                else:
                    fl = re.fullmatch(r"graph\.flip\((\w+)\)", m.group(1).strip())
                    cur.append([var[fl.group(1)], 1] if fl else [var[m.group(1).strip()], 0])
            f["walks"] = walks
            cases.append({"source": src, "name": name, "graph": graph, "call": "dagify", "args": [int_var(body, "preserved_length")], "facts": f})

    with open(OUT, "w") as fh:
        json.dump({"cases": cases}, fh, indent=1)
        fh.write("\n")
    print("%d cases -> %s" % (len(cases), OUT))
    for c in cases:
        print("  %-44s %-20s %s" % (c["source"], c["call"], sorted(c["facts"].keys())))


if __name__ == "__main__":
    main()
