#!/usr/bin/env python3
"""Transcribe the reference's known-answer tests for path simplification into JSON.

Run in the dev container only (needs /root/reference):

    python tests/golden/extract_simplify_tests.py      ->  tests/golden/ref_simplify.json

Sources: /root/reference/src/unittest/path.cpp:21-45 ("Path simplification tolerates adjacent insertions and deletions": simplify(path, false)),
/root/reference/src/unittest/alignment.cpp:57-87 ("Non-trim alignment simplification does not remove deletions on the edges of Mappings":
simplify(a, false)) and :89-102 ("Alignment simplification handles unaligned alignments": simplify(a) — no deletion anywhere, so
trim_internal_deletions makes no difference).  Each test is a JSON literal (the Path / Alignment) and REQUIRE lines on the result; this script
copies the literal's mappings and the REQUIREd numbers and never executes reference code.

Every case becomes {"source", "name", "mappings": [{"node_id": id|0, "offset", "is_reverse", "edits": [[from_length, to_length, has_sequence]...]}...],
"expect": {"mapping_size": n, "node_ids": [...]|null, "edit_sizes": {mapping index: n}}}.
"""
import json
import os
import re

REF = "/root/reference/src/unittest"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_simplify.json")


def test_case(path, title):
    text = open(path).read()
    at = text.index('TEST_CASE("%s"' % title)
    line = text.count("\n", 0, at) + 1
    end = text.index("\nTEST_CASE(", at + 1) if "\nTEST_CASE(" in text[at + 1:] else len(text)
    body = text[at:end]
    lit = re.search(r'R"\((.*?)\)"', body, re.S).group(1)
    doc = json.loads(lit)
    path_doc = doc["path"] if "path" in doc else doc
    mappings = []
    for m in path_doc["mapping"]:
        pos = m.get("position", {})
        mappings.append({"node_id": int(pos.get("node_id", 0)), "offset": int(pos.get("offset", 0)), "is_reverse": bool(pos.get("is_reverse", False)),
                         "edits": [[int(e.get("from_length", 0)), int(e.get("to_length", 0)), bool(e.get("sequence"))] for e in m["edit"]]})
    expect = {"mapping_size": None, "node_ids": None, "edit_sizes": {}}
    for req in re.findall(r"REQUIRE\((.*?)\);", body):
        m = re.match(r"simple(?:\.path\(\))?\.mapping_size\(\) == (\d+)", req)
        if m:
            expect["mapping_size"] = int(m.group(1)); continue
        m = re.match(r"simple\.mapping\((\d+)\)\.position\(\)\.node_id\(\) == (\d+)", req)
        if m:
            expect["node_ids"] = (expect["node_ids"] or []) + [int(m.group(2))]; continue
        m = re.match(r"simple\.path\(\)\.mapping\((\d+)\)\.edit_size\(\) == (\d+)", req)
        if m:
            expect["edit_sizes"][m.group(1)] = int(m.group(2))
    return {"source": "src/unittest/%s:%d" % (os.path.basename(path), line), "name": title, "mappings": mappings, "expect": expect}


cases = [test_case(REF + "/path.cpp", "Path simplification tolerates adjacent insertions and deletions"),
         test_case(REF + "/alignment.cpp", "Non-trim alignment simplification does not remove deletions on the edges of Mappings"),
         test_case(REF + "/alignment.cpp", "Alignment simplification handles unaligned alignments")]
with open(OUT, "w") as f:
    json.dump({"cases": cases}, f, indent=1)
print(OUT, [c["expect"] for c in cases])
