#!/usr/bin/env python3
"""Transcribe the reference's unit tests of algorithms::sample_minimal into JSON.

Run in the dev container only (needs /root/reference):

    python tests/golden/extract_sample_minimal_tests.py      ->  tests/golden/ref_sample_minimal.json

Source: /root/reference/src/unittest/sample_minimal.cpp:14-176 — six TEST_CASEs, each a literal call: sequence length, element length,
window size, the elements' starts (a vector, or "element i starts at i"), a should_beat lambda (nobody beats anybody; element 0 / element 1
beats every other; by a vector of goodness) and REQUIREs on the set of sampled elements.  This script reads those literals and keeps every
REQUIRE; it never executes reference code.  should_beat is stored as a goodness per element (a beats b iff goodness[a] > goodness[b]),
which expresses all four lambdas.

Every TEST_CASE becomes {"source": "src/unittest/sample_minimal.cpp:LINE", "name", "sequence_length", "element_length", "window_size",
"starts": [...], "goodness": [...], "sampled_count": n, "sampled_contains": [i, ...]}."""
import json
import os
import re

SRC = "/root/reference/src/unittest/sample_minimal.cpp"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_sample_minimal.json")


def main():
    text = open(SRC).read()
    lines = text.split("\n")
    heads = [(i, re.match(r'TEST_CASE\("([^"]*)"', l).group(1)) for i, l in enumerate(lines) if l.startswith("TEST_CASE(")]
    cases = []
    for k, (at, name) in enumerate(heads):
        end = heads[k + 1][0] if k + 1 < len(heads) else len(lines)
        body = "\n".join(lines[at:end])
        num = lambda what: int(re.search(r"size_t %s = (\d+);" % what, body).group(1))
        seq_len, elem_len, window = num("sequence_length"), num("element_length"), num("window_size")
        m = re.search(r"element_starts \{([^}]*)\}", body)
        if m:
            starts = [int(x) for x in m.group(1).split(",")]
        else:                                                   # "size_t element_count = sequence_length - element_length + 1;" and "return i;"
            assert "element_count = sequence_length - element_length + 1" in body and re.search(r"return i;", body)
            starts = list(range(seq_len - elem_len + 1))
        n = len(starts)
        g = re.search(r"element_goodness \{([^}]*)\}", body)
        if g:
            goodness = [int(x) for x in g.group(1).split(",")]
            assert "element_goodness.at(a) > element_goodness.at(b)" in body
        elif "return false;" in body:
            goodness = [0] * n
        else:
            w = re.search(r"return a == (\d+) && b != \1;", body)
            goodness = [1 if i == int(w.group(1)) else 0 for i in range(n)]
        count = re.search(r"REQUIRE\(sampled_elements\.size\(\) == ([a-z_0-9]+)\);", body).group(1)
        count = int(count) if count.isdigit() else {"window_count": seq_len - window + 1}[count]
        contains = [int(x) for x in re.findall(r"REQUIRE\(sampled_elements\.count\((\d+)\)\);", body)]
        cases.append({"source": "src/unittest/sample_minimal.cpp:%d" % (at + 1), "name": name, "sequence_length": seq_len, "element_length": elem_len,
                      "window_size": window, "starts": starts, "goodness": goodness, "sampled_count": count, "sampled_contains": contains})
    json.dump(cases, open(OUT, "w"), indent=1)
    print("%d cases -> %s" % (len(cases), OUT))


if __name__ == "__main__":
    main()
