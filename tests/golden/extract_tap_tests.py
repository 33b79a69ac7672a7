#!/usr/bin/env python3
"""Transcribe the known-answer cases of the reference's TAP test test/t/04_vg_align.t whose inputs are self-contained files
(.vg protobuf graphs + a read) into tests/golden/ref_tap_align.json, and test/tiny/tiny.gfa (the configs[0] graph) into
tests/golden/tiny_graph.json.  Run in the build container (needs /root/reference); the fixtures travel, the reference does not.

A .vg file is a gzip'd stream of protobuf messages (libvgio): varint group count, then per message a varint length and a
vg.Graph {repeated Node node = 1 {sequence = 1, name = 2, id = 3}; repeated Edge edge = 2 {from = 1, to = 2, from_start = 3,
to_end = 4}} — decoded here by hand (no protobuf module needed for five fields)."""
import gzip
import json
import os
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def varint(buf, at):
    v = shift = 0
    while True:
        b = buf[at]; at += 1
        v |= (b & 0x7f) << shift; shift += 7
        if not b & 0x80:
            return v, at


def fields(buf):
    at = 0
    while at < len(buf):
        key, at = varint(buf, at)
        num, wt = key >> 3, key & 7
        if wt == 0:
            val, at = varint(buf, at)
        elif wt == 2:
            ln, at = varint(buf, at); val = buf[at:at + ln]; at += ln
        elif wt == 1:
            val = buf[at:at + 8]; at += 8
        elif wt == 5:
            val = buf[at:at + 4]; at += 4
        else:
            raise ValueError("wire type %d" % wt)
        yield num, wt, val


def read_vg(path):
    raw = open(path, "rb").read()
    data = gzip.decompress(raw) if raw[:2] == b"\x1f\x8b" else raw
    nodes, edges = [], []
    at = 0
    while at < len(data):
        count, at = varint(data, at)
        for _ in range(count):
            ln, at = varint(data, at)
            msg = data[at:at + ln]; at += ln
            for num, wt, val in fields(msg):
                if num == 1 and wt == 2:
                    seq, nid = "", None
                    for n2, w2, v2 in fields(val):
                        if n2 == 1: seq = v2.decode()
                        elif n2 == 3: nid = v2
                    nodes.append([nid, seq])
                elif num == 2 and wt == 2:
                    e = {1: None, 2: None, 3: 0, 4: 0}
                    for n2, w2, v2 in fields(val):
                        if n2 in e: e[n2] = v2
                    assert not e[3] and not e[4], "reversing edge in %s" % path
                    edges.append([e[1], e[2]])
    return nodes, edges


def tap_cases():
    t = os.path.join(REF, "test")
    seq = lambda f: open(os.path.join(t, f)).read().strip()
    long_read = ("GGCTATGTCTGAACTAGGAGGGTAGAAAGAATATTCATTTTGGTTGCCACAAACCATCGAAACAAAGATGCAGGTCATTGATGTAAAACTACAGTTAGTTCCTACTGACTCCTTTTCAGCTTC"
                 "TCTTCATTGCTATGAGCCAGCGTCTCCT")
    spec = [
        ("test/t/04_vg_align.t:26", "alignment does not contain excessive soft clips under lenient scoring", "mapsoftclip/70211809-70211845.vg",
         seq("mapsoftclip/70211809-70211845.seq"), [2, 2, 3, 1, 0], [["node_id", 0, 70211814]]),
        ("test/t/04_vg_align.t:30", "alignment score does not overflow at 255 when using 8x16bit vectors", "mapsoftclip/113968116:113968146.vg",
         seq("mapsoftclip/113968116:113968146.seq"), [2, 2, 3, 1, 0], [["score", 274]]),
        ("test/t/04_vg_align.t:34", "Ns do not cause excessive soft clipping", "mapsoftclip/280136066-280136088.vg",
         seq("mapsoftclip/280136066-280136088.seq"), [1, 4, 6, 1, 5], [["node_id", 0, 280136076]]),
        ("test/t/04_vg_align.t:36", "nodes are only referenced if they have mappings", "graphs/59867692-59867698.vg", long_read, [1, 4, 6, 1, 5],
         [["node_id", 0, 59867694]]),
    ]
    cases = []
    for source, name, vg, read, scores, expect in spec:
        nodes, edges = read_vg(os.path.join(t, vg))
        cases.append({"source": source, "name": name, "nodes": nodes, "edges": edges, "read": read, "quality": None, "scores": scores,
                      "qual_adj": False, "call": "align", "args": ["graph", True], "aln": "aln", "expect": expect, "input": "test/" + vg})
    return cases


def tiny_graph():
    nodes, edges, path = {}, [], None
    for line in open(os.path.join(REF, "test", "tiny", "tiny.gfa")):
        f = line.rstrip("\n").split("\t")
        if f[0] == "S":
            nodes[int(f[1])] = f[2]
        elif f[0] == "L":
            assert f[2] == "+" and f[4] == "+"
            edges.append([int(f[1]), int(f[3])])
        elif f[0] == "P":
            path = [int(x[:-1]) for x in f[2].split(",")]
    return {"source": "test/tiny/tiny.gfa", "nodes": [[k, nodes[k]] for k in sorted(nodes)], "edges": sorted(edges), "reference_path": path}


if __name__ == "__main__":
    with open(os.path.join(OUT, "ref_tap_align.json"), "w") as f:
        json.dump(tap_cases(), f, indent=1)
    with open(os.path.join(OUT, "tiny_graph.json"), "w") as f:
        json.dump(tiny_graph(), f, indent=1)
    print("wrote ref_tap_align.json (%d cases), tiny_graph.json" % len(tap_cases()))
