#!/usr/bin/env python3
"""Golden vectors from the index files the reference keeps for its own tests in test/primers/ (used by test/t/56_vg_primers.t):

  y.gbwt   the haplotype index (GBWT, simple-sds serialization: header flag 0x4), 6 sequences = 3 haplotypes x 2 orientations
  y.gg     the GBWTGraph of the same graph: the sequences of its 66 nodes
  y.min    gbwtgraph's MinimizerIndex of that graph (k = 31, w = 50): 62 keys with one graph position each
  y.giraffe.gbz   the same GBWT and graph in gbwtgraph's GBZ container (carried as bytes only: what the engine's loader is held to
                  is the decoding of y.gbwt and y.gg below)

They are the only artefacts of gbwt / gbwtgraph (absent submodules) the snapshot holds, so they are what pins
  * the haplotype index (vgk_haplo_create, vgk_haplo_create_gbwt): record contents and search states,
  * the minimizer scheme (vgk_minimizer_index_create): key encoding, hash, canonical orientation, window rule, stored position.
This script only DECODES the three files (formats as published: simple-sds vectors, GBWT's byte-coded records, the hash table
of (key, position, payload) cells) and writes tests/golden/ref_primers_y.json; it computes no minimizer and builds no index.
Run in the build container:  python tests/golden/extract_primers_fixture.py   (reads /root/reference, which the GPU box lacks)."""
import json
import os
import struct
import sys

REF = os.environ.get("VG_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


class Reader:
    def __init__(self, data, at=0):
        self.d, self.at = data, at

    def u32(self):
        v = struct.unpack_from("<I", self.d, self.at)[0]; self.at += 4; return v

    def u64(self):
        v = struct.unpack_from("<Q", self.d, self.at)[0]; self.at += 8; return v

    def raw(self):                       # simple-sds RawVector: bit length, then a vector of words (count, words)
        bits, n = self.u64(), self.u64()
        big = int.from_bytes(self.d[self.at:self.at + 8 * n], "little"); self.at += 8 * n
        return bits, big

    def int_vector(self):                # IntVector: length, width, RawVector
        n, width = self.u64(), self.u64()
        bits, big = self.raw()
        assert bits == n * width
        return [(big >> (i * width)) & ((1 << width) - 1) for i in range(n)], width

    def skip_option(self):               # Option<T>: size in words, then T
        n = self.u64(); self.at += 8 * n

    def sparse_vector(self):             # SparseVector (Elias-Fano): length, high BitVector (ones, RawVector, three optional supports), low IntVector
        universe = self.u64()
        ones = self.u64(); bits, high = self.raw()
        self.skip_option(); self.skip_option(); self.skip_option()
        low, width = self.int_vector()
        assert len(low) == ones
        values, i = [], 0
        for p in range(bits):
            if (high >> p) & 1:
                values.append(((p - i) << width) | low[i]); i += 1
        return universe, values

    def byte_vector(self):               # Vec<u8>: length, bytes padded to a word
        n = self.u64(); b = self.d[self.at:self.at + n]; self.at += (n + 7) // 8 * 8; return b


def byte_code(b, i):                     # GBWT ByteCode: 7 bits per byte, low bits first, high bit = "continues"
    v = sh = 0
    while True:
        c = b[i]; i += 1; v |= (c & 0x7f) << sh; sh += 7
        if not c & 0x80:
            return v, i


def decode_gbwt(data):
    r = Reader(data)
    tag, version = r.u32(), r.u32()
    sequences, size, offset, alphabet_size, flags = (r.u64() for _ in range(5))
    assert tag == 0x6B376B37 and flags & 0x4, "not a simple-sds GBWT"
    r.sparse_vector(); r.byte_vector(); r.int_vector()          # tags: a compressed string array (index, alphabet, symbols)
    universe, starts = r.sparse_vector()                          # BWT: where each record starts in the byte array
    body = r.byte_vector()
    assert universe == len(body) and len(starts) == alphabet_size - offset
    records = []
    for c, lo in enumerate(starts):
        hi = starts[c + 1] if c + 1 < len(starts) else len(body)
        i = lo
        sigma, i = byte_code(body, i)
        edges, node = [], 0
        for _ in range(sigma):
            delta, i = byte_code(body, i); off, i = byte_code(body, i)
            node += delta; edges.append([node, off])
        runs = []
        while i < hi:                                             # Run coding: one byte = rank + sigma * (length - 1) while that fits
            if sigma >= 255:
                rank, i = byte_code(body, i); length, i = byte_code(body, i); length += 1
            else:
                code = body[i]; i += 1
                rank, length = code % sigma, code // sigma + 1
                if length >= 256 // sigma:
                    more, i = byte_code(body, i); length += more
            runs.append([rank, length])
        records.append({"edges": edges, "runs": runs})
    header = {"version": version, "sequences": sequences, "size": size, "offset": offset, "alphabet_size": alphabet_size, "flags": flags}
    return header, records


def extract_threads(header, records):
    offset = header["offset"]

    def lf(node, i):
        rec = records[0 if node == 0 else node - offset]
        seen = [o for _, o in rec["edges"]]
        at = 0
        for rank, length in rec["runs"]:
            if i < at + length:
                return rec["edges"][rank][0], seen[rank] + i - at
            seen[rank] += length; at += length
        raise ValueError("offset past the record")
    threads = []
    for s in range(header["sequences"]):
        node, i = lf(0, s)
        t = []
        while node:
            t.append(node); node, i = lf(node, i)
        threads.append(t)
    return threads


def decode_graph(data):
    """GBWTGraph, SDSL serialization (version 3): 'GBG ', tag, version, nodes, flags; then vector<char> sequences (both
    orientations of every node) and an int_vector of their offsets."""
    assert data[:4] == b"GBG "
    tag, version = struct.unpack_from("<II", data, 4)
    nodes, flags = struct.unpack_from("<QQ", data, 12)
    assert tag == 0x6B3764AF and version == 3 and flags == 0
    at = 28
    n = struct.unpack_from("<Q", data, at)[0]; at += 8
    chars = data[at:at + n]; at += n
    bits = struct.unpack_from("<Q", data, at)[0]; at += 8
    width = data[at]; at += 1
    big = int.from_bytes(data[at:at + (bits + 63) // 64 * 8], "little")
    offs = [(big >> (i * width)) & ((1 << width) - 1) for i in range(bits // width)]
    assert len(offs) == 2 * nodes + 1
    return [chars[offs[2 * i]:offs[2 * i + 1]].decode() for i in range(nodes)]


def decode_minimizers(data):
    """MinimizerIndex version 10: tag, version, then k, w, keys, -, max_keys, values, unique, flags; the hash table as a vector of
    32-byte cells (key, position | offset into the multi-value array, 16 bytes of payload); NO_KEY = 2^63 - 1 marks a free cell."""
    tag, version = struct.unpack_from("<II", data, 0)
    k, w, keys, _, _, values, unique, flags = struct.unpack_from("<8Q", data, 8)
    assert tag == 0x31513151 and version == 10 and keys == unique == values, "multi-occurrence keys are not decoded here"
    capacity = struct.unpack_from("<Q", data, 0x48)[0]
    entries = []
    for i in range(capacity):
        key, pos = struct.unpack_from("<QQ", data, 0x50 + 32 * i)
        if key != 0x7FFFFFFFFFFFFFFF:
            # gbwtgraph's Position: offset in the low 10 bits, then the orientation, then the node id
            entries.append({"cell": i, "key": key, "id": pos >> 11, "is_reverse": (pos >> 10) & 1, "offset": pos & 1023})
    assert len(entries) == keys
    return {"k": k, "w": w, "capacity": capacity, "entries": entries}


def main():
    d = os.path.join(REF, "test", "primers")
    gbwt = open(os.path.join(d, "y.gbwt"), "rb").read()
    header, records = decode_gbwt(gbwt)
    threads = extract_threads(header, records)
    nodes = decode_graph(open(os.path.join(d, "y.gg"), "rb").read())
    minimizers = decode_minimizers(open(os.path.join(d, "y.min"), "rb").read())
    assert sum(len(t) for t in threads) + len(threads) == header["size"]
    for s in range(0, len(threads), 2):      # bidirectional: sequence 2i + 1 is sequence 2i walked backwards on the other strand
        assert threads[s + 1] == [x ^ 1 for x in reversed(threads[s])]
    out = {
        "source": "test/primers/y.gbwt, y.gg, y.min of the reference (decoded by tests/golden/extract_primers_fixture.py)",
        "gbwt_header": header, "gbwt_records": records, "gbwt_threads": threads,
        "gbwt_file_hex": gbwt.hex(),
        "gbz_file_hex": open(os.path.join(d, "y.giraffe.gbz"), "rb").read().hex(),      # the same graph and haplotypes as one GBZ container (not decoded here)
        "node_sequences": nodes,             # node id i + 1 (GBWT node 2 * (i + 1) + is_reverse)
        "minimizer_index": minimizers,
    }
    path = os.path.join(HERE, "ref_primers_y.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
        f.write("\n")
    print("wrote", path, os.path.getsize(path), "bytes:", len(nodes), "nodes,", len(threads), "sequences,", len(minimizers["entries"]), "minimizers")


if __name__ == "__main__":
    sys.exit(main())
