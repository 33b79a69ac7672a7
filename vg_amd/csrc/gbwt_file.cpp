// gbwt_file.cpp — vgk_haplo_create_gbwt: the haplotype index from the file `vg gbwt` writes (the GBWT the
// reference's GaplessExtender / WFAExtender search lives in a gbwt::GBWT loaded from such a file, src/gbwt_helper.cpp, and is reached
// through gbwtgraph::GBWTGraph).
//
// gbwt is an absent submodule; what is decoded here is its published file format, simple-sds serialization (header flag 0x4, what
// current gbwt writes by default), pinned on the one GBWT the reference keeps for its own tests (test/primers/y.gbwt; decoded
// independently by tests/golden/extract_primers_fixture.py, compared in tests/test_gbwt_file.py):
//   header   tag 0x6B376B37, version, sequences, size, offset, alphabet_size, flags              (2 x u32, 5 x u64)
//   tags     a string array: sparse vector + byte vector + int vector                            (skipped)
//   BWT      sparse vector (Elias-Fano) of the records' start offsets + the byte array of all records
//   record   ByteCode outdegree; per edge ByteCode (successor - previous successor), ByteCode offset; then runs (rank of the edge,
//            length): one byte  rank + outdegree * (length - 1)  with the length continued in a ByteCode when it reaches
//            256 / outdegree, or two ByteCodes when the outdegree is >= 255
//   (document-array samples and metadata follow; not needed)
// Record 0 is the endmarker: its i-th visit starts sequence i.  Sequences are followed with LF (the visit's edge and its rank among
// the visits taking that edge, plus the edge's offset) until they return to the endmarker.  In a bidirectional GBWT sequence 2i + 1
// is sequence 2i on the other strand, which is exactly what vgk_haplo_create derives from a thread itself: the even sequences become
// the threads, and the index built from them holds the file's records again (the test compares search states with the file's).
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <vector>
#include "ctx.hpp"
#include "haplo.hpp"
#include "host_parallel.hpp"

namespace {

struct Cursor {
    const uint8_t* p; size_t n, at = 0; bool ok = true;
    uint64_t u64() { if (at + 8 > n) { ok = false; return 0; } uint64_t v; std::memcpy(&v, p + at, 8); at += 8; return v; }
    uint32_t u32() { if (at + 4 > n) { ok = false; return 0; } uint32_t v; std::memcpy(&v, p + at, 4); at += 4; return v; }
    const uint8_t* words(uint64_t count) {
        if (count > (n - at) / 8) { ok = false; return nullptr; }
        const uint8_t* q = p + at; at += 8 * (size_t)count; return q;
    }
    void skip_option() { const uint64_t w = u64(); words(w); }
};
struct Bits {                                 // a RawVector: bit length + words
    const uint8_t* w = nullptr; uint64_t bits = 0;
    bool get(uint64_t i) const { return (w[i >> 3] >> (i & 7)) & 1; }
    uint64_t field(uint64_t at, uint32_t width) const {                    // width <= 64, little-endian bit order
        uint64_t v = 0;
        for (uint32_t b = 0; b < width; ++b) v |= (uint64_t)get(at + b) << b;
        return v;
    }
};
bool read_raw(Cursor& c, Bits& out) {
    out.bits = c.u64(); const uint64_t n = c.u64();
    if (!c.ok || out.bits > 64 * n) { c.ok = false; return false; }
    out.w = c.words(n);
    return c.ok;
}
// IntVector: length, width, RawVector
bool read_int_vector(Cursor& c, Bits& data, uint64_t& len, uint64_t& width) {
    len = c.u64(); width = c.u64();
    if (!read_raw(c, data) || width > 64 || data.bits != len * width) { c.ok = false; return false; }
    return true;
}
// SparseVector: universe; high bits as a BitVector (ones, RawVector, three optional support structures); low bits as an IntVector.
// The i-th one of `high` at position p stands for the value ((p - i) << width) | low[i].
bool read_sparse(Cursor& c, uint64_t& universe, std::vector<uint64_t>* values) {
    universe = c.u64();
    const uint64_t ones = c.u64();
    Bits high; if (!read_raw(c, high)) return false;
    c.skip_option(); c.skip_option(); c.skip_option();
    Bits low; uint64_t len = 0, width = 0;
    // a well-formed vector has one high bit per value, none of them shiftable out of 64 bits, inside the bits that are really there
    if (!read_int_vector(c, low, len, width) || len != ones || ones > high.bits || width >= 64) { c.ok = false; return false; }
    if (values) {
        values->clear(); values->reserve((size_t)ones);
        uint64_t i = 0;
        for (uint64_t p = 0; p < high.bits && i < ones; ++p) if (high.get(p)) { values->push_back(((p - i) << width) | low.field(i * width, (uint32_t)width)); ++i; }
        if (i != ones) { c.ok = false; return false; }
    }
    return c.ok;
}
bool read_bytes(Cursor& c, const uint8_t*& data, uint64_t& len) {
    len = c.u64();
    if (!c.ok || len > c.n - c.at) { c.ok = false; return false; }
    data = c.words((len + 7) / 8);
    return c.ok;
}

struct Body { const uint8_t* b; size_t end; };
inline bool byte_code(const Body& s, size_t& i, uint64_t& v) {
    v = 0;
    for (uint32_t shift = 0; shift < 64; shift += 7) {
        if (i >= s.end) return false;
        const uint8_t c = s.b[i++];
        v |= (uint64_t)(c & 0x7f) << shift;
        if (!(c & 0x80)) return true;
    }
    return false;
}
struct Edge { uint64_t node, offset; };
// One LF step inside the record [lo, hi): the visit `i` of the record leaves along which edge, and as which visit of the successor.
// edges is scratch.  false: the record is malformed or has fewer than i + 1 visits.
bool lf(const uint8_t* body, size_t lo, size_t hi, uint64_t i, std::vector<Edge>& edges, uint64_t& next_node, uint64_t& next_i) {
    const Body s{body, hi};
    size_t at = lo; uint64_t sigma;
    if (!byte_code(s, at, sigma) || sigma == 0 || sigma > hi - lo) return false;
    edges.resize((size_t)sigma);
    uint64_t node = 0;
    for (uint64_t e = 0; e < sigma; ++e) {
        uint64_t delta, off;
        if (!byte_code(s, at, delta) || !byte_code(s, at, off)) return false;
        node += delta; edges[(size_t)e] = Edge{node, off};
    }
    const uint64_t continues = sigma < 255 ? 256 / sigma : 0;
    uint64_t seen = 0;
    while (at < hi) {
        uint64_t rank, length;
        if (continues == 0) { if (!byte_code(s, at, rank) || !byte_code(s, at, length)) return false; ++length; }
        else {
            const uint8_t code = body[at++];
            rank = code % sigma; length = code / sigma + 1;
            if (length >= continues) { uint64_t more; if (!byte_code(s, at, more)) return false; length += more; }
        }
        if (rank >= sigma) return false;
        if (i < seen + length) { next_node = edges[(size_t)rank].node; next_i = edges[(size_t)rank].offset + (i - seen); return true; }
        edges[(size_t)rank].offset += length; seen += length;
    }
    return false;
}

// The GBWT at the cursor: header, tags, BWT (GBWT node (offset + 1) + o <-> oriented node o; record 0 is the endmarker's).
// whole: also step over what follows the BWT (document-array samples, metadata: both optional structures with a size in front).
struct Bwt { uint64_t sequences = 0, size = 0, offset = 0, alphabet = 0; uint32_t n_nodes = 0; std::vector<uint64_t> starts; const uint8_t* body = nullptr; uint64_t body_len = 0; };
int read_bwt(Cursor& c, bool whole, Bwt& B) {
    const uint32_t tag = c.u32(); c.u32();
    const uint64_t sequences = c.u64(), size = c.u64(), offset = c.u64(), alphabet = c.u64(), flags = c.u64();
    if (!c.ok || tag != 0x6B376B37u) return VGK_EINVAL;
    if (!(flags & 0x4u)) return VGK_EUNSUPPORTED;                           // SDSL serialization (older files): `vg gbwt` rewrites them
    if (!(flags & 0x1u)) return VGK_EUNSUPPORTED;                           // unidirectional: the extenders need both strands
    if (alphabet <= offset + 1 || ((offset + 1) & 1) || ((alphabet - offset - 1) & 1) || alphabet - offset - 1 > 0xfffffff0ull || (sequences & 1) || size > 0xfffffff0ull) return VGK_EINVAL;
    if (sequences > size) return VGK_EINVAL;                                // every sequence visits the endmarker once, and `size` counts those visits too
    B.sequences = sequences; B.size = size; B.offset = offset; B.alphabet = alphabet;
    B.n_nodes = (uint32_t)((alphabet - offset - 1) / 2);
    std::vector<uint64_t>& starts = B.starts;
    uint64_t universe = 0;
    if (!read_sparse(c, universe, nullptr)) return VGK_EINVAL;               // tags: index ...
    { const uint8_t* a; uint64_t n; if (!read_bytes(c, a, n)) return VGK_EINVAL; }      // ... alphabet ...
    { Bits d; uint64_t n, w; if (!read_int_vector(c, d, n, w)) return VGK_EINVAL; }     // ... symbols
    if (!read_sparse(c, universe, &starts)) return VGK_EINVAL;
    const uint8_t* body = nullptr; uint64_t body_len = 0;
    if (!read_bytes(c, body, body_len) || body_len != universe || starts.size() != alphabet - offset) return VGK_EINVAL;
    for (size_t r = 0; r < starts.size(); ++r) if (starts[r] > body_len || (r && starts[r] < starts[r - 1])) return VGK_EINVAL;
    starts.push_back(body_len);
    B.body = body; B.body_len = body_len;
    if (whole) { c.skip_option(); c.skip_option(); if (!c.ok) return VGK_EINVAL; }
    return VGK_OK;
}

// -> the even sequences as threads of oriented nodes: each followed with LF from the endmarker until it returns there
int read_gbwt(Cursor& c, bool whole, uint32_t& n_nodes, std::vector<uint32_t>& thread_off, std::vector<uint32_t>& thread_nodes) {
    Bwt B;
    if (int rc = read_bwt(c, whole, B)) return rc;
    n_nodes = B.n_nodes;
    const uint64_t sequences = B.sequences, size = B.size, offset = B.offset, alphabet = B.alphabet;
    const std::vector<uint64_t>& starts = B.starts; const uint8_t* body = B.body;
    // the even sequences, one host task each: (endmarker, s) -> first node -> ... -> endmarker
    const uint32_t n_threads = (uint32_t)(sequences / 2);
    std::vector<std::vector<uint32_t>> walks(n_threads);
    std::vector<int> status(n_threads, VGK_OK);
    vgk::parallel_tasks(n_threads, [&](uint32_t t) {
        std::vector<Edge> edges;
        std::vector<uint32_t>& walk = walks[t];
        uint64_t node = 0, i = 2ull * t;
        for (uint64_t steps = 0;; ++steps) {
            const uint64_t rec = node == 0 ? 0 : node - offset;
            uint64_t nn, ni;
            if (steps > size || !lf(body, (size_t)starts[(size_t)rec], (size_t)starts[(size_t)rec + 1], i, edges, nn, ni)) { status[t] = VGK_EINVAL; return; }
            if (nn == 0) return;
            if (nn <= offset || nn >= alphabet) { status[t] = VGK_EINVAL; return; }
            walk.push_back((uint32_t)(nn - offset - 1));
            node = nn; i = ni;
        }
    });
    for (int rc : status) if (rc) return rc;
    thread_off.assign((size_t)n_threads + 1, 0); thread_nodes.clear();
    uint64_t visits = 0;                                                    // summed in 64 bits: the offsets are 32-bit
    for (uint32_t t = 0; t < n_threads; ++t) {
        visits += walks[t].size();
        if (visits > 0xfffffff0ull) return VGK_ETOOBIG;
        thread_off[t + 1] = (uint32_t)visits;
    }
    thread_nodes.reserve(thread_off[n_threads]);
    for (auto& w : walks) thread_nodes.insert(thread_nodes.end(), w.begin(), w.end());
    return VGK_OK;
}

// The records taken over as they are (round 3): a GBWT record already IS what the engine's index holds per oriented node — the visits in
// GBWT order as runs of (edge, length), the edges in successor order, each with the rank of its first visit in the successor's record —
// so the tables of vgk_haplo_from_tables are one decoding pass per record, on the host threads; no sequence is followed.
int gbwt_tables(const Bwt& B, HaploTables& T) {
    const uint32_t O = 2 * B.n_nodes;
    T.count.assign(O, 0); T.body_off.assign(O + 1, 0); T.edge_off.assign(O + 1, 0);
    std::vector<int> bad(O, 0);
    auto decode = [&](uint32_t o, bool fill) {
        const size_t lo = (size_t)B.starts[(size_t)o + 1], hi = (size_t)B.starts[(size_t)o + 2];
        if (lo == hi) return;                                                // (a node no sequence visits)
        const Body s{B.body, hi};
        size_t at = lo; uint64_t sigma;
        if (!byte_code(s, at, sigma) || sigma > hi - lo) { bad[o] = 1; return; }
        if (sigma == 0) return;
        if (sigma > 255) { bad[o] = 2; return; }                             // (edge numbers are bytes in the engine's records)
        uint64_t node = 0;
        for (uint64_t e = 0; e < sigma; ++e) {
            uint64_t delta, off;
            if (!byte_code(s, at, delta) || !byte_code(s, at, off)) { bad[o] = 1; return; }
            node += delta;
            if (node != 0 && (node <= B.offset || node >= B.alphabet)) { bad[o] = 1; return; }
            if (e && delta == 0) { bad[o] = 1; return; }
            if (off > 0xfffffff0ull) { bad[o] = 1; return; }
            if (fill) { T.edge_to[T.edge_off[o] + e] = node == 0 ? -1 : (int32_t)(node - B.offset - 1); T.edge_base[T.edge_off[o] + e] = node == 0 ? 0u : (uint32_t)off; }
        }
        const uint64_t continues = sigma < 255 ? 256 / sigma : 0;
        uint64_t seen = 0;
        uint64_t through[256] = {0};                                          // visits that leave over each edge
        while (at < hi) {
            uint64_t rank, length;
            if (continues == 0) { if (!byte_code(s, at, rank) || !byte_code(s, at, length)) { bad[o] = 1; return; } ++length; }
            else {
                const uint8_t code = B.body[at++];
                rank = code % sigma; length = code / sigma + 1;
                if (length >= continues) { uint64_t more; if (!byte_code(s, at, more)) { bad[o] = 1; return; } length += more; }
            }
            if (rank >= sigma || length > 0xfffffff0ull - seen) { bad[o] = 1; return; }
            if (fill) for (uint64_t k = 0; k < length; ++k) T.body[T.body_off[o] + seen + k] = (uint32_t)rank;
            seen += length; through[rank] += length;
        }
        if (!fill) {
            if (seen == 0) { bad[o] = 1; return; }                            // edges but no visit: not a record a GBWT writes
            T.count[o] = (uint32_t)seen; T.edge_off[o + 1] = (uint32_t)sigma;
            return;
        }
        // An edge's offset is the rank of its first visit in the successor's record, and the visits it carries follow from there: they must
        // lie inside that record (every count is known since the first pass) — the kernels form their search ranges from exactly these
        // numbers, and the engine's records keep the offset in 16 bits (vgk_haplo_from_tables).  A file that says otherwise is refused here.
        for (uint64_t e = 0; e < sigma; ++e) {
            const int32_t to = T.edge_to[T.edge_off[o] + e];
            if (to < 0) continue;
            const uint64_t base = T.edge_base[T.edge_off[o] + e];
            if (base + through[e] > T.count[(uint32_t)to]) { bad[o] = 1; return; }
            if (base > 65535u) { bad[o] = 2; return; }
        }
    };
    vgk::parallel_for(O, [&](uint32_t o, unsigned) { decode(o, false); });
    for (uint32_t o = 0; o < O; ++o) if (bad[o]) return bad[o] == 2 ? VGK_ETOOBIG : VGK_EINVAL;
    { // the endmarker's own record is not part of the index, but it must be what the header says: one visit per sequence, every edge a node
      const Body s{B.body, (size_t)B.starts[1]};
      size_t at = (size_t)B.starts[0]; uint64_t sigma = 0, node = 0, seen = 0;
      if (!byte_code(s, at, sigma) || sigma > s.end - (size_t)B.starts[0] || (sigma == 0 && B.sequences)) return VGK_EINVAL;
      for (uint64_t e = 0; e < sigma; ++e) {
          uint64_t delta, off;
          if (!byte_code(s, at, delta) || !byte_code(s, at, off)) return VGK_EINVAL;
          node += delta;
          if (node <= B.offset || node >= B.alphabet) return VGK_EINVAL;
      }
      const uint64_t continues = sigma && sigma < 255 ? 256 / sigma : 0;
      while (sigma && at < s.end) {
          uint64_t rank, length;
          if (continues == 0) { if (!byte_code(s, at, rank) || !byte_code(s, at, length)) return VGK_EINVAL; ++length; }
          else { const uint8_t code = B.body[at++]; rank = code % sigma; length = code / sigma + 1; if (length >= continues) { uint64_t more; if (!byte_code(s, at, more)) return VGK_EINVAL; length += more; } }
          if (rank >= sigma || length > B.size) return VGK_EINVAL;
          seen += length;
      }
      if (seen != B.sequences) return VGK_EINVAL; }
    uint64_t visits = 0, edges = 0;
    for (uint32_t o = 0; o < O; ++o) {
        T.body_off[o] = (uint32_t)visits; visits += T.count[o];
        const uint32_t ne = T.edge_off[o + 1]; T.edge_off[o] = (uint32_t)edges; edges += ne;
        if (visits > 0xfffffff0ull || edges > 0xfffffff0ull) return VGK_ETOOBIG;
    }
    T.body_off[O] = (uint32_t)visits; T.edge_off[O] = (uint32_t)edges;
    if (visits + B.sequences != B.size) return VGK_EINVAL;                   // `size` counts every visit, the endmarker's included
    T.body.assign((size_t)visits, 0); T.edge_to.assign((size_t)edges, -1); T.edge_base.assign((size_t)edges + 1, 0);
    vgk::parallel_for(O, [&](uint32_t o, unsigned) { decode(o, true); });
    for (uint32_t o = 0; o < O; ++o) if (bad[o]) return bad[o] == 2 ? VGK_ETOOBIG : VGK_EINVAL;
    return VGK_OK;
}

struct Loaded { vgk_haplotypes h; std::vector<uint32_t> node_len, thread_off, thread_nodes; std::string seq; };

}  // namespace

extern "C" {

int vgk_haplo_create_gbwt(vgk_ctx* ctx, const void* gbwt, size_t bytes, uint32_t n_nodes, const uint32_t* node_len, const char* seq, vgk_haplo** out) {
    if (!ctx || !gbwt || !out || !n_nodes || !node_len || !seq) return VGK_EINVAL;
    *out = nullptr;
    try {
        Cursor c{(const uint8_t*)gbwt, bytes};
        if (!std::getenv("VGAMD_GBWT_VIA_THREADS")) {
            // the file's records become the index's as they are (no sequence is followed: what a whole-genome GBWT needs)
            Bwt B;
            if (int rc = read_bwt(c, false, B)) return rc;
            if (B.n_nodes != n_nodes) return VGK_EINVAL;
            HaploTables T;
            if (int rc = gbwt_tables(B, T)) return rc;
            std::vector<uint32_t> len, seq_off; std::vector<char> strands; uint64_t total = 0;
            if (int rc = vgk_haplo_strands(n_nodes, node_len, seq, len, seq_off, strands, total)) return rc;
            return vgk_haplo_from_tables(ctx, 2 * n_nodes, len, seq_off, strands, (uint32_t)total, T, out);
        }
        // (the round-2 way, kept for comparison: every sequence followed with LF, the index rebuilt from the threads)
        uint32_t nodes_in_file = 0; std::vector<uint32_t> thread_off, thread_nodes;
        const int rc = read_gbwt(c, false, nodes_in_file, thread_off, thread_nodes);
        if (rc) return rc;
        if (nodes_in_file != n_nodes) return VGK_EINVAL;
        vgk_haplotypes d{};
        d.n_nodes = n_nodes; d.node_len = node_len; d.seq = seq;
        d.n_threads = (uint32_t)(thread_off.size() - 1); d.thread_off = thread_off.data(); d.thread_nodes = thread_nodes.data();
        return vgk_haplo_create(ctx, &d, out);
    } catch (const std::bad_alloc&) { return VGK_ENOMEM; }
    catch (...) { return VGK_EINVAL; }                                     // (a malformed image that slipped past the checks above)
}

// GBZ (gbwtgraph's container, what `vg giraffe -Z` takes): 'GBZ ' header, tags, the GBWT whole, then the GBWTGraph — header (tag
// 0x6B3764AF, version, nodes, flags) and the node sequences, forward strands only, as a compressed string array: sparse vector of the
// strings' start offsets, the alphabet, the symbols as an int vector over it.  (Segment names / translation follow: not needed.)
int vgk_gbz_load(const void* gbz, size_t bytes, vgk_haplotypes** out) {
    if (!gbz || !out) return VGK_EINVAL;
    *out = nullptr;
    try {
    Cursor c{(const uint8_t*)gbz, bytes};
    const uint32_t tag = c.u32(); c.u32(); c.u64();
    if (!c.ok || tag != 0x205A4247u) return VGK_EINVAL;
    uint64_t universe = 0;
    if (!read_sparse(c, universe, nullptr)) return VGK_EINVAL;               // the container's tags
    { const uint8_t* a; uint64_t n; if (!read_bytes(c, a, n)) return VGK_EINVAL; }
    { Bits d; uint64_t n, w; if (!read_int_vector(c, d, n, w)) return VGK_EINVAL; }
    std::unique_ptr<Loaded> L(new (std::nothrow) Loaded());
    if (!L) return VGK_ENOMEM;
    uint32_t n_nodes = 0;
    int rc = read_gbwt(c, true, n_nodes, L->thread_off, L->thread_nodes);
    if (rc) return rc;
    const uint32_t gtag = c.u32(); c.u32();
    const uint64_t nodes = c.u64(), gflags = c.u64();
    if (!c.ok || gtag != 0x6B3764AFu || nodes != n_nodes) return VGK_EINVAL;
    if (!(gflags & 0x2u)) return VGK_EUNSUPPORTED;                          // SDSL-serialized graph
    std::vector<uint64_t> starts;
    if (!read_sparse(c, universe, &starts) || starts.size() != nodes) return VGK_EINVAL;
    const uint8_t* alpha = nullptr; uint64_t n_alpha = 0;
    if (!read_bytes(c, alpha, n_alpha)) return VGK_EINVAL;
    Bits sym; uint64_t n_sym = 0, width = 0;
    if (!read_int_vector(c, sym, n_sym, width) || n_sym > 0xfffffff0ull) return VGK_EINVAL;
    L->seq.resize((size_t)n_sym);
    for (uint64_t i = 0; i < n_sym; ++i) { const uint64_t s = width ? sym.field(i * width, (uint32_t)width) : 0; if (s >= n_alpha) return VGK_EINVAL; L->seq[(size_t)i] = (char)alpha[s]; }
    L->node_len.resize(n_nodes);
    for (uint32_t v = 0; v < n_nodes; ++v) {
        const uint64_t a = starts[v], b = v + 1 < n_nodes ? starts[v + 1] : n_sym;
        if (a > b || b > n_sym) return VGK_EINVAL;
        L->node_len[v] = (uint32_t)(b - a);
    }
    L->h.n_nodes = n_nodes; L->h.node_len = L->node_len.data(); L->h.seq = L->seq.data();
    L->h.n_threads = (uint32_t)(L->thread_off.size() - 1); L->h.thread_off = L->thread_off.data(); L->h.thread_nodes = L->thread_nodes.data();
    *out = &L.release()->h;                                                  // (h is the first member: vgk_haplotypes_free casts back)
    return VGK_OK;
    } catch (const std::bad_alloc&) { return VGK_ENOMEM; }
    catch (...) { return VGK_EINVAL; }
}
void vgk_haplotypes_free(vgk_haplotypes* h) { delete reinterpret_cast<Loaded*>(h); }

}  // extern "C"
