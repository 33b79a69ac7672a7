// gaf_output.cpp — see gaf_output.hpp.
#include "gaf_output.hpp"
#include <atomic>
#include <thread>
#include <vector>
#include <algorithm>
#include <cstring>

namespace vgamd {

namespace {

inline char complement(char c) {
    switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A';
                 case 'a': return 'T'; case 'c': return 'G'; case 'g': return 'C'; case 't': return 'A'; default: return 'N'; }
}
inline char upper(char c) { return (c >= 'a' && c <= 'z') ? (char)(c - 'a' + 'A') : c; }

// base `at` of an oriented node, read along its strand
inline char oriented_base(const GafGraphView& g, uint32_t oriented, uint32_t at) {
    const uint32_t v = oriented >> 1;
    const uint64_t b = g.node_off[v], e = g.node_off[v + 1];
    return (oriented & 1u) ? complement(g.node_seq[e - 1 - at]) : upper(g.node_seq[b + at]);
}
inline uint32_t node_length(const GafGraphView& g, uint32_t oriented) { const uint32_t v = oriented >> 1; return (uint32_t)(g.node_off[v + 1] - g.node_off[v]); }

void put_number(std::string& out, long long x) { char b[24]; int n = 0; bool neg = x < 0; unsigned long long u = neg ? 0ull - (unsigned long long)x : (unsigned long long)x;
    do { b[n++] = (char)('0' + u % 10); u /= 10; } while (u); if (neg) out.push_back('-'); while (n) out.push_back(b[--n]); }

}  // namespace

bool gaf_record(std::string& out, const char* name, const char* seq, uint32_t seq_len, const vgk_chain_result& res, const vgk_chain_mapping* maps,
                const uint32_t* runs, const GafGraphView& graph, int mapq, const int32_t* score) {
    const bool aligned = res.status == VGK_OK && res.n_mappings > 0;
    // the path: a step per mapping that consumes graph bases (a mapping of insertions alone — a soft clip on a node of its own — is no step)
    std::string path, cs;
    uint64_t path_length = 0, path_start = 0, path_end = 0, matches = 0, block = 0;
    uint32_t read_at = 0;
    uint64_t pending_match = 0;
    auto flush_match = [&]() { if (pending_match) { cs.push_back(':'); put_number(cs, (long long)pending_match); pending_match = 0; } };
    bool first_step = true;
    uint32_t last_node = 0, last_offset = 0, last_from = 0;
    if (aligned) {
        for (uint32_t k = 0; k < res.n_mappings; ++k) {
            const vgk_chain_mapping& m = maps[res.mapping_begin + k];
            uint32_t from = 0;
            for (uint32_t e = 0; e < m.n_edits; ++e) { const uint32_t r = runs[m.edit_begin + e]; if ((r & 3u) != (uint32_t)VGK_WFA_INSERTION) from += r >> 2; }
            const bool placed = m.node != VGK_WFA_NO_NODE;
            if (placed && (m.node >> 1) >= graph.n_nodes) return false;
            if (from && !placed) return false;
            if (from) {
                if (m.offset + from > node_length(graph, m.node)) return false;
                path.push_back((m.node & 1u) ? '<' : '>');
                put_number(path, graph.node_ids ? (long long)graph.node_ids[m.node >> 1] : (long long)(m.node >> 1) + 1);
                path_length += node_length(graph, m.node);
                if (first_step) { path_start = m.offset; first_step = false; }
                last_node = m.node; last_offset = m.offset; last_from = from;
            }
            uint32_t ref_at = m.offset;
            for (uint32_t e = 0; e < m.n_edits; ++e) {
                const uint32_t r = runs[m.edit_begin + e], kind = r & 3u, len = r >> 2;
                if (!len) continue;
                if (kind != (uint32_t)VGK_WFA_DELETION && read_at + len > seq_len) return false;
                switch (kind) {
                case VGK_WFA_MATCH: pending_match += len; matches += len; block += len; read_at += len; ref_at += len; break;
                case VGK_WFA_MISMATCH:
                    flush_match();
                    for (uint32_t j = 0; j < len; ++j) { cs.push_back('*'); cs.push_back(oriented_base(graph, m.node, ref_at + j)); cs.push_back(upper(seq[read_at + j])); }
                    block += len; read_at += len; ref_at += len; break;
                case VGK_WFA_INSERTION:
                    flush_match(); cs.push_back('+');
                    for (uint32_t j = 0; j < len; ++j) cs.push_back(upper(seq[read_at + j]));
                    block += len; read_at += len; break;
                default:
                    flush_match(); cs.push_back('-');
                    for (uint32_t j = 0; j < len; ++j) cs.push_back(oriented_base(graph, m.node, ref_at + j));
                    block += len; ref_at += len; break;
                }
            }
        }
        flush_match();
        if (read_at != seq_len) return false;
        if (!first_step) path_end = path_length - (node_length(graph, last_node) - last_offset - last_from);
    } else {
        cs.push_back('+');
        for (uint32_t j = 0; j < seq_len; ++j) cs.push_back(upper(seq[j]));
    }
    out.append(name && *name ? name : "*"); out.push_back('\t');
    put_number(out, seq_len); out.append("\t0\t"); put_number(out, seq_len); out.append("\t+\t");
    out.append(path.empty() ? "*" : path); out.push_back('\t');
    put_number(out, (long long)path_length); out.push_back('\t'); put_number(out, (long long)path_start); out.push_back('\t'); put_number(out, (long long)path_end); out.push_back('\t');
    put_number(out, (long long)matches); out.push_back('\t'); put_number(out, (long long)block); out.push_back('\t'); put_number(out, mapq);
    if (score) { out.append("\tAS:i:"); put_number(out, *score); }
    out.append("\tcs:Z:"); out.append(cs);
    return true;
}

}  // namespace vgamd

// ---- the batch form behind the C ABI of the host shim --------------------------------------------------------------------------------------------
extern "C" {
// GAF lines (each ends in '\n') of a batch of composed alignments into `out`; line_off[r] .. line_off[r + 1] = read r's line.  names: n_reads C strings, or
// null ("*"); seqs / seq_off: the reads' bases; res / maps / runs: vgk_chain_stitch's output (vgh_chain_stage_view); node_seq / node_off / node_ids: the graph
// (GafGraphView); mapq / score: per read, or null (0 / no AS tag).  -> the bytes all lines need (nothing is written past out_cap: call again with room), or -1
// with *bad_read = the first read whose alignment does not fit the graph or its own length.
int64_t vgh_gaf_lines(uint32_t n_reads, const char* const* names, const char* seqs, const uint64_t* seq_off, const vgk_chain_result* res,
                      const vgk_chain_mapping* maps, const uint32_t* runs, const char* node_seq, const uint64_t* node_off, uint32_t n_nodes,
                      const int64_t* node_ids, const int32_t* mapq, const int32_t* score, char* out, uint64_t out_cap, uint64_t* line_off, int threads,
                      uint32_t* bad_read) {
    using namespace vgamd;
    const GafGraphView g{node_seq, node_off, n_nodes, node_ids};
    const unsigned T = (unsigned)std::max(1, std::min<int>(threads, (int)std::max<uint32_t>(1u, n_reads / 64u)));
    std::vector<std::string> part(T);
    std::vector<std::vector<uint32_t>> lens(T);
    std::atomic<uint32_t> bad{0xffffffffu};
    auto work = [&](unsigned t) {
        const uint32_t b = (uint32_t)((uint64_t)n_reads * t / T), e = (uint32_t)((uint64_t)n_reads * (t + 1) / T);
        std::string& s = part[t]; lens[t].reserve(e - b);
        for (uint32_t r = b; r < e; ++r) {
            const size_t before = s.size();
            const bool ok = gaf_record(s, names ? names[r] : nullptr, seqs + seq_off[r], (uint32_t)(seq_off[r + 1] - seq_off[r]), res[r], maps, runs, g,
                                       mapq ? mapq[r] : 0, score ? score + r : nullptr);
            if (!ok) { uint32_t seen = bad.load(); while (r < seen && !bad.compare_exchange_weak(seen, r)) {} s.resize(before); }
            s.push_back('\n');
            lens[t].push_back((uint32_t)(s.size() - before));
        }
    };
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < T; ++t) pool.emplace_back(work, t);
    work(0);
    for (std::thread& th : pool) th.join();
    if (bad.load() != 0xffffffffu) { if (bad_read) *bad_read = bad.load(); return -1; }
    uint64_t total = 0, r = 0;
    for (unsigned t = 0; t < T; ++t) {
        if (total + part[t].size() <= out_cap && out) std::memcpy(out + total, part[t].data(), part[t].size());
        for (uint32_t l : lens[t]) { if (line_off) line_off[r] = total; total += l; ++r; }
    }
    if (line_off) line_off[n_reads] = total;
    return (int64_t)total;
}
}  // extern "C"
