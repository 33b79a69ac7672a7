// gaf_output.hpp — GAF records of composed alignments (the output half of the mapping stage).
//
// What it replaces: `alignment_to_gaf(graph, aln, …)` + the emitter's `<<` of the record (called from src/hts_alignment_emitter.cpp and
// src/multipath_alignment_emitter.cpp:167,288).  The function itself lives in libvgio, an EMPTY submodule of the reference snapshot, so what is restated here
// is the GAF format (lh3/gfatools, doc/rGFA.md "the Graph Alignment Format") as the reference's OWN tests pin it:
//   * src/unittest/alignment.cpp:398-468  "Conversion to GAF removes an unused final node": query interval = the whole read (soft clips are insertions of
//     the difference string), a mapping that consumes no graph base contributes no path step, path_end counts from the first step's start, and the difference
//     string merges matches across mappings — ":5*AT:1+AC", bases in upper case;
//   * src/unittest/alignment.cpp:792-816  an alignment without a path: no steps, difference string "+" + the read;
//   * test/surject/opposite_strands.gaf   the column layout of a record, a path of eight steps on either strand, matches / block length / quality columns.
// PARITY-UNPINNED beyond those vectors (tests/golden/ref_gaf.json): the optional tags' choice and order, block length when a read has indels.
//
// MI355X-first: the long-read stage leaves its alignments as flat mappings and edit runs (vgk_chain_stitch); a record is formatted straight from them — no
// Alignment object, no protobuf — by as many host threads as the caller gives, each into its own stretch of one output buffer.
#pragma once
#include <cstdint>
#include <string>
#include "../../include/vgk.h"

namespace vgamd {

struct GafGraphView {
    const char* node_seq;            // forward strands of the nodes, behind each other
    const uint64_t* node_off;        // [n_nodes + 1] node v = node_seq[node_off[v], node_off[v + 1])
    uint32_t n_nodes;
    const int64_t* node_ids;         // [n_nodes] the names to print, or null: v + 1
};

// One record (no newline) appended to `out`.  maps / runs: the arrays vgk_chain_stitch filled (a mapping's edit_begin indexes `runs`); a mapping's node is an
// ORIENTED node (2 v + reverse) or VGK_WFA_NO_NODE.  score: printed as AS:i when not null.  -> false: a mapping or run that does not fit the graph / the read.
bool gaf_record(std::string& out, const char* name, const char* seq, uint32_t seq_len, const vgk_chain_result& res, const vgk_chain_mapping* maps,
                const uint32_t* runs, const GafGraphView& graph, int mapq, const int32_t* score);

}  // namespace vgamd
