/*
 * vgk.h — C ABI of the MI355X-native per-read graph-alignment engine.
 *
 * This is the drop-in boundary for vg's alignment hot path.  Every entry point
 * replaces a group of third-party C calls that vg's `Aligner` family makes
 * today (citations are into the reference tree, vgteam/vg):
 *
 *   vgk_gssw_*      replaces  gssw_graph_create / gssw_node_create /
 *                             gssw_nodes_add_edge / gssw_graph_fill_pinned /
 *                             gssw_graph_trace_back /
 *                             gssw_graph_trace_back_pinned_multi
 *                             (src/aligner.cpp:30-85, 396-435, 537-557, 575-611)
 *   vgk_gssw_* with mode VGK_XDROP_PINNED
 *                   replaces  dz_init / dz_pack_query_* / dz_extend / dz_trace
 *                             as driven by DozeuInterface::align_pinned
 *                             (src/dozeu_interface.cpp:210-307, 687-766)
 *   vgk_banded_*    replaces  BandedGlobalAligner<IntType>(...).align(...) as driven by
 *                             Aligner::align_global_banded / _multi
 *                             (src/aligner.cpp:662-760, src/banded_global_aligner.cpp:250-742)
 *   vgk_haplo_*, vgk_gapless_extend
 *                   replaces  GaplessExtender::extend over a GBWTGraph
 *                             (src/gbwt_extender.cpp:533-737)
 *   vgk_wfa_extend  replaces  WFAExtender::connect / prefix / suffix
 *                             (src/gbwt_extender.cpp:2052-2263)
 *   vgk_chain_stitch
 *                   replaces  the Path composition of MinimizerMapper::find_chain_alignment
 *                             (src/minimizer_mapper_from_chains.cpp:2606-3295: append_path of every piece's Path, simplify)
 *
 * Everything is plain pointers and sizes.  All "graphs" handed over are DAGs
 * whose nodes are ALREADY in the topological order the reference would use
 * (src/aligner.cpp:32 `lazier_topological_order`), forward strand only, with
 * predecessor lists in CSR form — exactly the information
 * `create_gssw_graph` (src/aligner.cpp:30-85) extracts from a HandleGraph.
 *
 * Batch model: vg calls one (read, subgraph) problem at a time from OpenMP
 * threads; a GPU needs thousands per launch.  The ABI is therefore
 * batch-first: pack N problems (host → HBM), run (kernels only), fetch
 * (HBM → host).  A batch of 1 gives the reference's synchronous semantics.
 *
 * Error model: return codes (0 = OK, <0 = VGK_E*), never exit()/abort(), and no C++ exception crosses this boundary (an allocation
 * that fails inside a call comes back as VGK_ENOMEM);
 * per-problem failures are reported in vgk_result.status.
 */
#ifndef VGK_H
#define VGK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VGK_ABI_VERSION 6

/* ---- status codes ------------------------------------------------------- */
enum {
    VGK_OK           = 0,
    VGK_EINVAL       = -1,  /* malformed input (e.g. non-topological edge)      */
    VGK_ENODEV       = -2,  /* no usable HIP device / HIP runtime error         */
    VGK_ENOMEM       = -3,  /* host or device allocation failed                 */
    VGK_ETOOLONG     = -4,  /* read longer than the engine supports             */
    VGK_EOVERFLOW    = -5,  /* score left the int16 range gssw/dozeu support    */
    VGK_EOPS         = -6,  /* CIGAR did not fit the per-problem op budget      */
    VGK_ETOOBIG      = -7,  /* banded matrices exceed max_cells
                               (BandMatricesTooBigException,
                                src/banded_global_aligner.hpp:40-43)            */
    VGK_ENOBAND      = -8,  /* no alignment inside the band
                               (NoAlignmentInBandException, :31-38)             */
    VGK_EUNSUPPORTED = -9   /* scoring parameters outside the kernels' range    */
};

/* ---- scoring (mirrors MatrixAlignmentScorer, src/alignment_scorer.cpp:284-314) */
typedef struct vgk_scoring {
    int8_t  matrix[25];      /* 5x5 row-major, index 5*nt[ref]+nt[read]; row/col 4 = N = 0 */
    uint8_t gap_open;        /* gap of length n costs gap_open + (n-1)*gap_extend          */
    uint8_t gap_extend;
    int8_t  full_length_bonus;
    uint8_t reserved;
} vgk_scoring;

/* ---- a DAG in topological order (what create_gssw_graph builds) ---------- */
typedef struct vgk_graph {
    uint32_t        n_nodes;
    const uint32_t* node_len;   /* [n_nodes]   bases per node                          */
    const char*     seq;        /* concatenated node sequences, ASCII, order = nodes   */
    const uint32_t* pred_off;   /* [n_nodes+1] CSR offsets into pred_idx               */
    const uint32_t* pred_idx;   /* predecessor node indices, each < the node's index   */
} vgk_graph;

/* ---- CIGAR element: one run on one node ---------------------------------- */
enum { VGK_OP_M = 0,   /* match-or-mismatch run (gssw 'M'/'X'/'N'; the caller
                          re-splits by character compare, src/aligner.cpp:171-205) */
       VGK_OP_I = 1,   /* insertion (read bases, no graph bases)                   */
       VGK_OP_D = 2,   /* deletion  (graph bases, no read bases)                   */
       VGK_OP_S = 3 }; /* soft clip (only first / last element)                    */

typedef struct vgk_op {
    uint32_t node;      /* index into the problem's node order                      */
    uint16_t len;
    uint8_t  op;        /* VGK_OP_*                                                 */
    uint8_t  pad;
} vgk_op;

/* ---- result header (16 B payload + bookkeeping) --------------------------- */
typedef struct vgk_result {
    int32_t  score;         /* gm->score / alignment->score1                        */
    int32_t  status;        /* VGK_OK or VGK_E* for this problem                    */
    int32_t  end_node;      /* node index of the last aligned graph base            */
    int32_t  end_offset;    /* offset of that base in end_node (gssw ref_end1)      */
    int32_t  end_read;      /* last aligned read base (gssw read_end1)              */
    int32_t  first_offset;  /* offset in the first node where the alignment starts
                               (gssw_graph_mapping.position)                       */
    uint32_t n_ops;         /* number of vgk_op written for this problem            */
    uint32_t ops_begin;     /* index of the first one in the batch's op array       */
} vgk_result;

/* ---- graph Smith-Waterman (gssw semantics) -------------------------------- */
enum { VGK_GSSW_LOCAL       = 0,  /* Aligner::align: bonus at both ends (src/aligner.cpp:399-402) */
       VGK_GSSW_PINNED      = 1,  /* Aligner::align_pinned (right-pinned; the caller reverses
                                     graph+read for pin_left, src/aligner.cpp:365-383)            */
       VGK_XDROP_PINNED     = 2,  /* Aligner::align_pinned(..., xdrop = true): dozeu semantics
                                     (DozeuInterface::align_pinned, src/dozeu_interface.cpp:724-766):
                                     LEFT-pinned semi-global extension from every source node, bonus
                                     on consuming the last read base only, no local restart; the
                                     caller reverses graph+read for a right pin.  Runs on the same
                                     kernels and entry points as the gssw modes.                  */
       VGK_GSSW_TRACEBACK   = 16  /* OR-ed in: produce CIGAR; otherwise score + end only
                                     (src/aligner.cpp:550-557)                                    */ };

typedef struct vgk_gssw_problem {
    const char*    read;         /* ASCII read                                         */
    uint32_t       read_len;
    uint32_t       flags;        /* VGK_GSSW_*                                         */
    vgk_graph      graph;
    const uint8_t* pinning;      /* PINNED only: [n_nodes] 1 = pinning node
                                    (identify_pinning_points, src/aligner.cpp:87-118)  */
    uint32_t       max_gap_length; /* XDROP only: dozeu max_gap_length (dz_align_init; clamped to >= 1,
                                      src/aligner.cpp:638) — bounds the leading insertion         */
    uint32_t       reserved;
    const uint8_t* qual;         /* quality-adjusted contexts only: [read_len] raw phred bytes (not ASCII-33),
                                    as in Alignment.quality (src/aligner.cpp:942-952)              */
} vgk_gssw_problem;

/* ---- quality-adjusted scoring (QualAdjAlignmentScorer, src/alignment_scorer.cpp:419-513) ------------- */
typedef struct vgk_qual_adj {
    const int8_t* matrix;    /* [256][25]: index 25*qual + 5*nt[ref] + nt[read]  (src/banded_global_aligner.cpp:685) */
    const int8_t* bonuses;   /* [256]: full-length bonus by the quality of the end base (:494-513, :555-563)          */
} vgk_qual_adj;

typedef struct vgk_ctx   vgk_ctx;     /* one per (device, scoring) — like one Aligner      */
typedef struct vgk_batch vgk_batch;   /* packed problems resident in HBM                   */

int         vgk_abi_version(void);
const char* vgk_strerror(int code);

/* Create/destroy an engine bound to HIP device `device`.  Mirrors constructing
 * an Aligner (src/aligner.hpp:164-168); thread-safe for concurrent batches. */
int  vgk_create(int device, const vgk_scoring* scoring, vgk_ctx** out);
/* Same for a QualAdjAligner (src/aligner.hpp:218-258): every problem must then carry `qual`; the bonus at a read
 * end is bonuses[quality of that end base] (gssw: both ends / unpinned end; X-drop: the far end). */
int  vgk_create_qual_adj(int device, const vgk_scoring* scoring, const vgk_qual_adj* qual_adj, vgk_ctx** out);
void vgk_destroy(vgk_ctx* ctx);
int  vgk_device_info(vgk_ctx* ctx, char* name_out, size_t name_cap,
                     int* compute_units, size_t* hbm_bytes);
/* A caller that streams batches out of buffers it keeps (reads, problem arrays) can page-lock them once: copies from a registered range
 * run at the link's rate (~50 GB/s on PCIe 5 x16) instead of being staged through the runtime's bounce buffers (~10 GB/s — 150 MB of reads
 * were 12 of the seeding call's 18 ms).  The range must stay mapped until it is unregistered; registering costs about a millisecond per
 * 100 MB, so it pays for buffers that are reused.  Purely a transfer-rate matter: every entry point takes unregistered memory as well. */
int  vgk_host_register(vgk_ctx* ctx, const void* ptr, size_t bytes);
int  vgk_host_unregister(vgk_ctx* ctx, const void* ptr);

/* gssw path.  pack = validate + encode + H2D (host threads; the copies run on a copy stream of their own, so another
 * thread may pack the next batch while this one runs); run = kernels only (asynchronous on the context's HIP stream,
 * bracketed by HIP events); fetch = wait + the CIGAR ops packed behind each other on the device + D2H.  Freed batches leave
 * their device arenas and page-locked blocks with the context for the next pack (released by vgk_destroy). */
int  vgk_gssw_pack (vgk_ctx* ctx, const vgk_gssw_problem* problems, uint32_t n,
                    uint32_t ops_per_problem /* 0 = engine default */,
                    vgk_batch** out);
int  vgk_gssw_run  (vgk_batch* batch);
int  vgk_gssw_fetch(vgk_batch* batch, vgk_result* results /* [n] */,
                    vgk_op* ops, size_t ops_cap, size_t* ops_written);
/* one-call convenience = pack + run + fetch + free, in sub-batches that fit half of HBM.  Problems the kernels cannot take
 * (VGK_ETOOLONG, VGK_EUNSUPPORTED, VGK_EINVAL) are answered in their own result's status and the rest of the call goes
 * ahead; vgk_gssw_pack itself refuses a batch that holds such a problem. */
int  vgk_gssw_align(vgk_ctx* ctx, const vgk_gssw_problem* problems, uint32_t n,
                    vgk_result* results, vgk_op* ops, size_t ops_cap,
                    size_t* ops_written);
/* The wide route's share of the last vgk_gssw_align call on this context (problems beyond the packed kernels' range: reads of more than 1024
 * rows, scores beyond 11 bits — four wavefronts per problem, int32 cells, strips through HBM): 0 = fill kernels ms, 1 = traceback kernel ms,
 * 2 = DP cells, 3 = cells whose traceback codes were stored, 4 = launches. */
double vgk_gssw_wide_last(vgk_ctx* ctx, int which);
/* ---- one graph resident in HBM, problems as windows of it (device-side packing) --------------------------------------
 * `vg map` aligns every read against a subgraph cut out of ONE graph around its seed cluster (the id range around the MEMs,
 * src/mapper.cpp:2445-2518) and converts that subgraph node by node on the CPU (create_gssw_graph, src/aligner.cpp:30-85).
 * When the graph is a DAG whose node order is topological, such a subgraph is a run of consecutive nodes: the graph is
 * handed over once (vgk_graph_create), a problem shrinks to {read, first node, node count} = the induced subgraph on
 * nodes [first_node, first_node + n_nodes) (edges that enter the window from outside are dropped, exactly as in the
 * extracted subgraph), and vgk_gssw_pack_windows derives everything the kernels read from the resident tables ON THE
 * DEVICE: the host copies two flat buffers (reads, problems) and touches no problem.  Results are those of vgk_gssw_pack on
 * the same induced subgraphs; op.node counts from the window's first node.
 * Modes: VGK_GSSW_LOCAL and VGK_XDROP_PINNED (| VGK_GSSW_TRACEBACK); plain contexts only (quality-adjusted: VGK_EUNSUPPORTED).
 * Nodes must be non-empty and at most 65535 bases long. */
typedef struct vgk_dgraph vgk_dgraph;
int  vgk_graph_create(vgk_ctx* ctx, const vgk_graph* graph, vgk_dgraph** out);
void vgk_graph_destroy(vgk_dgraph* graph);
typedef struct vgk_window_problem {
    uint64_t read_off;        /* offset of the read in `reads`                                   */
    uint32_t read_len;
    uint32_t flags;           /* VGK_GSSW_*                                                      */
    uint32_t first_node;      /* the window: nodes [first_node, first_node + n_nodes)            */
    uint32_t n_nodes;
    uint32_t max_gap_length;  /* XDROP only (see vgk_gssw_problem)                               */
    uint32_t reserved;
} vgk_window_problem;
/* A batch for vgk_gssw_run / vgk_gssw_fetch / vgk_batch_free like one from vgk_gssw_pack.  The first malformed or
 * out-of-range problem (in index order) decides the return code, as a serial scan would. */
int  vgk_gssw_pack_windows(vgk_ctx* ctx, const vgk_dgraph* graph, const char* reads, size_t reads_bytes,
                           const vgk_window_problem* problems, uint32_t n, uint32_t ops_per_problem, vgk_batch** out);
/* ---- extension windows: the passes of a seeded X-drop alignment inside a window of the resident graph ---------------------------------
 * Aligner::align_xdrop (DozeuInterface::align, src/dozeu_interface.cpp:608-685) runs two pinned extensions over ONE subgraph: from the
 * seed (a position inside a node) towards one end of the read to find the "head", then from the head the other way, traced.  Each is
 * dz_extend over the part of the subgraph that lies in its direction: dozeu is handed the start node's sequence from the offset on
 * (`seq + ref_offset`, :236-243; a leftward pass: the bases before it, backwards, :178-185), then every node its forefronts reach, in
 * topological order.  giraffe's mate rescue (MinimizerMapper::attempt_rescue, src/minimizer_mapper.cpp:3264-3440) does this for every
 * lost mate against the nodes at the fragment's distance from its partner.
 * With the graph resident such a pass is {read, window, start position, direction}: the engine derives the sub-DAG ON THE DEVICE — the
 * start node cut at the offset (left out when nothing of it lies that way: its neighbours then start from the root column), the window's
 * nodes reachable from it in that direction, in extension order (leftward: descending, sequences reversed, successors as predecessors —
 * no complementing), the read part on that side of query_offset (leftward: reversed) — and runs VGK_XDROP_PINNED over it on the same
 * kernels as every other batch.  Results are those of vgk_gssw_pack on the sub-DAG a caller would build by hand
 * (vg_amd/host/aligner.cpp xdrop_extend_prepare), except that end_node and op.node count from the WINDOW's first node (translated at
 * fetch).  A leftward pass comes back in extension order: its caller flips it as for a right pin (src/aligner.cpp:443-449).  When nothing
 * lies in the direction (a start at the window's last base, say) the result is score 0, no ops: "did not run".
 * flags: VGK_XDROP_PINNED (| VGK_GSSW_TRACEBACK).  start_node is an index of the resident graph inside the window; start_offset at most
 * the node's length; query_offset at most read_len, with a non-empty read part on the extension's side (VGK_EINVAL otherwise, as for a
 * malformed window).  Graphs from vgk_graph_create only (a forest graph: VGK_EUNSUPPORTED). */
typedef struct vgk_extension_problem {
    uint64_t read_off;        /* the whole read in `reads`                                              */
    uint32_t read_len;
    uint32_t flags;
    uint32_t first_node;      /* the window: nodes [first_node, first_node + n_nodes)                    */
    uint32_t n_nodes;
    uint32_t max_gap_length;  /* as in vgk_gssw_problem                                                  */
    uint32_t start_node;      /* the extension starts on this node of the resident graph ...             */
    uint32_t start_offset;    /* ... rightward: at its base start_offset; leftward: before it            */
    uint32_t query_offset;    /* rightward: aligns read[query_offset, read_len); leftward: read[0, query_offset) */
    uint32_t leftward;        /* 0 / 1                                                                   */
    uint32_t reserved;
} vgk_extension_problem;
int  vgk_gssw_pack_extensions(vgk_ctx* ctx, const vgk_dgraph* graph, const char* reads, size_t reads_bytes,
                              const vgk_extension_problem* problems, uint32_t n, uint32_t ops_per_problem, vgk_batch** out);
/* Which pairs of the batch the last vgk_gapless_extend / vgk_gapless_extend_seeded call on this context extended (reads 2 i and 2 i + 1 are
 * a pair) are rescue candidates, and what the rescue of each is given: a pair with a full-length extension set for exactly one mate
 * (MinimizerMapper::map_paired, src/minimizer_mapper.cpp:1793-1901, rescues the end without alignments from the other), the rescue nodes
 * [node_lo, node_hi) of the resident graph at the fragment's distance from the mapped mate (attempt_rescue, :3264-3300, takes them from
 * subgraph_in_distance_range over the SnarlDistanceIndex — an absent dependency: here the nodes whose columns lie within
 * [mean - k sd - read length, (mean + k sd) 1.1 + 40] of the mapped mate's start, downstream of a forward-mapped mate and upstream of a
 * reverse-mapped one, on a graph whose node order is topological; a stated stand-in), and dozeu's seed: the best gapless extension of the
 * lost mate inside those nodes on the strand it is rescued on, the earlier among equals (:3322-3348), as the mate reads along the forward
 * strand of the subgraph.  The sets never leave HBM for this: one lane per pair reads them where the extension kernels left them; only the
 * table comes down, in pair order.  `graph`: the resident graph whose node v is the index's node v (any context of the same device).
 * *written = the entries needed (VGK_EOPS when cap is too small). */
typedef struct vgk_rescue_request {
    uint32_t mapped, lost;           /* reads of the batch */
    uint32_t node_lo, node_hi;       /* the rescue nodes */
    int32_t  seed_begin, seed_end;   /* the seed's read interval (0, 0 without a seed) ... */
    int32_t  seed_node;              /* ... the node it starts on (-1: none) ... */
    int32_t  seed_offset;            /* ... and the offset there */
    uint32_t reverse;                /* 1: the lost mate is rescued as its reverse complement (its partner maps forward) */
    uint32_t reserved;
} vgk_rescue_request;
int  vgk_rescue_requests(vgk_ctx* ctx, const vgk_dgraph* graph, double fragment_mean, double fragment_sd, double rescue_stdevs,
                         vgk_rescue_request* requests, size_t cap, size_t* written);
/* ---- tail forests: the subgraphs giraffe aligns read tails to (MinimizerMapper::get_tail_forest, src/minimizer_mapper.cpp:5745-5860;
 * dfs_gbwt :5909-6013) ---------------------------------------------------------------------------------------------------------------
 * For an extension that does not reach an end of the read, giraffe walks the haplotypes that continue it — a depth-first search over
 * the GBWT from the extension's search state, out to (longest detectable gap + tail length) bases — and collects the nodes it enters
 * as a tree of (parent, handle) pairs: the haplotype-consistent subgraph the tail is then aligned to, pinned at the root
 * (get_best_alignment_against_any_tree :5626-5741 -> align_pinned on a TreeSubgraph).  When the cut lies at the very end of the start
 * node the root is not entered and every child of it starts a tree of its own (a forest).
 * vgk_tail_forest does that walk on the device, over the haplotype index of vgk_haplo_create, for a batch of tails; the trees stay in
 * HBM and are turned THERE into one resident graph (vgk_forest_graph) in which every tree is a run of consecutive nodes in entry
 * order — which is a topological order, and the order TreeSubgraph numbers its nodes in — so that the alignments are window problems
 * of that graph (vgk_gssw_pack_windows, VGK_XDROP_PINNED): first_node = the tree's first node, n_nodes = its size.  The root of a tree
 * carries only the bases behind the cut.  Children are entered in the order the reference's stack pops them: the node's outgoing
 * edges from the last to the first.
 * Per problem: VGK_EINVAL for a node or cut outside the index, VGK_ETOOBIG when the walk's stack outgrows the kernel's (512 frames).
 * [PARITY-UNPINNED: the reference holds no test vectors for get_tail_forest; pinned here by an independent construction from the
 * thread lists in tests/test_tail_forest.py.] */
typedef struct vgk_tail_problem {
    uint32_t node;            /* the search state the walk starts from: oriented node ...                                          */
    int32_t  lo, hi;          /* ... and its range of visits (vgk_extension.state[0..2] for a right tail, [3..5] for a left tail)   */
    uint32_t offset;          /* the cut on that node: its bases [offset, length) belong to the tail (from.offset(), :5758-5775)     */
    uint32_t walk_distance;   /* the search limit in bases (:5816)                                                                  */
} vgk_tail_problem;
typedef struct vgk_tail_result {
    int32_t  status;
    uint32_t first_node;      /* the problem's tree nodes are [first_node, first_node + n_nodes) of the forest                      */
    uint32_t n_nodes, n_trees;
    uint32_t root_trim;       /* bases cut off its root(s): `offset`, or 0 when the root was skipped (:5800, :5838)                  */
    uint32_t bases;           /* bases of its tree nodes together (the caller's max_dozeu_cells test, :5693)                        */
} vgk_tail_result;
typedef struct vgk_forest vgk_forest;
typedef struct vgk_haplo vgk_haplo;   /* the haplotype index (vgk_haplo_create, below) */
int      vgk_tail_forest(vgk_ctx* ctx, const vgk_haplo* index, const vgk_tail_problem* problems, uint32_t n,
                         vgk_tail_result* results, vgk_forest** out);
uint64_t vgk_forest_size(const vgk_forest* forest);                      /* tree nodes over all problems */
/* per tree node (each array nullable): its parent (index in the forest, -1 = the root of a tree), the oriented node of the index
 * it stands for (TreeSubgraph::translate_down), its length in the forest graph (the root's: what is left behind the cut) */
int      vgk_forest_fetch(const vgk_forest* forest, int32_t* parent, uint32_t* node, uint32_t* length);
const vgk_dgraph* vgk_forest_graph(const vgk_forest* forest);            /* owned by the forest */
void     vgk_forest_destroy(vgk_forest* forest);
double   vgk_tail_last_ms(vgk_ctx* ctx);                                 /* device time of the last vgk_tail_forest call: walks + graph construction */
/* The tails of the extension sets the last vgk_gapless_extend / vgk_gapless_extend_seeded call on this context produced, aligned, without
 * the sets, the trees or the tails' bases crossing PCIe: what MinimizerMapper does per extension at src/minimizer_mapper.cpp:5480-5535
 * (get_tail_forest for either open end of every extension of a read without a full-length extension, get_best_alignment_against_any_tree,
 * the total score) as kernels over what that call left in HBM — tails derived from the extensions' search states, one forest, one window
 * per tree, left-pinned X-drop, the best tree of a tail, the totals.  ext_total[e] = extension e's score + the best alignment of either
 * open tail (a tail nothing aligns to, or one the engine declines, adds 0: the soft clip); read_score[i] = the best total of read i.
 * stats (nullable): tails, trees, tree nodes, tails + windows the engine declined.  This entry point returns scores only (giraffe ranks
 * its extensions by them before it builds alignments); vgk_tail_stage_aligned returns the tails' alignments as well. */
int      vgk_tail_stage(vgk_ctx* ctx, const vgk_haplo* index, uint32_t ops_per_problem, int32_t* ext_total, size_t ext_cap, int32_t* read_score, uint64_t stats[4]);
/* The same, and per tail the alignment of its best tree — what get_best_alignment_against_any_tree (:5626-5741) hands back, chosen on the
 * device: only the winners' ops cross PCIe, with their nodes already translated from tree nodes to oriented nodes of the index
 * (TreeSubgraph::translate_down).  Tails come in the order right tails (by extension), then left tails.  A right tail's ops run along
 * the read; a LEFT tail's are those of the reverse-complemented tail on the other strand, running from the extension outwards — the
 * alignment the reference has before it reverse-complements it back (:5726).  score 0 / n_ops 0: the soft clip (nothing aligned better).
 * Of several trees with the best score the first one wins (the reference asks deterministic_beats, a hash of the alignments: unpinned).
 * written (nullable): tails, ops — set also on VGK_EOPS (tails_cap or ops_cap too small). */
typedef struct vgk_tail_alignment {
    uint32_t ext;             /* the extension it belongs to (index into that call's `extensions`)                                   */
    uint32_t left;            /* 0: the right tail, read bases [read_begin, read_end) behind the extension; 1: the left tail before it */
    uint32_t read_begin, read_end;
    int32_t  score;
    int32_t  status;          /* the tail's forest status (VGK_OK, VGK_ETOOBIG: walk declined -> soft clip)                           */
    uint32_t ops_begin, n_ops;    /* in `ops`                                                                                          */
    uint32_t first_offset;    /* where on the first op's node (on that strand) the alignment starts                                   */
    uint32_t n_trees;         /* trees the tail was aligned to                                                                        */
} vgk_tail_alignment;
int      vgk_tail_stage_aligned(vgk_ctx* ctx, const vgk_haplo* index, uint32_t ops_per_problem, int32_t* ext_total, size_t ext_cap, int32_t* read_score,
                                vgk_tail_alignment* tails, size_t tails_cap, vgk_op* ops, size_t ops_cap, size_t written[2], uint64_t stats[4]);
double   vgk_tail_stage_last_ms(vgk_ctx* ctx, int which);             /* 0 tails derived, 1 forest, 2 windows packed, 3 kernels + totals */

/* ---- X-drop with dozeu's band (src/dozeu_interface.cpp:226, :261-283; src/xdrop_aligner.cpp:95-109) ------------------------------
 * VGK_XDROP_PINNED through vgk_gssw_* keeps every cell: it returns the exact semi-global optimum, which is dozeu's answer whenever
 * dozeu's band contains the optimal path.  This entry point restates the band itself [PARITY-UNPINNED: dozeu's source is not in the
 * reference snapshot; the rules below are this engine's reading of the published algorithm, stated identically in
 * oracle/vgo_xdrop.c]: rows are grouped in dozeu's 8-cell vectors; a column keeps the vectors from the first to the last one that
 * holds a cell within xt = (gap_open - gap_extend) + gap_extend * max_gap_length of the best score reached so far on the way to
 * that column (the maximum over the predecessors' fronts, updated after every column); cells outside are unreachable from then on;
 * a node whose incoming fronts are all empty is skipped.  Everything else — root column, seeds, bonus, end cell, traceback
 * preferences — is as in VGK_XDROP_PINNED.  Problems must be VGK_XDROP_PINNED (| VGK_GSSW_TRACEBACK); reads up to 511 bases.
 * stats (nullable): [0] cells inside the bands, [1] cells of the full read x graph rectangles.
 * The device fills the columns (one 8-row vector per lane; tails of up to 127 bases four to a wavefront, 16 lanes each — with VGAMD_XBAND_EIGHTS=1
 * those of up to 63 bases eight to a wavefront, 8 lanes each: measured slower, off by default —, longer reads a wavefront each; only a column's front is kept in HBM) and picks the end cell; a second kernel, one lane per problem, walks the
 * tracebacks; ops are packed on the device and only results and ops come back.  VGAMD_XBAND_TIMING=1 prints the call's host laps. */
int  vgk_xdrop_band_align(vgk_ctx* ctx, const vgk_gssw_problem* problems, uint32_t n,
                          vgk_result* results, vgk_op* ops, size_t ops_cap, size_t* ops_written, uint64_t stats[2]);
double vgk_xdrop_band_last_ms(vgk_ctx* ctx);     /* kernel time (fills + tracebacks) of the last vgk_xdrop_band_align call on this context */
/* How that call's fill kept its cells: 4 = int32 cells and arithmetic; 2 = 16-bit cells, two rows to a register (taken when
 * (read + graph + 10) x (largest |score| + gap_open + gap_extend + bonus) < 16 000 for every problem of the call and the scores fit bytes:
 * no reachable cell can then leave the 16-bit range); 3 = 16-bit cells under int32 arithmetic (VGAMD_XBAND_ARITH32=1; VGAMD_XBAND_CELLS32=1
 * forces 4).  The answers are the same in every form (tests/test_xdrop_band.py); what differs is the bytes a cell costs: 2 x 4 or 2 x 2. */
int    vgk_xdrop_band_last_cells(vgk_ctx* ctx);
/* ... and how its problems shared wavefronts (summed over the call's sub-batches): which = 0: tails of at most 63 bases, EIGHT to a wavefront
 * (8 lanes of 8 rows each; the packed fill only); 1: tails of at most 127 bases, four to a wavefront; 2: the rest, a wavefront each. */
uint64_t vgk_xdrop_band_last_class(vgk_ctx* ctx, int which);

/* k-best pinned alignments (Aligner::align_pinned_multi -> gssw_graph_trace_back_pinned_multi, src/aligner.cpp:423-435, :455-480).
 * Every problem must be VGK_GSSW_PINNED.  results[i * max_alt_alns + k] is the k-th best alignment of problem i (k <
 * n_alignments[i] <= max_alt_alns), scores non-increasing and > 0, the first one the alignment vgk_gssw_align returns; a problem
 * that failed has n_alignments[i] = 0 and its status in results[i * max_alt_alns].  The device fills and keeps the H / E / F
 * matrices and enumerates the alternates over them, one lane per problem (deflections from earlier tracebacks, best first — gssw's
 * own rules are not in the reference snapshot: DESIGN.md §13); only the alignments come back.  A problem the kernel's fixed tables
 * cannot hold (a node with more than 15 predecessors, an alternate with more than 24 deflections, max_alt_alns > 62) is walked by a
 * host thread over its own matrices under the same rules; vgk_gssw_multi_host_walks counts those of the last k-best call (this one or
 * vgk_banded_align_multi). */
int  vgk_gssw_align_multi(vgk_ctx* ctx, const vgk_gssw_problem* problems, uint32_t n, uint32_t max_alt_alns,
                          vgk_result* results /* [n * max_alt_alns] */, uint32_t* n_alignments /* [n] */,
                          vgk_op* ops, size_t ops_cap, size_t* ops_written);
uint64_t vgk_gssw_multi_host_walks(const vgk_ctx* ctx);

/* ---- banded global alignment (BandedGlobalAligner, src/banded_global_aligner.cpp) -------------------
 * Replaces, inside Aligner::align_global_banded / QualAdjAligner::align_global_banded (src/aligner.cpp:699-760,
 * :1189-1248), the construction of BandedGlobalAligner<IntType> (:1961-2110: band geometry, masking, cell budget)
 * and its align() (:2296-2326: fill of the three band matrices per node, choice of the end cell, traceback).
 * The primary alignment only (max_multi_alns == 1).  The graph is handed over in the order
 * handlealgs::lazier_topological_order gives (:1976); predecessors in follow_edges(node, true) order (:2032-2036),
 * which the traceback's tie rules depend on.  Nodes may be empty (node_len 0).  The read must not be empty —
 * vg routes empty reads to DeletionAligner before it gets here (src/aligner.cpp:703-706).
 * Result per problem: score, status (VGK_OK, VGK_ENOBAND = NoAlignmentInBandException, VGK_ETOOBIG =
 * BandMatricesTooBigException), and one (node, op, len) run per edit from the first node of the path to the last;
 * an empty node on the path contributes one op of length 0.  Every alignment starts at offset 0 of its first node
 * and ends at the end of its last (global), so end_node / end_offset / first_offset are left 0. */
#define VGK_BANDED_PERMISSIVE 1u      /* permissive_banding (:2196-2208)                                  */
typedef struct vgk_banded_problem {
    const char*    read;              /* alignment.sequence()                                            */
    const uint8_t* qual;              /* alignment.quality(), raw phred bytes; NULL unless the context is quality-adjusted */
    uint32_t       read_len;
    uint32_t       flags;             /* VGK_BANDED_PERMISSIVE                                            */
    vgk_graph      graph;
    int32_t        band_padding;      /* (:2203-2218)                                                     */
    uint32_t       reserved;
    uint64_t       max_cells;         /* cell budget (:1999-2015); 0 = unlimited                          */
} vgk_banded_problem;

int  vgk_banded_align(vgk_ctx* ctx, const vgk_banded_problem* problems, uint32_t n,
                      vgk_result* results, vgk_op* ops, size_t ops_cap, size_t* ops_written);
/* The k best alignments of every problem (Aligner::align_global_banded_multi, src/aligner.cpp:763-831: the constructor with
 * max_multi_alns and the AltTracebackStack of src/banded_global_aligner.cpp:2426-2790): results[i * max_alt_alns + k] is the
 * k-th best alignment of problem i, k < n_alignments[i], scores in descending order; a problem that fails has n_alignments 0 and
 * its status in results[i * max_alt_alns].  The fill runs on the device and so does the enumeration of the alternates (one lane per
 * problem over the matrices the fill keeps; only the alignments come back).  What that walk declines — a problem with a chain of empty
 * nodes from source to sink, a traceback with more than 24 deflections, max_alt_alns > 63 — a host thread walks over the problem's
 * matrices under the same rules; vgk_gssw_multi_host_walks counts those of the last k-best call (pinned or banded). */
int  vgk_banded_align_multi(vgk_ctx* ctx, const vgk_banded_problem* problems, uint32_t n, uint32_t max_alt_alns,
                            vgk_result* results, uint32_t* n_alignments, vgk_op* ops, size_t ops_cap, size_t* ops_written);
/* timing of the last vgk_banded_align call on this context: 0 = fill kernel ms, 1 = traceback kernel ms,
 * 2 = band cells filled, 3 = algorithmic bytes */
double vgk_banded_last(vgk_ctx* ctx, int which);
/* Launch the kernels of the last vgk_banded_align call again on its inputs, which stay resident in HBM (the counterpart of
 * vgk_gssw_run for this path; results stay on the device).  VGK_EINVAL unless that call fitted one sub-batch — a call of 32 768 problems or
 * more runs as four sub-batches, two in flight (the next one's geometry and arenas made while one runs), and leaves nothing resident:
 * VGAMD_BANDED_ONE_BATCH=1 in the environment keeps such a call in one piece. */
int    vgk_banded_rerun(vgk_ctx* ctx);

/* ---- haplotype-consistent gapless extension (GaplessExtender, src/gbwt_extender.cpp:533-737) ---------
 * Replaces GaplessExtender::extend(cluster, sequence, cache, max_mismatches, overlap_threshold, trim)
 * (src/gbwt_extender.hpp:205): for every seed of a cluster the best gapless extension along the indexed haplotypes
 * (best-first over `follow_paths`, mismatch limits :605-607 / :649-651), then either the non-overlapping full-length
 * extensions or the trimmed, de-duplicated partial ones.
 * The haplotype index stands in for the GBWTGraph the reference walks (gbwt / gbwtgraph are absent submodules): node
 * sequences plus the threads as lists of oriented nodes (2 * node index + is_reverse).  Node indices must follow the
 * order of the graph's node ids (the order of `follow_paths` is the order of the GBWT node encoding).  Threads may
 * revisit nodes (cycles). */
typedef struct vgk_haplotypes {
    uint32_t        n_nodes;
    const uint32_t* node_len;
    const char*     seq;             /* forward strands of all nodes, concatenated */
    uint32_t        n_threads;
    const uint32_t* thread_off;      /* n_threads + 1 offsets into thread_nodes */
    const uint32_t* thread_nodes;    /* oriented nodes along each thread */
} vgk_haplotypes;
typedef struct vgk_haplo vgk_haplo;  /* the index, resident in HBM */
/* Sharing an index: a vgk_haplo and a vgk_minimizer_index are read-only tables in the HBM of the device their context drives (complete
 * when the create call returns; vgk_minimizer_set_policy aside).  Every context ON THE SAME DEVICE may be handed them — vg calls one
 * aligner from many threads; here a caller that keeps several batches in flight opens a context per batch (its own streams, scratch and
 * stage state) over ONE copy of the indexes: 8 ranks x 2 contexts need 8 uploads, not 16.  The creating context must be destroyed last.
 * (A context of another device: VGK_EINVAL.) */
int  vgk_haplo_create(vgk_ctx* ctx, const vgk_haplotypes* haplotypes, vgk_haplo** out);
void vgk_haplo_destroy(vgk_haplo* index);
/* The same index from the image of a GBWT file (what the reference loads with gbwt_helper's load_gbwt and reaches through
 * gbwtgraph::GBWTGraph): simple-sds serialization (header flag 0x4 — what current `vg gbwt` writes; SDSL-serialized and unidirectional
 * files: VGK_EUNSUPPORTED), bidirectional.  The records are taken over as they lie in the file (edge lists and runs decoded once per
 * record, in parallel; nothing is walked out with LF); GBWT
 * node (offset + 1) + o becomes oriented node o, so n_nodes must be (alphabet_size - offset - 1) / 2 and node_len / seq describe those
 * nodes in id order.  Malformed or truncated image: VGK_EINVAL.  [gbwt is an absent submodule: format as published, pinned on the
 * reference's test/primers/y.gbwt — tests/test_gbwt_file.py.] */
int  vgk_haplo_create_gbwt(vgk_ctx* ctx, const void* gbwt, size_t bytes, uint32_t n_nodes, const uint32_t* node_len, const char* seq, vgk_haplo** out);
/* A GBZ image (gbwtgraph's container of a GBWT and the graph's node sequences — what `vg giraffe -Z` loads; simple-sds serialization)
 * decoded into the plain form vgk_haplo_create and vgk_minimizer_index_create take: nodes in id order with their forward sequences,
 * the even GBWT sequences as threads.  Needs no context (host work only); the result owns its arrays: vgk_haplotypes_free.
 * [Pinned on the reference's test/primers/y.giraffe.gbz: sequences = those of y.gg, threads = those of y.gbwt — tests/test_gbwt_file.py.] */
int  vgk_gbz_load(const void* gbz, size_t bytes, vgk_haplotypes** out);
void vgk_haplotypes_free(vgk_haplotypes* haplotypes);

typedef struct vgk_seed {            /* GaplessExtender::seed_type (src/gbwt_extender.hpp:33): (handle, read_offset - node_offset) */
    uint32_t node;                   /* oriented node */
    int32_t  diff;
} vgk_seed;
#define VGK_GAPLESS_TRIM 1u          /* extend(..., trim = true) */
#define VGK_GAPLESS_DEFER 2u         /* vgk_gapless_extend_seeded (in `flags`) and vgk_gapless_extend (in the flags of problems[0]: it is a
                                        property of the call): return as soon as the sets are laid out on the device and their sizes
                                        (`written`) are known; the copies into results / extensions / nodes / mismatches run on a side stream and
                                        are complete when the NEXT vgk_tail_stage / vgk_tail_stage_aligned call on the context returns — they
                                        travel while that call keeps the device busy — or after vgk_gapless_fetch_deferred.  The arrays must
                                        stay valid until then.  (Arrays too small: not deferred, VGK_EOPS as usual.) */
typedef struct vgk_gapless_problem {
    const char*     read;            /* masked by the engine like ReadMasker (:160-176): non-ACGT never matches */
    uint32_t        read_len;
    uint32_t        n_seeds;
    const vgk_seed* seeds;           /* visited in this order (the reference iterates a hash set: order unpinned) */
    uint32_t        max_mismatches;  /* GaplessExtender::MAX_MISMATCHES = 4 */
    uint32_t        flags;
    double          overlap_threshold;   /* GaplessExtender::OVERLAP_THRESHOLD = 0.8 */
} vgk_gapless_problem;
typedef struct vgk_extension {       /* GaplessExtension (src/gbwt_extender.hpp:30-90) */
    uint32_t path_begin, path_len;   /* oriented nodes, in the `nodes` output array */
    uint32_t offset;                 /* in the first node */
    uint32_t read_begin, read_end;   /* read_interval */
    uint32_t mism_begin, n_mismatches;   /* mismatch_positions, in the `mismatches` output array */
    int32_t  score;
    uint8_t  left_full, right_full, pad[2];
    uint32_t state[6];               /* forward node, range first, last; backward node, range first, last */
} vgk_extension;
typedef struct vgk_gapless_result {
    int32_t  status;
    uint32_t ext_begin, n_ext;       /* in the `extensions` output array */
    uint32_t full_length;            /* GaplessExtender::full_length_extensions() of the set */
} vgk_gapless_result;
int  vgk_gapless_extend(vgk_ctx* ctx, const vgk_haplo* index, const vgk_gapless_problem* problems, uint32_t n,
                        vgk_gapless_result* results, vgk_extension* extensions, size_t ext_cap,
                        uint32_t* nodes, size_t nodes_cap, uint32_t* mismatches, size_t mism_cap,
                        size_t written[3] /* extensions, nodes, mismatches */);
int    vgk_gapless_fetch_deferred(vgk_ctx* ctx);   /* wait for and finish the copies of a VGK_GAPLESS_DEFER call (no-op when none is pending) */
int    vgk_gapless_rerun(vgk_ctx* ctx);      /* launch the kernel of the last vgk_gapless_extend call again on its resident inputs */
double vgk_gapless_last_ms(vgk_ctx* ctx);    /* kernel time of the last vgk_gapless_extend call on this context */
/* The search walks an index in which UNARY RUNS are merged at build time: consecutive nodes v, v + 1, ... where every haplotype that visits
 * one goes on to the next and every visit of the next arrives that way (so a search state's ranges map through unchanged), at most 255 bases
 * together — what the reference does per problem for WFA (WFANode: unary paths up to 1 024 bp, src/gbwt_extender.cpp:1431-1487), done once
 * per index: a 150-base read over 32-base nodes crosses one or two records per direction instead of five to ten dependent hops.  Seeds come
 * in and sets go out in the ORIGINAL nodes.  A search extends every partial extension until nothing is left and keeps the best finished one,
 * the first among equals: the one thing that depends on the order of the steps, and so on their granularity.  A search whose best score two
 * finished extensions share is therefore run again on the original index — the results are those of the node-by-node search, tie for tie.
 * Measured: it does not pay yet (the search kernel is bound by the bytes it moves, not by its hops: DESIGN.md §28.3), so an index is built with
 * merged runs only under VGAMD_HAPLO_MERGE=1.  vgk_haplo_search_nodes: nodes of the index the search walks (= the graph's when nothing merged);
 * vgk_gapless_last_redone: seeds of the last call whose search ran twice. */
uint64_t vgk_haplo_search_nodes(const vgk_haplo* index);
/* The merged-run form of the index is built with it always (VGAMD_HAPLO_NO_MERGE=1: not at all): the WFA wavefront kernel walks it — a trie node
 * there is a non-branching path, and walking one is a single lane's chain of record fetches, so hops are its time; positions are taken onto
 * the runs at a problem's start, paths come back in the original nodes, and the trie is the node-by-node walk's (a run is cut where WFANode's
 * walk would end a node inside it: at the target, at 1 024 bases) — results identical, tie for tie.  vgk_haplo_run_nodes: nodes of that form
 * (= the graph's when no run could be merged). */
uint64_t vgk_haplo_run_nodes(const vgk_haplo* index);
uint64_t vgk_gapless_last_redone(vgk_ctx* ctx);
uint64_t vgk_gapless_last_retried(vgk_ctx* ctx);   /* reads of that call whose search outgrew the fast (in-LDS) kernel and ran in the slab kernel */

/* ---- minimizer seeding (MinimizerMapper::find_minimizers / find_seeds over gbwtgraph::MinimizerIndex, src/minimizer_mapper.cpp:3918-3965,
 * :4109-4290): the step that produces the clusters vgk_gapless_extend takes ------------------------------------------------------
 * vgk_minimizer_index_create indexes the (k, w)-minimizers of every haplotype thread (k <= 31, w <= 64; giraffe's defaults 29, 11 for short reads and 31, 50 for long ones)
 * with the graph positions they start at; vgk_minimizer_seeds finds the minimizers of a batch of reads on the device, looks each up
 * and turns every hit into a seed (oriented node, read offset - node offset) on the strand the read reads forward on.  Per read the
 * seeds come in the order of their minimizers' read offsets; a (node, diagonal) pair hit twice is reported once (a cluster is a set);
 * minimizers with more than `hit_cap` hits give no seeds (hard_hit_cap); at most 64 seeds per read (what a cluster of the extension
 * stage holds).  [gbwtgraph is not in the reference snapshot; vg_amd/csrc/minimizer_device.hpp states the scheme
 * restated here — 2-bit keys, Wang's 64-bit hash, the smaller-hash orientation canonical, leftmost minimum per window.  It is PINNED on the
 * one MinimizerIndex the reference keeps (test/primers/y.min, k = 31, w = 50: all 62 keys and positions reproduced from y.gg + y.gbwt,
 * tests/test_minimizer.py); what a read's minimizers turn into below that — seed orientation, de-duplication — is PARITY-UNPINNED.  Seed
 * SCORING and the downsampling / hit-cap policies of find_seeds (:4140-4290), and the clustering of seeds by graph distance
 * (SnarlDistanceIndexClusterer), are the caller's.] */
typedef struct vgk_minimizer_index vgk_minimizer_index;
int  vgk_minimizer_index_create(vgk_ctx* ctx, const vgk_haplotypes* haplotypes, uint32_t k, uint32_t w, vgk_minimizer_index** out);
void vgk_minimizer_index_destroy(vgk_minimizer_index* index);
uint64_t vgk_minimizer_index_keys(const vgk_minimizer_index* index);      /* distinct minimizer k-mers */
/* What the index holds, as gbwtgraph's MinimizerIndex would hold it: (canonical key, oriented node, offset on that strand) for every
 * indexed occurrence, sorted by key, then node, then offset.  vgk_minimizer_index_fetch: VGK_EOPS when cap < vgk_minimizer_index_hits. */
typedef struct vgk_minimizer_hit { uint64_t key; uint32_t node, offset; } vgk_minimizer_hit;
uint64_t vgk_minimizer_index_hits(const vgk_minimizer_index* index);
int  vgk_minimizer_index_fetch(const vgk_minimizer_index* index, vgk_minimizer_hit* hits, size_t cap);
/* reads: flat, read i = reads[read_off[i], read_off[i + 1]).  seed_off[n + 1] and, nullable, minimizers[n] (minimizers per read; with
 * VGK_MINIMIZERS_TRUNCATED or'ed in when the read reached the cap of 64 seeds with hits of its minimizers left unexamined — the
 * reference's find_seeds has no such cap, so a caller that sees the flag takes that read through its own path) are filled always; seeds up to seeds_cap (VGK_EOPS when that is too small; *written = the number needed); seeds = NULL with seeds_cap = 0
 * leaves them on the device only (for vgk_gapless_extend_seeded). */
#define VGK_MINIMIZERS_TRUNCATED 0x80000000u
int  vgk_minimizer_seeds(vgk_ctx* ctx, const vgk_minimizer_index* index, const vgk_haplo* graph, const char* reads, const uint64_t* read_off, uint32_t n,
                         uint32_t hit_cap, uint32_t* seed_off, uint32_t* minimizers, vgk_seed* seeds, size_t seeds_cap, size_t* written);
double vgk_minimizer_last_ms(vgk_ctx* ctx);                              /* device time of the last vgk_minimizer_seeds call */
/* find_seeds' choice of minimizers, applied by vgk_minimizer_seeds on the device (src/minimizer_mapper.cpp:4109-4440 with giraffe's
 * short-read parameters): a minimizer's score is 1 + ln(hard_hit_cap) - ln(hits) (1 beyond the hard cap, 0 without hits, :3927-3937);
 * a read's minimizers are taken in order of descending score, runs of one key together, equal scores by key, and the runs that share the
 * BEST score shuffled as the reference shuffles them (sort_shuffling_ties, src/utility.hpp:771-799: Knuth's shuffle over std::minstd_rand
 * seeded from the read's sequence, src/utility.cpp:911-927 — the single-end rule, :620-627; the paired path's one generator over both mates,
 * :1529-1541, is not restated); a minimizer gives seeds iff it has hits, its
 * RUN (all occurrences of its key in the read) has at most hard_hit_cap hits, and it has at most hit_cap hits or the scores selected so
 * far plus its own stay within minimizer_score_fraction of the read's total or an earlier occurrence of its key was taken; the first
 * minimizer that fails the last test closes it for everything that follows (:4358-4378).  The filters left out are off in those
 * parameters (window downsampling, exclude-overlapping) or cannot fire below 500 taken minimizers (max-unique-min); the host shim's
 * select_minimizers (vg_amd/host/seed_policy.cpp) has them all.  A read with more than 64 minimizers — or one with a base other than A, C,
 * G, T whose shuffle could change the choice: the engine keeps reads masked and cannot seed the generator from its bytes — is seeded as
 * without a policy and flagged VGK_MINIMIZERS_POLICY_SKIPPED in minimizers[].  With a policy set the call's `hit_cap` argument is ignored.
 * policy = NULL: none (every minimizer with at most `hit_cap` hits gives seeds).  VGK_EINVAL: hard_hit_cap 0 or above 65 535, a fraction
 * outside [0, 1]. */
/* paired != 0: reads 2 i and 2 i + 1 are the mates of a pair.  The reference's paired path draws both mates' shuffles from ONE generator seeded
 * from mate 1's + mate 2's sequence (src/minimizer_mapper.cpp:1529-1541), which the per-read kernels do not restate: a read whose top tie can change
 * the choice is then not chosen for but flagged VGK_MINIMIZERS_POLICY_SKIPPED, and the caller takes its PAIR through its own choice (the host shim:
 * vgh_select_minimizers_of_pair).  Reads whose tie cannot matter — nearly all — are chosen for as before: their choice does not depend on any draw. */
typedef struct vgk_seed_policy { uint32_t hit_cap, hard_hit_cap; double minimizer_score_fraction; uint32_t paired, reserved; } vgk_seed_policy;
#define VGK_MINIMIZERS_POLICY_SKIPPED 0x40000000u
int  vgk_minimizer_set_policy(vgk_minimizer_index* index, const vgk_seed_policy* policy);
/* ---- seeding reads of ANY length (giraffe's long-read path: a 15 kbp read has ~2 500 minimizers and as many seeds as they have hits) ----------
 * vgk_minimizer_seeds above is the short-read form: at most 64 minimizers go through the device's policy, at most 64 seeds make a cluster.  These
 * two calls have no such caps and leave find_seeds' choice (src/minimizer_mapper.cpp:4109-4440, every filter: hit caps, score fraction,
 * max_unique_min / num_bp_per_min :4162,4312-4320, window downsampling, exclude-overlapping) to the caller between them — the host shim's
 * select_minimizers (vg_amd/host/seed_policy.cpp) restates it for any number of minimizers:
 *   vgk_minimizer_list      every minimizer of every read, in read order: key, offset of the k-mer's first base in the read, hits in the index,
 *                           orientation (minimizer_regions + find, :3918-3965); minimizer_off[n + 1], VGK_EOPS / *written as usual;
 *   vgk_minimizer_seeds_of  the seeds of the minimizers the caller TAKES (take[j] != 0), one per hit in index order (key, node, offset), nothing
 *                           de-duplicated (:4290-4340: one Seed per hit): seed_off[n_minimizers + 1] = where minimizer j's seeds start.
 * A seed is (oriented node, read offset - node offset) on the strand the read reads forward on, as above. */
typedef struct vgk_read_minimizer { uint64_t key; uint32_t offset, hits, flags, reserved; } vgk_read_minimizer;
#define VGK_MINIMIZER_REVERSE 1u       /* flags: the canonical k-mer is the reverse complement of the read's */
int  vgk_minimizer_list(vgk_ctx* ctx, const vgk_minimizer_index* index, const char* reads, const uint64_t* read_off, uint32_t n,
                        uint64_t* minimizer_off, vgk_read_minimizer* minimizers, size_t cap, size_t* written);
int  vgk_minimizer_seeds_of(vgk_ctx* ctx, const vgk_minimizer_index* index, const vgk_read_minimizer* minimizers, const uint8_t* take, size_t n_minimizers,
                            uint64_t* seed_off, vgk_seed* seeds, size_t cap, size_t* written);
/* The clusters of the last vgk_minimizer_seeds call on this context, extended as vgk_gapless_extend would extend them — without the
 * reads or the seeds crossing PCIe again: they are still in HBM (reads masked and padded as the extension kernels want them), and the
 * problem descriptors and the hand-out order are made there.  `index` must be the haplotype index that call was given; one
 * max_mismatches / overlap_threshold / flags (VGK_GAPLESS_TRIM) for all reads.  Outputs as vgk_gapless_extend. */
int  vgk_gapless_extend_seeded(vgk_ctx* ctx, const vgk_haplo* index, uint32_t max_mismatches, double overlap_threshold, uint32_t flags,
                               vgk_gapless_result* results, vgk_extension* extensions, size_t ext_cap,
                               uint32_t* nodes, size_t nodes_cap, uint32_t* mismatches, size_t mism_cap, size_t written[3]);

/* ---- haplotype-consistent wavefront alignment (WFAExtender, src/gbwt_extender.cpp:2052-2263) ------------
 * Replaces WFAExtender::connect(sequence, from, to), ::suffix(sequence, from) and ::prefix(sequence, to)
 * (src/gbwt_extender.hpp:427-455): gap-affine WFA over the trie of haplotypes that leave `from` (WFATree :1567-2046),
 * bounded by the error model's score cap (:1631-1634).  `from` and `to` are exclusive: the alignment starts one base
 * after `from` and ends one base before `to`.  Positions are (oriented node = 2 * node index + is_reverse, offset on
 * that strand).  suffix = connect without a target, keeping the best partial alignment when the cap is hit, plus the
 * full-length bonus when the last edit is a match/mismatch and the whole sequence is aligned (:2240-2243); prefix = the
 * same on the other strand, flipped back (:2248-2263).
 * Per problem: status (VGK_OK also for "no alignment": ok = 0, like WFAAlignment::ok; VGK_ETOOBIG when the haplotype
 * trie or the wavefronts outgrow the kernel's per-problem tables (`score` then names the table: 1 wavefront points,
 * 2 trie nodes, 3 path pool, 4 edit runs, 5 node length); VGK_EOPS when `paths` / `edits` are full; VGK_ENOBAND when the
 * winning candidate lies behind the distance band, where next() (:1761-1776) records it without storing its wavefront
 * point and the reference's backtrace (:2156-2202) does not terminate).
 * One of several equally good partial alignments is kept by WFATree::trim in hash-map order in the reference; here the
 * one with the smallest (trie node, penalty, diagonal). */
typedef struct vgk_wfa_event { double per_base; int32_t min, max; } vgk_wfa_event;   /* WFAExtender::ErrorModel::Event (gbwt_extender.hpp:361-374) */
typedef struct vgk_wfa_error_model { vgk_wfa_event mismatches, gaps, gap_length, distance; } vgk_wfa_error_model;
enum { VGK_WFA_CONNECT = 0, VGK_WFA_SUFFIX = 1, VGK_WFA_PREFIX = 2 };
enum { VGK_WFA_MATCH = 0, VGK_WFA_MISMATCH = 1, VGK_WFA_INSERTION = 2, VGK_WFA_DELETION = 3 };   /* WFAAlignment::Edit (gbwt_extender.hpp:234) */
#define VGK_WFA_NO_NODE 0xffffffffu
typedef struct vgk_wfa_problem {
    const char* seq;                 /* masked by the engine like ReadMasker (:160-170): non-ACGT never matches */
    uint32_t    seq_len;
    uint32_t    mode;                /* VGK_WFA_CONNECT / SUFFIX / PREFIX */
    uint32_t    from_node, from_offset;   /* unused by PREFIX */
    uint32_t    to_node, to_offset;       /* unused by SUFFIX */
} vgk_wfa_problem;
typedef struct vgk_wfa_result {      /* WFAAlignment (src/gbwt_extender.hpp:233-270) */
    int32_t  status;
    int32_t  ok;
    int32_t  score;
    uint32_t node_offset;            /* in the first node of the path */
    uint32_t seq_offset, length;     /* aligned interval of the sequence */
    uint32_t path_begin, path_len;   /* oriented nodes, in the `paths` output array */
    uint32_t edit_begin, n_edits;    /* in the `edits` output array: length << 2 | VGK_WFA_* ; runs of one kind are merged */
} vgk_wfa_result;
/* paths = edits = NULL with path_cap = edit_cap = 0: scores only — results[] without paths and edit runs (path_len = n_edits = 0), nothing else comes down */
int  vgk_wfa_extend(vgk_ctx* ctx, const vgk_haplo* index, const vgk_wfa_error_model* model /* NULL = WFAExtender::default_error_model */,
                    const vgk_wfa_problem* problems, uint32_t n, vgk_wfa_result* results,
                    uint32_t* paths, size_t path_cap, uint32_t* edits, size_t edit_cap, size_t written[2] /* paths, edits */);
int    vgk_wfa_rerun(vgk_ctx* ctx);          /* launch the kernel of the last vgk_wfa_extend call again on its resident inputs */
double vgk_wfa_last_ms(vgk_ctx* ctx);        /* kernel time of the last vgk_wfa_extend call on this context */
double vgk_wfa_last_wave(vgk_ctx* ctx, int which);   /* of that call (wavefront form): 0 = ms of the launch, 2 = problems that outgrew the small tables (LDS) and were run again with the large ones (HBM) */
/* A launch lasts as long as its slowest problem, and a problem that is going to outgrow the kernel's wavefront table (1024 stored
 * points) — a connect across a structural variant, say — is the slowest by far.  The budget makes the kernel give such a problem up
 * early: beyond `points` stored wavefront points it is reported VGK_ETOOBIG (score = 1) exactly as when the table is full, and the
 * caller takes the route it takes for every declined problem (BandedGlobalAligner between the anchors, as vg does
 * when WFAExtender fails).  0 = the table's size (the default).  Results of the problems that stay within the budget do not change. */
int    vgk_wfa_set_point_budget(vgk_ctx* ctx, uint32_t points);
/* The same with one budget for connects — which have that fallback — and one for prefixes / suffixes, which a caller wants answered
 * (vgk_wfa_set_point_budget(p) = vgk_wfa_set_point_budgets(p, p)). */
int    vgk_wfa_set_point_budgets(vgk_ctx* ctx, uint32_t connect_points, uint32_t tail_points);
/* Three kernels answer vgk_wfa_extend with the same results (every tie included): one THREAD per problem (64 problems per instruction: the
 * easy majority's best), one WAVEFRONT per problem with the lanes as the (diagonal, haplotype) items of a penalty step and tables sized for
 * links with long gaps (a heavy problem's critical path is ~100 x shorter), and HYBRID (the default): the thread kernel hands what reaches
 * 16 stored points (about the median problem's size) to the wavefront kernel.  A stage whose problems are long and often heavy (giraffe's long-read links) does better with
 * WAVE throughout: the heavy problems then start at once instead of behind the thread launch. */
enum { VGK_WFA_FORM_HYBRID = 0, VGK_WFA_FORM_THREAD = 1, VGK_WFA_FORM_WAVE = 2 };
int    vgk_wfa_set_form(vgk_ctx* ctx, int form);
int    vgk_wfa_get_form(vgk_ctx* ctx);          /* the form in force (a stage that changes it puts it back) */
/* What the caller expects problem i of the NEXT vgk_wfa_extend call (of exactly n problems) to cost beyond its sequence length, in
 * bases: problems are handed to the kernels most expensive first, and a launch ends with its heaviest problem — one that starts last
 * adds its whole run to the launch.  giraffe knows the graph distance between the two anchors of a connect: a sequence 40 bases longer
 * than that holds a 40-base insertion and will fill the wavefront tables whatever its length.  A hint changes the order only, never a
 * result; it is used once. */
int    vgk_wfa_set_cost_hints(vgk_ctx* ctx, const uint32_t* extra_bases, uint32_t n);

/* ---- one Path per read out of a chain's pieces ------------------------------------------------------------------------------------------
 * vgk_chain_stitch replaces the composition inside MinimizerMapper::find_chain_alignment (reference
 * src/minimizer_mapper_from_chains.cpp): every piece of a read's chain — left tail, anchor, link, anchor, ..., right tail — becomes a
 * Path (WFAAlignment::to_path, src/gbwt_extender.cpp:954-1070; align_sequence_between's Alignment::path as it is), the Paths are appended
 * in read order (append_path :2606 / :2662 / :2892 / :3035 / :3103 / :3147 / :3262; src/path.cpp:284-287) and the whole is simplified
 * (`simplify(composed_path, false)` :3295; src/path.cpp:1314-1497 with Mapping simplify :1509-1563 and concat_mappings :1499-1507):
 * edits of one kind merged, insertions pushed onto the previous mapping, mappings that continue each other on one node joined, leading and
 * trailing deletions removed.  For a whole batch of reads at once, on the device: the WFA results of the context's last vgk_wfa_extend
 * call never leave HBM as paths and edit runs — only the composed alignments come back.
 * A piece is one of
 *   LINK       result `link` of the LAST vgk_wfa_extend call on this context (with the same index), which must be ok; unused fields 0;
 *   ALIGNMENT  a WFAAlignment the caller states: node path nodes[path_begin .. +path_len) (oriented nodes), node_offset in its first node,
 *              edit runs edits[edit_begin .. +n_edits) as length << 2 | VGK_WFA_*  (an anchor = one match run, to_wfa_alignment :4083-4104;
 *              path_len = 0 with one insertion run = WFAAlignment::make_unlocalized_insertion: a mapping without a position);
 *   PATH       a Path the caller states: mappings[path_begin .. +path_len), each with its position and its own edit runs (what
 *              align_sequence_between_consistently answered for a link that WFA declined).
 * A zero-length run is not an edit (to_path refuses it; a Path's empty edits are skipped as simplify skips them).
 * Results, dense and in read order: per read {status, mapping_begin, n_mappings, edit_begin, n_edits, from_length, to_length} — status
 * VGK_EINVAL when a LINK piece names a problem that is not ok or a piece walks off its node path (the reference throws), VGK_EOPS when
 * out_mappings / out_edits are too small (*written = what is needed) —; per mapping {node (oriented; VGK_WFA_NO_NODE: no position),
 * offset, edit_begin, n_edits}; per edit length << 2 | VGK_WFA_* (a mismatch run of several bases is ONE edit when its source — a PATH piece
 * of BandedGlobalAligner's — or simplify's merging made it one).  Mapping ranks are their indices + 1; substituted / inserted bases are the
 * read's own at the edit's read offset, so they are not repeated here. */
enum { VGK_PIECE_LINK = 0, VGK_PIECE_ALIGNMENT = 1, VGK_PIECE_PATH = 2 };
typedef struct vgk_chain_piece {
    uint32_t kind;
    uint32_t link;                    /* LINK */
    uint32_t node_offset;             /* ALIGNMENT */
    uint32_t path_begin, path_len;    /* ALIGNMENT: in `nodes`; PATH: in `mappings` */
    uint32_t edit_begin, n_edits;     /* ALIGNMENT: in `edits` */
    uint32_t reserved;
} vgk_chain_piece;
typedef struct vgk_chain_mapping { uint32_t node, offset, edit_begin, n_edits; } vgk_chain_mapping;
typedef struct vgk_chain_result {
    int32_t  status;
    uint32_t mapping_begin, n_mappings;
    uint32_t edit_begin, n_edits;
    uint32_t from_length, to_length;  /* path_from_length / path_to_length of the result */
    uint32_t reserved;
} vgk_chain_result;
int  vgk_chain_stitch(vgk_ctx* ctx, const vgk_haplo* index,
                      const vgk_chain_piece* pieces, const uint64_t* piece_off /* n_reads + 1: read r = pieces[piece_off[r], piece_off[r + 1]) */, uint32_t n_reads,
                      const uint32_t* nodes, size_t n_nodes, const vgk_chain_mapping* mappings, size_t n_mappings, const uint32_t* edits, size_t n_edits,
                      vgk_chain_result* results, vgk_chain_mapping* out_mappings, size_t mapping_cap, uint32_t* out_edits, size_t edit_cap,
                      size_t written[2] /* mappings, edits */);
double vgk_chain_stitch_last_ms(vgk_ctx* ctx);   /* device time of the last call's kernels */

/* batch introspection (used by bench.py for the roofline line) */
void     vgk_batch_free(vgk_batch* batch);
int      vgk_batch_sync(vgk_batch* batch);
double   vgk_batch_kernel_ms(vgk_batch* batch, int which /* 0 = fill kernels (sum over launches), 1 = traceback tail after the last
                                                               fill, 2 = number of fill launches, -1 = fill + traceback; 3 = inside the traceback tail, the second fill of a speculative batch — one
                                                               geometry, mostly local alignments: the first fill builds no traceback codes, the reads whose alignment is not one
                                                               diagonal run are laid out again and filled with codes — 0 for other batches */);
uint64_t vgk_batch_cells(vgk_batch* batch);          /* DP cells computed per run            */
uint64_t vgk_batch_alg_bytes(vgk_batch* batch);      /* algorithmic bytes per run */
uint64_t vgk_batch_device_bytes(vgk_batch* batch);   /* HBM footprint of the batch            */
int      vgk_batch_lane(vgk_batch* batch);           /* launch lane (stream) of the batch: consecutive batches of a context alternate between
                                                        two, so that one batch's traceback runs under the next one's fill */
/* ---- the speculative fill and its feedback (no analogue in the reference: the engine's own risk) ------------------------------------------
 * A batch of mostly local alignments with tracebacks in one lane geometry may be filled WITHOUT traceback codes first: the alignments
 * that are one diagonal run are settled from the end cells, only the rest are laid out and filled again with codes.  That pays while few
 * reads miss (configs[1]: one in eight).  Each speculative run reports how many wavefronts it filled twice; the context stops speculating
 * when that share passes 0.25 (VGAMD_SPEC_MISS_MAX) and probes again after 16 runs (VGAMD_SPEC_PROBE_EVERY), doubling the wait up to 1024
 * while the probes keep failing.  Results are identical either way (tests/test_gssw_emu_parity.py).
 * vgk_set_speculation: 0 = by feedback (default), 1 = whenever a batch allows it, 2 = never.  vgk_speculation_state returns 1 while the
 * context speculates, 0 while it does not; counters (nullable) = runs observed, times turned off, times turned on, runs between probes now.
 * vgk_batch_speculated: did the batch's last run speculate. */
int      vgk_set_speculation(vgk_ctx* ctx, int mode);
int      vgk_speculation_state(vgk_ctx* ctx, uint64_t counters[4], double* last_miss_share);
int      vgk_batch_speculated(vgk_batch* batch);
uint64_t vgk_batch_wave_steps(vgk_batch* batch);     /* fill steps summed over the batch's wavefronts (one step = one graph column for
                                                        each of a wavefront's 64 lanes): the unit of the VALU-issue model in DESIGN.md */

#ifdef __cplusplus
}
#endif
#endif /* VGK_H */
