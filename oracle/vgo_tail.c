/*
 * vgo_tail.c — CPU ORACLE for giraffe's tail forests (the extraction half).
 *
 * TEST INFRASTRUCTURE ONLY (see vgo_engine.c): never linked or loaded by the product path.
 *
 * Restates, over the oracle's haplotype index (vgo_haplo.h; gbwt / gbwtgraph themselves are un-vendored submodules, absent from
 * the snapshot), the reference's src/minimizer_mapper.cpp:
 *   dfs_gbwt :5909-6013 — an iterative depth-first walk over GBWT search states with a stack of (state, distance used, visited)
 *     frames: on the first visit of a frame enter its handle (unless it is the root and nothing of the root is left behind the cut),
 *     add the node's length (the root: what is left of it) to the distance used and, while that is below the walk distance, push
 *     every non-empty one-node extension of the state (follow_paths: the node's outgoing edges in order); on the second visit — or
 *     when nothing was pushed — leave the handle and pop;
 *   get_tail_forest :5745-5860 — the enter / exit handlers: a list of (parent index, handle) pairs per tree, a stack of the
 *     indices of the open ancestors; a handle entered with no open ancestor starts a new tree.
 * Kept in the reference's shape (one tree vector, a parent stack, an explicit frame stack that grows as needed); the engine's kernel
 * (vg_amd/csrc/tail_device.hpp) stores the pushing tree node in the frame instead and runs the walk twice.
 *
 * Parity status: PARITY-UNPINNED — the reference holds no known-answer test for get_tail_forest / dfs_gbwt.  tests/test_tail_forest.py
 * pins this file on an independent construction: the trie of the continuations of the threads that pass through the start state,
 * built from the explicit thread lists.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../include/vgk.h"
#include "vgo_haplo.h"

typedef struct { SState here; uint32_t used; int visit; } Frame;
typedef struct { int32_t* parent; uint32_t* node; uint32_t* len; size_t n, cap; } Forest;

static int forest_push(Forest* f, int32_t parent, uint32_t node, uint32_t len) {
    if (f->n == f->cap) {
        const size_t cap = f->cap ? 2 * f->cap : 64;
        int32_t* p = (int32_t*)realloc(f->parent, sizeof(int32_t) * cap); if (!p) return 0; f->parent = p;
        uint32_t* q = (uint32_t*)realloc(f->node, sizeof(uint32_t) * cap); if (!q) return 0; f->node = q;
        uint32_t* r = (uint32_t*)realloc(f->len, sizeof(uint32_t) * cap); if (!r) return 0; f->len = r;
        f->cap = cap;
    }
    f->parent[f->n] = parent; f->node[f->n] = node; f->len[f->n] = len; ++f->n;
    return 1;
}

/* one tail; the tree nodes are appended to *f with parents as indices into *f (or -1) */
int vgo_tail_forest(const vgk_haplo* h, const vgk_tail_problem* pb, vgk_tail_result* out, int32_t** parent, uint32_t** node, uint32_t** len, size_t* n, size_t* cap) {
    Forest f = { *parent, *node, *len, *n, *cap };
    out->status = VGK_OK; out->first_node = (uint32_t)f.n; out->n_nodes = 0; out->n_trees = 0; out->root_trim = 0; out->bases = 0;
    if (pb->node >= h->n_oriented) { out->status = VGK_EINVAL; return VGK_OK; }
    const uint32_t root_len = h->len[pb->node];
    if (pb->offset > root_len || pb->lo < 0 || pb->hi >= (int32_t)h->count[pb->node]) { out->status = VGK_EINVAL; return VGK_OK; }
    if (pb->lo > pb->hi) return VGK_OK;                                   /* start_state.empty() (:5912) */
    const uint32_t remaining_root = root_len - pb->offset;               /* (:5925) */
    out->root_trim = remaining_root ? pb->offset : 0u;                    /* start_included ? from.offset() : 0 (:5800, :5838) */
    const size_t base = f.n;
    size_t scap = 64, sp = 0, pcap = 64, pp = 0;
    Frame* stack = (Frame*)malloc(sizeof(Frame) * scap); int64_t* parents = (int64_t*)malloc(sizeof(int64_t) * pcap);
    if (!stack || !parents) { free(stack); free(parents); return VGK_ENOMEM; }
    int rc = VGK_OK;
    stack[sp].here.node = (int32_t)pb->node; stack[sp].here.lo = pb->lo; stack[sp].here.hi = pb->hi; stack[sp].used = 0; stack[sp].visit = 0; ++sp;
    while (sp && rc == VGK_OK) {
        Frame* fr = &stack[sp - 1];
        const int is_root = sp == 1, hidden = is_root && remaining_root == 0;
        if (!fr->visit) {
            fr->visit = 1;
            const uint32_t o = (uint32_t)fr->here.node;
            const uint32_t node_length = is_root ? remaining_root : h->len[o];
            if (!hidden) {                                                /* enter_handle (:5823-5846) */
                const int64_t par = pp ? parents[pp - 1] : -1;
                if (pp == 0) ++out->n_trees;
                if (!forest_push(&f, par < 0 ? -1 : (int32_t)par, o, node_length)) { rc = VGK_ENOMEM; break; }
                if (pp == pcap) { pcap *= 2; int64_t* q = (int64_t*)realloc(parents, sizeof(int64_t) * pcap); if (!q) { rc = VGK_ENOMEM; break; } parents = q; }
                parents[pp++] = (int64_t)(f.n - 1);
                out->bases += node_length;
            }
            fr->used += node_length;
            if (fr->used < pb->walk_distance) {
                /* follow_paths: every outgoing edge in order, the states that are not empty (:5975-5983) */
                const uint32_t ne = h->edge_off[o + 1] - h->edge_off[o];
                const int32_t* et = h->edge_to + h->edge_off[o]; const uint32_t* body = h->body + h->body_off[o];
                const SState here = fr->here; const uint32_t used = fr->used;
                for (uint32_t e = 0; e < ne; ++e) {
                    if (et[e] < 0) continue;
                    int32_t before = 0, inside = 0;
                    for (int32_t i = 0; i <= here.hi; ++i) if (body[i] == e) { if (i < here.lo) ++before; else ++inside; }
                    if (!inside) continue;
                    if (sp == scap) { scap *= 2; Frame* q = (Frame*)realloc(stack, sizeof(Frame) * scap); if (!q) { rc = VGK_ENOMEM; break; } stack = q; }
                    Frame* c = &stack[sp++];
                    c->here.node = et[e]; c->here.lo = (int32_t)h->edge_base[h->edge_off[o] + e] + before; c->here.hi = c->here.lo + inside - 1;
                    c->used = used; c->visit = 0;
                }
                continue;
            }
        }
        if (!hidden) --pp;                                                /* exit_handle (:5848-5851) */
        --sp;
    }
    free(stack); free(parents);
    *parent = f.parent; *node = f.node; *len = f.len; *cap = f.cap;
    if (rc != VGK_OK) { *n = base; return rc; }
    *n = f.n;
    out->n_nodes = (uint32_t)(f.n - base);
    return VGK_OK;
}
/* the bases of a tree node: those of its oriented node behind `trim` */
void vgo_tail_copy_bases(const vgk_haplo* h, uint32_t node, uint32_t trim, uint32_t len, char* dst) { memcpy(dst, h->seq + h->seq_off[node] + trim, len); }
