/*
 * vgo_chain.c — CPU ORACLE for vgk_chain_stitch: one Path per read out of the pieces of its chain.
 * TEST INFRASTRUCTURE ONLY: never included, linked or loaded by the product path (tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline / checker legs load oracle/libvgoracle.so; vg_amd/ never does).
 *
 * A literal restatement, object by object as the reference does it (protobuf Paths copied and rebuilt), of
 *   WFAAlignment::to_path                 /root/reference/src/gbwt_extender.cpp:954-1070
 *   append_path                           /root/reference/src/path.cpp:284-287
 *   simplify(const Path&, bool)           /root/reference/src/path.cpp:1314-1497
 *   simplify(const Mapping&, bool)        /root/reference/src/path.cpp:1509-1563
 *   concat_mappings                       /root/reference/src/path.cpp:1499-1507
 *   edits_are_compatible / merge_edits_in_place   /root/reference/src/path.cpp:1565-1580
 * as MinimizerMapper::find_chain_alignment drives them (/root/reference/src/minimizer_mapper_from_chains.cpp:2606, :2662, :2892, :3035,
 * :3103, :3147, :3262: append_path(composed_path, piece) ; :3295: simplify(composed_path, false)).
 * The engine's kernel (vg_amd/csrc/chain_device.hpp) streams mappings through the same rules in place; this file builds the objects.
 *
 * Pinned on the reference's own vectors for simplify (tests/golden/ref_simplify.json, extracted from src/unittest/path.cpp:21-45,
 * src/unittest/alignment.cpp:57-102) and, for to_path, on the [wfa_extender] cases whose alignments the reference's tests check
 * (tests/golden/ref_wfa_extender.json: check_alignment walks the same path / edits).  libvgio's edit predicates (edit_is_match ...) are an absent
 * dependency: restated from their use [prior knowledge] — match: from == to, no sequence; sub: from == to with sequence; insertion: from == 0 < to;
 * deletion: to == 0 < from; empty: from == to == 0 without sequence.
 */
#include <stdlib.h>
#include <string.h>
#include "../include/vgk.h"
#include "vgo_haplo.h"

typedef struct { uint32_t kind, len; } OEdit;                    /* kind = VGK_WFA_*; from_length / to_length / "has a sequence" follow from it */
typedef struct { int has_position; uint32_t node_id /* oriented node + 1; 0 = none */, offset; OEdit* edit; uint32_t n, cap; } OMapping;
typedef struct { OMapping* mapping; uint32_t n, cap; } OPath;

static uint32_t e_from(const OEdit* e) { return e->kind == VGK_WFA_INSERTION ? 0 : e->len; }
static uint32_t e_to(const OEdit* e) { return e->kind == VGK_WFA_DELETION ? 0 : e->len; }
static int edit_is_insertion(const OEdit* e) { return e->kind == VGK_WFA_INSERTION && e->len > 0; }
static int edit_is_deletion(const OEdit* e) { return e->kind == VGK_WFA_DELETION && e->len > 0; }
static int edit_is_empty(const OEdit* e) { return e->len == 0; }
static int edits_are_compatible(const OEdit* e, const OEdit* f) { return e->kind == f->kind; }           /* (:1565-1570: both matches, both subs, both deletions or both insertions) */

static void add_edit(OMapping* m, OEdit e) {
    if (m->n == m->cap) { m->cap = m->cap ? 2 * m->cap : 4; m->edit = (OEdit*)realloc(m->edit, sizeof(OEdit) * m->cap); }
    m->edit[m->n++] = e;
}
static OMapping copy_mapping(const OMapping* m) {
    OMapping c = *m; c.cap = m->n ? m->n : 1; c.edit = (OEdit*)malloc(sizeof(OEdit) * c.cap);
    if (m->n) memcpy(c.edit, m->edit, sizeof(OEdit) * m->n);
    return c;
}
static OMapping* add_mapping(OPath* p) {
    if (p->n == p->cap) { p->cap = p->cap ? 2 * p->cap : 8; p->mapping = (OMapping*)realloc(p->mapping, sizeof(OMapping) * p->cap); }
    OMapping* m = &p->mapping[p->n++]; memset(m, 0, sizeof *m);
    return m;
}
static void free_path(OPath* p) { for (uint32_t i = 0; i < p->n; ++i) free(p->mapping[i].edit); free(p->mapping); p->mapping = NULL; p->n = p->cap = 0; }
static uint32_t mapping_from_length(const OMapping* m) { uint32_t f = 0; for (uint32_t i = 0; i < m->n; ++i) f += e_from(&m->edit[i]); return f; }
static uint32_t mapping_to_length(const OMapping* m) { uint32_t t = 0; for (uint32_t i = 0; i < m->n; ++i) t += e_to(&m->edit[i]); return t; }

/* WFAAlignment::to_path (:954-1070) onto the end of `result` (which is what append_path(composed_path, to_path(...)) comes to).  -> VGK_OK, or
 * VGK_EINVAL where the reference throws */
static int wfa_to_path(const vgk_haplo* graph, const uint32_t* path, uint32_t path_len, uint32_t node_offset, const uint32_t* edits, uint32_t n_edits, OPath* result) {
    if (!path_len && n_edits == 1 && (edits[0] & 3u) == VGK_WFA_INSERTION) {        /* unlocalized_insertion() (:964-970): a mapping without a position */
        OMapping* m = add_mapping(result);
        OEdit e = { VGK_WFA_INSERTION, edits[0] >> 2 }; add_edit(m, e);
        return VGK_OK;
    }
    if (!path_len) return n_edits ? VGK_EINVAL : VGK_OK;                             /* (:972-974) */
    if (path[0] >= graph->n_oriented) return VGK_EINVAL;
    size_t node_cursor = node_offset, path_it = 0;
    size_t node_end = graph->len[path[0]];
    if (node_offset >= node_end) return VGK_EINVAL;                                  /* "offset to or past end of first node" */
    if (!n_edits) return VGK_EINVAL;                                                 /* "has no edits" */
    size_t edit_it = 0, current_edit_used = 0;
    OMapping* mapping_in_progress = add_mapping(result);
    mapping_in_progress->has_position = 1; mapping_in_progress->node_id = path[0] + 1; mapping_in_progress->offset = (uint32_t)node_cursor;
    while (edit_it != n_edits) {
        const uint32_t edit_type = edits[edit_it] & 3u, edit_len = edits[edit_it] >> 2;
        if (current_edit_used == edit_len) return VGK_EINVAL;                        /* "has empty edit" */
        size_t length_to_resolve = edit_len - current_edit_used;
        const int on_graph = edit_type == VGK_WFA_MATCH || edit_type == VGK_WFA_MISMATCH || edit_type == VGK_WFA_DELETION;
        if (on_graph) {
            if (path_it == path_len) return VGK_EINVAL;                              /* "tried to go past end of path" */
            if (node_cursor == node_end) return VGK_EINVAL;                          /* "tried to go past end of node" */
            if (node_end - node_cursor < length_to_resolve) length_to_resolve = node_end - node_cursor;
        }
        OEdit created = { edit_type, (uint32_t)length_to_resolve };
        add_edit(mapping_in_progress, created);
        if (on_graph) node_cursor += length_to_resolve;
        current_edit_used += length_to_resolve;
        if (current_edit_used == edit_len) { ++edit_it; current_edit_used = 0; }
        if (on_graph && node_cursor == node_end) {
            node_cursor = 0; ++path_it;
            if (path_it != path_len) {
                if (path[path_it] >= graph->n_oriented) return VGK_EINVAL;
                node_end = graph->len[path[path_it]];
                if (node_cursor == node_end) return VGK_EINVAL;                      /* "has empty node" */
                mapping_in_progress = add_mapping(result);
                mapping_in_progress->has_position = 1; mapping_in_progress->node_id = path[path_it] + 1; mapping_in_progress->offset = 0;
            } else node_end = 0;
        }
    }
    return VGK_OK;
}

/* Mapping simplify(const Mapping& m, bool trim_internal_deletions = false) (:1509-1563) */
static OMapping simplify_mapping(const OMapping* m) {
    OMapping n; memset(&n, 0, sizeof n);
    if (m->has_position) { n.has_position = 1; n.node_id = m->node_id; n.offset = m->offset; }
    size_t j = 0;
    if (j < m->n) {
        OEdit e = m->edit[j++];
        for (; j < m->n; ++j) {
            const OEdit* f = &m->edit[j];
            if (edit_is_empty(f)) continue;
            else if (edits_are_compatible(&e, f)) e.len += f->len;                   /* merge_edits_in_place */
            else { add_edit(&n, e); e = *f; }
        }
        add_edit(&n, e);
    }
    return n;
}
/* concat_mappings (:1499-1507) */
static OMapping concat_mappings(const OMapping* m, const OMapping* n) {
    OMapping c = copy_mapping(m);
    for (uint32_t i = 0; i < n->n; ++i) add_edit(&c, n->edit[i]);
    OMapping s = simplify_mapping(&c);
    free(c.edit);
    return s;
}
/* Path simplify(const Path& p, bool trim_internal_deletions = false) (:1314-1497) */
static OPath simplify_path(const OPath* p) {
    OPath s; memset(&s, 0, sizeof s);
    for (size_t i = 0; i < p->n; ++i) {
        OMapping m = simplify_mapping(&p->mapping[i]);
        if (m.n == 0) { free(m.edit); continue; }                                    /* empty mappings are redundant (:1334) */
        if (s.n) {
            OMapping* l = &s.mapping[s.n - 1];
            size_t edits_moved = 0;                                                  /* insertions at the start of m go to l (:1345-1352) */
            while (edits_moved < m.n && edit_is_insertion(&m.edit[edits_moved])) { add_edit(l, m.edit[edits_moved]); edits_moved++; }
            memmove(m.edit, m.edit + edits_moved, sizeof(OEdit) * (m.n - edits_moved)); m.n -= (uint32_t)edits_moved;
            if ((!l->has_position || l->node_id == 0) && (m.has_position && m.node_id != 0)) {                 /* (:1361-1369) */
                l->has_position = 1; l->node_id = m.node_id; l->offset = m.offset;
            } else if ((!m.has_position || m.node_id == 0) && (l->has_position && l->node_id != 0)) {          /* (:1371-1380) */
                m.has_position = 1; m.node_id = l->node_id;
                m.offset = mapping_from_length(l);                                   /* set_offset(from_length(*l)) — as written there */
            }
            if ((!l->has_position && !m.has_position)
                || (l->has_position && m.has_position && l->node_id == m.node_id     /* (is_reverse is part of the oriented node) */
                    && l->offset + mapping_from_length(l) == m.offset)) {
                OMapping joined = concat_mappings(l, &m);
                free(l->edit); *l = joined; free(m.edit);
            } else if (mapping_from_length(&m) || mapping_to_length(&m)) { *add_mapping(&s) = m; }
            else free(m.edit);
        } else *add_mapping(&s) = m;
    }
    OPath r; memset(&r, 0, sizeof r);                                                /* edit-less mappings go; empty positions are cleared (:1408-1420) */
    for (size_t i = 0; i < s.n; ++i) {
        const OMapping* m = &s.mapping[i];
        if (!m->n) continue;
        OMapping* l = add_mapping(&r); *l = copy_mapping(m);
        if (l->has_position && l->node_id == 0) l->has_position = 0;
    }
    free_path(&s);
    OPath q; memset(&q, 0, sizeof q);                                                /* leading and trailing deletions go (:1422-1475) */
    uint32_t total_to_length = 0, seen_to_length = 0;
    for (size_t i = 0; i < r.n; ++i) total_to_length += mapping_to_length(&r.mapping[i]);
    for (size_t i = 0; i < r.n; ++i) {
        const OMapping* m = &r.mapping[i];
        const uint32_t curr_to_length = mapping_to_length(m);
        if ((!seen_to_length && !curr_to_length) || seen_to_length == total_to_length) continue;
        OMapping n; memset(&n, 0, sizeof n);
        n.has_position = 1; n.node_id = m->has_position ? m->node_id : 0; n.offset = m->has_position ? m->offset : 0;     /* *n.mutable_position() = m.position() */
        if (seen_to_length) {
            if (seen_to_length + curr_to_length == total_to_length) {
                long j = (long)m->n - 1;
                for (; j >= 0; --j) if (!edit_is_deletion(&m->edit[j])) { ++j; break; }
                for (long h = 0; h < j; ++h) add_edit(&n, m->edit[h]);
            } else { free(n.edit); n = copy_mapping(m); }
        } else if (mapping_to_length(m)) {
            size_t j = 0, seen = 0;
            for (; j < m->n; ++j) { if (!edit_is_deletion(&m->edit[j])) break; seen += e_from(&m->edit[j]); }
            n.offset += (uint32_t)seen;
            for (; j < m->n; ++j) add_edit(&n, m->edit[j]);
        }
        *add_mapping(&q) = n;
        seen_to_length += mapping_to_length(&n);
    }
    free_path(&r);
    for (size_t i = 0; i < q.n; ++i) {                                               /* ranks (= index + 1); empty positions and empty edits go (:1479-1494) */
        OMapping* m = &q.mapping[i];
        if (m->node_id == 0) m->has_position = 0;
        uint32_t w = 0;
        for (uint32_t k = 0; k < m->n; ++k) if (!edit_is_empty(&m->edit[k])) m->edit[w++] = m->edit[k];
        m->n = w;
    }
    return q;
}

/* one read: its pieces appended (append_path), the whole simplified, flattened.  link_*: the last vgk_wfa_extend call's results (problem order). */
int vgo_chain_stitch_one(const vgk_haplo* index, const vgk_chain_piece* pieces, uint64_t n_pieces, const uint32_t* nodes, size_t n_nodes,
                         const vgk_chain_mapping* mappings, size_t n_mappings, const uint32_t* edits, size_t n_edits,
                         const vgk_wfa_result* link_res, const uint32_t* link_paths, const uint32_t* link_edits, uint32_t n_links,
                         vgk_chain_result* res, vgk_chain_mapping** out_m, uint32_t** out_e) {
    OPath composed; memset(&composed, 0, sizeof composed);
    int rc = VGK_OK;
    for (uint64_t k = 0; k < n_pieces && rc == VGK_OK; ++k) {
        const vgk_chain_piece* pc = &pieces[k];
        if (pc->kind == VGK_PIECE_LINK) {
            if (pc->link >= n_links || link_res[pc->link].status != VGK_OK || !link_res[pc->link].ok) { rc = VGK_EINVAL; break; }   /* "is not OK and cannot become a path" */
            const vgk_wfa_result* w = &link_res[pc->link];
            rc = wfa_to_path(index, link_paths + w->path_begin, w->path_len, w->node_offset, link_edits + w->edit_begin, w->n_edits, &composed);
        } else if (pc->kind == VGK_PIECE_ALIGNMENT) {
            if ((size_t)pc->path_begin + pc->path_len > n_nodes || (size_t)pc->edit_begin + pc->n_edits > n_edits) { rc = VGK_EINVAL; break; }
            rc = wfa_to_path(index, nodes + pc->path_begin, pc->path_len, pc->node_offset, edits + pc->edit_begin, pc->n_edits, &composed);
        } else if (pc->kind == VGK_PIECE_PATH) {
            if ((size_t)pc->path_begin + pc->path_len > n_mappings) { rc = VGK_EINVAL; break; }
            for (uint32_t q = 0; q < pc->path_len && rc == VGK_OK; ++q) {
                const vgk_chain_mapping* gm = &mappings[pc->path_begin + q];
                if ((size_t)gm->edit_begin + gm->n_edits > n_edits || (gm->node != VGK_WFA_NO_NODE && gm->node >= index->n_oriented)) { rc = VGK_EINVAL; break; }
                OMapping* m = add_mapping(&composed);
                if (gm->node != VGK_WFA_NO_NODE) { m->has_position = 1; m->node_id = gm->node + 1; m->offset = gm->offset; }
                for (uint32_t x = 0; x < gm->n_edits; ++x) { OEdit e = { edits[gm->edit_begin + x] & 3u, edits[gm->edit_begin + x] >> 2 }; if (e.len) add_edit(m, e); }   /* (a zero-length run is not an edit: include/vgk.h) */
            }
        } else rc = VGK_EINVAL;
    }
    memset(res, 0, sizeof *res); res->status = rc; *out_m = NULL; *out_e = NULL;
    if (rc == VGK_OK) {
        OPath q = simplify_path(&composed);
        uint32_t ne = 0;
        for (uint32_t i = 0; i < q.n; ++i) ne += q.mapping[i].n;
        *out_m = (vgk_chain_mapping*)malloc(sizeof(vgk_chain_mapping) * (q.n + 1)); *out_e = (uint32_t*)malloc(sizeof(uint32_t) * (ne + 1));
        uint32_t we = 0;
        for (uint32_t i = 0; i < q.n; ++i) {
            const OMapping* m = &q.mapping[i];
            vgk_chain_mapping* o = &(*out_m)[i];
            o->node = m->has_position ? m->node_id - 1 : VGK_WFA_NO_NODE; o->offset = m->has_position ? m->offset : 0; o->edit_begin = we; o->n_edits = m->n;
            for (uint32_t k = 0; k < m->n; ++k) (*out_e)[we++] = m->edit[k].len << 2 | m->edit[k].kind;
            res->from_length += mapping_from_length(m); res->to_length += mapping_to_length(m);
        }
        res->n_mappings = q.n; res->n_edits = ne;
        free_path(&q);
    }
    free_path(&composed);
    return rc;
}
