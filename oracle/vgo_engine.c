/*
 * vgo_engine.c — the CPU ORACLE behind the same C ABI as the product
 * (include/vgk.h), so that tests can (a) diff the HIP engine against it call
 * for call and (b) exercise the host shim (vg_amd/host) without a GPU.
 *
 * TEST INFRASTRUCTURE ONLY.  liboracle is never loaded by the product path:
 * vg_amd/host/engine.cpp binds libvgamd.so by default and fails loudly if the
 * HIP library or a GPU is missing; only tests/ and bench.py's cpu_baseline leg
 * name this library explicitly.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../include/vgk.h"

int vgo_gssw_align_q(const vgk_scoring* sc, const vgk_qual_adj* qa, const vgk_gssw_problem* p,
                     vgk_result* res, vgk_op* ops, uint32_t ops_cap);
int vgo_xdrop_pinned_align_q(const vgk_scoring* sc, const vgk_qual_adj* qa, const vgk_gssw_problem* p,
                             vgk_result* res, vgk_op* ops, uint32_t ops_cap);

int vgo_banded_align(const vgk_scoring* sc, const vgk_qual_adj* qa, const vgk_banded_problem* p,
                     vgk_result* res, vgk_op* ops, uint32_t ops_cap);

int vgo_haplo_create(const vgk_haplotypes* d, vgk_haplo** out);
void vgo_haplo_destroy(vgk_haplo* h);
int vgo_gapless_extend(const vgk_scoring* sc, const vgk_haplo* h, const vgk_gapless_problem* p, vgk_gapless_result* res,
                       vgk_extension* ext_out, uint32_t ext_cap, uint32_t* nodes_out, uint32_t nodes_cap,
                       uint32_t* mism_out, uint32_t mism_cap, uint32_t* n_nodes_out, uint32_t* n_mism_out);

int vgo_wfa_one(const vgk_scoring* sc, const vgk_haplo* h, const vgk_wfa_error_model* model, const vgk_wfa_problem* p,
                vgk_wfa_result* res, int32_t** path_out, uint32_t** edits_out);

int vgo_gssw_pinned_multi(const vgk_scoring* sc, const vgk_qual_adj* qa, const vgk_gssw_problem* p, uint32_t max_alt_alns,
                          vgk_result** results_out, uint32_t* n_out, vgk_op** ops_out, uint32_t* n_ops_out);

struct vgk_ctx { vgk_scoring sc; int has_qa; vgk_qual_adj qa; int8_t qmat[256 * 25]; int8_t qbon[256];
                 /* what the last vgk_wfa_extend call answered, in problem order: vgk_chain_stitch's LINK pieces name these */
                 vgk_wfa_result* wfa_res; uint32_t* wfa_paths; uint32_t* wfa_edits; uint32_t wfa_n; const vgk_haplo* wfa_index; };

static int vgo_dispatch(const vgk_ctx* c, const vgk_gssw_problem* p, vgk_result* res, vgk_op* ops, uint32_t ops_cap) {
    const vgk_qual_adj* qa = c->has_qa ? &c->qa : NULL;
    return (p->flags & 15u) == VGK_XDROP_PINNED ? vgo_xdrop_pinned_align_q(&c->sc, qa, p, res, ops, ops_cap)
                                                : vgo_gssw_align_q(&c->sc, qa, p, res, ops, ops_cap);
}
struct vgk_batch {
    vgk_ctx* ctx; const vgk_gssw_problem* probs; uint32_t n; uint32_t ops_per;
    vgk_result* res; vgk_op* ops; int ran; uint64_t cells;
    /* window batches own the problems built for them */
    vgk_gssw_problem* own_probs; char* own_reads; uint32_t* own_pred_off; uint32_t* own_pred_idx;
    /* extension batches (vgk_gssw_pack_extensions): sub-DAG node k of problem i is window node ext_node[ext_off[i] + k]; ext_off[i + 1] == ext_off[i]:
       nothing lay in the extension's direction (the problem built for it is a placeholder whose result is dropped) */
    uint32_t* own_node_len; char* own_seq; uint32_t* ext_node; uint64_t* ext_off;
};

int vgk_abi_version(void) { return VGK_ABI_VERSION; }

const char* vgk_strerror(int code) {
    switch (code) {
        case VGK_OK: return "ok";
        case VGK_EINVAL: return "invalid argument";
        case VGK_ENODEV: return "no device";
        case VGK_ENOMEM: return "out of memory";
        case VGK_ETOOLONG: return "read too long";
        case VGK_EOVERFLOW: return "score overflow";
        case VGK_EOPS: return "cigar buffer too small";
        case VGK_ETOOBIG: return "band matrices too big";
        case VGK_ENOBAND: return "no alignment in band";
        case VGK_EUNSUPPORTED: return "unsupported scoring";
        default: return "unknown";
    }
}

int vgk_create(int device, const vgk_scoring* scoring, vgk_ctx** out) {
    (void)device;
    if (!scoring || !out) return VGK_EINVAL;
    vgk_ctx* c = (vgk_ctx*)calloc(1, sizeof *c);
    if (!c) return VGK_ENOMEM;
    c->sc = *scoring; *out = c; return VGK_OK;
}
int vgk_create_qual_adj(int device, const vgk_scoring* scoring, const vgk_qual_adj* qa, vgk_ctx** out) {
    if (!qa || !qa->matrix || !qa->bonuses) return VGK_EINVAL;
    int rc = vgk_create(device, scoring, out);
    if (rc) return rc;
    memcpy((*out)->qmat, qa->matrix, sizeof (*out)->qmat); memcpy((*out)->qbon, qa->bonuses, sizeof (*out)->qbon);
    (*out)->has_qa = 1; (*out)->qa.matrix = (*out)->qmat; (*out)->qa.bonuses = (*out)->qbon;
    return VGK_OK;
}
void vgk_destroy(vgk_ctx* ctx) { if (ctx) { free(ctx->wfa_res); free(ctx->wfa_paths); free(ctx->wfa_edits); } free(ctx); }

int vgk_host_register(vgk_ctx* ctx, const void* ptr, size_t bytes) { (void)ctx; (void)ptr; (void)bytes; return VGK_OK; }
int vgk_host_unregister(vgk_ctx* ctx, const void* ptr) { (void)ctx; (void)ptr; return VGK_OK; }
int vgk_device_info(vgk_ctx* ctx, char* name_out, size_t name_cap, int* cus, size_t* hbm) {
    (void)ctx;
    if (name_out && name_cap) { strncpy(name_out, "cpu-oracle", name_cap - 1); name_out[name_cap - 1] = 0; }
    if (cus) *cus = 0;
    if (hbm) *hbm = 0;
    return VGK_OK;
}

static uint32_t default_ops(const vgk_gssw_problem* p) {
    uint64_t R = 0; for (uint32_t i = 0; i < p->graph.n_nodes; ++i) R += p->graph.node_len[i];
    return (uint32_t)(p->read_len + R + p->graph.n_nodes + 4);
}

int vgk_gssw_pack(vgk_ctx* ctx, const vgk_gssw_problem* problems, uint32_t n,
                  uint32_t ops_per_problem, vgk_batch** out) {
    if (!ctx || (!problems && n) || !out) return VGK_EINVAL;
    vgk_batch* b = (vgk_batch*)calloc(1, sizeof *b);
    if (!b) return VGK_ENOMEM;
    b->ctx = ctx; b->probs = problems; b->n = n;
    uint32_t per = ops_per_problem;
    if (!per) for (uint32_t i = 0; i < n; ++i) { uint32_t d = default_ops(&problems[i]); if (d > per) per = d; }
    b->ops_per = per ? per : 1;
    b->res = (vgk_result*)calloc(n ? n : 1, sizeof(vgk_result));
    b->ops = (vgk_op*)calloc((size_t)(n ? n : 1) * b->ops_per, sizeof(vgk_op));
    if (!b->res || !b->ops) { free(b->res); free(b->ops); free(b); return VGK_ENOMEM; }
    for (uint32_t i = 0; i < n; ++i) {
        uint64_t R = 0; for (uint32_t k = 0; k < problems[i].graph.n_nodes; ++k) R += problems[i].graph.node_len[k];
        b->cells += R * problems[i].read_len;
    }
    *out = b; return VGK_OK;
}

int vgk_gssw_run(vgk_batch* b) {
    if (!b) return VGK_EINVAL;
    #pragma omp parallel for schedule(dynamic, 16)
    for (int64_t i = 0; i < (int64_t)b->n; ++i)
        vgo_dispatch(b->ctx, &b->probs[i], &b->res[i], b->ops + (size_t)i * b->ops_per, b->ops_per);
    b->ran = 1; return VGK_OK;
}

int vgo_xdrop_band_align_q(const vgk_scoring* sc, const vgk_qual_adj* qa, const vgk_gssw_problem* p,
                           vgk_result* res, vgk_op* ops, uint32_t ops_cap, uint64_t* stats);
int vgk_xdrop_band_align(vgk_ctx* ctx, const vgk_gssw_problem* problems, uint32_t n,
                         vgk_result* results, vgk_op* ops, size_t ops_cap, size_t* ops_written, uint64_t stats[2]) {
    if (!ctx || (!problems && n) || (!results && n)) return VGK_EINVAL;
    for (uint32_t i = 0; i < n; ++i) if ((problems[i].flags & 15u) != VGK_XDROP_PINNED) return VGK_EINVAL;
    size_t* slot = (size_t*)malloc(sizeof(size_t) * ((size_t)n + 1));
    slot[0] = 0;
    for (uint32_t i = 0; i < n; ++i) slot[i + 1] = slot[i] + default_ops(&problems[i]);
    vgk_op* scratch = (vgk_op*)malloc(sizeof(vgk_op) * (slot[n] + 1));
    uint64_t s0 = 0, s1 = 0;
    const vgk_qual_adj* qa = ctx->has_qa ? &ctx->qa : NULL;
    #pragma omp parallel for schedule(dynamic, 16) reduction(+:s0, s1)
    for (uint32_t i = 0; i < n; ++i) {
        uint64_t st[2] = {0, 0};
        if (problems[i].read_len > 511) { memset(&results[i], 0, sizeof results[i]); results[i].status = VGK_ETOOLONG; continue; }
        vgo_xdrop_band_align_q(&ctx->sc, qa, &problems[i], &results[i], scratch + slot[i], (uint32_t)(slot[i + 1] - slot[i]), st);
        s0 += st[0]; s1 += st[1];
    }
    size_t used = 0; int rc = VGK_OK;
    for (uint32_t i = 0; i < n; ++i) {
        if (results[i].status == VGK_OK && results[i].n_ops) {
            if (!ops || used + results[i].n_ops > ops_cap) { results[i].status = VGK_EOPS; results[i].n_ops = 0; rc = VGK_EOPS; }
            else memcpy(ops + used, scratch + slot[i], sizeof(vgk_op) * results[i].n_ops);
        } else results[i].n_ops = 0;
        results[i].ops_begin = (uint32_t)used; used += results[i].n_ops;
    }
    free(scratch); free(slot);
    if (ops_written) *ops_written = used;
    if (stats) { stats[0] = s0; stats[1] = s1; }
    return rc;
}

double vgk_xdrop_band_last_ms(vgk_ctx* ctx) { (void)ctx; return 0.0; }
int vgk_xdrop_band_last_cells(vgk_ctx* ctx) { (void)ctx; return 4; }      /* the checker's cells are int32 */
uint64_t vgk_xdrop_band_last_class(vgk_ctx* ctx, int which) { (void)ctx; (void)which; return 0; }      /* (no wavefronts to share here) */

/* threads of the OpenMP loops below (bench.py sets the CPUs the container may really use; the default is every hardware thread) */
#include <omp.h>
void vgo_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }

/* bench.py's cpu_baseline leg: the same batch through the SIMD restatement (vgo_gssw_fast.c).  Not part of the C ABI. */
int vgo_gssw_fast_batch(const vgk_scoring* sc, const vgk_gssw_problem* probs, uint32_t n, vgk_result* results, vgk_op* ops, uint32_t ops_per_problem);
int vgo_gssw_run_fast(vgk_batch* b) {
    if (!b) return VGK_EINVAL;
    if (b->ctx->has_qa) return VGK_EUNSUPPORTED;
    int rc = vgo_gssw_fast_batch(&b->ctx->sc, b->probs, b->n, b->res, b->ops, b->ops_per);
    if (rc == VGK_EUNSUPPORTED) return rc;
    for (uint32_t i = 0; i < b->n; ++i) b->res[i].ops_begin = 0;
    b->ran = 1; return VGK_OK;
}

int vgk_gssw_fetch(vgk_batch* b, vgk_result* results, vgk_op* ops, size_t ops_cap, size_t* ops_written) {
    if (!b || !results) return VGK_EINVAL;
    if (!b->ran) { int rc = vgk_gssw_run(b); if (rc) return rc; }
    size_t w = 0;
    for (uint32_t i = 0; i < b->n; ++i) {
        results[i] = b->res[i];
        results[i].ops_begin = (uint32_t)w;
        if (w + b->res[i].n_ops > ops_cap) { results[i].status = VGK_EOPS; results[i].n_ops = 0; continue; }
        if (ops) memcpy(ops + w, b->ops + (size_t)i * b->ops_per, sizeof(vgk_op) * b->res[i].n_ops);
        if (b->ext_off && results[i].status == VGK_OK) {                 /* the sub-DAG's nodes in the window's numbering */
            const uint64_t e0 = b->ext_off[i], cnt = b->ext_off[i + 1] - e0;
            if (!cnt) { results[i].score = 0; results[i].n_ops = 0; results[i].end_node = results[i].end_offset = results[i].end_read = -1; results[i].first_offset = 0; continue; }
            if (results[i].end_node >= 0 && (uint64_t)results[i].end_node < cnt) results[i].end_node = (int32_t)b->ext_node[e0 + (uint64_t)results[i].end_node];
            if (ops) for (uint32_t k = 0; k < results[i].n_ops; ++k) if (ops[w + k].node < cnt) ops[w + k].node = b->ext_node[e0 + ops[w + k].node];
        }
        w += b->res[i].n_ops;
    }
    if (ops_written) *ops_written = w;
    return VGK_OK;
}

int vgk_gssw_align(vgk_ctx* ctx, const vgk_gssw_problem* problems, uint32_t n,
                   vgk_result* results, vgk_op* ops, size_t ops_cap, size_t* ops_written) {
    vgk_batch* b = NULL;
    int rc = vgk_gssw_pack(ctx, problems, n, 0, &b);
    if (rc) return rc;
    rc = vgk_gssw_run(b);
    if (!rc) rc = vgk_gssw_fetch(b, results, ops, ops_cap, ops_written);
    vgk_batch_free(b);
    return rc;
}

int vgk_banded_align(vgk_ctx* ctx, const vgk_banded_problem* problems, uint32_t n,
                     vgk_result* results, vgk_op* ops, size_t ops_cap, size_t* ops_written) {
    if (!ctx || (!problems && n) || !results) return VGK_EINVAL;
    const vgk_qual_adj* qa = ctx->has_qa ? &ctx->qa : NULL;
    /* every problem into its own scratch slot (OpenMP over problems), then compacted in order */
    size_t* slot = (size_t*)malloc(sizeof(size_t) * ((size_t)n + 1));
    if (!slot) return VGK_ENOMEM;
    slot[0] = 0;
    for (uint32_t i = 0; i < n; ++i) {
        uint64_t R = 0; for (uint32_t v = 0; v < problems[i].graph.n_nodes; ++v) R += problems[i].graph.node_len[v];
        slot[i + 1] = slot[i] + problems[i].read_len + R + 2u * problems[i].graph.n_nodes + 8;
    }
    vgk_op* scratch = (vgk_op*)malloc(sizeof(vgk_op) * (slot[n] + 1));
    if (!scratch) { free(slot); return VGK_ENOMEM; }
    #pragma omp parallel for schedule(dynamic, 16)
    for (uint32_t i = 0; i < n; ++i)
        vgo_banded_align(&ctx->sc, qa, &problems[i], &results[i], scratch + slot[i], (uint32_t)(slot[i + 1] - slot[i]));
    size_t used = 0; int rc = VGK_OK;
    for (uint32_t i = 0; i < n; ++i) {
        if (results[i].status == VGK_OK) {
            if (!ops || used + results[i].n_ops > ops_cap) { results[i].status = VGK_EOPS; results[i].n_ops = 0; rc = VGK_EOPS; }
            else memcpy(ops + used, scratch + slot[i], sizeof(vgk_op) * results[i].n_ops);
        }
        results[i].ops_begin = (uint32_t)used; used += results[i].n_ops;
    }
    free(scratch); free(slot);
    if (ops_written) *ops_written = used;
    return rc;
}
int vgo_banded_align_multi(const vgk_scoring* sc, const vgk_qual_adj* qa, const vgk_banded_problem* p, uint32_t max_alns,
                           vgk_result* results, uint32_t* n_alns, vgk_op* ops, uint32_t ops_cap);
int vgk_banded_align_multi(vgk_ctx* ctx, const vgk_banded_problem* problems, uint32_t n, uint32_t max_alt_alns,
                           vgk_result* results, uint32_t* n_alignments, vgk_op* ops, size_t ops_cap, size_t* ops_written) {
    if (!ctx || (!problems && n) || !results || !n_alignments || !max_alt_alns) return VGK_EINVAL;
    const vgk_qual_adj* qa = ctx->has_qa ? &ctx->qa : NULL;
    size_t used = 0; int rc = VGK_OK;
    for (uint32_t i = 0; i < n; ++i) {
        vgk_result* r = results + (size_t)i * max_alt_alns;
        const size_t room = ops_cap - used;
        int st = vgo_banded_align_multi(&ctx->sc, qa, &problems[i], max_alt_alns, r, &n_alignments[i], ops ? ops + used : NULL, room > 0xffffffffu ? 0xffffffffu : (uint32_t)room);
        if (st == VGK_EOPS) rc = VGK_EOPS;
        size_t w = 0;
        for (uint32_t k = 0; k < n_alignments[i]; ++k) { r[k].ops_begin += (uint32_t)used; w += r[k].n_ops; }
        used += w;
    }
    if (ops_written) *ops_written = used;
    return rc;
}
double vgk_banded_last(vgk_ctx* ctx, int which) { (void)ctx; (void)which; return 0.0; }

int vgk_haplo_create(vgk_ctx* ctx, const vgk_haplotypes* haplotypes, vgk_haplo** out) { (void)ctx; return vgo_haplo_create(haplotypes, out); }
void vgk_haplo_destroy(vgk_haplo* index) { vgo_haplo_destroy(index); }
double vgk_gapless_last_ms(vgk_ctx* ctx) { (void)ctx; return 0.0; }
uint64_t vgk_gapless_last_retried(vgk_ctx* ctx) { (void)ctx; return 0; }
uint64_t vgk_gapless_last_redone(vgk_ctx* ctx) { (void)ctx; return 0; }      /* (the oracle walks the graph's own nodes, one by one) */
int vgk_gapless_rerun(vgk_ctx* ctx) { (void)ctx; return VGK_EINVAL; }     /* nothing is resident on the CPU */
int vgk_banded_rerun(vgk_ctx* ctx) { (void)ctx; return VGK_EINVAL; }
int vgk_gapless_extend(vgk_ctx* ctx, const vgk_haplo* index, const vgk_gapless_problem* problems, uint32_t n,
                       vgk_gapless_result* results, vgk_extension* extensions, size_t ext_cap,
                       uint32_t* nodes, size_t nodes_cap, uint32_t* mismatches, size_t mism_cap, size_t written[3]) {
    if (!ctx || !index || (!problems && n) || (!results && n)) return VGK_EINVAL;
    /* every problem into its own scratch slots (OpenMP over problems), then compacted in order */
    size_t* se = (size_t*)malloc(sizeof(size_t) * ((size_t)n + 1)); size_t* sn = (size_t*)malloc(sizeof(size_t) * ((size_t)n + 1));
    size_t* sm = (size_t*)malloc(sizeof(size_t) * ((size_t)n + 1));
    uint32_t* wn = (uint32_t*)calloc((size_t)n + 1, sizeof(uint32_t)); uint32_t* wm = (uint32_t*)calloc((size_t)n + 1, sizeof(uint32_t));
    se[0] = sn[0] = sm[0] = 0;
    for (uint32_t i = 0; i < n; ++i) {
        se[i + 1] = se[i] + problems[i].n_seeds; sn[i + 1] = sn[i] + (size_t)problems[i].n_seeds * (problems[i].read_len + 2);
        sm[i + 1] = sm[i] + (size_t)problems[i].n_seeds * problems[i].read_len;
    }
    vgk_extension* te = (vgk_extension*)malloc(sizeof(vgk_extension) * (se[n] + 1));
    uint32_t* tn = (uint32_t*)malloc(sizeof(uint32_t) * (sn[n] + 1)); uint32_t* tm = (uint32_t*)malloc(sizeof(uint32_t) * (sm[n] + 1));
    #pragma omp parallel for schedule(dynamic, 64)
    for (uint32_t i = 0; i < n; ++i)
        vgo_gapless_extend(&ctx->sc, index, &problems[i], &results[i], te + se[i], (uint32_t)(se[i + 1] - se[i]), tn + sn[i], (uint32_t)(sn[i + 1] - sn[i]),
                           tm + sm[i], (uint32_t)(sm[i + 1] - sm[i]), &wn[i], &wm[i]);
    size_t ne = 0, nn = 0, nm = 0; int rc = VGK_OK;
    for (uint32_t i = 0; i < n; ++i) {
        vgk_gapless_result* r = &results[i];
        if (r->status == VGK_OK && (ne + r->n_ext > ext_cap || nn + wn[i] > nodes_cap || nm + wm[i] > mism_cap)) { r->status = VGK_EOPS; r->n_ext = 0; rc = VGK_EOPS; }
        r->ext_begin = (uint32_t)ne;
        if (r->status != VGK_OK) { r->n_ext = 0; continue; }
        for (uint32_t k = 0; k < r->n_ext; ++k) {
            vgk_extension x = te[se[i] + k];
            x.path_begin += (uint32_t)nn; x.mism_begin += (uint32_t)nm;
            extensions[ne + k] = x;
        }
        memcpy(nodes + nn, tn + sn[i], sizeof(uint32_t) * wn[i]); memcpy(mismatches + nm, tm + sm[i], sizeof(uint32_t) * wm[i]);
        ne += r->n_ext; nn += wn[i]; nm += wm[i];
    }
    free(se); free(sn); free(sm); free(wn); free(wm); free(te); free(tn); free(tm);
    if (written) { written[0] = ne; written[1] = nn; written[2] = nm; }
    return rc;
}

double vgk_wfa_last_ms(vgk_ctx* ctx) { (void)ctx; return 0.0; }
double vgk_wfa_last_wave(vgk_ctx* ctx, int which) { (void)ctx; (void)which; return 0.0; }
uint64_t vgk_gssw_multi_host_walks(const vgk_ctx* ctx) { (void)ctx; return 0; }      /* (the oracle has no device: every problem is walked here) */
int vgk_wfa_set_form(vgk_ctx* ctx, int form) { (void)ctx; (void)form; return VGK_OK; }
int vgk_wfa_get_form(vgk_ctx* ctx) { (void)ctx; return 0; }
int vgk_wfa_set_cost_hints(vgk_ctx* ctx, const uint32_t* extra_bases, uint32_t n) { (void)ctx; (void)extra_bases; (void)n; return VGK_OK; }   /* (no launch to order) */
int vgk_wfa_set_point_budget(vgk_ctx* ctx, uint32_t points) { (void)points; return ctx ? VGK_OK : VGK_EINVAL; }      /* the oracle has no tables to outgrow */
int vgk_wfa_set_point_budgets(vgk_ctx* ctx, uint32_t connect_points, uint32_t tail_points) { (void)ctx; (void)connect_points; (void)tail_points; return VGK_OK; }
int vgk_wfa_rerun(vgk_ctx* ctx) { (void)ctx; return VGK_EINVAL; }
int vgk_wfa_extend(vgk_ctx* ctx, const vgk_haplo* index, const vgk_wfa_error_model* model, const vgk_wfa_problem* problems, uint32_t n,
                   vgk_wfa_result* results, uint32_t* paths, size_t path_cap, uint32_t* edits, size_t edit_cap, size_t written[2]) {
    if (!ctx || !index || (!problems && n) || (!results && n)) return VGK_EINVAL;
    if (model) { const vgk_wfa_event* ev[3] = { &model->mismatches, &model->gaps, &model->gap_length };
                 for (int k = 0; k < 3; ++k) if (ev[k]->per_base < 0 || ev[k]->min < 0 || ev[k]->max < ev[k]->min) return VGK_EINVAL; }   /* the constructor's asserts (:1262-1270) */
    if (ctx->sc.matrix[0] < 0 || ctx->sc.matrix[1] >= 0 || ctx->sc.gap_open < ctx->sc.gap_extend || ctx->sc.gap_extend == 0) return VGK_EUNSUPPORTED;   /* (:1256-1259) */
    int32_t** tp = (int32_t**)calloc((size_t)n + 1, sizeof(int32_t*)); uint32_t** te = (uint32_t**)calloc((size_t)n + 1, sizeof(uint32_t*));
    #pragma omp parallel for schedule(dynamic, 64)
    for (uint32_t i = 0; i < n; ++i) vgo_wfa_one(&ctx->sc, index, model, &problems[i], &results[i], &tp[i], &te[i]);
    size_t np = 0, ne = 0; int rc = VGK_OK;
    const int scores_only = !paths && !edits && !path_cap && !edit_cap;           /* (include/vgk.h: results without paths and edit runs) */
    {   /* the call's alignments stay with the context (the engine keeps them in HBM) for vgk_chain_stitch */
        size_t tp_all = 0, te_all = 0;
        for (uint32_t i = 0; i < n; ++i) if (results[i].status == VGK_OK && results[i].ok) { tp_all += results[i].path_len; te_all += results[i].n_edits; }
        free(ctx->wfa_res); free(ctx->wfa_paths); free(ctx->wfa_edits);
        ctx->wfa_res = (vgk_wfa_result*)malloc(sizeof(vgk_wfa_result) * ((size_t)n + 1)); ctx->wfa_paths = (uint32_t*)malloc(sizeof(uint32_t) * (tp_all + 1));
        ctx->wfa_edits = (uint32_t*)malloc(sizeof(uint32_t) * (te_all + 1)); ctx->wfa_n = n; ctx->wfa_index = index;
        size_t kp = 0, ke = 0;
        for (uint32_t i = 0; i < n; ++i) {
            vgk_wfa_result r = results[i];
            if (r.status != VGK_OK || !r.ok) { r.ok = 0; r.path_len = r.n_edits = 0; }
            r.path_begin = (uint32_t)kp; r.edit_begin = (uint32_t)ke;
            for (uint32_t k = 0; k < r.path_len; ++k) ctx->wfa_paths[kp + k] = (uint32_t)tp[i][k];
            if (r.n_edits) memcpy(ctx->wfa_edits + ke, te[i], sizeof(uint32_t) * r.n_edits);
            kp += r.path_len; ke += r.n_edits;
            ctx->wfa_res[i] = r;
        }
    }
    for (uint32_t i = 0; i < n; ++i) {
        vgk_wfa_result* r = &results[i];
        if (scores_only) { if (r->status != VGK_OK) r->ok = 0; r->path_begin = r->path_len = r->edit_begin = r->n_edits = 0; free(tp[i]); free(te[i]); continue; }
        if (r->status == VGK_OK && (np + r->path_len > path_cap || ne + r->n_edits > edit_cap)) { r->status = VGK_EOPS; rc = VGK_EOPS; }
        r->path_begin = (uint32_t)np; r->edit_begin = (uint32_t)ne;
        if (r->status != VGK_OK) { r->ok = 0; r->path_len = r->n_edits = 0; }
        for (uint32_t k = 0; k < r->path_len; ++k) paths[np + k] = (uint32_t)tp[i][k];
        if (r->n_edits) memcpy(edits + ne, te[i], sizeof(uint32_t) * r->n_edits);
        np += r->path_len; ne += r->n_edits;
        free(tp[i]); free(te[i]);
    }
    free(tp); free(te);
    if (written) { written[0] = np; written[1] = ne; }
    return rc;
}

/* vgk_chain_stitch: read by read through vgo_chain.c */
int vgo_chain_stitch_one(const vgk_haplo* index, const vgk_chain_piece* pieces, uint64_t n_pieces, const uint32_t* nodes, size_t n_nodes,
                         const vgk_chain_mapping* mappings, size_t n_mappings, const uint32_t* edits, size_t n_edits,
                         const vgk_wfa_result* link_res, const uint32_t* link_paths, const uint32_t* link_edits, uint32_t n_links,
                         vgk_chain_result* res, vgk_chain_mapping** out_m, uint32_t** out_e);
double vgk_chain_stitch_last_ms(vgk_ctx* ctx) { (void)ctx; return 0.0; }
int vgk_chain_stitch(vgk_ctx* ctx, const vgk_haplo* index, const vgk_chain_piece* pieces, const uint64_t* piece_off, uint32_t n_reads,
                     const uint32_t* nodes, size_t n_nodes, const vgk_chain_mapping* mappings, size_t n_mappings, const uint32_t* edits, size_t n_edits,
                     vgk_chain_result* results, vgk_chain_mapping* out_mappings, size_t mapping_cap, uint32_t* out_edits, size_t edit_cap, size_t written[2]) {
    if (!ctx || !index || !piece_off || (!results && n_reads) || (!pieces && n_reads && piece_off[n_reads]) || (!nodes && n_nodes) || (!mappings && n_mappings) || (!edits && n_edits)
        || (!out_mappings && mapping_cap) || (!out_edits && edit_cap)) return VGK_EINVAL;
    for (uint32_t r = 0; r < n_reads; ++r) if (piece_off[r + 1] < piece_off[r]) return VGK_EINVAL;
    vgk_chain_mapping** tm = (vgk_chain_mapping**)calloc((size_t)n_reads + 1, sizeof *tm); uint32_t** te = (uint32_t**)calloc((size_t)n_reads + 1, sizeof *te);
    const uint32_t n_links = ctx->wfa_index == index ? ctx->wfa_n : 0;
    #pragma omp parallel for schedule(dynamic, 16)
    for (uint32_t r = 0; r < n_reads; ++r)
        vgo_chain_stitch_one(index, pieces + piece_off[r], piece_off[r + 1] - piece_off[r], nodes, n_nodes, mappings, n_mappings, edits, n_edits,
                             ctx->wfa_res, ctx->wfa_paths, ctx->wfa_edits, n_links, &results[r], &tm[r], &te[r]);
    size_t nm = 0, ne = 0; int rc = VGK_OK;
    for (uint32_t r = 0; r < n_reads; ++r) {
        vgk_chain_result* o = &results[r];
        o->mapping_begin = (uint32_t)nm; o->edit_begin = (uint32_t)ne;
        if (nm + o->n_mappings <= mapping_cap && ne + o->n_edits <= edit_cap) {
            for (uint32_t k = 0; k < o->n_mappings; ++k) { out_mappings[nm + k] = tm[r][k]; out_mappings[nm + k].edit_begin += (uint32_t)ne; }
            if (o->n_edits) memcpy(out_edits + ne, te[r], sizeof(uint32_t) * o->n_edits);
        } else { if (o->status == VGK_OK) o->status = VGK_EOPS; rc = VGK_EOPS; }
        nm += o->n_mappings; ne += o->n_edits;
        free(tm[r]); free(te[r]);
    }
    free(tm); free(te);
    if (written) { written[0] = nm; written[1] = ne; }
    return rc;
}

int vgk_gssw_align_multi(vgk_ctx* ctx, const vgk_gssw_problem* problems, uint32_t n, uint32_t max_alt_alns,
                         vgk_result* results, uint32_t* n_alignments, vgk_op* ops, size_t ops_cap, size_t* ops_written) {
    if (!ctx || (!problems && n) || (!results && n) || (!n_alignments && n) || !max_alt_alns) return VGK_EINVAL;
    vgk_result** tr = (vgk_result**)calloc((size_t)n + 1, sizeof(vgk_result*)); vgk_op** to = (vgk_op**)calloc((size_t)n + 1, sizeof(vgk_op*));
    uint32_t* tn = (uint32_t*)calloc((size_t)n + 1, sizeof(uint32_t)); uint32_t* tno = (uint32_t*)calloc((size_t)n + 1, sizeof(uint32_t));
    int* st = (int*)calloc((size_t)n + 1, sizeof(int));
    const vgk_qual_adj* qa = ctx->has_qa ? &ctx->qa : NULL;
    #pragma omp parallel for schedule(dynamic, 16)
    for (uint32_t i = 0; i < n; ++i) st[i] = vgo_gssw_pinned_multi(&ctx->sc, qa, &problems[i], max_alt_alns, &tr[i], &tn[i], &to[i], &tno[i]);
    size_t used = 0; int rc = VGK_OK;
    for (uint32_t i = 0; i < n; ++i) {
        vgk_result* r = results + (size_t)i * max_alt_alns;
        memset(r, 0, sizeof(vgk_result) * max_alt_alns);
        n_alignments[i] = 0;
        if (st[i] != VGK_OK) r->status = st[i];
        else if (used + tno[i] > ops_cap || (tno[i] && !ops)) { r->status = VGK_EOPS; rc = VGK_EOPS; }
        else {
            for (uint32_t k = 0; k < tn[i]; ++k) { r[k] = tr[i][k]; r[k].ops_begin += (uint32_t)used; }
            if (tno[i]) memcpy(ops + used, to[i], sizeof(vgk_op) * tno[i]);
            used += tno[i]; n_alignments[i] = tn[i];
        }
        free(tr[i]); free(to[i]);
    }
    free(tr); free(to); free(tn); free(tno); free(st);
    if (ops_written) *ops_written = used;
    return rc;
}

void vgk_batch_free(vgk_batch* b) { if (b) { free(b->res); free(b->ops); free(b->own_probs); free(b->own_reads); free(b->own_pred_off); free(b->own_pred_idx);
                                        free(b->own_node_len); free(b->own_seq); free(b->ext_node); free(b->ext_off); free(b); } }

/* ---- windows of one graph (vgk_graph_create / vgk_gssw_pack_windows) ---------------------------------------------------
 * The oracle's reading of "a window is the induced subgraph on nodes [first, first + n)": every problem gets its own
 * predecessor CSR with the edges from outside the window dropped and the indices re-based, then runs through the ordinary
 * per-problem oracle.  (The engine derives its arenas from resident tables instead; the two constructions share nothing.) */
struct vgk_dgraph { uint32_t n_nodes; uint32_t* node_len; uint64_t* seq_off; char* seq; uint32_t* pred_off; uint32_t* pred_idx; };
int vgk_graph_create(vgk_ctx* ctx, const vgk_graph* g, vgk_dgraph** out) {
    if (!ctx || !g || !out || !g->n_nodes || !g->node_len || !g->seq || !g->pred_off) return VGK_EINVAL;
    *out = NULL;
    const uint32_t n = g->n_nodes;
    uint64_t bases = 0;
    for (uint32_t v = 0; v < n; ++v) {
        if (g->node_len[v] == 0 || g->pred_off[v + 1] < g->pred_off[v]) return VGK_EINVAL;
        if (g->node_len[v] > 65535u) return VGK_ETOOBIG;
        for (uint32_t k = g->pred_off[v]; k < g->pred_off[v + 1]; ++k) if (!g->pred_idx || g->pred_idx[k] >= v) return VGK_EINVAL;
        bases += g->node_len[v];
    }
    if (bases >= (1ull << 32) - 16) return VGK_ETOOBIG;
    vgk_dgraph* d = (vgk_dgraph*)calloc(1, sizeof *d);
    if (!d) return VGK_ENOMEM;
    const uint32_t ne = g->pred_off[n] - g->pred_off[0];
    d->n_nodes = n;
    d->node_len = (uint32_t*)malloc(sizeof(uint32_t) * n); d->seq_off = (uint64_t*)malloc(sizeof(uint64_t) * ((size_t)n + 1));
    d->seq = (char*)malloc(bases + 1); d->pred_off = (uint32_t*)malloc(sizeof(uint32_t) * ((size_t)n + 1)); d->pred_idx = (uint32_t*)malloc(sizeof(uint32_t) * ((size_t)ne + 1));
    if (!d->node_len || !d->seq_off || !d->seq || !d->pred_off || !d->pred_idx) { vgk_graph_destroy(d); return VGK_ENOMEM; }
    memcpy(d->node_len, g->node_len, sizeof(uint32_t) * n); memcpy(d->seq, g->seq, bases);
    d->seq_off[0] = 0;
    for (uint32_t v = 0; v < n; ++v) d->seq_off[v + 1] = d->seq_off[v] + g->node_len[v];
    for (uint32_t v = 0; v <= n; ++v) d->pred_off[v] = g->pred_off[v] - g->pred_off[0];
    if (ne) memcpy(d->pred_idx, g->pred_idx + g->pred_off[0], sizeof(uint32_t) * ne);
    *out = d; return VGK_OK;
}
void vgk_graph_destroy(vgk_dgraph* d) { if (d) { free(d->node_len); free(d->seq_off); free(d->seq); free(d->pred_off); free(d->pred_idx); free(d); } }

int vgk_gssw_pack_windows(vgk_ctx* ctx, const vgk_dgraph* g, const char* reads, size_t reads_bytes,
                          const vgk_window_problem* problems, uint32_t n, uint32_t ops_per_problem, vgk_batch** out) {
    if (!ctx || !g || !out || (!problems && n) || (!reads && reads_bytes)) return VGK_EINVAL;
    *out = NULL;
    if (ctx->has_qa) return VGK_EUNSUPPORTED;
    int maxs = 0;
    for (int k = 0; k < 25; ++k) if (ctx->sc.matrix[k] > maxs) maxs = ctx->sc.matrix[k];
    uint64_t tot_nodes = 0, tot_edges = 0;
    for (uint32_t i = 0; i < n; ++i) {      /* the first problem that fails decides, with the engine's codes */
        const vgk_window_problem* p = &problems[i];
        const uint32_t mode = p->flags & 15u;
        const uint32_t rows = p->read_len + (mode == VGK_XDROP_PINNED ? 1u : 0u);
        if (p->read_len == 0 || p->n_nodes == 0 || (uint64_t)p->first_node + p->n_nodes > g->n_nodes || p->read_off + p->read_len > reads_bytes) return VGK_EINVAL;
        if (mode != VGK_GSSW_LOCAL && mode != VGK_XDROP_PINNED) return VGK_EINVAL;
        if (rows > 1024) return VGK_ETOOLONG;
        if ((int64_t)rows * maxs + 2 * (int64_t)ctx->sc.full_length_bonus > 2046) return VGK_EUNSUPPORTED;
        if (mode == VGK_XDROP_PINNED && (int64_t)p->read_len * maxs + ctx->sc.full_length_bonus >= 1023) return VGK_EUNSUPPORTED;
        if (g->seq_off[p->first_node + p->n_nodes] - g->seq_off[p->first_node] >= (1u << 20)) return VGK_ETOOBIG;
        tot_nodes += p->n_nodes; tot_edges += g->pred_off[p->first_node + p->n_nodes] - g->pred_off[p->first_node];
    }
    vgk_gssw_problem* pr = (vgk_gssw_problem*)calloc(n ? n : 1, sizeof *pr);
    uint32_t* po = (uint32_t*)malloc(sizeof(uint32_t) * (tot_nodes + n + 1)); uint32_t* pi = (uint32_t*)malloc(sizeof(uint32_t) * (tot_edges + 1));
    char* rd = (char*)malloc(reads_bytes + 1);
    if (!pr || !po || !pi || !rd) { free(pr); free(po); free(pi); free(rd); return VGK_ENOMEM; }
    if (reads_bytes) memcpy(rd, reads, reads_bytes);
    uint64_t at_o = 0, at_e = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const vgk_window_problem* p = &problems[i];
        vgk_gssw_problem* q = &pr[i];
        q->read = rd + p->read_off; q->read_len = p->read_len; q->flags = p->flags; q->max_gap_length = p->max_gap_length;
        q->graph.n_nodes = p->n_nodes; q->graph.node_len = g->node_len + p->first_node; q->graph.seq = g->seq + g->seq_off[p->first_node];
        q->graph.pred_off = po + at_o; q->graph.pred_idx = pi + at_e;
        uint32_t ne = 0;
        for (uint32_t k = 0; k < p->n_nodes; ++k) {
            const uint32_t v = p->first_node + k;
            po[at_o + k] = ne;
            for (uint32_t e = g->pred_off[v]; e < g->pred_off[v + 1]; ++e)
                if (g->pred_idx[e] >= p->first_node) pi[at_e + ne++] = g->pred_idx[e] - p->first_node;
        }
        po[at_o + p->n_nodes] = ne;
        at_o += p->n_nodes + 1; at_e += ne;
    }
    vgk_batch* b = NULL;
    int rc = vgk_gssw_pack(ctx, pr, n, ops_per_problem, &b);
    if (rc) { free(pr); free(po); free(pi); free(rd); return rc; }
    b->own_probs = pr; b->own_reads = rd; b->own_pred_off = po; b->own_pred_idx = pi;
    *out = b; return VGK_OK;
}
/* ---- extension windows (vgk_gssw_pack_extensions): one pass of a seeded X-drop alignment from a position inside a window ------------------
 * The oracle's construction of the sub-DAG, problem by problem, as DozeuInterface hands dozeu its nodes (src/dozeu_interface.cpp:236-243: the
 * start node's sequence from the offset on; :178-185: a leftward pass reads the bases before the offset backwards; :261-283: a node is
 * visited when a forefront of a neighbour on the start's side reaches it) and as the host shim builds it for one subgraph
 * (vg_amd/host/aligner.cpp, xdrop_extend_prepare): explicit strings, an explicit predecessor CSR, then the ordinary per-problem oracle.
 * (The engine derives the same sub-DAG on the device from the resident tables: gssw_pack_device.hpp ext_size_one / ext_emit_one.) */
int vgk_gssw_pack_extensions(vgk_ctx* ctx, const vgk_dgraph* g, const char* reads, size_t reads_bytes,
                             const vgk_extension_problem* problems, uint32_t n, uint32_t ops_per_problem, vgk_batch** out) {
    if (!ctx || !g || !out || (!problems && n) || (!reads && reads_bytes)) return VGK_EINVAL;
    *out = NULL;
    if (ctx->has_qa) return VGK_EUNSUPPORTED;
    int maxs = 0;
    for (int k = 0; k < 25; ++k) if (ctx->sc.matrix[k] > maxs) maxs = ctx->sc.matrix[k];
    uint64_t tot_nodes = 0, tot_edges = 0, tot_bases = 0, tot_query = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const vgk_extension_problem* p = &problems[i];
        if (p->read_len == 0 || p->n_nodes == 0 || (uint64_t)p->first_node + p->n_nodes > g->n_nodes || p->read_off + p->read_len > reads_bytes) return VGK_EINVAL;
        if ((p->flags & 15u) != VGK_XDROP_PINNED || p->query_offset > p->read_len) return VGK_EINVAL;
        const uint32_t qlen = p->leftward ? p->query_offset : p->read_len - p->query_offset;
        if (qlen == 0 || p->start_node < p->first_node || p->start_node >= p->first_node + p->n_nodes || p->start_offset > g->node_len[p->start_node]) return VGK_EINVAL;
        if (qlen + 1 > 1024) return VGK_ETOOLONG;
        if ((int64_t)(qlen + 1) * maxs + 2 * (int64_t)ctx->sc.full_length_bonus > 2046) return VGK_EUNSUPPORTED;
        if ((int64_t)qlen * maxs + ctx->sc.full_length_bonus >= 1023) return VGK_EUNSUPPORTED;
        tot_nodes += p->n_nodes; tot_edges += g->pred_off[p->first_node + p->n_nodes] - g->pred_off[p->first_node];
        tot_bases += g->seq_off[p->first_node + p->n_nodes] - g->seq_off[p->first_node] + 1; tot_query += qlen;
    }
    vgk_gssw_problem* pr = (vgk_gssw_problem*)calloc(n ? n : 1, sizeof *pr);
    uint32_t* po = (uint32_t*)malloc(sizeof(uint32_t) * (tot_nodes + 2ull * n + 1)); uint32_t* pi = (uint32_t*)malloc(sizeof(uint32_t) * (tot_edges + 1));
    uint32_t* nl = (uint32_t*)malloc(sizeof(uint32_t) * (tot_nodes + n + 1)); char* sq = (char*)malloc(tot_bases + n + 1);
    char* rd = (char*)malloc(tot_query + 1);
    uint32_t* en = (uint32_t*)malloc(sizeof(uint32_t) * (tot_nodes + 1)); uint64_t* eo = (uint64_t*)malloc(sizeof(uint64_t) * ((size_t)n + 1));
    int32_t* where = (int32_t*)malloc(sizeof(int32_t) * (g->n_nodes + 1ull));      /* window node -> sub-DAG node, per problem (reset after use) */
    if (!pr || !po || !pi || !nl || !sq || !rd || !en || !eo || !where) { free(pr); free(po); free(pi); free(nl); free(sq); free(rd); free(en); free(eo); free(where); return VGK_ENOMEM; }
    for (uint32_t v = 0; v <= g->n_nodes; ++v) where[v] = -2;                     /* -2: not reached; -1: reached but left out (the start, cut to nothing) */
    uint64_t at_o = 0, at_e = 0, at_n = 0, at_s = 0, at_q = 0, at_x = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const vgk_extension_problem* p = &problems[i];
        const uint32_t a = p->first_node, b = p->first_node + p->n_nodes, s = p->start_node;
        const int left = p->leftward != 0;
        const uint32_t qlen = left ? p->query_offset : p->read_len - p->query_offset;
        /* the read part on the extension's side, in extension order */
        for (uint32_t r = 0; r < qlen; ++r) rd[at_q + r] = left ? reads[p->read_off + p->query_offset - 1 - r] : reads[p->read_off + p->query_offset + r];
        vgk_gssw_problem* q = &pr[i];
        q->read = rd + at_q; q->read_len = qlen; q->flags = p->flags; q->max_gap_length = p->max_gap_length;
        q->graph.node_len = nl + at_n; q->graph.seq = sq + at_s; q->graph.pred_off = po + at_o; q->graph.pred_idx = pi + at_e;
        uint32_t cnt = 0, ne = 0; uint64_t bases = 0;
        eo[i] = at_x;
        /* the window's nodes in extension order from the start; the edges that lead back towards it */
        for (uint32_t step = 0; left ? s >= a + step : s + step < b; ++step) {
            const uint32_t v = left ? s - step : s + step;
            uint32_t len = g->node_len[v]; const char* src = g->seq + g->seq_off[v];
            int reached = v == s;
            if (v == s) { if (left) len = p->start_offset; else { src += p->start_offset; len -= p->start_offset; } }
            else {
                /* neighbours on the start's side: predecessors (rightward) or successors (leftward: found by scanning the later nodes' lists) */
                if (!left) { for (uint32_t e = g->pred_off[v]; e < g->pred_off[v + 1]; ++e) { const uint32_t u = g->pred_idx[e]; if (u >= s && u < b && where[u] != -2) reached = 1; } }
                else for (uint32_t u = v + 1; u <= s; ++u) { if (where[u] == -2) continue; for (uint32_t e = g->pred_off[u]; e < g->pred_off[u + 1]; ++e) if (g->pred_idx[e] == v) reached = 1; }
            }
            if (!reached) continue;
            if (v == s && len == 0) { where[v] = -1; continue; }                  /* pinned exactly at the node's end: its neighbours start from the root */
            where[v] = (int32_t)cnt;
            po[at_o + cnt] = ne;
            if (v != s) {
                if (!left) { for (uint32_t e = g->pred_off[v]; e < g->pred_off[v + 1]; ++e) { const uint32_t u = g->pred_idx[e]; if (u >= s && u < b && where[u] >= 0) pi[at_e + ne++] = (uint32_t)where[u]; } }
                else for (uint32_t u = v + 1; u <= s; ++u) { if (where[u] < 0) continue; for (uint32_t e = g->pred_off[u]; e < g->pred_off[u + 1]; ++e) if (g->pred_idx[e] == v) pi[at_e + ne++] = (uint32_t)where[u]; }
            }
            nl[at_n + cnt] = len;
            for (uint32_t k = 0; k < len; ++k) sq[at_s + bases + k] = left ? src[len - 1 - k] : src[k];
            bases += len; en[at_x + cnt] = v - a; ++cnt;
        }
        for (uint32_t v = a; v < b; ++v) where[v] = -2;
        if (cnt == 0) {                                                           /* nothing lies that way: a placeholder the fetch drops */
            po[at_o] = 0; nl[at_n] = 1; sq[at_s] = 'N'; cnt = 1; bases = 1;
            po[at_o + 1] = 0; q->graph.n_nodes = 1;
            at_o += 2; at_n += 1; at_s += 1; at_q += qlen;                        /* (eo[i + 1] == eo[i]) */
            continue;
        }
        po[at_o + cnt] = ne; q->graph.n_nodes = cnt;
        at_o += cnt + 1; at_e += ne; at_n += cnt; at_s += bases; at_q += qlen; at_x += cnt;
    }
    eo[n] = at_x;
    free(where);
    vgk_batch* bt = NULL;
    int rc = vgk_gssw_pack(ctx, pr, n, ops_per_problem, &bt);
    if (rc) { free(pr); free(po); free(pi); free(nl); free(sq); free(rd); free(en); free(eo); return rc; }
    bt->own_probs = pr; bt->own_reads = rd; bt->own_pred_off = po; bt->own_pred_idx = pi; bt->own_node_len = nl; bt->own_seq = sq; bt->ext_node = en; bt->ext_off = eo;
    *out = bt; return VGK_OK;
}
int  vgk_batch_sync(vgk_batch* b) { (void)b; return VGK_OK; }
double vgk_batch_kernel_ms(vgk_batch* b, int which) { (void)b; (void)which; return 0.0; }
uint64_t vgk_batch_cells(vgk_batch* b) { return b ? b->cells : 0; }
uint64_t vgk_batch_alg_bytes(vgk_batch* b) { (void)b; return 0; }
uint64_t vgk_batch_device_bytes(vgk_batch* b) { (void)b; return 0; }
uint64_t vgk_batch_wave_steps(vgk_batch* b) { (void)b; return 0; }
int vgk_batch_lane(vgk_batch* b) { (void)b; return 0; }
double vgk_gssw_wide_last(vgk_ctx* ctx, int which) { (void)ctx; (void)which; return 0.0; }      /* (one scalar route for every read length here) */
/* the engine's speculative fill has no counterpart here (one scalar fill with its traceback): the checker never speculates */
int vgk_batch_speculated(vgk_batch* b) { (void)b; return 0; }
int vgk_set_speculation(vgk_ctx* ctx, int mode) { (void)ctx; return (mode < 0 || mode > 2) ? VGK_EINVAL : VGK_OK; }
int vgk_speculation_state(vgk_ctx* ctx, uint64_t counters[4], double* last_miss_share) {
    (void)ctx; if (counters) counters[0] = counters[1] = counters[2] = counters[3] = 0; if (last_miss_share) *last_miss_share = 0.0; return 0; }

/* the oracle keeps no sets between calls (every entry point takes and returns host arrays): the table over host arrays that the engine's
 * vgk_rescue_requests is checked against is vg_amd/host/rescue_requests.cpp */
int vgk_rescue_requests(vgk_ctx* ctx, const vgk_dgraph* graph, double fragment_mean, double fragment_sd, double rescue_stdevs, vgk_rescue_request* requests, size_t cap, size_t* written) {
    (void)ctx; (void)graph; (void)fragment_mean; (void)fragment_sd; (void)rescue_stdevs; (void)requests; (void)cap; if (written) *written = 0; return VGK_EUNSUPPORTED; }

/* ---- tail forests (vgo_tail.c) behind the engine's entry points: the walks, then the forest as one graph through the oracle's
 * own vgk_graph_create (node lengths, the bases behind the cuts copied out of the index, one predecessor per non-root node) ---- */
int vgo_tail_forest(const vgk_haplo* h, const vgk_tail_problem* pb, vgk_tail_result* out, int32_t** parent, uint32_t** node, uint32_t** len, size_t* n, size_t* cap);
void vgo_tail_copy_bases(const vgk_haplo* h, uint32_t node, uint32_t trim, uint32_t len, char* dst);
struct vgk_forest { size_t n; int32_t* parent; uint32_t* node; uint32_t* len; vgk_dgraph* graph; };
void vgk_forest_destroy(vgk_forest* f) { if (f) { free(f->parent); free(f->node); free(f->len); vgk_graph_destroy(f->graph); free(f); } }
int vgk_tail_forest(vgk_ctx* ctx, const vgk_haplo* index, const vgk_tail_problem* problems, uint32_t n, vgk_tail_result* results, vgk_forest** out) {
    if (!ctx || !index || !out || (n && (!problems || !results))) return VGK_EINVAL;
    *out = NULL;
    vgk_forest* f = (vgk_forest*)calloc(1, sizeof *f);
    if (!f) return VGK_ENOMEM;
    size_t cap = 0;
    uint32_t* trim = NULL; size_t trim_cap = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const size_t before = f->n;
        const int rc = vgo_tail_forest(index, &problems[i], &results[i], &f->parent, &f->node, &f->len, &f->n, &cap);
        if (rc) { free(trim); vgk_forest_destroy(f); return rc; }
        if (f->n > trim_cap) { trim_cap = 2 * f->n + 64; uint32_t* q = (uint32_t*)realloc(trim, sizeof(uint32_t) * trim_cap); if (!q) { free(trim); vgk_forest_destroy(f); return VGK_ENOMEM; } trim = q; }
        for (size_t v = before; v < f->n; ++v) trim[v] = 0;
        /* the root of the walk (if it was entered: it is the problem's first tree node and no other root exists then) carries the cut */
        if (f->n > before && results[i].root_trim) trim[before] = results[i].root_trim;
    }
    if (f->n) {
        size_t bases = 0, ne = 0;
        for (size_t v = 0; v < f->n; ++v) { bases += f->len[v]; ne += f->parent[v] >= 0; }
        char* seq = (char*)malloc(bases + 1); uint32_t* po = (uint32_t*)malloc(sizeof(uint32_t) * (f->n + 1)); uint32_t* pi = (uint32_t*)malloc(sizeof(uint32_t) * (ne + 1));
        if (!seq || !po || !pi) { free(seq); free(po); free(pi); free(trim); vgk_forest_destroy(f); return VGK_ENOMEM; }
        size_t at = 0, e = 0;
        for (size_t v = 0; v < f->n; ++v) {
            vgo_tail_copy_bases(index, f->node[v], trim[v], f->len[v], seq + at); at += f->len[v];
            po[v] = (uint32_t)e;
            if (f->parent[v] >= 0) pi[e++] = (uint32_t)f->parent[v];
        }
        po[f->n] = (uint32_t)e;
        vgk_graph g; g.n_nodes = (uint32_t)f->n; g.node_len = f->len; g.seq = seq; g.pred_off = po; g.pred_idx = pi;
        const int rc = vgk_graph_create(ctx, &g, &f->graph);
        free(seq); free(po); free(pi);
        if (rc) { free(trim); vgk_forest_destroy(f); return rc; }
    }
    free(trim);
    *out = f;
    return VGK_OK;
}
uint64_t vgk_forest_size(const vgk_forest* f) { return f ? f->n : 0; }
int vgk_forest_fetch(const vgk_forest* f, int32_t* parent, uint32_t* node, uint32_t* length) {
    if (!f) return VGK_EINVAL;
    if (parent) memcpy(parent, f->parent, sizeof(int32_t) * f->n);
    if (node) memcpy(node, f->node, sizeof(uint32_t) * f->n);
    if (length) memcpy(length, f->len, sizeof(uint32_t) * f->n);
    return VGK_OK;
}
const vgk_dgraph* vgk_forest_graph(const vgk_forest* f) { return f ? f->graph : NULL; }
double vgk_tail_last_ms(vgk_ctx* ctx) { (void)ctx; return 0.0; }

/* reading a GBWT file is the engine's; the independent decoder the tests hold against it is tests/golden/extract_primers_fixture.py,
 * and the oracle's index is built from the threads that script extracts */
int vgk_haplo_create_gbwt(vgk_ctx* ctx, const void* gbwt, size_t bytes, uint32_t n_nodes, const uint32_t* node_len, const char* seq, vgk_haplo** out) {
    (void)ctx; (void)gbwt; (void)bytes; (void)n_nodes; (void)node_len; (void)seq; if (out) *out = NULL;
    return VGK_EUNSUPPORTED;
}

int vgk_gapless_fetch_deferred(vgk_ctx* ctx) { (void)ctx; return VGK_OK; }
int vgk_gbz_load(const void* gbz, size_t bytes, vgk_haplotypes** out) { (void)gbz; (void)bytes; if (out) *out = NULL; return VGK_EUNSUPPORTED; }     /* file formats are the engine's */
void vgk_haplotypes_free(vgk_haplotypes* h) { (void)h; }

/* the oracle keeps nothing between calls: the seeded form is not available on it (callers use vgk_gapless_extend) */
int vgk_gapless_extend_seeded(vgk_ctx* ctx, const vgk_haplo* index, uint32_t max_mismatches, double overlap_threshold, uint32_t flags,
                              vgk_gapless_result* results, vgk_extension* extensions, size_t ext_cap,
                              uint32_t* nodes, size_t nodes_cap, uint32_t* mismatches, size_t mism_cap, size_t written[3]) {
    (void)ctx; (void)index; (void)max_mismatches; (void)overlap_threshold; (void)flags; (void)results; (void)extensions; (void)ext_cap;
    (void)nodes; (void)nodes_cap; (void)mismatches; (void)mism_cap; (void)written;
    return VGK_EUNSUPPORTED;
}

int vgk_tail_stage(vgk_ctx* ctx, const vgk_haplo* index, uint32_t ops_per_problem, int32_t* ext_total, size_t ext_cap, int32_t* read_score, uint64_t stats[4]) {
    (void)ctx; (void)index; (void)ops_per_problem; (void)ext_total; (void)ext_cap; (void)read_score; (void)stats;
    return VGK_EUNSUPPORTED;                                             /* the oracle keeps nothing between calls: tests use the host-side stage */
}
int vgk_tail_stage_aligned(vgk_ctx* ctx, const vgk_haplo* index, uint32_t ops_per_problem, int32_t* ext_total, size_t ext_cap, int32_t* read_score,
                           vgk_tail_alignment* tails, size_t tails_cap, vgk_op* ops, size_t ops_cap, size_t written[2], uint64_t stats[4]) {
    (void)ctx; (void)index; (void)ops_per_problem; (void)ext_total; (void)ext_cap; (void)read_score; (void)tails; (void)tails_cap; (void)ops; (void)ops_cap; (void)written; (void)stats;
    return VGK_EUNSUPPORTED;                                             /* as vgk_tail_stage: the tests assemble the stage from the oracle's entry points */
}
double vgk_tail_stage_last_ms(vgk_ctx* ctx, int which) { (void)ctx; (void)which; return 0.0; }
