/*
 * vgo_haplo.h — the haplotype index of the CPU ORACLE (shared by vgo_gapless.c and vgo_wfa.c).
 * TEST INFRASTRUCTURE ONLY: never included, linked or loaded by the product path.
 * The index restates the published GBWT design (Siren et al. 2020) [prior knowledge]; see vgo_gapless.c.
 */
#ifndef VGO_HAPLO_H
#define VGO_HAPLO_H
#include <stdint.h>
#include <stddef.h>
#include "../include/vgk.h"

typedef struct { int32_t node; int32_t lo, hi; } SState;          /* visits [lo, hi] of an oriented node; empty when lo > hi */
typedef struct { SState f, b; } BState;

struct vgk_haplo {
    uint32_t n_nodes, n_oriented;
    uint32_t* len;            /* per oriented node */
    size_t*   seq_off;        /* per oriented node, into seq */
    char*     seq;            /* forward strands then reverse complements */
    uint32_t* count;          /* visits per oriented node */
    uint32_t* edge_off;       /* per oriented node, n_oriented + 1 */
    int32_t*  edge_to;        /* successor (oriented node) or -1 = thread ends here; ascending */
    uint32_t* edge_base;      /* where this node's visits start inside the successor's record */
    size_t*   body_off;       /* per oriented node, n_oriented + 1 */
    uint32_t* body;           /* per visit: index of its edge within the node's edge list */
};

#endif
