/*
 * qldpc_graph.h -- host-side Tanner graph of libqldpc (plain C; internal to the library).
 *
 * Orientation and edge orders follow what the reference harness gets from AFF3CT's
 * tools::Sparse_matrix (rows = variable nodes; BS/src/main.cpp:175-178): a check's edges are in
 * add_connection() order, a variable's message slots are in ascending-check order.
 */
#ifndef QLDPC_GRAPH_H
#define QLDPC_GRAPH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

struct qldpc_code {
    int N, M, E;
    int *cn_ptr;     /* [M+1] CN-major ranges                                        */
    int *cn_var;     /* [E]   VN of CN-major edge k                                   */
    int *vn_ptr;     /* [N+1] VN-major ranges                                        */
    int *vn_chk;     /* [E]   check of VN-major slot                                  */
    int *transpose;  /* [E]   CN-major edge k -> VN-major slot                        */
    int max_dc, max_dv;
    int ira_K;       /* > 0: VNs ira_K..N-1 are a dual-diagonal accumulator chain    */
    /* horizontal-layered execution order: layers of mutually VN-disjoint checks     */
    int n_layers;
    int *layer_ptr;   /* [n_layers+1] into layer_order                                */
    int *layer_order; /* [M] checks in execution order                                */
    int layer_natural;/* 1: order is equivalent to c = 0..M-1 (level schedule)        */
};

void qldpc_set_error(const char *fmt, ...);

/* GF(2) systematic form: returns rank r; pivots[r] parity positions (ascending pivot search),
 * free_pos[N-r] info positions, A[r][ceil((N-r)/64)] bit rows with x_pivot[j] = <A[j], x_free>. */
int qldpc_gf2_systematic(const struct qldpc_code *code, int **pivots, int **free_pos, uint64_t **A, int *words_per_row);
int qldpc_gf2_systematic_ord(const struct qldpc_code *code, int order, int **pivots, int **free_pos, uint64_t **A, int *words_per_row);

#ifdef __cplusplus
}
#endif
#endif
