/*
 * qldpc_oracle.c -- CPU ORACLE (test infrastructure only; see qldpc_oracle.h).
 *
 * Restates, in scalar float32 and in the same operation order, the AFF3CT v2.3.5 decoder that
 * the reference harness instantiates (AFF3CT itself is an un-vendored third-party dependency:
 * MP/lib/.gitkeep, MP/ci/build-linux-macos.sh:50,56).  Each function cites the reference call
 * site whose behaviour it follows and the AFF3CT module it restates.
 */
#include "qldpc_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

struct orc_graph {
    int N, M, E;
    int *vn_ptr;    /* [N+1] VN-major edge ranges (AFF3CT H.get_row_to_cols())                */
    int *vn_chk;    /* [E]   check of each VN-major edge, in AFF3CT branch order               */
    int *cn_ptr;    /* [M+1] CN-major edge ranges (AFF3CT H.get_col_to_rows())                */
    int *cn_var;    /* [E]   VN of each CN-major edge, in add_connection() order              */
    int *transpose; /* [E]   CN-major edge k -> VN-major edge id (Decoder_LDPC_BP_flooding ctor) */
    int max_dc, max_dv;
};

/* per-thread decode context: target syndrome of the frame being decoded (coset / syndrome form) or NULL */
static const int *g_synd = 0;
#pragma omp threadprivate(g_synd)

/* ------------------------------------------------------------------ graph ---------------- */

/*
 * AFF3CT tools::Sparse_matrix::add_connection(row=VN, col=CN) appends CN to row_to_cols[VN] and
 * VN to col_to_rows[CN]; H is held with rows = variable nodes (BS/src/main.cpp:178 uses
 * H.get_cols_max_degree() as the max CN degree).  The flooding decoder's `transpose` gives the
 * k-th CN-major edge the next free VN-major slot of its variable node, visiting CNs in ascending
 * order -- so a VN's slots are in ascending-check order of that traversal.
 */
orc_graph *orc_graph_from_edges(int N, int M, int E, const int *var, const int *chk)
{
    if (N <= 0 || M <= 0 || E <= 0 || !var || !chk) return NULL;
    orc_graph *g = (orc_graph *)calloc(1, sizeof(*g));
    g->N = N; g->M = M; g->E = E;
    g->vn_ptr = (int *)calloc((size_t)N + 1, sizeof(int));
    g->cn_ptr = (int *)calloc((size_t)M + 1, sizeof(int));
    g->vn_chk = (int *)malloc((size_t)E * sizeof(int));
    g->cn_var = (int *)malloc((size_t)E * sizeof(int));
    g->transpose = (int *)malloc((size_t)E * sizeof(int));
    for (int e = 0; e < E; e++) {
        if (var[e] < 0 || var[e] >= N || chk[e] < 0 || chk[e] >= M) { orc_graph_free(g); return NULL; }
        g->vn_ptr[var[e] + 1]++;
        g->cn_ptr[chk[e] + 1]++;
    }
    for (int v = 0; v < N; v++) { if (g->vn_ptr[v + 1] > g->max_dv) g->max_dv = g->vn_ptr[v + 1]; g->vn_ptr[v + 1] += g->vn_ptr[v]; }
    for (int c = 0; c < M; c++) { if (g->cn_ptr[c + 1] > g->max_dc) g->max_dc = g->cn_ptr[c + 1]; g->cn_ptr[c + 1] += g->cn_ptr[c]; }
    int *fill = (int *)calloc((size_t)M, sizeof(int));
    for (int e = 0; e < E; e++) { int c = chk[e]; g->cn_var[g->cn_ptr[c] + fill[c]++] = var[e]; }
    free(fill);
    int *conn = (int *)calloc((size_t)N, sizeof(int));
    for (int c = 0, k = 0; c < M; c++)
        for (int j = g->cn_ptr[c]; j < g->cn_ptr[c + 1]; j++, k++) {
            int v = g->cn_var[j];
            int slot = g->vn_ptr[v] + conn[v]++;
            g->transpose[k] = slot;
            g->vn_chk[slot] = c;
        }
    free(conn);
    return g;
}

void orc_graph_free(orc_graph *g)
{
    if (!g) return;
    free(g->vn_ptr); free(g->vn_chk); free(g->cn_ptr); free(g->cn_var); free(g->transpose); free(g);
}
int orc_graph_N(const orc_graph *g) { return g->N; }
int orc_graph_M(const orc_graph *g) { return g->M; }
int orc_graph_E(const orc_graph *g) { return g->E; }
int orc_graph_max_cn_degree(const orc_graph *g) { return g->max_dc; }
int orc_graph_max_vn_degree(const orc_graph *g) { return g->max_dv; }
void orc_graph_export(const orc_graph *g, int *cn_ptr, int *cn_var, int *vn_ptr, int *vn_chk, int *transpose)
{
    if (cn_ptr) memcpy(cn_ptr, g->cn_ptr, sizeof(int) * ((size_t)g->M + 1));
    if (cn_var) memcpy(cn_var, g->cn_var, sizeof(int) * (size_t)g->E);
    if (vn_ptr) memcpy(vn_ptr, g->vn_ptr, sizeof(int) * ((size_t)g->N + 1));
    if (vn_chk) memcpy(vn_chk, g->vn_chk, sizeof(int) * (size_t)g->E);
    if (transpose) memcpy(transpose, g->transpose, sizeof(int) * (size_t)g->E);
}

/*
 * MacKay alist as read by tools::LDPC_matrix_handler::read (VAR/main.cpp (alist-v1.0.1):324,338):
 * line 1 "N M" (N listed first = variable nodes), line 2 max degrees, line 3 VN degrees, line 4 CN
 * degrees, then N per-VN lists (1-based check ids, 0 = padding) which drive add_connection(), then
 * M per-CN lists (only cross-checked).
 */
orc_graph *orc_graph_from_alist(const char *path)
{
    FILE *f = fopen(path, "r");
    if (!f) return NULL;
    int N, M, dvmax, dcmax;
    if (fscanf(f, "%d %d %d %d", &N, &M, &dvmax, &dcmax) != 4 || N <= 0 || M <= 0) { fclose(f); return NULL; }
    int *dv = (int *)malloc(sizeof(int) * (size_t)N), *dc = (int *)malloc(sizeof(int) * (size_t)M);
    long E = 0;
    for (int v = 0; v < N; v++) { if (fscanf(f, "%d", &dv[v]) != 1) goto bad; E += dv[v]; }
    for (int c = 0; c < M; c++) { if (fscanf(f, "%d", &dc[c]) != 1) goto bad; }
    {
        int *var = (int *)malloc(sizeof(int) * (size_t)E), *chk = (int *)malloc(sizeof(int) * (size_t)E);
        long e = 0;
        int ok = 1;
        /* Each VN line holds dv[v] entries, optionally zero-padded up to dvmax on the same line.
         * Read line-wise so both padded and unpadded files parse. */
        int ch;
        while ((ch = fgetc(f)) != EOF && ch != '\n') {}
        char *line = NULL; size_t cap = 0;
        for (int v = 0; v < N && ok; v++) {
            if (getline(&line, &cap, f) < 0) { ok = 0; break; }
            char *p = line; int got = 0;
            for (;;) {
                char *end; long x = strtol(p, &end, 10);
                if (end == p) break;
                p = end;
                if (x > 0) { if (e >= E || x > M) { ok = 0; break; } var[e] = v; chk[e] = (int)x - 1; e++; got++; }
            }
            if (got != dv[v]) ok = 0;
        }
        /* cross-check the CN part */
        long e2 = 0;
        for (int c = 0; c < M && ok; c++) {
            if (getline(&line, &cap, f) < 0) { ok = 0; break; }
            char *p = line; int got = 0;
            for (;;) { char *end; long x = strtol(p, &end, 10); if (end == p) break; p = end; if (x > 0) { got++; e2++; } }
            if (got != dc[c]) ok = 0;
        }
        free(line);
        orc_graph *g = (ok && e == E && e2 == E) ? orc_graph_from_edges(N, M, (int)E, var, chk) : NULL;
        free(var); free(chk); free(dv); free(dc); fclose(f);
        return g;
    }
bad:
    free(dv); free(dc); fclose(f);
    return NULL;
}

/*
 * AFF3CT .qc (EC/ldpc_examples/README.md:7, EC/README_LDPC.md:494-498): header "cols rows Z",
 * then `rows` lines of `cols` entries; -1 = zero block, s >= 0 = ZxZ identity right-shifted by s
 * (block row r has its 1 in block column (r + s) mod Z -- ML/mul_sh.m:6-10, ML/check_cword.m).
 * Connections are added block by block, row-block major, z ascending.
 */
orc_graph *orc_graph_from_qc(const char *path)
{
    FILE *f = fopen(path, "r");
    if (!f) return NULL;
    int nb, mb, Z;
    if (fscanf(f, "%d %d %d", &nb, &mb, &Z) != 3 || nb <= 0 || mb <= 0 || Z <= 0) { fclose(f); return NULL; }
    int *B = (int *)malloc(sizeof(int) * (size_t)nb * mb);
    long blocks = 0;
    for (int i = 0; i < mb * nb; i++) { if (fscanf(f, "%d", &B[i]) != 1) { free(B); fclose(f); return NULL; } if (B[i] >= 0) blocks++; }
    fclose(f);
    long E = blocks * Z;
    int *var = (int *)malloc(sizeof(int) * (size_t)E), *chk = (int *)malloc(sizeof(int) * (size_t)E);
    long e = 0;
    for (int i = 0; i < mb; i++)
        for (int j = 0; j < nb; j++) {
            int s = B[i * nb + j];
            if (s < 0) continue;
            for (int z = 0; z < Z; z++) { chk[e] = i * Z + z; var[e] = j * Z + (z + s) % Z; e++; }
        }
    orc_graph *g = orc_graph_from_edges(nb * Z, mb * Z, (int)E, var, chk);
    free(var); free(chk); free(B);
    return g;
}

int orc_syndrome(const orc_graph *g, const int *x, int *s)
{
    int w = 0;
    for (int c = 0; c < g->M; c++) {
        int p = 0;
        for (int j = g->cn_ptr[c]; j < g->cn_ptr[c + 1]; j++) p ^= (x[g->cn_var[j]] & 1);
        if (s) s[c] = p;
        w += p;
    }
    return w;
}

/* ------------------------------------------------------------- update rules -------------- */

static inline int sgnbit(float x) { return signbit(x) ? -1 : 0; }   /* AFF3CT: std::signbit(v) ? -1 : 0 */
static inline float csign(float mag, int s) { return copysignf(mag, s ? -1.0f : 1.0f); }

/* tools::min_star / min_star_linear2 (AFF3CT Tools/Math/utils.h) -- parity unpinned */
/* std::min / std::max semantics: min(a,b) = (b < a) ? b : a ; max(a,b) = (a < b) ? b : a */
static inline float f_min(float a, float b) { return (b < a) ? b : a; }
static inline float f_max(float a, float b) { return (a < b) ? b : a; }
static inline float f_min_star(float a, float b)
{
    return (a < b ? a : b) + logf(1.0f + expf(-(a + b))) - logf(1.0f + expf(-fabsf(a - b)));
}
static inline float corr_l2(float x) { float t = 0.6f - 0.24f * fabsf(x); return t > 0.0f ? t : 0.0f; }
static inline float f_min_star_l2(float a, float b)
{
    float r = (a < b ? a : b) + corr_l2(a + b) - corr_l2(a - b);
    return r > 0.0f ? r : 0.0f;
}

/*
 * One check node: in[0..deg) are the incoming var->chk messages in col_to_rows order, out[] the
 * outgoing chk->var messages.  Restates tools::Update_rule_{MS,OMS,NMS,SPA,LSPA,AMS}
 * (begin_chk_node_in / compute_chk_node_in xdeg / end_chk_node_in / compute_chk_node_out xdeg),
 * selected at VAR/main.cpp (alist-v1.0.1):203-218 and BS/src/main.cpp:193.
 */
static int g_cn_sign0 = 0;         /* initial sign of the fold: -1 when the check's target parity is 1 */
#pragma omp threadprivate(g_cn_sign0)
static void cn_update(int rule, float param, int deg, const float *in, float *out, float *scratch)
{
    int sign = g_cn_sign0;
    switch (rule) {
    case ORC_RULE_MS:
    case ORC_RULE_OMS:
    case ORC_RULE_NMS: {
        float min1 = FLT_MAX, min2 = FLT_MAX;
        for (int i = 0; i < deg; i++) {
            const float a = fabsf(in[i]);
            sign ^= sgnbit(in[i]);
            min2 = f_min(min2, f_max(a, min1));   /* min2 = min(min2, max(var_abs, min1)) */
            min1 = f_min(min1, a);                /* min1 = min(min1, var_abs)            */
        }
        float cst1, cst2;
        if (rule == ORC_RULE_MS)       { cst1 = f_max(0.0f, min2);         cst2 = f_max(0.0f, min1); }
        else if (rule == ORC_RULE_OMS) { cst1 = f_max(0.0f, min2 - param); cst2 = f_max(0.0f, min1 - param); }
        else                           { cst1 = min2 * param;              cst2 = min1 * param; }
        for (int i = 0; i < deg; i++) {
            const float a = fabsf(in[i]);
            const float r = (a == min1) ? cst1 : cst2;
            out[i] = csign(r, sign ^ sgnbit(in[i]));
        }
        break;
    }
    case ORC_RULE_SPA: {
        float product = 1.0f;
        for (int i = 0; i < deg; i++) {
            const float a = fabsf(in[i]);
            const float r = tanhf(a * 0.5f);
            sign ^= sgnbit(in[i]);
            product *= r;
            scratch[i] = r;
        }
        for (int i = 0; i < deg; i++) {
            float t = product / scratch[i];
            t = (t < 1.0f) ? t : 1.0f - FLT_EPSILON;
            const float r = 2.0f * atanhf(t);
            out[i] = csign(r, sign ^ sgnbit(in[i]));
        }
        break;
    }
    case ORC_RULE_LSPA: {
        float sum = 0.0f;
        for (int i = 0; i < deg; i++) {
            const float a = fabsf(in[i]);
            const float t = tanhf(a * 0.5f);
            const float r = (t != 0.0f) ? logf(t) : FLT_MIN;
            sign ^= sgnbit(in[i]);
            sum += r;
            scratch[i] = r;
        }
        for (int i = 0; i < deg; i++) {
            float t = sum - scratch[i];
            t = (t != 0.0f) ? expf(t) : 1.0f - FLT_EPSILON;
            const float r = 2.0f * atanhf(t);
            out[i] = csign(r, sign ^ sgnbit(in[i]));
        }
        break;
    }
    default: { /* AMS<MIN>: delta = MIN over all |in|, delta_min = MIN over all but the minimum */
        float min = FLT_MAX, delta_min = FLT_MAX;
        for (int i = 0; i < deg; i++) {
            const float a = fabsf(in[i]);
            sign ^= sgnbit(in[i]);
            float other;
            if (a < min) { other = min; min = a; } else other = a;
            if (rule == ORC_RULE_AMS_MIN) delta_min = f_min(delta_min, other);
            else if (rule == ORC_RULE_AMS_MINSTAR) delta_min = f_min_star(delta_min, other);
            else delta_min = f_min_star_l2(delta_min, other);
        }
        float delta;
        if (rule == ORC_RULE_AMS_MIN) delta = f_min(delta_min, min);
        else if (rule == ORC_RULE_AMS_MINSTAR) delta = f_min_star(delta_min, min);
        else delta = f_min_star_l2(delta_min, min);
        delta = f_max(0.0f, delta);
        delta_min = f_max(0.0f, delta_min);
        for (int i = 0; i < deg; i++) {
            const float a = fabsf(in[i]);
            const float r = (a == min) ? delta_min : delta;
            out[i] = csign(r, sign ^ sgnbit(in[i]));
        }
        break;
    }
    }
}

/* ------------------------------------------------------------- syndrome ------------------ */

/* Decoder_LDPC_BP::check_syndrome_soft: XOR of signbit(post) over each check's VNs. */
static int syndrome_is_zero(const orc_graph *g, const float *post)
{
    for (int c = 0; c < g->M; c++) {
        int s = (g_synd && g_synd[c]) ? -1 : 0;
        for (int j = g->cn_ptr[c]; j < g->cn_ptr[c + 1]; j++) s ^= sgnbit(post[g->cn_var[j]]);
        if (s) return 0;
    }
    return 1;
}

/* ------------------------------------------------------------- fp16 message storage ------ */

/* round-to-nearest-even float -> IEEE binary16 -> float, in integer arithmetic (what the GPU's __float2half_rn /
 * __half2float pair does); used when the decoder under test stores its messages as binary16 */
static int g_msg_fp16 = 0;
#pragma omp threadprivate(g_msg_fp16)
static float rne16(float f)
{
    union { float f; unsigned u; } v = { f };
    const unsigned sign = v.u & 0x80000000u;
    unsigned a = v.u & 0x7fffffffu;
    if (a >= 0x7f800000u) return f;                                  /* inf / nan */
    if (a >= 0x477ff000u) { v.u = sign | 0x7f800000u; return v.f; }  /* >= 65520 rounds to inf */
    if (a < 0x33000001u) { v.u = sign; return v.f; }                 /* < 2^-25 (or = 2^-25, tie to even 0) rounds to 0 */
    unsigned h;
    if (a < 0x38800000u) {                                           /* binary16 subnormal: quantum 2^-24 */
        const int shift = 126 - (int)(a >> 23);                      /* 14 .. 24 */
        const unsigned m = (a & 0x7fffffu) | 0x800000u;
        const unsigned q = m >> shift, rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
        h = q + ((rem > half) || (rem == half && (q & 1u)));
        /* back to float: h * 2^-24 */
        v.f = (float)h * 5.9604644775390625e-08f;
        v.u |= sign;
        return v.f;
    }
    const unsigned rem = a & 0x1fffu, q = a >> 13;
    a = (q + ((rem > 0x1000u) || (rem == 0x1000u && (q & 1u)))) << 13;
    v.u = sign | a;
    return v.f;
}
static inline float stq(float x) { return g_msg_fp16 ? rne16(x) : x; }
float orc_round_fp16(float x) { return rne16(x); }      /* exported so the tests can pin it against numpy's float16 */

/* ------------------------------------------------------------- flooding ------------------ */

/*
 * module::Decoder_LDPC_BP_flooding<B,Q,Rule>::_decode_siho for one frame after reset()
 * (BS/src/main.cpp:365,389).  Per iteration:
 *   _initialize_var_to_chk : tmp = Y[v] + (sum of chk_to_var over the VN's slots, in slot order,
 *                            starting from 0);  var_to_chk[slot] = tmp - chk_to_var[slot]
 *   _decode_single_ite     : per CN in order, gather via transpose, rule in/out, scatter
 *   if enable_syndrome && ite != n_ite-1: post = Y + sum; break when the syndrome has been zero
 *                            `syndrome_depth` consecutive times
 * After the loop (no break) post is computed once more.  V = !(post >= 0).
 */
static int decode_flooding(const orc_graph *g, int rule, float param, int n_ite, int enable_syndrome,
                           int syndrome_depth, const float *Y, float *post, float *c2v, float *v2c,
                           float *in, float *out, float *scratch)
{
    const int N = g->N, M = g->M, E = g->E;
    memset(c2v, 0, sizeof(float) * (size_t)E);   /* _load() with init_flag set by reset() */
    int cur_depth = 0, ite = 0;
    for (; ite < n_ite; ite++) {
        for (int v = 0; v < N; v++) {
            float sum = 0.0f;
            for (int e = g->vn_ptr[v]; e < g->vn_ptr[v + 1]; e++) sum += c2v[e];
            const float tmp = Y[v] + sum;
            for (int e = g->vn_ptr[v]; e < g->vn_ptr[v + 1]; e++) v2c[e] = stq(tmp - c2v[e]);
        }
        for (int c = 0; c < M; c++) {
            const int b = g->cn_ptr[c], deg = g->cn_ptr[c + 1] - b;
            for (int i = 0; i < deg; i++) in[i] = v2c[g->transpose[b + i]];
            g_cn_sign0 = (g_synd && g_synd[c]) ? -1 : 0;
            cn_update(rule, param, deg, in, out, scratch);
            for (int i = 0; i < deg; i++) c2v[g->transpose[b + i]] = stq(out[i]);
        }
        if (enable_syndrome && ite != n_ite - 1) {
            for (int v = 0; v < N; v++) {
                float sum = 0.0f;
                for (int e = g->vn_ptr[v]; e < g->vn_ptr[v + 1]; e++) sum += c2v[e];
                post[v] = Y[v] + sum;
            }
            const int z = syndrome_is_zero(g, post);
            cur_depth = z ? (cur_depth + 1) % syndrome_depth : 0;
            if (z && cur_depth == 0) { ite++; return ite; }
        }
    }
    for (int v = 0; v < N; v++) {
        float sum = 0.0f;
        for (int e = g->vn_ptr[v]; e < g->vn_ptr[v + 1]; e++) sum += c2v[e];
        post[v] = Y[v] + sum;
    }
    return ite;
}

/* ------------------------------------------------------------- horizontal layered -------- */

/*
 * module::Decoder_LDPC_BP_horizontal_layered<B,Q,Rule> (VAR/main.cpp (alist-v1.0.1):223-237):
 * var_nodes = Y; messages = 0; per iteration, per CN in order:
 *   contributions[i] = var_nodes[v_i] - messages[kr++]; rule in;  messages[kw] = rule out;
 *   var_nodes[v_i] = contributions[i] + messages[kw++]
 * then check_syndrome_soft(var_nodes) after EVERY iteration (also the last one).
 * The same recursion as ML/BPSK_nrldpc_sim.m:29-69 (layer = one check row).
 */
static int decode_hlayered(const orc_graph *g, int rule, float param, int n_ite, int enable_syndrome,
                           int syndrome_depth, const float *Y, float *post, float *msg,
                           float *in, float *out, float *scratch)
{
    const int N = g->N, M = g->M, E = g->E;
    memcpy(post, Y, sizeof(float) * (size_t)N);
    memset(msg, 0, sizeof(float) * (size_t)E);
    int cur_depth = 0, ite = 0;
    for (; ite < n_ite; ite++) {
        for (int c = 0; c < M; c++) {
            const int b = g->cn_ptr[c], deg = g->cn_ptr[c + 1] - b;
            for (int i = 0; i < deg; i++) in[i] = post[g->cn_var[b + i]] - msg[b + i];
            g_cn_sign0 = (g_synd && g_synd[c]) ? -1 : 0;
            cn_update(rule, param, deg, in, out, scratch);
            for (int i = 0; i < deg; i++) { msg[b + i] = out[i]; post[g->cn_var[b + i]] = in[i] + out[i]; }
        }
        if (enable_syndrome) {
            const int z = syndrome_is_zero(g, post);
            cur_depth = z ? (cur_depth + 1) % syndrome_depth : 0;
            if (z && cur_depth == 0) { ite++; return ite; }
        }
    }
    return ite;
}

/* ------------------------------------------------------------- entry --------------------- */

int orc_decode(const orc_graph *g, int schedule, int rule, float rule_param, int n_ite,
               int enable_syndrome, int syndrome_depth, const float *Y_N, int n_frames,
               float *post_out, int *hard, int *iters, int *synd_ok, int n_threads)
{
    return orc_decode_coset(g, schedule, rule, rule_param, n_ite, enable_syndrome, syndrome_depth, Y_N, 0, n_frames, post_out, hard, iters, synd_ok, n_threads);
}

/* Coset ("syndrome form") decoding: each frame f must satisfy H x = target[f] instead of H x = 0; the fold of check c
 * starts with sign (-1)^target[f][c].  Not an AFF3CT mode: the oracle for the product's qldpc_load_syndrome_dev. */
int orc_decode_coset(const orc_graph *g, int schedule, int rule, float rule_param, int n_ite,
                     int enable_syndrome, int syndrome_depth, const float *Y_N, const int *target, int n_frames,
                     float *post_out, int *hard, int *iters, int *synd_ok, int n_threads)
{
    if (!g || !Y_N || n_frames < 0 || n_ite < 0) return -1;
    const int fp16 = (rule & ORC_MSG_FP16) != 0;      /* flag bit: messages stored as binary16 (flooding only) */
    rule &= ~ORC_MSG_FP16;
    if (rule < ORC_RULE_MS || rule > ORC_RULE_AMS_MINSTAR) return -2;
    if (fp16 && schedule != ORC_SCHED_FLOODING) return -4;
    if (schedule != ORC_SCHED_FLOODING && schedule != ORC_SCHED_HLAYERED) return -3;
    if (syndrome_depth < 1) syndrome_depth = 1;
    const int N = g->N, E = g->E;
#ifdef _OPENMP
    if (n_threads < 1) n_threads = 1;
#pragma omp parallel num_threads(n_threads)
#endif
    {
        float *post = (float *)malloc(sizeof(float) * (size_t)N);
        float *a = (float *)malloc(sizeof(float) * (size_t)E);
        float *b = (float *)malloc(sizeof(float) * (size_t)E);
        float *in = (float *)malloc(sizeof(float) * (size_t)(g->max_dc + 1) * 3);
        float *out = in + g->max_dc + 1, *scratch = out + g->max_dc + 1;
        int *hd = (int *)malloc(sizeof(int) * (size_t)N);
        g_msg_fp16 = fp16;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (int f = 0; f < n_frames; f++) {
            const float *Y = Y_N + (size_t)f * N;
            g_synd = target ? target + (size_t)f * g->M : 0;
            g_cn_sign0 = 0;
            int it;
            if (schedule == ORC_SCHED_FLOODING)
                it = decode_flooding(g, rule, rule_param, n_ite, enable_syndrome, syndrome_depth, Y, post, a, b, in, out, scratch);
            else
                it = decode_hlayered(g, rule, rule_param, n_ite, enable_syndrome, syndrome_depth, Y, post, a, in, out, scratch);
            for (int v = 0; v < N; v++) hd[v] = !(post[v] >= 0.0f);   /* V_K[i] = !(post[k] >= 0) */
            if (post_out) memcpy(post_out + (size_t)f * N, post, sizeof(float) * (size_t)N);
            if (hard) memcpy(hard + (size_t)f * N, hd, sizeof(int) * (size_t)N);
            if (iters) iters[f] = it;
            if (synd_ok) {
                int okf = 1;
                for (int c = 0; c < g->M && okf; c++) {
                    int pz = g_synd ? (g_synd[c] & 1) : 0;
                    for (int j = g->cn_ptr[c]; j < g->cn_ptr[c + 1]; j++) pz ^= hd[g->cn_var[j]] & 1;
                    if (pz) okf = 0;
                }
                synd_ok[f] = okf;
            }
        }
        free(post); free(a); free(b); free(in); free(hd);
    }
    return 0;
}

/* ------------------------------------------------------------- 8-bit fixed point --------- */

/*
 * Saturating fixed-point flooding min-sum, the integer decoder the product's msg_dtype = 2 kernels must equal bit for
 * bit.  Not an AFF3CT float mode (FER-tolerance class against it); the recipe -- quantise, subtract, saturate,
 * min1/min2/parity, offset or scale, saturate -- is the one of the reference's fixed-point MATLAB decoder
 * (ldpc_examples/.../BPSK_nrldpc_sim_RM_FP.m:37-98) applied to the flooding schedule above:
 *   Yq = clamp(rint(Y * scale), +-127); tmp = Yq + sum chk_to_var; var_to_chk = clamp(tmp - chk_to_var, +-127);
 *   |chk_to_var| = rule(min over the other edges), MS: m, OMS: max(0, m - rint(offset * scale)), NMS: (m * rint(factor * 128)) >> 7;
 *   decision = tmp < 0; early exit and iteration count exactly as decode_flooding.
 */
static int i8_quant(float y, float scale)
{
    const float t = y * scale;
    if (!(t < 127.0f)) return 127;
    if (t < -127.0f) return -127;
    return (int)lrintf(t);
}
static int i8_norm(int m, int rule, int p)
{
    if (rule == ORC_RULE_MS) return m;
    if (rule == ORC_RULE_OMS) return m > p ? m - p : 0;
    return (m * p) >> 7;
}
static int i8_syndrome_is_zero(const orc_graph *g, const int *post, const int *target)
{
    for (int c = 0; c < g->M; c++) {
        int s = target ? (target[c] & 1) : 0;
        for (int j = g->cn_ptr[c]; j < g->cn_ptr[c + 1]; j++) s ^= post[g->cn_var[j]] < 0;
        if (s) return 0;
    }
    return 1;
}
static int decode_flooding_i8(const orc_graph *g, int rule, int p, int n_ite, int enable_syndrome, int syndrome_depth,
                              const int *Yq, const int *target, int *post, int *c2v, int *v2c)
{
    const int N = g->N, M = g->M, E = g->E;
    memset(c2v, 0, sizeof(int) * (size_t)E);
    int cur_depth = 0, ite = 0;
    for (; ite < n_ite; ite++) {
        for (int v = 0; v < N; v++) {
            int sum = 0;
            for (int e = g->vn_ptr[v]; e < g->vn_ptr[v + 1]; e++) sum += c2v[e];
            const int tmp = Yq[v] + sum;
            for (int e = g->vn_ptr[v]; e < g->vn_ptr[v + 1]; e++) {
                int x = tmp - c2v[e];
                v2c[e] = x > 127 ? 127 : (x < -127 ? -127 : x);
            }
        }
        for (int c = 0; c < M; c++) {
            const int b = g->cn_ptr[c], deg = g->cn_ptr[c + 1] - b;
            int sign = target ? (target[c] & 1) : 0, min1 = 127, min2 = 127;
            for (int i = 0; i < deg; i++) {
                const int x = v2c[g->transpose[b + i]], a = x < 0 ? -x : x;
                sign ^= x < 0;
                const int t = a < min2 ? a : min2;
                min2 = t > min1 ? t : min1;
                min1 = t < min1 ? t : min1;
            }
            const int n1 = i8_norm(min1, rule, p), n2 = i8_norm(min2, rule, p);
            for (int i = 0; i < deg; i++) {
                const int x = v2c[g->transpose[b + i]], a = x < 0 ? -x : x;
                const int mag = (a == min1) ? n2 : n1;
                c2v[g->transpose[b + i]] = (sign ^ (x < 0)) ? -mag : mag;
            }
        }
        if (enable_syndrome && ite != n_ite - 1) {
            for (int v = 0; v < N; v++) {
                int sum = 0;
                for (int e = g->vn_ptr[v]; e < g->vn_ptr[v + 1]; e++) sum += c2v[e];
                post[v] = Yq[v] + sum;
            }
            const int z = i8_syndrome_is_zero(g, post, target);
            cur_depth = z ? (cur_depth + 1) % syndrome_depth : 0;
            if (z && cur_depth == 0) { ite++; return ite; }
        }
    }
    for (int v = 0; v < N; v++) {
        int sum = 0;
        for (int e = g->vn_ptr[v]; e < g->vn_ptr[v + 1]; e++) sum += c2v[e];
        post[v] = Yq[v] + sum;
    }
    return ite;
}

/* Fixed-point horizontal layered sweep, posterior kept in 8 bits: the recursion of BPSK_nrldpc_sim_RM_FP.m:50-93
 * (L = L - R; saturate to +-31 (maxqr); min-sum; R = new; L = saturate(L + R) to +-127 (maxqL)) with one check per layer
 * row, syndrome test after every iteration as decode_hlayered. */
static int decode_hlayered_i8(const orc_graph *g, int rule, int p, int n_ite, int enable_syndrome, int syndrome_depth,
                              const int *Yq, const int *target, int *post, int *msg, int *contr)
{
    const int N = g->N, M = g->M, E = g->E;
    memcpy(post, Yq, sizeof(int) * (size_t)N);
    memset(msg, 0, sizeof(int) * (size_t)E);
    int cur_depth = 0, ite = 0;
    for (; ite < n_ite; ite++) {
        for (int c = 0; c < M; c++) {
            const int b = g->cn_ptr[c], deg = g->cn_ptr[c + 1] - b;
            int sign = target ? (target[c] & 1) : 0, min1 = 31, min2 = 31;
            for (int i = 0; i < deg; i++) {
                contr[i] = post[g->cn_var[b + i]] - msg[b + i];
                const int x = contr[i] > 31 ? 31 : (contr[i] < -31 ? -31 : contr[i]), a = x < 0 ? -x : x;      /* maxqr = 31 */
                sign ^= x < 0;
                const int t = a < min2 ? a : min2;
                min2 = t > min1 ? t : min1;
                min1 = t < min1 ? t : min1;
            }
            const int n1 = i8_norm(min1, rule, p), n2 = i8_norm(min2, rule, p);
            for (int i = 0; i < deg; i++) {
                const int x = contr[i] > 31 ? 31 : (contr[i] < -31 ? -31 : contr[i]), a = x < 0 ? -x : x;      /* maxqr = 31 */
                const int mag = (a == min1) ? n2 : n1;
                const int out = (sign ^ (x < 0)) ? -mag : mag;
                msg[b + i] = out;
                const int np = contr[i] + out;
                post[g->cn_var[b + i]] = np > 127 ? 127 : (np < -127 ? -127 : np);
            }
        }
        if (enable_syndrome) {
            const int z = i8_syndrome_is_zero(g, post, target);
            cur_depth = z ? (cur_depth + 1) % syndrome_depth : 0;
            if (z && cur_depth == 0) { ite++; return ite; }
        }
    }
    return ite;
}

int orc_decode_i8(const orc_graph *g, int schedule, int rule, float rule_param, float quant_scale, int n_ite, int enable_syndrome, int syndrome_depth,
                  const float *Y_N, const int *target, int n_frames, float *post_out, int *hard, int *iters, int *synd_ok, int n_threads)
{
    if (!g || !Y_N || n_frames < 0 || n_ite < 0) return -1;
    if (rule < ORC_RULE_MS || rule > ORC_RULE_NMS) return -2;
    if (schedule != ORC_SCHED_FLOODING && schedule != ORC_SCHED_HLAYERED) return -3;
    if (syndrome_depth < 1) syndrome_depth = 1;
    int p = 0;
    if (rule == ORC_RULE_OMS) p = (int)lrintf(rule_param * quant_scale);
    if (rule == ORC_RULE_NMS) p = (int)lrintf(rule_param * 128.0f);
    if (p < 0) p = 0;
    if (p > 128) p = 128;
    const int N = g->N, E = g->E;
#ifdef _OPENMP
    if (n_threads < 1) n_threads = 1;
#pragma omp parallel num_threads(n_threads)
#endif
    {
        int *yq = (int *)malloc(sizeof(int) * (size_t)N * 2);
        int *post = yq + N;
        int *a = (int *)malloc(sizeof(int) * ((size_t)E * 2 + (size_t)g->max_dc + 1));
        int *b = a + E, *contr = b + E;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (int f = 0; f < n_frames; f++) {
            const int *tg = target ? target + (size_t)f * g->M : 0;
            for (int v = 0; v < N; v++) yq[v] = i8_quant(Y_N[(size_t)f * N + v], quant_scale);
            const int it = schedule == ORC_SCHED_FLOODING ? decode_flooding_i8(g, rule, p, n_ite, enable_syndrome, syndrome_depth, yq, tg, post, a, b)
                                                          : decode_hlayered_i8(g, rule, p, n_ite, enable_syndrome, syndrome_depth, yq, tg, post, a, contr);
            if (post_out) for (int v = 0; v < N; v++) post_out[(size_t)f * N + v] = (float)post[v];
            if (hard) for (int v = 0; v < N; v++) hard[(size_t)f * N + v] = post[v] < 0;
            if (iters) iters[f] = it;
            if (synd_ok) synd_ok[f] = i8_syndrome_is_zero(g, post, tg);
        }
        free(yq); free(a);
    }
    return 0;
}

/* ------------------------------------------------------------- privacy amplification ----- */

/* rnd_getPrngValue2_32 (EC/subcomponents/rnd.c:118-127): 32 x { b = parity(state & 0xe0000200); state <<= 1; state += b } */
unsigned int orc_lfsr32(unsigned int *state)
{
    for (int k = 32; k; k--) {
        const unsigned int b = (unsigned int)__builtin_parity(*state & 0xe0000200u);
        *state <<= 1;
        *state += b;
    }
    return *state;
}

/* privAmp_doPrivAmp, EC/subcomponents/priv_amp.c:194-218: mask the tail of the key, then for every target bit
 * m = XOR_j key[j] & prng32(); bit i = parity(m); words MSB-first (uint32AllZeroExceptAtN, helpers.h:68) */
void orc_privamp(const unsigned int *key_words, int workbits, unsigned int seed, int final_bits, unsigned int *final_words)
{
    const int numwords = (workbits + 31) / 32;
    unsigned int *key = (unsigned int *)malloc(sizeof(unsigned int) * (size_t)numwords);
    memcpy(key, key_words, sizeof(unsigned int) * (size_t)numwords);
    if (workbits & 31) key[numwords - 1] &= 0xffffffffu << (32 - (workbits & 31));
    memset(final_words, 0, sizeof(unsigned int) * (size_t)((final_bits + 31) / 32));
    unsigned int state = seed;
    for (int i = 0; i < final_bits; i++) {
        unsigned int m = 0;
        for (int j = 0; j < numwords; j++) m ^= key[j] & orc_lfsr32(&state);
        if (__builtin_parity(m)) final_words[i / 32] |= 1u << (31 - (i & 31));
    }
    free(key);
}
