/*
 * qldpc_graph.c -- H-matrix layer of libqldpc (plain C host code).
 *
 * Replaces, for the LDPC path of the reference harness:
 *   tools::LDPC_matrix_handler::read_matrix_size / read   VAR/main.cpp (alist-v1.0.1):324,338
 *   .qc reading for Encoder_LDPC_from_QC                   VAR/main.cpp (qc):145
 *   tools::build_dvbs2 + tools::build_H                    BS/src/main.cpp:175-176  (DVB-like IRA generator)
 *   H.get_cols_max_degree()                                BS/src/main.cpp:178
 *   h(), min_cr(), parity_bits_to_punct(), LLR()           BS/src/main.cpp:19-34
 * (BS = errorcorrection/ldpc_examples/my_project_with_aff3ct/examples/bootstrap, VAR = its
 * "src/variants (copy out as main.cpp to use)").
 */
#define _GNU_SOURCE
#include "qldpc_graph.h"
#include "../../include/qldpc.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stddef.h>
#include <unistd.h>

/* ------------------------------------------------------------------ errors ------------------- */

static __thread char g_err[512];

void qldpc_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

const char *qldpc_last_error(void) { return g_err; }

const char *qldpc_strerror(int status)
{
    switch (status) {
    case QLDPC_OK: return "ok";
    case QLDPC_EINVAL: return "invalid argument";
    case QLDPC_ENOMEM: return "out of memory";
    case QLDPC_EIO: return "matrix file unreadable or malformed";
    case QLDPC_EHIP: return "HIP runtime error";
    case QLDPC_ENODEV: return "no usable HIP device";
    case QLDPC_ESIZE: return "size mismatch";
    case QLDPC_EUNSUPPORTED: return "unsupported";
    case QLDPC_ESTATE: return "call sequence error";
    case QLDPC_EDECODE: return "reconciliation failed";
    default: return "unknown status";
    }
}

int qldpc_version(void) { return QLDPC_VERSION; }

/* ------------------------------------------------------------------ scalar helpers ----------- */

/* LLR(BER) = -log(BER / (1 - BER)), BS/src/main.cpp:20 (double arithmetic, narrowed like the harness). */
float qldpc_llr_from_ber(float ber) { return (float)(-log((double)ber / (1.0 - (double)ber))); }

/* Modem_OOK_BSC::set_noise + demodulate (BS/src/main.cpp:317,348): |LLR| = log((1-p)/p) in float. */
float qldpc_bsc_llr(float ber) { return logf((1.0f - ber) / ber); }

/* h(QBER), BS/src/main.cpp:23 */
float qldpc_binary_entropy(float q)
{
    if (q <= 0.0f || q >= 1.0f) return 0.0f;
    return (float)((-(double)q) * log2((double)q) - (1.0 - (double)q) * log2(1.0 - (double)q));
}

/* min_cr(QBER, EFF) = 1 / (1 + EFF * h(QBER)), BS/src/main.cpp:29 */
float qldpc_min_code_rate(float qber, float efficiency)
{
    return (float)(1.0 / (1.0 + (double)efficiency * (double)qldpc_binary_entropy(qber)));
}

/* parity_bits_to_punct(INFO_B, TTL_B, GOAL_CR) = -((INFO_B) - GOAL_CR * TTL_B) / GOAL_CR, truncated
 * to int as the harness does (BS/src/main.cpp:34,280). */
int qldpc_parity_bits_to_punct(int N, int K, float target_cr)
{
    return (int)(-((float)K - target_cr * (float)N) / target_cr);
}

/* ------------------------------------------------------------------ construction ------------- */

void qldpc_code_free(qldpc_code *c)
{
    if (!c) return;
    free(c->cn_ptr); free(c->cn_var); free(c->vn_ptr); free(c->vn_chk); free(c->transpose);
    free(c->layer_ptr); free(c->layer_order);
    free(c);
}

/*
 * Level schedule for the horizontal-layered decoder: level(c) = 1 + max level of any earlier check
 * sharing a VN.  Checks of one level are VN-disjoint, and running levels in order is
 * operation-for-operation the sequential c = 0..M-1 sweep of Decoder_LDPC_BP_horizontal_layered.
 * When that has (almost) no parallelism -- a dual-diagonal chain makes level(c) = c -- fall back to
 * a greedy colouring, which is the sequential sweep of a row-permuted H (order exported through
 * qldpc_code_layer_order so a reference decoder can be given the same row order).
 */
/*
 * DSATUR colouring of the check-conflict graph (two checks conflict when they share a VN), at most 64 colours: the uncoloured check whose VNs
 * already carry the most distinct colours goes next (ties: the larger conflict degree, then the lower index) and takes the lowest colour none of
 * its VNs carries.  A check's forbidden set is the OR of its VNs' used-colour words, its saturation the popcount of that; a bucket queue by
 * saturation holds the uncoloured checks.  On the IRA codes of the bench (conflict degree ~ 78, largest VN degree 11) first-fit in index order needs
 * 30 classes with a long tail of small ones (10 756 ... 283, 26 checks at N = 10^6); DSATUR needs 24, of which 22 hold 5 700 - 10 300 checks
 * (tools/dsatur_probe.c; iterated-greedy passes on top do not get below 24).  Fewer, fuller layer launches per sweep.  Returns the number of
 * colours with lvl[c] = colour + 1, or 0 when 64 colours do not suffice / memory runs out (the caller then colours first-fit).
 */
static int dsatur_cmp(const void *a, const void *b)      /* keys carry their check's index: no state outside the array (codes are built on several threads) */
{
    const uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}
static int colour_dsatur(const qldpc_code *g, int *lvl)
{
    const int N = g->N, M = g->M;
    uint64_t *used = (uint64_t *)calloc((size_t)N, sizeof(uint64_t));
    int *sat = (int *)calloc((size_t)M, sizeof(int)), *nxt = (int *)malloc(sizeof(int) * (size_t)M), *prv = (int *)malloc(sizeof(int) * (size_t)M);
    uint64_t *keys = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)M);
    int k = 0, head[65];
    if (!used || !sat || !nxt || !prv || !keys) { k = -1; goto out; }
    for (int c = 0; c < M; c++) {
        uint64_t d = 0;
        for (int j = g->cn_ptr[c]; j < g->cn_ptr[c + 1]; j++) d += (uint64_t)(g->vn_ptr[g->cn_var[j] + 1] - g->vn_ptr[g->cn_var[j]] - 1);
        keys[c] = (d << 32) | (uint64_t)(0xffffffffu - (uint32_t)c);      /* conflict degree, then the index (descending) */
        lvl[c] = 0;
    }
    qsort(keys, (size_t)M, sizeof(uint64_t), dsatur_cmp);      /* ascending: pushed to the front one by one, the largest degree (lowest index among equals) ends up first */
    for (int s = 0; s < 65; s++) head[s] = -1;
    for (int i = 0; i < M; i++) { const int c = (int)(0xffffffffu - (uint32_t)(keys[i] & 0xffffffffu)); nxt[c] = head[0]; prv[c] = -1; if (head[0] >= 0) prv[head[0]] = c; head[0] = c; }
    for (int done = 0; done < M; done++) {
        int s = 64;
        while (s > 0 && head[s] < 0) s--;
        const int c = head[s];
        if (c < 0) { k = 0; goto out; }      /* cannot happen: every uncoloured check sits in a bucket */
        head[s] = nxt[c];
        if (nxt[c] >= 0) prv[nxt[c]] = -1;
        uint64_t forb = 0;
        for (int j = g->cn_ptr[c]; j < g->cn_ptr[c + 1]; j++) forb |= used[g->cn_var[j]];
        if (!~forb) { k = 0; goto out; }      /* a 65th colour: not for this routine */
        const int q = __builtin_ctzll(~forb);
        lvl[c] = q + 1;
        if (q + 1 > k) k = q + 1;
        for (int j = g->cn_ptr[c]; j < g->cn_ptr[c + 1]; j++) {
            const int v = g->cn_var[j];
            if ((used[v] >> q) & 1ull) continue;
            used[v] |= 1ull << q;
            for (int t = g->vn_ptr[v]; t < g->vn_ptr[v + 1]; t++) {
                const int d = g->vn_chk[t];
                if (lvl[d]) continue;
                uint64_t f = 0;
                for (int jj = g->cn_ptr[d]; jj < g->cn_ptr[d + 1]; jj++) f |= used[g->cn_var[jj]];
                const int ns = __builtin_popcountll(f);
                if (ns != sat[d]) {
                    if (prv[d] >= 0) nxt[prv[d]] = nxt[d]; else head[sat[d]] = nxt[d];
                    if (nxt[d] >= 0) prv[nxt[d]] = prv[d];
                    nxt[d] = head[ns]; prv[d] = -1;
                    if (head[ns] >= 0) prv[head[ns]] = d;
                    head[ns] = d;
                    sat[d] = ns;
                }
            }
        }
    }
out:
    free(used); free(sat); free(nxt); free(prv); free(keys);
    return k > 0 ? k : 0;
}

/* colours (bit q = colour q + 1 of lvl) carried by the checks on c's VNs, c itself and `skip` left out */
static uint64_t colours_around(const qldpc_code *g, const int *lvl, int c, int skip)
{
    uint64_t f = 0;
    for (int j = g->cn_ptr[c]; j < g->cn_ptr[c + 1]; j++) {
        const int v = g->cn_var[j];
        for (int t = g->vn_ptr[v]; t < g->vn_ptr[v + 1]; t++) {
            const int d = g->vn_chk[t];
            if (d != c && d != skip && lvl[d]) f |= 1ull << (lvl[d] - 1);
        }
    }
    return f;
}
/*
 * DSATUR leaves a last class of a handful of checks (23 of 200 000 at N = 10^6, 1 of 13 107 at N = 65 536): a launch of its own per sweep for nothing.
 * Each of its checks c is moved into a class q where it conflicts with at most three checks, after each of those has moved to another class that none of
 * its neighbours (c included, at its new colour) carries.  Returns the new number of classes (k - 1 if the class could be emptied, else k with lvl untouched).
 */
static int dissolve_last_class(const qldpc_code *g, int *lvl, int k)
{
    const int M = g->M;
    int n_last = 0;
    for (int c = 0; c < M; c++) n_last += lvl[c] == k;
    if (k < 3 || k > 64 || n_last == 0 || n_last > 256) return k;
    int *saved = (int *)malloc(sizeof(int) * (size_t)M);
    if (!saved) return k;
    memcpy(saved, lvl, sizeof(int) * (size_t)M);
    for (int c = 0; c < M; c++) {
        if (lvl[c] != k) continue;
        int moved = 0;
        for (int q = 0; q < k - 1 && !moved; q++) {
            /* the checks of class q + 1 that share a VN with c: up to three of them may be moved out of the way */
            int blk[3], cnt = 0;
            for (int j = g->cn_ptr[c]; j < g->cn_ptr[c + 1] && cnt <= 3; j++) {
                const int v = g->cn_var[j];
                for (int t = g->vn_ptr[v]; t < g->vn_ptr[v + 1]; t++) {
                    const int d = g->vn_chk[t];
                    if (d == c || lvl[d] != q + 1) continue;
                    int seen = 0;
                    for (int i = 0; i < cnt && i < 3; i++) seen |= blk[i] == d;
                    if (!seen) { if (cnt < 3) blk[cnt] = d; cnt++; }
                }
            }
            if (cnt > 3) continue;
            int old[3], ok = 1;
            for (int i = 0; i < cnt && ok; i++) {
                old[i] = lvl[blk[i]];
                uint64_t f = colours_around(g, lvl, blk[i], c) | (1ull << q) | (1ull << (k - 1));      /* c will carry q; the class being dissolved is not a target */
                if (k - 1 < 64) f |= ~0ull << (k - 1);
                if (!~f) { for (int u = 0; u < i; u++) lvl[blk[u]] = old[u]; ok = 0; break; }
                lvl[blk[i]] = __builtin_ctzll(~f) + 1;      /* (the next blocker sees this one at its new colour) */
            }
            if (!ok) continue;
            lvl[c] = q + 1;
            moved = 1;
        }
        if (!moved) { memcpy(lvl, saved, sizeof(int) * (size_t)M); free(saved); return k; }
    }
    free(saved);
    return k - 1;
}

static int build_layers(qldpc_code *g)
{
    const int N = g->N, M = g->M;
    int *lvl = (int *)malloc(sizeof(int) * (size_t)M);
    int *last = (int *)calloc((size_t)N, sizeof(int));
    if (!lvl || !last) { free(lvl); free(last); return QLDPC_ENOMEM; }
    int nlev = 0;
    for (int c = 0; c < M; c++) {
        int l = 0;
        for (int j = g->cn_ptr[c]; j < g->cn_ptr[c + 1]; j++) if (last[g->cn_var[j]] > l) l = last[g->cn_var[j]];
        l += 1;
        lvl[c] = l;
        for (int j = g->cn_ptr[c]; j < g->cn_ptr[c + 1]; j++) last[g->cn_var[j]] = l;
        if (l > nlev) nlev = l;
    }
    g->layer_natural = 1;
    int dsat = 0;
    if (nlev > 256 && nlev > M / 16 && !getenv("QLDPC_FIRST_FIT_LAYERS") && (dsat = colour_dsatur(g, lvl)) > 0) {
        for (int again = 1; again; ) { const int k2 = dissolve_last_class(g, lvl, dsat); again = k2 < dsat; dsat = k2; }
        nlev = dsat;
        g->layer_natural = 0;
    }
    if (!dsat && nlev > 256 && nlev > M / 16) {
        /* greedy colouring, first-fit, with a per-VN bitset of used colours */
        int words = 4;
        uint64_t *used = (uint64_t *)calloc((size_t)N * words, sizeof(uint64_t));
        uint64_t *acc = (uint64_t *)malloc(sizeof(uint64_t) * 64);
        if (!used || !acc) { free(used); free(acc); free(lvl); free(last); return QLDPC_ENOMEM; }
        nlev = 0;
        for (int c = 0; c < M; c++) {
            int col = -1;
            for (;;) {
                memset(acc, 0, sizeof(uint64_t) * (size_t)words);
                for (int j = g->cn_ptr[c]; j < g->cn_ptr[c + 1]; j++) {
                    const uint64_t *u = used + (size_t)g->cn_var[j] * words;
                    for (int w = 0; w < words; w++) acc[w] |= u[w];
                }
                for (int w = 0; w < words && col < 0; w++)
                    if (~acc[w]) col = w * 64 + __builtin_ctzll(~acc[w]);
                if (col >= 0) break;
                if (words >= 64) { free(used); free(acc); free(lvl); free(last); qldpc_set_error("layer colouring needs > 4096 colours"); return QLDPC_EUNSUPPORTED; }
                /* grow the bitsets */
                int nw = words * 2;
                uint64_t *nu = (uint64_t *)calloc((size_t)N * nw, sizeof(uint64_t));
                if (!nu) { free(used); free(acc); free(lvl); free(last); return QLDPC_ENOMEM; }
                for (int v = 0; v < N; v++) memcpy(nu + (size_t)v * nw, used + (size_t)v * words, sizeof(uint64_t) * (size_t)words);
                free(used); used = nu; words = nw;
            }
            lvl[c] = col + 1;
            for (int j = g->cn_ptr[c]; j < g->cn_ptr[c + 1]; j++) used[(size_t)g->cn_var[j] * words + col / 64] |= 1ull << (col % 64);
            if (col + 1 > nlev) nlev = col + 1;
        }
        free(used); free(acc);
        g->layer_natural = 0;
    }
    free(last);
    g->n_layers = nlev;
    g->layer_ptr = (int *)calloc((size_t)nlev + 1, sizeof(int));
    g->layer_order = (int *)malloc(sizeof(int) * (size_t)M);
    if (!g->layer_ptr || !g->layer_order) { free(lvl); return QLDPC_ENOMEM; }
    for (int c = 0; c < M; c++) g->layer_ptr[lvl[c]]++;
    for (int l = 0; l < nlev; l++) g->layer_ptr[l + 1] += g->layer_ptr[l];
    int *fill = (int *)calloc((size_t)nlev, sizeof(int));
    if (!fill) { free(lvl); return QLDPC_ENOMEM; }
    for (int c = 0; c < M; c++) { int l = lvl[c] - 1; g->layer_order[g->layer_ptr[l] + fill[l]++] = c; }
    free(fill); free(lvl);
    return QLDPC_OK;
}

static int detect_ira(const qldpc_code *g)
{
    /* parity VNs K..N-1 with K = N - M: VN K+c touches exactly checks {c, c+1} (last one only {M-1}) */
    const int K = g->N - g->M;
    if (K <= 0) return 0;
    for (int c = 0; c < g->M; c++) {
        const int v = K + c, d = g->vn_ptr[v + 1] - g->vn_ptr[v];
        const int *s = g->vn_chk + g->vn_ptr[v];
        if (c < g->M - 1) { if (d != 2 || s[0] != c || s[1] != c + 1) return 0; }
        else if (d != 1 || s[0] != c) return 0;
    }
    return K;
}

int qldpc_code_from_edges(int N, int M, int E, const int *var, const int *chk, qldpc_code **out)
{
    if (!out) return QLDPC_EINVAL;
    *out = NULL;
    if (N <= 0 || M <= 0 || E <= 0 || !var || !chk) { qldpc_set_error("code_from_edges: bad sizes N=%d M=%d E=%d", N, M, E); return QLDPC_EINVAL; }
    qldpc_code *g = (qldpc_code *)calloc(1, sizeof(*g));
    if (!g) return QLDPC_ENOMEM;
    g->N = N; g->M = M; g->E = E;
    g->cn_ptr = (int *)calloc((size_t)M + 1, sizeof(int));
    g->vn_ptr = (int *)calloc((size_t)N + 1, sizeof(int));
    g->cn_var = (int *)malloc(sizeof(int) * (size_t)E);
    g->vn_chk = (int *)malloc(sizeof(int) * (size_t)E);
    g->transpose = (int *)malloc(sizeof(int) * (size_t)E);
    if (!g->cn_ptr || !g->vn_ptr || !g->cn_var || !g->vn_chk || !g->transpose) { qldpc_code_free(g); return QLDPC_ENOMEM; }
    for (int e = 0; e < E; e++) {
        if (var[e] < 0 || var[e] >= N || chk[e] < 0 || chk[e] >= M) {
            qldpc_set_error("code_from_edges: edge %d = (%d,%d) out of range", e, var[e], chk[e]);
            qldpc_code_free(g);
            return QLDPC_EINVAL;
        }
        g->cn_ptr[chk[e] + 1]++;
        g->vn_ptr[var[e] + 1]++;
    }
    for (int c = 0; c < M; c++) { if (g->cn_ptr[c + 1] > g->max_dc) g->max_dc = g->cn_ptr[c + 1]; g->cn_ptr[c + 1] += g->cn_ptr[c]; }
    for (int v = 0; v < N; v++) { if (g->vn_ptr[v + 1] > g->max_dv) g->max_dv = g->vn_ptr[v + 1]; g->vn_ptr[v + 1] += g->vn_ptr[v]; }
    int *cur = (int *)calloc((size_t)(M > N ? M : N), sizeof(int));
    if (!cur) { qldpc_code_free(g); return QLDPC_ENOMEM; }
    for (int e = 0; e < E; e++) g->cn_var[g->cn_ptr[chk[e]] + cur[chk[e]]++] = var[e];
    memset(cur, 0, sizeof(int) * (size_t)(M > N ? M : N));
    /* VN slots are handed out while sweeping checks in ascending order (flooding-decoder `transpose`) */
    for (int c = 0; c < M; c++)
        for (int k = g->cn_ptr[c]; k < g->cn_ptr[c + 1]; k++) {
            const int v = g->cn_var[k];
            const int slot = g->vn_ptr[v] + cur[v]++;
            g->transpose[k] = slot;
            g->vn_chk[slot] = c;
        }
    free(cur);
    /* duplicate edges inside a check would make messages alias */
    for (int v = 0; v < N; v++)
        for (int s = g->vn_ptr[v] + 1; s < g->vn_ptr[v + 1]; s++)
            if (g->vn_chk[s] == g->vn_chk[s - 1]) {
                qldpc_set_error("code_from_edges: duplicate edge (var %d, chk %d)", v, g->vn_chk[s]);
                qldpc_code_free(g);
                return QLDPC_EINVAL;
            }
    g->ira_K = detect_ira(g);
    int rc = build_layers(g);
    if (rc != QLDPC_OK) { qldpc_code_free(g); return rc; }
    *out = g;
    return QLDPC_OK;
}

/* MacKay alist: "N M" / max degrees / VN degrees / CN degrees / N VN lists (1-based, 0-padded) / M CN lists. */
int qldpc_code_from_alist(const char *path, qldpc_code **out)
{
    if (!out || !path) return QLDPC_EINVAL;
    *out = NULL;
    FILE *f = fopen(path, "r");
    if (!f) { qldpc_set_error("alist: cannot open %s", path); return QLDPC_EIO; }
    int N = 0, M = 0, a = 0, b = 0, rc = QLDPC_EIO;
    int *dv = NULL, *dc = NULL, *var = NULL, *chk = NULL;
    char *line = NULL; size_t cap = 0;
    if (fscanf(f, "%d %d %d %d", &N, &M, &a, &b) != 4 || N <= 0 || M <= 0) { qldpc_set_error("alist: bad header in %s", path); goto done; }
    dv = (int *)malloc(sizeof(int) * (size_t)N); dc = (int *)malloc(sizeof(int) * (size_t)M);
    if (!dv || !dc) { rc = QLDPC_ENOMEM; goto done; }
    long E = 0, Ec = 0;
    for (int v = 0; v < N; v++) { if (fscanf(f, "%d", &dv[v]) != 1 || dv[v] < 0) { qldpc_set_error("alist: bad VN degree list"); goto done; } E += dv[v]; }
    for (int c = 0; c < M; c++) { if (fscanf(f, "%d", &dc[c]) != 1 || dc[c] < 0) { qldpc_set_error("alist: bad CN degree list"); goto done; } Ec += dc[c]; }
    if (E != Ec || E <= 0 || E > 0x7fffffff) { qldpc_set_error("alist: degree sums differ (%ld vs %ld)", E, Ec); goto done; }
    var = (int *)malloc(sizeof(int) * (size_t)E); chk = (int *)malloc(sizeof(int) * (size_t)E);
    if (!var || !chk) { rc = QLDPC_ENOMEM; goto done; }
    { int ch; while ((ch = fgetc(f)) != EOF && ch != '\n') {} }
    long e = 0;
    for (int v = 0; v < N; v++) {
        if (getline(&line, &cap, f) < 0) { qldpc_set_error("alist: truncated VN part"); goto done; }
        int got = 0;
        for (char *p = line;;) {
            char *end; long x = strtol(p, &end, 10);
            if (end == p) break;
            p = end;
            if (x == 0) continue;
            if (x < 0 || x > M || e >= E) { qldpc_set_error("alist: VN %d lists check %ld", v, x); goto done; }
            var[e] = v; chk[e] = (int)x - 1; e++; got++;
        }
        if (got != dv[v]) { qldpc_set_error("alist: VN %d has %d entries, degree says %d", v, got, dv[v]); goto done; }
    }
    for (int c = 0; c < M; c++) {
        if (getline(&line, &cap, f) < 0) { qldpc_set_error("alist: truncated CN part"); goto done; }
        int got = 0;
        for (char *p = line;;) { char *end; long x = strtol(p, &end, 10); if (end == p) break; p = end; if (x > 0) got++; }
        if (got != dc[c]) { qldpc_set_error("alist: CN %d has %d entries, degree says %d", c, got, dc[c]); goto done; }
    }
    rc = qldpc_code_from_edges(N, M, (int)E, var, chk, out);
done:
    free(line); free(dv); free(dc); free(var); free(chk); fclose(f);
    return rc;
}

/* base matrix B[mb][nb] of circulant shifts (-1 = zero block) -> lifted code: row i*Z+z meets column j*Z + (z+s) % Z */
static int qc_expand(int nb, int mb, int Z, const int *B, qldpc_code **out)
{
    long blocks = 0;
    for (long i = 0; i < (long)nb * mb; i++) if (B[i] >= 0) blocks++;
    const long E = blocks * Z;
    if (E <= 0 || E > 0x7fffffff) { qldpc_set_error("qc: empty or oversized matrix"); return QLDPC_EIO; }
    int *var = (int *)malloc(sizeof(int) * (size_t)E), *chk = (int *)malloc(sizeof(int) * (size_t)E);
    if (!var || !chk) { free(var); free(chk); return QLDPC_ENOMEM; }
    long e = 0;
    for (int i = 0; i < mb; i++)
        for (int j = 0; j < nb; j++) {
            const int s = B[(long)i * nb + j];
            if (s < 0) continue;
            for (int z = 0; z < Z; z++, e++) { chk[e] = i * Z + z; var[e] = j * Z + (z + s) % Z; }
        }
    int rc = qldpc_code_from_edges(nb * Z, mb * Z, (int)E, var, chk, out);
    free(var); free(chk);
    return rc;
}

/* AFF3CT .qc: "cols rows Z", then rows x cols shifts; -1 = zero block, s = identity right-shifted by s. */
int qldpc_code_from_qc(const char *path, qldpc_code **out)
{
    if (!out || !path) return QLDPC_EINVAL;
    *out = NULL;
    FILE *f = fopen(path, "r");
    if (!f) { qldpc_set_error("qc: cannot open %s", path); return QLDPC_EIO; }
    int nb = 0, mb = 0, Z = 0;
    if (fscanf(f, "%d %d %d", &nb, &mb, &Z) != 3 || nb <= 0 || mb <= 0 || Z <= 0) { fclose(f); qldpc_set_error("qc: bad header in %s", path); return QLDPC_EIO; }
    int *B = (int *)malloc(sizeof(int) * (size_t)nb * mb);
    if (!B) { fclose(f); return QLDPC_ENOMEM; }
    for (long i = 0; i < (long)nb * mb; i++) {
        if (fscanf(f, "%d", &B[i]) != 1 || B[i] < -1) { free(B); fclose(f); qldpc_set_error("qc: bad entry %ld", i); return QLDPC_EIO; }
    }
    fclose(f);
    int rc = qc_expand(nb, mb, Z, B, out);
    free(B);
    return rc;
}

/* ------------------------------------------------------------------ IRA generator ------------ */

static uint64_t splitmix64(uint64_t *s)
{
    uint64_t z = (*s += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
typedef struct { uint64_t s[4]; } xoshiro;
static inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static uint64_t xo_next(xoshiro *r)
{
    const uint64_t res = rotl64(r->s[1] * 5, 7) * 9, t = r->s[1] << 17;
    r->s[2] ^= r->s[0]; r->s[3] ^= r->s[1]; r->s[1] ^= r->s[2]; r->s[0] ^= r->s[3];
    r->s[2] ^= t; r->s[3] = rotl64(r->s[3], 45);
    return res;
}
static uint64_t xo_below(xoshiro *r, uint64_t n)   /* unbiased, n >= 1 */
{
    const uint64_t lim = UINT64_MAX - UINT64_MAX % n;
    uint64_t x;
    do x = xo_next(r); while (x >= lim);
    return x % n;
}

int qldpc_code_ira(int N, int K, float hi_frac, int dv_hi, int dv_lo, uint64_t seed, qldpc_code **out)
{
    if (!out) return QLDPC_EINVAL;
    *out = NULL;
    const int M = N - K;
    if (N <= 0 || K <= 0 || M <= 1 || dv_hi < dv_lo || dv_lo < 1 || hi_frac < 0.0f || hi_frac > 1.0f) {
        qldpc_set_error("code_ira: bad parameters N=%d K=%d", N, K);
        return QLDPC_EINVAL;
    }
    const int n_hi = (int)floor((double)hi_frac * (double)K);
    long base = (long)n_hi * dv_hi + (long)(K - n_hi) * dv_lo;
    const long dci = (base + M - 1) / M;           /* info edges per check */
    long deficit = dci * M - base;                 /* raised one degree at a time on the low-degree VNs */
    if (dci > K) { qldpc_set_error("code_ira: %ld info edges per check > K", dci); return QLDPC_EINVAL; }
    const long T = dci * M, E = T + 2L * M - 1;
    if (E > 0x7fffffff) { qldpc_set_error("code_ira: too many edges"); return QLDPC_EINVAL; }
    int *deg = (int *)malloc(sizeof(int) * (size_t)K);
    int *stub = (int *)malloc(sizeof(int) * (size_t)T);
    int *var = (int *)malloc(sizeof(int) * (size_t)E), *chk = (int *)malloc(sizeof(int) * (size_t)E);
    if (!deg || !stub || !var || !chk) { free(deg); free(stub); free(var); free(chk); return QLDPC_ENOMEM; }
    for (int v = 0; v < K; v++) deg[v] = v < n_hi ? dv_hi : dv_lo;
    for (int v = n_hi; deficit > 0; v++) {          /* wrap over the low-degree VNs if needed */
        if (v >= K) v = n_hi < K ? n_hi : 0;
        deg[v]++; deficit--;
    }
    long t = 0;
    for (int v = 0; v < K; v++) for (int d = 0; d < deg[v]; d++) stub[t++] = v;
    xoshiro rng; uint64_t sm = seed;
    for (int i = 0; i < 4; i++) rng.s[i] = splitmix64(&sm);
    for (long i = T - 1; i > 0; i--) { long j = (long)xo_below(&rng, (uint64_t)i + 1); int x = stub[i]; stub[i] = stub[j]; stub[j] = x; }
    /* duplicate-in-row repair: swap an offending stub with a later one that fits (wrapping around) */
    int rc = QLDPC_OK;
    for (int c = 0; c < M && rc == QLDPC_OK; c++) {
        int *row = stub + (long)c * dci;
        for (long i = 1; i < dci; i++) {
            int dup = 0;
            for (long j = 0; j < i; j++) if (row[j] == row[i]) { dup = 1; break; }
            if (!dup) continue;
            long tries = 0, p = (long)c * dci + i;
            for (long q = (p + 1) % T; tries < T; q = (q + 1) % T, tries++) {
                const long qc = q / dci;
                if (qc == c) continue;
                const int cand = stub[q];
                int bad = 0;
                for (long j = 0; j < i; j++) if (row[j] == cand) { bad = 1; break; }          /* cand fits row c?   */
                if (bad) continue;
                const int *qrow = stub + qc * dci;
                for (long j = 0; j < dci; j++) if (qrow + j != stub + q && qrow[j] == row[i]) { bad = 1; break; } /* row[i] fits row qc? */
                if (bad) continue;
                stub[q] = row[i]; row[i] = cand;
                break;
            }
            if (tries >= T) { qldpc_set_error("code_ira: could not repair duplicate in check %d", c); rc = QLDPC_EINVAL; break; }
        }
    }
    if (rc == QLDPC_OK) {
        long e = 0;
        for (int c = 0; c < M; c++) {
            /* ascending VN order inside a check, as add_connection() sweeps give it */
            int *row = stub + (long)c * dci;
            for (long i = 1; i < dci; i++) { int x = row[i]; long j = i - 1; while (j >= 0 && row[j] > x) { row[j + 1] = row[j]; j--; } row[j + 1] = x; }
            for (long i = 0; i < dci; i++, e++) { var[e] = row[i]; chk[e] = c; }
            if (c > 0) { var[e] = K + c - 1; chk[e] = c; e++; }
            var[e] = K + c; chk[e] = c; e++;
        }
        rc = qldpc_code_from_edges(N, M, (int)E, var, chk, out);
    }
    free(deg); free(stub); free(var); free(chk);
    return rc;
}

/* ------------------------------------------------------------------ accessors ---------------- */

int qldpc_code_n(const qldpc_code *c) { return c ? c->N : QLDPC_EINVAL; }
int qldpc_code_m(const qldpc_code *c) { return c ? c->M : QLDPC_EINVAL; }
int qldpc_code_e(const qldpc_code *c) { return c ? c->E : QLDPC_EINVAL; }
int qldpc_code_max_cn_degree(const qldpc_code *c) { return c ? c->max_dc : QLDPC_EINVAL; }
int qldpc_code_max_vn_degree(const qldpc_code *c) { return c ? c->max_dv : QLDPC_EINVAL; }
int qldpc_code_is_ira(const qldpc_code *c) { return c ? (c->ira_K > 0) : QLDPC_EINVAL; }

int qldpc_code_export_edges(const qldpc_code *c, int *var, int *chk)
{
    if (!c) return QLDPC_EINVAL;
    if (var) memcpy(var, c->cn_var, sizeof(int) * (size_t)c->E);
    if (chk) for (int m = 0; m < c->M; m++) for (int k = c->cn_ptr[m]; k < c->cn_ptr[m + 1]; k++) chk[k] = m;
    return QLDPC_OK;
}

int qldpc_code_layer_count(const qldpc_code *c) { return c ? c->n_layers : QLDPC_EINVAL; }

int qldpc_code_layer_order(const qldpc_code *c, int *check_order, int *layer_ptr)
{
    if (!c) return QLDPC_EINVAL;
    if (check_order) memcpy(check_order, c->layer_order, sizeof(int) * (size_t)c->M);
    if (layer_ptr) memcpy(layer_ptr, c->layer_ptr, sizeof(int) * ((size_t)c->n_layers + 1));
    return c->layer_natural;
}

int qldpc_code_syndrome_host(const qldpc_code *c, const int *x, int *s)
{
    if (!c || !x) return QLDPC_EINVAL;
    int w = 0;
    for (int m = 0; m < c->M; m++) {
        int p = 0;
        for (int k = c->cn_ptr[m]; k < c->cn_ptr[m + 1]; k++) p ^= x[c->cn_var[k]] & 1;
        if (s) s[m] = p;
        w += p;
    }
    return w;
}

/* ------------------------------------------------------------------ GF(2) systematic form ---- */

/*
 * Encoder_LDPC_from_H(K, N, H, "IDENTITY", ...) (VAR/main.cpp (alist-v1.0.1):142-145) needs a
 * generator for an arbitrary H.  Row-reduce H over GF(2), searching pivots in ascending column
 * order: pivot columns become the parity positions, the rest the info positions, and
 * x_pivot[j] = XOR over free columns f with A[j][f] = 1 of x_free[f].
 */
int qldpc_gf2_systematic(const qldpc_code *g, int **pivots_out, int **free_out, uint64_t **A_out, int *wpr_out)
{
    return qldpc_gf2_systematic_ord(g, 0, pivots_out, free_out, A_out, wpr_out);
}

/*
 * The same elimination with the pivot search in another column order:
 *   order 0  ascending  ("IDENTITY": the KAT's info_bits_pos = 504..1007 comes out of this one)
 *   order 1  descending ("LU_DEC" of Encoder_LDPC_from_H, VAR/main.cpp (alist-v1.0.1):135-145: parity at the END of the codeword where H allows)
 *   order 2  the last M columns only ("QC" = Encoder_LDPC_from_QC, VAR/main.cpp (qc):145: x = [u | H2^-1 H1 u]; H2 singular is an error there too)
 * Pivots come back sorted by column so that parity index j means the same thing for every order.
 */
static int cmp_int(const void *a, const void *b) { return *(const int *)a - *(const int *)b; }
int qldpc_gf2_systematic_ord(const qldpc_code *g, int order, int **pivots_out, int **free_out, uint64_t **A_out, int *wpr_out)
{
    const int N = g->N, M = g->M;
    const size_t W = ((size_t)N + 63) / 64;
    if ((double)M * (double)W * 8.0 > 2.0e9) { qldpc_set_error("IDENTITY encoder: H too large for dense elimination (%d x %d)", M, N); return QLDPC_EUNSUPPORTED; }
    uint64_t *R = (uint64_t *)calloc((size_t)M * W, sizeof(uint64_t));
    int *piv = (int *)malloc(sizeof(int) * (size_t)M);
    if (!R || !piv) { free(R); free(piv); return QLDPC_ENOMEM; }
    for (int m = 0; m < M; m++) for (int k = g->cn_ptr[m]; k < g->cn_ptr[m + 1]; k++) R[(size_t)m * W + g->cn_var[k] / 64] ^= 1ull << (g->cn_var[k] % 64);
    int r = 0;
    if (order < 0 || order > 2) { free(R); free(piv); return QLDPC_EINVAL; }
    for (int step = 0; step < N && r < M; step++) {
        const int col = order == 0 ? step : N - 1 - step;
        if (order == 2 && col < N - M) break;
        int p = -1;
        for (int m = r; m < M; m++) if (R[(size_t)m * W + col / 64] >> (col % 64) & 1) { p = m; break; }
        if (p < 0) continue;
        if (p != r) for (size_t w = 0; w < W; w++) { uint64_t t = R[(size_t)p * W + w]; R[(size_t)p * W + w] = R[(size_t)r * W + w]; R[(size_t)r * W + w] = t; }
        for (int m = 0; m < M; m++)
            if (m != r && (R[(size_t)m * W + col / 64] >> (col % 64) & 1))
                for (size_t w = order == 0 ? (size_t)col / 64 : 0; w < W; w++) R[(size_t)m * W + w] ^= R[(size_t)r * W + w];
        piv[r++] = col;
    }
    if (order == 2 && r < M) {
        free(R); free(piv);
        qldpc_set_error("QC encoder: the last %d columns of H (H2) are not invertible (rank %d)", M, r);
        return QLDPC_EUNSUPPORTED;
    }
    if (order != 0) {      /* row j of the reduced matrix belongs to pivot piv[j]: sort both by column */
        int *ord = (int *)malloc(sizeof(int) * 2 * (size_t)(r > 0 ? r : 1));
        uint64_t *R2 = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(r > 0 ? r : 1) * W);
        if (!ord || !R2) { free(ord); free(R2); free(R); free(piv); return QLDPC_ENOMEM; }
        for (int j = 0; j < r; j++) { ord[2 * j] = piv[j]; ord[2 * j + 1] = j; }
        qsort(ord, (size_t)r, 2 * sizeof(int), cmp_int);
        for (int j = 0; j < r; j++) { memcpy(R2 + (size_t)j * W, R + (size_t)ord[2 * j + 1] * W, sizeof(uint64_t) * W); piv[j] = ord[2 * j]; }
        memcpy(R, R2, sizeof(uint64_t) * (size_t)r * W);
        free(ord); free(R2);
    }
    const int K = N - r;
    int *fr = (int *)malloc(sizeof(int) * (size_t)(K > 0 ? K : 1));
    const int wpr = (K + 63) / 64;
    uint64_t *A = (uint64_t *)calloc((size_t)r * (wpr > 0 ? wpr : 1), sizeof(uint64_t));
    if (!fr || !A) { free(R); free(piv); free(fr); free(A); return QLDPC_ENOMEM; }
    for (int col = 0, pi = 0, fi = 0; col < N; col++) {
        if (pi < r && piv[pi] == col) { pi++; continue; }
        fr[fi++] = col;
    }
    for (int j = 0; j < r; j++)
        for (int fi = 0; fi < K; fi++)
            if (R[(size_t)j * W + fr[fi] / 64] >> (fr[fi] % 64) & 1) A[(size_t)j * wpr + fi / 64] |= 1ull << (fi % 64);
    free(R);
    *pivots_out = piv; *free_out = fr; *A_out = A; *wpr_out = wpr;
    return r;
}

/* ------------------------------------------------------------------ PEG construction ---------- */

/*
 * Progressive-edge-growth construction of the information part of a DVB-like IRA code
 * (what EC/ldpc_examples/improved-peg.py:136-195 and psd-peg.py set out to do):
 * variable nodes are connected one edge at a time, low degrees first; each new edge goes to a check of
 * the lowest current degree among those NOT reached from the variable node within `depth` check levels
 * of the graph built so far, so no cycle shorter than 2*(depth+1) is closed while that is possible
 * (depth 2 = no 4-cycles).  The dual-diagonal parity chain is in place from the start.  Ties are broken
 * by a seeded PRNG: both ends of a link build the same code from (N, K, profile, depth, seed).
 */
typedef struct { int *a; int n, cap; } ivec;
static int ivec_push(ivec *v, int x)
{
    if (v->n == v->cap) {
        int nc = v->cap ? v->cap * 2 : 8;
        int *na = (int *)realloc(v->a, sizeof(int) * (size_t)nc);
        if (!na) return -1;
        v->a = na; v->cap = nc;
    }
    v->a[v->n++] = x;
    return 0;
}

/*
 * Edge-list cache on disk.  Progressive edge growth is deterministic in (N, K, profile, depth, seed) but takes 0.1 - 10 s per code
 * (the depth-2 search through the degree-38 checks of a rate-0.9 mother is the slow case), and the daemon builds 32 mother codes in
 * ldpc_init: with QLDPC_CODE_CACHE=<directory> in the environment a code is built once, its edge list written there
 * (<dir>/ira_peg_N.._K.._h.._dv.._.._d.._s...edges: header, var[E], chk[E], FNV-1a checksum; written to a temporary name and renamed) and
 * read back on every later start after its header and checksum have been verified -- a file that fails either is ignored and rebuilt.
 * Alice and Bob need not share the directory: both derive the same code from the same parameters.
 */
static uint64_t fnv1a64(const void *data, size_t n, uint64_t h)
{
    const unsigned char *p = (const unsigned char *)data;
    for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 0x100000001b3ull; }
    return h;
}
#define PEG_CACHE_MAGIC 0x3147444551444c51ull      /* "QLDQEDG1" */
typedef struct { uint64_t magic; int32_t N, K, dv_hi, dv_lo, depth, E; uint32_t hi_frac_bits; uint64_t seed; } peg_cache_hdr;

static int peg_cache_path(char *buf, size_t cap, int N, int K, float hi_frac, int dv_hi, int dv_lo, int depth, uint64_t seed)
{
    const char *dir = getenv("QLDPC_CODE_CACHE");
    if (!dir || !*dir) return 0;
    const int n = snprintf(buf, cap, "%s/ira_peg_N%d_K%d_h%.6f_dv%d_%d_d%d_s%llu.edges", dir, N, K, (double)hi_frac, dv_hi, dv_lo, depth, (unsigned long long)seed);
    return n > 0 && (size_t)n < cap;
}

static int peg_cache_load(const char *path, const peg_cache_hdr *want, qldpc_code **out)
{
    FILE *f = fopen(path, "rb");
    if (!f) return 0;
    peg_cache_hdr h;
    int ok = 0, *var = NULL, *chk = NULL;
    uint64_t sum = 0;
    if (fread(&h, sizeof(h), 1, f) == 1 && !memcmp(&h, want, offsetof(peg_cache_hdr, E)) && h.hi_frac_bits == want->hi_frac_bits && h.seed == want->seed &&
        h.E > 0 && h.E < (1 << 28)) {
        var = (int *)malloc(sizeof(int) * (size_t)h.E); chk = (int *)malloc(sizeof(int) * (size_t)h.E);
        if (var && chk && fread(var, sizeof(int), (size_t)h.E, f) == (size_t)h.E && fread(chk, sizeof(int), (size_t)h.E, f) == (size_t)h.E && fread(&sum, sizeof(sum), 1, f) == 1) {
            uint64_t s2 = fnv1a64(&h, sizeof(h), 0xcbf29ce484222325ull);
            s2 = fnv1a64(var, sizeof(int) * (size_t)h.E, s2);
            s2 = fnv1a64(chk, sizeof(int) * (size_t)h.E, s2);
            ok = s2 == sum;
            for (int e = 0; ok && e < h.E; e++) ok = var[e] >= 0 && var[e] < h.N && chk[e] >= 0 && chk[e] < h.N - h.K;
            if (ok) ok = qldpc_code_from_edges(h.N, h.N - h.K, h.E, var, chk, out) == QLDPC_OK;
        }
    }
    fclose(f);
    free(var); free(chk);
    return ok;
}

static void peg_cache_store(const char *path, peg_cache_hdr *h, const qldpc_code *code)
{
    char tmp[1100];
    if (snprintf(tmp, sizeof(tmp), "%s.tmp%ld", path, (long)getpid()) >= (int)sizeof(tmp)) return;
    int *chk = (int *)malloc(sizeof(int) * (size_t)code->E);
    FILE *f = chk ? fopen(tmp, "wb") : NULL;
    if (f) {
        for (int c = 0; c < code->M; c++) for (int k = code->cn_ptr[c]; k < code->cn_ptr[c + 1]; k++) chk[k] = c;
        h->E = code->E;
        uint64_t sum = fnv1a64(h, sizeof(*h), 0xcbf29ce484222325ull);
        sum = fnv1a64(code->cn_var, sizeof(int) * (size_t)code->E, sum);
        sum = fnv1a64(chk, sizeof(int) * (size_t)code->E, sum);
        const int ok = fwrite(h, sizeof(*h), 1, f) == 1 && fwrite(code->cn_var, sizeof(int), (size_t)code->E, f) == (size_t)code->E &&
                       fwrite(chk, sizeof(int), (size_t)code->E, f) == (size_t)code->E && fwrite(&sum, sizeof(sum), 1, f) == 1;
        if (fclose(f) == 0 && ok) { if (rename(tmp, path) != 0) remove(tmp); }
        else remove(tmp);
    }
    free(chk);
}

static int ira_peg_build(int N, int K, float hi_frac, int dv_hi, int dv_lo, int depth, uint64_t seed, qldpc_code **out);

int qldpc_code_ira_peg(int N, int K, float hi_frac, int dv_hi, int dv_lo, int depth, uint64_t seed, qldpc_code **out)
{
    char path[1024];
    peg_cache_hdr h;
    if (!out) return QLDPC_EINVAL;
    *out = NULL;
    memset(&h, 0, sizeof(h));
    h.magic = PEG_CACHE_MAGIC; h.N = N; h.K = K; h.dv_hi = dv_hi; h.dv_lo = dv_lo; h.depth = depth; h.seed = seed;
    memcpy(&h.hi_frac_bits, &hi_frac, sizeof(h.hi_frac_bits));
    const int cached = peg_cache_path(path, sizeof(path), N, K, hi_frac, dv_hi, dv_lo, depth, seed);
    if (cached && peg_cache_load(path, &h, out)) return QLDPC_OK;
    const int rc = ira_peg_build(N, K, hi_frac, dv_hi, dv_lo, depth, seed, out);
    if (rc == QLDPC_OK && cached) peg_cache_store(path, &h, *out);
    return rc;
}

static int ira_peg_build(int N, int K, float hi_frac, int dv_hi, int dv_lo, int depth, uint64_t seed, qldpc_code **out)
{
    if (!out) return QLDPC_EINVAL;
    *out = NULL;
    const int M = N - K;
    if (N <= 0 || K <= 0 || M <= 1 || dv_hi < dv_lo || dv_lo < 1 || hi_frac < 0.0f || hi_frac > 1.0f || depth < 1 || depth > 4 || dv_hi > M) {
        qldpc_set_error("code_ira_peg: bad parameters N=%d K=%d depth=%d", N, K, depth);
        return QLDPC_EINVAL;
    }
    const int n_hi = (int)floor((double)hi_frac * (double)K);
    int rc = QLDPC_ENOMEM;
    ivec *cvn = (ivec *)calloc((size_t)M, sizeof(ivec));     /* VNs of each check (info + parity) */
    ivec *vcn = (ivec *)calloc((size_t)N, sizeof(ivec));     /* checks of each VN */
    int *cdeg = (int *)calloc((size_t)M, sizeof(int));       /* info degree of each check */
    int *stamp_c = (int *)calloc((size_t)M, sizeof(int)), *stamp_v = (int *)calloc((size_t)N, sizeof(int)), *lvl_c = (int *)calloc((size_t)M, sizeof(int));
    int *fr = (int *)malloc(sizeof(int) * (size_t)N), *fr2 = (int *)malloc(sizeof(int) * (size_t)N);
    ivec *bucket = NULL; int *bpos = (int *)malloc(sizeof(int) * (size_t)M);
    int maxdeg_cap = 0;
    long T = (long)n_hi * dv_hi + (long)(K - n_hi) * dv_lo;
    int *var = NULL, *chk = NULL;
    if (!cvn || !vcn || !cdeg || !stamp_c || !stamp_v || !lvl_c || !fr || !fr2 || !bpos) goto done;
    maxdeg_cap = (int)(T / M) + 64;
    bucket = (ivec *)calloc((size_t)maxdeg_cap + 1, sizeof(ivec));
    if (!bucket) goto done;
    for (int c = 0; c < M; c++) {                              /* parity chain: VN K+c on checks c and c+1 */
        if (ivec_push(&cvn[c], K + c) || ivec_push(&vcn[K + c], c)) goto done;
        if (c + 1 < M && (ivec_push(&cvn[c + 1], K + c) || ivec_push(&vcn[K + c], c + 1))) goto done;
        bpos[c] = bucket[0].n;
        if (ivec_push(&bucket[0], c)) goto done;
    }
    {
        xoshiro rng; uint64_t sm = seed;
        for (int i = 0; i < 4; i++) rng.s[i] = splitmix64(&sm);
        int tick = 0, min_d = 0;
        for (int pass = 0; pass < 2; pass++) {                 /* low degrees first (PEG order) */
            const int v_lo = pass == 0 ? n_hi : 0, v_hi = pass == 0 ? K : n_hi, dv = pass == 0 ? dv_lo : dv_hi;
            for (int v = v_lo; v < v_hi; v++) {
                for (int k = 0; k < dv; k++) {
                    /* breadth-first expansion from v: record the level at which each check is first reached */
                    tick++;
                    int nf = 1; fr[0] = v; stamp_v[v] = tick;
                    long cum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                    for (int lvl = 0; lvl < depth && nf > 0; lvl++) {
                        int nf2 = 0;
                        long newly = 0;
                        for (int i = 0; i < nf; i++) {
                            const ivec *cl = &vcn[fr[i]];
                            for (int j = 0; j < cl->n; j++) {
                                const int c = cl->a[j];
                                if (stamp_c[c] == tick) continue;
                                stamp_c[c] = tick; lvl_c[c] = lvl; newly++;
                                const ivec *vl = &cvn[c];
                                for (int t2 = 0; t2 < vl->n; t2++) { const int u = vl->a[t2]; if (stamp_v[u] != tick) { stamp_v[u] = tick; fr2[nf2++] = u; } }
                            }
                        }
                        cum[lvl] = (lvl ? cum[lvl - 1] : 0) + newly;
                        for (int l2 = lvl + 1; l2 < 8; l2++) cum[l2] = cum[lvl];
                        { int *tmp = fr; fr = fr2; fr2 = tmp; nf = nf2; }
                    }
                    /* exclude every level that still leaves a candidate: the deepest l with cum[l] < M (level 0 always) */
                    int excl = 0;
                    for (int l2 = 1; l2 < depth; l2++) if (cum[l2] < M) excl = l2;
                    /* lowest-degree check that is not excluded; random start inside the bucket */
                    int pick = -1;
                    for (int d = min_d; d <= maxdeg_cap && pick < 0; d++) {
                        const ivec *b = &bucket[d];
                        if (b->n == 0) { if (d == min_d) min_d++; continue; }
                        const int start = (int)xo_below(&rng, (uint64_t)b->n);
                        for (int i = 0; i < b->n; i++) {
                            const int c = b->a[(start + i) % b->n];
                            if (stamp_c[c] != tick || lvl_c[c] > excl) { pick = c; break; }
                        }
                    }
                    if (pick < 0) { qldpc_set_error("code_ira_peg: no admissible check for VN %d", v); rc = QLDPC_EINVAL; goto done; }
                    /* move the check one bucket up and record the edge */
                    {
                        const int d = cdeg[pick];
                        ivec *b = &bucket[d];
                        const int last = b->a[b->n - 1];
                        b->a[bpos[pick]] = last; bpos[last] = bpos[pick]; b->n--;
                        if (d + 1 > maxdeg_cap) { qldpc_set_error("code_ira_peg: degree overflow"); rc = QLDPC_EINVAL; goto done; }
                        bpos[pick] = bucket[d + 1].n;
                        if (ivec_push(&bucket[d + 1], pick)) goto done;
                        cdeg[pick] = d + 1;
                    }
                    if (ivec_push(&cvn[pick], v) || ivec_push(&vcn[v], pick)) goto done;
                }
            }
        }
    }
    {
        long E = 0;
        for (int c = 0; c < M; c++) E += cvn[c].n;
        var = (int *)malloc(sizeof(int) * (size_t)E); chk = (int *)malloc(sizeof(int) * (size_t)E);
        if (!var || !chk) goto done;
        long e = 0;
        for (int c = 0; c < M; c++) {
            int *row = cvn[c].a; const int n = cvn[c].n;
            for (int i = 1; i < n; i++) { int x = row[i], j = i - 1; while (j >= 0 && row[j] > x) { row[j + 1] = row[j]; j--; } row[j + 1] = x; }
            for (int i = 0; i < n; i++, e++) { var[e] = row[i]; chk[e] = c; }
        }
        rc = qldpc_code_from_edges(N, M, (int)E, var, chk, out);
    }
done:
    if (cvn) for (int c = 0; c < M; c++) free(cvn[c].a);
    if (vcn) for (int v = 0; v < N; v++) free(vcn[v].a);
    if (bucket) for (int d = 0; d <= maxdeg_cap; d++) free(bucket[d].a);
    free(cvn); free(vcn); free(cdeg); free(stamp_c); free(stamp_v); free(lvl_c); free(fr); free(fr2); free(bucket); free(bpos); free(var); free(chk);
    return rc;
}

/* ------------------------------------------------------------------ QC base graph by PEG + shift design ------- */

/*
 * What the reference's ldpc_examples/psd-peg.py sets out to do (its header cites the PSD-PEG paper; the script's loop is at
 * :281-414): grow a base graph of n_cols information columns of degree dv over m_rows rows by progressive edge growth
 * -- first edge to the lowest-degree row, then rows not reached from the column come first, deeper ones next, ties to the
 * lower degree (:386-389) -- and give every edge a circulant shift such that the cycle the new edge closes in the base graph
 * does not close in the lifted graph: the alternating sum of shifts around it must not vanish mod Z (:397-402).  The
 * parity part is the identity, as the script prints it (:444-447): H = [lift(P) | I], N = (n_cols + m_rows) Z.
 * Differences, on purpose: the alternating sum is accumulated along the whole tree path (the script keeps the last two
 * edges only), every 4-cycle through the new edge is tested explicitly (the tree holds one path per row), shifts are
 * drawn from [0, Z) (the script's randint(0, L) includes L), and the generator is seeded.
 */
typedef struct { int *v; int n, cap; } qcp_list;
static int qcp_push(qcp_list *l, int x)
{
    if (l->n == l->cap) { int nc = l->cap ? 2 * l->cap : 8; int *t = (int *)realloc(l->v, sizeof(int) * (size_t)nc); if (!t) return 0; l->v = t; l->cap = nc; }
    l->v[l->n++] = x;
    return 1;
}
static long qcp_mod(long a, long Z) { a %= Z; return a < 0 ? a + Z : a; }

int qldpc_code_qc_peg(int n_cols, int m_rows, int dv, int Z, uint64_t seed, const char *qc_path, qldpc_code **out, int *base_girth)
{
    if (!out) return QLDPC_EINVAL;
    *out = NULL;
    if (n_cols <= 0 || m_rows <= 0 || Z <= 1 || dv < 1 || dv > m_rows) { qldpc_set_error("qc_peg: need n_cols, m_rows > 0, Z > 1, 1 <= dv <= m_rows"); return QLDPC_EINVAL; }
    if ((long)(n_cols + m_rows) * Z > 0x3fffffffL) { qldpc_set_error("qc_peg: lifted size too large"); return QLDPC_ESIZE; }
    const int nb = n_cols + m_rows;
    int *B = (int *)malloc(sizeof(int) * (size_t)nb * m_rows);
    qcp_list *row_cols = (qcp_list *)calloc((size_t)m_rows, sizeof(qcp_list));      /* information columns of each row */
    qcp_list *col_rows = (qcp_list *)calloc((size_t)n_cols, sizeof(qcp_list));
    int *level = (int *)malloc(sizeof(int) * (size_t)m_rows), *order = (int *)malloc(sizeof(int) * (size_t)m_rows);
    long *S = (long *)malloc(sizeof(long) * (size_t)m_rows);
    int *queue = (int *)malloc(sizeof(int) * (size_t)m_rows);
    char *col_seen = (char *)malloc((size_t)n_cols);
    int rc = QLDPC_OK, girth = 0;
    if (!B || !row_cols || !col_rows || !level || !order || !S || !queue || !col_seen) { rc = QLDPC_ENOMEM; goto done; }
    for (long i = 0; i < (long)nb * m_rows; i++) B[i] = -1;
    xoshiro rng;
    { uint64_t sm = seed ^ 0x51c0de5eedull; for (int i = 0; i < 4; i++) rng.s[i] = splitmix64(&sm); }
#define QCP_B(i, j) B[(long)(i) * nb + (j)]
    for (int j = 0; j < n_cols; j++) {
        /* first edge: lowest-degree row (lowest index on ties), random shift */
        int c0 = 0;
        for (int i = 1; i < m_rows; i++) if (row_cols[i].n < row_cols[c0].n) c0 = i;
        QCP_B(c0, j) = (int)xo_below(&rng, (uint64_t)Z);
        if (!qcp_push(&row_cols[c0], j) || !qcp_push(&col_rows[j], c0)) { rc = QLDPC_ENOMEM; goto done; }
        if (dv == 1) continue;
        /* breadth-first tree from column j: level of each reached row and the alternating shift sum along its tree path */
        for (int i = 0; i < m_rows; i++) level[i] = -1;
        memset(col_seen, 0, (size_t)n_cols);
        col_seen[j] = 1;
        int qh = 0, qt = 0;
        level[c0] = 0; S[c0] = QCP_B(c0, j); queue[qt++] = c0;
        while (qh < qt) {
            const int a = queue[qh++];
            for (int x = 0; x < row_cols[a].n; x++) {
                const int k = row_cols[a].v[x];
                if (col_seen[k]) continue;
                col_seen[k] = 1;
                const long sk = S[a] - QCP_B(a, k);
                for (int y = 0; y < col_rows[k].n; y++) {
                    const int b = col_rows[k].v[y];
                    if (level[b] >= 0) continue;
                    level[b] = level[a] + 1; S[b] = sk + QCP_B(b, k); queue[qt++] = b;
                }
            }
        }
        /* rows by preference: unreached first, then deeper, then lower degree, then lower index (insertion sort, m_rows is small) */
        for (int i = 0; i < m_rows; i++) order[i] = i;
        for (int i = 1; i < m_rows; i++) {
            const int r = order[i];
            const long kr = (level[r] < 0 ? 0x40000000L : level[r]);
            int t = i - 1;
            while (t >= 0) {
                const int q = order[t];
                const long kq = (level[q] < 0 ? 0x40000000L : level[q]);
                const int before = kr > kq || (kr == kq && (row_cols[r].n < row_cols[q].n || (row_cols[r].n == row_cols[q].n && r < q)));
                if (!before) break;
                order[t + 1] = q; t--;
            }
            order[t + 1] = r;
        }
        int taken = 0;
        for (int oi = 0; oi < m_rows && taken < dv - 1; oi++) {
            const int c = order[oi];
            if (QCP_B(c, j) >= 0) continue;
            if (level[c] >= 0) { const int cyc = 2 * level[c] + 2; if (girth == 0 || cyc < girth) girth = cyc; }
            int perm = 0, ok = 0;
            for (int attempt = 0; attempt < 4096 && !ok; attempt++) {
                perm = (int)xo_below(&rng, (uint64_t)Z);
                ok = 1;
                if (level[c] >= 0 && qcp_mod(S[c] - perm, Z) == 0) ok = 0;             /* the tree path's cycle would close in the lifted graph */
                for (int y = 0; ok && y < col_rows[j].n; y++) {                          /* every 4-cycle j - a - k - c - j */
                    const int a = col_rows[j].v[y];
                    for (int x = 0; ok && x < row_cols[a].n; x++) {
                        const int k = row_cols[a].v[x];
                        if (k == j || QCP_B(c, k) < 0) continue;
                        if (qcp_mod((long)QCP_B(a, j) - QCP_B(a, k) + QCP_B(c, k) - perm, Z) == 0) ok = 0;
                    }
                }
            }
            if (!ok) { qldpc_set_error("qc_peg: no admissible shift for block (%d, %d) with Z = %d", c, j, Z); rc = QLDPC_EUNSUPPORTED; goto done; }
            QCP_B(c, j) = perm;
            if (!qcp_push(&row_cols[c], j) || !qcp_push(&col_rows[j], c)) { rc = QLDPC_ENOMEM; goto done; }
            taken++;
        }
    }
    for (int i = 0; i < m_rows; i++) QCP_B(i, n_cols + i) = 0;                           /* H2 = I */
#undef QCP_B
    if (qc_path) {
        FILE *f = fopen(qc_path, "w");
        if (!f) { qldpc_set_error("qc_peg: cannot write %s", qc_path); rc = QLDPC_EIO; goto done; }
        fprintf(f, "%d %d %d\n\n", nb, m_rows, Z);
        for (int i = 0; i < m_rows; i++) { for (int c = 0; c < nb; c++) fprintf(f, "%d ", B[(long)i * nb + c]); fprintf(f, "\n"); }
        fclose(f);
    }
    rc = qc_expand(nb, m_rows, Z, B, out);
    if (base_girth) *base_girth = girth;      /* shortest cycle an edge closed in the BASE graph (0: none closed) */
done:
    if (row_cols) for (int i = 0; i < m_rows; i++) free(row_cols[i].v);
    if (col_rows) for (int i = 0; i < n_cols; i++) free(col_rows[i].v);
    free(row_cols); free(col_rows); free(B); free(level); free(order); free(S); free(queue); free(col_seen);
    return rc;
}
