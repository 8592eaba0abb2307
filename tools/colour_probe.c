/* developer probe: colourings of the check-conflict graph of an IRA code -- first-fit, largest-degree-first, iterated greedy, balancing passes.
 * gcc -O2 -std=gnu11 -Iinclude -o /tmp/colour_probe tools/colour_probe.c -Lqcrypto-ldpc_amd -lqldpc -Wl,-rpath,$PWD/qcrypto-ldpc_amd -Wl,-rpath,/opt/rocm/lib -lm ; /tmp/colour_probe 1000000 800000 [wp] */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "qldpc.h"
/* iterated greedy colouring of the check-conflict graph (two checks conflict when they share a VN) */
static int N, M, E; static int *cvar, *cchk, *cptr, *vptr, *vchk;
static int greedy(const int *order, int *col, int balance, int kmax, int *sizes)
{
    /* colours used by neighbours collected via stamp array */
    static int *stamp = NULL; static int tick = 0;
    if (!stamp) stamp = calloc(4096, sizeof(int));
    int k = 0;
    for (int c = 0; c < M; c++) col[c] = -1;
    if (sizes) memset(sizes, 0, sizeof(int) * 4096);
    for (int i = 0; i < M; i++) {
        const int c = order[i];
        tick++;
        for (int j = cptr[c]; j < cptr[c + 1]; j++) { const int v = cvar[j]; for (int t = vptr[v]; t < vptr[v + 1]; t++) { const int d = vchk[t]; if (col[d] >= 0) stamp[col[d]] = tick; } }
        int pick = -1;
        if (!balance) { for (int q = 0; q < 4096; q++) if (stamp[q] != tick) { pick = q; break; } }
        else { int best = 1 << 30; for (int q = 0; q < kmax; q++) if (stamp[q] != tick && sizes[q] < best) { best = sizes[q]; pick = q; } if (pick < 0) for (int q = kmax; q < 4096; q++) if (stamp[q] != tick) { pick = q; break; } }
        col[c] = pick; if (sizes) sizes[pick]++;
        if (pick + 1 > k) k = pick + 1;
    }
    return k;
}
int main(int argc, char **argv)
{
    int n = argc > 1 ? atoi(argv[1]) : 1000000, kk = argc > 2 ? atoi(argv[2]) : 800000;
    qldpc_code *code; if (qldpc_code_ira(n, kk, 0.125f, 11, 3, 7, &code)) return 1;
    N = qldpc_code_n(code); M = qldpc_code_m(code); E = qldpc_code_e(code);
    cvar = malloc(4 * E); cchk = malloc(4 * E); qldpc_code_export_edges(code, cvar, cchk);
    cptr = calloc(M + 1, 4); vptr = calloc(N + 1, 4); vchk = malloc(4 * E);
    for (int e = 0; e < E; e++) { cptr[cchk[e] + 1]++; vptr[cvar[e] + 1]++; }
    for (int c = 0; c < M; c++) cptr[c + 1] += cptr[c];
    for (int v = 0; v < N; v++) vptr[v + 1] += vptr[v];
    int *fill = calloc(N, 4); for (int e = 0; e < E; e++) vchk[vptr[cvar[e]] + fill[cvar[e]]++] = cchk[e];
    printf("N %d M %d E %d layers(lib) %d\n", N, M, E, qldpc_code_layer_count(code));
    int *order = malloc(4 * M), *col = malloc(4 * M), *sizes = calloc(4096, 4), *cnt = calloc(4097, 4), *ord2 = malloc(4 * M);
    for (int c = 0; c < M; c++) order[c] = c;
    if (argc > 3) {
        long *deg = malloc(8 * M);
        for (int c = 0; c < M; c++) { long d = 0; for (int j = cptr[c]; j < cptr[c + 1]; j++) d += vptr[cvar[j] + 1] - vptr[cvar[j]] - 1; deg[c] = d * 1000000L + (M - c); }
        int cmp(const void *a, const void *b) { long x = deg[*(const int *)a], y = deg[*(const int *)b]; return x < y ? 1 : (x > y ? -1 : 0); }
        qsort(order, M, sizeof(int), cmp);
        printf("welsh-powell order, max conflict degree %ld min %ld\n", deg[order[0]] / 1000000L, deg[order[M - 1]] / 1000000L);
    }
    int k = greedy(order, col, 0, 0, sizes);
    printf("first-fit: %d\n", k);
    for (int it = 0; it < 9; it++) {
        /* reorder: colour classes contiguous; alternate: reverse class order / largest class first / smallest first */
        int idx[4096]; for (int q = 0; q < k; q++) idx[q] = q;
        const int mode = it % 3;
        if (mode == 0) { for (int q = 0; q < k / 2; q++) { int t = idx[q]; idx[q] = idx[k - 1 - q]; idx[k - 1 - q] = t; } }
        else { for (int a = 0; a < k; a++) for (int b = a + 1; b < k; b++) if ((mode == 1) ? sizes[idx[b]] > sizes[idx[a]] : sizes[idx[b]] < sizes[idx[a]]) { int t = idx[a]; idx[a] = idx[b]; idx[b] = t; } }
        int pos[4096], at = 0; for (int q = 0; q < k; q++) { pos[idx[q]] = at; at += sizes[idx[q]]; }
        for (int c = 0; c < M; c++) ord2[pos[col[c]]++] = c;
        memcpy(order, ord2, 4 * M);
        int k2 = greedy(order, col, 0, 0, sizes);
        if (k2 != k || it % 10 == 9) printf("iter %d mode %d: %d colours\n", it, mode, k2);
        k = k2;
    }
    /* balance within k colours */
    for (int it = 0; it < 4; it++) {
        int pos[4096], at = 0; for (int q = 0; q < k; q++) { pos[q] = at; at += sizes[q]; }
        for (int c = 0; c < M; c++) ord2[pos[col[c]]++] = c;
        memcpy(order, ord2, 4 * M);
        int k2 = greedy(order, col, 1, k, sizes);
        int mn = 1 << 30, mx = 0; for (int q = 0; q < k2; q++) { if (sizes[q] < mn) mn = sizes[q]; if (sizes[q] > mx) mx = sizes[q]; }
        printf("balance pass %d: %d colours, class sizes %d .. %d\n", it, k2, mn, mx); k = k2;
    }
    return 0;
}
