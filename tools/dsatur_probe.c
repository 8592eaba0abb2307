/* developer probe: DSATUR colouring of the check-conflict graph of an IRA code (two checks conflict when they share a VN) --
 * how many VN-disjoint layers a horizontal-layered sweep needs.  A check's forbidden colours are the OR of its VNs' used-colour sets; saturation = popcount.
 * gcc -O2 -std=gnu11 -Iinclude -o /tmp/dsatur_probe tools/dsatur_probe.c -Lqcrypto-ldpc_amd -lqldpc -Wl,-rpath,$PWD/qcrypto-ldpc_amd -Wl,-rpath,/opt/rocm/lib -lm ; /tmp/dsatur_probe 1000000 800000 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "qldpc.h"
static int N, M, E; static int *cvar, *cchk, *cptr, *vptr, *vchk;
int main(int argc, char **argv)
{
    int n = argc > 1 ? atoi(argv[1]) : 1000000, kk = argc > 2 ? atoi(argv[2]) : 800000;
    qldpc_code *code; if (qldpc_code_ira(n, kk, 0.125f, 11, 3, 7, &code)) return 1;
    N = qldpc_code_n(code); M = qldpc_code_m(code); E = qldpc_code_e(code);
    cvar = malloc(4 * (size_t)E); cchk = malloc(4 * (size_t)E); qldpc_code_export_edges(code, cvar, cchk);
    cptr = calloc(M + 1, 4); vptr = calloc(N + 1, 4); vchk = malloc(4 * (size_t)E);
    int *cv = malloc(4 * (size_t)E);
    for (int e = 0; e < E; e++) { cptr[cchk[e] + 1]++; vptr[cvar[e] + 1]++; }
    for (int c = 0; c < M; c++) cptr[c + 1] += cptr[c];
    for (int v = 0; v < N; v++) vptr[v + 1] += vptr[v];
    int *fv = calloc(N, 4), *fc = calloc(M, 4);
    for (int e = 0; e < E; e++) { vchk[vptr[cvar[e]] + fv[cvar[e]]++] = cchk[e]; cv[cptr[cchk[e]] + fc[cchk[e]]++] = cvar[e]; }
    printf("N %d M %d E %d layers(lib, first-fit) %d\n", N, M, E, qldpc_code_layer_count(code));
    uint64_t *used = calloc(N, 8);      /* colours taken on each VN (<= 64 colours) */
    int *col = malloc(4 * (size_t)M), *sat = calloc(M, 4);
    long *deg = malloc(8 * (size_t)M);
    for (int c = 0; c < M; c++) { col[c] = -1; long d = 0; for (int j = cptr[c]; j < cptr[c + 1]; j++) d += vptr[cv[j] + 1] - vptr[cv[j]] - 1; deg[c] = d; }
    /* bucket queue by saturation: doubly linked lists */
    int *nxt = malloc(4 * (size_t)M), *prv = malloc(4 * (size_t)M), head[65];
    for (int s = 0; s < 65; s++) head[s] = -1;
    /* insert in order of conflict degree so that ties inside a bucket prefer high degree: sort descending, push to front in ascending order */
    int *order = malloc(4 * (size_t)M); for (int c = 0; c < M; c++) order[c] = c;
    int cmp(const void *a, const void *b) { long x = deg[*(const int *)a], y = deg[*(const int *)b]; return x < y ? -1 : (x > y ? 1 : 0); }
    qsort(order, M, sizeof(int), cmp);
    for (int i = 0; i < M; i++) { const int c = order[i]; nxt[c] = head[0]; prv[c] = -1; if (head[0] >= 0) prv[head[0]] = c; head[0] = c; }
    int top = 0, k = 0; int sizes[64] = {0};
    for (int done = 0; done < M; done++) {
        while (top > 0 && head[top] < 0) top--;
        int s = 64; while (s >= 0 && head[s] < 0) s--;
        const int c = head[s];
        /* unlink */
        head[s] = nxt[c]; if (nxt[c] >= 0) prv[nxt[c]] = -1;
        uint64_t forb = 0; for (int j = cptr[c]; j < cptr[c + 1]; j++) forb |= used[cv[j]];
        int q = 0; while ((forb >> q) & 1ull) q++;
        if (q >= 64) { printf("more than 64 colours\n"); return 1; }
        col[c] = q; sizes[q]++; if (q + 1 > k) k = q + 1;
        for (int j = cptr[c]; j < cptr[c + 1]; j++) {
            const int v = cv[j];
            if ((used[v] >> q) & 1ull) continue;
            used[v] |= 1ull << q;
            for (int t = vptr[v]; t < vptr[v + 1]; t++) {
                const int d = vchk[t];
                if (col[d] >= 0) continue;
                uint64_t f = 0; for (int jj = cptr[d]; jj < cptr[d + 1]; jj++) f |= used[cv[jj]];
                const int ns = __builtin_popcountll(f);
                if (ns != sat[d]) {
                    const int os = sat[d];
                    if (prv[d] >= 0) nxt[prv[d]] = nxt[d]; else head[os] = nxt[d];
                    if (nxt[d] >= 0) prv[nxt[d]] = prv[d];
                    nxt[d] = head[ns]; prv[d] = -1; if (head[ns] >= 0) prv[head[ns]] = d; head[ns] = d;
                    sat[d] = ns;
                }
            }
        }
    }
    printf("DSATUR: %d colours; sizes:", k);
    for (int q = 0; q < k; q++) printf(" %d", sizes[q]);
    printf("\n");
    /* iterated greedy on top: first-fit over the checks class by class (never more colours than before), class order varied */
    int *ord2 = malloc(4 * (size_t)M), *col2 = malloc(4 * (size_t)M);
    unsigned rs = 12345u;
    for (int it = 0; it < 12; it++) {
        int idx[64]; for (int q = 0; q < k; q++) idx[q] = q;
        const int mode = it % 3;
        if (mode == 0) { for (int q = 0; q < k / 2; q++) { int t = idx[q]; idx[q] = idx[k - 1 - q]; idx[k - 1 - q] = t; } }
        else if (mode == 1) { for (int a = 0; a < k; a++) for (int b = a + 1; b < k; b++) if (sizes[idx[b]] < sizes[idx[a]]) { int t = idx[a]; idx[a] = idx[b]; idx[b] = t; } }      /* smallest class first */
        else { for (int a = k - 1; a > 0; a--) { rs = rs * 1664525u + 1013904223u; int b = (int)((rs >> 8) % (unsigned)(a + 1)); int t = idx[a]; idx[a] = idx[b]; idx[b] = t; } }
        int pos[64], at = 0; for (int q = 0; q < k; q++) { pos[idx[q]] = at; at += sizes[idx[q]]; }
        for (int c = 0; c < M; c++) ord2[pos[col[c]]++] = c;
        memset(used, 0, 8 * (size_t)N);
        int k2 = 0, sz2[64] = {0};
        for (int i = 0; i < M; i++) {
            const int c = ord2[i];
            uint64_t forb = 0; for (int j = cptr[c]; j < cptr[c + 1]; j++) forb |= used[cv[j]];
            int q = 0; while ((forb >> q) & 1ull) q++;
            col2[c] = q; sz2[q]++; if (q + 1 > k2) k2 = q + 1;
            for (int j = cptr[c]; j < cptr[c + 1]; j++) used[cv[j]] |= 1ull << q;
        }
        printf("  iterated greedy pass %d (mode %d): %d colours, smallest classes %d %d\n", it, mode, k2, sz2[k2 - 1], k2 > 1 ? sz2[k2 - 2] : 0);
        memcpy(col, col2, 4 * (size_t)M); k = k2; memcpy(sizes, sz2, sizeof(sizes));
    }
    return 0;
}
