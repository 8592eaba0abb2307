/* Sanitizer pass over the plain-C graph layer (no HIP): built with -fsanitize=address,undefined by tests/test_sanitize.py */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "qldpc.h"
#include "../../qcrypto-ldpc_amd/csrc/qldpc_graph.h"

#define CHECK(x) do { if (!(x)) { fprintf(stderr, "FAILED %s:%d: %s (%s)\n", __FILE__, __LINE__, #x, qldpc_last_error()); return 1; } } while (0)

int main(int argc, char **argv)
{
    const char *gold = argc > 1 ? argv[1] : "tests/golden";
    char path[512];
    qldpc_code *c = NULL;
    snprintf(path, sizeof(path), "%s/PEGReg504x1008.alist", gold);
    CHECK(qldpc_code_from_alist(path, &c) == QLDPC_OK && qldpc_code_n(c) == 1008 && qldpc_code_e(c) == 3024);
    {
        int *piv, *fr, wpr; uint64_t *A;
        const int r = qldpc_gf2_systematic(c, &piv, &fr, &A, &wpr);
        CHECK(r == 504 && piv[0] == 0 && piv[503] == 503 && fr[0] == 504);
        free(piv); free(fr); free(A);
        int *order = malloc(sizeof(int) * 504), *ptr = malloc(sizeof(int) * ((size_t)qldpc_code_layer_count(c) + 1));
        CHECK(qldpc_code_layer_order(c, order, ptr) >= 0);
        free(order); free(ptr);
    }
    qldpc_code_free(c);
    snprintf(path, sizeof(path), "%s/NR_2_3_112.qc", gold);
    CHECK(qldpc_code_from_qc(path, &c) == QLDPC_OK);
    qldpc_code_free(c);
    snprintf(path, sizeof(path), "%s/does-not-exist.alist", gold);
    CHECK(qldpc_code_from_alist(path, &c) == QLDPC_EIO && c == NULL);
    CHECK(qldpc_code_ira(65536, 52429, 0.125f, 11, 3, 7, &c) == QLDPC_OK && qldpc_code_e(c) == 235925 && qldpc_code_is_ira(c));
    {
        int *var = malloc(sizeof(int) * 235925), *chk = malloc(sizeof(int) * 235925);
        CHECK(qldpc_code_export_edges(c, var, chk) == QLDPC_OK);
        qldpc_code *c2 = NULL;
        CHECK(qldpc_code_from_edges(65536, 13107, 235925, var, chk, &c2) == QLDPC_OK && qldpc_code_is_ira(c2));
        var[5] = var[4];   /* duplicate edge inside a check */
        chk[5] = chk[4];
        qldpc_code *c3 = NULL;
        CHECK(qldpc_code_from_edges(65536, 13107, 235925, var, chk, &c3) == QLDPC_EINVAL && c3 == NULL);
        qldpc_code_free(c2); free(var); free(chk);
    }
    qldpc_code_free(c);
    CHECK(qldpc_code_ira_peg(8192, 6554, 0.125f, 11, 3, 2, 7, &c) == QLDPC_OK && qldpc_code_is_ira(c));
    qldpc_code_free(c);
    CHECK(qldpc_code_ira_peg(8192, 6554, 0.125f, 11, 3, 3, 9, &c) == QLDPC_OK);
    qldpc_code_free(c);
    CHECK(qldpc_code_ira(1000, 900, 0.4f, 14, 4, 3, &c) == QLDPC_OK);     /* high rate, many repairs */
    qldpc_code_free(c);
    CHECK(qldpc_code_ira(100, 99, 0.1f, 3, 3, 1, &c) == QLDPC_EINVAL);
    {
        int girth = -1;
        CHECK(qldpc_code_qc_peg(12, 6, 3, 601, 3, NULL, &c, &girth) == QLDPC_OK && qldpc_code_n(c) == 18 * 601 && (girth == 4 || girth == 6));
        qldpc_code_free(c);
        CHECK(qldpc_code_qc_peg(40, 8, 4, 53, 1, "/tmp/qldpc_sanitize.qc", &c, NULL) == QLDPC_OK);
        qldpc_code_free(c);
        CHECK(qldpc_code_from_qc("/tmp/qldpc_sanitize.qc", &c) == QLDPC_OK && qldpc_code_m(c) == 8 * 53);
        qldpc_code_free(c);
        remove("/tmp/qldpc_sanitize.qc");
        CHECK(qldpc_code_qc_peg(4, 2, 2, 2, 1, NULL, &c, NULL) != QLDPC_OK && c == NULL);      /* Z = 2: no admissible shifts */
        CHECK(qldpc_code_qc_peg(12, 6, 7, 601, 3, NULL, &c, NULL) == QLDPC_EINVAL);
    }
    CHECK(qldpc_parity_bits_to_punct(64800, 48600, 0.8f) == 4050);
    printf("graph layer: sanitizer pass ok\n");
    return 0;
}
