/*
 * qldpc_oracle.h -- CPU ORACLE for the LDPC reconciliation path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C, scalar float32 restatement of the decoder the reference harness calls:
 * AFF3CT v2.3.5 (git 1ceddfc, NOT vendored in /root/reference; pinned by
 * errorcorrection/ldpc_examples/my_project_with_aff3ct/ci/build-linux-macos.sh:34-35,56).
 * Call sites that fix its semantics:
 *   BS/src/main.cpp:172-195 (modules), :335-393 (frame loop)
 *   VAR/main.cpp (alist-v1.0.1):119-285 (decoder ctors), :408-465 (frame loop + KAT)
 *   VAR/main.cpp (dvb-v1.0.2):179-313,427-458
 * where BS = errorcorrection/ldpc_examples/my_project_with_aff3ct/examples/bootstrap and
 * VAR = BS/src/variants (copy out as main.cpp to use).
 *
 * PARITY PIN: the reference's embedded known-answer vector
 *   VAR/main.cpp (alist-v1.0.1):445,447,456,460 on BS/matrices/H/PEGReg504x1008.alist
 * (tests/golden/kat_peg504x1008.json).  Flooding SPA is pinned by it exactly; MS/OMS/NMS and the
 * layered schedule reproduce the same decoded word but have no reference-held vector of their own;
 * LSPA / AMS rules are "parity unpinned" (restated from the published AFF3CT algorithm only).
 * Further pins held by tests/test_oracle.py: the float layered schedule against a double-precision restatement of the
 * reference's MATLAB recursion (ldpc_examples/.../BPSK_nrldpc_sim.m:29-69); the integer layered decoder (orc_decode_i8)
 * against a literal restatement of its fixed-point MATLAB decoder (BPSK_nrldpc_sim_RM_FP.m:37-98, exact) and against the
 * frame-error table the reference publishes for it (sim_results.m:8-13); the LFSR against the reference's own rnd.c.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (libqldpc.so) never links, loads or calls it.
 */
#ifndef QLDPC_ORACLE_H
#define QLDPC_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* Update rules (AFF3CT tools::Update_rule_*; VAR/main.cpp (alist-v1.0.1):203-218) */
enum {
    ORC_RULE_MS = 0,       /* min-sum                                   */
    ORC_RULE_OMS = 1,      /* offset min-sum, param = offset            */
    ORC_RULE_NMS = 2,      /* normalised min-sum, param = factor        */
    ORC_RULE_SPA = 3,      /* sum-product (tanh / atanh)                */
    ORC_RULE_LSPA = 4,     /* log sum-product                           */
    ORC_RULE_AMS_MIN = 5,  /* approximate min*, MIN = min               */
    ORC_RULE_AMS_MINSTAR_L2 = 6, /* MIN = min_star_linear2              */
    ORC_RULE_AMS_MINSTAR = 7     /* MIN = min_star                      */
};

/* OR into `rule`: store var_to_chk / chk_to_var rounded to IEEE binary16 (round-to-nearest-even), arithmetic in
 * fp32 -- mirrors the product's fp16 message-storage mode; not an AFF3CT configuration of the reference harness */
#define ORC_MSG_FP16 0x100

/* Schedules (AFF3CT module::Decoder_LDPC_BP_{flooding,horizontal_layered}) */
enum { ORC_SCHED_FLOODING = 0, ORC_SCHED_HLAYERED = 1 };

typedef struct orc_graph orc_graph;

/* Build from a list of (var, chk) connections in AFF3CT add_connection() order. */
orc_graph *orc_graph_from_edges(int N, int M, int E, const int *var, const int *chk);
/* MacKay alist (first part = per-VN lists drives insertion order) / AFF3CT .qc readers. */
orc_graph *orc_graph_from_alist(const char *path);
orc_graph *orc_graph_from_qc(const char *path);
void orc_graph_free(orc_graph *g);
int orc_graph_N(const orc_graph *g);
int orc_graph_M(const orc_graph *g);
int orc_graph_E(const orc_graph *g);
int orc_graph_max_cn_degree(const orc_graph *g);
int orc_graph_max_vn_degree(const orc_graph *g);
/* Copy out the CN-major structure: cn_ptr[M+1], cn_var[E]; and VN-major: vn_ptr[N+1], vn_chk[E]. */
void orc_graph_export(const orc_graph *g, int *cn_ptr, int *cn_var, int *vn_ptr, int *vn_chk, int *transpose);

/*
 * decode_siho for n_frames independent frames (each starts from a reset() decoder).
 *   Y_N   [n_frames][N]  channel LLRs
 *   post  [n_frames][N]  (optional, may be NULL) final a-posteriori LLRs
 *   hard  [n_frames][N]  (optional) hard decision of every VN: !(post >= 0)
 *   iters [n_frames]     (optional) iterations executed (1-based count of CN sweeps)
 *   synd_ok [n_frames]   (optional) 1 if H*hard == 0 at exit
 * n_threads > 1 uses OpenMP over frames.  Returns 0, or a negative code on bad arguments.
 */
int orc_decode(const orc_graph *g, int schedule, int rule, float rule_param, int n_ite,
               int enable_syndrome, int syndrome_depth, const float *Y_N, int n_frames,
               float *post, int *hard, int *iters, int *synd_ok, int n_threads);

/* Coset ("syndrome form") decoding: target[n_frames][M] of 0/1; frame f must satisfy H x = target[f]. */
int orc_decode_coset(const orc_graph *g, int schedule, int rule, float rule_param, int n_ite,
                     int enable_syndrome, int syndrome_depth, const float *Y_N, const int *target, int n_frames,
                     float *post, int *hard, int *iters, int *synd_ok, int n_threads);

/* 8-bit fixed-point min-sum, flooding or horizontal layered (rule MS / OMS / NMS only; layered keeps the posterior in 8 bits too): channel LLRs quantised with quant_scale, messages
 * saturating at +-127, integer arithmetic; target may be NULL (H x = 0).  post_out holds the integer posteriors as floats. */
int orc_decode_i8(const orc_graph *g, int schedule, int rule, float rule_param, float quant_scale, int n_ite, int enable_syndrome, int syndrome_depth,
                  const float *Y_N, const int *target, int n_frames, float *post_out, int *hard, int *iters, int *synd_ok, int n_threads);

/*
 * Privacy amplification hash, restating privAmp_doPrivAmp's loop (EC/subcomponents/priv_amp.c:213-218)
 * and the bit-serial LFSR of rnd_getPrngValue2_32 (EC/subcomponents/rnd.c:118-127, PRNG_FEEDBACK
 * 0xe0000200 rnd.h:46).  Pinned against the reference's own rnd.c compiled into oracle/_ref/librefrnd.so.
 */
unsigned int orc_lfsr32(unsigned int *state);
void orc_privamp(const unsigned int *key_words, int workbits, unsigned int seed, int final_bits, unsigned int *final_words);

/* H * x over GF(2) for one word x[N] of 0/1 ints -> s[M]. Returns syndrome weight. */
int orc_syndrome(const orc_graph *g, const int *x, int *s);

#ifdef __cplusplus
}
#endif
#endif
