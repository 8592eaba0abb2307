/*
 * qldpc.h -- C ABI of libqldpc.so: MI355X (gfx950) batched LDPC belief-propagation reconciliation.
 *
 * This is the drop-in boundary for the LDPC path of JarryChou/qcrypto-ldpc.  Plain C types only
 * (pointers, sizes, ints, floats); no C++/torch types.  Every entry point returns QLDPC_OK (0) or a
 * negative qldpc_status; nothing throws.  One decoder instance is not re-entrant (neither is an
 * AFF3CT module); use one per host thread / stream.
 *
 * Reference interfaces replaced (paths under /root/reference/errorcorrection/, BS =
 * ldpc_examples/my_project_with_aff3ct/examples/bootstrap, VAR = BS/src/variants (copy out as main.cpp to use)):
 *
 *   qldpc_code_from_alist / _from_qc      tools::LDPC_matrix_handler::read_matrix_size / read
 *                                         VAR/main.cpp (alist-v1.0.1):324,338 ; VAR/main.cpp (qc):145
 *   qldpc_code_ira                        tools::build_dvbs2 + tools::build_H        BS/src/main.cpp:175-176
 *                                         (DVB-S2 tables are AFF3CT built-ins and not in the tree; this
 *                                         builds a DVB-like IRA code of any (N,K) instead)
 *   qldpc_code_max_cn_degree              H.get_cols_max_degree()                    BS/src/main.cpp:178
 *   qldpc_decoder_create                  module::Decoder_LDPC_BP_flooding<B,Q,Rule>(K, N, n_ite, H, info_bits_pos,
 *                                         rule(param), enable_syndrome, syndrome_depth, n_frames)
 *                                         BS/src/main.cpp:193 ; VAR/main.cpp (alist-v1.0.1):179-256
 *   qldpc_decode_siho                     decoder->decode_siho(LLRs, dec_bits)       BS/src/main.cpp:365
 *   qldpc_decoder_reset                   (*(m.decoder)).reset()                     BS/src/main.cpp:389
 *   qldpc_llr_from_ber, QLDPC_CONFIRMED_BIT_LLR   LLR(BER), CONFIRMED_BIT_LLR        BS/src/main.cpp:19-20
 *   qldpc_load_bits_*  (frame formation)  modem->demodulate + parity pinning + puncturing
 *                                         BS/src/main.cpp:348-362 ; VAR/main.cpp (5g-qc):514-531
 *   qldpc_encoder_* / qldpc_encode_*      m.encoder->encode(ref_bits, enc_bits)      BS/src/main.cpp:341 ;
 *                                         Encoder_LDPC_from_H(K,N,H,"IDENTITY",...)  VAR/main.cpp (alist-v1.0.1):142-145
 *   qldpc_min_code_rate, qldpc_parity_bits_to_punct   min_cr(), parity_bits_to_punct()   BS/src/main.cpp:23-34
 *   packed-bit layout (bit i <-> word[i/32] & (1u << (31 - i%32)))                  subcomponents/helpers.h:65-70
 *   qldpc_code_ira_peg / qldpc_code_qc_peg   ldpc_examples/improved-peg.py:136-195 / psd-peg.py:281-447 (H-matrix construction)
 *   qldpc_decoder_cfg.msg_dtype = 2       the fixed-point layered min-sum of ldpc_examples/.../BPSK_nrldpc_sim_RM_FP.m:37-98
 *   qldpc_recon_* (sessions)              what the two `return 81` arms of subcomponents/qber_estim.c:337-340,420-423 need:
 *                                         rate choice (BS/src/main.cpp:29,235-266), frame formation, verification; bound by
 *                                         qcrypto-ldpc_amd/host/ldpc_reconcile.c (packet handlers, cascade fallback, batched ingest)
 *   qldpc_privamp*                        the hash loop of privAmp_doPrivAmp         subcomponents/priv_amp.c:190-218
 */
#ifndef QLDPC_H
#define QLDPC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QLDPC_VERSION 100

typedef enum qldpc_status {
    QLDPC_OK = 0,
    QLDPC_EINVAL = -1,       /* bad argument                                                    */
    QLDPC_ENOMEM = -2,       /* host or device allocation failed                                */
    QLDPC_EIO = -3,          /* matrix file unreadable / malformed                              */
    QLDPC_EHIP = -4,         /* a HIP call failed (qldpc_last_error() has the text)             */
    QLDPC_ENODEV = -5,       /* no gfx950-capable device visible                                */
    QLDPC_ESIZE = -6,        /* size mismatch (AFF3CT throws tools::length_error here)          */
    QLDPC_EUNSUPPORTED = -7, /* valid request this build does not implement                     */
    QLDPC_ESTATE = -8,       /* call sequence error (e.g. run before load)                      */
    QLDPC_EDECODE = -9       /* reconciliation failed: no codeword found or CRC mismatch        */
} qldpc_status;

/* tools::Update_rule_* selected at VAR/main.cpp (alist-v1.0.1):203-218 */
typedef enum qldpc_rule {
    QLDPC_RULE_MS = 0,            /* Update_rule_MS                                  */
    QLDPC_RULE_OMS = 1,           /* Update_rule_OMS(offset)        param = offset   */
    QLDPC_RULE_NMS = 2,           /* Update_rule_NMS(norm_factor)   param = factor   */
    QLDPC_RULE_SPA = 3,           /* Update_rule_SPA(max_CN_degree)                  */
    QLDPC_RULE_LSPA = 4,          /* Update_rule_LSPA                                */
    QLDPC_RULE_AMS_MIN = 5,       /* Update_rule_AMS<min>                            */
    QLDPC_RULE_AMS_MINSTAR_L2 = 6,/* Update_rule_AMS<min_star_linear2>               */
    QLDPC_RULE_AMS_MINSTAR = 7    /* Update_rule_AMS<min_star>                       */
} qldpc_rule;

/* Decoder_LDPC_BP_{flooding,horizontal_layered}; VAR/main.cpp (alist-v1.0.1):203-237 */
typedef enum qldpc_schedule {
    QLDPC_SCHED_FLOODING = 0,
    QLDPC_SCHED_HLAYERED = 1      /* horizontal layered, checks visited in the code's layer order */
#define QLDPC_RECON_SCHED_AUTO 2  /* qldpc_recon_cfg.schedule only: chosen by the batch size of the session's decoders */
} qldpc_schedule;

/*
 * Two kernel families behind the same calls:
 *   FRAMES  frame-interleaved: a wavefront lane = a frame; thousands of frames per launch (HBM-bound)
 *   EDGES   edge-parallel: lanes run over the edges of ONE frame, check groups staged in LDS, min/sign
 *           fold by wavefront shuffles; the daemon's one-block-at-a-time case (flooding; MS/OMS/NMS/SPA;
 *           check degree <= 64)
 */
typedef enum qldpc_engine { QLDPC_ENGINE_AUTO = 0, QLDPC_ENGINE_FRAMES = 1, QLDPC_ENGINE_EDGES = 2 } qldpc_engine;

/* Per-VN class for QKD frame formation (BS/src/main.cpp:348-362) */
enum {
    QLDPC_VN_CHANNEL = 0,   /* sifted-key bit seen through the BSC:  LLR = (1-2y) ln((1-p)/p)      */
    QLDPC_VN_PINNED = 1,    /* bit disclosed by Alice (parity):      LLR = y ? -23.02585 : +23.02585 */
    QLDPC_VN_PUNCTURED = 2  /* punctured parity VN:                  LLR = 0                         */
};

#define QLDPC_CONFIRMED_BIT_LLR 23.025850929840455f /* -log(1e-10 / (1 - 1e-10)), BS/src/main.cpp:19 */

typedef struct qldpc_code qldpc_code;
typedef struct qldpc_decoder qldpc_decoder;
typedef struct qldpc_encoder qldpc_encoder;

/* ------------------------------------------------------------------ library ------------------ */
int qldpc_version(void);
const char *qldpc_strerror(int status);
const char *qldpc_last_error(void);           /* thread-local text of the last failure          */
int qldpc_device_count(void);                 /* HIP devices visible (0 on a CPU-only host)     */
/* Measurement aid (bench.py): the rate at which `device` copies `bytes` (read once + written once, non-temporal) in the decoder's own
 * access shape -- wide = 0: 256-byte rows, one dword per lane, eight rows in flight per wavefront; wide = 1: 16 bytes per lane.
 * The figure counts bytes read + bytes written, like the kernels' rooflines.  No reference counterpart. */
int qldpc_copy_probe(int device, size_t bytes, int reps, int wide, double *gbytes_per_s);

/* ------------------------------------------------------------------ scalar helpers ----------- */
float qldpc_llr_from_ber(float ber);                          /* LLR(BER) = -log(p/(1-p)) in double, BS/src/main.cpp:20 */
float qldpc_bsc_llr(float ber);                               /* Modem_OOK_BSC: logf((1-p)/p) in float, BS/src/main.cpp:317,348 */
float qldpc_binary_entropy(float q);                          /* h(q), BS/src/main.cpp:23-26       */
float qldpc_min_code_rate(float qber, float efficiency);      /* 1/(1+f h(q))... BS/src/main.cpp:29 */
int qldpc_parity_bits_to_punct(int N, int K, float target_cr);/* BS/src/main.cpp:34                */

/* ------------------------------------------------------------------ code (H matrix) ---------- */
int qldpc_code_from_alist(const char *path, qldpc_code **out);
int qldpc_code_from_qc(const char *path, qldpc_code **out);
/* (var[e], chk[e]) in add_connection order: that order is each check's edge order. */
int qldpc_code_from_edges(int N, int M, int E, const int *var, const int *chk, qldpc_code **out);
/*
 * DVB-like IRA code: info VNs 0..K-1 (the first hi_frac*K have degree dv_hi, the rest dv_lo),
 * parity VNs K..N-1 on a dual diagonal, every check the same number of info edges, seeded
 * socket shuffle.  (N=65536,K=52429,hi_frac=.125,dv_hi=11,dv_lo=3,seed=7) is BASELINE config 2.
 */
int qldpc_code_ira(int N, int K, float hi_frac, int dv_hi, int dv_lo, uint64_t seed, qldpc_code **out);
/*
 * The same profile with the information part built by progressive edge growth (what the reference's
 * ldpc_examples/improved-peg.py:136-195 / psd-peg.py set out to do): each new edge goes to a lowest-degree
 * check not reached from its variable node within `depth` check levels, so no cycle shorter than
 * 2 (depth + 1) is closed while avoidable (depth 2: no 4-cycles).  Deterministic in (profile, depth, seed).
 */
int qldpc_code_ira_peg(int N, int K, float hi_frac, int dv_hi, int dv_lo, int depth, uint64_t seed, qldpc_code **out);
/*
 * Quasi-cyclic code whose base graph is grown by PEG with a circulant shift per edge chosen so that the cycles closed in
 * the base graph stay open in the lifted graph (what the reference's ldpc_examples/psd-peg.py:281-447 does, regular
 * information-column degree dv, parity part = identity, output also as an AFF3CT .qc file when qc_path != NULL):
 * N = (n_cols + m_rows) Z, M = m_rows Z.  *base_girth (optional) = shortest cycle closed in the base graph (0 = none);
 * no cycle of length 4 exists in the lifted graph.  Deterministic in (n_cols, m_rows, dv, Z, seed).
 */
int qldpc_code_qc_peg(int n_cols, int m_rows, int dv, int Z, uint64_t seed, const char *qc_path, qldpc_code **out, int *base_girth);
void qldpc_code_free(qldpc_code *code);
int qldpc_code_n(const qldpc_code *code);
int qldpc_code_m(const qldpc_code *code);
int qldpc_code_e(const qldpc_code *code);
int qldpc_code_max_cn_degree(const qldpc_code *code);
int qldpc_code_max_vn_degree(const qldpc_code *code);
int qldpc_code_is_ira(const qldpc_code *code);     /* 1 if parity VNs K..N-1 form a dual diagonal */
/* CN-major edge list, E entries each (pass NULL to skip one). */
int qldpc_code_export_edges(const qldpc_code *code, int *var, int *chk);
/* Number of conflict-free layers of the horizontal-layered order, and that order (M entries). */
int qldpc_code_layer_count(const qldpc_code *code);
int qldpc_code_layer_order(const qldpc_code *code, int *check_order, int *layer_ptr /* layers+1 */);
/* H x over GF(2) for one host word of 0/1 ints; returns the syndrome weight (>= 0) or a status. */
int qldpc_code_syndrome_host(const qldpc_code *code, const int *x, int *s);

/* ------------------------------------------------------------------ decoder ------------------ */
typedef struct qldpc_decoder_cfg {
    int schedule;        /* qldpc_schedule                                                       */
    int rule;            /* qldpc_rule                                                           */
    float rule_param;    /* OMS offset / NMS factor                                              */
    int n_ite;           /* maximum BP iterations (AFF3CT n_ite)                                 */
    int enable_syndrome; /* stop a frame once its syndrome is zero (AFF3CT enable_syndrome)      */
    int syndrome_depth;  /* consecutive zero syndromes required (AFF3CT syndrome_depth), >= 1    */
    int max_frames;      /* capacity: frames decoded concurrently in one call (AFF3CT n_frames)  */
    int device;          /* HIP device ordinal                                                   */
    int frames_per_lane; /* 0 = auto; 1, 2 or 4 frames per wavefront lane (64/128/256-frame groups) */
    int engine;          /* qldpc_engine: 0 = auto (edge-parallel for <= 8 frames when supported) */
    int freeze_messages; /* FRAMES engine with enable_syndrome: 1 = a converged frame's messages are frozen bit-for-bit
                            (lane-masked stores; qldpc_fetch_post_dev is then exact for every frame, ~15 % slower);
                            0 = only its hard decisions / iteration count / success flag are frozen (default)      */
    int msg_dtype;       /* 0 = fp32 messages (the AFF3CT float build, bit-exact class); 1 = messages rounded to
                            binary16 in HBM, fp32 arithmetic (half the bytes per iteration; FER-tolerance class against
                            AFF3CT, bit-exact against the oracle run with the same rounding; FRAMES engine, flooding);
                            2 = 8-bit fixed point: channel LLRs quantised to clamp(rint(LLR * quant_scale), +-127), messages
                            saturating at +-127, integer min-sum (MS / OMS / NMS; flooding, or horizontal layered with the posterior
                            kept in 8 bits as well; FRAMES engine, 4 frames per lane):
                            a quarter of the bytes per iteration, FER-tolerance class against AFF3CT, bit-exact against the
                            oracle's integer decoder; qldpc_fetch_post_dev then returns the integer posteriors           */
    float quant_scale;   /* msg_dtype 2: quantiser steps per LLR unit (0 = 8.0); OMS offset = rint(rule_param * quant_scale)
                            steps, NMS factor = rint(rule_param * 128) / 128                                          */
    int compact;         /* FRAMES engine, flooding, enable_syndrome, freeze_messages = 0: active-frame compaction.
                            Once the frames that have not converged fit into <= 0.6 of the groups in flight they are dealt into fewer,
                            full groups (the message arrays are not copied: the next check pass reads them through a slot map), so a
                            group no longer runs until its slowest frame.  Decisions, iteration counts and success flags are unchanged;
                            qldpc_fetch_post_dev is refused after a run that compacted.  0 = auto (batches of >= 4 groups), 1 = whenever
                            a group can be saved, 2 = never                                                          */
    int layer_chain;     /* horizontal layered, fp32 messages, 64-frame groups, freeze_messages = 0, check degree <= 40: run a sweep as ONE launch in
                            which a check waits for the earlier checks on its own variable nodes (per-VN version counters, agent-coherent posterior
                            rows) instead of one launch per layer of mutually VN-disjoint checks.  Same results bit for bit.  0 = auto (fixed-iteration
                            runs with 2 to 8 frame groups and a layer launch of 8 192 .. 65 535 waves: measured +10 .. 13 % on the N = 10^6 code with 128 - 256
                            frames; no gain with one group or with the per-sweep early exit, a loss with many groups or several decoders side by side),
                            1 = on, 2 = off.  The one-launch sweep works on explicit messages.  Layered MS / OMS / NMS sweeps (fp32, 64-frame groups,
                            check degree <= 32, freeze_messages = 0) otherwise keep a COMPRESSED CHECK STATE -- the two magnitudes a check's messages
                            take and two dc-bit masks per frame instead of the dc messages (csrc/qldpc_kernels_cst.h): the same floats, 0.61 x the
                            bytes on the N = 10^6 code -- so for those rules auto never picks the one-launch sweep, and 1 gives up the state for it */
    int reserved[1];     /* must be zero                                                         */
} qldpc_decoder_cfg;

void qldpc_decoder_cfg_default(qldpc_decoder_cfg *cfg);
/* info_bits_pos may be NULL (= 0..K-1, the harness default VAR/main.cpp (alist-v1.0.1):147-159). */
int qldpc_decoder_create(const qldpc_code *code, int K, const int *info_bits_pos,
                         const qldpc_decoder_cfg *cfg, qldpc_decoder **out);
void qldpc_decoder_free(qldpc_decoder *dec);
/* hipStream_t to launch on (NULL = the null stream).  All *_dev calls are asynchronous on it. */
int qldpc_decoder_set_stream(qldpc_decoder *dec, void *hip_stream);
int qldpc_decoder_reset(qldpc_decoder *dec);
size_t qldpc_decoder_device_bytes(const qldpc_decoder *dec);   /* HBM held by this decoder        */
/* Allocate now the buffers the load calls would otherwise allocate on first use (per-frame erasure ballots). */
int qldpc_decoder_reserve(qldpc_decoder *dec);

/* AFF3CT mirror, host pointers: Y_N[n_frames][N] -> V_K[n_frames][K] (one int per bit). Synchronous. */
int qldpc_decode_siho(qldpc_decoder *dec, const float *Y_N, int *V_K, int n_frames);

/* ---- staged, HBM-resident path.  d_* are device pointers on the decoder's device. ----------- */
/* load: channel LLRs [n_frames][N] float (frame-major, as decode_siho takes them). */
int qldpc_load_llr_dev(qldpc_decoder *dec, const float *d_llr, int n_frames);
/*
 * load: QKD-native frame formation on the device.  d_bits[n_frames][ceil(N/32)] packed MSB-first
 * (helpers.h:65-70): Bob's sifted-key bits at channel VNs, Alice's disclosed bits at pinned VNs.
 * d_llr_mag[n_frames] = |LLR| of a channel bit of that frame, i.e. qldpc_bsc_llr(estimated QBER)
 * (ProcessBlock.localError) computed by the host; d_vn_class[N] (NULL = all QLDPC_VN_CHANNEL) is
 * shared by all frames.
 */
int qldpc_load_bits_dev(qldpc_decoder *dec, const uint32_t *d_bits, const float *d_llr_mag,
                        const uint8_t *d_vn_class, int n_frames);
/* The same with a per-frame count of channel VNs: class-0 VNs at index >= d_n_channel[f] are known (shortened) bits of
 * frame f and are pinned like class 1.  Lets blocks of different length share one code and one launch. */
int qldpc_load_bits_short_dev(qldpc_decoder *dec, const uint32_t *d_bits, const float *d_llr_mag,
                              const uint8_t *d_vn_class, const int *d_n_channel, int n_frames);
/*
 * Syndrome form: instead of pinning disclosed parity VNs, every check c must come out with the
 * parity s_c that Alice computed on her key (s = H x_A).  d_synd_bits[n_frames][ceil(M/32)], MSB-first.  Call after
 * qldpc_load_* (which clears it) and before qldpc_run; the success flag / early exit then test H x = s.  Works with
 * any H (no encoder needed); not an AFF3CT configuration, so it is checked against the oracle's own coset mode.
 */
int qldpc_load_syndrome_dev(qldpc_decoder *dec, const uint32_t *d_synd_bits, int n_frames);
/*
 * Per-frame puncturing (BS/src/main.cpp:359-362, the harness's `LLRs[pattern[i]] = 0` with a pattern of its own for every frame):
 * d_erase_bits[n_frames][ceil(N/32)] packed MSB-first, a set bit makes that VN of that frame an erasure (channel LLR 0) whatever
 * its class in d_vn_class.  Lets blocks punctured to different efficiencies share one code and one launch.  Call after
 * qldpc_load_bits_* / qldpc_load_llr_dev for the frames just loaded; the next load clears it.
 */
int qldpc_load_erasures_dev(qldpc_decoder *dec, const uint32_t *d_erase_bits, int n_frames);
/* s = H x for packed words d_bits[n_frames][ceil(N/32)] -> d_synd_bits[n_frames][ceil(M/32)] (Alice's side). */
int qldpc_syndrome_dev(qldpc_decoder *dec, const uint32_t *d_bits, uint32_t *d_synd_bits, int n_frames);
/* run the BP iterations on what was loaded. */
int qldpc_run(qldpc_decoder *dec);
/* fetch: hard decision of every VN, packed MSB-first, d_out[n_frames][ceil(N/32)]. */
int qldpc_fetch_packed_dev(qldpc_decoder *dec, uint32_t *d_out);
/* fetch: V_K[n_frames][K] ints at info_bits_pos (AFF3CT layout). */
int qldpc_fetch_info_dev(qldpc_decoder *dec, int *d_V_K);
/* fetch: per-frame iterations executed and 1/0 "syndrome of the hard decision is zero". */
int qldpc_fetch_status_dev(qldpc_decoder *dec, int *d_iters, int *d_ok);
/* fetch: a-posteriori LLRs [n_frames][N] (debug / parity tests; costs one extra pass). */
int qldpc_fetch_post_dev(qldpc_decoder *dec, float *d_post);
/* block until everything queued on the decoder's stream is done. */
int qldpc_sync(qldpc_decoder *dec);

/* ---- measurement hooks ---------------------------------------------------------------------- */
typedef struct qldpc_kernel_stat {
    char name[32];        /* kernel family: "cn_update", "vn_update", ...                         */
    uint64_t launches;
    double total_ms;      /* hipEvent time summed over launches                                   */
    double alg_bytes;     /* algorithmic bytes summed over launches         */
    double moved_bytes;   /* bytes those launches have to move in the data form in use (e.g. with coded LLRs a
                             variable-node pass fetches N/8 bytes of received-bit ballots per frame, not the 4 N bytes of an
                             LLR array that alg_bytes prices): time these for the bandwidth actually sustained */
} qldpc_kernel_stat;
/* When on, every kernel launch is bracketed by hipEvents on the decoder's stream. */
int qldpc_profile_enable(qldpc_decoder *dec, int on);
/* Sync, fold events into stats; writes up to cap entries, returns the count (or a status). */
int qldpc_profile_read(qldpc_decoder *dec, qldpc_kernel_stat *out, int cap);
int qldpc_profile_clear(qldpc_decoder *dec);
/* Launches actually issued by the last qldpc_run (converged groups make later ones no-ops). */
int qldpc_last_run_iterations(const qldpc_decoder *dec);
/* Early-exit bookkeeping of the last qldpc_run (FRAMES engine; zeros otherwise): out[0] = lane-iterations executed (every
 * iteration a group ran counts its whole width, converged lanes included -- divide the sum of the frames' iteration counts by it
 * for the useful-work fraction), out[1] = compactions, out[2] = groups in flight at the end, out[3] = frames per group.
 * Synchronises the decoder's stream. */
int qldpc_last_run_stats(qldpc_decoder *dec, long long out[4]);

/* ------------------------------------------------------------------ encoder (Alice) ---------- */
/* method: "IRA" (dual-diagonal accumulate), "IDENTITY" / "LU_DEC" (Encoder_LDPC_from_H's two G_methods, VAR/main.cpp (alist-v1.0.1):135-145:
 * GF(2) elimination of any H, parity positions searched from the first / from the last column) or "QC" (Encoder_LDPC_from_QC,
 * VAR/main.cpp (qc):145: info bits first, parity = H2^-1 H1 u; QLDPC_EUNSUPPORTED when H2 is singular). */
int qldpc_encoder_create(const qldpc_code *code, const char *method, int device, qldpc_encoder **out);
void qldpc_encoder_free(qldpc_encoder *enc);
/* Size the encoder's device workspace for calls of up to max_frames frames now (it otherwise grows on first need): a caller that must not
 * allocate later -- the daemon after ldpc_init -- says so here.  One encoder is not re-entrant (neither is an AFF3CT module). */
int qldpc_encoder_reserve(qldpc_encoder *enc, int max_frames);
int qldpc_encoder_k(const qldpc_encoder *enc);
int qldpc_encoder_info_bits_pos(const qldpc_encoder *enc, int *pos /* K */);
/* host mirror of encoder->encode: U_K[n_frames][K] ints -> X_N[n_frames][N] ints. */
int qldpc_encode(qldpc_encoder *enc, const int *U_K, int *X_N, int n_frames);
/* device, packed MSB-first: d_info[n_frames][ceil(K/32)] -> d_cw[n_frames][ceil(N/32)]. */
int qldpc_encode_packed_dev(qldpc_encoder *enc, const uint32_t *d_info, uint32_t *d_cw, int n_frames,
                            void *hip_stream);

/* ------------------------------------------------------------------ reconciliation sessions -- */
/*
 * What an ecd2 LDPC handler does between QBER estimation and privacy amplification, i.e. the
 * replacement of the cascade_biconf exchange (subcomponents/cascade_biconf.c:427-940, ~55 packets
 * each way) by ONE parity message: the two `return 81` arms at subcomponents/qber_estim.c:337-340
 * and :420-423 call into this.  Buffers are the daemon's own: ProcessBlock.mainBufPtr words,
 * MSB-first (helpers.h:65-70), `workbits` valid bits (helpers.c:31-69), QBER = localError.
 *
 * Per block (qldpc_recon_plan): the QBER estimate is clamped to [0.001, 0.25]; target rate R* = min(min_cr(qber, efficiency),
 * capacity - rate_gap (65536/K)^0.4) (BS/src/main.cpp:29); table rate = largest entry <= R* (:235-266); code = IRA MOTHER code with K a multiple of
 * `mother_step` (blocks above `mother_max` bits: K = workbits rounded up to `key_quantum`) and M = round(K (1-R)/R) parity VNs, the
 * information VNs past the block's length are shortened (known 0, pinned per frame); of the M parity bits only
 * d = ceil(workbits (1/R* - 1)) are disclosed and the other M - d are punctured (BS/src/main.cpp:34,305-311,359-362:
 * parity_bits_to_punct, LLR 0), evenly spaced along the accumulator.  Alice sends the d parity bits + CRC-32 of her key; Bob pins
 * them (+-23.03), erases the punctured ones, decodes and verifies.  Leak = d + 32 bits.
 */
typedef struct qldpc_recon qldpc_recon;

typedef struct qldpc_recon_cfg {
    int device;
    float efficiency;      /* f in min_cr(q, f); 1.4                        */
    int n_rates;           /* <= 8                                                               */
    float rates[8];        /* ascending; default {0.5, 0.7, 0.8, 0.9}                            */
    int n_ite;             /* 50                                                                 */
    int rule;              /* QLDPC_RULE_NMS                                                     */
    float rule_param;      /* 0.75                                                               */
    int key_quantum;       /* 1024 (multiple of 32)                                              */
    int max_blocks;        /* blocks decoded concurrently by qldpc_recon_decode_batch            */
    uint64_t seed;         /* IRA construction seed shared by both sides (7)                     */
    int schedule;          /* schedule of Bob's decoder: QLDPC_RECON_SCHED_AUTO (default) = horizontal layered when the decoders take
                              batches (max_blocks > 8: half the iterations for the same bytes per sweep; config-3 stream 16.1 -> 14.4 ms,
                              0 instead of 0 - 1 first-round failures in 2 048 epochs), flooding for max_blocks <= 8 (the one-block
                              edge-parallel engine is flooding); QLDPC_SCHED_FLOODING / QLDPC_SCHED_HLAYERED force one */
    int mother_step;       /* 8192: blocks of up to mother_max bits use mother codes whose K is a multiple of this (a block
                              is shortened to its length per frame), so a handful of codes serve every block; 0 = a code per size */
    int mother_max;        /* 65536 */
    float rate_gap;        /* the effective rate stays rate_gap (65536 / K)^0.4 below the BSC capacity 1 - h(q); 0 = by rule
                              (0.035 SPA / LSPA, 0.05 min-sum family), times a factor per mother rate and size: gap_profile below */
    int puncture;          /* 1 (default): puncture parity VNs down to the target efficiency; 0 / 2: disclose all M    */
    int preload;           /* 1: build every (mother size, rate) entry in qldpc_recon_create -- no code construction and no
                              device allocation afterwards for blocks of up to mother_max bits                          */
    int peg_depth;         /* 0: the information part of every code is the seeded socket shuffle (qldpc_code_ira); 1..4: grown by
                              progressive edge growth to that depth (qldpc_code_ira_peg; 2 = no 4-cycles), SURVEY.md 8f #3.  Both
                              sides must use the same value: the codes are derived from (size, rate, peg_depth, seed)          */
    int gap_profile;       /* how far below capacity the plan stays, as a multiple c(R, K) of rate_gap (65536 / K)^0.4 per mother rate R:
                              0 = as calibrated for the construction in use (PEG: 0.10 for R <= 0.75 on mothers of K >= 32 768, more on
                              shorter ones, 0.85 for R <= 0.85, 1 above -- leak 0.29 of the key on the config-3 stream; seeded shuffle:
                              0.6 / 0.9 / 1.0); 1 = round 2's 0.6 / 0.9 / 1.0 whatever the construction (fewer iterations, leak 0.305).
                              Both sides must use the same value                                                            */
} qldpc_recon_cfg;

/* Travels in the parity packet (all fields uint32, little-endian like every ecd2 header). */
typedef struct qldpc_recon_msg {
    uint32_t rate_index;   /* index into the rate table                                          */
    uint32_t key_bits;     /* workbits                                                           */
    uint32_t code_k;       /* info VNs of the (mother) code, >= key_bits                         */
    uint32_t code_m;       /* parity VNs of the code                                             */
    uint32_t crc32;        /* CRC-32 (IEEE) of Alice's key words, tail bits masked               */
    uint32_t n_punct;      /* parity VNs punctured: code_m - n_punct bits are disclosed          */
} qldpc_recon_msg;

void qldpc_recon_cfg_default(qldpc_recon_cfg *cfg);
int qldpc_recon_create(const qldpc_recon_cfg *cfg, qldpc_recon **out);
void qldpc_recon_free(qldpc_recon *r);
/* Rate choice, code dimensions and puncturing for a block; fills everything but crc32.  qber in [0, 0.5) (0 is clamped). */
int qldpc_recon_plan(const qldpc_recon *r, int key_bits, float qber, qldpc_recon_msg *msg);
int qldpc_recon_parity_words(const qldpc_recon_msg *msg);    /* words of disclosed parity a message carries: ceil((code_m - n_punct)/32) */
int qldpc_recon_leaked_bits(const qldpc_recon_msg *msg);     /* code_m - n_punct + 32 (CRC)                                             */
int qldpc_recon_check_header(const qldpc_recon *r, const qldpc_recon_msg *msg, int key_bits);      /* QLDPC_OK / QLDPC_ESIZE: as the decode calls check it */
long qldpc_recon_entries_created(const qldpc_recon *r);      /* (code, encoder, decoder) sets built so far: constant after a preload   */
/* qldpc_profile_enable / _read of Bob's decoders, summed over the session's codes by kernel kind */
int qldpc_recon_profile_enable(qldpc_recon *r, int on);
int qldpc_recon_profile_read(qldpc_recon *r, qldpc_kernel_stat *out, int cap);
/* Alice: the disclosed parity bits (qldpc_recon_parity_words(msg_out) words, MSB-first, in position order) + message header. */
int qldpc_recon_encode(qldpc_recon *r, const uint32_t *key_words, int key_bits, float qber,
                       qldpc_recon_msg *msg_out, uint32_t *parity_words, int parity_cap_words);
/* Bob: corrects key_words in place.  QLDPC_OK = decoded and CRC verified; QLDPC_EDECODE = failed,
 * key untouched.  corrected_bits / leaked_bits / iterations may be NULL. */
int qldpc_recon_decode(qldpc_recon *r, uint32_t *key_words, int key_bits, float qber, const qldpc_recon_msg *msg,
                       const uint32_t *parity_words, int *corrected_bits, int *leaked_bits, int *iterations);
/* Alice, second round (incremental redundancy): the parity bits of a plan already made -- `msg` as qldpc_recon_encode left it, with
 * n_punct lowered by the caller (0 = every parity bit of the mother code).  Same codeword, lower effective rate; Bob decodes again with
 * qldpc_recon_decode* and the new header.  The ecd2 handlers use it when a verdict asks for the withheld bits (ldpc_reconcile.c). */
int qldpc_recon_encode_planned(qldpc_recon *r, const uint32_t *key_words, int key_bits, qldpc_recon_msg *msg, uint32_t *parity_words, int cap);
/* Bob, n blocks that share one plan (same key_bits / rate / code dims) in one launch.
 * key_words[n][ceil(key_bits/32)], parity_words[n][ceil(code_m/32)] (row i holds qldpc_recon_parity_words(&msgs[i]) words),
 * status[n] = QLDPC_OK / QLDPC_EDECODE / QLDPC_ESIZE (that block's header does not match it). */
/* Alice's side for many blocks of any mix of lengths / plans in one call: every msgs[i] is planned and filled as by
 * qldpc_recon_encode, blocks are grouped by plan and encoded in launches of up to max_blocks frames. */
int qldpc_recon_encode_blocks(qldpc_recon *r, int n, const uint32_t *const *key_words, const int *key_bits, const float *qber,
                              qldpc_recon_msg *msgs, uint32_t *const *parity_words, const int *parity_cap);
int qldpc_recon_decode_batch(qldpc_recon *r, int n_blocks, uint32_t *key_words, int key_bits, const float *qber,
                             const qldpc_recon_msg *msgs, const uint32_t *parity_words, int *status,
                             int *corrected_bits, int *iterations);
/* Blocks of any mix of lengths and plans, one pointer per block (keys are decoded in place): grouped by plan and decoded in
 * launches of up to max_blocks frames; within a plan the blocks may differ in length.  status[i] = QLDPC_OK | QLDPC_EDECODE. */
int qldpc_recon_decode_blocks(qldpc_recon *r, int n, uint32_t *const *key_words, const int *key_bits, const float *qber,
                              const qldpc_recon_msg *msgs, const uint32_t *const *parity_words, int *status, int *corrected,
                              int *iterations);
uint32_t qldpc_crc32_words(const uint32_t *words, int n_bits);
/* The same CRC-32 the way the device verification computes it (rk_verify / rk_crc in qldpc_recon.hip): `lanes` (a power of two) equal
 * chunks, each run through the byte-wise recurrence from a zero register, folded pairwise with x^len multipliers mod the CRC polynomial,
 * start value and final inversion applied at the end.  Host mirror for tests: equals qldpc_crc32_words for every length. */
uint32_t qldpc_crc32_words_chunked(const uint32_t *words, int n_bits, int lanes);

/* ------------------------------------------------------------------ privacy amplification ---- */
/*
 * The hash loop of privAmp_doPrivAmp (subcomponents/priv_amp.c:213-218) with the LFSR word stream of
 * rnd_getPrngValue2_32 (subcomponents/rnd.c:118-127, feedback 0xe0000200): final key bit i =
 * parity(XOR_j key[j] & w[i*numwords + j]).  key = mainBufPtr words (bits past workbits are ignored),
 * seed = EcPktHdr_StartPrivAmp.seed (definitions/packets.h:170-175), out = ceil(final_bits/32) words.
 * Pure integer arithmetic: bit-identical to the reference.
 */
int qldpc_privamp(int device, const uint32_t *key_words, int workbits, uint32_t seed, int final_bits, uint32_t *final_words);
/* device pointers; the key's tail bits past workbits must already be zero */
int qldpc_privamp_dev(const uint32_t *d_key_words, int workbits, uint32_t seed, int final_bits, uint32_t *d_final_words, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* QLDPC_H */
