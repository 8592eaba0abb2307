/*
 * qldpc_engine_int.h -- internals shared by the translation units of the decoder (not part of the C ABI): the decoder object,
 * the per-generation state of the early-exit run, and the kernel-launch dispatchers.  The dispatchers instantiate every
 * (frames per lane, degree cap, rule family, message type) variant of the hot kernels; they are compiled once per frames-per-lane
 * value (qldpc_launch.hip with -DQL_V=1|2|4) so that the build runs in parallel.
 */
#ifndef QLDPC_ENGINE_INT_H
#define QLDPC_ENGINE_INT_H

#include <hip/hip_runtime.h>

#include <algorithm>
#include <vector>

#include "../../include/qldpc.h"
#include "qldpc_graph.h"
#include "qldpc_kernels.h"
#include "qldpc_kernels_i8.h"

#define HIPCHK(expr)                                                                                    \
    do {                                                                                                \
        hipError_t e__ = (expr);                                                                        \
        if (e__ != hipSuccess) {                                                                        \
            qldpc_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e__));     \
            return QLDPC_EHIP;                                                                          \
        }                                                                                               \
    } while (0)

enum { KS_CN = 0, KS_VN, KS_LAYER, KS_SYND, KS_STATUS, KS_LOAD, KS_FETCH, KS_COUNT };
static const char *const ks_names[KS_COUNT] = {"cn_update", "vn_update", "layer_update", "syndrome", "status", "load", "fetch"};

struct prof_rec { int kind; double bytes, moved; hipEvent_t a, b; };

struct bucket { int cap; int n; int *d_list; int *d_rec; };   /* cap = register-resident degree bound, 0 = any degree */
/* d_rec (layer buckets with cap > 0, else NULL): entry i of the list as ONE aligned record {check, first edge, degree, 0, vn[0 .. cap)} --
 * a wave of a layer kernel learns everything it needs to ask for its rows in one scalar round trip instead of three dependent ones
 * (list -> cn_ptr -> cn_var); the launches of a layered sweep are small and bound by the latency of that chain (DESIGN section 8 #6h);
 * QK_REC_HDR (qldpc_kernels.h) = 4 header words */

/*
 * Per-frame state of one generation of the early-exit run (qldpc_kernels_compact.h).  Generation 0 is the batch as loaded (its
 * pointers alias the decoder's base arrays); every compaction opens the next one with fewer groups.  A retired generation
 * keeps the results of the frames that converged in it.
 */
struct gen_state {
    int G, cap;                          /* groups in use / allocated */
    u64 *sgn, *hard, *unsat, *done, *ybits, *synd, *ebits;
    int *depth, *iters, *origin, *src;   /* origin[slot] = frame index in the caller's batch (-1: padding); src[slot] = slot in the previous generation */
    float *fmag; int *fnch;
    float *llr; uint32_t *llr8;          /* channel LLR rows in this generation's layout (NULL with coded LLRs) */
};
#define QLDPC_MAX_GENS 6

struct qldpc_decoder {
    qldpc_decoder_cfg cfg;
    int N, M, E, K;
    int V, FG, G;
    int device;
    hipStream_t stream;
    /* graph on device */
    int *d_cn_ptr, *d_cn_tr, *d_cn_var, *d_vn_ptr, *d_info_pos;
    int *d_vn_tr;                    /* VN-major slot -> CN-major edge (inverse of cn_tr, padded): where chk_to_var of a slot lives */
    int *d_cn_var_t, max_dc;         /* cn_var transposed to [edge position][check], -1 padded (syndrome pass) */
    std::vector<bucket> cn_buckets, vn_buckets;
    std::vector<std::vector<bucket>> layer_buckets;   /* per layer */
    int n_layers;
    /* one launch per sweep for small batches (qldpc_kernels_chain.h): execution order, per-edge {dv, rank}, per-VN version counters, ticket / fault words */
    int layer_cst;      /* layered min-sum keeps {cst1, cst2} per check and two ballot words per edge instead of dc messages (qldpc_kernels_cst.h) */
    int chain, chain_blocks, chain_lds; int *d_chain_order, *d_chain_dep, *d_chain_ver, *d_chain_ctl; int chain_sweeps;
    int layer_first;                 /* layered fp32 run, sweep 0, messages not frozen: the layer kernels treat the messages as zero instead of reading a cleared array */
    /* state */
    float *d_llr, *d_a, *d_b;        /* flooding: a = v2c, b = c2v ; layered: a = post, b = msg */
    float *d_post;                   /* lazily allocated by fetch_post */
    u64 *d_sgn, *d_hard, *d_unsat, *d_done;
    int *d_depth, *d_iters, *d_active;   /* d_active[0] = groups, [1] = frames still unconverged after the last status pass */
    int *h_active, *h_active_dev;    /* mapped pinned memory {groups, frames, sequence number} and its device address: qk_status reports, the host spins */
    int poll_seq;
    unsigned long long *d_work;      /* group-iterations executed in the current run (qk_status) */
    /* active-frame compaction: generations of the per-frame state; the d_* pointers above and G are the CURRENT generation's */
    int G0;
    std::vector<gen_state> gens;     /* [0] aliases the base arrays */
    int cur_gen, compact_mode, compactions;
    int live_lanes;                  /* lanes of the groups still active at the last poll (byte accounting of the profile; 0 = not polled yet) */
    float compact_ratio;             /* compact when the active frames fit into <= ratio * G groups */
    const int *remap_src;            /* != NULL: the next check pass reads var_to_chk through this map (set by a compaction) */
    int *d_gcount, *d_goff;          /* [G0], [G0 + 1] */
    int llr_alt_cap[2]; float *d_llr_alt[2]; uint32_t *d_llr8_alt[2];   /* side buffers for compacted LLR rows (generations ping-pong between them; capacity in groups) */
    size_t bytes;
    int n_frames;                    /* loaded */
    int loaded, ran;
    int last_iters;
    int poll_every;
    u64 *d_synd;                     /* [G][M][V] target-syndrome ballots (syndrome form), NULL until used */
    uint32_t *e_synd;                /* edge engine: packed target syndromes [F][Wm] */
    int has_synd;
    float *h_in; int *h_out;         /* device staging of qldpc_decode_siho's host vectors */
    int msg_half;                    /* 1: v2c / c2v stored as binary16 (flooding, frames engine) */
    /* coded channel LLRs (flooding, fp32 / binary16 messages, after qldpc_load_bits_*): no LLR array is read, see qk_coded_llr */
    u64 *d_ybits; float *d_fmag; int *d_fnch; uint8_t *d_vcls; int llr_coded;
    u64 *d_ebits; int has_erase;     /* [G][N][V] per-frame erasure ballots of the coded form (allocated on first use), set by qldpc_load_erasures_dev */
    int post_closes_run;             /* set around the _compute_post that ends an early-exit run (not for posterior read-back) */
    int packed_h16;                  /* binary16 variant: use the packed check-node kernel when V == 2 (QLDPC_PACKED_H16=0 turns it off) */
    int msg_i8;                      /* 1: 8-bit fixed-point messages and integer arithmetic (flooding min-sum family, frames engine, V = 4) */
    uint32_t *d_llr8;                /* [G][N][256] quantised channel LLRs, four frames of a lane per dword */
    float quant_scale;
    int freeze;                      /* 1: lane-masked stores keep converged frames' messages bit-frozen (exact posteriors, slower) */
    /* edge-parallel engine (one block at a time): llr [F][N], d_a = v2c / d_b = c2v [F][E] */
    int engine, eW, eS, e_stride;
    size_t e_lds;
    uint32_t *e_sgn, *e_hard;
    float *e_c2v1;                   /* odd-iteration chk_to_var buffer (d_b is the even one) */
    int *e_unsat, *e_done_at;
    int *h_done;
    hipEvent_t e_ev[2];
    int use_graphs, graph_frames;
    int persist, persist_blocks;     /* one cooperative launch per decode (qe_persist): enabled / co-resident workgroups (0 = not probed yet) */
    int *e_ctl; int iters_pending;   /* control words of the one-launch decode (qe_xcd); its iteration count / fault flag are read back on demand */
    hipStream_t cap_stream;
    std::vector<hipGraphExec_t> e_graphs;   /* one per chunk of poll_every iterations */
    /* profiling */
    int prof_on;
    std::vector<prof_rec> prof_pending;
    qldpc_kernel_stat stats[KS_COUNT];
};

#define LAUNCHCHK()                                                                                     \
    do {                                                                                                \
        hipError_t e__ = hipGetLastError();                                                             \
        if (e__ != hipSuccess) { qldpc_set_error("%s:%d: kernel launch -> %s", __FILE__, __LINE__, hipGetErrorString(e__)); return QLDPC_EHIP; } \
    } while (0)

static inline int grid_x(int n_items, int per_wave)
{
    const int per_block = QK_WAVES * per_wave;
    return std::max(1, (n_items + per_block - 1) / per_block);
}

static inline int family_of(int rule)
{
    switch (rule) {
    case QLDPC_RULE_MS: case QLDPC_RULE_OMS: case QLDPC_RULE_NMS: return QK_FAM_MS;
    case QLDPC_RULE_SPA: return QK_FAM_SPA;
    case QLDPC_RULE_LSPA: return QK_FAM_LSPA;
    default: return QK_FAM_AMS;
    }
}

/* the rule in quantiser units (qldpc.h: quant_scale) */
static inline qi_rule qi_rule_of(const qldpc_decoder *d)
{
    qi_rule qr{d->cfg.rule, 0};
    if (d->cfg.rule == QLDPC_RULE_OMS) qr.param = (int)lrintf(d->cfg.rule_param * d->quant_scale);
    if (d->cfg.rule == QLDPC_RULE_NMS) qr.param = (int)lrintf(d->cfg.rule_param * 128.0f);
    qr.param = std::min(128, std::max(0, qr.param));
    return qr;
}

/* the in-between variable-node passes only need to leave ballots when the syndrome test reads them; _compute_post always does */
static inline int want_ballots(const qldpc_decoder *d, int mode)
{
    if (mode == QK_VN_POST) return 1 | (d->post_closes_run ? 2 : 0);      /* bit 1: skip groups that converged as a whole (their ballots are final) */
    return d->cfg.enable_syndrome ? 1 : 0;
}

/* kernel-launch dispatchers (qldpc_launch.hip); `first`: the check pass of iteration 0 in coded-LLR mode */
template <int V> void qldpc_launch_cn(qldpc_decoder *d, const bucket &b, bool first);
template <int V> void qldpc_launch_layer(qldpc_decoder *d, const bucket &b);
void qldpc_launch_layer_chain(qldpc_decoder *d, int sweep);      /* V = 1 only */
int qldpc_chain_resident_blocks(qldpc_decoder *d);
template <int V, int MODE> void qldpc_launch_vn(qldpc_decoder *d, const bucket &b, float *post_out);
#define QLDPC_DECLARE_LAUNCH(V)                                                                   \
    extern template void qldpc_launch_cn<V>(qldpc_decoder *, const bucket &, bool);                \
    extern template void qldpc_launch_layer<V>(qldpc_decoder *, const bucket &);                   \
    extern template void qldpc_launch_vn<V, QK_VN_FIRST>(qldpc_decoder *, const bucket &, float *); \
    extern template void qldpc_launch_vn<V, QK_VN_NORMAL>(qldpc_decoder *, const bucket &, float *); \
    extern template void qldpc_launch_vn<V, QK_VN_POST>(qldpc_decoder *, const bucket &, float *);

#endif /* QLDPC_ENGINE_INT_H */
