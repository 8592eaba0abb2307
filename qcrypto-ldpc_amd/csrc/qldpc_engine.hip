/*
 * qldpc_engine.hip -- decoder object, launch sequencing and the C ABI of the batched BP decoder.
 *
 * Mirrors module::Decoder_LDPC_BP_flooding / _horizontal_layered as the reference harness drives
 * them (BS/src/main.cpp:193,365,389; VAR/main.cpp (alist-v1.0.1):179-256,438): create(K, N, n_ite,
 * H, info_bits_pos, rule, enable_syndrome, syndrome_depth, n_frames), decode_siho(Y_N, V_K), reset().
 * Everything heavy is a HIP kernel from qldpc_kernels.h; there is no CPU fallback.
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "qldpc_engine_int.h"
#include "qldpc_kernels_edge.h"
#include "qldpc_kernels_chain.h"
#include "qldpc_kernels_compact.h"

extern "C" int qldpc_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

/* ------------------------------------------------------------------ bandwidth probe ---------- */

/* What this device sustains for the access shape the decoder is made of: 256-byte rows, one dword per lane, non-temporal, read + written
 * once.  bench.py prints it next to the kernels' rates (the 8 TB/s of the data sheet is not reachable by any kernel: DESIGN section 3.1). */
typedef float qk_probe_f4 __attribute__((ext_vector_type(4)));
template <int ROWS, bool WIDE>
static __global__ __launch_bounds__(256) void qk_copy_probe(const float *__restrict__ src, float *__restrict__ dst, size_t n_rows)
{
    /* one wavefront per ROWS consecutive 256-byte rows (WIDE: ROWS / 4 KiB-rows), every load issued before the first store: the shape of a
     * variable-node pass without its arithmetic */
    const int lane = threadIdx.x & 63;
    const size_t r0 = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * ROWS;
    if (r0 >= n_rows) return;
    if constexpr (WIDE) {
        qk_probe_f4 v[ROWS / 4];
#pragma unroll
        for (int k = 0; k < ROWS / 4; k++) v[k] = __builtin_nontemporal_load(reinterpret_cast<const qk_probe_f4 *>(src + (r0 + 4 * k) * 64) + lane);
#pragma unroll
        for (int k = 0; k < ROWS / 4; k++) __builtin_nontemporal_store(v[k], reinterpret_cast<qk_probe_f4 *>(dst + (r0 + 4 * k) * 64) + lane);
    } else {
        float v[ROWS];
#pragma unroll
        for (int k = 0; k < ROWS; k++) v[k] = __builtin_nontemporal_load(src + (r0 + k) * 64 + lane);
#pragma unroll
        for (int k = 0; k < ROWS; k++) __builtin_nontemporal_store(v[k], dst + (r0 + k) * 64 + lane);
    }
}

extern "C" int qldpc_copy_probe(int device, size_t bytes, int reps, int wide, double *gbytes_per_s)
{
    int ndev = 0;
    if (!gbytes_per_s || reps < 1 || bytes < (1u << 20)) return QLDPC_EINVAL;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { qldpc_set_error("no HIP device visible: libqldpc has no CPU fallback"); return QLDPC_ENODEV; }
    if (device < 0 || device >= ndev) return QLDPC_ENODEV;
    HIPCHK(hipSetDevice(device));
    const size_t n_rows = bytes / 256 / 64 * 64;      /* whole workgroups of 4 x 16 rows */
    const unsigned blocks = (unsigned)(n_rows / 64);
    float *a = nullptr, *b = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = QLDPC_OK;
    float ms = 0.0f;
    if (hipMalloc((void **)&a, n_rows * 256) != hipSuccess || hipMalloc((void **)&b, n_rows * 256) != hipSuccess) { rc = QLDPC_ENOMEM; goto out; }
    if (hipMemset(a, 0, n_rows * 256) != hipSuccess || hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { rc = QLDPC_EHIP; goto out; }
    if (wide) hipLaunchKernelGGL((qk_copy_probe<16, true>), dim3(blocks), dim3(256), 0, 0, a, b, n_rows);      /* warm-up */
    else hipLaunchKernelGGL((qk_copy_probe<16, false>), dim3(blocks), dim3(256), 0, 0, a, b, n_rows);
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < reps; i++) {
        const float *from = (i & 1) ? b : a;
        float *to = (i & 1) ? a : b;
        if (wide) hipLaunchKernelGGL((qk_copy_probe<16, true>), dim3(blocks), dim3(256), 0, 0, from, to, n_rows);
        else hipLaunchKernelGGL((qk_copy_probe<16, false>), dim3(blocks), dim3(256), 0, 0, from, to, n_rows);
    }
    (void)hipEventRecord(e1, 0);
    if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess || hipGetLastError() != hipSuccess) { rc = QLDPC_EHIP; goto out; }
    *gbytes_per_s = 2.0 * (double)(n_rows * 256) * reps / ((double)ms * 1e-3) / 1e9;
out:
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(a); (void)hipFree(b);
    if (rc == QLDPC_EHIP) qldpc_set_error("copy probe: HIP error");
    return rc;
}

/* ------------------------------------------------------------------ decoder object ----------- */

QLDPC_DECLARE_LAUNCH(1)
QLDPC_DECLARE_LAUNCH(2)
QLDPC_DECLARE_LAUNCH(4)

static const int CN_CAPS[] = {8, 12, 20, 40};
static const int VN_CAPS[] = {4, 12};

template <typename T> static int dev_alloc(qldpc_decoder *d, T **p, size_t n)
{
    *p = nullptr;
    if (n == 0) n = 1;
    hipError_t e = hipMalloc((void **)p, n * sizeof(T));
    if (e != hipSuccess) { qldpc_set_error("hipMalloc(%zu bytes): %s", n * sizeof(T), hipGetErrorString(e)); return QLDPC_ENOMEM; }
    d->bytes += n * sizeof(T);
    return QLDPC_OK;
}

static int make_buckets(qldpc_decoder *d, const int *ptr, const int *ids, int n_ids, const int *caps, int n_caps, std::vector<bucket> &out, const int *var = nullptr)
{
    std::vector<std::vector<int>> lists(n_caps + 1);
    for (int i = 0; i < n_ids; i++) {
        const int id = ids ? ids[i] : i;
        const int deg = ptr[id + 1] - ptr[id];
        int b = n_caps;
        for (int k = 0; k < n_caps; k++) if (deg <= caps[k]) { b = k; break; }
        lists[b].push_back(id);
    }
    for (int k = 0; k <= n_caps; k++) {
        if (lists[k].empty()) continue;
        bucket bk;
        bk.cap = k < n_caps ? caps[k] : 0;
        bk.n = (int)lists[k].size();
        bk.d_rec = nullptr;
        int rc = dev_alloc(d, &bk.d_list, lists[k].size());
        if (rc != QLDPC_OK) return rc;
        out.push_back(bk);      /* owned by the decoder from here on: an error below leaves nothing behind that decoder_free does not release */
        HIPCHK(hipMemcpy(bk.d_list, lists[k].data(), lists[k].size() * sizeof(int), hipMemcpyHostToDevice));
        if (var && bk.cap > 0) {      /* the list once more as records {id, first edge, degree, 0, var[0 .. cap)} (padding: VN 0, never asked for) */
            const size_t stride = (size_t)(QK_REC_HDR + bk.cap);
            std::vector<int> rec(lists[k].size() * stride, 0);
            for (size_t i = 0; i < lists[k].size(); i++) {
                const int id = lists[k][i], b = ptr[id], deg = ptr[id + 1] - b;
                int *r = rec.data() + i * stride;
                r[0] = id; r[1] = b; r[2] = deg;
                for (int e = 0; e < deg; e++) r[QK_REC_HDR + e] = var[b + e];
            }
            rc = dev_alloc(d, &bk.d_rec, rec.size());
            if (rc != QLDPC_OK) return rc;
            out.back().d_rec = bk.d_rec;
            HIPCHK(hipMemcpy(bk.d_rec, rec.data(), rec.size() * sizeof(int), hipMemcpyHostToDevice));
        }
    }
    return QLDPC_OK;
}

extern "C" void qldpc_decoder_cfg_default(qldpc_decoder_cfg *cfg)
{
    if (!cfg) return;
    memset(cfg, 0, sizeof(*cfg));
    cfg->schedule = QLDPC_SCHED_FLOODING;
    cfg->rule = QLDPC_RULE_SPA;        /* the harness default, BS/src/main.cpp:193 */
    cfg->rule_param = 0.0f;
    cfg->n_ite = 100;                  /* BS/src/main.cpp:95-104 */
    cfg->enable_syndrome = 1;
    cfg->syndrome_depth = 1;
    cfg->max_frames = 1;
    cfg->device = 0;
    cfg->frames_per_lane = 0;
}

static void view_reset(qldpc_decoder *d);

extern "C" void qldpc_decoder_free(qldpc_decoder *d)
{
    if (!d) return;
    (void)hipSetDevice(d->device);
    view_reset(d);      /* the d_* pointers freed below are the base arrays */
    if (d->stream) (void)hipStreamSynchronize(d->stream); else (void)hipDeviceSynchronize();
    for (auto &r : d->prof_pending) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (auto &b : d->cn_buckets) (void)hipFree(b.d_list);
    for (auto &b : d->vn_buckets) (void)hipFree(b.d_list);
    for (auto &l : d->layer_buckets) for (auto &b : l) { (void)hipFree(b.d_list); (void)hipFree(b.d_rec); }
    (void)hipFree(d->d_cn_ptr); (void)hipFree(d->d_cn_tr); (void)hipFree(d->d_cn_var); (void)hipFree(d->d_vn_ptr); (void)hipFree(d->d_info_pos); (void)hipFree(d->d_cn_var_t); (void)hipFree(d->d_vn_tr);
    (void)hipFree(d->d_chain_order); (void)hipFree(d->d_chain_dep); (void)hipFree(d->d_chain_ver); (void)hipFree(d->d_chain_ctl);
    (void)hipFree(d->d_llr); (void)hipFree(d->d_llr8); (void)hipFree(d->d_ybits); (void)hipFree(d->d_ebits); (void)hipFree(d->d_fmag); (void)hipFree(d->d_fnch); (void)hipFree(d->d_vcls); (void)hipFree(d->d_a); (void)hipFree(d->d_b); (void)hipFree(d->d_post);
    (void)hipFree(d->d_sgn); if (d->d_hard != d->d_sgn) (void)hipFree(d->d_hard); (void)hipFree(d->d_unsat); (void)hipFree(d->d_done);
    (void)hipFree(d->d_depth); (void)hipFree(d->d_iters); (void)hipFree(d->d_active); (void)hipFree(d->h_in); (void)hipFree(d->h_out); (void)hipFree(d->d_synd); (void)hipFree(d->e_synd);
    (void)hipFree(d->e_c2v1); (void)hipFree(d->e_ctl); (void)hipFree(d->e_sgn); (void)hipFree(d->e_hard); (void)hipFree(d->e_unsat); (void)hipFree(d->e_done_at);
    if (d->h_done) (void)hipHostFree(d->h_done);
    for (auto &g : d->e_graphs) if (g) (void)hipGraphExecDestroy(g);
    if (d->cap_stream) (void)hipStreamDestroy(d->cap_stream);
    if (d->e_ev[0]) (void)hipEventDestroy(d->e_ev[0]);
    if (d->e_ev[1]) (void)hipEventDestroy(d->e_ev[1]);
    if (d->h_active) (void)hipHostFree(d->h_active);
    (void)hipFree(d->d_work); (void)hipFree(d->d_gcount); (void)hipFree(d->d_goff); (void)hipFree(d->d_llr_alt[0]); (void)hipFree(d->d_llr_alt[1]); (void)hipFree(d->d_llr8_alt[0]); (void)hipFree(d->d_llr8_alt[1]);
    for (size_t k = 1; k < d->gens.size(); k++) {
        gen_state &n = d->gens[k];
        (void)hipFree(n.sgn); if (n.hard != n.sgn) (void)hipFree(n.hard); (void)hipFree(n.unsat); (void)hipFree(n.done); (void)hipFree(n.ybits); (void)hipFree(n.synd); (void)hipFree(n.ebits);
        (void)hipFree(n.depth); (void)hipFree(n.iters); (void)hipFree(n.origin); (void)hipFree(n.src); (void)hipFree(n.fmag); (void)hipFree(n.fnch);
    }
    delete d;
}

static int create_impl(const qldpc_code *code, int K, const int *info_bits_pos, const qldpc_decoder_cfg *cfg, qldpc_decoder *d)
{
    d->cfg = *cfg;
    d->N = code->N; d->M = code->M; d->E = code->E; d->K = K;
    d->device = cfg->device;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { qldpc_set_error("no HIP device visible: libqldpc has no CPU fallback"); return QLDPC_ENODEV; }
    if (cfg->device < 0 || cfg->device >= ndev) { qldpc_set_error("device %d out of range (%d visible)", cfg->device, ndev); return QLDPC_ENODEV; }
    HIPCHK(hipSetDevice(cfg->device));
    int V = cfg->frames_per_lane;
    if (V == 0) V = (cfg->msg_dtype == 1) ? 2 : 1;      /* fp16 storage: 2 frames per lane keep the rows at 256 bytes; measured on MI355X: 256-byte rows (V = 1) are 2-5 % faster than V = 2 / 4 at every batch size, and exit earlier */
    if (const char *e = getenv("QLDPC_FRAMES_PER_LANE")) { int x = atoi(e); if (x == 1 || x == 2 || x == 4) V = x; }
    if (cfg->msg_dtype == 2) V = QI_V;                   /* 8-bit messages: the four frames of a lane are the four bytes of a dword */
    d->V = V; d->FG = 64 * V; d->G = (cfg->max_frames + d->FG - 1) / d->FG;
    d->freeze = cfg->freeze_messages ? 1 : 0;
    if (const char *e = getenv("QLDPC_FREEZE")) d->freeze = atoi(e) ? 1 : 0;
    /* early exit: the host looks at the active-group counter every 2 iterations once an iteration is long enough to hide the
     * ~30 us round trip (>= 2e7 message updates); small decodes run their iterations as early-returning launches instead */
    d->poll_every = (d->G >= 8 || (double)d->E * d->G * d->FG >= 2e7) ? 2 : 0;
    if (const char *e = getenv("QLDPC_POLL_EVERY")) d->poll_every = atoi(e);

    int rc;
    if ((rc = dev_alloc(d, &d->d_cn_ptr, (size_t)d->M + 1))) return rc;
    if ((rc = dev_alloc(d, &d->d_cn_tr, (size_t)d->E + QK_IDX_PAD))) return rc;
    if ((rc = dev_alloc(d, &d->d_cn_var, (size_t)d->E + QK_IDX_PAD))) return rc;
    HIPCHK(hipMemset(d->d_cn_tr, 0, sizeof(int) * ((size_t)d->E + QK_IDX_PAD)));
    HIPCHK(hipMemset(d->d_cn_var, 0, sizeof(int) * ((size_t)d->E + QK_IDX_PAD)));
    if ((rc = dev_alloc(d, &d->d_vn_ptr, (size_t)d->N + 1))) return rc;
    if ((rc = dev_alloc(d, &d->d_info_pos, (size_t)K))) return rc;
    HIPCHK(hipMemcpy(d->d_cn_ptr, code->cn_ptr, sizeof(int) * ((size_t)d->M + 1), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d->d_cn_tr, code->transpose, sizeof(int) * (size_t)d->E, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d->d_cn_var, code->cn_var, sizeof(int) * (size_t)d->E, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d->d_vn_ptr, code->vn_ptr, sizeof(int) * ((size_t)d->N + 1), hipMemcpyHostToDevice));
    {   /* chk_to_var of the frames engine is CN-major: the variable-node passes find the row of slot s at vn_tr[s] */
        std::vector<int> vt((size_t)d->E + QK_IDX_PAD, 0);
        for (int k = 0; k < d->E; k++) vt[(size_t)code->transpose[k]] = k;
        if ((rc = dev_alloc(d, &d->d_vn_tr, vt.size()))) return rc;
        HIPCHK(hipMemcpy(d->d_vn_tr, vt.data(), sizeof(int) * vt.size(), hipMemcpyHostToDevice));
    }
    {   /* check -> VN table transposed to [edge position][check] for the syndrome pass (qk_syndrome) */
        d->max_dc = code->max_dc;
        std::vector<int> t((size_t)std::max(1, code->max_dc) * d->M, -1);
        for (int c = 0; c < d->M; c++)
            for (int k = code->cn_ptr[c]; k < code->cn_ptr[c + 1]; k++) t[(size_t)(k - code->cn_ptr[c]) * d->M + c] = code->cn_var[k];
        if ((rc = dev_alloc(d, &d->d_cn_var_t, t.size()))) return rc;
        HIPCHK(hipMemcpy(d->d_cn_var_t, t.data(), sizeof(int) * t.size(), hipMemcpyHostToDevice));
    }
    {
        std::vector<int> pos((size_t)K);
        for (int i = 0; i < K; i++) {
            pos[i] = info_bits_pos ? info_bits_pos[i] : i;
            if (pos[i] < 0 || pos[i] >= d->N) { qldpc_set_error("info_bits_pos[%d] = %d out of range", i, pos[i]); return QLDPC_EINVAL; }
        }
        HIPCHK(hipMemcpy(d->d_info_pos, pos.data(), sizeof(int) * (size_t)K, hipMemcpyHostToDevice));
    }
    for (int k = 0; k < KS_COUNT; k++) { memset(&d->stats[k], 0, sizeof(d->stats[k])); snprintf(d->stats[k].name, sizeof(d->stats[k].name), "%s", ks_names[k]); }
    /* engine choice: lanes over frames (batch) or lanes over edges (one block at a time) */
    {
        const bool edge_ok = cfg->schedule == QLDPC_SCHED_FLOODING && cfg->rule <= QLDPC_RULE_SPA && code->max_dc <= 64 &&
                             (size_t)QE_THREADS * (size_t)code->max_dv * sizeof(float) <= 64 * 1024;
        int eng = cfg->engine;
        if (const char *e = getenv("QLDPC_ENGINE")) { int x = atoi(e); if (x >= 0 && x <= 2) eng = x; }
        if (eng == QLDPC_ENGINE_EDGES && !edge_ok) {
            qldpc_set_error("edge-parallel engine needs flooding, MS/OMS/NMS/SPA, check degree <= 64 (have %d) and VN degree <= 64 (have %d)", code->max_dc, code->max_dv);
            return QLDPC_EUNSUPPORTED;
        }
        if (eng == QLDPC_ENGINE_AUTO) eng = (edge_ok && cfg->max_frames <= 8) ? QLDPC_ENGINE_EDGES : QLDPC_ENGINE_FRAMES;
        d->engine = eng;
        d->msg_half = cfg->msg_dtype == 1;
        if (const char *e = getenv("QLDPC_MSG_HALF")) d->msg_half = atoi(e) ? 1 : 0;
        d->packed_h16 = 1;
        if (const char *e = getenv("QLDPC_PACKED_H16")) d->packed_h16 = atoi(e) ? 1 : 0;
        d->msg_i8 = cfg->msg_dtype == 2;
        d->quant_scale = cfg->quant_scale > 0.0f ? cfg->quant_scale : 8.0f;      /* measured on the config-2 code: FER at QBER 3.0 % 0.097 (fp32) / 0.119 (scale 8) / 0.168 (scale 4) */
        if (d->msg_i8) {
            if (cfg->rule > QLDPC_RULE_NMS || cfg->engine == QLDPC_ENGINE_EDGES || cfg->freeze_messages ||
                (cfg->frames_per_lane != 0 && cfg->frames_per_lane != QI_V)) {
                qldpc_set_error("8-bit messages are implemented for MS/OMS/NMS on the FRAMES engine (frames_per_lane 0 or 4, freeze_messages 0)");
                return QLDPC_EUNSUPPORTED;
            }
            if (code->max_dv > 256) { qldpc_set_error("8-bit messages: VN degree %d > 256 would overflow the 16-bit posterior", code->max_dv); return QLDPC_EUNSUPPORTED; }
            d->engine = eng = QLDPC_ENGINE_FRAMES;
            d->msg_half = 0;
        }
        if (d->msg_half && (eng != QLDPC_ENGINE_FRAMES || cfg->schedule != QLDPC_SCHED_FLOODING)) {
            if (cfg->engine == QLDPC_ENGINE_AUTO && cfg->schedule == QLDPC_SCHED_FLOODING) d->engine = QLDPC_ENGINE_FRAMES;
            else { qldpc_set_error("fp16 message storage is implemented for the flooding schedule on the FRAMES engine"); return QLDPC_EUNSUPPORTED; }
        }
    }
    if (d->engine == QLDPC_ENGINE_EDGES) {
        const size_t F = (size_t)cfg->max_frames;
        d->eW = (d->N + 31) / 32;
        d->eS = code->max_dc <= 8 ? 8 : (code->max_dc <= 16 ? 16 : (code->max_dc <= 32 ? 32 : 64));
        d->e_lds = (size_t)QE_THREADS * (size_t)code->max_dv * sizeof(float);
        d->e_stride = cfg->n_ite + 2;
        if ((rc = dev_alloc(d, &d->d_llr, F * d->N))) return rc;
        if ((rc = dev_alloc(d, &d->d_a, F * d->E))) return rc;
        if ((rc = dev_alloc(d, &d->d_b, F * d->E))) return rc;
        if ((rc = dev_alloc(d, &d->e_c2v1, F * d->E))) return rc;
        if ((rc = dev_alloc(d, &d->e_sgn, F * d->eW))) return rc;
        if ((rc = dev_alloc(d, &d->e_hard, F * d->eW))) return rc;
        if ((rc = dev_alloc(d, &d->e_unsat, F * d->e_stride))) return rc;
        if ((rc = dev_alloc(d, &d->e_done_at, F))) return rc;
        HIPCHK(hipHostMalloc((void **)&d->h_done, sizeof(int) * F * 2));
        HIPCHK(hipEventCreateWithFlags(&d->e_ev[0], hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&d->e_ev[1], hipEventDisableTiming));
        d->poll_every = 2;
        if (const char *e = getenv("QLDPC_POLL_EVERY")) d->poll_every = atoi(e);
        /* measured: replaying the chunks as hipGraphs is not faster than plain launches here (345 vs 321 us per
         * 65 536-VN block; the 2 x ~8 us kernels per iteration are latency-bound on the GPU, not launch-bound on
         * the host), so graphs stay opt-in */
        d->use_graphs = 0; d->graph_frames = -1;
        if (const char *e = getenv("QLDPC_GRAPH")) d->use_graphs = atoi(e) ? 1 : 0;
        /* QLDPC_EDGE_PERSIST=1: the whole decode of up to 8 blocks as ONE launch, one XCD per block (qe_xcd).  Off by default: measured
         * 964 us per 65 536-VN block against 325 us with a launch per pass (a chunk of 64 checks is a chain of dependent L2 round trips,
         * ~8 us, and one XCD has a fifth of the workgroup slots the chunks of a block would fill) */
        d->persist = 0;
        if (const char *e = getenv("QLDPC_EDGE_PERSIST")) d->persist = (atoi(e) && cfg->max_frames <= QE_PERSIST_MAX_FRAMES) ? 1 : 0;
        if ((rc = dev_alloc(d, &d->e_ctl, QE_CTL_WORDS))) return rc;      /* claim / rank / barrier / fault words of qe_xcd */
        return QLDPC_OK;
    }
    if (cfg->schedule == QLDPC_SCHED_FLOODING) {
        if ((rc = make_buckets(d, code->cn_ptr, nullptr, d->M, CN_CAPS, 4, d->cn_buckets))) return rc;
        if ((rc = make_buckets(d, code->vn_ptr, nullptr, d->N, VN_CAPS, 2, d->vn_buckets))) return rc;
    } else {
        d->n_layers = code->n_layers;
        d->layer_buckets.resize((size_t)code->n_layers);
        /* QLDPC_LAYER_REC=0: the layer kernels walk list -> cn_ptr -> cn_var as in rounds 1 - 2 (A/B) */
        const char *rec_env = getenv("QLDPC_LAYER_REC");
        const int *rec_var = (rec_env && atoi(rec_env) == 0) ? nullptr : code->cn_var;
        for (int l = 0; l < code->n_layers; l++)
            if ((rc = make_buckets(d, code->cn_ptr, code->layer_order + code->layer_ptr[l], code->layer_ptr[l + 1] - code->layer_ptr[l], CN_CAPS, 4, d->layer_buckets[(size_t)l], rec_var))) return rc;
        /* A sweep as ONE launch in which a check waits for the earlier checks on its own VNs instead of for the whole layer before it
         * (qldpc_kernels_chain.h).  fp32 messages, 64-frame groups, messages never frozen, check degree <= 40.  Measured on the N = 10^6 code
         * (fixed 50 sweeps, fraction of the HBM peak, launch per layer -> one launch): 64 frames 0.575 -> 0.566, 128: 0.62 -> 0.68, 256: 0.63 -> 0.71,
         * 512: 0.67 -> 0.70, 1 024: 0.69 -> 0.69 (early exit loses from 512 frames on: finished groups still draw tickets); config-2 batch (64
         * groups, 437 checks per layer) 1 528 -> 1 300 Mbit/s; four session decoders side by side 19.0 -> 23.9 ms; with the per-sweep early exit the gain
         * at 128 - 256 frames is inside the run-to-run spread (+2 % .. -9 %).  So: auto = fixed-iteration runs with 2 .. 8 groups and a layer launch of
         * 8 192 .. 65 535 waves; cfg.layer_chain / QLDPC_LAYER_CHAIN = 1 / 0 force it on / off.  (Fewer or more waves in flight per SIMD,
         * QLDPC_CHAIN_WAVES: no difference outside the box-to-box spread, tools/gpu/r3_g47.sh.)  It works on explicit messages: the min-sum rules,
         * which run on the compressed check state below, do not use it. */
        /* Min-sum and AMS sweeps on a compressed check state (qldpc_kernels_cst.h): bit-identical, 0.59 x the bytes on the N = 10^6 code.  fp32 messages,
         * 64-frame groups, messages never frozen, check degree <= 32, and the state must fit the message array.  QLDPC_LAYER_CST = 0 keeps the dc messages. */
        d->layer_cst = 0;
        const char *chain_env = getenv("QLDPC_LAYER_CHAIN");
        const bool chain_asked = chain_env ? atoi(chain_env) != 0 : cfg->layer_chain == 1;      /* an explicit request for the one-launch sweep keeps the explicit messages it works on */
        if (!chain_asked && !d->msg_i8 && !d->msg_half && d->V == 1 && !d->freeze && (family_of(cfg->rule) == QK_FAM_MS || family_of(cfg->rule) == QK_FAM_AMS) && d->max_dc <= 32 &&
            (size_t)d->M * 1024 <= (size_t)d->E * 256)
            d->layer_cst = 1;
        if (const char *e = getenv("QLDPC_LAYER_CST")) d->layer_cst = d->layer_cst && atoi(e) != 0;
        d->chain = 0;
        {
            const long per_layer = (long)d->M / std::max(1, code->n_layers) * d->G;
            bool want = !cfg->enable_syndrome && d->G >= 2 && d->G <= 8 && per_layer >= 8192 && per_layer < 65536 && code->n_layers > 1;
            if (cfg->layer_chain == 1) want = true;
            if (cfg->layer_chain == 2) want = false;
            if (const char *e = getenv("QLDPC_LAYER_CHAIN")) want = atoi(e) != 0;
            if (want && !d->layer_cst && !d->msg_i8 && !d->msg_half && d->V == 1 && !d->freeze && d->max_dc <= 40 && (long)d->M * d->G < (1L << 30) / 64) {
                std::vector<int> depv((size_t)d->E), seen((size_t)d->N, 0), chain_order((size_t)d->M), lastpos((size_t)d->N, -1);
                /* Execution order of the one-launch sweep: the code's layers in their order (so every VN sees its checks in the order of the
                 * launch-per-layer sweep and of the oracle), but INSIDE a layer -- whose checks share no VN, so their order is free -- the checks
                 * whose predecessors sit early in the order go first.  The waves in flight cover a window of a few thousand consecutive
                 * tickets; with the layer in arbitrary order 4 in 10 checks found a predecessor from the tail of the layer before still in that
                 * window and had to wait for it, sorted this way a check's predecessors are about one whole layer behind it. */
                {
                    std::vector<std::pair<int, int>> keyed;
                    int at = 0;
                    for (int l = 0; l < code->n_layers; l++) {
                        keyed.clear();
                        for (int i = code->layer_ptr[l]; i < code->layer_ptr[l + 1]; i++) {
                            const int c = code->layer_order[i];
                            int key = -1;
                            for (int k = code->cn_ptr[c]; k < code->cn_ptr[c + 1]; k++) key = std::max(key, lastpos[(size_t)code->cn_var[k]]);
                            keyed.emplace_back(key, c);
                        }
                        std::stable_sort(keyed.begin(), keyed.end(), [](const std::pair<int, int> &x, const std::pair<int, int> &y) { return x.first < y.first; });
                        for (auto &kc : keyed) {
                            const int c = kc.second;
                            chain_order[(size_t)at] = c;
                            for (int k = code->cn_ptr[c]; k < code->cn_ptr[c + 1]; k++) lastpos[(size_t)code->cn_var[k]] = at;
                            at++;
                        }
                    }
                }
                /* rank of check c among the checks of VN v in execution order: walk the checks in that order, count per VN */
                for (int p = 0; p < d->M; p++) {
                    const int c = chain_order[(size_t)p];
                    for (int k = code->cn_ptr[c]; k < code->cn_ptr[c + 1]; k++) {
                        const int v = code->cn_var[k];
                        const int dv = code->vn_ptr[v + 1] - code->vn_ptr[v];
                        depv[(size_t)k] = (dv << 16) | seen[(size_t)v]++;
                    }
                }
                if (code->max_dv < 32768) {
                    if ((rc = dev_alloc(d, &d->d_chain_order, (size_t)d->M)) || (rc = dev_alloc(d, &d->d_chain_dep, (size_t)d->E)) ||
                        (rc = dev_alloc(d, &d->d_chain_ver, (size_t)d->G * d->N)) || (rc = dev_alloc(d, &d->d_chain_ctl, QC_CTL_WORDS))) return rc;
                    HIPCHK(hipMemcpy(d->d_chain_order, chain_order.data(), sizeof(int) * (size_t)d->M, hipMemcpyHostToDevice));
                    HIPCHK(hipMemcpy(d->d_chain_dep, depv.data(), sizeof(int) * (size_t)d->E, hipMemcpyHostToDevice));
                    d->chain = 1;
                }
            }
        }
    }
    const size_t G = (size_t)d->G, FG = (size_t)d->FG;
    /* the fp32 LLR array [G][N][FG] is allocated on first use (ensure_llr): flooding decoders fed through qldpc_load_bits_* never read one */
    if (cfg->schedule == QLDPC_SCHED_FLOODING) {
        const size_t elems = d->msg_i8 ? (G * d->E * FG + 3) / 4 : (d->msg_half ? (G * d->E * FG + 1) / 2 : G * d->E * FG);      /* floats of storage */
        if (d->msg_i8 && (rc = dev_alloc(d, &d->d_llr8, G * d->N * 64))) return rc;
        const char *coded_env = getenv("QLDPC_CODED_LLR");      /* =0: keep the fp32 LLR array on the load_bits path too (A/B measurements) */
        /* measured: the 8-bit VN pass gets slower with coded LLRs (5.54 vs 6.0 TB/s: four ballot words and selects per VN to save one
         * of nine bytes), so it keeps its quantised LLR array unless QLDPC_CODED_LLR=1 asks otherwise */
        if (!(coded_env && coded_env[0] == '0') && (!d->msg_i8 || (coded_env && coded_env[0] == '1'))) {
            if ((rc = dev_alloc(d, &d->d_ybits, G * d->N * V))) return rc;
            if ((rc = dev_alloc(d, &d->d_fmag, G * FG))) return rc;
            if ((rc = dev_alloc(d, &d->d_fnch, G * FG))) return rc;
            if ((rc = dev_alloc(d, &d->d_vcls, ((size_t)d->N + 3) / 4 * 4))) return rc;      /* read as dwords by the kernels */
        }
        if ((rc = dev_alloc(d, &d->d_a, elems))) return rc;
        if ((rc = dev_alloc(d, &d->d_b, elems))) return rc;
    } else if (d->msg_i8) {
        if ((rc = dev_alloc(d, &d->d_llr8, G * d->N * 64))) return rc;
        if ((rc = dev_alloc(d, &d->d_a, G * d->N * 64))) return rc;      /* post8 */
        if ((rc = dev_alloc(d, &d->d_b, G * d->E * 64))) return rc;      /* msg8, CN-major */
    } else {
        if ((rc = dev_alloc(d, &d->d_a, G * d->N * FG))) return rc;
        if ((rc = dev_alloc(d, &d->d_b, G * d->E * FG))) return rc;
    }
    if ((rc = dev_alloc(d, &d->d_sgn, G * d->N * V))) return rc;
    if (d->msg_i8) d->d_hard = d->d_sgn;      /* integer posteriors: signbit(post) and !(post >= 0) are the same ballot */
    else if ((rc = dev_alloc(d, &d->d_hard, G * d->N * V))) return rc;
    if ((rc = dev_alloc(d, &d->d_unsat, G * V))) return rc;
    if ((rc = dev_alloc(d, &d->d_done, G * V))) return rc;
    if ((rc = dev_alloc(d, &d->d_depth, G * FG))) return rc;
    if ((rc = dev_alloc(d, &d->d_iters, G * FG))) return rc;
    if ((rc = dev_alloc(d, &d->d_active, 4))) return rc;
    HIPCHK(hipMemset(d->d_active, 0, 4 * sizeof(int)));
    if ((rc = dev_alloc(d, &d->d_work, 1))) return rc;
    /* coherent (fine-grained): the host spins on a word the kernel releases at system scope while the stream is still running */
    HIPCHK(hipHostMalloc((void **)&d->h_active, 4 * sizeof(int), hipHostMallocMapped | hipHostMallocCoherent));
    memset(d->h_active, 0, 4 * sizeof(int));
    HIPCHK(hipHostGetDevicePointer((void **)&d->h_active_dev, d->h_active, 0));
    /* active-frame compaction (early exit, flooding, messages not frozen): reserved[0] = 0 auto (batches of >= 4 groups), 1 always, 2 never */
    d->G0 = d->G;
    d->compact_mode = cfg->compact;
    if (const char *e = getenv("QLDPC_COMPACT")) d->compact_mode = atoi(e);
    d->compact_ratio = 0.6f;
    if (const char *e = getenv("QLDPC_COMPACT_RATIO")) { const float x = (float)atof(e); if (x > 0.0f && x <= 1.0f) d->compact_ratio = x; }
    if (!cfg->enable_syndrome || cfg->schedule != QLDPC_SCHED_FLOODING || d->freeze || (d->compact_mode == 0 && d->G < 4)) d->compact_mode = 2;
    if (d->compact_mode != 2) {
        if ((rc = dev_alloc(d, &d->d_gcount, (size_t)d->G))) return rc;
        if ((rc = dev_alloc(d, &d->d_goff, (size_t)d->G + 1))) return rc;
        if (d->poll_every != 1 && !getenv("QLDPC_POLL_EVERY")) d->poll_every = 1;      /* a compaction is decided on the count the status pass leaves */
    }
    return QLDPC_OK;
}

extern "C" int qldpc_decoder_create(const qldpc_code *code, int K, const int *info_bits_pos, const qldpc_decoder_cfg *cfg, qldpc_decoder **out)
{
    if (!out) return QLDPC_EINVAL;
    *out = nullptr;
    if (!code || !cfg) { qldpc_set_error("decoder_create: null code/cfg"); return QLDPC_EINVAL; }
    /* AFF3CT's ctor throws on these (Decoder_LDPC_BP: 'N' vs H.get_n_rows(), K <= N, n_ite > 0) */
    if (K <= 0 || K > code->N) { qldpc_set_error("decoder_create: K=%d not in (0, N=%d]", K, code->N); return QLDPC_ESIZE; }
    if (cfg->n_ite <= 0) { qldpc_set_error("decoder_create: n_ite=%d", cfg->n_ite); return QLDPC_EINVAL; }
    if (cfg->max_frames <= 0) { qldpc_set_error("decoder_create: max_frames=%d", cfg->max_frames); return QLDPC_EINVAL; }
    if (cfg->rule < QLDPC_RULE_MS || cfg->rule > QLDPC_RULE_AMS_MINSTAR) { qldpc_set_error("decoder_create: rule=%d", cfg->rule); return QLDPC_EINVAL; }
    if (cfg->schedule != QLDPC_SCHED_FLOODING && cfg->schedule != QLDPC_SCHED_HLAYERED) { qldpc_set_error("decoder_create: schedule=%d", cfg->schedule); return QLDPC_EINVAL; }
    if (cfg->enable_syndrome && cfg->syndrome_depth < 1) { qldpc_set_error("decoder_create: syndrome_depth=%d", cfg->syndrome_depth); return QLDPC_EINVAL; }
    if (cfg->frames_per_lane != 0 && cfg->frames_per_lane != 1 && cfg->frames_per_lane != 2 && cfg->frames_per_lane != 4) { qldpc_set_error("decoder_create: frames_per_lane=%d", cfg->frames_per_lane); return QLDPC_EINVAL; }
    if (cfg->msg_dtype < 0 || cfg->msg_dtype > 2) { qldpc_set_error("decoder_create: msg_dtype=%d", cfg->msg_dtype); return QLDPC_EINVAL; }
    if (cfg->compact < 0 || cfg->compact > 2 || cfg->layer_chain < 0 || cfg->layer_chain > 2 || cfg->reserved[0]) { qldpc_set_error("decoder_create: compact=%d, layer_chain=%d (0 auto, 1 on, 2 off), reserved words must be zero", cfg->compact, cfg->layer_chain); return QLDPC_EINVAL; }
    if (!(cfg->quant_scale >= 0.0f) || cfg->quant_scale > 64.0f) { qldpc_set_error("decoder_create: quant_scale=%g", (double)cfg->quant_scale); return QLDPC_EINVAL; }
    qldpc_decoder *d = new (std::nothrow) qldpc_decoder();
    if (!d) return QLDPC_ENOMEM;
    int rc = create_impl(code, K, info_bits_pos, cfg, d);
    if (rc != QLDPC_OK) { qldpc_decoder_free(d); return rc; }
    *out = d;
    return QLDPC_OK;
}

/* ---- generations of the per-frame state (active-frame compaction) ---------------------------- */

static void use_gen(qldpc_decoder *d, int k)
{
    const gen_state &n = d->gens[(size_t)k];
    d->G = n.G;
    d->d_sgn = n.sgn; d->d_hard = n.hard; d->d_unsat = n.unsat; d->d_done = n.done; d->d_ybits = n.ybits; d->d_synd = n.synd; d->d_ebits = n.ebits;
    d->d_depth = n.depth; d->d_iters = n.iters; d->d_fmag = n.fmag; d->d_fnch = n.fnch; d->d_llr = n.llr; d->d_llr8 = n.llr8;
}
/* back to the batch as loaded (generation 0 = the decoder's base arrays) */
static void view_reset(qldpc_decoder *d)
{
    if (d->cur_gen != 0) { use_gen(d, 0); d->cur_gen = 0; }
    d->remap_src = nullptr;
}
static void gen0_capture(qldpc_decoder *d)
{
    if (d->gens.empty()) d->gens.resize(1);
    gen_state &n = d->gens[0];
    n.G = n.cap = d->G0;
    n.sgn = d->d_sgn; n.hard = d->d_hard; n.unsat = d->d_unsat; n.done = d->d_done; n.ybits = d->d_ybits; n.synd = d->d_synd; n.ebits = d->d_ebits;
    n.depth = d->d_depth; n.iters = d->d_iters; n.origin = nullptr; n.src = nullptr; n.fmag = d->d_fmag; n.fnch = d->d_fnch; n.llr = d->d_llr; n.llr8 = d->d_llr8;
}

static int ensure_llr(qldpc_decoder *d)
{
    if (d->d_llr) return QLDPC_OK;
    return dev_alloc(d, &d->d_llr, (size_t)d->G * d->N * d->FG);
}

extern "C" int qldpc_decoder_set_stream(qldpc_decoder *d, void *s) { if (!d) return QLDPC_EINVAL; d->stream = (hipStream_t)s; return QLDPC_OK; }
extern "C" size_t qldpc_decoder_device_bytes(const qldpc_decoder *d) { return d ? d->bytes : 0; }
extern "C" int qldpc_last_run_iterations(const qldpc_decoder *dc)
{
    qldpc_decoder *d = const_cast<qldpc_decoder *>(dc);
    if (!d) return QLDPC_EINVAL;
    return d->last_iters;
}

/* early-exit bookkeeping of the last run (FRAMES engine): out[0] = lane-iterations executed (groups that ran an iteration x frames
 * per group, counted by the status pass), out[1] = compactions, out[2] = groups in the last generation, out[3] = frames per group.
 * Synchronises the stream. */
extern "C" int qldpc_last_run_stats(qldpc_decoder *d, long long out[4])
{
    if (!d || !out) return QLDPC_EINVAL;
    out[0] = out[1] = out[2] = out[3] = 0;
    if (d->engine == QLDPC_ENGINE_EDGES || !d->d_work) return QLDPC_OK;
    HIPCHK(hipSetDevice(d->device));
    unsigned long long w = 0;
    HIPCHK(hipMemcpyAsync(&w, d->d_work, sizeof(w), hipMemcpyDeviceToHost, d->stream));
    HIPCHK(hipStreamSynchronize(d->stream));
    out[0] = (long long)w * d->FG; out[1] = d->compactions; out[2] = d->G; out[3] = d->FG;
    return QLDPC_OK;
}

/* reset(): the next decode starts from chk_to_var = 0 (BS/src/main.cpp:389).  Every qldpc_run starts
 * from that state anyway (frames are independent decodes), so this only drops loaded frames. */
extern "C" int qldpc_decoder_reset(qldpc_decoder *d)
{
    if (!d) return QLDPC_EINVAL;
    view_reset(d);
    d->loaded = 0; d->ran = 0; d->n_frames = 0;
    return QLDPC_OK;
}

extern "C" int qldpc_sync(qldpc_decoder *d)
{
    if (!d) return QLDPC_EINVAL;
    HIPCHK(hipSetDevice(d->device));
    HIPCHK(hipStreamSynchronize(d->stream));
    return QLDPC_OK;
}

/* ------------------------------------------------------------------ profiling ---------------- */

struct prof_scope {
    qldpc_decoder *d; int kind; double bytes, moved; hipEvent_t a, b; bool on;
    prof_scope(qldpc_decoder *d_, int kind_, double bytes_, double moved_ = -1.0) : d(d_), kind(kind_), bytes(bytes_), moved(moved_ < 0.0 ? bytes_ : moved_), a(nullptr), b(nullptr), on(d_->prof_on != 0)
    {
        if (on) { (void)hipEventCreate(&a); (void)hipEventCreate(&b); (void)hipEventRecord(a, d->stream); }
    }
    ~prof_scope()
    {
        if (on) { (void)hipEventRecord(b, d->stream); d->prof_pending.push_back({kind, bytes, moved, a, b}); }
    }
};

extern "C" int qldpc_profile_enable(qldpc_decoder *d, int on) { if (!d) return QLDPC_EINVAL; d->prof_on = on; return QLDPC_OK; }

static int prof_fold(qldpc_decoder *d)
{
    if (d->prof_pending.empty()) return QLDPC_OK;
    HIPCHK(hipStreamSynchronize(d->stream));
    for (auto &r : d->prof_pending) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            d->stats[r.kind].launches++;
            d->stats[r.kind].total_ms += ms;
            d->stats[r.kind].alg_bytes += r.bytes;
            d->stats[r.kind].moved_bytes += r.moved;
        }
        (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
    }
    d->prof_pending.clear();
    return QLDPC_OK;
}

extern "C" int qldpc_profile_read(qldpc_decoder *d, qldpc_kernel_stat *out, int cap)
{
    if (!d || (!out && cap > 0)) return QLDPC_EINVAL;
    HIPCHK(hipSetDevice(d->device));
    int rc = prof_fold(d);
    if (rc) return rc;
    int n = 0;
    for (int k = 0; k < KS_COUNT && n < cap; k++) if (d->stats[k].launches) out[n++] = d->stats[k];
    return n;
}

extern "C" int qldpc_profile_clear(qldpc_decoder *d)
{
    if (!d) return QLDPC_EINVAL;
    int rc = prof_fold(d);
    if (rc) return rc;
    for (int k = 0; k < KS_COUNT; k++) { d->stats[k].launches = 0; d->stats[k].total_ms = 0; d->stats[k].alg_bytes = 0; d->stats[k].moved_bytes = 0; }
    return QLDPC_OK;
}

/* ------------------------------------------------------------------ launch helpers ----------- */

/* algorithmic bytes: only live (non-padding) frames are counted */
static double msg_b(const qldpc_decoder *d) { return d->msg_i8 ? 1.0 : (d->msg_half ? 2.0 : 4.0); }
static double llr_b(const qldpc_decoder *d) { return d->msg_i8 ? 1.0 : 4.0; }
/* frames the next launches work on: all loaded frames, or -- once the early-exit loop has polled -- the lanes of the groups still active */
static double live_frames(const qldpc_decoder *d) { return (double)(d->live_lanes > 0 && d->live_lanes < d->n_frames ? d->live_lanes : d->n_frames); }
static double bytes_cn(const qldpc_decoder *d) { return 2.0 * d->E * msg_b(d) * live_frames(d); }
static double bytes_vn(const qldpc_decoder *d, int mode)
{
    if (mode == QK_VN_FIRST) return ((double)d->E * msg_b(d) + d->N * llr_b(d)) * live_frames(d);
    if (mode == QK_VN_POST) return ((double)d->E * msg_b(d) + d->N * llr_b(d)) * live_frames(d);
    return (2.0 * d->E * msg_b(d) + d->N * llr_b(d)) * live_frames(d);
}
/* what a variable-node pass has to fetch in the data form in use: with coded LLRs the channel values are N/8 bytes of received-bit
 * ballots per frame (plus a class byte per VN, shared by all frames) instead of an N * llr_b array */
static double moved_vn(const qldpc_decoder *d, int mode)
{
    if (!d->llr_coded) return bytes_vn(d, mode);
    const double msgs = (mode == QK_VN_NORMAL ? 2.0 : 1.0) * d->E * msg_b(d);
    return (msgs + d->N / 8.0) * live_frames(d) + d->N;
}
static double bytes_layer(const qldpc_decoder *d) { return 4.0 * d->E * msg_b(d) * live_frames(d); }
/* what a sweep moves: with the compressed check state 2 E posterior rows + 8 M state rows per 64-frame group */
static double moved_layer(const qldpc_decoder *d)
{
    if (!d->layer_cst) return bytes_layer(d);
    return (2.0 * d->E * 4.0 + 8.0 * d->M * 4.0) * live_frames(d);
}

template <int V, int MODE>
static int vn_pass(qldpc_decoder *d, float *post_out)
{
    prof_scope ps(d, KS_VN, bytes_vn(d, MODE), moved_vn(d, MODE));
    for (auto &b : d->vn_buckets) { qldpc_launch_vn<V, MODE>(d, b, post_out); LAUNCHCHK(); }
    return QLDPC_OK;
}
template <int V>
static int cn_pass(qldpc_decoder *d, bool first = false)
{
    /* the first pass with coded LLRs reads ballots instead of var_to_chk rows */
    prof_scope ps(d, KS_CN, bytes_cn(d), first && !d->msg_i8 ? ((double)d->E * msg_b(d) + d->N / 8.0) * live_frames(d) : -1.0);
    for (auto &b : d->cn_buckets) { qldpc_launch_cn<V>(d, b, first); LAUNCHCHK(); }
    d->remap_src = nullptr;      /* chk_to_var is in the current generation's layout now, and so is everything after it */
    return QLDPC_OK;
}
template <int V>
static int synd_pass(qldpc_decoder *d, const u64 *mask, int skip_done)
{
    prof_scope ps(d, KS_SYND, (double)d->E * 8.0 * d->G * V);
    const int g8 = (d->G + 7) / 8 * 8;      /* groups rounded up to the XCD count: see qk_syndrome */
    /* early-exit passes go in two launches: an eighth of the checks first; the second launch returns at once for every group in which
     * all running frames already show an unsatisfied check (measured: 1.17 -> 0.69 ms of syndrome passes per config-2 step).  The closing success-flag pass
     * (skip_done = 0) and small codes take one launch. */
    const int Mp = (skip_done && d->M >= 4096) ? ((d->M / 8 + 255) / 256) * 256 : d->M;
    for (int part = 0; part < (Mp < d->M ? 2 : 1); part++) {
        const int lo = part ? Mp : 0, hi = part ? d->M : Mp;
        const int bx = std::max(1, std::min((hi - lo + 255) / 256, 4096 / std::max(1, d->G)));
        hipLaunchKernelGGL((qk_syndrome<V>), dim3((unsigned)((d->G < 8 ? d->G : g8) * bx)), dim3(256), 0, d->stream, mask, d->d_cn_var_t, d->max_dc, d->M, d->N,
                           d->d_unsat, d->d_done, skip_done, d->has_synd ? d->d_synd : nullptr, d->G, bx, lo, hi, part);
    }
    LAUNCHCHK();
    return QLDPC_OK;
}
template <int V>
static int status_pass(qldpc_decoder *d, int ite_done)
{
    prof_scope ps(d, KS_STATUS, 0.0);
    hipLaunchKernelGGL((qk_status<V>), dim3((unsigned)d->G), dim3(64), 0, d->stream, d->d_unsat, d->d_done, d->d_depth, d->d_iters, d->G,
                       d->cfg.syndrome_depth, ite_done, d->d_active, d->d_work, (volatile int *)d->h_active_dev, ++d->poll_seq);
    LAUNCHCHK();
    return QLDPC_OK;
}
/* blocking: how many groups still have unconverged frames after the status pass launched last, and how many frames those are.
 * The kernel writes the counts and then its sequence number into mapped pinned memory; the host spins on the sequence word. */
static int poll_active(qldpc_decoder *d, int *active, int *active_frames = nullptr)
{
    int *h = d->h_active;
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (__atomic_load_n(&h[2], __ATOMIC_ACQUIRE) != d->poll_seq) {
        if ((++spins & 0xfffu) == 0) {
            hipError_t e = hipStreamQuery(d->stream);      /* a failed launch / device fault must not leave the host spinning */
            if (e != hipSuccess && e != hipErrorNotReady) { qldpc_set_error("early-exit poll: %s", hipGetErrorString(e)); return QLDPC_EHIP; }
            if (e == hipSuccess && __atomic_load_n(&h[2], __ATOMIC_ACQUIRE) != d->poll_seq) {
                /* the stream has drained: the report is on its way through the host's memory system, or was never written */
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) { qldpc_set_error("early-exit poll: status report never arrived"); return QLDPC_EHIP; }
            }
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) { qldpc_set_error("early-exit poll: timed out"); return QLDPC_EHIP; }
        }
    }
    *active = h[0];
    if (active_frames) *active_frames = h[1];
    return QLDPC_OK;
}

/*
 * Open the next generation: the `active_frames` frames that have not converged move into ceil(active_frames / FG) full groups
 * (qldpc_kernels_compact.h).  The message arrays are not touched: the next check pass reads through d->remap_src.
 */
template <typename T> static int gen_alloc(qldpc_decoder *d, T **p, size_t n) { return *p ? QLDPC_OK : dev_alloc(d, p, n); }

template <int V>
static int compact(qldpc_decoder *d, int active_frames)
{
    int rc;
    const int FG = d->FG, k = d->cur_gen + 1;
    const int Gn = (active_frames + FG - 1) / FG;
    if ((int)d->gens.size() <= k) d->gens.resize((size_t)k + 1);
    const gen_state o = d->gens[(size_t)k - 1];
    gen_state &n = d->gens[(size_t)k];
    if (n.cap < Gn) {      /* first use (or a larger need than any run before): sized for the largest batch this generation can get */
        (void)hipStreamSynchronize(d->stream);
        (void)hipFree(n.sgn); if (n.hard != n.sgn) (void)hipFree(n.hard); (void)hipFree(n.unsat); (void)hipFree(n.done); (void)hipFree(n.ybits); (void)hipFree(n.synd); (void)hipFree(n.ebits);
        (void)hipFree(n.depth); (void)hipFree(n.iters); (void)hipFree(n.origin); (void)hipFree(n.src); (void)hipFree(n.fmag); (void)hipFree(n.fnch);
        n = gen_state{};
        n.cap = std::max(Gn, std::min(o.cap, (int)(d->compact_ratio * (float)o.cap) + 1));
    }
    const size_t C = (size_t)n.cap;
    if ((rc = gen_alloc(d, &n.sgn, C * d->N * V))) return rc;
    if (d->msg_i8) n.hard = n.sgn;
    else if ((rc = gen_alloc(d, &n.hard, C * d->N * V))) return rc;
    if ((rc = gen_alloc(d, &n.unsat, C * V)) || (rc = gen_alloc(d, &n.done, C * V))) return rc;
    if ((rc = gen_alloc(d, &n.depth, C * FG)) || (rc = gen_alloc(d, &n.iters, C * FG)) || (rc = gen_alloc(d, &n.origin, C * FG)) || (rc = gen_alloc(d, &n.src, C * FG))) return rc;
    if (o.fmag && ((rc = gen_alloc(d, &n.fmag, C * FG)) || (rc = gen_alloc(d, &n.fnch, C * FG)))) return rc;
    if (d->llr_coded && (rc = gen_alloc(d, &n.ybits, C * d->N * V))) return rc;
    if (d->has_synd && (rc = gen_alloc(d, &n.synd, C * d->M * V))) return rc;
    if (d->llr_coded && d->has_erase && (rc = gen_alloc(d, &n.ebits, C * d->N * V))) return rc;
    n.G = Gn;
    n.llr = nullptr; n.llr8 = nullptr;
    const bool rows8 = !d->llr_coded && d->msg_i8, rows32 = !d->llr_coded && !d->msg_i8;
    if (rows8 || rows32) {      /* the decoder reads an LLR array: its rows move too, ping-pong between two side buffers (the loaded batch stays intact) */
        const int side = k & 1;
        if (d->llr_alt_cap[side] < n.cap) {      /* sized by the first generation that uses it (later ones are smaller); grows if a later batch needs more */
            (void)hipStreamSynchronize(d->stream);
            (void)hipFree(d->d_llr_alt[side]); (void)hipFree(d->d_llr8_alt[side]);
            d->d_llr_alt[side] = nullptr; d->d_llr8_alt[side] = nullptr;
            d->llr_alt_cap[side] = n.cap;
        }
        const size_t need = (size_t)d->llr_alt_cap[side] * d->N * (rows8 ? 64 : (size_t)FG);
        if (rows8 && !d->d_llr8_alt[side] && (rc = dev_alloc(d, &d->d_llr8_alt[side], need))) return rc;
        if (rows32 && !d->d_llr_alt[side] && (rc = dev_alloc(d, &d->d_llr_alt[side], need))) return rc;
        n.llr8 = rows8 ? d->d_llr8_alt[side] : nullptr; n.llr = rows32 ? d->d_llr_alt[side] : nullptr;
    }
    prof_scope ps(d, KS_STATUS, 0.0);
    hipLaunchKernelGGL((qk_compact_count<V>), dim3((unsigned)((o.G + 255) / 256)), dim3(256), 0, d->stream, o.done, d->d_gcount, o.G);
    hipLaunchKernelGGL(qk_compact_scan, dim3(1), dim3(256), 0, d->stream, d->d_gcount, d->d_goff, o.G);
    hipLaunchKernelGGL((qk_compact_scatter<V>), dim3((unsigned)o.G), dim3(64), 0, d->stream, o.done, d->d_goff, n.src);
    hipLaunchKernelGGL((qk_compact_fill<V>), dim3((unsigned)Gn), dim3(64), 0, d->stream, n.src, d->d_goff + o.G, o.origin, n.origin, o.fmag, n.fmag, o.fnch, n.fnch,
                       o.depth, n.depth, n.iters, n.done, n.unsat, d->cfg.n_ite, d->N);
    LAUNCHCHK();
    if (d->llr_coded) hipLaunchKernelGGL((qk_compact_ballots<V>), dim3((unsigned)((d->N + 64 * QK_WAVES - 1) / (64 * QK_WAVES)), (unsigned)Gn), dim3(QK_THREADS), 0, d->stream, o.ybits, n.ybits, n.src, d->N);
    if (d->has_synd) hipLaunchKernelGGL((qk_compact_ballots<V>), dim3((unsigned)((d->M + 64 * QK_WAVES - 1) / (64 * QK_WAVES)), (unsigned)Gn), dim3(QK_THREADS), 0, d->stream, o.synd, n.synd, n.src, d->M);
    if (d->llr_coded && d->has_erase) hipLaunchKernelGGL((qk_compact_ballots<V>), dim3((unsigned)((d->N + 64 * QK_WAVES - 1) / (64 * QK_WAVES)), (unsigned)Gn), dim3(QK_THREADS), 0, d->stream, o.ebits, n.ebits, n.src, d->N);
    const unsigned bx = (unsigned)std::max(1, std::min((d->N + QK_WAVES - 1) / QK_WAVES, 8192 / std::max(1, Gn)));
    if (rows32) hipLaunchKernelGGL((qk_compact_rows<V, float>), dim3(bx, (unsigned)Gn), dim3(QK_THREADS), 0, d->stream, o.llr, n.llr, n.src, d->N, 1.0f);
    if (rows8) hipLaunchKernelGGL((qk_compact_rows<V, uint8_t>), dim3(bx, (unsigned)Gn), dim3(QK_THREADS), 0, d->stream, (const uint8_t *)o.llr8, (uint8_t *)n.llr8, n.src, d->N, (uint8_t)0);
    LAUNCHCHK();
    d->cur_gen = k;
    use_gen(d, k);
    d->remap_src = n.src;
    d->compactions++;
    return QLDPC_OK;
}

/* ------------------------------------------------------------------ run ---------------------- */

template <int V>
static int run_flooding(qldpc_decoder *d)
{
    int rc;
    const int n_ite = d->cfg.n_ite;
    /* coded LLRs (fp32 / binary16 messages): the first check pass rebuilds its inputs itself, so _initialize_var_to_chk of
     * iteration 0 (E rows written, E rows read back) is not run at all */
    const bool skip_first = d->msg_i8 ? !d->llr_coded : (d->llr_coded != 0);      /* 8-bit: the first check pass reads llr8 itself */
    if (!skip_first && (rc = vn_pass<V, QK_VN_FIRST>(d, nullptr))) return rc;
    int ite = 0;
    for (; ite < n_ite; ite++) {
        if ((rc = cn_pass<V>(d, skip_first && ite == 0))) return rc;
        if (ite == n_ite - 1) { ite++; break; }
        if ((rc = vn_pass<V, QK_VN_NORMAL>(d, nullptr))) return rc;
        if (d->cfg.enable_syndrome) {
            if ((rc = synd_pass<V>(d, d->d_sgn, 1))) return rc;
            if ((rc = status_pass<V>(d, ite + 1))) return rc;
            if (d->poll_every > 0 && ((ite + 1) % d->poll_every) == 0) {
                int active = 1, left = 0;
                if ((rc = poll_active(d, &active, &left))) return rc;
                d->live_lanes = active * d->FG;
                if (active == 0) { ite++; break; }
                /* few enough frames left: deal them into fewer, full groups (the next check pass reads through the slot map) */
                const int Gn = (left + d->FG - 1) / d->FG;
                if (d->compact_mode != 2 && d->cur_gen + 1 < QLDPC_MAX_GENS && ite + 2 < n_ite && Gn < d->G &&
                    (d->compact_mode == 1 || (float)Gn <= d->compact_ratio * (float)d->G)) {
                    if ((rc = compact<V>(d, left))) return rc;
                }
            }
        }
    }
    d->last_iters = ite;
    /* final _compute_post for every frame (frozen frames recompute the posterior they stopped at) */
    d->post_closes_run = d->cfg.enable_syndrome ? 1 : 0;
    rc = vn_pass<V, QK_VN_POST>(d, nullptr);
    d->post_closes_run = 0;
    return rc;
}

static int bx_of(const qldpc_decoder *d) { return std::max(1, std::min((d->N + QK_WAVES - 1) / QK_WAVES, 8192 / std::max(1, d->G))); }

template <int V>
static int run_layered(qldpc_decoder *d)
{
    int rc;
    const int n_ite = d->cfg.n_ite;
    /* A decoder sized for more frames than the call brought (the sessions' decoders take every batch up to max_blocks) runs its sweeps over the
     * groups that hold frames only.  The layered schedule has no generations: d->G only sizes grids, copies and the status pass, the arrays are
     * strided per group.  Empty groups were skipped inside the kernels before, but their workgroups were still dispatched -- 23 small launches per
     * sweep each carrying up to 8 x the workgroups: the config-3 stream on session decoders sized for 512 blocks 15.7 -> 13.65 ms (sized for 256: 13.9 -> 13.5). */
    struct live_groups {
        qldpc_decoder *d; int saved;
        explicit live_groups(qldpc_decoder *d_) : d(d_), saved(d_->G) { const int gl = std::max(1, (d->n_frames + d->FG - 1) / d->FG); if (gl < d->G) d->G = gl; }
        ~live_groups() { d->G = saved; }
    } live_guard(d);
    const size_t G = (size_t)d->G, FG = (size_t)d->FG;
    const size_t cell = d->msg_i8 ? 1 : sizeof(float);
    HIPCHK(hipMemcpyAsync(d->d_a, d->msg_i8 ? (const void *)d->d_llr8 : (const void *)d->d_llr, G * d->N * FG * cell, hipMemcpyDeviceToDevice, d->stream));   /* var_nodes = Y_N */
    /* messages = 0: fp32 sweeps that never mask a store (freeze = 0) do not clear and re-read E rows per frame group for that -- sweep 0's
     * layer kernels take the messages as zero and write every row (config 5: 0.92 GB not written and not read per decode) */
    const bool skip_clear = !d->msg_i8 && !d->freeze;
    if (!skip_clear) HIPCHK(hipMemsetAsync(d->d_b, 0, G * d->E * FG * cell, d->stream));
    bool chain = false;
    if constexpr (V == 1) {
        chain = d->chain != 0;
        if (chain) {
            if (d->chain_blocks == 0) { d->chain_blocks = qldpc_chain_resident_blocks(d); if (d->chain_blocks < 64) chain = false, d->chain = 0;
                if (getenv("QLDPC_DEBUG")) fprintf(stderr, "libqldpc: one-launch layered sweep: %d resident workgroups of %d waves\n", d->chain_blocks, QK_WAVES); }
            if (chain) {
                HIPCHK(hipMemsetAsync(d->d_chain_ver, 0, sizeof(int) * G * d->N, d->stream));
                HIPCHK(hipMemsetAsync(d->d_chain_ctl, 0, sizeof(int) * QC_CTL_WORDS, d->stream));
            }
        }
    }
    auto ballots = [&]() {
        if (d->msg_i8) hipLaunchKernelGGL(qi_post_ballots, dim3((unsigned)bx_of(d), (unsigned)d->G), dim3(QK_THREADS), 0, d->stream, (const uint32_t *)d->d_a, d->d_sgn, (u64 *)nullptr, d->N, d->d_done);
        else hipLaunchKernelGGL((qk_post_ballots<V>), dim3((unsigned)std::max(1, std::min((d->N + 32 * QK_WAVES - 1) / (32 * QK_WAVES), 8192 / std::max(1, d->G))), (unsigned)d->G), dim3(QK_THREADS), 0, d->stream, d->d_a, d->d_sgn, d->d_hard, d->N, d->d_done);
    };
    int ite = 0;
    for (; ite < n_ite; ite++) {
        {
            prof_scope ps(d, KS_LAYER, bytes_layer(d), moved_layer(d));
            d->layer_first = (skip_clear && ite == 0) ? 1 : 0;
            if (chain) {
                HIPCHK(hipMemsetAsync(d->d_chain_ctl + QC_CTL_SHARD0, 0, sizeof(int) * 32 * QC_SHARDS, d->stream));      /* the ticket counters; the fault word stays */
                qldpc_launch_layer_chain(d, ite);
                LAUNCHCHK();
            }
            else
                for (int l = 0; l < d->n_layers; l++)
                    for (auto &b : d->layer_buckets[(size_t)l]) { qldpc_launch_layer<V>(d, b); LAUNCHCHK(); }
        }
        if (d->cfg.enable_syndrome) {
            {
                prof_scope ps(d, KS_SYND, 0.0, (double)d->N * 4.0 * d->n_frames);      /* the sign ballots of the posteriors: N rows read */
                ballots();
                LAUNCHCHK();
            }
            if ((rc = synd_pass<V>(d, d->d_sgn, 1))) return rc;
            if ((rc = status_pass<V>(d, ite + 1))) return rc;
            if (d->poll_every > 0 && ((ite + 1) % d->poll_every) == 0) {
                int active = 1;
                if ((rc = poll_active(d, &active))) return rc;
                if (active == 0) { ite++; break; }
            }
        }
    }
    d->last_iters = std::min(ite, n_ite);
    ballots();
    LAUNCHCHK();
    if (chain) {
        /* a wait that ran into its bound left the fault word set: the decode cannot be trusted; say so and go back to a launch per layer */
        int ctl[4] = {0, 0, 0, 0};
        HIPCHK(hipMemcpyAsync(ctl, d->d_chain_ctl, sizeof(ctl), hipMemcpyDeviceToHost, d->stream));
        HIPCHK(hipStreamSynchronize(d->stream));
        if (getenv("QLDPC_DEBUG")) fprintf(stderr, "libqldpc: one-launch layered sweeps: %d sweeps x %d checks x %d groups, %d checks waited for a predecessor, %d polls on top\n", d->last_iters, d->M, d->G, ctl[2], ctl[3]);
        if (ctl[1]) {
            d->chain = 0;
            qldpc_set_error("layered decode: a dependency wait of the one-launch sweep timed out (decoder switched back to a launch per layer; run again)");
            return QLDPC_EHIP;
        }
    }
    return QLDPC_OK;
}

template <int V>
static int run_v(qldpc_decoder *d)
{
    view_reset(d);
    gen0_capture(d);
    d->compactions = 0;
    d->live_lanes = 0;
    {
        prof_scope ps(d, KS_STATUS, 0.0);
        hipLaunchKernelGGL((qk_status_init<V>), dim3((unsigned)d->G), dim3(64), 0, d->stream, d->d_unsat, d->d_done, d->d_depth, d->d_iters, d->n_frames, d->cfg.n_ite, d->d_work, d->d_active);
        LAUNCHCHK();
    }
    int rc = d->cfg.schedule == QLDPC_SCHED_FLOODING ? run_flooding<V>(d) : run_layered<V>(d);
    if (rc) return rc;
    /* success flag: syndrome of the hard decision, in every generation (a frame's result lives where it converged) */
    const int last = d->cur_gen;
    for (int k = 0; k <= last && !rc; k++) {
        use_gen(d, k);
        if (hipMemsetAsync(d->d_unsat, 0, sizeof(u64) * (size_t)d->G * V, d->stream) != hipSuccess) rc = QLDPC_EHIP;
        if (!rc) rc = synd_pass<V>(d, d->d_hard, 0);
    }
    use_gen(d, last);
    return rc;
}

/* fn(origin, final_mask, n_slots) once per generation, with that generation's view in place: which slots hold a frame's result
 * and where it belongs in the caller's batch (qk_result_frame) */
template <typename F>
static int for_each_gen(qldpc_decoder *d, F fn)
{
    const int last = d->cur_gen;
    int rc = QLDPC_OK;
    for (int k = 0; k <= last && !rc; k++) {
        use_gen(d, k);
        const gen_state &n = d->gens[(size_t)k];
        rc = fn(n.origin, k < last ? (const u64 *)n.done : (const u64 *)nullptr, k == 0 ? d->n_frames : n.G * d->FG);
    }
    use_gen(d, last);
    return rc;
}

/* ---- edge-parallel engine ----------------------------------------------------------------- */

template <int S>
static void launch_qe_cn(qldpc_decoder *d, const uint32_t *bits, int slot, int syndrome_only)
{
    dim3 grid((unsigned)((d->M + QE_CPB - 1) / QE_CPB), (unsigned)d->n_frames);
    qk_rule r{d->cfg.rule, d->cfg.rule_param};
    float *c2v_out = (slot & 1) ? d->e_c2v1 : d->d_b;        /* ping-pong by iteration parity */
    if (d->cfg.rule == QLDPC_RULE_SPA)
        hipLaunchKernelGGL((qe_cn<S, QK_FAM_SPA>), grid, dim3(QE_THREADS), 0, d->stream, d->d_a, c2v_out, d->d_cn_ptr, d->d_cn_tr, d->d_cn_var, bits,
                           d->M, d->E, d->eW, d->e_unsat, d->e_stride, slot, d->e_done_at, r, syndrome_only, d->has_synd ? d->e_synd : nullptr, (d->M + 31) / 32);
    else
        hipLaunchKernelGGL((qe_cn<S, QK_FAM_MS>), grid, dim3(QE_THREADS), 0, d->stream, d->d_a, c2v_out, d->d_cn_ptr, d->d_cn_tr, d->d_cn_var, bits,
                           d->M, d->E, d->eW, d->e_unsat, d->e_stride, slot, d->e_done_at, r, syndrome_only, d->has_synd ? d->e_synd : nullptr, (d->M + 31) / 32);
}
static int edge_cn(qldpc_decoder *d, const uint32_t *bits, int slot, int syndrome_only)
{
    prof_scope ps(d, syndrome_only ? KS_SYND : KS_CN, syndrome_only ? (double)d->E * 4.0 * d->n_frames : bytes_cn(d));
    switch (d->eS) {
    case 8: launch_qe_cn<8>(d, bits, slot, syndrome_only); break;
    case 16: launch_qe_cn<16>(d, bits, slot, syndrome_only); break;
    case 32: launch_qe_cn<32>(d, bits, slot, syndrome_only); break;
    default: launch_qe_cn<64>(d, bits, slot, syndrome_only); break;
    }
    LAUNCHCHK();
    return QLDPC_OK;
}
/* ite = index of the check pass whose messages are consumed (its parity selects the buffer); sel < 0: per-frame choice */
template <int MODE>
static int edge_vn(qldpc_decoder *d, int ite, int check, int force, float *post_out, int sel)
{
    prof_scope ps(d, KS_VN, bytes_vn(d, MODE));
    dim3 grid((unsigned)((d->N + QE_THREADS - 1) / QE_THREADS), (unsigned)d->n_frames);
    hipLaunchKernelGGL((qe_vn<MODE>), grid, dim3(QE_THREADS), d->e_lds, d->stream, d->d_b, d->e_c2v1, sel, d->cfg.n_ite, d->d_llr, d->d_a, d->e_sgn, d->e_hard, post_out, d->d_vn_ptr,
                       d->N, d->E, d->eW, d->e_unsat, d->e_stride, ite, d->cfg.syndrome_depth, check, d->e_done_at, force);
    LAUNCHCHK();
    return QLDPC_OK;
}

/* launches of iterations [ite0, ite1) on d->stream (+ the decoder-reset prologue in front of iteration 0) */
static int edge_emit(qldpc_decoder *d, int ite0, int ite1)
{
    int rc;
    const int n_ite = d->cfg.n_ite, synd = d->cfg.enable_syndrome;
    if (ite0 == 0) {
        hipLaunchKernelGGL(qe_init, dim3((unsigned)d->n_frames), dim3(64), 0, d->stream, d->e_unsat, d->e_stride, d->e_done_at, d->n_frames);
        LAUNCHCHK();
        if ((rc = edge_vn<QK_VN_FIRST>(d, 0, 0, 0, nullptr, 0))) return rc;
    }
    for (int ite = ite0; ite < ite1; ite++) {
        if ((rc = edge_cn(d, d->e_sgn, ite, 0))) return rc;
        if (ite == n_ite - 1) rc = edge_vn<QK_VN_POST>(d, ite, synd, 0, nullptr, ite & 1);
        else rc = edge_vn<QK_VN_NORMAL>(d, ite, synd, 0, nullptr, ite & 1);
        if (rc) return rc;
    }
    if (ite1 == n_ite + 1) return edge_cn(d, d->e_hard, n_ite + 1, 1);   /* epilogue chunk: success flag of the final hard decisions */
    return QLDPC_OK;
}

/*
 * The launch-bound inner loop as hipGraphs: one instantiated graph per chunk of `poll_every` iterations
 * (kernel arguments differ per iteration, so each chunk has its own graph), captured once per frame count
 * on a private stream and replayed on the caller's stream.  The host only polls between chunks.
 */
static int edge_chunk_graph(qldpc_decoder *d, int chunk, int ite0, int ite1, hipGraphExec_t *out)
{
    if (d->graph_frames != d->n_frames) {
        for (auto &g : d->e_graphs) if (g) (void)hipGraphExecDestroy(g);
        d->e_graphs.clear();
        d->graph_frames = d->n_frames;
    }
    if ((int)d->e_graphs.size() <= chunk) d->e_graphs.resize((size_t)chunk + 1, nullptr);
    if (!d->e_graphs[(size_t)chunk]) {
        if (!d->cap_stream) HIPCHK(hipStreamCreateWithFlags(&d->cap_stream, hipStreamNonBlocking));
        hipStream_t user = d->stream;
        d->stream = d->cap_stream;
        hipGraph_t g = nullptr;
        hipError_t e = hipStreamBeginCapture(d->cap_stream, hipStreamCaptureModeRelaxed);
        int rc = QLDPC_EHIP;
        if (e == hipSuccess) {
            rc = edge_emit(d, ite0, ite1);
            e = hipStreamEndCapture(d->cap_stream, &g);
        }
        d->stream = user;
        if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
        if (e != hipSuccess || !g) { qldpc_set_error("graph capture: %s", hipGetErrorString(e)); return QLDPC_EHIP; }
        hipGraphExec_t x = nullptr;
        e = hipGraphInstantiate(&x, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (e != hipSuccess) { qldpc_set_error("graph instantiate: %s", hipGetErrorString(e)); return QLDPC_EHIP; }
        d->e_graphs[(size_t)chunk] = x;
    }
    *out = d->e_graphs[(size_t)chunk];
    return QLDPC_OK;
}

template <int S, int FAM>
static int launch_persist(qldpc_decoder *d)
{
    const void *fn = (const void *)&qe_xcd<S, FAM>;
    const size_t lds = std::max((size_t)(2 * QE_MAX_EDGES + 16 + QE_CPB) * sizeof(float), d->e_lds);
    if (d->persist_blocks == 0) {
        int per_cu = 0, cus = 0;
        if (lds > 48 * 1024) HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, QE_THREADS, lds));
        HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, d->device));
        /* workgroups that can be resident on ONE XCD (an eighth of the CUs); the occupancy query is advisory (it can over-report by one
         * workgroup per CU), so one per CU is given away and at most 4 are counted on */
        const int reported = per_cu;
        per_cu = std::min(4, per_cu - 1);
        d->persist_blocks = per_cu * (cus / 8);
        if (const char *e = getenv("QLDPC_EDGE_WGS")) d->persist_blocks = atoi(e);      /* workgroups per block (experiments) */
        if (getenv("QLDPC_DEBUG")) fprintf(stderr, "libqldpc: one-launch edge decode: occupancy query %d workgroups per CU, %d CUs, %d workgroups per block, %zu B LDS\n", reported, cus, d->persist_blocks, lds);
        if (d->persist_blocks < 1 || cus < 8) { d->persist = 0; return QLDPC_EUNSUPPORTED; }
    }
    const int nCN = (d->M + QE_CPB - 1) / QE_CPB, nVN = (d->N + QE_THREADS - 1) / QE_THREADS;
    int nb = std::max(1, std::min(d->persist_blocks, std::max(nCN, nVN)));
    qk_rule r{d->cfg.rule, d->cfg.rule_param};
    float *v2c = d->d_a, *c2v0 = d->d_b, *c2v1 = d->e_c2v1;
    const float *llr = d->d_llr;
    const int *cn_ptr = d->d_cn_ptr, *cn_tr = d->d_cn_tr, *cn_var = d->d_cn_var, *vn_ptr = d->d_vn_ptr;
    uint32_t *sgn = d->e_sgn, *hard = d->e_hard;
    int N = d->N, M = d->M, E = d->E, W = d->eW, F = d->n_frames, n_ite = d->cfg.n_ite, synd_on = d->cfg.enable_syndrome, depth = d->cfg.syndrome_depth, stride = d->e_stride;
    int Wm = (d->M + 31) / 32;
    int *unsat = d->e_unsat, *done_at = d->e_done_at, *ctl = d->e_ctl;
    const uint32_t *synd = d->has_synd ? d->e_synd : nullptr;
    static const int ctl_init[QE_CTL_WORDS] = {-1, -1, -1, -1, -1, -1, -1, -1};      /* the rest zero */
    HIPCHK(hipMemcpyAsync(d->e_ctl, ctl_init, sizeof(ctl_init), hipMemcpyHostToDevice, d->stream));
    /* twice the workgroups the 8 XCDs could use: the surplus (and the whole share of an XCD that got no block) exits at once */
    const unsigned grid = (unsigned)(8 * nb * 2);
    hipLaunchKernelGGL((qe_xcd<S, FAM>), dim3(grid), dim3(QE_THREADS), lds, d->stream, v2c, c2v0, c2v1, llr, cn_ptr, cn_tr, cn_var, vn_ptr, sgn, hard,
                       N, M, E, W, F, nb, n_ite, synd_on, depth, unsat, stride, done_at, r, synd, Wm, ctl);
    LAUNCHCHK();
    /* the host needs the verdict now anyway (the launch-per-pass path polls as well): read the control words back; a decode that did
     * not complete (a workgroup that never became resident: the grid drains on its bounded waits and says so) is repeated with a
     * launch per pass by the caller, and the one-launch path is switched off for this decoder */
    int ctl_h[QE_CTL_WORDS];
    HIPCHK(hipMemcpyAsync(ctl_h, d->e_ctl, sizeof(ctl_h), hipMemcpyDeviceToHost, d->stream));
    HIPCHK(hipStreamSynchronize(d->stream));
    if (ctl_h[QE_CTL_FAULT] || ctl_h[QE_CTL_NEXT] < F) {
        fprintf(stderr, "libqldpc: one-launch edge decode did not complete (barrier time-out %d, %d of %d blocks claimed): falling back to a launch per pass\n",
                ctl_h[QE_CTL_FAULT], ctl_h[QE_CTL_NEXT], F);
        d->persist = 0;
        return QLDPC_EUNSUPPORTED;
    }
    d->last_iters = ctl_h[QE_CTL_ITERS];
    return QLDPC_OK;
}

static int run_edges_persist(qldpc_decoder *d)
{
    const bool spa = d->cfg.rule == QLDPC_RULE_SPA;
    switch (d->eS) {
    case 8: return spa ? launch_persist<8, QK_FAM_SPA>(d) : launch_persist<8, QK_FAM_MS>(d);
    case 16: return spa ? launch_persist<16, QK_FAM_SPA>(d) : launch_persist<16, QK_FAM_MS>(d);
    case 32: return spa ? launch_persist<32, QK_FAM_SPA>(d) : launch_persist<32, QK_FAM_MS>(d);
    default: return spa ? launch_persist<64, QK_FAM_SPA>(d) : launch_persist<64, QK_FAM_MS>(d);
    }
}

static int run_edges(qldpc_decoder *d)
{
    int rc;
    d->iters_pending = 0;
    if (d->persist && !d->prof_on && d->n_frames <= QE_PERSIST_MAX_FRAMES) {
        rc = run_edges_persist(d);
        if (rc != QLDPC_EUNSUPPORTED) return rc;      /* not co-resident on this device: the launch-per-pass path below */
    }
    const int n_ite = d->cfg.n_ite, F = d->n_frames, synd = d->cfg.enable_syndrome;
    const int P = (synd && d->poll_every > 0) ? d->poll_every : n_ite;
    const bool graphs = d->use_graphs && !d->prof_on;
    int ite = 0, pending = -1, flip = 0, chunk = 0;
    bool stopped = false;
    while (ite < n_ite) {
        const int ite1 = std::min(n_ite, ite + P);
        if (graphs) {
            hipGraphExec_t x;
            if ((rc = edge_chunk_graph(d, chunk, ite, ite1, &x))) return rc;
            HIPCHK(hipGraphLaunch(x, d->stream));
        } else if ((rc = edge_emit(d, ite, ite1))) return rc;
        ite = ite1; chunk++;
        if (synd && ite < n_ite) {
            /* look-ahead polling: wait for the snapshot queued one chunk ago, so the GPU always has work queued */
            if (pending >= 0) {
                HIPCHK(hipEventSynchronize(d->e_ev[pending]));
                bool all = true;
                for (int f = 0; f < F; f++) all = all && d->h_done[(size_t)pending * F + f] >= 0;
                if (all) { stopped = true; break; }
            }
            HIPCHK(hipMemcpyAsync(d->h_done + (size_t)flip * F, d->e_done_at, sizeof(int) * (size_t)F, hipMemcpyDeviceToHost, d->stream));
            HIPCHK(hipEventRecord(d->e_ev[flip], d->stream));
            pending = flip; flip ^= 1;
        }
    }
    (void)stopped;
    d->last_iters = ite;
    /* success flag: syndrome of the final hard decisions into the last slot */
    return edge_cn(d, d->e_hard, n_ite + 1, 1);
}

extern "C" int qldpc_run(qldpc_decoder *d)
{
    if (!d) return QLDPC_EINVAL;
    if (!d->loaded) { qldpc_set_error("qldpc_run: nothing loaded"); return QLDPC_ESTATE; }
    HIPCHK(hipSetDevice(d->device));
    int rc;
    if (d->engine == QLDPC_ENGINE_EDGES) {
        rc = run_edges(d);
        if (rc == QLDPC_OK) d->ran = 1;
        return rc;
    }
    switch (d->V) {
    case 1: rc = run_v<1>(d); break;
    case 2: rc = run_v<2>(d); break;
    default: rc = run_v<4>(d); break;
    }
    if (rc == QLDPC_OK) d->ran = 1;
    return rc;
}

/* ------------------------------------------------------------------ load --------------------- */

static int check_frames(qldpc_decoder *d, int n_frames, const char *who)
{
    if (n_frames <= 0 || n_frames > d->cfg.max_frames) {
        qldpc_set_error("%s: n_frames=%d not in [1, max_frames=%d]", who, n_frames, d->cfg.max_frames);
        return QLDPC_ESIZE;
    }
    return QLDPC_OK;
}

extern "C" int qldpc_load_llr_dev(qldpc_decoder *d, const float *d_llr, int n_frames)
{
    if (!d || !d_llr) return QLDPC_EINVAL;
    int rc = check_frames(d, n_frames, "qldpc_load_llr_dev");
    if (rc) return rc;
    HIPCHK(hipSetDevice(d->device));
    view_reset(d);
    d->n_frames = n_frames;
    d->has_synd = 0;
    d->has_erase = 0;
    d->llr_coded = 0;
    if (d->engine == QLDPC_ENGINE_EDGES) {
        prof_scope ps(d, KS_LOAD, 2.0 * d->N * 4.0 * n_frames);
        HIPCHK(hipMemcpyAsync(d->d_llr, d_llr, sizeof(float) * (size_t)n_frames * d->N, hipMemcpyDeviceToDevice, d->stream));
        d->loaded = 1; d->ran = 0;
        return QLDPC_OK;
    }
    if ((rc = ensure_llr(d))) return rc;
    {
        prof_scope ps(d, KS_LOAD, 2.0 * d->N * 4.0 * n_frames);
        dim3 grid((unsigned)((d->N + 63) / 64), (unsigned)d->G);
        switch (d->V) {
        case 1: hipLaunchKernelGGL((qk_load_llr<1>), grid, dim3(256), 0, d->stream, d_llr, d->d_llr, d->N, n_frames); break;
        case 2: hipLaunchKernelGGL((qk_load_llr<2>), grid, dim3(256), 0, d->stream, d_llr, d->d_llr, d->N, n_frames); break;
        default: hipLaunchKernelGGL((qk_load_llr<4>), grid, dim3(256), 0, d->stream, d_llr, d->d_llr, d->N, n_frames); break;
        }
        LAUNCHCHK();
        if (d->msg_i8) {
            const size_t n_dwords = (size_t)d->G * d->N * 64;
            hipLaunchKernelGGL(qi_quant_llr, dim3((unsigned)std::min<size_t>((n_dwords + 255) / 256, 16384)), dim3(256), 0, d->stream, d->d_llr, d->d_llr8, n_dwords, d->quant_scale);
            LAUNCHCHK();
        }
    }
    d->loaded = 1; d->ran = 0;
    return QLDPC_OK;
}

extern "C" int qldpc_load_bits_dev(qldpc_decoder *d, const uint32_t *d_bits, const float *d_llr_mag, const uint8_t *d_vn_class, int n_frames)
{
    return qldpc_load_bits_short_dev(d, d_bits, d_llr_mag, d_vn_class, nullptr, n_frames);
}

/* as qldpc_load_bits_dev, with a per-frame count of channel VNs: class-0 VNs at index >= d_n_channel[f] are known bits of
 * frame f (shortening: blocks of different length sharing one code), i.e. pinned like class 1 */
extern "C" int qldpc_load_bits_short_dev(qldpc_decoder *d, const uint32_t *d_bits, const float *d_llr_mag, const uint8_t *d_vn_class, const int *d_n_channel, int n_frames)
{
    if (!d || !d_bits || !d_llr_mag) return QLDPC_EINVAL;
    int rc = check_frames(d, n_frames, "qldpc_load_bits_dev");
    if (rc) return rc;
    HIPCHK(hipSetDevice(d->device));
    view_reset(d);
    d->n_frames = n_frames;
    d->has_synd = 0;
    d->has_erase = 0;
    const int W = (d->N + 31) / 32;
    if (d->engine == QLDPC_ENGINE_EDGES) {
        prof_scope ps(d, KS_LOAD, ((double)W * 4.0 + d->N * 4.0) * n_frames);
        hipLaunchKernelGGL(qe_load_bits, dim3((unsigned)std::min((d->N + 255) / 256, 256), (unsigned)n_frames), dim3(256), 0, d->stream, d_bits, d_llr_mag, d_vn_class,
                           d->d_llr, d->N, W, d_n_channel);
        LAUNCHCHK();
        d->loaded = 1; d->ran = 0;
        return QLDPC_OK;
    }
    d->llr_coded = 0;
    if (d->d_ybits) {
        /* flooding, fp32 / binary16 messages: keep the received bits as ballots and rebuild Y in the VN passes (qk_coded_llr) */
        prof_scope ps(d, KS_LOAD, ((double)W * 4.0 + d->N / 8.0) * n_frames);
        dim3 grid((unsigned)std::max(1, std::min((W + QK_WAVES - 1) / QK_WAVES, 4096 / std::max(1, d->G))), (unsigned)d->G);
        switch (d->V) {
        case 1: hipLaunchKernelGGL((qk_load_syndrome<1>), grid, dim3(QK_THREADS), 0, d->stream, d_bits, d->d_ybits, d->N, W, n_frames); break;
        case 2: hipLaunchKernelGGL((qk_load_syndrome<2>), grid, dim3(QK_THREADS), 0, d->stream, d_bits, d->d_ybits, d->N, W, n_frames); break;
        default: hipLaunchKernelGGL((qk_load_syndrome<4>), grid, dim3(QK_THREADS), 0, d->stream, d_bits, d->d_ybits, d->N, W, n_frames); break;
        }
        LAUNCHCHK();
        const int total = d->G * d->FG;
        hipLaunchKernelGGL(qk_load_frame_consts, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, d->stream, d_llr_mag, d_n_channel, d->d_fmag, d->d_fnch, n_frames, total, d->N);
        LAUNCHCHK();
        if (d_vn_class) HIPCHK(hipMemcpyAsync(d->d_vcls, d_vn_class, (size_t)d->N, hipMemcpyDeviceToDevice, d->stream));
        else HIPCHK(hipMemsetAsync(d->d_vcls, 0, (size_t)d->N, d->stream));
        d->llr_coded = 1;
        d->loaded = 1; d->ran = 0;
        return QLDPC_OK;
    }
    if (d->msg_i8) {
        /* 8-bit variant: straight into the quantised array, no fp32 LLRs in between */
        prof_scope ps(d, KS_LOAD, ((double)W * 4.0 + d->N) * n_frames);
        dim3 grid((unsigned)std::max(1, std::min((W + QK_WAVES - 1) / QK_WAVES, 4096 / std::max(1, d->G))), (unsigned)d->G);
        hipLaunchKernelGGL(qi_load_bits, grid, dim3(QK_THREADS), 0, d->stream, d_bits, d_llr_mag, d_vn_class, d->d_llr8, d->N, W, n_frames, d_n_channel, d->quant_scale);
        LAUNCHCHK();
        d->loaded = 1; d->ran = 0;
        return QLDPC_OK;
    }
    if ((rc = ensure_llr(d))) return rc;
    {
        prof_scope ps(d, KS_LOAD, ((double)W * 4.0 + d->N * 4.0) * n_frames);
        dim3 grid((unsigned)std::max(1, std::min((W + QK_WAVES - 1) / QK_WAVES, 4096 / std::max(1, d->G))), (unsigned)d->G);
        switch (d->V) {
        case 1: hipLaunchKernelGGL((qk_load_bits<1>), grid, dim3(QK_THREADS), 0, d->stream, d_bits, d_llr_mag, d_vn_class, d->d_llr, d->N, W, n_frames, d_n_channel); break;
        case 2: hipLaunchKernelGGL((qk_load_bits<2>), grid, dim3(QK_THREADS), 0, d->stream, d_bits, d_llr_mag, d_vn_class, d->d_llr, d->N, W, n_frames, d_n_channel); break;
        default: hipLaunchKernelGGL((qk_load_bits<4>), grid, dim3(QK_THREADS), 0, d->stream, d_bits, d_llr_mag, d_vn_class, d->d_llr, d->N, W, n_frames, d_n_channel); break;
        }
        LAUNCHCHK();
        if (d->msg_i8) {
            const size_t n_dwords = (size_t)d->G * d->N * 64;
            hipLaunchKernelGGL(qi_quant_llr, dim3((unsigned)std::min<size_t>((n_dwords + 255) / 256, 16384)), dim3(256), 0, d->stream, d->d_llr, d->d_llr8, n_dwords, d->quant_scale);
            LAUNCHCHK();
        }
    }
    d->loaded = 1; d->ran = 0;
    return QLDPC_OK;
}

/* syndrome form: target syndromes for the frames just loaded (cleared again by the next qldpc_load_*) */
extern "C" int qldpc_load_syndrome_dev(qldpc_decoder *d, const uint32_t *d_synd_bits, int n_frames)
{
    if (!d || !d_synd_bits) return QLDPC_EINVAL;
    if (!d->loaded || n_frames != d->n_frames) { qldpc_set_error("qldpc_load_syndrome_dev: load %d frames first (have %d)", n_frames, d->loaded ? d->n_frames : 0); return QLDPC_ESTATE; }
    HIPCHK(hipSetDevice(d->device));
    view_reset(d);
    const int Wm = (d->M + 31) / 32;
    int rc;
    if (d->engine == QLDPC_ENGINE_EDGES) {
        if (!d->e_synd && (rc = dev_alloc(d, &d->e_synd, (size_t)d->cfg.max_frames * Wm))) return rc;
        HIPCHK(hipMemcpyAsync(d->e_synd, d_synd_bits, sizeof(uint32_t) * (size_t)n_frames * Wm, hipMemcpyDeviceToDevice, d->stream));
    } else {
        if (!d->d_synd && (rc = dev_alloc(d, &d->d_synd, (size_t)d->G * d->M * d->V))) return rc;
        dim3 grid((unsigned)std::max(1, std::min((Wm + QK_WAVES - 1) / QK_WAVES, 4096 / std::max(1, d->G))), (unsigned)d->G);
        switch (d->V) {
        case 1: hipLaunchKernelGGL((qk_load_syndrome<1>), grid, dim3(QK_THREADS), 0, d->stream, d_synd_bits, d->d_synd, d->M, Wm, n_frames); break;
        case 2: hipLaunchKernelGGL((qk_load_syndrome<2>), grid, dim3(QK_THREADS), 0, d->stream, d_synd_bits, d->d_synd, d->M, Wm, n_frames); break;
        default: hipLaunchKernelGGL((qk_load_syndrome<4>), grid, dim3(QK_THREADS), 0, d->stream, d_synd_bits, d->d_synd, d->M, Wm, n_frames); break;
        }
        LAUNCHCHK();
    }
    d->has_synd = 1; d->ran = 0;
    return QLDPC_OK;
}

/*
 * Per-frame puncturing: d_erase_bits[n_frames][ceil(N/32)] packed MSB-first, a set bit makes that VN of that frame an erasure
 * (channel LLR 0, BS/src/main.cpp:359-362) whatever its class.  For the frames just loaded (cleared again by the next qldpc_load_*).
 */
extern "C" int qldpc_load_erasures_dev(qldpc_decoder *d, const uint32_t *d_erase_bits, int n_frames)
{
    if (!d || !d_erase_bits) return QLDPC_EINVAL;
    if (!d->loaded || n_frames != d->n_frames) { qldpc_set_error("qldpc_load_erasures_dev: load %d frames first (have %d)", n_frames, d->loaded ? d->n_frames : 0); return QLDPC_ESTATE; }
    HIPCHK(hipSetDevice(d->device));
    view_reset(d);
    const int W = (d->N + 31) / 32;
    int rc;
    if (d->engine == QLDPC_ENGINE_EDGES) {
        hipLaunchKernelGGL(qe_erase, dim3((unsigned)std::min((d->N + 255) / 256, 256), (unsigned)n_frames), dim3(256), 0, d->stream, d_erase_bits, d->d_llr, d->N, W);
        LAUNCHCHK();
        d->ran = 0;
        return QLDPC_OK;
    }
    dim3 grid((unsigned)std::max(1, std::min((W + QK_WAVES - 1) / QK_WAVES, 4096 / std::max(1, d->G))), (unsigned)d->G);
    if (d->llr_coded) {
        if (!d->d_ebits && (rc = dev_alloc(d, &d->d_ebits, (size_t)d->G * d->N * d->V))) return rc;
        switch (d->V) {
        case 1: hipLaunchKernelGGL((qk_load_syndrome<1>), grid, dim3(QK_THREADS), 0, d->stream, d_erase_bits, d->d_ebits, d->N, W, n_frames); break;
        case 2: hipLaunchKernelGGL((qk_load_syndrome<2>), grid, dim3(QK_THREADS), 0, d->stream, d_erase_bits, d->d_ebits, d->N, W, n_frames); break;
        default: hipLaunchKernelGGL((qk_load_syndrome<4>), grid, dim3(QK_THREADS), 0, d->stream, d_erase_bits, d->d_ebits, d->N, W, n_frames); break;
        }
        LAUNCHCHK();
        d->has_erase = 1; d->ran = 0;
        return QLDPC_OK;
    }
    if (d->msg_i8) hipLaunchKernelGGL((qk_erase_rows<4, uint8_t>), grid, dim3(QK_THREADS), 0, d->stream, d_erase_bits, (uint8_t *)d->d_llr8, d->N, W, n_frames);
    else {
        switch (d->V) {
        case 1: hipLaunchKernelGGL((qk_erase_rows<1, float>), grid, dim3(QK_THREADS), 0, d->stream, d_erase_bits, d->d_llr, d->N, W, n_frames); break;
        case 2: hipLaunchKernelGGL((qk_erase_rows<2, float>), grid, dim3(QK_THREADS), 0, d->stream, d_erase_bits, d->d_llr, d->N, W, n_frames); break;
        default: hipLaunchKernelGGL((qk_erase_rows<4, float>), grid, dim3(QK_THREADS), 0, d->stream, d_erase_bits, d->d_llr, d->N, W, n_frames); break;
        }
    }
    LAUNCHCHK();
    d->ran = 0;
    return QLDPC_OK;
}

/* allocate now what the load calls would allocate on first use (the per-frame erasure ballots of the coded-LLR form), so that a
 * caller who must not allocate later -- the daemon after ldpc_init -- can say so */
extern "C" int qldpc_decoder_reserve(qldpc_decoder *d)
{
    if (!d) return QLDPC_EINVAL;
    HIPCHK(hipSetDevice(d->device));
    view_reset(d);
    int rc;
    if (d->engine != QLDPC_ENGINE_EDGES && d->d_ybits && !d->d_ebits && (rc = dev_alloc(d, &d->d_ebits, (size_t)d->G0 * d->N * d->V))) return rc;
    return QLDPC_OK;
}

/* s = H x on the device for packed words (Alice's side of the syndrome form; also a codeword test) */
extern "C" int qldpc_syndrome_dev(qldpc_decoder *d, const uint32_t *d_bits, uint32_t *d_synd_bits, int n_frames)
{
    if (!d || !d_bits || !d_synd_bits || n_frames <= 0) return QLDPC_EINVAL;
    HIPCHK(hipSetDevice(d->device));
    const int Wn = (d->N + 31) / 32, Wm = (d->M + 31) / 32;
    hipLaunchKernelGGL(qk_syndrome_of_bits, dim3((unsigned)((Wm + 255) / 256), (unsigned)n_frames), dim3(256), 0, d->stream, d_bits, d->d_cn_ptr, d->d_cn_var, d_synd_bits,
                       d->M, Wn, Wm);
    LAUNCHCHK();
    return QLDPC_OK;
}

/* ------------------------------------------------------------------ fetch -------------------- */

static int need_ran(qldpc_decoder *d, const char *who)
{
    if (!d->ran) { qldpc_set_error("%s: no completed qldpc_run", who); return QLDPC_ESTATE; }
    return QLDPC_OK;
}

extern "C" int qldpc_fetch_packed_dev(qldpc_decoder *d, uint32_t *d_out)
{
    if (!d || !d_out) return QLDPC_EINVAL;
    int rc = need_ran(d, "qldpc_fetch_packed_dev");
    if (rc) return rc;
    HIPCHK(hipSetDevice(d->device));
    const int W = (d->N + 31) / 32;
    prof_scope ps(d, KS_FETCH, (double)W * 4.0 * d->n_frames);
    if (d->engine == QLDPC_ENGINE_EDGES) {
        HIPCHK(hipMemcpyAsync(d_out, d->e_hard, sizeof(uint32_t) * (size_t)d->n_frames * W, hipMemcpyDeviceToDevice, d->stream));
        return QLDPC_OK;
    }
    return for_each_gen(d, [&](const int *origin, const u64 *fin, int n_slots) -> int {
        dim3 grid((unsigned)std::max(1, std::min((W + QK_WAVES - 1) / QK_WAVES, 4096 / std::max(1, d->G))), (unsigned)d->G);
        switch (d->V) {
        case 1: hipLaunchKernelGGL((qk_fetch_packed<1>), grid, dim3(QK_THREADS), 0, d->stream, d->d_hard, d_out, d->N, W, n_slots, origin, fin); break;
        case 2: hipLaunchKernelGGL((qk_fetch_packed<2>), grid, dim3(QK_THREADS), 0, d->stream, d->d_hard, d_out, d->N, W, n_slots, origin, fin); break;
        default: hipLaunchKernelGGL((qk_fetch_packed<4>), grid, dim3(QK_THREADS), 0, d->stream, d->d_hard, d_out, d->N, W, n_slots, origin, fin); break;
        }
        LAUNCHCHK();
        return (int)QLDPC_OK;
    });
}

extern "C" int qldpc_fetch_info_dev(qldpc_decoder *d, int *d_V_K)
{
    if (!d || !d_V_K) return QLDPC_EINVAL;
    int rc = need_ran(d, "qldpc_fetch_info_dev");
    if (rc) return rc;
    HIPCHK(hipSetDevice(d->device));
    prof_scope ps(d, KS_FETCH, (double)d->K * 4.0 * d->n_frames);
    dim3 grid((unsigned)std::max(1, std::min((d->K + 255) / 256, 64)), (unsigned)d->n_frames);
    if (d->engine == QLDPC_ENGINE_EDGES) {
        hipLaunchKernelGGL(qe_fetch_info, grid, dim3(256), 0, d->stream, d->e_hard, d->d_info_pos, d_V_K, d->K, d->eW);
        LAUNCHCHK();
        return QLDPC_OK;
    }
    return for_each_gen(d, [&](const int *origin, const u64 *fin, int n_slots) -> int {
        dim3 gk((unsigned)std::max(1, std::min((d->K + 255) / 256, 64)), (unsigned)n_slots);
        switch (d->V) {
        case 1: hipLaunchKernelGGL((qk_fetch_info<1>), gk, dim3(256), 0, d->stream, d->d_hard, d->d_info_pos, d_V_K, d->N, d->K, n_slots, origin, fin); break;
        case 2: hipLaunchKernelGGL((qk_fetch_info<2>), gk, dim3(256), 0, d->stream, d->d_hard, d->d_info_pos, d_V_K, d->N, d->K, n_slots, origin, fin); break;
        default: hipLaunchKernelGGL((qk_fetch_info<4>), gk, dim3(256), 0, d->stream, d->d_hard, d->d_info_pos, d_V_K, d->N, d->K, n_slots, origin, fin); break;
        }
        LAUNCHCHK();
        return (int)QLDPC_OK;
    });
}

extern "C" int qldpc_fetch_status_dev(qldpc_decoder *d, int *d_iters, int *d_ok)
{
    if (!d) return QLDPC_EINVAL;
    int rc = need_ran(d, "qldpc_fetch_status_dev");
    if (rc) return rc;
    HIPCHK(hipSetDevice(d->device));
    dim3 grid((unsigned)((d->n_frames + 255) / 256));
    if (d->engine == QLDPC_ENGINE_EDGES) {
        hipLaunchKernelGGL(qe_status_out, grid, dim3(256), 0, d->stream, d->e_unsat, d->e_stride, d->cfg.n_ite + 1, d->e_done_at, d->cfg.n_ite, d_iters, d_ok, d->n_frames);
        LAUNCHCHK();
        return QLDPC_OK;
    }
    return for_each_gen(d, [&](const int *origin, const u64 *fin, int n_slots) -> int {
        dim3 gs((unsigned)((n_slots + 255) / 256));
        switch (d->V) {
        case 1: hipLaunchKernelGGL((qk_status_out<1>), gs, dim3(256), 0, d->stream, d->d_unsat, d->d_iters, d_iters, d_ok, n_slots, origin, fin); break;
        case 2: hipLaunchKernelGGL((qk_status_out<2>), gs, dim3(256), 0, d->stream, d->d_unsat, d->d_iters, d_iters, d_ok, n_slots, origin, fin); break;
        default: hipLaunchKernelGGL((qk_status_out<4>), gs, dim3(256), 0, d->stream, d->d_unsat, d->d_iters, d_iters, d_ok, n_slots, origin, fin); break;
        }
        LAUNCHCHK();
        return (int)QLDPC_OK;
    });
}

extern "C" int qldpc_fetch_post_dev(qldpc_decoder *d, float *d_post_out)
{
    if (!d || !d_post_out) return QLDPC_EINVAL;
    int rc = need_ran(d, "qldpc_fetch_post_dev");
    if (rc) return rc;
    HIPCHK(hipSetDevice(d->device));
    if (d->engine == QLDPC_ENGINE_EDGES) {
        /* posterior of the last executed check pass, written frame-major directly (exact in fixed-iteration mode) */
        return edge_vn<QK_VN_POST>(d, 0, 0, 1, d_post_out, -1);
    }
    if (d->cur_gen != 0) {
        qldpc_set_error("qldpc_fetch_post_dev: the run compacted its active frames (%d times), the messages of frames that converged earlier are gone; "
                        "create the decoder with compact = 2 (or freeze_messages = 1) to read posteriors", d->compactions);
        return QLDPC_EUNSUPPORTED;
    }
    const float *src;
    if (d->cfg.schedule == QLDPC_SCHED_FLOODING) {
        if (!d->d_post) { if ((rc = dev_alloc(d, &d->d_post, (size_t)d->G * d->N * d->FG))) return rc; }
        switch (d->V) {
        case 1: rc = vn_pass<1, QK_VN_POST>(d, d->d_post); break;
        case 2: rc = vn_pass<2, QK_VN_POST>(d, d->d_post); break;
        default: rc = vn_pass<4, QK_VN_POST>(d, d->d_post); break;
        }
        if (rc) return rc;
        src = d->d_post;
    } else if (d->msg_i8) {
        if (!d->d_post) { if ((rc = dev_alloc(d, &d->d_post, (size_t)d->G * d->N * d->FG))) return rc; }
        const size_t n_dwords = (size_t)d->G * d->N * 64;
        hipLaunchKernelGGL(qi_post_to_f32, dim3((unsigned)std::min<size_t>((n_dwords + 255) / 256, 16384)), dim3(256), 0, d->stream, (const uint32_t *)d->d_a, d->d_post, n_dwords);
        LAUNCHCHK();
        src = d->d_post;
    } else {
        src = d->d_a;
    }
    dim3 grid((unsigned)((d->N + 63) / 64), (unsigned)d->G);
    switch (d->V) {
    case 1: hipLaunchKernelGGL((qk_unload_f32<1>), grid, dim3(256), 0, d->stream, src, d_post_out, d->N, d->n_frames); break;
    case 2: hipLaunchKernelGGL((qk_unload_f32<2>), grid, dim3(256), 0, d->stream, src, d_post_out, d->N, d->n_frames); break;
    default: hipLaunchKernelGGL((qk_unload_f32<4>), grid, dim3(256), 0, d->stream, src, d_post_out, d->N, d->n_frames); break;
    }
    LAUNCHCHK();
    return QLDPC_OK;
}

/* ------------------------------------------------------------------ AFF3CT mirror ------------ */

/* decoder->decode_siho(LLRs, dec_bits) with host vectors (BS/src/main.cpp:365): H2D, run, D2H. */
extern "C" int qldpc_decode_siho(qldpc_decoder *d, const float *Y_N, int *V_K, int n_frames)
{
    if (!d || !Y_N || !V_K) return QLDPC_EINVAL;
    int rc = check_frames(d, n_frames, "qldpc_decode_siho");
    if (rc) return rc;
    HIPCHK(hipSetDevice(d->device));
    /* staging buffers for the host-pointer call live with the decoder (sized for max_frames on first use) */
    if (!d->h_in) {
        if ((rc = dev_alloc(d, &d->h_in, (size_t)d->cfg.max_frames * d->N))) return rc;
        if ((rc = dev_alloc(d, &d->h_out, (size_t)d->cfg.max_frames * d->K))) return rc;
    }
    float *din = d->h_in; int *dout = d->h_out;
    rc = QLDPC_OK;
    if (hipMemcpyAsync(din, Y_N, sizeof(float) * (size_t)n_frames * d->N, hipMemcpyHostToDevice, d->stream) != hipSuccess) rc = QLDPC_EHIP;
    if (!rc) rc = qldpc_load_llr_dev(d, din, n_frames);
    if (!rc) rc = qldpc_run(d);
    if (!rc) rc = qldpc_fetch_info_dev(d, dout);
    if (!rc && hipMemcpyAsync(V_K, dout, sizeof(int) * (size_t)n_frames * d->K, hipMemcpyDeviceToHost, d->stream) != hipSuccess) rc = QLDPC_EHIP;
    if (hipStreamSynchronize(d->stream) != hipSuccess && !rc) rc = QLDPC_EHIP;
    if (rc == QLDPC_EHIP) qldpc_set_error("decode_siho: HIP failure (%s)", hipGetErrorString(hipGetLastError()));
    return rc;
}
