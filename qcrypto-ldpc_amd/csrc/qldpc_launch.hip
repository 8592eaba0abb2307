/*
 * qldpc_launch.hip -- dispatch from the decoder's runtime configuration to the template instances of the hot kernels
 * (check-node, variable-node and layer passes of qldpc_kernels*.h) for ONE frames-per-lane value: compiled three times,
 * -DQL_V=1, 2 and 4.
 */
#include "qldpc_engine_int.h"
#include "qldpc_kernels_h16.h"
#include "qldpc_kernels_cst.h"

#ifndef QL_V
#error "compile with -DQL_V=1, 2 or 4"
#endif

template <int V, int CAP, int FAM>
static void launch_cn_one(qldpc_decoder *d, const bucket &b, bool first)
{
    dim3 grid((unsigned)grid_x(b.n, 1), (unsigned)d->G);
    qk_rule r{d->cfg.rule, d->cfg.rule_param};
    if (first && !d->msg_i8) {      /* iteration 0 with coded LLRs: inputs rebuilt from the received bits, var_to_chk is not read (see qk_cn_flood FIRST) */
        qk_coded_llr c{d->d_ybits, d->d_fmag, d->d_fnch, d->d_vcls, d->has_erase ? d->d_ebits : nullptr};
        if (d->msg_half)
            hipLaunchKernelGGL((qk_cn_flood<V, CAP, FAM, __half, true>), grid, dim3(QK_THREADS), 0, d->stream, (const __half *)d->d_a, (__half *)d->d_b, b.d_list, b.n,
                               d->d_cn_ptr, d->d_cn_tr, (size_t)d->E * d->FG, d->d_done, r, d->freeze, d->has_synd ? d->d_synd : nullptr, d->M, d->d_cn_var, d->N, c);
        else
            hipLaunchKernelGGL((qk_cn_flood<V, CAP, FAM, float, true>), grid, dim3(QK_THREADS), 0, d->stream, (const float *)d->d_a, d->d_b, b.d_list, b.n,
                               d->d_cn_ptr, d->d_cn_tr, (size_t)d->E * d->FG, d->d_done, r, d->freeze, d->has_synd ? d->d_synd : nullptr, d->M, d->d_cn_var, d->N, c);
        return;
    }
    if (d->remap_src) {      /* the check pass right after a compaction: var_to_chk is read through the slot map (never the first pass) */
        if (d->msg_i8) {
            if constexpr (V == QI_V && FAM == QK_FAM_MS)
                hipLaunchKernelGGL((qi_cn_flood<CAP, false, true>), grid, dim3(QK_THREADS), 0, d->stream, (const uint32_t *)d->d_a, (uint32_t *)d->d_b, b.d_list, b.n,
                                   d->d_cn_ptr, d->d_cn_tr, (size_t)d->E * 64, d->d_done, qi_rule_of(d), d->has_synd ? d->d_synd : nullptr, d->M, (const uint32_t *)nullptr, (const int *)nullptr, 0,
                                   d->remap_src);
        } else if (d->msg_half)
            hipLaunchKernelGGL((qk_cn_flood<V, CAP, FAM, __half, false, true>), grid, dim3(QK_THREADS), 0, d->stream, (const __half *)d->d_a, (__half *)d->d_b, b.d_list, b.n,
                               d->d_cn_ptr, d->d_cn_tr, (size_t)d->E * d->FG, d->d_done, r, 0, d->has_synd ? d->d_synd : nullptr, d->M, (const int *)nullptr, 0, qk_coded_llr{}, d->remap_src);
        else
            hipLaunchKernelGGL((qk_cn_flood<V, CAP, FAM, float, false, true>), grid, dim3(QK_THREADS), 0, d->stream, (const float *)d->d_a, d->d_b, b.d_list, b.n,
                               d->d_cn_ptr, d->d_cn_tr, (size_t)d->E * d->FG, d->d_done, r, 0, d->has_synd ? d->d_synd : nullptr, d->M, (const int *)nullptr, 0, qk_coded_llr{}, d->remap_src);
        return;
    }
    if (d->msg_i8) {
        if constexpr (V == QI_V && FAM == QK_FAM_MS) {
            if (first)
                hipLaunchKernelGGL((qi_cn_flood<CAP, true>), grid, dim3(QK_THREADS), 0, d->stream, (const uint32_t *)d->d_a, (uint32_t *)d->d_b, b.d_list, b.n,
                                   d->d_cn_ptr, d->d_cn_tr, (size_t)d->E * 64, d->d_done, qi_rule_of(d), d->has_synd ? d->d_synd : nullptr, d->M, d->d_llr8, d->d_cn_var, d->N);
            else
                hipLaunchKernelGGL((qi_cn_flood<CAP, false>), grid, dim3(QK_THREADS), 0, d->stream, (const uint32_t *)d->d_a, (uint32_t *)d->d_b, b.d_list, b.n,
                                   d->d_cn_ptr, d->d_cn_tr, (size_t)d->E * 64, d->d_done, qi_rule_of(d), d->has_synd ? d->d_synd : nullptr, d->M, (const uint32_t *)nullptr, (const int *)nullptr, 0);
        }
        return;
    }
    if (d->msg_half && !d->freeze && d->packed_h16) {
        if constexpr (V == 2 && FAM == QK_FAM_MS && CAP > 0) {      /* packed binary16 fold (qldpc_kernels_h16.h), bit-identical */
            hipLaunchKernelGGL((qh_cn_flood<CAP>), grid, dim3(QK_THREADS), 0, d->stream, (const uint32_t *)d->d_a, (uint32_t *)d->d_b, b.d_list, b.n,
                               d->d_cn_ptr, d->d_cn_tr, (size_t)d->E * 64, d->d_done, r, d->has_synd ? d->d_synd : nullptr, d->M);
            return;
        }
    }
    if (d->msg_half)
        hipLaunchKernelGGL((qk_cn_flood<V, CAP, FAM, __half>), grid, dim3(QK_THREADS), 0, d->stream, (const __half *)d->d_a, (__half *)d->d_b, b.d_list, b.n,
                           d->d_cn_ptr, d->d_cn_tr, (size_t)d->E * d->FG, d->d_done, r, d->freeze, d->has_synd ? d->d_synd : nullptr, d->M);
    else
        hipLaunchKernelGGL((qk_cn_flood<V, CAP, FAM, float>), grid, dim3(QK_THREADS), 0, d->stream, (const float *)d->d_a, d->d_b, b.d_list, b.n,
                           d->d_cn_ptr, d->d_cn_tr, (size_t)d->E * d->FG, d->d_done, r, d->freeze, d->has_synd ? d->d_synd : nullptr, d->M);
}
template <int V, int FAM>
static void launch_cn_fam(qldpc_decoder *d, const bucket &b, bool first)
{
    switch (b.cap) {
    case 8: launch_cn_one<V, 8, FAM>(d, b, first); break;
    case 12: launch_cn_one<V, 12, FAM>(d, b, first); break;
    case 20: launch_cn_one<V, 20, FAM>(d, b, first); break;
    case 40: launch_cn_one<V, 40, FAM>(d, b, first); break;
    default: launch_cn_one<V, 0, FAM>(d, b, first); break;
    }
}
template <int V>
void qldpc_launch_cn(qldpc_decoder *d, const bucket &b, bool first)
{
    switch (family_of(d->cfg.rule)) {
    case QK_FAM_MS: launch_cn_fam<V, QK_FAM_MS>(d, b, first); break;
    case QK_FAM_SPA: launch_cn_fam<V, QK_FAM_SPA>(d, b, first); break;
    case QK_FAM_LSPA: launch_cn_fam<V, QK_FAM_LSPA>(d, b, first); break;
    default: launch_cn_fam<V, QK_FAM_AMS>(d, b, first); break;
    }
}

template <int V, int CAP, int FAM>
static void launch_layer_one(qldpc_decoder *d, const bucket &b)
{
    dim3 grid((unsigned)grid_x(b.n, 1), (unsigned)d->G);
    qk_rule r{d->cfg.rule, d->cfg.rule_param};
    if (d->msg_i8) {
        if constexpr (V == QI_V && FAM == QK_FAM_MS)
            hipLaunchKernelGGL((qi_cn_layer<CAP>), grid, dim3(QK_THREADS), 0, d->stream, (uint32_t *)d->d_a, (uint32_t *)d->d_b, b.d_list, b.n, d->d_cn_ptr, d->d_cn_var,
                               d->N, (size_t)d->E * 64, d->d_done, qi_rule_of(d), d->has_synd ? d->d_synd : nullptr, d->M);
        return;
    }
    if (d->layer_cst) {      /* min-sum / AMS on the compressed check state (qldpc_kernels_cst.h): the host set this only for V = 1, degrees <= 32 */
        if constexpr (V == 1 && (FAM == QK_FAM_MS || FAM == QK_FAM_AMS) && CAP > 0)
            hipLaunchKernelGGL((qk_cn_layer_cst<(CAP > 32 ? 32 : CAP), FAM>), grid, dim3(QK_THREADS), 0, d->stream, d->d_a, d->d_b, b.d_list, b.n, d->d_cn_ptr, d->d_cn_var,
                               d->N, (size_t)d->E * d->FG, d->d_done, r, d->has_synd ? d->d_synd : nullptr, d->M, d->layer_first, b.d_rec, QK_REC_HDR + b.cap);
        return;
    }
    hipLaunchKernelGGL((qk_cn_layer<V, CAP, FAM>), grid, dim3(QK_THREADS), 0, d->stream, d->d_a, d->d_b, b.d_list, b.n, d->d_cn_ptr, d->d_cn_var,
                       d->N, (size_t)d->E * d->FG, d->d_done, r, d->freeze, d->has_synd ? d->d_synd : nullptr, d->M, d->layer_first, CAP > 0 ? b.d_rec : nullptr, QK_REC_HDR + b.cap);
}
template <int V, int FAM>
static void launch_layer_fam(qldpc_decoder *d, const bucket &b)
{
    switch (b.cap) {
    case 8: launch_layer_one<V, 8, FAM>(d, b); break;
    case 12: launch_layer_one<V, 12, FAM>(d, b); break;
    case 20: launch_layer_one<V, 20, FAM>(d, b); break;
    case 40: launch_layer_one<V, 40, FAM>(d, b); break;
    default: launch_layer_one<V, 0, FAM>(d, b); break;
    }
}
template <int V>
void qldpc_launch_layer(qldpc_decoder *d, const bucket &b)
{
    switch (family_of(d->cfg.rule)) {
    case QK_FAM_MS: launch_layer_fam<V, QK_FAM_MS>(d, b); break;
    case QK_FAM_SPA: launch_layer_fam<V, QK_FAM_SPA>(d, b); break;
    case QK_FAM_LSPA: launch_layer_fam<V, QK_FAM_LSPA>(d, b); break;
    default: launch_layer_fam<V, QK_FAM_AMS>(d, b); break;
    }
}

template <int V, int CAP, int UNX, int MODE, typename MT>
static void launch_vn_k(qldpc_decoder *d, const bucket &b, float *post_out)
{
    dim3 grid((unsigned)grid_x(b.n, UNX), (unsigned)d->G);
    if (d->llr_coded) {
        qk_coded_llr c{d->d_ybits, d->d_fmag, d->d_fnch, d->d_vcls, d->has_erase ? d->d_ebits : nullptr};
        hipLaunchKernelGGL((qk_vn_flood<V, CAP, UNX, MODE, MT, true>), grid, dim3(QK_THREADS), 0, d->stream, (const MT *)d->d_b, (const float *)nullptr, (MT *)d->d_a, d->d_sgn, d->d_hard,
                           post_out, b.d_list, b.n, d->d_vn_ptr, d->N, (size_t)d->E * d->FG, d->d_done, c, want_ballots(d, MODE), d->d_vn_tr);
        return;
    }
    hipLaunchKernelGGL((qk_vn_flood<V, CAP, UNX, MODE, MT, false>), grid, dim3(QK_THREADS), 0, d->stream, (const MT *)d->d_b, d->d_llr, (MT *)d->d_a, d->d_sgn, d->d_hard,
                       post_out, b.d_list, b.n, d->d_vn_ptr, d->N, (size_t)d->E * d->FG, d->d_done, qk_coded_llr{}, want_ballots(d, MODE), d->d_vn_tr);
}
template <int V, int CAP, int MODE>
static void launch_vn_one(qldpc_decoder *d, const bucket &b, float *post_out)
{
    constexpr int UN = (CAP > 0 && CAP <= 4) ? 4 : 2;
    if (d->msg_i8) {
        if constexpr (V == QI_V) {
            dim3 grid((unsigned)grid_x(b.n, UN), (unsigned)d->G);
            if (d->llr_coded) {
                qk_coded_llr c{d->d_ybits, d->d_fmag, d->d_fnch, d->d_vcls, d->has_erase ? d->d_ebits : nullptr};
                hipLaunchKernelGGL((qi_vn_flood<CAP, UN, MODE, true>), grid, dim3(QK_THREADS), 0, d->stream, (const uint32_t *)d->d_b, (const uint32_t *)nullptr, (uint32_t *)d->d_a, d->d_sgn, (u64 *)nullptr,
                                   post_out, b.d_list, b.n, d->d_vn_ptr, d->N, (size_t)d->E * 64, d->d_done, c, d->quant_scale, want_ballots(d, MODE), d->d_vn_tr);
            } else
                hipLaunchKernelGGL((qi_vn_flood<CAP, UN, MODE, false>), grid, dim3(QK_THREADS), 0, d->stream, (const uint32_t *)d->d_b, d->d_llr8, (uint32_t *)d->d_a, d->d_sgn, (u64 *)nullptr,
                                   post_out, b.d_list, b.n, d->d_vn_ptr, d->N, (size_t)d->E * 64, d->d_done, qk_coded_llr{}, 0.0f, want_ballots(d, MODE), d->d_vn_tr);
        }
        return;
    }
    if (d->msg_half) launch_vn_k<V, CAP, UN, MODE, __half>(d, b, post_out);
    else launch_vn_k<V, CAP, UN, MODE, float>(d, b, post_out);
}
template <int V, int MODE>
void qldpc_launch_vn(qldpc_decoder *d, const bucket &b, float *post_out)
{
    if (MODE == QK_VN_FIRST) { launch_vn_one<V, 0, MODE>(d, b, post_out); return; }
    switch (b.cap) {
    case 4: launch_vn_one<V, 4, MODE>(d, b, post_out); break;
    case 12: launch_vn_one<V, 12, MODE>(d, b, post_out); break;
    default: launch_vn_one<V, 0, MODE>(d, b, post_out); break;
    }
}

template void qldpc_launch_cn<QL_V>(qldpc_decoder *, const bucket &, bool);
template void qldpc_launch_layer<QL_V>(qldpc_decoder *, const bucket &);
template void qldpc_launch_vn<QL_V, QK_VN_FIRST>(qldpc_decoder *, const bucket &, float *);
template void qldpc_launch_vn<QL_V, QK_VN_NORMAL>(qldpc_decoder *, const bucket &, float *);
template void qldpc_launch_vn<QL_V, QK_VN_POST>(qldpc_decoder *, const bucket &, float *);

#if QL_V == 1
#include "qldpc_kernels_chain.h"

/* one sweep of the horizontal-layered schedule as ONE launch (qldpc_kernels_chain.h); compiled with the V = 1 instances only */
template <int DCMAX, int FAM>
static void launch_chain_one(qldpc_decoder *d, int sweep)
{
    qk_rule r{d->cfg.rule, d->cfg.rule_param};
    const int total = d->M * d->G;
    const unsigned grid = (unsigned)std::max(1, std::min((total + QK_WAVES - 1) / QK_WAVES, d->chain_blocks));
    hipLaunchKernelGGL((qk_cn_layer_chain<DCMAX, FAM>), dim3(grid), dim3(QK_THREADS), (size_t)d->chain_lds, d->stream, d->d_a, d->d_b, d->d_chain_order, d->M, d->G, d->d_cn_ptr, d->d_cn_var,
                       d->d_chain_dep, d->d_chain_ver, d->d_chain_ctl, sweep, d->N, (size_t)d->E * d->FG, d->d_done, r,
                       d->has_synd ? d->d_synd : nullptr, d->layer_first, getenv("QLDPC_DEBUG") ? 1 : 0);
}
template <int FAM>
static void launch_chain_fam(qldpc_decoder *d, int sweep)
{
    if (d->max_dc <= 12) launch_chain_one<12, FAM>(d, sweep);
    else if (d->max_dc <= 20) launch_chain_one<20, FAM>(d, sweep);
    else launch_chain_one<40, FAM>(d, sweep);
}
void qldpc_launch_layer_chain(qldpc_decoder *d, int sweep)
{
    switch (family_of(d->cfg.rule)) {
    case QK_FAM_MS: launch_chain_fam<QK_FAM_MS>(d, sweep); break;
    case QK_FAM_SPA: launch_chain_fam<QK_FAM_SPA>(d, sweep); break;
    case QK_FAM_LSPA: launch_chain_fam<QK_FAM_LSPA>(d, sweep); break;
    default: launch_chain_fam<QK_FAM_AMS>(d, sweep); break;
    }
}
/* resident workgroups the chip holds for the instance this decoder uses (the grid of the persistent launch).  QLDPC_CHAIN_WAVES = n caps it at n
 * workgroups per CU (n waves per SIMD), QLDPC_CHAIN_LDS = bytes additionally makes every workgroup ask for that much LDS it never touches, so that the
 * dispatcher cannot stack more on a CU.  Measured on the N = 10^6 code (tools/gpu/r3_g46.sh, r3_g47.sh): 3 / 4 / 5 waves per SIMD by the grid alone
 * 0.58 / 0.58 / 0.56 of the HBM peak at 64 frames, 1 377 / 1 482 / 1 470 Mbit/s at 128, 1 418 / 1 557 / 1 548 at 256 -- differences at 64 frames are inside
 * the box-to-box spread, so the default stays what the occupancy query gives. */
int qldpc_chain_resident_blocks(qldpc_decoder *d)
{
    int per_cu = 0, cus = 0;
    int n = 8;
    if (const char *e = getenv("QLDPC_CHAIN_WAVES")) n = std::max(1, std::min(8, atoi(e)));
    d->chain_lds = 0;
    if (const char *e = getenv("QLDPC_CHAIN_LDS")) d->chain_lds = std::max(0, std::min(65536, atoi(e)));
    const void *fn = nullptr;
    const int fam = family_of(d->cfg.rule);
#define QC_PICK(F) (d->max_dc <= 12 ? (const void *)&qk_cn_layer_chain<12, F> : (d->max_dc <= 20 ? (const void *)&qk_cn_layer_chain<20, F> : (const void *)&qk_cn_layer_chain<40, F>))
    fn = fam == QK_FAM_MS ? QC_PICK(QK_FAM_MS) : (fam == QK_FAM_SPA ? QC_PICK(QK_FAM_SPA) : (fam == QK_FAM_LSPA ? QC_PICK(QK_FAM_LSPA) : QC_PICK(QK_FAM_AMS)));
#undef QC_PICK
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, QK_THREADS, (size_t)d->chain_lds) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, d->device) != hipSuccess) return 0;
    return std::min(per_cu, n) * cus;
}
#endif
