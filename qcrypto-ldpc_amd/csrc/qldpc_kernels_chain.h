/*
 * qldpc_kernels_chain.h -- horizontal-layered sweep of a SMALL batch as ONE launch: the global barrier between colour layers
 * replaced by per-VN version counters (route (b)).
 *
 * The launch-per-layer sweep (qk_cn_layer) costs one kernel boundary per colour class: 30 launches of 6 667 waves each on the
 * N = 10^6 code with one 64-frame group, every one a single generation of waves that all read and then all write (4.5 TB/s).  But a
 * check does not depend on the whole layer before it, only on the <= dc earlier checks that share a variable node with it
 * (BPSK_nrldpc_sim.m:29-69's recursion: the posterior a check reads is the one the previous check ON THAT VN left).  Here:
 *
 *   - the wavefronts of persistent workgroups draw tickets, each for itself; ticket t = (position p in the execution order, frame group g).  ONE
 *     counter would serialise the sweep (an agent-scope atomic on one address takes ~12 ns: 200 000 tickets = 2.4 ms, measured), so there are
 *     64 counters, counter j handing out tickets j, j + 64, ... to the waves bound to it.  A wave only ever waits for LOWER tickets; a counter
 *     hands its tickets out in rising order, so the lowest unfinished ticket is either held by a running wave, which then waits for
 *     nothing, or the next one a free wave of its counter draws: no wait can last;
 *   - ver[g][v] counts the check updates applied to VN v of group g since the run began; the edge (c, v) carries the rank r of c
 *     among v's dv checks in execution order, and in sweep s check c may read v's posterior once ver[g][v] == s * dv + r;
 *   - the posterior rows are the data handed between waves inside the launch: they are stored and loaded agent-coherently
 *     (global_store / global_load ... sc1: 256-byte rows, two 128-byte lines each written whole by one store instruction of one
 *     wave) and every wave drains its stores (s_waitcnt vmcnt(0)) before one instruction of <= dc lanes adds 1 to the counters of
 *     its VNs -- the hand-off form MI355X_MICROARCH.md measures as valid (counter adds by each storing wave for itself, sc1 poll,
 *     the polling wave loads only after its poll has matched).  tools/sc1_rows.hip: such rows move at the plain rate (5.4-5.8 TB/s).
 *     The check's own messages are private to it within a sweep (the next reader is the next LAUNCH): plain streaming accesses;
 *   - every wait is bounded: a wave that waits too long sets the fault word and goes on (the decode is then wrong, the host sees
 *     the word after the run, reports an error and switches this decoder back to a launch per layer).
 *
 * The arithmetic of a check and the order of the updates on every VN are those of the layer-per-launch sweep, so results are
 * bit-identical to it and to the oracle given the same row order.  fp32 messages, 64-frame groups (V = 1), freeze_messages = 0.
 */
#ifndef QLDPC_KERNELS_CHAIN_H
#define QLDPC_KERNELS_CHAIN_H

#include "qldpc_kernels.h"

#define QC_CTL_FAULT 1
#define QC_CTL_WAITS 2                /* checks whose first poll did not pass / polls they made on top (diagnostics) */
#define QC_CTL_SPINS 3
#define QC_SHARDS 64                  /* ticket counters: shard j hands out tickets j, j + 64, j + 128, ...; each on a 128-byte line of its own */
#define QC_CTL_SHARD0 32
#define QC_CTL_WORDS (QC_CTL_SHARD0 + 32 * QC_SHARDS)
#define QC_SPIN_LIMIT (1u << 20)      /* x s_sleep(2): ~ a second; a healthy wait is microseconds */

__device__ __forceinline__ float qc_ld_sc1(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void qc_st_sc1(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int DCMAX, int FAM>
__global__ __launch_bounds__(QK_THREADS) void qk_cn_layer_chain(float *__restrict__ post, float *__restrict__ msg, const int *__restrict__ order, int M, int G,
                                                        const int *__restrict__ cn_ptr, const int *__restrict__ cn_var, const int *__restrict__ dep,
                                                        int *__restrict__ ver, int *__restrict__ ctl, int sweep,
                                                        int N, size_t group_stride, const u64 *__restrict__ done, qk_rule rule,
                                                        const u64 *__restrict__ synd, int first, int diag)
{
    const int lane = threadIdx.x & 63;      /* every wavefront of the workgroup works for itself: no workgroup-wide step anywhere below */
    const int total = M * G;
    const int shard = (int)((blockIdx.x * QK_WAVES + (threadIdx.x >> 6)) % QC_SHARDS);
    int *my_ctr = ctl + QC_CTL_SHARD0 + 32 * shard;
    /* Software pipeline of one wave.  While ticket t is being worked on the wave already holds ticket t' (its atomic was issued first) and t''s
     * check, edge range, VN ids and dependency words arrive with t's rows.  A check then costs two dependent round trips: {the poll of its VNs'
     * counters, issued in front of the loads of its own message rows, which depend on nothing} and {its posterior rows}.  The stores of a
     * check are not waited for on their own: its counters are raised (QC_DEFER) once the NEXT check's rows have arrived -- s_waitcnt vmcnt(0)
     * there covers the stores too -- unless the next check's poll does not pass at once, in which case the pending counts go out first (the
     * next check may be waiting for this very wave).  Measured alternatives on the N = 10^6 code, one 64-frame group (fraction of the HBM peak):
     * this form 0.566; counts raised right after a drain of the check's own stores 0.47; raised once the next check's poll has returned
     * (stores drained there) 0.32; once poll AND message rows have returned 0.26 -- the acknowledgement of write-through stores is slow, and
     * anything that waits for it in line costs more than the late publication does (which makes half the checks of a one-group sweep
     * wait for a predecessor: 625 000 of 1.2 million, QLDPC_DEBUG counters; 2 % with four groups per check). */
    auto draw = [&]() -> int {
        int t = 0;
        if (lane == 0) t = shard + QC_SHARDS * __hip_atomic_fetch_add(my_ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return t;      /* lane 0 holds it; read with readfirstlane when needed */
    };
    int t = __builtin_amdgcn_readfirstlane(draw());
    if (t >= total) return;
    int p = t / G, g = t - p * G;
    int c = order[p];
    int b = cn_ptr[c], deg = cn_ptr[c + 1] - b;
    int my_v = lane < deg ? cn_var[b + lane] : 0;
    int my_dp = lane < deg ? dep[b + lane] : 0;
    int *pend_ctr = nullptr;      /* counters still to raise for the previous check (this lane's VN), or NULL */
    for (;;) {
        const int tn_raw = draw();                       /* next ticket: in flight from here on */
        const bool live = !qk_group_done<1>(done, g);    /* every check of a finished group skips alike: no waits on it, no counts from it */
        int *vg = ver + (size_t)g * N;
        /* wave-uniform row bases + the lane as a 32-bit offset: the loads and stores take their base from SGPRs (one VGPR of address for all of them) */
        float *pgrp = post + (size_t)g * N * 64;
        float *mrow = msg + (size_t)g * group_stride + (size_t)b * 64;      /* the check's message rows are contiguous */
        float x[DCMAX], m[DCMAX];
        int vn[DCMAX];
        if (live) {
            /* lane k < deg looks after edge k: its VN's counter must have reached sweep * dv + rank */
            const int need = sweep * (my_dp >> 16) + (my_dp & 0xffff);
            bool ok = lane >= deg;
            int seen = 0;
            if (!ok) seen = __hip_atomic_load(vg + my_v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      /* the poll goes first ... */
#pragma unroll
            for (int k = 0; k < DCMAX; k++) vn[k] = cn_var[b + k];      /* scalar loads (the index arrays are padded) */
#pragma unroll
            for (int k = 0; k < DCMAX; k++)                             /* ... the check's own messages right behind it */
                if (k < deg) {
                    if (!first) m[k] = __builtin_nontemporal_load(mrow + k * 64 + lane);
                    else m[k] = 0.0f;
                }
            ok = ok || seen >= need;
            if (!__all(ok)) {
                /* somebody this check depends on has not published yet -- possibly this wave itself: raise the pending counts, then wait */
                if (pend_ctr != nullptr || __any(pend_ctr != nullptr)) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (pend_ctr) __hip_atomic_fetch_add(pend_ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    pend_ctr = nullptr;
                }
                unsigned spins = 0;
                if (diag && lane == 0) __hip_atomic_fetch_add(ctl + QC_CTL_WAITS, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      /* diagnostics (QLDPC_DEBUG only: one address, it serialises): checks that had to wait */
                for (;;) {
                    if (!ok) ok = __hip_atomic_load(vg + my_v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need;
                    if (__all(ok)) { if (diag && lane == 0) __hip_atomic_fetch_add(ctl + QC_CTL_SPINS, (int)spins, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > QC_SPIN_LIMIT) {
                        if (lane == 0) __hip_atomic_store(ctl + QC_CTL_FAULT, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                }
            }
            /* (the polling wave's own loads follow its matched poll in program order: the hand-off form's first row) */
#pragma unroll
            for (int k = 0; k < DCMAX; k++)
                if (k < deg) x[k] = qc_ld_sc1(pgrp + (size_t)vn[k] * 64 + lane);
        }
        /* the next ticket's metadata travels with this ticket's rows */
        const int tn = __builtin_amdgcn_readfirstlane(tn_raw);
        const bool more = tn < total;
        const int pn = more ? tn / G : 0, gn = more ? tn - pn * G : 0;
        const int cnx = order[pn];
        const int bn = cn_ptr[cnx], degn = cn_ptr[cnx + 1] - bn;
        const int my_vn = (more && lane < degn) ? cn_var[bn + lane] : 0;
        const int my_dpn = (more && lane < degn) ? dep[bn + lane] : 0;
        /* everything issued so far has arrived: the previous check's rows have left too -> its counts */
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (pend_ctr) __hip_atomic_fetch_add(pend_ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pend_ctr = nullptr;
        if (live) {
            qk_acc<FAM> acc;
            acc.begin();
            if (synd) acc.sign = (uint32_t)((synd[(size_t)g * M + c] >> lane) & 1ull) << 31;
#pragma unroll
            for (int k = 0; k < DCMAX; k++)
                if (k < deg) { x[k] = x[k] - m[k]; qk_acc_in<FAM>(acc, qk_prep<FAM>(x[k]), rule); }
            acc.finish(rule);
#pragma unroll
            for (int k = 0; k < DCMAX; k++)
                if (k < deg) {
                    const float o = acc.out(qk_prep<FAM>(x[k]), rule);
                    __builtin_nontemporal_store(o, mrow + k * 64 + lane);
                    qc_st_sc1(pgrp + (size_t)vn[k] * 64 + lane, x[k] + o);
                }
            if (lane < deg) pend_ctr = vg + my_v;
        }
        if (!more) break;
        t = tn; p = pn; g = gn; c = cnx; b = bn; deg = degn; my_v = my_vn; my_dp = my_dpn;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      /* the last check's rows have left before its VNs' counters move */
    if (pend_ctr) __hip_atomic_fetch_add(pend_ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#endif /* QLDPC_KERNELS_CHAIN_H */
