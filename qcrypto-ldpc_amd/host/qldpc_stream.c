/*
 * qldpc_stream.c -- BASELINE config 3 timed from C: a stream of sifted-key epochs whose code rate is picked per epoch from the
 * estimated QBER (BS/src/main.cpp:29,235-266), through the reconciliation sessions the ecd2 handlers call
 * (qldpc_recon_encode_blocks on Alice's side, ONE qldpc_recon_decode_blocks call on Bob's), host buffers in and out.
 * Nothing but the C ABI is inside the timed region: no interpreter, no list building.
 *
 *   epochs x key_bits sifted bits, QBER per epoch ~ U[qmin, qmax] (what the daemon would have estimated), Bob's copy = Alice's
 *   through a BSC of that QBER; reconciled = status OK and Bob's words == Alice's.
 *
 * usage: qldpc_stream [-e epochs] [-k key_bits] [-b max_blocks] [-S seed] [-r reps] [-q qmin:qmax] [-l (layered) | -f (flooding); default: the sessions' choice = layered for batches] [-P depth (PEG mothers)]
 *                     [-g rate_gap] [-G gap_profile] [-p (per-kernel profile of Bob's decoders in a second pass)]
 *                     (defaults: 512 epochs x 52 429 bits, max_blocks 512, PEG depth 2 = the library's default)
 * prints one JSON object on stdout.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "qldpc.h"

static uint64_t rng_state;
static inline uint64_t rng_next(void)
{
    uint64_t z = (rng_state += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
static inline double rng_unit(void) { return (double)(rng_next() >> 11) * (1.0 / 9007199254740992.0); }

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static int die(const char *what, int rc)
{
    fprintf(stderr, "qldpc_stream: %s: %s (%d) %s\n", what, qldpc_strerror(rc), rc, qldpc_last_error());
    return 1;
}

struct part { qldpc_recon *r; int n; uint32_t **keys; int *kb; float *qber; qldpc_recon_msg *msgs; const uint32_t **pars; int *status, *corrected, *iters; int rc; };
static void *part_main(void *arg)
{
    struct part *p = arg;
    p->rc = qldpc_recon_decode_blocks(p->r, p->n, p->keys, p->kb, p->qber, p->msgs, p->pars, p->status, p->corrected, p->iters);
    return NULL;
}

int main(int argc, char **argv)
{
    int epochs = 512, key_bits = 52429, batch = 512, reps = 3, layered = -1, profile = 0, peg = 2, opt, a_lanes = 0, b_lanes = 0, verbose = 0, split = 1, gap_profile = 0;
    uint64_t seed = 42;
    double qmin = 0.005, qmax = 0.06, gap = 0.0;
    while ((opt = getopt(argc, argv, "e:k:b:S:r:q:lfpP:g:A:B:vT:G:")) != -1) {
        switch (opt) {
        case 'e': epochs = atoi(optarg); break;
        case 'k': key_bits = atoi(optarg); break;
        case 'b': batch = atoi(optarg); break;
        case 'S': seed = strtoull(optarg, NULL, 0); break;
        case 'r': reps = atoi(optarg); break;
        case 'q': if (sscanf(optarg, "%lf:%lf", &qmin, &qmax) != 2) { fprintf(stderr, "-q qmin:qmax\n"); return 2; } break;
        case 'l': layered = 1; break;
        case 'f': layered = 0; break;
        case 'p': profile = 1; break;
        case 'P': peg = atoi(optarg); break;
        case 'g': gap = atof(optarg); break;
        case 'A': a_lanes = atoi(optarg); break;      /* lanes of Alice's / Bob's calls (QLDPC_RECON_LANES, set per call): diagnostics */
        case 'B': b_lanes = atoi(optarg); break;
        case 'v': verbose = 1; break;
        case 'G': gap_profile = atoi(optarg); break;      /* qldpc_recon_cfg.gap_profile */
        case 'T': split = atoi(optarg); break;      /* experiment: T sessions on T host threads, each decoding every T-th epoch */
        default: fprintf(stderr, "usage: see the head of qldpc_stream.c\n"); return 2;
        }
    }
    if (epochs < 1 || key_bits < 32 || batch < 1 || reps < 1) return 2;
    rng_state = seed;
    const int W = (key_bits + 31) / 32;
    uint32_t *alice = calloc((size_t)epochs * W, 4), *bob = calloc((size_t)epochs * W, 4), *work = calloc((size_t)epochs * W, 4);
    float *qber = calloc((size_t)epochs, sizeof(float));
    if (!alice || !bob || !work || !qber) return 1;
    for (int e = 0; e < epochs; e++) {
        qber[e] = (float)(qmin + (qmax - qmin) * rng_unit());
        for (int w = 0; w < W; w++) alice[(size_t)e * W + w] = (uint32_t)rng_next();
        if (key_bits & 31) alice[(size_t)e * W + W - 1] &= 0xFFFFFFFFu << (32 - (key_bits & 31));
        memcpy(bob + (size_t)e * W, alice + (size_t)e * W, (size_t)W * 4);
        for (int i = 0; i < key_bits; i++)
            if (rng_unit() < qber[e]) bob[(size_t)e * W + (i >> 5)] ^= 1u << (31 - (i & 31));
    }

    qldpc_recon_cfg cfg;
    qldpc_recon_cfg_default(&cfg);
    cfg.max_blocks = batch;
    if (layered >= 0) cfg.schedule = layered ? QLDPC_SCHED_HLAYERED : QLDPC_SCHED_FLOODING;      /* else the default: layered for batches of more than 8 blocks */
    else layered = batch > 8;
    cfg.rate_gap = (float)gap;
    cfg.peg_depth = peg;
    cfg.gap_profile = gap_profile;
    qldpc_recon *ra = NULL, *rb = NULL;
    int rc;
    if ((rc = qldpc_recon_create(&cfg, &ra)) || (rc = qldpc_recon_create(&cfg, &rb))) return die("recon_create", rc);

    qldpc_recon_msg *msgs = calloc((size_t)epochs, sizeof(*msgs));
    const uint32_t **akeys = calloc((size_t)epochs, sizeof(*akeys));
    uint32_t **bkeys = calloc((size_t)epochs, sizeof(*bkeys)), **pars = calloc((size_t)epochs, sizeof(*pars));
    int *kb = calloc((size_t)epochs, sizeof(int)), *cap = calloc((size_t)epochs, sizeof(int));
    int *status = calloc((size_t)epochs, sizeof(int)), *corrected = calloc((size_t)epochs, sizeof(int)), *iters = calloc((size_t)epochs, sizeof(int));
    int max_par = 0;
    for (int e = 0; e < epochs; e++) {
        qldpc_recon_msg m;
        if ((rc = qldpc_recon_plan(ra, key_bits, qber[e], &m))) return die("recon_plan", rc);
        if ((int)((m.code_m + 31) / 32) > max_par) max_par = (int)((m.code_m + 31) / 32);
    }
    uint32_t *parbuf = calloc((size_t)epochs * max_par, 4);
    for (int e = 0; e < epochs; e++) {
        akeys[e] = alice + (size_t)e * W; bkeys[e] = work + (size_t)e * W; pars[e] = parbuf + (size_t)e * max_par;
        kb[e] = key_bits; cap[e] = max_par;
    }

    /* Alice: first call builds the codes of the table, the second is timed */
    char num[16];
    if (a_lanes) { snprintf(num, sizeof(num), "%d", a_lanes); setenv("QLDPC_RECON_LANES", num, 1); }
    if ((rc = qldpc_recon_encode_blocks(ra, epochs, akeys, kb, qber, msgs, pars, cap))) return die("recon_encode_blocks", rc);
    uint32_t *par_first = NULL;
    qldpc_recon_msg *msg_first = NULL;
    if (verbose) {
        par_first = malloc((size_t)epochs * max_par * 4); msg_first = malloc((size_t)epochs * sizeof(*msgs));
        memcpy(par_first, parbuf, (size_t)epochs * max_par * 4); memcpy(msg_first, msgs, (size_t)epochs * sizeof(*msgs));
    }
    double t0 = now_s();
    if ((rc = qldpc_recon_encode_blocks(ra, epochs, akeys, kb, qber, msgs, pars, cap))) return die("recon_encode_blocks", rc);
    const double t_enc = now_s() - t0;
    if (verbose) {
        int diff = 0;
        for (int e = 0; e < epochs; e++)
            if (memcmp(&msg_first[e], &msgs[e], sizeof(*msgs)) || memcmp(par_first + (size_t)e * max_par, parbuf + (size_t)e * max_par, 4 * (size_t)qldpc_recon_parity_words(&msgs[e]))) {
                if (diff < 8) fprintf(stderr, "alice: block %d (rate index %u) differs between her two calls (crc %08x vs %08x)\n", e, msgs[e].rate_index, msg_first[e].crc32, msgs[e].crc32);
                diff++;
            }
        fprintf(stderr, "alice: %d of %d blocks differ between two identical encode calls\n", diff, epochs);
        for (int e = 0; e < epochs; e++)
            if (msgs[e].crc32 != qldpc_crc32_words(akeys[e], key_bits)) { fprintf(stderr, "alice: block %d: device CRC %08x, host CRC %08x\n", e, msgs[e].crc32, qldpc_crc32_words(akeys[e], key_bits)); break; }
    }
    if (b_lanes) { snprintf(num, sizeof(num), "%d", b_lanes); setenv("QLDPC_RECON_LANES", num, 1); }

    if (split > 1) {
        /* experiment: the stream dealt to `split` sessions, each on its own host thread (every split-th epoch), timed together */
        struct part P[8];
        pthread_t th[8];
        if (split > 8) split = 8;
        for (int t = 0; t < split; t++) {
            struct part *p = &P[t];
            memset(p, 0, sizeof(*p));
            if ((rc = qldpc_recon_create(&cfg, &p->r))) return die("recon_create", rc);
            p->keys = calloc((size_t)epochs, sizeof(*p->keys)); p->pars = calloc((size_t)epochs, sizeof(*p->pars)); p->kb = calloc((size_t)epochs, sizeof(int));
            p->qber = calloc((size_t)epochs, sizeof(float)); p->msgs = calloc((size_t)epochs, sizeof(*msgs));
            p->status = calloc((size_t)epochs, sizeof(int)); p->corrected = calloc((size_t)epochs, sizeof(int)); p->iters = calloc((size_t)epochs, sizeof(int));
            int seen[8] = {0};
            for (int e = 0; e < epochs; e++) {
                const int ri = msgs[e].rate_index & 7;
                if (seen[ri]++ % split != t) continue;      /* every split-th epoch OF EACH RATE: the rate groups are halved */
                p->keys[p->n] = bkeys[e]; p->pars[p->n] = pars[e]; p->kb[p->n] = kb[e]; p->qber[p->n] = qber[e]; p->msgs[p->n] = msgs[e]; p->n++;
            }
        }
        double bestT = 1e30;
        int goodT = 0;
        for (int rep = -1; rep < reps; rep++) {
            memcpy(work, bob, (size_t)epochs * W * 4);
            t0 = now_s();
            for (int t = 0; t < split; t++) pthread_create(&th[t], NULL, part_main, &P[t]);
            for (int t = 0; t < split; t++) pthread_join(th[t], NULL);
            const double dt = now_s() - t0;
            if (rep >= 0 && dt < bestT) bestT = dt;
            goodT = 0;
            for (int t = 0; t < split; t++) { if (P[t].rc) return die("decode (split)", P[t].rc); for (int i = 0; i < P[t].n; i++) goodT += P[t].status[i] == QLDPC_OK; }
        }
        printf("{\"experiment\": \"%d sessions on %d threads\", \"reconciled\": %d, \"ms_best\": %.3f, \"Mbit_s_best\": %.1f}\n", split, split, goodT, bestT * 1e3, (double)goodT * key_bits / bestT / 1e6);
        return 0;
    }
    /* Bob: one untimed pass builds his decoders, then `reps` timed passes over fresh copies of his keys */
    double best = 1e30, sum = 0.0;
    int good = 0;
    long leaked = 0;
    double it_sum = 0.0;
    for (int rep = -1; rep < reps; rep++) {
        memcpy(work, bob, (size_t)epochs * W * 4);
        t0 = now_s();
        rc = qldpc_recon_decode_blocks(rb, epochs, bkeys, kb, qber, msgs, (const uint32_t *const *)pars, status, corrected, iters);
        const double dt = now_s() - t0;
        if (rc) return die("recon_decode_blocks", rc);
        if (verbose) {
            int bad = 0, pos[8] = {0};
            fprintf(stderr, "bob rep %d:", rep);
            for (int e = 0; e < epochs; e++) {
                const int ri = msgs[e].rate_index & 7;
                if (status[e] != QLDPC_OK) { if (bad < 12) fprintf(stderr, " [blk %d rate %d pos-in-group %d it %d]", e, ri, pos[ri], iters[e]); bad++; }
                pos[ri]++;
            }
            fprintf(stderr, " -> %d failed\n", bad);
        }
        if (rep < 0) continue;
        sum += dt;
        if (dt < best) best = dt;
        good = 0; leaked = 0; it_sum = 0.0;
        for (int e = 0; e < epochs; e++) {
            it_sum += iters[e];
            if (status[e] == QLDPC_OK && !memcmp(work + (size_t)e * W, alice + (size_t)e * W, (size_t)W * 4)) { good++; leaked += qldpc_recon_leaked_bits(&msgs[e]); }
        }
    }
    int per_rate[8] = {0}, fail_rate[8] = {0}, maxed[8] = {0}, it_max[8] = {0};
    double it_mean[8] = {0};
    for (int e = 0; e < epochs; e++) {
        if (iters[e] > it_max[msgs[e].rate_index & 7]) it_max[msgs[e].rate_index & 7] = iters[e];
        it_mean[msgs[e].rate_index & 7] += iters[e];
        per_rate[msgs[e].rate_index & 7]++;
        if (!(status[e] == QLDPC_OK && !memcmp(work + (size_t)e * W, alice + (size_t)e * W, (size_t)W * 4))) fail_rate[msgs[e].rate_index & 7]++;
        if (iters[e] >= cfg.n_ite) maxed[msgs[e].rate_index & 7]++;
    }
    const double mean = sum / reps;
    printf("{\"workload\": \"%d epochs x %d bits, QBER ~ U[%.3f, %.3f] seed %llu, batches of <= %d blocks, %s, one decode_blocks call\", "
           "\"reconciled\": %d, \"epochs\": %d, \"ms_mean\": %.3f, \"ms_best\": %.3f, \"Mbit_s_mean\": %.1f, \"Mbit_s_best\": %.1f, "
           "\"leaked_fraction\": %.4f, \"avg_iterations\": %.2f, \"alice_encode_ms\": %.3f, \"epochs_per_rate\": [%d, %d, %d, %d], \"failed_per_rate\": [%d, %d, %d, %d], "
           "\"at_max_iterations_per_rate\": [%d, %d, %d, %d], \"mean_iterations_per_rate\": [%.1f, %.1f, %.1f, %.1f], \"max_iterations_per_rate\": [%d, %d, %d, %d]",
           epochs, key_bits, qmin, qmax, (unsigned long long)seed, batch, layered ? "layered" : "flooding", good, epochs, mean * 1e3, best * 1e3,
           (double)good * key_bits / mean / 1e6, (double)good * key_bits / best / 1e6, (double)leaked / fmax(1.0, (double)good * key_bits), it_sum / epochs,
           t_enc * 1e3, per_rate[0], per_rate[1], per_rate[2], per_rate[3], fail_rate[0], fail_rate[1], fail_rate[2], fail_rate[3], maxed[0], maxed[1], maxed[2], maxed[3],
           it_mean[0] / fmax(1, per_rate[0]), it_mean[1] / fmax(1, per_rate[1]), it_mean[2] / fmax(1, per_rate[2]), it_mean[3] / fmax(1, per_rate[3]), it_max[0], it_max[1], it_max[2], it_max[3]);
    if (profile) {
        qldpc_kernel_stat st[16];
        qldpc_recon_profile_enable(rb, 1);
        memcpy(work, bob, (size_t)epochs * W * 4);
        if ((rc = qldpc_recon_decode_blocks(rb, epochs, bkeys, kb, qber, msgs, (const uint32_t *const *)pars, status, corrected, iters))) return die("recon_decode_blocks", rc);
        const int n = qldpc_recon_profile_read(rb, st, 16);
        printf(", \"kernels\": {");
        for (int i = 0; i < n; i++)
            printf("%s\"%s\": {\"launches\": %llu, \"ms\": %.3f, \"alg_GB\": %.3f}", i ? ", " : "", st[i].name, (unsigned long long)st[i].launches, st[i].total_ms, st[i].alg_bytes / 1e9);
        printf("}");
    }
    printf("}\n");
    qldpc_recon_free(ra);
    qldpc_recon_free(rb);
    return good == epochs ? 0 : 3;
}
