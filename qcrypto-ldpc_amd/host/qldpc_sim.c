/*
 * qldpc_sim.c -- the reference's QKD-over-BSC simulation loop (BS/src/main.cpp:213-409,
 * VAR/main.cpp (dvb-v1.0.2):427-458) written in C against libqldpc's C ABI.
 *
 *   for each BER: pick the code, then per batch of frames
 *     source -> encoder -> BSC(ber) on the key VNs -> LLR = +-ln((1-p)/p)            (main.cpp:340-348)
 *     parity VNs pinned to +-CONFIRMED_BIT_LLR                                        (main.cpp:351-354)
 *     decoder->decode_siho, compare the K info bits, count bit / frame errors         (main.cpp:365-388)
 *   one AFF3CT-style row per BER: FRA | BE | FE | BER | FER | SIM_THR (Mb/s)          (Reporter_BFER/_throughput)
 *
 * usage: qldpc_sim [-N n] [-K k | -a alist | -q qc] [-r MS|OMS|NMS|SPA|LSPA|AMS_MIN|AMS_MINSTAR_L2|AMS_MINSTAR] [-p param]
 *                  [-i n_ite] [-f frames_per_ber] [-b batch] [-s ber_min:ber_max:ber_step] [-S seed] [-l (layered)] [-n (no syndrome)]
 *                  [-P depth (progressive-edge-growth information part instead of the seeded socket shuffle)]
 *                  [-d parity_ber (dirty disclosed parity bits, BS/data_dvb/data5)]
 *                  [-G IDENTITY|LU_DEC|QC (encoder construction: p.G_method / Encoder_LDPC_from_QC)]
 *                  [-R (with -e: report every random pattern -- one per batch -- and keep the best; -o file writes its VN indices)]
 *                  [-Q 32|16|8 (message storage: fp32 = the AFF3CT float build, binary16, 8-bit fixed-point min-sum)]
 *                  [-e f (puncture parity bits to reach the rate min_cr(ber, f); random pattern re-drawn per batch, main.cpp:321-333,359-362)]
 */
#include <math.h>
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
    fprintf(stderr, "qldpc_sim: %s: %s (%d) %s\n", what, qldpc_strerror(rc), rc, qldpc_last_error());
    return 1;
}

int main(int argc, char **argv)
{
    int N = 8192, K = 6554, n_ite = 50, frames = 256, batch = 256, layered = 0, synd = 1, peg = 0, msg_bits = 32, search = 0, opt;
    const char *pattern_out = NULL;
    double parity_ber = 0.0;      /* > 0: the disclosed parity bits are themselves wrong with this probability (main.cpp (test effect of dirty parities)) */
    double target_eff = 0.0;      /* > 0: puncture parity bits up to min_cr(ber, f), as BS/src/main.cpp:235-333 does */
    const char *alist = NULL, *qc = NULL, *rule_name = "NMS";
    float param = 0.75f;
    double ber_min = 0.01, ber_max = 0.03, ber_step = 0.005;
    uint64_t seed = 0;
    const char *g_method = NULL;
    while ((opt = getopt(argc, argv, "N:K:a:q:r:p:i:f:b:s:S:P:e:d:Q:o:G:Rln")) != -1) {
        switch (opt) {
        case 'N': N = atoi(optarg); break;
        case 'K': K = atoi(optarg); break;
        case 'a': alist = optarg; break;
        case 'q': qc = optarg; break;
        case 'r': rule_name = optarg; break;
        case 'p': param = (float)atof(optarg); break;
        case 'i': n_ite = atoi(optarg); break;
        case 'f': frames = atoi(optarg); break;
        case 'b': batch = atoi(optarg); break;
        case 's': if (sscanf(optarg, "%lf:%lf:%lf", &ber_min, &ber_max, &ber_step) != 3) { fprintf(stderr, "-s min:max:step\n"); return 2; } break;
        case 'S': seed = strtoull(optarg, NULL, 0); break;
        case 'P': peg = atoi(optarg); break;
        case 'e': target_eff = atof(optarg); break;
        case 'd': parity_ber = atof(optarg); break;
        case 'Q': msg_bits = atoi(optarg); break;
        case 'R': search = 1; break;
        case 'o': pattern_out = optarg; break;
        case 'G': g_method = optarg; break;      /* p.G_method (VAR/main.cpp (alist-v1.0.1):135): IDENTITY | LU_DEC; QC = Encoder_LDPC_from_QC ((qc):145) */
        case 'l': layered = 1; break;
        case 'n': synd = 0; break;
        default: fprintf(stderr, "see the header of qldpc_sim.c for usage\n"); return 2;
        }
    }
    static const char *names[] = {"MS", "OMS", "NMS", "SPA", "LSPA", "AMS_MIN", "AMS_MINSTAR_L2", "AMS_MINSTAR"};
    int rule = -1;
    for (int i = 0; i < 8; i++) if (!strcmp(rule_name, names[i])) rule = i;
    if (rule < 0) { fprintf(stderr, "unknown rule %s\n", rule_name); return 2; }

    qldpc_code *H = NULL;
    int rc = alist ? qldpc_code_from_alist(alist, &H) : qc ? qldpc_code_from_qc(qc, &H) : peg ? qldpc_code_ira_peg(N, K, 0.125f, 11, 3, peg, 7, &H) : qldpc_code_ira(N, K, 0.125f, 11, 3, 7, &H);
    if (rc) return die("code", rc);
    N = qldpc_code_n(H);
    qldpc_encoder *enc = NULL;
    if ((rc = qldpc_encoder_create(H, g_method ? g_method : qldpc_code_is_ira(H) ? "IRA" : "IDENTITY", 0, &enc))) return die("encoder", rc);
    K = qldpc_encoder_k(enc);
    int *pos = (int *)malloc(sizeof(int) * (size_t)K);
    qldpc_encoder_info_bits_pos(enc, pos);
    char *is_info = (char *)calloc((size_t)N, 1);
    for (int i = 0; i < K; i++) is_info[pos[i]] = 1;

    qldpc_decoder_cfg cfg;
    qldpc_decoder_cfg_default(&cfg);
    cfg.schedule = layered ? QLDPC_SCHED_HLAYERED : QLDPC_SCHED_FLOODING;
    cfg.rule = rule; cfg.rule_param = param; cfg.n_ite = n_ite; cfg.enable_syndrome = synd; cfg.syndrome_depth = 1; cfg.max_frames = batch;
    if (msg_bits != 32 && msg_bits != 16 && msg_bits != 8) { fprintf(stderr, "-Q 32 | 16 | 8\n"); return 2; }
    cfg.msg_dtype = msg_bits == 16 ? 1 : (msg_bits == 8 ? 2 : 0);      /* 16: binary16 message storage; 8: fixed-point min-sum */
    qldpc_decoder *dec = NULL;
    if ((rc = qldpc_decoder_create(H, K, pos, &cfg, &dec))) return die("decoder", rc);

    printf("# * libqldpc %d on HIP device 0; Decoder_LDPC_BP_%s_Update_rule_%s (param %g), n_ite %d, syndrome %d, %d-bit messages\n", qldpc_version(),
           layered ? "horizontal_layered" : "flooding", rule_name, (double)param, n_ite, synd, msg_bits);
    printf("#    ** Info. bits (K) = %d\n#    ** Frame size (N) = %d\n#    ** Code rate  (R) = %f\n#    ** max CN degree   = %d\n", K, N, (double)K / N, qldpc_code_max_cn_degree(H));
    printf("#    ** Est. QKD Key Rate After Priv Amp = %f\n", (double)(K - (N - K)) / (double)K);
    printf("# %8s | %8s | %8s | %8s | %9s | %9s | %10s\n", "EP", "FRA", "BE", "FE", "BER", "FER", "SIM_THR");
    printf("# %8s | %8s | %8s | %8s | %9s | %9s | %10s\n", "", "", "", "", "", "", "(Mb/s)");

    int *ref_bits = (int *)malloc(sizeof(int) * (size_t)batch * K), *enc_bits = (int *)malloc(sizeof(int) * (size_t)batch * N);
    int *dec_bits = (int *)malloc(sizeof(int) * (size_t)batch * K);
    float *llr = (float *)malloc(sizeof(float) * (size_t)batch * N);
    rng_state = seed;
    int *par_pos = (int *)malloc(sizeof(int) * (size_t)(N - K + 1));
    int n_par = 0;
    for (int v = 0; v < N; v++) if (!is_info[v]) par_pos[n_par++] = v;
    for (double ber = ber_min; ber <= ber_max + 1e-12; ber += ber_step) {
        const float L = qldpc_bsc_llr((float)ber);
        /* parity_bits_to_punct(INFO_B, TTL_B, GOAL_CR) with GOAL_CR = min_cr(QBER, EFF)   (BS/src/main.cpp:29,34,280) */
        int n_punct = 0;
        if (target_eff > 0.0) {
            n_punct = qldpc_parity_bits_to_punct(N, K, qldpc_min_code_rate((float)ber, (float)target_eff));
            if (n_punct < 0) { printf("# ber %.4f: mother code rate already above the goal, nothing to puncture\n", ber); n_punct = 0; }
            if (n_punct > n_par) n_punct = n_par;
            printf("# ber %.4f: puncturing %d of %d parity bits -> rate %.4f, efficiency f = %.3f\n", ber, n_punct, n_par, (double)K / (N - n_punct),
                   ((double)(n_par - n_punct) / K) / (double)qldpc_binary_entropy((float)ber));
        }
        long fra = 0, be = 0, fe = 0, best_fe = -1, best_be = -1, n_pat = 0;
        int *best_pat = (search && n_punct > 0) ? (int *)malloc(sizeof(int) * (size_t)n_punct) : NULL;
        double t_dec = 0.0;
        while (fra < frames) {
            long pat_be = 0, pat_fe = 0;
            const int nb = frames - fra < batch ? (int)(frames - fra) : batch;
            for (long i = 0; i < (long)nb * K; i++) ref_bits[i] = (int)(rng_next() & 1);
            if ((rc = qldpc_encode(enc, ref_bits, enc_bits, nb))) return die("encode", rc);
            for (int f = 0; f < nb; f++)
                for (int v = 0; v < N; v++) {
                    const int x = enc_bits[(size_t)f * N + v];
                    if (is_info[v]) { const int y = x ^ (rng_unit() < ber); llr[(size_t)f * N + v] = y ? -L : L; }      /* BSC + demodulate */
                    else { const int y = x ^ (parity_ber > 0.0 && rng_unit() < parity_ber); llr[(size_t)f * N + v] = y ? -QLDPC_CONFIRMED_BIT_LLR : QLDPC_CONFIRMED_BIT_LLR; }   /* disclosed parity (possibly dirty) */
                }
            if (n_punct > 0) {      /* new random pattern per batch: partial Fisher-Yates over the parity positions */
                for (int i = 0; i < n_punct; i++) { const int j = i + (int)(rng_next() % (uint64_t)(n_par - i)); const int t = par_pos[i]; par_pos[i] = par_pos[j]; par_pos[j] = t; }
                for (int f = 0; f < nb; f++) for (int i = 0; i < n_punct; i++) llr[(size_t)f * N + par_pos[i]] = 0.0f;          /* main.cpp:359-362 */
            }
            const double t0 = now_s();
            if ((rc = qldpc_decode_siho(dec, llr, dec_bits, nb))) return die("decode_siho", rc);
            t_dec += now_s() - t0;
            qldpc_decoder_reset(dec);
            for (int f = 0; f < nb; f++) {
                long e = 0;
                for (int i = 0; i < K; i++) e += dec_bits[(size_t)f * K + i] != ref_bits[(size_t)f * K + i];
                be += e; fe += e > 0; pat_be += e; pat_fe += e > 0;
            }
            fra += nb;
            if (best_pat) {      /* the reference's random search (main.cpp:321-333): one shuffled pattern per simulation round, keep the best */
                printf("#   pattern %3ld: FE %ld / %d, BE %ld\n", n_pat, pat_fe, nb, pat_be);
                if (best_fe < 0 || pat_fe < best_fe || (pat_fe == best_fe && pat_be < best_be)) { best_fe = pat_fe; best_be = pat_be; memcpy(best_pat, par_pos, sizeof(int) * (size_t)n_punct); }
                n_pat++;
            }
        }
        if (best_pat) {
            printf("# ber %.4f: best of %ld patterns: FE %ld, BE %ld per %d frames\n", ber, n_pat, best_fe, best_be, batch);
            if (pattern_out) {
                FILE *fo = fopen(pattern_out, "w");
                if (!fo) { perror(pattern_out); return 1; }
                fprintf(fo, "# qldpc_sim puncture pattern: N %d K %d ber %.4f punctured %d FE %ld\n", N, K, ber, n_punct, best_fe);
                for (int i = 0; i < n_punct; i++) fprintf(fo, "%d\n", best_pat[i]);
                fclose(fo);
            }
            free(best_pat);
        }
        printf("  %8.4f | %8ld | %8ld | %8ld | %9.2e | %9.2e | %10.3f\n", ber, fra, be, fe, (double)be / ((double)fra * K), (double)fe / (double)fra,
               (double)fra * K / t_dec / 1e6);
        fflush(stdout);
    }
    qldpc_decoder_free(dec); qldpc_encoder_free(enc); qldpc_code_free(H);
    free(par_pos); free(pos); free(is_info); free(ref_bits); free(enc_bits); free(dec_bits); free(llr);
    return 0;
}
