/*
 * ldpc_reconcile.c -- LDPC reconciliation handlers for the qcrypto `ecd2` daemon (plain C host code).
 * See ldpc_reconcile.h for where it hooks in.  Shape follows the sibling algorithm:
 * cascade_initiateAfterQber (subcomponents/cascade_biconf.c:427-476) for the initiator,
 * chooseEcAlgorithmAsQberFollower (subcomponents/qber_estim.c:293-345) for the role set-up, and the
 * PA hand-over at cascade_biconf.c:892,939.  Every bit of arithmetic is behind libqldpc's C ABI.
 */
#include "ldpc_reconcile.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "subcomponents/cascade_biconf.h"
#include "subcomponents/comms.h"
#include "subcomponents/debug.h"
#include "subcomponents/helpers.h"
#include "subcomponents/priv_amp.h"
#include "subcomponents/processblock_mgmt.h"
#include "definitions/proc_state.h"

/* one engine per device (`-L D<n>`: blocks go round-robin over n devices by epoch, SURVEY.md section 8e "replicas only") */
static qldpc_recon *g_recons[LDPC_MAX_DEVICES];
static int g_devices = 1;
static int ldpc_batchSize(void);
static qldpc_recon *ldpc_engineFor(const ProcessBlock *pb) { return g_recons[ldpc_deviceOf(pb->startEpoch, g_devices)]; }

/* ---- options: the daemon's -L letter (ecd2.c:26), with the environment as the fall-back for unmodified command lines ---------- */
static int g_opt_select = -1, g_opt_gpu_pa = -1, g_opt_fallback = -1, g_opt_max_packet = -1, g_opt_margin = 0, g_opt_second = 1;
#ifdef LDPC_TEST_HOOKS
/* fault injection for the loopback tests (oracle/build_ref_ecd2.sh sets -DLDPC_TEST_HOOKS for the test binary only): a plugin built as
 * INTEGRATION.md section 2 describes does not parse these letters and has none of the code below them */
static int g_opt_fault = -1, g_opt_dup = 0, g_opt_badhdr = 0, g_opt_noplan = 0, g_opt_badfrag = 0;
#endif
static int g_batch = -1, g_wait_ms = -1, g_gap_profile = -1;

static int ldpc_envInt(const char *name, int dflt)
{
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}

int ldpc_parseOption(const char *optarg)
{
    const char *p = optarg;
    if (!p) return 1;
    while (*p) {
        char *end;
        switch (*p) {
        case '0': g_opt_select = 0; end = (char *)p + 1; break;
        case '1': g_opt_select = 1; end = (char *)p + 1; break;
        case 'g': g_opt_gpu_pa = 1; end = (char *)p + 1; break;
        case 'b': g_batch = (int)strtol(p + 1, &end, 10); if (end == p + 1 || g_batch < 1) return 1; break;
        case 'w': g_wait_ms = (int)strtol(p + 1, &end, 10); if (end == p + 1 || g_wait_ms < 0) return 1; break;
        case 'f': g_opt_fallback = (int)strtol(p + 1, &end, 10); if (end == p + 1) return 1; break;
        case 'p': g_opt_max_packet = (int)strtol(p + 1, &end, 10); if (end == p + 1 || g_opt_max_packet < 256) return 1; break;
#ifdef LDPC_TEST_HOOKS
        case 'x': g_opt_fault = (int)strtol(p + 1, &end, 10); if (end == p + 1) return 1; break;      /* flip n disclosed parity bits */
        case 'd': g_opt_dup = (int)strtol(p + 1, &end, 10); if (end == p + 1) return 1; break;        /* send every parity packet n more times */
        case 'y': g_opt_badhdr = (int)strtol(p + 1, &end, 10); if (end == p + 1) return 1; break;     /* claim another rate index in the header */
        case 'n': g_opt_noplan = (int)strtol(p + 1, &end, 10); if (end == p + 1) return 1; break;     /* the initiator's plan comes back "no code for this QBER" */
        case 'z': g_opt_badfrag = (int)strtol(p + 1, &end, 10); if (end == p + 1) return 1; break;    /* fragment 1 claims a word offset that is off by n */
#endif
        case 'G': g_gap_profile = (int)strtol(p + 1, &end, 10); if (end == p + 1 || g_gap_profile < 0 || g_gap_profile > 1) return 1; break;   /* qldpc_recon_cfg.gap_profile (both daemons alike) */
        case 'D': g_devices = (int)strtol(p + 1, &end, 10); if (end == p + 1 || g_devices < 1 || g_devices > LDPC_MAX_DEVICES) return 1; break;   /* blocks round-robin over n devices */
        case 'm': g_opt_margin = (int)strtol(p + 1, &end, 10); if (end == p + 1 || g_opt_margin < 0 || g_opt_margin > 100) return 1; break;   /* plan for qber + n/10 sigma of its estimate */
        case 'r': g_opt_second = (int)strtol(p + 1, &end, 10); if (end == p + 1) return 1; break;     /* r0: no second round, a failed decode goes straight to cascade */
        default: return 1;
        }
        p = end;
        if (*p == ',') p++;
        else if (*p) return 1;
    }
    return 0;
}

/* The error rate the initiator plans the code for.  localError comes from a sample of n = initialBits - workbits revealed bits
 * (qber_estim.c:203-238), so on short blocks it is off by a sigma = sqrt(q (1 - q) / n) that is not small against the distance the
 * plan keeps from capacity (a 1 000-bit sample at 3 %: +-0.54 %); -L m<n> plans for q + n/10 sigma.  The follower's LLRs keep its own
 * estimate: only the disclosed amount changes, and it travels in the header. */
static float ldpc_planQber(const ProcessBlock *pb)
{
    const float q = pb->localError;
    const int n = pb->initialBits - pb->workbits;
    if (g_opt_margin <= 0 || n <= 0 || !(q > 0.0f)) return q;
    return q + 0.1f * (float)g_opt_margin * sqrtf(q * (1.0f - q) / (float)n);
}

int ldpc_selected(void) { return g_opt_select >= 0 ? g_opt_select : (getenv("ECD2_LDPC") != NULL); }
/* The QBER follower's choice for ONE block (qber_estim.c:301): LDPC when it is selected and the rate table has a code for the estimated
 * error rate.  Beyond the lowest rate (0.5: about 8 % at the configured efficiency) the plan has no code; such a block goes to cascade
 * from the start instead of ending the daemon with LDPC_ERR_RATE -- unless it is too large for cascade's 16-bit indices. */
int ldpc_selectedFor(const ProcessBlock *pb)
{
    qldpc_recon_msg msg;
    int i, kept = 0;
    if (!ldpc_selected()) return 0;
    if (!g_recons[0] || pb->initialBits >= (1 << 16)) return 1;
    /* the length the initiator will plan on: workbits after helper_cleanupRevealedBits (helpers.c:31-69) = the bits of the block
     * whose marker is clear.  Planning on initialBits could pick another mother size, whose gap from capacity differs, and choose
     * LDPC for a block the initiator then cannot plan (it answers with a "no plan" header in that case, see ldpc_initiateAfterQber). */
    for (i = 0; i < pb->initialBits; i++)
        if (!(pb->testedBitsMarker[wordIndex(i)] & uint32AllZeroExceptAtN(i))) kept++;
    if (kept < 32) return 0;
    return qldpc_recon_plan(ldpc_engineFor(pb), kept, pb->localError, &msg) != QLDPC_EUNSUPPORTED;
}
int ldpc_gpuPrivAmp(void) { return g_opt_gpu_pa >= 0 ? g_opt_gpu_pa : (getenv("ECD2_GPU_PA") != NULL); }
static int ldpc_maxPacketBytes(void)
{
    int v = g_opt_max_packet >= 0 ? g_opt_max_packet : ldpc_envInt("ECD2_LDPC_MAX_PACKET", LDPC_MAX_PACKET_BYTES);
    if (v > LDPC_MAX_PACKET_BYTES) v = LDPC_MAX_PACKET_BYTES;
    if (v < (int)sizeof(EcPktHdr_LdpcParity) + 64) v = (int)sizeof(EcPktHdr_LdpcParity) + 64;
    return v;
}

int ldpc_init(int device)
{
    qldpc_recon_cfg cfg;
    int d;
    if (g_recons[0]) return 0;
    for (d = 0; d < g_devices; d++) {
        qldpc_recon_cfg_default(&cfg);
        cfg.device = device + d;
        cfg.preload = 1;                   /* every mother code / encoder / decoder now: no construction, no device allocation per block */
        cfg.max_blocks = ldpc_batchSize();
        if (g_gap_profile >= 0) cfg.gap_profile = g_gap_profile;
        if (qldpc_recon_create(&cfg, &g_recons[d]) != QLDPC_OK) {
            fprintf(stderr, "ldpc_init: device %d: %s\n", device + d, qldpc_last_error());
            ldpc_shutdown();
            return LDPC_ERR_ENGINE;
        }
    }
    printf("ldpc: engine ready, %ld code / encoder / decoder sets built, up to %d blocks per decode call, %d device(s)\n", qldpc_recon_entries_created(g_recons[0]), cfg.max_blocks, g_devices);
    fflush(stdout);
    return 0;
}

void ldpc_shutdown(void)
{
    int d;
    for (d = 0; d < LDPC_MAX_DEVICES; d++) { qldpc_recon_free(g_recons[d]); g_recons[d] = NULL; }
}

int ldpc_deviceForBlock(const ProcessBlock *pb) { return ldpc_deviceOf(pb->startEpoch, g_devices); }

/* ---- data manager (definitions/algorithms/data_manager.h:28-32) ---------------------------- */

static int initLdpcData(ProcessBlock *processBlock)
{
    if (processBlock->algorithmDataPtr) return 84;
    processBlock->algorithmDataPtr = malloc2(sizeof(LdpcData));
    if (!processBlock->algorithmDataPtr) return 34;
    memset(processBlock->algorithmDataPtr, 0, sizeof(LdpcData));
    return 0;
}

static int freeLdpcData(ProcessBlock *processBlock)
{
    LdpcData *ld = (LdpcData *)processBlock->algorithmDataPtr;
    if (ld && ld->parityWords) free2(ld->parityWords);
    free2(processBlock->algorithmDataPtr);
    processBlock->algorithmDataPtr = NULL;
    processBlock->algorithmDataMngr = NULL;
    return 0;
}

const ALGORITHM_DATA_MNGR ALG_DATA_MNGR_LDPC = {
    ALG_DATATYPE_LDPC,
    (const int (*)(ProcessBlock *))&initLdpcData,
    (const int (*)(ProcessBlock *))&freeLdpcData
};

/* ---- packet managers: subtypes 8..10 on both sides ------------------------------------------ */

static const PacketHandlerArray ALG_PKTHNDLRS_LDPC = {
    privAmp_receivePrivAmpMsg,   /* subtype 8  */
    ldpc_receiveParity,          /* subtype 9  */
    ldpc_receiveVerdict          /* subtype 10 */
};
const ALGORITHM_PKT_MNGR ALG_PKT_MNGR_LDPC_INITIATOR = { &ALG_PKTHNDLRS_LDPC, SUBTYPE_START_PRIV_AMP, SUBTYPE_LDPC_VERDICT, False };
const ALGORITHM_PKT_MNGR ALG_PKT_MNGR_LDPC_FOLLOWER = { &ALG_PKTHNDLRS_LDPC, SUBTYPE_START_PRIV_AMP, SUBTYPE_LDPC_VERDICT, False };

/* ---- helpers ---------------------------------------------------------------------------------- */

/* comms_createEcHeader (subcomponents/comms.c:150-187) only knows subtypes 0..8; build ours the same way */
static int ldpc_createHeader(char **buf, unsigned int subtype, unsigned int totalBytes, ProcessBlock *pb)
{
    EcPktHdr_Base *h;
    *buf = malloc2(totalBytes);
    if (!*buf) return 43;
    memset(*buf, 0, totalBytes);
    h = (EcPktHdr_Base *)*buf;
    h->tag = EC_PACKET_TAG;
    h->subtype = subtype;
    h->totalLengthInBytes = totalBytes;
    h->epoch = pb->startEpoch;
    h->numberOfEpochs = pb->numberOfEpochs;
    return 0;
}

static int ldpc_setup(ProcessBlock *pb, PROCESSOR_ROLE role)
{
    int errorCode;
    if ((errorCode = ldpc_init(0))) return errorCode;
    pb->processorRole = role;
    pb->algorithmPktMngr = (ALGORITHM_PKT_MNGR *)(role == PROC_ROLE_EC_INITIATOR ? &ALG_PKT_MNGR_LDPC_INITIATOR : &ALG_PKT_MNGR_LDPC_FOLLOWER);
    pb->algorithmDataMngr = (ALGORITHM_DATA_MNGR *)&ALG_DATA_MNGR_LDPC;
    if ((errorCode = pb->algorithmDataMngr->initData(pb))) return errorCode;
    /* compact out the bits revealed during QBER estimation: sets workbits, zeroes leakageBits
     * (subcomponents/helpers.c:31-69); both sides do it, so both see the same workbits */
    helper_cleanupRevealedBits(pb);
    return 0;
}

/* A block LDPC could not reconcile (no codeword within the iteration budget, or CRC mismatch) is handed to
 * the sibling algorithm instead of being dropped: same role, the cascade managers, and the set-up the
 * cascade arms of qber_estim.c:306-334 / :397-421 do.  The parity and CRC bits already disclosed stay in
 * leakageBits, so privacy amplification accounts for both attempts.  ECD2_LDPC_FALLBACK=0 restores "drop". */
static int ldpc_fallbackEnabled(void)
{
    return g_opt_fallback >= 0 ? g_opt_fallback != 0 : ldpc_envInt("ECD2_LDPC_FALLBACK", 1) != 0;
}
/* cascade keeps its permutations in unsigned short indices (definitions/processblock.h:126-127): a block above 2^16 - 1 bits, which
 * only the LDPC path can take (MAX_BITS_PER_PROCESSBLOCK raised for it), cannot be handed over */
static int ldpc_canFallBack(const ProcessBlock *pb) { return ldpc_fallbackEnabled() && pb->initialBits < (1 << 16); }

/* cascade begins with helper_cleanupRevealedBits once more (helper_prepPermutationWrapper, helpers.c:79-82), which
 * (a) would overwrite the already compacted key at the marked positions and (b) zeroes leakageBits.  (a): rewrite the
 * marker so that a second clean-up is the identity on mainBufPtr[0..workbits).  (b): the bits LDPC spent are carried
 * past that reset -- by the caller on the initiator side, by a wrapped subtype-4 handler on the follower side. */
static void ldpc_normaliseMarkers(ProcessBlock *pb)
{
    int i;
    for (i = 0; i < pb->initialBits; i++) {
        const unsigned int bm = uint32AllZeroExceptAtN(i);
        if (i < pb->workbits) pb->testedBitsMarker[wordIndex(i)] &= ~bm;
        else pb->testedBitsMarker[wordIndex(i)] |= bm;
    }
}

#define LDPC_FALLBACK_SLOTS 64
static struct { unsigned int epoch; int spentBits; int used; } g_fallback[LDPC_FALLBACK_SLOTS];

static int ldpc_fallbackStartBinSearch(ProcessBlock *pb, char *receivebuf)
{
    int i, spent = 0, errorCode;
    for (i = 0; i < LDPC_FALLBACK_SLOTS; i++)
        if (g_fallback[i].used && g_fallback[i].epoch == pb->startEpoch) { spent = g_fallback[i].spentBits; g_fallback[i].used = 0; break; }
    errorCode = cascade_startBinSearch(pb, receivebuf);
    pb->leakageBits += spent;
    return errorCode;
}

/* LDPC packets that are still on the wire when a block goes over to cascade (the later fragments of a parity message whose first
 * fragment was refused or malformed, a repeated verdict) must not end in ecd2.c:505-510's "subtype outside the manager's range"
 * (error 45): the fallback tables run on to subtype 10 and drop them */
static int ldpc_ignoreStraggler(ProcessBlock *pb, char *receivebuf)
{
    printf("ldpc: epoch %08x: LDPC packet (subtype %u) after the block went to cascade, ignored\n", pb->startEpoch, ((EcPktHdr_Base *)receivebuf)->subtype);
    return 0;
}

/* the cascade follower's table (definitions/algorithms/algorithms.c:80-93) with subtype 4 wrapped */
static const PacketHandlerArray ALG_PKTHNDLRS_LDPC_FALLBACK_FOLLOWER = {
    ldpc_fallbackStartBinSearch,            /* subtype 4 */
    cascade_followerBob_processBinSearch,   /* subtype 5 */
    cascade_generateBiconfReply,            /* subtype 6 */
    cascade_receiveBiconfReply,             /* subtype 7 */
    privAmp_receivePrivAmpMsg,              /* subtype 8 */
    ldpc_ignoreStraggler,                   /* subtype 9 */
    ldpc_ignoreStraggler                    /* subtype 10 */
};
static const ALGORITHM_PKT_MNGR ALG_PKT_MNGR_LDPC_FALLBACK_FOLLOWER = { &ALG_PKTHNDLRS_LDPC_FALLBACK_FOLLOWER, SUBTYPE_CASCADE_PARITY_LIST, SUBTYPE_LDPC_VERDICT, False };
/* the cascade initiator's table (algorithms.c:62-68), likewise */
static const PacketHandlerArray ALG_PKTHNDLRS_LDPC_FALLBACK_INITIATOR = {
    cascade_startBinSearch,                     /* subtype 4 */
    cascade_initiatorAlice_processBinSearch,    /* subtype 5 */
    cascade_generateBiconfReply,                /* subtype 6 */
    cascade_receiveBiconfReply,                 /* subtype 7 */
    privAmp_receivePrivAmpMsg,                  /* subtype 8 */
    ldpc_ignoreStraggler,                       /* subtype 9 */
    ldpc_ignoreStraggler                        /* subtype 10 */
};
static const ALGORITHM_PKT_MNGR ALG_PKT_MNGR_LDPC_FALLBACK_INITIATOR = { &ALG_PKTHNDLRS_LDPC_FALLBACK_INITIATOR, SUBTYPE_CASCADE_PARITY_LIST, SUBTYPE_LDPC_VERDICT, False };

static int ldpc_fallBackToCascade(ProcessBlock *pb, PROCESSOR_ROLE role)
{
    const int spent = pb->leakageBits;
    int errorCode, i;
    if ((errorCode = pb->algorithmDataMngr->freeData(pb))) return errorCode;
    pb->processorRole = role;
    pb->algorithmPktMngr = (ALGORITHM_PKT_MNGR *)(role == PROC_ROLE_EC_INITIATOR ? &ALG_PKT_MNGR_LDPC_FALLBACK_INITIATOR : &ALG_PKT_MNGR_LDPC_FALLBACK_FOLLOWER);
    pb->algorithmDataMngr = (ALGORITHM_DATA_MNGR *)&ALG_DATA_MNGR_CASCADE;
    if ((errorCode = pb->algorithmDataMngr->initData(pb))) return errorCode;
    ldpc_normaliseMarkers(pb);
    printf("ldpc: epoch %08x: falling back to cascade as EC %s, %d bits already leaked\n", pb->startEpoch,
           role == PROC_ROLE_EC_INITIATOR ? "initiator" : "follower", spent);
    fflush(stdout);
    if (role == PROC_ROLE_EC_INITIATOR) {
        errorCode = cascade_initiateAfterQber(pb);
        pb->leakageBits += spent;
        return errorCode;
    }
    for (i = 0; i < LDPC_FALLBACK_SLOTS; i++)      /* a free slot, or one whose block is gone (dropped before its parity list arrived) */
        if (!g_fallback[i].used || !pBlkMgmt_getProcessBlk(g_fallback[i].epoch)) break;
    if (i == LDPC_FALLBACK_SLOTS) return LDPC_ERR_ENGINE;
    g_fallback[i].epoch = pb->startEpoch; g_fallback[i].spentBits = spent; g_fallback[i].used = 1;
    cascade_calck0k1(pb);
    return 0;           /* await the initiator's parity list (subtype 4) */
}

/* ---- hooks for qber_estim.c ------------------------------------------------------------------- */

int ldpc_prepareAsQberFollower(ProcessBlock *pb, ALGORITHM_DECISION chosenAlgorithm, char *ackToSend, unsigned int ackLength)
{
    int errorCode;
    if (chosenAlgorithm == ALG_LDPC_CONTINUE_ROLES) {
        /* QBER follower stays EC follower: ACK, then wait for the parity packet */
        if ((errorCode = ldpc_setup(pb, PROC_ROLE_EC_FOLLOWER))) return errorCode;
        return comms_insertSendPacket(ackToSend, ackLength);
    }
    /* ALG_LDPC_FLIP_ROLES: ACK first, then act as EC initiator */
    if ((errorCode = ldpc_setup(pb, PROC_ROLE_EC_INITIATOR))) return errorCode;
    if ((errorCode = comms_insertSendPacket(ackToSend, ackLength))) return errorCode;
    return ldpc_initiateAfterQber(pb);
}

int ldpc_prepareAsQberInitiator(ProcessBlock *pb, ALGORITHM_DECISION chosenAlgorithm)
{
    int errorCode;
    if (chosenAlgorithm == ALG_LDPC_CONTINUE_ROLES) {
        if ((errorCode = ldpc_setup(pb, PROC_ROLE_EC_INITIATOR))) return errorCode;
        return ldpc_initiateAfterQber(pb);
    }
    return ldpc_setup(pb, PROC_ROLE_EC_FOLLOWER);      /* wait for the other side's parity packet */
}

/* ---- EC initiator ("Alice"): one parity packet ------------------------------------------------- */

/* the parity words of one message as subtype-9 packets: one if it fits under transferd's cap, else consecutive slices with the same header */
static int ldpc_sendParityPackets(ProcessBlock *pb, const qldpc_recon_msg *msg, const uint32_t *parity, unsigned int parityWords, unsigned int *fragCountOut)
{
    const unsigned int perPacket = ((unsigned int)ldpc_maxPacketBytes() - sizeof(EcPktHdr_LdpcParity)) / WORD_SIZE;
    unsigned int fragCount = (parityWords + perPacket - 1) / perPacket, f;
    int errorCode = 0;
    if (fragCount < 1) fragCount = 1;
    if (fragCount > LDPC_MAX_FRAGMENTS) return LDPC_ERR_PKT_SIZE;
    for (f = 0; f < fragCount && !errorCode; f++) {
        const unsigned int off = f * perPacket, words = (parityWords - off < perPacket) ? parityWords - off : perPacket;
        EcPktHdr_LdpcParity *h9;
        if ((errorCode = ldpc_createHeader((char **)&h9, SUBTYPE_LDPC_PARITY, sizeof(EcPktHdr_LdpcParity) + words * WORD_SIZE, pb))) break;
        h9->rateIndex = msg->rate_index; h9->keyBits = msg->key_bits; h9->codeK = msg->code_k; h9->codeM = msg->code_m;
        h9->crc32 = msg->crc32; h9->nPunct = msg->n_punct;
        h9->fragIndex = f; h9->fragCount = fragCount; h9->fragWordOffset = off;
        if (words) memcpy(&h9[1], parity + off, words * WORD_SIZE);
#ifdef LDPC_TEST_HOOKS
        if (msg->rate_index != LDPC_NO_PLAN) h9->rateIndex += (unsigned int)g_opt_badhdr;
        if (f == 1) h9->fragWordOffset += (unsigned int)g_opt_badfrag;
        {
            int d;
            for (d = 0; d < g_opt_dup && !errorCode; d++) {      /* tests only: the same packet again */
                char *copy = malloc2(h9->base.totalLengthInBytes);
                if (!copy) { errorCode = 43; break; }
                memcpy(copy, h9, h9->base.totalLengthInBytes);
                errorCode = comms_insertSendPacket(copy, h9->base.totalLengthInBytes);
            }
        }
#endif
        if (!errorCode) errorCode = comms_insertSendPacket((char *)h9, h9->base.totalLengthInBytes);
        else free2(h9);
    }
    *fragCountOut = fragCount;
    return errorCode;
}

int ldpc_initiateAfterQber(ProcessBlock *pb)
{
    LdpcData *ld = (LdpcData *)pb->algorithmDataPtr;
    qldpc_recon_msg msg;
    uint32_t *parity;
    unsigned int parityWords, fragCount = 0;
    int rc, errorCode = 0;
    float qplan;

    qplan = ldpc_planQber(pb);
    rc = qldpc_recon_plan(ldpc_engineFor(pb), pb->workbits, qplan, &msg);
    if (rc == QLDPC_EUNSUPPORTED && qplan > pb->localError) {      /* the margin pushed the plan off the rate table: plan for the estimate itself */
        qplan = pb->localError;
        rc = qldpc_recon_plan(ldpc_engineFor(pb), pb->workbits, qplan, &msg);
    }
#ifdef LDPC_TEST_HOOKS
    if (g_opt_noplan) rc = QLDPC_EUNSUPPORTED;
#endif
    if (rc == QLDPC_EUNSUPPORTED) {
        /* The follower chose LDPC for a block this side has no code for (its estimate sits at the edge of the rate table).  The main
         * loop drops a handler's return value (ecd2.c:524), so an error code alone would leave both daemons waiting: send a header that
         * says "no plan" instead.  The follower refuses it like any header its own table does not give, answers with a failed verdict,
         * and both sides go on with cascade (or drop the block where cascade cannot take it). */
        memset(&msg, 0, sizeof(msg));
        msg.rate_index = LDPC_NO_PLAN; msg.key_bits = (uint32_t)pb->workbits;
        printf("ldpc: epoch %08x: no code in the rate table for QBER %.4f, telling the follower (error %d would be: %s)\n", pb->startEpoch, (double)qplan, LDPC_ERR_RATE,
               "QBER too high for the LDPC rate table");
        fflush(stdout);
        if ((errorCode = ldpc_sendParityPackets(pb, &msg, NULL, 0, &fragCount))) return errorCode;
        ld->msg = msg; ld->planQber = qplan; ld->round = 1;      /* nothing withheld: no second round */
        pb->processingState = PSTATE_PERFORMED_PARITY;
        pb->leakageBits += qldpc_recon_leaked_bits(&msg);          /* what the follower charges for a refused header: the (unused) CRC */
        return 0;
    }
    if (rc != QLDPC_OK) { fprintf(stderr, "ldpc: %s\n", qldpc_last_error()); return LDPC_ERR_ENGINE; }
    parityWords = (unsigned int)qldpc_recon_parity_words(&msg);
    parity = (uint32_t *)malloc2(parityWords * WORD_SIZE + WORD_SIZE);
    if (!parity) return 43;
    rc = qldpc_recon_encode(ldpc_engineFor(pb), pb->mainBufPtr, pb->workbits, qplan, &msg, parity, (int)parityWords);
    if (rc != QLDPC_OK) { fprintf(stderr, "ldpc: %s\n", qldpc_last_error()); free2(parity); return LDPC_ERR_ENGINE; }
#ifdef LDPC_TEST_HOOKS
    {   /* fallback-path tests: -L x<n> / ECD2_LDPC_FAULT=n flips n disclosed parity bits (of the FIRST message) */
        const int n = g_opt_fault >= 0 ? g_opt_fault : ldpc_envInt("ECD2_LDPC_FAULT", 0), disclosed = (int)(msg.code_m - msg.n_punct);
        int i;
        for (i = 0; i < n && i < disclosed; i++) {
            const int pos = i * 97 % disclosed;
            parity[pos / 32] ^= 1u << (31 - pos % 32);
        }
    }
#endif
    errorCode = ldpc_sendParityPackets(pb, &msg, parity, parityWords, &fragCount);
    free2(parity);
    if (errorCode) return errorCode;
    ld->rateIndex = msg.rate_index; ld->codeK = msg.code_k; ld->codeM = msg.code_m;
    ld->msg = msg; ld->planQber = qplan; ld->round = 0;
    pb->processingState = PSTATE_PERFORMED_PARITY;
    pb->leakageBits += qldpc_recon_leaked_bits(&msg);            /* disclosed parity bits + CRC */
    printf("ldpc: epoch %08x: sent parity in %u packet(s), %d key bits, rate index %u, K %u, M %u, %u punctured, %d bits disclosed\n", pb->startEpoch, fragCount,
           pb->workbits, msg.rate_index, msg.code_k, msg.code_m, msg.n_punct, qldpc_recon_leaked_bits(&msg));
    fflush(stdout);
    return 0;
}

/* second round (verdict 2): the parity bits the plan withheld -- the whole parity of the same codeword, header with nPunct = 0.  The bits of
 * the first message are among them (evenly spaced puncturing), so the block's leak becomes codeM + 32. */
static int ldpc_sendWithheldParity(ProcessBlock *pb)
{
    LdpcData *ld = (LdpcData *)pb->algorithmDataPtr;
    qldpc_recon_msg msg = ld->msg;
    const int already = qldpc_recon_leaked_bits(&ld->msg);
    uint32_t *parity;
    unsigned int parityWords, fragCount = 0;
    int rc, errorCode;
    msg.n_punct = 0;
    parityWords = (unsigned int)qldpc_recon_parity_words(&msg);
    parity = (uint32_t *)malloc2(parityWords * WORD_SIZE + WORD_SIZE);
    if (!parity) return 43;
    rc = qldpc_recon_encode_planned(ldpc_engineFor(pb), pb->mainBufPtr, pb->workbits, &msg, parity, (int)parityWords);
    if (rc != QLDPC_OK) { fprintf(stderr, "ldpc: %s\n", qldpc_last_error()); free2(parity); return LDPC_ERR_ENGINE; }
    errorCode = ldpc_sendParityPackets(pb, &msg, parity, parityWords, &fragCount);
    free2(parity);
    if (errorCode) return errorCode;
    ld->round = 1;
    pb->leakageBits += qldpc_recon_leaked_bits(&msg) - already;
    printf("ldpc: epoch %08x: second round, sent the %u withheld parity bits too (%u packet(s), %d bits disclosed in all)\n", pb->startEpoch, ld->msg.n_punct, fragCount,
           qldpc_recon_leaked_bits(&msg));
    fflush(stdout);
    return 0;
}

/* ---- EC follower ("Bob"): decode, verify, verdict, privacy amplification --------------------- */

static int ldpc_finishBlock(ProcessBlock *pb, const qldpc_recon_msg *msg, int decoded, int corrected, int iterations);

/* one fragment of a block's parity: validate it against the block and against the fragments seen so far, copy its words.
 * returns 0 and *complete = 1 once every fragment is there; a packet for a block that is not waiting for parity (a duplicate after
 * completion, a packet for the wrong role) is ignored (*complete = 0).  LDPC_ERR_PKT_SIZE = a malformed packet for a block that IS
 * waiting: the caller answers it with a failed verdict (the main loop drops handler return values, ecd2.c:524). */
static int ldpc_acceptFragment(ProcessBlock *pb, const char *receivebuf, int *complete)
{
    const EcPktHdr_LdpcParity *in = (const EcPktHdr_LdpcParity *)receivebuf;
    LdpcData *ld = (LdpcData *)pb->algorithmDataPtr;
    unsigned int totalWords, words, per;
    *complete = 0;
    if (pb->processorRole != PROC_ROLE_EC_FOLLOWER || pb->algorithmDataMngr != (ALGORITHM_DATA_MNGR *)&ALG_DATA_MNGR_LDPC || !ld) return 0;   /* not ours to answer */
    if (ld->parityState != 0) return 0;                              /* already complete (queued or decoding): a repeated packet */
    if (ld->round == 1 && in->nPunct != 0) return 0;                 /* a late copy of the first message after the second round was asked for */
    if (in->base.totalLengthInBytes < sizeof(EcPktHdr_LdpcParity) || in->base.totalLengthInBytes > LDPC_MAX_PACKET_BYTES) return LDPC_ERR_PKT_SIZE;
    if ((int)in->keyBits != pb->workbits || in->nPunct > in->codeM || in->codeM > (1u << 26)) return LDPC_ERR_PKT_SIZE;
    if (in->fragCount < 1 || in->fragCount > LDPC_MAX_FRAGMENTS || in->fragIndex >= in->fragCount) return LDPC_ERR_PKT_SIZE;
    totalWords = (in->codeM - in->nPunct + 31) / 32;
    words = (in->base.totalLengthInBytes - sizeof(EcPktHdr_LdpcParity)) / WORD_SIZE;
    if (in->base.totalLengthInBytes != sizeof(EcPktHdr_LdpcParity) + words * WORD_SIZE) return LDPC_ERR_PKT_SIZE;
    if (in->fragWordOffset > totalWords || words > totalWords - in->fragWordOffset) return LDPC_ERR_PKT_SIZE;
    /* placement is tied to the fragment index: every fragment but the last carries `per` words and starts at fragIndex * per, the
     * last one ends the payload -- no gaps, no overlaps, whatever order they arrive in */
    if (in->fragIndex + 1 < in->fragCount) per = words;
    else per = in->fragCount > 1 ? in->fragWordOffset / (in->fragCount - 1) : totalWords;
    if (in->fragWordOffset != in->fragIndex * per || (in->fragIndex + 1 == in->fragCount && in->fragWordOffset + words != totalWords) ||
        (in->fragCount > 1 && per == 0)) return LDPC_ERR_PKT_SIZE;
    if (!ld->parityWords) {
        qldpc_recon_msg first;
        memset(&first, 0, sizeof(first));
        first.rate_index = in->rateIndex; first.key_bits = in->keyBits; first.code_k = in->codeK; first.code_m = in->codeM;
        first.crc32 = in->crc32; first.n_punct = in->nPunct;
        /* the dimensions must be what THIS side's rate table gives for the block before anything is allocated for the payload */
        if (qldpc_recon_check_header(ldpc_engineFor(pb), &first, pb->workbits) != QLDPC_OK) {
            /* answer it (the initiator waits for a verdict): failed, i.e. both sides fall back to cascade; what the packet says was disclosed
             * is charged to the leakage account.  The initiator's "no plan" header (LDPC_NO_PLAN) ends here too. */
            if (in->rateIndex == LDPC_NO_PLAN) printf("ldpc: epoch %08x: the initiator has no code for this block\n", pb->startEpoch);
            else printf("ldpc: epoch %08x: parity header refused (%s)\n", pb->startEpoch, qldpc_last_error());
            ld->parityState = 1; ld->round = 1;
            return ldpc_finishBlock(pb, &first, 0, 0, 0);
        }
        ld->parityWords = (unsigned int *)malloc2(totalWords * WORD_SIZE + WORD_SIZE);
        if (!ld->parityWords) return 43;
        memset(ld->parityWords, 0, totalWords * WORD_SIZE + WORD_SIZE);
        memset(&ld->msg, 0, sizeof(ld->msg));
        ld->msg.rate_index = in->rateIndex; ld->msg.key_bits = in->keyBits; ld->msg.code_k = in->codeK; ld->msg.code_m = in->codeM;
        ld->msg.crc32 = in->crc32; ld->msg.n_punct = in->nPunct;
        ld->fragCount = in->fragCount; ld->fragsSeen = 0; ld->fragWords = per; ld->wordsSeen = 0;
    } else if (ld->msg.rate_index != in->rateIndex || ld->msg.code_k != in->codeK || ld->msg.code_m != in->codeM || ld->msg.crc32 != in->crc32 ||
               ld->msg.n_punct != in->nPunct || ld->fragCount != in->fragCount || ld->fragWords != per) return LDPC_ERR_PKT_SIZE;      /* fragments must agree */
    if (ld->fragsSeen & (1u << in->fragIndex)) return 0;             /* this fragment again */
    memcpy(ld->parityWords + in->fragWordOffset, receivebuf + sizeof(EcPktHdr_LdpcParity), words * WORD_SIZE);
    ld->fragsSeen |= 1u << in->fragIndex;
    ld->wordsSeen += words;
    if (ld->fragsSeen == (ld->fragCount >= 32 ? 0xFFFFFFFFu : (1u << ld->fragCount) - 1u)) {
        if (ld->wordsSeen != totalWords) return LDPC_ERR_PKT_SIZE;   /* cannot happen given the placement rule; kept as the invariant it is */
        ld->parityState = 1; *complete = 1;
    }
    return 0;
}

/* after the decode: verdict packet, then privacy amplification (or the fallback) */
static int ldpc_finishBlock(ProcessBlock *pb, const qldpc_recon_msg *msg, int decoded, int corrected, int iterations)
{
    LdpcData *ld = (LdpcData *)pb->algorithmDataPtr;
    EcPktHdr_LdpcVerdict *h10;
    const int leaked = qldpc_recon_leaked_bits(msg);     /* disclosed parity bits + the CRC */
    int errorCode;

    if ((errorCode = ldpc_createHeader((char **)&h10, SUBTYPE_LDPC_VERDICT, sizeof(EcPktHdr_LdpcVerdict), pb))) return errorCode;
    h10->decoded = decoded ? 1 : ((g_opt_second && ld && ld->round == 0 && msg->n_punct > 0 && ld->parityWords) ? 2 : 0);
    h10->correctedBits = (unsigned int)corrected;
    h10->iterations = (unsigned int)iterations;
    if ((errorCode = comms_insertSendPacket((char *)h10, h10->base.totalLengthInBytes))) return errorCode;

    if (!decoded && g_opt_second && ld && ld->round == 0 && msg->n_punct > 0 && ld->parityWords) {
        /* first failure of a punctured plan: ask for the withheld parity bits instead of giving the block to cascade (the verdict went out
         * with decoded = 2 below).  mainBufPtr is untouched; the leak is charged when the block ends (the second message repeats the first's bits). */
        printf("ldpc: epoch %08x: no verified codeword after %d iterations, asking for the %u withheld parity bits\n", pb->startEpoch, iterations, msg->n_punct);
        fflush(stdout);
        free2(ld->parityWords);
        ld->parityWords = NULL; ld->fragsSeen = 0; ld->fragCount = 0; ld->parityState = 0; ld->round = 1;
        return 0;
    }
    if (!decoded) {
        /* no codeword found or CRC mismatch: mainBufPtr is untouched, the disclosed bits are spent */
        printf("ldpc: epoch %08x: no verified codeword after %d iterations\n", pb->startEpoch, iterations);
        pb->leakageBits += leaked;
        if (ldpc_canFallBack(pb)) return ldpc_fallBackToCascade(pb, PROC_ROLE_EC_FOLLOWER);
        pBlkMgmt_removeProcessBlk(pb->startEpoch);
        return (arguments.runtimeErrorMode == END_ON_ERR) ? LDPC_ERR_DECODE_FAILED : 0;
    }
    ld->rateIndex = msg->rate_index; ld->codeK = msg->code_k; ld->codeM = msg->code_m; ld->iterations = iterations;
    printf("ldpc: epoch %08x: decoded %d key bits in %d iterations, %d errors corrected, %d bits leaked (code sets built so far: %ld)\n", pb->startEpoch, pb->workbits,
           iterations, corrected, leaked, qldpc_recon_entries_created(ldpc_engineFor(pb)));
    fflush(stdout);
    /* Leakage account.  privAmp_doPrivAmp (priv_amp.c:86-90,166) ends with finalKeyBits = workbits - (leakageBits + sneakloss) +
     * correctedErrors: the last term is cascade's -- every error its binary search locates made one disclosed parity bit redundant.
     * An LDPC message holds no such redundancy (every disclosed parity bit and the CRC count in full), so the credit is charged
     * back here, on both sides alike (the initiator does the same in ldpc_receiveVerdict): the key ends workbits - leaked - sneakloss
     * bits long.  correctedErrors itself stays what it is: the sneakloss term is made of it. */
    pb->correctedErrors = corrected;
    pb->leakageBits += leaked + corrected;
    pb->processingState = PSTATE_PERFORMED_PARITY;
    /* same hand-over as cascade_biconf.c:892,939: send message 8 and do the PA locally */
    return privAmp_sendPrivAmpMsgAndPrivAmp(pb);
}

/*
 * Batched ingest: with ECD2_LDPC_BATCH=n (n > 1) a parity packet is only queued; the
 * main loop calls ldpc_tick() once per iteration and the queue is decoded in ONE qldpc_recon_decode_blocks call when it
 * holds n blocks, or when the receive list has drained and the oldest entry has waited ECD2_LDPC_BATCH_WAIT_MS (default 0).
 * The handler owns receivebuf only until it returns (ecd2.c:546-548), so the packet is copied.
 */
#define LDPC_BATCH_MAX 64
static unsigned int g_queue[LDPC_BATCH_MAX];      /* epochs of the blocks whose parity is complete; the words live in their LdpcData */
static int g_queued = 0;
static struct timespec g_first;

static int ldpc_batchSize(void)
{
    if (g_batch < 0) g_batch = ldpc_envInt("ECD2_LDPC_BATCH", 1);
    if (g_wait_ms < 0) g_wait_ms = ldpc_envInt("ECD2_LDPC_BATCH_WAIT_MS", 0);
    if (g_batch < 1) g_batch = 1;
    if (g_batch > LDPC_BATCH_MAX) g_batch = LDPC_BATCH_MAX;
    return g_batch;
}

int ldpc_pending(void) { return g_queued; }

/* decode everything queued: ONE qldpc_recon_decode_blocks call per device in use.  Every block is looked up by epoch again (it may
 * have been removed since it queued), validated and answered on its own: a header the engine refuses gets a failed verdict, it does
 * not fail the batch; an engine error fails every block of the call the same way (verdict 0 -> cascade or drop) instead of leaving
 * them queued for ever -- the main loop would not see a return code (ecd2.c:524). */
static int ldpc_flush(void)
{
    uint32_t *keys[LDPC_BATCH_MAX];
    const uint32_t *pars[LDPC_BATCH_MAX];
    unsigned int epochs[LDPC_BATCH_MAX], all[LDPC_BATCH_MAX];
    qldpc_recon_msg msgs[LDPC_BATCH_MAX];
    int bits[LDPC_BATCH_MAX], status[LDPC_BATCH_MAX], corrected[LDPC_BATCH_MAX], iterations[LDPC_BATCH_MAX];
    float qber[LDPC_BATCH_MAX];
    int n, i, d, rc, errorCode = 0, total = g_queued, decodedBlocks = 0;

    memcpy(all, g_queue, sizeof(unsigned int) * (size_t)g_queued);
    g_queued = 0;
    for (d = 0; d < g_devices; d++) {
        n = 0;
        for (i = 0; i < total; i++) {
            ProcessBlock *pb = pBlkMgmt_getProcessBlk(all[i]);
            LdpcData *ld;
            if (!pb || ldpc_deviceOf(all[i], g_devices) != d) continue;
            if (pb->algorithmDataMngr != (ALGORITHM_DATA_MNGR *)&ALG_DATA_MNGR_LDPC || !(ld = (LdpcData *)pb->algorithmDataPtr) || ld->parityState != 1) continue;
            epochs[n] = all[i]; keys[n] = pb->mainBufPtr; bits[n] = pb->workbits; qber[n] = pb->localError;
            msgs[n] = ld->msg; pars[n] = ld->parityWords;
            n++;
        }
        if (n == 0) continue;
        rc = qldpc_recon_decode_blocks(g_recons[d], n, keys, bits, qber, msgs, pars, status, corrected, iterations);
        if (rc != QLDPC_OK) {
            fprintf(stderr, "ldpc: device %d: %s\n", d, qldpc_last_error());
            printf("ldpc: engine error on a batch of %d blocks (%s): failed verdicts\n", n, qldpc_last_error());
            for (i = 0; i < n; i++) { status[i] = QLDPC_EDECODE; corrected[i] = 0; iterations[i] = 0; }
            if (!errorCode) errorCode = LDPC_ERR_ENGINE;
        }
        decodedBlocks += n;
        for (i = 0; i < n; i++) {
            ProcessBlock *pb = pBlkMgmt_getProcessBlk(epochs[i]);      /* finishing a block removes it (privacy amplification): never keep the pointer */
            int e;
            if (!pb) continue;
            if (status[i] == QLDPC_ESIZE || rc != QLDPC_OK) {
                if (rc == QLDPC_OK) printf("ldpc: epoch %08x: parity header refused (%s)\n", epochs[i], qldpc_last_error());
                if (pb->algorithmDataPtr) ((LdpcData *)pb->algorithmDataPtr)->round = 1;      /* nothing to ask a second round of: failed verdict */
            }
            e = ldpc_finishBlock(pb, &msgs[i], status[i] == QLDPC_OK, corrected[i], iterations[i]);
            if (e && !errorCode) errorCode = e;
        }
    }
    if (ldpc_batchSize() > 1 && decodedBlocks) { printf("ldpc: decoded a batch of %d blocks in one call\n", decodedBlocks); fflush(stdout); }
    return errorCode;
}

/* once per main-loop iteration (ecd2.c, after the received-packet part): receiveQueueEmpty = no packet waiting */
int ldpc_tick(int receiveQueueEmpty)
{
    if (g_queued == 0) return 0;
    if (g_queued < ldpc_batchSize()) {
        struct timespec now;
        if (!receiveQueueEmpty) return 0;
        clock_gettime(CLOCK_MONOTONIC, &now);
        if ((now.tv_sec - g_first.tv_sec) * 1000L + (now.tv_nsec - g_first.tv_nsec) / 1000000L < g_wait_ms) return 0;
    }
    return ldpc_flush();
}

int ldpc_receiveParity(ProcessBlock *pb, char *receivebuf)
{
    int complete = 0, errorCode, i;
    if ((errorCode = ldpc_acceptFragment(pb, receivebuf, &complete))) {
        LdpcData *ld = (LdpcData *)pb->algorithmDataPtr;
        if (errorCode == LDPC_ERR_PKT_SIZE && ld && ld->parityState == 0) {
            /* a malformed packet for a block that waits for its parity: say so and answer with a failed verdict (the initiator must not be
             * left waiting, and the main loop drops this return value).  What the block has been told was disclosed is charged: the plan
             * of the fragments accepted so far, else the header's own figures when they are sane. */
            const EcPktHdr_LdpcParity *in = (const EcPktHdr_LdpcParity *)receivebuf;
            qldpc_recon_msg m;
            memset(&m, 0, sizeof(m));
            if (ld->parityWords) m = ld->msg;
            else if (in->base.totalLengthInBytes >= sizeof(EcPktHdr_LdpcParity) && in->nPunct <= in->codeM && in->codeM <= (1u << 26)) { m.code_m = in->codeM; m.n_punct = in->nPunct; }
            printf("ldpc: epoch %08x: malformed parity packet (error %d), failed verdict\n", pb->startEpoch, errorCode);
            fflush(stdout);
            ld->parityState = 1; ld->round = 1;
            return ldpc_finishBlock(pb, &m, 0, 0, 0);
        }
        return errorCode;
    }
    if (!complete) return 0;                                         /* more fragments to come, or a packet that changes nothing */
    for (i = 0; i < g_queued; i++) if (g_queue[i] == pb->startEpoch) return 0;
    if (g_queued == 0) clock_gettime(CLOCK_MONOTONIC, &g_first);
    g_queue[g_queued++] = pb->startEpoch;
    /* one block at a time (the default): decode now, straight from the block's own buffers; with -L b<n> ldpc_tick() decides */
    return (ldpc_batchSize() <= 1 || g_queued >= LDPC_BATCH_MAX) ? ldpc_flush() : 0;
}

/* ---- EC initiator: verdict ---------------------------------------------------------------------- */

int ldpc_receiveVerdict(ProcessBlock *pb, char *receivebuf)
{
    EcPktHdr_LdpcVerdict *in_head = (EcPktHdr_LdpcVerdict *)receivebuf;
    LdpcData *ld = (LdpcData *)pb->algorithmDataPtr;
    /* both roles share one handler table: a verdict is only meaningful to an EC initiator that sent parity and holds LDPC data */
    if (pb->processorRole != PROC_ROLE_EC_INITIATOR || pb->algorithmDataMngr != (ALGORITHM_DATA_MNGR *)&ALG_DATA_MNGR_LDPC || !ld ||
        pb->processingState != PSTATE_PERFORMED_PARITY) {
        printf("ldpc: epoch %08x: verdict packet ignored (not an LDPC initiator waiting for one)\n", pb->startEpoch);
        return 0;
    }
    if (in_head->base.totalLengthInBytes != sizeof(EcPktHdr_LdpcVerdict)) { printf("ldpc: epoch %08x: verdict packet of %u bytes ignored\n", pb->startEpoch, in_head->base.totalLengthInBytes); return LDPC_ERR_PKT_SIZE; }
    if (in_head->decoded == 2) {      /* the follower asks for the withheld parity bits (once) */
        if (ld->round != 0 || ld->msg.n_punct == 0) { printf("ldpc: epoch %08x: second-round request ignored\n", pb->startEpoch); return LDPC_ERR_PKT_SIZE; }
        return ldpc_sendWithheldParity(pb);
    }
    if (!in_head->decoded) {
        if (ldpc_canFallBack(pb)) return ldpc_fallBackToCascade(pb, PROC_ROLE_EC_INITIATOR);
        pBlkMgmt_removeProcessBlk(pb->startEpoch);
        return (arguments.runtimeErrorMode == END_ON_ERR) ? LDPC_ERR_DECODE_FAILED : 0;
    }
    pb->correctedErrors = (int)in_head->correctedBits;
    pb->leakageBits += (int)in_head->correctedBits;      /* cancels privAmp_doPrivAmp's cascade-specific credit, as the follower does (ldpc_finishBlock) */
    ld->iterations = (int)in_head->iterations;
    return 0;   /* message 8 (privAmp_receivePrivAmpMsg) follows and finishes the block */
}
