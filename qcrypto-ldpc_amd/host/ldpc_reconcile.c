/*
 * ldpc_reconcile.c -- LDPC reconciliation handlers for the qcrypto `ecd2` daemon (plain C host code).
 * See ldpc_reconcile.h for where it hooks in.  Shape follows the sibling algorithm:
 * cascade_initiateAfterQber (subcomponents/cascade_biconf.c:427-476) for the initiator,
 * chooseEcAlgorithmAsQberFollower (subcomponents/qber_estim.c:293-345) for the role set-up, and the
 * PA hand-over at cascade_biconf.c:892,939.  Every bit of arithmetic is behind libqldpc's C ABI.
 */
#include "ldpc_reconcile.h"

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

static qldpc_recon *g_recon = NULL;
static int ldpc_batchSize(void);

int ldpc_init(int device)
{
    qldpc_recon_cfg cfg;
    if (g_recon) return 0;
    qldpc_recon_cfg_default(&cfg);
    cfg.device = device;
    cfg.max_blocks = ldpc_batchSize();
    if (qldpc_recon_create(&cfg, &g_recon) != QLDPC_OK) {
        fprintf(stderr, "ldpc_init: %s\n", qldpc_last_error());
        return LDPC_ERR_ENGINE;
    }
    return 0;
}

void ldpc_shutdown(void)
{
    qldpc_recon_free(g_recon);
    g_recon = NULL;
}

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
    const char *e = getenv("ECD2_LDPC_FALLBACK");
    return !(e && e[0] == '0');
}

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

/* the cascade follower's table (definitions/algorithms/algorithms.c:80-93) with subtype 4 wrapped */
static const PacketHandlerArray ALG_PKTHNDLRS_LDPC_FALLBACK_FOLLOWER = {
    ldpc_fallbackStartBinSearch,            /* subtype 4 */
    cascade_followerBob_processBinSearch,   /* subtype 5 */
    cascade_generateBiconfReply,            /* subtype 6 */
    cascade_receiveBiconfReply,             /* subtype 7 */
    privAmp_receivePrivAmpMsg               /* subtype 8 */
};
static const ALGORITHM_PKT_MNGR ALG_PKT_MNGR_LDPC_FALLBACK_FOLLOWER = { &ALG_PKTHNDLRS_LDPC_FALLBACK_FOLLOWER, SUBTYPE_CASCADE_PARITY_LIST, SUBTYPE_START_PRIV_AMP, False };

static int ldpc_fallBackToCascade(ProcessBlock *pb, PROCESSOR_ROLE role)
{
    const int spent = pb->leakageBits;
    int errorCode, i;
    if ((errorCode = pb->algorithmDataMngr->freeData(pb))) return errorCode;
    pb->processorRole = role;
    pb->algorithmPktMngr = (ALGORITHM_PKT_MNGR *)(role == PROC_ROLE_EC_INITIATOR ? &ALG_PKT_MNGR_CASCADE_INITIATOR : &ALG_PKT_MNGR_LDPC_FALLBACK_FOLLOWER);
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

int ldpc_initiateAfterQber(ProcessBlock *pb)
{
    LdpcData *ld = (LdpcData *)pb->algorithmDataPtr;
    qldpc_recon_msg msg;
    EcPktHdr_LdpcParity *h9;
    unsigned int parityWords, totalBytes;
    int rc, errorCode;

    rc = qldpc_recon_plan(g_recon, pb->workbits, pb->localError, &msg);
    if (rc == QLDPC_EUNSUPPORTED) return LDPC_ERR_RATE;
    if (rc != QLDPC_OK) { fprintf(stderr, "ldpc: %s\n", qldpc_last_error()); return LDPC_ERR_ENGINE; }
    parityWords = (msg.code_m + 31) / 32;
    totalBytes = sizeof(EcPktHdr_LdpcParity) + parityWords * WORD_SIZE;
    if ((errorCode = ldpc_createHeader((char **)&h9, SUBTYPE_LDPC_PARITY, totalBytes, pb))) return errorCode;

    rc = qldpc_recon_encode(g_recon, pb->mainBufPtr, pb->workbits, pb->localError, &msg, (uint32_t *)&h9[1], (int)parityWords);
    if (rc != QLDPC_OK) { fprintf(stderr, "ldpc: %s\n", qldpc_last_error()); free2(h9); return LDPC_ERR_ENGINE; }
    h9->rateIndex = msg.rate_index;
    h9->keyBits = msg.key_bits;
    h9->codeK = msg.code_k;
    h9->codeM = msg.code_m;
    h9->crc32 = msg.crc32;
    {   /* fault injection for tests of the fallback path: ECD2_LDPC_FAULT=n flips n disclosed parity bits */
        const char *fault = getenv("ECD2_LDPC_FAULT");
        int n = fault ? atoi(fault) : 0, i;
        for (i = 0; i < n && i < (int)msg.code_m; i++) {
            const int pos = i * 97 % (int)msg.code_m;
            ((uint32_t *)&h9[1])[pos / 32] ^= 1u << (31 - pos % 32);
        }
    }
    ld->rateIndex = msg.rate_index; ld->codeK = msg.code_k; ld->codeM = msg.code_m;

    pb->processingState = PSTATE_PERFORMED_PARITY;
    pb->leakageBits += (int)msg.code_m + 32;            /* disclosed parity bits + CRC */
    printf("ldpc: epoch %08x: sent parity, %d key bits, rate index %u, K %u, M %u\n", pb->startEpoch, pb->workbits, msg.rate_index, msg.code_k, msg.code_m);
    fflush(stdout);
    return comms_insertSendPacket((char *)h9, h9->base.totalLengthInBytes);
}

/* ---- EC follower ("Bob"): decode, verify, verdict, privacy amplification --------------------- */

static int ldpc_parseParity(ProcessBlock *pb, const char *receivebuf, qldpc_recon_msg *msg)
{
    const EcPktHdr_LdpcParity *in_head = (const EcPktHdr_LdpcParity *)receivebuf;
    const unsigned int parityWords = (in_head->codeM + 31) / 32;
    if (in_head->base.totalLengthInBytes != sizeof(EcPktHdr_LdpcParity) + parityWords * WORD_SIZE) return LDPC_ERR_PKT_SIZE;
    if ((int)in_head->keyBits != pb->workbits) return LDPC_ERR_PKT_SIZE;
    memset(msg, 0, sizeof(*msg));
    msg->rate_index = in_head->rateIndex;
    msg->key_bits = in_head->keyBits;
    msg->code_k = in_head->codeK;
    msg->code_m = in_head->codeM;
    msg->crc32 = in_head->crc32;
    return 0;
}

/* after the decode: verdict packet, then privacy amplification (or the fallback) */
static int ldpc_finishBlock(ProcessBlock *pb, const qldpc_recon_msg *msg, int decoded, int corrected, int iterations)
{
    LdpcData *ld = (LdpcData *)pb->algorithmDataPtr;
    EcPktHdr_LdpcVerdict *h10;
    const int leaked = (int)msg->code_m + 32;            /* disclosed parity bits + the CRC */
    int errorCode;

    if ((errorCode = ldpc_createHeader((char **)&h10, SUBTYPE_LDPC_VERDICT, sizeof(EcPktHdr_LdpcVerdict), pb))) return errorCode;
    h10->decoded = decoded ? 1 : 0;
    h10->correctedBits = (unsigned int)corrected;
    h10->iterations = (unsigned int)iterations;
    if ((errorCode = comms_insertSendPacket((char *)h10, h10->base.totalLengthInBytes))) return errorCode;

    if (!decoded) {
        /* no codeword found or CRC mismatch: mainBufPtr is untouched, the disclosed bits are spent */
        printf("ldpc: epoch %08x: no verified codeword after %d iterations\n", pb->startEpoch, iterations);
        pb->leakageBits += leaked;
        if (ldpc_fallbackEnabled()) return ldpc_fallBackToCascade(pb, PROC_ROLE_EC_FOLLOWER);
        pBlkMgmt_removeProcessBlk(pb->startEpoch);
        return (arguments.runtimeErrorMode == END_ON_ERR) ? LDPC_ERR_DECODE_FAILED : 0;
    }
    ld->rateIndex = msg->rate_index; ld->codeK = msg->code_k; ld->codeM = msg->code_m; ld->iterations = iterations;
    printf("ldpc: epoch %08x: decoded %d key bits in %d iterations, %d errors corrected, %d bits leaked\n", pb->startEpoch, pb->workbits, iterations, corrected, leaked);
    fflush(stdout);
    pb->correctedErrors = corrected;
    pb->leakageBits += leaked;
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
static struct { unsigned int epoch; char *packet; } g_queue[LDPC_BATCH_MAX];
static int g_queued = 0, g_batch = -1, g_wait_ms = 0;
static struct timespec g_first;

static int ldpc_batchSize(void)
{
    if (g_batch < 0) {
        const char *e = getenv("ECD2_LDPC_BATCH"), *w = getenv("ECD2_LDPC_BATCH_WAIT_MS");
        g_batch = e ? atoi(e) : 1;
        if (g_batch < 1) g_batch = 1;
        if (g_batch > LDPC_BATCH_MAX) g_batch = LDPC_BATCH_MAX;
        g_wait_ms = w ? atoi(w) : 0;
    }
    return g_batch;
}

int ldpc_pending(void) { return g_queued; }

static int ldpc_flush(void)
{
    uint32_t *keys[LDPC_BATCH_MAX];
    const uint32_t *pars[LDPC_BATCH_MAX];
    ProcessBlock *pbs[LDPC_BATCH_MAX];
    qldpc_recon_msg msgs[LDPC_BATCH_MAX];
    int bits[LDPC_BATCH_MAX], status[LDPC_BATCH_MAX], corrected[LDPC_BATCH_MAX], iterations[LDPC_BATCH_MAX];
    float qber[LDPC_BATCH_MAX];
    int n = 0, i, rc, errorCode = 0;

    for (i = 0; i < g_queued; i++) {
        ProcessBlock *pb = pBlkMgmt_getProcessBlk(g_queue[i].epoch);
        if (!pb || ldpc_parseParity(pb, g_queue[i].packet, &msgs[n])) continue;      /* block gone or packet inconsistent: drop the entry */
        pbs[n] = pb; keys[n] = pb->mainBufPtr; bits[n] = pb->workbits; qber[n] = pb->localError;
        pars[n] = (const uint32_t *)(g_queue[i].packet + sizeof(EcPktHdr_LdpcParity));
        n++;
    }
    if (n > 0) {
        rc = qldpc_recon_decode_blocks(g_recon, n, keys, bits, qber, msgs, pars, status, corrected, iterations);
        if (rc != QLDPC_OK) { fprintf(stderr, "ldpc: %s\n", qldpc_last_error()); errorCode = LDPC_ERR_ENGINE; }
        else {
            printf("ldpc: decoded a batch of %d blocks in one call\n", n);
            for (i = 0; i < n; i++) {
                const int e = ldpc_finishBlock(pbs[i], &msgs[i], status[i] == QLDPC_OK, corrected[i], iterations[i]);
                if (e && !errorCode) errorCode = e;
            }
        }
    }
    for (i = 0; i < g_queued; i++) { free2(g_queue[i].packet); g_queue[i].packet = NULL; }
    g_queued = 0;
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
    EcPktHdr_LdpcParity *in_head = (EcPktHdr_LdpcParity *)receivebuf;
    qldpc_recon_msg msg;
    int corrected = 0, leaked = 0, iterations = 0, rc, errorCode;

    if ((errorCode = ldpc_parseParity(pb, receivebuf, &msg))) return errorCode;
    if (ldpc_batchSize() > 1) {
        char *copy = malloc2(in_head->base.totalLengthInBytes);
        if (!copy) return 43;
        memcpy(copy, receivebuf, in_head->base.totalLengthInBytes);
        if (g_queued == 0) clock_gettime(CLOCK_MONOTONIC, &g_first);
        g_queue[g_queued].epoch = pb->startEpoch; g_queue[g_queued].packet = copy;
        g_queued++;
        return g_queued >= LDPC_BATCH_MAX ? ldpc_flush() : 0;      /* otherwise ldpc_tick() decides */
    }
    /* the handler owns receivebuf only until it returns (ecd2.c:546-548): decode straight from it */
    rc = qldpc_recon_decode(g_recon, pb->mainBufPtr, pb->workbits, pb->localError, &msg, (const uint32_t *)&in_head[1], &corrected, &leaked, &iterations);
    if (rc != QLDPC_OK && rc != QLDPC_EDECODE) { fprintf(stderr, "ldpc: %s\n", qldpc_last_error()); return LDPC_ERR_ENGINE; }
    return ldpc_finishBlock(pb, &msg, rc == QLDPC_OK, corrected, iterations);
}

/* ---- EC initiator: verdict ---------------------------------------------------------------------- */

int ldpc_receiveVerdict(ProcessBlock *pb, char *receivebuf)
{
    EcPktHdr_LdpcVerdict *in_head = (EcPktHdr_LdpcVerdict *)receivebuf;
    if (in_head->base.totalLengthInBytes != sizeof(EcPktHdr_LdpcVerdict)) return LDPC_ERR_PKT_SIZE;
    if (!in_head->decoded) {
        if (ldpc_fallbackEnabled()) return ldpc_fallBackToCascade(pb, PROC_ROLE_EC_INITIATOR);
        pBlkMgmt_removeProcessBlk(pb->startEpoch);
        return (arguments.runtimeErrorMode == END_ON_ERR) ? LDPC_ERR_DECODE_FAILED : 0;
    }
    pb->correctedErrors = (int)in_head->correctedBits;
    ((LdpcData *)pb->algorithmDataPtr)->iterations = (int)in_head->iterations;
    return 0;   /* message 8 (privAmp_receivePrivAmpMsg) follows and finishes the block */
}
