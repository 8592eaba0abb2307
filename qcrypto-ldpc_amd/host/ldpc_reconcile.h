/*
 * ldpc_reconcile.h -- LDPC reconciliation handlers for the qcrypto `ecd2` daemon (plain C).
 *
 * Drop this pair into errorcorrection/subcomponents/ next to cascade_biconf.{c,h}.  It is written
 * against the daemon's own headers and fills the two arms the reference left as `return 81`:
 *   subcomponents/qber_estim.c:337-340  (QBER follower chose ALG_LDPC_*)   -> ldpc_prepareAsQberFollower()
 *   subcomponents/qber_estim.c:420-423  (QBER initiator told ALG_LDPC_*)   -> ldpc_prepareAsQberInitiator()
 * The arithmetic is libqldpc (include/qldpc.h): one parity packet replaces the cascade exchange.
 *
 * New packet subtypes continue the table of definitions/packets.h:46-57 (0..8 are taken):
 *   9  SUBTYPE_LDPC_PARITY   EC initiator -> EC follower : plan + CRC-32 + parity words
 *   10 SUBTYPE_LDPC_VERDICT  EC follower  -> EC initiator: decoded / failed
 * and both packet managers also carry subtype 8 (privAmp_receivePrivAmpMsg), because a manager's
 * subtypes must be consecutive (definitions/algorithms/packet_manager.h:41-46, ecd2.c:505-526).
 */
#ifndef ECD2_LDPC_RECONCILE
#define ECD2_LDPC_RECONCILE

/* `-L D<n>`: which of n devices takes the block that starts at `epoch` (in the daemon multi-GPU means replicas only --
 * blocks go round-robin to devices, nothing is exchanged between them).  Each side chooses for itself; the two need not agree. */
#define LDPC_MAX_DEVICES 8
static inline int ldpc_deviceOf(unsigned int epoch, int n_devices) { return n_devices > 1 ? (int)(epoch % (unsigned int)n_devices) : 0; }

#ifndef LDPC_DEVICE_CHOICE_ONLY      /* (a unit test compiles the function above without the daemon's headers) */

#include "definitions/algorithms/algorithms.h"   /* ProcessBlock, packet/data managers, ALGORITHM_DECISION */
#include "definitions/packets.h"
#include "qldpc.h"

#define SUBTYPE_LDPC_PARITY 9
#define SUBTYPE_LDPC_VERDICT 10
#define ALG_DATATYPE_LDPC ((ALGORITHM_DATATYPE)2)      /* next free value of data_manager.h:17-20 */

/* error codes appended to errormessage[] (ecd2.h:251-337 ends at 84) */
#define LDPC_ERR_ENGINE 85        /* "LDPC engine error (libqldpc)"                  */
#define LDPC_ERR_PKT_SIZE 86      /* "LDPC packet size mismatch"                     */
#define LDPC_ERR_DECODE_FAILED 87 /* "LDPC decoding failed, block dropped"           */
#define LDPC_ERR_RATE 88          /* "QBER too high for the LDPC rate table"         */

/** @brief subtype 9: parity message.  The payload of a block is ceil((codeM - nPunct)/32) words of disclosed parity bits,
 *  MSB-first, zero padded; it travels in fragCount packets of at most LDPC_MAX_PACKET_BYTES each (transferd forwards packets of up
 *  to 10 000 bytes, remotecrypto/transferd.h:139, transferd.c:850), every one with the full header and its slice of the words. */
typedef struct ERRC_LDPC_9 {
    EcPktHdr_Base base;
    unsigned int rateIndex;       /**< index into the shared rate table                    */
    unsigned int keyBits;         /**< workbits after helper_cleanupRevealedBits           */
    unsigned int codeK;           /**< info VNs of the (mother) code                       */
    unsigned int codeM;           /**< parity VNs of the code                              */
    unsigned int crc32;           /**< CRC-32 of the initiator's key                       */
    unsigned int nPunct;          /**< parity VNs punctured: codeM - nPunct bits disclosed */
    unsigned int fragIndex;       /**< 0 .. fragCount-1                                    */
    unsigned int fragCount;       /**< packets this block's parity is split into (<= 32)   */
    unsigned int fragWordOffset;  /**< first payload word of this packet in the block's parity words */
} EcPktHdr_LdpcParity;
#define LDPC_MAX_PACKET_BYTES 10000
#define LDPC_MAX_FRAGMENTS 32
#define LDPC_NO_PLAN 0xFFFFFFFFu  /**< rateIndex of a header without payload: "this side's rate table has no code for the block" -- the follower
                                       answers with a failed verdict and both sides continue with cascade */

/** @brief subtype 10: verdict */
typedef struct ERRC_LDPC_10 {
    EcPktHdr_Base base;
    unsigned int decoded;         /**< 1: follower holds the initiator's key, PA message follows; 0: both sides fall back to cascade (or drop the block if ECD2_LDPC_FALLBACK=0);
                                       2: not decoded, send the parity bits the plan withheld (second round: the same header with nPunct = 0), the follower waits for them */
    unsigned int correctedBits;
    unsigned int iterations;
} EcPktHdr_LdpcVerdict;

/** @brief per-block LDPC working data (ProcessBlock.algorithmDataPtr) */
typedef struct ALGORITHM_LDPC_DATA {
    unsigned int rateIndex, codeK, codeM;
    int iterations;
    /* EC follower: the parity words as their fragments arrive */
    qldpc_recon_msg msg;          /**< header of the first fragment seen (the others must repeat it) */
    unsigned int *parityWords;    /**< malloc2'ed on the first fragment, ceil((codeM - nPunct)/32) words */
    unsigned int fragsSeen;       /**< bit i = fragment i has arrived                                 */
    unsigned int fragCount;
    unsigned int fragWords;       /**< words every fragment but the last carries (fragment i starts at i * fragWords)  */
    unsigned int wordsSeen;       /**< payload words placed so far                                                      */
    int parityState;              /**< 0 awaiting fragments, 1 complete and queued / being decoded   */
    int round;                    /**< 0: first parity message; 1: the withheld bits were asked for (follower) / sent (initiator) */
    float planQber;               /**< EC initiator: the error rate the block was planned for         */
} LdpcData;

extern const ALGORITHM_PKT_MNGR ALG_PKT_MNGR_LDPC_INITIATOR;
extern const ALGORITHM_PKT_MNGR ALG_PKT_MNGR_LDPC_FOLLOWER;
extern const ALGORITHM_DATA_MNGR ALG_DATA_MNGR_LDPC;

/** engine life cycle: call once from main() (device = HIP ordinal); ldpc_shutdown at exit.  Builds every mother code, encoder and
 *  decoder (qldpc_recon_cfg.preload): nothing is constructed or allocated on the device per block afterwards. */
int ldpc_init(int device);
/** the daemon's `-L` option (free in ecd2.c:26's getopt string): `-L 1` = choose LDPC after QBER estimation (qber_estim.c:301),
 *  `-L b<n>` batch size of the batched ingest, `-L w<ms>` its wait, `-L g` privacy amplification on the GPU, `-L f0` no cascade
 *  fallback, `-L r0` no second round (the withheld parity bits after a failed decode), `-L p<bytes>` largest parity packet, `-L D<n>` blocks round-robin over n devices (ldpc_deviceOf), `-L G<0|1>` qldpc_recon_cfg.gap_profile (0 = gaps calibrated for the mother codes, 1 = round 2's wider gaps; both daemons alike); several may be given comma separated: `-L 1,b8,g`.  Returns 0 or an error code. */
int ldpc_parseOption(const char *optarg);
int ldpc_selected(void);          /**< 1 after `-L 1` (or ECD2_LDPC=1 in the environment) */
int ldpc_selectedFor(const ProcessBlock *pb);      /**< the per-block choice: selected AND the rate table covers the block's estimated QBER */
int ldpc_gpuPrivAmp(void);        /**< 1 after `-L g` (or ECD2_GPU_PA=1)                   */
int ldpc_deviceForBlock(const ProcessBlock *pb);   /**< the device (0 .. n-1 of `-L D<n>`) this block's decode / hash runs on */
void ldpc_shutdown(void);

/** batched ingest (ECD2_LDPC_BATCH=n): call ldpc_tick(receivedPacketLinkedList == NULL) once per main-loop iteration (ecd2.c, after
 *  the received-packet part) and let the select() time-out be the short one while ldpc_pending() != 0 */
int ldpc_tick(int receiveQueueEmpty);
int ldpc_pending(void);

/** hooks for qber_estim.c */
int ldpc_prepareAsQberFollower(ProcessBlock *processBlock, ALGORITHM_DECISION chosenAlgorithm, char *ackToSend, unsigned int ackLength);
int ldpc_prepareAsQberInitiator(ProcessBlock *processBlock, ALGORITHM_DECISION chosenAlgorithm);

/** EC initiator: build and queue the parity packet */
int ldpc_initiateAfterQber(ProcessBlock *processBlock);
/** packet handlers (PacketHandlerArray entries) */
int ldpc_receiveParity(ProcessBlock *processBlock, char *receivebuf);
int ldpc_receiveVerdict(ProcessBlock *processBlock, char *receivebuf);

#endif /* LDPC_DEVICE_CHOICE_ONLY */
#endif
