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

/** @brief subtype 9: parity message (payload = ceil(code_m/32) words, MSB-first, zero padded) */
typedef struct ERRC_LDPC_9 {
    EcPktHdr_Base base;
    unsigned int rateIndex;       /**< index into the shared rate table                    */
    unsigned int keyBits;         /**< workbits after helper_cleanupRevealedBits           */
    unsigned int codeK;           /**< info VNs of the code                                */
    unsigned int codeM;           /**< parity VNs = disclosed bits                         */
    unsigned int crc32;           /**< CRC-32 of the initiator's key                       */
} EcPktHdr_LdpcParity;

/** @brief subtype 10: verdict */
typedef struct ERRC_LDPC_10 {
    EcPktHdr_Base base;
    unsigned int decoded;         /**< 1: follower holds the initiator's key, PA message follows; 0: both sides fall back to cascade (or drop the block if ECD2_LDPC_FALLBACK=0) */
    unsigned int correctedBits;
    unsigned int iterations;
} EcPktHdr_LdpcVerdict;

/** @brief per-block LDPC working data (ProcessBlock.algorithmDataPtr) */
typedef struct ALGORITHM_LDPC_DATA {
    unsigned int rateIndex, codeK, codeM;
    int iterations;
} LdpcData;

extern const ALGORITHM_PKT_MNGR ALG_PKT_MNGR_LDPC_INITIATOR;
extern const ALGORITHM_PKT_MNGR ALG_PKT_MNGR_LDPC_FOLLOWER;
extern const ALGORITHM_DATA_MNGR ALG_DATA_MNGR_LDPC;

/** engine life cycle: call once from main() (device = HIP ordinal); ldpc_shutdown at exit */
int ldpc_init(int device);
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

#endif
