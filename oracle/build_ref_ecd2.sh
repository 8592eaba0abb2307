#!/bin/bash
# oracle/build_ref_ecd2.sh -- build the reference's ecd2 daemon from its own sources (gcc, no build system)
# into oracle/_ref/ (git-ignored, travels to the GPU box).  TEST INFRASTRUCTURE ONLY.
#
#   oracle/_ref/ecd2_cascade  pristine reference daemon (cascade_biconf): the integration oracle of SURVEY.md section 4
#   oracle/_ref/ecd2_ldpc     the same sources with the four maintainer edits of INTEGRATION.md section 2 applied to a
#                             scratch copy (the two `return 81` arms of subcomponents/qber_estim.c:337-340,420-423, the
#                             algorithm choice at :301 made switchable with ECD2_LDPC=1, four appended error messages, the
#                             ldpc_tick() call + ldpc_pending() time-out term in ecd2.c's main loop for batched ingest) and
#                             linked with qcrypto-ldpc_amd/host/ldpc_reconcile.c + libqldpc.so
# Nothing from /root/reference is copied into the repository; the scratch copy lives in a temp dir and is deleted.
set -euo pipefail
REF=${REF:-/root/reference}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/oracle/_ref
[ -d "$REF/errorcorrection" ] || { echo "no reference tree at $REF: nothing to build"; exit 0; }
mkdir -p "$OUT"
TMP=$(mktemp -d)
trap 'rm -rf "$TMP"' EXIT
cp -r "$REF/errorcorrection" "$REF/packetheaders" "$TMP/"
chmod -R u+w "$TMP"
cd "$TMP/errorcorrection"
SRCS="subcomponents/rnd.c subcomponents/debug.c subcomponents/helpers.c subcomponents/comms.c subcomponents/cascade_biconf.c subcomponents/priv_amp.c subcomponents/qber_estim.c subcomponents/processblock_mgmt.c definitions/algorithms/algorithms.c ecd2.c"
gcc -O2 -g -w -o "$OUT/ecd2_cascade" $SRCS -lm
# the reference's own PRNG (subcomponents/rnd.c) as a shared object: pins the oracle's LFSR restatement
gcc -O2 -w -shared -fPIC -o "$OUT/librefrnd.so" subcomponents/rnd.c

# ---- maintainer edits on the scratch copy ----
python3 - <<'PY'
import re
p = "subcomponents/qber_estim.c"
s = open(p).read()
s = s.replace('#include "qber_estim.h"', '#include "qber_estim.h"\n#include "ldpc_reconcile.h"\n#include <stdlib.h>', 1)
s = s.replace("chosenAlgorithm = ALG_CASCADE_CONTINUE_ROLES;",
              'chosenAlgorithm = getenv("ECD2_LDPC") ? ALG_LDPC_CONTINUE_ROLES : ALG_CASCADE_CONTINUE_ROLES;', 1)
arm = "    case ALG_LDPC_CONTINUE_ROLES:\n      return 81;\n    case ALG_LDPC_FLIP_ROLES:\n      return 81;\n"
assert s.count(arm) == 2, s.count(arm)
first = s.index(arm)
s = s[:first] + ("    case ALG_LDPC_CONTINUE_ROLES:\n    case ALG_LDPC_FLIP_ROLES:\n"
                 "      return ldpc_prepareAsQberFollower(processBlock, chosenAlgorithm, (char *)(bufferToSend), bufferLengthInBytes);\n") + s[first + len(arm):]
second = s.index(arm)
s = s[:second] + ("    case ALG_LDPC_CONTINUE_ROLES:\n    case ALG_LDPC_FLIP_ROLES:\n"
                  "      return ldpc_prepareAsQberInitiator(processBlock, (ALGORITHM_DECISION)in_head->algorithmEnum);\n") + s[second + len(arm):]
open(p, "w").write(s)
p = "subcomponents/priv_amp.c"
s = open(p).read()
loop = ("    for (i = 0; i < pb->finalKeyBits; i++) { /* go through all targetbits */\n"
        "      m = 0;                                 /* initial word */\n"
        "      for (j = 0; j < numwords; j++)\n"
        "        m ^= (pb->mainBufPtr[j] & rnd_getPrngValue2_32(&pb->rngState));\n"
        "      if (calcParity(m)) finalkey[wordIndex(i)] |= uint32AllZeroExceptAtN(i);\n"
        "    }\n")
assert s.count(loop) == 1
s = s.replace(loop, "    if (getenv(\"ECD2_GPU_PA\")) {\n"
                    "      if (qldpc_privamp(0, pb->mainBufPtr, pb->workbits, seed, pb->finalKeyBits, finalkey)) return 85;\n"
                    "    } else {\n" + loop + "    }\n", 1)
s = s.replace('#include "priv_amp.h"', '#include "priv_amp.h"\n#include "qldpc.h"\n#include <stdlib.h>', 1)
open(p, "w").write(s)
p = "ecd2.c"
s = open(p).read()
s = s.replace('#include "ecd2.h"', '#include "ecd2.h"\n#include "subcomponents/ldpc_reconcile.h"', 1)
tail = ("      free2(tmpRecvdPktNode);                            /* ...and pointer entry */\n"
        "    }\n")
assert s.count(tail) == 1
s = s.replace(tail, tail + "    { int ldpcErr = ldpc_tick(receivedPacketLinkedList == NULL);      /* batched ingest: decode what queued up */\n"
                           "      if (ldpcErr) { emsg(ldpcErr); if (arguments.runtimeErrorMode == END_ON_ERR) return -ldpcErr; } }\n", 1)
sel = "(cmdInput[0] || receivedPacketLinkedList) ? TENMILLISEC : HALFSECOND"
assert s.count(sel) == 1
s = s.replace(sel, "(cmdInput[0] || receivedPacketLinkedList || ldpc_pending()) ? TENMILLISEC : HALFSECOND", 1)
open(p, "w").write(s)
p = "ecd2.h"
s = open(p).read()
s = s.replace('    "Algorithm specific data ptr not null"\n};',
              '    "Algorithm specific data ptr not null",\n    "LDPC engine error (libqldpc)", /* 85 */\n    "LDPC packet size mismatch",\n'
              '    "LDPC decoding failed, block dropped",\n    "QBER too high for the LDPC rate table"\n};', 1)
open(p, "w").write(s)
PY
cp "$ROOT/qcrypto-ldpc_amd/host/ldpc_reconcile.c" "$ROOT/qcrypto-ldpc_amd/host/ldpc_reconcile.h" subcomponents/
gcc -O2 -g -w -I. -Isubcomponents -I"$ROOT/include" -o "$OUT/ecd2_ldpc" $SRCS subcomponents/ldpc_reconcile.c \
    -L"$ROOT/qcrypto-ldpc_amd" -lqldpc -Wl,-rpath,'$ORIGIN/../../qcrypto-ldpc_amd' -Wl,-rpath,/opt/rocm/lib -lm
# optional: the same LDPC daemon with AddressSanitizer on the C sources (host code only), for chasing memory errors: ECD2_ASAN=1
if [ -n "${ECD2_ASAN:-}" ]; then
  gcc -O1 -g -w -fsanitize=address -fno-omit-frame-pointer -I. -Isubcomponents -I"$ROOT/include" -o "$OUT/ecd2_ldpc_asan" $SRCS subcomponents/ldpc_reconcile.c \
      -L"$ROOT/qcrypto-ldpc_amd" -lqldpc -Wl,-rpath,'$ORIGIN/../../qcrypto-ldpc_amd' -Wl,-rpath,/opt/rocm/lib -lm
fi
echo "built $OUT/ecd2_cascade and $OUT/ecd2_ldpc"
