#!/bin/bash
# oracle/build_ref_ecd2.sh -- build the reference's ecd2 daemon from its own sources (gcc, no build system)
# into oracle/_ref/ (git-ignored, travels to the GPU box).  TEST INFRASTRUCTURE ONLY.
#
#   oracle/_ref/ecd2_cascade  pristine reference daemon (cascade_biconf): the integration oracle of SURVEY.md section 4
#   Both test daemons are compiled with -DFIXED_RNG_SEED=<n>, the reference's own switch (subcomponents/rnd.c:145-166, debug.h:62): every
#   seed the daemon would read from /dev/urandom (QBER sample positions, cascade permutations, the PA hash) is that constant, so a
#   loopback run is reproducible bit for bit and the tests assert exact key lengths.  ecd2_ldpc additionally gets -DLDPC_TEST_HOOKS
#   (the fault-injection items of -L: x, d, y).  oracle/_ref/ecd2_ldpc_urandom is the plugin as a maintainer builds it -- seeds from
#   /dev/urandom, no hooks -- for the soak runs of tests/ecd2_loop.py only.
#   oracle/_ref/ecd2_ldpc     the same sources with the four maintainer edits of INTEGRATION.md section 2 applied to a
#                             scratch copy (the two `return 81` arms of subcomponents/qber_estim.c:337-340,420-423, the
#                             algorithm choice at :301 taken from the new `-L` option (ldpc_selected(); ECD2_LDPC=1 still works),
#                             `L:` added to ecd2.c:26's getopt string, ldpc_init() before the main loop when LDPC is selected, four
#                             appended error messages, the ldpc_tick() call + ldpc_pending() time-out term in ecd2.c's main loop
#                             for batched ingest, MAX_BITS_PER_PROCESSBLOCK / TEMP_ARRAY_SIZE raised to 2^18 bits for the LDPC path) and
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
SEED=${ECD2_FIXED_SEED:-0x5eed1234}
gcc -O2 -g -w -DFIXED_RNG_SEED=$SEED -o "$OUT/ecd2_cascade" $SRCS -lm
# the reference's own PRNG (subcomponents/rnd.c) as a shared object: pins the oracle's LFSR restatement
gcc -O2 -w -shared -fPIC -o "$OUT/librefrnd.so" subcomponents/rnd.c

# ---- maintainer edits on the scratch copy ----
python3 - <<'PY'
import re
p = "subcomponents/qber_estim.c"
s = open(p).read()
s = s.replace('#include "qber_estim.h"', '#include "qber_estim.h"\n#include "ldpc_reconcile.h"\n#include <stdlib.h>', 1)
s = s.replace("chosenAlgorithm = ALG_CASCADE_CONTINUE_ROLES;",
              'chosenAlgorithm = ldpc_selectedFor(processBlock) ? ALG_LDPC_CONTINUE_ROLES : ALG_CASCADE_CONTINUE_ROLES;', 1)
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
s = s.replace(loop, "    if (ldpc_gpuPrivAmp()) {\n"
                    "      if (qldpc_privamp(ldpc_deviceForBlock(pb), pb->mainBufPtr, pb->workbits, seed, pb->finalKeyBits, finalkey)) return 85;\n"
                    "    } else {\n" + loop + "    }\n", 1)
s = s.replace('#include "priv_amp.h"', '#include "priv_amp.h"\n#include "ldpc_reconcile.h"\n#include <stdlib.h>', 1)
open(p, "w").write(s)
p = "ecd2.c"
s = open(p).read()
s = s.replace('#include "ecd2.h"', '#include "ecd2.h"\n#include "subcomponents/ldpc_reconcile.h"', 1)
tail = ("      free2(tmpRecvdPktNode);                            /* ...and pointer entry */\n"
        "    }\n")
assert s.count(tail) == 1
s = s.replace(tail, tail + "    { int ldpcErr = ldpc_tick(receivedPacketLinkedList == NULL);      /* batched ingest: decode what queued up */\n"
                           "      if (ldpcErr) { emsg(ldpcErr); if (arguments.runtimeErrorMode == END_ON_ERR) return -ldpcErr; } }\n", 1)
# the -L option: one more letter in the getopt string and its case, and the engine (all mother codes) brought up before the main loop
opts = '"c:s:r:d:f:l:q:Q:e:E:kJ:T:V:Ipb:B:i"'
assert s.count(opts) == 1
s = s.replace(opts, '"c:s:r:d:f:l:q:Q:e:E:kJ:T:V:Ipb:B:iL:"', 1)
case_v = "      case 'V': /* verbosity parameter */"
assert s.count(case_v) == 1
s = s.replace(case_v, "      case 'L': /* LDPC reconciliation options (subcomponents/ldpc_reconcile.h) */\n"
                      "        if (ldpc_parseOption(optarg)) return -emsg(1);\n        break;\n" + case_v, 1)
loop_head = "  // Main loop\n"
assert s.count(loop_head) == 1
s = s.replace(loop_head, "  if (ldpc_selected()) { int e_ = ldpc_init(0); if (e_) return -emsg(e_); }\n" + loop_head, 1)
sel = "(cmdInput[0] || receivedPacketLinkedList) ? TENMILLISEC : HALFSECOND"
assert s.count(sel) == 1
s = s.replace(sel, "(cmdInput[0] || receivedPacketLinkedList || ldpc_pending()) ? TENMILLISEC : HALFSECOND", 1)
open(p, "w").write(s)
# blocks above 2^16 bits for the LDPC path (processblock_mgmt.c:94-95 checks against this; cascade's unsigned short indices keep their
# limit: ldpc_reconcile.c never hands such a block to cascade)
p = "definitions/defaultdefinitions.h"
s = open(p).read()
assert s.count("#define TEMP_ARRAY_SIZE (1 << 11)") == 1 and s.count("#define MAX_BITS_PER_PROCESSBLOCK (1 << 16)") == 1
s = s.replace("#define TEMP_ARRAY_SIZE (1 << 11)", "#define TEMP_ARRAY_SIZE (1 << 13)", 1).replace("#define MAX_BITS_PER_PROCESSBLOCK (1 << 16)", "#define MAX_BITS_PER_PROCESSBLOCK (1 << 18)", 1)
open(p, "w").write(s)
p = "ecd2.h"
s = open(p).read()
s = s.replace('    "Algorithm specific data ptr not null"\n};',
              '    "Algorithm specific data ptr not null",\n    "LDPC engine error (libqldpc)", /* 85 */\n    "LDPC packet size mismatch",\n'
              '    "LDPC decoding failed, block dropped",\n    "QBER too high for the LDPC rate table"\n};', 1)
open(p, "w").write(s)
PY
cp "$ROOT/qcrypto-ldpc_amd/host/ldpc_reconcile.c" "$ROOT/qcrypto-ldpc_amd/host/ldpc_reconcile.h" subcomponents/
gcc -O2 -g -w -DFIXED_RNG_SEED=$SEED -DLDPC_TEST_HOOKS -I. -Isubcomponents -I"$ROOT/include" -o "$OUT/ecd2_ldpc" $SRCS subcomponents/ldpc_reconcile.c \
    -L"$ROOT/qcrypto-ldpc_amd" -lqldpc -Wl,-rpath,'$ORIGIN/../../qcrypto-ldpc_amd' -Wl,-rpath,/opt/rocm/lib -lm
gcc -O2 -g -w -I. -Isubcomponents -I"$ROOT/include" -o "$OUT/ecd2_ldpc_urandom" $SRCS subcomponents/ldpc_reconcile.c \
    -L"$ROOT/qcrypto-ldpc_amd" -lqldpc -Wl,-rpath,'$ORIGIN/../../qcrypto-ldpc_amd' -Wl,-rpath,/opt/rocm/lib -lm
# optional: the same LDPC daemon with AddressSanitizer on the C sources (host code only), for chasing memory errors: ECD2_ASAN=1
if [ -n "${ECD2_ASAN:-}" ]; then
  gcc -O1 -g -w -DFIXED_RNG_SEED=$SEED -DLDPC_TEST_HOOKS -fsanitize=address -fno-omit-frame-pointer -I. -Isubcomponents -I"$ROOT/include" -o "$OUT/ecd2_ldpc_asan" $SRCS subcomponents/ldpc_reconcile.c \
      -L"$ROOT/qcrypto-ldpc_amd" -lqldpc -Wl,-rpath,'$ORIGIN/../../qcrypto-ldpc_amd' -Wl,-rpath,/opt/rocm/lib -lm
fi
echo "built $OUT/ecd2_cascade, $OUT/ecd2_ldpc (fixed seed $SEED, test hooks) and $OUT/ecd2_ldpc_urandom"
