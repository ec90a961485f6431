#!/bin/bash
# Kernel timeline of one k = 18 IPA round for this build and another: tools/ipa_round_trace.sh [other libtrh.so]
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OTHER=${1:-_ab/base/libtrh.so}
export TMPDIR=/tmp
export TRH_SELFTEST=0  # the self-test's own small launches (2^10 MSMs, 2^10 / 2^12 transforms) would be averaged into the per-kernel figures
cd /tmp
rm -rf $REPO/gpurun_out/ipa_tr_new $REPO/gpurun_out/ipa_tr_old
rocprofv3 --kernel-trace -d $REPO/gpurun_out/ipa_tr_new -o t -- python3 $REPO/tools/ipa_probe.py 18 > /dev/null 2>&1
TRH_LIB_PATH=$REPO/$OTHER rocprofv3 --kernel-trace -d $REPO/gpurun_out/ipa_tr_old -o t -- python3 $REPO/tools/ipa_probe.py 18 > /dev/null 2>&1
echo "=== this build"; python3 $REPO/tools/round_timeline.py $(dirname $(find $REPO/gpurun_out/ipa_tr_new -name "*.db" | head -1)) msm_recode_kernel 6
echo "=== $OTHER"; python3 $REPO/tools/round_timeline.py $(dirname $(find $REPO/gpurun_out/ipa_tr_old -name "*.db" | head -1)) msm_recode_kernel 6
