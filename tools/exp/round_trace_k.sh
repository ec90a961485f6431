#!/bin/bash
# Kernel timeline of one unfolded round of the opening at size k: tools/exp/round_trace_k.sh <k>
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp TRH_SELFTEST=0 TRH_IPA_FOLD=0
cd /tmp
rm -rf $REPO/gpurun_out/rt_k
rocprofv3 --kernel-trace -d $REPO/gpurun_out/rt_k -o t -- python3 $REPO/tools/ipa_probe.py $1 > /dev/null 2>&1
python3 $REPO/tools/round_timeline.py $(dirname $(find $REPO/gpurun_out/rt_k -name "*.db" | head -1)) ipa_round_front_kernel 6
