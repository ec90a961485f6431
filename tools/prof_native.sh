#!/bin/bash
# rocprofv3 kernel trace of the compiled replay driver: tools/prof_native.sh <tag> [replay args...]  -> gpurun_out/trace_native_<tag>/
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/trace_native_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export LD_LIBRARY_PATH=$REPO/tiny-ram-halo2_amd:${LD_LIBRARY_PATH:-}
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT -o trace -- $REPO/examples/replay "$@" > $OUT/log.txt 2>&1
tail -1 $OUT/log.txt | cut -c1-400
