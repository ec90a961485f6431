#!/bin/bash
# NOTE: drives the TRH_EXP_* knobs of the experiment build (git 780c803); the library no longer has them -- kept as the record of how profiles/r06_overlap_*.txt were made
# EXPERIMENT (round 6): window-group pipeline of a lone 2^24 MSM -- does the sort run UNDER the accumulation?
#   tools/exp/overlap_ab.sh -> gpurun_out/overlap_ab/report.txt  (copied to profiles/r06_overlap_ab.txt)
# Timelines come from rocprofv3 --kernel-trace alone (counter passes serialise dispatches); instruction counts from separate --pmc passes.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/overlap_ab
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
P="python3 $REPO/tools/msm_probe.py 24 pallas 0 0"
trace() { tag=$1; shift; env "$@" rocprofv3 --kernel-trace -d $OUT/$tag -o t -- $P > $OUT/$tag.log 2>&1; }
trace v0 A=0
trace g4_lean_acc2 TRH_EXP_GROUPS=4
trace g4_regular_acc2 TRH_EXP_GROUPS=4 TRH_EXP_LEAN=0
trace g4_lean_acc3 TRH_EXP_GROUPS=4 TRH_EXP_ACC2=0
trace g4_regular_acc3 TRH_EXP_GROUPS=4 TRH_EXP_ACC2=0 TRH_EXP_LEAN=0 TRH_EXP_TAILQ4=0
pmc() { tag=$1; shift; env "$@" rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY -d $OUT/$tag -o p -- $P > $OUT/$tag.log 2>&1; }
pmc pmc_v0 A=0
pmc pmc_g4_lean TRH_EXP_GROUPS=4
python3 $REPO/tools/exp/overlap_report.py $OUT > $OUT/report.txt 2>&1
cat $OUT/report.txt
