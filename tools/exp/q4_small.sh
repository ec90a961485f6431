#!/bin/bash
# NOTE: drives the TRH_EXP_* knobs of the experiment build (git 780c803); the library no longer has them -- kept as the record of how profiles/r06_overlap_*.txt were made
# EXPERIMENT (round 6): quad-lane bucket reduction for a lone 2^20 .. 2^22 MSM (16 bucket sets)
cd ${GRAFT_REPO_ROOT:-.}
run() { echo "== LN=$LN $*"; env "$@" python3 tools/msm_probe.py $LN pallas 0 0 2>&1 | tail -1; }
for LN in 20 21 22 18; do
run A=0
run TRH_EXP_Q4_SETS=16 TRH_EXP_Q4_LANES_LOG=17
run TRH_EXP_Q4_SETS=16 TRH_EXP_Q4_LANES_LOG=18
run TRH_EXP_Q4_SETS=32 TRH_EXP_Q4_LANES_LOG=19
done
