#!/bin/bash
# NOTE: drives the TRH_EXP_* knobs of the experiment build (git 780c803); the library no longer has them -- kept as the record of how profiles/r06_overlap_*.txt were made
# EXPERIMENT (round 6): window-group pipeline of a lone MSM -- sort / reduction of neighbouring groups beside the accumulation
cd ${GRAFT_REPO_ROOT:-.}
run() { echo "== $*"; env "$@" python3 tools/msm_probe.py $LN pallas 0 0 2>&1 | tail -1; }
LN=24
run A=0
run TRH_EXP_GROUPS=4 TRH_EXP_ACC2=0 TRH_EXP_LEAN=0 TRH_EXP_TAILQ4=0
run TRH_EXP_GROUPS=4
run TRH_EXP_GROUPS=4 TRH_EXP_LEAN=0
run TRH_EXP_GROUPS=4 TRH_EXP_ACC2=0
run TRH_EXP_GROUPS=4 TRH_EXP_TAILQ4=0
run TRH_EXP_GROUPS=4 TRH_EXP_PRIO=0
run TRH_EXP_GROUPS=2
run TRH_EXP_GROUPS=3
run TRH_EXP_GROUPS=5
run TRH_EXP_GROUPS=8
run TRH_EXP_GROUPS=2,6,11
run TRH_EXP_GROUPS=1,4,8,12
LN=22
run A=0
run TRH_EXP_GROUPS=2
run TRH_EXP_GROUPS=4
LN=20
run A=0
run TRH_EXP_GROUPS=2
run TRH_EXP_GROUPS=4
