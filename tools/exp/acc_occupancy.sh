#!/bin/bash
# NOTE: drives the TRH_EXP_* knobs of the experiment build (git 780c803); the library no longer has them -- kept as the record of how profiles/r06_overlap_*.txt were made
# EXPERIMENT (round 6): the accumulation at 3 / 2 / 1 workgroups (= waves per SIMD) per CU, by dynamic LDS
cd ${GRAFT_REPO_ROOT:-.}
for lds in 0 56000 65000; do
  for ln in 24 20; do
    echo "== TRH_EXP_ACC_LDS=$lds log_n=$ln"
    TRH_EXP_ACC_LDS=$lds python3 tools/msm_probe.py $ln pallas 0 0 2>&1 | tail -1
  done
done
