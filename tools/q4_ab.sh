#!/bin/bash
# A/B of the quad-lane bucket reduction (msm_reduce_q4_kernel, TRH_REDUCE_Q4=1) against the one-thread-per-slice kernels (=0), same box:
#   tools/q4_ab.sh [out-file]
OUT=${1:-gpurun_out/q4_ab.txt}
mkdir -p "$(dirname "$OUT")"
run() {
  python3 tools/ipa_probe.py 18 2>/dev/null | tail -1
  python3 tools/lone_sparse_probe.py 2>/dev/null | tail -1
  python3 tools/msm_probe.py 20 pallas 0 0 2>/dev/null | tail -1
  python3 tools/msm_probe.py 24 pallas 0 0 2>/dev/null | tail -1
}
{
  echo "# $(date -u)"
  for rep in 1 2; do
    echo "## TRH_REDUCE_Q4=0 (run $rep)"; TRH_REDUCE_Q4=0 run
    for lg in 15 16 17; do echo "## TRH_REDUCE_Q4=1 TRH_REDUCE_Q4_LANES_LOG=$lg (run $rep)"; TRH_REDUCE_Q4_LANES_LOG=$lg run; done
  done
} 2>&1 | tee "$OUT"
