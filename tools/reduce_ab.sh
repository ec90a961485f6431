#!/bin/bash
# A/B of the bucket reduction: the slice form at several slice lengths against the two-level row / column form
# (TRH_REDUCE_2L=1), on a lone 2^20 Pallas MSM (BASELINE config 2) and on the k = 18 opening (18 rounds of batch-2 fixed-base MSMs).
#   bash tools/reduce_ab.sh  -> table on stdout (kept as profiles/r04_reduce_ab.txt)
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
echo "# lone 2^20 Pallas MSM (tools/msm_probe.py 20 pallas 0 0): phases in ms (reduce_ms = bucket reduction + window sum), wall per MSM"
for v in "TRH_REDUCE_2L=0" "TRH_REDUCE_2L=0 TRH_REDUCE_TPW=2048" "TRH_REDUCE_2L=0 TRH_REDUCE_TPW=8192" "TRH_REDUCE_2L=0 TRH_REDUCE_TPW=16384" "TRH_REDUCE_2L=0 TRH_REDUCE_TPW=32768" "TRH_REDUCE_2L=1"; do
  echo "$v: $(env $v python3 tools/msm_probe.py 20 pallas 0 0 2>/dev/null | tail -1)"
done
echo "# the same at 2^22 and 2^24"
for lg in 22 24; do for v in "TRH_REDUCE_2L=0" "TRH_REDUCE_2L=1"; do echo "2^$lg $v: $(env $v python3 tools/msm_probe.py $lg pallas 0 0 2>/dev/null | tail -1)"; done; done
echo "# k = 18 opening (tools/ipa_probe.py 18: second run), fixed-base tables"
for v in "TRH_REDUCE_2L=0" "TRH_REDUCE_2L=1"; do echo "$v: $(env $v python3 tools/ipa_probe.py 18 2>/dev/null | tail -1)"; done
