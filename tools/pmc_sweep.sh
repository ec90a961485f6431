#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the dominant kernels at every size of bench.py's sweep (no roofline with traffic None):
#   tools/pmc_sweep.sh <tag>   -> gpurun_out/pmc_sweep_<tag>/{msm20,msm22,msm26,ntt20,ntt24}_{fetch,write}/ ; then, back in the repo,
#   python3 tools/summarize_sweep_pmc.py gpurun_out/pmc_sweep_<tag> <tag>   -> profiles/<tag>_pmc_sweep.md + profiles/traffic.json
# One counter per pass, kernel trace only beside it.
set -u
TAG=${1:-r05}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_sweep_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export TRH_SELFTEST=0  # the self-test's own small launches (2^10 MSMs, 2^10 / 2^12 transforms) would be averaged into the per-kernel figures
cd /tmp
python3 -c "import sys; sys.path.insert(0, '$REPO'); from tiny_ram_halo2_amd import api; print(api.lib().trh_version().decode())" > $OUT/version.txt 2>/dev/null
for lg in 20 22 26; do
  for c in FETCH_SIZE WRITE_SIZE; do
    d=$OUT/msm${lg}_$( [ $c = FETCH_SIZE ] && echo fetch || echo write )
    rocprofv3 --kernel-trace --pmc $c -d $d -o pmc -- python3 $REPO/tools/msm_probe.py $lg pallas 0 0 > $d.log 2>&1
  done
done
for lg in 20 24; do
  for c in FETCH_SIZE WRITE_SIZE; do
    d=$OUT/ntt${lg}_$( [ $c = FETCH_SIZE ] && echo fetch || echo write )
    rocprofv3 --kernel-trace --pmc $c -d $d -o pmc -- python3 $REPO/tools/ntt_probe.py $lg 5 > $d.log 2>&1
  done
done
ls $OUT
