#!/bin/bash
# Collects the rocprofv3 evidence for the headline bench on the GPU box.
#   tools/profile.sh <tag>     -> gpurun_out/prof_<tag>/{stats,pmc_fetch,pmc_write}
# Kernel trace/stats and PMC counters are collected in separate runs (never combined).
set -u
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export TRH_SELFTEST=0  # the self-test's own small launches (2^10 MSMs, 2^10 / 2^12 transforms) would be averaged into the per-kernel figures
cd /tmp
# the library build the counters are collected on (bench.py flags a traffic figure whose build is not the running library's)
python3 -c "import sys; sys.path.insert(0, '$REPO'); from tiny_ram_halo2_amd import api; print(api.lib().trh_version().decode())" > $OUT/version.txt 2>/dev/null
ARGS="--gpus 1 --steps 8 --warmup 2 --no-cpu-baseline --no-check --no-sweep"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o trace -- python3 $REPO/bench.py $ARGS > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- python3 $REPO/bench.py $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- python3 $REPO/bench.py $ARGS > $OUT/pmc_write.log 2>&1
find $OUT -name "*.csv" | head -20
# the witness-shaped k = 18 replay: the chunked-sort / heavy-bucket paths of the MSM, the batched lookups, the multiopen folds
rocprofv3 --kernel-trace --stats -d $OUT/stats_witness -o trace -- python3 $REPO/tools/replay_probe.py 32 witness > $OUT/stats_witness.log 2>&1
