#!/bin/bash
# Everything profiles/ holds for one tag, in two steps:
#   on the GPU box (through gpurun):   bash tools/round_profile.sh collect <tag>
#       rocprofv3 kernel stats + the two PMC passes of bench.py (tools/profile.sh), the bench line, the k = 18 / k = 10 replay
#       lines of the Python and native drivers, the instruction-rate microbenchmark -> gpurun_out/
#   back in the repo:                  bash tools/round_profile.sh install <tag>
#       summaries (tools/summarize_prof.py -> profiles/<tag>_kernel_stats.md, <tag>_pmc.md, traffic.json) and the JSON lines
# bench.py reads roofline.traffic from profiles/traffic.json, so after `install` re-run `python bench.py` once through gpurun
# if the traffic figure changed and copy that line over profiles/bench_<tag>.json.
set -u
MODE=${1:?collect|install}; TAG=${2:?tag}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$REPO"
if [ "$MODE" = collect ]; then
    bash tools/profile.sh "$TAG" > gpurun_out/profile_$TAG.log 2>&1
    python3 bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
    python3 -m tiny_ram_halo2_amd.replay --word-bits 32 2>/dev/null | tail -1 > gpurun_out/replay_$TAG.json
    python3 -m tiny_ram_halo2_amd.replay --word-bits 32 --columns witness 2>/dev/null | tail -1 > gpurun_out/replay_witness_$TAG.json
    LD_LIBRARY_PATH=tiny-ram-halo2_amd ./examples/replay --word-bits 32 --columns witness 2>/dev/null | tail -1 > gpurun_out/native_replay_witness_$TAG.json
    bash tools/pmc_msm.sh $TAG 24 > gpurun_out/msm_sq_counters_$TAG.txt 2>&1
    bash tools/pmc_ntt.sh $TAG > gpurun_out/ntt_sq_counters_$TAG.txt 2>&1
    python3 -m tiny_ram_halo2_amd.replay --word-bits 16 2>/dev/null | tail -1 > gpurun_out/replay_k10_$TAG.json
    LD_LIBRARY_PATH=tiny-ram-halo2_amd ./examples/replay --word-bits 32 2>/dev/null | tail -1 > gpurun_out/native_replay_$TAG.json
    [ -x tools/microbench ] && ./tools/microbench > gpurun_out/microbench_$TAG.txt 2>&1
    head -c 160 gpurun_out/bench_$TAG.json; echo
else
    python3 tools/summarize_prof.py gpurun_out/prof_$TAG "$TAG" | tail -2
    for f in bench replay replay_witness replay_k10 native_replay native_replay_witness; do cp gpurun_out/${f}_$TAG.json profiles/; done
    for f in msm_sq_counters ntt_sq_counters; do grep -v "^\[" gpurun_out/${f}_$TAG.txt > profiles/${TAG}_$f.txt; done
    [ -f gpurun_out/microbench_$TAG.txt ] && cp gpurun_out/microbench_$TAG.txt profiles/
    ls profiles
fi
