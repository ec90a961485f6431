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
    python3 -m tiny_ram_halo2_amd.replay --word-bits 32 --columns witness --overlap 2>/dev/null | tail -1 > gpurun_out/replay_witness_$TAG.json
    LD_LIBRARY_PATH=tiny-ram-halo2_amd ./examples/replay --word-bits 32 --columns witness 2>/dev/null | tail -1 > gpurun_out/native_replay_witness_$TAG.json
    bash tools/pmc_msm.sh $TAG 24 > gpurun_out/msm_sq_counters_$TAG.txt 2>&1
    bash tools/pmc_ntt.sh $TAG > gpurun_out/ntt_sq_counters_$TAG.txt 2>&1
    python3 -m tiny_ram_halo2_amd.replay --word-bits 16 2>/dev/null | tail -1 > gpurun_out/replay_k10_$TAG.json
    LD_LIBRARY_PATH=tiny-ram-halo2_amd ./examples/replay --word-bits 32 2>/dev/null | tail -1 > gpurun_out/native_replay_$TAG.json
    [ -x tools/microbench ] && ./tools/microbench > gpurun_out/microbench_$TAG.txt 2>&1
    # round 3: the host-pointer (drop-in) path, the issue-rate question, the 2^20 MSM
    for m in dropin dropin-batched; do
        python3 -m tiny_ram_halo2_amd.replay --word-bits 32 --columns witness --mode $m 2>/dev/null | tail -1 > gpurun_out/replay_${m}_$TAG.json
        python3 -m tiny_ram_halo2_amd.replay --word-bits 16 --columns witness --mode $m 2>/dev/null | tail -1 > gpurun_out/replay_${m}_k10_$TAG.json
        LD_LIBRARY_PATH=tiny-ram-halo2_amd ./examples/replay --word-bits 32 --columns witness --mode $m 2>/dev/null | tail -1 > gpurun_out/native_replay_${m}_$TAG.json
    done
    python3 -m tiny_ram_halo2_amd.replay --word-bits 32 --columns witness --extended full 2>/dev/null | tail -1 > gpurun_out/replay_witness_full_domain_$TAG.json
    python3 tools/dropin_probe.py 24 22 2>/dev/null | tail -1 > gpurun_out/dropin_probe_$TAG.json
    [ -x tools/pcie_probe ] && ./tools/pcie_probe 512 > gpurun_out/pcie_probe_$TAG.txt 2>&1
    [ -x tools/issue_probe ] && ./tools/issue_probe > gpurun_out/issue_probe_$TAG.txt 2>&1
    bash tools/pmc_valu.sh $TAG 24 > gpurun_out/msm_valu_counters_$TAG.txt 2>&1
    bash tools/prof_cmd.sh msm20_$TAG tools/msm_probe.py 20 pallas 0 0 > gpurun_out/msm_2_20_kernel_stats_$TAG.txt 2>&1
    python3 tools/products_probe.py 2>/dev/null > gpurun_out/products_probe_$TAG.txt
    # round 4: stall / instruction-cache counters of the accumulation, the reduction A/B, the column-sharded per-column phase on two contexts
    # (Python mirror and compiled driver), a lone commitment per witness class with and without the sparse-column path, the bank probe
    bash tools/pmc_stall.sh $TAG 24 > gpurun_out/msm_stall_counters_$TAG.txt 2>&1
    python3 -m tiny_ram_halo2_amd.replay --word-bits 32 --columns witness --devices 0,0 2>/dev/null | tail -1 > gpurun_out/replay_sharded_$TAG.json
    LD_LIBRARY_PATH=tiny-ram-halo2_amd ./examples/replay --word-bits 32 --columns witness --devices 0,0 2>/dev/null | tail -1 > gpurun_out/native_replay_sharded_$TAG.json
    (python3 tools/lone_sparse_probe.py 2>/dev/null | tail -1; TRH_SPARSE=0 python3 tools/lone_sparse_probe.py 2>/dev/null | tail -1) > gpurun_out/lone_sparse_probe_$TAG.txt
    [ -x tools/bank_probe ] && ./tools/bank_probe > gpurun_out/bank_probe_$TAG.txt 2>&1
    TRH_SPARSE=0 python3 -m tiny_ram_halo2_amd.replay --word-bits 32 --columns witness --no-keygen 2>/dev/null | tail -1 > gpurun_out/replay_witness_nosparse_$TAG.json
    python3 tools/io_trace_probe.py 22 2>&1 | grep -E "^full|^zero-padded|k=18 shapes" > gpurun_out/io_shapes_$TAG.txt
    # round 6: PMC traffic at the sweep sizes with the build id beside it, the driver diff, the soak of the final build
    # (tools/pmc_sweep.sh $TAG runs as its own gpurun call: its databases and this step's together exceed what one call brings back)
    bash tools/driver_diff.sh $TAG > /dev/null 2>&1
    # round 6, second half: the opening with its generators collapsed -- timeline around the collapse + where the opening's time goes, and the opening by size
    # with the collapse (default) and without (ipa_fold = 0)
    bash tools/exp/fold_trace.sh > gpurun_out/ipa_opening_timeline_$TAG.txt 2>&1
    (for k in 10 12 13 14 15 16 17 18 19; do for f in 1 0; do echo -n "k $k ipa_fold $f: "; TRH_IPA_FOLD=$f python3 tools/ipa_probe.py $k 2>&1 | tail -1; done; done) > gpurun_out/ipa_opening_by_k_$TAG.txt 2>&1
    head -c 160 gpurun_out/bench_$TAG.json; echo
else
    python3 tools/isa_regs.py | head -1
    python3 tools/summarize_prof.py gpurun_out/prof_$TAG "$TAG" | tail -2
    [ -d gpurun_out/pmc_sweep_$TAG ] && python3 tools/summarize_sweep_pmc.py gpurun_out/pmc_sweep_$TAG "$TAG" | tail -3
    [ -s gpurun_out/driver_diff_$TAG.md ] && echo "(driver diff table: gpurun_out/driver_diff_$TAG.md -- merge by hand into profiles/${TAG}_driver_diff.md)"
    for f in bench replay replay_witness replay_k10 native_replay native_replay_witness; do cp gpurun_out/${f}_$TAG.json profiles/; done
    for f in msm_sq_counters ntt_sq_counters; do grep -v "^\[" gpurun_out/${f}_$TAG.txt > profiles/${TAG}_$f.txt; done
    [ -f gpurun_out/microbench_$TAG.txt ] && cp gpurun_out/microbench_$TAG.txt profiles/
    for m in dropin dropin-batched; do for f in replay_${m} replay_${m}_k10 native_replay_${m}; do [ -s gpurun_out/${f}_$TAG.json ] && cp gpurun_out/${f}_$TAG.json profiles/; done; done
    for f in replay_witness_full_domain dropin_probe; do [ -s gpurun_out/${f}_$TAG.json ] && cp gpurun_out/${f}_$TAG.json profiles/; done
    for f in pcie_probe issue_probe products_probe; do [ -s gpurun_out/${f}_$TAG.txt ] && cp gpurun_out/${f}_$TAG.txt profiles/; done
    [ -s gpurun_out/msm_valu_counters_$TAG.txt ] && grep -v "^\[" gpurun_out/msm_valu_counters_$TAG.txt > profiles/${TAG}_msm_valu_counters.txt
    [ -s gpurun_out/msm_2_20_kernel_stats_$TAG.txt ] && cp gpurun_out/msm_2_20_kernel_stats_$TAG.txt profiles/${TAG}_msm_2_20_kernel_stats.txt
    for f in replay_sharded native_replay_sharded replay_witness_nosparse; do [ -s gpurun_out/${f}_$TAG.json ] && cp gpurun_out/${f}_$TAG.json profiles/; done
    for f in lone_sparse_probe bank_probe io_shapes; do [ -s gpurun_out/${f}_$TAG.txt ] && cp gpurun_out/${f}_$TAG.txt profiles/${TAG}_$f.txt; done
    [ -s gpurun_out/msm_stall_counters_$TAG.txt ] && grep -v "^\[\|^tail:" gpurun_out/msm_stall_counters_$TAG.txt > profiles/${TAG}_msm_stall_counters_raw.txt
    for f in ipa_opening_timeline ipa_opening_by_k; do [ -s gpurun_out/${f}_$TAG.txt ] && cp gpurun_out/${f}_$TAG.txt profiles/${TAG}_$f.txt; done
    ls profiles
fi
