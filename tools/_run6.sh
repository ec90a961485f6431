TRH_SPARSE=0 bash tools/prof_cmd.sh nosparse_r04b tools/replay_probe.py 32 witness > /dev/null 2>&1
python tools/commit_spans.py gpurun_out/trace_nosparse_r04b
for t in 17 18; do TRH_ADAPTIVE_TARGET_LOG=$t bash tools/prof_cmd.sh sparse_t${t}_r04b tools/replay_probe.py 32 witness > /dev/null 2>&1; echo target_log $t; python tools/commit_spans.py gpurun_out/trace_sparse_t${t}_r04b; done
