#!/bin/bash
# A/B of the persistent software-pipelined NTT pass (ntt_passp_kernel, TRH_NTT_PIPE=1) against ntt_passy_kernel (TRH_NTT_PIPE=0) on one box:
#   tools/ntt_pipe_ab.sh [out-file]        (optional: extra libtrh builds under _ab/*/libtrh.so are timed as well)
OUT=${1:-gpurun_out/ntt_pipe_ab.txt}
mkdir -p "$(dirname "$OUT")"
{
  echo "# $(date -u) $(python3 -c 'import sys; sys.path.insert(0,"."); from tiny_ram_halo2_amd import api; print(api.lib().trh_version().decode())')"
  for rep in 1 2; do
    for pipe in 0 1; do
      echo "## TRH_NTT_PIPE=$pipe (run $rep)"
      TRH_NTT_PIPE=$pipe python3 tools/ntt_probe.py 22 50
      TRH_NTT_PIPE=$pipe python3 tools/ntt_probe.py 20 50
      TRH_NTT_PIPE=$pipe python3 tools/ntt_probe.py 24 20
      TRH_NTT_PIPE=$pipe python3 tools/ntt_probe.py 18 20 320
      TRH_NTT_PIPE=$pipe python3 tools/ext_probe.py 18 64 6 5
    done
  done
  for lib in _ab/*/libtrh.so; do
    [ -f "$lib" ] || continue
    echo "## $lib TRH_NTT_PIPE=1"
    TRH_LIB_PATH=$lib TRH_NTT_PIPE=1 python3 tools/ntt_probe.py 22 50
    TRH_LIB_PATH=$lib TRH_NTT_PIPE=1 python3 tools/ntt_probe.py 18 20 320
    TRH_LIB_PATH=$lib TRH_NTT_PIPE=1 python3 tools/ext_probe.py 18 64 6 5
  done
  for wg in 1 3; do
    echo "## TRH_NTT_PIPE=1 TRH_NTT_PIPE_WG=$wg"
    TRH_NTT_PIPE_WG=$wg python3 tools/ntt_probe.py 22 50
  done
} 2>&1 | tee "$OUT"
