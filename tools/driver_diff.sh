#!/bin/bash
# Per-step times of the k = 18 witness-shaped replay from the two drivers on ONE box:
#   tools/driver_diff.sh <tag>  -> gpurun_out/driver_diff_<tag>.md   (copied to profiles/<tag>_driver_diff.md)
# Python mirror (torch events), compiled driver with device events (trh_event_*: the default since round 6) and with the host clock between
# stream synchronisations (--host-clock: what it reported until round 5).  Each twice, alternating, the second pass of each is tabulated.
set -u
TAG=${1:-r06}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
export LD_LIBRARY_PATH=tiny-ram-halo2_amd
for pass in 1 2; do
  python3 -m tiny_ram_halo2_amd.replay --word-bits 32 --columns witness --no-keygen 2>/dev/null | tail -1 > gpurun_out/dd_py_$pass.json
  ./examples/replay --word-bits 32 --columns witness 2>/dev/null | tail -1 > gpurun_out/dd_ev_$pass.json
  ./examples/replay --word-bits 32 --columns witness --host-clock 2>/dev/null | tail -1 > gpurun_out/dd_host_$pass.json
done
python3 - <<'PY' > gpurun_out/driver_diff_$TAG.md
import json
def load(p):
    try: return json.load(open(p))
    except Exception as e: return {"error": str(e)}
py, ev, ho = (load(f"gpurun_out/dd_{k}_2.json") for k in ("py", "ev", "host"))
a, b, c = py.get("gpu_ms", {}), ev.get("ms", {}), ho.get("ms", {})
print("# k = 18 witness-shaped create_proof replay, per step, one box, second pass of each driver (ms)\n")
print("| step | replay.py (torch events) | examples/replay, device events (trh_event_*) | examples/replay --host-clock (sync, host clock, sync) | host clock - events |")
print("|---|---|---|---|---|")
keys = [k for k in a if k in b or k == "product_columns"]
tot = [0.0, 0.0, 0.0]
for k in a:
    x, y, z = a.get(k, 0.0), b.get(k), c.get(k)
    print(f"| {k} | {x:.2f} | {'' if y is None else f'{y:.2f}'} | {'' if z is None else f'{z:.2f}'} | {'' if y is None or z is None else f'{z - y:+.2f}'} |")
print(f"| total | {py.get('gpu_ms_total', 0):.2f} | {ev.get('ms_total', 0):.2f} | {ho.get('ms_total', 0):.2f} | {ho.get('ms_total', 0) - ev.get('ms_total', 0):+.2f} |")
print("\n(the compiled driver has no separate `product_columns` step: its permutation / lookup products are inside `lookup_permute` and `h_eval`'s neighbours; totals are over each driver's own steps)")
print("\nfirst pass of each, totals:", load("gpurun_out/dd_py_1.json").get("gpu_ms_total"), load("gpurun_out/dd_ev_1.json").get("ms_total"), load("gpurun_out/dd_host_1.json").get("ms_total"))
PY
cat gpurun_out/driver_diff_$TAG.md
