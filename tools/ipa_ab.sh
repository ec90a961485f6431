#!/bin/bash
# Same-box A/B of the k = 18 opening and of lone commitments between two builds: tools/ipa_ab.sh <other libtrh.so>
cd ${GRAFT_REPO_ROOT:-.}
OTHER=${1:-_ab/base/libtrh.so}
for pass in 1 2 3; do
  echo "## this build (pass $pass)"; python3 tools/ipa_probe.py 18 2>&1 | tail -1; python3 tools/lone_sparse_probe.py 2>/dev/null | tail -1 | cut -c1-400
  echo "## $OTHER (pass $pass)"; TRH_LIB_PATH=$OTHER python3 tools/ipa_probe.py 18 2>&1 | tail -1; TRH_LIB_PATH=$OTHER python3 tools/lone_sparse_probe.py 2>/dev/null | tail -1 | cut -c1-400
done
