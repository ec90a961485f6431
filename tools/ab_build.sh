#!/bin/bash
# Build libtrh.so of another commit next to the working tree's, for same-box A/B timing (boxes differ by +-3 %):
#   tools/ab_build.sh <git-ref> <name>     -> _ab/<name>/libtrh.so      (then: TRH_LIB_PATH=_ab/<name>/libtrh.so python tools/msm_probe.py ...)
set -eu
REF=${1:?git ref}; NAME=${2:?name}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/_ab/$NAME
rm -rf "$OUT"; mkdir -p "$OUT/src"
git -C "$ROOT" archive "$REF" tiny-ram-halo2_amd/csrc include Makefile | tar -x -C "$OUT/src"
make -s -j8 -C "$OUT/src" tiny-ram-halo2_amd/libtrh.so
cp "$OUT/src/tiny-ram-halo2_amd/libtrh.so" "$OUT/libtrh.so"
rm -rf "$OUT/src"
echo "$OUT/libtrh.so"
