#!/bin/bash
# Build libsapcu_hip.so from the kernel sources of a git revision (for same-box A/B runs with profiles/step_ab.py):
#   bash profiles/build_rev_lib.sh <rev> <out.so>      e.g.  HEAD profiles/ab/libA.so   (*.so is git-ignored, but travels with gpurun)
set -e
REV=${1:-HEAD}
OUT=$(realpath -m ${2:-profiles/ab/libA.so})
ROOT=$(cd "$(dirname "$0")/.." && pwd)
PKG=c-users-sayakdutta-self-supervised-arbitrary-scale-point-cloud-upsampling-via-snn_amd
TMP=$(mktemp -d)
trap 'rm -rf "$TMP"' EXIT
git -C "$ROOT" archive "$REV" $PKG/csrc include | tar -x -C "$TMP"
make -C "$TMP/$PKG/csrc" -j8 all > "$TMP/build.log" 2>&1 || { tail -20 "$TMP/build.log"; exit 1; }
mkdir -p "$(dirname "$OUT")"
cp "$TMP/$PKG/csrc/libsapcu_hip.so" "$OUT"
echo "built $OUT from $REV"
