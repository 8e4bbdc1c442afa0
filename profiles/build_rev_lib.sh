#!/bin/bash
# Build libsapcu_hip.so from the kernel sources of a git revision (for same-box A/B runs with profiles/step_ab.py):
#   bash profiles/build_rev_lib.sh <rev|WORKTREE> <out.so>      e.g.  HEAD profiles/ab/libA.so   (*.so is git-ignored, but travels with gpurun)
#   EXTRA_CXXFLAGS=-DFE_STAMPS bash profiles/build_rev_lib.sh WORKTREE profiles/ab/lib_stamps.so     (profiles/fd_stamps.py)
set -e
REV=${1:-HEAD}
OUT=$(realpath -m ${2:-profiles/ab/libA.so})
ROOT=$(cd "$(dirname "$0")/.." && pwd)
PKG=c-users-sayakdutta-self-supervised-arbitrary-scale-point-cloud-upsampling-via-snn_amd
TMP=$(mktemp -d)
trap 'rm -rf "$TMP"' EXIT
if [ "$REV" == "WORKTREE" ]; then      # the sources as they are now (e.g. with EXTRA_CXXFLAGS=-DFE_STAMPS for a diagnostic build)
    mkdir -p "$TMP/$PKG/csrc" "$TMP/include"
    cp "$ROOT/$PKG/csrc/"*.hip "$ROOT/$PKG/csrc/"*.h "$ROOT/$PKG/csrc/"*.cpp "$ROOT/$PKG/csrc/Makefile" "$TMP/$PKG/csrc/"
    cp "$ROOT/include/"*.h "$TMP/include/"
else
    git -C "$ROOT" archive "$REV" $PKG/csrc include | tar -x -C "$TMP"
fi
make -C "$TMP/$PKG/csrc" -j8 all EXTRA_CXXFLAGS="$EXTRA_CXXFLAGS" > "$TMP/build.log" 2>&1 || { tail -20 "$TMP/build.log"; exit 1; }
mkdir -p "$(dirname "$OUT")"
cp "$TMP/$PKG/csrc/libsapcu_hip.so" "$OUT"
echo "built $OUT from $REV"
