#!/bin/bash
# Profiling ablations of the ring GEMM epilogue: builds variants of the library with one piece of the epilogue
# compiled out (-DSAPCU_ABL_*) into profiles/abl/ (git-ignored .so files) — run here, before gpurun.
set -e
cd "$(dirname "$0")/.."
SRC=c-users-sayakdutta-self-supervised-arbitrary-scale-point-cloud-upsampling-via-snn_amd/csrc
mkdir -p profiles/abl
for v in NO_GATHER NO_C2 NO_LIF "NO_GATHER -DSAPCU_ABL_NO_C2" ; do
  name=$(echo "$v" | sed 's/ -DSAPCU_ABL_/_/g')
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -DSAPCU_ABL_$v -c $SRC/gemm_sf16_ring.hip -o profiles/abl/ring_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o profiles/abl/libsapcu_$name.so profiles/abl/ring_$name.o \
      $(ls $SRC/*.o | grep -v gemm_sf16_ring.o)
done
ls -la profiles/abl/*.so
