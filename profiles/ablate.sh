#!/bin/bash
# NOTE (round 2): the diagnostic variants this script builds (-DSAPCU_ABL_*, -DSAPCU_RING_STAMPS, -DSAPCU_BT_STAMPS,
# -DSAPCU_BT_STORE_POLICY, -DSAPCU_BT_STAGGER), the two-workgroups-per-CU kernel (SAPCU_BT=2) and the ring kernel's fused
# softmax epilogue (EPI_SOFTMAX_AGG) were removed from the shipped sources after round 1; they live in git history at
# commit 61ec3ff.  To rebuild one: `git worktree add /tmp/abl 61ec3ff` and run this script there.  The measurements taken
# with them are recorded in DESIGN.md section 4.1 / 4.1b.
# Diagnostic builds of the ring GEMM (profiles/abl/*.so, git-ignored) — run here before gpurun, then point a
# microbenchmark at one with SAPCU_LIB=profiles/abl/libsapcu_<NAME>.so.
#   epilogue ablations   NO_GATHER  NO_C2  NO_LIF          one piece of the attention/LIF epilogue compiled out
#   delivery / compute   NO_MFMA (DMAs, waits, barriers only)   NO_DMA (LDS reads + MFMAs, rings never refilled)
#   latency              A_HOT (every tile reads the first row panel: all activation reads hit L2)
#   DRAM locality        A_CONTIG (each k-step's activation slot read as one contiguous 16 KiB run), A_CONTIG_NO_MFMA
#   tile order           CONTIG (-DSAPCU_RING_TILES_CONTIGUOUS: same speed, 3.8x the HBM fetches)
#   big-tile kernel      BT_NO_MFMA (delivery + epilogue)  BT_NO_EPI (k-loop only)   — run with SAPCU_BT=1
#   stamps               STAMPS (s_memtime per producer k-step segment; read with profiles/ring_stamps.py)
# Usage: bash profiles/ablate.sh [NAME ...]      (default: all)
set -e
cd "$(dirname "$0")/.."
SRC=c-users-sayakdutta-self-supervised-arbitrary-scale-point-cloud-upsampling-via-snn_amd/csrc
mkdir -p profiles/abl
make -C $SRC > /dev/null
build() {   # name, extra flags
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off $2 -c $SRC/gemm_sf16_ring.hip -o profiles/abl/ring_$1.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o profiles/abl/libsapcu_$1.so profiles/abl/ring_$1.o \
      $(ls $SRC/*.o | grep -v gemm_sf16_ring.o)
}
build_bt() {   # name, extra flags: diagnostic builds of the big-tile kernel
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off $2 -c $SRC/gemm_sf16_bt.hip -o profiles/abl/bt_$1.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o profiles/abl/libsapcu_$1.so profiles/abl/bt_$1.o \
      $(ls $SRC/*.o | grep -v gemm_sf16_bt.o)
}
for v in ${@:-NO_GATHER NO_C2 NO_LIF NO_MFMA NO_DMA A_HOT CONTIG STAMPS}; do
  case $v in
    CONTIG) build $v -DSAPCU_RING_TILES_CONTIGUOUS ;;
    STAMPS) build $v -DSAPCU_RING_STAMPS ;;
    BT_ST_NT) build_bt $v -DSAPCU_BT_STORE_POLICY=1 ;;
    BT_ST_SC1) build_bt $v -DSAPCU_BT_STORE_POLICY=2 ;;
    BT_ST_SC01) build_bt $v -DSAPCU_BT_STORE_POLICY=3 ;;
    BT_STAMPS) build_bt $v -DSAPCU_BT_STAMPS ;;
    BT_STAGGER*) build_bt $v -DSAPCU_BT_STAGGER=${v#BT_STAGGER} ;;
    BT_*) build_bt $v -DSAPCU_ABL_$v ;;
    A_CONTIG_NO_MFMA) build $v "-DSAPCU_ABL_A_CONTIG -DSAPCU_ABL_NO_MFMA" ;;
    *)      build $v -DSAPCU_ABL_$v ;;
  esac
done
ls -la profiles/abl/*.so
