// Shared by the GEMM kernels (gemm_f32.hip, gemm_sf16.hip): LDS-only barrier, and the consumer-side
// epilogue of four accumulator elements (bias, T-step neuron loop, activations, residuals, gathers).
#pragma once
#include "common.h"

namespace sapcu {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits vmcnt(0), i.e. for the
// consumer waves' global stores/gathers of the epilogue to complete, which would stall the MFMA waves
// at every k-step; the hardware barrier needs only the LDS writes (lgkmcnt) to have landed.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Make the compiler treat a loaded value as consumed here, so its s_waitcnt lands at this point and
// not (as a conservative vmcnt(0)) inside the epilogue loop behind every in-flight store.
__device__ __forceinline__ float settle(float x) {
    asm volatile("" : "+v"(x));
    return x;
}

// "Split rows": a [rows, K] activation tensor whose producer already split every value for the split-f16
// GEMM (gemm_sf16_ring.hip).  A row keeps the f32 row's footprint (pitch = ld floats = 4*ld bytes): the first
// K halves are hi = f16_rn(x), the halves starting at half-index ld (byte 2*ld) are lo = f16_rn(x - hi).
__device__ __forceinline__ void store_split(float* base, int64_t row, int ld, int col, float v) {
    _Float16* rp = reinterpret_cast<_Float16*>(base + row * ld);
    const _Float16 hi = (_Float16)v;
    rp[col] = hi;
    rp[ld + col] = (_Float16)(v - (float)hi);
}

// Epilogue of W accumulator elements of one lane: same column, rows row..row+W-1.  The W neuron chains
// are independent, so unrolling them gives the VALU W-way ILP; loads are issued for all before the first
// is used, stores after the last is computed.
template <int EPI, int W>
__device__ __forceinline__ void epilogue_group(const GemmArgs& g, const float (&acc)[W], int64_t row, int col,
                                               float bias, const NeuronP& np, const float (&q)[W],
                                               const float (&kf)[W]) {
    float v[W], res[W];
    bool ok[W];
#pragma unroll
    for (int u = 0; u < W; ++u) ok[u] = (row + u) < g.r;
    if (EPI == EPI_RESID || EPI == EPI_RESID_GELU) {
#pragma unroll
        for (int u = 0; u < W; ++u) res[u] = ok[u] ? g.resid[(row + u) * g.ldr + col] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < W; ++u) v[u] = __fadd_rn(acc[u], bias);
    if (EPI == EPI_LIF || EPI == EPI_LIF_ATTN) lif_selfloop_n<W>(v, np, g.lif_T);
#pragma unroll
    for (int u = 0; u < W; ++u) {
        if (EPI == EPI_GELU) v[u] = gelu_erf(v[u]);
        if (EPI == EPI_LRELU) v[u] = lrelu02(v[u]);
        if (EPI == EPI_RESID) v[u] = __fadd_rn(v[u], res[u]);
        if (EPI == EPI_RESID_GELU) v[u] = gelu_erf(__fadd_rn(v[u], res[u]));
    }
#pragma unroll
    for (int u = 0; u < W; ++u) {
        if (!ok[u]) continue;
        if (g.c_split) {
            store_split(g.c, row + u, g.ldc, col, v[u]);
            if (EPI != EPI_LIF && EPI != EPI_LIF_ATTN && !(fabsf(v[u]) < 65504.0f) && g.ovf) atomicAdd(g.ovf, 1);
        } else {
            g.c[(row + u) * g.ldc + col] = v[u];
        }
        // attn_in = q_i - k_j + pos_enc (fn/snn_coder.py:368), operand of the next GEMM
        if (EPI == EPI_LIF_ATTN) {
            const float ai = __fadd_rn(__fsub_rn(q[u], kf[u]), v[u]);
            if (g.c2_split) store_split(g.c2, row + u, g.ldc, col, ai);
            else g.c2[(row + u) * g.ldc + col] = ai;
        }
    }
}

template <int EPI>
__device__ __forceinline__ void epilogue_group4(const GemmArgs& g, const float (&acc)[4], int64_t row, int col,
                                                float bias, const NeuronP& np, const float (&q)[4],
                                                const float (&kf)[4]) {
    epilogue_group<EPI, 4>(g, acc, row, col, bias, np, q, kf);
}

}  // namespace sapcu
