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

// Epilogue of FOUR accumulator elements of one lane: same column, rows row..row+3.  The four neuron
// chains are independent, so unrolling them gives the VALU 4-way ILP; gathers are issued for all four
// before the first is used, stores after the last is computed.
template <int EPI>
__device__ __forceinline__ void epilogue_group4(const GemmArgs& g, const float (&acc)[4], int64_t row, int col,
                                                float bias, const NeuronP& np, const float (&q)[4],
                                                const float (&kf)[4]) {
    float v[4], res[4];
    bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) ok[u] = (row + u) < g.r;
    if (EPI == EPI_RESID || EPI == EPI_RESID_GELU) {
#pragma unroll
        for (int u = 0; u < 4; ++u) res[u] = ok[u] ? g.resid[(row + u) * g.ldr + col] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = __fadd_rn(acc[u], bias);
    if (EPI == EPI_LIF || EPI == EPI_LIF_ATTN) lif_selfloop_n<4>(v, np, g.lif_T);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        if (EPI == EPI_GELU) v[u] = gelu_erf(v[u]);
        if (EPI == EPI_LRELU) v[u] = lrelu02(v[u]);
        if (EPI == EPI_RESID) v[u] = __fadd_rn(v[u], res[u]);
        if (EPI == EPI_RESID_GELU) v[u] = gelu_erf(__fadd_rn(v[u], res[u]));
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        if (!ok[u]) continue;
        g.c[(row + u) * g.ldc + col] = v[u];
        // attn_in = q_i - k_j + pos_enc (fn/snn_coder.py:368), operand of the next GEMM
        if (EPI == EPI_LIF_ATTN) g.c2[(row + u) * g.ldc + col] = __fadd_rn(__fsub_rn(q[u], kf[u]), v[u]);
    }
}


}  // namespace sapcu
