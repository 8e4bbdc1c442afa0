// Shared by the GEMM kernels (gemm_f32.hip, gemm_sf16.hip): LDS-only barrier, and the consumer-side
// epilogue of four accumulator elements (bias, T-step neuron loop, activations, residuals, gathers).
#pragma once
#include "common.h"

namespace sapcu {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// The split-f16 product instruction of the ring / f32-A kernels.  -DSAPCU_TIMING_MFMA16 (diagnostic, WRONG results): the same
// flops issued as two v_mfma_f32_16x16x32_f16 on quarters of the accumulator — the timing-only prototype that bounds what moving
// these two kernels to the small MFMA shape could gain (profiles/r04_ab_mfma16_ring.txt).
typedef _Float16 epi_half8 __attribute__((ext_vector_type(8)));
typedef float epi_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x16 mfma_32x32x16_f16(epi_half8 a, epi_half8 b, f32x16 c) {
#ifdef SAPCU_TIMING_MFMA16
    epi_f32x4 q0 = {c[0], c[1], c[2], c[3]}, q1 = {c[8], c[9], c[10], c[11]};
    q0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, q0, 0, 0, 0);
    q1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, q1, 0, 0, 0);
    c[0] = q0[0]; c[1] = q0[1]; c[2] = q0[2]; c[3] = q0[3];
    c[8] = q1[0]; c[9] = q1[1]; c[10] = q1[2]; c[11] = q1[3];
    return c;
#else
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
#endif
}

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

__device__ __forceinline__ void settle4(float4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }

// "Split rows": a [rows, K] activation tensor whose producer already split every value for the split-f16
// GEMM (gemm_sf16_ring.hip, gemm_sf16_bt.hip): hi = f16_rn(x), lo = f16_rn(x - hi).  A row keeps the f32 row's footprint
// (pitch = ld floats = 4*ld bytes = 2*ld halves).  Layout inside the row:
//   ld % 32 == 0 (every tensor of the models): INTERLEAVED by groups of 32 elements — group q occupies one 128-byte line,
//                hi halves of elements 32q..32q+31 at half-index 64q, their lo halves at 64q + 32.  One k-step of the GEMMs
//                (32 elements) is then ONE line per row: with separate planes two consecutive k-steps shared every line (the
//                first missed to HBM, the second waited behind it), and an epilogue's 32 lanes fill a whole line.
//   otherwise:   hi halves at half-index col, lo halves at ld + col (two planes).
__device__ __forceinline__ bool split_interleaved(int ld) { return (ld & 31) == 0; }
__device__ __forceinline__ int split_hi_index(int ld, int col) { return split_interleaved(ld) ? ((col >> 5) << 6) + (col & 31) : col; }
__device__ __forceinline__ int split_lo_index(int ld, int col) { return split_interleaved(ld) ? ((col >> 5) << 6) + 32 + (col & 31) : ld + col; }

__device__ __forceinline__ void store_split(float* base, int64_t row, int ld, int col, float v) {
    _Float16* rp = reinterpret_cast<_Float16*>(base + row * ld);
    const _Float16 hi = (_Float16)v;
    rp[split_hi_index(ld, col)] = hi;
    rp[split_lo_index(ld, col)] = (_Float16)(v - (float)hi);
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
    if (EPI == EPI_LIF || EPI == EPI_LIF_ATTN || EPI == EPI_LIF_MAX) lif_selfloop_n<W>(v, np, g.lif_T);
#pragma unroll
    for (int u = 0; u < W; ++u) {
        if (EPI == EPI_GELU) v[u] = gelu_erf(v[u]);
        if (EPI == EPI_LRELU || EPI == EPI_LRELU_MAX) v[u] = lrelu02(v[u]);
        if (EPI == EPI_RESID) v[u] = __fadd_rn(v[u], res[u]);
        if (EPI == EPI_RESID_GELU) v[u] = gelu_erf(__fadd_rn(v[u], res[u]));
    }
    if (EPI == EPI_LRELU_MAX || EPI == EPI_LIF_MAX) {
        // max over the row groups instead of a store: consecutive rows of one group are combined in registers, one
        // integer atomicMax per (group, column) and lane (the maximum does not depend on the order: exact)
        int64_t cur = -1;
        float best = 0.f;
#pragma unroll
        for (int u = 0; u < W; ++u) {
            if (!ok[u]) continue;
            const int64_t grp = (row + u) / g.max_m;
            if (grp != cur) {
                if (cur >= 0) atomicMax(g.max_keys + cur * g.n + col, float_max_key(best));
                cur = grp;
                best = v[u];
            } else {
                best = fmaxf(best, v[u]);
            }
        }
        if (cur >= 0) atomicMax(g.max_keys + cur * g.n + col, float_max_key(best));
        return;
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

// ---------------------------------------------------------------------------------------------
// Row layout (ring GEMM consumers): a lane holds FOUR CONSECUTIVE COLUMNS of one row, so every global access
// of the epilogue is 8 or 16 bytes per lane (f32: one 16-byte store; split rows: two 8-byte stores; q/k
// gathers and residuals: 16-byte loads).  The vector-memory path handles a wave instruction at the same
// rate whatever its width, so this is 4x fewer address cycles than the one-column-per-lane layout above.
// ---------------------------------------------------------------------------------------------
struct ColParams4 {          // bias and RAW neuron parameters of 4 consecutive columns (clamped at use, so that a
    float4 bias;             // reload can stay in flight until the next piece needs it)
    float4 decay, adapt, rdecay, theta0;
};

// VEC = false: any n / alignment (element-wise accesses, columns >= n masked) — odd shapes only
template <bool VEC>
__device__ __forceinline__ float4 ld4_cols(const float* p, int col, int n) {
    if (VEC) return ld4(p + col);
    float4 r;
    r.x = col < n ? p[col] : 0.f;
    r.y = col + 1 < n ? p[col + 1] : 0.f;
    r.z = col + 2 < n ? p[col + 2] : 0.f;
    r.w = col + 3 < n ? p[col + 3] : 0.f;
    return r;
}

template <int EPI, bool VEC>
__device__ __forceinline__ ColParams4 load_col_params4(const GemmArgs& g, int col, bool valid) {
    ColParams4 c;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    c.bias = (valid && g.bias) ? ld4_cols<VEC>(g.bias, col, g.n) : z;
    c.decay = c.adapt = c.rdecay = c.theta0 = z;
    if (EPI == EPI_LIF || EPI == EPI_LIF_ATTN) {
        const int cc = valid ? col : 0;
        c.decay = ld4_cols<VEC>(g.lif, cc, g.n);
        c.adapt = ld4_cols<VEC>(g.lif + g.n, cc, g.n);
        c.rdecay = ld4_cols<VEC>(g.lif + 2 * (int64_t)g.n, cc, g.n);
        c.theta0 = ld4_cols<VEC>(g.lif + 3 * (int64_t)g.n, cc, g.n);
    }
    return c;
}

__device__ __forceinline__ void clamp_col_params4(const ColParams4& c, NeuronP2 (&np)[2]) {
    np[0].decay = f32x2{clampf(c.decay.x, 0.1f, 0.99f), clampf(c.decay.y, 0.1f, 0.99f)};
    np[1].decay = f32x2{clampf(c.decay.z, 0.1f, 0.99f), clampf(c.decay.w, 0.1f, 0.99f)};
    np[0].adapt = f32x2{clampf(c.adapt.x, 0.001f, 0.1f), clampf(c.adapt.y, 0.001f, 0.1f)};
    np[1].adapt = f32x2{clampf(c.adapt.z, 0.001f, 0.1f), clampf(c.adapt.w, 0.001f, 0.1f)};
    np[0].rdecay = f32x2{clampf(c.rdecay.x, 0.1f, 0.95f), clampf(c.rdecay.y, 0.1f, 0.95f)};
    np[1].rdecay = f32x2{clampf(c.rdecay.z, 0.1f, 0.95f), clampf(c.rdecay.w, 0.1f, 0.95f)};
    np[0].theta0 = f32x2{c.theta0.x, c.theta0.y};
    np[1].theta0 = f32x2{c.theta0.z, c.theta0.w};
}

typedef _Float16 half4 __attribute__((ext_vector_type(4)));

// 4 consecutive values of a row
template <bool VEC>
__device__ __forceinline__ void store_f32x4(float* base, int64_t row, int ld, int col, int n, const float (&v)[4]) {
    if (VEC) {
        *reinterpret_cast<float4*>(base + row * ld + col) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (col + u < n) base[row * ld + col + u] = v[u];
    }
}

// ... of a split row (8-byte stores)
template <bool VEC>
__device__ __forceinline__ void store_split4(float* base, int64_t row, int ld, int col, int n, const float (&v)[4]) {
    if (!VEC) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (col + u < n) store_split(base, row, ld, col + u, v[u]);
        return;
    }
    _Float16* rp = reinterpret_cast<_Float16*>(base + row * ld);
    half4 hi, lo;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        hi[u] = (_Float16)v[u];
        lo[u] = (_Float16)(v[u] - (float)hi[u]);
    }
    *reinterpret_cast<half4*>(rp + split_hi_index(ld, col)) = hi;      // col % 4 == 0: the 4 columns stay inside one group
    *reinterpret_cast<half4*>(rp + split_lo_index(ld, col)) = lo;
}

// acc = columns col..col+3 of `row` (row < g.r and col < g.n checked by the caller; n % 4 == 0).
// Two halves so that the caller can issue loads (next column parameters) between the arithmetic and the stores.
template <int EPI, bool VEC>
__device__ __forceinline__ void epilogue_row4_compute(const GemmArgs& g, const float4 acc, int64_t row, int col,
                                                      const ColParams4& cp, float (&v)[4]) {
    // acc carries the x16 of the pre-scaled weights: acc/16 is exact, so one FMA equals the separate multiply and add
    v[0] = __fmaf_rn(acc.x, 0.0625f, cp.bias.x);
    v[1] = __fmaf_rn(acc.y, 0.0625f, cp.bias.y);
    v[2] = __fmaf_rn(acc.z, 0.0625f, cp.bias.z);
    v[3] = __fmaf_rn(acc.w, 0.0625f, cp.bias.w);
    float4 res = make_float4(0.f, 0.f, 0.f, 0.f);
    if (EPI == EPI_RESID || EPI == EPI_RESID_GELU) res = ld4_cols<VEC>(g.resid + row * g.ldr, col, g.n);
    if (EPI == EPI_LIF || EPI == EPI_LIF_ATTN) {
        f32x2 pv[2] = {f32x2{v[0], v[1]}, f32x2{v[2], v[3]}};
        NeuronP2 np[2];
        clamp_col_params4(cp, np);
        lif_selfloop_pairs<2>(pv, np, g.lif_T);
        v[0] = pv[0].x; v[1] = pv[0].y; v[2] = pv[1].x; v[3] = pv[1].y;
    }
    const float rs[4] = {res.x, res.y, res.z, res.w};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        if (EPI == EPI_GELU) v[u] = gelu_erf(v[u]);
        if (EPI == EPI_LRELU) v[u] = lrelu02(v[u]);
        if (EPI == EPI_RESID) v[u] = __fadd_rn(v[u], rs[u]);
        if (EPI == EPI_RESID_GELU) v[u] = gelu_erf(__fadd_rn(v[u], rs[u]));
    }
}

template <int EPI, bool VEC>
__device__ __forceinline__ void epilogue_row4_store(const GemmArgs& g, const float (&v)[4], int64_t row, int col,
                                                    const float4 q, const float4 kf) {
    // attn_in = q_i - k_j + pos_enc (fn/snn_coder.py:368), operand of the next GEMM.  Formed BEFORE the first
    // store of this piece: the wait for the gathers then has only older stores ahead of it in the (in-order)
    // vector-memory counter, instead of draining this piece's own stores.
    float ai[4] = {0.f, 0.f, 0.f, 0.f};
    if (EPI == EPI_LIF_ATTN) {
        ai[0] = __fadd_rn(__fsub_rn(q.x, kf.x), v[0]);
        ai[1] = __fadd_rn(__fsub_rn(q.y, kf.y), v[1]);
        ai[2] = __fadd_rn(__fsub_rn(q.z, kf.z), v[2]);
        ai[3] = __fadd_rn(__fsub_rn(q.w, kf.w), v[3]);
        asm volatile("" : "+v"(ai[0]), "+v"(ai[1]), "+v"(ai[2]), "+v"(ai[3])::"memory");
    }
    if (g.c_split) {
        store_split4<VEC>(g.c, row, g.ldc, col, g.n, v);
        if (EPI != EPI_LIF && EPI != EPI_LIF_ATTN && g.ovf) {
            float big = 0.f;
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (VEC || col + u < g.n) big = fmaxf(big, fabsf(v[u]));     // fmaxf drops a NaN: test it separately
            const bool nan = (VEC || col < g.n) && !(v[0] == v[0] && v[1] == v[1] && v[2] == v[2] && v[3] == v[3]);
            if (!(big < 65504.0f) || nan) atomicAdd(g.ovf, 1);
        }
    } else {
        store_f32x4<VEC>(g.c, row, g.ldc, col, g.n, v);
    }
    if (EPI == EPI_LIF_ATTN) {
        if (g.c2_split) store_split4<VEC>(g.c2, row, g.ldc, col, g.n, ai);
        else store_f32x4<VEC>(g.c2, row, g.ldc, col, g.n, ai);
    }
}

template <int EPI>
__device__ __forceinline__ void epilogue_group4(const GemmArgs& g, const float (&acc)[4], int64_t row, int col,
                                                float bias, const NeuronP& np, const float (&q)[4],
                                                const float (&kf)[4]) {
    epilogue_group<EPI, 4>(g, acc, row, col, bias, np, q, kf);
}

}  // namespace sapcu
