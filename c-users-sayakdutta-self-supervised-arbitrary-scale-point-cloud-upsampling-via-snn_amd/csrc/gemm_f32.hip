// fp32 MFMA GEMM with fused operand prologues and neuron/activation epilogues (gfx950).
//
//   C[r, n] = epi( pro(A)[r, k] . W[n, k]^T + bias[n] )
//
// This one kernel carries every 1x1 convolution / Linear of both networks (BatchNorm is folded into
// W and bias by the packer): rows r are (patch, point[, neighbour]) tuples flattened over the whole
// query batch, so a launch has thousands of 128x128 tiles.
//
// MI355X mapping:
//   * v_mfma_f32_32x32x2_f32 — exact f32 products/accumulation (bitwise an fmaf chain), the only
//     MFMA that holds the 1e-4 parity contract without operand splitting; peak 157 TFLOP/s;
//   * 256 threads = 4 waves as 2x2, each wave owns a 64x64 sub-tile = 2x2 MFMA tiles (64 acc regs);
//   * operands are staged global -> registers -> LDS ([row][k], row stride 36 floats), double
//     buffered, ONE barrier per 32-deep k-step; a lane fetches 4 consecutive k with ds_read_b128 and
//     feeds them to 4 MFMAs (the k index inside an 8-wide group is permuted identically for A and W,
//     which is legal because the sum over k is order-free up to rounding);
//   * 73.7 KB LDS per workgroup -> 2 workgroups per CU, so one workgroup's VALU epilogue (the
//     T-step neuron loop, state in registers) overlaps the other's MFMA main loop;
//   * blockIdx is remapped so the n-tiles of one row panel run on the same XCD (shared L2).
#include "common.h"

namespace sapcu {

constexpr int BM = 128, BN = 128, BK = 32, LDT = BK + 4;
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

template <int EPI, int PRO>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float smem[2][(BM + BN) * LDT];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r32 = lane & 31, h = lane >> 5;

    // XCD-aware bijective remap: blocks with equal blockIdx%8 share an XCD (and its L2)
    const int ntn = (g.n + BN - 1) / BN;
    const int64_t nblk = gridDim.x;
    const int64_t qd = nblk >> 3, rem = nblk & 7;
    const int64_t xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int64_t logical = xcd * qd + (xcd < rem ? xcd : rem) + slot;
    const int tn = (int)(logical % ntn);
    const int64_t tm = logical / ntn;
    const int64_t row0 = tm * BM;
    const int col0 = tn * BN;

    const int srow = tid >> 3;
    const int skc = (tid & 7) * 4;
    const float* arow[4];
    const float* qrow[4];
    const float* krow[4];
    const float* wrow[4];
    bool aok[4], wok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t row = row0 + srow + 32 * i;
        aok[i] = row < g.r;
        const int64_t rr = aok[i] ? row : 0;
        arow[i] = g.a + rr * g.lda + skc;
        if (PRO == PRO_ATTN_IN) {
            const int64_t pt = rr / g.kk;
            const int64_t patch = pt / g.mpts;
            qrow[i] = g.q + pt * g.ldq + skc;
            krow[i] = g.kf + (patch * g.mpts + g.idx[rr]) * g.ldq + skc;
        }
        const int nn = col0 + srow + 32 * i;
        wok[i] = nn < g.n;
        wrow[i] = g.w + (int64_t)(wok[i] ? nn : 0) * g.k + skc;
    }
    float4 ra[4], rw[4];
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (aok[i]) {
                v = ld4(arow[i] + k0);
                if (PRO == PRO_ATTN_IN) {   // attn_in = q_i - k_j + pos_enc  (fn/snn_coder.py:368)
                    const float4 q = ld4(qrow[i] + k0);
                    const float4 kf = ld4(krow[i] + k0);
                    v.x = __fadd_rn(__fsub_rn(q.x, kf.x), v.x);
                    v.y = __fadd_rn(__fsub_rn(q.y, kf.y), v.y);
                    v.z = __fadd_rn(__fsub_rn(q.z, kf.z), v.z);
                    v.w = __fadd_rn(__fsub_rn(q.w, kf.w), v.w);
                }
            }
            ra[i] = v;
            rw[i] = wok[i] ? ld4(wrow[i] + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_tile = [&](int buf) {
        float* sA = smem[buf];
        float* sW = smem[buf] + BM * LDT;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<float4*>(&sA[(srow + 32 * i) * LDT + skc]) = ra[i];
            *reinterpret_cast<float4*>(&sW[(srow + 32 * i) * LDT + skc]) = rw[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk = g.k / BK;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) load_tile((kt + 1) * BK);
        const float* sA = smem[cur] + (wm * 64 + r32) * LDT + h * 4;
        const float* sW = smem[cur] + BM * LDT + (wn * 64 + r32) * LDT + h * 4;
#pragma unroll
        for (int k8 = 0; k8 < BK / 8; ++k8) {
            const float4 a0 = ld4(sA + k8 * 8);
            const float4 a1 = ld4(sA + 32 * LDT + k8 * 8);
            const float4 b0 = ld4(sW + k8 * 8);
            const float4 b1 = ld4(sW + 32 * LDT + k8 * 8);
            const float av[2][4] = {{a0.x, a0.y, a0.z, a0.w}, {a1.x, a1.y, a1.z, a1.w}};
            const float bv[2][4] = {{b0.x, b0.y, b0.z, b0.w}, {b1.x, b1.y, b1.z, b1.w}};
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][e], bv[j][e], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tile(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: lane owns column (n) r32 of each 32-wide tile, 16 rows per tile
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = col0 + wn * 64 + j * 32 + r32;
        if (col >= g.n) continue;
        const float bias = g.bias ? g.bias[col] : 0.f;
        NeuronP np;
        if (EPI == EPI_LIF) np = load_lif(g.lif, g.n, col);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t row = row0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (row >= g.r) continue;
                float v = __fadd_rn(acc[i][j][e], bias);
                if (EPI == EPI_LIF) v = lif_selfloop(v, np, g.lif_T);
                if (EPI == EPI_GELU) v = gelu_erf(v);
                if (EPI == EPI_LRELU) v = lrelu02(v);
                if (EPI == EPI_RESID) v = __fadd_rn(v, g.resid[row * g.ldr + col]);
                if (EPI == EPI_RESID_GELU) v = gelu_erf(__fadd_rn(v, g.resid[row * g.ldr + col]));
                g.c[row * g.ldc + col] = v;
            }
        }
    }
}

template <int EPI, int PRO>
static int launch_t(const GemmArgs& g, hipStream_t st) {
    const int64_t tm = (g.r + BM - 1) / BM;
    const int64_t tn = (g.n + BN - 1) / BN;
    const int64_t grid = tm * tn;
    if (grid > 0x7fffffffLL) {
        set_error("gemm: grid too large (%lld tiles)", (long long)grid);
        return SAPCU_ERR_ARG;
    }
    hipLaunchKernelGGL((gemm_kernel<EPI, PRO>), dim3((unsigned)grid), dim3(256), 0, st, g);
    SAPCU_CHECK_LAUNCH();
    return SAPCU_OK;
}

int launch_gemm(const GemmArgs& g, hipStream_t st) {
    if (g.r == 0 || g.n == 0) return SAPCU_OK;
    SAPCU_CHECK_ARG(g.k > 0 && g.k % BK == 0, "gemm: k=%d must be a positive multiple of %d", g.k, BK);
    SAPCU_CHECK_ARG(g.lda % 4 == 0 && ((uintptr_t)g.a & 15) == 0 && ((uintptr_t)g.w & 15) == 0,
                    "gemm: A/W must be 16-byte aligned with lda %% 4 == 0 (lda=%d)", g.lda);
    if (g.pro == PRO_ATTN_IN) {
        SAPCU_CHECK_ARG(g.epi == EPI_LIF, "gemm: PRO_ATTN_IN is only built with EPI_LIF");
        SAPCU_CHECK_ARG(g.ldq % 4 == 0 && g.kk > 0 && g.mpts > 0 && g.idx && g.q && g.kf, "gemm: bad attn_in operands");
        return launch_t<EPI_LIF, PRO_ATTN_IN>(g, st);
    }
    switch (g.epi) {
        case EPI_BIAS: return launch_t<EPI_BIAS, PRO_PLAIN>(g, st);
        case EPI_LIF: return launch_t<EPI_LIF, PRO_PLAIN>(g, st);
        case EPI_GELU: return launch_t<EPI_GELU, PRO_PLAIN>(g, st);
        case EPI_RESID: return launch_t<EPI_RESID, PRO_PLAIN>(g, st);
        case EPI_LRELU: return launch_t<EPI_LRELU, PRO_PLAIN>(g, st);
        case EPI_RESID_GELU: return launch_t<EPI_RESID_GELU, PRO_PLAIN>(g, st);
        default: set_error("gemm: unknown epilogue %d", g.epi); return SAPCU_ERR_ARG;
    }
}

}  // namespace sapcu
