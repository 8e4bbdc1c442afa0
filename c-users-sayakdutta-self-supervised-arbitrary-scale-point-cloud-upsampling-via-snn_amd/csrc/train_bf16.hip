// bf16-operand GEMMs of the fn training step (BASELINE config 5 "trainfn.py one epoch bf16"): the HIP
// counterpart of running the reference's Conv / Linear layers under torch.amp.autocast (fn/trainer.py:67-83) — operands
// rounded to bf16 (round to nearest even), products accumulated in f32 on v_mfma_f32_32x32x16_bf16.  Unlike the inference
// path's split-f16 GEMMs this IS a precision reduction, and it is opt-in (Trainer(use_amp=True)); the exact-f32 kernels of
// train_ops.hip stay the parity reference.  bf16 keeps f32's exponent range, which the backward needs: gradients reach 1e6
// (surrogate slope 10 through ~25 neuron layers) and leave the f16 range.
//
//   NT   c[r, n]  = sum_k a[r, k] * w[n, k] (+ bias[n])        forward (w = W) and data gradient (w = W^T)
//   TN   dw[n, k] = sum_r dy[r, n] * x[r, k]                   weight gradient: the reduction runs over the ROWS, which are
//                                                               the slow axis of both operands, so the tiles are transposed on
//                                                               their way into LDS; row slabs -> ordered slab sum (deterministic)
//
// One 256-thread workgroup per 128 x 128 output tile, k-steps of 32, wave tile 64 x 64 (2 x 2 MFMA blocks); operands are
// converted f32 -> bf16 in registers while staging into LDS ([row][32 halves], 16-byte chunks XOR-swizzled by (row >> 2) & 3:
// the conflict-free fragment layout of the inference GEMMs).  These launches are small (rows = 3 072 ... 147 456 at the
// reference's training batch): the step is bound by launch count and the element-wise kernels, not by this kernel.
#include "common.h"
#include "gemm_epi.h"

namespace sapcu {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int BF_BM = 128, BF_BN = 128, BF_BK = 32;

__device__ __forceinline__ unsigned bf_lds_off(int row, int k) {      // byte offset of element (row, k) of a [rows][32] bf16 tile
    return (unsigned)(row * 64 + ((((k >> 3) ^ ((row >> 2) & 3))) * 16) + (k & 7) * 2);
}

// TRANS = false: A(m, kk) = a[m * lda + kk], B(n, kk) = b[n * ldb + kk]                    (k contiguous)
// TRANS = true : A(m, kk) = a[kk * lda + m], B(n, kk) = b[kk * ldb + n], kk in this slab   (m / n contiguous)
template <bool TRANS>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb,
                                                        int64_t M, int N, int64_t K, int64_t kslab, const float* __restrict__ bias,
                                                        float* __restrict__ c, int64_t ldc, int64_t slab_stride) {
    __shared__ __attribute__((aligned(16))) unsigned char As[BF_BM * 64];
    __shared__ __attribute__((aligned(16))) unsigned char Bs[BF_BN * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r32 = lane & 31, h = lane >> 5;
    const int64_t m0 = (int64_t)blockIdx.x * BF_BM;
    const int n0 = blockIdx.y * BF_BN;
    const int64_t k_begin = (int64_t)blockIdx.z * kslab;
    const int64_t k_end = (k_begin + kslab) < K ? (k_begin + kslab) : K;
    const bool vec4 = TRANS && (lda & 3) == 0 && (ldb & 3) == 0 && (((uintptr_t)a | (uintptr_t)b) & 15) == 0;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    for (int64_t k0 = k_begin; k0 < k_end; k0 += BF_BK) {
        __syncthreads();
        if (!TRANS) {
            // 128 rows x 32 k of each operand: a thread loads 4 float4s (row = id >> 3, k = 4 (id & 7))
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int id = tid + 256 * p, row = id >> 3, kq = (id & 7) * 4;
                float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vb = va;
                if (m0 + row < M && k0 + kq < k_end) va = *reinterpret_cast<const float4*>(a + (m0 + row) * lda + k0 + kq);
                if (n0 + row < N && k0 + kq < k_end) vb = *reinterpret_cast<const float4*>(b + (int64_t)(n0 + row) * ldb + k0 + kq);
                const bf16x4 ha = {(__bf16)va.x, (__bf16)va.y, (__bf16)va.z, (__bf16)va.w};
                const bf16x4 hb = {(__bf16)vb.x, (__bf16)vb.y, (__bf16)vb.z, (__bf16)vb.w};
                *reinterpret_cast<bf16x4*>(As + bf_lds_off(row, kq)) = ha;
                *reinterpret_cast<bf16x4*>(Bs + bf_lds_off(row, kq)) = hb;
            }
        } else {
            // 32 k-rows x 128 columns: a thread loads 4 float4s along the contiguous axis and scatters them transposed
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int id = tid + 256 * p, kr = id >> 5, mq = (id & 31) * 4;
                const bool kok = k0 + kr < k_end;
                float va[4] = {0.f, 0.f, 0.f, 0.f}, vb[4] = {0.f, 0.f, 0.f, 0.f};
                if (kok && vec4 && m0 + mq + 3 < M) {
                    const float4 t = *reinterpret_cast<const float4*>(a + (k0 + kr) * lda + m0 + mq);
                    va[0] = t.x; va[1] = t.y; va[2] = t.z; va[3] = t.w;
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u)               // (element-wise: the column counts need not be multiples of 4)
                        if (kok && m0 + mq + u < M) va[u] = a[(k0 + kr) * lda + m0 + mq + u];
                }
                if (kok && vec4 && n0 + mq + 3 < N) {
                    const float4 t = *reinterpret_cast<const float4*>(b + (k0 + kr) * ldb + n0 + mq);
                    vb[0] = t.x; vb[1] = t.y; vb[2] = t.z; vb[3] = t.w;
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (kok && n0 + mq + u < N) vb[u] = b[(k0 + kr) * ldb + n0 + mq + u];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    *reinterpret_cast<__bf16*>(As + bf_lds_off(mq + u, kr)) = (__bf16)va[u];
                    *reinterpret_cast<__bf16*>(Bs + bf_lds_off(mq + u, kr)) = (__bf16)vb[u];
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int k16 = 0; k16 < 2; ++k16) {
            bf16x8 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(As + bf_lds_off(wm * 64 + i * 32 + r32, k16 * 16 + 8 * h));
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(Bs + bf_lds_off(wn * 64 + j * 32 + r32, k16 * 16 + 8 * h));
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
    }
    // accumulator layout: lane = column (n), register e = row 8 (e >> 2) + 4 h + (e & 3) of the 32 x 32 block
    float* cz = c + (int64_t)blockIdx.z * slab_stride;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + r32;
        if (n >= N) continue;
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t m = m0 + wm * 64 + i * 32 + 8 * (e >> 2) + 4 * h + (e & 3);
                if (m < M) cz[m * ldc + n] = __fadd_rn(acc[i][j][e], bv);
            }
    }
}

// out[i] = sum over slabs (ascending) of part[s][i]
__global__ __launch_bounds__(256) void slab_sum_kernel(const float* __restrict__ part, int slabs, int64_t count, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    float s = 0.f;
    for (int z = 0; z < slabs; ++z) s = __fadd_rn(s, part[(int64_t)z * count + i]);
    out[i] = s;
}

// column sums of dy [rows, n] in f32 (the bias gradient: not a product, no rounding to bf16): a workgroup sums 64 columns of
// one row slab (4 row lanes, fixed order); the slab partials are added in slab order by slab_sum_kernel
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ dy, int ld, int64_t rows, int n, int64_t slab,
                                                     float* __restrict__ part_out) {
    __shared__ float part[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + cl;
    const int64_t r0 = (int64_t)blockIdx.y * slab, r1 = (r0 + slab) < rows ? (r0 + slab) : rows;
    float s = 0.f;
    if (col < n)
        for (int64_t r = r0 + rl; r < r1; r += 4) s = __fadd_rn(s, dy[r * ld + col]);
    part[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && col < n)
        part_out[(int64_t)blockIdx.y * n + col] = __fadd_rn(__fadd_rn(part[0][cl], part[1][cl]), __fadd_rn(part[2][cl], part[3][cl]));
}

constexpr int64_t BF_WGRAD_SLAB = 512;       // rows per weight-gradient slab (these launches are small: more slabs = more workgroups)
constexpr int64_t BF_COLSUM_SLAB = 256;      // rows per bias-gradient slab

}  // namespace sapcu

using namespace sapcu;

extern "C" {

int sapcu_gemm_bf16(const float* a, int64_t r, int k, int lda, const float* w, int n, const float* bias, float* c, int ldc,
                    void* stream) {
    SAPCU_CHECK_ARG(a && w && c && r >= 0 && n >= 1 && k >= 1, "gemm_bf16: bad argument");
    SAPCU_CHECK_ARG(k % 4 == 0 && lda % 4 == 0 && lda >= k && ldc >= n && (((uintptr_t)a | (uintptr_t)w) & 15) == 0,
                    "gemm_bf16: k and lda must be multiples of 4, operands 16-byte aligned");
    if (r == 0) return SAPCU_OK;
    const dim3 grid((unsigned)((r + BF_BM - 1) / BF_BM), (unsigned)((n + BF_BN - 1) / BF_BN), 1);
    hipLaunchKernelGGL((gemm_bf16_kernel<false>), grid, dim3(256), 0, (hipStream_t)stream, a, lda, w, k, r, n, (int64_t)k, (int64_t)k,
                       bias, c, (int64_t)ldc, (int64_t)0);
    SAPCU_CHECK_LAUNCH();
    return SAPCU_OK;
}

int64_t sapcu_wgrad_bf16_workspace_bytes(int64_t rows, int n, int k) {
    if (rows < 0 || n < 1 || k < 1) return SAPCU_ERR_ARG;
    const int64_t slabs = (rows + BF_WGRAD_SLAB - 1) / BF_WGRAD_SLAB, cslabs = (rows + BF_COLSUM_SLAB - 1) / BF_COLSUM_SLAB;
    return (slabs > 1 ? slabs * (int64_t)n * k * 4 : 0) + cslabs * (int64_t)n * 4 + 512;
}

int sapcu_conv1x1_wgrad_bf16(const float* grad_y, int ldy, const float* x, int ldx, int64_t rows, int n, int k, float* grad_w,
                             float* grad_bias, void* workspace, int64_t workspace_bytes, void* stream) {
    SAPCU_CHECK_ARG(grad_y && x && grad_w && rows >= 0 && n >= 1 && k >= 1 && ldy >= n && ldx >= k, "wgrad_bf16: bad argument");
    if (rows == 0) {        // an empty batch: zero gradients (sapcu_gemm_bf16 accepts r == 0 too, so forward and backward agree)
        SAPCU_CHECK_HIP(hipMemsetAsync(grad_w, 0, (size_t)n * k * sizeof(float), (hipStream_t)stream));
        if (grad_bias) SAPCU_CHECK_HIP(hipMemsetAsync(grad_bias, 0, (size_t)n * sizeof(float), (hipStream_t)stream));
        return SAPCU_OK;
    }
    const int64_t slabs = (rows + BF_WGRAD_SLAB - 1) / BF_WGRAD_SLAB;
    const int64_t cslabs_chk = (rows + BF_COLSUM_SLAB - 1) / BF_COLSUM_SLAB;
    SAPCU_CHECK_ARG(slabs <= 65535 && cslabs_chk <= 65535, "wgrad_bf16: %lld rows exceed the grid limit (65535 slabs of %d rows)",
                    (long long)rows, (int)BF_WGRAD_SLAB);
    if (workspace_bytes < sapcu_wgrad_bf16_workspace_bytes(rows, n, k) || !workspace) {
        set_error("wgrad_bf16: workspace %lld B < required %lld B", (long long)workspace_bytes,
                  (long long)sapcu_wgrad_bf16_workspace_bytes(rows, n, k));
        return SAPCU_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    float* part = slabs > 1 ? reinterpret_cast<float*>(((uintptr_t)workspace + 255) & ~(uintptr_t)255) : grad_w;
    const dim3 grid((unsigned)((n + BF_BM - 1) / BF_BM), (unsigned)((k + BF_BN - 1) / BF_BN), (unsigned)slabs);
    // dw[n, k]: "A" = grad_y columns (m = output channel), "B" = x columns (n = input channel), reduction over the rows
    hipLaunchKernelGGL((gemm_bf16_kernel<true>), grid, dim3(256), 0, st, grad_y, ldy, x, ldx, (int64_t)n, k, rows, BF_WGRAD_SLAB,
                       (const float*)nullptr, part, (int64_t)k, (int64_t)n * k);
    SAPCU_CHECK_LAUNCH();
    if (slabs > 1) {
        const int64_t count = (int64_t)n * k;
        hipLaunchKernelGGL(slab_sum_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, part, (int)slabs, count, grad_w);
        SAPCU_CHECK_LAUNCH();
    }
    if (grad_bias) {
        const int64_t cslabs = (rows + BF_COLSUM_SLAB - 1) / BF_COLSUM_SLAB;
        float* cpart = reinterpret_cast<float*>(((uintptr_t)workspace + 255) & ~(uintptr_t)255) + (slabs > 1 ? slabs * (int64_t)n * k : 0);
        hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)((n + 63) / 64), (unsigned)cslabs), dim3(256), 0, st, grad_y, ldy, rows, n,
                           BF_COLSUM_SLAB, cpart);
        SAPCU_CHECK_LAUNCH();
        hipLaunchKernelGGL(slab_sum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, cpart, (int)cslabs, (int64_t)n, grad_bias);
        SAPCU_CHECK_LAUNCH();
    }
    return SAPCU_OK;
}

}  // extern "C"
