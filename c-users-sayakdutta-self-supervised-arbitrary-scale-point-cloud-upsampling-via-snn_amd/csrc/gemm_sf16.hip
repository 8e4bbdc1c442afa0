// Split-f16 MFMA GEMM ("3xf16"): f32-quality products on the 16x faster f16 matrix pipe (gfx950).
//
//   C[r, n] = epi( A[r, k] . W[n, k]^T + bias[n] ),   A, W, C in f32 memory
//
// Every f32 operand x is split into two halves  x = hi + lo  with hi = f16_rn(x), lo = f16_rn(x - hi):
// two round-to-nearest steps of 12 bits each leave |x - (hi + lo)| <= 2^-24 |x| (f32's own half-ulp) as
// long as lo stays a normal f16, and <= 2^-25 absolute when it is subnormal (|x| < 0.25).  A product
// then needs three f16 MFMAs into ONE f32 accumulator,
//        x.w = hi_x hi_w + hi_x lo_w + lo_x hi_w                  (dropped: lo_x lo_w <= 2^-24 |x w|)
// (v_mfma_f32_32x32x16_f16: products of two f16 are exact in f32, accumulation is f32).  Weights are
// pre-multiplied by 16 when they are split (model build), so that lo_w is normal for every |w| >= 2^-6,
// and the accumulator is scaled back by 1/16 (exact) at hand-off.  Measured against an f64 reference at
// K = 512 this is as accurate as the f32 MFMA's k-sequential fmaf chain, at 16/3 = 5.3x its rate.
// Range: |x| < 65504, |w| < 4094; weights are checked when the model is built, activations by a
// per-tile max test that raises the model's overflow counter (sapcu_model_gemm_mode).
//
// Structure (same producer/consumer idea as gemm_f32.hip, re-balanced because the k-loop is now short
// and the neuron epilogue is the long pole): ONE 1024-thread workgroup per CU, persistent over 128x128
// tiles; waves 0-7 are MFMA PRODUCERS (4x2, each a 32x64 sub-tile = 2 MFMA tiles, 32 accumulator
// registers), waves 8-15 are EPILOGUE CONSUMERS — two VALU waves per SIMD, which is what it takes to
// fill the vector pipe (one wave alone issues at half rate).  Producers convert the f32 A rows to
// hi/lo while staging them into LDS; W is pre-split at model build.  LDS: 2 x 40 KiB operand stages +
// 64 KiB accumulator hand-off = 144 KiB.  128 VGPRs per wave (4 waves per SIMD).
#include "common.h"
#include "gemm_epi.h"

namespace sapcu {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

constexpr int SBM = 128, SBN = 128, SBK = 32;
constexpr int LDH = SBK + 8;                         // halves per LDS row (80 B: conflict-free ds_read_b128)
constexpr int OPER_HALVES = 4 * SBM * LDH;           // A_hi | A_lo | W_hi | W_lo of one stage
constexpr int SF16_STAGE_FLOATS = 8 * 32 * 64;       // 8 producer waves x 32 results x 64 lanes
constexpr int SF16_LDS_BYTES = 2 * OPER_HALVES * 2 + SF16_STAGE_FLOATS * 4;

__device__ __forceinline__ void split8(const float4& x0, const float4& x1, half8& hi, half8& lo, float& amax) {
    const float xs[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const _Float16 h = (_Float16)xs[i];
        hi[i] = h;
        lo[i] = (_Float16)(xs[i] - (float)h);
        amax = fmaxf(amax, fabsf(xs[i]));
    }
}

template <int EPI>
__global__ __launch_bounds__(1024) void gemm_sf16_kernel(const GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    _Float16* oper = reinterpret_cast<_Float16*>(smem_raw);
    float* stage = reinterpret_cast<float*>(smem_raw + 2 * OPER_HALVES * 2);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const bool producer = wave < 8;
    const int pw = wave & 7;                 // producer index / the producer this consumer serves
    const int wm = pw >> 1, wn = pw & 1;     // 4 x 2 sub-tiles of 32 x 64
    const int r32 = lane & 31, h = lane >> 5;

    const int ntn = (g.n + SBN - 1) / SBN;
    const int64_t ntm = (g.r + SBM - 1) / SBM;
    const int64_t ntiles = ntm * ntn;
    const int nx = gridDim.x < 8 ? 1 : 8;
    const int xcd = nx == 1 ? 0 : (int)(blockIdx.x & 7);
    const int wg_in_x = nx == 1 ? (int)blockIdx.x : (int)(blockIdx.x >> 3);
    const int wgs_per_x = nx == 1 ? (int)gridDim.x : (int)((gridDim.x - xcd + 7) >> 3);
    const int64_t qd = ntiles / nx, rem = ntiles % nx;
    const int64_t x_begin = xcd * qd + (xcd < rem ? xcd : rem);
    const int64_t x_count = qd + (xcd < rem ? 1 : 0);
    const int nk = g.k / SBK;

    // The two roles are separate loops (separate register allocations: the producers' accumulators and
    // the consumers' neuron state never coexist) that execute the SAME sequence of workgroup barriers:
    // per tile 1 (operands of k-step 0 staged) + nk (one per k-step) + 1 (accumulators handed off).
    if (producer) {
        // thread t (0..511) owns row t>>2, k-octet (t&3)*8 of the A and W tiles
        const int srow = tid >> 2;
        const int skc = (tid & 3) * 8;
        float4 ra0, ra1;
        half8 rwh, rwl;
        f32x16 acc[2];
        bool prev = false;
        for (int64_t it = 0;; ++it) {
            const int64_t local = it * wgs_per_x + wg_in_x;
            const bool have = local < x_count;
            if (!have && !prev) break;
            const int64_t logical = x_begin + local;
            const int64_t row0 = have ? (logical / ntn) * SBM : 0;
            const int col0 = have ? (int)(logical % ntn) * SBN : 0;
            const int64_t row = row0 + srow;
            const bool aok = have && row < g.r;
            const float* arow = g.a + (aok ? row : 0) * g.lda + skc;
            const int nn = col0 + srow;
            const bool wok = have && nn < g.n;
            const _Float16* whrow = g.w16_hi + (int64_t)(wok ? nn : 0) * g.k + skc;
            const _Float16* wlrow = g.w16_lo + (int64_t)(wok ? nn : 0) * g.k + skc;
            float amax = 0.f;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
            auto load_tile = [&](int k0) {
                const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                ra0 = aok ? ld4(arow + k0) : z;
                ra1 = aok ? ld4(arow + k0 + 4) : z;
                if (wok) {
                    rwh = *reinterpret_cast<const half8*>(whrow + k0);
                    rwl = *reinterpret_cast<const half8*>(wlrow + k0);
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        rwh[i] = (_Float16)0.f;
                        rwl[i] = (_Float16)0.f;
                    }
                }
            };
            auto store_tile = [&](int buf) {
                _Float16* base = oper + buf * OPER_HALVES + srow * LDH + skc;
                half8 ah, al;
                split8(ra0, ra1, ah, al, amax);
                *reinterpret_cast<half8*>(base) = ah;
                *reinterpret_cast<half8*>(base + SBM * LDH) = al;
                *reinterpret_cast<half8*>(base + 2 * SBM * LDH) = rwh;
                *reinterpret_cast<half8*>(base + 3 * SBM * LDH) = rwl;
            };
            if (have) {
                load_tile(0);
                store_tile(0);
            }
            lds_barrier();
            int cur = 0;
            for (int kt = 0; kt < nk; ++kt) {
                if (have) {
                    if (kt + 1 < nk) load_tile((kt + 1) * SBK);
                    const _Float16* sA = oper + cur * OPER_HALVES + (wm * 32 + r32) * LDH + h * 8;
                    const _Float16* sW = oper + cur * OPER_HALVES + 2 * SBM * LDH + (wn * 64 + r32) * LDH + h * 8;
#pragma unroll
                    for (int k16 = 0; k16 < SBK / 16; ++k16) {
                        const half8 ah = *reinterpret_cast<const half8*>(sA + k16 * 16);
                        const half8 al = *reinterpret_cast<const half8*>(sA + SBM * LDH + k16 * 16);
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const half8 wh = *reinterpret_cast<const half8*>(sW + j * 32 * LDH + k16 * 16);
                            const half8 wl = *reinterpret_cast<const half8*>(sW + SBM * LDH + j * 32 * LDH + k16 * 16);
                            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wh, acc[j], 0, 0, 0);
                            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wl, acc[j], 0, 0, 0);
                            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wh, acc[j], 0, 0, 0);
                        }
                    }
                    if (kt + 1 < nk) store_tile(cur ^ 1);
                }
                lds_barrier();
                cur ^= 1;
            }
            // hand-off: the consumers drained the staging area before the last k-step barrier
            if (have) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        stage[(pw * 32 + j * 16 + e) * 64 + lane] = __fmul_rn(acc[j][e], 0.0625f);   // undo W x 16
                if (amax > 65504.0f && g.ovf) atomicAdd(g.ovf, 1);
            }
            lds_barrier();
            prev = have;
        }
    } else {
        int64_t prev_row0 = -1;
        int prev_col0 = 0;
        const int gper = (8 + nk - 1) / nk;
        for (int64_t it = 0;; ++it) {
            const int64_t local = it * wgs_per_x + wg_in_x;
            const bool have = local < x_count;
            if (!have && prev_row0 < 0) break;
            const int64_t logical = x_begin + local;
            // ---- state for the previous tile (32 results per lane = 8 groups of 4)
            const bool cons_work = prev_row0 >= 0;
            float cbias[2];
            NeuronP cnp[2];
            bool ccol[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = prev_col0 + wn * 64 + j * 32 + r32;
                ccol[j] = cons_work && col < g.n;
                cbias[j] = settle((ccol[j] && g.bias) ? g.bias[col] : 0.f);
                if (EPI == EPI_LIF || EPI == EPI_LIF_ATTN) {
                    cnp[j] = load_lif(g.lif, g.n, ccol[j] ? col : 0);
                    cnp[j].theta0 = settle(cnp[j].theta0);
                }
            }
            // EPI_LIF_ATTN: lane l holds the (q row, k row) pair of tile row wm*32 + (l & 31) (one 8-byte
            // load per tile); a group fetches its four pairs with ds_bpermute and its q/k gathers are issued
            // one group AHEAD, so their latency hides behind the previous group's neuron loop.
            int2 tabrow = make_int2(0, 0);
            float nq[4] = {0.f, 0.f, 0.f, 0.f}, nkf[4] = {0.f, 0.f, 0.f, 0.f};
            auto issue_gather = [&](int gi) {      // group gi: column tile j = gi>>2, rows 8*(gi&3) + 4h + 0..3
                if (EPI != EPI_LIF_ATTN) return;
                const int j = gi >> 2, e4 = gi & 3;
                const int col = prev_col0 + wn * 64 + j * 32 + r32;
                const int lrow = 8 * e4 + 4 * h;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int qr = __shfl(tabrow.x, lrow + u);
                    const int kr = __shfl(tabrow.y, lrow + u);
                    const bool ok = col < g.n && (prev_row0 + wm * 32 + lrow + u) < g.r;
                    nq[u] = ok ? g.q[(int64_t)qr * g.ldq + col] : 0.f;
                    nkf[u] = ok ? g.kf[(int64_t)kr * g.ldq + col] : 0.f;
                }
            };
            if (cons_work && EPI == EPI_LIF_ATTN) {
                const int64_t trow = prev_row0 + wm * 32 + (lane & 31);
                if (trow < g.r) tabrow = g.tab[trow];
                tabrow.x = __builtin_bit_cast(int, settle(__builtin_bit_cast(float, tabrow.x)));
                tabrow.y = __builtin_bit_cast(int, settle(__builtin_bit_cast(float, tabrow.y)));
                issue_gather(0);
            }
            lds_barrier();
            for (int kt = 0; kt < nk; ++kt) {
                if (cons_work) {
                    for (int gi = kt * gper; gi < (kt + 1) * gper && gi < 8; ++gi) {
                        const int j = gi >> 2, e4 = gi & 3;
                        float cq[4], ckf[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            cq[u] = nq[u];
                            ckf[u] = nkf[u];
                        }
                        if (gi + 1 < 8) issue_gather(gi + 1);
                        if (!(j ? ccol[1] : ccol[0])) continue;
                        const int64_t row = prev_row0 + wm * 32 + 8 * e4 + 4 * h;
                        if (row >= g.r) continue;
                        float a[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) a[u] = stage[(pw * 32 + gi * 4 + u) * 64 + lane];
                        NeuronP np;   // select, don't index: runtime-indexed register arrays would go to scratch
                        np.decay = j ? cnp[1].decay : cnp[0].decay;
                        np.adapt = j ? cnp[1].adapt : cnp[0].adapt;
                        np.rdecay = j ? cnp[1].rdecay : cnp[0].rdecay;
                        np.theta0 = j ? cnp[1].theta0 : cnp[0].theta0;
                        np.dT = 0.f;
                        np.rh = 0.f;
                        epilogue_group4<EPI>(g, a, row, prev_col0 + wn * 64 + j * 32 + r32, j ? cbias[1] : cbias[0], np, cq,
                                             ckf);
                    }
                }
                lds_barrier();
            }
            lds_barrier();
            prev_row0 = have ? (logical / ntn) * SBM : -1;
            prev_col0 = have ? (int)(logical % ntn) * SBN : 0;
        }
    }
}

// W [n*k] f32 -> hi/lo f16 of 16*w; flags |16 w| >= 65504
__global__ __launch_bounds__(256) void split_weights_kernel(const float* __restrict__ w, int64_t count,
                                                            _Float16* __restrict__ hi, _Float16* __restrict__ lo,
                                                            int* __restrict__ ovf) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const float x = w[t] * 16.0f;
    const _Float16 hh = (_Float16)x;
    hi[t] = hh;
    lo[t] = (_Float16)(x - (float)hh);
    if (!(fabsf(x) < 65504.0f)) atomicAdd(ovf, 1);
}

int launch_split_weights(const float* w, int64_t count, void* hi, void* lo, int* ovf, hipStream_t st) {
    if (count == 0) return SAPCU_OK;
    hipLaunchKernelGGL(split_weights_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, w, count,
                       (_Float16*)hi, (_Float16*)lo, ovf);
    SAPCU_CHECK_LAUNCH();
    return SAPCU_OK;
}

static int g_num_cus16 = 0;

template <int EPI>
static int launch_t16(const GemmArgs& g, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        SAPCU_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_sf16_kernel<EPI>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, SF16_LDS_BYTES));
        attr_set = true;
    }
    if (g_num_cus16 == 0) {
        int dev = 0;
        SAPCU_CHECK_HIP(hipGetDevice(&dev));
        hipDeviceProp_t prop;
        SAPCU_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
        g_num_cus16 = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int64_t tiles = ((g.r + SBM - 1) / SBM) * ((g.n + SBN - 1) / SBN);
    const int64_t grid = tiles < g_num_cus16 ? tiles : g_num_cus16;
    hipLaunchKernelGGL((gemm_sf16_kernel<EPI>), dim3((unsigned)grid), dim3(1024), SF16_LDS_BYTES, st, g);
    SAPCU_CHECK_LAUNCH();
    return SAPCU_OK;
}

int launch_gemm_sf16(const GemmArgs& g, hipStream_t st) {
    if (g.r == 0 || g.n == 0) return SAPCU_OK;
    SAPCU_CHECK_ARG(g.k > 0 && g.k % SBK == 0, "gemm_sf16: k=%d must be a positive multiple of %d", g.k, SBK);
    SAPCU_CHECK_ARG(g.lda % 4 == 0 && ((uintptr_t)g.a & 15) == 0 && g.w16_hi && g.w16_lo &&
                        ((uintptr_t)g.w16_hi & 15) == 0 && ((uintptr_t)g.w16_lo & 15) == 0,
                    "gemm_sf16: operands must be 16-byte aligned (lda=%d)", g.lda);
    switch (g.epi) {
        case EPI_BIAS: return launch_t16<EPI_BIAS>(g, st);
        case EPI_LIF: return launch_t16<EPI_LIF>(g, st);
        case EPI_GELU: return launch_t16<EPI_GELU>(g, st);
        case EPI_RESID: return launch_t16<EPI_RESID>(g, st);
        case EPI_LRELU: return launch_t16<EPI_LRELU>(g, st);
        case EPI_RESID_GELU: return launch_t16<EPI_RESID_GELU>(g, st);
        case EPI_LIF_ATTN:
            SAPCU_CHECK_ARG(g.ldq > 0 && g.tab && g.q && g.kf && g.c2, "gemm_sf16: bad attn operands");
            return launch_t16<EPI_LIF_ATTN>(g, st);
        default: set_error("gemm_sf16: unknown epilogue %d", g.epi); return SAPCU_ERR_ARG;
    }
}

}  // namespace sapcu
