// Split-f16 MFMA GEMM ("3xf16"): f32-quality products on the 16x faster f16 matrix pipe (gfx950).
//
//   C[r, n] = epi( A[r, k] . W[n, k]^T + bias[n] ),   A, W, C in f32 memory
//
// Every f32 operand x is split into two halves  x = hi + lo  with hi = f16_rn(x), lo = f16_rn(x - hi):
// two round-to-nearest steps of 12 bits each leave |x - (hi + lo)| <= 2^-24 |x| (f32's own half-ulp) as
// long as lo stays a normal f16, and <= 2^-25 absolute when it is subnormal (|x| < 0.25).  A product
// then needs three f16 MFMAs into ONE f32 accumulator,
//        x.w = hi_x hi_w + hi_x lo_w + lo_x hi_w                  (dropped: lo_x lo_w <= 2^-24 |x w|)
// (v_mfma_f32_32x32x16_f16: products of two f16 are exact in f32, accumulation is f32; the order inside a k32 step is the library's
// pass-major one — lo_x hi_w over both k16 halves, then hi_x lo_w, then hi_x hi_w —, see the k loop).  Weights are
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
// fill the vector pipe (one wave alone issues at half rate).
//   * k-step = 64: 12 f16 MFMAs x 4 k16 per producer wave = 1536 matrix-pipe cycles per SIMD between
//     barriers — long enough to cover the operand fetch of the next step and one 4-element neuron group
//     of each consumer wave (K = 512: 8 k-steps, 8 groups: balanced);
//   * W (pre-split f16 hi/lo planes) goes global -> LDS by LDS-DMA (global_load_lds_dwordx4, no
//     registers, no VALU); A rows are f32 in memory: staged through registers and split to hi/lo on the
//     way (1-step-ahead prefetch);
//   * operand tiles are [row][64 halves] = 128-byte rows with the 16-byte chunk index XOR-swizzled by
//     (row>>1)&7: conflict-free ds_read_b128 fragments without padding, which LDS-DMA requires
//     (its destination is lane-linear; the swizzle is applied to the per-lane SOURCE address);
//   * LDS: 2 stages x 64 KiB operands + 32 KiB.  The accumulator hand-off is 64 KiB: the half for column
//     tile 0 lives in the spare 32 KiB, the half for column tile 1 ALIASES stage 1 — consumers copy those
//     16 values to registers right after the hand-off barrier, before stage 1 is refilled at k-step 0.
#include "common.h"
#include "gemm_epi.h"

namespace sapcu {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) unsigned char lds_byte;
typedef const __attribute__((address_space(1))) void* gptr_t;

constexpr int SBM = 128, SBN = 128, SBK = 64;
constexpr int PLANE_BYTES = SBM * SBK * 2;           // one [128][64] f16 plane = 16 KiB
constexpr int STAGE_BYTES = 4 * PLANE_BYTES;         // A_hi | A_lo | W_hi | W_lo = 64 KiB
constexpr int HALF_HANDOFF_BYTES = 8 * 16 * 64 * 4;   // 8 producer waves x 16 results x 64 lanes = 32 KiB
constexpr int SF16_LDS_BYTES = 2 * STAGE_BYTES + HALF_HANDOFF_BYTES;   // column-tile-1 half of the hand-off aliases stage 1

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

// consumer: one 4-element group of column tile 1 with STATIC register indices (Q = group - 4)
template <int EPI, int Q>
__device__ __forceinline__ void consume_reg_group(const GemmArgs& g, const float (&cacc)[16], int64_t row, int col,
                                                  float bias, const NeuronP& np, const float (&cq)[4],
                                                  const float (&ckf)[4]) {
    const float a[4] = {cacc[Q * 4], cacc[Q * 4 + 1], cacc[Q * 4 + 2], cacc[Q * 4 + 3]};
    epilogue_group4<EPI>(g, a, row, col, bias, np, cq, ckf);
}

template <int EPI>
__global__ __launch_bounds__(1024) void gemm_sf16_kernel(const GemmArgs g) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem_raw[];
    float* stage1 = reinterpret_cast<float*>(smem_raw + STAGE_BYTES);        // hand-off, column tile 1: aliases stage 1
    float* stage0 = reinterpret_cast<float*>(smem_raw + 2 * STAGE_BYTES);    // hand-off, column tile 0: spare 32 KiB
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

    // The two roles are separate loops (separate register allocations) that execute the SAME sequence of
    // workgroup barriers: per tile 1 (hand-off copied out / k-step 0 staged) + nk (k-steps) + 1 (hand-off).
    if (producer) {
        // A: thread t (0..511) owns row t>>2 and 16-byte chunks 2*(t&3), 2*(t&3)+1 of both A planes
        const int srow = tid >> 2;
        const int sc0 = (tid & 3) * 2;
        const int ssw = (srow >> 1) & 7;
        // W: wave w issues LDS-DMA pieces (w*2 + i), i = 0,1, of each W plane; lane -> chunk p = piece*64 + lane
        int wrow_[2], wsrc_[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int p = (pw * 2 + i) * 64 + lane;
            wrow_[i] = p >> 3;
            wsrc_[i] = ((p & 7) ^ ((wrow_[i] >> 1) & 7)) * 8;      // source k offset (halves) of this LDS chunk
        }
        // fragment rows / swizzles
        const int arow_l = wm * 32 + r32;
        const int asw = (arow_l >> 1) & 7;
        int wrow_l[2], wsw[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            wrow_l[j] = wn * 64 + j * 32 + r32;
            wsw[j] = (wrow_l[j] >> 1) & 7;
        }
        float4 ra[4];
        f32x16 acc[2];
        bool prev = false;
        for (int64_t it = 0;; ++it) {
            const int64_t local = it * wgs_per_x + wg_in_x;
            const bool have = local < x_count;
            if (!have && !prev) break;
            const int64_t logical = x_begin + local;
            const int64_t row0 = have ? (logical / ntn) * SBM : 0;
            const int col0 = have ? (int)(logical % ntn) * SBN : 0;
            int64_t arow_g = row0 + srow;
            if (arow_g >= g.r) arow_g = g.r - 1;                  // clamped rows only feed masked outputs
            const float* aptr = g.a + arow_g * g.lda + sc0 * 8;
            int woff[2];   // element offset of this lane's source chunk inside both W planes (n*k < 2^31)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int nn = col0 + wrow_[i];
                if (nn >= g.n) nn = g.n - 1;
                woff[i] = nn * g.k + wsrc_[i];
            }
            float amax = 0.f;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
            auto load_a = [&](int k0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) ra[i] = ld4(aptr + k0 + 4 * i);
            };
            auto dma_w = [&](int buf, int k0) {
                lds_byte* sbase = (lds_byte*)(smem_raw + buf * STAGE_BYTES + 2 * PLANE_BYTES);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    lds_byte* dst = sbase + (pw * 2 + i) * 1024;
                    __builtin_amdgcn_global_load_lds((gptr_t)(g.w16_hi + woff[i] + k0), dst, 16, 0, 0);
                    __builtin_amdgcn_global_load_lds((gptr_t)(g.w16_lo + woff[i] + k0), dst + PLANE_BYTES, 16, 0, 0);
                }
            };
            auto store_a = [&](int buf) {
                unsigned char* base = smem_raw + buf * STAGE_BYTES + srow * (SBK * 2);
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    half8 ah, al;
                    split8(ra[2 * c], ra[2 * c + 1], ah, al, amax);
                    const int phys = ((sc0 + c) ^ ssw) * 16;
                    *reinterpret_cast<half8*>(base + phys) = ah;
                    *reinterpret_cast<half8*>(base + PLANE_BYTES + phys) = al;
                }
            };
            if (have) {
                dma_w(0, 0);
                load_a(0);
                store_a(0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            lds_barrier();
            int cur = 0;
            for (int kt = 0; kt < nk; ++kt) {
                if (have) {
                    if (kt + 1 < nk) {
                        dma_w(cur ^ 1, (kt + 1) * SBK);
                        load_a((kt + 1) * SBK);
                    }
                    const unsigned char* st = smem_raw + cur * STAGE_BYTES;
                    const unsigned char* sA = st + arow_l * (SBK * 2);
                    // pass-major order per k32 step (the library's accumulation order since round 3, gemm_sf16_bt.hip): a_lo.w_hi over
                    // both k16 halves, then a_hi.w_lo, then a_hi.w_hi
#pragma unroll
                    for (int k32 = 0; k32 < SBK / 32; ++k32) {
                        // one product at a time, its six fragments read just before it (w_hi is read twice: holding all twelve
                        // fragments of a k32 step cost 12 more spilled registers at this kernel's 128 and 10 % of its speed)
                        auto product = [&](int a_plane, int w_plane) {
                            half8 a2[2], w2[2][2];
#pragma unroll
                            for (int t = 0; t < 2; ++t) {
                                const int k16 = 2 * k32 + t;
                                a2[t] = *reinterpret_cast<const half8*>(sA + a_plane * PLANE_BYTES + ((k16 * 2 + h) ^ asw) * 16);
#pragma unroll
                                for (int j = 0; j < 2; ++j)
                                    w2[t][j] = *reinterpret_cast<const half8*>(st + (2 + w_plane) * PLANE_BYTES + wrow_l[j] * (SBK * 2) +
                                                                               (((k16 * 2 + h) ^ wsw[j]) * 16));
                            }
#pragma unroll
                            for (int t = 0; t < 2; ++t)
#pragma unroll
                                for (int j = 0; j < 2; ++j) acc[j] = mfma_32x32x16_f16(a2[t], w2[t][j], acc[j]);
                            __builtin_amdgcn_sched_barrier(0);
                        };
                        product(1, 0);      // a_lo . w_hi
                        product(0, 1);      // a_hi . w_lo
                        product(0, 0);      // a_hi . w_hi
                    }
                    if (kt + 1 < nk) {
                        store_a(cur ^ 1);
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the LDS-DMA pieces of the next stage have landed
                    }
                }
                lds_barrier();
                cur ^= 1;
            }
            // hand-off (into stage 1's bytes: every operand read of this tile is behind the last barrier)
            if (have) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    stage0[(pw * 16 + e) * 64 + lane] = __fmul_rn(acc[0][e], 0.0625f);   // undo W x 16
                    stage1[(pw * 16 + e) * 64 + lane] = __fmul_rn(acc[1][e], 0.0625f);
                }
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
            const bool cons_work = prev_row0 >= 0;
            // ---- copy the column-tile-1 half out of the hand-off area that aliases stage 1 (refilled at k-step 0)
            float cacc[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) cacc[q] = cons_work ? stage1[(pw * 16 + q) * 64 + lane] : 0.f;
            float cbias[2];
            NeuronP cnp[2];
            bool ccol[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = prev_col0 + wn * 64 + j * 32 + r32;
                ccol[j] = cons_work && col < g.n;
                cbias[j] = settle((ccol[j] && g.bias) ? g.bias[col] : 0.f);
                cnp[j] = NeuronP{0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if (EPI == EPI_LIF || EPI == EPI_LIF_ATTN || EPI == EPI_LIF_MAX) {
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
                        float cq[4], ckf[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            cq[u] = nq[u];
                            ckf[u] = nkf[u];
                        }
                        if (gi + 1 < 8) issue_gather(gi + 1);
                        const int j = gi >> 2, e4 = gi & 3;
                        if (!(j ? ccol[1] : ccol[0])) continue;
                        const int64_t row = prev_row0 + wm * 32 + 8 * e4 + 4 * h;
                        if (row >= g.r) continue;
                        const int col = prev_col0 + wn * 64 + j * 32 + r32;
                        if (gi < 4) {       // column tile 0: straight from the non-aliased half of the hand-off
                            float a[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) a[u] = stage0[(pw * 16 + gi * 4 + u) * 64 + lane];
                            epilogue_group4<EPI>(g, a, row, col, cbias[0], cnp[0], cq, ckf);
                        } else {            // column tile 1: from registers, static indices per group
                            switch (gi) {
                                case 4: consume_reg_group<EPI, 0>(g, cacc, row, col, cbias[1], cnp[1], cq, ckf); break;
                                case 5: consume_reg_group<EPI, 1>(g, cacc, row, col, cbias[1], cnp[1], cq, ckf); break;
                                case 6: consume_reg_group<EPI, 2>(g, cacc, row, col, cbias[1], cnp[1], cq, ckf); break;
                                default: consume_reg_group<EPI, 3>(g, cacc, row, col, cbias[1], cnp[1], cq, ckf); break;
                            }
                        }
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

template <int EPI>
static int launch_t16(const GemmArgs& g, hipStream_t st) {
    static DeviceOnce lds_once;                         // one per kernel instantiation, one bit per device
    SAPCU_SET_MAX_LDS(lds_once, (&gemm_sf16_kernel<EPI>), SF16_LDS_BYTES);
    const int g_num_cus16 = device_cu_count();
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
        case EPI_LRELU_MAX:
            SAPCU_CHECK_ARG(g.max_keys && g.max_m >= 1, "gemm_sf16: EPI_LRELU_MAX needs max_keys and max_m");
            return launch_t16<EPI_LRELU_MAX>(g, st);
        case EPI_LIF_MAX:
            SAPCU_CHECK_ARG(g.max_keys && g.max_m >= 1 && g.lif, "gemm_sf16: EPI_LIF_MAX needs max_keys, max_m and neuron parameters");
            return launch_t16<EPI_LIF_MAX>(g, st);
        case EPI_RESID_GELU: return launch_t16<EPI_RESID_GELU>(g, st);
        case EPI_LIF_ATTN:
            SAPCU_CHECK_ARG(g.ldq > 0 && g.tab && g.q && g.kf && g.c2, "gemm_sf16: bad attn operands");
            return launch_t16<EPI_LIF_ATTN>(g, st);
        default: set_error("gemm_sf16: unknown epilogue %d", g.epi); return SAPCU_ERR_ARG;
    }
}

}  // namespace sapcu
