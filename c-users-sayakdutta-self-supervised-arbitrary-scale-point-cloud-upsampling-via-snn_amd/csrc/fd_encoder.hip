// fd encoder: EnhancedTemporalSNN_DGCNN_fd.forward (fd/snn_coder.py:392-480) from the rotated patch to the per-step pooled
// features, in ONE kernel — one 512-thread workgroup per patch, the patch resident in LDS / registers from the first neighbour
// search to the last max over points:
//
//     block 0   xyz kNN (all scales are prefixes of one sorted list) -> multi-scale EdgeConv(6->64)+BN+LReLU, max over the
//               ks nearest -> scale_fusion (64 S -> 64)+BN+LReLU -> EIF                                     fd:411-444
//     block l   kNN in the feature space of block l-1's step-0 spikes -> EdgeConv(2C -> C')+BN+LReLU, max over the k
//     = 1..3    neighbours (factored: W.cat(xj - xi, xj) = (W1 + W2) xj - W1 xi, BN scale folded) -> EIF / LIF / LIF   fd:447-474
//     all t     cat(64 + 128 + 256 + 512 = 960) -> multi_scale_conv (960 -> emb)+BN+LReLU -> max over the points    fd:476-480
//
// The per-stage form (model.hip: ~20 launches) writes the [T, points, 960] spike tensor (3 GB at 4096 patches), the [points, 2C']
// EdgeConv operands and the block-0 features to HBM and reads them back.  Here nothing of the encoder reaches HBM: the kernel reads
// the patch (576 B), the packed weights (L2) and writes pooled [T, emb] per patch.
//
// What makes that fit.  In eval mode a neuron's input gate is closed from step 1 on (violations are counted),
// so ALL T spikes of a (point, channel) are a function of its ONE pre-activation x0 at t = 0.  The kernel therefore keeps x0 —
// 48 x 960 floats: blocks 0-2 in LDS (84 KiB), block 3 in registers (the current third of the patch parked in LDS, round 4) — and regenerates the spikes where they are
// consumed: step-0 spikes as the f32 features of the next block's neighbour search and as the split-f16 operand of its EdgeConv
// GEMM; all steps as the operand of multi_scale_conv.  That contraction takes the patch in thirds of 16 points with the T steps
// stacked as ROWS (row = 4 point + step: 64 rows, no padded row at T = 4), every wave 96 of ALL emb columns (96 accumulator
// registers: one sweep over K per third), K in rounds of 128 columns: per round every thread runs the T-step neuron loop of four
// (point, channel) elements with the state in registers, writes the spikes into a 32 KiB split-f16 panel and the waves multiply the
// panel with weight fragments streamed L2 -> registers in fragment order (the fn_edge_chain.hip recipe; v_mfma_f32_16x16x32_f16,
// pass-major, since round 3).  The max over the points is a running maximum of the raw accumulators (bias and LeakyReLU are
// monotone: once per result), kept in LDS between the thirds, like the GEMM epilogue it replaces.  Round 4: the
// production instantiation fd_encoder_kernel<false> (T = 4, no spike tap) needs 241 registers and no scratch; block 0's EdgeConv
// runs on the exact-f32 MFMA.
//
// Bit-identical to the per-stage path (tests/test_gpu_parity.py::test_fused_fd_encoder_equals_the_per_stage_path_bit_for_bit):
// same score chains and tie rule in the neighbour searches, same split-f16 products in the same k order, same neuron arithmetic
// (NeuronStep2 in every block), same max.  Taken for patches of <= 48 points, <= 4 scales, emb % 32 == 0 and
// emb >= 96 (fd_encoder_ok); anything else runs the per-stage kernels.
#include <type_traits>

#include "common.h"
#include "gemm_epi.h"
#include "ops.h"

namespace sapcu {
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// diagnostic build (-DFE_STAMPS, profiles/fd_stamps.py): s_memtime at the phase boundaries, wave 0 / lane 0, 32 slots per patch
// in the buffer passed as the spikes tap
#ifdef FE_STAMPS
#define FE_STAMP(IDX)                                                                                                   \
    do {                                                                                                                \
        if (a.tap_spikes && threadIdx.x == 0)                                                                           \
            reinterpret_cast<unsigned long long*>(a.tap_spikes)[(int64_t)blockIdx.x * 32 + (IDX)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define FE_STAMP(IDX) do { } while (0)
#endif

constexpr int FE_M = 48;                 // points per patch (rows of every per-patch table)
constexpr int FE_NT = 512;               // threads: 8 waves, 256 registers each
constexpr int FE_XLD = 448;              // x0 of blocks 0..2: [48][448] f32
// (the phase buffer comes FIRST: its addresses — panel rows, planes, steps, staging tiles — then fit the 16-bit offset field of the
//  LDS instructions and need no second base register per variant)
constexpr int FE_OFF_R2 = 0;                               // 64 KiB phase buffer (score keys / GEMM panels / staging)
constexpr int FE_R2_BYTES = 65536;
constexpr int FE_OFF_XS = FE_R2_BYTES;                     // x0 of blocks 0..2: [48][448] f32 = 86 016 B
constexpr int FE_OFF_XYZ = FE_OFF_XS + FE_M * FE_XLD * 4;  // float4[48]
constexpr int FE_OFF_IDX0 = FE_OFF_XYZ + FE_M * 16;        // u8[48][48]: xyz neighbours, ascending distance
constexpr int FE_OFF_IDXL = FE_OFF_IDX0 + FE_M * FE_M;     // u8[48][48]: feature-space neighbours of the current block
constexpr int FE_OFF_XX = FE_OFF_IDXL + FE_M * FE_M;       // float[64]
constexpr int FE_LDS = 163840;                             // all of it: the per-patch tables end at FE_OFF_XX + 256 = 157 184 B, and
                                                           // multi_scale_conv re-uses [FE_OFF_XYZ, 163 840) as FE_OFF_BEST
// multi_scale_conv's view of the LDS (the per-patch tables and the upper half of the phase buffer are dead by then):
constexpr int FE_OFF_X3L = FE_OFF_R2 + 32768;              // f32[16 values][512 threads]: block 3's x0 of the CURRENT third of the patch
constexpr int FE_OFF_BEST = FE_OFF_XYZ;                    // u32[6 column sub-blocks][4 steps][8 waves][16 lanes]: running maxima (keys)
static_assert(FE_OFF_BEST + 6 * 4 * 8 * 16 * 4 <= FE_LDS && FE_OFF_X3L + 16 * 512 * 4 <= FE_OFF_XYZ, "fd encoder: multi_scale_conv areas");
constexpr int FE_F_LD = 36;                                // score staging 2 x [64][36] f32 at R2 + 0, keys [m][m+1] behind them
constexpr int FE_KEYS_OFF = 2 * 64 * FE_F_LD * 4;         // (two staging buffers)
constexpr int FE_TLD = 36;                                 // EdgeConv staging tile of a wave: [48][36] f32 (32 channels + 4 pad)
constexpr int FE_TILE = FE_M * FE_TLD;
static_assert(8 * FE_TILE * 4 <= FE_R2_BYTES, "fd encoder: staging tiles");
static_assert(FE_OFF_XX + 256 <= FE_LDS && FE_LDS <= 163840, "fd encoder: LDS budget");
static_assert(FE_KEYS_OFF + FE_M * (FE_M + 1) * 4 <= FE_R2_BYTES, "fd encoder: score keys");

// ---- split-f16 operand panel: [k32 step][plane hi | lo][ROWS][32 halves], 16-byte chunks XOR-swizzled by (row >> 2) & 3 (the
// operand-slot layout of gemm_sf16_bt.hip / fn_edge_chain.hip: conflict-free ds_read_b128 fragments)
template <int ROWS>
__device__ __forceinline__ unsigned fe_panel_off(int row, int k) {
    return (unsigned)((k >> 5) * (ROWS * 128) + row * 64 + ((((k >> 3) & 3) ^ ((row >> 2) & 3)) << 4) + (k & 7) * 2);
}
// The value is made opaque first: with the producing FMA visible, the compiler folds "(f16) fma(a, b, c)" into v_fma_mixlo_f16 —
// ONE rounding of the exact result to f16 — where the per-stage path rounds to f32 (the spike it stores) and then to f16; the two
// differ in rare double-rounding cases (measured: 1 element in ~50 000), which is enough to break bit-identity downstream.
template <int ROWS>
__device__ __forceinline__ void fe_put(unsigned char* X, int row, int k, float v) {
    unsigned char* p = X + fe_panel_off<ROWS>(row, k);
    v = settle(v);
    const _Float16 hi = (_Float16)v;
    *reinterpret_cast<_Float16*>(p) = hi;
    *reinterpret_cast<_Float16*>(p + ROWS * 64) = (_Float16)(v - (float)hi);
}

// Clamped neuron parameters of a channel (8 floats per channel, pack_fd_neuron_kernel): carried from one chunk / round to the next as
// the two VECTORS they are loaded as — a carried struct of six scalars gets its loads merged by the load vectoriser into pieces that
// overlap the fields, and the compiler then rebuilds the fields through stack slots (scratch) — and taken apart only where used.
struct FeNP {
    f32x4 a;       // decay, adapt, rdecay, theta0
    f32x2 b;       // dT, rh
};
__device__ __forceinline__ FeNP fe_load_np(const float* __restrict__ nprm, int col) {
    const float* q = nprm + (int64_t)col * 8;
    FeNP v;
    v.a = *reinterpret_cast<const f32x4*>(q);
    v.b = *reinterpret_cast<const f32x2*>(q + 4);
    return v;
}
__device__ __forceinline__ NeuronP fe_np(const FeNP& v) {
    NeuronP p;
    p.decay = v.a.x;
    p.adapt = v.a.y;
    p.rdecay = v.a.z;
    p.theta0 = v.a.w;
    p.dT = v.b.x;
    p.rh = v.b.y;
    return p;
}

// neuron kinds: 1 = EIF (blocks 0 and 1), 2 = LIF (blocks 2 and 3), both in NeuronStep2's packed arithmetic — the arithmetic of the
// per-stage kernels fd_neuron_kernel / fd_edge_neuron_kernel.  (Until round 4 block 0 ran neuron_step<true>, the scalar form with
// its IEEE division per step: four times the instructions of the packed form, and — one 64-channel chunk per patch third, emitted
// by half the waves while the others waited at the barrier — a third of multi_scale_conv's whole emission time.)
// step-0 spikes of TWO elements of one channel
template <int KIND>
__device__ __forceinline__ f32x2 fe_spike0(f32x2 x, const NeuronP& p) {
    NeuronStep2<KIND == 1> ns(p);
    return ns.step(x, true);
}

// ---- fragment-ordered weights (16x16x32 B operand): frag (column sub-block cs, k32 step s, plane) of a [n, k] matrix = 64 lanes x 8 halves
__device__ __forceinline__ half8 fe_wfrag(const half8* __restrict__ wp, int nk32, int cs, int s, int plane, int lane) {
    return wp[(((int64_t)cs * nk32 + s) * 2 + plane) * 64 + lane];
}

// C[16 NRS rows, 16 NCS columns] = panel . W^T for this wave's column sub-blocks cs[]: v_mfma_f32_16x16x32_f16 (fn_edge_chain.hip:
// the clock the chip holds on it, and why it equals two chained 32x32x16 bit for bit), per k32 step and accumulator a_lo w_hi, then
// a_hi w_lo, then a_hi w_hi — the pass-major order of every split-f16 GEMM of the library.  acc[i][j] = rows 16 (rs0 + i) + 4 (lane >>
// 4) + e, column 16 cs[j] + (lane & 15).  One k32 step of weight fragments in registers; the column sub-blocks go in two halves whose
// fragments are refilled in place behind their MFMAs (NCS = 1: behind the step).  RH > 0: the operand fragments of RH row sub-blocks
// at a time (registers), re-read for the second column half.
template <int NCS, int NRS, int RH = 0, bool CARRY_HALF = false>
__device__ __forceinline__ void fe_gemm16(const unsigned char* X, int plane_bytes, const half8* __restrict__ wp, int nk32, const int (&cs)[NCS],
                                          int lane, f32x4 (&acc)[NRS][NCS], int rs0 = 0, int s_first = 0, int nsteps = -1, int nk32_total = -1,
                                          half8 (*Wh)[NCS > 1 && CARRY_HALF ? NCS / 2 : NCS] = nullptr,
                                          half8 (*Wl)[NCS > 1 && CARRY_HALF ? NCS / 2 : NCS] = nullptr, int nrc_used = 1 << 30) {
    // (nrc_used, wave-uniform: only the first nrc_used chunks of RH row sub-blocks are multiplied — a partly filled last tile)
    // (Wh / Wl: the caller keeps fragment registers across calls — multi_scale_conv's rounds; nullptr: local, fresh accumulators.
    //  CARRY_HALF: only the FIRST column half's fragments are the caller's (24 registers that stay live through the caller's
    //  emission phase instead of 48); the second half's first fragments are loaded on entry and land under the first half's MFMAs)
    const int c16 = lane & 15, g = lane >> 4;
    const unsigned char* xa = X + (16 * rs0 + c16) * 64 + ((g ^ ((c16 >> 2) & 3)) * 16);      // A operand: row 16 rs + c16, k chunk g
    const int kstep = 2 * plane_bytes;
    constexpr int HC = NCS > 1 ? NCS / 2 : 1, NH = NCS > 1 ? 2 : 1;
    constexpr int RC = RH > 0 ? RH : NRS, NRC = NRS / RC;
    constexpr int NCARRY = NCS > 1 && CARRY_HALF ? NCS / 2 : NCS;
    const int nk_all = nk32_total < 0 ? nk32 : nk32_total;
    if (nsteps < 0) nsteps = nk32;
    half8 wh[NCS], wl[NCS];
    if (!Wh) {
#pragma unroll
        for (int i = 0; i < NRS; ++i)
#pragma unroll
            for (int j = 0; j < NCS; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NCS; ++j) {
            wh[j] = fe_wfrag(wp, nk_all, cs[j], s_first, 0, lane);
            wl[j] = fe_wfrag(wp, nk_all, cs[j], s_first, 1, lane);
        }
    } else {
#pragma unroll
        for (int j = 0; j < NCS; ++j) {
            if (j < NCARRY) {
                wh[j] = (*Wh)[j];
                wl[j] = (*Wl)[j];
            } else {
                wh[j] = fe_wfrag(wp, nk_all, cs[j], s_first, 0, lane);
                wl[j] = fe_wfrag(wp, nk_all, cs[j], s_first, 1, lane);
            }
        }
    }
#pragma unroll 1
    for (int s = 0; s < nsteps; ++s) {
        const int sa = s_first + s + 1;
        const int sn = sa < nk_all ? sa : nk_all - 1;             // (the tail re-loads the last fragment: unconditional loads)
#pragma unroll
        for (int hc = 0; hc < NH; ++hc) {
#pragma unroll
            for (int rc = 0; rc < NRC; ++rc) {
                if (rc >= nrc_used) continue;
                half8 ah[RC], al[RC];
#pragma unroll
                for (int i = 0; i < RC; ++i) {
                    ah[i] = *reinterpret_cast<const half8*>(xa + s * kstep + (rc * RC + i) * 1024);
                    al[i] = *reinterpret_cast<const half8*>(xa + s * kstep + (rc * RC + i) * 1024 + plane_bytes);
                }
#pragma unroll
                for (int i = 0; i < RC; ++i)
#pragma unroll
                    for (int j = hc * HC; j < (hc + 1) * HC && j < NCS; ++j)
                        acc[rc * RC + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], wh[j], acc[rc * RC + i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < RC; ++i)
#pragma unroll
                    for (int j = hc * HC; j < (hc + 1) * HC && j < NCS; ++j)
                        acc[rc * RC + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], wl[j], acc[rc * RC + i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < RC; ++i)
#pragma unroll
                    for (int j = hc * HC; j < (hc + 1) * HC && j < NCS; ++j)
                        acc[rc * RC + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], wh[j], acc[rc * RC + i][j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = hc * HC; j < (hc + 1) * HC && j < NCS; ++j) {
                wh[j] = fe_wfrag(wp, nk_all, cs[j], sn, 0, lane);
                wl[j] = fe_wfrag(wp, nk_all, cs[j], sn, 1, lane);
            }
        }
    }
    if (Wh) {
#pragma unroll
        for (int j = 0; j < NCARRY; ++j) {
            (*Wh)[j] = wh[j];
            (*Wl)[j] = wl[j];
        }
    }
}

// sum of v over the 64 lanes (wave-uniform result): DPP butterflies inside rows of 16, then four scalar reads — the
// ds_bpermute shuffles of __shfl_xor cost an LDS round trip each
__device__ __forceinline__ int fe_wave_sum(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);      // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);      // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);     // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);     // row_mirror: every lane holds its row's sum
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) +
           __builtin_amdgcn_readlane(v, 48);
}

// ---------------------------------------------------------------------------------------------
// In-patch kNN (fd/snn_coder.py:25-32), the arithmetic of patch_knn_kernel (patch_ops.hip): score[i][j] = (-xx[j] - (-2 <xi,xj>))
// - xx[i] with <.,.> a channel-ascending f32 FMA chain and xx a sequential sum of rounded squares; top-k by descending score,
// equal scores by ascending index.  `fill(F, c0, cw, ftid)` (threads 256..511, ftid = tid - 256) stages channels c0 .. c0+cw-1
// of all 64 rows (rows >= m: 0; channels up to the next multiple of 4: 0) into F[row][36]; the staging area is double-buffered:
// waves 4-7 fill chunk n + 1 while waves 0-3 accumulate chunk n (one barrier per chunk).
//
// Round 3: the inner products run on the matrix pipe.  v_mfma_f32_16x16x4_f32 IS a k-ascending chain of IEEE f32 FMAs
// (profiles/micro/mfma_f32_exact.hip: 0 of 256 outputs differ from fmaf chains over 256 channels, likewise the 32x32x2 shape),
// so a 16 x 16 block of pairs takes one MFMA per 4 channels, operands one LDS dword per lane (row stride 36: conflict-free), and
// the scores stay bit-identical to the VALU form (9 FMAs per 6 LDS reads per lane: bound by the LDS port).  <xi,xj> = <xj,xi>
// bit for bit, so only the six blocks bi <= bj of the 3 x 3 block grid are computed: waves 0-2 two blocks each, wave 3 the squared
// norms.  Ranks: one wave per row, the row's keys broadcast through scalar registers.  Output: idx_out[i][rank] (bytes, row
// pitch 48) and the optional int32 tap [m][k].
// ---------------------------------------------------------------------------------------------
template <typename Fill>
__device__ __forceinline__ void fe_knn(unsigned char* R2, float* xx, unsigned char* idx_out, int m, int c, int k, Fill fill,
                                       int32_t* __restrict__ tap, int tid) {
    float* Fb = reinterpret_cast<float*>(R2);                          // two buffers of [64][36]
    unsigned* K = reinterpret_cast<unsigned*>(R2 + FE_KEYS_OFF);
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g4 = lane >> 4;
    // blocks of this wave (wave-uniform): wave 0: (0,0) (1,2), wave 1: (0,1) (2,2), wave 2: (0,2) (1,1)
    const int bi0 = 0, bj0 = wave < 3 ? wave : 0;
    const int bi1 = wave == 1 ? 2 : 1, bj1 = wave == 2 ? 1 : 2;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    float myxx = 0.f;
    __syncthreads();                                                   // the area is free (previous phase done)
    if (wave >= 4) fill(Fb, 0, min(32, c), tid - 256);
    __syncthreads();
    int buf = 0;
    for (int c0 = 0; c0 < c; c0 += 32, buf ^= 1) {
        const int cw = min(32, c - c0);
        const float* F = Fb + buf * (64 * FE_F_LD);
        if (wave >= 4) {
            if (c0 + 32 < c) fill(Fb + (buf ^ 1) * (64 * FE_F_LD), c0 + 32, min(32, c - c0 - 32), tid - 256);
        } else if (wave < 3) {
            const float* fa0 = F + (16 * bi0 + r16) * FE_F_LD + g4;
            const float* fb0 = F + (16 * bj0 + r16) * FE_F_LD + g4;
            const float* fa1 = F + (16 * bi1 + r16) * FE_F_LD + g4;
            const float* fb1 = F + (16 * bj1 + r16) * FE_F_LD + g4;
            if (cw == 32) {                                            // the 32 operand reads first, then the 16 MFMAs
                float a0[8], b0[8], a1[8], b1[8];
#pragma unroll
                for (int s4 = 0; s4 < 8; ++s4) {
                    a0[s4] = fa0[4 * s4];
                    b0[s4] = fb0[4 * s4];
                    a1[s4] = fa1[4 * s4];
                    b1[s4] = fb1[4 * s4];
                }
#pragma unroll
                for (int s4 = 0; s4 < 8; ++s4) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s4], b0[s4], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s4], b1[s4], acc1, 0, 0, 0);
                }
            } else {
                for (int cc = 0; cc < cw; cc += 4) {                   // (the fill zero-pads to a multiple of 4: + 0 is exact)
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa0[cc], fb0[cc], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa1[cc], fb1[cc], acc1, 0, 0, 0);
                }
            }
        } else if (lane < m) {                                         // wave 3: squared norms
            const float* fr = F + lane * FE_F_LD;
            for (int cc2 = 0; cc2 < cw; ++cc2) {
                const float sq = __fmul_rn(fr[cc2], fr[cc2]);
                myxx = (c0 == 0 && cc2 == 0) ? sq : __fadd_rn(myxx, sq);
            }
        }
        __syncthreads();                                               // chunk n consumed, chunk n + 1 staged
    }
    if (wave == 3 && lane < m) xx[lane] = myxx;
    __syncthreads();
    if (wave < 3) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int bi = q ? bi1 : bi0, bj = q ? bj1 : bj0;
            const f32x4 acc = q ? acc1 : acc0;
            const int j = 16 * bj + r16;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = 16 * bi + 4 * g4 + e;
                if (i < m && j < m) {
                    const float inner = __fmul_rn(-2.0f, acc[e]);
                    // stored as the order-preserving integer key of the score (+ 0.0f: -0 and +0 compare equal as floats)
                    K[i * (m + 1) + j] = float_max_key(__fadd_rn(__fsub_rn(__fsub_rn(-xx[j], inner), xx[i]), 0.0f));
                    if (bi != bj) K[j * (m + 1) + i] = float_max_key(__fadd_rn(__fsub_rn(__fsub_rn(-xx[i], inner), xx[j]), 0.0f));
                }
            }
        }
    }
    __syncthreads();
    const int full = m * (m - 1) / 2;
    for (int i = wave; i < m; i += FE_NT / 64) {
        const unsigned key0 = lane < m ? K[i * (m + 1) + lane] : 0u;
        int r0 = 0;
        for (int jb = 0; jb < m; jb += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) r0 += (unsigned)__builtin_amdgcn_readlane(key0, (jb + u) & 63) > key0;
        }
        const int tot = fe_wave_sum(lane < m ? r0 : 0);
        if (tot != full) {                      // (wave-uniform) equal scores in this row: exact (score, index) order
            const unsigned long long k0 = ((unsigned long long)key0 << 32) | (unsigned)~lane;
            r0 = 0;
            for (int jb = 0; jb < m; jb += 8) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int jp = jb + u;
                    const unsigned long long kv = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(key0, jp & 63) << 32) | (unsigned)~jp;
                    r0 += kv > k0;
                }
            }
        }
        if (lane < m && r0 < k) {
            idx_out[i * FE_M + r0] = (unsigned char)lane;
            if (tap) tap[i * k + r0] = lane;
        }
    }
}

// ---- multi_scale_conv operand panel: a THIRD of a patch (16 points) x 4 stacked steps = 64 rows (row = 4 point + step: a lane's
// accumulator quad is then the four steps of one point, and a point's swizzle key (row >> 2) & 3 does not depend on the step) x
// 128 columns of one K round: [k32 step (4)][plane hi | lo][64 rows][32 halves] = 32 KiB
constexpr int FE_PR = 64;                  // panel rows
constexpr int FE_TP = FE_M / 3;            // points per third of a patch

// split-f16 halves of v at a precomputed panel address (hi plane; lo plane 64 rows x 64 B further).  lo = f16(v - hi) written as
// the FMA it is (v - hi is exact in f32, so one rounding to f16 either way): the compiler then has v_fma_mixlo_f16 for it.
__device__ __forceinline__ void fe_put_at(unsigned char* p, float v) {
    v = settle(v);                                          // (see fe_put)
    const _Float16 hi = (_Float16)v;
    *reinterpret_cast<_Float16*>(p) = hi;
    *reinterpret_cast<_Float16*>(p + FE_PR * 64) = (_Float16)__builtin_fmaf((float)hi, -1.0f, v);
}

// T-step neuron loops of this thread's four (point, channel) elements of one K round of multi_scale_conv (four consecutive points
// of the third, one channel): the steps [t0, t0 + nemit) go to the panel as split-f16 rows 4 point + (t - t0) (pe[e] = panel
// address of point e's step-0 row in this thread's column; a step further = one row = 64 bytes); steps before t0 — a second
// group of steps when T > 4 — are run for the state only.  fd:432-474 with the closed gate.
// FAST: t0 = 0, nemit = 4, no tap — the production shape, fully unrolled.  pt0 = patch-level index of the first point (taps, m).
template <int KIND, bool FAST>
__device__ __forceinline__ void fe_emit4(const float (&x)[4], const NeuronP& p, int t0, int nemit, unsigned char* const (&pe)[4], int pt0,
                                         int m, bool count, int* __restrict__ gate, float* __restrict__ tap, int64_t tap_tstride) {
    const int total = FAST ? 4 : t0 + nemit;
    {
        NeuronStep2<KIND == 1> n0(p), n1(p);
        const f32x2 z = f32x2{0.f, 0.f};
        bool o0 = false, o1 = false;
        if (FAST) {
#pragma unroll
            for (int step = 0; step < 4; ++step) {
                if (step > 0) {
                    o0 = o0 || n0.gate_open();
                    o1 = o1 || n1.gate_open();
                }
                const f32x2 s0 = n0.step(step == 0 ? f32x2{x[0], x[1]} : z, step == 0);
                const f32x2 s1 = n1.step(step == 0 ? f32x2{x[2], x[3]} : z, step == 0);
                fe_put_at(pe[0] + step * 64, s0.x);
                fe_put_at(pe[1] + step * 64, s0.y);
                fe_put_at(pe[2] + step * 64, s1.x);
                fe_put_at(pe[3] + step * 64, s1.y);
            }
        } else {
            for (int step = 0; step < total; ++step) {
                const bool first = step == 0;
                if (!first) {
                    o0 = o0 || n0.gate_open();
                    o1 = o1 || n1.gate_open();
                }
                const f32x2 s0 = n0.step(first ? f32x2{x[0], x[1]} : z, first);
                const f32x2 s1 = n1.step(first ? f32x2{x[2], x[3]} : z, first);
                if (step >= t0) {
                    const float sv[4] = {s0.x, s0.y, s1.x, s1.y};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        fe_put_at(pe[e] + (step - t0) * 64, sv[e]);
                        if (tap && pt0 + e < m) tap[(int64_t)step * tap_tstride + (int64_t)(pt0 + e) * 960] = sv[e];
                    }
                }
            }
        }
        if (count) {                                        // a pair counts once
            const int open = (o0 && pt0 < m ? 1 : 0) + (o1 && pt0 + 2 < m ? 1 : 0);
            if (open) atomicAdd(gate, open);
        }
    }
}

// max(a, b) for finite values and -inf as v_med3_f32(a, b, FLT_MAX): one instruction.  fmaxf costs two more — the compiler quiets
// each operand it cannot prove non-signalling (MFMA results) with v_max_f32 x, x first — and med3(a, b, +inf) is folded back into
// that fmaxf.  (NOT inline asm: the hazard recogniser does not see into it, and a VALU read of a matrix-pipe result needs software
// wait states.)  No NaN and no +inf reaches these maxima.
__device__ __forceinline__ float fe_max2(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, 3.40282346638528859812e38f); }
__device__ __forceinline__ float fe_max3(float a, float b, float c) { return fe_max2(fe_max2(a, b), c); }
// Four per-lane values v[0..3], each to be maximised over the four 16-lane rows of the wave: a reduce-scatter in three swaps and
// three maxima — row g ends up with the full maximum of v[g] (v_permlane32_swap exchanges the upper half of its first operand
// with the lower half of its second, v_permlane16_swap the odd rows of the first with the even rows of the second).
__device__ __forceinline__ float fe_rows_max4(float v0, float v1, float v2, float v3) {
    const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v0), __float_as_uint(v2), false, false);
    const float x = fe_max2(__uint_as_float(a[0]), __uint_as_float(a[1]));          // rows 0, 1: v0 over the halves | rows 2, 3: v2
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v1), __float_as_uint(v3), false, false);
    const float y = fe_max2(__uint_as_float(b[0]), __uint_as_float(b[1]));          // rows 0, 1: v1 | rows 2, 3: v3
    const auto c = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return fe_max2(__uint_as_float(c[0]), __uint_as_float(c[1]));                   // row g: v[g] over all four rows
}


// ---------------------------------------------------------------------------------------------
// EdgeConv block L = 1..3 (fd:447-474 at t = 0): neighbour search on the step-0 spikes of block L-1, factored EdgeConv GEMM,
// max over the neighbours, shift + LeakyReLU -> the block's pre-activation x0 (LDS for L <= 2, registers for L = 3).
// ---------------------------------------------------------------------------------------------
template <int L>
__device__ __forceinline__ void fe_edge_block(const FdEncArgs& a, unsigned char* smem, int64_t patch_i, int tid, float (&x3)[3][4][4]) {
    constexpr int CIN = 64 << (L - 1), COUT = 128 << (L - 1);
    constexpr int COFF_IN = L == 1 ? 0 : (L == 2 ? 64 : 192), COFF_OUT = L == 1 ? 64 : (L == 2 ? 192 : 448);
    constexpr int KIND_IN = L <= 2 ? 1 : 2;                  // block L-1: EIF (blocks 0, 1) | LIF
    constexpr int NPW = L == 3 ? 2 : 1;                    // column-block pairs (W1 + W2 | W1) per wave
    float* XS = reinterpret_cast<float*>(smem + FE_OFF_XS);
    unsigned char* R2 = smem + FE_OFF_R2;
    unsigned char* IDXL = smem + FE_OFF_IDXL;
    float* xx = reinterpret_cast<float*>(smem + FE_OFF_XX);
    const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = a.m, kk = a.kk;

    // (1) neighbour tables: forced (parity protocol) or ranked in the feature space of block L-1's step-0 spikes
    const int64_t tab_off = (((int64_t)(L - 1) * a.b_total + a.s0 + patch_i) * m) * kk;
    if (a.knn_force) {
        __syncthreads();
        for (int e = tid; e < m * kk; e += FE_NT) {
            const int v = a.knn_force[tab_off + e];
            IDXL[(e / kk) * FE_M + (e % kk)] = (unsigned char)v;
            if (a.tap_knn) a.tap_knn[tab_off + e] = v;
        }
    } else {
        // thread (256 of them) -> channel c0 + (ftid & 31), rows (ftid >> 5) + 8 n; a chunk's parameters are loaded one chunk ahead
        FeNP pn = fe_load_np(a.nprm, COFF_IN + (tid & 31));
        auto fill = [&](float* F, int c0, int cw, int ftid) {
            const int cc = ftid & 31;
            const NeuronP p = fe_np(pn);
            pn = fe_load_np(a.nprm, COFF_IN + (c0 + 32 < CIN ? c0 + 32 : c0) + cc);
#pragma unroll
            for (int n = 0; n < 8; n += 2) {
                const int ia = (ftid >> 5) + 8 * n, ib = ia + 8;            // rows < 48 hold x0 (whatever it is beyond m)
                if (ia < FE_M) {                                            // (rows 48..63: zeros, ib < 48 whenever ia < 40)
                    const f32x2 sp = fe_spike0<KIND_IN>(f32x2{XS[ia * FE_XLD + COFF_IN + c0 + cc],
                                                              XS[(ib < FE_M ? ib : ia) * FE_XLD + COFF_IN + c0 + cc]}, p);
                    F[ia * FE_F_LD + cc] = ia < m ? sp.x : 0.f;
                    F[ib * FE_F_LD + cc] = (ib < m && ib < FE_M) ? sp.y : 0.f;
                } else {
                    F[ia * FE_F_LD + cc] = 0.f;
                    F[ib * FE_F_LD + cc] = 0.f;
                }
            }
            (void)cw;
        };
        fe_knn(R2, xx, IDXL, m, CIN, kk, fill, a.tap_knn ? a.tap_knn + tab_off : nullptr, tid);
    }
    __syncthreads();
    FE_STAMP(4 * L);
    // (2) step-0 spikes of block L-1 as the split-f16 operand panel [64 rows][CIN] (rows >= m: whatever x0 holds there — the
    //     GEMM's rows are independent and those outputs are never used)
    {
        const int cc = tid & 63;
        FeNP pn2 = fe_load_np(a.nprm, COFF_IN + cc);
        for (int c0 = 0; c0 < CIN; c0 += 64) {
            const NeuronP p = fe_np(pn2);
            pn2 = fe_load_np(a.nprm, COFF_IN + (c0 + 64 < CIN ? c0 + 64 : c0) + cc);
#pragma unroll
            for (int n = 0; n < 6; n += 2) {
                const int ia = (tid >> 6) + 8 * n, ib = ia + 8;            // rows 0..47
                const f32x2 sp = fe_spike0<KIND_IN>(f32x2{XS[ia * FE_XLD + COFF_IN + c0 + cc], XS[ib * FE_XLD + COFF_IN + c0 + cc]}, p);
                fe_put<64>(R2, ia, c0 + cc, sp.x);
                fe_put<64>(R2, ib, c0 + cc, sp.y);
#ifdef FE_DEBUG_PANEL       // diagnostic build: the f32 values behind the panel, in place of block L-1's x0 in the x0 tap
                if (a.tap_x0 && ia < m) a.tap_x0[((a.s0 + patch_i) * m + ia) * 960 + COFF_IN + c0 + cc] = sp.x;
                if (a.tap_x0 && ib < m) a.tap_x0[((a.s0 + patch_i) * m + ib) * 960 + COFF_IN + c0 + cc] = sp.y;
#endif
            }
        }
    }
    __syncthreads();
    FE_STAMP(4 * L + 1);
    // (3) GEMM: pairs pr = w + 8 q; accumulators acc[q][rows i][A' | B]
    const bool active = L != 1 || w < 4;                   // block 1 has only four pairs
    f32x4 acc[NPW][3][4];                                  // [pair][row sub-block][0, 1: A' = (W1 + W2) x | 2, 3: B = W1 x] (rows < 48 only)
    if (active) {
#pragma unroll
        for (int q = 0; q < NPW; ++q) {
            const int pr = w + 8 * q;
            const int css[4] = {2 * pr, 2 * pr + 1, 2 * (COUT / 32 + pr), 2 * (COUT / 32 + pr) + 1};
            fe_gemm16<4, 3>(R2, 4096, reinterpret_cast<const half8*>(a.edge_wp[L - 1]), CIN / 32, css, lane, acc[q]);
        }
    }
    __syncthreads();                                       // every wave has read the panel: R2 becomes the staging area
    FE_STAMP(4 * L + 2);
    // (4) per pair: stage A' (rows < 48) in this wave's private [48][36] tile (pitch 36 floats: 16-byte aligned rows that do not all
    //     start in the same bank); the max over the kk neighbours is taken with the lanes re-mapped to (channel quad, six rows) —
    //     one 16-byte tile read per neighbour serves four channels (a lane-per-channel walk spends 4 instructions per element and
    //     neighbour) —, written back over the tile, and picked up in the accumulator layout: pre = LeakyReLU((max - B) + shift)
    float* SAw = reinterpret_cast<float*>(R2) + w * FE_TILE;
    const int c16 = lane & 15, g4 = lane >> 4;                  // accumulator layout: column 16 cj + c16, rows 16 rs + 4 g4 + e
    const int qd = lane & 7, g6 = 6 * (lane >> 3);              // the max phase's mapping: channels 4 qd .. 4 qd + 3, rows g6 .. g6 + 5
#pragma unroll
    for (int q = 0; q < NPW; ++q) {
        float pre[3][2][4];
        if (active) {
            const int col0 = 32 * (w + 8 * q) + c16;           // channel of this lane inside the block (column sub-block cj: + 16 cj)
            const float sh[2] = {a.shift[L - 1][col0], a.shift[L - 1][col0 + 16]};
#pragma unroll
            for (int rs = 0; rs < 3; ++rs)
#pragma unroll
                for (int cj = 0; cj < 2; ++cj)
#pragma unroll
                    for (int e = 0; e < 4; ++e)               // the GEMM epilogue's value (no bias)
                        SAw[(16 * rs + 4 * g4 + e) * FE_TLD + 16 * cj + c16] = __fadd_rn(__fmul_rn(acc[q][rs][cj][e], 0.0625f), 0.0f);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            {
                float4 mx[6];
#pragma unroll
                for (int r = 0; r < 6; ++r) mx[r] = make_float4(-__builtin_huge_valf(), -__builtin_huge_valf(), -__builtin_huge_valf(), -__builtin_huge_valf());
                const unsigned char* ir0 = IDXL + g6 * FE_M;
                const float* tq = SAw + 4 * qd;
                int j = 0;
                for (; j + 4 <= kk; j += 4) {
                    unsigned pk[6];
#pragma unroll
                    for (int r = 0; r < 6; ++r) pk[r] = *reinterpret_cast<const unsigned*>(ir0 + r * FE_M + j);
#pragma unroll
                    for (int r = 0; r < 6; ++r) {
                        const float4 v0 = *reinterpret_cast<const float4*>(tq + (pk[r] & 255u) * FE_TLD);
                        const float4 v1 = *reinterpret_cast<const float4*>(tq + ((pk[r] >> 8) & 255u) * FE_TLD);
                        const float4 v2 = *reinterpret_cast<const float4*>(tq + ((pk[r] >> 16) & 255u) * FE_TLD);
                        const float4 v3 = *reinterpret_cast<const float4*>(tq + (pk[r] >> 24) * FE_TLD);
                        mx[r].x = fmaxf(fmaxf(mx[r].x, v0.x), fmaxf(v1.x, fmaxf(v2.x, v3.x)));
                        mx[r].y = fmaxf(fmaxf(mx[r].y, v0.y), fmaxf(v1.y, fmaxf(v2.y, v3.y)));
                        mx[r].z = fmaxf(fmaxf(mx[r].z, v0.z), fmaxf(v1.z, fmaxf(v2.z, v3.z)));
                        mx[r].w = fmaxf(fmaxf(mx[r].w, v0.w), fmaxf(v1.w, fmaxf(v2.w, v3.w)));
                    }
                }
                for (; j < kk; ++j) {
#pragma unroll
                    for (int r = 0; r < 6; ++r) {
                        const float4 v0 = *reinterpret_cast<const float4*>(tq + ir0[r * FE_M + j] * FE_TLD);
                        mx[r].x = fmaxf(mx[r].x, v0.x);
                        mx[r].y = fmaxf(mx[r].y, v0.y);
                        mx[r].z = fmaxf(mx[r].z, v0.z);
                        mx[r].w = fmaxf(mx[r].w, v0.w);
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();           // every lane has read the tile: the maxima go over it
#pragma unroll
                for (int r = 0; r < 6; ++r) *reinterpret_cast<float4*>(SAw + (g6 + r) * FE_TLD + 4 * qd) = mx[r];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
            }
#pragma unroll
            for (int rs = 0; rs < 3; ++rs)
#pragma unroll
                for (int cj = 0; cj < 2; ++cj)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int row = 16 * rs + 4 * g4 + e;
                        const float mxv = SAw[row * FE_TLD + 16 * cj + c16];
                        const float xb = __fadd_rn(__fmul_rn(acc[q][rs][2 + cj][e], 0.0625f), 0.0f);
                        pre[rs][cj][e] = lrelu02(__fadd_rn(__fsub_rn(mxv, xb), sh[cj]));
                        if (a.tap_x0 && row < m) a.tap_x0[((a.s0 + patch_i) * m + row) * 960 + COFF_OUT + col0 + 16 * cj] = pre[rs][cj][e];
                    }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();               // this wave's reads of its tile are done before it is rewritten
            if (L <= 2) {
#pragma unroll
                for (int rs = 0; rs < 3; ++rs)
#pragma unroll
                    for (int cj = 0; cj < 2; ++cj)
#pragma unroll
                        for (int e = 0; e < 4; ++e) XS[(16 * rs + 4 * g4 + e) * FE_XLD + COFF_OUT + col0 + 16 * cj] = pre[rs][cj][e];
            } else {
                // block 3: into the tile again, then every thread picks up its (channel, six rows) elements of multi_scale_conv's
                // K rounds: round j of this half = columns 64 j .. 64 j + 63 = the tiles of waves 2 j, 2 j + 1
#pragma unroll
                for (int rs = 0; rs < 3; ++rs)
#pragma unroll
                    for (int cj = 0; cj < 2; ++cj)
#pragma unroll
                        for (int e = 0; e < 4; ++e) SAw[(16 * rs + 4 * g4 + e) * FE_TLD + 16 * cj + c16] = pre[rs][cj][e];
            }
        }
        if (L == 3) {
            __syncthreads();
            const float* SA = reinterpret_cast<const float*>(R2);
            // multi_scale_conv's thread (wave w, lane): 64-column chunk parity cw = w & 1 (chunks 7 + c3 of the 15; cw = 1 takes the
            // even c3), points 4 (w >> 1) + e of each third of the patch; chunk c3 of this pair round = tiles 2 (c3 & 3), 2 (c3 & 3) + 1
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int c3l = 2 * jj + 1 - (w & 1);          // chunk inside this pair round (0..3): block chunk c3 = 4 q + c3l
                // round of chunk 7 + c3: odd waves R = 3 + 2 q + jj, even waves R = 4 + 2 q + jj: slot 2 q + jj = R - 4 + (w & 1) for both
#pragma unroll
                for (int th = 0; th < 3; ++th)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        x3[th][2 * q + jj][e] = SA[(2 * c3l + (lane >> 5)) * FE_TILE + (FE_TP * th + 4 * (w >> 1) + e) * FE_TLD + (lane & 31)];
            }
            __syncthreads();
        }
    }
    __syncthreads();
    FE_STAMP(4 * L + 3);
}

// GENERAL = false: the production shape — T = 4 and no spike tap, so every emission is the fully unrolled four-step form — compiled
// WITHOUT the general emission path (any T, taps): with both in one kernel the general path's extra live values pushed the
// allocation over 256 registers everywhere (37 spilled registers + block 3's x0 indexed in scratch, round 3); alone the kernel
// needs 241 registers and no scratch at all.  GENERAL = true: everything else (T != 4, stage taps), may spill.
template <bool GENERAL>
__global__ __launch_bounds__(FE_NT, 1) void fd_encoder_kernel(const FdEncArgs a) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    float* XS = reinterpret_cast<float*>(smem + FE_OFF_XS);
    unsigned char* R2 = smem + FE_OFF_R2;
    float4* XYZ = reinterpret_cast<float4*>(smem + FE_OFF_XYZ);
    unsigned char* IDX0 = smem + FE_OFF_IDX0;
    float* xx = reinterpret_cast<float*>(smem + FE_OFF_XX);
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = a.m;
    const int64_t patch_i = blockIdx.x;
    const float* __restrict__ pp = a.patch + patch_i * m * 3;

    // ---- patch coordinates; neighbour tables start as "point 0" so that rows >= m never index outside a staging tile
    for (int e = tid; e < FE_M; e += FE_NT)
        XYZ[e] = e < m ? make_float4(pp[3 * e], pp[3 * e + 1], pp[3 * e + 2], 0.f) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int e = tid; e < 2 * FE_M * FE_M / 4; e += FE_NT) reinterpret_cast<unsigned*>(IDX0)[e] = 0u;
    __syncthreads();
    FE_STAMP(0);
    // ---- block 0: xyz neighbours (one ranking serves all scales)                                          fd:411-417
    {
        auto fill = [&](float* F, int c0, int cw, int ftid) {
            {                                                          // 64 rows x (x, y, z, 0)
                const int i = ftid >> 2, cc = ftid & 3;
                // (one LDS dword at a computed address: a select chain over a float4's components becomes a dynamic vector
                //  extract, which the compiler lowers through a stack slot)
                const float v = reinterpret_cast<const float*>(XYZ)[4 * (i < FE_M ? i : 0) + cc];      // XYZ[.].w == 0
                F[i * FE_F_LD + cc] = i < m ? v : 0.f;
            }
            (void)c0; (void)cw;
        };
        fe_knn(R2, xx, IDX0, m, 3, a.kmax0, fill, nullptr, tid);
    }
    __syncthreads();
    FE_STAMP(1);
    // ---- block 0: EdgeConv(6 -> 64) per scale, max over the ks nearest, + bias, LeakyReLU -> split panel [64][64 S]   fd:413-420
    // Round 4: on the matrix pipe.  An edge's value for (scale, channel) is the 6-term dot product w . (xj - xi, xj), which the
    // per-stage kernel (fd_edge0_scalar_kernel) evaluates as a k-ascending chain w0 dx, fma(w1, dy, .), ... fma(w5, zj, .) — and a
    // k-ascending chain of IEEE f32 FMAs is exactly what v_mfma_f32_16x16x4_f32 computes (fe_knn; profiles/micro/mfma_f32_exact.hip;
    // fma(w0, dx, +0) = w0 dx up to the sign of a zero, which neither the max nor the + bias that follow can see).  So a 16 x 16
    // block of (edge, channel) pairs is two MFMAs (k = dx dy dz xj | yj zj 0 0): rows = 16 consecutive neighbours of the point's
    // sorted list, columns = 16 channels.  A point needs its 6 edge operands once — they serve all scales and channel blocks —
    // and 8 MFMAs per (scale, 16-neighbour tile): 56 per point at the reference's scales (8, 16, 32, 48), against 624 packed FMAs
    // + 208 maxima per lane pair before (2.7 x their issue floor: index -> coordinate chains of two LDS latencies per edge).
    {
        const float* XYZf = reinterpret_cast<const float*>(XYZ);
        const int r16 = lane & 15, g4 = lane >> 4;
        float wb[4][4][2], bb[4];                              // B operands: W_s[channel 16 ct + r16][k = 4 kstep + g4] (k >= 6: 0)
#pragma unroll
        for (int sc = 0; sc < 4; ++sc) {
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int kstep = 0; kstep < 2; ++kstep) {
                    const int k = 4 * kstep + g4;
                    wb[sc][ct][kstep] = (sc < a.nscale && k < 6) ? a.e0_w[((int64_t)sc * 64 + 16 * ct + r16) * 6 + k] : 0.f;
                }
            bb[sc] = sc < a.nscale ? a.e0_b[sc * 64 + lane] : 0.f;       // the channel this lane STORES: 16 g4 + r16 = lane
        }
        int ksv[4];
#pragma unroll
        for (int sc = 0; sc < 4; ++sc) ksv[sc] = sc < a.nscale ? a.ks[sc] : 0;
        float big = 0.f;
        for (int i = w; i < m; i += FE_NT / 64) {
            // edge operands of point i: A[row = neighbour 16 jt + r16][k = 4 kstep + g4]
            const float xi_c = g4 < 3 ? XYZf[4 * i + g4] : 0.f;                      // kstep 0: (xj - xi, yj - yi, zj - zi, xj)
            float ea[3][2];
#pragma unroll
            for (int jt = 0; jt < 3; ++jt) {
                const int nb = IDX0[i * FE_M + 16 * jt + r16];                       // (entries beyond kmax0: "point 0", masked below)
                ea[jt][0] = XYZf[4 * nb + (g4 < 3 ? g4 : 0)] - xi_c;                // (xj - 0 for g4 == 3: exact)
                ea[jt][1] = g4 < 2 ? XYZf[4 * nb + 1 + g4] : 0.f;                    // kstep 1: (yj, zj, 0, 0)
            }
#pragma unroll
            for (int sc = 0; sc < 4; ++sc) {
                if (sc < a.nscale) {                                                 // (wave-uniform)
                    const int nfull = ksv[sc] >> 4, rem = ksv[sc] & 15;               // whole 16-neighbour tiles + the edges of a last one
                    const int lim = rem - 4 * g4;                                     // ... of which this lane's quad holds e < lim
                    float mx[4];
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) mx[ct] = -__builtin_huge_valf();
#pragma unroll
                    for (int jt = 0; jt < 3; ++jt) {
                        const bool full = jt < nfull, part = jt == nfull && rem > 0;  // (wave-uniform)
                        if (full || part) {
                            f32x4 acc[4];
#pragma unroll
                            for (int ct = 0; ct < 4; ++ct)
                                acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(ea[jt][0], wb[sc][ct][0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
                            for (int ct = 0; ct < 4; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(ea[jt][1], wb[sc][ct][1], acc[ct], 0, 0, 0);
                            if (full) {                                              // acc[ct][e]: edge 16 jt + 4 g4 + e, channel 16 ct + r16
#pragma unroll
                                for (int ct = 0; ct < 4; ++ct)
                                    mx[ct] = fe_max3(fe_max3(mx[ct], acc[ct][0], acc[ct][1]), acc[ct][2], acc[ct][3]);
                            } else {
#pragma unroll
                                for (int ct = 0; ct < 4; ++ct) {
                                    const float ninf = -__builtin_huge_valf();
                                    mx[ct] = fe_max3(fe_max3(mx[ct], 0 < lim ? acc[ct][0] : ninf, 1 < lim ? acc[ct][1] : ninf),
                                                     2 < lim ? acc[ct][2] : ninf, 3 < lim ? acc[ct][3] : ninf);
                                }
                            }
                        }
                    }
                    // over the four lane groups' edges; lane group g4 keeps channel block ct = g4, i.e. channel 16 g4 + r16 = lane
                    const float va = lrelu02(__fadd_rn(fe_rows_max4(mx[0], mx[1], mx[2], mx[3]), bb[sc]));
                    fe_put<64>(R2, i, sc * 64 + lane, va);
                    big = fmaxf(big, fabsf(va));
                }
            }
        }
        if (!(big < 65504.0f) && a.ovf) atomicAdd(a.ovf, 1);
    }
    __syncthreads();
    FE_STAMP(2);
    // ---- block 0: scale_fusion (64 S -> 64) + BN + LeakyReLU = x0 of block 0                                fd:420-421
    if (w < 4) {                                           // 48 rows x 16 columns per wave (one wave per SIMD)
        f32x4 acc[3][1];
        const int css[1] = {w};
        fe_gemm16<1, 3>(R2, 4096, reinterpret_cast<const half8*>(a.fuse_wp), 2 * a.nscale, css, lane, acc);
        const int col = 16 * w + (lane & 15);
        const float bias = a.fuse_b[col];
#pragma unroll
        for (int rs = 0; rs < 3; ++rs)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = 16 * rs + 4 * (lane >> 4) + e;
                const float v = lrelu02(__fadd_rn(__fmul_rn(acc[rs][0][e], 0.0625f), bias));
                XS[row * FE_XLD + col] = v;
                if (a.tap_fused0 && row < m) a.tap_fused0[((a.s0 + patch_i) * m + row) * 64 + col] = v;
                if (a.tap_x0 && row < m) a.tap_x0[((a.s0 + patch_i) * m + row) * 960 + col] = v;
            }
    }
    __syncthreads();
    FE_STAMP(3);
    // ---- blocks 1..3                                                                                          fd:447-474
    // block 3's x0 of this thread's multi_scale_conv elements: [third of the patch][slot][point].  Odd waves emit the block's chunks
    // in rounds 3..6, even waves in rounds 4..7: slot = R - 4 + (w & 1), selected by wave-uniform compares (every array index static)
    float x3[3][4][4];
    fe_edge_block<1>(a, smem, patch_i, tid, x3);
    fe_edge_block<2>(a, smem, patch_i, tid, x3);
    fe_edge_block<3>(a, smem, patch_i, tid, x3);

    // ---- multi_scale_conv over all steps + max over the points                                               fd:476-480
    // The patch in thirds of 16 points (4 stacked steps = 64 rows = 2 row blocks); wave w owns column blocks 3 w .. 3 w + 2 of ALL
    // emb = 768 columns (2 x 3 accumulator blocks = 96 registers), so ONE sweep over K per third and every spike is generated
    // exactly once; K in rounds of 128 columns: wave w emits the 64-column chunk ch = 2 R + (w & 1) for the four points 4 (w >> 1) + e
    // of the third.  Chunk kinds: 0 block 0 | 1, 2 block 1 | 3..6 block 2 (x0 in LDS) | 7..14 block 3 (x0 in registers); round 7
    // has only chunk 14.  emb > 768: further sweeps of 24 column blocks.
    const int cw = w & 1, rg = w >> 1;
    const int nsweep = a.emb / 768 + (a.emb % 768 ? 1 : 0);
    const half8* __restrict__ mscw = reinterpret_cast<const half8*>(a.msc_wp);
    const int64_t tap_tstride = a.b_total * (int64_t)m * 960;
    unsigned char* pe[4];                                   // panel addresses: point 4 rg + e of the third (row 4 point), column 64 cw + lane
#pragma unroll
    for (int e = 0; e < 4; ++e) pe[e] = R2 + fe_panel_off<FE_PR>(4 * (4 * rg + e), 64 * cw + lane);
    // Registers.  The sweep below holds 96 accumulators, 48 of weight fragments and 16 of operand fragments per lane next to the
    // neuron loops of the emission; block 3's x0 (48 values per thread) and the 24 running maxima do not fit beside them (round 3:
    // the compiler indexed x3 dynamically, i.e. kept it in scratch, and spilled 37 more).  So: the x0 values of the CURRENT third sit
    // in LDS — 16 dwords per thread in the idle upper half of the phase buffer, slots only their owner touches: no barrier —, copied
    // from the registers when a third starts (static indices; the other thirds' 32 / 16 values stay in registers until then), and
    // the running maxima live in LDS too (the dead per-patch tables: one dword per (column, step), updated once per third by the
    // lane that owns it).
    float* X3L = reinterpret_cast<float*>(smem + FE_OFF_X3L) + tid;                    // value v of this thread: X3L[512 v]
    float* BESTL = reinterpret_cast<float*>(smem + FE_OFF_BEST) + (lane >> 4) * 128 + w * 16 + (lane & 15);   // (j, step = this lane's row): BESTL[512 j]
    FE_STAMP(16);
#ifdef FE_STAMPS
    unsigned long long fe_t_emit = 0, fe_t_mfma = 0, fe_t_start = 0, fe_t_epi = 0, fe_tp = 0;
#define FE_T0() fe_tp = __builtin_amdgcn_s_memtime()
#define FE_TACC(V) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); (V) += now_ - fe_tp; fe_tp = now_; } while (0)
#else
#define FE_T0() do { } while (0)
#define FE_TACC(V) do { } while (0)
#endif
    for (int sweep = 0; sweep < nsweep; ++sweep) {
        // column blocks of this wave in this sweep; blocks beyond emb are clamped to the last one (computed, never stored)
        const int ncb = a.emb / 32;
        const int cbw = sweep * 24 + 3 * w;
        const int cb0 = cbw + 2 < ncb ? cbw : (ncb >= 3 ? ncb - 3 : 0);
        // this wave's six 16-column sub-blocks: column 16 css[j] + (lane & 15)
        const int css[6] = {2 * cb0, 2 * cb0 + 1, 2 * cb0 + 2, 2 * cb0 + 3, 2 * cb0 + 4, 2 * cb0 + 5};
        for (int t0 = 0; t0 < a.T; t0 += 4) {
            const int nemit = a.T - t0 < 4 ? a.T - t0 : 4;
            const bool count = sweep == 0 && t0 + nemit == a.T;        // the gate check runs once, over all T steps
#ifdef FE_STAMPS
            float* tap = nullptr;                                      // (the spikes tap carries the stamps in this build)
#else
            float* tap = (a.tap_spikes && sweep == 0) ? a.tap_spikes + (a.s0 + patch_i) * (int64_t)m * 960 : nullptr;
#endif
            const bool fast = !GENERAL || (t0 == 0 && nemit == 4 && tap == nullptr);         // (workgroup-uniform)
            // running maxima of the RAW accumulators per (column sub-block, step), as order-preserving keys (0 = below every key):
            // x -> LeakyReLU(x / 16 + bias) -> integer key is monotone, so the max over the points commutes with it bit for bit —
            // bias, LeakyReLU and the final key once per result
#pragma unroll
            for (int j = 0; j < 6; ++j) BESTL[512 * j] = -__builtin_huge_valf();
            // one third of the patch; TH is compile-time (a generic lambda called with integral constants: x3 is indexed
            // statically and stays in registers), the rounds are a rolled loop
            auto third = [&](auto th_c) {
                constexpr int TH = decltype(th_c)::value;
                const int pt0 = FE_TP * TH + 4 * rg;                    // this thread's first point
                FE_T0();
#pragma unroll
                for (int sl = 0; sl < 4; ++sl)                          // this third's block-3 x0: registers -> this thread's LDS slots
#pragma unroll
                    for (int e = 0; e < 4; ++e) X3L[512 * (4 * sl + e)] = x3[TH][sl][e];
                f32x4 acc[4][6];                                        // [row sub-block: points 4 rs + (lane >> 4), step e][column sub-block]
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 6; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                half8 Wh[3], Wl[3];                                     // the first column half's weight fragments of the next k32 step,
#pragma unroll                                                          // carried from round to round (the second half: loaded on entry)
                for (int j = 0; j < 3; ++j) {
                    Wh[j] = fe_wfrag(mscw, 30, css[j], 0, 0, lane);
                    Wl[j] = fe_wfrag(mscw, 30, css[j], 0, 1, lane);
                }
                FeNP pnext = fe_load_np(a.nprm, 64 * cw + lane);         // a chunk's neuron parameters are loaded one round ahead
                FE_TACC(fe_t_start);
#pragma unroll 1
                for (int R = 0; R < 8; ++R) {
                    const int ch = 2 * R + cw;                          // this wave's chunk (15 = none: round 7, odd waves)
                    const int c = 64 * ch + lane;
                    const NeuronP p = fe_np(pnext);
                    pnext = fe_load_np(a.nprm, c + 128 < 960 ? c + 128 : (c < 960 ? c : 959));
                    FE_T0();
                    float* tp = tap ? tap + c : nullptr;
                    if (ch < 15) {
                        float x[4];
                        if (ch < 7) {                                   // blocks 0-2: x0 from LDS
#pragma unroll
                            for (int e = 0; e < 4; ++e) x[e] = XS[(pt0 + e) * FE_XLD + c];
                        } else {                                        // block 3: this thread's own slots; odd waves emit the block's
                            const float* xs = X3L + 2048 * (R - 4 + cw);    // chunks in rounds 3..6, even waves in rounds 4..7
#pragma unroll
                            for (int e = 0; e < 4; ++e) x[e] = xs[512 * e];
                        }
                        if (ch < 3) {                                   // blocks 0, 1: EIF (wave-uniform branches)
                            if (fast) fe_emit4<1, true>(x, p, 0, 4, pe, pt0, m, count, a.gate, nullptr, 0);
                            else if (GENERAL) fe_emit4<1, false>(x, p, t0, nemit, pe, pt0, m, count, a.gate, tp, tap_tstride);
                        } else {                                        // blocks 2, 3
                            if (fast) fe_emit4<2, true>(x, p, 0, 4, pe, pt0, m, count, a.gate, nullptr, 0);
                            else if (GENERAL) fe_emit4<2, false>(x, p, t0, nemit, pe, pt0, m, count, a.gate, tp, tap_tstride);
                        }
                    }
                    lds_barrier();
                    FE_TACC(fe_t_emit);
                    fe_gemm16<6, 4, 2, true>(R2, FE_PR * 64, mscw, 30, css, lane, acc, 0, 4 * R, R < 7 ? 4 : 2, 30, &Wh, &Wl);
                    lds_barrier();
                    FE_TACC(fe_t_mfma);
                }
                // epilogue of this third: maximum per step over the third's points (row = 4 point + step: register e of a quad is
                // step e) — first over this lane's own points, then over the four lane groups by a reduce-scatter that leaves step g
                // in lane group g (fe_rows_max4: 3 swaps + 3 maxima for the four steps) —, folded into the running maxima
                const bool allv = FE_TP * (TH + 1) <= m;                // (workgroup-uniform: every point of this third exists)
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = -__builtin_huge_valf();                  // (a lane group none of whose points exist: below every real value)
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (allv || FE_TP * TH + 4 * i + (lane >> 4) < m) v[e] = fe_max2(v[e], acc[i][j][e]);
                    }
                    BESTL[512 * j] = fe_max2(BESTL[512 * j], fe_rows_max4(v[0], v[1], v[2], v[3]));
                }
                FE_TACC(fe_t_epi);
            };
            third(std::integral_constant<int, 0>{});
            third(std::integral_constant<int, 1>{});
            third(std::integral_constant<int, 2>{});
            // a wave stores whenever it owns at least one column block of this sweep: a tail wave (cbw < ncb <= cbw + 2) was clamped to
            // the LAST three blocks, of which ncb - cbw are its own and the others are re-stored with the values their owners write
            // (same rows, same K order: identical bits).  emb = 800 / 896 / 1024 have such tails in the second sweep.
            if (cbw < ncb) {
                int lane_e = lane;                                      // (opaque: or the compiler forms the store / bias addresses
                asm volatile("" : "+v"(lane_e));                        //  before the sweep and keeps them — spilled — across it)
                const float* bl = BESTL;
                asm volatile("" : "+v"(bl));
                const int tt = lane_e >> 4, r16e = lane_e & 15;         // this lane's step and column inside a sub-block
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const float bias = a.msc_b[16 * css[j] + r16e];
                    const float raw = bl[512 * j];
                    if (tt < nemit)
                        a.pooled[((int64_t)(t0 + tt) * a.b + patch_i) * a.emb + 16 * css[j] + r16e] = lrelu02(__fadd_rn(__fmul_rn(raw, 0.0625f), bias));
                }
            }
        }
    }
#ifdef FE_STAMPS
    if (a.tap_spikes && threadIdx.x == 0) {
        unsigned long long* st = reinterpret_cast<unsigned long long*>(a.tap_spikes) + (int64_t)blockIdx.x * 32;
        st[17] = fe_t_emit;
        st[18] = fe_t_mfma;
        st[28] = fe_t_start;
        st[29] = fe_t_epi;
        st[19] = __builtin_amdgcn_s_memtime();
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// multi_scale_conv + max over the points for patches of ANY size (round 4): the second half of fd_encoder_kernel as a kernel of
// its own, for the per-stage path (patches of more than 48 points — the reference's default is 100, generation.py:68).  Until
// round 4 that path's neuron kernels wrote the spikes of all T steps as split rows ([T, points, 960]: 6.3 GB at 4000 x 100 points,
// T = 4; 11 GB at the reference's T = 7) for the big-tile GEMM to read back.  All T spikes of a (point, channel) are a function of
// its pre-activation x0 (closed gate), so the neuron kernels now write x0 [points, 960] and the step-0 spikes only, and this
// kernel regenerates every step where it is consumed: one workgroup per patch, the patch in tiles of 16 points with the 4 steps of
// a group stacked as rows (64 rows), every wave 96 of all emb columns, K in rounds of 128 columns — fd_encoder_kernel's scheme,
// with x0 read from global memory (L2: the neuron kernels have just written it) one round ahead instead of from LDS.  Same
// split-f16 products in the same order, same neuron arithmetic, same maximum: bit-identical to the T-slab path
// (tests: test_fd_x0_path_equals_the_spike_slab_path_bit_for_bit).  LDS: the 32 KiB panel + 12 KiB of running maxima.
// ---------------------------------------------------------------------------------------------
constexpr int FM_OFF_BEST = 32768;
constexpr int FM_LDS = FM_OFF_BEST + 6 * 4 * 8 * 16 * 4;

template <bool GENERAL>
__global__ __launch_bounds__(FE_NT, 1) void fd_msc_kernel(const FdMscArgs a) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char* R2 = smem;
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = a.m;
    const int64_t patch_i = blockIdx.x;
    const float* __restrict__ xp = a.x0 + patch_i * m * 960;
    const int cw = w & 1, rg = w >> 1;
    const int nsweep = a.emb / 768 + (a.emb % 768 ? 1 : 0);
    const int ntile = (m + 15) >> 4;
    const half8* __restrict__ mscw = reinterpret_cast<const half8*>(a.msc_wp);
    const int64_t tap_tstride = a.b_total * (int64_t)m * 960;
    unsigned char* pe[4];                                   // panel addresses: point 4 rg + e of the tile (row 4 point), column 64 cw + lane
#pragma unroll
    for (int e = 0; e < 4; ++e) pe[e] = R2 + fe_panel_off<FE_PR>(4 * (4 * rg + e), 64 * cw + lane);
    float* BESTL = reinterpret_cast<float*>(smem + FM_OFF_BEST) + (lane >> 4) * 128 + w * 16 + (lane & 15);   // (j, step = this lane's row): BESTL[512 j]
    for (int sweep = 0; sweep < nsweep; ++sweep) {
        const int ncb = a.emb / 32;
        const int cbw = sweep * 24 + 3 * w;
        const int cb0 = cbw + 2 < ncb ? cbw : (ncb >= 3 ? ncb - 3 : 0);
        const int css[6] = {2 * cb0, 2 * cb0 + 1, 2 * cb0 + 2, 2 * cb0 + 3, 2 * cb0 + 4, 2 * cb0 + 5};
        for (int t0 = 0; t0 < a.T; t0 += 4) {
            const int nemit = a.T - t0 < 4 ? a.T - t0 : 4;
            const bool count = sweep == 0 && t0 + nemit == a.T;        // the gate check runs once, over all T steps
            float* tap = (GENERAL && a.tap_spikes && sweep == 0) ? a.tap_spikes + (a.s0 + patch_i) * (int64_t)m * 960 : nullptr;
            const bool fast = !GENERAL || (t0 == 0 && nemit == 4 && tap == nullptr);         // (workgroup-uniform)
#pragma unroll
            for (int j = 0; j < 6; ++j) BESTL[512 * j] = -__builtin_huge_valf();
#pragma unroll 1
            for (int tile = 0; tile < ntile; ++tile) {
                // a last, partly filled tile (m = 100: 4 of 16 points) only multiplies the pairs of 16-row sub-blocks that hold points, and
                // only the waves whose four points exist emit
                const int npt = m - 16 * tile < 16 ? m - 16 * tile : 16;
                const int nrs = (npt + 3) >> 2;                         // row sub-blocks in use: points 4 rs .. 4 rs + 3, four steps each
                const int pt0 = 16 * tile + 4 * rg;                     // this thread's first point; rows beyond the patch: the last point
                const float* xr[4];                                     // (emitted like any other, never looked at: masked in the maximum)
#pragma unroll
                for (int e = 0; e < 4; ++e) xr[e] = xp + (int64_t)(pt0 + e < m ? pt0 + e : m - 1) * 960 + lane;
                f32x4 acc[4][6];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 6; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                half8 Wh[6], Wl[6];                                     // (203 registers without x0 in them: room for the whole k32 step)
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    Wh[j] = fe_wfrag(mscw, 30, css[j], 0, 0, lane);
                    Wl[j] = fe_wfrag(mscw, 30, css[j], 0, 1, lane);
                }
                FeNP pnext = fe_load_np(a.nprm, 64 * cw + lane);         // a chunk's neuron parameters and x0 values: one round ahead
                float xn[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) xn[e] = xr[e][64 * cw];
#pragma unroll 1
                for (int R = 0; R < 8; ++R) {
                    const int ch = 2 * R + cw;                          // this wave's chunk (15 = none: round 7, odd waves)
                    const int c = 64 * ch + lane;
                    const NeuronP p = fe_np(pnext);
                    float x[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[e] = xn[e];
                    const int chn = ch + 2 < 15 ? ch + 2 : (ch < 15 ? ch : 14);     // (the last rounds re-load their own chunk: unconditional loads)
                    pnext = fe_load_np(a.nprm, 64 * chn + lane);
#pragma unroll
                    for (int e = 0; e < 4; ++e) xn[e] = xr[e][64 * chn];
                    float* tp = tap ? tap + c : nullptr;
                    if (ch < 15 && 4 * rg < npt) {
                        if (ch < 3) {                                   // blocks 0, 1: EIF (wave-uniform branches)
                            if (fast) fe_emit4<1, true>(x, p, 0, 4, pe, pt0, m, count, a.gate, nullptr, 0);
                            else if (GENERAL) fe_emit4<1, false>(x, p, t0, nemit, pe, pt0, m, count, a.gate, tp, tap_tstride);
                        } else {                                        // blocks 2, 3: LIF
                            if (fast) fe_emit4<2, true>(x, p, 0, 4, pe, pt0, m, count, a.gate, nullptr, 0);
                            else if (GENERAL) fe_emit4<2, false>(x, p, t0, nemit, pe, pt0, m, count, a.gate, tp, tap_tstride);
                        }
                    }
                    lds_barrier();
                    fe_gemm16<6, 4, 2, false>(R2, FE_PR * 64, mscw, 30, css, lane, acc, 0, 4 * R, R < 7 ? 4 : 2, 30, &Wh, &Wl, (nrs + 1) >> 1);
                    lds_barrier();
                }
                const bool allv = 16 * (tile + 1) <= m;                 // (workgroup-uniform: every point of this tile exists)
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = -__builtin_huge_valf();
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (allv || 16 * tile + 4 * i + (lane >> 4) < m) v[e] = fe_max2(v[e], acc[i][j][e]);
                    }
                    BESTL[512 * j] = fe_max2(BESTL[512 * j], fe_rows_max4(v[0], v[1], v[2], v[3]));
                }
            }
            if (cbw < ncb) {
                int lane_e = lane;
                asm volatile("" : "+v"(lane_e));
                const float* bl = BESTL;
                asm volatile("" : "+v"(bl));
                const int tt = lane_e >> 4, r16e = lane_e & 15;
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const float bias = a.msc_b[16 * css[j] + r16e];
                    const float raw = bl[512 * j];
                    if (tt < nemit)
                        a.pooled[((int64_t)(t0 + tt) * a.b + patch_i) * a.emb + 16 * css[j] + r16e] = lrelu02(__fadd_rn(__fmul_rn(raw, 0.0625f), bias));
                }
            }
        }
    }
}

bool fd_msc_ok(int m, int emb, int T) { return m >= 1 && emb % 32 == 0 && emb >= 96 && T >= 1; }

int launch_fd_msc(const FdMscArgs& a, hipStream_t st) {
    if (a.b == 0) return SAPCU_OK;
    SAPCU_CHECK_ARG(fd_msc_ok(a.m, a.emb, a.T) && a.b < 0x7fffffffLL && a.x0 && a.msc_wp && a.msc_b && a.nprm && a.pooled && a.gate,
                    "fd_msc: unsupported shape or null operand (m=%d emb=%d T=%d)", a.m, a.emb, a.T);
    if (a.T == 4 && a.tap_spikes == nullptr) {
        hipLaunchKernelGGL(fd_msc_kernel<false>, dim3((unsigned)a.b), dim3(FE_NT), FM_LDS, st, a);
    } else {
        hipLaunchKernelGGL(fd_msc_kernel<true>, dim3((unsigned)a.b), dim3(FE_NT), FM_LDS, st, a);
    }
    SAPCU_CHECK_LAUNCH();
    return SAPCU_OK;
}

// ---------------------------------------------------------------------------------------------
// model-build helpers: weights in fragment order, clamped neuron parameters
// ---------------------------------------------------------------------------------------------
// out[((cs * nk32 + s) * 2 + plane) * 64 + lane][j] = w16_plane[16 cs + (lane & 15)][32 s + 8 (lane >> 4) + j],  w [n, k] row-major
__global__ __launch_bounds__(256) void pack_frag_weights_kernel(const _Float16* __restrict__ hi, const _Float16* __restrict__ lo,
                                                                int n, int k, _Float16* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int nk32 = k / 32;
    const int64_t total = (int64_t)(n / 16) * nk32 * 2 * 64;
    if (t >= total) return;
    const int lane = (int)(t & 63);
    const int plane = (int)((t >> 6) & 1);
    const int64_t cs_s = t >> 7;
    const int s = (int)(cs_s % nk32), cs = (int)(cs_s / nk32);
    const _Float16* src = (plane ? lo : hi) + (int64_t)(16 * cs + (lane & 15)) * k + 32 * s + 8 * (lane >> 4);
    *reinterpret_cast<half8*>(out + t * 8) = *reinterpret_cast<const half8*>(src);
}

int launch_pack_frag_weights(const void* w16_hi, const void* w16_lo, int n, int k, void* out, hipStream_t st) {
    SAPCU_CHECK_ARG(n % 16 == 0 && k % 32 == 0, "pack_frag_weights: need n %% 16 == 0 and k %% 32 == 0 (n=%d k=%d)", n, k);
    const int64_t total = (int64_t)(n / 16) * (k / 32) * 2 * 64;
    hipLaunchKernelGGL(pack_frag_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const _Float16*)w16_hi,
                       (const _Float16*)w16_lo, n, k, (_Float16*)out);
    SAPCU_CHECK_LAUNCH();
    return SAPCU_OK;
}

// out[(coff + c) * 8 + ..] = clamped (decay, adapt, rdecay, theta0, dT, rh, 0, 0) of channel c of a raw [4 | 6][C] parameter block
__global__ __launch_bounds__(256) void pack_fd_neuron_kernel(const float* __restrict__ raw, int C, int eif, int coff, float* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const NeuronP p = eif ? load_eif(raw, C, c) : load_lif(raw, C, c);
    float* o = out + (int64_t)(coff + c) * 8;
    o[0] = p.decay; o[1] = p.adapt; o[2] = p.rdecay; o[3] = p.theta0; o[4] = p.dT; o[5] = p.rh; o[6] = 0.f; o[7] = 0.f;
}

int launch_pack_fd_neuron(const float* raw, int C, int eif, int coff, float* out, hipStream_t st) {
    hipLaunchKernelGGL(pack_fd_neuron_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, st, raw, C, eif, coff, out);
    SAPCU_CHECK_LAUNCH();
    return SAPCU_OK;
}

bool fd_encoder_ok(int m, int nscale, int emb, int T) { return m >= 1 && m <= FE_M && nscale >= 1 && nscale <= 4 && emb % 32 == 0 && emb >= 96 && T >= 1; }

int launch_fd_encoder(const FdEncArgs& a, hipStream_t st) {
    if (a.b == 0) return SAPCU_OK;
    SAPCU_CHECK_ARG(fd_encoder_ok(a.m, a.nscale, a.emb, a.T), "fd_encoder: unsupported shape m=%d scales=%d emb=%d", a.m, a.nscale, a.emb);
    SAPCU_CHECK_ARG(a.kk >= 1 && a.kk <= a.m && a.kmax0 >= 1 && a.kmax0 <= a.m && a.b < 0x7fffffffLL, "fd_encoder: bad neighbour counts");
#ifdef FE_STAMPS
    const bool production = a.T == 4;                        // (diagnostic build: the spikes tap carries the stamps)
#else
    const bool production = a.T == 4 && a.tap_spikes == nullptr;
#endif
    if (production) {                                        // the scratch-free kernel
        static DeviceOnce lds_once;
        SAPCU_SET_MAX_LDS(lds_once, (&fd_encoder_kernel<false>), FE_LDS);
        hipLaunchKernelGGL(fd_encoder_kernel<false>, dim3((unsigned)a.b), dim3(FE_NT), FE_LDS, st, a);
    } else {
        static DeviceOnce lds_once;
        SAPCU_SET_MAX_LDS(lds_once, (&fd_encoder_kernel<true>), FE_LDS);
        hipLaunchKernelGGL(fd_encoder_kernel<true>, dim3((unsigned)a.b), dim3(FE_NT), FE_LDS, st, a);
    }
    SAPCU_CHECK_LAUNCH();
    return SAPCU_OK;
}

}  // namespace sapcu
