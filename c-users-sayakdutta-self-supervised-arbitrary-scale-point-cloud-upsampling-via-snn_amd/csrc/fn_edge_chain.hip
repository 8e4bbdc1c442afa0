// fn transformer block: the whole per-edge chain of MultiHeadSNNTransformerBlock.forward (fn/snn_coder.py:355-389) in ONE
// kernel, for all three blocks (d = 128, 256, 512):
//
//     pe1 = LIF(fc_delta(x_i - x_j))          VALU                       fn:310,355-358
//     pe  = LIF(fc_delta2(pe1))               GEMM 1 + neuron epilogue   fn:360-363
//     ain = q_i - k_j + pe                    (same epilogue)            fn:368
//     g   = LIF(fc_gamma(ain))                GEMM 2 + neuron epilogue   fn:373-376
//     a   = fc_gamma2(g)                      GEMM 3                     fn:378
//     res = sum_j softmax_j(a / sqrt(hd)) * (v_j + pe)                   fn:379-389
//
// The unfused form (model.hip: fn_pe1 -> gemm<EPI_LIF_ATTN> -> gemm<EPI_LIF> -> gemm<EPI_BIAS> -> fn_softmax_agg) moves every
// [rows, d] tensor through HBM: ten passes of rows*d*4 bytes per block.  Here a workgroup owns a GROUP of whole points
// (floor(ROWS / kk) points; ROWS = 96 / 128 / 64 edge rows at d = 128 / 256 / 512) and keeps the group's activation panel in LDS
// between the GEMMs, as the split-f16 A operand (hi | lo planes, [k32 step][plane][ROWS][32 halves], 16-byte chunks
// XOR-swizzled by (row>>2)&3 — the operand-slot layout of gemm_sf16_bt.hip, so the fragment reads are the same conflict-free
// ds_read_b128).  Nothing of the chain reaches HBM: the kernel reads xyz differences + neighbour rows (24 B per edge row), the
// q / k / v rows of the patch (L2) and the pre-packed weights (L2), and writes res [points, d].
//
// Shape of the work.  d/32 waves per workgroup (d = 512: d/64); wave w owns 32 (64) output columns of all three GEMMs and all ROWS rows.
// A wave's weight fragments are not shared with any other wave, so they bypass LDS: pre-packed at model build in fragment order (one
// contiguous KiB per (column sub-block, k32 step, plane)), streamed L2 -> registers half a k32 step ahead.  No barrier inside a GEMM;
// six workgroup barriers per group.  The GEMMs issue v_mfma_f32_16x16x32_f16 (see ChainShape below: why, and why the results still equal
// the unfused chain's bit for bit).  The epilogues run in the accumulator layout (lane = column: bias and neuron parameters are
// per-lane constants; register e of sub-block (rs, cs) = row 16 rs + 4 (lane >> 4) + e), pe stays in registers until the aggregation,
// and the panel's rows are SLOTS dealt so that a point's kk rows sit in ONE lane group: the per-point softmax of the first four points
// of a group reads nothing but its own lane's registers — every value equals the unfused chain's bit for bit (same split-f16 products
// in the same order, same neuron arithmetic, same softmax order).
// d = 128: 256-thread workgroups, 51 KiB of LDS, three per CU.  d = 256: 512 threads, 131 KiB, one per CU.  d = 512: 512 threads (wave
// tile 64 x 64), 131 KiB.  What bounds them (each pipe at its practical rate, the kernel time their sum) and the overlap designs that
// were measured without gain: DESIGN.md section 4.1c.
#include "common.h"
#include "gemm_epi.h"
#include "ops.h"

namespace sapcu {
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// Shape of a workgroup's work.  ROWS = MFMA rows per group: d = 128: 96 rows = 4 points of 24 neighbours (no idle row), d = 256: 128
// rows = 7 points of 18, d = 512: 64 rows = 5 points of 12 (a 128-row panel of 512 columns would be 256 KiB).  One plane of one k32
// step is [ROWS][32 halves]; a k32 step = hi plane | lo plane.
// History of the 32x32x16 form this replaced (ms per launch at 4096 x 48 points) — d = 512: 16 waves 17.3; 8 waves, exact wait counts
// in the rolled k loop, epilogue units of 8 elements 16.3; t = v_j + pe pinned to the first epilogue (no scratch) + softmax in point
// pairs + packed panel conversions + 32-bit gather offsets 15.2; this form 13.7.  d = 256: 8.27 -> 8.10 -> 7.56 -> 7.4.  d = 128:
// 4.64 -> 4.42 -> 3.93 -> 3.8 (three workgroups per CU: the k32 step's operand fragments half the row sub-blocks at a time, 160 registers).
#ifndef SAPCU_CHAIN_LB128
#define SAPCU_CHAIN_LB128 3
#endif
// Round 3, late: the GEMMs issue v_mfma_f32_16x16x32_f16.  Same flops per cycle as the 32x32x16 shape, but the chip holds a higher
// clock on it (profiles/micro/mfma_shape.hip: 1.82 against 1.60 GHz, two waves per SIMD, random operands), and one 16x16x32 equals
// two chained 32x32x16 over the same 32 k values BIT FOR BIT (profiles/micro/mfma_f16_shapes_bits.hip: 0 of 51 200 outputs differ)
// — so the unfused chain reproduces this kernel exactly by issuing its three split-f16 products per k32 step instead of per k16
// step (gemm_sf16_ring.hip, pass-major form).  Accumulator layout of the shape: a 16 x 16 block = 4 registers per lane, lane l ->
// column l & 15, rows 4 (l >> 4) + e: a wave tile ROWS x 32 CB is RS = ROWS / 16 row sub-blocks x CS = 2 CB column sub-blocks.
// SLOTS.  A row of the panel is a SLOT, and which edge row sits in it is the kernel's choice: lane group g = l >> 4 owns the rows
// 16 rs + 4 g + e, so point g of the group (g = 0..3) gets the first kk of lane group g's slots, in neighbour order — its whole
// softmax-aggregate then runs on that lane group's own registers, no cross-lane traffic — and the slots that are left (d = 512: 4 per
// lane group, d = 256: 14, d = 128: none) hold the group's remaining points (1 / 3 / 0) in lane-group-major order.
template <int D> struct ChainShape {
    static constexpr int ROWS = D == 128 ? 96 : (D == 256 ? 128 : 64);
    static constexpr int RS = ROWS / 16;              // 16-row sub-blocks of the wave tile
    static constexpr int CB = D == 512 ? 2 : 1;       // 32 columns per wave x CB
    static constexpr int CS = 2 * CB;                 // 16-column sub-blocks per wave
    static constexpr int NW = D / (32 * CB);          // waves per workgroup
    static constexpr int PLANE = ROWS * 64;
    static constexpr int KSTEP = 2 * PLANE;
    static constexpr int LDS = ROWS * D * 4 + ROWS * 16 + ROWS * 8;
    static constexpr int US = 8;                      // elements per lane of one epilogue unit: the quads of two row sub-blocks
};

// slot (rs, g, e) -> (point of the group, neighbour); point < 0: unused slot
template <int D, int KK> struct ChainSlots {
    static constexpr int GS = 4 * ChainShape<D>::RS;          // slots per lane group
    static constexpr int SP = GS - KK;                        // spare slots per lane group
    static constexpr int NUA = (4 * SP) / KK;                 // points living in spare slots
    static constexpr int PPG = 4 + NUA;                       // points per group
    static constexpr int point(int rs, int g, int e) {
        const int t = 4 * rs + e;
        if (t < KK) return g;
        const int p = 4 + (g * SP + t - KK) / KK;
        return p < PPG ? p : -1;
    }
    static constexpr int nbr(int rs, int g, int e) {
        const int t = 4 * rs + e;
        return t < KK ? t : (g * SP + t - KK) % KK;
    }
    // where neighbour jj of spare-slot point n (= point 4 + n) sits
    static constexpr int SP1 = SP > 0 ? SP : 1;
    static constexpr int ua_g(int n, int jj) { return (n * KK + jj) / SP1; }
    static constexpr int ua_rs(int n, int jj) { return (KK + (n * KK + jj) % SP1) >> 2; }
    static constexpr int ua_e(int n, int jj) { return (KK + (n * KK + jj) % SP1) & 3; }
};

// ---------------------------------------------------------------------------------------------
// edge preparation: per edge row (point i, neighbour slot j) the rows of point i and of its neighbour in the [points, .]
// tensors, and the position difference x_i - x_j (fn:310) — one coalesced 24-byte record per row for the chain kernel.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void edge_prep_kernel(const float* __restrict__ patch, const int32_t* __restrict__ idx,
                                                        int64_t rows, int m, int kk, int2* __restrict__ tab,
                                                        float4* __restrict__ pd) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const int64_t pt = r / kk;
    const int64_t nrow = (pt / m) * m + idx[r];
    tab[r] = make_int2((int)pt, (int)nrow);
    const float* pi = patch + pt * 3;
    const float* pj = patch + nrow * 3;
    pd[r] = make_float4(__fsub_rn(pi[0], pj[0]), __fsub_rn(pi[1], pj[1]), __fsub_rn(pi[2], pj[2]), 0.f);
}

// weights in fragment order of the 16x16x32 B operand:
//   out[((cs * nk32 + s) * 2 + plane) * 64 + lane][j] = w16_plane[16 cs + (lane & 15)][32 s + 8 (lane >> 4) + j]
__global__ __launch_bounds__(256) void pack_chain_weights_kernel(const _Float16* __restrict__ hi, const _Float16* __restrict__ lo,
                                                                 int d, _Float16* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // one 8-half fragment piece per thread
    const int nk32 = d / 32;
    const int64_t total = (int64_t)(d / 16) * nk32 * 2 * 64;
    if (t >= total) return;
    const int lane = (int)(t & 63);
    const int plane = (int)((t >> 6) & 1);
    const int64_t cs_s = t >> 7;
    const int s = (int)(cs_s % nk32), cs = (int)(cs_s / nk32);
    const _Float16* src = (plane ? lo : hi) + (int64_t)(16 * cs + (lane & 15)) * d + 32 * s + 8 * (lane >> 4);
    *reinterpret_cast<half8*>(out + t * 8) = *reinterpret_cast<const half8*>(src);
}

// the 16 values of lane group `s` (lanes 16 s .. 16 s + 15) of x, in all four lane groups (s compile-time)
__device__ __forceinline__ float group_bcast(float x, int s) {
    const unsigned a = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane16_swap(a, a, false, false);     // r[0] = groups [0,0,2,2], r[1] = [1,1,3,3]
    const unsigned z = (s & 1) ? r[1] : r[0];
    const auto q = __builtin_amdgcn_permlane32_swap(z, z, false, false);     // q[0] = [lo,lo], q[1] = [hi,hi]
    return __uint_as_float((s & 2) ? q[1] : q[0]);
}

template <int CS>
struct ChainLane {       // per-lane constants of the epilogues
    int c16, g;
    int col[CS];         // this lane's column of column sub-block cs
    unsigned xw[CS];     // LDS byte offset of (row 4 g, column col[cs] as k); + rs * 1024 + e * 64
};

template <int CS>
struct ChainW {           // weight fragments of one k32 step x CS column sub-blocks (hi, lo)
    half8 wh[CS], wl[CS];
};

// fragment (column sub-block cs, k32 step s, plane) of a packed matrix: [(cs * NK32 + s) * 2 + plane][64 lanes]
template <int D>
__device__ __forceinline__ half8 chain_w_frag(const half8* __restrict__ wp, int cs, int s, int plane, int lane) {
    return wp[(((int64_t)cs * (D / 32) + s) * 2 + plane) * 64 + lane];
}

// first k32 step of a GEMM's weight stream: issued well before the GEMM so that its L2 latency is covered
template <int D>
__device__ __forceinline__ void chain_w_prefetch(const half8* __restrict__ wp, int cs0, int lane, ChainW<ChainShape<D>::CS>& W) {
#pragma unroll
    for (int j = 0; j < ChainShape<D>::CS; ++j) {
        W.wh[j] = chain_w_frag<D>(wp, cs0 + j, 0, 0, lane);
        W.wl[j] = chain_w_frag<D>(wp, cs0 + j, 0, 1, lane);
    }
}

// acc[rs][cs] (+)= panel . W^T for this wave's columns.  Per k32 step the three split-f16 products (a_lo.w_hi, a_hi.w_lo, a_hi.w_hi) go
// into each accumulator in that order — the order of the pass-major ring kernel.  The column sub-blocks are taken in two halves:
// the weight fragments of a half are refilled IN PLACE with the next step's right behind the MFMAs that read them (unconditionally:
// the last step re-loads its own), so a fragment is in flight for half a step while the other half's MFMAs run.
template <int D>
__device__ __forceinline__ void chain_gemm(const unsigned char* X, const half8* __restrict__ wp, int cs0, int lane,
                                           ChainW<ChainShape<D>::CS>& W, f32x4 (&acc)[ChainShape<D>::RS][ChainShape<D>::CS]) {
    constexpr int NK32 = D / 32, RS = ChainShape<D>::RS, CS = ChainShape<D>::CS, HC = CS / 2;
    constexpr int CH_PLANE = ChainShape<D>::PLANE, CH_KSTEP = ChainShape<D>::KSTEP;
    const int c16 = lane & 15, g = lane >> 4;
    const unsigned char* xa = X + c16 * 64 + ((g ^ ((c16 >> 2) & 3)) * 16);      // A operand: row 16 rs + c16, k chunk g
#pragma unroll
    for (int i = 0; i < RS; ++i)
#pragma unroll
        for (int j = 0; j < CS; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (D == 128) {
        // d = 128 (three workgroups per CU: <= 170 registers): the operand fragments of HALF the row sub-blocks at a time (24 registers
        // instead of 48); a column sub-block's weight fragments are refilled behind the second half's MFMAs that read them — only a
        // quarter step of lead, which the three waves of a SIMD cover for each other.
        constexpr int RH = RS / 2;
#pragma unroll 1
        for (int s = 0; s < NK32; ++s) {
            const int sn = s + 1 < NK32 ? s + 1 : NK32 - 1;
#pragma unroll
            for (int rc = 0; rc < 2; ++rc) {
                half8 ah[RH], al[RH];
#pragma unroll
                for (int i = 0; i < RH; ++i) {
                    ah[i] = *reinterpret_cast<const half8*>(xa + s * CH_KSTEP + (rc * RH + i) * 1024);
                    al[i] = *reinterpret_cast<const half8*>(xa + s * CH_KSTEP + (rc * RH + i) * 1024 + CH_PLANE);
                }
#pragma unroll
                for (int hc = 0; hc < 2; ++hc) {
#pragma unroll
                    for (int i = 0; i < RH; ++i)
#pragma unroll
                        for (int j = hc * HC; j < (hc + 1) * HC; ++j)
                            acc[rc * RH + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], W.wh[j], acc[rc * RH + i][j], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < RH; ++i)
#pragma unroll
                        for (int j = hc * HC; j < (hc + 1) * HC; ++j)
                            acc[rc * RH + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], W.wl[j], acc[rc * RH + i][j], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < RH; ++i)
#pragma unroll
                        for (int j = hc * HC; j < (hc + 1) * HC; ++j)
                            acc[rc * RH + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], W.wh[j], acc[rc * RH + i][j], 0, 0, 0);
                    if (rc == 1) {
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int j = hc * HC; j < (hc + 1) * HC; ++j) {
                            W.wh[j] = chain_w_frag<D>(wp, cs0 + j, sn, 0, lane);
                            W.wl[j] = chain_w_frag<D>(wp, cs0 + j, sn, 1, lane);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        return;
    }
#pragma unroll 1
    for (int s = 0; s < NK32; ++s) {
        half8 ah[RS], al[RS];
#pragma unroll
        for (int i = 0; i < RS; ++i) {
            ah[i] = *reinterpret_cast<const half8*>(xa + s * CH_KSTEP + i * 1024);
            al[i] = *reinterpret_cast<const half8*>(xa + s * CH_KSTEP + i * 1024 + CH_PLANE);
        }
        const int sn = s + 1 < NK32 ? s + 1 : NK32 - 1;
#pragma unroll
        for (int hc = 0; hc < 2; ++hc) {
#pragma unroll
            for (int i = 0; i < RS; ++i)
#pragma unroll
                for (int j = hc * HC; j < (hc + 1) * HC; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], W.wh[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < RS; ++i)
#pragma unroll
                for (int j = hc * HC; j < (hc + 1) * HC; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], W.wl[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < RS; ++i)
#pragma unroll
                for (int j = hc * HC; j < (hc + 1) * HC; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], W.wh[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = hc * HC; j < (hc + 1) * HC; ++j) {
                W.wh[j] = chain_w_frag<D>(wp, cs0 + j, sn, 0, lane);
                W.wl[j] = chain_w_frag<D>(wp, cs0 + j, sn, 1, lane);
            }
        }
    }
}

// two elements of one column, rows e and e + 1 of a register quad, into the panel as the split-f16 operand of the next GEMM: the
// packed conversion of gfx950 for the hi halves (v_cvt_pk_f16_f32, round-to-nearest-even like the scalar form), and each lo half
// = f16(v - hi) as ONE instruction: v_fma_mixlo/mixhi_f16 read the f16 half in place, form -hi + v exactly (v - hi is exact in
// f32 anyway) and round once to f16 — the bits of cvt(f32(v) - f32(hi)); the compiler does not select them here on its own
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
template <int PLANE, int CS>
__device__ __forceinline__ void chain_put2(unsigned char* X, const ChainLane<CS>& L, int cs, int rs, int e, float v0, float v1) {
    unsigned char* p = X + L.xw[cs] + (unsigned)(rs * 1024 + e * 64);
    const f32x2 v = f32x2{v0, v1};
    const half2v hi = __builtin_convertvector(v, half2v);
    half2v lo;
    {
        unsigned lp;
        const unsigned hp = __builtin_bit_cast(unsigned, hi);
        asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lp) : "v"(hp), "v"(v.x));
        asm volatile("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lp) : "v"(hp), "v"(v.y));
        lo = __builtin_bit_cast(half2v, lp);
    }
    *reinterpret_cast<_Float16*>(p) = hi.x;
    *reinterpret_cast<_Float16*>(p + 64) = hi.y;
    *reinterpret_cast<_Float16*>(p + PLANE) = lo.x;
    *reinterpret_cast<_Float16*>(p + PLANE + 64) = lo.y;
}

// store_split (gemm_epi.h) with non-temporal stores: the result rows are streamed out once and must not push the weights and the
// k / v rows out of L2 (measured at d = 512 together with the non-temporal q loads: 29 % fewer L2 misses per launch)
__device__ __forceinline__ void chain_store_split_nt(float* base, int64_t row, int ld, int col, float v) {
    _Float16* rp = reinterpret_cast<_Float16*>(base + row * ld);
    const _Float16 hi = (_Float16)v;
    __builtin_nontemporal_store(hi, &rp[split_hi_index(ld, col)]);
    __builtin_nontemporal_store((_Float16)(v - (float)hi), &rp[split_lo_index(ld, col)]);
}

__device__ __forceinline__ NeuronP chain_lif(const float* __restrict__ lif, int d, int col) {
    NeuronP np;
    np.decay = clampf(lif[col], 0.1f, 0.99f);
    np.adapt = clampf(lif[d + col], 0.001f, 0.1f);
    np.rdecay = clampf(lif[2 * d + col], 0.1f, 0.95f);
    np.theta0 = lif[3 * d + col];
    np.dT = 0.f;
    np.rh = 0.f;
    return np;
}

// select by this lane's group g = 2 g1 + g0 among four values (compile-time equal ones fold away)
__device__ __forceinline__ float group_select(float v0, float v1, float v2, float v3, bool g0, bool g1) {
    const float a = g0 ? v1 : v0, b = g0 ? v3 : v2;
    return g1 ? b : a;
}

// per-point softmax over the kk neighbours and aggregation with t = v_j + pe, fn_softmax_agg_kernel's operation order
template <int KK>
__device__ __forceinline__ float chain_softmax_agg(float (&xs)[KK], const float (&ts)[KK]) {
    float mx = -__builtin_huge_valf();
#pragma unroll
    for (int jj = 0; jj < KK; ++jj) mx = fmaxf(mx, xs[jj]);
    // (kk is even: the element-wise steps — subtract, scale into exp2's argument, scale by 1 / den — on operand pairs; the two sums stay
    // sequential in neighbour order)
    float den = 0.f;
    const f32x2 mx2 = f32x2{mx, mx}, l2e = f32x2{1.4426950408889634074f, 1.4426950408889634074f};
#pragma unroll
    for (int jj = 0; jj < KK; jj += 2) {
        const f32x2 e2 = (f32x2{xs[jj], xs[jj + 1]} - mx2) * l2e;      // fast_exp(x - mx) = exp2((x - mx) * log2 e)
        xs[jj] = __builtin_amdgcn_exp2f(e2.x);
        xs[jj + 1] = __builtin_amdgcn_exp2f(e2.y);
        den = __fadd_rn(den, xs[jj]);
        den = __fadd_rn(den, xs[jj + 1]);
    }
    const float inv_den = __fdiv_rn(1.0f, den);
    const f32x2 inv2 = f32x2{inv_den, inv_den};
    float out = 0.f;
#pragma unroll
    for (int jj = 0; jj < KK; jj += 2) {
        const f32x2 w2 = f32x2{xs[jj], xs[jj + 1]} * inv2;
        out = __fmaf_rn(w2.x, ts[jj], out);
        out = __fmaf_rn(w2.y, ts[jj + 1], out);
    }
    return out;
}

// threads = 64 x (d / 32 / CB).  d = 128: three 256-thread workgroups per CU (51 KiB of LDS each, <= 170 registers per wave).
// O32: the q|k|v tensor is smaller than 4 GiB — the gathers address it as a scalar base + a 32-bit byte offset per lane (one add
// per gathered element; the 64-bit form spends a 64-bit multiply-add and two adds on each).
template <int D, int KK, bool O32>
__global__ __launch_bounds__(ChainShape<D>::NW * 64, (D == 128 ? SAPCU_CHAIN_LB128 : 1)) void fn_edge_chain_kernel(const ChainArgs a) {
    using S = ChainShape<D>;
    using M = ChainSlots<D, KK>;
    constexpr int CH_ROWS = S::ROWS, RS = S::RS, CS = S::CS, CH_PLANE = S::PLANE, CH_KSTEP = S::KSTEP;
    constexpr int PPG = M::PPG, NUA = M::NUA, SP = M::SP;
    constexpr int NUP = RS / 2;                            // epilogue units per column sub-block: the quads of row sub-blocks 2 up, 2 up + 1
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char* X = smem;
    // position differences of the group's slots, one array per coordinate: the four rows of a register quad are 16 contiguous bytes,
    // and two neighbouring rows one operand pair of the packed arithmetic
    float* pdx = reinterpret_cast<float*>(smem + CH_ROWS * D * 4);
    float* pdy = pdx + CH_ROWS;
    float* pdz = pdy + CH_ROWS;
    int2* rinfo = reinterpret_cast<int2*>(smem + CH_ROWS * D * 4 + CH_ROWS * 16);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cs0 = w * CS;                                // first column sub-block of this wave
    ChainLane<CS> L;
    L.c16 = lane & 15;
    L.g = lane >> 4;
    const bool g0 = (L.g & 1) != 0, g1 = (L.g & 2) != 0;
    // element (row = 16 rs + 4 g + e, k = col): k32 step = col / 32, chunk = (col % 32) / 8, half = col % 8; (row >> 2) & 3 = g
#pragma unroll
    for (int j = 0; j < CS; ++j) {
        L.col[j] = 16 * (cs0 + j) + L.c16;
        L.xw[j] = (unsigned)((L.col[j] >> 5) * CH_KSTEP + L.g * 256 + ((((L.col[j] >> 3) & 3) ^ L.g) * 16) + (L.col[j] & 7) * 2);
    }
    const int rowg = 4 * L.g;                              // this lane's rows: 16 rs + rowg + e

    // group of this workgroup: contiguous ranges of groups per XCD (blockIdx & 7), so that the tiles of one patch — which
    // gather the same q / k / v rows — share an L2
    const int64_t ngroups = (a.P + PPG - 1) / PPG;
    int64_t grp;
    {
        const int64_t nx = gridDim.x < 8 ? 1 : 8;
        const int64_t x = nx == 1 ? 0 : (blockIdx.x & 7), slot = nx == 1 ? blockIdx.x : (blockIdx.x >> 3);
        const int64_t qd = ngroups / nx, rem = ngroups % nx;
        grp = x * qd + (x < rem ? x : rem) + slot;
        if (slot >= qd + (x < rem ? 1 : 0)) return;
    }
    const int64_t pt0 = grp * PPG;
    const int npts = (int)((a.P - pt0) < PPG ? (a.P - pt0) : PPG);

    f32x4 acc[RS][CS], pe[RS][CS];
    ChainW<CS> W;
    // ---- phase 0: edge records of the group's slots; pe1 = LIF(fc_delta(x_i - x_j)) -> panel              fn:310,355-358
    if (tid < CH_ROWS) {
        const int rs = tid >> 4, sg = (tid >> 2) & 3, e = tid & 3, t = 4 * rs + e;
        int p, jj;
        if (t < KK) {
            p = sg;
            jj = t;
        } else {
            const int u = sg * SP + t - KK;
            p = 4 + u / KK;
            jj = u % KK;
        }
        const bool ok = p < npts;                                            // unused slots replay the group's first edge row
        const int64_t er = ok ? (pt0 + p) * KK + jj : pt0 * KK;
        const int2 t2 = a.tab[er];
        if (O32) reinterpret_cast<unsigned*>(rinfo)[tid] = (unsigned)t2.y * (unsigned)(a.ldq * 4);      // byte offset of the neighbour's row
        else rinfo[tid] = t2;
        const float4 dv = a.pd[er];
        pdx[tid] = dv.x;
        pdy[tid] = dv.y;
        pdz[tid] = dv.z;
    }
    lds_barrier();                                         // edge records are in
#pragma unroll
    for (int j = 0; j < CS; ++j) {
        // accumulator layout, like the epilogues: this lane's column for its rows (the neuron parameters are per-lane constants;
        // the position differences are LDS broadcasts)
        const float wx = a.wd[L.col[j] * 3], wy = a.wd[L.col[j] * 3 + 1], wz = a.wd[L.col[j] * 3 + 2], bd = a.bd[L.col[j]];
        const NeuronP nd = chain_lif(a.lifd, D, L.col[j]);
#pragma unroll
        for (int up = 0; up < NUP; ++up) {
            float v[8];
#pragma unroll
            for (int hq = 0; hq < 2; ++hq) {               // the two quads of the unit; element-wise mul, fma, fma, add as fn_pe1_kernel
                const int r0 = 16 * (2 * up + hq) + rowg;
                const float4 qx = *reinterpret_cast<const float4*>(pdx + r0), qy = *reinterpret_cast<const float4*>(pdy + r0),
                             qz = *reinterpret_cast<const float4*>(pdz + r0);
                const f32x2 wx2 = f32x2{wx, wx}, wy2 = f32x2{wy, wy}, wz2 = f32x2{wz, wz}, bd2 = f32x2{bd, bd};
                const f32x2 ta = pk_fma(wz2, f32x2{qz.x, qz.y}, pk_fma(wy2, f32x2{qy.x, qy.y}, wx2 * f32x2{qx.x, qx.y})) + bd2;
                const f32x2 tb = pk_fma(wz2, f32x2{qz.z, qz.w}, pk_fma(wy2, f32x2{qy.z, qy.w}, wx2 * f32x2{qx.z, qx.w})) + bd2;
                v[4 * hq] = ta.x;
                v[4 * hq + 1] = ta.y;
                v[4 * hq + 2] = tb.x;
                v[4 * hq + 3] = tb.y;
            }
            lif_selfloop_n<8>(v, nd, a.T);
#pragma unroll
            for (int z = 0; z < 8; z += 2) chain_put2<CH_PLANE, CS>(X, L, j, 2 * up + (z >> 2), z & 3, v[z], v[z + 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const half8* const wp1 = reinterpret_cast<const half8*>(a.w1p);
    const half8* const wp2 = reinterpret_cast<const half8*>(a.w2p);
    const half8* const wp3 = reinterpret_cast<const half8*>(a.w3p);
    // k_j and v_j of one epilogue unit (column sub-block j, row sub-blocks 2 up, 2 up + 1), this lane's column
    float kq[2][8], vq[2][8];
    auto gather_kv = [&](int j, int up, float (&kd)[8], float (&vd)[8]) {
#pragma unroll
        for (int z = 0; z < 8; ++z) {
            const int row = 16 * (2 * up + (z >> 2)) + rowg + (z & 3);
            if (O32) {
                const unsigned off = reinterpret_cast<const unsigned*>(rinfo)[row] + 4u * (unsigned)L.col[j];
                kd[z] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.qkv + D) + off);
                vd[z] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.qkv + 2 * D) + off);
            } else {
                const int nrow = rinfo[row].y;
                kd[z] = a.qkv[(int64_t)nrow * a.ldq + D + L.col[j]];
                vd[z] = a.qkv[(int64_t)nrow * a.ldq + 2 * D + L.col[j]];
            }
        }
    };
    // q_i (fn:368) of this lane's column: of lane group g's own point, and of the points that live in the spare slots
    auto load_q = [&](int j, float& qa, float (&qu)[NUA > 0 ? NUA : 1]) {
        qa = __builtin_nontemporal_load(&a.qkv[(pt0 + (L.g < npts ? L.g : 0)) * a.ldq + L.col[j]]);
#pragma unroll
        for (int n = 0; n < NUA; ++n) qu[n] = __builtin_nontemporal_load(&a.qkv[(pt0 + (4 + n < npts ? 4 + n : 0)) * a.ldq + L.col[j]]);
    };
    float qa[2], qu[2][NUA > 0 ? NUA : 1];
    load_q(0, qa[0], qu[0]);
    gather_kv(0, 0, kq[0], vq[0]);                         // in flight during GEMM 1
    chain_w_prefetch<D>(wp1, cs0, lane, W);
    lds_barrier();                                         // pe1 panel complete

    // ---- GEMM 1: fc_delta2; epilogue pe = LIF(.), attn_in = q_i - k_j + pe -> panel, t = v_j + pe stays      fn:360-368
    {
        chain_gemm<D>(X, wp1, cs0, lane, W, acc);
        lds_barrier();                                     // every wave has read the pe1 panel: it may be overwritten
        // software pipeline over the units: the gathers of unit u + 1 are issued before the neuron arithmetic of unit u
        // and consumed after the one of unit u + 1 (the in-order vector-memory counter then waits for loads that are one
        // arithmetic block old)
#pragma unroll
        for (int j = 0; j < CS; ++j) {
            const float b1 = a.b1[L.col[j]];
            const NeuronP n1 = chain_lif(a.lif1, D, L.col[j]);
            if (j + 1 < CS) load_q(j + 1, qa[(j + 1) & 1], qu[(j + 1) & 1]);
#pragma unroll
            for (int up = 0; up < NUP; ++up) {
                const int u = j * NUP + up;
                if (up + 1 < NUP) gather_kv(j, up + 1, kq[(u + 1) & 1], vq[(u + 1) & 1]);
                else if (j + 1 < CS) gather_kv(j + 1, 0, kq[(u + 1) & 1], vq[(u + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                float v[8];
#pragma unroll
                for (int z = 0; z < 8; ++z) v[z] = __fmaf_rn(acc[2 * up + (z >> 2)][j][z & 3], 0.0625f, b1);      // undoes the x16 of the pre-scaled weights (exact)
                lif_selfloop_n<8>(v, n1, a.T);
                __builtin_amdgcn_sched_barrier(0);
                float ain[8];                              // attn_in = q_i - k_j + pe
#pragma unroll
                for (int z = 0; z < 8; ++z) {
                    const int rs = 2 * up + (z >> 2), e = z & 3;
                    // the slot's point: lane group g's own point, or (spare slots) a compile-time point per lane group
                    float qv;
                    if (4 * rs + e < KK) {
                        qv = qa[j & 1];
                    } else {
                        float c[4];
#pragma unroll
                        for (int sg = 0; sg < 4; ++sg) {
                            const int p = M::point(rs, sg, e);
                            c[sg] = p >= 4 ? qu[j & 1][p - 4 < NUA ? p - 4 : 0] : qa[j & 1];       // (unused slots: any value)
                        }
                        qv = group_select(c[0], c[1], c[2], c[3], g0, g1);
                    }
                    ain[z] = __fadd_rn(__fsub_rn(qv, kq[u & 1][z]), v[z]);
                    // t = v_j + pe (fn:386-389).  settle: formed HERE — left alone the compiler sank some of these adds to the softmax
                    // and kept both operands alive until then, the gathered one in scratch behind a vmcnt(0) right after its load
                    pe[rs][j][e] = settle(__fadd_rn(vq[u & 1][z], v[z]));
                }
#pragma unroll
                for (int z = 0; z < 8; z += 2) chain_put2<CH_PLANE, CS>(X, L, j, 2 * up + (z >> 2), z & 3, ain[z], ain[z + 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        chain_w_prefetch<D>(wp2, cs0, lane, W);
    }
    lds_barrier();                                         // attn_in panel complete
    // ---- GEMM 2: fc_gamma; epilogue g = LIF(.) -> panel                                                   fn:373-376
    {
        chain_gemm<D>(X, wp2, cs0, lane, W, acc);
        lds_barrier();
#pragma unroll
        for (int j = 0; j < CS; ++j) {
            const float b2 = a.b2[L.col[j]];
            const NeuronP n2 = chain_lif(a.lif2, D, L.col[j]);
#pragma unroll
            for (int up = 0; up < NUP; ++up) {
                float v[8];
#pragma unroll
                for (int z = 0; z < 8; ++z) v[z] = __fmaf_rn(acc[2 * up + (z >> 2)][j][z & 3], 0.0625f, b2);
                lif_selfloop_n<8>(v, n2, a.T);
#pragma unroll
                for (int z = 0; z < 8; z += 2) chain_put2<CH_PLANE, CS>(X, L, j, 2 * up + (z >> 2), z & 3, v[z], v[z + 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        chain_w_prefetch<D>(wp3, cs0, lane, W);
    }
    lds_barrier();                                         // g panel complete
    // ---- GEMM 3: fc_gamma2; per-point softmax over the kk neighbours, aggregation with v_j + pe           fn:378-389
    {
        chain_gemm<D>(X, wp3, cs0, lane, W, acc);
#pragma unroll
        for (int j = 0; j < CS; ++j) {
            const float b3 = a.b3[L.col[j]];
            // own rows: x = (a + b) / sqrt(hd) in place of the accumulators (pe already holds t = v_j + pe)
#pragma unroll
            for (int i = 0; i < RS; ++i) {                 // (packed: fma, then mul, per element as fn_softmax_agg_kernel)
                const f32x2 k16 = f32x2{0.0625f, 0.0625f}, b32 = f32x2{b3, b3}, is2 = f32x2{a.inv_sqrt_hd, a.inv_sqrt_hd};
                const f32x2 lo = pk_fma(f32x2{acc[i][j][0], acc[i][j][1]}, k16, b32) * is2;
                const f32x2 hi = pk_fma(f32x2{acc[i][j][2], acc[i][j][3]}, k16, b32) * is2;
                acc[i][j] = f32x4{lo.x, lo.y, hi.x, hi.y};
            }
            // lane group g's own point: its kk rows are this lane's registers, in neighbour order
            {
                float xs[KK], ts[KK];
#pragma unroll
                for (int jj = 0; jj < KK; ++jj) {
                    xs[jj] = acc[jj >> 2][j][jj & 3];
                    ts[jj] = pe[jj >> 2][j][jj & 3];
                }
                const float out = chain_softmax_agg<KK>(xs, ts);
                if (L.g < npts) {
                    const int64_t pt = pt0 + L.g;
                    if (a.res_split) chain_store_split_nt(a.res, pt, D, L.col[j], out);
                    else __builtin_nontemporal_store(out, &a.res[pt * D + L.col[j]]);
                }
            }
            // the points in the spare slots: point 4 + n is summed by lane group n, which fetches the point's values from the lane
            // groups that hold them (two swaps per value and source; a value no lane group asks for costs nothing)
            if (NUA > 0) {
                float xs[KK], ts[KK];
#pragma unroll
                for (int jj = 0; jj < KK; ++jj) {
                    float cx[4] = {0.f, 0.f, 0.f, 0.f}, ct[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int n = 0; n < NUA; ++n) {
                        const int rs = M::ua_rs(n, jj), e = M::ua_e(n, jj), sg = M::ua_g(n, jj);
                        cx[n] = group_bcast(acc[rs][j][e], sg);
                        ct[n] = group_bcast(pe[rs][j][e], sg);
                    }
                    xs[jj] = NUA == 1 ? cx[0] : group_select(cx[0], cx[1], cx[2], cx[3], g0, g1);
                    ts[jj] = NUA == 1 ? ct[0] : group_select(ct[0], ct[1], ct[2], ct[3], g0, g1);
                }
                const float out = chain_softmax_agg<KK>(xs, ts);
                if (L.g < NUA && 4 + L.g < npts) {
                    const int64_t pt = pt0 + 4 + L.g;
                    if (a.res_split) chain_store_split_nt(a.res, pt, D, L.col[j], out);
                    else __builtin_nontemporal_store(out, &a.res[pt * D + L.col[j]]);
                }
            }
        }
    }
}

template <int D, int KK, bool O32>
static int launch_chain_t(const ChainArgs& a, hipStream_t st) {
    constexpr int lds = ChainShape<D>::LDS;
    static DeviceOnce lds_once;                         // one per kernel instantiation, one bit per device
    SAPCU_SET_MAX_LDS(lds_once, (&fn_edge_chain_kernel<D, KK, O32>), lds);
    constexpr int PPG = ChainSlots<D, KK>::PPG;
    const int64_t ngroups = (a.P + PPG - 1) / PPG;
    const int64_t grid = ngroups < 8 ? ngroups : ((ngroups + 7) / 8) * 8;      // 8 XCD ranges of equal slot count
    SAPCU_CHECK_ARG(grid < 0x7fffffffLL, "edge_chain: too many groups");
    hipLaunchKernelGGL((fn_edge_chain_kernel<D, KK, O32>), dim3((unsigned)grid), dim3(ChainShape<D>::NW * 64), lds, st, a);
    SAPCU_CHECK_LAUNCH();
    return SAPCU_OK;
}

bool fn_edge_chain_ok(int d, int kk) { return (d == 128 && kk == 24) || (d == 256 && kk == 18) || (d == 512 && kk == 12); }

int launch_fn_edge_chain(ChainArgs a, const float* patch, const int32_t* idx, int d, int kk, int2* tab, float4* pd,
                         hipStream_t st) {
    if (a.P == 0) return SAPCU_OK;
    SAPCU_CHECK_ARG(fn_edge_chain_ok(d, kk), "edge_chain: unsupported block shape d=%d kk=%d", d, kk);
    SAPCU_CHECK_ARG(a.P * (int64_t)kk < 0x7fffffffLL && a.P * (int64_t)(a.ldq) < (1LL << 40), "edge_chain: too many rows");
    const int64_t rows = a.P * kk;
    hipLaunchKernelGGL(edge_prep_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, patch, idx, rows, a.m, kk, tab, pd);
    SAPCU_CHECK_LAUNCH();
    a.tab = tab;
    a.pd = pd;
    // (rows x row pitch + the three column ranges) in bytes below 4 GiB: 32-bit gather offsets
    const bool o32 = !a.wide_offsets && (uint64_t)a.P * (uint64_t)a.ldq * 4u + 3u * (uint64_t)d * 4u < (1ull << 32) && a.ldq > 0;
    if (d == 128) return o32 ? launch_chain_t<128, 24, true>(a, st) : launch_chain_t<128, 24, false>(a, st);
    if (d == 256) return o32 ? launch_chain_t<256, 18, true>(a, st) : launch_chain_t<256, 18, false>(a, st);
    return o32 ? launch_chain_t<512, 12, true>(a, st) : launch_chain_t<512, 12, false>(a, st);
}

int launch_pack_chain_weights(const void* w16_hi, const void* w16_lo, int d, void* out, hipStream_t st) {
    const int64_t total = (int64_t)(d / 16) * (d / 32) * 2 * 64;
    hipLaunchKernelGGL(pack_chain_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       (const _Float16*)w16_hi, (const _Float16*)w16_lo, d, (_Float16*)out);
    SAPCU_CHECK_LAUNCH();
    return SAPCU_OK;
}

}  // namespace sapcu
