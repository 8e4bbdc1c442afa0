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
// Shape of the work.  d/32 waves per workgroup; wave w owns output columns 32w .. 32w+31 of all three GEMMs and all ROWS rows:
// wave tile ROWS x 32 = 3 / 4 MFMA blocks of 32x32 (d = 512: d/64 waves, 64 x 64 = 4 blocks).  A wave's weight fragments are not shared with any other wave,
// so they bypass LDS: pre-packed at model build in fragment order (one contiguous KiB per (column block, k16, plane)),
// streamed L2 -> registers two k16 steps ahead.  No barrier inside a GEMM; six workgroup barriers per group.
// The epilogues run in the accumulator layout (lane = column: bias and neuron parameters are per-lane constants; register
// e of block i = row 32i + 8(e>>2) + 4h + (e&3)), pe stays in registers until the aggregation, the per-point softmax gets
// its rows from the two lane halves with v_permlane32_swap and sums them in neighbour order — every value equals the
// unfused chain's bit for bit (same split-f16 products in the same order, same neuron arithmetic, same softmax order).
// d = 128: 256-thread workgroups, 51 KiB of LDS, two per CU (192 registers, see the kernel).  d = 256: 512 threads, 131 KiB, one per CU.  d = 512: 512
// threads (wave tile 64 x 64), 131 KiB.  The k loops are rolled (two k16 steps per iteration).  What bounds them (each pipe at its practical rate, the
// kernel time their sum) and the overlap designs that were measured without gain: DESIGN.md section 4.1c.
#include "common.h"
#include "gemm_epi.h"
#include "ops.h"

namespace sapcu {
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// Shape of a workgroup's work.  ROWS = MFMA rows per group = whole points: d = 128: 96 rows = 4 points of 24 neighbours (no idle
// row; 51 KiB of LDS; two 4-wave workgroups per CU at 192 registers per wave), d = 256: 128 rows = 7 points of 18 (one 8-wave
// workgroup, 256 registers), d = 512: 64 rows = 5 points of 12 (a 128-row panel of 512 columns would be 256 KiB; one 8-wave
// workgroup, 256 registers).  One plane of one k32 step is [ROWS][32 halves]; a k32 step = hi plane | lo plane.  A wave owns CB
// column blocks of 32 and all ROWS rows: RB x CB accumulator blocks.  CB = 2 at d = 512 (wave tile 64 x 64): half the LDS
// fragment reads of a 16-wave form and room for t = v + pe in registers (the 16-wave form, 128 registers per wave, parked it in
// scratch: 10 GB of HBM traffic per step).  Measured, ms per launch at 4096 x 48 points — d = 512: 16 waves 17.3; 16 waves with
// the weight ring refilled in place and unconditionally (exact wait counts in the rolled k loop) 17.2; 8 waves, same loop 16.6;
// 8 waves, epilogue units of 8 elements 16.3 (before the wait counts were exact the 8-wave form lost: 18.4 against 17.6).
// d = 256: fully unrolled k loop 8.27, rolled 8.10.  d = 128: 128-row groups (5 points + 8 idle rows), two workgroups per CU,
// unrolled loop 4.64; 96-row groups, three per CU (168 registers, 23 of them spilled), rolled loop 4.42; two per CU, no spill: 4.55;
// round 3, once the t = v_j + pe adds stopped being sunk to the end of the kernel (see the first epilogue): 144 / 182 / 198
// registers (d = 128 / 256 / 512), no scratch anywhere, three per CU again at d = 128.
#ifndef SAPCU_CHAIN_WD
#define SAPCU_CHAIN_WD 2
#endif
#ifndef SAPCU_CHAIN_LB128
#define SAPCU_CHAIN_LB128 3
#endif
template <int D> struct ChainShape {
    static constexpr int ROWS = D == 128 ? 96 : (D == 256 ? 128 : 64);
    static constexpr int RB = ROWS / 32;              // 32-row MFMA blocks per wave tile
    static constexpr int CB = D == 512 ? 2 : 1;       // 32-column blocks per wave
    static constexpr int NB = RB * CB;                // accumulator blocks per wave (block b: rows i = b / CB, columns j = b % CB)
    static constexpr int NW = D / (32 * CB);          // waves per workgroup
    static constexpr int PLANE = ROWS * 64;
    static constexpr int KSTEP = 2 * PLANE;
    static constexpr int LDS = ROWS * D * 4 + ROWS * 16 + ROWS * 8;
    static constexpr int US = 8;                      // elements per lane of one epilogue unit (d = 128 ran units of 4 while its
                                                      // registers were short: 150 of 170 now, units of 8 are 0.1 ms per step faster)
    static constexpr int WD = SAPCU_CHAIN_WD;                     // weight fragments this many k16 steps ahead (x CB column blocks x hi, lo) = the
                                                      // body of the rolled k loop
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

// weights in fragment order: out[((cb * nk16 + s) * 2 + plane) * 64 + lane][j] = w16_plane[32 cb + (lane & 31)][16 s + 8 (lane >> 5) + j]
__global__ __launch_bounds__(256) void pack_chain_weights_kernel(const _Float16* __restrict__ hi, const _Float16* __restrict__ lo,
                                                                 int d, _Float16* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // one 8-half fragment piece per thread
    const int nk16 = d / 16;
    const int64_t total = (int64_t)(d / 32) * nk16 * 2 * 64;
    if (t >= total) return;
    const int lane = (int)(t & 63);
    const int plane = (int)((t >> 6) & 1);
    const int64_t cs = t >> 7;
    const int s = (int)(cs % nk16), cb = (int)(cs / nk16);
    const _Float16* src = (plane ? lo : hi) + (int64_t)(32 * cb + (lane & 31)) * d + 16 * s + 8 * (lane >> 5);
    *reinterpret_cast<half8*>(out + t * 8) = *reinterpret_cast<const half8*>(src);
}

// value of the lane half `hr` (0: lanes 0-31, 1: lanes 32-63) of x, in BOTH halves (lane c and lane 32+c get lane (32 hr + c)'s)
__device__ __forceinline__ float half_bcast(float x, int hr) {
    const unsigned a = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(a, a, false, false);
    return __uint_as_float(hr ? r[1] : r[0]);
}

// lanes 0-31 get the lane half `ha` of a, lanes 32-63 the lane half `hb` of b (ha, hb compile-time; h = this lane's half)
__device__ __forceinline__ float half_pick(float a, int ha, float b, int hb, int h) {
    const unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
    if (ha == 0 && hb == 1) return h ? b : a;
    if (ha == hb) {
        const auto r = __builtin_amdgcn_permlane32_swap(ua, ub, false, false);       // r[0] = [a.lo, b.lo], r[1] = [a.hi, b.hi]
        return __uint_as_float(ha ? r[1] : r[0]);
    }
    const unsigned s = h ? ua : ub;                                                  // [b.lo, a.hi]
    const auto r = __builtin_amdgcn_permlane32_swap(s, s, false, false);             // [b.lo, b.lo], [a.hi, a.hi]
    return __uint_as_float(h ? r[0] : r[1]);
}

template <int CB>
struct ChainLane {       // per-lane constants of the epilogues
    int r32, h;
    int col[CB];         // this lane's column of column block j
    unsigned xw[CB][2];  // LDS byte offset of (row 4h + .., column col[j] as k) for rows with ((row>>3)&1) = 0 / 1
};

template <int WD, int CB>
struct ChainW {           // weight fragments of WD k16 steps x CB column blocks (hi, lo)
    half8 wh[WD][CB], wl[WD][CB];
};

// fragment (column block cb, k16 step s, plane) of a packed matrix: [(cb * NK16 + s) * 2 + plane][64 lanes]
template <int D>
__device__ __forceinline__ half8 chain_w_frag(const half8* __restrict__ wp, int cb, int s, int plane, int lane) {
    return wp[(((int64_t)cb * (D / 16) + s) * 2 + plane) * 64 + lane];
}

// first WD k16 steps of a GEMM's weight stream: issued well before the GEMM so that their L2 latency is covered
template <int D>
__device__ __forceinline__ void chain_w_prefetch(const half8* __restrict__ wp, int cb0, int lane,
                                                 ChainW<ChainShape<D>::WD, ChainShape<D>::CB>& W) {
#pragma unroll
    for (int s = 0; s < ChainShape<D>::WD; ++s)
#pragma unroll
        for (int j = 0; j < ChainShape<D>::CB; ++j) {
            W.wh[s][j] = chain_w_frag<D>(wp, cb0 + j, s, 0, lane);
            W.wl[s][j] = chain_w_frag<D>(wp, cb0 + j, s, 1, lane);
        }
}

template <int D>
__device__ __forceinline__ void chain_gemm(const unsigned char* X, const half8* __restrict__ wp, int cb0, int lane,
                                           ChainW<ChainShape<D>::WD, ChainShape<D>::CB>& W, f32x16 (&acc)[ChainShape<D>::NB]) {
    constexpr int NK16 = D / 16, WD = ChainShape<D>::WD;
    constexpr int RB = ChainShape<D>::RB, CB = ChainShape<D>::CB, NB = ChainShape<D>::NB;
    constexpr int CH_PLANE = ChainShape<D>::PLANE, CH_KSTEP = ChainShape<D>::KSTEP;
    const int r32 = lane & 31, h = lane >> 5;
    const int sw = (r32 >> 2) & 3;                         // (row >> 2) & 3 of rows 32 i + r32
    const unsigned char* xa = X + r32 * 64;
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[b][e] = 0.f;
    // One k16 step S; SJ = its slot of the weight ring (S % WD, a compile-time value: the loop is rolled with a body of WD steps,
    // which keeps the ring's indices static without letting the scheduler hoist all the steps' operands).  A ring slot is refilled
    // BEHIND the MFMAs that read it, in place (no copy of the fragments).
#define SAPCU_CHAIN_STEP_INPLACE(S, SJ)                                                                                     \
    {                                                                                                                       \
        const int s_ = (S);                                                                                                 \
        const unsigned ko = (unsigned)((s_ >> 1) * CH_KSTEP + ((((s_ & 1) * 2 + h) ^ sw) * 16));                            \
        half8 ah[RB], al[RB];                                                                                               \
        _Pragma("unroll") for (int i = 0; i < RB; ++i) {                                                                    \
            ah[i] = *reinterpret_cast<const half8*>(xa + ko + i * 2048);                                                    \
            al[i] = *reinterpret_cast<const half8*>(xa + ko + i * 2048 + CH_PLANE);                                         \
        }                                                                                                                   \
        _Pragma("unroll") for (int b = 0; b < NB; ++b)                                                                      \
            acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[b / CB], W.wh[SJ][b % CB], acc[b], 0, 0, 0);                 \
        _Pragma("unroll") for (int b = 0; b < NB; ++b)                                                                      \
            acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[b / CB], W.wl[SJ][b % CB], acc[b], 0, 0, 0);                 \
        _Pragma("unroll") for (int b = 0; b < NB; ++b)                                                                      \
            acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[b / CB], W.wh[SJ][b % CB], acc[b], 0, 0, 0);                 \
        __builtin_amdgcn_sched_barrier(0);                                                                                  \
        /* unconditional (the last WD steps re-load the last fragment): with a branch around the loads the compiler's wait  \
           counts at the loop head degrade to vmcnt(0), i.e. every iteration waits for the loads it has just issued */      \
        const int sn_ = s_ + WD < NK16 ? s_ + WD : NK16 - 1;                                                                \
        _Pragma("unroll") for (int j = 0; j < CB; ++j) {                                                                    \
            W.wh[SJ][j] = chain_w_frag<D>(wp, cb0 + j, sn_, 0, lane);                                                       \
            W.wl[SJ][j] = chain_w_frag<D>(wp, cb0 + j, sn_, 1, lane);                                                       \
        }                                                                                                                   \
    }
#pragma unroll 1
    for (int s0 = 0; s0 < NK16; s0 += WD) {
#pragma unroll
        for (int sj = 0; sj < WD; ++sj) SAPCU_CHAIN_STEP_INPLACE(s0 + sj, sj)
    }
#undef SAPCU_CHAIN_STEP_INPLACE
}

// write v (row = 32 i + 8 q + 4 h + u, k = this lane's column) into the panel as the split-f16 operand of the next GEMM
template <int PLANE, int CB>
__device__ __forceinline__ void chain_put(unsigned char* X, const ChainLane<CB>& L, int j, int i, int q, int u, float v) {
    unsigned char* p = X + L.xw[j][q & 1] + (unsigned)(i * 2048 + q * 512 + u * 64);
    const _Float16 hi = (_Float16)v;
    *reinterpret_cast<_Float16*>(p) = hi;
    *reinterpret_cast<_Float16*>(p + PLANE) = (_Float16)(v - (float)hi);
}

// two elements of one column, rows u and u + 1 of a register quad: the split with the packed conversions of gfx950 (v_cvt_pk_f16_f32:
// one instruction per PAIR and plane instead of one per element; round-to-nearest-even like the scalar form — the same bits)
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
template <int PLANE, int CB>
__device__ __forceinline__ void chain_put2(unsigned char* X, const ChainLane<CB>& L, int j, int i, int q, int u, float v0, float v1) {
    unsigned char* p = X + L.xw[j][q & 1] + (unsigned)(i * 2048 + q * 512 + u * 64);
    const f32x2 v = f32x2{v0, v1};
    const half2v hi = __builtin_convertvector(v, half2v);
    // lo = f16(v - hi) as ONE instruction per element: v_fma_mixlo/mixhi_f16 read the f16 half in place, form -hi + v exactly
    // (v - hi is exact in f32 anyway) and round once to f16 — the bits of cvt(f32(v) - f32(hi)); the compiler does not select
    // them here on its own (it unpacked the pair instead)
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

// threads = 64 x (d / 32 / CB).  d = 128: three 256-thread workgroups per CU (51 KiB of LDS each, <= 170 registers per wave: the
// kernel takes 144 and no scratch; three ran 1 % faster than two in the same-box A/B of round 3, 2.9 % in round 2's).
// O32: the q|k|v tensor is smaller than 4 GiB — the gathers address it as a scalar base + a 32-bit byte offset per lane (one add
// per gathered element; the 64-bit form spends a 64-bit multiply-add and two adds on each).
template <int D, int KK, bool O32>
__global__ __launch_bounds__(ChainShape<D>::NW * 64, (D == 128 ? SAPCU_CHAIN_LB128 : 1)) void fn_edge_chain_kernel(const ChainArgs a) {
    using S = ChainShape<D>;
    constexpr int CH_ROWS = S::ROWS, RB = S::RB, CB = S::CB, NB = S::NB, CH_PLANE = S::PLANE, CH_KSTEP = S::KSTEP;
    constexpr int PPG = CH_ROWS / KK;                      // points per group
    constexpr int US = S::US, UPB = 16 / US;               // elements per epilogue unit, units per accumulator block
    constexpr int NU = NB * UPB;                           // units per wave; unit u: block u / UPB, elements US (u % UPB) .. + US - 1
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char* X = smem;
    float4* pdl = reinterpret_cast<float4*>(smem + CH_ROWS * D * 4);
    int2* rinfo = reinterpret_cast<int2*>(smem + CH_ROWS * D * 4 + CH_ROWS * 16);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cb0 = w * CB;                                // first column block of this wave
    ChainLane<CB> L;
    L.r32 = lane & 31;
    L.h = lane >> 5;
    // element (row, k = col): k32 step = column block, chunk = r32 >> 3, half = r32 & 7; (row >> 2) & 3 = (2 q + h) & 3
#pragma unroll
    for (int j = 0; j < CB; ++j) {
        L.col[j] = 32 * (cb0 + j) + L.r32;
#pragma unroll
        for (int qo = 0; qo < 2; ++qo)
            L.xw[j][qo] = (unsigned)((cb0 + j) * CH_KSTEP + L.h * 256 + (((L.r32 >> 3) ^ ((2 * qo + L.h) & 3)) * 16) + (L.r32 & 7) * 2);
    }
    // row of element e of row block i (this lane half)
    auto row_of = [&](int i, int e) { return 32 * i + 8 * (e >> 2) + 4 * L.h + (e & 3); };

    // group of this workgroup: contiguous ranges of groups per XCD (blockIdx & 7), so that the tiles of one patch — which
    // gather the same q / k / v rows — share an L2
    const int64_t ngroups = (a.P + PPG - 1) / PPG;
    int64_t g;
    {
        const int64_t nx = gridDim.x < 8 ? 1 : 8;
        const int64_t x = nx == 1 ? 0 : (blockIdx.x & 7), slot = nx == 1 ? blockIdx.x : (blockIdx.x >> 3);
        const int64_t qd = ngroups / nx, rem = ngroups % nx;
        g = x * qd + (x < rem ? x : rem) + slot;
        if (slot >= qd + (x < rem ? 1 : 0)) return;
    }
    const int64_t pt0 = g * PPG;
    const int npts = (int)((a.P - pt0) < PPG ? (a.P - pt0) : PPG);

    f32x16 acc[NB], pe[NB];
    ChainW<S::WD, CB> W;
    // ---- phase 0: edge records of the group's rows; pe1 = LIF(fc_delta(x_i - x_j)) -> panel              fn:310,355-358
    float qp[CB][PPG];                                     // q_i of the group's points, this lane's columns
    if (tid < CH_ROWS) {
        const int pl = tid / KK;
        const bool ok = pl < npts;                                           // pad rows replay the group's first edge row
        const int64_t er = ok ? pt0 * KK + tid : pt0 * KK;
        const int2 t = a.tab[er];
        if (O32) reinterpret_cast<unsigned*>(rinfo)[tid] = (unsigned)t.y * (unsigned)(a.ldq * 4);      // byte offset of the neighbour's row
        else rinfo[tid] = t;
        pdl[tid] = a.pd[er];
    }
    {
        // accumulator layout, like the epilogues: this lane's column(s) for its rows (the neuron parameters are per-lane
        // constants; the position differences are LDS broadcasts)
        float wx[CB], wy[CB], wz[CB], bd[CB];
        NeuronP nd[CB];
#pragma unroll
        for (int j = 0; j < CB; ++j) {
            wx[j] = a.wd[L.col[j] * 3];
            wy[j] = a.wd[L.col[j] * 3 + 1];
            wz[j] = a.wd[L.col[j] * 3 + 2];
            bd[j] = a.bd[L.col[j]];
            nd[j] = chain_lif(a.lifd, D, L.col[j]);
        }
        lds_barrier();                                     // edge records are in
        // q_i of the group's points for the first epilogue (fn:368): issued here, consumed after GEMM 1
#pragma unroll
        for (int j = 0; j < CB; ++j)
#pragma unroll
            for (int p = 0; p < PPG; ++p) qp[j][p] = __builtin_nontemporal_load(&a.qkv[(pt0 + (p < npts ? p : 0)) * a.ldq + L.col[j]]);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int b = u / UPB, i = b / CB, j = b % CB, e0 = US * (u % UPB);
            float v[US];
#pragma unroll
            for (int z = 0; z < US; ++z) {
                const float4 dd = pdl[row_of(i, e0 + z)];
                float t0 = __fmul_rn(wx[j], dd.x);
                t0 = __fmaf_rn(wy[j], dd.y, t0);
                t0 = __fmaf_rn(wz[j], dd.z, t0);
                v[z] = __fadd_rn(t0, bd[j]);
            }
            lif_selfloop_n<US>(v, nd[j], a.T);
#pragma unroll
            for (int z = 0; z < US; z += 2) chain_put2<CH_PLANE, CB>(X, L, j, i, (e0 + z) >> 2, (e0 + z) & 3, v[z], v[z + 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const half8* const wp1 = reinterpret_cast<const half8*>(a.w1p);
    const half8* const wp2 = reinterpret_cast<const half8*>(a.w2p);
    const half8* const wp3 = reinterpret_cast<const half8*>(a.w3p);
    // k_j and v_j of one epilogue unit u, this lane's column of the unit's column block
    float kq[2][US], vq[2][US];
    auto gather_kv = [&](int u, float (&kd)[US], float (&vd)[US]) {
        const int b = u / UPB, i = b / CB, j = b % CB;
#pragma unroll
        for (int z = 0; z < US; ++z) {
            if (O32) {
                const unsigned off = reinterpret_cast<const unsigned*>(rinfo)[row_of(i, US * (u % UPB) + z)] + 4u * (unsigned)L.col[j];
                kd[z] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.qkv + D) + off);
                vd[z] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.qkv + 2 * D) + off);
            } else {
                const int nrow = rinfo[row_of(i, US * (u % UPB) + z)].y;
                kd[z] = a.qkv[(int64_t)nrow * a.ldq + D + L.col[j]];
                vd[z] = a.qkv[(int64_t)nrow * a.ldq + 2 * D + L.col[j]];
            }
        }
    };
    gather_kv(0, kq[0], vq[0]);                            // in flight during GEMM 1
    chain_w_prefetch<D>(wp1, cb0, lane, W);
    lds_barrier();                                         // pe1 panel complete

    // ---- GEMM 1: fc_delta2; epilogue pe = LIF(.), attn_in = q_i - k_j + pe -> panel, t = v_j + pe stays      fn:360-368
    {
        float b1[CB];
        NeuronP n1[CB];
#pragma unroll
        for (int j = 0; j < CB; ++j) {
            b1[j] = a.b1[L.col[j]];
            n1[j] = chain_lif(a.lif1, D, L.col[j]);
        }
        chain_gemm<D>(X, wp1, cb0, lane, W, acc);
        lds_barrier();                                     // every wave has read the pe1 panel: it may be overwritten
        // software pipeline over the units: the gathers of unit u + 1 are issued before the neuron arithmetic of unit u
        // and consumed after the one of unit u + 1 (the in-order vector-memory counter then waits for loads that are one
        // arithmetic block old)
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int b = u / UPB, i = b / CB, j = b % CB, e0 = US * (u % UPB);
            if (u + 1 < NU) gather_kv(u + 1, kq[(u + 1) & 1], vq[(u + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            float v[US];
#pragma unroll
            for (int z = 0; z < US; ++z) v[z] = __fmaf_rn(acc[b][e0 + z], 0.0625f, b1[j]);      // undoes the x16 of the pre-scaled weights (exact)
            lif_selfloop_n<US>(v, n1[j], a.T);
            __builtin_amdgcn_sched_barrier(0);
            float ain[US];                                 // attn_in = q_i - k_j + pe
#pragma unroll
            for (int z = 0; z < US; ++z) {
                // the row's point: row = 32 i + 8 q + 4 h + u, compile-time per lane half (pad rows: any point)
                const int e = e0 + z;
                const int r0 = 32 * i + 8 * (e >> 2) + (e & 3), r1 = r0 + 4;
                const int p0 = r0 / KK < PPG ? r0 / KK : PPG - 1, p1 = r1 / KK < PPG ? r1 / KK : PPG - 1;
                const float qv = p0 == p1 ? qp[j][p0] : (L.h ? qp[j][p1] : qp[j][p0]);
                ain[z] = __fadd_rn(__fsub_rn(qv, kq[u & 1][z]), v[z]);
                // t = v_j + pe (fn:386-389).  settle: formed HERE — left alone the compiler sank some of these adds to the softmax
                // and kept both operands alive until then, the gathered one in scratch behind a vmcnt(0) right after its load
                pe[b][e] = settle(__fadd_rn(vq[u & 1][z], v[z]));
            }
#pragma unroll
            for (int z = 0; z < US; z += 2) chain_put2<CH_PLANE, CB>(X, L, j, i, (e0 + z) >> 2, (e0 + z) & 3, ain[z], ain[z + 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
        chain_w_prefetch<D>(wp2, cb0, lane, W);
    }
    lds_barrier();                                         // attn_in panel complete
    // ---- GEMM 2: fc_gamma; epilogue g = LIF(.) -> panel                                                   fn:373-376
    {
        float b2[CB];
        NeuronP n2[CB];
#pragma unroll
        for (int j = 0; j < CB; ++j) {
            b2[j] = a.b2[L.col[j]];
            n2[j] = chain_lif(a.lif2, D, L.col[j]);
        }
        chain_gemm<D>(X, wp2, cb0, lane, W, acc);
        lds_barrier();
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int b = u / UPB, i = b / CB, j = b % CB, e0 = US * (u % UPB);
            float v[US];
#pragma unroll
            for (int z = 0; z < US; ++z) v[z] = __fmaf_rn(acc[b][e0 + z], 0.0625f, b2[j]);
            lif_selfloop_n<US>(v, n2[j], a.T);
#pragma unroll
            for (int z = 0; z < US; z += 2) chain_put2<CH_PLANE, CB>(X, L, j, i, (e0 + z) >> 2, (e0 + z) & 3, v[z], v[z + 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
        chain_w_prefetch<D>(wp3, cb0, lane, W);
    }
    lds_barrier();                                         // g panel complete
    // ---- GEMM 3: fc_gamma2; per-point softmax over the kk neighbours, aggregation with v_j + pe           fn:378-389
    {
        chain_gemm<D>(X, wp3, cb0, lane, W, acc);
#pragma unroll
        for (int j = 0; j < CB; ++j) {
            const float b3 = a.b3[L.col[j]];
            // own rows: x = (a + b) / sqrt(hd) in place of the accumulators (pe already holds t = v_j + pe)
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    acc[i * CB + j][e] = __fmul_rn(__fmaf_rn(acc[i * CB + j][e], 0.0625f, b3), a.inv_sqrt_hd);
            // per point: its kk rows alternate between the lane halves in quads.  Points are taken in PAIRS: lane half 0 runs the
            // neighbour-ordered sums (fn_softmax_agg_kernel's operation order) of point pp, half 1 those of point pp + 1, each
            // half fetching its point's values from whichever half holds them (an odd last point: both halves, as before r3 —
            // every point used to be summed by both halves).
#pragma unroll
            for (int pp = 0; pp < PPG; pp += 2) {
                const bool pair = pp + 1 < PPG;            // compile-time after unrolling
                float xs[KK], ts[KK];
#pragma unroll
                for (int jj = 0; jj < KK; ++jj) {
                    const int r = pp * KK + jj, i = r >> 5, rr = r & 31;
                    const int e = ((rr >> 3) << 2) | (rr & 3), hr = (rr >> 2) & 1;
                    if (pair) {
                        const int r1 = r + KK, i1 = r1 >> 5, rr1 = r1 & 31;
                        const int e1 = ((rr1 >> 3) << 2) | (rr1 & 3), hr1 = (rr1 >> 2) & 1;
                        xs[jj] = half_pick(acc[i * CB + j][e], hr, acc[i1 * CB + j][e1], hr1, L.h);
                        ts[jj] = half_pick(pe[i * CB + j][e], hr, pe[i1 * CB + j][e1], hr1, L.h);
                    } else {
                        xs[jj] = half_bcast(acc[i * CB + j][e], hr);
                        ts[jj] = half_bcast(pe[i * CB + j][e], hr);
                    }
                }
                float mx = -__builtin_huge_valf();
#pragma unroll
                for (int jj = 0; jj < KK; ++jj) mx = fmaxf(mx, xs[jj]);
                float den = 0.f;
#pragma unroll
                for (int jj = 0; jj < KK; ++jj) {
                    xs[jj] = fast_exp(__fsub_rn(xs[jj], mx));
                    den = __fadd_rn(den, xs[jj]);
                }
                const float inv_den = __fdiv_rn(1.0f, den);
                float out = 0.f;
#pragma unroll
                for (int jj = 0; jj < KK; ++jj) out = __fmaf_rn(__fmul_rn(xs[jj], inv_den), ts[jj], out);
                const int p = pair ? pp + L.h : pp;        // this half's point
                if (p < npts && (pair || L.h == 0)) {
                    const int64_t pt = pt0 + p;
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
    constexpr int PPG = ChainShape<D>::ROWS / KK;
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
    const int64_t total = (int64_t)(d / 32) * (d / 16) * 2 * 64;
    hipLaunchKernelGGL(pack_chain_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       (const _Float16*)w16_hi, (const _Float16*)w16_lo, d, (_Float16*)out);
    SAPCU_CHECK_LAUNCH();
    return SAPCU_OK;
}

}  // namespace sapcu
