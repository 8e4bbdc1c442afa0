// Split-f16 MFMA GEMM on split rows, BIG-TILE version (gfx950): 256 x BN output tile per 512-thread workgroup.
//
//   C[r, n] = epi( A[r, k] . W[n, k]^T + bias[n] ),   A in "split rows" (gemm_epi.h), W pre-split and x16
//
// Same arithmetic, in the same order, as gemm_sf16_ring.hip (per k16: a_lo.w_hi, a_hi.w_lo, a_hi.w_hi into one f32
// accumulator) — the results are bit-identical to the ring kernel's.  What changes is the shape of the work:
//
//   ring kernel   128x128 tile, 16 waves (8 MFMA + 8 epilogue), 128 registers per wave: every operand byte that reaches LDS
//                 feeds 128 rows/columns of the other operand; per MFMA one 16-byte LDS fragment read per lane.
//   this kernel   256xBN tile (BN = 256 or 128), 8 waves of 256 registers, wave tile 128 x BN/4 (128 accumulator registers):
//                 half the operand bytes per flop through the DMA path and through the LDS read port, four times the matrix
//                 work between two barriers.  Since round 3 the MFMAs are v_mfma_f32_16x16x32_f16 (8 x 4 sub-blocks of 16 x 16
//                 per wave; 4.5 % faster at the clock the chip holds on that shape), accumulated pass-major per k32 step.  The epilogue
//                 runs in the MFMA waves themselves, in the accumulator layout (lane = column: bias and neuron parameters
//                 are per-lane constants; a register quad = 4 consecutive rows -> gemm_epi.h epilogue_group4; a store
//                 instruction writes two full 128-byte lines).
//
// Pipeline: k-step = 32.  Waves 0-3 stream the activation operand (3 slots of 32 KiB, 2 k-steps ahead: first touches, HBM
// latency), waves 4-7 the weight operand (2 slots, 1 step ahead, L2 hits); one stream per wave because a wave's vector-
// memory counter completes in order.  The rings run on across tile boundaries (global step counter).  Per step:
// fragment reads of the tile's rows 64-127 -> the MFMAs of rows 0-63 -> own DMAs of the next step landed -> barrier -> refill the
// slot just read -> the next step's fragments, read under the MFMAs of rows 64-127 (rows 0-63's operand first, each half of the
// weight fragments behind the MFMAs that read it last).  LDS = 96 + 64 (BN = 256) KiB = all 160 KiB.
// The epilogue's global stores also count in vmcnt: the waits after a tile boundary over-wait for them (safe: completion is
// in order), which costs a short bubble per 256x256 tile.
#include "common.h"
#include "gemm_epi.h"

namespace sapcu {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) unsigned char lds_byte;
typedef const __attribute__((address_space(1))) void* gptr_t;

constexpr int TBM = 256, TBK = 32;
constexpr int TA_PLANE = TBM * TBK * 2;          // 16 KiB: [256 rows][32 halves]
constexpr int TA_SLOT = 2 * TA_PLANE;            // hi | lo
constexpr int TA_SLOTS = 3;
constexpr int TW_SLOTS = 2;

template <int N>
__device__ __forceinline__ void bt_wait_vm() {
    if (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// A global load the compiler does not see as one (it would protect the destination registers with a vmcnt(0) at the top of
// the next tile, which drains the DMA look-ahead and the epilogue's stores): the caller waits with bt_wait_vm<0>().
__device__ __forceinline__ float bt_load_f32(const float* p) {
    float v;
    asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// Epilogue of one wave tile, FAST path: every row of the tile exists (interior tile) and the output formats are compile-time
// — no per-element row masks, no run-time format branches, no overflow counter (the generic epilogue_group4 path costs about
// 40 instructions and 3 scalar branches per stored element, 128 elements per lane and tile).  Same arithmetic, bit for bit.
//   C_SPLIT / C2_SPLIT: c / c2 (attn_in) leave as split rows instead of f32.


template <int EPI, int CS, bool C_SPLIT, bool C2_SPLIT>
__device__ __forceinline__ void bt_epilogue_fast(const GemmArgs& g, const f32x4 (&acc)[8][CS], int64_t row0, int col0, int c16, int g4,
                                                 const float (&pbias)[CS], const NeuronP (&pnp)[CS]) {
    constexpr bool ATTN = EPI == EPI_LIF_ATTN;
    // Attention epilogue, software-pipelined over the 8 row groups (one register quad = rows 16 rs + 4 g4 + 0..3): the (point,
    // neighbour) rows of group n+1 are loaded and the q / k gathers of group n issued BEFORE the neuron arithmetic of group n, so
    // every wait of the in-order counter is for loads issued one arithmetic block earlier (and the stores issued before them).
    int2 t_nxt[4];
    if (ATTN) {
#pragma unroll
        for (int u = 0; u < 4; ++u) t_nxt[u] = g.tab[row0 + 4 * g4 + u];
    }
#pragma unroll
    for (int rs = 0; rs < 8; ++rs) {
        const int64_t row = row0 + rs * 16 + 4 * g4;
        float qv[CS][4], kv[CS][4];
        if (ATTN) {
            int2 t4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) t4[u] = t_nxt[u];
#pragma unroll
            for (int j = 0; j < CS; ++j) {
                const int col = col0 + j * 16 + c16;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    qv[j][u] = g.q[(int64_t)t4[u].x * g.ldq + col];
                    kv[j][u] = g.kf[(int64_t)t4[u].y * g.ldq + col];
                }
            }
            if (rs + 1 < 8) {
                const int64_t nrow = row0 + (rs + 1) * 16 + 4 * g4;
#pragma unroll
                for (int u = 0; u < 4; ++u) t_nxt[u] = g.tab[nrow + u];
            }
            __builtin_amdgcn_sched_barrier(0);              // loads first, arithmetic after
        }
        float v[CS][4];
#pragma unroll
        for (int j = 0; j < CS; ++j) {
#pragma unroll
            for (int u = 0; u < 4; ++u) v[j][u] = __fadd_rn(__fmul_rn(acc[rs][j][u], 0.0625f), pbias[j]);
            if (EPI == EPI_LIF || ATTN) lif_selfloop_n<4>(v[j], pnp[j], g.lif_T);
        }
        if (ATTN) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < CS; ++j) {
            const int col = col0 + j * 16 + c16;
            float* cp = g.c + row * g.ldc + col;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (C_SPLIT) {
                    _Float16* rp = reinterpret_cast<_Float16*>(g.c + (row + u) * g.ldc);
                    const _Float16 hi = (_Float16)v[j][u];
                    rp[split_hi_index(g.ldc, col)] = hi;
                    rp[split_lo_index(g.ldc, col)] = (_Float16)(v[j][u] - (float)hi);
                } else {
                    cp[(int64_t)u * g.ldc] = v[j][u];
                }
            }
            if (ATTN) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float ai = __fadd_rn(__fsub_rn(qv[j][u], kv[j][u]), v[j][u]);
                    if (C2_SPLIT) {
                        _Float16* rp = reinterpret_cast<_Float16*>(g.c2 + (row + u) * g.ldc);
                        const _Float16 hi = (_Float16)ai;
                        rp[split_hi_index(g.ldc, col)] = hi;
                        rp[split_lo_index(g.ldc, col)] = (_Float16)(ai - (float)hi);
                    } else {
                        g.c2[(row + u) * g.ldc + col] = ai;
                    }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);                  // keep the groups apart (128 accumulators live)
    }
}

template <int EPI, int BN>
__global__ __launch_bounds__(512) void gemm_bt_kernel(const GemmArgs g) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem_raw[];
    constexpr int W_PLANE = BN * TBK * 2;
    constexpr int W_SLOT = 2 * W_PLANE;
    constexpr int A_BYTES = TA_SLOTS * TA_SLOT;
    constexpr int CT = BN / 128;                       // 32-column blocks per wave (wave tile = 128 x 32*CT)
    constexpr int W_PIECES = BN / 64;                  // 16-row DMA pieces per weight wave and plane
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    const int ntn = g.n / BN;
    const int64_t ntm = (g.r + TBM - 1) / TBM;
    const int64_t ntiles = ntm * ntn;
    const int nk = g.k / TBK;
    // tile order: the column tiles of one 256-row panel run at the same time on workgroups of ONE XCD (blockIdx & 7), so the
    // activation panel leaves HBM once (see gemm_sf16_ring.hip)
    int64_t first_logical, my_tiles, step_tm;
    int step_tn;
    {
        const int nx = gridDim.x < 8 ? 1 : 8;
        const int xcd = nx == 1 ? 0 : (int)(blockIdx.x & 7);
        const int wg_in_x = nx == 1 ? (int)blockIdx.x : (int)(blockIdx.x >> 3);
        const int wgs_per_x = nx == 1 ? (int)gridDim.x : (int)((gridDim.x - xcd + 7) >> 3);
        const int64_t qd = ntiles / nx, rem = ntiles % nx;
        const int64_t x_begin = xcd * qd + (xcd < rem ? xcd : rem);
        const int64_t x_count = qd + (xcd < rem ? 1 : 0);
        my_tiles = x_count > wg_in_x ? (x_count - wg_in_x + wgs_per_x - 1) / wgs_per_x : 0;
        first_logical = x_begin + wg_in_x;
        step_tm = wgs_per_x / ntn;
        step_tn = wgs_per_x - (int)step_tm * ntn;
    }
    if (my_tiles == 0) return;
    const int64_t first_tm = first_logical / ntn;
    const int first_tn = (int)(first_logical - first_tm * ntn);

    // ---- DMA role of this wave
    const bool is_a = wave < 4;
    const int sub = wave & 3;
    const int dsb = ((lane & 3) ^ ((lane >> 4) & 3)) * 16;         // source byte offset of this lane's (swizzled) LDS chunk
    const int64_t pitch_b = is_a ? 4 * (int64_t)g.lda : 2 * (int64_t)g.k;
    const bool a_il = is_a && (g.lda & 31) == 0;                    // interleaved split rows (gemm_epi.h): one line per row and k-step
    const int64_t lo_delta = is_a ? (a_il ? 64 : 2 * (int64_t)g.lda)
                                  : reinterpret_cast<const char*>(g.w16_lo) - reinterpret_cast<const char*>(g.w16_hi);
    const int kstep_bytes = a_il ? 128 : TBK * 2;
    const char* const op_base = is_a ? reinterpret_cast<const char*>(g.a) : reinterpret_cast<const char*>(g.w16_hi);
    const int pieces = is_a ? 4 : W_PIECES;                         // wave-uniform
    const int rows_per_wave = 16 * pieces;
    const int depth = is_a ? TA_SLOTS : TW_SLOTS;
    const int slot_bytes = is_a ? TA_SLOT : W_SLOT;
    const int plane_bytes = is_a ? TA_PLANE : W_PLANE;
    const int64_t total_steps = my_tiles * nk;
    int64_t pf_tile = 0, pf_tm = first_tm;
    int pf_kt = 0, pf_tn = first_tn;
    const char* pf_base = nullptr;
    unsigned pf_off[4] = {0, 0, 0, 0};
    const int lrow = rows_per_wave * sub + (lane >> 2);
    auto pf_setup = [&]() {
        const int64_t first = is_a ? pf_tm * TBM : (int64_t)pf_tn * BN;
        const int64_t left = (is_a ? g.r : (int64_t)g.n) - first;    // >= 1
        pf_base = op_base + first * pitch_b;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            int rr = lrow + 16 * p;
            if (rr >= left) rr = (int)(left - 1);                     // edge tile: clamped rows only feed masked outputs
            pf_off[p] = (unsigned)(rr * (int)pitch_b + dsb);
        }
    };
    lds_byte* const ring0 = (lds_byte*)(smem_raw + (is_a ? 0 : A_BYTES) + sub * rows_per_wave * 64);
    int issue_slot = 0;
    int64_t issued = 0;
    auto issue = [&]() {
        lds_byte* sb = ring0 + issue_slot * slot_bytes;
        const char* src = pf_base + pf_kt * kstep_bytes;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if (p < pieces) {
                __builtin_amdgcn_global_load_lds((gptr_t)(src + pf_off[p]), sb + p * 1024, 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gptr_t)(src + lo_delta + pf_off[p]), sb + plane_bytes + p * 1024, 16, 0, 0);
            }
        }
        ++issued;
        if (++issue_slot == depth) issue_slot = 0;
        if (++pf_kt == nk) {
            pf_kt = 0;
            ++pf_tile;
            pf_tm += step_tm;
            pf_tn += step_tn;
            if (pf_tn >= ntn) { pf_tn -= ntn; ++pf_tm; }
            if (pf_tile < my_tiles) pf_setup();
        }
    };
    pf_setup();
    while (issued < depth && issued < total_steps) issue();
    // this wave's DMAs of `step` have landed: 2*pieces DMAs per step, in order
    int64_t landed_upto = 0;            // every step below this index is known to have landed (a vmcnt(0) since its issue)
    auto wait_landed = [&](int64_t step) {
        if (step < landed_upto) return;
        const int64_t ahead = issued - step - 1;
        if (is_a) {                                                   // 8 per step
            if (ahead >= 2) bt_wait_vm<16>();
            else if (ahead == 1) bt_wait_vm<8>();
            else bt_wait_vm<0>();
        } else if (W_PIECES == 4) {
            if (ahead >= 1) bt_wait_vm<8>();
            else bt_wait_vm<0>();
        } else {
            if (ahead >= 1) bt_wait_vm<4>();
            else bt_wait_vm<0>();
        }
    };

    // ---- MFMA role (round 3: v_mfma_f32_16x16x32_f16 — the chip holds a higher clock on it than on 32x32x16, and one of it equals
    // two chained 32x32x16 bit for bit, fn_edge_chain.hip): the wave tile 128 x 32 CT is RS = 8 row sub-blocks x CS = 2 CT column
    // sub-blocks of 16; lane l supplies row / weight row (l & 15), k chunk (l >> 4) of a k32 step and holds rows 4 (l >> 4) + e,
    // column (l & 15) of each sub-block.
    constexpr int CS = 2 * CT;
    const int c16 = lane & 15, g4 = lane >> 4;
    const unsigned frag_ko = (unsigned)((g4 ^ ((c16 >> 2) & 3)) * 16);           // (the rows' swizzle term (row >> 2) & 3 = (c16 >> 2) & 3)
    const unsigned a_frag0 = (unsigned)((wm * 128 + c16) * (TBK * 2)) + frag_ko;
    const unsigned w_frag0 = (unsigned)(A_BYTES + (wn * 32 * CT + c16) * (TBK * 2)) + frag_ko;
    half8 ah0[4], al0[4], ah1[4], al1[4], wh[CS], wl[CS];       // operand fragments of row sub-blocks 0-3 | 4-7 | all column sub-blocks
    auto read_a = [&](int a_slot, int half, half8 (&ah)[4], half8 (&al)[4]) {
        const unsigned char* sA = smem_raw + a_slot * TA_SLOT + a_frag0 + half * 4 * 16 * (TBK * 2);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ah[i] = *reinterpret_cast<const half8*>(sA + i * 16 * (TBK * 2));
            al[i] = *reinterpret_cast<const half8*>(sA + i * 16 * (TBK * 2) + TA_PLANE);
        }
    };
    auto read_w = [&](int w_slot, int chalf) {                 // column sub-blocks chalf * CT .. chalf * CT + CT - 1
        const unsigned char* sW = smem_raw + w_slot * W_SLOT + w_frag0;
#pragma unroll
        for (int j = chalf * CT; j < (chalf + 1) * CT; ++j) {
            wh[j] = *reinterpret_cast<const half8*>(sW + j * 16 * (TBK * 2));
            wl[j] = *reinterpret_cast<const half8*>(sW + j * 16 * (TBK * 2) + W_PLANE);
        }
    };
    f32x4 acc[8][CS];
    // the three products of a k32 step (a_lo.w_hi, a_hi.w_lo, a_hi.w_hi, per accumulator in that order) for 4 row sub-blocks x CT
    // column sub-blocks; consecutive MFMAs go to different accumulators
    auto mfma_group = [&](int rhalf, const half8 (&ah)[4], const half8 (&al)[4], int chalf) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = chalf * CT; j < (chalf + 1) * CT; ++j)
                acc[4 * rhalf + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], wh[j], acc[4 * rhalf + i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = chalf * CT; j < (chalf + 1) * CT; ++j)
                acc[4 * rhalf + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], wl[j], acc[4 * rhalf + i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = chalf * CT; j < (chalf + 1) * CT; ++j)
                acc[4 * rhalf + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], wh[j], acc[4 * rhalf + i][j], 0, 0, 0);
    };

    wait_landed(0);
    lds_barrier();
    int ca = 0, cw = 0, na = 1, nw = 1;
    int64_t gstep = 0;
    int64_t tm = first_tm;
    int tn = first_tn;
    for (int64_t ti = 0; ti < my_tiles; ++ti) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < CS; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // (not prefetched across the tile boundary: the epilogue needs the fragment registers)
        read_w(cw, 0);
        read_w(cw, 1);
        read_a(ca, 0, ah0, al0);
        for (int kt = 0; kt < nk; ++kt, ++gstep) {
            // One k32 step = one ring slot per operand.  Rows 0-63 of the wave tile first (their fragments were read during the previous
            // step), under whose MFMAs the fragments of rows 64-127 arrive; then the barrier / refill, after which the NEXT step's
            // fragments are read under the MFMAs of rows 64-127: rows 0-63's operand behind the barrier, each half of the weight
            // fragments behind the MFMAs that read it last.
            read_a(ca, 1, ah1, al1);
            mfma_group(0, ah0, al0, 0);
            mfma_group(0, ah0, al0, 1);
            const bool more = kt + 1 < nk;
            if (gstep + 1 < total_steps) {
                wait_landed(gstep + 1);
                lds_barrier();                                        // step gstep+1 is in for everyone; this step's slots fully read
                // the activation waves defer the refill of a tile's LAST step to the start of the epilogue (see there)
                if (issued < total_steps && !(is_a && kt + 1 == nk)) issue();
                if (more) read_a(na, 0, ah0, al0);
            } else {
                lds_barrier();
            }
            mfma_group(1, ah1, al1, 0);
            if (more) read_w(nw, 0);
            mfma_group(1, ah1, al1, 1);
            if (more) read_w(nw, 1);
            ca = na;
            cw = nw;
            if (++na == TA_SLOTS) na = 0;
            if (++nw == TW_SLOTS) nw = 0;
        }
        // ---- epilogue in the accumulator layout: lane = column (l & 15) of each column sub-block, register quad of row sub-block rs =
        // rows 16 rs + 4 (l >> 4) + 0..3
        const int64_t row0 = tm * TBM + wm * 128;
        const int col0 = tn * BN + wn * 32 * CT;
        // column parameters of this lane for ALL its column sub-blocks up front: a load waited for in the middle of the epilogue would
        // also wait (in-order counter) for every store issued before it
        float pbias[CS], praw[CS][4];
        NeuronP pnp[CS];
#pragma unroll
        for (int j = 0; j < CS; ++j) {
            const int col = col0 + j * 16 + c16;
            pbias[j] = 0.f;
            if (g.bias) pbias[j] = bt_load_f32(g.bias + col);
            if (EPI == EPI_LIF || EPI == EPI_LIF_ATTN) {
#pragma unroll
                for (int q = 0; q < 4; ++q) praw[j][q] = bt_load_f32(g.lif + (int64_t)q * g.n + col);
            }
        }
        // Tile-boundary protocol.  The counter completes in order and counts stores, so (a) waiting for the parameter loads
        // waits for every DMA issued before them and (b) any wait behind the epilogue's 128 stores waits for the stores.
        // (a): the activation waves issued their last refill one and a half k-steps ago (the one of the last step was
        // deferred), the weight refill is an L2 hit — this wait is short; after it every issued step has landed, which makes
        // the first waits of the next tile (steps issued before this point) unnecessary: landed_upto.  The deferred refill goes
        // out now, before the stores; the waits that follow it come one k-step or more after the last store.
        bt_wait_vm<0>();
        landed_upto = issued;
        if (is_a && issued < total_steps) issue();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < CS; ++j) {
            pnp[j] = NeuronP{0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (EPI == EPI_LIF || EPI == EPI_LIF_ATTN) {
                pnp[j].decay = clampf(praw[j][0], 0.1f, 0.99f);
                pnp[j].adapt = clampf(praw[j][1], 0.001f, 0.1f);
                pnp[j].rdecay = clampf(praw[j][2], 0.1f, 0.95f);
                pnp[j].theta0 = praw[j][3];
            }
        }
        // interior tile and a format the fast path has (wave-uniform): everything the models launch; else the generic path
        const bool interior = tm * TBM + TBM <= g.r;
        if (EPI == EPI_LRELU_MAX) {
            // LeakyReLU, then the max over groups of max_m rows instead of a store (fd/snn_coder.py:476-480).  A lane walks its
            // column's 32 rows in ascending order and keeps a running maximum per group: one integer atomicMax per (group,
            // column) and lane group instead of one per value (the maximum does not depend on the order: exact).
#pragma unroll
            for (int j = 0; j < CS; ++j) {
                const int col = col0 + j * 16 + c16;
                int64_t cur = -1, boundary = 0;
                float best = 0.f;
#pragma unroll
                for (int rs = 0; rs < 8; ++rs) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int64_t rr = row0 + rs * 16 + 4 * g4 + e;
                        if (rr >= g.r) continue;
                        const float v = lrelu02(__fadd_rn(__fmul_rn(acc[rs][j][e], 0.0625f), pbias[j]));
                        if (rr >= boundary) {                 // first row, or a new group starts
                            if (cur >= 0) atomicMax(g.max_keys + cur * g.n + col, float_max_key(best));
                            cur = rr / g.max_m;
                            boundary = (cur + 1) * g.max_m;
                            best = v;
                        } else {
                            best = fmaxf(best, v);
                        }
                    }
                }
                if (cur >= 0) atomicMax(g.max_keys + cur * g.n + col, float_max_key(best));
            }
        } else if (interior && EPI == EPI_LIF_ATTN && !g.c_split && g.c2_split) {
            bt_epilogue_fast<EPI, CS, false, true>(g, acc, row0, col0, c16, g4, pbias, pnp);
        } else if (interior && EPI == EPI_LIF && g.c_split) {
            bt_epilogue_fast<EPI, CS, true, false>(g, acc, row0, col0, c16, g4, pbias, pnp);
        } else if (interior && EPI != EPI_LIF_ATTN && !g.c_split) {
            bt_epilogue_fast<EPI, CS, false, false>(g, acc, row0, col0, c16, g4, pbias, pnp);
        } else
#pragma unroll
        for (int rs = 0; rs < 8; ++rs) {
            const int64_t row = row0 + rs * 16 + 4 * g4;
            if (row >= g.r) continue;
            int2 t4[4];                                      // (point row, neighbour row) of this group's 4 edge rows: once for all column sub-blocks
            if (EPI == EPI_LIF_ATTN) {
#pragma unroll
                for (int u = 0; u < 4; ++u) t4[u] = row + u < g.r ? g.tab[row + u] : make_int2(0, 0);
            }
#pragma unroll
            for (int j = 0; j < CS; ++j) {
                const int col = col0 + j * 16 + c16;
                float a4[4], cq[4] = {0.f, 0.f, 0.f, 0.f}, ckf[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < 4; ++u) a4[u] = __fmul_rn(acc[rs][j][u], 0.0625f);      // undo W x 16
                if (EPI == EPI_LIF_ATTN) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        cq[u] = g.q[(int64_t)t4[u].x * g.ldq + col];
                        ckf[u] = g.kf[(int64_t)t4[u].y * g.ldq + col];
                    }
                }
                epilogue_group4<EPI>(g, a4, row, col, pbias[j], pnp[j], cq, ckf);
            }
            __builtin_amdgcn_sched_barrier(0);              // keep the groups apart (128 accumulators live)
        }
        tm += step_tm;
        tn += step_tn;
        if (tn >= ntn) { tn -= ntn; ++tm; }
    }
}

template <int EPI, int BN>
static int launch_bt_t(const GemmArgs& g, hipStream_t st) {
    constexpr int LDS = TA_SLOTS * TA_SLOT + TW_SLOTS * 2 * BN * TBK * 2;
    static DeviceOnce lds_once;                         // one per kernel instantiation, one bit per device
    SAPCU_SET_MAX_LDS(lds_once, (&gemm_bt_kernel<EPI, BN>), LDS);
    const int g_num_cus_bt = device_cu_count();
    const int64_t tiles = ((g.r + TBM - 1) / TBM) * (g.n / BN);
    const int64_t grid = tiles < g_num_cus_bt ? tiles : g_num_cus_bt;
    hipLaunchKernelGGL((gemm_bt_kernel<EPI, BN>), dim3((unsigned)grid), dim3(512), LDS, st, g);
    SAPCU_CHECK_LAUNCH();
    return SAPCU_OK;
}

// shapes this kernel takes (everything else stays on the ring kernel)
bool gemm_sf16_bt_ok(const GemmArgs& g) {
    if (!g.a_split || !g.w16_hi || !g.w16_lo || g.k <= 0 || g.k % TBK || g.k > g.lda || g.lda % 8) return false;
    if (((uintptr_t)g.a & 15) || ((uintptr_t)g.w16_hi & 15) || ((uintptr_t)g.w16_lo & 15)) return false;
    if (g.n % 128) return false;
    if (4 * (int64_t)g.lda * TBM >= (1LL << 31)) return false;        // 32-bit per-lane DMA offsets inside a tile
    if (g.r < 4 * TBM) return false;                                   // small launches: the 128x128 tiles fill the chip better
    auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };       // the row-layout epilogue works on float4s
    if (g.ldc % 4 || !al16(g.c) || (g.bias && !al16(g.bias))) return false;
    if ((g.epi == EPI_LIF || g.epi == EPI_LIF_ATTN) && !al16(g.lif)) return false;
    if (g.epi == EPI_LIF_ATTN && (g.ldq % 4 || !al16(g.q) || !al16(g.kf) || !al16(g.c2))) return false;
    if (g.epi == EPI_LRELU_MAX) return g.max_keys != nullptr && g.max_m >= 1;
    return g.epi == EPI_BIAS || g.epi == EPI_LIF || g.epi == EPI_LIF_ATTN;
}

int launch_gemm_sf16_bt(const GemmArgs& g, hipStream_t st) {
    if (g.r == 0 || g.n == 0) return SAPCU_OK;
    SAPCU_CHECK_ARG(gemm_sf16_bt_ok(g), "gemm_bt: unsupported shape or epilogue");
    if (g.epi == EPI_LIF || g.epi == EPI_LIF_ATTN) SAPCU_CHECK_ARG(g.lif, "gemm_bt: missing neuron parameters");
    if (g.epi == EPI_LIF_ATTN) SAPCU_CHECK_ARG(g.ldq > 0 && g.tab && g.q && g.kf && g.c2, "gemm_bt: bad attn operands");
    const bool wide = g.n % 256 == 0;
    switch (g.epi) {
        case EPI_LRELU_MAX: return wide ? launch_bt_t<EPI_LRELU_MAX, 256>(g, st) : launch_bt_t<EPI_LRELU_MAX, 128>(g, st);
        case EPI_BIAS: return wide ? launch_bt_t<EPI_BIAS, 256>(g, st) : launch_bt_t<EPI_BIAS, 128>(g, st);
        case EPI_LIF: return wide ? launch_bt_t<EPI_LIF, 256>(g, st) : launch_bt_t<EPI_LIF, 128>(g, st);
        default: return wide ? launch_bt_t<EPI_LIF_ATTN, 256>(g, st) : launch_bt_t<EPI_LIF_ATTN, 128>(g, st);
    }
}


}  // namespace sapcu
