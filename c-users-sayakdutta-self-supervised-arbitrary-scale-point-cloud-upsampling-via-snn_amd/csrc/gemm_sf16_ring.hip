// Split-f16 MFMA GEMM, all-DMA ring version: the hot kernel of the edge GEMMs (gfx950).
//
//   C[r, n] = epi( A[r, k] . W[n, k]^T + bias[n] )
//
// Same arithmetic as gemm_sf16.hip (x = hi + lo halves, three v_mfma_f32_32x32x16_f16 per product into one
// f32 accumulator, weights pre-scaled by 16; since round 3 accumulated pass-major per k32 step — a_lo.w_hi over both k16 halves, then
// a_hi.w_lo, then a_hi.w_hi: what the 16x16x32 kernels of the library produce bit for bit), but A arrives ALREADY SPLIT ("split rows", gemm_epi.h: the
// producing epilogue / element-wise kernel wrote hi and lo halves instead of an f32), so BOTH operands are
// plain f16 planes that stream global -> LDS by LDS-DMA with no registers and no VALU in between.
//
// Why: with one k-step of prefetch the f32-A kernel was bound by operand latency (bias-only GEMM at
// r=294912, k=n=512: 690 us against ~200 us of matrix-pipe time; removing the A loads, the W loads or the
// MFMAs each saved only 15-25 %).  Here the rings keep several k-steps in flight across tile boundaries, and the
// producer waves execute nothing but ds_read + MFMA + 4 DMA issues per k-step.
//
// Structure: ONE 1024-thread workgroup per CU, persistent over 128x128 tiles.
//   waves 0-7   PRODUCERS, 4x2 sub-tiles of 32x64 (2 MFMA tiles, 32 accumulator registers each).
//               k-step = 32: per step  s_waitcnt vmcnt(own DMAs of the next step landed) -> barrier -> issue the DMAs
//               that refill the slots just read -> 12 MFMAs.  Waves 0-3 stream the activation operand (5-slot ring,
//               4 steps ahead: first touches, HBM latency), waves 4-7 the weight operand (3-slot ring, L2 hits).
//   waves 8-15  CONSUMERS (two VALU waves per SIMD): epilogue of the PREVIOUS tile, in 8 pieces of one float4
//               (one row x 4 consecutive columns per lane) between the k-step barriers, so that every global
//               access of the epilogue is 8-16 bytes per lane (gemm_epi.h, epilogue_row4).
//   hand-off    through a 32 KiB LDS area in two halves: first the column-tile-1 half (consumers copy their 16
//               values to registers), then the column-tile-0 half, which STAYS there and is read piece by piece
//               during the next tile — the rings keep streaming underneath.  The area is where the accumulators
//               change from the MFMA layout (lane = column, register = row) to the row layout: producers write
//               row rr of their 32x32 block at slot rr ^ ((rr>>2)&1) (pitch 32 floats: the two rows of one
//               ds_write_b32 land in different bank halves), consumers read float4s (two rows per 16 lanes).
//   LDS         8 operand slots x (hi | lo) x [128 rows][32 halves] = 128 KiB + 32 KiB hand-off = 160 KiB.
//               64-byte rows, 16-byte chunk index XOR-swizzled by (row>>2)&3: conflict-free ds_read_b128; the DMA
//               destination is lane-linear, so the swizzle is applied to each lane's SOURCE address.
#include "common.h"
#include "gemm_epi.h"

namespace sapcu {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) unsigned char lds_byte;
typedef const __attribute__((address_space(1))) void* gptr_t;

constexpr int RBM = 128, RBN = 128, RBK = 32;
// Tile order.  Interleaved (default): the n-tiles of one 128-row panel run at the same time on workgroups of ONE XCD, so
// the A panel leaves HBM once and its other readers hit that XCD's L2.  Contiguous (-DSAPCU_RING_TILES_CONTIGUOUS): each
// workgroup walks a contiguous run of tiles; measured the same speed (+-2 %) but 3.8x the HBM fetch traffic
// (FETCH_SIZE 10.1 GB vs 2.7 GB per launch at r=1179648, k=n=512), because 32 workgroups stream 16 MiB through a 4 MiB L2.
constexpr bool RING_TILES_INTERLEAVED = true;
constexpr int RPLANE = RBM * RBK * 2;            // 8 KiB
constexpr int ROPSLOT = 2 * RPLANE;              // 16 KiB: one operand's hi | lo planes of one k-step
// ring depths (8 operand slots = 128 KiB in all): activations 5 + weights 3; for k <= 128 activations 6 + weights 2
// (those launches are HBM-bound on the activation stream and a tile has only 4 k-steps).  Measured per launch on one box,
// 5+3 against the former 4+4 joint ring: d=512 bias 2768 -> 2592 us, d=256 bias 1300 -> 1214 us; 6+2 at d=512: 2775 us.
constexpr int RING_SLOTS = 8;
constexpr int RING_BYTES = RING_SLOTS * ROPSLOT;   // 128 KiB
constexpr int RHAND = 8 * 16 * 64 * 4;           // 32 KiB: one column-tile half of the accumulators
constexpr int RING_LDS_BYTES = RING_BYTES + RHAND;

// wait until at most N of this wave's vector-memory operations (the ring's DMAs) are outstanding
template <int N>
__device__ __forceinline__ void wait_vm() {
    if (N >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// PM: the pass-major accumulation order (see the k loop) — the only form launched since the whole library moved to it (round 3);
// the k16-major body below it is kept for reference and for A/B builds (PM = false).
template <int EPI, bool VEC, bool PM = false>
__global__ __launch_bounds__(1024) void gemm_ring_kernel(const GemmArgs g) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem_raw[];
    const int RA_SLOTS = g.k <= 128 ? 6 : 5, RW_SLOTS = RING_SLOTS - RA_SLOTS;     // wave-uniform
    const int RA_BYTES = RA_SLOTS * ROPSLOT;
    float* hand = reinterpret_cast<float*>(smem_raw + RING_BYTES);
    constexpr int row_step = RBM;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);    // wave-uniform: keep it (and all it feeds) in scalar registers
    const bool producer = wave < 8;
    const int pw = wave & 7;
    const int wm = pw >> 1, wn = pw & 1;
    const int r32 = lane & 31, h = lane >> 5;

    const int ntn = (g.n + RBN - 1) / RBN;
    const int64_t ntm = (g.r + row_step - 1) / row_step;
    const int64_t ntiles = ntm * ntn;
    const int nk = g.k / RBK;
    // tile order: see RING_TILES_INTERLEAVED
    int64_t first_logical, my_tiles, step_tm;
    int step_tn;
    if (RING_TILES_INTERLEAVED) {
        const int nx = gridDim.x < 8 ? 1 : 8;
        const int xcd = nx == 1 ? 0 : (int)(blockIdx.x & 7);
        const int wg_in_x = nx == 1 ? (int)blockIdx.x : (int)(blockIdx.x >> 3);
        const int wgs_per_x = nx == 1 ? (int)gridDim.x : (int)((gridDim.x - xcd + 7) >> 3);
        const int64_t qd = ntiles / nx, rem = ntiles % nx;
        const int64_t x_begin = xcd * qd + (xcd < rem ? xcd : rem);
        const int64_t x_count = qd + (xcd < rem ? 1 : 0);
        // this workgroup's tiles: local indices wg_in_x, wg_in_x + wgs_per_x, ... < x_count
        my_tiles = x_count > wg_in_x ? (x_count - wg_in_x + wgs_per_x - 1) / wgs_per_x : 0;
        first_logical = x_begin + wg_in_x;
        step_tm = wgs_per_x / ntn;
        step_tn = wgs_per_x - (int)step_tm * ntn;
    } else {
        const int64_t qd = ntiles / gridDim.x, rem = ntiles % gridDim.x;
        my_tiles = qd + ((int64_t)blockIdx.x < rem ? 1 : 0);
        first_logical = blockIdx.x * qd + ((int64_t)blockIdx.x < rem ? (int64_t)blockIdx.x : rem);
        step_tm = 0;
        step_tn = 1;
    }
    if (my_tiles == 0) return;
    // tile walk without divisions in the loops: (tile row, tile column) of the first tile + increment per tile
    const int64_t first_tm = first_logical / ntn;
    const int first_tn = (int)(first_logical - first_tm * ntn);

    if (producer) {
        // DMA role.  Waves 0-3 stream the ACTIVATION operand, waves 4-7 the WEIGHT operand: a wave's vector-memory
        // counter is in order, so two streams with different look-ahead cannot share a wave.  The activation rows are
        // first touches (HBM latency, ~2.7 us under load: measured 2272 us delivery-only against 1339 us when the A
        // panel is L2-hot), the weight tile is always an L2 hit: A runs 4 k-steps ahead in a 5-slot ring, W 2 k-steps
        // ahead in a 3-slot ring.  Per k-step a wave issues 4 DMAs of 1 KiB: rows 32*sub .. 32*sub+31 of its operand
        // (two 16-row pieces) x (hi plane, lo plane); lane -> (row = lane>>2, 16-byte chunk = lane&3).
        const bool is_a = pw < 4;
        const int sub = pw & 3;
        const int depth = is_a ? RA_SLOTS : RW_SLOTS;
        const int dsb = ((lane & 3) ^ ((lane >> 4) & 3)) * 16;     // source byte offset of this lane's (swizzled) LDS chunk
        // fragment rows / swizzles
        const int arow_l = wm * 32 + r32;
        const int asw = (arow_l >> 2) & 3;
        int wrow_l[2], wsw[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            wrow_l[j] = wn * 64 + j * 32 + r32;
            wsw[j] = (wrow_l[j] >> 2) & 3;
        }
        // DMA sources = wave-uniform base of the tile (scalar registers, all per-step arithmetic on the scalar unit)
        // + two 32-bit per-lane byte offsets (first / second 16-row piece) that only change with the tile (edge clamps)
        const int64_t pitch_b = is_a ? 4 * (int64_t)g.lda : 2 * (int64_t)g.k;     // bytes per operand row (hi plane)
        const bool a_il = is_a && (g.lda & 31) == 0;                // interleaved split rows (gemm_epi.h): one line per row and k-step
        const int kstep_bytes = a_il ? 128 : RBK * 2;
        const int64_t lo_delta = is_a ? (a_il ? 64 : 2 * (int64_t)g.lda)
                                      : reinterpret_cast<const char*>(g.w16_lo) - reinterpret_cast<const char*>(g.w16_hi);
        const char* const op_base = is_a ? reinterpret_cast<const char*>(g.a) : reinterpret_cast<const char*>(g.w16_hi);
        const int64_t total_steps = my_tiles * nk;

        // prefetch cursor of this wave's stream: global step index -> (tile, k-step)
        int64_t pf_tile = 0;
        int pf_kt = 0;
        const char* pf_base = nullptr;                             // uniform: first row of the tile's operand
        unsigned pf_off0 = 0, pf_off1 = 0;                         // per lane
        const int lrow = 32 * sub + (lane >> 2);
        const unsigned off0_full = (unsigned)(lrow * (int)pitch_b + dsb);          // interior tiles
        const unsigned off1_full = (unsigned)((lrow + 16) * (int)pitch_b + dsb);
        int64_t pf_tm = first_tm;                                  // tile coordinates of the cursor
        int pf_tn = first_tn;
        auto pf_setup = [&]() {
            const int64_t first = is_a ? pf_tm * row_step : (int64_t)pf_tn * RBN;   // first operand row of the tile
            const int64_t left = (is_a ? g.r : (int64_t)g.n) - first;             // >= 1
            pf_base = op_base + first * pitch_b;
            pf_off0 = off0_full;
            pf_off1 = off1_full;
            if (left < 128) {                                      // edge tile (uniform, rare): clamped rows only feed masked outputs
                const int r0 = lrow < left ? lrow : (int)(left - 1);
                const int r1 = lrow + 16 < left ? lrow + 16 : (int)(left - 1);
                pf_off0 = (unsigned)(r0 * (int)pitch_b + dsb);
                pf_off1 = (unsigned)(r1 * (int)pitch_b + dsb);
            }
        };
        lds_byte* const ring0 = (lds_byte*)(smem_raw + (is_a ? 0 : RA_BYTES) + sub * 2048);
        int issue_slot = 0;                                        // slot the next issued step goes to (wraps at depth)
        int64_t issued = 0;                                        // steps issued by this wave
        auto issue = [&]() {                                       // the 4 DMAs of this wave's next step
            lds_byte* sb = ring0 + issue_slot * ROPSLOT;
            const char* src = pf_base + pf_kt * kstep_bytes;
            {
            __builtin_amdgcn_global_load_lds((gptr_t)(src + pf_off0), sb, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(src + pf_off1), sb + 1024, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(src + lo_delta + pf_off0), sb + RPLANE, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(src + lo_delta + pf_off1), sb + RPLANE + 1024, 16, 0, 0);
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

        // Fragment pipeline (half a k-step deep): the ds_reads of one k16 half are in flight while the six MFMAs of
        // the previous half run, and the k-step barrier sits BETWEEN the two halves of a step, so neither the LDS
        // latency nor the barrier wait leaves the matrix pipe idle.
        struct Frags {
            half8 ah, al, wh[2], wl[2];
        };
        auto read_frags = [&](int a_slot, int w_slot, int k16, Frags& f) {
            const unsigned char* sA = smem_raw + a_slot * ROPSLOT + arow_l * (RBK * 2) + (((k16 * 2 + h) ^ asw) * 16);
            f.ah = *reinterpret_cast<const half8*>(sA);
            f.al = *reinterpret_cast<const half8*>(sA + RPLANE);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const unsigned char* sW = smem_raw + RA_BYTES + w_slot * ROPSLOT + wrow_l[j] * (RBK * 2) + (((k16 * 2 + h) ^ wsw[j]) * 16);
                f.wh[j] = *reinterpret_cast<const half8*>(sW);
                f.wl[j] = *reinterpret_cast<const half8*>(sW + RPLANE);
            }
        };
        f32x16 acc[2];
        auto mfma6 = [&](const Frags& f) {
            // alternate the two accumulators so consecutive MFMAs never depend on each other
            acc[0] = mfma_32x32x16_f16(f.al, f.wh[0], acc[0]);
            acc[1] = mfma_32x32x16_f16(f.al, f.wh[1], acc[1]);
            acc[0] = mfma_32x32x16_f16(f.ah, f.wl[0], acc[0]);
            acc[1] = mfma_32x32x16_f16(f.ah, f.wl[1], acc[1]);
            acc[0] = mfma_32x32x16_f16(f.ah, f.wh[0], acc[0]);
            acc[1] = mfma_32x32x16_f16(f.ah, f.wh[1], acc[1]);
        };
        auto wait_landed = [&](int64_t step) {     // this wave's DMAs of `step` have landed (4 DMAs per step, in order)
            const int64_t ahead = issued - step - 1;
            if (ahead >= 4) wait_vm<16>();
            else if (ahead == 3) wait_vm<12>();
            else if (ahead == 2) wait_vm<8>();
            else if (ahead == 1) wait_vm<4>();
            else wait_vm<0>();
        };
        // step 0 of the first tile
        wait_landed(0);
        lds_barrier();
        int ca = 0, cw = 0;                                         // ring slots of the current step
        int na = 1, nw = 1;                                         // ... of the next step (both rings have >= 2 slots)
        float* const hand_even = hand + pw * 1024 + 5 * h * 32 + r32;
        float* const hand_odd = hand + pw * 1024 + 3 * h * 32 + r32;
        Frags f0, f1;
        read_frags(ca, cw, 0, f0);
        int64_t gstep = 0;
        for (int64_t ti = 0; ti < my_tiles; ++ti) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
            for (int kt = 0; kt < nk; ++kt, ++gstep) {
                read_frags(ca, cw, 1, f1);                          // second half of this step: lands behind mfma6(f0)
                if (PM) {
                    // PASS-MAJOR form: per k32 step and accumulator a_lo.w_hi over both k16 halves, then a_hi.w_lo, then a_hi.w_hi —
                    // the accumulation order of a kernel that issues one v_mfma_f32_16x16x32_f16 per product (fn_edge_chain.hip,
                    // fd_encoder.hip): two chained 32x32x16 over the same 32 k values equal one 16x16x32 bit for bit
                    // (profiles/micro/mfma_f16_shapes_bits.hip)
                    acc[0] = mfma_32x32x16_f16(f0.al, f0.wh[0], acc[0]);
                    acc[1] = mfma_32x32x16_f16(f0.al, f0.wh[1], acc[1]);
                    acc[0] = mfma_32x32x16_f16(f1.al, f1.wh[0], acc[0]);
                    acc[1] = mfma_32x32x16_f16(f1.al, f1.wh[1], acc[1]);
                    acc[0] = mfma_32x32x16_f16(f0.ah, f0.wl[0], acc[0]);
                    acc[1] = mfma_32x32x16_f16(f0.ah, f0.wl[1], acc[1]);
                    acc[0] = mfma_32x32x16_f16(f1.ah, f1.wl[0], acc[0]);
                    acc[1] = mfma_32x32x16_f16(f1.ah, f1.wl[1], acc[1]);
                    acc[0] = mfma_32x32x16_f16(f0.ah, f0.wh[0], acc[0]);
                    acc[1] = mfma_32x32x16_f16(f0.ah, f0.wh[1], acc[1]);
                    acc[0] = mfma_32x32x16_f16(f1.ah, f1.wh[0], acc[0]);
                    acc[1] = mfma_32x32x16_f16(f1.ah, f1.wh[1], acc[1]);
                    if (gstep + 1 < total_steps) {
                        wait_landed(gstep + 1);
                        lds_barrier();
                        if (issued < total_steps) issue();
                        read_frags(na, nw, 0, f0);
                    } else {
                        lds_barrier();
                    }
                    ca = na;
                    cw = nw;
                    if (++na == RA_SLOTS) na = 0;
                    if (++nw == RW_SLOTS) nw = 0;
                    continue;
                }
                mfma6(f0);
                if (gstep + 1 < total_steps) {
                    wait_landed(gstep + 1);
                    lds_barrier();                                  // step gstep+1 is in for everyone; this step's slots fully read
                    if (issued < total_steps) issue();              // step gstep+depth -> the slot just read
                    read_frags(na, nw, 0, f0);                      // first half of the next step: lands behind mfma6(f1)
                } else {
                    lds_barrier();                                  // keep the barrier count per step uniform
                }
                mfma6(f1);
                ca = na;
                cw = nw;
                if (++na == RA_SLOTS) na = 0;
                if (++nw == RW_SLOTS) nw = 0;
            }
            // hand-off in two halves through the 32 KiB area (the ring keeps streaming underneath)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int j = 1 - jj;                               // column tile 1 first (goes to registers), then 0 (stays)
                lds_barrier();                                      // consumers are done with what the area held
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    // register e holds row rr = (e&3) + 8*(e>>2) + 4h of the 32x32 block; slot rr ^ h = rr + 5h - ... :
                    // even e -> (e&3) + 8*(e>>2) + 5h, odd e -> (e&3) + 8*(e>>2) + 3h: two lane bases + constant offsets
                    float* hb = (e & 1) ? hand_odd : hand_even;
                    hb[((e & 3) + 8 * (e >> 2)) * 32] = acc[j][e];  // still x16 (weights pre-scaled): undone in the epilogue
                }
                lds_barrier();                                      // half j is ready
            }
        }
    } else {
        // row layout of this wave's 32x64 sub-tile: lane = (row slot s = lane>>3, column group c4 = lane&7);
        // piece pi = 4*j + p covers row p*8 + s, columns j*32 + c4*4 .. +3 of the sub-tile
        const int c4 = lane & 7, rs = lane >> 3;
        auto hand_at = [&](int p) -> const float4* {
            const int rr = p * 8 + rs;
            return reinterpret_cast<const float4*>(hand + pw * 1024 + (rr ^ ((rr >> 2) & 1)) * 32 + c4 * 4);
        };
        float4 cacc[4];                // column tile 1 of the previous tile; column tile 0 stays in the hand-off area
        int64_t prev_row0 = -1;
        int prev_col0 = 0;
        int64_t cur_tm = first_tm;                                 // tile whose accumulators are handed over next
        int cur_tn = first_tn;
        const int pper = (8 + nk - 1) / nk;                        // pieces per k-step: all 8 within the next tile's k-loop
        lds_barrier();                                             // pairs with the producers' "step 0 has landed"
        for (int64_t ti = 0; ti <= my_tiles; ++ti) {               // last round = drain (no barriers on either side)
            const bool have = ti < my_tiles;
            const bool cons_work = prev_row0 >= 0;
            // Column parameters (bias + 4 neuron rows of this lane's 4 columns) are loaded at the first piece of each
            // column tile; the q/k gathers of a piece are issued at its start and consumed after the neuron loop
            // (~150 VALU instructions of cover).  Nothing loaded is carried across pieces: a loop-carried in-flight
            // register makes the compiler park a vmcnt(0) behind every piece's stores.
            bool ccol[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) ccol[j] = cons_work && (prev_col0 + wn * 64 + j * 32 + c4 * 4) < g.n;
            ColParams4 cp;                       // (no load here: anything in flight at loop entry is waited for in EVERY iteration)
            cp.bias = cp.decay = cp.adapt = cp.rdecay = cp.theta0 = make_float4(0.f, 0.f, 0.f, 0.f);
            int2 tabrow = make_int2(0, 0);       // EPI_LIF_ATTN: lane l holds the (q row, k row) pair of tile row wm*32 + (l & 31)
            if (cons_work && EPI == EPI_LIF_ATTN) {
                const int64_t trow = prev_row0 + wm * 32 + (lane & 31);
                if (trow < g.r) tabrow = g.tab[trow];
                tabrow.x = __builtin_bit_cast(int, settle(__builtin_bit_cast(float, tabrow.x)));
                tabrow.y = __builtin_bit_cast(int, settle(__builtin_bit_cast(float, tabrow.y)));
            }
            for (int kt = 0; kt < nk; ++kt) {
                if (cons_work)
                for (int pi = kt * pper; pi < (kt + 1) * pper && pi < 8; ++pi) {
                    const int j = pi >> 2, p4 = pi & 3, rr = p4 * 8 + rs;
                    const int64_t row = prev_row0 + wm * 32 + rr;
                    const int col = prev_col0 + wn * 64 + j * 32 + c4 * 4;
                    const bool active = (j ? ccol[1] : ccol[0]) && row < g.r;
                    if (p4 == 0) {
                        cp = load_col_params4<EPI, VEC>(g, col, j ? ccol[1] : ccol[0]);
                        // land ALL of them inside this branch: a value still in flight at the join would make the
                        // compiler wait on the join's other side too, i.e. behind the previous piece's stores
                        settle4(cp.bias);
                        if (EPI == EPI_LIF || EPI == EPI_LIF_ATTN) {
                            settle4(cp.decay);
                            settle4(cp.adapt);
                            settle4(cp.rdecay);
                            settle4(cp.theta0);
                        }
                    }
                    float4 cq = make_float4(0.f, 0.f, 0.f, 0.f), ckf = cq;
                    if (EPI == EPI_LIF_ATTN) {
                        const int qr = __shfl(tabrow.x, rr);
                        const int kr = __shfl(tabrow.y, rr);
                        if (active) {
                            cq = ld4_cols<VEC>(g.q + (int64_t)qr * g.ldq, col, g.n);
                            ckf = ld4_cols<VEC>(g.kf + (int64_t)kr * g.ldq, col, g.n);
                        }
                    }
                    if (!active) continue;
                    float4 a;
                    if (j == 0) a = *hand_at(p4);            // column tile 0: straight from the hand-off area
                    else switch (p4) {                        // column tile 1: registers, static indices
                        case 0: a = cacc[0]; break;
                        case 1: a = cacc[1]; break;
                        case 2: a = cacc[2]; break;
                        default: a = cacc[3]; break;
                    }
                    float v[4];
                    epilogue_row4_compute<EPI, VEC>(g, a, row, col, cp, v);
                    epilogue_row4_store<EPI, VEC>(g, v, row, col, cq, ckf);
                }
                if (have) lds_barrier();                            // the producers' mid-step barrier of this k-step
            }
            if (!have) break;
            // take this tile's accumulators: column tile 1 to registers, then column tile 0 stays in the area
            lds_barrier();                                          // (we are done with the area: producers may overwrite)
            lds_barrier();                                          // column tile 1 is in
#pragma unroll
            for (int p = 0; p < 4; ++p) cacc[p] = *hand_at(p);
            lds_barrier();                                          // copied out
            lds_barrier();                                          // column tile 0 is in (and stays)
            prev_row0 = cur_tm * RBM;
            prev_col0 = cur_tn * RBN;
            cur_tm += step_tm;
            cur_tn += step_tn;
            if (cur_tn >= ntn) { cur_tn -= ntn; ++cur_tm; }
        }
    }
}

template <int EPI, bool VEC, bool PM = false>
static int launch_ring_tv(const GemmArgs& g, hipStream_t st) {
    static DeviceOnce lds_once;                         // one per kernel instantiation, one bit per device
    SAPCU_SET_MAX_LDS(lds_once, (&gemm_ring_kernel<EPI, VEC, PM>), RING_LDS_BYTES);
    const int g_num_cus_ring = device_cu_count();
    const int row_step = RBM;
    const int64_t tiles = ((g.r + row_step - 1) / row_step) * ((g.n + RBN - 1) / RBN);
    const int64_t grid = tiles < g_num_cus_ring ? tiles : g_num_cus_ring;
    hipLaunchKernelGGL((gemm_ring_kernel<EPI, VEC, PM>), dim3((unsigned)grid), dim3(1024), RING_LDS_BYTES, st, g);
    SAPCU_CHECK_LAUNCH();
    return SAPCU_OK;
}

// the consumers work on float4 column groups (16-byte loads, 8-16-byte stores) when every operand allows it
static bool ring_vec_ok(const GemmArgs& g) {
    auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    bool ok = g.n % 4 == 0 && g.ldc % 4 == 0 && al16(g.c) && (!g.bias || al16(g.bias));
    if (g.epi == EPI_LIF || g.epi == EPI_LIF_ATTN) ok = ok && al16(g.lif);
    if (g.epi == EPI_RESID || g.epi == EPI_RESID_GELU) ok = ok && g.ldr % 4 == 0 && al16(g.resid);
    if (g.epi == EPI_LIF_ATTN) ok = ok && g.ldq % 4 == 0 && al16(g.q) && al16(g.kf) && al16(g.c2);
    return ok;
}

template <int EPI>
static int launch_ring_t(const GemmArgs& g, hipStream_t st) {
    return ring_vec_ok(g) ? launch_ring_tv<EPI, true, true>(g, st) : launch_ring_tv<EPI, false, true>(g, st);
}
int launch_gemm_sf16_ring(const GemmArgs& g, hipStream_t st) {
    if (g.r == 0 || g.n == 0) return SAPCU_OK;
    SAPCU_CHECK_ARG(g.a_split, "gemm_ring: A must be in split rows");
    SAPCU_CHECK_ARG(g.k > 0 && g.k % RBK == 0 && g.k <= g.lda, "gemm_ring: k=%d must be a multiple of %d and <= lda", g.k, RBK);
    SAPCU_CHECK_ARG(g.lda % 8 == 0 && ((uintptr_t)g.a & 15) == 0 && g.w16_hi && g.w16_lo &&
                        ((uintptr_t)g.w16_hi & 15) == 0 && ((uintptr_t)g.w16_lo & 15) == 0,
                    "gemm_ring: operands must be 16-byte aligned with lda %% 8 == 0 (lda=%d)", g.lda);
    if (g.epi == EPI_LIF || g.epi == EPI_LIF_ATTN) SAPCU_CHECK_ARG(g.lif, "gemm_ring: missing neuron parameters");
    if (g.epi == EPI_RESID || g.epi == EPI_RESID_GELU) SAPCU_CHECK_ARG(g.resid, "gemm_ring: missing residual");
    switch (g.epi) {
        case EPI_BIAS: return launch_ring_t<EPI_BIAS>(g, st);
        case EPI_LIF: return launch_ring_t<EPI_LIF>(g, st);
        case EPI_GELU: return launch_ring_t<EPI_GELU>(g, st);
        case EPI_RESID: return launch_ring_t<EPI_RESID>(g, st);
        case EPI_LRELU: return launch_ring_t<EPI_LRELU>(g, st);
        case EPI_RESID_GELU: return launch_ring_t<EPI_RESID_GELU>(g, st);
        case EPI_LIF_ATTN:
            SAPCU_CHECK_ARG(g.ldq > 0 && g.tab && g.q && g.kf && g.c2, "gemm_ring: bad attn operands");
            return launch_ring_t<EPI_LIF_ATTN>(g, st);
        default: set_error("gemm_ring: unknown epilogue %d", g.epi); return SAPCU_ERR_ARG;
    }
}


}  // namespace sapcu
