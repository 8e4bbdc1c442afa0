// Per-patch and element-wise kernels around the GEMMs: in-patch kNN, neuron loops, positional
// encoding, vector-attention softmax/aggregation, pooling, EdgeConv max, network tails.
// Everything here is HBM/VALU work with channel-fastest ([row][channel]) layouts so that a
// wavefront's 64 lanes touch 256 contiguous bytes.
#include "common.h"
#include "ops.h"
#include "gemm_epi.h"

namespace sapcu {

// =============================================================================================
// In-patch kNN  (fn/snn_coder.py:31-39, fd/snn_coder.py:25-32)
//   score[i][j] = (-xx[j] - (-2 * <xi,xj>)) - xx[i],  <.,.> = c-ascending f32 FMA chain (bitwise
//   what torch's matmul produces for c = 3), xx = sequential sum of rounded squares.
//   top-k by descending score, equal scores by ascending index.
// One 256-thread workgroup per patch; scores live in LDS.
// =============================================================================================
constexpr int PK_CH = 32;   // channel chunk staged in LDS

struct KnnOut {
    int32_t* idx[3];   // up to three tables [b, m, k[t]] written from ONE ranking (fn blocks share the xyz ranking)
    int k[3];
    int n;
};

// The inner products run on the matrix pipe (round 3): v_mfma_f32_16x16x4_f32 IS a k-ascending chain of IEEE f32 FMAs
// (profiles/micro/mfma_f32_exact.hip: 0 of 256 outputs differ from fmaf chains over 256 channels), so a 16 x 16 block of pairs
// takes one MFMA per 4 channels — operands: one LDS dword per lane, row stride 36: conflict-free — and every pair still sees ITS
// chain in ascending channel order: bit-identical to the register-tiled VALU form it replaces (BT x BT pairs per thread, BT + BT
// LDS reads per BT * BT FMAs: bound by the LDS port in feature space).  <xi,xj> = <xj,xi> bit for bit: only the blocks bi <= bj
// of the ceil(m / 16)^2 block grid are computed (m = 48: 6, m = 100: 28), dealt round-robin to the waves.  Channels are padded
// with zeros to a multiple of 4 (+ 0 * 0 is exact).
// NT = 256, or 512 for m > 64: the rank phase — m^3 compares per patch, one wave per row — then has eight waves per workgroup
// (the score keys of a 100-point patch take 40 KiB of LDS: two workgroups per CU either way).
constexpr int PK_LD = 36;   // staging row stride (floats)

template <int NT>
__global__ __launch_bounds__(NT) void patch_knn_kernel(const float* __restrict__ feat, int64_t pstride, int m,
                                                       int c, int ld, const KnnOut out) {
    extern __shared__ float sm[];
    unsigned* K = reinterpret_cast<unsigned*>(sm);   // [m][m+1] score keys
    float* xx = sm + m * (m + 1);       // [m]
    float* F = xx + m;                  // [16 nb][PK_LD]  (rows >= m, channels >= c of the last chunk: zero)
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g4 = lane >> 4;
    const float* base = feat + (int64_t)blockIdx.x * pstride;
    const int nb = (m + 15) >> 4, mpad = 16 * nb;
    constexpr int NW = NT / 64;
    constexpr int NBW = NT == 256 ? 3 : 5;          // blocks per wave: 10 / 4 (m <= 64), 36 / 8 (m <= 128)
    // this wave's blocks: t = wave, wave + NW, ... in the row-major enumeration of bi <= bj (wave-uniform)
    int bi[NBW], bj[NBW];
    {
        int t = 0, q = 0;
#pragma unroll
        for (int u = 0; u < NBW; ++u) bi[u] = bj[u] = -1;
        for (int i = 0; i < nb; ++i)
            for (int j = i; j < nb; ++j, ++t)
                if (t % NW == wave) {
#pragma unroll
                    for (int u = 0; u < NBW; ++u)
                        if (u == q) { bi[u] = i; bj[u] = j; }
                    ++q;
                }
    }
    f32x4 acc[NBW];
#pragma unroll
    for (int u = 0; u < NBW; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    float myxx = 0.f;
    for (int c0 = 0; c0 < c; c0 += PK_CH) {
        const int cw = min(PK_CH, c - c0), cw4 = (cw + 3) & ~3;
        __syncthreads();
        if (cw4 == PK_CH) {                             // (whole chunks: no integer division per element)
            for (int e = tid; e < mpad * PK_CH; e += NT) {
                const int i = e >> 5, cc = e & (PK_CH - 1);
                F[i * PK_LD + cc] = (i < m && cc < cw) ? base[(int64_t)i * ld + c0 + cc] : 0.f;
            }
        } else {
            for (int e = tid; e < mpad * cw4; e += NT) {
                const int i = e / cw4, cc = e % cw4;
                F[i * PK_LD + cc] = (i < m && cc < cw) ? base[(int64_t)i * ld + c0 + cc] : 0.f;
            }
        }
        __syncthreads();
        for (int cc = 0; cc < cw4; cc += 4) {
#pragma unroll
            for (int u = 0; u < NBW; ++u)
                if (bi[u] >= 0)
                    acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(F[(16 * bi[u] + r16) * PK_LD + cc + g4],
                                                                  F[(16 * bj[u] + r16) * PK_LD + cc + g4], acc[u], 0, 0, 0);
        }
        if (tid < m) {
            const float* fr = F + tid * PK_LD;
            for (int cc = 0; cc < cw; ++cc) {
                const float sq = __fmul_rn(fr[cc], fr[cc]);
                myxx = (c0 == 0 && cc == 0) ? sq : __fadd_rn(myxx, sq);
            }
        }
    }
    if (tid < m) xx[tid] = myxx;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NBW; ++u) {
        if (bi[u] < 0) continue;
        const int j = 16 * bj[u] + r16;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = 16 * bi[u] + 4 * g4 + e;
            if (i < m && j < m) {
                const float inner = __fmul_rn(-2.0f, acc[u][e]);
                // stored as the order-preserving integer key of the score (+ 0.0f: -0 and +0 compare equal as floats)
                K[i * (m + 1) + j] = float_max_key(__fadd_rn(__fsub_rn(__fsub_rn(-xx[j], inner), xx[i]), 0.0f));
                if (bi[u] != bj[u]) K[j * (m + 1) + i] = float_max_key(__fadd_rn(__fsub_rn(__fsub_rn(-xx[i], inner), xx[j]), 0.0f));
            }
        }
    }
    __syncthreads();
    // rank by counting: one wave per row, lane owns column lane (and lane + 64 when m > 64); the row's keys sit in the lanes'
    // registers and an inner iteration broadcasts one of them through a scalar register (v_readlane) — no LDS read — and costs
    // one compare and one add per owned column (the float form took ~10 VALU instructions and an LDS read per pair; this phase,
    // m^3 compares per patch, is what bounds the kernel).  Fast pass: 32-bit compares of the score keys alone.  Its ranks are a
    // permutation exactly when the row has no two equal scores (tied columns do not count each other, so the ranks then sum to
    // less than m (m - 1) / 2); otherwise the row is redone with (score, index) pairs compared as ONE 64-bit integer — key << 32 |
    // ~index: greater = higher score, or equal score and lower index.
    const int full = m * (m - 1) / 2;
    for (int i = wave; i < m; i += NT / 64) {
        const unsigned* row = K + i * (m + 1);
        const int j0 = lane, j1 = lane + 64;
        const unsigned key0 = j0 < m ? row[j0] : 0u, key1 = j1 < m ? row[j1] : 0u;
        int r0 = 0, r1 = 0;
        // chunks of 8 columns; columns >= m carry key 0, which is below every score's key, so padding a chunk is harmless
        if (m <= 64) {
            for (int jb = 0; jb < m; jb += 8) {
#pragma unroll
                for (int u = 0; u < 8; ++u) r0 += (unsigned)__builtin_amdgcn_readlane(key0, jb + u) > key0;
            }
        } else {
            for (int jb = 0; jb < 64; jb += 8) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const unsigned kv = (unsigned)__builtin_amdgcn_readlane(key0, jb + u);
                    r0 += kv > key0;
                    r1 += kv > key1;
                }
            }
            for (int jb = 64; jb < m; jb += 8) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const unsigned kv = (unsigned)__builtin_amdgcn_readlane(key1, jb + u - 64);
                    r0 += kv > key0;
                    r1 += kv > key1;
                }
            }
        }
        int tot = (j0 < m ? r0 : 0) + (j1 < m ? r1 : 0);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off);
        if (tot != full) {                              // (wave-uniform) equal scores in this row: exact (score, index) order
            const unsigned long long k0 = ((unsigned long long)key0 << 32) | (unsigned)~j0;
            const unsigned long long k1 = ((unsigned long long)key1 << 32) | (unsigned)~j1;
            r0 = 0;
            r1 = 0;
            for (int jb = 0; jb < m; jb += 8) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int jp = jb + u;
                    const unsigned kk32 = jp < 64 ? (unsigned)__builtin_amdgcn_readlane(key0, jp & 63)
                                                  : (unsigned)__builtin_amdgcn_readlane(key1, jp & 63);
                    const unsigned long long kv = ((unsigned long long)kk32 << 32) | (unsigned)~jp;
                    r0 += kv > k0;
                    r1 += kv > k1;
                }
            }
        }
        for (int t = 0; t < out.n; ++t) {
            const int k = out.k[t];
            int32_t* o = out.idx[t] + ((int64_t)blockIdx.x * m + i) * k;
            if (j0 < m && r0 < k) o[r0] = j0;
            if (j1 < m && r1 < k) o[r1] = j1;
        }
    }
}

int launch_patch_knn_multi(const float* feat, int64_t b, int64_t pstride, int m, int c, int ld, int ntab,
                           const int* ks, int32_t* const* idx, hipStream_t st) {
    if (b == 0) return SAPCU_OK;
    SAPCU_CHECK_ARG(m >= 1 && m <= 128 && c >= 1 && ntab >= 1 && ntab <= 3, "patch_knn: need 1<=m<=128, 1..3 tables (m=%d)", m);
    KnnOut out;
    out.n = ntab;
    for (int t = 0; t < 3; ++t) {
        out.idx[t] = t < ntab ? idx[t] : nullptr;
        out.k[t] = t < ntab ? ks[t] : 0;
        if (t < ntab) SAPCU_CHECK_ARG(ks[t] >= 1 && ks[t] <= m && idx[t], "patch_knn: need 1<=k<=m (k=%d m=%d)", ks[t], m);
    }
    const int nb = (m + 15) / 16;                                   // 1..8 blocks of 16 rows
    const size_t lds = (size_t)(m * (m + 1) + m + 16 * nb * PK_LD) * sizeof(float);
    // m > ~115 needs more than the default 64 KiB of dynamic LDS (m = 128: 83 KiB): raised once per device
#define SAPCU_PK(NT)                                                                                                  \
    do {                                                                                                              \
        static DeviceOnce once;                                                                                       \
        if (lds > 65536) SAPCU_SET_MAX_LDS(once, (&patch_knn_kernel<NT>), 98304);                                      \
        hipLaunchKernelGGL((patch_knn_kernel<NT>), dim3((unsigned)b), dim3(NT), lds, st, feat, pstride, m, c, ld, out); \
    } while (0)
    if (m <= 64) SAPCU_PK(256);
    else SAPCU_PK(512);
#undef SAPCU_PK
    SAPCU_CHECK_LAUNCH();
    return SAPCU_OK;
}

int launch_patch_knn_strided(const float* feat, int64_t b, int64_t pstride, int m, int c, int ld, int k,
                             int32_t* idx, hipStream_t st) {
    return launch_patch_knn_multi(feat, b, pstride, m, c, ld, 1, &k, &idx, st);
}

int launch_patch_knn(const float* feat, int64_t b, int m, int c, int ld, int k, int32_t* idx, hipStream_t st) {
    return launch_patch_knn_strided(feat, b, (int64_t)m * ld, m, c, ld, k, idx, st);
}

// =============================================================================================
// Neuron unit kernel for the STEPPING form fd's encoder runs (NeuronStep2 / NeuronStep2V: fd_encoder.hip, fd_edge_neuron_kernel):
// the input enters at step 0 only (closed gate), every step's spikes are kept, and the refractory state the gate would test is
// examined before each later step exactly as those kernels do.  A thread steps the pair (row 2i, row 2i + 1) of one channel
// (PAIRV: rows i of channels 2c, 2c + 1 — the NeuronStep2V form).  spikes [T, rows, ch]; gate: events with an open gate.
// =============================================================================================
template <bool EIF, bool PAIRV>
__global__ __launch_bounds__(256) void neuron_drive_kernel(const float* __restrict__ x, int64_t rows, int ch, int T, const float* md,
                                                           const float* ta, const float* rd, const float* tb, const float* dT,
                                                           const float* rh, float* __restrict__ so, float* mo, float* to, float* ro,
                                                           int* __restrict__ gate) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t npair = PAIRV ? rows * ((ch + 1) / 2) : ((rows + 1) / 2) * ch;
    if (t >= npair) return;
    int64_t ea, eb;                                   // the pair's two elements (eb == ea: an odd tail, stepped twice)
    int ca, cb;
    if (PAIRV) {
        const int hc = (ch + 1) / 2;
        const int64_t r = t / hc;
        ca = 2 * (int)(t % hc);
        cb = ca + 1 < ch ? ca + 1 : ca;
        ea = r * ch + ca;
        eb = r * ch + cb;
    } else {
        ca = cb = (int)(t % ch);
        const int64_t r = 2 * (t / ch), r1 = r + 1 < rows ? r + 1 : r;
        ea = r * ch + ca;
        eb = r1 * ch + cb;
    }
    auto prm = [&](int c) {
        NeuronP p;
        p.decay = clampf(md[c], 0.1f, 0.99f);
        p.adapt = clampf(ta[c], 0.001f, 0.1f);
        p.rdecay = clampf(rd[c], 0.1f, 0.95f);
        p.theta0 = tb[c];
        p.dT = EIF ? clampf(dT[c], 0.1f, 5.0f) : 0.f;
        p.rh = EIF ? clampf(rh[c], 0.1f, 2.0f) : 0.f;
        return p;
    };
    const f32x2 xin = f32x2{x[ea], x[eb]}, z = f32x2{0.f, 0.f};
    int open = 0;
    f32x2 m2, th2, r2;
    auto run = [&](auto& n) {
        for (int step = 0; step < T; ++step) {
            if (step > 0 && n.gate_open()) ++open;
            const f32x2 sp = n.step(step == 0 ? xin : z, step == 0);
            if (so) {
                so[(int64_t)step * rows * ch + ea] = sp.x;
                so[(int64_t)step * rows * ch + eb] = sp.y;
            }
        }
#ifdef SAPCU_LIF_EXACT_ORDER
        m2 = f32x2{n.sx.m, n.sy.m}; th2 = f32x2{n.sx.th, n.sy.th}; r2 = f32x2{n.sx.r, n.sy.r};
#else
        m2 = n.s.m; th2 = n.s.th; r2 = n.s.r;
#endif
    };
    if (PAIRV) {
        NeuronStep2V<EIF> n(prm(ca), prm(cb));
        run(n);
    } else {
        NeuronStep2<EIF> n(prm(ca));
        run(n);
    }
    if (mo) { mo[ea] = m2.x; mo[eb] = m2.y; }
    if (to) { to[ea] = th2.x; to[eb] = th2.y; }
    if (ro) { ro[ea] = r2.x; ro[eb] = r2.y; }
    if (open && gate) atomicAdd(gate, open);
}

int launch_neuron_drive(const float* x, int64_t rows, int ch, int T, const float* md, const float* ta, const float* rd, const float* tb,
                        const float* dT, const float* rh, int pairv, float* so, float* mo, float* to, float* ro, int* gate, hipStream_t st) {
    const int64_t npair = pairv ? rows * ((ch + 1) / 2) : ((rows + 1) / 2) * ch;
    if (npair == 0) return SAPCU_OK;
    const dim3 grid((unsigned)((npair + 255) / 256)), blk(256);
#define SAPCU_ND(E, V) hipLaunchKernelGGL((neuron_drive_kernel<E, V>), grid, blk, 0, st, x, rows, ch, T, md, ta, rd, tb, dT, rh, so, mo, to, ro, gate)
    if (dT && pairv) SAPCU_ND(true, true);
    else if (dT) SAPCU_ND(true, false);
    else if (pairv) SAPCU_ND(false, true);
    else SAPCU_ND(false, false);
#undef SAPCU_ND
    SAPCU_CHECK_LAUNCH();
    return SAPCU_OK;
}

// =============================================================================================
// Neuron unit kernel (parity tests of the step arithmetic; fn:87-153, fd:198-275)
// =============================================================================================
template <bool EIF>
__global__ __launch_bounds__(256) void neuron_selfloop_kernel(const float* __restrict__ x, int64_t total, int ch,
                                                              int T, const float* md, const float* ta,
                                                              const float* rd, const float* tb, const float* dT,
                                                              const float* rh, float* so, float* mo, float* to,
                                                              float* ro) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int c = (int)(t % ch);
    NeuronP p;
    p.decay = clampf(md[c], 0.1f, 0.99f);
    p.adapt = clampf(ta[c], 0.001f, 0.1f);
    p.rdecay = clampf(rd[c], 0.1f, 0.95f);
    p.theta0 = tb[c];
    p.dT = EIF ? clampf(dT[c], 0.1f, 5.0f) : 0.f;
    p.rh = EIF ? clampf(rh[c], 0.1f, 2.0f) : 0.f;
    NeuronS s = neuron_init(p);
    float v = x[t];
    for (int i = 0; i < T; ++i) v = neuron_step<EIF>(v, s, p);
    if (!EIF) v = lif_selfloop(x[t], p, T);   // LIF spikes come from the production (peeled) loop; states from the step form
    if (so) so[t] = v;
    if (mo) mo[t] = s.m;
    if (to) to[t] = s.th;
    if (ro) ro[t] = s.r;
}

int launch_neuron_selfloop(const float* x, int64_t rows, int ch, int T, const float* md, const float* ta,
                           const float* rd, const float* tb, const float* dT, const float* rh, float* so, float* mo,
                           float* to, float* ro, hipStream_t st) {
    const int64_t total = rows * ch;
    if (total == 0) return SAPCU_OK;
    const unsigned grid = (unsigned)((total + 255) / 256);
    if (dT)
        hipLaunchKernelGGL(neuron_selfloop_kernel<true>, dim3(grid), dim3(256), 0, st, x, total, ch, T, md, ta, rd, tb,
                           dT, rh, so, mo, to, ro);
    else
        hipLaunchKernelGGL(neuron_selfloop_kernel<false>, dim3(grid), dim3(256), 0, st, x, total, ch, T, md, ta, rd,
                           tb, dT, rh, so, mo, to, ro);
    SAPCU_CHECK_LAUNCH();
    return SAPCU_OK;
}

// =============================================================================================
// fn: stem conv1(3->64)+BN -> LIF x T_enc   (fn/snn_coder.py:453-457)
//     rows = b*m points, thread per (row, channel)
// =============================================================================================
__global__ __launch_bounds__(256) void fn_stem_kernel(const float* __restrict__ patch, int64_t rows,
                                                      const float* __restrict__ w /*[64][3]*/,
                                                      const float* __restrict__ bias, const float* __restrict__ lif,
                                                      int T, float* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= rows * 64) return;
    const int c = (int)(t & 63);
    const int64_t r = t >> 6;
    const float x = patch[r * 3], y = patch[r * 3 + 1], z = patch[r * 3 + 2];
    float a = __fmul_rn(w[c * 3], x);
    a = __fmaf_rn(w[c * 3 + 1], y, a);
    a = __fmaf_rn(w[c * 3 + 2], z, a);
    a = __fadd_rn(a, bias[c]);
    out[t] = lif_selfloop(a, load_lif(lif, 64, c), T);
}

int launch_fn_stem(const float* patch, int64_t rows, const float* w, const float* bias, const float* lif, int T,
                   float* out, hipStream_t st) {
    if (rows == 0) return SAPCU_OK;
    hipLaunchKernelGGL(fn_stem_kernel, dim3((unsigned)((rows * 64 + 255) / 256)), dim3(256), 0, st, patch, rows, w,
                       bias, lif, T, out);
    SAPCU_CHECK_LAUNCH();
    return SAPCU_OK;
}

// =============================================================================================
// fn: positional encoding stage 1: fc_delta(3->d)+BN on (x_i - x_j) -> LIF x 4  (fn:310,355-358)
//     rows = b*m*kk edges; output [rows, d].  A workgroup owns a strip of PE_ROWS edge rows and up to 256
//     channels (lane = channel: coalesced stores); the strip's position differences are staged in LDS
//     once; a thread keeps its channel's weights and neuron parameters in registers for the whole strip
//     and runs FOUR rows' neuron chains at a time (independent chains = VALU ILP).
// =============================================================================================
constexpr int PE_ROWS = 32;

__global__ __launch_bounds__(256) void fn_pe1_kernel(const float* __restrict__ patch, const int32_t* __restrict__ idx,
                                                     int64_t rows, int m, int kk, int d,
                                                     const float* __restrict__ w /*[d][3]*/,
                                                     const float* __restrict__ bias, const float* __restrict__ lif,
                                                     int T, float* __restrict__ out, int split) {
    __shared__ float pd[PE_ROWS][3];
    const int64_t row0 = (int64_t)blockIdx.x * PE_ROWS;
    const int c = blockIdx.y * blockDim.x + threadIdx.x;
    if (threadIdx.x < PE_ROWS) {
        const int64_t r = row0 + threadIdx.x;
        float dx = 0.f, dy = 0.f, dz = 0.f;
        if (r < rows) {
            const int64_t pt = r / kk;
            const int64_t patch_i = pt / m;
            const float* pi = patch + pt * 3;
            const float* pj = patch + (patch_i * m + idx[r]) * 3;
            dx = __fsub_rn(pi[0], pj[0]);
            dy = __fsub_rn(pi[1], pj[1]);
            dz = __fsub_rn(pi[2], pj[2]);
        }
        pd[threadIdx.x][0] = dx;
        pd[threadIdx.x][1] = dy;
        pd[threadIdx.x][2] = dz;
    }
    __syncthreads();
    if (c >= d) return;
    const float w0 = w[c * 3], w1 = w[c * 3 + 1], w2 = w[c * 3 + 2], bb = bias[c];
    const NeuronP np = load_lif(lif, d, c);
    const int nrow = (int)((rows - row0) < PE_ROWS ? (rows - row0) : PE_ROWS);
    for (int r4 = 0; r4 < nrow; r4 += 4) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int rr = r4 + u < PE_ROWS ? r4 + u : PE_ROWS - 1;
            float a = __fmul_rn(w0, pd[rr][0]);
            a = __fmaf_rn(w1, pd[rr][1], a);
            a = __fmaf_rn(w2, pd[rr][2], a);
            v[u] = __fadd_rn(a, bb);
        }
        lif_selfloop_n<4>(v, np, T);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (r4 + u < nrow) {
                if (split) store_split(out, row0 + r4 + u, d, c, v[u]);      // operand of the pos-enc ring GEMM
                else out[(row0 + r4 + u) * d + c] = v[u];
            }
    }
}

// Same stage with a lane owning FOUR CONSECUTIVE CHANNELS of a row (d/4 threads per row, 256/(d/4) rows per pass, two
// passes' rows stepped together = 4 packed neuron chains): the outputs leave as 8-byte (split rows) or 16-byte stores
// instead of 2/4-byte ones — the one-channel-per-lane form spent as long on store issue as on the neuron loop.
// Needs d % 4 == 0 with d/4 a power of two <= 256 (the model's 128/256/512); other widths use fn_pe1_kernel.
__global__ __launch_bounds__(256) void fn_pe1_vec4_kernel(const float* __restrict__ patch, const int32_t* __restrict__ idx,
                                                          int64_t rows, int m, int kk, int d,
                                                          const float* __restrict__ w /*[d][3]*/,
                                                          const float* __restrict__ bias, const float* __restrict__ lif,
                                                          int T, float* __restrict__ out, int split) {
    __shared__ float pd[PE_ROWS][3];
    const int tpr = d >> 2;                               // threads per row
    const int rpp = 256 / tpr;                            // rows per pass
    const int rsub = threadIdx.x / tpr;
    const int c = (threadIdx.x - rsub * tpr) * 4;
    // this thread's 4 channels: weights, bias and neuron parameters stay in registers for ALL strips of the workgroup
    // (a strip of 32 rows is only 4-16 elements per thread: reloading ~40 parameters per strip cost as much as the work)
    float wx[4], wy[4], wz[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        wx[u] = w[(c + u) * 3];
        wy[u] = w[(c + u) * 3 + 1];
        wz[u] = w[(c + u) * 3 + 2];
    }
    const float4 bb = ld4(bias + c);
    const float4 pdc = ld4(lif + c), pad = ld4(lif + d + c), prd = ld4(lif + 2 * (int64_t)d + c), pth = ld4(lif + 3 * (int64_t)d + c);
    NeuronP2 np[2];
    np[0].decay = f32x2{clampf(pdc.x, 0.1f, 0.99f), clampf(pdc.y, 0.1f, 0.99f)};
    np[1].decay = f32x2{clampf(pdc.z, 0.1f, 0.99f), clampf(pdc.w, 0.1f, 0.99f)};
    np[0].adapt = f32x2{clampf(pad.x, 0.001f, 0.1f), clampf(pad.y, 0.001f, 0.1f)};
    np[1].adapt = f32x2{clampf(pad.z, 0.001f, 0.1f), clampf(pad.w, 0.001f, 0.1f)};
    np[0].rdecay = f32x2{clampf(prd.x, 0.1f, 0.95f), clampf(prd.y, 0.1f, 0.95f)};
    np[1].rdecay = f32x2{clampf(prd.z, 0.1f, 0.95f), clampf(prd.w, 0.1f, 0.95f)};
    np[0].theta0 = f32x2{pth.x, pth.y};
    np[1].theta0 = f32x2{pth.z, pth.w};
    const NeuronP2 np4[4] = {np[0], np[1], np[0], np[1]};
    const float bs[4] = {bb.x, bb.y, bb.z, bb.w};
    const int64_t nstrips = (rows + PE_ROWS - 1) / PE_ROWS;
    for (int64_t strip = blockIdx.x; strip < nstrips; strip += gridDim.x) {
        const int64_t row0 = strip * PE_ROWS;
        __syncthreads();                                  // the previous strip's differences are no longer read
        if (threadIdx.x < PE_ROWS) {
            const int64_t r = row0 + threadIdx.x;
            float dx = 0.f, dy = 0.f, dz = 0.f;
            if (r < rows) {
                const int64_t pt = r / kk;
                const int64_t patch_i = pt / m;
                const float* pi = patch + pt * 3;
                const float* pj = patch + (patch_i * m + idx[r]) * 3;
                dx = __fsub_rn(pi[0], pj[0]);
                dy = __fsub_rn(pi[1], pj[1]);
                dz = __fsub_rn(pi[2], pj[2]);
            }
            pd[threadIdx.x][0] = dx;
            pd[threadIdx.x][1] = dy;
            pd[threadIdx.x][2] = dz;
        }
        __syncthreads();
        const int nrow = (int)((rows - row0) < PE_ROWS ? (rows - row0) : PE_ROWS);
        for (int r = rsub; r < nrow; r += 2 * rpp) {
            const int rb = r + rpp < PE_ROWS ? r + rpp : r;    // second row of the pair (may be past nrow: computed, not stored)
            float va[4], vb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float a = __fmul_rn(wx[u], pd[r][0]);
                a = __fmaf_rn(wy[u], pd[r][1], a);
                a = __fmaf_rn(wz[u], pd[r][2], a);
                va[u] = __fadd_rn(a, bs[u]);
                float b2 = __fmul_rn(wx[u], pd[rb][0]);
                b2 = __fmaf_rn(wy[u], pd[rb][1], b2);
                b2 = __fmaf_rn(wz[u], pd[rb][2], b2);
                vb[u] = __fadd_rn(b2, bs[u]);
            }
            f32x2 pv[4] = {f32x2{va[0], va[1]}, f32x2{va[2], va[3]}, f32x2{vb[0], vb[1]}, f32x2{vb[2], vb[3]}};
            lif_selfloop_pairs<4>(pv, np4, T);
            const float oa[4] = {pv[0].x, pv[0].y, pv[1].x, pv[1].y}, ob[4] = {pv[2].x, pv[2].y, pv[3].x, pv[3].y};
            if (split) store_split4<true>(out, row0 + r, d, c, d, oa);     // operand of the pos-enc ring GEMM
            else store_f32x4<true>(out, row0 + r, d, c, d, oa);
            if (rb != r && rb < nrow) {
                if (split) store_split4<true>(out, row0 + rb, d, c, d, ob);
                else store_f32x4<true>(out, row0 + rb, d, c, d, ob);
            }
        }
    }
}

int launch_fn_pe1(const float* patch, const int32_t* idx, int64_t rows, int m, int kk, int d, const float* w,
                  const float* bias, const float* lif, int T, float* out, int split, hipStream_t st) {
    if (rows == 0) return SAPCU_OK;
    const int64_t strips = (rows + PE_ROWS - 1) / PE_ROWS;
    SAPCU_CHECK_ARG(strips < 0x7fffffffLL, "pe1: too many rows");
    const int tpr = d / 4;
    const bool al16 = (((uintptr_t)bias | (uintptr_t)lif | (uintptr_t)out) & 15) == 0;
    if (d % 4 == 0 && tpr >= 1 && tpr <= 256 && (tpr & (tpr - 1)) == 0 && al16) {
        const int64_t grid = strips < 256 * 40 ? strips : 256 * 40;     // ~40 resident-or-queued workgroups per CU, each walking strips
        hipLaunchKernelGGL(fn_pe1_vec4_kernel, dim3((unsigned)grid), dim3(256), 0, st, patch, idx, rows, m, kk, d, w, bias, lif,
                           T, out, split);
        SAPCU_CHECK_LAUNCH();
        return SAPCU_OK;
    }
    const int bx = d < 256 ? ((d + 63) / 64) * 64 : 256;
    hipLaunchKernelGGL(fn_pe1_kernel, dim3((unsigned)strips, (unsigned)((d + bx - 1) / bx)), dim3(bx), 0, st, patch, idx,
                       rows, m, kk, d, w, bias, lif, T, out, split);
    SAPCU_CHECK_LAUNCH();
    return SAPCU_OK;
}

// =============================================================================================
// fn: edge-row table: row (patch, point i, neighbour slot j) -> (row of point i, row of its j-th neighbour)
//     in the [b*m, .] per-point tensors.  Lets the GEMM epilogue gather q_i / k_j without divisions.
// =============================================================================================
__global__ __launch_bounds__(256) void edge_table_kernel(const int32_t* __restrict__ idx, int64_t rows, int m, int kk,
                                                         int2* __restrict__ tab) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const int64_t pt = r / kk;
    const int64_t patch_i = pt / m;
    tab[r] = make_int2((int)pt, (int)(patch_i * m + idx[r]));
}

int launch_edge_table(const int32_t* idx, int64_t rows, int m, int kk, int2* tab, hipStream_t st) {
    if (rows == 0) return SAPCU_OK;
    SAPCU_CHECK_ARG(rows < 0x7fffffffLL, "edge_table: too many rows");
    hipLaunchKernelGGL(edge_table_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, idx, rows, m, kk, tab);
    SAPCU_CHECK_LAUNCH();
    return SAPCU_OK;
}

// =============================================================================================
// fn: per-channel softmax over the kk neighbours and aggregation (fn:379-389)
//     attn = softmax(a / sqrt(hd));  res[pt,c] = sum_j attn_j * (v[nbr_j, c] + pe[edge_j, c])
//     thread per (point, channel); the kk logits stay in registers (KK known at compile time for the
//     model's 24/18/12, generic loop otherwise), so `a` and `pe` are read exactly once.
// =============================================================================================
template <int KK>
__global__ __launch_bounds__(256) void fn_softmax_agg_kernel(const float* __restrict__ a, const float* __restrict__ pe,
                                                             const float* __restrict__ v, int ldv,
                                                             const int32_t* __restrict__ idx, int64_t pts, int m,
                                                             int kk_rt, int d, float sqrt_hd, float* __restrict__ res,
                                                             int split) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= pts * d) return;
    const int c = (int)(t % d);
    const int64_t pt = t / d;
    const int64_t patch_i = pt / m;
    const int kk = KK > 0 ? KK : kk_rt;
    const float* ar = a + pt * kk * d + c;
    const float* pr = pe + pt * kk * d + c;
    const int32_t* ir = idx + pt * kk;
    // 1/sqrt(hd) and 1/sum as correctly rounded reciprocals, then multiplications (two roundings where torch's division
    // has one: <= 1 ulp on a logit / weight); the fused form in gemm_sf16_ring.hip computes exactly the same
    const float inv_sqrt_hd = __fdiv_rn(1.0f, sqrt_hd);
    if (KK > 0) {
        float x[KK > 0 ? KK : 1];
        float mx = -__builtin_huge_valf();
#pragma unroll
        for (int j = 0; j < KK; ++j) {
            x[j] = __fmul_rn(ar[(int64_t)j * d], inv_sqrt_hd);
            mx = fmaxf(mx, x[j]);
        }
        float den = 0.f;
#pragma unroll
        for (int j = 0; j < KK; ++j) {
            x[j] = fast_exp(__fsub_rn(x[j], mx));
            den = __fadd_rn(den, x[j]);
        }
        const float inv_den = __fdiv_rn(1.0f, den);
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < KK; ++j) {
            const float vv = __fadd_rn(v[(patch_i * m + ir[j]) * ldv + c], pr[(int64_t)j * d]);
            acc = __fmaf_rn(__fmul_rn(x[j], inv_den), vv, acc);
        }
        if (split) store_split(res, pt, d, c, acc);
        else res[t] = acc;
    } else {
        float mx = -__builtin_huge_valf();
        for (int j = 0; j < kk; ++j) mx = fmaxf(mx, __fmul_rn(ar[(int64_t)j * d], inv_sqrt_hd));
        float den = 0.f;
        for (int j = 0; j < kk; ++j) den = __fadd_rn(den, fast_exp(__fsub_rn(__fmul_rn(ar[(int64_t)j * d], inv_sqrt_hd), mx)));
        const float inv_den = __fdiv_rn(1.0f, den);
        float acc = 0.f;
        for (int j = 0; j < kk; ++j) {
            const float wj = __fmul_rn(fast_exp(__fsub_rn(__fmul_rn(ar[(int64_t)j * d], inv_sqrt_hd), mx)), inv_den);
            const float vv = __fadd_rn(v[(patch_i * m + ir[j]) * ldv + c], pr[(int64_t)j * d]);
            acc = __fmaf_rn(wj, vv, acc);
        }
        if (split) store_split(res, pt, d, c, acc);
        else res[t] = acc;
    }
}

int launch_fn_softmax_agg(const float* a, const float* pe, const float* v, int ldv, const int32_t* idx, int64_t pts,
                          int m, int kk, int d, float sqrt_hd, float* res, int split, hipStream_t st) {
    if (pts == 0) return SAPCU_OK;
    const dim3 grid((unsigned)((pts * d + 255) / 256)), blk(256);
#define SAPCU_SMX(K) \
    hipLaunchKernelGGL((fn_softmax_agg_kernel<K>), grid, blk, 0, st, a, pe, v, ldv, idx, pts, m, kk, d, sqrt_hd, res, split)
    if (kk == 24) SAPCU_SMX(24);
    else if (kk == 18) SAPCU_SMX(18);
    else if (kk == 12) SAPCU_SMX(12);
    else SAPCU_SMX(0);
#undef SAPCU_SMX
    SAPCU_CHECK_LAUNCH();
    return SAPCU_OK;
}

// =============================================================================================
// max over groups of m consecutive rows: in [groups*m, c] -> out [groups, c]  (fn:472, fd:479)
// =============================================================================================
__global__ __launch_bounds__(256) void rowgroup_max_kernel(const float* __restrict__ in, int64_t groups, int m, int c,
                                                           float* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= groups * c) return;
    const int cc = (int)(t % c);
    const int64_t g = t / c;
    const float* p = in + g * m * c + cc;
    float mx = p[0];
    for (int i = 1; i < m; ++i) mx = fmaxf(mx, p[(int64_t)i * c]);
    out[t] = mx;
}

// keys written by the GEMM's EPI_LRELU_MAX epilogue (common.h float_max_key) -> the maxima as floats
__global__ __launch_bounds__(256) void decode_max_keys_kernel(const unsigned* __restrict__ keys, int64_t count, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = float_from_max_key(keys[i]);
}

int launch_decode_max_keys(const unsigned* keys, int64_t count, float* out, hipStream_t st) {
    if (count == 0) return SAPCU_OK;
    hipLaunchKernelGGL(decode_max_keys_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, keys, count, out);
    SAPCU_CHECK_LAUNCH();
    return SAPCU_OK;
}

int launch_rowgroup_max(const float* in, int64_t groups, int m, int c, float* out, hipStream_t st) {
    if (groups == 0) return SAPCU_OK;
    hipLaunchKernelGGL(rowgroup_max_kernel, dim3((unsigned)((groups * c + 255) / 256)), dim3(256), 0, st, in, groups,
                       m, c, out);
    SAPCU_CHECK_LAUNCH();
    return SAPCU_OK;
}

// =============================================================================================
// fn tail: Linear(256->3) -> LayerNorm(3, eps 1e-5) -> F.normalize (fn:545-548); one wave per row
// =============================================================================================
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ __launch_bounds__(256) void fn_tail_kernel(const float* __restrict__ h, int64_t b, int kdim,
                                                      const float* __restrict__ w /*[3][kdim]*/,
                                                      const float* __restrict__ bias, const float* __restrict__ lnw,
                                                      const float* __restrict__ lnb, float* __restrict__ logits,
                                                      float* __restrict__ normals) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= b) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int k = lane; k < kdim; k += 64) {
        const float x = h[row * kdim + k];
        s0 = fmaf(x, w[k], s0);
        s1 = fmaf(x, w[kdim + k], s1);
        s2 = fmaf(x, w[2 * kdim + k], s2);
    }
    s0 = wave_sum(s0) + bias[0];
    s1 = wave_sum(s1) + bias[1];
    s2 = wave_sum(s2) + bias[2];
    if (lane != 0) return;
    if (logits) {
        logits[row * 3] = s0;
        logits[row * 3 + 1] = s1;
        logits[row * 3 + 2] = s2;
    }
    const float mean = (s0 + s1 + s2) / 3.0f;
    const float d0 = s0 - mean, d1 = s1 - mean, d2 = s2 - mean;
    const float var = (d0 * d0 + d1 * d1 + d2 * d2) / 3.0f;
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
    const float y0 = d0 * rstd * lnw[0] + lnb[0];
    const float y1 = d1 * rstd * lnw[1] + lnb[1];
    const float y2 = d2 * rstd * lnw[2] + lnb[2];
    const float nn = fmaxf(sqrtf(y0 * y0 + y1 * y1 + y2 * y2), 1e-12f);
    normals[row * 3] = y0 / nn;
    normals[row * 3 + 1] = y1 / nn;
    normals[row * 3 + 2] = y2 / nn;
}

int launch_fn_tail(const float* h, int64_t b, int kdim, const float* w, const float* bias, const float* lnw,
                   const float* lnb, float* logits, float* normals, hipStream_t st) {
    if (b == 0) return SAPCU_OK;
    hipLaunchKernelGGL(fn_tail_kernel, dim3((unsigned)((b + 3) / 4)), dim3(256), 0, st, h, b, kdim, w, bias, lnw, lnb,
                       logits, normals);
    SAPCU_CHECK_LAUNCH();
    return SAPCU_OK;
}

// f32 [rows, k] (row pitch ld_in) -> split rows (row pitch ld_out floats; gemm_epi.h)
__global__ __launch_bounds__(256) void to_split_rows_kernel(const float* __restrict__ in, int64_t rows, int k, int ld_in,
                                                            float* __restrict__ out, int ld_out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= rows * k) return;
    const int64_t r = t / k;
    const int c = (int)(t % k);
    store_split(out, r, ld_out, c, in[r * ld_in + c]);
}

int launch_to_split_rows(const float* in, int64_t rows, int k, int ld_in, float* out, int ld_out, hipStream_t st) {
    if (rows == 0) return SAPCU_OK;
    hipLaunchKernelGGL(to_split_rows_kernel, dim3((unsigned)((rows * k + 255) / 256)), dim3(256), 0, st, in, rows, k, ld_in,
                       out, ld_out);
    SAPCU_CHECK_LAUNCH();
    return SAPCU_OK;
}

__global__ __launch_bounds__(256) void l2_normalize3_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                            int64_t b) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= b) return;
    const float x = in[t * 3], y = in[t * 3 + 1], z = in[t * 3 + 2];
    const float nn = fmaxf(sqrtf(x * x + y * y + z * z), 1e-12f);
    out[t * 3] = x / nn;
    out[t * 3 + 1] = y / nn;
    out[t * 3 + 2] = z / nn;
}

int launch_l2_normalize3(const float* in, float* out, int64_t b, hipStream_t st) {
    if (b == 0) return SAPCU_OK;
    hipLaunchKernelGGL(l2_normalize3_kernel, dim3((unsigned)((b + 255) / 256)), dim3(256), 0, st, in, out, b);
    SAPCU_CHECK_LAUNCH();
    return SAPCU_OK;
}

// =============================================================================================
// fd block 0: multi-scale EdgeConv(6->64)+BN+LeakyReLU, max over the ks nearest (fd:413-420)
//     feature = cat(x_j - x_i, x_j); LeakyReLU is monotone so it is applied once after the max.
//     thread per (point, scale*64 + channel); out [pts, 64*S]
// =============================================================================================
__global__ __launch_bounds__(256) void fd_edge0_kernel(const float* __restrict__ patch, const int32_t* __restrict__ idx,
                                                       int kmax, int64_t pts, int m, int nscale,
                                                       const int32_t* __restrict__ ks /*[S] device*/,
                                                       const float* __restrict__ w /*[S][64][6]*/,
                                                       const float* __restrict__ bias /*[S][64]*/,
                                                       float* __restrict__ out) {
    const int cs = 64 * nscale;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= pts * cs) return;
    const int col = (int)(t % cs);
    const int64_t pt = t / cs;
    const int64_t patch_i = pt / m;
    const int s = col >> 6;
    const float* ww = w + (int64_t)col * 6;
    const float xi = patch[pt * 3], yi = patch[pt * 3 + 1], zi = patch[pt * 3 + 2];
    const int kuse = min(ks[s], m);
    const int32_t* ir = idx + pt * kmax;
    float mx = -__builtin_huge_valf();
    for (int j = 0; j < kuse; ++j) {
        const float* pj = patch + (patch_i * m + ir[j]) * 3;
        const float xj = pj[0], yj = pj[1], zj = pj[2];
        float a = __fmul_rn(ww[0], __fsub_rn(xj, xi));
        a = __fmaf_rn(ww[1], __fsub_rn(yj, yi), a);
        a = __fmaf_rn(ww[2], __fsub_rn(zj, zi), a);
        a = __fmaf_rn(ww[3], xj, a);
        a = __fmaf_rn(ww[4], yj, a);
        a = __fmaf_rn(ww[5], zj, a);
        mx = fmaxf(mx, a);
    }
    out[t] = lrelu02(__fadd_rn(mx, bias[col]));
}

// Same stage, one workgroup per patch: the patch's coordinates and neighbour lists are staged in LDS once (the
// thread-per-output form re-loaded them from global memory for every channel), a thread keeps its column's six weights
// in registers and walks the patch's points.  Same operations in the same order per output.
constexpr int FDE0_MAXM = 64, FDE0_MAXK = 64;

__global__ __launch_bounds__(256) void fd_edge0_patch_kernel(const float* __restrict__ patch, const int32_t* __restrict__ idx,
                                                             int kmax, int m, int nscale,
                                                             const int32_t* __restrict__ ks /*[S] device*/,
                                                             const float* __restrict__ w /*[S][64][6]*/,
                                                             const float* __restrict__ bias /*[S][64]*/,
                                                             float* __restrict__ out) {
    __shared__ float sp[FDE0_MAXM][3];
    __shared__ int si[FDE0_MAXM][FDE0_MAXK];
    const int64_t pt0 = (int64_t)blockIdx.x * m;
    for (int e = threadIdx.x; e < m * 3; e += 256) sp[e / 3][e % 3] = patch[pt0 * 3 + e];
    for (int e = threadIdx.x; e < m * kmax; e += 256) si[e / kmax][e % kmax] = idx[pt0 * kmax + e];
    __syncthreads();
    const int cs = 64 * nscale;
    for (int col = threadIdx.x; col < cs; col += 256) {
        const int s = col >> 6;
        const float* ww = w + (int64_t)col * 6;
        const float w0 = ww[0], w1 = ww[1], w2 = ww[2], w3 = ww[3], w4 = ww[4], w5 = ww[5], bb = bias[col];
        const int kuse = min(ks[s], m);
        for (int i = 0; i < m; ++i) {
            const float xi = sp[i][0], yi = sp[i][1], zi = sp[i][2];
            float mx = -__builtin_huge_valf();
            for (int j = 0; j < kuse; ++j) {
                const int nb = si[i][j];
                const float xj = sp[nb][0], yj = sp[nb][1], zj = sp[nb][2];
                float a = __fmul_rn(w0, __fsub_rn(xj, xi));
                a = __fmaf_rn(w1, __fsub_rn(yj, yi), a);
                a = __fmaf_rn(w2, __fsub_rn(zj, zi), a);
                a = __fmaf_rn(w3, xj, a);
                a = __fmaf_rn(w4, yj, a);
                a = __fmaf_rn(w5, zj, a);
                mx = fmaxf(mx, a);
            }
            out[(pt0 + i) * cs + col] = lrelu02(__fadd_rn(mx, bb));
        }
    }
}

// Same stage once more, for up to four scales: a lane owns column c of EVERY scale (24 weights in registers), a wave walks pairs
// of points, and everything that is the same for all lanes — neighbour indices, coordinates — arrives through scalar loads (the
// scalar cache) instead of LDS broadcasts, which cost an LDS pass per wave and were what bounded the one-wave-per-scale form.
// The position differences are formed once per edge for all scales, the two points of a pair share packed multiply-adds, a scale
// stops at its own k (its neighbours are a prefix of the sorted list).  Same operations in the same order per output.
template <int NS>
__global__ __launch_bounds__(256) void fd_edge0_scalar_kernel(const float* __restrict__ patch, const int32_t* __restrict__ idx,
                                                              int kmax, int m, const int32_t* __restrict__ ks /*[NS] device*/,
                                                              const float* __restrict__ w /*[NS][64][6]*/,
                                                              const float* __restrict__ bias /*[NS][64]*/,
                                                              float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t pt0 = (int64_t)blockIdx.x * m;
    const float* __restrict__ pp = patch + pt0 * 3;
    const int32_t* __restrict__ ip = idx + pt0 * kmax;
    float wt[NS][6], bb[NS];
    int kuse[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
        for (int t = 0; t < 6; ++t) wt[s][t] = w[((int64_t)s * 64 + lane) * 6 + t];
        bb[s] = bias[s * 64 + lane];
        kuse[s] = min(ks[s], m);
    }
    constexpr int cs = 64 * NS;
    for (int i = 2 * wv; i < m; i += 8) {
        const int i1 = (i + 1 < m) ? i + 1 : i;
        const f32x2 xi = f32x2{pp[3 * i], pp[3 * i1]}, yi = f32x2{pp[3 * i + 1], pp[3 * i1 + 1]},
                    zi = f32x2{pp[3 * i + 2], pp[3 * i1 + 2]};
        const int32_t* __restrict__ ia = ip + (int64_t)i * kmax;
        const int32_t* __restrict__ ib = ip + (int64_t)i1 * kmax;
        f32x2 mx[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) mx[s] = f32x2{-__builtin_huge_valf(), -__builtin_huge_valf()};
        for (int j0 = 0; j0 < kmax; j0 += 4) {
            // four edges per point: their (scalar) index and coordinate loads are issued together
            int na[4], nb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = min(j0 + u, kmax - 1);
                na[u] = ia[j];
                nb[u] = ib[j];
            }
            f32x2 xj[4], yj[4], zj[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                xj[u] = f32x2{pp[3 * na[u]], pp[3 * nb[u]]};
                yj[u] = f32x2{pp[3 * na[u] + 1], pp[3 * nb[u] + 1]};
                zj[u] = f32x2{pp[3 * na[u] + 2], pp[3 * nb[u] + 2]};
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + u;
                const f32x2 dx = xj[u] - xi, dy = yj[u] - yi, dz = zj[u] - zi;
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    if (j < kuse[s]) {
                        f32x2 v = f32x2{wt[s][0], wt[s][0]} * dx;
                        v = pk_fma(f32x2{wt[s][1], wt[s][1]}, dy, v);
                        v = pk_fma(f32x2{wt[s][2], wt[s][2]}, dz, v);
                        v = pk_fma(f32x2{wt[s][3], wt[s][3]}, xj[u], v);
                        v = pk_fma(f32x2{wt[s][4], wt[s][4]}, yj[u], v);
                        v = pk_fma(f32x2{wt[s][5], wt[s][5]}, zj[u], v);
                        mx[s] = f32x2{fmaxf(mx[s].x, v.x), fmaxf(mx[s].y, v.y)};
                    }
                }
            }
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            out[(pt0 + i) * cs + s * 64 + lane] = lrelu02(__fadd_rn(mx[s].x, bb[s]));
            if (i1 != i) out[(pt0 + i1) * cs + s * 64 + lane] = lrelu02(__fadd_rn(mx[s].y, bb[s]));
        }
    }
}

int launch_fd_edge0(const float* patch, const int32_t* idx, int kmax, int64_t pts, int m, int nscale,
                    const int32_t* ks_dev, const float* w, const float* bias, float* out, hipStream_t st) {
    if (pts == 0) return SAPCU_OK;
    if (pts % m == 0 && nscale >= 1 && nscale <= 4 && pts * 3 < 0x7fffffffLL) {
        const dim3 g((unsigned)(pts / m)), b(256);
#define SAPCU_E0(NS) hipLaunchKernelGGL(fd_edge0_scalar_kernel<NS>, g, b, 0, st, patch, idx, kmax, m, ks_dev, w, bias, out)
        if (nscale == 1) SAPCU_E0(1);
        else if (nscale == 2) SAPCU_E0(2);
        else if (nscale == 3) SAPCU_E0(3);
        else SAPCU_E0(4);
#undef SAPCU_E0
        SAPCU_CHECK_LAUNCH();
        return SAPCU_OK;
    }
    if (m <= FDE0_MAXM && kmax <= FDE0_MAXK && pts % m == 0) {
        hipLaunchKernelGGL(fd_edge0_patch_kernel, dim3((unsigned)(pts / m)), dim3(256), 0, st, patch, idx, kmax, m, nscale,
                           ks_dev, w, bias, out);
        SAPCU_CHECK_LAUNCH();
        return SAPCU_OK;
    }
    const int64_t total = pts * 64 * nscale;
    hipLaunchKernelGGL(fd_edge0_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, patch, idx, kmax, pts,
                       m, nscale, ks_dev, w, bias, out);
    SAPCU_CHECK_LAUNCH();
    return SAPCU_OK;
}

// =============================================================================================
// fd neuron stage with external input over T outer steps (fd:408-474).
//   pre-activation at t = 0 comes from memory; for t >= 1 the refractory gate `x * (r <= 0)` is
//   closed (eval-mode spikes are > 0), so the EdgeConv input is multiplied by zero and the state
//   simply evolves; a violated gate increments *gate_violations (checked by the tests).
//   MODE 0: pre = in[row, c]                                  (block 0, after scale_fusion GEMM)
//   MODE 1: pre = lrelu(max_j AB[nbr_j, c] - AB[row, C + c] + shift[c])   (blocks 1..3, factored
//           EdgeConv: W.cat(xj - xi, xj) = (W1 + W2) xj - W1 xi, BN scale folded into W)
//   spikes of step t go to spk[(t*pts + row) * ldo + coff + c]
// =============================================================================================
template <bool EIF, int MODE>
__global__ __launch_bounds__(256) void fd_neuron_kernel(const float* __restrict__ in, int ldi,
                                                        const int32_t* __restrict__ idx, int kk, int m,
                                                        const float* __restrict__ shift, int64_t pts, int C,
                                                        const float* __restrict__ prm, int T, float* __restrict__ spk,
                                                        int ldo, int coff, float* __restrict__ pre_out,
                                                        int* __restrict__ gate_violations, float* __restrict__ spk_split,
                                                        float* __restrict__ x0_out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= pts * C) return;
    const int c = (int)(t % C);
    const int64_t row = t / C;
    float pre;
    if (MODE == 0) {
        pre = in[row * ldi + c];
    } else {
        const int64_t patch_i = row / m;
        const int32_t* ir = idx + row * kk;
        float mx = -__builtin_huge_valf();
        for (int j = 0; j < kk; ++j) mx = fmaxf(mx, in[(patch_i * m + ir[j]) * ldi + c]);
        pre = lrelu02(__fadd_rn(__fsub_rn(mx, in[row * ldi + C + c]), shift[c]));
    }
    if (pre_out) pre_out[t] = pre;
    const NeuronP p = EIF ? load_eif(prm, C, c) : load_lif(prm, C, c);
    // NeuronStep2's packed arithmetic, one element in both halves of the pair: element for element the arithmetic of
    // fd_edge_neuron_kernel and of the fused encoder (fd_encoder.hip) — the three must agree bit for bit.  (Round 4: until then
    // this kernel — block 0's stage — ran neuron_step<EIF>, the scalar form with an IEEE division per EIF step.)
    NeuronStep2<EIF> n(p);
    // x0_out (round 4, the x0 path): the pre-activation goes out as x0 [pts, ldo] and only step 0 is run here — fd_msc_kernel
    // regenerates every step from x0 (and checks the gate); spk / spk_split then hold the step-0 slab alone
    if (x0_out) {
        x0_out[row * ldo + coff + c] = pre;
        T = 1;
    }
    for (int step = 0; step < T; ++step) {
        if (step > 0 && n.gate_open()) atomicAdd(gate_violations, 1);
        const float sp = n.step(step == 0 ? f32x2{pre, pre} : f32x2{0.f, 0.f}, step == 0).x;
        // spk_split: every step as split rows (the operand of the multi_scale_conv GEMM), f32 only for step 0 (the features
        // the next blocks' neighbour search and EdgeConv read) — spk is then the [pts, ldo] slab of step 0 alone
        if (spk_split) {
            store_split(spk_split, (int64_t)step * pts + row, ldo, coff + c, sp);
            if (step == 0) spk[row * ldo + coff + c] = sp;
        } else {
            spk[((int64_t)step * pts + row) * ldo + coff + c] = sp;
        }
    }
}

// MODE 1 restructured: one workgroup per (patch, 128-channel chunk).  The patch's A' rows for the chunk are staged in LDS once
// (m x 128 floats, + the neighbour table as bytes), so the k-neighbour max reads LDS instead of k gathers from L2 per element.
// Round 3: a thread owns FOUR consecutive channels of a point — one 16-byte LDS read per neighbour serves four channels (a
// thread per channel spends three instructions per element and neighbour; at 100-point patches the kernel was 3.5 x its
// 48-point time for 2.1 x the rows) —, 32 channel quads x 8 point slots per workgroup, the two (channel, channel) pairs of a
// point as packed neuron chains (NeuronStep2V), 8 / 16-byte stores.  Same operations per element as before.
constexpr int FDE_CH = 128;
constexpr int FDE_SLOTS = 8;    // point slots: 32 quads x 8 = 256 threads

template <bool EIF>
__global__ __launch_bounds__(256) void fd_edge_neuron_kernel(const float* __restrict__ in, int ldi,
                                                                const int32_t* __restrict__ idx, int kk, int m,
                                                                const float* __restrict__ shift, int64_t pts, int C,
                                                                const float* __restrict__ prm, int T,
                                                                float* __restrict__ spk, int ldo, int coff,
                                                                int* __restrict__ gate_violations, float* __restrict__ spk_split,
                                                                float* __restrict__ x0_out) {
    extern __shared__ float sA[];                       // [m][FDE_CH] f32, then the neighbour table [m][kk] as bytes
    unsigned char* sI = reinterpret_cast<unsigned char*>(sA + (size_t)m * FDE_CH);
    const int tid = threadIdx.x;
    const int c0 = blockIdx.y * FDE_CH;
    const int64_t row0 = (int64_t)blockIdx.x * m;
    // staging: 16-byte loads of the [pts, 2C] operand's first half (C % 4 == 0; channels >= C of the last chunk: zeros)
    for (int e = tid; e < m * (FDE_CH / 4); e += 256) {
        const int i = e / (FDE_CH / 4), q = e % (FDE_CH / 4);
        const int c = c0 + 4 * q;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < C) v = *reinterpret_cast<const float4*>(in + (row0 + i) * ldi + c);
        *reinterpret_cast<float4*>(sA + i * FDE_CH + 4 * q) = v;
    }
    for (int e = tid; e < m * kk; e += 256) sI[e] = (unsigned char)idx[row0 * kk + e];
    __syncthreads();
    const int q4 = tid & 31, slot = tid >> 5;
    const int c = c0 + 4 * q4;
    if (c >= C) return;
    NeuronP p[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) p[u] = EIF ? load_eif(prm, C, c + u) : load_lif(prm, C, c + u);
    const float4 sh = *reinterpret_cast<const float4*>(shift + c);
    const float* tq = sA + 4 * q4;
    for (int i = slot; i < m; i += FDE_SLOTS) {
        const int64_t r = row0 + i;
        const float4 xb = *reinterpret_cast<const float4*>(in + r * ldi + C + c);          // the x_i term of the factored EdgeConv
        const unsigned char* ir = sI + i * kk;
        float4 mx = make_float4(-__builtin_huge_valf(), -__builtin_huge_valf(), -__builtin_huge_valf(), -__builtin_huge_valf());
        int j = 0;
        for (; j + 4 <= kk; j += 4) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(tq + ir[j + u] * FDE_CH);
            mx.x = fmaxf(fmaxf(mx.x, v[0].x), fmaxf(v[1].x, fmaxf(v[2].x, v[3].x)));
            mx.y = fmaxf(fmaxf(mx.y, v[0].y), fmaxf(v[1].y, fmaxf(v[2].y, v[3].y)));
            mx.z = fmaxf(fmaxf(mx.z, v[0].z), fmaxf(v[1].z, fmaxf(v[2].z, v[3].z)));
            mx.w = fmaxf(fmaxf(mx.w, v[0].w), fmaxf(v[1].w, fmaxf(v[2].w, v[3].w)));
        }
        for (; j < kk; ++j) {
            const float4 v = *reinterpret_cast<const float4*>(tq + ir[j] * FDE_CH);
            mx.x = fmaxf(mx.x, v.x);
            mx.y = fmaxf(mx.y, v.y);
            mx.z = fmaxf(mx.z, v.z);
            mx.w = fmaxf(mx.w, v.w);
        }
        const f32x2 pre0 = f32x2{lrelu02(__fadd_rn(__fsub_rn(mx.x, xb.x), sh.x)), lrelu02(__fadd_rn(__fsub_rn(mx.y, xb.y), sh.y))};
        const f32x2 pre1 = f32x2{lrelu02(__fadd_rn(__fsub_rn(mx.z, xb.z), sh.z)), lrelu02(__fadd_rn(__fsub_rn(mx.w, xb.w), sh.w))};
        NeuronStep2V<EIF> n0(p[0], p[1]), n1(p[2], p[3]);
        int nstep = T;
        if (x0_out) {                                    // (see fd_neuron_kernel: x0 out, step 0 only)
            *reinterpret_cast<float4*>(x0_out + r * ldo + coff + c) = make_float4(pre0.x, pre0.y, pre1.x, pre1.y);
            nstep = 1;
        }
        for (int step = 0; step < nstep; ++step) {
            if (step > 0 && (n0.gate_open() || n1.gate_open())) atomicAdd(gate_violations, 1);
            const f32x2 z = f32x2{0.f, 0.f};
            const f32x2 s0 = n0.step(step == 0 ? pre0 : z, step == 0), s1 = n1.step(step == 0 ? pre1 : z, step == 0);
            const float sv[4] = {s0.x, s0.y, s1.x, s1.y};
            if (spk_split) {                             // (see fd_neuron_kernel)
                store_split4<true>(spk_split, (int64_t)step * pts + r, ldo, coff + c, ldo, sv);
                if (step == 0) *reinterpret_cast<float4*>(spk + r * ldo + coff + c) = make_float4(sv[0], sv[1], sv[2], sv[3]);
            } else {
                *reinterpret_cast<float4*>(spk + ((int64_t)step * pts + r) * ldo + coff + c) = make_float4(sv[0], sv[1], sv[2], sv[3]);
            }
        }
    }
}

// debug tap (SAPCU_FD_TAP_X0) of the per-stage path: the MODE 1 pre-activation alone, out[row * ldo + coff + c]
__global__ __launch_bounds__(256) void fd_pre_kernel(const float* __restrict__ in, int ldi, const int32_t* __restrict__ idx, int kk, int m,
                                                     const float* __restrict__ shift, int64_t pts, int C, float* __restrict__ out, int ldo,
                                                     int coff) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= pts * C) return;
    const int c = (int)(t % C);
    const int64_t row = t / C;
    const int64_t patch_i = row / m;
    const int32_t* ir = idx + row * kk;
    float mx = -__builtin_huge_valf();
    for (int j = 0; j < kk; ++j) mx = fmaxf(mx, in[(patch_i * m + ir[j]) * ldi + c]);
    out[row * ldo + coff + c] = lrelu02(__fadd_rn(__fsub_rn(mx, in[row * ldi + C + c]), shift[c]));
}

int launch_fd_pre(const float* in, int ldi, const int32_t* idx, int kk, int m, const float* shift, int64_t pts, int C, float* out,
                  int ldo, int coff, hipStream_t st) {
    if (pts == 0) return SAPCU_OK;
    hipLaunchKernelGGL(fd_pre_kernel, dim3((unsigned)((pts * C + 255) / 256)), dim3(256), 0, st, in, ldi, idx, kk, m, shift, pts, C, out,
                       ldo, coff);
    SAPCU_CHECK_LAUNCH();
    return SAPCU_OK;
}

int launch_fd_neuron(bool eif, int mode, const float* in, int ldi, const int32_t* idx, int kk, int m,
                     const float* shift, int64_t pts, int C, const float* prm, int T, float* spk, int ldo, int coff,
                     float* pre_out, int* gate_violations, hipStream_t st, float* spk_split, float* x0_out) {
    if (pts == 0) return SAPCU_OK;
    // the 16-byte form needs 4-aligned channel geometry (every tensor of the models has it) and byte-sized neighbour indices
    if (mode == 1 && pre_out == nullptr && pts % m == 0 && C % 4 == 0 && ldi % 4 == 0 && ldo % 32 == 0 && coff % 4 == 0 && m <= 256 &&
        (((uintptr_t)in | (uintptr_t)spk | (uintptr_t)shift | (uintptr_t)x0_out) & 15) == 0) {
        const dim3 g2((unsigned)(pts / m), (unsigned)((C + FDE_CH - 1) / FDE_CH));
        const size_t lds = (size_t)m * FDE_CH * sizeof(float) + (((size_t)m * kk + 15) & ~(size_t)15);
        // patches of 121..256 points: the tile + the byte table pass the default 64 KiB dynamic-LDS limit (m = 128, kk = 32: 68 KiB;
        // m = 256, kk = 256: 192 KiB does not fit a CU at all and takes the scalar kernel below)
        if (lds <= 160 * 1024) {
            if (eif) {
                static DeviceOnce once;
                if (lds > 65536) SAPCU_SET_MAX_LDS(once, (&fd_edge_neuron_kernel<true>), 160 * 1024);
                hipLaunchKernelGGL(fd_edge_neuron_kernel<true>, g2, dim3(256), lds, st, in, ldi, idx, kk, m, shift, pts, C, prm,
                                   T, spk, ldo, coff, gate_violations, spk_split, x0_out);
            } else {
                static DeviceOnce once;
                if (lds > 65536) SAPCU_SET_MAX_LDS(once, (&fd_edge_neuron_kernel<false>), 160 * 1024);
                hipLaunchKernelGGL(fd_edge_neuron_kernel<false>, g2, dim3(256), lds, st, in, ldi, idx, kk, m, shift, pts, C,
                                   prm, T, spk, ldo, coff, gate_violations, spk_split, x0_out);
            }
            SAPCU_CHECK_LAUNCH();
            return SAPCU_OK;
        }
    }
    const dim3 grid((unsigned)((pts * C + 255) / 256)), blk(256);
#define SAPCU_FDN(E, M)                                                                                            \
    hipLaunchKernelGGL((fd_neuron_kernel<E, M>), grid, blk, 0, st, in, ldi, idx, kk, m, shift, pts, C, prm, T, spk, \
                       ldo, coff, pre_out, gate_violations, spk_split, x0_out)
    if (eif && mode == 0) SAPCU_FDN(true, 0);
    else if (eif) SAPCU_FDN(true, 1);
    else if (mode == 0) SAPCU_FDN(false, 0);
    else SAPCU_FDN(false, 1);
#undef SAPCU_FDN
    SAPCU_CHECK_LAUNCH();
    return SAPCU_OK;
}

// =============================================================================================
// fd: temporal integration softmax(w).pooled_t (fd:326-328,482-483) + one LIF step from the zero
//     state (fd:485-490).  pooled [T, b, emb]; thread per (patch, feature)
// =============================================================================================
__global__ __launch_bounds__(256) void fd_temporal_kernel(const float* __restrict__ pooled, int T, int64_t b, int emb,
                                                          const float* __restrict__ tw, const float* __restrict__ lif,
                                                          float* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= b * emb) return;
    const int f = (int)(t % emb);
    float mx = tw[0];
    for (int i = 1; i < T; ++i) mx = fmaxf(mx, tw[i]);
    float den = 0.f;
    for (int i = 0; i < T; ++i) den = __fadd_rn(den, expf(__fsub_rn(tw[i], mx)));
    float acc = 0.f;
    for (int i = 0; i < T; ++i) {
        const float w = __fdiv_rn(expf(__fsub_rn(tw[i], mx)), den);
        acc = __fmaf_rn(w, pooled[(int64_t)i * b * emb + t], acc);
    }
    const NeuronP p = load_lif(lif, emb, f);
    NeuronS s = neuron_init(p);
    out[t] = neuron_step<false>(acc, s, p);
}

int launch_fd_temporal(const float* pooled, int T, int64_t b, int emb, const float* tw, const float* lif, float* out,
                       hipStream_t st) {
    if (b == 0) return SAPCU_OK;
    hipLaunchKernelGGL(fd_temporal_kernel, dim3((unsigned)((b * emb + 255) / 256)), dim3(256), 0, st, pooled, T, b,
                       emb, tw, lif, out);
    SAPCU_CHECK_LAUNCH();
    return SAPCU_OK;
}

// =============================================================================================
// fd decoder tail (fd:777-798, 721-725): single-token "attention" over HEADS, to_out, residual,
// LayerNorm(64), fc_hidden(64->32)+BN+GELU, fc_distance(32->1), Softplus(beta 5, threshold 20).
// dim = 64 = one wavefront per row, lane = channel.  Transposed weights ([in][out]) are packed so
// a lane reads consecutive addresses.
// =============================================================================================
__global__ __launch_bounds__(256) void fd_tail_kernel(const float* __restrict__ x /*[b,64]*/,
                                                      const float* __restrict__ qkv /*[b,192]*/, int64_t b, int heads,
                                                      const float* __restrict__ wo_t /*[64][64] in-major*/,
                                                      const float* __restrict__ bo, const float* __restrict__ lnw,
                                                      const float* __restrict__ lnb,
                                                      const float* __restrict__ wh_t /*[64][32]*/,
                                                      const float* __restrict__ bh, const float* __restrict__ wd /*[32]*/,
                                                      const float* __restrict__ bd, float* __restrict__ attn_out,
                                                      float* __restrict__ dist) {
    __shared__ float sh[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;
    if (row >= b) return;   // whole wave exits together; no block-wide barrier below
    const int hd = 64 / heads;
    const float q = qkv[row * 192 + lane], k = qkv[row * 192 + 64 + lane], v = qkv[row * 192 + 128 + lane];
    // per-head dot product: reduce q*k over the hd lanes of a head
    float dot = q * k;
    for (int o = 1; o < hd; o <<= 1) dot += __shfl_xor(dot, o);
    const float logit = dot * (1.0f / sqrtf((float)hd));
    // softmax over heads (each head's logit is replicated on its hd lanes)
    float mx = logit;
    for (int o = hd; o < 64; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    const float e = expf(logit - mx);
    float den = e;
    for (int o = hd; o < 64; o <<= 1) den += __shfl_xor(den, o);
    const float o_c = (e / den) * v;
    sh[wave][lane] = o_c;
    __builtin_amdgcn_wave_barrier();
    float acc = bo[lane];
    for (int i = 0; i < 64; ++i) acc = fmaf(sh[wave][i], wo_t[i * 64 + lane], acc);
    const float y = acc + x[row * 64 + lane];
    const float mean = wave_sum(y) / 64.0f;
    const float dc = y - mean;
    const float var = wave_sum(dc * dc) / 64.0f;
    const float z = dc * (1.0f / sqrtf(var + 1e-5f)) * lnw[lane] + lnb[lane];
    if (attn_out) attn_out[row * 64 + lane] = z;
    __builtin_amdgcn_wave_barrier();
    sh[wave][lane] = z;
    __builtin_amdgcn_wave_barrier();
    float hsum = 0.f;
    if (lane < 32) {
        hsum = bh[lane];
        for (int i = 0; i < 64; ++i) hsum = fmaf(sh[wave][i], wh_t[i * 32 + lane], hsum);
        hsum = gelu_erf(hsum) * wd[lane];
    }
    const float tot = wave_sum(hsum) + bd[0];
    if (lane == 0) {
        const float bx = 5.0f * tot;
        dist[row] = bx > 20.0f ? tot : log1pf(expf(bx)) / 5.0f;
    }
}

int launch_fd_tail(const float* x, const float* qkv, int64_t b, int heads, const float* wo_t, const float* bo,
                   const float* lnw, const float* lnb, const float* wh_t, const float* bh, const float* wd,
                   const float* bd, float* attn_out, float* dist, hipStream_t st) {
    if (b == 0) return SAPCU_OK;
    SAPCU_CHECK_ARG(heads >= 1 && heads <= 64 && (heads & (heads - 1)) == 0, "fd tail: heads must be a power of two <= 64");
    hipLaunchKernelGGL(fd_tail_kernel, dim3((unsigned)((b + 3) / 4)), dim3(256), 0, st, x, qkv, b, heads, wo_t, bo, lnw,
                       lnb, wh_t, bh, wd, bd, attn_out, dist);
    SAPCU_CHECK_LAUNCH();
    return SAPCU_OK;
}

}  // namespace sapcu
