// fp32 MFMA GEMM with neuron/activation epilogues — persistent, wave-specialised (gfx950).
//
//   C[r, n] = epi( A[r, k] . W[n, k]^T + bias[n] )
//
// This one kernel carries every 1x1 convolution / Linear of both networks (BatchNorm is folded into
// W and bias by the packer): rows r are (patch, point[, neighbour]) tuples flattened over a chunk of
// the query batch, so a launch has thousands of 128x128 tiles.
//
// MI355X mapping:
//   * v_mfma_f32_32x32x2_f32 — exact f32 products/accumulation (bitwise an fmaf chain); peak
//     157 TFLOP/s, 64 cycles per instruction per SIMD;
//   * ONE 512-thread workgroup per CU, persistent over tiles.  Waves 0-3 are MFMA PRODUCERS (2x2 over
//     the tile, each 64x64 = 2x2 MFMA tiles, 64 accumulator registers); waves 4-7 are EPILOGUE
//     CONSUMERS.  When a tile's k-loop ends the producers park their accumulators in a 64 KiB LDS
//     staging area and start the next tile; the consumers run the epilogue of the PREVIOUS tile —
//     the T-step neuron loop with membrane/threshold/refractory in registers, gathers, stores — in
//     slices between the k-loop barriers.  Producer and consumer of a SIMD are different waves, so
//     the VALU/transcendental epilogue issues in the shadow of the 64-cycle MFMAs instead of after
//     them (the two-workgroup version ran both phases in lockstep: 2.28 ms vs 1.41 ms for bias-only);
//   * operands are staged global -> registers -> LDS ([row][k], row stride 36 floats), double
//     buffered, one barrier per 32-deep k-step; a lane fetches 4 consecutive k with ds_read_b128 and
//     feeds them to 4 MFMAs (the k order inside an 8-wide group is permuted identically for A and W —
//     legal because the sum over k is order-free up to rounding); fragments are register
//     double-buffered so LDS latency hides behind the previous 16 MFMAs;
//   * 139 KiB LDS of the CU's 160; tiles are dealt so that the n-tiles of one row panel run at the
//     same time on the same XCD (shared L2 for the A panel).
#include "common.h"
#include "gemm_epi.h"

namespace sapcu {

constexpr int BM = 128, BN = 128, BK = 32, LDT = BK + 4;
constexpr int OPER_FLOATS = 2 * (BM + BN) * LDT;          // double-buffered A|W tiles
constexpr int STAGE_FLOATS = 4 * 64 * 64;                 // 4 producer waves x 64 acc regs x 64 lanes
constexpr int LDS_BYTES = (OPER_FLOATS + STAGE_FLOATS) * 4;

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_kernel(const GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* stage = smem + OPER_FLOATS;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const bool producer = wave < 4;
    const int pw = wave & 3;                 // producer index / the producer this consumer serves
    const int wm = pw >> 1, wn = pw & 1;
    const int r32 = lane & 31, h = lane >> 5;

    const int ntn = (g.n + BN - 1) / BN;
    const int64_t ntm = (g.r + BM - 1) / BM;
    const int64_t ntiles = ntm * ntn;
    // XCD-aware dealing: workgroups with equal blockIdx%8 share an XCD; each XCD owns a contiguous range
    // of logical tiles (n fastest) and its workgroups walk it together.
    const int nx = gridDim.x < 8 ? 1 : 8;
    const int xcd = nx == 1 ? 0 : (int)(blockIdx.x & 7);
    const int wg_in_x = nx == 1 ? (int)blockIdx.x : (int)(blockIdx.x >> 3);
    const int wgs_per_x = nx == 1 ? (int)gridDim.x : (int)((gridDim.x - xcd + 7) >> 3);
    const int64_t qd = ntiles / nx, rem = ntiles % nx;
    const int64_t x_begin = xcd * qd + (xcd < rem ? xcd : rem);
    const int64_t x_count = qd + (xcd < rem ? 1 : 0);
    const int nk = g.k / BK;

    // ---- producer-side staging state
    const int srow = (tid & 255) >> 3;
    const int skc = (tid & 7) * 4;
    float4 ra[4], rw[4];
    f32x16 acc[2][2];

    int64_t prev_row0 = -1;   // tile whose accumulators sit in the staging area
    int prev_col0 = 0;

    for (int64_t it = 0;; ++it) {
        const int64_t local = it * wgs_per_x + wg_in_x;
        const bool have = local < x_count;
        if (!have && prev_row0 < 0) break;
        const int64_t logical = x_begin + local;
        const int tn = have ? (int)(logical % ntn) : 0;
        const int64_t tm = have ? logical / ntn : 0;
        const int64_t row0 = tm * BM;
        const int col0 = tn * BN;

        const float* arow[4];
        const float* wrow[4];
        bool aok[4], wok[4];
        if (producer && have) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int64_t row = row0 + srow + 32 * i;
                aok[i] = row < g.r;
                arow[i] = g.a + (aok[i] ? row : 0) * g.lda + skc;
                const int nn = col0 + srow + 32 * i;
                wok[i] = nn < g.n;
                wrow[i] = g.w + (int64_t)(wok[i] ? nn : 0) * g.k + skc;
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        }
        auto load_tile = [&](int k0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ra[i] = aok[i] ? ld4(arow[i] + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
                rw[i] = wok[i] ? ld4(wrow[i] + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        auto store_tile = [&](int buf) {
            float* sA = smem + buf * (BM + BN) * LDT;
            float* sW = sA + BM * LDT;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *reinterpret_cast<float4*>(&sA[(srow + 32 * i) * LDT + skc]) = ra[i];
                *reinterpret_cast<float4*>(&sW[(srow + 32 * i) * LDT + skc]) = rw[i];
            }
        };

        // ---- consumer-side state for the previous tile
        const bool cons_work = !producer && prev_row0 >= 0;
        float cbias[2];
        NeuronP cnp[2];
        bool ccol[2];
        if (cons_work) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = prev_col0 + wn * 64 + j * 32 + r32;
                ccol[j] = col < g.n;
                cbias[j] = settle((ccol[j] && g.bias) ? g.bias[col] : 0.f);
                if (EPI == EPI_LIF || EPI == EPI_LIF_ATTN) {
                    cnp[j] = load_lif(g.lif, g.n, ccol[j] ? col : 0);
                    cnp[j].theta0 = settle(cnp[j].theta0);
                }
            }
        }
        // consumer slice s: groups [s*gper, (s+1)*gper) of the 16 four-element groups of its 64 accumulators
        // (group gi = tile t4 = gi>>2 (= i*2+j), register quad e4 = gi&3 -> rows e4*8 + 4h + 0..3)
        const int gper = (16 + nk - 1) / nk;
        // EPI_LIF_ATTN: lane l holds the (q row, k row) pair of tile row wm*64 + l (one 8-byte load per
        // tile); a group fetches its four pairs with ds_bpermute and its q/k gathers are issued one
        // group AHEAD, so their latency hides behind the previous group's neuron loop.
        int2 tabrow = make_int2(0, 0);
        float nq[4] = {0.f, 0.f, 0.f, 0.f}, nkf[4] = {0.f, 0.f, 0.f, 0.f};
        auto issue_gather = [&](int gi) {
            if (EPI != EPI_LIF_ATTN) return;
            const int t4 = gi >> 2, e4 = gi & 3;
            const int i = t4 >> 1, j = t4 & 1;
            const int col = prev_col0 + wn * 64 + j * 32 + r32;
            const int lrow = i * 32 + 8 * e4 + 4 * h;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int qr = __shfl(tabrow.x, lrow + u);
                const int kr = __shfl(tabrow.y, lrow + u);
                const bool ok = col < g.n && (prev_row0 + wm * 64 + lrow + u) < g.r;
                nq[u] = ok ? g.q[(int64_t)qr * g.ldq + col] : 0.f;
                nkf[u] = ok ? g.kf[(int64_t)kr * g.ldq + col] : 0.f;
            }
        };
        if (cons_work && EPI == EPI_LIF_ATTN) {
            const int64_t trow = prev_row0 + wm * 64 + lane;
            if (trow < g.r) tabrow = g.tab[trow];
            tabrow.x = __builtin_bit_cast(int, settle(__builtin_bit_cast(float, tabrow.x)));
            tabrow.y = __builtin_bit_cast(int, settle(__builtin_bit_cast(float, tabrow.y)));
            issue_gather(0);
        }
        auto consume = [&](int s) {
            if (!cons_work) return;
            for (int gi = s * gper; gi < (s + 1) * gper && gi < 16; ++gi) {
                const int t4 = gi >> 2, e4 = gi & 3;
                const int i = t4 >> 1, j = t4 & 1;
                float cq[4], ckf[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    cq[u] = nq[u];
                    ckf[u] = nkf[u];
                }
                if (gi + 1 < 16) issue_gather(gi + 1);
                if (!(j ? ccol[1] : ccol[0])) continue;
                const int64_t row = prev_row0 + wm * 64 + i * 32 + 8 * e4 + 4 * h;
                if (row >= g.r) continue;
                float a[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) a[u] = stage[(pw * 64 + gi * 4 + u) * 64 + lane];
                // select, don't index: runtime-indexed register arrays would go to scratch
                NeuronP np;
                np.decay = j ? cnp[1].decay : cnp[0].decay;
                np.adapt = j ? cnp[1].adapt : cnp[0].adapt;
                np.rdecay = j ? cnp[1].rdecay : cnp[0].rdecay;
                np.theta0 = j ? cnp[1].theta0 : cnp[0].theta0;
                np.dT = 0.f;
                np.rh = 0.f;
                epilogue_group4<EPI>(g, a, row, prev_col0 + wn * 64 + j * 32 + r32, j ? cbias[1] : cbias[0], np, cq, ckf);
            }
        };

        if (producer && have) {
            load_tile(0);
            store_tile(0);
        }
        lds_barrier();
        int cur = 0;
        for (int kt = 0; kt < nk; ++kt) {
            if (producer) {
                if (have) {
                    if (kt + 1 < nk) load_tile((kt + 1) * BK);
                    const float* sA = smem + cur * (BM + BN) * LDT + (wm * 64 + r32) * LDT + h * 4;
                    const float* sW = smem + cur * (BM + BN) * LDT + BM * LDT + (wn * 64 + r32) * LDT + h * 4;
                    float4 fa[2][2], fb[2][2];   // [buffer][tile]
                    fa[0][0] = ld4(sA);
                    fa[0][1] = ld4(sA + 32 * LDT);
                    fb[0][0] = ld4(sW);
                    fb[0][1] = ld4(sW + 32 * LDT);
#pragma unroll
                    for (int k8 = 0; k8 < BK / 8; ++k8) {
                        const int cb = k8 & 1, nb = cb ^ 1;
                        if (k8 + 1 < BK / 8) {
                            fa[nb][0] = ld4(sA + (k8 + 1) * 8);
                            fa[nb][1] = ld4(sA + 32 * LDT + (k8 + 1) * 8);
                            fb[nb][0] = ld4(sW + (k8 + 1) * 8);
                            fb[nb][1] = ld4(sW + 32 * LDT + (k8 + 1) * 8);
                        }
                        const float av[2][4] = {{fa[cb][0].x, fa[cb][0].y, fa[cb][0].z, fa[cb][0].w},
                                                {fa[cb][1].x, fa[cb][1].y, fa[cb][1].z, fa[cb][1].w}};
                        const float bv[2][4] = {{fb[cb][0].x, fb[cb][0].y, fb[cb][0].z, fb[cb][0].w},
                                                {fb[cb][1].x, fb[cb][1].y, fb[cb][1].z, fb[cb][1].w}};
#pragma unroll
                        for (int e = 0; e < 4; ++e)
#pragma unroll
                            for (int i = 0; i < 2; ++i)
#pragma unroll
                                for (int j = 0; j < 2; ++j)
                                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][e], bv[j][e], acc[i][j], 0, 0, 0);
                    }
                    if (kt + 1 < nk) store_tile(cur ^ 1);
                }
            } else {
                consume(kt);
            }
            lds_barrier();
            cur ^= 1;
        }
        // hand-off: the consumers have drained the staging area (last slice ran before the last barrier)
        if (producer && have) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) stage[(pw * 64 + (i * 2 + j) * 16 + e) * 64 + lane] = acc[i][j][e];
        }
        lds_barrier();
        prev_row0 = have ? row0 : -1;
        prev_col0 = col0;
    }
}

template <int EPI>
static int launch_t(const GemmArgs& g, hipStream_t st) {
    static DeviceOnce lds_once;                         // one per kernel instantiation, one bit per device
    SAPCU_SET_MAX_LDS(lds_once, (&gemm_kernel<EPI>), LDS_BYTES);
    const int g_num_cus = device_cu_count();
    const int64_t tiles = ((g.r + BM - 1) / BM) * ((g.n + BN - 1) / BN);
    const int64_t grid = tiles < g_num_cus ? tiles : g_num_cus;
    hipLaunchKernelGGL((gemm_kernel<EPI>), dim3((unsigned)grid), dim3(512), LDS_BYTES, st, g);
    SAPCU_CHECK_LAUNCH();
    return SAPCU_OK;
}

int launch_gemm(const GemmArgs& g, hipStream_t st) {
    if (g.r == 0 || g.n == 0) return SAPCU_OK;
    SAPCU_CHECK_ARG(g.k > 0 && g.k % BK == 0, "gemm: k=%d must be a positive multiple of %d", g.k, BK);
    SAPCU_CHECK_ARG(g.lda % 4 == 0 && ((uintptr_t)g.a & 15) == 0 && ((uintptr_t)g.w & 15) == 0,
                    "gemm: A/W must be 16-byte aligned with lda %% 4 == 0 (lda=%d)", g.lda);
    switch (g.epi) {
        case EPI_BIAS: return launch_t<EPI_BIAS>(g, st);
        case EPI_LIF: return launch_t<EPI_LIF>(g, st);
        case EPI_GELU: return launch_t<EPI_GELU>(g, st);
        case EPI_RESID: return launch_t<EPI_RESID>(g, st);
        case EPI_LRELU: return launch_t<EPI_LRELU>(g, st);
        case EPI_RESID_GELU: return launch_t<EPI_RESID_GELU>(g, st);
        case EPI_LIF_ATTN:
            SAPCU_CHECK_ARG(g.ldq > 0 && g.tab && g.q && g.kf && g.c2, "gemm: bad attn operands");
            return launch_t<EPI_LIF_ATTN>(g, st);
        default: set_error("gemm: unknown epilogue %d", g.epi); return SAPCU_ERR_ARG;
    }
}

}  // namespace sapcu
