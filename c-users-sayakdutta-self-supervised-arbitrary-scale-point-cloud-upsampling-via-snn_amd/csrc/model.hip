// Model handles and forward orchestration (host code) + the extern "C" ABI of include/sapcu.h.
//
// A forward is a fixed sequence of launches on the caller's stream over caller-owned workspace:
// the 1x1 convolutions / Linears are batched over ALL rows of a chunk of patches and run on the
// MFMA GEMM (gemm_f32.hip) with the neuron loop fused as its epilogue; the irregular parts
// (in-patch kNN, gathers, softmax over neighbours, pooling) are the small kernels of
// patch_ops.hip.  Intermediates of a chunk ([rows, d] f32) live in HBM/Infinity Cache between
// launches — 288 GB of HBM is what lets [b*m*k, d] tensors exist at all (the reference must keep
// b <= 400 for them); the chunk size bounds the footprint.
#include <stdarg.h>
#include <stdlib.h>

#include <new>
#include <vector>

#include "common.h"
#include "ops.h"

namespace sapcu {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---------------------------------------------------------------- slot tables (== packing.py)
enum FnSlot {
    FN_STEM_W = 0, FN_STEM_B, FN_STEM_LIF,
    FN_BLK0 = 3,   // 21 slots per block, 3 blocks
    FN_FINAL_W = FN_BLK0 + 63, FN_FINAL_B, FN_FINAL_LIF,
    FN_FCOUT_W, FN_FCOUT_B,
    FN_MLP0_W, FN_MLP0_B, FN_MLP1_W, FN_MLP1_B, FN_MLP2_W, FN_MLP2_B,
    FN_HEAD_W, FN_HEAD_B, FN_LN_W, FN_LN_B,
    FN_SLOTS
};
enum FnBlkSlot {
    B_FC1_W = 0, B_FC1_B, B_SNN1,
    B_QKV_W, B_QKV_B, B_QKV_LIF,
    B_DELTA_W, B_DELTA_B, B_DELTA_LIF,
    B_DELTA2_W, B_DELTA2_B, B_DELTA2_LIF,
    B_GAMMA_W, B_GAMMA_B, B_GAMMA_LIF,
    B_GAMMA2_W, B_GAMMA2_B,
    B_OUT_W, B_OUT_B,
    B_FC2_W, B_FC2_B,
    B_SLOTS
};
static_assert(B_SLOTS == 21, "fn block slot count");
static_assert(FN_SLOTS == 81, "fn slot count");

enum FdSlot {
    FD_E0_W = 0, FD_E0_B,
    FD_FUSE_W, FD_FUSE_B, FD_SNN0,
    FD_EDGE1_W, FD_EDGE1_SHIFT, FD_SNN1,
    FD_EDGE2_W, FD_EDGE2_SHIFT, FD_SNN2,
    FD_EDGE3_W, FD_EDGE3_SHIFT, FD_SNN3,
    FD_MSC_W, FD_MSC_B,
    FD_TI_W, FD_SNNFC,
    FD_FCIN_W, FD_FCIN_B,
    FD_R0_FC0_W, FD_R0_FC0_B, FD_R0_FC4_W, FD_R0_FC4_B, FD_R0_PROJ_W, FD_R0_PROJ_B,
    FD_R1_FC0_W, FD_R1_FC0_B, FD_R1_FC4_W, FD_R1_FC4_B, FD_R1_PROJ_W, FD_R1_PROJ_B,
    FD_QKV_W, FD_QKV_B,
    FD_WO_T, FD_BO, FD_LN_W, FD_LN_B, FD_WH_T, FD_BH, FD_WD, FD_BD,
    FD_SLOTS
};
static_assert(FD_SLOTS == 42, "fd slot count");

}  // namespace sapcu

struct sapcu_model {
    int kind;
    // fn
    int kv[3];
    // fd
    int k, nscale;
    int ks[8];
    int32_t* ks_dev;
    int* gate_dev;
    // split-f16 GEMM path: the whole blob pre-split (same indexing), activation-range overflow counter
    bool sf16;
    void* w16_hi;
    void* w16_lo;
    void* chain_w;             // fn: fc_delta2 | fc_gamma | fc_gamma2 of the three blocks in MFMA-fragment order (fn_edge_chain.hip)
    int* ovf_dev;
    // common
    int emb, T, heads;
    int64_t chunk;             // patches per chunk: SAPCU_CHUNK, or 0 = from the workspace budget (ws_budget bytes per forward)
    int64_t ws_budget;
    // parity / ablation switches, read from the environment ONCE, at sapcu_model_create (a handle is immutable afterwards: a
    // forward never calls getenv; tests build a second handle under another environment instead of flipping it mid-process)
    bool opt_bt;               // SAPCU_BT=0: split-row GEMMs on the ring kernel only
    bool opt_chain;            // SAPCU_CHAIN=0: fn blocks as the five-kernel edge chain
    bool opt_chain_wide;       // SAPCU_CHAIN=wide: the fused chain with 64-bit gather addresses (the form tensors >= 4 GiB take)
    bool opt_fn_maxfuse;       // SAPCU_FN_MAXFUSE=0: conv_final GEMM + rowgroup_max
    bool opt_fd_maxfuse;       // SAPCU_FD_MAXFUSE=0: multi_scale_conv GEMM + rowgroup_max
    bool opt_fd_split;         // SAPCU_FD_SPLIT=0: fd spikes as f32 rows for every step
    bool opt_fd_fused;         // SAPCU_FD_FUSED=0: fd encoder on the per-stage kernels (through HBM) instead of fd_encoder.hip
    bool opt_fd_x0;            // SAPCU_FD_X0=0: the per-stage path writes T spike slabs for the big-tile GEMM instead of x0 for fd_msc_kernel
    // fd, fused encoder (fd_encoder.hip): scale_fusion | EdgeConv 1-3 | multi_scale_conv in MFMA-fragment order, clamped neuron
    // parameters of the 960 encoder channels
    void* fde_w;
    int64_t fde_off[5];        // offsets (halves) of the five matrices inside fde_w
    float* fde_nprm;
    float* blob;
    int64_t blob_floats;
    std::vector<int64_t> dir;
    const float* p(int slot) const { return blob + dir[slot]; }
};

namespace sapcu {

struct Arena {
    char* base;
    int64_t cap, off;
    template <typename T>
    T* take(int64_t count) {
        const int64_t bytes = ((count * (int64_t)sizeof(T)) + 255) & ~(int64_t)255;
        T* p = reinterpret_cast<T*>(base + off);
        off += bytes;
        return p;
    }
};

static inline int64_t imax(int64_t a, int64_t b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }

// GEMM on split rows: the big-tile kernel for the shapes it takes (allow_bt = false — a handle created under SAPCU_BT=0, or a raw
// entry point asked for the ring kernel — keeps everything on the ring kernel; the two are bit-identical), else the 128x128
// ring kernel.
static bool split_rows_gemm_on_big_tile(const GemmArgs& g, bool allow_bt) { return allow_bt && gemm_sf16_bt_ok(g); }

int launch_gemm_split_rows(const GemmArgs& g, hipStream_t st, bool allow_bt) {
    if (split_rows_gemm_on_big_tile(g, allow_bt)) return launch_gemm_sf16_bt(g, st);
    return launch_gemm_sf16_ring(g, st);
}

static bool env_off(const char* name) {
    const char* e = getenv(name);
    return e && strcmp(e, "0") == 0;
}

// Route a GEMM to the split-f16 kernel when the model carries pre-split weights, else to the f32 MFMA kernel.
static int run_gemm(const sapcu_model* m, GemmArgs& g, hipStream_t st) {
    const bool have16 = m && m->sf16 && g.w >= m->blob && g.w < m->blob + m->blob_floats;
    if (have16) {
        const int64_t off = g.w - m->blob;
        g.w16_hi = (const _Float16*)m->w16_hi + off;
        g.w16_lo = (const _Float16*)m->w16_lo + off;
        g.ovf = m->ovf_dev;
        if (g.a_split) return launch_gemm_split_rows(g, st, m->opt_bt);   // A already split by its producer: all-DMA kernels
        if (g.k % 64 == 0) return launch_gemm_sf16(g, st);
    }
    if (g.a_split || g.c_split || g.c2_split) {
        set_error("run_gemm: split-row operands need the split-f16 kernels");
        return SAPCU_ERR_ARG;
    }
    return launch_gemm(g, st);
}

// fmt bit 0: A is in split rows; bit 1: write C in split rows (both only in split-f16 mode)
static int gemm(const sapcu_model* m, const float* a, int64_t r, int k, int lda, const float* w, int n,
                const float* bias, float* c, int ldc, int epi, hipStream_t st, const float* lif = nullptr,
                int lifT = 0, const float* resid = nullptr, int ldr = 0, int fmt = 0) {
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.a = a; g.r = r; g.k = k; g.lda = lda; g.w = w; g.n = n; g.bias = bias; g.c = c; g.ldc = ldc;
    g.epi = epi; g.lif = lif; g.lif_T = lifT; g.resid = resid; g.ldr = ldr;
    g.a_split = fmt & 1; g.c_split = (fmt >> 1) & 1;
    return run_gemm(m, g, st);
}

#define SAPCU_TRY(expr)                 \
    do {                                \
        int _rc = (expr);               \
        if (_rc != SAPCU_OK) return _rc; \
    } while (0)

static int tap_copy(void* const* taps, int which, int64_t dst_off_bytes, const void* src, int64_t bytes,
                    hipStream_t st) {
    if (!taps || !taps[which] || bytes == 0) return SAPCU_OK;
    SAPCU_CHECK_HIP(hipMemcpyAsync((char*)taps[which] + dst_off_bytes, src, (size_t)bytes, hipMemcpyDeviceToDevice, st));
    return SAPCU_OK;
}

// ============================================================================ fn
struct FnPlan {
    int64_t cb;       // patches per chunk
    int kk[3];
    int64_t edge_floats, edge_floats23;   // per-chunk size of edge buffer 1 (also conv_final's output) and of buffers 2, 3
};

// does block l run its edge chain fused (fn_edge_chain.hip)?  The workspace plan and the forward use the same answer
// (SAPCU_CHAIN=0 at model creation keeps the five-kernel chain).
static bool fn_block_fused(const sapcu_model* m, int l, int mp) {
    return m->sf16 && m->chain_w && m->opt_chain && fn_edge_chain_ok(128 << l, imin(m->kv[l], mp));
}

// floats of one [rows, d] edge buffer per patch: only the unfused blocks materialise edge tensors; buffer 1 also holds
// conv_final's [points, emb] output
static void fn_edge_floats_per_patch(const sapcu_model* m, int mp, int64_t& first, int64_t& others) {
    first = (int64_t)mp * m->emb;
    others = 0;
    for (int l = 0; l < 3; ++l) {
        if (fn_block_fused(m, l, mp)) continue;
        const int64_t e = (int64_t)mp * imin(m->kv[l], mp) * (128 << l);
        first = imax(first, e);
        others = imax(others, e);
    }
}

// bytes of chunk workspace per patch (the footprint is linear in the chunk size): the edge buffers dominate when a block
// runs unfused
static int64_t fn_bytes_per_patch(const sapcu_model* m, int mp) {
    int64_t e1, e23;
    fn_edge_floats_per_patch(m, mp, e1, e23);
    int kmx = 1;
    int64_t idxs = 0;
    for (int l = 0; l < 3; ++l) {
        const int kk = imin(m->kv[l], mp);
        idxs += (int64_t)mp * kk * 4;
        kmx = kmx > kk ? kmx : kk;
    }
    return (e1 + 2 * e23) * 4 + idxs + (int64_t)mp * kmx * 24 + (int64_t)mp * (64 + 192 + 512 + 1536 + 512) * 4 +
           ((int64_t)m->emb + 2048 + 1024 + 512 + 256 + 3) * 4;
}

// patches per chunk: SAPCU_CHUNK when set, else as many as the workspace budget holds (a multiple of 64, at least 64)
static int64_t chunk_patches(const sapcu_model* m, int64_t b, int64_t bytes_per_patch) {
    int64_t cb = m->chunk;
    if (cb <= 0) {
        cb = m->ws_budget / (bytes_per_patch > 0 ? bytes_per_patch : 1);
        cb = cb < 64 ? 64 : (cb / 64) * 64;
    }
    if (cb > b) cb = b;
    return cb < 1 ? 1 : cb;
}

static FnPlan fn_plan(const sapcu_model* m, int64_t b, int mp) {
    FnPlan pl;
    pl.cb = chunk_patches(m, b, fn_bytes_per_patch(m, mp));
    for (int l = 0; l < 3; ++l) pl.kk[l] = imin(m->kv[l], mp);
    int64_t e1, e23;
    fn_edge_floats_per_patch(m, mp, e1, e23);
    pl.edge_floats = e1 * pl.cb;
    pl.edge_floats23 = e23 * pl.cb;
    return pl;
}

static int64_t fn_ws_bytes(const sapcu_model* m, int64_t b, int mp) {
    const FnPlan pl = fn_plan(m, b, mp);
    const int64_t P = pl.cb * mp;
    int64_t fl = 0;
    auto add = [&](int64_t n, int64_t esz) { fl += ((n * esz) + 255) & ~(int64_t)255; };
    for (int l = 0; l < 3; ++l) add(P * pl.kk[l], 4);          // idx
    { int kmx = 1; for (int l = 0; l < 3; ++l) kmx = kmx > pl.kk[l] ? kmx : pl.kk[l]; add(P * kmx, 8); add(P * kmx, 16); }   // edge table, position differences
    add(P * 64, 4); add(P * 192, 4); add(P * 512, 4); add(P * 1536, 4); add(P * 512, 4);
    add(pl.edge_floats, 4); add(pl.edge_floats23, 4); add(pl.edge_floats23, 4);
    add(pl.cb * m->emb, 4); add(pl.cb * 2048, 4); add(pl.cb * 1024, 4); add(pl.cb * 512, 4); add(pl.cb * 256, 4);
    add(pl.cb * 3, 4);
    return fl + 256;
}

static int fn_forward(const sapcu_model* m, const float* patch, int64_t b, int mp, const int32_t* knn_in,
                      int32_t* knn_out, float* normals, void* ws, int64_t ws_bytes, void* const* taps,
                      hipStream_t st) {
    const FnPlan pl = fn_plan(m, b, mp);
    if (ws_bytes < fn_ws_bytes(m, b, mp)) {
        set_error("fn_forward: workspace %lld B < required %lld B", (long long)ws_bytes, (long long)fn_ws_bytes(m, b, mp));
        return SAPCU_ERR_WORKSPACE;
    }
    // offsets of the three tables inside knn_in / knn_out ([b,m,k0] | [b,m,k1] | [b,m,k2])
    int64_t tab_off[3];
    tab_off[0] = 0;
    tab_off[1] = b * mp * pl.kk[0];
    tab_off[2] = tab_off[1] + b * mp * pl.kk[1];

    for (int64_t s = 0; s < b; s += pl.cb) {
        const int64_t cb = (b - s) < pl.cb ? (b - s) : pl.cb;
        const int64_t P = cb * mp;
        Arena A{(char*)(((uintptr_t)ws + 255) & ~(uintptr_t)255), ws_bytes, 0};
        int32_t* idx[3];
        for (int l = 0; l < 3; ++l) idx[l] = A.take<int32_t>(pl.cb * mp * pl.kk[l]);
        int kmx = 1;
        for (int l = 0; l < 3; ++l) kmx = kmx > pl.kk[l] ? kmx : pl.kk[l];
        int2* tab = A.take<int2>(pl.cb * mp * kmx);
        float4* pdiff = A.take<float4>(pl.cb * mp * kmx);
        float* feat0 = A.take<float>(pl.cb * mp * 64);
        float* cat = A.take<float>(pl.cb * mp * 192);
        float* X = A.take<float>(pl.cb * mp * 512);
        float* QKV = A.take<float>(pl.cb * mp * 1536);
        float* RES = A.take<float>(pl.cb * mp * 512);
        float* B1 = A.take<float>(pl.edge_floats);
        float* B2 = A.take<float>(pl.edge_floats23);
        float* B3 = A.take<float>(pl.edge_floats23);
        float* pooled = A.take<float>(pl.cb * m->emb);
        float* enc = A.take<float>(pl.cb * 2048);
        float* h1 = A.take<float>(pl.cb * 1024);
        float* h2 = A.take<float>(pl.cb * 512);
        float* h3 = A.take<float>(pl.cb * 256);
        float* logits = A.take<float>(pl.cb * 3);
        const float* pc = patch + s * mp * 3;

        // in-patch neighbour tables: replayed (reference KNNCache) or computed from this chunk (one xyz
        // ranking per patch serves the three blocks' k values)
        if (!knn_in) SAPCU_TRY(launch_patch_knn_multi(pc, cb, (int64_t)mp * 3, mp, 3, 3, 3, pl.kk, idx, st));
        for (int l = 0; l < 3; ++l) {
            const int64_t cnt = P * pl.kk[l];
            if (knn_in)
                SAPCU_CHECK_HIP(hipMemcpyAsync(idx[l], knn_in + tab_off[l] + s * mp * pl.kk[l], cnt * 4,
                                               hipMemcpyDeviceToDevice, st));
            if (knn_out)
                SAPCU_CHECK_HIP(hipMemcpyAsync(knn_out + tab_off[l] + s * mp * pl.kk[l], idx[l], cnt * 4,
                                               hipMemcpyDeviceToDevice, st));
        }
        SAPCU_TRY(launch_fn_stem(pc, P, m->p(FN_STEM_W), m->p(FN_STEM_B), m->p(FN_STEM_LIF), m->T, feat0, st));
        SAPCU_TRY(tap_copy(taps, SAPCU_FN_TAP_STEM, s * mp * 64 * 4, feat0, P * 64 * 4, st));

        const float* fin = feat0;
        int ldin = 64;
        for (int l = 0; l < 3; ++l) {
            const int d = 128 << l;
            const int kk = pl.kk[l];
            const int64_t R = P * kk;
            const int sb = FN_BLK0 + l * B_SLOTS;
            // Tensors that only feed another GEMM travel as split rows (SP) in split-f16 mode: their producer
            // writes f16 hi/lo halves and the consuming GEMM streams them by LDS-DMA (gemm_sf16_ring.hip).
            const int SP = m->sf16 ? 1 : 0;
            // x = LIF(fc1(feat))                                                    fn:317-320
            SAPCU_TRY(gemm(m, fin, P, 64, ldin, m->p(sb + B_FC1_W), d, m->p(sb + B_FC1_B), X, d, EPI_LIF, st,
                           m->p(sb + B_SNN1), 4, nullptr, 0, SP << 1));
            // q|k|v = LIF(w_qs|w_ks|w_vs (x))                                       fn:322-335
            SAPCU_TRY(gemm(m, X, P, d, d, m->p(sb + B_QKV_W), 3 * d, m->p(sb + B_QKV_B), QKV, 3 * d, EPI_LIF, st,
                           m->p(sb + B_QKV_LIF), 4, nullptr, 0, SP));
            const float sqrt_hd = (float)sqrt((double)(d / m->heads));
            if (fn_block_fused(m, l, mp)) {
                // the whole edge chain in one kernel, activations in LDS (fn_edge_chain.hip)      fn:355-389
                ChainArgs ca;
                memset(&ca, 0, sizeof(ca));
                ca.P = P; ca.m = mp; ca.qkv = QKV; ca.ldq = 3 * d;
                ca.wd = m->p(sb + B_DELTA_W); ca.bd = m->p(sb + B_DELTA_B); ca.lifd = m->p(sb + B_DELTA_LIF);
                static const int64_t cw_off[3] = {0, (int64_t)3 * 2 * 128 * 128, (int64_t)3 * 2 * (128 * 128 + 256 * 256)};
                const _Float16* cw = (const _Float16*)m->chain_w + cw_off[l];
                ca.w1p = cw; ca.b1 = m->p(sb + B_DELTA2_B); ca.lif1 = m->p(sb + B_DELTA2_LIF);
                ca.w2p = cw + (int64_t)d * d * 2; ca.b2 = m->p(sb + B_GAMMA_B); ca.lif2 = m->p(sb + B_GAMMA_LIF);
                ca.w3p = cw + (int64_t)2 * d * d * 2; ca.b3 = m->p(sb + B_GAMMA2_B);
                ca.inv_sqrt_hd = 1.0f / sqrt_hd;
                ca.res = RES; ca.res_split = SP; ca.T = 4; ca.wide_offsets = m->opt_chain_wide ? 1 : 0;
                SAPCU_TRY(launch_fn_edge_chain(ca, pc, idx[l], d, kk, tab, pdiff, st));
            } else {
                // pe1 = LIF(fc_delta(x_i - x_j))                                        fn:310,355-358
                SAPCU_TRY(launch_fn_pe1(pc, idx[l], R, mp, kk, d, m->p(sb + B_DELTA_W), m->p(sb + B_DELTA_B),
                                        m->p(sb + B_DELTA_LIF), 4, B1, SP, st));
                // pe = LIF(fc_delta2(pe1)) -> B2, and in the same epilogue attn_in = q_i - k_j + pe -> B3   fn:360-368
                {
                    GemmArgs g;
                    memset(&g, 0, sizeof(g));
                    g.a = B1; g.r = R; g.k = d; g.lda = d; g.w = m->p(sb + B_DELTA2_W); g.n = d;
                    g.bias = m->p(sb + B_DELTA2_B); g.c = B2; g.ldc = d; g.epi = EPI_LIF_ATTN;
                    g.lif = m->p(sb + B_DELTA2_LIF); g.lif_T = 4; g.c2 = B3;
                    g.q = QKV; g.kf = QKV + d; g.ldq = 3 * d; g.tab = tab;
                    g.a_split = SP; g.c2_split = SP;
                    SAPCU_TRY(launch_edge_table(idx[l], R, mp, kk, tab, st));
                    SAPCU_TRY(run_gemm(m, g, st));
                }
                // g = LIF(fc_gamma(attn_in)) -> B1                                      fn:373-376
                SAPCU_TRY(gemm(m, B3, R, d, d, m->p(sb + B_GAMMA_W), d, m->p(sb + B_GAMMA_B), B1, d, EPI_LIF, st,
                               m->p(sb + B_GAMMA_LIF), 4, nullptr, 0, SP | (SP << 1)));
                {
                    // a = fc_gamma2(g) -> B3                                            fn:378
                    SAPCU_TRY(gemm(m, B1, R, d, d, m->p(sb + B_GAMMA2_W), d, m->p(sb + B_GAMMA2_B), B3, d, EPI_BIAS, st, nullptr, 0,
                                   nullptr, 0, SP));
                    // res = sum_j softmax_j(a / sqrt(hd)) * (v_j + pe)                  fn:379-389
                    SAPCU_TRY(launch_fn_softmax_agg(B3, B2, QKV + 2 * d, 3 * d, idx[l], P, mp, kk, d, sqrt_hd, RES, SP, st));
                }
            }
            // out_proj, fc2 + residual                                              fn:393-394
            SAPCU_TRY(gemm(m, RES, P, d, d, m->p(sb + B_OUT_W), d, m->p(sb + B_OUT_B), X, d, EPI_BIAS, st, nullptr, 0, nullptr,
                           0, SP | (SP << 1)));
            SAPCU_TRY(gemm(m, X, P, d, d, m->p(sb + B_FC2_W), 64, m->p(sb + B_FC2_B), cat + 64 * l, 192, EPI_RESID, st,
                           nullptr, 0, fin, ldin, SP));
            if (taps && taps[SAPCU_FN_TAP_BLOCK1 + l]) {
                SAPCU_CHECK_HIP(hipMemcpy2DAsync((char*)taps[SAPCU_FN_TAP_BLOCK1 + l] + s * mp * 64 * 4, 64 * 4,
                                                 cat + 64 * l, 192 * 4, 64 * 4, (size_t)P, hipMemcpyDeviceToDevice, st));
            }
            fin = cat + 64 * l;
            ldin = 192;
        }
        // conv_final + LIF x T_enc, max over points, fc_out                          fn:465-475
        if (m->sf16 && m->opt_fn_maxfuse) {
            // the max over the patch's points inside the GEMM's epilogue (integer atomicMax on order-preserving keys, as fd's
            // multi_scale_conv): the [P, emb] activation is never written.  Keys live at the head of the unused B1 area.
            GemmArgs g;
            memset(&g, 0, sizeof(g));
            g.a = cat; g.r = P; g.k = 192; g.lda = 192; g.w = m->p(FN_FINAL_W); g.n = m->emb; g.bias = m->p(FN_FINAL_B);
            g.ldc = m->emb; g.epi = EPI_LIF_MAX; g.lif = m->p(FN_FINAL_LIF); g.lif_T = m->T;
            g.max_keys = reinterpret_cast<unsigned*>(B1); g.max_m = mp;
            SAPCU_CHECK_HIP(hipMemsetAsync(g.max_keys, 0, (size_t)cb * m->emb * 4, st));
            SAPCU_TRY(run_gemm(m, g, st));
            SAPCU_TRY(launch_decode_max_keys(g.max_keys, cb * m->emb, pooled, st));
        } else {
            SAPCU_TRY(gemm(m, cat, P, 192, 192, m->p(FN_FINAL_W), m->emb, m->p(FN_FINAL_B), B1, m->emb, EPI_LIF, st,
                           m->p(FN_FINAL_LIF), m->T));
            SAPCU_TRY(launch_rowgroup_max(B1, cb, mp, m->emb, pooled, st));
        }
        SAPCU_TRY(tap_copy(taps, SAPCU_FN_TAP_POOLED, s * m->emb * 4, pooled, cb * m->emb * 4, st));
        SAPCU_TRY(gemm(m, pooled, cb, m->emb, m->emb, m->p(FN_FCOUT_W), 2048, m->p(FN_FCOUT_B), enc, 2048, EPI_BIAS, st));
        SAPCU_TRY(tap_copy(taps, SAPCU_FN_TAP_ENC, s * 2048 * 4, enc, cb * 2048 * 4, st));
        // decoder MLP (Linear+BN+GELU) x3, Linear(256,3), LayerNorm(3), normalize    fn:542-549
        SAPCU_TRY(gemm(m, enc, cb, 2048, 2048, m->p(FN_MLP0_W), 1024, m->p(FN_MLP0_B), h1, 1024, EPI_GELU, st));
        SAPCU_TRY(gemm(m, h1, cb, 1024, 1024, m->p(FN_MLP1_W), 512, m->p(FN_MLP1_B), h2, 512, EPI_GELU, st));
        SAPCU_TRY(gemm(m, h2, cb, 512, 512, m->p(FN_MLP2_W), 256, m->p(FN_MLP2_B), h3, 256, EPI_GELU, st));
        SAPCU_TRY(launch_fn_tail(h3, cb, 256, m->p(FN_HEAD_W), m->p(FN_HEAD_B), m->p(FN_LN_W), m->p(FN_LN_B), logits,
                                 normals + s * 3, st));
        SAPCU_TRY(tap_copy(taps, SAPCU_FN_TAP_LOGITS, s * 3 * 4, logits, cb * 3 * 4, st));
    }
    return SAPCU_OK;
}

// ============================================================================ fd
struct FdPlan {
    int64_t cb;
    int kmax0, kk;
    bool fused;                // the whole encoder in fd_encoder.hip: no per-point intermediates in the workspace
    bool x0path;               // per-stage front + fd_msc_kernel: x0 [P, 960] and ONE spike slab instead of T slabs (round 4)
    bool maxfuse;              // multi_scale_conv's max over points inside the GEMM: the [T*P, emb] aggregate is never written
    int64_t agg_rows(int mp) const { return maxfuse ? 1 : mp; }   // rows of the AGG area per (step, patch): keys only, or the aggregate
};

// does the encoder run as ONE LDS-resident kernel per patch (fd_encoder.hip)?  Same answer for the workspace plan and the forward.
static bool fd_encoder_fused(const sapcu_model* m, int mp) {
    return m->sf16 && m->opt_fd_fused && m->fde_w && m->fde_nprm && fd_encoder_ok(mp, m->nscale, m->emb, m->T);
}

// does the per-stage path hand x0 to fd_msc_kernel (multi_scale_conv with the spikes regenerated on the CU) instead of writing the
// spikes of all T steps for the big-tile GEMM?  Patches of more than 48 points (the reference's default is 100), every T.
static bool fd_x0_path(const sapcu_model* m, int mp) {
    return !fd_encoder_fused(m, mp) && m->sf16 && m->opt_fd_fused && m->opt_fd_x0 && m->opt_fd_maxfuse && m->opt_fd_split && m->fde_w &&
           m->fde_nprm && fd_msc_ok(mp, m->emb, m->T);
}

static FdPlan fd_plan(const sapcu_model* m, int64_t b, int mp) {
    FdPlan pl;
    pl.fused = fd_encoder_fused(m, mp);
    pl.x0path = fd_x0_path(m, mp);
    pl.maxfuse = m->sf16 && m->opt_fd_maxfuse;
    if (pl.fused) {
        // the encoder's intermediates never leave the CU: per patch only pooled [T, emb], the encoding and the decoder's rows
        const int64_t per_patch_f = ((int64_t)(m->T + 1) * m->emb + 256 + 3 * 128 + 3 * 64 + 192 + 64) * 4;
        pl.cb = chunk_patches(m, b, per_patch_f);
        int kmaxf = 1;
        for (int i = 0; i < m->nscale; ++i) kmaxf = kmaxf > m->ks[i] ? kmaxf : m->ks[i];
        pl.kmax0 = imin(kmaxf, mp);
        pl.kk = imin(m->k, mp);
        return pl;
    }
    // fd intermediates: T x 960 spikes (+ the T x emb aggregate per point only when the max is NOT taken inside the GEMM) + 960 +
    // 1024 + block-0 features per point, neighbour tables
    int kmax0 = 1;
    for (int i = 0; i < m->nscale; ++i) kmax0 = kmax0 > m->ks[i] ? kmax0 : m->ks[i];
    const int64_t slabs = pl.x0path ? 2 : m->T;        // x0 path: the step-0 spike slab (split rows) + x0, whatever T
    const int64_t per_patch = (int64_t)mp * ((slabs * 960 + 960 + 1024 + 64 * (m->nscale + 1)) * 4 +
                                            (int64_t)(imin(kmax0, mp) + 3 * imin(m->k, mp)) * 4) +
                              (int64_t)m->T * pl.agg_rows(mp) * m->emb * 4 +
                              ((int64_t)(m->T + 1) * m->emb + 256 + 3 * 128 + 3 * 64 + 192 + 64) * 4;
    pl.cb = chunk_patches(m, b, per_patch);
    int kmax = 1;
    for (int i = 0; i < m->nscale; ++i) kmax = kmax > m->ks[i] ? kmax : m->ks[i];
    pl.kmax0 = imin(kmax, mp);
    pl.kk = imin(m->k, mp);
    return pl;
}

static int64_t fd_ws_bytes(const sapcu_model* m, int64_t b, int mp) {
    const FdPlan pl = fd_plan(m, b, mp);
    const int64_t P = pl.cb * mp;
    int64_t fl = 0;
    auto add = [&](int64_t n, int64_t esz) { fl += ((n * esz) + 255) & ~(int64_t)255; };
    if (!pl.fused) {
        add(P * pl.kmax0, 4); add(3 * P * pl.kk, 4);
        add(P * 64 * m->nscale, 4); add(P * 64, 4);
        add((pl.x0path ? 1 : (int64_t)m->T) * P * 960, 4); add(P * 960, 4); add(P * 1024, 4); add((int64_t)m->T * pl.cb * pl.agg_rows(mp) * m->emb, 4);
        if (pl.x0path) add(P * 960, 4);
    }
    add((int64_t)m->T * pl.cb * m->emb, 4); add(pl.cb * m->emb, 4);
    add(pl.cb * 256, 4);
    for (int i = 0; i < 3; ++i) add(pl.cb * 128, 4);
    for (int i = 0; i < 3; ++i) add(pl.cb * 64, 4);
    add(pl.cb * 192, 4); add(pl.cb * 64, 4);
    return fl + 256;
}

static int fd_forward(const sapcu_model* m, const float* patch, int64_t b, int mp, const int32_t* knn_force,
                      float* dist, void* ws, int64_t ws_bytes, void* const* taps, hipStream_t st) {
    const FdPlan pl = fd_plan(m, b, mp);
    if (ws_bytes < fd_ws_bytes(m, b, mp)) {
        set_error("fd_forward: workspace %lld B < required %lld B", (long long)ws_bytes, (long long)fd_ws_bytes(m, b, mp));
        return SAPCU_ERR_WORKSPACE;
    }
    const int T = m->T, emb = m->emb;
    static const int cin[4] = {0, 64, 128, 256}, cout[4] = {64, 128, 256, 512}, coff[4] = {0, 64, 192, 448};
    for (int64_t s = 0; s < b; s += pl.cb) {
        const int64_t cb = (b - s) < pl.cb ? (b - s) : pl.cb;
        const int64_t P = cb * mp;
        Arena A{(char*)(((uintptr_t)ws + 255) & ~(uintptr_t)255), ws_bytes, 0};
        // per-point intermediates exist only on the per-stage path (the fused encoder keeps them on the CU)
        const int64_t pp = pl.fused ? 0 : 1;
        int32_t* idx0 = A.take<int32_t>(pp * pl.cb * mp * pl.kmax0);
        int32_t* idxb = A.take<int32_t>(pp * 3 * pl.cb * mp * pl.kk);
        float* E0 = A.take<float>(pp * pl.cb * mp * 64 * m->nscale);
        float* FUSED = A.take<float>(pp * pl.cb * mp * 64);
        float* SPK = A.take<float>(pp * (pl.x0path ? 1 : (int64_t)T) * pl.cb * mp * 960);
        float* F0 = A.take<float>(pp * pl.cb * mp * 960);
        float* AB = A.take<float>(pp * pl.cb * mp * 1024);
        float* AGG = A.take<float>(pp * (int64_t)T * pl.cb * pl.agg_rows(mp) * emb);       // maxfuse: T*cb*emb keys only
        float* X0 = A.take<float>(pl.x0path ? pl.cb * mp * 960 : 0);                       // x0 path: [P, 960] pre-activations
        float* POOLED = A.take<float>((int64_t)T * pl.cb * emb);
        float* ENC = A.take<float>(pl.cb * emb);
        float* D1 = A.take<float>(pl.cb * 256);
        float* D2a = A.take<float>(pl.cb * 128);
        float* D2b = A.take<float>(pl.cb * 128);
        float* D2c = A.take<float>(pl.cb * 128);
        float* D3a = A.take<float>(pl.cb * 64);
        float* D3b = A.take<float>(pl.cb * 64);
        float* D3c = A.take<float>(pl.cb * 64);
        float* QKV = A.take<float>(pl.cb * 192);
        float* ATT = A.take<float>(pl.cb * 64);
        const float* pc = patch + s * mp * 3;

        if (pl.fused) {
            // the whole encoder up to pooled [T, cb, emb] in ONE launch, one workgroup per patch (fd_encoder.hip)   fd:408-480
            FdEncArgs ea;
            memset(&ea, 0, sizeof(ea));
            ea.patch = pc; ea.b = cb; ea.b_total = b; ea.s0 = s;
            ea.m = mp; ea.T = T; ea.kk = pl.kk; ea.kmax0 = pl.kmax0; ea.nscale = m->nscale; ea.emb = emb;
            for (int i = 0; i < 4; ++i) ea.ks[i] = i < m->nscale ? imin(m->ks[i], mp) : 0;
            ea.e0_w = m->p(FD_E0_W); ea.e0_b = m->p(FD_E0_B);
            const _Float16* fw = (const _Float16*)m->fde_w;
            ea.fuse_wp = fw + m->fde_off[0]; ea.fuse_b = m->p(FD_FUSE_B);
            for (int l = 0; l < 3; ++l) {
                ea.edge_wp[l] = fw + m->fde_off[1 + l];
                ea.shift[l] = m->p(FD_EDGE1_SHIFT + 3 * l);
            }
            ea.msc_wp = fw + m->fde_off[4]; ea.msc_b = m->p(FD_MSC_B);
            ea.nprm = m->fde_nprm;
            ea.pooled = POOLED;
            ea.knn_force = knn_force;
            ea.tap_knn = taps ? (int32_t*)taps[SAPCU_FD_TAP_KNN] : nullptr;
            ea.tap_fused0 = taps ? (float*)taps[SAPCU_FD_TAP_FUSED0] : nullptr;
            ea.tap_spikes = taps ? (float*)taps[SAPCU_FD_TAP_SPIKES] : nullptr;
            ea.tap_x0 = taps ? (float*)taps[SAPCU_FD_TAP_X0] : nullptr;
            ea.gate = m->gate_dev; ea.ovf = m->ovf_dev;
            SAPCU_TRY(launch_fd_encoder(ea, st));
        } else {
        // block 0: xyz kNN (all scales are prefixes of the sorted top-kmax list), EdgeConv x S,
        // scale fusion, EIF over T steps                                           fd:411-444
        SAPCU_TRY(launch_patch_knn(pc, cb, mp, 3, 3, pl.kmax0, idx0, st));
        SAPCU_TRY(launch_fd_edge0(pc, idx0, pl.kmax0, P, mp, m->nscale, m->ks_dev, m->p(FD_E0_W), m->p(FD_E0_B), E0, st));
        SAPCU_TRY(gemm(m, E0, P, 64 * m->nscale, 64 * m->nscale, m->p(FD_FUSE_W), 64, m->p(FD_FUSE_B), FUSED, 64,
                       EPI_LRELU, st));
        SAPCU_TRY(tap_copy(taps, SAPCU_FD_TAP_FUSED0, s * mp * 64 * 4, FUSED, P * 64 * 4, st));
        float* const X0T = (taps && taps[SAPCU_FD_TAP_X0]) ? (float*)taps[SAPCU_FD_TAP_X0] + s * mp * 960 : nullptr;
        if (X0T) SAPCU_CHECK_HIP(hipMemcpy2DAsync(X0T, 960 * 4, FUSED, 64 * 4, 64 * 4, (size_t)P, hipMemcpyDeviceToDevice, st));
        // multi_scale_conv's operand: split rows written by the neuron kernels themselves (the GEMM then streams them by LDS-DMA
        // on the big-tile kernel and takes the max over the points in its epilogue) whenever that kernel takes the shape; the
        // step-0 spikes also go to F0 as f32 for the next blocks' neighbour search and EdgeConv.  Otherwise (tiny batches,
        // SAPCU_GEMM=f32, SAPCU_FD_MAXFUSE=0 / SAPCU_FD_SPLIT=0, or a caller asking for the spike tap): f32 spikes for all steps.
        GemmArgs mg;
        memset(&mg, 0, sizeof(mg));
        mg.a = SPK; mg.r = (int64_t)T * P; mg.k = 960; mg.lda = 960; mg.w = m->p(FD_MSC_W); mg.n = emb; mg.bias = m->p(FD_MSC_B);
        mg.c = nullptr; mg.ldc = emb; mg.epi = EPI_LRELU_MAX; mg.max_keys = reinterpret_cast<unsigned*>(AGG); mg.max_m = mp;
        const bool maxfuse = pl.maxfuse;
        bool split_spikes = pl.x0path;                          // x0 path: ONE slab of step-0 split rows (the EdgeConv GEMMs' operand)
        if (!pl.x0path && maxfuse && m->opt_fd_split && !(taps && taps[SAPCU_FD_TAP_SPIKES])) {
            GemmArgs probe = mg;
            probe.a_split = 1;
            probe.w16_hi = (const _Float16*)m->w16_hi + (mg.w - m->blob);
            probe.w16_lo = (const _Float16*)m->w16_lo + (mg.w - m->blob);
            split_spikes = split_rows_gemm_on_big_tile(probe, m->opt_bt);     // (the ring kernel has no max-over-rows epilogue)
        }
        float* const SPKS = split_spikes ? SPK : nullptr;       // [T*P, 960] split rows (same buffer, other format)
        float* const SPK0 = split_spikes ? F0 : SPK;            // where the step-0 f32 spikes live ([P, 960] slab)
        float* const X0P = pl.x0path ? X0 : nullptr;            // the neuron kernels then write x0 and run step 0 only
        SAPCU_TRY(launch_fd_neuron(true, 0, FUSED, 64, nullptr, 0, mp, nullptr, P, 64, m->p(FD_SNN0), T, SPK0, 960, 0,
                                   nullptr, m->gate_dev, st, SPKS, X0P));
        // blocks 1..3: feature-space kNN on the t=0 spikes, factored EdgeConv, neuron  fd:447-474
        for (int l = 1; l <= 3; ++l) {
            int32_t* idl = idxb + (int64_t)(l - 1) * pl.cb * mp * pl.kk;
            const float* F = SPK0 + coff[l - 1];   // t = 0 slab, row stride 960
            if (knn_force) {
                SAPCU_CHECK_HIP(hipMemcpyAsync(idl, knn_force + ((int64_t)(l - 1) * b + s) * mp * pl.kk, P * pl.kk * 4,
                                               hipMemcpyDeviceToDevice, st));
            } else {
                SAPCU_TRY(launch_patch_knn_strided(F, cb, (int64_t)mp * 960, mp, cin[l], 960, pl.kk, idl, st));
            }
            if (taps && taps[SAPCU_FD_TAP_KNN])
                SAPCU_CHECK_HIP(hipMemcpyAsync((int32_t*)taps[SAPCU_FD_TAP_KNN] + ((int64_t)(l - 1) * b + s) * mp * pl.kk,
                                               idl, P * pl.kk * 4, hipMemcpyDeviceToDevice, st));
            const int ew = FD_EDGE1_W + 3 * (l - 1);
            // the factored EdgeConv's GEMM reads the step-0 spikes as split rows when the neuron kernels wrote them (rows 0..P-1 of
            // SPKS, same pitch and column offsets as the f32 slab): the all-DMA kernels instead of the f32-operand one, same sums
            if (split_spikes)
                SAPCU_TRY(gemm(m, SPKS + coff[l - 1], P, cin[l], 960, m->p(ew), 2 * cout[l], nullptr, AB, 2 * cout[l], EPI_BIAS, st,
                               nullptr, 0, nullptr, 0, 1));
            else
                SAPCU_TRY(gemm(m, F, P, cin[l], 960, m->p(ew), 2 * cout[l], nullptr, AB, 2 * cout[l], EPI_BIAS, st));
            SAPCU_TRY(launch_fd_neuron(l == 1, 1, AB, 2 * cout[l], idl, pl.kk, mp, m->p(ew + 1), P, cout[l], m->p(ew + 2),
                                       T, SPK0, 960, coff[l], nullptr, m->gate_dev, st, SPKS, X0P));
            if (X0T && !pl.x0path) SAPCU_TRY(launch_fd_pre(AB, 2 * cout[l], idl, pl.kk, mp, m->p(ew + 1), P, cout[l], X0T, 960, coff[l], st));
        }
        if (pl.x0path) {
            // multi_scale_conv over all T steps + max over the points straight from x0: every spike regenerated on the CU (fd_msc_kernel)
            if (X0T) SAPCU_CHECK_HIP(hipMemcpyAsync(X0T, X0, (size_t)P * 960 * 4, hipMemcpyDeviceToDevice, st));
            FdMscArgs ma;
            memset(&ma, 0, sizeof(ma));
            ma.x0 = X0; ma.b = cb; ma.b_total = b; ma.s0 = s; ma.m = mp; ma.T = T; ma.emb = emb;
            ma.msc_wp = (const _Float16*)m->fde_w + m->fde_off[4]; ma.msc_b = m->p(FD_MSC_B); ma.nprm = m->fde_nprm;
            ma.pooled = POOLED; ma.tap_spikes = taps ? (float*)taps[SAPCU_FD_TAP_SPIKES] : nullptr; ma.gate = m->gate_dev;
            SAPCU_TRY(launch_fd_msc(ma, st));
        } else {
        if (taps && taps[SAPCU_FD_TAP_SPIKES]) {
            for (int t = 0; t < T; ++t)
                SAPCU_CHECK_HIP(hipMemcpyAsync((float*)taps[SAPCU_FD_TAP_SPIKES] + ((int64_t)t * b + s) * mp * 960,
                                               SPK + (int64_t)t * P * 960, P * 960 * 4, hipMemcpyDeviceToDevice, st));
        }
        // multi_scale_conv + BN + LeakyReLU over all T*P rows, max over points        fd:476-480
        if (maxfuse) {
            // the max over the patch's points inside the GEMM's epilogue (integer atomicMax on order-preserving keys): the
            // [T*P, emb] aggregate is never written.  The key buffer is the head of the (otherwise unused) AGG area.
            SAPCU_CHECK_HIP(hipMemsetAsync(mg.max_keys, 0, (size_t)T * cb * emb * 4, st));
            mg.a_split = split_spikes ? 1 : 0;
            SAPCU_TRY(run_gemm(m, mg, st));
            SAPCU_TRY(launch_decode_max_keys(mg.max_keys, (int64_t)T * cb * emb, POOLED, st));
        } else {
            SAPCU_TRY(gemm(m, SPK, (int64_t)T * P, 960, 960, m->p(FD_MSC_W), emb, m->p(FD_MSC_B), AGG, emb, EPI_LRELU, st));
            SAPCU_TRY(launch_rowgroup_max(AGG, (int64_t)T * cb, mp, emb, POOLED, st));
        }
        }   // T spike slabs + GEMM
        }   // per-stage encoder
        if (taps && taps[SAPCU_FD_TAP_POOLED]) {
            for (int t = 0; t < T; ++t)
                SAPCU_CHECK_HIP(hipMemcpyAsync((float*)taps[SAPCU_FD_TAP_POOLED] + ((int64_t)t * b + s) * emb,
                                               POOLED + (int64_t)t * cb * emb, cb * emb * 4, hipMemcpyDeviceToDevice, st));
        }
        SAPCU_TRY(launch_fd_temporal(POOLED, T, cb, emb, m->p(FD_TI_W), m->p(FD_SNNFC), ENC, st));
        SAPCU_TRY(tap_copy(taps, SAPCU_FD_TAP_ENC, s * emb * 4, ENC, cb * emb * 4, st));
        // decoder                                                                    fd:711-725
        SAPCU_TRY(gemm(m, ENC, cb, emb, emb, m->p(FD_FCIN_W), 256, m->p(FD_FCIN_B), D1, 256, EPI_GELU, st));
        SAPCU_TRY(gemm(m, D1, cb, 256, 256, m->p(FD_R0_FC0_W), 128, m->p(FD_R0_FC0_B), D2a, 128, EPI_GELU, st));
        SAPCU_TRY(gemm(m, D1, cb, 256, 256, m->p(FD_R0_PROJ_W), 128, m->p(FD_R0_PROJ_B), D2b, 128, EPI_BIAS, st));
        SAPCU_TRY(gemm(m, D2a, cb, 128, 128, m->p(FD_R0_FC4_W), 128, m->p(FD_R0_FC4_B), D2c, 128, EPI_RESID_GELU, st,
                       nullptr, 0, D2b, 128));
        SAPCU_TRY(gemm(m, D2c, cb, 128, 128, m->p(FD_R1_FC0_W), 64, m->p(FD_R1_FC0_B), D3a, 64, EPI_GELU, st));
        SAPCU_TRY(gemm(m, D2c, cb, 128, 128, m->p(FD_R1_PROJ_W), 64, m->p(FD_R1_PROJ_B), D3b, 64, EPI_BIAS, st));
        SAPCU_TRY(gemm(m, D3a, cb, 64, 64, m->p(FD_R1_FC4_W), 64, m->p(FD_R1_FC4_B), D3c, 64, EPI_RESID_GELU, st, nullptr,
                       0, D3b, 64));
        SAPCU_TRY(gemm(m, D3c, cb, 64, 64, m->p(FD_QKV_W), 192, m->p(FD_QKV_B), QKV, 192, EPI_BIAS, st));
        SAPCU_TRY(launch_fd_tail(D3c, QKV, cb, m->heads, m->p(FD_WO_T), m->p(FD_BO), m->p(FD_LN_W), m->p(FD_LN_B),
                                 m->p(FD_WH_T), m->p(FD_BH), m->p(FD_WD), m->p(FD_BD), ATT, dist + s, st));
    }
    return SAPCU_OK;
}

}  // namespace sapcu

// ================================================================================== C ABI
using namespace sapcu;

extern "C" {

int sapcu_abi_version(void) { return SAPCU_ABI_VERSION; }
const char* sapcu_last_error(void) { return g_err; }

int sapcu_knn_gather_f64(const double* cloud, int64_t n, const double* queries, int64_t b, int k, int64_t* idx_out,
                         double* dist_out, float* patch_out, void* stream) {
    SAPCU_CHECK_ARG(cloud && queries && idx_out, "knn_gather: null pointer");
    SAPCU_CHECK_ARG(n >= 1 && b >= 0 && k >= 1 && k <= 128 && k <= n, "knn_gather: need 1 <= k <= min(128, n) (n=%lld k=%d)",
                    (long long)n, k);
    SAPCU_CHECK_ARG(n < 0x7fffffffLL, "knn_gather: n too large");
    return launch_knn_outer(cloud, n, queries, b, k, idx_out, dist_out, patch_out, (hipStream_t)stream);
}

int sapcu_gather_rotate_f64(const double* cloud, int64_t n, const double* queries, int64_t b, const int64_t* idx, int k,
                            const float* normals, float* patch_out, void* stream) {
    SAPCU_CHECK_ARG(cloud && queries && idx && patch_out, "gather_rotate: null pointer");
    SAPCU_CHECK_ARG(n >= 1 && b >= 0 && k >= 1, "gather_rotate: bad sizes");
    return launch_gather_rotate(cloud, n, queries, b, idx, k, normals, patch_out, (hipStream_t)stream);
}

int sapcu_displace_f64(const double* queries, const float* normals, const float* dist, int64_t b, double* out,
                       void* stream) {
    SAPCU_CHECK_ARG(queries && normals && dist && out && b >= 0, "displace: bad argument");
    return launch_displace(queries, normals, dist, b, out, (hipStream_t)stream);
}

int64_t sapcu_fps_workspace_bytes(int64_t npoint) { return npoint < 0 ? -1 : (int64_t)fps_workspace_bytes((int)npoint); }

int sapcu_fps_f32(const float* xyz, int64_t n, int64_t npoint, int64_t* idx_out, void* workspace, int64_t workspace_bytes,
                  void* stream) {
    SAPCU_CHECK_ARG(npoint >= 0 && npoint <= n && n < (int64_t)1 << 31, "fps: need 0 <= npoint <= n < 2^31 (n=%lld npoint=%lld)",
                    (long long)n, (long long)npoint);
    if (npoint == 0) return SAPCU_OK;
    SAPCU_CHECK_ARG(xyz && idx_out && workspace, "fps: null pointer");
    if (workspace_bytes < (int64_t)fps_workspace_bytes((int)npoint)) {
        set_error("fps: workspace of %lld bytes, need %lld", (long long)workspace_bytes,
                  (long long)fps_workspace_bytes((int)npoint));
        return SAPCU_ERR_WORKSPACE;
    }
    int rc = launch_fps(xyz, n, (int)npoint, (int)(n / 2), idx_out, workspace, (hipStream_t)stream);
    if (rc != SAPCU_OK) return rc;
    SAPCU_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
    int flag = 0;
    rc = fps_failed(workspace, (int)npoint, &flag);
    if (rc != SAPCU_OK) return rc;
    if (flag) {
        set_error("fps: a workgroup of the persistent grid never arrived at the step barrier (grid not resident)");
        return SAPCU_ERR_HIP;
    }
    return SAPCU_OK;
}

int sapcu_neuron_selfloop(const float* x, int64_t rows, int channels, int steps, const float* membrane_decay,
                          const float* threshold_adapt, const float* refractory_decay, const float* threshold_base,
                          const float* delta_T, const float* theta_rh, float* spikes_out, float* membrane_out,
                          float* threshold_out, float* refractory_out, void* stream) {
    SAPCU_CHECK_ARG(x && membrane_decay && threshold_adapt && refractory_decay && threshold_base, "neuron: null pointer");
    SAPCU_CHECK_ARG((delta_T == nullptr) == (theta_rh == nullptr), "neuron: delta_T and theta_rh go together");
    SAPCU_CHECK_ARG(rows >= 0 && channels >= 1 && steps >= 1, "neuron: bad sizes");
    return launch_neuron_selfloop(x, rows, channels, steps, membrane_decay, threshold_adapt, refractory_decay,
                                  threshold_base, delta_T, theta_rh, spikes_out, membrane_out, threshold_out,
                                  refractory_out, (hipStream_t)stream);
}

int sapcu_neuron_drive(const float* x, int64_t rows, int channels, int steps, const float* membrane_decay,
                       const float* threshold_adapt, const float* refractory_decay, const float* threshold_base,
                       const float* delta_T, const float* theta_rh, int channel_pairs, float* spikes_out, float* membrane_out,
                       float* threshold_out, float* refractory_out, int* gate_open_out, void* stream) {
    SAPCU_CHECK_ARG(x && membrane_decay && threshold_adapt && refractory_decay && threshold_base, "neuron_drive: null pointer");
    SAPCU_CHECK_ARG((delta_T == nullptr) == (theta_rh == nullptr), "neuron_drive: delta_T and theta_rh go together");
    SAPCU_CHECK_ARG(rows >= 0 && channels >= 1 && steps >= 1, "neuron_drive: bad sizes");
    return launch_neuron_drive(x, rows, channels, steps, membrane_decay, threshold_adapt, refractory_decay, threshold_base, delta_T,
                               theta_rh, channel_pairs, spikes_out, membrane_out, threshold_out, refractory_out, gate_open_out,
                               (hipStream_t)stream);
}

int sapcu_patch_knn(const float* feat, int64_t b, int m, int c, int ld, int k, int32_t* idx_out, void* stream) {
    SAPCU_CHECK_ARG(feat && idx_out && b >= 0 && ld >= c, "patch_knn: bad argument");
    return launch_patch_knn(feat, b, m, c, ld, k, idx_out, (hipStream_t)stream);
}

int sapcu_l2_normalize3(const float* in, float* out, int64_t b, void* stream) {
    SAPCU_CHECK_ARG(in && out && b >= 0, "l2_normalize3: bad argument");
    return launch_l2_normalize3(in, out, b, (hipStream_t)stream);
}

static int split_into_ws(const float* w, int64_t cnt, void* w16_ws, GemmArgs& g, hipStream_t st) {
    char* base = (char*)w16_ws;            // hi (2*cnt B) | lo (2*cnt B) | overflow counter
    int* ovf = (int*)(base + 4 * cnt);
    SAPCU_CHECK_HIP(hipMemsetAsync(ovf, 0, sizeof(int), st));
    SAPCU_TRY(launch_split_weights(w, cnt, base, base + 2 * cnt, ovf, st));
    g.w16_hi = (const _Float16*)base;
    g.w16_lo = (const _Float16*)(base + 2 * cnt);
    g.ovf = ovf;
    return SAPCU_OK;
}

int sapcu_gemm_f32(const float* a, int64_t r, int k, int lda, const float* w, int n, const float* bias,
                   const float* lif4, int lif_steps, float* c, int ldc, void* w16_ws, int a_split_rows, int c_split_rows,
                   void* stream) {
    SAPCU_CHECK_ARG(a && w && c && r >= 0 && n >= 1, "gemm: bad argument");
    SAPCU_CHECK_ARG(!lif4 || lif_steps >= 1, "gemm: lif_steps must be >= 1");
    SAPCU_CHECK_ARG(!(a_split_rows || c_split_rows) || w16_ws, "gemm: split rows need the split-f16 kernels (w16_ws)");
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.a = a; g.r = r; g.k = k; g.lda = lda; g.w = w; g.n = n; g.bias = bias; g.c = c; g.ldc = ldc;
    g.epi = lif4 ? EPI_LIF : EPI_BIAS; g.lif = lif4; g.lif_T = lif_steps;
    g.a_split = a_split_rows ? 1 : 0; g.c_split = c_split_rows ? 1 : 0;
    if (w16_ws && (g.a_split || k % 64 == 0)) {   // f32-A split-f16 kernel steps k by 64; other depths run on exact f32
        SAPCU_TRY(split_into_ws(w, (int64_t)n * k, w16_ws, g, (hipStream_t)stream));
        // a_split_rows = 2: the ring kernel even where the big-tile kernel takes the shape (bit-identical; parity tests)
        return g.a_split ? launch_gemm_split_rows(g, (hipStream_t)stream, a_split_rows != 2) : launch_gemm_sf16(g, (hipStream_t)stream);
    }
    SAPCU_CHECK_ARG(!g.c_split, "gemm: split-row output needs k %% 64 == 0 on the f32-A path");
    return launch_gemm(g, (hipStream_t)stream);
}

int sapcu_to_split_rows(const float* in, int64_t rows, int k, int ld_in, float* out, int ld_out, void* stream) {
    SAPCU_CHECK_ARG(in && out && rows >= 0 && k >= 1 && ld_in >= k && ld_out >= k, "to_split_rows: bad argument");
    return launch_to_split_rows(in, rows, k, ld_in, out, ld_out, (hipStream_t)stream);
}

int sapcu_posenc_gemm_f32(const float* pe1, int64_t r, int d, const float* w, const float* bias, const float* lif4,
                          int lif_steps, const float* qkv, const int32_t* idx, int kk, int m_pts, float* pe_out,
                          float* attn_in_out, void* edge_table_ws, void* w16_ws, int split_rows, void* stream) {
    SAPCU_CHECK_ARG(pe1 && w && lif4 && qkv && idx && pe_out && attn_in_out && edge_table_ws && r >= 0 && d >= 32 &&
                        lif_steps >= 1 && kk >= 1 && m_pts >= 1,
                    "posenc_gemm: bad argument");
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.a = pe1; g.r = r; g.k = d; g.lda = d; g.w = w; g.n = d; g.bias = bias; g.c = pe_out; g.ldc = d;
    g.epi = EPI_LIF_ATTN; g.lif = lif4; g.lif_T = lif_steps; g.c2 = attn_in_out;
    g.q = qkv; g.kf = qkv + d; g.ldq = 3 * d; g.tab = (const int2*)edge_table_ws;
    SAPCU_TRY(launch_edge_table(idx, r, m_pts, kk, (int2*)edge_table_ws, (hipStream_t)stream));
    SAPCU_CHECK_ARG(!split_rows || w16_ws, "posenc_gemm: split rows need the split-f16 kernels (w16_ws)");
    if (w16_ws && (split_rows || d % 64 == 0)) {   // split W into the caller's scratch (hi | lo | counter), then 3 x f16 MFMA
        SAPCU_TRY(split_into_ws(w, (int64_t)d * d, w16_ws, g, (hipStream_t)stream));
        if (split_rows) {     // the production form: pe1 arrives as split rows, attn_in leaves as split rows
            g.a_split = 1;
            g.c2_split = 1;
            return launch_gemm_split_rows(g, (hipStream_t)stream, split_rows != 2);     // split_rows = 2: ring kernel only
        }
        return launch_gemm_sf16(g, (hipStream_t)stream);
    }
    return launch_gemm(g, (hipStream_t)stream);
}

int64_t sapcu_fn_edge_chain_workspace_bytes(int64_t points, int d, int kk) {
    if (points < 0 || !fn_edge_chain_ok(d, kk)) {
        set_error("fn_edge_chain_workspace_bytes: unsupported shape (d=%d kk=%d)", d, kk);
        return SAPCU_ERR_ARG;
    }
    const int64_t rows = points * kk;
    auto up = [](int64_t b) { return (b + 255) & ~(int64_t)255; };
    return up(rows * 8) + up(rows * 16) + 3 * (up((int64_t)d * d * 4 + 16) + up((int64_t)d * d * 4)) + 256;
}

int sapcu_fn_edge_chain_f32(const float* patch, const int32_t* idx, int64_t points, int m_pts, int d, int kk,
                            const float* qkv, const float* w_delta, const float* b_delta, const float* lif_delta,
                            const float* w1, const float* b1, const float* lif1, const float* w2, const float* b2,
                            const float* lif2, const float* w3, const float* b3, int heads, int lif_steps,
                            float* res_out, void* workspace, int64_t workspace_bytes, void* stream) {
    SAPCU_CHECK_ARG(patch && idx && qkv && w_delta && b_delta && lif_delta && w1 && b1 && lif1 && w2 && b2 && lif2 && w3 && b3 &&
                        res_out && workspace, "fn_edge_chain: null pointer");
    SAPCU_CHECK_ARG(fn_edge_chain_ok(d, kk), "fn_edge_chain: unsupported shape (d=%d kk=%d)", d, kk);
    SAPCU_CHECK_ARG(points >= 0 && m_pts >= kk && heads >= 1 && d % heads == 0 && lif_steps >= 1, "fn_edge_chain: bad sizes");
    const int64_t need = sapcu_fn_edge_chain_workspace_bytes(points, d, kk);
    if (workspace_bytes < need) {
        set_error("fn_edge_chain: workspace %lld B < required %lld B", (long long)workspace_bytes, (long long)need);
        return SAPCU_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    Arena A{(char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255), workspace_bytes, 0};
    int2* tab = A.take<int2>(points * kk);
    float4* pd = A.take<float4>(points * kk);
    const float* ws[3] = {w1, w2, w3};
    const _Float16* packed[3];
    for (int q = 0; q < 3; ++q) {
        char* split = A.take<char>((int64_t)d * d * 4 + 16);              // hi | lo | overflow counter
        _Float16* pk = A.take<_Float16>((int64_t)d * d * 2);
        GemmArgs g;
        memset(&g, 0, sizeof(g));
        SAPCU_TRY(split_into_ws(ws[q], (int64_t)d * d, split, g, st));
        SAPCU_TRY(launch_pack_chain_weights(g.w16_hi, g.w16_lo, d, pk, st));
        packed[q] = pk;
    }
    ChainArgs ca;
    memset(&ca, 0, sizeof(ca));
    ca.P = points; ca.m = m_pts; ca.qkv = qkv; ca.ldq = 3 * d;
    ca.wd = w_delta; ca.bd = b_delta; ca.lifd = lif_delta;
    ca.w1p = packed[0]; ca.b1 = b1; ca.lif1 = lif1;
    ca.w2p = packed[1]; ca.b2 = b2; ca.lif2 = lif2;
    ca.w3p = packed[2]; ca.b3 = b3;
    ca.inv_sqrt_hd = 1.0f / (float)sqrt((double)(d / heads));
    ca.res = res_out; ca.res_split = 0; ca.T = lif_steps;
    return launch_fn_edge_chain(ca, patch, idx, d, kk, tab, pd, st);
}

int sapcu_model_create(int kind, const int32_t* hp, int n_hp, const float* blob, int64_t blob_floats,
                       const int64_t* dir_host, int n_dir, sapcu_model_t* out) {
    SAPCU_CHECK_ARG(hp && blob && dir_host && out && blob_floats > 0, "model_create: null pointer");
    sapcu_model* m = new (std::nothrow) sapcu_model();
    SAPCU_CHECK_ARG(m != nullptr, "model_create: out of host memory");
    m->kind = kind;
    m->ks_dev = nullptr;
    m->gate_dev = nullptr;
    m->blob = nullptr;
    m->w16_hi = nullptr;
    m->w16_lo = nullptr;
    m->chain_w = nullptr;
    m->ovf_dev = nullptr;
    m->fde_w = nullptr;
    m->fde_nprm = nullptr;
    const char* ge = getenv("SAPCU_GEMM");
    m->sf16 = !(ge && strcmp(ge, "f32") == 0);
    m->opt_bt = !env_off("SAPCU_BT");
    m->opt_chain = !env_off("SAPCU_CHAIN");
    {
        const char* v = getenv("SAPCU_CHAIN");
        m->opt_chain_wide = v && strcmp(v, "wide") == 0;
    }
    m->opt_fn_maxfuse = !env_off("SAPCU_FN_MAXFUSE");
    m->opt_fd_maxfuse = !env_off("SAPCU_FD_MAXFUSE");
    m->opt_fd_split = !env_off("SAPCU_FD_SPLIT");
    m->opt_fd_fused = !env_off("SAPCU_FD_FUSED");
    m->opt_fd_x0 = !env_off("SAPCU_FD_X0");
    const char* ce = getenv("SAPCU_CHUNK");
    m->chunk = ce ? atoll(ce) : 0;
    if (m->chunk < 0) m->chunk = 0;
    // workspace budget per forward (the caller owns the buffer; sapcu_workspace_bytes reports what a batch needs under it):
    // 20 GiB holds the whole 4096-patch benchmark batch at M = 48 in one chunk (16.9 GB fn, 6.7 GB fd) and cuts the reference's
    // default M = 100 (8.5 MB per patch) into chunks of ~2400 patches instead of a 35 GB workspace
    const char* be = getenv("SAPCU_WS_BUDGET_MB");
    m->ws_budget = (be ? atoll(be) : 20480) * (int64_t)(1 << 20);
    if (m->ws_budget < (int64_t)(64 << 20)) m->ws_budget = (int64_t)(64 << 20);
    int rc = SAPCU_OK;
    if (kind == SAPCU_KIND_FN) {
        if (n_hp != 6 || n_dir != FN_SLOTS) {
            set_error("model_create(fn): need 6 hparams and %d slots (got %d, %d)", (int)FN_SLOTS, n_hp, n_dir);
            rc = SAPCU_ERR_ARG;
        } else {
            m->kv[0] = hp[0]; m->kv[1] = hp[1]; m->kv[2] = hp[2];
            m->emb = hp[3]; m->T = hp[4]; m->heads = hp[5];
            if (m->kv[0] < 1 || m->kv[1] < 1 || m->kv[2] < 1 || m->emb < 32 || m->emb % 32 || m->T < 1 || m->heads < 1 ||
                128 % m->heads) {
                set_error("model_create(fn): unsupported hyper-parameters");
                rc = SAPCU_ERR_ARG;
            }
        }
    } else if (kind == SAPCU_KIND_FD) {
        if (n_hp < 6 || n_dir != FD_SLOTS || hp[4] < 1 || hp[4] > 8 || n_hp != 5 + hp[4]) {
            set_error("model_create(fd): need [k,emb,T,heads,S,ks...] and %d slots", (int)FD_SLOTS);
            rc = SAPCU_ERR_ARG;
        } else {
            m->k = hp[0]; m->emb = hp[1]; m->T = hp[2]; m->heads = hp[3]; m->nscale = hp[4];
            for (int i = 0; i < m->nscale; ++i) m->ks[i] = hp[5 + i];
            if (m->k < 1 || m->emb < 32 || m->emb % 32 || m->T < 1 || m->T > 64 || m->heads < 1) {
                set_error("model_create(fd): unsupported hyper-parameters");
                rc = SAPCU_ERR_ARG;
            }
        }
    } else {
        set_error("model_create: unknown kind %d", kind);
        rc = SAPCU_ERR_ARG;
    }
    if (rc == SAPCU_OK) {
        m->dir.assign(dir_host, dir_host + n_dir);
        for (int i = 0; i < n_dir; ++i)
            if (m->dir[i] < 0 || m->dir[i] >= blob_floats || (m->dir[i] & 3)) {
                set_error("model_create: slot %d offset %lld out of range / unaligned", i, (long long)m->dir[i]);
                rc = SAPCU_ERR_ARG;
                break;
            }
    }
    auto hip_ok = [&](hipError_t e, const char* what) {
        if (e != hipSuccess && rc == SAPCU_OK) {
            set_error("model_create: %s failed: %s", what, hipGetErrorString(e));
            rc = SAPCU_ERR_HIP;
        }
    };
    if (rc == SAPCU_OK) {
        m->blob_floats = blob_floats;
        hip_ok(hipMalloc((void**)&m->blob, (size_t)blob_floats * 4), "hipMalloc(blob)");
        if (rc == SAPCU_OK) hip_ok(hipMemcpy(m->blob, blob, (size_t)blob_floats * 4, hipMemcpyDeviceToDevice), "copy blob");
        if (rc == SAPCU_OK) hip_ok(hipMalloc((void**)&m->ovf_dev, 2 * sizeof(int)), "hipMalloc(ovf)");
        if (rc == SAPCU_OK) hip_ok(hipMemset(m->ovf_dev, 0, 2 * sizeof(int)), "memset ovf");
        if (rc == SAPCU_OK && m->sf16) {
            hip_ok(hipMalloc(&m->w16_hi, (size_t)blob_floats * 2), "hipMalloc(w16_hi)");
            if (rc == SAPCU_OK) hip_ok(hipMalloc(&m->w16_lo, (size_t)blob_floats * 2), "hipMalloc(w16_lo)");
            if (rc == SAPCU_OK) {
                int wovf = 0;
                if (launch_split_weights(m->blob, blob_floats, m->w16_hi, m->w16_lo, m->ovf_dev + 1, nullptr) != SAPCU_OK)
                    rc = SAPCU_ERR_HIP;
                if (rc == SAPCU_OK) hip_ok(hipMemcpy(&wovf, m->ovf_dev + 1, sizeof(int), hipMemcpyDeviceToHost), "read ovf");
                if (rc == SAPCU_OK && wovf != 0) m->sf16 = false;   // a parameter exceeds the f16 range: exact-f32 kernels
            }
            // fn blocks 1-3 (d = 128, 256, 512): the three d x d matrices of the edge chain again in MFMA-fragment order
            if (rc == SAPCU_OK && m->sf16 && kind == SAPCU_KIND_FN) {
                const int64_t halves = (int64_t)3 * 2 * (128 * 128 + 256 * 256 + 512 * 512);
                hip_ok(hipMalloc(&m->chain_w, (size_t)halves * 2), "hipMalloc(chain_w)");
                int64_t off = 0;
                for (int l = 0; l < 3 && rc == SAPCU_OK; ++l) {
                    const int d = 128 << l;
                    static const int slots[3] = {B_DELTA2_W, B_GAMMA_W, B_GAMMA2_W};
                    for (int q = 0; q < 3 && rc == SAPCU_OK; ++q) {
                        const int64_t wo = m->dir[FN_BLK0 + l * B_SLOTS + slots[q]];
                        if (launch_pack_chain_weights((const _Float16*)m->w16_hi + wo, (const _Float16*)m->w16_lo + wo, d,
                                                      (_Float16*)m->chain_w + off, nullptr) != SAPCU_OK)
                            rc = SAPCU_ERR_HIP;
                        off += (int64_t)d * d * 2;
                    }
                }
                if (rc == SAPCU_OK) hip_ok(hipDeviceSynchronize(), "pack chain weights");
            }
        }
        if (rc == SAPCU_OK && kind == SAPCU_KIND_FD) {
            hip_ok(hipMalloc((void**)&m->ks_dev, 8 * sizeof(int32_t)), "hipMalloc(ks)");
            if (rc == SAPCU_OK) hip_ok(hipMemcpy(m->ks_dev, m->ks, 8 * sizeof(int32_t), hipMemcpyHostToDevice), "copy ks");
            if (rc == SAPCU_OK) hip_ok(hipMalloc((void**)&m->gate_dev, sizeof(int)), "hipMalloc(gate)");
            if (rc == SAPCU_OK) hip_ok(hipMemset(m->gate_dev, 0, sizeof(int)), "memset gate");
            // the fused encoder's operands: five matrices in fragment order (hi | lo planes interleaved per fragment), neuron
            // parameters clamped once
            if (rc == SAPCU_OK && m->sf16 && m->nscale <= 4 && m->emb >= 96) {
                const int mats[5][3] = {{FD_FUSE_W, 64, 64 * m->nscale}, {FD_EDGE1_W, 256, 64}, {FD_EDGE2_W, 512, 128},
                                        {FD_EDGE3_W, 1024, 256}, {FD_MSC_W, m->emb, 960}};
                int64_t halves = 0;
                for (int i = 0; i < 5; ++i) {
                    m->fde_off[i] = halves;
                    halves += (int64_t)2 * mats[i][1] * mats[i][2];
                }
                hip_ok(hipMalloc(&m->fde_w, (size_t)halves * 2), "hipMalloc(fde_w)");
                if (rc == SAPCU_OK) hip_ok(hipMalloc((void**)&m->fde_nprm, 960 * 8 * sizeof(float)), "hipMalloc(fde_nprm)");
                for (int i = 0; i < 5 && rc == SAPCU_OK; ++i) {
                    const int64_t wo = m->dir[mats[i][0]];
                    if (launch_pack_frag_weights((const _Float16*)m->w16_hi + wo, (const _Float16*)m->w16_lo + wo, mats[i][1], mats[i][2],
                                                 (_Float16*)m->fde_w + m->fde_off[i], nullptr) != SAPCU_OK)
                        rc = SAPCU_ERR_HIP;
                }
                static const int nslot[4] = {FD_SNN0, FD_SNN1, FD_SNN2, FD_SNN3}, nch[4] = {64, 128, 256, 512}, noff[4] = {0, 64, 192, 448};
                for (int i = 0; i < 4 && rc == SAPCU_OK; ++i)
                    if (launch_pack_fd_neuron(m->p(nslot[i]), nch[i], i < 2 ? 1 : 0, noff[i], m->fde_nprm, nullptr) != SAPCU_OK) rc = SAPCU_ERR_HIP;
                if (rc == SAPCU_OK) hip_ok(hipDeviceSynchronize(), "pack fd encoder operands");
            }
        }
    }
    if (rc != SAPCU_OK) {
        if (m->w16_hi) (void)hipFree(m->w16_hi);
        if (m->w16_lo) (void)hipFree(m->w16_lo);
        if (m->chain_w) (void)hipFree(m->chain_w);
        if (m->ovf_dev) (void)hipFree(m->ovf_dev);
        if (m->blob) (void)hipFree(m->blob);
        if (m->ks_dev) (void)hipFree(m->ks_dev);
        if (m->gate_dev) (void)hipFree(m->gate_dev);
        if (m->fde_w) (void)hipFree(m->fde_w);
        if (m->fde_nprm) (void)hipFree(m->fde_nprm);
        delete m;
        return rc;
    }
    *out = m;
    return SAPCU_OK;
}

int sapcu_model_destroy(sapcu_model_t m) {
    if (!m) return SAPCU_OK;
    if (m->w16_hi) (void)hipFree(m->w16_hi);
    if (m->w16_lo) (void)hipFree(m->w16_lo);
    if (m->chain_w) (void)hipFree(m->chain_w);
    if (m->ovf_dev) (void)hipFree(m->ovf_dev);
    if (m->blob) (void)hipFree(m->blob);
    if (m->ks_dev) (void)hipFree(m->ks_dev);
    if (m->gate_dev) (void)hipFree(m->gate_dev);
    if (m->fde_w) (void)hipFree(m->fde_w);
    if (m->fde_nprm) (void)hipFree(m->fde_nprm);
    delete m;
    return SAPCU_OK;
}

int64_t sapcu_workspace_bytes(sapcu_model_t m, int64_t b, int m_pts) {
    if (!m || b < 0 || m_pts < 1 || m_pts > 128) {
        set_error("workspace_bytes: bad argument");
        return SAPCU_ERR_ARG;
    }
    return m->kind == SAPCU_KIND_FN ? fn_ws_bytes(m, b, m_pts) : fd_ws_bytes(m, b, m_pts);
}

int sapcu_model_gate_violations(sapcu_model_t m, int* count_host) {
    SAPCU_CHECK_ARG(m && count_host, "gate_violations: null pointer");
    *count_host = 0;
    if (m->gate_dev) SAPCU_CHECK_HIP(hipMemcpy(count_host, m->gate_dev, sizeof(int), hipMemcpyDeviceToHost));
    return SAPCU_OK;
}

int sapcu_model_gemm_mode(sapcu_model_t m, int* split_f16_host, int* range_overflows_host) {
    SAPCU_CHECK_ARG(m && split_f16_host && range_overflows_host, "gemm_mode: null pointer");
    *split_f16_host = m->sf16 ? 1 : 0;
    *range_overflows_host = 0;
    if (m->ovf_dev) SAPCU_CHECK_HIP(hipMemcpy(range_overflows_host, m->ovf_dev, sizeof(int), hipMemcpyDeviceToHost));
    return SAPCU_OK;
}

int sapcu_model_fused_blocks(sapcu_model_t m, int m_pts, int* mask_host) {
    SAPCU_CHECK_ARG(m && mask_host && m_pts >= 1 && m_pts <= 128, "fused_blocks: bad argument");
    int mask = 0;
    if (m->kind == SAPCU_KIND_FN) {
        for (int l = 0; l < 3; ++l)
            if (fn_block_fused(m, l, m_pts)) mask |= 1 << l;
    } else {
        if (fd_encoder_fused(m, m_pts)) mask = 1;
        else if (fd_x0_path(m, m_pts)) mask = 2;
    }
    *mask_host = mask;
    return SAPCU_OK;
}

int sapcu_fn_forward(sapcu_model_t m, const float* patch, int64_t b, int m_pts, const int32_t* knn_in, int32_t* knn_out,
                     float* normals_out, void* workspace, int64_t ws_bytes, void* const* taps_host, void* stream) {
    SAPCU_CHECK_ARG(m && m->kind == SAPCU_KIND_FN, "fn_forward: not an fn model handle");
    SAPCU_CHECK_ARG(patch && normals_out && workspace, "fn_forward: null pointer");
    SAPCU_CHECK_ARG(b >= 0 && m_pts >= 1 && m_pts <= 128, "fn_forward: need 1 <= m_pts <= 128 (got %d)", m_pts);
    return fn_forward(m, patch, b, m_pts, knn_in, knn_out, normals_out, workspace, ws_bytes, taps_host,
                      (hipStream_t)stream);
}

int sapcu_fd_forward(sapcu_model_t m, const float* patch, int64_t b, int m_pts, const int32_t* knn_force, float* dist_out,
                     void* workspace, int64_t ws_bytes, void* const* taps_host, void* stream) {
    SAPCU_CHECK_ARG(m && m->kind == SAPCU_KIND_FD, "fd_forward: not an fd model handle");
    SAPCU_CHECK_ARG(patch && dist_out && workspace, "fd_forward: null pointer");
    SAPCU_CHECK_ARG(b >= 0 && m_pts >= 1 && m_pts <= 128, "fd_forward: need 1 <= m_pts <= 128 (got %d)", m_pts);
    return fd_forward(m, patch, b, m_pts, knn_force, dist_out, workspace, ws_bytes, taps_host, (hipStream_t)stream);
}

}  // extern "C"
