// Launchers of patch_ops.hip (host side), used by model.hip and api.hip.
#pragma once
#include "common.h"

namespace sapcu {

int launch_patch_knn_strided(const float* feat, int64_t b, int64_t pstride, int m, int c, int ld, int k,
                             int32_t* idx, hipStream_t st);
int launch_patch_knn_multi(const float* feat, int64_t b, int64_t pstride, int m, int c, int ld, int ntab,
                           const int* ks, int32_t* const* idx, hipStream_t st);
int launch_patch_knn(const float* feat, int64_t b, int m, int c, int ld, int k, int32_t* idx, hipStream_t st);
int launch_neuron_selfloop(const float* x, int64_t rows, int ch, int T, const float* md, const float* ta,
                           const float* rd, const float* tb, const float* dT, const float* rh, float* so, float* mo,
                           float* to, float* ro, hipStream_t st);
int launch_neuron_drive(const float* x, int64_t rows, int ch, int T, const float* md, const float* ta, const float* rd, const float* tb,
                        const float* dT, const float* rh, int pairv, float* so, float* mo, float* to, float* ro, int* gate, hipStream_t st);
int launch_fn_stem(const float* patch, int64_t rows, const float* w, const float* bias, const float* lif, int T,
                   float* out, hipStream_t st);
int launch_fn_pe1(const float* patch, const int32_t* idx, int64_t rows, int m, int kk, int d, const float* w,
                  const float* bias, const float* lif, int T, float* out, int split, hipStream_t st);
// fused per-edge chain of an fn block (fn_edge_chain.hip): tab / pd are filled by the launcher
struct ChainArgs {
    int64_t P;                 // points (rows of qkv / res)
    int m;                     // points per patch
    const int2* tab;           // [P*kk] (point row, neighbour row)
    const float4* pd;          // [P*kk] x_i - x_j
    const float* qkv;          // [P, ldq]: q | k | v, d columns each
    int ldq;
    const float* wd;           // fc_delta  [d][3], bias [d], raw neuron parameters [4][d]
    const float* bd;
    const float* lifd;
    const _Float16* w1p;       // fc_delta2 in fragment order (launch_pack_chain_weights), bias, neuron parameters
    const float* b1;
    const float* lif1;
    const _Float16* w2p;       // fc_gamma
    const float* b2;
    const float* lif2;
    const _Float16* w3p;       // fc_gamma2
    const float* b3;
    float inv_sqrt_hd;
    float* res;                // [P, d] (split rows when res_split)
    int res_split;
    int T;                     // neuron self-loop steps (4)
    int wide_offsets;          // 1: 64-bit gather addresses even where 32-bit byte offsets would do (SAPCU_CHAIN=wide; parity tests)
};
bool fn_edge_chain_ok(int d, int kk);
int launch_fn_edge_chain(ChainArgs a, const float* patch, const int32_t* idx, int d, int kk, int2* tab_ws, float4* pd_ws,
                         hipStream_t st);      // tab_ws / pd_ws: P*kk entries each
int launch_pack_chain_weights(const void* w16_hi, const void* w16_lo, int d, void* out, hipStream_t st);
int launch_edge_table(const int32_t* idx, int64_t rows, int m, int kk, int2* tab, hipStream_t st);
int launch_fn_softmax_agg(const float* a, const float* pe, const float* v, int ldv, const int32_t* idx, int64_t pts,
                          int m, int kk, int d, float sqrt_hd, float* res, int split, hipStream_t st);
int launch_decode_max_keys(const unsigned* keys, int64_t count, float* out, hipStream_t st);   // float_from_max_key, element-wise
int launch_rowgroup_max(const float* in, int64_t groups, int m, int c, float* out, hipStream_t st);
int launch_fn_tail(const float* h, int64_t b, int kdim, const float* w, const float* bias, const float* lnw,
                   const float* lnb, float* logits, float* normals, hipStream_t st);
int launch_to_split_rows(const float* in, int64_t rows, int k, int ld_in, float* out, int ld_out, hipStream_t st);
int launch_l2_normalize3(const float* in, float* out, int64_t b, hipStream_t st);
int launch_fd_edge0(const float* patch, const int32_t* idx, int kmax, int64_t pts, int m, int nscale,
                    const int32_t* ks_dev, const float* w, const float* bias, float* out, hipStream_t st);
int launch_fd_neuron(bool eif, int mode, const float* in, int ldi, const int32_t* idx, int kk, int m,
                     const float* shift, int64_t pts, int C, const float* prm, int T, float* spk, int ldo, int coff,
                     float* pre_out, int* gate_violations, hipStream_t st, float* spk_split = nullptr,    // spk_split: see fd_neuron_kernel
                     float* x0_out = nullptr);            // x0_out [pts, ldo]: pre-activations out, step 0 only (the x0 path, fd_msc_kernel)
// fused fd encoder (fd_encoder.hip)
typedef _Float16 fe_half8 __attribute__((ext_vector_type(8)));
struct FdEncArgs {
    const float* patch;          // [b, m, 3] rotated patches of this launch
    int64_t b;                   // patches of this launch (pooled is [T, b, emb])
    int64_t b_total, s0;         // the forward's batch and this launch's first patch in it (tap / knn_force indexing)
    int m, T, kk, kmax0, nscale, emb;
    int ks[4];                   // clamped to m
    const float* e0_w;           // [S][64][6]
    const float* e0_b;           // [S][64]
    const _Float16* fuse_wp;     // scale_fusion [64, 64 S] in fragment order (launch_pack_frag_weights)
    const float* fuse_b;
    const _Float16* edge_wp[3];  // EdgeConv l: [2 C', C] in fragment order (rows 0..C'-1: W1 + W2, rows C'..: W1)
    const float* shift[3];
    const _Float16* msc_wp;      // multi_scale_conv [emb, 960] in fragment order
    const float* msc_b;
    const float* nprm;           // [960][8] clamped neuron parameters (decay, adapt, rdecay, theta0, dT, rh, 0, 0)
    float* pooled;               // [T, b, emb]
    const int32_t* knn_force;    // [3][b_total][m][kk] or null
    int32_t* tap_knn;            // [3][b_total][m][kk] or null
    float* tap_fused0;           // [b_total, m, 64] or null
    float* tap_spikes;           // [T, b_total, m, 960] or null
    float* tap_x0;               // [b_total, m, 960] or null
    int* gate;                   // refractory gate found open at t >= 1 (must stay 0)
    int* ovf;                    // a block-0 EdgeConv value beyond the f16 range of the split operand
};
// multi_scale_conv + max over the points from the pre-activations x0 [points, 960] (fd_msc_kernel, fd_encoder.hip): the per-stage
// path's replacement for "T spike slabs through HBM + big-tile GEMM", any patch size
struct FdMscArgs {
    const float* x0;             // [b * m, 960] neuron inputs at t = 0 of the four blocks (what SAPCU_FD_TAP_X0 shows)
    int64_t b;                   // patches of this launch (pooled is [T, b, emb])
    int64_t b_total, s0;         // the forward's batch and this launch's first patch in it (spike tap indexing)
    int m, T, emb;
    const _Float16* msc_wp;      // multi_scale_conv [emb, 960] in fragment order
    const float* msc_b;
    const float* nprm;           // [960][8] clamped neuron parameters
    float* pooled;               // [T, b, emb]
    float* tap_spikes;           // [T, b_total, m, 960] or null
    int* gate;
};
bool fd_msc_ok(int m, int emb, int T);
int launch_fd_msc(const FdMscArgs& a, hipStream_t st);
bool fd_encoder_ok(int m, int nscale, int emb, int T);
int launch_fd_encoder(const FdEncArgs& a, hipStream_t st);
int launch_pack_frag_weights(const void* w16_hi, const void* w16_lo, int n, int k, void* out, hipStream_t st);
int launch_pack_fd_neuron(const float* raw, int C, int eif, int coff, float* out, hipStream_t st);
int launch_fd_pre(const float* in, int ldi, const int32_t* idx, int kk, int m, const float* shift, int64_t pts, int C, float* out,
                  int ldo, int coff, hipStream_t st);      // x0 tap of the per-stage path (debug)
int launch_fd_temporal(const float* pooled, int T, int64_t b, int emb, const float* tw, const float* lif, float* out,
                       hipStream_t st);
int launch_fd_tail(const float* x, const float* qkv, int64_t b, int heads, const float* wo_t, const float* bo,
                   const float* lnw, const float* lnb, const float* wh_t, const float* bh, const float* wd,
                   const float* bd, float* attn_out, float* dist, hipStream_t st);

}  // namespace sapcu
