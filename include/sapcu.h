/*
 * sapcu.h — C ABI of libsapcu_hip.so: the MI355X (gfx950) implementation of the reference's
 * per-query-point inference hot path.
 *
 * Conventions:
 *   - every pointer is a DEVICE pointer unless its name ends in _host;
 *   - the caller owns all inputs, outputs and workspaces; the library owns only what sits
 *     behind a sapcu_model_t handle; no *_forward call allocates;
 *   - every entry point returns 0 on success and a negative sapcu_status otherwise; it never
 *     throws and never exits; sapcu_last_error() returns the text of the calling thread's last
 *     failure;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); all work is
 *     enqueued on it and nothing synchronises unless stated;
 *   - a handle is immutable after create: concurrent forwards on different streams with
 *     different workspaces are legal.
 *
 * Each entry point cites the reference lines (relative to /root/reference) it replaces.
 */
#ifndef SAPCU_H
#define SAPCU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The ABI version changes whenever a caller built against the previous header could misbehave with this library:
 *   1 -> 2: SAPCU_FD_TAP_COUNT grew from 5 to 6 (SAPCU_FD_TAP_X0 appended) and sapcu_fd_forward reads all SAPCU_FD_TAP_COUNT
 *           entries of a non-NULL `taps_host` array — a version-1 caller's 5-entry array would be read one pointer past its end.
 * A binding must refuse to run unless sapcu_abi_version() == the SAPCU_ABI_VERSION it was built against (sapcu_amd/_lib.py does). */
#define SAPCU_ABI_VERSION 2

typedef enum {
    SAPCU_OK = 0,
    SAPCU_ERR_ARG = -1,        /* bad pointer / size / hyper-parameter */
    SAPCU_ERR_WORKSPACE = -2,  /* workspace too small (see *_workspace_bytes) */
    SAPCU_ERR_HIP = -3,        /* a HIP runtime call or launch failed */
    SAPCU_ERR_UNSUPPORTED = -4 /* legal in the reference, not built here (e.g. use_snn_decoder) */
} sapcu_status;

typedef struct sapcu_model* sapcu_model_t;

int sapcu_abi_version(void);
const char* sapcu_last_error(void);

/* ---------------------------------------------------------------- geometry (float64) ---- */

/* Outer kNN: replaces KDTree(data).query(q, k) — generation.py:110,127,153.
 * cloud [n,3] f64, queries [b,3] f64 -> idx_out [b,k] int64 ascending (distance, index);
 * distances are sum_c (q_c-p_c)^2 accumulated x,y,z in f64 without FMA contraction.
 * dist_out [b,k] f64 (euclidean, i.e. sqrt) and patch_out [b,k,3] f32 are optional (NULL to
 * skip); patch_out = f32(cloud[idx] - q), the subtraction done in f64 (generation.py:128-129,137).
 * Requires 1 <= k <= 128 and k <= n. */
int sapcu_knn_gather_f64(const double* cloud, int64_t n, const double* queries, int64_t b, int k,
                         int64_t* idx_out, double* dist_out, float* patch_out, void* stream);

/* Gather + centre (+ optional per-patch Rodrigues rotation of the normal onto +x) —
 * generation.py:154-160,168 and rotation_matrix_from_vectors :30-47.
 * normals [b,3] f32 or NULL (no rotation). patch_out [b,k,3] f32; maths in f64. */
int sapcu_gather_rotate_f64(const double* cloud, int64_t n, const double* queries, int64_t b,
                            const int64_t* idx, int k, const float* normals, float* patch_out,
                            void* stream);

/* Displacement out = p + n*d (f32 product promoted to f64) — generation.py:171-172. */
int sapcu_displace_f64(const double* queries, const float* normals, const float* dist, int64_t b,
                       double* out, void* stream);

/* Farthest-point sampling of the refined cloud down to the target count — generate.py:56-74
 * (`farthest_point_sample`): f32 points, start index n/2, running minimum of the squared distance
 * ((dx^2+dy^2)+dz^2, separately rounded) to the chosen set, arg-max with ties to the smallest index.
 * One persistent launch (one workgroup per CU, points and running distances in registers; per step every
 * workgroup publishes its best candidate in a step-tagged mailbox and polls the others' — no atomics, no
 * separate barrier; csrc/fps.hip).  xyz [n,3] f32 device, idx_out [npoint] int64 device; the
 * workspace holds sapcu_fps_workspace_bytes(npoint) bytes.  n <= #CU * 8192 (2,097,152 on MI355X).
 * Synchronises `stream` before returning (the indices go to the host next, generate.py:74). */
int64_t sapcu_fps_workspace_bytes(int64_t npoint);
int sapcu_fps_f32(const float* xyz, int64_t n, int64_t npoint, int64_t* idx_out, void* workspace,
                  int64_t workspace_bytes, void* stream);

/* Seed generation in process (HOST pointers, CPU code) — replaces the `./dense <cell> <n>` subprocess and
 * its test.xyz / target.xyz text files (generation.py:112-118, dense.cpp:175-252): breadth-first voxel flood
 * from the occupied voxels, emitting (in the reference's order, rounded to 6 decimals like its "%lf" output)
 * the voxel centres whose distance to the local triangle fan lies in [0.011, 0.015].
 * seeds_out_host [capacity,3] f64; *count_host = number of seeds found (SAPCU_ERR_WORKSPACE if > capacity:
 * call again with a larger buffer).  Row (f-1) of SURVEY.md §8f: not part of the GPU hot path. */
int sapcu_dense_seeds_host(const double* cloud_host, int64_t n, double cell, double* seeds_out_host,
                           int64_t capacity, int64_t* count_host);

/* ---------------------------------------------------------------- neuron unit ----------- */

/* Self-feeding T-step neuron loop `for t: x,*st = snn(x,*st)` — fn/snn_coder.py:87-153,
 * 319-320 (LIF) and fd/snn_coder.py:198-275 (EIF when delta_T != NULL).
 * x [rows, channels] f32 (channel = fastest axis); params are per-channel RAW parameters
 * (clamped inside, as the reference does).  Outputs (each optional): spikes of the last step
 * and the final membrane / threshold / refractory state. */
int sapcu_neuron_selfloop(const float* x, int64_t rows, int channels, int steps,
                          const float* membrane_decay, const float* threshold_adapt,
                          const float* refractory_decay, const float* threshold_base,
                          const float* delta_T, const float* theta_rh,
                          float* spikes_out, float* membrane_out, float* threshold_out,
                          float* refractory_out, void* stream);

/* The STEPPING form of the same neurons as fd's encoder runs them (fd/snn_coder.py:432-474 in eval mode): x enters at step 0
 * only — from step 1 on the refractory gate `x * (r <= 0)` is closed, fd:133,249 — and every step's spikes are kept:
 * spikes_out [steps, rows, channels].  The packed two-chain arithmetic of csrc/fd_encoder.hip / fd_edge_neuron_kernel
 * (channel_pairs = 0: two rows of one channel per chain pair; 1: two channels of one row).  *gate_open_out (device int, the
 * caller zeroes it) += number of (pair, step >= 1) events at which the kernels' gate test found r <= 0 — the same test that feeds
 * sapcu_model_gate_violations; 0 for every finite input, as in the reference (its clamped spike is >= 3.85e-23). */
int sapcu_neuron_drive(const float* x, int64_t rows, int channels, int steps,
                       const float* membrane_decay, const float* threshold_adapt,
                       const float* refractory_decay, const float* threshold_base,
                       const float* delta_T, const float* theta_rh, int channel_pairs,
                       float* spikes_out, float* membrane_out, float* threshold_out,
                       float* refractory_out, int* gate_open_out, void* stream);

/* Training-mode neuron loop (first piece) — fn/snn_coder.py:87-151 with `self.training`, driven
 * as `for t: x, *st = snn(x, *st)` (:318-320): forward value = hard spikes (m - theta > 0), derivative = the soft
 * surrogate's (straight-through estimator, :148-151), the refractory gate is a constant mask.
 * x, spikes_out, grad_spikes, grad_x: [rows, channels] f32; parameters and their gradients: RAW per-channel values
 * (clamped inside; a parameter outside its clamp range gets zero gradient, as torch.clamp gives it).  1 <= steps <= 8.
 * The backward recomputes the forward from x (nothing is saved between the two calls); parameter gradients are summed
 * over rows in a fixed order (deterministic) through a workspace of sapcu_lif_train_workspace_bytes(rows, channels). */
int sapcu_lif_train_forward(const float* x, int64_t rows, int channels, int steps, const float* membrane_decay,
                            const float* threshold_adapt, const float* refractory_decay,
                            const float* threshold_base, float* spikes_out, void* stream);
int64_t sapcu_lif_train_workspace_bytes(int64_t rows, int channels);
int sapcu_lif_train_backward(const float* x, const float* grad_spikes, int64_t rows, int channels, int steps,
                             const float* membrane_decay, const float* threshold_adapt,
                             const float* refractory_decay, const float* threshold_base, float* grad_x,
                             float* grad_membrane_decay, float* grad_threshold_adapt,
                             float* grad_refractory_decay, float* grad_threshold_base, void* workspace,
                             int64_t workspace_bytes, void* stream);

/* Layer pieces of the training step (row f-4): BatchNorm in TRAINING mode over the rows of a [rows, channels] tensor (what
 * nn.BatchNorm1d/2d do to a 1x1 convolution's output, fn/snn_coder.py:225-252: batch mean and biased variance), its
 * backward, and the weight / bias gradient of the 1x1 convolution.  Deterministic reductions (fixed-order f64 partial
 * sums).  workspace: sapcu_train_workspace_bytes(rows, channels, k) bytes (k = 0 for the BatchNorm calls).
 *   forward : z = (y - mean) * invstd * gamma + beta;   mean/var (biased)/invstd [channels] are outputs (the running
 *             statistics update is the caller's: momentum, unbiased variance)
 *   backward: grad_y = gamma*invstd*(grad_z - mean_r(grad_z) - y_hat*mean_r(grad_z*y_hat)); grad_gamma, grad_beta
 *   wgrad   : grad_w[n,k] = sum_r grad_y[r,n] * x[r,k]  (exact-f32 MFMA);  grad_bias[n] = sum_r grad_y[r,n] (optional)
 * The data gradient of the convolution is a plain GEMM: sapcu_gemm_f32(grad_y, rows, n, ldy, W^T [k,n], k, ...). */
int64_t sapcu_train_workspace_bytes(int64_t rows, int channels, int k);
int sapcu_bn_train_forward(const float* y, int64_t rows, int channels, const float* gamma, const float* beta, float eps,
                           float* z_out, float* mean_out, float* var_out, float* invstd_out, void* workspace,
                           int64_t workspace_bytes, void* stream);
int sapcu_bn_train_backward(const float* y, const float* grad_z, int64_t rows, int channels, const float* gamma,
                            const float* mean, const float* invstd, float* grad_y, float* grad_gamma,
                            float* grad_beta, void* workspace, int64_t workspace_bytes, void* stream);
int sapcu_conv1x1_wgrad_f32(const float* grad_y, int ldy, const float* x, int ldx, int64_t rows, int n, int k,
                            float* grad_w, float* grad_bias, void* workspace, int64_t workspace_bytes, void* stream);

/* bf16-operand variants of the two training GEMMs (csrc/train_bf16.hip) — the counterpart of the reference's AMP training
 * (fn/trainer.py:67-83 torch.amp.autocast; BASELINE config 5): operands rounded to bf16 (nearest even), f32 accumulation on the
 * bf16 MFMA; tensors stay f32 in memory.  sapcu_gemm_bf16: c[r,n] = a[r,k] . w[n,k]^T + bias (forward; data gradient with
 * w = W^T); k % 4 == 0.  sapcu_conv1x1_wgrad_bf16: as sapcu_conv1x1_wgrad_f32 (grad_bias summed in f32); workspace
 * sapcu_wgrad_bf16_workspace_bytes(rows, n, k).  Opt-in (sapcu_amd.train.gemm_precision / Trainer(use_amp=True)): this IS a
 * precision reduction; the f32 entries above remain the parity reference. */
int sapcu_gemm_bf16(const float* a, int64_t r, int k, int lda, const float* w, int n, const float* bias, float* c, int ldc,
                    void* stream);
int64_t sapcu_wgrad_bf16_workspace_bytes(int64_t rows, int n, int k);
int sapcu_conv1x1_wgrad_bf16(const float* grad_y, int ldy, const float* x, int ldx, int64_t rows, int n, int k,
                             float* grad_w, float* grad_bias, void* workspace, int64_t workspace_bytes, void* stream);

/* Per-channel softmax over the k neighbours + weighted aggregation (fn/snn_coder.py:379-389), as its own differentiable op:
 *   res[pt, c] = sum_j softmax_j(a[pt, j, c] / sqrt_hd) * (v[nbr(pt, j), c] + pe[pt, j, c])
 * a, pe, grad_a, grad_pe: [pts*kk, d] (edge rows); v, grad_v: [pts, ld] rows; idx [pts*kk] = in-patch neighbour indices,
 * m points per patch (pts % m == 0).  keep: NULL, or [pts*kk, d] holding 0 or 1/(1-p) — the attention dropout of train()
 * mode applied to the softmax weights (fn/snn_coder.py:383); the caller draws it.  The backward zeroes grad_v[., 0:d] and
 * accumulates with float atomics. */
int sapcu_softmax_agg_forward(const float* a, const float* pe, const float* v, int ldv, const int32_t* idx, const float* keep,
                              int64_t pts, int m, int kk, int d, float sqrt_hd, float* res, void* stream);
int sapcu_softmax_agg_backward(const float* a, const float* pe, const float* v, int ldv, const int32_t* idx, const float* keep,
                               const float* grad_res, int64_t pts, int m, int kk, int d, float sqrt_hd,
                               float* grad_a, float* grad_pe, float* grad_v, int ldgv, void* stream);

/* Row gather out[r,:] = src[index[r],:] (index_points on flattened rows, fn/snn_coder.py:19-29) and its backward: zero
 * grad_src[0:src_rows, 0:d], then grad_src[index[r],:] += grad_out[r,:] (float atomics). */
int sapcu_gather_rows(const float* src, int ld_src, const int64_t* index, int64_t rows, int d, float* out, void* stream);
int sapcu_scatter_add_rows(const float* grad_out, const int64_t* index, int64_t rows, int d, float* grad_src, int ld_grad,
                           int64_t src_rows, void* stream);
/* The same sum for GROUPED indices, deterministic (no atomics, no zeroing pass): rows come in groups of group_rows consecutive
 * rows (a patch's m k edge rows) whose indices all point into the group_src_rows destination rows of the same group (the
 * patch's m points: index[r] / group_src_rows == r / group_rows).  One workgroup per group builds the inverse table in LDS and
 * sums every destination's sources in ascending source order — bit-identical from run to run, which the float-atomic form is
 * not.  This is what the training step uses (index_points' backward, and inside sapcu_softmax_agg_backward for grad_v).
 * bad_count: NULL or a device int that receives the number of entries pointing outside their group (skipped). */
int sapcu_scatter_add_rows_grouped(const float* grad_out, const int64_t* index, int64_t rows, int d, float* grad_src, int ld_grad,
                                   int64_t src_rows, int group_src_rows, int group_rows, int* bad_count, void* stream);

/* Max over the m points of each patch, x [groups*m, c] -> out [groups, c] (adaptive_max_pool1d, fn/snn_coder.py:472), with
 * the arg-max (ties -> first point) for the backward, which routes grad_out to that point. */
int sapcu_group_max_forward(const float* x, int64_t groups, int m, int c, float* out, int32_t* argmax_out, void* stream);
int sapcu_group_max_backward(const float* grad_out, const int32_t* argmax, int64_t groups, int m, int c, float* grad_x,
                             void* stream);

/* In-patch kNN `topk(-|xi|^2 + 2 xi.xj - |xj|^2)` — fn/snn_coder.py:31-39, fd/snn_coder.py:25-32.
 * feat [b, m, ld] f32 (point-major, first c columns used), 1 <= m <= 128, k <= m.
 * idx_out [b,m,k] int32, descending score, equal scores by ascending index. */
int sapcu_patch_knn(const float* feat, int64_t b, int m, int c, int ld, int k, int32_t* idx_out,
                    void* stream);

/* ---------------------------------------------------------------- models ---------------- */

#define SAPCU_KIND_FN 0 /* ImprovedSNNNormalEstimation   — fn/snn_coder.py:627-699 */
#define SAPCU_KIND_FD 1 /* EnhancedSNNDistanceEstimation — fd/snn_coder.py:805-871 */

/* hparams_host (int32):
 *   fn: [k0, k1, k2, emb_dims, time_steps_enc, num_heads]
 *   fd: [k, emb_dims, time_steps_enc, num_heads, n_scales, ks0, ks1, ...]
 * blob: packed f32 parameters on the device (BatchNorm folded, see sapcu_amd/packing.py);
 * dir_host: int64 offsets (in floats) into blob, one per slot of the kind's slot table
 * (SAPCU_FN_SLOTS / SAPCU_FD_SLOTS entries, order fixed by packing.py and model.hip).
 * The library copies the blob; the caller may free it after the call returns.
 * A handle is immutable after creation: the environment switches (SAPCU_GEMM=f32, SAPCU_CHUNK, SAPCU_WS_BUDGET_MB and the
 * parity / ablation switches SAPCU_BT, SAPCU_CHAIN (= 0, or "wide"), SAPCU_FN_MAXFUSE, SAPCU_FD_MAXFUSE, SAPCU_FD_SPLIT, SAPCU_FD_FUSED = 0) are
 * read HERE, once; a forward never reads the environment.  Forwards of one handle (or of several) may run concurrently on
 * different streams / host threads as long as each has its own workspace; launch attributes are set once per device. */
int sapcu_model_create(int kind, const int32_t* hparams_host, int n_hparams, const float* blob,
                       int64_t blob_floats, const int64_t* dir_host, int n_dir, sapcu_model_t* out);
int sapcu_model_destroy(sapcu_model_t m);

/* Workspace bytes needed by a forward of b patches of m_pts points. */
int64_t sapcu_workspace_bytes(sapcu_model_t m, int64_t b, int m_pts);

/* Debug taps: `taps` is NULL or a host array of SAPCU_*_TAP_COUNT device pointers (each may be
 * NULL); selected intermediates are copied out for stage-level parity tests. */
enum {
    SAPCU_FN_TAP_STEM = 0,   /* [b,m,64]  after snn_init            fn:453-457 */
    SAPCU_FN_TAP_BLOCK1 = 1, /* [b,m,64]  trans1 output             fn:460     */
    SAPCU_FN_TAP_BLOCK2 = 2, /* [b,m,64]                                       */
    SAPCU_FN_TAP_BLOCK3 = 3, /* [b,m,64]                                       */
    SAPCU_FN_TAP_POOLED = 4, /* [b,emb]   after max-pool            fn:472     */
    SAPCU_FN_TAP_ENC = 5,    /* [b,2048]  encoder output            fn:475     */
    SAPCU_FN_TAP_LOGITS = 6, /* [b,3]     before LayerNorm(3)       fn:545     */
    SAPCU_FN_TAP_COUNT = 7
};
enum {
    SAPCU_FD_TAP_FUSED0 = 0,  /* [b,m,64]      scale_fusion output at t=0      fd:421  */
    SAPCU_FD_TAP_SPIKES = 1,  /* [T,b,m,960]   the four spike tensors, all t   fd:476  */
    SAPCU_FD_TAP_KNN = 2,     /* [3][b,m,kk] int32 feature-space neighbours, blocks 1-3 (t=0) */
    SAPCU_FD_TAP_POOLED = 3,  /* [T,b,emb]     pooled_t                        fd:479  */
    SAPCU_FD_TAP_ENC = 4,     /* [b,emb]       encoder output                  fd:492  */
    SAPCU_FD_TAP_X0 = 5,      /* [b,m,960]     neuron inputs at t=0 of the four blocks (scale_fusion output | EdgeConv 1-3
                                 after max + BN + LeakyReLU)                      fd:421,455-470 */
    SAPCU_FD_TAP_COUNT = 6
};

/* fn forward — ImprovedSNNNormalEstimation.forward on [b,m,3] (fn/snn_coder.py:670-699)
 * followed by nothing else (the caller's extra F.normalize, generation.py:139, is idempotent
 * up to rounding and is applied by the Python layer).
 * knn_in:  NULL, or the three in-patch neighbour tables [b,m,k0],[b,m,k1],[b,m,k2] int32
 *          concatenated — the reference's shape-keyed KNNCache replay (fn:47-59);
 * knn_out: NULL, or receives the tables this call used (same layout). */
int sapcu_fn_forward(sapcu_model_t m, const float* patch, int64_t b, int m_pts,
                     const int32_t* knn_in, int32_t* knn_out, float* normals_out, void* workspace,
                     int64_t ws_bytes, void* const* taps_host, void* stream);

/* fd forward — EnhancedSNNDistanceEstimation.forward on [b,m,3] (fd/snn_coder.py:853-871).
 * knn_force: NULL, or [3][b,m,min(k,m)] int32 feature-space neighbour tables to use instead of
 * the ones computed in-kernel (test protocol for near-tie flips; DESIGN.md "kNN flips"). */
int sapcu_fd_forward(sapcu_model_t m, const float* patch, int64_t b, int m_pts,
                     const int32_t* knn_force, float* dist_out, void* workspace, int64_t ws_bytes,
                     void* const* taps_host, void* stream);

/* Number of (element, step>=1) events in fd forwards of this handle where the refractory gate
 * `x * (refractory <= 0)` (fd/snn_coder.py:133,249) was found OPEN.  The fd kernels rely on it
 * being closed for t >= 1 (eval-mode spikes are strictly positive) to skip the
 * dead EdgeConv stages; a non-zero count means an output may deviate.  Synchronises the device. */
int sapcu_model_gate_violations(sapcu_model_t m, int* count_host);

/* Which GEMM kernels the handle uses (1 = split-f16: every f32 operand as hi + lo*2^-11 halves, three f16
 * MFMAs per product, f32-quality results at 5.3x the f32-MFMA rate; 0 = exact-f32 MFMA, selected by the
 * environment variable SAPCU_GEMM=f32 at sapcu_model_create or automatically when a parameter exceeds the f16 range), and how
 * many activation tiles exceeded the f16 range in forwards so far (must be 0).  Synchronises the device. */
int sapcu_model_gemm_mode(sapcu_model_t m, int* split_f16_host, int* range_overflows_host);

/* Which stages of the handle run as fused LDS-resident kernels for patches of m_pts points (no device work):
 * fn: bit l (0..2) of *mask_host set = transformer block l+1 runs csrc/fn_edge_chain.hip (else the five-kernel chain);
 * fd: bit 0 set = the encoder (blocks 0-3 + multi_scale_conv) runs csrc/fd_encoder.hip (patches of <= 48 points); bit 1 set = the
 *     per-stage kernels hand the pre-activations x0 to fd_msc_kernel (multi_scale_conv with the spikes of all T steps regenerated
 *     on the CU: larger patches, e.g. the reference's default of 100 points) instead of writing T spike slabs for a GEMM; neither:
 *     the per-stage kernels through HBM.
 * The choice depends on the hyper-parameters, on m_pts and on the switches read at sapcu_model_create (SAPCU_CHAIN=0,
 * SAPCU_FD_FUSED=0, SAPCU_FD_X0=0, SAPCU_GEMM=f32 disable them); it changes speed and workspace size, not results. */
int sapcu_model_fused_blocks(sapcu_model_t m, int m_pts, int* mask_host);

/* out = in / max(||in||_2, 1e-12) row-wise for [b,3] — the extra F.normalize of generation.py:139. */
int sapcu_l2_normalize3(const float* in, float* out, int64_t b, void* stream);

/* The library's MFMA GEMMs, exposed for tests and roofline runs:
 *   C[r,n] = epi( A[r,k] * W[n,k]^T + bias[n] ),  k % 32 == 0, A/W 16-byte aligned, lda % 4 == 0.
 * lif4 == NULL: epi = identity.  Otherwise lif4 = raw neuron parameters [4][n] and the epilogue is
 * the lif_steps-step self-feeding LIF loop (the fused form of conv+BN -> snn loop, fn:317-320).
 * w16_ws == NULL: exact-f32 MFMA kernel.  Otherwise 4*n*k + 16 bytes of scratch: W is split into f16
 * hi/lo halves there and a split-f16 kernel runs (3 x f16 MFMA per product; last 4 bytes of the
 * scratch = count of activation values beyond the f16 range).
 * a_split_rows != 0: A is in "split rows" (see sapcu_to_split_rows) -> the all-DMA kernels (lda % 8 == 0): 1 = the kernel
 * the models would pick for the shape (big-tile for >= 1024 rows, else the 128x128 ring kernel), 2 = the ring kernel
 * whatever the shape (the two are bit-identical; the parity tests compare them);
 * c_split_rows: write C as split rows (needs w16_ws). */
int sapcu_gemm_f32(const float* a, int64_t r, int k, int lda, const float* w, int n, const float* bias,
                   const float* lif4, int lif_steps, float* c, int ldc, void* w16_ws, int a_split_rows,
                   int c_split_rows, void* stream);

/* f32 [rows,k] (row pitch ld_in floats) -> "split rows" (row pitch ld_out floats): each value x becomes two
 * f16 halves hi = f16_rn(x) and lo = f16_rn(x - hi) inside its row — the operand format of the split-f16 DMA GEMMs.
 * ld_out % 32 == 0: interleaved by groups of 32 elements, group q = one 128-byte line: hi halves of elements
 * 32q..32q+31 at half-index 64q, their lo halves at 64q + 32.  Otherwise: hi at half-index c, lo at ld_out + c.
 * Inside the models the producing kernels write it directly. */
int sapcu_to_split_rows(const float* in, int64_t rows, int k, int ld_in, float* out, int ld_out, void* stream);

/* The positional-encoding GEMM of one fn block — the heaviest single launch shape of the path:
 *   pe[row,:]      = LIF_x4( W . pe1[row,:] + bias )                         (fn/snn_coder.py:360-363)
 *   attn_in[row,:] = q[pt(row),:] - k[nbr(row),:] + pe[row,:]                (fn/snn_coder.py:367-368)
 * pe1 [r,d]; qkv [b*m, 3d] (q | k | v); idx [r] = flattened [b,m,kk] in-patch neighbours; w [d,d];
 * edge_table_ws: 8*r bytes of scratch (row -> (q row, k row) table, rebuilt by every call).
 * w16_ws: NULL -> exact-f32 MFMA kernel; else 4*d*d + 16 bytes of scratch -> split-f16 (3 x f16 MFMA) kernel.
 * split_rows != 0 (needs w16_ws): pe1 is in split rows and attn_in_out is written as split rows — the form the
 * models' unfused chain runs (all-DMA kernels; 1 = big-tile kernel where it takes the shape, 2 = ring kernel only). */
int sapcu_posenc_gemm_f32(const float* pe1, int64_t r, int d, const float* w, const float* bias,
                          const float* lif4, int lif_steps, const float* qkv, const int32_t* idx, int kk,
                          int m_pts, float* pe_out, float* attn_in_out, void* edge_table_ws, void* w16_ws,
                          int split_rows, void* stream);

/* The whole per-edge chain of one fn transformer block in ONE kernel, activations in LDS (csrc/fn_edge_chain.hip) — what the
 * models run for all three blocks at the reference's neighbour counts: (d, kk) = (128, 24), (256, 18), (512, 12) EXACTLY.  A
 * block with any other k_values entry — or a patch of fewer than kk points, which clamps kk — falls back to the five-kernel
 * chain (same results bit for bit, ~6x the workspace, slower); sapcu_model_fused_blocks reports which blocks of a handle run
 * fused for a given patch size.  fn/snn_coder.py:355-389:
 *   pe1 = LIF_x4(fc_delta(x_i - x_j)), pe = LIF_x4(fc_delta2(pe1)), attn_in = q_i - k_j + pe, g = LIF_x4(fc_gamma(attn_in)),
 *   a = fc_gamma2(g), res[i,:] = sum_j softmax_j(a / sqrt(d / heads)) * (v_j + pe)
 * patch [points,3] f32 (points = patches * m_pts, patch-major); idx [points*kk] int32 in-patch neighbours; qkv [points, 3d]
 * (q | k | v); w_delta [d,3]; w1 / w2 / w3 = fc_delta2 / fc_gamma / fc_gamma2 [d,d]; biases [d]; lif* = raw neuron parameters
 * [4,d] (membrane_decay, threshold_adapt, refractory_decay, threshold_base; clamped inside as the reference does);
 * res_out [points, d] f32.  Results equal the five-kernel chain (sapcu_posenc_gemm_f32 etc.) bit for bit.
 * workspace: sapcu_fn_edge_chain_workspace_bytes(points, d, kk) bytes (edge records + the split / fragment-ordered weights,
 * rebuilt by every call).  Returns SAPCU_ERR_ARG for any other (d, kk). */
int64_t sapcu_fn_edge_chain_workspace_bytes(int64_t points, int d, int kk);
int sapcu_fn_edge_chain_f32(const float* patch, const int32_t* idx, int64_t points, int m_pts, int d, int kk,
                            const float* qkv, const float* w_delta, const float* b_delta, const float* lif_delta,
                            const float* w1, const float* b1, const float* lif1, const float* w2, const float* b2,
                            const float* lif2, const float* w3, const float* b3, int heads, int lif_steps,
                            float* res_out, void* workspace, int64_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SAPCU_H */
