// Training-mode neuron loop, forward and backward: the first piece of SURVEY.md §8 row f-4 (training backward).
//
// /root/reference/fn/snn_coder.py:87-151 in `self.training` mode, driven as fn drives it (`for t: x, *st = snn(x, *st)`,
// :318-320): the forward value of a spike is the HARD threshold (m - theta > 0), its derivative the soft surrogate
// 0.5*N'(u) + 0.5*10*sigma'(10u) on clamp(u, +-10) (straight-through estimator, :148-151); the refractory gate
// `x * (r <= 0)` is a constant mask for the gradient.  Unlike the eval path nothing here is peeled: in training the spikes
// are 0/1 and the gate re-opens, so every step is evaluated as written.
//
// One thread = one (row, channel) element: the T-step forward keeps the per-step quantities the reverse sweep needs in
// registers (T <= 8), the reverse sweep runs in the same thread — no activations are saved between forward and backward,
// the backward kernel recomputes the T steps from x.  Parameter gradients are summed over rows deterministically:
// per-workgroup partial sums (fixed order inside a thread column, then across the workgroup's row slabs in LDS) go to a
// workspace, a second kernel adds the partials in ascending workgroup order.
#include "common.h"
#include "ops.h"
#include "../../include/sapcu.h"

namespace sapcu {

constexpr int LT_MAX_T = 8;
constexpr int LT_ROWS_PER_WG = 64;    // 4 row slabs of 16 rows x 64 channels per 256-thread workgroup

struct TrainP {
    float decay, adapt, rdecay, theta0;
};

__device__ __forceinline__ TrainP load_train_params(const float* md, const float* ta, const float* rd, const float* tb, int c) {
    return TrainP{fminf(fmaxf(md[c], 0.1f), 0.99f), fminf(fmaxf(ta[c], 0.001f), 0.1f), fminf(fmaxf(rd[c], 0.1f), 0.95f), tb[c]};
}

// soft surrogate value is not needed in training (the forward value is the hard spike); its derivative is
__device__ __forceinline__ float surrogate_grad(float u) {
    if (!(u >= -10.0f && u <= 10.0f)) return 0.f;              // torch.clamp passes the gradient on [min, max] only
    const float gauss = expf(-0.5f * u * u) * 0.3989422804014327f;
    const float sg = 1.0f / (1.0f + expf(-10.0f * u));
    return 0.5f * (-u * gauss) + 0.5f * (10.0f * sg * (1.0f - sg));
}

struct StepRec {      // what the reverse sweep needs from one forward step
    float m, r, th;   // state BEFORE the step
    float mm, u, sp;  // membrane after integration, threshold distance, (hard) spike
    float gate;       // (r <= 0)
};

// TT = number of steps at compile time: the loops unroll and rec[] stays in registers (a run-time trip count would put it
// in scratch memory).
template <int TT>
__device__ __forceinline__ float train_forward(float x, const TrainP& p, StepRec (&rec)[TT]) {
    float m = 0.f, r = 0.f, th = p.theta0, in = x, sp = 0.f;
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        StepRec& q = rec[t];
        q.m = m; q.r = r; q.th = th;
        q.gate = r <= 0.f ? 1.f : 0.f;
        const float xin = in * q.gate;
        q.mm = m * p.decay * (1.0f - r) + xin;
        q.u = q.mm - th;
        sp = q.u > 0.f ? 1.f : 0.f;
        q.sp = sp;
        m = q.mm * (1.0f - sp);
        r = r * p.rdecay + sp;
        th = p.theta0 + ((th + p.adapt * sp) - p.theta0) * 0.95f;
        in = sp;
    }
    return sp;
}

template <int TT>
__global__ __launch_bounds__(256) void lif_train_fwd_kernel(const float* __restrict__ x, int64_t rows, int ch,
                                                            const float* md, const float* ta, const float* rd, const float* tb,
                                                            float* __restrict__ spikes) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= rows * ch) return;
    const TrainP p = load_train_params(md, ta, rd, tb, (int)(t % ch));
    StepRec rec[TT];
    spikes[t] = train_forward<TT>(x[t], p, rec);
}

// grid: (ceil(rows / 64), ceil(ch / 64)); thread (slab = tid >> 6, lane = tid & 63) walks rows slab*16 .. +16 of its column
template <int TT>
__global__ __launch_bounds__(256) void lif_train_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gout,
                                                            int64_t rows, int ch, const float* md, const float* ta,
                                                            const float* rd, const float* tb, float* __restrict__ gx,
                                                            float* __restrict__ partial /*[gridDim.x][4][ch]*/) {
    __shared__ float red[4][4][64];
    const int lane = threadIdx.x & 63, slab = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + lane;
    const bool live = c < ch;
    float g_decay = 0.f, g_adapt = 0.f, g_rdecay = 0.f, g_theta = 0.f;
    if (live) {
        const TrainP p = load_train_params(md, ta, rd, tb, c);
        const int64_t row0 = (int64_t)blockIdx.x * LT_ROWS_PER_WG + slab * 16;
        float xv[16], gv[16];                             // all 32 loads in flight before the first dependent use
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const bool ok = row0 + i < rows;
            xv[i] = ok ? x[(row0 + i) * ch + c] : 0.f;
            gv[i] = ok ? gout[(row0 + i) * ch + c] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int64_t row = row0 + i;
            if (row >= rows) break;
            StepRec rec[TT];
            train_forward<TT>(xv[i], p, rec);
            // reverse sweep; adjoints of the state AFTER step t
            float a_sp = gv[i], a_m = 0.f, a_r = 0.f, a_th = 0.f;
#pragma unroll
            for (int t = TT - 1; t >= 0; --t) {
                const StepRec& q = rec[t];
                // th' = theta0 + ((th + adapt*sp) - theta0) * 0.95
                const float a_tht = 0.95f * a_th;
                g_theta += a_th * 0.05f;
                float b_th = a_tht;                       // adjoint of th (before the step)
                g_adapt += a_tht * q.sp;
                a_sp += a_tht * p.adapt;
                // r' = r*rdecay + sp
                float b_r = a_r * p.rdecay;
                g_rdecay += a_r * q.r;
                a_sp += a_r;
                // m' = mm * (1 - sp)
                float a_mm = a_m * (1.0f - q.sp);
                a_sp += -a_m * q.mm;
                // sp = H(u) with the surrogate derivative; u = mm - th
                const float a_u = a_sp * surrogate_grad(q.u);
                a_mm += a_u;
                b_th += -a_u;
                // mm = m*decay*(1-r) + x*gate
                const float b_m = a_mm * p.decay * (1.0f - q.r);
                g_decay += a_mm * q.m * (1.0f - q.r);
                b_r += -a_mm * q.m * p.decay;
                const float a_x = a_mm * q.gate;
                a_m = b_m; a_r = b_r; a_th = b_th;
                a_sp = a_x;                               // x_t = sp_{t-1}  (t >= 1);  for t = 0 it is dL/dx
            }
            g_theta += a_th;                              // the initial threshold IS theta0
            gx[row * ch + c] = a_sp;
        }
    }
    red[slab][0][lane] = g_decay;
    red[slab][1][lane] = g_adapt;
    red[slab][2][lane] = g_rdecay;
    red[slab][3][lane] = g_theta;
    __syncthreads();
    if (slab == 0 && live) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float s = ((red[0][q][lane] + red[1][q][lane]) + red[2][q][lane]) + red[3][q][lane];
            partial[((int64_t)blockIdx.x * 4 + q) * ch + c] = s;
        }
    }
}

// sums the per-workgroup partials and applies the clamp masks of the raw parameters.  Workgroup = 64 channels x 4 lanes:
// lane g adds partials g, g+4, ... in ascending order, the four lane sums are added in the order g = 0..3 (deterministic).
__global__ __launch_bounds__(256) void lif_train_param_reduce_kernel(const float* __restrict__ partial, int64_t nblocks, int ch,
                                                                     const float* md, const float* ta, const float* rd,
                                                                     float* __restrict__ g_md, float* __restrict__ g_ta,
                                                                     float* __restrict__ g_rd, float* __restrict__ g_tb) {
    __shared__ float red[4][4][64];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const bool live = c < ch;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    if (live) {
        for (int64_t b = g; b < nblocks; b += 16) {
            float v[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) v[i][q] = (b + 4 * i < nblocks) ? partial[((b + 4 * i) * 4 + q) * ch + c] : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) s[q] += v[i][q];
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) red[g][q][lane] = s[q];
    __syncthreads();
    if (g == 0 && live) {
#pragma unroll
        for (int q = 0; q < 4; ++q) s[q] = ((red[0][q][lane] + red[1][q][lane]) + red[2][q][lane]) + red[3][q][lane];
        g_md[c] = (md[c] >= 0.1f && md[c] <= 0.99f) ? s[0] : 0.f;
        g_ta[c] = (ta[c] >= 0.001f && ta[c] <= 0.1f) ? s[1] : 0.f;
        g_rd[c] = (rd[c] >= 0.1f && rd[c] <= 0.95f) ? s[2] : 0.f;
        g_tb[c] = s[3];
    }
}

// =============================================================================================
// Layer pieces of the training step: BatchNorm in training mode (batch statistics over the rows of a [rows, C] tensor —
// what nn.BatchNorm1d/2d do to a 1x1 convolution's output, fn/snn_coder.py:225-252), its backward, and the weight /
// bias gradients of the 1x1 convolution.  All column reductions are deterministic: fixed-order partial sums per
// workgroup (f64), then a fixed-order sum of the partials.
// =============================================================================================
constexpr int CR_ROWS = 256;     // rows per workgroup of the column reductions

// partial[b][q][c] (f64), q = 0: sum of u(row,c), q = 1: sum of u*v.   MODE 0: u = a, v = a (sum, sum of squares)
//                                                                    MODE 1: u = a (= dz), v = (b - mean) * invstd (= y_hat)
// Workgroup = 64 channels x 4 row lanes: thread (g = tid >> 6, lane = tid & 63) sums rows r0+g, r0+g+4, ... of its channel
// (8 loads in flight per round), the 4 lanes' sums are added in the order g = 0..3 through LDS.
template <int MODE>
__global__ __launch_bounds__(256) void col_partial_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t rows,
                                                          int ch, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                          double* __restrict__ partial) {
    __shared__ double red[2][4][64];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + lane;
    const bool live = c < ch;
    const int64_t r0 = (int64_t)blockIdx.x * CR_ROWS;
    const int64_t r1 = r0 + CR_ROWS < rows ? r0 + CR_ROWS : rows;
    double s0 = 0.0, s1 = 0.0;
    if (live) {
        const float mu = MODE == 1 ? mean[c] : 0.f, is = MODE == 1 ? invstd[c] : 0.f;
        for (int64_t r = r0 + g; r < r1; r += 32) {
            float u[8], w[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int64_t rr = r + 4 * i;
                u[i] = rr < r1 ? a[rr * ch + c] : 0.f;
                w[i] = (MODE == 1 && rr < r1) ? b[rr * ch + c] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (r + 4 * i >= r1) break;
                const float v = MODE == 0 ? u[i] : (w[i] - mu) * is;
                s0 += (double)u[i];
                s1 += (double)u[i] * (double)v;
            }
        }
    }
    red[0][g][lane] = s0;
    red[1][g][lane] = s1;
    __syncthreads();
    if (g == 0 && live) {
        partial[((int64_t)blockIdx.x * 2 + 0) * ch + c] = ((red[0][0][lane] + red[0][1][lane]) + red[0][2][lane]) + red[0][3][lane];
        partial[((int64_t)blockIdx.x * 2 + 1) * ch + c] = ((red[1][0][lane] + red[1][1][lane]) + red[1][2][lane]) + red[1][3][lane];
    }
}

__global__ __launch_bounds__(256) void col_final_kernel(const double* __restrict__ partial, int64_t nb, int ch,
                                                        double* __restrict__ sums /*[2][ch]*/) {
    __shared__ double red[2][4][64];                     // 64 channels x 4 lanes, lane sums combined in the order g = 0..3
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const bool live = c < ch;
    double s0 = 0.0, s1 = 0.0;
    if (live) {
        for (int64_t b = g; b < nb; b += 16) {
            double u[4], w[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool ok = b + 4 * i < nb;
                u[i] = ok ? partial[((b + 4 * i) * 2 + 0) * ch + c] : 0.0;
                w[i] = ok ? partial[((b + 4 * i) * 2 + 1) * ch + c] : 0.0;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                s0 += u[i];
                s1 += w[i];
            }
        }
    }
    red[0][g][lane] = s0;
    red[1][g][lane] = s1;
    __syncthreads();
    if (g == 0 && live) {
        sums[c] = ((red[0][0][lane] + red[0][1][lane]) + red[0][2][lane]) + red[0][3][lane];
        sums[ch + c] = ((red[1][0][lane] + red[1][1][lane]) + red[1][2][lane]) + red[1][3][lane];
    }
}

__global__ __launch_bounds__(256) void bn_train_apply_kernel(const float* __restrict__ y, int64_t rows, int ch,
                                                             const double* __restrict__ sums, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float eps, float* __restrict__ z,
                                                             float* __restrict__ mean_out, float* __restrict__ var_out,
                                                             float* __restrict__ invstd_out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= rows * ch) return;
    const int c = (int)(t % ch);
    const double mu = sums[c] / (double)rows;
    double var = sums[ch + c] / (double)rows - mu * mu;       // biased variance: what training-mode normalisation uses
    if (var < 0.0) var = 0.0;
    const float is = (float)(1.0 / sqrt(var + (double)eps));
    if (t < ch) {
        mean_out[c] = (float)mu;
        var_out[c] = (float)var;
        invstd_out[c] = is;
    }
    z[t] = (y[t] - (float)mu) * is * gamma[c] + beta[c];
}

// dy = gamma * invstd * (dz - sum(dz)/R - y_hat * sum(dz*y_hat)/R);  dgamma = sum(dz*y_hat), dbeta = sum(dz)
__global__ __launch_bounds__(256) void bn_train_bwd_apply_kernel(const float* __restrict__ y, const float* __restrict__ dz,
                                                                 int64_t rows, int ch, const double* __restrict__ sums,
                                                                 const float* __restrict__ gamma, const float* __restrict__ mean,
                                                                 const float* __restrict__ invstd, float* __restrict__ dy,
                                                                 float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= rows * ch) return;
    const int c = (int)(t % ch);
    const float sdz = (float)(sums[c] / (double)rows), sdzy = (float)(sums[ch + c] / (double)rows);
    if (t < ch) {
        dbeta[c] = (float)sums[c];
        dgamma[c] = (float)sums[ch + c];
    }
    const float yh = (y[t] - mean[c]) * invstd[c];
    dy[t] = gamma[c] * invstd[c] * (dz[t] - sdz - yh * sdzy);
}

// ---- weight gradient of a 1x1 convolution: dW[n,k] = sum_r dY[r,n] * X[r,k]  (exact-f32 MFMA, 2 rows per instruction).
// Workgroup = 4 waves = one 64x64 tile of dW over one slab of rows; wave (wn, wk) owns a 32x32 block: lane l supplies
// dY[r + (l>>5)][n0 + wn*32 + (l&31)] and X[r + (l>>5)][k0 + wk*32 + (l&31)] — row-major tensors are already in the
// operand layout of v_mfma_f32_32x32x2_f32 for this product, no transposition.  Partials [slab][n][k] are summed in slab
// order by wgrad_reduce_kernel (deterministic).  First version: operands come straight from global memory (L2 absorbs the
// n/64-fold and k/64-fold re-reads); an LDS-staged / larger-tile version is future work.
typedef float f32x16t __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void wgrad_partial_kernel(const float* __restrict__ dy, int ldy, const float* __restrict__ x,
                                                            int ldx, int64_t rows, int n, int k, int64_t rows_per_slab,
                                                            float* __restrict__ partial /*[slabs][n][k]*/) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wn = wave >> 1, wk = wave & 1;
    const int nn = blockIdx.x * 64 + wn * 32 + (lane & 31);
    const int kc = blockIdx.y * 64 + wk * 32 + (lane & 31);
    const int par = lane >> 5;
    const int64_t r0 = (int64_t)blockIdx.z * rows_per_slab;
    const int64_t r1 = r0 + rows_per_slab < rows ? r0 + rows_per_slab : rows;
    f32x16t acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const bool nok = nn < n, kok = kc < k;
    for (int64_t r = r0; r < r1; r += 16) {               // 8 row pairs per round: 16 loads in flight ahead of the 8 MFMAs
        float av[8], bv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int64_t rr = r + 2 * i + par;
            const bool rok = rr < r1;
            av[i] = (rok && nok) ? dy[rr * ldy + nn] : 0.f;
            bv[i] = (rok && kok) ? x[rr * ldx + kc] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[i], acc, 0, 0, 0);
    }
    // accumulator layout: lane = column (k index within the block) + 32 * h, register e = row (e&3) + 8*(e>>2) + 4h (n index)
    float* out = partial + (int64_t)blockIdx.z * n * k;
    const int col = blockIdx.y * 64 + wk * 32 + (lane & 31);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int row = blockIdx.x * 64 + wn * 32 + (e & 3) + 8 * (e >> 2) + 4 * par;
        if (row < n && col < k) out[(int64_t)row * k + col] = acc[e];
    }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, int slabs, int64_t count,
                                                           float* __restrict__ dw) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    float s = 0.f;
    for (int b = 0; b < slabs; ++b) s += partial[(int64_t)b * count + t];
    dw[t] = s;
}

__global__ __launch_bounds__(256) void colsum_to_float_kernel(const double* __restrict__ sums, int ch, float* __restrict__ out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < ch) out[c] = (float)sums[c];
}

static int64_t col_blocks(int64_t rows) { return rows > 0 ? (rows + CR_ROWS - 1) / CR_ROWS : 0; }
static int wgrad_slabs(int64_t rows) {                       // ~1024 rows per slab: enough workgroups at training batch sizes
    int64_t s = (rows + 1023) / 1024;
    return (int)(s < 1 ? 1 : (s > 64 ? 64 : s));
}

// ---- backward of the per-channel softmax over the k neighbours + weighted aggregation (fn/snn_coder.py:379-389):
//   forward (fn_softmax_agg_kernel): w = softmax_j(a_j / sqrt(hd)), u_j = v[nbr_j] + pe_j, res = sum_j w_j u_j
//   backward, g = d res:  d pe_j = w_j g;  d v[nbr_j] += w_j g (scatter-add);  d a_j = w_j g (u_j - res) / sqrt(hd)
// thread per (point, channel); the softmax is recomputed from `a` with the forward kernel's arithmetic.  grad_v is NOT written
// here: its contributions are exactly the d pe_j values, which scatter_sum_grouped_kernel sums per neighbour row in a fixed
// order afterwards (round 4; float atomics until then — the one run-to-run difference of a training step).
__global__ __launch_bounds__(256) void softmax_agg_bwd_kernel(const float* __restrict__ a, const float* __restrict__ pe,
                                                              const float* __restrict__ v, int ldv, const int32_t* __restrict__ idx,
                                                              const float* __restrict__ gres, int64_t pts, int m, int kk, int d,
                                                              float sqrt_hd, const float* __restrict__ keep,
                                                              float* __restrict__ ga, float* __restrict__ gpe) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= pts * d) return;
    const int c = (int)(t % d);
    const int64_t pt = t / d;
    const int64_t patch_i = pt / m;
    const float* ar = a + pt * kk * d + c;
    const float* pr = pe + pt * kk * d + c;
    const float* kr = keep ? keep + pt * kk * d + c : nullptr;
    const int32_t* ir = idx + pt * kk;
    const float inv_sqrt_hd = __fdiv_rn(1.0f, sqrt_hd);
    float mx = -__builtin_huge_valf();
    for (int j = 0; j < kk; ++j) mx = fmaxf(mx, __fmul_rn(ar[(int64_t)j * d], inv_sqrt_hd));
    float den = 0.f;
    for (int j = 0; j < kk; ++j) den = __fadd_rn(den, fast_exp(__fsub_rn(__fmul_rn(ar[(int64_t)j * d], inv_sqrt_hd), mx)));
    const float inv_den = __fdiv_rn(1.0f, den);
    float res = 0.f;
    for (int j = 0; j < kk; ++j) {
        float wj = __fmul_rn(fast_exp(__fsub_rn(__fmul_rn(ar[(int64_t)j * d], inv_sqrt_hd), mx)), inv_den);
        if (kr) wj = __fmul_rn(wj, kr[(int64_t)j * d]);
        res = __fmaf_rn(wj, __fadd_rn(v[(patch_i * m + ir[j]) * ldv + c], pr[(int64_t)j * d]), res);
    }
    const float g = gres[t];
    for (int j = 0; j < kk; ++j) {
        const float wj = __fmul_rn(fast_exp(__fsub_rn(__fmul_rn(ar[(int64_t)j * d], inv_sqrt_hd), mx)), inv_den);
        const float kj = kr ? kr[(int64_t)j * d] : 1.0f;                    // dropout on the softmax weights (fn:383)
        const int64_t nb = patch_i * m + ir[j];
        const float u = __fadd_rn(v[nb * ldv + c], pr[(int64_t)j * d]);
        const float wg = wj * kj * g;
        gpe[(pt * kk + j) * d + c] = wg;
        ga[(pt * kk + j) * d + c] = wj * g * (kj * u - res) * inv_sqrt_hd;
    }
}

// Forward with dropout on the softmax weights (train() mode, fn/snn_coder.py:381-388): keep holds 0 or 1/(1-p) per
// (edge row, channel).  Same arithmetic as the backward's recomputation above.
__global__ __launch_bounds__(256) void softmax_agg_keep_fwd_kernel(const float* __restrict__ a, const float* __restrict__ pe,
                                                                   const float* __restrict__ v, int ldv, const int32_t* __restrict__ idx,
                                                                   const float* __restrict__ keep, int64_t pts, int m, int kk, int d,
                                                                   float sqrt_hd, float* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= pts * d) return;
    const int c = (int)(t % d);
    const int64_t pt = t / d;
    const int64_t patch_i = pt / m;
    const float* ar = a + pt * kk * d + c;
    const float* pr = pe + pt * kk * d + c;
    const float* kr = keep + pt * kk * d + c;
    const int32_t* ir = idx + pt * kk;
    const float inv_sqrt_hd = __fdiv_rn(1.0f, sqrt_hd);
    float mx = -__builtin_huge_valf();
    for (int j = 0; j < kk; ++j) mx = fmaxf(mx, __fmul_rn(ar[(int64_t)j * d], inv_sqrt_hd));
    float den = 0.f;
    for (int j = 0; j < kk; ++j) den = __fadd_rn(den, fast_exp(__fsub_rn(__fmul_rn(ar[(int64_t)j * d], inv_sqrt_hd), mx)));
    const float inv_den = __fdiv_rn(1.0f, den);
    float res = 0.f;
    for (int j = 0; j < kk; ++j) {
        const float wj = __fmul_rn(__fmul_rn(fast_exp(__fsub_rn(__fmul_rn(ar[(int64_t)j * d], inv_sqrt_hd), mx)), inv_den), kr[(int64_t)j * d]);
        res = __fmaf_rn(wj, __fadd_rn(v[(patch_i * m + ir[j]) * ldv + c], pr[(int64_t)j * d]), res);
    }
    out[t] = res;
}

// zero the first d columns of `rows` rows (pitch ld floats).  A kernel rather than hipMemset2DAsync: inside a captured HIP graph
// the 2-D memset did not take effect on replay (the scatter-add targets then started from whatever the pool block held).
__global__ __launch_bounds__(256) void zero_rows_kernel(float* __restrict__ p, int64_t rows, int ld, int d) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= rows * d) return;
    const int64_t r = t / d;
    p[r * ld + (int)(t - r * d)] = 0.f;
}

static int launch_zero_rows(float* p, int64_t rows, int ld, int d, hipStream_t st) {
    if (rows <= 0) return SAPCU_OK;
    hipLaunchKernelGGL(zero_rows_kernel, dim3((unsigned)((rows * d + 255) / 256)), dim3(256), 0, st, p, rows, ld, d);
    SAPCU_CHECK_LAUNCH();
    return SAPCU_OK;
}

// ---- row gather (index_points, fn/snn_coder.py:19-29, on flattened rows) and its backward (scatter-add)
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, int lds_, const int64_t* __restrict__ index,
                                                          int64_t rows, int d, float* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= rows * d) return;
    const int64_t r = t / d;
    const int c = (int)(t - r * d);
    out[t] = src[index[r] * lds_ + c];
}

__global__ __launch_bounds__(256) void scatter_add_rows_kernel(const float* __restrict__ gout, const int64_t* __restrict__ index,
                                                               int64_t rows, int d, float* __restrict__ gsrc, int ldg) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= rows * d) return;
    const int64_t r = t / d;
    const int c = (int)(t - r * d);
    atomicAdd(gsrc + index[r] * ldg + c, gout[t]);
}

// ---- deterministic scatter-add for the patch-structured indices of the training step (round 4).  Every index tensor the step
// scatters through is grouped: the gr source rows of group g (the m k edge rows of a patch) point into the gs destination rows
// of the SAME group (the patch's m points).  One workgroup per group: the group's destinations (as local ids) go to LDS, the
// INVERSE table is built there — destination t's sources in ascending source order, by one thread per destination scanning the
// gr entries twice (count, then fill behind an exclusive prefix of the counts) — and every (destination, channel) is then summed
// by one thread over its sources in that order and STORED: no atomics, no zeroing, the same bits every run.  (The float
// atomicAdd form summed in arrival order: the last bits of the gradients, and with hard spikes whole trajectories, differed
// from run to run.)  LOCAL = true: index32 holds in-patch ids (softmax-aggregate's neighbour table); false: index64 holds global
// rows, local id = index - g gs.  Entries outside [0, gs) are skipped and counted in *bad (may be null).
// =============================================================================================
template <bool LOCAL>
__global__ __launch_bounds__(256) void scatter_sum_grouped_kernel(const float* __restrict__ gout, const int32_t* __restrict__ index32,
                                                                  const int64_t* __restrict__ index64, int gs, int gr, int d,
                                                                  float* __restrict__ gsrc, int ldg, int* __restrict__ bad) {
    extern __shared__ unsigned char sg_smem[];
    int* dst = reinterpret_cast<int*>(sg_smem);                 // [gr] local destination of each source row (-1: none)
    int* lst = dst + gr;                                        // [gr] source rows ordered by (destination, source)
    int* off = lst + gr;                                        // [gs + 1] start of each destination's list
    const int64_t g = blockIdx.x;
    const int tid = threadIdx.x;
    int nbad = 0;
    for (int e = tid; e < gr; e += 256) {
        const int64_t v = LOCAL ? (int64_t)index32[g * gr + e] : index64[g * gr + e] - g * gs;
        const bool ok = v >= 0 && v < gs;
        dst[e] = ok ? (int)v : -1;
        nbad += ok ? 0 : 1;
    }
    if (nbad && bad) atomicAdd(bad, nbad);
    __syncthreads();
    for (int t = tid; t < gs; t += 256) {                       // count
        int c = 0;
        for (int e = 0; e < gr; ++e) c += dst[e] == t;
        off[t + 1] = c;
    }
    __syncthreads();
    if (tid == 0) {                                             // exclusive prefix (gs <= a few hundred)
        off[0] = 0;
        for (int t = 0; t < gs; ++t) off[t + 1] += off[t];
    }
    __syncthreads();
    for (int t = tid; t < gs; t += 256) {                       // fill, ascending source order
        int w = off[t];
        for (int e = 0; e < gr; ++e)
            if (dst[e] == t) lst[w++] = e;
    }
    __syncthreads();
    const float* gb = gout + g * gr * (int64_t)d;
    float* ob = gsrc + g * gs * (int64_t)ldg;
    for (int64_t q = tid; q < (int64_t)gs * d; q += 256) {
        const int t = (int)(q / d), c = (int)(q - (int64_t)t * d);
        float s = 0.f;
        for (int w = off[t]; w < off[t + 1]; ++w) s = __fadd_rn(s, gb[(int64_t)lst[w] * d + c]);
        ob[(int64_t)t * ldg + c] = s;
    }
}

static int launch_scatter_sum_grouped(const float* gout, const int32_t* i32, const int64_t* i64, int64_t groups, int gs, int gr, int d,
                                      float* gsrc, int ldg, int* bad, hipStream_t st) {
    if (groups == 0) return SAPCU_OK;
    const size_t lds = ((size_t)2 * gr + gs + 1) * sizeof(int);
    SAPCU_CHECK_ARG(lds <= 64 * 1024 && groups < 0x7fffffffLL, "scatter_sum_grouped: group too large (%d sources, %d destinations)", gr, gs);
    if (i32)
        hipLaunchKernelGGL(scatter_sum_grouped_kernel<true>, dim3((unsigned)groups), dim3(256), lds, st, gout, i32, i64, gs, gr, d, gsrc, ldg, bad);
    else
        hipLaunchKernelGGL(scatter_sum_grouped_kernel<false>, dim3((unsigned)groups), dim3(256), lds, st, gout, i32, i64, gs, gr, d, gsrc, ldg, bad);
    SAPCU_CHECK_LAUNCH();
    return SAPCU_OK;
}

// ---- max over the points of a patch (adaptive_max_pool1d(., 1), fn/snn_coder.py:472) with the arg-max kept for the backward;
// ties go to the FIRST point (what torch's pooling does) — with hard 0/1 spikes ties are the rule, so the tie rule decides
// where the gradient goes.
__global__ __launch_bounds__(256) void group_max_fwd_kernel(const float* __restrict__ x, int64_t groups, int m, int c,
                                                            float* __restrict__ out, int32_t* __restrict__ arg) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= groups * c) return;
    const int cc = (int)(t % c);
    const int64_t g = t / c;
    const float* p = x + g * m * c + cc;
    float mx = p[0];
    int am = 0;
    for (int i = 1; i < m; ++i) {
        const float v = p[(int64_t)i * c];
        if (v > mx) { mx = v; am = i; }
    }
    out[t] = mx;
    arg[t] = am;
}

__global__ __launch_bounds__(256) void group_max_bwd_kernel(const float* __restrict__ gout, const int32_t* __restrict__ arg,
                                                            int64_t groups, int m, int c, float* __restrict__ gx) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;     // over [groups*m, c]
    if (t >= groups * m * c) return;
    const int cc = (int)(t % c);
    const int64_t row = t / c;
    const int64_t g = row / m;
    const int i = (int)(row - g * m);
    gx[t] = arg[g * c + cc] == i ? gout[g * c + cc] : 0.f;
}

}  // namespace sapcu

using namespace sapcu;

extern "C" {

int sapcu_lif_train_forward(const float* x, int64_t rows, int channels, int steps, const float* membrane_decay,
                            const float* threshold_adapt, const float* refractory_decay, const float* threshold_base,
                            float* spikes_out, void* stream) {
    SAPCU_CHECK_ARG(x && membrane_decay && threshold_adapt && refractory_decay && threshold_base && spikes_out,
                    "lif_train_forward: null pointer");
    SAPCU_CHECK_ARG(rows >= 0 && channels >= 1 && steps >= 1 && steps <= LT_MAX_T, "lif_train_forward: need 1 <= steps <= %d", LT_MAX_T);
    const int64_t total = rows * channels;
    if (total == 0) return SAPCU_OK;
#define SAPCU_LT_FWD(TT)                                                                                                       \
    case TT:                                                                                                                   \
        hipLaunchKernelGGL(lif_train_fwd_kernel<TT>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, \
                           rows, channels, membrane_decay, threshold_adapt, refractory_decay, threshold_base, spikes_out);     \
        break;
    switch (steps) {
        SAPCU_LT_FWD(1) SAPCU_LT_FWD(2) SAPCU_LT_FWD(3) SAPCU_LT_FWD(4) SAPCU_LT_FWD(5) SAPCU_LT_FWD(6) SAPCU_LT_FWD(7) SAPCU_LT_FWD(8)
    }
#undef SAPCU_LT_FWD
    SAPCU_CHECK_LAUNCH();
    return SAPCU_OK;
}

int64_t sapcu_lif_train_workspace_bytes(int64_t rows, int channels) {
    if (rows < 0 || channels < 1) return -1;
    const int64_t nb = (rows + LT_ROWS_PER_WG - 1) / LT_ROWS_PER_WG;
    return (nb > 0 ? nb : 1) * 4 * (int64_t)channels * (int64_t)sizeof(float);
}

int sapcu_lif_train_backward(const float* x, const float* grad_spikes, int64_t rows, int channels, int steps,
                             const float* membrane_decay, const float* threshold_adapt, const float* refractory_decay,
                             const float* threshold_base, float* grad_x, float* grad_membrane_decay,
                             float* grad_threshold_adapt, float* grad_refractory_decay, float* grad_threshold_base,
                             void* workspace, int64_t workspace_bytes, void* stream) {
    SAPCU_CHECK_ARG(x && grad_spikes && membrane_decay && threshold_adapt && refractory_decay && threshold_base && grad_x &&
                        grad_membrane_decay && grad_threshold_adapt && grad_refractory_decay && grad_threshold_base && workspace,
                    "lif_train_backward: null pointer");
    SAPCU_CHECK_ARG(rows >= 0 && channels >= 1 && steps >= 1 && steps <= LT_MAX_T, "lif_train_backward: need 1 <= steps <= %d", LT_MAX_T);
    if (workspace_bytes < sapcu_lif_train_workspace_bytes(rows, channels)) {
        set_error("lif_train_backward: workspace of %lld bytes, need %lld", (long long)workspace_bytes,
                  (long long)sapcu_lif_train_workspace_bytes(rows, channels));
        return SAPCU_ERR_WORKSPACE;
    }
    const int64_t nb = (rows + LT_ROWS_PER_WG - 1) / LT_ROWS_PER_WG;
    if (nb > 0) {
        SAPCU_CHECK_ARG(nb < 0x7fffffffLL, "lif_train_backward: too many rows");
#define SAPCU_LT_BWD(TT)                                                                                                       \
    case TT:                                                                                                                   \
        hipLaunchKernelGGL(lif_train_bwd_kernel<TT>, dim3((unsigned)nb, (unsigned)((channels + 63) / 64)), dim3(256), 0,           \
                           (hipStream_t)stream, x, grad_spikes, rows, channels, membrane_decay, threshold_adapt,               \
                           refractory_decay, threshold_base, grad_x, (float*)workspace);                                       \
        break;
        switch (steps) {
            SAPCU_LT_BWD(1) SAPCU_LT_BWD(2) SAPCU_LT_BWD(3) SAPCU_LT_BWD(4) SAPCU_LT_BWD(5) SAPCU_LT_BWD(6) SAPCU_LT_BWD(7) SAPCU_LT_BWD(8)
        }
#undef SAPCU_LT_BWD
        SAPCU_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(lif_train_param_reduce_kernel, dim3((unsigned)((channels + 63) / 64)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)workspace, nb, channels, membrane_decay, threshold_adapt, refractory_decay,
                       grad_membrane_decay, grad_threshold_adapt, grad_refractory_decay, grad_threshold_base);
    SAPCU_CHECK_LAUNCH();
    return SAPCU_OK;
}


int64_t sapcu_train_workspace_bytes(int64_t rows, int channels, int k) {
    if (rows < 0 || channels < 1 || k < 0) return -1;
    const int64_t col = (col_blocks(rows) * 2 + 2) * (int64_t)channels * (int64_t)sizeof(double);
    const int64_t wg = (int64_t)wgrad_slabs(rows) * channels * (int64_t)(k > 0 ? k : 1) * (int64_t)sizeof(float);
    return (col > wg ? col : wg) + 256;
}

static int column_sums(int mode, const float* a, const float* b, int64_t rows, int ch, const float* mean, const float* invstd,
                       double* ws, double** sums_out, hipStream_t st) {
    const int64_t nb = col_blocks(rows);
    double* partial = ws;
    double* sums = ws + nb * 2 * (int64_t)ch;
    const dim3 grid((unsigned)(nb > 0 ? nb : 1), (unsigned)((ch + 63) / 64));
    if (nb > 0) {
        if (mode == 0) hipLaunchKernelGGL(col_partial_kernel<0>, grid, dim3(256), 0, st, a, b, rows, ch, mean, invstd, partial);
        else hipLaunchKernelGGL(col_partial_kernel<1>, grid, dim3(256), 0, st, a, b, rows, ch, mean, invstd, partial);
        SAPCU_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(col_final_kernel, dim3((unsigned)((ch + 63) / 64)), dim3(256), 0, st, partial, nb, ch, sums);
    SAPCU_CHECK_LAUNCH();
    *sums_out = sums;
    return SAPCU_OK;
}

int sapcu_bn_train_forward(const float* y, int64_t rows, int channels, const float* gamma, const float* beta, float eps,
                           float* z_out, float* mean_out, float* var_out, float* invstd_out, void* workspace,
                           int64_t workspace_bytes, void* stream) {
    SAPCU_CHECK_ARG(y && gamma && beta && z_out && mean_out && var_out && invstd_out && workspace && rows >= 1 && channels >= 1,
                    "bn_train_forward: bad argument");
    SAPCU_CHECK_ARG(workspace_bytes >= sapcu_train_workspace_bytes(rows, channels, 0), "bn_train_forward: workspace too small");
    double* sums = nullptr;
    int rc = column_sums(0, y, nullptr, rows, channels, nullptr, nullptr, (double*)workspace, &sums, (hipStream_t)stream);
    if (rc != SAPCU_OK) return rc;
    const int64_t total = rows * channels;
    hipLaunchKernelGGL(bn_train_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, y, rows,
                       channels, sums, gamma, beta, eps, z_out, mean_out, var_out, invstd_out);
    SAPCU_CHECK_LAUNCH();
    return SAPCU_OK;
}

int sapcu_bn_train_backward(const float* y, const float* grad_z, int64_t rows, int channels, const float* gamma,
                            const float* mean, const float* invstd, float* grad_y, float* grad_gamma, float* grad_beta,
                            void* workspace, int64_t workspace_bytes, void* stream) {
    SAPCU_CHECK_ARG(y && grad_z && gamma && mean && invstd && grad_y && grad_gamma && grad_beta && workspace && rows >= 1 &&
                        channels >= 1, "bn_train_backward: bad argument");
    SAPCU_CHECK_ARG(workspace_bytes >= sapcu_train_workspace_bytes(rows, channels, 0), "bn_train_backward: workspace too small");
    double* sums = nullptr;
    int rc = column_sums(1, grad_z, y, rows, channels, mean, invstd, (double*)workspace, &sums, (hipStream_t)stream);
    if (rc != SAPCU_OK) return rc;
    const int64_t total = rows * channels;
    hipLaunchKernelGGL(bn_train_bwd_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, y, grad_z,
                       rows, channels, sums, gamma, mean, invstd, grad_y, grad_gamma, grad_beta);
    SAPCU_CHECK_LAUNCH();
    return SAPCU_OK;
}

int sapcu_conv1x1_wgrad_f32(const float* grad_y, int ldy, const float* x, int ldx, int64_t rows, int n, int k, float* grad_w,
                            float* grad_bias, void* workspace, int64_t workspace_bytes, void* stream) {
    SAPCU_CHECK_ARG(grad_y && x && grad_w && rows >= 0 && n >= 1 && k >= 1 && ldy >= n && ldx >= k, "conv1x1_wgrad: bad argument");
    if (rows == 0) {        // an empty batch: zero gradients (the forward GEMMs accept r == 0 as well)
        SAPCU_CHECK_HIP(hipMemsetAsync(grad_w, 0, (size_t)n * k * sizeof(float), (hipStream_t)stream));
        if (grad_bias) SAPCU_CHECK_HIP(hipMemsetAsync(grad_bias, 0, (size_t)n * sizeof(float), (hipStream_t)stream));
        return SAPCU_OK;
    }
    SAPCU_CHECK_ARG(workspace != nullptr, "conv1x1_wgrad: null workspace");
    SAPCU_CHECK_ARG(workspace_bytes >= sapcu_train_workspace_bytes(rows, n, k), "conv1x1_wgrad: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    if (grad_bias) {                                   // db = column sums of dY (needs a dense dY for the shared reduction)
        SAPCU_CHECK_ARG(ldy == n, "conv1x1_wgrad: the bias gradient needs ldy == n");
        double* sums = nullptr;
        int rc = column_sums(0, grad_y, nullptr, rows, n, nullptr, nullptr, (double*)workspace, &sums, st);
        if (rc != SAPCU_OK) return rc;
        hipLaunchKernelGGL(colsum_to_float_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, sums, n, grad_bias);
        SAPCU_CHECK_LAUNCH();
    }
    const int slabs = wgrad_slabs(rows);
    const int64_t rps = (((rows + slabs - 1) / slabs) + 1) & ~(int64_t)1;      // even: a slab's row pairs stay inside it
    hipLaunchKernelGGL(wgrad_partial_kernel, dim3((unsigned)((n + 63) / 64), (unsigned)((k + 63) / 64), (unsigned)slabs), dim3(256), 0,
                       st, grad_y, ldy, x, ldx, rows, n, k, rps, (float*)workspace);
    SAPCU_CHECK_LAUNCH();
    const int64_t count = (int64_t)n * k;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, (const float*)workspace, slabs,
                       count, grad_w);
    SAPCU_CHECK_LAUNCH();
    return SAPCU_OK;
}


int sapcu_softmax_agg_forward(const float* a, const float* pe, const float* v, int ldv, const int32_t* idx, const float* keep,
                              int64_t pts, int m, int kk, int d, float sqrt_hd, float* res, void* stream) {
    SAPCU_CHECK_ARG(a && pe && v && idx && res && pts >= 0 && m >= 1 && kk >= 1 && d >= 1 && ldv >= d && sqrt_hd > 0.f,
                    "softmax_agg_forward: bad argument");
    if (!keep) return launch_fn_softmax_agg(a, pe, v, ldv, idx, pts, m, kk, d, sqrt_hd, res, 0, (hipStream_t)stream);
    if (pts == 0) return SAPCU_OK;
    SAPCU_CHECK_ARG(pts % m == 0, "softmax_agg_forward: points must come in whole patches of m");
    hipLaunchKernelGGL(softmax_agg_keep_fwd_kernel, dim3((unsigned)((pts * d + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, pe, v, ldv,
                       idx, keep, pts, m, kk, d, sqrt_hd, res);
    SAPCU_CHECK_LAUNCH();
    return SAPCU_OK;
}

int sapcu_softmax_agg_backward(const float* a, const float* pe, const float* v, int ldv, const int32_t* idx, const float* keep,
                               const float* grad_res, int64_t pts, int m, int kk, int d, float sqrt_hd, float* grad_a,
                               float* grad_pe, float* grad_v, int ldgv, void* stream) {
    SAPCU_CHECK_ARG(a && pe && v && idx && grad_res && grad_a && grad_pe && grad_v && pts >= 0 && m >= 1 && kk >= 1 && d >= 1 &&
                        ldv >= d && ldgv >= d && sqrt_hd > 0.f, "softmax_agg_backward: bad argument");
    if (pts == 0) return SAPCU_OK;
    SAPCU_CHECK_ARG(pts % m == 0, "softmax_agg_backward: points must come in whole patches of m");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(softmax_agg_bwd_kernel, dim3((unsigned)((pts * d + 255) / 256)), dim3(256), 0, st, a, pe, v, ldv, idx, grad_res,
                       pts, m, kk, d, sqrt_hd, keep, grad_a, grad_pe);
    SAPCU_CHECK_LAUNCH();
    // grad_v[nbr] = sum over the (point, j) edges that list nbr of d pe_j, in ascending edge order (deterministic)
    return launch_scatter_sum_grouped(grad_pe, idx, nullptr, pts / m, m, m * kk, d, grad_v, ldgv, nullptr, st);
}


int sapcu_gather_rows(const float* src, int ld_src, const int64_t* index, int64_t rows, int d, float* out, void* stream) {
    SAPCU_CHECK_ARG(src && index && out && rows >= 0 && d >= 1 && ld_src >= d, "gather_rows: bad argument");
    if (rows == 0) return SAPCU_OK;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((rows * d + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, ld_src, index,
                       rows, d, out);
    SAPCU_CHECK_LAUNCH();
    return SAPCU_OK;
}

int sapcu_scatter_add_rows(const float* grad_out, const int64_t* index, int64_t rows, int d, float* grad_src, int ld_grad,
                           int64_t src_rows, void* stream) {
    SAPCU_CHECK_ARG(grad_out && index && grad_src && rows >= 0 && d >= 1 && ld_grad >= d && src_rows >= 0, "scatter_add_rows: bad argument");
    hipStream_t st = (hipStream_t)stream;
    { const int rc = launch_zero_rows(grad_src, src_rows, ld_grad, d, st); if (rc != SAPCU_OK) return rc; }
    if (rows == 0) return SAPCU_OK;
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3((unsigned)((rows * d + 255) / 256)), dim3(256), 0, st, grad_out, index, rows, d,
                       grad_src, ld_grad);
    SAPCU_CHECK_LAUNCH();
    return SAPCU_OK;
}


int sapcu_scatter_add_rows_grouped(const float* grad_out, const int64_t* index, int64_t rows, int d, float* grad_src, int ld_grad,
                                   int64_t src_rows, int group_src_rows, int group_rows, int* bad_count, void* stream) {
    SAPCU_CHECK_ARG(grad_out && index && grad_src && rows >= 0 && d >= 1 && ld_grad >= d && src_rows >= 0 && group_src_rows >= 1 &&
                        group_rows >= 1, "scatter_add_rows_grouped: bad argument");
    SAPCU_CHECK_ARG(rows % group_rows == 0 && src_rows % group_src_rows == 0 && rows / group_rows == src_rows / group_src_rows,
                    "scatter_add_rows_grouped: %lld rows / %d and %lld destination rows / %d are not the same number of whole groups",
                    (long long)rows, group_rows, (long long)src_rows, group_src_rows);
    return launch_scatter_sum_grouped(grad_out, nullptr, index, rows / group_rows, group_src_rows, group_rows, d, grad_src, ld_grad,
                                      bad_count, (hipStream_t)stream);
}

int sapcu_group_max_forward(const float* x, int64_t groups, int m, int c, float* out, int32_t* argmax_out, void* stream) {
    SAPCU_CHECK_ARG(x && out && argmax_out && groups >= 0 && m >= 1 && c >= 1, "group_max_forward: bad argument");
    if (groups == 0) return SAPCU_OK;
    hipLaunchKernelGGL(group_max_fwd_kernel, dim3((unsigned)((groups * c + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, groups, m, c,
                       out, argmax_out);
    SAPCU_CHECK_LAUNCH();
    return SAPCU_OK;
}

int sapcu_group_max_backward(const float* grad_out, const int32_t* argmax, int64_t groups, int m, int c, float* grad_x, void* stream) {
    SAPCU_CHECK_ARG(grad_out && argmax && grad_x && groups >= 0 && m >= 1 && c >= 1, "group_max_backward: bad argument");
    if (groups == 0) return SAPCU_OK;
    hipLaunchKernelGGL(group_max_bwd_kernel, dim3((unsigned)((groups * m * c + 255) / 256)), dim3(256), 0, (hipStream_t)stream, grad_out,
                       argmax, groups, m, c, grad_x);
    SAPCU_CHECK_LAUNCH();
    return SAPCU_OK;
}

}  // extern "C"
