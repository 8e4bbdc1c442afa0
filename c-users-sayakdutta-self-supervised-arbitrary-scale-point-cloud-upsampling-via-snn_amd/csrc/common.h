// Internal declarations shared by the HIP translation units of libsapcu_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/sapcu.h"

namespace sapcu {

void set_error(const char* fmt, ...);

#define SAPCU_CHECK_ARG(cond, ...)                      \
    do {                                                \
        if (!(cond)) {                                  \
            ::sapcu::set_error(__VA_ARGS__);            \
            return SAPCU_ERR_ARG;                       \
        }                                               \
    } while (0)

#define SAPCU_CHECK_HIP(expr)                                                            \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            ::sapcu::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),    \
                               __FILE__, __LINE__);                                      \
            return SAPCU_ERR_HIP;                                                        \
        }                                                                                \
    } while (0)

#define SAPCU_CHECK_LAUNCH() SAPCU_CHECK_HIP(hipGetLastError())

// ---------------------------------------------------------------------------------------------
// Neuron arithmetic.  Every product/sum is an explicitly rounded f32 operation in the op order of
// fn/snn_coder.py:125-146 (ATen evaluates each Python operator as its own rounded kernel), so the
// compiler must not contract mul+add into FMA here.
// ---------------------------------------------------------------------------------------------
struct NeuronP {   // clamped per-channel parameters
    float decay, adapt, rdecay, theta0;
    float dT, rh;  // EIF only
};

// v_med3_f32: one instruction; equals fminf(fmaxf(x, lo), hi) for every non-NaN x (lo <= hi)
__device__ __forceinline__ float clampf(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }

__device__ __forceinline__ NeuronP load_lif(const float* __restrict__ p4, int stride, int c) {
    // p4: [4][stride] raw membrane_decay, threshold_adapt, refractory_decay, threshold_base
    NeuronP p;
    p.decay = clampf(p4[c], 0.1f, 0.99f);
    p.adapt = clampf(p4[stride + c], 0.001f, 0.1f);
    p.rdecay = clampf(p4[2 * stride + c], 0.1f, 0.95f);
    p.theta0 = p4[3 * stride + c];
    p.dT = 0.f;
    p.rh = 0.f;
    return p;
}

__device__ __forceinline__ NeuronP load_eif(const float* __restrict__ p6, int stride, int c) {
    NeuronP p = load_lif(p6, stride, c);
    p.dT = clampf(p6[4 * stride + c], 0.1f, 5.0f);
    p.rh = clampf(p6[5 * stride + c], 0.1f, 2.0f);
    return p;
}

// eval-mode spike surrogate: 0.5*N(x) + 0.5*sigmoid(10 x), x clamped to +-10 (fn:135-146).
// Same operation order as the reference; the transcendentals are the hardware v_exp_f32 / v_rcp_f32
// (1 ulp each) instead of libm calls: |error| <= 2e-7 absolute on a value in (0,1), measured against the
// reference vectors in tests/golden/neuron_unit.npz (bar 1e-6).  The two halvings are exact (powers of
// two), so 0.5/sqrt(2 pi) is one constant and 0.5*s + g one FMA — bit-identical to mul, mul, add.
__device__ __forceinline__ float soft_spike(float d) {
    const float x = clampf(d, -10.0f, 10.0f);
    const float g = __fmul_rn(__builtin_amdgcn_exp2f(__fmul_rn(__fmul_rn(x, x), -0.72134752044448170368f)),
                              0.19947114020071633897f);                     // 0.5 * exp(-x^2/2) / sqrt(2 pi)
    const float e = __builtin_amdgcn_exp2f(__fmul_rn(x, -14.426950408889634074f));   // exp(-10 x)
    const float s = __builtin_amdgcn_rcpf(__fadd_rn(1.0f, e));
    return __fmaf_rn(0.5f, s, g);
}

__device__ __forceinline__ float fast_exp(float a) { return __builtin_amdgcn_exp2f(__fmul_rn(a, 1.4426950408889634074f)); }

struct NeuronS {
    float m, th, r;
};

__device__ __forceinline__ NeuronS neuron_init(const NeuronP& p) { return NeuronS{0.f, p.theta0, 0.f}; }

// One step.  EIF adds dT*exp(clamp((m_prev-rh)/(dT+1e-6),+-5)) un-gated (fd:245-252).
template <bool EIF>
__device__ __forceinline__ float neuron_step(float x, NeuronS& s, const NeuronP& p) {
    float extra = 0.f;
    if (EIF) {
        const float a = clampf(__fdiv_rn(__fsub_rn(s.m, p.rh), __fadd_rn(p.dT, 1e-6f)), -5.0f, 5.0f);
        extra = __fmul_rn(p.dT, fast_exp(a));
    }
    const float xin = (s.r <= 0.f) ? x : __fmul_rn(x, 0.f);
    float m = __fadd_rn(__fmul_rn(__fmul_rn(s.m, p.decay), __fsub_rn(1.0f, s.r)), xin);
    if (EIF) m = __fadd_rn(m, extra);
    const float sp = soft_spike(__fsub_rn(m, s.th));
    s.m = __fmul_rn(m, __fsub_rn(1.0f, sp));
    s.r = __fadd_rn(__fmul_rn(s.r, p.rdecay), sp);
    const float th = __fadd_rn(s.th, __fmul_rn(p.adapt, sp));
    s.th = __fadd_rn(p.theta0, __fmul_rn(__fsub_rn(th, p.theta0), 0.95f));
    return sp;
}

// `for t in range(T): x, *st = snn(x, *st)` — spikes fed back as the next input (fn:319-320), LIF.
// State lives in registers for all T steps.  Two exact simplifications of neuron_step<false>:
//   * step 0 starts from m = 0, r = 0:  m = x,  r = s0  (0*decay*(1-0) + x and 0*rdecay + s0, exactly);
//   * for t >= 1 the gate `x * (r <= 0)` is closed — soft_spike() >= 0.199 * 2^-72 > 0, so r > 0 — and the
//     fed-back input contributes exactly +0;  the state updates after the last spike are dead.
// W chains are advanced together (independent chains = VALU ILP for the GEMM consumers / pos-enc kernel).
typedef float f32x2 __attribute__((ext_vector_type(2)));

// two spikes at once: the multiplies/adds are written on 2-vectors so that hipcc emits the packed
// v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 (one instruction, two chains); clamp, exp2 and rcp are per lane.
// Element-wise the arithmetic is exactly soft_spike()'s.
__device__ __forceinline__ f32x2 soft_spike2(f32x2 d) {
    f32x2 x;
    x.x = clampf(d.x, -10.0f, 10.0f);
    x.y = clampf(d.y, -10.0f, 10.0f);
    const f32x2 a = (x * x) * -0.72134752044448170368f;
    const f32x2 b = x * -14.426950408889634074f;
    f32x2 g, e;
    g.x = __builtin_amdgcn_exp2f(a.x);
    g.y = __builtin_amdgcn_exp2f(a.y);
    e.x = __builtin_amdgcn_exp2f(b.x);
    e.y = __builtin_amdgcn_exp2f(b.y);
    g = g * 0.19947114020071633897f;
    const f32x2 den = e + 1.0f;
    f32x2 s;
    s.x = __builtin_amdgcn_rcpf(den.x);
    s.y = __builtin_amdgcn_rcpf(den.y);
    return __builtin_elementwise_fma(f32x2{0.5f, 0.5f}, s, g);
}

template <int W>
__device__ __forceinline__ void lif_selfloop_n(float (&v)[W], const NeuronP& p, int T) {
    if constexpr (W % 2 == 0) {
        // packed form: chains (2u, 2u+1) share every multiply/add instruction
        constexpr int H = W / 2;
        f32x2 m[H], r[H], th[H], s[H];
#pragma unroll
        for (int u = 0; u < H; ++u) {
            m[u] = f32x2{v[2 * u], v[2 * u + 1]};
            s[u] = soft_spike2(m[u] - p.theta0);
        }
        if (T > 1) {
#pragma unroll
            for (int u = 0; u < H; ++u) {
                m[u] = m[u] * (1.0f - s[u]);
                r[u] = s[u];
                const f32x2 t0 = p.theta0 + p.adapt * s[u];
                th[u] = p.theta0 + (t0 - p.theta0) * 0.95f;
            }
            for (int t = 1; t < T - 1; ++t) {
#pragma unroll
                for (int u = 0; u < H; ++u) {
                    const f32x2 mm = (m[u] * p.decay) * (1.0f - r[u]);
                    const f32x2 sp = soft_spike2(mm - th[u]);
                    m[u] = mm * (1.0f - sp);
                    r[u] = r[u] * p.rdecay + sp;
                    const f32x2 t0 = th[u] + p.adapt * sp;
                    th[u] = p.theta0 + (t0 - p.theta0) * 0.95f;
                    s[u] = sp;
                }
            }
#pragma unroll
            for (int u = 0; u < H; ++u) {
                const f32x2 mm = (m[u] * p.decay) * (1.0f - r[u]);
                s[u] = soft_spike2(mm - th[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < H; ++u) {
            v[2 * u] = s[u].x;
            v[2 * u + 1] = s[u].y;
        }
        return;
    }
    float m[W], r[W], th[W];
#pragma unroll
    for (int u = 0; u < W; ++u) {
        m[u] = v[u];
        v[u] = soft_spike(__fsub_rn(m[u], p.theta0));
    }
    if (T <= 1) return;
#pragma unroll
    for (int u = 0; u < W; ++u) {
        m[u] = __fmul_rn(m[u], __fsub_rn(1.0f, v[u]));
        r[u] = v[u];
        const float t0 = __fadd_rn(p.theta0, __fmul_rn(p.adapt, v[u]));
        th[u] = __fadd_rn(p.theta0, __fmul_rn(__fsub_rn(t0, p.theta0), 0.95f));
    }
    for (int t = 1; t < T - 1; ++t) {
#pragma unroll
        for (int u = 0; u < W; ++u) {
            const float mm = __fmul_rn(__fmul_rn(m[u], p.decay), __fsub_rn(1.0f, r[u]));
            const float sp = soft_spike(__fsub_rn(mm, th[u]));
            m[u] = __fmul_rn(mm, __fsub_rn(1.0f, sp));
            r[u] = __fadd_rn(__fmul_rn(r[u], p.rdecay), sp);
            const float t0 = __fadd_rn(th[u], __fmul_rn(p.adapt, sp));
            th[u] = __fadd_rn(p.theta0, __fmul_rn(__fsub_rn(t0, p.theta0), 0.95f));
            v[u] = sp;
        }
    }
#pragma unroll
    for (int u = 0; u < W; ++u) {
        const float mm = __fmul_rn(__fmul_rn(m[u], p.decay), __fsub_rn(1.0f, r[u]));
        v[u] = soft_spike(__fsub_rn(mm, th[u]));
    }
}

// Same loop for chains that sit in DIFFERENT channels (a lane of the ring GEMM's consumer holds 4 consecutive
// columns of one row): the parameters are 2-vectors too.  Element-wise identical to lif_selfloop_n.
struct NeuronP2 {
    f32x2 decay, adapt, rdecay, theta0;
};

template <int H>
__device__ __forceinline__ void lif_selfloop_pairs(f32x2 (&v)[H], const NeuronP2 (&p)[H], int T) {
    f32x2 m[H], r[H], th[H], s[H];
#pragma unroll
    for (int u = 0; u < H; ++u) {
        m[u] = v[u];
        s[u] = soft_spike2(m[u] - p[u].theta0);
    }
    if (T > 1) {
#pragma unroll
        for (int u = 0; u < H; ++u) {
            m[u] = m[u] * (1.0f - s[u]);
            r[u] = s[u];
            const f32x2 t0 = p[u].theta0 + p[u].adapt * s[u];
            th[u] = p[u].theta0 + (t0 - p[u].theta0) * 0.95f;
        }
        for (int t = 1; t < T - 1; ++t) {
#pragma unroll
            for (int u = 0; u < H; ++u) {
                const f32x2 mm = (m[u] * p[u].decay) * (1.0f - r[u]);
                const f32x2 sp = soft_spike2(mm - th[u]);
                m[u] = mm * (1.0f - sp);
                r[u] = r[u] * p[u].rdecay + sp;
                const f32x2 t0 = th[u] + p[u].adapt * sp;
                th[u] = p[u].theta0 + (t0 - p[u].theta0) * 0.95f;
                s[u] = sp;
            }
        }
#pragma unroll
        for (int u = 0; u < H; ++u) {
            const f32x2 mm = (m[u] * p[u].decay) * (1.0f - r[u]);
            s[u] = soft_spike2(mm - th[u]);
        }
    }
#pragma unroll
    for (int u = 0; u < H; ++u) v[u] = s[u];
}

__device__ __forceinline__ float lif_selfloop(float x, const NeuronP& p, int T) {
    float v[1] = {x};
    lif_selfloop_n<1>(v, p, T);
    return v[0];
}

// IEEE-correct f64 square root: the hardware/OCML result refined by one Markstein step
// (s + (d - s*s) / (2 s) with the residual from an FMA), so that distances equal libm's sqrt.
__device__ __forceinline__ double sqrt_cr(double d) {
    if (!(d > 0.0)) return d == 0.0 ? 0.0 : sqrt(d);
    const double s = sqrt(d);
    const double r = __fma_rn(-s, s, d);
    return __fma_rn(r, __ddiv_rn(0.5, s), s);
}

__device__ __forceinline__ float gelu_erf(float x) {   // nn.GELU() default (exact erf form)
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float lrelu02(float x) { return x >= 0.f ? x : 0.2f * x; }

// ---------------------------------------------------------------------------------------------
// GEMM (gemm_f32.hip):  C[r,n] = epi( pro(A)[r,k] * W[n,k]^T + bias[n] )
// ---------------------------------------------------------------------------------------------
enum GemmEpi { EPI_BIAS = 0, EPI_LIF = 1, EPI_GELU = 2, EPI_RESID = 3, EPI_LRELU = 4, EPI_RESID_GELU = 5, EPI_LIF_ATTN = 6 };

struct GemmArgs {
    const float* a;      // [r, lda]
    int64_t r;
    int k, lda;
    const float* w;      // [n, k] row-major (k % 32 == 0)
    int n;
    const float* bias;   // [n] or null
    float* c;            // [r, ldc]
    int ldc;
    int epi;
    // EPI_LIF / EPI_LIF_ATTN: raw neuron params [4][n], T self-loop steps
    const float* lif;
    int lif_T;
    // EPI_RESID / EPI_RESID_GELU: c = f(acc + bias + resid[r, ldr])
    const float* resid;
    int ldr;
    // EPI_LIF_ATTN: additionally c2[row,:] = q[tab[row].x,:] - kf[tab[row].y,:] + c[row,:]
    float* c2;
    const float* q;      // q rows at qkv + 0, kf rows at qkv + d, row stride ldq
    const float* kf;
    int ldq;
    const int2* tab;     // [r] (query-point row, neighbour row) of each edge row (launch_edge_table)
    // split-f16 path (gemm_sf16.hip): W pre-split into hi / lo halves, same [n, k] indexing as w
    const _Float16* w16_hi;
    const _Float16* w16_lo;
    int* ovf;            // device counter raised when an activation tile exceeds the f16 range (may be null)
    // "split rows" (gemm_epi.h): A already split by its producer -> all-DMA ring kernel; outputs to be split
    int a_split, c_split, c2_split;
};
int launch_gemm(const GemmArgs& g, hipStream_t st);        // f32 MFMA (exact f32 products)
int launch_gemm_sf16(const GemmArgs& g, hipStream_t st);   // 3 x f16 MFMA, f32-quality (needs w16_hi/lo); f32 A
int launch_gemm_sf16_ring(const GemmArgs& g, hipStream_t st);   // same arithmetic, A in split rows, 4-slot LDS-DMA ring
int launch_split_weights(const float* w, int64_t count, void* hi, void* lo, int* ovf, hipStream_t st);

// ---------------------------------------------------------------------------------------------
// knn / geometry (knn_outer.hip, geom.hip)
// ---------------------------------------------------------------------------------------------
int launch_knn_outer(const double* cloud, int64_t n, const double* q, int64_t b, int k, int64_t* idx,
                     double* dist, float* patch, hipStream_t st);
int launch_gather_rotate(const double* cloud, int64_t n, const double* q, int64_t b, const int64_t* idx,
                         int k, const float* normals, float* patch, hipStream_t st);
int launch_displace(const double* q, const float* nrm, const float* d, int64_t b, double* out, hipStream_t st);

// farthest-point sampling (fps.hip)
size_t fps_workspace_bytes(int npoint);
int launch_fps(const float* xyz, int64_t n, int npoint, int start, int64_t* out, void* ws, hipStream_t st);
int fps_failed(const void* ws, int npoint, int* flag);

}  // namespace sapcu
