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

// Per-device, thread-safe launch state (include/sapcu.h promises concurrent forwards on different streams and host threads).
//  * DeviceOnce: one bit per HIP device ordinal; `SAPCU_SET_MAX_LDS(once, kernel, bytes)` raises the kernel's dynamic-LDS limit
//    on the CURRENT device the first time the kernel is launched there.  hipFuncSetAttribute is idempotent, so two threads
//    racing through the first launch both set the same value; the bit is published after the call (release / acquire).
//  * device_cu_count(): multiProcessorCount of the current device, cached per ordinal.
struct DeviceOnce {
    unsigned long long done[4] = {0, 0, 0, 0};      // 256 device ordinals
    bool test(int dev) const { return (__atomic_load_n(&done[(dev >> 6) & 3], __ATOMIC_ACQUIRE) >> (dev & 63)) & 1ull; }
    void set(int dev) { __atomic_fetch_or(&done[(dev >> 6) & 3], 1ull << (dev & 63), __ATOMIC_RELEASE); }
};

#define SAPCU_SET_MAX_LDS(once, kernel, bytes)                                                                          \
    do {                                                                                                                \
        int _dev = 0;                                                                                                   \
        SAPCU_CHECK_HIP(hipGetDevice(&_dev));                                                                           \
        if (!(once).test(_dev)) {                                                                                       \
            SAPCU_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),                                  \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (bytes)));                  \
            (once).set(_dev);                                                                                           \
        }                                                                                                               \
    } while (0)

inline int device_cu_count() {
    static int cus[256];                              // 0 = not read yet; written once per ordinal with the same value by any thread
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    int v = __atomic_load_n(&cus[dev & 255], __ATOMIC_ACQUIRE);
    if (v == 0) {
        hipDeviceProp_t prop;
        v = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
        __atomic_store_n(&cus[dev & 255], v, __ATOMIC_RELEASE);
    }
    return v;
}

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

// ---------------------------------------------------------------------------------------------
// `for t in range(T): x, *st = snn(x, *st)` — spikes fed back as the next input (fn:319-320), LIF: THE production
// neuron loop (GEMM epilogues, pos-enc kernel, stem).  State lives in registers for all T steps, and the loop is
// peeled around two exact facts about neuron_step<false>:
//   * step 0 starts from m = 0, r = 0:  m = x,  r = s0;
//   * for t >= 1 the gate `x * (r <= 0)` is closed — the reference's clamped spike surrogate is >= 3.85e-23 > 0, so its r > 0
//     for every finite input — and the fed-back input contributes exactly +0 (the peel follows the reference's gate, not the
//     sign of this build's own r, which may underflow to 0 without the clamp: see soft_spike2);  the state updates after the
//     last spike are dead.
// Chains are processed as 2-vectors (v_pk_mul/add/fma_f32: one instruction, two chains); H pairs are advanced together
// (independent chains = VALU ILP).
//
// Arithmetic.  The kernels that run this loop are bound by VALU issue, so by default every update is
// written with fused multiply-adds — 12 packed operations per mid step instead of 19:
//     mm  = (m*decay)*(1-r)                         = fma(-(m*decay), r, m*decay)
//     g   = exp(-x^2/2) * 0.5/sqrt(2 pi)            = exp2(fma(x*x, -log2(e)/2, log2(0.5/sqrt(2 pi))))
//     m'  = mm*(1-s)                                = fma(-mm, s, mm)
//     r'  = r*rdecay + s                            = fma(r, rdecay, s)
//     th' = th0 + ((th + adapt*s) - th0)*0.95       = fma(th, 0.95, fma(s, 0.95*adapt, 0.05*th0))
// Each right-hand side is the left-hand side's value with fewer roundings; against the reference's separately rounded
// sequence the spikes move by <= 3e-7 (neuron-unit bar 1e-6: tests/test_gpu_parity.py::test_neuron_unit_*).
// -DSAPCU_LIF_EXACT_ORDER restores the reference's operation order op for op.
// ---------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct NeuronP2 {
    f32x2 decay, adapt, rdecay, theta0;
};

__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 pk_clamp10(f32x2 d) { return f32x2{clampf(d.x, -10.0f, 10.0f), clampf(d.y, -10.0f, 10.0f)}; }

// two spikes at once; clamp, exp2 and rcp are per lane.  Element-wise the arithmetic is exactly soft_spike()'s
// (exact-order build) or its fused form (default).
//
// The reference clamps x to +-10 before the two exponentials.  The default (fused) build leaves the clamp out — two of a
// pair-step's 20 instructions in kernels that are bound by VALU issue: beyond |x| = 10 the sigmoid term is 0 or 1 in f32
// either way (exp2(+-144) is inf / below 2^-126 next to 1), and the Gaussian term is 0.1995 exp(-50) = 3.9e-23 clamped
// against something smaller un-clamped: the spike moves by < 4e-23 absolute, the split-f16 operand made from it not at all
// (below half the smallest f16 subnormal).  x = +-inf gives 0 / 1 + 0 without the clamp as well (x*x = inf -> exp2(-inf) = 0).
//
// What the clamp also guaranteed: spike >= 0.1995 exp(-50) = 3.85e-23 > 0, hence refractory r > 0 from step 0 on — the closed gate
// that the peeled loops and fd's dead-stage elimination rest on.  Un-clamped, both exponentials underflow for x below about -13.2
// and the spike is exactly 0.  The VALUE path does not notice (1 - r and m * (1 - r) are the same f32 for r = 0 and r = 3.85e-23),
// but a refractory state built from it would read "gate open" where the reference's gate is closed.  The stepping forms that
// carry r and count gate violations (NeuronStep2 / NeuronStep2V) therefore floor the refractory at SPIKE_FLOOR, the reference's
// own minimum, after every step: r is then >= the reference's r up to rounding, and gate_open() tests what the reference's gate tests.
constexpr float SPIKE_FLOOR = 3.8e-23f;      // just below 0.5 exp(-50) / sqrt(2 pi) + 0.5 sigmoid(-100) = 3.847e-23 (fn:135-146 at x = -10)
__device__ __forceinline__ f32x2 soft_spike2(f32x2 d) {
#if defined(SAPCU_LIF_EXACT_ORDER) || defined(SAPCU_SPIKE_CLAMP)     // (the second macro: same-box A/B builds, profiles/step_ab.py)
    const f32x2 x = pk_clamp10(d);
#else
    const f32x2 x = d;
#endif
    const f32x2 b = x * -14.426950408889634074f;
    f32x2 g, e, s;
#ifdef SAPCU_LIF_EXACT_ORDER
    const f32x2 a = (x * x) * -0.72134752044448170368f;
    g.x = __builtin_amdgcn_exp2f(a.x);
    g.y = __builtin_amdgcn_exp2f(a.y);
    g = g * 0.19947114020071633897f;
#else
    // log2(0.5/sqrt(2 pi)) = -2.3257...: the scale rides in the exponent
    const f32x2 a = pk_fma(x * x, f32x2{-0.72134752044448170368f, -0.72134752044448170368f},
                           f32x2{-2.3257480647361593f, -2.3257480647361593f});
    g.x = __builtin_amdgcn_exp2f(a.x);
    g.y = __builtin_amdgcn_exp2f(a.y);
#endif
    e.x = __builtin_amdgcn_exp2f(b.x);
    e.y = __builtin_amdgcn_exp2f(b.y);
    const f32x2 den = e + 1.0f;
    s.x = __builtin_amdgcn_rcpf(den.x);
    s.y = __builtin_amdgcn_rcpf(den.y);
    return pk_fma(f32x2{0.5f, 0.5f}, s, g);
}

template <int H>
__device__ __forceinline__ void lif_selfloop_pairs(f32x2 (&v)[H], const NeuronP2 (&p)[H], int T) {
    f32x2 m[H], r[H], th[H], s[H];
#pragma unroll
    for (int u = 0; u < H; ++u) {
        m[u] = v[u];
        s[u] = soft_spike2(m[u] - p[u].theta0);
    }
    if (T > 1) {
#ifdef SAPCU_LIF_EXACT_ORDER
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
#else
        f32x2 a95[H], thc[H];
        const f32x2 k95 = f32x2{0.95f, 0.95f};
#pragma unroll
        for (int u = 0; u < H; ++u) {
            a95[u] = p[u].adapt * 0.95f;
            thc[u] = p[u].theta0 * 0.05f;
            m[u] = pk_fma(-m[u], s[u], m[u]);
            r[u] = s[u];
            th[u] = pk_fma(p[u].theta0, k95, pk_fma(s[u], a95[u], thc[u]));
        }
        for (int t = 1; t < T - 1; ++t) {
#pragma unroll
            for (int u = 0; u < H; ++u) {
                const f32x2 md = m[u] * p[u].decay;
                const f32x2 mm = pk_fma(-md, r[u], md);
                const f32x2 sp = soft_spike2(mm - th[u]);
                m[u] = pk_fma(-mm, sp, mm);
                r[u] = pk_fma(r[u], p[u].rdecay, sp);
                th[u] = pk_fma(th[u], k95, pk_fma(sp, a95[u], thc[u]));
                s[u] = sp;
            }
        }
#pragma unroll
        for (int u = 0; u < H; ++u) {
            const f32x2 md = m[u] * p[u].decay;
            s[u] = soft_spike2(pk_fma(-md, r[u], md) - th[u]);
        }
#endif
    }
#pragma unroll
    for (int u = 0; u < H; ++u) v[u] = s[u];
}

// W chains of ONE channel (an odd W repeats its last chain in the spare half of a pair)
template <int W>
__device__ __forceinline__ void lif_selfloop_n(float (&v)[W], const NeuronP& p, int T) {
    constexpr int H = (W + 1) / 2;
    f32x2 pv[H];
    NeuronP2 pp[H];
#pragma unroll
    for (int u = 0; u < H; ++u) {
        pv[u] = f32x2{v[2 * u], v[(2 * u + 1 < W) ? 2 * u + 1 : 2 * u]};
        pp[u] = NeuronP2{{p.decay, p.decay}, {p.adapt, p.adapt}, {p.rdecay, p.rdecay}, {p.theta0, p.theta0}};
    }
    lif_selfloop_pairs<H>(pv, pp, T);
#pragma unroll
    for (int u = 0; u < H; ++u) {
        v[2 * u] = pv[u].x;
        if (2 * u + 1 < W) v[2 * u + 1] = pv[u].y;
    }
}

__device__ __forceinline__ float lif_selfloop(float x, const NeuronP& p, int T) {
    float v[1] = {x};
    lif_selfloop_n<1>(v, p, T);
    return v[0];
}

// Two chains of ONE channel stepped together with per-step outputs (fd's encoder keeps the spikes of every step,
// fd:432-474): input only at step 0, zero afterwards (closed gate, counted by the caller).  Same fused arithmetic as
// lif_selfloop_pairs; EIF adds dT*exp(clamp((m_prev - rh)/(dT + 1e-6), +-5)) un-gated (fd:245-252), the division as a
// multiplication by the once-computed reciprocal.  -DSAPCU_LIF_EXACT_ORDER: neuron_step<EIF> per lane.
struct NeuronS2 {
    f32x2 m, th, r;
};

template <bool EIF>
struct NeuronStep2 {
    // (the scalar parameters the packed form needs are plain members: with the whole NeuronP embedded, the compiler's scalar
    //  replacement of the object sliced dT | rh as one 8-byte piece overlapping other pieces and rebuilt it through a stack slot)
#ifdef SAPCU_LIF_EXACT_ORDER
    NeuronP p;
#endif
    f32x2 decay, rdecay, a95, thc, theta0, dT2, rh2;
    float inv_dT;
    NeuronS2 s;
#ifdef SAPCU_LIF_EXACT_ORDER
    NeuronS sx, sy;
#endif
    __device__ __forceinline__ explicit NeuronStep2(const NeuronP& pp) {
#ifdef SAPCU_LIF_EXACT_ORDER
        p = pp;
#endif
        const NeuronP& p = pp;
        dT2 = f32x2{p.dT, p.dT};
        rh2 = f32x2{p.rh, p.rh};
        decay = f32x2{p.decay, p.decay};
        rdecay = f32x2{p.rdecay, p.rdecay};
        a95 = f32x2{p.adapt * 0.95f, p.adapt * 0.95f};
        thc = f32x2{p.theta0 * 0.05f, p.theta0 * 0.05f};
        theta0 = f32x2{p.theta0, p.theta0};
        inv_dT = EIF ? __fdiv_rn(1.0f, __fadd_rn(p.dT, 1e-6f)) : 0.f;
        s.m = f32x2{0.f, 0.f};
        s.r = f32x2{0.f, 0.f};
        s.th = theta0;
#ifdef SAPCU_LIF_EXACT_ORDER
        sx = neuron_init(p);
        sy = neuron_init(p);
#endif
    }
    __device__ __forceinline__ bool gate_open() const {
#ifdef SAPCU_LIF_EXACT_ORDER
        return sx.r <= 0.f || sy.r <= 0.f;
#else
        return s.r.x <= 0.f || s.r.y <= 0.f;
#endif
    }
    // one step; x is the input (added only while the gate is open, i.e. at step 0 in eval mode)
    __device__ __forceinline__ f32x2 step(f32x2 x, bool first) {
#ifdef SAPCU_LIF_EXACT_ORDER
        return f32x2{neuron_step<EIF>(x.x, sx, p), neuron_step<EIF>(x.y, sy, p)};
#else
        f32x2 mm;
        if (first) {
            mm = x;                                             // 0*decay*(1-0) + x
        } else {
            const f32x2 md = s.m * decay;
            mm = pk_fma(-md, s.r, md);
        }
        if (EIF) {
            f32x2 a = (s.m - rh2) * inv_dT;
            a = f32x2{clampf(a.x, -5.0f, 5.0f), clampf(a.y, -5.0f, 5.0f)} * 1.4426950408889634074f;
            const f32x2 e = f32x2{__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
            mm = pk_fma(e, dT2, mm);
        }
        const f32x2 sp = soft_spike2(mm - s.th);
        s.m = pk_fma(-mm, sp, mm);
        {   // (floor: see soft_spike2.  At EVERY step, as the reference's r >= its spike >= 3.85e-23 at every step: floored at step 0
            //  only, r * rdecay^t would underflow to 0 after ~15 silent steps at rdecay = 0.1 and read "gate open" at large T)
            const f32x2 rn = first ? sp : pk_fma(s.r, rdecay, sp);
            s.r = f32x2{fmaxf(rn.x, SPIKE_FLOOR), fmaxf(rn.y, SPIKE_FLOOR)};
        }
        s.th = pk_fma(s.th, f32x2{0.95f, 0.95f}, pk_fma(sp, a95, thc));
        return sp;
#endif
    }
};

// NeuronStep2 for a pair of DIFFERENT channels (lane .x = channel a, .y = channel b of one point): the same arithmetic per lane, the
// parameters as 2-vectors.  (fd's per-stage EdgeConv + neuron kernel walks four consecutive channels per thread.)
template <bool EIF>
struct NeuronStep2V {
#ifdef SAPCU_LIF_EXACT_ORDER
    NeuronP pa, pb;
#endif
    f32x2 decay, rdecay, a95, thc, theta0, dT, rh, inv_dT;
    NeuronS2 s;
#ifdef SAPCU_LIF_EXACT_ORDER
    NeuronS sx, sy;
#endif
    __device__ __forceinline__ NeuronStep2V(const NeuronP& a, const NeuronP& b) {
#ifdef SAPCU_LIF_EXACT_ORDER
        pa = a;
        pb = b;
#endif
        decay = f32x2{a.decay, b.decay};
        rdecay = f32x2{a.rdecay, b.rdecay};
        a95 = f32x2{a.adapt * 0.95f, b.adapt * 0.95f};
        thc = f32x2{a.theta0 * 0.05f, b.theta0 * 0.05f};
        theta0 = f32x2{a.theta0, b.theta0};
        dT = f32x2{a.dT, b.dT};
        rh = f32x2{a.rh, b.rh};
        inv_dT = EIF ? f32x2{__fdiv_rn(1.0f, __fadd_rn(a.dT, 1e-6f)), __fdiv_rn(1.0f, __fadd_rn(b.dT, 1e-6f))} : f32x2{0.f, 0.f};
        s.m = f32x2{0.f, 0.f};
        s.r = f32x2{0.f, 0.f};
        s.th = theta0;
#ifdef SAPCU_LIF_EXACT_ORDER
        sx = neuron_init(a);
        sy = neuron_init(b);
#endif
    }
    __device__ __forceinline__ bool gate_open() const {
#ifdef SAPCU_LIF_EXACT_ORDER
        return sx.r <= 0.f || sy.r <= 0.f;
#else
        return s.r.x <= 0.f || s.r.y <= 0.f;
#endif
    }
    __device__ __forceinline__ f32x2 step(f32x2 x, bool first) {
#ifdef SAPCU_LIF_EXACT_ORDER
        return f32x2{neuron_step<EIF>(x.x, sx, pa), neuron_step<EIF>(x.y, sy, pb)};
#else
        f32x2 mm;
        if (first) {
            mm = x;
        } else {
            const f32x2 md = s.m * decay;
            mm = pk_fma(-md, s.r, md);
        }
        if (EIF) {
            f32x2 a = (s.m - rh) * inv_dT;
            a = f32x2{clampf(a.x, -5.0f, 5.0f), clampf(a.y, -5.0f, 5.0f)} * 1.4426950408889634074f;
            const f32x2 e = f32x2{__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
            mm = pk_fma(e, dT, mm);
        }
        const f32x2 sp = soft_spike2(mm - s.th);
        s.m = pk_fma(-mm, sp, mm);
        {   // (floor: see soft_spike2.  At EVERY step, as the reference's r >= its spike >= 3.85e-23 at every step: floored at step 0
            //  only, r * rdecay^t would underflow to 0 after ~15 silent steps at rdecay = 0.1 and read "gate open" at large T)
            const f32x2 rn = first ? sp : pk_fma(s.r, rdecay, sp);
            s.r = f32x2{fmaxf(rn.x, SPIKE_FLOOR), fmaxf(rn.y, SPIKE_FLOOR)};
        }
        s.th = pk_fma(s.th, f32x2{0.95f, 0.95f}, pk_fma(sp, a95, thc));
        return sp;
#endif
    }
};

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
// order-preserving map float -> unsigned (a < b  <=>  key(a) < key(b) for non-NaN values); 0 is below every key
__device__ __forceinline__ unsigned float_max_key(float x) {
    const unsigned b = __float_as_uint(x);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float float_from_max_key(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

// ---------------------------------------------------------------------------------------------
// GEMM (gemm_f32.hip):  C[r,n] = epi( pro(A)[r,k] * W[n,k]^T + bias[n] )
// ---------------------------------------------------------------------------------------------
enum GemmEpi { EPI_BIAS = 0, EPI_LIF = 1, EPI_GELU = 2, EPI_RESID = 3, EPI_LRELU = 4, EPI_RESID_GELU = 5, EPI_LIF_ATTN = 6,
               EPI_LRELU_MAX = 7,     // LeakyReLU, then max over groups of max_m rows instead of storing C (gemm_sf16 / gemm_sf16_bt)
               EPI_LIF_MAX = 8 };     // gemm_sf16.hip only: T-step neuron, then the same max (fn conv_final -> max over points)

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
    // EPI_LRELU_MAX (fd/snn_coder.py:476-480: multi_scale_conv -> max over the patch's points) and EPI_LIF_MAX
    // (fn/snn_coder.py:465-472: conv_final + LIF -> max over the patch's points): C is never stored;
    // max_keys[(row / max_m) * n + col] = max over the group's rows of the order-preserving integer key of the value
    // (float_max_key; the buffer starts at 0 = below every key; launch_decode_max_keys turns it into floats)
    unsigned* max_keys;
    int max_m;
};
int launch_gemm(const GemmArgs& g, hipStream_t st);        // f32 MFMA (exact f32 products)
int launch_gemm_sf16(const GemmArgs& g, hipStream_t st);   // 3 x f16 MFMA, f32-quality (needs w16_hi/lo); f32 A
int launch_gemm_sf16_ring(const GemmArgs& g, hipStream_t st);   // same arithmetic, A in split rows, 4-slot LDS-DMA ring
int launch_gemm_sf16_bt(const GemmArgs& g, hipStream_t st);     // same arithmetic and results, 256 x 256/128 tiles (gemm_sf16_bt.hip)
bool gemm_sf16_bt_ok(const GemmArgs& g);                        // ... for the shapes / epilogues it takes
int launch_gemm_split_rows(const GemmArgs& g, hipStream_t st, bool allow_bt = true);  // picks between the two (model.hip; allow_bt = false: ring only)
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
