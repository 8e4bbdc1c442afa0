// Microbenchmark: VALU cost of the T-step LIF self-loop (common.h lif_selfloop_pairs, packed v_pk_*_f32 arithmetic) against
// the same arithmetic written with scalar f32 instructions, at 1 / 2 / 4 waves per SIMD.  Build + run on the GPU box:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I<csrc> profiles/micro/lif_rate.hip -o /tmp/lif_rate && /tmp/lif_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "common.h"
using namespace sapcu;

__device__ __forceinline__ float spike1(float d) {
#ifdef SAPCU_LIF_EXACT_ORDER
    const float x = clampf(d, -10.0f, 10.0f);
#else
    const float x = d;                                  // (as common.h soft_spike2 in the default build)
#endif
    const float b = x * -14.426950408889634074f;
    const float a = __fmaf_rn(x * x, -0.72134752044448170368f, -2.3257480647361593f);
    const float g = __builtin_amdgcn_exp2f(a);
    const float e = __builtin_amdgcn_exp2f(b);
    const float s = __builtin_amdgcn_rcpf(e + 1.0f);
    return __fmaf_rn(0.5f, s, g);
}

template <int W>
__device__ __forceinline__ void lif_scalar(float (&v)[W], const NeuronP& p, int T) {
    float m[W], r[W], th[W], s[W];
#pragma unroll
    for (int u = 0; u < W; ++u) { m[u] = v[u]; s[u] = spike1(m[u] - p.theta0); }
    if (T > 1) {
        const float a95 = p.adapt * 0.95f, thc = p.theta0 * 0.05f;
#pragma unroll
        for (int u = 0; u < W; ++u) {
            m[u] = __fmaf_rn(-m[u], s[u], m[u]);
            r[u] = s[u];
            th[u] = __fmaf_rn(p.theta0, 0.95f, __fmaf_rn(s[u], a95, thc));
        }
        for (int t = 1; t < T - 1; ++t) {
#pragma unroll
            for (int u = 0; u < W; ++u) {
                const float md = m[u] * p.decay;
                const float mm = __fmaf_rn(-md, r[u], md);
                const float sp = spike1(mm - th[u]);
                m[u] = __fmaf_rn(-mm, sp, mm);
                r[u] = __fmaf_rn(r[u], p.rdecay, sp);
                th[u] = __fmaf_rn(th[u], 0.95f, __fmaf_rn(sp, a95, thc));
                s[u] = sp;
            }
        }
#pragma unroll
        for (int u = 0; u < W; ++u) {
            const float md = m[u] * p.decay;
            s[u] = spike1(__fmaf_rn(-md, r[u], md) - th[u]);
        }
    }
#pragma unroll
    for (int u = 0; u < W; ++u) v[u] = s[u];
}

template <int MODE, int W>
__global__ __launch_bounds__(256) void k(const float* in, float* out, int iters, int T) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    NeuronP p{0.9f, 0.01f, 0.5f, 1.0f + 0.001f * (t & 31), 0.f, 0.f};
    float v[W];
    for (int u = 0; u < W; ++u) v[u] = in[(t * W + u) & 4095];
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        float x[W];
#pragma unroll
        for (int u = 0; u < W; ++u) x[u] = v[u] + acc * 1e-3f;
        if (MODE == 0) lif_selfloop_n<W>(x, p, T);
        else lif_scalar<W>(x, p, T);
#pragma unroll
        for (int u = 0; u < W; ++u) acc += x[u];
    }
    out[t] = acc;
}

template <int MODE, int W>
double run(int waves_per_simd, int iters, float* in, float* out) {
    const int blocks = 256 * waves_per_simd;      // 256 threads = 4 waves = one per SIMD of a CU
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, W>), dim3(blocks), dim3(256), 0, 0, in, out, iters, 4);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, W>), dim3(blocks), dim3(256), 0, 0, in, out, iters, 4);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double elems = (double)blocks * 256 * W * iters;
    return ms * 1e6 / (elems / 1000.0);           // ns per 1000 elements (chip-wide)
}

int main() {
    float *in, *out;
    hipMalloc(&in, 4096 * 4); hipMalloc(&out, 256 * 8 * 256 * 4);
    float h[4096]; for (int i = 0; i < 4096; ++i) h[i] = -2.0f + 4.0f * i / 4096;
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    // bit-identity of the two forms
    hipLaunchKernelGGL((k<0, 8>), dim3(4), dim3(256), 0, 0, in, out, 3, 4);
    hipLaunchKernelGGL((k<1, 8>), dim3(4), dim3(256), 0, 0, in, out + 1024, 3, 4);
    float a[2048]; hipMemcpy(a, out, sizeof(a), hipMemcpyDeviceToHost);
    int diff = 0; for (int i = 0; i < 1024; ++i) diff += a[i] != a[1024 + i];
    printf("packed vs scalar: %d of 1024 results differ\n", diff);
    for (int wps : {1, 2, 4}) {
        printf("waves/SIMD %d  packed W=8: %.3f  scalar W=8: %.3f  packed W=4: %.3f  scalar W=4: %.3f  scalar W=2: %.3f   ns per 1000 elements (4 steps)\n", wps,
               run<0, 8>(wps, 400, in, out), run<1, 8>(wps, 400, in, out), run<0, 4>(wps, 400, in, out), run<1, 4>(wps, 400, in, out), run<1, 2>(wps, 400, in, out));
    }
    return 0;
}
