// Microbenchmark: can ONE wave's MFMAs run under its own neuron arithmetic?  Each iteration issues NM dependent-chain MFMAs
// (v_mfma_f32_32x32x16_f16 on two accumulators, operands in registers) and one epilogue unit (lif_selfloop_n<4>, T = 4 unrolled:
// ~170 VALU instructions), in four arrangements:
//   0  MFMAs only            1  neuron arithmetic only
//   2  both, MFMA block then VALU block (what the chain kernels' phases do at a coarser grain)
//   3  both, interleaved by the scheduler: one MFMA every VPM VALU instructions (sched_group_barrier)
// at 1 / 2 / 4 waves per SIMD, every CU busy.  Build + run on the GPU box:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I<csrc> profiles/micro/mfma_valu_overlap.hip -o /tmp/mvo && /tmp/mvo
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "common.h"
using namespace sapcu;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int NM = 12;      // MFMAs per iteration (two accumulators alternate: dependent every other one)
constexpr int VPM = 9;      // VALU instructions between two MFMAs in the interleaved arrangement (the unit has ~115)

template <int MODE, int THREADS>
__global__ __launch_bounds__(THREADS) void k(const float* in, float* out, int iters) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const NeuronP p{0.9f, 0.01f, 0.5f, 1.0f + 0.001f * (t & 31), 0.f, 0.f};
    float v[4];
    for (int u = 0; u < 4; ++u) v[u] = in[(t * 4 + u) & 4095];
    half8 a, b;
    for (int u = 0; u < 8; ++u) {
        a[u] = (_Float16)in[(t + u) & 4095];
        b[u] = (_Float16)in[(t + 8 + u) & 4095];
    }
    f32x16 c0, c1;
    for (int e = 0; e < 16; ++e) c0[e] = c1[e] = 0.f;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (MODE != 1) {
#pragma unroll
            for (int q = 0; q < NM / 2; ++q) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c1, 0, 0, 0);
            }
        }
        if (MODE == 2) __builtin_amdgcn_sched_barrier(0);
        if (MODE != 0) {
            float w[4] = {v[0] + acc, v[1], v[2], v[3]};
            lif_selfloop_n<4>(w, p, 4);
            acc += w[0] + w[1] + w[2] + w[3];
        }
        if (MODE == 3) {
#pragma unroll
            for (int q = 0; q < NM; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = acc;
    for (int e = 0; e < 16; ++e) s += c0[e] + c1[e];
    out[t] = s;
}

template <int MODE, int THREADS>
static float run(const float* in, float* out, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int grid = 256 * 8;
    hipLaunchKernelGGL((k<MODE, THREADS>), dim3(grid), dim3(THREADS), 0, 0, in, out, 8);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, THREADS>), dim3(grid), dim3(THREADS), 0, 0, in, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float *in, *out;
    (void)hipMalloc(&in, 4096 * 4);
    (void)hipMalloc(&out, 256 * 8 * 1024 * 4);
    float h[4096];
    for (int i = 0; i < 4096; ++i) h[i] = 0.5f + 0.001f * (i % 97);
    (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    const int iters = 2000;
    // grid = 8 workgroups per CU; THREADS 256 / 512 / 1024 with __launch_bounds__ => the resident waves per SIMD differ by LDS-free
    // occupancy only; report ms per launch (same work per thread in every row of one column)
    printf("threads/wg  mfma_only  valu_only  serial  interleaved   (ms per launch, %d iterations, %d MFMAs + 1 unit each)\n", iters, NM);
    printf("%9d %10.3f %10.3f %7.3f %12.3f\n", 256, run<0, 256>(in, out, iters), run<1, 256>(in, out, iters), run<2, 256>(in, out, iters),
           run<3, 256>(in, out, iters));
    printf("%9d %10.3f %10.3f %7.3f %12.3f\n", 1024, run<0, 1024>(in, out, iters), run<1, 1024>(in, out, iters), run<2, 1024>(in, out, iters),
           run<3, 1024>(in, out, iters));
    return 0;
}
