// Microbenchmark: the two f16 MFMA shapes of gfx950 on RANDOM operands, same flops per wave — does the chip hold a higher clock on
// v_mfma_f32_16x16x32_f16 than on v_mfma_f32_32x32x16_f16 (the shape every GEMM of this library uses)?  Operands in registers,
// 4 independent accumulator chains per wave, 1 / 2 waves per SIMD, every CU busy.  Build + run on the GPU box:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 profiles/micro/mfma_shape.hip -o /tmp/mfma_shape && /tmp/mfma_shape
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, int THREADS>
__global__ __launch_bounds__(THREADS) void k(const _Float16* in, float* out, int iters, unsigned long long* clk) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    half8 a[4], b[4];
    for (int q = 0; q < 4; ++q)
        for (int u = 0; u < 8; ++u) {
            a[q][u] = in[(t * 8 + u + 64 * q) & 65535];
            b[q][u] = in[(t * 8 + u + 64 * q + 32768) & 65535];
        }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    if (SHAPE == 32) {
        f32x16 c[4];
        for (int q = 0; q < 4; ++q)
            for (int e = 0; e < 16; ++e) c[q][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int q = 0; q < 4; ++q) c[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(q + r) & 3], b[q], c[q], 0, 0, 0);
        }
        for (int q = 0; q < 4; ++q)
            for (int e = 0; e < 16; ++e) s += c[q][e];
    } else {
        // the same flops: one 32x32x16 (32 K MAC... 16384 MACs) = two 16x16x32 (8192 MACs each)
        f32x4 c[8];
        for (int q = 0; q < 8; ++q)
            for (int e = 0; e < 4; ++e) c[q][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int q = 0; q < 8; ++q) c[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(q + r) & 3], b[q & 3], c[q], 0, 0, 0);
        }
        for (int q = 0; q < 8; ++q)
            for (int e = 0; e < 4; ++e) s += c[q][e];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[t] = s;
    if (threadIdx.x == 0) {
        clk[2 * blockIdx.x] = t1 - t0;
        clk[2 * blockIdx.x + 1] = r1 - r0;
    }
}

template <int SHAPE, int THREADS>
static void run(const _Float16* in, float* out, unsigned long long* clk, int iters, int wgs_per_cu) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int grid = 256 * wgs_per_cu;
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<SHAPE, THREADS>), dim3(grid), dim3(THREADS), 0, 0, in, out, iters, clk);
    (void)hipEventRecord(e0);
    const int reps = 5;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k<SHAPE, THREADS>), dim3(grid), dim3(THREADS), 0, 0, in, out, iters, clk);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    unsigned long long h[2 * 256 * 8];
    (void)hipMemcpy(h, clk, sizeof(unsigned long long) * 2 * grid, hipMemcpyDeviceToHost);
    double ghz = 0;
    for (int i = 0; i < grid; ++i) ghz += (double)h[2 * i] / (double)h[2 * i + 1] * 0.1;
    ghz /= grid;
    const double flop = 2.0 * 16384 * 16 * (double)iters * (THREADS / 64) * grid;
    printf("shape %2d  %4d threads x %d wg/CU: %8.3f ms  %7.1f TFLOP/s  in-kernel clock %.2f GHz\n", SHAPE, THREADS, wgs_per_cu, ms, flop / ms / 1e9, ghz);
}

int main() {
    _Float16* in;
    float* out;
    unsigned long long* clk;
    (void)hipMalloc(&in, 65536 * 2);
    (void)hipMalloc(&out, 256 * 8 * 1024 * 4);
    (void)hipMalloc(&clk, 256 * 8 * 2 * 8);
    static _Float16 h[65536];
    srand(1);
    for (int i = 0; i < 65536; ++i) h[i] = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 2.0f);
    (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    const int iters = 20000;
    for (int pass = 0; pass < 2; ++pass) {
        run<32, 256>(in, out, clk, iters, 1);      // one wave per SIMD
        run<16, 256>(in, out, clk, iters, 1);
        run<32, 512>(in, out, clk, iters, 1);      // two
        run<16, 512>(in, out, clk, iters, 1);
    }
    return 0;
}
