// Is v_mfma_f32_32x32x2_f32 a k-ascending chain of IEEE f32 FMAs?  (the in-patch kNN scores are defined
// as one channel-ascending FMA chain per pair; if the f32 MFMA adds its two products in k order with one rounding each, the score
// phase could run on the matrix pipe bit for bit.)  C[32x32] = A[32xK] . B[Kx32] on one wave, against three host models:
//   chain   acc = fma(a[k], b[k], acc), k ascending (first step a plain product)
//   pair    acc = (a[k] b[k] + a[k+1] b[k+1], exactly, rounded once) + acc style: fma(a[k+1], b[k+1], fma(a[k], b[k], acc)) == chain
//   swapped acc = fma(a[k], b[k], fma(a[k+1], b[k+1], acc))
// Build + run on the GPU box: hipcc -O3 --offload-arch=gfx950 profiles/micro/mfma_f32_exact.hip -o /tmp/mfe && /tmp/mfe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int K = 256;

__global__ void k(const float* A, const float* B, float* C) {       // A [32][K], B [K][32]
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    f32x16 acc;
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * K + k0 + h], B[(k0 + h) * 32 + r], acc, 0, 0, 0);
    for (int e = 0; e < 16; ++e) C[(8 * (e >> 2) + 4 * h + (e & 3)) * 32 + r] = acc[e];
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k16(const float* A, const float* B, float* C) {     // the 16x16x4 shape on the top-left 16 x 16 block
    const int l = threadIdx.x, r = l & 15, g = l >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[r * K + k0 + g], B[(k0 + g) * 32 + r], acc, 0, 0, 0);
    for (int e = 0; e < 4; ++e) C[(4 * g + e) * 32 + r] = acc[e];
}

int main() {
    static float A[32 * K], B[K * 32], C[32 * 32];
    srand(3);
    for (int i = 0; i < 32 * K; ++i) A[i] = (rand() / (float)RAND_MAX - 0.5f) * 2.f;
    for (int i = 0; i < 32 * K; ++i) B[i] = (rand() / (float)RAND_MAX - 0.5f) * 2.f;
    float *dA, *dB, *dC;
    (void)hipMalloc(&dA, sizeof(A));
    (void)hipMalloc(&dB, sizeof(B));
    (void)hipMalloc(&dC, sizeof(C));
    (void)hipMemcpy(dA, A, sizeof(A), hipMemcpyHostToDevice);
    (void)hipMemcpy(dB, B, sizeof(B), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    (void)hipMemcpy(C, dC, sizeof(C), hipMemcpyDeviceToHost);
    int bad_chain = 0, bad_swapped = 0, bad_pairsum = 0;
    double worst = 0;
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            float c1 = 0.f, c2 = 0.f, c3 = 0.f;
            for (int kk = 0; kk < K; kk += 2) {
                const float a0 = A[i * K + kk], a1 = A[i * K + kk + 1], b0 = B[kk * 32 + j], b1 = B[(kk + 1) * 32 + j];
                c1 = fmaf(a1, b1, fmaf(a0, b0, c1));
                c2 = fmaf(a0, b0, fmaf(a1, b1, c2));
                c3 = (float)((double)a0 * b0 + (double)a1 * b1 + (double)c3);      // both products and the sum exact, one rounding
            }
            const float got = C[i * 32 + j];
            bad_chain += memcmp(&got, &c1, 4) != 0;
            bad_swapped += memcmp(&got, &c2, 4) != 0;
            bad_pairsum += memcmp(&got, &c3, 4) != 0;
            worst = fmax(worst, fabs((double)got - c1));
        }
    printf("v_mfma_f32_32x32x2_f32, K = %d, 1024 outputs: differ from the k-ascending FMA chain %d, from the k-descending pair order %d, "
           "from exact-pair-sum-rounded-once %d; max |mfma - chain| = %.3g\n", K, bad_chain, bad_swapped, bad_pairsum, worst);
    hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    (void)hipMemcpy(C, dC, sizeof(C), hipMemcpyDeviceToHost);
    int bad16 = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            float c1 = 0.f;
            for (int kk = 0; kk < K; ++kk) c1 = fmaf(A[i * K + kk], B[kk * 32 + j], c1);
            bad16 += memcmp(&C[i * 32 + j], &c1, 4) != 0;
        }
    printf("v_mfma_f32_16x16x4_f32, K = %d, 256 outputs: differ from the k-ascending FMA chain %d\n", K, bad16);
    return 0;
}
