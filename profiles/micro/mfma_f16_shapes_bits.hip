// Do the two f16 MFMA shapes of gfx950 round alike?  D = A[16 x 32] . B[32 x 16] + C computed as
//   (a) two chained v_mfma_f32_32x32x16_f16 (k 0..15, then k 16..31) — what every split-f16 GEMM of this library issues —
//   (b) one v_mfma_f32_16x16x32_f16,
// on random f16 operands with a wide exponent spread and a random f32 C, compared bit for bit on the 16 x 16 block both produce.
// If (b) == (a) the library could move to the faster-clocking 16x16x32 shape (profiles/micro/mfma_shape.hip) kernel by kernel
// without giving up bit-identity between its kernels.  Build + run: hipcc -O3 --offload-arch=gfx950 <this> -o /tmp/mfs && /tmp/mfs
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// A [32][32] (row, k), B [32][32] (col, k) f16; C [32][32] f32.  out32 / out16: [16][16]
__global__ void k(const _Float16* A, const _Float16* B, const float* C, float* out32, float* out16) {
    const int l = threadIdx.x;
    {   // (a)
        const int r = l & 31, h = l >> 5;
        f32x16 acc;
        for (int e = 0; e < 16; ++e) acc[e] = C[(8 * (e >> 2) + 4 * h + (e & 3)) * 32 + r];
        for (int s = 0; s < 2; ++s) {
            half8 a, b;
            for (int j = 0; j < 8; ++j) {
                a[j] = A[r * 32 + 16 * s + 8 * h + j];
                b[j] = B[r * 32 + 16 * s + 8 * h + j];
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        }
        for (int e = 0; e < 16; ++e) {
            const int row = 8 * (e >> 2) + 4 * h + (e & 3);
            if (row < 16 && r < 16) out32[row * 16 + r] = acc[e];
        }
    }
    {   // (b)
        const int r = l & 15, g = l >> 4;
        f32x4 acc;
        for (int e = 0; e < 4; ++e) acc[e] = C[(4 * g + e) * 32 + r];
        half8 a, b;
        for (int j = 0; j < 8; ++j) {
            a[j] = A[r * 32 + 8 * g + j];
            b[j] = B[r * 32 + 8 * g + j];
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
        for (int e = 0; e < 4; ++e) out16[(4 * g + e) * 16 + r] = acc[e];
    }
}

int main() {
    static _Float16 A[1024], B[1024];
    static float C[1024], o32[256], o16[256];
    _Float16 *dA, *dB;
    float *dC, *d32, *d16;
    (void)hipMalloc(&dA, sizeof(A));
    (void)hipMalloc(&dB, sizeof(B));
    (void)hipMalloc(&dC, sizeof(C));
    (void)hipMalloc(&d32, sizeof(o32));
    (void)hipMalloc(&d16, sizeof(o16));
    srand(7);
    int bad = 0, total = 0, bad_exact = 0;
    double worst = 0;
    for (int trial = 0; trial < 200; ++trial) {
        const int spread = trial % 4;                // 0: [-1,1), 1..3: exponents spread over 2^(+-4 spread)
        for (int i = 0; i < 1024; ++i) {
            const float ea = spread ? ldexpf(1.f, (rand() % (8 * spread + 1)) - 4 * spread) : 1.f;
            const float eb = spread ? ldexpf(1.f, (rand() % (8 * spread + 1)) - 4 * spread) : 1.f;
            A[i] = (_Float16)((rand() / (float)RAND_MAX * 2.f - 1.f) * ea);
            B[i] = (_Float16)((rand() / (float)RAND_MAX * 2.f - 1.f) * eb);
            C[i] = (trial & 1) ? (rand() / (float)RAND_MAX * 2.f - 1.f) * 4.f : 0.f;
        }
        (void)hipMemcpy(dA, A, sizeof(A), hipMemcpyHostToDevice);
        (void)hipMemcpy(dB, B, sizeof(B), hipMemcpyHostToDevice);
        (void)hipMemcpy(dC, C, sizeof(C), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, d32, d16);
        (void)hipMemcpy(o32, d32, sizeof(o32), hipMemcpyDeviceToHost);
        (void)hipMemcpy(o16, d16, sizeof(o16), hipMemcpyDeviceToHost);
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                double ex = C[i * 32 + j];
                for (int kk = 0; kk < 32; ++kk) ex += (double)(float)A[i * 32 + kk] * (double)(float)B[j * 32 + kk];
                const float exf = (float)ex;
                ++total;
                bad += memcmp(&o32[i * 16 + j], &o16[i * 16 + j], 4) != 0;
                bad_exact += memcmp(&o16[i * 16 + j], &exf, 4) != 0;
                worst = fmax(worst, fabs((double)o32[i * 16 + j] - o16[i * 16 + j]) / fmax(1e-30, fabs(ex)));
            }
    }
    printf("%d outputs: two chained 32x32x16 vs one 16x16x32 differ in %d (max relative difference %.3g); 16x16x32 differs from the "
           "exactly-summed-rounded-once value in %d\n", total, bad, worst, bad_exact);
    return 0;
}
