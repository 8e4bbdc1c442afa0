// Float64 geometry around the networks: patch gather + centre (+ Rodrigues rotation), displacement.
// Replaces /root/reference/generation.py:128-129,154-160 (+ rotation_matrix_from_vectors :30-47)
// and :171-172.  All arithmetic is explicitly rounded f64/f32 in numpy's operation order.
#include "common.h"

namespace sapcu {

// R = I + K + K.K * ((1-c)/s^2) for a = unit(n) (normalised in f32 like numpy does for an f32
// vector), b = +x.  With b = e_x: v = a x b = (0, a2, -a1), c = a0.  Identity when v == 0.
__device__ __forceinline__ void rotation_to_x(const float* __restrict__ nrm, double R[3][3]) {
    const float n0 = nrm[0], n1 = nrm[1], n2 = nrm[2];
    // np.linalg.norm(f32[3]) = sqrt(x.dot(x)); OpenBLAS' x86-64 sdot rounds each product to f32 and sums
    // them in a double (kernel/x86_64/sdot.c tail loop) — measured: 0 mismatches in 50k vectors.
    const double sq = ((double)__fmul_rn(n0, n0) + (double)__fmul_rn(n1, n1)) + (double)__fmul_rn(n2, n2);
    const float nn = __fsqrt_rn((float)sq);
    const double a0 = (double)__fdiv_rn(n0, nn);
    const double a1 = (double)__fdiv_rn(n1, nn);
    const double a2 = (double)__fdiv_rn(n2, nn);
    R[0][0] = 1.0; R[0][1] = 0.0; R[0][2] = 0.0;
    R[1][0] = 0.0; R[1][1] = 1.0; R[1][2] = 0.0;
    R[2][0] = 0.0; R[2][1] = 0.0; R[2][2] = 1.0;
    if (a1 == 0.0 && a2 == 0.0) return;   // `if any(v)` false: also for n = -x (reference quirk)
    const double v1 = a2, v2 = -a1;
    const double ss = __dadd_rn(__dmul_rn(v1, v1), __dmul_rn(v2, v2));
    const double s = sqrt_cr(ss);
    const double f = __ddiv_rn(__dsub_rn(1.0, a0), __dmul_rn(s, s));
    // K = [[0,-v2,v1],[v2,0,0],[-v1,0,0]];  K.K = [[-(v1^2+v2^2),0,0],[0,-v2^2,v1 v2],[0,v1 v2,-v1^2]]
    const double k00 = -__dadd_rn(__dmul_rn(v2, v2), __dmul_rn(v1, v1));
    const double k11 = -__dmul_rn(v2, v2);
    const double k22 = -__dmul_rn(v1, v1);
    const double k12 = __dmul_rn(v2, v1);
    R[0][0] = __dadd_rn(1.0, __dmul_rn(k00, f));
    R[0][1] = __dadd_rn(-v2, __dmul_rn(0.0, f));
    R[0][2] = __dadd_rn(v1, __dmul_rn(0.0, f));
    R[1][0] = __dadd_rn(v2, __dmul_rn(0.0, f));
    R[1][1] = __dadd_rn(1.0, __dmul_rn(k11, f));
    R[1][2] = __dadd_rn(0.0, __dmul_rn(k12, f));
    R[2][0] = __dadd_rn(-v1, __dmul_rn(0.0, f));
    R[2][1] = __dadd_rn(0.0, __dmul_rn(k12, f));
    R[2][2] = __dadd_rn(1.0, __dmul_rn(k22, f));
}

__global__ __launch_bounds__(256) void gather_rotate_kernel(const double* __restrict__ cloud, int64_t n,
                                                            const double* __restrict__ queries, int64_t b,
                                                            const int64_t* __restrict__ idx, int k,
                                                            const float* __restrict__ normals,
                                                            float* __restrict__ patch) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= b * k) return;
    const int64_t qi = t / k;
    int64_t pi = idx[t];
    pi = pi < 0 ? 0 : (pi >= n ? n - 1 : pi);
    const double px = __dsub_rn(cloud[pi * 3 + 0], queries[qi * 3 + 0]);
    const double py = __dsub_rn(cloud[pi * 3 + 1], queries[qi * 3 + 1]);
    const double pz = __dsub_rn(cloud[pi * 3 + 2], queries[qi * 3 + 2]);
    float* o = patch + t * 3;
    if (normals == nullptr) {
        o[0] = (float)px;
        o[1] = (float)py;
        o[2] = (float)pz;
        return;
    }
    double R[3][3];
    rotation_to_x(normals + qi * 3, R);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        double acc = __dmul_rn(R[r][0], px);
        acc = __fma_rn(R[r][1], py, acc);
        acc = __fma_rn(R[r][2], pz, acc);
        o[r] = (float)acc;
    }
}

__global__ __launch_bounds__(256) void displace_kernel(const double* __restrict__ q, const float* __restrict__ nrm,
                                                       const float* __restrict__ d, int64_t b,
                                                       double* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= b * 3) return;
    const float prod = __fmul_rn(nrm[t], d[t / 3]);
    out[t] = __dadd_rn(q[t], (double)prod);
}

int launch_gather_rotate(const double* cloud, int64_t n, const double* q, int64_t b, const int64_t* idx, int k,
                         const float* normals, float* patch, hipStream_t st) {
    const int64_t total = b * k;
    if (total == 0) return SAPCU_OK;
    hipLaunchKernelGGL(gather_rotate_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, cloud, n, q,
                       b, idx, k, normals, patch);
    SAPCU_CHECK_LAUNCH();
    return SAPCU_OK;
}

int launch_displace(const double* q, const float* nrm, const float* d, int64_t b, double* out, hipStream_t st) {
    if (b == 0) return SAPCU_OK;
    hipLaunchKernelGGL(displace_kernel, dim3((unsigned)((b * 3 + 255) / 256)), dim3(256), 0, st, q, nrm, d, b, out);
    SAPCU_CHECK_LAUNCH();
    return SAPCU_OK;
}

}  // namespace sapcu
