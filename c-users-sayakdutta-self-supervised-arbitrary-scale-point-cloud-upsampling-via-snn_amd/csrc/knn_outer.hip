// Outer kNN over the input cloud: brute-force float64, exact top-k, one query per wavefront.
//
// Replaces sklearn KDTree(data).query(q, k) at /root/reference/generation.py:110,127,153.
//
// Layout / mapping (gfx950):
//   * a 256-thread workgroup = 4 wavefronts = 4 queries; the cloud streams through LDS in
//     1024-point tiles (SoA x|y|z, 24 KiB) filled by coalesced global reads shared by the 4 waves;
//   * each lane owns one point of a 64-point slab per iteration; the squared distance is
//     ((qx-px)^2 + (qy-py)^2) + (qz-pz)^2 in f64 with separately rounded mul/add (the reduced
//     distance sklearn accumulates — no FMA contraction, or indices would differ in the last ulp);
//   * the running top-k is a sorted list DISTRIBUTED ACROSS THE WAVE's lanes (entry p in lane p%64,
//     slot p/64; k <= 128), so an insertion is one ballot+popcount (position) and one lane shift —
//     no LDS, no divergence beyond the wave-uniform candidate loop;
//   * candidates are visited in ascending point index and compared with strict '<' against the
//     current k-th distance, which yields the (distance, index) ascending order of a stable sort;
//   * the FIRST 64-point slab does not go through 64 serial insertions (every one of them passes an empty list's
//     threshold): it is sorted across the lanes by a bitonic network on (distance, index) keys and becomes the list.
#include "common.h"

namespace sapcu {

constexpr int KNN_TILE = 1024;
constexpr int KNN_WAVES = 4;

struct TopSlot {
    double d;
    int i;
};

// Cross-lane moves without the LDS crossbar: ds_bpermute (what __shfl compiles to) has ~100 cycles of
// latency and an insertion chains ~15 of them; a wave-uniform source lane is a v_readlane, and "take the
// value of lane-1" is one DPP move (wave_shr:1, lane 0 keeps its own).
__device__ __forceinline__ int shr1_i(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ double shr1_d(double v) {
    return __hiloint2double(shr1_i(__double2hiint(v)), shr1_i(__double2loint(v)));
}
__device__ __forceinline__ double readlane_d(double v, int lane_uniform) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane_uniform),
                            __builtin_amdgcn_readlane(__double2loint(v), lane_uniform));
}

// TWO = the list needs the second slot per lane (k > 64); the model's k = 48 and the outlier filter's k = 30 run the
// one-slot instance, whose insertion is a third shorter.
template <bool TWO>
__global__ __launch_bounds__(256) void knn_outer_kernel(const double* __restrict__ cloud, int64_t n,
                                                        const double* __restrict__ queries, int64_t b, int k,
                                                        int64_t* __restrict__ idx_out,
                                                        double* __restrict__ dist_out,
                                                        float* __restrict__ patch_out) {
    __shared__ double tile[3][KNN_TILE];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t qi = (int64_t)blockIdx.x * KNN_WAVES + wave;
    const bool active = qi < b;
    double qx = 0, qy = 0, qz = 0;
    if (active) {
        qx = queries[qi * 3 + 0];
        qy = queries[qi * 3 + 1];
        qz = queries[qi * 3 + 2];
    }
    const double INF = __builtin_huge_val();
    TopSlot s0{INF, -1}, s1{INF, -1};
    double tau = INF;   // distance of list entry k-1 (wave-uniform)

    for (int64_t base = 0; base < n; base += KNN_TILE) {
        const int cnt = (int)((n - base) < KNN_TILE ? (n - base) : KNN_TILE);
        __syncthreads();
        for (int pt = threadIdx.x; pt < cnt; pt += 256) {      // one point per thread and pass (no index arithmetic per element)
            const double* c3 = cloud + (base + pt) * 3;
            tile[0][pt] = c3[0];
            tile[1][pt] = c3[1];
            tile[2][pt] = c3[2];
        }
        __syncthreads();
        if (!active) continue;
        for (int off = 0; off < cnt; off += 64) {
            const int j = off + lane;
            double d = INF;
            if (j < cnt) {
                const double dx = __dsub_rn(qx, tile[0][j]);
                const double dy = __dsub_rn(qy, tile[1][j]);
                const double dz = __dsub_rn(qz, tile[2][j]);
                d = __dmul_rn(dx, dx);
                d = __dadd_rn(d, __dmul_rn(dy, dy));
                d = __dadd_rn(d, __dmul_rn(dz, dz));
            }
            if (base == 0 && off == 0) {
                // the list is empty: entries 0..63 = this slab in (distance, index) order (lanes past the cloud: INF, last)
                int si = j < cnt ? j : 0x7fffffff;
#pragma unroll
                for (int kb = 2; kb <= 64; kb <<= 1) {
#pragma unroll
                    for (int jb = kb >> 1; jb > 0; jb >>= 1) {
                        const double od = __shfl_xor(d, jb);
                        const int oi = __shfl_xor(si, jb);
                        const bool own_less = d < od || (d == od && si < oi);
                        const bool want_less = ((lane & jb) == 0) == ((lane & kb) == 0);     // lower lane of an ascending pair
                        if (own_less != want_less) {
                            d = od;
                            si = oi;
                        }
                    }
                }
                s0.d = d;
                s0.i = si == 0x7fffffff ? -1 : si;
                if (!TWO || k <= 64) tau = readlane_d(s0.d, (k - 1) & 63);      // k <= 64: the list is full (INF while cnt < k)
                continue;
            }
            unsigned long long mask = __ballot(d < tau);
            while (mask) {
                const int src = __builtin_ctzll(mask);
                mask &= mask - 1;
                const double cd = readlane_d(d, src);
                if (!(cd < tau)) continue;
                const int ci = (int)(base + off + src);
                // position = number of entries <= candidate (all have smaller point index)
                int pos = __popcll(__ballot(s0.d <= cd));
                if (TWO) pos += __popcll(__ballot(s1.d <= cd));
                // shift entries at positions >= pos up by one, drop the last
                const double u0d = shr1_d(s0.d);
                const int u0i = shr1_i(s0.i);
                const int p0 = lane, p1 = lane + 64;
                double l63d = 0.0;                                  // entry 63 moves into the second slot's lane 0
                int l63i = 0;
                if (TWO) {
                    l63d = readlane_d(s0.d, 63);
                    l63i = __builtin_amdgcn_readlane(s0.i, 63);
                }
                if (p0 == pos) {
                    s0.d = cd;
                    s0.i = ci;
                } else if (p0 > pos) {
                    s0.d = u0d;
                    s0.i = u0i;
                }
                if (TWO) {
                    double u1d = shr1_d(s1.d);
                    int u1i = shr1_i(s1.i);
                    if (lane == 0) {
                        u1d = l63d;
                        u1i = l63i;
                    }
                    if (p1 == pos) {
                        s1.d = cd;
                        s1.i = ci;
                    } else if (p1 > pos) {
                        s1.d = u1d;
                        s1.i = u1i;
                    }
                }
                // new k-th distance
                const int kl = (k - 1) & 63;
                tau = (!TWO || (k - 1) < 64) ? readlane_d(s0.d, kl) : readlane_d(s1.d, kl);
            }
        }
    }
    if (!active) return;
#pragma unroll
    for (int slot = 0; slot < 2; ++slot) {
        const int p = lane + 64 * slot;
        if (p >= k) continue;
        const TopSlot s = slot == 0 ? s0 : s1;
        idx_out[qi * k + p] = s.i;
        if (dist_out) dist_out[qi * k + p] = sqrt_cr(s.d);
        if (patch_out) {
            const double* pp = cloud + (int64_t)s.i * 3;
            float* o = patch_out + (qi * k + p) * 3;
            o[0] = (float)__dsub_rn(pp[0], qx);
            o[1] = (float)__dsub_rn(pp[1], qy);
            o[2] = (float)__dsub_rn(pp[2], qz);
        }
    }
}

int launch_knn_outer(const double* cloud, int64_t n, const double* q, int64_t b, int k, int64_t* idx,
                     double* dist, float* patch, hipStream_t st) {
    if (b == 0) return SAPCU_OK;
    const int64_t grid = (b + KNN_WAVES - 1) / KNN_WAVES;
    if (k > 64)
        hipLaunchKernelGGL(knn_outer_kernel<true>, dim3((unsigned)grid), dim3(256), 0, st, cloud, n, q, b, k, idx, dist, patch);
    else
        hipLaunchKernelGGL(knn_outer_kernel<false>, dim3((unsigned)grid), dim3(256), 0, st, cloud, n, q, b, k, idx, dist, patch);
    SAPCU_CHECK_LAUNCH();
    return SAPCU_OK;
}

}  // namespace sapcu
