// Farthest-point sampling to the target count — /root/reference/generate.py:56-74.
//
// The reference runs `npoint` sequential torch steps over the whole refined cloud (~385 k points): update every
// point's distance to the newest centroid (f32: ((dx^2 + dy^2) + dz^2), keep the minimum), take the arg-max.
// The loop is latency-bound (a step is ~1 MB of register-resident work), so ONE persistent launch does all
// steps: up to 256 workgroups (one per CU) keep their slice of the cloud and of the running distances IN
// REGISTERS for the whole run.  Per step a workgroup reduces its slice to one candidate
//     key = (distance bits << 32) | ~index          (largest distance wins, ties go to the smallest index —
//                                                     torch.max's documented rule)
// and PUBLISHES it together with the candidate's coordinates as five self-validating 64-bit words
// (step tag << 32 | payload) in a 2-row mailbox; every workgroup then polls all G mailboxes (one per thread) and
// reduces them itself.  One store + one polled load per step on the critical path: no atomics, no separate
// barrier, no re-read of the winner's coordinates.  A row is reused two steps later, which is safe because a
// workgroup can only publish step s+2 after every workgroup has published s+1, i.e. has finished reading s;
// a word from the row's previous use carries an older tag and is simply not accepted yet.
// Every workgroup of the grid must be resident (grid <= number of CUs); polls are bounded.
#include "common.h"

namespace sapcu {

constexpr unsigned FPS_SPIN_LIMIT = 1u << 22;   // ~seconds; a healthy step waits microseconds
constexpr int FPS_WORDS = 5;                    // dist, ~index, x, y, z — each tagged with step+1
constexpr int FPS_MAX_GRID = 256;               // one mailbox per polling thread

struct FpsCand {
    unsigned long long key;
    float x, y, z;
};

__device__ __forceinline__ unsigned long long fps_ld(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void fps_st(unsigned long long* p, unsigned tag, unsigned payload) {
    __hip_atomic_store(p, ((unsigned long long)tag << 32) | payload, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// all lanes end up with the wave's best candidate (keys are unique per point; 0 = "no point")
__device__ __forceinline__ FpsCand fps_wave_best(FpsCand c) {
    unsigned long long best = c.key;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const unsigned long long other = __shfl_xor(best, o);
        best = other > best ? other : best;
    }
    const unsigned long long owners = __ballot(c.key == best);
    const int src = __ffsll((long long)owners) - 1;      // several lanes only when all hold key 0
    FpsCand r;
    r.key = best;
    r.x = __shfl(c.x, src);
    r.y = __shfl(c.y, src);
    r.z = __shfl(c.z, src);
    return r;
}

// FPS_PPT = points per thread: a workgroup holds 256 * FPS_PPT points, the grid at most #CU workgroups
template <int FPS_PPT>
__global__ __launch_bounds__(256) void fps_kernel(const float* __restrict__ xyz, int64_t n, int npoint, int start,
                                                  int64_t* __restrict__ out, unsigned long long* __restrict__ mail,
                                                  int* __restrict__ err) {
    __shared__ FpsCand wbest[2][4];
    __shared__ int failed;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = gridDim.x;
    const int64_t per = (n + G - 1) / G;
    const int64_t base = (int64_t)blockIdx.x * per;
    const int64_t lim = (base + per) < n ? (base + per) : n;
    float px[FPS_PPT], py[FPS_PPT], pz[FPS_PPT], dist[FPS_PPT];
#pragma unroll
    for (int u = 0; u < FPS_PPT; ++u) {
        const int64_t i = base + tid + 256 * u;
        const bool ok = i < lim;
        px[u] = ok ? xyz[i * 3] : 0.f;
        py[u] = ok ? xyz[i * 3 + 1] : 0.f;
        pz[u] = ok ? xyz[i * 3 + 2] : 0.f;
        dist[u] = ok ? 1e32f : -1.f;          // torch.ones(N) * 1e32; padding (< 0) never produces a key
    }
    if (tid == 0) failed = 0;
    float cx = xyz[(int64_t)start * 3], cy = xyz[(int64_t)start * 3 + 1], cz = xyz[(int64_t)start * 3 + 2];
    if (blockIdx.x == 0 && tid == 0) out[0] = start;
    __syncthreads();
    for (int step = 0; step + 1 < npoint; ++step) {
        const unsigned tag = (unsigned)step + 1u;
        unsigned long long* row = mail + (size_t)(step & 1) * FPS_MAX_GRID * FPS_WORDS;
        FpsCand c = {0ull, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < FPS_PPT; ++u) {
            const int64_t i = base + tid + 256 * u;
            const float dx = __fsub_rn(px[u], cx), dy = __fsub_rn(py[u], cy), dz = __fsub_rn(pz[u], cz);
            const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
            if (d < dist[u]) dist[u] = d;
            if (dist[u] >= 0.f) {
                const unsigned long long key = ((unsigned long long)__float_as_uint(dist[u]) << 32) |
                                               (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
                if (key > c.key) { c.key = key; c.x = px[u]; c.y = py[u]; c.z = pz[u]; }
            }
        }
        c = fps_wave_best(c);
        if (lane == 0) wbest[0][wave] = c;
        __syncthreads();
        if (tid == 0) {                                   // publish this workgroup's candidate
            FpsCand b = wbest[0][0];
            for (int w = 1; w < 4; ++w) if (wbest[0][w].key > b.key) b = wbest[0][w];
            unsigned long long* m = row + (size_t)blockIdx.x * FPS_WORDS;
            fps_st(m + 0, tag, (unsigned)(b.key >> 32));
            fps_st(m + 1, tag, (unsigned)(b.key & 0xFFFFFFFFull));
            fps_st(m + 2, tag, __float_as_uint(b.x));
            fps_st(m + 3, tag, __float_as_uint(b.y));
            fps_st(m + 4, tag, __float_as_uint(b.z));
        }
        FpsCand r = {0ull, 0.f, 0.f, 0.f};
        if (tid < G) {                                    // collect: thread t waits for workgroup t's mailbox
            const unsigned long long* m = row + (size_t)tid * FPS_WORDS;
            unsigned long long w0, w1, w2, w3, w4;
            unsigned spins = 0;
            for (;;) {
                w0 = fps_ld(m); w1 = fps_ld(m + 1); w2 = fps_ld(m + 2); w3 = fps_ld(m + 3); w4 = fps_ld(m + 4);
                const bool ready = (unsigned)(w0 >> 32) == tag && (unsigned)(w1 >> 32) == tag && (unsigned)(w2 >> 32) == tag &&
                                   (unsigned)(w3 >> 32) == tag && (unsigned)(w4 >> 32) == tag;
                if (ready) break;
                if (++spins > FPS_SPIN_LIMIT) {           // a workgroup is not resident / died: give up, do not hang
                    atomicExch(err, 1);
                    failed = 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            r.key = (w0 << 32) | (w1 & 0xFFFFFFFFull);
            r.x = __uint_as_float((unsigned)w2);
            r.y = __uint_as_float((unsigned)w3);
            r.z = __uint_as_float((unsigned)w4);
        }
        r = fps_wave_best(r);
        if (lane == 0) wbest[1][wave] = r;
        __syncthreads();
        if (failed != 0) return;                          // uniform over the workgroup
        FpsCand b = wbest[1][0];
#pragma unroll
        for (int w = 1; w < 4; ++w) if (wbest[1][w].key > b.key) b = wbest[1][w];
        cx = b.x; cy = b.y; cz = b.z;
        if (blockIdx.x == 0 && tid == 0) out[step + 1] = (int64_t)(0xFFFFFFFFu - (unsigned)(b.key & 0xFFFFFFFFull));
    }
}

// 2 mailbox rows x 256 workgroups x 5 words, then the error flag
size_t fps_workspace_bytes(int) { return (size_t)2 * FPS_MAX_GRID * FPS_WORDS * 8 + 16; }

int launch_fps(const float* xyz, int64_t n, int npoint, int start, int64_t* out, void* ws, hipStream_t st) {
    if (npoint == 0) return SAPCU_OK;
    int dev = 0;
    SAPCU_CHECK_HIP(hipGetDevice(&dev));
    int cus = 0;
    SAPCU_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    if (cus <= 0) cus = 256;
    int64_t grid = (n + 255) / 256;              // small clouds: one point per thread; large ones: every CU
    if (grid > cus) grid = cus;
    if (grid > FPS_MAX_GRID) grid = FPS_MAX_GRID;
    if (grid < 1) grid = 1;
    const int64_t per = (n + grid - 1) / grid;   // points per workgroup
    SAPCU_CHECK_ARG(per <= 256 * 32, "fps: n=%lld exceeds %lld points (%lld resident workgroups x 8192 points)", (long long)n,
                    (long long)grid * 8192, (long long)grid);
    unsigned long long* mail = (unsigned long long*)ws;
    int* err = (int*)(mail + (size_t)2 * FPS_MAX_GRID * FPS_WORDS);
    SAPCU_CHECK_HIP(hipMemsetAsync(ws, 0, fps_workspace_bytes(npoint), st));
    const dim3 g((unsigned)grid), b(256);
    if (per <= 256 * 2) hipLaunchKernelGGL(fps_kernel<2>, g, b, 0, st, xyz, n, npoint, start, out, mail, err);
    else if (per <= 256 * 8) hipLaunchKernelGGL(fps_kernel<8>, g, b, 0, st, xyz, n, npoint, start, out, mail, err);
    else hipLaunchKernelGGL(fps_kernel<32>, g, b, 0, st, xyz, n, npoint, start, out, mail, err);
    SAPCU_CHECK_LAUNCH();
    return SAPCU_OK;
}

// reads the kernel's "a workgroup never arrived" flag; call after the stream has been synchronised
int fps_failed(const void* ws, int npoint, int* flag) {
    const char* p = (const char*)ws + (size_t)2 * FPS_MAX_GRID * FPS_WORDS * 8;
    SAPCU_CHECK_HIP(hipMemcpy(flag, p, sizeof(int), hipMemcpyDeviceToHost));
    return SAPCU_OK;
}

}  // namespace sapcu
