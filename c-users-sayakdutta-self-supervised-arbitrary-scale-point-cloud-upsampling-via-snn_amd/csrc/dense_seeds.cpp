// Seed generation, in process (host C++): the first "next" row of SURVEY.md §8f.
//
// Replaces the reference's `./dense <cell> <n>` subprocess and its two text files
// (/root/reference/generation.py:112-118 -> /root/reference/dense.cpp:175-252): a breadth-first flood over
// the 1/cell^3 voxel grid, started at the voxels that hold input points, that emits the centre of every
// visited voxel whose distance to a local triangle fan of the input cloud lies in [0.011, 0.015] and does
// not expand voxels farther than 0.015.  The ORDER of the emitted seeds is part of the contract (it decides
// the batch composition downstream), so the traversal keeps the reference's FIFO order and 6-neighbour
// order (+x,-x,+y,-y,+z,-z), its integer voxel-key arithmetic (including keys that leave the grid), and its
// quirks:
//   * the neighbour search runs over n+1 points — the reference builds its tree over po[0..n] inclusive,
//     i.e. with one extra all-zero point (dense.cpp:193 with the array of :64);
//   * the 10 nearest points come out of a max-heap, farthest first: pt[0..7] are fan vertices, pt[8] and
//     pt[9] (the two nearest) the shared edge (dense.cpp:214-225);
//   * seeds are written with "%lf" (6 decimals) and read back by np.loadtxt: the values handed on are the
//     6-decimal roundings — reproduced here with the same printf/strtod round trip.
// The exact 10-nearest search is a k-d tree of our own (any exact search returns the same set); distances
// are accumulated x, y, z in f64 like dense.cpp:94-96; closest-point-on-triangle is the classic Voronoi-
// region test (Ericson, Real-Time Collision Detection §5.1.5) in the operation order of dense.cpp:130-173.
// Built with g++ -O2 (no FMA contraction on x86-64), like oracle/_ref/dense which the parity test runs.
//
// Threads: a voxel's distance is a pure function of its key, and the flood's bookkeeping (FIFO order, visited set,
// emission) costs little next to the 10-NN search + 8 triangle tests.  So the flood runs level by level: the
// distances of every key waiting in the queue are computed by a pool of host threads (SAPCU_SEED_THREADS, default
// = the CPUs this process may run on, at most 16), then the queue is consumed sequentially with the reference's
// logic.  Same values, same order — only the wall time changes (385 k seeds: 2.4 s on one core).
#include <sched.h>

#include <algorithm>
#include <thread>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <queue>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../include/sapcu.h"

namespace {

struct P3 {
    double v[3];
};

inline P3 sub(const P3& a, const P3& b) { return {{a.v[0] - b.v[0], a.v[1] - b.v[1], a.v[2] - b.v[2]}}; }
inline P3 add(const P3& a, const P3& b) { return {{a.v[0] + b.v[0], a.v[1] + b.v[1], a.v[2] + b.v[2]}}; }
inline P3 mul(const P3& a, double s) { return {{a.v[0] * s, a.v[1] * s, a.v[2] * s}}; }
inline P3 divs(const P3& a, double s) { return {{a.v[0] / s, a.v[1] / s, a.v[2] / s}}; }
inline double dot(const P3& a, const P3& b) { return a.v[0] * b.v[0] + a.v[1] * b.v[1] + a.v[2] * b.v[2]; }
inline P3 cross(const P3& a, const P3& b) {
    return {{a.v[1] * b.v[2] - a.v[2] * b.v[1], a.v[2] * b.v[0] - a.v[0] * b.v[2], a.v[0] * b.v[1] - a.v[1] * b.v[0]}};
}
inline double dist(const P3& a, const P3& b) {
    return std::sqrt((a.v[0] - b.v[0]) * (a.v[0] - b.v[0]) + (a.v[1] - b.v[1]) * (a.v[1] - b.v[1]) +
                     (a.v[2] - b.v[2]) * (a.v[2] - b.v[2]));
}

// closest point of triangle (a,b,c) to p: vertex regions, edge regions, face (barycentric)
P3 closest_on_triangle(const P3& a, const P3& b, const P3& c, const P3& p) {
    const P3 ab = sub(b, a), ac = sub(c, a), bc = sub(c, b);
    const double snom = dot(sub(p, a), ab), sdenom = dot(sub(p, b), sub(a, b));
    const double tnom = dot(sub(p, a), ac), tdenom = dot(sub(p, c), sub(a, c));
    if (snom <= 0.0 && tnom <= 0.0) return a;
    const double unom = dot(sub(p, b), bc), udenom = dot(sub(p, c), sub(b, c));
    if (sdenom <= 0.0 && unom <= 0.0) return b;
    if (tdenom <= 0.0 && udenom <= 0.0) return c;
    const P3 n = cross(sub(b, a), sub(c, a));
    const double vc = dot(n, cross(sub(a, p), sub(b, p)));
    if (vc <= 0.0 && snom >= 0.0 && sdenom >= 0.0) return add(a, divs(mul(ab, snom), snom + sdenom));
    const double va = dot(n, cross(sub(b, p), sub(c, p)));
    if (va <= 0.0 && unom >= 0.0 && udenom >= 0.0) return add(b, divs(mul(bc, unom), unom + udenom));
    const double vb = dot(n, cross(sub(c, p), sub(a, p)));
    if (vb <= 0.0 && tnom >= 0.0 && tdenom >= 0.0) return add(a, divs(mul(ac, tnom), tnom + tdenom));
    const double u = va / (va + vb + vc);
    const double w2 = vb / (va + vb + vc);
    const double w3 = 1.0 - u - w2;
    return add(add(mul(a, u), mul(b, w2)), mul(c, w3));
}

// exact k-nearest search: implicit balanced k-d tree over an index permutation
struct KdTree {
    const std::vector<P3>& pts;
    std::vector<int> order;
    explicit KdTree(const std::vector<P3>& p) : pts(p), order(p.size()) {
        for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
        build(0, (int)order.size(), 0);
    }
    void build(int lo, int hi, int dim) {
        if (hi - lo <= 1) return;
        const int mid = (lo + hi) / 2;
        std::nth_element(order.begin() + lo, order.begin() + mid, order.begin() + hi,
                         [&](int a, int b) { return pts[a].v[dim] < pts[b].v[dim]; });
        build(lo, mid, (dim + 1) % 3);
        build(mid + 1, hi, (dim + 1) % 3);
    }
    // max-heap of (squared distance, point index), at most k entries
    typedef std::priority_queue<std::pair<double, int>> Heap;
    void query(const P3& q, int k, int lo, int hi, int dim, Heap& heap) const {
        if (hi <= lo) return;
        const int mid = (lo + hi) / 2;
        const P3& c = pts[order[mid]];
        const double delta = q.v[dim] - c.v[dim];
        const int nd = (dim + 1) % 3;
        if (delta < 0) query(q, k, lo, mid, nd, heap); else query(q, k, mid + 1, hi, nd, heap);
        double d = 0.0;
        for (int i = 0; i < 3; ++i) d += (c.v[i] - q.v[i]) * (c.v[i] - q.v[i]);
        if ((int)heap.size() < k) heap.push({d, order[mid]});
        else if (d < heap.top().first) { heap.pop(); heap.push({d, order[mid]}); }
        if ((int)heap.size() < k || delta * delta < heap.top().first) {
            if (delta < 0) query(q, k, mid + 1, hi, nd, heap); else query(q, k, lo, mid, nd, heap);
        }
    }
};

double six_decimals(double x) {   // the "%lf" -> loadtxt round trip of the reference's text hand-over
    char buf[64];
    std::snprintf(buf, sizeof(buf), "%lf", x);
    return std::strtod(buf, nullptr);
}

}  // namespace

namespace {

int seed_threads() {
    if (const char* e = std::getenv("SAPCU_SEED_THREADS")) {
        const int v = std::atoi(e);
        if (v >= 1) return v > 256 ? 256 : v;
    }
    cpu_set_t set;
    int n = 1;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
    return n < 1 ? 1 : (n > 16 ? 16 : n);
}

// distance of a voxel centre to the local triangle fan (dense.cpp:206-227)
double fan_distance(const KdTree& tree, const std::vector<P3>& pts, const P3& centre) {
    KdTree::Heap heap;
    tree.query(centre, 10, 0, (int)pts.size(), 0, heap);
    P3 near[10];
    int cnt = 0;
    for (; !heap.empty() && cnt < 10; ++cnt) {            // farthest first
        near[cnt] = pts[heap.top().second];
        heap.pop();
    }
    for (; cnt < 10; ++cnt) near[cnt] = {{0.0, 0.0, 0.0}};   // fewer than 10 points: the reference leaves zeros
    double best = 99999999999999.0;
    for (int i = 0; i < 8; ++i) {
        const double d = dist(closest_on_triangle(near[i], near[8], near[9], centre), centre);
        if (d < best) best = d;
    }
    return best;
}

inline P3 voxel_centre(int key, int boxsize, double cell, int* xyz) {
    int t = key;
    const int z = t % boxsize;
    t /= boxsize;
    const int y = t % boxsize;
    t /= boxsize;
    const int x = t;
    if (xyz) { xyz[0] = x; xyz[1] = y; xyz[2] = z; }
    return {{x * cell + 0.5 * cell - 0.5, y * cell + 0.5 * cell - 0.5, z * cell + 0.5 * cell - 0.5}};
}

}  // namespace

namespace {

// open-addressing set of the voxel keys met so far; replaces dense.cpp's std::map `ma` plus the duplicates its queue
// carries: a key enters the flood's queue when it is FIRST discovered (later pushes of a queued or processed key are
// no-ops in the reference too: it is processed at its first position in the queue, skipped afterwards).
struct KeySet {
    std::vector<int> keys;
    std::vector<uint8_t> full;
    size_t used = 0, mask = 0;
    explicit KeySet(size_t cap_pow2) : keys(cap_pow2), full(cap_pow2, 0), mask(cap_pow2 - 1) {}
    static size_t hash(int key) { return (size_t)(((uint32_t)key * 0x9E3779B97F4A7C15ull) >> 20); }
    bool insert(int key) {                                 // true if the key was new
        if ((used + 1) * 2 > keys.size()) grow();
        for (size_t i = hash(key) & mask;; i = (i + 1) & mask) {
            if (!full[i]) {
                full[i] = 1;
                keys[i] = key;
                ++used;
                return true;
            }
            if (keys[i] == key) return false;
        }
    }
    void grow() {
        std::vector<int> ok;
        std::vector<uint8_t> of;
        ok.swap(keys);
        of.swap(full);
        keys.resize(ok.size() * 2);
        full.assign(ok.size() * 2, 0);
        mask = keys.size() - 1;
        for (size_t j = 0; j < ok.size(); ++j)
            if (of[j]) {
                size_t i = hash(ok[j]) & mask;
                while (full[i]) i = (i + 1) & mask;
                full[i] = 1;
                keys[i] = ok[j];
            }
    }
};

}  // namespace

extern "C" int sapcu_dense_seeds_host(const double* cloud_host, int64_t n, double cell, double* seeds_out_host,
                                      int64_t capacity, int64_t* count_host) {
    if (!cloud_host || !count_host || n < 1 || !(cell > 0.0) || capacity < 0 || (capacity > 0 && !seeds_out_host))
        return SAPCU_ERR_ARG;
    std::vector<P3> pts((size_t)n + 1);                       // + the reference's extra all-zero point
    for (int64_t i = 0; i < n; ++i) pts[i] = {{cloud_host[3 * i], cloud_host[3 * i + 1], cloud_host[3 * i + 2]}};
    pts[n] = {{0.0, 0.0, 0.0}};
    const int boxsize = (int)std::round(1 / cell);
    KeySet met(1u << 16);
    std::vector<int> frontier, next;                          // one flood level: newly discovered keys in queue order
    for (int64_t i = 0; i < n; ++i) {
        const int key = std::floor(((pts[i].v[0] + 0.5) / cell)) * boxsize * boxsize +
                        std::floor(((pts[i].v[1] + 0.5) / cell)) * boxsize + std::floor(((pts[i].v[2] + 0.5) / cell));
        if (met.insert(key)) frontier.push_back(key);
    }
    const KdTree tree(pts);
    const int nthreads = seed_threads();
    static const int step[6][3] = {{1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
    int64_t count = 0;
    struct Computed {
        double best, c6[3];
    };
    std::vector<Computed> result;
    while (!frontier.empty()) {
        // 1. distances (and 6-decimal centres of band voxels) of this level, in parallel
        result.resize(frontier.size());
        auto work = [&](size_t lo, size_t hi) {
            for (size_t i = lo; i < hi; ++i) {
                const P3 centre = voxel_centre(frontier[i], boxsize, cell, nullptr);
                Computed& r = result[i];
                r.best = fan_distance(tree, pts, centre);
                if (r.best >= 0.0110 && r.best <= 0.0150)
                    for (int d = 0; d < 3; ++d) r.c6[d] = six_decimals(centre.v[d]);
            }
        };
        const int nt = (int)std::min<size_t>((size_t)nthreads, (frontier.size() + 255) / 256);
        if (nt <= 1) {
            work(0, frontier.size());
        } else {
            std::vector<std::thread> pool;
            const size_t per = (frontier.size() + nt - 1) / nt;
            for (int t = 0; t < nt; ++t) {
                const size_t lo = (size_t)t * per, hi = std::min(frontier.size(), lo + per);
                if (lo < hi) pool.emplace_back(work, lo, hi);
            }
            for (auto& th : pool) th.join();
        }
        // 2. the reference's sequential bookkeeping over this level, in queue order (dense.cpp:229-245)
        next.clear();
        for (size_t i = 0; i < frontier.size(); ++i) {
            const double best = result[i].best;
            if (best >= 0.0110 && best <= 0.0150) {
                if (count < capacity) {
                    seeds_out_host[3 * count] = result[i].c6[0];
                    seeds_out_host[3 * count + 1] = result[i].c6[1];
                    seeds_out_host[3 * count + 2] = result[i].c6[2];
                }
                ++count;
            } else if (best > 0.0150) {
                continue;
            }
            int c[3];
            voxel_centre(frontier[i], boxsize, cell, c);
            for (int d = 0; d < 6; ++d) {
                const int nkey = (c[0] + step[d][0]) * boxsize * boxsize + (c[1] + step[d][1]) * boxsize + (c[2] + step[d][2]);
                if (met.insert(nkey)) next.push_back(nkey);
            }
        }
        frontier.swap(next);
    }
    *count_host = count;
    return count <= capacity ? SAPCU_OK : SAPCU_ERR_WORKSPACE;
}
