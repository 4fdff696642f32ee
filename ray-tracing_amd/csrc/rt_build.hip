// rt_build.hip - the reference's BSP tree (a spatial-median kd-tree) built on the GPU.
//
// Restates BSPNode.build_bsp_tree / build_bsp_tree_rec (BSPNode.py:54-121) level by level instead of
// recursively.  The reference's rule for one node is
//     leaf  if  len(surfaces) <= 10  or  |big - small| < EPSILON                     (BSPNode.py:66-69)
//     axis = argmax(big - small);  cut = (big[axis] + small[axis]) / 2               (BSPNode.py:71-72)
//     a surface goes left if its big corner <= cut, right if its small corner >= cut,
//     to BOTH sides otherwise; the list order is kept on either side                 (BSPNode.py:78-86)
// All arithmetic is float64 with the same operations, so node bounds, leaf membership AND leaf order
// (which decides nearest-hit ties) are identical to the host builder's; tests/test_gpu_build.py checks
// the arrays bit for bit.
//
// One level = every node of that depth.  A level owns a reference list (primitive positions, segmented by
// node, reference order kept inside a segment) and runs
//     k_node_decide   leaf? / axis / cut per node                 -> scan over nodes  (children slots)
//     k_ref_classify  left / right / both per reference           -> scan over references (packed L|R counts)
//     k_node_children bounds and segment starts of the 2M children
//     k_ref_scatter   stable copy of the references into the children's segments
// then two sweeps turn the level arrays into the DFS pre-order layout of rt_scene_desc (left child = i + 1):
//     k_subtree_sizes (bottom-up)  nodes / leaf references below every node
//     k_assign        (top-down)   pre-order index and first leaf reference; writes the output rows
//     k_leaf_copy                  leaf segments -> leaf_prim
// The work is HBM-bound integer/index work (about 70 B per reference and level); nothing is reshaped
// into a GEMM.  Kernels are launched with one lane per reference (coalesced 4/8-byte streams; the only
// gathers are the two float64 corners of the reference's primitive along the node's axis).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rt_amd.h"

namespace {

constexpr int BT = 256;                 // threads per block everywhere
constexpr int SCAN_ITEMS = 8;           // consecutive items per thread in the scan
constexpr int SCAN_TILE = BT * SCAN_ITEMS;
constexpr int MAX_LEVELS = 512;         // a cell of 1e3 halves down to EPSILON = 1e-6 in ~90 levels
constexpr int64_t MAX_LEVEL_REFS = (int64_t)1 << 30;

struct LevelDev {
    int32_t n_nodes;
    uint32_t n_refs;
    double *lo, *hi;        // [n_nodes*3]
    double *where;          // [n_nodes] cut location (internal nodes)
    int32_t *axis;          // [n_nodes]
    int32_t *child;         // [n_nodes] left child's index in the next level (right = +1), -1 = leaf
    uint32_t *start;        // [n_nodes+1] segment starts in refs
    uint64_t *nscan;        // [n_nodes+1] exclusive count of internal nodes
    int32_t *refs;          // [n_refs] primitive positions (order of the caller's arrays)
    int32_t *ref_node;      // [n_refs] owning node (index in this level)
    uint64_t *rscan;        // [n_refs+1] exclusive counts, low word = copies going left, high word = right
    uint64_t *tiles;        // scan scratch
    int32_t *sub_nodes;     // [n_nodes] nodes in the subtree
    uint32_t *sub_refs;     // [n_nodes] leaf references in the subtree
    int32_t *pre;           // [n_nodes] pre-order index
    uint32_t *first;        // [n_nodes] first leaf reference of the subtree in DFS order
};

struct Out {
    double *lo, *hi;
    int32_t *right, *first, *count, *leaf_prim;
};

__device__ inline uint32_t lo32(uint64_t v) { return (uint32_t)v; }
__device__ inline uint32_t hi32(uint64_t v) { return (uint32_t)(v >> 32); }

// ---------------------------------------------------------------- root bounds (BSPNode.py:113-118)
__global__ void __launch_bounds__(BT) k_bounds_partial(const double *small, const double *big, int64_t n, double *partial) {
    double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = (int64_t)blockIdx.x * BT + threadIdx.x; i < n; i += (int64_t)gridDim.x * BT)
        for (int k = 0; k < 3; ++k) {
            mn[k] = fmin(mn[k], small[3 * i + k]);
            mx[k] = fmax(mx[k], big[3 * i + k]);
        }
    __shared__ double sh[BT / 64][6];
    for (int k = 0; k < 3; ++k)
        for (int d = 32; d; d >>= 1) {
            mn[k] = fmin(mn[k], __shfl_xor(mn[k], d));
            mx[k] = fmax(mx[k], __shfl_xor(mx[k], d));
        }
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 3; ++k) { sh[threadIdx.x >> 6][k] = mn[k]; sh[threadIdx.x >> 6][3 + k] = mx[k]; }
    __syncthreads();
    if (threadIdx.x < 6) {
        double v = sh[0][threadIdx.x];
        for (int w = 1; w < BT / 64; ++w) v = threadIdx.x < 3 ? fmin(v, sh[w][threadIdx.x]) : fmax(v, sh[w][threadIdx.x]);
        partial[6 * blockIdx.x + threadIdx.x] = v;
    }
}

__global__ void k_bounds_final(const double *partial, int nparts, const double *given_lo, const double *given_hi,
                               double *lo, double *hi) {
    // one wave per component: lanes stride over the partial results, then a butterfly
    const int k = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (k >= 6) return;
    double v = k < 3 ? INFINITY : -INFINITY;
    for (int b = lane; b < nparts; b += 64) v = k < 3 ? fmin(v, partial[6 * b + k]) : fmax(v, partial[6 * b + k]);
    for (int d = 32; d; d >>= 1) {
        const double o = __shfl_xor(v, d);
        v = k < 3 ? fmin(v, o) : fmax(v, o);
    }
    if (lane) return;
    if (k < 3) lo[k] = given_lo ? given_lo[k] : v;
    else hi[k - 3] = given_hi ? given_hi[k - 3] : v;
}

__global__ void __launch_bounds__(BT) k_init_refs(LevelDev L) {
    const uint32_t i = blockIdx.x * BT + threadIdx.x;
    if (i < L.n_refs) { L.refs[i] = (int32_t)i; L.ref_node[i] = 0; }
    if (i == 0) { L.start[0] = 0; L.start[1] = L.n_refs; L.pre[0] = 0; L.first[0] = 0; }
}

// ---------------------------------------------------------------- exclusive scan of uint64 (in place)
__global__ void __launch_bounds__(BT) k_scan_tiles(uint64_t *data, uint64_t *tile_sums, uint32_t n) {
    const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    uint64_t v[SCAN_ITEMS], run = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const uint64_t x = base + k < n ? data[base + k] : 0;
        v[k] = run;
        run += x;
    }
    uint64_t inc = run;                              // inclusive scan of the thread totals inside the wave
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t up = __shfl_up(inc, d);
        if (lane >= d) inc += up;
    }
    __shared__ uint64_t wsum[BT / 64];
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    uint64_t off = 0, total = 0;
#pragma unroll
    for (int w = 0; w < BT / 64; ++w) {
        if (w < wid) off += wsum[w];
        total += wsum[w];
    }
    const uint64_t excl = off + inc - run;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k)
        if (base + k < n) data[base + k] = excl + v[k];
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}

__global__ void __launch_bounds__(BT) k_scan_add(uint64_t *data, const uint64_t *tile_excl, uint32_t n) {
    const uint32_t i = blockIdx.x * BT + threadIdx.x;
    if (i < n) data[i] += tile_excl[i / SCAN_TILE];
}

// ---------------------------------------------------------------- one level
__global__ void __launch_bounds__(BT) k_node_decide(LevelDev L, int leaf_size, double eps) {
    const int32_t j = blockIdx.x * BT + threadIdx.x;
    if (j > L.n_nodes) return;
    if (j == L.n_nodes) { L.nscan[j] = 0; return; }
    const uint32_t cnt = L.start[j + 1] - L.start[j];
    const double e0 = L.hi[3 * j] - L.lo[3 * j], e1 = L.hi[3 * j + 1] - L.lo[3 * j + 1], e2 = L.hi[3 * j + 2] - L.lo[3 * j + 2];
    const double nrm = sqrt(e0 * e0 + e1 * e1 + e2 * e2);              // np.linalg.norm(diffs)
    const bool leaf = cnt <= (uint32_t)leaf_size || nrm < eps;
    int a = 0;                                                        // np.argmax: first index of the maximum
    double m = e0;
    if (e1 > m) { a = 1; m = e1; }
    if (e2 > m) { a = 2; }
    L.axis[j] = a;
    L.where[j] = (L.hi[3 * j + a] + L.lo[3 * j + a]) / 2;
    L.child[j] = leaf ? -1 : 0;
    L.nscan[j] = leaf ? 0 : 1;
}

__global__ void __launch_bounds__(BT) k_ref_classify(LevelDev L, const double *small, const double *big) {
    const uint32_t i = blockIdx.x * BT + threadIdx.x;
    if (i > L.n_refs) return;
    uint64_t f = 0;
    if (i < L.n_refs) {
        const int32_t j = L.ref_node[i];
        if (L.child[j] >= 0) {
            const int a = L.axis[j];
            const double w = L.where[j];
            const int64_t p = L.refs[i];
            const bool left_only = big[3 * p + a] <= w;
            const bool right_only = !left_only && small[3 * p + a] >= w;
            f = (uint64_t)(right_only ? 0 : 1) | ((uint64_t)(left_only ? 0 : 1) << 32);
        }
    }
    L.rscan[i] = f;
}

__global__ void __launch_bounds__(BT) k_node_children(LevelDev L, LevelDev N) {
    const int32_t j = blockIdx.x * BT + threadIdx.x;
    if (j == 0) {
        const uint64_t t = L.rscan[L.n_refs];
        N.start[N.n_nodes] = lo32(t) + hi32(t);
    }
    if (j >= L.n_nodes || L.child[j] < 0) return;
    const int32_t c = 2 * (int32_t)L.nscan[j];
    L.child[j] = c;
    const uint64_t s = L.rscan[L.start[j]], e = L.rscan[L.start[j + 1]];
    const uint32_t base = lo32(s) + hi32(s), n_left = lo32(e) - lo32(s);
    N.start[c] = base;
    N.start[c + 1] = base + n_left;
    const int a = L.axis[j];
    const double w = L.where[j];
    for (int k = 0; k < 3; ++k) {
        const double l = L.lo[3 * j + k], h = L.hi[3 * j + k];
        N.lo[3 * c + k] = l;
        N.hi[3 * c + k] = k == a ? w : h;            // BSPNode.py:88-90
        N.lo[3 * (c + 1) + k] = k == a ? w : l;      // BSPNode.py:92-94
        N.hi[3 * (c + 1) + k] = h;
    }
}

__global__ void __launch_bounds__(BT) k_ref_scatter(LevelDev L, LevelDev N) {
    const uint32_t i = blockIdx.x * BT + threadIdx.x;
    if (i >= L.n_refs) return;
    const int32_t j = L.ref_node[i];
    const int32_t c = L.child[j];
    if (c < 0) return;
    const uint64_t me = L.rscan[i], nx = L.rscan[i + 1], s = L.rscan[L.start[j]];
    const int32_t p = L.refs[i];
    if (lo32(nx) != lo32(me)) {
        const uint32_t pos = N.start[c] + (lo32(me) - lo32(s));
        N.refs[pos] = p;
        N.ref_node[pos] = c;
    }
    if (hi32(nx) != hi32(me)) {
        const uint32_t pos = N.start[c + 1] + (hi32(me) - hi32(s));
        N.refs[pos] = p;
        N.ref_node[pos] = c + 1;
    }
}

// ---------------------------------------------------------------- DFS pre-order layout
__global__ void __launch_bounds__(BT) k_subtree_sizes(LevelDev L, LevelDev N) {
    const int32_t j = blockIdx.x * BT + threadIdx.x;
    if (j >= L.n_nodes) return;
    const int32_t c = L.child[j];
    if (c < 0) {
        L.sub_nodes[j] = 1;
        L.sub_refs[j] = L.start[j + 1] - L.start[j];
    } else {
        L.sub_nodes[j] = 1 + N.sub_nodes[c] + N.sub_nodes[c + 1];
        L.sub_refs[j] = N.sub_refs[c] + N.sub_refs[c + 1];
    }
}

__global__ void __launch_bounds__(BT) k_assign(LevelDev L, LevelDev N, Out o) {
    const int32_t j = blockIdx.x * BT + threadIdx.x;
    if (j >= L.n_nodes) return;
    const int32_t me = L.pre[j], c = L.child[j];
    const uint32_t first = L.first[j];
    for (int k = 0; k < 3; ++k) { o.lo[3 * (int64_t)me + k] = L.lo[3 * j + k]; o.hi[3 * (int64_t)me + k] = L.hi[3 * j + k]; }
    if (c < 0) {
        o.right[me] = -1;
        o.first[me] = (int32_t)first;
        o.count[me] = (int32_t)(L.start[j + 1] - L.start[j]);
        return;
    }
    const int32_t right = me + 1 + N.sub_nodes[c];
    N.pre[c] = me + 1;
    N.first[c] = first;
    N.pre[c + 1] = right;
    N.first[c + 1] = first + N.sub_refs[c];
    o.right[me] = right;
    o.first[me] = -1;
    o.count[me] = 0;
}

__global__ void __launch_bounds__(BT) k_leaf_copy(LevelDev L, Out o) {
    const uint32_t i = blockIdx.x * BT + threadIdx.x;
    if (i >= L.n_refs) return;
    const int32_t j = L.ref_node[i];
    if (L.child[j] >= 0) return;
    o.leaf_prim[L.first[j] + (i - L.start[j])] = L.refs[i];
}

thread_local std::string g_build_error;

inline unsigned blocks_for(uint64_t n) { return (unsigned)((n + BT - 1) / BT); }
inline size_t up256(size_t b) { return (b + 255) & ~(size_t)255; }

}  // namespace

struct rt_bsp {
    int device = 0;
    std::string error;
    hipStream_t stream = nullptr;
    std::vector<void *> allocs;
    Out out{};
    rt_bsp_stats st{};
};

namespace {

int bfail(rt_bsp *t, int code, const std::string &msg) {
    if (t) t->error = msg;
    g_build_error = msg;
    return code;
}

#define HIP_B(expr)                                                                                          \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess)                                                                                \
            return bfail(t, e_ == hipErrorOutOfMemory ? RT_ERR_OOM : RT_ERR_HIP,                             \
                         std::string(#expr) + ": " + hipGetErrorString(e_));                                 \
    } while (0)

// in-place exclusive scan of data[0..n); `tiles` has room for every level of tile sums
int scan_u64(rt_bsp *t, uint64_t *data, uint64_t *tiles, uint32_t n) {
    const uint32_t nt = (n + SCAN_TILE - 1) / SCAN_TILE;
    hipLaunchKernelGGL(k_scan_tiles, dim3(nt), dim3(BT), 0, t->stream, data, tiles, n);
    t->st.launches++;
    if (nt > 1) {
        if (int rc = scan_u64(t, tiles, tiles + nt, nt)) return rc;
        hipLaunchKernelGGL(k_scan_add, dim3(blocks_for(n)), dim3(BT), 0, t->stream, data, tiles, n);
        t->st.launches++;
    }
    HIP_B(hipGetLastError());
    return RT_OK;
}

size_t scan_scratch_items(uint64_t n) {
    size_t total = 0;
    while (true) {
        const uint64_t nt = (n + SCAN_TILE - 1) / SCAN_TILE;
        total += nt;
        if (nt <= 1) break;
        n = nt;
    }
    return total + 1;
}

int alloc_level(rt_bsp *t, LevelDev &L, int32_t n_nodes, uint32_t n_refs) {
    L = LevelDev{};
    L.n_nodes = n_nodes;
    L.n_refs = n_refs;
    const size_t nn = (size_t)n_nodes, nr = (size_t)n_refs;
    const size_t tiles = scan_scratch_items(std::max<uint64_t>(nn + 1, nr + 1));
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off += up256(bytes); return o; };
    const size_t o_lo = carve(nn * 24), o_hi = carve(nn * 24), o_where = carve(nn * 8), o_axis = carve(nn * 4),
                 o_child = carve(nn * 4), o_start = carve((nn + 1) * 4), o_nscan = carve((nn + 1) * 8),
                 o_refs = carve(nr * 4), o_rnode = carve(nr * 4), o_rscan = carve((nr + 1) * 8), o_tiles = carve(tiles * 8),
                 o_sn = carve(nn * 4), o_sr = carve(nn * 4), o_pre = carve(nn * 4), o_first = carve(nn * 4);
    char *base = nullptr;
    HIP_B(hipMalloc((void **)&base, off ? off : 256));
    t->allocs.push_back(base);
    t->st.bytes_scratch += (int64_t)off;
    L.lo = (double *)(base + o_lo); L.hi = (double *)(base + o_hi); L.where = (double *)(base + o_where);
    L.axis = (int32_t *)(base + o_axis); L.child = (int32_t *)(base + o_child); L.start = (uint32_t *)(base + o_start);
    L.nscan = (uint64_t *)(base + o_nscan); L.refs = (int32_t *)(base + o_refs); L.ref_node = (int32_t *)(base + o_rnode);
    L.rscan = (uint64_t *)(base + o_rscan); L.tiles = (uint64_t *)(base + o_tiles); L.sub_nodes = (int32_t *)(base + o_sn);
    L.sub_refs = (uint32_t *)(base + o_sr); L.pre = (int32_t *)(base + o_pre); L.first = (uint32_t *)(base + o_first);
    return RT_OK;
}

void release(rt_bsp *t, bool keep_outputs) {
    // outputs are the LAST allocation when present
    const size_t keep = keep_outputs && !t->allocs.empty() ? 1 : 0;
    for (size_t i = 0; i + keep < t->allocs.size(); ++i) (void)hipFree(t->allocs[i]);
    if (keep) { void *o = t->allocs.back(); t->allocs.clear(); t->allocs.push_back(o); }
    else t->allocs.clear();
}

int build(rt_bsp *t, int64_t n, const double *small, const double *big, const double *root_lo, const double *root_hi,
          int leaf_size, double eps) {
    const auto wall0 = std::chrono::steady_clock::now();
    HIP_B(hipSetDevice(t->device));
    HIP_B(hipStreamCreateWithFlags(&t->stream, hipStreamNonBlocking));
    hipEvent_t ev0, ev1;
    HIP_B(hipEventCreate(&ev0));
    HIP_B(hipEventCreate(&ev1));

    // corners (+ optional root bounds) -> HBM
    double *d_small = nullptr, *d_big = nullptr, *d_given = nullptr, *d_partial = nullptr;
    const size_t cb = (size_t)n * 24;
    HIP_B(hipMalloc((void **)&d_small, cb)); t->allocs.push_back(d_small);
    HIP_B(hipMalloc((void **)&d_big, cb)); t->allocs.push_back(d_big);
    const int nparts = (int)std::min<int64_t>(1024, (n + BT - 1) / BT);
    HIP_B(hipMalloc((void **)&d_partial, (size_t)nparts * 48)); t->allocs.push_back(d_partial);
    HIP_B(hipMalloc((void **)&d_given, 48)); t->allocs.push_back(d_given);
    HIP_B(hipMemcpyAsync(d_small, small, cb, hipMemcpyHostToDevice, t->stream));
    HIP_B(hipMemcpyAsync(d_big, big, cb, hipMemcpyHostToDevice, t->stream));
    if (root_lo) HIP_B(hipMemcpyAsync(d_given, root_lo, 24, hipMemcpyHostToDevice, t->stream));
    if (root_hi) HIP_B(hipMemcpyAsync(d_given + 3, root_hi, 24, hipMemcpyHostToDevice, t->stream));
    t->st.bytes_scratch += (int64_t)(2 * cb);

    std::vector<LevelDev> lv;
    lv.reserve(64);
    lv.emplace_back();
    if (int rc = alloc_level(t, lv[0], 1, (uint32_t)n)) return rc;
    HIP_B(hipEventRecord(ev0, t->stream));
    hipLaunchKernelGGL(k_bounds_partial, dim3(nparts), dim3(BT), 0, t->stream, d_small, d_big, n, d_partial);
    hipLaunchKernelGGL(k_bounds_final, dim3(1), dim3(6 * 64), 0, t->stream, d_partial, nparts, root_lo ? d_given : nullptr,
                       root_hi ? d_given + 3 : nullptr, lv[0].lo, lv[0].hi);
    hipLaunchKernelGGL(k_init_refs, dim3(blocks_for(n)), dim3(BT), 0, t->stream, lv[0]);
    t->st.launches += 3;

    int64_t total_nodes = 1;
    for (int d = 0;; ++d) {
        if (d >= MAX_LEVELS) return bfail(t, RT_ERR_UNSUPPORTED, "tree deeper than 512 levels");
        LevelDev &L = lv[d];
        t->st.ref_visits += L.n_refs;
        hipLaunchKernelGGL(k_node_decide, dim3(blocks_for((uint64_t)L.n_nodes + 1)), dim3(BT), 0, t->stream, L, leaf_size, eps);
        t->st.launches++;
        if (int rc = scan_u64(t, L.nscan, L.tiles, (uint32_t)L.n_nodes + 1)) return rc;
        hipLaunchKernelGGL(k_ref_classify, dim3(blocks_for((uint64_t)L.n_refs + 1)), dim3(BT), 0, t->stream, L, d_small, d_big);
        t->st.launches++;
        if (int rc = scan_u64(t, L.rscan, L.tiles, L.n_refs + 1)) return rc;
        uint64_t h_counts[2];       // internal nodes of this level, packed L|R reference copies
        HIP_B(hipMemcpyAsync(&h_counts[0], L.nscan + L.n_nodes, 8, hipMemcpyDeviceToHost, t->stream));
        HIP_B(hipMemcpyAsync(&h_counts[1], L.rscan + L.n_refs, 8, hipMemcpyDeviceToHost, t->stream));
        HIP_B(hipStreamSynchronize(t->stream));
        const uint64_t internal = h_counts[0];
        const uint64_t next_refs = (h_counts[1] & 0xffffffffu) + (h_counts[1] >> 32);
        if (internal == 0) break;
        total_nodes += 2 * (int64_t)internal;
        if (next_refs > (uint64_t)MAX_LEVEL_REFS || total_nodes > ((int64_t)1 << 30))
            return bfail(t, RT_ERR_UNSUPPORTED, "tree exceeds 2^30 nodes or references on one level");
        lv.emplace_back();
        LevelDev &N = lv[d + 1];
        if (int rc = alloc_level(t, N, (int32_t)(2 * internal), (uint32_t)next_refs)) return rc;
        LevelDev &Lr = lv[d];
        hipLaunchKernelGGL(k_node_children, dim3(blocks_for((uint64_t)Lr.n_nodes)), dim3(BT), 0, t->stream, Lr, N);
        if (Lr.n_refs) hipLaunchKernelGGL(k_ref_scatter, dim3(blocks_for(Lr.n_refs)), dim3(BT), 0, t->stream, Lr, N);
        t->st.launches += 2;
    }
    const int depth = (int)lv.size();

    for (int d = depth - 1; d >= 0; --d) {
        hipLaunchKernelGGL(k_subtree_sizes, dim3(blocks_for((uint64_t)lv[d].n_nodes)), dim3(BT), 0, t->stream, lv[d],
                           d + 1 < depth ? lv[d + 1] : lv[d]);
        t->st.launches++;
    }
    int32_t h_nodes = 0;
    uint32_t h_refs = 0;
    HIP_B(hipMemcpyAsync(&h_nodes, lv[0].sub_nodes, 4, hipMemcpyDeviceToHost, t->stream));
    HIP_B(hipMemcpyAsync(&h_refs, lv[0].sub_refs, 4, hipMemcpyDeviceToHost, t->stream));
    HIP_B(hipStreamSynchronize(t->stream));
    if (h_refs > 0x7fffffffu) return bfail(t, RT_ERR_UNSUPPORTED, "more than 2^31 leaf references");

    {   // outputs: one allocation, pushed last so that release() can keep it
        const size_t nn = (size_t)h_nodes, nr = (size_t)h_refs;
        size_t off = 0;
        auto carve = [&](size_t bytes) { size_t o = off; off += up256(bytes); return o; };
        const size_t o_lo = carve(nn * 24), o_hi = carve(nn * 24), o_r = carve(nn * 4), o_f = carve(nn * 4), o_c = carve(nn * 4),
                     o_l = carve(nr * 4);
        char *base = nullptr;
        HIP_B(hipMalloc((void **)&base, off ? off : 256));
        t->allocs.push_back(base);
        t->out.lo = (double *)(base + o_lo); t->out.hi = (double *)(base + o_hi); t->out.right = (int32_t *)(base + o_r);
        t->out.first = (int32_t *)(base + o_f); t->out.count = (int32_t *)(base + o_c); t->out.leaf_prim = (int32_t *)(base + o_l);
    }
    for (int d = 0; d < depth; ++d) {
        hipLaunchKernelGGL(k_assign, dim3(blocks_for((uint64_t)lv[d].n_nodes)), dim3(BT), 0, t->stream, lv[d],
                           d + 1 < depth ? lv[d + 1] : lv[d], t->out);
        if (lv[d].n_refs) hipLaunchKernelGGL(k_leaf_copy, dim3(blocks_for(lv[d].n_refs)), dim3(BT), 0, t->stream, lv[d], t->out);
        t->st.launches += 2;
    }
    HIP_B(hipEventRecord(ev1, t->stream));
    HIP_B(hipStreamSynchronize(t->stream));
    HIP_B(hipGetLastError());
    float ms = 0;
    HIP_B(hipEventElapsedTime(&ms, ev0, ev1));
    (void)hipEventDestroy(ev0);
    (void)hipEventDestroy(ev1);
    t->st.ms_device = ms;
    t->st.n_nodes = h_nodes;
    t->st.n_leaf_refs = (int32_t)h_refs;
    t->st.depth = depth;
    release(t, true);
    t->st.ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
    return RT_OK;
}

}  // namespace

extern "C" const char *rt_bsp_last_error(const rt_bsp *t) { return t ? t->error.c_str() : g_build_error.c_str(); }

extern "C" int rt_bsp_destroy(rt_bsp *t) {
    if (!t) return RT_OK;
    (void)hipSetDevice(t->device);
    release(t, false);
    if (t->stream) (void)hipStreamDestroy(t->stream);
    delete t;
    return RT_OK;
}

extern "C" int rt_bsp_build(int device, int64_t n, const double *small_corner, const double *big_corner, const double *root_lo,
                            const double *root_hi, int leaf_size, double epsilon, rt_bsp **out) {
    if (!out) return bfail(nullptr, RT_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (n <= 0) return bfail(nullptr, RT_ERR_INVALID_ARG, "no finite primitives (the reference's np.min over an empty list raises too)");
    if (n >= MAX_LEVEL_REFS) return bfail(nullptr, RT_ERR_UNSUPPORTED, "more than 2^30 primitives");
    if (!small_corner || !big_corner) return bfail(nullptr, RT_ERR_INVALID_ARG, "corner arrays are NULL");
    if (leaf_size < 0 || !(epsilon >= 0)) return bfail(nullptr, RT_ERR_INVALID_ARG, "leaf_size / epsilon out of range");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return bfail(nullptr, RT_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= ndev) return bfail(nullptr, RT_ERR_NO_DEVICE, "device ordinal out of range");
    rt_bsp *t = new rt_bsp();
    t->device = device;
    const int rc = build(t, n, small_corner, big_corner, root_lo, root_hi, leaf_size, epsilon);
    if (rc != RT_OK) {
        g_build_error = t->error;
        rt_bsp_destroy(t);
        return rc;
    }
    *out = t;
    return RT_OK;
}

extern "C" int rt_bsp_stats_get(const rt_bsp *t, rt_bsp_stats *st) {
    if (!t || !st) return bfail(nullptr, RT_ERR_INVALID_ARG, "NULL argument");
    *st = t->st;
    return RT_OK;
}

extern "C" int rt_bsp_fetch(rt_bsp *t, double *node_lo, double *node_hi, int32_t *node_right, int32_t *node_first,
                            int32_t *node_count, int32_t *leaf_prim) {
    if (!t) return bfail(nullptr, RT_ERR_INVALID_ARG, "tree is NULL");
    HIP_B(hipSetDevice(t->device));
    const size_t nn = (size_t)t->st.n_nodes, nr = (size_t)t->st.n_leaf_refs;
    if (node_lo) HIP_B(hipMemcpyAsync(node_lo, t->out.lo, nn * 24, hipMemcpyDeviceToHost, t->stream));
    if (node_hi) HIP_B(hipMemcpyAsync(node_hi, t->out.hi, nn * 24, hipMemcpyDeviceToHost, t->stream));
    if (node_right) HIP_B(hipMemcpyAsync(node_right, t->out.right, nn * 4, hipMemcpyDeviceToHost, t->stream));
    if (node_first) HIP_B(hipMemcpyAsync(node_first, t->out.first, nn * 4, hipMemcpyDeviceToHost, t->stream));
    if (node_count) HIP_B(hipMemcpyAsync(node_count, t->out.count, nn * 4, hipMemcpyDeviceToHost, t->stream));
    if (leaf_prim && nr) HIP_B(hipMemcpyAsync(leaf_prim, t->out.leaf_prim, nr * 4, hipMemcpyDeviceToHost, t->stream));
    HIP_B(hipStreamSynchronize(t->stream));
    return RT_OK;
}
