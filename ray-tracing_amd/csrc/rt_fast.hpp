// rt_fast.hpp - the fast nearest-hit query: float32 conservative culling + float64 decisions.
//
// The reference's query (BSPNode.py:123-210) is, per ray, "argmin over all primitives of every
// leaf whose box the ray enters of the Euclidean hit distance, first in DFS order wins exact
// ties".  Because the tree is a spatial-median kd-tree whose child boxes tile their parent
// (BSPNode.py:71-94), that result equals the nearest hit over ALL primitives (BSP ==
// brute force), so any traversal that (a) never skips a leaf holding a hit that could be nearest
// and (b) resolves exact distance ties the way the reference's visiting order does, returns the
// same answer.  This file does:
//
//   * kd traversal, near child first, one split plane per node (8-byte nodes), per-lane stack in
//     LDS ([level][lane] columns of 8-byte entries: conflict-free ds_read/write_b64);
//   * all culling in float32 with explicit error margins (a cull may only say "cannot matter");
//   * leaf entries carry a 32-byte float32 record (sphere centre as hi+lo float pair, so the
//     centre-origin difference has no cancellation error; box bounds rounded outward);
//   * every primitive that survives the float32 reject is decided by the float64 functions of
//     rt_device.hpp (reference operation order), and distances are compared in float64;
//   * an exact tie between two DIFFERENT primitives (coincident surfaces) raises `tie`; the caller
//     then re-runs the reference-order walk (nearest_hit_exact) for that ray - rare, exact.
//
// Shadow rays additionally stop at the shaded point: the lit test of Light.py:219-221 only asks
// whether the nearest hit lies within 1e-6 of P, so nodes entirely beyond P are skipped and the
// query ends as soon as one float64-confirmed hit is closer than |P-src| - 3e-6.
#pragma once
#include "rt_device.hpp"

struct KNode {      // 8 B.  internal: a = float bits of the cut, b = (right child << 2) | axis
    uint32_t a;     //       leaf:     a = first leaf entry,     b = (count << 2) | 3
    uint32_t b;
};
struct LeafF {      // 32 B, one per leaf entry, in leaf-list order
    float x, y, z;  // sphere: centre (float part)            box: lo rounded down
    float r;        // sphere: radius rounded up
    float lx, ly, lz;  // sphere: centre - (double)float part  box: hi rounded up
    uint32_t pk;    // (primitive number << 2) | kind (1 sphere, 2 box)
};

#define RT_2M18 3.8146973e-06f
#define RT_2M19 1.9073486e-06f
#define RT_2M20 9.5367432e-07f
#define RT_2M38 3.6379788e-12f

struct FastScene {
    const KNode *knodes;
    const LeafF *leaves;
    const uint32_t *prim_entry;     // primitive number -> one of its leaf entries (its LeafF record)
    float root_lo[3], root_hi[3];   // rounded outward
    float extent;                   // max |coordinate| of the root box, rounded up
    int32_t depth;                  // tree depth = stack entries needed per lane
    int32_t ok;                     // 0: tree is not an exact kd tiling (or too deep) -> use the exact walk
    // every cut location and root bound of the tree, per axis, rounded to float32, sorted, distinct (on_node_plane)
    const float *cuts[3];
    int32_t n_cuts[3];
    // the internal nodes of the first levels once more, breadth first, 16 bytes each {cut bits, axis, left ref, right ref}: the
    // nearest-hit kernels copy this table into LDS (rt_top_stage) and take the first steps of every descent from there - a
    // step is a dependent load, ~1 us from L2 / Infinity Cache against ~0.1 us from LDS, and most steps of a short ray are
    // its first descent.  A ref with RT_TOP_FLAG is an index into this table, any other the node's index in knodes.
    const uint4 *top;
    int32_t n_top;                  // 0: no table
    // Exact ties (two primitives at the same float64 distance: coplanar faces of touching boxes) go to the primitive the reference's
    // walk visits FIRST.  For a tame ray the walk reaches a leaf exactly when the leaf's own box passes the slab test - an ancestor's
    // box contains it bound for bound, and every step of the test (difference, product, minimum, maximum) is monotone in the bounds -
    // and it visits leaf entries in the order of the leaf list.  So the winner is the tied primitive whose first leaf entry in a
    // leaf that passes comes first: tie_ent[tie_off[p] .. tie_off[p + 1]) = {leaf entry, node} of primitive p, ascending.
    const uint32_t *tie_off;        // nullptr: ties take the reference-order walk
    const uint2 *tie_ent;
};
#define RT_TOP_FLAG 0x80000000u
#define RT_TOP_MAX_NODES 255        // 8 levels, 4 KB of LDS per workgroup

// copy the table into the workgroup's LDS (all threads; ends with a barrier).  ltop: RT_TOP_MAX_NODES entries of dynamic LDS
__device__ __forceinline__ void rt_top_stage(const FastScene &F, uint4 *ltop) {
    for (int i = (int)threadIdx.x; i < F.n_top; i += (int)blockDim.x) ltop[i] = F.top[i];
    __syncthreads();
}

// Could coordinate v (float64 -> float32) on axis ax be EXACTLY a bound of some node?  A ray that starts in such a plane
// with an exactly zero direction component along ax is not served by "tree walk == brute force": the reference's slab
// test evaluates 0 * inf = NaN there and culls the node with its whole subtree (Cube.py:120-136, BSPNode.py:132).
// Equality in float32 is necessary for equality in float64, so a "no" is certain.
__device__ __forceinline__ bool on_node_plane(const FastScene &F, int ax, float v) {
    const float *a = F.cuts[ax];
    int lo = 0, hi = F.n_cuts[ax];
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1;
        else hi = mid;
    }
    return lo < F.n_cuts[ax] && a[lo] == v;
}

#ifdef RT_TRAV_STATS
#define RT_TS(x) x
#else
#define RT_TS(x)
#endif
struct FastState {
    RT_TS(uint32_t n_node = 0; uint32_t n_leaf = 0; uint32_t n_batch = 0; uint32_t n_conf = 0;)
    RT_TS(unsigned long long c_desc = 0; unsigned long long c_test = 0; unsigned long long c_conf = 0; unsigned long long c_pop = 0;)      // clocks: descents, float32 leaf tests, float64 confirmations, stack pops
    Hit best;
    int32_t best_prim;       // primitive number of the best hit, -1: a plane (or none)
    int32_t tie_prim;        // tie: the other primitive at the same distance
    bool tie_many;           // ... and there is a third
    bool tie;
    bool stop;       // shadow early-out taken
    float tbest;     // float32 upper bound of the best hit's ray parameter
};

// candidate confirmed in float64: strict-< update on the Euclidean distance recomputed from the
// hit point (ray_functions.py:61-77); the far clip of Background (Background.py:23) is applied
// lazily - only hits farther than 1e8 can lose against it.
__device__ __forceinline__ void fast_consider(const DScene &S, FastState &st, d3 o, d3 d, double t, int32_t id,
                                              double early_dist, int32_t prim) {
    d3 p = o + t * d;
    if (nan3_any(p)) return;
    double dist = norm3(p - o);
    if (dist >= RT_MAX_RENDER_DISTANCE && S.far_clip) {
        double far = norm3(mk3(RT_MAX_RENDER_DISTANCE, RT_MAX_RENDER_DISTANCE, RT_MAX_RENDER_DISTANCE) - o);
        if (!(dist < far)) return;
    }
    if (dist < st.best.dist) {
        st.best.p = p;
        st.best.dist = dist;
        st.best.id = id;
        st.best_prim = prim;
        st.tie = false;
        st.tie_many = false;
        float tf = __double2float_ru(t);
        st.tbest = fmaf(fabsf(tf), RT_2M20, tf);
        if (dist < early_dist) st.stop = true;
    } else if (dist == st.best.dist && id != st.best.id) {
        // (a plane that is the best hit stays: the reference looks at the planes before the tree and replaces a hit only by a
        // strictly nearer one; the planes come first here as well, so a plane never ties with an earlier primitive)
        if (st.best_prim >= 0) {
            if (st.tie && prim != st.tie_prim) st.tie_many = true;
            if (!st.tie) st.tie_prim = prim;
            st.tie = true;
        }
    }
}

// float32 view of a ray, shared by the plane loop, the root interval and the tree walk
struct FastRay {
    float of0, of1, of2;        // origin (float part)
    float ol0, ol1, ol2;        // origin - (double)float part
    float df0, df1, df2;        // direction
    float in0, in1, in2;        // reciprocal direction, clamped to +-1e30
    float E0, E1, E2, Emax;     // absolute slack of a plane distance along each axis
    float af, dnf, K;           // |d|^2, |d| rounded up, magnitude bound of the coordinates involved
};
__device__ __forceinline__ FastRay fast_ray(const FastScene &F, d3 o, d3 d) {
    FastRay R;
    R.of0 = (float)o.x; R.of1 = (float)o.y; R.of2 = (float)o.z;
    R.ol0 = (float)(o.x - (double)R.of0); R.ol1 = (float)(o.y - (double)R.of1); R.ol2 = (float)(o.z - (double)R.of2);
    R.df0 = (float)d.x; R.df1 = (float)d.y; R.df2 = (float)d.z;
    // reciprocal clamped to +-1e30: a zero (or float-underflowed) direction component behaves like a ray
    // parallel to that axis, and every product below stays finite or overflows to a harmless +-inf
    R.in0 = fminf(fmaxf(1.0f / R.df0, -1e30f), 1e30f);
    R.in1 = fminf(fmaxf(1.0f / R.df1, -1e30f), 1e30f);
    R.in2 = fminf(fmaxf(1.0f / R.df2, -1e30f), 1e30f);
    R.K = fmaxf(fmaxf(fabsf(R.of0), fabsf(R.of1)), fmaxf(fabsf(R.of2), F.extent));
    const float Eabs = R.K * RT_2M19;
    R.E0 = Eabs * fabsf(R.in0); R.E1 = Eabs * fabsf(R.in1); R.E2 = Eabs * fabsf(R.in2);
    R.Emax = fmaxf(R.E0, fmaxf(R.E1, R.E2));
    R.af = fmaf(R.df0, R.df0, fmaf(R.df1, R.df1, R.df2 * R.df2));             // |d|^2 (float32)
    R.dnf = sqrtf(R.af) * 1.000001f;                                           // |d| rounded up
    return R;
}

// tlim: float32 upper bound of the ray parameters that can matter (+inf for ordinary rays)
__device__ __forceinline__ void fast_init(FastState &st, float tlim) {
    st.best.p = mk3(dnan(), dnan(), dnan());
    st.best.dist = dinf();
    st.best.id = 0;
    st.tie = false;
    st.tie_many = false;
    st.best_prim = -1;
    st.tie_prim = -1;
    st.stop = false;
    st.tbest = tlim;
}

// ---- planes (root list, few, wave-uniform): float32 reject, float64 decision -------------
__device__ __forceinline__ void fast_planes(const DScene &S, const FastRay &R, d3 o, d3 d, double early_dist, FastState &st) {
    for (int k = 0; k < S.n_planes; ++k) {
        const DPlane &pl = S.planes[k];
        const float n0 = (float)pl.n[0], n1 = (float)pl.n[1], n2 = (float)pl.n[2];
        float vn = fmaf(R.df0, n0, fmaf(R.df1, n1, R.df2 * n2));
        float evn = RT_2M20 * (fabsf(R.df0 * n0) + fabsf(R.df1 * n1) + fabsf(R.df2 * n2));
        if (vn > evn) continue;                                   // certainly facing away: one-sided planes miss
        if (st.tbest < __int_as_float(0x7f800000) && vn < -evn) { // segment end known: is the plane beyond it?
            float num = -(fmaf(R.of0, n0, fmaf(R.of1, n1, R.of2 * n2)) + (float)pl.d);
            float enum_ = RT_2M19 * (fabsf(R.of0 * n0) + fabsf(R.of1 * n1) + fabsf(R.of2 * n2) + fabsf((float)pl.d)) + RT_2M19 * R.K;
            // t = num / vn with vn < 0: t > tbest  <=>  num < tbest * vn  (both sides bounded conservatively)
            float rhs = st.tbest * (vn - evn) * 1.000002f;         // more negative than tbest*vn
            if (num + enum_ < rhs) continue;                       // certainly beyond the segment
        }
        double t;
        if (plane_exact(o, d, pl, t)) fast_consider(S, st, o, d, t, pl.id, early_dist, -1);
        if (st.stop) return;
    }
}

// ---- root interval (float32, conservative): the part of [0, tcap] inside the tree's root box; false: none -----------
__device__ __forceinline__ bool fast_root_interval(const FastScene &F, const FastRay &R, float tcap, float &tmin, float &tmax) {
    float a0 = (F.root_lo[0] - R.of0) * R.in0, b0 = (F.root_hi[0] - R.of0) * R.in0;
    float a1 = (F.root_lo[1] - R.of1) * R.in1, b1 = (F.root_hi[1] - R.of1) * R.in1;
    float a2 = (F.root_lo[2] - R.of2) * R.in2, b2 = (F.root_hi[2] - R.of2) * R.in2;
    float n0 = fminf(a0, b0), x0 = fmaxf(a0, b0);
    float n1 = fminf(a1, b1), x1 = fmaxf(a1, b1);
    float n2 = fminf(a2, b2), x2 = fmaxf(a2, b2);
    n0 -= fmaf(fabsf(n0), RT_2M20, R.E0); x0 += fmaf(fabsf(x0), RT_2M20, R.E0);
    n1 -= fmaf(fabsf(n1), RT_2M20, R.E1); x1 += fmaf(fabsf(x1), RT_2M20, R.E1);
    n2 -= fmaf(fabsf(n2), RT_2M20, R.E2); x2 += fmaf(fabsf(x2), RT_2M20, R.E2);
    tmin = fmaxf(fmaxf(n0, n1), fmaxf(n2, 0.0f));
    tmax = fminf(fminf(x0, x1), fminf(x2, tcap));
    return !(tmin > tmax);
}

// ---- the tree, for the ray parameters [tmin, tmax] (inside the root interval; any sub-range of it: every leaf whose
// conservative interval meets the range is visited, and every primitive of a visited leaf is tested whole) -----------
// early_dist: stop as soon as a confirmed hit is closer than this (-inf for ordinary rays).
// stk: this lane's LDS stack column (stride STRIDE entries).
// REV: the far side of a cut first, the walk runs from the ray's END towards its origin.  The answer does not depend on the order
// (every leaf that the segment can meet is visited, the float64 minimum is the same); for shadow rays, which start at a light
// sample and end in the shaded point, it means that the rays of a (hit, light) pair - neighbours in a wave - start their walks
// in the same leaf, at P, instead of in s*s leaves spread over the light, and an occluder near P ends them at once.
template <int STRIDE, bool REV = false>
__device__ __forceinline__ void fast_tree(const DScene &S, const FastScene &F, uint2 *stk, const FastRay &R, d3 o, d3 d, float tmin, float tmax,
                                          double early_dist, FastState &st, const uint4 *ltop = nullptr) {
    const float of0 = R.of0, of1 = R.of1, of2 = R.of2, ol0 = R.ol0, ol1 = R.ol1, ol2 = R.ol2;
    const float df0 = R.df0, df1 = R.df1, df2 = R.df2, in0 = R.in0, in1 = R.in1, in2 = R.in2;
    const float E0 = R.E0, E1 = R.E1, E2 = R.E2, Emax = R.Emax, af = R.af, dnf = R.dnf;
    tmax = fminf(tmax, st.tbest);
    if (tmin > tmax) return;
    uint32_t node = (ltop && F.n_top > 0) ? RT_TOP_FLAG : 0u;
    int sp = 0;
    for (;;) {
        KNode nd;
        RT_TS(const unsigned long long tc0 = clock64();)
        // ---- descend to a leaf, near child first (the first levels from the LDS table) ----
        for (;;) {
            uint32_t ax, left, right;
            float cut;
            if (node & RT_TOP_FLAG) {
                const uint4 t = ltop[node & ~RT_TOP_FLAG];
                cut = __uint_as_float(t.x); ax = t.y; left = t.z; right = t.w;
            } else {
                nd = F.knodes[node];
                if ((nd.b & 3u) == 3u) break;
                cut = __uint_as_float(nd.a); ax = nd.b & 3u; left = node + 1u; right = nd.b >> 2;
            }
            RT_TS(++st.n_node;)
            const float oa = ax == 0 ? of0 : (ax == 1 ? of1 : of2);
            const float ia = ax == 0 ? in0 : (ax == 1 ? in1 : in2);
            const float ea = ax == 0 ? E0 : (ax == 1 ? E1 : E2);
            const float t = (cut - oa) * ia;
            const float m = fmaf(fabsf(t), RT_2M20, ea);
            const float tl = t - m, th = t + m;
            const bool dneg = ia < 0.0f;      // side the ray comes from, consistent with the sign used for t
            const uint32_t nearc = dneg ? right : left, farc = dneg ? left : right;
            if (tl > tmax) node = nearc;                 // plane beyond the segment: near side only
            else if (th < tmin) node = farc;             // plane before the segment: far side only
            else if (!REV) {                             // both (NaN lands here too): far side waits on the stack
                stk[sp * STRIDE] = make_uint2(farc, __float_as_uint(tmax));
                ++sp;
                node = nearc;
                tmax = fminf(tmax, th);
            } else {                                     // ... or the near side, with the start of the segment it will see
                stk[sp * STRIDE] = make_uint2(nearc, __float_as_uint(tmin));
                ++sp;
                node = farc;
                tmin = fmaxf(tmin, tl);
            }
        }
        RT_TS(const unsigned long long tc1 = clock64(); st.c_desc += tc1 - tc0; unsigned long long tcc = 0;)
        // ---- leaf ----
        {
            // four entries are requested together and rejected in float32 together: a leaf (<= 10 entries unless it
            // ended through the EPSILON rule) costs one to three memory round trips instead of one per entry.  The
            // survivors are confirmed one after the other, in leaf order, by a single copy of the float64 code.
            const uint32_t first = nd.a, last = nd.a + (nd.b >> 2) - 1u;
            RT_TS(++st.n_leaf;)
            for (uint32_t e0 = first; e0 <= last; e0 += 4u) {
                RT_TS(++st.n_batch;)
                LeafF lf4[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) lf4[k] = F.leaves[min(e0 + (uint32_t)k, last)];
                const float tcur = st.tbest;
                uint32_t mask = 0u;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const LeafF &lf = lf4[k];
                    const uint32_t kind = lf.pk & 3u;
                    bool cand;
                    if (kind == 1u) {
                        const float px = (lf.x - of0) + (lf.lx - ol0);
                        const float py = (lf.y - of1) + (lf.ly - ol1);
                        const float pz = (lf.z - of2) + (lf.lz - ol2);
                        const float b = fmaf(px, df0, fmaf(py, df1, pz * df2));
                        const float pp = fmaf(px, px, fmaf(py, py, pz * pz));
                        const float r2 = lf.r * lf.r;
                        const float disc = fmaf(b, b, -af * (pp - r2));
                        const float ed = af * (RT_2M18 * (pp + r2));
                        cand = !(disc < -ed);
                        // both roots certainly behind the origin
                        if ((b < 0.0f) && (b * b > RT_2M38 * pp * af) && (pp - r2 > RT_2M19 * (pp + r2))) cand = false;
                        // sphere certainly beyond the part of the ray that can matter
                        const float far = fmaf(tcur, dnf, lf.r);
                        if (pp > far * far * 1.00001f) cand = false;
                    } else {
                        float a0 = (lf.x - of0) * in0, b0 = (lf.lx - of0) * in0;
                        float a1 = (lf.y - of1) * in1, b1 = (lf.ly - of1) * in1;
                        float a2 = (lf.z - of2) * in2, b2 = (lf.lz - of2) * in2;
                        float n0 = fminf(a0, b0), x0 = fmaxf(a0, b0);
                        float n1 = fminf(a1, b1), x1 = fmaxf(a1, b1);
                        float n2 = fminf(a2, b2), x2 = fmaxf(a2, b2);
                        n0 -= fmaf(fabsf(n0), RT_2M20, E0); x0 += fmaf(fabsf(x0), RT_2M20, E0);
                        n1 -= fmaf(fabsf(n1), RT_2M20, E1); x1 += fmaf(fabsf(x1), RT_2M20, E1);
                        n2 -= fmaf(fabsf(n2), RT_2M20, E2); x2 += fmaf(fabsf(x2), RT_2M20, E2);
                        const float tn = fmaxf(fmaxf(n0, n1), n2), tf = fminf(fminf(x0, x1), x2);
                        cand = !(tn > tf) && !(tf < 0.0f) && !(tn > tcur);
                    }
                    if (cand && e0 + (uint32_t)k <= last) mask |= 1u << k;
                }
                RT_TS(const unsigned long long tc2 = clock64();)
                while (mask) {
                    const int k = __builtin_ctz(mask);
                    mask &= mask - 1u;
                    RT_TS(++st.n_conf;)
                    const uint32_t pk = k == 0 ? lf4[0].pk : (k == 1 ? lf4[1].pk : (k == 2 ? lf4[2].pk : lf4[3].pk));
                    const uint32_t kind = pk & 3u;
                    const DPrim *pr = S.prims + (pk >> 2);
                    double t;
                    if (kind == 1u) {
                        if (sphere_exact(o, d, dot3(d, d), pr->a, pr->b[0], t)) fast_consider(S, st, o, d, t, pr->id, early_dist, (int32_t)(pk >> 2));
                    } else {
                        d3 inv = mk3(d.x != 0 ? 1.0 / d.x : dinf(), d.y != 0 ? 1.0 / d.y : dinf(), d.z != 0 ? 1.0 / d.z : dinf());
                        if (slab_exact(o, inv, pr->a, pr->b, true, t)) fast_consider(S, st, o, d, t, pr->id, early_dist, (int32_t)(pk >> 2));
                    }
                    if (st.stop) return;
                }
                RT_TS(tcc += clock64() - tc2;)
            }
        }
        RT_TS(const unsigned long long tc3 = clock64(); st.c_conf += tcc; st.c_test += (tc3 - tc1) - tcc;)
        // ---- next pending far node: it starts where the leaf just visited ended ----
        if (sp == 0) return;
        if (!REV) {
            float prev = tmax;
            tmin = fmaxf(prev - fmaf(fabsf(prev), RT_2M19, 2.0f * Emax), 0.0f);
            if (tmin > st.tbest) return;                     // everything still pending is farther than the best hit
            --sp;
            const uint2 top = stk[sp * STRIDE];
            node = top.x;
            tmax = fminf(__uint_as_float(top.y), st.tbest);
            while (tmin > tmax) {                            // entry emptied by a better hit found meanwhile
                if (sp == 0) return;
                --sp;
                const uint2 t2 = stk[sp * STRIDE];
                node = t2.x;
                tmax = fminf(__uint_as_float(t2.y), st.tbest);
            }
        } else {
            // ... next pending NEAR node: it ends where the leaf just visited began (and never beyond the best hit)
            float prev = tmin;
            tmax = fminf(prev + fmaf(fabsf(prev), RT_2M19, 2.0f * Emax), st.tbest);
            --sp;
            const uint2 top = stk[sp * STRIDE];
            node = top.x;
            tmin = __uint_as_float(top.y);
            while (tmin > tmax) {
                if (sp == 0) return;
                --sp;
                const uint2 t2 = stk[sp * STRIDE];
                node = t2.x;
                tmin = __uint_as_float(t2.y);
            }
        }
    }
}

// o, d: the float64 ray.  tlim: float32 upper bound of the ray parameters that can matter
// (+inf for ordinary rays).  early_dist: stop as soon as a confirmed hit is closer than this
// (-inf for ordinary rays).  stk: this lane's LDS stack column (stride STRIDE entries).
template <int STRIDE, bool REV = false>
__device__ __forceinline__ void fast_nearest(const DScene &S, const FastScene &F, uint2 *stk, d3 o, d3 d, float tlim,
                                             double early_dist, FastState &st, const uint4 *ltop = nullptr) {
    fast_init(st, tlim);
    const FastRay R = fast_ray(F, o, d);
    fast_planes(S, R, o, d, early_dist, st);
    if (st.stop) return;
    float tmin, tmax;
    if (!fast_root_interval(F, R, st.tbest, tmin, tmax)) return;
    fast_tree<STRIDE, REV>(S, F, stk, R, o, d, tmin, tmax, early_dist, st, ltop);
}
