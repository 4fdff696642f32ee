// rt_amd.hip - HIP kernels (gfx950) and the C-ABI of include/rt_amd.h.
//
// Wavefront pipeline, one recursion level of the reference's ray_tracing()
// (ray_tracer.py:36-133) per iteration, no host synchronisation inside a frame:
//
//   per level:
//     k_intersect  nearest hit of every queued ray, miss -> background, hit -> compacted hit list; the hits'
//                  reflection / transparency rays are appended to the next level's queue (parked per wave in LDS,
//                  64 at a time); level 0 also makes the camera rays, 8x8 pixel tile per wave, zeroes the frame's
//                  other framebuffers and writes the block boxes of the tile beams
//                                                        (ray_functions.py:7-43, BSPNode.py:123-210, ray_tracer.py:108-130)
//     soft shadows (Light.py:178-288), rt_shadow.hpp:
//       k_bin_zone, k_bin_tiles   (level 0 of large frames) candidates per (8x8 pixel tile, light) by binning
//       k_beam_setup   one lane per (hit, light) pair: one beam traversal for all its s*s rays - or the tile's candidate
//                      row filtered - and whole-pair verdicts; a pair that needs per-ray work takes a record slot
//       k_shadow_beam  one lane per ray of the pairs that need it: float32 filtered predicates on the pair's candidates
//       k_shadow_walk  rays of pairs with too many candidates: each walks the tree itself, same predicates
//       k_shadow_queue what the filters could not decide, in float64
//       (k_shadow: one float64 decision per ray, the diagnostic / fallback path)
//     k_shade      lit counts -> Phong terms, framebuffer accumulation               (Light.py:35-175, ray_tracer.py:77-106)
//   k_resolve    clip + float32 / uint8 quantisation (k_accumulate: the same over K frames)
//                                                                              (ray_tracer.py:30, util.py:98)
// enqueue_levels() lays the levels out over four streams (level 0's shadows | nearest hits of levels 1.. | their
// shadows and shading, odd and even levels apart); a frame enqueued behind a pending one starts with its camera pass on the
// second of them, beside the tail of that frame (AltL0: a second copy of the level-0 state).  DESIGN.md section 5 and 8.
//
// The colour recursion  colour = through*T + (diff+spec)*(1-T) + refl*Kr  is linear in the child
// colours, so each queued ray carries a throughput and adds throughput*local to its pixel.
#include <hip/hip_runtime.h>

#include <chrono>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>
#include <algorithm>
#include <atomic>
#include <thread>
#include <mutex>

#include "../../include/rt_amd.h"
#include "rt_device.hpp"
#include "rt_fast.hpp"
#include "rt_shadow.hpp"
#include "rt_small.hpp"

#define RT_BLOCK 256
#define RT_HIT_RUN 256u        // hits a wave parks before it appends them to the hit list (k_intersect, camera-ray pass)
#define RT_MAX_LEVELS 64
#define RT_INVALID_SLOT 0xFFFFFFFFu

// ------------------------------------------------------------------------------------------------
// device-side frame state
// ------------------------------------------------------------------------------------------------
struct FrameCtl {
    uint32_t n_rays[RT_MAX_LEVELS + 2];
    uint32_t n_hits[RT_MAX_LEVELS + 2];
    uint32_t overflow;
    uint32_t overflow_q;                    // ... and it was the float64 shadow-ray queue (bb.queue) that ran out
    uint32_t overflow_deep;                 // ... and it was the queue of a level >= 1 (sized cap_deep) that ran out
    uint32_t overflow_r;                    // ... and it was the pair records of a buffer set (BeamBuf::rec_cap) that ran out
    unsigned long long reflect_rays, transmit_rays, primary_rays;
    uint32_t n_listed[RT_MAX_LEVELS + 2];   // diagnostics: pairs that needed the per-ray filter / rays that needed
    uint32_t n_queued[RT_MAX_LEVELS + 2];   // the float64 queue, per level (copied by k_shade)
    uint32_t n_walked[RT_MAX_LEVELS + 2];   // rays of overflowing pairs (k_shadow_walk)
    // diagnostics build only (make stats: -DRT_DIAG; render flag 0x1000 counters, 0x2000 clocks of k_bin_tiles)
    uint32_t dbg_max_trips, dbg_pad;        // longest / total beam traversal of the frame
    uint32_t dbg_why[4];                    // pairs sent to the queue: delicate basis, touched list, units, dedup list
    unsigned long long dbg_tcyc[4];
    uint32_t dbg_tile[8];                   // flag 0x1000, tile beams: (zone, light) rows / entries, (region, light) rows / entries / overflows, (block, light) rows / entries / overflows
    unsigned long long dbg_sum_trips;
    unsigned long long dbg_bcyc[8];         // flag 0x4000, k_beam_setup, clocks of lane 0 of every wave: pair set-up, tile rows, tree walk, touched primitives, verdict + record, trips
};

struct RayQ {            // SoA ray queue in HBM
    double *ox, *oy, *oz, *dx, *dy, *dz;   // origin, direction (float64: they feed decisions)
    double *tr, *tg, *tb;                  // throughput
    uint32_t *slot;                        // output slot (RT_INVALID_SLOT = padding lane)
    uint32_t *gpix;                        // global pixel index: jitter hash key
    uint32_t *path;                        // path code: root 1, through 2k, reflect 2k+1
};

struct HitBuf {          // per ray slot of the CURRENT queue
    double *px, *py, *pz;
    int32_t *id;
    uint32_t *list;      // compacted indices of rays that hit something
    uint32_t *lit;       // [cap * n_shadow_lights] lit-sample counters
};

struct GenParams {
    double pos[3], X[3], Y[3], p00[3];
    double hg, wg;
    int32_t W, H, tiles_x, n_tiles, rank, world, tw, th;
    uint32_t n_slots;
    float rcp_tiles_x;       // 1 / tiles_x (div_small)
};

struct LevelParams {
    int32_t level, last_level;   // last_level: children are not traced, they return the background
    int32_t sh_rays, jitter;
    uint32_t seed, cap;                  // cap: capacity of THIS level's ray queue and per-hit / per-pair buffers
    uint32_t cap_next;                   // capacity of the next level's ray queue (the levels >= 1 live in smaller buffers)
    int32_t flags;
    // beam path: a chunk of RT_BLOCK lanes holds spb whole (hit, light) pairs of s2 rays each
    uint32_t spb, magic_s2, magic_s;     // magic_x = ceil(2^32 / x): n / x == __umulhi(n, magic_x) for n < 2^16
    int32_t beam;                        // this frame's soft shadows go through the beam path
    int32_t prio;                        // overlapped frame, deeper levels: raise the waves' issue priority
    int32_t unit_thr;                    // level 0 holds camera rays of the fused pass: throughput 1, not stored (gen_ray<false>)
    int32_t beam_early;                  // beam traversal near side first with the umbra test inside it (scenes with long beams)
    uint32_t walk_min;                   // rays of overflowing pairs: k_shadow_walk takes them when a level has at least this
                                         // many, else k_shadow_queue expands them (same verdicts either way)
    int32_t tiles;                       // tile beams: the k_bin_* kernels have listed this level's candidates per (block, light)
    uint32_t tile_stride;                // blocks per image row of blocks at level 0 of a camera frame (regions are squares), else 0
};

// ------------------------------------------------------------------------------------------------
// tile beams: block / region / zone geometry and the union beam of a group of hit points towards a light
// ------------------------------------------------------------------------------------------------
// half size of the box around a light's position that holds every sample origin of Light.get_shadow_rays for any shaded
// point: |coefficient of up|, |coefficient of right| <= (radius + g) / 2 with the deviations of Light.py:277-279, and
// |up_k| + |right_k| <= 2.  The binning kernels build the base box of a union beam from it, k_beam_setup checks its own
// sample box against it (same expression in both).
__device__ __forceinline__ float tile_light_half(float radius, float g) { return (radius + g) * 1.001f + 1e-5f; }
#define RT_2M17 7.6293945e-06f

struct TileGrid {                       // blocks of 64 ray slots -> regions of 4 x 4 blocks -> zones of 8 x 8 regions
    uint32_t nblk, ts, rpr, nreg, zpr, nzone;
    enum : uint32_t { rxs = 2u, rys = 2u };
    // stride: blocks per image row of blocks (0: no image structure, blocks in sequence)
    __host__ __device__ TileGrid(uint32_t n_rays, uint32_t stride) {
        nblk = (n_rays + 63u) >> 6;
        ts = stride;
        if (ts) {
            const uint32_t rrows = (((nblk + ts - 1u) / ts) + (1u << rys) - 1u) >> rys;
            rpr = (ts + (1u << rxs) - 1u) >> rxs; nreg = rpr * rrows;
            zpr = (rpr + 7u) >> 3; nzone = zpr * ((rrows + 7u) >> 3);
        } else { rpr = 0u; zpr = 0u; nreg = (nblk + RT_REGION_BLOCKS - 1u) / RT_REGION_BLOCKS; nzone = (nreg + 63u) >> 6; }
    }
    __device__ uint32_t region_of(uint32_t blk) const {
        if (!ts) return blk / RT_REGION_BLOCKS;
        const uint32_t by = blk / ts, bx = blk - by * ts;
        return (by >> rys) * rpr + (bx >> rxs);
    }
    // block j (0..15) of region r, or 0xffffffff
    __device__ uint32_t block_of(uint32_t r, uint32_t j) const {
        uint32_t blk;
        if (!ts) blk = r * RT_REGION_BLOCKS + j;
        else {
            const uint32_t ry = r / rpr, rx = r - ry * rpr, bx = (rx << rxs) + (j & ((1u << rxs) - 1u)), by = (ry << rys) + (j >> rxs);
            if (bx >= ts) return 0xffffffffu;
            blk = by * ts + bx;
        }
        return blk < nblk ? blk : 0xffffffffu;
    }
    __device__ uint32_t zone_of(uint32_t r) const {
        if (!ts) return r >> 6;
        const uint32_t ry = r / rpr, rx = r - ry * rpr;
        return (ry >> 3) * zpr + (rx >> 3);
    }
    // region j (0..63) of zone z, or 0xffffffff
    __device__ uint32_t region_in(uint32_t z, uint32_t j) const {
        uint32_t r;
        if (!ts) r = z * 64u + j;
        else {
            const uint32_t zy = z / zpr, zx = z - zy * zpr, rx = zx * 8u + (j & 7u), ry = zy * 8u + (j >> 3);
            if (rx >= rpr) return 0xffffffffu;
            r = ry * rpr + rx;
        }
        return r < nreg ? r : 0xffffffffu;
    }
};

// Union of the beams of every shaded point in the box [pl, ph] towards one light: the points (1 - lambda) p + lambda s, p in
// the box, s in the box [sl, sh] of the light's sample origins; per axis k they lie in [pl_k + lambda L_k, ph_k + lambda U_k]
// (L = sl - pl, U = sh - ph) - linear in lambda like a single beam, so the interval clipping of beam_collect applies with the
// apex taken on the side that is tested.  [l0, l1]: the part inside the tree's root box; [alo, ahi]: its bounding box.
struct UBeam {
    float pl0, pl1, pl2, ph0, ph1, ph2, L0, L1, L2, U0, U1, U2, iL0, iL1, iL2, iU0, iU1, iU2, E, l0, l1;
    float alo0, alo1, alo2, ahi0, ahi1, ahi2;
    bool ok;                            // false: no hit point, or the union misses the root box
    bool wild;                          // coordinates outside the float32 filters' range: no list (the pairs are delicate anyway)
};
struct ULight { float c0, c1, c2, radius; };       // a shadow light as the binning kernels need it (loaded once per row)
__device__ __forceinline__ ULight ulight_load(const DScene &S, uint32_t ml) {
    const DLight &Lt = S.lights[S.shadow_lights[ml]];
    ULight u;
    u.c0 = (float)Lt.pos[0]; u.c1 = (float)Lt.pos[1]; u.c2 = (float)Lt.pos[2]; u.radius = (float)Lt.radius;
    return u;
}
__device__ __forceinline__ UBeam ubeam_make(const ULight &Lt, const FastScene &F, float4 lo, float4 hi, int sh_rays) {
    UBeam B;
    B.ok = lo.x <= hi.x;
    const float radius = Lt.radius, g = radius / (float)sh_rays;
    const float Rs = tile_light_half(radius, g);
    const float c0 = Lt.c0, c1 = Lt.c1, c2 = Lt.c2;
    // + what the pairs' own float32 arithmetic can add (their sample boxes carry 2^-18 relative slack)
    const float k0 = fmaxf(fabsf(lo.x), fabsf(hi.x)), k1 = fmaxf(fabsf(lo.y), fabsf(hi.y)), k2 = fmaxf(fabsf(lo.z), fabsf(hi.z));
    const float r0 = Rs + RT_2M17 * (fmaxf(fabsf(c0 - lo.x), fabsf(c0 - hi.x)) + Rs + fabsf(c0) + k0) + 2e-5f;
    const float r1 = Rs + RT_2M17 * (fmaxf(fabsf(c1 - lo.y), fabsf(c1 - hi.y)) + Rs + fabsf(c1) + k1) + 2e-5f;
    const float r2 = Rs + RT_2M17 * (fmaxf(fabsf(c2 - lo.z), fabsf(c2 - hi.z)) + Rs + fabsf(c2) + k2) + 2e-5f;
    B.wild = !(fmaxf(fmaxf(k0, k1), k2) < 1e7f) || !(fmaxf(fmaxf(fabsf(c0), fabsf(c1)), fabsf(c2)) < 1e7f);
    B.pl0 = lo.x; B.pl1 = lo.y; B.pl2 = lo.z; B.ph0 = hi.x; B.ph1 = hi.y; B.ph2 = hi.z;
    const float Kp = fmaxf(fmaxf(k0, k1), fmaxf(k2, F.extent));
    B.E = Kp * RT_2M19 + 2e-6f;          // absolute slack on every plane position, as in beam_collect
    B.L0 = (c0 - r0) - lo.x; B.L1 = (c1 - r1) - lo.y; B.L2 = (c2 - r2) - lo.z;
    B.U0 = (c0 + r0) - hi.x; B.U1 = (c1 + r1) - hi.y; B.U2 = (c2 + r2) - hi.z;
    B.iL0 = __builtin_amdgcn_rcpf(B.L0); B.iL1 = __builtin_amdgcn_rcpf(B.L1); B.iL2 = __builtin_amdgcn_rcpf(B.L2);
    B.iU0 = __builtin_amdgcn_rcpf(B.U0); B.iU1 = __builtin_amdgcn_rcpf(B.U1); B.iU2 = __builtin_amdgcn_rcpf(B.U2);
    float l0 = 0.0f, l1 = 1.0f;
    const float E = B.E;
    clip_mul(l0, l1, -B.U0, -B.iU0, -((F.root_lo[0] - B.ph0) - E)); clip_mul(l0, l1, B.L0, B.iL0, (F.root_hi[0] - B.pl0) + E);
    clip_mul(l0, l1, -B.U1, -B.iU1, -((F.root_lo[1] - B.ph1) - E)); clip_mul(l0, l1, B.L1, B.iL1, (F.root_hi[1] - B.pl1) + E);
    clip_mul(l0, l1, -B.U2, -B.iU2, -((F.root_lo[2] - B.ph2) - E)); clip_mul(l0, l1, B.L2, B.iL2, (F.root_hi[2] - B.pl2) + E);
    B.ok = B.ok && l0 <= l1;
    B.l0 = fmaxf(l0 - 1e-6f, 0.0f); B.l1 = fminf(l1 + 1e-6f, 1.0f);
    B.alo0 = B.pl0 + fminf(B.l0 * B.L0, B.l1 * B.L0) - E; B.ahi0 = B.ph0 + fmaxf(B.l0 * B.U0, B.l1 * B.U0) + E;
    B.alo1 = B.pl1 + fminf(B.l0 * B.L1, B.l1 * B.L1) - E; B.ahi1 = B.ph1 + fmaxf(B.l0 * B.U1, B.l1 * B.U1) + E;
    B.alo2 = B.pl2 + fminf(B.l0 * B.L2, B.l1 * B.L2) - E; B.ahi2 = B.ph2 + fmaxf(B.l0 * B.U2, B.l1 * B.U2) + E;
    return B;
}
// bounding box of a leaf record (sphere {centre, r}; box {lo, -, hi})
__device__ __forceinline__ void leaf_box(const LeafF &lf, float &e0, float &e1, float &e2, float &f0, float &f1, float &f2) {
    const bool sph = (lf.pk & 3u) == 1u;
    const float rr = sph ? lf.r : 0.0f;
    e0 = lf.x - rr; e1 = lf.y - rr; e2 = lf.z - rr;
    f0 = sph ? lf.x + rr : lf.lx; f1 = sph ? lf.y + rr : lf.ly; f2 = sph ? lf.z + rr : lf.lz;
}
// the primitive's box against the bounding box of the union's part inside the root box (zones, regions)
__device__ __forceinline__ bool ubeam_near(const UBeam &B, const LeafF &lf) {
    float e0, e1, e2, f0, f1, f2;
    leaf_box(lf, e0, e1, e2, f0, f1, f2);
    return !(e0 > B.ahi0) && !(f0 < B.alo0) && !(e1 > B.ahi1) && !(f1 < B.alo1) && !(e2 > B.ahi2) && !(f2 < B.alo2);
}
// ... against the moving box itself: some lambda of [l0, l1] at which the two boxes meet (blocks)
__device__ __forceinline__ bool ubeam_touch(const UBeam &B, const LeafF &lf) {
    float e0, e1, e2, f0, f1, f2;
    leaf_box(lf, e0, e1, e2, f0, f1, f2);
    const float E = B.E;
    float x0 = B.l0, x1 = B.l1, y0 = B.l0, y1 = B.l1, z0 = B.l0, z1 = B.l1;       // three independent per-axis intervals
    clip_mul(x0, x1, B.L0, B.iL0, (f0 - B.pl0) + E); clip_mul(x0, x1, -B.U0, -B.iU0, -((e0 - B.ph0) - E));
    clip_mul(y0, y1, B.L1, B.iL1, (f1 - B.pl1) + E); clip_mul(y0, y1, -B.U1, -B.iU1, -((e1 - B.ph1) - E));
    clip_mul(z0, z1, B.L2, B.iL2, (f2 - B.pl2) + E); clip_mul(z0, z1, -B.U2, -B.iU2, -((e2 - B.ph2) - E));
    return fmaxf(fmaxf(x0, y0), z0) <= fminf(fminf(x1, y1), z1) && x0 <= x1 && y0 <= y1 && z0 <= z1;
}

// XCD-aware chunk walk.  A chunk is RT_BLOCK consecutive items; chunks are dealt to the 8 XCDs in groups of
// RT_XCD_GROUP (group g -> XCD g % 8; workgroups b and b+8 share an XCD and its L2): items that are neighbours in
// the queue (= in the image) keep hitting the same L2, and every XCD still sees every part of the image - a
// contiguous eighth per XCD left the XCDs that drew sky idle while the one with the horizon worked.
// Within its XCD, workgroup j of nb takes the XCD's chunks j, j + nb, j + 2 nb, ...  Speed only, never correctness.
#define RT_XCD_GROUP 8u
struct ChunkWalk {
    uint32_t m, step, xcd, nchunks, per;
    __device__ ChunkWalk(uint64_t n_items, uint32_t per_chunk = RT_BLOCK) {      // per_chunk: items a workgroup takes at a time
        per = per_chunk;
        nchunks = (uint32_t)((n_items + per - 1) / per);
        xcd = blockIdx.x & 7u;
        m = blockIdx.x >> 3;                 // index among the XCD's chunks
        step = gridDim.x >> 3;               // workgroups per XCD (grids are multiples of 8)
    }
    __device__ bool next(uint64_t &first_item) {
        const uint32_t c = ((m / RT_XCD_GROUP) * 8u + xcd) * RT_XCD_GROUP + (m % RT_XCD_GROUP);
        m += step;
        if (c >= nchunks) return false;      // c grows with m: nothing further either
        first_item = (uint64_t)c * per;
        return true;
    }
};

// "A queue overflowed earlier in this frame: the frame is void" - read ONCE per workgroup and handed to all its waves, for the
// kernels that hold workgroup barriers: another stream may raise the flag between the reads of two waves, and an exit that
// splits a workgroup in front of a __syncthreads() must not depend on how this hardware treats ended waves at s_barrier.
__device__ __forceinline__ bool frame_voided_uniform(const uint32_t *overflow_flag) {
    __shared__ uint32_t voided_wg;
    if (threadIdx.x == 0) voided_wg = *overflow_flag;
    __syncthreads();
    return voided_wg != 0u;
}

__device__ __forceinline__ void fb_add(double *fb, uint32_t slot, d3 c) {
    atomicAdd(fb + 3ull * slot + 0, c.x);
    atomicAdd(fb + 3ull * slot + 1, c.y);
    atomicAdd(fb + 3ull * slot + 2, c.z);
}

// ------------------------------------------------------------------------------------------------
// nearest hit (BSPNode.intersect_vectorize, BSPNode.py:123-210, for one ray)
// ------------------------------------------------------------------------------------------------
// Semantics: best = Background point (far clip) -> root planes in list order -> DFS, left child
// first, over every node whose box passes the slab test -> leaf primitives in list order; a
// candidate replaces the current best only when its Euclidean distance is STRICTLY smaller.
// Pre-order layout + skip links make the left-first DFS a linear walk without a stack.
__device__ __forceinline__ Hit nearest_hit_exact(const DScene &S, d3 o, d3 d) {
    Hit best;
    if (S.far_clip) {
        best.p = mk3(RT_MAX_RENDER_DISTANCE, RT_MAX_RENDER_DISTANCE, RT_MAX_RENDER_DISTANCE);
        best.dist = norm3(best.p - o);
        if (!(best.dist < dinf())) { best.p = mk3(dnan(), dnan(), dnan()); best.dist = dinf(); }
    } else {
        best.p = mk3(dnan(), dnan(), dnan());
        best.dist = dinf();
    }
    best.id = 0;
    for (int k = 0; k < S.n_planes; ++k) {      // wave-uniform loop: scalar loads
        double t;
        if (plane_exact(o, d, S.planes[k], t)) consider_t(best, o, d, t, S.planes[k].id);
    }
    d3 inv = mk3(d.x != 0 ? 1.0 / d.x : dinf(), d.y != 0 ? 1.0 / d.y : dinf(), d.z != 0 ? 1.0 / d.z : dinf());
    double a = dot3(d, d);
    int i = 0;
    const int n = S.n_nodes;
    while (i < n) {
        const DNode *nd = S.nodes + i;
        double tn;
        if (!slab_exact(o, inv, nd->lo, nd->hi, false, tn)) {
            i = nd->skip;
            continue;
        }
        int first = nd->first;
        if (first >= 0) {
            int cnt = nd->count;
            for (int e = 0; e < cnt; ++e) {
                const DPrim *pr = S.prims + S.leaf_prim[first + e];
                double t;
                if (pr->kind == 1) {
                    if (sphere_exact(o, d, a, pr->a, pr->b[0], t)) consider_t(best, o, d, t, pr->id);
                } else {
                    if (slab_exact(o, inv, pr->a, pr->b, true, t)) consider_t(best, o, d, t, pr->id);
                }
            }
        }
        i = i + 1;
    }
    return best;
}

__device__ __forceinline__ bool ray_is_tame(d3 o, d3 d) {
    // the float32 culling assumes finite coordinates well inside the float range and away from the
    // Background point (1e8,1e8,1e8); anything else takes the exact walk
    // ... and so does a ray with an exactly zero direction component: if its origin lies in a node's bound plane the
    // reference culls that node (0 * inf = NaN in Cube.cube_intersection_mask), which only the reference-order walk repeats
    double m = fmax(fmax(fabs(o.x), fabs(o.y)), fabs(o.z));
    double n = fmax(fmax(fabs(d.x), fabs(d.y)), fabs(d.z));
    return (m < 1e7) && (n < 1e7) && (n > 1e-30) && (d.x != 0) && (d.y != 0) && (d.z != 0);
}

#ifdef RT_TRAV_STATS
// diagnostics build only (make stats): per level [rays, node steps, leaf visits, leaf batches, confirmations, max node steps, max leaf visits, max confirmations]
__device__ unsigned long long g_trav_stats[RT_MAX_LEVELS * 8];
__device__ unsigned long long g_trav_time[RT_MAX_LEVELS * 8];     // per level: [7] = longest wave of k_intersect in wall_clock64 ticks (100 MHz)
// one entry per wave of k_intersect (levels 0 - 15, first 4096 waves): clocks of the whole wave, inside its nearest-hit calls, and
// of its busiest lane: descents, float32 leaf tests, confirmations, reference-order walk
__device__ unsigned long long g_wave_log[16 * 4096 * 6];
__device__ uint32_t g_wave_n[16];
#endif
// nearest hit, fast path with exact fallback (ties between different primitives, wild rays, non-kd trees)
template <bool FAST>
__device__ __forceinline__ Hit nearest_hit(const DScene &S, const FastScene &F, uint2 *stk, d3 o, d3 d, int stats_level = 0, const uint4 *ltop = nullptr,
                                           unsigned long long *wp = nullptr) {
    if (FAST && ray_is_tame(o, d)) {
        FastState st;
        fast_nearest<RT_BLOCK>(S, F, stk, o, d, __int_as_float(0x7f800000), -dinf(), st, ltop);
#ifdef RT_TRAV_STATS
        {
            unsigned long long *ts = g_trav_stats + 8 * stats_level;
            atomicAdd(ts + 0, 1ull); atomicAdd(ts + 1, (unsigned long long)st.n_node); atomicAdd(ts + 2, (unsigned long long)st.n_leaf);
            atomicAdd(ts + 3, (unsigned long long)st.n_batch); atomicAdd(ts + 4, (unsigned long long)st.n_conf);
            atomicMax(ts + 5, (unsigned long long)st.n_node); atomicMax(ts + 6, (unsigned long long)st.n_leaf); atomicMax(ts + 7, (unsigned long long)st.n_conf);
            unsigned long long *tt = g_trav_time + 8 * stats_level;      // [0..2]: clocks in descents / float32 leaf tests / confirmations, summed over rays
            atomicAdd(tt + 0, st.c_desc); atomicAdd(tt + 1, st.c_test); atomicAdd(tt + 2, st.c_conf);
            if (wp) { wp[0] += st.c_desc; wp[1] += st.c_test; wp[2] += st.c_conf; }
        }
#endif
        if (!st.tie) {
            if (st.best.id == 0 && S.far_clip)
                st.best.p = mk3(RT_MAX_RENDER_DISTANCE, RT_MAX_RENDER_DISTANCE, RT_MAX_RENDER_DISTANCE);
            return st.best;
        }
        // an exact tie between two primitives: the one the reference's walk visits first (FastScene::tie_off), without the walk
        if (F.tie_off && !st.tie_many && st.best_prim >= 0 && st.tie_prim >= 0 &&
            fmin(fmin(fabs(d.x), fabs(d.y)), fabs(d.z)) > 1e-200) {                 // (finite reciprocals: no 0 * inf in the slab tests)
            const d3 inv = mk3(1.0 / d.x, 1.0 / d.y, 1.0 / d.z);
            uint32_t fv[2] = {0xffffffffu, 0xffffffffu};
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                const int32_t prim = w == 0 ? st.best_prim : st.tie_prim;
                for (uint32_t k = F.tie_off[prim]; k < F.tie_off[prim + 1]; ++k) {
                    const uint2 en = F.tie_ent[k];
                    const DNode *nd = S.nodes + en.y;
                    double tn;
                    if (slab_exact(o, inv, nd->lo, nd->hi, false, tn)) { fv[w] = en.x; break; }
                }
            }
            if (fv[0] != 0xffffffffu || fv[1] != 0xffffffffu) {
                if (fv[0] <= fv[1]) return st.best;
                const DPrim *pr = S.prims + st.tie_prim;
                double t;
                const bool hit = pr->kind == 1 ? sphere_exact(o, d, dot3(d, d), pr->a, pr->b[0], t) : slab_exact(o, inv, pr->a, pr->b, true, t);
                if (hit) {
                    Hit h;
                    h.p = o + t * d;
                    h.dist = norm3(h.p - o);
                    h.id = pr->id;
                    if (h.dist == st.best.dist) return h;
                }
            }
        }
    }
#ifdef RT_TRAV_STATS
    {
        const unsigned long long e0 = clock64();
        const Hit h = nearest_hit_exact(S, o, d);
        unsigned long long *tt = g_trav_time + 8 * stats_level;      // [3]: clocks in the reference-order walk, [4]: rays that took it
        const unsigned long long e1 = clock64();
        atomicAdd(tt + 3, e1 - e0); atomicAdd(tt + 4, 1ull);
        if (FAST && ray_is_tame(o, d)) atomicAdd(tt + 5, 1ull);      // [5]: ... of them because of an exact tie (the others: untame rays)
        if (wp) wp[3] += e1 - e0;
        return h;
    }
#endif
    return nearest_hit_exact(S, o, d);
}

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
// camera ray of queue item k (ray_functions.py:31-41), written to the queue; false for a padding lane.
// n / d for n < 2^22 (tile numbers) and any d >= 1, rcp = 1 / d in float32: the float quotient (relative error < 2^-22) is off
// by at most one either way, one correction step makes it exact - ~10 instructions against ~40 (32-bit) or ~100 (64-bit) of
// the general integer division, four of which a camera ray used to need
__device__ __forceinline__ uint32_t div_small(uint32_t n, uint32_t d, float rcp, uint32_t &rem) {
    uint32_t q = (uint32_t)((float)n * rcp);
    int32_t r = (int32_t)n - (int32_t)(q * d);
    if (r < 0) { --q; r += (int32_t)d; }
    else if (r >= (int32_t)d) { ++q; r -= (int32_t)d; }
    rem = (uint32_t)r;
    return q;
}
// FULL = false (the fused camera-ray pass): origin and throughput stay unwritten - nobody traces the ray again, and
// k_shade knows that a camera ray starts with throughput 1 (LevelParams::unit_thr): 48 of 190 bytes per pixel less.
template <bool FULL>
__device__ __forceinline__ bool gen_ray(const GenParams &g, const RayQ &q, uint64_t k, d3 &o, d3 &dn, uint32_t &slot) {
    uint32_t tile_local, lane, lx, ly;
    const uint32_t k32 = (uint32_t)k;                 // (n_slots < 2^32)
    if (g.tw == 8 && g.th == 8) { tile_local = k32 >> 6; lane = k32 & 63u; lx = lane & 7u; ly = lane >> 3; }
    else {
        tile_local = k32 / (uint32_t)(g.tw * g.th);
        lane = k32 - tile_local * (uint32_t)(g.tw * g.th);
        ly = lane / (uint32_t)g.tw; lx = lane - ly * (uint32_t)g.tw;
    }
    uint32_t tile = g.world > 1 ? (uint32_t)g.rank + tile_local * (uint32_t)g.world : tile_local;
    uint32_t tx, ty;
    if (tile < (1u << 22)) ty = div_small(tile, (uint32_t)g.tiles_x, g.rcp_tiles_x, tx);
    else { ty = tile / (uint32_t)g.tiles_x; tx = tile - ty * (uint32_t)g.tiles_x; }
    int x = (int)tx * g.tw + (int)lx;
    int y = (int)ty * g.th + (int)ly;
    bool valid = tile < (uint32_t)g.n_tiles && x < g.W && y < g.H;
    if (!valid) {
        q.slot[k] = RT_INVALID_SLOT;
        slot = RT_INVALID_SLOT;
        return false;
    }
    d3 pos = ld3(g.pos), X = ld3(g.X), Y = ld3(g.Y), p00 = ld3(g.p00);
    d3 dest = (p00 - ((double)y * g.hg) * Y) + ((double)x * g.wg) * X;
    d3 dir = dest - pos;
    dn = dir / norm3(dir);
    o = pos;
    if (FULL) { q.ox[k] = pos.x; q.oy[k] = pos.y; q.oz[k] = pos.z; q.tr[k] = 1.0; q.tg[k] = 1.0; q.tb[k] = 1.0; }
    uint32_t gp = (uint32_t)y * (uint32_t)g.W + (uint32_t)x;
    slot = g.world > 1 ? (uint32_t)k : gp;
    q.dx[k] = dn.x; q.dy[k] = dn.y; q.dz[k] = dn.z;
    q.slot[k] = slot;
    q.gpix[k] = gp;
    q.path[k] = 1u;
    return true;
}

__global__ void __launch_bounds__(RT_BLOCK) k_gen(GenParams g, RayQ q, FrameCtl *ctl) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        ctl->n_rays[0] = g.n_slots;
    }
    ChunkWalk walk(g.n_slots);
    uint64_t first;
    unsigned long long valid_total = 0;
    while (walk.next(first)) {
        uint64_t k = first + threadIdx.x;
        if (k >= g.n_slots) continue;
        d3 o, dn;
        uint32_t slot;
        if (gen_ray<true>(g, q, k, o, dn, slot)) valid_total++;
    }
    // primary-ray count (padding lanes excluded)
    for (int off = 32; off > 0; off >>= 1) valid_total += __shfl_down(valid_total, off);
    if ((threadIdx.x & 63) == 0 && valid_total) atomicAdd(&ctl->primary_rays, valid_total);
}

// arbitrary ray batch (rt_trace / rt_intersect): AoS float64 -> queue
__global__ void __launch_bounds__(RT_BLOCK) k_load_rays(const double *o, const double *d, uint32_t n, RayQ q,
                                                        FrameCtl *ctl) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        ctl->n_rays[0] = n;
        ctl->primary_rays = n;
    }
    for (uint64_t k = (uint64_t)blockIdx.x * RT_BLOCK + threadIdx.x; k < n; k += (uint64_t)gridDim.x * RT_BLOCK) {
        q.ox[k] = o[3 * k]; q.oy[k] = o[3 * k + 1]; q.oz[k] = o[3 * k + 2];
        q.dx[k] = d[3 * k]; q.dy[k] = d[3 * k + 1]; q.dz[k] = d[3 * k + 2];
        q.tr[k] = 1.0; q.tg[k] = 1.0; q.tb[k] = 1.0;
        q.slot[k] = (uint32_t)k;
        q.gpix[k] = (uint32_t)k;
        q.path[k] = 1u;
    }
}

// Wave-private append: values are parked in the wave's LDS buffer (128 entries) and get their slots 64 at a time -
// ONE returning atomic on the counter per 64 values (returning atomics on a single address complete at ~30 ns
// each; one per wave and chunk was the bottleneck of the kernels that append).  All 64 lanes call these together.
template <typename T, typename Store>
__device__ __forceinline__ void wave_append(bool flag, T value, T *buf, uint32_t &n, uint32_t *counter, int lane, Store store) {
    const unsigned long long m = __ballot(flag);
    if (!m) return;
    if (flag) buf[n + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = value;
    n += (uint32_t)__popcll(m);
    if (n >= 64u) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(counter, 64u);
        base = __shfl(base, 0);
        const T v = buf[lane], v2 = buf[64 + lane];
        store(base + (uint32_t)lane, v);
        n -= 64u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        if ((uint32_t)lane < n) buf[lane] = v2;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
}
template <typename T, typename Store>
__device__ __forceinline__ void wave_append_flush(T *buf, uint32_t n, uint32_t *counter, int lane, Store store) {
    if (!n) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(counter, n);
    base = __shfl(base, 0);
    if ((uint32_t)lane < n) store(base + (uint32_t)lane, buf[lane]);
}

// Children of the level's hits (ray_tracer.py:108-130), made by the nearest-hit kernel itself: the next level's query is the
// frame's longest dependency chain (hits(L) -> children -> hits(L + 1) -> ...), and a launch of its own per level for the
// children - over EVERY hit, to find the few whose material spawns - sat in the middle of it.  A lane that finds a hit on a
// spawning surface parks (queue index of the parent << 1 | kind) in the wave's LDS buffer; 64 parked entries leave together:
// ONE returning atomic on the next queue's tail, then every lane makes one child from its parent's records (direction, hit
// point, id, throughput, keys - written by this wave earlier in the kernel, hence the fences) - dense work for all 64 lanes.
// kind 0: straight through (T != 0), kind 1: mirror (|Kr| != 0).
struct ChildOut { uint32_t n_refl, n_trans; };      // per lane and kernel: < 2^32
__device__ __forceinline__ void spawn_children(const DScene &S, const LevelParams &lp, const RayQ &q, const RayQ &qn, const HitBuf &hb, FrameCtl *ctl,
                                               const uint32_t *cbuf, uint32_t count, int lane, bool unit_thr, ChildOut &co) {
    // the parents' records were stored by other lanes of THIS wave: workgroup scope orders them with the loads below (one CU, one
    // vector L1, write-through) - a wider scope would write back and invalidate L2 lines of everything running on the XCD
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&ctl->n_rays[lp.level + 1], count);
    base = __shfl(base, 0);
    if (base + count > lp.cap_next) {
        if (lane == 0) ctl->overflow = ctl->overflow_deep = 1;     // frame is invalid; host re-renders with larger queues
        return;
    }
    if ((uint32_t)lane >= count) return;
    const uint32_t e = cbuf[lane], k = e >> 1;
    const bool mirror = (e & 1u) != 0u;
    const d3 d = mk3(q.dx[k], q.dy[k], q.dz[k]);
    const d3 P = mk3(hb.px[k], hb.py[k], hb.pz[k]);
    d3 thr = mk3(1.0, 1.0, 1.0);
    if (!unit_thr) thr = mk3(q.tr[k], q.tg[k], q.tb[k]);
    int32_t mi;
    const d3 N = surface_normal(S, hb.id[k], P, mi);                 // ray_tracer.py:77-78
    const DMat &M = S.mats[(mi >= 0 && mi < S.n_mats) ? mi : 0];     // (a parked entry has a valid material: its flags were read)
    const uint32_t w = base + (uint32_t)lane;
    d3 o_c, d_c, thr_c;
    if (!mirror) {                                                   // ray_tracer.py:108-116
        const double T = M.T;
        thr_c = thr * mk3(T, T, T);
        d_c = d;
        o_c = P + RT_EPSILON * d;
        co.n_trans++;
    } else {                                                         // ray_tracer.py:119-129
        thr_c = thr * ld3(M.kr);
        const d3 V = neg3(d);
        const d3 Nh = N / norm3(N);                                  // ray_functions.py:98-99
        const double dp = dot3(V, Nh);
        d_c = (2 * dp) * Nh - V;                                     // ray_functions.py:103
        o_c = P + RT_EPSILON * d_c;
        co.n_refl++;
    }
    qn.ox[w] = o_c.x; qn.oy[w] = o_c.y; qn.oz[w] = o_c.z;
    qn.dx[w] = d_c.x; qn.dy[w] = d_c.y; qn.dz[w] = d_c.z;
    qn.tr[w] = thr_c.x; qn.tg[w] = thr_c.y; qn.tb[w] = thr_c.z;
    qn.slot[w] = q.slot[k]; qn.gpix[w] = q.gpix[k]; qn.path[w] = child_path(q.path[k], mirror ? 1u : 0u);
}

// what the camera-ray pass writes for the tile beams of its level (k_block_box's work, folded in: a wave of the pass IS a
// block of 64 ray slots and holds its hit points in registers): the blocks' boxes, and the zones' list counters cleared
struct TileOut {
    float *blockbox;         // nullptr: no tile beams in this frame
    uint32_t *zcnt;
    uint32_t zrows;
};

// Nearest hit of every queued ray; miss -> background, hit -> hit record + compacted hit list + the hit's children appended to
// the next level's queue.
// GEN (level 0 of a frame): the camera rays are made here and written to the queue on the way - their stores
// overlap the traversal instead of costing a launch of their own (k_gen) before it.
template <bool FAST, bool GEN>
__global__ void __launch_bounds__(RT_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 4))) k_intersect(DScene S, FastScene F, LevelParams lp, RayQ q, RayQ qn, HitBuf hb,
                                                        FrameCtl *ctl, double *fb, int32_t *ids_out, GenParams gp, uint32_t *beam_counters,
                                                        double *fb_side, double *fb_side2, TileOut to) {
    extern __shared__ uint2 lds_stack[];
    uint2 *stk = lds_stack + threadIdx.x;
#ifdef RT_TRAV_STATS
    const long long tk0 = wall_clock64();
    const unsigned long long tc_start = clock64();
    unsigned long long wprof[4] = {0, 0, 0, 0}, tw_hit = 0;
#endif
    if (frame_voided_uniform(&ctl->overflow)) return;       // a queue overflowed earlier in this frame: the frame is void
    // the first levels of the tree, in LDS behind the stacks (rt_fast.hpp)
    uint4 *ltop = nullptr;
    if (FAST && F.n_top > 0) {
        ltop = (uint4 *)(lds_stack + (size_t)(F.depth > 2 ? F.depth : 2) * RT_BLOCK);
        rt_top_stage(F, ltop);
    }
    // the level's soft-shadow kernels follow on this stream: their append counters start at zero (saves a memset launch)
    if (beam_counters && blockIdx.x == 0 && threadIdx.x < 3) beam_counters[threadIdx.x] = 0u;
    if (GEN && to.blockbox && blockIdx.x == 0)
        for (uint32_t i = threadIdx.x; i < to.zrows; i += RT_BLOCK) to.zcnt[i] = 0u;
    if (lp.prio) __builtin_amdgcn_s_setprio(3);
    const uint32_t n = GEN ? gp.n_slots : min(ctl->n_rays[lp.level], lp.cap);
    if (GEN && blockIdx.x == 0 && threadIdx.x == 0) ctl->n_rays[0] = gp.n_slots;
    ChunkWalk walk(n);
    uint64_t first;
    const int lane = threadIdx.x & 63;
    unsigned long long valid_total = 0;
    // Compaction of the hit list.  The hits of a wave are parked in its LDS buffer and leave HRUN at a time: ONE returning
    // atomic on the list counter per run.  (One per wave and chunk - 32 400 of them in a 1920x1080 camera-ray pass, all on
    // one address, ~12 ns each - was what the pass took, whatever the scene: 0.40 ms.)  The deeper levels hold few rays and
    // run beside the level-0 kernels, where every KB of LDS counts: runs of 64 there.
    constexpr uint32_t HRUN = GEN ? RT_HIT_RUN : 64u;
    __shared__ uint32_t hbuf_all[(RT_BLOCK / 64) * (HRUN + 64)];      // per wave: up to HRUN - 1 parked hits + the 64 of one more chunk
    uint32_t *hbuf = hbuf_all + (threadIdx.x >> 6) * (HRUN + 64);
    uint32_t hn = 0u;
    __shared__ uint32_t cbuf_all[(RT_BLOCK / 64) * 192];              // per wave: parked children (<= 63 + 2 x 64)
    uint32_t *cbuf = cbuf_all + (threadIdx.x >> 6) * 192;
    uint32_t cn = 0u;
    ChildOut co;
    co.n_refl = 0; co.n_trans = 0;
    const bool unit_thr = GEN || (lp.unit_thr && lp.level == 0);
    for (;;) {
      const bool more = walk.next(first);
      if (more) {
        uint64_t k = first + threadIdx.x;
        bool active = k < n;
        uint32_t slot = RT_INVALID_SLOT;
        d3 o = mk3(0, 0, 0), d = mk3(0, 0, 1);
        if (GEN) {
            // every output slot is visited exactly once here: the framebuffers are initialised by plain stores
            // (fb: 0 or the background of a miss, below; fb_side: 0) instead of two memsets before the frame
            const bool in_range = active;
            if (active) active = gen_ray<false>(gp, q, k, o, d, slot);
            valid_total += active ? 1ull : 0ull;
            const uint32_t zs = active ? slot : ((in_range && gp.world > 1) ? (uint32_t)k : RT_INVALID_SLOT);
            if (zs != RT_INVALID_SLOT) {
                if (!active) { fb[3ull * zs] = 0.0; fb[3ull * zs + 1] = 0.0; fb[3ull * zs + 2] = 0.0; }
                if (fb_side) { fb_side[3ull * zs] = 0.0; fb_side[3ull * zs + 1] = 0.0; fb_side[3ull * zs + 2] = 0.0; }
                if (fb_side2) { fb_side2[3ull * zs] = 0.0; fb_side2[3ull * zs + 1] = 0.0; fb_side2[3ull * zs + 2] = 0.0; }
            }
        } else {
            slot = active ? q.slot[k] : RT_INVALID_SLOT;
            active = active && slot != RT_INVALID_SLOT;
            if (active) {
                o = mk3(q.ox[k], q.oy[k], q.oz[k]);
                d = mk3(q.dx[k], q.dy[k], q.dz[k]);
            }
        }
        // the nearest hit; hid: its id, hp: its point (also stored to the hit record as soon as it is known)
        int32_t hid = 0;
        d3 hp = mk3(dnan(), dnan(), dnan());
#ifdef RT_TRAV_STATS
        const unsigned long long th0 = clock64();
#endif
        if (active) {
#ifdef RT_TRAV_STATS
            const Hit h = nearest_hit<FAST>(S, F, stk, o, d, lp.level, ltop, wprof);
#else
            const Hit h = nearest_hit<FAST>(S, F, stk, o, d, lp.level, ltop);
#endif
            hid = h.id;
            hp = h.p;
            if (hid != 0) { hb.px[k] = hp.x; hb.py[k] = hp.y; hb.pz[k] = hp.z; }
        }
#ifdef RT_TRAV_STATS
        tw_hit += clock64() - th0;
#endif
        bool has_hit = false;
        uint32_t mflags = 0u;
        if (active) {
            if (ids_out && lp.level == 0) ids_out[slot] = hid;
            if (hid == 0) {
                // ray_tracer.py:65,72,132: misses keep the background colour
                if (GEN) { fb[3ull * slot] = S.bg[0]; fb[3ull * slot + 1] = S.bg[1]; fb[3ull * slot + 2] = S.bg[2]; }   // = 0 + 1 * bg
                else fb_add(fb, slot, mk3(q.tr[k], q.tg[k], q.tb[k]) * ld3(S.bg));
            } else {
                // (with side buffers the frame is an overlapped one: k_shade<.. | RT_SHADE_STORE> stores this slot's value later)
                if (GEN && !fb_side) { fb[3ull * slot] = 0.0; fb[3ull * slot + 1] = 0.0; fb[3ull * slot + 2] = 0.0; }
                hb.id[k] = hid;
                has_hit = true;
                // does the material spawn?  (SurfaceAbs.py:61,83 look the surface up by LIST POSITION = hit id, Q8: an id past the
                // list reads as material 0, like surface_normal)
                if (!lp.last_level) mflags = (uint32_t)((hid > 0 && hid < S.n_surfs) ? S.surfs[hid].mflags : (S.n_mats > 0 ? S.mats[0].flags : 0));
            }
        }
        if (!GEN && !has_hit && k < n) hb.id[k] = 0;      // (k_block_box reads the hit record of every slot)
        if (GEN && to.blockbox) {
            // box of the block's hit points, rounded outward (a wave of the camera pass = one block of 64 ray slots)
            const float inf = __int_as_float(0x7f800000);
            float b0 = inf, b1 = inf, b2 = inf, B0 = -inf, B1 = -inf, B2 = -inf;
            if (has_hit) {
                b0 = __double2float_rd(hp.x); b1 = __double2float_rd(hp.y); b2 = __double2float_rd(hp.z);
                B0 = __double2float_ru(hp.x); B1 = __double2float_ru(hp.y); B2 = __double2float_ru(hp.z);
            }
            for (int off = 32; off > 0; off >>= 1) {
                b0 = fminf(b0, __shfl_xor(b0, off)); b1 = fminf(b1, __shfl_xor(b1, off)); b2 = fminf(b2, __shfl_xor(b2, off));
                B0 = fmaxf(B0, __shfl_xor(B0, off)); B1 = fmaxf(B1, __shfl_xor(B1, off)); B2 = fmaxf(B2, __shfl_xor(B2, off));
            }
            if (lane == 0 && k < n) {
                float4 *bx = (float4 *)(to.blockbox + 8ull * (k >> 6));
                bx[0] = make_float4(b0, b1, b2, 0.0f);
                bx[1] = make_float4(B0, B1, B2, 0.0f);
            }
        }
        // ---- the hit list ----
        const unsigned long long mask = __ballot(has_hit);
        if (mask) {
            if (has_hit) hbuf[hn + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = (uint32_t)k;
            hn += (uint32_t)__popcll(mask);
            if (hn >= HRUN) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&ctl->n_hits[lp.level], HRUN);
                base = __shfl(base, 0);
#pragma unroll
                for (uint32_t j = 0; j < HRUN / 64u; ++j) hb.list[base + j * 64u + (uint32_t)lane] = hbuf[j * 64u + (uint32_t)lane];
                const uint32_t rest = hbuf[HRUN + (uint32_t)lane];
                hn -= HRUN;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                if ((uint32_t)lane < hn) hbuf[lane] = rest;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
            }
        }
        // ---- the children: parked ----
        {
            const uint32_t nc = (mflags & 1u) + ((mflags >> 1) & 1u);
            const unsigned long long m1 = __ballot(nc >= 1u), m2 = __ballot(nc == 2u);
            if (m1) {
                const unsigned long long below = (1ull << lane) - 1ull;
                const uint32_t pos = cn + (uint32_t)__popcll(m1 & below) + (uint32_t)__popcll(m2 & below);
                if (nc >= 1u) cbuf[pos] = ((uint32_t)k << 1) | ((mflags & 1u) ? 0u : 1u);     // through first, like ray_tracer.py:108-129
                if (nc == 2u) cbuf[pos + 1u] = ((uint32_t)k << 1) | 1u;
                cn += (uint32_t)__popcll(m1) + (uint32_t)__popcll(m2);
            }
        }
      }
      // ---- 64 parked children leave together (at the end: the rest) ----
      while (cn >= 64u || (!more && cn > 0u)) {
          const uint32_t count = cn < 64u ? cn : 64u;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
          spawn_children(S, lp, q, qn, hb, ctl, cbuf, count, lane, unit_thr, co);
          const uint32_t v1 = cbuf[64 + lane], v2 = cbuf[128 + lane];
          cn -= count;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          __builtin_amdgcn_wave_barrier();
          cbuf[lane] = v1; cbuf[64 + lane] = v2;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          __builtin_amdgcn_wave_barrier();
      }
      if (!more) break;
    }
    if (hn) {                                     // the rest of the wave's hits
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&ctl->n_hits[lp.level], hn);
        base = __shfl(base, 0);
        for (uint32_t j = (uint32_t)lane; j < hn; j += 64u) hb.list[base + j] = hbuf[j];
    }
    {
        unsigned long long n_refl = co.n_refl, n_trans = co.n_trans;
        for (int off = 32; off > 0; off >>= 1) {
            n_refl += __shfl_down(n_refl, off);
            n_trans += __shfl_down(n_trans, off);
        }
        if (lane == 0) {
            if (n_refl) atomicAdd(&ctl->reflect_rays, n_refl);
            if (n_trans) atomicAdd(&ctl->transmit_rays, n_trans);
        }
    }
    if (GEN) {                                    // primary-ray count (padding lanes excluded)
        for (int off = 32; off > 0; off >>= 1) valid_total += __shfl_down(valid_total, off);
        if (lane == 0 && valid_total) atomicAdd(&ctl->primary_rays, valid_total);
    }
#ifdef RT_TRAV_STATS
    if (lane == 0) atomicMax(g_trav_time + 8 * lp.level + 7, (unsigned long long)(wall_clock64() - tk0));
    if (lp.level < 16) {
        // the wave's busiest lane
        unsigned long long mine = wprof[0] + wprof[1] + wprof[2] + wprof[3], best = mine;
        for (int off = 32; off > 0; off >>= 1) { const unsigned long long o2 = __shfl_xor(best, off); best = o2 > best ? o2 : best; }
        const unsigned long long who = __ballot(mine == best);
        if (lane == __ffsll((long long)who) - 1 && tw_hit > 0) {
            const uint32_t at = atomicAdd(&g_wave_n[lp.level], 1u);
            if (at < 4096u) {
                unsigned long long *e = g_wave_log + ((size_t)lp.level * 4096 + at) * 6;
                e[0] = clock64() - tc_start; e[1] = tw_hit; e[2] = wprof[0]; e[3] = wprof[1]; e[4] = wprof[2]; e[5] = wprof[3];
            }
        }
    }
#endif
}

// exact light-plane basis of one (hit, light) pair (Light.py:48-50,252-260)
__device__ __forceinline__ void exact_basis(const DLight &Lt, int sh_rays, d3 P, d3 &up, d3 &right, d3 &p00) {
    d3 lpos = ld3(Lt.pos);
    d3 dl = lpos - P;                              // Light.py:48-50
    d3 L = dl / norm3(dl);
    light_basis(neg3(L), up, right);               // Light.py:257
    double radius = Lt.radius;
    double gsz = radius / (double)sh_rays;         // Light.py:254
    double c0 = (radius - gsz) / 2;
    p00 = (lpos + c0 * up) - c0 * right;           // Light.py:260
}

// ------------------------------------------------------------------------------------------------
// tile beams: candidates per (block, light) by binning - zone -> region -> block, no tree walk, no long dependent chains
// ------------------------------------------------------------------------------------------------
// The pairs of a block of 64 ray slots (an 8x8 pixel tile at level 0) send nearly the same beam towards a light.  Every
// primitive is tested against the union beam of a ZONE (64 regions) and listed there; a REGION (4 x 4 blocks) keeps what of
// its zone's list its own union beam can touch, a BLOCK what of its region's list it can touch (exact test against the moving
// box) - at most RT_TILE_NC leaf records per (block, light), which k_beam_setup<.., true> filters per pair instead of walking
// the tree.  All passes are flat data-parallel loops.

// box of the hit points of every block (one wave per block), rounded outward; the zones' list counters are cleared on the way
__global__ void __launch_bounds__(RT_BLOCK) k_block_box(DScene S, LevelParams lp, HitBuf hb, BeamBuf bb, FrameCtl *ctl) {
    if (ctl->overflow) return;
    const uint32_t n = min(ctl->n_rays[lp.level], lp.cap);
    const TileGrid G(n, lp.tile_stride);
    const uint32_t zrows = G.nzone * (uint32_t)S.n_shadow_lights;
    for (uint32_t i = blockIdx.x * RT_BLOCK + threadIdx.x; i < zrows; i += gridDim.x * RT_BLOCK) bb.zcnt[i] = 0u;
    const int lane = threadIdx.x & 63;
    for (uint64_t k0 = (uint64_t)blockIdx.x * RT_BLOCK; k0 < n; k0 += (uint64_t)gridDim.x * RT_BLOCK) {
        const uint64_t k = k0 + threadIdx.x;
        const uint64_t kk = k < n ? k : (uint64_t)(n - 1u);       // (coordinates of a miss: whatever the buffers hold, not used)
        const double x = hb.px[kk], y = hb.py[kk], z = hb.pz[kk];
        const bool has_hit = k < n && hb.id[kk] != 0;
        const float inf = __int_as_float(0x7f800000);
        float b0 = inf, b1 = inf, b2 = inf, B0 = -inf, B1 = -inf, B2 = -inf;
        if (has_hit) {
            b0 = __double2float_rd(x); b1 = __double2float_rd(y); b2 = __double2float_rd(z);
            B0 = __double2float_ru(x); B1 = __double2float_ru(y); B2 = __double2float_ru(z);
        }
        for (int off = 32; off > 0; off >>= 1) {
            b0 = fminf(b0, __shfl_xor(b0, off)); b1 = fminf(b1, __shfl_xor(b1, off)); b2 = fminf(b2, __shfl_xor(b2, off));
            B0 = fmaxf(B0, __shfl_xor(B0, off)); B1 = fmaxf(B1, __shfl_xor(B1, off)); B2 = fmaxf(B2, __shfl_xor(B2, off));
        }
        if (lane == 0 && k < n) {
            float4 *bx = (float4 *)(bb.blockbox + 8ull * (k >> 6));
            bx[0] = make_float4(b0, b1, b2, 0.0f);
            bx[1] = make_float4(B0, B1, B2, 0.0f);
        }
    }
}

// union of the boxes of blocks [first, first + count) given by index function f (lane-strided), reduced over the wave
template <typename Fn>
__device__ __forceinline__ void wave_union_box(const float *blockbox, uint32_t count, Fn block_at, int lane, float4 &lo, float4 &hi,
                                               float4 *mylo = nullptr, float4 *myhi = nullptr) {
    const float inf = __int_as_float(0x7f800000);
    float b0 = inf, b1 = inf, b2 = inf, B0 = -inf, B1 = -inf, B2 = -inf;
    for (uint32_t j = (uint32_t)lane; j < count; j += 64u) {
        const uint32_t blk = block_at(j);
        if (blk == 0xffffffffu) continue;
        const float4 l = ((const float4 *)(blockbox + 8ull * blk))[0], h = ((const float4 *)(blockbox + 8ull * blk))[1];
        if (mylo) { *mylo = l; *myhi = h; }      // (count <= 64: the lane's one box)
        if (!(l.x <= h.x)) continue;
        b0 = fminf(b0, l.x); b1 = fminf(b1, l.y); b2 = fminf(b2, l.z);
        B0 = fmaxf(B0, h.x); B1 = fmaxf(B1, h.y); B2 = fmaxf(B2, h.z);
    }
    for (int off = 32; off > 0; off >>= 1) {
        b0 = fminf(b0, __shfl_xor(b0, off)); b1 = fminf(b1, __shfl_xor(b1, off)); b2 = fminf(b2, __shfl_xor(b2, off));
        B0 = fmaxf(B0, __shfl_xor(B0, off)); B1 = fmaxf(B1, __shfl_xor(B1, off)); B2 = fmaxf(B2, __shfl_xor(B2, off));
    }
    lo = make_float4(b0, b1, b2, 0.0f); hi = make_float4(B0, B1, B2, 0.0f);
}

// zones: every primitive against the union beam of each (zone, light); work item = (row, group of 8 chunks of RT_BLOCK primitives)
__global__ void __launch_bounds__(RT_BLOCK) k_bin_zone(DScene S, FastScene F, LevelParams lp, BeamBuf bb, FrameCtl *ctl) {
    if (frame_voided_uniform(&ctl->overflow)) return;
    const uint32_t n = min(ctl->n_rays[lp.level], lp.cap);
    const TileGrid G(n, lp.tile_stride);
    const uint32_t nsl = (uint32_t)S.n_shadow_lights, np = (uint32_t)S.n_prims, zcap = min(np, RT_ZONE_NC);
    const uint32_t ngroup = (np + 8u * RT_BLOCK - 1u) / (8u * RT_BLOCK), items = G.nzone * nsl * ngroup;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ float4 zbox[2 * (RT_BLOCK / 64)];
    for (uint32_t it = blockIdx.x; it < items; it += gridDim.x) {
        const uint32_t row = it / ngroup, group = it - row * ngroup;
        const uint32_t zone = row / nsl, ml = row - zone * nsl;
        // the zone's box: 1024 block boxes, 256 per wave (4 per lane), combined through LDS
        float4 lo, hi;
        wave_union_box(bb.blockbox, 256u, [&](uint32_t jj) {
            const uint32_t j = (uint32_t)wave * 256u + jj, r = G.region_in(zone, j >> 4);
            return r != 0xffffffffu ? G.block_of(r, j & 15u) : 0xffffffffu; }, lane, lo, hi);
        __syncthreads();
        if (lane == 0) { zbox[2 * wave] = lo; zbox[2 * wave + 1] = hi; }
        __syncthreads();
        for (int w = 0; w < RT_BLOCK / 64; ++w) {
            const float4 l = zbox[2 * w], h = zbox[2 * w + 1];
            lo.x = fminf(lo.x, l.x); lo.y = fminf(lo.y, l.y); lo.z = fminf(lo.z, l.z);
            hi.x = fmaxf(hi.x, h.x); hi.y = fmaxf(hi.y, h.y); hi.z = fmaxf(hi.z, h.z);
        }
        const UBeam B = ubeam_make(ulight_load(S, ml), F, lo, hi, lp.sh_rays);
        if (!B.ok) continue;
        // (all loads of the group's 8 chunks are in flight together; one returning atomic per wave)
        uint32_t ce[8];
        LeafF lf[8];
#pragma unroll
        for (uint32_t c = 0; c < 8u; ++c) {
            const uint32_t p = (group * 8u + c) * RT_BLOCK + threadIdx.x;
            ce[c] = p < np ? F.prim_entry[p] : 0u;
        }
#pragma unroll
        for (uint32_t c = 0; c < 8u; ++c) lf[c] = F.leaves[ce[c]];
        uint32_t keepm = 0u, total = 0u, pos_of[8];
#pragma unroll
        for (uint32_t c = 0; c < 8u; ++c) {
            const uint32_t p = (group * 8u + c) * RT_BLOCK + threadIdx.x;
            const bool keep = p < np && (B.wild || ubeam_near(B, lf[c]));
            const unsigned long long m = __ballot(keep);
            if (keep) keepm |= 1u << c;
            pos_of[c] = total + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            total += (uint32_t)__popcll(m);
        }
        if (total) {
            uint32_t base = 0u;
            if (lane == 0) base = atomicAdd(&bb.zcnt[row], total);
            base = __shfl(base, 0);
#pragma unroll
            for (uint32_t c = 0; c < 8u; ++c)
                if (((keepm >> c) & 1u) && base + pos_of[c] < zcap) bb.zlist[(uint64_t)row * zcap + base + pos_of[c]] = ce[c];
        }
    }
}

// regions and blocks: one workgroup per (region, light) keeps what of its zone's list the region's union beam comes near - as
// leaf records, in LDS -, then each wave filters that for 4 of the region's 16 blocks (exact test against the block's moving
// box) and writes the blocks' candidate rows
__global__ void __launch_bounds__(RT_BLOCK) k_bin_tiles(DScene S, FastScene F, LevelParams lp, BeamBuf bb, FrameCtl *ctl) {
    __shared__ uint32_t cnt;
    __shared__ float4 recs[2 * RT_REGION_NC];
    if (frame_voided_uniform(&ctl->overflow)) return;
    const uint32_t n = min(ctl->n_rays[lp.level], lp.cap);
    const TileGrid G(n, lp.tile_stride);
    const uint32_t nsl = (uint32_t)S.n_shadow_lights, zcap = min((uint32_t)S.n_prims, RT_ZONE_NC), rows = G.nreg * nsl;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t row = blockIdx.x; row < rows; row += gridDim.x) {
#ifdef RT_DIAG
        const long long tc0 = wall_clock64();
#endif
        const uint32_t reg = row / nsl, ml = row - reg * nsl;
        // the zone's list: its length and the lane's first two entries are requested before anything else of the row
        const uint32_t zrow = G.zone_of(reg) * nsl + ml;
        const uint32_t *zl = bb.zlist + (uint64_t)zrow * zcap;
        const uint32_t zc = bb.zcnt[zrow], zn = min(zc, zcap);
        uint32_t ce_a = threadIdx.x < zcap ? zl[threadIdx.x] : 0u, ce_b = threadIdx.x + RT_BLOCK < zcap ? zl[threadIdx.x + RT_BLOCK] : 0u;
        float4 lo, hi;
        const float inf = __int_as_float(0x7f800000);
        float4 mylo = make_float4(inf, inf, inf, 0.0f), myhi = make_float4(-inf, -inf, -inf, 0.0f);
        wave_union_box(bb.blockbox, RT_REGION_BLOCKS, [&](uint32_t j) { return G.block_of(reg, j); }, lane, lo, hi, &mylo, &myhi);
        const ULight UL = ulight_load(S, ml);
        const UBeam B = ubeam_make(UL, F, lo, hi, lp.sh_rays);
        __syncthreads();                 // (the last readers of the counter and of the records are through)
        if (threadIdx.x == 0) cnt = 0u;
        __syncthreads();
#ifdef RT_DIAG
        const long long tc1 = wall_clock64();
#endif
        bool over = B.ok && (zc > zcap || B.wild);
        if (B.ok && !over) {
            for (uint32_t i0 = 0; i0 < zn; i0 += 2u * RT_BLOCK) {
                const uint32_t ia = i0 + threadIdx.x, ib = ia + RT_BLOCK;
                if (i0) { ce_a = ia < zn ? zl[ia] : 0u; ce_b = ib < zn ? zl[ib] : 0u; }
                const LeafF la = F.leaves[ia < zn ? ce_a : 0u], lb = F.leaves[ib < zn ? ce_b : 0u];
                const bool ka = ia < zn && ubeam_near(B, la), kb = ib < zn && ubeam_near(B, lb);
                const unsigned long long ma = __ballot(ka), mb = __ballot(kb);
                if (ma | mb) {
                    uint32_t base = 0u;
                    if (lane == 0) base = atomicAdd(&cnt, (uint32_t)(__popcll(ma) + __popcll(mb)));
                    base = __shfl(base, 0);
                    const uint32_t pa = base + (uint32_t)__popcll(ma & ((1ull << lane) - 1ull));
                    const uint32_t pb = base + (uint32_t)__popcll(ma) + (uint32_t)__popcll(mb & ((1ull << lane) - 1ull));
                    if (ka && pa < RT_REGION_NC) {
                        recs[2u * pa] = make_float4(la.x, la.y, la.z, la.r);
                        recs[2u * pa + 1u] = make_float4(la.lx, la.ly, la.lz, __uint_as_float(la.pk));
                    }
                    if (kb && pb < RT_REGION_NC) {
                        recs[2u * pb] = make_float4(lb.x, lb.y, lb.z, lb.r);
                        recs[2u * pb + 1u] = make_float4(lb.lx, lb.ly, lb.lz, __uint_as_float(lb.pk));
                    }
                }
            }
        }
        __syncthreads();
        const uint32_t rn = cnt;
        over = over || rn > RT_REGION_NC;
#ifdef RT_DIAG
        const long long tc2 = wall_clock64();
        if ((lp.flags & 0x1000) && threadIdx.x == 0 && B.ok) {
            atomicAdd(&ctl->dbg_tile[1], zn);
            atomicAdd(&ctl->dbg_tile[2], 1u); atomicAdd(&ctl->dbg_tile[3], min(rn, RT_REGION_NC)); if (over) atomicAdd(&ctl->dbg_tile[4], 1u);
        }
#endif
        {
            // a wave takes 4 of the region's blocks at once: 16 lanes per block, 16 records per step
            const uint32_t sub = (uint32_t)lane >> 4, li = (uint32_t)lane & 15u, j = (uint32_t)wave * 4u + sub;
            const uint32_t blk = G.block_of(reg, j);
            const uint64_t trow = (uint64_t)(blk != 0xffffffffu ? blk : 0u) * nsl + ml;
            float4 blo, bhi;             // (lane j of wave_union_box above held block j's box)
            blo.x = __shfl(mylo.x, (int)j); blo.y = __shfl(mylo.y, (int)j); blo.z = __shfl(mylo.z, (int)j); blo.w = 0.0f;
            bhi.x = __shfl(myhi.x, (int)j); bhi.y = __shfl(myhi.y, (int)j); bhi.z = __shfl(myhi.z, (int)j); bhi.w = 0.0f;
            const UBeam Bb = ubeam_make(UL, F, blo, bhi, lp.sh_rays);
            const bool go = blk != 0xffffffffu && Bb.ok && !over;
            uint32_t c = 0u;
            float4 *out = bb.trec + trow * (2u * RT_TILE_NC);
            for (uint32_t i0 = 0; i0 < rn; i0 += 16u) {
                if (!__any(go && c != RT_TILE_OVERFLOW)) break;
                const uint32_t i = i0 + li;
                bool keep = false;
                float4 ra = make_float4(0.0f, 0.0f, 0.0f, 0.0f), rb = ra;
                if (go && c != RT_TILE_OVERFLOW && i < rn) {
                    ra = recs[2u * i]; rb = recs[2u * i + 1u];
                    LeafF lf;
                    lf.x = ra.x; lf.y = ra.y; lf.z = ra.z; lf.r = ra.w; lf.lx = rb.x; lf.ly = rb.y; lf.lz = rb.z; lf.pk = __float_as_uint(rb.w);
                    keep = ubeam_near(Bb, lf) && ubeam_touch(Bb, lf);
                }
                const uint32_t m16 = (uint32_t)(__ballot(keep) >> (sub * 16u)) & 0xffffu;
                const uint32_t pos = c + (uint32_t)__popc(m16 & ((1u << li) - 1u)), tot = (uint32_t)__popc(m16);
                if (c != RT_TILE_OVERFLOW) {
                    if (c + tot > RT_TILE_NC) c = RT_TILE_OVERFLOW;
                    else {
                        if (keep) { out[2u * pos] = ra; out[2u * pos + 1u] = rb; }
                        c += tot;
                    }
                }
            }
            if (blk != 0xffffffffu && li == 0u) {
                const uint32_t cf = !Bb.ok ? 0u : (over ? RT_TILE_OVERFLOW : c);
                bb.tcnt[trow] = cf;
#ifdef RT_DIAG
                if ((lp.flags & 0x1000) && Bb.ok) {
                    atomicAdd(&ctl->dbg_tile[5], 1u);
                    if (cf == RT_TILE_OVERFLOW) atomicAdd(&ctl->dbg_tile[7], 1u); else atomicAdd(&ctl->dbg_tile[6], cf);
                }
#endif
            }
        }
#ifdef RT_DIAG
        if ((lp.flags & 0x2000) && threadIdx.x == 0) {
            const long long tc3 = wall_clock64();
            atomicAdd(&ctl->dbg_tcyc[0], (unsigned long long)(tc1 - tc0)); atomicAdd(&ctl->dbg_tcyc[1], (unsigned long long)(tc2 - tc1));
            atomicAdd(&ctl->dbg_tcyc[2], (unsigned long long)(tc3 - tc2)); atomicMax(&ctl->dbg_tcyc[3], (unsigned long long)(tc3 - tc0));
            atomicMax(&ctl->dbg_tile[0], zn);
        }
#endif
    }
}

// beam path: one lane per (hit, shadow light) pair (rt_shadow.hpp).  The basis is computed in float32 from the
// float64 difference light - P; its error is part of the direction-error budget RT_RHO of the per-ray predicates.
// A pair ends in one of three ways:
//   decided as a whole (every ray lit / every ray unlit): only pair_slot[pair] is written - umbra, fully lit
//     surfaces and P's own sphere on either side of the terminator never reach the per-ray kernel;
//   listed: header + candidate units go to rec[pair], the pair gets a slot in bb.list (wave-aggregated append);
//   numerically delicate basis / too many candidates: listed with the overflow mark, all its rays go to the
//     float64 queue.
// TILES (level 0 of scenes with short beams): the pair filters the candidate row of its (block, light) (k_bin_block) instead of
// walking the tree; the wave stays converged for that (rows are staged by all 64 lanes): lanes past the end of the hit list
// repeat its last pair and store nothing.
template <bool EARLY, bool BOXES, bool TILES>
__global__ void __launch_bounds__(RT_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 4))) k_beam_setup(DScene S, FastScene F, LevelParams lp, RayQ q, HitBuf hb, BeamBuf bb,
                                                         FrameCtl *ctl) {
    extern __shared__ __align__(16) uint2 lds_stack[];
    __shared__ unsigned long long obuf_all[(RT_BLOCK / 64) * 128];
    __shared__ uint32_t tl_all[RT_TOUCH * RT_BLOCK];     // [entry][lane]: touched leaf entries of the lane's traversal
    if (ctl->overflow) return;
    if (lp.prio) __builtin_amdgcn_s_setprio(3);
    const uint32_t n_hits = min(ctl->n_hits[lp.level], lp.cap);
    const uint32_t nsl = (uint32_t)S.n_shadow_lights;
    // (n_hits * nsl < 2^28: checked on the host, enqueue_levels)
    const int lane = threadIdx.x & 63;
    // overflowing pairs are parked in a wave-private LDS buffer and get their list places 64 at a time (one returning
    // atomic on the counter per 64 pairs; see k_shadow_beam)
    unsigned long long *obuf = obuf_all + (threadIdx.x >> 6) * 128;
    uint32_t on = 0u;
    auto store_ovf = [&](uint32_t at, unsigned long long v) { bb.ovlist[at] = v; };
    // record slots: the wave owns the blk_left slots below blk_end (taken bb.rec_blk at a time) and hands them to its pairs in order
    uint32_t blk_end = 0u, blk_left = 0u;
    // A wave takes 64 CONSECUTIVE HITS and ONE light: neighbouring hits (neighbouring pixels at level 0) send
    // nearly the same beam through the tree, the lanes of a wave stay on one path.
    const uint32_t hit_blocks = (n_hits + 63u) >> 6;
    ChunkWalk walk((uint64_t)hit_blocks * nsl * 64u);
    uint64_t first;
    RT_DIAG_ONLY(unsigned long long bsum0 = 0, bsum1 = 0, bsum2 = 0, bsum3 = 0, bsum4 = 0, bsum5 = 0;)
    while (walk.next(first)) {
        RT_DIAG_ONLY(const unsigned long long bc0 = clock64(); unsigned long long bc1 = bc0, bc2 = bc0, bc3 = bc0, bc4 = bc0;)
        const uint32_t wg = (uint32_t)(first >> 6) + (threadIdx.x >> 6);     // (hit block, light)
        const uint32_t hblk = wg / nsl, ml = wg - hblk * nsl;
        const uint32_t h = (hblk << 6) + (uint32_t)lane;
        const bool valid = hblk < hit_blocks && h < n_hits;
        const uint32_t seg = h * nsl + ml;
        bool listed = false, ovf = false;
        uint32_t list_nu = 0u, keep = 0u;
        float4 hdr0 = make_float4(0, 0, 0, 0), hdr1 = hdr0, hdr2 = hdr0, hdr3 = hdr0;
        d3 Pk = mk3(0, 0, 0);
        if (valid || TILES) {
            const uint32_t r = hb.list[(valid || !TILES) ? h : n_hits - 1u];
            const int m = S.shadow_lights[ml];
            // tile beams: the candidate row of the wave's first pair is requested now and used after the pair set-up below
            // (most waves hold pairs of one block: one row)
            TileRow T;
            if (TILES) {
                T.trec = bb.trec; T.tcnt = bb.tcnt; T.row = 0xffffffffu; T.lane = lane;
                // the half row being filtered lives in the wave's columns of the first two stack rows (64 lanes x 8 bytes = 32 units each):
                // the stack is empty until the walk of the pairs without a row starts, after the rows
                T.rowbuf = (float4 *)(lds_stack + (threadIdx.x & ~63u));
                T.rowbuf_hi = (float4 *)(lds_stack + RT_BLOCK + (threadIdx.x & ~63u)) - RT_TILE_HALF;
                T.pre_row = (uint32_t)__builtin_amdgcn_readfirstlane((int)((r >> 6) * nsl + ml));
                T.pre = bb.trec[(uint64_t)T.pre_row * (2u * RT_TILE_NC) + (uint32_t)lane];
                T.pre_cnt = bb.tcnt[T.pre_row];
            }
            const d3 P = mk3(hb.px[r], hb.py[r], hb.pz[r]);
            const DLight &Lt = S.lights[m];
            const d3 dl = ld3(Lt.pos) - P;
            // D = -(light - P)/|light - P|  (Light.py:81: light_direction = -L)
            const float e0 = (float)dl.x, e1 = (float)dl.y, e2 = (float)dl.z;
            // (rsq / rcp: 1 ulp each, like the correctly rounded sqrt + divide they replace; all inside RT_RHO)
            const float il = __builtin_amdgcn_rsqf(fmaf(e0, e0, fmaf(e1, e1, e2 * e2)));
            const float D0 = -e0 * il, D1 = -e1 * il, D2 = -e2 * il;
            // util.get_orthonormal_vector: the branch is taken on the float64 value of D.z (Q7)
            const double az = fabs(dl.z), an = fabs(dl.x) + fabs(dl.y) + az;
            bool delicate = !(an > 0.0) || !(an < 1e7) || !ray_is_tame(P, mk3(1.0, 1.0, 1.0)) || !(an > 1e-30);
            float u0 = 1.0f, u1 = 1.0f, u2 = 0.0f;
            if (dl.z != 0.0) {
                if (az < 1e-280 * an || fabsf(D2) < 1e-12f) delicate = true;   // D.z would underflow, or (1,1,u3) overflows float32
                u2 = -(D0 + D1) * __builtin_amdgcn_rcpf(D2);
            }
            const float iu = __builtin_amdgcn_rsqf(2.0f + u2 * u2);
            u0 *= iu; u1 *= iu; u2 *= iu;
            float r0 = u1 * D2 - u2 * D1, r1 = u2 * D0 - u0 * D2, r2 = u0 * D1 - u1 * D0;     // np.cross(up, D)
            const float ir = __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(fmaf(r0, r0, fmaf(r1, r1, r2 * r2))) + 1e-6f);
            r0 *= ir; r1 *= ir; r2 *= ir;
            const float radius = (float)Lt.radius;
            const float g = radius / (float)lp.sh_rays;
            const float c0 = (radius - g) * 0.5f;
            const float q00 = e0 + c0 * (u0 - r0), q01 = e1 + c0 * (u1 - r1), q02 = e2 + c0 * (u2 - r2);
            float pd[RT_PMAX];
#pragma unroll
            for (int k = 0; k < RT_PMAX; ++k) {
                float v = 0.0f;
                if (k < S.n_planes) v = (float)(dot3(ld3(S.planes[k].n), P) + S.planes[k].d);
                pd[k] = v;
            }
            // box of every sample origin of the pair, relative to P: grid points a,b in [0,s-1] and the
            // component-wise deviations of Light.py:277-279 (|u| <= g on each of up and right)
            const float sm1 = (float)(lp.sh_rays - 1);
            const float cB0 = q00 + 0.5f * sm1 * g * (r0 - u0), cB1 = q01 + 0.5f * sm1 * g * (r1 - u1), cB2 = q02 + 0.5f * sm1 * g * (r2 - u2);
            float h0 = (0.5f * sm1 + 1.0f) * g * (fabsf(u0) + fabsf(r0)), h1 = (0.5f * sm1 + 1.0f) * g * (fabsf(u1) + fabsf(r1)),
                  h2 = (0.5f * sm1 + 1.0f) * g * (fabsf(u2) + fabsf(r2));
            h0 = fmaf(RT_2M18, fabsf(cB0) + h0, h0) + 1e-5f;
            h1 = fmaf(RT_2M18, fabsf(cB1) + h1, h1) + 1e-5f;
            h2 = fmaf(RT_2M18, fabsf(cB2) + h2, h2) + 1e-5f;
            // Some sample may share a coordinate with P (a shadow ray with an exactly zero direction component) and P lies
            // in a bound plane of some node there: the reference culls that node for such a ray (on_node_plane), and the
            // float64 queue repeats that; no filter verdicts for the pair.  (Under a light, in scenes built on round numbers.)
            {
                const bool z0 = fabsf(cB0) <= h0, z1 = fabsf(cB1) <= h1, z2 = fabsf(cB2) <= h2;
                if (z0 || z1 || z2) {
                    if ((z0 && on_node_plane(F, 0, (float)P.x)) || (z1 && on_node_plane(F, 1, (float)P.y)) || (z2 && on_node_plane(F, 2, (float)P.z)))
                        delicate = true;
                }
            }
            uint2 *bs = lds_stack + threadIdx.x;
            uint32_t pcls = 0, n_eval = 0, n_atp = 0;      // whole-pair verdicts of the planes
#pragma unroll
            for (int k = 0; k < RT_PMAX; ++k)
                if (k < S.n_planes) {
                    const uint32_t c = sh_plane_pair((float)S.planes[k].n[0], (float)S.planes[k].n[1], (float)S.planes[k].n[2], pd[k],
                                                     cB0, cB1, cB2, h0, h1, h2);
                    pcls |= c << (2 * k);
                    n_eval += c == RT_PLANE_EVAL;
                    n_atp += c == RT_PLANE_ATP;
                }
            BeamOut bo;
            RT_DIAG_ONLY(bc1 = clock64(); bo.clk_rows = bo.clk_walk = bc1;)
            bo.nu = 0; bo.boxmask = 0; bo.keep = 0; bo.overflow = true; bo.occ_all = false; bo.atp_all = false; bo.trips = 0; bo.why = 0;
            if (!delicate || TILES) {
                {
                    LightQuad Q;
                    Q.u0 = u0; Q.u1 = u1; Q.u2 = u2; Q.r0 = r0; Q.r1 = r1; Q.r2 = r2;
                    const float ur = fabsf(u0 * r0) + fabsf(u1 * r1) + fabsf(u2 * r2);
                    Q.aq = (0.5f * sm1 * g + g * (1.0f + ur)) * 1.0001f + 1e-5f;
                    Q.ja = (g * ((fabsf(u0) + fabsf(r0)) * fabsf(D0) + (fabsf(u1) + fabsf(r1)) * fabsf(D1) + (fabsf(u2) + fabsf(r2)) * fabsf(D2))) * 1.0001f + 1e-5f;
                    if (TILES) {
                        // the pair's block has a candidate row when its sample box lies inside the base box the binning kernels
                        // gave the block's union beam (always, up to rounding; a pair outside walks the tree itself)
                        const float Rs = tile_light_half(radius, g);
                        const float lim0 = Rs + RT_2M17 * (fabsf(cB0) + Rs) + 1e-5f, lim1 = Rs + RT_2M17 * (fabsf(cB1) + Rs) + 1e-5f, lim2 = Rs + RT_2M17 * (fabsf(cB2) + Rs) + 1e-5f;
                        if (h0 <= lim0 && h1 <= lim1 && h2 <= lim2) T.row = (r >> 6) * nsl + ml;
                        const bool active = valid && !delicate;
                        bo = beam_collect<RT_BLOCK, EARLY, BOXES, true>(S, F, bs, tl_all + threadIdx.x, P, cB0, cB1, cB2, h0, h1, h2, Q, lp.flags, active, &T);
                        if (delicate) bo.overflow = true;
                    } else
                    bo = beam_collect<RT_BLOCK, EARLY, BOXES>(S, F, bs, tl_all + threadIdx.x, P, cB0, cB1, cB2, h0, h1, h2, Q, lp.flags);
                }
            }
#ifdef RT_DEBUG_BEAM
            if (valid && (lp.flags & 0x40000)) {
                printf("[setup] pair %u light %d P %.17g %.17g %.17g delicate %d overflow %d why %u occ_all %d atp_all %d nu %u boxmask %x pcls %x trips %u cB %g %g %g h %g %g %g\n", seg, m, P.x, P.y, P.z,
                       (int)delicate, (int)bo.overflow, bo.why, (int)bo.occ_all, (int)bo.atp_all, bo.nu, bo.boxmask, pcls, bo.trips, cB0, cB1, cB2, h0, h1, h2);
                for (uint32_t j = 0; j < RT_TOUCH; ++j) printf("[setup] touched[%u] = prim %u\n", j, tl_all[j * RT_BLOCK + threadIdx.x]);
            }
#endif
#ifdef RT_DIAG
            if (valid && (lp.flags & 0x1000)) {
                atomicMax(&ctl->dbg_max_trips, bo.trips); atomicAdd(&ctl->dbg_sum_trips, (unsigned long long)bo.trips);
                if (bo.overflow) atomicAdd(&ctl->dbg_why[delicate ? 0 : bo.why], 1u);
            }
#endif
            RT_DIAG_ONLY(bc2 = bo.clk_rows; bc3 = bo.clk_walk; bc4 = clock64();)
            // every |sample - P| inside the per-ray filter's range (k_shadow_beam: 1e-4 < nv < 1e7)
            const float sc = __builtin_amdgcn_sqrtf(fmaf(cB0, cB0, fmaf(cB1, cB1, cB2 * cB2))), sh = __builtin_amdgcn_sqrtf(fmaf(h0, h0, fmaf(h1, h1, h2 * h2)));
            const bool range_ok = sc - sh > 1.1e-3f && sc + sh < 0.9e6f;
            uint32_t whole = 0;                             // RT_PAIR_LIT / RT_PAIR_UNLIT when decided
            if (!bo.overflow && range_ok) {
                if (bo.occ_all) whole = RT_PAIR_UNLIT;
                else if (bo.nu == 0 && n_eval == 0) whole = (n_atp + (bo.atp_all ? 1u : 0u)) > 0 ? RT_PAIR_LIT : RT_PAIR_UNLIT;
            }
            // Samples outside the per-ray filter's range (P a million units from the light: a reflection that left the scene and
            // met the ground plane far out): nothing the beam found may be dropped - a pair that is "occluded as a whole" would
            // otherwise be listed WITHOUT the occluder.  Every ray of such a pair is decided in float64.
            if (!range_ok) { bo.overflow = true; delicate = true; }
            if (!valid) {}
            else if (whole) bb.pair_slot[(uint64_t)ml * lp.cap + h] = whole;
            else {
                // a record: listed pairs (header + candidate units, 13 % of the pairs) and overflowing pairs - too many primitives
                // along the beam, or a delicate basis: k_shadow_walk sends each ray through the tree on its own and needs the
                // header only; units = RT_META_OVERFLOW: float64 for every ray
                ovf = bo.overflow;
                listed = !ovf;
                const uint32_t meta = ovf ? ((delicate ? RT_META_OVERFLOW : 0u) | ((uint32_t)m << 16) | (pcls << 24))
                                          : (bo.nu | (bo.boxmask << 4) | ((uint32_t)m << 16) | (pcls << 24) | (bo.atp_all ? 1u << 30 : 0u));
                list_nu = ovf ? 0u : bo.nu;
                keep = ovf ? 0u : bo.keep;
                Pk = P;
                hdr0 = make_float4(q00, q01, q02, g);
                hdr1 = make_float4(u0, u1, u2, __uint_as_float(meta));
                hdr2 = make_float4(r0, r1, r2, __uint_as_float(q.gpix[r]));
                hdr3 = make_float4(pd[0], pd[1], pd[2], __uint_as_float(q.path[r]));
            }
        }
        // ---- record slots, in order from the wave's block (one returning atomic per bb.rec_blk records) ----
        uint32_t slot = RT_LIST_NONE;
        {
            const bool need = listed || ovf;
            const unsigned long long nm = __ballot(need);
            if (nm) {
                const uint32_t c = (uint32_t)__popcll(nm), rk = (uint32_t)__popcll(nm & ((1ull << lane) - 1ull));
                if (c <= blk_left) {
                    slot = blk_end - blk_left + rk;
                    blk_left -= c;
                } else {
                    const uint32_t more = c - blk_left, take = ((more + bb.rec_blk - 1u) / bb.rec_blk) * bb.rec_blk;
                    uint32_t nb = 0;
                    if (lane == 0) nb = atomicAdd(&bb.queue_n[1], take);
                    nb = __shfl(nb, 0);
                    slot = rk < blk_left ? blk_end - blk_left + rk : nb + (rk - blk_left);
                    blk_end = nb + take;
                    blk_left = take - more;
                    if (blk_end > bb.rec_cap) {                // out of records: the frame is void, the host renders it again with more
                        if (lane == 0) ctl->overflow = ctl->overflow_r = 1u;
                        blk_left = 0u;
                        if (slot >= nb) slot = RT_LIST_NONE;
                    }
                }
                if (need && slot != RT_LIST_NONE) {
                    float4 *rec = bb.rec + (uint64_t)slot * RT_PAIR_UNITS;
                    rec[0] = hdr0; rec[1] = hdr1; rec[2] = hdr2; rec[3] = hdr3;
                    if (listed) {
                        beam_write_units<RT_BLOCK, BOXES>(S, tl_all + threadIdx.x, keep, Pk, rec + 4);
                        bb.list[slot] = seg | (list_nu << RT_LIST_NU_SHIFT);
                        bb.pair_slot[(uint64_t)ml * lp.cap + h] = slot;    // pair_slot is laid out [shadow light][hit]: a wave's 64 hits of one light store one 256-byte run
                    } else {
                        bb.list[slot] = RT_LIST_NONE;
                        bb.pair_slot[(uint64_t)ml * lp.cap + h] = RT_PAIR_UNLIT;   // nothing from the per-pair filter: the lit count is hb.lit alone
                    }
                } else if (need) { ovf = false; listed = false; }
            }
        }
        wave_append(ovf, (unsigned long long)seg | ((unsigned long long)slot << 32), obuf, on, &bb.queue_n[2], lane, store_ovf);
#ifdef RT_DIAG
        {
            const unsigned long long bc5 = clock64();
            bsum0 += bc1 - bc0; bsum1 += bc2 - bc1; bsum2 += bc3 - bc2; bsum3 += bc4 - bc3; bsum4 += bc5 - bc4; ++bsum5;
        }
#endif
    }
#ifdef RT_DIAG
    if ((lp.flags & 0x4000) && lane == 0) {
        atomicAdd(&ctl->dbg_bcyc[0], bsum0); atomicAdd(&ctl->dbg_bcyc[1], bsum1); atomicAdd(&ctl->dbg_bcyc[2], bsum2);
        atomicAdd(&ctl->dbg_bcyc[3], bsum3); atomicAdd(&ctl->dbg_bcyc[4], bsum4); atomicAdd(&ctl->dbg_bcyc[5], bsum5);
    }
#endif
    // the unused tail of the wave's last block: no pair there
    if ((uint32_t)lane < blk_left) bb.list[blk_end - blk_left + (uint32_t)lane] = RT_LIST_NONE;
    wave_append_flush(obuf, on, &bb.queue_n[2], lane, store_ovf);
}

// float64-confirmed verdict for shadow ray g (rt_fast.hpp), or the reference-order walk when !FAST
template <bool FAST>
__device__ __forceinline__ bool shadow_ray_lit(const DScene &S, const FastScene &F, uint2 *stk, const LevelParams &lp,
                                               const RayQ &q, const HitBuf &hb, uint64_t g) {
    const uint32_t s = (uint32_t)lp.sh_rays, s2 = s * s;
    const uint32_t nsl = (uint32_t)S.n_shadow_lights;
    const uint64_t seg = g / s2;
    uint32_t h = (uint32_t)(seg / nsl), ml = (uint32_t)(seg % nsl), kk = (uint32_t)(g % s2);
    uint32_t a = kk / s, b = kk % s;
    uint32_t r = hb.list[h];
    int m = S.shadow_lights[ml];
    d3 P = mk3(hb.px[r], hb.py[r], hb.pz[r]);
    d3 up, right, p00;
    exact_basis(S.lights[m], lp.sh_rays, P, up, right, p00);
    double gsz = S.lights[m].radius / (double)(int)s;          // Light.py:254
    d3 src = (p00 - ((double)b * gsz) * up) + ((double)a * gsz) * right;   // Light.py:273-274
    d3 u1 = mk3(0, 0, 0), u2 = mk3(0, 0, 0);
    if (lp.jitter) {
        u1 = jitter3(lp.seed, q.gpix[r], q.path[r], (uint32_t)m, kk, 0u, gsz);
        u2 = jitter3(lp.seed, q.gpix[r], q.path[r], (uint32_t)m, kk, 1u, gsz);
    }
    src = src + (u1 * up + u2 * right);            // Light.py:277-279
    d3 vec = P - src;
    double nv = norm3(vec);
    d3 dir = vec / (nv + RT_EPSILON);              // Light.py:283-286
    if (FAST && ray_is_tame(src, dir) && nv < 1e7) {
        // P = src + (nv + 1e-6) * dir.  Nothing beyond P can be "the nearest hit within 1e-6 of P"
        // unless it is the nearest hit overall, and then the ray is unlit either way; a confirmed hit
        // closer than nv - 3e-6 makes the ray unlit whatever else lies on it.
        float tP = __double2float_ru(nv + RT_EPSILON);
        float tlim = fmaf(tP, RT_2M20, tP + 8e-6f);
        FastState st;
        fast_nearest<RT_BLOCK, true>(S, F, stk, src, dir, tlim, nv - 3e-6, st);      // (walked from P's end: the rays of a pair, neighbours in the wave, start in one leaf)
#ifdef RT_DEBUG_BEAM
        if (lp.flags & 0x40000) printf("[exact] pair %llu sample %u lit %d stop %d nearest id %d dist %.17g nv %.17g src %.17g %.17g %.17g\n", (unsigned long long)seg, kk,
                                        (int)(!st.stop && (norm3(st.best.p - P) < RT_EPSILON)), (int)st.stop, st.best.id, st.best.dist, nv, src.x, src.y, src.z);
#endif
        return !st.stop && (norm3(st.best.p - P) < RT_EPSILON);
    }
    Hit x = nearest_hit_exact(S, src, dir);
    return norm3(x.p - P) < RT_EPSILON;            // Light.py:219-221 (NaN -> unlit)
}

// segmented popcount: consecutive lanes of one (hit, light) pair share a counter, one atomic per segment
__device__ __forceinline__ void add_lit_counts(uint32_t *lit_ctr, bool active, bool lit, uint64_t seg, int lane) {
    unsigned long long litmask = __ballot(lit);
    uint64_t prev = __shfl_up(seg, 1);
    bool leader = active && (lane == 0 || prev != seg);
    unsigned long long leaders = __ballot(leader);
    if (leader) {
        unsigned long long above = (lane == 63) ? 0ull : (leaders >> (lane + 1));
        int next = above ? (lane + 1 + (__ffsll((long long)above) - 1)) : 64;
        unsigned long long segmask = (next == 64 ? ~0ull : ((1ull << next) - 1ull)) & ~((1ull << lane) - 1ull);
        uint32_t cnt = (uint32_t)__popcll(litmask & segmask);
        if (cnt) atomicAdd(&lit_ctr[seg], cnt);
    }
}

// one lane per soft-shadow ray, every ray decided in float64 (Light.compute_shadows, Light.py:178-288)
template <bool FAST>
__global__ void __launch_bounds__(RT_BLOCK) k_shadow(DScene S, FastScene F, LevelParams lp, RayQ q, HitBuf hb, FrameCtl *ctl) {
    extern __shared__ uint2 lds_stack[];
    uint2 *stk = lds_stack + threadIdx.x;
    if (ctl->overflow) return;
    if (lp.prio) __builtin_amdgcn_s_setprio(3);
    const uint32_t n_hits = min(ctl->n_hits[lp.level], lp.cap);
    const uint32_t s2 = (uint32_t)(lp.sh_rays * lp.sh_rays);
    const uint64_t total = (uint64_t)n_hits * (uint32_t)S.n_shadow_lights * s2;
    ChunkWalk walk(total);
    uint64_t first;
    const int lane = threadIdx.x & 63;
    while (walk.next(first)) {
        uint64_t g = first + threadIdx.x;
        bool active = g < total;
        uint64_t seg = active ? g / s2 : ~0ull;       // (hit, light) pair: h * nsl + ml
        bool lit = active && shadow_ray_lit<FAST>(S, F, stk, lp, q, hb, g);
        add_lit_counts(hb.lit, active, lit, seg, lane);
    }
}

// one lane per soft-shadow ray of the LISTED pairs (k_beam_setup), float32 filtered predicates on the pair's
// candidate list (rt_shadow.hpp); undecided rays go to the queue.  A chunk of RT_BLOCK lanes holds lp.spb whole
// pairs = consecutive slots of bb.list.  Staging is per WAVE: the <= 9 pairs a wave's 64 lanes belong to
// (s*s >= 9) are <= 108 units = two 16-byte loads per lane into the wave's own LDS slice, so no workgroup barrier
// is needed (a wave's DS operations execute in order).  Software pipeline, per wave: while chunk i is classified,
// the records of chunk i+1 are in flight (parked in registers until the top of the next iteration) and so are
// the pair numbers of chunk i+2.
// The lane -> (pair, sample) map is the same in every chunk.
#define RT_WAVE_UNITS 192
__global__ void __launch_bounds__(RT_BLOCK) k_shadow_beam(DScene S, LevelParams lp, HitBuf hb, BeamBuf bb, FrameCtl *ctl) {
    __shared__ float4 pair_lds_all[(RT_BLOCK / 64) * RT_WAVE_UNITS];
    __shared__ unsigned long long qbuf_all[(RT_BLOCK / 64) * 128];
    if (ctl->overflow) return;
    if (lp.prio) __builtin_amdgcn_s_setprio(3);
    const uint32_t NONE = 0xffffffffu;
    const uint32_t s = (uint32_t)lp.sh_rays, s2 = s * s;
    const uint32_t queue_cap = bb.queue_n[3];
    const uint32_t nlist = min(bb.queue_n[1], bb.rec_cap);      // record slots handed out (RT_LIST_NONE where no listed pair sits)
    const uint32_t nchunks = (nlist + lp.spb - 1) / lp.spb;
    ChunkWalk walk((uint64_t)nchunks * RT_BLOCK);
    uint64_t first;
    const int lane = threadIdx.x & 63;
    const uint32_t wave = threadIdx.x >> 6, wave0 = threadIdx.x & ~63u;
    float4 *pair_lds = pair_lds_all + wave * RT_WAVE_UNITS;
    // Undecided rays are parked in a wave-private LDS buffer and leave 64 at a time: ONE returning atomic on the
    // queue counter per 64 rays.  (One per wave and chunk was the bottleneck of this kernel: returning atomics
    // on a single address complete at ~30 ns each.)
    unsigned long long *qbuf = qbuf_all + wave * 128;
    uint32_t qn = 0u;
    const uint32_t sib = __umulhi(threadIdx.x, lp.magic_s2);          // pair within the chunk
    const uint32_t kk = threadIdx.x - sib * s2;                        // sample within the pair
    const uint32_t a = s == 1 ? kk : __umulhi(kk, lp.magic_s), b = kk - a * s;   // Light.py:265: jj = a, ii = b
    const float fa = (float)a, fb = (float)b;
    // pairs of this wave: first .. last (within the chunk); lane l < npw fetches the pair number of pair wp0 + l
    const uint32_t wp0 = __umulhi(wave0, lp.magic_s2);
    const uint32_t wp1 = min(__umulhi(wave0 + 63u, lp.magic_s2), lp.spb - 1u);
    const uint32_t npw = wp1 >= wp0 ? wp1 - wp0 + 1u : 0u;             // <= 9 (host checks s*s >= 9): <= RT_WAVE_UNITS units
    const uint32_t mypair = sib < lp.spb ? sib - wp0 : 63u;            // lane 63 never fetches a pair number: NONE
    const float4 *rec = pair_lds + (sib < lp.spb ? (sib - wp0) * RT_PAIR_UNITS : 0u);
    // staging: unit i of the wave's slice = unit i % 12 of its pair i / 12
    const uint32_t i0u = (uint32_t)lane, i1u = (uint32_t)lane + 64u, i2u = (uint32_t)lane + 128u;
    const uint32_t sp0 = i0u / RT_PAIR_UNITS, su0 = i0u % RT_PAIR_UNITS, sp1 = i1u / RT_PAIR_UNITS, su1 = i1u % RT_PAIR_UNITS,
                   sp2 = i2u / RT_PAIR_UNITS, su2 = i2u % RT_PAIR_UNITS;
    const bool has0 = sp0 < npw, has1 = sp1 < npw, has2 = sp2 < npw;
    // Lit verdicts leave as one 64-bit ballot per wave and chunk (k_shade counts the bits of a pair).  No global
    // atomic or store sits in the steady-state loop - the wait for the staged loads at the top of an iteration
    // would otherwise also wait for them (one vmcnt) - : lane l of the wave parks the ballot of iteration
    // 64 w + l, and the wave stores 64 of them at once.
    unsigned long long keep = 0ull;
    uint32_t it = 0u, keep_chunk = 0u;
    float pn[RT_PMAX][3];
    for (int k = 0; k < RT_PMAX; ++k)
        for (int c = 0; c < 3; ++c) pn[k][c] = k < S.n_planes ? (float)S.planes[k].n[c] : 0.0f;
    const uint32_t zkey = lp.seed * 0x85EBCA6Bu + kk;
    float4 st0 = make_float4(0, 0, 0, 0), st1 = st0, st2 = st0;
#define RT_LOAD_PAIRNO(dst_, chunk_) do { const uint32_t sl_ = (chunk_) * lp.spb + wp0 + (uint32_t)lane; \
        dst_ = ((uint32_t)lane < npw && sl_ < nlist) ? bb.list[sl_] : NONE; } while (0)
    // a list entry = pair | units used << 28: only the header and the used candidate units of a record are fetched
#define RT_LOAD_RECS(pairno_, chunk_) do { const uint32_t g0_ = __shfl((pairno_), (int)sp0), g1_ = __shfl((pairno_), (int)sp1), g2_ = __shfl((pairno_), (int)sp2); \
        const uint64_t b_ = (uint64_t)(chunk_) * lp.spb + wp0;      /* the record of a listed pair sits at its list slot */ \
        if (has0 && g0_ != NONE && su0 < 4u + (g0_ >> RT_LIST_NU_SHIFT)) st0 = bb.rec[(b_ + sp0) * RT_PAIR_UNITS + su0]; \
        if (has1 && g1_ != NONE && su1 < 4u + (g1_ >> RT_LIST_NU_SHIFT)) st1 = bb.rec[(b_ + sp1) * RT_PAIR_UNITS + su1]; \
        if (has2 && g2_ != NONE && su2 < 4u + (g2_ >> RT_LIST_NU_SHIFT)) st2 = bb.rec[(b_ + sp2) * RT_PAIR_UNITS + su2]; } while (0)
    uint32_t cur = 0u, nxt = 0u, pno_cur = NONE, pno_nxt = NONE;
    bool have = walk.next(first), have_n = false;
    if (have) {
        cur = (uint32_t)(first / RT_BLOCK);
        RT_LOAD_PAIRNO(pno_cur, cur);
        RT_LOAD_RECS(pno_cur, cur);
        have_n = walk.next(first);
        if (have_n) {
            nxt = (uint32_t)(first / RT_BLOCK);
            RT_LOAD_PAIRNO(pno_nxt, nxt);
        }
    }
    while (have) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");       // earlier reads of the slice are done (WAR)
        __builtin_amdgcn_wave_barrier();
        if (has0) pair_lds[i0u] = st0;
        if (has1) pair_lds[i1u] = st1;
        if (has2) pair_lds[i2u] = st2;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        uint32_t n2 = 0u, pno_n2 = NONE;
        bool have_n2 = false;
        if (have_n) {                                // in flight during the arithmetic below
            RT_LOAD_RECS(pno_nxt, nxt);
            have_n2 = walk.next(first);
            if (have_n2) {
                n2 = (uint32_t)(first / RT_BLOCK);
                RT_LOAD_PAIRNO(pno_n2, n2);
            }
        }
        const uint32_t ent = __shfl(pno_cur, (int)mypair);            // the lane's (hit, light) pair
        const bool active = ent != NONE;
        const uint32_t seg = ent & ((1u << RT_LIST_NU_SHIFT) - 1u);
        bool lit = false, unsure = false;
        if (active) {
            const float4 h0 = rec[0], h1 = rec[1], h2 = rec[2], h3 = rec[3];
            const float g = h0.w;
            const uint32_t meta = __float_as_uint(h1.w);
            const uint32_t nu = meta & 0xfu, boxmask = (meta >> 4) & 0xfffu;
            if (nu == RT_META_OVERFLOW) unsure = true;
            else {
                float j10 = 0, j11 = 0, j12 = 0, j20 = 0, j21 = 0, j22 = 0;   // Light.py:277-278: one deviation per component
                if (lp.jitter) {                     // jitter6() of the oracle, in float32
                    const uint32_t m = (meta >> 16) & 0xffu;
                    const float k16 = 1.0f / 65536.0f, g2 = 2.0f * g;
                    uint32_t x = __float_as_uint(h2.w), y = __float_as_uint(h3.w) * 0x9E3779B1u + m, z = zkey;
                    pcg3d(x, y, z);
                    j10 = fmaf(g2, (float)(x >> 16) * k16, -g); j11 = fmaf(g2, (float)(y >> 16) * k16, -g);
                    j12 = fmaf(g2, (float)(z >> 16) * k16, -g);
                    j20 = fmaf(g2, (float)(x & 0xffffu) * k16, -g); j21 = fmaf(g2, (float)(y & 0xffffu) * k16, -g);
                    j22 = fmaf(g2, (float)(z & 0xffffu) * k16, -g);
                }
                // w = sample - P (Light.py:273-279), float32
                const float bg = fb * g, ag = fa * g;
                float w0 = h0.x - bg * h1.x + ag * h2.x + (j10 * h1.x + j20 * h2.x);
                float w1 = h0.y - bg * h1.y + ag * h2.y + (j11 * h1.y + j21 * h2.y);
                float w2 = h0.z - bg * h1.z + ag * h2.z + (j12 * h1.z + j22 * h2.z);
                const float ww = fmaf(w0, w0, fmaf(w1, w1, w2 * w2));
                const float inv_nv = __builtin_amdgcn_rsqf(ww);      // 1 ulp: inside the direction-error budget RT_RHO
                const float nv = ww * inv_nv;
                const float nvlim = nv - fmaf(nv, 4.0f * RT_RHO, 1e-5f);
                w0 *= inv_nv; w1 *= inv_nv; w2 *= inv_nv;
                int verdict = (!(nv > 1e-4f) || !(nv < 1e7f)) ? SH_UNSURE : 0;
                int n_atp = (int)((meta >> 30) & 1u);                // P's own sphere, decided for the whole pair
                const float pdk[RT_PMAX] = {h3.x, h3.y, h3.z};
                const uint32_t pcls = (meta >> 24) & 0x3fu;
#pragma unroll
                for (int k = 0; k < RT_PMAX; ++k) {
                    if (k < S.n_planes) {
                        const uint32_t cls = (pcls >> (2 * k)) & 3u;
                        if (cls == RT_PLANE_EVAL) {
                            const int v = sh_plane(pn[k][0], pn[k][1], pn[k][2], pdk[k], w0, w1, w2, nv, nvlim, S.plane_kw);
                            verdict |= v;
                            n_atp += v & SH_ATP;
                        } else n_atp += cls == RT_PLANE_ATP;
                    }
                }
                RayInv qi;
                if (boxmask) qi = sh_ray_inv(w0, w1, w2);
                for (uint32_t j = 0; j < nu && !(verdict & SH_OCCLUDE);) {
                    const float4 ua = rec[4 + j];
                    int v;
                    if ((boxmask >> j) & 1u) {
                        const float4 ub = rec[5 + j];
                        v = sh_box(ua.x, ua.y, ua.z, ua.w, ub.x, ub.y, qi, nvlim);
                        j += 2;
                    } else {
                        v = sh_sphere(ua.x, ua.y, ua.z, ua.w, w0, w1, w2, nv, nvlim);
                        j += 1;
                    }
                    verdict |= v;
                    n_atp += v & SH_ATP;
                }
#ifdef RT_DEBUG_BEAM
                if (lp.flags & 0x40000) printf("[beam] pair %u sample %u verdict %d n_atp %d nu %u nv %.9g w %.9g %.9g %.9g\n", seg, kk, verdict, n_atp, nu, nv, w0, w1, w2);
#endif
                if (verdict & SH_OCCLUDE) unsure = false;        // one certain occluder settles the ray
                else if (verdict & SH_UNSURE) unsure = true;
                else lit = n_atp > 0;
            }
        }
        // undecided rays -> wave buffer -> queue
        const unsigned long long um = __ballot(unsure);
        if (um) {
            if (unsure) qbuf[qn + (uint32_t)__popcll(um & ((1ull << lane) - 1ull))] = (unsigned long long)seg * s2 + kk;
            qn += (uint32_t)__popcll(um);
            if (qn >= 64u) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&bb.queue_n[0], 64u);
                base = __shfl(base, 0);
                const unsigned long long v = qbuf[lane], v2 = qbuf[64 + lane];
                if (base + (uint32_t)lane < queue_cap) bb.queue[base + (uint32_t)lane] = v;
                else ctl->overflow = ctl->overflow_q = 1u;             // frame is void; the host re-renders with a larger queue
                qn -= 64u;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                if ((uint32_t)lane < qn) qbuf[lane] = v2;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
            }
        }
        const unsigned long long lm = __ballot(lit);
        if ((uint32_t)lane == (it & 63u)) { keep = lm; keep_chunk = cur; }
        ++it;
        if ((it & 63u) == 0u) bb.litbits[(uint64_t)keep_chunk * (RT_BLOCK / 64) + wave] = keep;
        cur = nxt; pno_cur = pno_nxt; have = have_n;
        nxt = n2; pno_nxt = pno_n2; have_n = have_n2;
    }
    if ((uint32_t)lane < (it & 63u)) bb.litbits[(uint64_t)keep_chunk * (RT_BLOCK / 64) + wave] = keep;
    if (qn) {                                        // the rest of the wave's undecided rays
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&bb.queue_n[0], qn);
        base = __shfl(base, 0);
        if ((uint32_t)lane < qn) {
            if (base + (uint32_t)lane < queue_cap) bb.queue[base + (uint32_t)lane] = qbuf[lane];
            else ctl->overflow = ctl->overflow_q = 1u;
        }
    }
}

// one lane per soft-shadow ray of the pairs in ovlist (their beam met more primitives than a pair record holds):
// the ray is derived from the pair header exactly as in k_shadow_beam and walks the tree itself (ray_walk,
// rt_shadow.hpp) - float32 filtered predicates on float64-derived operands, undecided rays to the float64 queue.
// Lanes of one pair are neighbours: s*s consecutive items.
__global__ void __launch_bounds__(RT_BLOCK) k_shadow_walk(DScene S, FastScene F, LevelParams lp, HitBuf hb, BeamBuf bb, FrameCtl *ctl) {
    extern __shared__ uint2 lds_stack[];
    __shared__ unsigned long long qbuf_all[(RT_BLOCK / 64) * 128];
    uint2 *stk = lds_stack + threadIdx.x;
    const uint32_t s = (uint32_t)lp.sh_rays, s2 = s * s;
    const uint64_t total = (uint64_t)bb.queue_n[2] * s2;
    if (ctl->overflow || total < (uint64_t)lp.walk_min || total == 0) return;     // few such rays: k_shadow_queue takes them
    if (lp.prio) __builtin_amdgcn_s_setprio(3);
    const uint32_t queue_cap = bb.queue_n[3];
    const int lane = threadIdx.x & 63;
    unsigned long long *qbuf = qbuf_all + (threadIdx.x >> 6) * 128;
    uint32_t qn = 0u;
    auto store_q = [&](uint32_t slot, unsigned long long v) {
        if (slot < queue_cap) bb.queue[slot] = v;
        else ctl->overflow = ctl->overflow_q = 1u;                       // frame is void; the host re-renders with a larger queue
    };
    float pn[RT_PMAX][3];
    for (int k = 0; k < RT_PMAX; ++k)
        for (int c = 0; c < 3; ++c) pn[k][c] = k < S.n_planes ? (float)S.planes[k].n[c] : 0.0f;
    ChunkWalk walk(total);
    uint64_t first;
    while (walk.next(first)) {
        const uint64_t i = first + threadIdx.x;
        const bool active = i < total;
        bool lit = false, unsure = false;
        uint32_t seg = 0u, kk = 0u;
        if (active) {
            const uint32_t pi = (uint32_t)(i / s2);
            kk = (uint32_t)(i - (uint64_t)pi * s2);
            const unsigned long long ov = bb.ovlist[pi];
            seg = (uint32_t)ov;
            const float4 *rec = bb.rec + (ov >> 32) * RT_PAIR_UNITS;
            const float4 h0 = rec[0], h1 = rec[1], h2 = rec[2], h3 = rec[3];
            const float g = h0.w;
            const uint32_t meta = __float_as_uint(h1.w);
            if ((meta & 0xfu) == RT_META_OVERFLOW) unsure = true;
            else {
                const uint32_t a = kk / s, b = kk - a * s;            // Light.py:265: jj = a, ii = b
                float j10 = 0, j11 = 0, j12 = 0, j20 = 0, j21 = 0, j22 = 0;
                if (lp.jitter) {                     // jitter6() of the oracle, in float32 (as in k_shadow_beam)
                    const uint32_t m = (meta >> 16) & 0xffu;
                    const float k16 = 1.0f / 65536.0f, g2 = 2.0f * g;
                    uint32_t x = __float_as_uint(h2.w), y = __float_as_uint(h3.w) * 0x9E3779B1u + m, z = lp.seed * 0x85EBCA6Bu + kk;
                    pcg3d(x, y, z);
                    j10 = fmaf(g2, (float)(x >> 16) * k16, -g); j11 = fmaf(g2, (float)(y >> 16) * k16, -g);
                    j12 = fmaf(g2, (float)(z >> 16) * k16, -g);
                    j20 = fmaf(g2, (float)(x & 0xffffu) * k16, -g); j21 = fmaf(g2, (float)(y & 0xffffu) * k16, -g);
                    j22 = fmaf(g2, (float)(z & 0xffffu) * k16, -g);
                }
                const float bg = (float)b * g, ag = (float)a * g;
                const float sw0 = h0.x - bg * h1.x + ag * h2.x + (j10 * h1.x + j20 * h2.x);
                const float sw1 = h0.y - bg * h1.y + ag * h2.y + (j11 * h1.y + j21 * h2.y);
                const float sw2 = h0.z - bg * h1.z + ag * h2.z + (j12 * h1.z + j22 * h2.z);
                const float ww = fmaf(sw0, sw0, fmaf(sw1, sw1, sw2 * sw2));
                const float inv_nv = __builtin_amdgcn_rsqf(ww);
                const float nv = ww * inv_nv;
                const float nvlim = nv - fmaf(nv, 4.0f * RT_RHO, 1e-5f);
                const float w0 = sw0 * inv_nv, w1 = sw1 * inv_nv, w2 = sw2 * inv_nv;
                int verdict = (!(nv > 1e-4f) || !(nv < 1e7f)) ? SH_UNSURE : 0;
                int n_atp = 0;
                const float pdk[RT_PMAX] = {h3.x, h3.y, h3.z};
                const uint32_t pcls = (meta >> 24) & 0x3fu;
#pragma unroll
                for (int k = 0; k < RT_PMAX; ++k) {
                    if (k < S.n_planes) {
                        const uint32_t cls = (pcls >> (2 * k)) & 3u;
                        if (cls == RT_PLANE_EVAL) {
                            const int v = sh_plane(pn[k][0], pn[k][1], pn[k][2], pdk[k], w0, w1, w2, nv, nvlim, S.plane_kw);
                            verdict |= v;
                            n_atp += v & SH_ATP;
                        } else n_atp += cls == RT_PLANE_ATP;
                    }
                }
                if (!(verdict & (SH_OCCLUDE | SH_UNSURE))) {     // (an unsure ray is decided in float64 whatever the walk finds)
                    const uint32_t r = hb.list[seg / (uint32_t)S.n_shadow_lights];
                    const d3 P = mk3(hb.px[r], hb.py[r], hb.pz[r]);
                    verdict |= ray_walk<RT_BLOCK>(S, F, stk, P, sw0, sw1, sw2, w0, w1, w2, nv, nvlim, n_atp);
                }
                if (verdict & SH_OCCLUDE) unsure = false;
                else if (verdict & SH_UNSURE) unsure = true;
                else lit = n_atp > 0;
            }
        }
        if (lit) atomicAdd(&hb.lit[seg], 1u);
        wave_append(unsure, (unsigned long long)seg * s2 + kk, qbuf, qn, &bb.queue_n[0], lane, store_q);
    }
    wave_append_flush(qbuf, qn, &bb.queue_n[0], lane, store_q);
}

// the rays the filter could not decide, in float64
__global__ void __launch_bounds__(RT_BLOCK) k_shadow_queue(DScene S, FastScene F, LevelParams lp, RayQ q, HitBuf hb, BeamBuf bb, FrameCtl *ctl) {
    extern __shared__ uint2 lds_stack[];
    uint2 *stk = lds_stack + threadIdx.x;
    if (ctl->overflow) return;
    if (lp.prio) __builtin_amdgcn_s_setprio(3);
    // work = the rays the filter queued, then every ray of the pairs that never reached the filter (candidate
    // overflow, delicate basis): those are expanded here, s*s consecutive items per pair
    const uint32_t n = min(bb.queue_n[0], bb.queue_n[3]);
    const uint32_t s2 = (uint32_t)(lp.sh_rays * lp.sh_rays);
    const uint64_t n_ovf = (uint64_t)bb.queue_n[2] * s2;
    // walk_min == 0xffffffff: k_shadow_walk was not launched at all (a rule of the scene, enqueue_levels) - every such ray is
    // taken here, however many there are; else k_shadow_walk has dealt with them from walk_min rays on
    const bool take_ovf = lp.walk_min == 0xffffffffu || n_ovf < (uint64_t)lp.walk_min;
    const uint64_t total = (uint64_t)n + (take_ovf ? n_ovf : 0ull);
    ChunkWalk walk(total);
    uint64_t first;
    while (walk.next(first)) {
        uint64_t i = first + threadIdx.x;
        if (i >= total) continue;
        uint64_t g;
        if (i < n) g = bb.queue[i];
        else {
            const uint64_t j = i - n;
            const uint32_t pi = (uint32_t)(j / s2);
            g = (uint64_t)(uint32_t)bb.ovlist[pi] * s2 + (j - (uint64_t)pi * s2);
        }
        if (shadow_ray_lit<true>(S, F, stk, lp, q, hb, g)) atomicAdd(&hb.lit[g / s2], 1u);
    }
}

// Phong shading for every hit of the level (the children are made by k_intersect).
#define RT_SHADE_LIGHTS 1
#define RT_SHADE_STORE 4     // level 0 of a frame: fb is this pass's own buffer and holds 0 at every hit's slot (k_intersect's
                             // camera-ray pass), one hit per slot - a plain store replaces three float64 atomics per hit
// fbc: where the background goes that the children of the LAST level's hits would return (ray_tracer.py:62-63)
template <int MODE>
__global__ void __launch_bounds__(RT_BLOCK) k_shade(DScene S, LevelParams lp, RayQ q, HitBuf hb, FrameCtl *ctl,
                                                    double *fb, double *fbc, const unsigned long long *litbits, const uint32_t *pair_slot, const uint32_t *queue_n) {
    if (ctl->overflow) return;
    if (lp.prio) __builtin_amdgcn_s_setprio(3);
    if (lp.beam && blockIdx.x == 0 && threadIdx.x == 0) { ctl->n_queued[lp.level] = queue_n[0]; ctl->n_listed[lp.level] = queue_n[1]; ctl->n_walked[lp.level] = queue_n[2] * (uint32_t)(lp.sh_rays * lp.sh_rays); }
    const uint32_t n_hits = min(ctl->n_hits[lp.level], lp.cap);
    const uint32_t nsl = (uint32_t)S.n_shadow_lights;
    const double s2 = (double)(lp.sh_rays * lp.sh_rays);
    ChunkWalk walk(n_hits);
    uint64_t first;
    while (walk.next(first)) {
        uint64_t h = first + threadIdx.x;
        if (h >= n_hits) continue;
        uint32_t r = hb.list[h];
        d3 d = mk3(q.dx[r], q.dy[r], q.dz[r]);
        d3 thr = mk3(1.0, 1.0, 1.0);
        if (!(lp.unit_thr && lp.level == 0)) thr = mk3(q.tr[r], q.tg[r], q.tb[r]);
        d3 P = mk3(hb.px[r], hb.py[r], hb.pz[r]);
        const uint32_t slot = q.slot[r];
        int32_t id = hb.id[r], mi;
        d3 N = surface_normal(S, id, P, mi);                         // ray_tracer.py:77-78
        d3 Kd = mk3(0, 0, 0), Ks = mk3(0, 0, 0), Kr = mk3(0, 0, 0);
        double T = 0;
        int32_t phong = 0, mflags = 0;
        if (mi >= 0 && mi < S.n_mats) {                              // Material.py:23-62
            const DMat &M = S.mats[mi];
            Kd = ld3(M.kd); Ks = ld3(M.ks); Kr = ld3(M.kr);
            T = M.T; phong = M.phong; mflags = M.flags;
        }
        d3 V = neg3(d);
        d3 Nh = N / norm3(N);                                        // ray_functions.py:98-99
        d3 diff = mk3(0, 0, 0), spec = mk3(0, 0, 0);
        uint32_t ml = 0;
        for (int m = 0; m < S.n_lights; ++m) {
            const DLight &Lt = S.lights[m];
            d3 dl = ld3(Lt.pos) - P;                                 // Light.py:35-53
            d3 L = dl / norm3(dl);
            double I = 1.0;                                          // Light.py:199-200
            if (Lt.shadow != 0) {
                uint64_t seg = (uint64_t)h * nsl + ml;
                uint32_t cnt = hb.lit[seg];
                hb.lit[seg] = 0;                                     // ready for the next level
                if (lp.beam) {                                       // + the beam path's verdicts
                    const uint32_t ps = pair_slot[(uint64_t)ml * lp.cap + h];
                    const uint32_t ss = (uint32_t)(lp.sh_rays * lp.sh_rays);
                    if (ps == RT_PAIR_LIT) cnt += ss;
                    else if (ps != RT_PAIR_UNLIT) {                  // listed: lanes [lo, lo + s*s) of the slot's chunk
                        const uint32_t chunk = ps / lp.spb, sib = ps - chunk * lp.spb;
                        const uint32_t lo = sib * ss, hi = lo + ss;
                        const unsigned long long *bits = litbits + (uint64_t)chunk * (RT_BLOCK / 64);
                        for (uint32_t w = lo >> 6; w <= ((hi - 1u) >> 6); ++w) {
                            const uint32_t b0 = w << 6;
                            const uint32_t l = lo > b0 ? lo - b0 : 0u, u = hi < b0 + 64u ? hi - b0 : 64u;   // bit range [l, u) of word w
                            const unsigned long long m = (u == 64u ? ~0ull : ((1ull << u) - 1ull)) & ~((1ull << l) - 1ull);
                            cnt += (uint32_t)__popcll(bits[w] & m);
                        }
                    }
                }
                ml++;
                double frac = (double)cnt / s2;                  // Light.py:223
                I = (1.0 - Lt.shadow) + Lt.shadow * frac;            // Light.py:226
            }
            I = I < 0 ? 0 : (I > 1 ? 1 : I);                         // Light.py:84
            d3 col = ld3(Lt.color);
            d3 Ims = (Lt.spec * I) * col;                            // Light.py:87
            d3 Imd = I * col;                                        // Light.py:91
            double ldn = dot3(L, N);                                 // Light.py:123-125
            ldn = isnan(ldn) ? ldn : (ldn > 0 ? ldn : 0);
            d3 dterm = (ldn * Kd) * Imd;
            diff = (m == 0) ? dterm : diff + dterm;
            double dp = dot3(L, Nh);                                 // ray_functions.py:101-103
            d3 R = (2 * dp) * Nh - L;
            double rv = dot3(R, V);                                  // Light.py:169
            double pw = ipow(rv, phong);
            d3 sterm = (pw * Ks) * Ims;                              // Light.py:171
            spec = (m == 0) ? sterm : spec + sterm;
        }
        spec = mk3(nan_to_num(spec.x), nan_to_num(spec.y), nan_to_num(spec.z));   // Light.py:173
        double nt = 1.0 - T;
        d3 base = (diff + spec) * mk3(nt, nt, nt);                   // ray_tracer.py:105-106
        if (MODE & RT_SHADE_STORE) {
            const d3 v = thr * base;
            fb[3ull * slot] = v.x; fb[3ull * slot + 1] = v.y; fb[3ull * slot + 2] = v.z;
        } else fb_add(fb, slot, thr * base);
        if (lp.last_level) {                                         // the children would return the background (ray_tracer.py:62-63)
            if (mflags & 1) fb_add(fbc, slot, (thr * mk3(T, T, T)) * ld3(S.bg));      // ray_tracer.py:108-116
            if (mflags & 2) fb_add(fbc, slot, (thr * Kr) * ld3(S.bg));                // ray_tracer.py:119-129
        }
    }
}

// One recursion level of a small scene, fused (rt_small.hpp): nearest hit, soft shadows, Phong and child rays of every
// queued ray in one kernel, the scene tables in LDS.  GEN: level 0 of a frame, camera rays made here (ray_functions.py:7-43);
// its shading is STORED into fb (= the frame's fb0, one hit per slot) and the other two framebuffers are zeroed on the way.
// Framebuffers as in the general pipeline: fb <- shading (+ background the children of the last level would return),
// fbm <- background of rays of levels >= 1 that miss.
template <bool GEN>
__global__ void __launch_bounds__(RT_BLOCK) __attribute__((amdgpu_waves_per_eu(2, 2))) k_small_level(DScene S, SmallLayout L, LevelParams lp, RayQ q, RayQ qn, FrameCtl *ctl, double *fb,
                                                          double *fbm, double *fb_side, int32_t *ids_out, GenParams gp) {
    extern __shared__ uint2 lds_stack[];
    const LdsScene V = stage_scene(S, L, (RT_LDS unsigned char *)lds_stack);
    if (ctl->overflow) return;
    const uint32_t n = GEN ? gp.n_slots : min(ctl->n_rays[lp.level], lp.cap);
    if (GEN && blockIdx.x == 0 && threadIdx.x == 0) ctl->n_rays[0] = gp.n_slots;
    // Thin levels (a level of pool.txt from the fourth on holds a few thousand rays): the kernel's time is then ONE ray's chain -
    // nearest hit, every light's shadow rays one after the other, Phong - in one lane of a nearly empty GPU.  With at most an
    // eighth of the grid's lanes in rays, a ray takes eight lanes: all find its hit, lane m makes light m's shadow rays and
    // Phong terms, and the terms are summed in light order (shuffles; the same float64 additions in the same order), the
    // first lane of the eight writes.
    const bool spread = !GEN && S.n_lights <= 8 && (uint64_t)n * 8u <= (uint64_t)gridDim.x * RT_BLOCK;
    const uint32_t sub = spread ? (threadIdx.x & 7u) : 0u;
    ChunkWalk walk(n, spread ? RT_BLOCK / 8 : RT_BLOCK);
    uint64_t first;
    const int lane = threadIdx.x & 63;
    unsigned long long valid_total = 0, n_refl = 0, n_trans = 0, n_hit = 0;
    const uint32_t s = (uint32_t)lp.sh_rays, s2 = s * s;
    while (walk.next(first)) {
        const uint64_t k = first + (spread ? (threadIdx.x >> 3) : threadIdx.x);
        bool active = k < n;
        uint32_t slot = RT_INVALID_SLOT, gpix = 0, path = 1u;
        d3 o = mk3(0, 0, 0), d = mk3(0, 0, 1), thr = mk3(1.0, 1.0, 1.0);
        if (GEN) {
            const bool in_range = active;
            if (active) active = gen_ray<false>(gp, q, k, o, d, slot);      // (the queue entry is written for rt_initial_rays-style readers: direction, slot, pixel)
            valid_total += active ? 1ull : 0ull;
            const uint32_t zs = active ? slot : ((in_range && gp.world > 1) ? (uint32_t)k : RT_INVALID_SLOT);
            if (zs != RT_INVALID_SLOT) {
                if (!active) { fb[3ull * zs] = 0.0; fb[3ull * zs + 1] = 0.0; fb[3ull * zs + 2] = 0.0; }
                fbm[3ull * zs] = 0.0; fbm[3ull * zs + 1] = 0.0; fbm[3ull * zs + 2] = 0.0;
                fb_side[3ull * zs] = 0.0; fb_side[3ull * zs + 1] = 0.0; fb_side[3ull * zs + 2] = 0.0;
            }
            if (active) gpix = gp.world > 1 ? q.gpix[k] : slot;           // (whole frames: the slot IS the global pixel index)
        } else {
            slot = active ? q.slot[k] : RT_INVALID_SLOT;
            active = active && slot != RT_INVALID_SLOT;
            if (active) {
                o = mk3(q.ox[k], q.oy[k], q.oz[k]);
                d = mk3(q.dx[k], q.dy[k], q.dz[k]);
                thr = mk3(q.tr[k], q.tg[k], q.tb[k]);
                gpix = q.gpix[k]; path = q.path[k];
            }
        }
        int n_child = 0;
        d3 o_t, d_t, thr_t, o_r, d_r, thr_r;
        bool want_t = false, want_r = false;
        double *fb_children = GEN ? fb_side : fb;       // where the background of the last level's would-be children goes (k_shade: ctx->fb)
        if (active) {
            const Hit hit = nearest_hit_small(S, V, o, d);
            if (ids_out && lp.level == 0) ids_out[slot] = hit.id;
            if (hit.id == 0) {
                // ray_tracer.py:65,72,132: misses keep the background colour
                if (GEN) { fb[3ull * slot] = S.bg[0]; fb[3ull * slot + 1] = S.bg[1]; fb[3ull * slot + 2] = S.bg[2]; }
                else if (sub == 0u) fb_add(fbm, slot, thr * ld3(S.bg));
            } else {
                if (sub == 0u) n_hit++;
                const d3 P = hit.p;
                int32_t mi;
                const d3 N = surface_normal_small(S, V, hit.id, P, mi);      // ray_tracer.py:77-78
                d3 Kd = mk3(0, 0, 0), Ks = mk3(0, 0, 0), Kr = mk3(0, 0, 0);
                double T = 0;
                int32_t phong = 0, mflags = 0;
                if (mi >= 0 && mi < S.n_mats) {                              // Material.py:23-62
                    const DMat M = lds_load(V.mats + mi);
                    Kd = ld3(M.kd); Ks = ld3(M.ks); Kr = ld3(M.kr);
                    T = M.T; phong = M.phong; mflags = M.flags;
                }
                const d3 Vv = neg3(d);
                const d3 Nh = N / norm3(N);                                  // ray_functions.py:98-99
                d3 diff = mk3(0, 0, 0), spec = mk3(0, 0, 0);
                for (int m = (int)sub; m < S.n_lights; m += spread ? 8 : 1) {
                    const DLight Lt = lds_load(V.lights + m);
                    const d3 dl = ld3(Lt.pos) - P;                           // Light.py:35-53
                    const d3 Ld = dl / norm3(dl);
                    double I = 1.0;                                          // Light.py:199-200
                    if (Lt.shadow != 0) {
                        d3 up, right, p00;
                        exact_basis(Lt, lp.sh_rays, P, up, right, p00);      // Light.py:252-260
                        const double gsz = Lt.radius / (double)(int)s;
                        uint32_t cnt = 0;
                        for (uint32_t kk = 0; kk < s2; ++kk) {
                            const uint32_t a = kk / s, b = kk % s;
                            d3 src = (p00 - ((double)b * gsz) * up) + ((double)a * gsz) * right;   // Light.py:273-274
                            d3 u1 = mk3(0, 0, 0), u2 = mk3(0, 0, 0);
                            if (lp.jitter) {
                                u1 = jitter3(lp.seed, gpix, path, (uint32_t)m, kk, 0u, gsz);
                                u2 = jitter3(lp.seed, gpix, path, (uint32_t)m, kk, 1u, gsz);
                            }
                            src = src + (u1 * up + u2 * right);              // Light.py:277-279
                            const d3 vec = P - src;
                            const d3 dir = vec / (norm3(vec) + RT_EPSILON);  // Light.py:283-286
                            const Hit x = nearest_hit_small(S, V, src, dir);
                            cnt += (norm3(x.p - P) < RT_EPSILON) ? 1u : 0u;  // Light.py:219-221 (NaN -> unlit)
                        }
                        const double frac = (double)cnt / (double)(int)s2;   // Light.py:223
                        I = (1.0 - Lt.shadow) + Lt.shadow * frac;            // Light.py:226
                    }
                    I = I < 0 ? 0 : (I > 1 ? 1 : I);                         // Light.py:84
                    const d3 col = ld3(Lt.color);
                    const d3 Ims = (Lt.spec * I) * col;                      // Light.py:87
                    const d3 Imd = I * col;                                  // Light.py:91
                    double ldn = dot3(Ld, N);                                // Light.py:123-125
                    ldn = isnan(ldn) ? ldn : (ldn > 0 ? ldn : 0);
                    const d3 dterm = (ldn * Kd) * Imd;
                    diff = (m == 0 || spread) ? dterm : diff + dterm;         // (spread: this lane's one light)
                    const double dp = dot3(Ld, Nh);                          // ray_functions.py:101-103
                    const d3 R = (2 * dp) * Nh - Ld;
                    const double rv = dot3(R, Vv);                           // Light.py:169
                    const double pw = ipow(rv, phong);
                    const d3 sterm = (pw * Ks) * Ims;                        // Light.py:171
                    spec = (m == 0 || spread) ? sterm : spec + sterm;
                }
                if (spread) {
                    const int g0 = lane & ~7;
                    const d3 my_d = diff, my_s = spec;
                    for (int m = 0; m < S.n_lights; ++m) {
                        const d3 dt = mk3(__shfl(my_d.x, g0 + m), __shfl(my_d.y, g0 + m), __shfl(my_d.z, g0 + m));
                        const d3 st = mk3(__shfl(my_s.x, g0 + m), __shfl(my_s.y, g0 + m), __shfl(my_s.z, g0 + m));
                        diff = (m == 0) ? dt : diff + dt;
                        spec = (m == 0) ? st : spec + st;
                    }
                }
                spec = mk3(nan_to_num(spec.x), nan_to_num(spec.y), nan_to_num(spec.z));   // Light.py:173
                const double nt = 1.0 - T;
                const d3 base = (diff + spec) * mk3(nt, nt, nt);             // ray_tracer.py:105-106
                if (GEN) {
                    const d3 v = thr * base;
                    fb[3ull * slot] = v.x; fb[3ull * slot + 1] = v.y; fb[3ull * slot + 2] = v.z;
                } else if (sub == 0u) fb_add(fb, slot, thr * base);
                if (sub != 0u) mflags = 0;                                   // (children and their background: the first lane)
                if (mflags & 1) {                                            // ray_tracer.py:108-116
                    thr_t = thr * mk3(T, T, T);
                    if (lp.last_level) fb_add(fb_children, slot, thr_t * ld3(S.bg));
                    else { want_t = true; d_t = d; o_t = P + RT_EPSILON * d; }
                }
                if (mflags & 2) {                                            // ray_tracer.py:119-129
                    thr_r = thr * Kr;
                    if (lp.last_level) fb_add(fb_children, slot, thr_r * ld3(S.bg));
                    else {
                        want_r = true;
                        const double dp = dot3(Vv, Nh);
                        d_r = (2 * dp) * Nh - Vv;                            // ray_functions.py:103
                        o_r = P + RT_EPSILON * d_r;
                    }
                }
                n_child = (want_t ? 1 : 0) + (want_r ? 1 : 0);
            }
        }
        // wave-level exclusive prefix sum of n_child -> one atomic per wave on the next queue's tail
        int incl = n_child;
        for (int off = 1; off < 64; off <<= 1) {
            int v = __shfl_up(incl, off);
            if (lane >= off) incl += v;
        }
        const int total = __shfl(incl, 63);
        if (total) {
            uint32_t base = 0;
            if (lane == 63) base = atomicAdd(&ctl->n_rays[lp.level + 1], (uint32_t)total);
            base = __shfl(base, 63);
            uint32_t w = base + (uint32_t)(incl - n_child);
            if (base + (uint32_t)total > lp.cap_next) {
                if (lane == 63) ctl->overflow = ctl->overflow_deep = 1;     // frame is invalid; host re-renders with larger queues
            } else {
                if (want_t) {
                    qn.ox[w] = o_t.x; qn.oy[w] = o_t.y; qn.oz[w] = o_t.z;
                    qn.dx[w] = d_t.x; qn.dy[w] = d_t.y; qn.dz[w] = d_t.z;
                    qn.tr[w] = thr_t.x; qn.tg[w] = thr_t.y; qn.tb[w] = thr_t.z;
                    qn.slot[w] = slot; qn.gpix[w] = gpix; qn.path[w] = child_path(path, 0u);
                    w++;
                    n_trans++;
                }
                if (want_r) {
                    qn.ox[w] = o_r.x; qn.oy[w] = o_r.y; qn.oz[w] = o_r.z;
                    qn.dx[w] = d_r.x; qn.dy[w] = d_r.y; qn.dz[w] = d_r.z;
                    qn.tr[w] = thr_r.x; qn.tg[w] = thr_r.y; qn.tb[w] = thr_r.z;
                    qn.slot[w] = slot; qn.gpix[w] = gpix; qn.path[w] = child_path(path, 1u);
                    n_refl++;
                }
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        n_refl += __shfl_down(n_refl, off);
        n_trans += __shfl_down(n_trans, off);
        n_hit += __shfl_down(n_hit, off);
        valid_total += __shfl_down(valid_total, off);
    }
    if (lane == 0) {
        if (n_refl) atomicAdd(&ctl->reflect_rays, n_refl);
        if (n_trans) atomicAdd(&ctl->transmit_rays, n_trans);
        if (n_hit) atomicAdd(&ctl->n_hits[lp.level], (uint32_t)n_hit);
        if (GEN && valid_total) atomicAdd(&ctl->primary_rays, valid_total);
    }
}

// clip(0,1) (ray_tracer.py:30) and save_image's truncating quantisation (util.py:98)
// fb0: what the level-0 rays add themselves (background of the misses, Phong term of the hits); fb: everything the
// deeper levels add.  Two buffers keep the sum independent of how the two halves of the frame are scheduled.
// fbm: the background that the rays of levels 1.. add when they miss (k_intersect).  It is kept apart from fb because
// the nearest-hit query of level L+1 runs beside the shading of level L: with one buffer the order of the float64
// additions on a pixel would depend on the schedule.
__global__ void __launch_bounds__(RT_BLOCK) k_resolve(const double *fb, const double *fb0, const double *fbm, uint64_t n_values, float *rgb, uint8_t *rgb8,
                                                      const FrameCtl *ctl, uint32_t *voided) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && ctl->overflow) atomicAdd(voided, 1u);     // rt_voided_frames
    for (uint64_t k = (uint64_t)blockIdx.x * RT_BLOCK + threadIdx.x; k < n_values; k += (uint64_t)gridDim.x * RT_BLOCK) {
        double v = (fb0[k] + fb[k]) + fbm[k];
        double c = v < 0 ? 0 : (v > 1 ? 1 : v);
        if (rgb) rgb[k] = (float)c;
        if (rgb8) rgb8[k] = isnan(c) ? (uint8_t)0 : (uint8_t)(c * 255);
    }
}

// end of a frame that writes no image: count it when a queue overflow voided it (rt_voided_frames)
__global__ void k_frame_end(const FrameCtl *ctl, uint32_t *voided) {
    if (threadIdx.x == 0 && ctl->overflow) atomicAdd(voided, 1u);
}

// mean of K clipped frames: acc (+)= clip(frame); the last frame divides and quantises like k_resolve
__global__ void __launch_bounds__(RT_BLOCK) k_accumulate(const double *fb, const double *fb0, const double *fbm, double *acc, uint64_t n_values,
                                                         int first, double frames_if_last, float *rgb, uint8_t *rgb8) {
    for (uint64_t k = (uint64_t)blockIdx.x * RT_BLOCK + threadIdx.x; k < n_values; k += (uint64_t)gridDim.x * RT_BLOCK) {
        double v = (fb0[k] + fb[k]) + fbm[k];
        double c = v < 0 ? 0 : (v > 1 ? 1 : v);
        double a = first ? c : acc[k] + c;
        acc[k] = a;
        if (frames_if_last > 0) {
            double m = a / frames_if_last;
            if (rgb) rgb[k] = (float)m;
            if (rgb8) rgb8[k] = isnan(m) ? (uint8_t)0 : (uint8_t)(m * 255);
        }
    }
}

template <bool FAST>
__global__ void __launch_bounds__(RT_BLOCK) k_intersect_only(DScene S, FastScene F, const double *o, const double *d,
                                                             uint64_t n, double *pts, int64_t *ids) {
    extern __shared__ uint2 lds_stack[];
    uint2 *stk = lds_stack + threadIdx.x;
    for (uint64_t k = (uint64_t)blockIdx.x * RT_BLOCK + threadIdx.x; k < n; k += (uint64_t)gridDim.x * RT_BLOCK) {
        Hit h = nearest_hit<FAST>(S, F, stk, ld3(o + 3 * k), ld3(d + 3 * k));
        pts[3 * k] = h.p.x; pts[3 * k + 1] = h.p.y; pts[3 * k + 2] = h.p.z;
        ids[k] = h.id;
    }
}

// The any-hit query behind Light.compute_shadows (Light.py:214-221) for an arbitrary batch: ray k is "lit" iff the
// nearest hit X of (o, d) lies within EPSILON of its target, |X - P| < 1e-6 (NaN -> unlit).  |X - P| >= | |P - o| -
// |X - o| |, so nothing farther than |P - o| + 1e-6 can be that hit, and one float64-confirmed hit closer than
// |P - o| - 3e-6 settles "unlit" whatever else lies on the ray: the walk ends there.  P need not lie on the ray.
template <bool FAST>
__device__ __forceinline__ bool ray_reaches(const DScene &S, const FastScene &F, uint2 *stk, d3 src, d3 dir, d3 P) {
    if (FAST && ray_is_tame(src, dir)) {
        const double nv = norm3(P - src), dn = norm3(dir);
        if (nv < 1e7 && dn > 0.5 && dn < 2.0) {                      // ray parameter ~ distance: shadow rays are unit length
            const float tP = __double2float_ru((nv + RT_EPSILON) / dn);
            const float tlim = fmaf(tP, RT_2M20, tP + 1.6e-5f);
            FastState st;
            fast_nearest<RT_BLOCK>(S, F, stk, src, dir, tlim, nv - 3e-6, st);
            if (st.stop) return false;
            if (!st.tie) return norm3(st.best.p - P) < RT_EPSILON;
        }
    }
    Hit x = nearest_hit_exact(S, src, dir);
    return norm3(x.p - P) < RT_EPSILON;
}

template <bool FAST>
__global__ void __launch_bounds__(RT_BLOCK) k_visibility(DScene S, FastScene F, const double *o, const double *d,
                                                         const double *tg, uint64_t n, uint8_t *lit) {
    extern __shared__ uint2 lds_stack[];
    uint2 *stk = lds_stack + threadIdx.x;
    for (uint64_t k = (uint64_t)blockIdx.x * RT_BLOCK + threadIdx.x; k < n; k += (uint64_t)gridDim.x * RT_BLOCK)
        lit[k] = ray_reaches<FAST>(S, F, stk, ld3(o + 3 * k), ld3(d + 3 * k), ld3(tg + 3 * k)) ? 1 : 0;
}

// append counters of one buffer set: [0] queue entries, [1] listed pairs, [2] overflowing pairs = 0, [3] = queue capacity
__global__ void k_init_beam_counters(uint32_t *queue_n, uint32_t queue_cap) {
    if (threadIdx.x < 4) queue_n[threadIdx.x] = threadIdx.x == 3 ? queue_cap : 0u;
}

// rec_max < 0 (ray_tracer.py:62-63): every slot is background
__global__ void __launch_bounds__(RT_BLOCK) k_fill_background(double *fb, uint64_t n_slots, double r, double g, double b) {
    for (uint64_t k = (uint64_t)blockIdx.x * RT_BLOCK + threadIdx.x; k < n_slots; k += (uint64_t)gridDim.x * RT_BLOCK) {
        fb[3 * k] = r; fb[3 * k + 1] = g; fb[3 * k + 2] = b;
    }
}

__global__ void k_deinterleave_u8(const uint8_t *packed, int64_t stride, uint8_t *image, int W, int H, int tw, int th,
                                  int tiles_x, int n_tiles, int world, int tiles_per_rank) {
    uint64_t total = (uint64_t)world * tiles_per_rank * tw * th;
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t per_rank = (uint32_t)tiles_per_rank * tw * th;
        uint32_t r = (uint32_t)(k / per_rank), loc = (uint32_t)(k % per_rank);
        uint32_t tile = r + (loc / (uint32_t)(tw * th)) * (uint32_t)world, lane = loc % (uint32_t)(tw * th);
        int x = (int)(tile % (uint32_t)tiles_x) * tw + (int)(lane % (uint32_t)tw);
        int y = (int)(tile / (uint32_t)tiles_x) * th + (int)(lane / (uint32_t)tw);
        if (tile >= (uint32_t)n_tiles || x >= W || y >= H) continue;
        const uint8_t *src = packed + (uint64_t)r * stride + 3ull * loc;
        uint8_t *dst = image + 3ull * ((uint64_t)y * W + x);
        dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
    }
}

// ---- unit kernels (kernel-level parity tests; same device functions as the frame path) --------
__global__ void k_unit_prim(int kind, const double *par, const double *o, const double *d, uint64_t n, double *out) {
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) {
        d3 oo = ld3(o + 3 * k), dd = ld3(d + 3 * k);
        d3 inv = mk3(dd.x != 0 ? 1.0 / dd.x : dinf(), dd.y != 0 ? 1.0 / dd.y : dinf(), dd.z != 0 ? 1.0 / dd.z : dinf());
        double t = 0;
        bool ok = false;
        if (kind == RT_SURF_SPHERE) ok = sphere_exact(oo, dd, dot3(dd, dd), par, par[3], t);
        else if (kind == RT_SURF_BOX) ok = slab_exact(oo, inv, par, par + 3, true, t);
        else if (kind == RT_SURF_PLANE) {
            DPlane pl;
            pl.n[0] = par[0]; pl.n[1] = par[1]; pl.n[2] = par[2]; pl.d = par[3];
            ok = plane_exact(oo, dd, pl, t);
        }
        d3 p = ok ? oo + t * dd : mk3(dnan(), dnan(), dnan());
        out[3 * k] = p.x; out[3 * k + 1] = p.y; out[3 * k + 2] = p.z;
    }
}
__global__ void k_unit_normals(int kind, const double *par, const double *pts, uint64_t n, double *out) {
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) {
        d3 P = ld3(pts + 3 * k), N = mk3(0, 0, 0);
        if (kind == RT_SURF_SPHERE) {
            d3 dir = P - ld3(par);
            N = dir / norm3(dir);
        } else if (kind == RT_SURF_BOX) N = box_normal(P, ld3(par + 6), par[9]);
        else if (kind == RT_SURF_PLANE) N = ld3(par);
        out[3 * k] = N.x; out[3 * k + 1] = N.y; out[3 * k + 2] = N.z;
    }
}
__global__ void k_unit_slab(const double *lo, const double *hi, int outside_only, const double *o, const double *d,
                            uint64_t n, uint8_t *mask, double *tn) {
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) {
        d3 oo = ld3(o + 3 * k), dd = ld3(d + 3 * k);
        d3 inv = mk3(dd.x != 0 ? 1.0 / dd.x : dinf(), dd.y != 0 ? 1.0 / dd.y : dinf(), dd.z != 0 ? 1.0 / dd.z : dinf());
        double t;
        mask[k] = slab_exact(oo, inv, lo, hi, outside_only != 0, t) ? 1 : 0;
        tn[k] = t;
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_create_error;

// Everything a frame's kernels write apart from the caller's outputs: ray queues, hit records, per-pair soft-shadow state,
// the three framebuffers and the counter block.  The allocating entry points carve it out of hipMalloc'ed buffers
// (ensure_frame); rt_render_device_noalloc carves it out of the library's static device array (see `scratch` below) and
// swaps it in for the duration of the enqueue.  Trivially copyable on purpose.
struct FrameState {
    uint64_t cap = 0;       // ray-queue capacity of level 0 (and of every level when the levels share one buffer set)
    uint64_t cap_deep = 0;  // ... of the levels >= 1: queues 1.., buffer sets 1.. (cap / deep_div)
    uint64_t fb_slots = 0;
    uint32_t beam_queue_cap = 0;
    uint32_t tile_nblk = 0, tile_nreg = 0, tile_nzone = 0;      // what the binning buffers of the tile beams were carved for
    double qgrowth_carved = 1.0;    // float64 shadow-ray queue of a set = its capacity * this
    double rec_div_carved = 1.0;    // pair records of a set = its (hit, light) pairs / this (+ the waves' block slack)
    uint32_t spb_carved = 1;        // lit masks for chunks of at least this many pairs (RT_BLOCK / (s * s))
    RayQ q[6]{};            // level 0 -> q[0]; level k >= 1 -> q[1 + (k - 1) % 2] (% 3 / % 5 when the deeper levels are pipelined over 2 / 4 sets)
    HitBuf hb{}, hb1{}, hb2{}, hb3{}, hb4{};   // level 0 / deeper levels, rotating (so are seg*, beam*)
    BeamBuf beam{}, beam1{}, beam2{}, beam3{}, beam4{};
    bool five_sets = false;                 // hb3, hb4 / .. are separate allocations: a level's buffers are reused four levels later
    double *fb = nullptr, *fb0 = nullptr, *fbm = nullptr;   // see k_resolve
    bool two_sets = false;                  // hb1 / seg1 / beam1 are separate allocations
    bool three_sets = false;                // hb2 / seg2 / beam2 are separate allocations: levels 1.. can be pipelined
    FrameCtl *ctl = nullptr;
};

static HitBuf *hit_set(FrameState &fs, int set) { HitBuf *a[5] = {&fs.hb, &fs.hb1, &fs.hb2, &fs.hb3, &fs.hb4}; return a[set]; }
static BeamBuf *beam_set(FrameState &fs, int set) { BeamBuf *a[5] = {&fs.beam, &fs.beam1, &fs.beam2, &fs.beam3, &fs.beam4}; return a[set]; }

// RT_AMD_* environment switches, read ONCE in rt_create (a frame takes ~2 ms, a getenv per launch decision is host time on its
// critical path).  What is left are the switches the tests and tools use to force a path the defaults choose by scene or frame:
//   RT_AMD_DEBUG            per-level diagnostics on stderr
//   RT_AMD_SETS             buffer sets (1, 2, 3 or 5): fewer sets -> levels share sets and run in sequence
//   RT_AMD_NO_PIPELINE      deeper levels on one stream
//   RT_AMD_WALK_MIN         k_shadow_walk takes a level's rays of overflowing pairs from this many on (1 = always, -1 = never)
//   RT_AMD_BEAM_EARLY       0 | 1: beam walk near side first with the umbra test (default: scenes of >= 100 k primitives)
//   RT_AMD_TILES_MIN(_PRIMS) tile beams for frames of at least this many camera rays / scenes of at least this many primitives
//   RT_AMD_CREATE_THREADS   host threads of rt_create's table loops;  RT_AMD_PARSE_THREADS (rt_scene.cpp)
//   RT_AMD_NO_EARLY_CAM     the camera pass of a frame waits for the end of the frame before it (AltL0 below)
struct EnvCfg {
    long long walk_min = 3ll << 20;
    int beam_early = -1;
    bool no_pipeline = false;
    long long tiles_min = 1500000;      // (the three small kernels in front of k_beam_setup cost more than they save on smaller frames)
    long long tiles_min_early = 400000; // ... unless the frame's camera pass ran early (AltL0): the other side of a small frame is shorter then, and the
                                        // level-0 side decides - frames back to back, 400 k against 1.5 M: an eighth of 3840x2160 0.965 -> 0.933 ms,
                                        // 1280x720 1.00 -> 0.92, 960x540 0.70 -> 0.67, 640x360 0.535 -> 0.536 (100 k: 0.563)
    int tiles_min_prims = 256;          // (tile_scene_ok)
    bool debug = false;
    bool no_early_cam = false;
    void read() {
        no_early_cam = getenv("RT_AMD_NO_EARLY_CAM") != nullptr;
        if (const char *e = getenv("RT_AMD_WALK_MIN")) walk_min = atoll(e);
        if (const char *e = getenv("RT_AMD_BEAM_EARLY")) beam_early = atoi(e) != 0;
        no_pipeline = getenv("RT_AMD_NO_PIPELINE") != nullptr;
        if (const char *v = getenv("RT_AMD_TILES_MIN")) tiles_min = tiles_min_early = atoll(v);
        if (const char *v = getenv("RT_AMD_TILES_MIN_PRIMS")) tiles_min_prims = atoi(v);
        debug = getenv("RT_AMD_DEBUG") != nullptr;
    }
};
// What a camera pass writes, a second time.  Frames enqueued back to back on one stream: the head of a frame is its camera pass -
// a quarter of a millisecond with nothing beside it, everything else of the frame waits for its hits - and the tail is the float64
// queue, the Phong kernel and the resolve of level 0, with a GPU that the deeper levels have already left.  With its own copy of
// the level-0 state the camera pass of frame N + 1 runs beside the tail of frame N (on stream2, behind frame N's nearest-hit
// chain); two copies alternate.  Everything a camera pass stores to is here: the camera rays' queue entries (origin and throughput
// of camera rays are never stored: shared), the hit records and hit list of level 0 (the lit counters belong to the shadow
// kernels: shared), the three framebuffers, the counter block, the tile beams' block boxes and zone counters, the append counters
// of level 0's soft-shadow kernels.  The children go to q[1]: the pass waits for the last level of frame N that reads it.
#define RT_EARLY_CAM_MAX_SLOTS 2200000ull
struct AltL0 {
    RayQ q0{};
    HitBuf hb{};
    double *fb = nullptr, *fb0 = nullptr, *fbm = nullptr;
    FrameCtl *ctl = nullptr;
    float *blockbox = nullptr;
    uint32_t *zcnt = nullptr, *queue_n = nullptr;
    bool ok = false, failed = false;
};

// workgroups per CU of the two sides of an overlapped frame (of a CU's four 128-register slots): level-0 side / deeper levels
#define RT_OVL_MAIN 3
#define RT_OVL_SIDE 1

struct rt_ctx : FrameState {
    int device = 0;
    EnvCfg env;
    int64_t early_frames = 0;               // frames enqueued with an early camera pass (rt_debug_early_frames)
    int64_t tile_frames = 0;                // frames rendered with tile beams whose counters were fetched and that no queue overflow voided (rt_debug_tile_frames)
    bool last_tiles = false;                // the last frame enqueued took the tile beams
    unsigned char *stage = nullptr;         // grow-only device staging for the blocking host-buffer entry points (no hipMalloc per call)
    size_t stage_bytes = 0;
    uint32_t *d_voided = nullptr;           // frames voided by a queue overflow since the last rt_voided_frames(reset) (device counter)
    std::vector<hipEvent_t> prof_bs;        // RT_FLAG_PROFILE: [start, end] of k_beam_setup per level
    int prof_bs_levels = 0;
    hipStream_t stream = nullptr;
    std::string error;
    std::vector<void *> scene_allocs;
    DScene S{};
    FastScene F{};
    size_t lds_bytes = 0;   // per-lane traversal stacks: F.depth * RT_BLOCK * 8
    size_t lds_top_bytes = 0;   // + the table of the tree's first levels, for the kernels that stage it (rt_top_stage)
    SmallLayout small{};    // scene tables staged in LDS by k_small_level (rt_small.hpp); small_ok: they fit
    bool small_ok = false;
    int g_small = 2048;
    rt_scene_desc host{};   // scalar fields only (camera, settings)
    uint64_t fb_slots_frame = 0;   // primary slots of the frame being enqueued
    double growth = 2.0;    // queue capacity = growth * primary slots
    double deep_div = 2.0;  // cap_deep = cap / deep_div: a level >= 1 rarely holds more rays than there are camera rays (halved when one does)
    double qgrowth = 1.0;   // float64 shadow-ray queue = qgrowth * queue capacity (grows by itself: large s, many undecided rays)
    double rec_div = 5.0;   // pair records for 1 / rec_div of a buffer set's (hit, light) pairs (halved when a level runs out: RT_ERR_QUEUE_OVERFLOW)
    std::vector<void *> frame_allocs;
    uint64_t frame_bytes = 0;               // device bytes of the frame state (ensure_frame)
    int env_sets = 0;                       // RT_AMD_SETS / RT_AMD_ONE_SET, read once in rt_create (0 = no limit)
    FrameCtl *ctl_own = nullptr;            // the context's own counter block (FrameState::ctl points here outside a no-alloc frame)
    hipStream_t stream3 = nullptr;          // soft shadows + Phong of the odd levels 1, 3, .. in a pipelined frame
    hipStream_t stream4 = nullptr;          // ... of the even levels 2, 4, ..
    std::vector<hipEvent_t> ev_hit, ev_lit; // per level: nearest hits known / shading added
    double *acc = nullptr;                  // rt_render_mean: sum of the clipped frames
    uint64_t acc_slots = 0;
    hipStream_t stream2 = nullptr;          // the deeper levels of an overlapped frame
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    AltL0 alt;                              // the other copy of the level-0 state (early camera pass)
    std::vector<void *> alt_allocs;
    uint64_t alt_bytes = 0;
    int alt_cur = 0;                        // which of the two copies the context's own pointers name
    hipEvent_t ev_cam = nullptr, ev_prev = nullptr, ev_end[2] = {nullptr, nullptr};   // camera pass done / stream order handed to stream2 / end of the last frame on copy 0, 1
    bool ev_end_set[2] = {false, false};
    bool early_prev = false;                // the last use of the frame state was a pipelined camera frame (its levels recorded ev_lit) ...
    hipStream_t early_st = nullptr;         // ... on this stream,
    int early_prev_rec = 0, early_prev_q1 = 0;   // ... with these levels: the last one / the last one that reads q[1]
    int nb_intersect = 1, nb_beam_setup = 1, nb_beam = 1, nb_queue = 1, nb_shade = 1, nb_setup = 1, nb_shadow = 1, nb_walk = 1;   // workgroups per CU
    int g_walk = 2048;
    int g_beam_setup_t = 2048, nb_beam_setup_t = 1;
    FrameCtl *h_ctl = nullptr;   // pinned
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<hipEvent_t> prof_ev;     // RT_FLAG_PROFILE: [gen0, gen1, (i0,i1,s0,s1,h0,h1) per level ..., res0, res1]
    int prof_levels = -1;                // levels bracketed in the last frame, -1 = not profiled
    bool frame_pending = false;
    bool counters_dirty = false;         // a voided frame left lit counts behind: clear them before the next frame
    bool last_noalloc = false;           // the last frame came through rt_render_device_noalloc: its queues cannot grow
    rt_render_params last{};
    int32_t last_sh = 0, last_rec = 0;
    uint64_t last_slots = 0;
    bool last_gen = false;      // the last frame enqueued was a camera frame (rt_render_device), not an rt_trace batch
    int grid = 2048;
    int n_cu = 256;
    int n_boxes = 0;
    int g_intersect = 2048, g_intersect_x = 2048, g_shadow = 2048, g_shadow_x = 2048, g_beam_setup = 2048, g_beam = 2048,
        g_queue = 2048, g_shade = 2048, g_setup = 2048;
};

static int fail(rt_ctx *ctx, int code, const std::string &msg) {
    if (ctx) ctx->error = msg;
    else g_create_error = msg;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                                   \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) {                                                                              \
            int code_ = (e_ == hipErrorOutOfMemory) ? RT_ERR_OOM : RT_ERR_HIP;                               \
            return fail(ctx, code_, std::string(#expr) + ": " + hipGetErrorString(e_));                      \
        }                                                                                                    \
    } while (0)

// every allocating HIP call of this library goes through these (rt_debug_alloc_calls counts them: the no-alloc entry point's
// test asserts that a call leaves the count unchanged)
static std::atomic<long long> g_alloc_calls(0);
template <typename T>
static hipError_t rt_malloc(T **p, size_t bytes) { g_alloc_calls++; return hipMalloc((void **)p, bytes); }
static hipError_t rt_host_malloc(void **p, size_t bytes) { g_alloc_calls++; return hipHostMalloc(p, bytes, hipHostMallocDefault); }
static hipError_t rt_event_create(hipEvent_t *e, bool timing) { g_alloc_calls++; return timing ? hipEventCreate(e) : hipEventCreateWithFlags(e, hipEventDisableTiming); }

// The no-alloc entry point's memory (rt_render_device_noalloc): every buffer ensure_frame would hipMalloc for a small frame is
// carved out of this array instead.  In host code the NAME `scratch` is only a stand-in for the symbol: the device address comes
// from hipGetSymbolAddress, once per device, and only that pointer is ever handed to kernels, memsets or copies.
// Each GPU the library runs on holds its own copy (RT_SCRATCH_BYTES of HBM) until the process ends.
#define RT_SCRATCH_BYTES (64ull << 20)
#define RT_MAX_DEVICES 64
static __device__ __align__(256) unsigned char scratch[RT_SCRATCH_BYTES];
static unsigned char *g_scratch_ptr[RT_MAX_DEVICES];      // per device ordinal; nullptr = not looked up yet
static std::mutex g_scratch_mutex;

// device address of `scratch` on `device` (which must be the current device)
static int scratch_address(rt_ctx *ctx, int device, unsigned char **out) {
    if (device < 0 || device >= RT_MAX_DEVICES) { if (out) *out = nullptr; return RT_OK; }     // no no-alloc frames on such a device
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void *p = nullptr;
        HIP_TRY(ctx, hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p || ((uintptr_t)p & 255u)) return fail(ctx, RT_ERR_HIP, "hipGetSymbolAddress(scratch) gave a null or misaligned address");
        g_scratch_ptr[device] = (unsigned char *)p;
    }
    if (out) *out = g_scratch_ptr[device];
    return RT_OK;
}

extern "C" int64_t rt_debug_alloc_calls(void) { return (int64_t)g_alloc_calls.load(); }
#ifdef RT_TRAV_STATS
// diagnostics build only: copy out (and clear) the traversal counters, 8 per level
extern "C" int rt_debug_trav_stats(unsigned long long *out, int n) {
    if (hipDeviceSynchronize() != hipSuccess) return RT_ERR_HIP;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trav_stats), (size_t)n * sizeof(unsigned long long)) != hipSuccess) return RT_ERR_HIP;
    static unsigned long long zero[RT_MAX_LEVELS * 8];
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_trav_stats), zero, sizeof(zero)) != hipSuccess) return RT_ERR_HIP;
    return RT_OK;
}
extern "C" int rt_debug_wave_log(unsigned long long *out, uint32_t *counts) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_log), sizeof(unsigned long long) * 16 * 4096 * 6) != hipSuccess) return RT_ERR_HIP;
    if (hipMemcpyFromSymbol(counts, HIP_SYMBOL(g_wave_n), sizeof(uint32_t) * 16) != hipSuccess) return RT_ERR_HIP;
    static const uint32_t zero[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_wave_n), zero, sizeof(zero)) != hipSuccess) return RT_ERR_HIP;
    return RT_OK;
}
extern "C" int rt_debug_trav_time(unsigned long long *out, int n) {
    if (hipDeviceSynchronize() != hipSuccess) return RT_ERR_HIP;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trav_time), (size_t)n * sizeof(unsigned long long)) != hipSuccess) return RT_ERR_HIP;
    static unsigned long long zero[RT_MAX_LEVELS * 8];
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_trav_time), zero, sizeof(zero)) != hipSuccess) return RT_ERR_HIP;
    return RT_OK;
}
#endif
extern "C" int64_t rt_debug_tile_frames(const rt_ctx *ctx) { return ctx ? ctx->tile_frames : 0; }
extern "C" int64_t rt_debug_early_frames(const rt_ctx *ctx) { return ctx ? ctx->early_frames : 0; }
extern "C" int64_t rt_noalloc_scratch_bytes(void) { return (int64_t)RT_SCRATCH_BYTES; }

// host loops of rt_create over millions of independent records: cut into contiguous pieces, one thread each
// (nothing below 64 k records; at most 16 threads, the GPU box's CPU share)
template <typename Fn>
static void parallel_for(size_t n, Fn fn) {
    unsigned hw = std::thread::hardware_concurrency();
    size_t nt = std::min<size_t>(std::min<size_t>(hw ? hw : 1, 16), n / 65536);
    if (const char *e = getenv("RT_AMD_CREATE_THREADS")) nt = std::min<size_t>((size_t)std::max(atoi(e), 1), n ? n : 1);
    if (nt <= 1) { fn((size_t)0, n); return; }
    std::vector<std::thread> th;
    const size_t step = (n + nt - 1) / nt;
    for (size_t t = 0; t < nt; ++t) {
        const size_t b = t * step, e = std::min(n, b + step);
        if (b < e) th.emplace_back([=]() { fn(b, e); });
    }
    for (auto &x : th) x.join();
}

template <typename T>
static int upload(rt_ctx *ctx, std::vector<void *> &owner, const std::vector<T> &h, const T **out) {
    void *p = nullptr;
    size_t bytes = (h.size() ? h.size() : 1) * sizeof(T);
    HIP_TRY(ctx, rt_malloc(&p, bytes));
    owner.push_back(p);
    if (!h.empty()) HIP_TRY(ctx, hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = (const T *)p;
    return RT_OK;
}

static float round_down(double v) {
    float f = (float)v;
    if ((double)f > v) f = nextafterf(f, -INFINITY);
    return f;
}
static float round_up(double v) {
    float f = (float)v;
    if ((double)f < v) f = nextafterf(f, INFINITY);
    return f;
}

static int check_scene(const rt_scene_desc *s, std::string &why) {
#define NEED(cond, text) if (!(cond)) { why = text; return RT_ERR_INVALID_ARG; }
    NEED(s, "scene is NULL");
    NEED(s->abi_version == RT_AMD_ABI_VERSION, "rt_scene_desc.abi_version mismatch");
    NEED(s->n_surfaces >= 0 && s->n_spheres >= 0 && s->n_boxes >= 0 && s->n_planes >= 0 && s->n_materials >= 0 &&
         s->n_lights >= 0 && s->n_nodes >= 1 && s->n_leaf_refs >= 0, "negative count (or no BSP node)");
    NEED(s->n_surfaces == 0 || (s->surf_kind && s->surf_geom && s->surf_mat), "surface table pointer is NULL");
    NEED(s->n_spheres == 0 || (s->sph_center && s->sph_radius && s->sph_r2 && s->sph_id), "sphere table pointer is NULL");
    NEED(s->n_boxes == 0 || (s->box_center && s->box_scale && s->box_lo && s->box_hi && s->box_id), "box table pointer is NULL");
    NEED(s->n_planes == 0 || (s->pln_normal && s->pln_d && s->pln_id), "plane table pointer is NULL");
    NEED(s->n_materials == 0 || (s->mat_kd && s->mat_ks && s->mat_kr && s->mat_phong && s->mat_transparency), "material table pointer is NULL");
    NEED(s->n_lights == 0 || (s->lgt_pos && s->lgt_color && s->lgt_spec && s->lgt_shadow && s->lgt_radius), "light table pointer is NULL");
    NEED(s->node_lo && s->node_hi && s->node_right && s->node_first && s->node_count, "node table pointer is NULL");
    NEED(s->n_leaf_refs == 0 || s->leaf_prim, "leaf_prim is NULL");
    const int np = s->n_spheres + s->n_boxes;
    for (int i = 0; i < s->n_leaf_refs; ++i) NEED(s->leaf_prim[i] >= 0 && s->leaf_prim[i] < np, "leaf_prim entry out of range");
    for (int i = 0; i < s->n_nodes; ++i) {
        if (s->node_right[i] >= 0) {
            NEED(s->node_right[i] > i + 1 && s->node_right[i] < s->n_nodes && i + 1 < s->n_nodes, "node_right out of range / not pre-order");
        } else {
            NEED(s->node_first[i] >= 0 && s->node_count[i] >= 0 && s->node_first[i] + s->node_count[i] <= s->n_leaf_refs, "leaf range out of bounds");
        }
    }
    for (int i = 0; i < s->n_surfaces; ++i) {
        int k = s->surf_kind[i], g = s->surf_geom[i];
        int lim = k == RT_SURF_SPHERE ? s->n_spheres : k == RT_SURF_BOX ? s->n_boxes : k == RT_SURF_PLANE ? s->n_planes : 0;
        NEED(k >= 0 && k <= 3, "surf_kind out of range");
        NEED(k == RT_SURF_BACKGROUND || (g >= 0 && g < lim), "surf_geom out of range");
    }
#undef NEED
    return RT_OK;
}

extern "C" int rt_abi_version(void) { return RT_AMD_ABI_VERSION; }

extern "C" const char *rt_last_error(const rt_ctx *ctx) { return ctx ? ctx->error.c_str() : g_create_error.c_str(); }

extern "C" int rt_create(const rt_scene_desc *s, int device, rt_ctx **out) {
    if (!out) return fail(nullptr, RT_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    std::string why;
    const bool trace_create = getenv("RT_AMD_DEBUG") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto tick = [&](const char *what) {       // RT_AMD_DEBUG: where the set-up time of a large scene goes
        if (!trace_create) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[rt_amd] rt_create %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    if (int rc = check_scene(s, why)) return fail(nullptr, rc, why);
    tick("check_scene");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(nullptr, RT_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= ndev) return fail(nullptr, RT_ERR_NO_DEVICE, "device ordinal out of range");
    rt_ctx *ctx = new rt_ctx();
    ctx->device = device;
    auto bail = [&](int rc) {
        g_create_error = ctx->error;
        rt_destroy(ctx);
        return rc;
    };
#define TRY_C(expr) do { int rc_ = (expr); if (rc_ != RT_OK) return bail(rc_); } while (0)
#define HIP_C(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { ctx->error = std::string(#expr) + ": " + hipGetErrorString(e_); return bail(e_ == hipErrorOutOfMemory ? RT_ERR_OOM : RT_ERR_HIP); } } while (0)
    HIP_C(hipSetDevice(device));
    HIP_C(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    HIP_C(rt_event_create(&ctx->ev0, true));
    HIP_C(rt_event_create(&ctx->ev1, true));
    // Plain streams, all of one priority.  (Round 1 tried a high-priority stream for the deeper levels; a run with it hung and no
    // log was kept.  DESIGN.md section 8, "The recorded hang", goes through what the schedule itself can and cannot do - it holds
    // no wait that a kernel of another stream must resolve - and why the experiment is not repeated; s_setprio in the kernels
    // gives the same bias without a second queue priority.)
    HIP_C(hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
    HIP_C(hipStreamCreateWithFlags(&ctx->stream3, hipStreamNonBlocking));
    HIP_C(rt_event_create(&ctx->ev_fork, false));
    HIP_C(rt_event_create(&ctx->ev_join, false));
    HIP_C(rt_event_create(&ctx->ev_cam, false));
    HIP_C(rt_event_create(&ctx->ev_prev, false));
    HIP_C(rt_event_create(&ctx->ev_end[0], false));
    HIP_C(rt_event_create(&ctx->ev_end[1], false));
    ctx->env.read();
    HIP_C(rt_malloc(&ctx->d_voided, sizeof(uint32_t)));
    HIP_C(hipMemset(ctx->d_voided, 0, sizeof(uint32_t)));
    HIP_C(rt_malloc(&ctx->ctl_own, sizeof(FrameCtl)));
    ctx->ctl = ctx->ctl_own;
    if (int rc = scratch_address(ctx, device, nullptr)) return bail(rc);      // the no-alloc entry point's buffer on this device, looked up once
    HIP_C(rt_host_malloc((void **)&ctx->h_ctl, sizeof(FrameCtl)));
    hipDeviceProp_t prop;
    HIP_C(hipGetDeviceProperties(&prop, device));
    ctx->n_cu = prop.multiProcessorCount;
    ctx->grid = prop.multiProcessorCount * 8;
    ctx->grid = (ctx->grid + 7) & ~7;

    tick("context, stream, events");
    // ---- primitives: spheres first, then boxes ----
    const int ns = s->n_spheres, nb = s->n_boxes;
    ctx->n_boxes = nb;
    std::vector<DPrim> prims((size_t)ns + nb);
    std::vector<FPrim> fprims((size_t)ns + nb);
    parallel_for((size_t)ns, [&](size_t b0, size_t e0) {
        for (size_t i = b0; i < e0; ++i) {
            DPrim &p = prims[i];
            memset(&p, 0, sizeof(p));
            for (int k = 0; k < 3; ++k) p.a[k] = s->sph_center[3 * i + k];
            p.b[0] = s->sph_r2[i]; p.b[1] = s->sph_radius[i]; p.b[2] = 0;
            p.kind = 1; p.id = s->sph_id[i];
            FPrim &f = fprims[i];
            memset(&f, 0, sizeof(f));
            for (int k = 0; k < 3; ++k) f.a[k] = (float)p.a[k];
            f.r = round_up(fabs(s->sph_radius[i]));
            f.kind = 1;
        }
    });
    parallel_for((size_t)nb, [&](size_t b0, size_t e0) {
        for (size_t i = b0; i < e0; ++i) {
            DPrim &p = prims[(size_t)ns + i];
            memset(&p, 0, sizeof(p));
            for (int k = 0; k < 3; ++k) { p.a[k] = s->box_lo[3 * i + k]; p.b[k] = s->box_hi[3 * i + k]; }
            p.kind = 2; p.id = s->box_id[i];
            FPrim &f = fprims[(size_t)ns + i];
            memset(&f, 0, sizeof(f));
            for (int k = 0; k < 3; ++k) { f.a[k] = round_down(p.a[k]); f.b[k] = round_up(p.b[k]); }
            f.kind = 2;
        }
    });
    TRY_C(upload(ctx, ctx->scene_allocs, prims, &ctx->S.prims));
    TRY_C(upload(ctx, ctx->scene_allocs, fprims, &ctx->S.fprims));

    tick("primitive tables + upload");
    // ---- nodes, with skip links ----
    const int nn = s->n_nodes;
    std::vector<DNode> nodes(nn);
    std::vector<FNode> fnodes(nn);
    std::vector<int32_t> skip(nn, 0);
    {
        // subtree end of node i in pre-order: leaf -> i+1, internal -> end(right child); right children
        // come later in the array, so a reverse sweep resolves them first
        for (int i = nn - 1; i >= 0; --i) skip[i] = s->node_right[i] < 0 ? i + 1 : skip[s->node_right[i]];
    }
    parallel_for((size_t)nn, [&](size_t b0, size_t e0) {
        for (size_t i = b0; i < e0; ++i) {
            DNode &n = nodes[i];
            for (int k = 0; k < 3; ++k) { n.lo[k] = s->node_lo[3 * i + k]; n.hi[k] = s->node_hi[3 * i + k]; }
            n.skip = skip[i];
            bool leaf = s->node_right[i] < 0;
            n.first = leaf ? s->node_first[i] : -1;
            n.count = leaf ? s->node_count[i] : 0;
            n.pad = 0;
            FNode &f = fnodes[i];
            for (int k = 0; k < 3; ++k) { f.lo[k] = round_down(n.lo[k]); f.hi[k] = round_up(n.hi[k]); }
            f.skip = skip[i];
            f.first_count = !leaf ? -1 : ((n.first < (1 << 22) && n.count < 256) ? ((n.first << 8) | n.count) : -2);
        }
    });
    TRY_C(upload(ctx, ctx->scene_allocs, nodes, &ctx->S.nodes));
    TRY_C(upload(ctx, ctx->scene_allocs, fnodes, &ctx->S.fnodes));
    std::vector<int32_t> leaf(s->leaf_prim, s->leaf_prim + s->n_leaf_refs);
    TRY_C(upload(ctx, ctx->scene_allocs, leaf, &ctx->S.leaf_prim));

    tick("node tables + upload");
    // ---- kd view of the same tree for the fast path (rt_fast.hpp) ----
    {
        std::vector<KNode> kn(nn);
        std::vector<LeafF> lf((size_t)s->n_leaf_refs);
        std::vector<uint32_t> prim_entry((size_t)ns + nb + 1, 0u);
        bool ok = true;
        std::vector<int> depth(nn, 0);
        int max_depth = 1;
        for (int i = 0; i < nn && ok; ++i) {
            const double *lo = s->node_lo + 3 * i, *hi = s->node_hi + 3 * i;
            if (s->node_right[i] < 0) {
                if ((unsigned)s->node_count[i] >= (1u << 30)) ok = false;
                kn[i].a = (uint32_t)s->node_first[i];
                kn[i].b = ((uint32_t)s->node_count[i] << 2) | 3u;
                continue;
            }
            const int L = i + 1, R = s->node_right[i];
            depth[L] = depth[R] = depth[i] + 1;
            if (depth[i] + 2 > max_depth) max_depth = depth[i] + 2;
            const double *llo = s->node_lo + 3 * L, *lhi = s->node_hi + 3 * L, *rlo = s->node_lo + 3 * R, *rhi = s->node_hi + 3 * R;
            int axis = -1;
            for (int k = 0; k < 3; ++k) if (lhi[k] != hi[k] || rlo[k] != lo[k]) { if (axis >= 0) ok = false; else axis = k; }
            if (axis < 0) {   // degenerate split (cut == bound): both children equal the parent box
                axis = 0;
                double ext = hi[0] - lo[0];
                for (int k = 1; k < 3; ++k) if (hi[k] - lo[k] > ext) { ext = hi[k] - lo[k]; axis = k; }
            }
            for (int k = 0; k < 3 && ok; ++k) {
                if (llo[k] != lo[k] || rhi[k] != hi[k]) ok = false;               // children share the outer faces
                if (k != axis && (lhi[k] != hi[k] || rlo[k] != lo[k])) ok = false;
            }
            const double cut = lhi[axis];
            if (!(rlo[axis] == cut) || !(lo[axis] <= cut && cut <= hi[axis])) ok = false; // and tile the parent
            const float cf = (float)cut;
            memcpy(&kn[i].a, &cf, 4);
            kn[i].b = ((uint32_t)R << 2) | (uint32_t)axis;
        }
        if (max_depth > 64) ok = false;   // per-lane stacks (8 B per level and lane) must fit the 160 KB LDS
        for (int e = s->n_leaf_refs - 1; e >= 0; --e) {           // (descending: prim_entry ends up with the first entry)
            const int p = s->leaf_prim[e];
            if (p >= 0 && p < ns + nb) prim_entry[p] = (uint32_t)e;
        }
        std::atomic<bool> r2_ok(true);
        parallel_for((size_t)s->n_leaf_refs, [&](size_t b0, size_t e0) {
            for (size_t e = b0; e < e0; ++e) {
                const int p = s->leaf_prim[e];
                LeafF &f = lf[e];
                memset(&f, 0, sizeof(f));
                if (p < ns) {
                    const double *c = s->sph_center + 3 * (size_t)p;
                    f.x = (float)c[0]; f.y = (float)c[1]; f.z = (float)c[2];
                    f.lx = (float)(c[0] - (double)f.x); f.ly = (float)(c[1] - (double)f.y); f.lz = (float)(c[2] - (double)f.z);
                    f.r = round_up(fabs(s->sph_radius[p]));
                    f.pk = ((uint32_t)p << 2) | 1u;
                    if (!(s->sph_r2[p] <= (double)f.r * (double)f.r * 1.0000001)) r2_ok = false;   // r2 must be radius**2
                } else {
                    const double *blo = s->box_lo + 3 * (size_t)(p - ns), *bhi = s->box_hi + 3 * (size_t)(p - ns);
                    f.x = round_down(blo[0]); f.y = round_down(blo[1]); f.z = round_down(blo[2]);
                    f.lx = round_up(bhi[0]); f.ly = round_up(bhi[1]); f.lz = round_up(bhi[2]);
                    f.pk = ((uint32_t)p << 2) | 2u;
                }
            }
        });
        if (!r2_ok) ok = false;
        if ((size_t)ns + nb >= (1u << 30)) ok = false;
        for (int k = 0; k < 3; ++k) {
            std::vector<float> cuts;
            cuts.push_back((float)s->node_lo[k]);
            cuts.push_back((float)s->node_hi[k]);
            for (int i = 0; i < nn; ++i)
                if (s->node_right[i] >= 0) {          // the children's shared bound on the split axis; their other bounds are the parent's
                    const double *lhi = s->node_hi + 3 * (size_t)(i + 1), *rlo = s->node_lo + 3 * (size_t)s->node_right[i];
                    if (lhi[k] != s->node_hi[3 * (size_t)i + k]) cuts.push_back((float)lhi[k]);
                    if (rlo[k] != s->node_lo[3 * (size_t)i + k]) cuts.push_back((float)rlo[k]);
                }
            std::sort(cuts.begin(), cuts.end());
            cuts.erase(std::unique(cuts.begin(), cuts.end()), cuts.end());
            ctx->F.n_cuts[k] = (int32_t)cuts.size();
            TRY_C(upload(ctx, ctx->scene_allocs, cuts, &ctx->F.cuts[k]));
        }
        {
            // first levels, breadth first (rt_fast.hpp: FastScene::top)
            const int levels = 8;      // (255 nodes, 4 KB: the camera-ray pass keeps its four workgroups per CU)
            std::vector<uint4> top;
            if (ok && levels > 0 && s->node_right[0] >= 0) {
                std::vector<int> order, lvl, pos(nn, -1);      // internal nodes of depth < levels, breadth first
                order.push_back(0); lvl.push_back(0); pos[0] = 0;
                for (size_t h = 0; h < order.size(); ++h) {
                    const int i = order[h];
                    const int ch[2] = {i + 1, s->node_right[i]};
                    for (int c = 0; c < 2; ++c)
                        if (s->node_right[ch[c]] >= 0 && lvl[h] + 1 < levels && (int)order.size() < RT_TOP_MAX_NODES) {
                            pos[ch[c]] = (int)order.size();
                            order.push_back(ch[c]); lvl.push_back(lvl[h] + 1);
                        }
                }
                top.resize(order.size());
                for (size_t h = 0; h < order.size(); ++h) {
                    const int i = order[h], L = i + 1, R = s->node_right[i];
                    uint4 t;
                    t.x = kn[i].a; t.y = kn[i].b & 3u;
                    t.z = pos[L] >= 0 ? (RT_TOP_FLAG | (uint32_t)pos[L]) : (uint32_t)L;
                    t.w = pos[R] >= 0 ? (RT_TOP_FLAG | (uint32_t)pos[R]) : (uint32_t)R;
                    top[h] = t;
                }
            }
            ctx->F.n_top = (int32_t)top.size();
            TRY_C(upload(ctx, ctx->scene_allocs, top, &ctx->F.top));
            ctx->lds_top_bytes = top.empty() ? 0 : (size_t)RT_TOP_MAX_NODES * sizeof(uint4);
        }
        TRY_C(upload(ctx, ctx->scene_allocs, kn, &ctx->F.knodes));
        TRY_C(upload(ctx, ctx->scene_allocs, lf, &ctx->F.leaves));
        TRY_C(upload(ctx, ctx->scene_allocs, prim_entry, &ctx->F.prim_entry));
        {
            // leaf entries per primitive, ascending, with their leaf's node (FastScene::tie_off); only when the leaf list follows the
            // nodes' pre-order, which is what "smaller entry = visited earlier" rests on
            bool ordered = ok;
            long long prev = -1;
            for (int i = 0; i < nn && ordered; ++i)
                if (s->node_right[i] < 0 && s->node_count[i] > 0) {
                    if ((long long)s->node_first[i] <= prev) ordered = false;
                    prev = (long long)s->node_first[i] + s->node_count[i] - 1;
                }
            ctx->F.tie_off = nullptr; ctx->F.tie_ent = nullptr;
            if (ordered && ns + nb > 0) {
                const size_t np = (size_t)ns + nb;
                std::vector<uint32_t> off(np + 1, 0u);
                for (int e = 0; e < s->n_leaf_refs; ++e) { const int p = s->leaf_prim[e]; if (p >= 0 && (size_t)p < np) ++off[p + 1]; }
                for (size_t p = 0; p < np; ++p) off[p + 1] += off[p];
                std::vector<uint2> ent(off[np]);
                std::vector<uint32_t> fill(off.begin(), off.end() - 1);
                for (int i = 0; i < nn; ++i)
                    if (s->node_right[i] < 0)
                        for (int e = s->node_first[i]; e < s->node_first[i] + s->node_count[i]; ++e) {
                            const int p = s->leaf_prim[e];
                            if (p >= 0 && (size_t)p < np) ent[fill[p]++] = make_uint2((uint32_t)e, (uint32_t)i);
                        }
                TRY_C(upload(ctx, ctx->scene_allocs, off, &ctx->F.tie_off));
                TRY_C(upload(ctx, ctx->scene_allocs, ent, &ctx->F.tie_ent));
            }
        }
        float ext = 0;
        for (int k = 0; k < 3; ++k) {
            ctx->F.root_lo[k] = round_down(s->node_lo[k]);
            ctx->F.root_hi[k] = round_up(s->node_hi[k]);
            ext = fmaxf(ext, fmaxf(fabsf(ctx->F.root_lo[k]), fabsf(ctx->F.root_hi[k])));
        }
        ctx->F.extent = ext * 1.000001f;
        ctx->F.depth = max_depth;
        ctx->F.ok = ok ? 1 : 0;
        ctx->lds_bytes = ok ? (size_t)std::max(max_depth, 2) * RT_BLOCK * sizeof(uint2) : 0;     // (>= 2 rows: k_beam_setup stages tile rows there)
        if (ok) {
            // per-lane stacks can exceed the default 64 KB dynamic-LDS limit on deep trees
            const int big = 160 * 1024;
            HIP_C(hipFuncSetAttribute((const void *)&k_intersect<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, big - 8192));   // 5 KB of static LDS (hit and child buffers)
            HIP_C(hipFuncSetAttribute((const void *)&k_intersect<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big - 12288));   // 8 KB of static LDS (hit and child buffers)
            HIP_C(hipFuncSetAttribute((const void *)&k_shadow<true>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
            HIP_C(hipFuncSetAttribute((const void *)&k_shadow_queue, hipFuncAttributeMaxDynamicSharedMemorySize, big));
            HIP_C(hipFuncSetAttribute((const void *)&k_shadow_walk, hipFuncAttributeMaxDynamicSharedMemorySize, big - 8192));   // 4 KB of static LDS (queue buffer)
            // 14 KB of static LDS (list buffer, touched entries)
            HIP_C(hipFuncSetAttribute((const void *)&k_beam_setup<false, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, big - 16384));
            HIP_C(hipFuncSetAttribute((const void *)&k_beam_setup<false, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, big - 16384));
            HIP_C(hipFuncSetAttribute((const void *)&k_beam_setup<true, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, big - 16384));
            HIP_C(hipFuncSetAttribute((const void *)&k_beam_setup<true, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, big - 16384));
            HIP_C(hipFuncSetAttribute((const void *)&k_beam_setup<false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big - 16384));
            HIP_C(hipFuncSetAttribute((const void *)&k_beam_setup<false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big - 16384));
            HIP_C(hipFuncSetAttribute((const void *)&k_intersect_only<true>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
            HIP_C(hipFuncSetAttribute((const void *)&k_visibility<true>, hipFuncAttributeMaxDynamicSharedMemorySize, big));
        }
        if (getenv("RT_AMD_DEBUG")) fprintf(stderr, "[rt_amd] kd view: ok=%d depth=%d lds=%zu B\n", ctx->F.ok, max_depth, ctx->lds_bytes);
    }

    tick("kd view + upload");
    // ---- grid = exactly the workgroups that are resident at once (every kernel walks its queue with a fixed
    // grid; a second, partially filled round of workgroups would run at a fraction of the occupancy) ----
    {
        int last_nb = 1;
        auto grid_for = [&](const void *fn, size_t lds) {
            int nb = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, RT_BLOCK, lds) != hipSuccess || nb < 1) nb = 1;
            if (nb > 6) nb = 6;       // ~100 SGPRs per wave admit 6 workgroups of 256 per CU (MI355X_MICROARCH.md, residency)
            last_nb = nb;
            int g = ctx->n_cu * nb;
            return (g + 7) & ~7;
        };
        const size_t lds = ctx->lds_bytes;
        ctx->g_intersect = grid_for((const void *)&k_intersect<true, true>, lds + ctx->lds_top_bytes); ctx->nb_intersect = last_nb;
        ctx->g_intersect_x = grid_for((const void *)&k_intersect<false, false>, 0);
        ctx->g_shadow = grid_for((const void *)&k_shadow<true>, lds); ctx->nb_shadow = last_nb;
        ctx->g_shadow_x = grid_for((const void *)&k_shadow<false>, 0);
        ctx->g_beam_setup = grid_for((const void *)&k_beam_setup<false, true, false>, lds); ctx->nb_beam_setup = last_nb;
        ctx->g_beam_setup_t = grid_for((const void *)&k_beam_setup<false, true, true>, lds); ctx->nb_beam_setup_t = last_nb;
        ctx->g_beam = grid_for((const void *)&k_shadow_beam, 0); ctx->nb_beam = last_nb;
        ctx->g_queue = grid_for((const void *)&k_shadow_queue, lds); ctx->nb_queue = last_nb;
        ctx->g_walk = grid_for((const void *)&k_shadow_walk, lds); ctx->nb_walk = last_nb;
        ctx->g_shade = grid_for((const void *)&k_shade<RT_SHADE_LIGHTS>, 0); ctx->nb_shade = last_nb;
        if (getenv("RT_AMD_DEBUG"))
            fprintf(stderr, "[rt_amd] grids: intersect %d beam_setup %d beam %d queue %d shade %d\n", ctx->g_intersect,
                    ctx->g_beam_setup, ctx->g_beam, ctx->g_queue, ctx->g_shade);
    }

    tick("occupancy queries");
    // ---- planes / surfaces / materials / lights ----
    std::vector<DPlane> planes(s->n_planes);
    for (int i = 0; i < s->n_planes; ++i) {
        for (int k = 0; k < 3; ++k) planes[i].n[k] = s->pln_normal[3 * i + k];
        planes[i].d = s->pln_d[i]; planes[i].id = s->pln_id[i]; planes[i].pad = 0;
    }
    TRY_C(upload(ctx, ctx->scene_allocs, planes, &ctx->S.planes));
    std::vector<DMat> mats(s->n_materials);
    for (int i = 0; i < s->n_materials; ++i) {
        DMat &m = mats[i];
        for (int k = 0; k < 3; ++k) { m.kd[k] = s->mat_kd[3 * i + k]; m.ks[k] = s->mat_ks[3 * i + k]; m.kr[k] = s->mat_kr[3 * i + k]; }
        m.T = s->mat_transparency[i];
        m.phong = s->mat_phong[i];
        // ray_tracer.py:108 (any(T != 0)) and :119 (norm(Kr) != 0, evaluated like np.linalg.norm)
        double nk = sqrt((m.kr[0] * m.kr[0] + m.kr[1] * m.kr[1]) + m.kr[2] * m.kr[2]);
        m.flags = (m.T != 0 ? 1 : 0) | (nk != 0 ? 2 : 0);
    }
    std::vector<DSurf> surfs(s->n_surfaces);
    parallel_for((size_t)s->n_surfaces, [&](size_t b0, size_t e0) {
        for (size_t i = b0; i < e0; ++i) {
            DSurf &d = surfs[i];
            memset(&d, 0, sizeof(d));
            d.kind = s->surf_kind[i]; d.mat = s->surf_mat[i];
            d.mflags = (d.mat >= 0 && d.mat < s->n_materials) ? mats[d.mat].flags : 0;      // k_intersect: does a hit here spawn children?
            const size_t g = (size_t)s->surf_geom[i];
            if (d.kind == RT_SURF_SPHERE) for (int k = 0; k < 3; ++k) d.p[k] = s->sph_center[3 * g + k];
            if (d.kind == RT_SURF_BOX) { for (int k = 0; k < 3; ++k) d.p[k] = s->box_center[3 * g + k]; d.p[3] = s->box_scale[g]; }
            if (d.kind == RT_SURF_PLANE) for (int k = 0; k < 3; ++k) d.p[k] = s->pln_normal[3 * g + k];
        }
    });
    TRY_C(upload(ctx, ctx->scene_allocs, surfs, &ctx->S.surfs));
    TRY_C(upload(ctx, ctx->scene_allocs, mats, &ctx->S.mats));
    std::vector<DLight> lights(s->n_lights);
    std::vector<int32_t> sl;
    for (int i = 0; i < s->n_lights; ++i) {
        DLight &l = lights[i];
        for (int k = 0; k < 3; ++k) { l.pos[k] = s->lgt_pos[3 * i + k]; l.color[k] = s->lgt_color[3 * i + k]; }
        l.spec = s->lgt_spec[i]; l.shadow = s->lgt_shadow[i]; l.radius = s->lgt_radius[i];
        if (l.shadow != 0) sl.push_back(i);   // Light.py:199: shadow_intensity == 0 casts no rays
    }
    TRY_C(upload(ctx, ctx->scene_allocs, lights, &ctx->S.lights));
    TRY_C(upload(ctx, ctx->scene_allocs, sl, &ctx->S.shadow_lights));
    ctx->S.n_prims = ns + nb; ctx->S.n_nodes = nn; ctx->S.n_planes = s->n_planes; ctx->S.n_surfs = s->n_surfaces;
    ctx->S.n_mats = s->n_materials; ctx->S.n_lights = s->n_lights; ctx->S.n_shadow_lights = (int)sl.size();
    ctx->S.far_clip = s->far_clip;
    {
        double kw = 0;
        for (int i = 0; i < s->n_lights; ++i)
            for (int k = 0; k < 3; ++k) kw = std::max(kw, fabs(s->lgt_pos[3 * i + k]) + 2.0 * fabs(s->lgt_radius[i]));
        double dmax = 0;
        for (int i = 0; i < s->n_planes; ++i) dmax = std::max(dmax, fabs(s->pln_d[i]));
        ctx->S.plane_kw = round_up((kw + dmax) * 1.74);     // |o|_2 <= 1.74 |o|_inf
    }
    for (int k = 0; k < 3; ++k) ctx->S.bg[k] = s->background[k];
    tick("surface / material / light tables");
    {
        // small scenes: the tables k_small_level copies into LDS (rt_small.hpp)
        auto up16 = [](size_t b) { return (uint32_t)((b + 15) & ~(size_t)15); };
        SmallLayout &L = ctx->small;
        uint32_t off = 0;
        L.planes = off; off += up16((size_t)s->n_planes * sizeof(DPlane));
        L.nodes = off; off += up16((size_t)nn * sizeof(DNode));
        L.leaf = off; off += up16((size_t)s->n_leaf_refs * sizeof(int32_t));
        L.prims = off; off += up16((size_t)(ns + nb) * sizeof(DPrim));
        L.surfs = off; off += up16((size_t)s->n_surfaces * sizeof(DSurf));
        L.mats = off; off += up16((size_t)s->n_materials * sizeof(DMat));
        L.lights = off; off += up16((size_t)s->n_lights * sizeof(DLight));
        L.shadow_lights = off; off += up16(sl.size() * sizeof(int32_t));
        L.total = off;
        const size_t raw = (size_t)s->n_planes * sizeof(DPlane) + (size_t)nn * sizeof(DNode) + (size_t)(ns + nb) * sizeof(DPrim) +
                           (size_t)s->n_surfaces * sizeof(DSurf);
        ctx->small_ok = raw <= RT_SMALL_LDS_MAX && off <= RT_SMALL_LDS_MAX;
        if (ctx->small_ok) {
            int nbk = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nbk, (const void *)&k_small_level<true>, RT_BLOCK, L.total) != hipSuccess || nbk < 1) nbk = 1;
            if (nbk > 6) nbk = 6;
            ctx->g_small = (ctx->n_cu * nbk + 7) & ~7;
        }
        if (ctx->env.debug) fprintf(stderr, "[rt_amd] small-scene path: %s (%u B of tables), grid %d\n", ctx->small_ok ? "on" : "off", L.total, ctx->g_small);
    }
    if (const char *lim = getenv("RT_AMD_SETS")) ctx->env_sets = atoi(lim);      // 1, 2 or 3 buffer sets (diagnostic)
    ctx->host = *s;   // pointers inside are NOT used after this call; scalars (camera, settings) are
#undef TRY_C
#undef HIP_C
    *out = ctx;
    return RT_OK;
}

static void free_frame(rt_ctx *ctx) {
    for (void *p : ctx->frame_allocs) (void)hipFree(p);
    ctx->frame_allocs.clear();
    for (void *p : ctx->alt_allocs) (void)hipFree(p);      // (whichever copy the context's pointers named: both are gone now)
    ctx->alt_allocs.clear();
    ctx->alt = AltL0{};
    ctx->alt_bytes = 0;
    ctx->early_prev = false;
    ctx->ev_end_set[0] = ctx->ev_end_set[1] = false;
    FrameState empty;
    empty.ctl = ctx->ctl_own;
    static_cast<FrameState &>(*ctx) = empty;
}

extern "C" int rt_destroy(rt_ctx *ctx) {
    if (!ctx) return RT_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    free_frame(ctx);
    for (void *p : ctx->scene_allocs) (void)hipFree(p);
    if (ctx->ctl_own) (void)hipFree(ctx->ctl_own);
    if (ctx->d_voided) (void)hipFree(ctx->d_voided);
    if (ctx->stage) (void)hipFree(ctx->stage);
    for (hipEvent_t e : ctx->prof_bs) (void)hipEventDestroy(e);
    if (ctx->acc) (void)hipFree(ctx->acc);
    if (ctx->h_ctl) (void)hipHostFree(ctx->h_ctl);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    for (hipEvent_t e : ctx->prof_ev) (void)hipEventDestroy(e);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->ev_cam) (void)hipEventDestroy(ctx->ev_cam);
    if (ctx->ev_prev) (void)hipEventDestroy(ctx->ev_prev);
    for (hipEvent_t e : ctx->ev_end) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->ev_hit) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->ev_lit) (void)hipEventDestroy(e);
    if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
    if (ctx->stream3) (void)hipStreamDestroy(ctx->stream3);
    if (ctx->stream4) (void)hipStreamDestroy(ctx->stream4);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return RT_OK;
}

// ------------------------------------------------------------------------------------------------
// frame state: one carving routine, two sources of memory
// ------------------------------------------------------------------------------------------------
// Memory source of carve_frame: one hipMalloc per buffer (owned by the context), or a bump pointer inside `scratch`.
struct Carver {
    rt_ctx *ctx;
    unsigned char *base = nullptr;   // bump mode: start of the region (device address), else nullptr
    uint64_t used = 0, limit = 0;
    template <typename T>
    int get(T **p, uint64_t n) {
        const uint64_t bytes = ((n ? n : 1) * sizeof(T) + 255ull) & ~255ull;
        if (base) {
            if (used + bytes > limit) return fail(ctx, RT_ERR_UNSUPPORTED, "frame does not fit the static scratch buffer of the no-alloc entry point");
            *p = (T *)(base + used);
            used += bytes;
            return RT_OK;
        }
        void *v = nullptr;
        HIP_TRY(ctx, rt_malloc(&v, bytes));
        ctx->frame_allocs.push_back(v);
        *p = (T *)v;
        used += bytes;
        return RT_OK;
    }
};

// Tile beams pay in scenes whose tree takes a pair tens of node steps to walk: measured -11 % (1920x1080) to -16 % (3840x2160) on
// the 10 k lattice, -6 % on the 90 k lattice (a sixth of its rows overflow), +5 % on the 20-primitive pool scene (nothing to save
// there, break-even at 100 primitives, -7 % at 400 -> from 256 primitives on).  Scenes of >= 100 k primitives keep the walk with the umbra test inside it (lp.beam_early).  A rule of the scene, like
// every other choice of kernels here.
static bool tile_scene_ok(const rt_ctx *ctx) {
    const bool long_beams = ctx->env.beam_early >= 0 ? ctx->env.beam_early != 0 : ctx->S.n_prims >= 100000;      // (lp.beam_early: the walk with the umbra test)
    return ctx->F.ok && !long_beams && ctx->S.n_prims >= ctx->env.tiles_min_prims && ctx->S.n_prims <= (int)RT_ZONE_NC;
}

// Record slots a wave of k_beam_setup takes at a time (BeamBuf::rec_blk): one returning atomic per block - 64 where a level
// holds millions of pairs (one atomic per wave and chunk on one address was a bottleneck there), fewer in small frames, where
// the unused tail of every wave's last block is what counts.
static uint32_t rec_block_size(uint64_t pairs) { return pairs >= (1ull << 20) ? 64u : (pairs >= (1ull << 17) ? 32u : 8u); }
// Record slots of a buffer set of `pairs` (hit, light) pairs: pairs / rec_div, plus the blocks' tails (a wave that takes part
// handles at least one run of 64 pairs and leaves less than a block unused), a multiple of 64.
static uint32_t rec_capacity(const rt_ctx *ctx, uint64_t pairs, double rec_div) {
    const uint64_t waves = (uint64_t)(RT_BLOCK / 64) * (uint64_t)(ctx->n_cu * 6 + 8);        // the largest grid a kernel is launched with
    const uint64_t slack = (uint64_t)rec_block_size(pairs) * std::min<uint64_t>(waves, pairs / 64 + 1);
    // (up to a million pairs - 256 MB of records - every pair may have one: the fraction is for the frames whose memory counts)
    const double div = pairs <= (1ull << 20) ? 1.0 : (rec_div >= 1.0 ? rec_div : 1.0);
    const uint64_t want = std::max<uint64_t>((uint64_t)((double)pairs / div), std::min<uint64_t>(pairs, 1ull << 20)) + slack + 64;
    return (uint32_t)std::min<uint64_t>((want + 63) & ~63ull, 0xFFFFFFC0ull);
}

// Carve the state of a frame of `slots` primary slots with queue capacity `cap` into fs.  sets = 1..3 buffer sets (level 0 /
// deeper levels odd / even); queues = 3 (levels in sequence) or 4 (pipelined deeper levels).  Nothing is initialised here.
// spb: the fewest pairs a chunk of RT_BLOCK shadow rays will hold in frames on this state (RT_BLOCK / (s * s); 1 = any s): the lit
// masks of k_shadow_beam are one word per wave and chunk of listed pairs.
static int carve_frame(rt_ctx *ctx, Carver &mem, FrameState &fs, uint64_t slots, uint64_t cap0, uint64_t cap_deep, double qgrowth, double rec_div, uint32_t spb, int sets,
                       int queues, bool tiles, uint64_t qbase0 = 0) {
#define TRY_A(expr) do { int rc_ = (expr); if (rc_ != RT_OK) return rc_; } while (0)
    if (sets < 2) cap_deep = cap0;          // one set: every level lives in it
    auto qcap = [&](uint64_t c) { const double v = (double)c * qgrowth; return (uint32_t)(v < 2147483647.0 ? v : 2147483647.0); };
    for (int i = 0; i < queues; ++i) {
        RayQ &q = fs.q[i];
        const uint64_t cap = i == 0 ? cap0 : cap_deep;
        TRY_A(mem.get(&q.ox, cap)); TRY_A(mem.get(&q.oy, cap)); TRY_A(mem.get(&q.oz, cap));
        TRY_A(mem.get(&q.dx, cap)); TRY_A(mem.get(&q.dy, cap)); TRY_A(mem.get(&q.dz, cap));
        TRY_A(mem.get(&q.tr, cap)); TRY_A(mem.get(&q.tg, cap)); TRY_A(mem.get(&q.tb, cap));
        TRY_A(mem.get(&q.slot, cap)); TRY_A(mem.get(&q.gpix, cap)); TRY_A(mem.get(&q.path, cap));
    }
    for (int i = queues; i < 6; ++i) fs.q[i] = fs.q[queues - 1];
    const uint64_t nsl1 = (uint64_t)(ctx->S.n_shadow_lights > 0 ? ctx->S.n_shadow_lights : 1);
    if (!qbase0) qbase0 = cap0;             // the float64 queue of the first set: qgrowth entries per qbase0
    fs.beam_queue_cap = qcap(qbase0);
    for (int set = 0; set < sets; ++set) {
        const uint64_t cap = set == 0 ? cap0 : cap_deep, nlit = cap * nsl1;
        const uint32_t queue_cap = qcap(set == 0 ? qbase0 : cap);
        HitBuf &hb = *hit_set(fs, set);
        TRY_A(mem.get(&hb.px, cap)); TRY_A(mem.get(&hb.py, cap)); TRY_A(mem.get(&hb.pz, cap));
        TRY_A(mem.get(&hb.id, cap)); TRY_A(mem.get(&hb.list, cap));
        TRY_A(mem.get(&hb.lit, nlit));
        BeamBuf &bb = *beam_set(fs, set);
        const uint64_t rcap = rec_capacity(ctx, nlit, rec_div);
        bb.rec_cap = (uint32_t)rcap;
        bb.rec_blk = rec_block_size(nlit);
        TRY_A(mem.get(&bb.rec, rcap * RT_PAIR_UNITS));
        TRY_A(mem.get(&bb.queue, (uint64_t)queue_cap)); TRY_A(mem.get(&bb.queue_n, 4));
        TRY_A(mem.get(&bb.litbits, (rcap / (spb ? spb : 1u) + 2) * (RT_BLOCK / 64)));
        TRY_A(mem.get(&bb.list, rcap)); TRY_A(mem.get(&bb.pair_slot, nlit)); TRY_A(mem.get(&bb.ovlist, rcap));
        bb.blockbox = nullptr; bb.trec = nullptr; bb.tcnt = nullptr; bb.zlist = bb.zcnt = nullptr;
        if (set == 0) { fs.tile_nblk = fs.tile_nreg = fs.tile_nzone = 0; }
        if (set == 0 && tiles && tile_scene_ok(ctx)) {
            // tile beams serve level 0, which always runs on the first set.  Regions / zones: twice what a frame of square-ish
            // tiles needs (ragged edges of the block grid); a frame that needs more runs without tile beams
            const uint64_t nblk = (slots + 63) / 64, nreg = nblk / 8 + 64, nzone = nreg / 16 + 8;
            const uint64_t zcap = std::min<uint64_t>((uint64_t)ctx->S.n_prims, RT_ZONE_NC);
            TRY_A(mem.get(&bb.blockbox, nblk * 8));
            TRY_A(mem.get(&bb.trec, nblk * nsl1 * 2 * RT_TILE_NC)); TRY_A(mem.get(&bb.tcnt, nblk * nsl1));
            TRY_A(mem.get(&bb.zlist, nzone * nsl1 * zcap)); TRY_A(mem.get(&bb.zcnt, nzone * nsl1));
            fs.tile_nblk = (uint32_t)nblk; fs.tile_nreg = (uint32_t)nreg; fs.tile_nzone = (uint32_t)nzone;
        }
    }
    if (sets < 2) { fs.hb1 = fs.hb; fs.beam1 = fs.beam; }
    if (sets < 3) { fs.hb2 = fs.hb1; fs.beam2 = fs.beam1; }
    if (sets < 5) { fs.hb3 = fs.hb1; fs.beam3 = fs.beam1; fs.hb4 = fs.hb2; fs.beam4 = fs.beam2; }
    fs.two_sets = sets >= 2;
    fs.three_sets = sets >= 3;
    fs.five_sets = sets >= 5;
    TRY_A(mem.get(&fs.fb0, slots * 3));
    TRY_A(mem.get(&fs.fb, slots * 3));
    TRY_A(mem.get(&fs.fbm, slots * 3));
#undef TRY_A
    fs.cap = cap0;
    fs.cap_deep = cap_deep;
    fs.qgrowth_carved = qgrowth;
    fs.rec_div_carved = rec_div;
    fs.spb_carved = spb ? spb : 1u;
    fs.fb_slots = slots;
    return RT_OK;
}

// what must hold before the first frame on freshly carved (or reused static) buffers: lit counters zero - k_shade resets them
// only as it consumes them, a frame voided by a queue overflow leaves counts behind -, append counters zero, queue capacity set
static int init_frame_counters(rt_ctx *ctx, FrameState &fs, hipStream_t st) {
    const uint64_t nsl1 = (uint64_t)(ctx->S.n_shadow_lights > 0 ? ctx->S.n_shadow_lights : 1);
    const int sets = fs.five_sets ? 5 : (fs.three_sets ? 3 : (fs.two_sets ? 2 : 1));
    for (int set = 0; set < sets; ++set) {
        HitBuf &hb = *hit_set(fs, set);
        BeamBuf &bb = *beam_set(fs, set);
        const uint64_t cap = set == 0 ? fs.cap : fs.cap_deep;
        const double qd = set == 0 ? (double)fs.beam_queue_cap : (double)cap * fs.qgrowth_carved;
        HIP_TRY(ctx, hipMemsetAsync(hb.lit, 0, cap * nsl1 * sizeof(uint32_t), st));
        hipLaunchKernelGGL(k_init_beam_counters, dim3(1), dim3(64), 0, st, bb.queue_n, (uint32_t)(qd < 2147483647.0 ? qd : 2147483647.0));
    }
    HIP_TRY(ctx, hipGetLastError());
    return RT_OK;
}

// (re)allocate queues for `slots` primary slots; only grows
// sh: shadow rays per light and axis of the frame to come (0: none will be cast)
static int ensure_frame(rt_ctx *ctx, uint64_t slots, int32_t sh) {
    const uint32_t s2n = (uint32_t)(sh > 0 ? sh * sh : 0), spb = (s2n >= 9u && s2n <= RT_BLOCK) ? RT_BLOCK / s2n : RT_BLOCK;      // (the beam path takes 9 <= s * s <= RT_BLOCK)
    // Level 0 holds the camera rays (or the caller's rays) and nothing else: its queue and buffer set are sized by the slots.  The
    // levels >= 1 may hold up to `growth` rays per slot (two children per hit) and usually hold a fraction of one: their queues and
    // sets start at growth / deep_div = 1 ray per slot; deep_div is halved, then growth doubled, when a level overflows.
    const uint64_t want_full = (uint64_t)((double)slots * ctx->growth) + 1024;
    if (want_full > 0x7FFFFFF0ull) return fail(ctx, RT_ERR_UNSUPPORTED, "ray queue would exceed 2^31 entries");      // (k_intersect parks children as queue index << 1 | kind)
    const double want_qd = (double)want_full * ctx->qgrowth;
    const uint32_t want_q = (uint32_t)(want_qd < 2147483647.0 ? want_qd : 2147483647.0);
    const uint64_t want_deep = std::min<uint64_t>(want_full, (uint64_t)((double)want_full / ctx->deep_div) + 1024);
    uint64_t want_cap = slots + 1024;
    if (ctx->frame_bytes && !ctx->two_sets) want_cap = std::max(want_cap, want_deep);      // one buffer set: every level lives in it
    if (ctx->cap >= want_cap && ctx->cap_deep >= want_deep && ctx->fb_slots >= slots && ctx->beam_queue_cap >= want_q && ctx->rec_div_carved <= ctx->rec_div && ctx->spb_carved <= spb) {
        if (ctx->counters_dirty) {
            // the last frame was voided by a queue overflow: its shading kernels returned early and left lit counts behind
            // (k_shade clears them only as it consumes them); a smaller frame that fits the old buffers must not see them
            HIP_TRY(ctx, hipDeviceSynchronize());
            if (int rc = init_frame_counters(ctx, *ctx, ctx->stream)) return rc;
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            ctx->counters_dirty = false;
        }
        return RT_OK;
    }
    ctx->counters_dirty = false;
    HIP_TRY(ctx, hipDeviceSynchronize());   // rare: buffers of an in-flight frame are about to be replaced
    const uint32_t spb_keep = ctx->cap ? std::min(spb, ctx->spb_carved) : spb;      // (frames of fewer pairs per chunk were rendered on the old state: keep serving them)
    free_frame(ctx);
    const uint64_t cap_l0 = slots + 1024;
    const uint64_t nsl1 = (uint64_t)(ctx->S.n_shadow_lights > 0 ? ctx->S.n_shadow_lights : 1);
    // Buffer sets two to five are speed-ups only (levels overlap; a level's buffers are reused two / four levels later), and so
    // are the tile beams: without the memory for them levels share sets and run in sequence, pairs walk the tree themselves.
    // The estimate decides what is tried first; an allocation that fails all the same is tried again with less.
    int sets = 5;
    bool tiles = true;
    auto set_bytes = [&](uint64_t c) {
        const uint64_t pairs = c * nsl1, rcap = rec_capacity(ctx, pairs, ctx->rec_div);
        return c * 40 + pairs * 8 + rcap * (16 * RT_PAIR_UNITS + 12) + (rcap / spb_keep + 2) * 8 * (RT_BLOCK / 64) + (uint64_t)((double)c * ctx->qgrowth) * 8;
    };
    auto tile_bytes = [&]() -> uint64_t {
        if (!tiles || !tile_scene_ok(ctx)) return 0;
        const uint64_t nblk = (slots + 63) / 64, nreg = nblk / 8 + 64, nzone = nreg / 16 + 8;
        const uint64_t zcap = std::min<uint64_t>((uint64_t)ctx->S.n_prims, RT_ZONE_NC);
        return nblk * 32 + nblk * nsl1 * (32 * RT_TILE_NC + 4) + nzone * nsl1 * (zcap + 1) * 4;
    };
    {
        size_t free_b = 0, total_b = 0;
        const uint64_t queues_b = (cap_l0 + 5 * want_deep) * 84 + (uint64_t)want_qd * 8, fb_b = slots * 72;
        const uint64_t set0_b = set_bytes(cap_l0), setd_b = set_bytes(want_deep);
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) sets = 1;
        else {
            const uint64_t avail = (uint64_t)free_b > queues_b + fb_b ? (uint64_t)free_b - queues_b - fb_b : 0;
            auto need = [&](int n) { const uint64_t b = set0_b + (uint64_t)(n - 1) * setd_b + tile_bytes(); return b + b / 16; };
            while (sets > 1 && avail < need(sets)) --sets;
            if (sets == 1 && avail < need(1)) tiles = false;
        }
        if (ctx->env_sets > 0 && ctx->env_sets < sets) sets = ctx->env_sets;      // RT_AMD_SETS = 1, 2, 3 or 5 / RT_AMD_ONE_SET (diagnostic)
        if (sets == 4) sets = 3;
    }
    int rc = RT_OK;
    for (;;) {
        const uint64_t cap = sets < 2 ? std::max(cap_l0, want_deep) : cap_l0;        // one buffer set: every level lives in it
        Carver mem{ctx};
        FrameState fs;
        fs.ctl = ctx->ctl_own;
        rc = carve_frame(ctx, mem, fs, slots, cap, want_deep, ctx->qgrowth, ctx->rec_div, spb_keep, sets, sets >= 5 ? 6 : 4, tiles, want_full);
        if (rc == RT_OK) rc = init_frame_counters(ctx, fs, ctx->stream);
        if (rc == RT_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = fail(ctx, RT_ERR_HIP, "hipStreamSynchronize failed after frame set-up");
        if (rc == RT_OK) {
            static_cast<FrameState &>(*ctx) = fs;
            ctx->frame_bytes = mem.used;
            return RT_OK;
        }
        free_frame(ctx);
        (void)hipGetLastError();
        if (rc != RT_ERR_OOM) return rc;
        // out of memory although the estimate fitted (other contexts, a PyTorch pool): fewer sets, then no tile beams
        if (sets > 1) sets = sets == 5 ? 3 : sets - 1;
        else if (tiles && tile_scene_ok(ctx)) tiles = false;
        else return rc;
    }
}

struct TileGeom {
    int W, H, tw, th, tiles_x, tiles_y, n_tiles, world, rank, tiles_per_rank;
    uint64_t slots;      // ray slots processed by this rank (tile padded)
    uint64_t out_slots;  // entries in this rank's output buffers
};

static int tile_geom(const rt_render_params *p, TileGeom &g, std::string &why) {
    if (!p) { why = "params is NULL"; return RT_ERR_INVALID_ARG; }
    if (p->width <= 0 || p->height <= 0) { why = "width/height must be positive"; return RT_ERR_INVALID_ARG; }
    g.W = p->width; g.H = p->height;
    g.tw = p->tile_w > 0 ? p->tile_w : 8;
    g.th = p->tile_h > 0 ? p->tile_h : 8;
    g.world = p->world > 1 ? p->world : 1;
    g.rank = g.world > 1 ? p->rank : 0;
    if (g.rank < 0 || g.rank >= g.world) { why = "rank out of range"; return RT_ERR_INVALID_ARG; }
    g.tiles_x = (g.W + g.tw - 1) / g.tw;
    g.tiles_y = (g.H + g.th - 1) / g.th;
    g.n_tiles = g.tiles_x * g.tiles_y;
    g.tiles_per_rank = (g.n_tiles + g.world - 1) / g.world;
    g.slots = (uint64_t)g.tiles_per_rank * g.tw * g.th;
    g.out_slots = g.world > 1 ? g.slots : (uint64_t)g.W * g.H;
    if ((uint64_t)g.W * g.H > 0xFFFFFFF0ull) { why = "image too large"; return RT_ERR_UNSUPPORTED; }
    return RT_OK;
}

extern "C" int64_t rt_tile_pixels(const rt_render_params *p) {
    TileGeom g;
    std::string why;
    if (tile_geom(p, g, why) != RT_OK) return -1;
    return (int64_t)g.out_slots;
}

extern "C" int rt_tile_layout(const rt_render_params *p, int64_t *pixel_index) {
    TileGeom g;
    std::string why;
    if (int rc = tile_geom(p, g, why)) return fail(nullptr, rc, why);
    if (!pixel_index) return fail(nullptr, RT_ERR_INVALID_ARG, "pixel_index is NULL");
    if (g.world == 1) {
        for (int64_t k = 0; k < (int64_t)g.out_slots; ++k) pixel_index[k] = k;
        return RT_OK;
    }
    for (uint64_t k = 0; k < g.slots; ++k) {
        uint32_t tile = (uint32_t)g.rank + (uint32_t)(k / (uint32_t)(g.tw * g.th)) * (uint32_t)g.world;
        uint32_t lane = (uint32_t)(k % (uint32_t)(g.tw * g.th));
        int x = (int)(tile % (uint32_t)g.tiles_x) * g.tw + (int)(lane % (uint32_t)g.tw);
        int y = (int)(tile / (uint32_t)g.tiles_x) * g.th + (int)(lane / (uint32_t)g.tw);
        pixel_index[k] = (tile < (uint32_t)g.n_tiles && x < g.W && y < g.H) ? (int64_t)y * g.W + x : -1;
    }
    return RT_OK;
}

static int resolve_settings(rt_ctx *ctx, const rt_render_params *p, int32_t &sh, int32_t &rec) {
    sh = p->sh_rays == RT_USE_SCENE_VALUE ? ctx->host.sh_rays : p->sh_rays;
    rec = p->rec_max == RT_USE_SCENE_VALUE ? ctx->host.rec_max : p->rec_max;
    if (sh < 1 && ctx->S.n_shadow_lights > 0) return fail(ctx, RT_ERR_INVALID_ARG, "root_number_shadow_rays must be >= 1");
    if (sh > 1024) return fail(ctx, RT_ERR_UNSUPPORTED, "root_number_shadow_rays > 1024");
    if (rec > RT_MAX_LEVELS - 2) return fail(ctx, RT_ERR_UNSUPPORTED, "max_recursions > 62");
    return RT_OK;
}

static int prof_prepare(rt_ctx *ctx, int levels) {
    size_t need = 4 + 6 * (size_t)levels;
    while (ctx->prof_ev.size() < need) {
        hipEvent_t e;
        HIP_TRY(ctx, rt_event_create(&e, true));
        ctx->prof_ev.push_back(e);
    }
    ctx->prof_levels = levels;
    return RT_OK;
}
// events around the k_beam_setup launches alone (RT_FLAG_PROFILE and RT_FLAG_TIME_DOMINANT)
static int prof_bs_prepare(rt_ctx *ctx, int levels) {
    while (ctx->prof_bs.size() < 2 * (size_t)levels) {
        hipEvent_t e;
        HIP_TRY(ctx, rt_event_create(&e, true));
        ctx->prof_bs.push_back(e);
    }
    ctx->prof_bs_levels = 0;
    return RT_OK;
}
#define PROF_MARK(idx) do { if (prof) HIP_TRY(ctx, hipEventRecord(ctx->prof_ev[(idx)], st)); } while (0)

// enqueue levels 0..rec; queue 0 must hold the level-0 rays.  fb0 receives what the level-0 rays add themselves.
//
// Overlapped schedule (fb0 != fb, rec >= 1, no profiling): the deeper levels hold few rays but long ones (reflections
// skimming through the field) - their kernels are latency chains that leave the GPU idle, ~1/3 of a sequential
// frame.  Level 0's children only need the hits, so after k_intersect(0) the frame forks:
//     st      : soft shadows of level 0 -> Phong term of level 0 (fb0)                 [hb, seg, beam, q[0] read-only]
//     stream2 : children of level 0 -> levels 1..rec complete (fb)                     [hb1, seg1, beam1, q[1], q[2]]
// and joins before the resolve.  Both sides use fixed grids; the level-0 side leaves one workgroup slot per CU to
// the other.  Results do not depend on the schedule: the two sides add into different framebuffers.
// gen != nullptr: queue 0 is empty, k_intersect(0) makes the camera rays itself
//
// Pipelined deeper levels (three buffer sets, rec >= 2): a level's nearest-hit query is a latency chain as long as its
// longest ray, and so is its soft-shadow pass; run back to back they made the deeper levels the longer half of small
// frames.  The children of level L only need its hits, so
//     stream2 : spawn(0) -> hits(1) -> spawn(1) -> hits(2) -> ...                       [misses add into fbm]
//     stream3 : for L = 1..rec, as soon as hits(L) are known: soft shadows(L) -> Phong(L) [adds into fb, in level order]
// hits(L) reuses the buffer set and ray queue of level L-2 and waits for that level's shading.
// fbm: where the rays of levels 1.. add the background when they miss (may be fb itself when nothing overlaps).
// Which schedule a frame takes (one place: enqueue_frame decides about the early camera pass before enqueue_levels lays the
// levels out).  own_fb0: level 0 has a framebuffer of its own (camera frames).
struct Sched { bool fast, overlap, small, pipelined; };
static Sched plan_schedule(const rt_ctx *ctx, const rt_render_params *p, int32_t sh, int32_t rec, bool own_fb0) {
    Sched sc;
    const bool prof = (p->flags & RT_FLAG_PROFILE) != 0;
    const uint32_t s2 = (uint32_t)(sh * sh);
    sc.fast = ctx->F.ok && !(p->flags & RT_FLAG_EXACT_TRAVERSAL);
    sc.overlap = sc.fast && !prof && rec >= 1 && own_fb0 && ctx->two_sets && !(p->flags & RT_FLAG_NO_OVERLAP) && ctx->n_cu >= 8;
    // small scene, few shadow rays per hit: one fused kernel per level over the LDS-resident scene (rt_small.hpp)
    sc.small = ctx->small_ok && !prof && (s2 <= 4 || ctx->S.n_shadow_lights == 0) &&
               !(p->flags & (RT_FLAG_EXACT_TRAVERSAL | RT_FLAG_NO_BEAM | RT_FLAG_NO_SMALL));
    sc.pipelined = sc.overlap && !sc.small && ctx->three_sets && rec >= 2 && !ctx->env.no_pipeline;
    return sc;
}

// the second copy of the level-0 state (AltL0), made when the first frame that can use it arrives; false: no memory for it
static bool ensure_alt(rt_ctx *ctx) {
    AltL0 &a = ctx->alt;
    if (a.ok) return true;
    if (a.failed || !ctx->cap || !ctx->fb_slots) return false;
    uint64_t used = 0;
    bool good = true;
    auto get = [&](auto **pp, uint64_t n) {
        if (!good) return;
        const uint64_t bytes = ((n ? n : 1) * sizeof(**pp) + 255ull) & ~255ull;
        void *v = nullptr;
        if (rt_malloc(&v, bytes) != hipSuccess) { (void)hipGetLastError(); good = false; return; }
        ctx->alt_allocs.push_back(v);
        *pp = (decltype(*pp))v;
        used += bytes;
    };
    const uint64_t cap = ctx->cap, nsl1 = (uint64_t)(ctx->S.n_shadow_lights > 0 ? ctx->S.n_shadow_lights : 1);
    a.q0 = ctx->q[0];                        // origin and throughput of camera rays are never stored: shared
    get(&a.q0.dx, cap); get(&a.q0.dy, cap); get(&a.q0.dz, cap);
    get(&a.q0.slot, cap); get(&a.q0.gpix, cap); get(&a.q0.path, cap);
    a.hb = ctx->hb;                          // lit counters: shared
    get(&a.hb.px, cap); get(&a.hb.py, cap); get(&a.hb.pz, cap); get(&a.hb.id, cap); get(&a.hb.list, cap);
    get(&a.fb0, ctx->fb_slots * 3); get(&a.fb, ctx->fb_slots * 3); get(&a.fbm, ctx->fb_slots * 3);
    get(&a.ctl, 1);
    get(&a.queue_n, 4);
    if (ctx->beam.blockbox) { get(&a.blockbox, (uint64_t)ctx->tile_nblk * 8); get(&a.zcnt, (uint64_t)ctx->tile_nzone * nsl1); }
    if (good) {
        hipLaunchKernelGGL(k_init_beam_counters, dim3(1), dim3(64), 0, ctx->stream, a.queue_n, ctx->beam_queue_cap);
        good = hipStreamSynchronize(ctx->stream) == hipSuccess;
    }
    if (!good) {
        for (void *v : ctx->alt_allocs) (void)hipFree(v);
        ctx->alt_allocs.clear();
        a = AltL0{};
        a.failed = true;                     // (until the frame state is carved again)
        return false;
    }
    a.ok = true;
    ctx->alt_bytes = used;
    return true;
}
static void swap_alt(rt_ctx *ctx) {
    AltL0 &a = ctx->alt;
    std::swap(ctx->q[0].dx, a.q0.dx); std::swap(ctx->q[0].dy, a.q0.dy); std::swap(ctx->q[0].dz, a.q0.dz);
    std::swap(ctx->q[0].slot, a.q0.slot); std::swap(ctx->q[0].gpix, a.q0.gpix); std::swap(ctx->q[0].path, a.q0.path);
    std::swap(ctx->hb.px, a.hb.px); std::swap(ctx->hb.py, a.hb.py); std::swap(ctx->hb.pz, a.hb.pz);
    std::swap(ctx->hb.id, a.hb.id); std::swap(ctx->hb.list, a.hb.list);
    std::swap(ctx->fb, a.fb); std::swap(ctx->fb0, a.fb0); std::swap(ctx->fbm, a.fbm);
    std::swap(ctx->ctl, a.ctl);
    std::swap(ctx->beam.queue_n, a.queue_n);
    std::swap(ctx->beam.blockbox, a.blockbox); std::swap(ctx->beam.zcnt, a.zcnt);
    ctx->alt_cur ^= 1;
}

static int enqueue_levels(rt_ctx *ctx, const rt_render_params *p, int32_t sh, int32_t rec, int32_t *d_ids, hipStream_t st,
                          double *fb0, const GenParams *gen, double *fbm, const GenParams *cam, bool early_cam = false) {
    LevelParams lp{};
    bool tiles = false;
    ctx->last_tiles = false;
    lp.sh_rays = sh; lp.jitter = p->jitter; lp.seed = p->seed; lp.cap = (uint32_t)ctx->cap; lp.cap_next = (uint32_t)ctx->cap_deep; lp.flags = p->flags;
    auto set_level = [&](int level) {
        lp.level = level;
        lp.last_level = (level == rec) ? 1 : 0;
        lp.cap = (uint32_t)(level == 0 ? ctx->cap : ctx->cap_deep);
        lp.tiles = (tiles && level == 0) ? 1 : 0;
    };
    lp.unit_thr = gen ? 1 : 0;
    {
        // k_shadow_walk pays when a level has millions of rays in overflowing pairs (1 M lattice: 17 M); with few of them
        // a second latency-bound kernel only adds its tail.  RT_AMD_WALK_MIN: 1 = always, -1 = never (diagnostic).
        const long long v = ctx->env.walk_min;
        lp.walk_min = v < 0 || v > 0xfffffffell ? 0xffffffffu : (uint32_t)v;
        // beams that cross many primitives before they leave the scene (large lattices under low lights): stop at the first
        // sphere that covers the light.  Costs ~7 % on scenes whose beams are short; chosen by size, RT_AMD_BEAM_EARLY=0|1 overrides
        lp.beam_early = ctx->env.beam_early >= 0 ? ctx->env.beam_early : (ctx->S.n_prims >= 100000);
    }
    dim3 block(RT_BLOCK);
    const bool prof = (p->flags & RT_FLAG_PROFILE) != 0;
    const bool time_bs = (prof || (p->flags & RT_FLAG_TIME_DOMINANT)) && ctx->prof_bs.size() >= 2 * (size_t)(rec + 1);
    const Sched sc = plan_schedule(ctx, p, sh, rec, fb0 != ctx->fb);
    const bool fast = sc.fast;
    // "the frame before was a pipelined frame on this stream": what an early camera pass and the deeper levels behind it may
    // wait for level by level (ev_lit) instead of for everything enqueued so far.  Any other use of the frame state in between
    // (a ray batch, a frame on another schedule) clears it.
    const bool prev_pipelined = ctx->early_prev && ctx->early_st == st;
    const int prev_rec = ctx->early_prev_rec;
    ctx->early_prev = false;
    const uint32_t s2 = (uint32_t)(sh * sh);
    const bool beam = fast && ctx->S.n_planes <= RT_PMAX && !(p->flags & RT_FLAG_NO_BEAM) && s2 >= 9 && s2 <= RT_BLOCK &&
                      ctx->S.n_lights <= 256 &&   // pair record: 8-bit light index
                      std::max(ctx->cap, ctx->cap_deep) * (uint64_t)(ctx->S.n_shadow_lights > 0 ? ctx->S.n_shadow_lights : 1) < (1ull << RT_LIST_NU_SHIFT);   // list entry: pair | units << 28
    // tile beams (level 0 of a camera frame in 8 x 8 tiles: the rays of a block of 64 slots are the pixels of a tile, their beams
    // towards a light nearly one beam); the binning buffers were carved for the frame sizes this context has seen
    if (beam && cam && cam->tw * cam->th == 64 && (long long)cam->n_slots >= (early_cam ? ctx->env.tiles_min_early : ctx->env.tiles_min) && !lp.beam_early && !(p->flags & RT_FLAG_NO_TILES) && ctx->beam.trec != nullptr) {
        lp.tile_stride = (uint32_t)std::max(1, (cam->tiles_x + cam->world / 2) / std::max(cam->world, 1));
        const TileGrid G(cam->n_slots, lp.tile_stride);
        tiles = G.nblk <= ctx->tile_nblk && G.nreg <= ctx->tile_nreg && G.nzone <= ctx->tile_nzone;
        ctx->last_tiles = tiles;
    }
    if (s2 > 0 && s2 <= RT_BLOCK) {
        lp.spb = RT_BLOCK / s2;
        lp.beam = beam ? 1 : 0;
        lp.magic_s2 = (uint32_t)((0x100000000ull + s2 - 1) / s2);
        lp.magic_s = (uint32_t)((0x100000000ull + (uint32_t)sh - 1) / (uint32_t)sh);
    }
    // Both sides of an overlapped frame are fixed grids of resident workgroups: the level-0 side may hold at most ovl_main
    // (ovl_main2) of a CU's 4 workgroup slots (amdgpu_waves_per_eu(4, 4), 256 threads each), the deeper levels ovl_side - together
    // never more than the slots there are, so that neither side ever waits for the other to retire (no kernel here spins on
    // another one's results, but a side that cannot get a slot runs after the other instead of beside it).
    const bool overlap = sc.overlap;
    const int ovl_main = RT_OVL_MAIN, ovl_side = RT_OVL_SIDE, ovl_main2 = RT_OVL_MAIN;      // (side 3: the level-0 kernels after the beam set-up; more of them per CU was measured: no gain)
    // grid of a kernel with nb workgroups per CU at full occupancy: side 0 = sequential, 1 = level-0 side, 2 = deeper levels
    auto gsz = [&](int full, int nb, int side) {
        if (!overlap || side == 0) return full;
        int lim = side == 1 ? ovl_main : (side == 3 ? ovl_main2 : ovl_side);
        int use = nb < lim ? nb : lim;
        if (use < 1) use = 1;
        return (ctx->n_cu * use + 7) & ~7;
    };
    auto shadows = [&](int level, hipStream_t s_, RayQ &qc, HitBuf &hb, BeamBuf &bb, int side) -> int {
        set_level(level);
        if (ctx->S.n_shadow_lights <= 0) return RT_OK;
        if (beam) {
            {
                // the kernel template that fits the scene: long beams or not, boxes or spheres only
                const dim3 gs(gsz(ctx->g_beam_setup, ctx->nb_beam_setup, side));
                const bool boxes = ctx->n_boxes > 0;
                if (lp.tiles) {
                    // candidates per (block, light) by binning; small flat kernels in front of the pairs' own pass
                    // Small kernels of few registers: 8 workgroups per CU fit.  In an overlapped frame they keep to the level-0 side's
                    // share of the CUs all the same: the spawn and the first nearest-hit pass of the deeper levels start beside
                    // them, and a CU full of binning workgroups makes that chain wait (measured, 3 against 8 per CU: 1920x1080
                    // 1.67 / 1.74 ms, half of 3840x2160 2.82 / 2.97, a quarter 1.57 / 1.61; the whole 3840x2160 frame, where the
                    // level-0 side is the long one, 5.45 / 5.38 - there they take the CUs).
                    const int bw = (overlap && side == 1 && ctx->fb_slots_frame <= (4u << 20)) ? ovl_main : 8;
                    const dim3 gt((ctx->n_cu * bw + 7) & ~7);
                    if (!gen) hipLaunchKernelGGL(k_block_box, gt, block, 0, s_, ctx->S, lp, hb, bb, ctx->ctl);      // (else: k_intersect<.., true> wrote the boxes)
                    hipLaunchKernelGGL(k_bin_zone, gt, block, 0, s_, ctx->S, ctx->F, lp, bb, ctx->ctl);
                    hipLaunchKernelGGL(k_bin_tiles, gt, block, 0, s_, ctx->S, ctx->F, lp, bb, ctx->ctl);
                }
                if (time_bs) HIP_TRY(ctx, hipEventRecord(ctx->prof_bs[2 * level], s_));      // the frame's dominant kernel on its own (bench.py: roofline)
                if (lp.tiles) {
                    const dim3 gst(gsz(ctx->g_beam_setup_t, ctx->nb_beam_setup_t, side));
                    if (boxes) hipLaunchKernelGGL((k_beam_setup<false, true, true>), gst, block, ctx->lds_bytes, s_, ctx->S, ctx->F, lp, qc, hb, bb, ctx->ctl);
                    else hipLaunchKernelGGL((k_beam_setup<false, false, true>), gst, block, ctx->lds_bytes, s_, ctx->S, ctx->F, lp, qc, hb, bb, ctx->ctl);
                }
                else if (lp.beam_early && boxes) hipLaunchKernelGGL((k_beam_setup<true, true, false>), gs, block, ctx->lds_bytes, s_, ctx->S, ctx->F, lp, qc, hb, bb, ctx->ctl);
                else if (lp.beam_early) hipLaunchKernelGGL((k_beam_setup<true, false, false>), gs, block, ctx->lds_bytes, s_, ctx->S, ctx->F, lp, qc, hb, bb, ctx->ctl);
                else if (boxes) hipLaunchKernelGGL((k_beam_setup<false, true, false>), gs, block, ctx->lds_bytes, s_, ctx->S, ctx->F, lp, qc, hb, bb, ctx->ctl);
                else hipLaunchKernelGGL((k_beam_setup<false, false, false>), gs, block, ctx->lds_bytes, s_, ctx->S, ctx->F, lp, qc, hb, bb, ctx->ctl);
                if (time_bs) { HIP_TRY(ctx, hipEventRecord(ctx->prof_bs[2 * level + 1], s_)); ctx->prof_bs_levels = level + 1; }
            }
            hipLaunchKernelGGL(k_shadow_beam, dim3(gsz(ctx->g_beam, ctx->nb_beam, side == 1 ? 3 : side)), block, 0, s_, ctx->S, lp, hb, bb, ctx->ctl);
            // k_shadow_walk is launched for scenes of >= 100 k primitives only (where a level can hold millions of rays of
            // overflowing pairs; RT_AMD_WALK_MIN = 1 forces it): a rule of the scene, not of earlier frames.  Without the launch
            // k_shadow_queue is told that there is no walk (walk_min = never) and expands the overflowing pairs itself - the
            // same verdicts, one latency-bound launch less per level.
            LevelParams lq = lp;
            if (ctx->S.n_prims < 100000 && ctx->env.walk_min != 1) lq.walk_min = 0xffffffffu;
            if (lq.walk_min != 0xffffffffu)
                hipLaunchKernelGGL(k_shadow_walk, dim3(gsz(ctx->g_walk, ctx->nb_walk, side == 1 ? 3 : side)), block, ctx->lds_bytes, s_, ctx->S, ctx->F, lq, hb, bb, ctx->ctl);
            hipLaunchKernelGGL(k_shadow_queue, dim3(gsz(ctx->g_queue, ctx->nb_queue, side == 1 ? 3 : side)), block, ctx->lds_bytes, s_, ctx->S, ctx->F, lq, qc, hb, bb, ctx->ctl);
        } else {
            if (fast) hipLaunchKernelGGL(k_shadow<true>, dim3(gsz(ctx->g_shadow, ctx->nb_shadow, side)), block, ctx->lds_bytes, s_, ctx->S, ctx->F, lp, qc, hb, ctx->ctl);
            else hipLaunchKernelGGL(k_shadow<false>, dim3(ctx->g_shadow_x), block, 0, s_, ctx->S, ctx->F, lp, qc, hb, ctx->ctl);
        }
        return RT_OK;
    };
    // nearest hits of `level` (queue qc) + its hits' children into qn
    auto intersect = [&](int level, hipStream_t s_, RayQ &qc, RayQ &qn, HitBuf &hb, double *fbx, int side_, BeamBuf &bb) {
        uint32_t *bc = beam ? bb.queue_n : nullptr;
        set_level(level);
        static const GenParams none{};
        int32_t *ids = level == 0 ? d_ids : nullptr;
        TileOut to{nullptr, nullptr, 0u};
        if (level == 0 && gen) {
            // the camera-ray pass also zeroes the other framebuffers of the frame, and writes the block boxes of the tile beams
            double *side = fbx != ctx->fb ? ctx->fb : nullptr, *side2 = (fbm != ctx->fb && fbm != fbx) ? fbm : nullptr;
            if (lp.tiles) {
                const TileGrid G(gen->n_slots, lp.tile_stride);
                to.blockbox = bb.blockbox; to.zcnt = bb.zcnt; to.zrows = G.nzone * (uint32_t)ctx->S.n_shadow_lights;
            }
            if (fast) hipLaunchKernelGGL((k_intersect<true, true>), dim3(gsz(ctx->g_intersect, ctx->nb_intersect, side_)), block, ctx->lds_bytes + ctx->lds_top_bytes, s_, ctx->S, ctx->F, lp, qc, qn, hb, ctx->ctl, fbx, ids, *gen, bc, side, side2, to);
            else hipLaunchKernelGGL((k_intersect<false, true>), dim3(ctx->g_intersect_x), block, 0, s_, ctx->S, ctx->F, lp, qc, qn, hb, ctx->ctl, fbx, ids, *gen, bc, side, side2, to);
        } else if (fast) hipLaunchKernelGGL((k_intersect<true, false>), dim3(gsz(ctx->g_intersect, ctx->nb_intersect, side_)), block, ctx->lds_bytes + ctx->lds_top_bytes, s_, ctx->S, ctx->F, lp, qc, qn, hb, ctx->ctl, fbx, ids, none, bc, nullptr, nullptr, to);
        else hipLaunchKernelGGL((k_intersect<false, false>), dim3(ctx->g_intersect_x), block, 0, s_, ctx->S, ctx->F, lp, qc, qn, hb, ctx->ctl, fbx, ids, none, bc, nullptr, nullptr, to);
    };
    auto queue_of = [&](int level) -> RayQ & { return level == 0 ? ctx->q[0] : ctx->q[1 + ((level - 1) & 1)]; };

    // Small scene, few shadow rays per hit: one fused kernel per level over the LDS-resident scene (rt_small.hpp).  The
    // diagnostic flags keep selecting the general pipeline, which the parity tests compare this path with.
    const bool small = sc.small;
    if (small) {
        static const GenParams none{};
        for (int level = 0; level <= rec; ++level) {
            set_level(level);
            RayQ &qc = queue_of(level), &qn = queue_of(level + 1);
            if (level == 0 && gen)
                hipLaunchKernelGGL(k_small_level<true>, dim3(ctx->g_small), block, ctx->small.total, st, ctx->S, ctx->small, lp, qc, qn, ctx->ctl, fb0, fbm,
                                   ctx->fb, d_ids, *gen);
            else
                hipLaunchKernelGGL(k_small_level<false>, dim3(ctx->g_small), block, ctx->small.total, st, ctx->S, ctx->small, lp, qc, qn, ctx->ctl, ctx->fb, fbm,
                                   ctx->fb, level == 0 ? d_ids : nullptr, none);
        }
        HIP_TRY(ctx, hipGetLastError());
        return RT_OK;
    }

    if (!overlap) {
        for (int level = 0; level <= rec; ++level) {
            RayQ &qc = queue_of(level), &qn = queue_of(level + 1);
            HitBuf &hb = level == 0 ? ctx->hb : ctx->hb1;
            BeamBuf &bb = level == 0 ? ctx->beam : ctx->beam1;
            const int e = 2 + 6 * level;
            PROF_MARK(e + 0);
            intersect(level, st, qc, qn, hb, level == 0 ? fb0 : fbm, 0, bb);
            PROF_MARK(e + 1);
            PROF_MARK(e + 2);
            if (int rc = shadows(level, st, qc, hb, bb, 0)) return rc;
            PROF_MARK(e + 3);
            PROF_MARK(e + 4);
            if (level == 0 && fb0 != ctx->fb) hipLaunchKernelGGL((k_shade<RT_SHADE_LIGHTS | RT_SHADE_STORE>), dim3(ctx->g_shade), block, 0, st, ctx->S, lp, qc, hb, ctx->ctl, fb0, ctx->fb, bb.litbits, bb.pair_slot, bb.queue_n);
            else hipLaunchKernelGGL(k_shade<RT_SHADE_LIGHTS>, dim3(ctx->g_shade), block, 0, st, ctx->S, lp, qc, hb, ctx->ctl, level == 0 ? fb0 : ctx->fb, ctx->fb, bb.litbits, bb.pair_slot, bb.queue_n);
            PROF_MARK(e + 5);
        }
    } else {
        hipStream_t s2_ = ctx->stream2;
        if (early_cam) {
            // the camera pass on stream2, behind the nearest-hit chain of the frame before and beside that frame's tail on st
            // (enqueue_frame has ordered it behind what it overwrites)
            // (three workgroups per CU, like the level-0 kernels it runs beside: with all four the last, small kernels of the
            // frame before wait for a slot until the first workgroups of the pass retire - measured, 160 us for a 5 us kernel)
            intersect(0, s2_, ctx->q[0], ctx->q[1], ctx->hb, fb0, 1, ctx->beam);
            HIP_TRY(ctx, hipEventRecord(ctx->ev_cam, s2_));
            HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->ev_cam, 0));
            // the deeper levels' buffer sets and queues: the shading of the frame before is through with them
            if (prev_pipelined) HIP_TRY(ctx, hipStreamWaitEvent(s2_, ctx->ev_lit[prev_rec], 0));
        } else {
            intersect(0, st, ctx->q[0], ctx->q[1], ctx->hb, fb0, 0, ctx->beam);
            HIP_TRY(ctx, hipEventRecord(ctx->ev_fork, st));
            HIP_TRY(ctx, hipStreamWaitEvent(s2_, ctx->ev_fork, 0));
        }
        // level 0 (caller's stream) is submitted first: the host needs ~10 us per launch, and the long chain of the
        // deeper levels must not delay the submission of the frame's largest kernel
        if (int rc = shadows(0, st, ctx->q[0], ctx->hb, ctx->beam, 1)) return rc;
        hipLaunchKernelGGL((k_shade<RT_SHADE_LIGHTS | RT_SHADE_STORE>), dim3(gsz(ctx->g_shade, ctx->nb_shade, 3)), block, 0, st, ctx->S, lp, ctx->q[0], ctx->hb, ctx->ctl, fb0, ctx->fb,
                           ctx->beam.litbits, ctx->beam.pair_slot, ctx->beam.queue_n);
        // deeper levels (stream2): few, long rays - their waves ask for issue priority over the level-0 waves
        // (in frames of many million pixels the level-0 side is the long one and the priority only slows it down)
        lp.prio = ctx->fb_slots_frame <= (4u << 20) ? 1 : 0;
        if (sc.pipelined) {
            // Soft shadows of consecutive levels run side by side (odd levels on stream3, even levels on stream4: each is a latency
            // chain of its own); the Phong terms still reach fb in level order - shade(L) waits for shade(L - 1) - so the
            // float64 sums do not depend on the schedule.
            // A frame uses four streams, the caller's included - the hardware queues a process has here (GPU_MAX_HW_QUEUES = 4);
            // a fifth would share a hardware queue with one of the others and wait behind its kernels.  The even levels take
            // the context's own stream when the caller brought one, else a stream made on first use.
            hipStream_t s4_ = ctx->stream;
            if (st == ctx->stream) {
                if (!ctx->stream4) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->stream4, hipStreamNonBlocking));
                s4_ = ctx->stream4;
            }
            while ((int)ctx->ev_hit.size() <= rec) {
                hipEvent_t a, b;
                HIP_TRY(ctx, rt_event_create(&a, false));
                ctx->ev_hit.push_back(a);
                HIP_TRY(ctx, rt_event_create(&b, false));
                ctx->ev_lit.push_back(b);
            }
            // deeper levels rotate over nd = 2 or 4 buffer sets and nd + 1 ray queues
            const int nd = ctx->five_sets ? 4 : 2;
            auto pq = [&](int level) -> RayQ & { return ctx->q[1 + (level - 1) % (nd + 1)]; };
            for (int level = 1; level <= rec; ++level) {
                const bool odd = (level & 1) != 0;
                const int set = 1 + (level - 1) % nd;
                HitBuf &hb = *hit_set(*ctx, set);
                BeamBuf &bb = *beam_set(*ctx, set);
                RayQ &qc = pq(level), &qn = pq(level + 1);
                // set and queue of level - nd are reused: its shading must be through
                if (level > nd) HIP_TRY(ctx, hipStreamWaitEvent(s2_, ctx->ev_lit[level - nd], 0));
                intersect(level, s2_, qc, qn, hb, fbm, 2, bb);       // (+ the children: the next link of the frame's critical chain needs nothing else)
                HIP_TRY(ctx, hipEventRecord(ctx->ev_hit[level], s2_));
                hipStream_t s3_ = odd ? ctx->stream3 : s4_;
                HIP_TRY(ctx, hipStreamWaitEvent(s3_, ctx->ev_hit[level], 0));
                if (int rc = shadows(level, s3_, qc, hb, bb, 2)) return rc;
                if (level >= 2) HIP_TRY(ctx, hipStreamWaitEvent(s3_, ctx->ev_lit[level - 1], 0));
                hipLaunchKernelGGL(k_shade<RT_SHADE_LIGHTS>, dim3(gsz(ctx->g_shade, ctx->nb_shade, 2)), block, 0, s3_, ctx->S, lp, qc, hb, ctx->ctl, ctx->fb, ctx->fb,
                                   bb.litbits, bb.pair_slot, bb.queue_n);
                HIP_TRY(ctx, hipEventRecord(ctx->ev_lit[level], s3_));
            }
            lp.prio = 0;
            HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->ev_lit[rec], 0));      // shade(L) follows shade(L - 1): the last one covers all
            if (gen) {
                ctx->early_prev = true; ctx->early_st = st; ctx->early_prev_rec = rec;
                ctx->early_prev_q1 = 1 + ((rec - 1) / (nd + 1)) * (nd + 1);     // the last level whose queue is q[1]
            }
        } else {
            for (int level = 1; level <= rec; ++level) {
                RayQ &qc = queue_of(level), &qn = queue_of(level + 1);
                intersect(level, s2_, qc, qn, ctx->hb1, fbm, 2, ctx->beam1);
                if (int rc = shadows(level, s2_, qc, ctx->hb1, ctx->beam1, 2)) return rc;
                hipLaunchKernelGGL(k_shade<RT_SHADE_LIGHTS>, dim3(gsz(ctx->g_shade, ctx->nb_shade, 2)), block, 0, s2_, ctx->S, lp, qc, ctx->hb1, ctx->ctl, ctx->fb, ctx->fb,
                                   ctx->beam1.litbits, ctx->beam1.pair_slot, ctx->beam1.queue_n);
            }
            lp.prio = 0;
        }
        HIP_TRY(ctx, hipEventRecord(ctx->ev_join, s2_));
        HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->ev_join, 0));
    }
    HIP_TRY(ctx, hipGetLastError());
    return RT_OK;
}

// camera constants of a frame (ray_functions.py:17-25; host, once per frame)
static GenParams gen_params(const rt_ctx *ctx, const TileGeom &g) {
    GenParams gp{};
    const rt_scene_desc &s = ctx->host;
    double w = s.cam_screen_width;
    double h = w / g.W * g.H;
    gp.hg = h / g.H;
    gp.wg = w / g.W;
    for (int k = 0; k < 3; ++k) {
        gp.pos[k] = s.cam_pos[k]; gp.X[k] = s.cam_x[k]; gp.Y[k] = s.cam_y[k];
        double center = s.cam_pos[k] + s.cam_z[k] * s.cam_screen_distance;
        gp.p00[k] = (center + ((h - gp.hg) / 2 * s.cam_y[k])) - ((w - gp.wg) / 2 * s.cam_x[k]);
    }
    gp.W = g.W; gp.H = g.H; gp.tiles_x = g.tiles_x; gp.n_tiles = g.n_tiles; gp.rank = g.rank; gp.world = g.world;
    gp.tw = g.tw; gp.th = g.th; gp.n_slots = (uint32_t)g.slots;
    gp.rcp_tiles_x = 1.0f / (float)g.tiles_x;
    return gp;
}

// enqueue one camera frame on st; the frame state (queues, framebuffers, counters) is whatever ctx currently holds
static int enqueue_frame(rt_ctx *ctx, const rt_render_params *p, const TileGeom &g, int32_t sh, int32_t rec, float *d_rgb,
                         uint8_t *d_rgb8, int32_t *d_ids, hipStream_t st, bool prof) {
    ctx->fb_slots_frame = g.slots;
    const GenParams gp = gen_params(ctx, g);
    HIP_TRY(ctx, hipEventRecord(ctx->ev0, st));
    // Early camera pass (AltL0): a pipelined frame without primary ids (they are the caller's buffer, written by the pass) takes
    // the other copy of the level-0 state, and its camera pass goes to stream2.
    bool early_cam = false;
    // Measured (tools/b2b.py, frames back to back): -4 to -6 % for frames of up to 2 M camera rays whose two sides end together
    // (1280x720, a quarter or an eighth of 3840x2160, original.txt, pool.txt), -1 % at 1920x1080, +1 to +3 % from 4 M camera
    // rays on (the pass of a large frame takes more from the tail it runs beside than it gains): a rule of the frame's size.
    // Scenes of >= 100 k primitives keep the order "level 0's beam set-up takes the CUs first": their kernels are bound by
    // the latency of a tree that misses L2, and the deeper levels started ahead of it thrash it (1 M lattice: 22.1 -> 25.7 ms).
    // A frame enqueued while no other is pending (the caller has fetched the last one's status) has no tail to run beside
    // and takes the plain order, its camera pass alone on all four workgroup slots of a CU.
    // (pending: enqueued, status not fetched, and its last event not reached yet - a caller that synchronises the stream itself
    // gets the plain order as well.  The choice is one of schedule only: the frames are the same either way.)
    bool pending = false;
    if (ctx->frame_pending) {
        pending = hipEventQuery(ctx->ev1) == hipErrorNotReady;
        (void)hipGetLastError();                                     // ("not ready" is an answer, not an error of this call)
    }
    if (!prof && rec >= 2 && !d_ids && !ctx->env.no_early_cam && g.slots <= RT_EARLY_CAM_MAX_SLOTS && ctx->S.n_prims < 100000 && pending) {
        const Sched sc = plan_schedule(ctx, p, sh, rec, true);
        if (sc.pipelined && ensure_alt(ctx)) early_cam = true;
    }
    hipStream_t s_head = st;
    if (early_cam) {
        swap_alt(ctx);
        ++ctx->early_frames;
        s_head = ctx->stream2;
        // ... behind what it overwrites: this copy's last frame (its Phong kernel and resolve ran on that frame's stream),
        if (ctx->ev_end_set[ctx->alt_cur]) HIP_TRY(ctx, hipStreamWaitEvent(s_head, ctx->ev_end[ctx->alt_cur], 0));
        // and the readers of q[1], where the pass queues the children: the frame before, whose nearest-hit chain sits in front
        // of the pass on stream2 anyway - its soft shadows and Phong kernels of the levels on q[1] do not
        if (ctx->early_prev && ctx->early_st == st) HIP_TRY(ctx, hipStreamWaitEvent(s_head, ctx->ev_lit[ctx->early_prev_q1], 0));
        else {
            HIP_TRY(ctx, hipEventRecord(ctx->ev_prev, st));        // (no such frame: everything enqueued on st so far)
            HIP_TRY(ctx, hipStreamWaitEvent(s_head, ctx->ev_prev, 0));
        }
    }
    HIP_TRY(ctx, hipMemsetAsync(ctx->ctl, 0, sizeof(FrameCtl), s_head));
    if (prof) {      // otherwise k_intersect(0) initialises the framebuffers while it makes the camera rays
        HIP_TRY(ctx, hipMemsetAsync(ctx->fb, 0, g.out_slots * 3 * sizeof(double), st));
        HIP_TRY(ctx, hipMemsetAsync(ctx->fb0, 0, g.out_slots * 3 * sizeof(double), st));
        HIP_TRY(ctx, hipMemsetAsync(ctx->fbm, 0, g.out_slots * 3 * sizeof(double), st));
    }
    if (d_ids) HIP_TRY(ctx, hipMemsetAsync(d_ids, 0, g.out_slots * sizeof(int32_t), st));
    dim3 grid(ctx->grid), block(RT_BLOCK);
    if (rec < 0) {
        // ray_tracer.py:62-63: everything is background; expressed as a frame with no levels
        const rt_scene_desc &s = ctx->host;
        HIP_TRY(ctx, hipMemsetAsync(ctx->fb0, 0, g.out_slots * 3 * sizeof(double), st));
        HIP_TRY(ctx, hipMemsetAsync(ctx->fbm, 0, g.out_slots * 3 * sizeof(double), st));
        hipLaunchKernelGGL(k_fill_background, grid, block, 0, st, ctx->fb, g.out_slots, s.background[0], s.background[1], s.background[2]);
    } else {
        // camera rays: inside k_intersect(0); as a launch of their own only when the kernel families are timed
        PROF_MARK(0);
        if (prof) hipLaunchKernelGGL(k_gen, grid, block, 0, st, gp, ctx->q[0], ctx->ctl);
        PROF_MARK(1);
        if (int rc = enqueue_levels(ctx, p, sh, rec, d_ids, st, ctx->fb0, prof ? nullptr : &gp, ctx->fbm, &gp, early_cam)) return rc;
    }
    PROF_MARK(2 + 6 * (rec + 1));
    if (d_rgb || d_rgb8) hipLaunchKernelGGL(k_resolve, grid, block, 0, st, ctx->fb, ctx->fb0, ctx->fbm, g.out_slots * 3, d_rgb, d_rgb8, ctx->ctl, ctx->d_voided);
    else hipLaunchKernelGGL(k_frame_end, dim3(1), dim3(64), 0, st, ctx->ctl, ctx->d_voided);
    PROF_MARK(3 + 6 * (rec + 1));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_ctl, ctx->ctl, sizeof(FrameCtl), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipEventRecord(ctx->ev1, st));
    HIP_TRY(ctx, hipEventRecord(ctx->ev_end[ctx->alt_cur], st));      // (the copy of the level-0 state this frame ran on is free from here on)
    ctx->ev_end_set[ctx->alt_cur] = true;
    HIP_TRY(ctx, hipGetLastError());
    ctx->frame_pending = true;
    ctx->last = *p; ctx->last_sh = sh; ctx->last_rec = rec; ctx->last_slots = g.slots; ctx->last_gen = true;
    return RT_OK;
}

extern "C" int rt_render_device(rt_ctx *ctx, const rt_render_params *p, float *d_rgb, uint8_t *d_rgb8, int32_t *d_ids,
                                void *stream) {
    if (!ctx) return RT_ERR_INVALID_ARG;
    TileGeom g;
    std::string why;
    if (int rc = tile_geom(p, g, why)) return fail(ctx, rc, why);
    int32_t sh, rec;
    if (int rc = resolve_settings(ctx, p, sh, rec)) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    uint64_t need = g.slots > g.out_slots ? g.slots : g.out_slots;
    if (int rc = ensure_frame(ctx, need, ctx->S.n_shadow_lights > 0 ? sh : 0)) return rc;
    const bool prof = (p->flags & RT_FLAG_PROFILE) != 0 && rec >= 0;
    ctx->prof_levels = -1;
    if (prof) { if (int rc = prof_prepare(ctx, rec + 1)) return rc; }
    ctx->prof_bs_levels = 0;
    if ((prof || (p->flags & RT_FLAG_TIME_DOMINANT)) && rec >= 0) { if (int rc = prof_bs_prepare(ctx, rec + 1)) return rc; }
    ctx->last_noalloc = false;
    return enqueue_frame(ctx, p, g, sh, rec, d_rgb, d_rgb8, d_ids, st, prof);
}

// rt_render_device for callers that cannot allocate (include/rt_amd.h): the frame state lives in `scratch`, carved by the same
// routine that lays out the hipMalloc'ed buffers; levels run in sequence on ONE buffer set (the overlapped schedule needs two
// more sets and per-level events).  No hipMalloc / hipFree / event or stream creation on this path.  Every call on a device
// reuses that device's array from offset 0: what must start at zero (lit counters, append counters, the counter block) is
// cleared on the stream before the frame, so a voided or larger earlier frame leaves nothing behind.
extern "C" int rt_render_device_noalloc(rt_ctx *ctx, const rt_render_params *p, float *d_rgb, uint8_t *d_rgb8, int32_t *d_ids,
                                        void *stream) {
    if (!ctx) return RT_ERR_INVALID_ARG;
    TileGeom g;
    std::string why;
    if (int rc = tile_geom(p, g, why)) return fail(ctx, rc, why);
    int32_t sh, rec;
    if (int rc = resolve_settings(ctx, p, sh, rec)) return rc;
    if (p->flags & (RT_FLAG_PROFILE | RT_FLAG_TIME_DOMINANT)) return fail(ctx, RT_ERR_UNSUPPORTED, "RT_FLAG_PROFILE / RT_FLAG_TIME_DOMINANT create events: not available on the no-alloc entry point");
    ctx->prof_bs_levels = 0;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    unsigned char *base = nullptr;
    if (int rc = scratch_address(ctx, ctx->device, &base)) return rc;
    if (!base) return fail(ctx, RT_ERR_UNSUPPORTED, "no static scratch buffer on this device ordinal");
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    const uint64_t slots = g.slots > g.out_slots ? g.slots : g.out_slots;
    const uint64_t cap = 2 * slots + 1024;             // the allocating path's initial growth; it cannot grow here
    Carver mem{ctx};
    mem.base = base;
    mem.limit = RT_SCRATCH_BYTES;
    FrameState fs;
    if (int rc = mem.get(&fs.ctl, 1)) return rc;
    const uint32_t s2n = (uint32_t)(sh * sh);
    if (int rc = carve_frame(ctx, mem, fs, slots, cap, cap, 1.0, 1.0, (s2n >= 9u && s2n <= RT_BLOCK) ? RT_BLOCK / s2n : RT_BLOCK, 1, 3, false)) {
        char msg[256];
        snprintf(msg, sizeof(msg), "rt_render_device_noalloc: a frame of %llu pixel slots with %d shadow-casting lights does not fit the %llu MiB static scratch buffer",
                 (unsigned long long)slots, ctx->S.n_shadow_lights, (unsigned long long)(RT_SCRATCH_BYTES >> 20));
        return fail(ctx, rc, msg);
    }
    const FrameState saved = static_cast<FrameState &>(*ctx);
    static_cast<FrameState &>(*ctx) = fs;
    rt_render_params q = *p;
    q.flags |= RT_FLAG_NO_OVERLAP;
    int rc = init_frame_counters(ctx, fs, st);
    ctx->prof_levels = -1;
    if (rc == RT_OK) rc = enqueue_frame(ctx, &q, g, sh, rec, d_rgb, d_rgb8, d_ids, st, false);
    static_cast<FrameState &>(*ctx) = saved;            // the kernels hold their pointers by value
    // (enqueue_frame set frame_pending / last_* only when the whole frame was enqueued; a failed call leaves the context
    // describing the previous frame, growth policy included)
    if (rc == RT_OK) ctx->last_noalloc = true;
    return rc;
}

extern "C" int rt_sync_stats(rt_ctx *ctx, rt_stats *stats) {
    if (!ctx) return RT_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipEventSynchronize(ctx->ev1));
    ctx->frame_pending = false;
    const FrameCtl &c = *ctx->h_ctl;
    if (stats) {
        memset(stats, 0, sizeof(*stats));
        stats->primary_rays = (int64_t)c.primary_rays;
        stats->reflect_rays = (int64_t)c.reflect_rays;
        stats->transmit_rays = (int64_t)c.transmit_rays;
        int64_t hits = 0, levels = 0, maxq = 0;
        for (int l = 0; l <= ctx->last_rec && l < RT_MAX_LEVELS; ++l) {
            hits += c.n_hits[l];
            if (c.n_rays[l]) levels = l + 1;
            if ((int64_t)c.n_rays[l] > maxq) maxq = c.n_rays[l];
        }
        stats->hits = hits;
        stats->levels = levels;
        stats->max_queue = maxq;
        stats->shadow_rays = hits * (int64_t)ctx->S.n_shadow_lights * ctx->last_sh * ctx->last_sh;
        float ms = 0;
        if (hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1) == hipSuccess) stats->ms_total = ms;
        if (ctx->prof_levels >= 0) {
            auto span = [&](int a) {
                float t = 0;
                return hipEventElapsedTime(&t, ctx->prof_ev[a], ctx->prof_ev[a + 1]) == hipSuccess ? (double)t : 0.0;
            };
            stats->ms_gen = span(0);
            for (int l = 0; l < ctx->prof_levels; ++l) {
                stats->ms_intersect += span(2 + 6 * l);
                stats->ms_shadow += span(4 + 6 * l);
                stats->ms_shade += span(6 + 6 * l);
            }
            stats->ms_resolve = span(2 + 6 * ctx->prof_levels);
            stats->launches_per_family = ctx->prof_levels;
        }
        for (int l = 0; l < ctx->prof_bs_levels; ++l) {
            float t = 0;
            if (hipEventElapsedTime(&t, ctx->prof_bs[2 * l], ctx->prof_bs[2 * l + 1]) == hipSuccess) stats->ms_beam_setup += t;
        }
        stats->beam_setup_launches = ctx->prof_bs_levels;
    }
    if (ctx->env.debug && ctx->beam.rec && c.n_hits[0] > 0) {
        // beam path diagnostics: per level, pairs / pairs that needed the per-ray filter / rays that needed float64
        for (int l = 0; l <= ctx->last_rec && l < RT_MAX_LEVELS; ++l)
            fprintf(stderr, "[rt_amd] level %d: %llu pairs, %u listed, %u rays of overflowing pairs, %u rays queued by the filters\n", l,
                    (unsigned long long)c.n_hits[l] * (unsigned long long)ctx->S.n_shadow_lights, c.n_listed[l], c.n_walked[l], c.n_queued[l]);
#ifdef RT_DIAG
        fprintf(stderr, "[rt_amd] beam traversal trips (flag 0x1000): max %u, total %llu; pairs to the queue: delicate %u, touched %u, units %u, dedup %u\n",
                c.dbg_max_trips, c.dbg_sum_trips, c.dbg_why[0], c.dbg_why[1], c.dbg_why[2], c.dbg_why[3]);
        if (c.dbg_tile[2])
            fprintf(stderr, "[rt_amd] tile beams (flag 0x1000): %u (region, light) rows scanned %u zone-list entries and kept %u (%u overflowing); %u (block, light) rows list %u candidates (%u overflowing)\n",
                    c.dbg_tile[2], c.dbg_tile[1], c.dbg_tile[3], c.dbg_tile[4], c.dbg_tile[5], c.dbg_tile[6], c.dbg_tile[7]);
        if (c.dbg_bcyc[5])
            fprintf(stderr, "[rt_amd] k_beam_setup, all levels, clocks per (64 hits, light) trip of a wave (flag 0x4000, %llu trips): pair set-up %.0f, tile rows %.0f, tree walk %.0f, touched primitives %.0f, verdict + record %.0f\n",
                    c.dbg_bcyc[5], (double)c.dbg_bcyc[0] / c.dbg_bcyc[5], (double)c.dbg_bcyc[1] / c.dbg_bcyc[5], (double)c.dbg_bcyc[2] / c.dbg_bcyc[5],
                    (double)c.dbg_bcyc[3] / c.dbg_bcyc[5], (double)c.dbg_bcyc[4] / c.dbg_bcyc[5]);
        if (c.dbg_tcyc[3])
            fprintf(stderr, "[rt_amd] tile beams: longest zone list %u; clocks of thread 0 over all rows: set-up %llu, region filter %llu, block filter %llu; longest row %llu\n",
                    c.dbg_tile[0], c.dbg_tcyc[0], c.dbg_tcyc[1], c.dbg_tcyc[2], c.dbg_tcyc[3]);
#endif
        // candidate units of the listed pairs of the LAST level (render with rec_max = 0 to see level 0)
        uint32_t qn[4] = {0, 0, 0, 0};
        (void)hipMemcpy(qn, ctx->beam.queue_n, sizeof(qn), hipMemcpyDeviceToHost);
        uint32_t n = qn[1] < 200000u ? qn[1] : 200000u;
        std::vector<uint32_t> lst(n);
        if (n && hipMemcpy(lst.data(), ctx->beam.list, n * sizeof(uint32_t), hipMemcpyDeviceToHost) == hipSuccess) {
            uint64_t hist[RT_CUNITS + 2] = {0}, evalp = 0, base = 0;
            for (uint32_t i = 0; i < n; ++i) {
                float4 u1;
                if (lst[i] == RT_LIST_NONE) continue;
                if (hipMemcpy(&u1, ctx->beam.rec + (uint64_t)i * RT_PAIR_UNITS + 1, sizeof(u1), hipMemcpyDeviceToHost) != hipSuccess) break;
                uint32_t meta;
                memcpy(&meta, &u1.w, 4);
                uint32_t nu = meta & 0xf;
                hist[nu > RT_CUNITS ? RT_CUNITS + 1 : nu]++;
                evalp += ((meta >> 24) & 3u) == 0u;
                base += (meta >> 30) & 1u;
                if (i >= 20000) { n = i + 1; break; }
            }
            fprintf(stderr, "[rt_amd] last level, %u listed pairs sampled: units", n);
            for (int k = 0; k <= RT_CUNITS + 1; ++k) fprintf(stderr, " %d:%.1f%%", k, 100.0 * hist[k] / n);
            fprintf(stderr, " | plane0 per ray %.1f%% | own sphere at P %.1f%%\n", 100.0 * evalp / n, 100.0 * base / n);
        }
    }
    if (!c.overflow && ctx->last_tiles) ++ctx->tile_frames;
    ctx->last_tiles = false;
    if (c.overflow) {
        // the next frame gets larger queues: the one that ran out
        if (!ctx->last_noalloc) ctx->counters_dirty = true;
        if (ctx->last_noalloc) return fail(ctx, RT_ERR_QUEUE_OVERFLOW, "secondary-ray queue overflow in a no-alloc frame (its queues hold 2 x the pixel slots and cannot grow); nothing valid was written");
        if (c.overflow_r && ctx->rec_div <= 1.0) return fail(ctx, RT_ERR_HIP, "pair records ran out although every pair has one");
        if (c.overflow_r && ctx->rec_div > 1.0) ctx->rec_div = std::max(1.0, ctx->rec_div * 0.5);
        else if (c.overflow_q && ctx->beam_queue_cap < 0x7fffffffu) ctx->qgrowth *= 8.0;
        else if (c.overflow_deep && ctx->deep_div > 1.0) ctx->deep_div *= 0.5;
        else ctx->growth *= 2.0;
        return fail(ctx, RT_ERR_QUEUE_OVERFLOW, "secondary-ray queue overflow; queue growth factor doubled, render again");
    }
    // Pair records close to running out (frames of one scene differ by a few percent from seed to seed): give the next frames
    // room now, at this synchronisation point, rather than void one of them later in a loop that does not synchronise.
    if (!ctx->last_noalloc && ctx->rec_div_carved > 1.0 && ctx->rec_div >= ctx->rec_div_carved) {
        double worst = 0.0;
        for (int l = 0; l <= ctx->last_rec && l < RT_MAX_LEVELS; ++l) {
            const BeamBuf &bb = *beam_set(*ctx, (l == 0 || !ctx->two_sets) ? 0 : 1);
            if (bb.rec_cap) worst = std::max(worst, (double)c.n_listed[l] / (double)bb.rec_cap);
        }
        if (worst > 0.8) ctx->rec_div = std::max(1.0, ctx->rec_div_carved * 0.6 / worst);
    }
    return RT_OK;
}

// Device staging of the host-buffer entry points: one grow-only allocation per context, carved per call (these calls block,
// so one call's staging is free again when the next one starts).
struct Stage {
    rt_ctx *ctx;
    size_t used = 0;
    int reserve(size_t bytes) {
        if (ctx->stage_bytes >= bytes) return RT_OK;
        HIP_TRY(ctx, hipDeviceSynchronize());
        if (ctx->stage) (void)hipFree(ctx->stage);
        ctx->stage = nullptr; ctx->stage_bytes = 0;
        const size_t want = bytes + bytes / 4 + 4096;
        HIP_TRY(ctx, rt_malloc(&ctx->stage, want));
        ctx->stage_bytes = want;
        return RT_OK;
    }
    template <typename T> T *take(size_t n) {
        T *p = (T *)(ctx->stage + used);
        used += (n * sizeof(T) + 255) & ~(size_t)255;
        return p;
    }
    static size_t pad(size_t b) { return (b + 255) & ~(size_t)255; }
};

extern "C" int64_t rt_frame_bytes(const rt_ctx *ctx) { return ctx ? (int64_t)(ctx->frame_bytes + ctx->alt_bytes) : -1; }

extern "C" int rt_voided_frames(rt_ctx *ctx, int reset, int64_t *count) {
    if (!ctx || !count) return RT_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipDeviceSynchronize());
    uint32_t v = 0;
    HIP_TRY(ctx, hipMemcpy(&v, ctx->d_voided, sizeof(v), hipMemcpyDeviceToHost));
    if (reset) HIP_TRY(ctx, hipMemset(ctx->d_voided, 0, sizeof(v)));
    *count = (int64_t)v;
    return RT_OK;
}

extern "C" int rt_render(rt_ctx *ctx, const rt_render_params *p, float *rgb_out, uint8_t *rgb8_out, int32_t *ids_out,
                         rt_stats *stats) {
    if (!ctx) return RT_ERR_INVALID_ARG;
    TileGeom g;
    std::string why;
    if (int rc = tile_geom(p, g, why)) return fail(ctx, rc, why);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    Stage sg{ctx};
    const bool want8 = rgb8_out || !rgb_out;
    if (int rc = sg.reserve((rgb_out ? Stage::pad(g.out_slots * 3 * sizeof(float)) : 0) + (want8 ? Stage::pad(g.out_slots * 3) : 0) +
                            (ids_out ? Stage::pad(g.out_slots * sizeof(int32_t)) : 0))) return rc;
    float *d_rgb = rgb_out ? sg.take<float>(g.out_slots * 3) : nullptr;
    uint8_t *d_rgb8 = want8 ? sg.take<uint8_t>(g.out_slots * 3) : nullptr;
    int32_t *d_ids = ids_out ? sg.take<int32_t>(g.out_slots) : nullptr;
    int rc = RT_OK;
    for (int attempt = 0; attempt < 10; ++attempt) {      // a frame voided by RT_ERR_QUEUE_OVERFLOW is rendered again with the enlarged queues
        rc = rt_render_device(ctx, p, d_rgb, d_rgb8, d_ids, nullptr);
        if (rc != RT_OK) break;
        rc = rt_sync_stats(ctx, stats);
        if (rc != RT_ERR_QUEUE_OVERFLOW) break;
    }
    if (rc == RT_OK) {
        if (rgb_out) HIP_TRY(ctx, hipMemcpy(rgb_out, d_rgb, g.out_slots * 3 * sizeof(float), hipMemcpyDeviceToHost));
        if (rgb8_out) HIP_TRY(ctx, hipMemcpy(rgb8_out, d_rgb8, g.out_slots * 3, hipMemcpyDeviceToHost));
        if (ids_out) HIP_TRY(ctx, hipMemcpy(ids_out, d_ids, g.out_slots * sizeof(int32_t), hipMemcpyDeviceToHost));
    }
    return rc;
}

extern "C" int rt_render_mean(rt_ctx *ctx, const rt_render_params *p, int32_t frames, float *rgb_out, uint8_t *rgb8_out,
                              rt_stats *stats) {
    if (!ctx) return RT_ERR_INVALID_ARG;
    if (frames < 1 || frames > 65536) return fail(ctx, RT_ERR_INVALID_ARG, "frames must be 1..65536");
    TileGeom g;
    std::string why;
    if (int rc = tile_geom(p, g, why)) return fail(ctx, rc, why);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ctx->acc_slots < g.out_slots) {
        if (ctx->acc) (void)hipFree(ctx->acc);
        ctx->acc = nullptr;
        ctx->acc_slots = 0;
        HIP_TRY(ctx, rt_malloc(&ctx->acc, g.out_slots * 3 * sizeof(double)));
        ctx->acc_slots = g.out_slots;
    }
    Stage sg{ctx};
    if (int rc0 = sg.reserve((rgb_out ? Stage::pad(g.out_slots * 3 * sizeof(float)) : 0) + (rgb8_out ? Stage::pad(g.out_slots * 3) : 0))) return rc0;
    float *d_rgb = rgb_out ? sg.take<float>(g.out_slots * 3) : nullptr;
    uint8_t *d_rgb8 = rgb8_out ? sg.take<uint8_t>(g.out_slots * 3) : nullptr;
    int rc = RT_OK;
    auto cleanup = [&]() {};
#define TRY_M(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { cleanup(); return fail(ctx, e_ == hipErrorOutOfMemory ? RT_ERR_OOM : RT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } } while (0)
    rt_stats sum;
    memset(&sum, 0, sizeof(sum));
    for (int32_t k = 0; k < frames && rc == RT_OK; ++k) {
        rt_render_params q = *p;
        q.seed = p->seed + (uint32_t)k;                       // frame k of the mean = the frame rt_render gives for seed + k
        rt_stats st;
        for (int attempt = 0; attempt < 10; ++attempt) {
            rc = rt_render_device(ctx, &q, nullptr, nullptr, nullptr, nullptr);
            if (rc != RT_OK) break;
            rc = rt_sync_stats(ctx, &st);
            if (rc != RT_ERR_QUEUE_OVERFLOW) break;
        }
        if (rc != RT_OK) break;
        const bool last = k + 1 == frames;
        hipLaunchKernelGGL(k_accumulate, dim3(ctx->grid), dim3(RT_BLOCK), 0, ctx->stream, ctx->fb, ctx->fb0, ctx->fbm, ctx->acc,
                           g.out_slots * 3, k == 0 ? 1 : 0, last ? (double)frames : 0.0, d_rgb, d_rgb8);
        sum.primary_rays += st.primary_rays; sum.shadow_rays += st.shadow_rays; sum.reflect_rays += st.reflect_rays;
        sum.transmit_rays += st.transmit_rays; sum.hits += st.hits; sum.ms_total += st.ms_total;
        if (st.levels > sum.levels) sum.levels = st.levels;
        if (st.max_queue > sum.max_queue) sum.max_queue = st.max_queue;
    }
    if (rc == RT_OK) {
        TRY_M(hipStreamSynchronize(ctx->stream));
        TRY_M(hipGetLastError());
        if (rgb_out) TRY_M(hipMemcpy(rgb_out, d_rgb, g.out_slots * 3 * sizeof(float), hipMemcpyDeviceToHost));
        if (rgb8_out) TRY_M(hipMemcpy(rgb8_out, d_rgb8, g.out_slots * 3, hipMemcpyDeviceToHost));
        if (stats) *stats = sum;
    }
#undef TRY_M
    cleanup();
    return rc;
}

extern "C" int rt_trace(rt_ctx *ctx, const rt_render_params *p, const double *origins, const double *directions,
                        int64_t n, double *rgb_out, rt_stats *stats) {
    if (!ctx) return RT_ERR_INVALID_ARG;
    if (!p || !origins || !directions || !rgb_out || n < 0) return fail(ctx, RT_ERR_INVALID_ARG, "NULL argument or negative n");
    if (n == 0) return RT_OK;
    if ((uint64_t)n > 0xFFFFFFF0ull / 4) return fail(ctx, RT_ERR_UNSUPPORTED, "too many rays in one batch");
    int32_t sh, rec;
    if (int rc = resolve_settings(ctx, p, sh, rec)) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    int rc = RT_OK;
    auto cleanup = [&]() {};
#define TRY_T(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { cleanup(); return fail(ctx, e_ == hipErrorOutOfMemory ? RT_ERR_OOM : RT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } } while (0)
    Stage sg{ctx};
    if (int rc0 = sg.reserve(2 * Stage::pad((size_t)n * 3 * sizeof(double)))) return rc0;
    double *d_o = sg.take<double>((size_t)n * 3), *d_d = sg.take<double>((size_t)n * 3);
    TRY_T(hipMemcpy(d_o, origins, (size_t)n * 3 * sizeof(double), hipMemcpyHostToDevice));
    TRY_T(hipMemcpy(d_d, directions, (size_t)n * 3 * sizeof(double), hipMemcpyHostToDevice));
    for (int attempt = 0; attempt < 10; ++attempt) {
        rc = ensure_frame(ctx, (uint64_t)n, ctx->S.n_shadow_lights > 0 ? sh : 0);
        if (rc != RT_OK) break;
        TRY_T(hipEventRecord(ctx->ev0, st));
        TRY_T(hipMemsetAsync(ctx->ctl, 0, sizeof(FrameCtl), st));
        TRY_T(hipMemsetAsync(ctx->fb, 0, (size_t)n * 3 * sizeof(double), st));
        if (rec < 0) {     // ray_tracer.py:62-63 (on the frame's stream, after the memset above)
            hipLaunchKernelGGL(k_fill_background, dim3(ctx->grid), dim3(RT_BLOCK), 0, st, ctx->fb, (uint64_t)n, ctx->host.background[0],
                               ctx->host.background[1], ctx->host.background[2]);
        } else {
            hipLaunchKernelGGL(k_load_rays, dim3(ctx->grid), dim3(RT_BLOCK), 0, st, d_o, d_d, (uint32_t)n, ctx->q[0], ctx->ctl);
            rt_render_params pp = *p;
            pp.flags &= ~(RT_FLAG_PROFILE | RT_FLAG_TIME_DOMINANT);
            ctx->prof_bs_levels = 0;
            ctx->prof_levels = -1;
            rc = enqueue_levels(ctx, &pp, sh, rec, nullptr, st, ctx->fb, nullptr, ctx->fb, nullptr);   // one framebuffer: rgb_out is its raw float64 content
            if (rc != RT_OK) break;
        }
        hipLaunchKernelGGL(k_frame_end, dim3(1), dim3(64), 0, st, ctx->ctl, ctx->d_voided);
        TRY_T(hipMemcpyAsync(ctx->h_ctl, ctx->ctl, sizeof(FrameCtl), hipMemcpyDeviceToHost, st));
        TRY_T(hipEventRecord(ctx->ev1, st));
        ctx->last = *p; ctx->last_sh = sh; ctx->last_rec = rec; ctx->last_slots = (uint64_t)n; ctx->last_gen = false; ctx->last_noalloc = false;
        rc = rt_sync_stats(ctx, stats);
        if (rc != RT_ERR_QUEUE_OVERFLOW) break;
    }
    if (rc == RT_OK) TRY_T(hipMemcpy(rgb_out, ctx->fb, (size_t)n * 3 * sizeof(double), hipMemcpyDeviceToHost));
#undef TRY_T
    cleanup();
    return rc;
}

// BSPNode.intersect_vectorize on device buffers: only enqueued on `stream` (NULL = the context's), nothing is synchronised
extern "C" int rt_intersect_device(rt_ctx *ctx, const double *d_origins, const double *d_directions, int64_t n, double *d_points_out,
                                   int64_t *d_ids_out, void *stream) {
    if (!ctx) return RT_ERR_INVALID_ARG;
    if (!d_origins || !d_directions || !d_points_out || !d_ids_out || n < 0) return fail(ctx, RT_ERR_INVALID_ARG, "NULL argument or negative n");
    if (n == 0) return RT_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    if (ctx->F.ok)
        hipLaunchKernelGGL(k_intersect_only<true>, dim3(ctx->grid), dim3(RT_BLOCK), ctx->lds_bytes, st, ctx->S, ctx->F, d_origins, d_directions, (uint64_t)n, d_points_out, d_ids_out);
    else
        hipLaunchKernelGGL(k_intersect_only<false>, dim3(ctx->grid), dim3(RT_BLOCK), 0, st, ctx->S, ctx->F, d_origins, d_directions, (uint64_t)n, d_points_out, d_ids_out);
    HIP_TRY(ctx, hipGetLastError());
    return RT_OK;
}

extern "C" int rt_intersect(rt_ctx *ctx, const double *origins, const double *directions, int64_t n, double *points_out,
                            int64_t *ids_out) {
    if (!ctx) return RT_ERR_INVALID_ARG;
    if (!origins || !directions || !points_out || !ids_out || n < 0) return fail(ctx, RT_ERR_INVALID_ARG, "NULL argument or negative n");
    if (n == 0) return RT_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t vb = (size_t)n * 3 * sizeof(double);
    Stage sg{ctx};
    if (int rc = sg.reserve(3 * Stage::pad(vb) + Stage::pad((size_t)n * sizeof(int64_t)))) return rc;
    double *d_o = sg.take<double>((size_t)n * 3), *d_d = sg.take<double>((size_t)n * 3), *d_p = sg.take<double>((size_t)n * 3);
    int64_t *d_i = sg.take<int64_t>((size_t)n);
    HIP_TRY(ctx, hipMemcpyAsync(d_o, origins, vb, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(d_d, directions, vb, hipMemcpyHostToDevice, ctx->stream));
    if (int rc = rt_intersect_device(ctx, d_o, d_d, n, d_p, d_i, ctx->stream)) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(points_out, d_p, vb, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ids_out, d_i, (size_t)n * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

// the any-hit query on device buffers (see rt_visibility): only enqueued on `stream`
extern "C" int rt_visibility_device(rt_ctx *ctx, const double *d_origins, const double *d_directions, const double *d_targets,
                                    int64_t n, uint8_t *d_lit_out, void *stream) {
    if (!ctx) return RT_ERR_INVALID_ARG;
    if (!d_origins || !d_directions || !d_targets || !d_lit_out || n < 0) return fail(ctx, RT_ERR_INVALID_ARG, "NULL argument or negative n");
    if (n == 0) return RT_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    if (ctx->F.ok)
        hipLaunchKernelGGL(k_visibility<true>, dim3(ctx->grid), dim3(RT_BLOCK), ctx->lds_bytes, st, ctx->S, ctx->F, d_origins, d_directions, d_targets, (uint64_t)n, d_lit_out);
    else
        hipLaunchKernelGGL(k_visibility<false>, dim3(ctx->grid), dim3(RT_BLOCK), 0, st, ctx->S, ctx->F, d_origins, d_directions, d_targets, (uint64_t)n, d_lit_out);
    HIP_TRY(ctx, hipGetLastError());
    return RT_OK;
}

extern "C" int rt_visibility(rt_ctx *ctx, const double *origins, const double *directions, const double *targets,
                             int64_t n, uint8_t *lit_out) {
    if (!ctx) return RT_ERR_INVALID_ARG;
    if (!origins || !directions || !targets || !lit_out || n < 0) return fail(ctx, RT_ERR_INVALID_ARG, "NULL argument or negative n");
    if (n == 0) return RT_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t vn = (size_t)n * 3, vb = vn * sizeof(double);
    Stage sg{ctx};
    if (int rc = sg.reserve(3 * Stage::pad(vb) + Stage::pad((size_t)n))) return rc;
    double *d_o = sg.take<double>(vn), *d_d = sg.take<double>(vn), *d_t = sg.take<double>(vn);
    uint8_t *d_l = sg.take<uint8_t>((size_t)n);
    HIP_TRY(ctx, hipMemcpyAsync(d_o, origins, vb, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(d_d, directions, vb, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(d_t, targets, vb, hipMemcpyHostToDevice, ctx->stream));
    if (int rc = rt_visibility_device(ctx, d_o, d_d, d_t, n, d_l, ctx->stream)) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(lit_out, d_l, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

extern "C" int rt_deinterleave_u8(rt_ctx *ctx, const rt_render_params *p, const uint8_t *d_packed, int64_t stride_bytes,
                                  uint8_t *d_image, void *stream) {
    if (!ctx) return RT_ERR_INVALID_ARG;
    TileGeom g;
    std::string why;
    if (int rc = tile_geom(p, g, why)) return fail(ctx, rc, why);
    if (!d_packed || !d_image) return fail(ctx, RT_ERR_INVALID_ARG, "NULL buffer");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    hipLaunchKernelGGL(k_deinterleave_u8, dim3(ctx->grid), dim3(RT_BLOCK), 0, st, d_packed, stride_bytes, d_image, g.W, g.H,
                       g.tw, g.th, g.tiles_x, g.n_tiles, g.world, g.tiles_per_rank);
    HIP_TRY(ctx, hipGetLastError());
    return RT_OK;
}

// ---- unit entry points: one primitive / one box against a ray batch (kernel-level parity) ----
static int unit_run(int device, const double *in_a, const double *in_b, int64_t n, size_t out_elems_per_ray,
                    void (*launch)(const double *, const double *, uint64_t, void *, void *, const void *),
                    const void *extra, void *out_a, size_t out_a_bytes_per_ray, void *out_b, size_t out_b_bytes_per_ray) {
    (void)out_elems_per_ray;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return fail(nullptr, RT_ERR_NO_DEVICE, "no HIP device");
    if (n <= 0) return RT_OK;
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, RT_ERR_HIP, "hipSetDevice failed");
    double *d_a = nullptr, *d_b = nullptr;
    void *d_oa = nullptr, *d_ob = nullptr;
    size_t vb = (size_t)n * 3 * sizeof(double);
    int rc = RT_OK;
    hipError_t e = hipSuccess;
    if (in_a && (e = rt_malloc(&d_a, vb)) == hipSuccess) e = hipMemcpy(d_a, in_a, vb, hipMemcpyHostToDevice);
    if (e == hipSuccess && in_b && (e = rt_malloc(&d_b, vb)) == hipSuccess) e = hipMemcpy(d_b, in_b, vb, hipMemcpyHostToDevice);
    if (e == hipSuccess && out_a) e = rt_malloc(&d_oa, (size_t)n * out_a_bytes_per_ray);
    if (e == hipSuccess && out_b) e = rt_malloc(&d_ob, (size_t)n * out_b_bytes_per_ray);
    if (e == hipSuccess) {
        launch(d_a, d_b, (uint64_t)n, d_oa, d_ob, extra);
        e = hipDeviceSynchronize();
    }
    if (e == hipSuccess && out_a) e = hipMemcpy(out_a, d_oa, (size_t)n * out_a_bytes_per_ray, hipMemcpyDeviceToHost);
    if (e == hipSuccess && out_b) e = hipMemcpy(out_b, d_ob, (size_t)n * out_b_bytes_per_ray, hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = fail(nullptr, RT_ERR_HIP, std::string("unit kernel: ") + hipGetErrorString(e));
    if (d_a) (void)hipFree(d_a);
    if (d_b) (void)hipFree(d_b);
    if (d_oa) (void)hipFree(d_oa);
    if (d_ob) (void)hipFree(d_ob);
    return rc;
}

struct UnitExtra {
    int kind;
    int outside_only;
    double *d_par;
};

static double *unit_params(const double *par, int n) {
    double *d = nullptr;
    if (rt_malloc(&d, n * sizeof(double)) != hipSuccess) return nullptr;
    if (hipMemcpy(d, par, n * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); return nullptr; }
    return d;
}

extern "C" int rt_unit_prim_intersect(int device, int kind, const double *params12, const double *origins,
                                      const double *directions, int64_t n, double *points_out) {
    if (!params12 || !origins || !directions || !points_out) return fail(nullptr, RT_ERR_INVALID_ARG, "NULL argument");
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, RT_ERR_NO_DEVICE, "no HIP device");
    UnitExtra ex{kind, 0, unit_params(params12, 12)};
    if (!ex.d_par) return fail(nullptr, RT_ERR_HIP, "parameter upload failed");
    int rc = unit_run(device, origins, directions, n, 3,
                      [](const double *o, const double *d, uint64_t nn, void *oa, void *, const void *x) {
                          const UnitExtra *e = (const UnitExtra *)x;
                          hipLaunchKernelGGL(k_unit_prim, dim3(256), dim3(256), 0, 0, e->kind, e->d_par, o, d, nn, (double *)oa);
                      },
                      &ex, points_out, 3 * sizeof(double), nullptr, 0);
    (void)hipFree(ex.d_par);
    return rc;
}

extern "C" int rt_unit_prim_normals(int device, int kind, const double *params12, const double *points, int64_t n,
                                    double *normals_out) {
    if (!params12 || !points || !normals_out) return fail(nullptr, RT_ERR_INVALID_ARG, "NULL argument");
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, RT_ERR_NO_DEVICE, "no HIP device");
    UnitExtra ex{kind, 0, unit_params(params12, 12)};
    if (!ex.d_par) return fail(nullptr, RT_ERR_HIP, "parameter upload failed");
    int rc = unit_run(device, points, nullptr, n, 3,
                      [](const double *pts, const double *, uint64_t nn, void *oa, void *, const void *x) {
                          const UnitExtra *e = (const UnitExtra *)x;
                          hipLaunchKernelGGL(k_unit_normals, dim3(256), dim3(256), 0, 0, e->kind, e->d_par, pts, nn, (double *)oa);
                      },
                      &ex, normals_out, 3 * sizeof(double), nullptr, 0);
    (void)hipFree(ex.d_par);
    return rc;
}

extern "C" int rt_unit_slab_mask(int device, const double *lo, const double *hi, int outside_only, const double *origins,
                                 const double *directions, int64_t n, uint8_t *mask_out, double *t_near_out) {
    if (!lo || !hi || !origins || !directions || !mask_out || !t_near_out) return fail(nullptr, RT_ERR_INVALID_ARG, "NULL argument");
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, RT_ERR_NO_DEVICE, "no HIP device");
    double par[6] = {lo[0], lo[1], lo[2], hi[0], hi[1], hi[2]};
    UnitExtra ex{0, outside_only, unit_params(par, 6)};
    if (!ex.d_par) return fail(nullptr, RT_ERR_HIP, "parameter upload failed");
    int rc = unit_run(device, origins, directions, n, 1,
                      [](const double *o, const double *d, uint64_t nn, void *oa, void *ob, const void *x) {
                          const UnitExtra *e = (const UnitExtra *)x;
                          hipLaunchKernelGGL(k_unit_slab, dim3(256), dim3(256), 0, 0, e->d_par, e->d_par + 3, e->outside_only, o, d, nn,
                                             (uint8_t *)oa, (double *)ob);
                      },
                      &ex, mask_out, 1, t_near_out, sizeof(double));
    (void)hipFree(ex.d_par);
    return rc;
}

// get_initial_rays (ray_functions.py:7-43) as a standalone query: sources/directions [W*H*3] float64, row-major
extern "C" int rt_initial_rays(rt_ctx *ctx, int width, int height, double *sources_out, double *directions_out) {
    if (!ctx) return RT_ERR_INVALID_ARG;
    if (width <= 0 || height <= 0 || !sources_out || !directions_out) return fail(ctx, RT_ERR_INVALID_ARG, "bad argument");
    rt_render_params p{};
    p.width = width; p.height = height; p.world = 1; p.tile_w = 8; p.tile_h = 8;
    TileGeom g;
    std::string why;
    if (int rc = tile_geom(&p, g, why)) return fail(ctx, rc, why);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (int rc = ensure_frame(ctx, g.slots > g.out_slots ? g.slots : g.out_slots, 0)) return rc;
    GenParams gp{};
    const rt_scene_desc &s = ctx->host;
    double w = s.cam_screen_width;
    double h = w / g.W * g.H;
    gp.hg = h / g.H;
    gp.wg = w / g.W;
    for (int k = 0; k < 3; ++k) {
        gp.pos[k] = s.cam_pos[k]; gp.X[k] = s.cam_x[k]; gp.Y[k] = s.cam_y[k];
        double center = s.cam_pos[k] + s.cam_z[k] * s.cam_screen_distance;
        gp.p00[k] = (center + ((h - gp.hg) / 2 * s.cam_y[k])) - ((w - gp.wg) / 2 * s.cam_x[k]);
    }
    gp.W = g.W; gp.H = g.H; gp.tiles_x = g.tiles_x; gp.n_tiles = g.n_tiles; gp.rank = 0; gp.world = 1;
    gp.tw = g.tw; gp.th = g.th; gp.n_slots = (uint32_t)g.slots;
    gp.rcp_tiles_x = 1.0f / (float)g.tiles_x;
    hipStream_t st = ctx->stream;
    HIP_TRY(ctx, hipMemsetAsync(ctx->ctl, 0, sizeof(FrameCtl), st));
    hipLaunchKernelGGL(k_gen, dim3(ctx->grid), dim3(RT_BLOCK), 0, st, gp, ctx->q[0], ctx->ctl);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(st));
    uint64_t ns = g.slots;
    std::vector<double> buf(ns * 6);
    std::vector<uint32_t> slot(ns);
    RayQ &q = ctx->q[0];
    double *src[6] = {q.ox, q.oy, q.oz, q.dx, q.dy, q.dz};
    for (int c = 0; c < 6; ++c) HIP_TRY(ctx, hipMemcpy(buf.data() + c * ns, src[c], ns * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(slot.data(), q.slot, ns * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (uint64_t k = 0; k < ns; ++k) {
        if (slot[k] == RT_INVALID_SLOT) continue;
        for (int c = 0; c < 3; ++c) {
            sources_out[3ull * slot[k] + c] = buf[c * ns + k];
            directions_out[3ull * slot[k] + c] = buf[(3 + c) * ns + k];
        }
    }
    return RT_OK;
}
