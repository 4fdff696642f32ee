// rt_shadow.hpp - soft shadows as filtered predicates (Light.compute_shadows, Light.py:178-227).
//
// The reference decides a shadow ray by  lit <=> |X - P| < 1e-6  where X is the nearest hit of the
// ray from the light sample to the shaded point P.  Seen from P (tau = distance from P toward the
// sample), every primitive the line meets has a point tau_out where the light ray ENTERS it; X is
// the entry with the largest tau_out below the sample, so
//
//     some primitive has  tau_out >  1e-6                      -> unlit   ("OCCLUDE")
//     all primitives have tau_out < -1e-6 or miss the line,
//       and one has |tau_out| < 1e-6                           -> lit     ("AT_P": P's own surface)
//
// All s*s rays of one (hit, light) pair end in the same P and start inside one small box around the
// light, so (1) ONE conservative beam traversal per pair (k_beam_setup) collects the few
// primitives any of its rays can touch, with their coordinates taken RELATIVE TO P in float64 and
// only then rounded to float32 - the own surface sits at ~1e-16, not at an ulp of a world
// coordinate; (2) each ray classifies those candidates in float32 with explicit error margins
// (k_shadow_beam).  A verdict is only taken when every margin is cleared; anything else (grazing
// silhouettes, |w.n| ~ 0, sample inside a primitive, candidate-list overflow) is appended to a
// queue and decided by the float64 path of rt_fast.hpp (k_shadow_queue).  So the outcome is that
// of the float64 reference-order evaluation, at float32 VALU cost for ~99.9 % of the rays.
#pragma once
#include <hip/hip_fp16.h>
#include "rt_fast.hpp"

#ifdef RT_DIAG
#define RT_DIAG_ONLY(...) __VA_ARGS__
#else
#define RT_DIAG_ONLY(...)
#endif
#define RT_CUNITS 12         // 16-byte candidate units kept per (hit, light) pair: a sphere takes 1, a box 2
#define RT_CMAX RT_CUNITS
#define RT_TOUCH 8          // leaf entries a pair's traversal may park before its rays go to the float64 queue
#define RT_PMAX 3            // planes handled by the filtered path
#define RT_PAIR_UNITS (4 + RT_CUNITS)   // one pair record = 16 float4 = 256 B, contiguous; only 4 + units used are read

// Pair record (float4 units), coordinates relative to P:
//   u0 = {q0.xyz (p00 - P, Light.py:260), g = radius / s}
//   u1 = {up.xyz, meta}      meta: bits 0-7 candidate units used (0xFF = overflow: every ray of the pair goes to
//                                   the queue), bits 8-15 bit j set = unit j starts a box, bits 16-31 light index
//   u2 = {right.xyz, pix}    pix / path: jitter-hash keys of the ray that produced the hit
//   u3 = {pd[0..2], path}    plane k: n_k . P + d_k (from float64; ~1e-16 on P's own plane)
//   u4.. candidates          sphere: {centre - P, |centre - P|^2 - r^2 (float64, then rounded)}
//                            box:    {lo - P rounded down, hi.x - P.x rounded up}, {hi.y, hi.z rounded up, 0, 0}
struct CandF {               // decoded candidate
    float a[3];
    float c;
    float b[3];
    uint32_t kind;
};
struct BeamBuf {
    // Records live in SLOTS, handed out to the pairs that need one (listed pairs: header + candidate units; overflowing pairs:
    // the header, for k_shadow_walk) - rec_cap of them, a fraction of the pairs (13 % of the pairs of the 10k-sphere scene are
    // listed); a level that needs more raises FrameCtl::overflow_r and the frame is rendered again with more.  Slots are taken
    // 64 at a time per wave (one returning atomic on queue_n[1]) and used up in order, so list[] is dense up to the unused tail
    // of each wave's last block (RT_LIST_NONE there, and at the slots of overflowing pairs).
    float4 *rec;             // [slot][RT_PAIR_UNITS]
    unsigned long long *queue;   // shadow-ray numbers g the filter could not decide
    uint32_t *queue_n;       // [0] = queue entries appended, [1] = slots handed out, [2] = pairs in ovlist, [3] = queue capacity
    unsigned long long *ovlist;  // pairs whose rays all go to the float64 kernel (candidate overflow, delicate basis): pair | slot << 32
    uint32_t *list;          // [slot] -> pair | units used << 28 of a listed pair (its rays need the per-ray filter), or RT_LIST_NONE
    uint32_t *pair_slot;     // [pair] -> slot, or RT_PAIR_LIT / RT_PAIR_UNLIT when the whole pair is decided
    uint32_t rec_cap;        // slots in rec / list (a multiple of 64)
    uint32_t rec_blk;        // slots a wave of k_beam_setup takes at a time (8 .. 64)
    unsigned long long *litbits; // [chunk of slots][wave]: which lanes' rays the filter found lit (every word written per level)
    // tile beams (level 0 of scenes with short beams; set 0 only): candidates per (block of 64 ray slots, shadow light), found by
    // binning the primitives against the union of the block's beams, zone -> region -> block (k_bin_*, rt_amd.hip)
    float *blockbox;         // [block][8]: min xyz, max xyz of the block's hit points, rounded outward (min > max: no hit)
    float4 *trec;            // [block * n_shadow_lights + light][2 * RT_TILE_NC]: the row's candidates, as their 32-byte leaf records
    uint32_t *tcnt;          // [block * n_shadow_lights + light]: candidates in the row, or RT_TILE_OVERFLOW
    uint32_t *zlist, *zcnt;  // [zone * n_shadow_lights + light][min(n_prims, RT_ZONE_NC)]: canonical leaf entries of the primitives near the zone's union beam / their number
};
#define RT_TILE_NC 64u                  // candidates a (block, light) row holds: 2 KB
#define RT_TILE_HALF 32u                // ... and a wave stages at a time: 1 KB, one float4 per lane
#define RT_REGION_NC 512u               // ... a (region, light) keeps in LDS
#define RT_ZONE_NC 131072u              // ... a (zone, light) row (or every primitive of a smaller scene)
#define RT_REGION_BLOCKS 16u            // a region = 4 x 4 blocks (32 x 32 pixels), a zone = 64 regions
#define RT_TILE_OVERFLOW 0xFFFFFFFFu    // more candidates than the row holds: the pairs concerned walk the tree themselves
#define RT_LIST_NONE 0xFFFFFFFFu
#define RT_PAIR_LIT 0xFFFFFFFFu
#define RT_PAIR_UNLIT 0xFFFFFFFEu
#define RT_META_OVERFLOW 0xFu
#define RT_LIST_NU_SHIFT 28   // bb.list entry = pair | units used << 28 (the host keeps pairs < 2^28 on this path)

#define RT_T1 1e-6f          // the reference's EPSILON (Light.py:221)
#define RT_RHO 1.9073486e-06f   // 2^-19: bound on the ABSOLUTE error of the float32 unit direction w (measured need ~5e-7)

enum { SH_MISS = 0, SH_ATP = 1, SH_OCCLUDE = 2, SH_UNSURE = 4 };   // verdict bits: a ray ORs them over its candidates

// Everything below is written without branches: the lanes of a wave look at different candidates of different
// pairs, so every path is paid for anyway, and selects cost less than the exec-mask bookkeeping of nested ifs.

// verdict for one primitive from conservative bounds [lo, hi] of tau_out.  nvlim = nv - (4 rho nv + 1e-5), nv = |sample - P|:
// an entry point below nvlim certainly lies between P and the sample.
__device__ __forceinline__ int sh_classify(float lo, float hi, float nvlim) {
    const bool ok = lo <= hi;                                       // false: NaN somewhere
    const bool miss = hi < -1.1f * RT_T1;                           // entirely behind P
    const bool occ = lo > 1.1f * RT_T1 && hi < nvlim;
    const bool atp = lo > -0.9f * RT_T1 && hi < 0.9f * RT_T1;
    return !ok ? SH_UNSURE : miss ? SH_MISS : occ ? SH_OCCLUDE : atp ? SH_ATP : SH_UNSURE;
}

// sphere {a = centre - P, c = |centre - P|^2 - r^2}.  With b = w.a and D = b^2 - c the light ray enters at
// tau_out = b + sqrt(D); for b < 0 the stable form is  -c / (|b| + sqrt(D)).
// The verdict to reproduce is the REFERENCE's, and the reference evaluates its discriminant b'^2 - 4 a' c' from the
// sample, R = |sample - centre| <= nv + |a| away (Sphere.py:63-107): float64 rounding of <= 84 * 2^-53 R^2 there, which
// is nz = 2.4e-15 R^2 on D (a squared half chord) and moves its entry point by nz / (2 sqrt(D)).  Near a silhouette
// that is no longer small against the 1e-6 of the lit test (R = 700: a line within a half chord of 7e-3 of grazing):
// "hits" needs a half chord of gz = 1.5e-8 R^2 (entry point then within 0.8e-7 of the exact one, inside the margins of
// sh_classify), "misses" needs D < -nz; everything between goes to the float64 queue, which repeats the reference's
// arithmetic.
__device__ __forceinline__ int sh_sphere(float a0, float a1, float a2, float c, float w0, float w1, float w2, float nv, float nvlim) {
    const float b = fmaf(w0, a0, fmaf(w1, a1, w2 * a2));
    const float an1 = fabsf(a0) + fabsf(a1) + fabsf(a2);
    const float Rb = nv + an1, R2 = Rb * Rb;
    const float nz = 2.6e-15f * R2, gz = 1.5e-8f * R2;
    // clear miss, the common verdict: ed below is <= 1.1e-5 |a|^2 when D < 0 (then 0 <= c <= |a|^2)
    if (fmaf(b, b, -c) < -1.5e-5f * fmaf(a0, a0, fmaf(a1, a1, a2 * a2)) - nz) return SH_MISS;
    const float eb = RT_2M20 * (fabsf(w0 * a0) + fabsf(w1 * a1) + fabsf(w2 * a2)) + RT_RHO * an1;
    const float disc = fmaf(b, b, -c);
    const float ed = fmaf(2.0f * fabsf(b), eb, eb * eb) + 4.8e-7f * (b * b + fabsf(c));
    const bool miss = disc < -(ed + nz);                            // the line misses the sphere, for the reference too
    const float sl = __builtin_amdgcn_sqrtf(fmaxf(disc - ed, 0.0f)) * (1.0f - 6e-7f);
    const float sh = __builtin_amdgcn_sqrtf(disc + ed) * (1.0f + 6e-7f);
    const bool clear = disc > ed && sl >= gz;                       // else grazing
    const float ab = fabsf(b);
    const float qa = (ab - eb) + sl, qb = (ab + eb) + sh;           // bounds of |b| + sqrt(D)
    const float cw = fabsf(c) * 4.8e-7f;
    const float nlo = -c - cw, nhi = -c + cw;                       // bounds of -c
    const float rqa = __builtin_amdgcn_rcpf(qa), rqb = __builtin_amdgcn_rcpf(qb);   // 1 ulp; covered by the 7e-7 below
    float lo = nlo * (nlo >= 0.0f ? rqb : rqa), hi = nhi * (nhi >= 0.0f ? rqa : rqb);
    lo -= fabsf(lo) * 7e-7f;
    hi += fabsf(hi) * 7e-7f;
    const bool front = b > 0.0f;                                    // centre on the light side of P
    lo = front ? qa : lo;
    hi = front ? qb : hi;
    const bool sure = clear && ab >= eb && (front || qa > 0.0f);
    const int v = sh_classify(lo, hi, nvlim);
    return miss ? SH_MISS : sure ? v : SH_UNSURE;
}

struct RayInv {              // per ray, shared by all its box candidates
    float i0, i1, i2;        // 1 / w_k (1 ulp)
    float r0, r1, r2;        // relative error of a slab distance along axis k: the direction error is ABSOLUTE
                             // (<= RT_RHO per component), so it is RT_RHO / |w_k| (+ arithmetic)
    bool ok;                 // false: nearly axis parallel
};
__device__ __forceinline__ RayInv sh_ray_inv(float w0, float w1, float w2) {
    RayInv v;
    v.ok = !(fminf(fminf(fabsf(w0), fabsf(w1)), fabsf(w2)) < 4e-6f);
    v.i0 = __builtin_amdgcn_rcpf(w0); v.i1 = __builtin_amdgcn_rcpf(w1); v.i2 = __builtin_amdgcn_rcpf(w2);
    v.r0 = fmaf(RT_RHO, fabsf(v.i0), RT_2M20); v.r1 = fmaf(RT_RHO, fabsf(v.i1), RT_2M20); v.r2 = fmaf(RT_RHO, fabsf(v.i2), RT_2M20);
    return v;
}
// box {lo - P rounded down, hi - P rounded up}
__device__ __forceinline__ int sh_box(float lo0, float lo1, float lo2, float hi0, float hi1, float hi2, const RayInv &q, float nvlim) {
    const float a0 = lo0 * q.i0, b0 = hi0 * q.i0, a1 = lo1 * q.i1, b1 = hi1 * q.i1, a2 = lo2 * q.i2, b2 = hi2 * q.i2;
    const float n0 = fminf(a0, b0), x0 = fmaxf(a0, b0), n1 = fminf(a1, b1), x1 = fmaxf(a1, b1), n2 = fminf(a2, b2), x2 = fmaxf(a2, b2);
    const float dn0 = fabsf(n0) * q.r0, dn1 = fabsf(n1) * q.r1, dn2 = fabsf(n2) * q.r2;
    const float dx0 = fabsf(x0) * q.r0, dx1 = fabsf(x1) * q.r1, dx2 = fabsf(x2) * q.r2;
    const float en_lo = fmaxf(fmaxf(n0 - dn0, n1 - dn1), n2 - dn2), en_hi = fmaxf(fmaxf(n0 + dn0, n1 + dn1), n2 + dn2);
    const float ex_lo = fminf(fminf(x0 - dx0, x1 - dx1), x2 - dx2), ex_hi = fminf(fminf(x0 + dx0, x1 + dx1), x2 + dx2);
    const bool miss = en_lo > ex_hi;                                // the line misses the box
    const bool clean = en_hi < ex_lo;                               // else touching an edge / corner
    const int v = sh_classify(ex_lo, ex_hi, nvlim);                 // the light ray enters at tau = exit
    return !q.ok ? SH_UNSURE : miss ? SH_MISS : clean ? v : SH_UNSURE;
}

// plane: pd = n.P + d (float64, then rounded); tau = -pd / (w.n).  kw >= |coordinates of the sample| + |d|: the reference
// divides o.n + d (float64 rounding <= 4 * 2^-53 kw) and nv-long ray parameters (relative 3 * 2^-53 / (w.n)) by w.n
// (InfinitePlane.py:60-86); a verdict needs w.n large enough that both stay below 1e-7.
__device__ __forceinline__ int sh_plane(float n0, float n1, float n2, float pd, float w0, float w1, float w2, float nv, float nvlim, float kw) {
    const float vn = fmaf(w0, n0, fmaf(w1, n1, w2 * n2));
    const float ev = fmaxf(2.5f * RT_RHO, 5e-9f * (kw + nv));       // |n|_1 <= 1.74, plus the rounding of n and the dot product
    const bool back = vn < -2.5f * RT_RHO;                          // the light ray sees the back: one-sided (InfinitePlane.py:83)
    const bool facing = vn > ev;
    const float rp = __builtin_amdgcn_rcpf(vn + 2.5f * RT_RHO), rm = __builtin_amdgcn_rcpf(vn - 2.5f * RT_RHO);   // 1 ulp; covered by the 7e-7 below
    const float num = -pd, nu = num * (1.0f + 4.8e-7f), nd = num * (1.0f - 4.8e-7f);
    const bool pos = num >= 0.0f;
    float lo = pos ? nd * rp : nu * rm, hi = pos ? nu * rm : nd * rp;
    lo -= fabsf(lo) * 7e-7f;
    hi += fabsf(hi) * 7e-7f;
    const bool beyond = lo > nv + (nv - nvlim);                     // plane beyond the sample (t < 0 from the light)
    const int v = sh_classify(lo, hi, nvlim);
    return back ? SH_MISS : !facing ? SH_UNSURE : beyond ? SH_MISS : v;
}

// Whole-pair verdict for a plane, from the box [cB - h, cB + h] of all sample origins (relative to P):
// 1 = every ray of the pair misses it, 2 = it is P's own surface for every ray, 0 = decide per ray.
#define RT_PLANE_EVAL 0u
#define RT_PLANE_MISS 1u
#define RT_PLANE_ATP 2u
__device__ __forceinline__ uint32_t sh_plane_pair(float n0, float n1, float n2, float pd, float cB0, float cB1, float cB2,
                                                  float h0, float h1, float h2) {
    const float smax = (fabsf(cB0) + fabsf(cB1) + fabsf(cB2)) + (h0 + h1 + h2);      // >= |s| for every sample s
    const float sc = fmaf(cB0, n0, fmaf(cB1, n1, cB2 * n2));
    const float sh = fabsf(n0) * h0 + fabsf(n1) * h1 + fabsf(n2) * h2;
    const float thr = 1e-5f * smax;                                  // float32 rounding of s.n is ~1e-6 smax
    const float a = sc - sh, b = sc + sh;                            // s.n in [a, b];  w.n = s.n / |s|
    if (!(smax < 1e30f)) return RT_PLANE_EVAL;
    if (b < -thr) return RT_PLANE_MISS;                              // every light ray sees the back
    if (a > thr && pd > 2e-6f) return RT_PLANE_MISS;                 // tau = -pd / (w.n) < -pd <= -2e-6: behind P
    if (a > 1e-3f * smax && fabsf(pd) < 1e-10f) return RT_PLANE_ATP; // |tau| < 1e-10 / 1e-3
    return RT_PLANE_EVAL;
}

// ---------------------------------------------------------------------------------------------
// beam traversal: all points  lambda * s,  s in [cB - h, cB + h], lambda in [0, 1]  (P at the origin)
// ---------------------------------------------------------------------------------------------
// keep the part of [l0, l1] where  lambda * coef <= rhs.  icoef = 1 / coef to 1 ulp (anything when coef == 0);
// the new bound is rounded outward, the interval only ever comes out too large
__device__ __forceinline__ void clip_mul(float &l0, float &l1, float coef, float icoef, float rhs) {
    const float t = rhs * icoef, w = fabsf(t) * RT_2M20;
    l1 = coef > 0.0f ? fminf(l1, t + w) : l1;        // fminf / fmaxf drop a NaN t (0 * inf)
    l0 = coef < 0.0f ? fmaxf(l0, t - w) : l0;
    l1 = (coef == 0.0f && rhs < 0.0f) ? -1.0f : l1;  // 0 <= rhs fails: empty
}

// A primitive's box [c - rr, d + rr] (sphere: c = d = centre, rr = radius; box: c = lo, d = hi, rr = 0) lies apart from the box
// [blo, bhi] on some axis.  One maximum over the six signed gaps instead of six compares ANDed together (x > y <=> x - y > 0;
// a NaN gap is dropped by the maximum as it was by its compare: never a reason to cull).
__device__ __forceinline__ bool boxes_apart(float c0, float c1, float c2, float d0, float d1, float d2, float rr, float blo0, float blo1, float blo2,
                                            float bhi0, float bhi1, float bhi2) {
    const float g0 = (c0 - rr) - bhi0, g1 = (c1 - rr) - bhi1, g2 = (c2 - rr) - bhi2;
    const float h0 = blo0 - (d0 + rr), h1 = blo1 - (d1 + rr), h2 = blo2 - (d2 + rr);
    return fmaxf(fmaxf(fmaxf(g0, g1), g2), fmaxf(fmaxf(h0, h1), h2)) > 0.0f;
}

// Stack entry of the beam traversal: (node, lambda interval as two float16, rounded outward by up to 1.5 ulp).
// 8 bytes per level and lane, like the stacks of rt_fast.hpp; lambda is in [0, 1], 2^-11 relative is ~0.01
// length units at the far end of a beam.
__device__ __forceinline__ uint32_t pack_lambda(float l0, float l1) {
    uint32_t a = __half_as_ushort(__float2half_rn(l0)), b = __half_as_ushort(__float2half_rn(l1));
    a = a ? a - 1u : 0u;     // l0 >= 0: one float16 ulp down
    b = b + 1u;              // l1 <= 1: one ulp up, cannot overflow
    return a | (b << 16);
}
__device__ __forceinline__ void unpack_lambda(uint32_t p, float &l0, float &l1) {
    l0 = __half2float(__ushort_as_half((unsigned short)(p & 0xffffu)));
    l1 = __half2float(__ushort_as_half((unsigned short)(p >> 16)));
}

__device__ __forceinline__ float sel3(uint32_t ax, float v0, float v1, float v2) { return ax == 0 ? v0 : (ax == 1 ? v1 : v2); }

// The sample origins of a pair in the frame (up, right, axis = cB / |cB|; up, right are orthogonal to the axis):
// |s.up|, |s.right| <= aq and |s.axis - |cB|| <= ja for every sample s, deviations of Light.py:277-279 included.
struct LightQuad {
    float u0, u1, u2, r0, r1, r2;
    float aq, ja;
};

// Result of the beam traversal of one (hit, light) pair.
struct BeamOut {
    uint32_t nu, boxmask;    // candidate units the pair's record needs, which of them start a box
    uint32_t keep;           // bit k: touched primitive tl[k] is a candidate (beam_write_units writes their units, in this order)
    bool overflow;           // the units do not suffice: every ray of the pair goes to the float64 queue
    bool occ_all;            // one sphere occludes EVERY ray of the pair (traversal stopped there)
    bool atp_all;            // P lies on a sphere that every ray of the pair leaves through P (not stored in out[])
    uint32_t trips;          // diagnostics: node steps + leaf entries looked at
    uint32_t why;            // diagnostics: 1 touched list full, 2 candidate units full, 3 dedup list full
#ifdef RT_DIAG
    unsigned long long clk_rows, clk_walk;     // clock at the end of the tile rows / of the tree walk (make stats)
#endif
};

// c* / h*: centre and half size of the box of all sample origins, relative to P.  The candidates' units are NOT written here:
// a pair's record gets its place in memory only once the pair turns out to need one (13 % do), then beam_write_units
// re-expresses the kept primitives (BeamOut::keep).  Everything here is a conservative float32 filter; the absolute slack E covers the
// rounding of P, of the cut planes / leaf records and of the differences (<= 6 Kp 2^-24 in all); the candidates
// it lets through are re-expressed from float64.
//
// Whole-pair verdicts for spheres.  All sample origins lie in the cone around axis cB with tan(half angle) = kt;
// with C = centre - P, ta = C.axis, dp = distance of C from the axis, c = |C|^2 - r^2 (float64, rounded):
//   P outside (c > 1e-4 r^2): the sphere is seen from P under the half angle beta, tan(beta) = r / sqrt(c); every
//     ray hits it iff  angle(C, axis) + alpha < beta  <=>  dp + kt ta < tan(beta) (ta - kt dp);  the light ray
//     enters it at tau_out >= sqrt(c) from P and before |C| + r  ->  OCCLUDE for every ray when |C| + r < min |sample|;
//   P on the sphere (|c| tiny): b = w.C is <= (ta + kt dp)/sqrt(1+kt^2) and >= (ta - kt dp)/sqrt(1+kt^2) over the
//     cone;  b <= -bm for all rays: tau_out = -c / (|b| + sqrt(b^2 - c)), |tau_out| <= |c| / bm < 0.4e-6 -> AT_P;
//     b >= bm for all rays: tau_out >= b >= bm > 1.1e-6 (the ray crosses P's own sphere) -> OCCLUDE.
// The margins (1e-3 relative on the angles, bm = 1e-3 r) dwarf float32 and float64 rounding.
// EARLY (scenes with long beams): near side first, umbra test inside the traversal (see the loop).
// BOXES = false (scenes without boxes): every leaf entry is a sphere, the box arithmetic folds away.
// TILES (k_beam_setup<.., true>, level 0): called by all 64 lanes of a wave together (on = false: the lane has no pair, or a
// delicate one, and only helps staging); the lane's (block, light) row - trec / tcnt / row, row == 0xffffffff: none - lists the
// primitives that the union of the block's beams can touch, and the lane keeps those its own beam can touch instead of walking
// the tree.  rowbuf / rowbuf_hi: the wave's LDS slice for one row (2 * RT_TILE_NC float4, in two halves: the wave's columns of the
// first two rows of the traversal stack, which is not in use while rows are filtered); pre_row / pre / pre_cnt: a row already in registers.
struct TileRow {
    const float4 *__restrict__ trec;
    const uint32_t *__restrict__ tcnt;
    uint32_t row;
    float4 *rowbuf, *rowbuf_hi;       // units 0..RT_TILE_HALF-1 / RT_TILE_HALF..2*RT_TILE_HALF-1 of the staged half (rowbuf_hi is biased: index it with the unit number)
    int lane;
    uint32_t pre_row, pre_cnt;
    float4 pre;
};
template <int STRIDE, bool EARLY, bool BOXES, bool TILES = false>
__device__ __forceinline__ BeamOut beam_collect(const DScene &S, const FastScene &F, uint2 *stk, uint32_t *tl, d3 P, float cB0, float cB1,
                                                float cB2, float h0, float h1, float h2, const LightQuad &Q, int dbg,
                                                bool on = true, const TileRow *T = nullptr) {
    BeamOut R;
    R.nu = 0; R.boxmask = 0; R.keep = 0; R.overflow = false; R.occ_all = false; R.atp_all = false; R.trips = 0; R.why = 0;
    RT_DIAG_ONLY(R.clk_rows = R.clk_walk = clock64();)
    const float Pf0 = (float)P.x, Pf1 = (float)P.y, Pf2 = (float)P.z;
    const float Kp = fmaxf(fmaxf(fabsf(Pf0), fabsf(Pf1)), fmaxf(fabsf(Pf2), F.extent));
    const float E = Kp * RT_2M19 + 2e-6f;            // absolute slack on every plane position (and the 1e-6 offsets)
    const float L0 = cB0 - h0, L1 = cB1 - h1, L2 = cB2 - h2, U0 = cB0 + h0, U1 = cB1 + h1, U2 = cB2 + h2;
    // (tile beams: the reciprocals and the root-box clip belong to the tree walk, which few pairs need - computed there)
    float iL0 = 0.0f, iL1 = 0.0f, iL2 = 0.0f, iU0 = 0.0f, iU1 = 0.0f, iU2 = 0.0f;
    if (!TILES) {
        iL0 = __builtin_amdgcn_rcpf(L0); iL1 = __builtin_amdgcn_rcpf(L1); iL2 = __builtin_amdgcn_rcpf(L2);
        iU0 = __builtin_amdgcn_rcpf(U0); iU1 = __builtin_amdgcn_rcpf(U1); iU2 = __builtin_amdgcn_rcpf(U2);
    }
    // the beam also lies inside the cone with apex P and axis cB: a sample s = cB + e, |e| <= |h|, makes an angle
    // with sin <= rho = |h| / |cB| with the axis, so tan <= rho / sqrt(1 - rho^2)
    const float cl2 = fmaf(cB0, cB0, fmaf(cB1, cB1, cB2 * cB2));
    const float hh2 = fmaf(h0, h0, fmaf(h1, h1, h2 * h2));
    const float icl = __builtin_amdgcn_rsqf(cl2);
    const float ax0 = cB0 * icl, ax1 = cB1 * icl, ax2 = cB2 * icl;                    // |ax| = 1 +- 3e-7
    const float ax_len = cl2 * icl;
    const float rho2 = hh2 * icl * icl;
    const bool cone_ok = rho2 < 0.25f;               // else: no cone culling, no whole-pair verdicts
    const float kt = cone_ok ? __builtin_amdgcn_sqrtf(rho2) * __builtin_amdgcn_rsqf(1.0f - rho2) * 1.0001f : 1e30f;
    const float smin = (ax_len - __builtin_amdgcn_sqrtf(hh2) * 1.000001f) * (1.0f - 1e-6f);   // <= |sample| for every sample
    const float smaxv = (ax_len + __builtin_amdgcn_sqrtf(hh2) * 1.000001f) * (1.0f + 1e-6f);  // >= |sample| for every sample
    float l0 = 0.0f, l1 = 1.0f;
    // root box relative to P
    auto clip_root = [&]() {
        clip_mul(l0, l1, -U0, -iU0, -((F.root_lo[0] - Pf0) - E)); clip_mul(l0, l1, L0, iL0, (F.root_hi[0] - Pf0) + E);
        clip_mul(l0, l1, -U1, -iU1, -((F.root_lo[1] - Pf1) - E)); clip_mul(l0, l1, L1, iL1, (F.root_hi[1] - Pf1) + E);
        clip_mul(l0, l1, -U2, -iU2, -((F.root_lo[2] - Pf2) - E)); clip_mul(l0, l1, L2, iL2, (F.root_hi[2] - Pf2) + E);
    };
    if (!TILES) clip_root();
    // The light's sample box is wider than it is far (a light of radius 10 inside a scene of extent 4: original.txt): the beam is
    // no beam, it meets everything near P - its rays are decided one by one, the walk below would only end in an overflow.
    const bool wide = rho2 > 1.0f;
    const bool live = on && l0 <= l1;                // else: no pair here, or its beam misses the root box (tile beams: seen by the walk, if any)
    if (!TILES) {
        if (!live) return R;
        if (wide) { R.overflow = true; R.why = 1; return R; }
    }
    const uint32_t NONE = 0xffffffffu;
    uint32_t nt = 0, ntu = 0;                        // touched entries, and the candidate units they would need
    bool use_walk = true;
    if (TILES) {
        // ---- phase 1, tile beams.  Lanes of a wave share one or two rows: a row is staged into the wave's LDS slice by all 64
        // lanes (one float4 each), distinct rows take turns.  The lane tests the row's primitives against its own beam with the
        // box and cone tests of the walk's leaf loop - no tree, no duplicates.  An overflowing row sends its lanes through the tree.
        use_walk = live && !wide && T->row == NONE;
        // the beam's box from P (lambda 0) to the samples (lambda 1), unclipped: what lies outside the root box holds no primitive
        const float blo0 = fminf(0.0f, L0) - E, bhi0 = fmaxf(0.0f, U0) + E;
        const float blo1 = fminf(0.0f, L1) - E, bhi1 = fmaxf(0.0f, U1) + E;
        const float blo2 = fminf(0.0f, L2) - E, bhi2 = fmaxf(0.0f, U2) + E;
        bool pending = live && !wide && T->row != NONE;
        for (;;) {
            const unsigned long long pm = __ballot(pending);
            if (!pm) break;
            const uint32_t cur = (uint32_t)__builtin_amdgcn_readlane((int)T->row, __ffsll((long long)pm) - 1);
            const bool mine = pending && T->row == cur;
            float4 part = T->pre;                        // the row requested ahead (k_beam_setup), or
            uint32_t cnt = T->pre_cnt;
            if (cur != T->pre_row) {
                part = T->trec[(uint64_t)cur * (2u * RT_TILE_NC) + (uint32_t)T->lane];      // (entries past the count: never read)
                cnt = T->tcnt[cur];
            }
            cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)cnt);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");       // earlier reads of the slice are done (WAR)
            __builtin_amdgcn_wave_barrier();
            (T->lane < (int)RT_TILE_HALF ? T->rowbuf : T->rowbuf_hi)[T->lane] = part;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if (cnt == RT_TILE_OVERFLOW) {
                if (mine) use_walk = true;
            } else for (uint32_t half = 0; half * RT_TILE_HALF < cnt; ++half) {
                if (half) {                              // (few rows are longer than one staged half: the second is fetched on demand)
                    part = T->trec[(uint64_t)cur * (2u * RT_TILE_NC) + 2u * RT_TILE_HALF * half + (uint32_t)T->lane];
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __builtin_amdgcn_wave_barrier();
                    (T->lane < (int)RT_TILE_HALF ? T->rowbuf : T->rowbuf_hi)[T->lane] = part;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                }
                const uint32_t nh = min(cnt - half * RT_TILE_HALF, RT_TILE_HALF);
                for (uint32_t j = 0; j < nh; ++j) {
                    const float4 *rw = 2u * j < RT_TILE_HALF ? T->rowbuf : T->rowbuf_hi;
                    const float4 ra = rw[2u * j], rb4 = rw[2u * j + 1u];
                    const uint32_t pk = __float_as_uint(rb4.w);
                    const bool sph = !BOXES || (pk & 3u) == 1u;     // sphere {centre, r, (low parts)}; box {lo, -, hi}
                    const float rr = sph ? ra.w : 0.0f;
                    const float c0 = ra.x - Pf0, c1 = ra.y - Pf1, c2 = ra.z - Pf2;
                    const float d0 = sph ? c0 : rb4.x - Pf0, d1 = sph ? c1 : rb4.y - Pf1, d2 = sph ? c2 : rb4.z - Pf2;
                    bool keep = mine && !boxes_apart(c0, c1, c2, d0, d1, d2, rr, blo0, blo1, blo2, bhi0, bhi1, bhi2);
                    if (cone_ok) {                           // as in the walk's leaf loop
                        const float m0_ = sph ? c0 : 0.5f * (c0 + d0), m1_ = sph ? c1 : 0.5f * (c1 + d1), m2_ = sph ? c2 : 0.5f * (c2 + d2);
                        const float e0_ = d0 - c0, e1_ = d1 - c1, e2_ = d2 - c2;
                        const float rb = sph ? ra.w : 0.5f * __builtin_amdgcn_sqrtf(fmaf(e0_, e0_, fmaf(e1_, e1_, e2_ * e2_))) * 1.000001f;
                        const float ta = fmaf(m0_, ax0, fmaf(m1_, ax1, m2_ * ax2));
                        const float k0 = fmaf(m1_, ax2, -(m2_ * ax1)), k1 = fmaf(m2_, ax0, -(m0_ * ax2)), k2 = fmaf(m0_, ax1, -(m1_ * ax0));
                        const float dp2 = fmaf(k0, k0, fmaf(k1, k1, k2 * k2));
                        const float cn = (fabsf(m0_) + fabsf(m1_)) + fabsf(m2_);      // >= |centre|
                        const float lim = fmaf(rb, 1.0f + kt, fmaxf(ta, 0.0f) * kt) + E + 2e-6f * cn;
                        keep = keep && !(dp2 > lim * lim * 1.0001f) && !(ta < -(rb + E)) && !(ta > ax_len * (1.0f + kt) + rb + E);
                    }
                    if (keep) {
                        if (nt < RT_TOUCH) tl[nt * STRIDE] = pk >> 2;
                        ++nt;
                        ntu += sph ? 1u : 2u;
                    }
                }
            }
            if (mine) pending = false;
        }
    }

    // ---- phase 1: traversal.  The loops carry only the traversal state; a leaf entry that passes the float32
    // tests is parked (its primitive number, once) in the lane's LDS list tl[] (RT_TOUCH entries) and looked at
    // after the traversal.
    RT_DIAG_ONLY(R.clk_rows = clock64();)
    if (TILES && use_walk) {
        iL0 = __builtin_amdgcn_rcpf(L0); iL1 = __builtin_amdgcn_rcpf(L1); iL2 = __builtin_amdgcn_rcpf(L2);
        iU0 = __builtin_amdgcn_rcpf(U0); iU1 = __builtin_amdgcn_rcpf(U1); iU2 = __builtin_amdgcn_rcpf(U2);
        clip_root();
        use_walk = l0 <= l1;                         // else: the beam misses the root box
    }
    if (use_walk) {
    uint32_t node = 0;
    int sp = 0;
    for (;;) {
        // descend to a leaf (left first; the right child waits on the stack when the beam reaches both)
        KNode nd = F.knodes[node];
        while ((nd.b & 3u) != 3u) {
            const uint32_t ax = nd.b & 3u;
            RT_DIAG_ONLY(++R.trips;)
            const float cut = __uint_as_float(nd.a) - sel3(ax, Pf0, Pf1, Pf2);
            const float La = sel3(ax, L0, L1, L2), Ua = sel3(ax, U0, U1, U2), iLa = sel3(ax, iL0, iL1, iL2), iUa = sel3(ax, iU0, iU1, iU2);
            float a0 = l0, a1 = l1, b0 = l0, b1 = l1;
            clip_mul(a0, a1, La, iLa, cut + E);          // left child: some beam point with x <= cut
            clip_mul(b0, b1, -Ua, -iUa, -(cut - E));     // right child: some beam point with x >= cut
            const bool goL = a0 <= a1, goR = b0 <= b1;
            const uint32_t left = node + 1u, right = nd.b >> 2;
            if (goL && goR) {
                // long beams: the side whose part of the beam starts nearer to P first - a sphere that hides the whole
                // light from P (below) is usually met within a few leaves, and the rest of the beam is never walked.
                // Short beams (the usual case) go left first and skip the test: there it costs more than it saves.
                const bool left_first = !EARLY || a0 <= b0;
                stk[sp * STRIDE] = left_first ? make_uint2(right, pack_lambda(b0, b1)) : make_uint2(left, pack_lambda(a0, a1));
                ++sp;
                node = left_first ? left : right;
                l0 = left_first ? a0 : b0;
                l1 = left_first ? a1 : b1;
            } else if (goL) { node = left; l0 = a0; l1 = a1; }
            else if (goR) { node = right; l0 = b0; l1 = b1; }
            else { node = NONE; break; }
            nd = F.knodes[node];
        }
        if (node != NONE) {
            // leaf: beam bounding box over this leaf's lambda range (slightly widened)
            const float m0 = fmaxf(l0 - 1e-6f, 0.0f), m1 = fminf(l1 + 1e-6f, 1.0f);
            const float blo0 = fminf(m0 * L0, m1 * L0) - E, bhi0 = fmaxf(m0 * U0, m1 * U0) + E;
            const float blo1 = fminf(m0 * L1, m1 * L1) - E, bhi1 = fmaxf(m0 * U1, m1 * U1) + E;
            const float blo2 = fminf(m0 * L2, m1 * L2) - E, bhi2 = fmaxf(m0 * U2, m1 * U2) + E;
            // one leaf entry against the beam's part in this leaf; true: a sphere hides the whole light (EARLY), R is final
            auto entry = [&](const LeafF &lf) -> bool {
                RT_DIAG_ONLY(++R.trips;)
                const bool sph = !BOXES || (lf.pk & 3u) == 1u;     // sphere {centre, r, (low parts)}; box {lo, -, hi}
                const float rr = sph ? lf.r : 0.0f;
                const float c0 = lf.x - Pf0, c1 = lf.y - Pf1, c2 = lf.z - Pf2;
                const float d0 = sph ? c0 : lf.lx - Pf0, d1 = sph ? c1 : lf.ly - Pf1, d2 = sph ? c2 : lf.lz - Pf2;
                const bool touch = !boxes_apart(c0, c1, c2, d0, d1, d2, rr, blo0, blo1, blo2, bhi0, bhi1, bhi2);
                if (touch) {
                    bool in_cone = true;
                    if (cone_ok) {
                        // sphere (a box: its bounding sphere) against the cone: distance to the axis <= r (1 + tan) +
                        // axial position * tan (conservative).  The distance comes from the cross product: its error
                        // is ~1e-7 |c|, inside the 2e-6 |c| below.
                        const float m0_ = sph ? c0 : 0.5f * (c0 + d0), m1_ = sph ? c1 : 0.5f * (c1 + d1), m2_ = sph ? c2 : 0.5f * (c2 + d2);
                        const float e0_ = d0 - c0, e1_ = d1 - c1, e2_ = d2 - c2;
                        const float rb = sph ? lf.r : 0.5f * __builtin_amdgcn_sqrtf(fmaf(e0_, e0_, fmaf(e1_, e1_, e2_ * e2_))) * 1.000001f;
                        const float ta = fmaf(m0_, ax0, fmaf(m1_, ax1, m2_ * ax2));
                        const float k0 = fmaf(m1_, ax2, -(m2_ * ax1)), k1 = fmaf(m2_, ax0, -(m0_ * ax2)), k2 = fmaf(m0_, ax1, -(m1_ * ax0));
                        const float dp2 = fmaf(k0, k0, fmaf(k1, k1, k2 * k2));
                        const float cn = (fabsf(m0_) + fabsf(m1_)) + fabsf(m2_);      // >= |centre|
                        const float lim = fmaf(rb, 1.0f + kt, fmaxf(ta, 0.0f) * kt) + E + 2e-6f * cn;
                        in_cone = !(dp2 > lim * lim * 1.0001f) && !(ta < -(rb + E)) && !(ta > ax_len * (1.0f + kt) + rb + E);
                        // A sphere well away from P that covers the whole cone (the umbra verdict of phase 2, with
                        // |C|^2 - r^2 taken in float32: fine while it is >= 1 % of |C|^2): every ray is occluded, stop.
                        if (EARLY && in_cone && sph) {
                            const float cc = fmaf(c0, c0, fmaf(c1, c1, c2 * c2)), cf = cc - lf.r * lf.r;
                            if (cf > 0.01f * cc && ta > 0.0f) {
                                const float dp = __builtin_amdgcn_sqrtf(dp2), Cn = __builtin_amdgcn_sqrtf(cc);
                                const float tb = lf.r * __builtin_amdgcn_rsqf(cf) * (1.0f - 1e-3f);
                                const float num = fmaf(kt, ta, dp), den = fmaf(-kt, dp, ta);
                                if (den > 0.0f && fmaf(num, 1.001f, 1e-6f * cn) < tb * den && (Cn + lf.r) * 1.0001f + 1e-4f < smin) {
                                    R.occ_all = true;
                                    return true;
                                }
                            }
                        }
                    }
                    if (in_cone) {
                        const uint32_t prim = lf.pk >> 2;
                        bool seen = false;               // a primitive is listed in every leaf it reaches into
#pragma unroll
                        for (uint32_t j = 0; j < RT_TOUCH; ++j) seen = seen || (j < nt && tl[j * STRIDE] == prim);
                        if (!seen) {
                            if (nt < RT_TOUCH) tl[nt * STRIDE] = prim;
                            ++nt;
                            ntu += sph ? 1u : 2u;
                        }
                    }
                }
                return false;
            };
            const uint32_t e_end = nd.a + (nd.b >> 2);
            if (!BOXES && !EARLY) {
                // four entries are requested together: one memory round trip instead of four, four independent tests in flight
                // (the templates with boxes or the umbra test would spill registers over it)
                for (uint32_t e0 = nd.a; e0 < e_end && nt <= RT_TOUCH && ntu <= RT_CUNITS + 4u; e0 += 4u) {
                    LeafF lf4[4];
#pragma unroll
                    for (int k4 = 0; k4 < 4; ++k4) lf4[k4] = F.leaves[min(e0 + (uint32_t)k4, e_end - 1u)];
#pragma unroll
                    for (int k4 = 0; k4 < 4; ++k4)
                        if (e0 + (uint32_t)k4 < e_end) (void)entry(lf4[k4]);
                }
            } else {
                LeafF nxt = F.leaves[nd.a];                  // one entry ahead: the loop is bound by memory latency
                for (uint32_t e = nd.a; e < e_end && nt <= RT_TOUCH && ntu <= RT_CUNITS + 4u; ++e) {   // (leaves at the depth limit hold hundreds of entries)
                    const LeafF lf = nxt;
                    nxt = F.leaves[e + 1u < e_end ? e + 1u : e];
                    if (entry(lf)) return R;
                }
            }
        }
        if (sp == 0 || nt > RT_TOUCH || ntu > RT_CUNITS + 4u) break;   // the pair overflows whatever else the beam meets (boxes crowding around P)
        --sp;
        const uint2 t = stk[sp * STRIDE];
        node = t.x;
        unpack_lambda(t.y, l0, l1);
    }
    }
    RT_DIAG_ONLY(R.clk_walk = clock64();)
    if (TILES) {
        if (!live) return R;
        if (wide) { R.overflow = true; R.why = 1; return R; }
    }
    if (nt > RT_TOUCH || ntu > RT_CUNITS + 4u) { R.overflow = true; R.why = 1; return R; }

    // ---- phase 2: the touched primitives.  Whole-pair verdicts, float64 re-expression.
    uint32_t nu = 0, boxmask = 0;                    // units used, which units start a box
    for (uint32_t k = 0; k < nt; ++k) {
        // everything comes from the primitive's float64 record: ONE memory round trip per touched primitive (its
        // float32 leaf record would cost two more, prim_entry -> leaves, in a loop that is bound by them)
        const DPrim *pr = S.prims + tl[k * STRIDE];
        const uint32_t kind = BOXES ? (uint32_t)pr->kind : 1u;
        const bool sph = !BOXES || kind == 1u;
        const double pa0 = pr->a[0], pa1 = pr->a[1], pa2 = pr->a[2], pb0 = pr->b[0], pb1 = pr->b[1];
        double dx = 0, dy = 0, dz = 0, cd = 0;
        float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f, lfr = 0.0f;
        if (sph) {
            dx = pa0 - P.x; dy = pa1 - P.y; dz = pa2 - P.z;
            c0 = (float)dx; c1 = (float)dy; c2 = (float)dz;
            lfr = __double2float_ru(fabs(pb1));           // radius, rounded up like the leaf record's
        }
        float ta = 0.0f, dp2 = 0.0f, cn = 0.0f;
        if (sph && cone_ok) {                        // (passed the cone test in phase 1)
            ta = fmaf(c0, ax0, fmaf(c1, ax1, c2 * ax2));
            const float k0 = fmaf(c1, ax2, -(c2 * ax1)), k1 = fmaf(c2, ax0, -(c0 * ax2)), k2 = fmaf(c0, ax1, -(c1 * ax0));
            dp2 = fmaf(k0, k0, fmaf(k1, k1, k2 * k2));
            cn = (fabsf(c0) + fabsf(c1)) + fabsf(c2);                             // >= |c|
        }
        bool store = true;
        if (sph) {
            cd = ((dx * dx + dy * dy) + dz * dz) - pb0;
            if (cone_ok) {                       // whole-pair verdict (see above)
                // Every ray of the pair misses the sphere?  At the centre's axial depth ta the rays cross the
                // rectangle |x|, |y| <= aq t, t <= ta / (|cB| - ja); a line at angle phi to the axis stays
                // >= d cos(phi) away from a point at lateral distance d, and 1 / cos(phi) <= 1 + kt^2 / 2.
                if (ta > 0.0f) {
                    const float tm = (ta + E) * __builtin_amdgcn_rcpf(ax_len - Q.ja) * 1.00001f * Q.aq + E;
                    const float cu = fabsf(fmaf(c0, Q.u0, fmaf(c1, Q.u1, c2 * Q.u2))), cr = fabsf(fmaf(c0, Q.r0, fmaf(c1, Q.r1, c2 * Q.r2)));
                    const float ex = fmaxf(cu - tm - 2e-6f * cn, 0.0f), ey = fmaxf(cr - tm - 2e-6f * cn, 0.0f);
                    const float rl = fmaf(lfr, fmaf(0.5f * kt, kt, 1.001f), E);
                    if (fmaf(ex, ex, ey * ey) > rl * rl) continue;
                }
                const float cf = (float)cd, rr = lfr, dp = __builtin_amdgcn_sqrtf(dp2);
                const float Cn = __builtin_amdgcn_sqrtf(fmaf(c0, c0, fmaf(c1, c1, c2 * c2)));
                // (the reference's own rounding at a silhouette, see sh_sphere: a whole-pair verdict on P's own sphere needs
                // every ray's half chord |b| above 1.5e-8 R^2, R <= max |sample| + r)
                const float Rw = smaxv + rr;
                const float bm = fmaxf(1e-3f * rr, 1.5e-8f * Rw * Rw);
                const bool ends_before = (Cn + rr) * 1.0001f + 1e-4f < smin;
                const float num = fmaf(kt, ta, dp), den = fmaf(-kt, dp, ta);       // dp + kt ta,  ta - kt dp
                if (cf > 1e-4f * rr * rr) {
                    const float tb = rr * __builtin_amdgcn_rsqf(cf) * (1.0f - 1e-3f);
                    if (ta > 0.0f && den > 0.0f && fmaf(num, 1.001f, 1e-6f * cn) < tb * den && ends_before) { R.occ_all = true; return R; }
                } else if (fabsf(cf) < 1e-3f * bm * bm && fabsf(cf) < 0.4e-6f * bm) {
                    const float guard = bm * (1.0f + kt) * 1.001f + 1e-6f * cn;
                    if (ta + kt * dp <= -guard) { R.atp_all = true; store = false; }
                    else if (den >= guard && bm > 1e-5f && ends_before) { R.occ_all = true; return R; }
                }
            }
        }
        float bl0 = 0, bl1 = 0, bl2 = 0, bh0 = 0, bh1 = 0, bh2 = 0;
        if (!sph) {
            // box relative to P, rounded outward (float64 difference first)
            bl0 = __double2float_rd(pr->a[0] - P.x); bl1 = __double2float_rd(pr->a[1] - P.y); bl2 = __double2float_rd(pr->a[2] - P.z);
            bh0 = __double2float_ru(pr->b[0] - P.x); bh1 = __double2float_ru(pr->b[1] - P.y); bh2 = __double2float_ru(pr->b[2] - P.z);
            if (cone_ok) {
                // whole-pair verdicts for a box.  (1) every ray misses its bounding sphere (same rectangle test as for
                // spheres).  (2) P lies in the interior of one of its faces (|face - P| < 1e-9 along axis k, > 1e-5 from
                // the other five planes): with every sample on the outer side of that face (|w_k| >= 1e-2) the light ray
                // arrives at P from outside, tau_out = (face - P)_k / w_k, |tau_out| < 1e-7 -> AT_P; with every sample on
                // the inner side it has to cross the box first, tau_out >= min(thickness, 1e-5) and <= its diagonal ->
                // OCCLUDE (when the box ends before the nearest sample).
                const float m0_ = 0.5f * (bl0 + bh0), m1_ = 0.5f * (bl1 + bh1), m2_ = 0.5f * (bl2 + bh2);
                const float e0_ = bh0 - bl0, e1_ = bh1 - bl1, e2_ = bh2 - bl2;
                const float diag = __builtin_amdgcn_sqrtf(fmaf(e0_, e0_, fmaf(e1_, e1_, e2_ * e2_))) * 1.000001f, rb = 0.5f * diag;
                const float tab = fmaf(m0_, ax0, fmaf(m1_, ax1, m2_ * ax2));
                const float cnb = (fabsf(m0_) + fabsf(m1_)) + fabsf(m2_);
                if (tab > 0.0f) {
                    const float tm = (tab + E) * __builtin_amdgcn_rcpf(ax_len - Q.ja) * 1.00001f * Q.aq + E;
                    const float cu = fabsf(fmaf(m0_, Q.u0, fmaf(m1_, Q.u1, m2_ * Q.u2))), cr = fabsf(fmaf(m0_, Q.r0, fmaf(m1_, Q.r1, m2_ * Q.r2)));
                    const float ex = fmaxf(cu - tm - 2e-6f * cnb, 0.0f), ey = fmaxf(cr - tm - 2e-6f * cnb, 0.0f);
                    const float rl = fmaf(rb, fmaf(0.5f * kt, kt, 1.001f), E);
                    if (fmaf(ex, ex, ey * ey) > rl * rl) continue;
                }
                const float smax = ax_len + __builtin_amdgcn_sqrtf(hh2) * 1.000001f;            // >= |sample|
                const float wm = 1e-2f * smax;
                const float tiny = 1e-9f, clear = 1e-5f;
                const bool in0 = bl0 < -clear && bh0 > clear, in1 = bl1 < -clear && bh1 > clear, in2 = bl2 < -clear && bh2 > clear;
                // which face, if any: 0 none, +-(k+1)
                int face = 0;
                if (in1 && in2) { if (fabsf(bh0) < tiny && bl0 < -clear) face = 1; else if (fabsf(bl0) < tiny && bh0 > clear) face = -1; }
                if (in0 && in2) { if (fabsf(bh1) < tiny && bl1 < -clear) face = 2; else if (fabsf(bl1) < tiny && bh1 > clear) face = -2; }
                if (in0 && in1) { if (fabsf(bh2) < tiny && bl2 < -clear) face = 3; else if (fabsf(bl2) < tiny && bh2 > clear) face = -3; }
                if (face != 0) {
                    const int k = face > 0 ? face - 1 : -face - 1;
                    const float Lk = sel3((uint32_t)k, L0, L1, L2), Uk = sel3((uint32_t)k, U0, U1, U2);
                    const bool outer = face > 0 ? Lk >= wm : Uk <= -wm;          // every sample beyond the face, outside the box
                    const bool inner = face > 0 ? Uk <= -wm : Lk >= wm;          // every sample on the box's side of the face
                    if (outer) { R.atp_all = true; store = false; }
                    else if (inner && diag * 1.0001f + 1e-4f < smin) { R.occ_all = true; return R; }
                }
            }
        }
        if (store) {
            if (nu + kind > RT_CUNITS) { R.overflow = true; R.why = 2; return R; }          // kind == units needed (sphere 1, box 2)
            R.keep |= 1u << k;
            if (!sph) boxmask |= 1u << nu;
            nu += kind;
        }
    }
    R.nu = nu; R.boxmask = boxmask;
    return R;
}

// The candidate units of a listed pair (k_beam_setup, once the pair has its record): the kept primitives of beam_collect,
// in its order, re-expressed relative to P from their float64 records - sphere {centre - P, |centre - P|^2 - r^2}, box two
// units {lo - P rounded down, hi - P rounded up}.
template <int STRIDE, bool BOXES>
__device__ __forceinline__ void beam_write_units(const DScene &S, const uint32_t *tl, uint32_t keep, d3 P, float4 *out) {
    uint32_t nu = 0;
    while (keep) {
        const uint32_t k = (uint32_t)__builtin_ctz(keep);
        keep &= keep - 1u;
        const DPrim *pr = S.prims + tl[k * STRIDE];
        const bool sph = !BOXES || pr->kind == 1;
        if (sph) {
            const double dx = pr->a[0] - P.x, dy = pr->a[1] - P.y, dz = pr->a[2] - P.z;
            const double cd = ((dx * dx + dy * dy) + dz * dz) - pr->b[0];
            out[nu] = make_float4((float)dx, (float)dy, (float)dz, (float)cd);
            nu += 1;
        } else {
            const float bl0 = __double2float_rd(pr->a[0] - P.x), bl1 = __double2float_rd(pr->a[1] - P.y), bl2 = __double2float_rd(pr->a[2] - P.z);
            const float bh0 = __double2float_ru(pr->b[0] - P.x), bh1 = __double2float_ru(pr->b[1] - P.y), bh2 = __double2float_ru(pr->b[2] - P.z);
            out[nu] = make_float4(bl0, bl1, bl2, bh0);
            out[nu + 1] = make_float4(bh1, bh2, 0.0f, 0.0f);
            nu += 2;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// One shadow ray of a pair whose beam met too many primitives (k_beam_setup's candidate overflow): the ray walks
// the tree itself, as a beam that is only as wide as the float32 uncertainty of its sample, and classifies every
// primitive it may touch on the spot with the same predicates and the same float64-derived operands as the listed
// pairs' rays.  Primitives the thin beam provably cannot touch are misses for the exact ray as well; one that
// appears in several leaves is classified several times, which changes nothing (verdict bits are ORed, only
// n_atp > 0 matters).  Near child first (smaller lambda): an occluder close to P ends the walk early.
//   sw  = sample - P (float32, error <= nv RT_RHO per component),  w = sw / nv,  qi = sh_ray_inv(w)
// Returns the ORed verdict; n_atp counts AT_P verdicts.
template <int STRIDE>
__device__ __forceinline__ int ray_walk(const DScene &S, const FastScene &F, uint2 *stk, d3 P, float sw0, float sw1, float sw2, float w0,
                                        float w1, float w2, float nv, float nvlim, int &n_atp) {
    int verdict = 0;
    const float Pf0 = (float)P.x, Pf1 = (float)P.y, Pf2 = (float)P.z;
    const float Kp = fmaxf(fmaxf(fabsf(Pf0), fabsf(Pf1)), fmaxf(fabsf(Pf2), F.extent));
    const float E = Kp * RT_2M19 + 2e-6f;
    const float hs = fmaf(nv, 2.0f * RT_RHO, 1e-6f);                 // the exact sample lies in sw +- hs
    const float L0 = sw0 - hs, L1 = sw1 - hs, L2 = sw2 - hs, U0 = sw0 + hs, U1 = sw1 + hs, U2 = sw2 + hs;
    const float iL0 = __builtin_amdgcn_rcpf(L0), iL1 = __builtin_amdgcn_rcpf(L1), iL2 = __builtin_amdgcn_rcpf(L2);
    const float iU0 = __builtin_amdgcn_rcpf(U0), iU1 = __builtin_amdgcn_rcpf(U1), iU2 = __builtin_amdgcn_rcpf(U2);
    // cone around w with tan(half angle) kt: sin <= sqrt(3) hs / nv
    const float kt = 1.7321f * hs * __builtin_amdgcn_rcpf(nv) * 1.01f;
    float l0 = 0.0f, l1 = 1.0f;
    clip_mul(l0, l1, -U0, -iU0, -((F.root_lo[0] - Pf0) - E)); clip_mul(l0, l1, L0, iL0, (F.root_hi[0] - Pf0) + E);
    clip_mul(l0, l1, -U1, -iU1, -((F.root_lo[1] - Pf1) - E)); clip_mul(l0, l1, L1, iL1, (F.root_hi[1] - Pf1) + E);
    clip_mul(l0, l1, -U2, -iU2, -((F.root_lo[2] - Pf2) - E)); clip_mul(l0, l1, L2, iL2, (F.root_hi[2] - Pf2) + E);
    if (!(l0 <= l1)) return verdict;
    const RayInv qi = sh_ray_inv(w0, w1, w2);
    const uint32_t NONE = 0xffffffffu;
    uint32_t node = 0;
    int sp = 0;
    for (;;) {
        KNode nd = F.knodes[node];
        while ((nd.b & 3u) != 3u) {
            const uint32_t ax = nd.b & 3u;
            const float cut = __uint_as_float(nd.a) - sel3(ax, Pf0, Pf1, Pf2);
            const float La = sel3(ax, L0, L1, L2), Ua = sel3(ax, U0, U1, U2), iLa = sel3(ax, iL0, iL1, iL2), iUa = sel3(ax, iU0, iU1, iU2);
            float a0 = l0, a1 = l1, b0 = l0, b1 = l1;
            clip_mul(a0, a1, La, iLa, cut + E);          // left child: some beam point with x <= cut
            clip_mul(b0, b1, -Ua, -iUa, -(cut - E));     // right child: some beam point with x >= cut
            const bool goL = a0 <= a1, goR = b0 <= b1;
            const uint32_t left = node + 1u, right = nd.b >> 2;
            if (goL && goR) {
                const bool left_first = a0 <= b0;        // the side that holds the part nearer to P
                stk[sp * STRIDE] = left_first ? make_uint2(right, pack_lambda(b0, b1)) : make_uint2(left, pack_lambda(a0, a1));
                ++sp;
                node = left_first ? left : right;
                l0 = left_first ? a0 : b0;
                l1 = left_first ? a1 : b1;
            } else if (goL) { node = left; l0 = a0; l1 = a1; }
            else if (goR) { node = right; l0 = b0; l1 = b1; }
            else { node = NONE; break; }
            nd = F.knodes[node];
        }
        if (node != NONE) {
            const float m0 = fmaxf(l0 - 1e-6f, 0.0f), m1 = fminf(l1 + 1e-6f, 1.0f);
            const float blo0 = fminf(m0 * L0, m1 * L0) - E, bhi0 = fmaxf(m0 * U0, m1 * U0) + E;
            const float blo1 = fminf(m0 * L1, m1 * L1) - E, bhi1 = fmaxf(m0 * U1, m1 * U1) + E;
            const float blo2 = fminf(m0 * L2, m1 * L2) - E, bhi2 = fmaxf(m0 * U2, m1 * U2) + E;
            // four entries are requested and tested against the thin beam together (one memory round trip instead of
            // four); the ones it may touch are then classified one after the other
            const uint32_t e_last = nd.a + (nd.b >> 2) - 1u;
            for (uint32_t e0 = nd.a; e0 <= e_last; e0 += 4u) {
                LeafF lf4[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) lf4[k] = F.leaves[min(e0 + (uint32_t)k, e_last)];
                uint32_t mask = 0u;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const LeafF &lf = lf4[k];
                    const bool sph = (lf.pk & 3u) == 1u;
                    const float rr = sph ? lf.r : 0.0f;
                    const float c0 = lf.x - Pf0, c1 = lf.y - Pf1, c2 = lf.z - Pf2;
                    const float d0 = sph ? c0 : lf.lx - Pf0, d1 = sph ? c1 : lf.ly - Pf1, d2 = sph ? c2 : lf.lz - Pf2;
                    bool touch = !boxes_apart(c0, c1, c2, d0, d1, d2, rr, blo0, blo1, blo2, bhi0, bhi1, bhi2);
                    // bounding sphere against the thin cone (same test as the beam's, beam_collect)
                    const float m0_ = sph ? c0 : 0.5f * (c0 + d0), m1_ = sph ? c1 : 0.5f * (c1 + d1), m2_ = sph ? c2 : 0.5f * (c2 + d2);
                    const float e0_ = d0 - c0, e1_ = d1 - c1, e2_ = d2 - c2;
                    const float rb = sph ? lf.r : 0.5f * __builtin_amdgcn_sqrtf(fmaf(e0_, e0_, fmaf(e1_, e1_, e2_ * e2_))) * 1.000001f;
                    const float ta = fmaf(m0_, w0, fmaf(m1_, w1, m2_ * w2));
                    const float k0 = fmaf(m1_, w2, -(m2_ * w1)), k1 = fmaf(m2_, w0, -(m0_ * w2)), k2 = fmaf(m0_, w1, -(m1_ * w0));
                    const float dp2 = fmaf(k0, k0, fmaf(k1, k1, k2 * k2));
                    const float cn = (fabsf(m0_) + fabsf(m1_)) + fabsf(m2_);
                    const float lim = fmaf(rb, 1.0f + kt, fmaxf(ta, 0.0f) * kt) + E + 4e-6f * cn;      // (w itself is off by RT_RHO)
                    touch = touch && !(dp2 > lim * lim * 1.0001f) && !(ta < -(rb + E + 4e-6f * cn)) && !(ta > nv * (1.0f + kt) + rb + E + 4e-6f * cn);
                    if (touch && e0 + (uint32_t)k <= e_last) mask |= 1u << k;
                }
                while (mask) {
                    const int k = __builtin_ctz(mask);
                    mask &= mask - 1u;
                    const uint32_t pk = k == 0 ? lf4[0].pk : (k == 1 ? lf4[1].pk : (k == 2 ? lf4[2].pk : lf4[3].pk));
                    const DPrim *pr = S.prims + (pk >> 2);
                    int v;
                    if ((pk & 3u) == 1u) {
                        const double dx = pr->a[0] - P.x, dy = pr->a[1] - P.y, dz = pr->a[2] - P.z;
                        const double cd = ((dx * dx + dy * dy) + dz * dz) - pr->b[0];
                        v = sh_sphere((float)dx, (float)dy, (float)dz, (float)cd, w0, w1, w2, nv, nvlim);
                    } else {
                        v = sh_box(__double2float_rd(pr->a[0] - P.x), __double2float_rd(pr->a[1] - P.y), __double2float_rd(pr->a[2] - P.z),
                                   __double2float_ru(pr->b[0] - P.x), __double2float_ru(pr->b[1] - P.y), __double2float_ru(pr->b[2] - P.z), qi, nvlim);
                    }
                    verdict |= v;
                    n_atp += v & SH_ATP;
                    if (verdict & SH_OCCLUDE) return verdict;        // one certain occluder settles the ray
                }
            }
        }
        if (sp == 0) break;
        --sp;
        const uint2 t = stk[sp * STRIDE];
        node = t.x;
        unpack_lambda(t.y, l0, l1);
    }
    return verdict;
}
