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
// light, so (1) ONE conservative beam traversal per pair (k_shadow_setup) collects the few
// primitives any of its rays can touch, with their coordinates taken RELATIVE TO P in float64 and
// only then rounded to float32 - the own surface sits at ~1e-16, not at an ulp of a world
// coordinate; (2) each ray classifies those candidates in float32 with explicit error margins
// (k_shadow_beam).  A verdict is only taken when every margin is cleared; anything else (grazing
// silhouettes, |w.n| ~ 0, sample inside a primitive, candidate-list overflow) is appended to a
// queue and decided by the float64 path of rt_fast.hpp (k_shadow_queue).  So the outcome is that
// of the float64 reference-order evaluation, at float32 VALU cost for ~99.9 % of the rays.
#pragma once
#include "rt_fast.hpp"

#define RT_CUNITS 8          // 16-byte candidate units kept per (hit, light) pair: a sphere takes 1, a box 2
#define RT_CMAX RT_CUNITS
#define RT_PMAX 3            // planes handled by the filtered path
#define RT_PAIR_UNITS (4 + RT_CUNITS)   // one pair record = 12 float4 = 192 B, contiguous

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
    float4 *rec;             // [pair][RT_PAIR_UNITS]
    unsigned long long *queue;   // shadow-ray numbers g the filter could not decide
    uint32_t *queue_n;       // [0] = entries appended, [1] = capacity
};
#define RT_META_OVERFLOW 0xFFu

#define RT_T1 1e-6f          // the reference's EPSILON (Light.py:221)
#define RT_RHO 1.9073486e-06f   // 2^-19: bound on the ABSOLUTE error of the float32 unit direction w (measured need ~5e-7)

enum { SH_MISS = 0, SH_ATP = 1, SH_OCCLUDE = 2, SH_UNSURE = 3 };

// verdict for one primitive from conservative bounds [lo, hi] of tau_out; nv = |sample - P|
__device__ __forceinline__ int sh_classify(float lo, float hi, float nv) {
    if (!(lo <= hi)) return SH_UNSURE;                             // NaN somewhere
    if (hi < -1.1f * RT_T1) return SH_MISS;                         // entirely behind P
    if (lo > 1.1f * RT_T1) return (hi < nv - fmaf(nv, 4.0f * RT_RHO, 1e-5f)) ? SH_OCCLUDE : SH_UNSURE;
    if (lo > -0.9f * RT_T1 && hi < 0.9f * RT_T1) return SH_ATP;
    return SH_UNSURE;
}

__device__ __forceinline__ int sh_sphere(const CandF &k, float w0, float w1, float w2, float nv) {
    const float b = fmaf(w0, k.a[0], fmaf(w1, k.a[1], w2 * k.a[2]));
    const float eb = RT_2M20 * (fabsf(w0 * k.a[0]) + fabsf(w1 * k.a[1]) + fabsf(w2 * k.a[2])) +
                     RT_RHO * (fabsf(k.a[0]) + fabsf(k.a[1]) + fabsf(k.a[2]));
    const float disc = fmaf(b, b, -k.c);
    const float ed = fmaf(2.0f * fabsf(b), eb, eb * eb) + 4.8e-7f * (b * b + fabsf(k.c));
    if (disc < -ed) return SH_MISS;                                 // the line misses the sphere
    if (!(disc > ed)) return SH_UNSURE;                             // grazing
    const float sl = __builtin_amdgcn_sqrtf(disc - ed) * (1.0f - 6e-7f), sh = __builtin_amdgcn_sqrtf(disc + ed) * (1.0f + 6e-7f);
    float lo, hi;
    if (b >= eb) {                    // centre on the light side: tau_out = b + sqrt(disc)
        lo = (b - eb) + sl;
        hi = (b + eb) + sh;
    } else if (b <= -eb) {            // centre away from the light: tau_out = c / (b - sqrt(disc)), stable form
        const float qlo = (b - eb) - sh, qhi = (b + eb) - sl;      // both negative
        const float clo = k.c - fabsf(k.c) * 4.8e-7f, chi = k.c + fabsf(k.c) * 4.8e-7f;
        if (!(qhi < 0.0f)) return SH_UNSURE;
        const float rqhi = __builtin_amdgcn_rcpf(qhi), rqlo = __builtin_amdgcn_rcpf(qlo);   // 1 ulp; covered below
        if (clo >= 0.0f) { lo = chi * rqhi; hi = clo * rqlo; }       // c >= 0: tau_out <= 0
        else if (chi <= 0.0f) { lo = chi * rqlo; hi = clo * rqhi; }  // c <= 0: tau_out >= 0
        else { lo = chi * rqhi; hi = clo * rqhi; }
        lo -= fabsf(lo) * 7e-7f;
        hi += fabsf(hi) * 7e-7f;
    } else return SH_UNSURE;
    return sh_classify(lo, hi, nv);
}

__device__ __forceinline__ int sh_box(const CandF &k, float w0, float w1, float w2, float nv) {
    if (fabsf(w0) < 4e-6f || fabsf(w1) < 4e-6f || fabsf(w2) < 4e-6f) return SH_UNSURE;   // nearly axis parallel
    const float i0 = __builtin_amdgcn_rcpf(w0), i1 = __builtin_amdgcn_rcpf(w1), i2 = __builtin_amdgcn_rcpf(w2);   // 1 ulp, inside r_k below
    float a0 = k.a[0] * i0, b0 = k.b[0] * i0, a1 = k.a[1] * i1, b1 = k.b[1] * i1, a2 = k.a[2] * i2, b2 = k.b[2] * i2;
    float n0 = fminf(a0, b0), x0 = fmaxf(a0, b0), n1 = fminf(a1, b1), x1 = fmaxf(a1, b1), n2 = fminf(a2, b2), x2 = fmaxf(a2, b2);
    // the direction error is ABSOLUTE (<= RT_RHO per component), so the relative error of a slab distance
    // along axis k is RT_RHO / |w_k| (+ arithmetic)
    const float r0 = fmaf(RT_RHO, fabsf(i0), RT_2M20), r1 = fmaf(RT_RHO, fabsf(i1), RT_2M20), r2 = fmaf(RT_RHO, fabsf(i2), RT_2M20);
    float en_lo = fmaxf(fmaxf(n0 - fabsf(n0) * r0, n1 - fabsf(n1) * r1), n2 - fabsf(n2) * r2);
    float en_hi = fmaxf(fmaxf(n0 + fabsf(n0) * r0, n1 + fabsf(n1) * r1), n2 + fabsf(n2) * r2);
    float ex_lo = fminf(fminf(x0 - fabsf(x0) * r0, x1 - fabsf(x1) * r1), x2 - fabsf(x2) * r2);
    float ex_hi = fminf(fminf(x0 + fabsf(x0) * r0, x1 + fabsf(x1) * r1), x2 + fabsf(x2) * r2);
    if (en_lo > ex_hi) return SH_MISS;                              // the line misses the box
    if (!(en_hi < ex_lo)) return SH_UNSURE;                         // touching an edge / corner
    return sh_classify(ex_lo, ex_hi, nv);                           // the light ray enters at tau = exit
}

__device__ __forceinline__ int sh_plane(float n0, float n1, float n2, float pd, float w0, float w1, float w2, float nv) {
    const float vn = fmaf(w0, n0, fmaf(w1, n1, w2 * n2));
    const float ev = 2.5f * RT_RHO;                                 // |n|_1 <= 1.74, plus the rounding of n and the dot product
    if (vn < -ev) return SH_MISS;                                   // the light ray sees the back: one-sided (InfinitePlane.py:83)
    if (!(vn > ev)) return SH_UNSURE;
    // tau = -pd / vn
    const float num = -pd;
    float lo, hi;
    const float rp = __builtin_amdgcn_rcpf(vn + ev), rm = __builtin_amdgcn_rcpf(vn - ev);   // 1 ulp; covered below
    if (num >= 0.0f) { lo = num * (1.0f - 4.8e-7f) * rp; hi = num * (1.0f + 4.8e-7f) * rm; }
    else { lo = num * (1.0f + 4.8e-7f) * rm; hi = num * (1.0f - 4.8e-7f) * rp; }
    lo -= fabsf(lo) * 7e-7f;
    hi += fabsf(hi) * 7e-7f;
    if (lo > nv + fmaf(nv, 4.0f * RT_RHO, 1e-5f)) return SH_MISS;   // plane beyond the sample (t < 0 from the light)
    return sh_classify(lo, hi, nv);
}

// ---------------------------------------------------------------------------------------------
// beam traversal: all points  lambda * s,  s in [cB - h, cB + h], lambda in [0, 1]  (P at the origin)
// ---------------------------------------------------------------------------------------------
// keep the part of [l0, l1] where  lambda * coef <= rhs
__device__ __forceinline__ void clip_le(float &l0, float &l1, float coef, float rhs) {
    if (coef > 0.0f) l1 = fminf(l1, rhs / coef);
    else if (coef < 0.0f) l0 = fmaxf(l0, rhs / coef);
    else if (rhs < 0.0f) l1 = -1.0f;                 // 0 <= rhs fails: empty
}
__device__ __forceinline__ void clip_ge(float &l0, float &l1, float coef, float rhs) { clip_le(l0, l1, -coef, -rhs); }

struct BeamStack {           // LDS columns, one entry per tree level
    uint2 *a;                // (node, lambda0 bits)
    float *b;                // lambda1
};

__device__ __forceinline__ float sel3(uint32_t ax, float v0, float v1, float v2) { return ax == 0 ? v0 : (ax == 1 ? v1 : v2); }

// returns units used | box mask << 8 (RT_META_OVERFLOW when the units do not suffice).  c* / h*: centre and half
// size of the box of all sample origins, relative to P.  out: the pair record's RT_CUNITS candidate units.
template <int STRIDE>
__device__ __forceinline__ uint32_t beam_collect(const DScene &S, const FastScene &F, BeamStack stk, d3 P, float cB0, float cB1,
                                                 float cB2, float h0, float h1, float h2, float4 *out) {
    const float Pf0 = (float)P.x, Pf1 = (float)P.y, Pf2 = (float)P.z;
    const float Pl0 = (float)(P.x - (double)Pf0), Pl1 = (float)(P.y - (double)Pf1), Pl2 = (float)(P.z - (double)Pf2);
    const float Kp = fmaxf(fmaxf(fabsf(Pf0), fabsf(Pf1)), fmaxf(fabsf(Pf2), F.extent));
    const float E = Kp * RT_2M19 + 2e-6f;            // absolute slack on every plane position (rounding of cut, P, and 1e-6 offsets)
    const float L0 = cB0 - h0, L1 = cB1 - h1, L2 = cB2 - h2, U0 = cB0 + h0, U1 = cB1 + h1, U2 = cB2 + h2;
    // the beam also lies inside the cone with apex P, axis cB and half-angle tan = |h| / |cB| (used for spheres)
    const float cl2 = fmaf(cB0, cB0, fmaf(cB1, cB1, cB2 * cB2));
    const float icl = 1.0f / sqrtf(cl2);
    const float ax0 = cB0 * icl, ax1 = cB1 * icl, ax2 = cB2 * icl;
    const float kt = sqrtf(fmaf(h0, h0, fmaf(h1, h1, h2 * h2))) * icl * 1.0001f;      // tan(half angle), rounded up
    const float ax_len = cl2 * icl;
    float l0 = 0.0f, l1 = 1.0f;
    // root box relative to P
    clip_ge(l0, l1, U0, (F.root_lo[0] - Pf0) - Pl0 - E); clip_le(l0, l1, L0, (F.root_hi[0] - Pf0) - Pl0 + E);
    clip_ge(l0, l1, U1, (F.root_lo[1] - Pf1) - Pl1 - E); clip_le(l0, l1, L1, (F.root_hi[1] - Pf1) - Pl1 + E);
    clip_ge(l0, l1, U2, (F.root_lo[2] - Pf2) - Pl2 - E); clip_le(l0, l1, L2, (F.root_hi[2] - Pf2) - Pl2 + E);
    uint32_t n = 0, nu = 0, boxmask = 0;             // candidates, units used, which units start a box
    uint32_t ids[RT_CMAX];                           // registers (static indices only): the pair's candidates so far
#pragma unroll
    for (uint32_t j = 0; j < RT_CMAX; ++j) ids[j] = 0xffffffffu;
    if (!(l0 <= l1)) return 0;
    // One loop, one memory access per trip: either a node step or a leaf entry.  Lanes of a wave sit at different
    // places of their own traversals; a flat loop costs the longest single traversal, not the sum of the longest phases.
    const uint32_t NONE = 0xffffffffu;
    uint32_t node = 0, e = 0, e_end = 0;
    int sp = 0;
    float blo0 = 0, blo1 = 0, blo2 = 0, bhi0 = 0, bhi1 = 0, bhi2 = 0;
    for (;;) {
        if (e < e_end) {
            // ---- one leaf entry ----
            const LeafF lf = F.leaves[e];
            ++e;
            const uint32_t kind = lf.pk & 3u, prim = lf.pk >> 2;
            float plo0, plo1, plo2, phi0, phi1, phi2;
            bool in_cone = true;
            if (kind == 1u) {
                const float c0 = (lf.x - Pf0) + (lf.lx - Pl0), c1 = (lf.y - Pf1) + (lf.ly - Pl1), c2 = (lf.z - Pf2) + (lf.lz - Pl2);
                plo0 = c0 - lf.r; plo1 = c1 - lf.r; plo2 = c2 - lf.r;
                phi0 = c0 + lf.r; phi1 = c1 + lf.r; phi2 = c2 + lf.r;
                // sphere against the cone: distance to the axis <= r (1 + tan) + axial position * tan (conservative)
                const float ta = fmaf(c0, ax0, fmaf(c1, ax1, c2 * ax2));
                const float cc = fmaf(c0, c0, fmaf(c1, c1, c2 * c2));
                const float dp2 = fmaxf(cc - ta * ta, 0.0f);
                const float lim = fmaf(lf.r, 1.0f + kt, fmaxf(ta, 0.0f) * kt) + E + 2e-6f * sqrtf(cc);
                in_cone = !(dp2 > lim * lim * 1.0001f) && !(ta < -(lf.r + E)) && !(ta > ax_len * (1.0f + kt) + lf.r + E);
            } else {
                plo0 = (lf.x - Pf0) - Pl0; plo1 = (lf.y - Pf1) - Pl1; plo2 = (lf.z - Pf2) - Pl2;
                phi0 = (lf.lx - Pf0) - Pl0; phi1 = (lf.ly - Pf1) - Pl1; phi2 = (lf.lz - Pf2) - Pl2;
            }
            const bool touch = !(plo0 > bhi0) && !(phi0 < blo0) && !(plo1 > bhi1) && !(phi1 < blo1) && !(plo2 > bhi2) && !(phi2 < blo2);
            if (!touch || !in_cone) continue;
            bool seen = false;                       // the same primitive is listed in several leaves
#pragma unroll
            for (uint32_t j = 0; j < RT_CMAX; ++j) seen = seen || (j < n && ids[j] == prim);
            if (seen) continue;
            if (nu + kind > RT_CUNITS) return RT_META_OVERFLOW;       // kind == units needed (sphere 1, box 2)
            const DPrim *pr = S.prims + prim;
            if (kind == 1u) {
                const double dx = pr->a[0] - P.x, dy = pr->a[1] - P.y, dz = pr->a[2] - P.z;
                out[nu] = make_float4((float)dx, (float)dy, (float)dz, (float)(((dx * dx + dy * dy) + dz * dz) - pr->b[0]));
                nu += 1;
            } else {
                out[nu] = make_float4(__double2float_rd(pr->a[0] - P.x), __double2float_rd(pr->a[1] - P.y),
                                      __double2float_rd(pr->a[2] - P.z), __double2float_ru(pr->b[0] - P.x));
                out[nu + 1] = make_float4(__double2float_ru(pr->b[1] - P.y), __double2float_ru(pr->b[2] - P.z), 0.0f, 0.0f);
                boxmask |= 1u << nu;
                nu += 2;
            }
#pragma unroll
            for (uint32_t j = 0; j < RT_CMAX; ++j) if (j == n) ids[j] = prim;
            ++n;
            continue;
        }
        if (node == NONE) {
            if (sp == 0) break;
            --sp;
            const uint2 t = stk.a[sp * STRIDE];
            node = t.x;
            l0 = __uint_as_float(t.y);
            l1 = stk.b[sp * STRIDE];
        }
        // ---- one node ----
        const KNode nd = F.knodes[node];
        if ((nd.b & 3u) == 3u) {
            // leaf: beam bounding box over this leaf's lambda range (slightly widened)
            const float m0 = fmaxf(l0 - 1e-6f, 0.0f), m1 = fminf(l1 + 1e-6f, 1.0f);
            blo0 = fminf(m0 * L0, m1 * L0) - E; bhi0 = fmaxf(m0 * U0, m1 * U0) + E;
            blo1 = fminf(m0 * L1, m1 * L1) - E; bhi1 = fmaxf(m0 * U1, m1 * U1) + E;
            blo2 = fminf(m0 * L2, m1 * L2) - E; bhi2 = fmaxf(m0 * U2, m1 * U2) + E;
            e = nd.a;
            e_end = nd.a + (nd.b >> 2);
            node = NONE;
            continue;
        }
        const uint32_t ax = nd.b & 3u;
        const float cut = (__uint_as_float(nd.a) - sel3(ax, Pf0, Pf1, Pf2)) - sel3(ax, Pl0, Pl1, Pl2);
        float a0 = l0, a1 = l1, b0 = l0, b1 = l1;
        clip_le(a0, a1, sel3(ax, L0, L1, L2), cut + E);      // left child: some beam point with x <= cut
        clip_ge(b0, b1, sel3(ax, U0, U1, U2), cut - E);      // right child
        const bool goL = a0 <= a1, goR = b0 <= b1;
        if (goL && goR) {
            stk.a[sp * STRIDE] = make_uint2(nd.b >> 2, __float_as_uint(b0));
            stk.b[sp * STRIDE] = b1;
            ++sp;
        }
        if (goL) { node = node + 1; l0 = a0; l1 = a1; }
        else if (goR) { node = nd.b >> 2; l0 = b0; l1 = b1; }
        else node = NONE;
    }
    return nu | (boxmask << 8);
}
