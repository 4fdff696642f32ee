// rt_device.hpp - device-side data layout and per-ray arithmetic (gfx950).
//
// Precision plan: every value that feeds a binary decision of
// the reference (hit / miss, nearest-of-two, the |X-P| < 1e-6 lit test, ray
// origins of secondary rays) is evaluated in float64 with the reference's own
// operation order - the translation unit is built with -ffp-contract=off, so a
// result is bit-identical to NumPy's wherever NumPy does not go through BLAS or
// libm.  float32 is used only for conservative CULLING (node slab tests,
// primitive early rejects): a float32 test may only say "cannot matter"; every
// candidate it lets through is decided in float64.  MI355X issues v_fma_f64 at
// half the v_fma_f32 rate, so the float64 confirmations (a few per ray) are cheap
// next to the float32 bulk.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RT_EPSILON 1e-6               // util.py:13
#define RT_MAX_RENDER_DISTANCE 1e8    // util.py:15

struct d3 {
    double x, y, z;
};
__device__ __forceinline__ d3 mk3(double x, double y, double z) { return d3{x, y, z}; }
__device__ __forceinline__ d3 ld3(const double *p) { return d3{p[0], p[1], p[2]}; }
__device__ __forceinline__ d3 operator+(d3 a, d3 b) { return d3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ d3 operator-(d3 a, d3 b) { return d3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ d3 operator*(d3 a, d3 b) { return d3{a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ d3 operator*(double s, d3 a) { return d3{s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ d3 operator/(d3 a, double s) { return d3{a.x / s, a.y / s, a.z / s}; }
__device__ __forceinline__ d3 neg3(d3 a) { return d3{-a.x, -a.y, -a.z}; }
// np.sum(a*b, axis=-1) / np.linalg.norm(x, axis=-1): sequential over the three components
__device__ __forceinline__ double dot3(d3 a, d3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ double norm3(d3 a) { return sqrt(dot3(a, a)); }
__device__ __forceinline__ bool nan3_any(d3 a) { return isnan(a.x) || isnan(a.y) || isnan(a.z); }
__device__ __forceinline__ bool nan3_all(d3 a) { return isnan(a.x) && isnan(a.y) && isnan(a.z); }
__device__ __forceinline__ double dnan() { return __longlong_as_double(0x7ff8000000000000LL); }
__device__ __forceinline__ double dinf() { return __longlong_as_double(0x7ff0000000000000LL); }

// ---- scene tables in HBM (built by rt_create) ---------------------------------------------
struct DPrim {            // 64 B: finite primitive, float64 (decisions)
    double a[3];          // sphere: centre           box: lo = position - scale/2
    double b[3];          // sphere: b[0] = radius**2 box: hi = position + scale/2
    int32_t kind;         // 1 sphere, 2 box
    int32_t id;           // reference surface.index
};
struct FPrim {            // 32 B: float32 shadow of DPrim for conservative rejects
    float a[3];           // sphere: centre, box: lo rounded down
    float r;              // sphere: radius rounded up
    float b[3];           // box: hi rounded up
    int32_t kind;
};
struct DNode {            // 64 B, DFS pre-order (left child = i + 1)
    double lo[3], hi[3];
    int32_t skip;         // next node in pre-order when this subtree is skipped
    int32_t first;        // leaf: first entry in leaf_prim, -1 for internal nodes
    int32_t count;
    int32_t pad;
};
struct FNode {            // 32 B: float32 bounds rounded outward
    float lo[3], hi[3];
    int32_t skip;
    int32_t first_count;  // leaf: (first << 8) | count when both fit, else -2 (read DNode); internal: -1
};
struct DPlane {
    double n[3], d;
    int32_t id, pad;
};
struct DSurf {            // by list position (Q8): what the shading stage reads for hit id k
    double p[4];          // sphere: centre; box: centre + scale; plane: normal
    int32_t kind, mat;
    int32_t mflags, pad;  // DMat::flags of the surface's material (0 when it has none): does a hit here spawn children?
};
struct DMat {
    double kd[3], ks[3], kr[3], T;
    int32_t phong;
    int32_t flags;        // bit0: T != 0, bit1: |Kr| != 0
};
struct DLight {
    double pos[3], color[3], spec, shadow, radius;
};

struct DScene {
    const DPrim *prims;
    const FPrim *fprims;
    const DNode *nodes;
    const FNode *fnodes;
    const int32_t *leaf_prim;
    const DPlane *planes;
    const DSurf *surfs;
    const DMat *mats;
    const DLight *lights;
    const int32_t *shadow_lights;   // light indices with shadow_intensity != 0
    int32_t n_prims, n_nodes, n_planes, n_surfs, n_mats, n_lights, n_shadow_lights, far_clip;
    double bg[3];
    float plane_kw;                 // >= |coordinate| of any light sample + |d| of any plane (sh_plane, rt_shadow.hpp)
};

struct Hit {
    d3 p;
    double dist;
    int32_t id;
};

// ---- reference-exact primitives --------------------------------------------------------------
// Cube.cube_intersection_mask (Cube.py:95-138).  inv = where(d != 0, 1/d, +inf) is per ray.
// np.minimum/np.maximum/np.max/np.min propagate NaN (0*inf when the origin lies in a slab plane
// and the direction component is 0): any NaN makes both t_near and t_far NaN -> reject.
__device__ __forceinline__ bool slab_exact(d3 o, d3 inv, const double *lo, const double *hi, bool outside_only,
                                           double &t_near) {
    double ax = (lo[0] - o.x) * inv.x, bx = (hi[0] - o.x) * inv.x;
    double ay = (lo[1] - o.y) * inv.y, by = (hi[1] - o.y) * inv.y;
    double az = (lo[2] - o.z) * inv.z, bz = (hi[2] - o.z) * inv.z;
    bool bad = isnan(ax) || isnan(bx) || isnan(ay) || isnan(by) || isnan(az) || isnan(bz);
    double n = fmax(fmax(fmin(ax, bx), fmin(ay, by)), fmin(az, bz));
    double f = fmin(fmin(fmax(ax, bx), fmax(ay, by)), fmax(az, bz));
    t_near = n;
    bool ok = (n < f) && (outside_only ? (n > 0) : (f > 0));
    return ok && !bad;
}

// Sphere.intersect_vectorized (Sphere.py:63-107); returns false on a miss, else the ray parameter
__device__ __forceinline__ bool sphere_exact(d3 o, d3 d, double a, const double *c, double r2, double &t) {
    d3 p = o - ld3(c);
    double b = 2 * dot3(d, p);
    double cc = dot3(p, p) - r2;
    double disc = b * b - 4 * a * cc;
    if (!(0 <= disc)) return false;
    double sq = sqrt(disc);
    double den = 1 / (2 * a);
    double s1 = (-b - sq) * den;
    double s2 = (-b + sq) * den;
    double smin = (s1 < s2) ? s1 : s2;
    if (smin < 0) smin = (isnan(s1) || isnan(s2)) ? dnan() : ((s1 > s2) ? s1 : s2);
    if (!(smin >= 0)) return false;
    t = smin;
    return true;
}

// InfinitePlane.intersect_vectorized (InfinitePlane.py:60-86)
__device__ __forceinline__ bool plane_exact(d3 o, d3 d, const DPlane &pl, double &t) {
    d3 n = ld3(pl.n);
    double p0n = dot3(o, n);
    double vn = dot3(d, n);
    double tt = -(p0n + pl.d) / vn;
    if ((tt < 0) || (vn >= 0) || isnan(tt)) return false;
    t = tt;
    return true;
}

// strict-< update in list order (ray_functions.py:61-77): the hit point is o + t*d and the
// distance is recomputed from the point, exactly as every level of BSPNode.py:146,164,175,191 does.
__device__ __forceinline__ void consider_t(Hit &best, d3 o, d3 d, double t, int32_t id) {
    d3 p = o + t * d;
    if (nan3_all(p)) return;
    double dist = norm3(p - o);
    if (dist < best.dist && !nan3_any(p)) {
        best.p = p;
        best.dist = dist;
        best.id = id;
    }
}

// Cube.calculate_normals (Cube.py:177-207): face whose CENTRE is nearest, first wins ties
__device__ __forceinline__ d3 box_normal(d3 P, d3 c, double scale) {
    double half = scale / 2;
    d3 fc[6] = {mk3(c.x + half, c.y, c.z), mk3(c.x - half, c.y, c.z), mk3(c.x, c.y + half, c.z),
                mk3(c.x, c.y - half, c.z), mk3(c.x, c.y, c.z + half), mk3(c.x, c.y, c.z - half)};
    int best = 0;
    double bd = 0;
    bool done = false;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double dist = norm3(P - fc[i]);
        if (!done) {
            if (isnan(dist)) { best = i; done = true; }
            else if (i == 0 || dist < bd) { bd = dist; best = i; }
        }
    }
    d3 f = fc[0];
#pragma unroll
    for (int i = 1; i < 6; ++i) if (best == i) f = fc[i];
    return (f - c) / half;
}

// SurfaceAbs.get_surfaces_normals (SurfaceAbs.py:45-66): lookup by list position = hit id
__device__ __forceinline__ d3 surface_normal(const DScene &S, int32_t id, d3 P, int32_t &mat) {
    mat = 0;
    if (id < 0 || id >= S.n_surfs) return mk3(0, 0, 0);
    DSurf s = S.surfs[id];
    mat = s.mat;
    if (s.kind == 1) {                       // Sphere.py:121-133
        d3 dir = P - mk3(s.p[0], s.p[1], s.p[2]);
        return dir / norm3(dir);
    }
    if (s.kind == 2) return box_normal(P, mk3(s.p[0], s.p[1], s.p[2]), s.p[3]);
    if (s.kind == 3) return mk3(s.p[0], s.p[1], s.p[2]);   // InfinitePlane.py:91-92
    return mk3(0, 0, 0);
}

// ---- jitter hash: must stay bit-identical to jitter6() in oracle/oracle.cpp ----------------
__device__ __forceinline__ void pcg3d(uint32_t &x, uint32_t &y, uint32_t &z) {
    x = x * 1664525u + 1013904223u;
    y = y * 1664525u + 1013904223u;
    z = z * 1664525u + 1013904223u;
    x += y * z; y += z * x; z += x * y;
    x ^= x >> 16; y ^= y >> 16; z ^= z >> 16;
    x += y * z; y += z * x; z += x * y;
}
// stream 0 / 1 = the two draws of Light.py:277-278; ONE hash per shadow ray, 16 bits per deviate
__device__ __forceinline__ d3 jitter3(uint32_t seed, uint32_t pixel, uint32_t path, uint32_t light, uint32_t sample,
                                      uint32_t stream, double g) {
    uint32_t x = pixel, y = path * 0x9E3779B1u + light, z = seed * 0x85EBCA6Bu + sample;
    pcg3d(x, y, z);
    if (stream == 0u) { x >>= 16; y >>= 16; z >>= 16; }
    else { x &= 0xffffu; y &= 0xffffu; z &= 0xffffu; }
    const double k = 1.0 / 65536.0;
    return mk3(-g + (2 * g) * ((double)x * k), -g + (2 * g) * ((double)y * k), -g + (2 * g) * ((double)z * k));
}

// Path code of a child ray (a key of the jitter hash): root 1, through 2k, mirror 2k + 1 - the path's turns as a bit string
// under a leading 1 - down to recursion level 31.  Deeper, the string no longer fits 32 bits (shifting on would drop the leading
// turns, and different paths of one pixel would draw the same deviations): from there the code is a hash of (parent code, turn),
// kept above 2^31 so that it cannot meet a shallow code.  Must stay identical to child_path() in oracle/oracle.cpp.
__device__ __forceinline__ uint32_t child_path(uint32_t path, uint32_t mirror) {
    if (path < 0x80000000u) return path * 2u + mirror;
    uint32_t h = (path ^ (mirror * 0x85EBCA6Bu + 0x9E3779B1u)) * 0x2C1B3C6Du;
    h ^= h >> 15; h *= 0x297A2D39u; h ^= h >> 15;
    return h | 0x80000000u;
}

// util.get_orthonormal_vector (util.py:113-132) + create_orthonormal_basis (util.py:65-75)
__device__ __forceinline__ void light_basis(d3 D, d3 &up, d3 &right) {
    up = mk3(1, 1, 0);
    if (D.z != 0) up.z = -(D.x + D.y) / D.z;
    up = up / norm3(up);
    d3 r = mk3(up.y * D.z - up.z * D.y, up.z * D.x - up.x * D.z, up.x * D.y - up.y * D.x);
    right = r / (norm3(r) + RT_EPSILON);
}

// integer power by squaring; np.power(float64, int) goes through libm pow, which agrees to a few ulp
__device__ __forceinline__ double ipow(double x, int32_t n) {
    uint32_t e = n < 0 ? (uint32_t)(-(int64_t)n) : (uint32_t)n;
    double r = 1.0, b = x;
    while (e) {
        if (e & 1u) r *= b;
        b *= b;
        e >>= 1;
    }
    return n < 0 ? 1.0 / r : r;
}
__device__ __forceinline__ double nan_to_num(double x) {   // np.nan_to_num(x, nan=0.0)
    if (isnan(x)) return 0.0;
    if (isinf(x)) return x > 0 ? 1.7976931348623157e308 : -1.7976931348623157e308;
    return x;
}
