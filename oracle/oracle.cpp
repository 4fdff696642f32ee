/*
 * oracle.cpp - CPU restatement (float64) of the reference's per-pixel hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (ray-tracing_amd/,
 * ray_tracer.py) may import, link or call this file; it is used by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker and as
 * the timed CPU counterpart ("port").  The product fails loudly without its HIP
 * library; it never falls back to this code.
 *
 * Parity status: PINNED.  The reference ships no tests or golden vectors
 * (SURVEY.md §4), so this restatement is pinned against outputs of the
 * reference itself generated in the build container by tools/make_golden.py
 * (the .npz files under tests/golden/); tests/test_oracle.py checks every function below
 * against them.
 *
 * The reference is batch-vectorised NumPy; this file restates it PER RAY, in
 * the same operation order (no FMA contraction: build with -ffp-contract=off),
 * so that results agree with the reference to the last bit except where the
 * reference goes through BLAS (np.dot in InfinitePlane.py:77-78) or libm pow.
 * One deliberate deviation: SURVEY §8a Q4 (BSPNode.py:156 tests index VALUES;
 * batch-order dependent) is treated as a bug and not replicated; the goldens
 * carry a "q4fix" variant and the number of Q4 events.
 *
 * Every function cites the reference lines it follows (paths relative to
 * /root/reference).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <limits>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/rt_amd.h"

namespace {

const double EPSILON = 1e-6;               /* util.py:13 */
const double MAX_RENDER_DISTANCE = 1e8;    /* util.py:15 */
const double NaN = std::numeric_limits<double>::quiet_NaN();
const double INF = std::numeric_limits<double>::infinity();

struct V3 {
    double x, y, z;
};
inline V3 v3(const double *p) { return V3{p[0], p[1], p[2]}; }
inline V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
inline V3 operator*(double s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
inline V3 operator/(V3 a, double s) { return V3{a.x / s, a.y / s, a.z / s}; }
inline V3 neg(V3 a) { return V3{-a.x, -a.y, -a.z}; }
/* np.sum(a*b, axis=-1): sequential over the 3 components */
inline double dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
/* np.linalg.norm(x, axis=-1): sqrt(add.reduce(x*x)) */
inline double norm(V3 a) { return sqrt(dot(a, a)); }
inline bool isnan3_all(V3 a) { return std::isnan(a.x) && std::isnan(a.y) && std::isnan(a.z); }
inline bool isnan3_any(V3 a) { return std::isnan(a.x) || std::isnan(a.y) || std::isnan(a.z); }
/* np.minimum / np.maximum propagate NaN */
inline double np_min(double a, double b) { return (std::isnan(a) || std::isnan(b)) ? NaN : (a < b ? a : b); }
inline double np_max(double a, double b) { return (std::isnan(a) || std::isnan(b)) ? NaN : (a > b ? a : b); }

struct Counters {
    int64_t primary = 0, shadow = 0, reflect = 0, transmit = 0;
    int64_t hits = 0, node_tests = 0, sphere_tests = 0, box_tests = 0, plane_tests = 0, shade_lights = 0;
    /* share of the four test counters spent on shadow rays (subset of the totals above) */
    int64_t sh_node_tests = 0, sh_sphere_tests = 0, sh_box_tests = 0, sh_plane_tests = 0;
    void add(const Counters &o) {
        sh_node_tests += o.sh_node_tests; sh_sphere_tests += o.sh_sphere_tests; sh_box_tests += o.sh_box_tests;
        sh_plane_tests += o.sh_plane_tests;
        primary += o.primary; shadow += o.shadow; reflect += o.reflect; transmit += o.transmit;
        hits += o.hits; node_tests += o.node_tests; sphere_tests += o.sphere_tests; box_tests += o.box_tests;
        plane_tests += o.plane_tests; shade_lights += o.shade_lights;
    }
};

/* Sphere.intersect_vectorized, Sphere.py:63-107.  r2 = radius ** 2 from the host. */
V3 sphere_intersect(V3 o, V3 d, V3 c, double r2) {
    V3 p = o - c;
    double a = dot(d, d);
    double b = 2 * dot(d, p);
    double cc = dot(p, p) - r2;
    double disc = b * b - 4 * a * cc;
    if (!(0 <= disc)) return V3{NaN, NaN, NaN};
    double sq = sqrt(disc);
    double den = 1 / (2 * a);
    double s1 = (-b - sq) * den;
    double s2 = (-b + sq) * den;
    double smin = (s1 < s2) ? s1 : s2;
    if (smin < 0) smin = np_max(s1, s2);
    if (!(smin >= 0)) return V3{NaN, NaN, NaN};
    return o + smin * d;
}

/* Cube.cube_intersection_mask, Cube.py:95-138 */
bool slab_mask(V3 o, V3 d, V3 lo, V3 hi, bool outside_hits_only, double *t_near_out) {
    double inv[3], tmin[3], tmax[3];
    const double od[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
    const double l[3] = {lo.x, lo.y, lo.z}, h[3] = {hi.x, hi.y, hi.z};
    double t_near = 0, t_far = 0;
    for (int k = 0; k < 3; ++k) {
        inv[k] = (dd[k] != 0) ? 1.0 / dd[k] : INF;
        tmin[k] = (l[k] - od[k]) * inv[k];
        tmax[k] = (h[k] - od[k]) * inv[k];
        double t1 = np_min(tmin[k], tmax[k]);
        double t2 = np_max(tmin[k], tmax[k]);
        t_near = (k == 0) ? t1 : np_max(t_near, t1);
        t_far = (k == 0) ? t2 : np_min(t_far, t2);
    }
    if (t_near_out) *t_near_out = t_near;
    bool valid = t_near < t_far;
    if (outside_hits_only) valid = valid && (t_near > 0);
    else valid = valid && (t_far > 0);
    return valid;
}

/* Cube.intersect_vectorized, Cube.py:78-93 (lo/hi = position -/+ scale/2 from the host) */
V3 box_intersect(V3 o, V3 d, V3 lo, V3 hi) {
    double t;
    if (!slab_mask(o, d, lo, hi, true, &t)) return V3{NaN, NaN, NaN};
    return o + t * d;
}

/* InfinitePlane.intersect_vectorized, InfinitePlane.py:60-86.  n, dd are the
 * stored (normalised) normal and offset.  np.dot goes through BLAS in the
 * reference; here it is the plain left-to-right sum. */
V3 plane_intersect(V3 o, V3 d, V3 n, double dd) {
    double p0n = dot(o, n);
    double vn = dot(d, n);
    double t = -(p0n + dd) / vn;
    if ((t < 0) || (vn >= 0) || std::isnan(t)) return V3{NaN, NaN, NaN};
    return o + t * d;
}

/* Cube.calculate_normals, Cube.py:177-207; face centres from Cube.__init__, Cube.py:10-14 */
V3 box_normal(V3 P, V3 c, double scale) {
    double half = scale / 2;
    /* order: right(+x), up(+y), forward(+z), left(-x), down(-y), backward(-z) is the construction
       order; the distance stack (Cube.py:192) is right,left,up,down,forward,backward */
    V3 fc[6] = {V3{c.x + half * 1, c.y + half * 0, c.z + half * 0}, V3{c.x - half * 1, c.y - half * 0, c.z - half * 0},
                V3{c.x + half * 0, c.y + half * 1, c.z + half * 0}, V3{c.x - half * 0, c.y - half * 1, c.z - half * 0},
                V3{c.x + half * 0, c.y + half * 0, c.z + half * 1}, V3{c.x - half * 0, c.y - half * 0, c.z - half * 1}};
    int best = 0;
    double bd = 0;
    for (int i = 0; i < 6; ++i) { /* np.argmin: first minimum; the first NaN wins outright */
        double dist = norm(P - fc[i]);
        if (std::isnan(dist)) { best = i; break; }
        if (i == 0 || dist < bd) { bd = dist; best = i; }
    }
    return (fc[best] - c) / half;
}

struct Scene {
    const rt_scene_desc *s;
    explicit Scene(const rt_scene_desc *d) : s(d) {}
};

struct Best {
    V3 p;
    int64_t id;
    double dist;
};

/* ray_functions.compute_rays_interactions (ray_functions.py:46-79): strict-< update in list order */
inline void consider(Best &b, V3 o, V3 hit, int64_t id) {
    if (isnan3_all(hit)) return;
    double dist = norm(hit - o);
    if (dist < b.dist) {
        b.p = hit;
        b.dist = dist;
        b.id = id;
    }
}

/* the merge at every tree level (BSPNode.py:145-153,164-170,175-181,191-197): a child's result
 * replaces the current one iff no component is NaN and its distance is strictly smaller */
inline void merge(Best &into, V3 o, const Best &child) {
    if (isnan3_any(child.p)) return;
    double dist = norm(child.p - o);
    if (dist < into.dist) {
        into.p = child.p;
        into.id = child.id;
        into.dist = dist;
    }
}

void prim_test(const rt_scene_desc *s, int prim, V3 o, V3 d, Best &b, Counters &c) {
    if (prim < s->n_spheres) {
        c.sphere_tests++;
        consider(b, o, sphere_intersect(o, d, v3(s->sph_center + 3 * prim), s->sph_r2[prim]), s->sph_id[prim]);
    } else {
        int k = prim - s->n_spheres;
        c.box_tests++;
        consider(b, o, box_intersect(o, d, v3(s->box_lo + 3 * k), v3(s->box_hi + 3 * k)), s->box_id[k]);
    }
}

/* BSPNode.intersect_vectorize_rec, BSPNode.py:123-199, for one ray */
Best intersect_rec(const rt_scene_desc *s, int node, V3 o, V3 d, Counters &c) {
    Best best{V3{NaN, NaN, NaN}, 0, INF};
    c.node_tests++;
    bool relevant = slab_mask(o, d, v3(s->node_lo + 3 * node), v3(s->node_hi + 3 * node), false, nullptr);
    if (node == 0) {
        /* root only: inf_plains = [Background, planes...] (BSPNode.py:111,143-153) */
        Best pb{V3{NaN, NaN, NaN}, 0, INF};
        if (s->far_clip) consider(pb, o, V3{MAX_RENDER_DISTANCE, MAX_RENDER_DISTANCE, MAX_RENDER_DISTANCE}, 0);
        for (int k = 0; k < s->n_planes; ++k) {
            c.plane_tests++;
            consider(pb, o, plane_intersect(o, d, v3(s->pln_normal + 3 * k), s->pln_d[k]), s->pln_id[k]);
        }
        merge(best, o, pb);
    }
    if (!relevant) return best; /* Q4 not replicated: per-ray "this ray misses the node" */
    if (s->node_right[node] >= 0) {
        Best l = intersect_rec(s, node + 1, o, d, c);
        merge(best, o, l);
        Best r = intersect_rec(s, s->node_right[node], o, d, c);
        merge(best, o, r);
    } else {
        Best lb{V3{NaN, NaN, NaN}, 0, INF};
        int first = s->node_first[node], cnt = s->node_count[node];
        for (int e = 0; e < cnt; ++e) prim_test(s, s->leaf_prim[first + e], o, d, lb, c);
        merge(best, o, lb);
    }
    return best;
}

/* BSPNode.intersect_vectorize, BSPNode.py:201-210 */
inline Best nearest_hit(const rt_scene_desc *s, V3 o, V3 d, Counters &c) { return intersect_rec(s, 0, o, d, c); }

/* SurfaceAbs.get_surfaces_normals (SurfaceAbs.py:45-66): lookup by LIST POSITION = hit id (Q8) */
V3 surface_normal(const rt_scene_desc *s, int64_t id, V3 P) {
    if (id < 0 || id >= s->n_surfaces) return V3{0, 0, 0};
    int g = s->surf_geom[id];
    switch (s->surf_kind[id]) {
    case RT_SURF_SPHERE: { /* Sphere.py:121-133 */
        V3 dir = P - v3(s->sph_center + 3 * g);
        return dir / norm(dir);
    }
    case RT_SURF_BOX:
        return box_normal(P, v3(s->box_center + 3 * g), s->box_scale[g]);
    case RT_SURF_PLANE: /* InfinitePlane.py:91-92 */
        return v3(s->pln_normal + 3 * g);
    default: /* Background.py:35-44 */
        return V3{0, 0, 0};
    }
}

inline int material_of(const rt_scene_desc *s, int64_t id) { /* SurfaceAbs.py:69-87 */
    if (id < 0 || id >= s->n_surfaces) return 0;
    return s->surf_mat[id];
}

/* --- counter-based jitter (no counterpart in the reference, which uses the global NumPy stream,
 * Light.py:277-278; only the DISTRIBUTION - 6 independent U(-g,g) per shadow ray - is reproduced).
 * Must stay bit-identical to rt_hash3 in ray-tracing_amd/csrc. */
inline void pcg3d(uint32_t &x, uint32_t &y, uint32_t &z) {
    x = x * 1664525u + 1013904223u;
    y = y * 1664525u + 1013904223u;
    z = z * 1664525u + 1013904223u;
    x += y * z; y += z * x; z += x * y;
    x ^= x >> 16; y ^= y >> 16; z ^= z >> 16;
    x += y * z; y += z * x; z += x * y;
}
/* One hash per shadow ray; the high halves of (x, y, z) are the first draw of Light.py:277, the low halves the
 * second (16 bits per deviate: steps of g / 32768). */
/* path code of a child ray: the turns as a bit string under a leading 1 down to level 31, a hash of (parent code, turn) kept
 * above 2^31 beyond - the same function as child_path() in ray-tracing_amd/csrc/rt_device.hpp (restated, not included: the
 * oracle shares no code with the product) */
inline uint32_t child_path(uint32_t path, uint32_t mirror) {
    if (path < 0x80000000u) return path * 2u + mirror;
    uint32_t h = (path ^ (mirror * 0x85EBCA6Bu + 0x9E3779B1u)) * 0x2C1B3C6Du;
    h ^= h >> 15; h *= 0x297A2D39u; h ^= h >> 15;
    return h | 0x80000000u;
}
inline void jitter6(uint32_t seed, uint32_t pixel, uint32_t path, uint32_t light, uint32_t sample, double g,
                    double out[6]) {
    uint32_t x = pixel, y = path * 0x9E3779B1u + light, z = seed * 0x85EBCA6Bu + sample;
    pcg3d(x, y, z);
    const uint32_t h[6] = {x >> 16, y >> 16, z >> 16, x & 0xffffu, y & 0xffffu, z & 0xffffu};
    for (int k = 0; k < 6; ++k) {
        double u = (double)h[k] * (1.0 / 65536.0);
        out[k] = -g + (2 * g) * u;
    }
}

struct Ctx {
    const rt_scene_desc *s;
    int sh_rays;
    int jitter;
    uint32_t seed;
};

/* util.get_orthonormal_vector (util.py:113-132) + create_orthonormal_basis (util.py:65-75) */
void light_basis(V3 D, V3 &up, V3 &right) {
    up = V3{1, 1, 0};
    if (D.z != 0) up.z = -(1 * D.x + 1 * D.y) / D.z;
    up = up / norm(up);
    /* np.cross(up, D) */
    V3 r{up.y * D.z - up.z * D.y, up.z * D.x - up.x * D.z, up.x * D.y - up.y * D.x};
    right = r / (norm(r) + EPSILON);
}

/* Light.get_shadow_rays, Light.py:230-288, for one (hit, a, b) */
void shadow_ray(const rt_scene_desc *s, int m, V3 P, V3 up, V3 right, int n, int a, int b, const double *jit,
                V3 &src, V3 &dir) {
    double radius = s->lgt_radius[m];
    double g = radius / n;
    V3 lp = v3(s->lgt_pos + 3 * m);
    double c0 = (radius - g) / 2;
    V3 p00 = (lp + c0 * up) - c0 * right;
    src = (p00 - ((double)b * g) * up) + ((double)a * g) * right;
    if (jit) {
        V3 u1{jit[0], jit[1], jit[2]}, u2{jit[3], jit[4], jit[5]};
        src = src + (u1 * up + u2 * right);
    } else {
        /* np.zeros * up + np.zeros * right added as in Light.py:277-279 (keeps signed zeros identical) */
        V3 z{0, 0, 0};
        src = src + (z * up + z * right);
    }
    V3 vec = P - src;
    dir = vec / (norm(vec) + EPSILON);
}

/* Light.compute_shadows, Light.py:178-227: returns the light intensity factor (before clip) */
double shadow_factor(const Ctx &cx, int m, V3 L, V3 P, uint32_t pixel, uint32_t path, Counters &c) {
    const rt_scene_desc *s = cx.s;
    double si = s->lgt_shadow[m];
    if (si == 0) return 1.0;
    int n = cx.sh_rays;
    V3 D = neg(L), up, right;
    light_basis(D, up, right);
    int lit = 0;
    double g = s->lgt_radius[m] / n;
    for (int a = 0; a < n; ++a)
        for (int b = 0; b < n; ++b) {
            double jit[6];
            if (cx.jitter) jitter6(cx.seed, pixel, path, (uint32_t)m, (uint32_t)(a * n + b), g, jit);
            V3 src, dir;
            shadow_ray(s, m, P, up, right, n, a, b, cx.jitter ? jit : nullptr, src, dir);
            c.shadow++;
            const int64_t n0 = c.node_tests, s0 = c.sphere_tests, b0 = c.box_tests, p0 = c.plane_tests;
            Best h = nearest_hit(s, src, dir, c);
            c.sh_node_tests += c.node_tests - n0; c.sh_sphere_tests += c.sphere_tests - s0;
            c.sh_box_tests += c.box_tests - b0; c.sh_plane_tests += c.plane_tests - p0;
            double dd = norm(h.p - P);
            if (dd < EPSILON) lit++;
        }
    double frac = (double)lit / (double)(n * n);
    return (1.0 - si) + si * frac;
}

inline double clip01(double x) { return x < 0 ? 0 : (x > 1 ? 1 : x); } /* NaN stays NaN like np.clip */

/* np.nan_to_num(x, nan=0.0): +-inf become +-DBL_MAX */
inline double nan_to_num(double x) {
    if (std::isnan(x)) return 0.0;
    if (std::isinf(x)) return x > 0 ? std::numeric_limits<double>::max() : -std::numeric_limits<double>::max();
    return x;
}

/* ray_tracer.ray_tracing, ray_tracer.py:36-133, for one ray */
V3 trace(const Ctx &cx, V3 o, V3 d, int rec, uint32_t pixel, uint32_t path, Counters &c, int32_t *first_id,
         int kind = 0) {
    const rt_scene_desc *s = cx.s;
    V3 bg = v3(s->background);
    if (rec < 0) return bg;                               /* ray_tracer.py:62-63: no intersect call at all */
    if (kind == 1) c.transmit++;                          /* rays are counted when handed to intersect_vectorize */
    if (kind == 2) c.reflect++;
    Best h = nearest_hit(s, o, d, c);                      /* :70 */
    if (first_id) *first_id = (int32_t)h.id;
    if (h.id == 0) return bg;                              /* :72-75,132 */
    c.hits++;
    V3 P = h.p;
    V3 N = surface_normal(s, h.id, P);                     /* :77 */
    int mi = material_of(s, h.id);                         /* :78 */
    V3 Kd{0, 0, 0}, Ks{0, 0, 0}, Kr{0, 0, 0};
    double T = 0;
    int phong = 0;
    if (mi >= 0 && mi < s->n_materials) {                  /* Material.py:23-62 */
        Kd = v3(s->mat_kd + 3 * mi); Ks = v3(s->mat_ks + 3 * mi); Kr = v3(s->mat_kr + 3 * mi);
        T = s->mat_transparency[mi];
        phong = s->mat_phong[mi];
    }
    V3 diff{0, 0, 0}, spec{0, 0, 0};
    V3 V = neg(d);
    double nn = norm(N);
    V3 Nh = N / nn;                                        /* ray_functions.py:98-99 */
    for (int m = 0; m < s->n_lights; ++m) {
        c.shade_lights++;
        V3 dl = v3(s->lgt_pos + 3 * m) - P;                /* Light.py:35-53 */
        V3 L = dl / norm(dl);
        double I = clip01(shadow_factor(cx, m, L, P, pixel, path, c)); /* Light.py:80-84 */
        V3 col = v3(s->lgt_color + 3 * m);
        V3 Ims = (s->lgt_spec[m] * I) * col;               /* Light.py:87 */
        V3 Imd = I * col;                                  /* Light.py:91 */
        double ldn = dot(L, N);                            /* Light.py:123-125 */
        ldn = np_max(ldn, 0);
        V3 dterm = (ldn * Kd) * Imd;
        diff = (m == 0) ? dterm : diff + dterm;
        double dp = dot(L, Nh);                            /* ray_functions.py:101-103 */
        V3 R = (2 * dp) * Nh - L;
        double rv = dot(R, V);                             /* Light.py:169-171 */
        double pw = pow(rv, (double)phong);
        V3 sterm = (pw * Ks) * Ims;
        spec = (m == 0) ? sterm : spec + sterm;
    }
    spec = V3{nan_to_num(spec.x), nan_to_num(spec.y), nan_to_num(spec.z)}; /* Light.py:173 */
    V3 Tv{T, T, T};
    V3 base = (diff + spec) * V3{1.0 - T, 1.0 - T, 1.0 - T};       /* ray_tracer.py:105-106 */
    V3 through{0, 0, 0};
    if (T != 0) {                                                   /* :108-116 */
        V3 o2 = P + EPSILON * d;
        through = trace(cx, o2, d, rec - 1, pixel, child_path(path, 0u), c, nullptr, 1);
    }
    V3 back = through * Tv;                                         /* :117 */
    V3 refl{0, 0, 0};
    if (norm(Kr) != 0) {                                            /* :119-129 */
        double dp = dot(V, Nh);
        V3 Rd = (2 * dp) * Nh - V;
        V3 o2 = P + EPSILON * Rd;
        refl = trace(cx, o2, Rd, rec - 1, pixel, child_path(path, 1u), c, nullptr, 2);
    }
    refl = refl * Kr;                                               /* :130 */
    return (back + base) + refl;                                    /* :132 */
}

/* ray_functions.get_initial_rays, ray_functions.py:7-43, for pixel (i=row, j=col) */
void initial_ray(const rt_scene_desc *s, int W, int H, int i, int j, V3 &o, V3 &d) {
    double w = s->cam_screen_width;
    double h = w / W * H;
    double hg = h / H, wg = w / W;
    V3 pos = v3(s->cam_pos), X = v3(s->cam_x), Y = v3(s->cam_y), Z = v3(s->cam_z);
    V3 center = pos + s->cam_screen_distance * Z;
    V3 p00 = (center + ((h - hg) / 2) * Y) - ((w - wg) / 2) * X;
    V3 dest = (p00 - ((double)i * hg) * Y) + ((double)j * wg) * X;
    V3 dir = dest - pos;
    o = pos;
    d = dir / norm(dir);
}

/* pixel order of a rank's output (include/rt_amd.h, rt_tile_layout): row-major for world == 1, else the
 * rank's tiles t = rank, rank+world, ... each as a full tw x th block (-1 = padding outside the image) */
void tile_layout(const rt_render_params *p, std::vector<int64_t> &pix) {
    int W = p->width, H = p->height;
    int world = p->world > 1 ? p->world : 1, rank = p->world > 1 ? p->rank : 0;
    int tw = p->tile_w > 0 ? p->tile_w : 8, th = p->tile_h > 0 ? p->tile_h : 8;
    pix.clear();
    if (world == 1) {
        pix.resize((size_t)W * H);
        for (int64_t k = 0; k < (int64_t)W * H; ++k) pix[k] = k;
        return;
    }
    int tx = (W + tw - 1) / tw, ty = (H + th - 1) / th;
    int per_rank = (tx * ty + world - 1) / world;
    for (int j = 0; j < per_rank; ++j) {
        int t = rank + j * world;
        int x0 = (t % tx) * tw, y0 = (t / tx) * th;
        for (int y = y0; y < y0 + th; ++y)
            for (int x = x0; x < x0 + tw; ++x)
                pix.push_back((t < tx * ty && y < H && x < W) ? (int64_t)y * W + x : -1);
    }
}

void fill_counters(const Counters &c, int64_t *out) {
    if (!out) return;
    out[0] = c.primary; out[1] = c.shadow; out[2] = c.reflect; out[3] = c.transmit; out[4] = c.hits;
    out[5] = c.node_tests; out[6] = c.sphere_tests; out[7] = c.box_tests; out[8] = c.plane_tests;
    out[9] = c.shade_lights;
    out[10] = c.sh_node_tests; out[11] = c.sh_sphere_tests; out[12] = c.sh_box_tests; out[13] = c.sh_plane_tests;
}

}  // namespace

extern "C" {

/* counters_out[14]: primary, shadow, reflect, transmit rays, hits, node tests, sphere tests, box tests,
 * plane tests, (hit,light) shading terms, then the shadow-ray share of the four test counters */

int orc_intersect(const rt_scene_desc *s, const double *o, const double *d, int64_t n, double *pts, int64_t *ids,
                  int64_t *counters_out, int threads) {
    Counters total;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel
    {
        Counters c;
#pragma omp for schedule(dynamic, 256)
        for (int64_t k = 0; k < n; ++k) {
            Best b = nearest_hit(s, v3(o + 3 * k), v3(d + 3 * k), c);
            pts[3 * k] = b.p.x; pts[3 * k + 1] = b.p.y; pts[3 * k + 2] = b.p.z;
            ids[k] = b.id;
        }
#pragma omp critical
        total.add(c);
    }
    total.primary = n;
    fill_counters(total, counters_out);
    return 0;
}

/* rgb_f64 [npix*3] unclipped colours, rgb8 [npix*3] quantised like util.save_image (util.py:94-98),
 * ids [npix] primary hit ids; any may be NULL.  Pixel order = tile layout of the params. */
int orc_render(const rt_scene_desc *s, const rt_render_params *p, double *rgb_f64, uint8_t *rgb8, int32_t *ids,
               int64_t *counters_out, int threads) {
    std::vector<int64_t> pix;
    tile_layout(p, pix);
    Ctx cx{s, p->sh_rays == RT_USE_SCENE_VALUE ? s->sh_rays : p->sh_rays, p->jitter, p->seed};
    int rec = p->rec_max == RT_USE_SCENE_VALUE ? s->rec_max : p->rec_max;
    Counters total;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel
    {
        Counters c;
#pragma omp for schedule(dynamic, 64)
        for (int64_t k = 0; k < (int64_t)pix.size(); ++k) {
            if (pix[k] < 0) {   /* tile padding: stays zero */
                if (ids) ids[k] = 0;
                if (rgb_f64) { rgb_f64[3 * k] = 0; rgb_f64[3 * k + 1] = 0; rgb_f64[3 * k + 2] = 0; }
                if (rgb8) { rgb8[3 * k] = 0; rgb8[3 * k + 1] = 0; rgb8[3 * k + 2] = 0; }
                continue;
            }
            int i = (int)(pix[k] / p->width), j = (int)(pix[k] % p->width);
            V3 o, d;
            initial_ray(s, p->width, p->height, i, j, o, d);
            int32_t id0 = 0;
            c.primary++;
            V3 col = trace(cx, o, d, rec, (uint32_t)pix[k], 1u, c, &id0);
            if (ids) ids[k] = id0;
            if (rgb_f64) { rgb_f64[3 * k] = col.x; rgb_f64[3 * k + 1] = col.y; rgb_f64[3 * k + 2] = col.z; }
            if (rgb8) {
                const double cc[3] = {col.x, col.y, col.z};
                for (int ch = 0; ch < 3; ++ch) {
                    double v = clip01(cc[ch]) * 255;        /* ray_tracer.py:30, util.py:98 */
                    rgb8[3 * k + ch] = std::isnan(v) ? 0 : (uint8_t)v;
                }
            }
        }
#pragma omp critical
        total.add(c);
    }
    fill_counters(total, counters_out);
    return 0;
}

int orc_trace(const rt_scene_desc *s, const rt_render_params *p, const double *o, const double *d, int64_t n,
              double *rgb_out, int64_t *counters_out, int threads) {
    Ctx cx{s, p->sh_rays == RT_USE_SCENE_VALUE ? s->sh_rays : p->sh_rays, p->jitter, p->seed};
    int rec = p->rec_max == RT_USE_SCENE_VALUE ? s->rec_max : p->rec_max;
    Counters total;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel
    {
        Counters c;
#pragma omp for schedule(dynamic, 64)
        for (int64_t k = 0; k < n; ++k) {
            c.primary++;
            V3 col = trace(cx, v3(o + 3 * k), v3(d + 3 * k), rec, (uint32_t)k, 1u, c, nullptr);
            rgb_out[3 * k] = col.x; rgb_out[3 * k + 1] = col.y; rgb_out[3 * k + 2] = col.z;
        }
#pragma omp critical
        total.add(c);
    }
    fill_counters(total, counters_out);
    return 0;
}

int64_t orc_tile_layout(const rt_render_params *p, int64_t *pixel_index) {
    std::vector<int64_t> pix;
    tile_layout(p, pix);
    if (pixel_index) memcpy(pixel_index, pix.data(), pix.size() * sizeof(int64_t));
    return (int64_t)pix.size();
}

/* ---- unit-level entry points for the kernel-level goldens ---- */
void orc_initial_rays(const rt_scene_desc *s, int W, int H, double *src, double *dir) {
    for (int i = 0; i < H; ++i)
        for (int j = 0; j < W; ++j) {
            V3 o, d;
            initial_ray(s, W, H, i, j, o, d);
            int64_t k = (int64_t)i * W + j;
            src[3 * k] = o.x; src[3 * k + 1] = o.y; src[3 * k + 2] = o.z;
            dir[3 * k] = d.x; dir[3 * k + 1] = d.y; dir[3 * k + 2] = d.z;
        }
}
void orc_sphere_hits(const double *o, const double *d, int64_t n, const double *c, double r2, double *out) {
    for (int64_t k = 0; k < n; ++k) {
        V3 h = sphere_intersect(v3(o + 3 * k), v3(d + 3 * k), v3(c), r2);
        out[3 * k] = h.x; out[3 * k + 1] = h.y; out[3 * k + 2] = h.z;
    }
}
void orc_box_hits(const double *o, const double *d, int64_t n, const double *lo, const double *hi, double *out) {
    for (int64_t k = 0; k < n; ++k) {
        V3 h = box_intersect(v3(o + 3 * k), v3(d + 3 * k), v3(lo), v3(hi));
        out[3 * k] = h.x; out[3 * k + 1] = h.y; out[3 * k + 2] = h.z;
    }
}
void orc_plane_hits(const double *o, const double *d, int64_t n, const double *nrm, double dd, double *out) {
    for (int64_t k = 0; k < n; ++k) {
        V3 h = plane_intersect(v3(o + 3 * k), v3(d + 3 * k), v3(nrm), dd);
        out[3 * k] = h.x; out[3 * k + 1] = h.y; out[3 * k + 2] = h.z;
    }
}
void orc_slab_mask(const double *o, const double *d, int64_t n, const double *lo, const double *hi, int outside_only,
                   uint8_t *mask, double *t_near) {
    for (int64_t k = 0; k < n; ++k) {
        double t;
        mask[k] = slab_mask(v3(o + 3 * k), v3(d + 3 * k), v3(lo), v3(hi), outside_only != 0, &t) ? 1 : 0;
        if (t_near) t_near[k] = t;
    }
}
void orc_box_normals(const double *pts, int64_t n, const double *c, double scale, double *out) {
    for (int64_t k = 0; k < n; ++k) {
        V3 nn = box_normal(v3(pts + 3 * k), v3(c), scale);
        out[3 * k] = nn.x; out[3 * k + 1] = nn.y; out[3 * k + 2] = nn.z;
    }
}
void orc_sphere_normals(const double *pts, int64_t n, const double *c, double *out) {
    for (int64_t k = 0; k < n; ++k) {
        V3 dir = v3(pts + 3 * k) - v3(c);
        V3 nn = dir / norm(dir);
        out[3 * k] = nn.x; out[3 * k + 1] = nn.y; out[3 * k + 2] = nn.z;
    }
}
/* shadow-ray grid of one light for n hits, jitter 0: src/dir [n][s][s][3] in the reference's (hit, a, b) order */
void orc_shadow_grid(const double *hits, int64_t n, const double *light_pos, double radius, int sh, double *src,
                     double *dir) {
    rt_scene_desc s;
    memset(&s, 0, sizeof(s));
    s.n_lights = 1;
    s.lgt_pos = light_pos;
    s.lgt_radius = &radius;
    for (int64_t k = 0; k < n; ++k) {
        V3 P = v3(hits + 3 * k);
        V3 dl = v3(light_pos) - P;
        V3 L = dl / norm(dl);
        V3 up, right;
        light_basis(neg(L), up, right);
        for (int a = 0; a < sh; ++a)
            for (int b = 0; b < sh; ++b) {
                V3 sr, dr;
                shadow_ray(&s, 0, P, up, right, sh, a, b, nullptr, sr, dr);
                int64_t q = ((k * sh + a) * sh + b) * 3;
                src[q] = sr.x; src[q + 1] = sr.y; src[q + 2] = sr.z;
                dir[q] = dr.x; dir[q + 1] = dr.y; dir[q + 2] = dr.z;
            }
    }
}
/* Light.compute_diffuse_color / compute_specular_colors (Light.py:99-175) on explicit inputs:
 * N,V,Kd,Ks [n][3]; L,Imd,Ims [Lc][n][3]; alpha [n] */
void orc_shade_terms(int64_t n, int Lc, const double *N, const double *L, const double *V, const double *Kd,
                     const double *Ks, const double *Imd, const double *Ims, const int64_t *alpha, double *diffuse,
                     double *specular) {
    for (int64_t k = 0; k < n; ++k) {
        V3 Nk = v3(N + 3 * k), Vk = v3(V + 3 * k), kd = v3(Kd + 3 * k), ks = v3(Ks + 3 * k);
        V3 Nh = Nk / norm(Nk);
        V3 diff{0, 0, 0}, spec{0, 0, 0};
        for (int m = 0; m < Lc; ++m) {
            V3 Lm = v3(L + 3 * (m * n + k));
            V3 imd = v3(Imd + 3 * (m * n + k)), ims = v3(Ims + 3 * (m * n + k));
            double ldn = np_max(dot(Lm, Nk), 0);
            V3 dterm = (ldn * kd) * imd;
            diff = (m == 0) ? dterm : diff + dterm;
            double dp = dot(Lm, Nh);
            V3 R = (2 * dp) * Nh - Lm;
            double pw = pow(dot(R, Vk), (double)alpha[k]);
            V3 sterm = (pw * ks) * ims;
            spec = (m == 0) ? sterm : spec + sterm;
        }
        diffuse[3 * k] = diff.x; diffuse[3 * k + 1] = diff.y; diffuse[3 * k + 2] = diff.z;
        specular[3 * k] = nan_to_num(spec.x); specular[3 * k + 1] = nan_to_num(spec.y);
        specular[3 * k + 2] = nan_to_num(spec.z);
    }
}
void orc_jitter6(uint32_t seed, uint32_t pixel, uint32_t path, uint32_t light, uint32_t sample, double g,
                 double *out) {
    jitter6(seed, pixel, path, light, sample, g, out);
}

int orc_abi_version(void) { return RT_AMD_ABI_VERSION; }

}  // extern "C"
