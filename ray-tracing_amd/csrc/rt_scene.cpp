// rt_scene.cpp - scene text -> rt_scene_desc without Python objects.
//
// Restates Parser.parse_scene_file (Parser.py:25-71) and the constructors it calls, emitting the SoA tables
// of include/rt_amd.h directly:
//   * one index counter shared by sph / pln / box AND lgt lines, starting at 1; surfaces[0] is the Background
//     (Parser.py:29,52-66); material ids count from 1, the background material (index 0) is made by the `set`
//     line and the list is sorted by index at the end (Parser.py:46-47,69);
//   * every derived constant is computed with the operations the Python objects use, in float64 and in the same
//     order, so the tables are bit-identical to scene_pack.pack_scene(parse_scene_file(...)):
//       sphere  radius ** 2 (libm pow, as CPython's float.__pow__), corners position -/+ radius (Sphere.py:135-139)
//       cube    position -/+ scale / 2 (Cube.py:89-90, 209-213)
//       plane   n / |n|, -offset / |n|  (InfinitePlane.py:8-10)
//       camera  look direction and the Gram-Schmidt basis of util.create_orthonormal_basis (Camera.py:11-13,
//               util.py:48-75, both normalisations with + EPSILON)
//     np.linalg.norm of a 1-D vector is a BLAS dot product whose 3-element tail is an FMA chain
//     (fma(z,z, fma(y,y, x*x))); norms taken along an axis are plain sums of squares.  Both are restated as such.
// The tokenizer splits the text at line ends into one piece per thread; the index counters are applied when
// the pieces are put together in file order.  tests/test_scene_native.py compares every table with the Python
// path on all fixture scenes and on random scene files.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rt_amd.h"

namespace {

thread_local std::string g_scene_error;

enum Kind : uint8_t { K_CAM, K_SET, K_MTL, K_SPH, K_PLN, K_BOX, K_LGT };
const int NEED[] = {11, 5, 11, 5, 5, 5, 9};      // parameters each line type reads (Parser.py:41-66)

struct Rec {
    double v[11];
    Kind kind;
};

struct Piece {
    const char *begin = nullptr, *end = nullptr;
    std::vector<Rec> recs;
    int64_t lines = 0;
    int64_t err_line = -1;          // 1-based inside the piece
    std::string err;
};

inline bool is_space(unsigned char c) { return c == ' ' || (c >= 9 && c <= 13) || (c >= 0x1c && c <= 0x1f); }
inline bool is_eol(char c) { return c == '\n' || c == '\r'; }

// Python's float(token): decimal / exponent forms, inf / nan, optional sign; the whole token must be used
bool parse_number(const char *b, const char *e, double &out) {
    if (b < e && *b == '+') {
        ++b;
        if (b < e && (*b == '+' || *b == '-')) return false;
    }
    if (b >= e) return false;
    auto r = std::from_chars(b, e, out, std::chars_format::general);
    if (r.ec == std::errc::result_out_of_range) {          // float('1e999') is inf, float('1e-999') is 0.0
        std::string tmp(b, e);
        char *endp = nullptr;
        out = strtod(tmp.c_str(), &endp);
        return endp == tmp.c_str() + tmp.size();
    }
    return r.ec == std::errc() && r.ptr == e;
}

void tokenize(Piece &p) {
    const char *s = p.begin;
    p.recs.reserve((size_t)(p.end - p.begin) / 48 + 4);
    while (s < p.end) {
        const char *eol = s;
        while (eol < p.end && !is_eol(*eol)) ++eol;
        ++p.lines;
        const char *c = s;
        while (c < eol && is_space((unsigned char)*c)) ++c;
        if (c < eol && *c != '#') {
            const char *t = c;
            while (t < eol && !is_space((unsigned char)*t)) ++t;
            const size_t n = (size_t)(t - c);
            int kind = -1;
            static const char *names[] = {"cam", "set", "mtl", "sph", "pln", "box", "lgt"};
            if (n == 3) {
                for (int k = 0; k < 7; ++k)
                    if (!memcmp(c, names[k], 3)) kind = k;
            }
            if (kind < 0) {
                p.err_line = p.lines;
                p.err = "Unknown object type: " + std::string(c, t);
                return;
            }
            Rec r;
            r.kind = (Kind)kind;
            int got = 0;
            c = t;
            while (true) {
                while (c < eol && is_space((unsigned char)*c)) ++c;
                if (c >= eol) break;
                t = c;
                while (t < eol && !is_space((unsigned char)*t)) ++t;
                double v;
                if (!parse_number(c, t, v)) {
                    p.err_line = p.lines;
                    p.err = "could not convert string to float: '" + std::string(c, t) + "'";
                    return;
                }
                if (got < 11) r.v[got] = v;
                ++got;
                c = t;
            }
            if (got < NEED[kind]) {
                p.err_line = p.lines;
                p.err = std::string("a '") + names[kind] + "' line needs " + std::to_string(NEED[kind]) + " numbers";
                return;
            }
            p.recs.push_back(r);
        }
        s = eol;
        if (s < p.end) {                                   // "\r\n" is one line end (universal newlines)
            if (*s == '\r' && s + 1 < p.end && s[1] == '\n') ++s;
            ++s;
        }
    }
}

inline double norm_dot3(const double *x) { return std::sqrt(std::fma(x[2], x[2], std::fma(x[1], x[1], x[0] * x[0]))); }
inline double norm_axis3(const double *x) { return std::sqrt((x[0] * x[0] + x[1] * x[1]) + x[2] * x[2]); }

inline int32_t py_int(double v) {                  // int(float): truncation toward zero
    if (!(v > -2147483648.0 && v < 2147483648.0)) return 0;
    return (int32_t)v;
}

}  // namespace

struct rt_scene {
    rt_scene_desc d;
    std::vector<int32_t> surf_kind, surf_geom, surf_mat, sph_id, box_id, pln_id, mat_phong, lgt_id, node_right, node_first,
        node_count, leaf_prim, prim_of_finite;
    std::vector<double> sph_center, sph_radius, sph_r2, box_center, box_scale, box_lo, box_hi, pln_normal, pln_d, mat_kd,
        mat_ks, mat_kr, mat_transparency, lgt_pos, lgt_color, lgt_spec, lgt_shadow, lgt_radius, node_lo, node_hi, small, big;
    std::vector<int32_t> mat_index;
    rt_scene_info info;
    rt_bsp_stats build;
    std::string error;

    void link() {
        d.abi_version = RT_AMD_ABI_VERSION;
        d.n_surfaces = (int32_t)surf_kind.size();
        d.surf_kind = surf_kind.data(); d.surf_geom = surf_geom.data(); d.surf_mat = surf_mat.data();
        d.n_spheres = (int32_t)sph_id.size();
        d.sph_center = sph_center.data(); d.sph_radius = sph_radius.data(); d.sph_r2 = sph_r2.data(); d.sph_id = sph_id.data();
        d.n_boxes = (int32_t)box_id.size();
        d.box_center = box_center.data(); d.box_scale = box_scale.data(); d.box_lo = box_lo.data(); d.box_hi = box_hi.data();
        d.box_id = box_id.data();
        d.n_planes = (int32_t)pln_id.size();
        d.pln_normal = pln_normal.data(); d.pln_d = pln_d.data(); d.pln_id = pln_id.data();
        d.n_materials = (int32_t)mat_phong.size();
        d.mat_kd = mat_kd.data(); d.mat_ks = mat_ks.data(); d.mat_kr = mat_kr.data(); d.mat_phong = mat_phong.data();
        d.mat_transparency = mat_transparency.data();
        d.n_lights = (int32_t)lgt_spec.size();
        d.lgt_pos = lgt_pos.data(); d.lgt_color = lgt_color.data(); d.lgt_spec = lgt_spec.data(); d.lgt_shadow = lgt_shadow.data();
        d.lgt_radius = lgt_radius.data();
        d.n_nodes = (int32_t)node_right.size();
        d.node_lo = node_lo.data(); d.node_hi = node_hi.data(); d.node_right = node_right.data(); d.node_first = node_first.data();
        d.node_count = node_count.data();
        d.n_leaf_refs = (int32_t)leaf_prim.size();
        d.leaf_prim = leaf_prim.data();
    }
};

namespace {

int sfail(int code, const std::string &msg) {
    g_scene_error = msg;
    return code;
}

void put3(std::vector<double> &dst, const double *v) { dst.insert(dst.end(), v, v + 3); }

void camera(rt_scene &s, const double *v) {
    // Camera.py:5-13
    const double *pos = v, *look_at = v + 3, *up = v + 6;
    double toward[3], z[3], y[3], x[3];
    for (int k = 0; k < 3; ++k) toward[k] = look_at[k] - pos[k];
    const double len = norm_dot3(toward);
    for (int k = 0; k < 3; ++k) z[k] = toward[k] / len;
    // util.create_orthonormal_basis(z, up): util.py:64-73
    const double along = (z[0] * up[0] + z[1] * up[1]) + z[2] * up[2];
    for (int k = 0; k < 3; ++k) y[k] = up[k] - along * z[k];
    const double ny = norm_axis3(y) + 1e-6;
    for (int k = 0; k < 3; ++k) y[k] = y[k] / ny;
    x[0] = y[1] * z[2] - y[2] * z[1];
    x[1] = y[2] * z[0] - y[0] * z[2];
    x[2] = y[0] * z[1] - y[1] * z[0];
    const double nx = norm_axis3(x) + 1e-6;
    for (int k = 0; k < 3; ++k) x[k] = x[k] / nx;
    for (int k = 0; k < 3; ++k) { s.d.cam_pos[k] = pos[k]; s.d.cam_x[k] = x[k]; s.d.cam_y[k] = y[k]; s.d.cam_z[k] = z[k]; }
    s.d.cam_screen_distance = v[9];
    s.d.cam_screen_width = v[10];
    s.info.has_camera = 1;
}

int assemble(rt_scene &s, std::vector<Piece> &pieces) {
    size_t total = 0;
    for (auto &p : pieces) total += p.recs.size();
    int64_t index = 1;            // Parser.py:31: shared by surfaces and lights
    int32_t mat = 1;
    struct MatRow { int32_t index; double v[11]; };
    std::vector<MatRow> mats;
    size_t per_kind[7] = {0, 0, 0, 0, 0, 0, 0};
    for (auto &p : pieces)
        for (const Rec &r : p.recs) per_kind[r.kind]++;
    const size_t n_sph = per_kind[K_SPH], n_box = per_kind[K_BOX], n_fin = n_sph + n_box;
    s.surf_kind.reserve(total + 1); s.surf_geom.reserve(total + 1); s.surf_mat.reserve(total + 1);
    s.sph_center.reserve(3 * n_sph); s.sph_radius.reserve(n_sph); s.sph_r2.reserve(n_sph); s.sph_id.reserve(n_sph);
    s.box_center.reserve(3 * n_box); s.box_scale.reserve(n_box); s.box_lo.reserve(3 * n_box); s.box_hi.reserve(3 * n_box);
    s.box_id.reserve(n_box);
    s.small.reserve(3 * n_fin); s.big.reserve(3 * n_fin); s.prim_of_finite.reserve(n_fin);
    s.surf_kind.push_back(RT_SURF_BACKGROUND);       // surfaces = [Background()]  (Parser.py:29)
    s.surf_geom.push_back(-1);
    s.surf_mat.push_back(0);
    s.d.far_clip = 1;
    volatile double two = 2.0;                        // keeps radius ** 2 a libm pow() call like CPython's
    for (auto &p : pieces)
        for (const Rec &r : p.recs) {
            const double *v = r.v;
            switch (r.kind) {
            case K_CAM: camera(s, v); break;
            case K_SET:
                for (int k = 0; k < 3; ++k) s.d.background[k] = v[k];
                s.d.sh_rays = py_int(v[3]);
                s.d.rec_max = py_int(v[4]);
                s.info.has_settings = 1;
                {   // the background material (Parser.py:46-47)
                    MatRow m{};
                    m.index = 0;
                    for (int k = 0; k < 3; ++k) m.v[k] = v[k];
                    mats.push_back(m);
                }
                break;
            case K_MTL: {
                MatRow m;
                m.index = mat++;
                memcpy(m.v, v, sizeof(m.v));
                mats.push_back(m);
                break;
            }
            case K_SPH: {
                if (index > 0x7ffffff0) return sfail(RT_ERR_UNSUPPORTED, "more than 2^31 indexed lines");
                s.surf_kind.push_back(RT_SURF_SPHERE);
                s.surf_geom.push_back((int32_t)s.sph_id.size());
                s.surf_mat.push_back(py_int(v[4]));
                put3(s.sph_center, v);
                s.sph_radius.push_back(v[3]);
                s.sph_r2.push_back(std::pow(v[3], two));
                s.sph_id.push_back((int32_t)index++);
                for (int k = 0; k < 3; ++k) { s.small.push_back(v[k] - v[3]); }
                for (int k = 0; k < 3; ++k) { s.big.push_back(v[k] + v[3]); }
                s.prim_of_finite.push_back(-(int32_t)s.sph_id.size());      // spheres: -(row+1), fixed up below
                break;
            }
            case K_BOX: {
                if (index > 0x7ffffff0) return sfail(RT_ERR_UNSUPPORTED, "more than 2^31 indexed lines");
                s.surf_kind.push_back(RT_SURF_BOX);
                s.surf_geom.push_back((int32_t)s.box_id.size());
                s.surf_mat.push_back(py_int(v[4]));
                put3(s.box_center, v);
                s.box_scale.push_back(v[3]);
                const double half = v[3] / 2;
                for (int k = 0; k < 3; ++k) { s.box_lo.push_back(v[k] - half); s.small.push_back(v[k] - half); }
                for (int k = 0; k < 3; ++k) { s.box_hi.push_back(v[k] + half); s.big.push_back(v[k] + half); }
                s.box_id.push_back((int32_t)index++);
                s.prim_of_finite.push_back((int32_t)s.box_id.size());       // boxes: +(row+1)
                break;
            }
            case K_PLN: {
                if (index > 0x7ffffff0) return sfail(RT_ERR_UNSUPPORTED, "more than 2^31 indexed lines");
                s.surf_kind.push_back(RT_SURF_PLANE);
                s.surf_geom.push_back((int32_t)s.pln_id.size());
                s.surf_mat.push_back(py_int(v[4]));
                const double len = norm_dot3(v);                            // InfinitePlane.py:8-10
                for (int k = 0; k < 3; ++k) s.pln_normal.push_back(v[k] / len);
                s.pln_d.push_back(-v[3] / len);
                s.pln_id.push_back((int32_t)index++);
                break;
            }
            case K_LGT:
                put3(s.lgt_pos, v);
                put3(s.lgt_color, v + 3);
                s.lgt_spec.push_back(v[6]);
                s.lgt_shadow.push_back(v[7]);
                s.lgt_radius.push_back(v[8]);
                s.lgt_id.push_back((int32_t)index++);
                break;
            }
        }
    const int32_t ns = (int32_t)s.sph_id.size();
    for (auto &p : s.prim_of_finite) p = p < 0 ? -p - 1 : ns + p - 1;      // spheres first, then boxes
    std::stable_sort(mats.begin(), mats.end(), [](const MatRow &a, const MatRow &b) { return a.index < b.index; });
    for (const MatRow &m : mats) {                                          // Material.py:4-11
        put3(s.mat_kd, m.v);
        put3(s.mat_ks, m.v + 3);
        put3(s.mat_kr, m.v + 6);
        s.mat_phong.push_back(py_int(m.v[9]));                             // Material.py:49,58: astype(int)
        s.mat_transparency.push_back(m.v[10]);
        s.mat_index.push_back(m.index);
    }
    s.info.n_finite = (int64_t)s.prim_of_finite.size();
    s.info.n_records = (int64_t)total;
    s.link();
    return RT_OK;
}

}  // namespace

extern "C" const char *rt_scene_last_error(void) { return g_scene_error.c_str(); }

extern "C" void rt_scene_free(rt_scene *s) { delete s; }

extern "C" int rt_scene_parse(const char *text, int64_t len, int threads, rt_scene **out) {
    if (!out) return sfail(RT_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (len < 0 || (!text && len)) return sfail(RT_ERR_INVALID_ARG, "text is NULL");
    if (threads <= 0) {
        const char *env = getenv("RT_AMD_PARSE_THREADS");
        threads = env ? atoi(env) : (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
    }
    threads = (int)std::max<int64_t>(1, std::min<int64_t>(threads, len / (1 << 20) + 1));
    std::vector<Piece> pieces((size_t)threads);
    const char *end = text + len, *at = text;
    for (int t = 0; t < threads; ++t) {
        const char *stop = t + 1 == threads ? end : text + len * (t + 1) / threads;
        if (stop < at) stop = at;
        auto line_start = [&](const char *p) { return p == text || p[-1] == '\n' || (p[-1] == '\r' && *p != '\n'); };
        while (stop < end && !line_start(stop)) ++stop;
        pieces[t].begin = at;
        pieces[t].end = stop;
        at = stop;
    }
    if (threads == 1) tokenize(pieces[0]);
    else {
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; ++t) pool.emplace_back(tokenize, std::ref(pieces[t]));
        for (auto &th : pool) th.join();
    }
    int64_t line0 = 0;
    for (auto &p : pieces) {
        if (p.err_line >= 0) return sfail(RT_ERR_INVALID_ARG, "line " + std::to_string(line0 + p.err_line) + ": " + p.err);
        line0 += p.lines;
    }
    rt_scene *s = new rt_scene();
    memset(&s->d, 0, sizeof(s->d));
    memset(&s->info, 0, sizeof(s->info));
    memset(&s->build, 0, sizeof(s->build));
    s->info.lines = line0;
    s->info.threads = threads;
    if (int rc = assemble(*s, pieces)) {
        delete s;
        return rc;
    }
    *out = s;
    return RT_OK;
}

extern "C" int rt_scene_parse_file(const char *path, int threads, rt_scene **out) {
    if (!out) return sfail(RT_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (!path) return sfail(RT_ERR_INVALID_ARG, "path is NULL");
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return sfail(RT_ERR_INVALID_ARG, std::string("cannot open ") + path);
    struct stat sb;
    if (fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > 0) {       // regular file: map it, no copy
        void *m = mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
        close(fd);
        if (m == MAP_FAILED) return sfail(RT_ERR_INVALID_ARG, std::string("cannot map ") + path);
        const int rc = rt_scene_parse((const char *)m, (int64_t)sb.st_size, threads, out);
        munmap(m, (size_t)sb.st_size);
        return rc;
    }
    std::string text;                                                          // pipes and empty files
    char buf[1 << 16];
    while (true) {
        const ssize_t n = read(fd, buf, sizeof(buf));
        if (n <= 0) break;
        text.append(buf, (size_t)n);
    }
    close(fd);
    return rt_scene_parse(text.data(), (int64_t)text.size(), threads, out);
}

extern "C" const rt_scene_desc *rt_scene_get_desc(const rt_scene *s) { return s ? &s->d : nullptr; }

extern "C" int rt_scene_get_info(const rt_scene *s, rt_scene_info *info) {
    if (!s || !info) return sfail(RT_ERR_INVALID_ARG, "NULL argument");
    *info = s->info;
    return RT_OK;
}

extern "C" int rt_scene_corners(const rt_scene *s, const double **small_corner, const double **big_corner, int64_t *n) {
    if (!s) return sfail(RT_ERR_INVALID_ARG, "scene is NULL");
    if (small_corner) *small_corner = s->small.data();
    if (big_corner) *big_corner = s->big.data();
    if (n) *n = s->info.n_finite;
    return RT_OK;
}

extern "C" int rt_scene_light_ids(const rt_scene *s, const int32_t **ids, const int32_t **material_ids) {
    if (!s) return sfail(RT_ERR_INVALID_ARG, "scene is NULL");
    if (ids) *ids = s->lgt_id.data();
    if (material_ids) *material_ids = s->mat_index.data();
    return RT_OK;
}

extern "C" int rt_scene_set_tree(rt_scene *s, int32_t n_nodes, const double *node_lo, const double *node_hi,
                                 const int32_t *node_right, const int32_t *node_first, const int32_t *node_count,
                                 int32_t n_leaf_refs, const int32_t *leaf_positions) {
    if (!s) return sfail(RT_ERR_INVALID_ARG, "scene is NULL");
    if (n_nodes <= 0 || n_leaf_refs < 0 || !node_lo || !node_hi || !node_right || !node_first || !node_count ||
        (n_leaf_refs && !leaf_positions))
        return sfail(RT_ERR_INVALID_ARG, "tree arrays are NULL or empty");
    const int64_t nf = s->info.n_finite;
    for (int32_t e = 0; e < n_leaf_refs; ++e)
        if (leaf_positions[e] < 0 || leaf_positions[e] >= nf) return sfail(RT_ERR_INVALID_ARG, "leaf position out of range");
    s->node_lo.assign(node_lo, node_lo + 3 * (size_t)n_nodes);
    s->node_hi.assign(node_hi, node_hi + 3 * (size_t)n_nodes);
    s->node_right.assign(node_right, node_right + n_nodes);
    s->node_first.assign(node_first, node_first + n_nodes);
    s->node_count.assign(node_count, node_count + n_nodes);
    s->leaf_prim.resize((size_t)n_leaf_refs);
    for (int32_t e = 0; e < n_leaf_refs; ++e) s->leaf_prim[e] = s->prim_of_finite[leaf_positions[e]];
    s->link();
    return RT_OK;
}

extern "C" int rt_scene_build_tree(rt_scene *s, int device, rt_bsp_stats *stats) {
    if (!s) return sfail(RT_ERR_INVALID_ARG, "scene is NULL");
    rt_bsp *t = nullptr;
    int rc = rt_bsp_build(device, s->info.n_finite, s->small.data(), s->big.data(), nullptr, nullptr, 10, 1e-6, &t);
    if (rc != RT_OK) return sfail(rc, rt_bsp_last_error(nullptr));
    rt_bsp_stats st;
    rt_bsp_stats_get(t, &st);
    s->node_lo.resize(3 * (size_t)st.n_nodes);
    s->node_hi.resize(3 * (size_t)st.n_nodes);
    s->node_right.resize((size_t)st.n_nodes);
    s->node_first.resize((size_t)st.n_nodes);
    s->node_count.resize((size_t)st.n_nodes);
    s->leaf_prim.resize((size_t)st.n_leaf_refs);
    rc = rt_bsp_fetch(t, s->node_lo.data(), s->node_hi.data(), s->node_right.data(), s->node_first.data(), s->node_count.data(),
                      s->leaf_prim.data());
    if (rc != RT_OK) {
        sfail(rc, rt_bsp_last_error(t));
        rt_bsp_destroy(t);
        return rc;
    }
    rt_bsp_destroy(t);
    for (auto &p : s->leaf_prim) p = s->prim_of_finite[p];                  // list positions -> primitive numbers
    s->build = st;
    if (stats) *stats = st;
    s->link();
    return RT_OK;
}
