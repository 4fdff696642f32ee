// The whole library with AddressSanitizer on its HOST code (device code is not instrumented), driven through the C-ABI:
// scene text -> tables -> tree on the GPU -> context -> frames, batched queries, error paths -> destroy.
//   tools/sanitize/build_asan_host.sh            (here: hipcc cross-compiles)
//   ASAN_OPTIONS=detect_leaks=0:quarantine_size_mb=0:thread_local_quarantine_size_kb=0 tools/sanitize/asan_host tests/golden/scenes/*.txt     (on the GPU box)
// detect_leaks=0: the HIP runtime keeps process-lifetime allocations; leaks of this library are watched by
// tools/leak_check.py instead.  No quarantine: with one, ROCm's ASan runtime fails a CHECK of its own device allocator
// (sanitizer_allocator_device.h:125) as soon as any thread of the process exits.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../include/rt_amd.h"

#define MUST(expr) do { int rc_ = (expr); if (rc_ != RT_OK) { fprintf(stderr, "%s:%d %s -> %d\n", __FILE__, __LINE__, #expr, rc_); exit(2); } } while (0)
#define MUST_FAIL(expr) do { int rc_ = (expr); if (rc_ == RT_OK) { fprintf(stderr, "%s:%d %s succeeded\n", __FILE__, __LINE__, #expr); exit(3); } } while (0)

static rt_render_params params(int w, int h, int sh, int rec, int jitter, unsigned seed, int rank = 0, int world = 1, int flags = 0) {
    rt_render_params p = {};
    p.width = w; p.height = h; p.sh_rays = sh; p.rec_max = rec; p.jitter = jitter; p.seed = seed;
    p.rank = rank; p.world = world; p.flags = flags;
    return p;
}

int main(int argc, char **argv) {
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    for (int a = 1; a < argc; ++a) {
        rt_scene *sc = nullptr;
        MUST(rt_scene_parse_file(argv[a], 4, &sc));
        rt_bsp_stats bs;
        MUST(rt_scene_build_tree(sc, 0, &bs));
        const rt_scene_desc *d = rt_scene_get_desc(sc);
        rt_ctx *ctx = nullptr;
        MUST(rt_create(d, 0, &ctx));
        rt_stats st;
        // frames: whole image, odd sizes, partitions, diagnostics
        const int sizes[][2] = {{64, 48}, {1, 1}, {37, 5}, {130, 71}};
        for (auto &wh : sizes) {
            for (int world : {1, 3}) {
                rt_render_params p = params(wh[0], wh[1], RT_USE_SCENE_VALUE, RT_USE_SCENE_VALUE, 1, 5, world - 1, world);
                const int64_t n = rt_tile_pixels(&p);
                std::vector<float> rgb((size_t)n * 3);
                std::vector<uint8_t> rgb8((size_t)n * 3);
                std::vector<int32_t> ids((size_t)n);
                std::vector<int64_t> layout((size_t)n);
                MUST(rt_tile_layout(&p, layout.data()));
                MUST(rt_render(ctx, &p, rgb.data(), rgb8.data(), ids.data(), &st));
                MUST(rt_render(ctx, &p, nullptr, rgb8.data(), nullptr, nullptr));
                p.flags = RT_FLAG_EXACT_TRAVERSAL;
                MUST(rt_render(ctx, &p, rgb.data(), nullptr, ids.data(), &st));
                p.flags = RT_FLAG_NO_BEAM | RT_FLAG_PROFILE;
                MUST(rt_render(ctx, &p, rgb.data(), nullptr, nullptr, &st));
                p.flags = RT_FLAG_NO_OVERLAP;
                p.sh_rays = 16;
                MUST(rt_render(ctx, &p, nullptr, rgb8.data(), nullptr, &st));
            }
        }
        {
            rt_render_params p = params(48, 32, 3, 4, 1, 9);
            std::vector<float> rgb(48 * 32 * 3);
            std::vector<uint8_t> rgb8(48 * 32 * 3);
            MUST(rt_render_mean(ctx, &p, 5, rgb.data(), rgb8.data(), &st));
            MUST_FAIL(rt_render_mean(ctx, &p, 0, rgb.data(), rgb8.data(), &st));
        }
        {
            // frames in flight (the early camera pass: a second copy of the level-0 state, made on the way, freed with the context;
            // frame sizes that change re-carve the state while frames are pending)
            const int sizes[4][2] = {{96, 64}, {96, 64}, {200, 120}, {64, 40}};
            for (int round = 0; round < 2; ++round)
                for (int k = 0; k < 8; ++k) {
                    rt_render_params p = params(sizes[k % 4][0], sizes[k % 4][1], 5, 2 + k % 3, 1, 40 + k);
                    MUST(rt_render_device(ctx, &p, nullptr, nullptr, nullptr, nullptr));
                }
            int rc = rt_sync_stats(ctx, &st);
            if (rc != RT_OK && rc != RT_ERR_QUEUE_OVERFLOW) exit(5);
        }
        // batched queries
        const int64_t n = 3001;
        std::vector<double> o((size_t)n * 3), dir((size_t)n * 3), tg((size_t)n * 3), pts((size_t)n * 3), col((size_t)n * 3);
        std::vector<int64_t> ids((size_t)n);
        std::vector<uint8_t> lit((size_t)n);
        for (auto &v : o) v = 10.0 * U(rng);
        for (auto &v : dir) v = U(rng);
        for (auto &v : tg) v = 10.0 * U(rng);
        dir[0] = dir[1] = dir[2] = 0.0;
        dir[4] = NAN;
        MUST(rt_intersect(ctx, o.data(), dir.data(), n, pts.data(), ids.data()));
        MUST(rt_visibility(ctx, o.data(), dir.data(), tg.data(), n, lit.data()));
        MUST(rt_intersect(ctx, o.data(), dir.data(), 0, pts.data(), ids.data()));
        rt_render_params tp = params(1, 1, 3, 3, 1, 2);
        MUST(rt_trace(ctx, &tp, o.data(), dir.data(), n, col.data(), &st));
        std::vector<double> src(40 * 30 * 3), dd(40 * 30 * 3);
        MUST(rt_initial_rays(ctx, 40, 30, src.data(), dd.data()));
        // error paths
        rt_render_params bad = params(0, 10, 1, 1, 0, 0);
        MUST_FAIL(rt_render(ctx, &bad, nullptr, nullptr, nullptr, nullptr));
        bad = params(8, 8, 0, 1, 0, 0);
        if (d->n_lights > 0) (void)rt_render(ctx, &bad, nullptr, nullptr, nullptr, nullptr);
        bad = params(8, 8, 2000, 1, 0, 0);
        MUST_FAIL(rt_render(ctx, &bad, nullptr, nullptr, nullptr, nullptr));
        bad = params(8, 8, 1, 1, 0, 0, 5, 2);
        MUST_FAIL(rt_render(ctx, &bad, nullptr, nullptr, nullptr, nullptr));
        MUST_FAIL(rt_intersect(ctx, nullptr, dir.data(), n, pts.data(), ids.data()));
        MUST_FAIL(rt_visibility(ctx, o.data(), dir.data(), nullptr, n, lit.data()));
        rt_scene_desc broken = *d;
        broken.abi_version = 999;
        rt_ctx *c2 = nullptr;
        MUST_FAIL(rt_create(&broken, 0, &c2));
        broken = *d;
        broken.n_nodes = 0;
        MUST_FAIL(rt_create(&broken, 0, &c2));
        MUST_FAIL(rt_create(d, 99, &c2));
        if (!rt_last_error(nullptr) || !*rt_last_error(nullptr)) exit(4);
        printf("%s: ok (%d primitives, %d nodes)\n", argv[a], d->n_spheres + d->n_boxes + d->n_planes, (int)bs.n_nodes);
        MUST(rt_destroy(ctx));
        rt_scene_free(sc);
    }
    // the stand-alone tree build and the single-primitive kernels
    {
        const int64_t n = 5000;
        std::vector<double> lo((size_t)n * 3), hi((size_t)n * 3);
        for (int64_t i = 0; i < n * 3; ++i) { lo[(size_t)i] = 50.0 * U(rng); hi[(size_t)i] = lo[(size_t)i] + 0.3 + 0.3 * U(rng) * U(rng); }
        rt_bsp *t = nullptr;
        MUST(rt_bsp_build(0, n, lo.data(), hi.data(), nullptr, nullptr, 10, 1e-6, &t));
        rt_bsp_stats bs;
        MUST(rt_bsp_stats_get(t, &bs));
        std::vector<double> nlo((size_t)bs.n_nodes * 3), nhi((size_t)bs.n_nodes * 3);
        std::vector<int32_t> nr((size_t)bs.n_nodes), nf((size_t)bs.n_nodes), nc((size_t)bs.n_nodes), lp((size_t)bs.n_leaf_refs);
        MUST(rt_bsp_fetch(t, nlo.data(), nhi.data(), nr.data(), nf.data(), nc.data(), lp.data()));
        MUST(rt_bsp_destroy(t));
        MUST_FAIL(rt_bsp_build(0, 0, lo.data(), hi.data(), nullptr, nullptr, 10, 1e-6, &t));
        double par[12] = {0, 0, 0, 1.0, 1.0};
        std::vector<double> o(300), dd(300), out(300);
        for (auto &v : o) v = 3.0 * U(rng);
        for (auto &v : dd) v = U(rng);
        MUST(rt_unit_prim_intersect(0, RT_SURF_SPHERE, par, o.data(), dd.data(), 100, out.data()));
        MUST(rt_unit_prim_normals(0, RT_SURF_SPHERE, par, o.data(), 100, out.data()));
        double blo[3] = {-1, -1, -1}, bhi[3] = {1, 1, 1};
        std::vector<uint8_t> m(100);
        std::vector<double> tn(100);
        MUST(rt_unit_slab_mask(0, blo, bhi, 1, o.data(), dd.data(), 100, m.data(), tn.data()));
    }
    printf("host AddressSanitizer run complete\n");
    return 0;
}
