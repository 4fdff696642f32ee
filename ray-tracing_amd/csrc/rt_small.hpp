// rt_small.hpp - small scenes staged in LDS: one fused kernel per recursion level.
//
// A scene of a few hundred primitives (pool.txt: 1 plane + 6 spheres in one leaf) has no use for the wavefront pipeline's
// five launches per level, its hit lists, pair records and float32 filters: the whole scene fits the LDS of a CU.  Each
// workgroup copies the scene tables (planes, tree nodes, leaf lists, primitives, surfaces, materials, lights) into LDS once
// and then, one lane per queued ray, does the level's work in registers:
//
//   nearest hit   the reference-order float64 walk (BSPNode.py:123-210) over the LDS-resident tables - the same arithmetic
//                 as nearest_hit_exact, so ids, hit points and tie-breaks are the reference's by construction;
//   soft shadows  for every shadow-casting light the s*s sample rays of Light.get_shadow_rays (Light.py:230-288), each
//                 decided by the same walk from the sample (Light.py:214-221);
//   Phong         Light.py:35-175, ray_tracer.py:77-106;
//   children      through / mirror rays appended to the next level's queue (ray_tracer.py:108-130).
//
// Used for s*s <= 4 (BASELINE config 2: pool.txt at 1920x1080 with one shadow ray per light) where the per-hit shadow
// work is a handful of rays; from 3x3 samples on the beam path's one-traversal-per-pair wins.  The framebuffer protocol
// is the pipeline's (k_resolve sums (fb0 + fb) + fbm): level 0 stores into fb0, deeper levels add their shading into fb
// and the background of their misses into fbm, so frames are bit-identical to every other mode.
#pragma once
#include "rt_device.hpp"

#define RT_LDS __attribute__((address_space(3)))
#define RT_SMALL_LDS_MAX (48 * 1024)      // scene tables beyond this stay in HBM (general pipeline)

struct SmallLayout {         // byte offsets of the tables inside the workgroup's LDS copy (all multiples of 16)
    uint32_t planes, nodes, leaf, prims, surfs, mats, lights, shadow_lights, total;
};

struct LdsScene {            // the scene, read from LDS
    const RT_LDS DPlane *planes;
    const RT_LDS DNode *nodes;
    const RT_LDS int32_t *leaf_prim;
    const RT_LDS DPrim *prims;
    const RT_LDS DSurf *surfs;
    const RT_LDS DMat *mats;
    const RT_LDS DLight *lights;
    const RT_LDS int32_t *shadow_lights;
};

typedef uint32_t rt_u4 __attribute__((ext_vector_type(4)));
typedef uint32_t rt_u2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void lds_copy(RT_LDS unsigned char *dst, const void *src, uint32_t bytes) {
    const rt_u4 *s = (const rt_u4 *)src;
    RT_LDS rt_u4 *d = (RT_LDS rt_u4 *)dst;
    for (uint32_t i = threadIdx.x; i < (bytes + 15u) / 16u; i += blockDim.x) d[i] = s[i];      // (device allocations are 256-byte granular: the tail read stays inside)
}

// one record from LDS into registers (the records are plain structs of 8-byte multiples; C++ copy construction does not
// take a __shared__-qualified source)
template <typename T>
__device__ __forceinline__ T lds_load(const RT_LDS T *p) {
    static_assert(sizeof(T) % 8 == 0, "record size");
    T out;
    const RT_LDS rt_u2 *w = (const RT_LDS rt_u2 *)p;
    rt_u2 *o = (rt_u2 *)&out;
#pragma unroll
    for (int i = 0; i < (int)(sizeof(T) / 8); ++i) o[i] = w[i];
    return out;
}

// stage the scene; every thread of the workgroup calls this, ends with a barrier
__device__ __forceinline__ LdsScene stage_scene(const DScene &S, const SmallLayout &L, RT_LDS unsigned char *lds) {
    lds_copy(lds + L.planes, S.planes, (uint32_t)S.n_planes * (uint32_t)sizeof(DPlane));
    lds_copy(lds + L.nodes, S.nodes, (uint32_t)S.n_nodes * (uint32_t)sizeof(DNode));
    lds_copy(lds + L.leaf, S.leaf_prim, L.prims - L.leaf);
    lds_copy(lds + L.prims, S.prims, (uint32_t)S.n_prims * (uint32_t)sizeof(DPrim));
    lds_copy(lds + L.surfs, S.surfs, (uint32_t)S.n_surfs * (uint32_t)sizeof(DSurf));
    lds_copy(lds + L.mats, S.mats, (uint32_t)S.n_mats * (uint32_t)sizeof(DMat));
    lds_copy(lds + L.lights, S.lights, (uint32_t)S.n_lights * (uint32_t)sizeof(DLight));
    lds_copy(lds + L.shadow_lights, S.shadow_lights, L.total - L.shadow_lights);
    __syncthreads();
    LdsScene V;
    V.planes = (const RT_LDS DPlane *)(lds + L.planes);
    V.nodes = (const RT_LDS DNode *)(lds + L.nodes);
    V.leaf_prim = (const RT_LDS int32_t *)(lds + L.leaf);
    V.prims = (const RT_LDS DPrim *)(lds + L.prims);
    V.surfs = (const RT_LDS DSurf *)(lds + L.surfs);
    V.mats = (const RT_LDS DMat *)(lds + L.mats);
    V.lights = (const RT_LDS DLight *)(lds + L.lights);
    V.shadow_lights = (const RT_LDS int32_t *)(lds + L.shadow_lights);
    return V;
}

// nearest_hit_exact (rt_amd.hip) over the LDS copy: same order, same arithmetic.  Records are copied to registers first
// (the shared helpers take generic pointers / references).
__device__ __forceinline__ Hit nearest_hit_small(const DScene &S, const LdsScene &V, d3 o, d3 d) {
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
    for (int k = 0; k < S.n_planes; ++k) {
        const DPlane pl = lds_load(V.planes + k);
        double t;
        if (plane_exact(o, d, pl, t)) consider_t(best, o, d, t, pl.id);
    }
    const d3 inv = mk3(d.x != 0 ? 1.0 / d.x : dinf(), d.y != 0 ? 1.0 / d.y : dinf(), d.z != 0 ? 1.0 / d.z : dinf());
    const double a = dot3(d, d);
    int i = 0;
    const int n = S.n_nodes;
    while (i < n) {
        const DNode nd = lds_load(V.nodes + i);
        double tn;
        if (!slab_exact(o, inv, nd.lo, nd.hi, false, tn)) {
            i = nd.skip;
            continue;
        }
        if (nd.first >= 0) {
            for (int e = 0; e < nd.count; ++e) {
                const DPrim pr = lds_load(V.prims + V.leaf_prim[nd.first + e]);
                double t;
                if (pr.kind == 1) {
                    if (sphere_exact(o, d, a, pr.a, pr.b[0], t)) consider_t(best, o, d, t, pr.id);
                } else {
                    if (slab_exact(o, inv, pr.a, pr.b, true, t)) consider_t(best, o, d, t, pr.id);
                }
            }
        }
        i = i + 1;
    }
    return best;
}

// SurfaceAbs.get_surfaces_normals (SurfaceAbs.py:45-66) from the LDS copy
__device__ __forceinline__ d3 surface_normal_small(const DScene &S, const LdsScene &V, int32_t id, d3 P, int32_t &mat) {
    mat = 0;
    if (id < 0 || id >= S.n_surfs) return mk3(0, 0, 0);
    const DSurf s = lds_load(V.surfs + id);
    mat = s.mat;
    if (s.kind == 1) {
        d3 dir = P - mk3(s.p[0], s.p[1], s.p[2]);
        return dir / norm3(dir);
    }
    if (s.kind == 2) return box_normal(P, mk3(s.p[0], s.p[1], s.p[2]), s.p[3]);
    if (s.kind == 3) return mk3(s.p[0], s.p[1], s.p[2]);
    return mk3(0, 0, 0);
}
