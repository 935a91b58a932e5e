/* hpt/types.h — wire-format PODs of the render seam.
 *
 * Byte layouts are those of the reference's host/device mirror structs
 * (reference pathtracer/common.hpp:13-17, 116-186; device views in
 * pathtracer/ray_kernel.cl:35-54, 99-109, 157-161, 189-227).  They are what a
 * caller of OpenCL_Manager already holds, so the HIP backend takes them as is.
 * Plain C, no OpenCL / HIP / torch types.
 */
#ifndef HPT_TYPES_H
#define HPT_TYPES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* cl_float4 (16 B, 16-B aligned like the OpenCL typedef) */
typedef struct { float x, y, z, w; } __attribute__((aligned(16))) hpt_float4;

/* common.hpp:13-17 `Pixel` — 12 B RGB float */
typedef struct { float red, green, blue; } hpt_pixel;

/* common.hpp:126-129 `BVH_Node_BBox` — 24 B */
typedef struct {
    float minboundsx, minboundsy, minboundsz;
    float maxboundsx, maxboundsy, maxboundsz;
} hpt_bvh_bbox;

/* common.hpp:131-136 `BVH_Node_Info` — 16 B.  Depth-first order: the first
 * child of node k is k+1, the second is secondChildOffset (bvh.cpp:382-386).
 * Leaf: nPrimitives > 0 and offset = first triangle; `axis`/`secondChildOffset`
 * of a leaf and `offset` of an interior node are don't-care. */
typedef struct {
    int32_t offset;
    int32_t secondChildOffset;
    int32_t nPrimitives;
    int32_t axis;
} hpt_bvh_info;

/* common.hpp:168-172 `Triangle` — {v1, e1 = v2-v1, e2 = v3-v1}, w = 0; 48 B */
typedef struct { hpt_float4 v1, e1, e2; } hpt_triangle;

/* common.hpp:174-179 `Material` — 64 B.
 * refractive.x = IOR, refractive.w == 1 -> transmissive; emissive.y/z/w > 0 -> light */
typedef struct { hpt_float4 refractive, specular, diffuse, emissive; } hpt_material;

/* common.hpp:153-166 `Primitive` — 128 B. type.x: 0 sphere, 1 plane, other = never hit */
typedef struct {
    hpt_float4 refractive, specular, diffuse, emissive;
    hpt_float4 center;
    hpt_float4 type;
    hpt_float4 plane_normal;
    float radius;
    float p1, p2, p3;
} hpt_primitive;

/* common.hpp:181-186 `Camera` — 64 B */
typedef struct { hpt_float4 eye, n, u, v; } hpt_camera;

#ifdef __cplusplus
static_assert(sizeof(hpt_pixel) == 12, "Pixel");
static_assert(sizeof(hpt_bvh_bbox) == 24, "BVH_Node_BBox");
static_assert(sizeof(hpt_bvh_info) == 16, "BVH_Node_Info");
static_assert(sizeof(hpt_triangle) == 48, "Triangle");
static_assert(sizeof(hpt_material) == 64, "Material");
static_assert(sizeof(hpt_primitive) == 128, "Primitive");
static_assert(sizeof(hpt_camera) == 64, "Camera");
#else
_Static_assert(sizeof(hpt_pixel) == 12, "Pixel");
_Static_assert(sizeof(hpt_bvh_bbox) == 24, "BVH_Node_BBox");
_Static_assert(sizeof(hpt_bvh_info) == 16, "BVH_Node_Info");
_Static_assert(sizeof(hpt_triangle) == 48, "Triangle");
_Static_assert(sizeof(hpt_material) == 64, "Material");
_Static_assert(sizeof(hpt_primitive) == 128, "Primitive");
_Static_assert(sizeof(hpt_camera) == 64, "Camera");
#endif

/* Render constants that are compile-time literals in the reference
 * (SURVEY.md §5 "Config / flags"); defaults reproduce them. */
typedef struct {
    int32_t  max_depth;      /* MAXDEPTH, pathtracer_kernel.cl:11 (15); >= 1 */
    int32_t  rr_start;       /* RR applies when bounce index b > rr_start, pathtracer_kernel.cl:836 (3) */
    float    diffuse_scale;  /* factor on dot(n,dir) for a diffuse bounce: 2.0 in pathtracer_kernel.cl:822,
                                1.0 in ray_kernel.cl:785-786 */
    uint32_t seed;           /* global key of the counter-based RNG (replaces seed_memory, pathtracer.cpp:79-82) */
} hpt_config;

/* hit_id encoding of hpt_primary_hits: >= 0 triangle index in BVH leaf order
 * (node.offset + i, ray_kernel.cl:426); -1 miss; -(2+k) analytic primitive k. */
#define HPT_HIT_MISS (-1)
#define HPT_HIT_PRIMITIVE(k) (-(2 + (k)))

/* Counters and per-stage device times of the last calls (rays = closest-hit queries).  Times are
 * HIP-event times on the context's launch stream, in ms; a stage that has not run since hpt_create reads 0. */
typedef struct {
    uint64_t rays_primary;
    uint64_t rays_bounce;
    uint64_t paths;
    double   last_kernel_ms;   /* render_kernel launches of the last hpt_render / hpt_kernel_execute / hpt_render_sample, summed */
    int32_t  wide_nodes;       /* 64-B nodes resident on the device */
    int32_t  tree_depth;
    int32_t  grid_waves;       /* waves launched by the persistent kernel */
    int32_t  launches;         /* render_kernel launches the last render was cut into (sample buffer) */
    double   fold_ms;          /* fold_kernel launches of the same render, summed */
    double   trace_ms;         /* iter_trace_kernel of the last hpt_first_bounce / hpt_bounce */
    double   shade_ms;         /* iter_first_shade_kernel / iter_bounce_shade_kernel of the same call */
    double   reconstruct_ms;   /* the last hpt_reconstruct_paths or hpt_reconstruct */
    uint64_t lane_loads;       /* 16-B per-lane loads issued by the traversal and the shading of the counted rays (4 per node
                                  visit that is not served from the LDS copy of the tree's top, 5 per pair record / 3 per single record, 2 per shaded triangle hit): the unit of the
                                  vector-L1 gather roofline.  Counted only by contexts created with
                                  HPT_COUNT_LANE_LOADS=1 in the environment (a counting build of the kernel, 1.4 % slower);
                                  0 otherwise */
} hpt_stats;

#ifdef __cplusplus
}
#endif
#endif /* HPT_TYPES_H */
