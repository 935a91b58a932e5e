/* oracle/pt_oracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * Scalar CPU restatement of the reference's OpenCL kernels and CPU scene
 * preparation.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this; the product (pathtracer_amd/) never does.
 *
 * PARITY STATUS: "parity unpinned" against a real run of the reference.
 * The reference has no tests, golden vectors or fixtures, its
 * hot path is OpenCL C that needs a GPU OpenCL device (none here), and its CPU
 * scene-prep sources include macOS-only headers (<Opencl/cl.hpp>,
 * <OpenGL/gl.h>) that this image lacks, so they are unbuildable here without
 * stand-in headers.  Every function below cites the reference lines it follows.
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H
#include <stdint.h>
#include "../include/hpt/types.h"

typedef struct {
    const hpt_triangle  *triangles;   int32_t n_triangles;
    const hpt_bvh_bbox  *bbox;
    const hpt_bvh_info  *info;        int32_t n_nodes;
    const hpt_primitive *primitives;  int32_t n_primitives;
    const hpt_material  *materials;   int32_t n_materials;
    hpt_camera camera;
    int32_t width, height;
    hpt_config cfg;
} orc_scene;

typedef struct {
    uint64_t rays_primary, rays_bounce, paths;
    uint64_t nodes_primary, nodes_bounce;   /* get_node() calls, ray_kernel.cl:407 */
    uint64_t tris_primary, tris_bounce;     /* intersect_tri() calls, ray_kernel.cl:426 */
} orc_counters;

/* closest hit of the camera ray for pixels [px_begin, px_end) */
void orc_primary_hits(const orc_scene *s, int iteration, int px_begin, int px_end,
                      int32_t *hit_id, float *t, orc_counters *c);
/* one radiance sample per pixel (pathtracer_kernel.cl:855-950); rgb = 3 floats/pixel, full image */
void orc_kernel_execute(const orc_scene *s, int iteration, int px_begin, int px_end,
                        float *rgb, orc_counters *c, int n_threads);
/* running mean of Pathtracer::render() (pathtracer.cpp:200-217), `iterations` 1-based */
void orc_accumulate(float *image, const float *sample, int64_t n_floats, int iterations);
/* spp loop: image = running mean over iterations [spp_begin, spp_end) */
void orc_render(const orc_scene *s, int spp_begin, int spp_end, int px_begin, int px_end,
                float *image, orc_counters *c, int n_threads);
/* The iterative pipeline with its intended semantics (initial_ray_kernel.cl:802-902,
 * ray_kernel.cl:817-929, reconstruct_kernel.cl:10-37); see the comment above orc_iter_first_bounce. */
typedef struct {
    float o[3], d[3], weight[3], color[3];          /* current path of the pixel */
    uint32_t ctr; int32_t bounce;
    float c_o[3], c_d[3], c_weight[3], c_color[3];  /* cache: state after bounce 0 */
    uint32_t c_ctr; int32_t c_finished;
    int32_t completed, fresh;                       /* paths folded so far; contribution waits for reconstruct */
    float contribution[3];
} orc_iter_pixel;
typedef struct { orc_iter_pixel *px; uint64_t rays_primary, rays_bounce; } orc_iter_state;
void orc_iter_first_bounce(const orc_scene *s, orc_iter_state *st);
void orc_iter_bounce(const orc_scene *s, orc_iter_state *st);
void orc_iter_reconstruct(const orc_scene *s, orc_iter_state *st, float *image);

/* png_writer.cpp:7-17 */
void orc_tonemap(const float *image, uint8_t *rgb8, int64_t n_floats);

/* bvh.cpp:61-123, 141-390 + common.hpp:39-56, 94-106.  verts = n*9 floats.
 * Outputs (caller allocates): bbox/info [2n-1], order[n] = input index of the
 * k-th triangle in leaf order.  Returns n_nodes, or -1 if n < 12 (the
 * reference reads vtriangles[0..11] unconditionally, bvh.cpp:242-261). */
int orc_build_bvh(const float *verts, int n, hpt_bvh_bbox *bbox, hpt_bvh_info *info, int32_t *order);
/* pathtracer.cpp:46-58 */
void orc_make_triangles(const float *verts, const int32_t *order, int n, hpt_triangle *out);
/* obj_reader.cpp:6-84; returns number of vertices written (3 per face), -1 on open failure,
 * -2 on a face the reference parser rejects. out may be NULL to count. */
int64_t orc_load_obj(const char *path, float *out_xyz, int64_t cap_vertices);

/* unit probes for KATs */
int   orc_intersect_box(const float o[3], const float d[3], const float lo[3], const float hi[3], float t);
int   orc_intersect_tri(const float o[3], const float d[3], const hpt_triangle *tri, float *t);
int   orc_intersect_prim(const float o[3], const float d[3], const hpt_primitive *p, float *t);
float orc_rand(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t ctr);
void  orc_sincos(float phi, float *s, float *c);
void  orc_camera_ray(const orc_scene *s, int gid, int iteration, float o[3], float d[3]);
#endif
