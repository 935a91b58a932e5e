/* hpt/host.h — C entry points of the host side that surrounds the render path: the order-exact BVH builder,
 * the OBJ reader and the image writer (C++ in pathtracer_amd/csrc/host/, exported from libhpt.so by
 * host/host_capi.cpp).  They replace the reference's CPU scene preparation and output, which are not part of
 * OpenCL_Manager but feed it and read from it:
 *   BoundingVolumeHierarchy::buildTree + get_bvh_bbox/get_bvh_info/get_triangles (reference bvh.cpp:141-390)
 *   create_triangle (pathtracer.cpp:46-58), load_obj (obj_reader.cpp:6-84), save_png_to_file (png_writer.cpp:11-116).
 * Plain C, host pointers only.
 */
#ifndef HPT_HOST_H
#define HPT_HOST_H

#include "types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Builds the reference's BVH (node and triangle ORDER identical to bvh.cpp:141-390) over n triangles given as
 * n*9 floats (three vertices each).  Outputs: bbox and info sized for 2n-1 nodes, order[n] = input index of the
 * k-th triangle in leaf order, triangles[n] = {v1, e1, e2} in leaf order (may be NULL), *depth = tree depth (may
 * be NULL).  Returns the node count, or -1 for n < 12 (the reference reads vtriangles[0..11], bvh.cpp:242-261).
 * Parallel over HPT_BUILD_THREADS host threads (default min(cores, 16)) with the sequential result. */
int hpt_host_build_bvh(const float *verts, int n, hpt_bvh_bbox *bbox, hpt_bvh_info *info, int32_t *order,
                       hpt_triangle *triangles, int *depth);

/* Reads a Wavefront OBJ (what the reference's load_obj reads, plus a/b/c face syntax, negative indices and
 * polygons as fans).  Returns the number of vertices (3 per triangle) or -1; out_xyz may be NULL to count, the
 * next call with the same path then returns the list without parsing again. */
long hpt_host_load_obj(const char *path, float *out_xyz, long cap_vertices);

/* Writes width*height Pixels through the reference's 8-bit conversion (uint8)(255*clip(c,0,1)), png_writer.cpp:7-17.
 * 0 on success. */
int hpt_host_save_png(const float *rgb, int width, int height, const char *path);
int hpt_host_save_ppm(const float *rgb, int width, int height, const char *path);
int hpt_host_save_png_rgb8(const uint8_t *rgb, int width, int height, const char *path);

#ifdef __cplusplus
}
#endif
#endif /* HPT_HOST_H */
