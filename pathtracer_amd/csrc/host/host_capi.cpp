// host/host_capi.cpp — C entry points over the C++ host side (BVH builder, OBJ
// reader, image writer, Pathtracer facade) so that Python tests and bench.py
// can drive exactly the code a C++ application links.
#include <cstring>
#include <string>
#include <vector>

#include "bvh.hpp"
#include "image_writer.hpp"
#include "obj_reader.hpp"
#include "pathtracer.hpp"

using namespace hpt;

extern "C" {

// verts: n*9 floats. Outputs sized 2n-1 / 2n-1 / n / n. Returns n_nodes or -1 (n < 12).
int hpt_host_build_bvh(const float *verts, int n, hpt_bvh_bbox *bbox, hpt_bvh_info *info, int32_t *order,
                       hpt_triangle *triangles, int *depth) {
    if (n < 0) return -1;
    const int threads = build_threads();
    std::vector<VectorTriangle> vt((size_t)n);
    parallel_chunks(threads, n, [&](int, long b, long e) {
        for (long i = b; i < e; i++) {
            const float *p = verts + 9 * (size_t)i;
            vt[i] = VectorTriangle(Vec3f(p[0], p[1], p[2]), Vec3f(p[3], p[4], p[5]), Vec3f(p[6], p[7], p[8]));
        }
    });
    BoundingVolumeHierarchy b(std::move(vt));
    if (!b.buildTree()) return -1;
    int nn = b.get_n_nodes();
    std::memcpy(bbox, b.get_bvh_bbox(), sizeof(hpt_bvh_bbox) * (size_t)nn);
    std::memcpy(info, b.get_bvh_info(), sizeof(hpt_bvh_info) * (size_t)nn);
    std::memcpy(order, b.get_order().data(), sizeof(int32_t) * (size_t)n);
    if (triangles) {
        const std::vector<VectorTriangle> &o = b.get_triangles();
        parallel_chunks(threads, n, [&](int, long b, long e) {
            for (long i = b; i < e; i++) {                  // create_triangle, pathtracer.cpp:46-58
                hpt_triangle &t = triangles[i];
                const VectorTriangle &s = o[i];
                t.v1.x = s.v1[0]; t.v1.y = s.v1[1]; t.v1.z = s.v1[2]; t.v1.w = 0;
                t.e1.x = s.v2[0] - s.v1[0]; t.e1.y = s.v2[1] - s.v1[1]; t.e1.z = s.v2[2] - s.v1[2]; t.e1.w = 0;
                t.e2.x = s.v3[0] - s.v1[0]; t.e2.y = s.v3[1] - s.v1[1]; t.e2.z = s.v3[2] - s.v1[2]; t.e2.w = 0;
            }
        });
    }
    if (depth) *depth = b.get_depth();
    return nn;
}

// Returns the number of vertices (3 per triangle), -1 on failure.  out may be NULL to count: the parsed list
// is then kept (per thread) for the following call with the same path, so count + fetch parses the file once.
long hpt_host_load_obj(const char *path, float *out_xyz, long cap_vertices) {
    static thread_local std::string kept_path;
    static thread_local std::vector<Vec3f> kept;
    std::vector<Vec3f> v;
    if (out_xyz && !kept_path.empty() && kept_path == path) v.swap(kept);
    else if (!load_obj(path, v)) { kept_path.clear(); kept.clear(); return -1; }
    kept_path.clear(); kept.clear(); kept.shrink_to_fit();
    const long n = (long)v.size();
    if (out_xyz) {
        const long m = n < cap_vertices ? n : cap_vertices;
        parallel_chunks(build_threads(), m, [&](int, long b, long e) {
            for (long i = b; i < e; i++) { out_xyz[3 * i] = v[i][0]; out_xyz[3 * i + 1] = v[i][1]; out_xyz[3 * i + 2] = v[i][2]; }
        });
    } else {
        kept_path = path;
        kept.swap(v);
    }
    return n;
}

int hpt_host_save_png(const float *rgb, int width, int height, const char *path) {
    Bitmap b{(hpt_pixel *)rgb, width, height};
    return save_png_to_file(&b, path);
}
int hpt_host_save_ppm(const float *rgb, int width, int height, const char *path) {
    Bitmap b{(hpt_pixel *)rgb, width, height};
    return save_ppm_to_file(&b, path);
}
int hpt_host_save_png_rgb8(const uint8_t *rgb, int width, int height, const char *path) {
    return save_png_rgb8(rgb, width, height, path);
}

// ---- Pathtracer facade ----
void *hpt_pathtracer_new(int w, int h, int device) {
    Pathtracer *p = new Pathtracer(w, h, device);
    if (!p->ok()) { delete p; return nullptr; }
    return p;
}
// several devices behind the same facade (hpt_group); NULL on failure, text in hpt_group_last_error(NULL)
void *hpt_pathtracer_new_group(int w, int h, const int *devices, int n_devices, int sample_shards) {
    if (!devices || n_devices < 1) return nullptr;
    Pathtracer *p = new Pathtracer(w, h, std::vector<int>(devices, devices + n_devices), sample_shards != 0);
    if (!p->ok()) { delete p; return nullptr; }
    return p;
}
void hpt_pathtracer_delete(void *p) { delete (Pathtracer *)p; }
const char *hpt_pathtracer_error(void *p) { return ((Pathtracer *)p)->error(); }
int hpt_pathtracer_set_scene_default(void *p) { return ((Pathtracer *)p)->set_scene(); }
int hpt_pathtracer_set_scene(void *p, const hpt_primitive *prims, int np, const hpt_material *mats, int nm) {
    return ((Pathtracer *)p)->set_scene(std::vector<hpt_primitive>(prims, prims + np), std::vector<hpt_material>(mats, mats + nm));
}
int hpt_pathtracer_set_camera_default(void *p) { return ((Pathtracer *)p)->set_camera(); }
int hpt_pathtracer_set_camera(void *p, const hpt_camera *cam) { return ((Pathtracer *)p)->set_camera(*cam); }
int hpt_pathtracer_set_config(void *p, const hpt_config *cfg) { return ((Pathtracer *)p)->set_config(*cfg); }
int hpt_pathtracer_set_triangles(void *p, const float *xyz, long n_vertices) {
    std::vector<Vec3f> v((size_t)n_vertices);
    for (long i = 0; i < n_vertices; i++) v[i] = Vec3f(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
    return ((Pathtracer *)p)->set_triangles(std::move(v));
}
const float *hpt_pathtracer_render(void *p) { Pixel *r = ((Pathtracer *)p)->render(); return r ? &r->red : nullptr; }
const float *hpt_pathtracer_render_n(void *p, int n) { Pixel *r = ((Pathtracer *)p)->render(n); return r ? &r->red : nullptr; }
void hpt_pathtracer_iterative_render_first_bounce(void *p) { ((Pathtracer *)p)->iterative_render_first_bounce(); }
void hpt_pathtracer_iterative_render(void *p) { ((Pathtracer *)p)->iterative_render(); }
void hpt_pathtracer_reconstruct(void *p) { ((Pathtracer *)p)->reconstruct(); }
const float *hpt_pathtracer_get_image(void *p) { return &((Pathtracer *)p)->get_image()->red; }
int hpt_pathtracer_get_iterations(void *p) { return ((Pathtracer *)p)->get_iterations(); }
int hpt_pathtracer_get_width(void *p) { return ((Pathtracer *)p)->get_width(); }
int hpt_pathtracer_get_height(void *p) { return ((Pathtracer *)p)->get_height(); }
void *hpt_pathtracer_backend(void *p) { return ((Pathtracer *)p)->backend(); }

}  // extern "C"
