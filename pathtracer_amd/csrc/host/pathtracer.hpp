// host/pathtracer.hpp — the renderer facade, same public methods as the
// reference's class Pathtracer (reference pathtracer/pathtracer.hpp:55-76),
// driving the HIP backend through the C-ABI of include/hpt/hpt.h where the
// reference drives OpenCL_Manager (pathtracer.cpp:69-72, 142, 166, 180, 192, 301, 316).
#pragma once
#include <string>
#include <vector>

#include "../../../include/hpt/hpt.h"
#include "bvh.hpp"

namespace hpt {

typedef hpt_pixel Pixel;

class Pathtracer {
public:
    Pathtracer(int w, int h, int device = 0);
    // Several GPUs behind the same facade ("multi-GPU handled inside the context"): an hpt_group over
    // `devices` -- tile shards and one all-gather per render() (bit-identical images for any number of devices), or
    // with sample_shards iterations s % n == member and one all-reduce.  The live-loop calls need a single device.
    Pathtracer(int w, int h, const std::vector<int> &devices, bool sample_shards = false);
    ~Pathtracer();
    Pathtracer(const Pathtracer &) = delete;             // owns the backend context and the device image
    Pathtracer &operator=(const Pathtracer &) = delete;
    bool ok() const { return ctx != nullptr; }
    const char *error() const;

    // setters (reference: hard-coded scene and camera, pathtracer.cpp:224-263, 322-382)
    int set_scene();
    int set_scene(const std::vector<hpt_primitive> &prims, const std::vector<hpt_material> &mats);
    int set_camera();
    int set_camera(const hpt_camera &cam);
    int set_triangles(std::vector<Vec3f> vertices);
    int set_config(const hpt_config &cfg);

    // One more sample per pixel folded into the running mean; returns the image
    // (reference Pathtracer::render(), pathtracer.cpp:184-221).
    Pixel *render();
    // `n` samples in one device-resident batch; same image as n calls of render().
    Pixel *render(int n);

    // Progressive interface of the live GLUT loop (main.cpp:112, 130-131), with the
    // semantics the reference intends.  It runs with the same hpt_config as render(): by default
    // the mega-kernel's constants (diffuse factor 2*cos, depth limit 15), not ray_kernel.cl's own (factor 1, no
    // limit, ray_kernel.cl:785-786); set_config({.., .diffuse_scale = 1.0f, ..}) selects those.  The first call caches bounce 0
    // of every pixel, each iterative_render() advances every pixel's path by one bounce
    // (a path that ends restarts from the cache), reconstruct() folds the paths that
    // ended into the per-pixel running mean and refreshes get_image().
    void iterative_render_first_bounce();
    void iterative_render();
    void reconstruct();

    int get_width() const { return width; }
    int get_height() const { return height; }
    int get_iterations() const { return iterations; }
    Pixel *get_image() { return image_data.data(); }
    hpt_ctx *backend() { return ctx; }                  // (member 0 of a group)
    hpt_group *backend_group() { return group; }

private:
    hpt_ctx *ctx = nullptr;
    hpt_group *group = nullptr;              // several devices: owns the contexts (ctx = member 0, not to be destroyed by itself)
    int width, height, iterations = 0;
    std::vector<Pixel> image_data, sample_data;
    std::vector<hpt_triangle> triangles;
    BoundingVolumeHierarchy bvh_constructor;
    std::string create_error;
    float *d_image = nullptr;                // device-resident running mean for render(n)
};

}  // namespace hpt
