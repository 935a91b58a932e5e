// host/pathtracer.cpp — see pathtracer.hpp.
#include "pathtracer.hpp"

#include <hip/hip_runtime_api.h>

#include <cstring>

namespace hpt {

static hpt_float4 f4(float x, float y, float z, float w) { hpt_float4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }

// create_triangle(v1,v2,v3): {v1, v2-v1, v3-v1}, w = 0 (reference pathtracer.cpp:46-58)
static hpt_triangle create_triangle(const Vec3f &v1, const Vec3f &v2, const Vec3f &v3) {
    Vec3f e1 = v2 - v1, e2 = v3 - v1;
    hpt_triangle t;
    t.v1 = f4(v1[0], v1[1], v1[2], 0.0f);
    t.e1 = f4(e1[0], e1[1], e1[2], 0.0f);
    t.e2 = f4(e2[0], e2[1], e2[2], 0.0f);
    return t;
}

Pathtracer::Pathtracer(int w, int h, int device) : width(w), height(h) {
    if (hpt_create(w, h, device, &ctx) != HPT_OK) { create_error = hpt_last_error(nullptr); ctx = nullptr; return; }
    image_data.assign((size_t)w * h, Pixel{0, 0, 0});
    sample_data.assign((size_t)w * h, Pixel{0, 0, 0});
}

Pathtracer::Pathtracer(int w, int h, const std::vector<int> &devices, bool sample_shards) : width(w), height(h) {
    if (hpt_group_create(w, h, devices.data(), (int)devices.size(), &group) != HPT_OK) { create_error = hpt_group_last_error(nullptr); group = nullptr; return; }
    if (sample_shards && hpt_group_set_sharding(group, HPT_SHARD_SAMPLES) != HPT_OK) {
        create_error = hpt_group_last_error(group);
        hpt_group_destroy(group); group = nullptr;
        return;
    }
    ctx = hpt_group_ctx(group, 0);
    image_data.assign((size_t)w * h, Pixel{0, 0, 0});
}

Pathtracer::~Pathtracer() {
    if (d_image) (void)hipFree(d_image);
    if (group) hpt_group_destroy(group);          // destroys its members, ctx among them
    else hpt_destroy(ctx);
}

const char *Pathtracer::error() const { return group ? hpt_group_last_error(group) : ctx ? hpt_last_error(ctx) : create_error.c_str(); }

// The reference's fixed scene (pathtracer.cpp:322-382): one diffuse material, two
// dead planes (type 4), a light sphere and two spheres.
int Pathtracer::set_scene() {
    auto prim = []() {
        hpt_primitive p;
        std::memset(&p, 0, sizeof p);
        p.refractive = f4(1.0f, 0, 0, 0);
        p.radius = 1.0f;
        return p;
    };
    std::vector<hpt_material> mats(1);
    std::memset(&mats[0], 0, sizeof mats[0]);
    mats[0].refractive = f4(1.0f, 0, 0, 0);
    mats[0].diffuse = f4(0.6f, 0.4f, 0.2f, 0.0f);
    std::vector<hpt_primitive> prims(5, prim());
    prims[0].type = f4(4, 0, 0, 0); prims[0].center = f4(0, 8, 0, 0); prims[0].plane_normal = f4(0, -1, 0, 0);
    prims[0].diffuse = f4(0.6f, 0.6f, 0.6f, 0);
    prims[4].type = f4(4, 0, 0, 0); prims[4].center = f4(0, 0, 1.5f, 0); prims[4].plane_normal = f4(0, 0, 1, 0);
    prims[4].diffuse = f4(0.8f, 0.8f, 0.95f, 0);
    prims[1].center = f4(-0.5f, 0.0f, 0, 0); prims[1].radius = .4f; prims[1].emissive = f4(10, 10, 10, 10);
    prims[2].center = f4(-0.5f, 0.6f, 0, 0); prims[2].radius = .4f; prims[2].diffuse = f4(0.7f, 0.7f, 0.2f, 0);
    prims[2].specular = f4(0.7f, 0.7f, 0.2f, 0); prims[2].refractive = f4(1.9f, 0, 0, 0.0f);
    prims[3].center = f4(0.5f, 0.6f, 0.0f, 0); prims[3].radius = .4f; prims[3].diffuse = f4(0.4f, 0.4f, 0.4f, 0);
    return set_scene(prims, mats);
}

int Pathtracer::set_scene(const std::vector<hpt_primitive> &prims, const std::vector<hpt_material> &mats) {
    if (!ctx) return 0;
    if (group) return hpt_group_set_scene(group, prims.data(), (int)prims.size(), mats.data(), (int)mats.size()) == HPT_OK;
    return hpt_set_scene(ctx, prims.data(), (int)prims.size(), mats.data(), (int)mats.size()) == HPT_OK;
}

// The reference's fixed camera (pathtracer.cpp:224-263): eye (0,0,-200) looking at the
// origin; after the sign flips n = (0,0,-1), u = (1,0,0), v = (0,1,0).
int Pathtracer::set_camera() {
    hpt_camera cam;
    cam.eye = f4(0, 0, -200, 0); cam.n = f4(0, 0, -1, 0); cam.u = f4(1, 0, 0, 0); cam.v = f4(0, 1, 0, 0);
    return set_camera(cam);
}

int Pathtracer::set_camera(const hpt_camera &cam) {
    if (!ctx) return 0;
    if (group) return hpt_group_set_camera(group, &cam) == HPT_OK;
    return hpt_set_camera(ctx, &cam) == HPT_OK;
}

int Pathtracer::set_config(const hpt_config &cfg) {
    if (!ctx) return 0;
    if (group) return hpt_group_set_config(group, &cfg) == HPT_OK;
    return hpt_set_config(ctx, &cfg) == HPT_OK;
}

// reference pathtracer.cpp:271-319
int Pathtracer::set_triangles(std::vector<Vec3f> vertices) {
    if (!ctx || vertices.size() == 0 || (vertices.size() % 3) != 0) return 0;
    std::vector<VectorTriangle> vt;
    vt.reserve(vertices.size() / 3);
    for (size_t i = 0; i < vertices.size(); i += 3) vt.push_back(VectorTriangle(vertices[i], vertices[i + 1], vertices[i + 2]));
    bvh_constructor = BoundingVolumeHierarchy(std::move(vt));
    if (!bvh_constructor.buildTree()) return 0;
    const std::vector<VectorTriangle> &ovt = bvh_constructor.get_triangles();
    triangles.resize(ovt.size());
    for (size_t i = 0; i < ovt.size(); i++) triangles[i] = create_triangle(ovt[i].v1, ovt[i].v2, ovt[i].v3);
    if (group)
        return hpt_group_upload_bvh(group, bvh_constructor.get_bvh_bbox(), bvh_constructor.get_bvh_info(), bvh_constructor.get_n_nodes()) == HPT_OK &&
               hpt_group_upload_triangles(group, triangles.data(), (int)triangles.size()) == HPT_OK;
    if (hpt_upload_bvh(ctx, bvh_constructor.get_bvh_bbox(), bvh_constructor.get_bvh_info(), bvh_constructor.get_n_nodes()) != HPT_OK) return 0;
    return hpt_upload_triangles(ctx, triangles.data(), (int)triangles.size()) == HPT_OK;
}

Pixel *Pathtracer::render() {
    if (!ctx) return nullptr;
    if (group) return render(1);
    if (hpt_render_sample(ctx, iterations, &image_data[0].red) != HPT_OK) return nullptr;
    iterations++;
    return image_data.data();
}

Pixel *Pathtracer::render(int n) {
    if (!ctx || n <= 0) return nullptr;
    if (group) {                      // every member renders its shard into its own device image; the exchange completes the frame
        if (hpt_group_render(group, iterations, iterations + n) != HPT_OK) return nullptr;
        if (hpt_group_read_image(group, &image_data[0].red) != HPT_OK) return nullptr;
        iterations += n;
        return image_data.data();
    }
    size_t bytes = image_data.size() * sizeof(Pixel);
    if (!d_image && hipMalloc((void **)&d_image, bytes) != hipSuccess) return nullptr;
    if (hipMemcpy(d_image, image_data.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    if (hpt_render(ctx, iterations, iterations + n, d_image) != HPT_OK) return nullptr;
    if (hpt_synchronize(ctx) != HPT_OK) return nullptr;
    if (hipMemcpy(image_data.data(), d_image, bytes, hipMemcpyDeviceToHost) != hipSuccess) return nullptr;
    iterations += n;
    return image_data.data();
}

// The live frame loop of main.cpp:121-151 with the semantics the reference intends, on
// device-resident state: hpt_first_bounce caches bounce 0, every iterative_render() advances each pixel's
// path by one bounce, reconstruct() folds the paths that ended into the per-pixel running mean.
void Pathtracer::iterative_render_first_bounce() {       // reference pathtracer.cpp:117-157
    if (!ctx || group) return;                           // the live loop runs on one device
    size_t bytes = image_data.size() * sizeof(Pixel);
    if (!d_image && hipMalloc((void **)&d_image, bytes) != hipSuccess) return;
    if (hipMemset(d_image, 0, bytes) != hipSuccess) return;
    for (Pixel &px : image_data) px.red = px.green = px.blue = 0.0f;
    (void)hpt_first_bounce(ctx);
}

void Pathtracer::iterative_render() {                    // reference pathtracer.cpp:159-176
    if (!ctx || group) return;
    (void)hpt_bounce(ctx);
}

void Pathtracer::reconstruct() {                         // reference pathtracer.cpp:178-182
    if (!ctx || !d_image) return;
    if (hpt_reconstruct_paths(ctx, d_image) != HPT_OK) return;
    if (hpt_synchronize(ctx) != HPT_OK) return;
    (void)hipMemcpy(image_data.data(), d_image, image_data.size() * sizeof(Pixel), hipMemcpyDeviceToHost);
}

}  // namespace hpt
