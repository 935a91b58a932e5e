// hpt_comm.hip — the multi-GPU exchange behind the C-ABI (include/hpt/hpt.h, "Multi-GPU exchange inside the library"):
// per-context communicators for one process per GPU, and hpt_group for one process that drives several GPUs.
//
// pixels and samples are independent, scene and BVH are replicated, and the only data that crosses GPUs
// is the accumulation buffer -- disjoint tile slabs through ONE ncclAllGather, or (n_r / N)-weighted running means
// through ONE ncclAllReduce.  The reference has no counterpart (platform 0 / device 0 only, opencl_manager.cpp:57-73).
//
// RCCL is reached through dlopen on first use: a one-GPU caller never loads it, and inside a process that already
// holds an RCCL (PyTorch ships its own librccl.so.1) the loader hands back that copy, so that one HIP runtime and one
// RCCL serve both.  Collectives are queued on the context's stream: in order with the render, no host synchronisation.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "hpt_internal.h"

namespace {

// ---- RCCL, loaded on first use ---------------------------------------------------------------------------------
struct Rccl {
    void *handle = nullptr;
    std::string error;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

Rccl g_rccl;
std::once_flag g_rccl_once;

template <class F>
bool bind_symbol(F &fn, const char *name) {
    fn = reinterpret_cast<F>(dlsym(g_rccl.handle, name));
    if (!fn) g_rccl.error = std::string("librccl: missing symbol ") + name;
    return fn != nullptr;
}

// nullptr when RCCL cannot be loaded (g_rccl.error says why)
Rccl *rccl() {
    std::call_once(g_rccl_once, [] {
        const char *names[] = {getenv("HPT_RCCL_LIB"), "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
        for (const char *n : names) {
            if (!n || !*n) continue;
            g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (g_rccl.handle) break;
            const char *e = dlerror();
            g_rccl.error = std::string("dlopen ") + n + ": " + (e ? e : "failed");
        }
        if (!g_rccl.handle) return;
        bool ok = bind_symbol(g_rccl.GetUniqueId, "ncclGetUniqueId") && bind_symbol(g_rccl.CommInitRank, "ncclCommInitRank") &&
                  bind_symbol(g_rccl.CommInitAll, "ncclCommInitAll") && bind_symbol(g_rccl.CommDestroy, "ncclCommDestroy") &&
                  bind_symbol(g_rccl.AllGather, "ncclAllGather") && bind_symbol(g_rccl.AllReduce, "ncclAllReduce") &&
                  bind_symbol(g_rccl.GroupStart, "ncclGroupStart") && bind_symbol(g_rccl.GroupEnd, "ncclGroupEnd") &&
                  bind_symbol(g_rccl.GetErrorString, "ncclGetErrorString");
        if (!ok) { dlclose(g_rccl.handle); g_rccl.handle = nullptr; }
    });
    return g_rccl.handle ? &g_rccl : nullptr;
}

#define HIP_TRY(ctx, expr)                                                                                         \
    do {                                                                                                           \
        hipError_t e_ = (expr);                                                                                    \
        if (e_ != hipSuccess) return hpt_internal_fail(ctx, HPT_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

#define NCCL_TRY(ctx, R, expr)                                                                                          \
    do {                                                                                                                \
        ncclResult_t r_ = (expr);                                                                                       \
        if (r_ != ncclSuccess) return hpt_internal_fail(ctx, HPT_ERR_DEVICE, "%s: %s", #expr, (R)->GetErrorString(r_));  \
    } while (0)

__global__ void scale_image_kernel(const float *__restrict__ src, float *__restrict__ dst, long n, float s) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i] * s;
}

}  // namespace

// per-context communicator state (hpt_ctx::comm)
struct hpt_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    float *d_slabs = nullptr;          // `world` slabs end to end: the all-gather runs in place, this rank packs into slab `rank`
    size_t slabs_floats = 0;
};

namespace {

int require_comm(hpt_ctx *ctx, hpt_comm *&c, Rccl *&R, const char *who) {
    if (!ctx) return HPT_ERR_INVALID;
    c = hpt_internal_comm(ctx);
    if (!c || !c->comm) return hpt_internal_fail(ctx, HPT_ERR_INVALID, "%s: no communicator (hpt_comm_init or hpt_group_create first)", who);
    R = rccl();
    if (!R) return hpt_internal_fail(ctx, HPT_ERR_DEVICE, "%s: %s", who, g_rccl.error.c_str());
    return HPT_OK;
}

// The three steps of hpt_gather_tiles; a group puts ncclGroupStart/End around the middle one of all its members.
int gather_pack(hpt_ctx *ctx, hpt_comm *c, const float *d_image) {
    const hpt_ctx_view v = hpt_internal_view(ctx);
    if (v.rank != c->rank || v.world != c->world)
        return hpt_internal_fail(ctx, HPT_ERR_INVALID, "hpt_gather_tiles: tile shard %d of %d but communicator rank %d of %d "
                                                       "(hpt_set_shard must match hpt_comm_init)", v.rank, v.world, c->rank, c->world);
    const size_t n = (size_t)hpt_slab_floats(ctx);
    if (n * (size_t)c->world != c->slabs_floats) {
        HIP_TRY(ctx, hipSetDevice(v.device));
        if (c->d_slabs) { HIP_TRY(ctx, hipStreamSynchronize(v.stream)); HIP_TRY(ctx, hipFree(c->d_slabs)); c->d_slabs = nullptr; c->slabs_floats = 0; }
        hipError_t e = hipMalloc((void **)&c->d_slabs, n * (size_t)c->world * sizeof(float));
        if (e != hipSuccess) { (void)hipGetLastError(); return hpt_internal_fail(ctx, e == hipErrorOutOfMemory ? HPT_ERR_NOMEM : HPT_ERR_DEVICE, "hipMalloc of the slab buffer: %s", hipGetErrorString(e)); }
        c->slabs_floats = n * (size_t)c->world;
    }
    return hpt_pack_tiles(ctx, d_image, c->d_slabs + (size_t)c->rank * n);
}

int gather_collective(hpt_ctx *ctx, hpt_comm *c, Rccl *R) {
    const hpt_ctx_view v = hpt_internal_view(ctx);
    const size_t n = c->slabs_floats / (size_t)c->world;
    HIP_TRY(ctx, hipSetDevice(v.device));
    NCCL_TRY(ctx, R, R->AllGather(c->d_slabs + (size_t)c->rank * n, c->d_slabs, n, ncclFloat, c->comm, v.stream));
    return HPT_OK;
}

int reduce_scale(hpt_ctx *ctx, hpt_comm *c, const float *d_image, float *d_frame, int n_local, int n_total) {
    const hpt_ctx_view v = hpt_internal_view(ctx);
    if (v.sample_rank != c->rank || v.sample_world != c->world)
        return hpt_internal_fail(ctx, HPT_ERR_INVALID, "hpt_reduce_samples: sample shard %d of %d but communicator rank %d of %d "
                                                       "(hpt_set_sample_shard must match hpt_comm_init)", v.sample_rank, v.sample_world, c->rank, c->world);
    if (!d_image || !d_frame || n_local < 0 || n_total < 1 || n_local > n_total)
        return hpt_internal_fail(ctx, HPT_ERR_INVALID, "hpt_reduce_samples: bad arguments (n_local %d, n_total %d)", n_local, n_total);
    const long n = 3L * v.width * v.height;
    const float s = (float)((double)n_local / (double)n_total);       // one rounding to fp32, then one fp32 multiply per component
    HIP_TRY(ctx, hipSetDevice(v.device));
    hipLaunchKernelGGL(scale_image_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, v.stream, d_image, d_frame, n, s);
    HIP_TRY(ctx, hipGetLastError());
    return HPT_OK;
}

int reduce_collective(hpt_ctx *ctx, hpt_comm *c, Rccl *R, float *d_frame) {
    const hpt_ctx_view v = hpt_internal_view(ctx);
    HIP_TRY(ctx, hipSetDevice(v.device));
    NCCL_TRY(ctx, R, R->AllReduce(d_frame, d_frame, (size_t)3 * v.width * v.height, ncclFloat, ncclSum, c->comm, v.stream));
    return HPT_OK;
}

int samples_below(int s, int r, int G) { return s <= r ? 0 : (s - r + G - 1) / G; }      // iterations < s that rank r of G renders

}  // namespace

extern "C" {

int hpt_comm_unique_id(void *id_out) {
    if (!id_out) return hpt_internal_fail(nullptr, HPT_ERR_INVALID, "hpt_comm_unique_id: id_out is NULL");
    Rccl *R = rccl();
    if (!R) return hpt_internal_fail(nullptr, HPT_ERR_DEVICE, "hpt_comm_unique_id: %s", g_rccl.error.c_str());
    static_assert(sizeof(ncclUniqueId) == HPT_COMM_ID_BYTES, "HPT_COMM_ID_BYTES");
    ncclUniqueId id;
    NCCL_TRY(nullptr, R, R->GetUniqueId(&id));
    memcpy(id_out, &id, sizeof id);
    return HPT_OK;
}

int hpt_comm_init(hpt_ctx *ctx, const void *id, int rank, int world) {
    if (!ctx) return HPT_ERR_INVALID;
    if (!id || world < 1 || rank < 0 || rank >= world) return hpt_internal_fail(ctx, HPT_ERR_INVALID, "hpt_comm_init: bad rank %d of %d", rank, world);
    if (hpt_internal_comm(ctx)) return hpt_internal_fail(ctx, HPT_ERR_INVALID, "hpt_comm_init: the context has a communicator already (hpt_comm_destroy first)");
    Rccl *R = rccl();
    if (!R) return hpt_internal_fail(ctx, HPT_ERR_DEVICE, "hpt_comm_init: %s", g_rccl.error.c_str());
    const hpt_ctx_view v = hpt_internal_view(ctx);
    HIP_TRY(ctx, hipSetDevice(v.device));
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    ncclComm_t comm = nullptr;
    NCCL_TRY(ctx, R, R->CommInitRank(&comm, world, uid, rank));
    hpt_comm *c = new hpt_comm;
    c->comm = comm; c->rank = rank; c->world = world;
    hpt_internal_comm(ctx) = c;
    return HPT_OK;
}

int hpt_comm_destroy(hpt_ctx *ctx) {
    if (!ctx) return HPT_OK;
    hpt_comm *&slot = hpt_internal_comm(ctx);
    hpt_comm *c = slot;
    if (!c) return HPT_OK;
    slot = nullptr;
    const hpt_ctx_view v = hpt_internal_view(ctx);
    (void)hipSetDevice(v.device);
    (void)hipStreamSynchronize(v.stream);          // a collective that still reads the slabs
    if (c->comm) { Rccl *R = rccl(); if (R) (void)R->CommDestroy(c->comm); }
    if (c->d_slabs) (void)hipFree(c->d_slabs);
    delete c;
    return HPT_OK;
}

int hpt_gather_tiles(hpt_ctx *ctx, float *d_image) {
    hpt_comm *c; Rccl *R;
    int rc = require_comm(ctx, c, R, "hpt_gather_tiles");
    if (rc) return rc;
    if (!d_image) return hpt_internal_fail(ctx, HPT_ERR_INVALID, "hpt_gather_tiles: d_image is NULL");
    if ((rc = gather_pack(ctx, c, d_image))) return rc;
    if ((rc = gather_collective(ctx, c, R))) return rc;
    return hpt_unpack_tiles(ctx, c->d_slabs, d_image);
}

int hpt_reduce_samples(hpt_ctx *ctx, const float *d_image, float *d_frame, int n_local, int n_total) {
    hpt_comm *c; Rccl *R;
    int rc = require_comm(ctx, c, R, "hpt_reduce_samples");
    if (rc) return rc;
    if ((rc = reduce_scale(ctx, c, d_image, d_frame, n_local, n_total))) return rc;
    return reduce_collective(ctx, c, R, d_frame);
}

}  // extern "C"

// ---- one process, several GPUs -------------------------------------------------------------------------------------
struct hpt_group {
    int width = 0, height = 0;
    std::vector<int> devices;
    std::vector<hpt_ctx *> ctx;
    std::vector<hipStream_t> streams;
    std::vector<float *> images;       // per member: running mean of ITS shard (tiles: the gathered frame after a render)
    std::vector<float *> frames;       // sample shards only: the reduced frame (the member's own mean must survive for the next slice)
    int mode = HPT_SHARD_TILES;
    int spp_done = 0;                  // sample shards: iterations [0, spp_done) are folded into the members' means
    std::string err;
};

namespace {

std::string g_group_create_error;

int gfail(hpt_group *g, int code, const std::string &msg) {
    if (g) g->err = msg; else g_group_create_error = msg;
    return code;
}

// a member's failure, with its text, becomes the group's
int member_failed(hpt_group *g, int member, int code, const char *what) {
    char buf[640];
    snprintf(buf, sizeof buf, "%s, member %d (device %d): %s", what, member, g->devices[member], hpt_last_error(g->ctx[member]));
    return gfail(g, code, buf);
}

// the same call on every member, one host thread each (the uploads lay out and copy the scene: seconds at 10 M triangles)
template <class F>
int for_members_parallel(hpt_group *g, const char *what, F call) {
    const int n = (int)g->ctx.size();
    std::vector<int> rc(n, HPT_OK);
    if (n == 1) rc[0] = call(g->ctx[0]);
    else {
        std::vector<std::thread> th;
        for (int i = 0; i < n; i++) th.emplace_back([&, i] { rc[i] = call(g->ctx[i]); });
        for (std::thread &t : th) t.join();
    }
    for (int i = 0; i < n; i++) if (rc[i]) return member_failed(g, i, rc[i], what);
    return HPT_OK;
}

}  // namespace

extern "C" {

const char *hpt_group_last_error(const hpt_group *g) { return g ? g->err.c_str() : g_group_create_error.c_str(); }

int hpt_group_destroy(hpt_group *g) {
    if (!g) return HPT_OK;
    for (size_t i = 0; i < g->ctx.size(); i++) {
        if (!g->ctx[i]) continue;
        (void)hpt_synchronize(g->ctx[i]);
        if (i < g->images.size() && g->images[i]) (void)hpt_image_free(g->ctx[i], g->images[i]);
        if (i < g->frames.size() && g->frames[i]) (void)hpt_image_free(g->ctx[i], g->frames[i]);
        (void)hpt_destroy(g->ctx[i]);                     // releases the member's communicator
        if (i < g->streams.size() && g->streams[i]) { (void)hipSetDevice(g->devices[i]); (void)hipStreamDestroy(g->streams[i]); }
    }
    delete g;
    return HPT_OK;
}

int hpt_group_create(int width, int height, const int *devices, int n_devices, hpt_group **out) {
    if (!out) return gfail(nullptr, HPT_ERR_INVALID, "hpt_group_create: out is NULL");
    *out = nullptr;
    if (!devices || n_devices < 1 || n_devices > 64) return gfail(nullptr, HPT_ERR_INVALID, "hpt_group_create: 1..64 devices");
    // (HPT_GROUP_SHARED_DEVICE=1: rehearsal of a group of several members on a box with fewer GPUs, with a stand-in for RCCL
    // named by HPT_RCCL_LIB -- tests/native/fake_rccl.hip; real RCCL refuses two ranks on one GPU by itself)
    const char *shared = getenv("HPT_GROUP_SHARED_DEVICE");
    for (int i = 0; i < n_devices && !(shared && atoi(shared) != 0); i++)
        for (int j = 0; j < i; j++)
            if (devices[i] == devices[j]) return gfail(nullptr, HPT_ERR_INVALID, "hpt_group_create: device " + std::to_string(devices[i]) + " is named twice");
    hpt_group *g = new hpt_group;
    g->width = width; g->height = height;
    g->devices.assign(devices, devices + n_devices);
    g->ctx.assign(n_devices, nullptr); g->streams.assign(n_devices, nullptr);
    g->images.assign(n_devices, nullptr); g->frames.assign(n_devices, nullptr);
    auto bail = [&](int code, const std::string &msg) { gfail(nullptr, code, msg); hpt_group_destroy(g); return code; };
    for (int i = 0; i < n_devices; i++) {
        int rc = hpt_create(width, height, devices[i], &g->ctx[i]);
        if (rc) return bail(rc, std::string("hpt_group_create: device ") + std::to_string(devices[i]) + ": " + hpt_last_error(nullptr));
        hipError_t e = hipSetDevice(devices[i]);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&g->streams[i], hipStreamNonBlocking);
        if (e != hipSuccess) return bail(HPT_ERR_DEVICE, std::string("hpt_group_create: hipStreamCreate: ") + hipGetErrorString(e));
        if ((rc = hpt_set_stream(g->ctx[i], g->streams[i])) || (rc = hpt_set_shard(g->ctx[i], i, n_devices)) ||
            (rc = hpt_image_alloc(g->ctx[i], &g->images[i])))
            return bail(rc, std::string("hpt_group_create: ") + hpt_last_error(g->ctx[i]));
    }
    Rccl *R = rccl();
    if (!R) return bail(HPT_ERR_DEVICE, "hpt_group_create: " + g_rccl.error);
    std::vector<ncclComm_t> comms(n_devices, nullptr);
    ncclResult_t r = R->CommInitAll(comms.data(), n_devices, devices);
    if (r != ncclSuccess) return bail(HPT_ERR_DEVICE, std::string("hpt_group_create: ncclCommInitAll: ") + R->GetErrorString(r));
    for (int i = 0; i < n_devices; i++) {
        hpt_comm *c = new hpt_comm;
        c->comm = comms[i]; c->rank = i; c->world = n_devices;
        hpt_internal_comm(g->ctx[i]) = c;
    }
    *out = g;
    return HPT_OK;
}

int hpt_group_size(const hpt_group *g) { return g ? (int)g->ctx.size() : HPT_ERR_INVALID; }

hpt_ctx *hpt_group_ctx(hpt_group *g, int member) {
    return (g && member >= 0 && member < (int)g->ctx.size()) ? g->ctx[member] : nullptr;
}

float *hpt_group_frame(hpt_group *g, int member) {
    if (!g || member < 0 || member >= (int)g->ctx.size()) return nullptr;
    return g->mode == HPT_SHARD_SAMPLES ? g->frames[member] : g->images[member];
}

int hpt_group_set_sharding(hpt_group *g, int mode) {
    if (!g) return HPT_ERR_INVALID;
    if (mode != HPT_SHARD_TILES && mode != HPT_SHARD_SAMPLES) return gfail(g, HPT_ERR_INVALID, "hpt_group_set_sharding: mode " + std::to_string(mode));
    const int n = (int)g->ctx.size();
    for (int i = 0; i < n; i++) {
        int rc;
        if (mode == HPT_SHARD_TILES) { if ((rc = hpt_set_shard(g->ctx[i], i, n)) || (rc = hpt_set_sample_shard(g->ctx[i], 0, 1))) return member_failed(g, i, rc, "hpt_group_set_sharding"); }
        else {
            if ((rc = hpt_set_shard(g->ctx[i], 0, 1)) || (rc = hpt_set_sample_shard(g->ctx[i], i, n))) return member_failed(g, i, rc, "hpt_group_set_sharding");
            if (!g->frames[i] && (rc = hpt_image_alloc(g->ctx[i], &g->frames[i]))) return member_failed(g, i, rc, "hpt_group_set_sharding");
        }
    }
    g->mode = mode; g->spp_done = 0;
    return HPT_OK;
}

int hpt_group_upload_triangles(hpt_group *g, const hpt_triangle *triangles, int n_triangles) {
    if (!g) return HPT_ERR_INVALID;
    return for_members_parallel(g, "hpt_upload_triangles", [&](hpt_ctx *c) { return hpt_upload_triangles(c, triangles, n_triangles); });
}

int hpt_group_upload_bvh(hpt_group *g, const hpt_bvh_bbox *bbox, const hpt_bvh_info *info, int n_nodes) {
    if (!g) return HPT_ERR_INVALID;
    return for_members_parallel(g, "hpt_upload_bvh", [&](hpt_ctx *c) { return hpt_upload_bvh(c, bbox, info, n_nodes); });
}

int hpt_group_set_scene(hpt_group *g, const hpt_primitive *primitives, int n_primitives, const hpt_material *materials, int n_materials) {
    if (!g) return HPT_ERR_INVALID;
    return for_members_parallel(g, "hpt_set_scene", [&](hpt_ctx *c) { return hpt_set_scene(c, primitives, n_primitives, materials, n_materials); });
}

int hpt_group_set_camera(hpt_group *g, const hpt_camera *camera) {
    if (!g) return HPT_ERR_INVALID;
    for (size_t i = 0; i < g->ctx.size(); i++) { int rc = hpt_set_camera(g->ctx[i], camera); if (rc) return member_failed(g, (int)i, rc, "hpt_set_camera"); }
    return HPT_OK;
}

int hpt_group_set_config(hpt_group *g, const hpt_config *cfg) {
    if (!g) return HPT_ERR_INVALID;
    for (size_t i = 0; i < g->ctx.size(); i++) { int rc = hpt_set_config(g->ctx[i], cfg); if (rc) return member_failed(g, (int)i, rc, "hpt_set_config"); }
    return HPT_OK;
}

int hpt_group_render(hpt_group *g, int spp_begin, int spp_end) {
    if (!g) return HPT_ERR_INVALID;
    if (spp_begin < 0 || spp_end < spp_begin) return gfail(g, HPT_ERR_INVALID, "hpt_group_render: bad range");
    if (spp_end == spp_begin) return HPT_OK;
    const int n = (int)g->ctx.size();
    Rccl *R = rccl();
    if (!R) return gfail(g, HPT_ERR_DEVICE, "hpt_group_render: " + g_rccl.error);
    int rc;
    if (g->mode == HPT_SHARD_SAMPLES) {
        if (spp_begin != 0 && spp_begin != g->spp_done)
            return gfail(g, HPT_ERR_INVALID, "hpt_group_render: sample shards continue at iteration " + std::to_string(g->spp_done) + " or restart at 0");
        for (int i = 0; i < n; i++) {
            if (spp_begin == 0) {                 // a member without a sample yet contributes 0 * image: keep that finite
                hipError_t e = hipSetDevice(g->devices[i]);
                if (e == hipSuccess) e = hipMemsetAsync(g->images[i], 0, (size_t)3 * g->width * g->height * sizeof(float), g->streams[i]);
                if (e != hipSuccess) return gfail(g, HPT_ERR_DEVICE, std::string("hpt_group_render: hipMemsetAsync: ") + hipGetErrorString(e));
            }
            if ((rc = hpt_render(g->ctx[i], spp_begin, spp_end, g->images[i]))) return member_failed(g, i, rc, "hpt_render");
        }
        for (int i = 0; i < n; i++)
            if ((rc = reduce_scale(g->ctx[i], hpt_internal_comm(g->ctx[i]), g->images[i], g->frames[i], samples_below(spp_end, i, n), spp_end)))
                return member_failed(g, i, rc, "hpt_reduce_samples");
        ncclResult_t r = R->GroupStart();
        if (r != ncclSuccess) return gfail(g, HPT_ERR_DEVICE, std::string("ncclGroupStart: ") + R->GetErrorString(r));
        for (int i = 0; i < n; i++)
            if ((rc = reduce_collective(g->ctx[i], hpt_internal_comm(g->ctx[i]), R, g->frames[i]))) { (void)R->GroupEnd(); return member_failed(g, i, rc, "ncclAllReduce"); }
        if ((r = R->GroupEnd()) != ncclSuccess) return gfail(g, HPT_ERR_DEVICE, std::string("ncclGroupEnd: ") + R->GetErrorString(r));
        g->spp_done = spp_end;
        return HPT_OK;
    }
    for (int i = 0; i < n; i++)
        if ((rc = hpt_render(g->ctx[i], spp_begin, spp_end, g->images[i]))) return member_failed(g, i, rc, "hpt_render");
    for (int i = 0; i < n; i++)
        if ((rc = gather_pack(g->ctx[i], hpt_internal_comm(g->ctx[i]), g->images[i]))) return member_failed(g, i, rc, "hpt_pack_tiles");
    ncclResult_t r = R->GroupStart();
    if (r != ncclSuccess) return gfail(g, HPT_ERR_DEVICE, std::string("ncclGroupStart: ") + R->GetErrorString(r));
    for (int i = 0; i < n; i++)
        if ((rc = gather_collective(g->ctx[i], hpt_internal_comm(g->ctx[i]), R))) { (void)R->GroupEnd(); return member_failed(g, i, rc, "ncclAllGather"); }
    if ((r = R->GroupEnd()) != ncclSuccess) return gfail(g, HPT_ERR_DEVICE, std::string("ncclGroupEnd: ") + R->GetErrorString(r));
    for (int i = 0; i < n; i++)
        if ((rc = hpt_unpack_tiles(g->ctx[i], hpt_internal_comm(g->ctx[i])->d_slabs, g->images[i]))) return member_failed(g, i, rc, "hpt_unpack_tiles");
    return HPT_OK;
}

int hpt_group_synchronize(hpt_group *g) {
    if (!g) return HPT_ERR_INVALID;
    for (size_t i = 0; i < g->ctx.size(); i++) { int rc = hpt_synchronize(g->ctx[i]); if (rc) return member_failed(g, (int)i, rc, "hpt_synchronize"); }
    return HPT_OK;
}

int hpt_group_read_image(hpt_group *g, float *image_rgb) {
    if (!g || !image_rgb) return HPT_ERR_INVALID;
    int rc = hpt_group_synchronize(g);             // every member's part of the collective is done, not only member 0's
    if (rc) return rc;
    if ((rc = hpt_read_image(g->ctx[0], hpt_group_frame(g, 0), image_rgb))) return member_failed(g, 0, rc, "hpt_read_image");
    return HPT_OK;
}

int hpt_group_get_stats(hpt_group *g, hpt_stats *stats) {
    if (!g || !stats) return HPT_ERR_INVALID;
    memset(stats, 0, sizeof *stats);
    for (size_t i = 0; i < g->ctx.size(); i++) {
        hpt_stats s;
        int rc = hpt_get_stats(g->ctx[i], &s);
        if (rc) return member_failed(g, (int)i, rc, "hpt_get_stats");
        stats->rays_primary += s.rays_primary; stats->rays_bounce += s.rays_bounce; stats->paths += s.paths; stats->lane_loads += s.lane_loads;
        if (s.last_kernel_ms > stats->last_kernel_ms) stats->last_kernel_ms = s.last_kernel_ms;
        if (s.fold_ms > stats->fold_ms) stats->fold_ms = s.fold_ms;
        if (s.launches > stats->launches) stats->launches = s.launches;
        stats->grid_waves += s.grid_waves;
        stats->wide_nodes = s.wide_nodes; stats->tree_depth = s.tree_depth;
    }
    return HPT_OK;
}

}  // extern "C"
