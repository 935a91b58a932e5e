// hpt_backend.hip — the C-ABI of include/hpt/hpt.h over the gfx950 kernels.
//
// Replaces the reference's device backend class OpenCL_Manager
// (reference pathtracer/opencl_manager.{hpp,cpp}).  Differences in kind, not
// in results: scene data is uploaded once and stays resident (the reference
// re-creates every buffer and copies all per-pixel state both ways on every
// call, opencl_manager.cpp:368-431, 515-523), errors come back as status codes
// instead of exit(1), and the image/accumulation buffer lives in HBM.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/hpt/hpt.h"
#include "hpt_internal.h"
#include "render_kernels.h"
#include "host/bvh.hpp"            // parallel_chunks, build_threads

namespace {

std::string g_create_error;

constexpr int kMaxTreeDepth = 64;   // the reference's own bound: int todo[64], ray_kernel.cl:389
constexpr int kBlock = 256;
constexpr size_t kLdsGranule = 1280;  // LDS allocation granule per block (measured: choose_render())
constexpr int kLdsTopMax = 1024;    // nodes of the top of the tree that are numbered first, breadth first (hpt_upload_bvh); render_kernel keeps a prefix in LDS
#define HPT_N_COUNTERS 16
#define HPT_COUNTER_LOADS 15        // render_kernels.h: 16-B lane-loads

// Scratch of hpt_render_noalloc (include/hpt/hpt.h): the traversal-stack overflow and the sample buffer of a render
// that may not allocate.  One array per GPU (every device that loads this code object holds its own copy at its own
// address until the process ends); the host code never uses the NAME of the array as an address, only the pointer
// hipGetSymbolAddress() returned for the device at hand (scratch_of()).
constexpr size_t kScratchBytes = (size_t)HPT_NOALLOC_SCRATCH_MB << 20;
constexpr size_t kScratchSpillMax = kScratchBytes / 2;        // the stack overflow may take at most half
constexpr int kMaxDevices = 64;

// occupancy of a persistent kernel at a dynamic-LDS size: asked of the runtime once, not on every launch
struct KernelShape {
    const void *kernel;
    size_t lds;
    int blocks_per_cu;
};

}  // namespace

static __device__ __align__(256) unsigned char hpt_noalloc_scratch[kScratchBytes];

namespace {
std::mutex g_scratch_mutex;
void *g_scratch_ptr[kMaxDevices] = {};          // per device: the address of ITS copy of hpt_noalloc_scratch

// Address of the current device's scratch array, looked up once per device (with that device current).
hipError_t scratch_of(int device, void **out) {
    *out = nullptr;
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void *ptr = nullptr;
        hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(hpt_noalloc_scratch));
        if (e != hipSuccess) return e;
        g_scratch_ptr[device] = ptr;
    }
    *out = g_scratch_ptr[device];
    return hipSuccess;
}
}  // namespace

struct hpt_ctx {
    int device = 0;
    int width = 0, height = 0;
    hipStream_t stream = nullptr;
    int rank = 0, world = 1;
    int sample_rank = 0, sample_world = 1;   // hpt_set_sample_shard: this context renders iterations s % sample_world == sample_rank
    hpt_config cfg{15, 3, 2.0f, 0x5EEDu};
    hpt_camera cam{};
    bool have_camera = false;

    WideNode *d_nodes = nullptr;
    int n_wide = 0;
    int tree_depth = 0;
    int max_leaf_end = 0;            // largest offset+count over the leaves, checked against n_tris at render
    int root_ref = HPT_REF_DONE;
    float root_lo[3] = {0, 0, 0}, root_hi[3] = {0, 0, 0};

    float4 *d_tris = nullptr;        // wire-format triangles (shading reads the hit triangle's edges)
    int n_tris = 0;
    // The traversal reads triangles as 80-B records of two consecutive triangles of a leaf (render_kernels.h),
    // which needs both the triangle array and the tree: built lazily from these host copies, in either upload order.
    std::vector<hpt_triangle> h_tris;
    std::vector<WideNode> h_wide;    // leaf references still (first triangle << 6 | count-1)
    int root_ref_tris = HPT_REF_DONE;
    float4 *d_pairs = nullptr;
    bool pairs_dirty = true;
    // Upload generations: a scene REPLACED in a resident context arrives as two uploads, and building the records after the
    // first would pair the new half with the old one, only to be thrown away by the second (1.26 GB of records at C4).  An
    // upload builds eagerly only when the other half is newer than the last build; otherwise the build waits for the other
    // upload or for the next render.
    unsigned tris_gen = 0, bvh_gen = 0, built_tris_gen = 0, built_bvh_gen = 0;
    bool singles = false;            // record form of the current tree: one triangle per 48-B record (ensure_pairs)
    size_t tree_bytes = 0;           // nodes + triangle records of the current tree as the traversal reads them (ensure_pairs)
    int leaf_records = 0;            // HPT_LEAF_RECORDS: 0 = per tree, 1 = singles, 2 = pairs
    hpt_primitive *d_prims = nullptr;
    PrimCompact *d_prim_compact = nullptr;   // centre, radius, type, plane normal of each primitive (render_kernels.h)
    PrimPair *d_prim_pairs = nullptr;        // the same, two primitives side by side
    int n_prims = 0;
    hpt_material *d_mats = nullptr;
    int n_mats = 0;

    unsigned int *d_work = nullptr;          // persistent-kernel work counter
    unsigned long long *d_counters = nullptr;
    float *d_sample = nullptr;               // width*height*3, kernel_execute / render_sample staging
    float *d_accum = nullptr;                // width*height*3, render_sample staging
    int32_t *d_hit_id = nullptr;
    float *d_hit_t = nullptr;

    // Per-stage timing (hpt_get_stats): HIP events on the launch stream.  A render records ev_render[3b], [3b+1], [3b+2]
    // around the render_kernel and the fold_kernel of its b-th launch (the pool grows on demand); the iterative calls
    // have fixed events.
    std::vector<hipEvent_t> ev_render;
    int timed_launches = 0;          // launches of the last render whose events were recorded
    int launches = 0;                // launches of render_kernel by the last render
    hipEvent_t ev_iter[3] = {nullptr, nullptr, nullptr};     // before trace, between trace and shade, after shade
    hipEvent_t ev_recon[2] = {nullptr, nullptr};
    bool timed_iter = false, timed_recon = false;
    std::vector<KernelShape> shapes;
    unsigned char *scratch = nullptr;        // this device's hpt_noalloc_scratch (hipGetSymbolAddress at hpt_create)
    int n_cu = 256;
    int grid_waves = 0;
    int shade_threshold = 24;        // scheduling knob of render_kernel (results do not depend on it): 24, up to 40 with the length of the
                                     // primitive list, which every ray that starts in a SHADE step walks (fill_params); HPT_SHADE_THRESHOLD fixes it
    bool shade_threshold_fixed = false;
    int tris_per_trip = 8;
    int node_repeat_min = 24;
    int pair_fetch = -1;             // HPT_PAIR_FETCH=0/1 forces the helper-lane node fetch off / on (default: single-record trees of 1 GB and more)
    int tri_weight = 8;              // HPT_TRI_WEIGHT: a lane in the TRI state counts tri_weight / 4 NODE lanes when the wave picks its step
    int fetch_chunk = 256;           // work items a WAVE reserves per atomic; its lanes take them in order (HPT_FETCH_CHUNK overrides)
    int tail_chunks = 4;             // the last tail_chunks x waves x fetch_chunk work items of a launch are reserved 64 at a time (HPT_TAIL_CHUNKS; 2/4/8/16: 701.8/700.7/700.5/700.6 ms at C3)
    int fetch_threshold = 32;        // iter_trace_kernel's FETCH step threshold (HPT_FETCH_THRESHOLD overrides)
    int stack_cap = 0;               // LDS stack entries per lane of the persistent kernels; 0 = per tree (stack_cap_of()), HPT_STACK_CAP fixes it
    uint2 *d_spill = nullptr;        // HBM overflow of the traversal stack
    size_t spill_bytes = 0;
    float *d_samples = nullptr;      // [iteration in batch][pixel slot][3] radiance samples of one launch
    size_t samples_bytes = 0;
    size_t sample_budget_bytes = (size_t)HPT_SAMPLE_BUDGET_DEFAULT_MB << 20;   // HPT_SAMPLE_BUDGET_MB overrides; bounds the paths per launch
    int blocks_per_cu_cap = 0;       // 0 = occupancy limit (HPT_BLOCKS_PER_CU overrides)
    size_t extra_lds = 0;            // LDS a launch wants behind its stacks (render_kernel's copy of the top of the tree)
    int lds_top = -1;                // nodes of the top of the tree that render_kernel keeps in LDS: -1 = as many as fit beside the stacks at the
                                     // kernel's occupancy (C3: 104 beside eight stack entries per lane at 7 blocks per CU); HPT_LDS_TOP_NODES fixes it (0 = none)
    size_t lds_per_cu = 160 << 10;
    bool count_loads = false;        // HPT_COUNT_LANE_LOADS=1: renders run the counting instantiation of render_kernel (hpt_stats.lane_loads)
    void *d_iter = nullptr;          // per-pixel state of the iterative pipeline (one allocation), made by hpt_first_bounce
    IterState iter{};
    int iter_ready = 0;              // 1 after hpt_first_bounce
    int iter_pending = 0;            // 1 between hpt_bounce and hpt_reconstruct_paths
    hpt_comm *comm = nullptr;        // communicator of the multi-GPU exchange (hpt_comm.hip), released by hpt_destroy
    std::string err;
};

// ---- hpt_internal.h: what the library's other translation units see of a context ----
int hpt_internal_fail(hpt_ctx *ctx, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf; else g_create_error = buf;
    return code;
}
hpt_ctx_view hpt_internal_view(const hpt_ctx *c) {
    return hpt_ctx_view{c->device, c->width, c->height, c->stream, c->rank, c->world, c->sample_rank, c->sample_world, c->cfg};
}
hpt_comm *&hpt_internal_comm(hpt_ctx *ctx) { return ctx->comm; }

namespace {

int fail(hpt_ctx *ctx, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf; else g_create_error = buf;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                            \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) return fail(ctx, HPT_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// device allocation: out of memory is its own status (HPT_ERR_NOMEM), anything else a device error
#define HIP_ALLOC(ctx, pptr, bytes)                                                                     \
    do {                                                                                              \
        hipError_t e_ = hipMalloc((void **)(pptr), (bytes));                                          \
        if (e_ == hipErrorOutOfMemory) {                                                              \
            (void)hipGetLastError();                                                                  \
            return fail(ctx, HPT_ERR_NOMEM, "out of device memory allocating %zu bytes", (size_t)(bytes)); \
        }                                                                                             \
        if (e_ != hipSuccess) return fail(ctx, HPT_ERR_DEVICE, "hipMalloc: %s", hipGetErrorString(e_)); \
    } while (0)

int bind(hpt_ctx *ctx) {
    if (!ctx) return HPT_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return HPT_OK;
}

template <class T>
int replace_buffer(hpt_ctx *ctx, T *&dptr, const void *host, size_t bytes) {
    if (dptr) { HIP_TRY(ctx, hipFree(dptr)); dptr = nullptr; }
    if (bytes == 0) return HPT_OK;
    HIP_ALLOC(ctx, &dptr, bytes);
    HIP_TRY(ctx, hipMemcpyAsync(dptr, host, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // copy-at-call-time semantics of CL_MEM_COPY_HOST_PTR
    return HPT_OK;
}

int tiles_x(const hpt_ctx *c) { return (c->width + 7) / 8; }
int tiles_y(const hpt_ctx *c) { return (c->height + 7) / 8; }
int n_tiles(const hpt_ctx *c) { return tiles_x(c) * tiles_y(c); }
// tiles of the largest shard: every rank's slab has this many (all-gather wants equal sizes)
int slab_tiles(const hpt_ctx *c) { return (n_tiles(c) + c->world - 1) / c->world; }

// Builds the traversal's triangle records (render_kernels.h) and uploads the nodes with their leaf references
// re-pointed at them.  Leaves are numbered in node order, so a near-first descent walks the records forward.
// Record form, per tree: two consecutive triangles of a leaf per 80 B, or -- when the leaves hold fewer than
// 1.25 triangles on average (HPT_LEAF_RECORDS=pairs|singles overrides) -- one triangle per 48 B.
typedef void (*render_fn)(const RenderParams);
int choose_render(hpt_ctx *ctx, render_fn &fn, uint32_t &n_lds_nodes);

int ensure_pairs(hpt_ctx *ctx) {
    if (!ctx->pairs_dirty) return HPT_OK;
    if (ctx->root_ref_tris == HPT_REF_DONE) { ctx->pairs_dirty = false; return HPT_OK; }     // no tree
    if (ctx->max_leaf_end > ctx->n_tris) return HPT_OK;       // fill_params reports it
    std::vector<WideNode> wide(ctx->h_wide);
    size_t n_leaves = ctx->root_ref_tris < 0 ? 1 : 0, leaf_tris = ctx->root_ref_tris < 0 ? (size_t)((~ctx->root_ref_tris & 63) + 1) : 0;
    for (const WideNode &w : wide)
        for (int32_t ref : {w.ref0, w.ref1})
            if (ref < 0) { n_leaves++; leaf_tris += (size_t)((~ref & 63) + 1); }
    bool singles = leaf_tris * 4 < n_leaves * 5;
    if (ctx->leaf_records > 0) singles = ctx->leaf_records == 1;
    const int rec_floats = singles ? 4 * HPT_SINGLE_FLOAT4S : 4 * HPT_PAIR_FLOAT4S;
    auto records_of = [&](int32_t ref) -> size_t {             // records a leaf reference needs
        if (ref >= 0) return 0;
        const size_t n = (size_t)((~ref & 63) + 1);
        return singles ? n : (n + 1) / 2;
    };
    // first record of each node's leaves (root leaf first, then node order): sequential prefix sum, parallel fill
    int32_t root = ctx->root_ref_tris;
    std::vector<size_t> first_rec(wide.size() + 1);
    size_t n_rec = records_of(root);
    for (size_t i = 0; i < wide.size(); i++) { first_rec[i] = n_rec; n_rec += records_of(wide[i].ref0) + records_of(wide[i].ref1); }
    first_rec[wide.size()] = n_rec;
    if (n_rec >= (size_t)(1 << 25)) return fail(ctx, HPT_ERR_BVH, "too many triangle records");
    std::vector<float> rec((n_rec + 1) * (size_t)rec_floats, 0.0f);   // one spare record: a pair load of the last single stays inside
    auto repoint = [&](int32_t &ref, size_t at) {
        if (ref >= 0) return;
        const int packed = ~ref, first = packed >> 6, n = (packed & 63) + 1;
        const size_t first_at = at;
        for (int i = 0; i < n; i += singles ? 1 : 2, at++) {
            float *r = rec.data() + at * (size_t)rec_floats;
            for (int h = 0; h < (singles ? 1 : 2) && i + h < n; h++) {
                const hpt_triangle &t = ctx->h_tris[first + i + h];
                float *f = r + (h ? 11 : 0);
                f[0] = t.v1.x; f[1] = t.v1.y; f[2] = t.v1.z; f[3] = t.e1.x; f[4] = t.e1.y; f[5] = t.e1.z;
                f[6] = t.e2.x; f[7] = t.e2.y; f[8] = t.e2.z;
                const int32_t id = first + i + h;
                memcpy(r + 9 + h, &id, 4);
            }
        }
        ref = HPT_PAIR_REF((int)first_at, n);
    };
    repoint(root, 0);
    hpt::parallel_chunks(hpt::build_threads(), (long)wide.size(), [&](int, long b, long e) {
        for (long i = b; i < e; i++) {
            repoint(wide[i].ref0, first_rec[i]);
            repoint(wide[i].ref1, first_rec[i] + records_of(ctx->h_wide[i].ref0));
        }
    });
    int rc = replace_buffer(ctx, ctx->d_nodes, wide.data(), wide.size() * sizeof(WideNode));
    if (rc) return rc;
    if ((rc = replace_buffer(ctx, ctx->d_pairs, rec.data(), rec.size() * sizeof(float)))) return rc;
    ctx->root_ref = root;
    ctx->singles = singles;
    ctx->tree_bytes = wide.size() * sizeof(WideNode) + rec.size() * sizeof(float);
    ctx->pairs_dirty = false;
    ctx->built_tris_gen = ctx->tris_gen; ctx->built_bvh_gen = ctx->bvh_gen;
    render_fn fn; uint32_t n_top;
    return choose_render(ctx, fn, n_top);      // attribute and occupancy queries now, not inside the first render call
}

constexpr size_t kStreamingTreeBytes = (size_t)256 << 20;     // the Infinity Cache
constexpr size_t kPairFetchTreeBytes = (size_t)1 << 30;       // helper-lane node fetch: measured to pay at 1.26 GB, to cost 1-3 % at 378 MB

int fill_params(hpt_ctx *ctx, RenderParams &p, float *d_image, int spp_begin, int spp_end, int raw) {
    if (!ctx->have_camera) return fail(ctx, HPT_ERR_INVALID, "hpt_set_camera has not been called");
    if (ctx->root_ref != HPT_REF_DONE && ctx->max_leaf_end > ctx->n_tris)
        return fail(ctx, HPT_ERR_BVH, "BVH leaves reference triangle %d but only %d triangles are uploaded",
                    ctx->max_leaf_end - 1, ctx->n_tris);
    if ((long)ctx->width * ctx->height >= (1L << 24))
        return fail(ctx, HPT_ERR_INVALID, "width*height must stay below 2^24 (pixel index is converted to float, "
                                          "initial_ray_kernel.cl:863)");
    int rc_pairs = ensure_pairs(ctx);
    if (rc_pairs) return rc_pairs;
    memset(&p, 0, sizeof p);
    p.pairs = ctx->d_pairs;
    p.singles = ctx->singles ? 1 : 0;
    p.nodes = ctx->d_nodes; p.tris = ctx->d_tris; p.prims = ctx->d_prims; p.prim_compact = ctx->d_prim_compact; p.prim_pairs = ctx->d_prim_pairs; p.mats = ctx->d_mats;
    p.image = d_image; p.hit_id = ctx->d_hit_id; p.hit_t = ctx->d_hit_t;
    p.work_counter = ctx->d_work; p.counters = ctx->d_counters;
    p.cam = ctx->cam;
    for (int k = 0; k < 3; k++) { p.root_lo[k] = ctx->root_lo[k]; p.root_hi[k] = ctx->root_hi[k]; }
    p.root_ref = ctx->root_ref;
    p.width = ctx->width; p.height = ctx->height; p.n_prims = ctx->n_prims;
    p.spp_begin = spp_begin; p.spp_end = spp_end;
    p.max_depth = ctx->cfg.max_depth; p.rr_start = ctx->cfg.rr_start;
    p.diffuse_scale = ctx->cfg.diffuse_scale; p.seed = ctx->cfg.seed;
    p.rank = ctx->rank; p.world = ctx->world;
    p.tiles_x = tiles_x(ctx); p.n_tiles = n_tiles(ctx);
    int local_tiles = (p.n_tiles - ctx->rank + ctx->world - 1) / ctx->world;
    if (local_tiles < 0) local_tiles = 0;
    p.n_slots = (uint32_t)local_tiles * 64u;
    p.n_work = p.n_slots;
    p.batch = 1; p.spp_base = spp_begin; p.spp_first = spp_begin; p.spp_stride = 1;
    p.raw_sample = raw;
    // the longer the primitive list, the dearer a SHADE step (129 primitives: ~6 000 instructions), the more lanes
    // it should wait for: C5 24 / 32 / 40 / 48 -> 2 115 / 2 235 / 2 259 / 2 204 Mrays/s; C3 (1 primitive) is best at 24
    // ... and a tree that streams from HBM waits for misses, not for lanes: its finished rays can wait for more company
    // (C4 20 / 24 / 28 / 32 / 36 / 40 / 48: 163.6 / 162.6 / 161.7 / 161.4 / 161.3 / 161.8 / 166.3 ms)
    p.shade_threshold = ctx->shade_threshold_fixed ? ctx->shade_threshold
                                                   : ctx->shade_threshold + (ctx->n_prims / 8 < 16 ? ctx->n_prims / 8 : 16) +
                                                     (ctx->tree_bytes > kStreamingTreeBytes ? 8 : 0);
    p.tris_per_trip = ctx->tris_per_trip;
    p.node_repeat_min = ctx->node_repeat_min;
    p.tri_weight = ctx->tri_weight;
    p.fetch_chunk = ctx->fetch_chunk;
    p.fetch_threshold = ctx->fetch_threshold;
    return HPT_OK;
}

// LDS stack entries per lane.  Cache-resident trees: 8, which leaves room for 104 nodes of the tree's top in LDS at 7 blocks
// per CU (8 + 104 beat 10 + 40 and 9 + 72 at C3; without the node copy 8, 9 and 10 tie: profiles/r03/lds_top/).  Trees that
// stream from HBM get no node copy and all the LDS of their six blocks per CU as stack: 13 entries (13 x 2 KB = 26 624 B, 21 granules
// of 1 280 B, six times in 160 KB).  At C4 every entry that overflows to HBM is one more miss in a kernel that waits for misses:
// 8 / 10 / 11 / 12 / 13 entries 169.6 / 164.5 / 163.3 / 162.9 / 162.7 ms; the node copy on top of 10: -2.4 %.
bool streaming_tree(const hpt_ctx *ctx) { return ctx->tree_bytes > kStreamingTreeBytes; }
int stack_cap_of(const hpt_ctx *ctx) { return ctx->stack_cap > 0 ? ctx->stack_cap : streaming_tree(ctx) ? 13 : 8; }

size_t stack_lds_bytes(const hpt_ctx *ctx, int block) {
    int depth = ctx->tree_depth > 0 ? ctx->tree_depth : 1;
    return (size_t)depth * block * sizeof(uint2);
}

// Dynamic LDS any kernel here may ask for: a full-depth stack (kMaxTreeDepth entries of 8 B for 256 lanes).  Set once
// per kernel (the attribute is per function, so every context sets the same value), with the occupancy at the
// launch's actual LDS size cached beside it: neither is asked of the runtime again on later launches.
constexpr size_t kMaxDynamicLds = (size_t)kMaxTreeDepth * kBlock * sizeof(uint2);

int kernel_shape(hpt_ctx *ctx, const void *kernel, size_t lds, int &blocks_per_cu) {
    bool attribute_set = false;
    for (const KernelShape &k : ctx->shapes) {
        if (k.kernel != kernel) continue;
        attribute_set = true;
        if (k.lds == lds) { blocks_per_cu = k.blocks_per_cu; return HPT_OK; }
    }
    if (!attribute_set)
        HIP_TRY(ctx, hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxDynamicLds));
    int per_cu = 0;
    HIP_TRY(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kBlock, lds));
    ctx->shapes.push_back({kernel, lds, per_cu});
    blocks_per_cu = per_cu;
    return HPT_OK;
}

// The instantiation of render_kernel that renders the resident scene, and the number of nodes of the tree's top that it
// keeps in LDS.  Asked at every render and once when a scene becomes resident (ensure_pairs(), hpt_set_scene), so that the
// attribute and occupancy queries of kernel_shape() are made outside the render calls (hpt_render_noalloc: "calls no
// allocating routine" -- the queries cache their answers in a host vector).
int choose_render(hpt_ctx *ctx, render_fn &fn, uint32_t &n_lds_nodes) {
    // one instantiation per (record form, counting build, long primitive list)
    static const render_fn kRender[2][2][2] = {
        {{render_kernel<false, false, false>, render_kernel<false, false, true>}, {render_kernel<false, true, false>, render_kernel<false, true, true>}},
        {{render_kernel<true, false, false>, render_kernel<true, false, true>}, {render_kernel<true, true, false>, render_kernel<true, true, true>}}};
    // single-record trees of a gigabyte and more (short primitive lists): the instantiation whose NODE step fetches with helper lanes
    static const render_fn kRenderPair[2] = {render_kernel<true, false, false, true>, render_kernel<true, true, false, true>};
    const bool long_list = ctx->n_prims >= HPT_TWO_PHASE_MIN_PRIMS;
    const bool pair_fetch = ctx->singles && !long_list &&
                            (ctx->pair_fetch >= 0 ? ctx->pair_fetch != 0 : ctx->tree_bytes >= kPairFetchTreeBytes);
    fn = pair_fetch ? kRenderPair[ctx->count_loads ? 1 : 0]
                    : kRender[ctx->singles ? 1 : 0][ctx->count_loads ? 1 : 0][long_list ? 1 : 0];
    // The top of the tree in LDS, behind the stacks: as many nodes as fit without costing the kernel a block per CU (LDS is
    // handed out in granules of 1 280 B = 1/128 of the CU's 160 KB, which fits every observation of profiles/r03/lds_top/:
    // 7 x 23 040 B and 8 x 20 480 B run, 7 x 23 296 B and 6 x 27 136 B lose a block; C3 at 7 blocks per CU and 8 stack entries
    // per lane has room for 104 nodes; 105 and the seventh block is gone, -6 %, though the occupancy query still says 7).  C1 +3.6 %, C2 +2.0 %, C3 +4.9 %, C5 +-0;
    // trees that stream from HBM get none (C4 -2.4 % with it: stack_cap_of()).
    const int depth = ctx->tree_depth > 0 ? ctx->tree_depth : 1;
    const size_t stack_lds = (size_t)(stack_cap_of(ctx) < depth ? stack_cap_of(ctx) : depth) * kBlock * sizeof(uint2);
    int per_cu = 0;
    int rc = kernel_shape(ctx, (const void *)fn, stack_lds, per_cu);
    if (rc) return rc;
    if (ctx->blocks_per_cu_cap > 0 && per_cu > ctx->blocks_per_cu_cap) per_cu = ctx->blocks_per_cu_cap;
    if (streaming_tree(ctx) && ctx->blocks_per_cu_cap == 0 && per_cu > 6) per_cu = 6;
    long n_top = (ctx->lds_top < 0 && streaming_tree(ctx)) ? 0 : ctx->lds_top;
    if (n_top < 0 && per_cu >= 1) n_top = ((long)(ctx->lds_per_cu / (size_t)per_cu / kLdsGranule * kLdsGranule) - (long)stack_lds) / (long)sizeof(WideNode);
    if (n_top > ctx->n_wide) n_top = ctx->n_wide;
    if (n_top > kLdsTopMax) n_top = kLdsTopMax;
    if (n_top < 0 || ctx->root_ref < 0 || ctx->root_ref == HPT_REF_DONE) n_top = 0;
    for (; n_top > 0 && ctx->lds_top < 0; n_top -= 8) {   // the occupancy query has the last word (a count fixed by HPT_LDS_TOP_NODES is taken as is: fewer blocks then)
        int with_top = 0;
        if ((rc = kernel_shape(ctx, (const void *)fn, stack_lds + (size_t)n_top * sizeof(WideNode), with_top))) return rc;
        if (with_top >= per_cu) break;
    }
    n_lds_nodes = (uint32_t)(n_top > 0 ? n_top : 0);
    return HPT_OK;
}

// Launch shape of a persistent traversal kernel (render_kernel, iter_trace_kernel): LDS for the capped
// traversal stack, as many blocks as stay resident, HBM overflow of the stack sized for that grid.
// noalloc: the overflow lives at the start of the device's static scratch array (and the grid shrinks until it
// fits half of it); *scratch_used receives the bytes taken.
int prepare_persistent(hpt_ctx *ctx, const void *kernel, RenderParams &p, size_t &lds, long &blocks_max,
                       bool noalloc = false, size_t *scratch_used = nullptr) {
    const int depth = ctx->tree_depth > 0 ? ctx->tree_depth : 1;
    const int cap = stack_cap_of(ctx) < depth ? stack_cap_of(ctx) : depth;
    lds = (size_t)cap * kBlock * sizeof(uint2) + ctx->extra_lds;
    int per_cu = 0;
    int rc = kernel_shape(ctx, kernel, lds, per_cu);
    if (rc) return rc;
    if (per_cu < 1) return fail(ctx, HPT_ERR_INVALID, "render kernel does not fit: %zu B of LDS per block", lds);
    if (ctx->blocks_per_cu_cap > 0 && per_cu > ctx->blocks_per_cu_cap) per_cu = ctx->blocks_per_cu_cap;
    blocks_max = (long)per_cu * ctx->n_cu;
    p.stack_cap = cap;
    const size_t per_block = (size_t)(depth - cap) * kBlock * sizeof(uint2);
    if (noalloc) {
        if (per_block > 0 && (size_t)blocks_max * per_block > kScratchSpillMax) blocks_max = (long)(kScratchSpillMax / per_block);
        if (blocks_max < ctx->n_cu)
            return fail(ctx, HPT_ERR_INVALID, "hpt_render_noalloc: a tree of depth %d needs %zu B of stack overflow per block; "
                                              "fewer than one block per CU fits the %zu MB scratch", depth, per_block, kScratchBytes >> 20);
        const size_t used = ((size_t)blocks_max * per_block + 255) & ~(size_t)255;
        if (scratch_used) *scratch_used = used;
        p.spill = reinterpret_cast<uint2 *>(ctx->scratch);
        return HPT_OK;
    }
    const size_t spill_need = (size_t)blocks_max * per_block;
    if (spill_need > ctx->spill_bytes) {
        if (ctx->d_spill) { HIP_TRY(ctx, hipFree(ctx->d_spill)); ctx->d_spill = nullptr; ctx->spill_bytes = 0; }
        HIP_ALLOC(ctx, &ctx->d_spill, spill_need);
        ctx->spill_bytes = spill_need;
    }
    p.spill = ctx->d_spill;
    return HPT_OK;
}

// events [3b], [3b+1], [3b+2] of the render's b-th launch
int render_events(hpt_ctx *ctx, int launch) {
    while ((int)ctx->ev_render.size() < 3 * (launch + 1)) {
        hipEvent_t e = nullptr;
        HIP_TRY(ctx, hipEventCreate(&e));
        ctx->ev_render.push_back(e);
    }
    return HPT_OK;
}

// One render = one or more launches of render_kernel, each followed by fold_kernel.  A launch covers a range of the
// shard's pixel slots times a range of iterations whose samples fit the sample buffer: the whole shard and every
// iteration when that fits; else all iterations of as many 8x8 tiles as fit (a pixel is then folded once); else,
// for very long renders, one tile range and several iteration batches.  The split changes nothing in the image:
// fold_kernel folds a pixel's samples in iteration order whatever launch made them.
// noalloc (hpt_render_noalloc): nothing is allocated -- the stack overflow and the samples live in the device's
// static scratch array, no events are created or recorded.
int launch_render(hpt_ctx *ctx, float *d_image, int spp_begin, int spp_end, int raw, bool noalloc = false) {
    RenderParams p;
    if (noalloc && ctx->pairs_dirty && ctx->root_ref_tris != HPT_REF_DONE && ctx->max_leaf_end <= ctx->n_tris)
        return fail(ctx, HPT_ERR_INVALID, "hpt_render_noalloc: the scene is not resident yet (upload the BVH and the triangles first; after replacing only one of "
                                              "the two in a resident context, upload the other as well or call hpt_render once)");
    int rc = fill_params(ctx, p, d_image, spp_begin, spp_end, raw);
    if (rc) return rc;
    if (ctx->n_mats < 1 && ctx->root_ref != HPT_REF_DONE)
        return fail(ctx, HPT_ERR_INVALID, "triangles need materials[0] (ray_kernel.cl:752-754): call hpt_set_scene");
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_counters, 0, HPT_N_COUNTERS * sizeof(unsigned long long), ctx->stream));
    ctx->timed_launches = 0; ctx->launches = 0;
    const uint32_t shard_slots = p.n_slots;
    if (shard_slots == 0) return HPT_OK;

    // sample shards: this rank renders the iterations s = sample_rank + sample_world * l of the range, and folds
    // them as ITS running mean (count l + 1); hpt_kernel_execute (raw) names its iteration itself
    const int G = raw ? 1 : ctx->sample_world, r = raw ? 0 : ctx->sample_rank;
    auto locals_below = [&](int s) { return s <= r ? 0 : (s - r + G - 1) / G; };      // # of l with r + G*l < s
    const int l_begin = locals_below(spp_begin), l_end = locals_below(spp_end);
    const int total = l_end - l_begin;
    if (total <= 0) return HPT_OK;

    size_t lds; long blocks_max;
    size_t scratch_used = 0;
    render_fn fn; uint32_t n_top;
    if ((rc = choose_render(ctx, fn, n_top))) return rc;
    p.n_lds_nodes = n_top;
    ctx->extra_lds = (size_t)n_top * sizeof(WideNode);
    rc = prepare_persistent(ctx, (const void *)fn, p, lds, blocks_max, noalloc, &scratch_used);
    ctx->extra_lds = 0;
    if (rc) return rc;

    // paths (= 12-B samples) per launch: the sample buffer, and 32-bit work indices with headroom for the work
    // counter's overshoot (one chunk of <= 4096 per wave, < 2^16 waves)
    const size_t sample_bytes = 3 * sizeof(float);
    size_t capacity = (noalloc ? kScratchBytes - scratch_used : ctx->sample_budget_bytes) / sample_bytes;
    if (capacity > 0xF0000000u) capacity = 0xF0000000u;
    if (capacity < 64) capacity = 64;                     // (a budget below one tile: one tile, one iteration at a time)
    uint32_t range_slots; int batch;
    if ((size_t)shard_slots * (size_t)total <= capacity) { range_slots = shard_slots; batch = total; }
    else if ((capacity / (size_t)total) >= 64) { range_slots = (uint32_t)((capacity / (size_t)total) & ~(size_t)63); batch = total; }
    else { range_slots = 64; batch = (int)(capacity / 64); }
    if (range_slots > shard_slots) range_slots = shard_slots;

    if (noalloc) p.samples = reinterpret_cast<float *>(ctx->scratch + scratch_used);
    else {
        size_t samples_need = (size_t)range_slots * (size_t)batch * sample_bytes;
        while (samples_need > ctx->samples_bytes) {
            if (ctx->d_samples) { HIP_TRY(ctx, hipFree(ctx->d_samples)); ctx->d_samples = nullptr; ctx->samples_bytes = 0; }
            hipError_t e = hipMalloc((void **)&ctx->d_samples, samples_need);
            if (e == hipSuccess) { ctx->samples_bytes = samples_need; break; }
            (void)hipGetLastError();
            if (e != hipErrorOutOfMemory) return fail(ctx, HPT_ERR_DEVICE, "hipMalloc: %s", hipGetErrorString(e));
            // a busy device: halve the launch (fewer tiles, then fewer iterations) before giving up
            if (range_slots > 64) range_slots = ((range_slots / 2) + 63) & ~63u;
            else if (batch > 1) batch = (batch + 1) / 2;
            else return fail(ctx, HPT_ERR_NOMEM, "out of device memory allocating %zu bytes of samples", samples_need);
            samples_need = (size_t)range_slots * (size_t)batch * sample_bytes;
        }
        p.samples = ctx->d_samples;
    }

    int launch = 0;
    for (uint32_t slot0 = 0; slot0 < shard_slots; slot0 += range_slots) {
        p.slot_base = slot0;
        p.n_slots = (shard_slots - slot0) < range_slots ? (shard_slots - slot0) : range_slots;
        for (int base = l_begin; base < l_end; base += batch, launch++) {
            p.spp_base = base;
            p.spp_first = r + G * base; p.spp_stride = G;
            p.batch = (l_end - base) < batch ? (l_end - base) : batch;
            p.n_work = p.n_slots * (uint32_t)p.batch;
            long blocks = blocks_max;
            // a tree that does not fit the 256-MB Infinity Cache streams from HBM, and a seventh block per CU only adds
            // misses in flight (C4 166 -> 171 ms; 378 MB -1.1 %, 126 MB +0.9 %): six blocks per CU for those
            if (streaming_tree(ctx) && ctx->blocks_per_cu_cap == 0 && blocks > 6L * ctx->n_cu) blocks = 6L * ctx->n_cu;      // (HPT_BLOCKS_PER_CU overrides)
            // a launch of fewer than four paths per lane is mostly ramp and drain: one block per CU fewer then
            // (C1's 1 M paths: 0.737 -> 0.708 ms at 6 instead of 7 blocks per CU; from ~8 paths per lane the full grid wins)
            if (blocks >= 7L * ctx->n_cu && (long)p.n_work < 4L * blocks * kBlock) blocks -= ctx->n_cu;
            long need = ((long)p.n_work + kBlock - 1) / kBlock;
            if (blocks > need) blocks = need;
            ctx->grid_waves = (int)(blocks * (kBlock / 64));
            p.spill_stride = (uint32_t)(blocks * kBlock);
            p.tail_items = (uint32_t)ctx->tail_chunks * (uint32_t)ctx->grid_waves * (uint32_t)p.fetch_chunk;
            HIP_TRY(ctx, hipMemsetAsync(ctx->d_work, 0, sizeof(unsigned int), ctx->stream));
            if (!noalloc) {
                if ((rc = render_events(ctx, launch))) return rc;
                HIP_TRY(ctx, hipEventRecord(ctx->ev_render[3 * launch], ctx->stream));
            }
            hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(kBlock), lds, ctx->stream, p);
            HIP_TRY(ctx, hipGetLastError());
            if (!noalloc) HIP_TRY(ctx, hipEventRecord(ctx->ev_render[3 * launch + 1], ctx->stream));
            hipLaunchKernelGGL(fold_kernel, dim3((p.n_slots + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream, p);
            HIP_TRY(ctx, hipGetLastError());
            if (!noalloc) { HIP_TRY(ctx, hipEventRecord(ctx->ev_render[3 * launch + 2], ctx->stream)); ctx->timed_launches = launch + 1; }
        }
    }
    ctx->launches = launch;
    return HPT_OK;
}

}  // namespace

extern "C" {

const char *hpt_last_error(const hpt_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int hpt_create(int width, int height, int device, hpt_ctx **out) {
    if (!out) return fail(nullptr, HPT_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (width <= 0 || height <= 0) return fail(nullptr, HPT_ERR_INVALID, "bad image size %dx%d", width, height);
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev == 0)
        return fail(nullptr, HPT_ERR_DEVICE, "no HIP device (%s): the render path has no CPU fallback",
                    e == hipSuccess ? "count 0" : hipGetErrorString(e));
    if (device < 0 || device >= n_dev) return fail(nullptr, HPT_ERR_INVALID, "device %d of %d", device, n_dev);
    hpt_ctx *ctx = new hpt_ctx;
    ctx->device = device; ctx->width = width; ctx->height = height;
    auto bail = [&](const char *what, hipError_t err) {
        const int code = err == hipErrorOutOfMemory ? HPT_ERR_NOMEM : HPT_ERR_DEVICE;
        (void)hipGetLastError();          // the failed call's error is reported here; do not leave it for a later launch check
        fail(nullptr, code, "%s: %s", what, hipGetErrorString(err));
        hpt_destroy(ctx);
        return code;
    };
    if ((e = hipSetDevice(device)) != hipSuccess) return bail("hipSetDevice", e);
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) return bail("hipGetDeviceProperties", e);
    ctx->n_cu = prop.multiProcessorCount;
    if (const char *e = getenv("HPT_SHADE_THRESHOLD")) { int v = atoi(e); if (v >= 1 && v <= 64) { ctx->shade_threshold = v; ctx->shade_threshold_fixed = true; } }
    if (const char *e = getenv("HPT_TRIS_PER_TRIP")) { int v = atoi(e); if (v >= 1 && v <= 64) ctx->tris_per_trip = v; }
    if (const char *e = getenv("HPT_LEAF_RECORDS")) ctx->leaf_records = !strcmp(e, "singles") ? 1 : !strcmp(e, "pairs") ? 2 : 0;
    if (const char *e = getenv("HPT_NODE_REPEAT_MIN")) { int v = atoi(e); if (v >= 1 && v <= 65) ctx->node_repeat_min = v; }
    if (const char *e = getenv("HPT_PAIR_FETCH")) ctx->pair_fetch = atoi(e) != 0;
    if (const char *e = getenv("HPT_TRI_WEIGHT")) { int v = atoi(e); if (v >= 1 && v <= 256) ctx->tri_weight = v; }
    if (const char *e = getenv("HPT_SAMPLE_BUDGET_MB")) { long v = atol(e); if (v >= 1) ctx->sample_budget_bytes = (size_t)v << 20; }
    if (const char *e = getenv("HPT_FETCH_CHUNK")) { int v = atoi(e); if (v >= 1 && v <= 4096) ctx->fetch_chunk = v; }
    if (const char *e = getenv("HPT_TAIL_CHUNKS")) { int v = atoi(e); if (v >= 0 && v <= 64) ctx->tail_chunks = v; }
    if (const char *e = getenv("HPT_FETCH_THRESHOLD")) { int v = atoi(e); if (v >= 1 && v <= 64) ctx->fetch_threshold = v; }
    if (const char *e = getenv("HPT_STACK_CAP")) { int v = atoi(e); if (v >= 1 && v <= 64) ctx->stack_cap = v; }
    if (const char *e = getenv("HPT_BLOCKS_PER_CU")) { int v = atoi(e); if (v >= 1) ctx->blocks_per_cu_cap = v; }
    if (const char *e = getenv("HPT_COUNT_LANE_LOADS")) ctx->count_loads = atoi(e) != 0;
    if (const char *e = getenv("HPT_LDS_TOP_NODES")) { int v = atoi(e); if (v >= -1 && v <= kLdsTopMax) ctx->lds_top = v; }
    if (prop.maxSharedMemoryPerMultiProcessor > 0) ctx->lds_per_cu = (size_t)prop.maxSharedMemoryPerMultiProcessor;
    size_t img = (size_t)width * height * 3 * sizeof(float);
    if ((e = hipMalloc((void **)&ctx->d_work, sizeof(unsigned int))) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMalloc((void **)&ctx->d_counters, HPT_N_COUNTERS * sizeof(unsigned long long))) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMalloc((void **)&ctx->d_sample, img)) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMalloc((void **)&ctx->d_accum, img)) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMalloc((void **)&ctx->d_hit_id, (size_t)width * height * 4)) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMalloc((void **)&ctx->d_hit_t, (size_t)width * height * 4)) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMemset(ctx->d_counters, 0, HPT_N_COUNTERS * sizeof(unsigned long long))) != hipSuccess) return bail("hipMemset", e);
    for (hipEvent_t *ev : {&ctx->ev_iter[0], &ctx->ev_iter[1], &ctx->ev_iter[2], &ctx->ev_recon[0], &ctx->ev_recon[1]})
        if ((e = hipEventCreate(ev)) != hipSuccess) return bail("hipEventCreate", e);
    ctx->ev_render.reserve(96);
    for (int k = 0; k < 3; k++) {
        hipEvent_t ev = nullptr;
        if ((e = hipEventCreate(&ev)) != hipSuccess) return bail("hipEventCreate", e);
        ctx->ev_render.push_back(ev);
    }
    ctx->shapes.reserve(16);
    // this device's copy of the no-allocation scratch array (hpt_render_noalloc), looked up with the device current
    void *scratch = nullptr;
    if ((e = scratch_of(device, &scratch)) != hipSuccess) return bail("hipGetSymbolAddress(hpt_noalloc_scratch)", e);
    ctx->scratch = static_cast<unsigned char *>(scratch);
    *out = ctx;
    return HPT_OK;
}

int hpt_destroy(hpt_ctx *ctx) {
    if (!ctx) return HPT_OK;
    (void)hipSetDevice(ctx->device);
    (void)hpt_comm_destroy(ctx);
    (void)hipFree(ctx->d_nodes); (void)hipFree(ctx->d_tris); (void)hipFree(ctx->d_prims); (void)hipFree(ctx->d_mats);
    (void)hipFree(ctx->d_work); (void)hipFree(ctx->d_counters); (void)hipFree(ctx->d_sample); (void)hipFree(ctx->d_accum);
    (void)hipFree(ctx->d_hit_id); (void)hipFree(ctx->d_hit_t); (void)hipFree(ctx->d_spill); (void)hipFree(ctx->d_samples);
    (void)hipFree(ctx->d_iter); (void)hipFree(ctx->d_pairs); (void)hipFree(ctx->d_prim_compact); (void)hipFree(ctx->d_prim_pairs);
    for (hipEvent_t ev : ctx->ev_render) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : ctx->ev_iter) if (ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : ctx->ev_recon) if (ev) (void)hipEventDestroy(ev);
    delete ctx;
    return HPT_OK;
}

int hpt_set_stream(hpt_ctx *ctx, void *hip_stream) {
    if (!ctx) return HPT_ERR_INVALID;
    ctx->stream = (hipStream_t)hip_stream;
    return HPT_OK;
}

int hpt_set_shard(hpt_ctx *ctx, int rank, int world) {
    if (!ctx) return HPT_ERR_INVALID;
    if (world < 1 || rank < 0 || rank >= world) return fail(ctx, HPT_ERR_INVALID, "bad shard %d of %d", rank, world);
    ctx->rank = rank; ctx->world = world;
    return HPT_OK;
}

int hpt_set_sample_shard(hpt_ctx *ctx, int rank, int world) {
    if (!ctx) return HPT_ERR_INVALID;
    if (world < 1 || rank < 0 || rank >= world) return fail(ctx, HPT_ERR_INVALID, "bad sample shard %d of %d", rank, world);
    ctx->sample_rank = rank; ctx->sample_world = world;
    return HPT_OK;
}

int hpt_upload_triangles(hpt_ctx *ctx, const hpt_triangle *triangles, int n_triangles) {
    int rc = bind(ctx);
    if (rc) return rc;
    if (n_triangles < 0 || (n_triangles > 0 && !triangles)) return fail(ctx, HPT_ERR_INVALID, "bad triangle array");
    if (n_triangles >= (1 << 25)) return fail(ctx, HPT_ERR_INVALID, "at most 2^25-1 triangles (leaf reference packing)");
    rc = replace_buffer(ctx, ctx->d_tris, triangles, (size_t)n_triangles * sizeof(hpt_triangle));
    if (rc) return rc;
    ctx->n_tris = n_triangles;
    ctx->h_tris.assign(triangles, triangles + n_triangles);
    ctx->pairs_dirty = true;
    ctx->tris_gen++;
    // builds the traversal's records once tree and triangles are both here -- unless these triangles are the first half of a
    // replacement (the tree is the one of the last build): then the new tree's upload builds, or the next render
    if (ctx->built_bvh_gen == ctx->bvh_gen && ctx->built_tris_gen != 0) return HPT_OK;
    return ensure_pairs(ctx);
}

// Depth-first reference arrays -> 64-B two-child nodes.  Interior node k of the
// reference becomes one WideNode holding the boxes of k+1 and secondChildOffset;
// leaves live only as references in their parent.  Wide nodes are numbered in
// the same pre-order, so a near-first descent walks forward in memory.
int hpt_upload_bvh(hpt_ctx *ctx, const hpt_bvh_bbox *bbox, const hpt_bvh_info *info, int n_nodes) {
    int rc = bind(ctx);
    if (rc) return rc;
    if (n_nodes < 0 || (n_nodes > 0 && (!bbox || !info))) return fail(ctx, HPT_ERR_INVALID, "bad BVH arrays");
    if (ctx->d_nodes) { HIP_TRY(ctx, hipFree(ctx->d_nodes)); ctx->d_nodes = nullptr; }
    ctx->n_wide = 0; ctx->tree_depth = 0; ctx->max_leaf_end = 0; ctx->root_ref = HPT_REF_DONE; ctx->root_ref_tris = HPT_REF_DONE;
    ctx->h_wide.clear();
    ctx->pairs_dirty = true;
    ctx->bvh_gen++;
    if (n_nodes == 0) return HPT_OK;

    auto leaf_ref = [&](int k, int &ref) -> int {
        const hpt_bvh_info &n = info[k];
        if (n.nPrimitives > HPT_LEAF_MAX_PRIMS)
            return fail(ctx, HPT_ERR_BVH, "leaf %d has %d triangles (limit %d)", k, n.nPrimitives, HPT_LEAF_MAX_PRIMS);
        if (n.offset < 0 || n.offset >= (1 << 25)) return fail(ctx, HPT_ERR_BVH, "leaf %d: bad triangle offset %d", k, n.offset);
        if (n.offset + n.nPrimitives > ctx->max_leaf_end) ctx->max_leaf_end = n.offset + n.nPrimitives;
        ref = HPT_LEAF_REF(n.offset, n.nPrimitives);
        return HPT_OK;
    };

    for (int k = 0; k < n_nodes; k++) {          // lo <= hi, no NaN: the min3/max3 slab test relies on it
        const hpt_bvh_bbox &b = bbox[k];
        if (!(b.minboundsx <= b.maxboundsx && b.minboundsy <= b.maxboundsy && b.minboundsz <= b.maxboundsz))
            return fail(ctx, HPT_ERR_BVH, "node %d: bounding box with min > max or NaN", k);
    }
    // pass 1: walk the tree from node 0, number interior nodes in pre-order, validate
    std::vector<int> wide_of(n_nodes, -1);
    std::vector<unsigned char> seen(n_nodes, 0);
    struct Item { int node, depth; };
    std::vector<Item> todo;
    todo.push_back({0, 0});
    int n_wide = 0, depth_max = 0;
    while (!todo.empty()) {
        Item it = todo.back(); todo.pop_back();
        int k = it.node;
        if (k < 0 || k >= n_nodes) return fail(ctx, HPT_ERR_BVH, "child index %d out of range [0,%d)", k, n_nodes);
        if (seen[k]) return fail(ctx, HPT_ERR_BVH, "node %d is reachable twice (not a tree)", k);
        seen[k] = 1;
        if (info[k].nPrimitives > 0) continue;
        if (info[k].nPrimitives < 0) return fail(ctx, HPT_ERR_BVH, "node %d: negative nPrimitives", k);
        if (info[k].axis < 0 || info[k].axis > 2) return fail(ctx, HPT_ERR_BVH, "node %d: axis %d", k, info[k].axis);
        if (it.depth + 1 > kMaxTreeDepth)
            return fail(ctx, HPT_ERR_BVH, "tree deeper than %d (the reference's todo[64] bound)", kMaxTreeDepth);
        if (it.depth + 1 > depth_max) depth_max = it.depth + 1;
        wide_of[k] = n_wide++;
        todo.push_back({info[k].secondChildOffset, it.depth + 1});   // popped after the first child: pre-order
        todo.push_back({k + 1, it.depth + 1});
    }

    // the NODE step addresses nodes by a 32-bit byte offset (render_kernels.h trav_node_steps)
    if (n_wide >= (1 << 25)) return fail(ctx, HPT_ERR_BVH, "%d interior nodes: at most 2^25-1 (32-bit node byte offsets)", n_wide);

    // pass 2: fill
    std::vector<WideNode> wide(n_wide);
    for (int k = 0; k < n_nodes; k++) {
        if (wide_of[k] < 0) continue;
        WideNode &w = wide[wide_of[k]];
        int c0 = k + 1, c1 = info[k].secondChildOffset;
        w.lo0x = bbox[c0].minboundsx; w.lo0y = bbox[c0].minboundsy; w.lo0z = bbox[c0].minboundsz;
        w.hi0x = bbox[c0].maxboundsx; w.hi0y = bbox[c0].maxboundsy; w.hi0z = bbox[c0].maxboundsz;
        w.lo1x = bbox[c1].minboundsx; w.lo1y = bbox[c1].minboundsy; w.lo1z = bbox[c1].minboundsz;
        w.hi1x = bbox[c1].maxboundsx; w.hi1y = bbox[c1].maxboundsy; w.hi1z = bbox[c1].maxboundsz;
        w.axis = info[k].axis; w.pad = 0;
        if (info[c0].nPrimitives > 0) { if ((rc = leaf_ref(c0, w.ref0))) return rc; } else w.ref0 = wide_of[c0];
        if (info[c1].nPrimitives > 0) { if ((rc = leaf_ref(c1, w.ref1))) return rc; } else w.ref1 = wide_of[c1];
    }
    if (info[0].nPrimitives > 0) { if ((rc = leaf_ref(0, ctx->root_ref_tris))) return rc; }
    else ctx->root_ref_tris = wide_of[0];
    ctx->root_ref = ctx->root_ref_tris;                  // "there is a tree"; ensure_pairs() re-points a leaf root
    ctx->root_lo[0] = bbox[0].minboundsx; ctx->root_lo[1] = bbox[0].minboundsy; ctx->root_lo[2] = bbox[0].minboundsz;
    ctx->root_hi[0] = bbox[0].maxboundsx; ctx->root_hi[1] = bbox[0].maxboundsy; ctx->root_hi[2] = bbox[0].maxboundsz;
    if (ctx->root_ref_tris >= 0) {
        // Number the top of the tree first, breadth first: any prefix of that order is "the top n nodes", which is what
        // render_kernel copies to LDS.  Child references are explicit, so any numbering is a valid tree; the rest keeps its
        // pre-order (a near-first descent still walks forward in memory there).
        const int K = kLdsTopMax < n_wide ? kLdsTopMax : n_wide;
        std::vector<int> order; order.reserve(n_wide);
        std::vector<unsigned char> taken(n_wide, 0);
        order.push_back(ctx->root_ref_tris); taken[ctx->root_ref_tris] = 1;
        for (size_t head = 0; head < order.size() && (int)order.size() < K; head++)
            for (int32_t ref : {wide[order[head]].ref0, wide[order[head]].ref1})
                if (ref >= 0 && (int)order.size() < K) { order.push_back(ref); taken[ref] = 1; }
        for (int i = 0; i < n_wide; i++) if (!taken[i]) order.push_back(i);
        std::vector<int> new_of(n_wide);
        for (int i = 0; i < n_wide; i++) new_of[order[i]] = i;
        std::vector<WideNode> re(n_wide);
        for (int i = 0; i < n_wide; i++) {
            WideNode w = wide[order[i]];
            if (w.ref0 >= 0) w.ref0 = new_of[w.ref0];
            if (w.ref1 >= 0) w.ref1 = new_of[w.ref1];
            re[i] = w;
        }
        wide.swap(re);
        ctx->root_ref_tris = new_of[ctx->root_ref_tris];
        ctx->root_ref = ctx->root_ref_tris;
    }
    ctx->h_wide.swap(wide);                              // uploaded by ensure_pairs() with leaf references re-pointed
    ctx->n_wide = n_wide;
    ctx->tree_depth = depth_max;
    if (ctx->built_tris_gen == ctx->tris_gen && ctx->built_bvh_gen != 0) return HPT_OK;   // first half of a replacement (see tris_gen)
    return ensure_pairs(ctx);           // (a no-op until the triangles the leaves name are uploaded too)
}

int hpt_set_scene(hpt_ctx *ctx, const hpt_primitive *primitives, int n_primitives,
                  const hpt_material *materials, int n_materials) {
    int rc = bind(ctx);
    if (rc) return rc;
    if (n_primitives < 0 || n_materials < 0 || (n_primitives > 0 && !primitives) || (n_materials > 0 && !materials))
        return fail(ctx, HPT_ERR_INVALID, "bad scene arrays");
    if ((rc = replace_buffer(ctx, ctx->d_prims, primitives, (size_t)n_primitives * sizeof(hpt_primitive)))) return rc;
    std::vector<PrimCompact> compact((size_t)n_primitives);
    for (int i = 0; i < n_primitives; i++) {
        const hpt_primitive &q = primitives[i];
        compact[i] = PrimCompact{q.center.x, q.center.y, q.center.z, q.radius, q.type.x, q.plane_normal.x, q.plane_normal.y, q.plane_normal.z};
    }
    if ((rc = replace_buffer(ctx, ctx->d_prim_compact, compact.data(), compact.size() * sizeof(PrimCompact)))) return rc;
    std::vector<PrimPair> pairs(((size_t)n_primitives + 1) / 2);
    for (size_t j = 0; j < pairs.size(); j++)
        for (int h = 0; h < 2; h++) {
            PrimPair &pp = pairs[j];
            const size_t i = 2 * j + (size_t)h;
            if (i >= (size_t)n_primitives) { pp.cx[h] = pp.cy[h] = pp.cz[h] = pp.rr[h] = pp.nx[h] = pp.ny[h] = pp.nz[h] = 0.0f; pp.type[h] = -1.0f; continue; }
            const PrimCompact &c = compact[i];
            const volatile float r = c.radius;            // one fp32 multiply, rounded once (no contraction, no excess precision)
            const volatile float rr = r * r;
            pp.cx[h] = c.cx; pp.cy[h] = c.cy; pp.cz[h] = c.cz; pp.rr[h] = rr; pp.type[h] = c.type; pp.nx[h] = c.nx; pp.ny[h] = c.ny; pp.nz[h] = c.nz;
        }
    if ((rc = replace_buffer(ctx, ctx->d_prim_pairs, pairs.data(), pairs.size() * sizeof(PrimPair)))) return rc;
    if ((rc = replace_buffer(ctx, ctx->d_mats, materials, (size_t)n_materials * sizeof(hpt_material)))) return rc;
    ctx->n_prims = n_primitives; ctx->n_mats = n_materials;
    if (!ctx->pairs_dirty && ctx->root_ref_tris != HPT_REF_DONE) {   // a long primitive list selects another instantiation: query it now
        render_fn fn; uint32_t n_top;
        return choose_render(ctx, fn, n_top);
    }
    return HPT_OK;
}

int hpt_set_camera(hpt_ctx *ctx, const hpt_camera *camera) {
    if (!ctx || !camera) return HPT_ERR_INVALID;
    memcpy(&ctx->cam, camera, sizeof(hpt_camera));
    ctx->have_camera = true;
    return HPT_OK;
}

int hpt_set_config(hpt_ctx *ctx, const hpt_config *cfg) {
    if (!ctx || !cfg) return HPT_ERR_INVALID;
    if (cfg->max_depth < 1) return fail(ctx, HPT_ERR_INVALID, "max_depth %d: a path has at least its camera ray", cfg->max_depth);
    ctx->cfg = *cfg;
    return HPT_OK;
}

int hpt_get_config(const hpt_ctx *ctx, hpt_config *cfg) {
    if (!ctx || !cfg) return HPT_ERR_INVALID;
    *cfg = ctx->cfg;
    return HPT_OK;
}

int hpt_render(hpt_ctx *ctx, int spp_begin, int spp_end, float *d_image) {
    int rc = bind(ctx);
    if (rc) return rc;
    if (!d_image || spp_begin < 0 || spp_end < spp_begin) return fail(ctx, HPT_ERR_INVALID, "bad render range [%d,%d)", spp_begin, spp_end);
    if (spp_end == spp_begin) return HPT_OK;
    return launch_render(ctx, d_image, spp_begin, spp_end, 0);
}

int hpt_image_alloc(hpt_ctx *ctx, float **d_image) {
    int rc = bind(ctx);
    if (rc) return rc;
    if (!d_image) return fail(ctx, HPT_ERR_INVALID, "bad image_alloc arguments");
    *d_image = nullptr;
    const size_t bytes = (size_t)ctx->width * ctx->height * 3 * sizeof(float);
    float *ptr = nullptr;
    HIP_ALLOC(ctx, &ptr, bytes);
    hipError_t e = hipMemsetAsync(ptr, 0, bytes, ctx->stream);
    if (e != hipSuccess) { (void)hipFree(ptr); return fail(ctx, HPT_ERR_DEVICE, "hipMemsetAsync: %s", hipGetErrorString(e)); }
    *d_image = ptr;
    return HPT_OK;
}

int hpt_image_free(hpt_ctx *ctx, float *d_image) {
    int rc = bind(ctx);
    if (rc) return rc;
    if (!d_image) return HPT_OK;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));      // work that still writes it
    HIP_TRY(ctx, hipFree(d_image));
    return HPT_OK;
}

int hpt_read_image(hpt_ctx *ctx, const float *d_image, float *image_rgb) {
    int rc = bind(ctx);
    if (rc) return rc;
    if (!d_image || !image_rgb) return fail(ctx, HPT_ERR_INVALID, "bad read_image arguments");
    HIP_TRY(ctx, hipMemcpyAsync(image_rgb, d_image, (size_t)ctx->width * ctx->height * 3 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return HPT_OK;
}

int hpt_render_noalloc(hpt_ctx *ctx, int spp_begin, int spp_end, float *d_image) {
    int rc = bind(ctx);
    if (rc) return rc;
    if (!d_image || spp_begin < 0 || spp_end < spp_begin) return fail(ctx, HPT_ERR_INVALID, "bad render range [%d,%d)", spp_begin, spp_end);
    if (spp_end == spp_begin) return HPT_OK;
    return launch_render(ctx, d_image, spp_begin, spp_end, 0, true);
}

int64_t hpt_noalloc_scratch_bytes(void) { return (int64_t)kScratchBytes; }

int hpt_kernel_execute(hpt_ctx *ctx, hpt_pixel *image, int iteration) {
    int rc = bind(ctx);
    if (rc) return rc;
    if (!image || iteration < 0) return fail(ctx, HPT_ERR_INVALID, "bad kernel_execute arguments");
    size_t img = (size_t)ctx->width * ctx->height * 3 * sizeof(float);
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_sample, 0, img, ctx->stream));
    if ((rc = launch_render(ctx, ctx->d_sample, iteration, iteration + 1, 1))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(image, ctx->d_sample, img, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return HPT_OK;
}

int hpt_reconstruct(hpt_ctx *ctx, float *d_image, const float *d_sample, int n_pixels, int iterations) {
    int rc = bind(ctx);
    if (rc) return rc;
    if (!d_image || !d_sample || n_pixels < 0) return fail(ctx, HPT_ERR_INVALID, "bad reconstruct arguments");
    long n = 3L * n_pixels;
    ctx->timed_recon = false;
    if (n == 0) return HPT_OK;
    HIP_TRY(ctx, hipEventRecord(ctx->ev_recon[0], ctx->stream));
    hipLaunchKernelGGL(reconstruct_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_image, d_sample, n, iterations);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(ctx->ev_recon[1], ctx->stream));
    ctx->timed_recon = true;
    return HPT_OK;
}

// ---- iterative pipeline -------------------------------------------------------------------------------
static const size_t kIterBytesPerPixel = 7 * 16 + 3 * 8 + 4;    // 7 float4, 2 uint2 + 1 int2, 1 int32

static int iter_launch_params(hpt_ctx *ctx, RenderParams &p, unsigned &blocks) {
    int rc = fill_params(ctx, p, nullptr, 0, 1, 0);
    if (rc) return rc;
    if (ctx->n_mats < 1 && ctx->root_ref != HPT_REF_DONE)
        return fail(ctx, HPT_ERR_INVALID, "triangles need materials[0] (ray_kernel.cl:752-754): call hpt_set_scene");
    blocks = (p.n_slots + kBlock - 1) / kBlock;
    return HPT_OK;
}

// traversal half of a call: persistent iter_trace_kernel over this shard's pixel slots
static int iter_trace(hpt_ctx *ctx, RenderParams &p, int first) {
    typedef void (*trace_fn)(const RenderParams, const IterState, int);
    static const trace_fn kTrace[2][2] = {{iter_trace_kernel<false, false>, iter_trace_kernel<false, true>},
                                          {iter_trace_kernel<true, false>, iter_trace_kernel<true, true>}};
    const trace_fn fn = kTrace[ctx->singles ? 1 : 0][ctx->n_prims >= HPT_TWO_PHASE_MIN_PRIMS ? 1 : 0];
    size_t lds; long blocks_max;
    int rc = prepare_persistent(ctx, (const void *)fn, p, lds, blocks_max);
    if (rc) return rc;
    long blocks = blocks_max;
    long need = ((long)p.n_slots + kBlock - 1) / kBlock;
    if (blocks > need) blocks = need;
    p.spill_stride = (uint32_t)(blocks * kBlock);
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_work, 0, sizeof(unsigned int), ctx->stream));
    HIP_TRY(ctx, hipEventRecord(ctx->ev_iter[0], ctx->stream));
    hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(kBlock), lds, ctx->stream, p, ctx->iter, first);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(ctx->ev_iter[1], ctx->stream));
    return HPT_OK;
}

int hpt_first_bounce(hpt_ctx *ctx) {
    int rc = bind(ctx);
    if (rc) return rc;
    RenderParams p; unsigned blocks;
    if ((rc = iter_launch_params(ctx, p, blocks))) return rc;
    const size_t npx = (size_t)ctx->width * ctx->height;
    if (!ctx->d_iter) {
        HIP_ALLOC(ctx, &ctx->d_iter, npx * kIterBytesPerPixel);
        char *q = (char *)ctx->d_iter;                       // 16-B members first: every array stays aligned
        ctx->iter.cur_a = (float4 *)q; q += npx * 16; ctx->iter.cur_b = (float4 *)q; q += npx * 16;
        ctx->iter.cur_c = (float4 *)q; q += npx * 16; ctx->iter.c_a = (float4 *)q; q += npx * 16;
        ctx->iter.c_b = (float4 *)q; q += npx * 16; ctx->iter.c_c = (float4 *)q; q += npx * 16;
        ctx->iter.contrib = (float4 *)q; q += npx * 16;
        ctx->iter.cur_d = (uint2 *)q; q += npx * 8; ctx->iter.c_d = (uint2 *)q; q += npx * 8;
        ctx->iter.hit = (int2 *)q; q += npx * 8;
        ctx->iter.completed = (int32_t *)q;
    }
    // pixels outside the shard keep zeroed state (count 0, no contribution)
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_iter, 0, npx * kIterBytesPerPixel, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_counters, 0, HPT_N_COUNTERS * sizeof(unsigned long long), ctx->stream));
    ctx->timed_launches = 0; ctx->launches = 0; ctx->timed_iter = false; ctx->timed_recon = false;
    ctx->iter_ready = 1; ctx->iter_pending = 0;
    if (blocks == 0) return HPT_OK;
    if ((rc = iter_trace(ctx, p, 1))) return rc;
    hipLaunchKernelGGL(iter_first_shade_kernel, dim3(blocks), dim3(kBlock), 0, ctx->stream, p, ctx->iter);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(ctx->ev_iter[2], ctx->stream));
    ctx->timed_iter = true;
    return HPT_OK;
}

int hpt_bounce(hpt_ctx *ctx) {
    int rc = bind(ctx);
    if (rc) return rc;
    if (!ctx->iter_ready) return fail(ctx, HPT_ERR_INVALID, "hpt_bounce before hpt_first_bounce");
    if (ctx->iter_pending) return fail(ctx, HPT_ERR_INVALID, "contributions of the previous hpt_bounce are not folded yet: call hpt_reconstruct_paths");
    RenderParams p; unsigned blocks;
    if ((rc = iter_launch_params(ctx, p, blocks))) return rc;
    ctx->iter_pending = 1;
    ctx->timed_iter = false;
    if (blocks == 0) return HPT_OK;
    if ((rc = iter_trace(ctx, p, 0))) return rc;
    hipLaunchKernelGGL(iter_bounce_shade_kernel, dim3(blocks), dim3(kBlock), 0, ctx->stream, p, ctx->iter);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(ctx->ev_iter[2], ctx->stream));
    ctx->timed_iter = true;
    return HPT_OK;
}

int hpt_reconstruct_paths(hpt_ctx *ctx, float *d_image) {
    int rc = bind(ctx);
    if (rc) return rc;
    if (!d_image) return fail(ctx, HPT_ERR_INVALID, "bad reconstruct_paths arguments");
    if (!ctx->iter_ready) return fail(ctx, HPT_ERR_INVALID, "hpt_reconstruct_paths before hpt_first_bounce");
    RenderParams p; unsigned blocks;
    if ((rc = iter_launch_params(ctx, p, blocks))) return rc;
    ctx->iter_pending = 0;
    ctx->timed_recon = false;
    if (blocks == 0) return HPT_OK;
    HIP_TRY(ctx, hipEventRecord(ctx->ev_recon[0], ctx->stream));
    hipLaunchKernelGGL(iter_reconstruct_kernel, dim3(blocks), dim3(kBlock), 0, ctx->stream, p, ctx->iter, d_image);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(ctx->ev_recon[1], ctx->stream));
    ctx->timed_recon = true;
    return HPT_OK;
}

int hpt_path_counts(hpt_ctx *ctx, int32_t *counts) {
    int rc = bind(ctx);
    if (rc) return rc;
    if (!counts) return fail(ctx, HPT_ERR_INVALID, "bad path_counts arguments");
    if (!ctx->iter_ready) return fail(ctx, HPT_ERR_INVALID, "hpt_path_counts before hpt_first_bounce");
    HIP_TRY(ctx, hipMemcpyAsync(counts, ctx->iter.completed, (size_t)ctx->width * ctx->height * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return HPT_OK;
}

int hpt_render_sample(hpt_ctx *ctx, int iteration, float *image_rgb) {
    int rc = bind(ctx);
    if (rc) return rc;
    if (!image_rgb || iteration < 0) return fail(ctx, HPT_ERR_INVALID, "bad render_sample arguments");
    size_t img = (size_t)ctx->width * ctx->height * 3 * sizeof(float);
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_sample, 0, img, ctx->stream));
    if ((rc = launch_render(ctx, ctx->d_sample, iteration, iteration + 1, 1))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_accum, image_rgb, img, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = hpt_reconstruct(ctx, ctx->d_accum, ctx->d_sample, ctx->width * ctx->height, iteration + 1))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(image_rgb, ctx->d_accum, img, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return HPT_OK;
}

int hpt_primary_hits(hpt_ctx *ctx, int iteration, int32_t *hit_id, float *t) {
    int rc = bind(ctx);
    if (rc) return rc;
    if (!hit_id || !t || iteration < 0) return fail(ctx, HPT_ERR_INVALID, "bad primary_hits arguments");
    RenderParams p;
    if ((rc = fill_params(ctx, p, nullptr, iteration, iteration + 1, 1))) return rc;
    int npx = ctx->width * ctx->height;
    size_t lds = stack_lds_bytes(ctx, kBlock);
    int unused_per_cu = 0;
    if ((rc = kernel_shape(ctx, (const void *)primary_hits_kernel, lds, unused_per_cu))) return rc;
    hipLaunchKernelGGL(primary_hits_kernel, dim3((unsigned)((npx + kBlock - 1) / kBlock)), dim3(kBlock), lds, ctx->stream, p, iteration);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(hit_id, ctx->d_hit_id, (size_t)npx * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(t, ctx->d_hit_t, (size_t)npx * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return HPT_OK;
}

int64_t hpt_slab_floats(const hpt_ctx *ctx) {
    if (!ctx) return HPT_ERR_INVALID;
    return (int64_t)slab_tiles(ctx) * 64 * 3;
}

int hpt_pack_tiles(hpt_ctx *ctx, const float *d_image, float *d_slab) {
    int rc = bind(ctx);
    if (rc) return rc;
    if (!d_image || !d_slab) return fail(ctx, HPT_ERR_INVALID, "bad pack arguments");
    RenderParams p;
    memset(&p, 0, sizeof p);
    p.width = ctx->width; p.height = ctx->height; p.rank = ctx->rank; p.world = ctx->world;
    p.tiles_x = tiles_x(ctx); p.n_tiles = n_tiles(ctx);
    unsigned n = (unsigned)slab_tiles(ctx) * 64u;
    hipLaunchKernelGGL(pack_tiles_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, p, d_image, d_slab, n);
    HIP_TRY(ctx, hipGetLastError());
    return HPT_OK;
}

int hpt_unpack_tiles(hpt_ctx *ctx, const float *d_slabs, float *d_image) {
    int rc = bind(ctx);
    if (rc) return rc;
    if (!d_slabs || !d_image) return fail(ctx, HPT_ERR_INVALID, "bad unpack arguments");
    RenderParams p;
    memset(&p, 0, sizeof p);
    p.width = ctx->width; p.height = ctx->height; p.rank = 0; p.world = ctx->world;
    p.tiles_x = tiles_x(ctx); p.n_tiles = n_tiles(ctx);
    unsigned n = (unsigned)slab_tiles(ctx) * 64u;
    unsigned total = n * (unsigned)ctx->world;
    hipLaunchKernelGGL(unpack_tiles_kernel, dim3((total + 255) / 256), dim3(256), 0, ctx->stream, p, d_slabs, d_image, n);
    HIP_TRY(ctx, hipGetLastError());
    return HPT_OK;
}

int hpt_tonemap(hpt_ctx *ctx, const float *d_image, uint8_t *d_rgb8, int n_pixels) {
    int rc = bind(ctx);
    if (rc) return rc;
    if (!d_image || !d_rgb8 || n_pixels < 0) return fail(ctx, HPT_ERR_INVALID, "bad tonemap arguments");
    long n = 3L * n_pixels;
    if (n == 0) return HPT_OK;
    hipLaunchKernelGGL(tonemap_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_image, d_rgb8, n);
    HIP_TRY(ctx, hipGetLastError());
    return HPT_OK;
}

int hpt_synchronize(hpt_ctx *ctx) {
    int rc = bind(ctx);
    if (rc) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return HPT_OK;
}

int hpt_get_stats(hpt_ctx *ctx, hpt_stats *stats) {
    int rc = bind(ctx);
    if (rc) return rc;
    if (!stats) return HPT_ERR_INVALID;
    memset(stats, 0, sizeof *stats);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    unsigned long long c[HPT_N_COUNTERS];
    HIP_TRY(ctx, hipMemcpy(c, ctx->d_counters, sizeof c, hipMemcpyDeviceToHost));
    stats->rays_primary = c[0]; stats->rays_bounce = c[1]; stats->paths = c[2];
    stats->lane_loads = c[HPT_COUNTER_LOADS];
#ifdef HPT_PROFILE_STEPS
    fprintf(stderr, "[hpt profile] trips S/T/N %llu %llu %llu  lane-active S/T/N %llu %llu %llu  util %.3f %.3f %.3f\n",
            c[3] / 64, c[4] / 64, c[5] / 64, c[6], c[7], c[8], c[3] ? (double)c[6] / c[3] : 0.0, c[4] ? (double)c[7] / c[4] : 0.0,
            c[5] ? (double)c[8] / c[5] : 0.0);
    fprintf(stderr, "[hpt profile] NODE iterations %llu (%.1f lanes each), all lanes at one node: %.3f of iterations; distinct nodes per quad / lane-visits %.3f\n",
            c[11], c[11] ? (double)c[13] / c[11] : 0.0, c[11] ? (double)(c[12] & 0xfffffull) / c[11] : 0.0, c[13] ? (double)c[14] / c[13] : 0.0);
    fprintf(stderr, "[hpt profile] distinct nodes per WAVE / lane-visits %.3f\n", c[13] ? (double)(c[12] >> 20) / c[13] : 0.0);
    fprintf(stderr, "[hpt profile] of all lane-trips: waiting for SHADE %.3f, out of work %.3f\n",
            (double)c[9] / (double)(c[3] + c[4] + c[5]), (double)c[10] / (double)(c[3] + c[4] + c[5]));
#endif
    for (int b = 0; b < ctx->timed_launches; b++) {
        float ms = 0;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev_render[3 * b], ctx->ev_render[3 * b + 1]));
        stats->last_kernel_ms += ms;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev_render[3 * b + 1], ctx->ev_render[3 * b + 2]));
        stats->fold_ms += ms;
    }
    if (ctx->timed_iter) {
        float ms = 0;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev_iter[0], ctx->ev_iter[1]));
        stats->trace_ms = ms;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev_iter[1], ctx->ev_iter[2]));
        stats->shade_ms = ms;
    }
    if (ctx->timed_recon) {
        float ms = 0;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev_recon[0], ctx->ev_recon[1]));
        stats->reconstruct_ms = ms;
    }
    stats->wide_nodes = ctx->n_wide; stats->tree_depth = ctx->tree_depth; stats->grid_waves = ctx->grid_waves;
    stats->launches = ctx->launches;
    return HPT_OK;
}

}  // extern "C"
