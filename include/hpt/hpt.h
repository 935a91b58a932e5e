/* hpt/hpt.h — C-ABI of the MI355X (gfx950) HIP render backend.
 *
 * This is the drop-in boundary: one entry point per public method of the
 * reference's device backend `class OpenCL_Manager`
 * (reference pathtracer/opencl_manager.hpp:11-50), whose only caller is
 * `class Pathtracer` (pathtracer/pathtracer.cpp:69-72, 142, 166, 180, 192, 301, 316).
 * extern "C", plain pointers and sizes, opaque context.  Every function
 * returns HPT_OK (0) or a negative hpt_status; hpt_last_error() gives the
 * text.  Nothing in the library calls exit() (the reference does:
 * opencl_manager.cpp:52-65, 113-130).
 *
 * Pointers named `d_*` are DEVICE pointers (HIP memory of the context's
 * device); all others are host pointers.  One host thread per context.
 */
#ifndef HPT_HPT_H
#define HPT_HPT_H

#include "types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hpt_ctx hpt_ctx;

typedef enum {
    HPT_OK = 0,
    HPT_ERR_INVALID = -1,   /* bad argument / call order */
    HPT_ERR_DEVICE = -2,    /* HIP runtime error (text in hpt_last_error) */
    HPT_ERR_NOMEM = -3,
    HPT_ERR_BVH = -4        /* malformed BVH arrays (cycle, child out of range, depth > limit) */
} hpt_status;

/* Replaces OpenCL_Manager::initialize(width,height) + kernel_load()/new_kernel_load()
 * (opencl_manager.hpp:13-19, opencl_manager.cpp:46-104, 107-187): kernels are
 * compiled ahead of time for gfx950, so the load calls disappear.
 * `device` is the HIP device ordinal. */
int hpt_create(int width, int height, int device, hpt_ctx **out);
int hpt_destroy(hpt_ctx *ctx);

/* Text of the last error on this context (ctx == NULL: last hpt_create error). */
const char *hpt_last_error(const hpt_ctx *ctx);

/* Launch stream (a hipStream_t passed as void*; NULL = the null stream).
 * No counterpart in the reference (it makes a new CommandQueue per call,
 * opencl_manager.cpp:284, 394, 547, 590). */
int hpt_set_stream(hpt_ctx *ctx, void *hip_stream);

/* Image-tile shard of this context: rank r of `world` renders the 8x8-pixel
 * tiles t with t % world == r.  Default (0,1) = whole image. */
int hpt_set_shard(hpt_ctx *ctx, int rank, int world);

/* Sample-range shard of this context (the secondary mode, for progressive rendering of one
 * image by several GPUs): rank r of `world` renders the iterations s with s % world == r of every pixel it
 * owns, and hpt_render's image is the running mean over ITS samples (pathtracer.cpp:200-217 with its own
 * count).  Possible because the random stream is keyed by (pixel, iteration), not sequential per pixel as in
 * the reference (ray_kernel.cl:259-266).  The frame is sum_r (n_r / N) * image_r, one all-reduce; the fp32
 * summation order differs from the unsharded mean, so that sum is within 1e-5 relative of it, not
 * bit-identical (each rank's image is).  Default (0,1).  No counterpart in the reference. */
int hpt_set_sample_shard(hpt_ctx *ctx, int rank, int world);

/* Replaces OpenCL_Manager::triangles_to_texture (opencl_manager.hpp:21,
 * opencl_manager.cpp:207-221).  Copies at call time. Triangles are in BVH leaf order. */
int hpt_upload_triangles(hpt_ctx *ctx, const hpt_triangle *triangles, int n_triangles);

/* Replaces OpenCL_Manager::bvh_to_texture (opencl_manager.hpp:22-24,
 * opencl_manager.cpp:223-253).  Copies at call time; validates the tree and
 * re-lays it out as 64-B two-child nodes. */
int hpt_upload_bvh(hpt_ctx *ctx, const hpt_bvh_bbox *bbox, const hpt_bvh_info *info, int n_nodes);

/* Hoists the per-call re-uploads of primitives/materials/camera that every
 * *_execute does (opencl_manager.cpp:287-291, 399-403, 595-599). */
int hpt_set_scene(hpt_ctx *ctx, const hpt_primitive *primitives, int n_primitives,
                  const hpt_material *materials, int n_materials);
int hpt_set_camera(hpt_ctx *ctx, const hpt_camera *camera);

/* The reference's compile-time literals (types.h hpt_config). */
int hpt_set_config(hpt_ctx *ctx, const hpt_config *cfg);
int hpt_get_config(const hpt_ctx *ctx, hpt_config *cfg);

/* Replaces OpenCL_Manager::kernel_execute (opencl_manager.hpp:27-31,
 * opencl_manager.cpp:569-648): ONE radiance sample per pixel of this shard
 * for 0-based `iteration` (it selects the 6x6 jitter cell and keys the RNG).
 * `image` receives width*height Pixels (pixels outside the shard are left 0). */
int hpt_kernel_execute(hpt_ctx *ctx, hpt_pixel *image, int iteration);

/* Replaces OpenCL_Manager::reconstruct_kernel_execute (opencl_manager.hpp:50,
 * opencl_manager.cpp:532-566) with the arithmetic of the CPU running mean in
 * Pathtracer::render() (pathtracer.cpp:200-217):
 *   iterations == 1: image = sample;  else image = (1/it)*sample + ((it-1)/it)*image.
 * Device pointers, n_pixels Pixels each. */
int hpt_reconstruct(hpt_ctx *ctx, float *d_image, const float *d_sample, int n_pixels, int iterations);

/* The reference's live frame loop (what main.cpp runs) with its intended semantics; all
 * per-pixel path state stays on the device between calls.
 *
 * hpt_first_bounce replaces OpenCL_Manager::initial_ray_kernel_execute (opencl_manager.hpp:33-37,
 * opencl_manager.cpp:255-357; kernel initial_ray_kernel.cl:802-902): traces the camera ray (jitter cell 0)
 * and the first shading event of every pixel of this shard, caches that state, and starts path 0 from it.
 * Resets the per-pixel path counters.
 *
 * hpt_bounce replaces OpenCL_Manager::ray_kernel_execute (opencl_manager.hpp:39-48, opencl_manager.cpp:360-525;
 * kernel ray_kernel.cl:817-929): advances every pixel's current path by ONE bounce (Russian roulette when
 * bounce > rr_start, depth limit max_depth).  A path that ends leaves its colour as the pixel's contribution,
 * the pixel's path counter goes up by one, and the pixel restarts from the cache with a fresh RNG key
 * (seed, pixel, path number).  The committed kernel never restarts a finished pixel (ray_kernel.cl:899-921);
 * that defect is not reproduced.  Returns HPT_ERR_INVALID when contributions of the previous hpt_bounce
 * have not been folded yet (they would be lost), or before hpt_first_bounce.
 *
 * hpt_reconstruct_paths replaces OpenCL_Manager::reconstruct_kernel_execute as the live loop uses it
 * (reconstruct_kernel.cl:10-37 without its gid < 1000 guard and G/B swap): for every pixel whose path ended in
 * the last hpt_bounce, d_image = (1/n)*contribution + ((n-1)/n)*d_image with n = the pixel's path counter
 * (n == 1: copy).  d_image: device pointer, width*height*3 floats.
 *
 * hpt_path_counts copies the per-pixel path counters (the reference's iteration_memory) to the host.
 *
 * Constants: these calls use the context's hpt_config like every other render call, whose DEFAULTS are the
 * mega-kernel's (diffuse factor 2*dot(n,dir), pathtracer_kernel.cl:822; depth limit 15).  The kernel they replace
 * weighs a diffuse bounce by dot(n,dir) alone and has no depth limit (ray_kernel.cl:785-786): a caller that wants
 * the live loop's own brightness sets hpt_config.diffuse_scale = 1.0 (and a large max_depth) first.  Deliberate:
 * one set of constants per context, so that the two pipelines converge to the same image. */
int hpt_first_bounce(hpt_ctx *ctx);
int hpt_bounce(hpt_ctx *ctx);
int hpt_reconstruct_paths(hpt_ctx *ctx, float *d_image);
int hpt_path_counts(hpt_ctx *ctx, int32_t *counts);

/* kernel_execute + the running mean of Pathtracer::render() in one call,
 * host image in/out (width*height*3 floats): what render() returns after
 * iteration `iteration` given the image after `iteration-1`. */
int hpt_render_sample(hpt_ctx *ctx, int iteration, float *image_rgb);

/* Batched form: iterations spp_begin .. spp_end-1 accumulated into the
 * device-resident running-mean image d_image (width*height*3 floats; read
 * when spp_begin > 0).  Bit-identical to spp_end-spp_begin calls of
 * hpt_render_sample.  Asynchronous on the context's stream. */
int hpt_render(hpt_ctx *ctx, int spp_begin, int spp_end, float *d_image);

/* Device images for callers without the HIP runtime API (plain C, another language's FFI): hpt_image_alloc gives a
 * zero-filled width*height*3 float image on the context's device for hpt_render / hpt_reconstruct_paths,
 * hpt_read_image copies one to the host in stream order and waits, hpt_image_free releases it.  The reference
 * keeps the image on the host and copies it both ways on every call (opencl_manager.cpp:576-648); SURVEY.md 8b
 * lists hpt_read_image for the device-resident form. */
int hpt_image_alloc(hpt_ctx *ctx, float **d_image);
int hpt_image_free(hpt_ctx *ctx, float *d_image);
int hpt_read_image(hpt_ctx *ctx, const float *d_image, float *image_rgb);

/* hpt_render for callers that cannot allocate (code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc).  NO counterpart in the reference: its kernel_execute creates and destroys every buffer per call
 * (opencl_manager.cpp:576-604).  Same arguments, same image bit for bit as hpt_render.  Where hpt_render keeps a
 * sample buffer and a traversal-stack overflow buffer that it allocates (and grows) on first use, this entry point
 * places both in a static device array of hpt_noalloc_scratch_bytes() = 128 MiB that every GPU the library runs on
 * holds until the process ends, and cuts the render into as many launches as that takes.  It calls no hipMalloc,
 * hipFree, hipEventCreate or any other allocating routine, and records no events (hpt_get_stats reports the ray
 * counters of the call, times read 0).  The kernel's launch shape (function attribute, occupancy at its LDS size)
 * is queried when the scene becomes resident -- by the uploads and hpt_set_scene -- so that no call here is the first to ask
 * the runtime; a caller under graph capture needs no warm-up call.  Limits: the scene must be resident (hpt_upload_bvh and hpt_upload_triangles
 * done: the uploads build the device data; after replacing only ONE of the two in a resident context, upload the other as
 * well or call hpt_render once), else HPT_ERR_INVALID; the stack overflow may take half of the array --
 * the grid is shrunk until it does, and a tree for which not even one block per CU fits is refused with
 * HPT_ERR_INVALID (no tree that hpt_upload_bvh accepts, depth <= 64, comes near that on a 256-CU device); the rest
 * holds the samples of one launch, at least one 8x8 tile x one iteration, so any frame size and sample count
 * renders, in more launches than hpt_render needs (C3 full frame: 50 launches, 10 % slower).  Every context on a
 * GPU shares that GPU's array and every call overwrites what the last one left there (nothing in it needs to
 * start at zero): two calls that overlap on one GPU (two streams, two contexts, two threads) race; calls in
 * stream order do not. */
int hpt_render_noalloc(hpt_ctx *ctx, int spp_begin, int spp_end, float *d_image);
int64_t hpt_noalloc_scratch_bytes(void);

/* Test hook for the bit-exact criterion: closest hit of the camera ray of
 * `iteration` for every pixel of the full image (hit_id encoding in types.h,
 * t = 1000 on a miss). Host arrays of width*height. */
int hpt_primary_hits(hpt_ctx *ctx, int iteration, int32_t *hit_id, float *t);

/* Multi-GPU exchange helpers (device pointers).  pack: this shard's tiles of
 * d_image -> compact slab of hpt_slab_floats() floats; unpack: `world` slabs
 * laid end to end (an all-gather result) -> full image. */
int64_t hpt_slab_floats(const hpt_ctx *ctx);
int hpt_pack_tiles(hpt_ctx *ctx, const float *d_image, float *d_slab);
int hpt_unpack_tiles(hpt_ctx *ctx, const float *d_slabs, float *d_image);

/* ---- Multi-GPU exchange inside the library (SURVEY.md §8b "multi-GPU handled inside the context", §8e) ----
 * No counterpart in the reference: it drives platform 0 / GPU device 0 only (opencl_manager.cpp:57-73).  RCCL is
 * called from C++ (librccl.so.1 is loaded on first use, so a one-GPU caller never needs it); the collectives are queued
 * on the context's stream, in order with the render, without host synchronisation.
 *
 * One process (or host thread) per GPU: rank 0 calls hpt_comm_unique_id, hands the HPT_COMM_ID_BYTES bytes to the other
 * ranks by any means (a file, MPI, a torch.distributed broadcast), and every rank calls hpt_comm_init with them; the
 * call is collective (it returns when all `world` ranks have made it).  hpt_destroy releases the communicator.
 *
 * hpt_gather_tiles: tile shards (hpt_set_shard(rank, world) must equal the communicator's rank and size).  This
 * rank's tiles of d_image -> slab, ONE ncclAllGather, every rank's slab -> d_image: afterwards every rank holds the
 * whole frame, bit-identical to a one-GPU render (each pixel's samples are folded on one GPU in iteration order).
 *
 * hpt_reduce_samples: sample-range shards (hpt_set_sample_shard(rank, world) likewise).  d_image is this rank's
 * running mean over its n_local samples; d_frame (may equal d_image when the rank's own mean is no longer needed)
 * receives sum_r (n_r / n_total) * image_r through ONE ncclAllReduce: the frame's mean over n_total samples, within
 * 1e-5 relative of the unsharded running mean (fp32 summation order).  A progressive render keeps d_image, calls
 * hpt_render for the next slice of iterations and reduces into d_frame again ("reduce every K spp", BASELINE config 5). */
#define HPT_COMM_ID_BYTES 128
int hpt_comm_unique_id(void *id_out);
int hpt_comm_init(hpt_ctx *ctx, const void *id, int rank, int world);
int hpt_comm_destroy(hpt_ctx *ctx);
int hpt_gather_tiles(hpt_ctx *ctx, float *d_image);
int hpt_reduce_samples(hpt_ctx *ctx, const float *d_image, float *d_frame, int n_local, int n_total);

/* One process, several GPUs: a group owns one context per device (each with a stream of its own, so the devices
 * run side by side), one communicator over them (ncclCommInitAll) and one device image per member.  This is the form
 * SURVEY.md §8b gives hpt_create (`n_devices`), kept apart so that hpt_create stays the one-device call the reference's
 * initialize() maps onto.  The scene calls go to every member.  hpt_group_render = every member's hpt_render of its
 * shard + the exchange, all asynchronous; afterwards every member's frame (hpt_group_frame) is the whole image.
 * Sharding: HPT_SHARD_TILES (default; bit-identical to one GPU for any group size) or HPT_SHARD_SAMPLES (member r
 * renders the iterations s % n == r; renders must then start at 0 or continue where the last one ended, and the frame
 * is the reduced mean over all samples so far).  hpt_group_read_image copies member 0's frame to the host and waits;
 * hpt_group_get_stats sums the members' ray counters and takes the longest of their times.  One host thread per group. */
typedef struct hpt_group hpt_group;
enum { HPT_SHARD_TILES = 0, HPT_SHARD_SAMPLES = 1 };
int hpt_group_create(int width, int height, const int *devices, int n_devices, hpt_group **out);
int hpt_group_destroy(hpt_group *g);
const char *hpt_group_last_error(const hpt_group *g);      /* g == NULL: last hpt_group_create error */
int hpt_group_size(const hpt_group *g);
hpt_ctx *hpt_group_ctx(hpt_group *g, int member);
float *hpt_group_frame(hpt_group *g, int member);          /* device pointer, width*height*3 floats */
int hpt_group_set_sharding(hpt_group *g, int mode);
int hpt_group_upload_triangles(hpt_group *g, const hpt_triangle *triangles, int n_triangles);
int hpt_group_upload_bvh(hpt_group *g, const hpt_bvh_bbox *bbox, const hpt_bvh_info *info, int n_nodes);
int hpt_group_set_scene(hpt_group *g, const hpt_primitive *primitives, int n_primitives,
                        const hpt_material *materials, int n_materials);
int hpt_group_set_camera(hpt_group *g, const hpt_camera *camera);
int hpt_group_set_config(hpt_group *g, const hpt_config *cfg);
int hpt_group_render(hpt_group *g, int spp_begin, int spp_end);
int hpt_group_synchronize(hpt_group *g);
int hpt_group_read_image(hpt_group *g, float *image_rgb);
int hpt_group_get_stats(hpt_group *g, hpt_stats *stats);

/* Checkpoint of a progressive render (the reference has only the 'p'-key PNG snapshot, main.cpp:194-203):
 * the running-mean image, the number of iterations folded into it, the context's config and shards, and a checksum.
 * hpt_checkpoint_save waits for the stream and writes `path` atomically (temporary file + rename);
 * hpt_checkpoint_load refuses a file of another image size, config or shard (the continuation would be another
 * render) or with a bad checksum, else fills d_image and *spp_done: hpt_render(ctx, *spp_done, n, d_image) then
 * continues bit-identically to a render that was never interrupted. */
int hpt_checkpoint_save(hpt_ctx *ctx, const float *d_image, int spp_done, const char *path);
int hpt_checkpoint_load(hpt_ctx *ctx, float *d_image, int *spp_done, const char *path);

/* 8-bit output of png_writer.cpp:7-17 / main.cpp:147-149:
 * (uint8)(255*clip(c,0,1)), truncation, no gamma. d_rgb8 = n_pixels*3 bytes. */
int hpt_tonemap(hpt_ctx *ctx, const float *d_image, uint8_t *d_rgb8, int n_pixels);

/* Blocks until the context's stream is idle. */
int hpt_synchronize(hpt_ctx *ctx);

/* Counters of the last hpt_render / hpt_kernel_execute and device time per stage (types.h hpt_stats): render_kernel,
 * fold_kernel, and the iterative calls' trace / shade / reconstruct kernels, each from HIP events around its launches
 * (the reference has only the wall clock of one display() iteration, main.cpp:124-140).  Synchronizes. */
int hpt_get_stats(hpt_ctx *ctx, hpt_stats *stats);

#ifdef __cplusplus
}
#endif
#endif /* HPT_HPT_H */
