// host/checkpoint.cpp — hpt_checkpoint_save / hpt_checkpoint_load (include/hpt/hpt.h): the accumulation buffer and its
// sample count on disk, so that a long progressive render (BASELINE config 5: 1024 spp at 4K) can be stopped and
// continued.  The reference has only the viewer's 'p'-key PNG snapshot (main.cpp:194-203), which cannot be continued:
// 8-bit, clipped, no sample count.
//
// File: "HPTCKPT1", then little-endian int32 width, height, spp_done, tile rank, tile world, sample rank, sample world,
// the 16 bytes of hpt_config, a 64-bit FNV-1a of the image bytes, and width*height*3 floats.
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../hpt_internal.h"

namespace {

const char kMagic[8] = {'H', 'P', 'T', 'C', 'K', 'P', 'T', '1'};

struct Header {
    char magic[8];
    int32_t width, height, spp_done, rank, world, sample_rank, sample_world;
    hpt_config cfg;
    uint64_t checksum;
};
static_assert(sizeof(Header) == 64, "checkpoint header layout: 8 + 7*4 + 16, 4 of padding (written as zero), 8");

uint64_t fnv1a(const void *data, size_t n) {
    const unsigned char *p = static_cast<const unsigned char *>(data);
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}

}  // namespace

extern "C" {

int hpt_checkpoint_save(hpt_ctx *ctx, const float *d_image, int spp_done, const char *path) {
    if (!ctx) return HPT_ERR_INVALID;
    if (!d_image || !path || spp_done < 0) return hpt_internal_fail(ctx, HPT_ERR_INVALID, "hpt_checkpoint_save: bad arguments");
    const hpt_ctx_view v = hpt_internal_view(ctx);
    std::vector<float> img((size_t)v.width * v.height * 3);
    int rc = hpt_read_image(ctx, d_image, img.data());        // stream order, waits
    if (rc) return rc;
    Header h;
    memset(&h, 0, sizeof h);
    memcpy(h.magic, kMagic, 8);
    h.width = v.width; h.height = v.height; h.spp_done = spp_done;
    h.rank = v.rank; h.world = v.world; h.sample_rank = v.sample_rank; h.sample_world = v.sample_world;
    h.cfg = v.cfg;
    h.checksum = fnv1a(img.data(), img.size() * sizeof(float));
    const std::string tmp = std::string(path) + ".tmp";
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return hpt_internal_fail(ctx, HPT_ERR_INVALID, "hpt_checkpoint_save: cannot open %s", tmp.c_str());
    bool ok = fwrite(&h, sizeof h, 1, f) == 1 && fwrite(img.data(), sizeof(float), img.size(), f) == img.size();
    ok = (fclose(f) == 0) && ok;
    if (!ok || rename(tmp.c_str(), path) != 0) { remove(tmp.c_str()); return hpt_internal_fail(ctx, HPT_ERR_INVALID, "hpt_checkpoint_save: cannot write %s", path); }
    return HPT_OK;
}

int hpt_checkpoint_load(hpt_ctx *ctx, float *d_image, int *spp_done, const char *path) {
    if (!ctx) return HPT_ERR_INVALID;
    if (!d_image || !spp_done || !path) return hpt_internal_fail(ctx, HPT_ERR_INVALID, "hpt_checkpoint_load: bad arguments");
    const hpt_ctx_view v = hpt_internal_view(ctx);
    FILE *f = fopen(path, "rb");
    if (!f) return hpt_internal_fail(ctx, HPT_ERR_INVALID, "hpt_checkpoint_load: cannot open %s", path);
    Header h;
    std::vector<float> img((size_t)v.width * v.height * 3);
    bool ok = fread(&h, sizeof h, 1, f) == 1 && memcmp(h.magic, kMagic, 8) == 0;
    if (!ok) { fclose(f); return hpt_internal_fail(ctx, HPT_ERR_INVALID, "hpt_checkpoint_load: %s is not a checkpoint", path); }
    if (h.width != v.width || h.height != v.height) {
        fclose(f);
        return hpt_internal_fail(ctx, HPT_ERR_INVALID, "hpt_checkpoint_load: checkpoint is %dx%d, the context %dx%d", h.width, h.height, v.width, v.height);
    }
    if (memcmp(&h.cfg, &v.cfg, sizeof(hpt_config)) != 0 || h.rank != v.rank || h.world != v.world || h.sample_rank != v.sample_rank ||
        h.sample_world != v.sample_world || h.spp_done < 0) {
        fclose(f);
        return hpt_internal_fail(ctx, HPT_ERR_INVALID, "hpt_checkpoint_load: config or shard of the checkpoint differ from the context's "
                                                       "(depth %d/%d, rr %d/%d, seed %u/%u, tile shard %d of %d / %d of %d, sample shard %d of %d / %d of %d)",
                                 h.cfg.max_depth, v.cfg.max_depth, h.cfg.rr_start, v.cfg.rr_start, h.cfg.seed, v.cfg.seed, h.rank, h.world,
                                 v.rank, v.world, h.sample_rank, h.sample_world, v.sample_rank, v.sample_world);
    }
    ok = fread(img.data(), sizeof(float), img.size(), f) == img.size() && fgetc(f) == EOF;
    fclose(f);
    if (!ok) return hpt_internal_fail(ctx, HPT_ERR_INVALID, "hpt_checkpoint_load: %s is truncated or too long", path);
    if (fnv1a(img.data(), img.size() * sizeof(float)) != h.checksum)
        return hpt_internal_fail(ctx, HPT_ERR_INVALID, "hpt_checkpoint_load: checksum of %s does not match", path);
    hipError_t e = hipSetDevice(v.device);
    if (e == hipSuccess) e = hipMemcpyAsync(d_image, img.data(), img.size() * sizeof(float), hipMemcpyHostToDevice, v.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(v.stream);          // img goes out of scope
    if (e != hipSuccess) return hpt_internal_fail(ctx, HPT_ERR_DEVICE, "hpt_checkpoint_load: %s", hipGetErrorString(e));
    *spp_done = h.spp_done;
    return HPT_OK;
}

}  // extern "C"
