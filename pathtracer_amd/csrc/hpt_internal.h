// hpt_internal.h — what the library's other translation units (hpt_comm.hip: the multi-GPU exchange, hpt_checkpoint.cpp)
// may know of a context.  struct hpt_ctx itself stays private to hpt_backend.hip (it names kernel-side types); these
// accessors are not part of the C-ABI (include/hpt/hpt.h) and not meant for callers.
#pragma once
#include <hip/hip_runtime_api.h>

#include "../../include/hpt/hpt.h"

struct hpt_comm;                       // per-context communicator state, owned by hpt_comm.hip

struct hpt_ctx_view {
    int device, width, height;
    hipStream_t stream;
    int rank, world;                   // tile shard (hpt_set_shard)
    int sample_rank, sample_world;     // sample-range shard (hpt_set_sample_shard)
    hpt_config cfg;
};

int hpt_internal_fail(hpt_ctx *ctx, int code, const char *fmt, ...) __attribute__((format(printf, 3, 4)));
hpt_ctx_view hpt_internal_view(const hpt_ctx *ctx);
hpt_comm *&hpt_internal_comm(hpt_ctx *ctx);
