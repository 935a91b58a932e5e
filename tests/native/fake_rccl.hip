// fake_rccl.hip — TEST INFRASTRUCTURE, not part of the product: a stand-in for the nine RCCL entry points that
// pathtracer_amd/csrc/hpt_comm.hip binds, for communicators whose ranks all live in ONE process and may share ONE GPU.
// It lets the one-GPU box execute hpt_group's loops over SEVERAL members (shards, slab offsets, ncclGroupStart/End
// bracketing, per-member streams) -- what real RCCL refuses there ("duplicate GPU").  Selected with HPT_RCCL_LIB=<this
// library> together with HPT_GROUP_SHARED_DEVICE=1; the product never loads it otherwise.  It says nothing about RCCL's own
// data movement over xGMI, which first runs on a multi-GPU node.
//
// Semantics kept: collectives are asynchronous on the stream given for each rank; a rank's send buffer is read after the
// work already queued on ITS stream and not overwritten by later work on it before every peer has read it; in-place
// all-gather (send == recv + rank * count) and in-place all-reduce.  Calls of one collective by all ranks must sit inside
// one ncclGroupStart/End (what a single host thread has to do with real RCCL too).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <vector>

namespace {

struct World { int n; };

struct Op {
    int kind;                 // 0 all-gather, 1 all-reduce (float, sum)
    const void *send; void *recv; size_t count; hipStream_t stream; ncclComm_t comm;
};

thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;

__global__ void sum_ranks_kernel(const float *const *send, float *recv, size_t count, int n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    float s = send[0][i];
    for (int r = 1; r < n; r++) s += send[r][i];
    recv[i] = s;
}

}  // namespace

struct ncclComm { World *world; int rank; int device; };

namespace {

#define CHECK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { std::fprintf(stderr, "fake_rccl: %s: %s\n", #expr, hipGetErrorString(e_)); return ncclUnhandledCudaError; } } while (0)

// every stream waits for the work queued so far on every other stream of the collective
ncclResult_t fence_all(const std::vector<Op> &ops) {
    std::vector<hipEvent_t> ev(ops.size());
    for (size_t i = 0; i < ops.size(); i++) {
        CHECK(hipSetDevice(ops[i].comm->device));
        CHECK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
        CHECK(hipEventRecord(ev[i], ops[i].stream));
    }
    for (size_t i = 0; i < ops.size(); i++)
        for (size_t j = 0; j < ops.size(); j++)
            if (i != j) { CHECK(hipSetDevice(ops[i].comm->device)); CHECK(hipStreamWaitEvent(ops[i].stream, ev[j], 0)); }
    for (hipEvent_t e : ev) CHECK(hipEventDestroy(e));
    return ncclSuccess;
}

ncclResult_t run(std::vector<Op> &ops) {
    if (ops.empty()) return ncclSuccess;
    const int n = ops[0].comm->world->n;
    if ((int)ops.size() != n) { std::fprintf(stderr, "fake_rccl: %zu calls for a communicator of %d ranks\n", ops.size(), n); return ncclInvalidUsage; }
    std::vector<Op> by_rank(n);
    for (const Op &o : ops) {
        if (o.kind != ops[0].kind || o.count != ops[0].count || o.comm->world != ops[0].comm->world) return ncclInvalidUsage;
        by_rank[o.comm->rank] = o;
    }
    ncclResult_t r = fence_all(by_rank);
    if (r != ncclSuccess) return r;
    if (ops[0].kind == 0) {
        const size_t bytes = ops[0].count * sizeof(float);
        for (int d = 0; d < n; d++)
            for (int s = 0; s < n; s++) {
                void *dst = static_cast<char *>(by_rank[d].recv) + (size_t)s * bytes;
                if (dst == by_rank[s].send) continue;                       // in place
                CHECK(hipSetDevice(by_rank[d].comm->device));
                CHECK(hipMemcpyAsync(dst, by_rank[s].send, bytes, hipMemcpyDeviceToDevice, by_rank[d].stream));
            }
    } else {
        // sum into a temporary per destination first (in place: the sends are the receives), then copy out
        std::vector<const float *> h_send(n);
        for (int s = 0; s < n; s++) h_send[s] = static_cast<const float *>(by_rank[s].send);
        const float **d_send = nullptr; float *tmp = nullptr;
        CHECK(hipSetDevice(by_rank[0].comm->device));
        CHECK(hipMalloc((void **)&d_send, n * sizeof(float *)));
        CHECK(hipMalloc((void **)&tmp, ops[0].count * sizeof(float)));
        CHECK(hipMemcpyAsync(d_send, h_send.data(), n * sizeof(float *), hipMemcpyHostToDevice, by_rank[0].stream));
        CHECK(hipStreamSynchronize(by_rank[0].stream));                      // (h_send is a local)
        hipLaunchKernelGGL(sum_ranks_kernel, dim3((unsigned)((ops[0].count + 255) / 256)), dim3(256), 0, by_rank[0].stream, d_send, tmp, ops[0].count, n);
        CHECK(hipGetLastError());
        CHECK(hipStreamSynchronize(by_rank[0].stream));                      // test stand-in: simplicity over overlap
        for (int d = 0; d < n; d++) {
            CHECK(hipSetDevice(by_rank[d].comm->device));
            CHECK(hipMemcpyAsync(by_rank[d].recv, tmp, ops[0].count * sizeof(float), hipMemcpyDeviceToDevice, by_rank[d].stream));
        }
        for (int d = 0; d < n; d++) { CHECK(hipSetDevice(by_rank[d].comm->device)); CHECK(hipStreamSynchronize(by_rank[d].stream)); }
        CHECK(hipFree(tmp)); CHECK(hipFree(d_send));
    }
    return fence_all(by_rank);          // nobody overwrites its send buffer before every peer has read it
}

ncclResult_t submit(const Op &op) {
    if (g_depth > 0) { g_ops.push_back(op); return ncclSuccess; }
    if (op.comm->world->n != 1) { std::fprintf(stderr, "fake_rccl: a collective of several ranks outside ncclGroupStart/End\n"); return ncclInvalidUsage; }
    std::vector<Op> one{op};
    return run(one);
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) { std::memset(id, 0x5a, sizeof *id); return ncclSuccess; }

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId, int rank) {
    if (nranks != 1 || rank != 0) { std::fprintf(stderr, "fake_rccl: ncclCommInitRank supports one rank (ranks of one process: ncclCommInitAll)\n"); return ncclInvalidUsage; }
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return ncclUnhandledCudaError;
    *comm = new ncclComm{new World{1}, 0, dev};
    return ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t *comms, int ndev, const int *devlist) {
    World *w = new World{ndev};
    for (int i = 0; i < ndev; i++) comms[i] = new ncclComm{w, i, devlist ? devlist[i] : i};
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) { delete comm; return ncclSuccess; }      // (the World leaks: test processes are short)

ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t type, ncclComm_t comm, hipStream_t stream) {
    if (type != ncclFloat) return ncclInvalidArgument;
    return submit(Op{0, sendbuff, recvbuff, sendcount, stream, comm});
}

ncclResult_t ncclAllReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t type, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream) {
    if (type != ncclFloat || op != ncclSum) return ncclInvalidArgument;
    return submit(Op{1, sendbuff, recvbuff, count, stream, comm});
}

ncclResult_t ncclGroupStart() { g_depth++; return ncclSuccess; }

ncclResult_t ncclGroupEnd() {
    if (g_depth <= 0) return ncclInvalidUsage;
    if (--g_depth > 0) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(g_ops);
    return run(ops);
}

const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : r == ncclInvalidUsage ? "invalid usage (fake_rccl)" : "error (fake_rccl)"; }

}  // extern "C"
