// Micro-benchmark (development): rate at which one CU's vector L1 (TCP) serves 16-B-per-lane gathers when
//  mode 0: every lane of a wave reads a different 64-B line  (one-node-per-lane traversal: 4 such loads per node)
//  mode 1: each group of 4 adjacent lanes reads the 4 quarters of one line (one node fetched by 4 lanes)
// for an L1-resident table (16 KB) and an L2-resident one (2.8 MB, like the C3 node array).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void gather(const float4 *__restrict__ tab, unsigned mask, int mode, int iters, float *out) {
    unsigned lane = threadIdx.x & 63, gl = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned state = (mode == 0 ? gl : (gl >> 2)) * 2654435761u + 12345u;
    unsigned quarter = (mode == 0) ? (gl & 3u) : (lane & 3u);
    float acc = 0;
    for (int i = 0; i < iters; i++) {
        unsigned s0 = state * 1664525u + 1013904223u, s1 = s0 * 1664525u + 1013904223u;
        unsigned s2 = s1 * 1664525u + 1013904223u, s3 = s2 * 1664525u + 1013904223u;
        state = s3;
        float4 a = tab[(size_t)((s0 >> 8) & mask) * 4 + quarter];
        float4 b = tab[(size_t)((s1 >> 8) & mask) * 4 + quarter];
        float4 c = tab[(size_t)((s2 >> 8) & mask) * 4 + quarter];
        float4 d = tab[(size_t)((s3 >> 8) & mask) * 4 + quarter];
        acc += (a.x + b.y) + (c.z + d.w);
    }
    out[gl] = acc;
}
int main() {
    const unsigned sizes[2] = {256, 32768};      // lines of 64 B: 16 KB, 2 MB
    int blocks = 256 * 6, threads = 256, iters = 500;
    float *out; CK(hipMalloc(&out, (size_t)blocks * threads * 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int si = 0; si < 2; si++) {
        unsigned n_lines = sizes[si];
        std::vector<float4> h((size_t)n_lines * 4, make_float4(1, 2, 3, 4));
        float4 *tab; CK(hipMalloc(&tab, h.size() * sizeof(float4)));
        CK(hipMemcpy(tab, h.data(), h.size() * sizeof(float4), hipMemcpyHostToDevice));
        for (int mode = 0; mode < 2; mode++) {
            gather<<<blocks, threads>>>(tab, n_lines - 1, mode, 10, out);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(a)); gather<<<blocks, threads>>>(tab, n_lines - 1, mode, iters, out); CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            double loads = (double)blocks * threads * iters * 4;
            printf("table %7u KB mode %d: %.3f ms, %.0f G lane-loads/s = %.2f lane-loads/clk/CU (2.4 GHz) = %.3f lines(nodes)/clk/CU\n",
                   n_lines / 16, mode, ms, loads / ms / 1e6, loads / (ms * 1e-3) / 256 / 2.4e9,
                   loads / (ms * 1e-3) / 256 / 2.4e9 / 4);
        }
        CK(hipFree(tab));
    }
    return 0;
}
