// Micro-benchmark (development): which address patterns does the vector L1's address unit coalesce for
// global_load_dwordx4?  Each wave-instruction's 64 lanes are split into groups of G adjacent lanes; a group
// reads G consecutive 16-B pieces starting at a random (G*16-B aligned) place of the table.
//   G = 1: every lane its own random piece (render_kernel's node fetch)      G = 4: one 64-B node per lane quad
//   G = 64: fully contiguous 1 KB per instruction                             G = 0: all lanes the same 16 B
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void gather(const float4 *__restrict__ tab, unsigned mask, int G, int iters, float *out) {
    unsigned lane = threadIdx.x & 63, gl = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned group = G ? (gl / (unsigned)G) : (gl >> 6), within = G ? (lane % (unsigned)G) : 0u, gsz = G ? (unsigned)G : 1u;
    unsigned state = group * 2654435761u + 12345u;
    float acc = 0;
    for (int i = 0; i < iters; i++) {
        unsigned s0 = state * 1664525u + 1013904223u, s1 = s0 * 1664525u + 1013904223u;
        unsigned s2 = s1 * 1664525u + 1013904223u, s3 = s2 * 1664525u + 1013904223u;
        state = s3;
        float4 a = tab[(((s0 >> 8) * gsz) & mask) + within];
        float4 b = tab[(((s1 >> 8) * gsz) & mask) + within];
        float4 c = tab[(((s2 >> 8) * gsz) & mask) + within];
        float4 d = tab[(((s3 >> 8) * gsz) & mask) + within];
        acc += (a.x + b.y) + (c.z + d.w);
    }
    out[gl] = acc;
}

int main() {
    const unsigned sizes[2] = {1024, 131072};      // pieces of 16 B: 16 KB, 2 MB
    int blocks = 256 * 6, threads = 256, iters = 400;
    float *out; CK(hipMalloc(&out, (size_t)blocks * threads * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int Gs[] = {1, 2, 4, 8, 16, 32, 64, 0};
    for (int si = 0; si < 2; si++) {
        unsigned n = sizes[si];
        std::vector<float4> h((size_t)n + 64, make_float4(1, 2, 3, 4));
        float4 *tab; CK(hipMalloc(&tab, h.size() * sizeof(float4)));
        CK(hipMemcpy(tab, h.data(), h.size() * sizeof(float4), hipMemcpyHostToDevice));
        for (int g : Gs) {
            float ms = 0;
            for (int rep = 0; rep < 2; rep++) {
                int it = rep ? iters : 10;
                CK(hipEventRecord(e0));
                gather<<<blocks, threads>>>(tab, n - 1, g, it, out);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
            }
            double winstr = (double)blocks * threads / 64 * iters * 4;
            double clk = ms * 1e-3 * 2.4e9;
            printf("table %5u KB group %2d: %8.3f ms  %6.1f clk/CU per wave-instr  %.2f lane-loads/clk/CU  %.1f B/clk/CU\n",
                   n / 64, g, ms, clk / (winstr / 256), winstr * 64 / 256 / clk, winstr * 1024 / 256 / clk);
        }
        CK(hipFree(tab));
    }
    return 0;
}
