// Micro-benchmark (development): does a 16-B-per-lane gather cost the vector L1 per ACTIVE LANE or per
// wave instruction?  Every wave issues the same number of global_load_dwordx4 instructions; only `active` of
// its 64 lanes take part (contiguous low lanes, or scattered one per group of 64/active).  Also: node-shaped
// fetches (4 loads of the 4 quarters of ONE line per lane, as render_kernel's NODE step does) against
// 4 loads of 4 different lines, and narrower loads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// shape 0: 4 loads, 4 different lines (quarter fixed per lane); shape 1: 4 loads, the 4 quarters of one line
template <int SHAPE>
__global__ void gather(const float4 *__restrict__ tab, unsigned mask, int active, int scattered, int iters, float *out) {
    unsigned lane = threadIdx.x & 63, gl = blockIdx.x * blockDim.x + threadIdx.x;
    bool on = scattered ? ((lane % (64u / (unsigned)active)) == 0u) : (lane < (unsigned)active);
    unsigned state = gl * 2654435761u + 12345u;
    float acc = 0;
    if (on) {
        for (int i = 0; i < iters; i++) {
            unsigned s0 = state * 1664525u + 1013904223u, s1 = s0 * 1664525u + 1013904223u;
            unsigned s2 = s1 * 1664525u + 1013904223u, s3 = s2 * 1664525u + 1013904223u;
            state = s3;
            float4 a, b, c, d;
            if (SHAPE == 0) {
                a = tab[(size_t)((s0 >> 8) & mask) * 4 + (gl & 3u)];
                b = tab[(size_t)((s1 >> 8) & mask) * 4 + (gl & 3u)];
                c = tab[(size_t)((s2 >> 8) & mask) * 4 + (gl & 3u)];
                d = tab[(size_t)((s3 >> 8) & mask) * 4 + (gl & 3u)];
            } else {
                const float4 *n = tab + (size_t)((s3 >> 8) & mask) * 4;
                a = n[0]; b = n[1]; c = n[2]; d = n[3];
            }
            acc += (a.x + b.y) + (c.z + d.w);
        }
    }
    out[gl] = acc;
}

// one dword / dwordx2 per lane per load, 4 different lines
template <typename T>
__global__ void gather_narrow(const T *__restrict__ tab, unsigned mask, int iters, float *out) {
    unsigned gl = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned state = gl * 2654435761u + 12345u;
    float acc = 0;
    const unsigned per_line = 64 / sizeof(T);
    for (int i = 0; i < iters; i++) {
        unsigned s0 = state * 1664525u + 1013904223u, s1 = s0 * 1664525u + 1013904223u;
        unsigned s2 = s1 * 1664525u + 1013904223u, s3 = s2 * 1664525u + 1013904223u;
        state = s3;
        T a = tab[(size_t)((s0 >> 8) & mask) * per_line + (gl & 3u)];
        T b = tab[(size_t)((s1 >> 8) & mask) * per_line + (gl & 3u)];
        T c = tab[(size_t)((s2 >> 8) & mask) * per_line + (gl & 3u)];
        T d = tab[(size_t)((s3 >> 8) & mask) * per_line + (gl & 3u)];
        acc += (*(float *)&a + *(float *)&b) + (*(float *)&c + *(float *)&d);
    }
    out[gl] = acc;
}

int main() {
    const unsigned sizes[2] = {256, 32768};      // lines of 64 B: 16 KB, 2 MB
    int blocks = 256 * 6, threads = 256, iters = 400;
    float *out; CK(hipMalloc(&out, (size_t)blocks * threads * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int si = 0; si < 2; si++) {
        unsigned n_lines = sizes[si];
        std::vector<float4> h((size_t)n_lines * 4, make_float4(1, 2, 3, 4));
        float4 *tab; CK(hipMalloc(&tab, h.size() * sizeof(float4)));
        CK(hipMemcpy(tab, h.data(), h.size() * sizeof(float4), hipMemcpyHostToDevice));
        for (int shape = 0; shape < 2; shape++)
            for (int scattered = 0; scattered < 2; scattered++)
                for (int active = 64; active >= 4; active /= 2) {
                    if (active == 64 && scattered) continue;
                    float ms = 0;
                    for (int rep = 0; rep < 2; rep++) {
                        int it = rep ? iters : 10;
                        CK(hipEventRecord(e0));
                        if (shape == 0) gather<0><<<blocks, threads>>>(tab, n_lines - 1, active, scattered, it, out);
                        else            gather<1><<<blocks, threads>>>(tab, n_lines - 1, active, scattered, it, out);
                        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                        CK(hipEventElapsedTime(&ms, e0, e1));
                    }
                    double winstr = (double)blocks * threads / 64 * iters * 4;     // wave-level load instructions
                    double clk = ms * 1e-3 * 2.4e9;
                    printf("table %5u KB shape %d %s active %2d: %8.3f ms  %.1f clk/CU per wave-instr  %.2f lane-loads/clk/CU\n",
                           n_lines / 16, shape, scattered ? "scattered " : "contiguous", active, ms,
                           clk / (winstr / 256), winstr * active / 256 / clk);
                }
        for (int wdt = 0; wdt < 2; wdt++) {
            float ms = 0;
            for (int rep = 0; rep < 2; rep++) {
                int it = rep ? iters : 10;
                CK(hipEventRecord(e0));
                if (wdt == 0) gather_narrow<float><<<blocks, threads>>>((const float *)tab, n_lines - 1, it, out);
                else          gather_narrow<float2><<<blocks, threads>>>((const float2 *)tab, n_lines - 1, it, out);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
            }
            double winstr = (double)blocks * threads / 64 * iters * 4;
            double clk = ms * 1e-3 * 2.4e9;
            printf("table %5u KB %s all lanes: %8.3f ms  %.1f clk/CU per wave-instr  %.2f lane-loads/clk/CU\n",
                   n_lines / 16, wdt ? "dwordx2" : "dword  ", ms, clk / (winstr / 256), winstr * 64 / 256 / clk);
        }
        CK(hipFree(tab));
    }
    return 0;
}
