// Micro-benchmark (development): what does the vector L1 charge for a 16-B-per-lane load when G adjacent lanes
// read consecutive 16-B pieces of the same randomly chosen block (G = 1: every lane its own line, the NODE step's
// pattern today; G = 4: a quad reads one 64-B node; G = 8/16/64: longer runs)?  Same instruction count in every
// case; `active_groups` of the 64/G groups of a wave take part (the others are masked off).
// Second part: the same quad-cooperative fetch written straight into LDS (global_load_lds_dwordx4) and read back
// by the owning lane with four ds_read_b128.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int G>
__global__ void coop(const float4 *__restrict__ tab, unsigned mask, int active_lanes, int iters, float *out) {
    unsigned lane = threadIdx.x & 63, gl = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned grp = gl / G, sub = gl % G;
    bool on = lane < (unsigned)active_lanes;
    unsigned state = grp * 2654435761u + 12345u;
    float acc = 0;
    if (on) {
        for (int i = 0; i < iters; i++) {
            unsigned s0 = state * 1664525u + 1013904223u, s1 = s0 * 1664525u + 1013904223u;
            unsigned s2 = s1 * 1664525u + 1013904223u, s3 = s2 * 1664525u + 1013904223u;
            state = s3;
            // blocks of G*16 bytes; mask counts 64-B lines, so scale to keep the table size fixed
            unsigned bm = G >= 4 ? (mask / (G / 4 ? G / 4 : 1)) : mask;
            float4 a, b, c, d;
            if (G >= 4) {
                a = tab[(size_t)((s0 >> 8) & bm) * G + sub];
                b = tab[(size_t)((s1 >> 8) & bm) * G + sub];
                c = tab[(size_t)((s2 >> 8) & bm) * G + sub];
                d = tab[(size_t)((s3 >> 8) & bm) * G + sub];
            } else {                                   // G = 1 or 2: G lanes in one 64-B line, quarter = sub
                a = tab[(size_t)((s0 >> 8) & mask) * 4 + sub];
                b = tab[(size_t)((s1 >> 8) & mask) * 4 + sub];
                c = tab[(size_t)((s2 >> 8) & mask) * 4 + sub];
                d = tab[(size_t)((s3 >> 8) & mask) * 4 + sub];
            }
            acc += ((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w)) + ((c.x + c.y) + (c.z + c.w)) + ((d.x + d.y) + (d.z + d.w));
        }
    }
    out[gl] = acc;
}

// quad-cooperative fetch of 4 nodes per quad into LDS, then every lane reads ITS node (64 B) back.
// LDS layout per wave: load k (k = 0..3) lands at wave_base + k*(1024+16) + lane*16, so lane i = 4q+k finds its
// node at wave_base + k*1040 + q*64.
__global__ void coop_lds(const float4 *__restrict__ tab, unsigned mask, int iters, float *out) {
    extern __shared__ float4 lds[];
    unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6, gl = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned state = gl * 2654435761u + 12345u;
    float acc = 0;
    unsigned wave_base = wave * (4 * 1040);                                  // bytes
    unsigned q = lane >> 2, k = lane & 3u;
    for (int i = 0; i < iters; i++) {
        state = state * 1664525u + 1013904223u;
        unsigned mine = (state >> 8) & mask;                                  // this lane's node
#define COOP_ONE(kk) { unsigned node = __builtin_amdgcn_mov_dpp(mine, (kk) * 0x55, 0xf, 0xf, true);                \
            const float4 *src = tab + (size_t)node * 4 + k;                                                        \
            __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void *)((char *)lds + wave_base + (kk) * 1040), 16, 0, 0); }
        COOP_ONE(0) COOP_ONE(1) COOP_ONE(2) COOP_ONE(3)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const float4 *n = (const float4 *)((const char *)lds + wave_base + k * 1040 + q * 64);
        float4 a = n[0], b = n[1], c = n[2], d = n[3];
        acc += ((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w)) + ((c.x + c.y) + (c.z + c.w)) + ((d.x + d.y) + (d.z + d.w));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    out[gl] = acc;
}


// Which lanes does the L1 coalesce?  Two lanes that differ in bit D of the lane id read the two halves of... the same
// 64-B line (quarters 0 and 1); D = 1: neighbours (same quad), 4: neighbouring quads, 16 / 32: far apart.
// same = 1: both read the SAME 16 bytes.
__global__ void pairdist(const float4 *__restrict__ tab, unsigned mask, unsigned D, int same, int iters, float *out) {
    unsigned gl = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned grp = gl & ~D, sub = same ? 0u : ((gl & D) ? 1u : 0u);
    unsigned state = grp * 2654435761u + 12345u;
    float acc = 0;
    for (int i = 0; i < iters; i++) {
        unsigned s0 = state * 1664525u + 1013904223u, s1 = s0 * 1664525u + 1013904223u;
        unsigned s2 = s1 * 1664525u + 1013904223u, s3 = s2 * 1664525u + 1013904223u;
        state = s3;
        float4 a = tab[(size_t)((s0 >> 8) & mask) * 4 + sub], b = tab[(size_t)((s1 >> 8) & mask) * 4 + sub];
        float4 c = tab[(size_t)((s2 >> 8) & mask) * 4 + sub], d = tab[(size_t)((s3 >> 8) & mask) * 4 + sub];
        acc += ((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w)) + ((c.x + c.y) + (c.z + c.w)) + ((d.x + d.y) + (d.z + d.w));
    }
    out[gl] = acc;
}

template <int G> static float run(const float4 *tab, unsigned mask, int active, int iters, float *out, int blocks, int threads, hipEvent_t e0, hipEvent_t e1) {
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        coop<G><<<blocks, threads>>>(tab, mask, active, rep ? iters : 10, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    return ms;
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
        const int gs[6] = {1, 2, 4, 8, 16, 64};
        for (int gi = 0; gi < 6; gi++)
            for (int active = 64; active >= 16; active /= 2) {
                int G = gs[gi];
                float ms = G == 1 ? run<1>(tab, n_lines - 1, active, iters, out, blocks, threads, e0, e1)
                         : G == 2 ? run<2>(tab, n_lines - 1, active, iters, out, blocks, threads, e0, e1)
                         : G == 4 ? run<4>(tab, n_lines - 1, active, iters, out, blocks, threads, e0, e1)
                         : G == 8 ? run<8>(tab, n_lines - 1, active, iters, out, blocks, threads, e0, e1)
                         : G == 16 ? run<16>(tab, n_lines - 1, active, iters, out, blocks, threads, e0, e1)
                         : run<64>(tab, n_lines - 1, active, iters, out, blocks, threads, e0, e1);
                double winstr = (double)blocks * threads / 64 * iters * 4;
                double clk = ms * 1e-3 * 2.4e9;
                printf("table %5u KB G %2d active %2d: %8.3f ms  %5.1f clk/CU per wave-instr  %.2f lane-loads/clk/CU  %.1f B/clk/CU\n",
                       n_lines / 16, G, active, ms, clk / (winstr / 256), winstr * active / 256 / clk, winstr * active / 256 / clk * 16);
            }
        for (int same = 0; same < 2; same++)
            for (unsigned D = 1; D <= 32; D *= 2) {
                float ms = 0;
                for (int rep = 0; rep < 2; rep++) {
                    CK(hipEventRecord(e0));
                    pairdist<<<blocks, threads>>>(tab, n_lines - 1, D, same, rep ? iters : 10, out);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    CK(hipEventElapsedTime(&ms, e0, e1));
                }
                double winstr = (double)blocks * threads / 64 * iters * 4;
                double clk = ms * 1e-3 * 2.4e9;
                printf("table %5u KB pairs of lanes %2u apart on one line (%s): %8.3f ms  %5.1f clk/CU per wave-instr  %.2f lane-loads/clk/CU\n",
                       n_lines / 16, D, same ? "same 16 B" : "adjacent 16 B", ms, clk / (winstr / 256), winstr * 64 / 256 / clk);
            }
        {
            float ms = 0;
            for (int rep = 0; rep < 2; rep++) {
                CK(hipEventRecord(e0));
                coop_lds<<<blocks, threads, 4 * 4 * 1040>>>(tab, n_lines - 1, rep ? iters : 10, out);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
            }
            double winstr = (double)blocks * threads / 64 * iters * 4;
            double clk = ms * 1e-3 * 2.4e9;
            printf("table %5u KB quad-coop via LDS all lanes: %8.3f ms  %5.1f clk/CU per wave-instr  %.2f lane-loads/clk/CU (= nodes/clk/CU x4)\n",
                   n_lines / 16, ms, clk / (winstr / 256), winstr * 64 / 256 / clk);
        }
        CK(hipFree(tab));
    }
    return 0;
}
