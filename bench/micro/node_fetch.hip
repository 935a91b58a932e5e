// Micro-benchmark (development): three ways for the lanes of a wave that are in the NODE state (a random subset,
// probability `p`) to get the 64-B node each of them points at.
//   0  today's NODE step: every active lane issues four global_load_dwordx4 of its own node
//   1  quad-cooperative: load k (k = 0..3) fetches the node of quad-lane k with the quad's four lanes (one 64-B line
//      per quad), straight into LDS (global_load_lds_dwordx4); the owner reads its 64 B back with 4 ds_read_b128
//   4  in-register pairing (<= 32 active lanes): owner + idle helper fetch half a node each in two loads, 8 ds_bpermute
//   2  compacted: the active lanes' node indices are compacted through LDS, ceil(n/16) loads fetch 16 nodes each
//      (4 lanes per node) into LDS, the owner (rank r) reads node r back
// Node indices: 84 % from a 16-KB hot set, the rest uniform over the table (C3's measured L1 hit rate).
// Every variant sums the 16 dwords of the nodes it fetched; the sums must agree (checked on the host).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define LDS_AS __attribute__((address_space(3)))

__device__ __forceinline__ unsigned draw(unsigned &state) { state = state * 1664525u + 1013904223u; return state; }
__device__ __forceinline__ unsigned pick(unsigned r, unsigned n_nodes) {
    return ((r >> 4) % 100u) < 84u ? ((r >> 12) & 255u) : ((r >> 8) % n_nodes);
}
__device__ __forceinline__ float sum16(float4 a, float4 b, float4 c, float4 d) {
    return ((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w)) + ((c.x + c.y) + (c.z + c.w)) + ((d.x + d.y) + (d.z + d.w));
}

constexpr int kWaveLds = 4096 + 64 + 256;            // 64 nodes + padding for variant 1 + 64 indices

template <int V>
__global__ __launch_bounds__(256) void emu(const float4 *__restrict__ tab, unsigned n_nodes, unsigned thr, int iters, float *out) {
    extern __shared__ char lds[];
    unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6, gl = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned state = gl * 2654435761u + 12345u;
    char *wl = lds + wave * kWaveLds;
    unsigned *widx = (unsigned *)(wl + 4096 + 64);
    float acc = 0;
    for (int i = 0; i < iters; i++) {
        unsigned r = draw(state);
        bool on = (r & 0xffffu) < thr;
        unsigned node = pick(draw(state), n_nodes);
        if (V == 0) {
            if (on) {
                const float4 *n = tab + (size_t)node * 4;
                acc += sum16(n[0], n[1], n[2], n[3]);
            }
        } else if (V == 1) {
            unsigned q = lane >> 2, k = lane & 3u;
#define QUAD_ONE(kk) {                                                                                              \
                unsigned nd = __builtin_amdgcn_mov_dpp(node, (kk) * 0x55, 0xf, 0xf, true);                          \
                bool go = __builtin_amdgcn_mov_dpp((unsigned)on, (kk) * 0x55, 0xf, 0xf, true) != 0u;                \
                if (go) __builtin_amdgcn_global_load_lds(tab + (size_t)nd * 4 + k, (LDS_AS void *)(wl + (kk) * 1040), 16, 0, 0); }
            QUAD_ONE(0) QUAD_ONE(1) QUAD_ONE(2) QUAD_ONE(3)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (on) {
                const float4 *n = (const float4 *)(wl + k * 1040 + q * 64);
                acc += sum16(n[0], n[1], n[2], n[3]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (V == 4) {
            // in-register pairing: with at most 32 lanes in the NODE state every active lane gets an idle lane as a helper;
            // owner and helper fetch one half of the node each in TWO load instructions (instead of four), the owner
            // pulls the helper's half over with 8 ds_bpermute.  More than 32 active lanes: today's four loads.
            unsigned long long in = __ballot(on);
            unsigned n_in = __popcll(in);
            if (n_in > 32u) {
                if (on) {
                    const float4 *n = tab + (size_t)node * 4;
                    acc += sum16(n[0], n[1], n[2], n[3]);
                }
            } else {
                const unsigned below = __builtin_amdgcn_mbcnt_hi((unsigned)(in >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)in, 0u));
                const unsigned rank = below, irank = lane - below;               // among the active / among the idle lanes
                const int by_pos = __builtin_amdgcn_ds_permute((int)((on ? rank : n_in + irank) << 2), (int)lane);
                const int partner = __builtin_amdgcn_ds_bpermute((int)((on ? n_in + rank : irank) << 2), by_pos);
                const unsigned pnode = (unsigned)__builtin_amdgcn_ds_bpermute(partner << 2, (int)node);
                const bool helps = !on && irank < n_in;
                float4 r0 = make_float4(0, 0, 0, 0), r1 = r0;
                if (on || helps) {
                    const float4 *n = tab + (size_t)(on ? node : pnode) * 4 + (on ? 0 : 2);
                    r0 = n[0]; r1 = n[1];
                }
                float4 z, m;
                z.x = __int_as_float(__builtin_amdgcn_ds_bpermute(partner << 2, __float_as_int(r0.x)));
                z.y = __int_as_float(__builtin_amdgcn_ds_bpermute(partner << 2, __float_as_int(r0.y)));
                z.z = __int_as_float(__builtin_amdgcn_ds_bpermute(partner << 2, __float_as_int(r0.z)));
                z.w = __int_as_float(__builtin_amdgcn_ds_bpermute(partner << 2, __float_as_int(r0.w)));
                m.x = __int_as_float(__builtin_amdgcn_ds_bpermute(partner << 2, __float_as_int(r1.x)));
                m.y = __int_as_float(__builtin_amdgcn_ds_bpermute(partner << 2, __float_as_int(r1.y)));
                m.z = __int_as_float(__builtin_amdgcn_ds_bpermute(partner << 2, __float_as_int(r1.z)));
                m.w = __int_as_float(__builtin_amdgcn_ds_bpermute(partner << 2, __float_as_int(r1.w)));
                if (on) acc += sum16(r0, r1, z, m);
            }
        } else if (V == 3) {
            // variant 2 with the pieces of node c' rotated by c'>>2 inside its 64 B of LDS: 16 consecutive ranks then
            // read 16 different 16-B bank groups in every one of the four ds_read_b128
            unsigned long long in = __ballot(on);
            unsigned n_in = __popcll(in);
            unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(in >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)in, 0u));
            if (on) widx[rank] = node;
            unsigned c = lane >> 2, k = (lane + (lane >> 4)) & 3u;
            for (unsigned m = 0; m * 16u < n_in; m++) {
                if (m * 16u + c < n_in) {
                    unsigned nd = widx[m * 16u + c];
                    __builtin_amdgcn_global_load_lds(tab + (size_t)nd * 4 + k, (LDS_AS void *)(wl + m * 1024u), 16, 0, 0);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (on) {
                const char *n = wl + rank * 64u;
                unsigned s = (rank >> 2) & 3u;
                float4 a = *(const float4 *)(n + 16u * ((0u - s) & 3u)), b = *(const float4 *)(n + 16u * ((1u - s) & 3u));
                float4 cc = *(const float4 *)(n + 16u * ((2u - s) & 3u)), d = *(const float4 *)(n + 16u * ((3u - s) & 3u));
                acc += sum16(a, b, cc, d);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else {
            unsigned long long in = __ballot(on);
            unsigned n_in = __popcll(in);
            unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(in >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)in, 0u));
            if (on) widx[rank] = node;
            unsigned c = lane >> 2, k = lane & 3u;
            for (unsigned m = 0; m * 16u < n_in; m++) {               // wave-uniform trip count
                if (m * 16u + c < n_in) {
                    unsigned nd = widx[m * 16u + c];
                    __builtin_amdgcn_global_load_lds(tab + (size_t)nd * 4 + k, (LDS_AS void *)(wl + m * 1024u), 16, 0, 0);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (on) {
                const float4 *n = (const float4 *)(wl + rank * 64u);
                acc += sum16(n[0], n[1], n[2], n[3]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    out[gl] = acc;
}

int main(int argc, char **argv) {
    unsigned n_nodes = argc > 1 ? (unsigned)atoi(argv[1]) : 86921u;
    int bpc = argc > 2 ? atoi(argv[2]) : 6;
    int blocks = 256 * bpc, threads = 256, iters = 300;
    std::vector<float4> h((size_t)n_nodes * 4);
    for (size_t i = 0; i < h.size(); i++) h[i] = make_float4((float)(i % 251), (float)(i % 13), (float)(i % 7), 1.0f);
    float4 *tab; CK(hipMalloc(&tab, h.size() * sizeof(float4)));
    CK(hipMemcpy(tab, h.data(), h.size() * sizeof(float4), hipMemcpyHostToDevice));
    size_t n_out = (size_t)blocks * threads;
    float *out; CK(hipMalloc(&out, n_out * 4));
    std::vector<float> ref(n_out), got(n_out);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double ps[5] = {0.2, 0.38, 0.5, 0.75, 1.0};
    for (int pi = 0; pi < 5; pi++) {
        unsigned thr = (unsigned)(ps[pi] * 65536.0);
        for (int v = 0; v < 5; v++) {
            float ms = 0;
            for (int rep = 0; rep < 2; rep++) {
                int it = rep ? iters : 10;
                CK(hipEventRecord(e0));
                if (v == 0) emu<0><<<blocks, threads, 4 * kWaveLds>>>(tab, n_nodes, thr, it, out);
                else if (v == 1) emu<1><<<blocks, threads, 4 * kWaveLds>>>(tab, n_nodes, thr, it, out);
                else if (v == 2) emu<2><<<blocks, threads, 4 * kWaveLds>>>(tab, n_nodes, thr, it, out);
                else if (v == 3) emu<3><<<blocks, threads, 4 * kWaveLds>>>(tab, n_nodes, thr, it, out);
                else emu<4><<<blocks, threads, 4 * kWaveLds>>>(tab, n_nodes, thr, it, out);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
            }
            CK(hipMemcpy(got.data(), out, n_out * 4, hipMemcpyDeviceToHost));
            size_t bad = 0;
            if (v == 0) ref = got; else for (size_t i = 0; i < n_out; i++) bad += got[i] != ref[i];
            double trips = (double)blocks * threads / 64 * iters;               // NODE trips of a wave
            double clk = ms * 1e-3 * 2.4e9;
            printf("bpc %d nodes %u p %.2f variant %d: %8.3f ms  %6.1f clk/CU per NODE trip  %.3f nodes/clk/CU  mismatches %zu\n",
                   bpc, n_nodes, ps[pi], v, ms, clk / (trips / 256), trips * 64 * ps[pi] / 256 / clk, bad);
        }
    }
    return 0;
}
