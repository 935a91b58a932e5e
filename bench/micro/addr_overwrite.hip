// Micro-benchmark (development): what about the four loads of a 64-B node makes them slow?
// (profiles/r01_tuning.md: every build in which hipcc gave the last of render_kernel's four node loads its own offset
// register as part of the destination ran 20-25 % slower on C4.)
//
// Every lane chases node indices through a table of 64-B nodes: four global_load_dwordx4 of the four quarters
// of one node, next index = word 0 of the node (a random permutation step).  The four loads are written in
// assembly on fixed registers.  chase<0> / chase<1>: the destination of the last load apart from the offset register
// (v[56:59], offset in v40) or on top of it (v[40:43]) -- both with the `s_nop 0` that hipcc puts in front of such a load.
// chase<2>: the four loads WITHOUT that s_nop, as one uninterrupted group (in render_kernel one s_nop inside the
// group costs 23 % at C4).  Also: cache-policy bits on the loads, and visiting both halves of each 128-B line
// (FETCH_SIZE calibration).
//   hipcc -O3 --offload-arch=gfx950 bench/micro/addr_overwrite.hip -o build/addr_overwrite && build/addr_overwrite
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int OVERLAP>
__global__ void __launch_bounds__(256) chase(const uint4 *__restrict__ tab, unsigned mask, int active, int iters, int mixed, unsigned *out) {
    const unsigned lane = threadIdx.x & 63u, gl = blockIdx.x * blockDim.x + threadIdx.x;
    // mixed: odd lanes chase inside the first 128 nodes (8 KB: L1 hits), even lanes over the whole table (misses),
    // so every load instruction has hit lanes and miss lanes
    if (mixed == 1 && (lane & 1u)) mask = 127u;
    unsigned idx = (gl * 2654435761u) & mask;
    unsigned acc = 0;
    if (lane < (unsigned)active) {
        for (int i = 0; i < iters; i++) {
            const unsigned off = idx << 6;
            unsigned next, mix;
            if (OVERLAP == 2)
                asm volatile("v_mov_b32 v40, %2\n"
                             "s_nop 1\n"
                             "global_load_dwordx4 v[44:47], v40, %3 offset:16\n"
                             "global_load_dwordx4 v[48:51], v40, %3\n"
                             "global_load_dwordx4 v[52:55], v40, %3 offset:48\n"
                             "global_load_dwordx4 v[56:59], v40, %3 offset:32\n"
                             "s_waitcnt vmcnt(0)\n"
                             "v_mov_b32 %0, v48\n"
                             "v_xor_b32 %1, v45, v53\n"
                             "v_xor_b32 %1, %1, v57\n"
                             : "=&v"(next), "=&v"(mix) : "v"(off), "s"(tab)
                             : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55",
                               "v56", "v57", "v58", "v59", "memory");
            else if (OVERLAP)
                asm volatile("v_mov_b32 v40, %2\n"
                             "s_nop 1\n"
                             "global_load_dwordx4 v[44:47], v40, %3 offset:16\n"
                             "global_load_dwordx4 v[48:51], v40, %3\n"
                             "global_load_dwordx4 v[52:55], v40, %3 offset:48\n"
                             "s_nop 0\n"
                             "global_load_dwordx4 v[40:43], v40, %3 offset:32\n"
                             "s_waitcnt vmcnt(0)\n"
                             "v_mov_b32 %0, v48\n"
                             "v_xor_b32 %1, v45, v53\n"
                             "v_xor_b32 %1, %1, v41\n"
                             : "=&v"(next), "=&v"(mix) : "v"(off), "s"(tab)
                             : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55",
                               "v56", "v57", "v58", "v59", "memory");
            else
                asm volatile("v_mov_b32 v40, %2\n"
                             "s_nop 1\n"
                             "global_load_dwordx4 v[44:47], v40, %3 offset:16\n"
                             "global_load_dwordx4 v[48:51], v40, %3\n"
                             "global_load_dwordx4 v[52:55], v40, %3 offset:48\n"
                             "s_nop 0\n"
                             "global_load_dwordx4 v[56:59], v40, %3 offset:32\n"
                             "s_waitcnt vmcnt(0)\n"
                             "v_mov_b32 %0, v48\n"
                             "v_xor_b32 %1, v45, v53\n"
                             "v_xor_b32 %1, %1, v57\n"
                             : "=&v"(next), "=&v"(mix) : "v"(off), "s"(tab)
                             : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55",
                               "v56", "v57", "v58", "v59", "memory");
            idx = (mixed == 2 && (i & 1) == 0) ? (idx ^ 1u) : (next & mask);      // mixed 2: then the other half of the same 128-B line
            acc += mix;
        }
    }
    out[gl] = acc + idx;
}

// The same chase with cache-policy bits on the loads (MOD = "nt", "sc1", "sc0 sc1", "sc0 sc1 nt"): does any of them make
// an L2 miss fetch less than the whole 128-B line?
#define CHASE_MOD(NAME, MOD) \
__global__ void __launch_bounds__(256) NAME(const uint4 *__restrict__ tab, unsigned mask, int active, int iters, unsigned *out) { \
    const unsigned lane = threadIdx.x & 63u, gl = blockIdx.x * blockDim.x + threadIdx.x; \
    unsigned idx = (gl * 2654435761u) & mask, acc = 0; \
    if (lane < (unsigned)active) \
        for (int i = 0; i < iters; i++) { \
            const unsigned off = idx << 6; \
            unsigned next, mix; \
            asm volatile("v_mov_b32 v40, %2\n s_nop 1\n" \
                         "global_load_dwordx4 v[44:47], v40, %3 offset:16 " MOD "\n" \
                         "global_load_dwordx4 v[48:51], v40, %3 " MOD "\n" \
                         "global_load_dwordx4 v[52:55], v40, %3 offset:48 " MOD "\n" \
                         "global_load_dwordx4 v[56:59], v40, %3 offset:32 " MOD "\n" \
                         "s_waitcnt vmcnt(0)\n v_mov_b32 %0, v48\n v_xor_b32 %1, v45, v53\n v_xor_b32 %1, %1, v57\n" \
                         : "=&v"(next), "=&v"(mix) : "v"(off), "s"(tab) \
                         : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", \
                           "v56", "v57", "v58", "v59", "memory"); \
            idx = next & mask; acc += mix; \
        } \
    out[gl] = acc + idx; \
}
// What interrupts the group: the same chase with ONE other instruction between the third and the fourth load.
#define CHASE_GAP(NAME, GAP) \
__global__ void __launch_bounds__(256) NAME(const uint4 *__restrict__ tab, unsigned mask, int active, int iters, unsigned *out) { \
    const unsigned lane = threadIdx.x & 63u, gl = blockIdx.x * blockDim.x + threadIdx.x; \
    unsigned idx = (gl * 2654435761u) & mask, acc = 0; \
    if (lane < (unsigned)active) \
        for (int i = 0; i < iters; i++) { \
            const unsigned off = idx << 6; \
            unsigned next, mix; \
            asm volatile("v_mov_b32 v40, %2\n s_nop 1\n" \
                         "global_load_dwordx4 v[44:47], v40, %3 offset:16\n" \
                         "global_load_dwordx4 v[48:51], v40, %3\n" \
                         "global_load_dwordx4 v[52:55], v40, %3 offset:48\n" \
                         GAP "\n" \
                         "global_load_dwordx4 v[56:59], v40, %3 offset:32\n" \
                         "s_waitcnt vmcnt(0)\n v_mov_b32 %0, v48\n v_xor_b32 %1, v45, v53\n v_xor_b32 %1, %1, v57\n" \
                         : "=&v"(next), "=&v"(mix) : "v"(off), "s"(tab) \
                         : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", \
                           "v56", "v57", "v58", "v59", "s40", "s41", "memory"); \
            idx = next & mask; acc += mix; \
        } \
    out[gl] = acc + idx; \
}
CHASE_GAP(chase_gap_none, "")
CHASE_GAP(chase_gap_nop, "s_nop 0")
CHASE_GAP(chase_gap_valu, "v_mov_b32 v41, v40")
CHASE_GAP(chase_gap_salu, "s_mov_b32 s40, 0")
CHASE_GAP(chase_gap_exec, "s_mov_b64 s[40:41], exec")
CHASE_GAP(chase_gap_wait, "s_waitcnt vmcnt(3)")

CHASE_MOD(chase_nt, "nt")
CHASE_MOD(chase_sc1, "sc1")
CHASE_MOD(chase_sc01, "sc0 sc1")
CHASE_MOD(chase_sc01nt, "sc0 sc1 nt")

int main(int argc, char **argv) {
    // addr_overwrite [log2_nodes mixed active]: one configuration only (for PMC calibration runs)
    const int only_ln = argc > 3 ? atoi(argv[1]) : -1, only_mixed = argc > 3 ? atoi(argv[2]) : -1, only_active = argc > 3 ? atoi(argv[3]) : -1;
    int dev = 0;
    CK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, dev));
    const int blocks = prop.multiProcessorCount * 6, threads = 256;          // 24 waves per CU, as render_kernel
    unsigned *out;
    CK(hipMalloc(&out, (size_t)blocks * threads * sizeof(unsigned)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned log_nodes[3] = {15, 20, 23};                               // 2 MB, 64 MB, 512 MB of 64-B nodes
    for (unsigned ln : log_nodes) {
        if (only_ln >= 0 && (int)ln != only_ln) continue;
        const size_t n = (size_t)1 << ln;
        std::vector<uint4> h(n * 4);
        // word 0 of node k = a pseudo-random next node (full-period LCG step, so chains do not collapse)
        for (size_t k = 0; k < n; k++) {
            unsigned nx = (unsigned)((k * 1664525u + 1013904223u) & (n - 1));
            for (int q = 0; q < 4; q++) h[k * 4 + q] = make_uint4(q == 0 ? nx : (unsigned)k, (unsigned)k * 3u + q, 7u, 9u);
        }
        uint4 *tab;
        CK(hipMalloc(&tab, n * 64));
        CK(hipMemcpy(tab, h.data(), n * 64, hipMemcpyHostToDevice));
        for (int mixed : {0, 1, 2})
        for (int active : {64, 28}) {
            if (only_ln >= 0 && (mixed != only_mixed || active != only_active)) continue;
            const int iters = ln >= 20 ? 400 : 2000;
            float ms[3] = {0, 0, 0};
            for (int rep = 0; rep < 3; rep++)
                for (int v = 0; v < 3; v++) {
                    CK(hipEventRecord(e0));
                    if (v == 2) hipLaunchKernelGGL(chase<2>, dim3(blocks), dim3(threads), 0, 0, tab, (unsigned)(n - 1), active, iters, mixed, out);
                    else if (v) hipLaunchKernelGGL(chase<1>, dim3(blocks), dim3(threads), 0, 0, tab, (unsigned)(n - 1), active, iters, mixed, out);
                    else hipLaunchKernelGGL(chase<0>, dim3(blocks), dim3(threads), 0, 0, tab, (unsigned)(n - 1), active, iters, mixed, out);
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    CK(hipGetLastError());
                    float t; CK(hipEventElapsedTime(&t, e0, e1));
                    if (rep == 0 || t < ms[v]) ms[v] = t;
                }
            const double visits = (double)blocks * (threads / 64) * active * iters;
            printf("visits per launch %.0f (x 64 B = %.3f GB)\n", visits, visits * 64 / 1e9);
            printf("table %4zu MB, %s, %2d active lanes: s_nop in the group, separate dst %.3f ms (%.2f Gvisits/s), dst over address %.3f ms (ratio %.3f); "
                   "no s_nop %.3f ms (%.2f Gvisits/s, ratio %.3f)\n",
                   n * 64 >> 20, mixed == 2 ? "both halves of each line" : mixed ? "odd lanes in 8 KB" : "all lanes anywhere", active, ms[0], visits / ms[0] / 1e6,
                   ms[1], ms[1] / ms[0], ms[2], visits / ms[2] / 1e6, ms[2] / ms[0]);
        }
        if (only_ln < 0 || only_mixed == 4) {
            typedef void (*K)(const uint4 *, unsigned, int, int, unsigned *);
            const K ks[6] = {chase_gap_none, chase_gap_nop, chase_gap_valu, chase_gap_salu, chase_gap_exec, chase_gap_wait};
            const char *names[6] = {"nothing", "s_nop 0", "v_mov_b32", "s_mov_b32", "s_mov_b64 from exec", "s_waitcnt vmcnt(3)"};
            const int iters = ln >= 20 ? 400 : 2000;
            for (int k = 0; k < 6; k++) {
                float best = 0;
                for (int rep = 0; rep < 2; rep++) {
                    CK(hipEventRecord(e0));
                    hipLaunchKernelGGL(ks[k], dim3(blocks), dim3(threads), 0, 0, tab, (unsigned)(n - 1), 64, iters, out);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
                    float t; CK(hipEventElapsedTime(&t, e0, e1));
                    if (rep == 0 || t < best) best = t;
                }
                printf("table %4zu MB, between the third and fourth load %-20s: %.3f ms (%.2f Gvisits/s)\n", n * 64 >> 20, names[k], best,
                       (double)blocks * (threads / 64) * 64 * iters / best / 1e6);
            }
        }
        if (only_ln < 0 || only_mixed == 3) {
            typedef void (*K)(const uint4 *, unsigned, int, int, unsigned *);
            const K ks[4] = {chase_nt, chase_sc1, chase_sc01, chase_sc01nt};
            const char *names[4] = {"nt", "sc1", "sc0 sc1", "sc0 sc1 nt"};
            const int iters = ln >= 20 ? 400 : 2000;
            for (int k = 0; k < 4; k++) {
                float best = 0;
                for (int rep = 0; rep < 2; rep++) {
                    CK(hipEventRecord(e0));
                    hipLaunchKernelGGL(ks[k], dim3(blocks), dim3(threads), 0, 0, tab, (unsigned)(n - 1), 64, iters, out);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
                    float t; CK(hipEventElapsedTime(&t, e0, e1));
                    if (rep == 0 || t < best) best = t;
                }
                printf("table %4zu MB, loads with %-10s: %.3f ms (%.2f Gvisits/s)\n", n * 64 >> 20, names[k], best,
                       (double)blocks * (threads / 64) * 64 * iters / best / 1e6);
            }
        }
        CK(hipFree(tab));
    }
    return 0;
}
