// device_math.h — fp32 helpers of the HIP render path.
//
// Numerics contract: IEEE fp32 + - * / sqrt evaluated in
// the order written (the library is built with -ffp-contract=off and
// correctly-rounded divide/sqrt), dot(a,b) = (ax*bx + ay*by) + az*bz,
// normalize(v) = v * (1/sqrt(dot(v,v))), pow(x,2) = x*x, pow(x,5) = (x²·x²)·x,
// sin/cos by the polynomial below.  These replace the OpenCL built-ins
// normalize/dot/cross/pow/sin/cos used by the reference kernels
// (reference pathtracer/ray_kernel.cl:279-300, 615-663), whose precision is
// implementation-defined.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define HPT_EPSILON 0.001f            // ray_kernel.cl:4
#define HPT_ONE_THIRD_RT 0.57735026919f  // ray_kernel.cl:12

struct f3 { float x, y, z; };

__device__ __forceinline__ f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ f3 mk3(float4 v) { return mk3(v.x, v.y, v.z); }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ f3 cross3(f3 a, f3 b) {
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ f3 normalize3(f3 a) {
    float inv = 1.0f / sqrtf(dot3(a, a));
    return a * inv;
}
__device__ __forceinline__ float pow2f(float x) { return x * x; }
__device__ __forceinline__ float pow5f(float x) { float x2 = x * x; return (x2 * x2) * x; }

// sin/cos for phi in [0, 2*pi]: Cody-Waite reduction to [-pi/4, pi/4] by even
// octant, degree-7/8 minimax polynomials (<= 2 ulp).
__device__ __forceinline__ void sincos_2pi(float x, float &s, float &c) {
    int j = (int)(x * 1.27323954473516f);
    j = (j + 1) & ~1;
    float y = (float)j;
    x = ((x - y * 0.78515625f) - y * 2.4187564849853515625e-4f) - y * 3.77489497744594108e-8f;
    float z = x * x;
    float ps = -1.9515295891e-4f * z + 8.3321608736e-3f;
    ps = ps * z - 1.6666654611e-1f;
    ps = ps * z;
    ps = ps * x + x;
    float pc = 2.443315711809948e-5f * z - 1.388731625493765e-3f;
    pc = pc * z + 4.166664568298827e-2f;
    pc = pc * z;
    pc = pc * z;
    pc = pc - 0.5f * z;
    pc = pc + 1.0f;
    int q = (j >> 1) & 3;
    float ss = (q & 1) ? pc : ps;
    float cc = (q & 1) ? ps : pc;
    s = (q & 2) ? -ss : ss;
    c = (q == 1 || q == 2) ? -cc : cc;
}

// Counter-based RNG keyed (seed, pixel, sample, draw counter).  Replaces the
// per-pixel LCG of ray_kernel.cl:259-266 (unpinnable stream)
// while keeping its draw order.
__device__ __forceinline__ uint32_t fmix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; return h;
}
__device__ __forceinline__ uint32_t path_key(uint32_t seed, uint32_t pixel, uint32_t sample) {
    return fmix32(fmix32(seed + pixel * 0x9E3779B1u) + sample * 0x85EBCA77u);
}
__device__ __forceinline__ float draw(uint32_t key, uint32_t &ctr) {
    uint32_t h = fmix32(key + ctr * 0xC2B2AE3Du);
    ctr++;
    return (float)(h >> 8) * 5.9604644775390625e-8f;
}
