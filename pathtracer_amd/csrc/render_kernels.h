// render_kernels.h — gfx950 device code of the render path.
//
// Replaces the reference's four OpenCL kernels:
//   pathtracer_kernel  (pathtracer_kernel.cl:855-950)  -> render_kernel (whole path, fused running mean)
//   initial_ray_kernel (initial_ray_kernel.cl:802-902) -> camera_ray() + closest_hit(), primary_hits_kernel
//   ray_kernel         (ray_kernel.cl:817-929)         -> the bounce body in trace_path()
//   reconstruct_kernel (reconstruct_kernel.cl:10-37)   -> reconstruct_kernel (arithmetic of pathtracer.cpp:200-217)
// It is not a translation: the BVH is re-laid out as 64-B two-child nodes read
// with four dwordx4 loads, each node visit tests both children, the per-lane
// traversal stack lives in LDS, and pixels are handed to lanes by a persistent
// work counter.  Results are bit-identical to the oracle's reference-order
// traversal (argument in DESIGN.md §Traversal equivalence).
#pragma once
#include <type_traits>
#include "device_math.h"
#include "../../include/hpt/types.h"

// 64-byte node: the boxes of both children of one interior node of the
// reference tree (child0 = node k+1, child1 = secondChildOffset, bvh.cpp:382-386).
// ref >= 0: index of the child's own WideNode; ref < 0: leaf, ~ref = (first triangle PAIR << 6) | (count-1).
// Per axis one 16-B piece {lo of child 0, lo of child 1, hi of child 0, hi of child 1}: the two children's
// plane distances come out of one v_pk_add_f32 + one v_pk_mul_f32 per register pair.
struct __attribute__((aligned(64))) WideNode {
    float lo0x, lo1x, hi0x, hi1x;
    float lo0y, lo1y, hi0y, hi1y;
    float lo0z, lo1z, hi0z, hi1z;
    int32_t ref0, ref1, axis, pad;
};
static_assert(sizeof(WideNode) == 64, "WideNode must be one 64-B line");

#define HPT_REF_DONE 0x7fffffff
#define HPT_REF_IDLE 0x7ffffffe     // render_kernel only: the lane has no path and none is left to fetch
#define HPT_LEAF_MAX_PRIMS 64
#define HPT_LEAF_REF(offset, n) (~(((offset) << 6) | ((n) - 1)))

struct RenderParams {
    const WideNode *nodes;
    const float4 *tris;              // hpt_triangle[] viewed as 3 x float4 (shading: normal of the hit triangle)
    const float4 *pairs;             // triangles as the traversal reads them: 80-B records of two, or 48-B records of one
                                     // (singles != 0); leaf references point here
    const hpt_primitive *prims;
    const struct PrimCompact *prim_compact;   // the fields the closest-hit query reads of each primitive, 32 B each (long lists: two-phase loop)
    const struct PrimPair *prim_pairs;        // the same for primitives 2j, 2j+1 side by side: phase 1 tests two spheres per packed instruction
    const hpt_material *mats;
    float *image;                    // width*height*3
    int32_t *hit_id;                 // primary_hits only
    float *hit_t;
    unsigned int *work_counter;
    unsigned long long *counters;    // [0] primary rays [1] bounce rays [2] paths [15] 16-B lane-loads of the traversal and shading
    hpt_camera cam;
    float root_lo[3], root_hi[3];
    int32_t root_ref;                // HPT_REF_DONE when there is no tree
    int32_t width, height;
    int32_t n_prims;
    int32_t spp_begin, spp_end;
    int32_t max_depth, rr_start;
    float diffuse_scale;
    uint32_t seed;
    int32_t rank, world;
    int32_t tiles_x, n_tiles;
    uint32_t n_work;                 // local work items = local tiles * 64
    int32_t raw_sample;              // 1: store the sample itself (kernel_execute), 0: running mean
    int32_t shade_threshold;         // lanes that must want SHADE before the wave runs it (scheduling only)
    int32_t tris_per_trip;           // triangles a TRI step may test before the wave re-votes (scheduling only)
    int32_t node_repeat_min;         // a NODE step repeats while this many lanes still want it (scheduling only; > 64 = never)
    int32_t tri_weight;              // step choice: TRI when nT * tri_weight > nN * 4 (scheduling only; 8 = a TRI lane counts double)
    int32_t stack_cap;               // traversal-stack entries kept in LDS per lane; deeper entries go to `spill`
    uint2 *spill;                    // [entry - stack_cap][grid lane], HBM; only touched by rays deeper than stack_cap
    uint32_t spill_stride;           // lanes in the grid
    float *samples;                  // [batch sample][slot][3]: one radiance sample per (pixel, iteration) of this launch
    uint32_t n_slots;                // pixel slots of this launch: a range of this shard's slots (local tiles * 64 of them)
    uint32_t slot_base;              // first slot of the range (a multiple of 64): launches can cover a frame in slot ranges, so the
                                     // sample buffer bounds the work per launch, not the frame size
    int32_t batch;                   // samples per pixel in this launch: the rank's samples spp_base .. spp_base + batch - 1
    int32_t spp_base;                // (fold_kernel's running-mean count), which are the frame's iterations
    int32_t spp_first, spp_stride;   // spp_first + k * spp_stride (sample shards: stride = ranks; otherwise 1)
    int32_t fetch_chunk;             // work items a wave reserves per atomic, <= 4096 (scheduling only)
    uint32_t tail_items;             // the last tail_items of a launch are reserved 64 at a time (scheduling only)
    int32_t singles;                 // record form of this tree (chosen per tree by the host, see HPT_SINGLE_*)
    int32_t fetch_threshold;         // iter_trace_kernel: lanes that must want a new ray before the wave fetches (scheduling only)
    uint32_t n_lds_nodes;            // render_kernel: nodes 0 .. n_lds_nodes-1 (the top of the tree, breadth first: the host numbers them
                                     // first) are copied to LDS by every block, behind its stacks, and read from there; 0 = none
};

struct Ray { f3 o, d; };

// What intersect() reads of a primitive (ray_kernel.cl:524-602): centre, radius, type, plane normal.  Built by
// hpt_set_scene beside the wire-format array; 32 B = one s_load_dwordx8 (wave-uniform index) or two 16-B lane-loads.
struct __attribute__((aligned(32))) PrimCompact { float cx, cy, cz, radius, type, nx, ny, nz; };
// Primitives 2j and 2j+1 with each field as a register pair (phase 1 of prims_two_phase on packed fp32).  rr = radius *
// radius, rounded to fp32 on the host exactly as the device rounds it (one IEEE multiply).  An odd list ends in a
// primitive of type -1 (never hit, never a candidate).
struct __attribute__((aligned(64))) PrimPair { float cx[2], cy[2], cz[2], rr[2], type[2], nx[2], ny[2], nz[2]; };
#define HPT_TWO_PHASE_MIN_PRIMS 12   // shorter lists walk the wire-format array directly (one pass, no candidate masks)

// ---- camera ray: initial_ray_kernel.cl:859-883 (== pathtracer_kernel.cl:912-937) ----
__device__ __forceinline__ Ray camera_ray(const RenderParams &p, int gid, int iteration, uint32_t key, uint32_t &ctr) {
    const int n = 6;
    float x = (float)(gid % p.width);            // fmod((float)gid,(float)width): exact for gid < 2^24
    float y = (float)gid / (float)p.width;       // not floored
    x = (x / (float)p.width) - 0.5f;
    y = (y / (float)p.height) - 0.5f;
    float left_limit = x - 1.0f / (float)(p.width + p.width);
    float top_limit = y - 1.0f / (float)(p.height + p.height);
    int cell = iteration % (n * n);
    float inv_n_w = 1.0f / (float)(n * p.width);
    float inv_n_h = 1.0f / (float)(n * p.height);
    float r0 = draw(key, ctr);
    float r1 = draw(key, ctr);
    x = left_limit + ((float)(cell % n) + r0) * inv_n_w;
    y = top_limit + ((float)(cell / n) + r1) * inv_n_h;
    Ray r;
    r.o = mk3(p.cam.eye.x, p.cam.eye.y, p.cam.eye.z);
    f3 u = mk3(p.cam.u.x, p.cam.u.y, p.cam.u.z), v = mk3(p.cam.v.x, p.cam.v.y, p.cam.v.z);
    f3 nn = mk3(p.cam.n.x, p.cam.n.y, p.cam.n.z);
    r.d = normalize3((u * x + v * y) + (-nn));
    return r;
}

// ---- slab test of ray_kernel.cl:321-377 split into its t-independent part
// (returned) and tmin (compared with the current t by the caller) ----
__device__ __forceinline__ bool slab(f3 lo, f3 hi, f3 o, f3 inv, bool nx, bool ny, bool nz, float &tmin_out) {
    float tmin = ((nx ? hi.x : lo.x) - o.x) * inv.x;
    float tmax = ((nx ? lo.x : hi.x) - o.x) * inv.x;
    float tymin = ((ny ? hi.y : lo.y) - o.y) * inv.y;
    float tymax = ((ny ? lo.y : hi.y) - o.y) * inv.y;
    bool rej = (tmin > tymax) || (tymin > tmax);
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    float tzmin = ((nz ? hi.z : lo.z) - o.z) * inv.z;
    float tzmax = ((nz ? lo.z : hi.z) - o.z) * inv.z;
    rej = rej || (tmin > tzmax) || (tzmin > tmax);
    if (tzmin > tmin) tmin = tzmin;
    if (tzmax < tmax) tmax = tzmax;
    tmin_out = tmin;
    return !rej && (tmax > HPT_EPSILON);
}

// The same test with min3/max3 (v_max3_f32 / v_min3_f32) instead of compare+select chains.
// Equivalence with the form above, NaNs included (DESIGN.md §Slab test):
//  * a NaN on the x axis stays in tmin or tmax to the end in the reference (every later comparison with it is
//    false), so the box is neither hit nor pushed: `nanx` forces that outcome;
//  * a NaN on y or z is skipped by the reference (`tymin > tmin`, `tymax < tmax` are false, as are the reject
//    comparisons it appears in) and skipped by maxNum/minNum, so tmin and tmax come out the same;
//  * with the x values finite, "some near_i > far_j, i != j" (the reference's four reject comparisons) is
//    "max of nears > min of fars", because near_i <= far_i on each axis for a box with lo <= hi
//    (hpt_upload_bvh rejects others): fp32 subtract and multiply are monotone.
// Only the sign of a zero tmin can differ (maxNum(-0,+0) = +0), which no comparison with t > 0 sees.
__device__ __forceinline__ bool slab_minmax(f3 lo, f3 hi, f3 o, f3 inv, bool nx, bool ny, bool nz, float &tmin_out) {
    float txn = ((nx ? hi.x : lo.x) - o.x) * inv.x, txf = ((nx ? lo.x : hi.x) - o.x) * inv.x;
    float tyn = ((ny ? hi.y : lo.y) - o.y) * inv.y, tyf = ((ny ? lo.y : hi.y) - o.y) * inv.y;
    float tzn = ((nz ? hi.z : lo.z) - o.z) * inv.z, tzf = ((nz ? lo.z : hi.z) - o.z) * inv.z;
    bool nanx = __builtin_isunordered(txn, txf);
    float tmin = __builtin_fmaxf(__builtin_fmaxf(txn, tyn), tzn);
    float tmax = __builtin_fminf(__builtin_fminf(txf, tyf), tzf);
    tmin_out = tmin;
    return !nanx && !(tmin > tmax) && (tmax > HPT_EPSILON);
}
#ifdef HPT_SLAB_REFFORM
#define HPT_SLAB slab
#else
#define HPT_SLAB slab_minmax
#endif

// lane mask of a predicate without the int round trip of __ballot() (v_cndmask + v_cmp per call)
__device__ __forceinline__ unsigned long long hpt_ballot(bool pred) { return __builtin_amdgcn_ballot_w64(pred); }

// Both children of a WideNode at once.  X, Y, Z = {lo child0, lo child1, hi child0, hi child1} per axis, so
// (lo - o) * inv and (hi - o) * inv are one packed subtract and one packed multiply per register pair
// (v_pk_add_f32 / v_pk_mul_f32: per component the same IEEE operations as the scalar form, no contraction).
// The ray's origin and reciprocal direction travel as three register pairs {o.x,o.y} {o.z,inv.x} {inv.y,inv.z}
// whose halves the packed instructions broadcast with op_sel.  Per child the arithmetic and the comparisons
// are those of slab_minmax; the result is the lane mask of "box hit and tmin < t" per child, built from one
// v_cmp per comparison and scalar ANDs.
typedef float hpt_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void slab2_minmax(float4 X, float4 Y, float4 Z, hpt_v2f oxy, hpt_v2f ozix, hpt_v2f iyz,
                                             bool nx, bool ny, bool nz, float t,
                                             unsigned long long &M0, unsigned long long &M1, float &tmin0, float &tmin1) {
#ifdef HPT_NO_PK_ASM
    const hpt_v2f tlx = (hpt_v2f{X.x, X.y} - oxy.xx) * ozix.yy, thx = (hpt_v2f{X.z, X.w} - oxy.xx) * ozix.yy;
    const hpt_v2f tly = (hpt_v2f{Y.x, Y.y} - oxy.yy) * iyz.xx, thy = (hpt_v2f{Y.z, Y.w} - oxy.yy) * iyz.xx;
    const hpt_v2f tlz = (hpt_v2f{Z.x, Z.y} - ozix.xx) * iyz.yy, thz = (hpt_v2f{Z.z, Z.w} - ozix.xx) * iyz.yy;
#else
    // hipcc materialises each broadcast operand as its own register pair (12 VGPRs, which spill); written out,
    // op_sel broadcasts the half of the ray pair in place.  a - b is a + (-b) (neg_lo/neg_hi on src1), exactly.
    // (hipcc puts its own wait state after the block before the first use of a product.)
    hpt_v2f tlx = {X.x, X.y}, thx = {X.z, X.w}, tly = {Y.x, Y.y}, thy = {Y.z, Y.w}, tlz = {Z.x, Z.y}, thz = {Z.z, Z.w};
    asm("v_pk_add_f32 %0, %0, %6 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %1, %1, %6 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %2, %2, %6 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %3, %3, %6 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %4, %4, %7 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %5, %5, %7 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_mul_f32 %0, %0, %7 op_sel:[0,1] op_sel_hi:[1,1]\n\t"
        "v_pk_mul_f32 %1, %1, %7 op_sel:[0,1] op_sel_hi:[1,1]\n\t"
        "v_pk_mul_f32 %2, %2, %8 op_sel:[0,0] op_sel_hi:[1,0]\n\t"
        "v_pk_mul_f32 %3, %3, %8 op_sel:[0,0] op_sel_hi:[1,0]\n\t"
        "v_pk_mul_f32 %4, %4, %8 op_sel:[0,1] op_sel_hi:[1,1]\n\t"
        "v_pk_mul_f32 %5, %5, %8 op_sel:[0,1] op_sel_hi:[1,1]"
        : "+v"(tlx), "+v"(thx), "+v"(tly), "+v"(thy), "+v"(tlz), "+v"(thz)
        : "v"(oxy), "v"(ozix), "v"(iyz));
#endif
    const hpt_v2f txn = nx ? thx : tlx, txf = nx ? tlx : thx;
    const hpt_v2f tyn = ny ? thy : tly, tyf = ny ? tly : thy;
    const hpt_v2f tzn = nz ? thz : tlz, tzf = nz ? tlz : thz;
    tmin0 = __builtin_fmaxf(__builtin_fmaxf(txn.x, tyn.x), tzn.x);
    tmin1 = __builtin_fmaxf(__builtin_fmaxf(txn.y, tyn.y), tzn.y);
    const float tmax0 = __builtin_fminf(__builtin_fminf(txf.x, tyf.x), tzf.x);
    const float tmax1 = __builtin_fminf(__builtin_fminf(txf.y, tyf.y), tzf.y);
    M0 = hpt_ballot(!__builtin_isunordered(txn.x, txf.x)) & hpt_ballot(!(tmin0 > tmax0)) &
         hpt_ballot(tmax0 > HPT_EPSILON) & hpt_ballot(tmin0 < t);
    M1 = hpt_ballot(!__builtin_isunordered(txn.y, txf.y)) & hpt_ballot(!(tmin1 > tmax1)) &
         hpt_ballot(tmax1 > HPT_EPSILON) & hpt_ballot(tmin1 < t);
}
__device__ __forceinline__ void slab2_refform(float4 X, float4 Y, float4 Z, hpt_v2f oxy, hpt_v2f ozix, hpt_v2f iyz,
                                              bool nx, bool ny, bool nz, float t,
                                              unsigned long long &M0, unsigned long long &M1, float &tmin0, float &tmin1) {
    const f3 o = mk3(oxy.x, oxy.y, ozix.x), inv = mk3(ozix.y, iyz.x, iyz.y);
    const bool ok0 = slab(mk3(X.x, Y.x, Z.x), mk3(X.z, Y.z, Z.z), o, inv, nx, ny, nz, tmin0);
    const bool ok1 = slab(mk3(X.y, Y.y, Z.y), mk3(X.w, Y.w, Z.w), o, inv, nx, ny, nz, tmin1);
    M0 = hpt_ballot(ok0 && tmin0 < t);
    M1 = hpt_ballot(ok1 && tmin1 < t);
}
#ifdef HPT_SLAB_REFFORM
#define HPT_SLAB2 slab2_refform
#else
#define HPT_SLAB2 slab2_minmax
#endif

// ---- Moller-Trumbore, ray_kernel.cl:467-520 ----
// Straight-line form: every quantity is computed as the reference computes it when it gets that
// far, and the reference's early returns become one conjunction (a value the reference never
// reaches cannot change the outcome, and NaN comparisons keep their reference polarity).
__device__ __forceinline__ bool intersect_tri(f3 v1, f3 edge1, f3 edge2, f3 o, f3 d, float &t) {
    f3 pvec = cross3(d, edge2);
    float det = dot3(edge1, pvec);
    bool ok = !(fabsf(det) < HPT_EPSILON);
    float invDet = 1.0f / det;
    f3 tvec = o - v1;
    float u = dot3(tvec, pvec) * invDet;
    ok = ok && !(u < 0.0f || u > 1.0f);
    f3 qvec = cross3(tvec, edge1);
    float v = dot3(d, qvec) * invDet;
    ok = ok && !(v < 0.0f || u + v > 1.0f);
    float tt = dot3(edge2, qvec) * invDet;
    ok = ok && (tt < t && tt > 0.0f);
    if (ok) t = tt;
    return ok;
}

// Triangles as the traversal reads them: two consecutive triangles of one leaf per 80-B record, each as
// {v1, e1, e2} plus its index in the uploaded (leaf-order) array.  Five 16-B loads serve two triangle tests
// (2.5 lane-loads per triangle instead of 3); a leaf with an odd count ends in a record whose second half is
// never read.  Float slots: a = 0..8, id_a = 9, id_b = 10, b = 11..19.
#define HPT_PAIR_FLOAT4S 5
#define HPT_PAIR_REF(first_pair, n) (~(((first_pair) << 6) | ((n) - 1)))
// The other record form, for trees whose leaves hold (almost) one triangle each (C4: 10 M triangles, 1.02 per
// leaf): one triangle per 48-B record, the first half of a pair record on its own (a = 0..8, id = 9).  No
// half-empty records (480 MB instead of 800 MB at C4) and a TRI step without the second-half test; the host
// picks the form per tree and the traversal kernels exist in both.
#define HPT_SINGLE_FLOAT4S 3
__device__ __forceinline__ bool intersect_pair_half(const float4 *__restrict__ rec, int half, f3 o, f3 d, float &t, int &id) {
    const float *f = reinterpret_cast<const float *>(rec) + (half ? 11 : 0);
    id = __float_as_int(reinterpret_cast<const float *>(rec)[9 + half]);
    return intersect_tri(mk3(f[0], f[1], f[2]), mk3(f[3], f[4], f[5]), mk3(f[6], f[7], f[8]), o, d, t);
}

// Keeps a load's address register(s) alive until its data has arrived, so the register allocator cannot hand them
// to the load as its destination.  When hipcc does that to the last of the NODE step's four loads it must put an
// s_nop in front of it (the earlier loads still read the register), and that ONE s_nop inside the group of four
// loads of one line costs 23 % on C4 (tree streamed from HBM) and 1 % on C3: the loads no longer reach the L1
// back to back (profiles/r01_tuning.md; bench/micro/addr_overwrite.hip reproduces it;
// tests/test_host.py::test_node_loads_form_one_uninterrupted_group checks the built code object).
#ifdef HPT_NO_KEEP_ADDRESS           // development: the unguarded code, to reproduce the loss
#define HPT_KEEP_ADDRESS(addr, loaded) ((void)0)
#else
#define HPT_KEEP_ADDRESS(addr, loaded) asm volatile("" : : "v"(addr), "v"(loaded))
#endif

// ---- sphere / plane, ray_kernel.cl:524-602 ----
// The primitive index is wave-uniform wherever this is called (every lane walks the whole list), and the list
// is read-only for the kernel's lifetime: read through the constant address space it becomes scalar loads
// (s_load_dwordx4 into SGPRs, scalar branch on the type) instead of 64 identical vector loads.
typedef const __attribute__((address_space(4))) hpt_primitive *hpt_prim_cptr;
__device__ __forceinline__ bool intersect_prim(const hpt_primitive *prim, f3 o, f3 d, float &t) {
    const hpt_prim_cptr pr = (hpt_prim_cptr)prim;
    float type = pr->type.x;
    if (type == 0.0f) {
        f3 oc = o - mk3(pr->center.x, pr->center.y, pr->center.z);
        float radius = pr->radius;
        float a = dot3(d, d);
        float b = 2.0f * dot3(d, oc);
        float c = dot3(oc, oc) - (radius * radius);
        float disc = (b * b) - ((4.0f * a) * c);
        float tt;
        if (disc < 0.0f) return false;
        else if (disc < HPT_EPSILON) tt = -b / (2.0f * a);
        else {
            disc = sqrtf(disc);
            float t0 = (-b - disc) / (2.0f * a);
            if (t0 > HPT_EPSILON) tt = t0;
            else tt = (-b + disc) / (2.0f * a);
        }
        if (tt < t && tt > 0.0f) { t = tt; return true; }
    } else if (type == 1.0f) {
        f3 pn = mk3(pr->plane_normal.x, pr->plane_normal.y, pr->plane_normal.z);
        float denom = dot3(pn, d);
        if (fabsf(denom) > HPT_EPSILON) {
            f3 between = mk3(pr->center.x, pr->center.y, pr->center.z) - o;
            float tt = dot3(between, pn) / denom;
            if (tt < t && tt >= 0.0f) { t = tt; return true; }
        }
    }
    return false;
}

// The same test on values held by the lane (the second phase below): the arithmetic of intersect_prim, operand for operand.
__device__ __forceinline__ bool intersect_prim_values(float type, f3 center, float radius, f3 pn, f3 o, f3 d, float &t) {
    if (type == 0.0f) {
        f3 oc = o - center;
        float a = dot3(d, d);
        float b = 2.0f * dot3(d, oc);
        float c = dot3(oc, oc) - (radius * radius);
        float disc = (b * b) - ((4.0f * a) * c);
        float tt;
        if (disc < 0.0f) return false;
        else if (disc < HPT_EPSILON) tt = -b / (2.0f * a);
        else {
            disc = sqrtf(disc);
            float t0 = (-b - disc) / (2.0f * a);
            if (t0 > HPT_EPSILON) tt = t0;
            else tt = (-b + disc) / (2.0f * a);
        }
        if (tt < t && tt > 0.0f) { t = tt; return true; }
    } else if (type == 1.0f) {
        float denom = dot3(pn, d);
        if (fabsf(denom) > HPT_EPSILON) {
            f3 between = center - o;
            float tt = dot3(between, pn) / denom;
            if (tt < t && tt >= 0.0f) { t = tt; return true; }
        }
    }
    return false;
}

// The primitives loop of pathtrace() (ray_kernel.cl:726-739) for long lists, in two phases per group of 64
// primitives.  The reference walks the list per ray; here 64 lanes walk it together with the primitive index
// wave-uniform, so any step a single lane needs is executed by the whole wave -- and the costly steps (square root
// and two correctly rounded divisions of a sphere that the ray's line meets, the plane's division) are needed by few lanes per
// primitive but by SOME lane for a third of them.  Phase 1 computes, for every primitive of the group, only what
// decides whether the reference gets past its first return: the sphere's discriminant (same operations, same order:
// the same sign) or the plane's |denom| > EPSILON, and records it as one bit per lane and primitive.  Phase 2: each
// lane walks ITS candidates in index order through the full test, data fetched per lane from the compact table.
// A primitive that is no candidate cannot change t (the reference returns false for it before touching t), and
// candidates are tested in index order with the running t, so (hit, t) are those of the sequential loop.
// Phase 1 for two spheres at once: the discriminants of primitives 2j (low halves) and 2j+1 (high halves).  Per half
// the operations and their order are those of the scalar form -- oc = o - c; b = 2 * ((dx*ocx + dy*ocy) + dz*ocz);
// c = ((ocx*ocx + ocy*ocy) + ocz*ocz) - r*r; disc = b*b - (4a)*c -- as v_pk_add_f32 / v_pk_mul_f32 (IEEE per component,
// no contraction); the ray's components are broadcast from register pairs with op_sel, the spheres' come as SGPR pairs.
__device__ __forceinline__ hpt_v2f sphere_pair_disc(hpt_v2f oxy, hpt_v2f ozdx, hpt_v2f dyz, hpt_v2f a4,
                                                    hpt_v2f cx, hpt_v2f cy, hpt_v2f cz, hpt_v2f rr) {
    hpt_v2f ocx, ocy, ocz, u, v, w;
    asm("v_pk_add_f32 %0, %6, %10 op_sel:[0,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"     // ocx = o.x - cx
        "v_pk_add_f32 %1, %6, %11 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"     // ocy = o.y - cy
        "v_pk_add_f32 %2, %7, %12 op_sel:[0,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"     // ocz = o.z - cz
        "v_pk_mul_f32 %3, %7, %0 op_sel:[1,0] op_sel_hi:[1,1]\n\t"                               // u = d.x * ocx
        "v_pk_mul_f32 %4, %8, %1 op_sel:[0,0] op_sel_hi:[0,1]\n\t"                               // v = d.y * ocy
        "v_pk_mul_f32 %5, %0, %0\n\t"                                                           // w = ocx * ocx
        "v_pk_mul_f32 %0, %1, %1\n\t"                                                           // ocx <- ocy * ocy
        "v_pk_add_f32 %3, %3, %4\n\t"                                                           // u = dx*ocx + dy*ocy
        "v_pk_mul_f32 %4, %8, %2 op_sel:[1,0] op_sel_hi:[1,1]\n\t"                               // v = d.z * ocz
        "v_pk_add_f32 %5, %5, %0\n\t"                                                           // w = ocx^2 + ocy^2
        "v_pk_mul_f32 %0, %2, %2\n\t"                                                           // ocx <- ocz * ocz
        "v_pk_add_f32 %3, %3, %4\n\t"                                                           // u = dot(d, oc)
        "v_pk_add_f32 %5, %5, %0\n\t"                                                           // w = dot(oc, oc)
        "v_pk_add_f32 %3, %3, %3\n\t"                                                           // b = 2 * dot(d, oc)
        "v_pk_add_f32 %5, %5, %13 neg_lo:[0,1] neg_hi:[0,1]\n\t"                                 // c = dot(oc, oc) - r*r
        "v_pk_mul_f32 %3, %3, %3\n\t"                                                           // b * b
        "v_pk_mul_f32 %5, %9, %5\n\t"                                                           // (4a) * c
        "v_pk_add_f32 %3, %3, %5 neg_lo:[0,1] neg_hi:[0,1]"                                      // disc
        : "=&v"(ocx), "=&v"(ocy), "=&v"(ocz), "=&v"(u), "=&v"(v), "=&v"(w)
        : "v"(oxy), "v"(ozdx), "v"(dyz), "v"(a4), "s"(cx), "s"(cy), "s"(cz), "s"(rr));
    return u;
}

__device__ __forceinline__ bool prim_candidate(float type, float cx, float cy, float cz, float rr, float nx, float ny, float nz,
                                               f3 o, f3 d, float a) {
    if (type == 0.0f) {
        const f3 oc = o - mk3(cx, cy, cz);
        const float b = 2.0f * dot3(d, oc);
        const float c = dot3(oc, oc) - rr;
        const float disc = (b * b) - ((4.0f * a) * c);
        return !(disc < 0.0f);
    }
    if (type == 1.0f) return fabsf(dot3(mk3(nx, ny, nz), d)) > HPT_EPSILON;
    return false;
}

__device__ __forceinline__ void prims_two_phase(const RenderParams &p, f3 o, f3 d, float &t, int &hit) {
    typedef const __attribute__((address_space(4))) PrimPair *scalar_ptr;
    const float a = dot3(d, d);
    const float a4s = 4.0f * a;
    const hpt_v2f oxy = hpt_v2f{o.x, o.y}, ozdx = hpt_v2f{o.z, d.x}, dyz = hpt_v2f{d.y, d.z}, a4 = hpt_v2f{a4s, a4s};
    for (int base = 0; base < p.n_prims; base += 64) {
        const int n = (p.n_prims - base) < 64 ? (p.n_prims - base) : 64;
        unsigned m0 = 0u, m1 = 0u;                       // candidate bits of primitives base + 0..31, base + 32..63
        for (int k = 0; k < n; k += 2) {                 // primitives base + k, base + k + 1 (the pair (base + k) / 2)
            const scalar_ptr q = (scalar_ptr)(p.prim_pairs + ((base + k) >> 1));
            const float ty0 = q->type[0], ty1 = q->type[1];
            bool c0, c1;
            if (ty0 == 0.0f && ty1 == 0.0f) {            // (wave-uniform) two spheres: packed
                const hpt_v2f disc = sphere_pair_disc(oxy, ozdx, dyz, a4, hpt_v2f{q->cx[0], q->cx[1]}, hpt_v2f{q->cy[0], q->cy[1]},
                                                      hpt_v2f{q->cz[0], q->cz[1]}, hpt_v2f{q->rr[0], q->rr[1]});
                c0 = !(disc.x < 0.0f); c1 = !(disc.y < 0.0f);
            } else {
                c0 = prim_candidate(ty0, q->cx[0], q->cy[0], q->cz[0], q->rr[0], q->nx[0], q->ny[0], q->nz[0], o, d, a);
                c1 = prim_candidate(ty1, q->cx[1], q->cy[1], q->cz[1], q->rr[1], q->nx[1], q->ny[1], q->nz[1], o, d, a);
            }
            const unsigned bits = (c0 ? 1u : 0u) | (c1 ? 2u : 0u);
            if (k < 32) m0 |= bits << (k & 31); else m1 |= bits << (k & 31);
        }
        while (hpt_ballot((m0 | m1) != 0u) != 0ull) {
            if ((m0 | m1) != 0u) {
                int k;
                if (m0 != 0u) { k = __builtin_ctz(m0); m0 &= m0 - 1u; }
                else { k = 32 + __builtin_ctz(m1); m1 &= m1 - 1u; }
                const float4 *g = reinterpret_cast<const float4 *>(p.prim_compact + base + k);
                const float4 A = g[0], B = g[1];
                HPT_KEEP_ADDRESS(g, B.w);
                if (intersect_prim_values(B.x, mk3(A.x, A.y, A.z), A.w, mk3(B.y, B.z, B.w), o, d, t)) hit = HPT_HIT_PRIMITIVE(base + k);
            }
        }
    }
}

// ---- BVH traversal: same visits, same order, same t at every test as
// intersect_tri_bvh (ray_kernel.cl:382-464), on the two-child node layout.
// stack: LDS, entry e of this lane at stack[e * stride + lane]; .x = ref, .y = bits of the child's tmin.
__device__ __forceinline__ int traverse(const RenderParams &p, f3 o, f3 d, float &t,
                                        uint2 *__restrict__ stack, int stride) {
    int hit = -1;
    int cur = p.root_ref;
    if (cur == HPT_REF_DONE) return hit;
    f3 inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    bool nx = inv.x < 0.0f, ny = inv.y < 0.0f, nz = inv.z < 0.0f;
    {   // root: the node's own box, as the reference tests it
        float tm;
        bool ok = slab(mk3(p.root_lo[0], p.root_lo[1], p.root_lo[2]), mk3(p.root_hi[0], p.root_hi[1], p.root_hi[2]),
                       o, inv, nx, ny, nz, tm);
        if (!(ok && tm < t)) return hit;
    }
    int sp = 0;
    while (true) {
        bool pop = false;
        if (cur >= 0) {
            const float4 *q = reinterpret_cast<const float4 *>(p.nodes + cur);
            float4 a = q[0], b = q[1], c = q[2];
            int4 m = *reinterpret_cast<const int4 *>(q + 3);
            float t0, t1;
            bool ok0 = slab(mk3(a.x, b.x, c.x), mk3(a.z, b.z, c.z), o, inv, nx, ny, nz, t0);
            bool ok1 = slab(mk3(a.y, b.y, c.y), mk3(a.w, b.w, c.w), o, inv, nx, ny, nz, t1);
            bool h0 = ok0 && (t0 < t), h1 = ok1 && (t1 < t);
            // near child first: dirIsNeg[axis] -> second child first (ray_kernel.cl:440-451)
            bool second_first = (m.z == 0) ? nx : ((m.z == 1) ? ny : nz);
            int refN = second_first ? m.y : m.x, refF = second_first ? m.x : m.y;
            bool hN = second_first ? h1 : h0, hF = second_first ? h0 : h1;
            float tF = second_first ? t0 : t1;
            if (hN) {
                cur = refN;
                if (hF) { stack[sp * stride] = make_uint2((unsigned)refF, __float_as_uint(tF)); sp++; }
            } else if (hF) cur = refF;
            else pop = true;
        } else {
            int packed = ~cur;
            int first = packed >> 6, n = (packed & 63) + 1;     // first PAIR of the leaf, triangles in it
            for (int i = 0; i < n; i++) {
                int id;
                const float4 *rec = p.singles ? p.pairs + (size_t)HPT_SINGLE_FLOAT4S * (first + i)
                                              : p.pairs + (size_t)HPT_PAIR_FLOAT4S * (first + (i >> 1));
                if (intersect_pair_half(rec, p.singles ? 0 : (i & 1), o, d, t, id)) hit = id;
            }
            pop = true;
        }
        if (pop) {
            // the far child's box test, deferred by the reference to the pop, depends on t only through tmin < t
            cur = HPT_REF_DONE;
            while (sp > 0) {
                sp--;
                uint2 e = stack[sp * stride];
                if (__uint_as_float(e.y) < t) { cur = (int)e.x; break; }
            }
            if (cur == HPT_REF_DONE) break;
        }
    }
    return hit;
}

// closest hit: primitives loop, then the BVH (ray_kernel.cl:726-745)
__device__ __forceinline__ int closest_hit(const RenderParams &p, f3 o, f3 d, float &t, uint2 *stack, int stride) {
    int id = HPT_HIT_MISS;
    t = 1000.0f;
    for (int i = 0; i < p.n_prims; i++)
        if (intersect_prim(p.prims + i, o, d, t)) id = HPT_HIT_PRIMITIVE(i);
    int tri = traverse(p, o, d, t, stack, stride);
    if (tri >= 0) id = tri;
    return id;
}

// ---- shading helpers: ray_kernel.cl:244-256, 269-302, 615-706 ----
struct Mat { float4 refractive, specular, diffuse, emissive; };

__device__ __forceinline__ f3 sample_hemisphere(f3 normal, uint32_t key, uint32_t &ctr) {
    float z = draw(key, ctr);
    float r2 = draw(key, ctr);
    float r = sqrtf(z);
    float phi = 6.283185307179586f * r2;
    float sn, cs;
    sincos_2pi(phi, sn, cs);
    float x = cs * r;
    float y = sn * r;
    z = sqrtf(1.0f - z);
    f3 axis;
    if (fabsf(normal.x) < HPT_ONE_THIRD_RT) axis = mk3(1, 0, 0);
    else if (fabsf(normal.y) < HPT_ONE_THIRD_RT) axis = mk3(0, 1, 0);
    else axis = mk3(0, 0, 1);
    f3 u = normalize3(cross3(axis, normal));
    f3 v = cross3(normal, u);
    return (u * x + v * y) + normal * z;
}
__device__ __forceinline__ f3 reflect_dir(f3 n, f3 d) { return normalize3(d - n * (2.0f * dot3(n, d))); }
__device__ __forceinline__ bool refract_dir(f3 D, f3 N, float index, f3 &out) {
    float dn = dot3(D, N);
    float a = 1.0f - (pow2f(index) * (1.0f - pow2f(dn)));
    if (a < 0.0f) return false;
    out = normalize3((D - N * dn) * index - N * sqrtf(a));
    return true;
}
__device__ __forceinline__ float schlick_fresnel(f3 n, f3 d, float n2) {
    float F0 = pow2f((1.0f - n2) / (1.0f + n2));
    if (dot3(d, n) < 0.0f) return F0 + (1.0f - F0) * pow5f(1.0f - dot3(n, -d));
    return F0 + (1.0f - F0) * pow5f(1.0f - dot3(n, d));
}
__device__ __forceinline__ f3 refractive_dir(f3 N, f3 d, float n) {
    f3 tdir;
    bool ok;
    if (dot3(d, N) < 0.0f) ok = refract_dir(d, N, 1.0f / n, tdir);
    else ok = refract_dir(d, -N, n, tdir);
    if (ok) return tdir;
    float ndoti = dot3(N, -d);
    return N * (ndoti + ndoti) + d;
}

// ---- one bounce of the loop body pathtracer_kernel.cl:759-849, after the closest hit (id, t) of
// ray r is known.  Returns true when the path goes on with the new ray left in r (the caller then
// advances b and stops at max_depth), false when it ended (light, miss, Russian roulette).
__device__ __forceinline__ bool shade_bounce(const RenderParams &p, int id, float t, Ray &r, uint32_t key, uint32_t &ctr,
                                             int b, f3 &color, f3 &weight) {
    if (id == HPT_HIT_MISS) {
        color = color + weight * mk3(0.8f, 0.9f, 0.95f);           // BACKGROUND_COLOR, ray_kernel.cl:13
        return false;
    }
    f3 point = r.o + r.d * t;
    f3 normal;
    Mat m;
    if (id >= 0) {
        const float4 *q = p.tris + 3 * (size_t)id;
        const float4 e1 = q[1], e2 = q[2];
        HPT_KEEP_ADDRESS(q, e2.z);
        normal = normalize3(cross3(mk3(e1), mk3(e2)));
        if (dot3(normal, r.d) > 0.0f) normal = -normal;
        // triangles use materials[0]: one address for every lane, read with scalar loads
        const __attribute__((address_space(4))) hpt_material *mm = (const __attribute__((address_space(4))) hpt_material *)p.mats;
        m.refractive = make_float4(mm->refractive.x, mm->refractive.y, mm->refractive.z, mm->refractive.w);
        m.specular = make_float4(mm->specular.x, mm->specular.y, mm->specular.z, mm->specular.w);
        m.diffuse = make_float4(mm->diffuse.x, mm->diffuse.y, mm->diffuse.z, mm->diffuse.w);
        m.emissive = make_float4(mm->emissive.x, mm->emissive.y, mm->emissive.z, mm->emissive.w);
    } else {
        const hpt_primitive *pr = p.prims + (-id - 2);
        const float4 *pp = reinterpret_cast<const float4 *>(pr);
        m.refractive = pp[0]; m.specular = pp[1]; m.diffuse = pp[2]; m.emissive = pp[3];
        if (pr->type.x == 0.0f) normal = normalize3(point - mk3(pp[4]));
        else normal = normalize3(mk3(pp[6]));
    }
    if (m.emissive.y > 0.0f || m.emissive.z > 0.0f || m.emissive.w > 0.0f) {          // islight
        color = color + weight * mk3(m.emissive);
        return false;
    }
    bool fresnel = draw(key, ctr) < schlick_fresnel(normal, r.d, m.refractive.x);
    bool specular = (m.specular.x > HPT_EPSILON) || (m.specular.y > HPT_EPSILON) ||
                    (m.specular.z > HPT_EPSILON) || (m.specular.w > HPT_EPSILON);
    f3 nd;
    if (fresnel && specular) nd = reflect_dir(normal, r.d);
    else if (m.refractive.w == 1.0f) {
        if (fresnel) nd = reflect_dir(normal, r.d);
        else nd = refractive_dir(normal, r.d, m.refractive.x);
    } else {
        weight = weight * mk3(m.diffuse);
        nd = sample_hemisphere(normal, key, ctr);
        weight = weight * (dot3(normal, nd) * p.diffuse_scale);
    }
    f3 no = point + nd * HPT_EPSILON;
    if (b > p.rr_start) {                                           // Russian roulette, pathtracer_kernel.cl:836-845
        float mx = (weight.x > weight.y) ? weight.x : weight.y;
        mx = (weight.z > mx) ? weight.z : mx;
        float pterm = draw(key, ctr);
        if (mx < pterm) return false;
        weight = mk3(weight.x / mx, weight.y / mx, weight.z / mx);
    }
    r.o = no; r.d = nd;
    return true;
}

// work item -> pixel: 8x8 tiles, tile t belongs to rank t % world
__device__ __forceinline__ int work_to_pixel(const RenderParams &p, unsigned work) {
    int tile = (int)(work >> 6) * p.world + p.rank;
    int k = (int)(work & 63u);
    if (tile >= p.n_tiles) return -1;
    int px = (tile % p.tiles_x) * 8 + (k & 7), py = (tile / p.tiles_x) * 8 + (k >> 3);
    if (px >= p.width || py >= p.height) return -1;
    return py * p.width + px;
}

extern __shared__ uint2 hpt_lds_stack[];

// running mean of Pathtracer::render(), pathtracer.cpp:200-217; it = 1-based iteration
__device__ __forceinline__ f3 fold_sample(f3 mean, f3 c, int it, bool raw) {
    if (it == 1 || raw) return c;
    float i1 = 1.0f / (float)it;
    float i2 = (float)(it - 1) / (float)it;
    return mk3((i1 * c.x) + (i2 * mean.x), (i1 * c.y) + (i2 * mean.y), (i1 * c.z) + (i2 * mean.z));
}


// ---- per-lane traversal state of one ray and the two step kinds, shared by render_kernel and trace_kernel ----
// cur: >= 0 interior node to test; < 0 leaf, ~cur = (next triangle << 6) | (triangles left - 1);
// HPT_REF_DONE: traversal over; HPT_REF_IDLE: the lane has nothing left to do.  Each state is ONE comparison
// on cur, so its lane mask is one v_cmp.
struct Trav {
    hpt_v2f oxy, ozix, iyz;      // {o.x,o.y} {o.z,inv.x} {inv.y,inv.z}: the register pairs of slab2_minmax
    f3 rd;                       // direction (triangle tests, shading)
    float t;
    int hit, cur, sp;
    unsigned flags;              // 1,2,4 = dirIsNeg x,y,z (bits of one VGPR, not loop-carried bools)
};

// Stack entry e of a lane: LDS while e < stack_cap (the common case: near-first descent keeps few far
// children pending), HBM beyond.  Two independent predicated accesses (not an if/else that hipcc would merge
// into one flat access through a selected generic pointer); the HBM half is skipped by an execz branch.
__device__ __forceinline__ void trav_push(const RenderParams &p, Trav &s, uint2 *stack, int stride, uint2 *spill, uint2 e) {
    if (s.sp < p.stack_cap) stack[s.sp * stride] = e;
    if (s.sp >= p.stack_cap) spill[(size_t)(s.sp - p.stack_cap) * p.spill_stride] = e;
    s.sp++;
}
// deferred far-child test: tmin < t with the CURRENT t
__device__ __forceinline__ void trav_pop(const RenderParams &p, Trav &s, uint2 *stack, int stride, uint2 *spill) {
    int ref = HPT_REF_DONE;
    while (s.sp > 0) {
        s.sp--;
        uint2 e = make_uint2(0u, 0u);
        if (s.sp < p.stack_cap) e = stack[s.sp * stride];
        if (s.sp >= p.stack_cap) {
            const uint2 *a = spill + (size_t)(s.sp - p.stack_cap) * p.spill_stride;
            e = *a;
            HPT_KEEP_ADDRESS(a, e.x);
        }
        if (__uint_as_float(e.y) < s.t) { ref = (int)e.x; break; }
    }
    s.cur = ref;
}

// TRI step: up to tris_per_trip triangles of the current leaf (or of the next one, when the pop lands on a
// leaf), two per iteration from one pair record.  A lane is in the TRI state exactly when cur < 0, so the state
// is re-read from cur, not carried.  ~cur = (next pair << 6) | (triangles left - 1).
// COUNT (the counting instantiation of render_kernel, HPT_COUNT_LANE_LOADS=1): loads = the wave's count of 16-B
// lane-loads (a scalar: active lanes x loads of the group, from the ballots the step takes anyway); n_in = lanes in
// the TRI state at entry.  The count costs 1.4 % at C3, so the default kernel does not carry it.
template <bool SINGLES, bool COUNT>
__device__ __forceinline__ void trav_tri_steps(const RenderParams &p, Trav &s, uint2 *stack, int stride, uint2 *spill,
                                               int n_in, unsigned long long &loads) {
    if (SINGLES) {
        // one triangle per 48-B record: ~cur = (next record << 6) | (triangles left - 1)
        if (COUNT) loads += (unsigned)(n_in * HPT_SINGLE_FLOAT4S);
        for (int k = 0; k < p.tris_per_trip; k++) {
            if (s.cur < 0) {
                const int packed = ~s.cur, rec = packed >> 6;
                const unsigned rec_off = (unsigned)rec * 48u;
                const float4 *q = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(p.pairs) + rec_off);
                const float4 q0 = q[0], q1 = q[1], q2 = q[2];
                HPT_KEEP_ADDRESS(rec_off, q2.y);
                if (intersect_tri(mk3(q0.x, q0.y, q0.z), mk3(q0.w, q1.x, q1.y), mk3(q1.z, q1.w, q2.x),
                                  mk3(s.oxy.x, s.oxy.y, s.ozix.x), s.rd, s.t))
                    s.hit = __float_as_int(q2.y);
                if ((packed & 63) == 0) trav_pop(p, s, stack, stride, spill);
                else s.cur = ~(packed + 63);                      // next record, one fewer left
            }
            const unsigned long long left_in = hpt_ballot(s.cur < 0);
            if (left_in == 0ull) break;
            if (COUNT && k + 1 < p.tris_per_trip) loads += (unsigned)(__popcll(left_in) * HPT_SINGLE_FLOAT4S);
        }
        return;
    }
    if (COUNT) loads += (unsigned)(n_in * HPT_PAIR_FLOAT4S);
    for (int k = 0; k < p.tris_per_trip; k += 2) {
        if (s.cur < 0) {
            const int packed = ~s.cur, pair = packed >> 6, left = (packed & 63) + 1;
            const unsigned pair_off = (unsigned)pair * 80u;
            const float4 *q = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(p.pairs) + pair_off);
            const float4 q0 = q[0], q1 = q[1], q2 = q[2];
            const bool two = left >= 2;
            // all five quarters in ONE group of loads, also for a lane whose leaf has one triangle left (its record's
            // second half is zeros or the array's padding): a group interrupted by the exec-mask change for `two`
            // costs more than the two extra lane-loads (C3 89.7 -> 88.5 ms at 32 spp)
            const float4 q3 = q[3], q4 = q[4];
            HPT_KEEP_ADDRESS(pair_off, q4.w);
            const f3 o = mk3(s.oxy.x, s.oxy.y, s.ozix.x);
            if (intersect_tri(mk3(q0.x, q0.y, q0.z), mk3(q0.w, q1.x, q1.y), mk3(q1.z, q1.w, q2.x), o, s.rd, s.t))
                s.hit = __float_as_int(q2.y);
            if (two && intersect_tri(mk3(q2.w, q3.x, q3.y), mk3(q3.z, q3.w, q4.x), mk3(q4.y, q4.z, q4.w), o, s.rd, s.t))
                s.hit = __float_as_int(q2.z);
            if (left <= 2) trav_pop(p, s, stack, stride, spill);
            else s.cur = ~(packed + 64 - 2);                  // next pair, two fewer left
        }
        const unsigned long long left_in = hpt_ballot(s.cur < 0);
        if (left_in == 0ull) break;
        if (COUNT && k + 2 < p.tris_per_trip) loads += (unsigned)(__popcll(left_in) * HPT_PAIR_FLOAT4S);
    }
}

// NODE step: both children per visit; the wave stays here, without a full re-vote, while at least
// node_repeat_min lanes still have an interior node to test.
// PAIR (an instantiation of render_kernel of its own, picked by the host for single-record trees of a gigabyte and more,
// whose every node fetch is a miss to HBM): while at most 32 lanes are at a node, each of them takes an idle lane as helper (one
// ds_permute + two ds_bpermute to find each other and pass the node index); owner and helper fetch half the node each, in
// TWO load instructions instead of four, and the owner pulls the other half over with seven ds_bpermute.  Same 4 lane-loads
// per visit.  C4 +2.2 %; smaller trees lose 3-7 % to the three extra hops in front of every fetch (profiles/r02_tuning.md).
template <bool COUNT, bool PAIR = false>
__device__ __forceinline__ void trav_node_steps(const RenderParams &p, Trav &s, uint2 *stack, int stride, uint2 *spill,
                                                int n_in, unsigned long long &loads, const float4 *lds_nodes = nullptr) {
    // (counting build) global lane-loads only: a visit served from the LDS copy of the tree's top asks nothing of the L1
    if (COUNT) loads += 4u * (unsigned)__popcll(hpt_ballot((unsigned)s.cur < (unsigned)HPT_REF_IDLE && (unsigned)s.cur >= p.n_lds_nodes));
    int n_now = n_in;
    for (;;) {
#ifdef HPT_PROFILE_STEPS
        {   // development build: how many NODE iterations have all their lanes at ONE node (a scalar load would do)
            const bool in = (unsigned)s.cur < (unsigned)HPT_REF_IDLE;
            const unsigned long long act = hpt_ballot(in);
            const int first = __builtin_amdgcn_readfirstlane(in ? s.cur : -1);     // value of the first ACTIVE lane
            const bool uni = act != 0ull && hpt_ballot(in && s.cur != first) == 0ull && in;
            const unsigned long long U = hpt_ballot(uni);
            // lanes that are the first of their quad at their node: the 64-B lines the L1 looks up (it merges the
            // lanes of a quad that read one line, profiles/r02/coop_fetch/gather_coop.txt)
            const int mine = in ? s.cur : -1 - (int)(threadIdx.x & 3u);
            const unsigned qk = threadIdx.x & 3u;
            const int c0 = __builtin_amdgcn_mov_dpp(mine, 0x00, 0xf, 0xf, true), c1 = __builtin_amdgcn_mov_dpp(mine, 0x55, 0xf, 0xf, true);
            const int c2 = __builtin_amdgcn_mov_dpp(mine, 0xaa, 0xf, 0xf, true);
            const bool leader = in && !(qk > 0u && c0 == mine) && !(qk > 1u && c1 == mine) && !(qk > 2u && c2 == mine);
            const unsigned long long Lq = hpt_ballot(leader);
            // lanes that are the first of the WAVE at their node: what any re-ordering inside the wave could reach
            bool first_in_wave = in;
            for (unsigned long long mm = act; mm != 0ull; mm &= mm - 1ull) {
                const int j = __builtin_ctzll(mm);
                const int cj = __builtin_amdgcn_readlane(s.cur, j);
                if ((unsigned)j < (threadIdx.x & 63u) && cj == s.cur) first_in_wave = false;
            }
            const unsigned long long Lw = hpt_ballot(first_in_wave);
            if (act != 0ull && (threadIdx.x & 63u) == (unsigned)__builtin_ctzll(act)) {
                atomicAdd(p.counters + 11, 1ull); atomicAdd(p.counters + 13, (unsigned long long)__popcll(act));
                if (U != 0ull) atomicAdd(p.counters + 12, 1ull);
                atomicAdd(p.counters + 14, (unsigned long long)__popcll(Lq));
                atomicAdd(p.counters + 12, (unsigned long long)__popcll(Lw) << 20);
            }
        }
#endif
        const bool pair_mode = PAIR && n_now <= 32;        // wave-uniform
        float4 pA = make_float4(0, 0, 0, 0), pB = pA, pZ = pA; int4 pM = make_int4(0, 0, 0, 0);
        if (PAIR && pair_mode) {
            const unsigned lane = threadIdx.x & 63u;
            const bool in = (unsigned)s.cur < (unsigned)HPT_REF_IDLE;
            const unsigned long long IN = hpt_ballot(in);
            const unsigned below = __builtin_amdgcn_mbcnt_hi((unsigned)(IN >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)IN, 0u));
            const unsigned irank = lane - below, n_act = (unsigned)__popcll(IN);
            const int by_pos = __builtin_amdgcn_ds_permute((int)((in ? below : n_act + irank) << 2), (int)lane);
            const int partner = __builtin_amdgcn_ds_bpermute((int)((in ? n_act + below : irank) << 2), by_pos) << 2;
            const int pcur = __builtin_amdgcn_ds_bpermute(partner, s.cur);
            const bool helps = !in && irank < n_act;
            if (in || helps) {
                const unsigned off = ((unsigned)(in ? s.cur : pcur) << 6) + (in ? 0u : 32u);
                const float4 *q = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(p.nodes) + off);
                pA = q[0]; pB = q[1];
                HPT_KEEP_ADDRESS(off, pB.w);
            }
            pZ.x = __int_as_float(__builtin_amdgcn_ds_bpermute(partner, __float_as_int(pA.x)));
            pZ.y = __int_as_float(__builtin_amdgcn_ds_bpermute(partner, __float_as_int(pA.y)));
            pZ.z = __int_as_float(__builtin_amdgcn_ds_bpermute(partner, __float_as_int(pA.z)));
            pZ.w = __int_as_float(__builtin_amdgcn_ds_bpermute(partner, __float_as_int(pA.w)));
            pM.x = __builtin_amdgcn_ds_bpermute(partner, __float_as_int(pB.x));
            pM.y = __builtin_amdgcn_ds_bpermute(partner, __float_as_int(pB.y));
            pM.z = __builtin_amdgcn_ds_bpermute(partner, __float_as_int(pB.z));
            // landed before the step goes on (a permute in flight at the next TRI step would put an s_waitcnt into its load group)
            asm volatile("" : : "v"(pZ.x), "v"(pZ.y), "v"(pZ.z), "v"(pZ.w), "v"(pM.x), "v"(pM.y), "v"(pM.z));
        }
        if ((unsigned)s.cur < (unsigned)HPT_REF_IDLE) {
            // 32-bit byte offset (n_wide < 2^25): saddr + voffset addressing, no 64-bit address math
            float4 X, Y, Z; int4 m;
            if (PAIR && pair_mode) {
                X = pA; Y = pB; Z = pZ; m = pM;
            } else if ((unsigned)s.cur < p.n_lds_nodes) {
                // The top of the tree from this block's LDS copy (4 x ds_read_b128; n_lds_nodes = 0 where no copy is made):
                // C3 -18.5 % vector-L1 lookups, TA busy 96.8 -> 91.4 %, its stalls on the cache 29 -> 19 % of that, frame -5 %
                // (profiles/r03/lds_top/).  Node-major like the HBM array: one address, four instruction offsets (a
                // quarter-major copy, conflict-free for lanes at consecutive nodes, was 1.5 % slower: three more address
                // computations per visit).
                const float4 *q = lds_nodes + ((unsigned)s.cur << 2);
                X = q[0]; Y = q[1]; Z = q[2];
                m = *reinterpret_cast<const int4 *>(q + 3);
            } else {
                const unsigned node_off = (unsigned)s.cur << 6;
                const float4 *q = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(p.nodes) + node_off);
                X = q[0]; Y = q[1]; Z = q[2];
                m = *reinterpret_cast<const int4 *>(q + 3);
                HPT_KEEP_ADDRESS(node_off, m.w);
            }
            const bool nx = s.flags & 1u, ny = s.flags & 2u, nz = s.flags & 4u;
            float t0, t1;
            unsigned long long M0, M1;                    // lanes whose child 0 / child 1 is hit with tmin < t
            HPT_SLAB2(X, Y, Z, s.oxy, s.ozix, s.iyz, nx, ny, nz, s.t, M0, M1, t0, t1);
            // dirIsNeg[axis] (ray_kernel.cl:440): second child first.  Near/far hit flags are 64-bit
            // lane-mask arithmetic (scalar ops) instead of per-lane selects.
            const unsigned long long SF = hpt_ballot(((s.flags >> m.z) & 1u) != 0u);
            const bool second_first = __builtin_amdgcn_inverse_ballot_w64(SF);
            const int refN = second_first ? m.y : m.x, refF = second_first ? m.x : m.y;
            const float tF = second_first ? t0 : t1;
            const bool hN = __builtin_amdgcn_inverse_ballot_w64((M1 & SF) | (M0 & ~SF));
            const bool hF = __builtin_amdgcn_inverse_ballot_w64((M0 & SF) | (M1 & ~SF));
            if (hN) {
                if (hF) trav_push(p, s, stack, stride, spill, make_uint2((unsigned)refF, __float_as_uint(tF)));
                s.cur = refN;
            } else if (hF) s.cur = refF;
            else trav_pop(p, s, stack, stride, spill);
        }
        const int again = __popcll(hpt_ballot((unsigned)s.cur < (unsigned)HPT_REF_IDLE));
        if (again < p.node_repeat_min) break;
        n_now = again;
        if (COUNT) loads += 4u * (unsigned)__popcll(hpt_ballot((unsigned)s.cur < (unsigned)HPT_REF_IDLE && (unsigned)s.cur >= p.n_lds_nodes));
    }
}

// Starts the closest-hit query of ray (o, d): primitives first, then the tree (ray_kernel.cl:726-745).
// Leaves cur = root reference, or HPT_REF_DONE when the root box is missed / there is no tree.
// LONG: the kernel instantiation for long primitive lists (>= HPT_TWO_PHASE_MIN_PRIMS; chosen by the host per scene),
// which walks them in two phases; the other instantiation does not carry that code at all.
template <bool LONG>
__device__ __forceinline__ void trav_start(const RenderParams &p, Trav &s, f3 o, f3 d) {
    s.t = 1000.0f;
    s.hit = HPT_HIT_MISS;
    if (LONG) prims_two_phase(p, o, d, s.t, s.hit);
    else
        for (int i = 0; i < p.n_prims; i++)
            if (intersect_prim(p.prims + i, o, d, s.t)) s.hit = HPT_HIT_PRIMITIVE(i);
    s.sp = 0;
    s.cur = HPT_REF_DONE;
    s.rd = d;
    s.oxy = hpt_v2f{o.x, o.y};
    s.ozix.x = o.z;
    if (p.root_ref != HPT_REF_DONE) {
        const f3 inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
        s.ozix.y = inv.x; s.iyz = hpt_v2f{inv.y, inv.z};
        const bool nx = inv.x < 0.0f, ny = inv.y < 0.0f, nz = inv.z < 0.0f;
        s.flags = (nx ? 1u : 0u) | (ny ? 2u : 0u) | (nz ? 4u : 0u);
        float tm;
        bool ok = HPT_SLAB(mk3(p.root_lo[0], p.root_lo[1], p.root_lo[2]), mk3(p.root_hi[0], p.root_hi[1], p.root_hi[2]),
                           o, inv, nx, ny, nz, tm);
        if (ok && tm < s.t) s.cur = p.root_ref;
    }
}

// Re-grouping at the SHADE -> trav_start hand-off.  The vector L1 merges the lanes of a QUAD that read one 64-B
// line into one tag lookup (not lanes further apart: profiles/r02/coop_fetch/gather_coop.txt), so rays that walk the
// same nodes at the same time are cheaper when they sit in one quad.  The paths that start a ray in this step
// (`moving`: their traversal stack is empty, their whole state is the 16 registers below) are dealt out again
// among the lanes they occupy: camera rays first, in their pixel order, then the bounce rays by direction octant
// (same octant = same near/far child at every node), and the places are taken quads with four moving lanes first,
// then three, ..., so that runs of one octant fill whole quads.  Which lane computes a path does not enter its
// result (the RNG is keyed by path, the sample slot travels with it), so images stay bit-identical.
// C3: distinct nodes per quad / lane-visits 0.928 -> 0.827, 88.4 -> 86.3 ms at 32 spp; C5 +1.4 % once it ran 8 waves per SIMD
// (at 6 it lost 0.5 %) (profiles/r02_tuning.md).
__device__ __forceinline__ void regroup_starting(bool moving, unsigned lane, Ray &r, int &b, unsigned &out_idx,
                                                 uint32_t &key, uint32_t &ctr, f3 &color, f3 &weight) {
    const unsigned long long Mv = hpt_ballot(moving);
    // stable counting sort of the rays: key 0 camera ray, 1..8 octant of a bounce ray
    const unsigned sk = !moving ? 99u : (b == 0 ? 0u : 1u + ((r.d.x < 0.0f ? 1u : 0u) | (r.d.y < 0.0f ? 2u : 0u) | (r.d.z < 0.0f ? 4u : 0u)));
    unsigned rank = 0, n_moving = 0;
    for (unsigned k = 0; k < 9u; k++) {
        const unsigned long long B = hpt_ballot(sk == k);
        const unsigned below = __builtin_amdgcn_mbcnt_hi((unsigned)(B >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)B, 0u));
        if (sk == k) rank = n_moving + below;
        n_moving += (unsigned)__popcll(B);
    }
    // order of the places: by the number of moving lanes in the quad, then by lane
    const unsigned qn = (unsigned)__popcll((Mv >> (lane & 60u)) & 15ull);
    unsigned place = 0, n_placed = 0;
    for (unsigned c = 4u; c >= 1u; c--) {
        const unsigned long long B = hpt_ballot(moving && qn == c);
        const unsigned below = __builtin_amdgcn_mbcnt_hi((unsigned)(B >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)B, 0u));
        if (qn == c) place = n_placed + below;
        n_placed += (unsigned)__popcll(B);
    }
    // lane j < n_moving learns which lane holds the ray of rank j (the lanes that stay push to the lanes behind, so
    // the push is a permutation of all 64); place j then pulls from there, every other lane from itself
    const unsigned stay = lane - __builtin_amdgcn_mbcnt_hi((unsigned)(Mv >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)Mv, 0u));
    const int by_rank = __builtin_amdgcn_ds_permute((int)((moving ? rank : n_moving + stay) << 2), (int)lane);
    const int from = __builtin_amdgcn_ds_bpermute((int)(place << 2), by_rank);
    const int src = (moving ? from : (int)lane) << 2;
#define HPT_MOVE_F(x) x = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(x)))
#define HPT_MOVE_I(x) x = (std::remove_reference_t<decltype(x)>)__builtin_amdgcn_ds_bpermute(src, (int)(x))
    HPT_MOVE_F(r.o.x); HPT_MOVE_F(r.o.y); HPT_MOVE_F(r.o.z); HPT_MOVE_F(r.d.x); HPT_MOVE_F(r.d.y); HPT_MOVE_F(r.d.z);
    HPT_MOVE_F(color.x); HPT_MOVE_F(color.y); HPT_MOVE_F(color.z); HPT_MOVE_F(weight.x); HPT_MOVE_F(weight.y); HPT_MOVE_F(weight.z);
    HPT_MOVE_I(b); HPT_MOVE_I(out_idx); HPT_MOVE_I(key); HPT_MOVE_I(ctr);
    // all sixteen have arrived before the step ends: a permute still in flight at the next TRI / NODE step makes hipcc
    // put an s_waitcnt inside that step's group of loads (tests/test_host.py checks the groups)
    asm volatile("" : : "v"(r.o.x), "v"(r.o.y), "v"(r.o.z), "v"(r.d.x), "v"(r.d.y), "v"(r.d.z), "v"(color.x), "v"(color.y));
    asm volatile("" : : "v"(color.z), "v"(weight.x), "v"(weight.y), "v"(weight.z), "v"(b), "v"(out_idx), "v"(key), "v"(ctr));
#undef HPT_MOVE_F
#undef HPT_MOVE_I
}

// Persistent wavefront render kernel.
//
// Work item = one path: (pixel slot, iteration).  Items are numbered tile-major, then iteration, then pixel
// in the 8x8 tile, so the lanes that fetch together get coherent camera rays, and any number of lanes (or
// GPUs) shares a frame without a tail: a pixel's samples are NOT tied to one lane.  Each finished path
// stores its radiance sample to `samples`; fold_kernel then folds a pixel's samples in iteration order
// (the running mean of pathtracer.cpp:200-217 is order-dependent in fp32), so the image is bit-identical
// whatever the grid, the schedule or the shard.
//
// A lane is always in one of three states: NODE (an interior two-child node to test), TRI (a leaf with
// triangles left) or SHADE (its ray's traversal is over: shade, Russian roulette, next ray or next path
// from the global work counter).  Each trip of the loop the wave counts its lanes per state with three
// 64-bit ballots and executes ONE step kind, chosen wave-uniformly (scalar branch): a lane never waits for
// another lane's longer traversal, longer path or longer leaf, which is what held the one-path-per-lane
// form at 12 % lane utilisation (profiles/r01_v1_naive).  SHADE is the expensive, rare step, so it waits
// until p.shade_threshold lanes want it or nothing else is runnable.  Work order per path is unchanged,
// so results stay bit-identical to the oracle.
// Waves per SIMD of render_kernel: 7 (72 VGPRs, 94 SGPRs -- the hardware admits floor(800 / (ceil(sgpr/16)*16 + 16)) blocks of
// 256 per CU -- and 8-20 B of scratch per lane, loop-invariant values re-read only in SHADE): +1.7 % at C3, +1.5 % at C2 and
// +5.6 % at C5 over 6 since the re-grouping and the step weights took load off the L1 (before them 7 lost 0.7 %).  Trees that
// stream from HBM lose with the seventh block (C4 -3.3 %, 378 MB -1.1 %, 126 MB +0.9 %): the host launches six blocks per
// CU of the same kernel for them (launch_render(), as fast as a 6-wave build).  iter_trace_kernel: 6 (HPT_MIN_WAVES; 7 costs
// 1.7 % per call).  8 spills 52 B per lane and gains nothing at C3 (profiles/r02_tuning.md).
#ifndef HPT_MIN_WAVES
#define HPT_MIN_WAVES 6
#endif
#ifndef HPT_MIN_WAVES_PAIRS
#define HPT_MIN_WAVES_PAIRS 7
#endif
#ifndef HPT_MIN_WAVES_LONG           // pair-record trees with a long primitive list (C5): 8 (64 VGPRs, 56 B of scratch per lane; the eight
#define HPT_MIN_WAVES_LONG 8         // 20-KB LDS stacks fill the CU's 160 KB exactly): +2.0 % over 7, its SHADE steps leave the L1 idle
#endif
#ifdef HPT_MAX_SGPRS                 // development: cap the SGPRs (the hardware admits floor(800 / (ceil(sgpr/16)*16 + 16)) blocks of 256 per CU)
#define HPT_SGPR_CAP __attribute__((amdgpu_num_sgpr(HPT_MAX_SGPRS)))
#else
#define HPT_SGPR_CAP
#endif
template <bool SINGLES, bool COUNT, bool LONG, bool PAIR = false>
__global__ void __launch_bounds__(256, (SINGLES ? HPT_MIN_WAVES_PAIRS : LONG ? HPT_MIN_WAVES_LONG : HPT_MIN_WAVES_PAIRS)) HPT_SGPR_CAP render_kernel(const RenderParams p) {
    uint2 *stack = hpt_lds_stack + threadIdx.x;
    const int stride = blockDim.x;
    unsigned n_primary = 0, n_bounce = 0, n_paths = 0;   // per WAVE (from ballots: scalar registers, not three per lane)
    unsigned long long n_loads = 0;                      // per WAVE: 16-B lane-loads issued (nodes, triangle records, hit triangles' edges)
#ifdef HPT_PROFILE_STEPS
    unsigned long long prof_it[3] = {0, 0, 0}, prof_act[3] = {0, 0, 0};
    unsigned long long prof_waitS = 0, prof_idle = 0;     // per lane: trips spent waiting for SHADE / without work
#endif

    int b = -1;                          // b == -1: the lane needs a new path
    const unsigned lane = threadIdx.x & 63u;
    unsigned chunk_next = 0, chunk_end = 0;  // work items this WAVE has reserved from the work counter (wave-uniform)
    bool exhausted = false;                  // the work counter has passed n_work (wave-uniform)
    unsigned out_idx = 0;                // sample slot of the current path: (sample of the launch) * n_slots + pixel slot
    uint32_t key = 0, ctr = 0;
    f3 color = mk3(0, 0, 0), weight = mk3(1, 1, 1);
    Trav tv;
    tv.oxy = hpt_v2f{0, 0}; tv.ozix = hpt_v2f{0, 0}; tv.iyz = hpt_v2f{0, 0}; tv.rd = mk3(0, 0, 1);
    tv.t = 1000.0f; tv.hit = HPT_HIT_MISS; tv.cur = HPT_REF_DONE; tv.sp = 0; tv.flags = 0;
    uint2 *const spill = p.spill + (blockIdx.x * blockDim.x + threadIdx.x);
    // every block keeps the first n_lds_nodes nodes (the top of the tree, numbered first by the host) in LDS behind its
    // traversal stacks: as many as fit beside them at this kernel's occupancy (launch_render())
    float4 *const lds_nodes = reinterpret_cast<float4 *>(hpt_lds_stack + (size_t)p.stack_cap * blockDim.x);
    for (unsigned i = threadIdx.x; i < p.n_lds_nodes * 4u; i += blockDim.x) lds_nodes[i] = reinterpret_cast<const float4 *>(p.nodes)[i];
    __syncthreads();

    for (;;) {
        const bool wantS = tv.cur == HPT_REF_DONE;
        const bool wantN = (unsigned)tv.cur < (unsigned)HPT_REF_IDLE;
        const bool wantT = tv.cur < 0;
        const int nS = __popcll(hpt_ballot(wantS)), nN = __popcll(hpt_ballot(wantN)), nT = __popcll(hpt_ballot(wantT));
        if (nS + nN + nT == 0) break;
#ifdef HPT_PROFILE_STEPS
        {   // development build: per-kind trips and active lanes (bench/ab.py prints them)
            int kind = (nS >= p.shade_threshold || nN + nT == 0) ? 0 : ((nT * p.tri_weight > nN * 4) ? 1 : 2);
            prof_it[kind]++;
            if ((kind == 0 && wantS) || (kind == 1 && wantT) || (kind == 2 && wantN)) prof_act[kind]++;
            if (kind != 0 && wantS) prof_waitS++;
            if (tv.cur == HPT_REF_IDLE) prof_idle++;
        }
#endif

        if (nS >= p.shade_threshold || nN + nT == 0) {
            {
                Ray r;
                r.o = mk3(0, 0, 0); r.d = mk3(0, 0, 1);
                int gid = -1, s = 0;     // pixel and iteration of a path fetched in this step (only its key and camera ray need them)
                bool ended = false;
                n_primary += (unsigned)__popcll(hpt_ballot(wantS && b == 0));
                n_bounce += (unsigned)__popcll(hpt_ballot(wantS && b > 0));
                if (COUNT) n_loads += 2u * (unsigned)__popcll(hpt_ballot(wantS && b >= 0 && tv.hit >= 0));   // e1, e2 of the hit triangle
                if (wantS && b >= 0) {   // a ray has just finished: one bounce of the path loop
                    r.o = mk3(tv.oxy.x, tv.oxy.y, tv.ozix.x); r.d = tv.rd;
                    bool go_on = shade_bounce(p, tv.hit, tv.t, r, key, ctr, b, color, weight);
                    b++;
                    if (!go_on || b >= p.max_depth) {
                        float *o = p.samples + 3 * (size_t)out_idx;
                        o[0] = color.x; o[1] = color.y; o[2] = color.z;
                        ended = true;
                        b = -1;
                    }
                }
                // next path of this shard: (tile, iteration, pixel in tile).  Work items are reserved per WAVE in
                // chunks of fetch_chunk (one atomic per chunk on the single work counter) and handed to the lanes
                // in order, so lanes that fetch together still get adjacent pixels of one iteration.  (Wave-uniform
                // control flow: every lane takes part in the votes, lane 0 does the atomic.)
                bool want = wantS && b < 0;
                n_paths += (unsigned)__popcll(hpt_ballot(ended));
                for (;;) {
                    const unsigned long long need = hpt_ballot(want);
                    if (need == 0ull) break;
                    if (chunk_next >= chunk_end) {
                        // after the first miss a wave never touches the counter again, so the counter overshoots
                        // n_work by at most waves * fetch_chunk (the host keeps that much headroom below 2^32)
                        // Near the end of the launch (this wave's last reservation began inside the final
                        // tail_items) a wave takes 64 items at a time instead of fetch_chunk: the last waves to
                        // reserve then hold one path per lane, not four, and the grid drains sooner.
                        const unsigned take = (p.n_work - chunk_end) <= p.tail_items ? 64u : (unsigned)p.fetch_chunk;
                        unsigned at = 0xffffffffu;
                        if (!exhausted && lane == 0u) at = atomicAdd(p.work_counter, take);
                        at = __builtin_amdgcn_readfirstlane(at);
                        if (at >= p.n_work) { exhausted = true; if (want) tv.cur = HPT_REF_IDLE; break; }
                        chunk_next = at;
                        chunk_end = (p.n_work - at) < take ? p.n_work : at + take;
                    }
                    const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(need >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)need, 0u));
                    const unsigned avail = chunk_end - chunk_next;
                    if (want && rank < avail) {
                        const unsigned work = chunk_next + rank;
                        const unsigned group = work >> 6, tile = group / (unsigned)p.batch, sl = group - tile * (unsigned)p.batch;
                        const unsigned slot = tile * 64u + (work & 63u);       // within this launch's slot range
                        gid = work_to_pixel(p, p.slot_base + slot);
                        if (gid >= 0) { s = p.spp_first + (int)sl * p.spp_stride; out_idx = sl * p.n_slots + slot; want = false; }
                    }
                    const unsigned cnt = (unsigned)__popcll(need);
                    chunk_next += cnt < avail ? cnt : avail;
                }
                if (wantS && tv.cur != HPT_REF_IDLE) {
                    if (b < 0) {         // camera ray of (pixel, iteration)
                        key = path_key(p.seed, (uint32_t)gid, (uint32_t)s);
                        ctr = 0;
                        r = camera_ray(p, gid, s, key, ctr);
                        color = mk3(0, 0, 0); weight = mk3(1, 1, 1);
                        b = 0;
                    }
                }
                // The paths that start a ray in this step change lanes before they start it (regroup_starting()) -- in the
                // instantiation for gigabyte trees only.  Round 2 re-grouped everywhere (C3 +3 %): same-octant rays in one quad
                // share their node lines near the top of the tree.  Since round 3 the top of the tree is read from LDS, where
                // sharing buys nothing, and the re-grouping only costs its ~60 instructions per SHADE step: without it C1 +3.5 %,
                // C2 +2.6 %, C3 +0.4 %, C5 +-0; C4, which has no LDS copy, -0.7 % (profiles/r03_tuning.md).
                if (PAIR) regroup_starting(wantS && tv.cur != HPT_REF_IDLE, lane, r, b, out_idx, key, ctr, color, weight);
                if (wantS && tv.cur != HPT_REF_IDLE) trav_start<LONG>(p, tv, r.o, r.d);
            }
        } else if (nT * p.tri_weight > nN * 4) {     // a lane in the TRI state counts tri_weight / 4 lanes in the NODE state
            trav_tri_steps<SINGLES, COUNT>(p, tv, stack, stride, spill, nT, n_loads);
        } else {
            trav_node_steps<COUNT, PAIR>(p, tv, stack, stride, spill, nN, n_loads, lds_nodes);
        }
    }
    if (lane == 0u) {
        atomicAdd(p.counters + 0, (unsigned long long)n_primary);
        atomicAdd(p.counters + 1, (unsigned long long)n_bounce);
        atomicAdd(p.counters + 2, (unsigned long long)n_paths);
        if (COUNT) atomicAdd(p.counters + 15, n_loads);
    }
#ifdef HPT_PROFILE_STEPS
    for (int k = 0; k < 3; k++) { atomicAdd(p.counters + 3 + k, prof_it[k]); atomicAdd(p.counters + 6 + k, prof_act[k]); }
    atomicAdd(p.counters + 9, prof_waitS); atomicAdd(p.counters + 10, prof_idle);
#endif
}

// Folds the samples of one launch into the running-mean image, one lane per pixel slot, iterations in
// order: image = (1/it)*sample + ((it-1)/it)*image, it 1-based (pathtracer.cpp:200-217).  raw: the image
// receives the sample itself (hpt_kernel_execute).  Streams samples once: batch*12 B read per pixel.
__global__ void __launch_bounds__(256) fold_kernel(const RenderParams p) {
    unsigned slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= p.n_slots) return;
    int gid = work_to_pixel(p, p.slot_base + slot);
    if (gid < 0) return;
    float *px = p.image + 3 * (size_t)gid;
    f3 mean = mk3(0, 0, 0);
    if (p.spp_base > 0 && !p.raw_sample) mean = mk3(px[0], px[1], px[2]);
    for (int sl = 0; sl < p.batch; sl++) {
        const float *c = p.samples + 3 * ((size_t)sl * p.n_slots + slot);
        mean = fold_sample(mean, mk3(c[0], c[1], c[2]), p.spp_base + sl + 1, p.raw_sample != 0);
    }
    px[0] = mean.x; px[1] = mean.y; px[2] = mean.z;
}

// closest hit of the camera ray, one lane per pixel (test hook)
__global__ void __launch_bounds__(256) primary_hits_kernel(const RenderParams p, int iteration) {
    uint2 *stack = hpt_lds_stack + threadIdx.x;
    int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= p.width * p.height) return;
    uint32_t key = path_key(p.seed, (uint32_t)gid, (uint32_t)iteration), ctr = 0;
    Ray r = camera_ray(p, gid, iteration, key, ctr);
    float t;
    int id = closest_hit(p, r.o, r.d, t, stack, blockDim.x);
    p.hit_id[gid] = id;
    p.hit_t[gid] = t;
}

// ======================= iterative pipeline (hpt_first_bounce / hpt_bounce / hpt_reconstruct_paths) ==========
// The reference's live frame loop (initial_ray_kernel.cl:802-902, ray_kernel.cl:817-929,
// reconstruct_kernel.cl:10-37) with its INTENDED semantics (the committed state machine never
// restarts a finished pixel and is not reproduced): bounce 0 of every pixel is traced once and cached; each
// hpt_bounce advances every pixel's current path by one bounce; a path that ends leaves its colour as the
// pixel's contribution, counts as completed, and the pixel restarts from the cache; reconstruct folds fresh
// contributions into the per-pixel running mean.  Path n of a pixel draws from key(seed, pixel, n), counter
// continuing after the cached bounce.  All per-pixel state stays in HBM (the reference moves it over PCIe
// both ways on every call).  Each call is a traversal kernel (persistent, the step functions of render_kernel)
// followed by a shading kernel (one thread per pixel slot of the shard).
struct IterState {
    float4 *cur_a, *cur_b, *cur_c;   // current path: {o.xyz, weight.x} {d.xyz, weight.y} {color.xyz, weight.z}
    uint2 *cur_d;                    // {rng counter, bounces done}
    float4 *c_a, *c_b, *c_c;         // cache (state after bounce 0), same packing
    uint2 *c_d;                      // {rng counter, 1 if bounce 0 ended the path}
    int32_t *completed;              // paths folded per pixel (the reference's iteration_memory)
    float4 *contrib;                 // {rgb of the last finished path, 1 if it waits for reconstruct}
    int2 *hit;                       // {hit id, bits of t} of the ray traced by the current call
};

// Traversal half of one call: a persistent grid pulls the shard's pixel slots in chunks (one atomic per wave
// per chunk, lanes of a wave get neighbouring pixels), traces each pixel's current ray with the NODE / TRI
// steps of render_kernel and stores (hit, t).  first != 0: the ray is the camera ray of jitter cell 0, which
// is generated here and left in cur_a/cur_b (origin, direction) and cur_d.x (rng counter after the jitter).
// Pixels whose cached bounce already ended the path trace nothing.
template <bool SINGLES, bool LONG>
__global__ void __launch_bounds__(256, HPT_MIN_WAVES) iter_trace_kernel(const RenderParams p, const IterState it, int first) {
    uint2 *stack = hpt_lds_stack + threadIdx.x;
    const int stride = blockDim.x;
    uint2 *const spill = p.spill + (blockIdx.x * blockDim.x + threadIdx.x);
    const unsigned n = p.n_slots;
    const unsigned lane = threadIdx.x & 63u;
    const unsigned waves = gridDim.x * (blockDim.x >> 6);
    unsigned chunk = (n / (waves * 4u)) & ~63u;
    chunk = chunk < 64u ? 64u : (chunk > 1024u ? 1024u : chunk);
    unsigned chunk_next = 0, chunk_end = 0;              // wave-uniform
    unsigned traced = 0;
    unsigned long long n_loads = 0;                      // (not counted here)
    int gid = -1;
    Trav tv;
    tv.oxy = hpt_v2f{0, 0}; tv.ozix = hpt_v2f{0, 0}; tv.iyz = hpt_v2f{0, 0}; tv.rd = mk3(0, 0, 1);
    tv.t = 1000.0f; tv.hit = HPT_HIT_MISS; tv.cur = HPT_REF_DONE; tv.sp = 0; tv.flags = 0;
    for (;;) {
        const int nF = __popcll(hpt_ballot(tv.cur == HPT_REF_DONE));
        const int nN = __popcll(hpt_ballot((unsigned)tv.cur < (unsigned)HPT_REF_IDLE));
        const int nT = __popcll(hpt_ballot(tv.cur < 0));
        if (nF + nN + nT == 0) break;
        if (nF >= p.fetch_threshold || nN + nT == 0) {
            bool want = tv.cur == HPT_REF_DONE;
            if (want && gid >= 0) it.hit[gid] = make_int2(tv.hit, __float_as_int(tv.t));
            if (want) gid = -1;
            for (;;) {
                const unsigned long long need = hpt_ballot(want);
                if (need == 0ull) break;
                if (chunk_next >= chunk_end) {
                    unsigned at = 0;
                    if (lane == 0u) at = atomicAdd(p.work_counter, chunk);
                    at = __builtin_amdgcn_readfirstlane(at);
                    if (at >= n) { if (want) tv.cur = HPT_REF_IDLE; break; }
                    chunk_next = at;
                    chunk_end = (n - at) < chunk ? n : at + chunk;
                }
                const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(need >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)need, 0u));
                const unsigned avail = chunk_end - chunk_next;
                if (want && rank < avail) {
                    const int g = work_to_pixel(p, chunk_next + rank);
                    if (g >= 0 && (first || it.c_d[g].y == 0u)) { gid = g; want = false; }
                }
                const unsigned cnt = (unsigned)__popcll(need);
                chunk_next += cnt < avail ? cnt : avail;
            }
            if (tv.cur == HPT_REF_DONE && gid >= 0) {
                f3 o, d;
                if (first) {
                    const uint32_t key = path_key(p.seed, (uint32_t)gid, 0u);
                    uint32_t ctr = 0;
                    const Ray r = camera_ray(p, gid, 0, key, ctr);
                    o = r.o; d = r.d;
                    it.cur_a[gid] = make_float4(o.x, o.y, o.z, 1.0f);
                    it.cur_b[gid] = make_float4(d.x, d.y, d.z, 1.0f);
                    it.cur_d[gid] = make_uint2(ctr, 0u);
                } else {
                    const float4 a = it.cur_a[gid], b = it.cur_b[gid];
                    o = mk3(a.x, a.y, a.z); d = mk3(b.x, b.y, b.z);
                }
                traced++;
                trav_start<LONG>(p, tv, o, d);
            }
        } else if (nT * p.tri_weight > nN * 4) {
            trav_tri_steps<SINGLES, false>(p, tv, stack, stride, spill, nT, n_loads);
        } else {
            trav_node_steps<false>(p, tv, stack, stride, spill, nN, n_loads);
        }
    }
    __shared__ unsigned block_rays;
    if (threadIdx.x == 0) block_rays = 0;
    __syncthreads();
    if (traced) atomicAdd(&block_rays, traced);
    __syncthreads();
    if (threadIdx.x == 0 && block_rays) atomicAdd(p.counters + (first ? 0 : 1), (unsigned long long)block_rays);
}

// Shading half of hpt_first_bounce: bounce 0 of every pixel from the stored camera ray and hit; fills the cache
// and starts path 0 from it.
__global__ void __launch_bounds__(256) iter_first_shade_kernel(const RenderParams p, const IterState it) {
    const unsigned slot = blockIdx.x * blockDim.x + threadIdx.x;
    const int gid = slot < p.n_slots ? work_to_pixel(p, slot) : -1;
    if (gid < 0) return;
    const uint32_t key = path_key(p.seed, (uint32_t)gid, 0u);
    const float4 ra = it.cur_a[gid], rb = it.cur_b[gid];
    uint32_t ctr = it.cur_d[gid].x;
    const int2 h = it.hit[gid];
    Ray r; r.o = mk3(ra.x, ra.y, ra.z); r.d = mk3(rb.x, rb.y, rb.z);
    f3 color = mk3(0, 0, 0), weight = mk3(1, 1, 1);
    const bool go_on = shade_bounce(p, h.x, __int_as_float(h.y), r, key, ctr, 0, color, weight);
    const bool finished = !go_on || 1 >= p.max_depth;
    const float4 a = make_float4(r.o.x, r.o.y, r.o.z, weight.x), b = make_float4(r.d.x, r.d.y, r.d.z, weight.y);
    const float4 c = make_float4(color.x, color.y, color.z, weight.z);
    it.c_a[gid] = a; it.c_b[gid] = b; it.c_c[gid] = c; it.c_d[gid] = make_uint2(ctr, finished ? 1u : 0u);
    it.cur_a[gid] = a; it.cur_b[gid] = b; it.cur_c[gid] = c; it.cur_d[gid] = make_uint2(ctr, 1u);
    it.completed[gid] = 0;
    it.contrib[gid] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

// Shading half of hpt_bounce.
__global__ void __launch_bounds__(256) iter_bounce_shade_kernel(const RenderParams p, const IterState it) {
    const unsigned slot = blockIdx.x * blockDim.x + threadIdx.x;
    const int gid = slot < p.n_slots ? work_to_pixel(p, slot) : -1;
    if (gid < 0) return;
    const uint2 cd = it.c_d[gid];
    float4 c = it.cur_c[gid];
    f3 color = mk3(c.x, c.y, c.z);
    bool ended = cd.y != 0u;                     // the cached bounce already ended the path: one path per call
    if (!ended) {
        const float4 a = it.cur_a[gid], b = it.cur_b[gid];
        const uint2 d = it.cur_d[gid];
        const int2 h = it.hit[gid];
        const uint32_t key = path_key(p.seed, (uint32_t)gid, (uint32_t)it.completed[gid]);
        uint32_t ctr = d.x;
        int bounce = (int)d.y;
        Ray r; r.o = mk3(a.x, a.y, a.z); r.d = mk3(b.x, b.y, b.z);
        f3 weight = mk3(a.w, b.w, c.w);
        const bool go_on = shade_bounce(p, h.x, __int_as_float(h.y), r, key, ctr, bounce, color, weight);
        bounce++;
        ended = !go_on || bounce >= p.max_depth;
        if (!ended) {
            it.cur_a[gid] = make_float4(r.o.x, r.o.y, r.o.z, weight.x);
            it.cur_b[gid] = make_float4(r.d.x, r.d.y, r.d.z, weight.y);
            it.cur_c[gid] = make_float4(color.x, color.y, color.z, weight.z);
            it.cur_d[gid] = make_uint2(ctr, (unsigned)bounce);
        }
    }
    if (ended) {
        it.contrib[gid] = make_float4(color.x, color.y, color.z, 1.0f);
        it.completed[gid] = it.completed[gid] + 1;
        it.cur_a[gid] = it.c_a[gid]; it.cur_b[gid] = it.c_b[gid]; it.cur_c[gid] = it.c_c[gid];
        it.cur_d[gid] = make_uint2(cd.x, 1u);
    }
}

// image = (1/n)*contribution + ((n-1)/n)*image with n = the pixel's completed paths (pathtracer.cpp:200-217
// per pixel, as reconstruct_kernel.cl:25-36 intends), for pixels whose path ended in the last hpt_bounce.
__global__ void __launch_bounds__(256) iter_reconstruct_kernel(const RenderParams p, const IterState it, float *__restrict__ image) {
    const unsigned slot = blockIdx.x * blockDim.x + threadIdx.x;
    const int gid = slot < p.n_slots ? work_to_pixel(p, slot) : -1;
    if (gid < 0) return;
    const float4 c = it.contrib[gid];
    if (c.w == 0.0f) return;
    float *px = image + 3 * (size_t)gid;
    const f3 mean = fold_sample(mk3(px[0], px[1], px[2]), mk3(c.x, c.y, c.z), it.completed[gid], false);
    px[0] = mean.x; px[1] = mean.y; px[2] = mean.z;
    it.contrib[gid] = make_float4(c.x, c.y, c.z, 0.0f);
}

// running mean of Pathtracer::render(), pathtracer.cpp:200-217 (the committed
// reconstruct_kernel.cl swaps G/B and stops at gid 1000: not reproduced)
__global__ void reconstruct_kernel(float *__restrict__ image, const float *__restrict__ sample, long n_floats, int iterations) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_floats) return;
    if (iterations > 1) {
        float i1 = 1.0f / (float)iterations;
        float i2 = (float)(iterations - 1) / (float)iterations;
        image[i] = (i1 * sample[i]) + (i2 * image[i]);
    } else if (iterations == 1) image[i] = sample[i];
}

// png_writer.cpp:7-17: (uint8)(255 * max(0, min(c, 1))), truncation, NaN -> 0
__global__ void tonemap_kernel(const float *__restrict__ image, uint8_t *__restrict__ out, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float c = image[i];
    float mn = (1.0f < c) ? 1.0f : c;
    float cl = (0.0f < mn) ? mn : 0.0f;
    out[i] = (uint8_t)(255.0f * cl);
}

// shard exchange: slab[(local_tile*64 + k)*3 + c] <-> image pixel
__global__ void pack_tiles_kernel(const RenderParams p, const float *__restrict__ image, float *__restrict__ slab, unsigned n_slab_px) {
    unsigned w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_slab_px) return;
    int gid = work_to_pixel(p, w);
    float r = 0, g = 0, b = 0;
    if (gid >= 0) { r = image[3 * (size_t)gid]; g = image[3 * (size_t)gid + 1]; b = image[3 * (size_t)gid + 2]; }
    slab[3 * (size_t)w] = r; slab[3 * (size_t)w + 1] = g; slab[3 * (size_t)w + 2] = b;
}
__global__ void unpack_tiles_kernel(RenderParams p, const float *__restrict__ slabs, float *__restrict__ image, unsigned n_slab_px) {
    unsigned w = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned total = n_slab_px * (unsigned)p.world;
    if (w >= total) return;
    p.rank = (int)(w / n_slab_px);
    int gid = work_to_pixel(p, w % n_slab_px);
    if (gid < 0) return;
    image[3 * (size_t)gid] = slabs[3 * (size_t)w];
    image[3 * (size_t)gid + 1] = slabs[3 * (size_t)w + 1];
    image[3 * (size_t)gid + 2] = slabs[3 * (size_t)w + 2];
}
