/* oracle/pt_oracle.c — TEST INFRASTRUCTURE ONLY (see pt_oracle.h header).
 *
 * Scalar CPU restatement of the reference path tracer's arithmetic, in the
 * reference's own control structure: separate bbox/info node arrays, test the
 * node's own box when it is visited, stack of node indices.  The HIP product
 * uses a different node layout and scheduler; this file is what it is checked
 * against.  Build: gcc -O2 -ffp-contract=off (no FMA contraction, no fast-math).
 *
 * Numerics contract (shared with the HIP path, stated once here):
 *  - fp32 IEEE + - * / sqrt only, evaluated in the order written; dot(a,b) =
 *    (ax*bx + ay*by) + az*bz; normalize(v) = v * (1/sqrt(dot(v,v))).  OpenCL
 *    leaves the precision/order of dot/normalize/cross implementation-defined,
 *    so there is nothing more exact in the reference to match.
 *  - pow(x,2) = x*x and pow(x,5) = ((x*x)*(x*x))*x (refract/schlick,
 *    ray_kernel.cl:631, 654-663).
 *  - sin/cos(phi), phi in [0,2pi], by the Cody-Waite + minimax polynomial in
 *    orc_sincos() (<= 2 ulp; OpenCL allows 4 ulp).
 *  - RNG: counter-based hash keyed (seed, pixel, sample, draw counter) in the
 *    reference's draw order, because the reference stream (ray_kernel.cl:259-266,
 *    seeds from unseeded libc rand(), pathtracer.cpp:79-82) is unpinnable and
 *    cannot be sharded.
 */
#include "pt_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define EPSILON 0.001f                       /* ray_kernel.cl:4 */
static const float ONE_THIRD_RT = 0.57735026919f;      /* ray_kernel.cl:12 */
static const float BACKGROUND[3] = {0.8f, 0.9f, 0.95f}; /* ray_kernel.cl:13 */

typedef struct { float x, y, z; } v3;

static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 V4(hpt_float4 f) { return V(f.x, f.y, f.z); }
static inline v3 add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 scale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 neg(v3 a) { return V(-a.x, -a.y, -a.z); }
static inline float dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline v3 cross(v3 a, v3 b) {
    return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline v3 normalize(v3 a) { float inv = 1.0f / sqrtf(dot(a, a)); return scale(a, inv); }

/* ---- RNG (replaces rand(), ray_kernel.cl:259-266; draw order kept) ---- */
static inline uint32_t fmix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; return h;
}
static inline uint32_t path_key(uint32_t seed, uint32_t pixel, uint32_t sample) {
    return fmix32(fmix32(seed + pixel * 0x9E3779B1u) + sample * 0x85EBCA77u);
}
static inline float draw(uint32_t key, uint32_t *ctr) {
    uint32_t h = fmix32(key + (*ctr) * 0xC2B2AE3Du);
    (*ctr)++;
    return (float)(h >> 8) * 5.9604644775390625e-8f;   /* 2^-24, in [0,1) */
}
float orc_rand(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t ctr) {
    uint32_t k = path_key(seed, pixel, sample);
    return draw(k, &ctr);
}

/* ---- sin/cos for phi in [0, 2pi] ---- */
void orc_sincos(float x, float *s, float *c) {
    int j = (int)(x * 1.27323954473516f);        /* x * 4/pi */
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
    switch ((j >> 1) & 3) {
        case 0:  *s = ps;  *c = pc;  break;
        case 1:  *s = pc;  *c = -ps; break;
        case 2:  *s = -ps; *c = -pc; break;
        default: *s = -pc; *c = ps;  break;
    }
}

/* ---- camera ray: initial_ray_kernel.cl:859-883 == pathtracer_kernel.cl:912-937 ---- */
static void camera_ray(const orc_scene *s, int gid, int iteration, uint32_t key, uint32_t *ctr,
                       v3 *o, v3 *d) {
    int width = s->width, height = s->height;
    int n = 6;
    float x = fmodf((float)gid, (float)width);
    float y = gid / (float)width;                 /* not floored: row + col/width */
    x = (x / (float)width) - 0.5f;
    y = (y / (float)height) - 0.5f;
    float left_limit = x - 1 / (float)(width + width);
    float top_limit = y - 1 / (float)(height + height);
    int cell = iteration % (n * n);
    float inv_n_w = 1 / (float)(n * width);
    float inv_n_h = 1 / (float)(n * height);
    float r0 = draw(key, ctr);
    float r1 = draw(key, ctr);
    x = left_limit + ((cell % n) + r0) * inv_n_w;
    y = top_limit + ((cell / n) + r1) * inv_n_h;
    v3 u = V4(s->camera.u), v = V4(s->camera.v), nn = V4(s->camera.n);
    *o = V4(s->camera.eye);
    *d = normalize(add(add(scale(u, x), scale(v, y)), neg(nn)));
}
void orc_camera_ray(const orc_scene *s, int gid, int iteration, float o[3], float d[3]) {
    uint32_t key = path_key(s->cfg.seed, (uint32_t)gid, (uint32_t)iteration), ctr = 0;
    v3 oo, dd;
    camera_ray(s, gid, iteration, key, &ctr, &oo, &dd);
    o[0] = oo.x; o[1] = oo.y; o[2] = oo.z; d[0] = dd.x; d[1] = dd.y; d[2] = dd.z;
}

/* ---- slab test: ray_kernel.cl:321-377 ---- */
static int intersect_box(v3 o, v3 minb, v3 maxb, v3 invDir, const int dirIsNeg[3], float t) {
    float tmin, tmax, tymin, tymax, tzmin, tzmax;
    if (dirIsNeg[0]) { tmin = (maxb.x - o.x) * invDir.x; tmax = (minb.x - o.x) * invDir.x; }
    else             { tmin = (minb.x - o.x) * invDir.x; tmax = (maxb.x - o.x) * invDir.x; }
    if (dirIsNeg[1]) { tymin = (maxb.y - o.y) * invDir.y; tymax = (minb.y - o.y) * invDir.y; }
    else             { tymin = (minb.y - o.y) * invDir.y; tymax = (maxb.y - o.y) * invDir.y; }
    if ((tmin > tymax) || (tymin > tmax)) return 0;
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    if (dirIsNeg[2]) { tzmin = (maxb.z - o.z) * invDir.z; tzmax = (minb.z - o.z) * invDir.z; }
    else             { tzmin = (minb.z - o.z) * invDir.z; tzmax = (maxb.z - o.z) * invDir.z; }
    if ((tmin > tzmax) || (tzmin > tmax)) return 0;
    if (tzmin > tmin) tmin = tzmin;
    if (tzmax < tmax) tmax = tzmax;
    return (tmin < t) && (tmax > EPSILON);
}

/* ---- Moller-Trumbore: ray_kernel.cl:467-520 (hit id recorded instead of the Triangle copy) ---- */
static int intersect_tri(v3 o, v3 d, const hpt_triangle *tri, float *t) {
    v3 edge1 = V4(tri->e1), edge2 = V4(tri->e2);
    v3 pvec = cross(d, edge2);
    float det = dot(edge1, pvec);
    if (fabsf(det) < EPSILON) return 0;
    float invDet = 1 / det;
    v3 tvec = sub(o, V4(tri->v1));
    float u = dot(tvec, pvec) * invDet;
    if (u < 0 || u > 1) return 0;
    v3 qvec = cross(tvec, edge1);
    float v = dot(d, qvec) * invDet;
    if (v < 0 || u + v > 1) return 0;
    float tt = dot(edge2, qvec) * invDet;
    if (tt < *t && tt > 0.0f) { *t = tt; return 1; }
    return 0;
}

/* ---- sphere / plane: ray_kernel.cl:524-602 ---- */
static int intersect_prim(v3 o, v3 d, const hpt_primitive *p, float *t) {
    if (p->type.x == 0) {
        v3 oc = sub(o, V4(p->center));
        float a = dot(d, d);
        float b = 2 * dot(d, oc);
        float c = dot(oc, oc) - (p->radius * p->radius);
        float discriminant = (b * b) - (4 * a * c);
        float tt = 1000.0f;
        if (discriminant < 0.0f) return 0;
        else if (discriminant >= 0 && discriminant < EPSILON) tt = -b / (2 * a);
        else {
            discriminant = sqrtf(discriminant);
            float t0 = (-b - discriminant) / (2 * a);
            if (t0 > EPSILON) tt = t0;
            else tt = (-b + discriminant) / (2 * a);
        }
        if (tt < *t && tt > 0.0f) { *t = tt; return 1; }
    } else if (p->type.x == 1) {
        v3 pn = V4(p->plane_normal);
        float denom = dot(pn, d);
        if (fabsf(denom) > EPSILON) {
            v3 between = sub(V4(p->center), o);
            float tt = dot(between, pn) / denom;
            if (tt < *t && tt >= 0) { *t = tt; return 1; }
        }
    }
    return 0;
}

int orc_intersect_box(const float o[3], const float d[3], const float lo[3], const float hi[3], float t) {
    v3 inv = V(1.f / d[0], 1.f / d[1], 1.f / d[2]);
    int neg_[3] = {inv.x < 0, inv.y < 0, inv.z < 0};
    return intersect_box(V(o[0], o[1], o[2]), V(lo[0], lo[1], lo[2]), V(hi[0], hi[1], hi[2]), inv, neg_, t);
}
int orc_intersect_tri(const float o[3], const float d[3], const hpt_triangle *tri, float *t) {
    return intersect_tri(V(o[0], o[1], o[2]), V(d[0], d[1], d[2]), tri, t);
}
int orc_intersect_prim(const float o[3], const float d[3], const hpt_primitive *p, float *t) {
    return intersect_prim(V(o[0], o[1], o[2]), V(d[0], d[1], d[2]), p, t);
}

/* optional instrumentation for layout studies: visits per tree depth (not thread-safe: use n_threads=1) */
static const int32_t *g_depth_table = NULL;
static uint64_t g_depth_hist[64];
void orc_set_depth_table(const int32_t *table) { g_depth_table = table; memset(g_depth_hist, 0, sizeof g_depth_hist); }
void orc_get_depth_hist(uint64_t *out) { memcpy(out, g_depth_hist, sizeof g_depth_hist); }

/* ---- BVH traversal: ray_kernel.cl:382-464 ---- */
static int intersect_tri_bvh(const orc_scene *s, v3 o, v3 d, float *t, int *tri_id,
                             uint64_t *n_nodes, uint64_t *n_tris) {
    int current = 0, nodeNum = 0;
    int todo[64];
    int hit = 0;
    if (s->n_nodes <= 0) return 0;
    v3 invDir = V(1.f / d.x, 1.f / d.y, 1.f / d.z);
    int dirIsNeg[3] = {invDir.x < 0, invDir.y < 0, invDir.z < 0};
    for (;;) {
        const hpt_bvh_bbox *bb = &s->bbox[nodeNum];
        const hpt_bvh_info *n = &s->info[nodeNum];
        (*n_nodes)++;
        if (g_depth_table) g_depth_hist[g_depth_table[nodeNum] & 63]++;
        if (intersect_box(o, V(bb->minboundsx, bb->minboundsy, bb->minboundsz),
                          V(bb->maxboundsx, bb->maxboundsy, bb->maxboundsz), invDir, dirIsNeg, *t)) {
            if (n->nPrimitives > 0) {
                for (int i = 0; i < n->nPrimitives; i++) {
                    (*n_tris)++;
                    if (intersect_tri(o, d, &s->triangles[n->offset + i], t)) {
                        hit = 1; *tri_id = n->offset + i;
                    }
                }
                if (current == 0) break;
                nodeNum = todo[--current];
            } else {
                if (dirIsNeg[n->axis]) { todo[current++] = nodeNum + 1; nodeNum = n->secondChildOffset; }
                else                   { todo[current++] = n->secondChildOffset; nodeNum++; }
            }
        } else {
            if (current == 0) break;
            nodeNum = todo[--current];
        }
    }
    return hit;
}

/* closest hit = primitives loop then BVH, ray_kernel.cl:726-745 */
static int closest_hit(const orc_scene *s, v3 o, v3 d, float *t, uint64_t *nn, uint64_t *nt) {
    int id = HPT_HIT_MISS;
    *t = 1000;
    for (int i = 0; i < s->n_primitives; i++)
        if (intersect_prim(o, d, &s->primitives[i], t)) id = HPT_HIT_PRIMITIVE(i);
    int tri = -1;
    if (intersect_tri_bvh(s, o, d, t, &tri, nn, nt)) id = tri;
    return id;
}

/* ---- material flags: ray_kernel.cl:244-256 ---- */
static int islight(const hpt_material *m) { return (m->emissive.y > 0 || m->emissive.z > 0 || m->emissive.w > 0); }
static int isspecular(const hpt_material *m) {
    return (m->specular.x > EPSILON || m->specular.y > EPSILON || m->specular.z > EPSILON || m->specular.w > EPSILON);
}
static int isrefractive(const hpt_material *m) { return m->refractive.w == 1; }

static inline float pow2f(float x) { return x * x; }
static inline float pow5f(float x) { float x2 = x * x; return (x2 * x2) * x; }

/* ray_kernel.cl:269-302 */
static v3 random_sample_hemisphere(v3 normal, uint32_t key, uint32_t *ctr) {
    float z = draw(key, ctr);
    float r2 = draw(key, ctr);
    float r = sqrtf(z);
    float phi = 6.283185307179586f * r2;        /* 2 * M_PI_F * r2 */
    float sn, cs;
    orc_sincos(phi, &sn, &cs);
    float x = cs * r;
    float y = sn * r;
    z = sqrtf(1 - z);
    v3 most_perpendicular;
    if (fabsf(normal.x) < ONE_THIRD_RT) most_perpendicular = V(1, 0, 0);
    else if (fabsf(normal.y) < ONE_THIRD_RT) most_perpendicular = V(0, 1, 0);
    else most_perpendicular = V(0, 0, 1);
    v3 u = normalize(cross(most_perpendicular, normal));
    v3 v = cross(normal, u);
    return add(add(scale(u, x), scale(v, y)), scale(normal, z));
}

/* ray_kernel.cl:615-621 */
static v3 reflect_dir(v3 n, v3 d) {
    return normalize(sub(d, scale(n, 2 * dot(n, d))));
}
/* ray_kernel.cl:625-642; returns 0 on total internal reflection */
static int refract_dir(v3 D, v3 N, float index, v3 *out) {
    float dn = dot(D, N);
    float a = 1 - (pow2f(index) * (1 - pow2f(dn)));
    if (a < 0.0f) return 0;
    v3 tt = sub(scale(sub(D, scale(N, dn)), index), scale(N, sqrtf(a)));
    *out = normalize(tt);
    return 1;
}
/* ray_kernel.cl:650-668 */
static float schlick_fresnel(v3 n, v3 d, const hpt_material *m) {
    float n2 = m->refractive.x;
    float F0 = pow2f((1 - n2) / (1 + n2));
    if (dot(d, n) < 0.0f) return F0 + (1 - F0) * pow5f(1 - dot(n, neg(d)));
    return F0 + (1 - F0) * pow5f(1 - dot(n, d));
}
/* ray_kernel.cl:670-706 */
static v3 refractive_dir(v3 N, v3 d, const hpt_material *m) {
    float n = m->refractive.x;
    v3 t;
    int ok;
    if (dot(d, N) < 0.0f) ok = refract_dir(d, N, 1 / n, &t);
    else ok = refract_dir(d, neg(N), n, &t);
    if (ok) return t;
    float ndoti = dot(N, neg(d));
    return add(scale(N, ndoti + ndoti), d);
}

/* ---- one bounce of the loop body pathtracer_kernel.cl:759-849 after the closest hit (id, t) of ray (o, d):
 * returns 1 when the path goes on with the new ray left in (o, d), 0 when it ended (light, miss, roulette) ---- */
static int shade_bounce(const orc_scene *s, int id, float t, v3 *o, v3 *d, uint32_t key, uint32_t *ctr, int b,
                        v3 *color, v3 *weight) {
    v3 no, nd;
    if (id != HPT_HIT_MISS) {
        hpt_material m;
        v3 point = add(*o, scale(*d, t));          /* point_at, ray_kernel.cl:82-85 */
        v3 normal;
        if (id >= 0) {
            const hpt_triangle *tri = &s->triangles[id];
            m = s->materials[0];
            normal = normalize(cross(V4(tri->e1), V4(tri->e2)));
            if (dot(normal, *d) > 0.0f) normal = neg(normal);
        } else {
            const hpt_primitive *p = &s->primitives[-id - 2];
            m.refractive = p->refractive; m.specular = p->specular;
            m.diffuse = p->diffuse; m.emissive = p->emissive;
            if (p->type.x == 0) normal = normalize(sub(point, V4(p->center)));
            else normal = normalize(V4(p->plane_normal));
        }
        if (islight(&m)) {
            *color = add(*color, mul(*weight, V4(m.emissive)));
            return 0;
        }
        int fresnel = draw(key, ctr) < schlick_fresnel(normal, *d, &m);
        if (fresnel && isspecular(&m)) {
            nd = reflect_dir(normal, *d);
            no = add(point, scale(nd, EPSILON));
        } else if (isrefractive(&m)) {
            if (fresnel) nd = reflect_dir(normal, *d);
            else nd = refractive_dir(normal, *d, &m);
            no = add(point, scale(nd, EPSILON));
        } else {
            *weight = mul(*weight, V4(m.diffuse));
            nd = random_sample_hemisphere(normal, key, ctr);
            no = add(point, scale(nd, EPSILON));
            *weight = scale(*weight, dot(normal, nd) * s->cfg.diffuse_scale);
        }
    } else {
        *color = add(*color, mul(*weight, V(BACKGROUND[0], BACKGROUND[1], BACKGROUND[2])));
        return 0;
    }
    if (b > s->cfg.rr_start) {                    /* pathtracer_kernel.cl:836-845 */
        float m = (weight->x > weight->y) ? weight->x : weight->y;
        m = (weight->z > m) ? weight->z : m;
        float pterm = draw(key, ctr);
        if (m < pterm) return 0;
        *weight = V(weight->x / m, weight->y / m, weight->z / m);
    }
    *o = no; *d = nd;
    return 1;
}

/* ---- one radiance sample: pathtracer_kernel.cl:748-852 (loop form) ---- */
static void path_sample(const orc_scene *s, int gid, int iteration, float rgb[3], orc_counters *c) {
    uint32_t key = path_key(s->cfg.seed, (uint32_t)gid, (uint32_t)iteration), ctr = 0;
    v3 o, d;
    camera_ray(s, gid, iteration, key, &ctr, &o, &d);
    v3 color = V(0, 0, 0), weight = V(1, 1, 1);
    c->paths++;
    for (int b = 0; b < s->cfg.max_depth; b++) {
        float t;
        uint64_t nn = 0, nt = 0;
        int id = closest_hit(s, o, d, &t, &nn, &nt);
        if (b == 0) { c->rays_primary++; c->nodes_primary += nn; c->tris_primary += nt; }
        else        { c->rays_bounce++;  c->nodes_bounce += nn;  c->tris_bounce += nt; }
        if (!shade_bounce(s, id, t, &o, &d, key, &ctr, b, &color, &weight)) break;
    }
    rgb[0] = color.x; rgb[1] = color.y; rgb[2] = color.z;
}

/* ---- the iterative pipeline with its INTENDED semantics (initial_ray_kernel.cl:802-902,
 * ray_kernel.cl:817-929, reconstruct_kernel.cl:10-37): cache bounce 0 of every pixel; each call advances every
 * pixel's current path by one bounce; a path that ends leaves its colour as the pixel's contribution, counts as
 * completed, and the pixel restarts from the cache; reconstruct folds fresh contributions into the per-pixel
 * running mean.  Path n of a pixel draws from key(seed, pixel, n), counter continuing after the cached bounce. ---- */
void orc_iter_first_bounce(const orc_scene *s, orc_iter_state *st) {
    const int npx = s->width * s->height;
    for (int gid = 0; gid < npx; gid++) {
        orc_iter_pixel *q = &st->px[gid];
        uint32_t key = path_key(s->cfg.seed, (uint32_t)gid, 0u), ctr = 0;
        v3 o, d;
        camera_ray(s, gid, 0, key, &ctr, &o, &d);
        v3 color = V(0, 0, 0), weight = V(1, 1, 1);
        float t;
        uint64_t nn = 0, nt = 0;
        int id = closest_hit(s, o, d, &t, &nn, &nt);
        st->rays_primary++;
        int go_on = shade_bounce(s, id, t, &o, &d, key, &ctr, 0, &color, &weight);
        q->c_finished = (!go_on || 1 >= s->cfg.max_depth);
        q->c_o[0] = o.x; q->c_o[1] = o.y; q->c_o[2] = o.z; q->c_d[0] = d.x; q->c_d[1] = d.y; q->c_d[2] = d.z;
        q->c_weight[0] = weight.x; q->c_weight[1] = weight.y; q->c_weight[2] = weight.z;
        q->c_color[0] = color.x; q->c_color[1] = color.y; q->c_color[2] = color.z;
        q->c_ctr = ctr;
        memcpy(q->o, q->c_o, sizeof q->o); memcpy(q->d, q->c_d, sizeof q->d);
        memcpy(q->weight, q->c_weight, sizeof q->weight); memcpy(q->color, q->c_color, sizeof q->color);
        q->ctr = ctr; q->bounce = 1; q->completed = 0; q->fresh = 0;
        q->contribution[0] = q->contribution[1] = q->contribution[2] = 0.0f;
    }
}

void orc_iter_bounce(const orc_scene *s, orc_iter_state *st) {
    const int npx = s->width * s->height;
    for (int gid = 0; gid < npx; gid++) {
        orc_iter_pixel *q = &st->px[gid];
        int ended;
        v3 color = V(q->color[0], q->color[1], q->color[2]);
        if (q->c_finished) ended = 1;              /* the cached bounce already ended the path: one path per call */
        else {
            uint32_t key = path_key(s->cfg.seed, (uint32_t)gid, (uint32_t)q->completed), ctr = q->ctr;
            v3 o = V(q->o[0], q->o[1], q->o[2]), d = V(q->d[0], q->d[1], q->d[2]);
            v3 weight = V(q->weight[0], q->weight[1], q->weight[2]);
            float t;
            uint64_t nn = 0, nt = 0;
            int id = closest_hit(s, o, d, &t, &nn, &nt);
            st->rays_bounce++;
            int go_on = shade_bounce(s, id, t, &o, &d, key, &ctr, q->bounce, &color, &weight);
            q->bounce++;
            ended = (!go_on || q->bounce >= s->cfg.max_depth);
            if (!ended) {
                q->o[0] = o.x; q->o[1] = o.y; q->o[2] = o.z; q->d[0] = d.x; q->d[1] = d.y; q->d[2] = d.z;
                q->weight[0] = weight.x; q->weight[1] = weight.y; q->weight[2] = weight.z;
                q->color[0] = color.x; q->color[1] = color.y; q->color[2] = color.z;
                q->ctr = ctr;
            }
        }
        if (ended) {
            q->contribution[0] = color.x; q->contribution[1] = color.y; q->contribution[2] = color.z;
            q->completed++;
            q->fresh = 1;
            memcpy(q->o, q->c_o, sizeof q->o); memcpy(q->d, q->c_d, sizeof q->d);
            memcpy(q->weight, q->c_weight, sizeof q->weight); memcpy(q->color, q->c_color, sizeof q->color);
            q->ctr = q->c_ctr; q->bounce = 1;
        }
    }
}

void orc_iter_reconstruct(const orc_scene *s, orc_iter_state *st, float *image) {
    const int npx = s->width * s->height;
    for (int gid = 0; gid < npx; gid++) {
        orc_iter_pixel *q = &st->px[gid];
        if (!q->fresh) continue;
        orc_accumulate(image + 3 * (int64_t)gid, q->contribution, 3, q->completed);
        q->fresh = 0;
    }
}

static void counters_add(orc_counters *a, const orc_counters *b) {
    a->rays_primary += b->rays_primary; a->rays_bounce += b->rays_bounce; a->paths += b->paths;
    a->nodes_primary += b->nodes_primary; a->nodes_bounce += b->nodes_bounce;
    a->tris_primary += b->tris_primary; a->tris_bounce += b->tris_bounce;
}

void orc_primary_hits(const orc_scene *s, int iteration, int px_begin, int px_end,
                      int32_t *hit_id, float *t, orc_counters *c) {
    orc_counters loc; memset(&loc, 0, sizeof loc);
    for (int gid = px_begin; gid < px_end; gid++) {
        uint32_t key = path_key(s->cfg.seed, (uint32_t)gid, (uint32_t)iteration), ctr = 0;
        v3 o, d;
        camera_ray(s, gid, iteration, key, &ctr, &o, &d);
        float tt;
        hit_id[gid - px_begin] = closest_hit(s, o, d, &tt, &loc.nodes_primary, &loc.tris_primary);
        t[gid - px_begin] = tt;
        loc.rays_primary++;
    }
    if (c) counters_add(c, &loc);
}

void orc_kernel_execute(const orc_scene *s, int iteration, int px_begin, int px_end,
                        float *rgb, orc_counters *c, int n_threads) {
    orc_counters total; memset(&total, 0, sizeof total);
#ifdef _OPENMP
    if (n_threads <= 0) n_threads = omp_get_max_threads();
#pragma omp parallel num_threads(n_threads)
    {
        orc_counters loc; memset(&loc, 0, sizeof loc);
#pragma omp for schedule(dynamic, 256)
        for (int gid = px_begin; gid < px_end; gid++) path_sample(s, gid, iteration, rgb + 3 * (int64_t)gid, &loc);
#pragma omp critical
        counters_add(&total, &loc);
    }
#else
    (void)n_threads;
    for (int gid = px_begin; gid < px_end; gid++) path_sample(s, gid, iteration, rgb + 3 * (int64_t)gid, &total);
#endif
    if (c) counters_add(c, &total);
}

/* pathtracer.cpp:200-217 */
void orc_accumulate(float *image, const float *sample, int64_t n_floats, int iterations) {
    if (iterations > 1) {
        float i1 = 1 / (float)iterations;
        float i2 = (float)(iterations - 1) / (float)iterations;
        for (int64_t i = 0; i < n_floats; i++) image[i] = (i1 * sample[i]) + (i2 * image[i]);
    } else if (iterations == 1) {
        for (int64_t i = 0; i < n_floats; i++) image[i] = sample[i];
    }
}

void orc_render(const orc_scene *s, int spp_begin, int spp_end, int px_begin, int px_end,
                float *image, orc_counters *c, int n_threads) {
    int64_t npx = (int64_t)s->width * s->height;
    float *sample = (float *)calloc((size_t)npx * 3, sizeof(float));
    for (int it = spp_begin; it < spp_end; it++) {
        orc_kernel_execute(s, it, px_begin, px_end, sample, c, n_threads);
        orc_accumulate(image + 3 * (int64_t)px_begin, sample + 3 * (int64_t)px_begin,
                       3 * (int64_t)(px_end - px_begin), it + 1);
    }
    free(sample);
}

/* png_writer.cpp:7-17: clip = std::max(lower, std::min(n, upper)) */
void orc_tonemap(const float *image, uint8_t *rgb8, int64_t n_floats) {
    for (int64_t i = 0; i < n_floats; i++) {
        float n = image[i];
        float mn = (1.0f < n) ? 1.0f : n;         /* std::min(n, upper) */
        float cl = (0.0f < mn) ? mn : 0.0f;       /* std::max(lower, mn): NaN -> 0 */
        rgb8[i] = (uint8_t)(255 * cl);
    }
}

#include <stddef.h>
void orc_scene_layout(int *out) {   /* lets the ctypes binding check its mirror of orc_scene */
    out[0] = (int)sizeof(orc_scene); out[1] = (int)offsetof(orc_scene, camera);
    out[2] = (int)offsetof(orc_scene, width); out[3] = (int)offsetof(orc_scene, cfg);
}

/* ======================= BVH builder: bvh.cpp ======================= */
typedef struct { float mn[3], mx[3]; } bbox_t;
typedef struct {            /* VectorTriangle, common.hpp:83-106 */
    float centroid[3];
    bbox_t bbox;
    int32_t src;
} vtri_t;
typedef struct tnode {      /* TreeNode, bvh.hpp:17-25 */
    bbox_t bbox;
    struct tnode *children[2];
    int axis, index, nPrimitives;
} tnode;

static bbox_t combine(bbox_t a, bbox_t b) {   /* common.hpp:69-74 (std::min/max) */
    bbox_t r;
    for (int k = 0; k < 3; k++) {
        r.mn[k] = (b.mn[k] < a.mn[k]) ? b.mn[k] : a.mn[k];
        r.mx[k] = (a.mx[k] < b.mx[k]) ? b.mx[k] : a.mx[k];
    }
    return r;
}
static bbox_t combine_pt(bbox_t a, const float p[3]) {  /* common.hpp:76-81 */
    bbox_t r;
    for (int k = 0; k < 3; k++) {
        r.mn[k] = (p[k] < a.mn[k]) ? p[k] : a.mn[k];
        r.mx[k] = (a.mx[k] < p[k]) ? p[k] : a.mx[k];
    }
    return r;
}
static int longest_axis(const bbox_t *b) {   /* common.hpp:39-50 */
    int axis = 0;
    if ((b->mx[1] - b->mn[1]) > (b->mx[0] - b->mn[0])) axis = 1;
    if ((b->mx[2] - b->mn[2]) > (b->mx[axis] - b->mn[axis])) axis = 2;
    return axis;
}
static float surface_area(const bbox_t *b) {  /* common.hpp:52-56 */
    float v0 = b->mx[0] - b->mn[0], v1 = b->mx[1] - b->mn[1], v2 = b->mx[2] - b->mn[2];
    return 2.0f * (v0 * v1 + v0 * v2 + v1 * v2);
}

typedef struct {
    vtri_t *vt; int n;
    int32_t *ordered; int n_ordered;
    int totalNodes;
} builder_t;

#define MAX_TRIANGLES_LEAF 24   /* bvh.hpp:8 */
#define NUM_BUCKETS 12          /* bvh.hpp:9 */

static void create_leaf(builder_t *B, tnode *node, bbox_t bounds, int start, int end) {  /* bvh.cpp:126-139 */
    node->bbox = bounds;
    node->index = B->n_ordered;
    node->nPrimitives = end - start;
    for (int i = start; i < end; i++) B->ordered[B->n_ordered++] = B->vt[i].src;
}

static tnode *rec_build(builder_t *B, int start, int end) {   /* bvh.cpp:141-336 */
    vtri_t *vt = B->vt;
    B->totalNodes++;
    tnode *node = (tnode *)calloc(1, sizeof(tnode));
    bbox_t bounds = vt[start].bbox;
    for (int i = start + 1; i < end; i++) bounds = combine(bounds, vt[i].bbox);
    int nprimitives = end - start;
    if (nprimitives == 1) { create_leaf(B, node, bounds, start, end); return node; }

    bbox_t cb; memcpy(cb.mn, vt[start].centroid, sizeof cb.mn); memcpy(cb.mx, vt[start].centroid, sizeof cb.mx);
    for (int i = start + 1; i < end; i++) cb = combine_pt(cb, vt[i].centroid);
    int axis = longest_axis(&cb);
    int mid = (start + end) / 2;

    if (cb.mx[axis] == cb.mn[axis]) {           /* bvh.cpp:174-193 */
        if (nprimitives <= MAX_TRIANGLES_LEAF) { create_leaf(B, node, bounds, start, end); return node; }
        node->children[0] = rec_build(B, start, mid);
        node->children[1] = rec_build(B, mid, end);
        node->bbox = combine(node->children[0]->bbox, node->children[1]->bbox);
        node->axis = axis; node->nPrimitives = 0;
        return node;
    }

    /* "SAH" bucketing, bvh.cpp:208-232 */
    int count[NUM_BUCKETS]; memset(count, 0, sizeof count);
    int *bucket_of = (int *)malloc(sizeof(int) * (size_t)nprimitives);
    double span = cb.mx[axis] - cb.mn[axis];
    for (int i = start; i < end; i++) {
        int bucket = (int)((float)NUM_BUCKETS * (vt[i].centroid[axis] - cb.mn[axis]) / span);
        if (bucket == NUM_BUCKETS) bucket--;
        count[bucket]++;
        bucket_of[i - start] = bucket;
    }
    /* cost loop, bvh.cpp:235-271: b0/b1 are the bboxes of GLOBAL triangles 0 and i+1
     * (indexed by bucket number) and combine() results are discarded. */
    float cost[NUM_BUCKETS - 1];
    for (int i = 0; i < NUM_BUCKETS - 1; i++) {
        bbox_t b0 = vt[0].bbox, b1 = vt[i + 1].bbox;
        int count0 = 0, count1 = 0;
        for (int j = 0; j <= i; j++) count0 += count[j];
        for (int k = i + 1; k < NUM_BUCKETS; k++) count1 += count[k];
        cost[i] = 0.125f + (count0 * surface_area(&b0) + count1 * surface_area(&b1)) / surface_area(&bounds);
    }
    float bestCost = cost[0];
    int bestSplit = 0;
    for (int i = 1; i < NUM_BUCKETS - 1; i++)
        if (cost[i] < bestCost) { bestCost = cost[i]; bestSplit = i; }
    (void)bestSplit;

    if (nprimitives > MAX_TRIANGLES_LEAF || bestCost < nprimitives) {   /* bvh.cpp:285-316 */
        /* pre = buckets 0..bestSplit, post = the rest, each in bucket order then input
         * order, written back pre then post: the range ends up stably sorted by bucket. */
        vtri_t *tmp = (vtri_t *)malloc(sizeof(vtri_t) * (size_t)nprimitives);
        int w = 0;
        for (int k = 0; k < NUM_BUCKETS; k++)
            for (int i = 0; i < nprimitives; i++)
                if (bucket_of[i] == k) tmp[w++] = vt[start + i];
        memcpy(vt + start, tmp, sizeof(vtri_t) * (size_t)nprimitives);
        free(tmp);
    } else {
        free(bucket_of);
        create_leaf(B, node, bounds, start, end);
        return node;
    }
    free(bucket_of);
    node->children[0] = rec_build(B, start, mid);    /* still split at mid, bvh.cpp:325-326 */
    node->children[1] = rec_build(B, mid, end);
    node->bbox = combine(node->children[0]->bbox, node->children[1]->bbox);
    node->axis = axis; node->nPrimitives = 0;
    return node;
}

static int flatten(tnode *node, int *offset, hpt_bvh_bbox *bbox, hpt_bvh_info *info) {  /* bvh.cpp:338-390 */
    int k = (*offset)++;
    bbox[k].minboundsx = node->bbox.mn[0]; bbox[k].minboundsy = node->bbox.mn[1]; bbox[k].minboundsz = node->bbox.mn[2];
    bbox[k].maxboundsx = node->bbox.mx[0]; bbox[k].maxboundsy = node->bbox.mx[1]; bbox[k].maxboundsz = node->bbox.mx[2];
    memset(&info[k], 0, sizeof info[k]);   /* reference leaves the unused fields as malloc bytes */
    if (node->nPrimitives > 0) {
        info[k].offset = node->index;
        info[k].nPrimitives = node->nPrimitives;
    } else {
        info[k].axis = node->axis;
        info[k].nPrimitives = 0;
        flatten(node->children[0], offset, bbox, info);
        info[k].secondChildOffset = flatten(node->children[1], offset, bbox, info);
    }
    free(node);
    return k;
}

int orc_build_bvh(const float *verts, int n, hpt_bvh_bbox *bbox, hpt_bvh_info *info, int32_t *order) {
    if (n < NUM_BUCKETS) return -1;
    builder_t B;
    B.vt = (vtri_t *)malloc(sizeof(vtri_t) * (size_t)n);
    B.n = n; B.ordered = order; B.n_ordered = 0; B.totalNodes = 0;
    for (int i = 0; i < n; i++) {            /* VectorTriangle ctor, common.hpp:94-106 */
        const float *p = verts + 9 * (int64_t)i;
        vtri_t *v = &B.vt[i];
        for (int k = 0; k < 3; k++) {
            float a = p[k], b = p[3 + k], c = p[6 + k];
            float m1 = (b < a) ? b : a;  v->bbox.mn[k] = (c < m1) ? c : m1;
            float M1 = (a < b) ? b : a;  v->bbox.mx[k] = (M1 < c) ? c : M1;
            v->centroid[k] = (float)((v->bbox.mn[k] + v->bbox.mx[k]) * 0.5);
        }
        v->src = i;
    }
    tnode *root = rec_build(&B, 0, n);
    int offset = 0;
    flatten(root, &offset, bbox, info);
    free(B.vt);
    return B.totalNodes;
}

void orc_make_triangles(const float *verts, const int32_t *order, int n, hpt_triangle *out) {  /* pathtracer.cpp:46-58 */
    for (int i = 0; i < n; i++) {
        const float *p = verts + 9 * (int64_t)order[i];
        out[i].v1.x = p[0]; out[i].v1.y = p[1]; out[i].v1.z = p[2]; out[i].v1.w = 0.0f;
        out[i].e1.x = p[3] - p[0]; out[i].e1.y = p[4] - p[1]; out[i].e1.z = p[5] - p[2]; out[i].e1.w = 0.0f;
        out[i].e2.x = p[6] - p[0]; out[i].e2.y = p[7] - p[1]; out[i].e2.z = p[8] - p[2]; out[i].e2.w = 0.0f;
    }
}

/* ======================= OBJ reader: obj_reader.cpp:6-84 ======================= */
int64_t orc_load_obj(const char *path, float *out_xyz, int64_t cap_vertices) {
    FILE *file = fopen(path, "r");
    if (file == NULL) return -1;
    float *tv = NULL; int64_t ntv = 0, captv = 0;
    int32_t *idx = NULL; int64_t nidx = 0, capidx = 0;
    for (;;) {
        char lineHeader[128];
        int res = fscanf(file, "%127s", lineHeader);
        if (res == EOF) break;
        if (strcmp(lineHeader, "v") == 0) {
            float v[3] = {0, 0, 0};
            if (fscanf(file, "%f %f %f\n", &v[0], &v[1], &v[2]) == EOF) { /* keep what was parsed */ }
            if (ntv == captv) { captv = captv ? captv * 2 : 1024; tv = (float *)realloc(tv, sizeof(float) * 3 * (size_t)captv); }
            tv[3 * ntv] = v[0]; tv[3 * ntv + 1] = v[1]; tv[3 * ntv + 2] = v[2]; ntv++;
        } else if (strcmp(lineHeader, "f") == 0) {
            int vi[3];
            int matches = fscanf(file, "%d %d %d", &vi[0], &vi[1], &vi[2]);
            if (matches != 3) { free(tv); free(idx); fclose(file); return -2; }
            if (nidx + 3 > capidx) { capidx = capidx ? capidx * 2 : 3072; idx = (int32_t *)realloc(idx, sizeof(int32_t) * (size_t)capidx); }
            idx[nidx++] = vi[0]; idx[nidx++] = vi[1]; idx[nidx++] = vi[2];
        }
    }
    fclose(file);
    int64_t written = 0;
    for (int64_t i = 0; i < nidx; i++) {
        int64_t k = (int64_t)idx[i] - 1;
        if (k < 0 || k >= ntv) { free(tv); free(idx); return -2; }   /* reference: out-of-range read (UB) */
        if (out_xyz && written < cap_vertices) {
            out_xyz[3 * written] = tv[3 * k]; out_xyz[3 * written + 1] = tv[3 * k + 1]; out_xyz[3 * written + 2] = tv[3 * k + 2];
        }
        written++;
    }
    free(tv); free(idx);
    return written;
}
