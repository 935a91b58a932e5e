/* scenes/scenes.c — procedural stand-in scenes for the BASELINE.json configs.
 *
 * The reference ships no scene assets, so tests and
 * bench.py generate geometry here.  Bench/test infrastructure: not part of
 * the render path and not part of the oracle.  Everything is integer hashing
 * and + - * / sqrt on doubles (own sin/cos series, no libm transcendentals),
 * built with -ffp-contract=off, so the same floats come out on every box.
 * Geometry is scaled to O(100) units: the reference culls |det| < 1e-3 and
 * starts t at 1000 (ray_kernel.cl:476, 726).
 * Axes: camera default sits at z=-200 looking +z; +y is DOWN in the image.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { double x, y, z; } d3;
typedef struct { float *v; int64_t n, cap; } mesh_t;   /* n triangles written; v may be NULL (count only) */

static void emit(mesh_t *m, d3 a, d3 b, d3 c) {
    if (m->v && m->n < m->cap) {
        float *p = m->v + 9 * m->n;
        p[0] = (float)a.x; p[1] = (float)a.y; p[2] = (float)a.z;
        p[3] = (float)b.x; p[4] = (float)b.y; p[5] = (float)b.z;
        p[6] = (float)c.x; p[7] = (float)c.y; p[8] = (float)c.z;
    }
    m->n++;
}
static d3 D(double x, double y, double z) { d3 r = {x, y, z}; return r; }
static d3 dadd(d3 a, d3 b) { return D(a.x + b.x, a.y + b.y, a.z + b.z); }
static d3 dscale(d3 a, double s) { return D(a.x * s, a.y * s, a.z * s); }
static d3 dnorm(d3 a) { double l = sqrt(a.x * a.x + a.y * a.y + a.z * a.z); return D(a.x / l, a.y / l, a.z / l); }

static uint32_t mix(uint32_t h) { h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16; return h; }
static double u01(uint32_t *state) { *state = *state * 1664525u + 1013904223u; return (double)(mix(*state) >> 8) / 16777216.0; }

/* sin/cos by Taylor series after reduction to [-pi, pi]; deterministic */
static void dsincos(double a, double *s, double *c) {
    const double PI = 3.14159265358979323846, TWO_PI = 6.28318530717958647692;
    double k = floor(a / TWO_PI + 0.5);
    a = a - k * TWO_PI;
    int flip = 0;
    if (a > PI / 2) { a = PI - a; flip = 1; } else if (a < -PI / 2) { a = -PI - a; flip = 1; }
    double a2 = a * a, ts = a, tc = 1.0, ss = a, cc = 1.0;
    for (int i = 1; i <= 12; i++) {
        tc = -tc * a2 / (double)((2 * i - 1) * (2 * i)); cc += tc;
        ts = -ts * a2 / (double)((2 * i) * (2 * i + 1)); ss += ts;
    }
    *s = ss; *c = flip ? -cc : cc;
}

/* 3-D value noise in [0,1], trilinear with smoothstep */
static double lattice(int x, int y, int z, uint32_t seed) {
    uint32_t h = mix((uint32_t)x * 0x8da6b343u ^ mix((uint32_t)y * 0xd8163841u ^ mix((uint32_t)z * 0xcb1ab31fu ^ seed)));
    return (double)(h >> 8) / 16777216.0;
}
static double vnoise(d3 p, uint32_t seed) {
    double fx = floor(p.x), fy = floor(p.y), fz = floor(p.z);
    int ix = (int)fx, iy = (int)fy, iz = (int)fz;
    double tx = p.x - fx, ty = p.y - fy, tz = p.z - fz;
    tx = tx * tx * (3 - 2 * tx); ty = ty * ty * (3 - 2 * ty); tz = tz * tz * (3 - 2 * tz);
    double c[2][2][2];
    for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) for (int d = 0; d < 2; d++)
        c[a][b][d] = lattice(ix + a, iy + b, iz + d, seed);
    double x00 = c[0][0][0] + (c[1][0][0] - c[0][0][0]) * tx, x10 = c[0][1][0] + (c[1][1][0] - c[0][1][0]) * tx;
    double x01 = c[0][0][1] + (c[1][0][1] - c[0][0][1]) * tx, x11 = c[0][1][1] + (c[1][1][1] - c[0][1][1]) * tx;
    double y0 = x00 + (x10 - x00) * ty, y1 = x01 + (x11 - x01) * ty;
    return y0 + (y1 - y0) * tz;
}
static double fbm3(d3 p, uint32_t seed) {
    return (vnoise(p, seed) + 0.5 * vnoise(dscale(p, 2.0), seed + 1) + 0.25 * vnoise(dscale(p, 4.0), seed + 2)) / 1.75;
}

/* ---- noise-displaced icosphere ---- */
typedef struct { d3 center; double radius, amp, freq; uint32_t seed; int64_t limit, made; mesh_t *m; } sphere_job;
static d3 displace(const sphere_job *j, d3 unit) {
    double r = j->radius * (1.0 + j->amp * (fbm3(dadd(dscale(unit, j->freq), D(7.3, 1.9, 4.1)), j->seed) - 0.5));
    return dadd(j->center, dscale(unit, r));
}
static void subdivide(sphere_job *j, d3 a, d3 b, d3 c, int level) {
    if (j->made >= j->limit) return;
    if (level == 0) { emit(j->m, displace(j, a), displace(j, b), displace(j, c)); j->made++; return; }
    d3 ab = dnorm(dscale(dadd(a, b), 0.5)), bc = dnorm(dscale(dadd(b, c), 0.5)), ca = dnorm(dscale(dadd(c, a), 0.5));
    subdivide(j, a, ab, ca, level - 1); subdivide(j, ab, b, bc, level - 1);
    subdivide(j, ca, bc, c, level - 1); subdivide(j, ab, bc, ca, level - 1);
}
static void icosphere(mesh_t *m, d3 center, double radius, double amp, double freq, uint32_t seed, int level, int64_t limit) {
    const double t = 1.6180339887498949;
    d3 v[12] = {D(-1, t, 0), D(1, t, 0), D(-1, -t, 0), D(1, -t, 0), D(0, -1, t), D(0, 1, t),
                D(0, -1, -t), D(0, 1, -t), D(t, 0, -1), D(t, 0, 1), D(-t, 0, -1), D(-t, 0, 1)};
    static const int f[20][3] = {{0, 11, 5}, {0, 5, 1}, {0, 1, 7}, {0, 7, 10}, {0, 10, 11}, {1, 5, 9}, {5, 11, 4},
                                 {11, 10, 2}, {10, 7, 6}, {7, 1, 8}, {3, 9, 4}, {3, 4, 2}, {3, 2, 6}, {3, 6, 8},
                                 {3, 8, 9}, {4, 9, 5}, {2, 4, 11}, {6, 2, 10}, {8, 6, 7}, {9, 8, 1}};
    for (int i = 0; i < 12; i++) v[i] = dnorm(v[i]);
    sphere_job j = {center, radius, amp, freq, seed, limit, 0, m};
    for (int i = 0; i < 20; i++) subdivide(&j, v[f[i][0]], v[f[i][1]], v[f[i][2]], level);
}

static void quad(mesh_t *m, d3 a, d3 b, d3 c, d3 d) { emit(m, a, b, c); emit(m, a, c, d); }

/* grid over p(u,v), u,v in [0,1], nu x nv quads */
typedef d3 (*surf_fn)(double u, double v, const double *prm);
static void grid(mesh_t *m, surf_fn fn, const double *prm, int nu, int nv) {
    for (int j = 0; j < nv; j++) for (int i = 0; i < nu; i++) {
        double u0 = (double)i / nu, u1 = (double)(i + 1) / nu, v0 = (double)j / nv, v1 = (double)(j + 1) / nv;
        quad(m, fn(u0, v0, prm), fn(u1, v0, prm), fn(u1, v1, prm), fn(u0, v1, prm));
    }
}
static d3 plane_fn(double u, double v, const double *p) {   /* origin + u*A + v*B */
    return D(p[0] + u * p[3] + v * p[6], p[1] + u * p[4] + v * p[7], p[2] + u * p[5] + v * p[8]);
}
static void plane(mesh_t *m, d3 o, d3 a, d3 b, int nu, int nv) {
    double p[9] = {o.x, o.y, o.z, a.x, a.y, a.z, b.x, b.y, b.z};
    grid(m, plane_fn, p, nu, nv);
}
static d3 cyl_fn(double u, double v, const double *p) {     /* axis along y: center x,z; y0..y1; radius with entasis */
    double s, c; dsincos(u * 6.28318530717958647692, &s, &c);
    double r = p[4] * (1.0 + p[5] * (v - 0.5) * (v - 0.5));
    return D(p[0] + r * c, p[2] + v * (p[3] - p[2]), p[1] + r * s);
}
static d3 arch_fn(double u, double v, const double *p) {    /* tube of radius p[5] swept on a semicircle in the (z,y) plane */
    double s, c, st, ct;
    dsincos(v * 3.14159265358979323846, &s, &c);           /* along the arch */
    dsincos(u * 6.28318530717958647692, &st, &ct);         /* around the tube */
    double R = p[4] + p[5] * ct;
    return D(p[0] + p[5] * st, p[1] - R * s, p[2] - R * c); /* y up = negative */
}
static d3 cloth_fn(double u, double v, const double *p) {   /* hanging sheet with folds; p: x0,y0,z0,width(z),drop(y),amp,phase,seed */
    double s1, c1, s2, c2;
    dsincos(u * 25.0 + p[6], &s1, &c1);
    dsincos(v * 9.0 + u * 4.0 + p[6] * 1.7, &s2, &c2);
    double n = fbm3(D(u * 6.0, v * 6.0, p[6]), (uint32_t)p[7]) - 0.5;
    double sag = 4.0 * u * (1.0 - u);
    return D(p[0] + p[5] * (s1 * (0.3 + v) + 0.6 * s2) + 3.0 * n, p[1] + v * p[4] + 6.0 * sag * v, p[2] + u * p[3] + 0.8 * c2);
}

/* ---------------- C1: Cornell-like, 32 triangles ---------------- */
static void box5(mesh_t *m, double cx, double cz, double hw, double y_top, double y_bot, double cs, double sn) {
    /* 4 sides + top, rotated about y by (cs,sn) */
    d3 c[4]; const double sx[4] = {-1, 1, 1, -1}, sz[4] = {-1, -1, 1, 1};
    for (int i = 0; i < 4; i++) c[i] = D(cx + hw * (sx[i] * cs - sz[i] * sn), 0, cz + hw * (sx[i] * sn + sz[i] * cs));
    for (int i = 0; i < 4; i++) {
        d3 a = c[i], b = c[(i + 1) & 3];
        quad(m, D(a.x, y_bot, a.z), D(b.x, y_bot, b.z), D(b.x, y_top, b.z), D(a.x, y_top, a.z));
    }
    quad(m, D(c[0].x, y_top, c[0].z), D(c[1].x, y_top, c[1].z), D(c[2].x, y_top, c[2].z), D(c[3].x, y_top, c[3].z));
}
static void scene_cornell(mesh_t *m) {
    const double L = 80.0;
    quad(m, D(-L, L, -L), D(L, L, -L), D(L, L, L), D(-L, L, L));        /* floor (y down) */
    quad(m, D(-L, -L, -L), D(-L, -L, L), D(L, -L, L), D(L, -L, -L));    /* ceiling */
    quad(m, D(-L, -L, L), D(-L, L, L), D(L, L, L), D(L, -L, L));        /* back */
    quad(m, D(-L, -L, -L), D(-L, L, -L), D(-L, L, L), D(-L, -L, L));    /* left */
    quad(m, D(L, -L, -L), D(L, -L, L), D(L, L, L), D(L, L, -L));        /* right */
    box5(m, -28.0, 22.0, 24.0, -18.5, L, 0.95533649, 0.29552021);       /* tall box */
    box5(m, 30.0, -18.0, 22.0, 34.25, L, 0.95533649, -0.29552021);      /* short box */
    emit(m, D(-60.5, 79.5, -60.25), D(-40.25, 79.5, -55.5), D(-52.75, 79.5, -38.5));   /* 2 fillers on the floor */
    emit(m, D(55.5, 79.5, 50.25), D(68.25, 79.5, 58.5), D(50.75, 79.5, 66.5));
}

/* ---------------- C2: bunny-class 69 451-triangle closed mesh ---------------- */
static void scene_bunny(mesh_t *m) {
    icosphere(m, D(0, 0, 0), 60.0, 0.55, 2.2, 1u, 6, 69451);
}

/* ---------------- C3: Sponza-class atrium, exactly 262 267 triangles ---------------- */
#define SPONZA_TRIS 262267
static void scene_sponza(mesh_t *m) {
    const double X = 150, Y = 65, Z = 90;      /* 300 x 130 x 180 box, open at z = -Z and at the roof centre */
    plane(m, D(-X, Y, -Z), D(2 * X, 0, 0), D(0, 0, 2 * Z), 96, 64);                 /* floor      12288 */
    plane(m, D(-X, -Y, Z), D(2 * X, 0, 0), D(0, 2 * Y, 0), 64, 32);                 /* back wall   4096 */
    plane(m, D(-X, -Y, -Z), D(0, 0, 2 * Z), D(0, 2 * Y, 0), 48, 32);                /* left wall   3072 */
    plane(m, D(X, -Y, -Z), D(0, 0, 2 * Z), D(0, 2 * Y, 0), 48, 32);                 /* right wall  3072 */
    plane(m, D(-X, -Y, -Z), D(90, 0, 0), D(0, 0, 2 * Z), 24, 32);                   /* roof strips 3072 */
    plane(m, D(X - 90, -Y, -Z), D(90, 0, 0), D(0, 0, 2 * Z), 24, 32);
    /* two arcades: 8 columns each at x = +-62, arches between neighbours, upper gallery slab */
    for (int side = 0; side < 2; side++) {
        double x = side ? 62.0 : -62.0;
        for (int k = 0; k < 8; k++) {
            double z = -Z + 12.0 + k * (2 * Z - 24.0) / 7.0;
            double col[6] = {x, z, Y, -5.0, 5.5, 0.35};
            grid(m, cyl_fn, col, 40, 28);                                              /* 2240 each */
            if (k < 7) {
                double z1 = z + (2 * Z - 24.0) / 7.0;
                double arch[6] = {x, -5.0, 0.5 * (z + z1), 0, 0.5 * (z1 - z), 3.0};
                grid(m, arch_fn, arch, 24, 40);                                        /* 1920 each */
            }
        }
        double gx = side ? 62.0 : -X;
        plane(m, D(gx, -22.0, -Z), D(X - 62.0, 0, 0), D(0, 0, 2 * Z), 24, 48);      /* gallery floor 2304 */
    }
    /* draped cloths hanging between the arcades */
    for (int k = 0; k < 5; k++) {
        double cl[8] = {-40.0 + 20.0 * k, -60.0, -70.0 + 6.0 * k, 120.0, 55.0 + 4.0 * k, 2.5, 1.3 * k + 0.4, 20.0 + k};
        grid(m, cloth_fn, cl, 128, 128);                                               /* 32768 each */
    }
    /* ornaments: small noise spheres on the floor and gallery until the count is reached */
    uint32_t st = 2u;
    while (m->n < SPONZA_TRIS) {
        double px = -140.0 + 280.0 * u01(&st), pz = -80.0 + 160.0 * u01(&st);
        double r = 2.0 + 4.0 * u01(&st);
        int64_t left = SPONZA_TRIS - m->n;
        icosphere(m, D(px, Y - r * 0.8, pz), r, 0.5, 1.7, mix(st), 3, left < 1280 ? left : 1280);
    }
}

/* ---------------- C4: 10 000 x 1 000-triangle noise spheres on a jittered lattice ---------------- */
static void scene_10m(mesh_t *m, int copies) {
    uint32_t st = 3u;
    int side = 1; while (side * side * side < copies) side++;
    double pitch = 800.0 / side;
    for (int i = 0; i < copies; i++) {
        int ix = i % side, iy = (i / side) % side, iz = i / (side * side);
        d3 c = D(-400.0 + (ix + 0.5 + 0.6 * (u01(&st) - 0.5)) * pitch, -400.0 + (iy + 0.5 + 0.6 * (u01(&st) - 0.5)) * pitch,
                 -400.0 + (iz + 0.5 + 0.6 * (u01(&st) - 0.5)) * pitch);
        double r = pitch * (0.22 + 0.12 * u01(&st));
        icosphere(m, c, r, 0.6, 1.5, mix(st), 3, 1000);
    }
}

/* ---------------- small test scenes ---------------- */
static void scene_random(mesh_t *m, int n, uint32_t seed, double extent, double size) {
    uint32_t st = seed * 2654435761u + 12345u;
    for (int i = 0; i < n; i++) {
        d3 c = D((u01(&st) - 0.5) * 2 * extent, (u01(&st) - 0.5) * 2 * extent, (u01(&st) - 0.5) * 2 * extent);
        d3 p[3];
        for (int k = 0; k < 3; k++) p[k] = dadd(c, D((u01(&st) - 0.5) * size, (u01(&st) - 0.5) * size, (u01(&st) - 0.5) * size));
        emit(m, p[0], p[1], p[2]);
    }
}
/* many triangles sharing one bbox centre (bvh.cpp:174-193 branch) plus a few others */
static void scene_degenerate(mesh_t *m, int n_same, int n_other) {
    uint32_t st = 99u;
    for (int i = 0; i < n_same; i++) {
        double s = floor((4.0 + 28.0 * u01(&st)) * 8.0) / 8.0, h = floor((2.0 + 18.0 * u01(&st)) * 8.0) / 8.0;   /* exact in fp32 */
        /* symmetric about (10,-5,20): bbox centre identical for all */
        if (i & 1) emit(m, D(10 - s, -5 - h, 20 - s), D(10 + s, -5 - h, 20 + s), D(10 - s, -5 + h, 20 + s));
        else       emit(m, D(10 - s, -5 + h, 20 - s), D(10 + s, -5 - h, 20 - s), D(10 + s, -5 + h, 20 + s));
    }
    scene_random(m, n_other, 7u, 70.0, 25.0);
}

/* id: 1 cornell, 2 bunny, 3 sponza, 4 10M (p0 = copies, 0 -> 10000), 10 random (p0=n,p1=seed,p2=extent,p3=size),
 * 11 degenerate (p0=n_same,p1=n_other).  verts NULL -> returns the triangle count only. */
int64_t scn_generate(int id, double p0, double p1, double p2, double p3, float *verts, int64_t cap_tris) {
    mesh_t m = {verts, 0, cap_tris};
    switch (id) {
        case 1: scene_cornell(&m); break;
        case 2: scene_bunny(&m); break;
        case 3: scene_sponza(&m); break;
        case 4: scene_10m(&m, p0 > 0 ? (int)p0 : 10000); break;
        case 10: scene_random(&m, (int)p0, (uint32_t)p1, p2, p3); break;
        case 11: scene_degenerate(&m, (int)p0, (int)p1); break;
        default: return -1;
    }
    return m.n;
}
