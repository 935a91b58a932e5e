/* examples/minimal.c — the C-ABI from plain C: a procedural mesh -> BVH (host side) -> device -> progressive render
 * through the reference-shaped host-buffer call (hpt_render_sample = kernel_execute + the running mean of
 * Pathtracer::render(), reference pathtracer.cpp:185-220) -> PNG.
 *   gcc -std=c11 -Iinclude examples/minimal.c -Lpathtracer_amd -lhpt -Wl,-rpath,$PWD/pathtracer_amd -lm -o build/minimal
 *   build/minimal out.png [width height spp]
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hpt/hpt.h>
#include <hpt/host.h>

static int check(hpt_ctx *ctx, int rc, const char *what) {
    if (rc == HPT_OK) return 0;
    fprintf(stderr, "%s failed (%d): %s\n", what, rc, hpt_last_error(ctx));
    return 1;
}

int main(int argc, char **argv) {
    const char *out = argc > 1 ? argv[1] : "minimal.png";
    const int w = argc > 4 ? atoi(argv[2]) : 128, h = argc > 4 ? atoi(argv[3]) : 128, spp = argc > 4 ? atoi(argv[4]) : 8;

    /* a UV sphere of radius 60 around the origin: 8 x 16 quads = 256 triangles (the builder needs >= 12) */
    enum { RINGS = 8, SEGS = 16, NTRI = RINGS * SEGS * 2 };
    static float verts[NTRI * 9];
    const float pi = 3.14159265358979f;
    int n = 0;
    for (int r = 0; r < RINGS; r++)
        for (int s = 0; s < SEGS; s++) {
            float p[4][3];
            for (int k = 0; k < 4; k++) {
                const float th = pi * (float)(r + (k >> 1)) / RINGS, ph = 2.0f * pi * (float)(s + (k & 1)) / SEGS;
                p[k][0] = 60.0f * sinf(th) * cosf(ph); p[k][1] = 60.0f * cosf(th); p[k][2] = 60.0f * sinf(th) * sinf(ph);
            }
            const int idx[6] = {0, 1, 2, 2, 1, 3};
            for (int t = 0; t < 2; t++, n++)
                for (int k = 0; k < 3; k++) memcpy(verts + 9 * n + 3 * k, p[idx[3 * t + k]], 3 * sizeof(float));
        }

    hpt_bvh_bbox *bbox = malloc(sizeof *bbox * 2 * NTRI);
    hpt_bvh_info *info = malloc(sizeof *info * 2 * NTRI);
    int32_t *order = malloc(sizeof *order * NTRI);
    hpt_triangle *tris = malloc(sizeof *tris * NTRI);
    int depth = 0;
    const int n_nodes = hpt_host_build_bvh(verts, NTRI, bbox, info, order, tris, &depth);
    if (n_nodes < 0) { fprintf(stderr, "BVH build failed\n"); return 1; }

    hpt_ctx *ctx = NULL;
    if (check(NULL, hpt_create(w, h, 0, &ctx), "hpt_create")) return 1;
    if (check(ctx, hpt_upload_triangles(ctx, tris, NTRI), "hpt_upload_triangles")) return 1;
    if (check(ctx, hpt_upload_bvh(ctx, bbox, info, n_nodes), "hpt_upload_bvh")) return 1;

    hpt_material mat;                                   /* pathtracer.cpp:327-328: diffuse (0.6, 0.4, 0.2) */
    memset(&mat, 0, sizeof mat);
    mat.refractive.x = 1.0f; mat.diffuse.x = 0.6f; mat.diffuse.y = 0.4f; mat.diffuse.z = 0.2f;
    hpt_primitive light;                                /* one emissive sphere above the mesh (triangles cannot emit) */
    memset(&light, 0, sizeof light);
    light.refractive.x = 1.0f;
    light.center.x = 40.0f; light.center.y = -160.0f; light.center.z = -60.0f; light.radius = 50.0f;
    light.emissive.x = light.emissive.y = light.emissive.z = light.emissive.w = 12.0f;
    if (check(ctx, hpt_set_scene(ctx, &light, 1, &mat, 1), "hpt_set_scene")) return 1;

    hpt_camera cam;                                     /* the reference's fixed camera, pathtracer.cpp:224-263 */
    memset(&cam, 0, sizeof cam);
    cam.eye.z = -200.0f; cam.n.z = -1.0f; cam.u.x = 1.0f; cam.v.y = 1.0f;
    if (check(ctx, hpt_set_camera(ctx, &cam), "hpt_set_camera")) return 1;
    const hpt_config cfg = {15, 3, 2.0f, 0x5EEDu};      /* MAXDEPTH 15, RR after bounce 3, 2*cos factor */
    if (check(ctx, hpt_set_config(ctx, &cfg), "hpt_set_config")) return 1;

    float *image = calloc((size_t)w * h * 3, sizeof(float));
    for (int it = 0; it < spp; it++)
        if (check(ctx, hpt_render_sample(ctx, it, image), "hpt_render_sample")) return 1;
    hpt_stats st;
    if (check(ctx, hpt_get_stats(ctx, &st), "hpt_get_stats")) return 1;
    if (hpt_host_save_png(image, w, h, out) != 0) { fprintf(stderr, "cannot write %s\n", out); return 1; }
    printf("{\"triangles\": %d, \"nodes\": %d, \"depth\": %d, \"width\": %d, \"height\": %d, \"spp\": %d, \"rays_last_call\": %llu, \"out\": \"%s\"}\n",
           NTRI, n_nodes, depth, w, h, spp, (unsigned long long)(st.rays_primary + st.rays_bounce), out);
    hpt_destroy(ctx);
    free(image); free(tris); free(order); free(info); free(bbox);
    return 0;
}
