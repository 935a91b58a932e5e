// hpt_cli — headless replacement of the reference's GLUT viewer (reference pathtracer/main.cpp:57-223):
// same sequence — new Pathtracer(w,h), load_obj, set_triangles, set_scene, set_camera, render loop, PNG out —
// without a window.  The 'p' key of the viewer (main.cpp:194-203) becomes --out.
//
//   hpt_cli --obj model.obj [--w 100 --h 100] [--spp 16] [--depth 15] [--device 0] [--out image.png|.ppm]
//           [--gpus N | --devices a,b,c]          (several GPUs in this one process: an hpt_group, RCCL called from C++;
//                                                  tile shards + one all-gather per render, image bit-identical to one GPU)
//           [--shard tiles|samples]               (samples: member r renders the iterations s % N == r, one all-reduce)
//           [--reduce-every K]                    (progressive: the exchange after every K iterations)
//           [--checkpoint file [--resume]]        (one device: accumulation buffer + sample count after every slice;
//                                                  --resume continues bit-identically)
//           [--light cx cy cz radius intensity]   (adds an emissive sphere; triangles cannot emit)
//           [--progressive]                       (one render() per sample)
//           [--live N]                            (the viewer's frame loop, main.cpp:121-151: first bounce cached,
//                                                  then N x { iterative_render(); reconstruct(); })
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>


#include "image_writer.hpp"
#include "obj_reader.hpp"
#include "pathtracer.hpp"

using namespace hpt;

static hpt_float4 f4(float x, float y, float z, float w) { hpt_float4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }

int main(int argc, char **argv) {
    std::string obj, out = "out.png";
    int w = 100, h = 100, spp = 16, depth = 15, device = 0, live = 0;      // window 100x100: main.cpp:41-42
    bool progressive = false, have_light = false, sample_shards = false, resume = false;
    int gpus = 0, reduce_every = 0;
    std::vector<int> devices;
    std::string checkpoint;
    float light[5] = {0, -150, -40, 40, 10};
    for (int i = 1; i < argc; i++) {
        auto need = [&](int n) { if (i + n >= argc) { std::fprintf(stderr, "missing value after %s\n", argv[i]); std::exit(2); } };
        if (!std::strcmp(argv[i], "--obj")) { need(1); obj = argv[++i]; }
        else if (!std::strcmp(argv[i], "--out")) { need(1); out = argv[++i]; }
        else if (!std::strcmp(argv[i], "--w")) { need(1); w = std::atoi(argv[++i]); }
        else if (!std::strcmp(argv[i], "--h")) { need(1); h = std::atoi(argv[++i]); }
        else if (!std::strcmp(argv[i], "--spp")) { need(1); spp = std::atoi(argv[++i]); }
        else if (!std::strcmp(argv[i], "--depth")) { need(1); depth = std::atoi(argv[++i]); }
        else if (!std::strcmp(argv[i], "--device")) { need(1); device = std::atoi(argv[++i]); }
        else if (!std::strcmp(argv[i], "--gpus")) { need(1); gpus = std::atoi(argv[++i]); }
        else if (!std::strcmp(argv[i], "--devices")) { need(1); for (char *tok = std::strtok(argv[++i], ","); tok; tok = std::strtok(nullptr, ",")) devices.push_back(std::atoi(tok)); }
        else if (!std::strcmp(argv[i], "--shard")) { need(1); ++i; if (!std::strcmp(argv[i], "samples")) sample_shards = true; else if (std::strcmp(argv[i], "tiles")) { std::fprintf(stderr, "--shard tiles|samples\n"); return 2; } }
        else if (!std::strcmp(argv[i], "--reduce-every")) { need(1); reduce_every = std::atoi(argv[++i]); }
        else if (!std::strcmp(argv[i], "--checkpoint")) { need(1); checkpoint = argv[++i]; }
        else if (!std::strcmp(argv[i], "--resume")) resume = true;
        else if (!std::strcmp(argv[i], "--progressive")) progressive = true;
        else if (!std::strcmp(argv[i], "--live")) { need(1); live = std::atoi(argv[++i]); }
        else if (!std::strcmp(argv[i], "--light")) { need(5); for (int k = 0; k < 5; k++) light[k] = (float)std::atof(argv[++i]); have_light = true; }
        else { std::fprintf(stderr, "unknown option %s\n", argv[i]); return 2; }
    }
    if (obj.empty()) { std::fprintf(stderr, "usage: hpt_cli --obj model.obj [--w W --h H --spp N --depth D --out file.png --light cx cy cz r e --progressive --live N --gpus N --devices a,b --shard tiles|samples --reduce-every K --checkpoint file --resume]\n"); return 2; }

    if (devices.empty()) for (int d = 0; d < gpus; d++) devices.push_back(d);
    const bool grouped = !devices.empty();
    if (grouped && live > 0) { std::fprintf(stderr, "hpt_cli: --live runs on one device\n"); return 2; }
    if (grouped && !checkpoint.empty()) { std::fprintf(stderr, "hpt_cli: --checkpoint runs on one device (several: python -m pathtracer_amd.render)\n"); return 2; }
    std::unique_ptr<Pathtracer> ptp(grouped ? new Pathtracer(w, h, devices, sample_shards) : new Pathtracer(w, h, device));
    Pathtracer &pt = *ptp;
    if (!pt.ok()) { std::fprintf(stderr, "hpt_cli: %s\n", pt.error()); return 1; }
    std::vector<Vec3f> vertices;
    if (!load_obj(obj.c_str(), vertices)) { std::fprintf(stderr, "hpt_cli: cannot read %s\n", obj.c_str()); return 1; }
    auto t0 = std::chrono::steady_clock::now();
    if (!pt.set_triangles(vertices)) { std::fprintf(stderr, "hpt_cli: set_triangles failed (%zu vertices; at least 12 triangles needed): %s\n", vertices.size(), pt.error()); return 1; }
    double t_bvh = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (have_light) {
        std::vector<hpt_material> mats(1);
        std::memset(&mats[0], 0, sizeof mats[0]);
        mats[0].refractive = f4(1, 0, 0, 0); mats[0].diffuse = f4(0.6f, 0.4f, 0.2f, 0);        // pathtracer.cpp:327-328
        std::vector<hpt_primitive> prims(1);
        std::memset(&prims[0], 0, sizeof prims[0]);
        prims[0].refractive = f4(1, 0, 0, 0);
        prims[0].center = f4(light[0], light[1], light[2], 0); prims[0].radius = light[3];
        prims[0].emissive = f4(light[4], light[4], light[4], light[4]);
        if (!pt.set_scene(prims, mats)) { std::fprintf(stderr, "hpt_cli: %s\n", pt.error()); return 1; }
    } else if (!pt.set_scene()) { std::fprintf(stderr, "hpt_cli: %s\n", pt.error()); return 1; }
    pt.set_camera();
    hpt_config cfg{depth, 3, 2.0f, 0x5EEDu};
    pt.set_config(cfg);

    t0 = std::chrono::steady_clock::now();
    Pixel *img = nullptr;
    if (live > 0) {
        pt.iterative_render_first_bounce();
        for (int k = 0; k < live; k++) { pt.iterative_render(); pt.reconstruct(); }
        img = pt.get_image();
    } else if (progressive) for (int s = 0; s < spp; s++) img = pt.render();
    else if (!checkpoint.empty()) {
        // one device, device-resident image, a checkpoint after every slice; --resume picks up where the file ends
        hpt_ctx *ctx = pt.backend();
        float *d_image = nullptr;
        int done = 0;
        if (hpt_image_alloc(ctx, &d_image) != HPT_OK) { std::fprintf(stderr, "hpt_cli: %s\n", pt.error()); return 1; }
        if (resume) {
            if (FILE *f = std::fopen(checkpoint.c_str(), "rb")) {
                std::fclose(f);
                if (hpt_checkpoint_load(ctx, d_image, &done, checkpoint.c_str()) != HPT_OK) { std::fprintf(stderr, "hpt_cli: %s\n", pt.error()); return 1; }
            }
        }
        const int K = reduce_every > 0 ? reduce_every : spp;
        while (done < spp) {
            const int next = done + K < spp ? done + K : spp;
            if (hpt_render(ctx, done, next, d_image) != HPT_OK || hpt_checkpoint_save(ctx, d_image, next, checkpoint.c_str()) != HPT_OK) {
                std::fprintf(stderr, "hpt_cli: %s\n", pt.error());
                return 1;
            }
            done = next;
        }
        img = pt.get_image();
        if (hpt_read_image(ctx, d_image, &img[0].red) != HPT_OK) { std::fprintf(stderr, "hpt_cli: %s\n", pt.error()); return 1; }
        hpt_image_free(ctx, d_image);
        spp = done;
    } else if (reduce_every > 0) { for (int s = 0; s < spp && (s == 0 || img); s += reduce_every) img = pt.render(s + reduce_every < spp ? reduce_every : spp - s); }
    else img = pt.render(spp);
    double t_render = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (!img) { std::fprintf(stderr, "hpt_cli: render failed: %s\n", pt.error()); return 1; }

    Bitmap bmp{img, w, h};
    bool ppm = out.size() > 4 && out.compare(out.size() - 4, 4, ".ppm") == 0;
    if ((ppm ? save_ppm_to_file(&bmp, out.c_str()) : save_png_to_file(&bmp, out.c_str())) != 0) {
        std::fprintf(stderr, "hpt_cli: cannot write %s\n", out.c_str());
        return 1;
    }
    hpt_stats st;
    if (grouped) hpt_group_get_stats(pt.backend_group(), &st); else hpt_get_stats(pt.backend(), &st);
    std::printf("{\"triangles\": %zu, \"width\": %d, \"height\": %d, \"spp\": %d, \"gpus\": %d, \"bvh_s\": %.4f, \"render_s\": %.4f, "
                "\"rays_last_call\": %llu, \"out\": \"%s\"}\n",
                vertices.size() / 3, w, h, checkpoint.empty() ? pt.get_iterations() : spp, grouped ? (int)devices.size() : 1, t_bvh, t_render,
                (unsigned long long)(st.rays_primary + st.rays_bounce), out.c_str());
    return 0;
}
