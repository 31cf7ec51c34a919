// vpt_cli -- headless caller of the hot path: what the reference's main() (source/main.cpp:1131-1891)
// does around `volume_rt_kernel`, without GLFW / ImGui / OIDN, driving libvpt_hip.so through the C ABI
// (include/vpt_abi.h, include/vpt_io.h) exactly as a patched main.cpp would.
//
//   vpt_cli <scene.vdb | scene.ins> [options]
//     --assets DIR        BN0.bmp, blackbody_texture.exr, density_color_texture2.exr   (default ./assets)
//     --size W H          resolution                                                    (default 1920 1080)
//     --spp N             iterations                                                    (default 64)
//     --env FILE.hdr      lat-long HDRI -> environment_type 1                           (default: procedural sky)
//     --lights FILE.ins   "light" instance file (main.cpp:989-1017)
//     --integrator 0|1    direct_integrator / vol_integrator                           (default 0)
//     --sun AZ EL         degrees                                                       (default 120 30)
//     --fov F --aperture A --density-mult D --emission-scale E --g G --ray-depth N --volume-depth N
//     --out PREFIX        writes PREFIX.pfm (linear accum) and PREFIX.ppm (display)     (default render)
//     --png               ... and PREFIX.png (the display image, 8-bit RGB: fileIO.cpp:140-154)
//     --device N          first device
//     --ranks G           render on G GPUs (devices N .. N+G-1), one host thread + one context each: rank r renders
//                         iterations r, r+G, ... (spp / G of them) and the images are combined with ONE RCCL all-reduce
//                         under the C ABI (vpt_allreduce_accum); rank 0 writes the result           (default 1)
// Main-loop mapping: load grids (main.cpp:1283-1303) -> octree (:1313) -> camera (:1321, "F" framing
// :526-543) -> Kernel_params defaults (:1350-1376) -> LUT textures (:1383-1402) -> atmosphere init
// (:1469-1472) -> create_cdf (:1461) -> launches (:1822-1829) -> save (:1583-1650).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <atomic>
#include <thread>
#include <vector>

#include "vpt_abi.h"
#include "vpt_io.h"

#define CHECK(expr)                                                                                        \
    do {                                                                                                   \
        int rc_ = (expr);                                                                                  \
        if (rc_ != VPT_OK) {                                                                               \
            fprintf(stderr, "vpt_cli: %s -> %d: %s | %s\n", #expr, rc_, vpt_last_error(ctx), vpt_io_last_error()); \
            return 1;                                                                                      \
        }                                                                                                  \
    } while (0)
#define HIP(expr)                                                                             \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            fprintf(stderr, "vpt_cli: %s failed: %s\n", #expr, hipGetErrorString(e_));       \
            return 1;                                                                         \
        }                                                                                     \
    } while (0)

static bool ends_with(const std::string& s, const char* suf) {
    const size_t n = strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

struct Options {
    std::string scene, assets = "./assets", env, lights_file, out = "render";
    int W = 1920, H = 1080, spp = 64, device = 0, ranks = 1;
    bool png = false;
    float fov = 30.0f, aperture = 0.0f;
    vpt_kernel_params kp;
};

static int render_rank(const Options& opt, int rank, const unsigned char* comm_id);

int main(int argc, char** argv) {
    vpt_ctx* ctx = nullptr;
    if (argc < 2) {
        fprintf(stderr, "usage: vpt_cli <scene.vdb|scene.ins> [--assets DIR] [--size W H] [--spp N] [--env F.hdr] [--lights F.ins] "
                        "[--integrator 0|1] [--sun AZ EL] [--fov F] [--aperture A] [--density-mult D] [--emission-scale E] [--g G] "
                        "[--ray-depth N] [--volume-depth N] [--out PREFIX] [--png] [--device N] [--ranks G]\n");
        return 2;
    }
    Options opt;
    opt.scene = argv[1];
    std::string &scene = opt.scene, &assets = opt.assets, &env = opt.env, &lights_file = opt.lights_file, &out = opt.out;
    int &W = opt.W, &H = opt.H, &spp = opt.spp, &device = opt.device;
    float &fov = opt.fov, &aperture = opt.aperture;
    vpt_kernel_params& kp = opt.kp;
    vpt_kernel_params_default(&kp);
    kp.max_interactions = 1u << 30;
    for (int i = 2; i < argc; ++i) {
        const std::string a = argv[i];
        auto need = [&](int n) {
            if (i + n >= argc) { fprintf(stderr, "vpt_cli: %s needs %d argument(s)\n", a.c_str(), n); exit(2); }
        };
        if (a == "--assets") { need(1); assets = argv[++i]; }
        else if (a == "--size") { need(2); W = atoi(argv[++i]); H = atoi(argv[++i]); }
        else if (a == "--spp") { need(1); spp = atoi(argv[++i]); }
        else if (a == "--env") { need(1); env = argv[++i]; }
        else if (a == "--lights") { need(1); lights_file = argv[++i]; }
        else if (a == "--integrator") { need(1); kp.integrator = atoi(argv[++i]); }
        else if (a == "--sun") { need(2); kp.azimuth = (float)atof(argv[++i]); kp.elevation = (float)atof(argv[++i]); }
        else if (a == "--fov") { need(1); fov = (float)atof(argv[++i]); }
        else if (a == "--aperture") { need(1); aperture = (float)atof(argv[++i]); }
        else if (a == "--density-mult") { need(1); kp.density_mult = (float)atof(argv[++i]); }
        else if (a == "--emission-scale") { need(1); kp.emission_scale = (float)atof(argv[++i]); }
        else if (a == "--g") { need(1); kp.phase_g1 = (float)atof(argv[++i]); }
        else if (a == "--ray-depth") { need(1); kp.ray_depth = atoi(argv[++i]); }
        else if (a == "--volume-depth") { need(1); kp.volume_depth = atoi(argv[++i]); }
        else if (a == "--out") { need(1); out = argv[++i]; }
        else if (a == "--device") { need(1); device = atoi(argv[++i]); }
        else if (a == "--ranks") { need(1); opt.ranks = atoi(argv[++i]); }
        else if (a == "--png") { opt.png = true; }
        else { fprintf(stderr, "vpt_cli: unknown option %s\n", a.c_str()); return 2; }
    }
    if (W <= 0 || H <= 0 || spp <= 0) { fprintf(stderr, "vpt_cli: bad --size / --spp\n"); return 2; }
    if (opt.ranks < 1 || spp % opt.ranks != 0) { fprintf(stderr, "vpt_cli: --ranks must divide --spp\n"); return 2; }
    (void)ctx; (void)scene; (void)assets; (void)env; (void)lights_file; (void)out; (void)fov; (void)aperture; (void)device;
    if (opt.ranks == 1) return render_rank(opt, 0, nullptr);
    // one host thread per GPU; the RCCL id is made once and shared in-process (a multi-process host would ship the
    // 128 bytes over its own channel)
    unsigned char id[VPT_COMM_ID_BYTES];
    if (vpt_comm_unique_id(id) != VPT_OK) { fprintf(stderr, "vpt_cli: vpt_comm_unique_id: %s\n", vpt_last_error(nullptr)); return 1; }
    std::vector<int> rc((size_t)opt.ranks, 0);
    std::vector<std::thread> th;
    for (int r = 0; r < opt.ranks; ++r) th.emplace_back([&, r] { rc[(size_t)r] = render_rank(opt, r, id); });
    for (auto& t : th) t.join();
    for (int r : rc)
        if (r != 0) return r;
    return 0;
}

static std::atomic<int> g_setup_arrived{0}, g_setup_failed{0};      // --ranks: agreement before the collective communicator init

static int render_rank(const Options& opt, int rank, const unsigned char* comm_id) {
    vpt_ctx* ctx = nullptr;
    const std::string &scene = opt.scene, &assets = opt.assets, &env = opt.env, &lights_file = opt.lights_file, &out = opt.out;
    const int W = opt.W, H = opt.H, spp = opt.spp, ranks = opt.ranks, device = opt.device + rank;
    const float fov = opt.fov, aperture = opt.aperture;
    vpt_kernel_params kp = opt.kp;

    const int created = vpt_create(device, &ctx);
    if (ranks > 1) {
        // the communicator init is collective: every rank first says whether it has a context, and nobody enters
        // ncclCommInitRank unless all have (a rank that failed earlier would leave the others blocked in it for good)
        if (created != VPT_OK) g_setup_failed.fetch_add(1);
        g_setup_arrived.fetch_add(1);
        while (g_setup_arrived.load() < ranks) std::this_thread::yield();
        if (g_setup_failed.load() != 0) {
            if (created == VPT_OK) { fprintf(stderr, "vpt_cli: rank %d: another rank has no context, not joining the communicator\n", rank); vpt_destroy(ctx); return 1; }
        }
    }
    CHECK(created);
    if (ranks > 1) CHECK(vpt_comm_init_rank(ctx, ranks, rank, comm_id));

    // ---- volumes: one file or an instance file (main.cpp:1283-1303, 980-1102) --------------------------
    std::vector<vpt_gpu_vdb> instances;
    std::vector<vpt_io_volume*> files;
    std::vector<vpt_point_light> lights;
    auto load_unique = [&](const char* path, vpt_gpu_vdb* up) -> int {
        vpt_io_volume* v = nullptr;
        int rc = vpt_io_vdb_load(path, "density", "heat", "Cd", &v);      // main.cpp:1061
        if (rc != VPT_OK) return rc;
        files.push_back(v);
        return vpt_io_vdb_upload(ctx, v, up);
    };
    if (ends_with(scene, ".ins")) {
        vpt_io_ins* ins = nullptr;
        CHECK(vpt_io_ins_read(scene.c_str(), &ins));
        if (vpt_io_ins_is_light_file(ins)) { fprintf(stderr, "vpt_cli: %s is a light file; pass it with --lights\n", scene.c_str()); return 2; }
        for (int f = 0; f < vpt_io_ins_num_files(ins); ++f) {
            vpt_gpu_vdb unique;
            CHECK(load_unique(vpt_io_ins_file_name(ins, f), &unique));
            const vpt_io_instance* I = vpt_io_ins_instances(ins, f);
            for (int x = 0; x < vpt_io_ins_num_instances(ins, f); ++x) {
                vpt_gpu_vdb inst = unique;                                  // GPU_VDB(copy): shares the textures
                vpt_instance_xform(unique.xform, I[x].position, I[x].rotation, I[x].scale, inst.xform);
                instances.push_back(inst);
            }
        }
        vpt_io_ins_free(ins);
    } else {
        vpt_gpu_vdb v;
        CHECK(load_unique(scene.c_str(), &v));
        instances.push_back(v);
    }
    if (instances.empty()) { fprintf(stderr, "vpt_cli: no volumes\n"); return 1; }
    if (!lights_file.empty()) {
        vpt_io_ins* li = nullptr;
        CHECK(vpt_io_ins_read(lights_file.c_str(), &li));
        lights.assign(vpt_io_ins_lights(li), vpt_io_ins_lights(li) + vpt_io_ins_num_lights(li));
        vpt_io_ins_free(li);
    }
    CHECK(vpt_scene_set_volumes(ctx, instances.data(), (int)instances.size()));

    // ---- camera, sphere, kernel params -----------------------------------------------------------------
    vpt_camera cam;
    vpt_camera_default(&cam);
    vpt_float3 center;
    float dist = 0;
    vpt_camera_frame(&cam, instances.data(), (int)instances.size(), fov, (float)W / (float)H, aperture, &center, &dist);
    vpt_sphere sph = {{0, 1000, 0}, 1.0f, {10.0f, 0, 0}, 1.0f};              // main.cpp:1480-1484
    kp.resolution = {(unsigned)W, (unsigned)H};

    // ---- look-up tables (main.cpp:1383-1402) --------------------------------------------------------------
    float *bn = nullptr, *bb = nullptr, *dc = nullptr;
    int bw, bh, lw, lh, cw, ch;
    CHECK(vpt_io_load_bmp((assets + "/BN0.bmp").c_str(), &bn, &bw, &bh));
    CHECK(vpt_io_load_exr_rgb((assets + "/blackbody_texture.exr").c_str(), &bb, &lw, &lh));
    CHECK(vpt_io_load_exr_rgb((assets + "/density_color_texture2.exr").c_str(), &dc, &cw, &ch));
    if (bw != 256 || bh != 256 || lw * lh < 256 || cw * ch < 256) { fprintf(stderr, "vpt_cli: unexpected look-up texture sizes\n"); return 1; }
    const size_t n = (size_t)W * H;
    void *d_bn, *d_bb, *d_dc, *d_accum, *d_cost, *d_depth, *d_raw, *d_disp;
    HIP(hipSetDevice(device));
    HIP(hipMalloc(&d_bn, 65536 * 12)); HIP(hipMemcpy(d_bn, bn, 65536 * 12, hipMemcpyHostToDevice));
    HIP(hipMalloc(&d_bb, 256 * 12)); HIP(hipMemcpy(d_bb, bb, 256 * 12, hipMemcpyHostToDevice));
    HIP(hipMalloc(&d_dc, 256 * 12)); HIP(hipMemcpy(d_dc, dc, 256 * 12, hipMemcpyHostToDevice));
    HIP(hipMalloc(&d_accum, n * 12)); HIP(hipMemset(d_accum, 0, n * 12));       // main.cpp:596-637
    HIP(hipMalloc(&d_cost, n * 12)); HIP(hipMemset(d_cost, 0, n * 12));
    HIP(hipMalloc(&d_depth, n * 4)); HIP(hipMemset(d_depth, 0, n * 4));
    HIP(hipMalloc(&d_raw, n * 16)); HIP(hipMemset(d_raw, 0, n * 16));
    HIP(hipMalloc(&d_disp, n * 4)); HIP(hipMemset(d_disp, 0, n * 4));
    kp.blue_noise_buffer = (vpt_float3*)d_bn;
    kp.emission_texture = (vpt_float3*)d_bb;
    kp.density_color_texture = (vpt_float3*)d_dc;
    kp.accum_buffer = (vpt_float3*)d_accum;
    kp.cost_buffer = (vpt_float3*)d_cost;
    kp.depth_buffer = (float*)d_depth;
    kp.raw_buffer = (vpt_float4*)d_raw;
    kp.display_buffer = (unsigned int*)d_disp;

    // ---- environment (main.cpp:1441-1472) ---------------------------------------------------------------------
    vpt_atmosphere_parameters atm;
    CHECK(vpt_atmosphere_default_model(&atm));
    if (!env.empty()) {
        float* px = nullptr;
        int ew, eh;
        CHECK(vpt_io_load_hdr(env.c_str(), &px, &ew, &eh));
        vpt_texture_desc d = {ew, eh, 1, 4, 1, VPT_FILTER_LINEAR, {VPT_ADDR_WRAP, VPT_ADDR_CLAMP, VPT_ADDR_CLAMP}};   // main.cpp:967-976
        CHECK(vpt_texture_create(ctx, &d, px, &kp.env_tex));
        vpt_io_free(px);
        kp.environment_type = 1;
    }
    const auto t_pre = std::chrono::steady_clock::now();
    if (kp.environment_type == 0 || kp.integrator != 0) CHECK(vpt_atmosphere_precompute(ctx, &atm, 4, nullptr));
    if (kp.integrator != 0 && kp.environment_type == 0) CHECK(vpt_env_cdf_create(ctx, &kp));
    CHECK(vpt_sync(ctx));
    const double pre_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_pre).count();

    // ---- render: `spp` launches of volume_rt_kernel (main.cpp:1822-1829) -----------------------------------------
    vpt_light_list ll = {(unsigned)lights.size(), lights.empty() ? nullptr : lights.data()};
    // rank r of G: iterations r, r+G, ... with the blue-noise table advanced r steps; G = 1: all of them
    kp.iteration = (unsigned)rank;
    const auto t0 = std::chrono::steady_clock::now();
    if (rank > 0) CHECK(vpt_blue_noise_advance(ctx, kp.blue_noise_buffer, (unsigned)rank, (unsigned)n, nullptr));
    CHECK(vpt_render_batch(ctx, &cam, &ll, &sph, &atm, &kp, (unsigned)(spp / ranks), (unsigned)ranks, nullptr));
    if (ranks > 1) {
        // the one collective of the path, then the display image of the JOB's mean (the batch tonemapped this rank's)
        CHECK(vpt_allreduce_accum(ctx, (float*)d_accum, (unsigned long long)n * 3ull, (unsigned)(spp / ranks), nullptr));
        CHECK(vpt_resolve_display(ctx, &kp, nullptr));
    }
    CHECK(vpt_sync(ctx));
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (rank != 0) {
        for (vpt_io_volume* v : files) vpt_io_vdb_free(v);
        vpt_io_free(bn); vpt_io_free(bb); vpt_io_free(dc);
        vpt_destroy(ctx);
        return 0;
    }

    // ---- save (main.cpp:1583-1650: linear image + display image) ------------------------------------------------------
    std::vector<float> accum(n * 3);
    std::vector<unsigned int> disp(n);
    HIP(hipMemcpy(accum.data(), d_accum, n * 12, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(disp.data(), d_disp, n * 4, hipMemcpyDeviceToHost));
    CHECK(vpt_io_write_pfm((out + ".pfm").c_str(), accum.data(), 3, W, H));
    CHECK(vpt_io_write_ppm((out + ".ppm").c_str(), disp.data(), W, H));
    if (opt.png) CHECK(vpt_io_write_png((out + ".png").c_str(), disp.data(), W, H, 0));      // the reference's save_texture_png(uint32_t*), fileIO.cpp:140-154
    double mean = 0;
    for (float v : accum) mean += v;
    vpt_float3 lo, hi;
    float mx, mn;
    vpt_scene_get_root(ctx, &lo, &hi, &mx, &mn);
    printf("{\"scene\": \"%s\", \"instances\": %zu, \"width\": %d, \"height\": %d, \"spp\": %d, \"integrator\": %d, \"environment_type\": %u, "
           "\"ranks\": %d, \"render_s\": %.6f, \"msamples_per_s\": %.3f, \"precompute_s\": %.3f, \"mean\": %.6g, \"max_extinction\": %g, "
           "\"camera_dist\": %g, \"out\": \"%s.pfm\"}\n",
           scene.c_str(), instances.size(), W, H, spp, kp.integrator, kp.environment_type, ranks, s, (double)n * spp / s / 1e6, pre_s,
           mean / (double)(n * 3), mx, dist, out.c_str());
    for (vpt_io_volume* v : files) vpt_io_vdb_free(v);
    vpt_io_free(bn); vpt_io_free(bb); vpt_io_free(dc);
    vpt_destroy(ctx);
    return 0;
}
