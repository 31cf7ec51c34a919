// vpt_caches.hip -- the per-view caches of the environment tail, host side: what exists, what it is a function of (ONE key,
// vpt_ctx.h: ViewKey), when it is (re)built and what a render is pointed at.  The kernels are in vpt_tail.hip.
//
//   layer 1  camera-point scattering tables + SkyView (vpt_sky.h)       key.tables                        every render with the sky
//   layer 2  view-point ground tables, with their build-time checks      key.tables                        direct integrator, procedural sky
//   layer 3  per-pixel sky patches + never-traced pixel mask + sky dome  the whole key                     closed lens; a batch (>= 2 iterations)
//            + list of the pixels without a patch                                                           or a view that repeats
//   layer 3' one sky dome per table variant                              key.tables + sky_mult, sky_color   open lens; a batch
//   on top of layer 3: RESOLVED SAMPLES (vpt_device.h: ResolveParams::lean) -- the tracer resolves finished paths from the dome
//
// A layer is valid while its flag is set and the part of ctx->view_built it depends on equals the current key; building a lower layer
// clears the flags above it.  vpt_invalidate_sky_tables clears them all (the tables' CONTENTS changed behind unchanged addresses).
// Streams: one context's renders are serialised on the device (vpt_render_batch waits for the previous render's last kernel when the
// stream changes), so a rebuild never overwrites what an earlier render still reads.
#include <algorithm>
#include <cmath>
#include <cstddef>

#include "vpt_ctx.h"

using namespace vpt;

namespace {

// frees and reallocations need the device idle for this context: the stream of this render and of the previous one
int quiesce(vpt_ctx* ctx, hipStream_t stream) {
    HIPCHK(ctx, hipStreamSynchronize(stream));
    if (ctx->render_stream && ctx->render_stream != stream) HIPCHK(ctx, hipStreamSynchronize(ctx->render_stream));
    return VPT_OK;
}
#define CHK(expr) do { const int rc_ = (expr); if (rc_ != VPT_OK) return rc_; } while (0)

bool same_tables(const ViewKey& a, const ViewKey& b) { return std::memcmp(&a.tables, &b.tables, sizeof(a.tables)) == 0; }
bool same_view(const ViewKey& a, const ViewKey& b) { return std::memcmp(&a, &b, sizeof(ViewKey)) == 0; }

// The tiles (8x8 pixels) some NON-EMPTY octree leaf may be seen through: every such leaf's box -- the root's halved three times (divide_bbox,
// bvh_kernels.cu:150-202; child index: x high, y LOW, z high, as locate has it), in binary64, within an ulp of the tracer's binary32 planes --
// projected like the root (vpt_project_box) and grown by `margin` pixels.  false: a leaf corner at or behind the camera plane (no bound).
// occ: the 19 occupancy words of TraceParams::occ ([3..18]: level 3, bit = path).
bool build_leaf_tiles(const vpt_camera* cam, const double blo[3], const double bhi[3], const uint32_t* occ, uint32_t W, uint32_t H, double margin,
                      std::vector<unsigned char>& tiles) {
    const uint32_t tw = (W + 7u) / 8u, th = (H + 7u) / 8u;
    tiles.assign((size_t)tw * th, 0);
    for (int path = 0; path < 512; ++path) {
        if (((occ[(96 + path) >> 5] >> (path & 31)) & 1u) == 0u) continue;
        double lo[3] = {blo[0], blo[1], blo[2]}, hi[3] = {bhi[0], bhi[1], bhi[2]};
        for (int level = 0; level < 3; ++level) {
            const int c = (path >> (6 - 3 * level)) & 7;
            const bool high[3] = {(c & 1) != 0, (c & 2) == 0, (c & 4) != 0};
            for (int a = 0; a < 3; ++a) {
                const double mid = (lo[a] + hi[a]) * 0.5;
                if (high[a]) lo[a] = mid; else hi[a] = mid;
            }
        }
        double r[4];
        if (!vpt_project_box(cam, lo, hi, (double)W, (double)H, r)) return false;
        // a tile [8 t, 8 t + 8] meets [r0 - margin, r2 + margin]: t >= (r0 - margin) / 8 - 1, rounded DOWN once more where it is not an integer
        const long x0 = std::max(0L, (long)std::floor((r[0] - margin - 1.0) / 8.0)), y0 = std::max(0L, (long)std::floor((r[1] - margin - 1.0) / 8.0));
        const long x1 = std::min((long)tw - 1, (long)std::floor((r[2] + margin) / 8.0)), y1 = std::min((long)th - 1, (long)std::floor((r[3] + margin) / 8.0));
        for (long ty = y0; ty <= y1; ++ty)
            for (long tx = x0; tx <= x1; ++tx) tiles[(size_t)ty * tw + (size_t)tx] = 1;
    }
    return true;
}

}  // namespace

int vpt_view_caches_prepare(vpt_ctx* ctx, const vpt_camera* cam, const vpt_sphere* ref_sphere, const vpt_kernel_params* kp, bool compact,
                            unsigned int iter_count, ResolveParams& R, TraceParams& P, hipStream_t stream) {
    if (!R.has_atmosphere || ctx->no_cam_table) return VPT_OK;
    const uint32_t W = R.width, H = R.height, n_pixels = R.n_pixels;
    const bool direct_sky = kp->integrator == 0 && kp->environment_type == 0;

    // ---- the key of this render
    ViewKey key;
    std::memset(&key, 0, sizeof(key));
    key.tables.view_pos[0] = cam->origin.x; key.tables.view_pos[1] = cam->origin.y; key.tables.view_pos[2] = cam->origin.z;
    std::memcpy(key.tables.sun_dir, R.sun_dir, sizeof(float) * 3);
    std::memcpy(key.tables.atm_f, R.atm_f, sizeof(float) * 40);
    // variants: valid for samples whose env_pos has the camera origin's (r, mu_s) -- or, behind an open lens, an r within k binary32
    // steps of it (the lens disc spans lens_radius of height: k = that in steps of r, + 1)
    int view_k = 0;
    if (cam->lens_radius != 0.0f) {
        const double py = (double)cam->origin.y + (double)R.atm_f[0], r = std::sqrt((double)cam->origin.x * cam->origin.x + py * py + (double)cam->origin.z * cam->origin.z);
        const double step = std::ldexp(1.0, std::ilogb(r) - 23);                  // binary32 spacing at r
        view_k = (int)std::min<double>((double)SKY_VIEW_MAX_K, std::ceil(std::fabs((double)cam->lens_radius) / step) + 1.0);
    }
    key.tables.view_k = view_k;
    key.tables.tex[0] = R.transmittance_tex.data; key.tables.tex[1] = R.scattering_tex.data;
    key.tables.tex[2] = R.irradiance_tex.data; key.tables.tex[3] = R.single_mie_tex.data;
    const float frame[9] = {cam->lower_left_corner.x, cam->lower_left_corner.y, cam->lower_left_corner.z, cam->horizontal.x, cam->horizontal.y, cam->horizontal.z,
                            cam->vertical.x, cam->vertical.y, cam->vertical.z};
    std::memcpy(key.frame, frame, sizeof(frame));
    key.width = (float)W; key.height = (float)H;
    key.sky_mult = kp->sky_mult; key.sky_color[0] = kp->sky_color.x; key.sky_color[1] = kp->sky_color.y; key.sky_color[2] = kp->sky_color.z;
    key.ground_table = ctx->no_dir_table ? 0.0f : 1.0f + ctx->dir_tab_tol;

    // ---- layer 1: camera-point scattering tables.  A function of the view point, the lens (variants), the sun direction, the model
    // scalars and the two 4-D tables: a frame loop that changes none of them (main.cpp:1822-1829, one launch per iteration) builds them once
    if (!ctx->d_cam_tab) {
        HIPCHK(ctx, hipMalloc(&ctx->d_cam_tab, sky_cam_table_bytes()));
        HIPCHK(ctx, hipMalloc(&ctx->d_sky_view, sizeof(SkyView)));
    }
    R.cam_tab = ctx->d_cam_tab;
    R.sky_view = ctx->d_sky_view;
    std::memcpy(R.cam_tab_pos, key.tables.view_pos, sizeof(float) * 3);
    if (!ctx->cam_tab_built || !same_tables(key, ctx->view_built)) {
        HIPCHK(ctx, launch_sky_cam_table(R, ctx->d_sky_view, ctx->d_cam_tab, view_k, stream));
        ctx->view_built.tables = key.tables;
        ctx->cam_tab_built = true;
        ctx->dir_tab_built = false;
        ctx->sky_patch_built = false;
        ctx->lens_dome_built = false;
    }
    R.cam_tab_valid = 1;

    // ---- layer 2: view-point ground tables (vpt_sky.h): the direct integrator's procedural sky; which variants get one (view point
    // between the ground and the top of the atmosphere) is sky_view_kernel's decision
    if (!ctx->no_dir_table && direct_sky) {
        if (!ctx->d_dir_tab) {
            HIPCHK(ctx, hipMalloc(&ctx->d_dir_tab, sky_dir_table_bytes()));
            HIPCHK(ctx, hipMalloc(&ctx->d_dir_err, SKY_DIR_ERR_WORDS * sizeof(unsigned long long)));
        }
        R.dir_tab_tol = ctx->dir_tab_tol;
        R.dir_tab = ctx->d_dir_tab;                  // (set before the build: its check evaluates real rays through the table path)
        R.dir_tab_err = reinterpret_cast<const uint32_t*>(ctx->d_dir_err);
        if (!ctx->dir_tab_built) {
            HIPCHK(ctx, launch_sky_dir_table(R, ctx->d_sky_view, ctx->d_dir_tab, ctx->d_dir_err, view_k, stream));
            ctx->dir_tab_built = true;
        }
    }

    // ---- layer 3: per-pixel sky patches (ResolveParams::sky_patch), the never-traced pixel mask and the sky dome: untraced / traced
    // samples behind a closed lens, direct integrator, procedural sky, heads
    if (!ctx->no_sky_patch && compact && cam->lens_radius == 0.0f && direct_sky) {
        for (int i = 0; i < 3; ++i) { R.cam_llc[i] = frame[i]; R.cam_h[i] = frame[3 + i]; R.cam_v[i] = frame[6 + i]; }
        // never-traced pixels: screen-space bounds of the root box grown by 3 pixels (the rounding of a slab test moves a hit by ~eps / pixel
        // angle: a tenth of a pixel at most), the sphere by its inflated radius per pixel (sky_patch_kernel), and the line
        // dir . (origin - centre) = 0 of sphere::intersect's B == 0 case.  No culling when a box corner is at or behind the camera plane,
        // or the origin sits exactly on a slab plane (0 x inf in the slab test).
        R.render = kp->render ? 1 : 0;
        R.cull_enabled = 0;
        if (!ctx->no_pixel_cull) {
            const double blo[3] = {P.root_pmin[0], P.root_pmin[1], P.root_pmin[2]}, bhi[3] = {P.root_pmax[0], P.root_pmax[1], P.root_pmax[2]};
            double rb[4];
            const double o[3] = {cam->origin.x, cam->origin.y, cam->origin.z};
            bool on_plane = false;
            for (int i = 0; i < 3; ++i) on_plane = on_plane || o[i] == blo[i] || o[i] == bhi[i];
            if (!on_plane && vpt_project_box(cam, blo, bhi, (double)W, (double)H, rb)) {
                const double m = 3.0;
                for (int i = 0; i < 2; ++i) { R.cull_rect[i] = (float)(rb[i] - m); R.cull_rect[2 + i] = (float)(rb[2 + i] + m); }
                const double orig[3] = {o[0] - ref_sphere->center.x, o[1] - ref_sphere->center.y, o[2] - ref_sphere->center.z};
                const double A[3] = {cam->lower_left_corner.x - o[0], cam->lower_left_corner.y - o[1], cam->lower_left_corner.z - o[2]};
                R.cull_line[0] = (float)((cam->horizontal.x * orig[0] + cam->horizontal.y * orig[1] + cam->horizontal.z * orig[2]) / (double)W);
                R.cull_line[1] = (float)((cam->vertical.x * orig[0] + cam->vertical.y * orig[1] + cam->vertical.z * orig[2]) / (double)H);
                R.cull_line[2] = (float)(A[0] * orig[0] + A[1] * orig[1] + A[2] * orig[2]);
                R.cull_sph[0] = ref_sphere->center.x; R.cull_sph[1] = ref_sphere->center.y; R.cull_sph[2] = ref_sphere->center.z;
                R.cull_sph[3] = ref_sphere->radius;
                // worth its flag look-ups only when a good part of the frame lies outside the box's rectangle (config 3's fireball fills the
                // picture: raygen 4 % slower with the flags than without)
                const double ix0 = std::max(0.0, (double)R.cull_rect[0]), iy0 = std::max(0.0, (double)R.cull_rect[1]);
                const double ix1 = std::min((double)W, (double)R.cull_rect[2]), iy1 = std::min((double)H, (double)R.cull_rect[3]);
                const double inside = std::max(0.0, ix1 - ix0) * std::max(0.0, iy1 - iy0);
                R.cull_enabled = inside <= 0.7 * (double)W * (double)H ? 1 : 0;
                // ... and per 8x8 tile inside those bounds: the screen bounds of every NON-EMPTY octree leaf, grown by the same 3 pixels
                // (ResolveParams::cull_tiles, build_leaf_tiles above)
                const uint32_t tw = (W + 7u) / 8u, th = (H + 7u) / 8u;
                std::vector<unsigned char>& tiles = ctx->cull_tiles_host;
                bool refined = !ctx->no_leaf_cull && !ctx->counting;      // (a counting render keeps the skip counts of the rays this removes: the reference walks them)
                // the map is a function of the camera, the image size and the octree: a frame loop that changes none of them (one call per iteration,
                // main.cpp:1822-1829) must not pay ~0.1 ms of host time per call for it
                float in[9 + 3 + 6 + 2 + 19];
                std::memcpy(in, frame, sizeof(frame));
                in[9] = cam->origin.x; in[10] = cam->origin.y; in[11] = cam->origin.z;
                for (int i = 0; i < 3; ++i) { in[12 + i] = P.root_pmin[i]; in[15 + i] = P.root_pmax[i]; }
                in[18] = (float)W; in[19] = (float)H;
                std::memcpy(in + 20, ctx->occ, sizeof(uint32_t) * 19);
                const bool reuse = refined && ctx->cull_tiles_inputs_valid && std::memcmp(in, ctx->cull_tiles_inputs, sizeof(in)) == 0;
                if (refined && !reuse) {
                    if (ctx->cull_tiles_copy_pending) {           // (an upload of the previous map may still be queued: it reads this very buffer)
                        HIPCHK(ctx, hipEventSynchronize(ctx->cull_tiles_copied));
                        ctx->cull_tiles_copy_pending = false;
                    }
                    refined = build_leaf_tiles(cam, blo, bhi, ctx->occ, W, H, m, tiles);
                }
                if (refined && !reuse) {
                    uint32_t h0 = 2166136261u, h1 = 0x9747b28cu;
                    size_t covered = 0;
                    for (unsigned char t : tiles) { h0 = (h0 ^ t) * 16777619u; h1 = (h1 ^ (t + 1u)) * 0x01000193u + 0x9e3779b9u; covered += t; }
                    ctx->cull_tiles_hash[0] = h0 | 1u; ctx->cull_tiles_hash[1] = h1; ctx->cull_tiles_covered = covered;
                    std::memcpy(ctx->cull_tiles_inputs, in, sizeof(in));
                    ctx->cull_tiles_inputs_valid = true;
                }
                if (!refined && !ctx->no_leaf_cull && !ctx->counting) ctx->cull_tiles_inputs_valid = false;       // (a leaf behind the camera plane: nothing to reuse)
                if (refined) {
                    const size_t covered = ctx->cull_tiles_covered;
                    key.cull_tiles_hash[0] = ctx->cull_tiles_hash[0]; key.cull_tiles_hash[1] = ctx->cull_tiles_hash[1];
                    if (ctx->cull_tiles_bytes < tiles.size()) {
                        CHK(quiesce(ctx, stream));
                        (void)hipFree(ctx->d_cull_tiles); ctx->d_cull_tiles = nullptr; ctx->cull_tiles_bytes = 0;
                        HIPCHK(ctx, hipMalloc(&ctx->d_cull_tiles, tiles.size()));
                        ctx->cull_tiles_bytes = tiles.size();
                        ctx->sky_patch_built = false;
                    }
                    R.cull_tiles = ctx->d_cull_tiles;
                    R.cull_tiles_w = tw;
                    R.cull_enabled = (double)covered <= 0.7 * (double)tiles.size() ? 1 : 0;
                }
            }
        }
        key.cull_enabled = (float)R.cull_enabled; key.render = (float)R.render;
        std::memcpy(key.cull_rect, R.cull_rect, sizeof(float) * 4);
        std::memcpy(key.cull_line, R.cull_line, sizeof(float) * 3);
        std::memcpy(key.cull_sph, R.cull_sph, sizeof(float) * 4);
        if (ctx->sky_patch_pixels < (size_t)n_pixels) {
            CHK(quiesce(ctx, stream));
            (void)hipFree(ctx->d_sky_patch); ctx->d_sky_patch = nullptr; ctx->sky_patch_pixels = 0;
            (void)hipFree(ctx->d_never_traced); ctx->d_never_traced = nullptr;
            (void)hipFree(ctx->d_nopatch); ctx->d_nopatch = nullptr;
            HIPCHK(ctx, hipMalloc(&ctx->d_sky_patch, (size_t)n_pixels * 3u * sizeof(float4)));
            HIPCHK(ctx, hipMalloc(&ctx->d_never_traced, (size_t)n_pixels));
            HIPCHK(ctx, hipMalloc(&ctx->d_nopatch, ((size_t)n_pixels + 1u) * sizeof(uint32_t)));
            ctx->sky_patch_pixels = n_pixels;
            ctx->sky_patch_built = false;
        }
        // The patches, the mask and the dome cost ~0.3 ms to build: they are built for a batch (>= 2 iterations), or once a view repeats (the
        // progressive render of a still camera, main.cpp:1822-1829 frame after frame) -- a camera that moves every frame with one iteration
        // per frame never pays for them.
        const bool built_for_this = ctx->sky_patch_built && same_view(key, ctx->view_built);
        const bool view_repeats = ctx->view_seen_valid && same_view(key, ctx->view_seen);
        ctx->view_seen = key;
        ctx->view_seen_valid = true;
        const bool use_caches = built_for_this || iter_count >= 2u || view_repeats;
        if (use_caches && !built_for_this) {
            if (R.cull_tiles != nullptr)       // (the tile map the mask is built from; the host copy lives in the context until the next build)
            {
                HIPCHK(ctx, hipMemcpyAsync(ctx->d_cull_tiles, ctx->cull_tiles_host.data(), ctx->cull_tiles_host.size(), hipMemcpyHostToDevice, stream));
                if (!ctx->cull_tiles_copied) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->cull_tiles_copied, hipEventDisableTiming));
                HIPCHK(ctx, hipEventRecord(ctx->cull_tiles_copied, stream));
                ctx->cull_tiles_copy_pending = true;
            }
            HIPCHK(ctx, launch_sky_patch(R, ctx->d_sky_patch, ctx->d_never_traced, ctx->d_nopatch + 1, ctx->d_nopatch, stream));
            if (!ctx->no_sky_dome) {
                if (ctx->sky_dome_k < 0) {
                    HIPCHK(ctx, hipMalloc(&ctx->d_sky_dome, sky_dome_bytes(0)));
                    ctx->sky_dome_k = 0;
                }
                HIPCHK(ctx, launch_sky_dome(R, ctx->d_sky_view, ctx->d_sky_dome, 0, stream));
                ctx->lens_dome_built = false;
            }
            ctx->view_built = key;
            ctx->sky_patch_built = true;
        }
        if (use_caches) {
            R.sky_patch = ctx->d_sky_patch;
            R.sky_dome = ctx->no_sky_dome ? nullptr : ctx->d_sky_dome;
            R.blue_noise = ctx->d_bn_table;
            if (R.cull_enabled) {
                R.never_traced = ctx->d_never_traced;
                P.never_traced = ctx->d_never_traced;
            }
            // RESOLVED SAMPLES (vpt_device.h): patches + dome in use => the tracer resolves its finished paths from the dome and the tail
            // streams heads only.  (The pool tracer, an A/B harness, keeps the records.)
            if (R.sky_dome != nullptr && !ctx->no_lean_tail && !ctx->use_pool) {
                if (ctx->td_capacity < ctx->records_capacity) {
                    CHK(quiesce(ctx, stream));
                    (void)hipFree(ctx->d_td); ctx->d_td = nullptr; ctx->td_capacity = 0;
                    (void)hipFree(ctx->d_queue2); ctx->d_queue2 = nullptr;
                    HIPCHK(ctx, hipMalloc(&ctx->d_td, ctx->records_capacity * sizeof(float2)));
                    HIPCHK(ctx, hipMalloc(&ctx->d_queue2, ctx->records_capacity * sizeof(uint32_t)));
                    ctx->td_capacity = ctx->records_capacity;
                }
                R.lean = 1;
                R.td = ctx->d_td;
                R.queue2 = ctx->d_queue2;
                R.queue2_count = ctx->d_work_counter + 4;
                R.nopatch_list = ctx->d_nopatch + 1;
                R.nopatch_count = ctx->d_nopatch;
                ResolveInTracer rt;
                std::memset(&rt, 0, sizeof(rt));
                rt.sky_dome = R.sky_dome; rt.heads = ctx->d_heads; rt.td = ctx->d_td; rt.queue2 = ctx->d_queue2; rt.queue2_tail = ctx->d_work_counter + 4;
                rt.cam_origin[0] = cam->origin.x; rt.cam_origin[1] = cam->origin.y; rt.cam_origin[2] = cam->origin.z;
                P.resolve = rt;                     // by value, in the kernel-argument segment (no upload, nothing a later render could overwrite)
            }
        }
    }

    // ---- layer 3': open lens: one dome per table variant (vpt_tail.hip: sky_dome_kernel<true>), for batches; traced AND untraced samples use them
    if (!ctx->no_sky_dome && compact && cam->lens_radius != 0.0f && direct_sky) {
        const bool valid = ctx->lens_dome_built && ctx->sky_dome_k >= view_k && ctx->view_built.sky_mult == key.sky_mult &&
                           std::memcmp(ctx->view_built.sky_color, key.sky_color, sizeof(key.sky_color)) == 0;
        if (valid || iter_count >= 2u) {
            if (!valid) {
                if (ctx->sky_dome_k < view_k) {
                    CHK(quiesce(ctx, stream));
                    (void)hipFree(ctx->d_sky_dome); ctx->d_sky_dome = nullptr; ctx->sky_dome_k = -1;
                    HIPCHK(ctx, hipMalloc(&ctx->d_sky_dome, sky_dome_bytes(view_k)));
                    ctx->sky_dome_k = view_k;
                }
                HIPCHK(ctx, launch_sky_dome(R, ctx->d_sky_view, ctx->d_sky_dome, view_k, stream));
                ctx->view_built.sky_mult = key.sky_mult;
                std::memcpy(ctx->view_built.sky_color, key.sky_color, sizeof(key.sky_color));
                ctx->lens_dome_built = true;
                ctx->sky_patch_built = false;             // (the closed-lens dome in the same allocation is gone)
            }
            R.sky_dome = ctx->d_sky_dome;
            // RESOLVED SAMPLES behind the open lens (round 5): the variant's dome serves traced paths in the tracer and untraced samples in raygen
            if (!ctx->no_lean_tail && !ctx->use_pool && !ctx->no_lens_lean) {
                if (ctx->td_capacity < ctx->records_capacity) {
                    CHK(quiesce(ctx, stream));
                    (void)hipFree(ctx->d_td); ctx->d_td = nullptr; ctx->td_capacity = 0;
                    (void)hipFree(ctx->d_queue2); ctx->d_queue2 = nullptr;
                    HIPCHK(ctx, hipMalloc(&ctx->d_td, ctx->records_capacity * sizeof(float2)));
                    HIPCHK(ctx, hipMalloc(&ctx->d_queue2, ctx->records_capacity * sizeof(uint32_t)));
                    ctx->td_capacity = ctx->records_capacity;
                }
                R.lean = 1;
                R.td = ctx->d_td;
                R.queue2 = ctx->d_queue2;
                R.queue2_count = ctx->d_work_counter + 4;
                ResolveInTracer rt;
                std::memset(&rt, 0, sizeof(rt));
                rt.sky_dome = R.sky_dome; rt.heads = ctx->d_heads; rt.td = ctx->d_td; rt.queue2 = ctx->d_queue2; rt.queue2_tail = ctx->d_work_counter + 4;
                rt.cam_origin[0] = cam->origin.x; rt.cam_origin[1] = cam->origin.y; rt.cam_origin[2] = cam->origin.z;
                rt.lens = 1;
                rt.sky_view = ctx->d_sky_view;
                rt.sun_dir[0] = R.sun_dir[0]; rt.sun_dir[1] = R.sun_dir[1]; rt.sun_dir[2] = R.sun_dir[2];
                rt.earth_bottom = R.atm_f[0];
                P.resolve = rt;
            }
        }
    }
    return VPT_OK;
}

extern "C" {

// The per-view caches are keyed on the view point, the sun, the model scalars and the ADDRESSES of the four look-up tables -- not on
// the tables' contents.  Whatever can change those contents behind an unchanged address drops the caches: every texture create /
// destroy (a re-upload of the same size usually lands on the address just freed), vpt_atmosphere_precompute (it refills the buffers
// it is handed), and this entry point for a host that rewrites a device table in place.
int vpt_invalidate_sky_tables(vpt_ctx* ctx) {
    if (!ctx) return VPT_E_INVALID;
    ctx->cam_tab_built = false;
    ctx->dir_tab_built = false;
    ctx->sky_patch_built = false;
    ctx->lens_dome_built = false;
    ctx->ahead.key_valid = false;            // (rays traced ahead looked at the old tables)
    ctx->ahead.n = 0;
    return VPT_OK;
}

// include/vpt_abi.h "FRAME-AHEAD": device memory rewritten in place behind unchanged pointers is invisible to the key
int vpt_frame_ahead_invalidate(vpt_ctx* ctx) {
    if (!ctx) return VPT_E_INVALID;
    ctx->ahead.key_valid = false;
    ctx->ahead.n = 0;
    ctx->ahead.streak = 0;
    return VPT_OK;
}
int vpt_set_frame_ahead(vpt_ctx* ctx, int enable) {
    if (!ctx) return VPT_E_INVALID;
    ctx->ahead.off = enable == 0;
    return vpt_frame_ahead_invalidate(ctx);
}

int vpt_test_get_cache_state(vpt_ctx* ctx, int out[8]) {
    if (!ctx || !out) return VPT_E_INVALID;
    for (int i = 0; i < 8; ++i) out[i] = 0;
    if (!ctx->have_last_resolve) return VPT_OK;
    const ResolveParams& R = ctx->last_resolve;
    out[0] = R.sky_patch != nullptr;
    out[1] = R.never_traced != nullptr;
    out[2] = R.sky_dome != nullptr;
    out[3] = R.sky_dome != nullptr ? 2 * ctx->sky_dome_k + 1 : 0;
    out[4] = R.cam_tab_valid;
    out[5] = R.dir_tab != nullptr;
    out[6] = R.lean;
    out[7] = R.never_traced != nullptr && R.cull_tiles != nullptr;
    return VPT_OK;
}

int vpt_test_leaf_tiles(const vpt_camera* cam, const float root_lo[3], const float root_hi[3], const unsigned int occ[19], int width, int height,
                        float margin, unsigned char* tiles) {
    if (!cam || !root_lo || !root_hi || !occ || !tiles || width <= 0 || height <= 0) return VPT_E_INVALID;
    const double lo[3] = {root_lo[0], root_lo[1], root_lo[2]}, hi[3] = {root_hi[0], root_hi[1], root_hi[2]};
    std::vector<unsigned char> t;
    if (!build_leaf_tiles(cam, lo, hi, occ, (uint32_t)width, (uint32_t)height, (double)margin, t)) return VPT_E_UNSUPPORTED;
    std::memcpy(tiles, t.data(), t.size());
    return VPT_OK;
}

int vpt_test_count_never_traced(vpt_ctx* ctx, unsigned long long* pixels) {
    if (!ctx || !pixels) return VPT_E_INVALID;
    *pixels = 0;
    if (!ctx->have_last_resolve || ctx->last_resolve.never_traced == nullptr) return VPT_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipDeviceSynchronize());
    std::vector<unsigned char> h(ctx->last_resolve.n_pixels);
    HIPCHK(ctx, hipMemcpy(h.data(), ctx->last_resolve.never_traced, h.size(), hipMemcpyDeviceToHost));
    unsigned long long n = 0;
    for (unsigned char v : h) n += v != 0;
    *pixels = n;
    return VPT_OK;
}

}  // extern "C"
