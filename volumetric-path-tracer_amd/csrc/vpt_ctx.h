// vpt_ctx.h -- the context behind the C ABI (include/vpt_abi.h): shared by the ABI implementation (vpt_host.hip) and the per-view
// caches of the environment tail (vpt_caches.hip).  Host-side only.
#pragma once

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <string>
#include <vector>

#include "../../include/vpt_abi.h"
#include "../../include/vpt_testhooks.h"
#include "vpt_device.h"

namespace vpt {
hipError_t launch_trace(const TraceParams& P, bool multi, bool color, bool emit, int blocks, hipStream_t stream);
int trace_vol_blocks_per_cu(bool sky_in_tracer);      // sky_in_tracer: environment_type 0 (the SKYLUT instantiations)
size_t trace_vol_hist_floats_per_block();                          // > 0: the vol tracer keeps its first-walk density history in HBM (TraceParams::pool_hist)
int trace_blocks_per_cu();                                         // workgroups per CU the direct tracer is built for (its waves per SIMD)
#ifdef VPT_WITH_POOL              // study builds only (csrc/variants/vpt_trace_pool.hip, build.py --with-pool): the round-3 pool tracer
hipError_t launch_trace_pool(const TraceParams& P, bool multi, bool color, bool emit, int blocks, int threads, hipStream_t stream);
size_t trace_pool_hist_floats_per_block();
bool trace_pool_supports(const TraceParams& P);
#endif
hipError_t launch_trace_vol(const TraceParams& P, bool multi, bool color, bool emit, int blocks, hipStream_t stream);
hipError_t launch_raygen(const TraceParams& P, hipStream_t stream);
hipError_t launch_tail_resolve(const ResolveParams& R, hipStream_t stream);
hipError_t launch_tail_stream(const ResolveParams& R, hipStream_t stream);      // tail_stream_kernel alone
hipError_t launch_sky_fix(const ResolveParams& R, hipStream_t stream);          // sky_fix_kernel (before it, on the tracer's stream)
hipError_t launch_sky_cam_table(const ResolveParams& R, SkyView* view, float4* out, int k, hipStream_t stream);
size_t sky_cam_table_bytes();
size_t sky_dir_table_bytes();
hipError_t launch_sky_dir_table(const ResolveParams& R, SkyView* view, float4* tab, unsigned long long* err, int k, hipStream_t stream);
hipError_t launch_sky_samples(const ResolveParams& R, const float* origins, const float* dirs, float* out, uint32_t n, int use_table, hipStream_t stream);
hipError_t launch_sky_patch(const ResolveParams& R, float4* out, unsigned char* never, uint32_t* nopatch_list, uint32_t* nopatch_count, hipStream_t stream);
hipError_t launch_sky_dome(const ResolveParams& R, const SkyView* view, float4* out, int k, hipStream_t stream);
size_t sky_dome_bytes(int k);
hipError_t launch_display(const float* accum, unsigned int* display, float* raw, uint32_t n, float exposure_scale, hipStream_t stream);
hipError_t launch_blue_noise(float* bn, float2* table, uint32_t count, uint32_t stride, uint32_t live, hipStream_t stream);
}  // namespace vpt


struct TexEntry {
    vpt::DTexture t;
    void* owned;     // device allocation owned by the ctx (NULL when adopted)
    bool live;
};

struct Box { vpt::f3 lo, hi; };

// Everything the per-view caches of the environment tail are a function of, in ONE struct (compared bytewise: fill it from a zeroed
// one).  `tables`: the camera-point scattering tables, SkyView and the ground tables; the whole key: the sky patches, the never-traced
// pixel mask and the sky dome(s) built on top of them.
struct ViewKey {
    struct Tables {
        float view_pos[3], sun_dir[3], atm_f[40];
        int view_k;                        // table variants: one per binary32 step of r across an open lens (0: closed lens)
        const void* tex[4];                // ADDRESSES of the four look-up tables (their contents: vpt_invalidate_sky_tables)
    } tables;
    float frame[9];                        // camera: lower_left_corner, horizontal, vertical
    float width, height;
    float sky_mult, sky_color[3];
    float ground_table;                    // 0: ground hits in full; 1 + tolerance: through the ground tables
    float cull_enabled, render;            // never-traced pixels: on / off, kernel_params.render
    float cull_rect[4], cull_line[3], cull_sph[4];
    uint32_t cull_tiles_hash[2];           // FNV-1a of the leaf-level tile map (0, 0: none), i.e. of camera x octree occupancy
};

struct vpt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    int num_cus = 0;
    int blocks_per_cu = 0;         // VPT_BLOCKS_PER_CU: tracer workgroups per CU; 0 = what the tracer is built for (direct: 4, vpt_trace.hip; vol: 3)
    uint32_t regen_min = 8;        // direct_integrator tracer: refill once >= 8 lanes are idle
    uint32_t regen_min_vol = 1;    // vol_integrator tracer: walks are long (config 4: 71 steps per ray), refill at once
    uint32_t trans_min = 48;       // direct_integrator tracer: run the transition states once >= 48 lanes wait for them
    uint32_t trans_min_vol = 24;   // vol_integrator tracer (swept 8..48 on config 4)
    // The zero-footprint mask (DVolume::zmask) is OPT-IN (VPT_ZERO_MASK=1): exact, and it takes the look-ups' HBM traffic down, but its own dependent load ahead of the
    // two quad loads costs every config more than the skipped lines return (tracer +2.3 % on config 4, +1.3 % on 3, +3 % on 5, +1.7 % on 2: profiles/r06_zero_mask.txt)
    bool no_zero_mask = true;
    size_t zmask_min_bytes = (size_t)32 << 20;   // VPT_ZERO_MASK_MIN_BYTES: density grids at or above get a mask (below they stay cache resident: the mask's dependent load costs more than it saves)
    int zmask_shift = 0;              // VPT_ZERO_MASK_SHIFT: force the block edge (2..6: 4..64 origins; study switch)
    uint32_t chunk_entries = 0;       // VPT_CHUNK_ENTRIES: queue entries per claim (study switch; 0: 256, 128 for launches of a few iterations)
    int raygen_footprint = -1;        // VPT_RAYGEN_FOOTPRINT=rows|squares (study switch); -1: squares where the view has a never-traced mask, rows where it has none (round 6)
    uint32_t raygen_small_iters = 17; // VPT_RAYGEN_SMALL_ITERS: launches of fewer iterations run raygen over 16-row tiles (four times the blocks: 8 iterations 1.102 -> 1.045 ms, 16: 1.669 -> 1.611, 64: no difference; profiles/r05_batch_curve.txt)
    // pool tracer (csrc/variants/vpt_trace_pool.hip, study builds with -DVPT_WITH_POOL only): direct_integrator with the rays in an LDS pool per CU
    bool use_pool = false;         // VPT_TRACER=pool in such a build: measured slower than the lane-bound tracer, kept as the evidence and for A/B runs
    int pool_waves = 12;           // VPT_POOL_WAVES: waves of the one workgroup per CU (8..12)
    uint32_t pool_min_lanes = 40;  // VPT_POOL_MIN_LANES: fewest lanes a pass starts with while other waves still hold rays
    float* d_pool_hist = nullptr;
    size_t pool_hist_floats = 0;
    std::string last_error;
    std::vector<TexEntry> textures;
    // scene
    std::vector<vpt_gpu_vdb> host_volumes;
    std::vector<vpt::DVolume> host_dvolumes;
    vpt::DVolume* d_volumes = nullptr;
    float4* d_insts = nullptr;        // compact per-instance matrices (TraceParams::insts)
    bool single_file = false;
    std::vector<void*> bricked;       // re-tiled copies of large density grids (owned)
    float4* d_cam_tab = nullptr;      // camera-point scattering tables, (2 k + 1) x 8 x 128 x 2 float4 (vpt_sky.h)
    vpt::SkyView* d_sky_view = nullptr;    // their view point and variants
    float4* d_dir_tab = nullptr;      // view-point ground table (vpt_sky.h, GroundNode) ...
    unsigned long long* d_dir_err = nullptr;    // ... and its measured interpolation error (high word: float bits, low word: the cell)
    bool dir_tab_built = false;       // for the table part of view_built
    vpt::ResolveParams last_resolve;       // environment side of the last render (vpt_test_sky_samples)
    bool have_last_resolve = false;
    uint32_t* d_leaf_offsets = nullptr;
    uint32_t* d_leaf_indices = nullptr;
    uint32_t* d_sub_offsets = nullptr;     // single-file scenes: candidate lists per sub-cell of every leaf (512 * VPT_SUB3 + 1)
    float sub_inv[3] = {0.0f, 0.0f, 0.0f}; // VPT_SUB / leaf extent per axis
    uint32_t occ[19] = {0};
    Box root = {{0, 0, 0}, {0, 0, 0}};
    float max_ext = 0.0f, min_ext = 0.0f;
    int nonempty[3] = {0, 0, 0};
    bool scene_ready = false;
    bool any_color = false, any_emission = false;
    // scratch
    vpt::Record* d_records = nullptr;
    float4* d_heads = nullptr;             // 16-byte sample heads, same capacity as d_records
    float4* d_rays32 = nullptr;            // compact 32-byte ray records (TraceParams::rays32), same capacity, allocated on first use
    size_t rays32_capacity = 0;
    float2* d_td = nullptr;                // {alpha, depth} of the resolved samples (TraceParams::td), same capacity, allocated on first use
    size_t td_capacity = 0;
    uint32_t* d_queue2 = nullptr;          // record slots for sky_fix_kernel (TraceParams::queue2), same capacity, allocated with d_td
    uint32_t* d_nopatch = nullptr;         // [0]: count, [1..]: pixels without a usable sky patch (ResolveParams::nopatch_list), with d_sky_patch
    uint32_t piece_max = 512;              // VPT_PIECE_MAX: records per piece of the tracer's queue at most (vpt_device.h: queue of pieces); 0: a queue of entries
    bool last_queue_pieces = false;        // the last launch's queue held pieces (get_stats: where its ray count stands)
    uint32_t piece_waves_div = 32;         // VPT_PIECE_DIV: a piece is at most (samples left in the launch) / (waves of the tracer x this)
    bool no_compact_rays = false;          // VPT_NO_COMPACT_RAYS: 64-byte ray records behind a closed lens too (tests: same bits either way)
    bool no_lens_lean = false;             // VPT_NO_LENS_LEAN: behind an open lens the samples keep their records and the tail adds the environment (tests, A/B)
    bool no_lean_tail = false;             // VPT_NO_LEAN_TAIL: finished paths keep their 64-byte records and the tail adds the environment (A/B, tests)
    float4* d_head_org = nullptr;          // ray origins of the heads (thin lens: lens_radius != 0), allocated on first use
    size_t head_org_capacity = 0;
    size_t records_capacity = 0;           // in records
    float2* d_bn_table = nullptr;
    size_t bn_capacity = 0;                // in iterations
    uint32_t* d_work_counter = nullptr;     // [0] the tracer's dequeue cursor, [8] raygen's queue tail (own cache line apart)
    uint32_t* d_queue = nullptr;
    float* d_vdc = nullptr;
    vpt::Counters* d_counters = nullptr;
    vpt::DPointLight* d_lights = nullptr;
    size_t lights_capacity = 0;
    std::vector<vpt::DPointLight> lights_cache;
    // multi-GPU: one RCCL communicator per context (vpt_comm_init_rank); librccl.so is loaded on first use, so a
    // single-GPU host never maps it
    ncclComm_t comm = nullptr;
    int comm_nranks = 0, comm_rank = 0;
    float* d_comm_buf = nullptr;           // the collective's payload: this rank's weighted image + its iteration count in the last float
    size_t comm_buf_floats = 0;
    // tuning / test switches, read from the environment ONCE, when the context is created (vpt_create)
    size_t relaid_min_bytes = 0;                  // VPT_RELAID_MIN_BYTES: density / emission grids below this stay dense.  0 since round 5: the corner quads are two loads and a third of the address arithmetic of the dense layout's eight -- also for a grid that lives in L2 (config 2's 425 KB dragon: tracer -1.2 %, profiles/r05_compact_rays.txt); 8 MiB in rounds 2-4
    int grid_layout = -1;                  // VPT_GRID_LAYOUT (tests): force "dense" / "bricks" / "quads" for grids >= relaid_min_bytes; -1: quads, bricks if those do not fit
    bool force_no_addr24 = false;          // VPT_NO_ADDR24: tests force the 32-bit texel index arithmetic
    unsigned batch_iters = 0;              // VPT_BATCH_ITERS: iterations per record chunk (0 = the 16-GiB rule)
    bool no_heads = false;                 // VPT_NO_HEADS: every sample gets a 64-byte record (tests)
    bool no_cam_table = false;             // VPT_NO_CAM_TABLE: general sky look-ups only (tests)
    bool no_leaf_cull = false;             // VPT_NO_LEAF_CULL: the never-traced mask from the root box's bounds alone (tests: same image either way)
    bool tex_fixed8 = false;               // VPT_TEX_WEIGHTS=fixed8: diagnostic model of the CUDA texture unit's 1.8 fixed-point weights (vpt_trace_common.h: make_taps)
    bool no_fast_div = false;              // VPT_NO_FAST_DIV: every look-up divides by the grid extent (tests: both forms give the same bits)
    bool no_dir_table = false;             // VPT_NO_DIR_TABLE: every ground hit evaluated in full (tests)
    float dir_tab_tol = 5e-4f;             // VPT_DIR_TABLE_TOL: largest relative mid-cell error the ground table may show
    // the per-view caches of the environment tail (vpt_caches.hip): ONE key -- what they were built for -- and one flag per layer
    ViewKey view_built;                    // valid as far as the flags below say
    ViewKey view_seen;                     // the previous render call's key, built or not (a view that repeats gets its patches)
    bool view_seen_valid = false;
    bool cam_tab_built = false;            // camera-point scattering tables (+ SkyView) for view_built's table part
    // per-pixel sky patches of the untraced samples (ResolveParams::sky_patch): rebuilt when the sky tables or the camera frame change
    float4* d_sky_dome = nullptr;              // sky dome(s) (ResolveParams::sky_dome), rebuilt with the patches (closed lens) / the tables (open lens)
    int sky_dome_k = -1;                       // variants the allocation holds: 2 k + 1
    bool lens_dome_built = false;              // open lens: domes valid for the current camera-point tables
    bool no_sky_dome = false;                  // VPT_NO_SKY_DOME (tests)
    float4* d_sky_patch = nullptr;
    unsigned char* d_cull_tiles = nullptr;     // ResolveParams::cull_tiles (8x8-pixel tiles some non-empty octree leaf may be seen through)
    size_t cull_tiles_bytes = 0;
    std::vector<unsigned char> cull_tiles_host;
    hipEvent_t cull_tiles_copied = nullptr;    // recorded behind the map's upload: the host copy is not rewritten before it has been read
    bool cull_tiles_copy_pending = false;
    float cull_tiles_inputs[9 + 3 + 6 + 2 + 19] = {};   // what the host map was built from (camera frame + origin, root box, image size, occupancy words)
    bool cull_tiles_inputs_valid = false;
    uint32_t cull_tiles_hash[2] = {0, 0};
    size_t cull_tiles_covered = 0;
    unsigned char* d_never_traced = nullptr;   // per pixel: raygen emits nothing, the tail has the values (ResolveParams::never_traced)
    size_t sky_patch_pixels = 0;           // capacity, in pixels
    bool sky_patch_built = false;
    bool no_sky_patch = false;             // VPT_NO_SKY_PATCH: every untraced sample evaluated in full (tests)
    bool no_pixel_cull = false;            // VPT_NO_PIXEL_CULL: raygen emits every pixel's samples (tests)
    // One context's renders share its record / head / queue buffers and its per-view caches, so they are SERIALISED: a render issued on
    // another stream than the previous one first waits (on the device) for that one's last kernel.
    hipEvent_t render_event = nullptr;     // recorded behind every render's last kernel
    hipStream_t render_stream = nullptr;   // the stream of the previous render
    hipEvent_t comm_event = nullptr;       // recorded behind every vpt_allreduce_accum's last kernel
    hipStream_t comm_stream = nullptr;     // the stream of the previous collective
    // FRAME-AHEAD (round 5): the reference's host loop is ONE launch + device sync per iteration (main.cpp:1822-1829), and a one-iteration launch of the
    // persistent tracer lasts as long as its longest paths whatever the machine could do meanwhile (0.19 of the frame's 0.35 ms at 1080p).  A still camera
    // repeats the same call with iteration + 1: from the second such call on, a call traces the rays of the NEXT iterations as well (2, 4, 8, ... up to
    // max_k per call, one raygen + tracer launch at the batch's occupancy) but accumulates only its own; the following calls find their rays traced and
    // run only their tail.  Every sample is the same Philox-keyed function of (pixel, iteration) either way and the running means take the iterations in
    // the same order: every buffer after every frame is bit-identical to the frame-by-frame sequence (tests/test_gpu_edge.py).  A call that does not
    // continue the sequence (another camera, light, parameter, buffer, stream; a batch; a scene or texture change) discards what was traced ahead: the
    // cost of a camera move is the <= max_k iterations of rays traced for nothing, once.
    struct FrameAhead {
        bool off = false;                  // VPT_NO_FRAME_AHEAD
        unsigned max_k = 16;               // VPT_FRAME_AHEAD_MAX
        bool key_valid = false;
        std::vector<unsigned char> key;    // what a one-iteration call is a function of, its iteration index excluded (bytes of the argument structs)
        unsigned last_it = 0;              // iteration of the previous call
        unsigned streak = 0;               // consecutive calls with the same key and iteration = previous + 1
        unsigned it0 = 0, n = 0, next = 0; // rays of iterations it0 .. it0 + n - 1 are traced (buffer set 0); slices [next, n) still wait for their tail
        float* d_bn = nullptr;             // private copy of the caller's blue-noise state the batch's jitter tables are generated from
    } ahead;
    bool counters_dirty = true;            // d_counters holds counts of an earlier counted render
    // stats
    bool counting = false;
    std::vector<hipEvent_t> ev_pool;
    struct Span { int e0, e1; int kind; };
    std::vector<Span> spans;
    int ev_used = 0;
    unsigned long long last_samples = 0;
};

void vpt_set_error(vpt_ctx* ctx, const char* fmt, ...);
#define set_error vpt_set_error
#define HIPCHK(ctx, expr)                                                                        \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            vpt_set_error(ctx, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return VPT_E_HIP;                                                                    \
        }                                                                                        \
    } while (0)

// screen-space bounds (pixels) of the world box [lo, hi] through the closed-lens camera (vpt_host.hip); false: a corner at or behind the camera plane
extern "C" bool vpt_project_box(const vpt_camera* cam, const double lo[3], const double hi[3], double W, double H, double rect[4]);
// the per-view caches for this render: builds what is stale, points R / P at what is in use (vpt_caches.hip)
int vpt_view_caches_prepare(vpt_ctx* ctx, const vpt_camera* cam, const vpt_sphere* ref_sphere, const vpt_kernel_params* kp, bool compact,
                            unsigned int iter_count, vpt::ResolveParams& R, vpt::TraceParams& P, hipStream_t stream);
