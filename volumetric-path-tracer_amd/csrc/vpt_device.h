// vpt_device.h -- device-side views of the scene and the per-launch parameter blocks.
//
// HBM layout (DESIGN.md, section Data layout):
//   * density / emission grids: dense f32, x fastest (LayoutXYZ of the reference's
//     copyToDense, gpu_vdb.cpp:179-212); colour grids: dense float4;
//   * the 3-level instance octree is stored as three occupancy bit-sets (8 + 64 + 512
//     bits) plus a CSR list of instance indices per leaf -- child boxes are re-derived
//     arithmetically by halving the parent box (bvh_kernels.cu:150-202 divide_bbox), which
//     reproduces the reference's node boxes bit-for-bit without the 2.5 KB OCTNode;
//   * density and emission grids (>= VPT_RELAID_MIN_BYTES; default 0 since round 5, 8 MiB before: what does not stay in L2) are
//     re-laid once as CORNER QUADS (GRID_QUADS): entry (x, j, k), j in [-1, dy-1], k in [-1, dz-1], is the float4
//     (c(x,j,k), c(x,j+1,k), c(x,j,k+1), c(x,j+1,k+1)) with clamped j, k -- a trilinear footprint is then two float4
//     loads from ONE 32-byte run (x, x+1) instead of eight dwords on four scattered rows: 4x the memory (288 GB
//     of HBM is what it is spent on), about half the HBM traffic per look-up (1.1 lines of 128 B instead of ~2.5 with half of
//     them hitting L2).  Density grids whose quads would not fit fall back to 4x4x4 bricks of 256 B (GRID_BRICKS);
//   * path records: one 64-byte line per pixel-sample, written once by the trace
//     kernel, read once by the resolve kernel.
#pragma once

#include "vpt_math.h"

namespace vpt {

enum { GRID_DENSE = 0, GRID_BRICKS = 1, GRID_QUADS = 2 };

// view point of the environment tail's per-frame tables (vpt_sky.h), device-resident
enum { SKY_VIEW_MAX_K = 4 };
#ifndef VPT_SKY_DOME_NU
#define VPT_SKY_DOME_NU 2048     // (4096 x 2048 and 1024 x 512 measured: the same tail time; 2048 x 1024 = 33 MB, 0.18-degree cells)
#define VPT_SKY_DOME_NV 1024
#endif
enum { SKY_DOME_NU = VPT_SKY_DOME_NU, SKY_DOME_NV = VPT_SKY_DOME_NV };          // sky dome nodes (ResolveParams::sky_dome): float4 each
enum { SKY_DIR_ERR_STRIDE = 6 };      // u64 words per table variant: worst ray (key), rays, unflipped rays above 1e-3, flipped rays, their summed deviation (2^-24 units), spare
enum { SKY_DIR_ERR_WORDS = 8 + SKY_DIR_ERR_STRIDE * (2 * SKY_VIEW_MAX_K + 1) };   // u64 words of the ground table's build-time check (vpt_tail.hip: launch_sky_dir_table)
struct SkyView {
    float r, mu_s;         // of the camera origin, with the tail's own arithmetic
    int k;                 // variants: r + (-k .. k) binary32 steps
    int pad_;
    float4 tab[2 * SKY_VIEW_MAX_K + 1];   // per variant: 1 / (r - bottom), 1 / log2(horizon distance / (r - bottom)), log-distance coordinate up to
                                          // which the ground table is used (and was validated), 1 if the variant has a ground table
};

// WRITE-ONCE / READ-ONCE streams (ray and path records, sample heads, {alpha, depth} pairs, queue entries: 8.5 GB per 64-iteration launch at
// 1080p): accessed through these helpers so that a build can mark them NON-TEMPORAL (-DVPT_NT_STREAMS: the `nt` bit of global_load /
// global_store -- the lines are first in line for eviction and do not push the grid, the tables and the dome out of L2 and the Infinity
// Cache).  Same bytes either way; measured per config in profiles/r05_*.
typedef float vpt_v4f __attribute__((ext_vector_type(4)));
typedef float vpt_v2f __attribute__((ext_vector_type(2)));
#ifdef VPT_NT_STREAMS
VPT_D float4 ld_stream(const float4* p) { const vpt_v4f v = __builtin_nontemporal_load(reinterpret_cast<const vpt_v4f*>(p)); return make_float4(v.x, v.y, v.z, v.w); }
VPT_D float2 ld_stream(const float2* p) { const vpt_v2f v = __builtin_nontemporal_load(reinterpret_cast<const vpt_v2f*>(p)); return make_float2(v.x, v.y); }
VPT_D uint32_t ld_stream(const uint32_t* p) { return __builtin_nontemporal_load(p); }
VPT_D void st_stream(float4* p, float4 v) { vpt_v4f t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w; __builtin_nontemporal_store(t, reinterpret_cast<vpt_v4f*>(p)); }
VPT_D void st_stream(float2* p, float2 v) { vpt_v2f t; t.x = v.x; t.y = v.y; __builtin_nontemporal_store(t, reinterpret_cast<vpt_v2f*>(p)); }
VPT_D void st_stream(uint32_t* p, uint32_t v) { __builtin_nontemporal_store(v, p); }
VPT_D uint2 ld_stream(const uint2* p) { const float2 v = ld_stream(reinterpret_cast<const float2*>(p)); return make_uint2(__float_as_uint(v.x), __float_as_uint(v.y)); }
VPT_D void st_stream(uint2* p, uint2 v) { st_stream(reinterpret_cast<float2*>(p), make_float2(__uint_as_float(v.x), __uint_as_float(v.y))); }
#else
VPT_D float4 ld_stream(const float4* p) { return *p; }
VPT_D float2 ld_stream(const float2* p) { return *p; }
VPT_D uint32_t ld_stream(const uint32_t* p) { return *p; }
VPT_D void st_stream(float4* p, float4 v) { *p = v; }
VPT_D void st_stream(float2* p, float2 v) { *p = v; }
VPT_D void st_stream(uint32_t* p, uint32_t v) { *p = v; }
VPT_D uint2 ld_stream(const uint2* p) { return *p; }
VPT_D void st_stream(uint2* p, uint2 v) { *p = v; }
#endif

struct DVolume {
    const float* density;
    const float* emission;
    const f4* color;
    float m[12];   // world->index rows: q.x = m[0]*x + m[1]*y + m[2]*z + m[3], ...
    float bmin[3];
    float fdim[3];
    int dim[3];            // density texture extent
    int has_color;
    int has_emission;
    int edim[3];           // emission texture extent
    int cdim[3];           // colour texture extent
    int layout;            // of the density grid: GRID_DENSE (x fastest), GRID_BRICKS (4x4x4 bricks of 256 B, x fastest inside a
                           // brick, bricks x fastest) or GRID_QUADS (float4 corner quads, x fastest, (dy+1)(dz+1) rows)
    int bdim[2];           // GRID_BRICKS: bricks along x and y
    int elayout;           // of the emission grid: GRID_DENSE or GRID_QUADS
    int addr24;            // every texel-index product of this volume's grids fits the 24-bit multiplier (see imul)
    float rdim[3];         // RN(1 / fdim)
    int fast_div;          // q / fdim may be formed as y + (q - fdim y) rdim, y = q rdim: the host checked all three extents (vpt_fastdiv.h)
    // (float) of the three texture extents, host-converted: the compiler keeps a launch-uniform conversion in a VECTOR register for the whole
    // kernel (gfx950 has no scalar float unit); as descriptor fields they are scalar operands (profiles/r04_four_waves.txt (i))
    float dimf[3], edimf[3], cdimf[3];
    // ZERO-FOOTPRINT MASK of the density grid (round 6; null: none).  A trilinear footprint is the 2x2x2 texels at origin (i, j, k) = floor(u * dim - 0.5), each in
    // [-1, dim - 1], clamp addressing applied.  Bit (bx, by, bz) -- b = (origin + 1) >> zshift per axis -- is set when EVERY footprint whose origin lies in the block
    // consists of eight texels that are exactly 0: such a look-up returns 0 + (0 - 0) * a = +0 whatever the weights (bit-exact), so the two quad loads (128-byte
    // lines that are almost never reused: profiles/r05_c4_pmc.txt, 4.3x the must-move bytes) are skipped.  Word (bx >> 5, by, bz), x fastest; built next to the re-lay.
    const uint32_t* zmask;
    int zshift, znwx, znby;
};

struct DTexture {          // CUDA sampler state restated
    const float* data;
    int width, height, depth, channels;
    int normalized, linear;
    int addr[3];
};

// 64-byte path record: trace -> resolve
struct __attribute__((aligned(16))) Record {
    float L[3];
    float tr;
    float beta[3];
    float depth;
    float env_pos[3];
    uint32_t flags;        // bit0: sample was traced (render && iteration < max_interactions)
    float dir[3];
    float pad_;
};
static_assert(sizeof(Record) == 64, "Record must be one 64-byte line");

struct DPointLight {       // vpt_point_light
    float pos[3];
    float dir[3];
    float power;
    float color[3];
};

struct DCamera {           // vpt_camera, same field order
    float time1, time0;
    float origin[3];
    float focus_dist;
    float llc[3];
    float horizontal[3];
    float vertical[3];
    float u[3], v[3], w[3];
    float lens_radius;
    unsigned char viz_dof;
};

// words of vpt_ctx::d_work_counter: [0] the tracer's dequeue cursor, [4] second queue's tail, [8] raygen's queue tail, [32 + 32 c] claim counter c (128 bytes apart)
#define VPT_WORK_COUNTER_WORDS (32 + 32 * 32)
// QUEUE-ORDERED compact ray records (round 6; 0 = rounds 5's slot-indexed ones, the A/B switch): a compact record stands where its sample stands in its raygen block's
// queue -- block b owns records [b x 64 x ROWS, ...) of `rays32`, in enqueue order -- and holds {position reached | (t_hit, depth, t_box), packed word} + {direction, slot}:
// the tracer re-generates the Philox block from the counter in the word (one block per refilled lane) instead of reading it, reads no head, and the records of a refill
// are one contiguous run instead of three scattered 16-byte loads per ray.  The queue entry is the record's place.  Config 2 tracer -6 %, config 3 -3.6 %, config 4 -1.6 %.
#ifndef VPT_QREC
#define VPT_QREC 1
#endif
struct Counters {
    unsigned long long samples;
    unsigned long long density_lookups;
    unsigned long long color_lookups;
    unsigned long long emission_lookups;
    unsigned long long tracking_steps;
    unsigned long long skip_steps;
    // schedule histogram of the tracer (counting builds only; wave-level sums, see vpt_testhooks.h):
    // [0] loop passes, [1] walking lanes, [2] lanes parked in transition states, [3] idle lanes,
    // [4] passes that ran transitions, [5] inner transition passes, [6] lanes in them,
    // [7] lanes that executed the tracking step proper (after the empty-node loop)
    unsigned long long sched[8];
    // [0] refill, [1] Philox top-up, [2] walk step, [3] transitions: wave-level shader-clock cycles (s_memtime)
    unsigned long long cycles[4];
    // spatial coherence of the density look-ups of single-volume scenes (counting builds only; wave-level sums):
    // [0] wave-level look-up events, [1] lanes taking part, [2] distinct 8x8x8-voxel bricks among those lanes,
    // [3] distinct 4x4x4 bricks, [4] distinct 128-byte lines over one lane's 8 taps summed over lanes,
    // [5] distinct 128-byte lines over ALL taps of the wave (what one gather pass asks the L1 for)
    unsigned long long coh[8];
    // trilinear fetches actually issued (the point lies inside the instance's look-up domain and the value is used):
    // [0] density, [1] colour (float4 texels), [2] emission -- what the tracer really moves, as opposed to the
    // reference-defined look-up counts above (one per instance of the leaf and step, fetched or not)
    unsigned long long fetches[4];     // ([3]: density fetches the zero-footprint mask answered instead -- counted in [0] too)
    // vol_integrator's runs of empty sample() calls (vpt_walk.h: use_retries; counting builds only; wave-level sums of LANE-passes through
    // the tracking step proper): [0] lane-passes that reached a density look-up, [1] lane-passes that ended in retry spins only (the
    // Philox words buffered for the pass ran out, or VPT_RETRY_SPINS), [2] retry draws in all, [3] lane-passes whose walk ended at t >= distance
    unsigned long long retry[4];
};

#ifndef VPT_CHUNK
#define VPT_CHUNK 256        // queue entries a wave claims per global atomic (<= 256: a lane buffers 4 entries); TraceParams::chunk
#endif
#ifndef VPT_SUB
#define VPT_SUB 4            // sub-cells per leaf and axis of the refined candidate lists (host builder and tracer agree on it)
#endif
#define VPT_SUB3 (VPT_SUB * VPT_SUB * VPT_SUB)

struct ResolveInTracer {
    const float4* sky_dome;          // ResolveParams::sky_dome; NULL: off (every finished path writes its 64-byte record, the tail adds the environment)
    float4* heads;                   // == TraceParams::heads
    float2* td;                      // [iter_count][n_pixels] {alpha, depth} of the resolved samples
    uint32_t* queue2;                // record slots whose environment term needs the full evaluation
    uint32_t* queue2_tail;
    float cam_origin[3];
    // OPEN LENS (round 5): every sample starts somewhere on the lens disc and the sky sees that origin only through (r, mu_s) -- one dome per table variant
    // (SkyView: one per binary32 value of r within k steps of the camera origin's).  `lens` != 0: the dome that serves a sample is picked from its origin
    // (dome_variant, vpt_dome.h) instead of requiring the camera origin bit for bit; raygen resolves the UNTRACED samples the same way.
    int lens;
    const SkyView* sky_view;
    float sun_dir[3];
    float earth_bottom;              // AtmosphereParameters::bottom_radius
};

struct TraceParams {
    // work distribution
    uint32_t width, height, n_pixels;
    float inv_n_pixels;              // 1 / n_pixels (fp32), see split_slot
    float rcp_w, rcp_h;              // RN(1 / width), RN(1 / height)
    int tex_fixed8;                  // VPT_TEX_WEIGHTS=fixed8 (diagnostic): the grid look-ups quantise their interpolation weights to 1/256 (make_taps)
    int fast_uv;                     // get_ray's u and v may be formed with them (both extents checked, vpt_fastdiv.h)
    uint32_t iter_begin, iter_stride, iter_count;
    uint32_t max_interactions;
    int render;
    uint32_t regen_min;              // refill when at least this many lanes of a wave are idle
    uint32_t trans_min;              // run the transition states when at least this many lanes wait for them
    uint32_t chunk;                  // queue entries a wave claims per global atomic: VPT_CHUNK, half of it for launches of a few iterations
    // QUEUE OF PIECES (round 6; with queue-ordered records, piece_max != 0): the queue holds {first record, count} pairs instead of one word per ray -- the records of a raygen
    // block are consecutive, so a run of them needs no list.  A claim is then ONE piece: no dependent load of 256 entries, no entries kept in registers, and its size is free:
    // a block cuts its run into even pieces of at most clamp(samples of this and the later blocks / piece_div, piece_min, piece_max) records -- large pieces while the
    // launch has plenty of work left, small ones towards its end (raygen's blocks append in launch order), so that the last pieces finish together.  `chunk` is 1 then.
    uint32_t piece_min, piece_max, piece_div;
    uint32_t raygen_small_iters;     // launches of fewer iterations run raygen over 16-row tiles (four times the blocks)
    uint32_t raygen_squares;         // a raygen wave covers an 8 x 8 pixel square (a never-traced mask exists: live or skipped as a whole) or 64 pixels of a row (no mask: 1 KB store runs)
    // COMPACT RAY RECORDS (round 5; closed lens + heads): a queued ray's record is 32 bytes instead of 64 -- {a.x, a.y, a.z, word} + the Philox block, where
    // a = the position raygen's empty-node pushes reached, or (t_hit, depth, t_box), and word = obj [0:6] | advanced [7] | pushes [8:13] | Philox word index
    // [14:16] | Philox blocks since the sample's stream origin [17:31].  The origin is the camera's, the direction is the sample's 16-byte head (written
    // anyway), the counter follows from the iteration: raygen writes 48 bytes per traced sample instead of 80 (config 3's raygen is store-bound), the refill
    // reads 48 instead of 64.  The compact records live in their OWN dense array (`rays32`, 32-byte stride): written into the first half of the 64-byte record slots
    // they cost raygen +10-17 % (half-filled lines: profiles/r05_compact_rays.txt); records[] keeps the 64-byte path records of the paths the dome cannot serve.
    int compact_rays;
    float4* rays32;                  // VPT_QREC: [raygen block][place in the block's queue][2] over the image padded to 64 x 64 tiles; else [iter_count][n_pixels][2]
    uint32_t* work_counter;          // [0] next queue entry the tracer hands out (claim_chunk, vpt_trace_common.h); [32 + 32 c]: the interleaved claim counters (VPT_CLAIM_COUNTERS)
    uint32_t* queue;                 // [n_pixels*iter_count] record slots of the rays to trace (compacted); piece_max != 0: [pieces][2] = {first record, count}
    uint32_t* queue_tail;            // raygen's append cursor (entries, or pieces; then [1] counts the rays)
    const uint32_t* queue_count;     // == queue_tail, read by the tracer
    Record* records;                 // [iter_count][n_pixels]
    float4* heads;                   // [iter_count][n_pixels] 16-byte sample heads, or NULL (see ResolveParams)
    float4* head_org;                // ray origins of the heads when lens_radius != 0, else NULL (origin = camera)
    const float2* blue_noise;        // [iter_count][65536] (x,y) jitter of each iteration
    Counters* counters;              // may be NULL
    Counters* prof;                  // section cycle counters of -DVPT_PROFILE_SECTIONS builds (else unused)
    const unsigned char* never_traced;   // [n_pixels] or NULL: 1 = no primary ray of this pixel can start a walk and its samples' values come
                                     // from the pixel's sky patch (ResolveParams::never_traced): raygen emits nothing for it
    // RESOLVED SAMPLES (closed lens, direct integrator, procedural sky, per-view caches in use; ResolveParams::td): the tracer itself adds a
    // finished path's environment term -- a sky-dome look-up along its exit direction, at the ~44 lanes a transition pass finishes together --
    // and writes the sample as a 16-byte head {value, -1} + 8 bytes {alpha, depth} instead of a 64-byte path record; a path the dome cannot
    // serve (a flagged cell, an origin moved by the sphere bounce) keeps its record and its slot goes into queue2 for sky_fix_kernel.
    // What that takes travels BY VALUE in the kernel-argument segment and is read where a batch of paths finishes (load_resolve, vpt_trace_common.h:
    // scalar loads through the laundered segment pointer, like ColdConst -- not carried in the loop's live scalars).  Round 5: it sat behind a pointer
    // to device memory before, uploaded from pageable host memory per view change (a hidden host synchronisation for a moving camera), and the
    // finishing lanes read its fields with five dependent VECTOR loads.
    struct ResolveInTracer resolve;
    float* pool_hist;                // pool tracer (vpt_trace_pool.hip): density histories of the fused first walk, [workgroup][entry][ray]
    const float* vdc_tables;         // [2][101]: van der Corput radical inverses, bases 2 and 3
    // camera
    DCamera cam;
    // octree / scene
    float root_pmin[3], root_pmax[3];
    float max_ext, min_ext;
    // launch-uniform values of the walk, host-evaluated (IEEE single: the same bits as the device's correctly rounded divide; a uniform quotient
    // formed in the kernel lives in a vector register for the whole launch): 1 / max_ext (:1645), 1 / density_mult (:1646),
    // 1 / (max_ext - min_ext) (:1165), and the root's centre (root_pmin + root_pmax) * 0.5 (the first octree split)
    float inv_max_ext, inv_density_mult, sigma_r_inv;
    float root_mid[3];
    uint32_t occ[19];                // [0]: level-1, [1..2]: level-2, [3..18]: level-3 occupancy
    const uint32_t* leaf_offsets;    // 513 CSR offsets (multi-volume scenes)
    const uint32_t* leaf_indices;
    const DVolume* volumes;
    int num_volumes;
    // instances of ONE file (the .ins use case: main.cpp:1060-1095 copies the file's VDB_INFO and
    // textures into every instance and changes only the transform): the per-instance descriptor is then
    // just the 3x4 world->index matrix, 48 bytes in insts[] (stride 64), and everything else is vol0 in
    // SGPRs -- a quarter of the bytes the texture-data path has to return per instance visited
    const float4* insts;             // [sub-cell list entry][4]: matrix rows of that entry's instance, {0,0,0,0}
    const uint32_t* sub_offsets;     // [512 leaves * VPT_SUB3 sub-cells + 1]: CSR of the refined candidate lists (vpt_scene_set_volumes)
    float sub_inv[3];                // VPT_SUB / leaf extent: (p - leaf_lo) * sub_inv -> sub-cell coordinate in [0, VPT_SUB)
    int octree_full_single;          // one volume and no empty octree node: point location is the root test alone
    int single_file;
    int addr24;                      // every volume has DVolume::addr24: the tracer's A24 instantiation is launched
    DVolume vol0;                    // copy of volumes[0]: single-volume fast path reads it from SGPRs
    // reference sphere
    float sph_center[3];
    float sph_radius;
    float sph_color[3];
    float sph_roughness;
    // lights
    const DPointLight* lights;
    int num_lights;
    // Kernel_params subset
    int ray_depth, volume_depth;
    float phase_g1;
    float albedo[3], extinction[3];
    float tr_depth, density_mult;
    float emission_scale, emission_pivot;
    float sun_color[3];
    float sun_mult;
    float sun_dir[3];                // degree_to_cartesian(azimuth, elevation), host-evaluated
    float sun_inv[3];                // 1 / sun_dir per component, host-evaluated (IEEE single = the device's correctly rounded quotient): the sun's shadow rays (Tr prologue)
    float energy_inject;             // float(kernel_params.energy_inject)
    const float* emission_lut;       // float3[256]
    const float* density_color_lut;  // float3[256]
    unsigned int environment_type;
    // vol_integrator (integrator != 0) only: uniform_sample_one_light / estimate_sky inputs
    int integrator;
    float sky_mult;
    int env_sample_tex_res;
    float env_marginal_int;
    DTexture env_tex;                                      // lat-long HDRI (environment_type 1)
    DTexture env_func_tex, env_cdf_tex;                    // 2-D f32, unnormalised, point
    DTexture env_marginal_func_tex, env_marginal_cdf_tex;  // 1-D f32, unnormalised, point
    int has_atmosphere;
    int cam_tab_valid;                                     // always 0 in the tracer (estimate_sky looks from the interaction point)
    const float4* cam_tab;
    float cam_tab_pos[3];
    const float4* dir_tab;                                 // always NULL in the tracer
    const SkyView* sky_view;                               // always NULL in the tracer     // 1 / (r - bottom), 1 / log2(horizon distance / (r - bottom))
    float atm_f[40];                                       // packed vpt_atmosphere_parameters scalars
    DTexture transmittance_tex, scattering_tex, irradiance_tex, single_mie_tex;
};

struct ResolveParams {
    uint32_t width, height, n_pixels;
    float inv_n_pixels;              // 1 / n_pixels (fp32), see split_slot
    uint32_t iter_begin, iter_stride, iter_count;
    uint32_t max_interactions;
    // RN64(1 / n) for the running-mean divisor n = (float)(iter_begin / iter_stride + k + 1) of the launch's k-th iteration (a launch holds at
    // most 64), or 0 where n >= 2^27: the tail forms the quotients a / n as RN32(a * rcp_n[k]) (vpt_tail.hip: mul1_rn, with the proof)
    double rcp_n[64];
    const Record* records;
    // 16-byte sample heads (only when every primary ray starts at the camera origin, i.e. lens_radius
    // == 0): {dir0.xyz, w}.  w >= 0: the primary ray started no walk ("miss", 59 % of config 2) -- it
    // is final with L = 0, beta = 1, depth = w, env_pos = cam_origin and has NO 64-byte record;
    // w == -1: see records[]; w == -2: not rendered (value WHITE).  Cuts the record stream from 64 to
    // ~42 B per sample.
    const float4* heads;
    const float4* head_org;          // origin of a head's ray when the lens is open (thin-lens offset), else NULL: cam_origin
    float cam_origin[3];
    // Per-pixel SKY PATCH (vpt_tail.hip: sky_patch_kernel; closed lens + direct integrator + procedural sky): the value of an untraced
    // sample is a function of its primary direction alone, i.e. of the pixel and the sample's jitter (jx, jy) in [0,1)^2 -- smooth
    // across one pixel except where the horizon or the sun's disc cuts through it.  The patch holds that value at the pixel's four
    // corners (3 float4 per pixel: v00.rgb v10.rgb v01.rgb v11.rgb); a pixel whose exact centre value the bilinear patch misses by
    // more than 1e-3 (relative), or that lies within a pixel of the sun's disc, is marked (first word NaN) and evaluates every sample
    // in full.  Untraced samples of the other pixels -- 70 % of config 2's samples -- cost one jitter look-up and nine FMAs instead of
    // ~250 instructions of sample_atmosphere.  VALUE-ONLY, like the ground table: a cache of a pure function with a measured bound.
    const float4* sky_patch;         // [n_pixels][3], or NULL
    // SKY DOME (vpt_tail.hip: sky_dome_kernel; same conditions as the patches): a TRACED sample's environment term is
    // beta x sample_atmosphere(env_pos, dir) x sky_mult x sky_color with env_pos = the camera origin unless the path bounced off the
    // sphere, i.e. a function of the exit direction alone -- tabulated once per view over the whole sphere of directions
    // (SKY_DOME_NU x SKY_DOME_NV = 2048 x 1024 nodes: rows uniform in dir.y, columns in the L1 azimuth x / (|x| + |z|), no trigonometry either way),
    // with one flag per CELL: its bilinear interpolant reproduces the exact value at the cell's centre to 1e-3, no probe of the cell is
    // a ground hit evaluated in full, the sun's disc is more than a cell away.  A sample whose direction falls into a flagged cell
    // costs four float4 reads and nine FMAs instead of sample_atmosphere; the others are evaluated as before.  VALUE-ONLY.
    const float4* sky_dome;          // [SKY_DOME_NV][SKY_DOME_NU] {value.rgb, cell flag}, or NULL
    const float2* blue_noise;        // [iter_count][65536]: the chunk's jitter tables (what raygen read), for the patch
    // RESOLVED SAMPLES (TraceParams::resolve): with the per-view caches in use behind a closed lens the tail proper (tail_stream_kernel) evaluates
    // no sky at all -- a sample is a head {dir0, depth >= 0} (untraced: the pixel's patch at its jitter), {-, -, -, -2} (not rendered) or
    // {value, -1} + td {alpha, depth} (resolved by the tracer's dome look-up or by sky_fix_kernel) -- and streams its 16 + 8 bytes per sample
    // several iterations ahead of the ordered running means.  sky_fix_kernel runs between tracer and tail over (a) queue2, the paths the dome
    // could not serve, and (b) the untraced samples of the pixels WITHOUT a usable patch (nopatch_list, written with the patches).
    int lean;
    float2* td;
    const uint32_t* queue2;
    const uint32_t* queue2_count;
    const uint32_t* nopatch_list;    // pixels whose patch failed its check (horizon, sun's disc: a few hundred of a 1080p frame)
    const uint32_t* nopatch_count;
    float cam_llc[3], cam_h[3], cam_v[3];   // camera frame (lower_left_corner, horizontal, vertical), for the patch corners
    // NEVER-TRACED pixels (written by sky_patch_kernel next to the patches): a pixel whose whole jitter footprint lies outside the
    // screen-space bounds of the volumes' root box (cull_rect, in pixels, already grown by the margin), whose rays all pass the
    // reference sphere at more than its radius INFLATED by what the binary32 discriminant of sphere::intersect can lose
    // (B^2 - 4AC carries up to ~60 eps D^2 of error at distance D: false hits out to sqrt(r^2 + 15 eps D^2), several pixels at 1080p --
    // vpt_cull.h inflates the radius to r^2 + 64 eps D^2), away from
    // the line on which that function's `B == 0` case reports a hit at distance 0 whatever the sphere's place (geometry.h:118-121:
    // cull_line, a x + b y + c in pixels), and that has a patch: raygen skips it altogether (no ray, no head), the tail takes every
    // sample's value from the patch.  cull_enabled = 0: a box corner at or behind the camera plane, the origin on a slab plane, ...
    int cull_enabled;
    float cull_rect[4];              // root box: x0, y0, x1, y1
    // ... refined per 8x8-pixel TILE inside that rectangle (round 4): a tile no NON-EMPTY octree leaf's grown screen bounds touch holds only rays
    // that cross empty nodes -- sample() pushes them out of the root without a draw or a look-up (:1606-1616) and, with nothing behind, they end
    // exactly as rays that miss the box (raygen_kernel walks those pushes: 27 % of config 2's box hits).  cull_tiles[ty * cull_tiles_w + tx] != 0:
    // some leaf may be met.  NULL: no refinement (a leaf corner at or behind the camera plane).
    const unsigned char* cull_tiles;
    uint32_t cull_tiles_w;
    float cull_line[3];
    float cull_sph[4];               // reference sphere: centre, radius
    int render;                      // kernel_params.render (a pixel without heads has to know whether its samples are rendered)
    const unsigned char* never_traced;      // [n_pixels] or NULL (read by the tail); unsigned char* for sky_patch_kernel to write
    float* accum;          // float3[n_pixels]
    float* cost;           // float3[n_pixels] or NULL
    float* depth;          // float[n_pixels] or NULL
    uint32_t* display;     // or NULL
    float* raw;            // float4[n_pixels] or NULL
    float exposure_scale;
    // camera bits for viz_dof
    float lens_radius, focus_dist;
    int viz_dof;
    // environment
    unsigned int environment_type;
    int integrator;
    float sky_mult;
    float sky_color[3];
    float sun_dir[3];
    DTexture env_tex;
    // atmosphere
    int has_atmosphere;
    // camera-point scattering table (vpt_sky.h, CamTable): the two 4-D scattering tables pre-interpolated at
    // the r and mu_s of ONE view point -> [nu slice 8][mu row 128] x {scattering.xyz, single_mie.xyz}
    int cam_tab_valid;
    const float4* cam_tab;            // [8][128][2] float4
    float cam_tab_pos[3];             // the view point (relative to the scene, as env_pos) it was built for
    // view-point ground table (vpt_sky.h, GroundNode): the radiance of a ray from that view point that ends on the ground, as
    // {A.xyz, B.xyz} over [distance to the ground DT_NX][nu DT_NN]; NULL: evaluate every ground hit in full.  dir_tab_err:
    // device word, float bits of the largest relative mid-cell interpolation error found when the table was built
    const float4* dir_tab;
    const uint32_t* dir_tab_err;      // u64 words [SKY_DIR_ERR_WORDS] seen as u32: [0] cell of the largest interpolation error, [1] its float bits,
                                      // [2..7] the centre variant against real rays through the full path (max, rays, rays off by > 1e-3), [8] verdict,
                                      // [10] variants in use; from u64 word 8: per variant {max | cell, rays, rays off by > 1e-3, -}
    float dir_tab_tol;
    // Both tables exist in 2k+1 VARIANTS, one per binary32 value of the view point's radius r within k steps of the camera's:
    // with an open lens every sample starts on the lens disc, whose height above the ground spans a few binary32 steps of r
    // (0.5 m at earth-radius magnitude) while mu_s moves by 1e-7 -- a sample whose (r, mu_s) matches a variant looks from "the"
    // view point as far as the table coordinates can tell.  k = 0 with a closed lens.  Written by sky_view_kernel.
    const SkyView* sky_view;
    float atm_f[40];       // packed vpt_atmosphere_parameters scalars (see vpt_sky.h)
    DTexture transmittance_tex, scattering_tex, irradiance_tex, single_mie_tex;
};

}  // namespace vpt
