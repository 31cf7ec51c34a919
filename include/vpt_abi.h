/*
 * vpt_abi.h -- C ABI of libvpt_hip.so, the MI355X (gfx950) replacement for the one
 * device entry point of sergeneren/Volumetric-Path-Tracer: `volume_rt_kernel`
 * (reference source/render_kernel.cu:2216) together with the scene handles that
 * kernel dereferences (dense VDB textures, the instance octree, the look-up
 * textures).  Plain C: POD structs, raw pointers, sizes.  No C++ / torch types.
 *
 * Every struct below mirrors a reference POD field-for-field (same names, same
 * units).  Two deliberate differences, both forced by the platform:
 *   - `cudaTextureObject_t` handles become `vpt_texture_t` handles created with
 *     vpt_texture_create() (CDNA4 has no texture-filter path; filtering is ALU);
 *   - reference classes with a vptr (`point_light`, `sphere`) are restated as PODs
 *     without one -- the kernel never virtual-calls through them.
 *
 * Threading: one vpt_ctx per GPU, re-entrant per ctx, all work is enqueued on the
 * HIP stream the caller passes (NULL = the ctx's own stream).  Errors: every entry
 * point returns 0 or a negative VPT_E_* code and never exits the process
 * (reference: check_success() -> exit(EXIT_FAILURE), source/main.cpp:136-142).
 */
#ifndef VPT_ABI_H_
#define VPT_ABI_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VPT_ABI_VERSION 2

/* ---- error codes ------------------------------------------------------------- */
#define VPT_OK              0
#define VPT_E_INVALID      -1   /* bad argument / NULL pointer / inconsistent sizes    */
#define VPT_E_NO_DEVICE    -2   /* no usable gfx950 device                             */
#define VPT_E_HIP          -3   /* a HIP runtime call failed (see vpt_last_error)       */
#define VPT_E_NOMEM        -4
#define VPT_E_NOT_READY    -5   /* scene / octree / look-up tables not bound yet        */
#define VPT_E_UNSUPPORTED  -6
#define VPT_E_IO           -7   /* host-side loader could not read / parse a file       */

/* ---- small vector PODs (layout = CUDA float3/float4/int3/uint2) ------------------ */
typedef struct vpt_float2 { float x, y; } vpt_float2;
typedef struct vpt_float3 { float x, y, z; } vpt_float3;
typedef struct vpt_float4 { float x, y, z, w; } vpt_float4;
typedef struct vpt_int3   { int x, y, z; } vpt_int3;
typedef struct vpt_uint2  { unsigned int x, y; } vpt_uint2;

/* replaces cudaTextureObject_t (an opaque 64-bit handle in the reference too) */
typedef unsigned long long vpt_texture_t;

/* ---- camera : reference source/gpu_vdb/camera.h:94-148 --------------------------- */
typedef struct vpt_camera {
    float      time1, time0;          /* camera.h:139 (note the order: time1 first)   */
    vpt_float3 origin;
    float      focus_dist;
    vpt_float3 lower_left_corner;
    vpt_float3 horizontal;
    vpt_float3 vertical;
    vpt_float3 u, v, w;
    float      lens_radius;
    unsigned char viz_dof;            /* bool                                         */
} vpt_camera;

/* ---- lights : reference source/light.h:70-121,156-169 ---------------------------- */
typedef struct vpt_point_light {
    vpt_float3 pos;
    vpt_float3 dir;
    float      power;
    vpt_float3 color;
} vpt_point_light;

typedef struct vpt_light_list {
    unsigned int           num_lights;
    const vpt_point_light *light_ptr;  /* HOST pointer; copied by vpt_render (the
                                          reference used cudaMallocManaged,
                                          source/main.cpp:1000)                       */
} vpt_light_list;

/* ---- reference sphere : source/geometry/geometry.h:80-171 ------------------------ */
typedef struct vpt_sphere {
    vpt_float3 center;
    float      radius;
    vpt_float3 color;
    float      roughness;
} vpt_sphere;

/* ---- VDB_INFO / GPU_VDB : reference source/gpu_vdb/gpu_vdb.h:59-76,148-152 -------- */
typedef struct vpt_vdb_info {
    float      voxelsize;
    vpt_int3   dim;
    vpt_float3 bmin;
    vpt_float3 bmax;
    float      max_density;
    float      min_density;
    unsigned char has_color;
    unsigned char has_emission;
    unsigned char matte;
    vpt_texture_t density_texture;    /* 3-D f32, normalised coords, linear, clamp    */
    vpt_texture_t emission_texture;   /* 3-D f32                                      */
    vpt_texture_t color_texture;      /* 3-D float4                                   */
} vpt_vdb_info;

typedef struct vpt_gpu_vdb {
    vpt_vdb_info vdb_info;
    float        xform[4][4];         /* mat4::m[col][row], matrix_math.h:49-70: the
                                         OpenVDB index->world matrix in row-vector
                                         convention (gpu_vdb.cpp:81-92)               */
} vpt_gpu_vdb;

/* ---- atmosphere : reference source/atmosphere/definitions.h:35-99 ----------------- */
typedef struct vpt_density_profile_layer {
    float width, exp_term, exp_scale, linear_term, const_term;
} vpt_density_profile_layer;

typedef struct vpt_density_profile { vpt_density_profile_layer layers[2]; } vpt_density_profile;

typedef struct vpt_atmosphere_parameters {
    vpt_float3 sky_spectral_radiance_to_luminance;
    vpt_float3 sun_spectral_radiance_to_luminance;
    vpt_float3 solar_irradiance;
    float      angle;
    float      bottom_radius;
    float      top_radius;
    int        use_luminance;
    vpt_density_profile rayleigh_density;
    vpt_float3 rayleigh_scattering;
    vpt_density_profile mie_density;
    vpt_float3 mie_scattering;
    vpt_float3 mie_extinction;
    float      mie_phase_function_g;
    vpt_density_profile absorption_density;
    vpt_float3 absorption_extinction;
    vpt_float3 ground_albedo;
    float      sun_angular_radius;
    float      mu_s_min;
    float      exposure;
    vpt_float3 white_point;
    /* precompute scratch (device pointers; only used by vpt_atmosphere_precompute) */
    vpt_float4 *delta_irradience_buffer;
    vpt_float4 *delta_rayleigh_scattering_buffer;
    vpt_float4 *delta_mie_scattering_buffer;
    vpt_float4 *delta_scattering_density_buffer;
    vpt_float4 *delta_multiple_scattering_buffer;
    vpt_float4 *transmittance_buffer;
    vpt_float4 *irradiance_buffer;
    vpt_float4 *scattering_buffer;
    vpt_float4 *optional_mie_single_scattering_buffer;
    /* render-time look-up tables */
    vpt_texture_t transmittance_texture;          /* 2-D 256x64 float4, linear         */
    vpt_texture_t scattering_texture;             /* 3-D 256x128x32 float4, linear     */
    vpt_texture_t irradiance_texture;             /* 2-D 256x64 float4, linear         */
    vpt_texture_t single_mie_scattering_texture;  /* 3-D 256x128x32 float4, linear     */
} vpt_atmosphere_parameters;

/* LUT dimensions, reference source/atmosphere/constants.h:50-62 */
#define VPT_TRANSMITTANCE_W 256
#define VPT_TRANSMITTANCE_H 64
#define VPT_SCATTERING_R    32
#define VPT_SCATTERING_MU   128
#define VPT_SCATTERING_MU_S 32
#define VPT_SCATTERING_NU   8
#define VPT_IRRADIANCE_W    256
#define VPT_IRRADIANCE_H    64

/* ---- Kernel_params : reference source/kernel_params.h:39-109 ---------------------- */
/* Buffer pointers are DEVICE pointers owned by the caller, exactly as in the
 * reference (source/main.cpp:596-637 allocates them with cudaMalloc). */
typedef struct vpt_kernel_params {
    unsigned char render;             /* bool */
    unsigned char debug;              /* bool */
    vpt_uint2     resolution;
    float         exposure_scale;
    unsigned int *display_buffer;     /* W*H 0xffRRGGBB                                */
    vpt_float4   *raw_buffer;         /* W*H tonemapped rgb + alpha                    */
    vpt_float3   *blue_noise_buffer;  /* 256*256, advanced in place every iteration; values in [0, 1] (the reference's: an 8-bit image / 255).
                                       * A value outside is used CLAMPED to [0, 1] as the pixel jitter (its in-place advance is the reference's fmod). */
    vpt_float3   *emission_texture;   /* 256-entry blackbody LUT                       */
    float         emission_scale;
    float         emission_pivot;
    vpt_float3   *density_color_texture; /* 256-entry LUT                              */
    unsigned int  iteration;
    vpt_float3   *accum_buffer;       /* W*H running mean                              */
    float        *depth_buffer;       /* W*H running mean of first-hit distance        */
    unsigned int  max_interactions;
    int           ray_depth;
    int           volume_depth;
    float         min_extinction;
    float         phase_g1, phase_g2, phase_f;
    vpt_float3    albedo;
    vpt_float3    extinction;
    vpt_float3    transmittance;
    float         tr_depth;
    float         density_mult;
    unsigned int  environment_type;   /* 0 procedural sky, 1 HDRI                      */
    float         azimuth;            /* degrees                                       */
    float         elevation;          /* degrees                                       */
    vpt_float3    sun_color;
    vpt_float3    sky_color;
    float         sun_mult;
    float         sky_mult;
    double        energy_inject;
    vpt_texture_t env_tex;            /* 2-D float4 lat-long, linear, wrap/clamp       */
    int           env_sample_tex_res;
    vpt_texture_t sky_tex;
    vpt_texture_t env_func_tex;           /* 2-D f32, unnormalised, point, wrap/clamp  */
    vpt_texture_t env_cdf_tex;
    vpt_texture_t env_marginal_func_tex;  /* 1-D f32, unnormalised, point, wrap        */
    vpt_texture_t env_marginal_cdf_tex;
    float         env_marginal_int;
    vpt_float3   *debug_buffer;
    vpt_float3   *cost_buffer;        /* W*H running mean (always BLACK, :2249,:2280)  */
    int           integrator;         /* 0 direct_integrator, !=0 vol_integrator       */
} vpt_kernel_params;

/* ---- textures : replaces cudaMalloc(3D)Array + cudaCreateTextureObject ------------
 * call sites: source/gpu_vdb/gpu_vdb.cpp:214-248,293-327,373-407,
 * source/main.cpp:775-867,957-976, source/atmosphere/atmosphere.cpp:503-675        */
#define VPT_ADDR_WRAP   0
#define VPT_ADDR_CLAMP  1
#define VPT_FILTER_POINT  0
#define VPT_FILTER_LINEAR 1

typedef struct vpt_texture_desc {
    int width, height, depth;     /* height = depth = 1 for 1-D, depth = 1 for 2-D     */
    int channels;                 /* 1 (f32) or 4 (float4)                             */
    int normalized_coords;        /* cudaTextureDesc::normalizedCoords                 */
    int filter_mode;              /* VPT_FILTER_*                                      */
    int address_mode[3];          /* VPT_ADDR_* per axis                               */
} vpt_texture_desc;

typedef struct vpt_ctx vpt_ctx;

/* ---- context ----------------------------------------------------------------------- */
/* replaces init_cuda()/cuModuleLoadData/cuModuleGetFunction, source/main.cpp:343-355,
 * 1221-1244.  `device` is a HIP device ordinal. */
int  vpt_create(int device, vpt_ctx **out_ctx);
void vpt_destroy(vpt_ctx *ctx);
const char *vpt_last_error(const vpt_ctx *ctx);   /* ctx may be NULL: last global error */
int  vpt_abi_version(void);
/* returns the HIP stream (hipStream_t) owned by the ctx */
void *vpt_stream(vpt_ctx *ctx);
int  vpt_sync(vpt_ctx *ctx);                      /* cudaDeviceSynchronize, main.cpp:1829 */

/* ---- textures ---------------------------------------------------------------------- */
/* `data` is a HOST pointer to width*height*depth*channels floats, x fastest
 * (LayoutXYZ, gpu_vdb.cpp:179-212); it is copied to HBM. */
int  vpt_texture_create(vpt_ctx *ctx, const vpt_texture_desc *desc, const float *data,
                        vpt_texture_t *out_tex);
/* same, but `device_data` already lives in HBM: no upload, the caller keeps ownership and must keep it alive.
 * SNAPSHOT SEMANTICS for volume grids: vpt_scene_set_volumes re-lays every density / emission grid of 8 MiB or more into a
 * float4 corner-quad copy owned by the context (4x the grid's bytes; DESIGN.md "Data layout in HBM") and renders from THAT copy.
 * A host that rewrites such a grid in place (animated volumes, a tensor it keeps updating) calls vpt_scene_set_volumes again
 * before the next render -- it rebuilds the copies -- or creates the context with VPT_GRID_LAYOUT=dense in the environment,
 * which keeps rendering from the caller's memory at ~2x the look-up traffic.  Grids below 8 MiB, colour grids and the
 * sky / environment tables are always read in place. */
int  vpt_texture_create_device(vpt_ctx *ctx, const vpt_texture_desc *desc,
                               const float *device_data, vpt_texture_t *out_tex);
int  vpt_texture_destroy(vpt_ctx *ctx, vpt_texture_t tex);
/* The renderer caches two per-frame sky tables (camera-point scattering table, view-point ground table) keyed on the camera
 * origin, the sun direction, the model scalars and the device ADDRESSES of the four atmosphere tables.  Creating or destroying
 * a texture and vpt_atmosphere_precompute drop that cache themselves; a host that rewrites the CONTENTS of an adopted device
 * table in place (vpt_texture_create_device) calls this before its next render. */
int  vpt_invalidate_sky_tables(vpt_ctx *ctx);
/* FRAME-AHEAD.  The literal drop-in loop (one vpt_render + device sync per iteration, main.cpp:1822-1829) is served from rays traced ahead: from
 * the second identical one-iteration call on, a call traces the rays of the next 2, 4, 8, 16 iterations in ONE launch and the following calls run
 * only their tail (every buffer after every frame bit-identical to frame by frame).  "Identical" is decided on the BYTES of the argument structs
 * (camera, sphere, atmosphere, kernel params, lights, stream); scene / texture / sky-table changes made through this API void what was traced.
 * What the key cannot see is device memory rewritten IN PLACE behind unchanged pointers between two calls: the emission / density-colour
 * look-up tables, the blue-noise buffer (the ahead batch's jitter comes from a copy taken when it was traced), a grid or table adopted with
 * vpt_texture_create_device.  A host that does that calls vpt_frame_ahead_invalidate before its next vpt_render (the rays are traced again from
 * the current contents), or switches the mechanism off for the context with vpt_set_frame_ahead(ctx, 0) (also: VPT_NO_FRAME_AHEAD in the
 * environment when the context is created).  vpt_render_batch calls of more than one iteration never use it. */
int  vpt_frame_ahead_invalidate(vpt_ctx *ctx);
int  vpt_set_frame_ahead(vpt_ctx *ctx, int enable);

/* ---- scene: instances + octree ---------------------------------------------------------
 * replaces: cuMemAlloc+HtoD of instances[] (source/main.cpp:1301-1303) and
 * BVH_Builder::build_bvh -> build_octree<<<1,1>>> (source/bvh/bvh_builder.cpp:46-105,
 * source/bvh/bvh_kernels.cu:204-246,455,582).  Builds the fixed 3-level octree over
 * the instance bounds on the host and uploads it with the instance table.  The LBVH
 * is not built: volume_rt_kernel never traverses it (render_kernel.cu:2222). */
int  vpt_scene_set_volumes(vpt_ctx *ctx, const vpt_gpu_vdb *volumes, int num_volumes);
/* read back the root node facts the reference keeps on the host */
int  vpt_scene_get_root(vpt_ctx *ctx, vpt_float3 *pmin, vpt_float3 *pmax,
                        float *max_extinction, float *min_extinction);
/* number of octree nodes with num_volumes>0 per level (root excluded): out[3] */
int  vpt_scene_get_octree_stats(vpt_ctx *ctx, int out_nonempty[3]);

/* ---- the hot path ------------------------------------------------------------------------
 * vpt_render: ONE launch of volume_rt_kernel = one sample per pixel at
 * kernel_params->iteration (drop-in for cuLaunchKernel at source/main.cpp:1822-1829,
 * params[] = {cam, lights, volumes, sphere, geo_list, bvh, octree, atmosphere, kp};
 * volumes/octree come from vpt_scene_set_volumes, geo_list/bvh are unused by the
 * kernel).  Asynchronous on `stream` (hipStream_t, NULL = ctx stream). */
int  vpt_render(vpt_ctx *ctx, const vpt_camera *cam, const vpt_light_list *lights,
                const vpt_sphere *ref_sphere, const vpt_atmosphere_parameters *atmosphere,
                const vpt_kernel_params *kernel_params, void *stream);

/* vpt_render_batch: iterations kp->iteration + k*iter_stride, k = 0..iter_count-1, in
 * one call (what the reference's main loop does with iter_count launches).  Buffers
 * end in exactly the state iter_count successive launches would leave them in,
 * except display/raw which are tonemapped once at the end (they are overwritten
 * every launch in the reference, render_kernel.cu:2303-2316).  iter_stride > 1 is
 * the multi-GPU iteration striping: the running means then weight the samples of
 * this rank only (local index k), and blue noise advances iter_stride steps per k. */
int  vpt_render_batch(vpt_ctx *ctx, const vpt_camera *cam, const vpt_light_list *lights,
                      const vpt_sphere *ref_sphere, const vpt_atmosphere_parameters *atmosphere,
                      const vpt_kernel_params *kernel_params, unsigned int iter_count,
                      unsigned int iter_stride, void *stream);

/* advance a 256x256 float3 blue-noise buffer by `steps` golden-ratio increments
 * (render_kernel.cu:2320-2325), e.g. to position rank g of a striped render.  As in the
 * reference, a launch only advances the entries its pixels own: the first
 * min(num_pixels, 65536) (`if (idx < 256*256)` with idx < W*H); pass num_pixels = W*H. */
int  vpt_blue_noise_advance(vpt_ctx *ctx, vpt_float3 *blue_noise_buffer, unsigned int steps,
                            unsigned int num_pixels, void *stream);

/* per-launch statistics of the last render call (device counters, read back after a
 * sync): look-up counts feeding the algorithmic-bytes roofline */
typedef struct vpt_render_stats {
    unsigned long long samples;           /* pixel-samples traced                      */
    unsigned long long density_lookups;   /* N_d                                       */
    unsigned long long color_lookups;     /* N_c                                       */
    unsigned long long emission_lookups;  /* N_e                                       */
    unsigned long long tracking_steps;    /* RNG-consuming steps of sample/Tr/emission */
    unsigned long long skip_steps;        /* empty-node pushes                         */
    unsigned long long queued_rays;       /* rays the last batch handed to the tracer  */
    float              trace_ms;          /* HIP-event time of trace_kernel            */
    float              raygen_ms;         /* HIP-event time of raygen_kernel           */
    float              tail_ms;           /* HIP-event time of tail_resolve_kernel (environment tail + resolve, fused) */
    /* trilinear fetches the tracer actually issued (counting renders): the look-up point lies inside the instance's
     * domain and the value is used -- density / colour (float4 texels) / emission.  The N_* above are the reference-defined
     * counts (every instance of the leaf at every step, :1003-1014, and the colour at every step, :1662). */
    unsigned long long density_fetches;
    unsigned long long color_fetches;
    unsigned long long emission_fetches;
    /* of density_fetches: those a zero-footprint mask answered (all eight texels exactly 0: the value is +0 without the loads) */
    unsigned long long density_zero_skips;
} vpt_render_stats;
/* enable/disable look-up counting (off by default: counting costs atomics) */
int  vpt_set_counting(vpt_ctx *ctx, int enable);
int  vpt_get_stats(vpt_ctx *ctx, vpt_render_stats *out);

/* ---- multi-GPU (the reference is single-GPU, source/main.cpp:353) --------------------------
 * One process (or thread) per GPU, one context each.  Rank r of G renders iterations r, r+G, ... of every pixel
 * (vpt_render_batch with iter_stride = G after vpt_blue_noise_advance(r)); its accum buffer then holds the running
 * mean of ITS n_local iterations.  vpt_allreduce_accum turns every rank's buffer into the job's mean,
 *     accum <- sum_r n_r * accum_r / sum_r n_r,
 * with ONE RCCL all-reduce over xGMI of W*H*3 + 1 floats (the rank's weighted image with its iteration count in the last float of a
 * context-owned payload buffer), enqueued on `stream` (NULL = the context's stream) between a scale and a divide kernel -- no host synchronisation; a render
 * issued afterwards on the same stream is ordered behind it.  RCCL (librccl.so) is loaded on the first vpt_comm_* call.
 * The reduce is TERMINAL for a progressive render: afterwards `accum` holds the JOB's mean, not this rank's -- a further
 * vpt_render_batch into it would treat the job's mean as the rank's running mean, and a second reduce would count the other
 * ranks twice.  To go on rendering, keep the rank-local mean in its own buffer and reduce a copy.  n_local_iterations x ranks
 * must not exceed 2^24 (the count rides along as one binary32); a job in which every rank passes 0 leaves the buffers as they are.
 *   vpt_comm_unique_id: ncclGetUniqueId on ONE rank; the 128 bytes reach the other ranks by the host's own means
 *                       (MPI, a file, torch.distributed's store: see INTEGRATION.md).
 *   vpt_comm_init_rank: ncclCommInitRank for this context's device (collective: every rank calls it). */
#define VPT_COMM_ID_BYTES 128
int  vpt_comm_unique_id(unsigned char *out_id /* [VPT_COMM_ID_BYTES] */);
int  vpt_comm_init_rank(vpt_ctx *ctx, int nranks, int rank, const unsigned char *id /* [VPT_COMM_ID_BYTES] */);
int  vpt_comm_destroy(vpt_ctx *ctx);
int  vpt_allreduce_accum(vpt_ctx *ctx, float *accum_device, unsigned long long n_floats, unsigned int n_local_iterations, void *stream);
/* display_buffer / raw_buffer.xyz of kp from kp->accum_buffer as it is NOW (render_kernel.cu:2292-2316: ACES fit, gamma, 8-bit
 * pack): a render tonemaps this rank's running mean, so after vpt_allreduce_accum the display image is refreshed with this. */
int  vpt_resolve_display(vpt_ctx *ctx, const vpt_kernel_params *kp, void *stream);

/* ---- atmosphere (prerequisite of the procedural sky) -----------------------------
 * vpt_atmosphere_default_model: the scalars atmosphere::atmosphere() + init() + update_model() leave in
 * atmosphere_parameters with the reference's defaults (constant solar spectrum, ozone on, white
 * balance on, lambdas 680/550/440 nm, use_luminance NONE) -- source/atmosphere/atmosphere.cpp:698-784,
 * 1177-1230.  Buffers / texture handles are zeroed.
 * vpt_atmosphere_precompute: atmosphere::precompute (atmosphere.cpp:888-1116; kernels
 * atmosphere_kernels.cu:621-752) followed by copy_*_texture (:503-675): allocates any of the nine
 * scratch buffers that are NULL, fills them, and creates the four look-up textures.
 * vpt_atmosphere_read_lut: device->host copy of one table (0 transmittance 256x64, 1 irradiance
 * 256x64, 2 scattering 256x128x32, 3 single Mie 256x128x32; float4 texels). */
int  vpt_atmosphere_default_model(vpt_atmosphere_parameters *atm);
/* vpt_atmosphere_model: the same scalars for ANY setting of the reference's model switches -- what atmosphere::init
 * (spectra, atmosphere.cpp:1193-1224), precompute's luminance factors (:903-910) and update_model(lambdas) (:698-784) leave in
 * atmosphere_parameters: solar spectrum constant / ASTM, ozone on / off, white balance, luminance mode NONE (0) or
 * APPROXIMATE (1) (PRECOMPUTED (2) yields the scalars of its final pass here; its 15-wavelength table passes are
 * vpt_atmosphere_precompute_model's), the three wavelengths the
 * tables are computed for, exposure, the 102-degree sun-zenith limit of half-precision tables, the length unit.  The published
 * data tables behind it (solar irradiance, ozone cross-sections, CIE 1931 CMFs) are read from `spectra_file`
 * (NULL: data/atmosphere_spectra.bin next to the library).  Follow with vpt_atmosphere_precompute. */
typedef struct vpt_atmosphere_model_options {
    int    use_constant_solar_spectrum;   /* 1 */
    int    use_ozone;                     /* 1 */
    int    do_white_balance;              /* 1 */
    int    use_luminance;                 /* 0 NONE, 1 APPROXIMATE, 2 PRECOMPUTED (tables through vpt_atmosphere_precompute_model) */
    int    half_precision;                /* 0 */
    float  exposure;                      /* 1 */
    double lambdas[3];                    /* 680, 550, 440 nm */
    double length_unit_in_meters;         /* 1 */
} vpt_atmosphere_model_options;
void vpt_atmosphere_model_options_default(vpt_atmosphere_model_options *opt);
int  vpt_atmosphere_model(const vpt_atmosphere_model_options *opt, const char *spectra_file, vpt_atmosphere_parameters *atm);
int  vpt_atmosphere_precompute(vpt_ctx *ctx, vpt_atmosphere_parameters *atm, int num_scattering_orders, void *stream);
/* atmosphere::init's precomputation for any luminance mode (source/atmosphere/atmosphere.cpp:1227-1275): vpt_atmosphere_model +
 * the table passes.  use_luminance 0 / 1: one pass, i.e. vpt_atmosphere_model followed by vpt_atmosphere_precompute.
 * use_luminance 2 (PRECOMPUTED): five passes over 15 wavelengths with per-pass model scalars, luminance-from-radiance matrices and
 * blending as the reference runs them (its argument-passing quirks included, csrc/vpt_atmosphere.hip), then the transmittance
 * table for opt->lambdas (the reference's init() has no wavelength input: its final pass uses kDefaultLambdas = 680 / 550 / 440 nm, the
 * defaults of opt->lambdas; other wavelengths are this entry point's extension).  `atm`: the scalars are an OUTPUT; the nine device buffers
 * and the four texture handles are IN/OUT -- ZERO the struct before the first call (NULL buffers are allocated), a later call with the same
 * struct (another sun model, other options) refills the buffers and re-creates the handles it already holds. */
int  vpt_atmosphere_precompute_model(vpt_ctx *ctx, const vpt_atmosphere_model_options *opt, const char *spectra_file,
                                     vpt_atmosphere_parameters *atm, int num_scattering_orders, void *stream);
int  vpt_atmosphere_read_lut(vpt_ctx *ctx, const vpt_atmosphere_parameters *atm, int which, float *host_out, size_t n_floats);

/* ---- environment importance tables (prerequisite of estimate_sky on the procedural sky) -------
 * vpt_env_cdf_build: create_cdf's table fill (source/main.cpp:647-757) over the host
 * single-scattering sky `sample_atmosphere` (main.cpp:242-312) for kp->azimuth/elevation/sky_color:
 * val4 float4[res*res] (may be NULL), func/cdf float[res*res], marginal_func/marginal_cdf
 * float[res]; the reference uses res = 180.  Host only, no GPU needed.
 * vpt_env_cdf_create: the same plus the five texture objects of main.cpp:759-867; fills
 * kp->sky_tex, env_func_tex, env_cdf_tex, env_marginal_func_tex, env_marginal_cdf_tex,
 * env_sample_tex_res and env_marginal_int. */
int  vpt_env_cdf_build(const vpt_kernel_params *kp, int res, float *val4, float *func, float *cdf,
                       float *marginal_func, float *marginal_cdf, float *marginal_int);
int  vpt_env_cdf_create(vpt_ctx *ctx, vpt_kernel_params *kp);

/* ---- host-side helpers restating reference host code the path depends on --------------- */
/* camera::update_camera, source/gpu_vdb/camera.h:110-129 */
void vpt_camera_update(vpt_camera *cam, vpt_float3 lookfrom, vpt_float3 lookat, vpt_float3 vup,
                       float vfov, float aspect, float aperture);
/* the "F key" framing of source/main.cpp:526-543: bbox (seeded with the origin) of the transformed
 * bmin / bmax corners of every instance, lookat = centre, lookfrom = centre + |diagonal| (1,1,1),
 * vup = (0,1,0); then update_camera.  out_center / out_dist may be NULL. */
void vpt_camera_frame(vpt_camera *cam, const vpt_gpu_vdb *volumes, int num_volumes, float vfov, float aspect,
                      float aperture, vpt_float3 *out_center, float *out_dist);
/* camera() default ctor, camera.h:97-106 */
void vpt_camera_default(vpt_camera *cam);
/* GPU_VDB::Bounds, source/gpu_vdb/gpu_vdb.h:131-146 */
void vpt_gpu_vdb_bounds(const vpt_gpu_vdb *vdb, vpt_float3 *pmin, vpt_float3 *pmax);
/* the per-instance transform of the .ins loader, source/main.cpp:1060-1095:
 *   xform = base; xform.translate(-xform.extract_translate()); xform.scale(scale);
 *   xform = quaternion_to_mat4(rotation) * xform; xform.translate(position)
 * with mat4's own conventions (matrix_math.h:49-70,130-163,326-344,379-412: storage m[col][row],
 * scale() touches the diagonal only, operator* as written there).  base/out: float[4][4] as in
 * vpt_gpu_vdb::xform; position double[3]; rotation double[4] (x, y, z, w). */
void vpt_instance_xform(const float base[4][4], const double position[3], const double rotation[4],
                        double scale, float out[4][4]);
/* Kernel_params defaults of source/main.cpp:1350-1376 (+ the per-frame overrides at
 * :1533-1546: azimuth 120, elevation 30, energy_inject 1.0) */
void vpt_kernel_params_default(vpt_kernel_params *kp);

#ifdef __cplusplus
}
#endif
#endif /* VPT_ABI_H_ */
