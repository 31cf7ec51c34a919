/* vpt_oracle.h -- TEST INFRASTRUCTURE: C API of the CPU oracle (liborc.so).
 *
 * The oracle is a single-threaded-by-default CPU restatement of the reference's
 * `volume_rt_kernel` (source/render_kernel.cu:2216) and everything it calls.  It is
 * the checker for the HIP path and the "CPU ray-marching baseline" of bench.py.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * PARITY PINNED against the reference's own code: oracle/_ref/libvptref.so is the reference's
 * source/render_kernel.cu (+ source/bvh/octree.cpp) compiled unmodified for the CPU where it lies
 * (recipe: `make -C oracle ref`, stand-in CUDA headers in oracle/ref_shim/), and
 * tests/test_oracle_vs_ref.py requires orc_render's buffers to equal that library's BIT FOR BIT on
 * 23 scenes covering both integrators, point light / sun / sky / HDRI, emission, colour grids, 16
 * instances, thin lens + viz_dof, the sphere bounce, volume_depth 3, max_interactions and render=false -- live where the library
 * exists, and against tests/golden/ref_golden.npz (written from it) everywhere else.  Inside that
 * library only the texture fetch and the Philox block function are this oracle's; they are pinned on
 * their own (tests/test_oracle_pins.py): Philox4x32-10 against the Random123 known-answer vectors
 * and cuRAND's stream semantics, the sampler states on exact cases (linear fields, point / wrap /
 * unnormalised addressing), plus the dragon.vdb asset facts, the octree against brute-force point
 * location, the density look-up against a numpy trilinear reference and the fixed-sequence
 * log/sin/cos against libm.  (The reference ships no tests or golden images of its own)
 * Not covered by the pin: the atmosphere tables' CONTENT (GPU precompute, checked separately) and
 * the arithmetic of nvcc's default FMA contraction / approximate intrinsics -- both sides are built
 * with -ffp-contract=off and correctly rounded division/sqrt, see DESIGN.md "arithmetic contract".
 *
 * It re-uses the POD structs of include/vpt_abi.h (they restate the reference's PODs).
 * Texture handles inside those PODs are oracle handles made by orc_texture_create;
 * buffer pointers inside vpt_kernel_params are HOST pointers here.
 */
#ifndef VPT_ORACLE_H_
#define VPT_ORACLE_H_

#include "../include/vpt_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* look-up counters feeding SURVEY 8(d)'s algorithmic-bytes formula */
typedef struct orc_stats {
    unsigned long long samples;
    unsigned long long density_lookups;   /* N_d */
    unsigned long long color_lookups;     /* N_c */
    unsigned long long emission_lookups;  /* N_e */
    unsigned long long tracking_steps;
    unsigned long long skip_steps;
    unsigned long long rng_draws;
    unsigned long long max_draws_per_sample;
} orc_stats;

/* host texture with CUDA sampler semantics; data is NOT copied */
vpt_texture_t orc_texture_create(const vpt_texture_desc *desc, const float *data);
void orc_texture_destroy(vpt_texture_t tex);
/* sample a texture directly (unit tests of the sampler restatement) */
void orc_texture_sample(vpt_texture_t tex, float u, float v, float w, float out[4]);
/* DIAGNOSTIC (never on in the parity tests): 8 = the volume-grid look-ups quantise their linear-filter weights to 1/256, a model of the CUDA texture
 * unit's 1.8 fixed-point weights; 32 = binary32 weights (default, the parity contract).  Returns the previous setting. */
int orc_set_volume_tex_weights(int bits);

/* Philox4x32-10 block function + cuRAND stream semantics (unit tests) */
void orc_philox4x32_10(const unsigned int ctr[4], const unsigned int key[2], unsigned int out[4]);
/* n draws of curand_uniform from curand_init(seed, 0, offset) */
void orc_curand_uniform_stream(unsigned long long seed, unsigned long long offset, int n, float *out);

/* deterministic elementary functions (unit tests) */
float orc_det_logf(float x);
float orc_det_sinf(float x);
float orc_det_cosf(float x);

/* octree facts (unit tests / cross-check of the product's host builder) */
typedef struct orc_octree_info {
    vpt_float3 root_pmin, root_pmax;
    float max_extinction, min_extinction;
    int nonempty[3];          /* nodes with num_volumes>0 at levels 1..3 */
    int total_nodes;
} orc_octree_info;
int orc_octree_info_get(const vpt_gpu_vdb *volumes, int num_volumes, orc_octree_info *out);
/* point location: returns leaf path i*64+x*8+y (or -1) and whether the leaf is empty */
int orc_octree_locate(const vpt_gpu_vdb *volumes, int num_volumes, vpt_float3 p, int *num_volumes_in_leaf);

/* sum_density at a world position over all volumes (brute force, no octree) */
float orc_density_at(const vpt_gpu_vdb *volumes, int num_volumes, vpt_float3 p);

/* Render iterations kp->iteration + k*iter_stride, k=0..iter_count-1, exactly as
 * iter_count successive launches of volume_rt_kernel would (display/raw tonemapped
 * after every iteration like the reference).  nthreads<=1: single thread. */
int orc_render(const vpt_camera *cam, const vpt_light_list *lights,
               const vpt_gpu_vdb *volumes, int num_volumes,
               const vpt_sphere *ref_sphere, const vpt_atmosphere_parameters *atmosphere,
               const vpt_kernel_params *kp, unsigned int iter_count, unsigned int iter_stride,
               int nthreads, orc_stats *stats);

/* the same for the pixels whose index y*W + x is a multiple of pixel_step only (every other pixel's buffers stay untouched):
 * a bounded CPU sample of a big frame over MANY iterations (bench.py: parity of configs 3 and 5 across record chunks) */
int orc_render_subset(const vpt_camera *cam, const vpt_light_list *lights,
                      const vpt_gpu_vdb *volumes, int num_volumes,
                      const vpt_sphere *ref_sphere, const vpt_atmosphere_parameters *atmosphere,
                      const vpt_kernel_params *kp, unsigned int iter_count, unsigned int iter_stride,
                      int nthreads, unsigned int pixel_step, orc_stats *stats);

/* one pixel-sample, returning the integrator's value before accumulation (debugging
 * and per-sample parity tests): out = {L.x, L.y, L.z, tr, depth} */
int orc_sample_pixel(const vpt_camera *cam, const vpt_light_list *lights,
                     const vpt_gpu_vdb *volumes, int num_volumes,
                     const vpt_sphere *ref_sphere, const vpt_atmosphere_parameters *atmosphere,
                     const vpt_kernel_params *kp, int x, int y, float out[5]);

#ifdef __cplusplus
}
#endif
#endif
