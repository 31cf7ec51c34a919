/* vpt_testhooks.h -- NOT part of the drop-in boundary.  Small probes the test-suite uses to
 * pin the product's fixed-sequence arithmetic (csrc/vpt_math.h) against the oracle's,
 * on the host and on the device.  Exported by libvpt_hip.so next to the vpt_abi.h symbols. */
#ifndef VPT_TESTHOOKS_H_
#define VPT_TESTHOOKS_H_
#include "vpt_abi.h"
#ifdef __cplusplus
extern "C" {
#endif
#define VPT_OP_LOG 0
#define VPT_OP_SIN 1
#define VPT_OP_COS 2
#define VPT_OP_UNIFORM 3   /* in[i] reinterpreted as uint32 -> curand_uniform mapping */
#define VPT_OP_RCP 4       /* 1.0f / x  */
#define VPT_OP_SQRT 5
/* evaluate op element-wise with the HOST build of csrc/vpt_math.h */
int vpt_test_host_math(int op, const float *in, float *out, int n);
/* evaluate op element-wise in a gfx950 kernel (in/out are HOST arrays) */
int vpt_test_device_math(vpt_ctx *ctx, int op, const float *in, float *out, int n);
/* n raw Philox draws of rocRAND's philox4x32_10 from rocrand_init(seed, 0, offset), mapped
 * with the curand_uniform formula -- the stream the trace kernel consumes */
int vpt_test_device_uniform_stream(vpt_ctx *ctx, unsigned long long seed, unsigned long long offset, int n, float *out);
/* n draws of the product's own Philox stream (csrc/vpt_rng.h) for key = seed, offset, drawn
 * through the trace kernel's refill-point protocol */
int vpt_test_device_product_stream(vpt_ctx *ctx, unsigned int key, unsigned int offset, int n, float *out);
/* schedule histogram of the last counted render (vpt_set_counting(ctx, 1)): wave-level sums of
 * [0] tracer loop passes, [1] walking lanes, [2] lanes parked in transition states, [3] idle lanes,
 * [4] passes that ran transitions, [5] inner transition passes, [6] lanes in them, [7] lanes that
 * executed the tracking step proper; [8..11] wave-level shader-clock cycles spent in refill, Philox
 * top-up, the walk step and the transition states (direct tracer only) */
int vpt_test_get_schedule(vpt_ctx *ctx, unsigned long long out[12]);
/* vol_integrator's runs of empty sample() calls in the last counted render (wave-level sums of lane-passes through the tracking step of
 * its delta-tracking walks): [0] lane-passes that reached a density look-up, [1] lane-passes that ended in retry spins only, [2] retry
 * draws in all, [3] lane-passes whose walk ended at t >= distance with no retry left */
int vpt_test_get_retry_stats(vpt_ctx *ctx, unsigned long long out[4]);
/* spatial coherence of the density look-ups of the last counted render of a single-volume scene (a 1-in-16 sample of
 * the wave-level gather events): [0] events, [1] lanes taking part, [2] distinct 8x8x8-voxel bricks among them,
 * [3] distinct 4x4x4 bricks, [4] 128-byte lines per lane summed over lanes, [5] distinct 128-byte lines per event */
int vpt_test_get_coherence(vpt_ctx *ctx, unsigned long long out[8]);
/* view-point ground table of the environment tail (vpt_sky.h): *built = 1 when the last render built or reused one, *err = the
 * largest relative deviation of its bilinear interpolant from the full evaluation at the cell centres (the table is used while
 * err <= the tolerance, 5e-4 unless VPT_DIR_TABLE_TOL says otherwise), *cell = where: distance index * (DT_NN - 1) + nu index */
int vpt_test_get_dir_table_error(vpt_ctx *ctx, int *built, float *err, unsigned int *cell);
/* the whole build-time check of that table: out[0] = the interpolation error above; out[1] = the largest relative difference
 * between the table path and the FULL path (the reference's arithmetic) of sample_atmosphere along real view rays from the
 * camera origin, one per reachable cell centre; out[2] = rays compared; out[3] = rays that differ by more than 1e-3;
 * out[4] = 1 when the tail uses the tables (out[0] <= tolerance and at least one variant passed: worst ray <= 2e-2, at most 0.5 % of
 * its rays unflipped and off by more than 1e-3, the flipped ones costing at most 3e-4 on average -- a variant that fails loses its table); out[5] = variants in use (1 behind a closed lens, up to
 * 2 k + 1 behind an open one); out[6], out[7] = the worst ray and the largest share of rays above 1e-3 over ALL checked variants */
int vpt_test_get_dir_table_check(vpt_ctx *ctx, float out[8]);
/* round 5: the check above counts in out[3] / out[7] only UNFLIPPED rays (gate: at most 0.5 % of a variant's rays); a ray is FLIPPED when the full
 * path finds its binary32 ground point one step (0.5 m) above the ground -- the reference's own ray-to-ray noise, which a smooth table cannot and
 * should not follow.  out[0] = share of flipped rays of the centre variant, out[1] = the largest share over all checked variants; out[2], out[3] = what
 * they cost a smooth cache on average -- the flipped rays' summed deviation over ALL rays -- for the centre variant / the largest over all (gate: 3e-4) */
int vpt_test_get_dir_table_flips(vpt_ctx *ctx, float out[4]);
/* sample_atmosphere (render_kernel.cu:839-895) as the environment tail of the last render evaluates it, along n unit directions
 * dirs[3n] -> out[3n], from origins[3n] (scene coordinates) or, origins == NULL, from that render's view point; use_table: ground
 * hits through the view-point ground tables (when the last render had them within tolerance), else in full */
int vpt_test_sky_samples(vpt_ctx *ctx, int n, const float *origins, const float *dirs, int use_table, float *out);
/* screen-space bounds, in pixels (x0, y0, x1, y1, not grown by any margin), of the world box [lo, hi] as the closed-lens camera::get_ray
 * sees it -- what the never-traced pixel mask of the renderer is built from (csrc/vpt_host.hip: project_box).  Host only.
 * VPT_E_UNSUPPORTED: a corner lies at or behind the camera plane (no bound; the renderer then skips nothing). */
int vpt_test_project_box(const vpt_camera *cam, const float lo[3], const float hi[3], int width, int height, float rect[4]);
/* the sphere half of that mask (csrc/vpt_cull.h: sphere_may_hit): 1 when a primary ray within `diag` (chord) of the unit direction
 * dir_centre could make sphere::intersect report a hit -- its binary32 discriminant included --, 0 when none can.  Host only. */
int vpt_test_sphere_may_hit(const float org[3], const float dir_centre[3], float diag, const float sphere[4]);
/* the host's check of one grid extent d (csrc/vpt_fastdiv.h): 1 when y + (q - d y) r, y = q r, has the bits of q / d for every
 * significand q of a binade (the look-ups then form the quotient that way, csrc/vpt_trace_common.h: to_unit), 0 when some q differs,
 * d is outside [1, 2^16] or the host has no FMA.  r: the reciprocal to test -- RN(1 / d) in the product.  Host only. */
int vpt_test_fast_div_ok(float d, float r);
/* per-pixel sky patches of the last render (csrc/vpt_tail.hip: sky_patch_kernel): pixels of the frame, and how many of them passed
 * the patch's check (the others evaluate every untraced sample in full); both 0 when the render used no patches */
int vpt_test_get_sky_patch_coverage(vpt_ctx *ctx, unsigned long long *pixels, unsigned long long *with_patch);
/* which per-view caches of the environment tail the LAST render used (csrc/vpt_host.hip: built for a batch of >= 2 iterations or a repeated
 * view): out[0] per-pixel sky patches, [1] never-traced pixel mask (raygen skips those pixels), [2] sky dome(s), [3] dome variants (1 behind
 * a closed lens, 2 k + 1 behind an open one), [4] camera-point scattering table, [5] view-point ground table(s) bound, [6] resolved samples
 * (the tracer adds a finished path's environment term from the dome, sky_fix_kernel serves the rest, the tail streams 16 + 8 bytes per
 * sample: csrc/vpt_device.h, ResolveParams::lean); [7] the never-traced mask was refined per 8x8-pixel tile by the non-empty octree leaves' screen
 * bounds (ResolveParams::cull_tiles; not in counting renders, which keep the reference-defined skip counts of the rays it removes) */
int vpt_test_get_cache_state(vpt_ctx *ctx, int out[8]);
/* the 8x8-pixel tiles ((width + 7) / 8 per row, (height + 7) / 8 rows, one byte each) through which some NON-EMPTY leaf of the octree over
 * [root_lo, root_hi] may be seen by the closed-lens camera -- every leaf whose bit is set in occ[3..18] (path = 64 c1 + 8 c2 + c3, child index
 * c = x high | y LOW << 1 | z high << 2: TraceParams::occ), its box projected and grown by `margin` pixels; what the renderer refines its
 * never-traced mask with (csrc/vpt_caches.hip: build_leaf_tiles).  Host only.  VPT_E_UNSUPPORTED: a leaf corner at or behind the camera plane. */
int vpt_test_leaf_tiles(const vpt_camera *cam, const float root_lo[3], const float root_hi[3], const unsigned int occ[19], int width, int height,
                        float margin, unsigned char *tiles);
/* pixels the never-traced mask of the LAST render holds (0 when it had none); synchronises the device */
int vpt_test_count_never_traced(vpt_ctx *ctx, unsigned long long *pixels);
/* one c-blosc chunk (the compressed-buffer framing OpenVDB >= 224 writes) through the reader's own decoder (csrc/vpt_io.hip):
 * 0 on success, VPT_E_IO when the chunk is malformed (message in vpt_io_last_error) */
int vpt_io_test_blosc_decode(const unsigned char *src, size_t n, unsigned char *dst, size_t nbytes_out);
#ifdef __cplusplus
}
#endif
#endif
