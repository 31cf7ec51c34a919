// vpt_trace.hip -- the hot path: persistent wavefront tracer for gfx950 (wave64).
//
// What it computes: for every pixel-sample (pixel idx, iteration) the integrator part of
// the reference's `volume_rt_kernel` (source/render_kernel.cu:2227-2260): Philox stream
// init, jittered thin-lens primary ray, the depth pass (depth_calculator :1859), and
// direct_integrator (:1760) with delta tracking (`sample` :1556), residual-ratio
// tracking (`Tr` :1138), Henyey-Greenstein scattering (`sample_hg` :306), sun /
// point-light next-event estimation (:1478, :1445), emission (:1275) and the reference
// sphere bounce (:1807-1834).  The environment tail, accumulation and tonemap live in
// vpt_tail.hip (tail_resolve_kernel).
//
// How it is organised (nothing like the reference's one-thread-per-pixel megakernel):
//   * persistent waves; every lane runs a path STATE MACHINE whose heavy state is "do
//     one tracking step".  All walk kinds (delta tracking, ratio tracking for shadow rays,
//     emission marching) share one step body (octree point location -> empty-node push |
//     exponential step -> density look-up), so a wave executes the step body with lanes in
//     different paths, bounces and walk kinds side by side -- divergence is confined to the
//     short transition code, whose states are laid out in successor order so that a chain
//     of transitions completes in one pass;
//   * ray generation is its own kernel (raygen_kernel): one thread per pixel-sample at full
//     lane utilisation; primary rays that miss both the volume box and the sphere are final
//     there and never enter the tracer, the others are COMPACTED into a work queue with a wave
//     ballot + one atomic per wave;
//   * lanes whose path ended are refilled from that queue (popcount prefix over the idle
//     ballot, ONE atomic per wave); a refill is a 64-byte record load, so it is cheap enough
//     to run whenever >= regen_min lanes are idle;
//   * the depth pass and the integrator's first delta-tracking walk replay the same Philox
//     stream from the same ray (rng by value, :1860 vs :1761), so they are walked ONCE; the
//     only state the replay would change, Alpha (:1670), is re-accumulated from a per-lane
//     LDS history of the looked-up densities (exact; falls back to a real replay if the
//     history overflows);
//   * Philox4x32-10 blocks are generated at two fixed points per loop iteration (not at
//     every draw site): each lane keeps one block plus one carried-over word, enough for
//     the <= 2 draws any state consumes between refill points;
//   * the Philox counter, not the schedule, defines a sample: results are bit-identical
//     for any grid size / refill order;
//   * octree occupancy bits sit in LDS; node boxes are re-derived by halving.
#include "vpt_dome.h"
#include "vpt_trace_direct.h"

namespace vpt {

// ---- stage 0: ray generation + compaction ----------------------------------------------------------
// volume_rt_kernel :2227-2251 for every pixel-sample of the batch, one thread each (full lanes).
// Writes into records[slot] either
//   * the FINAL path record, when the sample is not rendered (:2254) or its primary ray hits
//     neither the volume box nor the sphere (get_closest_object == 0: every iteration of
//     direct_integrator's loop is then a no-op and depth = 0), or
//   * a 64-byte RAY record {org0, t_hit | dir0, obj | philox block | counter, word, depth} and the
//     slot index into the work queue (wave-ballot compaction, one atomic per wave).
// ROWS: pixel rows per raygen block (block = 64 x 4 threads, ROWS / 4 passes): 64 for batches (one queue-tail atomic per 4096
// samples), 16 for launches of a few iterations (the per-frame call, main.cpp:1822-1829: four times the blocks to fill the chip)
#ifndef VPT_RAYGEN_WAVES_PER_EU
#define VPT_RAYGEN_WAVES_PER_EU 7          // 72 registers, no spill (8: 64 registers + 8 spilled, slower; profiles/r04_four_waves.txt)
#endif
// LENSRES: behind an open lens with resolved samples raygen resolves the untraced ones from their origin's dome (its own instantiation: the look-up's registers
// would cost the closed-lens kernel two spilled dwords at seven waves per SIMD)
#ifndef VPT_TR_CONVEX_EXIT
#define VPT_TR_CONVEX_EXIT 1               // (round 6: vpt_walk.h TRX; 0 = every Tr walk pushes on to the root's far side, as rounds 1-5)
#endif
#ifndef VPT_RAYGEN_PUSH_MIN
#define VPT_RAYGEN_PUSH_MIN 1              // (study switch, round 6: see the push loop.  8: raygen -2.7 % on config 2 -- NOT adopted, it makes WHICH approximation serves a sample depend on the wave it sat in)
#endif
// (round 6: SIX waves per SIMD for the open-lens instantiation -- 74 registers, no spill; at seven it kept 72 and spilled 7 dwords since its footprint became a square: config 5's raygen -2.6 %)
#ifndef VPT_RAYGEN_LENS_WAVES
#define VPT_RAYGEN_LENS_WAVES 6
#endif
template <bool COUNT, int VPT_RAYGEN_ROWS, bool LENSRES>
__global__ __launch_bounds__(256, LENSRES ? VPT_RAYGEN_LENS_WAVES : VPT_RAYGEN_WAVES_PER_EU) void raygen_kernel(const TraceParams P) {
    // grid: tiles x iterations (1-D, tile-major); a block sweeps a 64x64 pixel tile in 16 passes and
    // compacts its active rays in LDS, so the global queue tail sees ONE atomic per 4096 samples
    __shared__ uint32_t s_q[64 * VPT_RAYGEN_ROWS];
    __shared__ uint32_t s_n, s_base;
    __shared__ uint32_t s_occ[20];
    if (threadIdx.x == 0 && threadIdx.y == 0) s_n = 0;
    if (threadIdx.y == 0 && threadIdx.x < 19) s_occ[threadIdx.x] = P.occ[threadIdx.x];
    __syncthreads();
    uint32_t n_empty_skips = 0;                       // counting builds: empty-node pushes of the rays resolved here
    // 1-D grid in TILE-major order, the batch's iterations of one 64x64 tile back to back: the queue then
    // holds a tile's rays of all iterations contiguously, so the rays the tracer has in flight at any time
    // come from a few neighbouring tiles and touch a small part of the grid (L2 / Infinity Cache locality
    // for grids that do not fit them)
    const uint32_t tiles_x = (P.width + 63u) / 64u;
    const uint32_t tile = blockIdx.x / P.iter_count;
    const uint32_t kiter = blockIdx.x - tile * P.iter_count;
    const uint32_t tile_y = tile / tiles_x, tile_x = tile - tile_y * tiles_x;
    const uint32_t iteration = P.iter_begin + kiter * P.iter_stride;
    const int lane = __lane_id();
    const bool rendered = iteration < P.max_interactions && P.render;
    uint32_t n_final = 0;
    // A WAVE'S FOOTPRINT IS AN 8 x 8 PIXEL SQUARE (round 5; rounds 1-4: 64 pixels of one row), aligned to the 8 x 8 tiles the never-traced mask is decided on
    // (sky_patch_kernel: cull_tiles[(y >> 3), (x >> 3)]): a wave is then live or skipped as a whole instead of running the whole body for the live eighths of its
    // strip, the empty-node pushes and the traced-only Philox blocks see rays that are neighbours in both directions, and the tracer's refills (consecutive queue
    // entries) take their rays from one square.  Per pass the block's four waves take the squares 4 pass .. 4 pass + 3 of the tile (row-major, 8 per row); every
    // store of a wave still covers whole 128-byte lines (8 pixels x 16 bytes per row of the square).  The sample is keyed by (pixel, iteration): results do not move.
    // WHERE NO MASK EXISTS (config 4, a view without sky patches) nothing is skipped and the square's 128-byte store runs cost ~0.2 ms per launch against the 1 KB runs of a
    // row: the footprint follows the mask (round 6, TraceParams::raygen_squares; launch-uniform).
    const bool squares = P.raygen_squares != 0u;
    const int sq_x = (int)(tile_x * 64u) + (squares ? (lane & 7) : lane), sq_y = (int)(tile_y * VPT_RAYGEN_ROWS) + (squares ? (lane >> 3) : 0);
    // the never-traced flags of this thread's pixels (one per pass), requested together up front: one memory latency per thread
    // instead of a dependent load at the head of every pass
    uint32_t never_bits = 0;
    if (P.never_traced) {
#pragma unroll
        for (int pass = 0; pass < VPT_RAYGEN_ROWS / 4; ++pass) {
            const int sq = pass * 4 + (int)threadIdx.y;
            const int xx = sq_x + (squares ? (sq & 7) * 8 : 0), yy = sq_y + (squares ? (sq >> 3) * 8 : sq);
            const uint32_t f = (xx < (int)P.width && yy < (int)P.height) ? (uint32_t)P.never_traced[(uint32_t)yy * P.width + (uint32_t)xx] : 0u;
            never_bits |= (f & 1u) << pass;
        }
    }
    for (int pass = 0; pass < VPT_RAYGEN_ROWS / 4; ++pass) {
        const int sq = pass * 4 + (int)threadIdx.y;
        const int x = sq_x + (squares ? (sq & 7) * 8 : 0), y = sq_y + (squares ? (sq >> 3) * 8 : sq);
        bool enqueue = false;
        uint32_t s = 0;
#if VPT_QREC
        bool store_q = false;
        float4 qrec0 = make_float4(0, 0, 0, 0), qrec1 = qrec0;
#endif
        bool live = x < (int)P.width && y < (int)P.height;
        if (live && ((never_bits >> pass) & 1u)) {
            // no ray of this pixel can start a walk and the tail has its samples' values (ResolveParams::never_traced): nothing to emit
            live = false;
            n_final++;
        }
        if (live) {
            const uint32_t pixel = (uint32_t)y * P.width + (uint32_t)x;
            s = kiter * P.n_pixels + pixel;
            const float2 bn = P.blue_noise[(size_t)kiter * 65536 + (y % 256) * 256 + (x % 256)];
            // volume_rt_kernel divides the jittered pixel position by the image extent (render_kernel.cu:2243-2244): per-launch constants, so the
            // quotient is formed as in to_unit (vpt_trace_common.h) where the host has checked both extents; a zero numerator divides
            const float qu = (float)(x + bn.x), qv = (float)(y + bn.y);
            float u, v;
            if (__all(P.fast_uv != 0 && fmin_(qu, qv) >= 0x1p-40f)) {
                const float yu = qu * P.rcp_w, yv = qv * P.rcp_h;
                u = __builtin_fmaf(__builtin_fmaf(-(float)P.width, yu, qu), P.rcp_w, yu);
                v = __builtin_fmaf(__builtin_fmaf(-(float)P.height, yv, qv), P.rcp_h, yv);
            } else {
                u = qu / (float)P.width;
                v = qv / (float)P.height;
            }
            // camera::get_ray, camera.h:131-136.  With a closed lens (lens_radius == 0) the lens sample is multiplied by zero:
            // offset = u * (0 * pd.x) + v * (0 * pd.y) = +-0, so the ray does not depend on the stream at all -- it is built
            // and tested first, and only a ray that goes on to the tracer (41 % of config 2) pays for the Philox block(s)
            // and the rejection loop that position its stream behind get_ray's draws.
            Rng rng;
            uint32_t draws = 0;
            const bool closed = P.cam.lens_radius == 0.0f;
            f3 offset = mk3(0.0f);
            if (!closed) {
                rng_init(rng, pixel, iteration * 4096u);
                f3 pd;
                do {
                    float a = van_der_corput(P.vdc_tables, rng, pixel, draws);
                    float b = van_der_corput(P.vdc_tables + 101, rng, pixel, draws);
                    pd = 2.0f * mk3(a, b, 0) - mk3(1.0f, 1.0f, 0.0f);
                } while (dot(pd, pd) >= 1.0f);
                const f3 rdk = P.cam.lens_radius * pd;
                offset = ld3(P.cam.u) * rdk.x + ld3(P.cam.v) * rdk.y;
                (void)rnd_simple(rng, pixel, draws);                          // the `time` draw
            }
            const f3 org0 = ld3(P.cam.origin) + offset;
            const f3 B = ld3(P.cam.llc) + u * ld3(P.cam.horizontal) + v * ld3(P.cam.vertical) - ld3(P.cam.origin) - offset;
            const f3 dir0 = normalize(B);
            float4* dst = reinterpret_cast<float4*>(P.records + s);
            int obj = 0;
            float t_hit = 0.0f, t_box = 0.0f;
            const f3 inv0 = rcp3(dir0);
            if (rendered) obj = closest_object(P, org0, dir0, inv0, t_hit);
            bool traced = obj != 0;
            if (P.integrator != 0) {
                // vol_integrator (:1732) enters its loop iff the ray hits the root box, whatever
                // the sphere does; depth_calculator (:1875) still uses get_closest_object
                float t_far;
                traced = rendered && box_intersect(ld3(P.root_pmin), ld3(P.root_pmax), org0, inv0, t_box, t_far);
            }
            // direct_integrator: a ray whose walk only ever crosses EMPTY octree nodes draws no random number and looks nothing up.
            // sample() pushes it from node to node (:1613-1616) until it is outside the root (:1606), returns WHITE without an
            // interaction, and if get_closest_object finds nothing from there (:1806) every later loop iteration is a no-op: the
            // path ends with L = 0, beta = 1, alpha = 0, depth = 0 and its primary ray -- exactly what a ray that misses the box
            // ends with.  27 % of config 2's traced rays are of this kind (the dragon fills a fraction of its padded bounding
            // box); their pushes are walked here, with the tracer's own operations, and they never enter the queue.
            // ... and a ray that does reach a leaf keeps the position its pushes took it to: the record of a box-first ray has three
            // idle words (t_hit is only used to form the start position, depth is 0, t_box is the vol_integrator's), so it carries the
            // advanced position in them (flag + push count in the obj word) and the tracer's first walk starts there instead of
            // repeating the pushes at its own lane occupancy.
            f3 adv_pos = mk3(0.0f);
            uint32_t adv = 0u;
            // (round 5, measured again: these pushes cost raygen 0.22 of its 1.13 ms on config 2 and save the tracer 0.46 ms, profiles/r05_compact_rays.txt)
            if (traced && P.integrator == 0 && obj == 1 && !P.octree_full_single) {
                f3 pos = org0;
                pos += dir0 * (t_hit + VPT_EPS);                                    // :1783-1785, as the tracer's refill does
                const OccTop occ_top = {s_occ[0], s_occ[1], s_occ[2]};
                f3 nmin = mk3(0.0f), nmax = mk3(0.0f);
                int leaf = 0, st = LOC_EMPTY;
                uint32_t pushes = 0;
#if VPT_RAYGEN_PUSH_MIN > 1
                // (study switch, round 6; as the tracer's skip loop: vpt_walk.h VPT_SKIP_MIN) the rounds go on while at least VPT_RAYGEN_PUSH_MIN lanes of the wave are still
                // crossing empty nodes; the few with longer runs leave with the position reached -- still inside an empty node -- and the tracer's own loop continues from
                // there with the same operations.  Walk decisions, depth and alpha cannot move -- but a ray that would have been found to cross empty nodes only (final here,
                // its sky value from the pixel's PATCH) is then finished by the tracer (its sky value from the DOME): two value-only approximations of the same
                // function, 1e-4 apart, and which one a sample gets would depend on its wave's other lanes -- on the footprint, on the never-traced mask being on or off.
                // tests/test_gpu_atmosphere.py::test_never_traced_pixels_change_nothing caught it; the 0.03 ms are not worth an image that depends on the schedule.
                bool pushing = true;
#pragma unroll 1
                for (int it = 0; it < 32; ++it) {
                    if (pushing) {
                        st = locate(P, s_occ, occ_top, pos, nmin, nmax, leaf);
                        if (st != LOC_EMPTY) pushing = false;
                        else {
                            float t_min, t_max;
                            box_intersect(nmin, nmax, pos, inv0, t_min, t_max);
                            t_max = fmax_(t_max, 0.1f);
                            pos += dir0 * t_max;
                            pushes++;
                        }
                    }
                    if ((int)__popcll(__ballot(pushing)) < VPT_RAYGEN_PUSH_MIN) break;
                }
#else
#pragma unroll 1
                for (int it = 0; it < 32; ++it) {
                    st = locate(P, s_occ, occ_top, pos, nmin, nmax, leaf);
                    if (st != LOC_EMPTY) break;
                    float t_min, t_max;
                    box_intersect(nmin, nmax, pos, inv0, t_min, t_max);
                    t_max = fmax_(t_max, 0.1f);
                    pos += dir0 * t_max;
                    pushes++;
                }
#endif
                if (st == LOC_OUTSIDE) {
                    float t2;
                    if (closest_object(P, pos, dir0, inv0, t2) == 0) {
                        traced = false;
                        if (COUNT) n_empty_skips += 2u * pushes;                    // depth pass + integrator: the reference walks them twice
                    }
                } else if (pushes != 0u) {
                    adv_pos = pos;
                    adv = 0x80u | (pushes << 8);
                }
            }
            if (closed && traced) {
#ifdef VPT_RAYGEN_REJECTION_LOOP
                // (rounds 1-5, kept as the A/B switch of the block-wise form below: same stream position, same bits)
                rng_init(rng, pixel, iteration * 4096u);
                f3 pd;
                do {
                    float a = van_der_corput(P.vdc_tables, rng, pixel, draws);
                    float b = van_der_corput(P.vdc_tables + 101, rng, pixel, draws);
                    pd = 2.0f * mk3(a, b, 0) - mk3(1.0f, 1.0f, 0.0f);
                } while (dot(pd, pd) >= 1.0f);
                (void)rnd_simple(rng, pixel, draws);                          // the `time` draw
#else
                // random_in_unit_disk's rejection loop (camera.h:65-75) decides nothing behind a closed lens but HOW MANY words get_ray consumes: attempt k reads words
                // 2k - 2 and 2k - 1 of the stream, the accepted attempt K is followed by the `time` draw (word 2K), and the tracer's stream starts at word 2K + 1.  A
                // Philox block holds two attempts, so the loop is run BLOCK-WISE (round 6): both attempts of a block are tested together (their four table reads in
                // flight at once instead of a dependent load pair per attempt), and one more block is generated while any lane is still undecided or accepted at a
                // block's second attempt (its `time` draw is word 0 of the next block).  Same words, same tests, same position: {block, counter, word index, draws}
                // are what the loop above leaves.  The draw-by-draw form cost raygen 0.18 of its 1.06 ms on config 2 (profiles/r06_raygen_blockwise.txt).
                uint32_t cb = iteration * 1024u, b0, b1, b2, b3;
                philox_block(cb, pixel, b0, b1, b2, b3);
                uint32_t accepted = 0u;                                       // K, 0 while undecided
                bool placed = false;                                          // the block that holds word 2K is in {b0..b3}
                for (uint32_t j = 0u;; ++j) {
                    if (accepted == 0u) {
                        const bool a1 = lens_sample_accepted(P.vdc_tables, b0, b1), a2 = lens_sample_accepted(P.vdc_tables, b2, b3);
                        accepted = a1 ? 2u * j + 1u : (a2 ? 2u * j + 2u : 0u);
                        placed = a1;                                          // K odd: the `time` draw is word 2 of this very block
                    }
#if VPT_QREC && !defined(VPT_RAYGEN_ALL_BLOCKS)
                    // QUEUE-ORDERED records carry the stream's POSITION, not its block (the tracer re-generates the block from the counter): a lane accepted at a block's
                    // second attempt only steps its counter -- one more block is generated only while a lane is still UNDECIDED (both attempts rejected: 4.6 % of the lanes
                    // per block, not the 21 % that also end on an even attempt).  Raygen -3 % on config 2 (profiles/r06_raygen.txt).  (-DVPT_RAYGEN_ALL_BLOCKS: as before, the A/B.)
                    if (P.compact_rays && accepted != 0u && !placed) { cb += 1u; placed = true; }
#endif
                    if (!__any(!placed)) break;
                    if (!placed) {
                        cb += 1u;
                        philox_block(cb, pixel, b0, b1, b2, b3);
                        placed = accepted != 0u;                              // K even: the `time` draw is word 0 of the block just generated
                    }
                }
                rng.c0 = cb; rng.o0 = b0; rng.o1 = b1; rng.o2 = b2; rng.o3 = b3;
                rng.idx = (accepted & 1u) ? 3u : 1u;                          // next word behind the `time` draw
                rng.carry = 0u; rng.has_carry = 0u;
                draws = 2u * accepted + 1u;
#endif
            }
            // :1883-1888: sphere first -> depth is the distance to it
            const float depth = (obj == 2) ? length(org0 - (org0 + dir0 * t_hit)) : 0.0f;
            // one 64-byte line per sample, written as four 16-byte stores whatever the branch (selecting the
            // VALUES per branch keeps the stores vectorised; per-branch stores degrade to 16 dword stores)
            const float l = rendered ? 0.0f : 1.0f, b = rendered ? 1.0f : 0.0f;
            float4 r0, r1, r2, r3;
            if (!traced) {
                // final: L = 0 (or WHITE when not rendering, :2248), beta = 1 (0), tr = 0
                r0 = make_float4(l, l, l, 0.0f);
                r1 = make_float4(b, b, b, depth);
                r2 = make_float4(org0.x, org0.y, org0.z, __uint_as_float(rendered ? 1u : 0u));
                r3 = make_float4(dir0.x, dir0.y, dir0.z, 0.0f);
                n_final++;
            } else {
                r0 = make_float4(org0.x, org0.y, org0.z, adv ? adv_pos.x : t_hit);
                r1 = make_float4(dir0.x, dir0.y, dir0.z, __uint_as_float((uint32_t)obj | adv));
                r2 = make_float4(__uint_as_float(rng.o0), __uint_as_float(rng.o1), __uint_as_float(rng.o2), __uint_as_float(rng.o3));
                r3 = make_float4(__uint_as_float(rng.c0), __uint_as_float(rng.idx), adv ? adv_pos.y : depth, adv ? adv_pos.z : t_box);
                enqueue = true;
            }
            if (LENSRES && P.heads && !traced && rendered) {
                // OPEN LENS + RESOLVED SAMPLES (round 5): an untraced sample's value is its environment term alone -- L = 0, beta = 1 -- and behind an open lens
                // that is a look-up in the dome of its origin's variant (what the tail did per sample, one dependent load after the other): done here, at full
                // lanes, the sample leaves as ONE 16-byte head {value, depth} (no origin stream).  What the dome cannot serve (no variant for this origin, a
                // flagged cell) keeps the 64-byte final record and goes to sky_fix_kernel through queue2, like a traced path the tracer could not resolve.
                f3 dv;
                const int dcv = dome_variant(P.resolve.sky_view, org0, ld3(P.resolve.sun_dir), P.resolve.earth_bottom);
                const bool served = dcv >= 0 && dome_lookup(P.resolve.sky_dome + (size_t)dcv * ((size_t)SKY_DOME_NU * SKY_DOME_NV), dir0, dv);
                if (served) {
                    const f3 val = mk3(0.0f) + dv * mk3(1.0f);                          // (the tail's `value += dv * beta` with value = L = 0, beta = 1)
                    st_stream(P.heads + s, make_float4(val.x, val.y, val.z, depth));
                } else {
                    st_stream(P.heads + s, make_float4(dir0.x, dir0.y, dir0.z, -1.0f));
                    st_stream(dst, r0); st_stream(dst + 1, r1); st_stream(dst + 2, r2); st_stream(dst + 3, r3);
                }
                const unsigned long long qm = __ballot(!served);
                if (qm != 0ull) {
                    const int ql = __ffsll((long long)qm) - 1;
                    uint32_t qb = 0;
                    if (lane == ql) qb = atomicAdd(P.resolve.queue2_tail, (uint32_t)__popcll(qm));
                    qb = (uint32_t)__shfl((int)qb, ql);
                    if (!served) P.resolve.queue2[qb + (uint32_t)__popcll(qm & ((1ull << lane) - 1ull))] = s;
                }
            } else if (P.heads) {
                // compact stream: a sample that starts no walk is fully described by its 16-byte head {dir0, depth} --
                // plus its origin when the lens is open (lens_radius == 0: org0 is the camera origin for every sample)
                // (with queue-ordered records the tracer no longer reads a traced sample's direction from its head -- but the head's w = -1 still tells sky_fix_kernel's
                // pass over the pixels without a patch, and the non-resolving tails, that the sample is the tracer's: it stays)
                st_stream(P.heads + s, make_float4(dir0.x, dir0.y, dir0.z, traced ? -1.0f : (rendered ? depth : -2.0f)));
                if (P.head_org && !traced) st_stream(P.head_org + s, make_float4(org0.x, org0.y, org0.z, 0.0f));
                if (traced && P.compact_rays) {
                    // 32 bytes (TraceParams::compact_rays): the origin is the camera's, the direction is in the head, the counter follows from the iteration
                    const uint32_t word = ((uint32_t)obj | adv) | (rng.idx << 14) | ((rng.c0 - iteration * 1024u) << 17);
#if VPT_QREC
                    // QUEUE-ORDERED records (round 6, vpt_device.h VPT_QREC): the record goes where the sample stands in its block's queue (below, once that place is known) and carries
                    // {direction, slot} instead of the Philox block -- the tracer re-generates the block from the counter, reads no head, and a refill's records are one contiguous run
                    qrec0 = make_float4(r0.w, r3.z, r3.w, __uint_as_float(word));
                    qrec1 = make_float4(dir0.x, dir0.y, dir0.z, __uint_as_float(s));
                    store_q = true;
#else
                    float4* d32 = P.rays32 + 2u * (size_t)s;
                    st_stream(d32, make_float4(r0.w, r3.z, r3.w, __uint_as_float(word)));
                    st_stream(d32 + 1, r2);
#endif
                } else if (traced) { st_stream(dst, r0); st_stream(dst + 1, r1); st_stream(dst + 2, r2); st_stream(dst + 3, r3); }
            } else {
                st_stream(dst, r0); st_stream(dst + 1, r1); st_stream(dst + 2, r2); st_stream(dst + 3, r3);
            }
        }
        const unsigned long long m = __ballot(enqueue);
        if (m != 0ull) {
            const int leader = __ffsll((long long)m) - 1;
            uint32_t base = 0;
            if (lane == leader) base = atomicAdd(&s_n, (uint32_t)__popcll(m));      // LDS atomic
            base = __shfl(base, leader);
#if VPT_QREC
            if (enqueue) {
                const uint32_t lidx = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                if (store_q) {
                    const uint32_t qpos = blockIdx.x * (64u * (uint32_t)VPT_RAYGEN_ROWS) + lidx;      // this block's own run of the (padded) record array
                    float4* d32 = P.rays32 + 2u * (size_t)qpos;
                    st_stream(d32, qrec0);
                    st_stream(d32 + 1, qrec1);
                    s_q[lidx] = qpos;
                } else {
                    s_q[lidx] = s;
                }
            }
#else
            if (enqueue) s_q[base + __popcll(m & ((1ull << lane) - 1ull))] = s;
#endif
        }
    }
    __syncthreads();
    const uint32_t n = s_n;
    const uint32_t tid = threadIdx.y * 64u + threadIdx.x;
#if VPT_QREC
    if (P.piece_max != 0u) {
        // QUEUE OF PIECES (vpt_device.h): this block's records are [run, run + n) of rays32 -- appended as even pieces of whole waves' worth, sized by how much of the launch is left
        if (n) {
            const uint32_t left = (gridDim.x - blockIdx.x) * (64u * (uint32_t)VPT_RAYGEN_ROWS);
            uint32_t size = min(max(left / P.piece_div, P.piece_min), P.piece_max);
            uint32_t np = (n + size - 1u) / size;
            size = ((n + np - 1u) / np + 63u) & ~63u;
            np = (n + size - 1u) / size;
            if (tid == 0) { s_base = atomicAdd(P.queue_tail, np); atomicAdd(P.queue_tail + 1, n); }       // (+ 1: the rays, for vpt_render_stats::queued_rays)
            __syncthreads();
            if (tid < np) {
                const uint32_t first = tid * size;
                st_stream(reinterpret_cast<uint2*>(P.queue) + s_base + tid, make_uint2(blockIdx.x * (64u * (uint32_t)VPT_RAYGEN_ROWS) + first, min(size, n - first)));
            }
        }
    } else
#endif
    {
    if (tid == 0 && n) s_base = atomicAdd(P.queue_tail, n);
    __syncthreads();
    const uint32_t gbase = s_base;
    for (uint32_t i = tid; i < n; i += 256u) st_stream(P.queue + gbase + i, s_q[i]);
    }
    if (COUNT && n_final) atomicAdd(&P.counters->samples, (unsigned long long)n_final);
    if (COUNT && n_empty_skips) atomicAdd(&P.counters->skip_steps, (unsigned long long)n_empty_skips);
}
// FOUR waves per SIMD.  Kept from forming packed-fp32 instructions (build.py: -fno-slp-vectorize) the compiler needs 128-147 registers for
// these kernels instead of 158, the plain and the instanced instantiations fit 128 without a spill (the others spill 2-4 words), and four
// workgroups per CU fit the LDS with 28 parked fields and a density history of 11 entries (8 next to the emission march's 3 KB table: an
// overflow replays the first walk, config 3 +4 % look-ups).  The fourth wave is worth +17 % throughput of the same code on config 2,
// +13 % on config 5, +10 % on config 3 (profiles/r04_four_waves.txt) -- at three waves the SIMDs issued 69 % of the time.
#ifndef VPT_TRACE_WAVES_PER_EU
#define VPT_TRACE_WAVES_PER_EU 4
#endif
#ifndef VPT_HIST_CAP_EMIT
#define VPT_HIST_CAP_EMIT 8
#endif
int trace_blocks_per_cu() { return VPT_TRACE_WAVES_PER_EU; }
// (study switch: -DVPT_NO_SUN_INV forms 1 / sun_dir in the Tr prologue again, as rounds 1-4 did)
#ifdef VPT_NO_SUN_INV
#define VPT_SUN_INV(C) rcp3((C).sun_dir)
#else
#define VPT_SUN_INV(C) (C).sun_inv
#endif
template <bool MULTI, bool COLOR, bool EMIT, bool COUNT, bool A24>
__global__ __launch_bounds__(256, VPT_TRACE_WAVES_PER_EU) void trace_kernel(const TraceParams P) {
    constexpr int HCAP = EMIT ? VPT_HIST_CAP_EMIT : VPT_HIST_CAP;
    __shared__ uint32_t s_occ[20];
    __shared__ float s_hist[HCAP * 256];              // [entry][thread]: densities seen by the fused first walk
    __shared__ float s_park[28 * 256];                // [field][thread]: path-level state parked in LDS (28 fields)
    if (threadIdx.x < 19) s_occ[threadIdx.x] = P.occ[threadIdx.x];
    if (EMIT) stage_emission_lut(P);
    __syncthreads();

    const uint32_t total = *P.queue_count;          // rays that entered the volume box / hit the sphere
    const int lane = __lane_id();
    const WalkConst K = make_walk_const(P);
    const uint32_t regen_min = P.regen_min;
    const uint32_t trans_min = P.trans_min;

    // ---- lane state -------------------------------------------------------------------
    uint32_t phase = PH_IDLE;
    uint32_t pixel = 0, kiter = 0;
    Rng rng;
    rng.c0 = rng.o0 = rng.o1 = rng.o2 = rng.o3 = rng.idx = rng.carry = rng.has_carry = 0u;
    uint32_t draws = 0;            // draws since the stream origin of this sample (offset iteration*4096)
    Walk w;                        // current walk ray + walk results
    w.pos = w.dir = w.inv = mk3(0.0f);
    w.t = w.distance = 0.0f;
    w.trw = 1.0f;
    w.alpha = 0.0f;
    w.wgt = mk3(1.0f);
    w.Ld = mk3(0.0f);
    w.mi = w.geo = w.obj2 = false;
    float* const park = s_park + threadIdx.x;
    const LdsF3 ppos = {park + 0 * 256}, pdir = {park + 3 * 256};    // path ray parked during shadow / emission walks
    const LdsF3 org0 = {park + 6 * 256}, dir0 = {park + 9 * 256};    // primary ray
    const LdsF3 env_pos = {park + 12 * 256};
    const LdsF3 beta = {park + 15 * 256}, L = {park + 18 * 256};
    const LdsF depth = {park + 21 * 256}, sph_factor = {park + 22 * 256};
    int* const parki = reinterpret_cast<int*>(park);
    const LdsI rd = {parki + 23 * 256}, vd = {parki + 24 * 256}, budget = {parki + 25 * 256}, light_index = {parki + 26 * 256};
    const LdsI cam_draws_p = {parki + 27 * 256};
    uint32_t n_hist = 0;
    int gco_obj = -1;              // cached get_closest_object result for the current (pos, dir), -1 = stale
    float gco_t = 0.0f;
    WalkCounts cnt;
    cnt.n_d = cnt.n_c = cnt.n_e = cnt.n_steps = cnt.n_skips = 0;
    bool more = true;
    int no_retry = 0;              // walk_step's retry count is the vol tracer's
    uint32_t chunk_next = 0, chunk_end = 0, chunk_base = 0;
    uint32_t qi0 = 0, qi1 = 0, qi2 = 0, qi3 = 0;       // this wave's chunk of queue entries, 4 per lane

    // section timing (perf studies): only in -DVPT_PROFILE_SECTIONS builds, and only in the non-counting
    // instantiation (the look-up counters' atomics would distort it)
#ifdef VPT_PROFILE_SECTIONS
    constexpr bool PROF = !COUNT;
#else
    constexpr bool PROF = false;
#endif
    unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, ts4 = 0;       // transitions, split (PROF builds)
    unsigned long long tskip = 0;                                         // skip loop inside the walk step (PROF builds)
    unsigned long long tr0 = 0, tr1 = 0, tr1a = 0, tr2 = 0, tr3 = 0, nr0 = 0, nr1 = 0, nr2 = 0;   // refill, split: idle test, claim, record wait, unpack; refills, lanes refilled, claims
    unsigned long long tc0 = 0, tc1 = 0, tc2 = 0, tc3 = 0, tstamp = PROF ? __builtin_readcyclecounter() : 0ull;
#define VPT_TICK(acc) do { if (PROF) { const unsigned long long now_ = __builtin_readcyclecounter(); acc += now_ - tstamp; tstamp = now_; } } while (0)
    for (;;) {
        // ==== refill idle lanes from the compacted ray queue ===================================
        // The wave owns a chunk [chunk_next, chunk_end) of queue entries at a time, so the global
        // dequeue counter sees one atomic per VPT_CHUNK rays.
        const unsigned long long idle = __ballot(phase == PH_IDLE);
        if (idle != 0ull) {
            const unsigned long long active = __ballot(1);
            const uint32_t n_idle = (uint32_t)__popcll(idle);
            if (PROF) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }       // (study builds: what the last pass left in flight is charged to the idle test, not to the claim)
            VPT_TICK(tr0);
            if (chunk_next == chunk_end && more && (n_idle >= regen_min || idle == active)) {
                if (PROF) nr2++;
                claim_chunk<MULTI ? 1 : VPT_CLAIM_COUNTERS>(P, total, lane, __ffsll((long long)active) - 1, chunk_next, chunk_end, more);
                if (PROF) { const unsigned long long now_ = __builtin_readcyclecounter(); tr1a += now_ - tstamp; }       // the atomic alone (tr1 keeps the whole claim)
                // the chunk's queue entries are fetched once, here (4 per lane), so that a refill pays one
                // memory latency (the ray record) instead of two dependent ones
                if (P.piece_max != 0u) {
                    // a queue of PIECES (vpt_device.h): the claim is a piece's number, the piece a run of consecutive records
                    if (chunk_next != chunk_end) {
                        const uint2 pc = ld_stream(reinterpret_cast<const uint2*>(P.queue) + chunk_next);
                        chunk_next = (uint32_t)__builtin_amdgcn_readfirstlane((int)pc.x);
                        chunk_end = chunk_next + (uint32_t)__builtin_amdgcn_readfirstlane((int)pc.y);
                    }
                } else {
                chunk_base = chunk_next;
                qi0 = chunk_base + (uint32_t)lane < chunk_end ? ld_stream(P.queue + chunk_base + (uint32_t)lane) : 0u;
                qi1 = chunk_base + 64u + (uint32_t)lane < chunk_end ? ld_stream(P.queue + chunk_base + 64u + (uint32_t)lane) : 0u;
                qi2 = chunk_base + 128u + (uint32_t)lane < chunk_end ? ld_stream(P.queue + chunk_base + 128u + (uint32_t)lane) : 0u;
                qi3 = chunk_base + 192u + (uint32_t)lane < chunk_end ? ld_stream(P.queue + chunk_base + 192u + (uint32_t)lane) : 0u;
                }
            }
            if (PROF) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }       // (study builds: charge the claim's latencies to the claim)
            VPT_TICK(tr1);
            const uint32_t avail = chunk_end - chunk_next;
            if (avail == 0u && !more && idle == active) {
                if (PROF && lane == 0) {
                    atomicAdd(&P.prof->coh[0], tr0); atomicAdd(&P.prof->coh[1], tr1); atomicAdd(&P.prof->coh[2], tr2); atomicAdd(&P.prof->coh[3], tr3);
                    atomicAdd(&P.prof->coh[4], nr0); atomicAdd(&P.prof->coh[5], nr1); atomicAdd(&P.prof->coh[6], nr2); atomicAdd(&P.prof->coh[7], tr1a);
                    atomicAdd(&P.prof->cycles[0], tc0 + tr0 + tr1 + tr2 + tr3); atomicAdd(&P.prof->cycles[1], tc1);
                    atomicAdd(&P.prof->cycles[2], tc2); atomicAdd(&P.prof->cycles[3], tc3 + ts0 + ts1 + ts2 + ts3 + ts4);
                    atomicAdd(&P.prof->sched[0], ts0); atomicAdd(&P.prof->sched[1], ts1); atomicAdd(&P.prof->sched[2], ts2);
                    atomicAdd(&P.prof->sched[3], ts3); atomicAdd(&P.prof->sched[4], ts4);
                    atomicAdd(&P.prof->sched[5], tskip + cnt.n_skips);
                }
                break;
            }
            if (avail != 0u && (n_idle >= regen_min || idle == active)) {
                const uint32_t first = chunk_next;
                const uint32_t take = min(n_idle, avail);
                chunk_next += take;
                // entry e of the chunk sits in word (e >> 6) of lane (e & 63); all lanes take part in the exchange
                const uint32_t rank = __popcll(idle & ((1ull << lane) - 1ull));
                const bool pieces = P.piece_max != 0u;
                const uint32_t rel = first + rank - chunk_base;
                const int src_lane = (int)(rel & 63u);
                uint32_t e0 = 0, e1 = 0, e2 = 0, e3 = 0;
                if (!pieces) { e0 = __shfl(qi0, src_lane); e1 = __shfl(qi1, src_lane); e2 = __shfl(qi2, src_lane); e3 = __shfl(qi3, src_lane); }
#ifdef VPT_PROFILE_SECTIONS
                float4 q0 = make_float4(0, 0, 0, 0), q1 = q0, q2 = q0, q3 = q0;
                uint32_t pk_ = 0, pp_ = 0;
#endif
                if (phase == PH_IDLE) {
                    if (rank < take) {
                        const uint32_t word = rel >> 6;
                        const uint32_t entry = pieces ? first + rank : (word == 0u ? e0 : (word == 1u ? e1 : (word == 2u ? e2 : e3)));
                        uint32_t new_kiter, new_pixel;
#ifdef VPT_PROFILE_SECTIONS
                        // (study builds: the wave waits for its records HERE, outside the divergent block, and times the wait apart from the unpacking)
                        load_ray_record(P, entry, new_kiter, new_pixel, q0, q1, q2, q3);
                        pk_ = new_kiter; pp_ = new_pixel;
                    }
                }
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                VPT_TICK(tr2);
                if (phase == PH_IDLE) {
                    if (rank < take) {
                        const uint32_t new_kiter = pk_, new_pixel = pp_;
#else
                        float4 q0, q1, q2, q3;
                        load_ray_record(P, entry, new_kiter, new_pixel, q0, q1, q2, q3);
#endif
                        kiter = new_kiter;
                        pixel = new_pixel;
                        const uint32_t iteration = P.iter_begin + kiter * P.iter_stride;
                        // obj word: bit 7 = (q0.w, q3.z, q3.w) is the position raygen's empty-node pushes reached (a box-first ray:
                        // depth 0, t_hit used up), bits 8.. = how many pushes that took
                        const uint32_t objw = __float_as_uint(q1.w);
                        const bool advanced = (objw & 0x80u) != 0u;
                        const f3 origin = mk3(q0.x, q0.y, q0.z);
                        org0 = origin;
                        gco_t = q0.w;
                        dir0 = mk3(q1.x, q1.y, q1.z);
                        gco_obj = (int)(objw & 0x7fu);
                        rng.o0 = __float_as_uint(q2.x); rng.o1 = __float_as_uint(q2.y);
                        rng.o2 = __float_as_uint(q2.z); rng.o3 = __float_as_uint(q2.w);
                        rng.c0 = __float_as_uint(q3.x);
                        rng.idx = __float_as_uint(q3.y);
                        rng.carry = 0u; rng.has_carry = 0u;
                        depth = advanced ? 0.0f : q3.z;
                        draws = rng.c0 * 4u + rng.idx - iteration * 4096u;
                        cam_draws_p = (int)draws;                    // draws consumed by camera::get_ray
                        // depth_calculator :1859-1889 and direct_integrator :1772-1785 start from the
                        // same ray with the same rng copy
                        w.alpha = 0.0f;
                        env_pos = origin;
                        w.pos = origin;
                        w.dir = mk3(q1.x, q1.y, q1.z);
                        w.inv = rcp3(w.dir);
                        L = mk3(0.0f);
                        beta = mk3(1.0f);
                        w.mi = false;
                        rd = 1;
                        if (PROF) tskip += cnt.n_skips;
                        cnt.n_d = cnt.n_c = cnt.n_e = cnt.n_steps = cnt.n_skips = 0;
                        if (gco_obj == 1) {
                            if (advanced) {
                                w.pos = mk3(q0.w, q3.z, q3.w);
                                if (COUNT) cnt.n_skips = objw >> 8;
                            } else {
                                w.pos += w.dir * (gco_t + VPT_EPS);
                            }
                            gco_obj = -1;
                            vd = 1;
                            w.t = 0.0f; w.geo = false; w.obj2 = false; w.wgt = mk3(1.0f);
                            n_hist = 0;
                            phase = PH_W_FIRST;
                        } else {
                            phase = PH_T_OUTER_TOP;      // sphere first: cached result is reused there
                        }
                    }
                }
                if (PROF) { nr0++; nr1 += take; }
                VPT_TICK(tr3);
                // (round 6, measured and closed: the NEXT refill's records touched here through global_load_lds -- no destination register, nothing waits -- so that
                // the refill's own loads hit L2: no change, profiles/r06_closed.txt)
            }
        }

        VPT_TICK(tc0);
        // ==== one tracking step for every walking lane =====================================
        rng_top_up(rng, pixel);
        VPT_TICK(tc1);
        if (phase >= PH_W_FIRST && phase <= PH_W_LAST) {
            const int kind = phase <= PH_W_TRACK ? WALK_SAMPLE : (phase == PH_W_EMIT ? WALK_EMIT : WALK_TR);
            // (one-piece step: moving the refill behind the step, as the split-phase look-up of the vol tracer needs, costs this tracer
            // 16 % -- its refilled lanes would idle for a pass -- against 1 % gained from the overlap; measured, not used here)
            Pending no_pd;
            constexpr bool TRX = VPT_TR_CONVEX_EXIT && !MULTI && !COUNT;
            const int wr = walk_step<MULTI, COLOR, EMIT, COUNT, EMIT, A24, false, 256, HCAP, TRX>(P, s_occ, K, kind, phase == PH_W_FIRST, s_hist + threadIdx.x, n_hist, w, rng, draws, cnt, no_retry, false, no_pd);
            const bool done = wr == WALK_DONE || wr == WALK_DONE_CLEAR;
            // (TRX: Walk::geo is dead once a `sample` walk has ended -- it carries "nothing ahead of this ray" to TRACK_DONE)
            if (TRX && done && kind == WALK_SAMPLE) w.geo = wr == WALK_DONE_CLEAR;
            if (done) {
                if (phase == PH_W_FIRST) phase = PH_T_FIRST_DONE;
                else if (phase == PH_W_TRACK) phase = PH_T_TRACK_DONE;
                else if (phase == PH_W_EMIT) phase = PH_T_EMIT_DONE;
                else {
                    w.trw = tr_end(K, w);
                    phase = (phase == PH_W_SUN) ? PH_T_SUN_DONE : (phase == PH_W_PL ? PH_T_PL_DONE : PH_T_SPH_DONE);
                }
            }
        }

        VPT_TICK(tc2);
        // ==== transitions: integrator control flow between walks ===========================
        // States are visited in successor order, so e.g. TRACK_DONE -> OUTER_SECOND -> OUTER_TOP
        // -> FINISH resolves in a single pass.  Each state draws at most 2 random numbers.
        // Transition code runs with few lanes, so it is batched like the refill: entered only when
        // >= trans_min lanes wait for it or nothing else can make progress.
        const unsigned long long tmask = __ballot(phase >= PH_T_FIRST);
        // (a lower threshold once the wave's queue is empty -- 1, 8, 16 lanes -- changes nothing measurable, not even on a one-iteration launch: profiles/r05_short_launches.txt)
        // (round 6, measured and closed: ... "or fewer than 8 / 16 / 24 lanes are walking" -- +0.4 / +0.5 / +2 % tracer time; threshold 40 / 56 instead of 48: +1.6 / +0.3 %.  profiles/r06_closed.txt)
        const bool run_trans = tmask != 0ull && ((uint32_t)__popcll(tmask) >= trans_min || !__any(phase >= PH_W_FIRST && phase <= PH_W_LAST));
        if (COUNT) {
            const unsigned long long wm = __ballot(phase >= PH_W_FIRST && phase <= PH_W_LAST), im = __ballot(phase == PH_IDLE);
            if (lane == 0) {
                atomicAdd(&P.counters->sched[0], 1ull);
                atomicAdd(&P.counters->sched[1], (unsigned long long)__popcll(wm));
                atomicAdd(&P.counters->sched[2], (unsigned long long)__popcll(tmask));
                atomicAdd(&P.counters->sched[3], (unsigned long long)__popcll(im));
                if (run_trans) atomicAdd(&P.counters->sched[4], 1ull);
            }
        }
        // (the launch constants of the transition states: scalar loads per pass, see ColdConst)
        if (run_trans) {
        const ColdConst C = load_cold_const();
        while (__any(phase >= PH_T_FIRST)) {
            if (COUNT) {
                const unsigned long long tm = __ballot(phase >= PH_T_FIRST);
                if (lane == 0) {
                    atomicAdd(&P.counters->sched[5], 1ull);
                    atomicAdd(&P.counters->sched[6], (unsigned long long)__popcll(tm));
                }
            }
            rng_top_up(rng, pixel);
            bool start_tr = false;
            uint32_t tr_walk_phase = PH_IDLE, tr_done_phase = PH_IDLE;
            f3 tr_dir = mk3(0.0f), tr_inv = mk3(0.0f);

            if (phase == PH_T_FIRST_DONE) {
                // the walk just finished IS depth_calculator's walk (:1879-1881) ...
                depth = w.mi ? length(f3(org0) - w.pos) : .0f;
                // ... and direct_integrator's first sample() call (:1789), which would add the same
                // densities to Alpha a second time (:1670)
                if (w.alpha < 1.0f) {
                    if (n_hist > (uint32_t)HCAP) {
                        phase = PH_T_REPLAY;
                    } else {
                        for (uint32_t i = 0; i < n_hist; ++i)
                            if (w.alpha < 1.0f) w.alpha += s_hist[i * 256 + threadIdx.x];
                    }
                }
                if (COUNT && phase == PH_T_FIRST_DONE) {
                    // the reference walks this segment twice (depth pass + integrator): count it twice
                    cnt.n_d += cnt.n_d; cnt.n_c += cnt.n_c; cnt.n_steps += cnt.n_steps; cnt.n_skips += cnt.n_skips;
                }
                if (phase == PH_T_FIRST_DONE) phase = PH_T_TRACK_DONE;
            }
            if (phase == PH_T_REPLAY) {
                // history overflow (long walk through thin medium): replay the integrator's first walk
                // for real, from the primary ray and the post-camera rng state
                const uint32_t iteration = C.iter_begin + kiter * C.iter_stride;
                const uint32_t cam_draws = (uint32_t)(int)cam_draws_p;
                rng_init(rng, pixel, iteration * 4096u + cam_draws);
                draws = cam_draws;
                w.pos = f3(org0);
                w.dir = f3(dir0);
                w.inv = rcp3(w.dir);
                w.mi = false;
                rd = 1;
                gco_obj = -1;
                phase = PH_T_OUTER_TOP;
            }
            VPT_TICK(ts0);                       // pass entry + FIRST_DONE / REPLAY
            if (phase == PH_T_TRACK_DONE) {
                // :1789-1796
                beta *= w.wgt;
                const bool brk = is_black(f3(beta)) || w.obj2;
                if (!brk && w.mi) {
                    sample_hg(w.dir, rng, draws, C.phase_g1);
                    w.inv = rcp3(w.dir);
                }
                // (a walk that ended WALK_DONE_CLEAR -- vpt_walk.h TRX -- has get_closest_object's answer with it: nothing)
                gco_obj = (VPT_TR_CONVEX_EXIT && !MULTI && !COUNT && w.geo) ? 0 : -1;
                vd++;
                if (!brk && (int)vd <= C.volume_depth) {
                    w.mi = false;
                    w.t = 0.0f; w.geo = false; w.obj2 = false; w.wgt = mk3(1.0f);
                    phase = PH_W_TRACK;
                } else if (w.mi) {
                    // estimate_sun :1478-1516
                    ppos = w.pos;
                    pdir = w.dir;
                    start_tr = true; tr_dir = C.sun_dir; tr_inv = VPT_SUN_INV(C); tr_walk_phase = PH_W_SUN; tr_done_phase = PH_T_SUN_DONE;
                } else {
                    phase = PH_T_OUTER_SECOND;
                }
            } else if (phase == PH_T_SUN_DONE) {
                const float cos_theta = dot(f3(pdir), C.sun_dir);
                const float phase_pdf = henyey_greenstein(cos_theta, C.phase_g1);
                const f3 Lsun = mk3(w.trw) * phase_pdf;
                L += (Lsun * C.sun_color * C.sun_mult) * f3(beta);             // :1514, :1798
                if (C.num_lights > 0) {
                    budget = 10;                                                    // :1459
                    w.Ld = mk3(0.0f);
                    phase = PH_T_PL_NEXT;
                } else {
                    phase = PH_T_EMIT_CHECK;
                }
            } else if (phase == PH_T_PL_DONE) {
                if ((int)budget < C.num_lights) {
                    // point_light::Le, light.h:104-121
                    const DPointLight& lt = C.lights[(int)light_index];
                    const f3 lp = ld3(lt.pos);
                    const f3 pp = ppos;
                    const f3 wi = normalize(lp - pp);
                    const float cos_theta = dot(f3(pdir), wi);
                    const float phase_pdf = henyey_greenstein(cos_theta, C.phase_g1);
                    const float sqr_dist = length(lp * lp - pp * pp);
                    const float falloff = 1 / sqr_dist;
                    w.Ld += ld3(lt.color) * lt.power * mk3(w.trw) * phase_pdf * falloff;
                }
                budget--;
                if ((int)budget >= 0) phase = PH_T_PL_NEXT;
                else {
                    L += w.Ld * f3(beta);                                           // :1799
                    phase = PH_T_EMIT_CHECK;
                }
            }
            if (phase == PH_T_PL_NEXT) {
                // estimate_point_light :1461-1466 (1 draw)
                int li = (int)floorf(rnd(rng, draws) * C.num_lights);
                if (li > C.num_lights - 1) li = C.num_lights - 1;                    // rand()==1.0f guard
                light_index = li;
                const DPointLight& lt = C.lights[li];
                start_tr = true; tr_dir = normalize(ld3(lt.pos) - f3(ppos)); tr_inv = rcp3(tr_dir); tr_walk_phase = PH_W_PL; tr_done_phase = PH_T_PL_DONE;
            } else if (phase == PH_T_EMIT_CHECK || phase == PH_T_EMIT_DONE || phase == PH_T_SPH_DONE) {
                if (phase == PH_T_EMIT_DONE) L += w.Ld;                             // :1803
                if (phase == PH_T_SPH_DONE) L += C.sun_color * C.sun_mult * mk3(w.trw) * (float)sph_factor * f3(beta);  // :1832
                w.pos = f3(ppos);
                w.dir = f3(pdir);
                w.inv = rcp3(w.dir);
                gco_obj = -1;
                if (phase == PH_T_EMIT_CHECK && EMIT && C.emission_scale > 0) {     // :1802 (mi is true here)
                    w.t = 0.0f;
                    w.Ld = mk3(0.0f);
                    phase = PH_W_EMIT;
                } else if (phase == PH_T_SPH_DONE) {
                    env_pos = w.pos;                                                // :1833
                    rd++;
                    phase = PH_T_OUTER_TOP;
                } else {
                    phase = PH_T_OUTER_SECOND;
                }
            }
            VPT_TICK(ts1);                       // TRACK_DONE .. EMIT / SPH
            // (round 6, measured and closed: get_closest_object of OUTER_SECOND and of OUTER_TOP hoisted into one place ahead of both states -- no change, profiles/r06_closed.txt)
            if (phase == PH_T_OUTER_SECOND) {
                if (gco_obj < 0) gco_obj = closest_object(K.root_lo, K.root_hi, C.sph_center, C.sph_radius, w.pos, w.dir, w.inv, gco_t); // :1806
                if (gco_obj == 2) {
                    // sphere bounce :1809-1833 (2 draws)
                    w.pos += w.dir * gco_t;
                    const f3 normal = normalize((w.pos - C.sph_center) / C.sph_radius);
                    const f3 nl = dot(normal, w.dir) < 0 ? normal : normal * -1;
                    const float phi = 2 * VPT_PI * rnd(rng, draws);
                    const float r2 = rnd(rng, draws);
                    const float r2s = sqrtf(r2);
                    const f3 ww = normalize(nl);
                    const f3 uu = normalize(cross(((double)fabsf(ww.x) > .1 ? mk3(0, 1, 0) : mk3(1, 0, 0)), ww));
                    const f3 vv = cross(ww, uu);
                    float sp, cp;
                    det_sincosf(phi, &sp, &cp);
                    const f3 hemisphere_dir = normalize(uu * cp * r2s + vv * sp * r2s + ww * sqrtf(1 - r2));
                    const f3 ref = reflect(w.dir, nl);
                    w.dir = lerp3(ref, hemisphere_dir, C.sph_roughness);
                    w.pos += normal * VPT_EPS;
                    beta *= C.sph_color;
                    sph_factor = fmax_(dot(C.sun_dir, normal), .0f);
                    ppos = w.pos;
                    pdir = w.dir;
                    gco_obj = -1;
                    start_tr = true; tr_dir = C.sun_dir; tr_inv = VPT_SUN_INV(C); tr_walk_phase = PH_W_SPH; tr_done_phase = PH_T_SPH_DONE;
                } else {
                    rd++;                          // same ray next iteration: the cached result stays valid
                    phase = PH_T_OUTER_TOP;
                }
            }
            if (phase == PH_T_OUTER_TOP) {
                if ((int)rd > C.ray_depth) {
                    phase = PH_T_FINISH;
                } else {
                    if (gco_obj < 0) gco_obj = closest_object(K.root_lo, K.root_hi, C.sph_center, C.sph_radius, w.pos, w.dir, w.inv, gco_t);   // :1782
                    if (gco_obj == 1) {
                        w.pos += w.dir * (gco_t + VPT_EPS);
                        gco_obj = -1;
                        vd = 1;
                        w.mi = false;
                        w.t = 0.0f; w.geo = false; w.obj2 = false; w.wgt = mk3(1.0f);
                        // A ray that sat INSIDE the box (a walk that stopped at t >= distance, or a
                        // scattered ray) is moved to the box's far side here, so the walk it starts
                        // usually finds itself outside the octree at once (get_quadrant(root) == -1,
                        // :1606): no draw, no look-up, sample() returns WHITE.  That empty walk is
                        // resolved right here instead of costing a pass of the walk loop.
                        f3 nmin, nmax;
                        int leaf;
                        const OccTop occ_top = {s_occ[0], s_occ[1], s_occ[2]};
                        if (locate(P, s_occ, occ_top, w.pos, nmin, nmax, leaf) == LOC_OUTSIDE) {
                            // ... and what follows such a walk is fixed: every one of the volume_depth inner iterations returns at
                            // once (beta *= WHITE, no interaction, :1789-1796), then get_closest_object is asked again from here
                            // (:1806).  Nothing there (the usual case: the ray has left the box for good) ends the path; it is
                            // finished in this very round instead of going round the state list once more (TRACK_DONE sits before
                            // this state).  A sphere ahead goes through OUTER_SECOND with the result cached.
                            gco_obj = closest_object(K.root_lo, K.root_hi, C.sph_center, C.sph_radius, w.pos, w.dir, w.inv, gco_t);
                            if (gco_obj == 0) {
                                rd++;
                                phase = PH_T_FINISH;
                            } else {
                                vd = C.volume_depth + 1;
                                phase = PH_T_OUTER_SECOND;
                            }
                        } else {
                            phase = PH_W_TRACK;
                        }
                    } else if (gco_obj == 0) {
                        // nothing ahead: the second get_closest_object (:1806) sees the same ray, so
                        // this and every later iteration is a no-op -> finish (exact)
                        phase = PH_T_FINISH;
                    } else {
                        phase = PH_T_OUTER_SECOND;   // sphere is closest: handled next pass
                    }
                }
            }
            VPT_TICK(ts2);                       // OUTER_SECOND + OUTER_TOP
            if (phase == PH_T_FINISH) {
                const f3 od = w.dir, oL = L, ob = beta, oe = env_pos;
                const size_t slot = (size_t)kiter * C.n_pixels + pixel;
                // RESOLVED SAMPLES (TraceParams::resolve): what the tail would add to L -- beta x the sky along the exit direction, seen from the
                // camera origin -- is a dome look-up, done here where ~44 lanes finish together; the sample then leaves as 24 bytes instead of 64.
                bool resolved = false;
                const ResolveInTracer rt = load_resolve();   // (its fields are fetched here, per batch of finishing paths, not held in the loop's live scalars)
                const bool resolving = rt.sky_dome != nullptr;
                if (resolving) {
                    f3 dv;
                    // the dome that serves this path's origin: the camera origin's (closed lens: env_pos must equal it bit for bit), or the lens variant of its r
                    int dcv = -1;
                    if (rt.lens) dcv = dome_variant(rt.sky_view, oe, mk3(rt.sun_dir[0], rt.sun_dir[1], rt.sun_dir[2]), rt.earth_bottom);
                    else if (oe.x == rt.cam_origin[0] && oe.y == rt.cam_origin[1] && oe.z == rt.cam_origin[2]) dcv = 0;
                    if (dcv >= 0 && dome_lookup(rt.sky_dome + (size_t)dcv * ((size_t)SKY_DOME_NU * SKY_DOME_NV), od, dv)) {
                        const f3 val = oL + dv * ob;                                    // (the tail's `value += dv * beta`, :1838-1842)
                        // (their scatter costs the tracer 1.2 %: the same two stores to a per-thread fixed place, profiles/r05_compact_rays.txt; merged into one 32-byte sector: no better)
                        st_stream(rt.heads + slot, make_float4(val.x, val.y, val.z, -1.0f));
                        st_stream(rt.td + slot, make_float2(fmin_(w.alpha, 1.0f), depth));
                        resolved = true;
                    }
                }
                if (!resolved) {
                    float4* dst = reinterpret_cast<float4*>(C.records + slot);
                    st_stream(dst, make_float4(oL.x, oL.y, oL.z, fmin_(w.alpha, 1.0f)));      // tr = fminf(tr, 1) :1854
                    st_stream(dst + 1, make_float4(ob.x, ob.y, ob.z, depth));
                    st_stream(dst + 2, make_float4(oe.x, oe.y, oe.z, __uint_as_float(1u)));
                    st_stream(dst + 3, make_float4(od.x, od.y, od.z, 0.0f));
                    if (resolving) {
                        // the full evaluation is sky_fix_kernel's (vpt_tail.hip): one queue entry per such path, one atomic per wave
                        const unsigned long long qm = __ballot(1);
                        const int ql = __ffsll((long long)qm) - 1;
                        uint32_t qb = 0;
                        if (lane == ql) qb = atomicAdd(rt.queue2_tail, (uint32_t)__popcll(qm));
                        qb = (uint32_t)__shfl((int)qb, ql);
                        rt.queue2[qb + (uint32_t)__popcll(qm & ((1ull << lane) - 1ull))] = (uint32_t)slot;
                    }
                }
                if (COUNT) {
                    atomicAdd(&P.counters->samples, 1ull);
                    if (cnt.n_steps == 0u) atomicAdd(&P.counters->coh[6], 1ull);       // traced rays that crossed empty nodes only (no draw, no look-up)
                    atomicAdd(&P.counters->density_lookups, (unsigned long long)cnt.n_d);
                    atomicAdd(&P.counters->color_lookups, (unsigned long long)cnt.n_c);
                    atomicAdd(&P.counters->emission_lookups, (unsigned long long)cnt.n_e);
                    atomicAdd(&P.counters->tracking_steps, (unsigned long long)cnt.n_steps);
                    atomicAdd(&P.counters->skip_steps, (unsigned long long)cnt.n_skips);
                }
                phase = PH_IDLE;
            }

            VPT_TICK(ts3);                       // FINISH
            // ---- Tr prologue :1153-1167 (shared by sun / point-light / sphere shadow rays) ---
            if (start_tr) {
                phase = tr_begin(C.sph_center, C.sph_radius, K, w, f3(ppos), tr_dir, tr_inv) ? tr_walk_phase : tr_done_phase;
                w.geo = false;                     // (of a Tr walk: "has been inside a non-empty leaf", vpt_walk.h TRX; every `sample` walk clears it where it starts)
            }
            VPT_TICK(ts4);                       // Tr prologue
        }
        }
        VPT_TICK(tc3);
    }
#undef VPT_TICK
}

// ---- launcher -------------------------------------------------------------------------------
template <bool MULTI, bool COLOR, bool EMIT>
static hipError_t launch_variant(const TraceParams& P, int blocks, hipStream_t stream) {
    // A24: every volume's texel indices fit the 24-bit multiplier (vpt_scene_set_volumes), see imul
    if (P.addr24) {
        if (P.counters) hipLaunchKernelGGL((trace_kernel<MULTI, COLOR, EMIT, true, true>), dim3(blocks), dim3(256), 0, stream, P);
        else hipLaunchKernelGGL((trace_kernel<MULTI, COLOR, EMIT, false, true>), dim3(blocks), dim3(256), 0, stream, P);
    } else {
        if (P.counters) hipLaunchKernelGGL((trace_kernel<MULTI, COLOR, EMIT, true, false>), dim3(blocks), dim3(256), 0, stream, P);
        else hipLaunchKernelGGL((trace_kernel<MULTI, COLOR, EMIT, false, false>), dim3(blocks), dim3(256), 0, stream, P);
    }
    return hipGetLastError();
}

hipError_t launch_raygen(const TraceParams& P, hipStream_t stream) {
    const bool small = P.iter_count < P.raygen_small_iters;      // (17 unless VPT_RAYGEN_SMALL_ITERS says otherwise)
    const uint32_t rows = small ? 16u : 64u;
    const dim3 grid(((P.width + 63u) / 64u) * ((P.height + rows - 1u) / rows) * P.iter_count), block(64, 4, 1);
    const bool lensres = P.heads != nullptr && P.resolve.lens != 0 && P.resolve.sky_dome != nullptr;
    if (small) {
        if (P.counters) { if (lensres) hipLaunchKernelGGL((raygen_kernel<true, 16, true>), grid, block, 0, stream, P); else hipLaunchKernelGGL((raygen_kernel<true, 16, false>), grid, block, 0, stream, P); }
        else { if (lensres) hipLaunchKernelGGL((raygen_kernel<false, 16, true>), grid, block, 0, stream, P); else hipLaunchKernelGGL((raygen_kernel<false, 16, false>), grid, block, 0, stream, P); }
    } else {
        if (P.counters) { if (lensres) hipLaunchKernelGGL((raygen_kernel<true, 64, true>), grid, block, 0, stream, P); else hipLaunchKernelGGL((raygen_kernel<true, 64, false>), grid, block, 0, stream, P); }
        else { if (lensres) hipLaunchKernelGGL((raygen_kernel<false, 64, true>), grid, block, 0, stream, P); else hipLaunchKernelGGL((raygen_kernel<false, 64, false>), grid, block, 0, stream, P); }
    }
    return hipGetLastError();
}

hipError_t launch_trace(const TraceParams& P, bool multi, bool color, bool emit, int blocks, hipStream_t stream) {
    if (!multi && !color && !emit) return launch_variant<false, false, false>(P, blocks, stream);
    if (!multi && !color && emit) return launch_variant<false, false, true>(P, blocks, stream);
    if (!multi && color && !emit) return launch_variant<false, true, false>(P, blocks, stream);
    if (!multi && color && emit) return launch_variant<false, true, true>(P, blocks, stream);
    if (multi && !color && !emit) return launch_variant<true, false, false>(P, blocks, stream);
    if (multi && !color && emit) return launch_variant<true, false, true>(P, blocks, stream);
    if (multi && color && !emit) return launch_variant<true, true, false>(P, blocks, stream);
    return launch_variant<true, true, true>(P, blocks, stream);
}

}  // namespace vpt
