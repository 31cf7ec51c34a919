// vpt_walk.h -- ONE tracking step of any walk kind, shared by the tracer kernels (vpt_trace.hip:
// direct_integrator, vpt_trace_vol.hip: vol_integrator): delta tracking `sample`
// (render_kernel.cu:1556), ratio tracking `Tr` (:1138), emission march `estimate_emission` (:1275),
// plus the Tr prologue / epilogue.  Strict arithmetic (vpt_math.h).
#pragma once

#include "vpt_trace_common.h"

namespace vpt {

// sample_hg that also returns the phase value the reference returns (henyey_greenstein(-cos_theta, g),
// :324) -- estimate_sky uses it (:1404)
VPT_D float sample_hg_pdf(f3& wo, Rng& rng, uint32_t& draws, float g) {
    float cos_theta;
    if (fabsf(g) < VPT_EPS) cos_theta = 1 - 2 * rnd(rng, draws);
    else {
        float sqr_term = (1 - g * g) / (1 - g + 2 * g * rnd(rng, draws));
        cos_theta = (1 + g * g - sqr_term * sqr_term) / (2 * g);
    }
    float sin_theta = sqrtf(fmax_(.0f, 1.0f - cos_theta * cos_theta));
    float phi = (2.0f * VPT_PI) * rnd(rng, draws);
    f3 v1 = wo * -1.0f, v2, v3;
    if (fabsf(v1.x) > fabsf(v1.y)) v2 = mk3(-v1.z, 0.0f, v1.x);
    else v2 = mk3(0.0f, v1.z, -v1.y);
    v2 = normalize(v2);
    v3 = normalize(cross(v1, v2));
    float sp, cp;
    det_sincosf(phi, &sp, &cp);
    wo = v2 * sin_theta * cp + v3 * sin_theta * sp + wo * cos_theta;
    return henyey_greenstein(-cos_theta, g);
}

// Constants of a launch that every step needs (kept in SGPRs).
struct WalkConst {
    f3 root_lo, root_hi;
    float inv_max;        // 1 / root->max_extinction                         (:1645)
    float inv_dm;         // 1 / density_mult                                 (:1646)
    float sigma_c;        // control variate = root->min_extinction           (:1164)
    float sigma_r_inv;    // 1 / (max_extinction - sigma_c)                   (:1165)
};
VPT_D WalkConst make_walk_const(const TraceParams& P) {
    WalkConst K;
    K.root_lo = ld3(P.root_pmin);
    K.root_hi = ld3(P.root_pmax);
    K.inv_max = P.inv_max_ext;
    K.inv_dm = P.inv_density_mult;
    K.sigma_c = P.min_ext;
    K.sigma_r_inv = P.sigma_r_inv;
    return K;
}

// Per-lane walk state.  The same body serves the three walk kinds, so a wave runs it with lanes
// in different paths, bounces and walk kinds side by side:
//   WALK_SAMPLE  delta tracking (`sample` :1603-1678): may end in a real collision (mi, wgt)
//   WALK_TR      ratio tracking (`Tr` :1169-1265): multiplies trw
//   WALK_EMIT    emission march (`estimate_emission` :1287-1337): adds to Ld
enum { WALK_SAMPLE = 0, WALK_TR = 1, WALK_EMIT = 2 };
#ifndef VPT_SKIP_LOOP
#define VPT_SKIP_LOOP 8
#endif
#ifndef VPT_SKIP_MIN
#define VPT_SKIP_MIN 8
#endif
#ifndef VPT_SAMPLE_CONVEX_EXIT
#define VPT_SAMPLE_CONVEX_EXIT 1
#endif
#ifndef VPT_RETRY_SPINS
#define VPT_RETRY_SPINS 4
#endif
struct Walk {
    f3 pos, dir, inv;     // walk ray (inv = 1 / dir)
    float t;              // cumulative step (Q-list 1: never reset inside a walk)
    float distance;       // exit distance / distance to the sphere
    float trw;            // running transmittance of a Tr walk
    float alpha;          // `tr` of volume_rt_kernel (:2251), fed to sample() as Alpha
    f3 wgt;               // return value of sample()
    f3 Ld;                // emission sum / light sum
    bool mi, geo, obj2;
};
struct WalkCounts {
    uint32_t n_d, n_c, n_e, n_steps, n_skips;
};

// Calls f(matrix, descriptor) for every instance listed in octree leaf `leaf` (in list order).
// `cell`: index of the point's sub-cell inside the leaf (sub_cell below); only the single-file path uses it.
template <bool MULTI, class F>
VPT_D void for_each_instance(const TraceParams& P, int leaf, int cell, F&& f) {
    if (!MULTI) {
        DVolume v0;
        load_vol0(v0);
        f(v0.m, v0);
        return;
    }
    // (the instance lists' pointers: read where they are used, like the descriptor -- load_vol0)
    KargPtr k = (KargPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(k));
    if (k->single_file) {
        DVolume v0;
        load_vol0(v0);
        // refined candidate list of the sub-cell (vpt_scene_set_volumes): a subset of the leaf's list in the same order;
        // the instances left out cannot contain the point and would add nothing
        const uint32_t sc = (uint32_t)leaf * (uint32_t)VPT_SUB3 + (uint32_t)cell;
        const uint32_t* sub_offsets = k->sub_offsets;
        const uint32_t b = sub_offsets[sc], e = sub_offsets[sc + 1u];
        // instances of one file: 48-byte matrix per list entry, the rest from vol0 (SGPRs)
        typedef float __attribute__((ext_vector_type(4))) v4;
        const __attribute__((address_space(1))) v4* ip = (const __attribute__((address_space(1))) v4*)k->insts;
        // insts[] is in leaf-list order (vpt_scene_set_volumes): entry q's matrix sits at q, no index to chase, and the
        // next entry's matrix is requested before the current one is used, so its latency overlaps the transform / fetch
        v4 n0, n1, n2;
        if (b < e) { n0 = ip[b * 4u]; n1 = ip[b * 4u + 1u]; n2 = ip[b * 4u + 2u]; }
        for (uint32_t q = b; q < e; ++q) {
            const v4 r0 = n0, r1 = n1, r2 = n2;
            if (q + 1u < e) { const uint32_t vi = (q + 1u) * 4u; n0 = ip[vi]; n1 = ip[vi + 1u]; n2 = ip[vi + 2u]; }
            const float m[12] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w};
            f(m, v0);
        }
    } else {
        const uint32_t* leaf_offsets = k->leaf_offsets;
        const uint32_t* leaf_indices = k->leaf_indices;
        const DVolume* volumes = k->volumes;
        const uint32_t b = leaf_offsets[leaf], e = leaf_offsets[leaf + 1];
        for (uint32_t q = b; q < e; ++q) {
            const DVolume& v = volumes[leaf_indices[q]];
            f(v.m, v);
        }
    }
}
// sub-cell of p inside its leaf box [lo, hi]: VPT_SUB^3 of them, x fastest.  The index may be off by one for a point within
// rounding of a cell plane; the host lists are built over boxes grown by 1e-3 of a cell, which covers it.
VPT_D int sub_cell(const TraceParams& P, f3 lo, f3 p) {
    const int ix = min(max((int)((p.x - lo.x) * P.sub_inv[0]), 0), VPT_SUB - 1);
    const int iy = min(max((int)((p.y - lo.y) * P.sub_inv[1]), 0), VPT_SUB - 1);
    const int iz = min(max((int)((p.z - lo.z) * P.sub_inv[2]), 0), VPT_SUB - 1);
    return (iz * VPT_SUB + iy) * VPT_SUB + ix;
}

// Counting builds, single-volume scenes: how coherent are the density look-ups a wave issues together?  (DESIGN.md section 4:
// the evidence behind not staging brick tiles in LDS.)  Executed by the lanes that are about to fetch; wave-level sums.
template <bool A24>
VPT_D void coherence_stats(const TraceParams& P, f3 p) {
    if (((__builtin_readcyclecounter() >> 5) & 15ull) != 0ull) return;       // a wave-uniform 1-in-16 sample of the events (all figures are ratios)
    f3 u;
    const bool inside = to_unit(P.vol0.m, P.vol0, p, u);
    if (!inside) return;
    const DVolume& v = P.vol0;
    const Taps t = make_taps(v.dim, v.dimf, u, P.tex_fixed8);
    // 128-byte line of each of the 8 taps in the layout actually used
    uint32_t a[8];
    if (v.layout == GRID_QUADS) {
        uint32_t e0, e1;
        quad_entries<A24>(v.dim, t, e0, e1);
        for (int q = 0; q < 8; ++q) a[q] = ((q & 1) ? e1 : e0) >> 3;            // 8 float4 entries per line
    } else if (v.layout == GRID_BRICKS) {
        const uint32_t row = (uint32_t)v.bdim[0] * 64u, slab = (uint32_t)v.bdim[1] * row;
        const uint32_t x[2] = {(((uint32_t)t.i0 >> 2) << 6) + ((uint32_t)t.i0 & 3u), (((uint32_t)t.i1 >> 2) << 6) + ((uint32_t)t.i1 & 3u)};
        const uint32_t y[2] = {((uint32_t)t.j0 >> 2) * row + (((uint32_t)t.j0 & 3u) << 2), ((uint32_t)t.j1 >> 2) * row + (((uint32_t)t.j1 & 3u) << 2)};
        const uint32_t z[2] = {((uint32_t)t.k0 >> 2) * slab + (((uint32_t)t.k0 & 3u) << 4), ((uint32_t)t.k1 >> 2) * slab + (((uint32_t)t.k1 & 3u) << 4)};
        for (int q = 0; q < 8; ++q) a[q] = (z[q >> 2] + y[(q >> 1) & 1] + x[q & 1]) >> 5;
    } else {
        const uint32_t dx = (uint32_t)v.dim[0], dy = (uint32_t)v.dim[1];
        const uint32_t x[2] = {(uint32_t)t.i0, (uint32_t)t.i1}, y[2] = {(uint32_t)t.j0, (uint32_t)t.j1}, z[2] = {(uint32_t)t.k0, (uint32_t)t.k1};
        for (int q = 0; q < 8; ++q) a[q] = ((z[q >> 2] * dy + y[(q >> 1) & 1]) * dx + x[q & 1]) >> 5;
    }
    uint32_t own_lines = 0;
    for (int q = 0; q < 8; ++q) {
        bool seen = false;
        for (int r = 0; r < q; ++r) seen = seen || a[r] == a[q];
        own_lines += seen ? 0u : 1u;
    }
    const uint32_t b8 = (((uint32_t)t.k0 >> 3) * 4096u + ((uint32_t)t.j0 >> 3)) * 4096u + ((uint32_t)t.i0 >> 3);
    const uint32_t b4 = (((uint32_t)t.k0 >> 2) * 4096u + ((uint32_t)t.j0 >> 2)) * 4096u + ((uint32_t)t.i0 >> 2);
    const unsigned long long act = __ballot(1);
    uint32_t d8 = 0, d4 = 0, dl = 0;
    for (unsigned long long m = act; m != 0ull;) {
        const uint32_t id = (uint32_t)__shfl((int)b8, __ffsll((long long)m) - 1);
        m &= ~__ballot(b8 == id);
        d8++;
    }
    for (unsigned long long m = act; m != 0ull;) {
        const uint32_t id = (uint32_t)__shfl((int)b4, __ffsll((long long)m) - 1);
        m &= ~__ballot(b4 == id);
        d4++;
    }
    // distinct lines over all taps of the wave: retire one line id per round
    uint32_t done = 0;                                   // bit q: tap q's line has been counted
    for (;;) {
        const unsigned long long m = __ballot(done != 0xffu);
        if (m == 0ull) break;
        uint32_t cand = 0;
        for (int q = 7; q >= 0; --q) cand = ((done >> q) & 1u) ? cand : a[q];       // lowest tap not yet counted
        const uint32_t id = (uint32_t)__shfl((int)cand, __ffsll((long long)m) - 1);
        for (int q = 0; q < 8; ++q) done |= (a[q] == id) ? (1u << q) : 0u;
        dl++;
    }
    const int leader = __ffsll((long long)act) - 1;
    // wave sum of own_lines
    unsigned long long sum_own = 0;
    for (unsigned long long m = act; m != 0ull; m &= m - 1ull) sum_own += (unsigned long long)__shfl((int)own_lines, __ffsll((long long)m) - 1);
    if (__lane_id() == leader) {
        atomicAdd(&P.counters->coh[0], 1ull);
        atomicAdd(&P.counters->coh[1], (unsigned long long)__popcll(act));
        atomicAdd(&P.counters->coh[2], (unsigned long long)d8);
        atomicAdd(&P.counters->coh[3], (unsigned long long)d4);
        atomicAdd(&P.counters->coh[4], sum_own);
        atomicAdd(&P.counters->coh[5], (unsigned long long)dl);
    }
}

// Returns true when the walk ended.  hist / n_hist: per-lane LDS history of the densities seen by
// the fused first walk (record_hist), stride 256 floats.
// retries (vol_integrator only: use_retries): how many further `sample()` calls the integrator's depth
// loop would make from this very position if the current one ends without an interaction at
// t >= distance (:1654 -> :1740 next iteration).  Such a call repeats the same point location, the
// same exit distance and sphere test and differs only in its exponential draw, so it is replayed
// right here (one draw + one log per retry, as many as the buffered Philox words allow) instead of costing
// a pass of the walk loop each.
// What follows a step's density look-up: sample() accepts or rejects the collision (:1667-1675), Tr() multiplies its estimate
// (:1239, :1261).  Shared by the one-piece step and the second half of the split-phase step.  Returns true when the walk ended.
// HS: floats between two entries of a lane's density history (256: [entry][thread] in LDS; the pool tracer keeps it in HBM); HCAP: its entries
template <bool MULTI, bool COLOR, bool COUNT, bool A24, int HS = 256, int HCAP = VPT_HIST_CAP>
VPT_D bool walk_decide(const TraceParams& P, const WalkConst& K, bool is_sample, bool record_hist, float* hist, uint32_t& n_hist, Walk& w, Rng& rng,
                       uint32_t& draws, float density, int leaf, int cell) {
    if (is_sample) {
        // The density-colour LUT value only matters on a real collision, so its index (one correctly rounded divide by
        // emission_pivot) and fetch are evaluated there.
        if (w.alpha < 1.0f) w.alpha += density;
        if (record_hist) {
            if (n_hist < (uint32_t)HCAP) hist[n_hist * HS] = density;
            n_hist++;
        }
        if (density * K.inv_max > rnd(rng, draws)) {
            f3 Cd = COLOR ? mk3(0.0f) : mk3(1.0f);
            if (COLOR) {
                uint32_t z0 = 0, z1 = 0, z2 = 0;
                float dz = 0.0f;
                f3 ez = mk3(0.0f);
                for_each_instance<MULTI>(P, leaf, cell, [&](const float* m, const DVolume& v) { // sum_color :931 (component-wise max)
                    lookup_volume<COLOR, false, false, false, A24, COUNT>(P, m, v, w.pos, false, true, false, dz, Cd, ez, z0, z1, z2);
                });
            }
            const int index = (int)floorf(fmin_(fmax_((density * K.inv_max * 255.0f / P.emission_pivot), 0.0f), 255.0f));
            const float* dc = P.density_color_lut + 3 * index;
            w.mi = true;
            w.wgt = (ld3(P.albedo) * Cd * mk3(dc[0], dc[1], dc[2]) / ld3(P.extinction)) * P.energy_inject;
            return true;
        }
    } else {
        w.trw *= 1 - ((density - K.sigma_c) * K.sigma_r_inv);                 // :1239
        const float s2 = w.trw * w.trw;
        if ((s2 + s2 + s2) < VPT_SQ_OF_EPS) return true;                      // :1261 length(tr) < EPS, without the root
    }
    return false;
}

// SPLIT (single-volume kernels): the density look-up of a delta- or ratio-tracking step is SPLIT-PHASE.  walk_step only requests the
// eight texels (returns WALK_PENDING, `pd` holds them) and the caller runs whatever else the wave has to do -- refilling idle
// lanes from the ray queue, itself a ~1 us record read -- before walk_finish interpolates and decides: the two memory
// latencies of a pass overlap instead of adding up.  Per lane the operations and their order are unchanged.
enum { WALK_GOES_ON = 0, WALK_DONE = 1, WALK_PENDING = 2, WALK_DONE_CLEAR = 3 };      // (DONE_CLEAR: TRX, a `sample` walk that ended with nothing ahead of its ray)
// TRX (round 6; single-volume scenes, timed instantiations of the direct tracer): a RATIO-TRACKING walk that has been inside a non-empty leaf and now stands in an empty
// node is over.  With one volume the non-empty leaves are a BOX of leaves (every leaf its bounds overlap), the walk moves forward along a straight line, and a line that
// has left a convex set does not come back: from here to the root's far side there are only empty nodes -- pushes (:1193-1227), which draw nothing and look nothing up --
// and Tr's value is trw x exp(-sigma_c x the distance taken at its start) (:1166, :1267): nothing the remaining pushes compute is ever read.  A DELTA-TRACKING walk's
// final position is read once more -- get_closest_object starts from it, :1806 -- so it ends early only where that call's answer is known (see VPT_SAMPLE_CONVEX_EXIT in
// the loop below); counting instantiations push on in both cases: their skip counts are the oracle's, and timed vs counting is the exactness A/B
// (tests/test_gpu_edge.py::test_convex_exit_changes_nothing).  Walk::geo, which only `sample` walks read while they run, carries "has been inside a leaf" for a Tr walk
// (cleared where the walk starts) and "ended with nothing ahead" from a `sample` walk's end to TRACK_DONE.  Config 2: tracer -1.5 % (Tr walks), -9 % (`sample` walks).
template <bool MULTI, bool COLOR, bool EMIT, bool COUNT, bool ELDS, bool A24, bool SPLIT = false, int HS = 256, int HCAP = VPT_HIST_CAP, bool TRX = false>
VPT_D int walk_step(const TraceParams& P, const uint32_t* s_occ, const WalkConst& K, int kind, bool record_hist,
                    float* hist, uint32_t& n_hist, Walk& w, Rng& rng, uint32_t& draws, WalkCounts& c,
                    int& retries, bool use_retries, Pending& pd) {
    const bool is_sample = kind == WALK_SAMPLE;
    const bool is_emit = EMIT && kind == WALK_EMIT;
    // Empty-node pushes are cheap and the tracking step below is expensive, so the wave first loops
    // on the pushes until none of its walking lanes stands in an empty node (at most VPT_SKIP_LOOP
    // rounds): the tracking step then runs with (nearly) all walkers participating instead of the
    // one third that would otherwise be at a leaf in any given pass.  Per lane the sequence of
    // operations is unchanged.
    int leaf = 0;
    int st = LOC_EMPTY;
    bool clear_exit = false;
    const OccTop occ_top = {s_occ[0], s_occ[1], s_occ[2]};
#ifdef VPT_PROFILE_SECTIONS
    const unsigned long long tp0_ = __builtin_readcyclecounter();
#endif
    f3 nmin = mk3(0.0f), nmax = mk3(0.0f);       // box of the node found: an empty node's (to push through) or the leaf's
#pragma unroll 1
    for (int it = 0; it < VPT_SKIP_LOOP; ++it) {
        if (st == LOC_EMPTY) {
            st = locate(P, s_occ, occ_top, w.pos, nmin, nmax, leaf);
            if (TRX && kind == WALK_TR) {
                if (st == LOC_LEAF) w.geo = true;
                else if (st == LOC_EMPTY && w.geo) st = LOC_OUTSIDE;          // left the box of non-empty leaves for good: nothing ahead can change trw
            }
#if VPT_SAMPLE_CONVEX_EXIT
            if (TRX && is_sample && st == LOC_EMPTY && w.t > 0.0f) {
                // A DELTA-TRACKING walk that has taken a tracking step (t > 0: it has been inside a leaf) and stands in an empty node will not interact any more
                // (same convexity).  What the reference still does with it: pushes to the root's far side, then get_closest_object from THAT position (:1806) -- the
                // box is behind it, so only the sphere can answer.  Where the ray's LINE misses the sphere by a margin a thousand times the rounding of the
                // discriminant, the answer is "nothing" from any point of it, every later outer iteration is a no-op, and the path ends with the L, beta, alpha,
                // depth and direction it has now: the pushes and both get_closest_object calls are skipped (WALK_DONE_CLEAR).  Anything less clear-cut pushes on.
                const f3 o = w.pos - ld3(P.sph_center);
                const float qa = dot(w.dir, w.dir), qb = 2.0f * dot(w.dir, o), oo = dot(o, o);
                const float disc = qb * qb - 4.0f * qa * (oo - P.sph_radius * P.sph_radius);
                if (disc < -1e-3f * (qb * qb + 4.0f * qa * oo)) {
                    // ... and sphere::intersect's OTHER way of answering: `B == 0` is a hit at distance 0 whatever the discriminant (find_discr, geometry.h:52-57).  B =
                    // 2 dir . (pos - centre) grows by 2 A per unit of path, so it can round to zero only next to the point of closest approach to the centre.  The
                    // position get_closest_object would start from lies where the ray leaves the root box, at most 0.1 (the pushes' minimum) + rounding beyond it: the
                    // rule cannot fire there when B is already positive by a margin here (closest approach behind), or still negative by a margin one unit past the
                    // root's far side (closest approach well ahead).  Anything in between pushes on.
                    const float bs = 2.0f * (fabsf(w.dir.x * o.x) + fabsf(w.dir.y * o.y) + fabsf(w.dir.z * o.z));
                    bool b_clear = qb > 1e-3f * bs;
                    if (!b_clear) {
                        float t_in, t_out;
                        box_intersect(K.root_lo, K.root_hi, w.pos, w.inv, t_in, t_out);
                        const float reach = 2.0f * qa * (t_out + 1.0f);
                        b_clear = qb + reach < -1e-3f * (bs + reach);
                    }
                    if (b_clear) { st = LOC_OUTSIDE; clear_exit = true; }
                }
            }
#endif
            if (st == LOC_EMPTY) {
                // empty node: push to its far side, at least 0.1 (:1613-1616)
                float t_min, t_max;
                box_intersect(nmin, nmax, w.pos, w.inv, t_min, t_max);
                t_max = fmax_(t_max, 0.1f);
                w.pos += w.dir * t_max;
                if (COUNT) c.n_skips++;
            }
        }
        // (round 6) ... and leaves the loop once fewer than VPT_SKIP_MIN lanes still stand in an empty node: a round costs the same ~80 instructions whatever
        // its lane count, and the last rounds of a pass used to run for the two or three lanes with the longest runs of pushes (the loop as a whole issued at
        // ~20 lanes).  The stragglers sit this pass's tracking step out and go on pushing in the next pass's loop, next to the lanes that need pushes then.
        // Per lane the operations and their order are unchanged.  Tracer -5.4 % on config 2, -3.2 % on 5, -0.5 / -0.9 % on 3 / 4 (profiles/r06_skip_min.txt).
#if VPT_SKIP_MIN > 1
        if ((int)__popcll(__ballot(st == LOC_EMPTY)) < VPT_SKIP_MIN) break;
#else
        if (!__any(st == LOC_EMPTY)) break;
#endif
    }
#ifdef VPT_PROFILE_SECTIONS
    if (!COUNT) c.n_skips += (uint32_t)(__builtin_readcyclecounter() - tp0_);      // cycles of the skip loop (perf study)
#endif
    if (st == LOC_EMPTY) return WALK_GOES_ON;    // still crossing empty nodes: next pass
    if (st == LOC_OUTSIDE) return (TRX && clear_exit) ? WALK_DONE_CLEAR : WALK_DONE;
    if (COUNT) {
        const unsigned long long m = __ballot(1);
        if (__lane_id() == __ffsll((long long)m) - 1) atomicAdd(&P.counters->sched[7], (unsigned long long)__popcll(m));
    }
    if (is_sample) {
        // :1647-1651
        float t_min, t_max, geo_dist;
        box_intersect(K.root_lo, K.root_hi, w.pos, w.inv, t_min, w.distance);
        if (sphere_intersect(P, w.pos, w.dir, geo_dist, t_max)) {
            w.distance = geo_dist;
            w.geo = true;
        }
    }
    int spins = 0;
    auto tally = [&](int which, unsigned long long n) {          // counting builds: one atomic per wave and event (Counters::retry)
        const unsigned long long m = __ballot(1);
        unsigned long long s = 0;
        for (unsigned long long mm = m; mm != 0ull; mm &= mm - 1ull) s += (unsigned long long)__shfl((int)n, __ffsll((long long)mm) - 1);
        if (__lane_id() == __ffsll((long long)m) - 1) atomicAdd(&P.counters->retry[which], s);
    };
    for (;;) {
        const float lg = det_logf(1 - rnd(rng, draws));
        if (COUNT) c.n_steps++;
        if (is_sample) w.t -= lg * K.inv_max * K.inv_dm;                          // :1652
        else if (is_emit) w.t -= lg * K.inv_max * P.tr_depth / P.extinction[0];   // :1331
        else w.t -= lg * K.sigma_r_inv * P.tr_depth;                              // :1231
        if (!is_emit && w.t >= w.distance) {
            if (is_sample && w.geo) w.obj2 = true;                                // :1654-1657
            if (use_retries && is_sample && retries > 0) {
                retries -= 1;                    // the next sample() call, same position: t = 0 again
                w.t = 0.0f;
                // a retry draws once and a step that goes on draws a second time: stay within the words
                // buffered since the pass's refill point (vpt_rng.h), and do not hold the wave up for long
                const uint32_t buffered = (rng.has_carry ? 1u : 0u) + (4u - rng.idx);
                if (buffered < 2u || ++spins >= VPT_RETRY_SPINS) {
                    if (COUNT) { tally(1, 1ull); tally(2, (unsigned long long)(spins > 0 ? spins : 1)); }
                    return WALK_GOES_ON;
                }
                continue;
            }
            if (COUNT && use_retries) { tally(3, 1ull); if (spins > 0) tally(2, (unsigned long long)spins); }
            return WALK_DONE;
        }
        break;
    }
    if (COUNT && use_retries) { tally(0, 1ull); if (spins > 0) tally(2, (unsigned long long)spins); }
    w.pos += w.dir * w.t;                                                     // cumulative t (Q-list 1)
    if (!contains(K.root_lo, K.root_hi, w.pos)) return WALK_DONE;
    if (SPLIT && !MULTI && !is_emit) {
        // request the texels, decide later (walk_finish)
        f3 u;
        DVolume v0;
        load_vol0(v0);
        const bool inside = to_unit(v0.m, v0, w.pos, u);
        if (COUNT) c.n_d++;
        if (COUNT) count_fetch(P, 0, inside);
        if (COLOR && COUNT && is_sample && P.vol0.has_color) c.n_c++;          // the reference looks the colour up here (:1662)
        pd.state = 1;
        if (inside) {
            const Taps t = make_taps(v0.dim, v0.dimf, u, COUNT ? P.tex_fixed8 : 0);
            const bool zero = footprint_is_zero<A24>(v0, t);     // (study builds: density known to be +0 -- state 1, like a point outside the domain)
#ifdef VPT_ZERO_MASK
            if (COUNT) count_fetch(P, 3, zero);
#endif
            if (!zero) {
                issue_f32<A24>(v0.density, v0, t, pd);
                pd.state = 2;
            }
        }
        if (COUNT) coherence_stats<A24>(P, w.pos);
        return WALK_PENDING;
    }
    float density = 0.0f;
    f3 Cd = COLOR ? mk3(0.0f) : mk3(1.0f);
    f3 em = mk3(0.0f);
    // sum_density / sum_emission over the instances of this leaf (:1003, :970).  The reference also
    // evaluates sum_color here at every step of sample() (:1662), but its value is only used on a real
    // collision (:1673): it is counted here and FETCHED there (8 float4 texels per instance -- most of
    // what the texture-data path returned per step in instanced scenes).
    // NOTE the reference looks up at the NEW position with the leaf found at the OLD one (:1659-1662); so does this, and
    // the sub-cell is the new position's inside that leaf's box (clamped: the point may have left the leaf)
    const bool refined = MULTI && P.single_file;
    const int cell = refined ? sub_cell(P, nmin, w.pos) : 0;
    if (refined && COUNT) {
        // the reference visits (and SURVEY 8d counts) every instance of the LEAF's list, whatever is skipped here
        const uint32_t n_leaf = P.leaf_offsets[leaf + 1] - P.leaf_offsets[leaf];
        if (!is_emit) c.n_d += n_leaf;
        if (COLOR && is_sample && P.vol0.has_color) c.n_c += n_leaf;
        if (EMIT && is_emit && P.vol0.has_emission) c.n_e += n_leaf;
        uint32_t z0 = 0, z1 = 0, z2 = 0;
        for_each_instance<MULTI>(P, leaf, cell, [&](const float* m, const DVolume& v) {
            lookup_volume<COLOR, EMIT, false, ELDS, A24, true>(P, m, v, w.pos, !is_emit, false, is_emit, density, Cd, em, z0, z1, z2, false);
        });
    } else {
        for_each_instance<MULTI>(P, leaf, cell, [&](const float* m, const DVolume& v) {
            lookup_volume<COLOR, EMIT, COUNT, ELDS, A24>(P, m, v, w.pos, !is_emit, false, is_emit, density, Cd, em, c.n_d, c.n_c, c.n_e, is_sample);
        });
    }
    if (COUNT && !MULTI && !is_emit) coherence_stats<A24>(P, w.pos);
    if (is_emit) {
        w.Ld += em;                                                           // :1335
        return WALK_GOES_ON;
    }
    return walk_decide<MULTI, COLOR, COUNT, A24, HS, HCAP>(P, K, is_sample, record_hist, hist, n_hist, w, rng, draws, density, leaf, cell) ? WALK_DONE : WALK_GOES_ON;
}

// second half of a split-phase step: the texels requested by walk_step have (had time to) arrive
template <bool COLOR, bool COUNT, bool A24, int HCAP = VPT_HIST_CAP>
VPT_D bool walk_finish(const TraceParams& P, const WalkConst& K, int kind, bool record_hist, float* hist, uint32_t& n_hist, Walk& w, Rng& rng,
                       uint32_t& draws, const Pending& pd) {
    const float density = pd.state == 2 ? lerp8(pd) : 0.0f;
    return walk_decide<false, COLOR, COUNT, A24, 256, HCAP>(P, K, kind == WALK_SAMPLE, record_hist, hist, n_hist, w, rng, draws, density, 0, 0);
}

// Tr prologue :1153-1167 (shared by sun / point-light / sky / sphere shadow rays): returns true
// when a walk is needed; otherwise w.trw holds the result (1: misses the box, 0: sphere in the way)
// tr_inv: 1 / tr_dir per component (the sun's is a launch constant, host-evaluated: three correctly rounded divisions per scatter event less)
VPT_D bool tr_begin(f3 sph_center, float sph_radius, const WalkConst& K, Walk& w, f3 from, f3 tr_dir, f3 tr_inv) {
    w.pos = from;
    w.dir = tr_dir;
    w.inv = tr_inv;
    float t_min, t_max;
    if (!contains(K.root_lo, K.root_hi, w.pos)) {
        if (box_intersect(K.root_lo, K.root_hi, w.pos, w.inv, t_min, t_max)) w.pos += w.dir * (t_min + VPT_EPS);
        else { w.trw = 1.0f; return false; }
    }
    float geo_dist;
    box_intersect(K.root_lo, K.root_hi, w.pos, w.inv, t_min, w.distance);
    if (sphere_intersect(sph_center, sph_radius, w.pos, w.dir, geo_dist, t_max)) { w.trw = 0.0f; return false; }   // :1160
    w.t = 0.0f;
    w.trw = 1.0f;
    return true;
}
VPT_D bool tr_begin(const TraceParams& P, const WalkConst& K, Walk& w, f3 from, f3 tr_dir) {
    return tr_begin(ld3(P.sph_center), P.sph_radius, K, w, from, tr_dir, rcp3(tr_dir));
}
// Tr epilogue :1166,:1267
VPT_D float tr_end(const WalkConst& K, const Walk& w) { return clampf(w.trw * expf(-K.sigma_c * w.distance), .0f, 1.0f); }

}  // namespace vpt
