// vpt_trace_common.h -- device helpers shared by the tracer kernels (vpt_trace.hip: direct_integrator,
// vpt_trace_vol.hip: vol_integrator): slab / sphere tests, octree point location, dense-grid
// look-ups, phase sampling, and ONE tracking step of any walk kind (delta tracking `sample`
// render_kernel.cu:1556, ratio tracking `Tr` :1138, emission march :1275).  Strict arithmetic
// (vpt_math.h): every translation unit including this file is built with -ffp-contract=off.
#pragma once

#include <hip/hip_runtime.h>

#include "vpt_device.h"
#include "vpt_rng.h"

namespace vpt {

// Path-level state that only the (infrequent) transition states touch lives in LDS, one column per
// thread ([field][thread]: conflict-free), so that the walk loop keeps fewer registers live and a
// third wave fits per SIMD.  The proxies make the integrator code read as if they were registers.
struct LdsF {
    float* p;
    VPT_D operator float() const { return *p; }
    VPT_D void operator=(float v) const { *p = v; }
};
struct LdsI {
    int* p;
    VPT_D operator int() const { return *p; }
    VPT_D void operator=(int v) const { *p = v; }
    VPT_D void operator++(int) const { *p += 1; }
    VPT_D void operator--(int) const { *p -= 1; }
};
template <int S>                                // S: floats between two fields of a column (threads of the block / rays of the pool)
struct LdsF3S {
    float* p;                                   // x at p[0], y at p[S], z at p[2 S]
    VPT_D operator f3() const { return mk3(p[0], p[S], p[2 * S]); }
    VPT_D void operator=(f3 v) const { p[0] = v.x; p[S] = v.y; p[2 * S] = v.z; }
    VPT_D void operator+=(f3 v) const { p[0] += v.x; p[S] += v.y; p[2 * S] += v.z; }
    VPT_D void operator*=(f3 v) const { p[0] *= v.x; p[S] *= v.y; p[2 * S] *= v.z; }
};
typedef LdsF3S<256> LdsF3;

#ifndef VPT_HIST_CAP
#define VPT_HIST_CAP 11
#endif

VPT_D f3 ld3(const float* p) { return mk3(p[0], p[1], p[2]); }

// AABB::Intersect (bvh/AABB.h:182-205) with the reciprocal direction cached per ray
VPT_D bool box_intersect(f3 pmin, f3 pmax, f3 o, f3 inv, float& tmin, float& tmax) {
    float t1 = (pmin.x - o.x) * inv.x;
    float t2 = (pmax.x - o.x) * inv.x;
    float t3 = (pmin.y - o.y) * inv.y;
    float t4 = (pmax.y - o.y) * inv.y;
    float t5 = (pmin.z - o.z) * inv.z;
    float t6 = (pmax.z - o.z) * inv.z;
    tmin = fmax_(fmax_(fmin_(t1, t2), fmin_(t3, t4)), fmin_(t5, t6));
    tmax = fmin_(fmin_(fmax_(t1, t2), fmax_(t3, t4)), fmax_(t5, t6));
    if (tmax <= 0.0f) return false;
    if (tmin > tmax) return false;
    if (tmin < 0) {
        tmin = tmax;
        if (tmin < 0) return false;
    }
    return true;
}
VPT_D f3 rcp3(f3 d) { return mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z); }

VPT_D bool contains(f3 pmin, f3 pmax, f3 p) {    // AABB.h:141-146
    return (p.x >= pmin.x && p.x <= pmax.x && p.y >= pmin.y && p.y <= pmax.y && p.z >= pmin.z && p.z <= pmax.z);
}

// sphere::intersect + find_discr (geometry/geometry.h:46-70,114-137)
VPT_D bool sphere_intersect(f3 sph_center, float sph_radius, f3 ray_pos, f3 ray_dir, float& t_min, float& t_max) {
    f3 orig = ray_pos - sph_center;
    float A = ray_dir.x * ray_dir.x + ray_dir.y * ray_dir.y + ray_dir.z * ray_dir.z;
    float B = 2 * (ray_dir.x * orig.x + ray_dir.y * orig.y + ray_dir.z * orig.z);
    float C = orig.x * orig.x + orig.y * orig.y + orig.z * orig.z - sph_radius * sph_radius;
    float x1, x2;
    if (B == 0) {
        if (A == 0) return false;
        x1 = 0;
        x2 = sqrtf(-C / A);
    } else {
        float discr = B * B - 4 * A * C;
        if (discr < 0) return false;
        float sq = sqrtf(discr);
        float q = (B < 0.f) ? -0.5f * (B - sq) : -0.5f * (B + sq);
        x1 = q / A;
        x2 = C / q;
    }
    t_min = x1;
    t_max = x2;
    if (t_min > t_max) {
        float tmp = t_max;
        t_max = t_min;
        t_min = tmp;
    }
    if (t_min < 0) {
        t_min = t_max;
        if (t_min < 0) return false;
    }
    return true;
}

VPT_D bool sphere_intersect(const TraceParams& P, f3 ray_pos, f3 ray_dir, float& t_min, float& t_max) {
    return sphere_intersect(ld3(P.sph_center), P.sph_radius, ray_pos, ray_dir, t_min, t_max);
}

// get_closest_object (render_kernel.cu:1118-1135): 0 none, 1 volume box, 2 sphere
VPT_D int closest_object(f3 root_lo, f3 root_hi, f3 sph_center, float sph_radius, f3 o, f3 d, f3 inv, float& t_min) {
    float tmin1 = VPT_M_INF, tmax1 = -VPT_M_INF, tmin2 = VPT_M_INF, tmax2 = -VPT_M_INF;
    bool i1 = box_intersect(root_lo, root_hi, o, inv, tmin1, tmax1);
    bool i2 = sphere_intersect(sph_center, sph_radius, o, d, tmin2, tmax2);
    if (i1 && !i2) { t_min = tmin1; return 1; }
    if (!i1 && i2) { t_min = tmin2; return 2; }
    if (i1 && i2) {
        if (tmin1 < tmin2) { t_min = tmin1; return 1; }
        if (tmin2 < tmin1) { t_min = tmin2; return 2; }
    }
    return 0;
}

VPT_D int closest_object(const TraceParams& P, f3 o, f3 d, f3 inv, float& t_min) {
    return closest_object(ld3(P.root_pmin), ld3(P.root_pmax), ld3(P.sph_center), P.sph_radius, o, d, inv, t_min);
}

// ---- COLD launch constants ---------------------------------------------------------------------------------------------------------
// What only the transition states read (sun, sphere, light list, depths, the output pointers).  As fields of the by-value kernel argument they are
// loaded in the prologue and compete for scalar registers with what the walk step reads on every pass: the register allocator kept 24 of them
// resident and spilled the world->index matrix of the look-up to VGPR lanes instead (50 v_readlane in the three inlined copies of locate / to_unit of
// config 2's kernel; profiles/r04_four_waves.txt (n)).  Read through a laundered pointer to the kernel-argument segment (constant address space:
// scalar loads, a few hundred ns per transition pass, hidden behind the other waves) they are live inside the transition pass only.
typedef const __attribute__((address_space(4))) TraceParams* KargPtr;
struct ColdConst {
    f3 sph_center, sph_color;
    float sph_radius, sph_roughness;
    f3 sun_color, sun_dir, sun_inv;
    float sun_mult, phase_g1, emission_scale;
    float sky_mult;
    int ray_depth, volume_depth, num_lights;
    const DPointLight* lights;
    uint32_t n_pixels, iter_begin, iter_stride;
    Record* records;
};
VPT_D ColdConst load_cold_const() {
    KargPtr k = (KargPtr)__builtin_amdgcn_kernarg_segment_ptr();       // the TraceParams is the kernel's only argument: offset 0 of the segment
    asm volatile("" : "+s"(k));                                        // (not hoisted out of the pass: that is the point)
    ColdConst c;
    c.sph_center = mk3(k->sph_center[0], k->sph_center[1], k->sph_center[2]);
    c.sph_color = mk3(k->sph_color[0], k->sph_color[1], k->sph_color[2]);
    c.sph_radius = k->sph_radius; c.sph_roughness = k->sph_roughness;
    c.sun_color = mk3(k->sun_color[0], k->sun_color[1], k->sun_color[2]);
    c.sun_dir = mk3(k->sun_dir[0], k->sun_dir[1], k->sun_dir[2]);
    c.sun_inv = mk3(k->sun_inv[0], k->sun_inv[1], k->sun_inv[2]);
    c.sky_mult = k->sky_mult;
    c.sun_mult = k->sun_mult; c.phase_g1 = k->phase_g1; c.emission_scale = k->emission_scale;
    c.ray_depth = k->ray_depth; c.volume_depth = k->volume_depth; c.num_lights = k->num_lights;
    c.lights = k->lights;
    c.n_pixels = k->n_pixels; c.iter_begin = k->iter_begin; c.iter_stride = k->iter_stride;
    c.records = k->records;
    return c;
}
// A queued ray's record as the refill unpacks it (the 64-byte layout raygen writes behind an open lens or without heads):
//   q0 = {origin, t_hit | adv.x}  q1 = {dir, obj word}  q2 = Philox block  q3 = {counter, word index, depth | adv.y, t_box | adv.z}
// -- loaded as such, or REBUILT from a compact 32-byte record + the sample's head + the camera origin (TraceParams::compact_rays): the same values, so
// everything behind this function is one code path.  Launch constants come through the laundered kernel-argument pointer (scalar loads at the refill,
// nothing carried through the loop).
VPT_D void split_slot(const TraceParams& P, uint32_t slot, uint32_t& kiter, uint32_t& pixel);
// `entry` is the queue entry: the sample's slot (kiter * n_pixels + pixel) -- or, with queue-ordered compact records (VPT_QREC), the record's own place, the slot then
// travels in the record.  Returns the sample's kiter / pixel with the record.
VPT_D void load_ray_record(const TraceParams& P, uint32_t entry, uint32_t& kiter, uint32_t& pixel, float4& q0, float4& q1, float4& q2, float4& q3) {
    KargPtr k = (KargPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(k));
    if (k->compact_rays) {
#if VPT_QREC
        const float4* src = k->rays32 + 2u * (size_t)entry;
        const float4 c0 = ld_stream(src), c1 = ld_stream(src + 1);
        split_slot(P, __float_as_uint(c1.w), kiter, pixel);
        const uint32_t iteration = P.iter_begin + kiter * P.iter_stride;
        const uint32_t word = __float_as_uint(c0.w);
        const uint32_t counter = iteration * 1024u + (word >> 17);
        uint32_t b0, b1, b2, b3;
        philox_block(counter, pixel, b0, b1, b2, b3);                        // (the block raygen stood in when get_ray's draws were done: re-generated, not carried)
        q0 = make_float4(k->cam.origin[0] + 0.0f, k->cam.origin[1] + 0.0f, k->cam.origin[2] + 0.0f, c0.x);
        q1 = make_float4(c1.x, c1.y, c1.z, __uint_as_float(word & 0x3fffu));
        q2 = make_float4(__uint_as_float(b0), __uint_as_float(b1), __uint_as_float(b2), __uint_as_float(b3));
        q3 = make_float4(__uint_as_float(counter), __uint_as_float((word >> 14) & 7u), c0.y, c0.z);
#else
        split_slot(P, entry, kiter, pixel);
        const uint32_t iteration = P.iter_begin + kiter * P.iter_stride;
        const float4* src = k->rays32 + 2u * (size_t)entry;
        const float4 c0 = ld_stream(src), c1 = ld_stream(src + 1), h = ld_stream(k->heads + entry);
        const uint32_t word = __float_as_uint(c0.w);
        q0 = make_float4(k->cam.origin[0] + 0.0f, k->cam.origin[1] + 0.0f, k->cam.origin[2] + 0.0f, c0.x);      // (raygen's `origin + offset` with the closed lens' offset of +0)
        q1 = make_float4(h.x, h.y, h.z, __uint_as_float(word & 0x3fffu));
        q2 = c1;
        q3 = make_float4(__uint_as_float(iteration * 1024u + (word >> 17)), __uint_as_float((word >> 14) & 7u), c0.y, c0.z);
#endif
    } else {
        split_slot(P, entry, kiter, pixel);
        const float4* src = reinterpret_cast<const float4*>(k->records + entry);
        q0 = ld_stream(src); q1 = ld_stream(src + 1); q2 = ld_stream(src + 2); q3 = ld_stream(src + 3);
    }
}
// TraceParams::resolve for one batch of finishing paths (PH_T_FINISH): 13 dwords from the kernel-argument segment
VPT_D ResolveInTracer load_resolve() {
    KargPtr k = (KargPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(k));
    ResolveInTracer r;
    r.sky_dome = k->resolve.sky_dome; r.heads = k->resolve.heads; r.td = k->resolve.td;
    r.queue2 = k->resolve.queue2; r.queue2_tail = k->resolve.queue2_tail;
    r.cam_origin[0] = k->resolve.cam_origin[0]; r.cam_origin[1] = k->resolve.cam_origin[1]; r.cam_origin[2] = k->resolve.cam_origin[2];
    r.lens = k->resolve.lens; r.sky_view = k->resolve.sky_view; r.earth_bottom = k->resolve.earth_bottom;
    r.sun_dir[0] = k->resolve.sun_dir[0]; r.sun_dir[1] = k->resolve.sun_dir[1]; r.sun_dir[2] = k->resolve.sun_dir[2];
    return r;
}

// The single-volume descriptor (TraceParams::vol0) for ONE look-up: ~35 dwords that would otherwise want scalar registers through the whole launch next
// to the octree's and the walk's constants -- where they all fit the allocator re-reads the descriptor per look-up anyway, where they do not it parks
// scalars in VGPR lanes (six of the twenty timed instantiations did: 330-470 v_readlane / v_writelane each, tests/test_kernel_resources.py).  Read here,
// at the look-up, from the kernel-argument segment: the fields a look-up uses arrive in three or four scalar loads.
VPT_D void load_vol0(DVolume& v) {
    KargPtr k = (KargPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(k));
    v.density = k->vol0.density; v.emission = k->vol0.emission; v.color = k->vol0.color;
#pragma unroll
    for (int i = 0; i < 12; ++i) v.m[i] = k->vol0.m[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        v.bmin[i] = k->vol0.bmin[i]; v.fdim[i] = k->vol0.fdim[i]; v.dim[i] = k->vol0.dim[i]; v.edim[i] = k->vol0.edim[i]; v.cdim[i] = k->vol0.cdim[i];
        v.rdim[i] = k->vol0.rdim[i]; v.dimf[i] = k->vol0.dimf[i]; v.edimf[i] = k->vol0.edimf[i]; v.cdimf[i] = k->vol0.cdimf[i];
    }
    v.has_color = k->vol0.has_color; v.has_emission = k->vol0.has_emission; v.layout = k->vol0.layout; v.bdim[0] = k->vol0.bdim[0]; v.bdim[1] = k->vol0.bdim[1];
    v.elayout = k->vol0.elayout; v.addr24 = k->vol0.addr24; v.fast_div = k->vol0.fast_div;
#ifdef VPT_ZERO_MASK
    v.zmask = k->vol0.zmask; v.zshift = k->vol0.zshift; v.znwx = k->vol0.znwx; v.znby = k->vol0.znby;
#endif
}

// Three-level point location (get_quadrant x3, render_kernel.cu:1102-1115 + :1193-1227).
// Child boxes follow divide_bbox (bvh_kernels.cu:150-202): child i covers
//   x: low half for i in {0,2,4,6}, high half otherwise
//   y: HIGH half for i in {0,1,4,5}, low half otherwise
//   z: low half for i < 4, high half otherwise
// and the first child (index order) whose CLOSED box contains p wins.
enum { LOC_LEAF = 0, LOC_EMPTY = 1, LOC_OUTSIDE = 2 };
struct OccTop { uint32_t w0, w1, w2; };          // occupancy words of octree levels 1 and 2 (occ[0..2])
VPT_D int locate(const TraceParams& P, const uint32_t* occ, const OccTop& o, f3 p, f3& nmin, f3& nmax, int& leaf) {
    f3 lo = ld3(P.root_pmin), hi = ld3(P.root_pmax);
    // The eight children tile their parent's closed box, so "the first child whose closed box contains p" needs a
    // containment test only once, against the root (the union of its children); below that, one comparison per axis
    // and level picks the half.  A point ON a splitting plane lies in both halves and the lower child index wins:
    // the low half in x and z, the HIGH half in y (children 0,1,4,5).  NaN positions fail the root test.
    if (!(p.x >= lo.x && p.x <= hi.x && p.y >= lo.y && p.y <= hi.y && p.z >= lo.z && p.z <= hi.z)) return LOC_OUTSIDE;
    // One volume whose bounds touch every leaf (a dense grid: configs 3 and 4): no node is empty and the single-volume
    // look-up does not use the leaf index, so "inside the root" is the whole answer.
    if (P.octree_full_single) {
        leaf = 0;
        nmin = lo;
        nmax = hi;
        return LOC_LEAF;
    }
    int path = 0;
#pragma unroll
    for (int level = 0; level < 3; ++level) {
        const float hx = level == 0 ? P.root_mid[0] : (lo.x + hi.x) * 0.5f;
        const float hy = level == 0 ? P.root_mid[1] : (lo.y + hi.y) * 0.5f;
        const float hz = level == 0 ? P.root_mid[2] : (lo.z + hi.z) * 0.5f;
        const bool xh = !(p.x <= hx);
        const bool yh = p.y >= hy;
        const bool zh = !(p.z <= hz);
        const int c = (xh ? 1 : 0) | (yh ? 0 : 2) | (zh ? 4 : 0);
        lo.x = xh ? hx : lo.x; hi.x = xh ? hi.x : hx;
        lo.y = yh ? hy : lo.y; hi.y = yh ? hi.y : hy;
        lo.z = zh ? hz : lo.z; hi.z = zh ? hi.z : hz;
        path = path * 8 + c;
        // occupancy bit of the node: level 1 and 2 words (8 + 64 bits) come in registers, level 3 (512 bits) from LDS
        uint32_t word;
        if (level == 0) word = o.w0;
        else if (level == 1) word = path < 32 ? o.w1 : o.w2;
        else word = occ[(96 + path) >> 5];
        if (((word >> (path & 31)) & 1u) == 0) {
            nmin = lo;
            nmax = hi;
            return LOC_EMPTY;
        }
    }
    leaf = path;
    nmin = lo;                 // the leaf's own box (the instance loop refines its candidate list per sub-cell of it)
    nmax = hi;
    return LOC_LEAF;
}

// world -> normalised texture coordinates (render_kernel.cu:987-997)
// m: the instance's 3x4 world->index matrix (v.m, or the compact copy of it in TraceParams::insts)
VPT_D bool to_unit(const float* m, const DVolume& v, f3 p, f3& u) {
    f3 q;
    q.x = m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3];
    q.y = m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7];
    q.z = m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11];
    q.x = q.x - v.bmin[0];
    q.y = q.y - v.bmin[1];
    q.z = q.z - v.bmin[2];
    // u = q / fdim, correctly rounded (:996).  The extents are constants of the volume: where the host has checked them (vpt_fastdiv.h)
    // a multiplication by the rounded reciprocal and one exact-residual correction give the same bits as the division, as long as no
    // intermediate is subnormal or overflows -- |q| in [2^-40, 2^40] (extents <= 2^16).  A position within 1e-12 voxels of a bmin
    // plane, a non-finite coordinate or an unchecked extent takes the division itself; the wave decides together.
    const float qlo = fmin_(fmin_(fabsf(q.x), fabsf(q.y)), fabsf(q.z)), qhi = fmax_(fmax_(fabsf(q.x), fabsf(q.y)), fabsf(q.z));
    const bool fast = v.fast_div != 0 && qlo >= 0x1p-40f && qhi <= 0x1p+40f;
    if (__all(fast)) {
        const float yx = q.x * v.rdim[0], yy = q.y * v.rdim[1], yz = q.z * v.rdim[2];
        u.x = __builtin_fmaf(__builtin_fmaf(-v.fdim[0], yx, q.x), v.rdim[0], yx);
        u.y = __builtin_fmaf(__builtin_fmaf(-v.fdim[1], yy, q.y), v.rdim[1], yy);
        u.z = __builtin_fmaf(__builtin_fmaf(-v.fdim[2], yz, q.z), v.rdim[2], yz);
    } else {
        u.x = q.x / v.fdim[0];
        u.y = q.y / v.fdim[1];
        u.z = q.z / v.fdim[2];
    }
    return !(u.x < .0f || u.y < .0f || u.z < .0f || u.x > 1.0f || u.y > 1.0f || u.z > 1.0f);
}

// slot = kiter * n_pixels + pixel with kiter < 64 and slot < 2^31 (record chunks are capped at 16 GiB): the quotient
// is estimated in fp32 (off by at most one) and corrected exactly, instead of a 40-instruction u32 division
VPT_D void split_slot(const TraceParams& P, uint32_t slot, uint32_t& kiter, uint32_t& pixel) {
    uint32_t k = (uint32_t)((float)slot * P.inv_n_pixels);
    int r = (int)(slot - k * P.n_pixels);
    if (r < 0) { k -= 1u; r += (int)P.n_pixels; }
    else if (r >= (int)P.n_pixels) { k += 1u; r -= (int)P.n_pixels; }
    kiter = k;
    pixel = (uint32_t)r;
}

// ---- work distribution ---------------------------------------------------------------------------------
// The compacted ray queue is in tile-major order (raygen_kernel); a wave claims VPT_CHUNK consecutive
// entries with one leader atomic on a single global cursor, so the rays in flight across the whole GPU
// always come from a few neighbouring tiles.  P.chunk = VPT_CHUNK for batches; launches that give a wave fewer than ~6000
// samples (the per-frame call: 600 k rays for 3072 waves) claim half as many at a time, so that every wave gets work and
// the last chunks are shorter (per-frame tracer 0.215 -> 0.17 ms on config 2; profiles/r03_small_launch.txt).  (Partitioning the queue per XCD -- blockIdx % 8, own L2
// -- with stealing was measured: load imbalance between image regions cost more than the L2 locality
// gained, +4..6 % tracer time on configs 2-4.)
#ifndef VPT_CLAIM_COUNTERS
#define VPT_CLAIM_COUNTERS 8
#endif
// NC cursors, 128 bytes apart, that hand out the chunks INTERLEAVED (round 6) -- cursor c owns chunks c, c + NC, c + 2 NC, ... -- so every cursor sweeps the whole
// queue at 1 / NC of the rate (no imbalance between image regions, unlike a partition); a workgroup starts at cursor blockIdx % NC and moves on to the next ones when its
// own is past the end.  ONE cursor takes ~27 atomics per microsecond from the 4096 waves of a config-2 launch and answers a claim in ~9 us (7.5 % of the tracer's wave
// cycles, profiles/r06_sections_c2_before_cursors.txt), and the first claims of a launch queue up behind each other for ~150 us: with 8 cursors config 2's tracer takes 3.72 -> 3.63 ms per
// 64 iterations and 0.71 -> 0.53 ms per 8 (profiles/r06_claim_cursors.txt).  NC = 1 (the single cursor) for the instantiations whose claims are rare anyway -- instanced
// scenes, the vol tracer: there the loop's scalar registers cost more than the cursors return (+1.5 % / +3 %).  Which wave traces which ray never mattered: results cannot move.
template <int NC>
VPT_D void claim_chunk(const TraceParams& P, uint32_t total, int lane, int leader, uint32_t& chunk_next, uint32_t& chunk_end, bool& more) {
    if (NC > 1) {
        uint32_t base = total;
        if (lane == leader) {
            const uint32_t c0 = blockIdx.x & (uint32_t)(NC - 1);
            // (32-bit arithmetic: total < 2^31 and a cursor overshoots the end by at most one chunk index per workgroup and attempt)
#pragma unroll 1
            for (uint32_t a = 0; a < (uint32_t)NC; ++a) {
                const uint32_t c = (c0 + a) & (uint32_t)(NC - 1);
                const uint32_t k = atomicAdd(P.work_counter + 32u + 32u * c, 1u);
                const uint32_t b = (k * (uint32_t)NC + c) * P.chunk;
                if (b < total) { base = b; break; }
            }
        }
        base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
        chunk_next = min(base, total);
        chunk_end = base < total ? min(base + P.chunk, total) : total;
        if (base >= total) more = false;
    } else {
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(P.work_counter, P.chunk);
        base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);     // wave-uniform: the claim lives in scalar registers
        chunk_next = min(base, total);
        chunk_end = min(base + P.chunk, total);
        if (chunk_end == total) more = false;
    }
}

// Grid pointers read from an instance descriptor in memory are generic pointers to the compiler, which
// then emits FLAT loads (slower path, ties up both memory counters); they are global: say so.
typedef const __attribute__((address_space(1))) float* gptr_f;
typedef float __attribute__((ext_vector_type(4))) v4f;
typedef const __attribute__((address_space(1))) v4f* gptr_f4;
VPT_D f4 ld_g4(gptr_f4 g, uint32_t i) { const v4f v = g[i]; return mk4(v.x, v.y, v.z, v.w); }

// Texel index arithmetic.  32-bit integer multiplies issue at a quarter of the rate of the 24-bit ones on CDNA; the
// host sets DVolume::addr24 when every product below has operands under 2^24 and a result under 2^32 (true for any
// grid whose y*z slice count and x extent are below 16.7 M: config 4's 1024x704x1216 included); when that holds for
// every volume of the scene the A24 instantiation of the tracer is launched, else the 32-bit one.
template <bool A24> VPT_D uint32_t imul(uint32_t a, uint32_t b) { return A24 ? __umul24(a, b) : a * b; }

struct Taps {
    int i0, i1, j0, j1, k0, k1;
    int ir, jr, kr;       // floor of the texel coordinates before clamping (GRID_QUADS rows; the zero-footprint mask's block)
    float ax, ay, az;
};
// fixed8 (TraceParams::tex_fixed8, VPT_TEX_WEIGHTS=fixed8): a DIAGNOSTIC model of the CUDA texture unit, which holds the interpolation weights in
// 9-bit fixed point with 8 fractional bits (CUDA C Programming Guide, "Linear Filtering"); the reference's tex3D calls (render_kernel.cu:999-1014)
// run on that hardware, the parity contract here is full binary32 weights.  Model: weight = floor(a * 256 + 0.5) / 256, same lerps.
// Only the COUNTING instantiations of the tracers carry the switch (the host runs them when it is set): a launch-uniform scalar branch per look-up
// moved the timed instantiation's schedule by 1.6 % on config 2 (profiles/r04_four_waves.txt (m)), and a diagnostic must cost the product nothing.
VPT_D float quant8(float a) { return floorf(a * 256.0f + 0.5f) * 0.00390625f; }
VPT_D Taps make_taps(const int* dim, const float* dimf, f3 u, int fixed8) {      // dimf[a] = (float)dim[a] (DVolume)
    Taps t;
    float xb = u.x * dimf[0] - 0.5f;
    float yb = u.y * dimf[1] - 0.5f;
    float zb = u.z * dimf[2] - 0.5f;
    float fx = floorf(xb), fy = floorf(yb), fz = floorf(zb);
    t.ax = xb - fx;
    t.ay = yb - fy;
    t.az = zb - fz;
    if (fixed8) { t.ax = quant8(t.ax); t.ay = quant8(t.ay); t.az = quant8(t.az); }      // launch-uniform: a scalar branch
    int i = (int)fx, j = (int)fy, k = (int)fz;
    t.i0 = min(max(i, 0), dim[0] - 1);
    t.i1 = min(max(i + 1, 0), dim[0] - 1);
    t.j0 = min(max(j, 0), dim[1] - 1);
    t.j1 = min(max(j + 1, 0), dim[1] - 1);
    t.k0 = min(max(k, 0), dim[2] - 1);
    t.k1 = min(max(k + 1, 0), dim[2] - 1);
    t.ir = i;
    t.jr = j;
    t.kr = k;
    return t;
}
// trilinear f32 fetch: CUDA "linear, normalised, clamp" addressing, fp32 weights, nested
// lerp x -> y -> z with lerp(a,b,t) = a + t*(b-a)
template <bool A24>
VPT_D float fetch_f32(const float* __restrict__ g_, const int* dim, const Taps& t) {
    const gptr_f g = (gptr_f)g_;
    const uint32_t dx = (uint32_t)dim[0];
    const uint32_t s0 = imul<A24>((uint32_t)t.k0, (uint32_t)dim[1]), s1 = imul<A24>((uint32_t)t.k1, (uint32_t)dim[1]);
    const uint32_t r00 = imul<A24>(s0 + (uint32_t)t.j0, dx);
    const uint32_t r10 = imul<A24>(s0 + (uint32_t)t.j1, dx);
    const uint32_t r01 = imul<A24>(s1 + (uint32_t)t.j0, dx);
    const uint32_t r11 = imul<A24>(s1 + (uint32_t)t.j1, dx);
    const float c000 = g[r00 + t.i0], c100 = g[r00 + t.i1];
    const float c010 = g[r10 + t.i0], c110 = g[r10 + t.i1];
    const float c001 = g[r01 + t.i0], c101 = g[r01 + t.i1];
    const float c011 = g[r11 + t.i0], c111 = g[r11 + t.i1];
    const float c00 = c000 + (c100 - c000) * t.ax;
    const float c10 = c010 + (c110 - c010) * t.ax;
    const float c01 = c001 + (c101 - c001) * t.ax;
    const float c11 = c011 + (c111 - c011) * t.ax;
    const float c0 = c00 + (c10 - c00) * t.ay;
    const float c1 = c01 + (c11 - c01) * t.ay;
    return c0 + (c1 - c0) * t.az;
}
// same trilinear fetch from the bricked layout (vpt_device.h): texel (i, j, k) lives at
//   (((k>>2) * by + (j>>2)) * bx + (i>>2)) * 64 + (k&3) * 16 + (j&3) * 4 + (i&3)
template <bool A24>
VPT_D float fetch_f32_bricked(const float* __restrict__ g_, const DVolume& v, const Taps& t) {
    const gptr_f g = (gptr_f)g_;
    const uint32_t row = (uint32_t)v.bdim[0] * 64u, slab = (uint32_t)v.bdim[1] * row;       // floats per brick row / brick slab
    const uint32_t x0 = (((uint32_t)t.i0 >> 2) << 6) + ((uint32_t)t.i0 & 3u), x1 = (((uint32_t)t.i1 >> 2) << 6) + ((uint32_t)t.i1 & 3u);
    const uint32_t y0 = imul<A24>((uint32_t)t.j0 >> 2, row) + (((uint32_t)t.j0 & 3u) << 2), y1 = imul<A24>((uint32_t)t.j1 >> 2, row) + (((uint32_t)t.j1 & 3u) << 2);
    const uint32_t z0 = imul<A24>((uint32_t)t.k0 >> 2, slab) + (((uint32_t)t.k0 & 3u) << 4), z1 = imul<A24>((uint32_t)t.k1 >> 2, slab) + (((uint32_t)t.k1 & 3u) << 4);
    const float c000 = g[z0 + y0 + x0], c100 = g[z0 + y0 + x1];
    const float c010 = g[z0 + y1 + x0], c110 = g[z0 + y1 + x1];
    const float c001 = g[z1 + y0 + x0], c101 = g[z1 + y0 + x1];
    const float c011 = g[z1 + y1 + x0], c111 = g[z1 + y1 + x1];
    const float c00 = c000 + (c100 - c000) * t.ax;
    const float c10 = c010 + (c110 - c010) * t.ax;
    const float c01 = c001 + (c101 - c001) * t.ax;
    const float c11 = c011 + (c111 - c011) * t.ax;
    const float c0 = c00 + (c10 - c00) * t.ay;
    const float c1 = c01 + (c11 - c01) * t.ay;
    return c0 + (c1 - c0) * t.az;
}
// GRID_QUADS (vpt_device.h): the footprint is the two float4 entries (i0, i1) of row (kr+1, jr+1); same texels, same lerp order.
// The entry index stays below 2^32 (host check), the byte offset does not: 64-bit addressing.
template <bool A24>
VPT_D void quad_entries(const int* dim, const Taps& t, uint32_t& e0, uint32_t& e1) {
    const uint32_t jc = (uint32_t)min(max(t.jr + 1, 0), dim[1]), kc = (uint32_t)min(max(t.kr + 1, 0), dim[2]);
    const uint32_t row = imul<A24>(imul<A24>(kc, (uint32_t)dim[1] + 1u) + jc, (uint32_t)dim[0]);
    e0 = row + (uint32_t)t.i0;
    e1 = row + (uint32_t)t.i1;
}
// DVolume::zmask: is every texel of this footprint exactly zero?  One word of an L2-resident bit-set (<= 512 KB) instead of two 128-byte lines from HBM.
// STUDY BUILDS ONLY (-DVPT_ZERO_MASK, `build.py --variant zmask -DVPT_ZERO_MASK`; at run time VPT_ZERO_MASK=1): exact, and slower on every config -- the mask word
// is one more DEPENDENT load ahead of the quads (EXPERIMENTS.md round 6, profiles/r06_zero_mask.txt); the product library does not carry the test.
typedef const __attribute__((address_space(1))) uint32_t* gptr_u;
template <bool A24>
VPT_D bool footprint_is_zero(const DVolume& v, const Taps& t) {
#ifdef VPT_ZERO_MASK
    if (v.zmask == nullptr) return false;              // (launch-uniform: a scalar branch)
    const uint32_t bx = (uint32_t)(t.ir + 1) >> v.zshift, by = (uint32_t)(t.jr + 1) >> v.zshift, bz = (uint32_t)(t.kr + 1) >> v.zshift;
    const uint32_t w = imul<A24>(imul<A24>(bz, (uint32_t)v.znby) + by, (uint32_t)v.znwx) + (bx >> 5);
    return ((((gptr_u)v.zmask)[w] >> (bx & 31u)) & 1u) != 0u;
#else
    return false;
#endif
}
// a corner quad is read once per look-up and (on the grids that are re-laid: those that do not stay in L2) almost never again by the same CU --
// 1.02 lanes of a wave share an 8^3 brick on config 4 (profiles/r02_lookup_coherence.txt).  -DVPT_NT_QUADS marks the two loads non-temporal, so
// that their 128-byte lines are the first to leave L2 / the Infinity Cache instead of the records, the HDRI and the dome (measured: profiles/r05_*).
VPT_D v4f ld_quad(gptr_f4 g, size_t e) {
#ifdef VPT_NT_QUADS
    return __builtin_nontemporal_load(g + e);
#else
    return g[e];
#endif
}
template <bool A24>
VPT_D float fetch_f32_quads(const float* __restrict__ g_, const int* dim, const Taps& t) {
    const gptr_f4 g = (gptr_f4)g_;
    uint32_t e0, e1;
    quad_entries<A24>(dim, t, e0, e1);
    const v4f q0 = ld_quad(g, (size_t)e0), q1 = ld_quad(g, (size_t)e1);
    const float c00 = q0.x + (q1.x - q0.x) * t.ax;
    const float c10 = q0.y + (q1.y - q0.y) * t.ax;
    const float c01 = q0.z + (q1.z - q0.z) * t.ax;
    const float c11 = q0.w + (q1.w - q0.w) * t.ax;
    const float c0 = c00 + (c10 - c00) * t.ay;
    const float c1 = c01 + (c11 - c01) * t.ay;
    return c0 + (c1 - c0) * t.az;
}
// The same fetch in two halves (split-phase look-up, vpt_walk.h): `issue` requests the eight texels, `lerp8` interpolates them with
// the operation order of fetch_f32 -- whatever runs between the two overlaps the memory latency.
struct Pending {
    float c[8];           // c000 c100 c010 c110 c001 c101 c011 c111
    float ax, ay, az;
    int state;            // 0 none, 1 density known to be 0 (point outside the grid's domain), 2 texels in flight
};
template <bool A24>
VPT_D void issue_f32(const float* __restrict__ g_, const DVolume& v, const Taps& t, Pending& pd) {
    const gptr_f g = (gptr_f)g_;
    if (v.layout == GRID_QUADS) {
        uint32_t e0, e1;
        quad_entries<A24>(v.dim, t, e0, e1);
        const v4f q0 = ld_quad((gptr_f4)g_, (size_t)e0), q1 = ld_quad((gptr_f4)g_, (size_t)e1);
        pd.c[0] = q0.x; pd.c[1] = q1.x;
        pd.c[2] = q0.y; pd.c[3] = q1.y;
        pd.c[4] = q0.z; pd.c[5] = q1.z;
        pd.c[6] = q0.w; pd.c[7] = q1.w;
    } else if (v.layout == GRID_BRICKS) {
        const uint32_t row = (uint32_t)v.bdim[0] * 64u, slab = (uint32_t)v.bdim[1] * row;
        const uint32_t x0 = (((uint32_t)t.i0 >> 2) << 6) + ((uint32_t)t.i0 & 3u), x1 = (((uint32_t)t.i1 >> 2) << 6) + ((uint32_t)t.i1 & 3u);
        const uint32_t y0 = imul<A24>((uint32_t)t.j0 >> 2, row) + (((uint32_t)t.j0 & 3u) << 2), y1 = imul<A24>((uint32_t)t.j1 >> 2, row) + (((uint32_t)t.j1 & 3u) << 2);
        const uint32_t z0 = imul<A24>((uint32_t)t.k0 >> 2, slab) + (((uint32_t)t.k0 & 3u) << 4), z1 = imul<A24>((uint32_t)t.k1 >> 2, slab) + (((uint32_t)t.k1 & 3u) << 4);
        pd.c[0] = g[z0 + y0 + x0]; pd.c[1] = g[z0 + y0 + x1];
        pd.c[2] = g[z0 + y1 + x0]; pd.c[3] = g[z0 + y1 + x1];
        pd.c[4] = g[z1 + y0 + x0]; pd.c[5] = g[z1 + y0 + x1];
        pd.c[6] = g[z1 + y1 + x0]; pd.c[7] = g[z1 + y1 + x1];
    } else {
        const uint32_t dx = (uint32_t)v.dim[0];
        const uint32_t s0 = imul<A24>((uint32_t)t.k0, (uint32_t)v.dim[1]), s1 = imul<A24>((uint32_t)t.k1, (uint32_t)v.dim[1]);
        const uint32_t r00 = imul<A24>(s0 + (uint32_t)t.j0, dx), r10 = imul<A24>(s0 + (uint32_t)t.j1, dx);
        const uint32_t r01 = imul<A24>(s1 + (uint32_t)t.j0, dx), r11 = imul<A24>(s1 + (uint32_t)t.j1, dx);
        pd.c[0] = g[r00 + t.i0]; pd.c[1] = g[r00 + t.i1];
        pd.c[2] = g[r10 + t.i0]; pd.c[3] = g[r10 + t.i1];
        pd.c[4] = g[r01 + t.i0]; pd.c[5] = g[r01 + t.i1];
        pd.c[6] = g[r11 + t.i0]; pd.c[7] = g[r11 + t.i1];
    }
    pd.ax = t.ax; pd.ay = t.ay; pd.az = t.az;
}
VPT_D float lerp8(const Pending& pd) {
    const float c00 = pd.c[0] + (pd.c[1] - pd.c[0]) * pd.ax;
    const float c10 = pd.c[2] + (pd.c[3] - pd.c[2]) * pd.ax;
    const float c01 = pd.c[4] + (pd.c[5] - pd.c[4]) * pd.ax;
    const float c11 = pd.c[6] + (pd.c[7] - pd.c[6]) * pd.ax;
    const float c0 = c00 + (c10 - c00) * pd.ay;
    const float c1 = c01 + (c11 - c01) * pd.ay;
    return c0 + (c1 - c0) * pd.az;
}
VPT_D f4 lerp4(f4 a, f4 b, float t) { return a + (b - a) * t; }
template <bool A24>
VPT_D f3 fetch_f4(const f4* __restrict__ g_, const int* dim, const Taps& t) {
    const gptr_f4 g = (gptr_f4)g_;
    const uint32_t dx = (uint32_t)dim[0];
    const uint32_t s0 = imul<A24>((uint32_t)t.k0, (uint32_t)dim[1]), s1 = imul<A24>((uint32_t)t.k1, (uint32_t)dim[1]);
    const uint32_t r00 = imul<A24>(s0 + (uint32_t)t.j0, dx);
    const uint32_t r10 = imul<A24>(s0 + (uint32_t)t.j1, dx);
    const uint32_t r01 = imul<A24>(s1 + (uint32_t)t.j0, dx);
    const uint32_t r11 = imul<A24>(s1 + (uint32_t)t.j1, dx);
    const f4 c00 = lerp4(ld_g4(g, r00 + t.i0), ld_g4(g, r00 + t.i1), t.ax);
    const f4 c10 = lerp4(ld_g4(g, r10 + t.i0), ld_g4(g, r10 + t.i1), t.ax);
    const f4 c01 = lerp4(ld_g4(g, r01 + t.i0), ld_g4(g, r01 + t.i1), t.ax);
    const f4 c11 = lerp4(ld_g4(g, r11 + t.i0), ld_g4(g, r11 + t.i1), t.ax);
    return xyz(lerp4(lerp4(c00, c10, t.ay), lerp4(c01, c11, t.ay), t.az));
}

// The 256-entry blackbody table (kernel_params.emission_texture) of the emission march, copied to LDS by the kernels
// that can emit and have the LDS to spare (template ELDS: the direct tracer): its three dwords are read right after the emission texel they depend on, at every
// step of estimate_emission -- from LDS that is one short round trip instead of a second global one.
static __shared__ float s_emission_lut[768];
VPT_D void stage_emission_lut(const TraceParams& P) {
    if (P.emission_lut)
        for (uint32_t i = threadIdx.x; i < 768u; i += blockDim.x) s_emission_lut[i] = P.emission_lut[i];
}

// one volume's contribution at world position p (get_density / get_color / get_emission)
// FC (counting builds): tally the trilinear fetches actually ISSUED (point inside the instance's domain, value wanted),
// wave-aggregated, into Counters::fetches -- the bytes the kernel really has to move, next to the reference-defined
// look-up counts n_d / n_c / n_e
VPT_D void count_fetch(const TraceParams& P, int which, bool issued) {
    const unsigned long long m = __ballot(issued);
    if (m != 0ull && __lane_id() == __ffsll((long long)m) - 1) atomicAdd(&P.counters->fetches[which], (unsigned long long)__popcll(m));
}
template <bool COLOR, bool EMIT, bool COUNT, bool ELDS, bool A24, bool FC = COUNT>
VPT_D void lookup_volume(const TraceParams& P, const float* m, const DVolume& v, f3 p, bool want_density, bool want_color, bool want_emission,
                         float& density, f3& color, f3& emission, uint32_t& n_d, uint32_t& n_c, uint32_t& n_e, bool count_color = false) {
    f3 u;
    const bool inside = to_unit(m, v, p, u);
    // every texture object has its own extent (the reference densifies each grid over its own
    // active bbox, gpu_vdb.cpp:179,262,343) but is addressed with the density grid's normalised
    // coordinates
    if (want_density) {
        if (COUNT) n_d++;
        if (FC) count_fetch(P, 0, inside);
        if (inside) {
            const Taps t = make_taps(v.dim, v.dimf, u, (COUNT || FC) ? P.tex_fixed8 : 0);
            // (v.zmask is launch-uniform: a scalar branch; a masked footprint adds +0 to a sum that started at +0 or holds a finite value: nothing to do)
            const bool zero = footprint_is_zero<A24>(v, t);
#ifdef VPT_ZERO_MASK
            if (FC) count_fetch(P, 3, zero);
#endif
            if (!zero)
                density += v.layout == GRID_QUADS    ? fetch_f32_quads<A24>(v.density, v.dim, t)
                           : v.layout == GRID_BRICKS ? fetch_f32_bricked<A24>(v.density, v, t)
                                                     : fetch_f32<A24>(v.density, v.dim, t);
        }
    }
    if (COLOR && COUNT && count_color && v.has_color) n_c++;      // the reference looks the colour up here (see walk_step)
    if (COLOR && want_color) {
        if (!v.has_color) {
            color = fmax3(color, mk3(1.0f));
        } else {
            if (COUNT) n_c++;
            if (FC) count_fetch(P, 1, inside);
            f3 c = inside ? fetch_f4<A24>(v.color, v.cdim, make_taps(v.cdim, v.cdimf, u, (COUNT || FC) ? P.tex_fixed8 : 0)) : mk3(0.0f);
            color = fmax3(color, c);
        }
    }
    if (EMIT && want_emission) {
        if (v.has_emission) {
            if (COUNT) n_e++;
            if (FC) count_fetch(P, 2, inside);
            if (inside) {
                const Taps t = make_taps(v.edim, v.edimf, u, (COUNT || FC) ? P.tex_fixed8 : 0);
                float index = v.elayout == GRID_QUADS ? fetch_f32_quads<A24>(v.emission, v.edim, t) : fetch_f32<A24>(v.emission, v.edim, t);
                index = clampf(index * 255.0f / P.emission_pivot, .0f, 255.0f);
                const int e = 3 * (int)index;
                if (ELDS) emission += mk3(s_emission_lut[e], s_emission_lut[e + 1], s_emission_lut[e + 2]) * P.emission_scale;
                else emission += mk3(P.emission_lut[e], P.emission_lut[e + 1], P.emission_lut[e + 2]) * P.emission_scale;
            }
        }
    }
}

// coordinate_system :92-102, spherical_direction :104-115, sample_hg :306-325 (2 draws)
VPT_D void sample_hg(f3& wo, Rng& rng, uint32_t& draws, float g) {
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
}

VPT_D float henyey_greenstein(float cos_theta, float g) {   // light.h:55-64 (pi/4 normalisation kept)
    float denominator = 1 + g * g - 2 * g * cos_theta;
    return VPT_PI_4 * (1 - g * g) / (denominator * sqrtf(denominator));
}

// vanDerCorput (camera.h:49-62): n = int(rand*100) is in [0, 100], so the radical inverse is read
// from a 101-entry table that the host fills with the reference's own float loop (vdc_table in
// vpt_host.hip) -- same bits, no data-dependent loop on the device.
VPT_D float van_der_corput(const float* table, Rng& rng, uint32_t key, uint32_t& draws) {
    int n = (int)(rnd_simple(rng, key, draws) * 100);
    return table[n];
}

// one attempt of random_in_unit_disk (camera.h:65-75) on two stream words: p = 2 (vdc_2(n_a), vdc_3(n_b), 0) - (1, 1, 0), accepted unless dot(p, p) >= 1.
// The same operations as van_der_corput + the loop's test, on words instead of a generator state (raygen's block-wise form).
VPT_D bool lens_sample_accepted(const float* tables, uint32_t wa, uint32_t wb) {
    const float ua = (float)wa * 2.3283064365386963e-10f + 1.1641532182693481e-10f, ub = (float)wb * 2.3283064365386963e-10f + 1.1641532182693481e-10f;   // curand_uniform
    const float a = tables[(int)(ua * 100)], b = tables[101 + (int)(ub * 100)];
    const f3 pd = 2.0f * mk3(a, b, 0) - mk3(1.0f, 1.0f, 0.0f);
    return !(dot(pd, pd) >= 1.0f);
}

}  // namespace vpt
