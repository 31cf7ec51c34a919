// vpt_trace.hip -- the hot path: persistent wavefront tracer for gfx950 (wave64).
//
// What it computes: for every pixel-sample (pixel idx, iteration) the integrator part of
// the reference's `volume_rt_kernel` (source/render_kernel.cu:2227-2260): Philox stream
// init, jittered thin-lens primary ray, the depth pass (depth_calculator :1859), and
// direct_integrator (:1760) with delta tracking (`sample` :1556), residual-ratio
// tracking (`Tr` :1138), Henyey-Greenstein scattering (`sample_hg` :306), sun /
// point-light next-event estimation (:1478, :1445), emission (:1275) and the reference
// sphere bounce (:1807-1834).  The environment tail, accumulation and tonemap live in
// vpt_resolve.hip.
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
#include <hip/hip_runtime.h>

#include "vpt_device.h"
#include "vpt_rng.h"

namespace vpt {

enum : uint32_t {
    PH_IDLE = 0,
    // walking phases
    PH_W_FIRST = 1,   // delta tracking: depth pass + first integrator walk, fused
    PH_W_TRACK = 2,   // delta tracking, integrator
    PH_W_SUN = 3,     // ratio tracking towards the sun
    PH_W_PL = 4,      // ratio tracking towards a point light
    PH_W_SPH = 5,     // ratio tracking after the sphere bounce
    PH_W_EMIT = 6,    // emission march
    PH_W_LAST = 6,
    // transition phases, in successor order
    PH_T_FIRST = 16,
    PH_T_FIRST_DONE = 16,
    PH_T_REPLAY = 17,       // history overflow: restart the integrator from the primary ray
    PH_T_TRACK_DONE = 18,
    PH_T_SUN_DONE = 19,
    PH_T_PL_DONE = 20,
    PH_T_PL_NEXT = 21,
    PH_T_EMIT_CHECK = 22,
    PH_T_EMIT_DONE = 23,
    PH_T_SPH_DONE = 24,
    PH_T_OUTER_SECOND = 25,
    PH_T_OUTER_TOP = 26,
    PH_T_FINISH = 27,
};

#define VPT_HIST_CAP 12
#define VPT_CHUNK 256        // queue entries a wave claims per global atomic

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
VPT_D bool sphere_intersect(const TraceParams& P, f3 ray_pos, f3 ray_dir, float& t_min, float& t_max) {
    f3 orig = ray_pos - ld3(P.sph_center);
    float A = ray_dir.x * ray_dir.x + ray_dir.y * ray_dir.y + ray_dir.z * ray_dir.z;
    float B = 2 * (ray_dir.x * orig.x + ray_dir.y * orig.y + ray_dir.z * orig.z);
    float C = orig.x * orig.x + orig.y * orig.y + orig.z * orig.z - P.sph_radius * P.sph_radius;
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

// get_closest_object (render_kernel.cu:1118-1135): 0 none, 1 volume box, 2 sphere
VPT_D int closest_object(const TraceParams& P, f3 o, f3 d, f3 inv, float& t_min) {
    float tmin1 = VPT_M_INF, tmax1 = -VPT_M_INF, tmin2 = VPT_M_INF, tmax2 = -VPT_M_INF;
    bool i1 = box_intersect(ld3(P.root_pmin), ld3(P.root_pmax), o, inv, tmin1, tmax1);
    bool i2 = sphere_intersect(P, o, d, tmin2, tmax2);
    if (i1 && !i2) { t_min = tmin1; return 1; }
    if (!i1 && i2) { t_min = tmin2; return 2; }
    if (i1 && i2) {
        if (tmin1 < tmin2) { t_min = tmin1; return 1; }
        if (tmin2 < tmin1) { t_min = tmin2; return 2; }
    }
    return 0;
}

// Three-level point location (get_quadrant x3, render_kernel.cu:1102-1115 + :1193-1227).
// Child boxes follow divide_bbox (bvh_kernels.cu:150-202): child i covers
//   x: low half for i in {0,2,4,6}, high half otherwise
//   y: HIGH half for i in {0,1,4,5}, low half otherwise
//   z: low half for i < 4, high half otherwise
// and the first child (index order) whose CLOSED box contains p wins.
enum { LOC_LEAF = 0, LOC_EMPTY = 1, LOC_OUTSIDE = 2 };
VPT_D int locate(const TraceParams& P, const uint32_t* occ, f3 p, f3& nmin, f3& nmax, int& leaf) {
    f3 lo = ld3(P.root_pmin), hi = ld3(P.root_pmax);
    int path = 0;
#pragma unroll
    for (int level = 0; level < 3; ++level) {
        const float hx = (lo.x + hi.x) * 0.5f;
        const float hy = (lo.y + hi.y) * 0.5f;
        const float hz = (lo.z + hi.z) * 0.5f;
        const uint32_t mx = ((p.x >= lo.x && p.x <= hx) ? 0x55u : 0u) | ((p.x >= hx && p.x <= hi.x) ? 0xAAu : 0u);
        const uint32_t my = ((p.y >= hy && p.y <= hi.y) ? 0x33u : 0u) | ((p.y >= lo.y && p.y <= hy) ? 0xCCu : 0u);
        const uint32_t mz = ((p.z >= lo.z && p.z <= hz) ? 0x0Fu : 0u) | ((p.z >= hz && p.z <= hi.z) ? 0xF0u : 0u);
        const uint32_t m = mx & my & mz;
        if (m == 0) return LOC_OUTSIDE;
        const int c = __ffs((int)m) - 1;
        const bool xh = (c & 1) != 0;
        const bool yh = (c & 2) == 0;
        const bool zh = (c & 4) != 0;
        lo.x = xh ? hx : lo.x; hi.x = xh ? hi.x : hx;
        lo.y = yh ? hy : lo.y; hi.y = yh ? hi.y : hy;
        lo.z = zh ? hz : lo.z; hi.z = zh ? hi.z : hz;
        path = path * 8 + c;
        const int bit = (level == 0 ? 0 : (level == 1 ? 32 : 96)) + path;
        if (((occ[bit >> 5] >> (bit & 31)) & 1u) == 0) {
            nmin = lo;
            nmax = hi;
            return LOC_EMPTY;
        }
    }
    leaf = path;
    return LOC_LEAF;
}

// world -> normalised texture coordinates (render_kernel.cu:987-997)
VPT_D bool to_unit(const DVolume& v, f3 p, f3& u) {
    f3 q;
    q.x = v.m[0] * p.x + v.m[1] * p.y + v.m[2] * p.z + v.m[3];
    q.y = v.m[4] * p.x + v.m[5] * p.y + v.m[6] * p.z + v.m[7];
    q.z = v.m[8] * p.x + v.m[9] * p.y + v.m[10] * p.z + v.m[11];
    q.x = q.x - v.bmin[0];
    q.y = q.y - v.bmin[1];
    q.z = q.z - v.bmin[2];
    u.x = q.x / v.fdim[0];
    u.y = q.y / v.fdim[1];
    u.z = q.z / v.fdim[2];
    return !(u.x < .0f || u.y < .0f || u.z < .0f || u.x > 1.0f || u.y > 1.0f || u.z > 1.0f);
}

struct Taps {
    int i0, i1, j0, j1, k0, k1;
    float ax, ay, az;
};
VPT_D Taps make_taps(const DVolume& v, f3 u) {
    Taps t;
    float xb = u.x * v.fdim[0] - 0.5f;
    float yb = u.y * v.fdim[1] - 0.5f;
    float zb = u.z * v.fdim[2] - 0.5f;
    float fx = floorf(xb), fy = floorf(yb), fz = floorf(zb);
    t.ax = xb - fx;
    t.ay = yb - fy;
    t.az = zb - fz;
    int i = (int)fx, j = (int)fy, k = (int)fz;
    t.i0 = min(max(i, 0), v.dim[0] - 1);
    t.i1 = min(max(i + 1, 0), v.dim[0] - 1);
    t.j0 = min(max(j, 0), v.dim[1] - 1);
    t.j1 = min(max(j + 1, 0), v.dim[1] - 1);
    t.k0 = min(max(k, 0), v.dim[2] - 1);
    t.k1 = min(max(k + 1, 0), v.dim[2] - 1);
    return t;
}
// trilinear f32 fetch: CUDA "linear, normalised, clamp" addressing, fp32 weights, nested
// lerp x -> y -> z with lerp(a,b,t) = a + t*(b-a)
VPT_D float fetch_f32(const float* __restrict__ g, const DVolume& v, const Taps& t) {
    const uint32_t dx = (uint32_t)v.dim[0];
    const uint32_t r00 = ((uint32_t)t.k0 * (uint32_t)v.dim[1] + (uint32_t)t.j0) * dx;
    const uint32_t r10 = ((uint32_t)t.k0 * (uint32_t)v.dim[1] + (uint32_t)t.j1) * dx;
    const uint32_t r01 = ((uint32_t)t.k1 * (uint32_t)v.dim[1] + (uint32_t)t.j0) * dx;
    const uint32_t r11 = ((uint32_t)t.k1 * (uint32_t)v.dim[1] + (uint32_t)t.j1) * dx;
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
VPT_D f4 lerp4(f4 a, f4 b, float t) { return a + (b - a) * t; }
VPT_D f3 fetch_f4(const f4* __restrict__ g, const DVolume& v, const Taps& t) {
    const uint32_t dx = (uint32_t)v.dim[0];
    const uint32_t r00 = ((uint32_t)t.k0 * (uint32_t)v.dim[1] + (uint32_t)t.j0) * dx;
    const uint32_t r10 = ((uint32_t)t.k0 * (uint32_t)v.dim[1] + (uint32_t)t.j1) * dx;
    const uint32_t r01 = ((uint32_t)t.k1 * (uint32_t)v.dim[1] + (uint32_t)t.j0) * dx;
    const uint32_t r11 = ((uint32_t)t.k1 * (uint32_t)v.dim[1] + (uint32_t)t.j1) * dx;
    const f4 c00 = lerp4(g[r00 + t.i0], g[r00 + t.i1], t.ax);
    const f4 c10 = lerp4(g[r10 + t.i0], g[r10 + t.i1], t.ax);
    const f4 c01 = lerp4(g[r01 + t.i0], g[r01 + t.i1], t.ax);
    const f4 c11 = lerp4(g[r11 + t.i0], g[r11 + t.i1], t.ax);
    return xyz(lerp4(lerp4(c00, c10, t.ay), lerp4(c01, c11, t.ay), t.az));
}

// one volume's contribution at world position p (get_density / get_color / get_emission)
template <bool COLOR, bool EMIT, bool COUNT>
VPT_D void lookup_volume(const TraceParams& P, const DVolume& v, f3 p, bool want_density, bool want_color, bool want_emission,
                         float& density, f3& color, f3& emission, uint32_t& n_d, uint32_t& n_c, uint32_t& n_e) {
    f3 u;
    const bool inside = to_unit(v, p, u);
    Taps t;
    if (inside) t = make_taps(v, u);
    if (want_density) {
        if (COUNT) n_d++;
        if (inside) density += fetch_f32(v.density, v, t);
    }
    if (COLOR && want_color) {
        if (!v.has_color) {
            color = fmax3(color, mk3(1.0f));
        } else {
            if (COUNT) n_c++;
            f3 c = inside ? fetch_f4(v.color, v, t) : mk3(0.0f);
            color = fmax3(color, c);
        }
    }
    if (EMIT && want_emission) {
        if (v.has_emission) {
            if (COUNT) n_e++;
            if (inside) {
                float index = fetch_f32(v.emission, v, t);
                index = clampf(index * 255.0f / P.emission_pivot, .0f, 255.0f);
                const float* e = P.emission_lut + 3 * (int)index;
                emission += mk3(e[0], e[1], e[2]) * P.emission_scale;
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

// ---- stage 0: ray generation + compaction ----------------------------------------------------------
// volume_rt_kernel :2227-2251 for every pixel-sample of the batch, one thread each (full lanes).
// Writes into records[slot] either
//   * the FINAL path record, when the sample is not rendered (:2254) or its primary ray hits
//     neither the volume box nor the sphere (get_closest_object == 0: every iteration of
//     direct_integrator's loop is then a no-op and depth = 0), or
//   * a 64-byte RAY record {org0, t_hit | dir0, obj | philox block | counter, word, depth} and the
//     slot index into the work queue (wave-ballot compaction, one atomic per wave).
#define VPT_RAYGEN_ROWS 64        // pixel rows per raygen block (block = 64 x 4 threads, 16 passes)
template <bool COUNT>
__global__ __launch_bounds__(256) void raygen_kernel(const TraceParams P) {
    // grid: (ceil(W/64), ceil(H/64), iterations); a block sweeps a 64x64 pixel tile in 16 passes and
    // compacts its active rays in LDS, so the global queue tail sees ONE atomic per 4096 samples
    __shared__ uint32_t s_q[64 * VPT_RAYGEN_ROWS];
    __shared__ uint32_t s_n, s_base;
    if (threadIdx.x == 0 && threadIdx.y == 0) s_n = 0;
    __syncthreads();
    const int x = (int)(blockIdx.x * 64u + threadIdx.x);
    const uint32_t kiter = blockIdx.z;
    const uint32_t iteration = P.iter_begin + kiter * P.iter_stride;
    const int lane = __lane_id();
    const bool rendered = iteration < P.max_interactions && P.render;
    uint32_t n_final = 0;
    for (int pass = 0; pass < VPT_RAYGEN_ROWS / 4; ++pass) {
        const int y = (int)(blockIdx.y * VPT_RAYGEN_ROWS) + pass * 4 + (int)threadIdx.y;
        bool enqueue = false;
        uint32_t s = 0;
        if (x < (int)P.width && y < (int)P.height) {
            const uint32_t pixel = (uint32_t)y * P.width + (uint32_t)x;
            s = kiter * P.n_pixels + pixel;
            Rng rng;
            rng_init(rng, pixel, iteration * 4096u);
            uint32_t draws = 0;
            const float2 bn = P.blue_noise[(size_t)kiter * 65536 + (y % 256) * 256 + (x % 256)];
            const float u = (float)(x + bn.x) / (float)P.width;
            const float v = (float)(y + bn.y) / (float)P.height;
            // camera::get_ray, camera.h:131-136
            f3 pd;
            do {
                float a = van_der_corput(P.vdc_tables, rng, pixel, draws);
                float b = van_der_corput(P.vdc_tables + 101, rng, pixel, draws);
                pd = 2.0f * mk3(a, b, 0) - mk3(1.0f, 1.0f, 0.0f);
            } while (dot(pd, pd) >= 1.0f);
            const f3 rdk = P.cam.lens_radius * pd;
            const f3 offset = ld3(P.cam.u) * rdk.x + ld3(P.cam.v) * rdk.y;
            (void)rnd_simple(rng, pixel, draws);                          // the `time` draw
            const f3 org0 = ld3(P.cam.origin) + offset;
            const f3 B = ld3(P.cam.llc) + u * ld3(P.cam.horizontal) + v * ld3(P.cam.vertical) - ld3(P.cam.origin) - offset;
            const f3 dir0 = normalize(B);
            float4* dst = reinterpret_cast<float4*>(P.records + s);
            int obj = 0;
            float t_hit = 0.0f;
            if (rendered) obj = closest_object(P, org0, dir0, rcp3(dir0), t_hit);
            if (obj == 0) {
                // final: L = 0 (or WHITE when not rendering, :2248), beta = 1 (0), tr = 0, depth = 0
                const float l = rendered ? 0.0f : 1.0f, b = rendered ? 1.0f : 0.0f;
                dst[0] = make_float4(l, l, l, 0.0f);
                dst[1] = make_float4(b, b, b, 0.0f);
                dst[2] = make_float4(org0.x, org0.y, org0.z, __uint_as_float(rendered ? 1u : 0u));
                dst[3] = make_float4(dir0.x, dir0.y, dir0.z, 0.0f);
                n_final++;
            } else {
                // :1883-1888: sphere first -> depth is the distance to it
                const float depth = (obj == 2) ? length(org0 - (org0 + dir0 * t_hit)) : 0.0f;
                dst[0] = make_float4(org0.x, org0.y, org0.z, t_hit);
                dst[1] = make_float4(dir0.x, dir0.y, dir0.z, __uint_as_float((uint32_t)obj));
                dst[2] = make_float4(__uint_as_float(rng.o0), __uint_as_float(rng.o1), __uint_as_float(rng.o2), __uint_as_float(rng.o3));
                dst[3] = make_float4(__uint_as_float(rng.c0), __uint_as_float(rng.idx), depth, 0.0f);
                enqueue = true;
            }
        }
        const unsigned long long m = __ballot(enqueue);
        if (m != 0ull) {
            const int leader = __ffsll((long long)m) - 1;
            uint32_t base = 0;
            if (lane == leader) base = atomicAdd(&s_n, (uint32_t)__popcll(m));      // LDS atomic
            base = __shfl(base, leader);
            if (enqueue) s_q[base + __popcll(m & ((1ull << lane) - 1ull))] = s;
        }
    }
    __syncthreads();
    const uint32_t n = s_n;
    const uint32_t tid = threadIdx.y * 64u + threadIdx.x;
    if (tid == 0 && n) s_base = atomicAdd(P.queue_tail, n);
    __syncthreads();
    const uint32_t gbase = s_base;
    for (uint32_t i = tid; i < n; i += 256u) P.queue[gbase + i] = s_q[i];
    if (COUNT && n_final) atomicAdd(&P.counters->samples, (unsigned long long)n_final);
}

#ifndef VPT_TRACE_WAVES_PER_EU
#define VPT_TRACE_WAVES_PER_EU 2
#endif
template <bool MULTI, bool COLOR, bool EMIT, bool COUNT>
__global__ __launch_bounds__(256, VPT_TRACE_WAVES_PER_EU) void trace_kernel(const TraceParams P) {
    __shared__ uint32_t s_occ[20];
    __shared__ float s_hist[VPT_HIST_CAP * 256];      // [entry][thread]: densities seen by the fused first walk
    if (threadIdx.x < 19) s_occ[threadIdx.x] = P.occ[threadIdx.x];
    __syncthreads();

    const uint32_t total = *P.queue_count;          // rays that entered the volume box / hit the sphere
    const int lane = __lane_id();
    const f3 root_lo = ld3(P.root_pmin), root_hi = ld3(P.root_pmax);
    const f3 sun_dir = ld3(P.sun_dir);
    const float inv_max = 1.0f / P.max_ext;                 // :1645
    const float inv_dm = 1.0f / P.density_mult;             // :1646
    const float sigma_c = P.min_ext;                        // :1164
    const float sigma_r_inv = 1.0f / (P.max_ext - sigma_c); // :1165
    const uint32_t regen_min = P.regen_min;
    const uint32_t trans_min = P.trans_min;

    // ---- lane state -------------------------------------------------------------------
    uint32_t phase = PH_IDLE;
    uint32_t pixel = 0, kiter = 0;
    Rng rng;
    rng.c0 = rng.o0 = rng.o1 = rng.o2 = rng.o3 = rng.idx = rng.carry = rng.has_carry = 0u;
    uint32_t draws = 0;            // draws since the stream origin of this sample (offset iteration*4096)
    uint32_t cam_draws = 0;        // draws consumed by camera::get_ray
    f3 pos = mk3(0.0f), dir = mk3(0.0f), inv = mk3(0.0f);      // current walk ray
    f3 ppos = mk3(0.0f), pdir = mk3(0.0f);                      // path ray parked during shadow / emission walks
    f3 org0 = mk3(0.0f), dir0 = mk3(0.0f);                      // primary ray
    f3 env_pos = mk3(0.0f);
    f3 beta = mk3(1.0f), L = mk3(0.0f), Ld = mk3(0.0f);
    float t = 0.0f, distance = 0.0f;
    float trw = 1.0f;              // running transmittance of a Tr walk / its result
    float alpha = 0.0f;            // `tr` of volume_rt_kernel (:2251), fed to sample() as Alpha
    float depth = 0.0f;
    f3 wgt = mk3(1.0f);            // return value of sample()
    int rd = 0, vd = 0, budget = 0, light_index = 0;
    uint32_t n_hist = 0;
    bool mi = false, geo = false, obj2 = false;
    int gco_obj = -1;              // cached get_closest_object result for the current (pos, dir), -1 = stale
    float gco_t = 0.0f;
    float sph_factor = 0.0f;
    uint32_t n_d = 0, n_c = 0, n_e = 0, n_steps = 0, n_skips = 0;
    bool more = true;
    uint32_t chunk_next = 0, chunk_end = 0;

    for (;;) {
        // ==== refill idle lanes from the compacted ray queue ===================================
        // The wave owns a chunk [chunk_next, chunk_end) of queue entries at a time, so the global
        // dequeue counter sees one atomic per VPT_CHUNK rays.
        const unsigned long long idle = __ballot(phase == PH_IDLE);
        if (idle != 0ull) {
            const unsigned long long active = __ballot(1);
            const uint32_t n_idle = (uint32_t)__popcll(idle);
            if (chunk_next == chunk_end && more && (n_idle >= regen_min || idle == active)) {
                const int leader = __ffsll((long long)active) - 1;
                uint32_t base = 0;
                if (lane == leader) base = atomicAdd(P.work_counter, (uint32_t)VPT_CHUNK);
                base = __shfl(base, leader);
                chunk_next = min(base, total);
                chunk_end = min(base + (uint32_t)VPT_CHUNK, total);
                if (chunk_end == total) more = false;
            }
            const uint32_t avail = chunk_end - chunk_next;
            if (avail == 0u && !more && idle == active) break;
            if (avail != 0u && (n_idle >= regen_min || idle == active)) {
                const uint32_t first = chunk_next;
                chunk_next += min(n_idle, avail);
                if (phase == PH_IDLE) {
                    const uint32_t rank = __popcll(idle & ((1ull << lane) - 1ull));
                    if (rank < avail) {
                        const uint32_t slot = P.queue[first + rank];
                        kiter = slot / P.n_pixels;
                        pixel = slot - kiter * P.n_pixels;
                        const uint32_t iteration = P.iter_begin + kiter * P.iter_stride;
                        const float4* src = reinterpret_cast<const float4*>(P.records + slot);
                        const float4 q0 = src[0], q1 = src[1], q2 = src[2], q3 = src[3];
                        org0 = mk3(q0.x, q0.y, q0.z);
                        gco_t = q0.w;
                        dir0 = mk3(q1.x, q1.y, q1.z);
                        gco_obj = (int)__float_as_uint(q1.w);
                        rng.o0 = __float_as_uint(q2.x); rng.o1 = __float_as_uint(q2.y);
                        rng.o2 = __float_as_uint(q2.z); rng.o3 = __float_as_uint(q2.w);
                        rng.c0 = __float_as_uint(q3.x);
                        rng.idx = __float_as_uint(q3.y);
                        rng.carry = 0u; rng.has_carry = 0u;
                        depth = q3.z;
                        draws = rng.c0 * 4u + rng.idx - iteration * 4096u;
                        cam_draws = draws;
                        // depth_calculator :1859-1889 and direct_integrator :1772-1785 start from the
                        // same ray with the same rng copy
                        alpha = 0.0f;
                        env_pos = org0;
                        pos = org0;
                        dir = dir0;
                        inv = rcp3(dir);
                        L = mk3(0.0f);
                        beta = mk3(1.0f);
                        mi = false;
                        rd = 1;
                        n_d = n_c = n_e = n_steps = n_skips = 0;
                        if (gco_obj == 1) {
                            pos += dir * (gco_t + VPT_EPS);
                            gco_obj = -1;
                            vd = 1;
                            t = 0.0f; geo = false; obj2 = false; wgt = mk3(1.0f);
                            n_hist = 0;
                            phase = PH_W_FIRST;
                        } else {
                            phase = PH_T_OUTER_TOP;      // sphere first: cached result is reused there
                        }
                    }
                }
            }
        }

        // ==== one tracking step for every walking lane =====================================
        rng_top_up(rng, pixel);
        if (phase >= PH_W_FIRST && phase <= PH_W_LAST) {
            const bool is_sample = phase <= PH_W_TRACK;
            const bool is_emit = EMIT && phase == PH_W_EMIT;
            f3 nmin, nmax;
            int leaf = 0;
            const int st = locate(P, s_occ, pos, nmin, nmax, leaf);
            bool done = false;
            if (st == LOC_OUTSIDE) {
                done = true;
            } else if (st == LOC_EMPTY) {
                // empty node: push to its far side, at least 0.1 (:1613-1616)
                float t_min, t_max;
                box_intersect(nmin, nmax, pos, inv, t_min, t_max);
                t_max = fmax_(t_max, 0.1f);
                pos += dir * t_max;
                if (COUNT) n_skips++;
            } else {
                if (is_sample) {
                    // :1647-1651
                    float t_min, t_max, geo_dist;
                    box_intersect(root_lo, root_hi, pos, inv, t_min, distance);
                    if (sphere_intersect(P, pos, dir, geo_dist, t_max)) {
                        distance = geo_dist;
                        geo = true;
                    }
                }
                const float lg = det_logf(1 - rnd(rng, draws));
                if (COUNT) n_steps++;
                if (is_sample) t -= lg * inv_max * inv_dm;                       // :1652
                else if (is_emit) t -= lg * inv_max * P.tr_depth / P.extinction[0];  // :1331
                else t -= lg * sigma_r_inv * P.tr_depth;                         // :1231
                if (!is_emit && t >= distance) {
                    if (is_sample && geo) obj2 = true;                           // :1654-1657
                    done = true;
                } else {
                    pos += dir * t;                                              // cumulative t (Q-list 1)
                    if (!contains(root_lo, root_hi, pos)) {
                        done = true;
                    } else {
                        float density = 0.0f;
                        f3 Cd = COLOR ? mk3(0.0f) : mk3(1.0f);
                        f3 em = mk3(0.0f);
                        if (!MULTI) {
                            lookup_volume<COLOR, EMIT, COUNT>(P, P.vol0, pos, !is_emit, is_sample, is_emit, density, Cd, em, n_d, n_c, n_e);
                        } else {
                            const uint32_t b = P.leaf_offsets[leaf], e = P.leaf_offsets[leaf + 1];
                            for (uint32_t q = b; q < e; ++q) {
                                const DVolume& v = P.volumes[P.leaf_indices[q]];
                                lookup_volume<COLOR, EMIT, COUNT>(P, v, pos, !is_emit, is_sample, is_emit, density, Cd, em, n_d, n_c, n_e);
                            }
                        }
                        if (is_sample) {
                            // :1667-1675
                            int index = (int)floorf(fmin_(fmax_((density * inv_max * 255.0f / P.emission_pivot), 0.0f), 255.0f));
                            const float* dc = P.density_color_lut + 3 * index;
                            if (alpha < 1.0f) alpha += density;
                            if (phase == PH_W_FIRST) {
                                if (n_hist < VPT_HIST_CAP) s_hist[n_hist * 256 + threadIdx.x] = density;
                                n_hist++;
                            }
                            if (density * inv_max > rnd(rng, draws)) {
                                mi = true;
                                wgt = (ld3(P.albedo) * Cd * mk3(dc[0], dc[1], dc[2]) / ld3(P.extinction)) * P.energy_inject;
                                done = true;
                            }
                        } else if (is_emit) {
                            Ld += em;                                            // :1335
                        } else {
                            trw *= 1 - ((density - sigma_c) * sigma_r_inv);      // :1239
                            const float s2 = trw * trw;
                            if (sqrtf(s2 + s2 + s2) < VPT_EPS) done = true;      // :1261
                        }
                    }
                }
            }
            if (done) {
                if (phase == PH_W_FIRST) phase = PH_T_FIRST_DONE;
                else if (phase == PH_W_TRACK) phase = PH_T_TRACK_DONE;
                else if (phase == PH_W_EMIT) phase = PH_T_EMIT_DONE;
                else {
                    // Tr epilogue :1166,:1267: clamp(tr * exp(-sigma_c * distance), 0, 1)
                    trw = clampf(trw * expf(-sigma_c * distance), .0f, 1.0f);
                    phase = (phase == PH_W_SUN) ? PH_T_SUN_DONE : (phase == PH_W_PL ? PH_T_PL_DONE : PH_T_SPH_DONE);
                }
            }
        }

        // ==== transitions: integrator control flow between walks ===========================
        // States are visited in successor order, so e.g. TRACK_DONE -> OUTER_SECOND -> OUTER_TOP
        // -> FINISH resolves in a single pass.  Each state draws at most 2 random numbers.
        // Transition code runs with few lanes, so it is batched like the refill: entered only when
        // >= trans_min lanes wait for it or nothing else can make progress.
        const unsigned long long tmask = __ballot(phase >= PH_T_FIRST);
        const bool run_trans = tmask != 0ull && ((uint32_t)__popcll(tmask) >= trans_min || !__any(phase >= PH_W_FIRST && phase <= PH_W_LAST));
        while (run_trans && __any(phase >= PH_T_FIRST)) {
            rng_top_up(rng, pixel);
            bool start_tr = false;
            uint32_t tr_walk_phase = PH_IDLE, tr_done_phase = PH_IDLE;
            f3 tr_dir = mk3(0.0f);

            if (phase == PH_T_FIRST_DONE) {
                // the walk just finished IS depth_calculator's walk (:1879-1881) ...
                depth = mi ? length(org0 - pos) : .0f;
                // ... and direct_integrator's first sample() call (:1789), which would add the same
                // densities to Alpha a second time (:1670)
                if (alpha < 1.0f) {
                    if (n_hist > VPT_HIST_CAP) {
                        phase = PH_T_REPLAY;
                    } else {
                        for (uint32_t i = 0; i < n_hist; ++i)
                            if (alpha < 1.0f) alpha += s_hist[i * 256 + threadIdx.x];
                    }
                }
                if (COUNT && phase == PH_T_FIRST_DONE) {
                    // the reference walks this segment twice (depth pass + integrator): count it twice
                    n_d += n_d; n_c += n_c; n_steps += n_steps; n_skips += n_skips;
                }
                if (phase == PH_T_FIRST_DONE) phase = PH_T_TRACK_DONE;
            }
            if (phase == PH_T_REPLAY) {
                // history overflow (long walk through thin medium): replay the integrator's first walk
                // for real, from the primary ray and the post-camera rng state
                const uint32_t iteration = P.iter_begin + kiter * P.iter_stride;
                rng_init(rng, pixel, iteration * 4096u + cam_draws);
                draws = cam_draws;
                pos = org0;
                dir = dir0;
                inv = rcp3(dir);
                mi = false;
                rd = 1;
                gco_obj = -1;
                phase = PH_T_OUTER_TOP;
            }
            if (phase == PH_T_TRACK_DONE) {
                // :1789-1796
                beta *= wgt;
                const bool brk = is_black(beta) || obj2;
                if (!brk && mi) {
                    sample_hg(dir, rng, draws, P.phase_g1);
                    inv = rcp3(dir);
                }
                gco_obj = -1;
                vd++;
                if (!brk && vd <= P.volume_depth) {
                    mi = false;
                    t = 0.0f; geo = false; obj2 = false; wgt = mk3(1.0f);
                    phase = PH_W_TRACK;
                } else if (mi) {
                    // estimate_sun :1478-1516
                    ppos = pos;
                    pdir = dir;
                    start_tr = true; tr_dir = sun_dir; tr_walk_phase = PH_W_SUN; tr_done_phase = PH_T_SUN_DONE;
                } else {
                    phase = PH_T_OUTER_SECOND;
                }
            } else if (phase == PH_T_SUN_DONE) {
                const float cos_theta = dot(pdir, sun_dir);
                const float phase_pdf = henyey_greenstein(cos_theta, P.phase_g1);
                const f3 Lsun = mk3(trw) * phase_pdf;
                L += (Lsun * ld3(P.sun_color) * P.sun_mult) * beta;                 // :1514, :1798
                if (P.num_lights > 0) {
                    budget = 10;                                                    // :1459
                    Ld = mk3(0.0f);
                    phase = PH_T_PL_NEXT;
                } else {
                    phase = PH_T_EMIT_CHECK;
                }
            } else if (phase == PH_T_PL_DONE) {
                if (budget < P.num_lights) {
                    // point_light::Le, light.h:104-121
                    const DPointLight& lt = P.lights[light_index];
                    const f3 lp = ld3(lt.pos);
                    const f3 wi = normalize(lp - ppos);
                    const float cos_theta = dot(pdir, wi);
                    const float phase_pdf = henyey_greenstein(cos_theta, P.phase_g1);
                    const float sqr_dist = length(lp * lp - ppos * ppos);
                    const float falloff = 1 / sqr_dist;
                    Ld += ld3(lt.color) * lt.power * mk3(trw) * phase_pdf * falloff;
                }
                budget--;
                if (budget >= 0) phase = PH_T_PL_NEXT;
                else {
                    L += Ld * beta;                                                 // :1799
                    phase = PH_T_EMIT_CHECK;
                }
            }
            if (phase == PH_T_PL_NEXT) {
                // estimate_point_light :1461-1466 (1 draw)
                light_index = (int)floorf(rnd(rng, draws) * P.num_lights);
                if (light_index > P.num_lights - 1) light_index = P.num_lights - 1;  // rand()==1.0f guard
                const DPointLight& lt = P.lights[light_index];
                start_tr = true; tr_dir = normalize(ld3(lt.pos) - ppos); tr_walk_phase = PH_W_PL; tr_done_phase = PH_T_PL_DONE;
            } else if (phase == PH_T_EMIT_CHECK || phase == PH_T_EMIT_DONE || phase == PH_T_SPH_DONE) {
                if (phase == PH_T_EMIT_DONE) L += Ld;                               // :1803
                if (phase == PH_T_SPH_DONE) L += ld3(P.sun_color) * P.sun_mult * mk3(trw) * sph_factor * beta;  // :1832
                pos = ppos;
                dir = pdir;
                inv = rcp3(dir);
                gco_obj = -1;
                if (phase == PH_T_EMIT_CHECK && EMIT && P.emission_scale > 0) {     // :1802 (mi is true here)
                    t = 0.0f;
                    Ld = mk3(0.0f);
                    phase = PH_W_EMIT;
                } else if (phase == PH_T_SPH_DONE) {
                    env_pos = pos;                                                  // :1833
                    rd++;
                    phase = PH_T_OUTER_TOP;
                } else {
                    phase = PH_T_OUTER_SECOND;
                }
            }
            if (phase == PH_T_OUTER_SECOND) {
                if (gco_obj < 0) gco_obj = closest_object(P, pos, dir, inv, gco_t); // :1806
                if (gco_obj == 2) {
                    // sphere bounce :1809-1833 (2 draws)
                    pos += dir * gco_t;
                    const f3 normal = normalize((pos - ld3(P.sph_center)) / P.sph_radius);
                    const f3 nl = dot(normal, dir) < 0 ? normal : normal * -1;
                    const float phi = 2 * VPT_PI * rnd(rng, draws);
                    const float r2 = rnd(rng, draws);
                    const float r2s = sqrtf(r2);
                    const f3 w = normalize(nl);
                    const f3 uu = normalize(cross(((double)fabsf(w.x) > .1 ? mk3(0, 1, 0) : mk3(1, 0, 0)), w));
                    const f3 vv = cross(w, uu);
                    float sp, cp;
                    det_sincosf(phi, &sp, &cp);
                    const f3 hemisphere_dir = normalize(uu * cp * r2s + vv * sp * r2s + w * sqrtf(1 - r2));
                    const f3 ref = reflect(dir, nl);
                    dir = lerp3(ref, hemisphere_dir, P.sph_roughness);
                    pos += normal * VPT_EPS;
                    beta *= ld3(P.sph_color);
                    sph_factor = fmax_(dot(sun_dir, normal), .0f);
                    ppos = pos;
                    pdir = dir;
                    gco_obj = -1;
                    start_tr = true; tr_dir = sun_dir; tr_walk_phase = PH_W_SPH; tr_done_phase = PH_T_SPH_DONE;
                } else {
                    rd++;                          // same ray next iteration: the cached result stays valid
                    phase = PH_T_OUTER_TOP;
                }
            }
            if (phase == PH_T_OUTER_TOP) {
                if (rd > P.ray_depth) {
                    phase = PH_T_FINISH;
                } else {
                    if (gco_obj < 0) gco_obj = closest_object(P, pos, dir, inv, gco_t);   // :1782
                    if (gco_obj == 1) {
                        pos += dir * (gco_t + VPT_EPS);
                        gco_obj = -1;
                        vd = 1;
                        mi = false;
                        t = 0.0f; geo = false; obj2 = false; wgt = mk3(1.0f);
                        phase = PH_W_TRACK;
                    } else if (gco_obj == 0) {
                        // nothing ahead: the second get_closest_object (:1806) sees the same ray, so
                        // this and every later iteration is a no-op -> finish (exact)
                        phase = PH_T_FINISH;
                    } else {
                        phase = PH_T_OUTER_SECOND;   // sphere is closest: handled next pass
                    }
                }
            }
            if (phase == PH_T_FINISH) {
                const f3 od = dir;
                float4* dst = reinterpret_cast<float4*>(P.records + ((size_t)kiter * P.n_pixels + pixel));
                dst[0] = make_float4(L.x, L.y, L.z, fmin_(alpha, 1.0f));           // tr = fminf(tr, 1) :1854
                dst[1] = make_float4(beta.x, beta.y, beta.z, depth);
                dst[2] = make_float4(env_pos.x, env_pos.y, env_pos.z, __uint_as_float(1u));
                dst[3] = make_float4(od.x, od.y, od.z, 0.0f);
                if (COUNT) {
                    atomicAdd(&P.counters->samples, 1ull);
                    atomicAdd(&P.counters->density_lookups, (unsigned long long)n_d);
                    atomicAdd(&P.counters->color_lookups, (unsigned long long)n_c);
                    atomicAdd(&P.counters->emission_lookups, (unsigned long long)n_e);
                    atomicAdd(&P.counters->tracking_steps, (unsigned long long)n_steps);
                    atomicAdd(&P.counters->skip_steps, (unsigned long long)n_skips);
                }
                phase = PH_IDLE;
            }

            // ---- Tr prologue :1153-1167 (shared by sun / point-light / sphere shadow rays) ---
            if (start_tr) {
                pos = ppos;
                dir = tr_dir;
                inv = rcp3(dir);
                bool walking = true;
                float t_min, t_max;
                if (!contains(root_lo, root_hi, pos)) {
                    if (box_intersect(root_lo, root_hi, pos, inv, t_min, t_max)) pos += dir * (t_min + VPT_EPS);
                    else { trw = 1.0f; walking = false; }                           // misses the volume box
                }
                if (walking) {
                    float geo_dist;
                    box_intersect(root_lo, root_hi, pos, inv, t_min, distance);
                    if (sphere_intersect(P, pos, dir, geo_dist, t_max)) { trw = 0.0f; walking = false; }   // :1160
                }
                if (walking) {
                    t = 0.0f;
                    trw = 1.0f;
                    phase = tr_walk_phase;
                } else {
                    phase = tr_done_phase;
                }
            }
        }
    }
}

// ---- launcher -------------------------------------------------------------------------------
template <bool MULTI, bool COLOR, bool EMIT>
static hipError_t launch_variant(const TraceParams& P, int blocks, hipStream_t stream) {
    if (P.counters) hipLaunchKernelGGL((trace_kernel<MULTI, COLOR, EMIT, true>), dim3(blocks), dim3(256), 0, stream, P);
    else hipLaunchKernelGGL((trace_kernel<MULTI, COLOR, EMIT, false>), dim3(blocks), dim3(256), 0, stream, P);
    return hipGetLastError();
}

hipError_t launch_raygen(const TraceParams& P, hipStream_t stream) {
    const dim3 grid((P.width + 63u) / 64u, (P.height + VPT_RAYGEN_ROWS - 1u) / VPT_RAYGEN_ROWS, P.iter_count), block(64, 4, 1);
    if (P.counters) hipLaunchKernelGGL((raygen_kernel<true>), grid, block, 0, stream, P);
    else hipLaunchKernelGGL((raygen_kernel<false>), grid, block, 0, stream, P);
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
