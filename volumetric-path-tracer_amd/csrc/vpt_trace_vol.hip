// vpt_trace_vol.hip -- the tracer for integrator != 0: the reference's PBRT-style `vol_integrator`
// (source/render_kernel.cu:1712-1756) with `uniform_sample_one_light` (:1519-1554), `estimate_sun`
// (:1478), `estimate_point_light` (:1445), `estimate_sky` (:1356-1443: MIS of the environment with
// `draw_sample_from_distribution` :167-253, `pdf_li` :1342, `sample_spherical` :292,
// `sample_env_tex` :897, `sample_atmosphere` :839) and `estimate_emission` (:1275), plus the depth
// pass (`depth_calculator` :1859).
//
// Same organisation as vpt_trace.hip (persistent waves, per-lane path state machine, one shared
// tracking-step body for every walk kind, queue-fed refill, fused depth pass + first walk); only
// the integrator control flow between walks differs:
//
//   for depth = 1 .. ray_depth:                       sample()  -> W_FIRST / W_TRACK
//       beta *= sample(); black -> stop                         -> T_VTRACK_DONE
//       on a medium interaction:
//           one of {sun, point lights, sky} (1 draw)            -> W_TR (1, 11 or <= 2 walks)
//           + estimate_emission                                 -> W_EMIT
//           sample_hg                                           -> T_VSCATTER
//   L += beta * sample_atmosphere(...)                          -> tail kernel (vpt_tail.hip)
//
// Exact early-out: a sample() that ends outside the root box makes every later loop iteration a
// no-op (get_quadrant(root) == -1 -> break before any draw, :1606/:1629), so the path is finished.
//
// Random numbers: states draw at most 2 numbers per pass and a pass starts with a refill point
// (vpt_rng.h), so states that draw never chain inside one pass (`drew`).  By-value rng copies of
// the reference (Q-list 2: draw_sample_from_distribution :169, sample_spherical :293) are peeks.
//
// Arithmetic: the walk, phase sampling and shadow-ray directions are strict; the environment
// values (sky radiance, HDRI texels, pdf_li's acos/atan2) are value-only (vpt_sky.h) -- their
// only influence on control flow is the `isBlack(Li)` / `pdf == 0` tests (:1383,:1393,:1416).
#include <hip/hip_runtime.h>

#include "vpt_sky.h"
#include "vpt_walk.h"

namespace vpt {

enum : uint32_t {
    VH_IDLE = 0,
    VH_W_FIRST = 1,    // delta tracking: depth pass + first integrator walk, fused
    VH_W_TRACK = 2,    // delta tracking
    VH_W_TR = 3,       // ratio tracking (shadow ray); tr_next says where to continue
    VH_W_EMIT = 4,     // emission march
    VH_W_LAST = 4,
    VH_T_FIRST = 16,
    VH_T_FIRST_DONE = 16,
    VH_T_REPLAY = 17,
    VH_T_VTRACK_DONE = 18,
    VH_T_SUN_DONE = 19,
    VH_T_PL_DONE = 20,
    VH_T_PL_NEXT = 21,
    VH_T_SKY_A0 = 22,
    VH_T_SKY_A1 = 23,
    VH_T_SKY_B = 24,
    VH_T_SKY_C = 25,
    VH_T_SKY_D = 26,
    VH_T_SKY_END = 27,
    VH_T_LIGHT_DONE = 28,
    VH_T_EMIT_DONE = 29,
    VH_T_SCATTER = 30,
    VH_T_FINISH = 31,
};

VPT_D float power_heuristic(float f_pdf, float g_pdf) {      // light.h:65-69 with nf = ng = 1
    const float f = 1 * f_pdf, g = 1 * g_pdf;
    return (f * f) / (f * f + g * g);
}
VPT_D float tex_f(const DTexture& t, float u, float v) { return tex2d(t, u, v).x; }

// draw_sample_from_distribution :167-253, on a COPY of the rng (2 draws peeked)
VPT_D float draw_sample_from_distribution(const TraceParams& P, Rng rng, f3& wo) {
    uint32_t d = 0;
    const float xi = rnd(rng, d);
    const float zeta = rnd(rng, d);
    const int res = P.env_sample_tex_res;
    int first = 0, len = res;
    while (len > 0) {
        const int half = len >> 1, middle = first + half;
        if (tex_f(P.env_marginal_cdf_tex, (float)middle, 0.0f) <= xi) {
            first = middle + 1;
            len -= half + 1;
        } else len = half;
    }
    const int v = min(max(first - 1, 0), res - 2);
    float dv = xi - tex_f(P.env_marginal_cdf_tex, (float)v, 0.0f);
    const float d_cdf_marginal = tex_f(P.env_marginal_cdf_tex, (float)(v + 1), 0.0f) - tex_f(P.env_marginal_cdf_tex, (float)v, 0.0f);
    if (d_cdf_marginal > .0f) dv /= d_cdf_marginal;
    const float marginal_pdf = tex_f(P.env_marginal_func_tex, v + dv, 0.0f) / P.env_marginal_int;
    const float theta = (((float)v + dv) / (float)res) * VPT_PI;
    first = 0, len = res;
    while (len > 0) {
        const int half = len >> 1, middle = first + half;
        if (tex_f(P.env_cdf_tex, (float)middle, (float)v) <= zeta) {
            first = middle + 1;
            len -= half + 1;
        } else len = half;
    }
    const int u = min(max(first - 1, 0), res - 2);
    float du = zeta - tex_f(P.env_cdf_tex, (float)u, (float)v);
    const float d_cdf_conditional = tex_f(P.env_cdf_tex, (float)(u + 1), (float)v) - tex_f(P.env_cdf_tex, (float)u, (float)v);
    if (d_cdf_conditional > 0) du /= d_cdf_conditional;
    const float conditional_pdf = tex_f(P.env_func_tex, u + du, (float)v) / tex_f(P.env_marginal_func_tex, (float)v, 0.0f);
    const float phi = (((float)u + du) / (float)res) * VPT_PI * 2.0f;
    float sin_theta, cos_theta, sin_phi, cos_phi;
    det_sincosf(theta, &sin_theta, &cos_theta);
    det_sincosf(phi, &sin_phi, &cos_phi);
    wo = normalize(mk3(sin_theta * cos_phi, sin_theta * sin_phi, cos_theta));
    return (marginal_pdf * conditional_pdf) / (2 * VPT_PI * VPT_PI * sin_theta);
}
// pdf_li :1342-1354 + draw_pdf_from_distribution :258-269 (INV_2_PI / INV_PI are unparenthesised
// macros, Q-list 12; the coordinates really are divided by 2 pi^2 sin(theta))
VPT_D float pdf_li(const TraceParams& P, f3 wi) {
    const float theta = acosf(clampf(wi.y, -1.0f, 1.0f));
    const float phi = atan2f(wi.z, wi.x);
    const float sin_theta = sinf(theta);
    if (sin_theta == .0f) return .0f;
    const float denom = 2.0f * VPT_PI * VPT_PI * sin_theta;
    const float px = (phi * 1.0f / (2.0f * VPT_PI)) / denom, py = (theta * 1.0f / VPT_PI) / denom;
    const int res = P.env_sample_tex_res;
    const int iu = min(max((int)(px * res), 0), res - 1);
    const int iv = min(max((int)(py * res), 0), res - 1);
    return tex_f(P.env_func_tex, (float)iu, (float)iv) / tex_f(P.env_marginal_func_tex, (float)iv, 0.0f);
}
// sample_spherical :292-303, on a COPY of the rng (2 draws peeked)
VPT_D float sample_spherical(Rng rng, f3& wi) {
    uint32_t d = 0;
    const float phi = (2.0f * VPT_PI) * rnd(rng, d);
    const float cos_theta = 1.0f - 2.0f * rnd(rng, d);
    const float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
    float sp, cp;
    det_sincosf(phi, &sp, &cp);
    wi = mk3(cp * sin_theta, sp * sin_theta, cos_theta);
    return 1.0f / (4.0f * VPT_PI);
}

// SKYLUT: environment_type == 0 -- estimate_sky evaluates the Bruneton sky inside the tracer (large
// code, high register pressure); the HDRI instantiation (BASELINE config 4) carries none of it.
// out of line on purpose: the sky evaluation needs ~130 registers of its own; inlined (twice) into
// the state machine it pushed the whole kernel into scratch spills inside the walk loop
__device__ __noinline__ void sky_eval(const TraceParams* P, float px, float py, float pz, float dx, float dy, float dz, float* out) {
    const Sky<TraceParams> sky = {*P};
    const f3 r = sky.sample(mk3(px, py, pz), mk3(dx, dy, dz), ld3(P->sun_dir));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
VPT_D f3 sky_at(const TraceParams& P, f3 pos, f3 dir) {
    float o[3];
    sky_eval(&P, pos.x, pos.y, pos.z, dir.x, dir.y, dir.z, o);
    return mk3(o[0], o[1], o[2]);
}

// FOUR waves per SIMD here too (round 4), which takes the density history of the fused first walk out of the LDS: next to the 38 parked
// fields only one history entry would fit four workgroups per CU (measured: the replayed first walks, +39 % look-ups on config 4, cost more
// than the wave returns).  So the history lives in HBM ([workgroup][entry][thread], TraceParams::pool_hist: one coalesced 4-byte store per
// lane and first-walk step, read back eight entries at a time when the walk ends), 32 entries deep.  Config 4's tracer 14.7 -> 13.9 ms per
// 16 spp; the same history at three waves: 15.0 (profiles/r04_four_waves.txt).
#ifndef VPT_VOL_WAVES_PER_EU
#define VPT_VOL_WAVES_PER_EU 4
#endif
// ... except the SKYLUT instantiations (environment_type 0: estimate_sky evaluates the Bruneton sky inside the tracer), which want registers, not waves:
// 256 registers at two waves, 168 + 143 spilled at three, 128 + 240 at four -- tracer 6.5 / 6.9 / 9.9 ms per 8 spp of the dragon at 1080p
// (tools/vol_sky_probe.py, profiles/r04_four_waves.txt (o)).
#ifndef VPT_VOL_SKY_WAVES_PER_EU
#define VPT_VOL_SKY_WAVES_PER_EU 2
#endif
#ifndef VPT_VOL_HIST_HBM
#define VPT_VOL_HIST_HBM 1
#endif
#ifndef VPT_VOL_HIST_CAP
#define VPT_VOL_HIST_CAP (VPT_VOL_HIST_HBM ? 32 : VPT_HIST_CAP)
#endif
int trace_vol_blocks_per_cu(bool sky_in_tracer) { return sky_in_tracer ? VPT_VOL_SKY_WAVES_PER_EU : VPT_VOL_WAVES_PER_EU; }
size_t trace_vol_hist_floats_per_block() { return VPT_VOL_HIST_HBM ? (size_t)VPT_VOL_HIST_CAP * 256u : 0u; }
template <bool MULTI, bool COLOR, bool EMIT, bool COUNT, bool SKYLUT, bool A24>
__global__ __launch_bounds__(256, SKYLUT ? VPT_VOL_SKY_WAVES_PER_EU : VPT_VOL_WAVES_PER_EU) void trace_vol_kernel(const TraceParams P) {
    constexpr int HCAP = VPT_VOL_HIST_CAP;
    __shared__ uint32_t s_occ[20];
#if VPT_VOL_HIST_HBM
    float* const hist = P.pool_hist + (size_t)blockIdx.x * (size_t)(HCAP * 256) + threadIdx.x;
#else
    __shared__ float s_hist[HCAP * 256];
    float* const hist = s_hist + threadIdx.x;
#endif
    __shared__ float s_park[38 * 256];                // [field][thread]: path-level state parked in LDS (vpt_trace_common.h), 38 fields
    if (threadIdx.x < 19) s_occ[threadIdx.x] = P.occ[threadIdx.x];
    __syncthreads();

    const uint32_t total = *P.queue_count;
    const int lane = __lane_id();
    const WalkConst K = make_walk_const(P);
    const uint32_t regen_min = P.regen_min;
    const uint32_t trans_min = P.trans_min;

    uint32_t phase = VH_IDLE;
    uint32_t tr_next = VH_IDLE;     // state entered when the current shadow walk ends
    uint32_t pixel = 0, kiter = 0;
    Rng rng;
    rng.c0 = rng.o0 = rng.o1 = rng.o2 = rng.o3 = rng.idx = rng.carry = rng.has_carry = 0u;
    uint32_t draws = 0;
    Walk w;
    w.pos = w.dir = w.inv = mk3(0.0f);
    w.t = w.distance = 0.0f;
    w.trw = 1.0f;
    w.alpha = 0.0f;
    w.wgt = mk3(1.0f);
    w.Ld = mk3(0.0f);
    w.mi = w.geo = w.obj2 = false;
    float* const park = s_park + threadIdx.x;
    int* const parki = reinterpret_cast<int*>(park);
    const LdsF3 ppos = {park + 0 * 256}, pdir = {park + 3 * 256};     // path ray parked during shadow / emission walks
    const LdsF3 org0 = {park + 6 * 256}, dir0 = {park + 9 * 256};
    const LdsF3 beta = {park + 12 * 256}, L = {park + 15 * 256};
    const LdsF3 Lsel = {park + 18 * 256};                              // what the selected light estimator returned
    const LdsF3 A = {park + 21 * 256};                                 // beta * uniform_sample_one_light(...)
    const LdsF3 Li = {park + 24 * 256}, wi = {park + 27 * 256};        // estimate_sky scratch
    const LdsF light_pdf = {park + 30 * 256}, phase_pdf = {park + 31 * 256}, mis_w = {park + 32 * 256};
    const LdsF depth = {park + 33 * 256}, t_box = {park + 34 * 256};
    const LdsI budget = {parki + 35 * 256}, light_index = {parki + 36 * 256}, cam_draws_p = {parki + 37 * 256};
    uint32_t n_hist = 0;
    int retry = 0;                 // vol_integrator's depth loop (:1737): iterations left after the sample() call being
                                   // walked, i.e. current depth = ray_depth - retry (retries run inline, vpt_walk.h)
    WalkCounts cnt;
    cnt.n_d = cnt.n_c = cnt.n_e = cnt.n_steps = cnt.n_skips = 0;
    bool more = true;
    uint32_t chunk_next = 0, chunk_end = 0, chunk_base = 0;
    uint32_t qi0 = 0, qi1 = 0, qi2 = 0, qi3 = 0;       // this wave's chunk of queue entries, 4 per lane

    for (;;) {
        // ==== one tracking step for every walking lane =====================================
        // The density look-up of the single-volume kernels is split-phase (vpt_walk.h): the step requests its texels here ...
        rng_top_up(rng, pixel);
        constexpr bool SPLIT = !MULTI;
        Pending pd;
        pd.state = 0;
        {
            int r = WALK_GOES_ON;
            if (phase >= VH_W_FIRST && phase <= VH_W_LAST) {
                const int kind = phase <= VH_W_TRACK ? WALK_SAMPLE : (phase == VH_W_EMIT ? WALK_EMIT : WALK_TR);
                r = walk_step<MULTI, COLOR, EMIT, COUNT, false, A24, SPLIT, 256, HCAP>(P, s_occ, K, kind, phase == VH_W_FIRST, hist, n_hist, w, rng, draws, cnt,
                                                                            retry, phase == VH_W_TRACK, pd);
            }
            if (r != WALK_PENDING) pd.state = 0;
            if (r == WALK_DONE) {
                if (phase == VH_W_FIRST) phase = VH_T_FIRST_DONE;
                else if (phase == VH_W_TRACK) phase = VH_T_VTRACK_DONE;
                else if (phase == VH_W_EMIT) phase = VH_T_EMIT_DONE;
                else {
                    w.trw = tr_end(K, w);
                    phase = tr_next;
                }
            }
        }

        // ==== ... idle lanes are refilled from the compacted ray queue while those texels travel (the record read is the other
        // long latency of a pass; a refilled lane takes its first step in the next pass) ...
        const unsigned long long idle = __ballot(phase == VH_IDLE);
        if (idle != 0ull) {
            const unsigned long long active = __ballot(1);
            const uint32_t n_idle = (uint32_t)__popcll(idle);
            if (chunk_next == chunk_end && more && (n_idle >= regen_min || idle == active)) {
                claim_chunk<1>(P, total, lane, __ffsll((long long)active) - 1, chunk_next, chunk_end, more);
                // the chunk's queue entries are fetched once, here (4 per lane), so that a refill pays one
                // memory latency (the ray record) instead of two dependent ones
                chunk_base = chunk_next;
                qi0 = chunk_base + (uint32_t)lane < chunk_end ? ld_stream(P.queue + chunk_base + (uint32_t)lane) : 0u;
                qi1 = chunk_base + 64u + (uint32_t)lane < chunk_end ? ld_stream(P.queue + chunk_base + 64u + (uint32_t)lane) : 0u;
                qi2 = chunk_base + 128u + (uint32_t)lane < chunk_end ? ld_stream(P.queue + chunk_base + 128u + (uint32_t)lane) : 0u;
                qi3 = chunk_base + 192u + (uint32_t)lane < chunk_end ? ld_stream(P.queue + chunk_base + 192u + (uint32_t)lane) : 0u;
            }
            const uint32_t avail = chunk_end - chunk_next;
            if (avail == 0u && !more && idle == active) break;
            if (avail != 0u && (n_idle >= regen_min || idle == active)) {
                const uint32_t first = chunk_next;
                chunk_next += min(n_idle, avail);
                // entry e of the chunk sits in word (e >> 6) of lane (e & 63); all lanes take part in the exchange
                const uint32_t rank = __popcll(idle & ((1ull << lane) - 1ull));
                const uint32_t rel = first + rank - chunk_base;
                const int src_lane = (int)(rel & 63u);
                const uint32_t e0 = __shfl(qi0, src_lane), e1 = __shfl(qi1, src_lane), e2 = __shfl(qi2, src_lane), e3 = __shfl(qi3, src_lane);
                if (phase == VH_IDLE) {
                    if (rank < avail) {
                        const uint32_t word = rel >> 6;
                        const uint32_t entry = word == 0u ? e0 : (word == 1u ? e1 : (word == 2u ? e2 : e3));
                        float4 q0, q1, q2, q3;
                        load_ray_record(P, entry, kiter, pixel, q0, q1, q2, q3);
                        const uint32_t iteration = P.iter_begin + kiter * P.iter_stride;
                        const f3 o0 = mk3(q0.x, q0.y, q0.z), d0 = mk3(q1.x, q1.y, q1.z);
                        org0 = o0;
                        dir0 = d0;
                        const int obj = (int)__float_as_uint(q1.w);      // get_closest_object of the primary ray
                        rng.o0 = __float_as_uint(q2.x); rng.o1 = __float_as_uint(q2.y);
                        rng.o2 = __float_as_uint(q2.z); rng.o3 = __float_as_uint(q2.w);
                        rng.c0 = __float_as_uint(q3.x);
                        rng.idx = __float_as_uint(q3.y);
                        rng.carry = 0u; rng.has_carry = 0u;
                        depth = q3.z;
                        t_box = q3.w;
                        draws = rng.c0 * 4u + rng.idx - iteration * 4096u;
                        cam_draws_p = (int)draws;
                        // vol_integrator :1732-1737: every queued ray hits the root box
                        w.alpha = 0.0f;
                        w.dir = d0;
                        w.inv = rcp3(w.dir);
                        w.pos = o0 + w.dir * (q3.w + VPT_EPS);
                        L = mk3(0.0f);
                        beta = mk3(1.0f);
                        w.mi = false;
                        w.t = 0.0f; w.geo = false; w.obj2 = false; w.wgt = mk3(1.0f);
                        n_hist = 0;
                        cnt.n_d = cnt.n_c = cnt.n_e = cnt.n_steps = cnt.n_skips = 0;
                        // depth_calculator (:1875-1881) walks the very same segment with the same
                        // rng copy iff the box is the closest object (then t_min == t_box)
                        retry = P.ray_depth - 1;
                        phase = obj == 1 ? VH_W_FIRST : VH_W_TRACK;
                    }
                }
            }
        }

        // ==== ... and the steps with texels in flight interpolate and decide ====================
        if (SPLIT && pd.state != 0) {
            const int kind = phase <= VH_W_TRACK ? WALK_SAMPLE : WALK_TR;
            const bool done = walk_finish<COLOR, COUNT, A24, HCAP>(P, K, kind, phase == VH_W_FIRST, hist, n_hist, w, rng, draws, pd);
            if (done) {
                if (phase == VH_W_FIRST) phase = VH_T_FIRST_DONE;
                else if (phase == VH_W_TRACK) phase = VH_T_VTRACK_DONE;
                else {
                    w.trw = tr_end(K, w);
                    phase = tr_next;
                }
            }
        }

        // ==== transitions ===================================================================
        const unsigned long long tmask = __ballot(phase >= VH_T_FIRST);
        const bool run_trans = tmask != 0ull && ((uint32_t)__popcll(tmask) >= trans_min || !__any(phase >= VH_W_FIRST && phase <= VH_W_LAST));
        if (COUNT) {                     // schedule statistics (vpt_test_get_schedule), as in trace_kernel
            const unsigned long long wm = __ballot(phase >= VH_W_FIRST && phase <= VH_W_LAST), im = __ballot(phase == VH_IDLE);
            if (lane == 0) {
                atomicAdd(&P.counters->sched[0], 1ull);
                atomicAdd(&P.counters->sched[1], (unsigned long long)__popcll(wm));
                atomicAdd(&P.counters->sched[2], (unsigned long long)__popcll(tmask));
                atomicAdd(&P.counters->sched[3], (unsigned long long)__popcll(im));
                if (run_trans) atomicAdd(&P.counters->sched[4], 1ull);
            }
        }
        if (run_trans) {
        const ColdConst C = load_cold_const();      // the transition states' launch constants: scalar loads per pass (vpt_trace_common.h)
        while (__any(phase >= VH_T_FIRST)) {
            if (COUNT) {
                const unsigned long long tm = __ballot(phase >= VH_T_FIRST);
                if (lane == 0) {
                    atomicAdd(&P.counters->sched[5], 1ull);
                    atomicAdd(&P.counters->sched[6], (unsigned long long)__popcll(tm));
                }
            }
            rng_top_up(rng, pixel);
            bool drew = false;            // this lane consumed / peeked random numbers in this pass
            bool start_tr = false;
            f3 tr_dir = mk3(0.0f);
            uint32_t tr_done = VH_IDLE;

            if (phase == VH_T_FIRST_DONE) {
                depth = w.mi ? length(f3(org0) - w.pos) : .0f;                      // :1879-1881
                // vol_integrator's first sample() (:1740) adds the same densities to Alpha again
                if (w.alpha < 1.0f) {
                    if (n_hist > (uint32_t)HCAP) {
                        phase = VH_T_REPLAY;
                    } else {
#if VPT_VOL_HIST_HBM
                        for (uint32_t i0 = 0; i0 < n_hist; i0 += 8u) {          // eight entries per round trip
                            float hv[8];
#pragma unroll
                            for (uint32_t j = 0; j < 8u; ++j) hv[j] = i0 + j < n_hist ? hist[(i0 + j) * 256u] : 0.0f;
#pragma unroll
                            for (uint32_t j = 0; j < 8u; ++j)
                                if (i0 + j < n_hist && w.alpha < 1.0f) w.alpha += hv[j];
                        }
#else
                        for (uint32_t i = 0; i < n_hist; ++i)
                            if (w.alpha < 1.0f) w.alpha += hist[i * 256];
#endif
                    }
                }
                if (phase == VH_T_FIRST_DONE) {
                    if (COUNT) { cnt.n_d += cnt.n_d; cnt.n_c += cnt.n_c; cnt.n_steps += cnt.n_steps; cnt.n_skips += cnt.n_skips; }
                    phase = VH_T_VTRACK_DONE;
                }
            }
            if (phase == VH_T_REPLAY) {
                // history overflow: walk the integrator's first segment for real
                const uint32_t iteration = C.iter_begin + kiter * C.iter_stride;
                const uint32_t cam_draws = (uint32_t)(int)cam_draws_p;
                rng_init(rng, pixel, iteration * 4096u + cam_draws);
                draws = cam_draws;
                w.dir = f3(dir0);
                w.inv = rcp3(w.dir);
                w.pos = f3(org0) + w.dir * ((float)t_box + VPT_EPS);
                w.mi = false;
                w.t = 0.0f; w.geo = false; w.obj2 = false; w.wgt = mk3(1.0f);
                retry = C.ray_depth - 1;
                phase = VH_W_TRACK;
            }
            if (phase == VH_T_VTRACK_DONE) {
                // :1740-1747
                beta *= w.wgt;
                if (is_black(f3(beta))) {
                    phase = VH_T_FINISH;
                } else if (!w.mi) {
                    // no interaction: either the walk left the root box (every later sample() is a
                    // no-op) or it stopped at the sphere / exit distance inside the box (:1654) and
                    // the next loop iteration walks again from here -- inline in walk_step for
                    // W_TRACK walks (retry), as a new walk after the fused first one
                    f3 nmin, nmax;
                    int leaf;
                    const OccTop occ_top = {s_occ[0], s_occ[1], s_occ[2]};
                    if (retry == 0 || locate(P, s_occ, occ_top, w.pos, nmin, nmax, leaf) == LOC_OUTSIDE) {
                        phase = VH_T_FINISH;
                    } else {
                        w.t = 0.0f; w.geo = false; w.obj2 = false; w.wgt = mk3(1.0f);
                        retry--;
                        phase = VH_W_TRACK;
                    }
                } else {
                    // uniform_sample_one_light :1531-1551 (1 draw)
                    ppos = w.pos;
                    pdir = w.dir;
                    const float light_num = rnd(rng, draws) * 3;
                    drew = true;
                    Lsel = mk3(0.0f);
                    if (light_num < 1) {
                        if (C.sun_mult > .0f) { start_tr = true; tr_dir = C.sun_dir; tr_done = VH_T_SUN_DONE; phase = VH_W_TR; }
                        else phase = VH_T_LIGHT_DONE;
                    } else if (light_num >= 1 && light_num < 2) {
                        if (C.num_lights > 0) {
                            budget = 10;                                            // :1459
                            w.Ld = mk3(0.0f);
                            phase = VH_T_PL_NEXT;
                        } else phase = VH_T_LIGHT_DONE;
                    } else {
                        if (C.sky_mult > .0f) {
                            w.Ld = mk3(0.0f);
                            phase = VH_T_SKY_A0;
                        } else phase = VH_T_LIGHT_DONE;
                    }
                }
            } else if (phase == VH_T_SUN_DONE) {
                // estimate_sun :1478-1516
                const float cos_theta = dot(f3(pdir), C.sun_dir);
                const float pp = henyey_greenstein(cos_theta, C.phase_g1);
                Lsel = (mk3(w.trw) * pp) * C.sun_color * C.sun_mult;
                phase = VH_T_LIGHT_DONE;
            } else if (phase == VH_T_PL_DONE) {
                if ((int)budget < C.num_lights) {
                    const DPointLight& lt = C.lights[(int)light_index];             // point_light::Le, light.h:104-121
                    const f3 lp = ld3(lt.pos), pq = ppos;
                    const f3 wl = normalize(lp - pq);
                    const float cos_theta = dot(f3(pdir), wl);
                    const float pp = henyey_greenstein(cos_theta, C.phase_g1);
                    const float sqr_dist = length(lp * lp - pq * pq);
                    const float falloff = 1 / sqr_dist;
                    w.Ld += ld3(lt.color) * lt.power * mk3(w.trw) * pp * falloff;
                }
                budget--;
                if ((int)budget >= 0) phase = VH_T_PL_NEXT;
                else {
                    Lsel = w.Ld;
                    phase = VH_T_LIGHT_DONE;
                }
            }
            if (phase == VH_T_PL_NEXT && !drew) {
                // estimate_point_light :1461-1466 (1 draw)
                int li = (int)floorf(rnd(rng, draws) * C.num_lights);
                drew = true;
                if (li > C.num_lights - 1) li = C.num_lights - 1;                    // rand()==1.0f guard
                light_index = li;
                start_tr = true; tr_dir = normalize(ld3(C.lights[li].pos) - f3(ppos)); tr_done = VH_T_PL_DONE; phase = VH_W_TR;
            } else if (phase == VH_T_SKY_A0 && !drew) {
                // estimate_sky :1373-1374: two draws that are never used
                (void)rnd(rng, draws);
                (void)rnd(rng, draws);
                drew = true;
                phase = VH_T_SKY_A1;
            } else if (phase == VH_T_SKY_A1 && !drew) {
                // light sampling :1378-1402; the samplers work on a copy of the rng (peek)
                drew = true;
                f3 wv, Lv;
                float lp;
                if (SKYLUT) {
                    lp = draw_sample_from_distribution(P, rng, wv);
                    Lv = sky_at(P, f3(ppos), wv);
                } else {
                    lp = sample_spherical(rng, wv);
                    Lv = env_lookup(P.env_tex, wv);
                }
                light_pdf = lp;
                wi = wv;
                Li = Lv;
                phase = VH_T_SKY_C;
                if (lp > .0f && !is_black(Lv)) {
                    const float pp = henyey_greenstein(dot(f3(pdir), wv), C.phase_g1);
                    phase_pdf = pp;
                    if (pp > .0f) { start_tr = true; tr_dir = wv; tr_done = VH_T_SKY_B; phase = VH_W_TR; }
                }
            } else if (phase == VH_T_SKY_B) {
                const f3 Lv = f3(Li) * mk3(w.trw);
                if (!is_black(Lv)) {
                    const float lp = light_pdf, pp = phase_pdf;
                    const float weight = power_heuristic(lp, pp);
                    w.Ld += Lv * pp * weight / lp;
                }
                phase = VH_T_SKY_C;
            }
            if (phase == VH_T_SKY_C && !drew) {
                // phase-function sampling :1404-1431 (2 draws)
                f3 wv = pdir;
                const float pp = sample_hg_pdf(wv, rng, draws, C.phase_g1);
                wi = wv;
                drew = true;
                phase = VH_T_SKY_END;
                if (pp > .0f) {
                    const float lp = SKYLUT ? pdf_li(P, wv) : 1.0f / (4.0f * VPT_PI);
                    if (lp != 0.0f) {                                               // :1416 `return Ld`
                        mis_w = power_heuristic(pp, lp);
                        start_tr = true; tr_dir = wv; tr_done = VH_T_SKY_D; phase = VH_W_TR;
                    }
                }
            } else if (phase == VH_T_SKY_D) {
                f3 Lv;
                if (SKYLUT) Lv = sky_at(P, f3(ppos), f3(wi));
                else Lv = env_lookup(P.env_tex, f3(wi));
                if (!is_black(Lv)) w.Ld += Lv * mk3(w.trw) * (float)mis_w;
                phase = VH_T_SKY_END;
            }
            if (phase == VH_T_SKY_END) {
                Lsel = w.Ld * C.sky_mult;                                           // :1548
                phase = VH_T_LIGHT_DONE;
            }
            if (phase == VH_T_LIGHT_DONE) {
                const f3 Av = f3(beta) * (f3(Lsel) * 3.0f);                         // :1553, :1745
                A = Av;
                w.pos = f3(ppos);
                w.dir = f3(pdir);
                w.inv = rcp3(w.dir);
                if (EMIT && C.emission_scale != 0) {                                // :1285
                    w.t = 0.0f;
                    w.Ld = mk3(0.0f);
                    phase = VH_W_EMIT;
                } else {
                    L += Av + mk3(0.0f);
                    phase = VH_T_SCATTER;
                }
            } else if (phase == VH_T_EMIT_DONE) {
                L += f3(A) + w.Ld;                                                  // :1745
                w.pos = f3(ppos);
                w.dir = f3(pdir);
                phase = VH_T_SCATTER;
            }
            if (phase == VH_T_SCATTER && !drew) {
                sample_hg(w.dir, rng, draws, C.phase_g1);                           // :1746 (2 draws)
                drew = true;
                w.inv = rcp3(w.dir);
                if (retry == 0) phase = VH_T_FINISH;                                // depth loop exhausted
                else {
                    retry--;
                    w.mi = false;
                    w.t = 0.0f; w.geo = false; w.obj2 = false; w.wgt = mk3(1.0f);
                    phase = VH_W_TRACK;
                }
            }
            if (phase == VH_T_FINISH) {
                const f3 od = normalize(w.dir);                                     // :1750
                const f3 ob = beta, oL = L;
                const f3 op = length(ob) > 0.9999f ? f3(org0) : w.pos;              // :1753
                float4* dst = reinterpret_cast<float4*>(C.records + ((size_t)kiter * C.n_pixels + pixel));
                st_stream(dst, make_float4(oL.x, oL.y, oL.z, fmin_(w.alpha, 1.0f)));      // :1755
                st_stream(dst + 1, make_float4(ob.x, ob.y, ob.z, depth));
                st_stream(dst + 2, make_float4(op.x, op.y, op.z, __uint_as_float(1u)));
                st_stream(dst + 3, make_float4(od.x, od.y, od.z, 0.0f));
                if (COUNT) {
                    atomicAdd(&P.counters->samples, 1ull);
                    atomicAdd(&P.counters->density_lookups, (unsigned long long)cnt.n_d);
                    atomicAdd(&P.counters->color_lookups, (unsigned long long)cnt.n_c);
                    atomicAdd(&P.counters->emission_lookups, (unsigned long long)cnt.n_e);
                    atomicAdd(&P.counters->tracking_steps, (unsigned long long)cnt.n_steps);
                    atomicAdd(&P.counters->skip_steps, (unsigned long long)cnt.n_skips);
                }
                phase = VH_IDLE;
            }

            if (start_tr) {
                if (tr_begin(C.sph_center, C.sph_radius, K, w, f3(ppos), tr_dir, rcp3(tr_dir))) {
                    tr_next = tr_done;
                    phase = VH_W_TR;
                } else {
                    phase = tr_done;
                }
            }
        }
        }
    }
}

template <bool GENERIC, bool SKYLUT>
static hipError_t launch_vol_variant(const TraceParams& P, int blocks, hipStream_t stream) {
    if (P.addr24) {              // every volume's texel indices fit the 24-bit multiplier, see imul
        if (P.counters) hipLaunchKernelGGL((trace_vol_kernel<GENERIC, GENERIC, GENERIC, true, SKYLUT, true>), dim3(blocks), dim3(256), 0, stream, P);
        else hipLaunchKernelGGL((trace_vol_kernel<GENERIC, GENERIC, GENERIC, false, SKYLUT, true>), dim3(blocks), dim3(256), 0, stream, P);
    } else {
        if (P.counters) hipLaunchKernelGGL((trace_vol_kernel<GENERIC, GENERIC, GENERIC, true, SKYLUT, false>), dim3(blocks), dim3(256), 0, stream, P);
        else hipLaunchKernelGGL((trace_vol_kernel<GENERIC, GENERIC, GENERIC, false, SKYLUT, false>), dim3(blocks), dim3(256), 0, stream, P);
    }
    return hipGetLastError();
}

// Two scene specialisations only (the generic one handles any mix of instances / colour / emission
// grids with the same results: COLOR with no colour grid yields Cd = 1, EMIT is gated by
// emission_scale), times the environment kind.
hipError_t launch_trace_vol(const TraceParams& P, bool multi, bool color, bool emit, int blocks, hipStream_t stream) {
    const bool generic = multi || color || emit;
    const bool skylut = P.environment_type == 0;
    if (!generic && !skylut) return launch_vol_variant<false, false>(P, blocks, stream);
    if (!generic && skylut) return launch_vol_variant<false, true>(P, blocks, stream);
    if (generic && !skylut) return launch_vol_variant<true, false>(P, blocks, stream);
    return launch_vol_variant<true, true>(P, blocks, stream);
}

}  // namespace vpt
