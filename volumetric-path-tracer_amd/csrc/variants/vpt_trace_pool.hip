// vpt_trace_pool.hip -- direct_integrator tracer with the rays in an LDS POOL (gfx950, wave64).
//
// Same per-ray work as vpt_trace.hip (the states of direct_integrator render_kernel.cu:1760, one tracking step of
// `sample` :1556 / `Tr` :1138 / `estimate_emission` :1275 per walk pass, the same operations in the same order: results are
// bit-identical), different binding of rays to lanes.  vpt_trace.hip binds a ray to a lane for its whole life; per pass a
// wave then holds walking rays, rays PARKED until enough of them wait for the (divergent, expensive) transition code, and
// idle lanes -- 29 / 31 / 5 of 64 on BASELINE config 2 -- and every vector instruction of the walk step is issued for the
// 29.  Here a ray is a COLUMN of LDS (47 words: walk state, Philox block, path state) and belongs to no lane.  One
// workgroup per CU owns R = 832 columns; its waves loop over
//     look at the pool -> pick a mode -> claim up to 64 rays of that mode -> load -> work -> store -> release
// with three modes: WALK (one tracking step for 64 walking rays), TRANSITION (the integrator's control flow between walks
// for 64 rays that finished one), FILL (64 free columns get new rays from the global queue).  A wave therefore runs each
// piece of code with (nearly) all lanes on rays that need exactly that piece, and a ray that finished a walk waits for the
// next wave in TRANSITION mode of ANY of the CU's waves instead of for 47 other lanes of its own wave.
//
// Claiming without queues: ray r = lane + 64 j can only be worked on by lane (r mod 64) of whichever wave -- the
// [field][ray] layout then stays bank-conflict free -- so the states of the J = 13 rays of a lane index fit one 32-bit
// word (2 bits each: FREE, WALK-ready, TRANSITION-ready, BUSY).  A lane reads its word, the wave ballots how many lanes could
// get a ray of each kind, picks the mode with the most, and each lane claims with ONE ds_or_rtn (setting BUSY; the returned
// old state says whether the claim won).  Release is one ds_and.  No queue, no counter, no barrier in the loop.
//
// What moved out of LDS to make 832 rays fit in 160 KB: the density history of the fused first walk (HBM, written once
// per look-up of that walk, read once), the primary ray (re-read from the ray record when depth_calculator's distance or
// a replay needs it), the draw counter (derived from the Philox state).
#include "../vpt_trace_direct.h"

namespace vpt {

enum { RS_FREE = 0u, RS_WALK = 1u, RS_TRANS = 2u, RS_BUSY = 3u };
// column fields ([field][ray]); the first PF_HOT are what a WALK pass loads
enum {
    PF_POS = 0, PF_DIR = 3, PF_INV = 6, PF_T = 9, PF_DIST = 10, PF_TRW = 11, PF_ALPHA = 12,
    PF_C0 = 13, PF_O0 = 14, PF_O1 = 15, PF_O2 = 16, PF_O3 = 17, PF_CARRY = 18, PF_PIXEL = 19, PF_FLAGS = 20,
    PF_WGT = 21, PF_LD = 24, PF_GCOT = 27, PF_PPOS = 28, PF_PDIR = 31, PF_ENV = 34, PF_BETA = 37, PF_L = 40,
    PF_DEPTH = 43, PF_SPHF = 44, PF_INTS = 45, PF_CAMDRAWS = 46, PF_CNT = 47      // counting builds: + 5 look-up counters
};
// PF_FLAGS: phase 0..4 | Philox idx 5..7 | has_carry 8 | mi 9 | geo 10 | obj2 11 | n_hist (saturating) 12..15 | kiter 16..21 | gco_obj + 1 22..23
VPT_D uint32_t pack_flags(uint32_t phase, const Rng& g, const Walk& w, uint32_t n_hist, uint32_t kiter, int gco_obj) {
    return phase | (g.idx << 5) | (g.has_carry << 8) | ((w.mi ? 1u : 0u) << 9) | ((w.geo ? 1u : 0u) << 10) | ((w.obj2 ? 1u : 0u) << 11) |
           (min(n_hist, 15u) << 12) | (kiter << 16) | ((uint32_t)(gco_obj + 1) << 22);
}
// PF_INTS: rd 0..7 | vd 8..15 | budget + 1 16..19 | light_index 20..31  (launch_trace_pool refuses scenes that do not fit)
VPT_D uint32_t pack_ints(int rd, int vd, int budget, int light_index) {
    return (uint32_t)rd | ((uint32_t)vd << 8) | ((uint32_t)(budget + 1) << 16) | ((uint32_t)light_index << 20);
}

template <bool MULTI, bool COLOR, bool EMIT, bool COUNT, bool A24>
__global__ __launch_bounds__(768) void trace_pool_kernel(const TraceParams P) {
    constexpr int J = COUNT ? 12 : 13;                 // rays per lane index
    constexpr int R = 64 * J;                          // rays of the pool
    constexpr int NF = COUNT ? PF_CNT + 5 : PF_CNT;
    constexpr uint32_t M55 = 0x55555555u & ((1u << (2 * J)) - 1u);
    typedef LdsF3S<R> F3;
    __shared__ float s_ray[NF * R];
    __shared__ uint32_t s_cand[64];
    __shared__ uint32_t s_occ[20];
    if (threadIdx.x < 19) s_occ[threadIdx.x] = P.occ[threadIdx.x];
    if (threadIdx.x < 64) s_cand[threadIdx.x] = 0u;
    if (EMIT) stage_emission_lut(P);
    __syncthreads();

    const uint32_t total = *P.queue_count;
    const int lane = __lane_id();
    const WalkConst K = make_walk_const(P);
    const f3 sun_dir = ld3(P.sun_dir);
    const uint32_t thr = P.trans_min;                  // fewest lanes a pass starts with while other waves still hold rays
    float* const col0 = s_ray + lane;
    float* const hist0 = P.pool_hist + (size_t)blockIdx.x * (size_t)(VPT_HIST_CAP * R) + lane;
    uint32_t* const candp = &s_cand[lane];
    bool more = true;
    uint32_t chunk_next = 0, chunk_end = 0, chunk_base = 0;
    uint32_t qi0 = 0, qi1 = 0, qi2 = 0, qi3 = 0;
    uint32_t rot = (threadIdx.x >> 6) % (uint32_t)J;   // where this wave starts looking among a lane's J rays
    uint32_t waited = 0;
    int no_retry = 0;
    uint32_t draws = 0;                                // (only counted; nothing reads it in this integrator)

    for (;;) {
        // ==== look at the pool, pick a mode ================================================================
        const uint32_t word = __hip_atomic_load(candp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint32_t lo = word & M55, hi = (word >> 1) & M55;
        const uint32_t mW = lo & ~hi, mT = hi & ~lo, mF = ~(lo | hi) & M55, mB = lo & hi;
        const bool can_fill = chunk_next != chunk_end || more;
        const uint32_t cW = (uint32_t)__popcll(__ballot(mW != 0u)), cT = (uint32_t)__popcll(__ballot(mT != 0u));
        const uint32_t cF = can_fill ? (uint32_t)__popcll(__ballot(mF != 0u)) : 0u;
        uint32_t want, best, m;
        if (cT >= cW && cT >= cF) { want = RS_TRANS; best = cT; m = mT; }
        else if (cW >= cF) { want = RS_WALK; best = cW; m = mW; }
        else { want = RS_FREE; best = cF; m = mF; }
        if (best == 0u) {
            if (!can_fill && !__any(word != 0u)) break;          // every ray of the pool is free and the queue is drained
            __builtin_amdgcn_s_sleep(8);
            continue;
        }
        if (best < thr && waited < 64u && __any(mB != 0u)) {      // few takers now, and other waves are about to release more
            waited++;
            __builtin_amdgcn_s_sleep(2);
            continue;
        }
        waited = 0;
        // ==== claim: one ds_or_rtn per lane ================================================================
        int j = -1;
        if (m != 0u) {
            const uint32_t from = m & (0xffffffffu << (2u * rot));
            const int jj = (__ffs((int)(from != 0u ? from : m)) - 1) >> 1;
            const uint32_t old = atomicOr(candp, 3u << (2 * jj));
            const uint32_t st = (old >> (2 * jj)) & 3u;
            if (st == want) j = jj;
            else if (st != RS_BUSY) atomicAnd(candp, ~((3u ^ st) << (2 * jj)));     // it moved on meanwhile: put its state back
        }
        rot = rot + 1u == (uint32_t)J ? 0u : rot + 1u;
        // (no fence: the column's address depends on the claim's result, and the LDS executes a wave's DS operations in order)
        asm volatile("" ::: "memory");
        bool active = j >= 0;
        float* const col = col0 + 64 * max(j, 0);
        float* const hist = hist0 + 64 * max(j, 0);
        uint32_t* const coli = reinterpret_cast<uint32_t*>(col);

        // ---- the ray, in registers --------------------------------------------------------------------
        Walk w;
        Rng rng;
        uint32_t pixel = 0, kiter = 0, phase = PH_IDLE, n_hist = 0;
        int gco_obj = -1;
        float gco_t = 0.0f;
        WalkCounts cnt;
        cnt.n_d = cnt.n_c = cnt.n_e = cnt.n_steps = cnt.n_skips = 0;
        w.pos = w.dir = w.inv = mk3(0.0f);
        w.t = w.distance = 0.0f; w.trw = 1.0f; w.alpha = 0.0f;
        w.wgt = mk3(1.0f); w.Ld = mk3(0.0f);
        w.mi = w.geo = w.obj2 = false;
        rng.c0 = rng.o0 = rng.o1 = rng.o2 = rng.o3 = rng.idx = rng.carry = rng.has_carry = 0u;
        const F3 c_pos = {col + PF_POS * R}, c_dir = {col + PF_DIR * R}, c_inv = {col + PF_INV * R}, c_wgt = {col + PF_WGT * R}, c_Ld = {col + PF_LD * R};
        const F3 ppos = {col + PF_PPOS * R}, pdir = {col + PF_PDIR * R}, env_pos = {col + PF_ENV * R}, beta = {col + PF_BETA * R}, L = {col + PF_L * R};
        const LdsF depth = {col + PF_DEPTH * R}, sph_factor = {col + PF_SPHF * R};

        if (want != RS_FREE && active) {
            w.pos = f3(c_pos); w.dir = f3(c_dir); w.inv = f3(c_inv);
            w.t = col[PF_T * R]; w.distance = col[PF_DIST * R]; w.trw = col[PF_TRW * R]; w.alpha = col[PF_ALPHA * R];
            rng.c0 = coli[PF_C0 * R]; rng.o0 = coli[PF_O0 * R]; rng.o1 = coli[PF_O1 * R]; rng.o2 = coli[PF_O2 * R]; rng.o3 = coli[PF_O3 * R];
            rng.carry = coli[PF_CARRY * R];
            pixel = coli[PF_PIXEL * R];
            const uint32_t fl = coli[PF_FLAGS * R];
            phase = fl & 31u; rng.idx = (fl >> 5) & 7u; rng.has_carry = (fl >> 8) & 1u;
            w.mi = (fl >> 9) & 1u; w.geo = (fl >> 10) & 1u; w.obj2 = (fl >> 11) & 1u;
            n_hist = (fl >> 12) & 15u; kiter = (fl >> 16) & 63u; gco_obj = (int)((fl >> 22) & 3u) - 1;
            if (EMIT || want == RS_TRANS) w.Ld = f3(c_Ld);
            if (want == RS_TRANS) { w.wgt = f3(c_wgt); gco_t = col[PF_GCOT * R]; }
            if (COUNT) {
                cnt.n_d = coli[(PF_CNT + 0) * R]; cnt.n_c = coli[(PF_CNT + 1) * R]; cnt.n_e = coli[(PF_CNT + 2) * R];
                cnt.n_steps = coli[(PF_CNT + 3) * R]; cnt.n_skips = coli[(PF_CNT + 4) * R];
            }
        }
        if (COUNT && lane == 0) atomicAdd(&P.counters->sched[0], 1ull);

        if (want == RS_WALK) {
            // ==== WALK: one tracking step for every claimed ray ============================================
            if (COUNT) {
                const unsigned long long am = __ballot(active);
                if (lane == 0) {
                    atomicAdd(&P.counters->sched[1], (unsigned long long)__popcll(am));
                    atomicAdd(&P.counters->sched[3], (unsigned long long)(64 - __popcll(am)));
                }
            }
            bool collided = false;
            const uint32_t n_hist_in = n_hist;
            if (active) {
                rng_top_up(rng, pixel);
                const int kind = phase <= PH_W_TRACK ? WALK_SAMPLE : (phase == PH_W_EMIT ? WALK_EMIT : WALK_TR);
                Pending no_pd;
                const bool done = walk_step<MULTI, COLOR, EMIT, COUNT, EMIT, A24, false, R>(P, s_occ, K, kind, phase == PH_W_FIRST, hist, n_hist, w, rng, draws, cnt,
                                                                                        no_retry, false, no_pd) == WALK_DONE;
                if (done) {
                    collided = w.mi;
                    if (phase == PH_W_FIRST) phase = PH_T_FIRST_DONE;
                    else if (phase == PH_W_TRACK) phase = PH_T_TRACK_DONE;
                    else if (phase == PH_W_EMIT) phase = PH_T_EMIT_DONE;
                    else {
                        w.trw = tr_end(K, w);
                        phase = (phase == PH_W_SUN) ? PH_T_SUN_DONE : (phase == PH_W_PL ? PH_T_PL_DONE : PH_T_SPH_DONE);
                    }
                }
                // what a step changes
                c_pos = w.pos;
                col[PF_T * R] = w.t; col[PF_DIST * R] = w.distance; col[PF_TRW * R] = w.trw; col[PF_ALPHA * R] = w.alpha;
                coli[PF_C0 * R] = rng.c0; coli[PF_O0 * R] = rng.o0; coli[PF_O1 * R] = rng.o1; coli[PF_O2 * R] = rng.o2; coli[PF_O3 * R] = rng.o3;
                coli[PF_CARRY * R] = rng.carry;
                coli[PF_FLAGS * R] = pack_flags(phase, rng, w, n_hist, kiter, gco_obj);
                if (EMIT) c_Ld = w.Ld;
                if (COUNT) {
                    coli[(PF_CNT + 0) * R] = cnt.n_d; coli[(PF_CNT + 1) * R] = cnt.n_c; coli[(PF_CNT + 2) * R] = cnt.n_e;
                    coli[(PF_CNT + 3) * R] = cnt.n_steps; coli[(PF_CNT + 4) * R] = cnt.n_skips;
                }
            }
            if (__any(collided)) {
                if (collided) c_wgt = w.wgt;                    // sample()'s return value, read by TRACK_DONE
            }
            // Release = one ds_and behind the column's ds_writes (same wave: executed in order).  The density history is the one
            // thing a later pass -- of any wave of this CU -- reads from HBM: a pass that appended to it waits for its stores.
            if (__any(active && n_hist != n_hist_in)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("" ::: "memory");
            if (active) atomicAnd(candp, ~((3u ^ (phase >= PH_T_FIRST ? RS_TRANS : RS_WALK)) << (2 * j)));
            continue;
        }

        int rd = 0, vd = 0, budget = 0, light_index = 0;
        uint32_t slot = 0;
        if (want == RS_FREE) {
            // ==== FILL: new rays from the compacted queue into free columns ================================
            // the wave owns a chunk of VPT_CHUNK queue entries at a time (one global atomic per 256 rays), 4 entries per lane
            if (chunk_next == chunk_end && more) {
                claim_chunk<1>(P, total, lane, 0, chunk_next, chunk_end, more);
                chunk_base = chunk_next;
                qi0 = chunk_base + (uint32_t)lane < chunk_end ? P.queue[chunk_base + (uint32_t)lane] : 0u;
                qi1 = chunk_base + 64u + (uint32_t)lane < chunk_end ? P.queue[chunk_base + 64u + (uint32_t)lane] : 0u;
                qi2 = chunk_base + 128u + (uint32_t)lane < chunk_end ? P.queue[chunk_base + 128u + (uint32_t)lane] : 0u;
                qi3 = chunk_base + 192u + (uint32_t)lane < chunk_end ? P.queue[chunk_base + 192u + (uint32_t)lane] : 0u;
            }
            const unsigned long long am = __ballot(active);
            const uint32_t avail = chunk_end - chunk_next;
            const uint32_t first = chunk_next;
            chunk_next += min((uint32_t)__popcll(am), avail);
            const uint32_t rank = (uint32_t)__popcll(am & ((1ull << lane) - 1ull));
            const uint32_t rel = first + rank - chunk_base;
            const int src_lane = (int)(rel & 63u);
            const uint32_t e0 = __shfl(qi0, src_lane), e1 = __shfl(qi1, src_lane), e2 = __shfl(qi2, src_lane), e3 = __shfl(qi3, src_lane);
            if (active && rank >= avail) {                       // the chunk ran out: the column stays free
                atomicAnd(candp, ~(3u << (2 * j)));
                active = false;
            }
            if (COUNT) {
                const unsigned long long fm = __ballot(active);
                if (lane == 0) atomicAdd(&P.counters->sched[3], (unsigned long long)(64 - __popcll(fm)));
            }
            if (active) {
                const uint32_t wordi = rel >> 6;
                slot = wordi == 0u ? e0 : (wordi == 1u ? e1 : (wordi == 2u ? e2 : e3));
                split_slot(P, slot, kiter, pixel);
                const uint32_t iteration = P.iter_begin + kiter * P.iter_stride;
                const float4* src = reinterpret_cast<const float4*>(P.records + slot);
                const float4 q0 = src[0], q1 = src[1], q2 = src[2], q3 = src[3];
                // the ray record raygen wrote (vpt_trace.hip: raygen_kernel); obj word: bit 7 = (q0.w, q3.z, q3.w) is the position raygen's
                // empty-node pushes reached, bits 8.. = how many pushes that took
                const uint32_t objw = __float_as_uint(q1.w);
                const bool advanced = (objw & 0x80u) != 0u;
                const f3 origin = mk3(q0.x, q0.y, q0.z);
                gco_t = q0.w;
                gco_obj = (int)(objw & 0x7fu);
                rng.o0 = __float_as_uint(q2.x); rng.o1 = __float_as_uint(q2.y); rng.o2 = __float_as_uint(q2.z); rng.o3 = __float_as_uint(q2.w);
                rng.c0 = __float_as_uint(q3.x);
                rng.idx = __float_as_uint(q3.y);
                rng.carry = 0u; rng.has_carry = 0u;
                depth = advanced ? 0.0f : q3.z;
                coli[PF_CAMDRAWS * R] = rng.c0 * 4u + rng.idx - iteration * 4096u;        // draws consumed by camera::get_ray
                // depth_calculator :1859-1889 and direct_integrator :1772-1785 start from the same ray with the same rng copy
                w.alpha = 0.0f;
                env_pos = origin;
                w.pos = origin;
                w.dir = mk3(q1.x, q1.y, q1.z);
                w.inv = rcp3(w.dir);
                L = mk3(0.0f);
                beta = mk3(1.0f);
                w.mi = false;
                rd = 1; vd = 0;
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
        } else {
            // ==== TRANSITION: integrator control flow between walks (vpt_trace.hip's states, same order) ===
            if (active) {
                const uint32_t iw = coli[PF_INTS * R];
                rd = (int)(iw & 255u); vd = (int)((iw >> 8) & 255u); budget = (int)((iw >> 16) & 15u) - 1; light_index = (int)(iw >> 20);
                slot = kiter * P.n_pixels + pixel;
            }
            if (COUNT) {
                const unsigned long long am = __ballot(active);
                if (lane == 0) {
                    atomicAdd(&P.counters->sched[2], (unsigned long long)__popcll(am));
                    atomicAdd(&P.counters->sched[3], (unsigned long long)(64 - __popcll(am)));
                    atomicAdd(&P.counters->sched[4], 1ull);
                }
            }
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
                f3 tr_dir = mk3(0.0f);

                if (phase == PH_T_FIRST_DONE) {
                    // the walk just finished IS depth_calculator's walk (:1879-1881) ...
                    if (w.mi) {
                        // (the primary ray's origin: the camera origin behind a closed lens, else the ray record still holds it)
                        f3 o0 = ld3(P.cam.origin);
                        if (P.cam.lens_radius != 0.0f) { const float4 q0 = *reinterpret_cast<const float4*>(P.records + slot); o0 = mk3(q0.x, q0.y, q0.z); }
                        depth = length(o0 - w.pos);
                    } else {
                        depth = .0f;
                    }
                    // ... and direct_integrator's first sample() call (:1789), which would add the same densities to Alpha a second time (:1670)
                    if (w.alpha < 1.0f) {
                        if (n_hist > VPT_HIST_CAP) {
                            phase = PH_T_REPLAY;
                        } else {
                            for (uint32_t i = 0; i < n_hist; ++i)
                                if (w.alpha < 1.0f) w.alpha += __hip_atomic_load(hist + i * R, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (not through the L1)
                        }
                    }
                    if (COUNT && phase == PH_T_FIRST_DONE) {
                        cnt.n_d += cnt.n_d; cnt.n_c += cnt.n_c; cnt.n_steps += cnt.n_steps; cnt.n_skips += cnt.n_skips;
                    }
                    if (phase == PH_T_FIRST_DONE) phase = PH_T_TRACK_DONE;
                }
                if (phase == PH_T_REPLAY) {
                    // history overflow (long walk through thin medium): replay the integrator's first walk for real, from the primary
                    // ray (still in the ray record) and the post-camera rng state
                    const uint32_t iteration = P.iter_begin + kiter * P.iter_stride;
                    const uint32_t cam_draws = coli[PF_CAMDRAWS * R];
                    rng_init(rng, pixel, iteration * 4096u + cam_draws);
                    const float4* src = reinterpret_cast<const float4*>(P.records + slot);
                    const float4 q0 = src[0], q1 = src[1];
                    w.pos = mk3(q0.x, q0.y, q0.z);
                    w.dir = mk3(q1.x, q1.y, q1.z);
                    w.inv = rcp3(w.dir);
                    w.mi = false;
                    rd = 1;
                    gco_obj = -1;
                    phase = PH_T_OUTER_TOP;
                }
                if (phase == PH_T_TRACK_DONE) {
                    // :1789-1796
                    beta *= w.wgt;
                    const bool brk = is_black(f3(beta)) || w.obj2;
                    if (!brk && w.mi) {
                        sample_hg(w.dir, rng, draws, P.phase_g1);
                        w.inv = rcp3(w.dir);
                    }
                    gco_obj = -1;
                    vd++;
                    if (!brk && vd <= P.volume_depth) {
                        w.mi = false;
                        w.t = 0.0f; w.geo = false; w.obj2 = false; w.wgt = mk3(1.0f);
                        phase = PH_W_TRACK;
                    } else if (w.mi) {
                        // estimate_sun :1478-1516
                        ppos = w.pos;
                        pdir = w.dir;
                        start_tr = true; tr_dir = sun_dir; tr_walk_phase = PH_W_SUN; tr_done_phase = PH_T_SUN_DONE;
                    } else {
                        phase = PH_T_OUTER_SECOND;
                    }
                } else if (phase == PH_T_SUN_DONE) {
                    const float cos_theta = dot(f3(pdir), sun_dir);
                    const float phase_pdf = henyey_greenstein(cos_theta, P.phase_g1);
                    const f3 Lsun = mk3(w.trw) * phase_pdf;
                    L += (Lsun * ld3(P.sun_color) * P.sun_mult) * f3(beta);             // :1514, :1798
                    if (P.num_lights > 0) {
                        budget = 10;                                                    // :1459
                        w.Ld = mk3(0.0f);
                        phase = PH_T_PL_NEXT;
                    } else {
                        phase = PH_T_EMIT_CHECK;
                    }
                } else if (phase == PH_T_PL_DONE) {
                    if (budget < P.num_lights) {
                        // point_light::Le, light.h:104-121
                        const DPointLight& lt = P.lights[light_index];
                        const f3 lp = ld3(lt.pos);
                        const f3 pp = ppos;
                        const f3 wi = normalize(lp - pp);
                        const float cos_theta = dot(f3(pdir), wi);
                        const float phase_pdf = henyey_greenstein(cos_theta, P.phase_g1);
                        const float sqr_dist = length(lp * lp - pp * pp);
                        const float falloff = 1 / sqr_dist;
                        w.Ld += ld3(lt.color) * lt.power * mk3(w.trw) * phase_pdf * falloff;
                    }
                    budget--;
                    if (budget >= 0) phase = PH_T_PL_NEXT;
                    else {
                        L += w.Ld * f3(beta);                                           // :1799
                        phase = PH_T_EMIT_CHECK;
                    }
                }
                if (phase == PH_T_PL_NEXT) {
                    // estimate_point_light :1461-1466 (1 draw)
                    int li = (int)floorf(rnd(rng, draws) * P.num_lights);
                    if (li > P.num_lights - 1) li = P.num_lights - 1;                    // rand()==1.0f guard
                    light_index = li;
                    const DPointLight& lt = P.lights[li];
                    start_tr = true; tr_dir = normalize(ld3(lt.pos) - f3(ppos)); tr_walk_phase = PH_W_PL; tr_done_phase = PH_T_PL_DONE;
                } else if (phase == PH_T_EMIT_CHECK || phase == PH_T_EMIT_DONE || phase == PH_T_SPH_DONE) {
                    if (phase == PH_T_EMIT_DONE) L += w.Ld;                             // :1803
                    if (phase == PH_T_SPH_DONE) L += ld3(P.sun_color) * P.sun_mult * mk3(w.trw) * (float)sph_factor * f3(beta);  // :1832
                    w.pos = f3(ppos);
                    w.dir = f3(pdir);
                    w.inv = rcp3(w.dir);
                    gco_obj = -1;
                    if (phase == PH_T_EMIT_CHECK && EMIT && P.emission_scale > 0) {     // :1802 (mi is true here)
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
                if (phase == PH_T_OUTER_SECOND) {
                    if (gco_obj < 0) gco_obj = closest_object(P, w.pos, w.dir, w.inv, gco_t); // :1806
                    if (gco_obj == 2) {
                        // sphere bounce :1809-1833 (2 draws)
                        w.pos += w.dir * gco_t;
                        const f3 normal = normalize((w.pos - ld3(P.sph_center)) / P.sph_radius);
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
                        w.dir = lerp3(ref, hemisphere_dir, P.sph_roughness);
                        w.pos += normal * VPT_EPS;
                        beta *= ld3(P.sph_color);
                        sph_factor = fmax_(dot(sun_dir, normal), .0f);
                        ppos = w.pos;
                        pdir = w.dir;
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
                        if (gco_obj < 0) gco_obj = closest_object(P, w.pos, w.dir, w.inv, gco_t);   // :1782
                        if (gco_obj == 1) {
                            w.pos += w.dir * (gco_t + VPT_EPS);
                            gco_obj = -1;
                            vd = 1;
                            w.mi = false;
                            w.t = 0.0f; w.geo = false; w.obj2 = false; w.wgt = mk3(1.0f);
                            // a ray that sat INSIDE the box is moved to the box's far side here, so the walk it starts usually finds itself
                            // outside the octree at once (:1606): resolved right here (see vpt_trace.hip), exact
                            f3 nmin, nmax;
                            int leaf;
                            const OccTop occ_top = {s_occ[0], s_occ[1], s_occ[2]};
                            if (locate(P, s_occ, occ_top, w.pos, nmin, nmax, leaf) == LOC_OUTSIDE) {
                                gco_obj = closest_object(P, w.pos, w.dir, w.inv, gco_t);
                                if (gco_obj == 0) {
                                    rd++;
                                    phase = PH_T_FINISH;
                                } else {
                                    vd = P.volume_depth + 1;
                                    phase = PH_T_OUTER_SECOND;
                                }
                            } else {
                                phase = PH_W_TRACK;
                            }
                        } else if (gco_obj == 0) {
                            // nothing ahead: this and every later iteration is a no-op -> finish (exact)
                            phase = PH_T_FINISH;
                        } else {
                            phase = PH_T_OUTER_SECOND;   // sphere is closest: handled next pass
                        }
                    }
                }
                if (phase == PH_T_FINISH) {
                    const f3 od = w.dir, oL = L, ob = beta, oe = env_pos;
                    float4* dst = reinterpret_cast<float4*>(P.records + slot);
                    dst[0] = make_float4(oL.x, oL.y, oL.z, fmin_(w.alpha, 1.0f));      // tr = fminf(tr, 1) :1854
                    dst[1] = make_float4(ob.x, ob.y, ob.z, depth);
                    dst[2] = make_float4(oe.x, oe.y, oe.z, __uint_as_float(1u));
                    dst[3] = make_float4(od.x, od.y, od.z, 0.0f);
                    if (COUNT) {
                        atomicAdd(&P.counters->samples, 1ull);
                        if (cnt.n_steps == 0u) atomicAdd(&P.counters->coh[6], 1ull);
                        atomicAdd(&P.counters->density_lookups, (unsigned long long)cnt.n_d);
                        atomicAdd(&P.counters->color_lookups, (unsigned long long)cnt.n_c);
                        atomicAdd(&P.counters->emission_lookups, (unsigned long long)cnt.n_e);
                        atomicAdd(&P.counters->tracking_steps, (unsigned long long)cnt.n_steps);
                        atomicAdd(&P.counters->skip_steps, (unsigned long long)cnt.n_skips);
                    }
                    phase = PH_IDLE;
                }
                // ---- Tr prologue :1153-1167 (shared by sun / point-light / sphere shadow rays) ---
                if (start_tr) phase = tr_begin(P, K, w, f3(ppos), tr_dir) ? tr_walk_phase : tr_done_phase;
            }
        }
        // ==== store the whole ray, release its column =========================================================
        if (active && phase != PH_IDLE) {
            c_pos = w.pos; c_dir = w.dir; c_inv = w.inv;
            col[PF_T * R] = w.t; col[PF_DIST * R] = w.distance; col[PF_TRW * R] = w.trw; col[PF_ALPHA * R] = w.alpha;
            coli[PF_C0 * R] = rng.c0; coli[PF_O0 * R] = rng.o0; coli[PF_O1 * R] = rng.o1; coli[PF_O2 * R] = rng.o2; coli[PF_O3 * R] = rng.o3;
            coli[PF_CARRY * R] = rng.carry;
            coli[PF_PIXEL * R] = pixel;
            coli[PF_FLAGS * R] = pack_flags(phase, rng, w, n_hist, kiter, gco_obj);
            c_wgt = w.wgt; c_Ld = w.Ld;
            col[PF_GCOT * R] = gco_t;
            coli[PF_INTS * R] = pack_ints(rd, vd, budget, light_index);
            if (COUNT) {
                coli[(PF_CNT + 0) * R] = cnt.n_d; coli[(PF_CNT + 1) * R] = cnt.n_c; coli[(PF_CNT + 2) * R] = cnt.n_e;
                coli[(PF_CNT + 3) * R] = cnt.n_steps; coli[(PF_CNT + 4) * R] = cnt.n_skips;
            }
        }
        asm volatile("" ::: "memory");                 // the ds_and below stays behind the column's ds_writes; nothing else is read by others
        if (active) {
            const uint32_t ns = phase == PH_IDLE ? RS_FREE : (phase >= PH_T_FIRST ? RS_TRANS : RS_WALK);
            atomicAnd(candp, ~((3u ^ ns) << (2 * j)));
        }
    }
}

// ---- launcher -------------------------------------------------------------------------------
template <bool MULTI, bool COLOR, bool EMIT>
static hipError_t launch_pool_variant(const TraceParams& P, int blocks, int threads, hipStream_t stream) {
    if (P.addr24) {
        if (P.counters) hipLaunchKernelGGL((trace_pool_kernel<MULTI, COLOR, EMIT, true, true>), dim3(blocks), dim3(threads), 0, stream, P);
        else hipLaunchKernelGGL((trace_pool_kernel<MULTI, COLOR, EMIT, false, true>), dim3(blocks), dim3(threads), 0, stream, P);
    } else {
        if (P.counters) hipLaunchKernelGGL((trace_pool_kernel<MULTI, COLOR, EMIT, true, false>), dim3(blocks), dim3(threads), 0, stream, P);
        else hipLaunchKernelGGL((trace_pool_kernel<MULTI, COLOR, EMIT, false, false>), dim3(blocks), dim3(threads), 0, stream, P);
    }
    return hipGetLastError();
}

// floats of TraceParams::pool_hist per workgroup
size_t trace_pool_hist_floats_per_block() { return (size_t)VPT_HIST_CAP * 64u * 13u; }
// the packed per-ray integers (PF_INTS) bound what the pool tracer takes; anything else goes to the lane-bound tracer
bool trace_pool_supports(const TraceParams& P) { return P.ray_depth < 254 && P.volume_depth < 254 && P.num_lights < 4096; }

hipError_t launch_trace_pool(const TraceParams& P, bool multi, bool color, bool emit, int blocks, int threads, hipStream_t stream) {
    if (!multi && !color && !emit) return launch_pool_variant<false, false, false>(P, blocks, threads, stream);
    if (!multi && !color && emit) return launch_pool_variant<false, false, true>(P, blocks, threads, stream);
    if (!multi && color && !emit) return launch_pool_variant<false, true, false>(P, blocks, threads, stream);
    if (!multi && color && emit) return launch_pool_variant<false, true, true>(P, blocks, threads, stream);
    if (multi && !color && !emit) return launch_pool_variant<true, false, false>(P, blocks, threads, stream);
    if (multi && !color && emit) return launch_pool_variant<true, false, true>(P, blocks, threads, stream);
    if (multi && color && !emit) return launch_pool_variant<true, true, false>(P, blocks, threads, stream);
    return launch_pool_variant<true, true, true>(P, blocks, threads, stream);
}

}  // namespace vpt
