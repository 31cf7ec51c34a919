// vpt_tail.hip -- last stage of the hot path, fused: the environment tail of the integrators
// (direct_integrator render_kernel.cu:1838-1850, vol_integrator :1752: Bruneton sky
// `sample_atmosphere` :839-895 or the lat-long HDRI) for every path record of a batch, followed by
// what volume_rt_kernel does with a sample value (:2263-2316): NaN guard, running means, tonemap.
//
// The environment value is VALUE-ONLY arithmetic: nothing downstream branches on it that feeds a
// random walk, the result is added to L once.  The reference evaluates it with `--use_fast_math`
// (source/CMakeLists.txt:133); this translation unit is likewise built with approximate fp32
// divide/sqrt and FMA contraction and uses the hardware exp/log/pow (see build.py) -- unlike the
// strict-arithmetic tracer.  Tolerance against the oracle: tests/test_gpu_atmosphere.py.
#include <hip/hip_runtime.h>

#include "vpt_sky.h"
#include "vpt_cull.h"

namespace vpt {

// IEEE binary32 divide regardless of this translation unit's relaxed flags (the running means of the
// reference, :2278-2287, are compared bit for bit): the quotient is formed in binary64, whose 53 bits
// (>= 2*24+2) make the second rounding innocuous, so (float)((double)a / b) IS the correctly rounded
// binary32 quotient.  __fdiv_rn is just `a / b` in HIP and follows the approximate-divide flag.
VPT_D float div1_rn(float a, float b) { return (float)((double)a / (double)b); }
VPT_D f3 div_rn(f3 a, float b) { return mk3(div1_rn(a.x, b), div1_rn(a.y, b), div1_rn(a.z, b)); }
// The same correctly rounded quotients a / n for the running means' divisor n = (float)(iteration + 1) -- an integer-valued binary32, the
// same for every pixel of an iteration -- by ONE binary64 reciprocal per iteration, rn = RN64(1 / n), made on the host (ResolveParams::rcp_n):
// RN32(RN64(a * rn)) == RN32(a / n) for every binary32 a whenever n < 2^27.  Proof: a = A 2^e, |A| < 2^24; a rounding boundary of binary32
// next to q = a / n is M 2^f with M an odd integer (|M| < 2^25) and e >= f, so q - M 2^f = 2^f (A 2^(e-f) - M n) / n is 0 or at least 2^f / n
// in magnitude, i.e. >= 2^-25 / n relative to q; it is 0 only if n is a power of two (M n odd times a power of two cannot equal an integer of
// 24 bits otherwise), and then a * rn is exact.  The computed product is within 2^-52 of q relatively (two binary64 roundings), which is
// below 2^-25 / n for n < 2^27: it lies on q's side of every boundary.  (Subnormal quotients: coarser boundaries, same argument.)  Four
// binary64 divisions per sample become four multiplications; host entries of 0 (n >= 2^27) keep the division.
VPT_D float mul1_rn(float a, double rn) { return (float)((double)a * rn); }
VPT_D f3 mul_rn(f3 a, double rn) { return mk3(mul1_rn(a.x, rn), mul1_rn(a.y, rn), mul1_rn(a.z, rn)); }
VPT_D f3 rtt_and_odt_fit(f3 v) {                                                       // :2208
    f3 a = v * (v + 0.0245786f) - 0.000090537f;
    f3 b = v * (0.983729f * v + 0.4329510f) + 0.238081f;
    return mk3(__fdiv_rn(a.x, b.x), __fdiv_rn(a.y, b.y), __fdiv_rn(a.z, b.z));
}

// ACES fit + gamma + 8-bit pack of a mean (:2292-2316); returns the linear display value (raw.xyz)
VPT_D f3 tonemap(f3 acc, float exposure_scale, unsigned int& packed) {
    f3 val = mk3(0.59719f * acc.x + 0.35458f * acc.y + 0.04823f * acc.z,
                 0.07600f * acc.x + 0.90834f * acc.y + 0.01566f * acc.z,
                 0.02840f * acc.x + 0.13383f * acc.y + 0.83777f * acc.z);
    val = rtt_and_odt_fit(val);
    val = mk3(1.60475f * val.x + -0.53108f * val.y + -0.07367f * val.z,
              -0.10208f * val.x + 1.10813f * val.y + -0.00605f * val.z,
              -0.00327f * val.x + -0.07276f * val.y + 1.07602f * val.z) * exposure_scale;
    const float ig = (float)(1.0 / 2.2);
    const unsigned int r = (unsigned int)(255.0f * fmin_(powf(fmax_(val.x, 0.0f), ig), 1.0f));
    const unsigned int g = (unsigned int)(255.0f * fmin_(powf(fmax_(val.y, 0.0f), ig), 1.0f));
    const unsigned int b = (unsigned int)(255.0f * fmin_(powf(fmax_(val.z, 0.0f), ig), 1.0f));
    packed = 0xff000000u | (r << 16) | (g << 8) | b;
    return val;
}

// stage 3 (last): one thread per PIXEL walks the batch's path records in iteration order:
//   environment tail of the integrator (direct :1838-1850 / vol :1752)  -> sample value
//   NaN/Inf guard (:2263-2264), viz_dof tint (:2266-2274), running means (:2278-2287)
// and, once per batch, ACES tonemap + gamma + 8-bit pack + raw buffer (:2292-2316).
// Adjacent lanes read adjacent 64-byte records (4 KiB contiguous per wave and iteration); the sky
// evaluation is the dominant cost and runs at full lane utilisation.
#ifndef VPT_TAIL_WAVES_PER_EU
#define VPT_TAIL_WAVES_PER_EU 4
#endif
// what volume_rt_kernel does with a sample value (:2263-2287): NaN guard, viz_dof tint, running means -- and with the means at the end of
// a launch (:2292-2316).  One pixel's state; shared by the two tail kernels so that they cannot drift apart.
struct RunningMeans {
    f3 acc, cst;
    float dep, tr_last;
    VPT_D void add(const ResolveParams& R, f3 value, float tr, float depth, uint32_t iteration, uint32_t local_it, double rn) {
        // :2263-2264
        if (isnan(value.x) || isnan(value.y) || isnan(value.z) || isinf(value.x) || isinf(value.y) || isinf(value.z)) value = acc;
        if (isnan(tr) || isinf(tr)) tr = 1.0f;
        // :2266-2274
        if (R.viz_dof) {
            float aof = clampf(__fdiv_rn(1.0f, R.lens_radius), .0f, 3.402823466e+38F);
            if (depth > (R.focus_dist + aof)) value = lerp3(value, mk3(1, 0, 0), 0.5f);
            if (depth < (R.focus_dist - aof)) value = lerp3(value, mk3(0, 0, 1), 0.5f);
            if (depth > (R.focus_dist - aof) && depth < (R.focus_dist + aof)) value = lerp3(value, mk3(0, 1, 0), 0.5f);
        }
        // :2278-2287 (cost is always BLACK, :2249)
        if (local_it == 0) {
            acc = value;
            cst = mk3(0.0f);
            dep = depth;
        } else if (iteration < R.max_interactions) {
            const float n = (float)(local_it + 1);
            if (rn != 0.0) {
                acc = acc + mul_rn(value - acc, rn);
                // cost is always BLACK: 0 + (0 - 0)/n == +0 exactly (also from -0), so the quotients are skipped then
                if (cst.x == 0.0f && cst.y == 0.0f && cst.z == 0.0f) cst = mk3(0.0f);
                else cst = cst + mul_rn(mk3(0.0f) - cst, rn);
                dep = dep + mul1_rn(depth - dep, rn);
            } else {
                acc = acc + div_rn(value - acc, n);
                if (cst.x == 0.0f && cst.y == 0.0f && cst.z == 0.0f) cst = mk3(0.0f);
                else cst = cst + div_rn(mk3(0.0f) - cst, n);
                dep = dep + div1_rn(depth - dep, n);
            }
        }
        tr_last = tr;
    }
    VPT_D static RunningMeans load(const ResolveParams& R, uint32_t idx) {
        RunningMeans m;
        m.acc = mk3(R.accum[3 * idx], R.accum[3 * idx + 1], R.accum[3 * idx + 2]);
        m.cst = R.cost ? mk3(R.cost[3 * idx], R.cost[3 * idx + 1], R.cost[3 * idx + 2]) : mk3(0.0f);
        m.dep = R.depth ? R.depth[idx] : 0.0f;
        m.tr_last = 0.0f;
        return m;
    }
    VPT_D void store(const ResolveParams& R, uint32_t idx) const {
        R.accum[3 * idx] = acc.x; R.accum[3 * idx + 1] = acc.y; R.accum[3 * idx + 2] = acc.z;
        if (R.cost) { R.cost[3 * idx] = cst.x; R.cost[3 * idx + 1] = cst.y; R.cost[3 * idx + 2] = cst.z; }
        if (R.depth) R.depth[idx] = dep;
        if (R.display || R.raw) {
            unsigned int packed;
            const f3 val = tonemap(acc, R.exposure_scale, packed);
            if (R.display) R.display[idx] = packed;
            if (R.raw) reinterpret_cast<float4*>(R.raw)[idx] = make_float4(val.x, val.y, val.z, tr_last);
        }
    }
};

// LENS: the tables have one variant per binary32 step of r across the lens disc (SkyView); else one, the camera origin's
template <bool LENS>
VPT_D void load_sky_view(const ResolveParams& R, Sky<ResolveParams>& sky) {
    sky.view_multi = LENS;
    if (R.cam_tab_valid) {
        sky.view_r = R.sky_view->r;
        sky.view_mu_s = R.sky_view->mu_s;
        sky.view_k = LENS ? R.sky_view->k : 0;
        if (!LENS) sky.view_vt0 = R.sky_view->tab[0];
    }
}

// HEADS: samples that start no walk arrive as 16-byte heads (+ origins when the lens is open) instead of 64-byte records
// (ResolveParams).  Two instantiations: each keeps its own register budget (the kernel spills at 4 waves per SIMD).
// per-pixel sky patch (ResolveParams::sky_patch): one thread per pixel evaluates the untraced-sample value at the pixel's four
// corners and at its centre, keeps the corners when the bilinear patch reproduces the centre to 1e-3
__global__ __launch_bounds__(256) void sky_patch_kernel(const ResolveParams R, float4* __restrict__ out, unsigned char* __restrict__ never,
                                                        uint32_t* __restrict__ nopatch_list, uint32_t* __restrict__ nopatch_count) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= R.n_pixels) return;
    const uint32_t y = idx / R.width, x = idx - y * R.width;
    Sky<ResolveParams> sky = {R};
    load_sky_view<false>(R, sky);
    const bool use_dir_tab = R.dir_tab != nullptr && R.dir_tab_err[8] != 0u;
    const f3 sun_dir = mk3(R.sun_dir[0], R.sun_dir[1], R.sun_dir[2]);
    const f3 org = mk3(R.cam_origin[0], R.cam_origin[1], R.cam_origin[2]);
    const f3 llc = mk3(R.cam_llc[0], R.cam_llc[1], R.cam_llc[2]), ch = mk3(R.cam_h[0], R.cam_h[1], R.cam_h[2]), cv = mk3(R.cam_v[0], R.cam_v[1], R.cam_v[2]);
    const f3 scale = mk3(R.sky_color[0], R.sky_color[1], R.sky_color[2]) * R.sky_mult;
    const float inv_w = frcp((float)R.width), inv_h = frcp((float)R.height);
    auto ray = [&](float jx, float jy) {                     // camera::get_ray with a closed lens (camera.h:131-136)
        const float u = ((float)x + jx) * inv_w, v = ((float)y + jy) * inv_h;
        return normalize(llc + ch * u + cv * v - org);
    };
    const f3 d00 = ray(0.0f, 0.0f), d11 = ray(1.0f, 1.0f), dc = ray(0.5f, 0.5f);
    int k00, k10, k01, k11, kc;
    const f3 v00 = sky.sample(org, d00, sun_dir, use_dir_tab, &k00) * scale, v10 = sky.sample(org, ray(1.0f, 0.0f), sun_dir, use_dir_tab, &k10) * scale;
    const f3 v01 = sky.sample(org, ray(0.0f, 1.0f), sun_dir, use_dir_tab, &k01) * scale, v11 = sky.sample(org, d11, sun_dir, use_dir_tab, &k11) * scale;
    const f3 vc = sky.sample(org, dc, sun_dir, use_dir_tab, &kc) * scale;
    const f3 pred = (v00 + v10 + v01 + v11) * 0.25f;
    const float dev = fmax_(fmax_(fabsf(pred.x - vc.x), fabsf(pred.y - vc.y)), fabsf(pred.z - vc.z));
    const float ref = fmax_(fmax_(fabsf(vc.x), fabsf(vc.y)), fabsf(vc.z));
    // 1e-3 of the centre value: sample_atmosphere itself is only that smooth -- the reference's binary32 `r^2 mu^2 - r^2 + bottom^2`
    // (vpt_sky.h: ScatteringUvwz) is a staircase in mu with steps of ~1e-4 of the radiance between neighbouring directions, while
    // what the check is for, the horizon or the sun's disc cutting through the pixel, moves the centre by 1e-2 and more
    bool ok = dev <= 1e-3f * ref || (dev == 0.0f && ref == 0.0f);               // (NaNs fail the comparison)
    // the sun's disc can be smaller than a pixel and miss all five probes: keep a pixel's diagonal away from it
    const f3 dd = d00 - d11;
    const float diag = fsqrt(dot(dd, dd));
    const float ang = __builtin_amdgcn_sqrtf(fmax_(2.0f - 2.0f * dot(dc, sun_dir), 0.0f));   // chord ~ angle between pixel centre and sun
    const float disc = fsqrt(fmax_(2.0f - 2.0f * sky.f(AF_COS_SUN), 0.0f));
    if (ang <= disc + diag) ok = false;
    // a ground hit evaluated in full (grazing rays, or no ground table) rounds its ground point to binary32 at earth-radius
    // magnitude and jumps by up to 1-2 % from ray to ray (vpt_sky.h): per sample that is noise which averages out, frozen into a
    // patch corner it would be a bias -- such pixels keep the per-sample evaluation
    if (k00 == 2 || k10 == 2 || k01 == 2 || k11 == 2 || kc == 2) ok = false;
    float4* o = out + 3u * (size_t)idx;
    o[0] = make_float4(ok ? v00.x : __uint_as_float(0x7fc00000u), v00.y, v00.z, v10.x);
    o[1] = make_float4(v10.y, v10.z, v01.x, v01.y);
    o[2] = make_float4(v01.z, v11.x, v11.y, v11.z);
    // the pixels without a usable patch, listed for sky_fix_kernel (ResolveParams::nopatch_list; the count was zeroed by the host)
    if (!ok && nopatch_list) nopatch_list[atomicAdd(nopatch_count, 1u)] = idx;
    // never-traced pixels (ResolveParams::cull_*): the pixel's square [x, x+1] x [y, y+1] against the grown bounds
    if (never) {
        bool nt = ok && R.cull_enabled != 0;
        const float fx = (float)x, fy = (float)y;
        if (fx + 1.0f >= R.cull_rect[0] && fx <= R.cull_rect[2] && fy + 1.0f >= R.cull_rect[1] && fy <= R.cull_rect[3]) {
            // inside the root box's bounds: traced unless no non-empty leaf's bounds touch the pixel's tile (ResolveParams::cull_tiles)
            if (R.cull_tiles == nullptr || R.cull_tiles[(y >> 3) * R.cull_tiles_w + (x >> 3)] != 0) nt = false;
        }
        if (sphere_may_hit(org, dc, diag, R.cull_sph)) nt = false;
        const float la = R.cull_line[0], lb = R.cull_line[1], lc = R.cull_line[2];
        const float ln = la * la + lb * lb;
        if (ln > 0.0f) {
            const float dist = fabsf(la * (fx + 0.5f) + lb * (fy + 0.5f) + lc) * __builtin_amdgcn_rsqf(ln);
            if (!(dist > 4.0f)) nt = false;                                            // half a diagonal + 3 pixels
        } else if (lc == 0.0f) {
            nt = false;
        }
        never[idx] = nt ? 1 : 0;
    }
}
hipError_t launch_sky_patch(const ResolveParams& R, float4* out, unsigned char* never, uint32_t* nopatch_list, uint32_t* nopatch_count, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(nopatch_count, 0, sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(sky_patch_kernel, dim3((R.n_pixels + 255u) / 256u), dim3(256), 0, stream, R, out, never, nopatch_list, nopatch_count);
    return hipGetLastError();
}

// ---- sky dome (ResolveParams::sky_dome): coordinates and look-up in vpt_dome.h ---------------------------------------------------
// one thread per cell (i, j): its four corner nodes and its centre, evaluated from the camera origin; writes node (i, j) and the cell's flag
// LENS: one dome per table variant (SkyView: one per binary32 value of r across the lens disc) -- the sky sees a sample's origin only through
// r and mu_s, so a dome built from the camera origin displaced by the variant's steps of r serves every lens origin of that r
template <bool LENS>
__global__ __launch_bounds__(256) void sky_dome_kernel(const ResolveParams R, const SkyView* view, float4* __restrict__ out) {
    const uint32_t tg = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t cells = (uint32_t)SKY_DOME_NU * (uint32_t)SKY_DOME_NV;
    const uint32_t variant = tg / cells, t = tg - variant * cells;
    const int kv = LENS ? view->k : 0;
    if (variant > 2u * (uint32_t)kv) return;
    const uint32_t j = t / (uint32_t)SKY_DOME_NU, i = t - j * (uint32_t)SKY_DOME_NU;
    Sky<ResolveParams> sky = {R};
    load_sky_view<LENS>(R, sky);
    const bool use_dir_tab = R.dir_tab != nullptr && R.dir_tab_err[8] != 0u;
    const f3 sun_dir = mk3(R.sun_dir[0], R.sun_dir[1], R.sun_dir[2]);
    f3 org = mk3(R.cam_origin[0], R.cam_origin[1], R.cam_origin[2]);
    bool variant_ok = true;
    if (LENS) {
        const f3 ec = mk3(0.0f, -sky.bottom(), 0.0f);
        const f3 up0 = normalize(org - ec);
        org = org + up0 * (__uint_as_float(__float_as_uint(view->r) + variant - (uint32_t)kv) - view->r);
        const f3 p = org - ec;
        const float r = length_rn(p);
        variant_ok = sky.CamVariant(r, dot(p, sun_dir) * frcp(r)) == (int)variant;       // (the displaced origin did land on this variant's r)
    }
    const f3 scale = mk3(R.sky_color[0], R.sky_color[1], R.sky_color[2]) * R.sky_mult;
    const float fi = (float)i, fj = (float)j;
    const f3 d00 = dome_direction(fi, fj);
    int k00 = 0, k10 = 0, k01 = 0, k11 = 0, kc = 0;
    const f3 v00 = sky.sample(org, d00, sun_dir, use_dir_tab, &k00) * scale;
    bool ok = false;
    if (j + 1u < (uint32_t)SKY_DOME_NV) {
        const f3 d11 = dome_direction(fi + 1.0f, fj + 1.0f), dc = dome_direction(fi + 0.5f, fj + 0.5f);
        const f3 v10 = sky.sample(org, dome_direction(fi + 1.0f, fj), sun_dir, use_dir_tab, &k10) * scale;
        const f3 v01 = sky.sample(org, dome_direction(fi, fj + 1.0f), sun_dir, use_dir_tab, &k01) * scale;
        const f3 v11 = sky.sample(org, d11, sun_dir, use_dir_tab, &k11) * scale;
        const f3 vc = sky.sample(org, dc, sun_dir, use_dir_tab, &kc) * scale;
        const f3 pred = (v00 + v10 + v01 + v11) * 0.25f;
        const float dev = fmax_(fmax_(fabsf(pred.x - vc.x), fabsf(pred.y - vc.y)), fabsf(pred.z - vc.z));
        const float ref = fmax_(fmax_(fabsf(vc.x), fabsf(vc.y)), fabsf(vc.z));
        ok = dev <= 1e-3f * ref || (dev == 0.0f && ref == 0.0f);
        const f3 dd = d00 - d11;
        const float diag = fsqrt(dot(dd, dd));
        const float ang = fsqrt(fmax_(2.0f - 2.0f * dot(dc, sun_dir), 0.0f));
        const float disc = fsqrt(fmax_(2.0f - 2.0f * sky.f(AF_COS_SUN), 0.0f));
        if (ang <= disc + diag) ok = false;
        if (k00 == 2 || k10 == 2 || k01 == 2 || k11 == 2 || kc == 2) ok = false;
    }
    out[tg] = make_float4(v00.x, v00.y, v00.z, ok && variant_ok ? 1.0f : 0.0f);
}
size_t sky_dome_bytes(int k) { return sizeof(float4) * (size_t)SKY_DOME_NU * (size_t)SKY_DOME_NV * (size_t)(2 * k + 1); }
hipError_t launch_sky_dome(const ResolveParams& R, const SkyView* view, float4* out, int k, hipStream_t stream) {
    const uint32_t n = (uint32_t)SKY_DOME_NU * (uint32_t)SKY_DOME_NV * (uint32_t)(2 * k + 1);
    if (k > 0) hipLaunchKernelGGL(sky_dome_kernel<true>, dim3((n + 255u) / 256u), dim3(256), 0, stream, R, view, out);
    else hipLaunchKernelGGL(sky_dome_kernel<false>, dim3((n + 255u) / 256u), dim3(256), 0, stream, R, view, out);
    return hipGetLastError();
}
template <bool HEADS, bool LENS>
__global__ __launch_bounds__(256, VPT_TAIL_WAVES_PER_EU) void tail_resolve_kernel(const ResolveParams R) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    // the pixel's sky patch, parked in LDS ([word][thread]); usable: the pixel has one and it passed its check
    __shared__ float s_patch[12 * 256];
    bool patch = false;
    bool never = false;              // no heads exist for this pixel (ResolveParams::never_traced): every sample is untraced, depth 0
    uint32_t bn_idx = 0;
    if (HEADS && !LENS && R.sky_patch != nullptr && idx < R.n_pixels) {
        const float4* pp = R.sky_patch + 3u * (size_t)idx;
        const float4 a = pp[0], b = pp[1], c = pp[2];
        patch = a.x == a.x;
        never = R.never_traced != nullptr && R.never_traced[idx] != 0;
        float* s = s_patch + threadIdx.x;
        s[0] = a.x; s[256] = a.y; s[512] = a.z; s[768] = a.w;
        s[1024] = b.x; s[1280] = b.y; s[1536] = b.z; s[1792] = b.w;
        s[2048] = c.x; s[2304] = c.y; s[2560] = c.z; s[2816] = c.w;
        const uint32_t py = idx / R.width, px = idx - py * R.width;
        bn_idx = (py & 255u) * 256u + (px & 255u);
    }
    if (idx >= R.n_pixels) return;
    const f3 sky_color = mk3(R.sky_color[0], R.sky_color[1], R.sky_color[2]);
    const f3 sun_dir = mk3(R.sun_dir[0], R.sun_dir[1], R.sun_dir[2]);
    Sky<ResolveParams> sky = {R};
    load_sky_view<LENS>(R, sky);
    // the ground table is used only while it passed its build-time checks: interpolation error within the tolerance AND real rays
    // through the full path agreeing (sky_dir_table_verdict_kernel)
    const bool use_dir_tab = R.dir_tab != nullptr && R.dir_tab_err[8] != 0u;

    // floor((iter_begin + k * stride) / stride) = floor(iter_begin / stride) + k: one division per launch, not per sample
    const uint32_t local_it0 = R.iter_begin / R.iter_stride;
    RunningMeans rm = RunningMeans::load(R, idx);
    auto accumulate = [&](f3 value, float tr, float depth, uint32_t iteration, uint32_t local_it, double rn) { rm.add(R, value, tr, depth, iteration, local_it, rn); };
    if (never) {
        // a pixel raygen emitted nothing for (ResolveParams::never_traced): every sample is untraced with depth 0, its value the patch at the
        // sample's jitter -- or WHITE when the sample is not rendered (:2248, :2254).  No head, no record, no sky code: waves over the background
        // run this loop only.
        const float* s = s_patch + threadIdx.x;
        const f3 v00 = mk3(s[0], s[256], s[512]), v10 = mk3(s[768], s[1024], s[1280]);
        const f3 v01 = mk3(s[1536], s[1792], s[2048]), v11 = mk3(s[2304], s[2560], s[2816]);
        for (uint32_t k = 0; k < R.iter_count; ++k) {
            const uint32_t iteration = R.iter_begin + k * R.iter_stride;
            f3 value = mk3(1.0f);
            if (iteration < R.max_interactions && R.render) {
                const float2 j = R.blue_noise[(size_t)k * 65536u + bn_idx];
                value = flerp3(flerp3(v00, v10, j.x), flerp3(v01, v11, j.x), j.y);
            }
            accumulate(value, 0.0f, 0.0f, iteration, local_it0 + k, R.rcp_n[k]);
        }
    } else {
    // the next iteration's head is requested while the current sample is evaluated (a streaming read from HBM)
    float4 h_next = make_float4(0.0f, 0.0f, 0.0f, -1.0f);
    if (HEADS) h_next = ld_stream(R.heads + idx);
    for (uint32_t k = 0; k < R.iter_count; ++k) {
        const uint32_t iteration = R.iter_begin + k * R.iter_stride;
        const uint32_t local_it = local_it0 + k;
        const size_t slot = (size_t)k * R.n_pixels + idx;
        float4 q0, q1, q2, q3;
        bool from_record = true;
        if (HEADS) {
            const float4 h = h_next;
            if (k + 1u < R.iter_count) h_next = ld_stream(R.heads + slot + R.n_pixels);
            if (!LENS && patch && h.w >= 0.0f) {
                // an untraced sample of a pixel with a sky patch: its value is the patch at the sample's jitter (L = 0, beta = 1)
                const float2 j = R.blue_noise[(size_t)k * 65536u + bn_idx];
                const float* s = s_patch + threadIdx.x;
                const f3 v00 = mk3(s[0], s[256], s[512]), v10 = mk3(s[768], s[1024], s[1280]);
                const f3 v01 = mk3(s[1536], s[1792], s[2048]), v11 = mk3(s[2304], s[2560], s[2816]);
                const f3 lo = flerp3(v00, v10, j.x), hi = flerp3(v01, v11, j.x);
                const f3 val = flerp3(lo, hi, j.y);
                q0 = make_float4(val.x, val.y, val.z, 0.0f);
                q1 = make_float4(1.0f, 1.0f, 1.0f, h.w);
                q2 = make_float4(R.cam_origin[0], R.cam_origin[1], R.cam_origin[2], __uint_as_float(0u));      // flags 0: nothing left to add
                q3 = make_float4(h.x, h.y, h.z, 0.0f);
                from_record = false;
            } else if (h.w != -1.0f) {
                // no 64-byte record: a primary ray that started no walk (or a sample that is not rendered)
                const bool rendered = h.w >= 0.0f;
                const float l = rendered ? 0.0f : 1.0f, b = rendered ? 1.0f : 0.0f;
                q0 = make_float4(l, l, l, 0.0f);
                q1 = make_float4(b, b, b, rendered ? h.w : 0.0f);
                if (R.head_org) {
                    const float4 o = ld_stream(R.head_org + slot);
                    q2 = make_float4(o.x, o.y, o.z, __uint_as_float(rendered ? 1u : 0u));
                } else {
                    q2 = make_float4(R.cam_origin[0], R.cam_origin[1], R.cam_origin[2], __uint_as_float(rendered ? 1u : 0u));
                }
                q3 = make_float4(h.x, h.y, h.z, 0.0f);
                from_record = false;
            }
        }
        if (from_record) {
            const float4* rec = reinterpret_cast<const float4*>(R.records + slot);
            q0 = ld_stream(rec); q1 = ld_stream(rec + 1); q2 = ld_stream(rec + 2); q3 = ld_stream(rec + 3);
        }
        f3 value = mk3(q0.x, q0.y, q0.z);
        float tr = q0.w;
        const f3 beta = mk3(q1.x, q1.y, q1.z);
        const float depth = q1.w;
        const f3 env_pos = mk3(q2.x, q2.y, q2.z);
        const uint32_t flags = __float_as_uint(q2.w);
        const f3 dir = mk3(q3.x, q3.y, q3.z);
        if (flags & 1u) {
            if (R.integrator != 0) {
                // vol_integrator :1752: L += beta * sample_atmosphere(ray_pos, ray_dir) -- always the
                // procedural sky, no sky_mult / sky_color, whatever environment_type says
                value += beta * sky.sample(env_pos, dir, sun_dir);
            } else if (R.environment_type == 0) {                                   // :1838-1842
                if (R.has_atmosphere) {
                    // a sample that looks from the camera origin (no sphere bounce) along a direction whose dome cell passed its check: the dome
                    f3 dv;
                    int dcv = -1;                        // the dome that serves this sample's origin: the camera origin's, or the lens variant of its r
                    if (R.sky_dome != nullptr) {
                        if (!LENS) {
                            dcv = (env_pos.x == R.cam_origin[0] && env_pos.y == R.cam_origin[1] && env_pos.z == R.cam_origin[2]) ? 0 : -1;
                        } else {
                            const f3 pe = env_pos - mk3(0.0f, -sky.bottom(), 0.0f);
                            const float re = length_rn(pe);
                            dcv = sky.CamVariant(re, dot(pe, sun_dir) * frcp(re));
                        }
                    }
                    if (dcv >= 0 && dome_lookup(R.sky_dome + (size_t)dcv * ((size_t)SKY_DOME_NU * SKY_DOME_NV), dir, dv))
                        value += dv * beta;
                    else
                        value += sky.sample(env_pos, dir, sun_dir, use_dir_tab) * beta * R.sky_mult * sky_color;
                }
            } else {                                                                 // :1843-1850
                value += env_lookup(R.env_tex, dir) * sky_color * beta * (1.0f / (4.0f * VPT_PI));
            }
        }
        accumulate(value, tr, depth, iteration, local_it, R.rcp_n[k]);
    }
    }
    rm.store(R, idx);
}

// ---- RESOLVED SAMPLES (ResolveParams::lean; TraceParams::resolve) ------------------------------------------------------------------
// sky_fix_kernel: the full sample_atmosphere for the few samples nothing cheaper serves, one thread each at full occupancy (inside the
// per-pixel loop of the tail a wave ran it as soon as ONE lane needed it -- a third of the waves over the volume, for 1-2 % of their samples):
//   (a) queue2: finished paths whose exit direction falls into a flagged dome cell, or whose origin the sphere bounce moved -- their 64-byte
//       record is read, the environment term added as the tail would (:1838-1842), the sample written as head + td;
//   (b) the untraced samples of the pixels without a usable patch (nopatch_list x the launch's iterations): L = 0, beta = 1, from the camera origin.
// LENS (round 5): behind an open lens queue2 also holds the UNTRACED samples raygen could not resolve from a dome (their 64-byte final record: L = 0, beta = 1); there
// are no patches, so no (b)
template <bool LENS>
__global__ __launch_bounds__(256) void sky_fix_kernel(const ResolveParams R, float4* __restrict__ heads) {
    Sky<ResolveParams> sky = {R};
    load_sky_view<LENS>(R, sky);
    const bool use_dir_tab = R.dir_tab != nullptr && R.dir_tab_err[8] != 0u;
    const f3 sky_color = mk3(R.sky_color[0], R.sky_color[1], R.sky_color[2]);
    const f3 sun_dir = mk3(R.sun_dir[0], R.sun_dir[1], R.sun_dir[2]);
    const uint32_t n_a = *R.queue2_count, n_b = LENS ? 0u : *R.nopatch_count * R.iter_count;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n_a + n_b; t += stride) {
        if (t < n_a) {
            const uint32_t slot = R.queue2[t];
            const float4* rec = reinterpret_cast<const float4*>(R.records + slot);
            const float4 q0 = ld_stream(rec), q1 = ld_stream(rec + 1), q2 = ld_stream(rec + 2), q3 = ld_stream(rec + 3);
            f3 value = mk3(q0.x, q0.y, q0.z);
            const f3 beta = mk3(q1.x, q1.y, q1.z);
            value += sky.sample(mk3(q2.x, q2.y, q2.z), mk3(q3.x, q3.y, q3.z), sun_dir, use_dir_tab) * beta * R.sky_mult * sky_color;
            st_stream(heads + slot, make_float4(value.x, value.y, value.z, -1.0f));
            st_stream(R.td + slot, make_float2(q0.w, q1.w));
        } else {
            const uint32_t e = t - n_a;
            const uint32_t p = e / R.iter_count, k = e - p * R.iter_count;
            const size_t slot = (size_t)k * R.n_pixels + R.nopatch_list[p];
            const float4 h = heads[slot];
            if (h.w >= 0.0f) {
                // (as the tail did for such a sample: L = 0, beta = 1, the dome where its cell passed, else in full)
                const f3 dir = mk3(h.x, h.y, h.z);
                f3 value = mk3(0.0f), dv;
                if (dome_lookup(R.sky_dome, dir, dv)) value += dv * mk3(1.0f);
                else value += sky.sample(mk3(R.cam_origin[0], R.cam_origin[1], R.cam_origin[2]), dir, sun_dir, use_dir_tab) * mk3(1.0f) * R.sky_mult * sky_color;
                heads[slot] = make_float4(value.x, value.y, value.z, -1.0f);
                R.td[slot] = make_float2(0.0f, h.w);
            }
        }
    }
}
// tail_stream_kernel: the tail proper once every sample is a head (ResolveParams::lean).  One thread per pixel; the running means are an
// ordered recurrence, so a pixel's iterations stay sequential -- but nothing it reads depends on what it computes: the heads, jitters and
// {alpha, depth} pairs of VPT_TAIL_GROUP iterations are requested together (two memory round trips per group instead of one or two per
// iteration: the old loop's 64 iterations were a chain of ~100 dependent latencies).
#ifndef VPT_TAIL_GROUP
#define VPT_TAIL_GROUP 4                   // x 6 waves per SIMD (8 x 4 waves: 4 % slower; 16: registers; profiles/r04_resolved_samples.txt, r04_four_waves.txt)
#endif
#ifndef VPT_TAIL_STREAM_WAVES_PER_EU
#define VPT_TAIL_STREAM_WAVES_PER_EU 6
#endif
// LENS (round 5): behind an open lens there are no patches -- an untraced sample's head already holds its VALUE (raygen resolved it from the dome of its origin's
// variant): {value, depth}; everything else as behind a closed lens
template <bool LENS>
__global__ __launch_bounds__(256, VPT_TAIL_STREAM_WAVES_PER_EU) void tail_stream_kernel(const ResolveParams R) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= R.n_pixels) return;
    float4 pa = make_float4(0.0f, 0.0f, 0.0f, 0.0f), pb = pa, pc = pa;
    if (!LENS) { const float4* pp = R.sky_patch + 3u * (size_t)idx; pa = pp[0]; pb = pp[1]; pc = pp[2]; }
    const f3 v00 = mk3(pa.x, pa.y, pa.z), v10 = mk3(pa.w, pb.x, pb.y), v01 = mk3(pb.z, pb.w, pc.x), v11 = mk3(pc.y, pc.z, pc.w);
    const bool never = !LENS && R.never_traced != nullptr && R.never_traced[idx] != 0;
    const uint32_t py = idx / R.width, px = idx - py * R.width;
    const float2* bnp = LENS ? nullptr : R.blue_noise + ((py & 255u) * 256u + (px & 255u));
    const uint32_t local_it0 = R.iter_begin / R.iter_stride;
    RunningMeans rm = RunningMeans::load(R, idx);
    for (uint32_t k0 = 0; k0 < R.iter_count; k0 += (uint32_t)VPT_TAIL_GROUP) {
        float4 h[VPT_TAIL_GROUP];
        float2 j[VPT_TAIL_GROUP], td[VPT_TAIL_GROUP];
#pragma unroll
        for (uint32_t u = 0; u < (uint32_t)VPT_TAIL_GROUP; ++u) {
            const uint32_t k = k0 + u;
            const bool live = k < R.iter_count;
            // a pixel raygen emitted nothing for has no heads: every sample is untraced with depth 0 (or not rendered)
            h[u] = (live && !never) ? ld_stream(R.heads + ((size_t)k * R.n_pixels + idx)) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            j[u] = (live && !LENS) ? bnp[(size_t)k * 65536u] : make_float2(0.0f, 0.0f);
            // {alpha, depth} is requested WITH the head, not behind it (round 5): only a resolved sample (head.w == -1) uses it, but waiting for the head to know that made
            // every group two dependent round trips (8 more bytes read per live untraced sample; whatever they hold is ignored).  Tail -3 %: profiles/r05_compact_rays.txt
            td[u] = (live && !never) ? ld_stream(R.td + ((size_t)k * R.n_pixels + idx)) : make_float2(0.0f, 0.0f);
        }
#pragma unroll
        for (uint32_t u = 0; u < (uint32_t)VPT_TAIL_GROUP; ++u) {
            const uint32_t k = k0 + u;
            if (k < R.iter_count) {
                const uint32_t iteration = R.iter_begin + k * R.iter_stride;
                f3 value;
                float tr = 0.0f, depth = 0.0f;
                if (never) {
                    value = (iteration < R.max_interactions && R.render) ? flerp3(flerp3(v00, v10, j[u].x), flerp3(v01, v11, j[u].x), j[u].y) : mk3(1.0f);
                } else if (h[u].w >= 0.0f) {
                    if (LENS) value = mk3(h[u].x, h[u].y, h[u].z);                                  // untraced, resolved by raygen from its origin's dome
                    else value = flerp3(flerp3(v00, v10, j[u].x), flerp3(v01, v11, j[u].x), j[u].y);      // untraced: the patch at the sample's jitter
                    depth = h[u].w;
                } else if (h[u].w == -1.0f) {
                    value = mk3(h[u].x, h[u].y, h[u].z);                                               // resolved by the tracer or by sky_fix_kernel
                    tr = td[u].x;
                    depth = td[u].y;
                } else {
                    value = mk3(1.0f);                                                                 // not rendered: WHITE (:2248, :2254)
                }
                rm.add(R, value, tr, depth, iteration, local_it0 + k, R.rcp_n[k]);
            }
        }
    }
    rm.store(R, idx);
}
hipError_t launch_sky_fix(const ResolveParams& R, hipStream_t stream) {
    if (R.lens_radius != 0.0f) hipLaunchKernelGGL(sky_fix_kernel<true>, dim3(512), dim3(256), 0, stream, R, const_cast<float4*>(R.heads));
    else hipLaunchKernelGGL(sky_fix_kernel<false>, dim3(512), dim3(256), 0, stream, R, const_cast<float4*>(R.heads));
    return hipGetLastError();
}
hipError_t launch_tail_stream(const ResolveParams& R, hipStream_t stream) {
    if (R.lens_radius != 0.0f) hipLaunchKernelGGL(tail_stream_kernel<true>, dim3((R.n_pixels + 255u) / 256u), dim3(256), 0, stream, R);
    else hipLaunchKernelGGL(tail_stream_kernel<false>, dim3((R.n_pixels + 255u) / 256u), dim3(256), 0, stream, R);
    return hipGetLastError();
}

// display / raw images of an accumulation buffer that was changed outside the render (vpt_allreduce_accum: the batch
// tonemapped this rank's running mean, the job's mean arrives afterwards); raw.w (alpha of the last iteration) is kept
__global__ void display_kernel(const float* __restrict__ accum, unsigned int* __restrict__ display, float4* __restrict__ raw, uint32_t n, float exposure_scale) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    unsigned int packed;
    const f3 val = tonemap(mk3(accum[3 * idx], accum[3 * idx + 1], accum[3 * idx + 2]), exposure_scale, packed);
    if (display) display[idx] = packed;
    if (raw) { raw[idx].x = val.x; raw[idx].y = val.y; raw[idx].z = val.z; }
}
hipError_t launch_display(const float* accum, unsigned int* display, float* raw, uint32_t n, float exposure_scale, hipStream_t stream) {
    hipLaunchKernelGGL(display_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, accum, display, reinterpret_cast<float4*>(raw), n, exposure_scale);
    return hipGetLastError();
}

// view point of the per-frame tables: (r, mu_s) of the camera origin with this file's own arithmetic, and per variant of r (one
// per binary32 step within k of it) the ground table's log-distance map -- thread v = variant
__global__ void sky_view_kernel(const ResolveParams R, SkyView* out, int k) {
    typedef Sky<ResolveParams> S;
    const int v = (int)threadIdx.x;
    const S sky = {R};
    const f3 p = mk3(R.cam_tab_pos[0], R.cam_tab_pos[1] + sky.bottom(), R.cam_tab_pos[2]);       // view point - earth centre
    const float r0 = length_rn(p);                       // (correctly rounded, like every r the ground geometry of vpt_sky.h forms)
    if (v == 0) {
        out->r = r0;
        out->mu_s = dot(p, mk3(R.sun_dir[0], R.sun_dir[1], R.sun_dir[2])) * frcp(r0);
        out->k = k;
        out->pad_ = 0;
    }
    if (v > 2 * k) return;
    const float r = __uint_as_float(__float_as_uint(r0) + (uint32_t)(v - k));
    const float b = sky.bottom();
    const float d_min = r - b, d_max = fsqrt(fmax_((r - b) * (r + b), 0.0f));
    // the ground table covers rays at least ~2 degrees below the horizon: d <= min(33 (r - bottom), 0.35 horizon distance); the
    // grazing rest takes the full path (vpt_sky.h, GroundFromTable).  No table from outside the atmosphere or on the ground.
    float4 t = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (r > b && r <= sky.top() && d_max > d_min) {
        const float inv_range = frcp(__builtin_amdgcn_logf(fdiv(d_max, d_min)));
        const float x_use = fmin_(__builtin_amdgcn_logf(fdiv(fmin_(33.0f * d_min, 0.35f * d_max), d_min)) * inv_range, 1.0f);
        t = make_float4(frcp(d_min), inv_range, x_use, x_use > 0.05f ? 1.0f : 0.0f);
    }
    out->tab[v] = t;
}
// camera-point scattering table (vpt_sky.h): per variant 8 nu slices x 128 mu rows, one thread per entry
__global__ void sky_cam_table_kernel(const ResolveParams R, const SkyView* view, float4* out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t variant = t >> 10, e = t & 1023u;
    if (variant > 2u * (uint32_t)view->k) return;
    const float r = __uint_as_float(__float_as_uint(view->r) + variant - (uint32_t)view->k);
    f3 sc, mie;
    sky_cam_table_entry(R, r, view->mu_s, e >> 7, e & 127u, sc, mie);
    out[2u * t] = make_float4(sc.x, sc.y, sc.z, 0.0f);
    out[2u * t + 1u] = make_float4(mie.x, mie.y, mie.z, 0.0f);
}
hipError_t launch_sky_cam_table(const ResolveParams& R, SkyView* view, float4* out, int k, hipStream_t stream) {
    hipLaunchKernelGGL(sky_view_kernel, dim3(1), dim3(64), 0, stream, R, view, k);
    hipLaunchKernelGGL(sky_cam_table_kernel, dim3(4 * (2 * k + 1)), dim3(256), 0, stream, R, view, out);
    return hipGetLastError();
}

// view-point ground table (vpt_sky.h, GroundNode): one thread per node and variant ...
__global__ void sky_dir_table_kernel(const ResolveParams R, const SkyView* view, float4* out) {
    typedef Sky<ResolveParams> S;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t nodes = (uint32_t)(S::DT_NX * S::DT_NN);
    const uint32_t variant = t / nodes, e = t % nodes;
    if (variant > 2u * (uint32_t)view->k || view->tab[variant].w == 0.0f) return;
    const S sky = {R};
    const float r = __uint_as_float(__float_as_uint(view->r) + variant - (uint32_t)view->k);
    const uint32_t ix = e / (uint32_t)S::DT_NN, in = e % (uint32_t)S::DT_NN;
    f3 A, B;
    float scale;
    sky.GroundNode(r, view->mu_s, (float)ix * (1.0f / (float)(S::DT_NX - 1)), -1.0f + (float)in * (2.0f / (float)(S::DT_NN - 1)), A, B, scale);
    out[2u * t] = make_float4(A.x, A.y, A.z, 0.0f);
    out[2u * t + 1u] = make_float4(B.x, B.y, B.z, 0.0f);
}
// ... and one per cell of the used part: the full evaluation at the cell centre against the bilinear interpolant, for the cells a
// view ray can reach (|nu - mu mu_s| <= sin(theta) sin(theta_s), widened by two cells); err = max relative deviation over all variants
__global__ void sky_dir_table_check_kernel(const ResolveParams R, const SkyView* view, const float4* tab, unsigned long long* err) {
    typedef Sky<ResolveParams> S;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t cells = (uint32_t)((S::DT_NX - 1) * (S::DT_NN - 1));
    const uint32_t variant = t / cells, c = t % cells;
    if (variant > 2u * (uint32_t)view->k || view->tab[variant].w == 0.0f) return;
    const S sky = {R};
    const float r = __uint_as_float(__float_as_uint(view->r) + variant - (uint32_t)view->k);
    const float mu_s = view->mu_s;
    const uint32_t ix = c / (uint32_t)(S::DT_NN - 1), in = c % (uint32_t)(S::DT_NN - 1);
    const float x = ((float)ix + 0.5f) * (1.0f / (float)(S::DT_NX - 1)), nu = -1.0f + ((float)in + 0.5f) * (2.0f / (float)(S::DT_NN - 1));
    if ((float)ix * (1.0f / (float)(S::DT_NX - 1)) > view->tab[variant].z) return;           // beyond the part that is used
    const float b = sky.bottom();
    const float h2 = (r - b) * (r + b), d_min = r - b, d = d_min * __builtin_amdgcn_exp2f(x * __builtin_amdgcn_logf(fdiv(fsqrt(fmax_(h2, 0.0f)), d_min)));
    const float mu = clampf(fdiv(-(h2 + d * d), 2.0f * r * d), -1.0f, 1.0f);
    const float band = fsqrt(fmax_((1.0f - mu * mu) * (1.0f - mu_s * mu_s), 0.0f)) + 2.0f * (2.0f / (float)(S::DT_NN - 1));
    if (fabsf(nu - mu * mu_s) > band) return;
    f3 A, B, Ai, Bi;
    float scale;
    sky.GroundNode(r, mu_s, x, nu, A, B, scale);
    S::DirTabLerp(tab + variant * (2u * S::DT_NX * S::DT_NN), (ix * (uint32_t)S::DT_NN + in) * 2u, 0.5f, 0.5f, Ai, Bi);
    const float ph = S::MiePhase(sky.f(AF_MIE_G), nu);
    const f3 Re = fscale_add3(B, ph, A), Ri = fscale_add3(Bi, ph, Ai);
    const float dev = fmax_(fmax_(fabsf(Ri.x - Re.x), fabsf(Ri.y - Re.y)), fabsf(Ri.z - Re.z)) / fmax_(scale, 1e-30f);
    // NaN compares false everywhere: let it through as a huge error
    const uint32_t bits = dev == dev ? __float_as_uint(dev) : 0x7fc00000u;
    if (bits != 0u) atomicMax(err, ((unsigned long long)bits << 32) | c);             // high word: the error; low word: where
}
// ... and against the path it replaces: for every reachable cell centre of every variant a real view ray with that distance to the
// ground and that nu -- from the camera origin, displaced along the vertical by the variant's binary32 steps of r (and, behind an
// open lens, horizontally within the lens radius) -- is evaluated
// twice through sample_atmosphere: in full (the reference's arithmetic: binary32 ground point, its radius as the reference finds
// it) and through the table; the two tone-curved radiances the tail would add to L are compared.  Per variant v, at err[8 + 4 v]:
// largest relative difference (high word) and its cell, rays compared, UNFLIPPED rays off by more than 1e-3, FLIPPED rays.  A ray is "flipped" when
// the full path finds its binary32 ground point one step (0.5 m) above the ground (vpt_sky.h: SkyRadianceToPoint) -- the reference's own ray-to-ray
// noise, 2.5 km of the tables' rho, ~5e-3 of the radiance: a table is a smooth function and returns such a ray's unflipped value by construction, so the
// two populations are gated apart (round 5: with the ground geometry formed as the strict side forms it, ~2 % of the rays flip, where the approximate
// roots of rounds 1-4 flipped 0.5 % of them on one side only and the single "share above 1e-3" gate sat at its limit).
template <bool LENS>
__global__ void sky_dir_table_rays_kernel(const ResolveParams R, const SkyView* view, unsigned long long* err) {
    typedef Sky<ResolveParams> S;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t cells = (uint32_t)((S::DT_NX - 1) * (S::DT_NN - 1));
    const uint32_t variant = t / cells, c = t % cells;
    const int k = view->k;
    bool valid = variant <= 2u * (uint32_t)k;              // (no early return: the wave-level sums below need every lane)
    S sky = {R};
    load_sky_view<LENS>(R, sky);
    const uint32_t ix = c / (uint32_t)(S::DT_NN - 1), in = c % (uint32_t)(S::DT_NN - 1);
    const float x = ((float)ix + 0.5f) * (1.0f / (float)(S::DT_NX - 1)), nu = -1.0f + ((float)in + 0.5f) * (2.0f / (float)(S::DT_NN - 1));
    const f3 cam = mk3(R.cam_tab_pos[0], R.cam_tab_pos[1], R.cam_tab_pos[2]);
    const f3 sun = mk3(R.sun_dir[0], R.sun_dir[1], R.sun_dir[2]);
    const f3 ec = mk3(0.0f, -sky.bottom(), 0.0f);
    const f3 up0 = normalize(cam - ec);
    // the origin: `variant - k` binary32 steps of r above / below the camera origin (which variant that lands on is the tail's own
    // decision, below)
    const float dr = valid ? __uint_as_float(__float_as_uint(view->r) + variant - (uint32_t)k) - view->r : 0.0f;
    f3 pos = cam + up0 * dr;
    if (LENS) {
        // ... and, like the samples behind an open lens, somewhere within a lens radius of it horizontally (a fixed pseudo-random
        // point per cell): the binary32 rounding of the ground point depends on where exactly the origin sits
        const uint32_t h = (t * 2654435761u) ^ (t >> 13) * 40503u;
        const float th = (float)(h & 0xffffu) * (6.2831853f / 65536.0f), rho = fsqrt((float)(h >> 16) * (1.0f / 65536.0f)) * fabsf(R.lens_radius);
        const f3 t1 = normalize(cross(up0, fabsf(up0.x) < 0.9f ? mk3(1.0f, 0.0f, 0.0f) : mk3(0.0f, 0.0f, 1.0f))), t2 = cross(up0, t1);
        pos = pos + (t1 * __builtin_cosf(th) + t2 * __builtin_sinf(th)) * rho;
    }
    const f3 p = pos - ec;
    const float r = length_rn(p);
    int cv = 0;
    if (LENS) {
        cv = sky.CamVariant(r, dot(p, sun) * frcp(r));
        valid = valid && cv >= 0;
        cv = max(cv, 0);
    } else {
        valid = valid && variant == 0u;
    }
    valid = valid && view->tab[cv].w != 0.0f && x <= view->tab[cv].z;                          // a table, and within the part that is used
    const f3 up = p * frcp(r);
    const float mu_s = dot(up, sun);
    const float b = sky.bottom();
    const float h2 = (r - b) * (r + b), d_min = r - b, d = d_min * __builtin_amdgcn_exp2f(x * __builtin_amdgcn_logf(fdiv(fsqrt(fmax_(h2, 0.0f)), d_min)));
    const float mu = clampf(fdiv(-(h2 + d * d), 2.0f * r * d), -1.0f, 1.0f);
    // a unit vector with view . up = mu and view . sun = nu (two solutions mirrored in the sun's vertical plane: take one)
    const float sv = fsqrt(fmax_(1.0f - mu * mu, 0.0f)), ss = fsqrt(fmax_(1.0f - mu_s * mu_s, 0.0f));
    valid = valid && sv * ss > 1e-6f;
    const float cphi = fdiv(nu - mu * mu_s, sv * ss);
    valid = valid && fabsf(cphi) <= 1.0f;                                                      // else: no view ray has this (mu, nu)
    float dev = 0.0f;
    bool flip = false;
    if (valid) {
        const f3 e1 = (sun - up * mu_s) * frcp(ss), e2 = cross(up, e1);
        const f3 dir = normalize(up * mu + (e1 * cphi + e2 * fsqrt(fmax_(1.0f - cphi * cphi, 0.0f))) * sv);
        const f3 full = sky.sample(pos, dir, sun, false, nullptr, &flip), tab = sky.sample(pos, dir, sun, true);
        dev = fmax_(fmax_(fabsf(tab.x - full.x), fabsf(tab.y - full.y)), fabsf(tab.z - full.z)) / fmax_(fmax_(fmax_(full.x, full.y), full.z), 1e-30f);
    }
    const uint32_t bits = dev == dev ? __float_as_uint(dev) : 0x7fc00000u;                      // NaN compares false everywhere: let it through as a huge error
    // per variant, one set of atomics per WAVE (a quarter of a million threads on three words otherwise)
    const int lane = __lane_id();
    for (int v = 0; v <= 2 * k; ++v) {
        const unsigned long long m = __ballot(valid && cv == v);
        if (m == 0ull) continue;
        const unsigned long long above = __ballot(valid && cv == v && !flip && !(dev <= 1e-3f));
        const unsigned long long above2 = __ballot(valid && cv == v && flip);
        unsigned long long key = (valid && cv == v) ? (((unsigned long long)bits << 32) | c) : 0ull;
        for (int s = 32; s >= 1; s >>= 1) {
            const unsigned long long o = ((unsigned long long)(uint32_t)__shfl_xor((int)(key >> 32), s) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)key, s);
            key = o > key ? o : key;
        }
        // the flipped rays' summed deviation, in units of 2^-24 (what they cost a smooth cache on average: share x amplitude)
        unsigned long long fsum = (valid && cv == v && flip && dev == dev) ? (unsigned long long)(fmin_(dev, 1.0f) * 16777216.0f) : 0ull;
        for (int s = 32; s >= 1; s >>= 1)
            fsum += ((unsigned long long)(uint32_t)__shfl_xor((int)(fsum >> 32), s) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)fsum, s);
        if (lane == 0) {
            unsigned long long* e = err + 8 + SKY_DIR_ERR_STRIDE * v;
            if ((key >> 32) != 0ull) atomicMax(e, key);
            atomicAdd(e + 1, (unsigned long long)__popcll(m));
            if (above != 0ull) atomicAdd(e + 2, (unsigned long long)__popcll(above));
            if (above2 != 0ull) atomicAdd(e + 3, (unsigned long long)__popcll(above2));
            if (fsum != 0ull) atomicAdd(e + 4, fsum);
        }
    }
}
// The verdicts the tail reads.  Per variant: against real rays through the full path no ray is off by more than 2 %, at most 0.5 % of them are
// UNFLIPPED rays off by more than 1e-3 (the image tolerance is 1e-3 rel. L2) and the flipped ones cost at most 3e-4 on average (their summed deviation
// over ALL rays: a smooth cache returns the mean of that noise; measured 1e-4 from a camera near the ground) -- a variant that fails loses its table (SkyView::tab[v].w = 0: its ground hits are evaluated in full).  err[4] = 1 when the interpolant follows its nodes (err[0] <= tol) and a variant is left;
// err[1..3], err[6] = the centre variant's figures (vpt_test_get_dir_table_check / _flips), err[5] = variants in use.
__global__ void sky_dir_table_verdict_kernel(unsigned long long* err, SkyView* view, float tol) {
    const int k = view->k;
    unsigned long long in_use = 0;
    for (int v = 0; v <= 2 * k; ++v) {
        const unsigned long long* e = err + 8 + SKY_DIR_ERR_STRIDE * v;
        const float worst = __uint_as_float((uint32_t)(e[0] >> 32));
        // flipped rays: what they cost on average, (sum of their deviations) / (all rays) <= 3e-4 -- e[4] is in units of 2^-24: 3e-4 x 2^24 = 5033
        const bool ok = view->tab[v].w != 0.0f && e[1] > 0ull && worst <= 2e-2f && e[2] * 200ull <= e[1] && e[4] <= e[1] * 5033ull;
        if (!ok) view->tab[v].w = 0.0f;
        in_use += ok ? 1ull : 0ull;
    }
    { const unsigned long long* c = err + 8 + SKY_DIR_ERR_STRIDE * k; err[1] = c[0]; err[2] = c[1]; err[3] = c[2]; err[6] = c[3]; err[7] = c[4]; }
    err[5] = in_use;
    err[4] = (__uint_as_float((uint32_t)(err[0] >> 32)) <= tol && in_use != 0ull) ? 1ull : 0ull;
}
size_t sky_cam_table_bytes() { return sizeof(float4) * 2u * 8u * 128u * (2u * SKY_VIEW_MAX_K + 1u); }
size_t sky_dir_table_bytes() { return sizeof(float4) * 2u * Sky<ResolveParams>::DT_NX * Sky<ResolveParams>::DT_NN * (2u * SKY_VIEW_MAX_K + 1u); }
hipError_t launch_sky_dir_table(const ResolveParams& R, SkyView* view, float4* tab, unsigned long long* err, int k, hipStream_t stream) {
    typedef Sky<ResolveParams> S;
    hipError_t e = hipMemsetAsync(err, 0, SKY_DIR_ERR_WORDS * sizeof(unsigned long long), stream);
    if (e != hipSuccess) return e;
    const int variants = 2 * k + 1;
    hipLaunchKernelGGL(sky_dir_table_kernel, dim3((variants * S::DT_NX * S::DT_NN + 255) / 256), dim3(256), 0, stream, R, view, tab);
    hipLaunchKernelGGL(sky_dir_table_check_kernel, dim3((variants * (S::DT_NX - 1) * (S::DT_NN - 1) + 255) / 256), dim3(256), 0, stream, R, view, tab, err);
    // real rays through both paths (R.dir_tab / R.sky_view / R.cam_tab must already point at the tables: the host sets them first)
    const dim3 rg((variants * (S::DT_NX - 1) * (S::DT_NN - 1) + 255) / 256);
    if (k > 0) hipLaunchKernelGGL(sky_dir_table_rays_kernel<true>, rg, dim3(256), 0, stream, R, view, err);
    else hipLaunchKernelGGL(sky_dir_table_rays_kernel<false>, rg, dim3(256), 0, stream, R, view, err);
    hipLaunchKernelGGL(sky_dir_table_verdict_kernel, dim3(1), dim3(1), 0, stream, err, view, R.dir_tab_tol);
    return hipGetLastError();
}

// test hook (vpt_test_sky_samples): sample_atmosphere from the table's view point along n given directions
__global__ void sky_samples_kernel(const ResolveParams R, const float* __restrict__ origins, const float* __restrict__ dirs, float* __restrict__ out, uint32_t n, int use_table) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Sky<ResolveParams> sky = {R};
    load_sky_view<true>(R, sky);
    const bool use_dir_tab = use_table && R.dir_tab != nullptr && R.dir_tab_err[8] != 0u;
    const f3 from = origins ? mk3(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2]) : mk3(R.cam_tab_pos[0], R.cam_tab_pos[1], R.cam_tab_pos[2]);
    const f3 v = sky.sample(from, mk3(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]),
                            mk3(R.sun_dir[0], R.sun_dir[1], R.sun_dir[2]), use_dir_tab);
    out[3 * i] = v.x; out[3 * i + 1] = v.y; out[3 * i + 2] = v.z;
}
hipError_t launch_sky_samples(const ResolveParams& R, const float* origins, const float* dirs, float* out, uint32_t n, int use_table, hipStream_t stream) {
    hipLaunchKernelGGL(sky_samples_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, R, origins, dirs, out, n, use_table);
    return hipGetLastError();
}

hipError_t launch_tail_resolve(const ResolveParams& R, hipStream_t stream) {
    const dim3 grid((R.n_pixels + 255u) / 256u), block(256);
    const bool lens = R.lens_radius != 0.0f;             // the host builds k > 0 table variants exactly then
    if (R.heads && lens) hipLaunchKernelGGL((tail_resolve_kernel<true, true>), grid, block, 0, stream, R);
    else if (R.heads) hipLaunchKernelGGL((tail_resolve_kernel<true, false>), grid, block, 0, stream, R);
    else if (lens) hipLaunchKernelGGL((tail_resolve_kernel<false, true>), grid, block, 0, stream, R);
    else hipLaunchKernelGGL((tail_resolve_kernel<false, false>), grid, block, 0, stream, R);
    return hipGetLastError();
}

}  // namespace vpt
