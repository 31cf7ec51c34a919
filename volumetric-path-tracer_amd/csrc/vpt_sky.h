// vpt_sky.h -- Bruneton precomputed atmospheric scattering, look-up side: `sample_atmosphere`
// (render_kernel.cu:839-895) and the functions under it (:369-835; published algorithm:
// E. Bruneton, "Precomputed Atmospheric Scattering", EGSR 2008 + the 2017 reference implementation
// `functions.glsl`).  Used by the environment tail (vpt_tail.hip) and by estimate_sky in the
// vol_integrator tracer (vpt_trace_vol.hip).  RP is any parameter block with atm_f[] and the four
// look-up textures (ResolveParams, TraceParams).
//
// VALUE-ONLY arithmetic: the result is added to L once and feeds no random walk.  The
// reference evaluates it under --use_fast_math; here, independent of the including translation
// unit's flags, it is written with
//   * v_rcp_f32 / v_sqrt_f32 based divide and sqrt (1 ulp each, like __fdividef / __fsqrt_rn-less
//     fast math) and hardware exp2 / log2 for the tone curve,
//   * FMA-contracted interpolation (a + t (b - a) as one subtract + one fma),
//   * float instead of double for the texture-coordinate maps (:419; the doubles come from
//     unsuffixed literals), but DOUBLE kept where the reference's doubles matter: the
//     r^2 (mu^2 - 1) + R^2 discriminants (:389, :401) cancel catastrophically in fp32,
//   * launch-uniform sub-expressions (H, 1/H, A of :553, ...) evaluated once on the host
//     (pack_atmosphere, vpt_host.hip) -- gfx950 has no scalar float ALU, so a uniform sqrt would
//     otherwise be repeated by every wave,
//   * compile-time table extents (constants.h:50-62), 32-bit unsigned texel offsets (SGPR base +
//     VGPR offset addressing), and the y/z taps shared between the two nu slices and the two 3-D
//     tables of GetCombinedScattering.
// Tolerance against the oracle (which restates the reference literally): tests/test_gpu_atmosphere.py.
#pragma once

#include "vpt_dome.h"
#include "vpt_tex.h"

namespace vpt {

// atm_f[] layout; written by pack_atmosphere (vpt_host.hip)
enum {
    AF_BOTTOM = 0, AF_TOP = 1, AF_USE_LUM = 2, AF_MIE_G = 3, AF_SUN_ANG = 4, AF_MU_S_MIN = 5, AF_EXPOSURE = 6,
    AF_H = 7,                 // sqrt(top^2 - bottom^2)
    AF_SKY_K = 8, AF_SUN_K = 11, AF_SOLAR = 14, AF_GROUND = 17, AF_WHITE = 20,
    AF_INV_H = 23,            // 1 / H
    AF_INV_DMUS = 24,         // 1 / (H - (top - bottom)): the mu_s map of :548-552
    AF_INV_A = 25,            // 1 / A, A = -2 mu_s_min bottom / (H - (top - bottom))     (:553)
    AF_INV_TB = 26,           // 1 / (top - bottom)
    AF_COS_SUN = 27,          // cos(sun_angular_radius)
    AF_SOLAR_RAD = 28,        // solar_irradiance / (pi sun_angular_radius^2) [* sun_k]  (GetSolarRadiance :835)
    AF_EXPO_W = 31,           // exposure [* 1e-5] / white_point                          (:883-885)
    AF_COUNT = 34,
};

// texel fetch through a 32-bit BYTE offset (tables are <= 16 MiB): SGPR base + VGPR offset addressing
VPT_D f3 ld_f3(const float4* __restrict__ p, uint32_t i) {
    const uint32_t off = i << 4;
    const float4 v = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(p) + off);
    return mk3(v.x, v.y, v.z);
}

struct Tap { uint32_t i0, i1; float a; };
template <int N, bool WRAP>
VPT_D Tap lut_tap(float u) {
    Tap t;
    const float xb = ffma(u, (float)N, -0.5f);
    const float fl = floorf(xb);
    t.a = xb - fl;
    const int i = (int)fl;
    if (WRAP) {
        t.i0 = (uint32_t)i & (uint32_t)(N - 1);
        t.i1 = (uint32_t)(i + 1) & (uint32_t)(N - 1);
    } else {
        t.i0 = (uint32_t)min(max(i, 0), N - 1);
        t.i1 = (uint32_t)min(max(i + 1, 0), N - 1);
    }
    return t;
}
// 2-D 256x64 float4 table, x wraps, y clamps (atmosphere.cpp:503-573)
VPT_D f3 lut2d(const float* data, float u, float v) {
    const float4* p = reinterpret_cast<const float4*>(data);
    const Tap tx = lut_tap<256, true>(u), ty = lut_tap<64, false>(v);
    f3 c = flerp3(ld_f3(p, ty.i0 * 256u + tx.i0), ld_f3(p, ty.i0 * 256u + tx.i1), tx.a);
    if (ty.a != 0.0f)          // a zero weight is exact: c0 + 0 (c1 - c0) == c0 (points ON the ground: rho = 0)
        c = flerp3(c, flerp3(ld_f3(p, ty.i1 * 256u + tx.i0), ld_f3(p, ty.i1 * 256u + tx.i1), tx.a), ty.a);
    return c;
}

template <class RP>
struct Sky {
    const RP& R;
    // view point of the per-frame tables (SkyView, loaded once per thread by the tail); view_k < 0: no tables (the tracer)
    float view_r = 0.0f, view_mu_s = 0.0f;
    int view_k = -1;
    bool view_multi = false;                        // view_k > 0 (open lens); a compile-time constant of the tail's instantiations
    float4 view_vt0 = {0.0f, 0.0f, 0.0f, 0.0f};     // SkyView::tab[0] when !view_multi (closed lens: one variant, kept in registers)
    VPT_D float f(int i) const { return R.atm_f[i]; }
    VPT_D f3 v(int i) const { return mk3(R.atm_f[i], R.atm_f[i + 1], R.atm_f[i + 2]); }
    VPT_D float bottom() const { return f(AF_BOTTOM); }
    VPT_D float top() const { return f(AF_TOP); }
    VPT_D bool lum() const { return f(AF_USE_LUM) != 0.0f; }

    VPT_D static float ClampCosine(float mu) { return clampf(mu, -1.0f, 1.0f); }
    VPT_D float ClampRadius(float r) const { return clampf(r, bottom(), top()); }
    VPT_D static float SafeSqrt(float a) { return fsqrt(fmax_(a, 0.0f)); }
    // r^2 (mu^2 - 1) + R^2 in double (:389, :401)
    VPT_D static double Disc(float r, float mu, float R_) { return (double)(r * r) * ((double)(mu * mu) - 1.0) + (double)(R_ * R_); }
    VPT_D float DistanceToTop(float r, float mu) const {                              // :389
        return fmax_(-r * mu + SafeSqrt((float)Disc(r, mu, top())), 0.0f);
    }
    // sqrt(d^2 + 2 r mu d + r^2) in DOUBLE, rounded once (:475, :781): at a ground point the result
    // decides between r = bottom and bottom + 1 ulp (0.5 m), i.e. rho = 0 or 2.5 km in the table maps
    VPT_D static float RadiusAt(float r, float mu, float d) {
        return (float)sqrt((double)(d * d) + 2.0 * (double)r * (double)mu * (double)d + (double)(r * r));
    }
    VPT_D bool HitsGround(float r, float mu) const { return mu < 0.0f && Disc(r, mu, bottom()) >= 0.0; }   // :401
    template <int N>
    VPT_D static float UnitToTex(float x) { return ffma(x, (float)(1.0 - 1.0 / (double)N), (float)(0.5 / (double)N)); }   // :419
    VPT_D f3 TransmittanceToTop(float r, float mu) const {                            // :429-470
        const float rho = SafeSqrt(r * r - bottom() * bottom());
        const float d = DistanceToTop(r, mu);
        const float d_min = top() - r;
        const float d_max = rho + f(AF_H);
        const float x_mu = fdiv(d - d_min, d_max - d_min);
        const float x_r = rho * f(AF_INV_H);
        return lut2d(R.transmittance_tex.data, UnitToTex<256>(x_mu), UnitToTex<64>(x_r));
    }
    // r_d = ClampRadius(RadiusAt(r, mu, d)) is passed in: GetSkyRadianceToPoint needs the very same
    // value again as r_p (:781; d >= 0 there, so its max(d, 0) changes nothing) and the double-precision
    // square root is the most expensive single operation of the ground path
    VPT_D f3 Transmittance(float r, float mu, float d, float r_d, bool ground) const {           // :472
        const float mu_d = ClampCosine(fdiv(r * mu + d, r_d));
        f3 num, den;
        if (ground) { num = TransmittanceToTop(r_d, -mu_d); den = TransmittanceToTop(r, -mu); }
        else { num = TransmittanceToTop(r, mu); den = TransmittanceToTop(r_d, mu_d); }
        return fmin3(mk3(fdiv(num.x, den.x), fdiv(num.y, den.y), fdiv(num.z, den.z)), mk3(1.0f));
    }
    VPT_D f3 TransmittanceToSun(float r, float mu_s) const {                          // :486
        const float sin_theta_h = fdiv(bottom(), r);
        const float cos_theta_h = -fsqrt(fmax_(1.0f - sin_theta_h * sin_theta_h, 0.0f));
        const float sa = f(AF_SUN_ANG);
        const float a = -sin_theta_h * sa, b = sin_theta_h * sa;
        const float y = clampf(fdiv(mu_s - cos_theta_h - a, b - a), 0.0f, 1.0f);     // smoothstep
        return TransmittanceToTop(r, mu_s) * (y * y * (3.0f - (2.0f * y)));
    }
    VPT_D static float RayleighPhase(float nu) {                                      // :508
        const float k = 3.0f / (16.0f * VPT_PI);
        return k * (1.0f + nu * nu);
    }
    VPT_D static float MiePhase(float g, float nu) {                                  // :514
        const float k = fdiv(3.0f / (8.0f * VPT_PI) * (1.0f - g * g), 2.0f + g * g);
        const float b = 1.0f + g * g - 2.0f * g * nu;
        return fdiv(k * (1.0f + nu * nu), b * fsqrt(b));                              // pow(b, 1.5)
    }
    VPT_D f4 ScatteringUvwz(float r, float mu, float mu_s, float nu, bool ground) const {   // :520-569
        const float H = f(AF_H);
        const float rho = SafeSqrt(r * r - bottom() * bottom());
        const float u_r = UnitToTex<32>(rho * f(AF_INV_H));
        const float r_mu = r * mu;
        const float disc = r_mu * r_mu - r * r + bottom() * bottom();
        float u_mu;
        if (ground) {
            const float d = -r_mu - SafeSqrt(disc);
            const float d_min = r - bottom();
            const float d_max = rho;
            u_mu = 0.5f - 0.5f * UnitToTex<64>(d_max == d_min ? 0.0f : fdiv(d - d_min, d_max - d_min));
        } else {
            const float d = -r_mu + SafeSqrt(disc + H * H);
            const float d_min = top() - r;
            const float d_max = rho + H;
            u_mu = 0.5f + 0.5f * UnitToTex<64>(fdiv(d - d_min, d_max - d_min));
        }
        const float d = DistanceToTop(bottom(), mu_s);
        const float a = (d - (top() - bottom())) * f(AF_INV_DMUS);
        const float u_mu_s = UnitToTex<32>(fdiv(fmax_(1.0f - a * f(AF_INV_A), 0.0f), 1.0f + a));
        const float u_nu = (nu + 1.0f) * 0.5f;
        return mk4(u_nu, u_mu_s, u_mu, u_r);
    }
    // bilinear (y, z) x linear (x) fetch of BOTH 3-D tables at one u, sharing the row offsets.
    // A zero interpolation weight is exact -- c0 + 0 (c1 - c0) == c0 -- so rows / slices with weight 0
    // are not fetched: a point ON the ground (r = bottom: rho = 0 and d_min = d_max) sits exactly on
    // the first r slice and on a mu row, and needs 2 of the 8 texels per table.
    VPT_D void fetch_pair(const Tap& tx, const uint32_t r00, const uint32_t r10, const uint32_t r01, const uint32_t r11, float ay, float az,
                          f3& sc, f3& mie) const {
        const float4* __restrict__ ps = reinterpret_cast<const float4*>(R.scattering_tex.data);
        const float4* __restrict__ pm = reinterpret_cast<const float4*>(R.single_mie_tex.data);
        sc = flerp3(ld_f3(ps, r00 + tx.i0), ld_f3(ps, r00 + tx.i1), tx.a);
        mie = flerp3(ld_f3(pm, r00 + tx.i0), ld_f3(pm, r00 + tx.i1), tx.a);
        if (ay != 0.0f) {
            sc = flerp3(sc, flerp3(ld_f3(ps, r10 + tx.i0), ld_f3(ps, r10 + tx.i1), tx.a), ay);
            mie = flerp3(mie, flerp3(ld_f3(pm, r10 + tx.i0), ld_f3(pm, r10 + tx.i1), tx.a), ay);
        }
        if (az != 0.0f) {
            f3 s1 = flerp3(ld_f3(ps, r01 + tx.i0), ld_f3(ps, r01 + tx.i1), tx.a);
            f3 m1 = flerp3(ld_f3(pm, r01 + tx.i0), ld_f3(pm, r01 + tx.i1), tx.a);
            if (ay != 0.0f) {
                s1 = flerp3(s1, flerp3(ld_f3(ps, r11 + tx.i0), ld_f3(ps, r11 + tx.i1), tx.a), ay);
                m1 = flerp3(m1, flerp3(ld_f3(pm, r11 + tx.i0), ld_f3(pm, r11 + tx.i1), tx.a), ay);
            }
            sc = flerp3(sc, s1, az);
            mie = flerp3(mie, m1, az);
        }
    }
    VPT_D f3 CombinedScattering(float r, float mu, float mu_s, float nu, bool ground, f3& single_mie) const {  // :672
        const f4 uvwz = ScatteringUvwz(r, mu, mu_s, nu, ground);
        const float tex_coord_x = uvwz.x * 7.0f;
        const float tex_x = floorf(tex_coord_x);
        const float lerp = tex_coord_x - tex_x;
        const float u0 = (tex_x + uvwz.y) * 0.125f;
        const float u1 = (tex_x + 1.0f + uvwz.y) * 0.125f;
        const Tap ty = lut_tap<128, false>(uvwz.z), tz = lut_tap<32, false>(uvwz.w);
        const uint32_t r00 = (tz.i0 * 128u + ty.i0) * 256u, r10 = (tz.i0 * 128u + ty.i1) * 256u;
        const uint32_t r01 = (tz.i1 * 128u + ty.i0) * 256u, r11 = (tz.i1 * 128u + ty.i1) * 256u;
        f3 s0, m0, s1, m1;
        fetch_pair(lut_tap<256, false>(u0), r00, r10, r01, r11, ty.a, tz.a, s0, m0);
        fetch_pair(lut_tap<256, false>(u1), r00, r10, r01, r11, ty.a, tz.a, s1, m1);
        single_mie = flerp3(m0, m1, lerp);            // m0 (1 - lerp) + m1 lerp
        return flerp3(s0, s1, lerp);
    }
    // ---- camera-point fast path -------------------------------------------------------------------
    // All but a few samples of a frame look from ONE point (env_pos is the primary-ray origin: the
    // camera position when the aperture is 0, sphere bounces aside), so r and mu_s -- two of the four
    // table coordinates -- are launch constants.  sky_cam_table_kernel pre-interpolates both tables
    // along those two axes; a look-up from the view point then needs 4 table entries (2 mu rows x 2 nu
    // slices) instead of 16 texels per table.  Multilinear interpolation commutes, so this is the same
    // value up to rounding (value-only arithmetic).  Any other view point takes the general path.
    // The tables exist in 2k+1 variants, one per binary32 value of r within k steps of the camera origin's (SkyView): behind an
    // open lens the samples start on the lens disc, whose height spans a few binary32 steps of r (0.5 m at earth-radius magnitude)
    // while mu_s moves by ~1e-7, far below what the mu_s axis of the tables resolves.  Returns the variant, or -1: general path.
    VPT_D int CamVariant(float r, float mu_s) const {
        if (view_k < 0) return -1;
        const int k = (int)(__float_as_uint(r) - __float_as_uint(view_r)) + view_k;
        return (k >= 0 && k <= 2 * view_k && fabsf(mu_s - view_mu_s) <= 1e-6f) ? k : -1;
    }
    VPT_D f3 CombinedScatteringCam(float r, float mu, float nu, bool ground, f3& single_mie, int cv) const {
        // u_mu of ScatteringUvwz (:520-546); u_r and u_mu_s are baked into the table
        const float H = f(AF_H);
        const float rho = SafeSqrt(r * r - bottom() * bottom());
        const float r_mu = r * mu;
        const float disc = r_mu * r_mu - r * r + bottom() * bottom();
        float u_mu;
        if (ground) {
            const float d = -r_mu - SafeSqrt(disc);
            const float d_min = r - bottom();
            const float d_max = rho;
            u_mu = 0.5f - 0.5f * UnitToTex<64>(d_max == d_min ? 0.0f : fdiv(d - d_min, d_max - d_min));
        } else {
            const float d = -r_mu + SafeSqrt(disc + H * H);
            const float d_min = top() - r;
            const float d_max = rho + H;
            u_mu = 0.5f + 0.5f * UnitToTex<64>(fdiv(d - d_min, d_max - d_min));
        }
        const float tex_coord_x = (nu + 1.0f) * 0.5f * 7.0f;
        const float tex_x = floorf(tex_coord_x);
        const float lerp = tex_coord_x - tex_x;
        const uint32_t n0 = (uint32_t)tex_x, n1 = min(n0 + 1u, 7u);
        const Tap ty = lut_tap<128, false>(u_mu);
        const float4* __restrict__ T = view_multi ? R.cam_tab + (uint32_t)cv * (8u * 128u * 2u) : R.cam_tab;
        auto row = [&](uint32_t n, f3& sc, f3& mie) {
            const uint32_t e0 = (n * 128u + ty.i0) * 2u, e1 = (n * 128u + ty.i1) * 2u;
            sc = ld_f3(T, e0);
            mie = ld_f3(T, e0 + 1u);
            if (ty.a != 0.0f) {
                sc = flerp3(sc, ld_f3(T, e1), ty.a);
                mie = flerp3(mie, ld_f3(T, e1 + 1u), ty.a);
            }
        };
        f3 s0, m0;
        row(n0, s0, m0);
        if (lerp != 0.0f) {
            f3 s1, m1;
            row(n1, s1, m1);
            s0 = flerp3(s0, s1, lerp);
            m0 = flerp3(m0, m1, lerp);
        }
        single_mie = m0;
        return s0;
    }
    VPT_D f3 Irradiance(float r, float mu_s) const {                                  // :633-654
        const float x_r = (r - bottom()) * f(AF_INV_TB);
        const float x_mu_s = ffma(mu_s, 0.5f, 0.5f);
        return lut2d(R.irradiance_tex.data, UnitToTex<256>(x_mu_s), UnitToTex<64>(x_r));
    }
    // want_tr: the transmittance is only read for rays inside the sun's disc (sample()); its table look-up is skipped otherwise
    VPT_D f3 SkyRadiance(f3 camera, f3 view_ray, f3 sun_direction, f3& transmittance, int cv, bool want_tr) const {   // :694 (shadow_length = 0)
        float r = length_rn(camera);                       // (the r the table variant `cv` was picked by: sample())
        float rmu = dot(camera, view_ray);
        const float dtop = -rmu - fsqrt(rmu * rmu - r * r + top() * top());
        if (dtop > 0.0f) {
            camera = camera + view_ray * dtop;
            r = top();
            rmu += dtop;
            cv = -1;
        } else if (r > top()) {
            transmittance = mk3(1.0f);
            return mk3(0.0f);
        }
        const float inv_r = frcp(r);
        const float mu = rmu * inv_r;
        const float mu_s = dot(camera, sun_direction) * inv_r;
        const float nu = dot(view_ray, sun_direction);
        const bool ground = HitsGround(r, mu);
        transmittance = mk3(0.0f);
        if (want_tr && !ground) transmittance = TransmittanceToTop(r, mu);
        f3 single_mie;
        const f3 scattering = cv >= 0 ? CombinedScatteringCam(r, mu, nu, ground, single_mie, cv) : CombinedScattering(r, mu, mu_s, nu, ground, single_mie);
        f3 sky = fscale_add3(single_mie, MiePhase(f(AF_MIE_G), nu), scattering * RayleighPhase(nu));
        if (lum()) sky *= v(AF_SKY_K);
        return sky;
    }
    // flipped (optional): the binary32 ground point did NOT land on the ground -- its radius, as :781 forms it, is a step (0.5 m) above `bottom`,
    // i.e. 2.5 km of the tables' rho: the reference's own ray-to-ray noise of a ground hit (~2 % of the rays, ~5e-3 of the radiance each)
    VPT_D f3 SkyRadianceToPoint(f3 camera, f3 point, f3 sun_direction, f3& transmittance, int cv, bool* flipped = nullptr) const {   // :749 (shadow_length = 0)
        // GEOMETRY in the reference's own operations, correctly rounded (round 5): the view ray, r, mu and d feed r_p below and the binary32
        // discriminant (r mu)^2 - r^2 + bottom^2 of the scattering row -- both staircases in which one ulp of a root or a quotient is 0.5 m of
        // radius = 2.5 km of rho, up to 2 % of the radiance next to the horizon.  With them formed as the strict side forms them the ground
        // hits of the two sides land on the same steps (per-pixel outliers against the oracle: see profiles/r05_c5_p99.txt).
        const f3 delta = point - camera;
        const f3 view_ray = normalize_rn(delta);
        float d = length_rn(delta);
        float r = length_rn(camera);
        float rmu = dot(camera, view_ray);
        const float dtop = -rmu - sqrt_rn(rmu * rmu - r * r + top() * top());
        if (dtop > 0.0f) {
            camera = camera + view_ray * dtop;
            r = top();
            rmu += dtop;
            d = length_rn(point - camera);
            cv = -1;
        }
        const float inv_r = frcp(r);
        const float mu = div_rn(rmu, r);
        const float mu_s = dot(camera, sun_direction) * inv_r;
        const float nu = dot(view_ray, sun_direction);
        const bool ground = HitsGround(r, mu);
        const float r_d = ClampRadius(RadiusAt(r, mu, d));
        if (flipped) *flipped = r_d > bottom();
        transmittance = Transmittance(r, mu, d, r_d, ground);
        f3 single_mie;
        f3 scattering = cv >= 0 ? CombinedScatteringCam(r, mu, nu, ground, single_mie, cv) : CombinedScattering(r, mu, mu_s, nu, ground, single_mie);
        d = fmax_(d, 0.0f);
        const float r_p = r_d;
        const float inv_rp = frcp(r_p);
        const float mu_p = (r * mu + d) * inv_rp;
        const float mu_s_p = (r * mu_s + d * nu) * inv_rp;
        f3 single_mie_p;
        const f3 scattering_p = CombinedScattering(r_p, mu_p, mu_s_p, nu, ground, single_mie_p);
        scattering = scattering - transmittance * scattering_p;
        single_mie = single_mie - transmittance * single_mie_p;
        const float y = clampf(mu_s * 100.0f, 0.0f, 1.0f);                           // smoothstep(0, 0.01, mu_s)
        single_mie = single_mie * (y * y * (3.0f - (2.0f * y)));
        f3 sky = fscale_add3(single_mie, MiePhase(f(AF_MIE_G), nu), scattering * RayleighPhase(nu));
        if (lum()) sky *= v(AF_SKY_K);
        return sky;
    }
    // ---- view-point ground table ------------------------------------------------------------------
    // A ray from the view point that ends on the ground (about half of a frame's samples look down) is the expensive case of
    // sample_atmosphere: ground irradiance + sun transmittance, a transmittance ratio, the scattering at both ends.  With the view
    // point and the sun fixed, everything but the view-point scattering (kept per sample: the camera-point table above) is a
    // function of two scalars -- the distance d to the ground (equivalently mu) and nu = view . sun -- apart from the Mie phase
    // function, which is sharply peaked in nu and stays analytic:
    //     radiance = A(d, nu) + S_view RayleighPhase(nu) + (M_view smoothstep(mu_s) + B(d, nu)) MiePhase(nu)
    //     A = ground_radiance T - T S_ground RayleighPhase(nu),   B = -T M_ground smoothstep(mu_s)
    // (S_view's mu row comes from r^2 mu^2 - r^2 + bottom^2 in binary32, as in the reference: a staircase in mu next to the
    // horizon that no table could follow; A and B are smooth.)  GroundNode evaluates A and B with the very functions of the full
    // path; sky_dir_table_kernel tabulates them on DT_NX x DT_NN nodes (log d uniform between its extremes r - bottom, straight
    // down, and the horizon distance: 1-2 % per cell both in the view angle, mu ~ -(r - bottom) / d near the nadir, and in the
    // path length that the transmittance decays with near the horizon; nu linear), a second kernel evaluates every reachable
    // cell CENTRE of the part that is used (the grazing end of the d range is left to the full path: there the reference's own
    // ground test flips between its binary32 and binary64 forms) in full and records the largest deviation of the bilinear
    // interpolant relative to the radiance there; the tail uses the table only while that figure is below
    // ResolveParams::dir_tab_tol (5e-4; the image tolerance is 1e-3) and evaluates in full otherwise.  VALUE-ONLY like
    // everything in this file: a cache of a pure function with a measured error bound, not a re-association.
#ifndef VPT_DT_NX
#define VPT_DT_NX 512
#define VPT_DT_NN 64
#endif
    enum { DT_NX = VPT_DT_NX, DT_NN = VPT_DT_NN };
    // scale: radiance scale of the node (|A| + |full ground radiance|), the denominator of the relative error
    VPT_D void GroundNode(float r, float mu_s, float x, float nu, f3& A, f3& B, float& scale) const {
        const float b = bottom();
        const float r_p = b;                                                         // the ground point is ON the ground (see below)
        const float h2 = (r - b) * (r + b);                                          // r^2 - bottom^2 without the cancellation
        const float d_min = r - b, d_max = fsqrt(fmax_(h2, 0.0f));
        const float d = d_min * __builtin_amdgcn_exp2f(x * __builtin_amdgcn_logf(fdiv(d_max, d_min)));      // d_min (d_max / d_min)^x
        const float mu = ClampCosine(fdiv(-(h2 + d * d), 2.0f * r * d));             // bottom^2 = r^2 + d^2 + 2 r d mu
        const float inv_rp = frcp(r_p);
        const float mu_p = (r * mu + d) * inv_rp;
        const float mu_s_p = (r * mu_s + d * nu) * inv_rp;
        f3 sky_irr = Irradiance(r_p, mu_s_p);                                        // (1 + normal . pt / r) / 2 = 1
        f3 sun_irr = v(AF_SOLAR) * TransmittanceToSun(r_p, mu_s_p) * fmax_(mu_s_p, 0.0f);
        if (lum()) { sky_irr *= v(AF_SKY_K); sun_irr *= v(AF_SUN_K); }
        const f3 ground = v(AF_GROUND) * (1.0f / VPT_PI) * (sun_irr + sky_irr);
        const f3 T = Transmittance(r, mu, d, r_p, true);
        f3 m_v, m_p;
        const f3 s_v = CombinedScattering(r, mu, mu_s, nu, true, m_v);               // only for the scale
        const f3 s_p = CombinedScattering(r_p, mu_p, mu_s_p, nu, true, m_p);
        const float y = clampf(mu_s * 100.0f, 0.0f, 1.0f);
        const float sm = y * y * (3.0f - (2.0f * y));
        f3 sc = T * s_p * RayleighPhase(nu);
        f3 mie = T * m_p * sm;
        f3 full = (s_v - T * s_p) * RayleighPhase(nu) + (m_v - T * m_p) * (sm * MiePhase(f(AF_MIE_G), nu));
        if (lum()) { sc *= v(AF_SKY_K); mie *= v(AF_SKY_K); full *= v(AF_SKY_K); }
        A = ground * T - sc;
        B = mie * -1.0f;
        full = full + ground * T;
        scale = fmax_(fmax_(fabsf(full.x), fabsf(full.y)), fabsf(full.z));
    }
    VPT_D static void DirTabCoords(float fx, float fn, uint32_t& e, float& ax, float& an) {
        const int ix = min((int)fx, DT_NX - 2), in = min((int)fn, DT_NN - 2);
        ax = fx - (float)ix;
        an = fn - (float)in;
        e = ((uint32_t)ix * DT_NN + (uint32_t)in) * 2u;
    }
    VPT_D static void DirTabLerp(const float4* __restrict__ T, uint32_t e, float ax, float an, f3& A, f3& B) {
        const uint32_t up = 2u * DT_NN;
        const f3 a0 = flerp3(ld_f3(T, e), ld_f3(T, e + 2u), an), b0 = flerp3(ld_f3(T, e + 1u), ld_f3(T, e + 3u), an);
        const f3 a1 = flerp3(ld_f3(T, e + up), ld_f3(T, e + up + 2u), an), b1 = flerp3(ld_f3(T, e + up + 1u), ld_f3(T, e + up + 3u), an);
        A = flerp3(a0, a1, ax);
        B = flerp3(b0, b1, ax);
    }
    // p: view point - earth centre, pt: ground point - earth centre as sample() forms it.  The geometry is that of the full
    // path: the ground point is rounded to binary32 at earth-radius magnitude (0.5 m grid) and the view ray, its length d,
    // mu and nu are re-derived from it (SkyRadianceToPoint) -- for a camera a few metres above the ground that quantisation IS
    // the reference's value.  What is NOT followed is the radius of that rounded point, which the full path clamps to bottom or
    // finds one binary32 step (0.5 m) above it: half a metre of radius is 2.5 km of the tables' rho coordinate, and behind it the
    // scattering row of the ground point becomes a staircase of binary32 cancellation (r mu)^2 - r^2 + bottom^2 (following the
    // step in the transmittance alone, through a table of ratios over d, changed nothing: measured) -- up to 2 % of the radiance
    // within 2 degrees of the horizon, <= 0.8 % (1e-5 typically) below that from the camera origin, ~1 % for a few per cent of the
    // origins on an open lens' disc.  The table is therefore used for rays at least ~2 degrees below the horizon only
    // (d <= min(33 (r - bottom), 0.35 horizon distance): SkyView::tab[].z) and takes the ground point on the ground; returns
    // false (evaluate in full) for everything else.
    VPT_D bool GroundFromTable(f3 p, float r, float mu_s, f3 pt, f3 sun_direction, int cv, f3& radiance) const {
        const f3 delta = pt - p;
        const float dist = length_rn(delta);
        const f3 view_ray = normalize_rn(delta);                                     // (geometry correctly rounded: SkyRadianceToPoint)
        const float mu = div_rn(dot(p, view_ray), r);
        const float4 vt = view_multi ? R.sky_view->tab[cv] : view_vt0;               // 1 / d_min, 1 / log2(d_max / d_min), x_use, has a table
        const float fx = __builtin_amdgcn_logf(dist * vt.x) * vt.y;
        // (within x_use |mu| is at least 1.6 times the horizon's: the double-precision ground test of :401 holds)
        if (!(fx <= vt.z) || vt.w == 0.0f) return false;
        const float nu = dot(view_ray, sun_direction);
        f3 m_v;
        f3 s_v = CombinedScatteringCam(r, mu, nu, true, m_v, cv);
        const float y = clampf(mu_s * 100.0f, 0.0f, 1.0f);
        m_v = m_v * (y * y * (3.0f - (2.0f * y)));
        if (lum()) { s_v *= v(AF_SKY_K); m_v *= v(AF_SKY_K); }
        const float fn = clampf(ffma(nu, 0.5f, 0.5f), 0.0f, 1.0f) * (float)(DT_NN - 1);
        uint32_t e; float ax, an;
        DirTabCoords(fmax_(fx, 0.0f) * (float)(DT_NX - 1), fn, e, ax, an);
        f3 A, B;
        DirTabLerp(view_multi ? R.dir_tab + (uint32_t)cv * (2u * DT_NX * DT_NN) : R.dir_tab, e, ax, an, A, B);
        radiance = fscale_add3(m_v + B, MiePhase(f(AF_MIE_G), nu), fscale_add3(s_v, RayleighPhase(nu), A));
        return true;
    }
    // sample_atmosphere :839-895.  use_dir_tab: ground hits seen from the table's view point come from the ground table
    // kind (optional): which evaluation the direction took -- 0 sky, 1 ground through the table, 2 ground in full (the one whose
    // binary32 ground point makes it noisy from ray to ray: the per-pixel sky patches keep away from it)
    VPT_D f3 sample(f3 ray_pos, f3 ray_dir, f3 sun_direction, bool use_dir_tab = false, int* kind = nullptr, bool* flipped = nullptr) const {
        const f3 earth_center = mk3(.0f, -bottom(), .0f);
        const f3 p = ray_pos - earth_center;
        const float p_dot_v = dot(p, ray_dir);
        const float p_dot_p = dot(p, p);
        const float d2 = p_dot_p - p_dot_v * p_dot_v;
        const float dist = -p_dot_v - sqrt_rn(earth_center.y * earth_center.y - d2);      // (correctly rounded: one ulp of this root moves the ground point 0.5 m along the ray)
        f3 radiance;
        int cv = -1;
        float r_view = 0.0f, mu_s_view = 0.0f;
        if (view_k >= 0 && !view_multi) {
            // closed lens: one variant, the camera origin itself
            if (ray_pos.x == R.cam_tab_pos[0] && ray_pos.y == R.cam_tab_pos[1] && ray_pos.z == R.cam_tab_pos[2]) {
                cv = 0;
                r_view = view_r;
                mu_s_view = view_mu_s;
            }
        } else if (view_k >= 0) {
            r_view = length_rn(p);                                                    // (its binary32 value picks the table variant)
            mu_s_view = dot(p, sun_direction) * frcp(r_view);
            cv = CamVariant(r_view, mu_s_view);
        }
        if (dist > 0.0f && use_dir_tab && cv >= 0 && GroundFromTable(p, r_view, mu_s_view, ray_pos + ray_dir * dist - earth_center, sun_direction, cv, radiance)) {
            // radiance from the view-point ground table
            if (kind) *kind = 1;
        } else if (dist > 0.0f) {
            if (kind) *kind = 2;
            const f3 pt = ray_pos + ray_dir * dist - earth_center;
            const float r = length_rn(pt);
            const float inv_r = frcp(r);
            const f3 normal = pt * inv_r;
            const float mu_s = dot(pt, sun_direction) * inv_r;
            f3 sky_irr = Irradiance(r, mu_s) * ((1.0f + dot(normal, pt) * inv_r) * 0.5f);     // :818
            f3 sun_irr = v(AF_SOLAR) * TransmittanceToSun(r, mu_s) * fmax_(dot(normal, sun_direction), 0.0f);
            if (lum()) { sky_irr *= v(AF_SKY_K); sun_irr *= v(AF_SUN_K); }
            radiance = v(AF_GROUND) * (1.0f / VPT_PI) * (sun_irr + sky_irr);
            f3 tr;
            const f3 in_scatter = SkyRadianceToPoint(p, pt, sun_direction, tr, cv, flipped);
            radiance = radiance * tr + in_scatter;
            // lerp(radiance_sky, ground_radiance, ground_alpha = 1) (:881) is ground_radiance to one
            // rounding: the sky-only branch below is not evaluated for ground hits
        } else {
            if (kind) *kind = 0;
            f3 tr_sky;
            const bool in_disc = dot(ray_dir, sun_direction) > f(AF_COS_SUN);
            radiance = SkyRadiance(p, ray_dir, sun_direction, tr_sky, cv, in_disc);
            if (in_disc) radiance = radiance + tr_sky * v(AF_SOLAR_RAD);
        }
        // pow(1 - exp(-radiance / white_point * exposure), 1 / 2.2)   (:883-885)
        const f3 e = radiance * v(AF_EXPO_W);
        const float l2e = 1.4426950408889634f, g = (float)(1.0 / 2.2);
        const f3 om = mk3(1.0f - __builtin_amdgcn_exp2f(-e.x * l2e), 1.0f - __builtin_amdgcn_exp2f(-e.y * l2e), 1.0f - __builtin_amdgcn_exp2f(-e.z * l2e));
        return mk3(__builtin_amdgcn_exp2f(g * __builtin_amdgcn_logf(om.x)), __builtin_amdgcn_exp2f(g * __builtin_amdgcn_logf(om.y)),
                   __builtin_amdgcn_exp2f(g * __builtin_amdgcn_logf(om.z)));
    }
};

// One entry of the camera-point table: both 4-D tables interpolated along r (z) and mu_s (x within a nu
// slice) at the view point's (r, mu_s), for mu row j and nu slice n.
template <class RP>
VPT_D void sky_cam_table_entry(const RP& R, float r, float mu_s, uint32_t n, uint32_t j, f3& sc, f3& mie) {
    const Sky<RP> sky = {R};
    const f4 uvwz = sky.ScatteringUvwz(r, 0.0f, mu_s, 0.0f, false);          // only u_mu_s (y) and u_r (w) are used
    const Tap tz = lut_tap<32, false>(uvwz.w);
    const Tap tm = lut_tap<256, false>(uvwz.y * 0.125f);                       // the tap inside nu slice 0: index < 32
    const float4* __restrict__ ps = reinterpret_cast<const float4*>(R.scattering_tex.data);
    const float4* __restrict__ pm = reinterpret_cast<const float4*>(R.single_mie_tex.data);
    const uint32_t x0 = n * 32u + tm.i0, x1 = n * 32u + min(tm.i1, 31u);
    const uint32_t r0 = (tz.i0 * 128u + j) * 256u, r1 = (tz.i1 * 128u + j) * 256u;
    sc = flerp3(flerp3(ld_f3(ps, r0 + x0), ld_f3(ps, r0 + x1), tm.a), flerp3(ld_f3(ps, r1 + x0), ld_f3(ps, r1 + x1), tm.a), tz.a);
    mie = flerp3(flerp3(ld_f3(pm, r0 + x0), ld_f3(pm, r0 + x1), tm.a), flerp3(ld_f3(pm, r1 + x0), ld_f3(pm, r1 + x1), tm.a), tz.a);
}

}  // namespace vpt
