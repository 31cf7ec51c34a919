// vpt_sky.h -- Bruneton precomputed atmospheric scattering, look-up side: `sample_atmosphere`
// (render_kernel.cu:839-895) and the functions under it (:369-835).  VALUE-ONLY arithmetic (see
// vpt_tail.hip); used by the environment tail (vpt_tail.hip) and by estimate_sky in the
// vol_integrator tracer (vpt_trace_vol.hip).  RP is any parameter block with atm_f[] and the four
// look-up textures (ResolveParams, TraceParams).
#pragma once

#include "vpt_tex.h"

namespace vpt {

// ---- Bruneton precomputed atmospheric scattering, look-up side ------------------------------
// (render_kernel.cu:369-895; published algorithm: E. Bruneton, "Precomputed Atmospheric
// Scattering", EGSR 2008 + 2017 reference implementation `functions.glsl`.)
// atm_f[] packing is defined in vpt_host.hip (pack_atmosphere).
//
// The four tables have fixed power-of-two extents (constants.h:50-62), so their samplers are
// specialised at compile time: "linear, normalised, wrap/clamp" for the 2-D tables
// (atmosphere.cpp:503-573) and "linear, normalised, clamp" for the 3-D ones (:575-675), with the
// y/z taps shared between the two nu slices and the two 3-D tables of GetCombinedScattering.
enum {
    AF_BOTTOM = 0, AF_TOP = 1, AF_USE_LUM = 2, AF_MIE_G = 3, AF_SUN_ANG = 4, AF_MU_S_MIN = 5, AF_EXPOSURE = 6,
    AF_SKY_K = 8, AF_SUN_K = 11, AF_SOLAR = 14, AF_GROUND = 17, AF_WHITE = 20,
};
VPT_D f3 ld_f3(const float4* p, int i) { const float4 v = p[i]; return mk3(v.x, v.y, v.z); }
VPT_D f3 lerp3r(f3 a, f3 b, float t) { return a + (b - a) * t; }
struct Tap { int i0, i1; float a; };
template <int N, bool WRAP>
VPT_D Tap lut_tap(float u) {
    Tap t;
    const float xb = u * (float)N - 0.5f;
    const float fl = floorf(xb);
    t.a = xb - fl;
    const int i = (int)fl;
    if (WRAP) {
        t.i0 = i & (N - 1);
        t.i1 = (i + 1) & (N - 1);
    } else {
        t.i0 = min(max(i, 0), N - 1);
        t.i1 = min(max(i + 1, 0), N - 1);
    }
    return t;
}
// 2-D 256x64 float4 table, x wraps, y clamps
VPT_D f3 lut2d(const float* data, float u, float v) {
    const float4* p = reinterpret_cast<const float4*>(data);
    const Tap tx = lut_tap<256, true>(u), ty = lut_tap<64, false>(v);
    const f3 c0 = lerp3r(ld_f3(p, ty.i0 * 256 + tx.i0), ld_f3(p, ty.i0 * 256 + tx.i1), tx.a);
    const f3 c1 = lerp3r(ld_f3(p, ty.i1 * 256 + tx.i0), ld_f3(p, ty.i1 * 256 + tx.i1), tx.a);
    return lerp3r(c0, c1, ty.a);
}
template <class RP>
struct Sky {
    const RP& R;
    VPT_D float f(int i) const { return R.atm_f[i]; }
    VPT_D f3 v(int i) const { return mk3(R.atm_f[i], R.atm_f[i + 1], R.atm_f[i + 2]); }
    VPT_D float bottom() const { return f(AF_BOTTOM); }
    VPT_D float top() const { return f(AF_TOP); }
    VPT_D bool lum() const { return f(AF_USE_LUM) != 0.0f; }

    VPT_D static float ClampCosine(float mu) { return clampf(mu, -1.0f, 1.0f); }
    VPT_D float ClampRadius(float r) const { return clampf(r, bottom(), top()); }
    VPT_D static float SafeSqrt(float a) { return sqrtf(fmax_(a, 0.0f)); }
    VPT_D float DistanceToTop(float r, float mu) const {                              // :389
        float disc = (float)((double)(r * r) * ((double)(mu * mu) - 1.0) + (double)(top() * top()));
        return fmax_(-r * mu + SafeSqrt(disc), 0.0f);
    }
    VPT_D bool HitsGround(float r, float mu) const {                                  // :401
        return mu < 0.0f && (double)(r * r) * ((double)(mu * mu) - 1.0) + (double)(bottom() * bottom()) >= 0.0;
    }
    template <int N>
    VPT_D static float UnitToTex(float x) {                                           // :419
        return (float)(0.5 / (double)N + (double)x * (1.0 - 1.0 / (double)N));
    }
    VPT_D f3 TransmittanceToTop(float r, float mu) const {                            // :429-470
        float H = sqrtf(top() * top() - bottom() * bottom());
        float rho = SafeSqrt(r * r - bottom() * bottom());
        float d = DistanceToTop(r, mu);
        float d_min = top() - r;
        float d_max = rho + H;
        float x_mu = (d - d_min) / (d_max - d_min);
        float x_r = rho / H;
        return lut2d(R.transmittance_tex.data, UnitToTex<256>(x_mu), UnitToTex<64>(x_r));
    }
    VPT_D f3 Transmittance(float r, float mu, float d, bool ground) const {           // :472
        float r_d = ClampRadius((float)sqrt((double)(d * d) + 2.0 * (double)r * (double)mu * (double)d + (double)(r * r)));
        float mu_d = ClampCosine((r * mu + d) / r_d);
        if (ground) return fmin3(TransmittanceToTop(r_d, -mu_d) / TransmittanceToTop(r, -mu), mk3(1.0f));
        return fmin3(TransmittanceToTop(r, mu) / TransmittanceToTop(r_d, mu_d), mk3(1.0f));
    }
    VPT_D f3 TransmittanceToSun(float r, float mu_s) const {                          // :486
        float sin_theta_h = bottom() / r;
        float cos_theta_h = -sqrtf(fmax_(1.0f - sin_theta_h * sin_theta_h, 0.0f));
        float sa = f(AF_SUN_ANG);
        return TransmittanceToTop(r, mu_s) * smoothstep(-sin_theta_h * sa, sin_theta_h * sa, mu_s - cos_theta_h);
    }
    VPT_D static float RayleighPhase(float nu) {                                      // :508
        float k = 3.0f / (16.0f * VPT_PI);
        return k * (1.0f + nu * nu);
    }
    VPT_D static float MiePhase(float g, float nu) {                                  // :514
        float k = 3.0f / (8.0f * VPT_PI) * (1.0f - g * g) / (2.0f + g * g);
        const float b = 1.0f + g * g - 2.0f * g * nu;
        return k * (1.0f + nu * nu) / (b * sqrtf(b));     // pow(b, 1.5)
    }
    VPT_D f4 ScatteringUvwz(float r, float mu, float mu_s, float nu, bool ground) const {   // :520-569
        float H = sqrtf(top() * top() - bottom() * bottom());
        float rho = SafeSqrt(r * r - bottom() * bottom());
        float u_r = UnitToTex<32>(rho / H);
        float r_mu = r * mu;
        float disc = r_mu * r_mu - r * r + bottom() * bottom();
        float u_mu;
        if (ground) {
            float d = -r_mu - SafeSqrt(disc);
            float d_min = r - bottom();
            float d_max = rho;
            u_mu = 0.5f - 0.5f * UnitToTex<64>(d_max == d_min ? 0.0f : (d - d_min) / (d_max - d_min));
        } else {
            float d = -r_mu + SafeSqrt(disc + H * H);
            float d_min = top() - r;
            float d_max = rho + H;
            u_mu = 0.5f + 0.5f * UnitToTex<64>((d - d_min) / (d_max - d_min));
        }
        float d = DistanceToTop(bottom(), mu_s);
        float d_min = top() - bottom();
        float d_max = H;
        float a = (d - d_min) / (d_max - d_min);
        float A = -2.0f * f(AF_MU_S_MIN) * bottom() / (d_max - d_min);
        float u_mu_s = UnitToTex<32>(fmax_(1.0f - a / A, 0.0f) / (1.0f + a));
        float u_nu = (nu + 1.0f) / 2.0f;
        return mk4(u_nu, u_mu_s, u_mu, u_r);
    }
    // bilinear (y, z) x linear (x) fetch of BOTH 3-D tables at one u, sharing the taps
    VPT_D void fetch_pair(const Tap& tx, const Tap& ty, const Tap& tz, f3& sc, f3& mie) const {
        const float4* ps = reinterpret_cast<const float4*>(R.scattering_tex.data);
        const float4* pm = reinterpret_cast<const float4*>(R.single_mie_tex.data);
        const int r00 = (tz.i0 * 128 + ty.i0) * 256, r10 = (tz.i0 * 128 + ty.i1) * 256;
        const int r01 = (tz.i1 * 128 + ty.i0) * 256, r11 = (tz.i1 * 128 + ty.i1) * 256;
        {
            const f3 c00 = lerp3r(ld_f3(ps, r00 + tx.i0), ld_f3(ps, r00 + tx.i1), tx.a);
            const f3 c10 = lerp3r(ld_f3(ps, r10 + tx.i0), ld_f3(ps, r10 + tx.i1), tx.a);
            const f3 c01 = lerp3r(ld_f3(ps, r01 + tx.i0), ld_f3(ps, r01 + tx.i1), tx.a);
            const f3 c11 = lerp3r(ld_f3(ps, r11 + tx.i0), ld_f3(ps, r11 + tx.i1), tx.a);
            sc = lerp3r(lerp3r(c00, c10, ty.a), lerp3r(c01, c11, ty.a), tz.a);
        }
        {
            const f3 c00 = lerp3r(ld_f3(pm, r00 + tx.i0), ld_f3(pm, r00 + tx.i1), tx.a);
            const f3 c10 = lerp3r(ld_f3(pm, r10 + tx.i0), ld_f3(pm, r10 + tx.i1), tx.a);
            const f3 c01 = lerp3r(ld_f3(pm, r01 + tx.i0), ld_f3(pm, r01 + tx.i1), tx.a);
            const f3 c11 = lerp3r(ld_f3(pm, r11 + tx.i0), ld_f3(pm, r11 + tx.i1), tx.a);
            mie = lerp3r(lerp3r(c00, c10, ty.a), lerp3r(c01, c11, ty.a), tz.a);
        }
    }
    VPT_D f3 CombinedScattering(float r, float mu, float mu_s, float nu, bool ground, f3& single_mie) const {  // :672
        f4 uvwz = ScatteringUvwz(r, mu, mu_s, nu, ground);
        float tex_coord_x = uvwz.x * 7.0f;
        float tex_x = floorf(tex_coord_x);
        float lerp = tex_coord_x - tex_x;
        float u0 = (tex_x + uvwz.y) / 8.0f;
        float u1 = (tex_x + 1.0f + uvwz.y) / 8.0f;
        float l0 = 1.0f - lerp;
        const Tap ty = lut_tap<128, false>(uvwz.z), tz = lut_tap<32, false>(uvwz.w);
        f3 s0, m0, s1, m1;
        fetch_pair(lut_tap<256, false>(u0), ty, tz, s0, m0);
        fetch_pair(lut_tap<256, false>(u1), ty, tz, s1, m1);
        single_mie = m0 * l0 + m1 * lerp;
        return s0 * l0 + s1 * lerp;
    }
    VPT_D f3 Irradiance(float r, float mu_s) const {                                  // :633-654
        float x_r = (r - bottom()) / (top() - bottom());
        float x_mu_s = mu_s * 0.5f + 0.5f;
        return lut2d(R.irradiance_tex.data, UnitToTex<256>(x_mu_s), UnitToTex<64>(x_r));
    }
    VPT_D f3 SkyRadiance(f3 camera, f3 view_ray, f3 sun_direction, f3& transmittance) const {   // :694 (shadow_length = 0)
        float r = length(camera);
        float rmu = dot(camera, view_ray);
        float dtop = -rmu - sqrtf(rmu * rmu - r * r + top() * top());
        if (dtop > 0.0f) {
            camera = camera + view_ray * dtop;
            r = top();
            rmu += dtop;
        } else if (r > top()) {
            transmittance = mk3(1.0f);
            return mk3(0.0f);
        }
        float mu = rmu / r;
        float mu_s = dot(camera, sun_direction) / r;
        float nu = dot(view_ray, sun_direction);
        bool ground = HitsGround(r, mu);
        transmittance = ground ? mk3(0.0f) : TransmittanceToTop(r, mu);
        f3 single_mie;
        f3 scattering = CombinedScattering(r, mu, mu_s, nu, ground, single_mie);
        f3 sky = scattering * RayleighPhase(nu) + single_mie * MiePhase(f(AF_MIE_G), nu);
        if (lum()) sky *= v(AF_SKY_K);
        return sky;
    }
    VPT_D f3 SkyRadianceToPoint(f3 camera, f3 point, f3 sun_direction, f3& transmittance) const {   // :749 (shadow_length = 0)
        f3 view_ray = normalize(point - camera);
        float r = length(camera);
        float rmu = dot(camera, view_ray);
        float dtop = -rmu - sqrtf(rmu * rmu - r * r + top() * top());
        if (dtop > 0.0f) {
            camera = camera + view_ray * dtop;
            r = top();
            rmu += dtop;
        }
        float mu = rmu / r;
        float mu_s = dot(camera, sun_direction) / r;
        float nu = dot(view_ray, sun_direction);
        float d = length(point - camera);
        bool ground = HitsGround(r, mu);
        transmittance = Transmittance(r, mu, d, ground);
        f3 single_mie;
        f3 scattering = CombinedScattering(r, mu, mu_s, nu, ground, single_mie);
        d = fmax_(d, 0.0f);
        float r_p = ClampRadius((float)sqrt((double)(d * d) + 2.0 * (double)r * (double)mu * (double)d + (double)(r * r)));
        float mu_p = (r * mu + d) / r_p;
        float mu_s_p = (r * mu_s + d * nu) / r_p;
        f3 single_mie_p;
        f3 scattering_p = CombinedScattering(r_p, mu_p, mu_s_p, nu, ground, single_mie_p);
        f3 shadow_t = transmittance;
        scattering = scattering - shadow_t * scattering_p;
        single_mie = single_mie - shadow_t * single_mie_p;
        single_mie = single_mie * smoothstep(0.0f, 0.01f, mu_s);
        f3 sky = scattering * RayleighPhase(nu) + single_mie * MiePhase(f(AF_MIE_G), nu);
        if (lum()) sky *= v(AF_SKY_K);
        return sky;
    }
    // sample_atmosphere :839-895
    VPT_D f3 sample(f3 ray_pos, f3 ray_dir, f3 sun_direction) const {
        f3 earth_center = mk3(.0f, -bottom(), .0f);
        f3 p = ray_pos - earth_center;
        float p_dot_v = dot(p, ray_dir);
        float p_dot_p = dot(p, p);
        float d2 = p_dot_p - p_dot_v * p_dot_v;
        float dist = -p_dot_v - sqrtf(earth_center.y * earth_center.y - d2);
        float ground_alpha = 0.0f;
        f3 ground_radiance = mk3(0.0f);
        if (dist > 0.0f) {
            f3 point = ray_pos + ray_dir * dist;
            f3 normal = normalize(point - earth_center);
            f3 pt = point - earth_center;
            float r = length(pt);
            float mu_s = dot(pt, sun_direction) / r;
            f3 sky_irr = Irradiance(r, mu_s) * ((1.0f + dot(normal, pt) / r) * 0.5f);     // :818
            f3 sun_irr = v(AF_SOLAR) * TransmittanceToSun(r, mu_s) * fmax_(dot(normal, sun_direction), 0.0f);
            if (lum()) { sky_irr *= v(AF_SKY_K); sun_irr *= v(AF_SUN_K); }
            ground_radiance = v(AF_GROUND) * (1.0f / VPT_PI) * (sun_irr + sky_irr);
            f3 tr;
            f3 in_scatter = SkyRadianceToPoint(ray_pos - earth_center, pt, sun_direction, tr);
            ground_radiance = ground_radiance * tr + in_scatter;
            ground_alpha = 1.0f;
        }
        if (ground_alpha == 0.0f) {
            // lerp(radiance_sky, ground_radiance, ground_alpha) (:881): with the ground hit the blend weight
            // is exactly 1 and the sky-only radiance only enters as a + (b - a), i.e. b to within one
            // rounding -- it is not evaluated then (value-only path)
            f3 tr_sky;
            f3 radiance_sky = SkyRadiance(ray_pos - earth_center, ray_dir, sun_direction, tr_sky);
            float sa = f(AF_SUN_ANG);
            if (dot(ray_dir, sun_direction) > cosf(sa)) {
                f3 solar = v(AF_SOLAR) / (VPT_PI * sa * sa);
                if (lum()) solar *= v(AF_SUN_K);
                radiance_sky = radiance_sky + tr_sky * solar;
            }
            ground_radiance = radiance_sky;
        }
        f3 exposure = lum() ? mk3(f(AF_EXPOSURE)) * 1e-5f : mk3(f(AF_EXPOSURE));
        f3 e = -ground_radiance / v(AF_WHITE) * exposure;
        f3 om = mk3(1.0f) - mk3(__expf(e.x), __expf(e.y), __expf(e.z));
        const float g = (float)(1.0 / 2.2);
        return mk3(__powf(om.x, g), __powf(om.y, g), __powf(om.z, g));
    }
};

}  // namespace vpt
