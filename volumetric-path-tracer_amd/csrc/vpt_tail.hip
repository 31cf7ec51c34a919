// vpt_tail.hip -- stage 2 of the hot path: the environment tail of direct_integrator
// (render_kernel.cu:1838-1850) for every pixel-sample of a batch: Bruneton sky `sample_atmosphere`
// (:839-895) or the lat-long HDRI.
//
// This is VALUE-ONLY arithmetic: nothing downstream branches on it that feeds a random walk, the
// result is added to L once.  The reference evaluates it with `--use_fast_math`
// (source/CMakeLists.txt:133); this translation unit is likewise built with approximate fp32
// divide/sqrt and FMA contraction and uses the hardware exp/log/pow (see build.py) -- unlike the
// strict-arithmetic tracer.  Tolerance against the oracle: tests/test_gpu_atmosphere.py.
#include <hip/hip_runtime.h>

#include "vpt_device.h"

namespace vpt {

// ---- generic sampler (CUDA texture addressing) -----------------------
VPT_D int tex_addr(int i, int n, int mode, int normalized) {
    if (mode == 0 && normalized) {            // wrap (only honoured for normalised coordinates)
        int r = i % n;
        return r < 0 ? r + n : r;
    }
    return i < 0 ? 0 : (i > n - 1 ? n - 1 : i);
}
struct AxisTap { int i0, i1; float a; };
VPT_D AxisTap axis_tap(float u, int n, int mode, int normalized, int linear) {
    AxisTap r;
    float x = normalized ? u * (float)n : u;
    if (linear) {
        float xb = x - 0.5f;
        float fl = floorf(xb);
        r.a = xb - fl;
        int i = (int)fl;
        r.i0 = tex_addr(i, n, mode, normalized);
        r.i1 = tex_addr(i + 1, n, mode, normalized);
    } else {
        int i = (int)floorf(x);
        r.a = 0.0f;
        r.i0 = r.i1 = tex_addr(i, n, mode, normalized);
    }
    return r;
}
VPT_D f4 texel4(const DTexture& t, int x, int y, int z) {
    size_t idx = ((size_t)z * t.height + y) * t.width + x;
    if (t.channels == 1) return mk4(t.data[idx], 0.0f, 0.0f, 0.0f);
    const float4 v = reinterpret_cast<const float4*>(t.data)[idx];
    return mk4(v.x, v.y, v.z, v.w);
}
VPT_D f4 lerp4r(f4 a, f4 b, float t) { return a + (b - a) * t; }
VPT_D f4 tex2d(const DTexture& t, float u, float v) {
    AxisTap ax = axis_tap(u, t.width, t.addr[0], t.normalized, t.linear);
    AxisTap ay = axis_tap(v, t.height, t.addr[1], t.normalized, t.linear);
    if (!t.linear) return texel4(t, ax.i0, ay.i0, 0);
    f4 c0 = lerp4r(texel4(t, ax.i0, ay.i0, 0), texel4(t, ax.i1, ay.i0, 0), ax.a);
    f4 c1 = lerp4r(texel4(t, ax.i0, ay.i1, 0), texel4(t, ax.i1, ay.i1, 0), ax.a);
    return lerp4r(c0, c1, ay.a);
}
VPT_D f4 tex3d(const DTexture& t, float u, float v, float w) {
    AxisTap ax = axis_tap(u, t.width, t.addr[0], t.normalized, t.linear);
    AxisTap ay = axis_tap(v, t.height, t.addr[1], t.normalized, t.linear);
    AxisTap az = axis_tap(w, t.depth, t.addr[2], t.normalized, t.linear);
    if (!t.linear) return texel4(t, ax.i0, ay.i0, az.i0);
    f4 c00 = lerp4r(texel4(t, ax.i0, ay.i0, az.i0), texel4(t, ax.i1, ay.i0, az.i0), ax.a);
    f4 c10 = lerp4r(texel4(t, ax.i0, ay.i1, az.i0), texel4(t, ax.i1, ay.i1, az.i0), ax.a);
    f4 c01 = lerp4r(texel4(t, ax.i0, ay.i0, az.i1), texel4(t, ax.i1, ay.i0, az.i1), ax.a);
    f4 c11 = lerp4r(texel4(t, ax.i0, ay.i1, az.i1), texel4(t, ax.i1, ay.i1, az.i1), ax.a);
    return lerp4r(lerp4r(c00, c10, ay.a), lerp4r(c01, c11, ay.a), az.a);
}

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
struct Sky {
    const ResolveParams& R;
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

// stage 2: environment tail, one thread per pixel-sample (every sample of the batch, including
// the primary-ray misses finalised by raygen).  Rewrites the first 16 bytes of the record to
// {value.xyz, tr}; the rest of the line is left alone.
__global__ __launch_bounds__(256) void tail_kernel(const ResolveParams R) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= R.n_pixels * R.iter_count) return;
    float4* rec = reinterpret_cast<float4*>(const_cast<Record*>(R.records) + s);
    const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2], q3 = rec[3];
    f3 value = mk3(q0.x, q0.y, q0.z);
    const f3 beta = mk3(q1.x, q1.y, q1.z);
    const f3 env_pos = mk3(q2.x, q2.y, q2.z);
    const uint32_t flags = __float_as_uint(q2.w);
    const f3 dir = mk3(q3.x, q3.y, q3.z);
    if (flags & 1u) {
        const f3 sky_color = mk3(R.sky_color[0], R.sky_color[1], R.sky_color[2]);
        if (R.environment_type == 0) {                                              // :1838-1842
            if (R.has_atmosphere) {
                const Sky sky = {R};
                const f3 sun_dir = mk3(R.sun_dir[0], R.sun_dir[1], R.sun_dir[2]);
                value += sky.sample(env_pos, dir, sun_dir) * beta * R.sky_mult * sky_color;
            }
        } else {                                                                     // :1843-1850
            f4 t = tex2d(R.env_tex, atan2f(dir.z, dir.x) * (float)(0.5 / (double)VPT_PI) + 0.5f,
                         acosf(fmax_(fmin_(dir.y, 1.0f), -1.0f)) * (float)(1.0 / (double)VPT_PI));
            value += xyz(t) * sky_color * beta * (1.0f / (4.0f * VPT_PI));
        }
    }
    rec[0] = make_float4(value.x, value.y, value.z, q0.w);
}

hipError_t launch_tail(const ResolveParams& R, hipStream_t stream) {
    const uint32_t total = R.n_pixels * R.iter_count;
    hipLaunchKernelGGL(tail_kernel, dim3((total + 255u) / 256u), dim3(256), 0, stream, R);
    return hipGetLastError();
}

}  // namespace vpt
