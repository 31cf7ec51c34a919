// vpt_env.hip -- host-side environment importance tables: the step the reference performs before
// a vol_integrator launch can run `estimate_sky` on the procedural sky.
//
//   vpt_env_cdf_build   create_cdf's table fill           (source/main.cpp:647-757)
//                       + the host single-scattering sky  (source/main.cpp:182-312)
//   vpt_env_cdf_create  ... + the five texture objects     (source/main.cpp:759-867)
//
// The reference's fill loop reads one element before its arrays (`*(cdf_p - 1)`, `*(func_p - 1)`
// at the first texel, `*(marginal_cdf_p - 1)` at the first row) and writes one past the end of
// marginal_cdf; those reads are defined as 0 here and the stray writes are dropped.  Everything
// else -- including the row-start `*(cdf_p - 1) = 0` that clears the previous row's last cdf
// entry before it is later forced to 1, the running `func` offset by one texel, and the
// `total_int` loop that adds marginal_func[0] `res` times -- is restated as written.
// Host code only (no kernels); strict arithmetic like the rest of the host side.
//
// NOTE on form: `solve_quadratic`, `ray_sphere`, `host_degree_to_cartesian` and `host_sample_atmosphere` below (lines 25-118) are a LITERAL
// RESTATEMENT of source/main.cpp:182-312, statement for statement with other names -- not a redesign.  They feed point-sampled CDF inversion on
// the decision path of estimate_sky, tests/test_env_cdf.py demands the tables BIT-identical to the reference's own lines compiled for the host,
// and for ~90 lines of scalar binary32 host arithmetic the operation order IS the specification.  The fill loop after them is re-expressed.
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/vpt_abi.h"
#include "vpt_math.h"

namespace vpt {
namespace {

// solveQuadratic, main.cpp:182-199
bool solve_quadratic(float a, float b, float c, float& x1, float& x2) {
    if (b == 0) {
        if (a == 0) return false;
        x1 = 0;
        x2 = sqrtf(-c / a);
        return true;
    }
    float discr = b * b - 4 * a * c;
    if (discr < 0) return false;
    float q = (b < 0.f) ? -0.5f * (b - sqrtf(discr)) : -0.5f * (b + sqrtf(discr));
    x1 = q / a;
    x2 = c / q;
    return true;
}
// raySphereIntersect, main.cpp:201-214
bool ray_sphere(const f3& orig, const f3& dir, float radius, float& t0, float& t1) {
    float A = dir.x * dir.x + dir.y * dir.y + dir.z * dir.z;
    float B = 2 * (dir.x * orig.x + dir.y * orig.y + dir.z * orig.z);
    float C = orig.x * orig.x + orig.y * orig.y + orig.z * orig.z - radius * radius;
    if (!solve_quadratic(A, B, C, t0, t1)) return false;
    if (t0 > t1) {
        float tmp = t1;
        t1 = t0;
        t0 = tmp;
    }
    return true;
}
// host degree_to_cartesian, main.cpp:222-236 (elevation clamped to [0, 90], Q-list 13)
f3 host_degree_to_cartesian(float azimuth, float elevation) {
    float az = clampf(azimuth, .0f, 360.0f);
    float el = clampf(elevation, .0f, 90.0f);
    az = az * VPT_PI / 180.0f;
    el = (90.0f - el) * VPT_PI / 180.0f;
    float x = sinf(el) * cosf(az);
    float y = cosf(el);
    float z = sinf(el) * sinf(az);
    return normalize(mk3(x, y, z));
}
// host sample_atmosphere, main.cpp:242-312: single-scattering sky, 16 view x 8 light samples
f3 host_sample_atmosphere(const vpt_kernel_params& kp, f3 orig, f3 dir, f3 intensity) {
    const float atmosphereRadius = 6420e3f;
    const f3 sunDirection = host_degree_to_cartesian(kp.azimuth, kp.elevation);
    const float earthRadius = 6360e3f;
    const float Hr = 7994.0f, Hm = 1200.0f;
    const f3 betaR = mk3(3.8e-6f, 13.5e-6f, 33.1e-6f);
    const f3 betaM = mk3(21e-6f);
    float t0, t1;
    float tmin, tmax = 3.402823466e+38f;
    f3 pos = orig;
    pos.y += 1000 + 6360e3f;
    if (ray_sphere(pos, dir, earthRadius, t0, t1) && t1 > .0f) tmax = fmax_(.0f, t0);
    tmin = .0f;
    if (!ray_sphere(pos, dir, atmosphereRadius, t0, t1) || t1 < 0) return mk3(1.0f, .0f, .0f);
    if (t0 > tmin && t0 > 0) tmin = t0;
    if (t1 < tmax) tmax = t1;
    const unsigned numSamples = 16, numSamplesLight = 8;
    float segmentLength = (tmax - tmin) / numSamples;
    float tCurrent = tmin;
    f3 sumR = mk3(0.0f), sumM = mk3(0.0f);
    float opticalDepthR = 0, opticalDepthM = 0;
    float mu = dot(dir, sunDirection);
    float phaseR = 3.f / (16.f * VPT_PI) * (1 + mu * mu);
    float g = 0.76f;
    float phaseM = 3.f / (8.f * VPT_PI) * ((1.f - g * g) * (1.f + mu * mu)) / ((2.f + g * g) * std::pow(1.f + g * g - 2.f * g * mu, 1.5f));
    for (unsigned i = 0; i < numSamples; ++i) {
        f3 samplePosition = pos + (tCurrent + segmentLength * 0.5f) * dir;
        float height = length(samplePosition) - earthRadius;
        float hr = std::exp(-height / Hr) * segmentLength;
        float hm = std::exp(-height / Hm) * segmentLength;
        opticalDepthR += hr;
        opticalDepthM += hm;
        float t0Light = 0, t1Light = 0;
        ray_sphere(samplePosition, sunDirection, atmosphereRadius, t0Light, t1Light);
        float segmentLengthLight = t1Light / numSamplesLight, tCurrentLight = 0;
        float opticalDepthLightR = 0, opticalDepthLightM = 0;
        unsigned j;
        for (j = 0; j < numSamplesLight; ++j) {
            f3 samplePositionLight = samplePosition + (tCurrentLight + segmentLengthLight * 0.5f) * sunDirection;
            float heightLight = length(samplePositionLight) - earthRadius;
            if (heightLight < 0) break;
            opticalDepthLightR += std::exp(-heightLight / Hr) * segmentLengthLight;
            opticalDepthLightM += std::exp(-heightLight / Hm) * segmentLengthLight;
            tCurrentLight += segmentLengthLight;
        }
        if (j == numSamplesLight) {
            f3 tau = betaR * (opticalDepthR + opticalDepthLightR) + betaM * 1.1f * (opticalDepthM + opticalDepthLightM);
            f3 attenuation = mk3(std::exp(-tau.x), std::exp(-tau.y), std::exp(-tau.z));
            sumR += attenuation * hr;
            sumM += attenuation * hm;
        }
        tCurrent += segmentLength;
    }
    return (sumR * betaR * phaseR + sumM * betaM * phaseM) * intensity;
}

}  // namespace
}  // namespace vpt

using namespace vpt;

extern "C" {

int vpt_env_cdf_build(const vpt_kernel_params* kp, int res_i, float* val4, float* func, float* cdf, float* marginal_func,
                      float* marginal_cdf, float* marginal_int_out) {
    if (!kp || res_i < 2 || !func || !cdf || !marginal_func || !marginal_cdf) return VPT_E_INVALID;
    const unsigned res = (unsigned)res_i;
    const size_t n2 = (size_t)res * res;
    std::vector<f3> val(n2, mk3(0.0f));
    std::memset(func, 0, sizeof(float) * n2);
    std::memset(cdf, 0, sizeof(float) * n2);
    std::memset(marginal_func, 0, sizeof(float) * res);
    std::memset(marginal_cdf, 0, sizeof(float) * res);
    const f3 pos = mk3(0.0f);
    const f3 sky_color = mk3(kp->sky_color.x, kp->sky_color.y, kp->sky_color.z);
    // main.cpp:684-697
    for (unsigned y = 0; y < res; ++y) {
        const float el = float(y) / float(res - 1) * VPT_PI;
        if (y > 0) cdf[(size_t)y * res - 1] = .0f;                      // *(cdf_p - 1) = .0f
        for (unsigned x = 0; x < res; ++x) {
            const size_t i = (size_t)y * res + x;
            const float az = float(x) / float(res - 1) * VPT_PI * 2.0f;
            const f3 dir = mk3(sinf(el) * cosf(az), cosf(el), sinf(el) * sinf(az));
            val[i] = host_sample_atmosphere(*kp, pos, dir, sky_color);
            func[i] = length(val[i]);
            const float prev_cdf = i > 0 ? cdf[i - 1] : .0f;            // reads before the array are 0
            const float prev_func = i > 0 ? func[i - 1] : .0f;
            cdf[i] = prev_cdf + prev_func / (res);
        }
        marginal_func[y] = cdf[(size_t)y * res + res - 1];
    }
    // main.cpp:705-731
    float total_int = 0.0f;
    for (unsigned j = 0; j < res; j++) total_int += marginal_func[0];     // the pointer is not advanced
    if (total_int == .0f) {
        for (unsigned y = 0; y < res; ++y)
            for (unsigned x = 0; x < res; ++x) cdf[(size_t)y * res + x] = (float(x) / float(res)) * (float(y) / float(res));
    } else {
        for (unsigned y = 0; y < res; y++)
            for (unsigned x = 0; x < res; ++x) {
                float& c = cdf[(size_t)y * res + x];
                c /= marginal_func[y];
                if (x == res - 1) c = 1.0f;
            }
    }
    // main.cpp:733-757
    for (unsigned y = 0; y < res; ++y) marginal_cdf[y] = (y > 0 ? marginal_cdf[y - 1] : .0f) + marginal_func[y] / res;
    const float marginal_int = marginal_cdf[res - 1];
    if (marginal_int > .0f)
        for (unsigned y = 0; y < res; ++y) marginal_cdf[y] /= fmax_(.000001f, marginal_int);
    else
        marginal_cdf[0] = 1.0f;                                         // `*marginal_cdf_p = 1.0f` with the pointer never advanced
    if (marginal_int_out) *marginal_int_out = marginal_int;
    if (val4)
        for (size_t i = 0; i < n2; ++i) {
            val4[4 * i + 0] = val[i].x; val4[4 * i + 1] = val[i].y; val4[4 * i + 2] = val[i].z; val4[4 * i + 3] = 1.0f;
        }
    return VPT_OK;
}

int vpt_env_cdf_create(vpt_ctx* ctx, vpt_kernel_params* kp) {
    if (!ctx || !kp) return VPT_E_INVALID;
    const int res = 180;                                                  // main.cpp:664
    const size_t n2 = (size_t)res * res;
    std::vector<float> val4(4 * n2), func(n2), cdf(n2), mfunc(res), mcdf(res);
    float mint = 0.0f;
    int rc = vpt_env_cdf_build(kp, res, val4.data(), func.data(), cdf.data(), mfunc.data(), mcdf.data(), &mint);
    if (rc != VPT_OK) return rc;
    // sampler states of main.cpp:775-867
    vpt_texture_desc d_val = {res, res, 1, 4, 1, VPT_FILTER_LINEAR, {VPT_ADDR_WRAP, VPT_ADDR_CLAMP, VPT_ADDR_WRAP}};
    vpt_texture_desc d_2d = {res, res, 1, 1, 0, VPT_FILTER_POINT, {VPT_ADDR_WRAP, VPT_ADDR_CLAMP, VPT_ADDR_WRAP}};
    vpt_texture_desc d_1d = {res, 1, 1, 1, 0, VPT_FILTER_POINT, {VPT_ADDR_WRAP, VPT_ADDR_CLAMP, VPT_ADDR_WRAP}};
    vpt_texture_t t[5] = {0, 0, 0, 0, 0};
    if ((rc = vpt_texture_create(ctx, &d_val, val4.data(), &t[0])) != VPT_OK) return rc;
    if ((rc = vpt_texture_create(ctx, &d_2d, func.data(), &t[1])) != VPT_OK) return rc;
    if ((rc = vpt_texture_create(ctx, &d_2d, cdf.data(), &t[2])) != VPT_OK) return rc;
    if ((rc = vpt_texture_create(ctx, &d_1d, mfunc.data(), &t[3])) != VPT_OK) return rc;
    if ((rc = vpt_texture_create(ctx, &d_1d, mcdf.data(), &t[4])) != VPT_OK) return rc;
    kp->env_sample_tex_res = res;
    kp->env_marginal_int = mint;
    kp->sky_tex = t[0];
    kp->env_func_tex = t[1];
    kp->env_cdf_tex = t[2];
    kp->env_marginal_func_tex = t[3];
    kp->env_marginal_cdf_tex = t[4];
    return VPT_OK;
}

}  // extern "C"
