// vpt_tex.h -- generic texture sampler restating the CUDA sampler states the reference creates
// (SURVEY appendix C): normalised / unnormalised coordinates, point / linear filter, wrap / clamp.
#pragma once

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

// lat-long environment look-up, sample_env_tex (render_kernel.cu:897-907) / :1845-1849
VPT_D f3 env_lookup(const DTexture& env, f3 wi) {
    f4 t = tex2d(env, atan2f(wi.z, wi.x) * (float)(0.5 / (double)VPT_PI) + 0.5f,
                 acosf(fmax_(fmin_(wi.y, 1.0f), -1.0f)) * (float)(1.0 / (double)VPT_PI));
    return xyz(t);
}

}  // namespace vpt
