// vpt_dome.h -- the value-only arithmetic helpers of the environment code, and the SKY DOME's coordinates and look-up
// (ResolveParams::sky_dome, vpt_device.h): shared by the environment tail (vpt_tail.hip, which builds the dome) and the direct
// tracer (vpt_trace.hip, which resolves a finished path's environment term from it).  Everything here is written with explicit
// intrinsics (v_rcp, v_sqrt, fma), so it evaluates to the same bits whatever the including translation unit's flags.
#pragma once

#include "vpt_device.h"

namespace vpt {

VPT_D float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
VPT_D float fdiv(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
VPT_D float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
VPT_D float ffma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
VPT_D float flerp(float a, float b, float t) { return ffma(t, b - a, a); }
VPT_D f3 flerp3(f3 a, f3 b, float t) { return mk3(flerp(a.x, b.x, t), flerp(a.y, b.y, t), flerp(a.z, b.z, t)); }
VPT_D f3 fscale_add3(f3 a, float s, f3 b) { return mk3(ffma(a.x, s, b.x), ffma(a.y, s, b.y), ffma(a.z, s, b.z)); }
// CORRECTLY ROUNDED binary32 root / quotient whatever the translation unit's flags (vpt_tail.hip is built with the approximate ones): through
// binary64, whose 53 bits make the second rounding innocuous for sqrt and / of binary32 operands (2 x 24 + 2 <= 53).  For the GEOMETRY of a ground hit
// only (vpt_sky.h: sample, SkyRadianceToPoint, GroundFromTable): there one ulp of a root decides which 0.5 m step of the earth's radius a point lands on.
VPT_D float sqrt_rn(float x) { return (float)__builtin_sqrt((double)x); }
VPT_D float div_rn(float a, float b) { return (float)((double)a / (double)b); }
VPT_D float length_rn(f3 v) { return sqrt_rn(dot(v, v)); }
VPT_D f3 normalize_rn(f3 v) { const float inv = div_rn(1.0f, sqrt_rn(dot(v, v))); return v * inv; }      // helper_math.h:1336, as the strict side forms it

// ---- sky dome (ResolveParams::sky_dome) ------------------------------------------------------------------------------------------
// direction <-> dome coordinates: v = dir.y in [-1, 1] (rows), u in [0, 4) the L1 azimuth in the xz plane: t = x / (|x| + |z|),
// u = 1 - t for z >= 0 (x runs +1 -> -1), u = 3 + t for z < 0 (x runs -1 -> +1); periodic, piecewise smooth with its kinks (the axes) on nodes
VPT_D void dome_coords(f3 d, float& fu, float& fv) {
    const float s = fabsf(d.x) + fabsf(d.z);
    const float tt = s > 0.0f ? d.x * frcp(s) : 1.0f;
    const float u = d.z >= 0.0f ? 1.0f - tt : 3.0f + tt;
    fu = u * (float)(SKY_DOME_NU / 4);
    fv = fmin_(fmax_(ffma(d.y, 0.5f, 0.5f), 0.0f), 1.0f) * (float)(SKY_DOME_NV - 1);
}
VPT_D f3 dome_direction(float fu, float fv) {
    const float v = clampf(ffma(fv, 2.0f / (float)(SKY_DOME_NV - 1), -1.0f), -1.0f, 1.0f);
    float u = fu * (4.0f / (float)SKY_DOME_NU);
    u = u >= 4.0f ? u - 4.0f : u;
    const bool front = u <= 2.0f;
    const float tt = front ? 1.0f - u : u - 3.0f;
    const float x = tt, z = (1.0f - fabsf(tt)) * (front ? 1.0f : -1.0f);
    const float rxz = fsqrt(fmax_(1.0f - v * v, 0.0f)) * frcp(fsqrt(x * x + z * z));
    return mk3(x * rxz, v, z * rxz);
}
// which dome (= which variant of the camera-point tables) serves a sample that looks from `pos`: the one whose r equals pos' binary32 distance from the earth's
// centre, if that lies within k steps of the camera origin's and mu_s agrees (Sky::CamVariant, vpt_sky.h, with the tail's own operations); -1: none
VPT_D int dome_variant(const SkyView* view, f3 pos, f3 sun_dir, float earth_bottom) {
    // (the view point's three scalars through the CONSTANT address space: written by sky_view_kernel before this launch, launch-uniform address -> scalar loads;
    // as plain global loads they were three dependent vector loads per batch of finishing paths)
    const __attribute__((address_space(4))) SkyView* v = (const __attribute__((address_space(4))) SkyView*)view;
    const float view_r = v->r, view_mu_s = v->mu_s;
    const int vk = v->k;
    const f3 pe = pos - mk3(0.0f, -earth_bottom, 0.0f);
    const float re = length_rn(pe);
    const float mu_s = dot(pe, sun_dir) * frcp(re);
    const int k = (int)(__float_as_uint(re) - __float_as_uint(view_r)) + vk;
    return (k >= 0 && k <= 2 * vk && fabsf(mu_s - view_mu_s) <= 1e-6f) ? k : -1;
}
// the dome's value along d (ResolveParams::sky_dome); false: the cell is flagged, evaluate in full
VPT_D bool dome_lookup(const float4* __restrict__ dome, f3 d, f3& value) {
    float fu, fv;
    dome_coords(d, fu, fv);
    const float flu = floorf(fu), flv = fminf(floorf(fv), (float)(SKY_DOME_NV - 2));
    const float au = fu - flu, av = fv - flv;
    uint32_t i0 = (uint32_t)flu;
    i0 = i0 >= (uint32_t)SKY_DOME_NU ? i0 - (uint32_t)SKY_DOME_NU : i0;
    const uint32_t i1 = i0 + 1u == (uint32_t)SKY_DOME_NU ? 0u : i0 + 1u;
    const uint32_t r0 = (uint32_t)flv * (uint32_t)SKY_DOME_NU, r1 = r0 + (uint32_t)SKY_DOME_NU;
    const float4 a = dome[r0 + i0], b = dome[r0 + i1], c = dome[r1 + i0], e = dome[r1 + i1];      // (one round trip: all four requested before the flag is looked at)
    if (a.w == 0.0f) return false;
    const f3 lo = flerp3(mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), au), hi = flerp3(mk3(c.x, c.y, c.z), mk3(e.x, e.y, e.z), au);
    value = flerp3(lo, hi, av);
    return true;
}

}  // namespace vpt
