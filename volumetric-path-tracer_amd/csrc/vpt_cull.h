// vpt_cull.h -- "can a primary ray of this pixel reach the reference sphere?", the conservative test behind the never-traced pixel
// mask (vpt_tail.hip: sky_patch_kernel).  Host + device, plain binary32 (the test hook runs it on the CPU).
#pragma once

#include "vpt_math.h"

namespace vpt {

// org: ray origin, dc: unit direction of the pixel's centre ray, diag: chord between the pixel's two opposite corner directions
// (>= its angular diagonal), sph: centre.xyz + radius.  false = NO ray of the pixel makes sphere::intersect (geometry.h:114-137)
// report a hit, whatever its binary32 rounding does:
//   * that function decides with discr = B^2 - 4AC, two numbers of size 4 D^2 (D = distance to the centre) whose difference carries an
//     absolute error of up to ~60 eps D^2, while the true value is 4 (r^2 - p^2) for a ray passing the centre at distance p: it can
//     report hits out to p^2 <= r^2 + 15 eps D^2.  The radius is therefore inflated to r^2 + 64 eps D^2;
//   * the pixel's rays lie within `diag` of the centre ray: at distance D they pass within D diag of it (taken 1.5 times);
//   * p^2 = D^2 - (oc . dc)^2 is itself formed with cancellation: 16 eps D^2 are conceded;
//   * a sphere entirely behind a ray that starts outside it has two negative roots and is never hit.
// The B == 0 quirk of that function (a "hit" at distance 0 wherever the sphere is) is a separate test (ResolveParams::cull_line).
VPT_HD bool sphere_may_hit(f3 org, f3 dc, float diag, const float* sph) {
    const f3 oc = mk3(sph[0], sph[1], sph[2]) - org;
    const float D2 = dot(oc, oc), tca = dot(oc, dc);
    const float eps = 1.1920929e-7f;
    const float r2 = sph[3] * sph[3] + 64.0f * eps * D2;
    if (!(tca > 0.0f || D2 <= r2)) return false;
    const float reach = sqrtf(r2) + sqrtf(D2) * diag * 1.5f;
    return D2 - tca * tca - 16.0f * eps * D2 <= reach * reach;
}

}  // namespace vpt
