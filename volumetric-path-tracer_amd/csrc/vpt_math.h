// vpt_math.h -- device/host math of the HIP path tracer (product code).
//
// "Strict arithmetic" contract (DESIGN.md, section Arithmetic): every expression on the
// decision path (ray generation, slab/sphere tests, tracking steps, density look-ups,
// phase sampling) is evaluated as individually rounded IEEE-754 binary32 operations in
// the order the reference source writes them -- this translation unit is compiled with
// -ffp-contract=off, HIP's default correctly-rounded fp32 divide/sqrt, and no
// fast-math.  log/sin/cos on that path are the fixed-sequence routines below (Cephes
// single-precision algorithms), so a sample's random walk is a pure function of
// (pixel, iteration) on any IEEE machine.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define VPT_HD __host__ __device__ __forceinline__
#define VPT_D __device__ __forceinline__

namespace vpt {

struct f3 {
    float x, y, z;
};
struct f4 {
    float x, y, z, w;
};

VPT_HD f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
VPT_HD f3 mk3(float s) { return mk3(s, s, s); }
VPT_HD f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
VPT_HD f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
VPT_HD f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
VPT_HD f3 operator/(f3 a, f3 b) { return mk3(a.x / b.x, a.y / b.y, a.z / b.z); }
VPT_HD f3 operator*(f3 a, float b) { return mk3(a.x * b, a.y * b, a.z * b); }
VPT_HD f3 operator*(float b, f3 a) { return mk3(b * a.x, b * a.y, b * a.z); }
VPT_HD f3 operator/(f3 a, float b) { return mk3(a.x / b, a.y / b, a.z / b); }
VPT_HD f3 operator+(f3 a, float b) { return mk3(a.x + b, a.y + b, a.z + b); }
VPT_HD f3 operator-(f3 a, float b) { return mk3(a.x - b, a.y - b, a.z - b); }
VPT_HD f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
VPT_HD void operator+=(f3& a, f3 b) { a = a + b; }
VPT_HD void operator-=(f3& a, f3 b) { a = a - b; }
VPT_HD void operator*=(f3& a, f3 b) { a = a * b; }
VPT_HD void operator*=(f3& a, float b) { a = a * b; }

VPT_HD f4 mk4(float x, float y, float z, float w) { f4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
VPT_HD f4 operator+(f4 a, f4 b) { return mk4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
VPT_HD f4 operator-(f4 a, f4 b) { return mk4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
VPT_HD f4 operator*(f4 a, float b) { return mk4(a.x * b, a.y * b, a.z * b, a.w * b); }
VPT_HD f3 xyz(f4 a) { return mk3(a.x, a.y, a.z); }

// IEEE minNum / maxNum
VPT_HD float fmin_(float a, float b) { return fminf(a, b); }
VPT_HD float fmax_(float a, float b) { return fmaxf(a, b); }
VPT_HD f3 fmin3(f3 a, f3 b) { return mk3(fmin_(a.x, b.x), fmin_(a.y, b.y), fmin_(a.z, b.z)); }
VPT_HD f3 fmax3(f3 a, f3 b) { return mk3(fmax_(a.x, b.x), fmax_(a.y, b.y), fmax_(a.z, b.z)); }
VPT_HD float clampf(float f, float a, float b) { return fmax_(a, fmin_(f, b)); }   // helper_math.h:1175
VPT_HD f3 clamp3(f3 v, float a, float b) { return mk3(clampf(v.x, a, b), clampf(v.y, a, b), clampf(v.z, a, b)); }
VPT_HD float lerpf(float a, float b, float t) { return a + t * (b - a); }           // helper_math.h:1153
VPT_HD f3 lerp3(f3 a, f3 b, float t) { return a + t * (b - a); }
VPT_HD float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }          // helper_math.h:1274
VPT_HD float length(f3 v) { return sqrtf(dot(v, v)); }
VPT_HD f3 normalize(f3 v) { float inv = 1.0f / sqrtf(dot(v, v)); return v * inv; } // helper_math.h:1336
VPT_HD f3 cross(f3 a, f3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
VPT_HD f3 reflect(f3 i, f3 n) { return i - 2.0f * n * dot(n, i); }                 // helper_math.h:1438
VPT_HD float smoothstep(float a, float b, float x) {
    float y = clampf((x - a) / (b - a), 0.0f, 1.0f);
    return (y * y * (3.0f - (2.0f * y)));
}
// sqrt-free forms of `sqrtf(x) < c`: a correctly rounded square root is monotone, so RN(sqrt(x)) < c  <=>  x < T with
// T the smallest float whose rounded root reaches c (tests/test_abi.py::test_sqrt_free_thresholds re-derives both)
#define VPT_SQ_OF_FLT_EPSILON 0x1p-46f          /* c = FLT_EPSILON (1.192092896e-07F) */
#define VPT_SQ_OF_EPS 0x1.0c6f7ap-20f           /* c = 0.001f */
VPT_HD bool is_black(f3 v) { return dot(v, v) < VPT_SQ_OF_FLT_EPSILON; }            // helper_math.h:1520: length(v) < FLT_EPSILON

#define VPT_PI    3.14159265358979323846f
#define VPT_PI_4  0.785398163397448309616f
#define VPT_M_INF 3.402823466e+38F
#define VPT_EPS   0.001f

// ---- fixed-sequence elementary functions (Cephes 2.8 single precision) ---------------
VPT_HD uint32_t f2u(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __float_as_uint(f);
#else
    union { float f; uint32_t u; } c; c.f = f; return c.u;
#endif
}
VPT_HD float u2f(uint32_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    union { float f; uint32_t u; } c; c.u = u; return c.f;
#endif
}

// natural log for x in [0, +inf): 0 -> -inf.  (NaN / negative inputs never occur on the
// path: the argument is always 1 - curand_uniform in [0, 1).)
VPT_HD float det_logf(float x) {
    if (x == 0.0f) return -__builtin_inff();
    uint32_t u = f2u(x);
    int e = 0;
    if ((u >> 23) == 0) {
        x = x * 16777216.0f;
        u = f2u(x);
        e = -24;
    }
    if ((u >> 23) >= 255u) return x;          // inf / nan pass through
    e += (int)(u >> 23) - 126;
    float m = u2f((u & 0x007fffffu) | 0x3f000000u);
    if (m < 0.707106781186547524f) {
        e = e - 1;
        m = m + m;
        m = m - 1.0f;
    } else {
        m = m - 1.0f;
    }
    float z = m * m;
    float p = 7.0376836292E-2f;
    p = p * m; p = p + -1.1514610310E-1f;
    p = p * m; p = p + 1.1676998740E-1f;
    p = p * m; p = p + -1.2420140846E-1f;
    p = p * m; p = p + 1.4249322787E-1f;
    p = p * m; p = p + -1.6668057665E-1f;
    p = p * m; p = p + 2.0000714765E-1f;
    p = p * m; p = p + -2.4999993993E-1f;
    p = p * m; p = p + 3.3333331174E-1f;
    float y = m * z;
    y = y * p;
    float fe = (float)e;
    float t = -2.12194440e-4f * fe;
    y = y + t;
    t = 0.5f * z;
    y = y - t;
    float r = m + y;
    t = 0.693359375f * fe;
    r = r + t;
    return r;
}

VPT_HD float det_reduce_pio4(float x, int* jout) {
    float fj = x * 1.27323954473516f;
    int j = (int)fj;
    if (j & 1) j = j + 1;
    float y = (float)j;
    float t = y * 0.78515625f;
    float r = x - t;
    t = y * 2.4187564849853515625e-4f;
    r = r - t;
    t = y * 3.77489497744594108e-8f;
    r = r - t;
    *jout = j;
    return r;
}
VPT_HD float det_sin_poly(float x, float z) {
    float p = -1.9515295891E-4f;
    p = p * z; p = p + 8.3321608736E-3f;
    p = p * z; p = p + -1.6666654611E-1f;
    p = p * z;
    p = p * x;
    return p + x;
}
VPT_HD float det_cos_poly(float z) {
    float p = 2.443315711809948E-005f;
    p = p * z; p = p + -1.388731625493765E-003f;
    p = p * z; p = p + 4.166664568298827E-002f;
    p = p * z;
    p = p * z;
    float t = 0.5f * z;
    p = p - t;
    return p + 1.0f;
}
// |x| <= 8192 (the path evaluates angles in [0, 2*pi])
VPT_HD float det_sinf(float x) {
    bool neg = false;
    if (x < 0.0f) { neg = true; x = -x; }
    int j;
    float r = det_reduce_pio4(x, &j);
    j &= 7;
    if (j > 3) { neg = !neg; j -= 4; }
    float z = r * r;
    float y = (j == 1 || j == 2) ? det_cos_poly(z) : det_sin_poly(r, z);
    return neg ? -y : y;
}
VPT_HD float det_cosf(float x) {
    bool neg = false;
    if (x < 0.0f) x = -x;
    int j;
    float r = det_reduce_pio4(x, &j);
    j &= 7;
    if (j > 3) { j -= 4; neg = !neg; }
    if (j > 1) neg = !neg;
    float z = r * r;
    float y = (j == 1 || j == 2) ? det_sin_poly(r, z) : det_cos_poly(z);
    return neg ? -y : y;
}
// sin and cos of the same angle share one range reduction
VPT_HD void det_sincosf(float x, float* s, float* c) {
    bool sneg = false;
    if (x < 0.0f) { sneg = true; x = -x; }
    int j;
    float r = det_reduce_pio4(x, &j);
    j &= 7;
    bool cneg = false;
    if (j > 3) { sneg = !sneg; cneg = !cneg; j -= 4; }
    if (j > 1) cneg = !cneg;
    float z = r * r;
    float ps = det_sin_poly(r, z);
    float pc = det_cos_poly(z);
    bool swap = (j == 1 || j == 2);
    float ys = swap ? pc : ps;
    float yc = swap ? ps : pc;
    *s = sneg ? -ys : ys;
    *c = cneg ? -yc : yc;
}

// ---- mat4 helpers (storage m[col][row], reference matrix_math.h:49-70) ---------------
struct mat4 {
    float m[4][4];
};
VPT_HD mat4 mat4_transpose(const mat4& a) {
    mat4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) r.m[i][j] = a.m[j][i];
    return r;
}
VPT_HD mat4 mat4_abs(const mat4& a) {
    mat4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) r.m[i][j] = fabsf(a.m[i][j]);
    return r;
}
VPT_HD f3 mat4_transform_point(const mat4& a, f3 p) {       // matrix_math.h:77-84,293-297
    f3 r;
    r.x = a.m[0][0] * p.x + a.m[1][0] * p.y + a.m[2][0] * p.z + a.m[3][0] * 1.0f;
    r.y = a.m[0][1] * p.x + a.m[1][1] * p.y + a.m[2][1] * p.z + a.m[3][1] * 1.0f;
    r.z = a.m[0][2] * p.x + a.m[1][2] * p.y + a.m[2][2] * p.z + a.m[3][2] * 1.0f;
    return r;
}
VPT_HD f3 mat4_transform_vector(const mat4& a, f3 p) {      // matrix_math.h:298-302
    f3 r;
    r.x = a.m[0][0] * p.x + a.m[1][0] * p.y + a.m[2][0] * p.z + a.m[3][0] * 0.0f;
    r.y = a.m[0][1] * p.x + a.m[1][1] * p.y + a.m[2][1] * p.z + a.m[3][1] * 0.0f;
    r.z = a.m[0][2] * p.x + a.m[1][2] * p.y + a.m[2][2] * p.z + a.m[3][2] * 0.0f;
    return r;
}

// 4x4 inverse by cofactors, operand order of reference matrix_math.h:214-253 (the
// result must carry the same bits as the per-lookup inverse of render_kernel.cu:987)
inline mat4 mat4_inverse(const mat4& a) {
    const float n11 = a.m[0][0], n12 = a.m[1][0], n13 = a.m[2][0], n14 = a.m[3][0];
    const float n21 = a.m[0][1], n22 = a.m[1][1], n23 = a.m[2][1], n24 = a.m[3][1];
    const float n31 = a.m[0][2], n32 = a.m[1][2], n33 = a.m[2][2], n34 = a.m[3][2];
    const float n41 = a.m[0][3], n42 = a.m[1][3], n43 = a.m[2][3], n44 = a.m[3][3];
    const float t11 = n23 * n34 * n42 - n24 * n33 * n42 + n24 * n32 * n43 - n22 * n34 * n43 - n23 * n32 * n44 + n22 * n33 * n44;
    const float t12 = n14 * n33 * n42 - n13 * n34 * n42 - n14 * n32 * n43 + n12 * n34 * n43 + n13 * n32 * n44 - n12 * n33 * n44;
    const float t13 = n13 * n24 * n42 - n14 * n23 * n42 + n14 * n22 * n43 - n12 * n24 * n43 - n13 * n22 * n44 + n12 * n23 * n44;
    const float t14 = n14 * n23 * n32 - n13 * n24 * n32 - n14 * n22 * n33 + n12 * n24 * n33 + n13 * n22 * n34 - n12 * n23 * n34;
    const float det = n11 * t11 + n21 * t12 + n31 * t13 + n41 * t14;
    const float idet = 1.0f / det;
    mat4 r;
    r.m[0][0] = t11 * idet;
    r.m[0][1] = (n24 * n33 * n41 - n23 * n34 * n41 - n24 * n31 * n43 + n21 * n34 * n43 + n23 * n31 * n44 - n21 * n33 * n44) * idet;
    r.m[0][2] = (n22 * n34 * n41 - n24 * n32 * n41 + n24 * n31 * n42 - n21 * n34 * n42 - n22 * n31 * n44 + n21 * n32 * n44) * idet;
    r.m[0][3] = (n23 * n32 * n41 - n22 * n33 * n41 - n23 * n31 * n42 + n21 * n33 * n42 + n22 * n31 * n43 - n21 * n32 * n43) * idet;
    r.m[1][0] = t12 * idet;
    r.m[1][1] = (n13 * n34 * n41 - n14 * n33 * n41 + n14 * n31 * n43 - n11 * n34 * n43 - n13 * n31 * n44 + n11 * n33 * n44) * idet;
    r.m[1][2] = (n14 * n32 * n41 - n12 * n34 * n41 - n14 * n31 * n42 + n11 * n34 * n42 + n12 * n31 * n44 - n11 * n32 * n44) * idet;
    r.m[1][3] = (n12 * n33 * n41 - n13 * n32 * n41 + n13 * n31 * n42 - n11 * n33 * n42 - n12 * n31 * n43 + n11 * n32 * n43) * idet;
    r.m[2][0] = t13 * idet;
    r.m[2][1] = (n14 * n23 * n41 - n13 * n24 * n41 - n14 * n21 * n43 + n11 * n24 * n43 + n13 * n21 * n44 - n11 * n23 * n44) * idet;
    r.m[2][2] = (n12 * n24 * n41 - n14 * n22 * n41 + n14 * n21 * n42 - n11 * n24 * n42 - n12 * n21 * n44 + n11 * n22 * n44) * idet;
    r.m[2][3] = (n13 * n22 * n41 - n12 * n23 * n41 - n13 * n21 * n42 + n11 * n23 * n42 + n12 * n21 * n43 - n11 * n22 * n43) * idet;
    r.m[3][0] = t14 * idet;
    r.m[3][1] = (n13 * n24 * n31 - n14 * n23 * n31 + n14 * n21 * n33 - n11 * n24 * n33 - n13 * n21 * n34 + n11 * n23 * n34) * idet;
    r.m[3][2] = (n14 * n22 * n31 - n12 * n24 * n31 - n14 * n21 * n32 + n11 * n24 * n32 + n12 * n21 * n34 - n11 * n22 * n34) * idet;
    r.m[3][3] = (n12 * n23 * n31 - n13 * n22 * n31 + n13 * n21 * n32 - n11 * n23 * n32 - n12 * n21 * n33 + n11 * n22 * n33) * idet;
    return r;
}

}  // namespace vpt
