// orc_math.h -- TEST INFRASTRUCTURE (CPU oracle).  Not part of the product: only
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build or call
// anything under oracle/.
//
// Vector / matrix helpers and the "strict arithmetic" the oracle is defined in:
//   * every C operator is one IEEE-754 binary32 operation, evaluated in the order
//     the reference source writes it (build with -ffp-contract=off, no fast-math);
//   * normalize() = v * (1.0f / sqrtf(dot)) (reference helper_math.h:1336-1340 uses
//     rsqrtf, whose host definition is 1.0f/sqrtf(x), helper_math.h:85-88);
//   * logf / sinf / cosf on the decision path are the deterministic Cephes-style
//     single-precision routines below (orc_logf, orc_sinf, orc_cosf), because the
//     reference's `--use_fast_math` intrinsics (source/CMakeLists.txt:133) are not
//     reproducible off an NVIDIA GPU (oracle/_ref is built with the same routines).  They are
//     checked against glibc in tests/test_oracle_math.py.
#ifndef ORC_MATH_H_
#define ORC_MATH_H_

#include <cmath>
#include <cstdint>
#include <cstring>
#include <cfloat>

namespace orc {

// reference source/common/helper_math.h:40-54
static const float kPi     = 3.14159265358979323846f;
static const float kPi4    = 0.785398163397448309616f;
static const float kInf    = 3.402823466e+38F;     // M_INF (FLT_MAX, not infinity)

struct f2 { float x, y; };
struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };

static inline f3 mk3(float x, float y, float z) { f3 r = {x, y, z}; return r; }
static inline f3 mk3(float s) { f3 r = {s, s, s}; return r; }
static inline f4 mk4(float x, float y, float z, float w) { f4 r = {x, y, z, w}; return r; }
static inline f3 xyz(f4 a) { return mk3(a.x, a.y, a.z); }

static inline f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline f3 operator/(f3 a, f3 b) { return mk3(a.x / b.x, a.y / b.y, a.z / b.z); }
static inline f3 operator*(f3 a, float b) { return mk3(a.x * b, a.y * b, a.z * b); }
static inline f3 operator*(float b, f3 a) { return mk3(b * a.x, b * a.y, b * a.z); }
static inline f3 operator/(f3 a, float b) { return mk3(a.x / b, a.y / b, a.z / b); }
static inline f3 operator+(f3 a, float b) { return mk3(a.x + b, a.y + b, a.z + b); }
static inline f3 operator-(f3 a, float b) { return mk3(a.x - b, a.y - b, a.z - b); }
static inline f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
static inline void operator+=(f3& a, f3 b) { a = a + b; }
static inline void operator-=(f3& a, f3 b) { a = a - b; }
static inline void operator*=(f3& a, f3 b) { a = a * b; }
static inline void operator*=(f3& a, float b) { a = a * b; }

static inline f4 operator+(f4 a, f4 b) { return mk4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
static inline f4 operator-(f4 a, f4 b) { return mk4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
static inline f4 operator*(f4 a, float b) { return mk4(a.x * b, a.y * b, a.z * b, a.w * b); }

// device fminf/fmaxf = IEEE-754 minNum/maxNum (a NaN operand is dropped); the host
// stand-ins at helper_math.h:63-71 are not what volume_rt_kernel runs with
static inline float fminf_(float a, float b) { return std::fmin(a, b); }
static inline float fmaxf_(float a, float b) { return std::fmax(a, b); }
static inline f3 fmin3(f3 a, f3 b) { return mk3(fminf_(a.x, b.x), fminf_(a.y, b.y), fminf_(a.z, b.z)); }
static inline f3 fmax3(f3 a, f3 b) { return mk3(fmaxf_(a.x, b.x), fmaxf_(a.y, b.y), fmaxf_(a.z, b.z)); }
// helper_math.h:1175-1178
static inline float clampf(float f, float a, float b) { return fmaxf_(a, fminf_(f, b)); }
static inline int   clampi(int f, int a, int b) { return f < a ? a : (f > b ? b : f); }
static inline f3 clamp3(f3 v, float a, float b) { return mk3(clampf(v.x, a, b), clampf(v.y, a, b), clampf(v.z, a, b)); }
// helper_math.h:1153-1164
static inline float lerpf(float a, float b, float t) { return a + t * (b - a); }
static inline f3 lerp3(f3 a, f3 b, float t) { return a + t * (b - a); }
// helper_math.h:1274-1277,1318-1321,1336-1340
static inline float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline float length(f3 v) { return sqrtf(dot(v, v)); }
static inline f3 normalize(f3 v) { float inv = 1.0f / sqrtf(dot(v, v)); return v * inv; }
// helper_math.h:1447-1450, 1438-1441
static inline f3 cross(f3 a, f3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static inline f3 reflect(f3 i, f3 n) { return i - 2.0f * n * dot(n, i); }
// helper_math.h:1479-1483
static inline float smoothstep(float a, float b, float x) {
    float y = clampf((x - a) / (b - a), 0.0f, 1.0f);
    return (y * y * (3.0f - (2.0f * y)));
}
// helper_math.h:1520-1535
static inline bool is_black(f3 v) { return length(v) < 1.192092896e-07F; }
static inline bool is_nan3(f3 v) { return std::isnan(v.x) || std::isnan(v.y) || std::isnan(v.z); }
static inline bool is_inf3(f3 v) { return std::isinf(v.x) || std::isinf(v.y) || std::isinf(v.z); }

// ---- deterministic single-precision elementary functions ----------------------------
// Published algorithm: S. Moshier, Cephes Math Library 2.8, single precision
// (logf.c, sinf.c).  Each line is one rounded binary32 operation.

static inline uint32_t f2u(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

static inline float orc_logf(float x) {
    if (std::isnan(x)) return x;
    if (x < 0.0f) return std::nanf("");
    if (x == 0.0f) return -INFINITY;
    if (std::isinf(x)) return x;
    uint32_t u = f2u(x);
    int e = 0;
    if ((u >> 23) == 0) {            // subnormal: scale by 2^24 (exact)
        x = x * 16777216.0f;
        u = f2u(x);
        e = -24;
    }
    e += (int)(u >> 23) - 126;       // x = m * 2^e, m in [0.5, 1)
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

// shared range reduction: x >= 0, returns octant j (after the "odd -> +1" map) and
// the reduced argument in [-pi/4, pi/4]
static inline float orc_reduce_pio4(float x, int* jout) {
    float fj = x * 1.27323954473516f;    // 4/pi
    int j = (int)fj;
    if (j & 1) j = j + 1;
    float y = (float)j;
    float t = y * 0.78515625f;            // DP1
    float r = x - t;
    t = y * 2.4187564849853515625e-4f;    // DP2
    r = r - t;
    t = y * 3.77489497744594108e-8f;      // DP3
    r = r - t;
    *jout = j;
    return r;
}
static inline float orc_sin_poly(float x, float z) {
    float p = -1.9515295891E-4f;
    p = p * z; p = p + 8.3321608736E-3f;
    p = p * z; p = p + -1.6666654611E-1f;
    p = p * z;
    p = p * x;
    return p + x;
}
static inline float orc_cos_poly(float z) {
    float p = 2.443315711809948E-005f;
    p = p * z; p = p + -1.388731625493765E-003f;
    p = p * z; p = p + 4.166664568298827E-002f;
    p = p * z;
    p = p * z;
    float t = 0.5f * z;
    p = p - t;
    return p + 1.0f;
}
// valid for |x| <= 8192 (the path only evaluates angles in [0, 2*pi])
static inline float orc_sinf(float x) {
    if (!(std::fabs(x) <= 8192.0f)) return std::sin(x);
    float sign = 1.0f;
    if (x < 0.0f) { sign = -1.0f; x = -x; }
    int j;
    float r = orc_reduce_pio4(x, &j);
    j &= 7;
    if (j > 3) { sign = -sign; j -= 4; }
    float z = r * r;
    float y = (j == 1 || j == 2) ? orc_cos_poly(z) : orc_sin_poly(r, z);
    return sign < 0.0f ? -y : y;
}
static inline float orc_cosf(float x) {
    if (!(std::fabs(x) <= 8192.0f)) return std::cos(x);
    float sign = 1.0f;
    if (x < 0.0f) x = -x;
    int j;
    float r = orc_reduce_pio4(x, &j);
    j &= 7;
    if (j > 3) { j -= 4; sign = -sign; }
    if (j > 1) sign = -sign;
    float z = r * r;
    float y = (j == 1 || j == 2) ? orc_sin_poly(r, z) : orc_cos_poly(z);
    return sign < 0.0f ? -y : y;
}

// ---- mat4: reference source/gpu_vdb/matrix_math.h:49-345, storage m[col][row] -----------
struct mat4 {
    float m[4][4];
};

static inline mat4 mat4_transpose(const mat4& a) {          // matrix_math.h:168-175
    mat4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) r.m[i][j] = a.m[j][i];
    return r;
}

static inline mat4 mat4_abs(const mat4& a) {                // matrix_math.h:270-281
    mat4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) r.m[i][j] = std::fabs(a.m[i][j]);
    return r;
}

// matrix_math.h:214-253 -- cofactor expansion with exactly the reference's term order
static inline mat4 mat4_inverse(const mat4& a) {
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

// matrix_math.h:77-84 (operator*(float4)) via transform_point/transform_vector :293-302
static inline f3 mat4_transform_point(const mat4& a, f3 p) {
    f3 r;
    r.x = a.m[0][0] * p.x + a.m[1][0] * p.y + a.m[2][0] * p.z + a.m[3][0] * 1.0f;
    r.y = a.m[0][1] * p.x + a.m[1][1] * p.y + a.m[2][1] * p.z + a.m[3][1] * 1.0f;
    r.z = a.m[0][2] * p.x + a.m[1][2] * p.y + a.m[2][2] * p.z + a.m[3][2] * 1.0f;
    return r;
}
static inline f3 mat4_transform_vector(const mat4& a, f3 p) {
    f3 r;
    r.x = a.m[0][0] * p.x + a.m[1][0] * p.y + a.m[2][0] * p.z + a.m[3][0] * 0.0f;
    r.y = a.m[0][1] * p.x + a.m[1][1] * p.y + a.m[2][1] * p.z + a.m[3][1] * 0.0f;
    r.z = a.m[0][2] * p.x + a.m[1][2] * p.y + a.m[2][2] * p.z + a.m[3][2] * 0.0f;
    return r;
}

}  // namespace orc
#endif
