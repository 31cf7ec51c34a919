// TEST INFRASTRUCTURE -- stand-in for the CUDA toolkit headers, so that the reference's own
// source/render_kernel.cu (and the headers it includes) can be compiled by g++ *where it lies under
// /root/reference* into oracle/_ref/libvptref.so.  Nothing here restates the reference: it only supplies
// what the CUDA toolkit would (qualifiers, vector types, texture fetches, launch indices).  The product
// never includes this directory.
//
//   - texture fetches go to the oracle's sampler restatement (orc_texture_sample, pinned by
//     tests/test_oracle_pins.py against exact cases),
//   - cuRAND's Philox stream is in curand_kernel.h (block function: orc_philox4x32_10, pinned against the
//     Random123 known-answer vectors),
//   - logf/sinf/cosf are routed to the oracle's fixed-sequence versions (REF_SHIM_DET_MATH, default on) so
//     that "reference source compiled here" and "oracle restatement" can be compared bit for bit; every other
//     libm call stays libm in both.
#ifndef REF_SHIM_CUDA_RUNTIME_H_
#define REF_SHIM_CUDA_RUNTIME_H_

// everything standard first: the reference later does `#define rand(state) ...` and we redirect logf & co
#include <math.h>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <cfloat>
#include <cstdint>
#include <string>
#include <vector>
#include <iostream>
#include <algorithm>

#define __CUDACC__ 1          /* helper_math.h: skip its host re-definitions of fminf/fmaxf (glibc has them) */
#define __host__
#define __device__
#define __global__
#define __constant__ static
#define __forceinline__ inline
#define __inline__ inline
#define __align__(n) __attribute__((aligned(n)))
#define __launch_bounds__(...)

struct float2 { float x, y; } __attribute__((aligned(8)));
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; } __attribute__((aligned(16)));
struct int2 { int x, y; } __attribute__((aligned(8)));
struct int3 { int x, y, z; };
struct int4 { int x, y, z, w; } __attribute__((aligned(16)));
struct uint2 { unsigned int x, y; } __attribute__((aligned(8)));
struct uint3 { unsigned int x, y, z; };
struct uint4 { unsigned int x, y, z, w; } __attribute__((aligned(16)));
struct uchar4 { unsigned char x, y, z, w; };
struct dim3 { unsigned int x = 1, y = 1, z = 1; dim3() {} dim3(unsigned a, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };

static inline float2 make_float2(float x, float y) { float2 r; r.x = x; r.y = y; return r; }
static inline float3 make_float3(float x, float y, float z) { float3 r; r.x = x; r.y = y; r.z = z; return r; }
static inline float4 make_float4(float x, float y, float z, float w) { float4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
static inline int2 make_int2(int x, int y) { int2 r; r.x = x; r.y = y; return r; }
static inline int3 make_int3(int x, int y, int z) { int3 r; r.x = x; r.y = y; r.z = z; return r; }
static inline int4 make_int4(int x, int y, int z, int w) { int4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
static inline uint2 make_uint2(unsigned x, unsigned y) { uint2 r; r.x = x; r.y = y; return r; }
static inline uint3 make_uint3(unsigned x, unsigned y, unsigned z) { uint3 r; r.x = x; r.y = y; r.z = z; return r; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { uint4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }

// helper_math.h declares unary minus on NON-CONST lvalue references (`operator-(float3 &a)`); the reference negates
// temporaries, which nvcc's MSVC-compatible front end lets bind to them.  Standard C++ needs these for rvalues
// (lvalues still pick helper_math.h's own overloads: a non-const reference is the better match).
static inline float2 operator-(const float2& a) { return make_float2(-a.x, -a.y); }
static inline float3 operator-(const float3& a) { return make_float3(-a.x, -a.y, -a.z); }
static inline float4 operator-(const float4& a) { return make_float4(-a.x, -a.y, -a.z, -a.w); }

// launch indices: the driver runs one "thread" at a time
extern thread_local uint3 blockIdx, threadIdx;
extern thread_local dim3 blockDim, gridDim;

// CUDA's overloaded min/max (device math API)
static inline int max(int a, int b) { return a > b ? a : b; }
static inline int min(int a, int b) { return a < b ? a : b; }
static inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
static inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
static inline float max(float a, float b) { return fmaxf(a, b); }
static inline float min(float a, float b) { return fminf(a, b); }
static inline double max(double a, double b) { return fmax(a, b); }
static inline double min(double a, double b) { return fmin(a, b); }
static inline double max(float a, double b) { return fmax((double)a, b); }
static inline double max(double a, float b) { return fmax(a, (double)b); }
static inline double min(float a, double b) { return fmin((double)a, b); }
static inline double min(double a, float b) { return fmin(a, (double)b); }
// the host implementation helper_math.h itself gives (helper_math.h:81-84 in the reference tree)
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline float __saturatef(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }

// ---- textures: handles are the oracle's, fetches are the oracle's sampler ----------------------------
typedef unsigned long long cudaTextureObject_t;
typedef unsigned long long cudaSurfaceObject_t;
typedef struct CUmod_st* CUmodule;
typedef struct CUfunc_st* CUfunction;
extern "C" void orc_texture_sample(unsigned long long tex, float u, float v, float w, float out[4]);

template <typename T> static inline T ref_shim_texel(const float o[4]);
template <> inline float ref_shim_texel<float>(const float o[4]) { return o[0]; }
template <> inline float4 ref_shim_texel<float4>(const float o[4]) { return make_float4(o[0], o[1], o[2], o[3]); }
// a null texture object (undefined in CUDA; the reference always binds its tables) reads as 0
static inline void ref_shim_fetch(cudaTextureObject_t t, float x, float y, float z, float o[4]) {
    if (t) orc_texture_sample(t, x, y, z, o);
    else o[0] = o[1] = o[2] = o[3] = 0.0f;
}
template <typename T> static inline T tex1D(cudaTextureObject_t t, float x) { float o[4]; ref_shim_fetch(t, x, 0.0f, 0.0f, o); return ref_shim_texel<T>(o); }
template <typename T> static inline T tex2D(cudaTextureObject_t t, float x, float y) { float o[4]; ref_shim_fetch(t, x, y, 0.0f, o); return ref_shim_texel<T>(o); }
template <typename T> static inline T tex3D(cudaTextureObject_t t, float x, float y, float z) { float o[4]; ref_shim_fetch(t, x, y, z, o); return ref_shim_texel<T>(o); }

// ---- fixed-sequence elementary functions, shared with the oracle -------------------------------------
extern "C" float orc_det_logf(float x);
extern "C" float orc_det_sinf(float x);
extern "C" float orc_det_cosf(float x);
#ifndef REF_SHIM_LIBM
#define logf orc_det_logf
#define sinf orc_det_sinf
#define cosf orc_det_cosf
#endif

#endif
