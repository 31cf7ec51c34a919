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

#include "vpt_sky.h"

namespace vpt {

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
        const f3 sun_dir = mk3(R.sun_dir[0], R.sun_dir[1], R.sun_dir[2]);
        if (R.integrator != 0) {
            // vol_integrator :1752: L += beta * sample_atmosphere(ray_pos, ray_dir) -- always the
            // procedural sky, no sky_mult / sky_color, whatever environment_type says
            const Sky<ResolveParams> sky = {R};
            value += beta * sky.sample(env_pos, dir, sun_dir);
        } else if (R.environment_type == 0) {                                       // :1838-1842
            if (R.has_atmosphere) {
                const Sky<ResolveParams> sky = {R};
                value += sky.sample(env_pos, dir, sun_dir) * beta * R.sky_mult * sky_color;
            }
        } else {                                                                     // :1843-1850
            value += env_lookup(R.env_tex, dir) * sky_color * beta * (1.0f / (4.0f * VPT_PI));
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
