// vpt_resolve.hip -- last stage of the hot path:
//   resolve_kernel (per pixel, in iteration order): NaN/Inf guard (:2263-2264), viz_dof
//   tint (:2266-2274), running-mean accumulation (:2278-2287) and -- once per batch --
//   ACES tonemap + gamma + 8-bit pack + raw buffer (:2292-2316).
// Plus the blue-noise table kernel (golden-ratio advance, :2320-2325).
//
// One thread = one pixel; records are 64-byte lines, so a wave reads 4 KiB contiguous
// per iteration.  Everything here is value-only arithmetic (nothing branches on it that
// feeds the random walk), so the fast device libm is used.
#include <hip/hip_runtime.h>

#include "vpt_device.h"

namespace vpt {

VPT_D f3 rtt_and_odt_fit(f3 v) {                                                       // :2208
    f3 a = v * (v + 0.0245786f) - 0.000090537f;
    f3 b = v * (0.983729f * v + 0.4329510f) + 0.238081f;
    return a / b;
}

// stage 3: per pixel, in iteration order: NaN guard, viz_dof, running means, tonemap
__global__ __launch_bounds__(256) void resolve_kernel(const ResolveParams R) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= R.n_pixels) return;
    f3 acc = mk3(R.accum[3 * idx], R.accum[3 * idx + 1], R.accum[3 * idx + 2]);
    f3 cst = R.cost ? mk3(R.cost[3 * idx], R.cost[3 * idx + 1], R.cost[3 * idx + 2]) : mk3(0.0f);
    float dep = R.depth ? R.depth[idx] : 0.0f;
    float tr_last = 0.0f;

    for (uint32_t k = 0; k < R.iter_count; ++k) {
        const uint32_t iteration = R.iter_begin + k * R.iter_stride;
        const uint32_t local_it = iteration / R.iter_stride;
        const float4* src = reinterpret_cast<const float4*>(R.records + ((size_t)k * R.n_pixels + idx));
        const float4 q0 = src[0];
        f3 value = mk3(q0.x, q0.y, q0.z);
        float tr = q0.w;
        float depth = src[1].w;
        // :2263-2264
        if (isnan(value.x) || isnan(value.y) || isnan(value.z) || isinf(value.x) || isinf(value.y) || isinf(value.z)) value = acc;
        if (isnan(tr) || isinf(tr)) tr = 1.0f;
        // :2266-2274
        if (R.viz_dof) {
            float aof = clampf(1 / R.lens_radius, .0f, 3.402823466e+38F);
            if (depth > (R.focus_dist + aof)) value = lerp3(value, mk3(1, 0, 0), 0.5f);
            if (depth < (R.focus_dist - aof)) value = lerp3(value, mk3(0, 0, 1), 0.5f);
            if (depth > (R.focus_dist - aof) && depth < (R.focus_dist + aof)) value = lerp3(value, mk3(0, 1, 0), 0.5f);
        }
        // :2278-2287 (cost is always BLACK, :2249)
        if (local_it == 0) {
            acc = value;
            cst = mk3(0.0f);
            dep = depth;
        } else if (iteration < R.max_interactions) {
            const float n = (float)(local_it + 1);
            acc = acc + (value - acc) / n;
            cst = cst + (mk3(0.0f) - cst) / n;
            dep = dep + (depth - dep) / n;
        }
        tr_last = tr;
    }
    R.accum[3 * idx] = acc.x; R.accum[3 * idx + 1] = acc.y; R.accum[3 * idx + 2] = acc.z;
    if (R.cost) { R.cost[3 * idx] = cst.x; R.cost[3 * idx + 1] = cst.y; R.cost[3 * idx + 2] = cst.z; }
    if (R.depth) R.depth[idx] = dep;

    if (R.display || R.raw) {
        // :2292-2316
        f3 val = mk3(0.59719f * acc.x + 0.35458f * acc.y + 0.04823f * acc.z,
                     0.07600f * acc.x + 0.90834f * acc.y + 0.01566f * acc.z,
                     0.02840f * acc.x + 0.13383f * acc.y + 0.83777f * acc.z);
        val = rtt_and_odt_fit(val);
        val = mk3(1.60475f * val.x + -0.53108f * val.y + -0.07367f * val.z,
                  -0.10208f * val.x + 1.10813f * val.y + -0.00605f * val.z,
                  -0.00327f * val.x + -0.07276f * val.y + 1.07602f * val.z) * R.exposure_scale;
        const float ig = (float)(1.0 / 2.2);
        const unsigned int r = (unsigned int)(255.0f * fmin_(powf(fmax_(val.x, 0.0f), ig), 1.0f));
        const unsigned int g = (unsigned int)(255.0f * fmin_(powf(fmax_(val.y, 0.0f), ig), 1.0f));
        const unsigned int b = (unsigned int)(255.0f * fmin_(powf(fmax_(val.z, 0.0f), ig), 1.0f));
        if (R.display) R.display[idx] = 0xff000000u | (r << 16) | (g << 8) | b;
        if (R.raw) reinterpret_cast<float4*>(R.raw)[idx] = make_float4(val.x, val.y, val.z, tr_last);
    }
}

// Blue-noise tables: entry i of iteration k = the caller's buffer advanced k*stride golden-ratio
// steps (render_kernel.cu:2320-2325, applied between launches; the reference's in-launch
// read/write race is resolved as "a launch reads the pre-update values").  Also leaves
// the caller's buffer advanced by count*stride steps, as `count` launches would.
__global__ __launch_bounds__(256) void blue_noise_kernel(float* bn /* float3[65536] */, float2* table, uint32_t count, uint32_t stride) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 65536u) return;
    float x = bn[3 * i], y = bn[3 * i + 1], z = bn[3 * i + 2];
    const float phi = (1.0f + sqrtf(5.0f)) / 2.0f;
    for (uint32_t k = 0; k < count; ++k) {
        if (table) table[(size_t)k * 65536u + i] = make_float2(x, y);
        for (uint32_t s = 0; s < stride; ++s) {
            x = fmodf(x + phi, 1.0f);
            y = fmodf(y + phi, 1.0f);
            z = fmodf(z + phi, 1.0f);
        }
    }
    bn[3 * i] = x; bn[3 * i + 1] = y; bn[3 * i + 2] = z;
}

hipError_t launch_resolve(const ResolveParams& R, hipStream_t stream) {
    const int blocks = (int)((R.n_pixels + 255u) / 256u);
    hipLaunchKernelGGL(resolve_kernel, dim3(blocks), dim3(256), 0, stream, R);
    return hipGetLastError();
}
hipError_t launch_blue_noise(float* bn, float2* table, uint32_t count, uint32_t stride, hipStream_t stream) {
    hipLaunchKernelGGL(blue_noise_kernel, dim3(256), dim3(256), 0, stream, bn, table, count, stride);
    return hipGetLastError();
}

}  // namespace vpt
