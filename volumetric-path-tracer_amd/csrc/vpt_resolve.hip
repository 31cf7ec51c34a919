// vpt_resolve.hip -- the blue-noise table kernel (golden-ratio advance, render_kernel.cu:2320-2325).
// (The per-pixel resolve -- NaN guard, running means, tonemap -- is fused with the environment tail
// in vpt_tail.hip.)
#include <hip/hip_runtime.h>

#include "vpt_device.h"

namespace vpt {

// Blue-noise tables: entry i of iteration k = the caller's buffer advanced k*stride golden-ratio
// steps (render_kernel.cu:2320-2325, applied between launches; the reference's in-launch
// read/write race is resolved as "a launch reads the pre-update values").  Also leaves
// the caller's buffer advanced by count*stride steps, as `count` launches would.
// Only the first `live` = min(W*H, 65536) entries advance (`if (idx < 256*256)` runs for idx < W*H only).
__global__ __launch_bounds__(256) void blue_noise_kernel(float* bn /* float3[65536] */, float2* table, uint32_t count, uint32_t stride, uint32_t live) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 65536u) return;
    float x = bn[3 * i], y = bn[3 * i + 1], z = bn[3 * i + 2];
    const float phi = (1.0f + sqrtf(5.0f)) / 2.0f;
    for (uint32_t k = 0; k < count; ++k) {
        // CONTRACT (vpt_abi.h): jitter values lie in [0, 1] (the reference's come from an 8-bit image / 255, then x -> fmod(x + phi, 1)); what
        // raygen and the tail READ is clamped to it -- the never-traced pixel mask and the sky patches are built for that footprint
        if (table) table[(size_t)k * 65536u + i] = make_float2(fminf(fmaxf(x, 0.0f), 1.0f), fminf(fmaxf(y, 0.0f), 1.0f));
        for (uint32_t s = 0; s < stride && i < live; ++s) {
            x = fmodf(x + phi, 1.0f);
            y = fmodf(y + phi, 1.0f);
            z = fmodf(z + phi, 1.0f);
        }
    }
    bn[3 * i] = x; bn[3 * i + 1] = y; bn[3 * i + 2] = z;
}

hipError_t launch_blue_noise(float* bn, float2* table, uint32_t count, uint32_t stride, uint32_t live, hipStream_t stream) {
    hipLaunchKernelGGL(blue_noise_kernel, dim3(256), dim3(256), 0, stream, bn, table, count, stride, live);
    return hipGetLastError();
}

}  // namespace vpt
