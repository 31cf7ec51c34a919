// vpt_testhooks.hip -- test probes (include/vpt_testhooks.h); not on the render path.
#include <hip/hip_runtime.h>
#include <rocrand/rocrand_philox4x32_10.h>

#include <vector>

#include "../../include/vpt_testhooks.h"
#include "vpt_math.h"
#include "vpt_rng.h"

using namespace vpt;

namespace {
VPT_HD float eval_op(int op, float x) {
    switch (op) {
        case VPT_OP_LOG: return det_logf(x);
        case VPT_OP_SIN: return det_sinf(x);
        case VPT_OP_COS: return det_cosf(x);
        case VPT_OP_UNIFORM: return (float)f2u(x) * 2.3283064365386963e-10f + 1.1641532182693481e-10f;
        case VPT_OP_RCP: return 1.0f / x;
        case VPT_OP_SQRT: return sqrtf(x);
    }
    return 0.0f;
}
__global__ void math_kernel(int op, const float* in, float* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = eval_op(op, in[i]);
}
__global__ void uniform_kernel(unsigned long long seed, unsigned long long offset, int n, float* out) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    rocrand_state_philox4x32_10 s;
    rocrand_init(seed, 0, offset, &s);
    for (int i = 0; i < n; ++i) out[i] = (float)rocrand(&s) * 2.3283064365386963e-10f + 1.1641532182693481e-10f;
}
// the product's own generator (csrc/vpt_rng.h), drawn through its refill-point protocol:
// top_up, then at most two draws, as the trace kernel does
__global__ void product_stream_kernel(unsigned int key, unsigned int offset, int n, float* out) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    Rng g;
    rng_init(g, key, offset);
    unsigned int draws = 0;
    int i = 0;
    while (i < n) {
        rng_top_up(g, key);
        out[i++] = rnd(g, draws);
        if (i < n && ((i * 2654435761u) >> 31)) out[i++] = rnd(g, draws);    // irregular 1-or-2 draw pattern
    }
}
}  // namespace

extern "C" {

int vpt_test_device_product_stream(vpt_ctx* ctx, unsigned int key, unsigned int offset, int n, float* out) {
    if (!ctx || !out || n <= 0) return VPT_E_INVALID;
    float* d_out = nullptr;
    if (hipMalloc(&d_out, sizeof(float) * n) != hipSuccess) return VPT_E_NOMEM;
    hipLaunchKernelGGL(product_stream_kernel, dim3(1), dim3(64), 0, 0, key, offset, n, d_out);
    hipError_t e = hipMemcpy(out, d_out, sizeof(float) * n, hipMemcpyDeviceToHost);
    hipFree(d_out);
    return e == hipSuccess ? VPT_OK : VPT_E_HIP;
}

int vpt_test_host_math(int op, const float* in, float* out, int n) {
    if (!in || !out || n < 0) return VPT_E_INVALID;
    for (int i = 0; i < n; ++i) out[i] = eval_op(op, in[i]);
    return VPT_OK;
}

int vpt_test_device_math(vpt_ctx* ctx, int op, const float* in, float* out, int n) {
    if (!ctx || !in || !out || n <= 0) return VPT_E_INVALID;
    float *d_in = nullptr, *d_out = nullptr;
    if (hipMalloc(&d_in, sizeof(float) * n) != hipSuccess || hipMalloc(&d_out, sizeof(float) * n) != hipSuccess) return VPT_E_NOMEM;
    hipMemcpy(d_in, in, sizeof(float) * n, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(math_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, op, d_in, d_out, n);
    hipError_t e = hipMemcpy(out, d_out, sizeof(float) * n, hipMemcpyDeviceToHost);
    hipFree(d_in);
    hipFree(d_out);
    return e == hipSuccess ? VPT_OK : VPT_E_HIP;
}

int vpt_test_device_uniform_stream(vpt_ctx* ctx, unsigned long long seed, unsigned long long offset, int n, float* out) {
    if (!ctx || !out || n <= 0) return VPT_E_INVALID;
    float* d_out = nullptr;
    if (hipMalloc(&d_out, sizeof(float) * n) != hipSuccess) return VPT_E_NOMEM;
    hipLaunchKernelGGL(uniform_kernel, dim3(1), dim3(64), 0, 0, seed, offset, n, d_out);
    hipError_t e = hipMemcpy(out, d_out, sizeof(float) * n, hipMemcpyDeviceToHost);
    hipFree(d_out);
    return e == hipSuccess ? VPT_OK : VPT_E_HIP;
}

}  // extern "C"
