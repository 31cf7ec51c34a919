// probe: what does v_mad_u64_u32 (Philox's 32 x 32 -> 64 product in one instruction) cost next to v_mul_lo_u32 + v_mul_hi_u32 on gfx950?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mad64_probe tools/probes/mad64_probe.hip && /tmp/mad64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
constexpr int REPS = 4000;
template <int KIND>
__global__ __launch_bounds__(256) void k(unsigned* sink, unsigned m) {
    unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    unsigned long long q0 = a0, q1 = a1, q2 = a2, q3 = a3;
#pragma unroll 1
    for (int r = 0; r < REPS; ++r) {
        if (KIND == 0) {          // 8 x v_mul_lo_u32
            asm volatile(".rept 2\n v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4\n .endr\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));
        } else if (KIND == 1) {   // 8 x v_mad_u64_u32 (64-bit product, as the compiler emits for Philox)
            asm volatile(".rept 2\n v_mad_u64_u32 %0, vcc, %4, %5, 0\n v_mad_u64_u32 %1, vcc, %4, %6, 0\n v_mad_u64_u32 %2, vcc, %4, %7, 0\n v_mad_u64_u32 %3, vcc, %4, %8, 0\n .endr\n" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(m), "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "vcc");
        } else {                  // 4 x (v_mul_hi_u32 + v_mul_lo_u32): the same 64-bit products from two instructions
            unsigned h0, h1, h2, h3;
            asm volatile("v_mul_hi_u32 %4, %0, %8\n v_mul_lo_u32 %0, %0, %8\n v_mul_hi_u32 %5, %1, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_hi_u32 %6, %2, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_hi_u32 %7, %3, %8\n v_mul_lo_u32 %3, %3, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&v"(h0), "=&v"(h1), "=&v"(h2), "=&v"(h3) : "v"(m));
            a0 ^= h0; a1 ^= h1; a2 ^= h2; a3 ^= h3;
        }
    }
    sink[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + (unsigned)(q0 + q1 + q2 + q3) + (unsigned)((q0 ^ q1 ^ q2 ^ q3) >> 32);
}
int main() {
    unsigned* sink; CK(hipMalloc(&sink, 4096 * 256 * 4));
    const char* names[3] = {"8 x v_mul_lo_u32", "8 x v_mad_u64_u32", "4 x (v_mul_hi_u32 + v_mul_lo_u32) + 4 x v_xor_b32"};
    for (int blocks : {1024, 2048}) for (int kind = 0; kind < 3; ++kind) {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int w = 0; w < 2; ++w) {
            CK(hipEventRecord(e0));
            if (kind == 0) k<0><<<blocks, 256>>>(sink, 0xD2511F53u); else if (kind == 1) k<1><<<blocks, 256>>>(sink, 0xD2511F53u); else k<2><<<blocks, 256>>>(sink, 0xD2511F53u);
            CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double waves_per_simd = blocks * 4.0 / 1024.0;
        printf("%d blocks: %-50s %.3f ms -> %.2f cycles at 2.4 GHz per loop body per wave (8 multiplies%s)\n", blocks, names[kind], ms, ms * 1e-3 * 2.4e9 / (REPS * waves_per_simd), kind == 2 ? " + 4 xor" : "");
    }
    return 0;
}
