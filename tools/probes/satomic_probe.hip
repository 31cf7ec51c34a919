// probe: does gfx950 execute scalar-memory atomics (s_atomic_add ... glc), and what does a returning one cost next to a vector atomicAdd
// under a gather load?   hipcc --offload-arch=gfx950 -O3 -o /tmp/satomic_probe tools/probes/satomic_probe.hip && /tmp/satomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ inline unsigned scalar_fetch_add(unsigned* p, unsigned v) {
    unsigned r = v;
    asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(r) : "s"(p) : "memory");
    return r;
}

template <bool SCALAR>
__global__ __launch_bounds__(256, 4) void k(unsigned* counter, unsigned* seen, const float4* table, unsigned mask, float* sink, unsigned long long* cycles, int claims, int gathers) {
    const int lane = threadIdx.x & 63;
    unsigned h = blockIdx.x * 256 + threadIdx.x;
    float acc = 0.0f;
    unsigned long long t = 0;
    for (int c = 0; c < claims; ++c) {
        for (int g = 0; g < gathers; ++g) {      // the tracer's background: scattered 16-byte loads
            h = h * 1664525u + 1013904223u;
            const float4 v = table[(h >> 4) & mask];
            acc += v.x + v.w;
        }
        const unsigned long long t0 = __builtin_readcyclecounter();
        unsigned base;
        if (SCALAR) {
            base = scalar_fetch_add(counter, 1u);
        } else {
            unsigned b = 0;
            if (lane == 0) b = atomicAdd(counter, 1u);
            base = (unsigned)__builtin_amdgcn_readfirstlane((int)b);
        }
        t += __builtin_readcyclecounter() - t0;
        if (lane == 0) seen[base] += 1u;
    }
    sink[blockIdx.x * 256 + threadIdx.x] = acc;
    if (lane == 0) atomicAdd(cycles, t);
}

int main() {
    const int blocks = 1024, claims = 40;
    const unsigned total = blocks * 4 * claims;
    unsigned *counter, *seen; float4* table; float* sink; unsigned long long* cycles;
    const unsigned mask = (1u << 24) - 1;      // 256 MB of float4
    CK(hipMalloc(&counter, 256)); CK(hipMalloc(&seen, total * 4)); CK(hipMalloc(&table, (size_t)(mask + 1) * 16)); CK(hipMalloc(&sink, blocks * 256 * 4)); CK(hipMalloc(&cycles, 8));
    CK(hipMemset(table, 0, (size_t)(mask + 1) * 16));
    for (int gathers : {0, 8, 32}) for (int scalar = 0; scalar < 2; ++scalar) {
        CK(hipMemset(counter, 0, 256)); CK(hipMemset(seen, 0, total * 4)); CK(hipMemset(cycles, 0, 8));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        if (scalar) k<true><<<blocks, 256>>>(counter, seen, table, mask, sink, cycles, claims, gathers);
        else k<false><<<blocks, 256>>>(counter, seen, table, mask, sink, cycles, claims, gathers);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned> h(total); unsigned cnt; unsigned long long cy;
        CK(hipMemcpy(h.data(), seen, total * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&cnt, counter, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&cy, cycles, 8, hipMemcpyDeviceToHost));
        unsigned bad = 0; for (unsigned v : h) bad += v != 1u;
        printf("%s atomic, %2d gathers between claims: %.3f ms, counter %u of %u, tickets not handed out exactly once %u, cycles per claim %.0f\n", scalar ? "scalar" : "vector", gathers, ms, cnt, total, bad, (double)cy / total);
    }
    return 0;
}
