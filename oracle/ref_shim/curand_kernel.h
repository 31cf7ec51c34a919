// TEST INFRASTRUCTURE -- the part of cuRAND's device API the reference path uses (curand_init / curand_uniform on
// curandStatePhilox4_32_10_t), stated from cuRAND's documented stream semantics: the state is a 128-bit counter, a
// 64-bit key (= seed), the current block of four outputs and a cursor; `offset` skips single 32-bit outputs,
// `subsequence` adds to the upper 64 bits of the counter.  The block function is the oracle's
// (orc_philox4x32_10, pinned against the Random123 known-answer vectors).
#ifndef REF_SHIM_CURAND_KERNEL_H_
#define REF_SHIM_CURAND_KERNEL_H_
#include "cuda_runtime.h"

extern "C" void orc_philox4x32_10(const unsigned int ctr[4], const unsigned int key[2], unsigned int out[4]);

struct curandStatePhilox4_32_10 {
    unsigned int ctr[4];
    unsigned int output[4];
    unsigned int key[2];
    unsigned int STATE;
};
typedef struct curandStatePhilox4_32_10 curandStatePhilox4_32_10_t;

static inline void ref_shim_philox_add(curandStatePhilox4_32_10_t* s, unsigned long long n, int word) {
    for (int i = word; i < 4 && n; i += 1) {
        const unsigned long long sum = (unsigned long long)s->ctr[i] + (n & 0xffffffffull);
        s->ctr[i] = (unsigned int)sum;
        n = (n >> 32) + (sum >> 32);
    }
}
static inline void curand_init(unsigned long long seed, unsigned long long subsequence, unsigned long long offset,
                               curandStatePhilox4_32_10_t* s) {
    s->ctr[0] = s->ctr[1] = s->ctr[2] = s->ctr[3] = 0u;
    s->key[0] = (unsigned int)seed;
    s->key[1] = (unsigned int)(seed >> 32);
    s->STATE = 0u;
    ref_shim_philox_add(s, subsequence, 2);
    s->STATE += (unsigned int)(offset & 3ull);
    ref_shim_philox_add(s, offset / 4ull, 0);
    orc_philox4x32_10(s->ctr, s->key, s->output);
}
static inline unsigned int curand(curandStatePhilox4_32_10_t* s) {
    const unsigned int r = s->output[s->STATE++];
    if (s->STATE == 4u) {
        ref_shim_philox_add(s, 1ull, 0);
        orc_philox4x32_10(s->ctr, s->key, s->output);
        s->STATE = 0u;
    }
    return r;
}
static inline float curand_uniform(curandStatePhilox4_32_10_t* s) {
    return curand(s) * 2.3283064e-10f + (2.3283064e-10f / 2.0f);
}
#endif
