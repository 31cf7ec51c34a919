// vpt_rng.h -- the sample stream of the hot path: Philox4x32-10 with cuRAND's stream semantics.
#pragma once

#include "vpt_math.h"

namespace vpt {

// ---- Philox4x32-10 with cuRAND stream semantics ----------------------------------------------------
// (D. E. Shaw Research Random123; cuRAND: key = (seed_lo, seed_hi), counter = offset / 4, four
// outputs per block consumed x,y,z,w -- call sites render_kernel.cu:2235, camera.h:45-46.)
// seed = pixel index < 2^32 and counter = iteration*1024 + draws/4 < 2^32 (checked on the host), so
// only the low key / counter words are ever non-zero.
struct Rng {
    uint32_t c0;            // counter word 0 of the block held in o[]
    uint32_t o0, o1, o2, o3;
    uint32_t idx;           // next word of the block (0..4)
    uint32_t carry;         // one word kept from the previous block
    uint32_t has_carry;
};
VPT_D void philox_block(uint32_t c0, uint32_t key, uint32_t& r0, uint32_t& r1, uint32_t& r2, uint32_t& r3) {
    uint32_t x0 = c0, x1 = 0u, x2 = 0u, x3 = 0u;
    uint32_t k0 = key, k1 = 0u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * x0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * x2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ x1 ^ k0;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ x3 ^ k1;
        x1 = (uint32_t)p1;
        x3 = (uint32_t)p0;
        x0 = n0;
        x2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    r0 = x0; r1 = x1; r2 = x2; r3 = x3;
}
// curand_init(seed = key, subsequence 0, offset): position the stream `offset` draws in
VPT_D void rng_init(Rng& g, uint32_t key, uint32_t offset) {
    g.c0 = offset >> 2;
    g.idx = offset & 3u;
    g.has_carry = 0u;
    g.carry = 0u;
    philox_block(g.c0, key, g.o0, g.o1, g.o2, g.o3);
}
// refill point: afterwards at least 4 draws are buffered
VPT_D void rng_top_up(Rng& g, uint32_t key) {
    if (!g.has_carry && g.idx >= 3u) {
        if (g.idx == 3u) {
            g.carry = g.o3;
            g.has_carry = 1u;
        }
        g.c0 += 1u;
        g.idx = 0u;
        philox_block(g.c0, key, g.o0, g.o1, g.o2, g.o3);
    }
}
VPT_D uint32_t rng_next(Rng& g) {
    uint32_t r;
    if (g.has_carry) {
        r = g.carry;
        g.has_carry = 0u;
    } else {
        r = g.idx == 0u ? g.o0 : (g.idx == 1u ? g.o1 : (g.idx == 2u ? g.o2 : g.o3));
        g.idx += 1u;
    }
    return r;
}
// curand_uniform: x * 2^-32 + 2^-33, (0, 1]   (camera.h:45-46)
VPT_D float rnd(Rng& g, uint32_t& draws) {
    ++draws;
    return (float)rng_next(g) * 2.3283064365386963e-10f + 1.1641532182693481e-10f;
}
// plain cuRAND-style draw with an inline block refill (ray generation, where the number of draws is
// open-ended); leaves has_carry == 0, so {c0, idx, o0..o3} is the whole state
VPT_D float rnd_simple(Rng& g, uint32_t key, uint32_t& draws) {
    if (g.idx == 4u) {
        g.c0 += 1u;
        g.idx = 0u;
        philox_block(g.c0, key, g.o0, g.o1, g.o2, g.o3);
    }
    return rnd(g, draws);
}

}  // namespace vpt
