// pvnet_rng.h -- counter-based, curand-free random bits for the voting layer (host + device).
//
// Replaces the two torch RNG calls of the reference driver: `random_(0, tn)` for the pixel pairs
// (lib/ransac_voting_gpu_layer/ransac_voting_gpu.py:547) and `uniform_(0,1)` for the Bernoulli subsample
// (:538).  A value is a pure function of (seed, tag, stream, counter), so results do not depend on launch
// geometry and the CPU oracle (oracle/ransac_voting_oracle.py, oracle/oracle_c/pvnet_vote_ref.c) restates
// it bit for bit.  Construction: three rounds of a 32-bit xorshift-multiply finaliser keyed by the operands.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define PVNET_HD __host__ __device__ __forceinline__
#else
#define PVNET_HD static inline
#endif

#define PVNET_TAG_HYP 0x48595031u /* pixel-pair draws   : stream = image, counter = (h*vn + k)*2 + j */
#define PVNET_TAG_SUB 0x53554231u /* subsample decisions: stream = image, counter = y*w + x          */

PVNET_HD uint32_t pvnet_mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x21F0AAADu;
    x ^= x >> 15; x *= 0x735A2D97u;
    x ^= x >> 15;
    return x;
}
// per-(seed, tag, stream) key: hoist out of per-element loops
PVNET_HD uint32_t pvnet_rng_key(uint64_t seed, uint32_t tag, uint32_t stream) {
    uint32_t x = pvnet_mix32((uint32_t)seed ^ tag);
    return pvnet_mix32((x ^ (stream * 0x9E3779B1u)) + (uint32_t)(seed >> 32));
}
PVNET_HD uint32_t pvnet_rng_at(uint32_t key, uint32_t counter) { return pvnet_mix32(key ^ (counter * 0x85EBCA77u)); }
PVNET_HD uint32_t pvnet_rng_u32(uint64_t seed, uint32_t tag, uint32_t stream, uint32_t counter) {
    return pvnet_rng_at(pvnet_rng_key(seed, tag, stream), counter);
}
// uniform integer in [0, n) (multiply-shift; n <= 2^31)
PVNET_HD uint32_t pvnet_rng_below(uint32_t r, uint32_t n) { return (uint32_t)(((uint64_t)r * (uint64_t)n) >> 32); }
