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

// ---- thinning (ransac_voting_gpu.py:537-540): keep a pixel <=> its random word r < threshold ---------------------------------
// The threshold has to come from a table that can be histogrammed before the image's foreground count tn0 is known (the mask
// kernel counts, per segment, how many pixels every possible threshold keeps).  Rounds 2-3 used the top ten bits of r: the
// probability max_num / tn0 rounded UP to a multiple of 1/1024 -- fine at the default max_num = 30 000 (+0.1 %), but +17 .. +58 %
// at the evaluation call site's max_num = 100 of 60 000 pixels.  Round 4: the same 1/1024 steps down to
// p = 1/64 and, below that, sixteen steps per octave of r -- the probability is rounded up by at most 1/1024 absolute AND at
// most 1/16 relative: PVNET_THIN_LAST + 1 = 1424 bins, monotone in r.
//   pvnet_thin_bin(r)  = bin of r;   keep <=> bin(r) < K,   K = pvnet_thin_bins_kept(max_num, tn0)   (<=> r < the first r of bin K)
#define PVNET_THIN_LOG_BINS 416           /* 26 octaves (leading one at bit 0 .. 25) x 16 */
#define PVNET_THIN_LAST (PVNET_THIN_LOG_BINS - 16 + 1023)   /* the highest bin index: 1423 */
PVNET_HD int pvnet_thin_bin(uint32_t r) {
    if (r >> 26) return PVNET_THIN_LOG_BINS - 16 + (int)(r >> 22);     // linear part (r >> 22 = 16 .. 1023): 416 .. 1423
    if (r == 0u) return 0;
#if defined(__HIP_DEVICE_COMPILE__)
    const int e = 31 - __clz((int)r);                                    // position of the leading one, 0 .. 25: branch-free
#else
    int e = 25;
    while (!((r >> e) & 1u)) --e;                                        // the same on the host (oracle builds compile this header with gcc)
#endif
    const uint32_t sub = e >= 4 ? (r >> (e - 4)) & 15u : (r << (4 - e)) & 15u;
    return e * 16 + (int)sub;
}
// bins kept at probability max_num / tn0 (tn0 > max_num >= 0): the smallest K whose bins 0 .. K - 1 cover every r < ceil(2^32 max_num / tn0)
PVNET_HD int pvnet_thin_bins_kept(long long max_num, long long tn0) {
    const unsigned long long t = (unsigned long long)(((unsigned __int128)max_num << 32) + (unsigned long long)tn0 - 1) / (unsigned long long)tn0;
    return t == 0ull ? 0 : pvnet_thin_bin((uint32_t)(t - 1ull)) + 1;
}
