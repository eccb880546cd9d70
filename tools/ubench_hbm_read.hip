// Micro-benchmark (development aid): how fast can ONE launch stream the 78.6 MB of a batch's int64 masks out of HBM?
// 8-byte vs 16-byte loads per lane, loads in flight per lane, workgroup size.  Two buffers alternate (> 256 MiB apart in
// total) so that the Infinity Cache does not serve the reads.
// hipcc --offload-arch=gfx950 -O3 tools/ubench_hbm_read.hip -o tools/ubench_hbm_read.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int VEC, int ILP>
__global__ void rd(const uint64_t* __restrict__ p, size_t n, unsigned long long* out) {
    // block handles blockDim.x * VEC * ILP consecutive elements; lane-contiguous VEC elements per load
    const size_t base = (size_t)blockIdx.x * blockDim.x * VEC * ILP + (size_t)threadIdx.x * VEC;
    unsigned long long acc = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) {
        const size_t idx = base + (size_t)i * blockDim.x * VEC;
        if (idx + VEC <= n) {
            if (VEC == 1) {
                acc |= __builtin_nontemporal_load(p + idx);
            } else {
                typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
                const u64x2 v = __builtin_nontemporal_load(reinterpret_cast<const u64x2*>(p + idx));
                acc |= v.x | (v.y << 1);
            }
        }
    }
    const unsigned long long m = __ballot(acc != 0);
    if ((threadIdx.x & 63) == 0 && m == 0x123456789ull) out[blockIdx.x] = m;
}

// round 5: what the mask kernel does BESIDES reading -- one 64-bit ballot per 64 pixels stored as a bit word, and (REDUCE) the
// workgroup's foreground count through LDS behind a barrier -- on the same cold buffers: is the barrier what K1 pays over a bare read?
template <int WAVES, bool REDUCE>
__global__ void k1_like(const uint64_t* __restrict__ p, size_t n, unsigned long long* __restrict__ bits, int* __restrict__ seg) {
    constexpr int WPW = 64 / WAVES;   // 64-pixel words per wave: a workgroup = one 4096-pixel segment
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t word0 = ((size_t)blockIdx.x * WAVES + wave) * WPW;
    bool f[WPW];
#pragma unroll
    for (int i = 0; i < WPW; ++i) {
        const size_t idx = (word0 + i) * 64 + lane;
        f[i] = idx < n ? (__builtin_nontemporal_load(p + idx) & 0xFFull) != 0 : false;
    }
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < WPW; ++i) {
        const unsigned long long m = __ballot(f[i]);
        if (lane == 0) bits[word0 + i] = m;
        cnt += __popcll(m);
    }
    if (REDUCE) {
        __shared__ int s_cnt[WAVES];
        if (lane == 0) s_cnt[wave] = cnt;
        __syncthreads();
        int t = 0;
#pragma unroll
        for (int i = 0; i < WAVES; ++i) t += s_cnt[i];
        if (threadIdx.x == 0) seg[blockIdx.x] = t;
    } else if (lane == 0) {
        seg[blockIdx.x * WAVES + wave] = cnt;
    }
}

// STORE = 0: no bit words at all (counts only); 1: the wave's words leave as ONE store, lane i holding word i; 2: as K1, one store per word
template <int WAVES, int STORE>
__global__ void k1_like2(const uint64_t* __restrict__ p, size_t n, unsigned long long* __restrict__ bits, int* __restrict__ seg) {
    constexpr int WPW = 64 / WAVES;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t word0 = ((size_t)blockIdx.x * WAVES + wave) * WPW;
    bool f[WPW];
#pragma unroll
    for (int i = 0; i < WPW; ++i) {
        const size_t idx = (word0 + i) * 64 + lane;
        f[i] = idx < n ? (__builtin_nontemporal_load(p + idx) & 0xFFull) != 0 : false;
    }
    int cnt = 0;
    unsigned long long mine = 0;
#pragma unroll
    for (int i = 0; i < WPW; ++i) {
        const unsigned long long m = __ballot(f[i]);
        if (STORE == 2 && lane == 0) bits[word0 + i] = m;
        if (STORE == 1) mine = lane == i ? m : mine;
        cnt += __popcll(m);
    }
    if (STORE == 1 && lane < WPW) bits[word0 + lane] = mine;
    if (lane == 0) seg[blockIdx.x * WAVES + wave] = cnt;
}

// as k1_like2<WAVES, 2> (one store per word), but load i of the workgroup's waves covers ONE contiguous 512 * WAVES bytes (wave w takes
// words w, w + WAVES, ...) instead of every wave walking its own 4 KB: the access pattern of the bare-read kernel above
template <int WAVES>
__global__ void k1_like3(const uint64_t* __restrict__ p, size_t n, unsigned long long* __restrict__ bits, int* __restrict__ seg) {
    constexpr int WPW = 64 / WAVES;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t seg0 = (size_t)blockIdx.x * 64;
    bool f[WPW];
#pragma unroll
    for (int i = 0; i < WPW; ++i) {
        const size_t idx = (seg0 + wave + i * WAVES) * 64 + lane;
        f[i] = idx < n ? (__builtin_nontemporal_load(p + idx) & 0xFFull) != 0 : false;
    }
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < WPW; ++i) {
        const unsigned long long m = __ballot(f[i]);
        if (lane == 0) bits[seg0 + wave + i * WAVES] = m;
        cnt += __popcll(m);
    }
    if (lane == 0) seg[blockIdx.x * WAVES + wave] = cnt;
}

int main() {
    const size_t n = (size_t)32 * 480 * 640;  // int64 elements of one batch of masks: 78.6 MB
    const int NBUF = 5;                       // 393 MB cycled: beyond the 256 MiB Infinity Cache
    uint64_t* buf[NBUF];
    for (auto& b : buf) { hipMalloc(&b, n * 8); hipMemset(b, 0, n * 8); }
    unsigned long long* out;
    hipMalloc(&out, 1 << 20);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto kernel, int threads, int per_block) {
        const int blocks = (int)((n + per_block - 1) / per_block);
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, buf[w % NBUF], n, out);
        hipDeviceSynchronize();
        float best = 1e9f, sum = 0;
        for (int r = 0; r < 20; ++r) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, buf[r % NBUF], n, out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
            sum += ms;
        }
        printf("%-46s blocks %6d x %4d thr: avg %6.1f us  best %6.1f us  -> %5.2f TB/s (avg)\n", name, blocks, threads,
               sum / 20 * 1e3, best * 1e3, n * 8 / (sum / 20 * 1e-3) / 1e12);
    };
    run("8 B/lane, 4 loads in flight, 1024 thr (as K1)", rd<1, 4>, 1024, 1024 * 4);
    run("8 B/lane, 8 loads in flight, 256 thr", rd<1, 8>, 256, 256 * 8);
    run("8 B/lane, 16 loads in flight, 256 thr", rd<1, 16>, 256, 256 * 16);
    run("16 B/lane, 2 loads in flight, 1024 thr", rd<2, 2>, 1024, 1024 * 4);
    run("16 B/lane, 4 loads in flight, 256 thr", rd<2, 4>, 256, 256 * 8);
    run("16 B/lane, 8 loads in flight, 256 thr", rd<2, 8>, 256, 256 * 16);
    run("16 B/lane, 4 loads in flight, 512 thr", rd<2, 4>, 512, 512 * 8);
    run("16 B/lane, 16 loads in flight, 256 thr", rd<2, 16>, 256, 256 * 32);
    run("8 B/lane, 8 loads in flight, 512 thr (K1's shape)", rd<1, 8>, 512, 512 * 8);
    unsigned long long* bits;
    int* seg;
    hipMalloc(&bits, n / 64 * 8 + 4096);
    hipMalloc(&seg, 1 << 20);
    auto run2 = [&](const char* name, auto kernel, int threads) {
        const int blocks = (int)((n + 4095) / 4096);
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, buf[w % NBUF], n, bits, seg);
        hipDeviceSynchronize();
        float best = 1e9f, sum = 0;
        for (int r = 0; r < 20; ++r) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, buf[r % NBUF], n, bits, seg);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
            sum += ms;
        }
        printf("%-46s blocks %6d x %4d thr: avg %6.1f us  best %6.1f us  -> %5.2f TB/s (avg)\n", name, blocks, threads,
               sum / 20 * 1e3, best * 1e3, n * 8 / (sum / 20 * 1e-3) / 1e12);
    };
    run2("K1-like: ballots + bit words + LDS count, 8 waves", k1_like<8, true>, 512);
    run2("K1-like: ballots + bit words, per-wave count, 8 w", k1_like<8, false>, 512);
    run2("K1-like: ballots + bit words + LDS count, 16 waves", k1_like<16, true>, 1024);
    run2("K1-like: ballots + bit words, per-wave count, 4 w", k1_like<4, false>, 256);
    run2("K1-like2: counts only, no bit words, 8 waves", k1_like2<8, 0>, 512);
    run2("K1-like2: one 64-byte store of 8 words per wave", k1_like2<8, 1>, 512);
    run2("K1-like2: one store per word (as K1), 8 waves", k1_like2<8, 2>, 512);
    run2("K1-like2: one store of 4 words per wave, 16 w", k1_like2<16, 1>, 1024);
    run2("K1-like3: interleaved words, store per word, 8 w", k1_like3<8>, 512);
    run2("K1-like3: interleaved words, store per word, 16 w", k1_like3<16>, 1024);
    run2("K1-like2: one store per word (as K1), 8 waves", k1_like2<8, 2>, 512);
    run2("K1-like3: interleaved words, store per word, 8 w", k1_like3<8>, 512);
    return 0;
}