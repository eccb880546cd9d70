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
    return 0;
}
