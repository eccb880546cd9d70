// Development aid: runs compact_kernel<1> from a code object (argv[1]) on the repro's inputs and counts bad runs.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
struct Params {
    const float* vertex; int64_t vs0, vs1, vs2, vs3, vs4;
    int b, h, w, vn, words, cap, nseg;
    const int32_t* seg; const uint64_t* bits; int32_t* pix; float4* rec;
};
int main(int argc, char** argv) {
    const int b = 3, h = 200, w = 280, vn = 9, npix = h * w, words = (npix + 63) / 64, nseg = (words + 63) / 64, cap = 30008;
    std::vector<uint64_t> bits((size_t)b * words, 0);
    std::vector<int32_t> seg((size_t)b * nseg, 0);
    std::vector<float> field((size_t)b * 2 * vn * npix);
    srand(5);
    for (auto& f : field) f = (float)rand() / RAND_MAX - 0.5f;
    std::vector<std::vector<int>> kept(b);
    for (int bi = 0; bi < b; ++bi) {
        const int cx = 90 + 40 * bi, cy = 100 + 10 * bi, R = 31;
        for (int p = 0; p < npix; ++p) {
            const int y = p / w, x = p % w;
            if ((x - cx) * (x - cx) + (y - cy) * (y - cy) <= R * R) {
                bits[(size_t)bi * words + p / 64] |= 1ull << (p % 64);
                seg[bi * nseg + p / 4096]++;
                kept[bi].push_back(p);
            }
        }
    }
    uint64_t* dbits; int32_t *dseg, *dpix; float* dfield; float4* drec;
    (void)hipMalloc(&dbits, bits.size() * 8); (void)hipMalloc(&dseg, seg.size() * 4); (void)hipMalloc(&dpix, (size_t)b * cap * 4);
    (void)hipMalloc(&dfield, field.size() * 4); (void)hipMalloc(&drec, (size_t)b * vn * cap * 16);
    (void)hipMemcpy(dbits, bits.data(), bits.size() * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(dseg, seg.data(), seg.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dfield, field.data(), field.size() * 4, hipMemcpyHostToDevice);
    Params P{dfield, (int64_t)2 * vn * npix, (int64_t)w, 1, (int64_t)2 * npix, (int64_t)npix, b, h, w, vn, words, cap, nseg, dseg, dbits, dpix, drec};
    hipModule_t mod; hipFunction_t fn;
    if (hipModuleLoad(&mod, argv[1]) != hipSuccess) { printf("load failed\n"); return 1; }
    if (hipModuleGetFunction(&fn, mod, "_Z14compact_kernelILi1EEv6Params") != hipSuccess) { printf("no function\n"); return 1; }
    std::vector<float4> rec((size_t)b * vn * cap);
    int badruns = 0, reps = argc > 2 ? atoi(argv[2]) : 200;
    // round 4: threads per workgroup, z extent of the grid and dynamic LDS of the variant (k2_repro_r4.hip: -DNT / -DDUAL / -DDYNLDS)
    const int threads = argc > 3 ? atoi(argv[3]) : 256, gz = argc > 4 ? atoi(argv[4]) : vn, lds = argc > 5 ? atoi(argv[5]) : 0;
    if (lds > 65536) (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    size_t psize = sizeof(P);
    void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &P, HIP_LAUNCH_PARAM_BUFFER_SIZE, &psize, HIP_LAUNCH_PARAM_END};
    long total_bad = 0;
    for (int rep = 0; rep < reps; ++rep) {
        (void)hipMemset(drec, 0xFF, rec.size() * 16);
        if (hipModuleLaunchKernel(fn, nseg, b, gz, threads, 1, 1, lds, 0, nullptr, cfg) != hipSuccess) { printf("launch failed\n"); return 1; }
        (void)hipMemcpy(rec.data(), drec, rec.size() * 16, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int bi = 0; bi < b; ++bi)
            for (int k = 0; k < vn; ++k)
                for (size_t i = 0; i < kept[bi].size(); ++i) {
                    const int p = kept[bi][i];
                    const float4 r = rec[((size_t)bi * vn + k) * cap + i];
                    if (r.x != (float)(p % w) || r.y != (float)(p / w)) ++bad;
                }
        if (bad) ++badruns;
        total_bad += bad;
    }
    printf("%s: bad runs %d of %d (%ld bad records)\n", argv[1], badruns, reps, total_bad);
    return 0;
}
