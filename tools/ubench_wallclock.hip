// Probe: wall_clock64() (s_memrealtime) against hipEvents on gfx950 -- rate, and whether per-workgroup min/max stamps
// reproduce a kernel's duration.   hipcc --offload-arch=gfx950 -O3 tools/ubench_wallclock.hip -o tools/ubench_wallclock.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void spin(unsigned long long* ts, long long cycles) {
    if (threadIdx.x == 0) atomicMin(&ts[0], (unsigned long long)wall_clock64());
    const long long t0 = clock64();
    while (clock64() - t0 < cycles) {}
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(&ts[1], (unsigned long long)wall_clock64());
}
int main() {
    int khz = 0, ckhz = 0;
    hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
    hipDeviceGetAttribute(&ckhz, hipDeviceAttributeClockRate, 0);
    printf("hipDeviceAttributeWallClockRate = %d kHz, ClockRate = %d kHz\n", khz, ckhz);
    unsigned long long* ts;
    hipMalloc(&ts, 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (long long cyc : {20000ll, 100000ll, 200000ll}) {
        for (int blocks : {256, 2048}) {
            unsigned long long init[2] = {~0ull, 0ull};
            hipMemcpy(ts, init, 16, hipMemcpyHostToDevice);
            hipEventRecord(e0);
            hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, 0, ts, cyc);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            unsigned long long out[2];
            hipMemcpy(out, ts, 16, hipMemcpyDeviceToHost);
            printf("spin %lld shader cycles x %d blocks: events %.2f us, wall_clock ticks %llu -> %.2f us at the reported rate\n",
                   cyc, blocks, ms * 1e3, out[1] - out[0], (double)(out[1] - out[0]) / khz * 1e3);
        }
    }
    return 0;
}
