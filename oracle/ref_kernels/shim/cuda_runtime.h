/* Build shim (test infrastructure, oracle/_ref only): lets hipcc compile the reference's CUDA kernel file
 * (lib/ransac_voting_gpu_layer/src/ransac_voting_kernel.cu, read in place from the reference tree at build time)
 * for gfx950.  Maps the four CUDA runtime names that file and cuda_common.h use onto HIP. */
#ifndef PVNET_REF_SHIM_CUDA_RUNTIME_H
#define PVNET_REF_SHIM_CUDA_RUNTIME_H
#include <hip/hip_runtime.h>
#include <assert.h>
#include <math.h>
#include <stdlib.h>
typedef hipError_t cudaError_t;
#define cudaSuccess hipSuccess
#define cudaGetErrorString hipGetErrorString
#define cudaGetLastError hipGetLastError
/* ... and the five that lib/utils/extend_utils/src/nearest_neighborhood.cu's host launcher adds */
#define cudaMalloc hipMalloc
#define cudaFree hipFree
#define cudaMemcpy hipMemcpy
#define cudaMemcpyHostToDevice hipMemcpyHostToDevice
#define cudaMemcpyDeviceToHost hipMemcpyDeviceToHost
#endif
