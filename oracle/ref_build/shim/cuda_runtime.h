// TEST INFRASTRUCTURE ONLY (oracle/_ref).  Name shim that lets hipcc compile the reference's
// src/cuda_block_solver.cu *where it lies* (never copied) so its kernels can run on the MI355X and serve as
// a second checker for our HIP path.  Maps the dozen CUDA runtime names the reference's device layer uses
// (src/device_buffer.h, src/macro.h, src/cuda_block_solver.cu) onto HIP.  Nothing in the product includes this.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
// everything the reference's .cu pulls from Thrust (rocThrust here) must be seen BEFORE __device__ is widened below
#include <thrust/device_ptr.h>
#include <thrust/scan.h>
#include <thrust/sort.h>
#include <thrust/gather.h>

#define cudaError_t hipError_t
#define cudaSuccess hipSuccess
#define cudaGetErrorString hipGetErrorString
#define cudaGetLastError hipGetLastError
#define cudaDeviceSynchronize hipDeviceSynchronize
#define cudaMalloc hipMalloc
#define cudaFree hipFree
#define cudaMemcpy hipMemcpy
#define cudaMemset hipMemset
#define cudaMemcpyHostToDevice hipMemcpyHostToDevice
#define cudaMemcpyDeviceToHost hipMemcpyDeviceToHost
#define cudaMemcpyDeviceToDevice hipMemcpyDeviceToDevice

// nvcc lets host code construct objects whose constructors are marked __device__ only (the reference builds its
// RobustKernelFunc<> kernel arguments that way, src/cuda_block_solver.cu:1199,1254); clang does not.  Make every
// `__device__` function of the reference translation unit host+device (device-only builtins inside them are fine:
// clang diagnoses those lazily, and the host never calls them).
#undef __device__
#define __device__ __attribute__((device)) __attribute__((host))
