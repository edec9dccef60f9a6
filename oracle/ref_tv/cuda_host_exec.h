// oracle/ref_tv/cuda_host_exec.h -- TEST INFRASTRUCTURE ONLY (never shipped, never on the product path).
//
// Lets the reference's two TV kernel sources
//   /root/reference/tomobar/cuda_kernels/primal_dual_for_total_variation.cu
//   /root/reference/tomobar/cuda_kernels/rudin_osher_fatemi_total_variation.cu
// be compiled, where they lie, as ordinary host C++ so that their arithmetic can be executed in
// this container (no CUDA toolkit, no NVIDIA GPU) to cross-check oracle/tomo_oracle.c and to emit
// the fixtures under tests/golden/.  It supplies the CUDA execution-model vocabulary the sources
// use (qualifiers, thread/block indices, __half, __fsqrt_rn) -- nothing of the algorithm.
// The kernels are race-free by construction (ping-pong buffers), so a serial sweep over the launch
// grid reproduces a GPU launch exactly.
#pragma once
#include <cmath>
#include <cstdint>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline

struct ref_uint3 { unsigned x, y, z; };
static thread_local ref_uint3 threadIdx, blockIdx, blockDim;

typedef _Float16 __half;
static inline float __half2float(__half h) { return (float)h; }
static inline __half __float2half(float f) { return (__half)f; }
static inline float __fsqrt_rn(float x) { return sqrtf(x); }
