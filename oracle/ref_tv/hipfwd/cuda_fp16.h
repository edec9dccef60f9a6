// Name forwarder only: the reference's TV sources say `#include <cuda_fp16.h>`; on ROCm the same types and conversions
// (__half, __float2half, __half2float) come from the toolchain's own header.  Nothing is defined here.
#include <hip/hip_fp16.h>
