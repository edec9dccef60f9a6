// oracle/ref_fft/ref_fft_host.cpp -- TEST INFRASTRUCTURE ONLY (never shipped, never on the product path).
// Host driver that sweeps the reference's Fourier-reconstruction kernels
//   /root/reference/tomobar/cuda_kernels/fft_us_kernels.cu      (included where it lies, not copied)
// over the launch grids that tomobar/methodsDIR_CuPy.py:645-836,851-967 passes to them, so that the reference's
// FOURIER_INV Python driver can be executed in this container (tests/golden/make_fourier_golden.py) to emit fixtures.
// Same execution-model vocabulary as oracle/ref_tv (cuda_host_exec.h) plus float2 / atomicAdd / the fast-math
// intrinsics the file uses (mapped to the libm functions: the fixtures are compared at 1e-5, far above their error).
#include <algorithm>
#include <cmath>
#include <cstring>

struct float2 { float x, y; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float atomicAdd(float *p, float v) { float o = *p; *p = o + v; return o; }
static inline float ref_fast_expf(float x) { return expf(x); }
static inline void ref_fast_sincosf(float a, float *s, float *c) { *s = sinf(a); *c = cosf(a); }
#define __expf ref_fast_expf        // glibc declares its own __expf / __sincosf
#define __sincosf ref_fast_sincosf
using std::max;
using std::min;

namespace ref_fft {
#include "fft_us_kernels.cu"
}

template <typename F>
static void sweep(const int *grid, const int *block, F &&kernel)
{
    blockDim = {(unsigned)block[0], (unsigned)block[1], (unsigned)block[2]};
    for (unsigned bz = 0; bz < (unsigned)grid[2]; ++bz)
        for (unsigned by = 0; by < (unsigned)grid[1]; ++by)
            for (unsigned bx = 0; bx < (unsigned)grid[0]; ++bx)
                for (unsigned tz = 0; tz < blockDim.z; ++tz)
                    for (unsigned ty = 0; ty < blockDim.y; ++ty)
                        for (unsigned tx = 0; tx < blockDim.x; ++tx) {
                            blockIdx = {bx, by, bz};
                            threadIdx = {tx, ty, tz};
                            kernel();
                        }
}

extern "C" {
void ref_r2c_c1dfftshift(const int *g, const int *b, float *in, float *data, int n, int nproj, int nz)
{ sweep(g, b, [&] { ref_fft::r2c_c1dfftshift(in, (float2 *)data, n, nproj, nz); }); }
void ref_c1dfftshift(const int *g, const int *b, float *data, float constant, int n, int nproj, int nz)
{ sweep(g, b, [&] { ref_fft::c1dfftshift((float2 *)data, constant, n, nproj, nz); }); }
void ref_c2dfftshift(const int *g, const int *b, float *f, int n, int nz)
{ sweep(g, b, [&] { ref_fft::c2dfftshift((float2 *)f, n, nz); }); }
void ref_gather_kernel(const int *g, const int *b, float *gd, float *f, float *theta, int m, float mu, int n, int nproj, int nz)
{ sweep(g, b, [&] { ref_fft::gather_kernel((float2 *)gd, (float2 *)f, theta, m, mu, n, nproj, nz); }); }
void ref_gather_kernel_partial(const int *g, const int *b, float *gd, float *f, float *theta, int m, float mu,
                               int center_size, int n, int nproj, int nz)
{ sweep(g, b, [&] { ref_fft::gather_kernel_partial((float2 *)gd, (float2 *)f, theta, m, mu, center_size, n, nproj, nz); }); }
void ref_gather_kernel_center_angle_based_prune(const int *g, const int *b, unsigned short *angle_range, int dimx,
                                                float *theta, int m, int center_size, int n, int nproj)
{ sweep(g, b, [&] { ref_fft::gather_kernel_center_angle_based_prune(angle_range, dimx, theta, m, center_size, n, nproj); }); }
void ref_gather_kernel_center(const int *g, const int *b, float *gd, float *f, unsigned short *angle_range, int dimx,
                              float *theta, long long *sorted_idx, int m, float mu, int center_size, int n, int nproj, int nz)
{ sweep(g, b, [&] { ref_fft::gather_kernel_center((float2 *)gd, (float2 *)f, angle_range, dimx, theta, sorted_idx, m, mu,
                                                  center_size, n, nproj, nz); }); }
void ref_unpadding_mul_phi(const int *g, const int *b, float *recon_up, float *f, float mu, int nproj, int unpad_recon_p,
                           int unpad_z, int unpad_recon_m, int n, int nz)
{ sweep(g, b, [&] { ref_fft::unpadding_mul_phi(recon_up, (float2 *)f, mu, nproj, unpad_recon_p, unpad_z, unpad_recon_m, n, nz); }); }
}
