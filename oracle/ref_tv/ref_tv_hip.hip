// oracle/ref_tv/ref_tv_hip.hip -- TEST INFRASTRUCTURE ONLY (second, independent execution of the reference TV kernels).
//
// The reference's two CUDA sources are compiled UNMODIFIED for gfx950 by hipcc (they are included from
// $(REF)/tomobar/cuda_kernels, never copied; the only header supplied is the one-line name forwarder
// hipfwd/cuda_fp16.h -> <hip/hip_fp16.h>) and their own `extern "C" __global__` entry points are launched on the
// MI355X with the launch geometry and ping-pong loops of tomobar/regularisersCuPy.py:93-167,235-296: block
// (128,1,1), grid (ceil(dx/128), dy, dz).  Same C interface as ref_tv_host.cpp (host pointers in and out), so
// tests/test_gpu_ref_tv.py can diff a real-GPU run of the reference source against tests/golden/tv_golden.npz (made by
// the host-executed build) and against libtomo_mi355x.so.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>  // at global scope first: the .cu files are included inside namespaces below

#include <cstring>
#include <vector>

namespace ref_pd {
#include "primal_dual_for_total_variation.cu"
}
namespace ref_rof {
#include "rudin_osher_fatemi_total_variation.cu"
}

#define CK(e) do { if ((e) != hipSuccess) return 1; } while (0)

template <typename T> struct Buf {
    T *p = nullptr;
    ~Buf() { if (p) (void)hipFree(p); }
    int alloc(size_t n, bool zero) {
        if (hipMalloc((void **)&p, n * sizeof(T)) != hipSuccess) return 1;
        return zero ? (hipMemset(p, 0, n * sizeof(T)) != hipSuccess) : 0;
    }
};

template <typename T>
static int run_pd(const float *in, float *out, int dx, int dy, int dz, int nd, float sigma, float tau, float lt,
                  float theta, int iters, int methodTV, int nonneg)
{
    if (nd == 2) dz = 1;
    const size_t n = (size_t)dx * dy * dz;
    Buf<float> I, U[2];
    Buf<T> P[3][2];
    if (I.alloc(n, false) || U[0].alloc(n, false) || U[1].alloc(n, true)) return 1;
    for (auto &c : P) for (auto &b : c) if (b.alloc(n, true)) return 1;
    CK(hipMemcpy(I.p, in, n * sizeof(float), hipMemcpyHostToDevice));
    CK(hipMemcpy(U[0].p, in, n * sizeof(float), hipMemcpyHostToDevice));
    const dim3 block(128, 1, 1), grid((dx + 127) / 128, dy, dz);
    constexpr bool H = sizeof(T) == 2;
    for (int it = 0; it < iters; ++it) {
        const int a = it & 1, b = a ^ 1;
        float *Ui = U[a].p, *Uo = U[b].p;
        T *p1i = P[0][a].p, *p2i = P[1][a].p, *p3i = P[2][a].p, *p1o = P[0][b].p, *p2o = P[1][b].p, *p3o = P[2][b].p;
        using namespace ref_pd;
        if (nd == 3) {
            if constexpr (!H) {
                if (!nonneg && !methodTV) primal_dual_for_total_variation_3D_float<<<grid, block>>>(I.p, Ui, Uo, p1i, p2i, p3i, p1o, p2o, p3o, sigma, tau, lt, theta, dx, dy, dz);
                if (nonneg && !methodTV) primal_dual_for_total_variation_3D_float_nonneg<<<grid, block>>>(I.p, Ui, Uo, p1i, p2i, p3i, p1o, p2o, p3o, sigma, tau, lt, theta, dx, dy, dz);
                if (!nonneg && methodTV) primal_dual_for_total_variation_3D_float_methodTV<<<grid, block>>>(I.p, Ui, Uo, p1i, p2i, p3i, p1o, p2o, p3o, sigma, tau, lt, theta, dx, dy, dz);
                if (nonneg && methodTV) primal_dual_for_total_variation_3D_float_nonneg_methodTV<<<grid, block>>>(I.p, Ui, Uo, p1i, p2i, p3i, p1o, p2o, p3o, sigma, tau, lt, theta, dx, dy, dz);
            } else {
                if (!nonneg && !methodTV) primal_dual_for_total_variation_3D_half<<<grid, block>>>(I.p, Ui, Uo, p1i, p2i, p3i, p1o, p2o, p3o, sigma, tau, lt, theta, dx, dy, dz);
                if (nonneg && !methodTV) primal_dual_for_total_variation_3D_half_nonneg<<<grid, block>>>(I.p, Ui, Uo, p1i, p2i, p3i, p1o, p2o, p3o, sigma, tau, lt, theta, dx, dy, dz);
                if (!nonneg && methodTV) primal_dual_for_total_variation_3D_half_methodTV<<<grid, block>>>(I.p, Ui, Uo, p1i, p2i, p3i, p1o, p2o, p3o, sigma, tau, lt, theta, dx, dy, dz);
                if (nonneg && methodTV) primal_dual_for_total_variation_3D_half_nonneg_methodTV<<<grid, block>>>(I.p, Ui, Uo, p1i, p2i, p3i, p1o, p2o, p3o, sigma, tau, lt, theta, dx, dy, dz);
            }
        } else {
            if constexpr (!H) {
                if (!nonneg && !methodTV) primal_dual_for_total_variation_2D_float<<<grid, block>>>(I.p, Ui, Uo, p1i, p2i, p1o, p2o, sigma, tau, lt, theta, dx, dy);
                if (nonneg && !methodTV) primal_dual_for_total_variation_2D_float_nonneg<<<grid, block>>>(I.p, Ui, Uo, p1i, p2i, p1o, p2o, sigma, tau, lt, theta, dx, dy);
                if (!nonneg && methodTV) primal_dual_for_total_variation_2D_float_methodTV<<<grid, block>>>(I.p, Ui, Uo, p1i, p2i, p1o, p2o, sigma, tau, lt, theta, dx, dy);
                if (nonneg && methodTV) primal_dual_for_total_variation_2D_float_nonneg_methodTV<<<grid, block>>>(I.p, Ui, Uo, p1i, p2i, p1o, p2o, sigma, tau, lt, theta, dx, dy);
            } else {
                if (!nonneg && !methodTV) primal_dual_for_total_variation_2D_half<<<grid, block>>>(I.p, Ui, Uo, p1i, p2i, p1o, p2o, sigma, tau, lt, theta, dx, dy);
                if (nonneg && !methodTV) primal_dual_for_total_variation_2D_half_nonneg<<<grid, block>>>(I.p, Ui, Uo, p1i, p2i, p1o, p2o, sigma, tau, lt, theta, dx, dy);
                if (!nonneg && methodTV) primal_dual_for_total_variation_2D_half_methodTV<<<grid, block>>>(I.p, Ui, Uo, p1i, p2i, p1o, p2o, sigma, tau, lt, theta, dx, dy);
                if (nonneg && methodTV) primal_dual_for_total_variation_2D_half_nonneg_methodTV<<<grid, block>>>(I.p, Ui, Uo, p1i, p2i, p1o, p2o, sigma, tau, lt, theta, dx, dy);
            }
        }
        CK(hipGetLastError());
    }
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(out, U[iters & 1].p, n * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

template <typename T>
static int run_rof(const float *in, float *out, int dx, int dy, int dz, int nd, float lambda, float tau, int iters)
{
    if (nd == 2) dz = 1;
    const size_t n = (size_t)dx * dy * dz;
    Buf<float> I, U[2];
    Buf<T> D1, D2, D3;
    if (I.alloc(n, false) || U[0].alloc(n, false) || U[1].alloc(n, true) || D1.alloc(n, true) || D2.alloc(n, true) ||
        D3.alloc(n, true))
        return 1;
    CK(hipMemcpy(I.p, in, n * sizeof(float), hipMemcpyHostToDevice));
    CK(hipMemcpy(U[0].p, in, n * sizeof(float), hipMemcpyHostToDevice));
    const dim3 block(128, 1, 1), grid((dx + 127) / 128, dy, dz);
    constexpr bool H = sizeof(T) == 2;
    for (int it = 0; it < iters; ++it) {
        float *Ui = U[it & 1].p, *Uo = U[(it + 1) & 1].p;
        using namespace ref_rof;
        if (nd == 3) {
            if constexpr (!H) {
                divergence_kernel_3D_float<<<grid, block>>>(Ui, D1.p, D2.p, D3.p, dx, dy, dz);
                TV_kernel_3D_float<<<grid, block>>>(Ui, Uo, I.p, D1.p, D2.p, D3.p, lambda, tau, dx, dy, dz);
            } else {
                divergence_kernel_3D_half<<<grid, block>>>(Ui, D1.p, D2.p, D3.p, dx, dy, dz);
                TV_kernel_3D_half<<<grid, block>>>(Ui, Uo, I.p, D1.p, D2.p, D3.p, lambda, tau, dx, dy, dz);
            }
        } else {
            if constexpr (!H) {
                divergence_kernel_2D_float<<<grid, block>>>(Ui, D1.p, D2.p, dx, dy);
                TV_kernel_2D_float<<<grid, block>>>(Ui, Uo, I.p, D1.p, D2.p, lambda, tau, dx, dy);
            } else {
                divergence_kernel_2D_half<<<grid, block>>>(Ui, D1.p, D2.p, dx, dy);
                TV_kernel_2D_half<<<grid, block>>>(Ui, Uo, I.p, D1.p, D2.p, lambda, tau, dx, dy);
            }
        }
        CK(hipGetLastError());
    }
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(out, U[iters & 1].p, n * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int ref_pdtv(const float *in, float *out, int dx, int dy, int dz, int nd, float sigma, float tau, float lt,
                        float theta, int iters, int methodTV, int nonneg, int half)
{
    return half ? run_pd<__half>(in, out, dx, dy, dz, nd, sigma, tau, lt, theta, iters, methodTV, nonneg)
                : run_pd<float>(in, out, dx, dy, dz, nd, sigma, tau, lt, theta, iters, methodTV, nonneg);
}

extern "C" int ref_roftv(const float *in, float *out, int dx, int dy, int dz, int nd, float lambda, float tau,
                         int iters, int half)
{
    return half ? run_rof<__half>(in, out, dx, dy, dz, nd, lambda, tau, iters)
                : run_rof<float>(in, out, dx, dy, dz, nd, lambda, tau, iters);
}
