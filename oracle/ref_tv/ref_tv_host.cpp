// oracle/ref_tv/ref_tv_host.cpp -- TEST INFRASTRUCTURE ONLY.
// Host driver that sweeps the reference TV kernels (included via -I$(REF)/tomobar/cuda_kernels, not copied) over the
// launch grid that tomobar/regularisersCuPy.py:93-106,235-249 uses (block (128,1,1),
// grid (ceil(dx/128), dy, dz)) and runs the ping-pong iteration loops of :108-167 / :252-296.
#include <cstdlib>
#include <cstring>
#include <vector>


namespace ref_pd {
#include "primal_dual_for_total_variation.cu"
}
namespace ref_rof {
#include "rudin_osher_fatemi_total_variation.cu"
}

template <typename F>
static void sweep(int dx, int dy, int dz, F &&kernel)
{
    blockDim = {128u, 1u, 1u};
    const unsigned gx = (unsigned)((dx + 127) / 128);
    for (unsigned bz = 0; bz < (unsigned)dz; ++bz)
        for (unsigned by = 0; by < (unsigned)dy; ++by)
            for (unsigned bx = 0; bx < gx; ++bx)
                for (unsigned tx = 0; tx < 128u; ++tx) {
                    blockIdx = {bx, by, bz};
                    threadIdx = {tx, 0u, 0u};
                    kernel();
                }
}

template <typename T>
static int run_pd(const float *in, float *out, int dx, int dy, int dz, int nd, float sigma, float tau, float lt,
                  float theta, int iters, int methodTV, int nonneg)
{
    if (nd == 2) dz = 1;
    size_t n = (size_t)dx * dy * dz;
    std::vector<float> U[2] = {std::vector<float>(in, in + n), std::vector<float>(n, 0.0f)};
    std::vector<T> P[3][2];
    for (auto &c : P) for (auto &b : c) b.assign(n, (T)0);
    float *I = const_cast<float *>(in);
    for (int it = 0; it < iters; ++it) {
        int a = it & 1, b = a ^ 1;
        float *Ui = U[a].data(), *Uo = U[b].data();
        T *p1i = P[0][a].data(), *p2i = P[1][a].data(), *p3i = P[2][a].data();
        T *p1o = P[0][b].data(), *p2o = P[1][b].data(), *p3o = P[2][b].data();
        sweep(dx, dy, dz, [&] {
            using namespace ref_pd;
            if (nd == 3) {
                if (!nonneg && !methodTV) primal_dual_for_total_variation_3D_impl<T, false, false>(I, Ui, Uo, p1i, p2i, p3i, p1o, p2o, p3o, sigma, tau, lt, theta, dx, dy, dz);
                if (nonneg && !methodTV) primal_dual_for_total_variation_3D_impl<T, true, false>(I, Ui, Uo, p1i, p2i, p3i, p1o, p2o, p3o, sigma, tau, lt, theta, dx, dy, dz);
                if (!nonneg && methodTV) primal_dual_for_total_variation_3D_impl<T, false, true>(I, Ui, Uo, p1i, p2i, p3i, p1o, p2o, p3o, sigma, tau, lt, theta, dx, dy, dz);
                if (nonneg && methodTV) primal_dual_for_total_variation_3D_impl<T, true, true>(I, Ui, Uo, p1i, p2i, p3i, p1o, p2o, p3o, sigma, tau, lt, theta, dx, dy, dz);
            } else {
                if (!nonneg && !methodTV) primal_dual_for_total_variation_2D_impl<T, false, false>(I, Ui, Uo, p1i, p2i, p1o, p2o, sigma, tau, lt, theta, dx, dy);
                if (nonneg && !methodTV) primal_dual_for_total_variation_2D_impl<T, true, false>(I, Ui, Uo, p1i, p2i, p1o, p2o, sigma, tau, lt, theta, dx, dy);
                if (!nonneg && methodTV) primal_dual_for_total_variation_2D_impl<T, false, true>(I, Ui, Uo, p1i, p2i, p1o, p2o, sigma, tau, lt, theta, dx, dy);
                if (nonneg && methodTV) primal_dual_for_total_variation_2D_impl<T, true, true>(I, Ui, Uo, p1i, p2i, p1o, p2o, sigma, tau, lt, theta, dx, dy);
            }
        });
    }
    std::memcpy(out, U[iters & 1].data(), n * sizeof(float));
    return 0;
}

template <typename T>
static int run_rof(const float *in, float *out, int dx, int dy, int dz, int nd, float lambda, float tau, int iters)
{
    if (nd == 2) dz = 1;
    size_t n = (size_t)dx * dy * dz;
    std::vector<float> U[2] = {std::vector<float>(in, in + n), std::vector<float>(n, 0.0f)};
    std::vector<T> D1(n), D2(n), D3(n);
    float *I = const_cast<float *>(in);
    for (int it = 0; it < iters; ++it) {
        float *Ui = U[it & 1].data(), *Uo = U[(it + 1) & 1].data();
        sweep(dx, dy, dz, [&] {
            using namespace ref_rof;
            if (nd == 3) divergence_kernel_3D_impl<T>(Ui, D1.data(), D2.data(), D3.data(), dx, dy, dz);
            else divergence_kernel_2D_impl<T>(Ui, D1.data(), D2.data(), dx, dy);
        });
        sweep(dx, dy, dz, [&] {
            using namespace ref_rof;
            if (nd == 3) TV_kernel_3D_impl<T>(Ui, Uo, I, D1.data(), D2.data(), D3.data(), lambda, tau, dx, dy, dz);
            else TV_kernel_2D_impl<T>(Ui, Uo, I, D1.data(), D2.data(), lambda, tau, dx, dy);
        });
    }
    std::memcpy(out, U[iters & 1].data(), n * sizeof(float));
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
