// Host (CPU) 2D parallel-beam projector pair behind RecToolsDIR(..., device_projector="cpu"): BASELINE configs[0], the
// reference's numpy / ASTRA-CPU plumbing path (tomobar/methodsDIR.py:71-175 -> astra_wrappers/astra_base.py:224-232,
// 310-372: ASTRA's CPU `line` projector and `BP` algorithm).  ASTRA is not part of this package; the same operator
// model as the GPU kernels is used instead (voxel-driven two-tap back projection, Joseph forward projection, the
// arithmetic of proj_kernels.hip), so CPU and GPU results of this package are interchangeable.  Pure host code: it
// needs no device (this is an explicitly requested CPU device, not a fallback of the GPU path).
#include "tomo_common.h"

#include <cmath>

extern "C" int tomo_host_bp2d(const float *sino_host, float *img_host, int n, int nu, int na, const double *angles_host,
                              double cor)
{
    TOMO_REQUIRE(sino_host && img_host && angles_host && n > 0 && nu > 0 && na > 0, "bad host back-projection arguments");
    std::vector<float> cs(na), sn(na);
    for (int a = 0; a < na; ++a) { cs[a] = (float)std::cos(angles_host[a]); sn[a] = (float)std::sin(angles_host[a]); }
    const float half_n = 0.5f * (float)n - 0.5f, half_u = 0.5f * (float)nu - 0.5f;
    const float off = half_u - (float)cor;
    for (int iy = 0; iy < n; ++iy) {
        const float yw = (float)iy - half_n;
        for (int ix = 0; ix < n; ++ix) {
            const float xw = (float)ix - half_n;
            float acc = 0.0f;
            for (int a = 0; a < na; ++a) {
                const float f = std::fmaf(xw, cs[a], std::fmaf(yw, sn[a], off));
                const float fl = std::floor(f);
                const float w = f - fl;
                const int i0 = (int)fl;
                const float *row = sino_host + (size_t)a * nu;
                const float s0 = (i0 >= 0 && i0 < nu) ? row[i0] : 0.0f;
                const float s1 = (i0 + 1 >= 0 && i0 + 1 < nu) ? row[i0 + 1] : 0.0f;
                acc = std::fmaf(1.0f - w, s0, acc);
                acc = std::fmaf(w, s1, acc);
            }
            img_host[(size_t)iy * n + ix] = acc;
        }
    }
    return TOMO_OK;
}

extern "C" int tomo_host_fp2d(const float *img_host, float *sino_host, int n, int nu, int na, const double *angles_host,
                              double cor)
{
    TOMO_REQUIRE(sino_host && img_host && angles_host && n > 0 && nu > 0 && na > 0, "bad host forward-projection arguments");
    const float half_n = 0.5f * (float)n - 0.5f, half_u = 0.5f * (float)nu - 0.5f;
    for (int a = 0; a < na; ++a) {
        const double c = std::cos(angles_host[a]), s = std::sin(angles_host[a]);
        const bool dirx = std::fabs(s) >= std::fabs(c);  // step along x, interpolate along y
        const float slope = (float)(dirx ? -c / s : -s / c);
        const float inv = (float)(dirx ? 1.0 / s : 1.0 / c);
        const float scale = (float)(1.0 / (dirx ? std::fabs(s) : std::fabs(c)));
        float *row = sino_host + (size_t)a * nu;
        for (int iu = 0; iu < nu; ++iu) {
            const float sd = ((float)iu - half_u) + (float)cor;
            const float offset = std::fmaf(sd, inv, half_n);
            float acc = 0.0f;
            for (int k = 0; k < n; ++k) {
                const float f = std::fmaf((float)k - half_n, slope, offset);
                const float fl = std::floor(f);
                const float w = f - fl;
                const int i0 = (int)fl;
                float v0 = 0.0f, v1 = 0.0f;
                if (dirx) {
                    if (i0 >= 0 && i0 < n) v0 = img_host[(size_t)i0 * n + k];
                    if (i0 + 1 >= 0 && i0 + 1 < n) v1 = img_host[(size_t)(i0 + 1) * n + k];
                } else {
                    if (i0 >= 0 && i0 < n) v0 = img_host[(size_t)k * n + i0];
                    if (i0 + 1 >= 0 && i0 + 1 < n) v1 = img_host[(size_t)k * n + i0 + 1];
                }
                acc = std::fmaf(1.0f - w, v0, acc);
                acc = std::fmaf(w, v1, acc);
            }
            row[iu] = acc * scale;
        }
    }
    return TOMO_OK;
}
