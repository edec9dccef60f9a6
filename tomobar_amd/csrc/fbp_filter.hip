// Sinc-ramp filtering of projection rows for FBP.
// Replaces tomobar/fourier.py:26-78 (_filtersinc3D_cupy: rfft along detX, multiply by the filter built by
// cuda_kernels/generate_filtersync.cu:5-82, unnormalised irfft with the 1/n and 1/Na factors folded into the filter).
// The FFTs are plain batched 1D real transforms -> hipFFT (library call, like the reference's cuFFT via CuPy); the filter
// table is tiny and is built on the host with the reference kernel's float32 formulas.
#include "tomo_common.h"

#include <hipfft/hipfft.h>

#include <cmath>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

namespace {

#define TOMO_FFT(expr)                                                                              \
    do {                                                                                            \
        hipfftResult r_ = (expr);                                                                   \
        if (r_ != HIPFFT_SUCCESS) return tomo_fail(TOMO_E_RUNTIME, "%s failed: hipfft status %d", #expr, (int)r_); \
    } while (0)

__global__ __launch_bounds__(256) void apply_filter_kernel(float2 *spec, const float *__restrict__ f, size_t rows, int nh)
{
    const size_t total = rows * (size_t)nh;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const float g = f[i % nh];
        float2 v = spec[i];
        v.x *= g;
        v.y *= g;
        spec[i] = v;
    }
}

// half-spectrum filter, fftshift-ed: f[(i + n/2) % n] = |2/a sin(a w_i/2)| * (sum sin(a w/2)(a w/2) / sum (a w/2)^2)^2 * mult
std::vector<float> make_filter(int n, float a, float mult)
{
    const float pi = 3.1415926535897932384626433832795f;
    const float dw = 2 * pi / n;
    float sum = 0.0f;
    for (int i = 0; i < n; ++i) {
        const float w = -pi + i * dw;
        const float rd = a * w / 2.0f;
        sum += rd * rd;
    }
    float dot = 0.0f;
    for (int i = 0; i < n; ++i) {
        const float w = -pi + i * dw;
        const float rd = a * w / 2.0f;
        dot += sinf(rd) * rd / sum;
    }
    const float dot2 = dot * dot;
    std::vector<float> f(n / 2 + 1, 0.0f);
    for (int i = 0; i < n; ++i) {
        const int out = (i + n / 2) % n;
        if (out >= n / 2 + 1) continue;
        const float w = -pi + i * dw;
        const float rd = a * w / 2.0f;
        const float rn1 = (float)std::fabs(2.0 / (double)a * (double)sinf(rd));
        f[out] = rn1 * dot2 * mult;
    }
    return f;
}

// hipFFT plan pairs per (device, stream, nu, rows) and device filter tables per (device, nu, cutoff, multiplier) are kept
// between calls (plan creation costs more than the transforms of a small batch); tomo_release_scratch frees them.
// The stream is part of the key: a plan owns ONE work area, so two streams must never execute the same plan concurrently
// (successive calls on one stream are ordered by the stream itself).  At most FBP_MAX_PLANS pairs are kept per process,
// least recently used first out (every distinct `rows` value is a plan pair with its own work area).
// `done` is recorded after every execution: eviction / release wait on THAT event before the work areas are freed, never
// on the stream handle -- the caller may have destroyed the stream since (CuPy / user-created streams), and an event
// owned by the library stays valid whatever happened to the stream it was recorded on.
struct fbp_plans { hipfftHandle fwd = 0, inv = 0; hipEvent_t done = nullptr; unsigned long long used = 0; };

void fbp_destroy(fbp_plans &p)
{
    if (p.done) { (void)hipEventSynchronize(p.done); (void)hipEventDestroy(p.done); }
    if (p.fwd) (void)hipfftDestroy(p.fwd);
    if (p.inv) (void)hipfftDestroy(p.inv);
    p = fbp_plans{};
}
std::mutex g_fbp_mu;
constexpr size_t FBP_MAX_PLANS = 8;
unsigned long long g_fbp_tick = 0;
std::map<std::tuple<int, hipStream_t, int, size_t>, fbp_plans> g_fbp_plans;
std::map<std::tuple<int, int, float, float>, float *> g_fbp_filters;

int fbp_get_plans(int device, hipStream_t st, int nu, size_t rows, fbp_plans &out)
{
    const auto key = std::make_tuple(device, st, nu, rows);
    auto it = g_fbp_plans.find(key);
    if (it != g_fbp_plans.end()) { it->second.used = ++g_fbp_tick; out = it->second; return TOMO_OK; }
    if (g_fbp_plans.size() >= FBP_MAX_PLANS) {
        auto victim = g_fbp_plans.begin();
        for (auto jt = g_fbp_plans.begin(); jt != g_fbp_plans.end(); ++jt)
            if (jt->second.used < victim->second.used) victim = jt;
        // the victim's last execution may still be running: wait for its event, then free the work areas
        tomo_device_guard guard(std::get<0>(victim->first));
        fbp_destroy(victim->second);
        g_fbp_plans.erase(victim);
    }
    const int nh = nu / 2 + 1;
    int n[1] = {nu};
    fbp_plans p;
    hipfftResult r = hipfftPlanMany(&p.fwd, 1, n, nullptr, 1, nu, nullptr, 1, nh, HIPFFT_R2C, (int)rows);
    if (r == HIPFFT_SUCCESS) r = hipfftPlanMany(&p.inv, 1, n, nullptr, 1, nh, nullptr, 1, nu, HIPFFT_C2R, (int)rows);
    if (r == HIPFFT_SUCCESS) r = hipfftSetStream(p.fwd, st);
    if (r == HIPFFT_SUCCESS) r = hipfftSetStream(p.inv, st);
    if (r == HIPFFT_SUCCESS && hipEventCreateWithFlags(&p.done, hipEventDisableTiming) != hipSuccess) { p.done = nullptr; r = HIPFFT_INTERNAL_ERROR; }
    if (r != HIPFFT_SUCCESS) {  // nothing half-built is kept or leaked
        fbp_destroy(p);
        return tomo_fail(TOMO_E_RUNTIME, "hipfftPlanMany failed: hipfft status %d", (int)r);
    }
    p.used = ++g_fbp_tick;
    g_fbp_plans[key] = p;
    out = p;
    return TOMO_OK;
}

int fbp_get_filter(int device, int nu, float cutoff, float multiplier, const float **out)
{
    const auto key = std::make_tuple(device, nu, cutoff, multiplier);
    auto it = g_fbp_filters.find(key);
    if (it != g_fbp_filters.end()) { *out = it->second; return TOMO_OK; }
    const std::vector<float> f = make_filter(nu, cutoff, multiplier);
    float *dev = nullptr;
    TOMO_HIP(hipMalloc((void **)&dev, f.size() * sizeof(float)));
    hipError_t e = hipMemcpy(dev, f.data(), f.size() * sizeof(float), hipMemcpyHostToDevice);  // synchronous, once
    if (e != hipSuccess) {
        (void)hipFree(dev);
        return tomo_fail(TOMO_E_RUNTIME, "filter upload failed: %s", hipGetErrorString(e));
    }
    g_fbp_filters[key] = dev;
    *out = dev;
    return TOMO_OK;
}

}  // namespace

void tomo_fbp_cache_release(int device)
{
    std::lock_guard<std::mutex> lk(g_fbp_mu);
    tomo_device_guard guard(device);
    for (auto it = g_fbp_plans.begin(); it != g_fbp_plans.end();) {
        if (std::get<0>(it->first) == device) {
            fbp_destroy(it->second);   // waits for the plan's last execution (its event), whatever became of the stream
            it = g_fbp_plans.erase(it);
        } else ++it;
    }
    for (auto it = g_fbp_filters.begin(); it != g_fbp_filters.end();) {
        if (std::get<0>(it->first) == device) {
            (void)hipFree(it->second);
            it = g_fbp_filters.erase(it);
        } else ++it;
    }
}

extern "C" int tomo_fbp_filter(int device, float *data_dev, size_t rows, int nu, float cutoff, float multiplier,
                               void *stream)
{
    TOMO_REQUIRE(device >= 0 && data_dev != nullptr && nu >= 2 && cutoff > 0.0f, "bad FBP filter arguments");
    if (rows == 0) return TOMO_OK;
    TOMO_REQUIRE(rows <= 0x7fffffffULL, "too many projection rows for one hipFFT plan");
    TOMO_ON_DEVICE(device);
    hipStream_t st = as_stream(stream);
    const int nh = nu / 2 + 1;
    const size_t spec_bytes = rows * (size_t)nh * sizeof(float2);
    void *base = nullptr;
    int rc = tomo_arena_get(device, st, ARENA_MAIN, spec_bytes, &base);
    if (rc != TOMO_OK) return rc;
    float2 *spec = (float2 *)base;
    std::lock_guard<std::mutex> lk(g_fbp_mu);  // guards the caches; a plan belongs to one (device, stream)
    const float *filt = nullptr;
    rc = fbp_get_filter(device, nu, cutoff, multiplier, &filt);
    if (rc != TOMO_OK) return rc;
    fbp_plans p;
    rc = fbp_get_plans(device, st, nu, rows, p);
    if (rc != TOMO_OK) return rc;
    TOMO_FFT(hipfftExecR2C(p.fwd, data_dev, (hipfftComplex *)spec));
    size_t total = rows * (size_t)nh;
    size_t grid = (total + 255) / 256;
    if (grid > 4096) grid = 4096;
    apply_filter_kernel<<<(unsigned)grid, 256, 0, st>>>(spec, filt, rows, nh);
    TOMO_LAUNCH_CHECK();
    TOMO_FFT(hipfftExecC2R(p.inv, (hipfftComplex *)spec, data_dev));
    TOMO_HIP(hipEventRecord(p.done, st));
    return TOMO_OK;  // asynchronous on `st`: no host synchronisation
}
