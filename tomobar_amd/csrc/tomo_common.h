// Internal declarations shared by the translation units of libtomo_mi355x.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "tomo_mi355x.h"

int tomo_fail(int code, const char *fmt, ...);

#define TOMO_HIP(expr)                                                                          \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return tomo_fail(e_ == hipErrorOutOfMemory ? TOMO_E_NOMEM : TOMO_E_RUNTIME,          \
                             "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__,   \
                             __LINE__);                                                         \
    } while (0)

#define TOMO_LAUNCH_CHECK()                                                                     \
    do {                                                                                        \
        hipError_t e_ = hipGetLastError();                                                      \
        if (e_ != hipSuccess)                                                                   \
            return tomo_fail(TOMO_E_RUNTIME, "kernel launch failed: %s (%s:%d)",                \
                             hipGetErrorString(e_), __FILE__, __LINE__);                        \
    } while (0)

#define TOMO_REQUIRE(cond, ...)                                                                 \
    do {                                                                                        \
        if (!(cond)) return tomo_fail(TOMO_E_INVALID, __VA_ARGS__);                              \
    } while (0)

struct tomo_subset {
    int size = 0;             // number of angles
    size_t table_offset = 0;  // element offset into the device angle table (and into the FP order table)
    int n_dirx = 0;           // how many angles step along x (FP)
    // FP stepping classes: the order table lists the subset-local indices of the y-stepping angles (class 0)
    // followed by the x-stepping ones (class 1), each sorted by angle; wbound = cached LDS window bound (-1 = unset)
    // class = 2*dirx + (inv < 0): angles whose detector axis runs the opposite way along the interpolation axis
    // (e.g. theta near 0 and near pi) sample opposite ends of a volume row and must not share a staged window
    int n_class[4] = {0, 0, 0, 0};
    int wbound[4] = {-1, -1, -1, -1};
    int wbound_wide[4] = {-1, -1, -1, -1};  // same for 1024-pixel detector tiles
    int wbound16[4] = {-1, -1, -1, -1};     // same for 256-pixel tiles x 16 angles (dense angle sets)
};

struct tomo_ctx {
    int device = 0;
    int nz = 0, n = 0, nu = 0, na = 0;
    int os = 1, bins = 0;
    unsigned flags = 0;
    std::vector<int64_t> newind;              // [os][bins]
    std::vector<tomo_angle_t> host_table;     // full set followed by every subset
    std::vector<tomo_subset> subsets;         // index 0 = full set, 1 + s = subset s
    tomo_angle_t *dev_table = nullptr;
    std::vector<int> host_fp_order;           // same indexing as host_table
    int *dev_fp_order = nullptr;
    int *dev_fp_mult = nullptr;               // whole-row FP form: lane -> pixel multiplier per entry of the order table (built on first use)
    int fp_mult_bt = 0;                       // ... for this tile width
    void *scratch = nullptr;                  // grow-only (FP: in-plane transposed volume)
    size_t scratch_bytes = 0;
    const float *volT_of = nullptr;           // volume whose transposed copy `scratch` holds (tomo_momentum_transposed)
    bool volT_valid = false;                  // ... valid for exactly the next forward projection of that volume
    hipStream_t volT_stream = nullptr;        // ... on the stream the copy was written on (another stream would race it)
    std::string last_fp_path, last_bp_path;   // which kernels the last FP / BP call ran (tomo_ctx_kernel_path)
    int res_layout = 0;                       // TOMO_RESIDUAL_*: layout of the residual between tomo_fp3d_residual and tomo_bp3d_fista* / _admm
};

// one message per process and key on stderr (a slow fallback kernel was taken)
void tomo_warn_once(const char *key, const char *msg);

// Makes `device` current for the lifetime of the object and restores the caller's current device afterwards: entry
// points never leave the calling thread on another device (torch's current device stays what the caller set).
struct tomo_device_guard {
    int prev = -1;
    hipError_t err = hipSuccess;
    explicit tomo_device_guard(int device)
    {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != device) err = hipSetDevice(device);
        else if (err == hipSuccess) prev = -1;  // nothing to restore
    }
    ~tomo_device_guard() { if (prev >= 0) (void)hipSetDevice(prev); }
    tomo_device_guard(const tomo_device_guard &) = delete;
    tomo_device_guard &operator=(const tomo_device_guard &) = delete;
};
#define TOMO_ON_DEVICE(dev)                 \
    tomo_device_guard device_guard_(dev);   \
    TOMO_HIP(device_guard_.err)

static inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// grow-only scratch arena per (device, stream, slot) (tomo_release_scratch frees a device's arenas)
enum { ARENA_MAIN = 0, ARENA_REDUCE = 1, ARENA_TV = 2 /* placed: the TV operators' work arrays */, ARENA_BPQ = 3 /* a planar sinogram re-laid quad-interleaved for the back projector */, ARENA_CALLER0 = 16 /* .. +7: tomo_placed_scratch (placed) */ };
int tomo_arena_get(int device, hipStream_t stream, int slot, size_t bytes, void **out, bool place = false);  // place: see tomo_api.hip
int tomo_arena_release_slot(int device, int slot);  // frees the arenas of one slot on a device (all streams)
void tomo_bp_relay_reset(int device);         // forget that the back projector's relay scratch was refused (proj_kernels.hip)
void tomo_fourier_cache_release(int device);  // cached hipFFT plans of fourier_inv.hip
void tomo_fbp_cache_release(int device);      // cached hipFFT plans / filter tables of fbp_filter.hip

// kernel-variant switches (tomo_set_variant).  Two flavours of the library are built from these sources:
//   libtomo_mi355x.so      (shipped): every kernel class runs its default (variant 0); "pdtv" additionally accepts 22 =
//                          the reference's roundings for float32 duals (bit-identical to the oracle; the default is within
//                          1e-5).  No measurement switches are compiled in: g_probe is the constant 0 and every
//                          `probe &` test folds away.
//   libtomo_mi355x_dev.so  (-DTOMO_DEV_VARIANTS; tests and tools/ only): the independent implementations and A/B builds
//                          (bp 1/2, fp 1/2/3/4, pdtv 1/2/3/21, roftv 1/2/3/4) and the "probe" bits of tools/*_probe.py.
extern thread_local int g_variant_bp, g_variant_fp, g_variant_pdtv, g_variant_roftv;
#ifdef TOMO_DEV_VARIANTS
#define TOMO_DEV 1
extern thread_local int g_probe;
#else
#define TOMO_DEV 0
constexpr int g_probe = 0;
#endif

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// ---- optional event timing of launch groups (tomo_profile_enable / tomo_profile_read)
enum { PROF_BP = 0, PROF_FP = 1, PROF_PDTV = 2, PROF_ROFTV = 3, PROF_CLASSES = 4 };
struct tomo_prof_scope {
    int cls;
    hipStream_t st;
    int launches;
    void *rec;  // opaque
    tomo_prof_scope(int cls, hipStream_t st, int launches);
    ~tomo_prof_scope();
};
