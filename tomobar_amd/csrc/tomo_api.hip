// Context, geometry tables, error reporting and scratch arenas of libtomo_mi355x.so.
#include "tomo_common.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <ctime>

static thread_local char g_err[512] = "";

int tomo_fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// test / A-B switches (tomo_set_variant), per host thread: one thread's choice never changes what another launches
thread_local int g_variant_bp = 0, g_variant_fp = 0, g_variant_pdtv = 0, g_variant_roftv = 0;
#if TOMO_DEV
thread_local int g_probe = 0;  // measurement-only switches of tools/ (tomo_set_variant("probe", bits)); dev flavour only
#endif

extern "C" int tomo_abi_version(void) { return TOMO_ABI_VERSION; }
extern "C" const char *tomo_build_flavour(void) { return TOMO_DEV ? "dev" : "shipped"; }
extern "C" const char *tomo_last_error(void) { return g_err; }

extern "C" int tomo_device_count(int *count)
{
    TOMO_REQUIRE(count != nullptr, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) {
        *count = 0;
        return tomo_fail(TOMO_E_NODEVICE, "no HIP device visible (%s); libtomo_mi355x has no CPU fallback",
                         e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    }
    *count = n;
    return TOMO_OK;
}

extern "C" int tomo_set_variant(const char *kernel, int variant)
{
    TOMO_REQUIRE(kernel != nullptr, "kernel name is NULL");
    std::string k(kernel);
    // what each flavour was compiled with (tomo_common.h): selecting a variant whose kernels are not in this library is an
    // error, never a silent fall-through to another kernel
    auto allowed = [&](std::initializer_list<int> shipped, std::initializer_list<int> dev) {
        for (int v : shipped) if (v == variant) return true;
        if (TOMO_DEV) for (int v : dev) if (v == variant) return true;
        return false;
    };
    int *slot = nullptr;
    bool ok = false;
    if (k == "bp") { slot = &g_variant_bp; ok = allowed({0}, {1, 2, 3}); }
    else if (k == "fp") { slot = &g_variant_fp; ok = allowed({0}, {1, 2, 3, 4}); }   // 4 = whole-row form without the per-angle lane multipliers (A/B)
    else if (k == "pdtv") { slot = &g_variant_pdtv; ok = allowed({0, 22}, {1, 2, 3, 21, 31, 32}); }
    else if (k == "roftv") { slot = &g_variant_roftv; ok = allowed({0}, {1, 2, 3, 4}); }
#if TOMO_DEV
    else if (k == "probe") { g_probe = variant; return TOMO_OK; }
#endif
    else return tomo_fail(TOMO_E_INVALID, "unknown kernel '%s'%s", kernel,
                          (!TOMO_DEV && k == "probe") ? " (measurement switches exist in libtomo_mi355x_dev.so only)" : "");
    if (!ok) return tomo_fail(TOMO_E_INVALID, "variant %d of '%s' is not part of this library (%s flavour)%s", variant, kernel,
                              tomo_build_flavour(), TOMO_DEV ? "" : ": the A/B variants live in libtomo_mi355x_dev.so");
    *slot = variant;
    return TOMO_OK;
}

// ---- per-angle record: rays (sin,-cos,0), u = (cos,sin,0), detector centre cor*u (supp/funcs.py:45-65)
static tomo_angle_t make_angle(double theta, double cor, int src)
{
    tomo_angle_t t;
    const double c = std::cos(theta), s = std::sin(theta);
    t.cs = (float)c;
    t.sn = (float)s;
    t.cor = (float)cor;
    t.dirx = std::fabs(s) >= std::fabs(c) ? 1 : 0;
    if (t.dirx) {
        t.slope = (float)(-c / s);
        t.inv = (float)(1.0 / s);
        t.scale = (float)(1.0 / std::fabs(s));
    } else {
        t.slope = (float)(-s / c);
        t.inv = (float)(1.0 / c);
        t.scale = (float)(1.0 / std::fabs(c));
    }
    t.src = src;
    return t;
}

extern "C" int tomo_ctx_create(int device, int nz, int n, int nu, int na, const double *angles_host,
                               const double *cor_host, int cor_stride, int os_number, unsigned flags,
                               tomo_ctx **out)
{
    TOMO_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    TOMO_REQUIRE(nu > 0, "The size of the horizontal detector cannot be negative or zero");
    TOMO_REQUIRE(nz > 0, "The size of the vertical detector cannot be negative or zero");
    TOMO_REQUIRE(n > 0, "The size of the reconstruction object cannot be zero");
    TOMO_REQUIRE(na > 0 && angles_host != nullptr, "The length of angles array cannot be zero");
    TOMO_REQUIRE(os_number > 0, "The number of ordered subsets cannot be negative or zero");
    TOMO_REQUIRE(cor_host != nullptr && cor_stride >= 0 && cor_stride <= 2, "bad centre-of-rotation argument");
    TOMO_REQUIRE(device >= 0, "The GPU device index must be >= 0");
    if (cor_stride == 2)
        for (int a = 0; a < na; ++a)
            TOMO_REQUIRE(cor_host[2 * a + 1] == 0.0,
                         "the context takes horizontal offsets only: apply the vertical component (angle %d) with "
                         "tomo_shift_rows around the projectors, as HipTools3D does", a);
    int ndev = 0;
    int rc = tomo_device_count(&ndev);
    if (rc != TOMO_OK) return rc;
    TOMO_REQUIRE(device < ndev, "device index %d out of range (%d devices)", device, ndev);
    TOMO_ON_DEVICE(device);

    tomo_ctx *ctx = new (std::nothrow) tomo_ctx();
    if (!ctx) return tomo_fail(TOMO_E_NOMEM, "out of host memory");
    ctx->device = device;
    ctx->nz = nz; ctx->n = n; ctx->nu = nu; ctx->na = na;
    ctx->os = os_number;
    ctx->flags = flags;
    ctx->bins = (int)std::ceil((double)na / (double)os_number);

    auto cor_of = [&](int a) { return cor_stride == 0 ? cor_host[0] : cor_host[(size_t)a * cor_stride]; };
    auto push_subset = [&](const std::vector<int64_t> &idx) {
        tomo_subset s;
        s.size = (int)idx.size();
        s.table_offset = ctx->host_table.size();
        for (int64_t a : idx) {
            ctx->host_table.push_back(make_angle(angles_host[a], cor_of((int)a), (int)a));
            s.n_dirx += ctx->host_table.back().dirx;
        }
        // FP order table: class 0 (y-stepping) then class 1 (x-stepping), each sorted by the march slope so that the
        // angles one workgroup handles together sample neighbouring parts of every volume row
        const tomo_angle_t *tab = ctx->host_table.data() + s.table_offset;
        std::vector<int> order[4];
        for (int i = 0; i < s.size; ++i) order[2 * tab[i].dirx + (tab[i].inv < 0.0f ? 1 : 0)].push_back(i);
        for (int c = 0; c < 4; ++c) {
            std::stable_sort(order[c].begin(), order[c].end(), [&](int p, int q) { return tab[p].slope < tab[q].slope; });
            s.n_class[c] = (int)order[c].size();
            ctx->host_fp_order.insert(ctx->host_fp_order.end(), order[c].begin(), order[c].end());
        }
        ctx->subsets.push_back(s);
    };
    std::vector<int64_t> all(na);
    for (int a = 0; a < na; ++a) all[a] = a;
    push_subset(all);
    // interleaved subsets s, s+OS, ... (astra_base.py:195-209); tail entries stay 0
    ctx->newind.assign((size_t)os_number * ctx->bins, 0);
    for (int s = 0; s < os_number; ++s)
        for (int k = 0; k < ctx->bins; ++k) {
            int64_t a = (int64_t)s + (int64_t)k * os_number;
            if (a < na) ctx->newind[(size_t)s * ctx->bins + k] = a;
        }
    if (os_number > 1) {
        for (int s = 0; s < os_number; ++s) {
            std::vector<int64_t> idx(ctx->newind.begin() + (size_t)s * ctx->bins,
                                     ctx->newind.begin() + (size_t)(s + 1) * ctx->bins);
            // consumers drop ONE trailing element when it is 0 (methodsIR_CuPy.py:454-456, astra_base.py:291-293)
            if (idx[ctx->bins - 1] == 0) idx.pop_back();
            push_subset(idx);
        }
    }
    size_t bytes = ctx->host_table.size() * sizeof(tomo_angle_t);
    hipError_t e = hipMalloc((void **)&ctx->dev_table, bytes > 0 ? bytes : sizeof(tomo_angle_t));
    if (e == hipSuccess && bytes)
        e = hipMemcpy(ctx->dev_table, ctx->host_table.data(), bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        const size_t ob = ctx->host_fp_order.size() * sizeof(int);
        e = hipMalloc((void **)&ctx->dev_fp_order, ob > 0 ? ob : sizeof(int));
        if (e == hipSuccess && ob)
            e = hipMemcpy(ctx->dev_fp_order, ctx->host_fp_order.data(), ob, hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        if (ctx->dev_table) (void)hipFree(ctx->dev_table);
        if (ctx->dev_fp_order) (void)hipFree(ctx->dev_fp_order);
        delete ctx;
        return tomo_fail(TOMO_E_RUNTIME, "angle table upload failed: %s", hipGetErrorString(e));
    }
    *out = ctx;
    return TOMO_OK;
}

extern "C" int tomo_ctx_destroy(tomo_ctx *ctx)
{
    if (!ctx) return TOMO_OK;
    tomo_device_guard guard(ctx->device);
    if (ctx->dev_table) (void)hipFree(ctx->dev_table);
    if (ctx->dev_fp_order) (void)hipFree(ctx->dev_fp_order);
    if (ctx->dev_fp_mult) (void)hipFree(ctx->dev_fp_mult);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    delete ctx;
    return TOMO_OK;
}

extern "C" int tomo_ctx_release_scratch(tomo_ctx *ctx)
{
    TOMO_REQUIRE(ctx != nullptr, "ctx is NULL");
    ctx->volT_valid = false;  // the transposed copy lives in the scratch that is about to go
    ctx->volT_of = nullptr;
    if (ctx->scratch) {
        TOMO_ON_DEVICE(ctx->device);
        TOMO_HIP(hipDeviceSynchronize());
        TOMO_HIP(hipFree(ctx->scratch));
        ctx->scratch = nullptr;
        ctx->scratch_bytes = 0;
    }
    // the back projector's relay block (a planar sinogram re-laid quad-interleaved: as large as the sinogram, e.g. 3.7 GB for
    // FBP of 1024^3 x 900) lives in the device's arena, outside the caller's allocator: a context that gives its scratch back
    // gives that back as well (it is re-made on the next tomo_bp3d of any context on this device)
    return tomo_arena_release_slot(ctx->device, ARENA_BPQ);
}

void tomo_warn_once(const char *key, const char *msg)
{
    static std::mutex mu;
    static std::map<std::string, bool> seen;
    std::lock_guard<std::mutex> lk(mu);
    if (seen[key]) return;
    seen[key] = true;
    fprintf(stderr, "libtomo_mi355x warning: %s\n", msg);
}

extern "C" const char *tomo_ctx_kernel_path(const tomo_ctx *ctx, const char *op)
{
    if (!ctx || !op) return "";
    return std::string(op) == "bp" ? ctx->last_bp_path.c_str() : ctx->last_fp_path.c_str();
}

extern "C" int tomo_ctx_os_number(const tomo_ctx *ctx) { return ctx ? ctx->os : -1; }
extern "C" int tomo_ctx_num_bins(const tomo_ctx *ctx) { return ctx ? ctx->bins : -1; }

extern "C" int tomo_ctx_newind_table(const tomo_ctx *ctx, int64_t *out_host)
{
    TOMO_REQUIRE(ctx != nullptr && out_host != nullptr, "NULL argument");
    std::memcpy(out_host, ctx->newind.data(), ctx->newind.size() * sizeof(int64_t));
    return TOMO_OK;
}

static const tomo_subset *find_subset(const tomo_ctx *ctx, int subset)
{
    if (!ctx) return nullptr;
    if (subset < 0 || ctx->os == 1) return &ctx->subsets[0];
    if (subset >= ctx->os) return nullptr;
    return &ctx->subsets[1 + subset];
}

extern "C" int tomo_ctx_subset_size(const tomo_ctx *ctx, int subset)
{
    const tomo_subset *s = find_subset(ctx, subset);
    return s ? s->size : -1;
}

extern "C" int tomo_ctx_angle_table(const tomo_ctx *ctx, int subset, tomo_angle_t *out_host, int capacity)
{
    const tomo_subset *s = find_subset(ctx, subset);
    TOMO_REQUIRE(s != nullptr && out_host != nullptr, "bad subset %d", subset);
    TOMO_REQUIRE(capacity >= s->size, "capacity %d < subset size %d", capacity, s->size);
    std::memcpy(out_host, ctx->host_table.data() + s->table_offset, (size_t)s->size * sizeof(tomo_angle_t));
    return TOMO_OK;
}

// ---- grow-only scratch arenas, one per (device, stream, slot): two streams of one device never alias scratch, and
//      growing an arena waits for THAT stream only
struct arena_t { void *ptr = nullptr; size_t bytes = 0; };
struct arena_key {
    int device; hipStream_t stream; int slot;
    bool operator<(const arena_key &o) const
    {
        if (device != o.device) return device < o.device;
        if (stream != o.stream) return stream < o.stream;
        return slot < o.slot;
    }
};
static std::mutex g_arena_mu;
static std::map<arena_key, arena_t> g_arenas;

// ---- placement of large scratch blocks (round 4).  On MI355X the speed of a plane-marching kernel depends on WHERE in HBM
// its arrays were allocated: the same PD_TV launch on the same volume takes 10.1 ms with its scratch arena in one block of
// device memory and 9.1-9.3 ms in another (same process, same kernel, profiles/archive/r4z_pd_arena_placement.txt); a flat copy does
// not see the difference (6.1-6.3 TB/s everywhere), a z-march over one array of 4 MB planes does (4.9 vs 5.3 TB/s per 8.6 GB
// chunk, two levels, roughly a quarter of the device in the slow one and a different quarter in every process).  So a block
// of >= 1 GiB is chosen among up to `tries` candidate allocations held at the same time: each is scored with a z-march
// probe over the whole block (~7 ms per 34 GB), the search stops at the first candidate that beats an earlier one by 8 %
// (blocks score 4.87, 5.12 or 5.29 TB/s -- the launch then takes 10.1, 9.5 or 9.2 ms --: that one is in the fast class), the best is kept and the others are freed.  TOMO_MI355X_PLACE_TRIES=1 (or
// tomo_set_placement_tries(1)) turns the search off; candidates are only taken while the device has room for them.
namespace {
// one array of planes of 1024 x 1024 floats: a workgroup (2 x 2 waves) owns 128 columns x 16 rows (+3 halo rows either side)
// and walks the planes of its z-chunk, reading 14 rows per wave and plane from the first half of the block and writing 8 into
// the second half -- the access shape of the TV kernels with one input and one output stream
__global__ __launch_bounds__(256) void place_probe_kernel(const float *__restrict__ a, float *__restrict__ b, int nz, int zchunk)
{
    const int n = 1024, gx = 8, gy = 64;
    const int tile = (int)blockIdx.x % (gx * gy), chunk = (int)blockIdx.x / (gx * gy);
    const int xb = tile % gx, yb = tile / gx;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = xb * 128 + (wave & 1) * 64 + lane, y0 = yb * 16 + (wave >> 1) * 8;
    const int z1 = min((chunk + 1) * zchunk, nz);
    for (int z = chunk * zchunk; z < z1; ++z) {
        const size_t pl = (size_t)z * n * n;
        float s = 0.0f;
#pragma unroll
        for (int r = -3; r < 11; ++r) s += a[pl + (size_t)min(max(y0 + r, 0), n - 1) * n + x];
#pragma unroll
        for (int r = 0; r < 8; ++r) b[pl + (size_t)(y0 + r) * n + x] = s;
    }
}
constexpr size_t PLACE_MIN_BYTES = (size_t)1 << 30;
constexpr size_t PLACE_HEADROOM = (size_t)4 << 30;  // left free on the device while candidates are held
constexpr double PLACE_HOLD_FRACTION = 0.85;         // ... and the candidates together never hold more than this share of
                                                     // what was free when the search began (co-tenants keep the rest)
constexpr int PLACE_MAX_TRIES = 10;
constexpr int PLACE_DEFAULT_TRIES = 8;
constexpr double PLACE_FAST_GAIN = 1.08;             // a candidate this much above an earlier one is in the fast class
int g_place_tries = -1;
struct place_rec { size_t bytes = 0; int tries = 0, chosen = -1, fast = 0; double score[PLACE_MAX_TRIES] = {0}; } g_place_last;
// one search at a time per process (two concurrent searches would each hold their candidates), and the record above;
// NOT the arena map's mutex: a search takes 0.1-4 s and other streams' arenas must stay reachable meanwhile
std::mutex g_place_mu;

int placement_tries()
{
    if (g_place_tries < 0) {
        const char *e = std::getenv("TOMO_MI355X_PLACE_TRIES");
        const int v = e ? std::atoi(e) : PLACE_DEFAULT_TRIES;
        g_place_tries = v < 1 ? 1 : (v > PLACE_MAX_TRIES ? PLACE_MAX_TRIES : v);
    }
    return g_place_tries;
}

// GB/s of the probe over the block (the better of two passes after a warm-up); 0 on any error
double place_score(void *p, size_t bytes, hipStream_t st)
{
    const int nz = (int)(bytes / 2 / ((size_t)4 << 20));
    if (nz < 16) return 0.0;
    const int chunks = 16, zchunk = (nz + chunks - 1) / chunks;
    const float *a = (const float *)p;
    float *b = (float *)((char *)p + (size_t)nz * ((size_t)4 << 20));
    hipEvent_t e[3];
    int made = 0;
    for (; made < 3; ++made)
        if (hipEventCreateWithFlags(&e[made], hipEventDefault) != hipSuccess) break;
    double best = 0.0;
    bool ok = made == 3;
    if (ok) {
        place_probe_kernel<<<512 * chunks, 256, 0, st>>>(a, b, nz, zchunk);
        ok = hipGetLastError() == hipSuccess;
    }
    for (int r = 0; r < 2 && ok; ++r) {
        ok = hipEventRecord(e[r], st) == hipSuccess;
        place_probe_kernel<<<512 * chunks, 256, 0, st>>>(a, b, nz, zchunk);
        ok = ok && hipGetLastError() == hipSuccess;
        ok = ok && hipEventRecord(e[r + 1], st) == hipSuccess && hipEventSynchronize(e[r + 1]) == hipSuccess;
        float ms = 0.0f;
        ok = ok && hipEventElapsedTime(&ms, e[r], e[r + 1]) == hipSuccess && ms > 0.0f;
        if (ok) best = std::max(best, 2.0 * nz * (double)((size_t)4 << 20) / ms / 1e6);
    }
    for (int k = 0; k < made; ++k) (void)hipEventDestroy(e[k]);
    if (!ok) (void)hipGetLastError();
    return ok ? best : 0.0;
}

// plain allocation; a first failure is retried a few times: another process sharing the GPU may be holding placement
// candidates for a fraction of a second (ranks that share a device in the functional multi-process tests)
int patient_malloc(size_t bytes, void **out, int attempts = 4)
{
    hipError_t e = hipSuccess;
    for (int attempt = 0; attempt < attempts; ++attempt) {
        e = hipMalloc(out, bytes);
        if (e == hipSuccess) return TOMO_OK;
        (void)hipGetLastError();
        if (e != hipErrorOutOfMemory || attempt + 1 == attempts) break;
        struct timespec ts = {0, 300 * 1000 * 1000};
        nanosleep(&ts, nullptr);
    }
    *out = nullptr;
    return tomo_fail(e == hipErrorOutOfMemory ? TOMO_E_NOMEM : TOMO_E_RUNTIME, "hipMalloc of %zu bytes of scratch failed: %s", bytes,
                     hipGetErrorString(e));
}

int placed_malloc(hipStream_t st, size_t bytes, void **out)
{
    std::lock_guard<std::mutex> lk(g_place_mu);
    const int tries = placement_tries();
    if (bytes < PLACE_MIN_BYTES || tries <= 1) return patient_malloc(bytes, out);
    size_t free0 = 0, total = 0;
    if (hipMemGetInfo(&free0, &total) != hipSuccess) { (void)hipGetLastError(); return patient_malloc(bytes, out); }
    const size_t hold_cap = (size_t)(PLACE_HOLD_FRACTION * (double)free0);
    void *cand[PLACE_MAX_TRIES];
    place_rec rec;
    rec.bytes = bytes;
    double lo = 0.0;
    for (int t = 0; t < tries; ++t) {
        if (t > 0) {
            size_t fr = 0, tot = 0;
            if ((size_t)(t + 1) * bytes > hold_cap) break;
            if (hipMemGetInfo(&fr, &tot) != hipSuccess || fr < bytes + PLACE_HEADROOM) { (void)hipGetLastError(); break; }
        }
        void *p = nullptr;
        if (t == 0) {
            const int rc = patient_malloc(bytes, &p);
            if (rc != TOMO_OK) return rc;
        } else if (hipMalloc(&p, bytes) != hipSuccess) {
            (void)hipGetLastError();
            break;
        }
        cand[t] = p;
        rec.score[t] = place_score(p, bytes, st);
        rec.tries = t + 1;
        if (rec.chosen < 0 || rec.score[t] > rec.score[rec.chosen]) rec.chosen = t;
        lo = (t == 0) ? rec.score[t] : std::min(lo, rec.score[t]);
        // levels 4.87 / 5.12 / 5.29 TB/s: the best is in the fast class once it beats an earlier candidate by 8 %
        if (t >= 1 && lo > 0.0 && rec.score[rec.chosen] >= PLACE_FAST_GAIN * lo) { rec.fast = 1; break; }
    }
    for (int t = 0; t < rec.tries; ++t)
        if (t != rec.chosen) (void)hipFree(cand[t]);
    *out = cand[rec.chosen];
    g_place_last = rec;
    return TOMO_OK;
}
}  // namespace

extern "C" int tomo_set_placement_tries(int tries)
{
    TOMO_REQUIRE(tries >= 1 && tries <= PLACE_MAX_TRIES, "placement tries must be 1 .. %d", PLACE_MAX_TRIES);
    std::lock_guard<std::mutex> lk(g_place_mu);
    g_place_tries = tries;
    return TOMO_OK;
}

extern "C" int tomo_placement_tries(void)
{
    std::lock_guard<std::mutex> lk(g_place_mu);
    return placement_tries();
}

extern "C" int tomo_placement_last(size_t *bytes, int *chosen, double *scores_GBps, int capacity)
{
    std::lock_guard<std::mutex> lk(g_place_mu);
    if (bytes) *bytes = g_place_last.bytes;
    if (chosen) *chosen = g_place_last.chosen;
    for (int t = 0; scores_GBps && t < capacity && t < g_place_last.tries; ++t) scores_GBps[t] = g_place_last.score[t];
    return g_place_last.tries;
}

extern "C" int tomo_placement_last_fast(void)
{
    std::lock_guard<std::mutex> lk(g_place_mu);
    return g_place_last.tries > 0 ? g_place_last.fast : -1;
}

// `place`: the block holds plane-marching work arrays (the TV kernels' arena, tomo_placed_scratch): choose it by the
// placement search.  FBP spectra, Fourier and reduction scratch are plain allocations.
int tomo_arena_get(int device, hipStream_t stream, int slot, size_t bytes, void **out, bool place)
{
    void *old = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_arena_mu);
        arena_t &a = g_arenas[arena_key{device, stream, slot}];
        if (a.bytes >= bytes) { *out = a.ptr; return TOMO_OK; }
        old = a.ptr;
        a.ptr = nullptr;
        a.bytes = 0;
    }
    // grow outside the map's mutex: the stream's own work is the only user of this arena (two host threads driving ONE
    // stream's scratch at once is a caller error, detected below)
    if (old) {   // the old block is released whatever happens next (the map no longer knows it)
        const hipError_t es = hipStreamSynchronize(stream);
        const hipError_t ef = hipFree(old);
        TOMO_HIP(es);
        TOMO_HIP(ef);
    }
    void *p = nullptr;
    // (the back projector's relay scratch is an optimisation it can do without: one attempt, no waiting)
    const int rc = place ? placed_malloc(stream, bytes, &p) : patient_malloc(bytes, &p, slot == ARENA_BPQ ? 1 : 4);
    if (rc != TOMO_OK) return rc;
    std::lock_guard<std::mutex> lk(g_arena_mu);
    arena_t &a = g_arenas[arena_key{device, stream, slot}];
    if (a.ptr != nullptr) {
        (void)hipFree(p);
        return tomo_fail(TOMO_E_INVALID, "two host threads grew the scratch arena of one (device, stream, slot) at the same time");
    }
    a.ptr = p;
    a.bytes = bytes;
    *out = p;
    return TOMO_OK;
}

extern "C" int tomo_reserve_scratch(int device, size_t bytes, void *stream)
{
    TOMO_REQUIRE(device >= 0 && bytes > 0, "bad scratch reservation");
    TOMO_ON_DEVICE(device);
    void *p = nullptr;
    return tomo_arena_get(device, as_stream(stream), ARENA_TV, bytes, &p, true);
}

extern "C" int tomo_placed_scratch(int device, int slot, size_t bytes, void *stream, void **out_dev)
{
    TOMO_REQUIRE(device >= 0 && slot >= 0 && slot < 8 && bytes > 0 && out_dev != nullptr, "bad placed-scratch request (slot 0 .. 7)");
    TOMO_ON_DEVICE(device);
    return tomo_arena_get(device, as_stream(stream), ARENA_CALLER0 + slot, bytes, out_dev, true);
}

int tomo_arena_release_slot(int device, int slot)
{
    std::lock_guard<std::mutex> lk(g_arena_mu);
    bool any = false;
    for (auto it = g_arenas.begin(); it != g_arenas.end();) {
        if (it->first.device == device && it->first.slot == slot && it->second.ptr) {
            tomo_device_guard g(device);
            if (!any) { TOMO_HIP(hipDeviceSynchronize()); any = true; }
            TOMO_HIP(hipFree(it->second.ptr));
            it = g_arenas.erase(it);
        } else {
            ++it;
        }
    }
    return TOMO_OK;
}

extern "C" int tomo_release_scratch(int device)
{
    tomo_fourier_cache_release(device);
    tomo_fbp_cache_release(device);
    tomo_bp_relay_reset(device);
    std::lock_guard<std::mutex> lk(g_arena_mu);
    bool any = false;
    for (auto it = g_arenas.begin(); it != g_arenas.end();) {
        if (it->first.device == device && it->second.ptr) {
            if (!any) {
                TOMO_ON_DEVICE(device);
                TOMO_HIP(hipDeviceSynchronize());
                any = true;
            }
            tomo_device_guard g(device);
            TOMO_HIP(hipFree(it->second.ptr));
            it = g_arenas.erase(it);
        } else {
            ++it;
        }
    }
    return TOMO_OK;
}

// ---- launch-group timing ------------------------------------------------------------------------
struct prof_rec { hipEvent_t a, b; long long launches; };
static std::mutex g_prof_mu;
static bool g_prof_on = false;
static std::vector<prof_rec> g_prof[PROF_CLASSES];

tomo_prof_scope::tomo_prof_scope(int cls_, hipStream_t st_, int launches_) : cls(cls_), st(st_), launches(launches_), rec(nullptr)
{
    if (!g_prof_on) return;
    prof_rec *r = new prof_rec();
    r->launches = launches;
    if (hipEventCreate(&r->a) != hipSuccess || hipEventCreate(&r->b) != hipSuccess) { delete r; return; }
    (void)hipEventRecord(r->a, st);
    rec = r;
}

tomo_prof_scope::~tomo_prof_scope()
{
    if (!rec) return;
    prof_rec *r = (prof_rec *)rec;
    (void)hipEventRecord(r->b, st);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof[cls].push_back(*r);
    delete r;
}

static void prof_clear()
{
    for (auto &v : g_prof) {
        for (auto &r : v) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
        v.clear();
    }
}

extern "C" int tomo_profile_enable(int on)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    prof_clear();
    g_prof_on = on != 0;
    return TOMO_OK;
}

extern "C" int tomo_profile_read(const char *kernel, long long *launches, double *total_ms)
{
    TOMO_REQUIRE(kernel && launches && total_ms, "NULL argument");
    std::string k(kernel);
    int cls = k == "bp" ? PROF_BP : k == "fp" ? PROF_FP : k == "pdtv" ? PROF_PDTV : k == "roftv" ? PROF_ROFTV : -1;
    TOMO_REQUIRE(cls >= 0, "unknown kernel '%s'", kernel);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    long long n = 0;
    double ms = 0.0;
    for (auto &r : g_prof[cls]) {
        TOMO_HIP(hipEventSynchronize(r.b));
        float t = 0.0f;
        TOMO_HIP(hipEventElapsedTime(&t, r.a, r.b));
        ms += t;
        n += r.launches;
    }
    *launches = n;
    *total_ms = ms;
    return TOMO_OK;
}
