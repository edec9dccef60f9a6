// Ceiling probe (round 4): what does a plain streaming copy reach on THIS box?  The MI355X guide quotes 6.29 TB/s for a
// float4 copy (MI355X_MICROARCH.md:35,293); tools/archive/probes/stream_probe.py measured 5.1-5.3 TB/s with the library's diag kernel.
// This probe sweeps everything a copy kernel can choose: 16 B per lane, one element per thread or grid-stride with 1/2/4/8
// independent loads in flight, 256..1024 threads, grid sizes, non-temporal loads / stores, array sizes, plus read-only
// and write-only streams and the runtime's own hipMemcpyAsync / hipMemsetAsync.
// build: hipcc --offload-arch=gfx950 -O3 -o hbm_copy_probe hbm_copy_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float v4f __attribute__((ext_vector_type(4)));

template <bool NT> __device__ __forceinline__ v4f ld(const v4f *p)
{
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
template <bool NT> __device__ __forceinline__ void st(v4f *p, v4f v)
{
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// one float4 per thread, no loop
template <bool NTL, bool NTS>
__global__ void copy_flat(const v4f *__restrict__ in, v4f *__restrict__ out, size_t n4)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) st<NTS>(out + i, ld<NTL>(in + i));
}

// grid-stride, U independent loads in flight per thread; consecutive blocks take consecutive 16 B * blockDim * U chunks
template <int U, bool NTL, bool NTS>
__global__ void copy_gs(const v4f *__restrict__ in, v4f *__restrict__ out, size_t n4)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x * U;
    for (size_t base = (size_t)blockIdx.x * blockDim.x * U + threadIdx.x; base < n4; base += stride) {
        v4f v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = base + (size_t)u * blockDim.x;
            v[u] = i < n4 ? ld<NTL>(in + i) : v4f{0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = base + (size_t)u * blockDim.x;
            if (i < n4) st<NTS>(out + i, v[u]);
        }
    }
}

template <int U, bool NTL>
__global__ void read_gs(const v4f *__restrict__ in, float *__restrict__ out, size_t n4)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x * U;
    v4f acc = v4f{0, 0, 0, 0};
    for (size_t base = (size_t)blockIdx.x * blockDim.x * U + threadIdx.x; base < n4; base += stride) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = base + (size_t)u * blockDim.x;
            if (i < n4) acc += ld<NTL>(in + i);
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

template <int U, bool NTS>
__global__ void write_gs(v4f *__restrict__ out, size_t n4, float a)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x * U;
    for (size_t base = (size_t)blockIdx.x * blockDim.x * U + threadIdx.x; base < n4; base += stride) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = base + (size_t)u * blockDim.x;
            if (i < n4) st<NTS>(out + i, v4f{a, a, a, a});
        }
    }
}

static hipEvent_t e0, e1;
template <typename F>
static double time_ms(F &&f, int reps = 5)
{
    f(); f();
    hipDeviceSynchronize();
    std::vector<float> t;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0);
        f();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

static void report(const char *name, int threads, int grid, double ms, double bytes)
{
    printf("%-34s threads=%4d grid=%8d  %8.3f ms  %7.1f GB/s  (%.3f of 8 TB/s)\n", name, threads, grid, ms, bytes / ms / 1e6,
           bytes / ms / 1e6 / 8000.0);
    fflush(stdout);
}

template <int U, bool NTL, bool NTS>
static void sweep_gs(const char *name, const v4f *in, v4f *out, size_t n4)
{
    double best = 1e30; int bt = 0, bg = 0;
    for (int threads : {256, 512, 1024})
        for (int grid : {256, 512, 1024, 2048, 4096, 8192, 16384, 65536}) {
            const double ms = time_ms([&] { copy_gs<U, NTL, NTS><<<grid, threads>>>(in, out, n4); });
            if (ms < best) { best = ms; bt = threads; bg = grid; }
        }
    report(name, bt, bg, best, 2.0 * n4 * 16);
}

// PD_TV's byte mix: NIN input streams, NOUT output streams, VEC floats per lane (1 = the dword-per-lane rows of the TV
// z-march, 4 = float4), FLAT: one element per thread and a huge grid, else grid-stride from `grid` workgroups
template <int NIN, int NOUT, int VEC, bool FLAT>
__global__ void mix(const float *const *in, float *const *out, size_t n)
{
    typedef float vt __attribute__((ext_vector_type(VEC)));
    const size_t nv = n / VEC;
    const size_t stride = FLAT ? ~(size_t)0 : (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) {
        vt acc = 0;
#pragma unroll
        for (int k = 0; k < NIN; ++k) acc += ((const vt *)in[k])[i];
#pragma unroll
        for (int k = 0; k < NOUT; ++k) ((vt *)out[k])[i] = acc + (float)k;
        if (FLAT) break;
    }
}

template <int NIN, int NOUT, int VEC>
static void run_mix(size_t n, size_t skew)
{
    float *base;
    const size_t per = n * 4 + skew;
    if (hipMalloc(&base, per * (NIN + NOUT) + 4096) != hipSuccess) { printf("alloc failed\n"); return; }
    hipMemset(base, 0, per * (NIN + NOUT));
    const float *hin[NIN]; float *hout[NOUT];
    for (int k = 0; k < NIN; ++k) hin[k] = (const float *)((char *)base + per * k);
    for (int k = 0; k < NOUT; ++k) hout[k] = (float *)((char *)base + per * (NIN + k));
    const float **din; float **dout;
    hipMalloc(&din, sizeof(hin)); hipMalloc(&dout, sizeof(hout));
    hipMemcpy(din, hin, sizeof(hin), hipMemcpyHostToDevice);
    hipMemcpy(dout, hout, sizeof(hout), hipMemcpyHostToDevice);
    const double bytes = 4.0 * n * (NIN + NOUT);
    char name[96];
    for (int threads : {256, 1024}) {
        const int grid = (int)((n / VEC + threads - 1) / threads);
        snprintf(name, sizeof name, "mix %dr/%dw vec%d flat, skew %zu", NIN, NOUT, VEC, skew);
        report(name, threads, grid, time_ms([&] { mix<NIN, NOUT, VEC, true><<<grid, threads>>>(din, dout, n); }, 3), bytes);
    }
    double best = 1e30; int bt = 0, bg = 0;
    for (int threads : {256, 512, 1024})
        for (int grid : {512, 2048, 8192, 32768}) {
            const double ms = time_ms([&] { mix<NIN, NOUT, VEC, false><<<grid, threads>>>(din, dout, n); }, 3);
            if (ms < best) { best = ms; bt = threads; bg = grid; }
        }
    snprintf(name, sizeof name, "mix %dr/%dw vec%d grid-stride, skew %zu", NIN, NOUT, VEC, skew);
    report(name, bt, bg, best, bytes);
    hipFree(base); hipFree(din); hipFree(dout);
}

// PD_TV's ADDRESS STREAM without its arithmetic: the tiling of the shipped K = 3 kernel (pd_zmarch_xk.inl) -- a 256-thread
// workgroup = 2 x 2 waves, a lane = one column, a wave = 8 rows + 3 halo rows either side and 58 + 6 columns, the
// workgroup marches a z-chunk plane by plane; per plane a wave loads 14 rows of U, 13 rows of each of three duals and 12
// rows of the input (dword per lane, rows 4 KB apart) and stores 8 rows of U and of the three duals.  Occupancy is pinned
// with dynamic LDS: 80 KiB per workgroup = two workgroups per CU = two waves per SIMD, like the kernel; 0 = whatever fits.
// What this reaches is the ceiling of the kernel's request stream at its own occupancy.
// MODE bits: 1 = the y-halo rows alias onto the workgroup's own 16 rows (no y over-fetch), 2 = the x-halo lanes alias onto the
// workgroup's own 116 columns, 4 = non-temporal stores, 8 = non-temporal loads
template <int MODE>
__global__ __launch_bounds__(256) void march(const float *const *in, float *const *out, int n, int nz, int gx, int gy, int zchunk,
                                             int tiles_per_xcd)
{
    const int j = (int)blockIdx.x >> 3, xcd = (int)blockIdx.x & 7;
    const int tq = xcd * tiles_per_xcd + (j % tiles_per_xcd), chunk = j / tiles_per_xcd;
    if (tq >= gx * gy) return;
    const int xb = tq % gx, yb = tq / gx;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int x = min(max((xb * 2 + (wave & 1)) * 58 - 3 + lane, 0), n - 1);
    if (MODE & 2) x = min(max(x, xb * 116), min(xb * 116 + 115, n - 1));
    const int y0 = (yb * 2 + (wave >> 1)) * 8;
    const int ylo = (MODE & 1) ? yb * 16 : 0, yhi = (MODE & 1) ? min(yb * 16 + 15, n - 1) : n - 1;
    const int z0 = chunk * zchunk, z1 = min(z0 + zchunk, nz);
    const bool emit = lane >= 3 && lane <= 60;
    auto ld = [](const float *p) { return (MODE & 8) ? __builtin_nontemporal_load(p) : *p; };
    for (int z = max(z0 - 3, 0); z < z1; ++z) {
        const size_t pl = (size_t)z * n * n;
        float acc = 0.0f;
#pragma unroll
        for (int r = 0; r < 14; ++r) {
            const size_t o = pl + (size_t)min(max(y0 + r - 3, ylo), yhi) * n + x;
            acc += ld(in[0] + o);
            if (r < 13) acc += ld(in[1] + o) + ld(in[2] + o) + ld(in[3] + o);
            if (r >= 1 && r < 13) acc += ld(in[4] + o);
        }
        __syncthreads();
        if (emit && z >= z0) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int y = y0 + r;
                if (y < n) {
                    const size_t o = pl + (size_t)y * n + x;
                    if (MODE & 4) {
                        __builtin_nontemporal_store(acc, out[0] + o); __builtin_nontemporal_store(acc, out[1] + o);
                        __builtin_nontemporal_store(acc, out[2] + o); __builtin_nontemporal_store(acc, out[3] + o);
                    } else { out[0][o] = acc; out[1][o] = acc; out[2][o] = acc; out[3][o] = acc; }
                }
            }
        }
    }
}

static void run_march(int n, int nz, size_t skew)
{
    const size_t vox = (size_t)n * n * nz, per = vox * 4 + skew;
    float *base;
    if (hipMalloc(&base, per * 9 + 4096) != hipSuccess) { printf("alloc failed\n"); return; }
    hipMemset(base, 0, per * 9);
    const float *hin[5]; float *hout[4];
    for (int k = 0; k < 5; ++k) hin[k] = (const float *)((char *)base + per * k);
    for (int k = 0; k < 4; ++k) hout[k] = (float *)((char *)base + per * (5 + k));
    const float **din; float **dout;
    hipMalloc(&din, sizeof(hin)); hipMalloc(&dout, sizeof(hout));
    hipMemcpy(din, hin, sizeof(hin), hipMemcpyHostToDevice);
    hipMemcpy(dout, hout, sizeof(hout), hipMemcpyHostToDevice);
    const int gx = ((n + 57) / 58 + 1) / 2, gy = (n + 15) / 16, tiles_per_xcd = (gx * gy + 7) / 8;
    auto go = [&](auto kern, const char *what, int lds_kib, int chunks) {
        const int zchunk = (nz + chunks - 1) / chunks;
        const unsigned blocks = 8u * tiles_per_xcd * chunks;
        const size_t dyn = (size_t)lds_kib * 1024;
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
        const double ms = time_ms([&] { kern<<<blocks, 256, dyn>>>(din, dout, n, nz, gx, gy, zchunk, tiles_per_xcd); }, 3);
        char name[128];
        snprintf(name, sizeof name, "PD stream %s, %d KiB LDS/WG, %d chunks", what, lds_kib, chunks);
        report(name, 256, (int)blocks, ms, 36.0 * vox);
    };
    for (int lds_kib : {80, 40, 0})
        for (int chunks : {32, 16}) go(march<0>, "as the kernel", lds_kib, chunks);
    go(march<1>, "y halo aliased", 80, 32);
    go(march<2>, "x halo aliased", 80, 32);
    go(march<3>, "no over-fetch (x+y aliased)", 80, 32);
    go(march<3>, "no over-fetch (x+y aliased)", 0, 32);
    go(march<4>, "nt stores", 80, 32);
    go(march<8>, "nt loads", 80, 32);
    go(march<12>, "nt loads + stores", 80, 32);
    hipFree(base); hipFree(din); hipFree(dout);
}

int main(int argc, char **argv)
{
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    printf("device: %s  CUs=%d  memclk=%d kHz  buswidth=%d bit  sclk=%d kHz\n", prop.name, prop.multiProcessorCount,
           prop.memoryClockRate, prop.memoryBusWidth, prop.clockRate);
    for (size_t gib : {1, 4}) {
        const size_t bytes = gib << 30, n4 = bytes / 16;
        v4f *in, *out;
        if (hipMalloc(&in, bytes) != hipSuccess || hipMalloc(&out, bytes + (1 << 20)) != hipSuccess) { printf("alloc failed\n"); return 1; }
        hipMemset(in, 1, bytes);
        hipMemset(out, 0, bytes);
        printf("---- arrays of %zu GiB (best (threads, grid) of a sweep per line; median of 5)\n", gib);
        for (int threads : {256, 512, 1024}) {
            const int grid = (int)((n4 + threads - 1) / threads);
            report("flat float4 copy", threads, grid, time_ms([&] { copy_flat<false, false><<<grid, threads>>>(in, out, n4); }), 2.0 * bytes);
            report("flat float4 copy, nt load+store", threads, grid, time_ms([&] { copy_flat<true, true><<<grid, threads>>>(in, out, n4); }), 2.0 * bytes);
        }
        sweep_gs<1, false, false>("grid-stride U=1", in, out, n4);
        sweep_gs<2, false, false>("grid-stride U=2", in, out, n4);
        sweep_gs<4, false, false>("grid-stride U=4", in, out, n4);
        sweep_gs<8, false, false>("grid-stride U=8", in, out, n4);
        sweep_gs<4, true, false>("grid-stride U=4 nt load", in, out, n4);
        sweep_gs<4, false, true>("grid-stride U=4 nt store", in, out, n4);
        sweep_gs<4, true, true>("grid-stride U=4 nt load+store", in, out, n4);
        sweep_gs<8, true, true>("grid-stride U=8 nt load+store", in, out, n4);
        // destination skewed by 68 KiB + 256 B (the TV scratch skew) -- same channels or not
        sweep_gs<4, false, false>("grid-stride U=4, dst + 69888 B", in, (v4f *)((char *)out + 69888), n4);
        {
            double best = 1e30; int bt = 0, bg = 0;
            float *sink; hipMalloc(&sink, 64);
            for (int threads : {256, 512, 1024})
                for (int grid : {1024, 2048, 4096, 8192, 16384}) {
                    const double ms = time_ms([&] { read_gs<4, false><<<grid, threads>>>(in, sink, n4); });
                    if (ms < best) { best = ms; bt = threads; bg = grid; }
                }
            report("read only U=4", bt, bg, best, 1.0 * bytes);
            best = 1e30;
            for (int threads : {256, 512, 1024})
                for (int grid : {1024, 2048, 4096, 8192, 16384}) {
                    const double ms = time_ms([&] { write_gs<4, false><<<grid, threads>>>(out, n4, 2.0f); });
                    if (ms < best) { best = ms; bt = threads; bg = grid; }
                }
            report("write only U=4", bt, bg, best, 1.0 * bytes);
            best = 1e30;
            for (int threads : {256, 512, 1024})
                for (int grid : {1024, 2048, 4096, 8192, 16384}) {
                    const double ms = time_ms([&] { write_gs<4, true><<<grid, threads>>>(out, n4, 2.0f); });
                    if (ms < best) { best = ms; bt = threads; bg = grid; }
                }
            report("write only U=4 nt", bt, bg, best, 1.0 * bytes);
            hipFree(sink);
        }
        report("hipMemcpyAsync D2D", 0, 0, time_ms([&] { hipMemcpyAsync(out, in, bytes, hipMemcpyDeviceToDevice, 0); }), 2.0 * bytes);
        report("hipMemsetAsync", 0, 0, time_ms([&] { hipMemsetAsync(out, 0, bytes, 0); }), 1.0 * bytes);
        hipFree(in); hipFree(out);
    }
    printf("---- PD_TV's stream mix on 1 GiB arrays: 5 reads + 4 writes (a middle launch), 2 + 1 (first / last launch shape)\n");
    const size_t n = (size_t)1 << 28;
    run_mix<5, 4, 4>(n, 0);
    run_mix<5, 4, 4>(n, 69888);
    run_mix<5, 4, 1>(n, 0);
    run_mix<5, 4, 1>(n, 69888);
    run_mix<2, 1, 4>(n, 69888);
    run_mix<2, 1, 1>(n, 69888);
    printf("---- PD_TV's address stream (tiling, row strides, halo re-reads) without its arithmetic, 1024^3; GB/s = COMPULSORY 36 B/voxel / time\n");
    run_march(1024, 1024, 69888);
    return 0;
}
