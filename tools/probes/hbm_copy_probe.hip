// Ceiling probe (round 4): what does a plain streaming copy reach on THIS box?  The MI355X guide quotes 6.29 TB/s for a
// float4 copy (MI355X_MICROARCH.md:35,293); tools/stream_probe.py measured 5.1-5.3 TB/s with the library's diag kernel.
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
    return 0;
}
