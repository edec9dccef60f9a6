// Micro-benchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 (and ds_read_b128 beside them) on gfx950.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_probe valu_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void probe(float *out, int iters, float a, float b)
{
    __shared__ v4f lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = v4f{a, b, a, b};
    __syncthreads();
    v2f acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = v2f{(float)i, (float)threadIdx.x};
    const v2f w = v2f{a, a};
    int idx = threadIdx.x & 63;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {  // 32 scalar fma
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                acc[i].x = __builtin_fmaf(a, acc[i].x, b);
                acc[i].y = __builtin_fmaf(a, acc[i].y, b);
            }
            asm volatile("" ::: "memory");
        } else if (MODE == 1) {  // 16 packed fma (same flops as MODE 0)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_elementwise_fma(w, acc[i], v2f{b, b});
            asm volatile("" ::: "memory");
        } else if (MODE == 2) {  // 8 ds_read_b128 + 16 packed fma (BP-like mix per half angle)
            v4f s[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) s[i] = lds[(idx + i * 40) & 1023];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc[2 * i] = __builtin_elementwise_fma(w, s[i].lo, acc[2 * i]);
                acc[2 * i + 1] = __builtin_elementwise_fma(w, s[i].hi, acc[2 * i + 1]);
            }
            idx = (idx + 1) & 63;
        } else {  // 8 ds_read_b128 only
            v4f s[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) s[i] = lds[(idx + i * 40) & 1023];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] += s[i].lo;
            idx = (idx + 1) & 63;
        }
    }
    float r = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += acc[i].x + acc[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int MODE>
void run(const char *name, int blocks_per_cu, double flop_per_iter_lane, double ldsB_per_iter_lane)
{
    const int blocks = 256 * blocks_per_cu, iters = 20000;
    float *out;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE><<<blocks, 256>>>(out, 100, 0.999f, 0.001f);
    hipEventRecord(e0);
    probe<MODE><<<blocks, 256>>>(out, iters, 0.999f, 0.001f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double lanes = (double)blocks * 256;
    printf("%-28s waves/SIMD=%d  %8.3f ms  %7.1f TFLOP/s  %7.1f TB/s LDS\n", name, blocks_per_cu, ms,
           flop_per_iter_lane * lanes * iters / ms / 1e9, ldsB_per_iter_lane * lanes * iters / ms / 1e9);
    hipFree(out);
}

int main()
{
    for (int w : {1, 2, 4}) {
        run<0>("32 x v_fma_f32", w, 64, 0);
        run<1>("16 x v_pk_fma_f32", w, 64, 0);
        run<2>("8 ds_read_b128 + 16 pk_fma", w, 64, 128);
        run<3>("8 ds_read_b128 + 8 pk_add", w, 0, 128);
    }
    return 0;
}
