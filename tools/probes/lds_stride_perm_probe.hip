// Probe (round 5): can a per-angle lane -> pixel permutation take the bank conflicts out of the forward projector's LDS gather?
// The FP kernel's lane samples the staged row at slot floor(x0 + s * pixel), s = 1/|cos| in [1, 1.41]; with pixel = lane the
// 16 lanes of a ds_read_b128 service group span up to 22 slots of the 16-slot bank row (conflict factor 1.46 on counters).
// A thread may own a DIFFERENT pixel per angle (its accumulators are independent), so pixel = (m * thread) mod 1024 with an odd
// m chosen per angle costs nothing inside the march.  This probe times the two-tap read pair for (s, m) with the march's
// address arithmetic (x += slope per row), 1024 threads per workgroup, one workgroup per CU, as the whole-row form runs.
// build: hipcc --offload-arch=gfx950 -O3 -o lds_stride_perm_probe lds_stride_perm_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float v4f __attribute__((ext_vector_type(4)));

// the FP kernel's lane -> logical pixel map: the 16 lanes of each ds_read_b128 service group ({0-3,12-15,20-27}, ...) get 16
// consecutive logical pixels (fp_tiled.inl: fp_lane_pixel)
__device__ __forceinline__ int lane_pixel(int lane)
{
    const int q = (lane >> 2) & 7;
    const int odd = (q ^ (q >> 1) ^ (q >> 2)) & 1;
    return (lane & 32) | (odd << 4) | ((q >> 1) << 2) | (lane & 3);
}

__global__ __launch_bounds__(1024) void probe(float *out, int iters, float s, int m, float slope, float x00)
{
    extern __shared__ v4f lds[];   // 4096 slots = 64 KB
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = v4f{1.0f, 2.0f, 3.0f, 4.0f};
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int pixel = (m * ((int)threadIdx.x - lane + lane_pixel(lane))) & 1023;
    // 8 angles of (nearly) the same slope per thread, as the kernel's angle group: 16 reads in flight per wait
    float x[8];
    v4f acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { x[i] = x00 + 1.7f * (float)i + s * (float)pixel; acc[i] = v4f{0, 0, 0, 0}; }
    for (int it = 0; it < iters; ++it) {
        v4f t0[8], t1[8];
        float w[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float fl = floorf(x[i]);
            w[i] = x[i] - fl;
            const unsigned addr = ((unsigned)(int)fl & 2047u) * 16u;
            asm volatile("ds_read_b128 %0, %1" : "=v"(t0[i]) : "v"(addr));
            asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(t1[i]) : "v"(addr));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            acc[i] += (1.0f - w[i]) * t0[i] + w[i] * t1[i];
            x[i] += slope;
        }
        if ((it & 7) == 7) {
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] -= 8.0f * slope - 0.37f;   // stay inside the staged window, new fractional phase
        }
        if ((it & 255) == 255) {
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = x00 + 1.7f * (float)i + s * (float)pixel;
        }
    }
    v4f a4 = acc[0] + acc[1] + acc[2] + acc[3] + acc[4] + acc[5] + acc[6] + acc[7];
    float4 dummy; (void)dummy;
    out[blockIdx.x * blockDim.x + threadIdx.x] = a4.x + a4.y + a4.z + a4.w;
}

int main()
{
    const int blocks = 256, threads = 1024, iters = 4096;
    float *out;
    hipMalloc(&out, sizeof(float) * blocks * threads);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t shm = 4096 * 16;
    hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    const int ms_list[] = {1, 3, 5, 7, 9, 11, 13, 15, 17, 19, 21, 23, 25, 27, 29, 31, 33, 37, 41, 45, 49, 53, 57, 61};
    const int nm = sizeof(ms_list) / sizeof(int);
    printf("# ns per wave-level read PAIR per CU (1024 threads, 1 workgroup per CU); rows: theta (deg), s = 1/cos; columns: multiplier m\n");
    printf("# theta    s   ");
    for (int j = 0; j < nm; ++j) printf(" m=%-3d", ms_list[j]);
    printf("  best  gain\n");
    double sum_base = 0, sum_best = 0;
    for (int deg2 = 0; deg2 <= 90; deg2 += 3) {   // theta = 0 .. 45 degrees in steps of 1.5
        const double th = deg2 * 0.5 * M_PI / 180.0;
        const float s = (float)(1.0 / cos(th)), slope = (float)tan(th);
        printf("  %5.1f %6.4f", deg2 * 0.5, s);
        double base = 0, best = 1e30; int bm = 1;
        for (int j = 0; j < nm; ++j) {
            probe<<<blocks, threads, shm>>>(out, 50, s, ms_list[j], slope, 3.3f);
            hipEventRecord(e0);
            probe<<<blocks, threads, shm>>>(out, iters, s, ms_list[j], slope, 3.3f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double ns = ms * 1e6 / ((double)(threads / 64) * iters * 8);   // 8 read pairs per iteration
            printf(" %5.2f", ns);
            if (j == 0) base = ns;
            if (ns < best) { best = ns; bm = ms_list[j]; }
        }
        printf("  m=%-3d %4.2f\n", bm, base / best);
        sum_base += base; sum_best += best;
    }
    printf("# mean over theta uniform in [0, 45] deg: m = 1: %.3f ns, best m per angle: %.3f ns  -> %.3f x\n", sum_base / 31, sum_best / 31, sum_base / sum_best);
    hipFree(out);
    return 0;
}
