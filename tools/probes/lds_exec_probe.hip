// Micro-benchmark: what does a ds_read_b128 cost when only part of the wave is active, and with gather strides > 1?
//   mode 0: all 64 lanes, unit stride (conflict-free)           mode 1: all lanes, stride 1.41 (FP's worst case)
//   mode 2: 41 % of the lanes active (scattered), stride 1.41    mode 3: 10 % active (scattered), stride 1.41
//   mode 4: lanes 0-15 only                                      mode 5: every 4th lane
// build: hipcc --offload-arch=gfx950 -O3 -o lds_exec_probe lds_exec_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(1024) void probe(float *out, int iters, float a)
{
    __shared__ v4f lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 1024) lds[i] = v4f{a, a, a, a};
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const float stride = (MODE == 0) ? 1.0f : 1.41f;
    int base = (int)(lane * stride) + (threadIdx.x >> 6) * 96;
    bool active = true;
    if (MODE == 2) active = ((lane * 37 + 11) % 100) < 41;
    if (MODE == 3) active = ((lane * 37 + 11) % 100) < 10;
    if (MODE == 4) active = lane < 16;
    if (MODE == 5) active = (lane & 3) == 0;
    v4f acc = v4f{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        if (active) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                v4f s = lds[(base + i * 7) & 4095];
                asm volatile("" : "+v"(s));
                acc += s;
            }
        }
        base = (base + 1) & 2047;
    }
    out[blockIdx.x * 1024 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

template <int MODE>
void run(const char *name)
{
    const int blocks = 256, iters = 20000;
    float *out;
    hipMalloc(&out, sizeof(float) * blocks * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE><<<blocks, 1024>>>(out, 100, 1.0f);
    hipEventRecord(e0);
    probe<MODE><<<blocks, 1024>>>(out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double reads = (double)blocks * 16 /*waves*/ * iters * 8;  // wave-level ds_read_b128 instructions
    printf("%-44s %8.3f ms  %6.2f ns per wave-read per CU  (%.1f clk at 2.0 GHz)\n", name, ms, ms * 1e6 / (reads / 256),
           ms * 1e6 / (reads / 256) * 2.0);
    hipFree(out);
}

int main()
{
    run<0>("all lanes, stride 1");
    run<1>("all lanes, stride 1.41");
    run<2>("41% lanes (scattered), stride 1.41");
    run<3>("10% lanes (scattered), stride 1.41");
    run<4>("lanes 0-15 only");
    run<5>("every 4th lane");
    return 0;
}
