// Ceiling probe (round 4): the issue rate of ds_read_b128 with NO other work in the loop.  tools/probes/lds_exec_probe.hip
// measured 6.0 clk per wave-level ds_read_b128 at unit stride, but its loop carries four v_add_f32 per read (the adds alone
// are 16 SIMD cycles per read and wave: with 4 waves per SIMD that is 64 cycles per 16 reads of a CU = the 4 clk/read the
// LDS would need) -- the 6.0 may be the VALU's.  Here the reads are inline asm into distinct registers, 16 per
// s_waitcnt, nothing else; then the same with the back projector's mix (2 v_pk_fma_f32 per read).
// The guide's figure: 4 clk per wave-instruction = 256 B/clk/CU from 4 waves per CU (MI355X_MICROARCH.md:328,352).
// build: hipcc --offload-arch=gfx950 -O3 -o lds_rate_probe lds_rate_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

#define RD(r, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(r) : "v"(addr))

// MODE 0: reads only; 1: + 2 v_pk_fma_f32 per read, consuming the PREVIOUS batch (so no wait inside a batch);
// MODE 2: + 2 v_pk_fma_f32 + 2 plain VALU (v_fmac_f32) per read = the back projector's real ratio, 4.1 VALU per LDS
//         instruction (SQ_INSTS_VALU 2.90e9 / SQ_INSTS_LDS 7.03e8, profiles/archive/r4b_fp_lane_permutation_ab.txt)
// STRIDE16: lane stride in 16-B slots x 100 (100 = unit stride, 141 = FP's worst case)
#define USE(r) acc0 = __builtin_elementwise_fma(w, v2f{r.x, r.y}, acc0); acc1 = __builtin_elementwise_fma(w, v2f{r.z, r.w}, acc1);
#define USE2(r) acc2 = __builtin_elementwise_fma(w, v2f{r.x, r.y}, acc2); acc3 = __builtin_elementwise_fma(w, v2f{r.z, r.w}, acc3);
#define EXTRA() if (MODE == 2) { sc0 = __builtin_fmaf(sc0, a, sc1); sc1 = __builtin_fmaf(sc1, a, sc0); }
template <int MODE, int STRIDE100, int THREADS>
__global__ __launch_bounds__(THREADS) void probe(float *out, int iters, float a)
{
    extern __shared__ v4f lds[];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = v4f{a, a, a, a};
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned addr = (unsigned)(((lane * STRIDE100) / 100 + wave * 128) & 2047) * 16u;
    v4f r0, r1, r2, r3, r4, r5, r6, r7, r8, r9, r10, r11, r12, r13, r14, r15;
    v2f acc0 = {0, 0}, acc1 = {0, 0}, acc2 = {0, 0}, acc3 = {0, 0};
    const v2f w = {a, a};
    float sc0 = a, sc1 = 2.0f * a;
    r0 = r1 = r2 = r3 = r4 = r5 = r6 = r7 = r8 = r9 = r10 = r11 = r12 = r13 = r14 = r15 = v4f{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        if (MODE >= 1) {
            // batch A in flight while the packed FMAs consume batch B, then the other way round (16 reads per iteration)
            RD(r0, 0); RD(r1, 2048); RD(r2, 4096); RD(r3, 6144); RD(r4, 8192); RD(r5, 10240); RD(r6, 12288); RD(r7, 14336);
            USE(r8) EXTRA() USE2(r9) EXTRA() USE(r10) EXTRA() USE2(r11) EXTRA() USE(r12) EXTRA() USE2(r13) EXTRA() USE(r14) EXTRA() USE2(r15) EXTRA()
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            RD(r8, 16384); RD(r9, 18432); RD(r10, 20480); RD(r11, 22528); RD(r12, 24576); RD(r13, 26624); RD(r14, 28672); RD(r15, 30720);
            USE(r0) EXTRA() USE2(r1) EXTRA() USE(r2) EXTRA() USE2(r3) EXTRA() USE(r4) EXTRA() USE2(r5) EXTRA() USE(r6) EXTRA() USE2(r7) EXTRA()
        } else {
            RD(r0, 0); RD(r1, 2048); RD(r2, 4096); RD(r3, 6144); RD(r4, 8192); RD(r5, 10240); RD(r6, 12288); RD(r7, 14336);
            RD(r8, 16384); RD(r9, 18432); RD(r10, 20480); RD(r11, 22528); RD(r12, 24576); RD(r13, 26624); RD(r14, 28672); RD(r15, 30720);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    v4f s = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + r8 + r9 + r10 + r11 + r12 + r13 + r14 + r15;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y + s.z + s.w + acc0.x + acc0.y + acc1.x + acc1.y + acc2.x + acc2.y + acc3.x + acc3.y + sc0 + sc1;
}

template <int MODE, int STRIDE100, int THREADS>
static void run(const char *name, int blocks_per_cu, double ghz)
{
    const int threads = THREADS;
    const int cus = 256, blocks = cus * blocks_per_cu, iters = 20000;
    float *out;
    hipMalloc(&out, sizeof(float) * blocks * threads);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t shm = 8192 * 16;
    hipFuncSetAttribute((const void *)probe<MODE, STRIDE100, THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    probe<MODE, STRIDE100, THREADS><<<blocks, threads, shm>>>(out, 200, 1.0f);
    hipEventRecord(e0);
    probe<MODE, STRIDE100, THREADS><<<blocks, threads, shm>>>(out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double reads_per_cu = (double)blocks_per_cu * (threads / 64) * iters * 16;  // wave-level ds_read_b128 per CU
    const double ns = ms * 1e6 / reads_per_cu;
    printf("%-34s waves/CU=%2d  %8.3f ms  %5.2f ns per wave-read per CU = %4.2f clk at %.2f GHz  -> %6.1f TB/s chip\n", name,
           blocks_per_cu * threads / 64, ms, ns, ns * ghz, ghz, 1024.0 / ns * 256 / 1000.0);
    hipFree(out);
}

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const double ghz = prop.clockRate / 1e6;
    printf("device: %s  CUs=%d  sclk(max)=%.2f GHz  (clk figures assume the max clock; the sustained clock under LDS load is lower)\n",
           prop.name, prop.multiProcessorCount, ghz);
#define ALL(T) run<0, 100, T>("reads only, stride 1", 1, ghz); run<0, 141, T>("reads only, stride 1.41", 1, ghz); \
               run<1, 100, T>("read + 2 v_pk_fma, stride 1", 1, ghz); run<1, 141, T>("read + 2 v_pk_fma, stride 1.41", 1, ghz); \
               run<2, 100, T>("read + 2 v_pk_fma + 2 VALU (BP mix)", 1, ghz);
    ALL(256) ALL(512) ALL(768) ALL(1024)
    return 0;
}
