// Forward projector, "tiled" variant.  Included inside the anonymous namespace of proj_kernels.hip.
//
// A ray-driven Joseph projector re-reads the whole volume once per angle; done naively that is V*Na_s*4 bytes of
// L2/Infinity-Cache traffic per call (322 GB for 1024^3 x 75 angles).  Here a workgroup owns 256 detector pixels x
// FP_A angles of ONE stepping class (all of them interpolate along the same, contiguous, in-plane axis) x 4 slices and
// marches the stepping axis.  For every row of the march the part of that volume row which the 256 x FP_A rays can
// touch is staged ONCE in LDS (as float4 over the 4 slices, zero outside the volume) and then sampled by all FP_A
// angles: ~1 byte of staging traffic per ray-step instead of ~6.  Staging is software-pipelined: the global loads of
// chunk c+1 are in flight (held in registers) while chunk c is sampled from the other LDS buffer; one barrier per
// chunk.  Each (pixel, angle, slice) accumulator lives in a register and is summed in march order, so the result is
// bit-identical to the sequential oracle.
constexpr int FP_A = 8;               // angles per workgroup
constexpr int FP_MAX_WPITCH = 1280;   // 5 passes of 256 columns

struct FpTiledArgs {
    const float *src;        // volume with the interpolation axis contiguous ([nz][n][n])
    const tomo_angle_t *tab; // subset table
    const int *order;        // subset-local angle indices of this stepping class
    const int *mult;         // per entry of `order`: odd lane -> pixel multiplier of the whole-row form (null: 1), see fp_lane_mult
    int n_class;             // angles in the class
    int nz, n, nu, na, na_full;
    float *out;
    const float *b, *w;
    const float *ring;       // Group-Huber offsets r_x [nz][nu] added to the residual (null: none)
    float ring_scale;        // ringGH_accelerate
    int fidelity, gathered;
    int robust;              // TOMO_ROBUST_* re-weighting of the LS / PWLS residual
    float rdelta;
    int zquad;               // residual epilogue: out is [ceil(nz/4)][na][nu][4] (TOMO_RESIDUAL_ZQUAD, see tomo_mi355x.h)
    int wpitch;              // LDS pitch (float4 units) per staged row, <= 256 * passes <= 1024
    int nut, ngroups, nzb;   // detector tiles, angle groups, slice quads
    int bt;                  // whole-row form: detector pixels per tile = threads launched (a multiple of 64, <= 1024)
#if TOMO_DEV
    int probe;               // measurement only (tools/archive/probes/fp_stage_probe.py): 16 = skip the staging (global loads + LDS writes), 32 = the LDS writes only
#else
    static constexpr int probe = 0;  // the shipped flavour carries no measurement switches
#endif
};

// BT = workgroup size = detector pixels per workgroup.  256: several workgroups per CU.  1024: the workgroup spans the
// whole detector row of a 1024-wide problem, so every volume row is staged once per angle group instead of once per
// (angle group, detector tile) with overlapping windows -- the 12-strided angles of an ordered subset spread a 256-pixel
// tile's window to ~480 columns, i.e. 4 x 480 instead of 1030 per row; measured L2->fabric traffic 55 GB per call with
// 256-pixel tiles (profiles/archive/r1_bp_fp_pmc.txt) because the 12-20 workgroups sharing a slice quad drift apart.
// PASSES = ceil(wpitch / BT) column passes per staged row; KC = M / PASSES rows per chunk (compile-time so that the
// staging index arithmetic is free of integer divisions).
// M = float4 staging items per thread and chunk (register prefetch depth).  DB: double-buffered tile, one barrier per
// chunk (narrow windows); !DB: one tile, two barriers per chunk -- half the LDS, so wide windows (the 12-strided
// angles of an ordered subset) still get several rows per chunk and 3 workgroups per CU.
// Pixel (0..63) inside the wave's 64-pixel segment that lane `lane` samples: the 16 lanes of each ds_read_b128 service group
// get 16 consecutive pixels.  With q = (lane >> 2) & 7 the groups of one half-wave are the quads {0, 3, 5, 6} (even
// number of set bits in q) and {1, 2, 4, 7} (odd); inside a group the quads rank in ascending order, i.e. q >> 1.
__device__ __forceinline__ int fp_lane_pixel(int lane)
{
    const int q = (lane >> 2) & 7;
    const int odd = (q ^ (q >> 1) ^ (q >> 2)) & 1;
    return (lane & 32) | (odd << 4) | ((q >> 1) << 2) | (lane & 3);
}

// A = angles per workgroup: 8 (FP_A) everywhere except the dense-angle form of round 4 (256 pixels x 16 angles: half the
// stagings per sample for angle sets whose neighbours in the slope order are a fraction of a degree apart, BASELINE configs[3])
template <bool LERP8, bool RESID, int PASSES, int M, bool DB, int BT, int A = 8>
__global__ __launch_bounds__(BT) void fp_tiled_kernel(FpTiledArgs a)
{
    constexpr int KC = M / PASSES;
    extern __shared__ __attribute__((aligned(16))) unsigned char fp_smem[];
    const int tile_items = KC * a.wpitch;
    float4 *tile0 = reinterpret_cast<float4 *>(fp_smem);
    float4 *tile1 = DB ? tile0 + tile_items : tile0;
    int *win_lo = reinterpret_cast<int *>(tile1 + tile_items);  // [n]
    int *win_wid = win_lo + a.n;                                // [n]

    // XCD-aware numbering: one slice-quad stream per XCD so that its L2 keeps that quad's rows
    const int per_zb = a.nut * a.ngroups;
    const int q = (int)blockIdx.x >> 3, xcd = (int)blockIdx.x & 7;
    const int per_xcd = (a.nzb * per_zb + 7) >> 3;  // a contiguous eighth of the (slice quad, tile, group) list per XCD
    const int wi = xcd * per_xcd + q;
    if (wi >= a.nzb * per_zb) return;
    const int zb = wi / per_zb;
    const int rest = wi % per_zb;
    const int ut = rest % a.nut, g = rest / a.nut;
    const int z0 = zb * 4;
    // the 1024-thread form is launched with a.bt <= 1024 threads: a 2560-wide detector runs 3 tiles of 896 pixels instead
    // of 2.5 tiles of 1024 (one sixth of the lanes idle)
    const int bt = (BT == 1024) ? a.bt : BT;
    const int u0 = ut * bt;
    const int tid = (int)threadIdx.x;
    // Which detector pixel a lane samples (round 4).  The LDS serves a wave-level ds_read_b128 in four groups of 16 lanes,
    // {0-3,12-15,20-27}, {4-11,16-19,28-31} and the same + 32 (MI355X_MICROARCH.md, LDS table), one cycle per group when its
    // 16 slots fall into 16 different banks.  Neighbouring pixels sample the staged row 1/|cos| in [1, 1.41] slots apart:
    // with pixel = lane a group's lanes span 28 pixels = up to 39 slots of a 16-slot bank row (three-way conflicts,
    // tools/probes/lds_rate_probe.hip: 12.1 clk per read at stride 1.41 against 4.0 at stride 1); with the 16 lanes of a
    // group on 16 CONSECUTIVE pixels the span is at most 22 slots, two-way at worst.  Model over the 900-angle set:
    // 2.05 -> 1.69 LDS cycles per group access.  The staging keeps its lane-linear columns (tid); only the ray a lane
    // owns -- and hence the 4-byte column it stores in the sinogram row -- moves inside the wave's 64-pixel segment.
    const int lane = tid & 63;
#ifdef TOMO_FP_NO_LANE_PERM   // A/B builds only (tools/run_ab.sh): pixel = lane, the round-3 mapping
    const int lane_pix = lane;
#else
    const int lane_pix = fp_lane_pixel(lane);
#endif
    const int t_log = (tid - lane) + lane_pix;   // the thread's logical position in the detector tile
    const int iu = u0 + t_log;                   // ... and the pixel it STORES (epilogue)
    const int n = a.n;
    const int ng = min(A, a.n_class - g * A);   // angles in this group (uniform)
    const int *ord = a.order + g * A;
    // Per-angle lane -> pixel permutation (round 6, whole-row form only): for angle i the thread marches pixel
    // (m_i * t_log) mod bt, m_i odd and coprime with bt, chosen per angle so that the 16 slots a ds_read_b128 service group
    // samples -- m_i / |cos| apart instead of 1 / |cos| -- fall into as many different bank rows as possible
    // (fp_lane_mult; docs/kernels/fp.md).  Free inside the march: a thread's accumulators are independent per angle.  The
    // epilogue hands every value back to the thread that stores it through LDS.
    const bool perm = (BT == 1024) && a.mult != nullptr;   // uniform
    const int *mul = perm ? a.mult + g * A : nullptr;

    const float half_n = 0.5f * (float)n - 0.5f, half_u = 0.5f * (float)a.nu - 0.5f, nf = (float)n;
    float offs[A], slope[A], acc[A][4];
#pragma unroll
    for (int i = 0; i < A; ++i) {
        const tomo_angle_t t = a.tab[ord[i < ng ? i : 0]];
        const int iu_m = perm ? u0 + (mul[i < ng ? i : 0] * t_log) % bt : iu;   // the pixel this thread MARCHES for angle i
        const float s = ((float)iu_m - half_u) + t.cor;
        offs[i] = fmaf(s, t.inv, half_n);
        slope[i] = t.slope;
#pragma unroll
        for (int zz = 0; zz < 4; ++zz) acc[i][zz] = 0.0f;
    }

    // ---- window of every march row (the sampling coordinate is monotone in the detector index, so the two ends of
    //      the detector tile bound it); two zero columns on either side (-2,-1 / n,n+1) absorb rays that miss the volume
    for (int k = tid; k < n; k += bt) {
        const float kw = (float)k - half_n;
        float fmin = 3.0e38f, fmax = -3.0e38f;
        for (int i = 0; i < ng; ++i) {
            const tomo_angle_t t = a.tab[ord[i]];
            const float o0 = fmaf(((float)u0 - half_u) + t.cor, t.inv, half_n);
            const float o1 = fmaf(((float)(u0 + bt - 1) - half_u) + t.cor, t.inv, half_n);
            const float f0 = fmaf(kw, t.slope, o0), f1 = fmaf(kw, t.slope, o1);
            fmin = fminf(fmin, fminf(f0, f1));
            fmax = fmaxf(fmax, fmaxf(f0, f1));
        }
        const int lo = (int)fminf(fmaxf(floorf(fmin), -2.0f), (float)n);
        const int hi = (int)fminf(fmaxf(floorf(fmax) + 1.0f, -1.0f), (float)(n + 1));
        win_lo[k] = lo;
        win_wid[k] = max(hi - lo + 1, 2);
    }
    __syncthreads();

    const size_t zstride = (size_t)n * n;
    // slices beyond nz read slice nz-1 (valid memory) through a zero-length descriptor, i.e. as zeros
    const float *p0 = a.src + (size_t)z0 * zstride;
    const float *p1 = a.src + (size_t)min(z0 + 1, a.nz - 1) * zstride;
    const float *p2 = a.src + (size_t)min(z0 + 2, a.nz - 1) * zstride;
    const float *p3 = a.src + (size_t)min(z0 + 3, a.nz - 1) * zstride;
    const int rowbytes = n * 4;
    const int nb1 = z0 + 1 < a.nz ? rowbytes : 0, nb2 = z0 + 2 < a.nz ? rowbytes : 0, nb3 = z0 + 3 < a.nz ? rowbytes : 0;

    float4 pre[M];
    // Gather of the chunk starting at row k0 into registers: pre[r * PASSES + p] <- (row k0+r, column tid + BT p).
    // One buffer descriptor per (row, slice) -- base = the row, num_records = n floats, built by scalar instructions --
    // lets the hardware's range check do what used to be 12 VALU per item: a column left of the volume (negative offset =
    // huge unsigned), right of it, beyond the window (offset forced out of range) or a row past the march loads 0.
    auto prefetch = [&](int k0) {
#pragma unroll
        for (int r = 0; r < KC; ++r) {
            // readfirstlane: the row index is wave-uniform, but in the 16-angle instantiation the compiler carries the chunk
            // counter in a VECTOR register, builds the four row descriptors there and wraps every one of the 16 staging loads of a
            // chunk in a waterfall loop (4 v_readfirstlane + 2 v_cmp_eq_u64 + exec juggling per load: 3.4 extra VALU per read pair
            // in a kernel bound by VALU issue; found in the ISA in round 6 -- tools/kisa.sh + kmix.py: 549 -> 4xx VALU per chunk)
            const int k = __builtin_amdgcn_readfirstlane(min(k0 + r, n - 1));
            const int lo4 = __builtin_amdgcn_readfirstlane(win_lo[k]) * 4;
            const int wid = (k0 + r < n) ? __builtin_amdgcn_readfirstlane(win_wid[k]) : 0;
            const size_t rowoff = (size_t)k * (size_t)n;
            const __amdgpu_buffer_rsrc_t d0 = __builtin_amdgcn_make_buffer_rsrc((void *)(p0 + rowoff), 0, rowbytes, 0x00020000);
            const __amdgpu_buffer_rsrc_t d1 = __builtin_amdgcn_make_buffer_rsrc((void *)(p1 + rowoff), 0, nb1, 0x00020000);
            const __amdgpu_buffer_rsrc_t d2 = __builtin_amdgcn_make_buffer_rsrc((void *)(p2 + rowoff), 0, nb2, 0x00020000);
            const __amdgpu_buffer_rsrc_t d3 = __builtin_amdgcn_make_buffer_rsrc((void *)(p3 + rowoff), 0, nb3, 0x00020000);
#pragma unroll
            for (int p = 0; p < PASSES; ++p) {
                const int j = tid + bt * p;
                const int off = (j < wid) ? (j << 2) + lo4 : (int)0x80000000;
                float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                // a wave whose 64 columns all lie beyond the window skips the pass (wave-uniform: scalar branch); with
                // BT = 1024 the second pass of a 1030-column row keeps one wave of sixteen busy
                if (p == 0 || __builtin_amdgcn_readfirstlane(j - (tid & 63)) < wid) {
                    v.x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(d0, off, 0, 0));
                    v.y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(d1, off, 0, 0));
                    v.z = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(d2, off, 0, 0));
                    v.w = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(d3, off, 0, 0));
                }
                pre[r * PASSES + p] = v;
            }
        }
    };

    const int nchunks = (n + KC - 1) / KC;
    const bool stage = !(a.probe & 16);  // uniform; false only in the staging-cost measurement (results are garbage then)
    if (!stage) {
#pragma unroll
        for (int i = 0; i < M; ++i) pre[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    if (stage) prefetch(0);
    for (int c = 0; c < nchunks; ++c) {
        float4 *tile = (c & 1) ? tile1 : tile0;
        if (!DB) __syncthreads();  // every wave is past the sampling of chunk c-1 (same buffer)
        if (stage && !(a.probe & 32)) {
#pragma unroll
            for (int r = 0; r < KC; ++r)
#pragma unroll
                for (int p = 0; p < PASSES; ++p) {
                    const int j = tid + bt * p;
                    if (p == 0 || __builtin_amdgcn_readfirstlane(j - (tid & 63)) < a.wpitch)
                        if (j < a.wpitch) tile[r * a.wpitch + j] = pre[r * PASSES + p];
                }
        } else if (stage) {  // measurement: the global loads are kept alive, the LDS writes are skipped
#pragma unroll
            for (int i = 0; i < M; ++i) asm volatile("" ::"v"(pre[i].x), "v"(pre[i].y), "v"(pre[i].z), "v"(pre[i].w));
        }
        __syncthreads();  // chunk c staged; every wave is past the sampling of chunk c-1 (other buffer)
        // readfirstlane: (float)row below is a VALU conversion, and from it some instantiations pull the whole chain of wave-uniform
        // row / chunk counters into vector registers (then every use of them costs v_min / v_add / v_readfirstlane instead of SALU)
        const int k0 = __builtin_amdgcn_readfirstlane(c * KC);
        if (stage && c + 1 < nchunks) prefetch(k0 + KC);  // in flight while chunk c is sampled
#pragma unroll
        for (int r = 0; r < KC; ++r) {
            // rows past the end of the march are staged as zeros and sampled with the last row's (valid) coordinates,
            // so they add exactly nothing and no tail branch is needed
            const int kr = min(k0 + r, n - 1);
            const float kw = (float)kr - half_n;
            // the row/window origin is wave-uniform: keep it in a scalar so that a tap address is one v_lshl_add
            int rb = __builtin_amdgcn_readfirstlane((r * a.wpitch - win_lo[kr]) * 16);
            asm("" : "+s"(rb));  // opaque: otherwise the *16 is factored back out and costs a second VALU op per tap
            const char *trow = reinterpret_cast<const char *>(tile);
#pragma unroll
            for (int i = 0; i < A; ++i) {  // slots >= ng repeat angle 0 (never stored)
                const float f = fmaf(kw, slope[i], offs[i]);
                const float fl = floorf(f);
                const float w = lerp_w<LERP8>(f, fl), omw = 1.0f - w;
                // rays outside [-2, n] sample the zero columns: one v_med3_f32, then one v_lshl_add for the address
                const int idx = (int)__builtin_amdgcn_fmed3f(fl, -2.0f, nf);
                const float4 *tap = reinterpret_cast<const float4 *>(trow + ((idx << 4) + rb));
                const float4 s0 = tap[0], s1 = tap[1];
                acc[i][0] = fmaf(omw, s0.x, acc[i][0]); acc[i][0] = fmaf(w, s1.x, acc[i][0]);
                acc[i][1] = fmaf(omw, s0.y, acc[i][1]); acc[i][1] = fmaf(w, s1.y, acc[i][1]);
                acc[i][2] = fmaf(omw, s0.z, acc[i][2]); acc[i][2] = fmaf(w, s1.z, acc[i][2]);
                acc[i][3] = fmaf(omw, s0.w, acc[i][3]); acc[i][3] = fmaf(w, s1.w, acc[i][3]);
            }
        }
    }
    if (BT == 1024 && perm) {
        // un-permute: angle by angle through two LDS rows of bt float4 (the tiles are free now), one barrier per angle;
        // afterwards acc[i] holds the four slices of the pixel this thread stores and the epilogue below is the usual one
        float4 *stage = tile0;
        __syncthreads();   // every wave is past its last tile read
#pragma unroll
        for (int i = 0; i < A; ++i) {
            float4 *row = stage + (i & 1) * bt;
            row[(mul[i < ng ? i : 0] * t_log) % bt] = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
            __syncthreads();
            const float4 v = row[t_log];
            acc[i][0] = v.x; acc[i][1] = v.y; acc[i][2] = v.z; acc[i][3] = v.w;
        }
    }
    if (iu >= a.nu) return;
    // (guards instead of `break`s: with 16 angles the unroller gives up on an early-exit loop and the accumulator array,
    // indexed by a run-time `i`, lands in scratch memory)
#pragma unroll
    for (int i = 0; i < A; ++i) {
        if (i < ng) {
            const int k_a = ord[i];
            const tomo_angle_t t = a.tab[k_a];
            float v4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int zz = 0; zz < 4; ++zz) {
                const int z = z0 + zz;
                if (z < a.nz) {
                    float val = acc[i][zz] * t.scale;
                    if (RESID) val = fp_residual_value(a, val, z, k_a, t.src, iu);
                    v4[zz] = val;
                    if (!(RESID && a.zquad)) a.out[((size_t)z * a.na + k_a) * a.nu + iu] = val;
                }
            }
            // private residual layout (TOMO_RESIDUAL_ZQUAD): the four slices of this workgroup's quad as one 16-byte word
            if (RESID && a.zquad)
                reinterpret_cast<float4 *>(a.out)[((size_t)zb * a.na + k_a) * a.nu + iu] = make_float4(v4[0], v4[1], v4[2], v4[3]);
        }
    }
}

// ---- per-angle lane -> pixel multiplier of the whole-row form (host).  Bank model of one ds_read_b128 service group: its 16
// lanes hold 16 consecutive logical positions t, i.e. pixels m*t mod bt, i.e. LDS slots floor(x0 + s*pixel) with s = 1/|cos|
// (|inv| of the angle record) in [1, 1.4143]; the LDS serves the group in as many cycles as the fullest of its 16 bank rows
// (slot mod 16) holds DIFFERENT slots.  m = 1 spans up to 22 slots (two-way conflicts: 1.5-2.0 cycles at 20-45 degrees), the
// best odd m < 64 per angle 1.5-1.7.  Measured on the kernel's own loop: tools/probes/fp_combo_probe.hip,
// profiles/r6_fp_combo_probe.txt (sampling loop x 1.09 alone, x 1.215 together with 8 slices per thread).
static double fp_bank_model(double s, int m, int bt)
{
    double total = 0.0;
    int cnt = 0;
    for (int ph = 0; ph < 8; ++ph) {
        const double x0 = 3.0 + ph * 0.91;
        for (int t0 = 0; t0 + 16 <= bt; t0 += 16 * 5) {   // every fifth service group
            int distinct[16], nd = 0, rows[16] = {0};
            for (int j = 0; j < 16; ++j) {
                const int slot = (int)std::floor(x0 + s * (double)((m * (t0 + j)) % bt));
                bool seen = false;
                for (int q = 0; q < nd; ++q) seen |= distinct[q] == slot;   // same slot: one broadcast
                if (!seen) { distinct[nd++] = slot; ++rows[slot & 15]; }
            }
            int worst = 1;
            for (int r = 0; r < 16; ++r) worst = std::max(worst, rows[r]);
            total += worst;
            ++cnt;
        }
    }
    return cnt ? total / cnt : 1.0;
}

// multiplier for stride s and tile width bt, memoised on a 1/512 grid of s (the model is smooth at that scale and a
// context asks for up to a few thousand angles)
static int fp_lane_mult(double s, int bt)
{
    static std::mutex mu;
    static std::map<std::pair<int, int>, int> memo;
    const int key = (int)std::lround((std::min(std::max(s, 1.0), 1.4143) - 1.0) * 512.0);
    std::lock_guard<std::mutex> lock(mu);
    auto it = memo.find({bt, key});
    if (it != memo.end()) return it->second;
    const double sq = 1.0 + key / 512.0;
    int best_m = 1;
    double best = fp_bank_model(sq, 1, bt);
    for (int m = 3; m < 64; m += 2) {
        if (std::gcd(m, bt) != 1) continue;   // m*t mod bt must be a bijection (bt = 896 = 2^7 * 7 rules out 7, 21, ...)
        const double c = fp_bank_model(sq, m, bt);
        if (c < best - 1e-3) { best = c; best_m = m; }
    }
    memo[{bt, key}] = best_m;
    return best_m;
}

// Upper bound (host, same float arithmetic as the kernel) of the staged window width over all groups / tiles / rows.
// The width is a max of affine functions of the row index minus a min of affine functions, hence convex: its maximum
// over the march is attained at the first or the last row.
static int fp_window_bound(const tomo_angle_t *tab, const int *order, int n_class, int n, int nu, int tile = 256,
                           int group = FP_A)
{
    const float half_n = 0.5f * (float)n - 0.5f, half_u = 0.5f * (float)nu - 0.5f;
    int bound = 2;
    const int nut = ceil_div(nu, tile);
    for (int g = 0; g * group < n_class; ++g) {
        const int ng = std::min(group, n_class - g * group);
        for (int ut = 0; ut < nut; ++ut) {
            for (int e = 0; e < 2; ++e) {
                const float kw = (float)(e ? n - 1 : 0) - half_n;
                float fmin = 3.0e38f, fmax = -3.0e38f;
                for (int i = 0; i < ng; ++i) {
                    const tomo_angle_t &t = tab[order[g * group + i]];
                    const float o0 = std::fmaf(((float)(ut * tile) - half_u) + t.cor, t.inv, half_n);
                    const float o1 = std::fmaf(((float)(ut * tile + tile - 1) - half_u) + t.cor, t.inv, half_n);
                    const float f0 = std::fmaf(kw, t.slope, o0), f1 = std::fmaf(kw, t.slope, o1);
                    fmin = std::min(fmin, std::min(f0, f1));
                    fmax = std::max(fmax, std::max(f0, f1));
                }
                // UNCLIPPED width: only that is convex in the row index (clipping to the volume can make the end rows
                // narrow while a middle row, fully inside the volume, is wide); the clip is applied once at the end
                const double wdt = (double)std::floor(fmax) + 1.0 - (double)std::floor(fmin) + 1.0;
                bound = std::max(bound, (int)std::min(wdt, (double)n + 4.0));
            }
        }
    }
    return std::min(bound + 2, n + 4);
}

// ---- synchronous (non-pipelined) form: stage kc rows, barrier, sample, barrier.  Small LDS footprint, so many
//      workgroups per CU hide the staging latency instead of a register prefetch.  Fallback for windows wider than the
//      pipelined kernel supports, and variant 2 for A/B measurement (1024^3 x 75 angles: 22.8 ms vs 15.6 ms pipelined).
template <bool LERP8, bool RESID, int BT, int A>
__global__ __launch_bounds__(BT) void fp_tiled_sync_kernel(FpTiledArgs a, int kc)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char fp_smem[];
    float4 *tile = reinterpret_cast<float4 *>(fp_smem);
    __shared__ int xlo_s[8], wid_s[8];
    const int per_zb = a.nut * a.ngroups;
    const int q = (int)blockIdx.x >> 3, xcd = (int)blockIdx.x & 7;
    const int per_xcd = (a.nzb * per_zb + 7) >> 3;  // a contiguous eighth of the (slice quad, tile, group) list per XCD
    const int wi = xcd * per_xcd + q;
    if (wi >= a.nzb * per_zb) return;
    const int zb = wi / per_zb;
    const int rest = wi % per_zb;
    const int ut = rest % a.nut, g = rest / a.nut;
    const int z0 = zb * 4, u0 = ut * BT, tid = (int)threadIdx.x, iu = u0 + tid, n = a.n;
    const int ng = min(A, a.n_class - g * A);
    const int *ord = a.order + g * A;
    const float half_n = 0.5f * (float)n - 0.5f, half_u = 0.5f * (float)a.nu - 0.5f, nf = (float)n;
    float offs[A], slope[A], acc[A][4];
#pragma unroll
    for (int i = 0; i < A; ++i) {
        const tomo_angle_t t = a.tab[ord[i < ng ? i : 0]];
        offs[i] = fmaf(((float)iu - half_u) + t.cor, t.inv, half_n);
        slope[i] = t.slope;
#pragma unroll
        for (int zz = 0; zz < 4; ++zz) acc[i][zz] = 0.0f;
    }
    const size_t zstride = (size_t)n * n;
    const float *p0 = a.src + (size_t)z0 * zstride;
    const float *p1 = a.src + (size_t)min(z0 + 1, a.nz - 1) * zstride;
    const float *p2 = a.src + (size_t)min(z0 + 2, a.nz - 1) * zstride;
    const float *p3 = a.src + (size_t)min(z0 + 3, a.nz - 1) * zstride;
    const unsigned k1 = z0 + 1 < a.nz ? 0xffffffffu : 0u, k2 = z0 + 2 < a.nz ? 0xffffffffu : 0u, k3 = z0 + 3 < a.nz ? 0xffffffffu : 0u;
    for (int k0 = 0; k0 < n; k0 += kc) {
        const int rows = min(kc, n - k0);
        __syncthreads();
        if (tid < rows) {
            const float kw = (float)(k0 + tid) - half_n;
            float fmin = 3.0e38f, fmax = -3.0e38f;
            for (int i = 0; i < ng; ++i) {
                const tomo_angle_t t = a.tab[ord[i]];
                const float o0 = fmaf(((float)u0 - half_u) + t.cor, t.inv, half_n);
                const float o1 = fmaf(((float)(u0 + BT - 1) - half_u) + t.cor, t.inv, half_n);
                const float f0 = fmaf(kw, t.slope, o0), f1 = fmaf(kw, t.slope, o1);
                fmin = fminf(fmin, fminf(f0, f1));
                fmax = fmaxf(fmax, fmaxf(f0, f1));
            }
            const int lo = (int)fminf(fmaxf(floorf(fmin), -2.0f), (float)n);
            const int hi = (int)fminf(fmaxf(floorf(fmax) + 1.0f, -1.0f), (float)(n + 1));
            xlo_s[tid] = lo;
            wid_s[tid] = max(hi - lo + 1, 2);
        }
        __syncthreads();
        for (int r = 0; r < rows; ++r) {
            const int lo = xlo_s[r], wid = wid_s[r];
            const unsigned rowoff = (unsigned)(k0 + r) * (unsigned)n;
            float4 *trow = tile + (size_t)r * a.wpitch;
            for (int j = tid; j < wid; j += BT) {
                const int x = lo + j;
                const unsigned mk = (x >= 0 && x < n) ? 0xffffffffu : 0u;
                const unsigned off = rowoff + (unsigned)min(max(x, 0), n - 1);
                float4 v;
                v.x = __uint_as_float(__float_as_uint(p0[off]) & mk);
                v.y = __uint_as_float(__float_as_uint(p1[off]) & (mk & k1));
                v.z = __uint_as_float(__float_as_uint(p2[off]) & (mk & k2));
                v.w = __uint_as_float(__float_as_uint(p3[off]) & (mk & k3));
                trow[j] = v;
            }
        }
        __syncthreads();
        for (int r = 0; r < rows; ++r) {
            const float kw = (float)(k0 + r) - half_n;
            int rb = __builtin_amdgcn_readfirstlane((r * a.wpitch - xlo_s[r]) * 16);  // see fp_tiled_kernel
            asm("" : "+s"(rb));
            const char *trow = reinterpret_cast<const char *>(tile);
#pragma unroll
            for (int i = 0; i < A; ++i) {
                const float f = fmaf(kw, slope[i], offs[i]);
                const float fl = floorf(f);
                const float w = lerp_w<LERP8>(f, fl), omw = 1.0f - w;
                const int idx = (int)__builtin_amdgcn_fmed3f(fl, -2.0f, nf);
                const float4 *tap = reinterpret_cast<const float4 *>(trow + ((idx << 4) + rb));
                const float4 s0 = tap[0], s1 = tap[1];
                acc[i][0] = fmaf(omw, s0.x, acc[i][0]); acc[i][0] = fmaf(w, s1.x, acc[i][0]);
                acc[i][1] = fmaf(omw, s0.y, acc[i][1]); acc[i][1] = fmaf(w, s1.y, acc[i][1]);
                acc[i][2] = fmaf(omw, s0.z, acc[i][2]); acc[i][2] = fmaf(w, s1.z, acc[i][2]);
                acc[i][3] = fmaf(omw, s0.w, acc[i][3]); acc[i][3] = fmaf(w, s1.w, acc[i][3]);
            }
        }
    }
    if (iu >= a.nu) return;
#pragma unroll
    for (int i = 0; i < A; ++i) {
        if (i >= ng) break;
        const int k_a = ord[i];
        const tomo_angle_t t = a.tab[k_a];
        float v4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int zz = 0; zz < 4; ++zz) {
            const int z = z0 + zz;
            if (z >= a.nz) break;
            float val = acc[i][zz] * t.scale;
            if (RESID) val = fp_residual_value(a, val, z, k_a, t.src, iu);
            v4[zz] = val;
            if (!(RESID && a.zquad)) a.out[((size_t)z * a.na + k_a) * a.nu + iu] = val;
        }
        if (RESID && a.zquad)
            reinterpret_cast<float4 *>(a.out)[((size_t)zb * a.na + k_a) * a.nu + iu] = make_float4(v4[0], v4[1], v4[2], v4[3]);
    }
}
