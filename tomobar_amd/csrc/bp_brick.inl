// Back projector, "brick" variant (default).  Included inside the anonymous namespace of proj_kernels.hip.
//
// A 256-thread workgroup owns a 32(x) x 16(y) x 16(z) voxel brick; a thread owns two voxel columns (same x, rows four
// apart) x 16 slices = 32 accumulators.  For every batch of 8 angles the part of the sinogram the brick can touch
// (<= 31|cos| + 15|sin| + 2 < 38 detector samples per angle and slice) is staged in LDS as float4 over z, then every
// tap is one ds_read_b128 serving 4 slices.  What the PMC counters said about the first tiled kernel (64 x 8 brick,
// lanes along x; profiles/archive/r1_bp_fp_pmc.txt) and what this kernel does about it:
//   * half of its VALU instructions were staging (div/mod item decoding, 64-bit addresses, four guarded loads per
//     item): here an item's LDS slot and global offset are loop invariants, loads are unconditional on clamped
//     addresses (slices >= nz are fed by a valid slice and never stored), 5 items per thread instead of 9;
//   * 28 % of its LDS cycles were bank conflicts: ds_read_b128 is served in 16-lane groups that are NOT contiguous
//     ({0-3,12-15,20-27}, ...), so with lanes along x a group spans up to 27|cos| slots, more than the 16 slots of a
//     bank row.  Here a wave is a 16(x) x 4(y) patch: every group spans <= 15|cos| + |sin| < 16 slots;
//   * staging latency was exposed (stage, barrier, sample, barrier): here the loads of batch b+1 are in flight (held
//     in registers) while batch b is sampled.
// Accumulation order (angles ascending, tap 0 then tap 1, fmaf) is that of the oracle: results are bit-identical.
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int BB_TX = 32, BB_TY = 16, BB_ZQ = 4, BB_AB = 8;
// columns per (angle, z-quad) row of the tile.  A brick sees at most 31|cos| + 15|sin| <= 34.5 detector pixels plus the
// second tap and the floor: 37 columns.  Planar staging: 40 (5 items per thread exactly).  Quad-interleaved staging (LDS-DMA,
// no register budget to respect): 39, so that two tile buffers + the window table stay within 160 KiB / 4 = 40 960 B and
// FOUR workgroups share a CU (the last of its 5 item rounds is 224 of 256 threads).
template <bool ZQ> constexpr int bb_pitch = ZQ ? 39 : 40;
template <bool ZQ> constexpr int bb_nitems = BB_AB * BB_ZQ * bb_pitch<ZQ>;
template <bool ZQ> constexpr int bb_items = (bb_nitems<ZQ> + 255) / 256;  // staging items per thread and batch (5)
static_assert(bb_nitems<false> == 256 * bb_items<false>, "planar staging items must divide evenly");
typedef __attribute__((address_space(3))) void *bb_lds_ptr;
// One LDS-DMA: 16 bytes per lane from buffer `rs` at byte offset `voff` (out of range -> zeros) to LDS byte address
// `lds_addr` + 16 * lane (`lds_addr` wave-uniform).  Issued through inline assembly on purpose: hipcc tracks an LDS-DMA issued
// by the builtin as a pending LDS WRITE and, unless it can prove the addresses apart (it cannot: the tile buffers alternate
// and the window slots rotate by run-time indices), waits vmcnt(0) before the next ds_read -- i.e. it drained the staging of
// batch b+1 before the first sample of batch b (seen in the ISA).  The kernel orders the hand-over itself: s_waitcnt vmcnt(0)
// + the batch barrier before a tile is sampled.  Loads return in order, so the compiler's own counted waits stay conservative.
// (s_nop: one wait state between an SALU write of M0 and an LDS-DMA that reads it.)
__device__ __forceinline__ void bb_dma16(__amdgpu_buffer_rsrc_t rs, unsigned lds_addr, int voff)
{
    // M0 cannot be listed as a clobber: to this compiler it is a RESERVED register ("inline asm clobber list contains reserved
    // registers: m0"), i.e. it never allocates it or keeps a value live in it -- it writes M0 itself immediately before each of
    // its own M0-reading instructions (LDS-direct, indirect register indexing; none in this kernel), so this statement cannot
    // break a live value; tests/test_gpu_parity.py pins the kernel's output bit for bit should a compiler change that.
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rs) : "memory");
}

// ZQ (round 5): the sinogram is the PRIVATE residual layout [z/4][angle][u][4] that tomo_fp3d_residual leaves on a context
// set to TOMO_RESIDUAL_ZQUAD: a staging item is ONE 16-byte word (whole 624-byte runs per angle and quad) instead of four
// dword gathers from rows nz * na * nu floats apart, and since an item is one word and the tile is lane-linear in the item
// index it goes from memory STRAIGHT INTO LDS (`buffer_load_dwordx4 ... lds`): no staging registers, no ds_write_b128 (13 clk
// each on the VGPR -> LDS path), the descriptor's range check zero-fills samples outside the detector, and the batch b+1 is
// in flight while batch b is sampled exactly as before.  The sampling and the epilogue are unchanged.
template <int EPI, bool LERP8, bool ZQ = false>
__global__ __launch_bounds__(256) void bp_brick_kernel(BpArgs a)
{
    constexpr int BB_PITCH = bb_pitch<ZQ>, BB_ITEMS = bb_items<ZQ>, BB_NITEMS = bb_nitems<ZQ>;
    // ONE __shared__ object: with a second one, however small, hipcc waits vmcnt(0) before the ds_reads of every angle while
    // an LDS-DMA is in flight (cdna_hip_programming.md, ".s-level traps"), i.e. the staging of batch b+1 would be drained before
    // batch b is sampled.  Layout: the double-buffered tile [2][angle][z-quad][u] (2 x 20 KiB), then the window table.
    // ... and, for the same reason, the angle records of a batch ((cos, sin, detector offset) per angle; ZQ only): read from
    // global memory inside the sampling loop they are ordinary VGPR-destination loads, and beside an LDS-DMA hipcc waits vmcnt(0)
    // for those.  The thread that computes an angle's window leaves the record in LDS, the sampling reads it from there.
    __shared__ float4 bb_smem[2 * BB_NITEMS + (3 * BB_AB * 4 + 15) / 16 + (ZQ ? 3 * BB_AB : 0)];
    float4(*tile2)[BB_NITEMS] = reinterpret_cast<float4(*)[BB_NITEMS]>(bb_smem);
    // three slots: a slow wave may still sample batch b-1 while batch b+1's window is written
    int(*umin_s)[BB_AB] = reinterpret_cast<int(*)[BB_AB]>(bb_smem + 2 * BB_NITEMS);
    [[maybe_unused]] float4(*ang_s)[BB_AB] = reinterpret_cast<float4(*)[BB_AB]>(bb_smem + 2 * BB_NITEMS + (3 * BB_AB * 4 + 15) / 16);

    // XCD-aware numbering: workgroup b lands on XCD b%8; give each XCD its own z-brick stream
    const int ntiles = a.ntx * a.nty;
    // workgroup b lands on XCD b % 8: every XCD gets one CONTIGUOUS eighth of the (z-brick, tile) list -- its own z-brick
    // stream (the sinogram rows of a z-brick stay in that XCD's L2) and the same number of workgroups whatever nzb is
    // (with "z-brick zb -> XCD zb % 8" a 270-slice slab = 17 z-bricks gave XCD 0 three bricks and the others two)
    const int b = blockIdx.x, xcd = b & 7, q = b >> 3;
    const int per_xcd = (a.nzb * ntiles + 7) >> 3;
    const int wi = xcd * per_xcd + q;
    if (wi >= a.nzb * ntiles) return;  // uniform for the workgroup
    const int zb = wi / ntiles;
    const int tq = wi % ntiles;
    const int tx0 = (tq % a.ntx) * BB_TX, ty0 = (tq / a.ntx) * BB_TY;
    const int z0 = zb * (4 * BB_ZQ);

    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ix = tx0 + (wave & 1) * 16 + (lane & 15);
    const int iy0 = ty0 + (wave >> 1) * 8 + (lane >> 4);  // rows iy0 and iy0 + 4
    const float half_n = 0.5f * (float)a.n - 0.5f, half_u = 0.5f * (float)a.nu - 0.5f;
    const float xw = (float)ix - half_n;
    const float yw0 = (float)iy0 - half_n, yw1 = (float)(iy0 + 4) - half_n;

    // accumulators as explicit 2-vectors (slices 2h, 2h+1): one v_pk_fma_f32 per tap and slice pair.  Left to itself the
    // SLP vectoriser pairs accumulators of DIFFERENT rows and pays ~35 v_mov per angle to assemble the operands.
    v2f acc[2][2 * BB_ZQ];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < 2 * BB_ZQ; ++j) acc[r][j] = v2f{0.0f, 0.0f};

    // ---- loop-invariant part of the staging items: item = tid + 256 m  ->  (angle slot, z-quad, column)
    const unsigned zstride = (unsigned)a.na * (unsigned)a.nu;  // elements; host guarantees 16 * zstride * 4 < 2^32
    const int zlim = a.nz - 1 - z0;                            // last valid slice relative to z0 (>= 0)
    const bool ragged = zlim < 4 * BB_ZQ - 1;                  // uniform: only the last z-brick
    int it_aa[BB_ITEMS], it_j[BB_ITEMS];
    unsigned it_off[BB_ITEMS];  // element offset of (slice z0 + 4 zq, angle slot, column 0) from the batch base
#pragma unroll
    for (int m = 0; m < BB_ITEMS; ++m) {
        const int item = tid + 256 * m;
        it_j[m] = item % BB_PITCH;
        const int zq = (item / BB_PITCH) % BB_ZQ;
        it_aa[m] = item / (BB_PITCH * BB_ZQ);
        if (ZQ) it_off[m] = (unsigned)min(zq, zlim >> 2) * zstride + (unsigned)it_aa[m] * (unsigned)a.nu;  // float4 units
        else it_off[m] = (unsigned)min(4 * zq, zlim) * zstride + (unsigned)it_aa[m] * (unsigned)a.nu;
    }
    const float *sino_z0 = a.sino + (size_t)z0 * zstride;  // ZQ: quad z0 / 4 starts at float4 index (z0 / 4) * zstride = float z0 * zstride, too

    auto window = [&](int a0, int buf) {  // detector window of the brick for the 8 angles of a batch
        if (tid < BB_AB) {
            // the coordinate is monotone in x and in y, so the four corners bound it
            const tomo_angle_t t = a.tab[min(a0 + tid, a.na - 1)];
            const float off = half_u - t.cor;
            const float x0 = (float)tx0 - half_n, x1 = (float)(tx0 + BB_TX - 1) - half_n;
            const float y0 = (float)ty0 - half_n, y1 = (float)(ty0 + BB_TY - 1) - half_n;
            const float f00 = fmaf(x0, t.cs, fmaf(y0, t.sn, off)), f10 = fmaf(x1, t.cs, fmaf(y0, t.sn, off));
            const float f01 = fmaf(x0, t.cs, fmaf(y1, t.sn, off)), f11 = fmaf(x1, t.cs, fmaf(y1, t.sn, off));
            // clamp so that the int conversion and the offsets below stay in range whatever the geometry
            const float lo = fminf(fminf(f00, f10), fminf(f01, f11));
            const int um = (int)fminf(fmaxf(floorf(lo), -1.0e6f), 1.0e6f);
            umin_s[buf][tid] = um;
            // (the window origin rides in the record's fourth word: the sampling gets it with the same ds_read_b128)
            if constexpr (ZQ) ang_s[buf][tid] = make_float4(t.cs, t.sn, off, __int_as_float(um));
        }
    };

    [[maybe_unused]] float4 pre[BB_ITEMS];  // planar staging only
    const int wave_item0 = __builtin_amdgcn_readfirstlane(tid & ~63);  // first item of this wave in every item round
    // ZQ: batch a0 -> LDS tile `dst` directly (one LDS-DMA per item, zero outside the detector)
    auto dma = [&](int a0, int buf, float4 *dst) {
        const int amax = a.na - 1 - a0;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(reinterpret_cast<const float4 *>(sino_z0) + (size_t)a0 * a.nu), 0,
            (int)(((unsigned)min((zlim >> 2) + 1, BB_ZQ) * zstride - (unsigned)a0 * (unsigned)a.nu) << 4), 0x00020000);
#pragma unroll
        for (int m = 0; m < BB_ITEMS; ++m) {
            // (the items of the partial last round beyond BB_NITEMS belong to an angle slot >= BB_AB: nothing is issued for them,
            // and their window read stays inside the table)
            const int u = umin_s[buf][min(it_aa[m], BB_AB - 1)] + it_j[m];
            unsigned off = it_off[m] + (unsigned)min(max(u, 0), a.nu - 1);
            if (it_aa[m] > amax) off -= (unsigned)(it_aa[m] - amax) * (unsigned)a.nu;
            const int boff = (u >= 0 && u < a.nu) ? (int)(off << 4) : (int)0x80000000;
            if (256 * (m + 1) <= BB_NITEMS || tid + 256 * m < BB_NITEMS)  // (the last round is partial)
                bb_dma16(rs, (unsigned)(uintptr_t)(bb_lds_ptr)(dst + wave_item0 + 256 * m), boff);
        }
    };
    auto prefetch = [&](int a0, int buf) {  // planar: batch a0 -> registers (zero outside the detector)
        const float *base = sino_z0 + (size_t)a0 * a.nu;
        const int amax = a.na - 1 - a0;  // angle slots beyond the subset re-read the last angle (never sampled)
#pragma unroll
        for (int m = 0; m < BB_ITEMS; ++m) {
            const int u = umin_s[buf][it_aa[m]] + it_j[m];
            const unsigned mk = (u >= 0 && u < a.nu) ? 0xffffffffu : 0u;
            unsigned off = it_off[m] + (unsigned)min(max(u, 0), a.nu - 1);
            if (it_aa[m] > amax) off -= (unsigned)(it_aa[m] - amax) * (unsigned)a.nu;
            float4 v;
            if (!ragged) {
                v.x = base[off];
                v.y = base[off + zstride];
                v.z = base[off + 2 * zstride];
                v.w = base[off + 3 * zstride];
            } else {  // slices past the end of the volume are fed by the last valid one; their accumulators are not stored
                const int zrel = min(4 * (((tid + 256 * m) / BB_PITCH) % BB_ZQ), zlim);
                v.x = base[off];
                v.y = base[off + (unsigned)(min(zrel + 1, zlim) - zrel) * zstride];
                v.z = base[off + (unsigned)(min(zrel + 2, zlim) - zrel) * zstride];
                v.w = base[off + (unsigned)(min(zrel + 3, zlim) - zrel) * zstride];
            }
            v.x = __uint_as_float(__float_as_uint(v.x) & mk);
            v.y = __uint_as_float(__float_as_uint(v.y) & mk);
            v.z = __uint_as_float(__float_as_uint(v.z) & mk);
            v.w = __uint_as_float(__float_as_uint(v.w) & mk);
            pre[m] = v;
        }
    };

    window(0, 0);
    __syncthreads();
    if constexpr (ZQ) dma(0, 0, tile2[0]);
    else prefetch(0, 0);
    int buf = 0;
    for (int a0 = 0; a0 < a.na; a0 += BB_AB, buf = (buf == 2 ? 0 : buf + 1)) {
        const int nb = min(BB_AB, a.na - a0);
        const bool more = a0 + BB_AB < a.na;
        const int nbuf = (buf == 2 ? 0 : buf + 1);
        if (more) window(a0 + BB_AB, nbuf);
        // double-buffered tile: waves still sampling batch b-1 read the other buffer, so one barrier per batch is enough
        float4 *tile = tile2[(a0 / BB_AB) & 1];
        if constexpr (ZQ) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's pieces of tile b have landed in LDS
        } else {
#pragma unroll
            for (int m = 0; m < BB_ITEMS; ++m) tile[tid + 256 * m] = pre[m];
        }
        __syncthreads();  // tile b and the window of batch b+1 visible; everyone is past the sampling of batch b-1
        if (more) {       // in flight while this batch is sampled
            if constexpr (ZQ) dma(a0 + BB_AB, nbuf, tile2[((a0 / BB_AB) + 1) & 1]);
            else prefetch(a0 + BB_AB, nbuf);
        }
        auto sample = [&](int aa) {
            float t_cs, t_sn, off;
            int um;
            if constexpr (ZQ) {
                // (cos, sin, detector offset, window origin) as ONE ds_read_b128 (4 LDS cycles per wave).  Round 6: with three words
                // used the compiler emitted a ds_read_b96, which takes 8, and the origin came from a separate ds_read_b32 -- together
                // an eighth of the tap reads' cycles on the pipe that bounds this kernel
                const float4 an = ang_s[buf][aa];
                t_cs = an.x; t_sn = an.y; off = an.z; um = __float_as_int(an.w);
            } else {
                const tomo_angle_t t = a.tab[a0 + aa];
                t_cs = t.cs; t_sn = t.sn; off = half_u - t.cor; um = umin_s[buf][aa];
            }
            // the window origin is wave-uniform: fold it into a scalar byte offset so that a tap address is one v_lshl_add
            int ab = __builtin_amdgcn_readfirstlane((aa * (BB_ZQ * BB_PITCH) - um) * 16);
            asm("" : "+s"(ab));  // opaque: otherwise the *16 is factored back out (a second VALU op per tap)
            const char *tb = reinterpret_cast<const char *>(tile);
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const float f = fmaf(xw, t_cs, fmaf(r ? yw1 : yw0, t_sn, off));
                const float fl = floorf(f);
                const float w = lerp_w<LERP8>(f, fl), omw = 1.0f - w;
                const int idx = (int)fl;
                const v2f w2 = v2f{w, w}, omw2 = v2f{omw, omw};
#pragma unroll
                for (int zq = 0; zq < BB_ZQ; ++zq) {
                    const v4f *tap = reinterpret_cast<const v4f *>(tb + ((idx << 4) + ab)) + zq * BB_PITCH;
                    const v4f s0 = tap[0], s1 = tap[1];
                    v2f &c0 = acc[r][zq * 2], &c1 = acc[r][zq * 2 + 1];
                    c0 = __builtin_elementwise_fma(omw2, s0.lo, c0); c0 = __builtin_elementwise_fma(w2, s1.lo, c0);
                    c1 = __builtin_elementwise_fma(omw2, s0.hi, c1); c1 = __builtin_elementwise_fma(w2, s1.hi, c1);
                }
            }
        };
        if (nb == BB_AB) {  // unrolled: the 8 angle records become one block of scalar loads ahead of the sampling
#pragma unroll
            for (int aa = 0; aa < BB_AB; ++aa) sample(aa);
        } else {
            for (int aa = 0; aa < nb; ++aa) sample(aa);
        }
    }
    // ---- epilogue.  A brick that lies inside the volume hands its 32 x 16 x 16 accumulators over through LDS (the tile
    // buffers are free now) so that a lane owns four consecutive voxels of a row: 8 dwordx4 accesses per thread and array
    // in whole 128-byte lines instead of 32 dword accesses in 64-byte row pieces (the fused FISTA epilogue cost +1.9 ms
    // per 1024^3 call for +0.9 ms of compulsory traffic).  Same arithmetic per voxel.
    const bool vec_ok = (tx0 + BB_TX <= a.n) && (ty0 + BB_TY <= a.n) && (z0 + 4 * BB_ZQ <= a.nz) && ((a.n & 3) == 0) && a.epi_aligned
                        && EPI != EPI_PLAIN;  // (plain stores gain nothing: 7.26 -> 7.32 ms)
    if (vec_ok) {  // uniform for the workgroup
        float *stg = reinterpret_cast<float *>(&tile2[0][0]);  // [16 z][16 y][32 x] floats = 32 KiB of the 40
        __syncthreads();  // every wave is past the sampling of the last batch
        const int xr = ix - tx0, yr = iy0 - ty0;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int j = 0; j < 4 * BB_ZQ; ++j) stg[(j * BB_TY + yr + 4 * r) * BB_TX + xr] = acc[r][j >> 1][j & 1];
        __syncthreads();
#pragma unroll
        for (int m = 0; m < 4 * BB_ZQ * BB_TY * (BB_TX / 4) / 256; ++m) {
            const int item = tid + 256 * m;  // float4 index = (z * 16 + y) * 8 + x4
            const int x4 = item & (BB_TX / 4 - 1), yy = (item / (BB_TX / 4)) & (BB_TY - 1), zz = item / (BB_TX / 4 * BB_TY);
            const float4 g = reinterpret_cast<const float4 *>(stg)[item];
            bp_epilogue4<EPI>(a, ((size_t)(z0 + zz) * a.n + (ty0 + yy)) * a.n + tx0 + 4 * x4, g);
        }
        return;
    }
    if (ix < a.n) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int iy = iy0 + 4 * r;
            if (iy < a.n) {
#pragma unroll
                for (int j = 0; j < 4 * BB_ZQ; ++j)
                    if (z0 + j < a.nz) bp_epilogue<EPI>(a, ((size_t)(z0 + j) * a.n + iy) * a.n + ix, acc[r][j >> 1][j & 1]);
            }
        }
    }
}
