// 16-bit MFMA GEMM for the encoder linears on gfx950:  C[m][n] = sum_k A[m][k] * W[n][k]  (+ epilogue)
//
//   A : activations [M][lda], W : torch Linear weight [N][ldw], both 16-bit (bf16 or fp16), K contiguous.
//   Tile 256 (m) x 256 (n), K consumed 64 at a time; 512 threads = 8 waves laid out 2 (m) x 4 (n), each
//   wave owning 128 (m) x 64 (n) as 4 x 2 tiles of v_mfma_f32_32x32x16.  The MFMA takes the W fragment as
//   its A operand and the activation fragment as its B operand, so the accumulator has n in registers
//   (4 consecutive n per register quad) and m on lanes: the epilogue moves 16 bytes per lane.
//
//   Staging.  Operands go global -> LDS by 16-byte LDS-DMA loads (no VGPR round trip).  Such a load writes
//   1 KiB linearly per wave, so the bank swizzle is put on the per-lane SOURCE address and undone on the
//   fragment read: 16-byte chunk c of tile row r lives at chunk c ^ ((r >> 1) & 7), which makes every
//   ds_read_b128 lane group hit 16 distinct 16-byte slots.
//
//   Pipeline.  At 256 x 256 the kernel needs ~128 flop per byte pulled from L2, i.e. most of what a CU can
//   pull, so the loads must never drain.  A K-tile is split in four 16 KiB half-tiles (A0/A1 = the first /
//   second 64 rows of every wave's 128, B0/B1 = the first / second 32 columns of every wave's 64), LDS holds
//   two K-tiles, and every K-tile is consumed in four phases of 8 MFMAs, one output quadrant each:
//       even tile: (A0,B0) (A0,B1) (A1,B1) (A1,B0)      odd tile: (A0,B1) (A0,B0) (A1,B0) (A1,B1)
//   so a half-tile is dead after a known phase and is refilled two phases later with the same half of the
//   tile two K-tiles ahead: six half-tiles (96 KiB) are in flight in steady state.  Each phase ends with a
//   counted "s_waitcnt vmcnt(10)" (all but the five youngest half-tiles of this wave have landed) and one
//   s_barrier; no wait drains the queue before the last two K-tiles.  Fragment registers are refilled
//   right after the MFMA that used them last, one phase before they are needed, so LDS latency is hidden
//   without a second register set.
//
//   Workgroup ids are remapped so that each of the 8 XCDs (private L2s) walks a contiguous range of tiles:
//   the tiles sharing an activation panel run on one XCD at about the same time.
//
//   Half-height variant (MS = 1).  A product with few output tiles (16384 rows x 512 columns: 128 tiles of 256 x 256
//   on 256 CUs) leaves half the chip idle, so launch_gemm then takes tiles of 128 (m) x 256 (n): the same eight waves
//   2 (m) x 4 (n), each owning 64 x 64 as 2 x 2 MFMA tiles, the same four phases of now 4 MFMAs, activation half-tiles
//   of 8 KiB (one DMA load per thread instead of two; they keep their 16 KiB slots).  It reads one fragment from LDS
//   per MFMA instead of three per four, which is about what the LDS delivers: slower per flop than the full tile,
//   faster than half a chip.
#include "common.h"
#include "elem.h"
#include "gemm.h"
#include "lds_dma.h"

namespace asrfe {

constexpr int BN = 256, BK = 64;  // (tile height: 128 * MS rows)
constexpr int kGemmThreads = 512;
constexpr int kHalfBytes = 128 * BK * 2;     // 16 KiB: one half-tile
constexpr int kSlotBytes = 4 * kHalfBytes;   // A0 | A1 | B0 | B1
constexpr int kRingBytes = 2 * kSlotBytes;   // 128 KiB
constexpr int kGemmLds = kRingBytes + 8 * 4096;  // + one 4 KiB epilogue slab per wave = 160 KiB

// End of a phase: this wave's loads for the next phase have landed and its own fragment reads have returned, then all
// waves meet.  One asm statement, so nothing is scheduled between the wait and the barrier.  The lgkmcnt(0) is not
// optional: the phase behind this barrier starts by refilling a half-tile that other waves read during the phase in
// front of it (the rolling activation fragments, whose last read is issued right before the barrier).  A read that
// is merely queued in the LDS pipeline when its wave reaches the barrier can still lose against the refill: when the
// LDS is saturated (the half-height variant) and the DMA data comes from a hot L2 line (activations the previous
// kernel has just written), the new data has been seen to land first, and the wave then multiplies part of K-tile
// g + 2 into K-tile g.  With every wave's reads returned before it arrives, no read of a half-tile is left anywhere
// when the first refill behind the barrier is issued.  (The reads outstanding at this point are the ones the next
// phase's first MFMAs need anyway; the drain costs 1 to 2 % of a product.)  The LDS-DMA loads are issued through
// inline assembly (lds_dma.h), so within a phase the compiler's waits for the fragment reads stay counted.
// The five youngest half-tiles are ten loads at full height.  At half height an activation half-tile is one load and a
// weight half-tile two, and the five youngest are, depending on the phase, three activation + two weight half-tiles (7
// loads) or two + three (8): the wait uses 7 everywhere, which at the second kind of phase also waits for one load more
// than needed.
#define PHASE_SYNC(steady)                                                                         \
    do {                                                                                           \
        if (steady) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(MS == 2 ? 10 : 7) : "memory"); \
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");                        \
    } while (0)

// 16-byte global store with a selectable cache policy: 0 plain, 1 non-temporal, 2 write-through (sc1)
__device__ __forceinline__ void store16(void* ptr, uint4 v, int policy) {
    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
    const u32x4 d = {v.x, v.y, v.z, v.w};
    if (policy == 1) {
        __builtin_nontemporal_store(d, reinterpret_cast<u32x4*>(ptr));
    } else if (policy == 2) {
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(ptr), "v"(d) : "memory");
    } else {
        *reinterpret_cast<u32x4*>(ptr) = d;
    }
}

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Epilogue of one output tile (the bias is already in the accumulator: it is its initial value).  In the accumulator a lane owns row m and register quad g owns the columns
// nb + 8 g + 4 h + {0..3}, so a direct store would scatter 32-byte pieces over 32 rows.  Instead every
// 32-row slab goes through a 4 KiB wave-private LDS tile (128-byte rows, 16-byte chunk c of row r kept at
// chunk c ^ (r & 7): conflict-free both ways) and comes back with 8 consecutive lanes holding one full
// 128-byte line, so each global access moves whole lines.  Only wave-level ordering is needed.
template <int EPI, bool F16, int MS>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, const f32x16 (&acc)[2][2 * MS], int m0, int n0, int wm, int wn,
                                              int lane, char* es, const float (&rsum)[2 * MS][4], const float (&rsq)[2 * MS][4]) {
    const int l31 = lane & 31, h = lane >> 5;
    const int wr_row = l31 * 128, wr_sw = l31 & 7;   // this lane's row in the slab when writing
    const int rd_row = lane >> 3, rd_c = lane & 7;   // row (+ 8 i) and chunk when reading back
    const bool live = !(p.flags & 1);
    const int policy = (p.flags >> 4) & 3;
    if constexpr (EPI == EPI_ARGMAX) {
        // a lane meets its 32 columns of the wave's 64-column strip in rising order (register e of tile ns is column
        // 32 ns + 8 (e >> 2) + 4 h + (e & 3)), so the strict comparison keeps the lowest column of equal values; the
        // other 32 columns of the row sit in lane ^ 32
        const int nb = n0 + wn * 64 + 4 * h;
        const int nv = p.n_valid > 0 ? p.n_valid : p.N;
        uint2* pairs = reinterpret_cast<uint2*>(p.out);
#pragma unroll
        for (int ms = 0; ms < 2 * MS; ++ms) {
            float best = -INFINITY;
            int bi = 0x7fffffff;
#pragma unroll
            for (int ns = 0; ns < 2; ++ns)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int n = nb + ns * 32 + 8 * (e >> 2) + (e & 3);
                    const float v = acc[ns][ms][e];
                    if (n < nv && v > best) { best = v; bi = n; }
                }
            const float ov = __shfl_xor(best, 32);
            const int oi = __shfl_xor(bi, 32);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
            const int m = m0 + wm * (64 * MS) + ms * 32 + l31;
            if (h == 0 && m < p.M && live)
                pairs[static_cast<int64_t>(m) * p.ldo + (n0 >> 6) + wn] = make_uint2(__float_as_uint(best), static_cast<uint32_t>(bi));
        }
        return;
    }
    if constexpr (EPI == EPI_F32_ADD || EPI == EPI_F32_NORM) {
#pragma unroll
        for (int ns = 0; ns < 2; ++ns) {
            const int nb = n0 + wn * 64 + ns * 32;
#pragma unroll
            for (int ms = 0; ms < 2 * MS; ++ms) {
                const f32x16 a = acc[ns][ms];
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4*>(es + wr_row + (((2 * g + h) ^ wr_sw) << 4)) =
                        make_float4(a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]);
                wave_lds_sync();
                const int mb = m0 + wm * (64 * MS) + ms * 32;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = rd_row + 8 * i;
                    float4 v = *reinterpret_cast<const float4*>(es + r * 128 + ((rd_c ^ (r & 7)) << 4));
                    const int m = mb + r;
                    if constexpr (EPI == EPI_F32_NORM) {
                        const float mean = rsum[ms][i], rstd = rsq[ms][i];
                        const float4 c4 = *reinterpret_cast<const float4*>(p.colsum + nb + rd_c * 4);
                        const float4 d4 = *reinterpret_cast<const float4*>(p.post_bias + nb + rd_c * 4);
                        v.x = (v.x - mean * c4.x) * rstd + d4.x; v.y = (v.y - mean * c4.y) * rstd + d4.y;
                        v.z = (v.z - mean * c4.z) * rstd + d4.z; v.w = (v.w - mean * c4.w) * rstd + d4.w;
                    }
                    if (m < p.M && live) {
                        const int n = nb + rd_c * 4;
                        if (p.add16) {
                            const uint2 t2 = *reinterpret_cast<const uint2*>(p.add16 + static_cast<int64_t>(m) * p.ld16 + n);
                            float e0, e1, e2, e3;
                            unpack_elem2<F16>(t2.x, e0, e1);
                            unpack_elem2<F16>(t2.y, e2, e3);
                            v.x += e0; v.y += e1; v.z += e2; v.w += e3;
                        }
                        if (p.add32) {
                            const float4 t4 = *reinterpret_cast<const float4*>(p.add32 + static_cast<int64_t>(m) * p.ld32 + n);
                            v.x += t4.x; v.y += t4.y; v.z += t4.z; v.w += t4.w;
                        }
                        store16(reinterpret_cast<float*>(p.out) + static_cast<int64_t>(m) * p.ldo + n,
                                make_uint4(__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)), policy);
                    }
                }
                wave_lds_sync();
            }
        }
    } else {
#pragma unroll
        for (int ms = 0; ms < 2 * MS; ++ms) {
            // (EPI_RELU_STAT) this lane's 32 columns of row l31, as packed pairs: the epilogue runs with the matrix pipe
            // idle, so its vector instructions are all exposed
            typedef float v2f __attribute__((ext_vector_type(2)));
            v2f s1[2] = {{0.0f, 0.0f}, {0.0f, 0.0f}}, s2[2] = {{0.0f, 0.0f}, {0.0f, 0.0f}};
#pragma unroll
            for (int ns = 0; ns < 2; ++ns) {
                const f32x16 a = acc[ns][ms];
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = a[8 * gp + e];
                    if constexpr (EPI == EPI_BF16_RELU || EPI == EPI_RELU_STAT) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.0f);
                    }
                    if constexpr (EPI == EPI_RELU_STAT) {
#pragma unroll
                        for (int e = 0; e < 8; e += 2) {
                            const v2f t = {v[e], v[e + 1]};
                            asm("v_pk_add_f32 %0, %0, %1" : "+v"(s1[(e >> 1) & 1]) : "v"(t));
                            asm("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(s2[(e >> 1) & 1]) : "v"(t));
                        }
                    }
                    uint32_t lo0 = pack_elem2<F16>(v[0], v[1]), lo1 = pack_elem2<F16>(v[2], v[3]);
                    uint32_t hi0 = pack_elem2<F16>(v[4], v[5]), hi1 = pack_elem2<F16>(v[6], v[7]);
                    // lanes 32..63 of lo swap with lanes 0..31 of hi: h = 0 ends with columns 0..7, h = 1 with 8..15
                    auto s0 = __builtin_amdgcn_permlane32_swap(lo0, hi0, false, false);
                    auto s1 = __builtin_amdgcn_permlane32_swap(lo1, hi1, false, false);
                    const int chunk = ns * 4 + 2 * gp + h;  // 16-byte chunk of the 128-byte (64-column) row
                    *reinterpret_cast<uint4*>(es + wr_row + ((chunk ^ wr_sw) << 4)) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                }
            }
            wave_lds_sync();
            const int mb = m0 + wm * (64 * MS) + ms * 32;
            if constexpr (EPI == EPI_RELU_STAT) {
                float t1 = (s1[0].x + s1[0].y) + (s1[1].x + s1[1].y), t2 = (s2[0].x + s2[0].y) + (s2[1].x + s2[1].y);
                t1 += __shfl_xor(t1, 32);
                t2 += __shfl_xor(t2, 32);
                if (h == 0 && mb + l31 < p.M && live)
                    reinterpret_cast<float2*>(p.stat_out)[static_cast<int64_t>((n0 >> 6) + wn) * p.M + mb + l31] = make_float2(t1, t2);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = rd_row + 8 * i;
                const uint4 v = *reinterpret_cast<const uint4*>(es + r * 128 + ((rd_c ^ (r & 7)) << 4));
                const int m = mb + r;
                if (m < p.M && live)
                    store16(reinterpret_cast<uint16_t*>(p.out) + static_cast<int64_t>(m) * p.ldo + n0 + wn * 64 + rd_c * 8, v, policy);
            }
            wave_lds_sync();
        }
    }
}

template <int EPI, bool F16, int MS>
__global__ __launch_bounds__(kGemmThreads) void gemm_bf16_kernel(const GemmParams p) {
    constexpr int BM = 128 * MS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    saturate_fp16_conversions<F16>();
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    // Persistent workgroups: the 8 XCDs (private L2s) each own a contiguous range of output tiles, and the
    // workgroups of an XCD walk that range together, so tiles sharing an activation panel meet in one L2.
    const int n_tiles = p.n_tiles_m * p.n_tiles_n;
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7;
    const int wg_in_xcd = bid >> 3, wgs_of_xcd = (nwg - xcd + 7) >> 3;
    const int tq = n_tiles >> 3, tr = n_tiles & 7;
    const int tile_lo = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
    const int tile_hi = tile_lo + tq + (xcd < tr ? 1 : 0);
    const int first_tile = tile_lo + wg_in_xcd;
    const int n_my = first_tile < tile_hi ? (tile_hi - first_tile + wgs_of_xcd - 1) / wgs_of_xcd : 0;
    if (n_my == 0) return;
    const int nk = p.K / BK;
    const int n_kt = n_my * nk;  // K-tiles this workgroup consumes, over all its output tiles

    // ---- staging: thread covers LDS slot (local row j * 64 + (tid >> 3), chunk tid & 7) of a half-tile
    //      (an activation half-tile has 64 * MS rows: local row wm * 32 MS + r is row r of wave row wm's share)
    const int srow = tid >> 3;
    const int schunk = (tid & 7) ^ ((tid >> 4) & 7);  // source chunk that belongs in that slot
    const int wave_slot = wave * 1024;
    const char* a_tile;
    const char* w_tile;
    uint32_t aoff[2][MS], boff[2][2];  // [half][j] byte offsets from the tile base
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int cb = (2 * j + (tid >> 8)) * 64 + hf * 32 + (srow & 31);  // tile column behind this local row of weight half hf
            boff[hf][j] = static_cast<uint32_t>(cb) * p.ldw * 2 + schunk * 16;
        }
    int st_tile = first_tile;  // output tile and K-tile the next staged K-tile belongs to
    int st_kt = 0;
    int st_left = n_kt;        // K-tiles still to stage
    auto stage_set_tile = [&]() {
        const int tm = st_tile / p.n_tiles_n, tn = st_tile - tm * p.n_tiles_n;
        const int sm0 = tm * BM;
        a_tile = reinterpret_cast<const char*>(p.A) + static_cast<int64_t>(sm0) * p.lda * 2;
        w_tile = reinterpret_cast<const char*>(p.W) + static_cast<int64_t>(tn * BN) * p.ldw * 2;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int j = 0; j < MS; ++j) {
                // tile row behind this local row of activation half hf
                int ra_ = MS == 2 ? j * 128 + hf * 64 + srow : (srow >> 5) * 64 + hf * 32 + (srow & 31);
                ra_ = (sm0 + ra_ < p.M) ? ra_ : (p.M - 1 - sm0);
                aoff[hf][j] = static_cast<uint32_t>(ra_) * p.lda * 2 + schunk * 16;
            }
    };
    // after the four halves of a K-tile have been issued
    auto stage_advance = [&]() {
        --st_left;
        if (++st_kt == nk) {
            st_kt = 0;
            st_tile += wgs_of_xcd;
            if (st_left > 0) stage_set_tile();
        }
    };
    // half: 0 A0, 1 A1, 2 B0, 3 B1
    const uint32_t smem_base = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    auto stage = [&](int half, int slot) {
        const uint32_t dst = __builtin_amdgcn_readfirstlane(smem_base + slot * kSlotBytes + half * kHalfBytes + wave_slot);
        const int kb = st_kt * (BK * 2);
        if (half < 2) {
            lds_dma16(a_tile + kb, aoff[half][0], dst);
            if constexpr (MS == 2) lds_dma16(a_tile + kb, aoff[half][MS - 1], dst + 8192);
        } else {
            lds_dma16(w_tile + kb, boff[half - 2][0], dst);
            lds_dma16(w_tile + kb, boff[half - 2][1], dst + 8192);
        }
    };

    // ---- fragment read offsets
    const int l31 = lane & 31, h = lane >> 5;
    const int swz = (l31 >> 1) & 7;
    int coff[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) coff[s] = ((2 * s + h) ^ swz) << 4;
    const int a_row_off = (wm * (32 * MS) + l31) * 128;            // + a * kHalfBytes + ms * 4096
    const int b_row_off = 2 * kHalfBytes + (wn * 32 + l31) * 128;  // + b * kHalfBytes

    // accumulators start from the bias: register e of acc[b][*] is column n0 + wn * 64 + b * 32 + 8 (e >> 2) + 4 h + (e & 3)
    auto init_acc = [&](f32x16 (&acc_)[2][2 * MS], int tile) {
        const int tn_ = tile % p.n_tiles_n;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            f32x16 bvec;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 t4 = *reinterpret_cast<const float4*>(p.bias + tn_ * BN + wn * 64 + b * 32 + 8 * g + 4 * (lane >> 5));
                bvec[4 * g] = t4.x; bvec[4 * g + 1] = t4.y; bvec[4 * g + 2] = t4.z; bvec[4 * g + 3] = t4.w;
            }
#pragma unroll
            for (int ms = 0; ms < 2 * MS; ++ms) acc_[b][ms] = bvec;
        }
    };

    // EPI_F32_NORM: (mean, rstd) of the A operand's rows that this lane stores in the epilogue (row ms * 32 + (lane >> 3)
    // + 8 i of the wave's share).  At half height they are requested when the tile begins, so that the loads fly behind
    // the main loop; at full height the main loop has no registers to spare and they are requested in front of the
    // epilogue.
    float rsum[2 * MS][4], rsq[2 * MS][4];
    auto gather_stats = [&](int tile) {
        if constexpr (EPI == EPI_F32_NORM) {
            const int tm_ = tile / p.n_tiles_n;
#pragma unroll
            for (int ms = 0; ms < 2 * MS; ++ms)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int m = tm_ * BM + wm * (64 * MS) + ms * 32 + (lane >> 3) + 8 * i;
                    const float2 t2 = reinterpret_cast<const float2*>(p.stat_in)[m < p.M ? m : p.M - 1];
                    rsum[ms][i] = t2.x;
                    rsq[ms][i] = t2.y;
                }
        }
    };
    if constexpr (MS == 1) gather_stats(first_tile);

    u16x8 ra[MS][4];        // activation fragments [ms][s] of the current A half
    u16x8 rb[2][4];         // weight fragments [b][s]: rb[0] always holds B0, rb[1] always holds B1
    f32x16 acc[2][2 * MS];  // [b][a * MS + ms]
    init_acc(acc, first_tile);

    auto read_a = [&](int a, int slot, int ms, int s) {
        ra[ms][s] = *reinterpret_cast<const u16x8*>(smem + slot * kSlotBytes + a * kHalfBytes + a_row_off + ms * 4096 + coff[s]);
    };
    auto read_b = [&](int b, int slot) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
            rb[b][s] = *reinterpret_cast<const u16x8*>(smem + slot * kSlotBytes + b * kHalfBytes + b_row_off + coff[s]);
    };

    // One phase: 4 MS MFMAs of quadrant (A_, B_).  ROLL_A >= 0: refill ra with half ROLL_A of ROLL_SLOT as it is
    // consumed.  FREE_B >= 0: load rb[FREE_B] (idle in this phase) from FREE_SLOT.  STAGE: issue that half of
    // the K-tile two ahead (if it exists).
#define PHASE(A_, B_, STAGE_HALF, STAGE_SLOT, FREE_B, FREE_SLOT, ROLL_A, ROLL_SLOT, SYNC, STEADY)            \
    do {                                                                                                    \
        if (STEADY) stage(STAGE_HALF, STAGE_SLOT);                                           \
        if (FREE_B >= 0) read_b(FREE_B < 0 ? 0 : FREE_B, FREE_SLOT);                            \
        _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                     \
            _Pragma("unroll") for (int ms = 0; ms < MS; ++ms) {                                             \
                acc[B_][(A_) * MS + ms] = mfma32<F16>(rb[B_][s], ra[ms][s], acc[B_][(A_) * MS + ms]);       \
                if (ROLL_A >= 0) read_a(ROLL_A < 0 ? 0 : ROLL_A, ROLL_SLOT, ms, s);             \
            }                                                                                               \
        }                                                                                                   \
        if (SYNC) PHASE_SYNC(STEADY);                                                         \
    } while (0)

    // ---- prologue: the first two K-tiles in flight, then the first fragments
    stage_set_tile();
    stage(0, 0); stage(2, 0); stage(3, 0); stage(1, 0);
    stage_advance();
    stage(0, 1); stage(3, 1); stage(2, 1); stage(1, 1);
    stage_advance();
    PHASE_SYNC(true);  // A0, B0, B1 of K-tile 0 landed (five half-tiles may still fly)
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int ms = 0; ms < MS; ++ms) read_a(0, 0, ms, s);
    read_b(0, 0);

    int cur_tile = first_tile;
    int kt_in_tile = 0;
    for (int g = 0; g < n_kt; g += 2) {
        const bool st0 = st_left > 0;  // K-tile g + 2 exists: its halves are issued during K-tile g
        //     use    stage(half, slot)   free read    rolling A     sync
        PHASE(0, 0,   0, 0,               1, 0,        -1, 0,        true, st0);
        PHASE(0, 1,   2, 0,              -1, 0,         1, 0,        false, st0);
        PHASE(1, 1,   3, 0,              -1, 0,        -1, 0,        true, st0);
        PHASE(1, 0,   1, 0,               1, 1,         0, 1,        true, st0);
        if (st0) stage_advance();
        const bool st1 = st_left > 0;  // K-tile g + 3
        PHASE(0, 1,   0, 1,               0, 1,        -1, 0,        true, st1);
        PHASE(0, 0,   3, 1,              -1, 0,         1, 1,        false, st1);
        PHASE(1, 0,   2, 1,              -1, 0,        -1, 0,        true, st1);
        PHASE(1, 1,   1, 1,               0, 0,         0, 0,        true, st1);
        if (st1) stage_advance();
        kt_in_tile += 2;
        if (kt_in_tile == nk) {  // this output tile is complete: store it while the next tile's loads fly
            const int tm = cur_tile / p.n_tiles_n, tn = cur_tile - tm * p.n_tiles_n;
            if constexpr (MS == 2) gather_stats(cur_tile);
            gemm_epilogue<EPI, F16, MS>(p, acc, tm * BM, tn * BN, wm, wn, lane, smem + kRingBytes + wave * 4096, rsum, rsq);
            kt_in_tile = 0;
            cur_tile += wgs_of_xcd;
            if (g + 2 < n_kt) {
                init_acc(acc, cur_tile);
                if constexpr (MS == 1) gather_stats(cur_tile);
            }
        }
    }
#undef PHASE
}

// 32 rows per workgroup; thread (q, r) adds strips q, q + 8, .. of row r (a wave reads runs of 256 bytes), the eight
// partial sums of a row meet in LDS
__global__ __launch_bounds__(256) void row_stats_finish_kernel(const float2* __restrict__ partial, int n_strips, int M, float eps,
                                                               float2* __restrict__ row_norm) {
    __shared__ float2 part[8][32];
    const int r = threadIdx.x & 31, q = threadIdx.x >> 5;
    const int m = blockIdx.x * 32 + r;
    const int mc = m < M ? m : M - 1;
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll 4
    for (int j = q; j < n_strips; j += 8) {
        const float2 t = partial[static_cast<int64_t>(j) * M + mc];
        s1 += t.x;
        s2 += t.y;
    }
    part[q][r] = make_float2(s1, s2);
    __syncthreads();
    if (q == 0 && m < M) {
#pragma unroll
        for (int k = 1; k < 8; ++k) {
            s1 += part[k][r].x;
            s2 += part[k][r].y;
        }
        const float inv = 1.0f / static_cast<float>(64 * n_strips);
        const float mean = s1 * inv;
        row_norm[m] = make_float2(mean, 1.0f / sqrtf(fmaxf(s2 * inv - mean * mean, 0.0f) + eps));
    }
}

int32_t launch_row_stats_finish(asrfe_ctx* ctx, const float* partial, int n_strips, int M, float eps, float* row_norm, hipStream_t s) {
    if (M <= 0) return ASRFE_OK;
    hipLaunchKernelGGL(row_stats_finish_kernel, dim3((M + 31) / 32), dim3(256), 0, s, reinterpret_cast<const float2*>(partial),
                       n_strips, M, eps, reinterpret_cast<float2*>(row_norm));
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_err(ctx, ASRFE_ERR_HIP, std::string("row_stats_finish launch: ") + hipGetErrorString(e));
    return ASRFE_OK;
}

// The dynamic-LDS limit of a kernel is a per-device attribute: asrfe_create calls this with the context's device
// current, before any compute call can run.
int32_t prepare_gemm(asrfe_ctx* ctx) {
    const void* fns[24] = {reinterpret_cast<const void*>(gemm_bf16_kernel<EPI_RELU_STAT, false, 2>),
                           reinterpret_cast<const void*>(gemm_bf16_kernel<EPI_RELU_STAT, true, 2>),
                           reinterpret_cast<const void*>(gemm_bf16_kernel<EPI_RELU_STAT, false, 1>),
                           reinterpret_cast<const void*>(gemm_bf16_kernel<EPI_RELU_STAT, true, 1>),
                           reinterpret_cast<const void*>(gemm_bf16_kernel<EPI_F32_NORM, false, 2>),
                           reinterpret_cast<const void*>(gemm_bf16_kernel<EPI_F32_NORM, true, 2>),
                           reinterpret_cast<const void*>(gemm_bf16_kernel<EPI_F32_NORM, false, 1>),
                           reinterpret_cast<const void*>(gemm_bf16_kernel<EPI_F32_NORM, true, 1>),
                           reinterpret_cast<const void*>(gemm_bf16_kernel<EPI_ARGMAX, false, 2>),
                           reinterpret_cast<const void*>(gemm_bf16_kernel<EPI_ARGMAX, true, 2>),
                           reinterpret_cast<const void*>(gemm_bf16_kernel<EPI_ARGMAX, false, 1>),
                           reinterpret_cast<const void*>(gemm_bf16_kernel<EPI_ARGMAX, true, 1>),
                           reinterpret_cast<const void*>(gemm_bf16_kernel<EPI_BF16, false, 2>),
                           reinterpret_cast<const void*>(gemm_bf16_kernel<EPI_BF16_RELU, false, 2>),
                           reinterpret_cast<const void*>(gemm_bf16_kernel<EPI_F32_ADD, false, 2>),
                           reinterpret_cast<const void*>(gemm_bf16_kernel<EPI_BF16, true, 2>),
                           reinterpret_cast<const void*>(gemm_bf16_kernel<EPI_BF16_RELU, true, 2>),
                           reinterpret_cast<const void*>(gemm_bf16_kernel<EPI_F32_ADD, true, 2>),
                           reinterpret_cast<const void*>(gemm_bf16_kernel<EPI_BF16, false, 1>),
                           reinterpret_cast<const void*>(gemm_bf16_kernel<EPI_BF16_RELU, false, 1>),
                           reinterpret_cast<const void*>(gemm_bf16_kernel<EPI_F32_ADD, false, 1>),
                           reinterpret_cast<const void*>(gemm_bf16_kernel<EPI_BF16, true, 1>),
                           reinterpret_cast<const void*>(gemm_bf16_kernel<EPI_BF16_RELU, true, 1>),
                           reinterpret_cast<const void*>(gemm_bf16_kernel<EPI_F32_ADD, true, 1>)};
    for (const void* f : fns)
        if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, kGemmLds) != hipSuccess)
            return set_err(ctx, ASRFE_ERR_HIP, "gemm: cannot raise the dynamic LDS limit to 160 KiB");
    return ASRFE_OK;
}

int32_t launch_gemm(asrfe_ctx* ctx, const GemmParams& pin, int epi, hipStream_t s) {
    GemmParams p = pin;
    if (p.M <= 0) return ASRFE_OK;
    if (p.N % BN != 0 || p.K % (2 * BK) != 0 || p.K <= 0 || p.lda % 8 != 0 || p.ldw % 8 != 0 || (p.ldo % 8 != 0 && epi != EPI_ARGMAX))
        return set_err(ctx, ASRFE_ERR_SHAPE, "gemm: N must be a multiple of 256, K of 128, leading dims of 8");
    if ((epi == EPI_RELU_STAT && !p.stat_out) || (epi == EPI_F32_NORM && (!p.stat_in || !p.colsum || !p.post_bias)))
        return set_err(ctx, ASRFE_ERR_BAD_ARG, "gemm: row statistics need their buffers");
    if (epi == EPI_ARGMAX && p.ldo < p.N / 64) return set_err(ctx, ASRFE_ERR_SHAPE, "gemm: a row of the arg-max output holds N / 64 pairs");
    p.n_tiles_n = p.N / BN;
    // Half-height tiles when they still fit the chip in one round and full-height tiles would leave CUs idle
    // (flags bits 8-9 force a height: 1 full, 2 half).
    const int tiles_full = (p.M + 255) / 256 * p.n_tiles_n, tiles_half = (p.M + 127) / 128 * p.n_tiles_n;
    const int force = (p.flags >> 8) & 3;
    const bool half = force == 2 || (force == 0 && tiles_half > tiles_full && tiles_half <= ctx->num_cus);
    p.n_tiles_m = half ? (p.M + 127) / 128 : (p.M + 255) / 256;
    const int n_tiles = p.n_tiles_m * p.n_tiles_n;
    dim3 grid(n_tiles < ctx->num_cus ? n_tiles : ctx->num_cus), block(kGemmThreads);
    const bool f16 = ctx->cfg.enc_compute_dtype == ASRFE_DT_F16;
#define LAUNCH_GEMM_H(E, MS_)                                                                       \
    do {                                                                                            \
        if (f16) hipLaunchKernelGGL((gemm_bf16_kernel<E, true, MS_>), grid, block, kGemmLds, s, p);  \
        else hipLaunchKernelGGL((gemm_bf16_kernel<E, false, MS_>), grid, block, kGemmLds, s, p);     \
    } while (0)
#define LAUNCH_GEMM(E)                   \
    do {                                 \
        if (half) LAUNCH_GEMM_H(E, 1);   \
        else LAUNCH_GEMM_H(E, 2);        \
    } while (0)
    switch (epi) {
        case EPI_BF16: LAUNCH_GEMM(EPI_BF16); break;
        case EPI_BF16_RELU: LAUNCH_GEMM(EPI_BF16_RELU); break;
        case EPI_F32_ADD: LAUNCH_GEMM(EPI_F32_ADD); break;
        case EPI_ARGMAX: LAUNCH_GEMM(EPI_ARGMAX); break;
        case EPI_RELU_STAT: LAUNCH_GEMM(EPI_RELU_STAT); break;
        case EPI_F32_NORM: LAUNCH_GEMM(EPI_F32_NORM); break;
        default: return set_err(ctx, ASRFE_ERR_BAD_ARG, "gemm: bad epilogue");
    }
#undef LAUNCH_GEMM
#undef LAUNCH_GEMM_H
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_err(ctx, ASRFE_ERR_HIP, std::string("gemm launch: ") + hipGetErrorString(e));
    return ASRFE_OK;
}

}  // namespace asrfe
