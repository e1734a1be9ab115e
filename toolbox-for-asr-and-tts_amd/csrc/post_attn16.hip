// The post-attention kernel (post_attn.hip: output projection + FSMN + residual, feed-forward, next LayerNorm and the
// next layer's QKV projection in one launch) on v_mfma_f32_16x16x32: the shape that does the same work with less energy
// per FLOP on gfx950, which is what bounds a kernel whose MFMA phases run on a power-limited clock.
//
// Same workgroup (128 rows, 4 waves, one per SIMD, a wave owns 32 complete rows), same weight stream geometry
// (16 KiB pieces, 8-slot ring, LDS-DMA, counted vmcnt waits: see post_attn.hip), same residual-stream handling (tiled
// form only: this kernel serves the throughput plan).  What differs is every register layout.  Lane l = 16 q + c:
//   - an MFMA multiplies a weight fragment (A: 16 weight rows x 32 k; lane holds row c, k = 8 q .. 8 q + 7) by an
//     activation fragment (B: 32 k x 16 rows; lane holds row c of the HALF mh of the wave's 32 rows, same k) into a
//     16 x 16 tile of four registers: register e = output column 4 q + e of row 16 mh + c.  A 1 KiB weight fragment
//     read from LDS feeds two MFMAs (mh = 0, 1);
//   - the 32 x 512 fp32 row tile is 32 column tiles x 2 halves of four accumulation registers (256 AGPRs): tile (j, mh)
//     holds columns 16 j + 4 q + e.  A row lives in the four lanes c, c + 16, c + 32, c + 48;
//   - operand fragments made from accumulators (n(x_mid), relu(H), n(x_out)) take element i < 4 from tile 2 s and
//     i >= 4 from tile 2 s + 1 of k-slice s: lane q, element i is column 32 s + 16 (i >> 2) + 4 q + (i & 3).  The host
//     stores the columns of W1, W2 and Wqkv in that order inside every block of 32 (asrfe_load_encoder_weights);
//   - the tiled residual stream: per 32-row group, 32-column block nt, four runs of 1 KiB: run 2 a + mh holds tile
//     (2 nt + a, mh), lane l its four columns.
#include <algorithm>
#include <utility>

#include "common.h"
#include "elem.h"
#include "encoder.h"
#include "lds_dma.h"

namespace asrfe {
namespace {

#ifndef ASRFE_FFN_TIMING
#define ASRFE_FFN_TIMING 0
#endif
#ifndef ASRFE_FFN_ABLATE
#define ASRFE_FFN_ABLATE 0
#endif

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

constexpr int kFfnThreads = 256;
constexpr int kFfnRows = 128;
constexpr int kPieceBytes = 16384;
constexpr int kFfnRing = 8 * kPieceBytes;  // 128 KiB; the fp32 b1, b2 and b_o vectors follow it
constexpr int kSlabFloats = 4096;          // four wave-private 4 KiB output slabs

// all but the N youngest vector-memory operations of this wave are done, then every wave of the workgroup meets
template <int N>
__device__ __forceinline__ void wait_vm_barrier() {
#if ASRFE_FFN_TIMING == 2  // timing experiment: no workgroup barrier (wrong results)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#else
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
#endif
}

__device__ __forceinline__ void ffn_wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// The MFMAs are issued through inline assembly so that the register class of each accumulator is fixed: the row tile
// fills the 256 accumulation registers (AGPRs), the hidden-chunk tiles stay in architectural VGPRs.
template <bool F16>
__device__ __forceinline__ void mfma_agpr(f32x4& c, const u16x8 a, const u16x8 b) {
    if constexpr (F16) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
template <bool F16>
__device__ __forceinline__ void mfma_vgpr(f32x4& c, const u16x8 a, const u16x8 b) {
    if constexpr (F16) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}

// Moves one tile between the accumulation registers and working VGPRs at a fixed point of the program.
__device__ __forceinline__ f32x4 tile_out(f32x4& acc) {
    asm volatile("" : "+a"(acc));
    f32x4 t = acc;
    asm volatile("" : "+v"(t));
    return t;
}
__device__ __forceinline__ void tile_in(f32x4& acc, const f32x4 t) {
    acc = t;
    asm volatile("" : "+a"(acc));
}

typedef float v2f __attribute__((ext_vector_type(2)));

// 16 bytes per lane from sbase + voff + OFF into registers.  Issued through inline assembly like the LDS-DMA loads:
// the kernel counts vector-memory operations itself (they complete in issue order), and a load the compiler could
// see would make it insert waits that know nothing of the DMA loads in flight behind it.  The destination holds
// the data only after a wait that names it (wait_vm_regs / wait_vm_barrier_regs below).
template <int OFF>
__device__ __forceinline__ void gload16(f32x4& dst, const void* sbase, uint32_t voff) {
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm_regs(f32x4 (&r)[4]) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm_barrier_regs(f32x4 (&r)[4]) {
    asm volatile("s_waitcnt vmcnt(%4)\n\ts_barrier" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : "n"(N) : "memory");
}

// ---- vector-memory bookkeeping of the output projection when the residual stream arrives beside it (TILED) ----
// Before the pass the wave has issued weight pieces 0..7, in that order.  Piece i >= 1 of the pass issues piece i + 7
// (four DMA loads, at MFMAs 0, 2, 4, 6 of its second half); an even piece i <= 24 also issues the four loads of a
// residual tile (at MFMAs 1, 3, 5, 7): the tile that piece i + 6 adds.  In the middle of piece i the wave needs piece
// i + 1 and, when i is even, the residual tile it adds there (loaded by piece i - 6, or before the pass).
// Even piece 2 k adds tile 4 + k for k < 12 (a tile of the NEXT column quarter, not yet started) and tile k - 12
// after that (the first quarter's tiles, long finished by then).
constexpr int xtile_of(int k) { return k < 12 ? 4 + k : k - 12; }
constexpr int xload_piece(int j) { return (j >= 0 && j <= 24 && j % 2 == 0) ? 1 : 0; }
// vector-memory operations piece j of the pass issues
constexpr int out_proj_ops(int j) { return (j >= 1 ? 4 : 0) + 4 * xload_piece(j); }
// operations younger than the last DMA load of piece i + 1 at the middle of piece i
constexpr int out_proj_vmcnt(int i) {
    int n = 0;
    if (i + 1 <= 7) {
        n = 4 * (7 - (i + 1));
        for (int j = 0; j < i; ++j) n += out_proj_ops(j);
    } else {
        n = xload_piece(i - 6);  // (its last load follows the last DMA load of piece i + 1)
        for (int j = i - 5; j < i; ++j) n += out_proj_ops(j);
    }
    return n;
}
// ... and younger than both that and the last load of the residual tile that even piece i adds: the tile was
// requested by piece i - 6 as its last operation (before the pass for i < 6, behind every weight piece up to 7 but
// in front of every slab, so it has long arrived)
constexpr int out_proj_vmcnt_x(int i) {
    if (i < 6) return out_proj_vmcnt(i);
    int n = 0;
    for (int j = i - 5; j < i; ++j) n += out_proj_ops(j);
    return n < out_proj_vmcnt(i) ? n : out_proj_vmcnt(i);
}

// Row statistics of a lane's 128 values of one row (32 tiles of 4): sums of (x - k) and (x - k)^2 with k = the mean
// of the lane's first tile, two values per packed instruction.  The shift keeps the variance formula free of
// cancellation when a row's mean is large against its spread.
struct RowStat {
    v2f s1, s2, nk2;
    float k;
};
__device__ __forceinline__ void stat_add_tile(RowStat& r, const int j, const f32x4& t) {
    if (j == 0) {  // j is a constant after unrolling
        r.k = ((t[0] + t[1]) + (t[2] + t[3])) * 0.25f;
        r.nk2 = v2f{-r.k, -r.k};
        r.s1 = v2f{0.0f, 0.0f};
        r.s2 = v2f{0.0f, 0.0f};
    }
#pragma unroll
    for (int e = 0; e < 4; e += 2) {
        const v2f d = v2f{t[e], t[e + 1]} + r.nk2;
        r.s1 += d;
        r.s2 = __builtin_elementwise_fma(d, d, r.s2);
    }
}
// -> mean and sum of squared deviations of the lane's 128 values
__device__ __forceinline__ void stat_lane(const RowStat& r, float& mean, float& m2) {
    const float s1 = r.s1.x + r.s1.y, s2 = r.s2.x + r.s2.y;
    mean = r.k + s1 * (1.0f / 128.0f);
    m2 = s2 - s1 * s1 * (1.0f / 128.0f);
}
// the four lanes of a row (128 values each) -> mean and 1 / sqrt(var + eps) of the 512
__device__ __forceinline__ void stat_finish_row(float& mean, float m2, float eps, float& rstd) {
    float mo = __shfl_xor(mean, 16), qo = __shfl_xor(m2, 16);
    float d = mo - mean;
    m2 = m2 + qo + d * d * 64.0f;
    mean = 0.5f * (mean + mo);
    mo = __shfl_xor(mean, 32);
    qo = __shfl_xor(m2, 32);
    d = mo - mean;
    m2 = m2 + qo + d * d * 128.0f;
    mean = 0.5f * (mean + mo);
    rstd = 1.0f / sqrtf(m2 * (1.0f / 512.0f) + eps);
}

__device__ __forceinline__ u16x8 relu16(const u16x8 v) {
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    return __builtin_bit_cast(u16x8, __builtin_elementwise_max(__builtin_bit_cast(s16x8, v), z));
}

template <bool F16>
__device__ __forceinline__ u16x8 pack8(float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7) {
    const u32x4 v = {pack_elem2<F16>(a0, a1), pack_elem2<F16>(a2, a3), pack_elem2<F16>(a4, a5), pack_elem2<F16>(a6, a7)};
    return __builtin_bit_cast(u16x8, v);
}


// OUT: 0 = x_out only, row-major (the encoder's last layer), 2 = x_out tiled and the next layer's qkv.  The residual
// stream x always arrives tiled (or not at all: layer 0).  No residual tile is waited for before the output projection
// starts: they arrive while it runs and are added between its MFMAs, twelve 32-column blocks to column quarters that
// have not started yet, the first quarter's four at the end.
template <bool F16, int OUT>
__global__ __launch_bounds__(kFfnThreads) void post_attn16_kernel(const FfnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    saturate_fp16_conversions<F16>();
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, q = lane >> 4;
    const int row0 = blockIdx.x * kFfnRows + wave * 32;
    const int nch = p.ff >> 6;
    // the bias vectors live in LDS: a global load per use would be waited for one by one.  The b1 area is at least
    // 16 KiB: after the feed-forward it holds the waves' output slabs (OUT 2).
    float* b1s = reinterpret_cast<float*>(smem + kFfnRing);
    float* b2s = b1s + max(p.ff, kSlabFloats);
    float* bos = b2s + 512;
    float* bqs = bos + 512;  // b_qkv of the next layer, 1536 (OUT 2)

#if ASRFE_FFN_TIMING
    int trace_n = 0;
    const bool tracing = (p.flags & 256) && p.trace && blockIdx.x == 100 && wave == 0;
#define FFN_STAMP() do { if (tracing) { const long long tsc = __builtin_readcyclecounter(); if (lane == 0) p.trace[trace_n] = tsc; ++trace_n; } } while (0)
#else
#define FFN_STAMP() do {} while (0)
#endif
    FFN_STAMP();
    // ---- staging: the weights arrive as a stream of ready-made 16 KiB LDS images, so a wave copies four
    //      contiguous 1 KiB runs per piece
    const char* wsrc = reinterpret_cast<const char*>(p.wstream);
    const uint32_t wlane = wave * 1024 + lane * 16;
    const uint32_t smem_base = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    const bool no_store = p.flags & 1;  // timing experiments only
    uint32_t wq[4] = {wlane, wlane + 4096, wlane + 8192, wlane + 12288};
    asm volatile("" : "+v"(wq[0]), "+v"(wq[1]), "+v"(wq[2]), "+v"(wq[3]));
    auto stage_quarter = [&](auto rc, const char* cbase, int i) {
        constexpr int R = decltype(rc)::value;
        const uint32_t dst = __builtin_amdgcn_readfirstlane(smem_base + R * kPieceBytes + wave * 1024);
        lds_dma16(cbase + R * kPieceBytes, wq[i], dst + i * 4096);
    };
    // the same in two parts around a pair of MFMAs (which are the M0 wait state: no s_nop in the loop)
    auto stage_quarter_m0 = [&](auto rc, int i) {
        constexpr int R = decltype(rc)::value;
        lds_dma_set_m0(__builtin_amdgcn_readfirstlane(smem_base + R * kPieceBytes + wave * 1024) + i * 4096);
    };
    auto stage_quarter_load = [&](auto rc, const char* cbase, int i) {
        constexpr int R = decltype(rc)::value;
        lds_dma16_at_m0(cbase + R * kPieceBytes, wq[i]);
    };
    // chunk c of the layer's stream: 0..3 are Wo, 4 + j is hidden chunk j
    auto chunk_base = [&](int ch) { return wsrc + static_cast<int64_t>(ch) * (8 * kPieceBytes); };
    auto stage = [&](auto rc, const char* cbase) {
#pragma unroll
        for (int i = 0; i < 4; ++i) stage_quarter(rc, cbase, i);
    };
    // the first pieces fly while the activations are loaded; slots 2..7 stage activations until then
    const char* cb_first = chunk_base(0);
    stage(IC<0>{}, cb_first); stage(IC<1>{}, cb_first);

    for (int i = tid * 4; i < p.ff; i += kFfnThreads * 4)
        *reinterpret_cast<float4*>(b1s + i) = *reinterpret_cast<const float4*>(p.b1 + i);
    if (tid < 128) *reinterpret_cast<float4*>(b2s + tid * 4) = *reinterpret_cast<const float4*>(p.b2 + tid * 4);
    else *reinterpret_cast<float4*>(bos + (tid - 128) * 4) = *reinterpret_cast<const float4*>(p.bo + (tid - 128) * 4);
    if constexpr (OUT == 2) {
        for (int i = tid * 4; i < 1536; i += kFfnThreads * 4)
            *reinterpret_cast<float4*>(bqs + i) = *reinterpret_cast<const float4*>(p.bqkv + i);
    }
    __syncthreads();

    // ---- activations.  att and fsmn come in by LDS-DMA (no staging registers) as 4 KiB slabs of 32 rows x 128 bytes,
    //      filled by four loads whose per-lane source addresses carry the bank swizzle; the residual stream does not pass
    //      through here at all.  A wave's slab slot j is the part of ring slot 2 + j that the same wave later fills with
    //      its share of weight piece 2 + j (the four 1 KiB runs at wave * 1 KiB + k * 4 KiB), so it can stage that piece
    //      as soon as it has read its own last slab there, without asking the other waves.  Six slots: five slabs in
    //      flight behind the one being read.  Slab I uses slot (I + 2) % 6, which makes the last six slabs free slots
    //      0..5 in order: pieces 2..7 are requested in stream order.
    f32x4 oacc[32][2];   // the row tile: [column tile j][half mh]
    u16x8 afrag[16][2];  // att, later n(x_mid) / n(x_out), as operand fragments: [k-slice of 32][half]
    const bool has_x = p.x != nullptr;  // layer 0 has no residual into the attention block: x counts as zero
    // three register sets that the residual blocks rotate through: set k % 3 for the block that piece 2 k adds
    f32x4 xr[3][4];
    const uint32_t lane16 = lane * 16;
    // the wave's 32-row group of the tiled residual stream (64 KiB); layer 0 reads (and ignores) the first 64 KiB of
    // the weight stream instead, so that the count of loads in flight is the same in every layer
    const char* xtile = has_x ? reinterpret_cast<const char*>(p.x) + static_cast<int64_t>(row0 >> 5) * 65536 : wsrc;
    auto xload = [&](auto setc, int nt, auto gc) {
        constexpr int S = decltype(setc)::value, G = decltype(gc)::value;
        gload16<G * 1024>(xr[S][G], xtile + nt * 4096, lane16);
    };
    {
        static_for<3>([&](auto sc) {
            static_for<4>([&](auto gc) { xload(sc, xtile_of(decltype(sc)::value), gc); });
        });
        const int rbase = min(row0, p.M - 1);
        const char* asrc = reinterpret_cast<const char*>(p.att) + static_cast<int64_t>(rbase) * 1024;
        const char* fsrc = reinterpret_cast<const char*>(p.fsmn) + static_cast<int64_t>(rbase) * 1024;
        uint32_t aoff[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = 8 * k + (lane >> 3);
            const int rr = min(row0 + r, p.M - 1) - rbase;
            aoff[k] = static_cast<uint32_t>(rr * 1024 + (((lane & 7) ^ (r & 7)) << 4));
        }
        const uint32_t sl_lds = __builtin_amdgcn_readfirstlane(smem_base + 2 * kPieceBytes + wave * 1024);
        // slabs 0..7: att k-slices of 64, slabs 8..15: fsmn column groups of 64; rows 8 k .. 8 k + 7 of a slab sit at
        // k * 4 KiB, 16-byte chunk ch of row r at chunk ch ^ (r & 7)
        auto slab_dma = [&](auto ic) {
            constexpr int I = decltype(ic)::value;
            const uint32_t dst = sl_lds + ((I + 2) % 6) * kPieceBytes;
            const char* src = I < 8 ? asrc + I * 128 : fsrc + (I - 8) * 128;
#pragma unroll
            for (int k = 0; k < 4; ++k) lds_dma16(src, aoff[k], dst + k * 4096);
        };
        static_for<6>([&](auto ic) { slab_dma(ic); });
        // this lane's row 16 mh + c of a slab: (2 mh + (c >> 3)) * 4 KiB + (c & 7) * 128
        const int srow = (c >> 3) * 4096 + (c & 7) * 128, ssw = c & 7;
        static_for<16>([&](auto ic) {
            constexpr int I = decltype(ic)::value;
            // behind slab I in the queue: the slabs up to I + 5 and the weight pieces requested so far, 20 loads in
            // every iteration
            asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
            const char* sl = smem + (2 + (I + 2) % 6) * kPieceBytes + wave * 1024 + srow;
            if constexpr (I < 8) {
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int mh = 0; mh < 2; ++mh)
                        afrag[2 * I + s][mh] = *reinterpret_cast<const u16x8*>(sl + mh * 8192 + (((4 * s + q) ^ ssw) << 4));
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(afrag[2 * I][0]), "+v"(afrag[2 * I][1]), "+v"(afrag[2 * I + 1][0]), "+v"(afrag[2 * I + 1][1]) : : "memory");
            } else {
                constexpr int P = I - 8;
                uint2 fs[2][4];  // [mh][jj]: the four values of tile (4 P + jj, mh)
#pragma unroll
                for (int mh = 0; mh < 2; ++mh)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj)
                        fs[mh][jj] = *reinterpret_cast<const uint2*>(sl + mh * 8192 + (((2 * jj + (q >> 1)) ^ ssw) << 4) + 8 * (q & 1));
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fs[0][0]), "+v"(fs[0][1]), "+v"(fs[0][2]), "+v"(fs[0][3]), "+v"(fs[1][0]), "+v"(fs[1][1]), "+v"(fs[1][2]), "+v"(fs[1][3]) : : "memory");
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const float4 bo4 = *reinterpret_cast<const float4*>(bos + 64 * P + 16 * jj + 4 * q);
#pragma unroll
                    for (int mh = 0; mh < 2; ++mh) {
                        float f0, f1, f2, f3;
                        unpack_elem2<F16>(fs[mh][jj].x, f0, f1);
                        unpack_elem2<F16>(fs[mh][jj].y, f2, f3);
                        const f32x4 t = {f0 + bo4.x, f1 + bo4.y, f2 + bo4.z, f3 + bo4.w};
                        tile_in(oacc[4 * P + jj][mh], t);
                    }
                }
            }
            if constexpr (I + 6 < 16) slab_dma(IC<I + 6>{});
            else stage(IC<I - 8>{}, cb_first);  // the slot just read takes this wave's share of weight piece I - 8
        });
    }

    // ---- fragment reads: kind 1 = W1 piece (64 rows of 256 bytes), kind 2 = Wo / W2 / Wqkv piece (128 rows of 128 bytes).
    //      Fragment (row group, k-slice of 32): lane reads row 16 group + c, 16-byte chunk 4 slice + q of the row.
    int rb1 = c * 256, sw1 = (q ^ c) << 4;                   // kind 1: chunk ch of row r sits at ch ^ (r & 15)
    int rb2 = c * 128, sw2 = (q ^ ((c >> 1) & 7)) << 4;      // kind 2: at ch ^ ((r >> 1) & 7)
    auto frag1 = [&](auto rc, auto fc) -> u16x8 {  // W1 rows 16 jj + c, k-slice ks of the piece's four
        constexpr int R = decltype(rc)::value, F = decltype(fc)::value;
        constexpr int jj = F & 3, ks = F >> 2;
        return *reinterpret_cast<const u16x8*>(smem + R * kPieceBytes + jj * 4096 + rb1 + (sw1 ^ (ks << 6)));
    };
    // Kind 2 addresses are one of four per-lane bases (k-slice kk; ring half: an LDS read's immediate offset ends at
    // 64 KiB) plus a compile-time offset.  The bases are made opaque where they are formed (frag2_bases), otherwise
    // the compiler re-associates the sum and spends a vector add on every fragment.
    int fb2[2][2];
    auto frag2_bases = [&]() {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            fb2[0][kk] = rb2 + (sw2 ^ (kk << 6));
            fb2[1][kk] = fb2[0][kk] + 4 * kPieceBytes;
            asm volatile("" : "+v"(fb2[0][kk]), "+v"(fb2[1][kk]));
        }
    };
    auto frag2 = [&](auto rc, auto fc) -> u16x8 {  // rows 16 jl + c, k-slice kk of the piece's two
        constexpr int R = decltype(rc)::value, F = decltype(fc)::value;
        constexpr int kk = F >> 3, jl = F & 7;
        return *reinterpret_cast<const u16x8*>(smem + fb2[R >> 2][kk] + (R & 3) * kPieceBytes + jl * 2048);
    };
    u16x8 wf[8];     // rolling weight fragments

    wait_vm_barrier<24>();  // piece 0 has landed everywhere

    // ---- a projection pass: the row tile += afrag W^T for 512 output columns, 4 chunks of 8 pieces starting at
    //      chunk c0 of the stream, whose piece 0 every wave has seen land.  A piece is 128 output columns x 64 k:
    //      16 fragments (column tile jl, k-slice kk), each feeding the two halves of the wave's rows.  TO_W1: W1 pieces
    //      follow the pass and their first fragments are read ahead.  FIRST: the output projection, whose chunk is a
    //      quarter of the output columns over all of k (the others: a quarter of k over all columns, two k-slices of
    //      64 per chunk); every even piece of it also adds a residual block in its middle and requests the block that
    //      the piece six further on adds (out_proj_vmcnt above has the schedule).
    auto proj_pass = [&](int c0, auto to_w1, auto first) {
        constexpr bool TO_W1 = decltype(to_w1)::value;
        constexpr bool FIRST = decltype(first)::value;
        asm volatile("" : "+v"(rb2), "+v"(sw2));  // see chunk() below
        frag2_bases();
        static_for<8>([&](auto f) { wf[decltype(f)::value] = frag2(IC<0>{}, f); });
        static_for<4>([&](auto cc) {
            constexpr int C = decltype(cc)::value;
            const char* cb0 = chunk_base(c0 + C);
            const char* cb1 = chunk_base(c0 + C + 1);
            static_for<8>([&](auto rc) {
                constexpr int R = decltype(rc)::value;
                constexpr int ks = FIRST ? R : 2 * C + (R >> 2), nq = FIRST ? C : (R & 3);
                constexpr int PI = 8 * C + R;                        // piece of the pass
                constexpr bool XADD = FIRST && PI % 2 == 0;
                constexpr bool XLOAD = FIRST && xload_piece(PI);
                constexpr int XSET = (PI / 2) % 3;
                static_for<8>([&](auto f) {
                    constexpr int F = decltype(f)::value;
                    mfma_agpr<F16>(oacc[8 * nq + F][0], wf[F], afrag[2 * ks][0]);
                    mfma_agpr<F16>(oacc[8 * nq + F][1], wf[F], afrag[2 * ks][1]);
                    wf[F] = frag2(rc, IC<F + 8>{});
                });
                if constexpr (XADD) {
                    // block nt of the residual stream = tiles (2 nt + a, mh), register set entry 2 a + mh
                    constexpr int nt = xtile_of(PI / 2);
                    wait_vm_barrier_regs<out_proj_vmcnt_x(PI)>(xr[XSET]);
                    if (has_x) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            f32x4 t = tile_out(oacc[2 * nt + (g >> 1)][g & 1]);
                            t += xr[XSET][g];
                            tile_in(oacc[2 * nt + (g >> 1)][g & 1], t);
                        }
                    }
                } else if constexpr (FIRST) {
                    wait_vm_barrier<out_proj_vmcnt(PI)>();
                } else {
                    wait_vm_barrier<20>();
                }
                static_for<8>([&](auto f) {
                    constexpr int F = decltype(f)::value;
                    // (the first pass finds its piece 7 already requested by the prologue)
                    constexpr bool STAGE = (F & 1) == 0 && !(R == 0 && FIRST && C == 0);
                    if constexpr (STAGE) stage_quarter_m0(IC<(R + 7) & 7>{}, F >> 1);
                    mfma_agpr<F16>(oacc[8 * nq + F][0], wf[F], afrag[2 * ks + 1][0]);
                    mfma_agpr<F16>(oacc[8 * nq + F][1], wf[F], afrag[2 * ks + 1][1]);
                    if constexpr (C == 3 && R == 7) {
                        if constexpr (TO_W1) wf[F] = frag1(IC<0>{}, f);
                    } else {
                        wf[F] = frag2(IC<((R + 1) & 7)>{}, f);
                    }
                    if constexpr ((F & 1) == 0) {
                        if constexpr (STAGE) stage_quarter_load(IC<(R + 7) & 7>{}, R == 0 ? cb0 : cb1, F >> 1);
                    } else if constexpr (XLOAD) {
                        xload(IC<XSET>{}, xtile_of(PI / 2 + 3), IC<(F >> 1)>{});
                    }
                });
            });
        });
    };

    // ---- output projection: the row tile += att Wo^T
    FFN_STAMP();  // prologue done
    proj_pass(0, std::true_type{}, std::true_type{});
    FFN_STAMP();  // output projection done

    // Row statistics of the row tile (both halves), then n(row) as the operand fragments of the next product.  The
    // compiler does not see inside the asm MFMAs, so the wait states between the last of them and the first read of their
    // results are covered by hand (volatile asm statements keep their order).
    float rstd[2], shift[2];
    auto row_stats = [&]() {
#pragma unroll
        for (int mh = 0; mh < 2; ++mh) {
            RowStat rs;
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                const f32x4 t = tile_out(oacc[j][mh]);
                stat_add_tile(rs, j, t);
            }
            float mean = 0.0f, m2 = 0.0f;
            stat_lane(rs, mean, m2);
            stat_finish_row(mean, m2, p.eps, rstd[mh]);
            shift[mh] = -mean * rstd[mh];
        }
    };
    // fragment s of half mh from tiles 2 s and 2 s + 1 (already in working registers)
    auto norm_frag = [&](const f32x4& t0, const f32x4& t1, int mh) -> u16x8 {
        const float r = rstd[mh], sh = shift[mh];
        return pack8<F16>(t0[0] * r + sh, t0[1] * r + sh, t0[2] * r + sh, t0[3] * r + sh,
                          t1[0] * r + sh, t1[1] * r + sh, t1[2] * r + sh, t1[3] * r + sh);
    };

    // ---- x_mid is complete: n(x_mid) as the operand fragments of W1, and b2 into the row tile
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    row_stats();
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const float4 u0 = *reinterpret_cast<const float4*>(b2s + 32 * s + 4 * q);
        const float4 u1 = *reinterpret_cast<const float4*>(b2s + 32 * s + 16 + 4 * q);
#pragma unroll
        for (int mh = 0; mh < 2; ++mh) {
            f32x4 t0 = tile_out(oacc[2 * s][mh]), t1 = tile_out(oacc[2 * s + 1][mh]);
            afrag[s][mh] = norm_frag(t0, t1, mh);
            t0 += f32x4{u0.x, u0.y, u0.z, u0.w};
            t1 += f32x4{u1.x, u1.y, u1.z, u1.w};
            tile_in(oacc[2 * s][mh], t0);
            tile_in(oacc[2 * s + 1][mh], t1);
        }
    }

    // ---- feed-forward
    f32x4 acc1[4][2];  // H^T of the current chunk: tile (jj, mh), register e = hidden unit 16 jj + 4 q + e of row 16 mh + c
    u16x8 hp[2][2];    // relu(H) as operand fragments of the second GEMM: [32 hidden units t][mh]
    auto init_acc1 = [&](int j) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const float4 t = *reinterpret_cast<const float4*>(b1s + 64 * j + 16 * jj + 4 * q);
            acc1[jj][0] = f32x4{t.x, t.y, t.z, t.w};
            acc1[jj][1] = f32x4{t.x, t.y, t.z, t.w};
        }
    };
    // the two MFMAs of fragment F (0..15) of piece R
    auto mma = [&](auto rc, auto fc) {
        constexpr int R = decltype(rc)::value, F = decltype(fc)::value;
        if constexpr (R < 4) {
            constexpr int jj = F & 3, ks = F >> 2;
            mfma_vgpr<F16>(acc1[jj][0], wf[F & 7], afrag[4 * R + ks][0]);
            mfma_vgpr<F16>(acc1[jj][1], wf[F & 7], afrag[4 * R + ks][1]);
        } else {
            constexpr int t = F >> 3, jl = F & 7;
            mfma_agpr<F16>(oacc[8 * (R - 4) + jl][0], wf[F & 7], hp[t][0]);
            mfma_agpr<F16>(oacc[8 * (R - 4) + jl][1], wf[F & 7], hp[t][1]);
        }
    };
    auto frag = [&](auto rc, auto fc) -> u16x8 {
        if constexpr (decltype(rc)::value < 4) return frag1(rc, fc);
        else return frag2(rc, fc);
    };
    // piece R of hidden chunk j; MODE 1: j is the final chunk of the stream (nothing further to stage), MODE 2: j is
    // the final hidden chunk and projection pieces (128-byte rows) follow it, MODE 0: another hidden chunk follows
    auto step = [&](auto rc, int j, const char* cb0, const char* cb1, auto mode) {
        constexpr int R = decltype(rc)::value;
        constexpr bool LAST = decltype(mode)::value == 1;
        constexpr bool TO_PROJ = decltype(mode)::value == 2;
        static_for<8>([&](auto f) {
            constexpr int F = decltype(f)::value;
            mma(rc, f);
            wf[F] = frag(rc, IC<F + 8>{});
        });
        if constexpr (!LAST) wait_vm_barrier<20>();
        else wait_vm_barrier<(R <= 1 ? 20 : (R <= 6 ? 4 * (6 - R) : 0))>();
        // the hidden-chunk accumulators are idle from the ReLU to the end of the chunk: reload them with the next
        // chunk's bias here, sixteen MFMAs ahead of their first use
        if constexpr (R == 7 && !LAST && !TO_PROJ) init_acc1(j + 1);
        // An LDS-DMA instruction holds the wave's issue for tens of cycles, so the four of a piece are spread
        // over the MFMA gaps of the second half instead of being issued back to back.
        static_for<8>([&](auto f) {
            constexpr int F = decltype(f)::value;
#if ASRFE_FFN_ABLATE & 2  // timing experiment: the feed-forward's weight pieces are not requested (wrong results)
            constexpr bool STAGE = false;
#else
            constexpr bool STAGE = (F & 1) == 0 && (R == 0 || !LAST);
#endif
            if constexpr (STAGE) stage_quarter_m0(IC<(R + 7) & 7>{}, F >> 1);
            mma(rc, IC<F + 8>{});
            if constexpr (!((LAST || TO_PROJ) && R == 7)) wf[F] = frag(IC<((R + 1) & 7)>{}, f);
            if constexpr (STAGE) stage_quarter_load(IC<(R + 7) & 7>{}, R == 0 ? cb0 : cb1, F >> 1);
        });
    };
    auto chunk = [&](int j, auto lastc) {
        // The fragment addresses are cheap to form from these four values (one add or xor each).  Making the values
        // opaque once per chunk keeps the compiler from hoisting all the addresses out of the loop, where they
        // do not fit in registers and come back as scratch reloads - whose vmcnt(0) waits would drain the DMA queue.
        asm volatile("" : "+v"(rb1), "+v"(sw1), "+v"(rb2), "+v"(sw2));
        frag2_bases();
        const char* cb0 = chunk_base(4 + j);
        const char* cb1 = chunk_base(5 + j);
        step(IC<0>{}, j, cb0, cb1, lastc);
        step(IC<1>{}, j, cb0, cb1, lastc);
        step(IC<2>{}, j, cb0, cb1, lastc);
        step(IC<3>{}, j, cb0, cb1, lastc);
        // MFMA-result-to-VALU wait states by hand again; the nops are tied to the tiles so that no read can move
        // above them
        asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc1[0][0]), "+v"(acc1[0][1]), "+v"(acc1[1][0]), "+v"(acc1[1][1]),
                                             "+v"(acc1[2][0]), "+v"(acc1[2][1]), "+v"(acc1[3][0]), "+v"(acc1[3][1]));
        // ReLU after the rounding, on the 16-bit pairs: a negative fp16 / bf16 is a negative int16, so one packed
        // integer max with 0 clamps two values (and turns -0 into +0)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int mh = 0; mh < 2; ++mh) {
                const f32x4& a = acc1[2 * t][mh];
                const f32x4& b = acc1[2 * t + 1][mh];
                hp[t][mh] = relu16(pack8<F16>(a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]));
            }
        step(IC<4>{}, j, cb0, cb1, lastc);
        step(IC<5>{}, j, cb0, cb1, lastc);
        step(IC<6>{}, j, cb0, cb1, lastc);
        step(IC<7>{}, j, cb0, cb1, lastc);
    };

    init_acc1(0);
    FFN_STAMP();  // LayerNorm pass done
    for (int j = 0; j < nch - 1; ++j) chunk(j, IC<0>{});
    chunk(nch - 1, IC<(OUT == 2 ? 2 : 1)>{});
    FFN_STAMP();  // feed-forward done

    // (the lane number is taken afresh here: derived from the values of the prologue it keeps those alive, or
    // spilled, over the whole kernel)
    const int lane_e = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int c_e = lane_e & 15, q_e = lane_e >> 4;
    const int rd_row = lane_e >> 3, rd_c = lane_e & 7;

    if constexpr (OUT == 2) {
        // ---- x_out leaves in register order while the ring keeps filling with Wqkv pieces; n(x_out) becomes the
        //      operand fragments of the QKV projection and the row tile restarts from b_qkv.  The tile is then 512
        //      columns of q, of k and of v in turn, each leaving through the wave's slab in the (now idle) b1 area.
        asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
        char* es = reinterpret_cast<char*>(b1s) + wave * 4096;
        // The slab rows leave through buffer stores: a row at or past M lies beyond the descriptor's size and is
        // dropped by the hardware, so no lane mask or branch surrounds a store, and the address is one 32-bit
        // per-lane offset plus an immediate.
        // (the tiled residual stream holds whole 128-row tiles: the rows past M of the last one are written too)
        const int xo_rows = (p.M + kFfnRows - 1) / kFfnRows * kFfnRows;
        const __amdgpu_buffer_rsrc_t xo_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.x_out, 0, no_store ? 0 : xo_rows * 2048, 0x00020000);
        const int xt_off = (row0 >> 5) * 65536 + lane_e * 16;
        const __amdgpu_buffer_rsrc_t qkv_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.qkv, 0, no_store ? 0 : p.M * p.ldq * 2, 0x00020000);
        int qkv_off[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) qkv_off[i] = (row0 + rd_row + 8 * i) * (p.ldq * 2) + rd_c * 16;
        // this lane's 8 bytes of slab row 16 mh + c: 16-byte chunk 2 jj + (q >> 1) (swizzled by the row), half q & 1
        const int wr_row = c_e * 128 + 8 * (q_e & 1), wr_sw = c_e & 7, wr_ch = q_e >> 1;
        row_stats();
        auto bias_tile = [&](int pass, int j) {
            const float4 u = *reinterpret_cast<const float4*>(bqs + 512 * pass + 16 * j + 4 * q_e);
            return f32x4{u.x, u.y, u.z, u.w};
        };
#pragma unroll
        for (int s = 0; s < 16; ++s) {
#pragma unroll
            for (int mh = 0; mh < 2; ++mh) {
                const f32x4 t0 = tile_out(oacc[2 * s][mh]), t1 = tile_out(oacc[2 * s + 1][mh]);
                // the tiles leave in register order: every tile is one contiguous 1 KiB run
                const u32x4 v0 = {__float_as_uint(t0[0]), __float_as_uint(t0[1]), __float_as_uint(t0[2]), __float_as_uint(t0[3])};
                const u32x4 v1 = {__float_as_uint(t1[0]), __float_as_uint(t1[1]), __float_as_uint(t1[2]), __float_as_uint(t1[3])};
                __builtin_amdgcn_raw_buffer_store_b128(v0, xo_rsrc, xt_off + 1024 * mh, 4096 * s, 0);
                __builtin_amdgcn_raw_buffer_store_b128(v1, xo_rsrc, xt_off + 1024 * (2 + mh), 4096 * s, 0);
                afrag[s][mh] = norm_frag(t0, t1, mh);
                // made here and now: left to sink to its first use, the compiler keeps all the tiles alive instead
                asm volatile("" : "+v"(afrag[s][mh]));
            }
#pragma unroll
            for (int mh = 0; mh < 2; ++mh) {
                tile_in(oacc[2 * s][mh], bias_tile(0, 2 * s));
                tile_in(oacc[2 * s + 1][mh], bias_tile(0, 2 * s + 1));
            }
        }
        FFN_STAMP();  // x_out stored, n(x_out) in fragments
#pragma unroll 1
        for (int pass = 0; pass < 3; ++pass) {
            proj_pass(4 + nch + 4 * pass, std::false_type{}, std::false_type{});
            FFN_STAMP();  // projection pass computed (the pass before it stored)
            asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
            // four column tiles x two halves make the slab's 32 rows of 128 bytes; it is read back as whole rows
#pragma unroll
            for (int P = 0; P < 8; ++P) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                    for (int mh = 0; mh < 2; ++mh) {
                        const f32x4 t = tile_out(oacc[4 * P + jj][mh]);
                        const uint2 w = make_uint2(pack_elem2<F16>(t[0], t[1]), pack_elem2<F16>(t[2], t[3]));
                        *reinterpret_cast<uint2*>(es + mh * 2048 + wr_row + (((2 * jj + wr_ch) ^ wr_sw) << 4)) = w;
                        if (pass < 2) tile_in(oacc[4 * P + jj][mh], bias_tile(pass + 1, 4 * P + jj));
                    }
                ffn_wave_lds_sync();
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = rd_row + 8 * i;
                    const u32x4 v = *reinterpret_cast<const u32x4*>(es + r * 128 + ((rd_c ^ (r & 7)) << 4));
                    __builtin_amdgcn_raw_buffer_store_b128(v, qkv_rsrc, qkv_off[i] + 128 * P, 1024 * pass, 0);
                }
                ffn_wave_lds_sync();
            }
        }
        FFN_STAMP();  // q, k, v stored
        // the padding chunk is still arriving: the LDS must not be handed to another workgroup before it has landed
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    __syncthreads();  // every wave is done with the ring: it becomes the epilogue's staging area
    // ---- epilogue (OUT 0): x_out row-major, 32-row slabs through a wave-private 4 KiB LDS tile so that global stores
    //      move whole 128-byte lines (8 consecutive lanes per row)
    char* es = smem + wave * 8192;
    const int wr_row = c_e * 128, wr_sw = c_e & 7;
#pragma unroll
    for (int nt = 0; nt < 16; ++nt) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int mh = 0; mh < 2; ++mh) {
                const f32x4 t = tile_out(oacc[2 * nt + a][mh]);
                *reinterpret_cast<float4*>(es + mh * 2048 + wr_row + (((4 * a + q_e) ^ wr_sw) << 4)) = make_float4(t[0], t[1], t[2], t[3]);
            }
        ffn_wave_lds_sync();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = rd_row + 8 * i;
            const float4 v = *reinterpret_cast<const float4*>(es + r * 128 + ((rd_c ^ (r & 7)) << 4));
            const int m = row0 + r;
            if (m < p.M && !no_store) *reinterpret_cast<float4*>(p.x_out + static_cast<int64_t>(m) * 512 + 32 * nt + rd_c * 4) = v;
        }
        ffn_wave_lds_sync();
    }
}

}  // namespace

// The dynamic-LDS limit of a kernel is a per-device attribute: asrfe_create calls this with the context's device
// current, before any compute call can run.
int32_t prepare_post_attn16(asrfe_ctx* ctx) {
    const void* fns[4] = {reinterpret_cast<const void*>(post_attn16_kernel<true, 0>),
                          reinterpret_cast<const void*>(post_attn16_kernel<true, 2>),
                          reinterpret_cast<const void*>(post_attn16_kernel<false, 0>),
                          reinterpret_cast<const void*>(post_attn16_kernel<false, 2>)};
    for (const void* f : fns)
        if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return set_err(ctx, ASRFE_ERR_HIP, "post_attn16: cannot raise the dynamic LDS limit to 160 KiB");
    return ASRFE_OK;
}

int32_t launch_post_attn16(asrfe_ctx* ctx, const FfnParams& p, hipStream_t s) {
    if (p.M <= 0) return ASRFE_OK;
    if (p.ff < 64 || p.ff % 64 != 0 || p.ff > 7168)
        return set_err(ctx, ASRFE_ERR_SHAPE, "post_attn16: linear_units must be a multiple of 64 in 64..7168");
    if (!p.att || !p.fsmn || !p.x_out || !p.wstream || !p.bo || !p.b1 || !p.b2)
        return set_err(ctx, ASRFE_ERR_BAD_ARG, "post_attn16: null argument");
    if (!p.tiled || p.y)
        return set_err(ctx, ASRFE_ERR_BAD_ARG, "post_attn16: serves the tiled residual stream with the qkv output or with x_out alone");
    if (p.qkv && (!p.bqkv || p.ldq % 8 != 0 || p.ff > kPostAttnQkvMaxFf))
        return set_err(ctx, ASRFE_ERR_BAD_ARG, "post_attn16: qkv output needs b_qkv and linear_units <= 5632");
    // (its stores address rows by signed 32-bit byte offsets, up to one row tile past M)
    if ((static_cast<int64_t>(p.M) + kFfnRows) * std::max<int64_t>(2048, 2 * static_cast<int64_t>(p.ldq)) >= (int64_t{1} << 31))
        return set_err(ctx, ASRFE_ERR_SHAPE, "post_attn16: x_out and qkv must each stay below 2 GiB per launch");
    const int out = p.qkv ? 2 : 0;
    const int lds = kFfnRing + std::max(p.ff, kSlabFloats) * 4 + 2 * 512 * 4 + (out == 2 ? 1536 * 4 : 0);
    const dim3 grid((p.M + kFfnRows - 1) / kFfnRows), block(kFfnThreads);
    const bool f16 = ctx->cfg.enc_compute_dtype == ASRFE_DT_F16;
    if (f16) {
        if (out == 2) hipLaunchKernelGGL((post_attn16_kernel<true, 2>), grid, block, lds, s, p);
        else hipLaunchKernelGGL((post_attn16_kernel<true, 0>), grid, block, lds, s, p);
    } else {
        if (out == 2) hipLaunchKernelGGL((post_attn16_kernel<false, 2>), grid, block, lds, s, p);
        else hipLaunchKernelGGL((post_attn16_kernel<false, 0>), grid, block, lds, s, p);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_err(ctx, ASRFE_ERR_HIP, std::string("post_attn16 launch: ") + hipGetErrorString(e));
    return ASRFE_OK;
}

}  // namespace asrfe
