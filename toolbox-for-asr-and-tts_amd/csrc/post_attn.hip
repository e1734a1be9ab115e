// Everything of a SAN-M encoder layer that follows the attention kernel, in one kernel on gfx950:
//
//     x_mid = x + fsmn + att Wo^T + b_o                   (attention output projection + FSMN memory + residual)
//     x_out = x_mid + W2 relu(W1 n(x_mid) + b1) + b2      (position-wise feed-forward, fp32 residual stream)
//     y     = n(x_out)  as 16-bit                          (OUT 1: the next layer's normalised input), or
//     qkv   = n(x_out) Wqkv^T + b_qkv  as 16-bit           (OUT 2: the next layer's q, k, v, so that layer needs no
//                                                           projection launch and y never exists in memory)
// n(.) normalises a row of 512 to zero mean and unit variance; the scale and shift of the two LayerNorms are
// folded into W1 / b1 and into the next layer's QKV projection when the weights are loaded.
//
// One workgroup = 128 rows, 4 waves, ONE wave per SIMD so that a wave may use the whole 512-entry register
// file.  A wave owns 32 complete rows for the whole kernel:
//   - the 32 x 512 fp32 row tile is 16 accumulators of v_mfma_f32_32x32x16 (256 registers).  It starts as
//     x + fsmn + b_o, the output projection accumulates into it, the row statistics are taken from it in
//     registers (a row lives in the two lanes l and l + 32), b2 is added and the feed-forward accumulates
//     into it again: neither x_mid nor the residual ever leaves registers;
//   - att (32 x 512) and later n(x_mid) are kept as the 32 MFMA operand fragments of the GEMM they feed
//     (128 registers, the same ones);
//   - the hidden layer is produced 64 units at a time: H^T[64][32] = W1_chunk n(x_mid)^T (2 accumulators),
//     ReLU + rounding turn it straight into the operand fragments of the second GEMM, which adds
//     W2[:, chunk] H^T into the row tile.  The 2048-wide intermediate never exists in memory.
//
// An accumulator holds 4 consecutive rows per register quad (row = 8 g + 4 (lane >> 5) + e), while an
// operand fragment holds 8 consecutive k per lane (k = 8 (lane >> 5) + i).  Feeding accumulator registers
// back as an operand therefore visits k in the order p -> swap of bits 2 and 3 of p inside every 16; the
// host stores the COLUMNS of W1 and W2 in that same order (asrfe_load_encoder_weights), which makes the
// product exact again without moving any data on the device.
//
// Weights stream through LDS as 16 KiB pieces, 8 per chunk; a layer's stream is 4 chunks of Wo followed by one
// chunk per 64 hidden units and, for OUT 2, by 12 chunks of the next layer's Wqkv (three passes of 512 output
// columns, each laid out like Wo but with the k order of accumulator-fed operands) and one chunk of padding that
// the last pass prefetches and nobody reads:
//     Wo chunk c, piece r : Wo[128 c .. +128][64 r .. +64]      (one chunk = one quarter of the output columns)
//     FFN pieces 0..3     : W1[chunk rows 64][128 k]   (256-byte rows, 16-byte slot c of row r holds chunk c ^ (r & 15))
//     FFN pieces 4..7     : W2[128 n][chunk cols 64]
// (128-byte rows: slot c of row r holds chunk c ^ ((r >> 1) & 7)).  The host stores the stream of these images
// in consumption order, so a piece is 16 KiB of contiguous memory copied by 16-byte LDS-DMA loads.  Piece r
// of every chunk uses ring slot r.  A wave issues 16 MFMAs per piece; the fragment for MFMA i + 8 is read right
// after MFMA i, so in the middle of piece q the wave needs piece q + 1: there it waits (counted vmcnt: five
// younger pieces stay in flight), meets the other waves at one s_barrier, and then refills the slot of piece
// q - 1, whose last read every wave finished before that barrier.
#include <algorithm>
#include <utility>

#include "common.h"
#include "elem.h"
#include "encoder.h"
#include "lds_dma.h"

namespace asrfe {

#ifndef ASRFE_FFN_TIMING
#define ASRFE_FFN_TIMING 0
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

// The MFMAs are issued through inline assembly so that the register class of each accumulator is fixed: the
// 32 x 512 output tile fills the 256 accumulation registers (AGPRs), the two hidden-chunk tiles stay in
// architectural VGPRs next to the operand fragments.  (Left to the register allocator, all 288 accumulator
// registers compete for the 256 AGPRs and the loop fills with copies and spills.)
template <bool F16>
__device__ __forceinline__ void mfma_agpr(f32x16& c, const u16x8 a, const u16x8 b) {
    if constexpr (F16) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
template <bool F16>
__device__ __forceinline__ void mfma_vgpr(f32x16& c, const u16x8 a, const u16x8 b) {
    if constexpr (F16) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}

// Moves one tile between the accumulation registers and working VGPRs at a fixed point of the program (the empty
// asm statements keep the compiler from merging or hoisting the copies, which would need all 256 at once).
__device__ __forceinline__ f32x16 tile_out(f32x16& acc) {
    asm volatile("" : "+a"(acc));
    f32x16 t = acc;
    asm volatile("" : "+v"(t));
    return t;
}
__device__ __forceinline__ void tile_in(f32x16& acc, const f32x16 t) {
    acc = t;
    asm volatile("" : "+a"(acc));
}

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

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

// Row statistics of a lane's 256 values (16 tiles of 16): sums of (x - k) and (x - k)^2 with k = the mean of the
// lane's first tile, two values per packed instruction (1.5 instructions per value).  The shift keeps the variance
// formula free of cancellation when a row's mean is large against its spread.
struct RowStat {
    v2f s1, s2, nk2;
    float k;
};
__device__ __forceinline__ void stat_add_tile(RowStat& r, const int nt, const f32x16& t) {
    if (nt == 0) {  // nt is a constant after unrolling
        float s = 0.0f;
#pragma unroll
        for (int e = 0; e < 16; e += 4) s += (t[e] + t[e + 1]) + (t[e + 2] + t[e + 3]);
        r.k = s * (1.0f / 16.0f);
        r.nk2 = v2f{-r.k, -r.k};
        r.s1 = v2f{0.0f, 0.0f};
        r.s2 = v2f{0.0f, 0.0f};
    }
#pragma unroll
    for (int e = 0; e < 16; e += 2) {
        const v2f d = v2f{t[e], t[e + 1]} + r.nk2;
        r.s1 += d;
        r.s2 = __builtin_elementwise_fma(d, d, r.s2);
    }
}
// -> mean and sum of squared deviations of the lane's 256 values
__device__ __forceinline__ void stat_lane(const RowStat& r, float& mean, float& m2) {
    const float s1 = r.s1.x + r.s1.y, s2 = r.s2.x + r.s2.y;
    mean = r.k + s1 * (1.0f / 256.0f);
    m2 = s2 - s1 * s1 * (1.0f / 256.0f);
}
// the two lanes of a row (256 values each) -> mean and 1 / sqrt(var + eps) of the 512
__device__ __forceinline__ void stat_finish_row(float& mean, float m2, float eps, float& rstd) {
    const float mo = __shfl_xor(mean, 32), qo = __shfl_xor(m2, 32);
    const float d = mo - mean;
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


// OUT: 0 = x_out only, 1 = x_out and y, 2 = x_out and the next layer's qkv
// TILED: the residual stream between two launches of this kernel is kept in the order the row tile has in registers
// (block (32-row group, tile nt, quad g) = 64 lanes x 16 bytes: lane l31 + 32 h holds columns 32 nt + 8 g + 4 h .. + 3
// of row l31), so a wave reads and writes it as whole 1 KiB runs straight from / into registers, with no trip through
// LDS.  x is read in that order; x_out is written in it for OUT 2 and row-major otherwise (the encoder's last layer).
// No residual tile is waited for before the output projection starts: they arrive while it runs and are added
// between its MFMAs, twelve to tiles of column quarters that have not started yet, the first quarter's four at the end.
template <bool F16, int OUT, bool TILED>
__global__ __launch_bounds__(kFfnThreads) void post_attn_kernel(const FfnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    saturate_fp16_conversions<F16>();
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int row0 = blockIdx.x * kFfnRows + wave * 32;
    const int nch = p.ff >> 6;
    // the bias vectors live in LDS: a global load per use would be waited for one by one.  The b1 area is at least
    // 16 KiB: after the feed-forward it holds the waves' output slabs (OUT 2).
    float* b1s = reinterpret_cast<float*>(smem + kFfnRing);
    float* b2s = b1s + max(p.ff, kSlabFloats);
    float* bos = b2s + 512;
    float* bqs = bos + 512;  // b_qkv of the next layer, 1536 (OUT 2)

#if ASRFE_FFN_TIMING
    // timing build, flags bit 8: wave 0 of workgroup 100 writes cycle-counter stamps to p.trace
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
    // quarter i (4 KiB: one 1 KiB run per wave) of piece R of the chunk whose eight pieces start at cbase
    // (the quarter's offset rides in the per-lane part of the address: one scalar base per piece instead of a 64-bit
    // scalar add per load; the four values are made opaque so that they stay four registers)
    uint32_t wq[4] = {wlane, wlane + 4096, wlane + 8192, wlane + 12288};
    asm volatile("" : "+v"(wq[0]), "+v"(wq[1]), "+v"(wq[2]), "+v"(wq[3]));
    auto stage_quarter = [&](auto rc, const char* cbase, int i) {
        constexpr int R = decltype(rc)::value;
        const uint32_t dst = __builtin_amdgcn_readfirstlane(smem_base + R * kPieceBytes + wave * 1024);
        lds_dma16(cbase + R * kPieceBytes, wq[i], dst + i * 4096);
    };
    // chunk c of the layer's stream: 0..3 are Wo, 4 + j is hidden chunk j
    auto chunk_base = [&](int c) { return wsrc + static_cast<int64_t>(c) * (8 * kPieceBytes); };
    auto stage = [&](auto rc, const char* cbase) {
#pragma unroll
        for (int i = 0; i < 4; ++i) stage_quarter(rc, cbase, i);
    };
    // the first pieces fly while the activations are loaded; the other slots serve as their staging until then
    const char* cb_first = chunk_base(0);
    stage(IC<0>{}, cb_first); stage(IC<1>{}, cb_first);
    if constexpr (!TILED) { stage(IC<2>{}, cb_first); stage(IC<3>{}, cb_first); }  // (TILED: slots 2..7 stage activations)

    for (int i = tid * 4; i < p.ff; i += kFfnThreads * 4)
        *reinterpret_cast<float4*>(b1s + i) = *reinterpret_cast<const float4*>(p.b1 + i);
    if (tid < 128) *reinterpret_cast<float4*>(b2s + tid * 4) = *reinterpret_cast<const float4*>(p.b2 + tid * 4);
    else *reinterpret_cast<float4*>(bos + (tid - 128) * 4) = *reinterpret_cast<const float4*>(p.bo + (tid - 128) * 4);
    if constexpr (OUT == 2) {
        for (int i = tid * 4; i < 1536; i += kFfnThreads * 4)
            *reinterpret_cast<float4*>(bqs + i) = *reinterpret_cast<const float4*>(p.bqkv + i);
    }
    __syncthreads();

    // ---- activations.  att, fsmn and x come in by LDS-DMA (no staging registers) as 4 KiB slabs of 32 rows x
    //      128 bytes, filled by four loads whose per-lane source addresses carry the bank swizzle.  Every wave owns
    //      ring slot 4 + wave as four such slabs, refilled as they are consumed, so three slabs are always in
    //      flight.  Slab order: att k-slices 0..7, then per 64 columns: fsmn, x tile 2 p, x tile 2 p + 1.
    //      Row tile layout: register 4 g + e of tile nt is column 32 nt + 8 g + 4 h + e of row l31.
    f32x16 oacc[16];
    u16x8 afrag[32];  // att, later n(x_mid), as operand fragments: fragment s covers k = 16 s .. 16 s + 15
    const bool has_x = p.x != nullptr;  // layer 0 has no residual into the attention block: x counts as zero
    // TILED: three register sets that the residual tiles rotate through: set k % 3 for the tile that piece 2 k adds
    f32x4 xr[3][4];
    const uint32_t lane16 = lane * 16;
    // the wave's 32-row group of the tiled residual stream (64 KiB); layer 0 reads (and ignores) the first 64 KiB of
    // the weight stream instead, so that the count of loads in flight is the same in every layer
    const char* xtile = has_x ? reinterpret_cast<const char*>(p.x) + static_cast<int64_t>(row0 >> 5) * 65536 : wsrc;
    auto xload = [&](auto setc, int nt, auto gc) {
        constexpr int S = decltype(setc)::value, G = decltype(gc)::value;
        gload16<G * 1024>(xr[S][G], xtile + nt * 4096, lane16);
    };
    if constexpr (TILED) {
        // att and fsmn come in by LDS-DMA as slabs of 32 rows x 128 bytes; the residual stream does not pass through
        // here at all.  A wave's slab slot j is the part of ring slot 2 + j that the same wave later fills with its
        // share of weight piece 2 + j (the four 1 KiB runs at wave * 1 KiB + k * 4 KiB), so it can stage that piece
        // as soon as it has read its own last slab there, without asking the other waves.  Six slots: five slabs in
        // flight behind the one being read.  Slab I uses slot (I + 2) % 6, which makes the last six slabs free slots
        // 0..5 in order: pieces 2..7 are requested in stream order.
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
        // slabs 0..7: att k-slices, slabs 8..15: fsmn column pairs of tiles (2 p, 2 p + 1); rows 8 k .. 8 k + 7 of a
        // slab sit at k * 4 KiB
        auto slab_dma = [&](auto ic) {
            constexpr int I = decltype(ic)::value;
            const uint32_t dst = sl_lds + ((I + 2) % 6) * kPieceBytes;
            const char* src = I < 8 ? asrc + I * 128 : fsrc + (I - 8) * 128;
#pragma unroll
            for (int k = 0; k < 4; ++k) lds_dma16(src, aoff[k], dst + k * 4096);
        };
        static_for<6>([&](auto ic) { slab_dma(ic); });
        uint2 fs[8];
        static_for<16>([&](auto ic) {
            constexpr int I = decltype(ic)::value;
            // behind slab I in the queue: the slabs up to I + 5 and the weight pieces requested so far, 20 loads in
            // every iteration
            asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
            const char* sl = smem + (2 + (I + 2) % 6) * kPieceBytes + wave * 1024 + (l31 >> 3) * 4096 + (l31 & 7) * 128;
            const int sw = l31 & 7;
            if constexpr (I < 8) {
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    afrag[4 * I + s] = *reinterpret_cast<const u16x8*>(sl + (((2 * s + h) ^ sw) << 4));
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(afrag[4 * I]), "+v"(afrag[4 * I + 1]), "+v"(afrag[4 * I + 2]), "+v"(afrag[4 * I + 3]) : : "memory");
            } else {
                constexpr int P = I - 8;
#pragma unroll
                for (int q = 0; q < 8; ++q) fs[q] = *reinterpret_cast<const uint2*>(sl + ((q ^ sw) << 4) + 8 * h);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fs[0]), "+v"(fs[1]), "+v"(fs[2]), "+v"(fs[3]), "+v"(fs[4]), "+v"(fs[5]), "+v"(fs[6]), "+v"(fs[7]) : : "memory");
                static_for<2>([&](auto nsc) {
                    constexpr int ns = decltype(nsc)::value, nt = 2 * P + ns;
                    f32x16 t;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float f0, f1, f2, f3;
                        unpack_elem2<F16>(fs[4 * ns + g].x, f0, f1);
                        unpack_elem2<F16>(fs[4 * ns + g].y, f2, f3);
                        const float4 bo4 = *reinterpret_cast<const float4*>(bos + 32 * nt + 8 * g + 4 * h);
                        t[4 * g] = f0 + bo4.x; t[4 * g + 1] = f1 + bo4.y; t[4 * g + 2] = f2 + bo4.z; t[4 * g + 3] = f3 + bo4.w;
                    }
                    tile_in(oacc[nt], t);
                });
            }
            if constexpr (I + 6 < 16) slab_dma(IC<I + 6>{});
            else stage(IC<I - 8>{}, cb_first);  // the slot just read takes this wave's share of weight piece I - 8
        });
    } else {
        const uint32_t xs_lds = __builtin_amdgcn_readfirstlane(smem_base + (4 + wave) * kPieceBytes);
        const char* xs = smem + (4 + wave) * kPieceBytes;
        // the per-lane offset is an UNSIGNED 32-bit addend of the scalar base, so the base is the wave's first valid
        // row and rows past M - 1 re-read row M - 1 (never a negative offset)
        const int rbase = min(row0, p.M - 1);
        const char* xsrc = reinterpret_cast<const char*>(has_x ? p.x : p.x_out) + static_cast<int64_t>(rbase) * 2048;
        const char* asrc = reinterpret_cast<const char*>(p.att) + static_cast<int64_t>(rbase) * 1024;
        const char* fsrc = reinterpret_cast<const char*>(p.fsmn) + static_cast<int64_t>(rbase) * 1024;
        uint32_t xoff[4], aoff[4];  // lane's source offsets for rows 8 k + (lane >> 3): 2 KiB rows (x) and 1 KiB rows
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = 8 * k + (lane >> 3);
            const int rr = min(row0 + r, p.M - 1) - rbase;
            xoff[k] = static_cast<uint32_t>(rr * 2048 + (((lane & 7) ^ (r & 7)) << 4));
            aoff[k] = static_cast<uint32_t>(rr * 1024 + (((lane & 7) ^ (r & 7)) << 4));
        }
        // slab i of the sequence -> its four loads into slab slot i & 3
        auto slab_dma = [&](auto ic) {
            constexpr int I = decltype(ic)::value;
            const uint32_t dst = xs_lds + (I & 3) * 4096;
            if constexpr (I < 8) {
#pragma unroll
                for (int k = 0; k < 4; ++k) lds_dma16(asrc + I * 128, aoff[k], dst + k * 1024);
            } else if constexpr ((I - 8) % 3 == 0) {
#pragma unroll
                for (int k = 0; k < 4; ++k) lds_dma16(fsrc + ((I - 8) / 3) * 128, aoff[k], dst + k * 1024);
            } else {
                constexpr int nt = 2 * ((I - 8) / 3) + ((I - 8) % 3 - 1);
#pragma unroll
                for (int k = 0; k < 4; ++k) lds_dma16(xsrc + nt * 128, xoff[k], dst + k * 1024);
            }
        };
        slab_dma(IC<0>{}); slab_dma(IC<1>{}); slab_dma(IC<2>{}); slab_dma(IC<3>{});
        uint2 fs[8];  // the current fsmn slab in row-tile layout: fs[4 ns + g] = 4 values of tile 2 p + ns
        static_for<32>([&](auto ic) {
            constexpr int I = decltype(ic)::value;
            // slab I has landed once at most the loads of the slabs behind it are outstanding (in-order counter)
            if constexpr (I <= 28) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if constexpr (I == 29) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if constexpr (I == 30) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const char* sl = xs + (I & 3) * 4096 + l31 * 128;
            const int sw = l31 & 7;
            if constexpr (I < 8) {
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    afrag[4 * I + s] = *reinterpret_cast<const u16x8*>(sl + (((2 * s + h) ^ sw) << 4));
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(afrag[4 * I]), "+v"(afrag[4 * I + 1]), "+v"(afrag[4 * I + 2]), "+v"(afrag[4 * I + 3]) : : "memory");
            } else if constexpr ((I - 8) % 3 == 0) {
#pragma unroll
                for (int q = 0; q < 8; ++q) fs[q] = *reinterpret_cast<const uint2*>(sl + ((q ^ sw) << 4) + 8 * h);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fs[0]), "+v"(fs[1]), "+v"(fs[2]), "+v"(fs[3]), "+v"(fs[4]), "+v"(fs[5]), "+v"(fs[6]), "+v"(fs[7]) : : "memory");
            } else {
                constexpr int ns = (I - 8) % 3 - 1, nt = 2 * ((I - 8) / 3) + ns;
                f32x16 t;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 v = *reinterpret_cast<const float4*>(sl + (((2 * g + h) ^ sw) << 4));
                    t[4 * g] = v.x; t[4 * g + 1] = v.y; t[4 * g + 2] = v.z; t[4 * g + 3] = v.w;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t) : : "memory");  // the slab is read before it is refilled
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float f0, f1, f2, f3;
                    unpack_elem2<F16>(fs[4 * ns + g].x, f0, f1);
                    unpack_elem2<F16>(fs[4 * ns + g].y, f2, f3);
                    const float4 bo4 = *reinterpret_cast<const float4*>(bos + 32 * nt + 8 * g + 4 * h);
                    if (has_x) {
                        t[4 * g] += f0 + bo4.x; t[4 * g + 1] += f1 + bo4.y; t[4 * g + 2] += f2 + bo4.z; t[4 * g + 3] += f3 + bo4.w;
                    } else {
                        t[4 * g] = f0 + bo4.x; t[4 * g + 1] = f1 + bo4.y; t[4 * g + 2] = f2 + bo4.z; t[4 * g + 3] = f3 + bo4.w;
                    }
                }
                tile_in(oacc[nt], t);
            }
            if constexpr (I + 4 < 32) slab_dma(IC<I + 4>{});
        });
        // every wave is done with its slabs: slots 4..6 can take their weight pieces
        __syncthreads();
        stage(IC<4>{}, cb_first); stage(IC<5>{}, cb_first); stage(IC<6>{}, cb_first);
    }

    // ---- fragment reads: kind 1 = W1 piece (256-byte rows), kind 2 = Wo / W2 piece (128-byte rows)
    int rb1 = l31 * 256, sw1 = (h ^ (l31 & 15)) << 4;
    int rb2 = l31 * 128, sw2 = (h ^ ((l31 >> 1) & 7)) << 4;
    auto frag1 = [&](auto rc, auto fc) -> u16x8 {  // W1 rows 32 ct + l31, k-step s of the piece
        constexpr int R = decltype(rc)::value, F = decltype(fc)::value;
        constexpr int s = F >> 1, ct = F & 1;
        return *reinterpret_cast<const u16x8*>(smem + R * kPieceBytes + ct * 8192 + rb1 + (sw1 ^ (s << 5)));
    };
    // Kind 2 addresses are one of eight per-lane bases (k-step kk; ring half: an LDS read's immediate offset ends at
    // 64 KiB) plus a compile-time offset.  The bases are made opaque where they are formed (frag2_bases), otherwise
    // the compiler re-associates the sum and spends a vector add on every fragment.
    int fb2[2][4];
    auto frag2_bases = [&]() {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            fb2[0][kk] = rb2 + (sw2 ^ (kk << 5));
            fb2[1][kk] = fb2[0][kk] + 4 * kPieceBytes;
            asm volatile("" : "+v"(fb2[0][kk]), "+v"(fb2[1][kk]));
        }
    };
    auto frag2 = [&](auto rc, auto fc) -> u16x8 {  // rows 32 ntl + l31, k-step kk of the piece
        constexpr int R = decltype(rc)::value, F = decltype(fc)::value;
        constexpr int kk = F >> 2, ntl = F & 3;
        return *reinterpret_cast<const u16x8*>(smem + fb2[R >> 2][kk] + (R & 3) * kPieceBytes + ntl * 4096);
    };
    u16x8 wf[8];     // rolling weight fragments

    wait_vm_barrier<24>();  // piece 0 has landed everywhere

    // ---- a projection pass: the row tile += afrag W^T for 512 output columns, 4 chunks of 8 pieces starting at
    //      chunk c0 of the stream, whose piece 0 every wave has seen land.  TO_W1: W1 pieces (256-byte rows) follow
    //      the pass and their first fragments are read ahead; otherwise the pass ends with nothing read ahead, which
    //      leaves the registers to the output code behind it.  FIRST: the output projection, whose chunk is a quarter
    //      of the output columns over all of k (the others: a quarter of k over all columns, two k-slices per chunk);
    //      with TILED every even piece also adds a residual tile in its middle and requests the tile that the
    //      piece six further on adds (out_proj_vmcnt above has the schedule).
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
                constexpr bool XADD = FIRST && TILED && PI % 2 == 0;
                constexpr bool XLOAD = FIRST && TILED && xload_piece(PI);
                constexpr int XSET = (PI / 2) % 3;
                static_for<8>([&](auto f) {
                    constexpr int F = decltype(f)::value;
                    mfma_agpr<F16>(oacc[4 * nq + (F & 3)], wf[F], afrag[4 * ks + (F >> 2)]);
                    wf[F] = frag2(rc, IC<F + 8>{});
                });
                if constexpr (XADD) {
                    // (the tile moves through working registers as a whole: element-wise updates of an accumulation
                    // tile make the compiler copy all sixteen registers per element)
                    constexpr int nt = xtile_of(PI / 2);
                    wait_vm_barrier_regs<out_proj_vmcnt_x(PI)>(xr[XSET]);
                    if (has_x) {
                        f32x16 t = tile_out(oacc[nt]);
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            t[4 * g] += xr[XSET][g][0]; t[4 * g + 1] += xr[XSET][g][1];
                            t[4 * g + 2] += xr[XSET][g][2]; t[4 * g + 3] += xr[XSET][g][3];
                        }
                        tile_in(oacc[nt], t);
                    }
                } else if constexpr (FIRST && TILED) {
                    wait_vm_barrier<out_proj_vmcnt(PI)>();
                } else {
                    wait_vm_barrier<20>();
                }
                static_for<8>([&](auto f) {
                    constexpr int F = decltype(f)::value;
                    mfma_agpr<F16>(oacc[4 * nq + (F & 3)], wf[F], afrag[4 * ks + 2 + (F >> 2)]);
                    if constexpr (C == 3 && R == 7) {
                        if constexpr (TO_W1) wf[F] = frag1(IC<0>{}, f);
                    } else {
                        wf[F] = frag2(IC<((R + 1) & 7)>{}, f);
                    }
                    if constexpr ((F & 1) == 0) {
                        if constexpr (R == 0) {
                            // (TILED: the first pass finds its piece 7 already requested by the prologue)
                            if constexpr (!(FIRST && TILED && C == 0)) stage_quarter(IC<7>{}, cb0, F >> 1);
                        } else {
                            stage_quarter(IC<(R + 7) & 7>{}, cb1, F >> 1);
                        }
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

    // ---- x_mid is complete: row statistics, n(x_mid) as the operand fragments of W1, and b2 into the row tile.
    //      The compiler does not see inside the asm MFMAs, so the wait states between the last of them and the
    //      first read of their results are covered by hand (volatile asm statements keep their order).
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    {
        float mean = 0.0f, m2 = 0.0f, rstd;
        RowStat rs;
#pragma unroll
        for (int nt = 0; nt < 16; ++nt) {
            const f32x16 t = tile_out(oacc[nt]);
            stat_add_tile(rs, nt, t);
        }
        stat_lane(rs, mean, m2);
        stat_finish_row(mean, m2, p.eps, rstd);
        const float shift = -mean * rstd;
#pragma unroll
        for (int nt = 0; nt < 16; ++nt) {
            f32x16 t = tile_out(oacc[nt]);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = t[8 * q + e] * rstd + shift;
                afrag[2 * nt + q] = pack8<F16>(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 u = *reinterpret_cast<const float4*>(b2s + 32 * nt + 8 * g + 4 * h);
                t[4 * g] += u.x; t[4 * g + 1] += u.y; t[4 * g + 2] += u.z; t[4 * g + 3] += u.w;
            }
            tile_in(oacc[nt], t);
        }
    }

    // ---- feed-forward
    f32x16 acc1[2];  // H^T of the current chunk: register 4 g + e of tile ct is hidden unit 32 ct + 8 g + 4 h + e
    u16x8 hp[4];     // relu(H) as operand fragments of the second GEMM
    auto init_acc1 = [&](int j) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 t = *reinterpret_cast<const float4*>(b1s + 64 * j + 32 * ct + 8 * g + 4 * h);
                acc1[ct][4 * g] = t.x; acc1[ct][4 * g + 1] = t.y; acc1[ct][4 * g + 2] = t.z; acc1[ct][4 * g + 3] = t.w;
            }
    };
    auto mma = [&](auto rc, auto fc) {
        constexpr int R = decltype(rc)::value, F = decltype(fc)::value;
        if constexpr (R < 4) {
            constexpr int s = F >> 1, ct = F & 1;
            mfma_vgpr<F16>(acc1[ct], wf[F & 7], afrag[8 * R + s]);
        } else {
            constexpr int kk = F >> 2, ntl = F & 3;
            mfma_agpr<F16>(oacc[4 * (R - 4) + ntl], wf[F & 7], hp[kk]);
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
        // chunk's bias here, eight MFMAs ahead of their first use
        if constexpr (R == 7 && !LAST && !TO_PROJ) init_acc1(j + 1);
        // An LDS-DMA instruction holds the wave's issue for tens of cycles, so the four of a piece are spread
        // over the MFMA gaps of the second half instead of being issued back to back.
        static_for<8>([&](auto f) {
            constexpr int F = decltype(f)::value;
            mma(rc, IC<F + 8>{});
            if constexpr (!((LAST || TO_PROJ) && R == 7)) wf[F] = frag(IC<((R + 1) & 7)>{}, f);
            if constexpr ((F & 1) == 0) {
                if constexpr (R == 0) stage_quarter(IC<7>{}, cb0, F >> 1);
                else if constexpr (!LAST) stage_quarter(IC<(R + 7) & 7>{}, cb1, F >> 1);
            }
        });
    };
    auto chunk = [&](int j, auto lastc) {
        // The fragment addresses are cheap to form from these four values (one add or xor each).  Making the values
        // opaque once per chunk keeps the compiler from hoisting all the addresses out of the loop, where they
        // do not fit in registers and come back as scratch reloads - whose vmcnt(0) waits would drain the DMA queue.
        asm volatile("" : "+v"(rb1), "+v"(sw1), "+v"(rb2), "+v"(sw2));
        frag2_bases();
        // this chunk's and the next chunk's pieces in the weight stream (uniform: scalar registers)
        const char* cb0 = chunk_base(4 + j);
        const char* cb1 = chunk_base(5 + j);
        step(IC<0>{}, j, cb0, cb1, lastc);
        step(IC<1>{}, j, cb0, cb1, lastc);
        step(IC<2>{}, j, cb0, cb1, lastc);
        step(IC<3>{}, j, cb0, cb1, lastc);
        // MFMA-result-to-VALU wait states by hand again; the nops are tied to both tiles so that no read can move
        // above them
        asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc1[0]), "+v"(acc1[1]));
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const f32x16& a = acc1[kk >> 1];
            const int o = 8 * (kk & 1);
            // ReLU after the rounding, on the 16-bit pairs: a negative fp16 / bf16 is a negative int16, so one packed
            // integer max with 0 clamps two values (and turns -0 into +0); rounding keeps the sign, so the result is
            // the one of ReLU before rounding
            hp[kk] = relu16(pack8<F16>(a[o], a[o + 1], a[o + 2], a[o + 3], a[o + 4], a[o + 5], a[o + 6], a[o + 7]));
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

    if constexpr (OUT == 2) {
        // ---- x_out leaves through the wave's slab in the (now idle) b1 area while the ring keeps filling with
        //      Wqkv pieces; n(x_out) becomes the operand fragments of the QKV projection and the row tile restarts
        //      from b_qkv.  The tile is then 512 columns of q, of k and of v in turn.
        asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
        char* es = reinterpret_cast<char*>(b1s) + wave * 4096;
        // (the lane number is taken afresh here: derived from the values of the prologue it keeps those alive, or
        // spilled, over the whole kernel)
        const int lane_e = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        const int l31_e = lane_e & 31;
        const int wr_row = l31_e * 128, wr_sw = l31_e & 7;
        const int rd_row = lane_e >> 3, rd_c = lane_e & 7;
        // The slab rows leave through buffer stores: a row at or past M lies beyond the descriptor's size and is
        // dropped by the hardware, so no lane mask or branch surrounds a store, and the address is one 32-bit
        // per-lane offset plus an immediate.
        // (the tiled residual stream holds whole 128-row tiles: the rows past M of the last one are written too)
        const int xo_rows = TILED ? (p.M + kFfnRows - 1) / kFfnRows * kFfnRows : p.M;
        const __amdgpu_buffer_rsrc_t xo_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.x_out, 0, no_store ? 0 : xo_rows * 2048, 0x00020000);
        const int xt_off = (row0 >> 5) * 65536 + lane_e * 16;
        const __amdgpu_buffer_rsrc_t qkv_rsrc = __builtin_amdgcn_make_buffer_rsrc(p.qkv, 0, no_store ? 0 : p.M * p.ldq * 2, 0x00020000);
        int xo_off[4], qkv_off[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            xo_off[i] = (row0 + rd_row + 8 * i) * 2048 + rd_c * 16;
            qkv_off[i] = (row0 + rd_row + 8 * i) * (p.ldq * 2) + rd_c * 16;
        }
        float mean = 0.0f, m2 = 0.0f, rstd;
        RowStat rs;
#pragma unroll
        for (int nt = 0; nt < 16; ++nt) {
            const f32x16 t = tile_out(oacc[nt]);
            stat_add_tile(rs, nt, t);
        }
        stat_lane(rs, mean, m2);
        stat_finish_row(mean, m2, p.eps, rstd);
        const float shift = -mean * rstd;
        auto bias_tile = [&](int pass, int nt) {
            f32x16 t;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 u = *reinterpret_cast<const float4*>(bqs + 512 * pass + 32 * nt + 8 * g + 4 * h);
                t[4 * g] = u.x; t[4 * g + 1] = u.y; t[4 * g + 2] = u.z; t[4 * g + 3] = u.w;
            }
            return t;
        };
#pragma unroll
        for (int nt = 0; nt < 16; ++nt) {
            const f32x16 t = tile_out(oacc[nt]);
            if constexpr (TILED) {
                // the tile leaves in register order: quad g of every lane is one contiguous 1 KiB run
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const u32x4 v = {__float_as_uint(t[4 * g]), __float_as_uint(t[4 * g + 1]), __float_as_uint(t[4 * g + 2]),
                                     __float_as_uint(t[4 * g + 3])};
                    __builtin_amdgcn_raw_buffer_store_b128(v, xo_rsrc, xt_off + 1024 * g, 4096 * nt, 0);
                }
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4*>(es + wr_row + (((2 * g + h) ^ wr_sw) << 4)) =
                        make_float4(t[4 * g], t[4 * g + 1], t[4 * g + 2], t[4 * g + 3]);
                ffn_wave_lds_sync();
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = rd_row + 8 * i;
                    const u32x4 v = *reinterpret_cast<const u32x4*>(es + r * 128 + ((rd_c ^ (r & 7)) << 4));
                    __builtin_amdgcn_raw_buffer_store_b128(v, xo_rsrc, xo_off[i] + 128 * nt, 0, 0);
                }
                ffn_wave_lds_sync();
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = t[8 * q + e] * rstd + shift;
                afrag[2 * nt + q] = pack8<F16>(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
            }
            // made here and now: left to sink to its first use, the compiler keeps all sixteen tiles alive instead
            asm volatile("" : "+v"(afrag[2 * nt]), "+v"(afrag[2 * nt + 1]));
            tile_in(oacc[nt], bias_tile(0, nt));
        }
        FFN_STAMP();  // x_out stored, n(x_out) in fragments
#pragma unroll 1
        for (int pass = 0; pass < 3; ++pass) {
            proj_pass(4 + nch + 4 * pass, std::false_type{}, std::false_type{});
            FFN_STAMP();  // projection pass computed (the pass before it stored)
            asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
            // two tiles make one 128-byte row piece; the slab is read back as whole rows
#pragma unroll
            for (int nt = 0; nt < 16; ++nt) {
                const f32x16 t = tile_out(oacc[nt]);
                const int ns = nt & 1;
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    const uint32_t lo0 = pack_elem2<F16>(t[8 * gp], t[8 * gp + 1]), lo1 = pack_elem2<F16>(t[8 * gp + 2], t[8 * gp + 3]);
                    const uint32_t hi0 = pack_elem2<F16>(t[8 * gp + 4], t[8 * gp + 5]), hi1 = pack_elem2<F16>(t[8 * gp + 6], t[8 * gp + 7]);
                    auto s0 = __builtin_amdgcn_permlane32_swap(lo0, hi0, false, false);
                    auto s1 = __builtin_amdgcn_permlane32_swap(lo1, hi1, false, false);
                    const int ch = ns * 4 + 2 * gp + h;
                    *reinterpret_cast<uint4*>(es + wr_row + ((ch ^ wr_sw) << 4)) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                }
                if (pass < 2) tile_in(oacc[nt], bias_tile(pass + 1, nt));
                if (ns == 1) {
                    ffn_wave_lds_sync();
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r = rd_row + 8 * i;
                        const u32x4 v = *reinterpret_cast<const u32x4*>(es + r * 128 + ((rd_c ^ (r & 7)) << 4));
                        __builtin_amdgcn_raw_buffer_store_b128(v, qkv_rsrc, qkv_off[i] + 64 * (nt - 1), 1024 * pass, 0);
                    }
                    ffn_wave_lds_sync();
                }
            }
        }
        FFN_STAMP();  // q, k, v stored
        // the padding chunk is still arriving: the LDS must not be handed to another workgroup before it has landed
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    constexpr bool HAS_Y = OUT == 1;
    __syncthreads();  // every wave is done with the ring: it becomes the epilogue's staging area
    // ---- epilogue: 32-row slabs through a wave-private 4 KiB LDS tile so that global stores move whole
    //      128-byte lines (8 consecutive lanes per row)
    char* es = smem + wave * 8192;  // fp32 slab of x_out
    char* ys = es + 4096;           // 16-bit slab of y
    const int lane_e = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));  // (see the OUT 2 branch)
    const int l31_e = lane_e & 31;
    const int wr_row = l31_e * 128, wr_sw = l31_e & 7;
    const int rd_row = lane_e >> 3, rd_c = lane_e & 7;
    float mean = 0.0f, m2 = 0.0f, rstd = 0.0f, shift = 0.0f;
    RowStat rs;
    if constexpr (HAS_Y) {
        // pass 1: row statistics of x_out
#pragma unroll
        for (int nt = 0; nt < 16; ++nt) {
            const f32x16 t = tile_out(oacc[nt]);
            stat_add_tile(rs, nt, t);
        }
        stat_lane(rs, mean, m2);
        stat_finish_row(mean, m2, p.eps, rstd);
        shift = -mean * rstd;
    }
    // pass 2: x_out and, from the same registers, y = the normalised rows rounded to 16 bits (two tiles make one
    // 128-byte row piece of y)
#pragma unroll
    for (int nt = 0; nt < 16; ++nt) {
        const f32x16 t = tile_out(oacc[nt]);
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(es + wr_row + (((2 * g + h) ^ wr_sw) << 4)) =
                make_float4(t[4 * g], t[4 * g + 1], t[4 * g + 2], t[4 * g + 3]);
        ffn_wave_lds_sync();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = rd_row + 8 * i;
            const float4 v = *reinterpret_cast<const float4*>(es + r * 128 + ((rd_c ^ (r & 7)) << 4));
            const int m = row0 + r;
            if (m < p.M && !no_store) *reinterpret_cast<float4*>(p.x_out + static_cast<int64_t>(m) * 512 + 32 * nt + rd_c * 4) = v;
        }
        ffn_wave_lds_sync();
        if constexpr (HAS_Y) {
            uint16_t* y = reinterpret_cast<uint16_t*>(p.y);
            const int ns = nt & 1;
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = t[8 * gp + e] * rstd + shift;
                const uint32_t lo0 = pack_elem2<F16>(v[0], v[1]), lo1 = pack_elem2<F16>(v[2], v[3]);
                const uint32_t hi0 = pack_elem2<F16>(v[4], v[5]), hi1 = pack_elem2<F16>(v[6], v[7]);
                // lanes 32..63 of lo swap with lanes 0..31 of hi: h = 0 ends with columns 0..7 of the 16, h = 1 with 8..15
                auto s0 = __builtin_amdgcn_permlane32_swap(lo0, hi0, false, false);
                auto s1 = __builtin_amdgcn_permlane32_swap(lo1, hi1, false, false);
                const int ch = ns * 4 + 2 * gp + h;
                *reinterpret_cast<uint4*>(ys + wr_row + ((ch ^ wr_sw) << 4)) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
            }
            if (ns == 1) {
                ffn_wave_lds_sync();
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = rd_row + 8 * i;
                    const uint4 v = *reinterpret_cast<const uint4*>(ys + r * 128 + ((rd_c ^ (r & 7)) << 4));
                    const int m = row0 + r;
                    if (m < p.M && !no_store) *reinterpret_cast<uint4*>(y + static_cast<int64_t>(m) * p.ldy + 32 * (nt - 1) + rd_c * 8) = v;
                }
                ffn_wave_lds_sync();
            }
        }
    }
}

// The dynamic-LDS limit of a kernel is a per-device attribute: asrfe_create calls this with the context's device
// current, before any compute call can run.
int32_t prepare_post_attn(asrfe_ctx* ctx) {
    const void* fns[10] = {reinterpret_cast<const void*>(post_attn_kernel<true, 0, false>),
                           reinterpret_cast<const void*>(post_attn_kernel<true, 1, false>),
                           reinterpret_cast<const void*>(post_attn_kernel<true, 2, false>),
                           reinterpret_cast<const void*>(post_attn_kernel<false, 0, false>),
                           reinterpret_cast<const void*>(post_attn_kernel<false, 1, false>),
                           reinterpret_cast<const void*>(post_attn_kernel<false, 2, false>),
                           reinterpret_cast<const void*>(post_attn_kernel<true, 0, true>),
                           reinterpret_cast<const void*>(post_attn_kernel<true, 2, true>),
                           reinterpret_cast<const void*>(post_attn_kernel<false, 0, true>),
                           reinterpret_cast<const void*>(post_attn_kernel<false, 2, true>)};
    for (const void* f : fns)
        if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return set_err(ctx, ASRFE_ERR_HIP, "post_attn: cannot raise the dynamic LDS limit to 160 KiB");
    return ASRFE_OK;
}

int32_t launch_post_attn(asrfe_ctx* ctx, const FfnParams& p, hipStream_t s) {
    if (p.M <= 0) return ASRFE_OK;
    if (p.ff < 64 || p.ff % 64 != 0 || p.ff > 7168)
        return set_err(ctx, ASRFE_ERR_SHAPE, "post_attn: linear_units must be a multiple of 64 in 64..7168");
    if (!p.att || !p.fsmn || !p.x_out || !p.wstream || !p.bo || !p.b1 || !p.b2 || (p.y && p.ldy % 8 != 0))
        return set_err(ctx, ASRFE_ERR_BAD_ARG, "post_attn: null or misaligned argument");
    if (p.qkv && (p.y || !p.bqkv || p.ldq % 8 != 0 || p.ff > kPostAttnQkvMaxFf))
        return set_err(ctx, ASRFE_ERR_BAD_ARG, "post_attn: qkv output excludes y, needs b_qkv and linear_units <= 5632");
    if (p.tiled && p.y)
        return set_err(ctx, ASRFE_ERR_BAD_ARG, "post_attn: the tiled residual stream goes with the qkv output or with x_out alone");
    // (its stores address rows by signed 32-bit byte offsets, up to one row tile past M)
    if ((p.qkv || p.tiled) && (static_cast<int64_t>(p.M) + kFfnRows) * std::max<int64_t>(2048, 2 * static_cast<int64_t>(p.ldq)) >= (int64_t{1} << 31))
        return set_err(ctx, ASRFE_ERR_SHAPE, "post_attn: x_out and qkv must each stay below 2 GiB per launch");
    const int out = p.qkv ? 2 : (p.y ? 1 : 0);
    const int lds = kFfnRing + std::max(p.ff, kSlabFloats) * 4 + 2 * 512 * 4 + (out == 2 ? 1536 * 4 : 0);
    const dim3 grid((p.M + kFfnRows - 1) / kFfnRows), block(kFfnThreads);
    const bool f16 = ctx->cfg.enc_compute_dtype == ASRFE_DT_F16;
#define POST_ATTN_LAUNCH(F, O, T) hipLaunchKernelGGL((post_attn_kernel<F, O, T>), grid, block, lds, s, p)
    if (p.tiled) {
        if (f16) {
            if (out == 2) POST_ATTN_LAUNCH(true, 2, true);
            else POST_ATTN_LAUNCH(true, 0, true);
        } else {
            if (out == 2) POST_ATTN_LAUNCH(false, 2, true);
            else POST_ATTN_LAUNCH(false, 0, true);
        }
    } else if (f16) {
        if (out == 2) POST_ATTN_LAUNCH(true, 2, false);
        else if (out == 1) POST_ATTN_LAUNCH(true, 1, false);
        else POST_ATTN_LAUNCH(true, 0, false);
    } else {
        if (out == 2) POST_ATTN_LAUNCH(false, 2, false);
        else if (out == 1) POST_ATTN_LAUNCH(false, 1, false);
        else POST_ATTN_LAUNCH(false, 0, false);
    }
#undef POST_ATTN_LAUNCH
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_err(ctx, ASRFE_ERR_HIP, std::string("post_attn launch: ") + hipGetErrorString(e));
    return ASRFE_OK;
}

}  // namespace asrfe
