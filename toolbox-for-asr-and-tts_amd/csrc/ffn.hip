// Fused position-wise feed-forward block of one SAN-M encoder layer on gfx950:
//
//     x_out = x + W2 relu(W1 LN(x) + b1) + b2            (fp32 residual stream, 512 columns)
//     y     = LN'(x_out)  as 16-bit                       (optional: the next layer's norm1)
//
// One workgroup = 128 rows of x, 4 waves, ONE wave per SIMD so that a wave may use the whole 512-entry
// register file.  A wave owns 32 complete rows for the whole kernel:
//   - the 32 x 512 fp32 output tile is 16 accumulators of v_mfma_f32_32x32x16 (256 registers) which start
//     as x + b2, so the residual never leaves registers;
//   - LN(x) is computed inside the wave (a row lives in the two lanes l and l + 32) and kept as the 32
//     MFMA operand fragments of the first GEMM (128 registers);
//   - the hidden layer is produced 64 units at a time: H^T[64][32] = W1_chunk LN(x)^T (2 accumulators),
//     ReLU + rounding turn it straight into the operand fragments of the second GEMM, which adds
//     W2[:, chunk] H^T into the output tile.  The 2048-wide intermediate never exists in memory.
//
// An accumulator holds 4 consecutive rows per register quad (row = 8 g + 4 (lane >> 5) + e), while an
// operand fragment holds 8 consecutive k per lane (k = 8 (lane >> 5) + i).  Feeding accumulator registers
// back as an operand therefore visits k in the order p -> swap of bits 2 and 3 of p inside every 16; the
// host stores the COLUMNS of W1 and W2 in that same order (asrfe_load_encoder_weights), which makes the
// product exact again without moving any data on the device.
//
// Weights stream through LDS as 16 KiB pieces, 8 per chunk of 64 hidden units:
//     pieces 0..3 : W1[chunk rows 64][128 k]    (256-byte rows, 16-byte slot c of row r holds chunk c ^ (r & 15))
//     pieces 4..7 : W2[128 n][chunk cols 64]    (128-byte rows, slot c of row r holds chunk c ^ ((r >> 1) & 7))
// filled by 16-byte LDS-DMA loads with the swizzle applied to the source address.  Piece r of every chunk
// uses ring slot r.  A wave issues 16 MFMAs per piece; the fragment for MFMA i + 8 is read right after
// MFMA i, so in the middle of piece q the wave needs piece q + 1: there it waits (counted vmcnt: five
// younger pieces stay in flight), meets the other waves at one s_barrier, and then refills the slot of
// piece q - 1, whose last read every wave finished before that barrier.
#include <utility>

#include "common.h"
#include "elem.h"
#include "encoder.h"

namespace asrfe {

constexpr int kFfnThreads = 256;
constexpr int kFfnRows = 128;
constexpr int kPieceBytes = 16384;
constexpr int kFfnRing = 8 * kPieceBytes;  // 128 KiB; the fp32 b1 vector follows it

// 16-byte LDS-DMA load: LDS[m0 + 16 * lane] = *gptr.  Issued through inline assembly on purpose: the compiler
// models a load that touches both global memory and LDS as a flat access and, while one is pending, turns
// every wait it inserts into vmcnt(0) / lgkmcnt(0); hidden from it, its LDS-read waits stay counted.  The
// kernel orders these loads itself (wait_vm_barrier).
__device__ __forceinline__ void ffn_glds16(const void* gptr, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_addr), "v"(gptr) : "memory");
}
__device__ __forceinline__ uint32_t lds_addr_of(const void* p) {
    return static_cast<uint32_t>(reinterpret_cast<uintptr_t>(p));  // low half of a generic LDS pointer = LDS offset
}

template <int I>
using IC = std::integral_constant<int, I>;

template <int... I, class Fn>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, Fn&& f) {
    (f(IC<I>{}), ...);
}
template <int N, class Fn>
__device__ __forceinline__ void static_for(Fn&& f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

// all but the N youngest vector-memory operations of this wave are done, then every wave of the workgroup meets
template <int N>
__device__ __forceinline__ void wait_vm_barrier() {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
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

__device__ __forceinline__ float relu(float v) { return __builtin_amdgcn_fmed3f(v, 0.0f, __builtin_inff()); }

template <bool F16>
__device__ __forceinline__ u16x8 pack8(float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7) {
    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
    const u32x4 v = {pack_elem2<F16>(a0, a1), pack_elem2<F16>(a2, a3), pack_elem2<F16>(a4, a5), pack_elem2<F16>(a6, a7)};
    return __builtin_bit_cast(u16x8, v);
}

template <bool F16, bool HAS_Y>
__global__ __launch_bounds__(kFfnThreads) void ffn_kernel(const FfnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int row0 = blockIdx.x * kFfnRows + wave * 32;
    const int nch = p.ff >> 6;
    const uint32_t ffb = static_cast<uint32_t>(p.ff) * 2;  // bytes per row of W2
    float* b1s = reinterpret_cast<float*>(smem + kFfnRing);

    // ---- staging: thread covers 16-byte LDS slot (i * 256 + tid) of a piece, i = 0..3
    const char* w1 = reinterpret_cast<const char*>(p.w1);
    const char* w2 = reinterpret_cast<const char*>(p.w2);
    const uint32_t g1off = static_cast<uint32_t>(tid >> 4) * 1024u + (((tid & 15) ^ ((tid >> 4) & 15)) << 4);
    const uint32_t g2off = static_cast<uint32_t>(tid >> 3) * ffb + (((tid & 7) ^ ((tid >> 4) & 7)) << 4);
    const uint32_t smem_base = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    auto stage = [&](auto rc, int j) {
        constexpr int R = decltype(rc)::value;
        const uint32_t dst = __builtin_amdgcn_readfirstlane(smem_base + R * kPieceBytes + wave * 1024);
        if constexpr (R < 4) {
            const char* src = w1 + static_cast<uint32_t>(j) * 65536u + R * 256 + g1off;
#pragma unroll
            for (int i = 0; i < 4; ++i) ffn_glds16(src + i * 16384, dst + i * 4096);
        } else {
            const char* src = w2 + static_cast<uint32_t>(R - 4) * 128u * ffb + static_cast<uint32_t>(j) * 128u + g2off;
#pragma unroll
            for (int i = 0; i < 4; ++i) ffn_glds16(src + static_cast<uint32_t>(i) * 32u * ffb, dst + i * 4096);
        }
    };
    // the first seven pieces fly while x is loaded and normalised
    stage(IC<0>{}, 0); stage(IC<1>{}, 0); stage(IC<2>{}, 0); stage(IC<3>{}, 0);
    stage(IC<4>{}, 0); stage(IC<5>{}, 0); stage(IC<6>{}, 0);

    for (int i = tid * 4; i < p.ff; i += kFfnThreads * 4)
        *reinterpret_cast<float4*>(b1s + i) = *reinterpret_cast<const float4*>(p.b1 + i);

    // ---- x in accumulator layout: register 4 g + e of tile nt is column 32 nt + 8 g + 4 h + e of row l31.
    //      Three passes, one 32-column tile in working registers at a time (tile_in / tile_out pin the moves).
    const int mrow = min(row0 + l31, p.M - 1);
    const float* xrow = p.x + static_cast<int64_t>(mrow) * 512 + 4 * h;
    f32x16 oacc[16];
    u16x8 afrag[32];  // LN(x) as operand fragments: afrag[2 nt + q] = registers 8 q .. 8 q + 7 of tile nt
    {
        float s = 0.0f;
#pragma unroll
        for (int nt = 0; nt < 16; ++nt) {
            f32x16 t;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 u = *reinterpret_cast<const float4*>(xrow + 32 * nt + 8 * g);
                t[4 * g] = u.x; t[4 * g + 1] = u.y; t[4 * g + 2] = u.z; t[4 * g + 3] = u.w;
                s += (u.x + u.y) + (u.z + u.w);
            }
            tile_in(oacc[nt], t);
        }
        s += __shfl_xor(s, 32);
        const float mean = s * (1.0f / 512.0f);
        float sq = 0.0f;
#pragma unroll
        for (int nt = 0; nt < 16; ++nt) {
            const f32x16 t = tile_out(oacc[nt]);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float dlt = t[e] - mean;
                sq += dlt * dlt;
            }
        }
        sq += __shfl_xor(sq, 32);
        const float rstd = 1.0f / sqrtf(sq * (1.0f / 512.0f) + p.eps);
#pragma unroll
        for (int nt = 0; nt < 16; ++nt) {
            f32x16 t = tile_out(oacc[nt]);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float v[8];
#pragma unroll
                for (int gg = 0; gg < 2; ++gg) {
                    const int col = 32 * nt + 8 * (2 * q + gg) + 4 * h;
                    const float4 gm = *reinterpret_cast<const float4*>(p.ln_g + col);
                    const float4 bt = *reinterpret_cast<const float4*>(p.ln_b + col);
                    v[4 * gg + 0] = (t[8 * q + 4 * gg + 0] - mean) * rstd * gm.x + bt.x;
                    v[4 * gg + 1] = (t[8 * q + 4 * gg + 1] - mean) * rstd * gm.y + bt.y;
                    v[4 * gg + 2] = (t[8 * q + 4 * gg + 2] - mean) * rstd * gm.z + bt.z;
                    v[4 * gg + 3] = (t[8 * q + 4 * gg + 3] - mean) * rstd * gm.w + bt.w;
                }
                afrag[2 * nt + q] = pack8<F16>(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
            }
            // the output tile starts as x + b2
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 u = *reinterpret_cast<const float4*>(p.b2 + 32 * nt + 8 * g + 4 * h);
                t[4 * g] += u.x; t[4 * g + 1] += u.y; t[4 * g + 2] += u.z; t[4 * g + 3] += u.w;
            }
            tile_in(oacc[nt], t);
        }
    }

    // ---- fragment reads
    const int rb1 = l31 * 256, sw1 = (h ^ (l31 & 15)) << 4;
    const int rb2 = l31 * 128, sw2 = (h ^ ((l31 >> 1) & 7)) << 4;
    auto frag = [&](auto rc, auto fc) -> u16x8 {
        constexpr int R = decltype(rc)::value, F = decltype(fc)::value;
        if constexpr (R < 4) {  // W1 rows 32 ct + l31, k-step s of the piece
            constexpr int s = F >> 1, ct = F & 1;
            return *reinterpret_cast<const u16x8*>(smem + R * kPieceBytes + ct * 8192 + rb1 + (sw1 ^ (s << 5)));
        } else {                // W2 rows 32 ntl + l31, k-step kk of the chunk
            constexpr int kk = F >> 2, ntl = F & 3;
            return *reinterpret_cast<const u16x8*>(smem + R * kPieceBytes + ntl * 4096 + rb2 + (sw2 ^ (kk << 5)));
        }
    };

    f32x16 acc1[2];  // H^T of the current chunk: register 4 g + e of tile ct is hidden unit 32 ct + 8 g + 4 h + e
    u16x8 hp[4];     // relu(H) as operand fragments of the second GEMM
    u16x8 wf[8];     // rolling weight fragments
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
    // piece R of chunk j; LAST: j is the final chunk (nothing further to stage)
    auto step = [&](auto rc, int j, auto lastc) {
        constexpr int R = decltype(rc)::value;
        constexpr bool LAST = decltype(lastc)::value;
        static_for<8>([&](auto f) {
            constexpr int F = decltype(f)::value;
            mma(rc, f);
            wf[F] = frag(rc, IC<F + 8>{});
        });
        if constexpr (!LAST) {
            wait_vm_barrier<20>();
            if constexpr (R == 0) stage(IC<7>{}, j);
            else stage(IC<(R + 7) & 7>{}, j + 1);
        } else {
            wait_vm_barrier<(R <= 1 ? 20 : (R <= 6 ? 4 * (6 - R) : 0))>();
            if constexpr (R == 0) stage(IC<7>{}, j);
        }
        // the hidden-chunk accumulators are idle from the ReLU to the end of the chunk: reload them with the next
        // chunk's bias here, eight MFMAs ahead of their first use
        if constexpr (R == 7 && !LAST) init_acc1(j + 1);
        static_for<8>([&](auto f) {
            constexpr int F = decltype(f)::value;
            mma(rc, IC<F + 8>{});
            if constexpr (!(LAST && R == 7)) wf[F] = frag(IC<((R + 1) & 7)>{}, f);
        });
    };
    auto chunk = [&](int j, auto lastc) {
        step(IC<0>{}, j, lastc);
        step(IC<1>{}, j, lastc);
        step(IC<2>{}, j, lastc);
        step(IC<3>{}, j, lastc);
        // The compiler does not see inside the asm MFMAs, so the wait states between an MFMA and a VALU read of
        // its result are covered by hand; the nops are tied to both tiles so that no read can move above them.
        asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc1[0]), "+v"(acc1[1]));
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const f32x16& a = acc1[kk >> 1];
            const int o = 8 * (kk & 1);
            hp[kk] = pack8<F16>(relu(a[o]), relu(a[o + 1]), relu(a[o + 2]), relu(a[o + 3]), relu(a[o + 4]), relu(a[o + 5]),
                                relu(a[o + 6]), relu(a[o + 7]));
        }
        step(IC<4>{}, j, lastc);
        step(IC<5>{}, j, lastc);
        step(IC<6>{}, j, lastc);
        step(IC<7>{}, j, lastc);
    };

    __syncthreads();  // b1 is in LDS
    init_acc1(0);
    wait_vm_barrier<24>();  // piece 0 has landed everywhere
    static_for<8>([&](auto f) { wf[decltype(f)::value] = frag(IC<0>{}, f); });

    for (int j = 0; j < nch - 1; ++j) chunk(j, std::false_type{});
    chunk(nch - 1, std::true_type{});
    __syncthreads();  // every wave is done with the ring: it becomes the epilogue's staging area

    // ---- epilogue: 32-row slabs through a wave-private 4 KiB LDS tile so that global stores move whole
    //      128-byte lines (8 consecutive lanes per row)
    char* es = smem + wave * 8192;  // fp32 slab of x_out
    char* ys = es + 4096;           // 16-bit slab of y
    const int wr_row = l31 * 128, wr_sw = l31 & 7;
    const int rd_row = lane >> 3, rd_c = lane & 7;
    float mean = 0.0f, rstd = 0.0f;
    if constexpr (HAS_Y) {
        float s = 0.0f;
#pragma unroll
        for (int nt = 0; nt < 16; ++nt) {
            const f32x16 t = tile_out(oacc[nt]);
#pragma unroll
            for (int e = 0; e < 16; e += 4) s += (t[e] + t[e + 1]) + (t[e + 2] + t[e + 3]);
        }
        s += __shfl_xor(s, 32);
        mean = s * (1.0f / 512.0f);
        float sq = 0.0f;
#pragma unroll
        for (int nt = 0; nt < 16; ++nt) {
            const f32x16 t = tile_out(oacc[nt]);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float dlt = t[e] - mean;
                sq += dlt * dlt;
            }
        }
        sq += __shfl_xor(sq, 32);
        rstd = 1.0f / sqrtf(sq * (1.0f / 512.0f) + p.eps);
    }
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
            if (m < p.M) *reinterpret_cast<float4*>(p.x_out + static_cast<int64_t>(m) * 512 + 32 * nt + rd_c * 4) = v;
        }
        ffn_wave_lds_sync();
        if constexpr (HAS_Y) {
            // the same tile, normalised and rounded: half of a 128-byte row piece of y (64 columns per pair of tiles)
            uint16_t* y = reinterpret_cast<uint16_t*>(p.y);
            const int ns = nt & 1;
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                float v[8];
#pragma unroll
                for (int gg = 0; gg < 2; ++gg) {
                    const int col = 32 * nt + 16 * gp + 8 * gg + 4 * h;
                    const float4 gm = *reinterpret_cast<const float4*>(p.y_g + col);
                    const float4 bt = *reinterpret_cast<const float4*>(p.y_b + col);
                    v[4 * gg + 0] = (t[8 * gp + 4 * gg + 0] - mean) * rstd * gm.x + bt.x;
                    v[4 * gg + 1] = (t[8 * gp + 4 * gg + 1] - mean) * rstd * gm.y + bt.y;
                    v[4 * gg + 2] = (t[8 * gp + 4 * gg + 2] - mean) * rstd * gm.z + bt.z;
                    v[4 * gg + 3] = (t[8 * gp + 4 * gg + 3] - mean) * rstd * gm.w + bt.w;
                }
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
                    if (m < p.M) *reinterpret_cast<uint4*>(y + static_cast<int64_t>(m) * p.ldy + 32 * (nt - 1) + rd_c * 8) = v;
                }
                ffn_wave_lds_sync();
            }
        }
    }
}

int32_t launch_ffn(asrfe_ctx* ctx, const FfnParams& p, hipStream_t s) {
    if (p.M <= 0) return ASRFE_OK;
    if (p.ff < 64 || p.ff % 64 != 0 || p.ff > 8192)
        return set_err(ctx, ASRFE_ERR_SHAPE, "ffn: linear_units must be a multiple of 64 in 64..8192");
    if (!p.x || !p.x_out || !p.w1 || !p.w2 || !p.b1 || !p.b2 || !p.ln_g || !p.ln_b || (p.y && (!p.y_g || !p.y_b || p.ldy % 8 != 0)))
        return set_err(ctx, ASRFE_ERR_BAD_ARG, "ffn: null or misaligned argument");
    const int lds = kFfnRing + p.ff * 4;
    static bool attr_done = false;
    if (!attr_done) {
        const void* fns[4] = {reinterpret_cast<const void*>(ffn_kernel<false, false>), reinterpret_cast<const void*>(ffn_kernel<false, true>),
                              reinterpret_cast<const void*>(ffn_kernel<true, false>), reinterpret_cast<const void*>(ffn_kernel<true, true>)};
        for (const void* f : fns)
            if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                return set_err(ctx, ASRFE_ERR_HIP, "ffn: cannot raise the dynamic LDS limit to 160 KiB");
        attr_done = true;
    }
    const dim3 grid((p.M + kFfnRows - 1) / kFfnRows), block(kFfnThreads);
    const bool f16 = ctx->cfg.enc_compute_dtype == ASRFE_DT_F16;
    if (f16) {
        if (p.y) hipLaunchKernelGGL((ffn_kernel<true, true>), grid, block, lds, s, p);
        else hipLaunchKernelGGL((ffn_kernel<true, false>), grid, block, lds, s, p);
    } else {
        if (p.y) hipLaunchKernelGGL((ffn_kernel<false, true>), grid, block, lds, s, p);
        else hipLaunchKernelGGL((ffn_kernel<false, false>), grid, block, lds, s, p);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_err(ctx, ASRFE_ERR_HIP, std::string("ffn launch: ") + hipGetErrorString(e));
    return ASRFE_OK;
}

}  // namespace asrfe
