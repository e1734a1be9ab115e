// Masked multi-head self-attention (d_k = 128) and the FSMN memory block of the SAN-M encoder for long query axes:
// persistent workgroups of four waves, ONE wave per SIMD, 64 query rows per wave.
//
//   One workgroup = 4 waves = 256 query rows of one (utterance, head) at a time; one workgroup per CU walks the
//   (utterance, head, 256-row query block) triples of the launch, a contiguous range per XCD so that the workgroups
//   that share a head's K and V hit in the same L2.
//
// The arithmetic is that of attention.hip (scores transposed, S^T = K Q^T, so a lane owns one query row; online
// softmax with a lazily raised maximum; O^T += V^T P^T with V read transposed from LDS).  What differs is who overlaps
// with whom.  There, two workgroups share a CU and nothing keeps the vector work of one out of step with the matrix
// work of the other; here a wave is alone on its SIMD and carries TWO 32-row query groups that run half a key tile
// apart inside one instruction stream, so that the exponentials of one group sit in the shadow of the other group's
// MFMAs (a wave alone on a SIMD hides about four vector instructions behind every MFMA):
//
//     per key tile t        matrix pipe                  vector ALU
//       A                   S0 = K(t) Q0^T               exponentials of group 1, tile t - 1 (second half)
//       B                   O1 += V(t-1)^T P1^T,         row maximum + exponentials of group 0, tile t
//                           S1 = K(t) Q1^T
//       C                   O0 += V(t)^T P0^T            row maximum + exponentials of group 1, tile t (first half)
//
// K tiles live in a ring of two LDS buffers, V tiles in a ring of three (V(t-1) is still read while V(t+1) arrives);
// the loads of tile t + 1 are issued right behind the one workgroup barrier of tile t and have a whole tile to land.
// The rings keep turning across query blocks: during a block's last key tile the next block's Q rows (into the
// wave's private 16 KiB stage), its first K / V tile and its FSMN taps are already on their way, so only the very
// first block of a workgroup sees its memory latency.  The stage then takes the finished O tile on its way to
// whole 256-byte rows in global memory.
//
// FSMN: fsmn[t][c] = mask_t (v_m[t][c] + sum_j w[j][c] v_m[t + j - 5][c]) needs no loads of its own: while tiles
// t - 1 and t of V are both in the ring the four waves compute the 64 output rows [64 t - 32, 64 t + 32) of the
// head's 128 channels from them (a thread owns a channel pair and slides the 11-tap window through registers).
// Window t belongs to the query block that holds most of its rows, the window behind the last tile takes zeros for
// the tile that does not exist, and rows past the last window are stored as zeros by the block that owns them.
#include "common.h"
#include "elem.h"
#include "encoder.h"
#include "lds_dma.h"

namespace asrfe {

namespace {
constexpr int kWThreads = 256;
constexpr int kWRows = 256;                       // query rows per block
constexpr int kWTile = 64 * 256;                  // 64 keys x 128 d, 16-bit
constexpr int kWKOfs = 0;                         // K ring: two tiles
constexpr int kWVOfs = 2 * kWTile;                // V ring: three tiles
constexpr int kWStageOfs = 5 * kWTile;            // four wave-private stages of 16 KiB (Q in, O out)
constexpr int kWTapOfs = 9 * kWTile;              // two buffers for the head's FSMN taps, f32 [11][128]
constexpr int kWTapBytes = 6 * 1024;              // 352 pieces of 16 bytes, rounded up to whole wave-instructions
constexpr int kWLds = kWTapOfs + 2 * kWTapBytes;  // 159744 of the CU's 163840 bytes
constexpr int kWHalo = 5;
constexpr int kWTaps = 11;
static_assert(kWLds <= 160 * 1024, "the wide attention kernel needs (almost) the whole LDS of a CU");

typedef short v4s __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

template <class T>
__device__ __forceinline__ T* uniform(T* ptr) {  // the pointer of the first active lane, as a scalar register pair
    const uint64_t v = reinterpret_cast<uint64_t>(ptr);
    const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v));
    const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v >> 32));
    return reinterpret_cast<T*>(static_cast<uint64_t>(lo) | (static_cast<uint64_t>(hi) << 32));
}

struct WBlock {     // one (utterance, head, query block) triple; everything here is wave-uniform
    int b;          // utterance
    int head;
    int qbi;        // index of the query block
    int qb0;        // its first query row
    int len;        // valid keys
    int n_kt;       // key tiles
};
}  // namespace

template <bool F16>
__global__ __launch_bounds__(kWThreads, 1) void attn_wide_kernel(const AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef ASRFE_ATTN_ABLATE
    // experiments (results are wrong when a bit is set): 1 no K / V requests inside the key loop, 2 no exponentials,
    // 4 no row maxima, 8 no V fragment reads, 16 no K fragment reads, 1024 stamps inside a tile as well
    const int xf = p.flags;
#elif defined(ASRFE_ATTN_TIMING)
    const int xf = p.flags & 1024;
#else
    constexpr int xf = 0;
#endif
    saturate_fp16_conversions<F16>();
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const uint32_t smem_base = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    char* const stage = smem + kWStageOfs + wave * kWTile;
    const uint32_t stage_addr = __builtin_amdgcn_readfirstlane(smem_base + kWStageOfs + wave * kWTile);
    const uint32_t ldb = static_cast<uint32_t>(p.ld_kv) * 2;   // bytes per key / value row
    const uint32_t ldqb = static_cast<uint32_t>(p.ld_q) * 2;   // bytes per query row
    const float c = p.scale_log2;

    // ---- the blocks of this workgroup: a contiguous range per XCD and round (workgroups are dealt round-robin
    //      over the 8 XCDs), query block fastest, so that the blocks of one head run side by side on one XCD
    const int G = gridDim.x, bid = blockIdx.x;
    const int n_blocks = p.B * 4 * p.n_qtiles;
    const int first = (G & 7) == 0 ? (bid & 7) * (G >> 3) + (bid >> 3) : bid;
    auto decode = [&](int L) __attribute__((always_inline)) {
        WBlock k;
        k.qbi = __builtin_amdgcn_readfirstlane(L % p.n_qtiles);
        k.qb0 = k.qbi * kWRows;
        k.head = __builtin_amdgcn_readfirstlane((L / p.n_qtiles) & 3);
        k.b = __builtin_amdgcn_readfirstlane(L / (p.n_qtiles * 4));
        int len = p.lens[k.b];
        len = len < 0 ? 0 : (len > p.Tk ? p.Tk : len);
        k.len = __builtin_amdgcn_readfirstlane(len);
        k.n_kt = (k.len + 63) >> 6;
        return k;
    };
    // base addresses of a block's rows, computed where they are used (scalar arithmetic; held in registers for a whole
    // block they crowd the scalar file).  Wave-uniform by construction; said explicitly, because the LDS-DMA base must
    // be a scalar register pair.
    auto q_base = [&](const WBlock& k) __attribute__((always_inline)) {
        return uniform(reinterpret_cast<const char*>(p.q + static_cast<int64_t>(k.b) * p.Tq * p.ld_q + k.head * 128));
    };
    auto k_base = [&](const WBlock& k) __attribute__((always_inline)) {
        return uniform(reinterpret_cast<const char*>(p.k + static_cast<int64_t>(k.b) * p.Tk * p.ld_kv + k.head * 128));
    };
    auto v_base = [&](const WBlock& k) __attribute__((always_inline)) {
        return uniform(reinterpret_cast<const char*>(p.v + static_cast<int64_t>(k.b) * p.Tk * p.ld_kv + k.head * 128));
    };
    auto o_base = [&](const WBlock& k) __attribute__((always_inline)) {
        return uniform(p.out + static_cast<int64_t>(k.b) * p.Tq * p.ld_out + k.head * 128);
    };
    auto f_base = [&](const WBlock& k) __attribute__((always_inline)) {  // (only called with p.fsmn set)
        return uniform(p.fsmn + static_cast<int64_t>(k.b) * p.Tk * 512 + k.head * 128);
    };
#ifdef ASRFE_ATTN_TIMING
    const bool with_fsmn = p.fsmn && !(p.flags & 512);
#else
    const bool with_fsmn = p.fsmn != nullptr;
#endif

    // ---- LDS-DMA requests (16 bytes per lane, 1 KiB per wave-instruction; counted by hand: see lds_dma.h) --------
    // K / V tile: thread covers LDS slot (row i * 16 + (tid >> 4), chunk tid & 15), i = 0..3.  K chunk c of row r sits at
    // chunk c ^ (r & 15) (row reads), V chunk c of row r at c ^ (((r & 3) << 2) | ((r >> 2) & 3)) (transposed reads).
    // (the per-lane offsets are derived from an opaque copy of the thread id at every use: computed once per block they
    // would be kept in registers across the key loop, which has none to spare)
    auto opaque_tid = [&]() __attribute__((always_inline)) {
        int t = tid;
        asm volatile("" : "+v"(t));
        return t;
    };
    // K / V tile requests: four lane constants stay in registers (this thread's byte offset inside a 16-row group of
    // K and of V, and the same for the utterance's last row: rows past it are clamped to it and masked when used)
    const uint32_t st_row = tid >> 4;
    const uint32_t koff_l = st_row * ldb + (((tid & 15) ^ ((tid >> 4) & 15)) << 4);
    const uint32_t voff_l = st_row * ldb + (((tid & 15) ^ ((((tid >> 4) & 3) << 2) | ((tid >> 6) & 3))) << 4);
    const uint32_t last_row = static_cast<uint32_t>(p.Tk - 1) * ldb;
    const uint32_t kclamp = last_row + (koff_l - st_row * ldb), vclamp = last_row + (voff_l - st_row * ldb);
    // load j of a tile: j >> 1 = row group i, j & 1 = V
    auto kv_issue_one = [&](const WBlock& k, int t, int kslot, int vslot, auto jc) __attribute__((always_inline)) {
        constexpr int J = decltype(jc)::value, I = J >> 1;
        const uint32_t rowofs = static_cast<uint32_t>(t * 64 + I * 16) * ldb;  // (scalar)
        if constexpr ((J & 1) == 0) {
            const uint32_t o = koff_l + rowofs;
            const uint32_t dst = __builtin_amdgcn_readfirstlane(smem_base + kWKOfs + kslot * kWTile + wave * 1024);
            lds_dma16(k_base(k), o < kclamp ? o : kclamp, dst + I * 4096);
        } else {
            const uint32_t o = voff_l + rowofs;
            const uint32_t dst = __builtin_amdgcn_readfirstlane(smem_base + kWVOfs + vslot * kWTile + wave * 1024);
            lds_dma16(v_base(k), o < vclamp ? o : vclamp, dst + I * 4096);
        }
    };
    // Q rows of this wave, 16 pieces of 4 rows: row r = 4 i + (lane >> 4) of the wave's 64, chunk c at c ^ (r & 15)
    auto q_issue_one = [&](const WBlock& k, auto ic) __attribute__((always_inline)) {
        constexpr int I = decltype(ic)::value;
        const int ll = opaque_tid() & 63;
        int qr = k.qb0 + wave * 64 + 4 * I + (ll >> 4);
        qr = qr < p.Tq ? qr : p.Tq - 1;
        const uint32_t chunk = (ll & 15) ^ (4 * (I & 3) + (ll >> 4));
        lds_dma16(q_base(k), static_cast<uint32_t>(qr) * ldqb + chunk * 16, stage_addr + I * 1024);
    };
    // the taps of the head's 128 channels: 11 rows of 512 bytes = 352 pieces; piece q = (tap q / 32, channels
    // 4 (q % 32) .. +3), one per thread and a second one for the first 96 threads
    auto taps_issue = [&](const WBlock& k, int tbuf) __attribute__((always_inline)) {
        const int tt = opaque_tid();
        const char* wbase = uniform(reinterpret_cast<const char*>(p.w_fsmn + k.head * 128));
        const uint32_t tdst = __builtin_amdgcn_readfirstlane(smem_base + kWTapOfs + tbuf * kWTapBytes + wave * 1024);
        lds_dma16(wbase, static_cast<uint32_t>((tt >> 5) * 2048 + (tt & 31) * 16), tdst);
        if (tt < 96) lds_dma16(wbase, static_cast<uint32_t>(((tt + 256) >> 5) * 2048 + (tt & 31) * 16), tdst + 4096);
    };

    // ---- per-lane LDS read addresses (kept for the whole kernel and moved from ring slot to ring slot in place) ----
    // K fragment of k-step s, key sub-tile kk: row 32 kk + l31, chunk (2 s + h) ^ (l31 & 15)
    const int ksw = l31 & 15;
    uint32_t kaddr[8];   // this lane's K fragment of k-step s in the K slot of the tile at hand (sub-tile 1: + 8 KiB)
#pragma unroll
    for (int s = 0; s < 8; ++s) kaddr[s] = smem_base + kWKOfs + l31 * 256 + (((2 * s + h) ^ ksw) << 4);
    // transposed V reads: lane 4 q + pp of a 16-lane group addresses key row q of a 4-key block, channels 4 pp .. + 3 of
    // the group's 16; row r = 16 n + 8 hi + 4 h + q keeps chunk c at c ^ ((q << 2) | (2 hi + h))
    uint32_t vaddr[2][4][2];  // [0: V(t - 1), 1: V(t)][32-channel tile][half]
    {
        const int q = (lane >> 2) & 3, pp = lane & 3, g1 = (lane >> 4) & 1;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int hi = 0; hi < 2; ++hi) {
                const int chunk = ((mt ^ q) << 2) | ((2 * g1 + (pp >> 1)) ^ (2 * hi + h));
                const uint32_t o = smem_base + kWVOfs + (8 * hi + 4 * h + q) * 256 + chunk * 16 + 8 * (pp & 1);
                vaddr[1][mt][hi] = o + 2 * kWTile;  // "the tile before the first": slot 2 (zeroed in the prologue)
                vaddr[0][mt][hi] = o + 1 * kWTile;
            }
    }

    // ---- state of the two query groups --------------------------------------------------------------------------
    // The MFMAs are inline assembly so that every operand's register class is fixed: Q fragments and the O tiles live
    // in the accumulation registers (192 of them), scores and probabilities in architectural VGPRs where the vector
    // ALU works on them.  Inline assembly also means that nobody pads the hazards around these instructions: the
    // comments at each guard() say which one it covers.
    u16x8 qf[2][8];    // Q fragments, pre-multiplied by log2(e) / sqrt(d_k): lane holds Q[q0 + 32 g + l31][16 s + 8 h .. + 8)
    f32x16 S[2][2];    // scores of the current tile minus the running maximum, in log2 units: [group][key sub-tile]
    f32x16 negm[2];    // minus the running maximum of the lane's row, in all 16 elements: the C operand that starts a score tile
    u16x8 P[2][4];     // probabilities as operand fragments: [group][2 kk + s2]
    f32x16 O[2][4];    // O^T accumulators: [group][32-channel tile]
    float l_run[2];

    // Instruction helpers.  LDS reads, their waits, the MFMAs and the exponentials are all assembly: the kernel counts
    // its own LDS operations (they return in order; lgkmcnt(N) = "all but my N youngest"), which keeps a fragment's
    // read several MFMAs ahead of its use without the compiler draining the queue at every reuse of a register.
    // Between a read and the statement that waits for it nothing may touch the destination (tools/check_loop_asm.py
    // looks for compiler moves of those registers and for scalar loads, which share the counter, inside the loop).
#define AW_MFMA(suffix) (F16 ? "v_mfma_f32_32x32x16_f16 " suffix : "v_mfma_f32_32x32x16_bf16 " suffix)
    // (the Q fragment is named as an in-out operand although the instruction only reads it: a value that every use
    // redefines in the accumulation registers stays there; as a plain input the register allocator kept the fragments
    // in architectural VGPRs and copied one over in front of every MFMA)
    auto mfma_s0 = [&](auto nc, f32x16& d, const u16x8& a, u16x8& bq, const f32x16& c0) __attribute__((always_inline)) {
        constexpr int N = decltype(nc)::value;
        if constexpr (F16) asm volatile("s_waitcnt lgkmcnt(%4)\n\tv_mfma_f32_32x32x16_f16 %0, %2, %1, %3" : "=&v"(d), "+a"(bq) : "v"(a), "v"(c0), "n"(N));
        else asm volatile("s_waitcnt lgkmcnt(%4)\n\tv_mfma_f32_32x32x16_bf16 %0, %2, %1, %3" : "=&v"(d), "+a"(bq) : "v"(a), "v"(c0), "n"(N));
    };
    auto mfma_s = [&](auto nc, f32x16& d, const u16x8& a, u16x8& bq) __attribute__((always_inline)) {
        constexpr int N = decltype(nc)::value;
        if constexpr (F16) asm volatile("s_waitcnt lgkmcnt(%3)\n\tv_mfma_f32_32x32x16_f16 %0, %2, %1, %0" : "+v"(d), "+a"(bq) : "v"(a), "n"(N));
        else asm volatile("s_waitcnt lgkmcnt(%3)\n\tv_mfma_f32_32x32x16_bf16 %0, %2, %1, %0" : "+v"(d), "+a"(bq) : "v"(a), "n"(N));
    };
    auto mfma_o = [&](auto nc, f32x16& d, const v4s& alo, const v4s& ahi, const u16x8& b) __attribute__((always_inline)) {
        constexpr int N = decltype(nc)::value;
        const u16x8 a = __builtin_bit_cast(u16x8, __builtin_shufflevector(alo, ahi, 0, 1, 2, 3, 4, 5, 6, 7));
        if constexpr (F16) asm volatile("s_waitcnt lgkmcnt(%3)\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(d) : "v"(a), "v"(b), "n"(N));
        else asm volatile("s_waitcnt lgkmcnt(%3)\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(d) : "v"(a), "v"(b), "n"(N));
    };
    auto lds_read128 = [&](u16x8& d, uint32_t addr, auto offc) __attribute__((always_inline)) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(decltype(offc)::value));
    };
    auto lds_read_tr = [&](v4s& d, uint32_t addr, auto offc) __attribute__((always_inline)) {
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(decltype(offc)::value));
    };

    // the next block's Q rows from the wave's stage, scaled, into the accumulation registers
    auto read_q = [&]() __attribute__((always_inline)) {
        const int ll = opaque_tid() & 63;  // (addresses from an opaque lane id: see opaque_tid)
        const char* qrow = stage + (ll & 31) * 256;
        const int hh = ll >> 5, sw = ll & 15;
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                u16x8 raw = *reinterpret_cast<const u16x8*>(qrow + g * 8192 + (((2 * s + hh) ^ sw) << 4));
                if constexpr (F16) {
                    f16x8 v = __builtin_bit_cast(f16x8, raw);
                    v = v * static_cast<_Float16>(c);
                    raw = __builtin_bit_cast(u16x8, v);
                } else {
#pragma unroll
                    for (int j = 0; j < 8; j += 2) {
                        const uint32_t w = pack_elem2<false>(__uint_as_float(static_cast<uint32_t>(raw[j]) << 16) * c,
                                                             __uint_as_float(static_cast<uint32_t>(raw[j + 1]) << 16) * c);
                        raw[j] = static_cast<uint16_t>(w & 0xffffu);
                        raw[j + 1] = static_cast<uint16_t>(w >> 16);
                    }
                }
                qf[g][s] = raw;
                asm volatile("" : "+a"(qf[g][s]));
            }
    };

    // ---- the segments of a key tile ---------------------------------------------------------------------------
    // A segment is 16 MFMAs whose operand fragments are read from LDS three MFMA pairs (score product) or six
    // MFMAs (O update) ahead.  pre() issues the first reads; it is called from a late gap of the segment before
    // (PRE_AT: the gap, PRE_N: how many reads that adds to this wave's queue), so that a segment starts with its
    // fragments on their way.  gap(i) is called behind MFMA i = 0..15 with whatever else belongs there.
    u16x8 kf[3][2];   // K fragments in flight: [k-step % 3][key sub-tile]
    v4s vfl[8], vfh[8];  // V fragments in flight: [MFMA % 8], the two halves of a transposed read
    // S[g] = K(t) Q_g^T - m
    auto qk_pre = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 3; ++s) if (!(xf & 16)) { lds_read128(kf[s][0], kaddr[s], IC<0>{}); lds_read128(kf[s][1], kaddr[s], IC<8192>{}); }
    };
    auto qk_run = [&](auto gc, auto preat, auto pren, auto&& gap) __attribute__((always_inline)) {
        constexpr int g = decltype(gc)::value, PRE_AT = decltype(preat)::value, PRE_N = decltype(pren)::value;
        static_for<8>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            // reads younger than fragment (s, kk) when its MFMA issues: the rest of the three k-steps in flight
            constexpr int Y0 = (s <= 5 ? 5 : s == 6 ? 3 : 1) + (2 * s > PRE_AT ? PRE_N : 0);
            constexpr int Y1 = (s <= 5 ? 4 : s == 6 ? 2 : 0) + (2 * s + 1 > PRE_AT ? PRE_N : 0);
            if constexpr (s == 0) mfma_s0(IC<Y0>{}, S[g][0], kf[0][0], qf[g][0], negm[g]);
            else mfma_s(IC<Y0>{}, S[g][0], kf[s % 3][0], qf[g][s]);
            gap(IC<2 * s>{});
            if constexpr (s == 0) mfma_s0(IC<Y1>{}, S[g][1], kf[0][1], qf[g][0], negm[g]);
            else mfma_s(IC<Y1>{}, S[g][1], kf[s % 3][1], qf[g][s]);
            if constexpr (s + 3 < 8) if (!(xf & 16)) { lds_read128(kf[s % 3][0], kaddr[s + 3], IC<0>{}); lds_read128(kf[s % 3][1], kaddr[s + 3], IC<8192>{}); }
            gap(IC<2 * s + 1>{});
        });
    };
    // O[g] += V^T P_g^T over the whole 64-key tile: MFMA i = 8 kk + 4 s2 + mt
    auto pv_read = [&](auto wc, auto ic) __attribute__((always_inline)) {
        constexpr int w = decltype(wc)::value, i = decltype(ic)::value;
        // element j of the fragment is key 32 kk + 16 s2 + 8 (j >> 2) + 4 h + (j & 3): rows 16 (i >> 2) .. of the tile
        if (xf & 8) return;
        lds_read_tr(vfl[i % 8], vaddr[w][i & 3][0], IC<(i >> 2) * 16 * 256>{});
        lds_read_tr(vfh[i % 8], vaddr[w][i & 3][1], IC<(i >> 2) * 16 * 256>{});
    };
    auto pv_pre = [&](auto wc) __attribute__((always_inline)) {
        static_for<6>([&](auto ic) { pv_read(wc, ic); });
    };
    auto pv_run = [&](auto gc, auto wc, auto preat, auto pren, auto&& gap) __attribute__((always_inline)) {
        constexpr int g = decltype(gc)::value, PRE_AT = decltype(preat)::value, PRE_N = decltype(pren)::value;
        static_for<16>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr int Y = 2 * (i <= 10 ? 5 : 15 - i) + (i > PRE_AT ? PRE_N : 0);  // fragments i + 1 .. i + 5 are younger
            if constexpr ((i & 7) == 0) {
                // (the probabilities were written by the vector ALU: two wait states before an MFMA may read them)
                u16x8 &p0 = P[g][i >> 2], &p1 = P[g][(i >> 2) + 1];
                asm volatile("s_nop 1" : "+v"(p0), "+v"(p1));
            }
            mfma_o(IC<Y>{}, O[g][i & 3], vfl[i % 8], vfh[i % 8], P[g][i >> 2]);
            if constexpr (i + 6 < 16) pv_read(wc, IC<i + 6>{});
            gap(ic);
        });
    };
    // The scores of group g were written by MFMAs in assembly: nothing of them may be read before this statement, which
    // the caller places behind two later MFMAs (the matrix pipe completes in order, so the score tile is written by then).
    auto scores_ready = [&](auto gc) __attribute__((always_inline)) {
        constexpr int g = decltype(gc)::value;
        f32x16 &s0 = S[g][0], &s1 = S[g][1];
        asm volatile("s_nop 0" : "+v"(s0), "+v"(s1));
    };
    // Row maximum of group g's tile (S already holds score - running maximum), a quarter of the 32 scores per step
    // (assembly: stays in its gap, and no canonicalising moves in front of the maxima)
    // (kmask >= 0: the last tile of a block, where keys at or past kmask = length - first key of the tile get -inf)
    float rmx[2];
    auto max_step = [&](auto gc, auto ic, int kmask) __attribute__((always_inline)) {
        constexpr int g = decltype(gc)::value, i = decltype(ic)::value, kk = i >> 1, e0 = 8 * (i & 1);
        if (xf & 4) { rmx[g] = 0.0f; return; }
        if (kmask >= 0) {
#pragma unroll
            for (int e = e0; e < e0 + 8; ++e)
                if (kk * 32 + (e & 3) + 8 * (e >> 2) + 4 * h >= kmask) S[g][kk][e] = -INFINITY;
        }
        float m, m2;
        asm volatile("v_max3_f32 %0, %2, %3, %4\n\tv_max3_f32 %1, %5, %6, %7\n\tv_max3_f32 %0, %0, %8, %9"
                     : "=&v"(m), "=&v"(m2)
                     : "v"(S[g][kk][e0]), "v"(S[g][kk][e0 + 1]), "v"(S[g][kk][e0 + 2]), "v"(S[g][kk][e0 + 3]), "v"(S[g][kk][e0 + 4]),
                       "v"(S[g][kk][e0 + 5]), "v"(S[g][kk][e0 + 6]), "v"(S[g][kk][e0 + 7]));
        if constexpr (i == 0) asm volatile("v_max_f32 %0, %1, %2" : "=v"(rmx[g]) : "v"(m), "v"(m2));
        else asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(rmx[g]) : "v"(m), "v"(m2));
    };
    // The decision whether the running maximum moves.  It only has to bound the scores well enough for exp2 to stay in
    // range, so it moves (and O, l are rescaled) only when the tile's maximum exceeds it by more than 2^8, or on a
    // block's first tile, where it is set.  No P of this group is pending when this runs (its last O update is complete:
    // a whole segment of other MFMAs lies between), so the rescale is applied on the spot.
    auto max_decide = [&](auto gc, bool first_tile) __attribute__((always_inline)) {
        constexpr int g = decltype(gc)::value;
        float rmax = rmx[g];
        {   // the other 32 keys of the row sit in lane ^ 32
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(rmax), __float_as_uint(rmax), false, false);
            float a = __uint_as_float(sw[0]);
            asm volatile("v_max_f32 %0, %0, %1" : "+v"(a) : "v"(__uint_as_float(sw[1])));
            rmax = a;
        }
        const bool grew = first_tile || rmax > 8.0f;
        if (__any(grew)) {
            const float delta = grew ? rmax : 0.0f;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int e = 0; e < 16; ++e) S[g][kk][e] -= delta;
#pragma unroll
            for (int e = 0; e < 16; ++e) negm[g][e] -= delta;
            if (!first_tile) {  // (on the first tile O and l are zero)
                const float alpha = __builtin_amdgcn_exp2f(-delta);
                l_run[g] *= alpha;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) O[g][mt][e] *= alpha;
                    f32x16& o = O[g][mt];
                    asm volatile("" : "+a"(o));
                }
            }
        }
    };
    // P = exp2(S) for scores 2 q and 2 q + 1 of the group's 32 (q = 0..15).  One statement per pair: the two
    // exponentials of this pair and, on the pair before (whose exponentials are complete by now), the two additions to
    // the row sums and the conversion into the operand fragment.  exp_flush() does the latter for pair 15.
    float pe[2][2];        // the exponentials of the pair before: [group][even / odd score]
    float psum[2][2];
    uint32_t Pw[2][16];    // the probabilities, two to a word: P[g][i] = words 4 i .. 4 i + 3
    auto exp_pair = [&](auto gc, auto qc) __attribute__((always_inline)) {
        constexpr int g = decltype(gc)::value, q = decltype(qc)::value, kk = q >> 3, r = 2 * (q & 7);
        if (xf & 2) return;
        float n0, n1;
        if constexpr (q == 0) {
            asm volatile("v_exp_f32 %0, %2\n\tv_exp_f32 %1, %3" : "=&v"(n0), "=&v"(n1) : "v"(S[g][kk][r]), "v"(S[g][kk][r + 1]));
            psum[g][0] = psum[g][1] = 0.0f;
        } else if constexpr (F16) {
            asm volatile("v_exp_f32 %0, %5\n\tv_exp_f32 %1, %6\n\tv_add_f32 %2, %2, %7\n\tv_add_f32 %3, %3, %8\n\tv_cvt_pk_f16_f32 %4, %7, %8"
                         : "=&v"(n0), "=&v"(n1), "+v"(psum[g][0]), "+v"(psum[g][1]), "=&v"(Pw[g][q - 1])
                         : "v"(S[g][kk][r]), "v"(S[g][kk][r + 1]), "v"(pe[g][0]), "v"(pe[g][1]));
        } else {
            asm volatile("v_exp_f32 %0, %5\n\tv_exp_f32 %1, %6\n\tv_add_f32 %2, %2, %7\n\tv_add_f32 %3, %3, %8\n\tv_cvt_pk_bf16_f32 %4, %7, %8"
                         : "=&v"(n0), "=&v"(n1), "+v"(psum[g][0]), "+v"(psum[g][1]), "=&v"(Pw[g][q - 1])
                         : "v"(S[g][kk][r]), "v"(S[g][kk][r + 1]), "v"(pe[g][0]), "v"(pe[g][1]));
        }
        pe[g][0] = n0;
        pe[g][1] = n1;
    };
    auto exp_flush = [&](auto gc) __attribute__((always_inline)) {
        constexpr int g = decltype(gc)::value;
        // (v_exp_f32 is a transcendental: one wait state before another instruction may read its result)
        if constexpr (F16) asm volatile("s_nop 0\n\tv_add_f32 %0, %0, %3\n\tv_add_f32 %1, %1, %4\n\tv_cvt_pk_f16_f32 %2, %3, %4"
                                        : "+v"(psum[g][0]), "+v"(psum[g][1]), "=&v"(Pw[g][15]) : "v"(pe[g][0]), "v"(pe[g][1]));
        else asm volatile("s_nop 0\n\tv_add_f32 %0, %0, %3\n\tv_add_f32 %1, %1, %4\n\tv_cvt_pk_bf16_f32 %2, %3, %4"
                          : "+v"(psum[g][0]), "+v"(psum[g][1]), "=&v"(Pw[g][15]) : "v"(pe[g][0]), "v"(pe[g][1]));
        l_run[g] += psum[g][0] + psum[g][1];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 w = {Pw[g][4 * i], Pw[g][4 * i + 1], Pw[g][4 * i + 2], Pw[g][4 * i + 3]};
            P[g][i] = __builtin_bit_cast(u16x8, w);
        }
    };

    // FSMN window t of a block: output rows [64 t - 32, 64 t + 32) from V tiles t - 1 (LDS address vprev) and t (vcur);
    // this wave takes 16 of them.  Input row j of the wave's 26 is row rel = 27 + 16 wave + j counted from the first row
    // of tile t - 1.  Rows outside the utterance count as zeros (by select: the LDS behind them may hold anything).
    auto fsmn_window = [&](const WBlock& k, int t, int vprev, int vcur, int tbuf) __attribute__((always_inline)) {  // vprev, vcur: byte offsets in smem
        const int cp = opaque_tid() & 63;  // channel pair (addresses from an opaque lane id: see opaque_tid)
        const char* taps = smem + kWTapOfs + tbuf * kWTapBytes;
        v2f wt[kWTaps];
#pragma unroll
        for (int j = 0; j < kWTaps; ++j) {
            const float2 w = *reinterpret_cast<const float2*>(taps + j * 512 + 8 * cp);
            wt[j] = v2f{w.x, w.y};
        }
        const int relb = 27 + 16 * wave;
        v2f xw[16 + kWTaps - 1];
#pragma unroll
        for (int j = 0; j < 16 + kWTaps - 1; ++j) {
            const int rel = relb + j;
            const int tt = 64 * (t - 1) + rel;
            const int rowofs = rel < 64 ? vprev + rel * 256 : vcur + (rel - 64) * 256;
            const int r16 = (11 + j) & 15;  // = rel & 15 for every wave
            const int swz = ((r16 & 3) << 2) | ((r16 >> 2) & 3);
            const uint32_t raw = *reinterpret_cast<const uint32_t*>(smem + rowofs + ((((cp >> 2) ^ swz) << 4) | ((cp & 3) << 2)));
            float x0, x1;
            unpack_elem2<F16>(raw, x0, x1);
            const bool ok = tt >= 0 && tt < k.len;
            xw[j] = v2f{ok ? x0 : 0.0f, ok ? x1 : 0.0f};
        }
        uint32_t* out = reinterpret_cast<uint32_t*>(f_base(k)) + cp;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int r = 64 * t - 32 + 16 * wave + i;
            v2f acc = xw[i + kWHalo];
#pragma unroll
            for (int j = 0; j < kWTaps; ++j) acc = __builtin_elementwise_fma(wt[j], xw[i + j], acc);
            if (r >= 0 && r < p.Tk) out[static_cast<int64_t>(r) * 256] = r < k.len ? pack_elem2<F16>(acc.x, acc.y) : 0u;
        }
    };
    // which query block computes window t: the one that holds its first valid row
    auto window_owner = [](int t) { return t == 0 ? 0 : (t - 1) >> 2; };

#ifdef ASRFE_ATTN_TIMING
    // experiment (flag 512): wave 0 of workgroup 37 writes cycle-counter stamps into the fsmn buffer (no FSMN then)
    long long* trace = reinterpret_cast<long long*>(p.fsmn);
    int trace_n = 0;
    const bool tracing = (p.flags & 512) && bid == 37 && wave == 0;
#define AW_STAMP() do { if (tracing) { const long long tsc = __builtin_readcyclecounter(); if (lane == 0 && trace_n < 120) trace[trace_n] = tsc; ++trace_n; } } while (0)
#else
#define AW_STAMP() do {} while (0)
#endif
    // ---- prologue: the first block's Q, first tile and taps --------------------------------------------------------
    if (first >= n_blocks) return;
    WBlock cur = decode(first);
    int gt = 0;    // tiles issued so far by this workgroup: tile number gt sits in K slot gt & 1, V slot gt % 3
    int vs3 = 0;   // gt % 3
    int tbuf = 0;
    {
        // the V slot "before" the first tile feeds the first (empty) O update of group 1: give it finite contents
        uint4* z = reinterpret_cast<uint4*>(smem + kWVOfs + 2 * kWTile);
#pragma unroll
        for (int i = 0; i < 4; ++i) z[i * 256 + tid] = make_uint4(0, 0, 0, 0);
        static_for<16>([&](auto ic) { q_issue_one(cur, ic); });
        if (cur.n_kt > 0) static_for<8>([&](auto jc) { kv_issue_one(cur, 0, 0, 0, jc); });
        if (with_fsmn) taps_issue(cur, 0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        read_q();
    }

    for (int L = first;;) {
        const int Ln = L + G;
        const bool has_next = Ln < n_blocks;
        WBlock nxt = cur;
        if (has_next) nxt = decode(Ln);
        // the next block's requests, spread over the last tile (or issued in one go when this block has no tile)
        // (kslot, vslot: the ring slots of the next tile this workgroup will consume)
        auto next_issue = [&](auto jc, int kslot, int vslot) __attribute__((always_inline)) {
            constexpr int J = decltype(jc)::value;  // 0..7: one per k-step of the first score product
            if (!has_next) return;
            q_issue_one(nxt, IC<2 * J>{});
            q_issue_one(nxt, IC<2 * J + 1>{});
            if (nxt.n_kt > 0) kv_issue_one(nxt, 0, kslot, vslot, jc);
            if (J == 7 && with_fsmn) taps_issue(nxt, tbuf ^ 1);
        };

#pragma unroll
        for (int g = 0; g < 2; ++g) {
            l_run[g] = 0.0f;
            psum[g][0] = psum[g][1] = 0.0f;
            pe[g][0] = pe[g][1] = 0.0f;
#pragma unroll
            for (int i = 0; i < 16; ++i) Pw[g][i] = 0u;
#pragma unroll
            for (int e = 0; e < 16; ++e) negm[g][e] = 0.0f;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
                for (int e = 0; e < 16; ++e) O[g][mt][e] = 0.0f;
                asm volatile("" : "+a"(O[g][mt]));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) P[g][i][e] = 0;
        }
        // group 1 starts half a tile behind: its "tile -1" has scores of -inf everywhere (probability 0)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int e = 0; e < 16; ++e) S[1][kk][e] = -INFINITY;

        auto key_tile = [&](const int t, auto lastc) __attribute__((always_inline)) {
            constexpr bool last = decltype(lastc)::value;
            // tile t (requested one tile ago) is complete in LDS: every wave saw its own requests land at the end of the
            // tile before (or in the prologue) and has now passed the barrier; behind the barrier nobody reads tile
            // t - 1's K or tile t - 2's V any more
            AW_STAMP();  // 0: tile start
            __builtin_amdgcn_s_barrier();
            if (xf & 1024) AW_STAMP();  // 1: past the barrier
            const int kslot = gt & 1, vslot = vs3;
            const int vprev = vs3 == 0 ? 2 : vs3 - 1, vnext = vs3 == 2 ? 0 : vs3 + 1;
            const bool first_tile = t == 0;
            const int kmask = last ? cur.len - t * 64 : -1;  // (compile-time -1 on every tile but a block's last)
            {   // the read addresses follow the rings: K toggles between its two slots, V(t) becomes V(t - 1)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int hi = 0; hi < 2; ++hi) {
                        vaddr[0][mt][hi] = vaddr[1][mt][hi];
                        vaddr[1][mt][hi] += vslot == 0 ? static_cast<uint32_t>(-2 * kWTile) : static_cast<uint32_t>(kWTile);
                    }
            }
            qk_pre();

            // A: S0 = K(t) Q0^T beside 12 of group 1's 16 exponential pairs; the next tile's loads in the other gaps
            qk_run(IC<0>{}, IC<13>{}, IC<12>{}, [&](auto ic) {
                constexpr int i = decltype(ic)::value;
                if constexpr ((i & 3) != 3) exp_pair(IC<1>{}, IC<4 + 3 * (i >> 2) + (i & 3)>{});
                else {
                    if constexpr (last) { next_issue(IC<2 * (i >> 2)>{}, kslot ^ 1, vnext); next_issue(IC<2 * (i >> 2) + 1>{}, kslot ^ 1, vnext); }
                    else if (!(xf & 1)) { kv_issue_one(cur, t + 1, kslot ^ 1, vnext, IC<2 * (i >> 2)>{}); kv_issue_one(cur, t + 1, kslot ^ 1, vnext, IC<2 * (i >> 2) + 1>{}); }
                }
                if constexpr (i == 13) pv_pre(IC<0>{});
                if constexpr (i == 15) exp_flush(IC<1>{});
            });
            if (xf & 1024) AW_STAMP();  // 2: A done
            // B: O1 += V(t-1)^T P1^T beside group 0's row maximum and its first exponentials
            pv_run(IC<1>{}, IC<0>{}, IC<13>{}, IC<6>{}, [&](auto ic) {
                constexpr int i = decltype(ic)::value;
                if constexpr (i == 1) scores_ready(IC<0>{});
                if constexpr (i >= 2 && i < 6) max_step(IC<0>{}, IC<i - 2>{}, kmask);
                if constexpr (i == 7) max_decide(IC<0>{}, first_tile);
                if constexpr (i >= 8 && i % 3 != 1) exp_pair(IC<0>{}, IC<2 * ((i - 8) / 3) + (i - 8) % 3>{});  // pairs 0..5
                if constexpr (i == 13) qk_pre();
            });
            // S1 = K(t) Q1^T beside the rest of group 0's exponentials
            qk_run(IC<1>{}, IC<13>{}, IC<12>{}, [&](auto ic) {
                constexpr int i = decltype(ic)::value;
                if constexpr (i % 3 != 2 && i < 15) exp_pair(IC<0>{}, IC<6 + 2 * (i / 3) + i % 3>{});  // pairs 6..15
                if constexpr (i == 13) pv_pre(IC<1>{});
                if constexpr (i == 15) exp_flush(IC<0>{});
            });
            if (xf & 1024) AW_STAMP();  // 3: B done
            // C: O0 += V(t)^T P0^T beside group 1's row maximum and its first exponentials
            pv_run(IC<0>{}, IC<1>{}, IC<99>{}, IC<0>{}, [&](auto ic) {
                constexpr int i = decltype(ic)::value;
                if constexpr (i == 1) scores_ready(IC<1>{});
                if constexpr (i >= 2 && i < 6) max_step(IC<1>{}, IC<i - 2>{}, kmask);
                if constexpr (i == 7) max_decide(IC<1>{}, first_tile);
                if constexpr (i >= 8 && (i & 1) == 0) exp_pair(IC<1>{}, IC<(i - 8) / 2>{});  // pairs 0..3
            });
            if (xf & 1024) AW_STAMP();  // 4: C done
            // the requests issued in A have had the tile to land.  Waited for here, in front of the FSMN window's stores:
            // the wait counts every vector-memory operation, and a store takes far longer to retire than these loads
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (xf & 1024) AW_STAMP();  // 5: loads landed
            if (with_fsmn && window_owner(t) == cur.qbi)
                fsmn_window(cur, t, kWVOfs + vprev * kWTile, kWVOfs + vslot * kWTile, tbuf);
#pragma unroll
            for (int s = 0; s < 8; ++s) kaddr[s] ^= kWTile;
            ++gt;
            vs3 = vnext;
        };
        for (int t = 0; t + 1 < cur.n_kt; ++t) key_tile(t, std::false_type{});
        if (cur.n_kt > 0) key_tile(cur.n_kt - 1, std::true_type{});
        else {
            static_for<8>([&](auto jc) { next_issue(jc, gt & 1, vs3); });
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }

        AW_STAMP();  // key loop done
        // ---- drain group 1: the rest of its last tile's exponentials and its last O update --------------------------
        {
            const int vprev = vs3 == 0 ? 2 : vs3 - 1;  // the last tile's V
            pv_pre(IC<1>{});
            static_for<12>([&](auto ic) { exp_pair(IC<1>{}, IC<4 + decltype(ic)::value>{}); });
            exp_flush(IC<1>{});
            pv_run(IC<1>{}, IC<1>{}, IC<99>{}, IC<0>{}, [&](auto) {});
            // the FSMN window behind the last tile (its second tile does not exist: those rows are past the length)
            if (with_fsmn && window_owner(cur.n_kt) == cur.qbi)
                fsmn_window(cur, cur.n_kt, kWVOfs + vprev * kWTile, kWVOfs + vs3 * kWTile, tbuf);
            // rows past the last window hold zeros: the block that owns them stores them
            if (with_fsmn) {
                const int z0 = 64 * cur.n_kt + 32, w0 = cur.qb0 + 64 * wave;
                uint32_t* out = reinterpret_cast<uint32_t*>(f_base(cur)) + lane;
                for (int r = (z0 > w0 ? z0 : w0); r < w0 + 64 && r < p.Tk; ++r) out[static_cast<int64_t>(r) * 256] = 0u;
            }
            // the O tiles were written by MFMAs in assembly: let the last one retire before anything reads them
            asm volatile("s_nop 15\n\ts_nop 7" : "+a"(O[0][0]), "+a"(O[0][1]), "+a"(O[0][2]), "+a"(O[0][3]),
                         "+a"(O[1][0]), "+a"(O[1][1]), "+a"(O[1][2]), "+a"(O[1][3]));
        }

        AW_STAMP();  // drain done
        // ---- the next block's Q has landed in the stage: take it, then the stage is free for this block's O ----------
        float inv[2];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const float l_tot = l_run[g] + __shfl_xor(l_run[g], 32);
            inv[g] = l_tot > 0.0f ? 1.0f / l_tot : 0.0f;
        }
        if (has_next) read_q();
        // (the stage is wave-private: the wave's own LDS operations execute in order.  One O tile at a time leaves the
        // accumulation registers: the statements with a memory clobber keep the compiler from fetching all eight first)
        {
            const int ll = opaque_tid() & 63;
            const int e31 = ll & 31, eh = ll >> 5;
            char* const erow = stage + e31 * 256;
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    f32x16& oa = O[g][mt];
                    asm volatile("" : "+a"(oa) : : "memory");
                    const f32x16 o = oa;
#pragma unroll
                    for (int gp = 0; gp < 2; ++gp) {
                        const uint32_t lo0 = pack_elem2<F16>(o[8 * gp] * inv[g], o[8 * gp + 1] * inv[g]);
                        const uint32_t lo1 = pack_elem2<F16>(o[8 * gp + 2] * inv[g], o[8 * gp + 3] * inv[g]);
                        const uint32_t hi0 = pack_elem2<F16>(o[8 * gp + 4] * inv[g], o[8 * gp + 5] * inv[g]);
                        const uint32_t hi1 = pack_elem2<F16>(o[8 * gp + 6] * inv[g], o[8 * gp + 7] * inv[g]);
                        const auto s0 = __builtin_amdgcn_permlane32_swap(lo0, hi0, false, false);
                        const auto s1 = __builtin_amdgcn_permlane32_swap(lo1, hi1, false, false);
                        const int chunk = 4 * mt + 2 * gp + eh;  // 16-byte chunk of the 256-byte row: d = 8 chunk .. + 8
                        *reinterpret_cast<uint4*>(erow + g * 8192 + ((chunk ^ (e31 & 15)) << 4)) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                    }
                }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            uint16_t* const ob = o_base(cur);
            const int r0 = ll >> 4, cch = ll & 15;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int r = i * 4 + r0;
                const uint4 v = *reinterpret_cast<const uint4*>(stage + r * 256 + ((cch ^ (r & 15)) << 4));
                const int qr = cur.qb0 + wave * 64 + r;
                if (qr < p.Tq) *reinterpret_cast<uint4*>(ob + static_cast<int64_t>(qr) * p.ld_out + cch * 8) = v;
            }
        }
        AW_STAMP();  // block done
        if (!has_next) break;
        L = Ln;
        cur = nxt;
        tbuf ^= 1;
    }
}

int32_t prepare_attention_wide(asrfe_ctx* ctx) {
    const void* fns[2] = {reinterpret_cast<const void*>(attn_wide_kernel<true>), reinterpret_cast<const void*>(attn_wide_kernel<false>)};
    for (const void* f : fns)
        if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, kWLds) != hipSuccess)
            return set_err(ctx, ASRFE_ERR_HIP, "attention: cannot raise the dynamic LDS limit to 160 KiB");
    return ASRFE_OK;
}

// called by launch_attention (attention.hip) with checked parameters
int32_t launch_attention_wide(asrfe_ctx* ctx, const AttnParams& p, hipStream_t s) {
    AttnParams q = p;
    q.n_qtiles = (p.Tq + kWRows - 1) / kWRows;
    const int n_blocks = q.n_qtiles * 4 * p.B;
    const int cus = ctx->num_cus > 0 ? ctx->num_cus : 256;
    dim3 grid(n_blocks < cus ? n_blocks : cus);
    if (ctx->cfg.enc_compute_dtype == ASRFE_DT_F16) hipLaunchKernelGGL(attn_wide_kernel<true>, grid, dim3(kWThreads), kWLds, s, q);
    else hipLaunchKernelGGL(attn_wide_kernel<false>, grid, dim3(kWThreads), kWLds, s, q);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_err(ctx, ASRFE_ERR_HIP, std::string("attention launch: ") + hipGetErrorString(e));
    return ASRFE_OK;
}

}  // namespace asrfe
