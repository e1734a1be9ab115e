// The attention + FSMN kernel for long query axes: the arithmetic of attention.hip (same tiles, same fragment
// layouts, same online softmax, same FSMN phase) with eight waves per workgroup that take turns on the matrix pipe.
//
//   One workgroup = 8 waves = 256 query rows of one (utterance, head); one workgroup per CU, two waves per SIMD.
//
// In attention.hip a wave walks a key tile as  S^T = K Q^T (16 MFMAs)  ->  row maximum, exponentials (vector ALU)  ->
// O^T += V^T P^T (16 MFMAs), each step waiting for the one before it; two workgroups share a CU, but nothing keeps
// their waves out of step, and a wave spends as long issuing its eight LDS-DMA loads per tile as on the MFMAs beside
// them.  Here the two waves of a SIMD belong to ONE workgroup and are held half a tile apart by the workgroup barrier:
//
//     iteration    group A (waves 0..3)           group B (waves 4..7)
//       2 t        PV(t - 1), QK(t)   [matrix]    softmax(t - 1)     [vector]
//       2 t + 1    softmax(t)         [vector]    PV(t - 1), QK(t)   [matrix]
//
// so that in every iteration one wave of a SIMD feeds the matrix pipe while the other runs its exponentials.  A K / V
// tile serves 256 query rows instead of 128: four LDS-DMA loads per thread and tile instead of eight, and every key
// and value row crosses from the L2 into LDS half as often.
//
// Buffers: K tile t sits in K buffer t & 1 (read in iterations 2 t and 2 t + 1), V tile t in V buffer t % 3 (read in
// iterations 2 t + 2 and 2 t + 3).  Tile t + 1 is requested in iteration 2 t (K first, then V): its K buffer was last
// read in iteration 2 t - 1, its V buffer in iteration 2 t - 1 too (V tile t - 2).  At the end of every odd iteration
// a wave waits until at most its two youngest loads (the V part of the newest tile) are outstanding; the barrier that
// closes the iteration then publishes K(t + 1) and V(t) to everyone.  The FSMN stage (266 rows of V: 256 + 2 x 5 halo)
// has LDS of its own behind the tile buffers and is requested in place of the tile after the last one.
//
// Measured (64 x 4 heads x 500 x 500, cycle stamps of one wave per group, timing build): a key tile of 256 query rows
// takes 3.7 k cycles here against 5.1 k for two 128-row workgroups of attention.hip; per iteration the matrix step is
// 1.5-1.9 k cycles (32 MFMAs = 1.0 k) and the softmax step 1.4-1.7 k: the key loop is bound by the vector ALU (about
// 100 instructions and 32 quarter-rate exponentials per wave and tile), not by the matrix pipe.  The whole launch is
// NOT faster, 57-60 us against 54-56 us: with one workgroup per CU nothing overlaps a workgroup's prologue (Q and the
// first tile), its output pass and its FSMN phase with another workgroup's key loop, which two independent 128-row
// workgroups per CU do for free.  The kernel is therefore selected only by ASRFE_ATTN_PINGPONG=1 (read at
// asrfe_create); a test runs it against the same fp64 evaluation as the default kernel.
#include "common.h"
#include "elem.h"
#include "encoder.h"
#include "lds_dma.h"

namespace asrfe {

namespace {
constexpr int kPpThreads = 512;
constexpr int kPpRows = 256;
constexpr int kPpTileBytes = 64 * 256;                 // 64 keys x 128 d, 16-bit
constexpr int kPpKOfs = 0;                             // two K buffers
constexpr int kPpVOfs = 2 * kPpTileBytes;              // three V buffers
constexpr int kPpStageOfs = 5 * kPpTileBytes;          // FSMN stage: 9 DMA rounds of 32 rows (288 >= 256 + 10)
constexpr int kPpStageRounds = 9;
constexpr int kPpTapOfs = kPpStageOfs + kPpStageRounds * 8192;
constexpr int kPpTapBytes = kPpThreads * 16;           // 352 pieces of taps, padded to one piece per thread
constexpr int kPpLds = kPpTapOfs + kPpTapBytes;        // 163840 = all of a CU's LDS
constexpr int kPpHalo = 5;
constexpr int kPpTaps = 11;
static_assert(kPpLds <= 160 * 1024, "the ping-pong attention kernel needs the whole LDS of a CU");

typedef short v4s __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
}  // namespace

template <bool F16>
__global__ __launch_bounds__(kPpThreads, 1) void attn_pp_kernel(const AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    saturate_fp16_conversions<F16>();
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;  // 0: group A, 1: group B (one iteration behind)
    const int l31 = lane & 31, h = lane >> 5;
    // contiguous ranges of (utterance, head, query tile) triples per XCD, as in attention.hip
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, xq = nwg >> 3, xr = nwg & 7;
    const int logical = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
    const int qtile = logical % p.n_qtiles;
    const int head = (logical / p.n_qtiles) & 3;
    const int b = logical / (p.n_qtiles * 4);
    int len = p.lens[b];
    len = len < 0 ? 0 : (len > p.Tk ? p.Tk : len);
    const int qb = qtile * kPpRows;
    const int q0 = qb + wave * 32;
    const int64_t row0 = static_cast<int64_t>(b) * p.Tq;
    const int64_t row0k = static_cast<int64_t>(b) * p.Tk;
    const int n_kt = (len + 63) >> 6;
    const uint32_t ldb = static_cast<uint32_t>(p.ld_kv) * 2;

    // ---- tile staging: thread covers LDS slot (row i * 32 + (tid >> 4), chunk tid & 15) of a tile, i = 0, 1
    const uint32_t smem_base = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    const char* kbase = reinterpret_cast<const char*>(p.k + row0k * p.ld_kv + head * 128);
    const char* vbase = reinterpret_cast<const char*>(p.v + row0k * p.ld_kv + head * 128);
    const int st_row = tid >> 4;  // 0..31
    const uint32_t k_chunk = (tid & 15) ^ ((tid >> 4) & 15);
    const uint32_t v_chunk = (tid & 15) ^ ((((tid >> 4) & 3) << 2) | ((tid >> 6) & 3));
    // load j of tile kt: j >> 1 = K / V, j & 1 = row group
    auto tile_dma = [&](int kt, int j) {
        int key = kt * 64 + (j & 1) * 32 + st_row;
        key = key < p.Tk ? key : p.Tk - 1;
        const uint32_t ro = static_cast<uint32_t>(key) * ldb;
        if (j < 2) {
            const uint32_t dst = __builtin_amdgcn_readfirstlane(smem_base + kPpKOfs + (kt & 1) * kPpTileBytes + wave * 1024);
            lds_dma16(kbase, ro + k_chunk * 16, dst + (j & 1) * 8192);
        } else {
            const uint32_t dst = __builtin_amdgcn_readfirstlane(smem_base + kPpVOfs + (kt % 3) * kPpTileBytes + wave * 1024);
            lds_dma16(vbase, ro + v_chunk * 16, dst + (j & 1) * 8192);
        }
    };
    if (n_kt > 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) tile_dma(0, j);
    }

    // Q fragments: lane holds Q[q0 + l31][16 s + 8 h .. +8), s = 0..7
    u16x8 qf[8];
    {
        int qr = q0 + l31;
        qr = qr < p.Tq ? qr : p.Tq - 1;
        const uint16_t* qp = p.q + (row0 + qr) * p.ld_q + head * 128 + 8 * h;
#pragma unroll
        for (int s = 0; s < 8; ++s) qf[s] = *reinterpret_cast<const u16x8*>(qp + 16 * s);
    }

    // ---- FSMN memory block for rows qb .. qb + 255 of this head (see attention.hip) --------------------------------
    auto fsmn_stage_issue = [&]() {
        const uint32_t fst = __builtin_amdgcn_readfirstlane(smem_base + kPpStageOfs + wave * 1024);
#pragma unroll
        for (int i = 0; i < kPpStageRounds; ++i) {
            int t = qb - kPpHalo + i * 32 + st_row;
            t = t < 0 ? 0 : (t > p.Tk - 1 ? p.Tk - 1 : t);
            lds_dma16(vbase, static_cast<uint32_t>(t) * ldb + (tid & 15) * 16, fst + i * 8192);
        }
        // the taps of the head's 128 channels: 11 rows of 512 bytes = 352 16-byte pieces, piece q = (tap q / 32,
        // channels 4 (q % 32) .. +3); threads past 352 repeat the last piece into the padding of the tap area, so
        // that every wave issues the same number of loads
        const char* wsrc = reinterpret_cast<const char*>(p.w_fsmn + head * 128);
        const uint32_t tdst = __builtin_amdgcn_readfirstlane(smem_base + kPpTapOfs + wave * 1024);
        const int q = tid < 352 ? tid : 351;
        lds_dma16(wsrc, static_cast<uint32_t>((q >> 5) * 2048 + (q & 31) * 16), tdst);
    };
    auto fsmn_phase = [&]() {
        const int cp = lane, rg = wave;  // channel pair, group of 32 rows
        v2f wt[kPpTaps];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int j = 0; j < kPpTaps; ++j) {
            const float2 w = *reinterpret_cast<const float2*>(smem + kPpTapOfs + j * 512 + 8 * cp);
            wt[j] = v2f{w.x, w.y};
        }
        const char* fs = smem + kPpStageOfs + (32 * rg) * 256 + 4 * cp;
        uint32_t raw[32 + kPpTaps - 1];
#pragma unroll
        for (int r = 0; r < 32 + kPpTaps - 1; ++r) raw[r] = *reinterpret_cast<const uint32_t*>(fs + r * 256);
        v2f xw[kPpTaps];
        auto put = [&](int r) {
            const int tt = qb + 32 * rg + r - kPpHalo;
            float a, bq;
            unpack_elem2<F16>(raw[r], a, bq);
            const float ok = (tt >= 0 && tt < len) ? 1.0f : 0.0f;
            xw[r % kPpTaps] = v2f{a, bq} * ok;
        };
#pragma unroll
        for (int r = 0; r < kPpTaps - 1; ++r) put(r);
        uint32_t* out = reinterpret_cast<uint32_t*>(p.fsmn + (row0 + qb + 32 * rg) * 512 + head * 128 + 2 * cp);
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int t = qb + 32 * rg + i;
            put(i + kPpTaps - 1);
            v2f acc = xw[(i + kPpHalo) % kPpTaps];
#pragma unroll
            for (int j = 0; j < kPpTaps; ++j) acc = __builtin_elementwise_fma(wt[j], xw[(i + j) % kPpTaps], acc);
            if (t < p.Tk) out[static_cast<int64_t>(i) * 256] = t < len ? pack_elem2<F16>(acc.x, acc.y) : 0u;
        }
    };

    // (see attention.hip: the compiler waits for the Q loads here, not inside the loop)
#pragma unroll
    for (int s = 0; s < 8; ++s) asm volatile("" : "+v"(qf[s]));

    // transposed V reads (see attention.hip)
    int vtr[4][2];
    {
        const int q = (lane >> 2) & 3, pp = lane & 3, g1 = (lane >> 4) & 1;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int hi = 0; hi < 2; ++hi) {
                const int chunk = ((mt ^ q) << 2) | ((2 * g1 + (pp >> 1)) ^ (2 * hi + h));
                vtr[mt][hi] = (8 * hi + 4 * h + q) * 256 + chunk * 16 + 8 * (pp & 1);
            }
    }

    f32x16 oacc[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[mt][e] = 0.0f;
    f32x16 sacc[2];  // scores of the tile between its QK and its softmax
    u16x8 pf[2][2];  // probabilities of the tile between its softmax and its PV: [32-key half][16-key step]
    float m_run = -INFINITY, l_run = 0.0f;
    const float c = p.scale_log2;
#ifdef ASRFE_ATTN_TIMING
    // timing build, flag 512: wave 0 (group A) and wave 4 (group B) of workgroup 100 write cycle-counter stamps into
    // the fsmn buffer (the FSMN phase is skipped): A at [0..), B at [64..)
    long long* trace = reinterpret_cast<long long*>(p.fsmn) + (grp ? 64 : 0);
    int trace_n = 0;
    const bool tracing = (p.flags & 512) && bid == 100 && (wave & 3) == 0;
    const bool with_fsmn = p.fsmn != nullptr && !(p.flags & 512);
#define PP_STAMP() do { if (tracing) { const long long tsc = __builtin_readcyclecounter(); if (lane == 0) trace[trace_n] = tsc; ++trace_n; } } while (0)
#else
    const bool with_fsmn = p.fsmn != nullptr;
#define PP_STAMP() do {} while (0)
#endif

    // ---- S^T = K Q^T for key tile t; the four loads of the tile `next` (if >= 0) go out between its MFMAs
    auto qk = [&](int t, int next) {
        const char* ks = smem + kPpKOfs + (t & 1) * kPpTileBytes;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int e = 0; e < 16; ++e) sacc[kk][e] = 0.0f;
        const char* krow = ks + l31 * 256;
        const int ksw = l31 & 15;
        auto kread = [&](int s, int kk) {
            return *reinterpret_cast<const u16x8*>(krow + kk * 8192 + (((2 * s + h) ^ ksw) << 4));
        };
        u16x8 kf[3][2];
#pragma unroll
        for (int s = 0; s < 3; ++s) { kf[s][0] = kread(s, 0); kf[s][1] = kread(s, 1); }
        static_for<8>([&](auto sc) {
            constexpr int S = decltype(sc)::value;
            sacc[0] = mfma32<F16>(kf[S % 3][0], qf[S], sacc[0]);
            sacc[1] = mfma32<F16>(kf[S % 3][1], qf[S], sacc[1]);
            if constexpr (S + 3 < 8) { kf[S % 3][0] = kread(S + 3, 0); kf[S % 3][1] = kread(S + 3, 1); }
            if constexpr ((S & 1) == 0) {
                if (next >= 0) tile_dma(next, S >> 1);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    // ---- row maximum, running rescale, P = exp2(...) as the operand fragments of PV (see attention.hip)
    auto softmax = [&](int t, auto last_c) {
        if constexpr (decltype(last_c)::value) {  // the last tile: mask keys at or past the utterance length
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int kidx = t * 64 + kk * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    if (kidx >= len) sacc[kk][e] = -INFINITY;
                }
        }
        float rmax = sacc[0][0];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int e = 0; e < 16; ++e) rmax = fmaxf(rmax, sacc[kk][e]);
        {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(rmax), __float_as_uint(rmax), false, false);
            rmax = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
        const bool grew = rmax * c > m_run * c + 8.0f;
        const float m_new = grew ? rmax : m_run;
        const float alpha = grew ? __builtin_amdgcn_exp2f((m_run - m_new) * c) : 1.0f;
        const float mc = m_new * c;
        m_run = m_new;
        if (__any(grew)) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int e = 0; e < 16; ++e) oacc[mt][e] *= alpha;
        }
        v2f psum2 = {0.0f, 0.0f};
        const v2f c2 = {c, c}, nmc2 = {-mc, -mc};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    const v2f sv = {sacc[kk][8 * s2 + j], sacc[kk][8 * s2 + j + 1]};
                    const v2f tt = __builtin_elementwise_fma(sv, c2, nmc2);
                    const v2f pv = v2f{__builtin_amdgcn_exp2f(tt.x), __builtin_amdgcn_exp2f(tt.y)};
                    psum2 += pv;
                    const uint32_t w = pack_elem2<F16>(pv.x, pv.y);
                    pf[kk][s2][j] = static_cast<uint16_t>(w & 0xffffu);
                    pf[kk][s2][j + 1] = static_cast<uint16_t>(w >> 16);
                }
        l_run = l_run * alpha + (psum2.x + psum2.y);
        // the probabilities are complete HERE: left to the compiler, the exponentials sink past the workgroup barrier
        // into the PV step that first uses them, and the two groups' matrix and vector steps no longer alternate
        asm volatile("" : "+v"(pf[0][0]), "+v"(pf[0][1]), "+v"(pf[1][0]), "+v"(pf[1][1]), "+v"(l_run) : : "memory");
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- O^T += V^T P^T for key tile t.  The sixteen V fragments are read four MFMAs ahead of their use (a fragment
    //      is two transposed reads); the scheduling barriers keep the compiler from folding the reads back to their
    //      uses, where every MFMA would wait a whole LDS round trip.
    auto pv = [&](int t) {
        const char* vs = smem + kPpVOfs + (t % 3) * kPpTileBytes;
        auto vread = [&](auto ic) -> u16x8 {  // fragment I: 32-key half I >> 3, 16-key step (I >> 2) & 1, channel tile I & 3
            constexpr int I = decltype(ic)::value;
            const char* blk = vs + ((I >> 3) * 32 + ((I >> 2) & 1) * 16) * 256;
            const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(blk + vtr[I & 3][0]));
            const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(blk + vtr[I & 3][1]));
            return __builtin_bit_cast(u16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        };
        u16x8 vf[4];
        static_for<4>([&](auto ic) { vf[decltype(ic)::value] = vread(ic); });
        __builtin_amdgcn_sched_barrier(0);
        static_for<16>([&](auto ic) {
            constexpr int I = decltype(ic)::value;
            oacc[I & 3] = mfma32<F16>(vf[I & 3], pf[I >> 3][(I >> 2) & 1], oacc[I & 3]);
            if constexpr (I + 4 < 16) vf[I & 3] = vread(IC<I + 4>{});
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    // tile 0 has landed for everyone
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    bool stage_out = false;  // the FSMN stage has been requested (it rides in the queue behind the last tile)
    // End of odd iteration 2 t + 1: K(t + 1) and V(t) must have landed.  The loads behind them in this wave's queue may
    // stay in flight: the V part of tile t + 1 (2 loads) while tiles are still being requested, the FSMN stage (10
    // loads) after the last tile, nothing when there is no FSMN output.  Then the barrier publishes them.
    auto close_odd = [&](int t) {
        if (t + 1 < n_kt) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else if (stage_out) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    // In even iteration 2 t tile t + 1 is requested, or the FSMN stage after the last tile.
    auto open_even = [&](int t) {
        if (t + 1 == n_kt && with_fsmn) { fsmn_stage_issue(); stage_out = true; }
    };
    if (n_kt > 0) {
        // Every wave passes 2 n_kt + 2 barriers, whichever group it is in.
        if (grp == 0) {
            // group A: iteration 2 t = PV(t - 1), QK(t) with the tile's loads between the MFMAs; 2 t + 1 = softmax(t)
            auto iter = [&](int t, auto last_c) {
                constexpr bool LAST = decltype(last_c)::value;
                open_even(t);
                PP_STAMP();
                if (t > 0) pv(t - 1);
                PP_STAMP();
                qk(t, LAST ? -1 : t + 1);
                PP_STAMP();
                __syncthreads();
                PP_STAMP();
                softmax(t, last_c);
                PP_STAMP();
                close_odd(t);
            };
            for (int t = 0; t + 1 < n_kt; ++t) iter(t, std::false_type{});
            iter(n_kt - 1, std::true_type{});
            pv(n_kt - 1);      // iteration 2 n_kt
            __syncthreads();
            close_odd(n_kt);   // iteration 2 n_kt + 1: group B's last PV
        } else {
            // group B, one iteration behind: 2 t = the tile's loads, softmax(t - 1); 2 t + 1 = PV(t - 1), QK(t)
            auto iter = [&](int t, auto first_c, auto last_c) {
                constexpr bool FIRST = decltype(first_c)::value, LAST = decltype(last_c)::value;
                open_even(t);
                if constexpr (!LAST) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) tile_dma(t + 1, j);
                }
                PP_STAMP();
                if constexpr (!FIRST) softmax(t - 1, std::false_type{});
                PP_STAMP();
                __syncthreads();
                PP_STAMP();
                if constexpr (!FIRST) pv(t - 1);
                PP_STAMP();
                qk(t, -1);
                PP_STAMP();
                close_odd(t);
            };
            if (n_kt == 1) {
                iter(0, std::true_type{}, std::true_type{});
            } else {
                iter(0, std::true_type{}, std::false_type{});
                for (int t = 1; t + 1 < n_kt; ++t) iter(t, std::false_type{}, std::false_type{});
                iter(n_kt - 1, std::false_type{}, std::true_type{});
            }
            softmax(n_kt - 1, std::true_type{});  // iteration 2 n_kt
            __syncthreads();
            pv(n_kt - 1);                         // iteration 2 n_kt + 1
            close_odd(n_kt);
        }
    }

    // ---- normalise; each wave routes its 32 x 128 tile through LDS so that 16 lanes store one full 256-byte row ----
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = l_tot > 0.0f ? 1.0f / l_tot : 0.0f;
    char* es = smem + wave * 8192;  // (over the K / V buffers: every wave is past the loop's last barrier)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
            uint32_t lo0 = pack_elem2<F16>(oacc[mt][8 * gp] * inv, oacc[mt][8 * gp + 1] * inv);
            uint32_t lo1 = pack_elem2<F16>(oacc[mt][8 * gp + 2] * inv, oacc[mt][8 * gp + 3] * inv);
            uint32_t hi0 = pack_elem2<F16>(oacc[mt][8 * gp + 4] * inv, oacc[mt][8 * gp + 5] * inv);
            uint32_t hi1 = pack_elem2<F16>(oacc[mt][8 * gp + 6] * inv, oacc[mt][8 * gp + 7] * inv);
            auto s0 = __builtin_amdgcn_permlane32_swap(lo0, hi0, false, false);
            auto s1 = __builtin_amdgcn_permlane32_swap(lo1, hi1, false, false);
            const int chunk = 4 * mt + 2 * gp + h;
            *reinterpret_cast<uint4*>(es + l31 * 256 + ((chunk ^ (l31 & 15)) << 4)) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
        }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = i * 4 + (lane >> 4), cch = lane & 15;
        const uint4 v = *reinterpret_cast<const uint4*>(es + r * 256 + ((cch ^ (r & 15)) << 4));
        const int qr = q0 + r;
        if (qr < p.Tq) *reinterpret_cast<uint4*>(p.out + (row0 + qr) * p.ld_out + head * 128 + cch * 8) = v;
    }
    if (with_fsmn) {
        if (!stage_out) fsmn_stage_issue();  // (an utterance without keys: nothing requested it)
        fsmn_phase();
    }
}

int32_t prepare_attention_pp(asrfe_ctx* ctx) {
    const void* fns[2] = {reinterpret_cast<const void*>(attn_pp_kernel<true>), reinterpret_cast<const void*>(attn_pp_kernel<false>)};
    for (const void* f : fns)
        if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, kPpLds) != hipSuccess)
            return set_err(ctx, ASRFE_ERR_HIP, "attention: cannot raise the dynamic LDS limit to 160 KiB");
    return ASRFE_OK;
}

// called by launch_attention (attention.hip) with checked parameters
int32_t launch_attention_pp(asrfe_ctx* ctx, const AttnParams& p, hipStream_t s) {
    AttnParams q = p;
    q.n_qtiles = (p.Tq + kPpRows - 1) / kPpRows;
    dim3 grid(q.n_qtiles * 4 * p.B);
    if (ctx->cfg.enc_compute_dtype == ASRFE_DT_F16) hipLaunchKernelGGL(attn_pp_kernel<true>, grid, dim3(kPpThreads), kPpLds, s, q);
    else hipLaunchKernelGGL(attn_pp_kernel<false>, grid, dim3(kPpThreads), kPpLds, s, q);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_err(ctx, ASRFE_ERR_HIP, std::string("attention launch: ") + hipGetErrorString(e));
    return ASRFE_OK;
}

}  // namespace asrfe
