// Masked multi-head self-attention (d_k = 128) and the FSMN memory block of the SAN-M encoder on gfx950.
//
//   One workgroup = 4 waves = 128 query rows of one (utterance, head).
//
//   FSMN: fsmn[t][c] = mask_t (v_m[t][c] + sum_j w[j][c] v_m[t + j - 5][c]) for the workgroup's 128 rows and
//   its head's 128 channels (v_m = v inside the utterance, else 0).  The 128 + 2 x 5 rows of V it needs are
//   brought into LDS by LDS-DMA; a thread owns a pair of channels and 32 consecutive rows and slides an 11-tap
//   window through registers, so a wave stores whole 256-byte rows.
//
//   Attention, flash style: each wave owns 32 query rows and walks the key axis in tiles of 64 with an online
//   softmax.  Scores are computed transposed, S^T = K Q^T (v_mfma_f32_32x32x16 with the K tile as the A operand
//   and Q as the B operand), so a lane holds 32 key scores of ONE query row: the row max and row sum are
//   register reductions plus one exchange with lane ^ 32, and the probabilities are already laid out as the B
//   operand of the second product O^T = V^T P^T, whose accumulator again has the query on the lane, so the
//   running rescale is a per-lane scalar.
//   K and V tiles ([64 keys][128 d], 256-byte rows) go global -> LDS by 16-byte LDS-DMA loads (lds_dma.h) with
//   the bank swizzle on the source address: K chunk c of row r sits at chunk c ^ (r & 15) (row reads,
//   ds_read_b128); V chunk c of row r sits at chunk c ^ (((r & 3) << 2) | ((r >> 2) & 3)) and is read
//   TRANSPOSED by ds_read_b64_tr_b16: a 16-lane group fetches a block of 4 keys x 16 channels and every lane
//   receives the 4 keys of one channel, which is the A operand of O^T = V^T P^T.  No transposed copy of V exists
//   in memory.  Keys at or past the utterance length get probability 0 (their V rows are finite, so 0 x V = 0).
#include "common.h"
#include "elem.h"
#include "encoder.h"
#include "lds_dma.h"

namespace asrfe {

constexpr int kAttnThreads = 256;
constexpr int kTileBytes = 64 * 256;       // 64 keys x 128 d, 16-bit
constexpr int kBufBytes = 2 * kTileBytes;  // K tile | V tile
constexpr int kFsmnHalo = 5;
constexpr int kFsmnTaps = 11;
constexpr int kFsmnTapBytes = kFsmnTaps * 128 * 4;  // the head's taps, f32 [11][128], behind the stage
constexpr int kFsmnTapOfs = 2 * kBufBytes + 4096;
constexpr int kFsmnStage = kBufBytes;      // byte offset of the FSMN stage: 9 DMA rounds of 16 rows (128 + 2 x 5 halo, rounded up)

typedef short v4s __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

template <bool F16>
__global__ __launch_bounds__(kAttnThreads, 2) void attn_kernel(const AttnParams p) {
    // two K/V buffers: tile kt + 1 is fetched while tile kt is multiplied.  The FSMN phase stages its 144 rows in
    // bytes [32 KiB, 68 KiB), i.e. over buffer 1 plus 4 KiB, so that K and V tile 0 fly into buffer 0 at the same time.
    __shared__ __attribute__((aligned(16))) char smem[2 * kBufBytes + 4096 + kFsmnTapBytes];
    saturate_fp16_conversions<F16>();
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    // Workgroups are dealt round-robin over the 8 XCDs (private L2s).  Remap the linear id so that each XCD
    // gets a contiguous range of (utterance, head, query tile) triples, query tile fastest: the workgroups that
    // share one head's K and V then run on the same XCD at about the same time and hit in its L2.
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, xq = nwg >> 3, xr = nwg & 7;
    const int logical = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
    const int qtile = logical % p.n_qtiles;
    const int head = (logical / p.n_qtiles) & 3;
    const int b = logical / (p.n_qtiles * 4);
    int len = p.lens[b];
    len = len < 0 ? 0 : (len > p.Tk ? p.Tk : len);
    const int qb = qtile * 128;          // first query row of the workgroup
    const int q0 = qb + wave * 32;       // first query row of the wave
    const int64_t row0 = static_cast<int64_t>(b) * p.Tq;   // first query / output row of the utterance
    const int64_t row0k = static_cast<int64_t>(b) * p.Tk;  // its first key / value row
    const int n_kt = (len + 63) >> 6;
    const uint32_t ldb = static_cast<uint32_t>(p.ld_kv) * 2;  // bytes per key / value row

    // ---- tile staging: thread covers LDS slot (row i * 16 + (tid >> 4), chunk tid & 15) of a tile, i = 0..3
    const uint32_t smem_base = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    const char* kbase = reinterpret_cast<const char*>(p.k + row0k * p.ld_kv + head * 128);
    const char* vbase = reinterpret_cast<const char*>(p.v + row0k * p.ld_kv + head * 128);
    const int st_row = tid >> 4;
    const uint32_t k_chunk = (tid & 15) ^ ((tid >> 4) & 15);
    const uint32_t v_chunk = (tid & 15) ^ ((((tid >> 4) & 3) << 2) | ((tid >> 6) & 3));
    auto kv_issue = [&](int kt, int buf, bool with_v) {
        const uint32_t dst = __builtin_amdgcn_readfirstlane(smem_base + buf * kBufBytes + wave * 1024);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int key = kt * 64 + i * 16 + st_row;
            key = key < p.Tk ? key : p.Tk - 1;
            const uint32_t ro = static_cast<uint32_t>(key) * ldb;
            lds_dma16(kbase, ro + k_chunk * 16, dst + i * 4096);
            if (with_v) lds_dma16(vbase, ro + v_chunk * 16, dst + kTileBytes + i * 4096);
        }
    };
    // tile kt uses buffer (n_kt - 1 - kt) & 1: the LAST tile always sits in buffer 0, which leaves buffer 1 and the
    // 4 KiB behind it free for the FSMN stage to arrive while that tile is computed
    const int buf_par = (n_kt - 1) & 1;
    if (n_kt > 0) kv_issue(0, buf_par, true);
    // Inside the key loop the same loads are issued one at a time between the MFMAs of the score product, from
    // running byte offsets (row i * 16 + st_row of the next tile; + 64 rows per tile) clamped to the last row:
    // no multiply and no burst of eight loads in front of the matrix work.
    const uint32_t tile_step = 64u * ldb;
    const uint32_t last_row = static_cast<uint32_t>(p.Tk - 1) * ldb;
    const uint32_t kclamp = last_row + k_chunk * 16, vclamp = last_row + v_chunk * 16;
    const uint32_t vdelta = (v_chunk - k_chunk) * 16;  // (wraps: added to a K offset it gives the V offset)
    uint32_t kro[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) kro[i] = static_cast<uint32_t>(64 + i * 16 + st_row) * ldb + k_chunk * 16;
    // load j of a tile: j >> 1 = row group i, j & 1 = V
    auto kv_issue_one = [&](auto jc, uint32_t dst_buf) {
        constexpr int J = decltype(jc)::value, I = J >> 1;
        if constexpr ((J & 1) == 0) {
            const uint32_t o = kro[I] < kclamp ? kro[I] : kclamp;
            lds_dma16(kbase, o, dst_buf + I * 4096);
        } else {
            const uint32_t ov = kro[I] + vdelta;
            const uint32_t o = ov < vclamp ? ov : vclamp;
            lds_dma16(vbase, o, dst_buf + kTileBytes + I * 4096);
            kro[I] += tile_step;
        }
    };

    // Q fragments: lane holds Q[q0 + l31][16 s + 8 h .. +8), s = 0..7
    u16x8 qf[8];
    {
        int qr = q0 + l31;
        qr = qr < p.Tq ? qr : p.Tq - 1;
        const uint16_t* qp = p.q + (row0 + qr) * p.ld_q + head * 128 + 8 * h;
#pragma unroll
        for (int s = 0; s < 8; ++s) qf[s] = *reinterpret_cast<const u16x8*>(qp + 16 * s);
    }

    // ---- FSMN memory block for rows qb .. qb + 127 of this head -----------------------------------------------
    // timing experiments (a build with -DASRFE_ATTN_TIMING; results are wrong when a bit is set):
    // 1 no FSMN phase, 2 no PV product, 4 no QK product, 8 no K / V prefetch, 16 no workgroup barrier per tile,
    // 32 no wait for the tile's loads, 64 no exponentials
#ifdef ASRFE_ATTN_TIMING
    const int xf = p.flags;
#else
    constexpr int xf = 0;
#endif
    // (runs after the attention proper: measured slightly faster than in front of it, where it delays the first
    // K / V tile of every workgroup at the same moment; its rows are fetched during the last key tile)
    auto fsmn_stage_issue = [&]() {
        // stage row rr (0..143) = V row qb - 5 + rr, clamped into the utterance's rows (masked when used)
        const uint32_t fst = __builtin_amdgcn_readfirstlane(smem_base + kFsmnStage + wave * 1024);
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            int t = qb - kFsmnHalo + i * 16 + st_row;
            t = t < 0 ? 0 : (t > p.Tk - 1 ? p.Tk - 1 : t);
            lds_dma16(vbase, static_cast<uint32_t>(t) * ldb + (tid & 15) * 16, fst + i * 4096);
        }
        // the taps of the head's 128 channels: 11 rows of 512 bytes = 352 16-byte pieces, piece q = (tap q / 32,
        // channels 4 (q % 32) .. +3), one per thread and a second one for the first 96 threads
        const char* wsrc = reinterpret_cast<const char*>(p.w_fsmn + head * 128);
        const uint32_t tdst = __builtin_amdgcn_readfirstlane(smem_base + kFsmnTapOfs + wave * 1024);
        lds_dma16(wsrc, static_cast<uint32_t>((tid >> 5) * 2048 + (tid & 31) * 16), tdst);
        if (tid < 96) lds_dma16(wsrc, static_cast<uint32_t>(((tid + 256) >> 5) * 2048 + (tid & 31) * 16), tdst + 4096);
    };
    auto fsmn_phase = [&]() {
        if (n_kt == 0) fsmn_stage_issue();  // (no key tile during which it could have been fetched)
        // taps of this thread's two channels
        const int cp = lane, rg = wave;  // channel pair, group of 32 rows (wave-uniform: the time masks are scalar)
        v2f wt[kFsmnTaps];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int j = 0; j < kFsmnTaps; ++j) {
            const float2 w = *reinterpret_cast<const float2*>(smem + kFsmnTapOfs + j * 512 + 8 * cp);
            wt[j] = v2f{w.x, w.y};
        }
        const char* fs = smem + kFsmnStage + (32 * rg) * 256 + 4 * cp;
        // all 42 stage rows of this thread's window walk are read up front (one LDS latency, not 42)
        uint32_t raw[32 + kFsmnTaps - 1];
#pragma unroll
        for (int r = 0; r < 32 + kFsmnTaps - 1; ++r) raw[r] = *reinterpret_cast<const uint32_t*>(fs + r * 256);
        // xw[r % 11] holds v_m of stage row r (time qb + 32 rg + r - 5); the loops are fully unrolled, so the
        // rotating index is a compile-time constant and the window never moves between registers
        // (both channels of the pair travel as one 2-vector: packed fp32 FMAs)
        v2f xw[kFsmnTaps];
        auto put = [&](int r) {
            const int tt = qb + 32 * rg + r - kFsmnHalo;
            float a, bq;
            unpack_elem2<F16>(raw[r], a, bq);
            const float ok = (tt >= 0 && tt < len) ? 1.0f : 0.0f;
            xw[r % kFsmnTaps] = v2f{a, bq} * ok;
        };
#pragma unroll
        for (int r = 0; r < kFsmnTaps - 1; ++r) put(r);
        uint32_t* out = reinterpret_cast<uint32_t*>(p.fsmn + (row0 + qb + 32 * rg) * 512 + head * 128 + 2 * cp);
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int t = qb + 32 * rg + i;
            put(i + kFsmnTaps - 1);
            v2f acc = xw[(i + kFsmnHalo) % kFsmnTaps];
#pragma unroll
            for (int j = 0; j < kFsmnTaps; ++j) acc = __builtin_elementwise_fma(wt[j], xw[(i + j) % kFsmnTaps], acc);
            if (t < p.Tk) out[static_cast<int64_t>(i) * 256] = t < len ? pack_elem2<F16>(acc.x, acc.y) : 0u;
        }
        __syncthreads();
    };

    // Use the Q fragments once here: the compiler then waits for their loads before the loop.  Left to their first
    // use inside the loop, its vmcnt waits there would also drain the K / V prefetch it does not know about.
#pragma unroll
    for (int s = 0; s < 8; ++s) asm volatile("" : "+v"(qf[s]));

    // transposed V reads: lane 4 q + pp of a 16-lane group addresses key row q of the 4-key block, channels
    // 4 pp .. 4 pp + 3 of the group's 16; groups 0, 1 (h = 0) and 2, 3 (h = 1) take keys 4 h .. 4 h + 3 of each 8.
    // Row r = 16 n + 8 hi + 4 h + q keeps chunk c at c ^ ((q << 2) | (2 hi + h)): the low bits are folded here.
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

#ifdef ASRFE_ATTN_TIMING
    // experiment (flag 512, with flag 1): wave 0 of workgroup 300 writes cycle-counter stamps into the unused fsmn buffer
    long long* trace = reinterpret_cast<long long*>(p.fsmn);
    int trace_n = 0;
    const bool tracing = (xf & 512) && bid == 300 && wave == 0;
#define ATTN_STAMP() do { if (tracing) { const long long tsc = __builtin_readcyclecounter(); if (lane == 0) trace[trace_n] = tsc; ++trace_n; } } while (0)
#else
#define ATTN_STAMP() do {} while (0)
#endif
    ATTN_STAMP();
    f32x16 oacc[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[mt][e] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f;
    const float c = p.scale_log2;

    // one key tile; MORE: another tile follows (its K / V loads are issued here), otherwise this is the last tile
    auto key_tile = [&](const int kt, auto more_c) __attribute__((always_inline)) {
        constexpr bool more = decltype(more_c)::value;
        const int cur = (kt & 1) ^ buf_par;
        // tile kt is complete in LDS (this thread's LDS-DMA loads are drained by hand: they are issued through
        // inline assembly, which keeps the compiler's LDS-read waits counted) and every wave is done with tile
        // kt - 1, whose buffer is refilled below
        if (!(xf & 32)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!(xf & 16)) __syncthreads();
        ATTN_STAMP();  // 0: past the barrier
        if (!more && !(xf & 1) && p.fsmn) fsmn_stage_issue();  // the last tile is in buffer 0: the stage's bytes are free
        ATTN_STAMP();  // 1: prefetch issued
        const char* ks = smem + ((xf & 8) ? 0 : cur) * kBufBytes;
        const char* vs = ks + kTileBytes;
        const uint32_t nxt = __builtin_amdgcn_readfirstlane(smem_base + (cur ^ 1) * kBufBytes + wave * 1024);

        // ---- S^T = K Q^T : two 32-key sub-tiles ---------------------------------------------
        // (consecutive MFMAs alternate between the two accumulators: none waits for the one right before it; the
        // K fragments are read three k-steps ahead of their use, and one load of the next tile is issued per k-step)
        f32x16 sacc[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int e = 0; e < 16; ++e) sacc[kk][e] = 0.0f;
        if (!(xf & 4)) {
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
                if (more && !(xf & 8)) kv_issue_one(sc, nxt);
                __builtin_amdgcn_sched_barrier(0);
            });
        } else if (more && !(xf & 8)) {
            static_for<8>([&](auto sc) { kv_issue_one(sc, nxt); });
        }
        ATTN_STAMP();  // 2: QK issued
        if constexpr (!more) {  // mask keys at or past the utterance length
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int kidx = kt * 64 + kk * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    if (kidx >= len) sacc[kk][e] = -INFINITY;
                }
        }
        float rmax = sacc[0][0];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int e = 0; e < 16; ++e) rmax = fmaxf(rmax, sacc[kk][e]);
        {   // the other 32 keys of the row sit in lane ^ 32: one register swap instead of a trip through the LDS
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(rmax), __float_as_uint(rmax), false, false);
            rmax = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        }
        // The running maximum only has to bound the scores well enough for exp2 to stay in range: it is raised
        // (and the accumulators rescaled) only when the tile's maximum exceeds it by more than 2^8 after scaling,
        // so most tiles skip the rescale.  Every tile holds at least one valid key, so rmax is finite.
        ATTN_STAMP();  // 3: row maximum known
        const bool grew = rmax * c > m_run * c + 8.0f;  // true on the first tile (m_run = -inf)
        const float m_new = grew ? rmax : m_run;
        const float alpha = grew ? __builtin_amdgcn_exp2f((m_run - m_new) * c) : 1.0f;
        const float mc = m_new * c;
        m_run = m_new;
        if (__any(grew)) {  // no lane raised its maximum: alpha is exactly 1 everywhere, skip the rescale
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int e = 0; e < 16; ++e) oacc[mt][e] *= alpha;
        }
        // ---- P = exp2(...) and O^T += V^T P^T, one 32-key sub-tile at a time: the MFMAs of sub-tile 0 have the
        //      exponentials of sub-tile 1 to overlap with (one straight-line block for the scheduler)
        // (scores travel as register pairs: one packed FMA and one packed add per two keys)
        v2f psum2 = {0.0f, 0.0f};
        const v2f c2 = {c, c}, nmc2 = {-mc, -mc};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            u16x8 pf[2];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    const v2f sv = {sacc[kk][8 * s2 + j], sacc[kk][8 * s2 + j + 1]};
                    const v2f t = __builtin_elementwise_fma(sv, c2, nmc2);
                    const v2f pv = (xf & 64) ? t : v2f{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
                    psum2 += pv;
                    const uint32_t w = pack_elem2<F16>(pv.x, pv.y);
                    pf[s2][j] = static_cast<uint16_t>(w & 0xffffu);
                    pf[s2][j + 1] = static_cast<uint16_t>(w >> 16);
                }
            if (!(xf & 2)) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        // element j of the fragment is key 32 kk + 16 s2 + 8 (j >> 2) + 4 h + (j & 3)
                        const char* blk = vs + (kk * 32 + s2 * 16) * 256;
                        const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (__attribute__((address_space(3))) v4s*)(blk + vtr[mt][0]));
                        const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (__attribute__((address_space(3))) v4s*)(blk + vtr[mt][1]));
                        const u16x8 vf = __builtin_bit_cast(u16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                        oacc[mt] = mfma32<F16>(vf, pf[s2], oacc[mt]);
                    }
            }
        }
        l_run = l_run * alpha + (psum2.x + psum2.y);
        ATTN_STAMP();  // 4: exponentials and PV issued
    };
    for (int kt = 0; kt + 1 < n_kt; ++kt) key_tile(kt, std::true_type{});
    if (n_kt > 0) key_tile(n_kt - 1, std::false_type{});
    ATTN_STAMP();

    // ---- normalise; each wave routes its 32 x 128 tile through LDS so that 16 lanes store one full 256-byte row ----
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = l_tot > 0.0f ? 1.0f / l_tot : 0.0f;
    __syncthreads();  // all waves are done with the K/V buffers
    char* es = smem + wave * 8192;
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
            const int chunk = 4 * mt + 2 * gp + h;  // 16-byte chunk of the 256-byte row: d = 8 chunk .. +8
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
    if (!(xf & 1) && p.fsmn) fsmn_phase();
}

int32_t launch_attention(asrfe_ctx* ctx, const AttnParams& p, hipStream_t s) {
    if (p.B <= 0 || p.Tq <= 0 || p.Tk <= 0) return ASRFE_OK;
    if (!p.q || !p.k || !p.v || !p.lens || !p.out || (p.fsmn && (!p.w_fsmn || p.Tq != p.Tk)))
        return set_err(ctx, ASRFE_ERR_BAD_ARG, "attention: null argument (or an FSMN output with Tq != Tk)");
    if (p.ld_q % 8 != 0 || p.ld_kv % 8 != 0 || p.ld_out % 8 != 0)
        return set_err(ctx, ASRFE_ERR_BAD_ARG, "attention: row strides must be multiples of 8 elements");
    // (the kernel's running 32-bit load offsets run up to two tiles past the last row before they are clamped)
    if ((static_cast<int64_t>(p.Tk) + 128) * p.ld_kv * 2 >= (int64_t{1} << 32))
        return set_err(ctx, ASRFE_ERR_SHAPE, "attention: one utterance's key / value rows must stay below 4 GiB");
    AttnParams q = p;
    q.n_qtiles = (p.Tq + 127) / 128;
    dim3 grid(q.n_qtiles * 4 * p.B);
    if (ctx->cfg.enc_compute_dtype == ASRFE_DT_F16) hipLaunchKernelGGL(attn_kernel<true>, grid, dim3(kAttnThreads), 0, s, q);
    else hipLaunchKernelGGL(attn_kernel<false>, grid, dim3(kAttnThreads), 0, s, q);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_err(ctx, ASRFE_ERR_HIP, std::string("attention launch: ") + hipGetErrorString(e));
    return ASRFE_OK;
}

}  // namespace asrfe
