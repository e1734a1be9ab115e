// HiFi-GAN generator kernels on gfx950: every convolution (plain, dilated, transposed) as an implicit GEMM on the
// matrix cores, plus the memory-bound pieces around them (input layout change, stage average, output convolution).
#include "common.h"
#include "elem.h"
#include "lds_dma.h"
#include "vocoder.h"

namespace asrfe {

namespace {
template <bool F16>
__device__ __forceinline__ float from_elem(uint16_t u) {
    if constexpr (F16) return static_cast<float>(__builtin_bit_cast(_Float16, u));
    else return __uint_as_float(static_cast<uint32_t>(u) << 16);
}
__device__ __forceinline__ float lrelu(float v, float slope) { return v > 0.0f ? v : v * slope; }
// leaky ReLU on eight packed fp16 values (0 < slope < 1: max(v, slope v)); the product is rounded to fp16, as the
// activated copies are that the producers store otherwise
template <bool F16>
__device__ __forceinline__ uint4 lrelu_packed(uint4 v, float slope) {
    if constexpr (F16) {
        typedef _Float16 h8 __attribute__((ext_vector_type(8)));
        const h8 x = __builtin_bit_cast(h8, v);
        const _Float16 s = static_cast<_Float16>(slope);
        const h8 y = __builtin_elementwise_max(x, x * s);
        return __builtin_bit_cast(uint4, y);
    } else {
        return v;  // (not used with bf16 operands: the launcher refuses act_in there)
    }
}
}  // namespace

// Convolution tile kernel.  A workgroup (4 waves) owns NC * 32 output channels x 4 * NP * 32 positions of one batch item
// and one phase; wave w owns positions [w * NP * 32, (w + 1) * NP * 32) as NC x NP accumulator tiles of 32 x 32.
//
//   Activations: the input rows the tile's taps touch (positions + (taps - 1) * |dtap| rows, at most 128 channels at a
//   time) are staged once in LDS, zero outside [0, Tin); rows are padded by 16 bytes so that the 32 rows of an operand
//   fragment fall in distinct banks.  Every tap reads its fragments from there at a row offset: no activation byte is
//   fetched from memory twice by a workgroup, whatever the kernel width and dilation.
//   Weights: streamed through an LDS ring in groups of GS k-steps (16 GS consecutive k of every output channel of the
//   tile; 16-byte chunks swizzled by row so that a fragment read covers all banks) by LDS-DMA loads, three groups
//   ahead; one counted wait + barrier per group.  Every workgroup of a launch reads the same weight rows: fetched once
//   per workgroup and far ahead, they arrive from L2 without the latency of a hot line showing up in the loop.
//   Accumulators start from the residual (loaded while the activations are staged): the epilogue adds the bias and stores.
//   In fp16 contexts the residual stream is itself kept in fp16 (ConvParams::res16): it is read as the accumulators'
//   start, stored rounded, and doubles as the next convolution's operand, which applies the leaky ReLU while staging
//   (ConvParams::act_in: packed fp16 max(v, slope v) on the 16-byte chunks in flight), so a convolution writes one
//   16-bit array instead of an f32 one and an activated 16-bit copy.
//   Per k-step (16 input channels of one tap) a wave issues NC * NP MFMAs on NC weight and NP activation fragments.
// MFMA operand layout: lane (l31, h) holds 8 consecutive k = 8 consecutive input channels of one tap, of position l31
// (A) and of weight row l31 (B); accumulator register 4 g + e is position 8 g + 4 h + e of output channel l31.
constexpr int kConvRingSlots = 4;
template <int NC>
constexpr int conv_group_steps() { return NC == 4 ? 2 : 4; }  // a ring slot is NC * 32 rows x 32 GS bytes: 8 KiB at most

template <bool F16, int NC, int NP, int OCC>
__global__ __launch_bounds__(256, OCC) void conv_tile_kernel(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) char xs[];
    saturate_fp16_conversions<F16>();
    constexpr int P = 4 * NP * 32;
    constexpr int NSLOT = kConvRingSlots;
    constexpr int GS = conv_group_steps<NC>();
    constexpr int kRowBytes = 32 * GS;              // one output channel's k of a group
    constexpr int kSlotBytes = NC * 32 * kRowBytes;
    constexpr int kCpr = 2 * GS;                    // 16-byte chunks per ring row
    constexpr int kRowsPerInstr = 64 / kCpr;        // rows one LDS-DMA instruction (1 KiB) covers
    constexpr int kInstrPerWave = NC * 32 / kRowsPerInstr / 4;
    constexpr int kSwzShift = GS == 4 ? 1 : 2;      // rows sharing a 256-byte bank line: 2 (128-byte rows) or 4
    constexpr int NSTG = OCC == 1 ? 16 : 10;       // 16-byte activation chunks a thread keeps in flight while staging
    static_assert(kInstrPerWave >= 1, "ring geometry");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int b = blockIdx.z / p.phases, ph = blockIdx.z - b * p.phases;
    const int co0 = blockIdx.y * (NC * 32);
    const int q0 = blockIdx.x * P;
    const int span = (p.taps - 1) * (p.dtap < 0 ? -p.dtap : p.dtap);
    const int lo = p.off0 + (p.dtap < 0 ? (p.taps - 1) * p.dtap : 0);  // smallest input offset of any tap
    const int R = P + span;
    const int CiT = p.ci_tile, nblk = CiT >> 4, cpr = CiT >> 3;  // channels per pass, k-steps per tap, 16-byte chunks per row
    const int stride = 2 * CiT + 16;
    const int K = p.taps * p.Ci;
    const int S = p.taps * nblk;
    const int G = (S + GS - 1) / GS;
    const int zero_at = R * stride;
    const int ring_at = (zero_at + 16 + 127) & ~127;
    const uint32_t ring_lds = __builtin_amdgcn_readfirstlane(lds_addr_of(xs) + ring_at);

    // timing experiments (a build with -DASRFE_VOC_TIMING and ASRFE_VOC_ABLATE in the environment; results are wrong when
    // a bit below 16 is set): 1 no stores, 2 no k-steps, 4 no activation staging, 8 no residual load, 32 phase stamps
#ifdef ASRFE_VOC_TIMING
    const int tflags = p.flags;
#else
    constexpr int tflags = 0;
#endif
#ifdef ASRFE_VOC_TIMING
    // wave 0 of one workgroup in the middle of the grid prints cycle stamps of its phases
    const bool tracing = (tflags & 32) && blockIdx.x == gridDim.x / 2 && blockIdx.y == 0 && blockIdx.z == gridDim.z / 2 && wave == 0;
    long long stamps[12];
    int n_st = 0;
#define VOC_STAMP() do { if (tracing) stamps[n_st] = __builtin_readcyclecounter(); ++n_st; } while (0)
#else
#define VOC_STAMP() do {} while (0)
#endif
    VOC_STAMP();  // 0: start
    // ---- accumulators start from the residual: requested first, needed at the first MFMA.  The activation
    // fragment is the MFMA's A operand and the weight fragment its B operand, so a lane owns ONE output channel
    // (co0 + nc * 32 + l31) and register 4 g + e of a tile is position 8 g + 4 h + e: 32 lanes read or write the 32
    // consecutive channels of a position, whole 128-byte lines of the f32 arrays.
    // Output rows are addressed through buffer descriptors that end behind the item's last valid row (for this phase):
    // an access to a row before the first or past the last is dropped (a load gives 0) by the hardware's range check,
    // which looks at the per-lane offset; that offset is row * row bytes + channel bytes, one add per access.
    f32x16 acc[NC][NP];
    const int n_rows = max(0, min(p.Tout, p.q_count * p.out_mul + p.out_add0 + ph * p.out_add_step));
    const int64_t item = static_cast<int64_t>(b) * p.Tout * p.Co;
    const uint32_t row4 = static_cast<uint32_t>(p.Co) * 4u;  // bytes per f32 output row
    // byte offset (f32 arrays) of register 0 of position tile 0 of this lane: row (first position + 4 h) * out_mul + ...
    const uint32_t lane_off4 = static_cast<uint32_t>((q0 + wave * (NP * 32) + 4 * h) * p.out_mul + p.out_add0 + ph * p.out_add_step) * row4 +
                               static_cast<uint32_t>(co0 + l31) * 4u;
    const uint32_t step4 = static_cast<uint32_t>(p.out_mul) * row4;  // one position further
    auto off4 = [&](int nc, int np, int r) { return lane_off4 + static_cast<uint32_t>(np * 32 + 8 * (r >> 2) + (r & 3)) * step4 + nc * 128; };
    auto rsrc_of = [&](const void* ptr, int elem_bytes) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(static_cast<const char*>(ptr)) + item * elem_bytes, 0,
                                                 n_rows * p.Co * elem_bytes, 0x00020000);
    };
    {   // accumulators start from the residual (the bias is added when they are stored: nothing here depends on a
        // load, so all of them are in flight together and are waited for at the first MFMA)
        const bool with_res = p.res_in && !(tflags & 8);
        if (p.res16 && with_res) {
            const __amdgpu_buffer_rsrc_t res_rsrc = rsrc_of(p.res_in, 2);
#pragma unroll
            for (int nc = 0; nc < NC; ++nc)
#pragma unroll
                for (int np = 0; np < NP; ++np)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        acc[nc][np][r] = from_elem<F16>(static_cast<uint16_t>(__builtin_amdgcn_raw_buffer_load_b16(res_rsrc, off4(nc, np, r) >> 1, 0, 0)));
        } else {
            const __amdgpu_buffer_rsrc_t res_rsrc = rsrc_of(p.res_in ? static_cast<const void*>(p.res_in) : p.bias, 4);
#pragma unroll
            for (int nc = 0; nc < NC; ++nc)
#pragma unroll
                for (int np = 0; np < NP; ++np)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        acc[nc][np][r] = with_res ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(res_rsrc, off4(nc, np, r), 0, 0)) : 0.0f;
        }
    }

    // weight staging: instruction w + 4 i of a group (i < kInstrPerWave) moves ring rows kRowsPerInstr (w + 4 i) ...;
    // this lane's row of instruction w and the source chunk that belongs in its slot
    const uint16_t* wtile = p.w + (static_cast<int64_t>(ph) * p.Co + co0) * K;
    const int srow = kRowsPerInstr * wave + lane / kCpr;             // + 4 kRowsPerInstr i
    const int schunk = (lane & (kCpr - 1)) ^ ((srow >> kSwzShift) & (kCpr - 1));
    const int wsw = (l31 >> kSwzShift) & (kCpr - 1);                 // swizzle of the fragment rows nc * 32 + l31

    const uint16_t* xitem = p.x + static_cast<int64_t>(b) * p.Tin * p.Ci;
    for (int ci0 = 0; ci0 < p.Ci; ci0 += CiT) {
        if (ci0) {  // every wave is done with the previous pass's rows and ring, and no load into the ring is pending
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        int gtap = 0, gcb = 0;  // first k-step of the next group to stage
        auto issue = [&](int slot) {
            const int kofs = gtap * p.Ci + ci0 + gcb * 16;
            const int ce = min(kofs + 8 * schunk, K - 8);  // (a group may reach past the last k: those steps meet zeros)
            const uint32_t dst = __builtin_amdgcn_readfirstlane(ring_lds + slot * kSlotBytes + wave * 1024);
#pragma unroll
            for (int i = 0; i < kInstrPerWave; ++i)
                lds_dma16(wtile, static_cast<uint32_t>((srow + 4 * kRowsPerInstr * i) * K + ce) * 2u, dst + i * 4096);
            gcb += GS;
            while (gcb >= nblk) { gcb -= nblk; ++gtap; }
            gtap = min(gtap, p.taps);  // (groups past the end keep re-reading the last valid chunk)
        };
#pragma unroll
        for (int i = 0; i < NSLOT - 1; ++i) issue(i);

        // ---- stage rows [q0 + lo, q0 + lo + R) x channels [ci0, ci0 + CiT)
        const int n_chunks = R * cpr;
        for (int i0 = tid; i0 < ((tflags & 4) ? 0 : n_chunks); i0 += NSTG * 256) {
            uint4 v[NSTG];
            int dst[NSTG];
#pragma unroll
            for (int u = 0; u < NSTG; ++u) {
                const int i = i0 + u * 256;
                const int r = i / cpr, c = i - r * cpr;
                const int ti = q0 + lo + r;
                dst[u] = i < n_chunks ? r * stride + c * 16 : -1;
                v[u] = make_uint4(0, 0, 0, 0);
                if (i < n_chunks && ti >= 0 && ti < p.Tin)
                    v[u] = *reinterpret_cast<const uint4*>(xitem + static_cast<int64_t>(ti) * p.Ci + ci0 + 8 * c);
            }
            if (p.act_in) {
#pragma unroll
                for (int u = 0; u < NSTG; ++u) v[u] = lrelu_packed<F16>(v[u], p.slope);
            }
#pragma unroll
            for (int u = 0; u < NSTG; ++u)
                if (dst[u] >= 0) *reinterpret_cast<uint4*>(xs + dst[u]) = v[u];
        }
        if (tid == 0) *reinterpret_cast<uint4*>(xs + zero_at) = make_uint4(0, 0, 0, 0);
        VOC_STAMP();  // 1 (+ 3 per pass): staging issued and written
        __syncthreads();
        VOC_STAMP();  // 2: staged everywhere

        // ---- k-steps s = tap * nblk + cb in groups of GS; a step past the last one multiplies by the zero chunk
        const int xbase = (wave * (NP * 32) + l31 + p.off0 - lo) * stride + h * 16;
        int tap = 0, cb = 0;    // the k-step the next activation read belongs to
        auto read_x = [&](u16x8 (&dst)[NP]) {
            const bool real = tap < p.taps;
            const int at = real ? xbase + tap * p.dtap * stride + cb * 32 : zero_at;
            const int step = real ? 32 * stride : 0;
#pragma unroll
            for (int np = 0; np < NP; ++np) dst[np] = *reinterpret_cast<const u16x8*>(xs + at + np * step);
            if (++cb == nblk) { cb = 0; ++tap; }
        };
        auto read_w = [&](u16x8 (&dst)[NC], int slot, int j) {
            const char* wp = xs + ring_at + slot * kSlotBytes + l31 * kRowBytes + (((2 * j + h) ^ wsw) << 4);
#pragma unroll
            for (int nc = 0; nc < NC; ++nc) dst[nc] = *reinterpret_cast<const u16x8*>(wp + nc * 32 * kRowBytes);
        };
        u16x8 xf[2][NP], wf[2][NC];
        read_x(xf[0]);
        int slot = 0;
        for (int g = 0; g < ((tflags & 2) ? 0 : G); ++g) {
            // this wave's part of group g has landed (the younger groups may still fly), then all waves meet: group g
            // is complete and nobody reads group g - 1 any more, whose slot takes group g + NSLOT - 1
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NSLOT - 2) * kInstrPerWave) : "memory");
            issue(slot == 0 ? NSLOT - 1 : slot - 1);
            read_w(wf[0], slot, 0);
#pragma unroll
            for (int j = 0; j < GS; ++j) {
                read_x(xf[(j + 1) & 1]);
                if (j + 1 < GS) read_w(wf[(j + 1) & 1], slot, j + 1);
                __builtin_amdgcn_sched_barrier(0);  // the next step's fragment reads stay ahead of this step's MFMAs
#pragma unroll
                for (int nc = 0; nc < NC; ++nc)
#pragma unroll
                    for (int np = 0; np < NP; ++np) acc[nc][np] = mfma32<F16>(xf[j & 1][np], wf[j & 1][nc], acc[nc][np]);
            }
            slot = slot + 1 == NSLOT ? 0 : slot + 1;
        }
        VOC_STAMP();  // 3: k-steps issued
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the ring's last (unused) groups

    // ---- epilogue: bias, stores (32 lanes = the 32 consecutive channels of one position)
    if (!((tflags & 1) && acc[0][0][0] != 12345.0f)) {
        const __amdgpu_buffer_rsrc_t ro_rsrc = rsrc_of(p.res_out ? static_cast<const void*>(p.res_out) : p.bias, p.res16 ? 2 : 4);
        const __amdgpu_buffer_rsrc_t y_rsrc = rsrc_of(p.y16 ? static_cast<const void*>(p.y16) : p.bias, 2);
        float bv[NC];
#pragma unroll
        for (int nc = 0; nc < NC; ++nc) bv[nc] = p.bias[co0 + nc * 32 + l31];
#pragma unroll
        for (int np = 0; np < NP; ++np)
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int nc = 0; nc < NC; ++nc) {
                    const uint32_t o = off4(nc, np, r);
                    const float v = acc[nc][np][r] + bv[nc];
                    if (p.res_out) {
                        if (p.res16) __builtin_amdgcn_raw_buffer_store_b16(to_elem<F16>(v), ro_rsrc, o >> 1, 0, 0);
                        else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), ro_rsrc, o, 0, 0);
                    }
                    if (p.y16) __builtin_amdgcn_raw_buffer_store_b16(to_elem<F16>(lrelu(v, p.slope)), y_rsrc, o >> 1, 0, 0);
                }
    }
#ifdef ASRFE_VOC_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    VOC_STAMP();
    if (tracing && lane == 0) {
        printf("conv Ci %d Co %d taps %d T %d NC %d NP %d OCC %d:", p.Ci, p.Co, p.taps, p.q_count, NC, NP, OCC);
        for (int i = 1; i < n_st && i < 12; ++i) printf(" %lld", stamps[i] - stamps[i - 1]);
        printf("\n");
    }
#endif
}

// Residual pair of a block with few channels (32 or 64), fused: out = x + conv2(lrelu(conv1(lrelu(x)) + b1)) + b2 with
// conv1 dilated and conv2 plain, both C -> C over k taps.  Below 128 channels a convolution is bound by its activation
// traffic; run as two launches a pair moves 16 bytes per element (16-bit copy in, 16-bit intermediate out and in again,
// f32 residual in, f32 and 16-bit results out), fused it moves 8: the f32 stream is read once (converted to the 16-bit
// operand type, leaky ReLU applied, while it is staged) and written once; the intermediate never leaves the LDS.
//   (With the residual stream in fp16, PairParams::res16, the pair moves 4 bytes per element: 16-byte chunks of x go
//   straight to the LDS through the same packed leaky ReLU as the two-launch form's staging, the accumulators start from
//   the 16-bit x and the result is stored rounded; the bits are again those of the two launches.)
//   A workgroup (4 waves) owns M = 4 * NP * 32 positions of conv1's output, of which the inner P = M - (k - 1) are the
//   positions of conv2's output it stores (conv2 needs (k - 1) / 2 rows of conv1 on either side; neighbouring tiles
//   recompute those).  A1 = lrelu(x) rows [p0 - h2 - h1, ... + M + (k - 1) d), A2 = lrelu(conv1 + b1) rows
//   [p0 - h2, ... + M) and zero outside the signal, both 16-bit in LDS with padded rows; A2 takes A1's place once conv1
//   has read it.  The weights of both convolutions stream through one LDS ring as one sequence of groups, so conv2's
//   first groups arrive while conv1 ends.  A wave owns NC x NP accumulator tiles per convolution.  64 channels: 2 x 2
//   (a k-step reads 4 fragments for 4 MFMAs; with 2 x 1 it reads 3 for 2 and the LDS, not the matrix pipe, sets the pace),
//   two workgroups per CU.  32 channels: 1 x 2 and four workgroups per CU: there the pair is bound by its memory traffic
//   and more workgroups in different phases hide more of it (measured against 1 x 4 with two per CU).
//   The arithmetic (operand rounding, accumulation order) is that of the two separate launches: results are identical.
struct PairParams {
    const void* x;        // [B][T][C] residual stream in: f32, or the 16-bit operand type with res16
    void* out;            // [B][T][C] residual stream out (a different buffer: tiles read their neighbours' rows of x)
    int res16;
    const uint16_t* w1;   // [C][k * C]
    const float* b1;
    const uint16_t* w2;
    const float* b2;
    int T, k, dil;
    float slope;
};

template <bool F16, int NC, int NP, int OCC>
__global__ __launch_bounds__(256, OCC) void conv_pair_kernel(const PairParams p) {
    extern __shared__ __attribute__((aligned(16))) char xs[];
    saturate_fp16_conversions<F16>();
    constexpr int C = NC * 32, M = 4 * NP * 32;
    constexpr int NSLOT = kConvRingSlots, GS = 4;
    constexpr int kRowBytes = 32 * GS, kSlotBytes = C * kRowBytes, kCpr = 2 * GS, kRowsPerInstr = 64 / kCpr;
    constexpr int kInstrPerWave = C / kRowsPerInstr / 4;
    constexpr int nblk = C / 16, cpr = C / 8, stride = 2 * C + 16;
    constexpr int NSTG = 6;  // 32-byte pieces of the f32 stream a thread keeps in flight while staging
    static_assert(kInstrPerWave >= 1, "ring geometry");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int b = blockIdx.z;
    const int h2 = (p.k - 1) / 2, h1 = h2 * p.dil;
    const int P = M - 2 * h2;
    const int p0 = blockIdx.x * P;
    const int R1 = M + 2 * h1, R2 = M + 2 * h2;
    const int K = p.k * C, S = p.k * nblk, G = (S + GS - 1) / GS;
    const int a2_at = 0, zero_at = max(R1, R2) * stride;  // A2 takes A1's rows once conv1 is done with them
    const int ring_at = (zero_at + 16 + 127) & ~127;
    const uint32_t ring_lds = __builtin_amdgcn_readfirstlane(lds_addr_of(xs) + ring_at);

    // ---- conv2's accumulators start from the residual x (rows at or past min(T, p0 + P) read as 0 and are never stored:
    // the buffer descriptors end there).  Requested first: they are the oldest loads when the first group is waited for.
    f32x16 acc1[NC][NP], acc2[NC][NP];
    const int n_rows = max(0, min(p.T, p0 + P));
    const int64_t item = static_cast<int64_t>(b) * p.T * C;
    constexpr uint32_t row4 = C * 4u;
    const uint32_t lane_off4 = static_cast<uint32_t>(p0 + wave * (NP * 32) + 4 * h) * row4 + static_cast<uint32_t>(l31) * 4u;
    auto off4 = [&](int nc, int np, int r) { return lane_off4 + static_cast<uint32_t>(np * 32 + 8 * (r >> 2) + (r & 3)) * row4 + nc * 128; };
    if (p.res16) {
        const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint16_t*>(static_cast<const uint16_t*>(p.x)) + item, 0, n_rows * C * 2, 0x00020000);
#pragma unroll
        for (int nc = 0; nc < NC; ++nc)
#pragma unroll
            for (int np = 0; np < NP; ++np)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    acc2[nc][np][r] = from_elem<F16>(static_cast<uint16_t>(__builtin_amdgcn_raw_buffer_load_b16(x_rsrc, off4(nc, np, r) >> 1, 0, 0)));
                    acc1[nc][np][r] = 0.0f;
                }
    } else {
        const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(static_cast<const float*>(p.x)) + item, 0, n_rows * C * 4, 0x00020000);
#pragma unroll
        for (int nc = 0; nc < NC; ++nc)
#pragma unroll
            for (int np = 0; np < NP; ++np)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    acc2[nc][np][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, off4(nc, np, r), 0, 0));
                    acc1[nc][np][r] = 0.0f;
                }
    }

    // ---- weight ring: groups 0 .. G - 1 are conv1's, G .. 2 G - 1 conv2's
    const int srow = kRowsPerInstr * wave + lane / kCpr;
    const int schunk = (lane & (kCpr - 1)) ^ ((srow >> 1) & (kCpr - 1));
    const int wsw = (l31 >> 1) & (kCpr - 1);
    int gi = 0;
    auto issue = [&](int slot) {
        const bool second = gi >= G;
        const uint16_t* wt = second ? p.w2 : p.w1;
        const int kofs = (second ? gi - G : gi) * (16 * GS);
        const int ce = min(kofs + 8 * schunk, K - 8);  // (the last group may reach past the last k: those steps meet zeros)
        const uint32_t dst = __builtin_amdgcn_readfirstlane(ring_lds + slot * kSlotBytes + wave * 1024);
#pragma unroll
        for (int i = 0; i < kInstrPerWave; ++i)
            lds_dma16(wt, static_cast<uint32_t>((srow + 4 * kRowsPerInstr * i) * K + ce) * 2u, dst + i * 4096);
        gi = min(gi + 1, 2 * G - 1);  // (issues past the end re-read the last group)
    };
#pragma unroll
    for (int i = 0; i < NSLOT - 1; ++i) issue(i);

    // ---- stage A1 = lrelu(x) rows [p0 - h2 - h1, + R1) as 16-bit, zero outside [0, T)
    if (p.res16) {
        // x is already in the operand type: 16-byte chunks straight through, the activation on the packed values (the
        // arithmetic of the two-launch form's staging, so that both forms give the same bits)
        const uint16_t* xitem = static_cast<const uint16_t*>(p.x) + item;
        const int first = p0 - h2 - h1;
        const int n_chunks = R1 * cpr;
        constexpr int NST16 = 10;
        for (int i0 = tid; i0 < n_chunks; i0 += NST16 * 256) {
            uint4 v[NST16];
            int dst[NST16];
#pragma unroll
            for (int u = 0; u < NST16; ++u) {
                const int i = i0 + u * 256;
                const int r = i / cpr, c = i - r * cpr;
                const int ti = first + r;
                dst[u] = i < n_chunks ? r * stride + c * 16 : -1;
                v[u] = make_uint4(0, 0, 0, 0);
                if (i < n_chunks && ti >= 0 && ti < p.T)
                    v[u] = *reinterpret_cast<const uint4*>(xitem + static_cast<int64_t>(ti) * C + 8 * c);
            }
#pragma unroll
            for (int u = 0; u < NST16; ++u)
                if (dst[u] >= 0) *reinterpret_cast<uint4*>(xs + dst[u]) = lrelu_packed<F16>(v[u], p.slope);
        }
        if (tid == 0) *reinterpret_cast<uint4*>(xs + zero_at) = make_uint4(0, 0, 0, 0);
    } else {
        const float* xitem = static_cast<const float*>(p.x) + item;
        const int first = p0 - h2 - h1;
        const int n_chunks = R1 * cpr;
        for (int i0 = tid; i0 < n_chunks; i0 += NSTG * 256) {
            float4 va[NSTG], vb[NSTG];
            int dst[NSTG];
#pragma unroll
            for (int u = 0; u < NSTG; ++u) {
                const int i = i0 + u * 256;
                const int r = i / cpr, c = i - r * cpr;
                const int ti = first + r;
                dst[u] = i < n_chunks ? r * stride + c * 16 : -1;
                va[u] = vb[u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (i < n_chunks && ti >= 0 && ti < p.T) {
                    const float4* src = reinterpret_cast<const float4*>(xitem + static_cast<int64_t>(ti) * C + 8 * c);
                    va[u] = src[0];
                    vb[u] = src[1];
                }
            }
#pragma unroll
            for (int u = 0; u < NSTG; ++u)
                if (dst[u] >= 0)
                    *reinterpret_cast<uint4*>(xs + dst[u]) =
                        make_uint4(pack_elem2<F16>(lrelu(va[u].x, p.slope), lrelu(va[u].y, p.slope)),
                                   pack_elem2<F16>(lrelu(va[u].z, p.slope), lrelu(va[u].w, p.slope)),
                                   pack_elem2<F16>(lrelu(vb[u].x, p.slope), lrelu(vb[u].y, p.slope)),
                                   pack_elem2<F16>(lrelu(vb[u].z, p.slope), lrelu(vb[u].w, p.slope)));
        }
        if (tid == 0) *reinterpret_cast<uint4*>(xs + zero_at) = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();

    // ---- one convolution's k-steps: activations from rows (row of the lane's position) + tap * dtap of the region at
    // `base`, weights from the ring; a step past the last one multiplies by the zero chunk
    int slot = 0;
    auto run = [&](int base, int dtap, f32x16 (&acc)[NC][NP]) {
        const int xbase = base + (wave * (NP * 32) + l31) * stride + h * 16;
        int tap = 0, cb = 0;
        auto read_x = [&](u16x8 (&dst)[NP]) {
            const bool real = tap < p.k;
            const int at = real ? xbase + tap * dtap * stride + cb * 32 : zero_at;
            const int step = real ? 32 * stride : 0;
#pragma unroll
            for (int np = 0; np < NP; ++np) dst[np] = *reinterpret_cast<const u16x8*>(xs + at + np * step);
            if (++cb == nblk) { cb = 0; ++tap; }
        };
        auto read_w = [&](u16x8 (&dst)[NC], int sl, int j) {
            const char* wp = xs + ring_at + sl * kSlotBytes + l31 * kRowBytes + (((2 * j + h) ^ wsw) << 4);
#pragma unroll
            for (int nc = 0; nc < NC; ++nc) dst[nc] = *reinterpret_cast<const u16x8*>(wp + nc * 32 * kRowBytes);
        };
        u16x8 xf[2][NP], wf[2][NC];
        read_x(xf[0]);
        for (int g = 0; g < G; ++g) {
            // this wave's part of the group has landed (the younger groups may still fly), then all waves meet: the group
            // is complete and nobody reads the one before it any more, whose slot takes the group NSLOT - 1 ahead
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NSLOT - 2) * kInstrPerWave) : "memory");
            issue(slot == 0 ? NSLOT - 1 : slot - 1);
            read_w(wf[0], slot, 0);
#pragma unroll
            for (int j = 0; j < GS; ++j) {
                read_x(xf[(j + 1) & 1]);
                if (j + 1 < GS) read_w(wf[(j + 1) & 1], slot, j + 1);
                __builtin_amdgcn_sched_barrier(0);  // the next step's fragment reads stay ahead of this step's MFMAs
#pragma unroll
                for (int nc = 0; nc < NC; ++nc)
#pragma unroll
                    for (int np = 0; np < NP; ++np) acc[nc][np] = mfma32<F16>(xf[j & 1][np], wf[j & 1][nc], acc[nc][np]);
            }
            slot = slot + 1 == NSLOT ? 0 : slot + 1;
        }
    };

    run(0, p.dil, acc1);

    // ---- A2 = lrelu(conv1 + b1) as 16-bit, zero where the position lies outside the signal (conv2's padding); its rows
    // past M (read by the taps of positions that are not stored) are cleared.  Every wave is done reading A1 first.
    __syncthreads();
    {
        for (int i = tid; i < 2 * h2 * cpr; i += 256) {
            const int r = M + i / cpr, c = i % cpr;
            *reinterpret_cast<uint4*>(xs + a2_at + r * stride + c * 16) = make_uint4(0, 0, 0, 0);
        }
        float bv[NC];
#pragma unroll
        for (int nc = 0; nc < NC; ++nc) bv[nc] = p.b1[nc * 32 + l31];
#pragma unroll
        for (int np = 0; np < NP; ++np)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ql = wave * (NP * 32) + np * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
                const int q = p0 - h2 + ql;
                const bool inside = q >= 0 && q < p.T;
#pragma unroll
                for (int nc = 0; nc < NC; ++nc) {
                    const uint16_t v = inside ? to_elem<F16>(lrelu(acc1[nc][np][r] + bv[nc], p.slope)) : uint16_t{0};
                    *reinterpret_cast<uint16_t*>(xs + a2_at + ql * stride + (nc * 32 + l31) * 2) = v;
                }
            }
    }
    __syncthreads();

    run(a2_at, 1, acc2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the ring's last (unused) groups

    // ---- epilogue: bias, rows out (32 lanes = the 32 consecutive channels of one position)
    {
        float bv[NC];
#pragma unroll
        for (int nc = 0; nc < NC; ++nc) bv[nc] = p.b2[nc * 32 + l31];
        if (p.res16) {
            const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc(static_cast<uint16_t*>(p.out) + item, 0, n_rows * C * 2, 0x00020000);
#pragma unroll
            for (int np = 0; np < NP; ++np)
#pragma unroll
                for (int r = 0; r < 16; ++r)
#pragma unroll
                    for (int nc = 0; nc < NC; ++nc)
                        __builtin_amdgcn_raw_buffer_store_b16(to_elem<F16>(acc2[nc][np][r] + bv[nc]), o_rsrc, off4(nc, np, r) >> 1, 0, 0);
        } else {
            const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc(static_cast<float*>(p.out) + item, 0, n_rows * C * 4, 0x00020000);
#pragma unroll
            for (int np = 0; np < NP; ++np)
#pragma unroll
                for (int r = 0; r < 16; ++r)
#pragma unroll
                    for (int nc = 0; nc < NC; ++nc)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, acc2[nc][np][r] + bv[nc]), o_rsrc, off4(nc, np, r), 0, 0);
        }
    }
}

template <bool F16>
__global__ __launch_bounds__(256) void mel_to_ntc_kernel(const float* __restrict__ mel, int n_mels, int T, uint16_t* __restrict__ out) {
    saturate_fp16_conversions<F16>();
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;  // over T * n_mels, channel fastest
    if (i >= T * n_mels) return;
    const int t = i / n_mels, c = i - t * n_mels;
    out[static_cast<int64_t>(b) * T * n_mels + i] = to_elem<F16>(mel[(static_cast<int64_t>(b) * n_mels + c) * T + t]);
}

struct ScaleActIn { const float* in[4]; };
template <bool F16>
__global__ __launch_bounds__(256) void scale_act_kernel(const ScaleActIn src, int n_in, float scale, float slope, int64_t n4,
                                                        uint16_t* __restrict__ y16) {
    saturate_fp16_conversions<F16>();
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n4) return;
    float4 v = reinterpret_cast<const float4*>(src.in[0])[i];
    for (int k = 1; k < n_in; ++k) {
        const float4 t = reinterpret_cast<const float4*>(src.in[k])[i];
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    reinterpret_cast<uint2*>(y16)[i] = make_uint2(pack_elem2<F16>(lrelu(v.x * scale, slope), lrelu(v.y * scale, slope)),
                                                  pack_elem2<F16>(lrelu(v.z * scale, slope), lrelu(v.w * scale, slope)));
}

// the same from 16-bit inputs (the residual stream in the operand type): eight values per thread
struct ScaleActIn16 { const uint16_t* in[4]; };
template <bool F16>
__global__ __launch_bounds__(256) void scale_act16_kernel(const ScaleActIn16 src, int n_in, float scale, float slope, int64_t n8,
                                                          uint16_t* __restrict__ y16) {
    saturate_fp16_conversions<F16>();
    const int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n8) return;
    float acc[8];
    {
        const u16x8 t = reinterpret_cast<const u16x8*>(src.in[0])[i];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = from_elem<F16>(t[e]);
    }
    for (int k = 1; k < n_in; ++k) {
        const u16x8 t = reinterpret_cast<const u16x8*>(src.in[k])[i];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += from_elem<F16>(t[e]);
    }
    u16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = to_elem<F16>(lrelu(acc[e] * scale, slope));
    reinterpret_cast<u16x8*>(y16)[i] = o;
}

template <bool F16>
__global__ __launch_bounds__(256) void conv_post_kernel(const uint16_t* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, int T, int C, float* __restrict__ wav) {
    const int b = blockIdx.y;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    float acc = bias[0];
    for (int tap = 0; tap < 7; ++tap) {
        const int ti = t + tap - 3;
        if (ti < 0 || ti >= T) continue;
        const u16x8* xr = reinterpret_cast<const u16x8*>(x + (static_cast<int64_t>(b) * T + ti) * C);
        const float* wr = w + tap * C;
        for (int c8 = 0; c8 < (C >> 3); ++c8) {
            const u16x8 v = xr[c8];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc = fmaf(wr[8 * c8 + e], from_elem<F16>(v[e]), acc);
        }
    }
    wav[static_cast<int64_t>(b) * T + t] = tanhf(acc);
}

#define VOC_CHECK_LAUNCH(who)                                                                                         \
    do {                                                                                                              \
        const hipError_t e__ = hipGetLastError();                                                                     \
        if (e__ != hipSuccess) return set_err(ctx, ASRFE_ERR_HIP, std::string(who " launch: ") + hipGetErrorString(e__)); \
    } while (0)

namespace {
constexpr int kConvLdsMax = 160 * 1024;

// Tile shapes: NC = output-channel tiles per workgroup (1, 2 or 4, by the channel count), NP = position tiles per wave,
// OCC = workgroups a CU holds at once.  With 128 channels and more the product is MFMA-bound and the largest tile wins
// (fewest operand bytes per MFMA); with fewer channels a convolution is bound by its activation traffic, and two
// workgroups per CU let one load and store while the other computes.
struct ConvShape { int np, occ; };
// activation rows + zero chunk (rounded up to 128 bytes) + weight ring
int64_t conv_lds_bytes(int rows, int stride, int nc) {
    return ((static_cast<int64_t>(rows) * stride + 16 + 127) & ~int64_t{127}) + kConvRingSlots * nc * 32 * 32 * (nc == 4 ? 2 : 4);
}
constexpr int kMaxShapes = 5;
const ConvShape kShapes[3][kMaxShapes] = {{{2, 2}, {4, 2}, {8, 2}, {1, 2}, {0, 0}},   // NC = 1 (small tiles: 3-4 workgroups per CU)
                                          {{4, 2}, {2, 2}, {1, 2}, {0, 0}, {0, 0}},   // NC = 2
                                          {{2, 2}, {3, 1}, {2, 1}, {1, 1}, {0, 0}}};  // NC = 4

template <bool F16>
const void* conv_fn(int nc, int np, int occ) {
#define CF(NCV, NPV, OCCV) if (nc == NCV && np == NPV && occ == OCCV) return reinterpret_cast<const void*>(conv_tile_kernel<F16, NCV, NPV, OCCV>)
    CF(4, 2, 2); CF(4, 1, 1); CF(4, 2, 1); CF(4, 3, 1);
    CF(2, 1, 2); CF(2, 2, 2); CF(2, 4, 2);
    CF(1, 1, 2); CF(1, 2, 2); CF(1, 4, 2); CF(1, 8, 2);
#undef CF
    return nullptr;
}
// channels per pass: all of them up to 128; beyond that 128 or 64 at a time (a group of k-steps must not straddle two
// taps unless the pass covers every channel).  Two workgroups per CU only fit with 64.
int conv_ci_tile(int ci, int nc, int occ) {
    if (nc == 4 && occ == 2) return ci <= 64 ? ci : (ci % 64 == 0 ? 64 : 0);
    return ci <= 128 ? ci : (ci % 128 == 0 ? 128 : (ci % 64 == 0 ? 64 : 0));
}
}  // namespace

// The dynamic-LDS limit of a kernel is a per-device attribute: called with the context's device current when the
// vocoder weights are loaded.
int32_t prepare_conv(asrfe_ctx* ctx) {
    for (int f16 = 0; f16 < 2; ++f16)
        for (int nci = 0; nci < 3; ++nci)
            for (int k = 0; k < kMaxShapes; ++k) {
                const int nc = 1 << nci, np = kShapes[nci][k].np, occ = kShapes[nci][k].occ;
                if (!np) continue;
                const void* fn = f16 ? conv_fn<true>(nc, np, occ) : conv_fn<false>(nc, np, occ);
                if (!fn || hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, kConvLdsMax / occ) != hipSuccess)
                    return set_err(ctx, ASRFE_ERR_HIP, "vocoder convolution: cannot raise the dynamic LDS limit");
            }
    const void* pair_fns[4] = {reinterpret_cast<const void*>(conv_pair_kernel<true, 2, 2, 2>), reinterpret_cast<const void*>(conv_pair_kernel<false, 2, 2, 2>),
                               reinterpret_cast<const void*>(conv_pair_kernel<true, 1, 2, 4>), reinterpret_cast<const void*>(conv_pair_kernel<false, 1, 2, 4>)};
    for (const void* fn : pair_fns)
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, kConvLdsMax / 2) != hipSuccess)
            return set_err(ctx, ASRFE_ERR_HIP, "vocoder pair: cannot raise the dynamic LDS limit");
    return ASRFE_OK;
}

int32_t launch_conv_mfma(asrfe_ctx* ctx, const ConvParams& pin, hipStream_t s) {
    ConvParams p = pin;
    if (p.B <= 0 || p.q_count <= 0) return ASRFE_OK;
    if (p.Ci % 16 != 0 || p.Co % 32 != 0 || p.taps < 1 || p.phases < 1 || !p.x || !p.w || !p.bias)
        return set_err(ctx, ASRFE_ERR_SHAPE, "vocoder convolution: input channels must be a multiple of 16, output channels of 32");
    if (static_cast<int64_t>(p.B) * p.phases > 65535) return set_err(ctx, ASRFE_ERR_SHAPE, "vocoder convolution: batch * stride too large");
    if (p.taps * p.Ci < 8) return set_err(ctx, ASRFE_ERR_SHAPE, "vocoder convolution: too few weights");
    if ((p.res16 || p.act_in) && ctx->cfg.enc_compute_dtype != ASRFE_DT_F16)
        return set_err(ctx, ASRFE_ERR_BAD_ARG, "vocoder convolution: the 16-bit residual stream goes with fp16 operands");
    // (output rows are addressed by 32-bit byte offsets inside an item, up to eight position tiles past its end)
    if ((static_cast<int64_t>(p.Tout) + 8192ll * p.out_mul) * p.Co * 4 >= (int64_t{1} << 31))
        return set_err(ctx, ASRFE_ERR_SHAPE, "vocoder convolution: one item's output must stay below 2 GiB");
    static const int ablate = [] { const char* e = getenv("ASRFE_VOC_ABLATE"); return e ? atoi(e) : 0; }();
    p.flags = ablate & (15 | 32);
    const int nci = p.Co % 128 == 0 ? 2 : (p.Co % 64 == 0 ? 1 : 0), nc = 1 << nci;
    const int span = (p.taps - 1) * (p.dtap < 0 ? -p.dtap : p.dtap);
    // positions per workgroup: the first shape of the list that fits in its share of the LDS and wastes under a tenth
    // of its positions, else the one wasting least
    int pick = -1, best = -1;
    double best_eff = -1.0;
    // (timing experiments: ASRFE_VOC_SHAPE_NC1 / _NC2 / _NC4 = index of the only shape of the list to consider)
    static const int forced[3] = {[] { const char* e = getenv("ASRFE_VOC_SHAPE_NC1"); return e ? atoi(e) : -1; }(),
                                  [] { const char* e = getenv("ASRFE_VOC_SHAPE_NC2"); return e ? atoi(e) : -1; }(),
                                  [] { const char* e = getenv("ASRFE_VOC_SHAPE_NC4"); return e ? atoi(e) : -1; }()};
    for (int pass = forced[nci] >= 0 ? 0 : 1; pass < 2 && pick < 0 && best < 0; ++pass)
    for (int k = 0; k < kMaxShapes; ++k) {
        const int cand = kShapes[nci][k].np, occ = kShapes[nci][k].occ, P = 128 * cand;
        if (!cand || ((ablate & 16) && nc == 4 && occ == 2) || (pass == 0 && forced[nci] != k)) continue;
        const int cit = conv_ci_tile(p.Ci, nc, occ);
        if (!cit || conv_lds_bytes(P + span, 2 * cit + 16, nc) > kConvLdsMax / occ) continue;
        const int tiles = (p.q_count + P - 1) / P;
        const double eff = static_cast<double>(p.q_count) / (static_cast<double>(tiles) * P);
        if (eff >= 0.9) { pick = k; break; }
        if (eff > best_eff) { best_eff = eff; best = k; }
    }
    if (pick < 0) pick = best;
    if (pick < 0) return set_err(ctx, ASRFE_ERR_SHAPE, "vocoder convolution: channel count or kernel width x dilation not supported");
    const int np = kShapes[nci][pick].np, occ = kShapes[nci][pick].occ, P = 128 * np;
    p.ci_tile = conv_ci_tile(p.Ci, nc, occ);
    const unsigned lds = static_cast<unsigned>(conv_lds_bytes(P + span, 2 * p.ci_tile + 16, nc));
    const dim3 grid((p.q_count + P - 1) / P, p.Co / (32 * nc), p.B * p.phases), block(256);
    const bool f16 = ctx->cfg.enc_compute_dtype == ASRFE_DT_F16;
    const void* fn = f16 ? conv_fn<true>(nc, np, occ) : conv_fn<false>(nc, np, occ);
    void* args[1] = {&p};
    if (hipLaunchKernel(fn, grid, block, args, lds, s) != hipSuccess) { /* reported below */ }
    VOC_CHECK_LAUNCH("vocoder convolution");
    return ASRFE_OK;
}

namespace {
template <int NC, int NP>
int64_t pair_lds_bytes(int k, int dil) {
    constexpr int C = NC * 32, M = 4 * NP * 32, stride = 2 * C + 16;
    const int h2 = (k - 1) / 2;
    const int64_t rows = M + 2 * h2 * std::max(dil, 1);  // A1; A2 (M + 2 h2 rows) takes its place
    return ((rows * stride + 16 + 127) & ~int64_t{127}) + kConvRingSlots * C * 128;
}
}  // namespace

bool conv_pair_supported(int C, int k, int dil) {
    if (k < 1 || k % 2 == 0 || dil < 1) return false;
    if (C == 64) return k <= 33 && pair_lds_bytes<2, 2>(k, dil) <= kConvLdsMax / 2;
    if (C == 32) return k <= 65 && pair_lds_bytes<1, 2>(k, dil) <= kConvLdsMax / 4;
    return false;
}

int32_t launch_conv_pair(asrfe_ctx* ctx, const void* x, void* out, int res16, const VocConv& c1, const VocConv& c2, int B, int T,
                         float slope, hipStream_t s) {
    if (B <= 0 || T <= 0) return ASRFE_OK;
    const int C = c1.Co;
    if (c1.Ci != C || c2.Ci != C || c2.Co != C || c2.k != c1.k || c2.dil != 1 || !conv_pair_supported(C, c1.k, c1.dil) || x == out)
        return set_err(ctx, ASRFE_ERR_SHAPE, "vocoder pair: not a residual pair this kernel is built for");
    if (static_cast<int64_t>(T + 1024) * C * 4 >= (int64_t{1} << 31) || B > 65535)
        return set_err(ctx, ASRFE_ERR_SHAPE, "vocoder pair: one item's rows must stay below 2 GiB, batch below 65536");
    if (res16 && ctx->cfg.enc_compute_dtype != ASRFE_DT_F16)
        return set_err(ctx, ASRFE_ERR_BAD_ARG, "vocoder pair: the 16-bit residual stream goes with fp16 operands");
    PairParams p{};
    p.res16 = res16;
    p.x = x; p.out = out; p.w1 = c1.w; p.b1 = c1.bias; p.w2 = c2.w; p.b2 = c2.bias; p.T = T; p.k = c1.k; p.dil = c1.dil; p.slope = slope;
    const bool f16 = ctx->cfg.enc_compute_dtype == ASRFE_DT_F16;
    const int M = 256, P = M - (c1.k - 1);
    const unsigned lds = static_cast<unsigned>(C == 64 ? pair_lds_bytes<2, 2>(c1.k, c1.dil) : pair_lds_bytes<1, 2>(c1.k, c1.dil));
    const dim3 grid((T + P - 1) / P, 1, B), block(256);
    if (C == 64) {
        if (f16) hipLaunchKernelGGL((conv_pair_kernel<true, 2, 2, 2>), grid, block, lds, s, p);
        else hipLaunchKernelGGL((conv_pair_kernel<false, 2, 2, 2>), grid, block, lds, s, p);
    } else {
        if (f16) hipLaunchKernelGGL((conv_pair_kernel<true, 1, 2, 4>), grid, block, lds, s, p);
        else hipLaunchKernelGGL((conv_pair_kernel<false, 1, 2, 4>), grid, block, lds, s, p);
    }
    VOC_CHECK_LAUNCH("vocoder pair");
    return ASRFE_OK;
}

int32_t launch_mel_to_ntc(asrfe_ctx* ctx, const float* mel, int B, int n_mels, int T, uint16_t* out, hipStream_t s) {
    if (B <= 0 || T <= 0) return ASRFE_OK;
    const dim3 grid((T * n_mels + 255) / 256, B), block(256);
    if (ctx->cfg.enc_compute_dtype == ASRFE_DT_F16) hipLaunchKernelGGL(mel_to_ntc_kernel<true>, grid, block, 0, s, mel, n_mels, T, out);
    else hipLaunchKernelGGL(mel_to_ntc_kernel<false>, grid, block, 0, s, mel, n_mels, T, out);
    VOC_CHECK_LAUNCH("mel layout");
    return ASRFE_OK;
}

int32_t launch_scale_act(asrfe_ctx* ctx, const void* const* in, int res16, int n_in, float scale, float slope, int64_t n, uint16_t* y16,
                         hipStream_t s) {
    if (n <= 0) return ASRFE_OK;
    if (n_in < 1 || n_in > 4) return set_err(ctx, ASRFE_ERR_BAD_ARG, "stage average: 1..4 inputs");
    if (res16) {
        ScaleActIn16 src16{};
        for (int k = 0; k < n_in; ++k) src16.in[k] = static_cast<const uint16_t*>(in[k]);
        const int64_t n8 = n / 8;  // (channel counts are multiples of 32)
        const dim3 grid16(static_cast<unsigned>((n8 + 255) / 256));
        if (ctx->cfg.enc_compute_dtype == ASRFE_DT_F16) hipLaunchKernelGGL(scale_act16_kernel<true>, grid16, dim3(256), 0, s, src16, n_in, scale, slope, n8, y16);
        else hipLaunchKernelGGL(scale_act16_kernel<false>, grid16, dim3(256), 0, s, src16, n_in, scale, slope, n8, y16);
        VOC_CHECK_LAUNCH("stage average");
        return ASRFE_OK;
    }
    ScaleActIn src{};
    for (int k = 0; k < n_in; ++k) src.in[k] = static_cast<const float*>(in[k]);
    const int64_t n4 = n / 4;  // (channel counts are multiples of 32)
    const dim3 grid(static_cast<unsigned>((n4 + 255) / 256)), block(256);
    if (ctx->cfg.enc_compute_dtype == ASRFE_DT_F16) hipLaunchKernelGGL(scale_act_kernel<true>, grid, block, 0, s, src, n_in, scale, slope, n4, y16);
    else hipLaunchKernelGGL(scale_act_kernel<false>, grid, block, 0, s, src, n_in, scale, slope, n4, y16);
    VOC_CHECK_LAUNCH("stage average");
    return ASRFE_OK;
}

int32_t launch_conv_post(asrfe_ctx* ctx, const uint16_t* x, const float* w, const float* bias, int B, int T, int C, float* wav,
                         hipStream_t s) {
    if (B <= 0 || T <= 0) return ASRFE_OK;
    const dim3 grid((T + 255) / 256, B), block(256);
    if (ctx->cfg.enc_compute_dtype == ASRFE_DT_F16) hipLaunchKernelGGL(conv_post_kernel<true>, grid, block, 0, s, x, w, bias, T, C, wav);
    else hipLaunchKernelGGL(conv_post_kernel<false>, grid, block, 0, s, x, w, bias, T, C, wav);
    VOC_CHECK_LAUNCH("output convolution");
    return ASRFE_OK;
}

}  // namespace asrfe
