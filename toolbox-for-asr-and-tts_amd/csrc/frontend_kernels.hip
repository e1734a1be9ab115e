// Speech frontend kernels for gfx950: Kaldi fbank (framing, DC removal, pre-emphasis, window,
// 512-point real FFT, power, mel, log) and LFR + CMVN, either as separate steps or fused.
//
// Work decomposition of the fbank kernel
//   * 16 lanes own one frame, so a 64-lane wave carries 4 consecutive frames and never needs a
//     workgroup barrier: every hand-off goes through the wave's private LDS region and LDS
//     operations of one wave execute in program order.
//   * The 512-point real FFT is a 256-point complex FFT of z[n] = y[2n] + i y[2n+1], split
//     16 x 16: lane n2 transforms z[16 n1 + n2] over n1 in registers, multiplies by
//     W256^(n2 k1), the 16 lanes transpose through LDS, lane k1 transforms over n2 and ends
//     with Z[k1 + 16 k2].  One pass over LDS pairs Z[k] with Z[256-k] to get 4|X[k]|^2.
//   * mel: each lane sums the 5 filters m = lane + 16 g from the power row in LDS; the weight
//     table is zero-padded per group to a wave-uniform multiple of 4 terms and read 4 terms at a
//     time.  For the standard 80-filter layout the trip counts are compile-time constants, so
//     every LDS read of the stage is in flight before the first multiply waits for one.
#include <type_traits>

#include "common.h"

namespace asrfe {

constexpr int kFramesPerWave = 4;
constexpr int kGroupsPerWave = 8;
constexpr int kWavesPerBlock = 4;
constexpr int kFramesPerBlock = kFramesPerWave * kGroupsPerWave * kWavesPerBlock;  // 128
constexpr int kStageSamples = kFrameShift * (kFramesPerWave - 1) + kFrameLen;       // 880
constexpr int kStageChunks = kStageSamples / 8;                                     // 110
constexpr int kExRow = 36;                             // floats per exchange row (16 complex + 16 B pad)
// Floats per frame region: 16 exchange rows plus 32.  The LDS serves 8-byte accesses 32 lanes (two frames) at a
// time over 64 banks, so neighbouring frames must lie 32 banks apart; 4-byte accesses go 32 lanes at a time over
// 32 banks, so the power row, the only thing accessed that way, is shifted by 16 floats in every other frame.
constexpr int kFrameLds = 16 * kExRow + 32;            // 608
constexpr int kWaveLds = kFramesPerWave * kFrameLds;   // 2432 floats = 9728 B per wave
constexpr int kConstFloats = (13 + 16 + 8) * 16 * 2;   // window and twiddle pairs kept in LDS
static_assert(kWaveLds >= kStageSamples + 16, "staging area must fit the wave region");
static_assert(kFrameLds >= 2 * kNc, "Z row must fit the frame region");

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Sum over the 16 lanes of a DPP row (= one frame), result in every lane: lane exchanges stay in the VALU.
__device__ __forceinline__ float row16_sum(float v) {
    auto dpp_add = [](float x, auto ctrl) {
        const int y = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, false);
        return x + __builtin_bit_cast(float, y);
    };
    v = dpp_add(v, std::integral_constant<int, 0xB1>{});   // quad_perm [1,0,3,2]
    v = dpp_add(v, std::integral_constant<int, 0x4E>{});   // quad_perm [2,3,0,1]
    v = dpp_add(v, std::integral_constant<int, 0x141>{});  // row_half_mirror
    v = dpp_add(v, std::integral_constant<int, 0x140>{});  // row_mirror
    return v;
}

// 4-term chunks per mel group (filters lane + 16 g) of the standard layout: 80 filters, 20 Hz .. 8 kHz, 512-point FFT
// (the last group's windows are one chunk wider than its widest filter: room to start them on different banks)
__device__ __host__ constexpr int std_mel_chunks(int g) { return g == 0 ? 1 : g == 1 ? 1 : g == 2 ? 2 : g == 3 ? 3 : 5; }
__device__ __host__ constexpr int std_mel_first(int g) { return g == 0 ? 0 : g == 1 ? 1 : g == 2 ? 2 : g == 3 ? 4 : 7; }

// Complex numbers travel as packed fp32 pairs (re, im) through v_pk_add / v_pk_mul / v_pk_fma_f32.  The half
// selectors (op_sel, op_sel_hi) and sign bits (neg_lo, neg_hi) of those instructions do the swaps and negations of
// complex arithmetic for free, which the compiler does not exploit on its own: hence the inline assembly.
typedef float v2f __attribute__((ext_vector_type(2)));

// Compile-time twiddles sit in SGPR pairs (one scalar operand per VOP3P instruction is allowed).
struct ctw { v2f w; };

// z * w = (zr wr - zi wi, zr wi + zi wr): t = (-zi wi, zi wr), then r = (zr wr, zr wi) + t
__device__ __forceinline__ v2f cmul(const v2f z, const v2f w) {
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[1,0]" : "=v"(t) : "v"(z), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(z), "v"(w), "v"(t));
    return r;
}
__device__ __forceinline__ v2f cmul(const v2f z, const ctw c) {
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[1,0]" : "=v"(t) : "v"(z), "s"(c.w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(z), "s"(c.w), "v"(t));
    return r;
}
// (-i z) * w = (zi wr + zr wi, zi wi - zr wr): t = (zr wi, -zr wr), then r = (zi wr, zi wi) + t
__device__ __forceinline__ v2f cmul_neg_i(const v2f z, const v2f w) {
    v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,0] neg_hi:[1,0]" : "=v"(t) : "v"(z), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(z), "v"(w), "v"(t));
    return r;
}
// b + (-i) e = (br + ei, bi - er)  and  b - (-i) e = (br - ei, bi + er)
__device__ __forceinline__ v2f add_neg_i(const v2f b, const v2f e) {
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(b), "v"(e));
    return r;
}
__device__ __forceinline__ v2f sub_neg_i(const v2f b, const v2f e) {
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(b), "v"(e));
    return r;
}
// a + conj(b) and a - conj(b)
__device__ __forceinline__ v2f add_conj(const v2f a, const v2f b) {
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ v2f sub_conj(const v2f a, const v2f b) {
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// 4-point forward DFT in place, natural output order.
__device__ __forceinline__ void dft4(v2f& z0, v2f& z1, v2f& z2, v2f& z3) {
    const v2f a = z0 + z2, b = z0 - z2, c = z1 + z3, e = z1 - z3;
    z0 = a + c;
    z2 = a - c;
    z1 = add_neg_i(b, e);
    z3 = sub_neg_i(b, e);
}

// 16-point forward DFT in registers.  Input x[n] at index n; output X[k] at index 4 (k & 3) + (k >> 2).
__device__ __forceinline__ void dft16(v2f (&z)[16]) {
    constexpr float c1 = 0.92387953251128674f;  // cos(pi/8)
    constexpr float s1 = 0.38268343236508977f;  // sin(pi/8)
    constexpr float h = 0.70710678118654752f;
    v2f y[16];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        v2f t0 = z[b], t1 = z[4 + b], t2 = z[8 + b], t3 = z[12 + b];
        dft4(t0, t1, t2, t3);
        y[b] = t0; y[4 + b] = t1; y[8 + b] = t2; y[12 + b] = t3;
    }
    // y[4 c + b] *= W16^(b c)
    y[5] = cmul(y[5], ctw{{c1, -s1}});    // b1 c1 : W^1
    y[6] = cmul(y[6], ctw{{h, -h}});      // b2 c1 : W^2
    y[7] = cmul(y[7], ctw{{s1, -c1}});    // b3 c1 : W^3
    y[9] = cmul(y[9], ctw{{h, -h}});      // b1 c2 : W^2
    y[10] = cmul(y[10], ctw{{0.0f, -1.0f}});  // b2 c2 : W^4 = -i
    y[11] = cmul(y[11], ctw{{-h, -h}});   // b3 c2 : W^6
    y[13] = cmul(y[13], ctw{{s1, -c1}});  // b1 c3 : W^3
    y[14] = cmul(y[14], ctw{{-h, -h}});   // b2 c3 : W^6
    y[15] = cmul(y[15], ctw{{-c1, s1}});  // b3 c3 : W^9
#pragma unroll
    for (int c = 0; c < 4; ++c) dft4(y[4 * c], y[4 * c + 1], y[4 * c + 2], y[4 * c + 3]);
#pragma unroll
    for (int k = 0; k < 16; ++k) z[k] = y[k];
}

__device__ __forceinline__ constexpr int pos16(int k) { return 4 * (k & 3) + (k >> 2); }

template <typename PcmT>
__device__ __forceinline__ float pcm_to_kaldi(PcmT v);
template <>
__device__ __forceinline__ float pcm_to_kaldi<int16_t>(int16_t v) { return static_cast<float>(v); }
template <>
__device__ __forceinline__ float pcm_to_kaldi<float>(float v) { return v * 32768.0f; }
// A float32 session sample in [-1, 1] (ASRFE_PCM_F32_WIRE): the value of the int16 the reference hands to its pipeline,
// int16(clip(v, -1, 1) * 32767), truncated toward zero (voice_interface.py:1997-2010).
struct UnitF32 {
    float v;
};
__device__ __forceinline__ float unit_to_wire(float v) {
    return __builtin_truncf(__builtin_fminf(__builtin_fmaxf(v, -1.0f), 1.0f) * 32767.0f);
}
template <>
__device__ __forceinline__ float pcm_to_kaldi<UnitF32>(UnitF32 s) { return unit_to_wire(s.v); }

// Eight consecutive samples as they lie in memory (one or two 16-byte loads), and their Kaldi-scaled floats.
template <typename PcmT>
struct Raw8;
template <>
struct Raw8<int16_t> {
    int4 q;
    __device__ __forceinline__ void load(const int16_t* p) { q = *reinterpret_cast<const int4*>(p); }
    __device__ __forceinline__ void get(float (&out)[8]) const {
        const int w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            out[2 * j] = static_cast<float>(static_cast<int16_t>(w[j] & 0xffff));
            out[2 * j + 1] = static_cast<float>(w[j] >> 16);
        }
    }
};
template <>
struct Raw8<float> {
    float4 a, b;
    __device__ __forceinline__ void load(const float* p) {
        a = *reinterpret_cast<const float4*>(p);
        b = *reinterpret_cast<const float4*>(p + 4);
    }
    __device__ __forceinline__ void get(float (&out)[8]) const {
        out[0] = a.x * 32768.0f; out[1] = a.y * 32768.0f; out[2] = a.z * 32768.0f; out[3] = a.w * 32768.0f;
        out[4] = b.x * 32768.0f; out[5] = b.y * 32768.0f; out[6] = b.z * 32768.0f; out[7] = b.w * 32768.0f;
    }
};

template <>
struct Raw8<UnitF32> {
    float4 a, b;
    __device__ __forceinline__ void load(const UnitF32* p) {
        a = *reinterpret_cast<const float4*>(p);
        b = *reinterpret_cast<const float4*>(p + 4);
    }
    __device__ __forceinline__ void get(float (&out)[8]) const {
        out[0] = unit_to_wire(a.x); out[1] = unit_to_wire(a.y); out[2] = unit_to_wire(a.z); out[3] = unit_to_wire(a.w);
        out[4] = unit_to_wire(b.x); out[5] = unit_to_wire(b.y); out[6] = unit_to_wire(b.z); out[7] = unit_to_wire(b.w);
    }
};

__device__ __forceinline__ uint16_t f32_to_bf16(float v) {
    return __builtin_bit_cast(uint16_t, static_cast<__bf16>(v));
}

// Writes the log-mel values mel[g] (filter l + 16 g) of frame t into every LFR/CMVN feature slot that holds them.
// Row i, slot j holds frame clamp(n i + j - left, 0, T-1), left = (m-1)/2; rows >= rows_out are dropped.  No global
// load happens here, so a wave never waits on its earlier stores or on the samples fetched ahead.
template <bool OUT_BF16>
__device__ __forceinline__ void lfr_scatter(void* __restrict__ feats_b, int t, int T, int Tp, int rows_out, int l,
                                            int n_mels, int lfr_m, int lfr_n, const float2* cmvn,
                                            const float (&mel)[kMelGroups]) {
    const int left = (lfr_m - 1) / 2;
    const int dim = lfr_m * n_mels;
    const int a = t + left - (lfr_m - 1);            // first row: n i >= a
    const int i_lo = a <= 0 ? 0 : (a + lfr_n - 1) / lfr_n;
    int i_hi = (t == T - 1) ? Tp - 1 : (t + left) / lfr_n;  // the last frame also fills every slot past the end
    if (i_hi > rows_out - 1) i_hi = rows_out - 1;
    for (int i = i_lo; i <= i_hi; ++i) {
        const int je = t + left - lfr_n * i;         // slot that holds t exactly
        int j_lo = (t == 0) ? 0 : je;                // frame 0 also fills the left padding slots
        int j_hi = (t == T - 1) ? lfr_m - 1 : je;
        if (j_lo < 0) j_lo = 0;
        if (j_hi > lfr_m - 1) j_hi = lfr_m - 1;
        for (int j = j_lo; j <= j_hi; ++j) {
            const int col0 = j * n_mels + l;
            const int64_t o = static_cast<int64_t>(i) * dim + col0;
#pragma unroll
            for (int g = 0; g < kMelGroups; ++g) {
                if (l + 16 * g < n_mels) {
                    const float2 ss = cmvn[col0 + 16 * g];  // (shift, scale), in LDS
                    const float y = (mel[g] + ss.x) * ss.y;
                    if constexpr (OUT_BF16) reinterpret_cast<uint16_t*>(feats_b)[o + 16 * g] = f32_to_bf16(y);
                    else reinterpret_cast<float*>(feats_b)[o + 16 * g] = y;
                }
            }
        }
    }
}

template <typename PcmT, bool FUSED, bool OUT_BF16, bool STD_MEL>
__global__ __launch_bounds__(kWavesPerBlock * 64, 3) void fbank_kernel(const FbankParams p) {
    __shared__ __attribute__((aligned(16))) float lds[kWavesPerBlock * kWaveLds + kMelMaxTerms * 16 + kMelGroups * 16 +
                                                      kConstFloats + (FUSED ? 2 * kMaxFeatDim : 0)];
    float* mel_w = lds + kWavesPerBlock * kWaveLds;  // float4 [chunk][lane]
    int* mel_sb = reinterpret_cast<int*>(mel_w + kMelMaxTerms * 16);
    // per-lane constants as (x, y) pairs, [row][lane of the frame]: 13 window rows, 16 + 8 twiddle rows
    v2f* c_win = reinterpret_cast<v2f*>(mel_sb + kMelGroups * 16);
    v2f* c_twa = c_win + 13 * 16;
    v2f* c_twp = c_twa + 16 * 16;
    float2* cmvn = reinterpret_cast<float2*>(c_twp + 8 * 16);  // (shift, scale) per feature column (fused mode)

    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const int lane = tid & 63;
    const int f = lane >> 4;   // frame of the group
    const int l = lane & 15;   // lane of the frame
    const int b = blockIdx.y;

    for (int i = tid; i < kMelMaxTerms * 16; i += kWavesPerBlock * 64) mel_w[i] = p.tab.mel_w[i];
    for (int i = tid; i < kMelGroups * 16; i += kWavesPerBlock * 64) mel_sb[i] = p.tab.mel_sb[i];

    const int64_t len = p.pcm_lens[b];
    // A clip with more frames than the output has rows is processed as if it had been cut after frame t_max - 1
    // (fused mode: t_max = tp_max * lfr_n); the reported counts are those of the cut clip.
    const int64_t t_all = len < kFrameLen ? 0 : 1 + (len - kFrameLen) / kFrameShift;
    const int T = static_cast<int>(t_all < p.t_max ? t_all : p.t_max);
    const int Tp = (T + p.lfr_n - 1) / p.lfr_n;
    if (blockIdx.x == 0 && tid == 0) {
        if (p.n_frames) p.n_frames[b] = T;
        if (p.lfr_lens) p.lfr_lens[b] = Tp;
    }
    const PcmT* __restrict__ row = reinterpret_cast<const PcmT*>(p.pcm) + static_cast<int64_t>(b) * p.pcm_stride;
    const bool vec_ok = p.aligned16 != 0;

    for (int i = tid; i < 13 * 16; i += kWavesPerBlock * 64) {
        const int n1 = i >> 4, ll = i & 15;
        c_win[i] = v2f{p.tab.window[32 * n1 + 2 * ll], p.tab.window[32 * n1 + 2 * ll + 1]};
    }
    for (int i = tid; i < 16 * 16; i += kWavesPerBlock * 64) {
        const float2 tw = p.tab.tw256[(i & 15) * 16 + (i >> 4)];  // row k1, lane n2
        c_twa[i] = v2f{tw.x, tw.y};
    }
    for (int i = tid; i < 8 * 16; i += kWavesPerBlock * 64) {
        const float2 tw = p.tab.tw512[1 + (i & 15) + 16 * (i >> 4)];
        c_twp[i] = v2f{tw.x, tw.y};
    }
    if constexpr (FUSED) {
        for (int i = tid; i < p.lfr_m * p.n_mels; i += kWavesPerBlock * 64)
            cmvn[i] = make_float2(p.tab.cmvn_shift[i], p.tab.cmvn_scale[i]);
    }
    __syncthreads();

    float* wl = lds + wave * kWaveLds;        // this wave's region
    float* fl = wl + f * kFrameLds;           // this frame's region
    float* prow = fl + 16 * (f & 1);          // its power row
    const float preemph = 0.97f;

    // Samples [160 t0, 160 t0 + 880) of a group are fetched one group ahead, 8 per lane and pass, and wait in
    // registers while the previous group is computed.  Chunks that cross the end of the clip, and clips whose
    // address does not allow 16-byte loads, are read sample by sample when the group is staged.
    const int t_lim = T;
    auto group_t0 = [&](int g) { return blockIdx.x * kFramesPerBlock + (g * kWavesPerBlock + wave) * kFramesPerWave; };
    auto fast_chunk = [&](int t0, int c) {
        return vec_ok && c < kStageChunks && static_cast<int64_t>(kFrameShift) * t0 + 8 * c + 8 <= len;
    };
    Raw8<PcmT> raw[2];
    auto prefetch = [&](int g) {
        const int t0 = group_t0(g);
        if (g >= kGroupsPerWave || t0 >= t_lim) return;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int c = lane + 64 * r;
            if (fast_chunk(t0, c)) raw[r].load(row + static_cast<int64_t>(kFrameShift) * t0 + 8 * c);
        }
    };
    prefetch(0);

    for (int g = 0; g < kGroupsPerWave; ++g) {
        const int t0 = group_t0(g);
        if (t0 >= p.t_max) break;  // wave-uniform
        const int t = t0 + f;
        if (t0 >= T) {
            // nothing to compute for these frames: zero the padding rows
            if constexpr (!FUSED) {
                if (t < p.t_max) {
                    float* o = p.logmel + (static_cast<int64_t>(b) * p.t_max + t) * p.n_mels;
                    for (int m = l; m < p.n_mels; m += 16) o[m] = 0.0f;
                }
            } else {
                if (t % p.lfr_n == 0) {
                    const int i = t / p.lfr_n;
                    if (i >= Tp && i < p.tp_max) {
                        const int dim = p.lfr_m * p.n_mels;
                        const int64_t off = (static_cast<int64_t>(b) * p.tp_max + i) * dim;
                        for (int c = l; c < dim; c += 16) {
                            if constexpr (OUT_BF16) reinterpret_cast<uint16_t*>(p.feats)[off + c] = 0;
                            else reinterpret_cast<float*>(p.feats)[off + c] = 0.0f;
                        }
                    }
                }
            }
            continue;
        }

        // ---- stage samples [160 t0, 160 t0 + 880) as Kaldi-scaled floats -------------------
        wave_sync();  // previous group's mel reads are done before the region is overwritten
        const int64_t s_base = static_cast<int64_t>(kFrameShift) * t0;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int c = lane + 64 * r;
            if (c < kStageChunks) {
                float v[8];
                if (fast_chunk(t0, c)) {
                    raw[r].get(v);
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int64_t si = s_base + 8 * c + j;
                        v[j] = si < len ? pcm_to_kaldi<PcmT>(row[si]) : 0.0f;
                    }
                }
                *reinterpret_cast<float4*>(wl + 8 * c) = make_float4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<float4*>(wl + 8 * c + 4) = make_float4(v[4], v[5], v[6], v[7]);
            }
        }
        prefetch(g + 1);
        wave_sync();

        // ---- frame samples -> DC removal, pre-emphasis, window ----------------------------------
        v2f z[16];
        {
            const float* xs = wl + kFrameShift * f;
            v2f sv[13];
            float pv[13];
            v2f sum2 = v2f{0.0f, 0.0f};
            // the sample in front of a lane's pair is the second sample of the lane before it (DPP rotate within the
            // frame's 16 lanes: no LDS read); for lane 0 that is lane 15's pair of the row before, and sample 0 stands
            // in for its own predecessor
            float ror_prev = 0.0f;
#pragma unroll
            for (int n1 = 0; n1 < 13; ++n1) {
                const int i0 = 32 * n1 + 2 * l;
                const float2 v = *reinterpret_cast<const float2*>(xs + i0);
                const float ror = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v.y), 0x121, 0xf, 0xf, false));
                const float pr = l > 0 ? ror : (n1 > 0 ? ror_prev : v.x);
                ror_prev = ror;
                const bool ok = (n1 < 12) || (l < 8);  // sample index < 400
                sv[n1] = ok ? v2f{v.x, v.y} : v2f{0.0f, 0.0f};
                pv[n1] = ok ? pr : 0.0f;
                sum2 += sv[n1];
            }
            const float mean = row16_sum(sum2.x + sum2.y) * (1.0f / 400.0f);
            // (a, c) = samples - mean; pre-emphasis subtracts 0.97 x the previous sample, (q, a); then the window.
            // Slots past sample 399 hold zeros times a zero window entry.
#pragma unroll
            for (int n1 = 0; n1 < 13; ++n1) {
                const v2f ac = sv[n1] - mean;
                const v2f qa = v2f{pv[n1], sv[n1].x} - mean;
                z[n1] = (ac - preemph * qa) * c_win[n1 * 16 + l];
            }
            z[13] = z[14] = z[15] = v2f{0.0f, 0.0f};
        }
        wave_sync();  // staged samples consumed; the region becomes the exchange buffer

        // ---- 256-point complex FFT, 16 x 16 --------------------------------------------------
        dft16(z);
#pragma unroll
        for (int k1 = 0; k1 < 16; ++k1) {
            v2f a = z[pos16(k1)];
            if (k1 > 0) a = cmul(a, c_twa[k1 * 16 + l]);
            *reinterpret_cast<v2f*>(fl + k1 * kExRow + 2 * l) = a;
        }
        wave_sync();
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float4 q = *reinterpret_cast<const float4*>(fl + l * kExRow + 4 * j);
            z[2 * j] = v2f{q.x, q.y};
            z[2 * j + 1] = v2f{q.z, q.w};
        }
        dft16(z);
        wave_sync();
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) *reinterpret_cast<v2f*>(fl + 2 * (l + 16 * k2)) = z[pos16(k2)];
        wave_sync();

        // ---- pair Z[k], Z[256-k] -> 4 |X[k]|^2, 4 |X[256-k]|^2 ---------------------------------
        //      X[k] = E + W O, X[256-k]* = E - W O with E = (Z[k] + Z[256-k]*) , O = -i (Z[k] - Z[256-k]*)
        v2f pw[8];  // (power at k, power at 256 - k)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = 1 + l + 16 * i;
            const v2f zk = *reinterpret_cast<const v2f*>(fl + 2 * k);
            const v2f zq = *reinterpret_cast<const v2f*>(fl + 2 * (kNc - k));
            const v2f e = add_conj(zk, zq);
            const v2f t = cmul_neg_i(sub_conj(zk, zq), c_twp[i * 16 + l]);
            const v2f x1 = e + t, x2 = e - t;
            const v2f s1 = x1 * x1, s2 = x2 * x2;
            pw[i] = v2f{s1.x + s1.y, s2.x + s2.y};
        }
        wave_sync();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = 1 + l + 16 * i;
            prow[k] = pw[i].x;
            prow[kNc - k] = pw[i].y;
        }
        // the padded tails of the filters run past the last bin: zero weights meet zeros there
        prow[kNc + l] = 0.0f;
        prow[kNc + 16 + l] = 0.0f;
        wave_sync();

        // ---- mel + log --------------------------------------------------------------------------
        float mel[kMelGroups];
#pragma unroll
        for (int gi = 0; gi < kMelGroups; ++gi) {
            const float* pr = prow + mel_sb[gi * 16 + l];
            const int nc = STD_MEL ? std_mel_chunks(gi) : p.tab.mel_gw[gi] >> 2;
            const float4* wq = reinterpret_cast<const float4*>(mel_w) +
                               (STD_MEL ? std_mel_first(gi) : p.tab.mel_go[gi] >> 2) * 16 + l;
            float acc = 0.0f;
#pragma unroll
            for (int c = 0; c < nc; ++c) {
                const float4 w = wq[c * 16];
                acc = fmaf(w.x, pr[4 * c], acc);
                acc = fmaf(w.y, pr[4 * c + 1], acc);
                acc = fmaf(w.z, pr[4 * c + 2], acc);
                acc = fmaf(w.w, pr[4 * c + 3], acc);
            }
            mel[gi] = __logf(fmaxf(acc, 1.1920929e-07f));
        }

        if (t < T && t < p.t_max) {
            if constexpr (!FUSED) {
                float* o = p.logmel + (static_cast<int64_t>(b) * p.t_max + t) * p.n_mels;
#pragma unroll
                for (int gi = 0; gi < kMelGroups; ++gi) {
                    const int m = l + 16 * gi;
                    if (m < p.n_mels) o[m] = mel[gi];
                }
            } else {
                const int dim = p.lfr_m * p.n_mels;
                char* fb = reinterpret_cast<char*>(p.feats) +
                           static_cast<int64_t>(b) * p.tp_max * dim * (OUT_BF16 ? 2 : 4);
                lfr_scatter<OUT_BF16>(fb, t, T, Tp, Tp < p.tp_max ? Tp : p.tp_max, l, p.n_mels, p.lfr_m, p.lfr_n,
                                      cmvn, mel);
            }
        } else if (t < p.t_max) {
            if constexpr (!FUSED) {
                float* o = p.logmel + (static_cast<int64_t>(b) * p.t_max + t) * p.n_mels;
                for (int m = l; m < p.n_mels; m += 16) o[m] = 0.0f;
            } else {
                if (t % p.lfr_n == 0) {
                    const int i = t / p.lfr_n;
                    if (i >= Tp && i < p.tp_max) {
                        const int dim = p.lfr_m * p.n_mels;
                        const int64_t off = (static_cast<int64_t>(b) * p.tp_max + i) * dim;
                        for (int c = l; c < dim; c += 16) {
                            if constexpr (OUT_BF16) reinterpret_cast<uint16_t*>(p.feats)[off + c] = 0;
                            else reinterpret_cast<float*>(p.feats)[off + c] = 0.0f;
                        }
                    }
                }
            }
        }
    }
}

// Stand-alone apply_lfr + apply_cmvn: one thread per output element group of 4 columns.
template <bool OUT_BF16>
__global__ __launch_bounds__(256) void lfr_cmvn_kernel(const float* __restrict__ logmel,
                                                        const int32_t* __restrict__ n_frames, int t_max, int tp_max,
                                                        int n_mels, int lfr_m, int lfr_n,
                                                        const float* __restrict__ shift,
                                                        const float* __restrict__ scale, void* __restrict__ feats,
                                                        int32_t* __restrict__ lens_out) {
    const int b = blockIdx.z;
    const int i = blockIdx.y;
    // frames past the rows of logmel do not exist, LFR rows past the rows of feats are dropped
    const int T = n_frames[b] < t_max ? (n_frames[b] < 0 ? 0 : n_frames[b]) : t_max;
    const int Tp_all = (T + lfr_n - 1) / lfr_n;
    const int Tp = Tp_all < tp_max ? Tp_all : tp_max;
    if (i == 0 && blockIdx.x == 0 && threadIdx.x == 0 && lens_out) lens_out[b] = Tp;
    const int dim = lfr_m * n_mels;
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= dim) return;
    const int64_t off = (static_cast<int64_t>(b) * tp_max + i) * dim + col;
    float y = 0.0f;
    if (i < Tp) {
        const int j = col / n_mels;
        const int m = col - j * n_mels;
        int src = lfr_n * i + j - (lfr_m - 1) / 2;
        src = src < 0 ? 0 : (src > T - 1 ? T - 1 : src);
        y = (logmel[(static_cast<int64_t>(b) * t_max + src) * n_mels + m] + shift[col]) * scale[col];
    }
    if constexpr (OUT_BF16) reinterpret_cast<uint16_t*>(feats)[off] = f32_to_bf16(y);
    else reinterpret_cast<float*>(feats)[off] = y;
}

int32_t launch_fbank(asrfe_ctx* ctx, const FbankParams& p, int pcm_dtype, bool fused, int out_dtype, hipStream_t s) {
    if (p.batch <= 0 || p.t_max <= 0) return ASRFE_OK;
    dim3 grid((p.t_max + kFramesPerBlock - 1) / kFramesPerBlock, p.batch, 1);
    dim3 block(kWavesPerBlock * 64);
    const bool bf = out_dtype == ASRFE_DT_BF16;
    bool std_mel = true;
    for (int g = 0; g < kMelGroups; ++g)
        std_mel = std_mel && p.tab.mel_gw[g] == 4 * std_mel_chunks(g) && p.tab.mel_go[g] == 4 * std_mel_first(g);
    auto go = [&](auto pcm_tag, auto fused_tag, auto bf_tag) {
        using PcmT = decltype(pcm_tag);
        if (std_mel)
            hipLaunchKernelGGL((fbank_kernel<PcmT, decltype(fused_tag)::value, decltype(bf_tag)::value, true>), grid, block,
                               0, s, p);
        else
            hipLaunchKernelGGL((fbank_kernel<PcmT, decltype(fused_tag)::value, decltype(bf_tag)::value, false>), grid,
                               block, 0, s, p);
    };
    auto go_pcm = [&](auto fused_tag, auto bf_tag) {
        if (pcm_dtype == ASRFE_PCM_I16) go(int16_t{}, fused_tag, bf_tag);
        else if (pcm_dtype == ASRFE_PCM_F32) go(float{}, fused_tag, bf_tag);
        else go(UnitF32{}, fused_tag, bf_tag);
    };
    if (!fused) go_pcm(std::false_type{}, std::false_type{});
    else if (bf) go_pcm(std::true_type{}, std::true_type{});
    else go_pcm(std::true_type{}, std::false_type{});
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_err(ctx, ASRFE_ERR_HIP, std::string("fbank launch: ") + hipGetErrorString(e));
    return ASRFE_OK;
}

int32_t launch_lfr_cmvn(asrfe_ctx* ctx, const float* logmel, const int32_t* n_frames, int batch, int t_max, int tp_max,
                        void* feats, int out_dtype, int32_t* lens_out, hipStream_t s) {
    if (batch <= 0 || tp_max <= 0) return ASRFE_OK;
    const int dim = ctx->cfg.lfr_m * ctx->cfg.n_mels;
    dim3 grid((dim + 255) / 256, tp_max, batch);
    if (out_dtype == ASRFE_DT_BF16)
        hipLaunchKernelGGL(lfr_cmvn_kernel<true>, grid, dim3(256), 0, s, logmel, n_frames, t_max, tp_max,
                           ctx->cfg.n_mels, ctx->cfg.lfr_m, ctx->cfg.lfr_n, ctx->tab.cmvn_shift, ctx->tab.cmvn_scale,
                           feats, lens_out);
    else
        hipLaunchKernelGGL(lfr_cmvn_kernel<false>, grid, dim3(256), 0, s, logmel, n_frames, t_max, tp_max,
                           ctx->cfg.n_mels, ctx->cfg.lfr_m, ctx->cfg.lfr_n, ctx->tab.cmvn_shift, ctx->tab.cmvn_scale,
                           feats, lens_out);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_err(ctx, ASRFE_ERR_HIP, std::string("lfr_cmvn launch: ") + hipGetErrorString(e));
    return ASRFE_OK;
}

}  // namespace asrfe
