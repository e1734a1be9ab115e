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
//     table is zero-padded per group so the trip counts are wave-uniform.
#include "common.h"

namespace asrfe {

constexpr int kFramesPerWave = 4;
constexpr int kGroupsPerWave = 8;
constexpr int kWavesPerBlock = 4;
constexpr int kFramesPerBlock = kFramesPerWave * kGroupsPerWave * kWavesPerBlock;  // 128
constexpr int kStageSamples = kFrameShift * (kFramesPerWave - 1) + kFrameLen;       // 880
constexpr int kStageChunks = kStageSamples / 8;                                     // 110
constexpr int kExRow = 36;                             // floats per exchange row (16 complex + 16 B pad)
constexpr int kFrameLds = 16 * kExRow;                 // 576 floats per frame
constexpr int kWaveLds = kFramesPerWave * kFrameLds;   // 2304 floats = 9216 B per wave
static_assert(kWaveLds >= kStageSamples + 16, "staging area must fit the wave region");
static_assert(kFrameLds >= 2 * kNc, "Z row must fit the frame region");

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// 4-point forward DFT in place, natural output order.
__device__ __forceinline__ void dft4(float& r0, float& i0, float& r1, float& i1, float& r2, float& i2,
                                     float& r3, float& i3) {
    const float ar = r0 + r2, ai = i0 + i2;
    const float br = r0 - r2, bi = i0 - i2;
    const float cr = r1 + r3, ci = i1 + i3;
    const float dr = r1 - r3, di = i1 - i3;
    r0 = ar + cr; i0 = ai + ci;
    r2 = ar - cr; i2 = ai - ci;
    r1 = br + di; i1 = bi - dr;
    r3 = br - di; i3 = bi + dr;
}

__device__ __forceinline__ void cmul(float& r, float& i, const float wr, const float wi) {
    const float tr = r * wr - i * wi;
    i = r * wi + i * wr;
    r = tr;
}

// 16-point forward DFT in registers.  Input x[n] at index n; output X[k] at index 4 (k & 3) + (k >> 2).
__device__ __forceinline__ void dft16(float (&re)[16], float (&im)[16]) {
    constexpr float c1 = 0.92387953251128674f;  // cos(pi/8)
    constexpr float s1 = 0.38268343236508977f;  // sin(pi/8)
    constexpr float h = 0.70710678118654752f;
    float yr[16], yi[16];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        float r0 = re[b], i0 = im[b], r1 = re[4 + b], i1 = im[4 + b];
        float r2 = re[8 + b], i2 = im[8 + b], r3 = re[12 + b], i3 = im[12 + b];
        dft4(r0, i0, r1, i1, r2, i2, r3, i3);
        yr[b] = r0; yi[b] = i0;
        yr[4 + b] = r1; yi[4 + b] = i1;
        yr[8 + b] = r2; yi[8 + b] = i2;
        yr[12 + b] = r3; yi[12 + b] = i3;
    }
    // y[4 c + b] *= W16^(b c)
    cmul(yr[5], yi[5], c1, -s1);    // b1 c1 : W^1
    cmul(yr[6], yi[6], h, -h);      // b2 c1 : W^2
    cmul(yr[7], yi[7], s1, -c1);    // b3 c1 : W^3
    cmul(yr[9], yi[9], h, -h);      // b1 c2 : W^2
    cmul(yr[10], yi[10], 0.f, -1.f);  // b2 c2 : W^4
    cmul(yr[11], yi[11], -h, -h);   // b3 c2 : W^6
    cmul(yr[13], yi[13], s1, -c1);  // b1 c3 : W^3
    cmul(yr[14], yi[14], -h, -h);   // b2 c3 : W^6
    cmul(yr[15], yi[15], -c1, s1);  // b3 c3 : W^9
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        dft4(yr[4 * c], yi[4 * c], yr[4 * c + 1], yi[4 * c + 1], yr[4 * c + 2], yi[4 * c + 2], yr[4 * c + 3],
             yi[4 * c + 3]);
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        re[k] = yr[k];
        im[k] = yi[k];
    }
}

__device__ __forceinline__ constexpr int pos16(int k) { return 4 * (k & 3) + (k >> 2); }

template <typename PcmT>
__device__ __forceinline__ float pcm_to_kaldi(PcmT v);
template <>
__device__ __forceinline__ float pcm_to_kaldi<int16_t>(int16_t v) { return static_cast<float>(v); }
template <>
__device__ __forceinline__ float pcm_to_kaldi<float>(float v) { return v * 32768.0f; }

// Loads samples [s, s + 8) of one clip as Kaldi-scaled floats, zero past the clip's end.
template <typename PcmT>
__device__ __forceinline__ void load8(const PcmT* __restrict__ row, int64_t s, int64_t len, bool vec_ok,
                                      float (&out)[8]) {
    if (vec_ok && s + 8 <= len) {
        if constexpr (sizeof(PcmT) == 2) {
            const int4 raw = *reinterpret_cast<const int4*>(row + s);
            const int w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                out[2 * j] = static_cast<float>(static_cast<int16_t>(w[j] & 0xffff));
                out[2 * j + 1] = static_cast<float>(w[j] >> 16);
            }
        } else {
            const float4 a = *reinterpret_cast<const float4*>(row + s);
            const float4 b = *reinterpret_cast<const float4*>(row + s + 4);
            out[0] = a.x * 32768.0f; out[1] = a.y * 32768.0f; out[2] = a.z * 32768.0f; out[3] = a.w * 32768.0f;
            out[4] = b.x * 32768.0f; out[5] = b.y * 32768.0f; out[6] = b.z * 32768.0f; out[7] = b.w * 32768.0f;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) out[j] = (s + j < len) ? pcm_to_kaldi<PcmT>(row[s + j]) : 0.0f;
    }
}

__device__ __forceinline__ uint16_t f32_to_bf16(float v) {
    return __builtin_bit_cast(uint16_t, static_cast<__bf16>(v));
}

// Writes one log-mel value of frame t into every LFR/CMVN feature slot that holds it.
// Row i, slot j holds frame clamp(n i + j - left, 0, T-1), left = (m-1)/2; rows >= rows_out are dropped.
template <bool OUT_BF16>
__device__ __forceinline__ void lfr_scatter(void* __restrict__ feats_b, int t, int T, int Tp, int rows_out, int mel,
                                            int n_mels, int lfr_m, int lfr_n, const float* __restrict__ shift,
                                            const float* __restrict__ scale, float v) {
    const int left = (lfr_m - 1) / 2;
    const int dim = lfr_m * n_mels;
    const int a = t + left - (lfr_m - 1);            // first row: n i >= a
    int i_lo = a <= 0 ? 0 : (a + lfr_n - 1) / lfr_n;
    int i_hi = (t == T - 1) ? Tp - 1 : (t + left) / lfr_n;  // the last frame also fills every slot past the end
    if (i_hi > rows_out - 1) i_hi = rows_out - 1;
    for (int i = i_lo; i <= i_hi; ++i) {
        const int je = t + left - lfr_n * i;         // slot that holds t exactly
        int j_lo = (t == 0) ? 0 : je;                // frame 0 also fills the left padding slots
        int j_hi = (t == T - 1) ? lfr_m - 1 : je;
        if (j_lo < 0) j_lo = 0;
        if (j_hi > lfr_m - 1) j_hi = lfr_m - 1;
        for (int j = j_lo; j <= j_hi; ++j) {
            const int col = j * n_mels + mel;
            const float y = (v + shift[col]) * scale[col];
            if constexpr (OUT_BF16)
                reinterpret_cast<uint16_t*>(feats_b)[static_cast<int64_t>(i) * dim + col] = f32_to_bf16(y);
            else
                reinterpret_cast<float*>(feats_b)[static_cast<int64_t>(i) * dim + col] = y;
        }
    }
}

template <typename PcmT, bool FUSED, bool OUT_BF16>
__global__ __launch_bounds__(kWavesPerBlock * 64) void fbank_kernel(const FbankParams p) {
    __shared__ __attribute__((aligned(16))) float lds[kWavesPerBlock * kWaveLds + kMelMaxTerms * 16 + kMelGroups * 16];
    float* mel_w = lds + kWavesPerBlock * kWaveLds;
    int* mel_sb = reinterpret_cast<int*>(mel_w + kMelMaxTerms * 16);

    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const int lane = tid & 63;
    const int f = lane >> 4;   // frame of the group
    const int l = lane & 15;   // lane of the frame
    const int b = blockIdx.y;

    for (int i = tid; i < kMelMaxTerms * 16; i += kWavesPerBlock * 64) mel_w[i] = p.tab.mel_w[i];
    for (int i = tid; i < kMelGroups * 16; i += kWavesPerBlock * 64) mel_sb[i] = p.tab.mel_sb[i];
    __syncthreads();

    const int64_t len = p.pcm_lens[b];
    const int T = len < kFrameLen ? 0 : 1 + static_cast<int>((len - kFrameLen) / kFrameShift);
    const int Tp = (T + p.lfr_n - 1) / p.lfr_n;
    if (blockIdx.x == 0 && tid == 0) {
        if (p.n_frames) p.n_frames[b] = T;
        if (p.lfr_lens) p.lfr_lens[b] = Tp;
    }
    const PcmT* __restrict__ row = reinterpret_cast<const PcmT*>(p.pcm) + static_cast<int64_t>(b) * p.pcm_stride;
    const bool vec_ok = p.aligned16 != 0;

    // per-lane constants
    float w0[13], w1[13];
#pragma unroll
    for (int n1 = 0; n1 < 13; ++n1) {
        w0[n1] = p.tab.window[32 * n1 + 2 * l];
        w1[n1] = p.tab.window[32 * n1 + 2 * l + 1];
    }
    float2 twa[16];
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) twa[k1] = p.tab.tw256[l * 16 + k1];
    float2 twp[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) twp[i] = p.tab.tw512[1 + l + 16 * i];

    float* wl = lds + wave * kWaveLds;        // this wave's region
    float* fl = wl + f * kFrameLds;           // this frame's region
    const float preemph = 0.97f;

    for (int g = 0; g < kGroupsPerWave; ++g) {
        const int t0 = blockIdx.x * kFramesPerBlock + (g * kWavesPerBlock + wave) * kFramesPerWave;
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
                load8<PcmT>(row, s_base + 8 * c, len, vec_ok, v);
                *reinterpret_cast<float4*>(wl + 8 * c) = make_float4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<float4*>(wl + 8 * c + 4) = make_float4(v[4], v[5], v[6], v[7]);
            }
        }
        wave_sync();

        // ---- frame samples -> DC removal, pre-emphasis, window ----------------------------------
        float re[16], im[16];
        {
            const float* xs = wl + kFrameShift * f;
            float sr[13], si[13], pv[13];
            float sum = 0.0f;
#pragma unroll
            for (int n1 = 0; n1 < 13; ++n1) {
                const int i0 = 32 * n1 + 2 * l;
                const float2 v = *reinterpret_cast<const float2*>(xs + i0);
                const float pr = xs[i0 > 0 ? i0 - 1 : 0];
                const bool ok = (n1 < 12) || (l < 8);  // sample index < 400
                sr[n1] = ok ? v.x : 0.0f;
                si[n1] = ok ? v.y : 0.0f;
                pv[n1] = ok ? pr : 0.0f;
                sum += sr[n1] + si[n1];
            }
            sum += __shfl_xor(sum, 1);
            sum += __shfl_xor(sum, 2);
            sum += __shfl_xor(sum, 4);
            sum += __shfl_xor(sum, 8);
            const float mean = sum / 400.0f;
#pragma unroll
            for (int n1 = 0; n1 < 13; ++n1) {
                const float a = sr[n1] - mean;
                const float c = si[n1] - mean;
                const float q = pv[n1] - mean;
                re[n1] = (a - preemph * q) * w0[n1];
                im[n1] = (c - preemph * a) * w1[n1];
            }
            re[13] = re[14] = re[15] = 0.0f;
            im[13] = im[14] = im[15] = 0.0f;
        }
        wave_sync();  // staged samples consumed; the region becomes the exchange buffer

        // ---- 256-point complex FFT, 16 x 16 --------------------------------------------------
        dft16(re, im);
#pragma unroll
        for (int k1 = 0; k1 < 16; ++k1) {
            float ar = re[pos16(k1)], ai = im[pos16(k1)];
            if (k1 > 0) cmul(ar, ai, twa[k1].x, twa[k1].y);
            *reinterpret_cast<float2*>(fl + k1 * kExRow + 2 * l) = make_float2(ar, ai);
        }
        wave_sync();
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float4 q = *reinterpret_cast<const float4*>(fl + l * kExRow + 4 * j);
            re[2 * j] = q.x; im[2 * j] = q.y;
            re[2 * j + 1] = q.z; im[2 * j + 1] = q.w;
        }
        dft16(re, im);
        wave_sync();
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2)
            *reinterpret_cast<float2*>(fl + 2 * (l + 16 * k2)) = make_float2(re[pos16(k2)], im[pos16(k2)]);
        wave_sync();

        // ---- pair Z[k], Z[256-k] -> 4 |X[k]|^2, 4 |X[256-k]|^2 ---------------------------------
        float pk[8], pq[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = 1 + l + 16 * i;
            const float2 zk = *reinterpret_cast<const float2*>(fl + 2 * k);
            const float2 zq = *reinterpret_cast<const float2*>(fl + 2 * (kNc - k));
            const float er = zk.x + zq.x, ei = zk.y - zq.y;
            const float orr = zk.y + zq.y, oi = zq.x - zk.x;
            const float tr = twp[i].x * orr - twp[i].y * oi;
            const float ti = twp[i].x * oi + twp[i].y * orr;
            const float x1r = er + tr, x1i = ei + ti;
            const float x2r = er - tr, x2i = ei - ti;
            pk[i] = x1r * x1r + x1i * x1i;
            pq[i] = x2r * x2r + x2i * x2i;
        }
        wave_sync();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = 1 + l + 16 * i;
            fl[k] = pk[i];
            fl[kNc - k] = pq[i];
        }
        wave_sync();

        // ---- mel + log --------------------------------------------------------------------------
        float mel[kMelGroups];
#pragma unroll
        for (int gi = 0; gi < kMelGroups; ++gi) {
            const int sb = mel_sb[gi * 16 + l];
            const int gw = p.tab.mel_gw[gi];
            const float* wrow = mel_w + p.tab.mel_go[gi] * 16 + l;
            float acc = 0.0f;
            for (int j = 0; j < gw; ++j) {
                int k = sb + j;
                k = k > kNc - 1 ? kNc - 1 : k;
                acc = fmaf(wrow[j * 16], fl[k], acc);
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
#pragma unroll
                for (int gi = 0; gi < kMelGroups; ++gi) {
                    const int m = l + 16 * gi;
                    if (m < p.n_mels)
                        lfr_scatter<OUT_BF16>(fb, t, T, Tp, Tp < p.tp_max ? Tp : p.tp_max, m, p.n_mels, p.lfr_m, p.lfr_n, p.tab.cmvn_shift,
                                              p.tab.cmvn_scale, mel[gi]);
                }
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
    const int T = n_frames[b];
    const int Tp = (T + lfr_n - 1) / lfr_n;
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
    const bool i16 = pcm_dtype == ASRFE_PCM_I16;
    const bool bf = out_dtype == ASRFE_DT_BF16;
    if (!fused) {
        if (i16) hipLaunchKernelGGL((fbank_kernel<int16_t, false, false>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((fbank_kernel<float, false, false>), grid, block, 0, s, p);
    } else if (bf) {
        if (i16) hipLaunchKernelGGL((fbank_kernel<int16_t, true, true>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((fbank_kernel<float, true, true>), grid, block, 0, s, p);
    } else {
        if (i16) hipLaunchKernelGGL((fbank_kernel<int16_t, true, false>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((fbank_kernel<float, true, false>), grid, block, 0, s, p);
    }
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
