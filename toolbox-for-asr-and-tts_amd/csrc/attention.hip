// Masked multi-head self-attention for the SAN-M encoder (d_k = 128) on gfx950, flash style.
//
//   One workgroup = 4 waves = 128 query rows of one (utterance, head); each wave owns 32 query rows
//   and walks the key axis in tiles of 64 with an online softmax.
//   Scores are computed transposed, S^T = K Q^T (v_mfma_f32_32x32x16_bf16 with the K tile as the A
//   operand and Q as the B operand), so a lane holds 32 key scores of ONE query row: the row max and
//   row sum are register reductions plus one exchange with lane ^ 32, and the probabilities are
//   already laid out as the B operand of the second product O^T = V^T P^T, whose accumulator again has
//   the query on the lane, so the running rescale is a per-lane scalar.
//   K tiles go global -> LDS by 16-byte LDS-DMA loads with the bank swizzle on the source address
//   (chunk c of key row r sits at chunk c ^ (r & 15)); V arrives pre-transposed ([d][t], written by the
//   FSMN kernel) and is register-staged into rows padded to 136 bytes so that the 8-byte fragment
//   reads of 32 lanes hit 32 different bank pairs.
#include "common.h"
#include "elem.h"
#include "encoder.h"

namespace asrfe {

constexpr int kAttnThreads = 256;
constexpr int kKTileBytes = 64 * 256;      // 64 keys x 128 d bf16
constexpr int kVRow = 136;                 // bytes per V^T row (64 keys bf16 + 8 pad)
constexpr int kVTileBytes = 128 * kVRow;

#define GLDS16(gptr, lptr)                                                                         \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),       \
                                     (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

template <bool F16>
__global__ __launch_bounds__(kAttnThreads) void attn_kernel(const AttnParams p) {
    __shared__ __attribute__((aligned(16))) char smem[kKTileBytes + kVTileBytes];
    char* ks = smem;
    char* vs = smem + kKTileBytes;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int b = blockIdx.z, head = blockIdx.y;
    int len = p.lens[b];
    len = len < 0 ? 0 : (len > p.Tp ? p.Tp : len);
    const int q0 = blockIdx.x * 128 + wave * 32;
    const int64_t row0 = static_cast<int64_t>(b) * p.Tp;
    const int n_kt = (len + 63) >> 6;

    // Q fragments: lane holds Q[q0 + l31][16 s + 8 h .. +8), s = 0..7
    u16x8 qf[8];
    {
        int qr = q0 + l31;
        qr = qr < p.Tp ? qr : p.Tp - 1;
        const uint16_t* qp = p.qkv + (row0 + qr) * p.ld_qkv + head * 128 + 8 * h;
#pragma unroll
        for (int s = 0; s < 8; ++s) qf[s] = *reinterpret_cast<const u16x8*>(qp + 16 * s);
    }

    f32x16 oacc[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[mt][e] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f;
    const float c = p.scale_log2;

    // staging addresses
    const int k_row = tid >> 4;                                   // + 16 i
    const int k_chunk = (tid & 15) ^ ((tid >> 4) & 15);           // source chunk for LDS slot (row, tid & 15)
    const int v_row = tid >> 3;                                   // + 32 i
    const int v_chunk = tid & 7;
    const uint16_t* vbase = p.vt + (static_cast<int64_t>(b) * 512 + head * 128) * p.Tk;

    for (int kt = 0; kt < n_kt; ++kt) {
        __syncthreads();  // previous tile fully consumed
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int key = kt * 64 + i * 16 + k_row;
            key = key < p.Tp ? key : p.Tp - 1;
            GLDS16(p.qkv + (row0 + key) * p.ld_qkv + 512 + head * 128 + k_chunk * 8, ks + i * 4096 + wave * 1024);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int d = i * 32 + v_row;
            const uint2* src = reinterpret_cast<const uint2*>(vbase + static_cast<int64_t>(d) * p.Tk + kt * 64 + v_chunk * 8);
            const uint2 a = src[0], bq = src[1];
            uint2* dst = reinterpret_cast<uint2*>(vs + d * kVRow + v_chunk * 16);
            dst[0] = a;
            dst[1] = bq;
        }
        __syncthreads();  // drains the LDS-DMA loads and publishes the V writes

        // ---- S^T = K Q^T : two 32-key sub-tiles ---------------------------------------------
        f32x16 sacc[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int e = 0; e < 16; ++e) sacc[kk][e] = 0.0f;
            const char* krow = ks + (kk * 32 + l31) * 256;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const u16x8 kf = *reinterpret_cast<const u16x8*>(krow + (((2 * s + h) ^ (l31 & 15)) << 4));
                sacc[kk] = mfma32<F16>(kf, qf[s], sacc[kk]);
            }
        }
        if (kt == n_kt - 1) {  // wave-uniform: mask keys at or past the utterance length
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
        rmax = fmaxf(rmax, __shfl_xor(rmax, 32));
        const float m_new = fmaxf(m_run, rmax);  // finite: every tile holds at least one valid key
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
        const float mc = m_new * c;
        m_run = m_new;
        float psum = 0.0f;
        u16x8 pf[2][2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float pv = __builtin_amdgcn_exp2f(sacc[kk][8 * s2 + j] * c - mc);
                    psum += pv;
                    pf[kk][s2][j] = to_elem<F16>(pv);
                }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int e = 0; e < 16; ++e) oacc[mt][e] *= alpha;

        // ---- O^T += V^T P^T ----------------------------------------------------------------------
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const char* vrow = vs + (mt * 32 + l31) * kVRow + 8 * h;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    // element j of the fragment is key 32 kk + 16 s2 + 8 (j >> 2) + 4 h + (j & 3)
                    const u16x4 lo = *reinterpret_cast<const u16x4*>(vrow + (kk * 32 + s2 * 16) * 2);
                    const u16x4 hi = *reinterpret_cast<const u16x4*>(vrow + (kk * 32 + s2 * 16 + 8) * 2);
                    const u16x8 vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    oacc[mt] = mfma32<F16>(vf, pf[kk][s2], oacc[mt]);
                }
        }
    }

    // ---- normalise and store: lane owns query row q0 + l31, register quad g owns d = 8 g + 4 h + {0..3} ----
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = l_tot > 0.0f ? 1.0f / l_tot : 0.0f;
    const int qr = q0 + l31;
    const bool ok = qr < p.Tp;
    uint16_t* orow = p.out + (row0 + (ok ? qr : 0)) * p.ld_out + head * 128;
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
            if (ok) *reinterpret_cast<uint4*>(orow + mt * 32 + 16 * gp + 8 * h) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
        }
}

int32_t launch_attention(asrfe_ctx* ctx, const AttnParams& p, hipStream_t s) {
    if (p.B <= 0 || p.Tp <= 0) return ASRFE_OK;
    if (p.Tk % 64 != 0 || p.Tk < p.Tp) return set_err(ctx, ASRFE_ERR_SHAPE, "attention: Tk must be Tp rounded up to 64");
    dim3 grid((p.Tp + 127) / 128, 4, p.B);
    if (ctx->cfg.enc_compute_dtype == ASRFE_DT_F16) hipLaunchKernelGGL(attn_kernel<true>, grid, dim3(kAttnThreads), 0, s, p);
    else hipLaunchKernelGGL(attn_kernel<false>, grid, dim3(kAttnThreads), 0, s, p);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_err(ctx, ASRFE_ERR_HIP, std::string("attention launch: ") + hipGetErrorString(e));
    return ASRFE_OK;
}

}  // namespace asrfe
