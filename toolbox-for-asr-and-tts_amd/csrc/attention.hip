// Masked multi-head self-attention for the SAN-M encoder (d_k = 128) on gfx950, flash style.
//
//   One workgroup = 4 waves = 128 query rows of one (utterance, head); each wave owns 32 query rows
//   and walks the key axis in tiles of 64 with an online softmax.
//   Scores are computed transposed, S^T = K Q^T (v_mfma_f32_32x32x16_bf16 with the K tile as the A
//   operand and Q as the B operand), so a lane holds 32 key scores of ONE query row: the row max and
//   row sum are register reductions plus one exchange with lane ^ 32, and the probabilities are
//   already laid out as the B operand of the second product O^T = V^T P^T, whose accumulator again has
//   the query on the lane, so the running rescale is a per-lane scalar.
//   K tiles go global -> LDS by 16-byte LDS-DMA loads (lds_dma.h) with the bank swizzle on the source address
//   (chunk c of key row r sits at chunk c ^ (r & 15)); V arrives pre-transposed ([d][t], written by the
//   FSMN kernel) and is register-staged into rows padded to 136 bytes so that the 8-byte fragment
//   reads of 32 lanes hit 32 different bank pairs.
#include "common.h"
#include "elem.h"
#include "encoder.h"
#include "lds_dma.h"

namespace asrfe {

constexpr int kAttnThreads = 256;
constexpr int kKTileBytes = 64 * 256;      // 64 keys x 128 d bf16
constexpr int kVRow = 136;                 // bytes per V^T row (64 keys bf16 + 8 pad)
constexpr int kVTileBytes = 128 * kVRow;

template <bool F16>
__global__ __launch_bounds__(kAttnThreads, 2) void attn_kernel(const AttnParams p) {
    // two K/V buffers: tile kt + 1 is fetched while tile kt is multiplied
    __shared__ __attribute__((aligned(16))) char smem[2 * (kKTileBytes + kVTileBytes)];
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
    len = len < 0 ? 0 : (len > p.Tp ? p.Tp : len);
    const int q0 = qtile * 128 + wave * 32;
    const int64_t row0 = static_cast<int64_t>(b) * p.Tp;
    const int n_kt = (len + 63) >> 6;

    // staging addresses
    const int k_row = tid >> 4;                                   // + 16 i
    const int k_chunk = (tid & 15) ^ ((tid >> 4) & 15);           // source chunk for LDS slot (row, tid & 15)
    const int v_row = tid >> 3;                                   // + 32 i
    const int v_chunk = tid & 7;
    const uint16_t* vbase = p.vt + (static_cast<int64_t>(b) * 512 + head * 128) * p.Tk;
    // K tile source: scalar base (this utterance, this head) + per-lane byte offset (key row, swizzled chunk)
    const char* kbase = reinterpret_cast<const char*>(p.qkv + row0 * p.ld_qkv + 512 + head * 128);
    const uint32_t smem_base = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));

    auto k_issue = [&](int kt, int buf) {
        const uint32_t ks = __builtin_amdgcn_readfirstlane(smem_base + buf * (kKTileBytes + kVTileBytes) + wave * 1024);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int key = kt * 64 + i * 16 + k_row;
            key = key < p.Tp ? key : p.Tp - 1;
            lds_dma16(kbase, static_cast<uint32_t>(key) * static_cast<uint32_t>(p.ld_qkv * 2) + k_chunk * 16, ks + i * 4096);
        }
    };
    uint4 vreg[4];
    auto v_load = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            vreg[i] = *reinterpret_cast<const uint4*>(vbase + static_cast<int64_t>(i * 32 + v_row) * p.Tk + kt * 64 + v_chunk * 8);
    };
    auto v_store = [&](int buf) {
        char* vs = smem + buf * (kKTileBytes + kVTileBytes) + kKTileBytes;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint2* dst = reinterpret_cast<uint2*>(vs + (i * 32 + v_row) * kVRow + v_chunk * 16);
            dst[0] = make_uint2(vreg[i].x, vreg[i].y);
            dst[1] = make_uint2(vreg[i].z, vreg[i].w);
        }
    };

    if (n_kt > 0) {
        k_issue(0, 0);
        v_load(0);
    }
    // Q fragments: lane holds Q[q0 + l31][16 s + 8 h .. +8), s = 0..7
    u16x8 qf[8];
    {
        int qr = q0 + l31;
        qr = qr < p.Tp ? qr : p.Tp - 1;
        const uint16_t* qp = p.qkv + (row0 + qr) * p.ld_qkv + head * 128 + 8 * h;
#pragma unroll
        for (int s = 0; s < 8; ++s) qf[s] = *reinterpret_cast<const u16x8*>(qp + 16 * s);
    }
    if (n_kt > 0) v_store(0);

    f32x16 oacc[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[mt][e] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f;
    const float c = p.scale_log2;

    for (int kt = 0; kt < n_kt; ++kt) {
        const int cur = kt & 1;
        // tile kt is complete in LDS (this thread's LDS-DMA loads are drained by hand: they are issued through
        // inline assembly, which keeps the compiler's LDS-read waits counted; its LDS writes by the barrier's fence)
        // and every wave is done with tile kt - 1, whose buffer is refilled below
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const bool more = kt + 1 < n_kt;
        if (more) {
            k_issue(kt + 1, cur ^ 1);
            v_load(kt + 1);
        }
        const char* ks = smem + cur * (kKTileBytes + kVTileBytes);
        const char* vs = ks + kKTileBytes;

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
        const bool grew = m_new != m_run;
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
        if (__any(grew)) {  // no lane's maximum moved: alpha is exactly 1 everywhere, skip the rescale
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int e = 0; e < 16; ++e) oacc[mt][e] *= alpha;
        }

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
        if (more) v_store(cur ^ 1);
    }

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
        if (qr < p.Tp) *reinterpret_cast<uint4*>(p.out + (row0 + qr) * p.ld_out + head * 128 + cch * 8) = v;
    }
}

int32_t launch_attention(asrfe_ctx* ctx, const AttnParams& p, hipStream_t s) {
    if (p.B <= 0 || p.Tp <= 0) return ASRFE_OK;
    if (p.Tk % 64 != 0 || p.Tk < p.Tp) return set_err(ctx, ASRFE_ERR_SHAPE, "attention: Tk must be Tp rounded up to 64");
    AttnParams q = p;
    q.n_qtiles = (p.Tp + 127) / 128;
    dim3 grid(q.n_qtiles * 4 * p.B);
    if (ctx->cfg.enc_compute_dtype == ASRFE_DT_F16) hipLaunchKernelGGL(attn_kernel<true>, grid, dim3(kAttnThreads), 0, s, q);
    else hipLaunchKernelGGL(attn_kernel<false>, grid, dim3(kAttnThreads), 0, s, q);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_err(ctx, ASRFE_ERR_HIP, std::string("attention launch: ") + hipGetErrorString(e));
    return ASRFE_OK;
}

}  // namespace asrfe
