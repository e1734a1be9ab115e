// What does an LDS-DMA load cost a wave that is alone on its SIMD and busy issuing MFMAs?  The loop of the
// post-attention kernel's feed-forward (two v_mfma_f32_16x16x32 and one ds_read_b128 per weight fragment) with, behind
// every second fragment, one 1 KiB load into LDS in one of several forms:
//   0  no load (the bare loop)
//   1  global_load_lds_dwordx4 with a per-lane 32-bit offset (what the kernels use)
//   2  buffer_load_dwordx4 ... offen lds (per-lane offset through a buffer descriptor)
//   3  buffer_load_dwordx4 off ... lds, the lane's address from the descriptor's ADD_TID_ENABLE (stride 16): no vector
//      register operand at all
//   4  global_load_dwordx4 into registers and ds_write_b128 eight fragments later, two per sixteen fragments (no DMA;
//      the wait for the load is part of the figure: two register sets do not cover the latency)
//   5  only the global_load_dwordx4 of form 4 (destination registers pinned for the whole loop, never waited for
//      inside an iteration), 6  only its ds_write_b128: the issue cost of the two halves of a register path
// and form 1 behind every 4th (the post-attention kernel's rate) and every 8th fragment.
// Prints cycles per fragment and checks that forms 1-3 delivered the right bytes.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/dma_issue.hip -o tools/micro/dma_issue && tools/micro/dma_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef unsigned int u4v __attribute__((ext_vector_type(4)));

constexpr int kLdsHalf = 32768;   // 64 KiB of fp16 the MFMAs read
constexpr int kRing = 32 * 1024;  // bytes the loads land in: 8 KiB per wave, eight 1 KiB slots

template <int FORM, int EVERY>
__global__ __launch_bounds__(256) void k(const _Float16* __restrict__ rnd, const char* __restrict__ src, long long* cyc, float* sink,
                                          unsigned* check, int iters) {
    __shared__ __attribute__((aligned(16))) _Float16 lds[kLdsHalf];
    __shared__ __attribute__((aligned(16))) char ring[kRing];
    __shared__ float pad[12 * 1024];  // one workgroup per CU
    for (int i = threadIdx.x; i < kLdsHalf; i += 256) lds[i] = rnd[(blockIdx.x * 977 + i) % (1 << 20)];
    for (int i = threadIdx.x; i < kRing / 4; i += 256) reinterpret_cast<unsigned*>(ring)[i] = 0;
    if (threadIdx.x == 0) pad[0] = 0.0f;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    h8 b[8], b2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            b[j][e] = rnd[(threadIdx.x * 64 + j * 8 + e + 12345) % (1 << 20)];
            b2[j][e] = rnd[(threadIdx.x * 64 + j * 8 + e + 54321) % (1 << 20)];
        }
    __syncthreads();
    const char* base0 = reinterpret_cast<const char*>(lds) + wave * 16384;
    f4v acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][e] = 0.0f;
    // this wave's source: 64 KiB runs of the buffer, 1 KiB per load, walking forward (L2-resident after the first pass)
    const char* wsrc = src + (static_cast<size_t>(blockIdx.x % 16) * 4 + wave) * 65536;
    const unsigned ring_lds = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(reinterpret_cast<uintptr_t>(ring)) + wave * 8192);
    const unsigned voff = lane * 16;
    // descriptors: plain raw buffer, and one with ADD_TID_ENABLE (bit 23 of word 3) and stride 16 (word 1 bits 16..29)
    const unsigned long long wa = reinterpret_cast<unsigned long long>(wsrc);
    u4v rs_raw = {static_cast<unsigned>(wa), static_cast<unsigned>(wa >> 32) & 0xffffu, 0x7fffffffu, 0x00020000u};
    u4v rs_tid = {static_cast<unsigned>(wa), (static_cast<unsigned>(wa >> 32) & 0xffffu) | (16u << 16), 0x7fffffffu, 0x00800000u};
    u4v held[2];
    held[0] = held[1] = u4v{0, 0, 0, 0};
    u4v pin[8];  // forms 5, 6: destinations / sources that stay allocated over the whole loop (tied at every iteration's end)
#pragma unroll
    for (int j = 0; j < 8; ++j) pin[j] = u4v{1u + j, 2u, 3u, 4u};
    long long t0 = __builtin_readcyclecounter();
    h8 wf[8];
    const int a0 = static_cast<int>(reinterpret_cast<uintptr_t>(base0)) + lane * 16;
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wf[i]) : "v"(a0), "n"(i * 1024));
    for (int it = 0; it < iters; ++it) {
        const int ad = static_cast<int>(reinterpret_cast<uintptr_t>(base0)) + ((lane + it) & 63) * 16;
        const unsigned soff = __builtin_amdgcn_readfirstlane((it & 7) * 8192);  // eight loads per iteration: 8 KiB of source
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const unsigned dst = ring_lds + (i >> 1) * 1024;
            if (FORM >= 1 && FORM <= 3 && (i % EVERY) == 0) asm volatile("s_mov_b32 m0, %0" ::"s"(dst) : "memory");
            asm volatile("s_waitcnt lgkmcnt(7)\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[2 * i]) : "v"(wf[i & 7]), "v"(b[i & 7]));
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[2 * i + 1]) : "v"(wf[i & 7]), "v"(b2[i & 7]));
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wf[i & 7]) : "v"(ad), "n"(((i + 8) & 15) * 1024));
            if ((i % EVERY) == 0) {
                const unsigned so = soff + (i >> 1) * 1024;
                if (FORM == 1) asm volatile("global_load_lds_dwordx4 %0, %1" ::"v"(voff + so), "s"(wsrc) : "memory");
                if (FORM == 2) asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(rs_raw), "s"(so) : "memory");
                if (FORM == 3) asm volatile("buffer_load_dwordx4 off, %0, %1 lds" ::"s"(rs_tid), "s"(so) : "memory");
                if (FORM == 5) asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(pin[i >> 1]) : "v"(voff + so), "s"(wsrc) : "memory");
                if (FORM == 6) asm volatile("ds_write_b128 %0, %1" ::"v"(static_cast<unsigned>(dst + voff)), "v"(pin[i >> 1]) : "memory");
                if (FORM == 4) {
                    if (i == 0 || i == 8) {  // (two loads per iteration in this form: registers are the limit)
                        asm volatile("s_waitcnt vmcnt(1)" : "+v"(held[(i >> 3) & 1]) : : "memory");
                        asm volatile("ds_write_b128 %0, %1" ::"v"(static_cast<unsigned>(dst + voff)), "v"(held[(i >> 3) & 1]) : "memory");
                        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(held[(i >> 3) & 1]) : "v"(voff + so), "s"(wsrc) : "memory");
                    }
                }
            }
        }
        if (FORM >= 1 && FORM <= 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        if (FORM == 5 || FORM == 6)
            asm volatile("s_waitcnt vmcnt(8)" : "+v"(pin[0]), "+v"(pin[1]), "+v"(pin[2]), "+v"(pin[3]), "+v"(pin[4]), "+v"(pin[5]), "+v"(pin[6]), "+v"(pin[7]) : : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    long long t1 = __builtin_readcyclecounter();
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][3];
    if (FORM >= 4) s += __builtin_bit_cast(float, held[0][0] ^ held[1][1] ^ pin[0][0] ^ pin[7][3]) * 0.0f;
    __syncthreads();
    // the ring now holds the last iteration's eight 1 KiB loads of every wave (forms 1-3): sum of its words
    unsigned cs = 0;
    for (int i = lane; i < 2048; i += 64) cs += reinterpret_cast<const unsigned*>(ring + wave * 8192)[i];
    for (int o = 32; o > 0; o >>= 1) cs += __shfl_xor(cs, o);
    sink[blockIdx.x * 256 + threadIdx.x] = s + pad[0];
    if (lane == 0) {
        cyc[blockIdx.x * 4 + wave] = t1 - t0;
        check[blockIdx.x * 4 + wave] = cs;
    }
}

template <int FORM, int EVERY = 2>
void run(const char* name, const _Float16* rnd, const char* src, const std::vector<unsigned>& hsrc, long long* cyc, float* sink, unsigned* check) {
    const int grid = 256, iters = 2000, launches = 20;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<FORM, EVERY>), dim3(grid), dim3(256), 0, 0, rnd, src, cyc, sink, check, iters);
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < launches; ++r) hipLaunchKernelGGL((k<FORM, EVERY>), dim3(grid), dim3(256), 0, 0, rnd, src, cyc, sink, check, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.0f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> c(grid * 4);
    std::vector<unsigned> ch(grid * 4);
    (void)hipMemcpy(c.data(), cyc, c.size() * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(ch.data(), check, ch.size() * 4, hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto v : c) mean += double(v);
    mean /= c.size();
    // expected ring content: the loads of the last iteration (it = iters - 1): source offset ((it & 7) * 8192 + j * 1024) of the wave's run
    int bad = 0;
    if (FORM >= 1 && FORM <= 3 && EVERY == 2) {
        for (int bl = 0; bl < grid; ++bl)
            for (int w = 0; w < 4; ++w) {
                const size_t base = (static_cast<size_t>(bl % 16) * 4 + w) * 65536 + ((iters - 1) & 7) * 8192;
                unsigned cs = 0;
                for (int i = 0; i < 2048; ++i) cs += hsrc[base / 4 + i];
                bad += cs != ch[bl * 4 + w];
            }
    }
    const double flop = double(launches) * grid * 4 * iters * 16.0 * 32768.0;
    printf("%-62s %7.2f cycles per fragment  %7.1f TFLOP/s  clock %.2f GHz  %s\n", name, mean / (iters * 16.0), flop / (ms * 1e-3) / 1e12,
           mean * launches / (ms * 1e-3) / 1e9, (FORM >= 1 && FORM <= 3 && EVERY == 2) ? (bad ? "DATA MISMATCH" : "data ok") : "");
}

int main() {
    std::vector<_Float16> h(1 << 20);
    srand(1);
    for (auto& v : h) {
        float u = 0.0f;
        for (int i = 0; i < 12; ++i) u += float(rand()) / RAND_MAX;
        v = (_Float16)((u - 6.0f) * 0.5f);
    }
    std::vector<unsigned> hsrc(64 * 65536 / 4);
    for (auto& v : hsrc) v = static_cast<unsigned>(rand());
    _Float16* rnd;
    char* src;
    long long* cyc;
    float* sink;
    unsigned* check;
    (void)hipMalloc(&rnd, h.size() * 2);
    (void)hipMalloc(&src, hsrc.size() * 4);
    (void)hipMalloc(&cyc, 256 * 4 * 8);
    (void)hipMalloc(&check, 256 * 4 * 4);
    (void)hipMalloc(&sink, 256 * 256 * 4);
    (void)hipMemcpy(rnd, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(src, hsrc.data(), hsrc.size() * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("0 bare loop (2 MFMA 16x16x32 + ds_read_b128 per fragment)", rnd, src, hsrc, cyc, sink, check);
        run<1>("1 + global_load_lds_dwordx4, per-lane offset, every 2nd fragment", rnd, src, hsrc, cyc, sink, check);
        run<2>("2 + buffer_load_dwordx4 offen lds", rnd, src, hsrc, cyc, sink, check);
        run<3>("3 + buffer_load_dwordx4 off lds, ADD_TID_ENABLE descriptor", rnd, src, hsrc, cyc, sink, check);
        run<4>("4 + global_load_dwordx4 to registers + ds_write_b128 (1/4 rate)", rnd, src, hsrc, cyc, sink, check);
        run<1, 4>("1 every 4th fragment (the post-attention kernel's rate)", rnd, src, hsrc, cyc, sink, check);
        run<1, 8>("1 every 8th fragment", rnd, src, hsrc, cyc, sink, check);
        run<3, 4>("3 every 4th fragment", rnd, src, hsrc, cyc, sink, check);
        run<5, 2>("5 + global_load_dwordx4 into registers only, every 2nd fragment", rnd, src, hsrc, cyc, sink, check);
        run<5, 4>("5 every 4th fragment", rnd, src, hsrc, cyc, sink, check);
        run<6, 2>("6 + ds_write_b128 only, every 2nd fragment", rnd, src, hsrc, cyc, sink, check);
        run<6, 4>("6 every 4th fragment", rnd, src, hsrc, cyc, sink, check);
    }
    return 0;
}
