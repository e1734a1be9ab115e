// Which fp16 MFMA shape does the chip run faster under sustained load?  Same FLOPs, same LDS bytes per FLOP, random
// operands, one wave per SIMD on every CU:
//   S32: v_mfma_f32_32x32x16_f16, one fresh 1 KiB weight fragment from LDS (ds_read_b128) per MFMA, 16 accumulators
//   S16: v_mfma_f32_16x16x32_f16, one fresh 1 KiB weight fragment per PAIR of MFMAs (two 16-row halves of the same
//        32 rows), 32 accumulators
// Prints TFLOP/s by wall time (HIP events over a few hundred ms of back-to-back launches), cycles per FLOP from the
// shader clock and the effective clock (cycles / wall time).  The power-limited chip lowers its clock under load, and by
// how much depends on the instruction mix, so cycles alone do not rank the two.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_shape.hip -o tools/micro/mfma_shape && tools/micro/mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

constexpr int kLdsHalf = 32768;  // 64 KiB of random fp16 per workgroup

template <int SHAPE, int ZERO>
__global__ __launch_bounds__(256) void k(const _Float16* __restrict__ rnd, long long* cyc, float* sink, int iters) {
    __shared__ __attribute__((aligned(16))) _Float16 lds[kLdsHalf];
    __shared__ float pad[16 * 1024];  // 128 KiB in all: one workgroup per CU
    for (int i = threadIdx.x; i < kLdsHalf; i += 256) lds[i] = ZERO ? (_Float16)0.0f : rnd[(blockIdx.x * 977 + i) % (1 << 20)];
    if (threadIdx.x == 0) pad[0] = 0.0f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    h8 b[8], b2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            b[j][e] = ZERO ? (_Float16)0.0f : rnd[(threadIdx.x * 64 + j * 8 + e + 12345) % (1 << 20)];
            b2[j][e] = ZERO ? (_Float16)0.0f : rnd[(threadIdx.x * 64 + j * 8 + e + 54321) % (1 << 20)];
        }
    __syncthreads();
    const char* base0 = reinterpret_cast<const char*>(lds) + wave * 16384;
    long long t0 = 0, t1 = 0;
    float s = 0.0f;
    if constexpr (SHAPE == 32) {
        f16v acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.0f;
        t0 = __builtin_readcyclecounter();
        h8 wf[8];  // fragments are read eight MFMAs ahead of their use (hand-placed: the compiler would regroup them)
        const int a0 = static_cast<int>(reinterpret_cast<uintptr_t>(base0)) + lane * 16;
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wf[i]) : "v"(a0), "n"(i * 1024));
        for (int it = 0; it < iters; ++it) {
            const int ad = static_cast<int>(reinterpret_cast<uintptr_t>(base0)) + ((lane + it) & 63) * 16;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                asm volatile("s_waitcnt lgkmcnt(7)\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(wf[i & 7]), "v"(b[i & 7]));
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wf[i & 7]) : "v"(ad), "n"(((i + 8) & 15) * 1024));
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        t1 = __builtin_readcyclecounter();
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][9];
    } else {
        f4v acc[32];
#pragma unroll
        for (int i = 0; i < 32; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][e] = 0.0f;
        t0 = __builtin_readcyclecounter();
        h8 wf[8];
        const int a0 = static_cast<int>(reinterpret_cast<uintptr_t>(base0)) + lane * 16;
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wf[i]) : "v"(a0), "n"(i * 1024));
        for (int it = 0; it < iters; ++it) {
            const int ad = static_cast<int>(reinterpret_cast<uintptr_t>(base0)) + ((lane + it) & 63) * 16;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                asm volatile("s_waitcnt lgkmcnt(7)\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[2 * i]) : "v"(wf[i & 7]), "v"(b[i & 7]));
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[2 * i + 1]) : "v"(wf[i & 7]), "v"(b2[i & 7]));
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wf[i & 7]) : "v"(ad), "n"(((i + 8) & 15) * 1024));
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        t1 = __builtin_readcyclecounter();
#pragma unroll
        for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][3];
    }
    sink[blockIdx.x * 256 + threadIdx.x] = s + pad[0];
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int SHAPE, int ZERO>
void run(const char* name, const _Float16* rnd, long long* cyc, float* sink) {
    const int grid = 256, iters = 4000, launches = 40;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL((k<SHAPE, ZERO>), dim3(grid), dim3(256), 0, 0, rnd, cyc, sink, iters);
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < launches; ++r) hipLaunchKernelGGL((k<SHAPE, ZERO>), dim3(grid), dim3(256), 0, 0, rnd, cyc, sink, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.0f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> c(grid * 4);
    (void)hipMemcpy(c.data(), cyc, c.size() * 8, hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto v : c) mean += double(v);
    mean /= c.size();
    const double flop = double(launches) * grid * 4 * iters * 16.0 * 32768.0;  // 16 x 32 KFLOP per wave and iteration
    const double sec = ms * 1e-3;
    printf("%-44s %8.1f TFLOP/s  %7.2f cycles per 32 KFLOP  clock %.2f GHz  (%.1f ms)\n", name, flop / sec / 1e12,
           mean / (iters * 16.0), mean * launches / sec / 1e9, ms);
}

int main() {
    std::vector<_Float16> h(1 << 20);
    srand(1);
    for (auto& v : h) {
        float u = 0.0f;
        for (int i = 0; i < 12; ++i) u += float(rand()) / RAND_MAX;
        v = (_Float16)((u - 6.0f) * 0.5f);
    }
    _Float16* rnd;
    long long* cyc;
    float* sink;
    (void)hipMalloc(&rnd, h.size() * 2);
    (void)hipMalloc(&cyc, 256 * 4 * 8);
    (void)hipMalloc(&sink, 256 * 256 * 4);
    (void)hipMemcpy(rnd, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<32, 0>("32x32x16 f16, random operands, LDS reads", rnd, cyc, sink);
        run<16, 0>("16x16x32 f16, random operands, LDS reads", rnd, cyc, sink);
    }
    run<32, 1>("32x32x16 f16, zero operands", rnd, cyc, sink);
    run<16, 1>("16x16x32 f16, zero operands", rnd, cyc, sink);
    return 0;
}
