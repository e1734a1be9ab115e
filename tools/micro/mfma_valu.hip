// Can the matrix pipe and the vector ALU of one gfx950 SIMD work at the same time?
//   (a) one wave: MFMAs back to back, then with K independent v_fma_f32 between consecutive MFMAs;
//   (b) two waves on one SIMD: one issues only MFMAs, the other only v_fma_f32 / v_exp_f32, alone and together.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/mfma_valu.hip -o tools/micro/mfma_valu && tools/micro/mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

// KIND of filler between MFMAs: 0 v_fma_f32, 1 s_waitcnt that is already satisfied, 2 s_add_u32, 3 s_nop 0,
// 4 ds_read_b128 (one wave-wide LDS read, results never waited for inside the block)
template <int K, int KIND = 0>
__device__ __forceinline__ void mfma_block(f16v (&acc)[4], h8 a, h8 b, float (&v)[8], float m, float c) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[i & 3]) : "v"(a), "v"(b));
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(i * K + k) & 7]) : "v"(m), "v"(c));
            if constexpr (KIND == 1) asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory");
            if constexpr (KIND == 2) asm volatile("s_add_u32 s20, s20, 1" ::: "s20", "scc");
            if constexpr (KIND == 3) asm volatile("s_nop 0");
            if constexpr (KIND == 4) {
                float4 t;
                asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"(static_cast<int>(threadIdx.x & 63) * 16));
            }
        }
    }
    if constexpr (KIND == 4) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// role: 0 = MFMA with K fillers, 1 = VALU fma only, 2 = exp only, 3 = idle
template <int K, int KIND = 0>
__global__ void k(long long* cyc, float* sink, int iters, int role_lo, int role_hi) {
    __shared__ float lds_pad[1024];
    if (threadIdx.x < 1024) lds_pad[threadIdx.x] = 0.f;
    const int wave = threadIdx.x >> 6;
    const int role = wave < 4 ? role_lo : role_hi;
    f16v acc[4];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    h8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.001f * threadIdx.x); b[e] = (_Float16)0.5f; }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.01f + i;
    const float m = 0.999f, c = 0.001f;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    if (role == 0) {
        for (int it = 0; it < iters; ++it) mfma_block<K, KIND>(acc, a, b, v, m, c);
    } else if (role == 1) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 128; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i & 7]) : "v"(m), "v"(c));
    } else if (role == 2) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i & 7]));
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
    for (int i = 0; i < 8; ++i) s += v[i];
    sink[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
}

template <int K, int KIND = 0>
void run(const char* name, int threads, int role_lo, int role_hi) {
    long long* d;
    float* sink;
    (void)hipMalloc(&d, 64);
    (void)hipMalloc(&sink, 4 * 512);
    const int iters = 500;
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<K, KIND>), dim3(1), dim3(threads), 0, 0, d, sink, iters, role_lo, role_hi);
    (void)hipDeviceSynchronize();
    long long c[8] = {0};
    (void)hipMemcpy(c, d, 64, hipMemcpyDeviceToHost);
    printf("%-58s wave0 %7.1f cycles/iter", name, double(c[0]) / iters);
    if (threads > 256) printf(", wave4 %7.1f cycles/iter", double(c[4]) / iters);
    printf("\n");
    (void)hipFree(d);
    (void)hipFree(sink);
}

int main() {
    printf("one iteration = 16 MFMA 32x32x16 f16 (+16 K v_fma) | 128 v_fma | 32 v_exp\n");
    run<0>("1 wave/SIMD: MFMA only", 256, 0, 3);
    run<1>("1 wave/SIMD: MFMA + 1 v_fma each", 256, 0, 3);
    run<2>("1 wave/SIMD: MFMA + 2 v_fma each", 256, 0, 3);
    run<4>("1 wave/SIMD: MFMA + 4 v_fma each", 256, 0, 3);
    run<6>("1 wave/SIMD: MFMA + 6 v_fma each", 256, 0, 3);
    run<8>("1 wave/SIMD: MFMA + 8 v_fma each", 256, 0, 3);
    run<4, 1>("1 wave/SIMD: MFMA + 4 satisfied s_waitcnt each", 256, 0, 3);
    run<8, 1>("1 wave/SIMD: MFMA + 8 satisfied s_waitcnt each", 256, 0, 3);
    run<16, 1>("1 wave/SIMD: MFMA + 16 satisfied s_waitcnt each", 256, 0, 3);
    run<8, 2>("1 wave/SIMD: MFMA + 8 s_add_u32 each", 256, 0, 3);
    run<16, 2>("1 wave/SIMD: MFMA + 16 s_add_u32 each", 256, 0, 3);
    run<16, 3>("1 wave/SIMD: MFMA + 16 s_nop 0 each", 256, 0, 3);
    run<1, 4>("1 wave/SIMD: MFMA + 1 ds_read_b128 each", 256, 0, 3);
    run<2, 4>("1 wave/SIMD: MFMA + 2 ds_read_b128 each", 256, 0, 3);
    run<4, 4>("1 wave/SIMD: MFMA + 4 ds_read_b128 each", 256, 0, 3);
    run<0>("1 wave/SIMD: 128 v_fma only", 256, 1, 3);
    run<0>("1 wave/SIMD: 32 v_exp only", 256, 2, 3);
    run<0>("2 waves/SIMD: MFMA | MFMA", 512, 0, 0);
    run<0>("2 waves/SIMD: MFMA | 128 v_fma", 512, 0, 1);
    run<0>("2 waves/SIMD: MFMA | 32 v_exp", 512, 0, 2);
    run<0>("2 waves/SIMD: 128 v_fma | 128 v_fma", 512, 1, 1);
    run<0>("2 waves/SIMD: 128 v_fma | 32 v_exp", 512, 1, 2);
    return 0;
}
