// Issue rate of fp32 VALU instructions on one SIMD of gfx950: scalar v_fma_f32 against the packed
// v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32, with 1, 2 and 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/valu_rate.hip -o build/valu_rate && build/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float v2f __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void rate_kernel(long long* cyc, float* sink, int iters) {
    v2f a[8];
    for (int i = 0; i < 8; ++i) a[i] = v2f{threadIdx.x * 0.001f + i, 1.0f + i};
    const v2f m = v2f{1.0001f, 0.9999f}, c = v2f{0.001f, -0.001f};
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(m.x), "v"(c.x));
                if constexpr (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                if constexpr (KIND == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if constexpr (KIND == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if constexpr (KIND == 4) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(c.x));
            }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int KIND>
void run(const char* name) {
    long long* d;
    float* sink;
    hipMalloc(&d, 8);
    hipMalloc(&sink, 4 * 1024 * 16);
    const int iters = 2000;
    for (int waves : {1, 2, 4}) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(rate_kernel<KIND>, dim3(1), dim3(256 * waves), 0, 0, d, sink, iters);
        hipEventRecord(e0);
        hipLaunchKernelGGL(rate_kernel<KIND>, dim3(1), dim3(256 * waves), 0, 0, d, sink, iters);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        long long c;
        hipMemcpy(&c, d, 8, hipMemcpyDeviceToHost);
        const double n = 32.0 * iters * waves;  // instructions per SIMD
        printf("%-14s %d wave/SIMD: %8.3f us, %6.2f ns per instr per SIMD, counter ticks/instr %.3f\n", name, waves,
               ms * 1e3, ms * 1e6 / n, c / n);
    }
}

int main() {
    run<0>("v_fma_f32");
    run<4>("v_add_f32");
    run<1>("v_pk_fma_f32");
    run<2>("v_pk_add_f32");
    run<3>("v_pk_mul_f32");
    return 0;
}
