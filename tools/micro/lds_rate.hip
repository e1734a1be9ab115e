// LDS cycles per wave-wide read for address patterns of the fbank kernel: 16 waves of one workgroup issue the same
// pattern back to back, so the figure is the LDS pipeline's throughput, not its latency.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/lds_rate.hip -o tools/micro/lds_rate && tools/micro/lds_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <functional>

template <int W>  // bytes per lane: 4, 8, 16; 40 = ds_read2_b32 (second dword 1 word on)
__global__ void rate_kernel(const int* addr_bytes, long long* cyc, float* sink, int iters) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = i;
    __syncthreads();
    const int a = addr_bytes[threadIdx.x & 63];
    float acc = 0;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if constexpr (W == 4) asm volatile("ds_read_b32 %0, %1" : "=v"(v[u].x) : "v"(a));
            if constexpr (W == 8) asm volatile("ds_read_b64 %0, %1" : "=v"(*reinterpret_cast<float2*>(&v[u])) : "v"(a));
            if constexpr (W == 16) asm volatile("ds_read_b128 %0, %1" : "=v"(v[u]) : "v"(a));
            if constexpr (W == 40) asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(*reinterpret_cast<float2*>(&v[u])) : "v"(a));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u].x;
    }
    __syncthreads();
    const long long t1 = __builtin_readcyclecounter();
    sink[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x == 0) *cyc = t1 - t0;
}

int main() {
    int* d_addr;
    long long* d_cyc;
    float* sink;
    (void)hipMalloc(&d_addr, 64 * 4);
    (void)hipMalloc(&d_cyc, 8);
    (void)hipMalloc(&sink, 1024 * 4);
    const int iters = 2000;
    auto run = [&](const char* name, int w, std::function<int(int, int)> word_addr) {
        std::vector<int> a(64);
        for (int lane = 0; lane < 64; ++lane) a[lane] = 4 * word_addr(lane >> 4, lane & 15);
        (void)hipMemcpy(d_addr, a.data(), 256, hipMemcpyHostToDevice);
        for (int rep = 0; rep < 2; ++rep) {
            if (w == 4) hipLaunchKernelGGL(rate_kernel<4>, dim3(1), dim3(1024), 65536, 0, d_addr, d_cyc, sink, iters);
            if (w == 8) hipLaunchKernelGGL(rate_kernel<8>, dim3(1), dim3(1024), 65536, 0, d_addr, d_cyc, sink, iters);
            if (w == 16) hipLaunchKernelGGL(rate_kernel<16>, dim3(1), dim3(1024), 65536, 0, d_addr, d_cyc, sink, iters);
            if (w == 40) hipLaunchKernelGGL(rate_kernel<40>, dim3(1), dim3(1024), 65536, 0, d_addr, d_cyc, sink, iters);
            (void)hipDeviceSynchronize();
        }
        long long c;
        (void)hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost);
        printf("%-52s w=%2d: %6.2f cycles per read\n", name, w, double(c) / (8.0 * iters * 16));
    };
    run("b32 contiguous", 4, [](int f, int l) { return 16 * f + l; });
    run("b64 contiguous", 8, [](int f, int l) { return 2 * (16 * f + l); });
    run("b128 contiguous", 16, [](int f, int l) { return 4 * (16 * f + l); });
    run("b32 all lanes one address", 4, [](int f, int l) { return 5; });
    run("b32 stride 32 words (worst)", 4, [](int f, int l) { return 32 * (16 * f + l) % 8192; });
    for (int fs : {576, 592, 608, 584}) {
        char nm[96];
        snprintf(nm, 96, "frame stride %d: b32 f*FS + l", fs); run(nm, 4, [fs](int f, int l) { return f * fs + l; });
        snprintf(nm, 96, "frame stride %d: b64 f*FS + 2l (exchange write)", fs); run(nm, 8, [fs](int f, int l) { return f * fs + 2 * l; });
        snprintf(nm, 96, "frame stride %d: b64 f*FS + 2(255-l) (pair read)", fs); run(nm, 8, [fs](int f, int l) { return f * fs + 2 * (255 - l); });
        snprintf(nm, 96, "frame stride %d: b128 f*FS + 36 l (exchange read)", fs); run(nm, 16, [fs](int f, int l) { return f * fs + 36 * l; });
        snprintf(nm, 96, "frame stride %d: read2_b32 f*FS + 8 l (mel grp 4)", fs); run(nm, 40, [fs](int f, int l) { return f * fs + 8 * l + 1; });
        snprintf(nm, 96, "frame stride %d: read2_b32 f*FS + 2 l (mel grp 1)", fs); run(nm, 40, [fs](int f, int l) { return f * fs + 2 * l + 1; });
    }
    run("samples b64 160 f + 2 l", 8, [](int f, int l) { return 160 * f + 2 * l; });
    run("prev sample b32 160 f + 2 l + 1", 4, [](int f, int l) { return 160 * f + 2 * l + 1; });
    run("constants b64 2 l (same for 4 frames)", 8, [](int f, int l) { return 2 * l; });
    run("mel weights b128 4 l (same for 4 frames)", 16, [](int f, int l) { return 4 * l; });
    run("b128 rows of 36 words, 16 lanes (l*36)", 16, [](int f, int l) { return 36 * l + 4 * f; });
    return 0;
}
