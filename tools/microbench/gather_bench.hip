// Microbenchmark: dependent random gathers of 80-byte records (5 x dwordx4 per lane), the memory access pattern of the
// 8-wide BVH traversal, with almost no ALU work.  Reports lane-fetches per second for several table sizes,
// waves per CU and record sizes, to find the memory-system ceiling for this pattern on MI355X.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int NLOADS>
__global__ __launch_bounds__(256) void gather_kernel(const float4* __restrict__ table, uint32_t numRecords, uint32_t recF4, int steps, uint32_t* __restrict__ out) {
    uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u % numRecords;
    uint32_t acc = 0;
    for (int s = 0; s < steps; ++s) {
        const float4* p = table + (size_t)idx * recF4;
        float4 v[NLOADS];
#pragma unroll
        for (int k = 0; k < NLOADS; ++k) v[k] = p[k];
        uint32_t h = 0;
#pragma unroll
        for (int k = 0; k < NLOADS; ++k) h ^= __float_as_uint(v[k].x) + __float_as_uint(v[k].w);
        acc += h;
        idx = (h * 2654435761u + idx * 40503u + 12345u) % numRecords;      // next record depends on the data (dependent chain)
    }
    out[blockIdx.x * 256u + threadIdx.x] = acc;
}

template <int NLOADS> double run(const float4* table, uint32_t numRecords, uint32_t recF4, int blocks, int steps, uint32_t* out) {
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(gather_kernel<NLOADS>, dim3(blocks), dim3(256), 0, 0, table, numRecords, recF4, 8, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(gather_kernel<NLOADS>, dim3(blocks), dim3(256), 0, 0, table, numRecords, recF4, steps, out);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    return (double)blocks * 256 * steps / (ms * 1e-3);
}

int main() {
    const size_t bytes = 512ull << 20;
    float4* table; uint32_t* out;
    CHECK(hipMalloc(&table, bytes)); CHECK(hipMalloc(&out, 256 * 8192 * 4));
    std::vector<uint32_t> h(bytes / 4);
    uint32_t x = 1; for (auto& v : h) { x = x * 1664525u + 1013904223u; v = x; }
    CHECK(hipMemcpy(table, h.data(), bytes, hipMemcpyHostToDevice));
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("CUs %d clock %d MHz\n", cus, prop.clockRate / 1000);
    const int steps = 256;
    for (size_t mb : {4, 16, 64, 256}) {
        for (int wavesPerCU : {8, 16, 32}) {
            const int blocks = cus * wavesPerCU / 4;
            const uint32_t rec80 = (uint32_t)((mb << 20) / 80), rec128 = (uint32_t)((mb << 20) / 128), rec48 = (uint32_t)((mb << 20) / 48);
            double r5 = run<5>(table, rec80, 5, blocks, steps, out);
            double r8 = run<8>(table, rec128, 8, blocks, steps, out);
            double r3 = run<3>(table, rec48, 3, blocks, steps, out);
            double r1 = run<1>(table, (uint32_t)((mb << 20) / 16), 1, blocks, steps, out);
            printf("table %4zu MB  waves/CU %2d : 80B-rec %.2f G/s (%.1f cyc/CU/fetch)  128B-rec %.2f G/s  48B-rec %.2f G/s  16B-rec %.2f G/s\n",
                   mb, wavesPerCU, r5 / 1e9, (double)cus * prop.clockRate * 1e3 / r5, r8 / 1e9, r3 / 1e9, r1 / 1e9);
        }
    }
    return 0;
}
