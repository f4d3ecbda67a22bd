// Microbenchmark 2 (round 2): what does a divergent dwordx4 gather cost the vector L1 -- per ACTIVE LANE or per wave
// instruction?  And what does the same gather cost from LDS?  Decides (a) whether idle lanes of the traversal loop are
// free on the memory side, (b) whether caching the top of the BVH in LDS can pay.
//   mode 0: all 64 lanes gather 80-B records (5 x dwordx4)            mode 1: lanes 0..31 only
//   mode 2: even lanes only                                           mode 3: lanes 0..15 only
//   mode 4: all lanes, records in LDS (5 x ds_read_b128, 512 records = 40 KB per workgroup)
//   mode 5: all lanes, 64-B records (4 x dwordx4)                     mode 6: all lanes, 48-B records
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int NLOADS, int MODE>
__global__ __launch_bounds__(256) void gather_kernel(const float4* __restrict__ table, uint32_t numRecords, int steps, uint32_t* __restrict__ out) {
    __shared__ float4 s_tab[MODE == 4 ? 512 * 5 : 1];
    if (MODE == 4) { for (int i = threadIdx.x; i < 512 * 5; i += 256) s_tab[i] = table[i]; __syncthreads(); numRecords = 512; }
    const uint32_t lane = threadIdx.x & 63u;
    const bool on = MODE == 1 ? lane < 32u : MODE == 2 ? (lane & 1u) == 0u : MODE == 3 ? lane < 16u : true;
    uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u % numRecords;
    uint32_t acc = 0;
    if (on) {
        for (int s = 0; s < steps; ++s) {
            const float4* p = (MODE == 4 ? s_tab : table) + (size_t)idx * NLOADS;
            float4 v[NLOADS];
#pragma unroll
            for (int k = 0; k < NLOADS; ++k) v[k] = p[k];
            uint32_t h = 0;
#pragma unroll
            for (int k = 0; k < NLOADS; ++k) h ^= __float_as_uint(v[k].x) + __float_as_uint(v[k].w);
            acc += h;
            idx = (h * 2654435761u + idx * 40503u + 12345u) % numRecords;
        }
    }
    out[blockIdx.x * 256u + threadIdx.x] = acc;
}

template <int NLOADS, int MODE> double run(const float4* table, uint32_t numRecords, int blocks, int steps, uint32_t* out, int activePer64) {
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL((gather_kernel<NLOADS, MODE>), dim3(blocks), dim3(256), 0, 0, table, numRecords, 8, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL((gather_kernel<NLOADS, MODE>), dim3(blocks), dim3(256), 0, 0, table, numRecords, steps, out);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    return (double)blocks * 256 * activePer64 / 64.0 * steps / (ms * 1e-3);
}

int main() {
    const size_t bytes = 64ull << 20;
    float4* table; uint32_t* out;
    CHECK(hipMalloc(&table, bytes)); CHECK(hipMalloc(&out, 256 * 8192 * 4));
    std::vector<uint32_t> h(bytes / 4);
    uint32_t x = 1; for (auto& v : h) { x = x * 1664525u + 1013904223u; v = x; }
    CHECK(hipMemcpy(table, h.data(), bytes, hipMemcpyHostToDevice));
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, blocks = cus * 8, steps = 256;
    for (size_t mb : {4, 16}) {
        const uint32_t rec80 = (uint32_t)((mb << 20) / 80), rec64 = (uint32_t)((mb << 20) / 64), rec48 = (uint32_t)((mb << 20) / 48);
        printf("table %zu MB, 32 waves/CU, G lane-records/s (active lanes only):\n", mb);
        printf("  80 B all lanes      %.1f\n", run<5, 0>(table, rec80, blocks, steps, out, 64) / 1e9);
        printf("  80 B lanes 0..31    %.1f\n", run<5, 1>(table, rec80, blocks, steps, out, 32) / 1e9);
        printf("  80 B even lanes     %.1f\n", run<5, 2>(table, rec80, blocks, steps, out, 32) / 1e9);
        printf("  80 B lanes 0..15    %.1f\n", run<5, 3>(table, rec80, blocks, steps, out, 16) / 1e9);
        printf("  64 B all lanes      %.1f\n", run<4, 5>(table, rec64, blocks, steps, out, 64) / 1e9);
        printf("  48 B all lanes      %.1f\n", run<3, 6>(table, rec48, blocks, steps, out, 64) / 1e9);
    }
    printf("LDS 80 B all lanes (512 records per workgroup, 8 workgroups/CU would need 320 KB: run with 3/CU)  %.1f\n",
           run<5, 4>(table, 512, cus * 3, steps, out, 64) / 1e9);
    return 0;
}
