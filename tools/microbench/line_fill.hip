// Microbenchmark (round 5): the traversal loop's fetches are bound by vector-L1 LINE FILLS (tools/microbench/coop_fetch.hip: gathering the same records cooperatively,
// one cache line touched once, costs the same as the per-lane gathers -- so it is the bytes that arrive, not the tag look-ups).  How many bytes does a record cost?
// Records of RS bytes at a stride of ST bytes, random indices, A of 64 lanes active, every fetch dependent on the previous one; 7 workgroups of 4 waves per CU.
//   80 @ 80   today's node: half of them straddle two 128-B lines          80 @ 128  one line each
//   64 @ 64 / 64 @ 128: is a fill a 128-B line or a 64-B sector?            48 @ 48 / 48 @ 64 / 96 @ 48 (two neighbouring triangle records in one step)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t mix(uint32_t h) { h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16; return h; }

template <int PIECES, int STRIDE>
__global__ __launch_bounds__(256, 7) void fill_kernel(const char* __restrict__ table, uint32_t numRecords, int steps, uint32_t pActive, uint32_t* __restrict__ out) {
    uint32_t state = mix(blockIdx.x * 256u + threadIdx.x + 1u), acc = 0;
    for (int s = 0; s < steps; ++s) {
        const bool want = (mix(state + (uint32_t)s) & 63u) < pActive;
        const uint32_t off = (mix(state ^ 0x9e3779b9u) % numRecords) * (uint32_t)STRIDE;
        uint32_t h = 0;
        if (want) {
            const v4f* p = (const v4f*)(table + off);
            v4f v[PIECES];
#pragma unroll
            for (int k = 0; k < PIECES; ++k) v[k] = p[k];
#pragma unroll
            for (int k = 0; k < PIECES; ++k) h ^= __float_as_uint(v[k].x) + __float_as_uint(v[k].w);
        }
        acc += h; state = mix(state + h);
    }
    out[blockIdx.x * 256u + threadIdx.x] = acc;
}

template <int PIECES, int STRIDE> void run(const char* table, size_t tableBytes, int cus, double clk, uint32_t pActive, uint32_t* out, const char* what) {
    const uint32_t numRecords = (uint32_t)(tableBytes / STRIDE) - 2u;
    const int blocks = cus * 7, steps = 2000;
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL((fill_kernel<PIECES, STRIDE>), dim3(blocks), dim3(256), 0, 0, table, numRecords, 8, pActive, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL((fill_kernel<PIECES, STRIDE>), dim3(blocks), dim3(256), 0, 0, table, numRecords, steps, pActive, out);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    const double iters = (double)blocks * 4 * steps / cus, cyc = ms * 1e-3 * clk / iters;
    printf("  %3d B @ %3d B, table %5.1f MB, %2u lanes: %7.3f ms  %6.1f cycles/iteration/CU  %5.2f cycles per record  %5.1f record-B/clk/CU   %s\n", PIECES * 16, STRIDE, tableBytes / 1048576.0, pActive, ms, cyc,
           cyc / pActive, PIECES * 16.0 * pActive / cyc, what);
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const size_t bytes = 64ull << 20;
    char* table; uint32_t* out;
    CHECK(hipMalloc(&table, bytes + 4096));
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount; const double clk = prop.clockRate * 1e3;
    CHECK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
    { std::vector<uint32_t> h(bytes / 4); uint32_t x = 1; for (auto& v : h) { x = x * 1664525u + 1013904223u; v = x; } CHECK(hipMemcpy(table, h.data(), bytes, hipMemcpyHostToDevice)); }
    const size_t nodeB = 26000ull * 80, nodeB128 = 26000ull * 128, nodeB64 = 26000ull * 64, triB = 262267ull * 48, triB64 = 262267ull * 64;
    printf("nodes (tree of Sponza-stat: 26 k records):\n");
    run<5, 80>(table, nodeB, cus, clk, 44, out, "today's node");
    run<5, 128>(table, nodeB128, cus, clk, 44, out, "node padded to a line");
    run<4, 64>(table, nodeB64, cus, clk, 44, out, "a 64-B node, two per line");
    run<4, 128>(table, nodeB128, cus, clk, 44, out, "a 64-B node alone in its line");
    run<5, 80>(table, nodeB, cus, clk, 26, out, "today's node, 26 lanes (the rest out of LDS)");
    run<5, 128>(table, nodeB128, cus, clk, 26, out, "padded, 26 lanes");
    run<5, 80>(table, nodeB, cus, clk, 64, out, "today's node, all lanes");
    run<5, 128>(table, nodeB128, cus, clk, 64, out, "padded, all lanes");
    run<8, 128>(table, nodeB128, cus, clk, 44, out, "a whole 128-B line per lane");
    printf("triangles (262 k records):\n");
    run<3, 48>(table, triB, cus, clk, 19, out, "today's triangle record");
    run<3, 64>(table, triB64, cus, clk, 19, out, "padded to 64 B");
    run<6, 48>(table, triB, cus, clk, 19, out, "two neighbouring records per step");
    run<6, 48>(table, triB, cus, clk, 10, out, "two neighbouring records, half the lanes");
    run<3, 48>(table, triB, cus, clk, 64, out, "all lanes");
    run<3, 64>(table, triB64, cus, clk, 64, out, "all lanes, padded");
    run<3, 48>(table, 2ull << 20, cus, clk, 19, out, "today's record, a 2-MB table (L2 hits)");
    run<3, 64>(table, 2ull << 20, cus, clk, 19, out, "padded, a 2-MB table");
    return 0;
}
