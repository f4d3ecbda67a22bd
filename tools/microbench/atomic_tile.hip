// Microbenchmark (round 5): what do the visibility scan's 64-bit atomics cost?  One wave = one 8 x 8 pixel tile per step (64 lanes, one atomicMax of 8 bytes each) on a
// 3840 x 2160 surface of 8-byte words, 4 workgroups of 4 waves per CU as the scan runs them, ~800 K tile steps in all (the 4K Sponza-stat frame's count).
//   layout   row-major (a tile = 8 half cache lines in 8 different lines) | tiled (a tile = 512 contiguous bytes = 4 whole lines)
//   place    any tile of the surface | tiles of the wave's own XCD only (surface cut into 8 interleaved sets of 64 x 64 pixel blocks, workgroup index modulo 8)
//   op       atomicMax, agent scope | the same, workgroup scope (diagnostic: not coherent across XCDs) | plain 8-byte store | nothing (the loop alone)
//   values   growing (every atomic writes) | constant after the first pass (every atomic loses: read-only in effect)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__device__ __forceinline__ uint32_t mix(uint32_t h) { h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16; return h; }
static constexpr uint32_t W = 3840, H = 2160, TX = W / 8, TY = H / 8;

template <int LAYOUT, int PLACE, int OP>
__global__ __launch_bounds__(256, 4) void atomic_kernel(unsigned long long* __restrict__ surf, int steps, uint32_t seed, int grow) {
    const uint32_t lane = threadIdx.x & 63u, wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    uint32_t state = mix(wave + seed);
    for (int s = 0; s < steps; ++s) {
        state = mix(state + 0x9e3779b9u);
        uint32_t tx = state % TX, ty = (state >> 12) % TY;
        if (PLACE == 1) {                       // 64 x 64 pixel blocks (8 x 8 tiles), block (bx, by) belongs to XCD (bx + by) & 7
            const uint32_t xcd = blockIdx.x & 7u, by = ty >> 3;
            uint32_t bx = tx >> 3;
            bx = (bx & ~7u) | ((xcd - by) & 7u);
            if (bx >= TX / 8) bx -= 8;
            tx = bx * 8 + (tx & 7u);
        }
        size_t idx;
        if (LAYOUT == 0) idx = (size_t)(ty * 8 + (lane >> 3)) * W + tx * 8 + (lane & 7u);
        else idx = ((size_t)ty * TX + tx) * 64 + lane;
        const unsigned long long v = grow ? ((unsigned long long)(seed + (uint32_t)s) << 32) | wave : 1ull;
        if (OP == 0) atomicMax(surf + idx, v);
        else if (OP == 1) __hip_atomic_fetch_max(surf + idx, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (OP == 2) surf[idx] = v;
        else if (v == 0x123456789ull) surf[idx] = v;
    }
}

template <int LAYOUT, int PLACE, int OP> void run(unsigned long long* surf, int cus, int grow, const char* what) {
    const int blocks = cus * 4, steps = 800000 / (blocks * 4) + 1;
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    CHECK(hipMemset(surf, 0, (size_t)W * H * 8));
    hipLaunchKernelGGL((atomic_kernel<LAYOUT, PLACE, OP>), dim3(blocks), dim3(256), 0, 0, surf, steps, 1u, 1);       // warm: every later value of a constant run loses
    CHECK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL((atomic_kernel<LAYOUT, PLACE, OP>), dim3(blocks), dim3(256), 0, 0, surf, steps, 100u + (uint32_t)rep * 1000u, grow);
        CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b)); best = ms < best ? ms : best;
    }
    const double tiles = (double)blocks * 4 * steps;
    printf("%-64s %.3f ms for %.0f K tile steps (%.1f M lane operations): %.2f ns per tile step over the chip\n", what, best, tiles / 1e3, tiles * 64 / 1e6, best * 1e6 / tiles);
    fflush(stdout);
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    unsigned long long* surf; CHECK(hipMalloc(&surf, (size_t)W * H * 8 + 4096));
    printf("%d CUs, surface %u x %u x 8 bytes = %.1f MB\n", cus, W, H, W * H * 8 / 1e6);
    run<0, 0, 3>(surf, cus, 1, "the loop alone");
    run<0, 0, 0>(surf, cus, 1, "row-major, any tile, atomicMax (agent), growing values");
    run<0, 0, 0>(surf, cus, 0, "row-major, any tile, atomicMax (agent), every atomic loses");
    run<1, 0, 0>(surf, cus, 1, "tiled, any tile, atomicMax (agent), growing values");
    run<1, 0, 0>(surf, cus, 0, "tiled, any tile, atomicMax (agent), every atomic loses");
    run<0, 1, 0>(surf, cus, 1, "row-major, own XCD's blocks, atomicMax (agent), growing");
    run<1, 1, 0>(surf, cus, 1, "tiled, own XCD's blocks, atomicMax (agent), growing");
    run<0, 0, 1>(surf, cus, 1, "row-major, any tile, atomicMax (workgroup scope), growing");
    run<1, 0, 1>(surf, cus, 1, "tiled, any tile, atomicMax (workgroup scope), growing");
    run<1, 1, 1>(surf, cus, 1, "tiled, own XCD's blocks, atomicMax (workgroup scope), growing");
    run<0, 0, 2>(surf, cus, 1, "row-major, any tile, plain store");
    run<1, 0, 2>(surf, cus, 1, "tiled, any tile, plain store");
    CHECK(hipFree(surf));
    return 0;
}
