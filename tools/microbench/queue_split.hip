// Microbenchmark 4 (round 4): the L2 work-queue split of the traversal that the round-3 review asked to "build or kill with a number".
// The proposal: instead of persistent waves that keep a ray in registers for its whole walk, a NODE kernel and a TRIANGLE kernel ping-pong
// over queues of 8-byte (ray, node) / (ray, triangle) items; every launch runs with all 64 lanes on one kind of step.  What a step then
// costs that it does not cost today: the item (8 B read), the RAY fetched again (32 B gather by ray index -- it no longer lives in
// registers), the node from L2 (no per-workgroup LDS copy of the top levels is possible: a wave's items are unrelated), and 8 B per
// surviving child written through a wave-aggregated atomic.  This kernel is the node pass in its most favourable form -- full waves,
// coalesced item reads, the real node test (pt_traverse.h node8_test), at most two children emitted per item, no stack / restart
// bookkeeping at all -- so its item rate is an UPPER bound for the split's step rate.  The persistent kernels' measured rate to beat:
// (726.7 M any-hit + 369.2 M closest-hit lane-steps per 4K frame, profiles/r03_lane_stats.txt) / (7.4 + 5.4 ms) = 86 G steps/s.
// MEASURED (profiles/r04_queue_split.txt): 33.6 G items/s with items in random ray order, 68 G items/s with the queue sorted by ray -- both below
// the 86 G steps/s of the persistent kernels, before the triangle pass, the per-ray stack in memory and any sorting are priced: not built.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -I../../incarnation-renderer_amd/csrc -o queue_split queue_split.hip && ./queue_split
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "pt_traverse.h"
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
using namespace incr;

__global__ __launch_bounds__(256) void node_pass(const uint2* __restrict__ itemsIn, uint32_t numItems, const float4* __restrict__ rays, const float4* __restrict__ nodes,
                                                 uint32_t numNodes, uint2* __restrict__ itemsOut, uint32_t* __restrict__ outCount, uint32_t outCap) {
    uint32_t wBase = 0u, wUsed = 256u;          // wave-uniform: the wave's current block of output slots
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < numItems; i += gridDim.x * 256u) {
        const uint2 it = itemsIn[i];
        const float4 ro = rays[(size_t)it.x * 2], rd = rays[(size_t)it.x * 2 + 1];
        const float4* __restrict__ np = nodes + (size_t)(it.y % numNodes) * 5;
        const float4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3], n4 = np[4];
        const RaySetup rs = ray_setup({ro.x, ro.y, ro.z}, {rd.x, rd.y, rd.z});
        Node8 N;
        node8_test(n0, n1, n2, n3, n4, rs, 0.1f, 1000.f, N);
        // emit up to two children (the traversal's average is ~1 new item per node step: nodes visited per ray / 1 root)
        uint32_t m = N.hitMask;
#pragma unroll
        for (int round = 0; round < 2; ++round) {
            const bool want = m != 0u;
            const unsigned long long b = __ballot(want);
            if (b == 0ull) break;
            const uint32_t lane = threadIdx.x & 63u, rank = (uint32_t)__popcll(b & ((1ull << lane) - 1ull)), n = (uint32_t)__popcll(b);
            // a wave reserves output space 256 slots at a time (one global atomic per 256 items; a first version with one atomic per wave and
            // round on a single counter ran at 7.7 G items/s: the same-address atomics alone took 4 ms)
            if (wUsed + n > 256u) {
                uint32_t nb = 0;
                if (lane == 0) nb = atomicAdd(outCount, 256u);
                wBase = __builtin_amdgcn_readfirstlane(nb); wUsed = 0u;
            }
            if (want) { const uint32_t k = (uint32_t)__builtin_ctz(m); m &= m - 1u; const uint32_t slot = wBase + wUsed + rank; if (slot < outCap) itemsOut[slot] = make_uint2(it.x, N.childBase + k); }
            wUsed += n;
        }
    }
}

int main() {
    const uint32_t numRays = 8u << 20, numNodes = 28065, numItems = 32u << 20;      // Sponza-stat's tree size; 32 M items per launch
    std::vector<float> hr((size_t)numRays * 8);
    uint32_t x = 1; auto rnd = [&]() { x = x * 1664525u + 1013904223u; return (float)(x >> 8) * (1.f / 16777216.f); };
    for (uint32_t r = 0; r < numRays; ++r) { float* p = &hr[(size_t)r * 8]; p[0] = rnd() * 20.f - 10.f; p[1] = rnd() * 8.f; p[2] = rnd() * 10.f - 5.f; p[3] = 0.f; p[4] = rnd() - 0.5f; p[5] = rnd() - 0.5f; p[6] = rnd() - 0.5f; p[7] = 0.f; }
    std::vector<uint32_t> hn((size_t)numNodes * 20);
    for (auto& v : hn) { x = x * 1664525u + 1013904223u; v = x; }
    for (uint32_t n = 0; n < numNodes; ++n) { float o[3] = {rnd() * 20.f - 10.f, rnd() * 8.f, rnd() * 10.f - 5.f}; memcpy(&hn[(size_t)n * 20], o, 12); hn[(size_t)n * 20 + 3] = 0x7c7c7cu | 0xffu << 24; hn[(size_t)n * 20 + 4] = (n * 8u + 1u) % numNodes; }
    std::vector<uint2> hi(numItems);
    for (uint32_t i = 0; i < numItems; ++i) { x = x * 1664525u + 1013904223u; hi[i] = make_uint2((x >> 4) % numRays, (x >> 9) % numNodes); }
    float4 *rays, *nodes; uint2 *in, *out; uint32_t* count;
    CHECK(hipMalloc(&rays, hr.size() * 4)); CHECK(hipMalloc(&nodes, hn.size() * 4)); CHECK(hipMalloc(&in, (size_t)numItems * 8)); CHECK(hipMalloc(&out, (size_t)numItems * 16)); CHECK(hipMalloc(&count, 4));
    CHECK(hipMemcpy(rays, hr.data(), hr.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(nodes, hn.data(), hn.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(in, hi.data(), (size_t)numItems * 8, hipMemcpyHostToDevice));
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int sorted = 0; sorted < 2; ++sorted) {
        if (sorted) {          // the friendliest case: items in ray order (a ray's bytes are fetched by neighbouring lanes) -- a queue the split could only get by sorting every pass
            for (uint32_t i = 0; i < numItems; ++i) hi[i].x = (uint32_t)((uint64_t)i * numRays / numItems);
            CHECK(hipMemcpy(in, hi.data(), (size_t)numItems * 8, hipMemcpyHostToDevice));
        }
        for (int blocksPerCU : {8, 16, 32}) {
            float best = 1e9f; uint32_t emitted = 0;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipMemset(count, 0, 4));
                CHECK(hipEventRecord(a));
                hipLaunchKernelGGL(node_pass, dim3(prop.multiProcessorCount * blocksPerCU), dim3(256), 0, 0, in, numItems, rays, nodes, numNodes, out, count, numItems * 2u);
                CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
                float ms; CHECK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
                CHECK(hipMemcpy(&emitted, count, 4, hipMemcpyDeviceToHost));
            }
            printf("node pass, %s ray order, %2d workgroups/CU: %u items in %.3f ms = %.1f G items/s (%.2f output slots reserved per item)\n", sorted ? "sorted" : "random", blocksPerCU, numItems, best,
                   numItems / (best * 1e-3) / 1e9, (double)emitted / numItems);
        }
    }
    return 0;
}
