// Microbenchmark (round 5): what does ONE ITERATION of the traversal loop's memory side cost, and does a COOPERATIVE fetch cut it?
//
// The persistent traversal waves (pathtrace_wf.hip wf_shadow_kernel / wf_extend_kernel) fetch, per iteration, an 80-B node record for the ~44
// lanes that take a node step (~26 once the hottest nodes come out of LDS) and a 48-B triangle record for the ~19 lanes that take a triangle
// step: 5 + 3 divergent dwordx4 gathers in which every lane's 16 bytes are a cache line of their own as far as the vector L1 can tell.
// tools/microbench/gather_lanes.hip priced such a gather at one L1 slot per active lane + ~22 per wave instruction.
//   mode 0  the loop's present form: per-lane 3 x dwordx4 (triangle) + 5 x dwordx4 (node)
//   mode 1  cooperative, LDS-DMA: the wave compacts its requests (ballot / mbcnt), lane l of round r fetches piece 64 r + l -- the 16-byte
//           pieces of one record go to neighbouring lanes, so a record is one or two cache lines touched once -- with global_load_lds_dwordx4
//           into a per-wave staging area; the owners then read their record with ds_read_b128
//   mode 2  the same through registers: global_load_dwordx4 + ds_write_b128 + ds_read_b128
// Every lane's next record index depends on what it loaded (a traversal is a chain of dependent fetches).
// Usage: coop_fetch [nodeLanes triLanes]    (expected active lanes per wave iteration; default sweeps 26/19, 44/19, 64/0)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4f* LdsV4;
typedef __attribute__((address_space(3))) uint32_t* LdsU32;
static constexpr int kMaxPieces = 288;          // 4.5 KB of staging per wave: 7 workgroups of 4 waves fit a CU beside their 10-KB stacks

__device__ __forceinline__ uint32_t mix(uint32_t h) { h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16; return h; }

template <int MODE, int WAVES>
__global__ __launch_bounds__(256, WAVES) void fetch_kernel(const char* __restrict__ nodes, uint32_t numNodes, const char* __restrict__ tris, uint32_t numTris,
                                                           int steps, uint32_t pNode, uint32_t pTri, uint32_t* __restrict__ out, unsigned long long* __restrict__ fetched) {
    __shared__ v4f s_stage[MODE ? 4 * kMaxPieces : 1];
    __shared__ uint32_t s_offs[MODE ? 4 * 128 : 1];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t state = mix(blockIdx.x * 256u + threadIdx.x + 1u);
    uint32_t acc = 0, nFetched = 0;
    LdsV4 stage = (LdsV4)s_stage + wave * kMaxPieces;
    LdsU32 offs = (LdsU32)s_offs + wave * 128;
    for (int s = 0; s < steps; ++s) {
        const uint32_t r0 = mix(state + (uint32_t)s);
        const bool wantNode = (r0 & 63u) < pNode, wantTri = ((r0 >> 6) & 63u) < pTri;
        const uint32_t nodeOff = (mix(state ^ 0x9e3779b9u) % numNodes) * 80u, triOff = (mix(state ^ 0x85ebca6bu) % numTris) * 48u;
        v4f n0 = {0, 0, 0, 0}, n1 = n0, n2 = n0, n3 = n0, n4 = n0, t0 = n0, t1 = n0, t2 = n0;
        bool gotTri = wantTri, gotNode = wantNode;
        if (MODE == 0) {
            if (wantTri) { const v4f* p = (const v4f*)(tris + triOff); t0 = p[0]; t1 = p[1]; t2 = p[2]; }
            if (wantNode) { const v4f* p = (const v4f*)(nodes + nodeOff); n0 = p[0]; n1 = p[1]; n2 = p[2]; n3 = p[3]; n4 = p[4]; }
        } else {
            const unsigned long long bn = __ballot(wantNode), bt = __ballot(wantTri);
            const uint32_t nN = (uint32_t)__popcll(bn), nT = (uint32_t)__popcll(bt);
            const uint32_t rankN = __builtin_amdgcn_mbcnt_hi((uint32_t)(bn >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bn, 0u));
            const uint32_t rankT = __builtin_amdgcn_mbcnt_hi((uint32_t)(bt >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bt, 0u));
            const uint32_t nodeFit = min(nN, (uint32_t)kMaxPieces / 5u);          // requests beyond the staging area wait for the next iteration
            const uint32_t nodePieces = nodeFit * 5u;
            const uint32_t triFit = min(nT, ((uint32_t)kMaxPieces - nodePieces) / 3u);
            const uint32_t total = nodePieces + triFit * 3u;
            if (wantNode) offs[rankN] = nodeOff;
            if (wantTri) offs[64 + rankT] = triOff;
            gotTri = wantTri && rankT < triFit; gotNode = wantNode && rankN < nodeFit;
            for (uint32_t base = 0; base < total; base += 64u) {
                const uint32_t g = base + lane;
                const bool isNode = g < nodePieces;
                const uint32_t h = isNode ? g : g - nodePieces;
                const uint32_t o = isNode ? (h * 205u) >> 10 : (h * 171u) >> 9;          // h / 5 (h < 1024), h / 3 (h < 512)
                const uint32_t p = h - o * (isNode ? 5u : 3u);
                if (g < total) {
                    const uint32_t off = offs[isNode ? o : 64u + o] + p * 16u;
                    const char* src = (isNode ? nodes : tris) + off;
                    if (MODE == 1) {
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(stage + base), 16, 0, 0);
                    } else {
                        stage[g] = *(const v4f*)src;
                    }
                }
            }
            if (MODE == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (gotNode) { LdsV4 q = (LdsV4)((__attribute__((address_space(3))) char*)stage + rankN * 80u); n0 = q[0]; n1 = q[1]; n2 = q[2]; n3 = q[3]; n4 = q[4]; }
            if (gotTri) { LdsV4 q = (LdsV4)((__attribute__((address_space(3))) char*)stage + nodePieces * 16u + rankT * 48u); t0 = q[0]; t1 = q[1]; t2 = q[2]; }
        }
        uint32_t h = 0;
        if (gotNode) { h ^= __float_as_uint(n0.x) + __float_as_uint(n1.y) + __float_as_uint(n2.z) + __float_as_uint(n3.w) + __float_as_uint(n4.x); nFetched++; }
        if (gotTri) { h ^= __float_as_uint(t0.x) + __float_as_uint(t1.y) + __float_as_uint(t2.z); nFetched++; }
        acc += h;
        state = mix(state + h);
    }
    out[blockIdx.x * 256u + threadIdx.x] = acc;
    atomicAdd(fetched, (unsigned long long)nFetched);
}

// reference result: mode 0 and the cooperative modes must load the same bytes (acc equal whenever no triangle request was deferred)
template <int MODE, int WAVES> double run(const char* nodes, uint32_t numNodes, const char* tris, uint32_t numTris, int blocks, int steps, uint32_t pN, uint32_t pT,
                                          uint32_t* out, unsigned long long* fetched, unsigned long long* nFetched, uint32_t* checksum) {
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL((fetch_kernel<MODE, WAVES>), dim3(blocks), dim3(256), 0, 0, nodes, numNodes, tris, numTris, 8, pN, pT, out, fetched);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemset(fetched, 0, 8));
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL((fetch_kernel<MODE, WAVES>), dim3(blocks), dim3(256), 0, 0, nodes, numNodes, tris, numTris, steps, pN, pT, out, fetched);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    CHECK(hipMemcpy(nFetched, fetched, 8, hipMemcpyDeviceToHost));
    std::vector<uint32_t> h((size_t)blocks * 256);
    CHECK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
    uint32_t c = 0; for (uint32_t v : h) c = c * 31u + v;
    *checksum = c;
    return ms;
}

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const uint32_t numNodes = 26000, numTris = 262267;          // Sponza-stat: 2 MB of nodes (L2), 12.6 MB of triangle records
    char *nodes, *tris; uint32_t* out; unsigned long long* fetched;
    CHECK(hipMalloc(&nodes, (size_t)numNodes * 80 + 256)); CHECK(hipMalloc(&tris, (size_t)numTris * 48 + 256));
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    CHECK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4)); CHECK(hipMalloc(&fetched, 8));
    { std::vector<uint32_t> h(((size_t)numTris * 48 + 256) / 4); uint32_t x = 1; for (auto& v : h) { x = x * 1664525u + 1013904223u; v = x; }
      CHECK(hipMemcpy(tris, h.data(), h.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(nodes, h.data(), (size_t)numNodes * 80, hipMemcpyHostToDevice)); }
    const int steps = 2000;
    const double clk = prop.clockRate * 1e3;
    std::vector<std::pair<uint32_t, uint32_t>> cases = {{26, 19}, {44, 19}, {64, 0}, {0, 19}, {44, 0}};
    if (argc >= 3) cases = {{(uint32_t)atoi(argv[1]), (uint32_t)atoi(argv[2])}};
    printf("%d CUs at %.2f GHz; one workgroup = 4 waves; per case: ms, G records/s, cycles per wave iteration per CU (= what the memory side of one loop iteration costs a CU)\n", cus, clk * 1e-9);
    for (auto [pN, pT] : cases) {
        printf("node lanes %u / 64, triangle lanes %u / 64:\n", pN, pT);
        unsigned long long nf; uint32_t cs0 = 0, cs;
#define RUN(MODE, WAVES, name) { const int blocks = cus * WAVES; const double ms = run<MODE, WAVES>(nodes, numNodes, tris, numTris, blocks, steps, pN, pT, out, fetched, &nf, &cs); \
            const double iters = (double)blocks * 4 * steps / cus; \
            printf("  %-44s %2d waves/SIMD  %7.3f ms  %6.1f G rec/s  %6.1f cycles/iteration/CU  checksum %08x%s\n", name, WAVES, ms, nf / (ms * 1e-3) / 1e9, ms * 1e-3 * clk / iters, cs, \
                   MODE == 0 ? "" : (cs == cs0 ? " (= mode 0)" : " (differs: deferred requests)")); if (MODE == 0) cs0 = cs; }
        RUN(0, 7, "per-lane gathers (the loop today)")
        RUN(1, 7, "cooperative, LDS-DMA")
        RUN(2, 7, "cooperative, through registers")
        RUN(0, 4, "per-lane gathers")
        RUN(1, 4, "cooperative, LDS-DMA")
        RUN(2, 4, "cooperative, through registers")
    }
    return 0;
}
