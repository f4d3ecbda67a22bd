// pathtrace_wf.hip -- the production path tracer: a WAVEFRONT (multi-kernel) formulation for gfx950.
//
// Why not one megakernel?  Measured on MI355X in round 1 (profiles/r01_v2_*, the variants have since been removed):
// the fused kernels need 147-169 VGPRs (2-3 waves per SIMD) and run at 31 % VALU lane utilisation.  MI355X has
// 288 GB of HBM3E at ~8 TB/s, so the per-path state lives in HBM (272 B + 8 x 48 B of shadow-ray queue per pixel)
// and the work is cut into lean kernels that each do one thing at high occupancy:
//
//   per bounce b = 0..3, over the ray list L_b (everything a kernel touches is indexed by LIST SLOT: coalesced streams):
//     extend     closest hit (TraceRay stand-in).  b = 0: wf_primary_kernel -- rgsMain's camera ray per pixel (8x8 pixel tiles per
//                wave) and the tile traced as ONE PACKET, then appended to L_0;  b > 0: persistent waves, one ray per lane,
//                lane refill                                                                          [traversal only]
//     shade      miss: msDynamicSky's term into the slot's record.  hit: chsMain without its light loop -- surface, material, emissive, sky
//                sample direction, BRDF-sampled continuation ray appended to L_{b+1}; then, per workgroup, the light loop's candidates
//                (shadow_gen_block): one lane per (vertex, light of the vertex's grid cell) decides whether a shadow ray is
//                needed (request mask), and one lane per RAY writes the queue record and the light's -- or the sky sample's --
//                radiance along it                                                                    [no traversal]
//     shadow     RayQuery stand-in: traces the queued rays, zeroes the visibility word of occluded ones [traversal only]
//     accumulate chsMain's sum IN THE REFERENCE'S LOOP ORDER: colour so far + the vertex's own term + the records left visible, the sky
//                sample's last; finished paths are stored + integrated                                  [no traversal]
//   The MAIN CHAIN of a frame is primary -> shade(0) -> extend(1) -> shade(1) -> ... -> shade(3): nothing on it needs the colour gathered so
//   far.  The shadow rays of all bounces go into one queue, traced by any-hit launches on a side stream beside the main chain, and the
//   accumulate kernels run once their rays are in (DEFERRED ACCUMULATION, launcher at the end of this file).
//   (Rounds 1-3 also carried the unfused forms -- wf_generate_kernel + wf_extend_packet_kernel, wf_shadow_gen_kernel -- behind environment
//   hooks for A/B runs; they were removed in round 4: commit 9a57a74 / 3220d1d hold them with their measurements.)
//
// A frame can be cut into PIPELINES (bands of tile rows on their own streams); with the side stream one pipeline is the fastest issue.
//
// The persistent traversal kernels refill lanes (Aila & Laine 2009, adapted to 64-wide waves): a wave owns a chunk of rays taken
// from a device-wide cursor; every loop iteration each lane tests one 8-wide node and / or one pending leaf triangle; lanes
// whose ray is done are handed the next rays of the chunk as soon as enough lanes are idle.  Every wave reaches the exit (cursor
// exhausted and no lane active), so the grid always drains.  They are VALU-ISSUE bound (profiles/r02_pmc_summary*.txt: issue
// cycles = elapsed cycles, 43-51 % of the lanes useful: a node step and a triangle step are different code, which is why a lane
// makes both in one iteration whenever it can -- pt_traverse.h trav_closest_step2 / trav_any_step2).
//
// The float operations per pixel and their order are exactly those of the reference HLSL; only the schedule differs.
//
// Light loop: chsMain loops over ALL scene lights per hit (pathtracing.hlsl:230-236).  A world-space light
// grid (scene_build.h LightGrid) yields, per hit, the ascending sub-list of lights that can be non-zero there;
// the skipped terms are exactly `attenuation * 0`, so the sum is unchanged (non-finite attenuation falls back to
// the full list).
#include "pt_shade.h"
#include <algorithm>

extern "C" hipError_t incr_launch_fold_ray_stats(unsigned long long* stripes, incr_ray_stats* stats, hipStream_t stream);   // raster.hip

namespace incr {

static constexpr int kWfBlock = 256;
// Waves per SIMD the shade / accumulate kernels are compiled for (measured at 4K Sponza-stat, ms per frame for
// 3 / 4 / 5 / 6 / 8 waves: shade 7.5 / 7.5 / 9.5 / 10.8 / 10.9, accumulate 4.6 / 3.8 / 5.0 / 5.3 / 5.9): one more resident
// wave pays for accumulate; below 128 VGPRs both kernels spill and lose more than the occupancy gives.
// Round 4: as single-pass kernels (no grid-stride loop for the compiler to hoist ~140 invariants out of) shade needs 117 VGPRs and
// accumulate 113; shade's scratch was trimmed to 40 KB so that four workgroups fit a CU (3 -> 4 waves: 4K frame 17.13 -> 16.60 ms).
#ifndef INCR_SHADE_WAVES
#define INCR_SHADE_WAVES 4
#endif
#ifndef INCR_EXTEND_WAVES
#define INCR_EXTEND_WAVES 7     // waves per SIMD of the closest-hit kernel with the 8-entry stack: at 8 (64 VGPRs) it spills 28 B inside the step; 12.3 / 11.8 / 12.0 ms per 4K frame at 8 / 7 / 6
#endif
#ifndef INCR_ACCUM_WAVES
#define INCR_ACCUM_WAVES 4
#endif
// LDS stack entries per lane in the traversal kernels: a template parameter, the launcher picks the smallest of
// 8 / 10 / 12 / 16 / 24 that holds the scene's tree (one entry per level below the root, pt_traverse.h): Sponza-stat and
// Bistro-scale (10 levels with the SAH-optimal collapse) take 10.
#ifndef INCR_EXTEND_WAVES10
#define INCR_EXTEND_WAVES10 7
#endif
#ifndef INCR_SHADOW_WAVES
#define INCR_SHADOW_WAVES 7     // the overlapped step (pt_traverse.h trav_any_step2) keeps a node and a triangle in flight: 72 VGPRs; at 8 waves (64) it spills 28 B and loses (shadow 9.4 ms against 7.9)
#endif
// WORKGROUPS OF THE PERSISTENT TRAVERSAL KERNELS (round 5: the workgroup's size is a parameter).  The LDS copy of the hottest nodes (NODE ORDER,
// scene_build.cpp) serves any number of waves, so fewer, larger workgroups per CU could keep more nodes: 7 workgroups of 4 waves hold 151 nodes each for the
// any-hit kernel (48 % of its node visits by tools/trav_sim) and 23 for the closest-hit kernel, whose stack carries keys (23 %); 4 workgroups of 7 waves hold
// 275 / 51 (55 % / 30 %), 2 workgroups of 14 waves 560 / 110 (64 % / 41 %).  Measured, same box, ms per 4K frame with every kernel alone, any-hit /
// closest-hit (profiles/r05_ab_trav_blocks.txt): 7 workgroups 7.33-7.41 / 7.43-7.49, 4 workgroups 7.44-7.52 / 7.78-7.82, 2 workgroups 8.44-8.55 / 8.71-8.81
// (frame 16.5 / 17.3 / 20.1 ms): what the larger copy takes off the vector memory path is less than what the larger workgroups cost (their waves spread
// unevenly over the four SIMDs and the CU's LDS is handed out in two or four pieces instead of seven).  Seven it stays.
#ifndef INCR_TRAV_BLOCKS_PER_CU
#define INCR_TRAV_BLOCKS_PER_CU 7
#endif
template <int N, bool SHADOW> struct TravCfg {
    static constexpr int kWaves = SHADOW ? (N <= 16 ? INCR_SHADOW_WAVES : 6) : (N <= 8 ? INCR_EXTEND_WAVES : (N <= 10 ? INCR_EXTEND_WAVES10 : (N <= 12 ? 6 : (N <= 16 ? 4 : 3))));      // waves per SIMD (VGPRs, LDS stack bytes)
    static constexpr int kBlocksPerCU = (kWaves * 4) % INCR_TRAV_BLOCKS_PER_CU == 0 ? INCR_TRAV_BLOCKS_PER_CU : kWaves;
    static constexpr int kBlock = kWaves * 256 / kBlocksPerCU;                  // threads per workgroup
    static constexpr int kStackBytes = N * kBlock * (SHADOW ? 4 : 6);           // 32-bit entries; the closest-hit kernel's 16-bit keys beside them (pt_traverse.h LaneStackT)
    static constexpr int kLdsBudget = 163840 / kBlocksPerCU - 1024;             // 160 KB per CU; 1 KB for allocation granularity
    static constexpr uint32_t kTop = (uint32_t)((kLdsBudget - kStackBytes) / 80 < 8 ? 8 : (kLdsBudget - kStackBytes) / 80);      // hottest nodes kept in LDS (80 B each)
    static_assert(kBlock % 64 == 0 && kBlock <= 1024, "traversal workgroup: whole waves, at most 16");
    static_assert(kStackBytes + (int)kTop * 80 <= 163840 / kBlocksPerCU, "traversal workgroup: LDS");
};
template <int N, bool SHADOW> using TravStack = LaneStackT<N, TravCfg<N, SHADOW>::kBlock, !SHADOW, true>;      // any-hit: plain; closest-hit: keyed, drops groups beyond the best hit
template <int BLOCK>
DI uint32_t load_top_nodes(const PtParams& P, float4* s_top, uint32_t capacity) {
    const uint32_t count = min(capacity, P.numNodes);
    for (uint32_t i = threadIdx.x; i < count * 5u; i += (uint32_t)BLOCK) s_top[i] = ((const float4*)P.nodes)[i];
    __syncthreads();
    return count;
}
#ifndef INCR_GEN_LDS_LIGHTS
#define INCR_GEN_LDS_LIGHTS 128
#endif
static constexpr uint32_t kGenLdsLights = INCR_GEN_LDS_LIGHTS;         // lights whose resolved records shadow_gen keeps in LDS (80 B each)
static constexpr uint32_t kSkySlot = 127;              // bit 127 of the request mask = the sky sample
static constexpr uint32_t kFlagTerminate = 1u, kFlagSkip = 2u, kFlagInlineShadow = 4u, kFlagManyLights = 8u, kFlagFullRecord = 16u, kFlagMiss = 32u;
// Items a wave takes from the device-wide cursor at a time, and the number of idle lanes that triggers a refill.
// Round 1, one setting for both kernels, 4K Sponza-stat, ms per frame for chunks of 32 / 64 / 96 / 128 / 256 / 512 / 1024
// items: 44.9 / 44.5 / 45.5 / 46.3 / 51.4 / 55.0 / 58.6 -- with 8192 persistent waves a 512-item chunk is a quarter of a
// wave's whole share of a 4K list and the kernel ends on the waves that drew a late one.  Round 2, per kernel
// (gpurun_out/ab_r2_1.log): extend 12.3 ms at 64 items / 12 lanes, 13.1 at 128 items, 12.6 at 32 lanes; shadow, whose
// refill is two coalesced loads, 12.2 ms at 64 / 12, 11.8 at 32 lanes, 11.5 at 128 items; with the LDS node copy 11.0 / 11.2 / 11.5 ms at
// 24 / 32 / 40 lanes and 11.6 at 256 items (gpurun_out/ab_r2_3.log).
#ifndef INCR_EXTEND_CHUNK
#define INCR_EXTEND_CHUNK 64
#endif
#ifndef INCR_SHADOW_CHUNK
#define INCR_SHADOW_CHUNK 128
#endif
#ifndef INCR_EXTEND_CHUNK_LARGE
#define INCR_EXTEND_CHUNK_LARGE 128      // closest-hit chunk of a pipeline with >= 4 M paths (launcher)
#endif
// Round 4, one pipeline with the any-hit launches beside the main chain (r04_ab_chunks: same box, two runs each): extend 16 / shadow 12 idle lanes 17.61-17.65 ms per
// 4K frame, 12 / 16: 17.43-17.48, 12 / 12: 17.56-17.58, 16 / 16: 17.53-17.62, 12 / 20: 17.58-17.59, 8 / 16: 17.50-17.52.
#ifndef INCR_EXTEND_REFILL
#define INCR_EXTEND_REFILL 12
#endif
#ifndef INCR_SHADOW_REFILL
#define INCR_SHADOW_REFILL 16
#endif
static constexpr uint32_t kExtendChunk = INCR_EXTEND_CHUNK, kShadowChunk = INCR_SHADOW_CHUNK;
static constexpr uint32_t kExtendRefill = INCR_EXTEND_REFILL, kShadowRefill = INCR_SHADOW_REFILL;
static constexpr uint32_t kMissTri = 0xffffffffu;
#ifndef INCR_WF_CURSOR_STRIPES
#define INCR_WF_CURSOR_STRIPES 8      // 1 = one cursor word per launch (rounds 1-3).  The chunk requests of 7,168 waves on ONE word are a queue of same-address atomics: with
#endif                                // 64-ray any-hit chunks the 4K frame took 21.5 ms on one cursor and 17.7 on eight; at the shipped chunk sizes eight stripes give 17.68 -> 17.55 ms (r04_ab_chunks)
// striped cursors of the persistent kernels (WaveChunk): counters[kStripeBase + (launch slot * stripes + stripe) * 32], launch slot = bounce for the
// closest-hit launches, 4 + launch index for the any-hit ones; zeroed by wf_init_counters_kernel
static constexpr int kStripeBase = 128;
// (Round 4 measured two ways of changing which rays share a shadow wave and removed both -- commit e325fc1 holds the code, profiles/r04_ab_ray_order.txt
// the numbers: a LIGHT-MAJOR queue order out of shadow_gen_block (LDS binning by scene light: shadow 7.48 -> 7.55 ms alone, shade + 0.19 ms for the
// binning: a lane-per-ray wave pays per iteration, not per divergence) and a per-lane OCCLUDER CACHE (the triangle that stopped the lane's previous ray
// tested first: 7.56 -> 7.82 ms vertex-major, 7.50 -> 7.57 light-major).  Also measured and removed: ONE RAY AHEAD PER LANE in wf_shadow_kernel -- the refill hands a lane
// its next ray while it still traces the current one, and a lane that finishes starts it in the same iteration: 63.8 live lanes in the steady state (53.6), 12.4 M wave
// iterations per frame (14.3 M) -- and the 4K frame 16.80-16.87 ms against 16.63-16.65, 16.56-16.65 with the refill at 32 used-up lanes: the ray start (three v_rcp_f32,
// ~15 moves, the visibility store) then runs in nearly every iteration instead of every fifth, and what the fuller lanes save it costs (profiles/r04_ab_lane_fill.txt).)
// counters: [b] = |L_b| for b = 0..4 ; cursors at [16 + 2b] (extend) and [16 + 2b + 1] (shadow) ; [32 + b] = shadow rays queued in bounce b
// (round 4: ONE shadow-ray queue for the four bounces -- [32] counts its rays; [40 + b] = its length before the rays of bounce b, [40 + b + 1] after
//  them, written by wf_snapshot_kernel behind wf_shade_kernel(b): a shadow launch traces the rays of one or several bounces as the range between two snapshots)
static constexpr int kCursorBase = 16, kRayCountBase = 32, kSnapBase = 40;
static constexpr int kQueueCountBase = 96;     // entries of the shadow-ray QUEUE (rayQ): the rays asked for (kRayCountBase: the RECORDS, rayC) less those the sun cover map answered;
                                               // what the snapshots copy and the any-hit launches walk.  On a cache line of its own: both counters take an atomic per workgroup.
static constexpr int kAnsweredBase = 33;      // shadow rays the sun cover map answered at their set-up (queued as null rays): incr_pathtrace_profile::shadow_rays_answered
// The dynamic sky's launch constants (Perez coefficients + the miss shader's sun: 32 floats) live in the counters block as well, written by
// wf_init_counters_kernel: wf_accumulate_kernel reads them with scalar loads inside its miss branch only, so that they hold no SGPRs on the
// (far more common) hit side -- as kernel arguments they are preloaded for the whole kernel and spill
static constexpr int kSkyConstBase = 4096;
struct WfSkyConsts { incr_perez_distribution perez; SkySun sunMiss; };
static_assert(sizeof(WfSkyConsts) % 4 == 0 && sizeof(WfSkyConsts) / 4 <= 64, "wf_init_counters_kernel copies WfSkyConsts with its 64 threads, one word each");

// Path state is indexed by LIST SLOT, not by pixel: the rays of a bounce are a flat list (generate / shade append to it), and
// everything a kernel of that bounce touches -- the ray, the colour / attenuation travelling with it, the hit, the vertex record --
// lives at the ray's slot.  Every kernel of a bounce walks the list in order, so all of these are coalesced streams; in round 1
// they were arrays indexed by pixel id, which the list order follows only in bounce 0 (shade ran 1.8x slower per hit on bounces
// 1-3).  Only the final pixel store is a scatter.
// vertex record, in two arrays.  B.vertex, 3 float4 per list slot, is all wf_accumulate_kernel reads of a vertex whose shadow rays went
// through the queue (the usual case: every ray's radiance is in its rayC record):
//   [kVColor] payload.color .xyz so far, slot of the continuation ray in the next bounce's list (0xffffffff: the path ends here)
//   [kVMask]  request mask (4 x u32): the shadow rays of this vertex, bit 127 = the sky sample
//   [kVMeta]  flags, path id, index of the vertex's first queued shadow ray, light-grid cell
// B.vertexFull, 5 float4 per list slot, is written and read only when something has to be re-evaluated per vertex (flags
// kFlagInlineShadow / kFlagManyLights / kFlagFullRecord):
//   [kFPos] hitPos.xyz, metallic   [kFNrm] N.xyz, roughness   [kFBase] basecolor.xyz, incoming ray direction .x
//   [kFAtt] attenuation at the vertex .xyz, incoming ray direction .y   [kFSky] skyDir.xyz, incoming ray direction .z
enum : int { kVColor = 0, kVMask = 1, kVMeta = 2 };
enum : int { kFPos = 0, kFNrm = 1, kFBase = 2, kFAtt = 3, kFSky = 4 };
// Round 4: the records of the FOUR bounces live side by side (B.vertex + bounce * numPaths * 3): wf_accumulate_kernel(b) runs after the shading of
// the later bounces (DEFERRED ACCUMULATION, launcher below), so a bounce cannot reuse the previous bounce's records.  [kVColor].xyz is no longer the
// colour so far but what THIS vertex adds to payload.color before its light loop (attenuation x emissive x 20000; for a miss: the miss shader's term).
DI float4* vertex_of(const WfBuffers& B, int bounce, uint32_t slot) { return B.vertex + ((size_t)bounce * B.numPaths + slot) * 3; }
DI float4* vertex_full_of(const WfBuffers& B, int bounce, uint32_t slot) { return B.vertexFull + ((size_t)bounce * B.numPaths + slot) * 5; }

DI void add_counters(const PtParams& P, const Counters& cnt) {
    if (!P.stats) return;
    uint32_t v[5] = {cnt.camera, cnt.bounce, cnt.skyShadow, cnt.lightShadow, cnt.hits};
#pragma unroll
    for (int k = 0; k < 5; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_down(v[k], off, 64);
    }
    if ((threadIdx.x & 63u) == 0) {
        // 64 striped counters (one cache line each, folded into P.stats after the frame) when the context provides them:
        // 8192 waves x up to 3 counters on ONE line were a queue of same-line atomics at the tail of every kernel
        unsigned long long* s = P.statStripes ? P.statStripes + (blockIdx.x & 63u) * 16u : (unsigned long long*)P.stats;
#pragma unroll
        for (int k = 0; k < 5; ++k) if (v[k]) atomicAdd(s + k, (unsigned long long)v[k]);
    }
}

// Block-aggregated append: every thread of the block calls this the same number of times.
// Returns the slot for threads with `want`, one global atomic per block.
DI uint32_t block_append(bool want, uint32_t* __restrict__ counter, uint32_t* s_waveCount, uint32_t* s_base) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned long long m = __ballot(want);
    const uint32_t rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) s_waveCount[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
#pragma unroll
        for (int w = 0; w < kWfBlock / 64; ++w) { const uint32_t c = s_waveCount[w]; s_waveCount[w] = total; total += c; }
        *s_base = total ? atomicAdd(counter, total) : 0u;
    }
    __syncthreads();
    const uint32_t slot = *s_base + s_waveCount[wave] + rank;
    __syncthreads();
    return slot;
}

DI uint32_t light_cell(const PtParams& P, f3 p) {
    const float fx = fminf(fmaxf((p.x - P.gridLo[0]) * P.gridInvCell[0], 0.f), (float)(P.gridDim[0] - 1u));
    const float fy = fminf(fmaxf((p.y - P.gridLo[1]) * P.gridInvCell[1], 0.f), (float)(P.gridDim[1] - 1u));
    const float fz = fminf(fmaxf((p.z - P.gridLo[2]) * P.gridInvCell[2], 0.f), (float)(P.gridDim[2] - 1u));
    return ((uint32_t)fz * P.gridDim[1] + (uint32_t)fy) * P.gridDim[0] + (uint32_t)fx;     // NaN -> fmaxf picks 0
}

// Wave-level work distribution: hands the next items of the wave's chunk to idle lanes (consecutive indices in
// lane order), taking a new chunk from the device-wide cursor when the current one is used up.
template <uint32_t kChunkDefault, int BLOCK> struct WaveChunk {
    uint32_t next, end;     // wave-uniform
    bool exhausted;
#if INCR_WF_CURSOR_STRIPES > 1
    uint32_t stripe = blockIdx.x % INCR_WF_CURSOR_STRIPES, tried = 0u;
#endif
    uint32_t kChunk = kChunkDefault;      // items per grab: wave-uniform; the closest-hit launches of a large frame take 128 (below), everything else the compile-time size
    // A wave's FIRST chunk is static -- chunk number = its index in the grid -- and the launcher starts the cursors behind
    // that round: the 8192 waves of the grid do not queue on one cursor word for their first item (or just to learn there
    // is nothing left: a wave without a static chunk leaves at once).  Chunks are kChunk items in both rounds, so a short
    // list (a 1/8 framebuffer shard, a late bounce) still spreads over every wave that can get one.
    // (Asking for every further chunk one chunk AHEAD, so that the returning atomic on the cursor hides behind a traversal
    // fetch, was measured and dropped: shadow 10.6 -> 11.1 ms per frame -- loads and atomics return in order, so every step
    // issued after the ask waits for the contended atomic instead of only the one step at the chunk boundary.)
    DI void init_static(uint32_t total) {
        const uint32_t b = __builtin_amdgcn_readfirstlane((blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6)) * kChunk);
        exhausted = b >= total;
        next = exhausted ? 0u : b; end = exhausted ? 0u : min(b + kChunk, total);
    }
    // returns the item index for this lane or 0xffffffff
    DI uint32_t take(bool idle, uint32_t* __restrict__ cursor, uint32_t total) {
        const unsigned long long m = __ballot(idle);
        if (m == 0ull) return 0xffffffffu;
        if (next >= end && !exhausted) {
#if INCR_WF_CURSOR_STRIPES > 1
            // STRIPED cursors: the part of the list behind the static round is cut into INCR_WF_CURSOR_STRIPES contiguous segments, each with a cursor of
            // its own (on its own cache line); a wave draws from its home segment (workgroup index modulo the stripe count) and moves on to the next
            // when that one is used up -- the grid's chunk requests no longer queue on one word
            const uint32_t dynBase = min(gridDim.x * (BLOCK / 64) * kChunk, total), dynTotal = total - dynBase;
            const uint32_t segLen = ((dynTotal + INCR_WF_CURSOR_STRIPES * kChunk - 1u) / (INCR_WF_CURSOR_STRIPES * kChunk)) * kChunk;
            for (;;) {
                if (tried >= (uint32_t)INCR_WF_CURSOR_STRIPES || segLen == 0u) { exhausted = true; break; }
                uint32_t b = 0;
                if ((threadIdx.x & 63u) == 0) b = atomicAdd(cursor + stripe * 32u, kChunk);
                b = __builtin_amdgcn_readfirstlane(b);
                const uint32_t segBegin = dynBase + stripe * segLen;
                if (b < segLen && segBegin + b < total) { next = segBegin + b; end = min(min(next + kChunk, segBegin + segLen), total); break; }
                ++tried; stripe = stripe + 1u == (uint32_t)INCR_WF_CURSOR_STRIPES ? 0u : stripe + 1u;
            }
#else
            uint32_t b = 0;
            if ((threadIdx.x & 63u) == 0) b = atomicAdd(cursor, kChunk);
            b = __builtin_amdgcn_readfirstlane(b);
            if (b >= total) exhausted = true;
            else { next = b; end = min(b + kChunk, total); }
#endif
        }
        if (next >= end) return 0xffffffffu;
        const uint32_t lane = threadIdx.x & 63u;
        const uint32_t rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        const uint32_t avail = end - next;
        const uint32_t item = (idle && rank < avail) ? next + rank : 0xffffffffu;
        next += min((uint32_t)__popcll(m), avail);
        return item;
    }
};

// ------------------------------------------------------------------------------------------------ generate
// rgsMain's ray for list item i (pathtracing.hlsl:89-119): 8x8 pixel tile per wave -- the primary rays of a wave walk the same BVH lines;
// samples in flight: the tile rows of sample s follow those of sample s - 1 (tilesY rows each).  false: no ray (outside the frame, or a
// shard's padding row, whose pixel is cleared here).
DI bool generate_lane(const PtParams& P, uint32_t i, uint32_t total, uint32_t tilesX, uint32_t tilesY, uint32_t tileBase, float4* rayO, float4* rayD) {
    const uint32_t lt = i >> 6, w = i & 63u;
    // (LIST ORDER of bounce 0 in column strips of 8 / 16 / 32 tiles instead of full-width tile rows -- what the grid works on at any time is then a compact window of the
    // image -- was measured in round 5 for the L2's sake: 4K frame 16.33-16.47 ms against 16.44-16.49, every kernel within its noise: profiles/r05_ab_tile_strip.txt.)
    const uint32_t tile = lt + tileBase;
    const uint32_t tyv = tile / tilesX, smp = tyv / tilesY;
    const uint32_t px = (tile - tyv * tilesX) * 8u + (w & 7u), ly = (tyv - smp * tilesY) * 8u + (w >> 3);
    *rayO = make_float4(0.f, 0.f, 0.f, 0.f); *rayD = make_float4(0.f, 0.f, 1.f, 0.f);
    if (!(i < total && px < P.W && ly < P.localRows)) return false;
    const int64_t gy = shard_global_row(P, ly);
    if (gy < 0) { store_pixel(P, px, ly, smp, {0.f, 0.f, 0.f, 0.f}); return false; }     // shard padding row
    const uint32_t pid = smp * P.localPixels + ly * P.W + px;
    const uint32_t pixelIdx = (uint32_t)gy * P.W + px, sampleIdx = P.sampleIdx + smp;
    uint32_t raygenSampleSetIdx = 0;
    const f2 subPixelJitter = SamplePoint(pixelIdx, sampleIdx, raygenSampleSetIdx, P.divSqrtN, P.divN, P.numPixels);
    const f2 apertureSample = SamplePoint(pixelIdx, sampleIdx, raygenSampleSetIdx, P.divSqrtN, P.divN, P.numPixels);
    const Ray ray = GenerateCameraRay(P, {(float)px + subPixelJitter.x, (float)(uint32_t)gy + subPixelJitter.y}, apertureSample);
    *rayO = make_float4(ray.o.x, ray.o.y, ray.o.z, __uint_as_float(pid));
    *rayD = make_float4(ray.d.x, ray.d.y, ray.d.z, 0.f);
    return true;
}
DI void store_primary(const WfBuffers& B, uint32_t slot, float4 rayO, float4 rayD) {
    B.extA[(size_t)slot * 2] = rayO; B.extA[(size_t)slot * 2 + 1] = rayD;
    // (payload.color = 0 and payload.attenuation = 1 of a camera ray are not stored: wf_shade_kernel knows them at bounce 0)
}
#ifdef INCR_WF_LANE_STATS
// Developer build only (tools/lane_stats.py): where the lanes of the traversal waves go.  Per kernel (0 extend, 1 shadow) 8 words:
// iterations, live lanes, lanes taking a node step, lanes taking a triangle step (a lane can take both), iterations after the cursor ran dry, live lanes in those,
// iterations that ran a refill, iterations in which only one kind of step was present.
__device__ unsigned long long g_laneStats[16];
struct LaneStats {
    uint32_t it = 0, live = 0, node = 0, tri = 0, drainIt = 0, drainLive = 0, refills = 0, pure = 0;
    // node / tri = the lanes that take a node step / a triangle step in this iteration (trav_*_step2: haveCur && !triMask1 / triMask != 0; a lane can take both)
    DI void step(bool active, bool wantNode, bool wantTri, bool exhausted) {
        const uint32_t a = (uint32_t)__popcll(__ballot(active)), n = (uint32_t)__popcll(__ballot(active && wantNode)), t = (uint32_t)__popcll(__ballot(active && wantTri));
        it++; live += a; node += n; tri += t; if (exhausted) { drainIt++; drainLive += a; } if (!n || !t) pure++;
    }
    DI void flush(int k) {
        if ((threadIdx.x & 63u) == 0) {
            const uint32_t v[8] = {it, live, node, tri, drainIt, drainLive, refills, pure};
            for (int i = 0; i < 8; ++i) atomicAdd(&g_laneStats[k * 8 + i], (unsigned long long)v[i]);
        }
    }
};
extern "C" int incr_debug_lane_stats(unsigned long long* out16) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_laneStats), sizeof(g_laneStats)) != hipSuccess) return -1;
    unsigned long long zero[16] = {};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_laneStats), zero, sizeof(zero)) == hipSuccess ? 0 : -1;
}
#define LANE_STATS(x) x
#else
#define LANE_STATS(x)
#endif

// ------------------------------------------------------------------------------------------------ extend
// (INCR_EXTEND_WAVES10 = 7, with TravCfg above: the 10-entry keyed stack takes 70 VGPRs since the hit's shading index is its flattened id and 1 / d comes from v_rcp)
template <int N> __global__ __launch_bounds__((TravCfg<N, false>::kBlock), (TravCfg<N, false>::kWaves)) void wf_extend_kernel(const PtParams P, const WfBuffers B, int bounce, uint32_t chunkSize) {
    typedef TravCfg<N, false> Cfg;
    __shared__ uint32_t s_stack[N * Cfg::kBlock];
    __shared__ uint16_t s_keys[N * Cfg::kBlock];
    __shared__ float4 s_top[Cfg::kTop * 5];
    const uint32_t total = B.counters[bounce];
    // a workgroup whose first wave has no static chunk has none at all (WaveChunk::init_static) and never takes one from the cursor: it leaves
    // before copying the top nodes -- a short list (a small frame, a late bounce) is then a short launch
    if (blockIdx.x * (Cfg::kBlock / 64) * chunkSize >= total) return;
    TravStack<N, false> st; st.bind(s_stack + threadIdx.x, s_keys + threadIdx.x); st.sp = 0;
    st.bind_top(s_top, load_top_nodes<Cfg::kBlock>(P, s_top, Cfg::kTop));
    // the rays of this bounce: a flat queue in list order (origin + path id, direction), written by generate / shade -- a refill
    // is two coalesced loads, not a path id and two gathers behind it
    const float4* __restrict__ ext = (bounce & 1) ? B.extB : B.extA;
#if INCR_WF_CURSOR_STRIPES > 1
    uint32_t* cursor = B.counters + kStripeBase + bounce * INCR_WF_CURSOR_STRIPES * 32;
#else
    uint32_t* cursor = B.counters + kCursorBase + 2 * bounce;
#endif
    WaveChunk<kExtendChunk, Cfg::kBlock> chunk; chunk.kChunk = chunkSize; chunk.init_static(total);
    bool active = false; uint32_t slot = 0;
    TravClosest2 T; trav_closest_begin(T, {0.f, 0.f, 0.f}, {0.f, 0.f, 1.f}, 0.001f, 10000.0f); T.triBase1 = 0; T.triMask1 = 0;      // the interval every ray of this kernel has: tmin / tmax stay compile-time constants
    uint32_t rays = 0;
    LANE_STATS(LaneStats ls;)
    // (The software-pipelined form of wf_shadow_kernel -- loads issued by the refill, rays started after the step, hits stored
    // by the next refill -- was measured here too: 12.2 ms per frame against 11.8; the closest-hit state already fills the
    // 72 VGPRs of 7 waves per SIMD and the eight registers of a ray in flight spill.)
    for (;;) {
        const unsigned long long act = __ballot(active);
        const uint32_t idle = 64u - (uint32_t)__popcll(act);
        if (!chunk.exhausted && (idle >= kExtendRefill || act == 0ull)) {
            const uint32_t item = chunk.take(!active, cursor, total);
            LANE_STATS(ls.refills++;)
            if (item != 0xffffffffu) {
                const float4 o = ext[(size_t)item * 2], d = ext[(size_t)item * 2 + 1];
                slot = item;
                trav_closest_begin(T, {o.x, o.y, o.z}, {d.x, d.y, d.z}, 0.001f, 10000.0f);     // TMin / TMax of TraceRay (common.hlsli:49-50,119-120)
                T.triBase1 = 0; T.triMask1 = 0;
                st.reset();
                active = true; rays++;
            }
        }
        if (__ballot(active) == 0ull) { if (chunk.exhausted) break; else continue; }
        LANE_STATS(ls.step(active, T.haveCur && T.triMask1 == 0u, T.triMask != 0u, chunk.exhausted);)
        if (active && trav_closest_step2(P, st, T)) {           // node step and triangle step in one iteration (pt_traverse.h)
            B.hit[slot] = T.found ? make_uint4(T.bestFlat, __float_as_uint(T.best), __float_as_uint(T.bu), __float_as_uint(T.bv))
                                 : make_uint4(kMissTri, 0u, 0u, 0u);
            active = false;
        }
    }
    LANE_STATS(ls.flush(0);)
    Counters cnt = {bounce == 0 ? rays : 0u, bounce == 0 ? 0u : rays, 0, 0, 0};
    add_counters(P, cnt);
}

// ------------------------------------------------------------------------------------------------ extend, camera rays
// Bounce 0 only.  The 64 camera rays of a wave are an 8 x 8 pixel tile (generate_lane) -- lens samples 0.01 apart, directions
// a pixel apart -- so they walk the tree as ONE PACKET: node index, stack, leaf and triangle loops and the 80 / 48-byte fetches are
// wave-uniform (scalar registers and scalar loads), and every lane tests ITS ray against the eight boxes / the triangle, with all
// lanes useful.  The lane-per-ray state machine above spends two thirds of its vector instructions on lanes that are idle or on
// the other kind of step (34 % VALU lane utilisation) -- the price of incoherent rays, which camera rays are not.
// Same result as wf_extend_kernel ray by ray: a lane's closest hit is the minimum over the triangles it tests under the
// (t, flattened id) order, the packet tests a superset of the triangles the lane's own walk would reach, and a triangle outside the
// boxes a ray overlaps cannot be hit by it.  Children are taken in the first live lane's near-to-far order; a node is entered when
// any live ray overlaps it closer than that ray's best hit.
static constexpr int kPacketStack = 32;              // one entry per tree level (kMaxBvhDepth = 25)
struct PacketHit { bool found; float best, bu, bv; uint32_t bestFlat; };
DI PacketHit packet_closest(const PtParams& P, uint32_t* stack, bool have, f3 o, f3 d) {
    const RaySetup rs = ray_setup(o, d);
    const float tmin = 0.001f;                                      // TMin / TMax of TraceRay (common.hlsli:49-50)
    float best = 10000.0f, bu = 0.f, bv = 0.f; uint32_t bestFlat = 0xffffffffu; bool found = false;
    const uint32_t negDir = (d.x < 0.f ? 1u : 0u) | (d.y < 0.f ? 2u : 0u) | (d.z < 0.f ? 4u : 0u);
    const unsigned long long haveMask = __ballot(have);
    if (haveMask != 0ull) {
        const uint32_t lead = (uint32_t)__builtin_ctzll(haveMask);                       // the lane whose ray orders the children
        const uint32_t leadNeg = (uint32_t)__builtin_amdgcn_readlane((int)negDir, (int)lead);
        int sp = 0;
        uint32_t cur = 0; bool haveCur = true;
        for (;;) {
            if (!haveCur) {
                if (sp == 0) break;
                uint32_t e = __builtin_amdgcn_readfirstlane(stack[sp - 1]);
                const uint32_t m = e & 0xffu;
                const uint32_t k = (e & 0x100u) ? 31u - (uint32_t)__builtin_clz(m) : (uint32_t)__builtin_ctz(m);
                cur = (e >> 9) + k;
                e &= ~(1u << k);
                if (e & 0xffu) stack[sp - 1] = e; else --sp;
            }
            haveCur = false;
            const float4* __restrict__ np = (const float4*)P.nodes + (size_t)cur * 5;      // wave-uniform address: scalar loads
            const float4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3], n4 = np[4];
            Node8 N;
            node8_test(n0, n1, n2, n3, n4, rs, tmin, best, N);                               // per lane: its ray, its best hit so far
            const uint32_t mine = have ? N.hitMask : 0u;
            uint32_t any = 0u;
#pragma unroll
            for (int k = 0; k < 8; ++k) any |= (__ballot((mine >> k) & 1u) != 0ull) ? (1u << k) : 0u;
            const uint32_t imask = __builtin_amdgcn_readfirstlane(N.imask), childBase = __builtin_amdgcn_readfirstlane(N.childBase);
            Node8 U = N; U.hitMask = any; U.imask = imask;
            U.leafTris = __builtin_amdgcn_readfirstlane(N.leafTris); U.sortAxis = __builtin_amdgcn_readfirstlane(N.sortAxis);
            uint32_t triMask = node8_tri_mask_uniform(U.hitMask, U.leafTris);
            const uint32_t triBase = __builtin_amdgcn_readfirstlane(N.triBase);
            while (triMask) {
                const uint32_t ti = triBase + (uint32_t)__builtin_ctz(triMask);
                triMask &= triMask - 1u;
                const float4* __restrict__ tp = (const float4*)(P.tris + ti);
                const float4 a = tp[0], b = tp[1], c = tp[2];
                float t, u, v; uint32_t flat, flags;
                if (have && tri_intersect(a, b, c, rs.o, d, tmin, 10000.0f, &t, &u, &v, &flat, &flags)) {
                    const bool better = found ? (t < best || (t == best && flat < bestFlat)) : (t < best);      // as trav_closest_step
                    if (better) {
                        if (flags & kTriMirrored) { const float tmp = u; u = v; v = tmp; }
                        if ((flags & kTriOpaque) || any_hit_accepts(P, flat, u, v)) { found = true; best = t; bestFlat = flat; bu = u; bv = v; }
                    }
                }
            }
            const uint32_t inner = any & imask;
            if (inner) {
                // the lead lane's nearest overlapped child first (if it overlaps none, the first in slot order); the others wait as
                // one group that leaves in slot order from the lead ray's side (device_types.h SLOT ORDER)
                uint32_t nearK = (uint32_t)__builtin_ctz(inner); float nearT = __uint_as_float(0x7f800000u);
                const uint32_t leadHit = (uint32_t)__builtin_amdgcn_readlane((int)mine, (int)lead) & imask;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float tk = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(N.tn[k]), (int)lead));
                    if (((leadHit >> k) & 1u) && tk < nearT) { nearT = tk; nearK = (uint32_t)k; }
                }
                const uint32_t rest = inner & ~(1u << nearK);
                if (rest) {
                    const uint32_t axis = U.sortAxis;
                    if (sp < kPacketStack) stack[sp++] = stack_entry(childBase, (leadNeg >> axis) & 1u, rest);
                }
                cur = childBase + nearK; haveCur = true;
            }
        }
    }
    return {found, best, bu, bv, bestFlat};
}
DI uint4 hit_record(const PacketHit& h) {
    return h.found ? make_uint4(h.bestFlat, __float_as_uint(h.best), __float_as_uint(h.bu), __float_as_uint(h.bv)) : make_uint4(kMissTri, 0u, 0u, 0u);
}
// Camera-ray generation and the packet traversal in one: a lane makes its camera ray, the wave traces its tile as a packet, and only then are
// the ray and the hit appended to bounce 0's list -- a separate generate pass (rounds 1-2) wrote 96 B per pixel that the packet kernel read
// straight back (0.38 ms of a 4K frame); here those stores leave a kernel that is bound by its vector ALU.
__global__ __launch_bounds__(kWfBlock, 6) void wf_primary_kernel(const PtParams P, const WfBuffers B, uint32_t total, uint32_t tilesX, uint32_t tilesY, uint32_t tileBase) {
    __shared__ uint32_t s_stack[(kWfBlock / 64) * kPacketStack];
    __shared__ uint32_t s_waveCount[kWfBlock / 64];
    __shared__ uint32_t s_base;
    float4 rayO, rayD;
    const bool have = generate_lane(P, blockIdx.x * kWfBlock + threadIdx.x, total, tilesX, tilesY, tileBase, &rayO, &rayD);
    const PacketHit h = packet_closest(P, s_stack + (threadIdx.x >> 6) * kPacketStack, have, {rayO.x, rayO.y, rayO.z}, {rayD.x, rayD.y, rayD.z});
    const uint32_t slot = block_append(have, B.counters + 0, s_waveCount, &s_base);
    if (have) { store_primary(B, slot, rayO, rayD); B.hit[slot] = hit_record(h); }
    Counters cnt = {have ? 1u : 0u, 0, 0, 0, 0};
    add_counters(P, cnt);
}

// ------------------------------------------------------------------------------------------------ shadow-ray set-up (called by shade, below)
DI uint32_t nth_set_bit(uint32_t word, uint32_t j) {          // position of the j-th (0-based) set bit of `word`
    uint32_t pos = 0u;
#pragma unroll
    for (uint32_t sft = 16u; sft > 0u; sft >>= 1) {
        const uint32_t c = (uint32_t)__popc((word >> pos) & ((1u << sft) - 1u));
        if (j >= c) { j -= c; pos += sft; }
    }
    return pos;
}
// The shared-memory scratch of the ray set-up (one workgroup = 256 list slots).  With wf_shade_kernel's own few words it has to stay within 40 KB:
// four workgroups per CU (INCR_SHADE_WAVES).
struct ShadowGenLds {
    uint32_t prefix[kWfBlock + 1];
    uint32_t waveTotal[kWfBlock / 64];
    uint4 mask[kWfBlock];
    float4 pos[kWfBlock], sky[kWfBlock] /* .w = incoming ray direction .y */, nrm[kWfBlock], col[kWfBlock], att[kWfBlock], skyRad[kWfBlock] /* .w = incoming ray direction .z */;
    float4 lights[kGenLdsLights * 5];          // the resolved light table (pt_shade.h), when the scene's lights fit
    uint32_t base, baseQ;
    unsigned long long ansBallot[kWfBlock / 64];      // per wave: which vertices have a ray the sun cover map answers
    uint8_t ansSlot[kWfBlock];                        // ... and its request slot (0xff: none): the map is of ONE light, a vertex has at most one such ray
    SunCoverHeader cover;                             // the map's header (copied with the light table): out of HBM it was a scalar load per candidate in a divergent loop
};
// What a lane brings for its list slot: the vertex wf_shade_kernel made (or nothing: a miss, a slot past the list's end).
struct ShadowGenVertex {
    bool have; uint32_t cellLo, cellHi; uint4 req;
    float4 pos /* .w = cellLo */, sky, nrm /* .w = roughness */, col /* basecolor, metallic */, att /* attenuation at the vertex, incoming ray direction .x */;
    float4 skyRad /* skyBrightness x the environment's radiance along sky.xyz (the caller has it from the request test) */;
    float2 vyz /* incoming ray direction .yz */;
};
DI void shadow_gen_load_lights(const PtParams& P, ShadowGenLds& S) {          // visible after the next __syncthreads
    if (P.cover.header && threadIdx.x < sizeof(SunCoverHeader) / 4u) ((uint32_t*)&S.cover)[threadIdx.x] = ((const uint32_t*)P.cover.header)[threadIdx.x];
    if ((uint32_t)P.lightCount <= kGenLdsLights) for (uint32_t k = threadIdx.x; k < (uint32_t)P.lightCount * 5u; k += kWfBlock) S.lights[k] = P.resolvedLights[k];
}
// The ray set-up of one workgroup's 256 list slots (every thread of the workgroup must call it: it synchronises).  `pid` = the lane's
// list slot.  Run by wf_shade_kernel on the vertices it has just made (they are still in registers; rounds 1-2 had a kernel of its own
// that read 112 B of every vertex record back).
// Returns what the caller files in the lane's vertex record: the complete request mask, kFlagManyLights / kFlagInlineShadow, and the index of
// the vertex's first queued ray.
struct ShadowGenResult { uint4 req; uint32_t flags, rayBase; };
DI ShadowGenResult shadow_gen_block(const PtParams& P, const WfBuffers& B, int bounce, ShadowGenLds& S, uint32_t pid, const ShadowGenVertex& V) {
    uint32_t* const s_prefix = S.prefix; uint32_t* const s_waveTotal = S.waveTotal; uint4* const s_mask = S.mask;
    float4* const s_pos = S.pos; float4* const s_sky = S.sky; float4* const s_nrm = S.nrm; float4* const s_col = S.col; float4* const s_att = S.att;
    float4* const s_lights = S.lights; uint32_t& s_base = S.base; float4* const s_skyRad = S.skyRad;
    uint8_t* const s_ansSlot = S.ansSlot; unsigned long long* const s_ansBallot = S.ansBallot; uint32_t& s_baseQ = S.baseQ;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const bool ldsLights = (uint32_t)P.lightCount <= kGenLdsLights;
    auto fetch_light = [&](uint32_t li, incr_light& light, float mm[16]) {
        if (ldsLights) load_light_resolved((const float4*)s_lights, li, light, mm); else load_light_resolved(P.resolvedLights, li, light, mm);
    };
    __syncthreads();          // the scratch may still be read by the workgroup's previous call (wf_shade_kernel's slot loop); also publishes S.lights
    const bool haveVtx = V.have; const uint32_t cellLo = V.cellLo, cellHi = V.cellHi; uint4 req = V.req;
    ShadowGenResult out = {req, 0u, 0u};
    const float4 vpos = V.pos, vsky = V.sky, vnrm = V.nrm, vcol = V.col, vatt = V.att; const float2 vyz = V.vyz;
    // ---- phase 1: which lights of its cell does every vertex need a shadow ray to?  One lane per (vertex, candidate light) pair:
    //      GetDirectRadiance's NoL > 0 && any(radianceIn > 0) (lighting/common.hlsli:84-95), the request bit set in LDS
    const uint32_t cand = haveVtx ? (cellHi - cellLo) : 0u;
    {
        uint32_t incl = cand;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const uint32_t v = __shfl_up(incl, off, 64); if (lane >= (uint32_t)off) incl += v; }
        if (lane == 63u) s_waveTotal[wave] = incl;
        s_mask[tid] = req; s_pos[tid] = vpos; s_sky[tid] = make_float4(vsky.x, vsky.y, vsky.z, vyz.x); s_nrm[tid] = vnrm; s_col[tid] = vcol; s_att[tid] = vatt;
        s_skyRad[tid] = make_float4(V.skyRad.x, V.skyRad.y, V.skyRad.z, vyz.y);
        s_ansSlot[tid] = 0xffu;
        __syncthreads();
        uint32_t waveBase = 0, totalCand = 0;
#pragma unroll
        for (int w = 0; w < kWfBlock / 64; ++w) { const uint32_t c = s_waveTotal[w]; if ((uint32_t)w < wave) waveBase += c; totalCand += c; }
        s_prefix[tid] = waveBase + incl - cand;
        if (tid == 0) s_prefix[kWfBlock] = totalCand;
        __syncthreads();
        uint32_t lightShadow = 0;
        for (uint32_t c = tid; c < totalCand; c += kWfBlock) {
            uint32_t lo = 0u, hi = kWfBlock;
#pragma unroll
            for (int it = 0; it < 8; ++it) { const uint32_t mid = (lo + hi) >> 1; if (s_prefix[mid] <= c) lo = mid; else hi = mid; }
            const uint32_t k = c - s_prefix[lo];
            const float4 v0 = s_pos[lo], n4 = s_nrm[lo];
            incr_light light; float lm[16];
            fetch_light(P.gridCellLights[__float_as_uint(v0.w) + k], light, lm);
            const LightSample ls = light_setup(light, lm, true, {v0.x, v0.y, v0.z}, {n4.x, n4.y, n4.z});
            if (ls.needsShadow) {
                ++lightShadow;
                if (k < kSkySlot) {
                    atomicOr(&((uint32_t*)&s_mask[lo])[k >> 5], 1u << (k & 31u));      // k >= 127: traced inside accumulate, counted here
                    // the sun cover map (pt_params.h SunCover) proves this ray unoccluded: it gets its record ("visible") and no queue entry
                    if (P.cover.header && sun_cover_is_light(S.cover, light, ls.L) && sun_cover_clear(S.cover, P.cover.grid, {v0.x, v0.y, v0.z}, 0.1f)) s_ansSlot[lo] = (uint8_t)k;
                }
            }
        }
        Counters cnt = {0, 0, 0, lightShadow, 0};
        add_counters(P, cnt);
        __syncthreads();
        req = s_mask[tid];          // the complete request mask (accumulate counts the rays with it)
        if (haveVtx && cand > kSkySlot) out.flags |= kFlagManyLights;
    }
    out.req = req;
    const uint32_t n = (uint32_t)(__popc(req.x) + __popc(req.y) + __popc(req.z) + __popc(req.w));
    // ---- phase 2: the rays.  Exclusive scan of n over the workgroup: wave scan by shuffles, then the four wave totals
    uint32_t incl = n;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const uint32_t v = __shfl_up(incl, off, 64); if (lane >= (uint32_t)off) incl += v; }
    __syncthreads();
    if (lane == 63u) s_waveTotal[wave] = incl;
    __syncthreads();
    uint32_t waveBase = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kWfBlock / 64; ++w) { const uint32_t c = s_waveTotal[w]; if ((uint32_t)w < wave) waveBase += c; total += c; }
    __syncthreads();
    s_prefix[tid] = waveBase + incl - n;
    const uint32_t myAns = s_ansSlot[tid];                        // (phase 1's barrier is behind us)
    const unsigned long long ansWave = __ballot(n != 0u && myAns != 0xffu);
    if (lane == 0u) s_ansBallot[wave] = ansWave;
    __syncthreads();
    if (tid == 0) {
        uint32_t answeredTotal = 0;
#pragma unroll
        for (int w = 0; w < kWfBlock / 64; ++w) answeredTotal += (uint32_t)__popcll(s_ansBallot[w]);
        s_prefix[kWfBlock] = total; s_base = total ? atomicAdd(B.counters + kRayCountBase, total) : 0u;
        // records that fit get their queue entries (fewer: the answered rays have none); records beyond the capacity are traced by accumulate and claim no queue space
        const bool fits = !(s_base + total > B.rayCap || s_base + total < s_base);
        s_baseQ = (fits && total > answeredTotal) ? atomicAdd(B.counters + kQueueCountBase, total - answeredTotal) : 0u;
    }
    __syncthreads();
    if (total == 0u) return out;
    const uint32_t base = s_base, baseQ = s_baseQ;
    if (base + total > B.rayCap || base + total < base) {
        if (n) out.flags |= kFlagInlineShadow;                    // queue full: these vertices are traced by accumulate
        return out;
    }
    // queue entries before vertex v's: its records' offset less the answered rays of the vertices before it
    auto answered_before = [&](uint32_t v) {
        uint32_t c = 0;
#pragma unroll
        for (uint32_t w = 0; w < (uint32_t)(kWfBlock / 64); ++w) { const unsigned long long b = s_ansBallot[w]; c += w < (v >> 6) ? (uint32_t)__popcll(b) : (w == (v >> 6) ? (uint32_t)__popcll(b & ((1ull << (v & 63u)) - 1ull)) : 0u); }
        return c;
    };
    if (n) out.rayBase = base + s_prefix[tid];                                  // the vertex's RECORDS (rayC) are [rayBase, rayBase + popc(mask)), ascending slot
    // Which ray is record r?  Owner vertex = the last v with s_prefix[v] <= r, request slot = the (r - s_prefix[v])-th set bit of its mask.
    auto decode = [&](uint32_t r, uint32_t* owner, uint32_t* slotOut) {
        uint32_t lo = 0u, hi = kWfBlock;
#pragma unroll
        for (int it = 0; it < 8; ++it) { const uint32_t mid = (lo + hi) >> 1; if (s_prefix[mid] <= r) lo = mid; else hi = mid; }
        uint32_t j = r - s_prefix[lo];
        const uint4 m = s_mask[lo];
        uint32_t word = m.x, wbase = 0u;
        const uint32_t c0 = (uint32_t)__popc(m.x), c1 = (uint32_t)__popc(m.y), c2 = (uint32_t)__popc(m.z);
        if (j >= c0) { j -= c0; word = m.y; wbase = 32u; if (j >= c1) { j -= c1; word = m.z; wbase = 64u; if (j >= c2) { j -= c2; word = m.w; wbase = 96u; } } }
        *owner = lo; *slotOut = wbase + nth_set_bit(word, j);
    };
    // The RECORD of a ray (rayC: its radiance and visibility, what accumulate sums in chsMain's loop order) and its QUEUE entry (rayQ: what the
    // traversal kernel walks) share the index base + r; the queue entry names the record in direction.w.
    uint32_t answeredRays = 0;
    for (uint32_t r = tid; r < total; r += kWfBlock) {
        uint32_t lo, slot; const uint32_t dest = r;
        decode(r, &lo, &slot);
        const uint32_t vp = pid - tid + lo;          // the owner's list slot: the workgroup's slots are consecutive
        const float4 v0 = s_pos[lo];
        const f3 origin = {v0.x, v0.y, v0.z};
        f3 L, contribution = {0.f, 0.f, 0.f};
        const uint32_t ansOfOwner = s_ansSlot[lo];
        const bool answered = ansOfOwner == slot;       // the sun cover map proved the ray unoccluded (phase 1): a record, no queue entry
        if (slot == kSkySlot) {
            // chsMain's sky sample (pathtracing.hlsl:239-244): GetSkyRadiance with lightVisibility = 1 (lighting/common.hlsli:145-202) over the
            // pdf, times the path attenuation -- evaluated HERE like the lights', so that accumulate only sums
            const float4 v4 = s_sky[lo], n4 = s_nrm[lo], c4 = s_col[lo], a4 = s_att[lo];
            L = {v4.x, v4.y, v4.z};
            const f3 N = {n4.x, n4.y, n4.z};
            MatProps matInfo = {};
            matInfo.basecolor = mk3(c4.x, c4.y, c4.z); matInfo.metallic = c4.w; matInfo.roughness = n4.w;
            const float4 sr = s_skyRad[lo];
            const float2 vyz2 = make_float2(v4.w, sr.w);
            const f3 skyRadianceIn = {sr.x, sr.y, sr.z};
            const f3 radianceOut = sky_shade(skyRadianceIn, L, saturate(dot(N, L)), matInfo, N, -mk3(a4.w, vyz2.x, vyz2.y));
            contribution = mk3(a4.x, a4.y, a4.z) * (radianceOut / CosineHemispherePdf(L.z));
        } else {
            // chsMain's light loop body for this (vertex, light): GetDirectRadiance with lightVisibility = 1 (lighting/common.hlsli:8-143),
            // times the path attenuation -- evaluated HERE, one lane per ray, instead of once more per cell light in accumulate
            const uint32_t li = P.gridCellLights[__float_as_uint(v0.w) + slot];
            incr_light light; float lm[16];
            fetch_light(li, light, lm);
            const float4 n4 = s_nrm[lo], c4 = s_col[lo], a4 = s_att[lo]; const float2 vyz2 = make_float2(s_sky[lo].w, s_skyRad[lo].w);
            const f3 N = {n4.x, n4.y, n4.z};
            const LightSample ls = light_setup(light, lm, true, origin, N);
            L = ls.L;
            const f3 V = -mk3(a4.w, vyz2.x, vyz2.y);
            const f3 radianceOut = light_shade(ls, mk3(c4.x, c4.y, c4.z), c4.w, n4.w, N, V);
            contribution = mk3(a4.x, a4.y, a4.z) * radianceOut;
        }
        if (!answered) {
            const size_t q = (size_t)baseQ + dest - answered_before(lo) - ((ansOfOwner != 0xffu && slot > ansOfOwner) ? 1u : 0u);
            B.rayQ[q * 2] = make_float4(origin.x, origin.y, origin.z, __uint_as_float(vp));
            B.rayQ[q * 2 + 1] = make_float4(L.x, L.y, L.z, __uint_as_float(base + r));      // .w: the ray's record
        }
        B.rayC[base + r] = make_float4(contribution.x, contribution.y, contribution.z, 1.f);      // .w: 1 = unoccluded until the shadow kernel says otherwise
        answeredRays += answered ? 1u : 0u;
    }
    if (P.cover.header) {                                         // kernel-uniform
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) answeredRays += __shfl_down(answeredRays, off, 64);
        // one word of the ray statistics' stripes (word 5 of 16; the first five are incr_ray_stats): an atomic per wave on ONE counter word was 355 K same-address
        // atomics a frame, +2.7 ms on the shade kernel
        if (lane == 0 && answeredRays) {
            if (P.statStripes) atomicAdd(P.statStripes + (blockIdx.x & 63u) * 16u + 5u, (unsigned long long)answeredRays);
            else atomicAdd(B.counters + kAnsweredBase, answeredRays);
        }
    }
    return out;
}

// ------------------------------------------------------------------------------------------------ shade
// ENV: the caller passed the sky cube (P.envmap != NULL: bench, the C++ renderer) -- the kernel then carries no Preetham code at all.  The analytic sky's ~35 scalar
// constants, pre-loaded and spilled to vector-register lanes, cost the kernel 4 % even where no lane ever evaluated it (round 4: SGPR spills 197 -> 134, shade 4.43 ->
// 4.24 ms per frame alone with the miss shading moved to wf_accumulate_kernel as well).
#ifndef INCR_SHADE_LOOP
#define INCR_SHADE_LOOP 0
#endif
template <bool ENV> __global__ __launch_bounds__(kWfBlock, INCR_SHADE_WAVES) void wf_shade_kernel(const PtParams P, const WfBuffers B, int bounce, int fuseShadowGen) {
    __shared__ uint32_t s_waveCount[kWfBlock / 64];
    __shared__ uint32_t s_base;
    // fuseShadowGen: the shadow-ray set-up of this workgroup's vertices (shadow_gen_block) runs here, on the values still in registers,
    // (0 only for the debug views, which make no vertices)
    __shared__ ShadowGenLds s_gen;
    if (fuseShadowGen) shadow_gen_load_lights(P, s_gen);
    Counters cnt = {0, 0, 0, 0, 0};
    const uint32_t count = B.counters[bounce];
    const float4* __restrict__ extIn = (bounce & 1) ? B.extB : B.extA;
    float4* extOut = (bounce & 1) ? B.extA : B.extB;
    const float4* __restrict__ stIn = (bounce & 1) ? B.stB : B.stA;
    float4* stOut = (bounce & 1) ? B.stA : B.stB;
    uint32_t* liveCounter = B.counters + bounce + 1;
    // one pass per workgroup: the launcher's grid covers every path of the pipeline (blocksAll).  A grid-stride loop here has the compiler hoist
    // ~140 loop-invariant scalars (addresses, launch constants) out of it, which then live in spilled SGPRs for the whole kernel
#if INCR_SHADE_LOOP
    for (uint32_t base = blockIdx.x * kWfBlock; base < count; base += gridDim.x * kWfBlock) {
#else
    const uint32_t base = blockIdx.x * kWfBlock;
    if (base < count) {
#endif
        const uint32_t i = base + threadIdx.x;
        bool cont = false, haveVertex = false;
        ShadowGenVertex SV = {};
        uint32_t pid = 0, vflags = 0u, vcell = 0u;
        float4 nextO = make_float4(0.f, 0.f, 0.f, 0.f), nextD = nextO, nextA = nextO;
        f4 color = {0.f, 0.f, 0.f, 0.f};
        if (i < count) {
            const float4 ro = extIn[(size_t)i * 2], rd = extIn[(size_t)i * 2 + 1];
            pid = __float_as_uint(ro.w);
            const uint32_t smp = P.numSamples > 1u ? fast_div(pid, P.divLocalPixels) : 0u, lp = pid - smp * P.localPixels;      // path id = sample * localPixels + local pixel
            const uint32_t ly = fast_div(lp, P.divW), px = lp - ly * P.W;
            const uint32_t pixelIdx = (uint32_t)shard_global_row(P, ly) * P.W + px, sampleIdx = P.sampleIdx + smp;
            const uint4 h = B.hit[i];
            // rgsMain's payload (pathtracing.hlsl:110-114): colour 0, attenuation 1.  The colour does not travel through this kernel any more: what a
            // vertex adds to it goes into its record and wf_accumulate_kernel -- deferred to the end of the frame -- does every addition in chsMain's order
            float4 a4 = make_float4(1.f, 1.f, 1.f, 0.f);
            if (bounce > 0) a4 = stIn[(size_t)i * 2 + 1];
            const f3 rayD = {rd.x, rd.y, rd.z};
            color = {0.f, 0.f, 0.f, 0.f};
            f3 attenuation = {a4.x, a4.y, a4.z};
            float4* vtx = vertex_of(B, bounce, i);
            if (P.hitChain && h.x == kMissTri) { uint32_t* hc = P.hitChain + (size_t)lp * 12u + 3u * (uint32_t)bounce; hc[0] = 0xffffffffu; hc[1] = 0xffffffffu; hc[2] = 0u; }
            if (h.x == kMissTri) {   // msEnvmap / msDynamicSky -- pathtracing.hlsl:304-319: evaluated by wf_accumulate_kernel (the analytic sky stays out of this kernel)
                vtx[kVColor] = make_float4(attenuation.x, attenuation.y, attenuation.z, __uint_as_float(0xffffffffu));
                vtx[kVMask] = make_float4(rayD.x, rayD.y, rayD.z, 0.f);
                *(uint4*)(vtx + kVMeta) = make_uint4(kFlagMiss | kFlagTerminate, pid, 0u, 0u);
            } else {
                const f3 rayO = {ro.x, ro.y, ro.z};
                const float hitT = __uint_as_float(h.y), bx = __uint_as_float(h.z), by = __uint_as_float(h.w);
                // chsMain -- pathtracing.hlsl:121-268
                cnt.hits++;
                const int pathLength = bounce + 1;
                // one SamplePoint per earlier hit (Q3) -- except the mirror vertex of VIEWMODE 7 (below), which consumes none
                uint32_t sampleSetIndex = (uint32_t)bounce - ((P.viewmode == 7 && bounce > 0) ? 1u : 0u);
                const float4* __restrict__ sp = (const float4*)(P.shade + h.x);
                const float4 s0 = sp[0], s1 = sp[1], s2 = sp[2], s3 = sp[3], s4 = sp[4], s5 = sp[5], s6 = sp[6];
                const uint4 s7 = *(const uint4*)(sp + 7);
                const f3 vn0 = {s0.x, s0.y, s0.z}, vn1 = {s0.w, s1.x, s1.y}, vn2 = {s1.z, s1.w, s2.x};
                const f4 vt0 = {s2.y, s2.z, s2.w, s3.x}, vt1 = {s3.y, s3.z, s3.w, s4.x}, vt2 = {s4.y, s4.z, s4.w, s5.x};
                const f2 uv0 = {s5.y, s5.z}, uv1 = {s5.w, s6.x}, uv2 = {s6.y, s6.z};
                const uint32_t materialIndex = __float_as_uint(s6.w), meshIndex = s7.x;
                if (P.hitChain) { uint32_t* hc = P.hitChain + (size_t)lp * 12u + 3u * (uint32_t)bounce; hc[0] = s7.y << 16 | (s7.z & 0xffffu); hc[1] = s7.w; }      // InstanceIndex, GeometryIndex, PrimitiveIndex (word 2: wf_accumulate_kernel)
                const float* __restrict__ localToWorld = P.meshLocalToWorld + (size_t)meshIndex * 16;
                const f3 hitPosition = rayO + hitT * rayD;
                const f4 packedT = vt0 + bx * (vt1 - vt0) + by * (vt2 - vt0);
                f3 N = normalize(vn0 + bx * (vn1 - vn0) + by * (vn2 - vn0));
                f3 T = normalize(mk3(packedT.x, packedT.y, packedT.z));
                f3 Bt = cross(N, T) * packedT.w;
                N = mul_dir(N, localToWorld);
                T = mul_dir(T, localToWorld);
                Bt = mul_dir(Bt, localToWorld);
                const f3 Ntbn = N;
                const f2 uv = uv0 + bx * (uv1 - uv0) + by * (uv2 - uv0);
                const MaterialTexels mtx = P.materialTexels[materialIndex];
                MatProps matInfo = mtx.texels ? EvaluateMaterialPropertiesInterleaved(P.materials[materialIndex], mtx, P.textures, P.srgbLut, uv)
                                              : EvaluateMaterialProperties(P.materials[materialIndex], P.textures, P.srgbLut, uv);
                bool done = false;
                if (P.viewmode == 1) matInfo.basecolor = mk3(0.5f);
                else if (P.viewmode == 2) { color = {matInfo.roughness, matInfo.roughness, matInfo.roughness, matInfo.roughness}; done = true; }
                else if (P.viewmode == 3) { color = {matInfo.metallic, matInfo.metallic, matInfo.metallic, matInfo.metallic}; done = true; }
                else if (P.viewmode == 4) { color = {matInfo.basecolor.x, matInfo.basecolor.y, matInfo.basecolor.z, 1.f}; done = true; }
                else if (P.viewmode == 5) { color = {matInfo.emissive.x, matInfo.emissive.y, matInfo.emissive.z, 1.f}; done = true; }
                if (done) {
                    store_pixel(P, px, ly, smp, color);
                    vtx[kVMeta] = make_float4(__uint_as_float(kFlagSkip), 0.f, 0.f, 0.f);
                } else {
                    const f3 add = attenuation * matInfo.emissive * 20000;
                    color = {add.x, add.y, add.z, 0.f};                      // payload.color.xyz += add happens in wf_accumulate_kernel, before the vertex's light loop
                    attenuation = attenuation * saturate(1.f - matInfo.emissive);
                    if (matInfo.bHasNormalmap) {
                        const f3 nm = matInfo.normalmap;
                        N = normalize(mk3((nm.x * T.x + nm.y * Bt.x) + nm.z * N.x, (nm.x * T.y + nm.y * Bt.y) + nm.z * N.y, (nm.x * T.z + nm.y * Bt.z) + nm.z * N.z));
                    }
                    const f3 V = -rayD;
                    const f3 attenV = attenuation;
                    if (P.viewmode == 7 && pathLength < 2) {
                        // VIEWMODE 7, Reflections (pathtracing.hlsl:212-224): the first hit sends one mirror ray with the same payload and returns --
                        // no lights, no sky sample, no attenuation test.  The vertex carries an empty request mask: accumulate hands the colour on.
                        const f3 Rm = normalize(reflect(-V, N));
                        nextO = make_float4(hitPosition.x, hitPosition.y, hitPosition.z, __uint_as_float(pid));
                        nextD = make_float4(Rm.x, Rm.y, Rm.z, 0.f);
                        nextA = make_float4(attenuation.x, attenuation.y, attenuation.z, 0.f);
                        cont = true; haveVertex = true; vflags = 0u; vcell = 0u;
                    } else {
                    // ---- enumerate the shadow rays of this vertex as a request mask ----
                    uint32_t req3 = 0u;
                    uint32_t flags = 0u;
                    const bool finiteAtt = fabsf(attenV.x) < 3.0e38f && fabsf(attenV.y) < 3.0e38f && fabsf(attenV.z) < 3.0e38f;
                    const uint32_t numCells = P.gridDim[0] * P.gridDim[1] * P.gridDim[2];
                    uint32_t cell = light_cell(P, hitPosition);
                    if (!finiteAtt) cell = numCells;                      // the catch-all cell lists every light
                    // (which lights of the cell need a shadow ray -- chsMain's loop over the lights, pathtracing.hlsl:230-236 -- is decided by
                    // shadow_gen_block with one lane per (vertex, light) pair: inside the per-hit code the loop ran to the longest light list
                    // of the wave at 3 waves per SIMD and cost 2.6 of its 6.7 ms per frame)
                    // sky sample :239-244
                    const f2 us = SamplePoint(pixelIdx, sampleIdx, sampleSetIndex, P.divSqrtN, P.divN, P.numPixels);
                    const f3 skyDir = CosineSampleHemisphere(us);
                    const f3 skyRadianceIn = P.skyBrightness * (ENV ? cube_sample_level(CubeChain{P.envmapRes, P.envmapMips, P.envmap}, skyDir, 0u) : sky_env_radiance(P, skyDir));
                    const float skyNoL = saturate(dot(N, skyDir));
                    if (skyNoL > 0.f && (skyRadianceIn.x > 0.f || skyRadianceIn.y > 0.f || skyRadianceIn.z > 0.f)) { req3 |= 0x80000000u; cnt.skyShadow++; }
                    // ---- BRDF-sampled continuation (raytracing/common.hlsli:55-123) ----
                    if (pathLength < 4) {
                        uint32_t setCopy = sampleSetIndex;
                        f2 u = SamplePoint(pixelIdx, sampleIdx, setCopy, P.divSqrtN, P.divN, P.numPixels);
                        f3 L, H;
                        const float t = 0.5f;
                        if (u.x < t) {
                            u.x = 2.f * u.x;
                            const f3 l = CosineSampleHemisphere(u);
                            L = normalize(mk3((l.x * T.x + l.y * Bt.x) + l.z * Ntbn.x, (l.x * T.y + l.y * Bt.y) + l.z * Ntbn.y, (l.x * T.z + l.y * Bt.z) + l.z * Ntbn.z));
                            H = normalize(L + V);
                        } else {
                            u.x = 2.f * (u.x - 0.5f);
                            const f3 hh = SampleGGX(u, matInfo.roughness);
                            H = normalize(mk3((hh.x * T.x + hh.y * Bt.x) + hh.z * Ntbn.x, (hh.x * T.y + hh.y * Bt.y) + hh.z * Ntbn.y, (hh.x * T.z + hh.y * Bt.z) + hh.z * Ntbn.z));
                            L = normalize(reflect(-V, H));
                        }
                        const float NoL = hmax(dot(N, L), 0.001f);
                        const float NoV = hmax(dot(N, V), 0.001f);
                        const float NoH = hmax(dot(N, H), 0.001f);
                        const float VoH = hmax(dot(V, H), 0.001f);
                        const float pdf = t * CosineHemispherePdf(NoL) + QDIV((1.f - t) * GGXPdf(NoH, matInfo.roughness), 4.f * VoH);
                        const f3 F0 = matInfo.metallic * matInfo.basecolor + (1.f - matInfo.metallic) * 0.04f;
                        const f3 albedo = (1.f - matInfo.metallic) * matInfo.basecolor;
                        const float D = GGX(NoH, matInfo.roughness);
                        const f3 F = F_Schlick(VoH, F0);
                        const float G = G_Smith_Direct(NoV, NoL, matInfo.roughness);
                        const f3 Fd = albedo * K_INV_PI;
                        const f3 Fr = (D * F * G) / hmax(4.f * NoV * NoL, 0.001f);
                        const f3 brdf = (Fr + (1.f - F) * Fd);
                        const f3 outAttenuation = brdf * NoL / pdf;
                        attenuation = attenuation * outAttenuation;
                        if (attenuation.x != 0.f || attenuation.y != 0.f || attenuation.z != 0.f) {     // Q2
                            nextO = make_float4(hitPosition.x, hitPosition.y, hitPosition.z, __uint_as_float(pid));
                            nextD = make_float4(L.x, L.y, L.z, 0.f);
                            nextA = make_float4(attenuation.x, attenuation.y, attenuation.z, 0.f);
                            cont = true;
                        }
                    }
                    if (!cont) flags |= kFlagTerminate;
                    haveVertex = true;
                    // the vertex: its fields stay in registers for the shadow-ray set-up below; the record gets them there
                    vflags = flags; vcell = cell;
                    if (!finiteAtt || !(CosineHemispherePdf(skyDir.z) > 0.f)) vflags |= kFlagFullRecord;      // the sky term is not "+ 0" when unrequested or occluded: accumulate re-evaluates it
                    SV.have = true; SV.req = make_uint4(0u, 0u, 0u, req3);
                    if (P.directLighting && P.lightCount > 0) { SV.cellLo = P.gridCellOffsets[cell]; SV.cellHi = P.gridCellOffsets[cell + 1]; }
                    SV.pos = make_float4(hitPosition.x, hitPosition.y, hitPosition.z, __uint_as_float(SV.cellLo));
                    SV.sky = make_float4(skyDir.x, skyDir.y, skyDir.z, 0.f); SV.nrm = make_float4(N.x, N.y, N.z, matInfo.roughness);
                    SV.skyRad = make_float4(skyRadianceIn.x, skyRadianceIn.y, skyRadianceIn.z, 0.f);
                    SV.col = make_float4(matInfo.basecolor.x, matInfo.basecolor.y, matInfo.basecolor.z, matInfo.metallic);
                    SV.att = make_float4(attenV.x, attenV.y, attenV.z, rayD.x); SV.vyz = make_float2(rayD.y, rayD.z);
                    }
                }
            }
        }
        // (Ordering the block's continuation rays by direction octant -- a wave of the next extend launch then holds one or two octants
        // instead of eight -- was measured and dropped: extend 11.9 -> 11.8 ms per frame, the next bounce's shade 6.65 -> 6.9 ms.)
        ShadowGenResult sg = {SV.req, 0u, 0u};
        if (fuseShadowGen) sg = shadow_gen_block(P, B, bounce, s_gen, i, SV);
        if (haveVertex) {
            float4* vtx = vertex_of(B, bounce, i);
            const uint32_t f = vflags | sg.flags;
            *(uint4*)(vtx + kVMask) = sg.req;
            *(uint4*)(vtx + kVMeta) = make_uint4(f, pid, sg.rayBase, vcell);
            if (!fuseShadowGen || (f & (kFlagInlineShadow | kFlagManyLights | kFlagFullRecord))) {       // someone re-evaluates this vertex: the rest of the record
                float4* full = vertex_full_of(B, bounce, i);
                full[kFPos] = make_float4(SV.pos.x, SV.pos.y, SV.pos.z, SV.col.w);
                full[kFNrm] = SV.nrm;
                full[kFBase] = make_float4(SV.col.x, SV.col.y, SV.col.z, SV.att.w);
                full[kFAtt] = make_float4(SV.att.x, SV.att.y, SV.att.z, SV.vyz.x);
                full[kFSky] = make_float4(SV.sky.x, SV.sky.y, SV.sky.z, SV.vyz.y);
            }
        }
        const uint32_t slot = block_append(cont, liveCounter, s_waveCount, &s_base);
        // the continuation ray and its attenuation go to the next bounce's list; its colour is written there by accumulate(bounce), which runs
        // after the whole path has been traced
        if (cont) { extOut[(size_t)slot * 2] = nextO; extOut[(size_t)slot * 2 + 1] = nextD; stOut[(size_t)slot * 2 + 1] = nextA; }
        if (haveVertex) vertex_of(B, bounce, i)[kVColor] = make_float4(color.x, color.y, color.z, __uint_as_float(cont ? slot : 0xffffffffu));
    }
    add_counters(P, cnt);
}

// ------------------------------------------------------------------------------------------------ shadow rays
// Round 1 gave every lane of the shadow kernel one path VERTEX and let it trace that vertex's requested rays one after
// the other, the ~200-instruction light / ray set-up of each ray included: set-up ran for a handful of lanes on nearly
// every iteration of the traversal loop (34 % VALU lane utilisation).  Now the set-up is its own streaming kernel with
// one lane per RAY and the traversal kernel reads a flat, coalesced ray queue (32 B per ray: origin + path id, direction
// + request slot); rays of one vertex are neighbours in the queue, so neighbouring lanes leave from the same point.
//
// Queue capacity is a soft limit: ONE queue holds the shadow rays of a path's four bounces, rayCap = paths x rays per path, where the rays per path come
// from the scene (api.hip wf_rays_per_path: 4 vertices x (mean lights per occupied light-grid cell + the sky sample) + 4, at least 12: twice what the bench
// scene's paths ask for).  A workgroup whose rays do not fit flags its vertices kFlagInlineShadow and accumulate traces those requests itself -- slower, same
// result, nothing is dropped; wf_overflow_tally_kernel (below) counts such rays when a profile is being recorded, incr_pathtrace_profile reports them.

// Pure any-hit traversal over the ray queue: idle lanes take the next rays of the wave's chunk; an occluded ray zeroes the
// visibility word of its own record (a plain 4-byte store; unoccluded rays -- the majority of the light rays -- write nothing).
// The loop is software-pipelined around the wave's one blocking wait per iteration (the traversal fetch): a refill only
// ISSUES its two queue loads, the step of the lanes still tracing runs, and the new rays start after it -- their HBM
// latency hides behind the step's fetch instead of stalling 40 busy lanes; results of finished lanes are written by the
// next refill, all of them with one instruction.
// `launch` = the index of this launch within the frame (its cursor); it traces the rays of bounces loBounce .. hiBounce = the queue range between the
// snapshots [kSnapBase + loBounce, kSnapBase + hiBounce + 1).
template <int N> __global__ __launch_bounds__((TravCfg<N, true>::kBlock), (TravCfg<N, true>::kWaves)) void wf_shadow_kernel(const PtParams P, const WfBuffers B, int launch, int loBounce, int hiBounce) {
    typedef TravCfg<N, true> Cfg;
    __shared__ uint32_t s_stack[N * Cfg::kBlock];
    __shared__ float4 s_top[Cfg::kTop * 5];
    const uint32_t first = min(B.counters[kSnapBase + loBounce], B.rayCap);
    const uint32_t total = min(B.counters[kSnapBase + hiBounce + 1], B.rayCap) - first;
    if (blockIdx.x * (Cfg::kBlock / 64) * kShadowChunk >= total) return;          // no static chunk for any wave of this workgroup (see wf_extend_kernel)
    TravStack<N, true> st; st.bind(s_stack + threadIdx.x, nullptr); st.sp = 0;
    st.bind_top(s_top, load_top_nodes<Cfg::kBlock>(P, s_top, Cfg::kTop));
#if INCR_WF_CURSOR_STRIPES > 1
    uint32_t* cursor = B.counters + kStripeBase + (4 + launch) * INCR_WF_CURSOR_STRIPES * 32;
#else
    uint32_t* cursor = B.counters + kCursorBase + 2 * launch + 1;
#endif
    WaveChunk<kShadowChunk, Cfg::kBlock> chunk; chunk.init_static(total);
    bool active = false, pending = false, occludedDone = false;
    uint32_t rayIndex = 0;
    float4 qa = make_float4(0.f, 0.f, 0.f, 0.f), qb = qa;
    TravAny2 T; trav_any_begin(T, {0.f, 0.f, 0.f}, {0.f, 0.f, 1.f}, 0.1f, 1000.f); T.triBase1 = 0; T.triMask1 = 0;      // the interval every ray of this kernel has: tmin / tmax stay compile-time constants
    LANE_STATS(LaneStats ls;)
    for (;;) {
        const unsigned long long act = __ballot(active);
        const uint32_t idle = 64u - (uint32_t)__popcll(act);
        if (!chunk.exhausted && (idle >= kShadowRefill || act == 0ull)) {
            LANE_STATS(ls.refills++;)
            if (occludedDone) { ((float*)(B.rayC + rayIndex))[3] = 0.f; occludedDone = false; }          // rayIndex = the ray's RECORD (direction.w of its queue entry)
            const uint32_t item = chunk.take(!active, cursor, total);
            if (item != 0xffffffffu) { qa = B.rayQ[(size_t)(first + item) * 2]; qb = B.rayQ[(size_t)(first + item) * 2 + 1]; pending = true; }
        }
        LANE_STATS(ls.step(active, T.haveCur && T.triMask1 == 0u, T.triMask != 0u, chunk.exhausted);)
        if (active && trav_any_step2(P, st, T)) { occludedDone = T.occluded; active = false; }      // node step and triangle step in one iteration
        if (pending) {
            pending = false;
            rayIndex = __float_as_uint(qb.w);
            if (__float_as_uint(qa.w) != 0xffffffffu) {
                trav_any_begin(T, {qa.x, qa.y, qa.z}, {qb.x, qb.y, qb.z}, 0.1f, 1000.f);     // lighting/common.hlsli:96-100
                T.haveCur = true; T.triBase1 = 0; T.triMask1 = 0;
                st.reset();
                active = true;
            }
        }
        if (__ballot(active) == 0ull && chunk.exhausted) break;
    }
    if (occludedDone) ((float*)(B.rayC + rayIndex))[3] = 0.f;
    LANE_STATS(ls.flush(1);)
}

// ------------------------------------------------------------------------------------------------ accumulate
__global__ __launch_bounds__(kWfBlock, INCR_ACCUM_WAVES) void wf_accumulate_kernel(const PtParams P, const WfBuffers B, int bounce) {
    const uint32_t count = B.counters[bounce];
    const float4* __restrict__ stIn = (bounce & 1) ? B.stB : B.stA;
    float4* stOut = (bounce & 1) ? B.stA : B.stB;
    // one trip (the grid covers every slot, as in wf_shade_kernel): `i = ~0u` ends the loop after its first pass and lets the body keep its `continue`s
    for (uint32_t i = blockIdx.x * kWfBlock + threadIdx.x; i < count; i = INCR_SHADE_LOOP ? i + gridDim.x * kWfBlock : 0xffffffffu) {
        const float4* vtx = vertex_of(B, bounce, i);
        const uint4 meta = *(const uint4*)(vtx + kVMeta);
        const uint32_t flags = meta.x;
        if (flags & kFlagSkip) continue;
        const float4 v6 = vtx[kVColor];
        // payload.color as the path brings it to this vertex (rgsMain: 0; later: what accumulate(bounce - 1) left for this slot), plus what the vertex
        // adds before its light loop -- chsMain's emissive term (pathtracing.hlsl:200), or the miss shader's sky term, after which rgsMain stores the pixel
        float4 cin = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bounce > 0) cin = stIn[(size_t)i * 2];
        if (flags & kFlagMiss) {          // the miss shader's term (msEnvmap / msDynamicSky, pathtracing.hlsl:304-319) on the attenuation and direction the ray arrived with
            const uint32_t mpid = meta.y;
            const uint32_t smp = P.numSamples > 1u ? fast_div(mpid, P.divLocalPixels) : 0u, lp = mpid - smp * P.localPixels;
            const uint32_t ly = fast_div(lp, P.divW);
            const float4 md = vtx[kVMask];
            f3 add;
            if (P.envSkyMode == 0) add = sky_miss_add(P, mk3(v6.x, v6.y, v6.z), mk3(md.x, md.y, md.z));
            else {          // sky_miss_add's msDynamicSky arm on the constants wf_init_counters_kernel left in the workspace
                typedef const __attribute__((address_space(4))) float* ConstFloats;
                const ConstFloats sc = (ConstFloats)(uintptr_t)(B.counters + kSkyConstBase);
                WfSkyConsts k;
#pragma unroll
                for (uint32_t w = 0; w < sizeof(WfSkyConsts) / 4u; ++w) ((float*)&k)[w] = sc[w];
                add = mk3(v6.x, v6.y, v6.z) * 1000 * CalculateSkyRadianceRGB(k.sunMiss, mk3(md.x, md.y, md.z), k.perez);
            }
            store_pixel(P, lp - ly * P.W, ly, smp, {cin.x + add.x, cin.y + add.y, cin.z + add.z, cin.w});
            continue;
        }
        const uint4 vis = *(const uint4*)(vtx + kVMask);
        const uint32_t pid = meta.y, rayBase = meta.z, cell = meta.w, nextSlot = __float_as_uint(v6.w);
        f4 color = {cin.x + v6.x, cin.y + v6.y, cin.z + v6.z, 0.f};          // payload.color.w is 0 on every path that reaches a light loop
        // the vertex's queued rays: [rayBase, rayBase + popc(request mask)) in ascending slot order, the sky ray (slot 127) last;
        // each record carries attenuation * GetDirectRadiance of its light -- for the sky ray attenuation * GetSkyRadiance / pdf
        // (shadow_gen_block) -- and the visibility the shadow kernel left
        const uint32_t requested = (uint32_t)(__popc(vis.x) + __popc(vis.y) + __popc(vis.z) + __popc(vis.w));
        const bool skyRequested = vis.w >> 31;
        bool skyVisible = skyRequested;
        uint32_t visWord = 0u;        // debug_hit_chain word 2: which of the vertex's shadow queries came back unoccluded (sum of (loop index + 1) * 0x9E3779B1)
        const bool perVertex = flags & (kFlagInlineShadow | kFlagManyLights | kFlagFullRecord);
        if (P.hitChain && !(flags & (kFlagInlineShadow | kFlagManyLights)) && P.directLighting && P.lightCount > 0) {
            const uint32_t lo = P.gridCellOffsets[cell];
            uint32_t rank = 0;
            for (uint32_t k = 0; k < kSkySlot; ++k) {
                const uint32_t w = k < 32u ? vis.x : (k < 64u ? vis.y : (k < 96u ? vis.z : vis.w));
                if (!((w >> (k & 31u)) & 1u)) continue;
                if (B.rayC[rayBase + rank].w != 0.f) visWord += (P.gridCellLights[lo + k] + 1u) * 0x9E3779B1u;
                ++rank;
            }
        }
        if (!perVertex) {
            // chsMain's light loop (pathtracing.hlsl:230-236) and sky sample (:239-244) in their order: lights that asked for no ray add
            // exactly attenuation * 0, occluded ones too, and so does a sky sample that is unrequested or occluded (finite attenuation and
            // pdf > 0: wf_shade_kernel flags the vertices where that does not hold); the others add the value computed for their ray
            const uint32_t lights = requested - (skyRequested ? 1u : 0u);
            for (uint32_t r = 0; r < lights; ++r) {
                const float4 c = B.rayC[rayBase + r];
                if (c.w != 0.f) { color.x += c.x; color.y += c.y; color.z += c.z; }
            }
            if (skyRequested) {
                const float4 c = B.rayC[rayBase + lights];
                skyVisible = c.w != 0.f;
                if (skyVisible) { color.x += c.x; color.y += c.y; color.z += c.z; }
            }
        } else {
            // per-vertex evaluation out of the full record: the ray queue was full for this vertex (every request is traced here), its cell
            // lists more than 127 lights (the first 127 went through the queue, the rest is traced here), or "+ 0" is not exact for it
            const bool inlineShadow = flags & kFlagInlineShadow;
            const float4* full = vertex_full_of(B, bounce, i);
            const float4 v0 = full[kFPos], v1 = full[kFNrm], vb = full[kFBase], va = full[kFAtt], vs = full[kFSky];
            const f3 hitPosition = {v0.x, v0.y, v0.z}, N = {v1.x, v1.y, v1.z}, basecolor = {vb.x, vb.y, vb.z}, attenV = {va.x, va.y, va.z};
            const f3 skyDir = {vs.x, vs.y, vs.z};
            const f3 V = -mk3(vb.w, va.w, vs.w);
            MatProps matInfo = {};
            matInfo.basecolor = basecolor; matInfo.metallic = v0.w; matInfo.roughness = v1.w;
            if (P.directLighting && P.lightCount > 0) {
                const uint32_t lo = P.gridCellOffsets[cell], hi = P.gridCellOffsets[cell + 1];
                const uint32_t n = hi - lo;
                uint32_t rank = 0;
                if (flags & (kFlagInlineShadow | kFlagManyLights)) visWord = 0u;
                for (uint32_t k = 0; k < n; ++k) {
                    incr_light light; const float* m; bool haveT;
                    load_light(P, (int)P.gridCellLights[lo + k], light, m, haveT);
                    const LightSample s = light_setup(light, m, haveT, hitPosition, N);
                    f3 radianceOut = {0.f, 0.f, 0.f};
                    bool visible = false;
                    if (k < kSkySlot && !inlineShadow) {          // selects, not words[k >> 5]: a dynamically indexed array would live in scratch
                        const uint32_t w = k < 32u ? vis.x : (k < 64u ? vis.y : (k < 96u ? vis.z : vis.w));
                        if ((w >> (k & 31u)) & 1u) { visible = B.rayC[rayBase + rank].w != 0.f; ++rank; }
                    } else if (s.needsShadow) { PrivStackT<kStackLds> ps; ps.sp = 0; visible = !trace_any(P, ps, hitPosition, s.L, 0.1f, 1000.f); }
                    if ((flags & (kFlagInlineShadow | kFlagManyLights)) && s.needsShadow && visible) visWord += (P.gridCellLights[lo + k] + 1u) * 0x9E3779B1u;
                    if (s.needsShadow && visible) radianceOut = light_shade(s, matInfo.basecolor, matInfo.metallic, matInfo.roughness, N, V);
                    const f3 r = attenV * radianceOut;
                    color.x += r.x; color.y += r.y; color.z += r.z;
                }
            }
            if (skyRequested) {
                if (inlineShadow) { PrivStackT<kStackLds> ps; ps.sp = 0; skyVisible = !trace_any(P, ps, hitPosition, skyDir, 0.1f, 1000.f); }
                else skyVisible = B.rayC[rayBase + requested - 1u].w != 0.f;
            }
            // sky sample (pathtracing.hlsl:239-244 + lighting/common.hlsli:145-202)
            const f3 skyRadianceIn = P.skyBrightness * sky_env_radiance(P, skyDir);
            const float skyNoL = saturate(dot(N, skyDir));
            f3 radianceOut = {0.f, 0.f, 0.f};
            if (skyVisible) radianceOut = sky_shade(skyRadianceIn, skyDir, skyNoL, matInfo, N, V);
            const f3 r = attenV * (radianceOut / CosineHemispherePdf(skyDir.z));
            color.x += r.x; color.y += r.y; color.z += r.z;
        }
        if (P.hitChain) {
            const uint32_t smpc = P.numSamples > 1u ? fast_div(pid, P.divLocalPixels) : 0u;
            P.hitChain[(size_t)(pid - smpc * P.localPixels) * 12u + 3u * (uint32_t)bounce + 2u] = (visWord & 0x7fffffffu) | (skyVisible ? 0x80000000u : 0u);
        }
        if (flags & kFlagTerminate) {
            const uint32_t smp = P.numSamples > 1u ? fast_div(pid, P.divLocalPixels) : 0u, lp = pid - smp * P.localPixels;
            const uint32_t ly = fast_div(lp, P.divW);
            store_pixel(P, lp - ly * P.W, ly, smp, color);
        }
        else stOut[(size_t)nextSlot * 2] = make_float4(color.x, color.y, color.z, color.w);
    }
}

// ------------------------------------------------------------------------------------------------ samples in flight: integrate
// incr_pathtrace_desc::num_samples > 1: every finished path left its colour (RGBA16F bits, what destUav would hold) in
// sampleStage[sample][pixel]; one lane per pixel replays pathtrace-integrate.hlsl:16-31 for its samples IN SAMPLE ORDER, each step
// re-quantised to R11G11B10 in a register as the history surface would have been in memory: the history equals the one that
// num_samples successive calls leave, bit for bit.  The target keeps the last sample.
__global__ __launch_bounds__(kWfBlock) void wf_integrate_samples_kernel(const PtParams P) {
    const uint32_t li = blockIdx.x * kWfBlock + threadIdx.x;
    if (li >= P.localPixels) return;
    uint32_t packed = (P.history && P.sampleIdx > 1u) ? P.history[li] : 0u;
    uint2 last = make_uint2(0u, 0u);
    for (uint32_t s = 0; s < P.numSamples; ++s) {
        last = P.sampleStage[(size_t)s * P.localPixels + li];
        const f3 cur = {f16_bits_to_f32(last.x & 0xffffu), f16_bits_to_f32(last.x >> 16), f16_bits_to_f32(last.y & 0xffffu)};
        packed = pack_r11g11b10(integrate_sample(P, unpack_r11g11b10(packed), cur, P.sampleIdx + s), P.storageRounding);
    }
    if (P.target) *(uint2*)(P.target + (size_t)li * 4) = last;
    if (P.history) P.history[li] = packed;
}

} // namespace incr

// ---- launcher ------------------------------------------------------------------------------------------
namespace incr {
__global__ __launch_bounds__(64) void wf_init_counters_kernel(uint32_t* counters, uint32_t extendStart, uint32_t shadowStart, const WfSkyConsts sky) {
    const uint32_t i = threadIdx.x;
    if (i < sizeof(WfSkyConsts) / 4u) counters[kSkyConstBase + i] = ((const uint32_t*)&sky)[i];
    uint32_t v = 0u;
    if (i >= (uint32_t)kCursorBase && i < (uint32_t)kCursorBase + 8u) v = ((i - kCursorBase) & 1u) ? shadowStart : extendStart;   // behind the static first round
    counters[i] = v;
    if (i == 0u) counters[kQueueCountBase] = 0u;
#if INCR_WF_CURSOR_STRIPES > 1
    for (uint32_t k = i; k < 8u * INCR_WF_CURSOR_STRIPES; k += 64u) counters[kStripeBase + k * 32u] = 0u;
#endif
}
template <int N> static void launch_extend(uint32_t blocks, hipStream_t stream, const PtParams& p, const WfBuffers& b, int bounce, uint32_t chunk) {
    hipLaunchKernelGGL(wf_extend_kernel<N>, dim3(blocks), dim3(TravCfg<N, false>::kBlock), 0, stream, p, b, bounce, chunk);
}
template <int N> static void launch_shadow(uint32_t blocks, hipStream_t stream, const PtParams& p, const WfBuffers& b, int launch, int lo, int hi) {
    hipLaunchKernelGGL(wf_shadow_kernel<N>, dim3(blocks), dim3(TravCfg<N, true>::kBlock), 0, stream, p, b, launch, lo, hi);
}
// workgroups per CU and threads per workgroup of the traversal kernels for a stack of `sv` entries (TravCfg)
template <bool SHADOW> static void trav_geometry(int sv, uint32_t* perCU, uint32_t* block) {
    if (sv == 8) { *perCU = TravCfg<8, SHADOW>::kBlocksPerCU; *block = TravCfg<8, SHADOW>::kBlock; }
    else if (sv == 10) { *perCU = TravCfg<10, SHADOW>::kBlocksPerCU; *block = TravCfg<10, SHADOW>::kBlock; }
    else if (sv == 12) { *perCU = TravCfg<12, SHADOW>::kBlocksPerCU; *block = TravCfg<12, SHADOW>::kBlock; }
    else if (sv == 16) { *perCU = TravCfg<16, SHADOW>::kBlocksPerCU; *block = TravCfg<16, SHADOW>::kBlock; }
    else { *perCU = TravCfg<24, SHADOW>::kBlocksPerCU; *block = TravCfg<24, SHADOW>::kBlock; }
}
// profiled calls only: the rays the frame asked for beyond the queue's capacity (they were traced inline by wf_accumulate_kernel), and one count per call that had any
__global__ __launch_bounds__(64) void wf_overflow_tally_kernel(const uint32_t* counters, uint32_t rayCap, unsigned long long* tally, unsigned long long* statStripes) {
    unsigned long long answered = statStripes ? atomicExch(statStripes + threadIdx.x * 16u + 5u, 0ull) : 0ull;       // shared by the pipelines: whichever tallies first takes them
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) answered += __shfl_down(answered, off, 64);
    if (threadIdx.x == 0) { const uint32_t asked = counters[kRayCountBase]; tally[0] += asked; if (asked > rayCap) { tally[1] += asked - rayCap; tally[2] += 1ull; } tally[3] += answered + counters[kAnsweredBase]; }
}
// behind wf_shade_kernel(bounce): the shadow-ray queue's length with the rays of bounces 0 .. bounce in it
__global__ __launch_bounds__(64) void wf_snapshot_kernel(uint32_t* counters, int bounce) {
    if (threadIdx.x == 0) counters[kSnapBase + bounce + 1] = counters[kQueueCountBase];
}
} // namespace incr

// One frame = `numPipes` independent PIPELINES: pipeline i owns the tile rows [tileRow[i], tileRow[i + 1]) of the (local) frame, its own
// path-state buffers pipes[i] and its own stream streams[i] (streams[0] = the caller's; the others wait for `fork`, recorded on the
// caller's stream here, and the caller's stream waits for their `joins` at the end).  Why: a traversal kernel ends on its longest
// rays -- a ray is a chain of ~30 (max > 150) dependent fetch + test steps, ~2 us each once its wave runs alone -- and with one
// pipeline the whole GPU waits for that tail 8 times per frame; with two, the other pipeline's kernels fill the machine meanwhile.
// The effect grows as the frame shrinks (a 1/8 framebuffer shard of an 8-GPU run: profiles/r02_shard_emulation.json).
//
// DEFERRED ACCUMULATION (round 4).  chsMain is a recursion: it traces a vertex's shadow rays, adds their light, then traces the continuation
// ray.  Nothing a later vertex computes depends on the colour gathered so far -- payload.color is only added to -- so the pipeline's MAIN chain is
//     primary -> shade(0) -> extend(1) -> shade(1) -> extend(2) -> shade(2) -> extend(3) -> shade(3)
// and the shadow rays of all four bounces go into ONE queue that the any-hit launches trace as ranges (one launch per bounce group, `shadowAfter`)
// on the pipeline's SIDE stream, beside the main chain (`overlap`; without it they are issued in place on the main stream: every launch alone).
// wf_accumulate_kernel(0 .. 3) then replay payload.color in chsMain's order: colour so far + the vertex's own term + its lights in loop order + its
// sky sample, bounce after bounce -- the same additions in the same order, so the image is the one of the in-order issue bit for bit.  They run at the
// end of the main stream (a large frame) or each right behind the any-hit launch of its bounce on the side stream (a small one), by measurement.  What it buys: the critical path of a frame loses its four any-hit launches and their drains (a frame of few rays is a chain of
// launches each as long as its longest ray: helmet 1080p, profiles/r04_small_frame_trace.txt), and on a large frame the any-hit work fills
// the drains of the closest-hit launches and vice versa.
// `events` (optional, kWfProfEvents per pipeline): see pt_params.h.
extern "C" hipError_t incr_launch_pathtrace_wf(const incr::PtParams* p, const incr::WfBuffers* pipes, int numPipes, const uint32_t* tileRow, hipStream_t* streams,
                                               hipEvent_t fork, hipEvent_t* joins, int numCUs, uint32_t stackEntries, hipEvent_t* events,
                                               hipStream_t* sideStreams, hipEvent_t* bounceFork, hipEvent_t* bounceJoin, int overlap, int accPlacement, unsigned long long* overflowTally) {
    using namespace incr;
    const uint32_t tilesX = (p->W + 7u) / 8u, tilesY = (p->localRows + 7u) / 8u;      // tileRow[] counts rows of all samples in flight: tilesY per sample
    if (numPipes < 1 || numPipes > 8) return hipErrorInvalidValue;
    if (tileRow[numPipes] == 0) {          // an empty shard: nothing to launch, but a profiled call still owns its event record
        if (events) for (int k = 0; k < kWfProfEvents * 8; ++k) (void)hipEventRecord(events[k], streams[0]);
        return hipSuccess;
    }
    if (stackEntries > 24u) return hipErrorInvalidValue;
    const int sv = stackEntries <= 8u ? 8 : (stackEntries <= 10u ? 10 : (stackEntries <= 12u ? 12 : (stackEntries <= 16u ? 16 : 24)));
    const uint32_t cus = (uint32_t)std::max(numCUs, 1);
    // persistent traversal grids: enough waves to fill every CU at the kernels' occupancy (LDS stack bytes per workgroup decide it)
    uint32_t extendPerCU, extendBlock, shadowPerCU, shadowBlock;
    trav_geometry<false>(sv, &extendPerCU, &extendBlock); trav_geometry<true>(sv, &shadowPerCU, &shadowBlock);
    const uint32_t blocksShadow = cus * shadowPerCU;        // the length of the ray queue is only known on the device
    uint32_t total[8], blocksAll[8], blocksExtend[8], extendChunk[8];
    for (int i = 0; i < numPipes; ++i) {
        total[i] = (tileRow[i + 1] - tileRow[i]) * tilesX * 64u;
        // rays a closest-hit wave takes from the cursor at a time: 128 for the long lists of a large frame (fewer same-address atomics on the cursor: 4K frame
        // 17.74 -> 17.59 ms), 64 for short ones (helmet 1080p: 1.08 ms against 1.14 with 128: a chunk is then too large a share of a wave's work) -- r04_ab_chunks
        extendChunk[i] = total[i] >= 4000000u ? (uint32_t)INCR_EXTEND_CHUNK_LARGE : kExtendChunk;
        // generate / shade / shadow_gen / accumulate: one workgroup per 256 list slots and the hardware dispatcher as the load balancer
        // (capped at 4 / 8 / 16 / 32 workgroups per CU with a grid-stride loop the 4K frame took 46.0 / 43.8 / 43.9 / 43.1 ms, uncapped 42.7)
        blocksAll[i] = std::max(1u, (total[i] + kWfBlock - 1) / kWfBlock);
        blocksExtend[i] = std::max(1u, std::min(cus * extendPerCU, (total[i] + extendBlock - 1) / extendBlock));
    }
    if (numPipes > 1) {
        (void)hipEventRecord(fork, streams[0]);
        for (int i = 1; i < numPipes; ++i) (void)hipStreamWaitEvent(streams[i], fork, 0);
    }
    const bool firstHitOnly = p->viewmode >= 2 && p->viewmode <= 5;      // VIEWMODE 2-5 return at the first hit (pathtracing.hlsl:184-196); every other mode is a full path
    const int fuseGen = firstHitOnly ? 0 : 1;             // the shadow-ray set-up of a vertex runs at the end of wf_shade_kernel's workgroup; the material views make no vertices
    const bool side = overlap && sideStreams && bounceFork && bounceJoin && !firstHitOnly;
    // any-hit launches: one per bounce.  (Fewer, longer launches -- the rays of bounces 2 + 3, of 0 + 1 and 2 + 3, or of all four in one -- were measured:
    // alone, one launch over all 50 M rays takes 6.0 ms against 7.5 for the four, but a launch that waits for several bounces' rays has less of the
    // main chain left to run beside: 4K frame 17.83 ms per bounce, 18.13 / 18.41 / 18.63 / 18.78 ms for the groupings above, profiles/r04_ab_deferred.txt.)
    const uint32_t shadowAfterMask = 0xfu;
    // accumulate(b) right behind the any-hit launch that traced bounce b's rays (on its stream), or all four at the end of the main stream
    // Measured on one box (profiles/r04_ab_deferred.txt): 4K, one pipeline: 17.90 ms at the end, 18.00 behind the launches (the summation competes
    // with the main chain for the same wave slots); helmet 1080p, a chain of short launches: 1.13 ms at the end, 1.08 behind.  So: by size.
    uint64_t allPaths = 0; for (int i = 0; i < numPipes; ++i) allPaths += (uint64_t)(tileRow[i + 1] - tileRow[i]) * tilesX * 64u;
    // (A third placement -- the summation on a stream of its own, accumulate(b) waiting for the any-hit launch of bounce b by an event and running beside the main chain and
    // the later any-hit launches -- was built and measured: the four summations do move off the tail of the frame (rocprofv3 timeline: 0.19 ms left after the last any-hit launch
    // instead of 0.95), and the frame takes what it took, 16.675 against 16.70 ms over six interleaved runs each: the waves the summation occupies are waves the traversal
    // kernels do not run.  profiles/r04_ab_deferred.txt.)  accPlacement: -1 by size, 0 / 1 forced (INCR_WF_ACC_PLACEMENT).
    const int placement = !side ? 1 : (accPlacement >= 0 ? std::min(accPlacement, 1) : (allPaths < 3000000ull ? 1 : 0));
    const bool accBehindShadow = placement == 1;
    bool rec[8][kWfProfEvents] = {};
    auto ev = [&](int i, int bounce, int kind, int end, hipStream_t st) {          // kind: 0 extend, 1 shade, 2 shadow, 3 accumulate
        if (!events) return;
        const int k = 1 + ((bounce * 4 + kind) * 2 + end);
        (void)hipEventRecord(events[i * kWfProfEvents + k], st); rec[i][k] = true;
    };
    auto extend = [&](int i, int bounce, hipStream_t st) {
        const WfBuffers& b = pipes[i];
        const uint32_t ck = extendChunk[i];
        if (sv == 8) launch_extend<8>(blocksExtend[i], st, *p, b, bounce, ck); else if (sv == 10) launch_extend<10>(blocksExtend[i], st, *p, b, bounce, ck); else if (sv == 12) launch_extend<12>(blocksExtend[i], st, *p, b, bounce, ck); else if (sv == 16) launch_extend<16>(blocksExtend[i], st, *p, b, bounce, ck); else launch_extend<24>(blocksExtend[i], st, *p, b, bounce, ck);
    };
    auto shadow = [&](int i, int launch, int lo, int hi, hipStream_t st) {
        const WfBuffers& b = pipes[i];
        if (sv == 8) launch_shadow<8>(blocksShadow, st, *p, b, launch, lo, hi); else if (sv == 10) launch_shadow<10>(blocksShadow, st, *p, b, launch, lo, hi); else if (sv == 12) launch_shadow<12>(blocksShadow, st, *p, b, launch, lo, hi); else if (sv == 16) launch_shadow<16>(blocksShadow, st, *p, b, launch, lo, hi); else launch_shadow<24>(blocksShadow, st, *p, b, launch, lo, hi);
    };
    WfSkyConsts skyConsts;
    skyConsts.perez = p->perez; skyConsts.sunMiss = p->sunMiss;
    // stage-major issue order: the pipelines' kernels of one stage sit next to each other in the queues
    for (int i = 0; i < numPipes; ++i) {
        if (!total[i]) continue;
        hipLaunchKernelGGL(wf_init_counters_kernel, dim3(1), dim3(64), 0, streams[i], pipes[i].counters, blocksExtend[i] * (extendBlock / 64) * extendChunk[i], blocksShadow * (shadowBlock / 64) * kShadowChunk, skyConsts);
        if (events) { (void)hipEventRecord(events[i * kWfProfEvents], streams[i]); rec[i][0] = true; }
    }
    int groupLo[8] = {}, launches[8] = {};
    bool forked[8] = {};
    for (int bounce = 0; bounce < 4; ++bounce) {
        for (int i = 0; i < numPipes; ++i) {
            if (!total[i]) continue;
            hipStream_t st = streams[i]; const WfBuffers& b = pipes[i];
            ev(i, bounce, 0, 0, st);
            if (bounce == 0) hipLaunchKernelGGL(wf_primary_kernel, dim3(blocksAll[i]), dim3(kWfBlock), 0, st, *p, b, total[i], tilesX, tilesY, tileRow[i] * tilesX);
            else extend(i, bounce, st);
            ev(i, bounce, 0, 1, st);
            ev(i, bounce, 1, 0, st);
            if (p->envmap) hipLaunchKernelGGL(wf_shade_kernel<true>, dim3(blocksAll[i]), dim3(kWfBlock), 0, st, *p, b, bounce, fuseGen);
            else hipLaunchKernelGGL(wf_shade_kernel<false>, dim3(blocksAll[i]), dim3(kWfBlock), 0, st, *p, b, bounce, fuseGen);
            if (!firstHitOnly) hipLaunchKernelGGL(wf_snapshot_kernel, dim3(1), dim3(64), 0, st, b.counters, bounce);
            ev(i, bounce, 1, 1, st);
        }
        if (firstHitOnly) break;
        if (!((shadowAfterMask >> bounce) & 1u) && bounce < 3) continue;
        for (int i = 0; i < numPipes; ++i) {
            if (!total[i]) continue;
            hipStream_t st = streams[i];
            if (side) {          // the any-hit launch of this bounce group on the side stream, behind the shading that filled its range of the queue
                st = sideStreams[i];
                (void)hipEventRecord(bounceFork[i * 4 + bounce], streams[i]);
                (void)hipStreamWaitEvent(st, bounceFork[i * 4 + bounce], 0);
                forked[i] = true;
            }
            ev(i, bounce, 2, 0, st);
            shadow(i, launches[i]++, groupLo[i], bounce, st);
            ev(i, bounce, 2, 1, st);
            // the bounces whose rays are now traced can be summed: accumulate(b) needs the visibility of bounce b's rays and the colour accumulate(b - 1)
            // left -- both are behind it on this stream
            for (int b = groupLo[i]; b <= bounce && accBehindShadow; ++b) {
                ev(i, b, 3, 0, st);
                hipLaunchKernelGGL(wf_accumulate_kernel, dim3(blocksAll[i]), dim3(kWfBlock), 0, st, *p, pipes[i], b);
                ev(i, b, 3, 1, st);
            }
            groupLo[i] = bounce + 1;
        }
    }
    {
        for (int i = 0; i < numPipes; ++i) if (total[i] && forked[i]) { (void)hipEventRecord(bounceJoin[i * 4], sideStreams[i]); (void)hipStreamWaitEvent(streams[i], bounceJoin[i * 4], 0); }
        if (!firstHitOnly && !accBehindShadow) for (int bounce = 0; bounce < 4; ++bounce) for (int i = 0; i < numPipes; ++i) {
            if (!total[i]) continue;
            ev(i, bounce, 3, 0, streams[i]);
            hipLaunchKernelGGL(wf_accumulate_kernel, dim3(blocksAll[i]), dim3(kWfBlock), 0, streams[i], *p, pipes[i], bounce);
            ev(i, bounce, 3, 1, streams[i]);
        }
        if (firstHitOnly) for (int i = 0; i < numPipes; ++i) {          // material views: the camera rays that missed still get their sky
            if (!total[i]) continue;
            ev(i, 0, 3, 0, streams[i]);
            hipLaunchKernelGGL(wf_accumulate_kernel, dim3(blocksAll[i]), dim3(kWfBlock), 0, streams[i], *p, pipes[i], 0);
            ev(i, 0, 3, 1, streams[i]);
        }
    }
    if (events) for (int i = 0; i < numPipes; ++i) for (int k = 0; k < kWfProfEvents; ++k) if (!rec[i][k]) (void)hipEventRecord(events[i * kWfProfEvents + k], streams[i]);      // kernels that did not run: zero-length intervals
    if (overflowTally) for (int i = 0; i < numPipes; ++i) if (total[i]) hipLaunchKernelGGL(wf_overflow_tally_kernel, dim3(1), dim3(64), 0, streams[i], pipes[i].counters, pipes[i].rayCap, overflowTally + 4 * i, p->statStripes);
    for (int i = 1; i < numPipes; ++i) { (void)hipEventRecord(joins[i], streams[i]); (void)hipStreamWaitEvent(streams[0], joins[i], 0); }
    if (p->numSamples > 1u && p->localPixels) hipLaunchKernelGGL(wf_integrate_samples_kernel, dim3((p->localPixels + kWfBlock - 1) / kWfBlock), dim3(kWfBlock), 0, streams[0], *p);
    if (p->stats && p->statStripes) (void)incr_launch_fold_ray_stats(p->statStripes, p->stats, streams[0]);
    return hipGetLastError();
}
