// pathtrace_wf.hip -- the production path tracer: a WAVEFRONT (multi-kernel) formulation for gfx950.
//
// Why not one megakernel (pathtrace.hip / pathtrace_wave.hip)?  Measured on MI355X (profiles/r01_v2_*):
// the fused kernels need 147-169 VGPRs (2-3 waves per SIMD), waves spend 64 % of their cycles waiting on
// memory, VALU lanes are 31 % utilised, and HBM/L2-miss traffic is only ~150 GB/s -- the path is LATENCY
// bound, not bandwidth bound.  MI355X has 288 GB of HBM3E at ~8 TB/s, so the cure is to keep the per-path
// state in HBM (176 B per pixel) and cut the work into lean kernels that each do one thing at high occupancy:
//
//   per bounce b = 0..3:
//     extend     closest hit for every live path (b = 0 also generates the camera ray: rgsMain)    [traversal only]
//     shade      chsMain part 1: surface, material, emissive, sky sample, ENUMERATE shadow rays as a request
//                bit-mask per vertex, BRDF-sampled continuation ray                                   [no traversal]
//     shadow     every wave takes 64 vertices, expands their request masks into an LDS job queue and traces
//                the queue 64 rays at a time (RayQuery stand-in), clearing the bits of occluded requests [traversal only]
//     accumulate chsMain part 2: evaluates the unoccluded lights IN THE REFERENCE'S LOOP ORDER, then the sky
//                sample, adds them to payload.color; finished paths are stored + integrated              [no traversal]
//
// The float operations per pixel and their order are exactly those of pathtrace.hip (= the reference HLSL),
// so the three kernels variants are bit-identical; only the schedule differs.  Lists are compacted with one
// global atomic per 256-thread block (LDS-aggregated); entry order varies run to run, results do not.
//
// Light loop: chsMain loops over ALL scene lights per hit (pathtracing.hlsl:230-236).  A world-space light
// grid (scene_build.h LightGrid) yields, per hit, the ascending sub-list of lights that can be non-zero there;
// the skipped terms are exactly `attenuation * 0`, so the sum is unchanged (non-finite attenuation falls back to
// the full list).
#include "pt_shade.h"
#include <algorithm>

namespace incr {

static constexpr int kWfBlock = 256;
static constexpr int kWfStackLds = 16;                  // LDS stack entries per lane in the lean traversal kernels
using WfStack = LaneStackT<kWfStackLds, kWfBlock>;
static constexpr uint32_t kSkySlot = 127;               // bit 127 of the request mask = the sky sample
static constexpr uint32_t kFlagTerminate = 1u, kFlagSkip = 2u, kFlagAllLights = 4u;

// vertex record: 6 float4 per path
//   [0] hitPos.xyz, metallic   [1] N.xyz, roughness   [2] basecolor.xyz, flags   [3] attenuation at the vertex .xyz, cell
//   [4] skyDir.xyz, -          [5] request mask (4 x u32): shade sets the bits of the shadow rays it needs,
//                                  shadow clears the occluded ones, accumulate shades the lights whose bit survived
DI float4* vertex_of(const WfBuffers& B, uint32_t pid) { return B.vertex + (size_t)pid * 6; }

DI void add_counters(const PtParams& P, const Counters& cnt) {
    if (!P.stats) return;
    uint32_t v[5] = {cnt.camera, cnt.bounce, cnt.skyShadow, cnt.lightShadow, cnt.hits};
#pragma unroll
    for (int k = 0; k < 5; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_down(v[k], off, 64);
    }
    if ((threadIdx.x & 63u) == 0) {
        unsigned long long* s = (unsigned long long*)P.stats;
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

// ------------------------------------------------------------------------------------------------ extend
template <bool FIRST>
__global__ __launch_bounds__(kWfBlock) void wf_extend_kernel(const PtParams P, const WfBuffers B, int bounce, uint32_t totalFirst, uint32_t tilesX) {
    __shared__ uint32_t s_stack[kWfStackLds * kWfBlock];
    __shared__ uint32_t s_waveCount[kWfBlock / 64];
    __shared__ uint32_t s_base;
    WfStack st; st.lds = s_stack + threadIdx.x; st.sp = 0;
    Counters cnt = {0, 0, 0, 0, 0};
    const uint32_t* __restrict__ liveIn = (bounce & 1) ? B.liveB : B.liveA;
    const uint32_t count = FIRST ? totalFirst : B.counters[(bounce - 1) * 8 + 2];
    uint32_t* hitCounter = B.counters + bounce * 8 + 1;
    for (uint32_t base = blockIdx.x * kWfBlock; base < count; base += gridDim.x * kWfBlock) {
        const uint32_t i = base + threadIdx.x;
        bool active = i < count;
        uint32_t pid = 0, px = 0, ly = 0;
        Ray ray = {{0.f, 0.f, 0.f}, {0.f, 0.f, 1.f}, 0.001f, 10000.0f};
        if (active) {
            if (FIRST) {
                // 8x8 pixel tile per wave: primary rays of a wave walk the same BVH lines
                const uint32_t tile = i >> 6, w = i & 63u;
                px = (tile % tilesX) * 8u + (w & 7u); ly = (tile / tilesX) * 8u + (w >> 3);
                active = px < P.W && ly < P.localRows;
                if (active) {
                    const int64_t gy = shard_global_row(P, ly);
                    if (gy < 0) { store_pixel(P, px, ly, {0.f, 0.f, 0.f, 0.f}); active = false; }     // shard padding row
                    else {
                        pid = ly * P.W + px;
                        const uint32_t pixelIdx = (uint32_t)gy * P.W + px;
                        uint32_t raygenSampleSetIdx = 0;      // rgsMain -- pathtracing.hlsl:89-119
                        const f2 subPixelJitter = SamplePoint(pixelIdx, P.sampleIdx, raygenSampleSetIdx, P.sqrtSampleCount, P.numPixels);
                        const f2 apertureSample = SamplePoint(pixelIdx, P.sampleIdx, raygenSampleSetIdx, P.sqrtSampleCount, P.numPixels);
                        ray = GenerateCameraRay(P, {(float)px + subPixelJitter.x, (float)(uint32_t)gy + subPixelJitter.y}, apertureSample);
                    }
                }
            } else {
                pid = liveIn[i];
                px = pid % P.W; ly = pid / P.W;
                const float4 o = B.rayO[pid], d = B.rayD[pid];
                ray.o = {o.x, o.y, o.z}; ray.d = {d.x, d.y, d.z};
            }
        }
        bool gotHit = false;
        if (active) {
            if (FIRST) cnt.camera++; else cnt.bounce++;
            Hit hit;
            gotHit = trace_closest(P, st, ray.o, ray.d, ray.tmin, ray.tmax, &hit);
            if (gotHit) {
                B.hit[pid] = make_uint4(hit.tri, __float_as_uint(hit.t), __float_as_uint(hit.u), __float_as_uint(hit.v));
                if (FIRST) {
                    B.rayO[pid] = make_float4(ray.o.x, ray.o.y, ray.o.z, 0.f);
                    B.rayD[pid] = make_float4(ray.d.x, ray.d.y, ray.d.z, 0.f);
                    B.color[pid] = make_float4(0.f, 0.f, 0.f, 0.f);
                    B.atten[pid] = make_float4(1.f, 1.f, 1.f, 0.f);
                }
            } else {   // msDynamicSky -- pathtracing.hlsl:311-319
                f4 color = {0.f, 0.f, 0.f, 0.f}; f3 attenuation = {1.f, 1.f, 1.f};
                if (!FIRST) { const float4 c = B.color[pid], a = B.atten[pid]; color = {c.x, c.y, c.z, c.w}; attenuation = {a.x, a.y, a.z}; }
                const f3 sky = CalculateSkyRadianceRGB(P.sunMiss, ray.d, P.perez);
                const f3 add = attenuation * 1000 * sky;
                color.x += add.x; color.y += add.y; color.z += add.z;
                store_pixel(P, px, ly, color);
            }
        }
        const uint32_t slot = block_append(gotHit, hitCounter, s_waveCount, &s_base);
        if (gotHit) B.hitList[slot] = pid;
    }
    add_counters(P, cnt);
}

// ------------------------------------------------------------------------------------------------ shade
__global__ __launch_bounds__(kWfBlock) void wf_shade_kernel(const PtParams P, const WfBuffers B, int bounce) {
    __shared__ uint32_t s_waveCount[kWfBlock / 64];
    __shared__ uint32_t s_base;
    Counters cnt = {0, 0, 0, 0, 0};
    const uint32_t count = B.counters[bounce * 8 + 1];
    uint32_t* liveOut = (bounce & 1) ? B.liveA : B.liveB;          // extend(bounce + 1) reads the other list
    uint32_t* liveCounter = B.counters + bounce * 8 + 2;
    for (uint32_t base = blockIdx.x * kWfBlock; base < count; base += gridDim.x * kWfBlock) {
        const uint32_t i = base + threadIdx.x;
        bool cont = false;
        uint32_t pid = 0;
        if (i < count) {
            pid = B.hitList[i];
            const uint32_t px = pid % P.W, ly = pid / P.W;
            const uint32_t pixelIdx = (uint32_t)shard_global_row(P, ly) * P.W + px;
            const uint4 h = B.hit[pid];
            const float4 ro = B.rayO[pid], rd = B.rayD[pid], c4 = B.color[pid], a4 = B.atten[pid];
            const f3 rayO = {ro.x, ro.y, ro.z}, rayD = {rd.x, rd.y, rd.z};
            f4 color = {c4.x, c4.y, c4.z, c4.w};
            f3 attenuation = {a4.x, a4.y, a4.z};
            const float hitT = __uint_as_float(h.y), bx = __uint_as_float(h.z), by = __uint_as_float(h.w);
            // chsMain -- pathtracing.hlsl:121-268
            cnt.hits++;
            const int pathLength = bounce + 1;
            uint32_t sampleSetIndex = (uint32_t)bounce;               // one SamplePoint per earlier hit (Q3)
            const float4* __restrict__ sp = (const float4*)(P.shade + h.x);
            const float4 s0 = sp[0], s1 = sp[1], s2 = sp[2], s3 = sp[3], s4 = sp[4], s5 = sp[5], s6 = sp[6];
            const uint4 s7 = *(const uint4*)(sp + 7);
            const f3 vn0 = {s0.x, s0.y, s0.z}, vn1 = {s0.w, s1.x, s1.y}, vn2 = {s1.z, s1.w, s2.x};
            const f4 vt0 = {s2.y, s2.z, s2.w, s3.x}, vt1 = {s3.y, s3.z, s3.w, s4.x}, vt2 = {s4.y, s4.z, s4.w, s5.x};
            const f2 uv0 = {s5.y, s5.z}, uv1 = {s5.w, s6.x}, uv2 = {s6.y, s6.z};
            const uint32_t materialIndex = __float_as_uint(s6.w), meshIndex = s7.x;
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
            MatProps matInfo = EvaluateMaterialProperties(P.materials[materialIndex], P.textures, P.srgbLut, uv);
            float4* vtx = vertex_of(B, pid);
            bool done = false;
            if (P.viewmode == 1) matInfo.basecolor = mk3(0.5f);
            else if (P.viewmode == 2) { color = {matInfo.roughness, matInfo.roughness, matInfo.roughness, matInfo.roughness}; done = true; }
            else if (P.viewmode == 3) { color = {matInfo.metallic, matInfo.metallic, matInfo.metallic, matInfo.metallic}; done = true; }
            else if (P.viewmode == 4) { color = {matInfo.basecolor.x, matInfo.basecolor.y, matInfo.basecolor.z, 1.f}; done = true; }
            else if (P.viewmode == 5) { color = {matInfo.emissive.x, matInfo.emissive.y, matInfo.emissive.z, 1.f}; done = true; }
            if (done) {
                store_pixel(P, px, ly, color);
                vtx[2] = make_float4(0.f, 0.f, 0.f, __uint_as_float(kFlagSkip));
                *(uint4*)(vtx + 5) = make_uint4(0u, 0u, 0u, 0u);
            } else {
                const f3 add = attenuation * matInfo.emissive * 20000;
                color.x += add.x; color.y += add.y; color.z += add.z;
                attenuation = attenuation * saturate(1.f - matInfo.emissive);
                if (matInfo.bHasNormalmap) {
                    const f3 nm = matInfo.normalmap;
                    N = normalize(mk3((nm.x * T.x + nm.y * Bt.x) + nm.z * N.x, (nm.x * T.y + nm.y * Bt.y) + nm.z * N.y, (nm.x * T.z + nm.y * Bt.z) + nm.z * N.z));
                }
                const f3 V = -rayD;
                const f3 attenV = attenuation;
                // ---- enumerate the shadow rays of this vertex as a request mask ----
                uint32_t req[4] = {0u, 0u, 0u, 0u};
                uint32_t flags = 0u;
                const bool finiteAtt = fabsf(attenV.x) < 3.0e38f && fabsf(attenV.y) < 3.0e38f && fabsf(attenV.z) < 3.0e38f;
                const uint32_t numCells = P.gridDim[0] * P.gridDim[1] * P.gridDim[2];
                uint32_t cell = light_cell(P, hitPosition);
                if (!finiteAtt) { cell = numCells; flags |= kFlagAllLights; }
                if (P.directLighting && P.lightCount > 0) {
                    const uint32_t lo = P.gridCellOffsets[cell], hi = P.gridCellOffsets[cell + 1];
                    const uint32_t n = min(hi - lo, kSkySlot);
                    for (uint32_t k = 0; k < n; ++k) {
                        incr_light light; const float* m; bool haveT;
                        load_light(P, (int)P.gridCellLights[lo + k], light, m, haveT);
                        const LightSample s = light_setup(light, m, haveT, hitPosition, N);
                        if (s.needsShadow) { req[k >> 5] |= 1u << (k & 31u); cnt.lightShadow++; }
                    }
                    for (uint32_t k = n; k < hi - lo; ++k) {      // > 127 candidates in one cell: traced inside accumulate, counted here
                        incr_light light; const float* m; bool haveT;
                        load_light(P, (int)P.gridCellLights[lo + k], light, m, haveT);
                        if (light_setup(light, m, haveT, hitPosition, N).needsShadow) cnt.lightShadow++;
                    }
                }
                // sky sample :239-244
                const f2 us = SamplePoint(pixelIdx, P.sampleIdx, sampleSetIndex, P.sqrtSampleCount, P.numPixels);
                const f3 skyDir = CosineSampleHemisphere(us);
                const f3 skyRadianceIn = P.skyBrightness * (0.04f * CalculateSkyRadianceRGB(P.sunEnv, skyDir, P.perez));
                const float skyNoL = saturate(dot(N, skyDir));
                if (skyNoL > 0.f && (skyRadianceIn.x > 0.f || skyRadianceIn.y > 0.f || skyRadianceIn.z > 0.f)) { req[3] |= 0x80000000u; cnt.skyShadow++; }
                // ---- BRDF-sampled continuation (raytracing/common.hlsli:55-123) ----
                if (pathLength < 4) {
                    uint32_t setCopy = sampleSetIndex;
                    f2 u = SamplePoint(pixelIdx, P.sampleIdx, setCopy, P.sqrtSampleCount, P.numPixels);
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
                    const float pdf = t * CosineHemispherePdf(NoL) + (1.f - t) * GGXPdf(NoH, matInfo.roughness) / (4.f * VoH);
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
                        B.rayO[pid] = make_float4(hitPosition.x, hitPosition.y, hitPosition.z, 0.f);
                        B.rayD[pid] = make_float4(L.x, L.y, L.z, 0.f);
                        B.atten[pid] = make_float4(attenuation.x, attenuation.y, attenuation.z, 0.f);
                        cont = true;
                    }
                }
                if (!cont) flags |= kFlagTerminate;
                B.color[pid] = make_float4(color.x, color.y, color.z, color.w);
                vtx[0] = make_float4(hitPosition.x, hitPosition.y, hitPosition.z, matInfo.metallic);
                vtx[1] = make_float4(N.x, N.y, N.z, matInfo.roughness);
                vtx[2] = make_float4(matInfo.basecolor.x, matInfo.basecolor.y, matInfo.basecolor.z, __uint_as_float(flags));
                vtx[3] = make_float4(attenV.x, attenV.y, attenV.z, __uint_as_float(cell));
                // accumulate needs V = -rayD, but rayD[pid] now holds the continuation ray: keep the incoming direction in the
                // record (x) and in the no longer needed t/u slots of the hit buffer (y, z)
                vtx[4] = make_float4(skyDir.x, skyDir.y, skyDir.z, rayD.x);
                *(uint4*)(vtx + 5) = make_uint4(req[0], req[1], req[2], req[3]);
                B.hit[pid] = make_uint4(h.x, __float_as_uint(rayD.y), __float_as_uint(rayD.z), 0u);
            }
        }
        const uint32_t slot = block_append(cont, liveCounter, s_waveCount, &s_base);
        if (cont) liveOut[slot] = pid;
    }
    add_counters(P, cnt);
}

// ------------------------------------------------------------------------------------------------ shadow
struct ShadowShared {
    uint16_t jobs[64 * 16];         // (owner lane << 8) | bit   (one 16-bit half of a mask word per round)
    float hitPos[3][64];
    float skyDir[3][64];
    uint32_t cell[64];
    uint32_t occluded[64];
    uint32_t jobCount;
    uint32_t pad[3];
};

DI void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ __launch_bounds__(kWfBlock) void wf_shadow_kernel(const PtParams P, const WfBuffers B, int bounce) {
    __shared__ uint32_t s_stack[kWfStackLds * kWfBlock];
    __shared__ ShadowShared s_wave[kWfBlock / 64];
    const uint32_t lane = threadIdx.x & 63u;
    ShadowShared& ws = s_wave[threadIdx.x >> 6];
    WfStack st; st.lds = s_stack + threadIdx.x; st.sp = 0;
    const uint32_t count = B.counters[bounce * 8 + 1];
    const uint32_t waveId = (blockIdx.x * kWfBlock + threadIdx.x) >> 6, numWaves = (gridDim.x * kWfBlock) >> 6;
    for (uint32_t base = waveId * 64u; base < count; base += numWaves * 64u) {      // wave-uniform loop: no block barrier inside
        const uint32_t i = base + lane;
        const bool valid = i < count;
        uint32_t pid = 0; uint4 req = make_uint4(0u, 0u, 0u, 0u);
        float4* vtx = nullptr;
        if (valid) {
            pid = B.hitList[i];
            vtx = vertex_of(B, pid);
            req = *(const uint4*)(vtx + 5);
            const float4 v0 = vtx[0], v3 = vtx[3], v4 = vtx[4];
            ws.hitPos[0][lane] = v0.x; ws.hitPos[1][lane] = v0.y; ws.hitPos[2][lane] = v0.z;
            ws.skyDir[0][lane] = v4.x; ws.skyDir[1][lane] = v4.y; ws.skyDir[2][lane] = v4.z;
            ws.cell[lane] = __float_as_uint(v3.w);
        }
        uint32_t words[4] = {req.x, req.y, req.z, req.w};
        bool changed = false;
#pragma unroll
        for (int hw = 0; hw < 8; ++hw) {
            const int w = hw >> 1; const uint32_t shift = (uint32_t)(hw & 1) * 16u;
            uint32_t bits = (words[w] >> shift) & 0xffffu;
            if (__ballot(bits != 0u) == 0ull) continue;                // wave-uniform skip of empty half-words
            ws.occluded[lane] = 0u;
            if (lane == 0) ws.jobCount = 0u;
            wave_lds_sync();
            while (bits) {
                const uint32_t b = (uint32_t)__builtin_ctz(bits); bits &= bits - 1u;
                const uint32_t slot = atomicAdd(&ws.jobCount, 1u);
                ws.jobs[slot] = (uint16_t)((lane << 8) | b);
            }
            wave_lds_sync();
            const uint32_t jobCount = ws.jobCount;
            for (uint32_t jb = 0; jb < jobCount; jb += 64u) {
                const uint32_t j = jb + lane;
                if (j < jobCount) {
                    const uint32_t job = ws.jobs[j];
                    const uint32_t owner = job >> 8, bit = job & 0xffu, slotIdx = (uint32_t)w * 32u + shift + bit;
                    const f3 origin = {ws.hitPos[0][owner], ws.hitPos[1][owner], ws.hitPos[2][owner]};
                    f3 L;
                    if (slotIdx == kSkySlot) L = {ws.skyDir[0][owner], ws.skyDir[1][owner], ws.skyDir[2][owner]};
                    else {
                        const uint32_t li = P.gridCellLights[P.gridCellOffsets[ws.cell[owner]] + slotIdx];
                        incr_light light; const float* m; bool haveT;
                        load_light(P, (int)li, light, m, haveT);
                        L = light_setup(light, m, haveT, origin, {0.f, 0.f, 0.f}).L;      // same L as shade computed (N does not enter L)
                    }
                    if (trace_any(P, st, origin, L, 0.1f, 1000.f)) atomicOr(&ws.occluded[owner], 1u << bit);
                }
            }
            wave_lds_sync();
            const uint32_t occ = ws.occluded[lane];
            if (occ) { words[w] &= ~(occ << shift); changed = true; }
            wave_lds_sync();
        }
        if (valid && changed) *(uint4*)(vtx + 5) = make_uint4(words[0], words[1], words[2], words[3]);
    }
}

// ------------------------------------------------------------------------------------------------ accumulate
__global__ __launch_bounds__(kWfBlock) void wf_accumulate_kernel(const PtParams P, const WfBuffers B, int bounce) {
    const uint32_t count = B.counters[bounce * 8 + 1];
    for (uint32_t i = blockIdx.x * kWfBlock + threadIdx.x; i < count; i += gridDim.x * kWfBlock) {
        const uint32_t pid = B.hitList[i];
        const float4* vtx = vertex_of(B, pid);
        const float4 v2 = vtx[2];
        const uint32_t flags = __float_as_uint(v2.w);
        if (flags & kFlagSkip) continue;
        const float4 v0 = vtx[0], v1 = vtx[1], v3 = vtx[3], v4 = vtx[4];
        const uint4 vis = *(const uint4*)(vtx + 5);
        const uint4 h = B.hit[pid];
        const float4 c4 = B.color[pid];
        f4 color = {c4.x, c4.y, c4.z, c4.w};
        const f3 hitPosition = {v0.x, v0.y, v0.z}, N = {v1.x, v1.y, v1.z}, basecolor = {v2.x, v2.y, v2.z}, attenV = {v3.x, v3.y, v3.z};
        const f3 skyDir = {v4.x, v4.y, v4.z};
        const f3 V = -mk3(v4.w, __uint_as_float(h.y), __uint_as_float(h.z));
        MatProps matInfo = {};
        matInfo.basecolor = basecolor; matInfo.metallic = v0.w; matInfo.roughness = v1.w;
        const uint32_t cell = __float_as_uint(v3.w);
        const uint32_t words[4] = {vis.x, vis.y, vis.z, vis.w};
        if (P.directLighting && P.lightCount > 0) {
            const uint32_t lo = P.gridCellOffsets[cell], hi = P.gridCellOffsets[cell + 1];
            const uint32_t n = hi - lo;
            for (uint32_t k = 0; k < n; ++k) {
                incr_light light; const float* m; bool haveT;
                load_light(P, (int)P.gridCellLights[lo + k], light, m, haveT);
                const LightSample s = light_setup(light, m, haveT, hitPosition, N);
                f3 radianceOut = {0.f, 0.f, 0.f};
                bool visible;
                if (k < kSkySlot) visible = (words[k >> 5] >> (k & 31u)) & 1u;
                else {   // more than 127 candidate lights in one cell: traced here (slow path, private-memory stack)
                    visible = false;
                    if (s.needsShadow) { LaneStackT<0, 1> ps; ps.lds = nullptr; ps.sp = 0; visible = !trace_any(P, ps, hitPosition, s.L, 0.1f, 1000.f); }
                }
                if (s.needsShadow && visible) radianceOut = light_shade(s, matInfo.basecolor, matInfo.metallic, matInfo.roughness, N, V);
                const f3 r = attenV * radianceOut;
                color.x += r.x; color.y += r.y; color.z += r.z;
            }
        }
        {   // sky sample (pathtracing.hlsl:239-244 + lighting/common.hlsli:145-202)
            const f3 skyRadianceIn = P.skyBrightness * (0.04f * CalculateSkyRadianceRGB(P.sunEnv, skyDir, P.perez));
            const float skyNoL = saturate(dot(N, skyDir));
            f3 radianceOut = {0.f, 0.f, 0.f};
            if (words[3] >> 31) radianceOut = sky_shade(skyRadianceIn, skyDir, skyNoL, matInfo, N, V);
            const f3 r = attenV * (radianceOut / CosineHemispherePdf(skyDir.z));
            color.x += r.x; color.y += r.y; color.z += r.z;
        }
        if (flags & kFlagTerminate) store_pixel(P, pid % P.W, pid / P.W, color);
        else B.color[pid] = make_float4(color.x, color.y, color.z, color.w);
    }
}

} // namespace incr

// ---- launcher ------------------------------------------------------------------------------------------
extern "C" hipError_t incr_launch_pathtrace_wf(const incr::PtParams* p, const incr::WfBuffers* b, int numCUs, hipStream_t stream) {
    using namespace incr;
    const uint32_t tilesX = (p->W + 7u) / 8u, tilesY = (p->localRows + 7u) / 8u;
    const uint32_t totalFirst = tilesX * tilesY * 64u;
    if (totalFirst == 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(b->counters, 0, 4 * 8 * sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    const uint32_t maxBlocks = (uint32_t)std::max(numCUs, 1) * 8u;
    const uint32_t blocksAll = std::max(1u, std::min(maxBlocks, (totalFirst + kWfBlock - 1) / kWfBlock));
    for (int bounce = 0; bounce < 4; ++bounce) {
        if (bounce == 0) hipLaunchKernelGGL((wf_extend_kernel<true>), dim3(blocksAll), dim3(kWfBlock), 0, stream, *p, *b, bounce, totalFirst, tilesX);
        else hipLaunchKernelGGL((wf_extend_kernel<false>), dim3(blocksAll), dim3(kWfBlock), 0, stream, *p, *b, bounce, totalFirst, tilesX);
        hipLaunchKernelGGL(wf_shade_kernel, dim3(blocksAll), dim3(kWfBlock), 0, stream, *p, *b, bounce);
        if (p->viewmode < 2) {
            hipLaunchKernelGGL(wf_shadow_kernel, dim3(blocksAll), dim3(kWfBlock), 0, stream, *p, *b, bounce);
            hipLaunchKernelGGL(wf_accumulate_kernel, dim3(blocksAll), dim3(kWfBlock), 0, stream, *p, *b, bounce);
        } else break;   // VIEWMODE >= 2 paths end at the first hit
    }
    return hipGetLastError();
}
