// pathtrace_wave.hip -- persistent-wavefront path-tracing kernel for gfx950 (the production kernel).
//
// Same arithmetic per pixel as pathtrace.hip (and therefore as the reference's rgsMain / chsMain /
// ahsMain / msDynamicSky, shaders/raytracing/pathtracing.hlsl:89-319, plus the fused integrate pass), but
// scheduled for 64-wide CDNA4 wavefronts instead of one-thread-per-pixel:
//
//  * PERSISTENT WAVES + TILE QUEUE.  Every wave loops: lanes whose path ended take the next pixel of the
//    wave's pending 8x8 tile; when that is used up one lane pops the next tile index from a device-wide
//    atomic counter.  All lanes therefore enter trace_closest together on every iteration, whatever
//    their bounce depth, and the frame tail is one tile per wave instead of the longest path per pixel block.
//    Every wave reaches the exit (queue empty and no lane alive): the grid always drains.
//  * WAVE-COOPERATIVE SHADOW RAYS.  chsMain traces one RayQuery per contributing light and one for the
//    sky sample (lighting/common.hlsli:102-115,161-174).  Executed lane-serially that is ~3 % SIMD
//    utilisation on a 65-light scene (most lights are out of range for most hits).  Here each lane only
//    ENUMERATES its shadow rays into an LDS job queue (wave ballot-free: ds_add_rtn), then the whole wave
//    traces the queue 64 jobs at a time and ORs the occlusion bit into the owner lane's LDS mask; finally each
//    lane evaluates its lights IN THE REFERENCE ORDER with the visibility looked up -- the float sums are
//    the same sequence of operations as the lane-serial form, so results stay bit-identical to pathtrace.hip.
//  * LDS: traversal stacks [entry][lane], per-wave job queue / hit positions / sky directions / masks.
//    No __syncthreads(): the four waves of a workgroup are independent; wave-scope fences order LDS traffic.
#include "pt_shade.h"
#include <algorithm>

namespace incr {

static constexpr int kWaves = kBlock / 64;
static constexpr int kChunkLights = 31;                 // lights per chunk; bit 31 of the mask is the sky sample
static constexpr int kMaxJobs = 64 * (kChunkLights + 1);

struct WaveShared {
    uint16_t jobs[kMaxJobs];        // (owner lane << 8) | slot   slot < 31: light chunkBase+slot, slot 31: sky sample
    float hitPos[3][64];
    float skyDir[3][64];
    uint32_t occluded[64];          // per owner lane: bit s set <=> job slot s is occluded
    uint32_t jobCount;
    uint32_t pad[3];
};

DI void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ __launch_bounds__(kBlock) void pathtrace_wave_kernel(const PtParams P, uint32_t* __restrict__ tileQueue, uint32_t numTiles, uint32_t tilesX) {
    __shared__ uint32_t s_stack[kStackLds * kBlock];
    __shared__ WaveShared s_wave[kWaves];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    WaveShared& ws = s_wave[tid >> 6];
    LaneStack st; st.bind(s_stack + tid, nullptr); st.sp = 0;
    Counters cnt = {0, 0, 0, 0, 0};
    const unsigned long long laneBit = 1ull << lane, laneMaskLt = laneBit - 1ull;

    // per-lane path state (pathtracing.hlsl:42-49 RayPayload + the current ray)
    bool alive = false;
    uint32_t px = 0, ly = 0, pixelIdx = 0, sampleSetIndex = 0;
    int pathLength = 0;
    f4 color = {0.f, 0.f, 0.f, 0.f};
    f3 attenuation = {1.f, 1.f, 1.f};
    Ray ray = {{0.f, 0.f, 0.f}, {0.f, 0.f, 1.f}, 0.f, 0.f};
    bool isCameraRay = true;

    // wave-uniform refill state
    uint32_t pendTile = 0, pendNext = 64;       // pixels [pendNext, 64) of tile pendTile are unassigned
    bool queueEmpty = false;
    const int numChunks = (P.directLighting && P.lightCount > 0) ? (P.lightCount + kChunkLights - 1) / kChunkLights : 0;

    for (;;) {
        // ---------------- refill idle lanes from the pending tile / the device-wide tile queue ----------------
        for (;;) {
            const unsigned long long idle = __ballot(!alive);
            if (idle == 0ull) break;
            if (pendNext >= 64u) {
                if (queueEmpty) break;
                uint32_t t = 0;
                if (lane == 0) t = atomicAdd(tileQueue, 1u);
                t = __builtin_amdgcn_readfirstlane(t);
                if (t >= numTiles) { queueEmpty = true; break; }
                pendTile = t; pendNext = 0;
            }
            const uint32_t rank = (uint32_t)__popcll(idle & laneMaskLt);
            const uint32_t avail = 64u - pendNext;
            if (!alive && rank < avail) {
                const uint32_t i = pendNext + rank;
                const uint32_t x = (pendTile % tilesX) * 8u + (i & 7u), row = (pendTile / tilesX) * 8u + (i >> 3);
                const int64_t gy = row < P.localRows ? shard_global_row(P, row) : -2;
                if (x < P.W && gy >= 0) {
                    // rgsMain -- pathtracing.hlsl:89-119
                    px = x; ly = row; pixelIdx = (uint32_t)gy * P.W + x;
                    uint32_t raygenSampleSetIdx = 0;
                    const f2 subPixelJitter = SamplePoint(pixelIdx, P.sampleIdx, raygenSampleSetIdx, P.sqrtSampleCount, P.numPixels);
                    const f2 apertureSample = SamplePoint(pixelIdx, P.sampleIdx, raygenSampleSetIdx, P.sqrtSampleCount, P.numPixels);
                    ray = GenerateCameraRay(P, {(float)x + subPixelJitter.x, (float)(uint32_t)gy + subPixelJitter.y}, apertureSample);
                    color = {0.f, 0.f, 0.f, 0.f}; attenuation = {1.f, 1.f, 1.f};
                    pathLength = 0; sampleSetIndex = 0; isCameraRay = true;
                    alive = true;
                } else if (x < P.W && gy == -1) {
                    store_pixel(P, x, row, {0.f, 0.f, 0.f, 0.f});          // padding row of a shard: stays zero
                }
            }
            const uint32_t taken = min((uint32_t)__popcll(idle), avail);
            pendNext += taken;
        }
        if (__ballot(alive) == 0ull) break;        // queue drained and every path finished

        // ---------------- closest hit for every live lane ----------------
        Hit hit; bool gotHit = false;
        if (alive) {
            if (isCameraRay) cnt.camera++; else cnt.bounce++;
            isCameraRay = false;
            gotHit = trace_closest(P, st, ray.o, ray.d, ray.tmin, ray.tmax, &hit);
            if (!gotHit) {   // msEnvmap / msDynamicSky -- pathtracing.hlsl:304-319
                const f3 add = sky_miss_add(P, attenuation, ray.d);
                color.x += add.x; color.y += add.y; color.z += add.z;
                store_pixel(P, px, ly, color);
                alive = false;
            }
        }
        // ---------------- chsMain, part 1: surface + material ----------------
        bool shading = alive && gotHit;
        MatProps matInfo = {};
        f3 N = {0.f, 0.f, 1.f}, T = {1.f, 0.f, 0.f}, B = {0.f, 1.f, 0.f}, Ntbn = {0.f, 0.f, 1.f}, V = {0.f, 0.f, 1.f}, hitPosition = {0.f, 0.f, 0.f};
        if (shading) {
            cnt.hits++;
            pathLength += 1;
            const float4* __restrict__ sp = (const float4*)(P.shade + hit.tri);
            const float4 s0 = sp[0], s1 = sp[1], s2 = sp[2], s3 = sp[3], s4 = sp[4], s5 = sp[5], s6 = sp[6];
            const uint4 s7 = *(const uint4*)(sp + 7);
            const f3 vn0 = {s0.x, s0.y, s0.z}, vn1 = {s0.w, s1.x, s1.y}, vn2 = {s1.z, s1.w, s2.x};
            const f4 vt0 = {s2.y, s2.z, s2.w, s3.x}, vt1 = {s3.y, s3.z, s3.w, s4.x}, vt2 = {s4.y, s4.z, s4.w, s5.x};
            const f2 uv0 = {s5.y, s5.z}, uv1 = {s5.w, s6.x}, uv2 = {s6.y, s6.z};
            const uint32_t materialIndex = __float_as_uint(s6.w), meshIndex = s7.x;
            const float* __restrict__ localToWorld = P.meshLocalToWorld + (size_t)meshIndex * 16;
            hitPosition = ray.o + hit.t * ray.d;
            const float bx = hit.u, by = hit.v;
            const f4 packedT = vt0 + bx * (vt1 - vt0) + by * (vt2 - vt0);
            N = normalize(vn0 + bx * (vn1 - vn0) + by * (vn2 - vn0));
            T = normalize(mk3(packedT.x, packedT.y, packedT.z));
            B = cross(N, T) * packedT.w;
            N = mul_dir(N, localToWorld);
            T = mul_dir(T, localToWorld);
            B = mul_dir(B, localToWorld);
            Ntbn = N;
            const f2 uv = uv0 + bx * (uv1 - uv0) + by * (uv2 - uv0);
            matInfo = EvaluateMaterialProperties(P.materials[materialIndex], P.textures, P.srgbLut, uv);
            bool done = false;
            if (P.viewmode == 1) matInfo.basecolor = mk3(0.5f);
            else if (P.viewmode == 2) { color = {matInfo.roughness, matInfo.roughness, matInfo.roughness, matInfo.roughness}; done = true; }
            else if (P.viewmode == 3) { color = {matInfo.metallic, matInfo.metallic, matInfo.metallic, matInfo.metallic}; done = true; }
            else if (P.viewmode == 4) { color = {matInfo.basecolor.x, matInfo.basecolor.y, matInfo.basecolor.z, 1.f}; done = true; }
            else if (P.viewmode == 5) { color = {matInfo.emissive.x, matInfo.emissive.y, matInfo.emissive.z, 1.f}; done = true; }
            if (done) { store_pixel(P, px, ly, color); alive = false; shading = false; }
            else {
                const f3 add = attenuation * matInfo.emissive * 20000;
                color.x += add.x; color.y += add.y; color.z += add.z;
                attenuation = attenuation * saturate(1.f - matInfo.emissive);
                if (matInfo.bHasNormalmap) {
                    const f3 nm = matInfo.normalmap;
                    N = normalize(mk3((nm.x * T.x + nm.y * B.x) + nm.z * N.x, (nm.x * T.y + nm.y * B.y) + nm.z * N.y, (nm.x * T.z + nm.y * B.z) + nm.z * N.z));
                }
                V = -ray.d;
                ws.hitPos[0][lane] = hitPosition.x; ws.hitPos[1][lane] = hitPosition.y; ws.hitPos[2][lane] = hitPosition.z;
            }
        }
        if (__ballot(shading) == 0ull) continue;

        // ---------------- chsMain, part 2: next-event estimation, shadow rays traced by the whole wave ----------------
        // sky sample (:239-244) is enumerated with the last chunk so that its contribution is added after every light
        f3 skyDir = {0.f, 0.f, 1.f}, skyRadianceIn = {0.f, 0.f, 0.f}; float skyNoL = 0.f; bool skyNeedsShadow = false;
        if (shading) {
            const f2 u = SamplePoint(pixelIdx, P.sampleIdx, sampleSetIndex, P.sqrtSampleCount, P.numPixels);
            skyDir = CosineSampleHemisphere(u);
            skyRadianceIn = P.skyBrightness * sky_env_radiance(P, skyDir);
            skyNoL = saturate(dot(N, skyDir));
            skyNeedsShadow = skyNoL > 0.f && (skyRadianceIn.x > 0.f || skyRadianceIn.y > 0.f || skyRadianceIn.z > 0.f);
            ws.skyDir[0][lane] = skyDir.x; ws.skyDir[1][lane] = skyDir.y; ws.skyDir[2][lane] = skyDir.z;
        }
        const int rounds = numChunks > 0 ? numChunks : 1;
        for (int chunk = 0; chunk < rounds; ++chunk) {
            const int lightBase = chunk * kChunkLights;
            const int lightEnd = numChunks > 0 ? min(lightBase + kChunkLights, P.lightCount) : 0;
            const bool lastRound = chunk == rounds - 1;
            ws.occluded[lane] = 0u;
            if (lane == 0) ws.jobCount = 0u;
            wave_lds_sync();
            // phase 1: enumerate
            if (shading) {
                for (int li = lightBase; li < lightEnd; ++li) {
                    incr_light light; const float* m; bool haveT;
                    load_light(P, li, light, m, haveT);
                    const LightSample s = light_setup(light, m, haveT, hitPosition, N);
                    if (s.needsShadow) {
                        cnt.lightShadow++;
                        const uint32_t slot = atomicAdd(&ws.jobCount, 1u);
                        ws.jobs[slot] = (uint16_t)((lane << 8) | (uint32_t)(li - lightBase));
                    }
                }
                if (lastRound && skyNeedsShadow) {
                    cnt.skyShadow++;
                    const uint32_t slot = atomicAdd(&ws.jobCount, 1u);
                    ws.jobs[slot] = (uint16_t)((lane << 8) | 31u);
                }
            }
            wave_lds_sync();
            // phase 2: the whole wave traces the queue, 64 shadow rays at a time
            const uint32_t jobCount = ws.jobCount;
            for (uint32_t base = 0; base < jobCount; base += 64u) {
                const uint32_t j = base + lane;
                if (j < jobCount) {
                    const uint32_t job = ws.jobs[j];
                    const uint32_t owner = job >> 8, slot = job & 0xffu;
                    const f3 origin = {ws.hitPos[0][owner], ws.hitPos[1][owner], ws.hitPos[2][owner]};
                    f3 L;
                    if (slot == 31u) L = {ws.skyDir[0][owner], ws.skyDir[1][owner], ws.skyDir[2][owner]};
                    else {
                        incr_light light; const float* m; bool haveT;
                        load_light(P, lightBase + (int)slot, light, m, haveT);
                        L = light_setup(light, m, haveT, origin, {0.f, 0.f, 0.f}).L;      // same L as the owner computes (N is not an input of L)
                    }
                    if (trace_any(P, st, origin, L, 0.1f, 1000.f)) atomicOr(&ws.occluded[owner], 1u << slot);
                }
            }
            wave_lds_sync();
            // phase 3: shade in the reference's order with the looked-up visibility
            if (shading) {
                const uint32_t occ = ws.occluded[lane];
                for (int li = lightBase; li < lightEnd; ++li) {
                    incr_light light; const float* m; bool haveT;
                    load_light(P, li, light, m, haveT);
                    const LightSample s = light_setup(light, m, haveT, hitPosition, N);
                    f3 radianceOut = {0.f, 0.f, 0.f};
                    if (s.needsShadow && !((occ >> (uint32_t)(li - lightBase)) & 1u))
                        radianceOut = light_shade(s, matInfo.basecolor, matInfo.metallic, matInfo.roughness, N, V);
                    const f3 r = attenuation * radianceOut;
                    color.x += r.x; color.y += r.y; color.z += r.z;
                }
                if (lastRound) {
                    f3 radianceOut = {0.f, 0.f, 0.f};
                    if (skyNeedsShadow && !(occ >> 31)) radianceOut = sky_shade(skyRadianceIn, skyDir, skyNoL, matInfo, N, V);
                    const f3 r = attenuation * (radianceOut / CosineHemispherePdf(skyDir.z));
                    color.x += r.x; color.y += r.y; color.z += r.z;
                }
            }
            wave_lds_sync();
        }

        // ---------------- chsMain, part 3: BRDF-sampled continuation (raytracing/common.hlsli:55-123) ----------------
        if (shading) {
            bool cont = false;
            if (pathLength < 4) {
                uint32_t setCopy = sampleSetIndex;
                f2 u = SamplePoint(pixelIdx, P.sampleIdx, setCopy, P.sqrtSampleCount, P.numPixels);
                f3 L, H;
                const float t = 0.5f;
                if (u.x < t) {
                    u.x = 2.f * u.x;
                    const f3 l = CosineSampleHemisphere(u);
                    L = normalize(mk3((l.x * T.x + l.y * B.x) + l.z * Ntbn.x, (l.x * T.y + l.y * B.y) + l.z * Ntbn.y, (l.x * T.z + l.y * B.z) + l.z * Ntbn.z));
                    H = normalize(L + V);
                } else {
                    u.x = 2.f * (u.x - 0.5f);
                    const f3 h = SampleGGX(u, matInfo.roughness);
                    H = normalize(mk3((h.x * T.x + h.y * B.x) + h.z * Ntbn.x, (h.x * T.y + h.y * B.y) + h.z * Ntbn.y, (h.x * T.z + h.y * B.z) + h.z * Ntbn.z));
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
                    ray.o = hitPosition; ray.d = L; ray.tmin = 0.001f; ray.tmax = 10000.0f;
                    cont = true;
                }
            }
            if (!cont) { store_pixel(P, px, ly, color); alive = false; }
        }
    }

    if (P.stats) {
        uint32_t v[5] = {cnt.camera, cnt.bounce, cnt.skyShadow, cnt.lightShadow, cnt.hits};
#pragma unroll
        for (int k = 0; k < 5; ++k) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_down(v[k], off, 64);
        }
        if (lane == 0) {
            unsigned long long* s = (unsigned long long*)P.stats;
#pragma unroll
            for (int k = 0; k < 5; ++k) if (v[k]) atomicAdd(s + k, (unsigned long long)v[k]);
        }
    }
}

} // namespace incr

extern "C" hipError_t incr_launch_pathtrace_wave(const incr::PtParams* p, uint32_t* tileQueue, int numCUs, hipStream_t stream) {
    const uint32_t tilesX = (p->W + 7u) / 8u, tilesY = (p->localRows + 7u) / 8u;
    const uint32_t numTiles = tilesX * tilesY;
    if (numTiles == 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(tileQueue, 0, sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    // persistent grid: enough workgroups to fill every CU at the kernel's occupancy, never more waves than tiles
    const uint32_t wantBlocks = (uint32_t)std::max(numCUs, 1) * 4u;
    const uint32_t blocks = std::max(1u, std::min(wantBlocks, (numTiles + incr::kWaves - 1) / incr::kWaves));
    hipLaunchKernelGGL(incr::pathtrace_wave_kernel, dim3(blocks), dim3(incr::kBlock), 0, stream, *p, tileQueue, numTiles, tilesX);
    return hipGetLastError();
}
