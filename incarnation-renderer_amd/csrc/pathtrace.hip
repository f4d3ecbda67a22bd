// pathtrace.hip -- the path-tracing integrator as a HIP kernel for gfx950 (MI355X).
//
// Replaces, for RenderJob::PathTracing::Execute (reference: source/demo-dll/src/render-jobs/path-tracing.inl:17-299):
//   DispatchRays(W,H,1) of rgsMain / chsMain / ahsMain / msDynamicSky   (shaders/raytracing/pathtracing.hlsl:89-319)
//   + the fixed-function DXR traversal and ray/triangle test              
//   + Dispatch(ceil(W/16), ceil(H/16)) of pathtrace-integrate cs_main     (shaders/raytracing/pathtrace-integrate.hlsl:16-31)
// in ONE launch: the TraceRay recursion of chsMain is a tail call, so it is run as a loop that carries
// {color, attenuation, pathLength, sampleSetIndex}; the running-mean integrate is fused into the
// epilogue (the thread that produced the RGBA16F texel is the one that reads it back).
//
// Mapping to CDNA4: one lane = one pixel path; a 64-lane wavefront covers an 8x8 pixel tile so primary
// rays of a wave walk the same BVH lines; 256-thread workgroups (4 waves = a 16x16 tile).  Traversal is a
// 4-wide BVH (one 128-B line per node, one 48-B record per candidate triangle) with the per-lane
// traversal stack staged in LDS ([entry][lane] layout: conflict-free ds_read/ds_write_b32) and a
// private-memory overflow that is only touched by pathological rays.
#include "pt_traverse.h"
#include "ibl_device.h"
#include <algorithm>

namespace incr {


// lighting/common.hlsli:8-143 (PATH_TRACING variant: every light traces a shadow ray)
DI f3 GetDirectRadiance(const PtParams& P, LaneStack& st, const incr_light& light, const float* __restrict__ m, bool haveTransform,
                        f3 worldPos, f3 basecolor, float metallic, float roughness, f3 N, f3 V, Counters& cnt) {
    f3 L = {0.f, 0.f, 0.f};
    f3 radianceIn = {0.f, 0.f, 0.f};
    const f3 color = {light.color[0], light.color[1], light.color[2]};
    float mm[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) mm[i] = haveTransform ? m[i] : 0.f;     // out-of-range transform loads read 0 (D3D robust access)
    if (light.type == INCR_LIGHT_DIRECTIONAL) {
        L = normalize(mk3((0.f * mm[0] + 0.f * mm[4]) + -1.f * mm[8], (0.f * mm[1] + 0.f * mm[5]) + -1.f * mm[9], (0.f * mm[2] + 0.f * mm[6]) + -1.f * mm[10]));
        radianceIn = 100 * light.intensity * color;
    } else if (light.type == INCR_LIGHT_POINT) {
        const f3 lightPosition = {mm[12], mm[13], mm[14]};
        const f3 lightVec = lightPosition - worldPos;
        const float distanceSquared = dot(lightVec, lightVec);
        const float distance = sqrtf(distanceSquared);
        L = lightVec / distance;
        float radialAttenuation;
        if (light.range > 0.f) radialAttenuation = hmax(hmin(1.f, 1.f - pow4(distance / light.range)), 0.f) / distanceSquared;
        else radialAttenuation = 1.f / hmax(distanceSquared, 0.0001f);
        radianceIn = 10000 * light.intensity * color * radialAttenuation;
    } else if (light.type == INCR_LIGHT_SPOT) {
        const f3 spotPosition = {mm[12], mm[13], mm[14]};
        const f3 spotDirection = normalize(mk3((0.f * mm[0] + 0.f * mm[4]) + -1.f * mm[8], (0.f * mm[1] + 0.f * mm[5]) + -1.f * mm[9], (0.f * mm[2] + 0.f * mm[6]) + -1.f * mm[10]));
        const f3 lightVec = spotPosition - worldPos;
        const float distanceSquared = dot(lightVec, lightVec);
        const float distance = sqrtf(distanceSquared);
        L = lightVec / distance;
        float angularAttenuation = 0.f;
        const float cd = dot(spotDirection, L);
        const float cosineOuterAngle = cosf(light.spotAngles[1]);
        if (cd > cosineOuterAngle) {
            const float cosineInnerAngle = cosf(light.spotAngles[0]);
            const float lightAngleScale = 1.f / hmax(0.001f, cosineInnerAngle - cosineOuterAngle);
            const float lightAngleOffset = -cosineOuterAngle * lightAngleScale;
            angularAttenuation = saturate(cd * lightAngleScale + lightAngleOffset);
            angularAttenuation *= angularAttenuation;
        }
        float radialAttenuation = 0.f;
        if (angularAttenuation > 0.f) {
            if (light.range > 0.f) { /* Q9: shadowed redeclaration at common.hlsli:73 leaves 0 */ }
            else radialAttenuation = 1.f / hmax(distanceSquared, 0.0001f);
        }
        radianceIn = 10000 * light.intensity * color * angularAttenuation * radialAttenuation;
    }
    f3 radianceOut = {0.f, 0.f, 0.f};
    const float NoL = saturate(dot(N, L));
    if (NoL > 0.f && (radianceIn.x > 0.f || radianceIn.y > 0.f || radianceIn.z > 0.f)) {
        float lightVisibility = 1.f;
        cnt.lightShadow++;
        if (trace_any(P, st, worldPos, L, 0.1f, 1000.f)) lightVisibility = 0.f;
        if (lightVisibility > 0.f) {
            const float NoV = saturate(dot(N, V));
            const f3 F0 = metallic * basecolor + (1.f - metallic) * 0.04f;
            const f3 albedo = (1.f - metallic) * basecolor;
            const f3 H = normalize(L + V);
            const float NoH = saturate(dot(N, H));
            const float VoH = saturate(dot(V, H));
            const float D = GGX(NoH, roughness);
            const f3 F = F_Schlick(VoH, F0);
            const float G = G_SmithGGXCorrelated(NoV, NoL, roughness);
            const f3 Fd = albedo * K_INV_PI;
            const f3 Fr = (D * F * G) / hmax(4.f * NoV * NoL, 0.001f);
            const f3 irradiance = radianceIn * NoL;
            radianceOut = (Fr + (1.f - F) * Fd) * irradiance * lightVisibility;
        }
    }
    return radianceOut;
}

// lighting/common.hlsli:145-202
DI f3 GetSkyRadiance(const PtParams& P, LaneStack& st, f3 radianceIn, f3 L, f3 worldPos, const MatProps& matInfo, f3 N, f3 V, Counters& cnt) {
    f3 radianceOut = {0.f, 0.f, 0.f};
    const float NoL = saturate(dot(N, L));
    if (NoL > 0.f && (radianceIn.x > 0.f || radianceIn.y > 0.f || radianceIn.z > 0.f)) {
        float lightVisibility = 1.f;
        cnt.skyShadow++;
        if (trace_any(P, st, worldPos, L, 0.1f, 1000.f)) lightVisibility = 0.f;
        if (lightVisibility > 0.f) {
            const float NoV = dot(N, V);
            const f3 F0 = matInfo.metallic * matInfo.basecolor + (1.f - matInfo.metallic) * 0.04f;
            const f3 albedo = (1.f - matInfo.metallic) * matInfo.basecolor;
            const f3 H = normalize(L + V);
            const float NoH = hmax(0.0001f, dot(N, H));
            const float VoH = hmax(0.0001f, dot(V, H));
            const float D = GGX(NoH, matInfo.roughness);
            const f3 F = F_Schlick(VoH, F0);
            const float G = G_Smith_Direct(NoV, NoL, matInfo.roughness);
            const f3 Fd = albedo * K_INV_PI;
            const f3 Fr = (D * F * G) / hmax(4.f * NoV * NoL, 0.001f);
            const f3 irradiance = radianceIn * NoL;
            radianceOut = (Fr + (1.f - F) * Fd) * irradiance * lightVisibility;
        }
    }
    return radianceOut / CosineHemispherePdf(L.z);
}


DI f4 trace_pixel(const PtParams& P, LaneStack& st, uint32_t px, uint32_t py, Counters& cnt) {
    // rgsMain -- pathtracing.hlsl:89-119
    uint32_t raygenSampleSetIdx = 0;
    const uint32_t pixelIdx = py * P.W + px;
    const f2 subPixelJitter = SamplePoint(pixelIdx, P.sampleIdx, raygenSampleSetIdx, P.sqrtSampleCount, P.numPixels);
    const f2 apertureSample = SamplePoint(pixelIdx, P.sampleIdx, raygenSampleSetIdx, P.sqrtSampleCount, P.numPixels);
    Ray ray = GenerateCameraRay(P, {(float)px + subPixelJitter.x, (float)py + subPixelJitter.y}, apertureSample);

    f4 color = {0.f, 0.f, 0.f, 0.f};
    int pathLength = 0;
    f3 attenuation = {1.f, 1.f, 1.f};
    uint32_t sampleSetIndex = 0;
    bool first = true;

    for (;;) {
        if (first) cnt.camera++; else cnt.bounce++;
        first = false;
        Hit hit;
        if (!trace_closest(P, st, ray.o, ray.d, ray.tmin, ray.tmax, &hit)) {
            // msEnvmap / msDynamicSky -- pathtracing.hlsl:304-319
            const f3 add = sky_miss_add(P, attenuation, ray.d);
            color.x += add.x; color.y += add.y; color.z += add.z;
            break;
        }
        // chsMain -- pathtracing.hlsl:121-268
        cnt.hits++;
        pathLength += 1;
        // one 128-byte line: the three vertices' normal/tangent/uv + ids (baked GetUint3/GetFloat2/3/4 results)
        const float4* __restrict__ sp = (const float4*)(P.shade + hit.tri);
        const float4 s0 = sp[0], s1 = sp[1], s2 = sp[2], s3 = sp[3], s4 = sp[4], s5 = sp[5], s6 = sp[6];
        const uint4 s7 = *(const uint4*)(sp + 7);
        const f3 vn0 = {s0.x, s0.y, s0.z}, vn1 = {s0.w, s1.x, s1.y}, vn2 = {s1.z, s1.w, s2.x};
        const f4 vt0 = {s2.y, s2.z, s2.w, s3.x}, vt1 = {s3.y, s3.z, s3.w, s4.x}, vt2 = {s4.y, s4.z, s4.w, s5.x};
        const f2 uv0 = {s5.y, s5.z}, uv1 = {s5.w, s6.x}, uv2 = {s6.y, s6.z};
        const uint32_t materialIndex = __float_as_uint(s6.w), meshIndex = s7.x;
        const float* __restrict__ localToWorld = P.meshLocalToWorld + (size_t)meshIndex * 16;

        const f3 hitPosition = ray.o + hit.t * ray.d;
        const float bx = hit.u, by = hit.v;
        const f4 packedT = vt0 + bx * (vt1 - vt0) + by * (vt2 - vt0);
        f3 N = normalize(vn0 + bx * (vn1 - vn0) + by * (vn2 - vn0));
        f3 T = normalize(mk3(packedT.x, packedT.y, packedT.z));
        f3 B = cross(N, T) * packedT.w;
        N = mul_dir(N, localToWorld);
        T = mul_dir(T, localToWorld);
        B = mul_dir(B, localToWorld);
        const f3 Ntbn = N;                                  // float3x3 tangentToWorld = float3x3(T, B, N)  (:166)
        const f2 uv = uv0 + bx * (uv1 - uv0) + by * (uv2 - uv0);
        MatProps matInfo = EvaluateMaterialProperties(P.materials[materialIndex], P.textures, P.srgbLut, uv);

        if (P.viewmode == 1) matInfo.basecolor = mk3(0.5f);
        else if (P.viewmode == 2) { color = {matInfo.roughness, matInfo.roughness, matInfo.roughness, matInfo.roughness}; break; }
        else if (P.viewmode == 3) { color = {matInfo.metallic, matInfo.metallic, matInfo.metallic, matInfo.metallic}; break; }
        else if (P.viewmode == 4) { color = {matInfo.basecolor.x, matInfo.basecolor.y, matInfo.basecolor.z, 1.f}; break; }
        else if (P.viewmode == 5) { color = {matInfo.emissive.x, matInfo.emissive.y, matInfo.emissive.z, 1.f}; break; }

        {
            const f3 add = attenuation * matInfo.emissive * 20000;
            color.x += add.x; color.y += add.y; color.z += add.z;
            attenuation = attenuation * saturate(1.f - matInfo.emissive);
        }
        if (matInfo.bHasNormalmap) {
            const f3 nm = matInfo.normalmap;
            N = normalize(mk3((nm.x * T.x + nm.y * B.x) + nm.z * N.x, (nm.x * T.y + nm.y * B.y) + nm.z * N.y, (nm.x * T.z + nm.y * B.z) + nm.z * N.z));
        }
        const f3 V = -ray.d;

        if (P.directLighting) {
            for (int lightIndex = 0; lightIndex < P.lightCount; ++lightIndex) {
                const int lightId = (uint32_t)lightIndex < P.numSceneLights ? P.lightIndices[lightIndex] : 0;
                incr_light light = {};
                if ((uint32_t)lightId < P.numGlobalLights) light = P.globalLights[lightId];
                const bool haveT = (uint32_t)lightId < P.numSceneLights;
                const f3 r = attenuation * GetDirectRadiance(P, st, light, P.lightTransforms + (haveT ? (size_t)lightId * 16 : 0), haveT,
                                                             hitPosition, matInfo.basecolor, matInfo.metallic, matInfo.roughness, N, V, cnt);
                color.x += r.x; color.y += r.y; color.z += r.z;
            }
        }
        {   // skylight :239-244
            const f2 u = SamplePoint(pixelIdx, P.sampleIdx, sampleSetIndex, P.sqrtSampleCount, P.numPixels);
            const f3 sampleDir = CosineSampleHemisphere(u);
            const f3 radianceIn = P.skyBrightness * sky_env_radiance(P, sampleDir);
            const f3 r = attenuation * GetSkyRadiance(P, st, radianceIn, sampleDir, hitPosition, matInfo, N, V, cnt);
            color.x += r.x; color.y += r.y; color.z += r.z;
        }
        if (pathLength < 4) {
            // GenerateIndirectRadianceRay -- raytracing/common.hlsli:55-123
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
                continue;
            }
        }
        break;
    }
    return color;
}


__global__ __launch_bounds__(kBlock) void pathtrace_kernel(const PtParams P) {
    __shared__ uint32_t s_stack[kStackLds * kBlock];
    const uint32_t tid = threadIdx.x;

    // workgroup = 16x16 pixel tile, wave w = 8x8 quadrant, lane = (lane & 7, lane >> 3) inside it
    const uint32_t tilesX = (P.W + 15u) / 16u;
    const uint32_t tileX = blockIdx.x % tilesX, tileY = blockIdx.x / tilesX;
    const uint32_t wave = tid >> 6, lane = tid & 63u;
    const uint32_t x = tileX * 16u + (wave & 1u) * 8u + (lane & 7u);
    const uint32_t ly = tileY * 16u + (wave >> 1) * 8u + (lane >> 3);
    Counters cnt = {0, 0, 0, 0, 0};
    if (x < P.W && ly < P.localRows) {
        LaneStack st; st.bind(s_stack + tid, nullptr); st.sp = 0;
        const int64_t gy = shard_global_row(P, ly);
        f4 c = {0.f, 0.f, 0.f, 0.f};
        if (gy >= 0) c = trace_pixel(P, st, x, (uint32_t)gy, cnt);
        const size_t li = (size_t)ly * P.W + x;
        // destUav[xy] = payload.color (RGBA16F typed store, pathtracing.hlsl:118)
        const uint32_t hx = f32_to_f16_bits(c.x, P.storageRounding), hy = f32_to_f16_bits(c.y, P.storageRounding);
        const uint32_t hz = f32_to_f16_bits(c.z, P.storageRounding), hw = f32_to_f16_bits(c.w, P.storageRounding);
        if (P.target) *(uint2*)(P.target + li * 4) = make_uint2(hx | (hy << 16), hz | (hw << 16));
        // pathtrace-integrate.hlsl:16-31 fused: history = ((n-1)*history + current) / n
        if (P.history) {
            const f3 cur = {f16_bits_to_f32(hx), f16_bits_to_f32(hy), f16_bits_to_f32(hz)};
            const float n = (float)P.sampleIdx;
            f3 avg;
            if (P.sampleIdx <= 1) avg = cur / n;
            else avg = ((n - 1.f) * unpack_r11g11b10(P.history[li]) + cur) / n;
            P.history[li] = pack_r11g11b10(avg, P.storageRounding);
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

// incr_trace_rays: batch of inline ray queries (debug / RayQuery stand-in); one lane per ray
__global__ __launch_bounds__(kBlock) void trace_rays_kernel(const PtParams P, uint32_t n, const float* __restrict__ origins, const float* __restrict__ dirs,
                                                            float tmin, float tmax, int anyHit, uint32_t* __restrict__ ids, float* __restrict__ tuv) {
    __shared__ uint32_t s_stack[kStackLds * kBlock];
    const uint32_t tid = threadIdx.x;
    const uint32_t i = blockIdx.x * kBlock + tid;
    if (i >= n) return;
    LaneStack st; st.bind(s_stack + tid, nullptr); st.sp = 0;
    const f3 o = {origins[i * 3], origins[i * 3 + 1], origins[i * 3 + 2]}, d = {dirs[i * 3], dirs[i * 3 + 1], dirs[i * 3 + 2]};
    uint4 id = make_uint4(0, 0, 0, 0); f3 r = {0.f, 0.f, 0.f};
    if (anyHit) id.x = trace_any(P, st, o, d, tmin, tmax) ? 1u : 0u;
    else {
#ifdef INCR_TRAV_STATS
        TravClosest T; trav_closest_begin(T, o, d, tmin, tmax); st.reset();
        while (!trav_closest_step(P, st, T)) {}
        id = make_uint4(T.found ? 1u : 0u, T.nNodes, T.nTris, 0u);
        r = {T.best, 0.f, 0.f};
#else
        Hit h;
        if (trace_closest(P, st, o, d, tmin, tmax, &h)) {
            const TriShade* __restrict__ sh = P.shade + h.tri;
            id = make_uint4(1u, sh->instance, sh->geometry, sh->primitive);
            r = {h.t, h.u, h.v};
        }
#endif
    }
    *(uint4*)(ids + (size_t)i * 4) = id;
    tuv[(size_t)i * 3] = r.x; tuv[(size_t)i * 3 + 1] = r.y; tuv[(size_t)i * 3 + 2] = r.z;
}

// TonemapPass: shaders/postprocess/tonemap.hlsl:30-59 with the R8G8B8A8_UNORM_SRGB render-target encode
DI uint32_t linear_to_srgb8(float c) {
    if (!(c > 0.f)) c = 0.f;
    if (c > 1.f) c = 1.f;
    const float e = c <= 0.0031308f ? 12.92f * c : 1.055f * powf(c, 1.f / 2.4f) - 0.055f;
    return (uint32_t)floorf(e * 255.f + 0.5f);
}
__global__ __launch_bounds__(256) void tonemap_kernel(const uint32_t* __restrict__ hdr, uint32_t* __restrict__ ldr, size_t n, int ev100, int viewmode) {
    const float e = Exposure(ev100);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const f3 h = unpack_r11g11b10(hdr[i]);
        f3 l;
        if (viewmode != 0 && viewmode != 1) l = h; else l = ACESFilm(h * e);
        ldr[i] = linear_to_srgb8(l.x) | (linear_to_srgb8(l.y) << 8) | (linear_to_srgb8(l.z) << 16) | 0xff000000u;
    }
}

// un-interleave gathered shards: src [numShards][localRows][W] -> dst [H][W]
__global__ __launch_bounds__(256) void unshard_kernel(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, uint32_t W, uint32_t H,
                                                      uint32_t h, uint32_t numShards, uint32_t localRows, uint32_t texelBytes) {
    const size_t rowBytes = (size_t)W * texelBytes;
    const size_t total = (size_t)H * rowBytes / 4;       // dword granularity (texelBytes is a multiple of 4)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t byte = i * 4;
        const uint32_t y = (uint32_t)(byte / rowBytes); const size_t xo = byte % rowBytes;
        const uint32_t blk = y / h, r = y % h;
        const uint32_t shard = blk % numShards, k = blk / numShards;
        const size_t srcOff = ((size_t)shard * localRows + (size_t)k * h + r) * rowBytes + xo;
        *(uint32_t*)(dst + byte) = *(const uint32_t*)(src + srcOff);
    }
}

} // namespace incr

// ---- launchers (called from api.cpp) -----------------------------------------------------------------
extern "C" hipError_t incr_launch_pathtrace(const incr::PtParams* p, hipStream_t stream) {
    const uint32_t tilesX = (p->W + 15u) / 16u, tilesY = (p->localRows + 15u) / 16u;
    if (tilesX == 0 || tilesY == 0) return hipSuccess;
    hipLaunchKernelGGL(incr::pathtrace_kernel, dim3(tilesX * tilesY), dim3(incr::kBlock), 0, stream, *p);
    return hipGetLastError();
}
extern "C" hipError_t incr_launch_trace_rays(const incr::PtParams* p, uint32_t n, const float* origins, const float* dirs, float tmin, float tmax,
                                             int anyHit, uint32_t* ids, float* tuv, hipStream_t stream) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(incr::trace_rays_kernel, dim3((n + incr::kBlock - 1) / incr::kBlock), dim3(incr::kBlock), 0, stream, *p, n, origins, dirs, tmin, tmax, anyHit, ids, tuv);
    return hipGetLastError();
}
extern "C" hipError_t incr_launch_tonemap(const uint32_t* hdr, uint32_t* ldr, size_t n, int ev100, int viewmode, hipStream_t stream) {
    if (!n) return hipSuccess;
    const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 2048);
    hipLaunchKernelGGL(incr::tonemap_kernel, dim3(blocks), dim3(256), 0, stream, hdr, ldr, n, ev100, viewmode);
    return hipGetLastError();
}
extern "C" hipError_t incr_launch_unshard(uint8_t* dst, const uint8_t* src, uint32_t W, uint32_t H, uint32_t h, uint32_t numShards,
                                          uint32_t localRows, uint32_t texelBytes, hipStream_t stream) {
    const size_t total = (size_t)H * W * texelBytes / 4;
    if (!total) return hipSuccess;
    const unsigned blocks = (unsigned)std::min<size_t>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(incr::unshard_kernel, dim3(blocks), dim3(256), 0, stream, dst, src, W, H, h, numShards, localRows, texelBytes);
    return hipGetLastError();
}
