// pt_shade.h -- next-event-estimation pieces shared by the wavefront kernels: GetDirectRadiance / GetSkyRadiance
// (shaders/lighting/common.hlsli:8-202) split into "which shadow ray" (light_setup) and "how much light"
// (light_shade / sky_shade) with identical arithmetic, the light-loop fetch, and the typed-UAV pixel store +
// fused integrate pass (shaders/raytracing/pathtrace-integrate.hlsl:16-31).
#pragma once
#include "pt_traverse.h"
#include "ibl_device.h"

namespace incr {

// ---- lighting/common.hlsli:8-143 split into "which ray" and "how much light" (identical arithmetic) ----
struct LightSample { f3 L; f3 radianceIn; float NoL; bool needsShadow; };

DI LightSample light_setup(const incr_light& light, const float* __restrict__ m, bool haveTransform, f3 worldPos, f3 N) {
    LightSample s;
    s.L = {0.f, 0.f, 0.f};
    s.radianceIn = {0.f, 0.f, 0.f};
    const f3 color = {light.color[0], light.color[1], light.color[2]};
    float mm[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) mm[i] = haveTransform ? m[i] : 0.f;
    if (light.type == INCR_LIGHT_DIRECTIONAL) {
        s.L = normalize(mk3((0.f * mm[0] + 0.f * mm[4]) + -1.f * mm[8], (0.f * mm[1] + 0.f * mm[5]) + -1.f * mm[9], (0.f * mm[2] + 0.f * mm[6]) + -1.f * mm[10]));
        s.radianceIn = 100 * light.intensity * color;
    } else if (light.type == INCR_LIGHT_POINT) {
        const f3 lightPosition = {mm[12], mm[13], mm[14]};
        const f3 lightVec = lightPosition - worldPos;
        const float distanceSquared = dot(lightVec, lightVec);
        const float distance = sqrtf(distanceSquared);
        s.L = lightVec / distance;
        float radialAttenuation;
        if (light.range > 0.f) radialAttenuation = QDIV(hmax(hmin(1.f, 1.f - pow4(QDIV(distance, light.range))), 0.f), distanceSquared);
        else radialAttenuation = QDIV(1.f, hmax(distanceSquared, 0.0001f));
        s.radianceIn = 10000 * light.intensity * color * radialAttenuation;
    } else if (light.type == INCR_LIGHT_SPOT) {
        const f3 spotPosition = {mm[12], mm[13], mm[14]};
        const f3 spotDirection = normalize(mk3((0.f * mm[0] + 0.f * mm[4]) + -1.f * mm[8], (0.f * mm[1] + 0.f * mm[5]) + -1.f * mm[9], (0.f * mm[2] + 0.f * mm[6]) + -1.f * mm[10]));
        const f3 lightVec = spotPosition - worldPos;
        const float distanceSquared = dot(lightVec, lightVec);
        const float distance = sqrtf(distanceSquared);
        s.L = lightVec / distance;
        float angularAttenuation = 0.f;
        const float cd = dot(spotDirection, s.L);
        const float cosineOuterAngle = cosf(light.spotAngles[1]);
        if (cd > cosineOuterAngle) {
            const float cosineInnerAngle = cosf(light.spotAngles[0]);
            const float lightAngleScale = QDIV(1.f, hmax(0.001f, cosineInnerAngle - cosineOuterAngle));
            const float lightAngleOffset = -cosineOuterAngle * lightAngleScale;
            angularAttenuation = saturate(cd * lightAngleScale + lightAngleOffset);
            angularAttenuation *= angularAttenuation;
        }
        float radialAttenuation = 0.f;
        if (angularAttenuation > 0.f) {
            if (light.range > 0.f) { /* Q9: shadowed redeclaration at common.hlsli:73 leaves 0 */ }
            else radialAttenuation = QDIV(1.f, hmax(distanceSquared, 0.0001f));
        }
        s.radianceIn = 10000 * light.intensity * color * angularAttenuation * radialAttenuation;
    }
    s.NoL = saturate(dot(N, s.L));
    s.needsShadow = s.NoL > 0.f && (s.radianceIn.x > 0.f || s.radianceIn.y > 0.f || s.radianceIn.z > 0.f);
    return s;
}

// ---- sun cover map (pt_params.h SunCover) ----
// heights as unsigned keys that order like the floats (atomicMax); 0 = below everything
DI uint32_t height_key(float h) { const uint32_t b = __float_as_uint(h); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
// true: nothing of the scene lies above `p` toward the sun beyond the query's tmin -- the shadow ray from p is unoccluded
DI bool sun_cover_clear(const SunCoverHeader& H, const uint32_t* __restrict__ grid, f3 p, float tmin) {
    if (!H.valid) return false;
    const float u = ((p.x * H.U[0] + p.y * H.U[1]) + p.z * H.U[2] - H.u0) * H.invCell, v = ((p.x * H.V[0] + p.y * H.V[1]) + p.z * H.V[2] - H.v0) * H.invCell;
    if (!(u >= 0.f && v >= 0.f && u < (float)kSunCoverG && v < (float)kSunCoverG)) return false;
    const float h = (p.x * H.L[0] + p.y * H.L[1]) + p.z * H.L[2];
    return grid[(size_t)(uint32_t)v * kSunCoverG + (uint32_t)u] < height_key(h + tmin - 2.f * H.eps);
}
DI bool sun_cover_clear(const SunCover& C, f3 p, float tmin) { return sun_cover_clear(*C.header, C.grid, p, tmin); }

// the map's light?  Its direction, bit for bit as light_setup gives it for a directional light
DI bool sun_cover_is_light(const SunCoverHeader& H, const incr_light& light, f3 L) {
    return light.type == INCR_LIGHT_DIRECTIONAL && L.x == H.L[0] && L.y == H.L[1] && L.z == H.L[2];
}

DI f3 light_shade(const LightSample& s, f3 basecolor, float metallic, float roughness, f3 N, f3 V) {     // :118-139, lightVisibility == 1
    const float NoV = saturate(dot(N, V));
    const f3 F0 = metallic * basecolor + (1.f - metallic) * 0.04f;
    const f3 albedo = (1.f - metallic) * basecolor;
    const f3 H = normalize(s.L + V);
    const float NoH = saturate(dot(N, H));
    const float VoH = saturate(dot(V, H));
    const float D = GGX(NoH, roughness);
    const f3 F = F_Schlick(VoH, F0);
    const float G = G_SmithGGXCorrelated(NoV, s.NoL, roughness);
    const f3 Fd = albedo * K_INV_PI;
    const f3 Fr = (D * F * G) / hmax(4.f * NoV * s.NoL, 0.001f);
    const f3 irradiance = s.radianceIn * s.NoL;
    return (Fr + (1.f - F) * Fd) * irradiance * 1.f;
}

DI f3 sky_shade(f3 radianceIn, f3 L, float NoL, const MatProps& matInfo, f3 N, f3 V) {                   // :177-198, lightVisibility == 1
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
    return (Fr + (1.f - F) * Fd) * irradiance * 1.f;
}

DI void load_light(const PtParams& P, int lightIndex, incr_light& light, const float*& m, bool& haveT) {
    const int lightId = (uint32_t)lightIndex < P.numSceneLights ? P.lightIndices[lightIndex] : 0;     // D3D robust access: OOB reads 0
    light = {};
    if ((uint32_t)lightId < P.numGlobalLights) light = P.globalLights[lightId];
    haveT = (uint32_t)lightId < P.numSceneLights;
    m = P.lightTransforms + (haveT ? (size_t)lightId * 16 : 0);
}

// The same fetch out of the RESOLVED light table (api.hip build_resolved_lights: load_light's result per loop index, 5 x float4):
// [0] type, range, spotAngles  [1] color, intensity  [2] m0 m1 m2 m4  [3] m5 m6 m8 m9  [4] m10 m12 m13 m14 -- the matrix entries
// light_setup reads; a light without a transform has them zero, as light_setup's `haveTransform ? m[i] : 0` makes them.
template <class Table> DI void load_light_resolved(Table table, uint32_t lightIndex, incr_light& light, float mm[16]) {
    const float4 r0 = table[lightIndex * 5u], r1 = table[lightIndex * 5u + 1u], r2 = table[lightIndex * 5u + 2u], r3 = table[lightIndex * 5u + 3u], r4 = table[lightIndex * 5u + 4u];
    light.type = (int32_t)__float_as_uint(r0.x); light.range = r0.y; light.spotAngles[0] = r0.z; light.spotAngles[1] = r0.w;
    light.color[0] = r1.x; light.color[1] = r1.y; light.color[2] = r1.z; light.intensity = r1.w;
#pragma unroll
    for (int i = 0; i < 16; ++i) mm[i] = 0.f;
    mm[0] = r2.x; mm[1] = r2.y; mm[2] = r2.z; mm[4] = r2.w; mm[5] = r3.x; mm[6] = r3.y; mm[8] = r3.z; mm[9] = r3.w;
    mm[10] = r4.x; mm[12] = r4.y; mm[13] = r4.z; mm[14] = r4.w;
}

// destUav[xy] = payload.color (pathtracing.hlsl:118) + the integrate pass on this pixel (pathtrace-integrate.hlsl:16-31).
// s = the path's sample within the call (0 unless incr_pathtrace_desc::num_samples > 1).
DI f3 integrate_sample(const PtParams& P, f3 history, f3 cur, uint32_t sampleIdx) {
    const float n = (float)sampleIdx;
    if (sampleIdx <= 1) return cur / n;
    return ((n - 1.f) * history + cur) / n;
}
DI void store_pixel(const PtParams& P, uint32_t x, uint32_t ly, uint32_t s, f4 c) {
    const size_t li = (size_t)ly * P.W + x;
    const uint32_t hx = f32_to_f16_bits(c.x, P.storageRounding), hy = f32_to_f16_bits(c.y, P.storageRounding);
    const uint32_t hz = f32_to_f16_bits(c.z, P.storageRounding), hw = f32_to_f16_bits(c.w, P.storageRounding);
    if (P.numSamples > 1u) { P.sampleStage[(size_t)s * P.localPixels + li] = make_uint2(hx | (hy << 16), hz | (hw << 16)); return; }      // integrated in sample order later
    if (P.target) *(uint2*)(P.target + li * 4) = make_uint2(hx | (hy << 16), hz | (hw << 16));
    if (P.history) {
        const f3 cur = {f16_bits_to_f32(hx), f16_bits_to_f32(hy), f16_bits_to_f32(hz)};
        const f3 avg = integrate_sample(P, P.sampleIdx <= 1 ? mk3(0.f) : unpack_r11g11b10(P.history[li]), cur, P.sampleIdx);
        P.history[li] = pack_r11g11b10(avg, P.storageRounding);
    }
}

} // namespace incr
