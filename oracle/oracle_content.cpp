// TEST INFRASTRUCTURE -- CPU oracle of the texture-ingest GPU step.  Parity unpinned (the reference ships no fixtures).
// Restates shaders/content-pipeline/prefilter-normal-roughness.hlsl as dispatched by FScene::PrefilterNormalRoughnessTextures
// (src/scene.cpp:1157-1320): one pass per mip, every destination texel looping over its whole base-level footprint with a
// single running sum, exactly as the shader does.  Only tests/, __graft_entry__.smoke() and bench.py may use this file.
#include "oracle_math.h"

#include <cstdint>

using namespace orc;

namespace {

// float4 -> R8G8B8A8_UNORM UAV store (NaN -> 0, clamp, round to nearest)
uint32_t to_unorm8(float v) {
    if (!(v > 0.f)) return 0u;
    if (v > 1.f) v = 1.f;
    return (uint32_t)(v * 255.f + 0.5f);
}
// Texture2D<unorm> load: c / 255 written as the product the texture decode uses everywhere else in this oracle
float un(uint32_t c) { return (float)c * (1.f / 255.f); }

} // namespace

extern "C" {

// backend-d3d12.h:456-471, bCompressed = true
uint32_t orc_compressed_mip_count(uint32_t w, uint32_t h) { uint32_t n = 0; while (w >= 4 && h >= 4) { ++n; w >>= 1; h >>= 1; } return n; }

// outN / outMR: all levels packed.  Returns the number of levels written.
uint32_t orc_prefilter_normal_roughness(const uint32_t* srcN, const uint32_t* srcMR, uint32_t W, uint32_t H, uint32_t* outN, uint32_t* outMR) {
    const uint32_t numMips = orc_compressed_mip_count(W, H);                       // scene.cpp:1229,1239,1285
    uint64_t off = 0;
    for (uint32_t mip = 0; mip < numMips; ++mip) {                                 // scene.cpp:1288-1317
        const uint32_t mw = W >> mip, mh = H >> mip;                               // hlsl:46
        for (uint32_t ty = 0; ty < mh; ++ty) for (uint32_t tx = 0; tx < mw; ++tx) {
            const uint64_t o = off + (uint64_t)ty * mw + tx;
            if (mip == 0) {                                                        // hlsl:38-42
                const uint32_t c = srcMR[(uint64_t)ty * W + tx];
                outN[o] = srcN[(uint64_t)ty * W + tx];
                outMR[o] = (c & 0xff00ff00u) | ((c & 0xffu) << 16) | ((c >> 16) & 0xffu);   // .bgra
                continue;
            }
            float3 rAvg = {0.f, 0.f, 0.f}; float metalnessAvg = 0.f;               // hlsl:51-52
            const uint32_t texelFootprint = 1u << mip;
            const float posx = (float)texelFootprint * ((float)tx + 0.5f), posy = (float)texelFootprint * ((float)ty + 0.5f);
            const float tl = -(float)texelFootprint / 2.f;
            for (uint32_t y = 0; y < texelFootprint; ++y) for (uint32_t x = 0; x < texelFootprint; ++x) {
                const uint32_t sx = (uint32_t)floorf(posx + (tl + (float)x)), sy = (uint32_t)floorf(posy + (tl + (float)y));   // hlsl:58-59
                const uint32_t n = srcN[(uint64_t)sy * W + sx], m = srcMR[(uint64_t)sy * W + sx];
                float3 sampleNormal = {2.f * un(n & 0xffu) - 1.f, 2.f * un((n >> 8) & 0xffu) - 1.f, 2.f * un((n >> 16) & 0xffu) - 1.f};
                sampleNormal = normalize(sampleNormal);                            // hlsl:61
                const float roughnessAlpha = un((m >> 8) & 0xffu), metalness = un((m >> 16) & 0xffu);
                metalnessAvg += metalness;                                         // hlsl:66
                const float invLambda = 0.5f * roughnessAlpha * roughnessAlpha;    // hlsl:71-75
                const float exp2L = pinned_exp_neg(-2.f / invLambda);                // exp(): pinned evaluation (oracle_math.h)
                const float cothLambda = invLambda > 0.1f ? (1.f + exp2L) / (1.f - exp2L) : 1.f;
                const float s = cothLambda - invLambda;
                rAvg.x += s * sampleNormal.x; rAvg.y += s * sampleNormal.y; rAvg.z += s * sampleNormal.z;
            }
            const float count = (float)(texelFootprint * texelFootprint);          // hlsl:79-80
            rAvg = {rAvg.x / count, rAvg.y / count, rAvg.z / count};
            metalnessAvg /= count;
            float r2 = dot(rAvg, rAvg);                                            // hlsl:83-86
            r2 = r2 < 1e-8f ? 1e-8f : (r2 > 1.f ? 1.f : r2);
            const float invLambda = (1.f / sqrtf(r2)) * (1.f - r2) / (3.f - r2);
            const float roughnessAlpha = sqrtf(2.f * invLambda);
            const float3 nn = normalize(rAvg);
            outN[o] = to_unorm8(0.5f * nn.x + 0.5f) | (to_unorm8(0.5f * nn.y + 0.5f) << 8) | (to_unorm8(0.5f * nn.z + 0.5f) << 16);   // hlsl:88
            outMR[o] = to_unorm8(metalnessAvg) | (to_unorm8(roughnessAlpha) << 8);                                                    // hlsl:89
        }
        off += (uint64_t)mw * mh;
    }
    return numMips;
}

} // extern "C"
