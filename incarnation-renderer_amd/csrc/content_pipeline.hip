// content_pipeline.hip -- the GPU step of the reference's texture ingest: vMF prefilter of a normal map + its
// metallic-roughness map into full mip chains (FScene::PrefilterNormalRoughnessTextures, src/scene.cpp:1157-1320;
// shaders/content-pipeline/prefilter-normal-roughness.hlsl).
//
// The reference dispatches one pass per mip and lets every destination texel of mip k loop over its 2^k x 2^k footprint of the
// BASE level, i.e. it re-reads the whole base image once per level (O(texels * mips)).  Here the per-texel vMF vectors r are
// summed ONCE into a float4 pyramid (level k = sum of the four level k-1 sums, children added in the reference's y-outer /
// x-inner order), so every base texel is read once and the whole chain costs ~ (8 B read + 8/3 B write + 32/3 B pyramid) per
// base texel.  Level 1 adds its four texels in exactly the reference's order; deeper levels differ from the reference's single
// running sum by float re-association only (<= 1 LSB of the RGBA8 result, the tolerance tests/test_gpu_content.py states).
#include <hip/hip_runtime.h>

#include <cstring>

#include "api_internal.h"
#include "pt_device.h"

using namespace incr;

namespace {

struct VmfLevel { uint32_t w, h; uint64_t texelOffset; };

// float4 -> R8G8B8A8_UNORM UAV store: NaN -> 0, clamp, round to nearest (D3D11.3 functional spec 3.2.3.6 FLOAT -> UNORM)
DI uint32_t to_unorm8(float v) {
    if (!(v > 0.f)) return 0u;
    if (v > 1.f) v = 1.f;
    return (uint32_t)(v * 255.f + 0.5f);
}

// prefilter-normal-roughness.hlsl:60-75 -- one base texel as a vMF mean vector r (xyz) + its metalness (w)
DI float4 vmf_texel(uint32_t n, uint32_t mr) {
    const float inv255 = 1.f / 255.f;
    f3 sn = {2.f * ((float)(n & 0xffu) * inv255) - 1.f, 2.f * ((float)((n >> 8) & 0xffu) * inv255) - 1.f, 2.f * ((float)((n >> 16) & 0xffu) * inv255) - 1.f};
    sn = normalize(sn);
    const float roughnessAlpha = (float)((mr >> 8) & 0xffu) * inv255;
    const float metalness = (float)((mr >> 16) & 0xffu) * inv255;
    const float invLambda = 0.5f * roughnessAlpha * roughnessAlpha;
    const float exp2L = pinned_exp_neg(-2.f / invLambda);
    const float cothLambda = invLambda > 0.1f ? (1.f + exp2L) / (1.f - exp2L) : 1.f;
    const float s = cothLambda - invLambda;
    return make_float4(s * sn.x, s * sn.y, s * sn.z, metalness);
}

DI float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// :80-91 -- footprint sum -> averaged normal + roughness, stored as the two UAV texels
DI void vmf_store(float4 sum, float count, uint32_t* __restrict__ outN, uint32_t* __restrict__ outMR) {
    const f3 rAvg = {sum.x / count, sum.y / count, sum.z / count};
    const float metalnessAvg = sum.w / count;
    float r2 = dot(rAvg, rAvg);
    r2 = r2 < 1e-8f ? 1e-8f : (r2 > 1.f ? 1.f : r2);
    const float invLambda = (1.f / sqrtf(r2)) * (1.f - r2) / (3.f - r2);
    const float roughnessAlpha = sqrtf(2.f * invLambda);
    const f3 nn = normalize(rAvg);
    const f3 normal = {0.5f * nn.x + 0.5f, 0.5f * nn.y + 0.5f, 0.5f * nn.z + 0.5f};
    *outN = to_unorm8(normal.x) | (to_unorm8(normal.y) << 8) | (to_unorm8(normal.z) << 16);          // float4(normal, 0)
    *outMR = to_unorm8(metalnessAvg) | (to_unorm8(roughnessAlpha) << 8);                              // float4(metal, rough, 0, 0)
}

// mip 0 (:38-42): normal map copied, metallic-roughness stored .bgra
__global__ __launch_bounds__(256) void vmf_mip0_kernel(const uint32_t* __restrict__ srcN, const uint32_t* __restrict__ srcMR,
                                                       uint32_t* __restrict__ outN, uint32_t* __restrict__ outMR, uint64_t texels) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= texels) return;
    outN[i] = srcN[i];
    const uint32_t c = srcMR[i];
    outMR[i] = (c & 0xff00ff00u) | ((c & 0xffu) << 16) | ((c >> 16) & 0xffu);
}

// mip 1: four base texels per thread, added in the reference's loop order (y outer, x inner)
__global__ __launch_bounds__(256) void vmf_mip1_kernel(const uint32_t* __restrict__ srcN, const uint32_t* __restrict__ srcMR, uint32_t W,
                                                       VmfLevel L, float4* __restrict__ sums, uint32_t* __restrict__ outN, uint32_t* __restrict__ outMR) {
    const uint32_t x = blockIdx.x * 32u + (threadIdx.x & 31u), y = blockIdx.y * 8u + (threadIdx.x >> 5);
    if (x >= L.w || y >= L.h) return;
    const uint64_t b0 = (uint64_t)(2u * y) * W + 2u * x, b1 = b0 + W;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    s = add4(s, vmf_texel(srcN[b0], srcMR[b0])); s = add4(s, vmf_texel(srcN[b0 + 1], srcMR[b0 + 1]));
    s = add4(s, vmf_texel(srcN[b1], srcMR[b1])); s = add4(s, vmf_texel(srcN[b1 + 1], srcMR[b1 + 1]));
    const uint64_t o = (uint64_t)y * L.w + x;
    if (sums) sums[o] = s;
    vmf_store(s, 4.f, outN + L.texelOffset + o, outMR + L.texelOffset + o);
}

// mip k >= 2: the four level k-1 sums under this texel, same child order
__global__ __launch_bounds__(256) void vmf_level_kernel(const float4* __restrict__ prev, uint32_t prevW, VmfLevel L, float count,
                                                        float4* __restrict__ sums, uint32_t* __restrict__ outN, uint32_t* __restrict__ outMR) {
    const uint32_t x = blockIdx.x * 32u + (threadIdx.x & 31u), y = blockIdx.y * 8u + (threadIdx.x >> 5);
    if (x >= L.w || y >= L.h) return;
    const uint64_t b0 = (uint64_t)(2u * y) * prevW + 2u * x, b1 = b0 + prevW;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    s = add4(s, prev[b0]); s = add4(s, prev[b0 + 1]); s = add4(s, prev[b1]); s = add4(s, prev[b1 + 1]);
    const uint64_t o = (uint64_t)y * L.w + x;
    if (sums) sums[o] = s;
    vmf_store(s, count, outN + L.texelOffset + o, outMR + L.texelOffset + o);
}

// Exact-order variant (incr_prefilter_normal_roughness_desc::reserved = 1): every destination texel walks its 2^k x 2^k footprint of
// the BASE level with ONE running sum, y outer / x inner, exactly as the shader (and the oracle) does -- a float sum cannot be
// re-associated without changing bits, so a texel of level k is 4^k dependent additions.  What CAN run in parallel does: the vMF
// vector of every base texel is evaluated once (vmf_base_kernel) instead of once per level, and ONE launch sums all levels, the
// long footprints first (a 4 x 4 top level of a 1024^2 pair is 16 lanes x 65,536 additions; the lower levels fill the machine
// meanwhile; several pairs run on parallel streams at scene creation).  Round 2 launched one kernel per level that re-evaluated
// the base texels inside the loop: 0.8 s per Sponza-stat scene; this form takes a few milliseconds.  The scene's textures are built
// with it: the raster passes sample these levels and their G-buffers are compared with the oracle's bit for bit.
__global__ __launch_bounds__(256) void vmf_base_kernel(const uint32_t* __restrict__ srcN, const uint32_t* __restrict__ srcMR, float4* __restrict__ base, uint64_t texels) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i < texels) base[i] = vmf_texel(srcN[i], srcMR[i]);
}
struct VmfLevels { VmfLevel L[16]; uint32_t first[17]; uint32_t mips; };      // first[k]: threads in front of level k when the levels are walked top-down (first[mips - 1] = 0)
__global__ __launch_bounds__(64) void vmf_levels_exact_kernel(const float4* __restrict__ base, uint32_t W, VmfLevels V, uint32_t total,
                                                               uint32_t* __restrict__ outN, uint32_t* __restrict__ outMR) {
    const uint32_t t = blockIdx.x * 64u + threadIdx.x;
    if (t >= total) return;
    uint32_t mip = V.mips - 1u;
    while (mip > 1u && t >= V.first[mip - 1u]) --mip;          // top level first: its threads are the longest
    const VmfLevel L = V.L[mip];
    const uint32_t local = t - V.first[mip];
    const uint32_t y = local / L.w, x = local - y * L.w;
    const uint32_t fp = 1u << mip;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (uint32_t yy = 0; yy < fp; ++yy) {
        const float4* __restrict__ row = base + (uint64_t)(y * fp + yy) * W + (uint64_t)x * fp;
        if (fp >= 8u) {
            for (uint32_t xx = 0; xx < fp; xx += 8u) {         // eight loads in flight, eight additions in the shader's order
                const float4 a0 = row[xx], a1 = row[xx + 1], a2 = row[xx + 2], a3 = row[xx + 3], a4 = row[xx + 4], a5 = row[xx + 5], a6 = row[xx + 6], a7 = row[xx + 7];
                s = add4(s, a0); s = add4(s, a1); s = add4(s, a2); s = add4(s, a3); s = add4(s, a4); s = add4(s, a5); s = add4(s, a6); s = add4(s, a7);
            }
        } else {
            for (uint32_t xx = 0; xx < fp; ++xx) s = add4(s, row[xx]);
        }
    }
    const uint64_t o = (uint64_t)y * L.w + x;
    vmf_store(s, (float)(fp * fp), outN + L.texelOffset + o, outMR + L.texelOffset + o);
}

uint32_t compressed_mip_count(uint32_t w, uint32_t h) { uint32_t n = 0; while (w >= 4 && h >= 4) { ++n; w >>= 1; h >>= 1; } return n; }   // CalcMipCount(w, h, true), backend-d3d12.h:456-471

} // namespace

extern "C" {

uint32_t incr_compressed_mip_count(uint32_t width, uint32_t height) { return compressed_mip_count(width, height); }

uint64_t incr_mip_chain_texels(uint32_t width, uint32_t height, uint32_t num_mips) {
    uint64_t t = 0;
    for (uint32_t k = 0; k < num_mips; ++k) t += (uint64_t)(width >> k) * (height >> k);
    return t;
}

int incr_prefilter_normal_roughness(incr_context* ctx, const incr_prefilter_normal_roughness_desc* d, incr_stream stream) {
    INCR_RANGE();
    if (!ctx) return fail(INCR_ERR_NOT_INITIALIZED, "context is NULL");
    if (!d || !d->normalmap || !d->metallic_roughness || !d->out_normalmap || !d->out_metallic_roughness)
        return fail(INCR_ERR_INVALID_ARGUMENT, "NULL argument");
    const uint32_t W = d->width, H = d->height;
    if (W > 16384u || H > 16384u) return fail(INCR_ERR_INVALID_ARGUMENT, "texture %ux%u larger than 16384", W, H);
    const uint32_t mips = compressed_mip_count(W, H);
    if (!mips) return fail(INCR_ERR_INVALID_ARGUMENT, "texture %ux%u is below the 4x4 block size (no mip to write)", W, H);
    if (d->num_mips != mips) return fail(INCR_ERR_INVALID_ARGUMENT, "num_mips %u, expected incr_compressed_mip_count = %u", d->num_mips, mips);
    if (((uintptr_t)d->normalmap | (uintptr_t)d->metallic_roughness | (uintptr_t)d->out_normalmap | (uintptr_t)d->out_metallic_roughness) & 3u)
        return fail(INCR_ERR_INVALID_ARGUMENT, "texel pointers must be 4-byte aligned");
    (void)hipSetDevice(ctx->device);
    hipStream_t s = (hipStream_t)stream;
    const uint32_t* srcN = (const uint32_t*)d->normalmap; const uint32_t* srcMR = (const uint32_t*)d->metallic_roughness;
    uint32_t* outN = (uint32_t*)d->out_normalmap; uint32_t* outMR = (uint32_t*)d->out_metallic_roughness;

    VmfLevel L[16]; uint64_t off = 0, sumTexels = 0; uint64_t sumOff[16] = {0};
    for (uint32_t k = 0; k < mips; ++k) {
        L[k] = {W >> k, H >> k, off}; off += (uint64_t)L[k].w * L[k].h;
        if (k >= 1 && k + 1 < mips) { sumOff[k] = sumTexels; sumTexels += (uint64_t)L[k].w * L[k].h; }    // the last level's sums are never read
    }
    const uint64_t texels = (uint64_t)W * H;
    vmf_mip0_kernel<<<dim3((uint32_t)((texels + 255) / 256)), dim3(256), 0, s>>>(srcN, srcMR, outN, outMR, texels);
    HIP_TRY(hipGetLastError());
    if (mips == 1) return INCR_OK;
    if (d->reserved == 1u) {      // exact order; stream-ordered from the scratch allocation to its release: the caller synchronises
        float4* baseV = nullptr;
        HIP_TRY(hipMallocAsync((void**)&baseV, texels * sizeof(float4), s));
        vmf_base_kernel<<<dim3((uint32_t)((texels + 255) / 256)), dim3(256), 0, s>>>(srcN, srcMR, baseV, texels);
        VmfLevels V; V.mips = mips;
        uint32_t total = 0;
        for (uint32_t m = mips; m-- > 1u;) { V.L[m] = L[m]; V.first[m] = total; total += L[m].w * L[m].h; }
        V.L[0] = L[0]; V.first[0] = total;
        vmf_levels_exact_kernel<<<dim3((total + 63u) / 64u), dim3(64), 0, s>>>(baseV, W, V, total, outN, outMR);
        hipError_t e = hipGetLastError();
        const hipError_t e2 = hipFreeAsync(baseV, s);
        if (e == hipSuccess) e = e2;
        if (e != hipSuccess) return fail(INCR_ERR_HIP, "prefilter_normal_roughness: %s", hipGetErrorString(e));
        return INCR_OK;
    }
    float4* sums = nullptr;
    if (sumTexels) HIP_TRY(hipMalloc((void**)&sums, sumTexels * sizeof(float4)));
    vmf_mip1_kernel<<<dim3((L[1].w + 31) / 32, (L[1].h + 7) / 8), dim3(256), 0, s>>>(srcN, srcMR, W, L[1], mips > 2 ? sums + sumOff[1] : nullptr, outN, outMR);
    hipError_t e = hipGetLastError();
    for (uint32_t k = 2; k < mips && e == hipSuccess; ++k) {
        vmf_level_kernel<<<dim3((L[k].w + 31) / 32, (L[k].h + 7) / 8), dim3(256), 0, s>>>(sums + sumOff[k - 1], L[k - 1].w, L[k], (float)(1u << k) * (float)(1u << k),
                                                                                         k + 1 < mips ? sums + sumOff[k] : nullptr, outN, outMR);
        e = hipGetLastError();
    }
    // the pyramid is scratch: wait for the chain, then release it (a load-time step; the reference blocks on a read-back here, scene.cpp:1330-1345)
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (sums) (void)hipFree(sums);
    if (e != hipSuccess) return fail(INCR_ERR_HIP, "prefilter_normal_roughness: %s", hipGetErrorString(e));
    return INCR_OK;
}

} // extern "C"
