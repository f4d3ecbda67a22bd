// texture_mips.hip -- mip chains of the scene's textures, built on the device at scene creation.
// Reference: FScene::LoadTexture (src/scene.cpp:1072-1090) calls DirectX::GenerateMipMaps(srcImage, TEX_FILTER_LINEAR, numMips)
// with numMips = levels down to 4x4, then block-compresses (not reproduced: the texture unit here filters RGBA8);
// FScene::PrefilterNormalRoughnessTextures (:1157-1320) replaces the chains of a material's normal + metallic-roughness pair
// by the vMF-filtered ones (content_pipeline.hip).  The raster passes sample these chains (pt_device.h SampleGrad); the
// path tracer keeps SampleLevel(0), as the reference's PATH_TRACING variant does.
//
// Pinned: level l is max(W >> l, 1) x max(H >> l, 1); a texel of level l + 1 is the mean of the 2x2 texels
// of level l under it (the second column / row clamps to the last one for odd sizes) -- DirectXTex's linear filter for a
// 2:1 reduction; sRGB colour is averaged in linear light (decode table of pt_device.h) and re-encoded to the NEAREST table
// entry (search of the same table: no pow, identical on host and device), alpha and the other formats average the bytes.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/incr_types.h"
#include "device_types.h"

namespace incr {

__device__ __forceinline__ uint32_t srgb_encode_nearest(const float* __restrict__ lut, float v) {
    uint32_t lo = 0u, hi = 255u;                         // largest i with lut[i] <= v (lut[0] = 0 <= v)
#pragma unroll
    for (int it = 0; it < 8; ++it) { const uint32_t mid = (lo + hi + 1u) >> 1; if (lut[mid] <= v) lo = mid; else hi = mid - 1u; }
    if (lo < 255u && (lut[lo + 1u] - v) < (v - lut[lo])) ++lo;
    return lo;
}

__global__ __launch_bounds__(256) void mip_box_kernel(const uint32_t* __restrict__ src, uint32_t sw, uint32_t sh, uint32_t* __restrict__ dst,
                                                      uint32_t dw, uint32_t dh, uint32_t format, const float* __restrict__ srgbLut) {
    const uint32_t x = blockIdx.x * 32u + (threadIdx.x & 31u), y = blockIdx.y * 8u + (threadIdx.x >> 5);
    if (x >= dw || y >= dh) return;
    const uint32_t x0 = min(2u * x, sw - 1u), x1 = min(2u * x + 1u, sw - 1u), y0 = min(2u * y, sh - 1u), y1 = min(2u * y + 1u, sh - 1u);
    const uint32_t c[4] = {src[(size_t)y0 * sw + x0], src[(size_t)y0 * sw + x1], src[(size_t)y1 * sw + x0], src[(size_t)y1 * sw + x1]};
    uint32_t out = 0u;
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
        const uint32_t b[4] = {(c[0] >> (8 * ch)) & 0xffu, (c[1] >> (8 * ch)) & 0xffu, (c[2] >> (8 * ch)) & 0xffu, (c[3] >> (8 * ch)) & 0xffu};
        uint32_t r;
        if (format == INCR_TEX_RGBA8_SRGB && ch < 3) r = srgb_encode_nearest(srgbLut, ((srgbLut[b[0]] + srgbLut[b[1]]) + (srgbLut[b[2]] + srgbLut[b[3]])) * 0.25f);
        else r = (uint32_t)((float)(b[0] + b[1] + b[2] + b[3]) * 0.25f + 0.5f);
        out |= r << (8 * ch);
    }
    dst[(size_t)y * dw + x] = out;
}

// levels >= 1 of a vMF-prefiltered pair into the scene's chains: the normal chain as is, the metallic-roughness chain from the
// prefilter's (metal, roughness, 0, 0) back to the glTF channel order the sampler swizzles (roughness in G, metalness in B)
__global__ __launch_bounds__(256) void mip_adopt_vmf_kernel(const uint32_t* __restrict__ vmfN, const uint32_t* __restrict__ vmfMR, uint32_t* __restrict__ dstN,
                                                            uint32_t* __restrict__ dstMR, uint64_t first, uint64_t count) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= count) return;
    dstN[first + i] = vmfN[first + i];
    const uint32_t m = vmfMR[first + i];
    dstMR[first + i] = (m & 0x0000ff00u) | ((m & 0xffu) << 16);
}

// one level of a row-major chain -> the tiled storage the samplers read (device_types.h texel_tile_index); pad texels of edge tiles stay 0
__global__ __launch_bounds__(256) void mip_tile_kernel(const uint32_t* __restrict__ src, uint32_t w, uint32_t h, uint32_t* __restrict__ dst) {
    const uint32_t x = blockIdx.x * 32u + (threadIdx.x & 31u), y = blockIdx.y * 8u + (threadIdx.x >> 5);
    if (x >= w || y >= h) return;
    dst[texel_tile_index(x, y, w)] = src[(size_t)y * w + x];
}

// the inverse, for incr_scene_texture_download: the row-major chains are freed once the tiled copy exists
__global__ __launch_bounds__(256) void mip_untile_kernel(const uint32_t* __restrict__ src, uint32_t w, uint32_t h, uint32_t* __restrict__ dst) {
    const uint32_t x = blockIdx.x * 32u + (threadIdx.x & 31u), y = blockIdx.y * 8u + (threadIdx.x >> 5);
    if (x >= w || y >= h) return;
    dst[(size_t)y * w + x] = src[texel_tile_index(x, y, w)];
}

// level 0 of a material's three main textures, interleaved (device_types.h MaterialTexels)
__global__ __launch_bounds__(256) void material_texels_kernel(const uint32_t* __restrict__ base, const uint32_t* __restrict__ normal, const uint32_t* __restrict__ mr,
                                                              uint32_t w, uint32_t h, uint2* __restrict__ dst) {
    const uint32_t x = blockIdx.x * 32u + (threadIdx.x & 31u), y = blockIdx.y * 8u + (threadIdx.x >> 5);
    if (x >= w || y >= h) return;
    const size_t i = (size_t)y * w + x;
    const uint32_t n = normal[i], m = mr[i];
    dst[material_texel_index(x, y, w)] = make_uint2(base[i], (n & 0xffffu) | (((m >> 16) & 0xffu) << 16) | (((m >> 8) & 0xffu) << 24));      // n.r, n.g, m.b (metal), m.g (roughness)
}

// ALPHA QUADS of one level of a tiled RGBA8 chain: word (x, y) = the alpha bytes of the bilinear footprint whose top-left texel is (x, y) -- a(x, y), a(x + 1, y),
// a(x, y + 1), a(x + 1, y + 1), with the sampler's wrap -- in the level's own tiled order.  The visibility scan's alpha test reads ONE word per tap and level where the
// colour texture costs four loads (pt_device.h trilinear_alpha_quads_at).
__global__ __launch_bounds__(256) void alpha_quads_kernel(const uint32_t* __restrict__ level, uint32_t w, uint32_t h, uint32_t* __restrict__ dst) {
    const uint32_t x = blockIdx.x * 32u + (threadIdx.x & 31u), y = blockIdx.y * 8u + (threadIdx.x >> 5);
    if (x >= w || y >= h) return;
    const uint32_t x1 = x + 1u == w ? 0u : x + 1u, y1 = y + 1u == h ? 0u : y + 1u;
    const uint32_t a00 = level[texel_tile_index(x, y, w)] >> 24, a10 = level[texel_tile_index(x1, y, w)] >> 24;
    const uint32_t a01 = level[texel_tile_index(x, y1, w)] >> 24, a11 = level[texel_tile_index(x1, y1, w)] >> 24;
    dst[texel_tile_index(x, y, w)] = a00 | (a10 << 8) | (a01 << 16) | (a11 << 24);
}
// the whole tiled chain of the three textures, interleaved word by word (MaterialTexels::chain)
__global__ __launch_bounds__(256) void material_chain_kernel(const uint32_t* __restrict__ base, const uint32_t* __restrict__ normal, const uint32_t* __restrict__ mr,
                                                             uint64_t words, uint2* __restrict__ dst) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= words) return;
    const uint32_t n = normal[i], m = mr[i];
    dst[i] = make_uint2(base[i], (n & 0xffffu) | (((m >> 16) & 0xffu) << 16) | (((m >> 8) & 0xffu) << 24));
}

} // namespace incr

namespace incr {
// ---- block-compressed chains handed over by the integrator (incr_types.h formats 4-7): decoded ONCE, at scene creation, into the words
// the samplers read.  The decode is fixed by the format (D3D11.3 functional spec 19.5: BC1 colour block, BC4 block); what a texture
// unit keeps of the palette's precision is pinned:
//   BC3 -> RGBA8 as the texture caches of current GPUs expand BC1-3: colour endpoints 5:6:5 widened by bit replication, the two
//          interpolants (2 a + b) / 3 and (a + 2 b) / 3 rounded to nearest (+ 1 before the division); alpha = the BC4 palette rounded
//          to 8 bits ((n + 17) / 35).  The result is a plain INCR_TEX_RGBA8_SRGB texture.
//   BC4 / BC5 -> EXACT: every palette entry of an 8-bit-endpoint block is a multiple of 1/35 of an endpoint step ((7 - k) e0 + k e1) / 7
//          or ((5 - k) e0 + k e1) / 5, so the texel stores the integer numerator n over 35 x 255 = 8925 (UNORM) or 35 x 127 = 4445
//          (SNORM; -128 reads as -127) in 16 bits per channel and the sampler scales it back (INCR_TEX_RG16_OVER_8925 / INCR_TEX_RG16S_OVER_4445, pt_device.h decode_texel).
__device__ __forceinline__ int32_t bc4_numerator(const uint8_t* __restrict__ block, uint32_t texel, bool snorm) {      // palette value x 35
    int32_t e0 = block[0], e1 = block[1];
    if (snorm) { e0 = (int8_t)block[0]; e1 = (int8_t)block[1]; e0 = e0 < -127 ? -127 : e0; e1 = e1 < -127 ? -127 : e1; }
    uint64_t bits = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) bits |= (uint64_t)block[2 + k] << (8 * k);
    const int32_t i = (int32_t)((bits >> (3u * texel)) & 7u);
    if (i == 0) return e0 * 35;
    if (i == 1) return e1 * 35;
    if (e0 > e1) return ((8 - i) * e0 + (i - 1) * e1) * 5;             // six interpolants in sevenths
    if (i < 6) return ((6 - i) * e0 + (i - 1) * e1) * 7;               // four interpolants in fifths
    const int32_t one = snorm ? 127 * 35 : 255 * 35;
    return i == 6 ? (snorm ? -one : 0) : one;
}
__global__ __launch_bounds__(256) void bc_decode_kernel(const uint8_t* __restrict__ blocks, uint32_t w, uint32_t h, uint32_t format, uint32_t* __restrict__ dst) {
    const uint32_t x = blockIdx.x * 32u + (threadIdx.x & 31u), y = blockIdx.y * 8u + (threadIdx.x >> 5);
    if (x >= w || y >= h) return;
    const uint32_t bw = (w + 3u) >> 2, texel = (y & 3u) * 4u + (x & 3u);
    const size_t bi = (size_t)(y >> 2) * bw + (x >> 2);
    uint32_t word;
    if (format == INCR_TEX_BC4_UNORM) {
        word = (uint32_t)bc4_numerator(blocks + bi * 8, texel, false);
    } else if (format == INCR_TEX_BC5_UNORM) {
        word = (uint32_t)bc4_numerator(blocks + bi * 16, texel, false) | ((uint32_t)bc4_numerator(blocks + bi * 16 + 8, texel, false) << 16);
    } else if (format == INCR_TEX_BC5_SNORM) {
        word = ((uint32_t)bc4_numerator(blocks + bi * 16, texel, true) & 0xffffu) | ((uint32_t)bc4_numerator(blocks + bi * 16 + 8, texel, true) << 16);
    } else {                                                              // INCR_TEX_BC3_SRGB: alpha block, then colour block
        const uint8_t* __restrict__ b = blocks + bi * 16;
        const uint32_t a = (uint32_t)(bc4_numerator(b, texel, false) + 17) / 35u;
        const uint32_t c0 = b[8] | ((uint32_t)b[9] << 8), c1 = b[10] | ((uint32_t)b[11] << 8);
        const uint32_t idx = ((b[12] | ((uint32_t)b[13] << 8) | ((uint32_t)b[14] << 16) | ((uint32_t)b[15] << 24)) >> (2u * texel)) & 3u;
        auto widen = [](uint32_t c, uint32_t* r, uint32_t* g, uint32_t* bl) {
            const uint32_t r5 = (c >> 11) & 31u, g6 = (c >> 5) & 63u, b5 = c & 31u;
            *r = (r5 << 3) | (r5 >> 2); *g = (g6 << 2) | (g6 >> 4); *bl = (b5 << 3) | (b5 >> 2);
        };
        uint32_t r0, g0, b0, r1, g1, b1;
        widen(c0, &r0, &g0, &b0); widen(c1, &r1, &g1, &b1);
        uint32_t r, g, bl;
        if (idx == 0u) { r = r0; g = g0; bl = b0; }
        else if (idx == 1u) { r = r1; g = g1; bl = b1; }
        else if (idx == 2u) { r = (2u * r0 + r1 + 1u) / 3u; g = (2u * g0 + g1 + 1u) / 3u; bl = (2u * b0 + b1 + 1u) / 3u; }
        else { r = (r0 + 2u * r1 + 1u) / 3u; g = (g0 + 2u * g1 + 1u) / 3u; bl = (b0 + 2u * b1 + 1u) / 3u; }
        word = r | (g << 8) | (bl << 16) | (a << 24);
    }
    dst[(size_t)y * w + x] = word;
}

} // namespace incr

extern "C" hipError_t incr_launch_mip_untile(const uint32_t* src, uint32_t w, uint32_t h, uint32_t* dst, hipStream_t stream) {
    hipLaunchKernelGGL(incr::mip_untile_kernel, dim3((w + 31u) / 32u, (h + 7u) / 8u), dim3(256), 0, stream, src, w, h, dst);
    return hipGetLastError();
}
extern "C" hipError_t incr_launch_material_texels(const uint32_t* base, const uint32_t* normal, const uint32_t* mr, uint32_t w, uint32_t h, void* dst, hipStream_t stream) {
    hipLaunchKernelGGL(incr::material_texels_kernel, dim3((w + 31u) / 32u, (h + 7u) / 8u), dim3(256), 0, stream, base, normal, mr, w, h, (uint2*)dst);
    return hipGetLastError();
}
extern "C" hipError_t incr_launch_alpha_quads(const uint32_t* level, uint32_t w, uint32_t h, uint32_t* dst, hipStream_t stream) {
    hipLaunchKernelGGL(incr::alpha_quads_kernel, dim3((w + 31u) / 32u, (h + 7u) / 8u), dim3(256), 0, stream, level, w, h, dst);
    return hipGetLastError();
}
extern "C" hipError_t incr_launch_material_chain(const uint32_t* base, const uint32_t* normal, const uint32_t* mr, uint64_t words, void* dst, hipStream_t stream) {
    if (!words) return hipSuccess;
    hipLaunchKernelGGL(incr::material_chain_kernel, dim3((uint32_t)((words + 255u) / 256u)), dim3(256), 0, stream, base, normal, mr, words, (uint2*)dst);
    return hipGetLastError();
}
extern "C" hipError_t incr_launch_mip_tile(const uint32_t* src, uint32_t w, uint32_t h, uint32_t* dst, hipStream_t stream) {
    hipLaunchKernelGGL(incr::mip_tile_kernel, dim3((w + 31u) / 32u, (h + 7u) / 8u), dim3(256), 0, stream, src, w, h, dst);
    return hipGetLastError();
}
extern "C" hipError_t incr_launch_mip_box(const uint32_t* src, uint32_t sw, uint32_t sh, uint32_t* dst, uint32_t dw, uint32_t dh, uint32_t format,
                                          const float* srgbLut, hipStream_t stream) {
    hipLaunchKernelGGL(incr::mip_box_kernel, dim3((dw + 31u) / 32u, (dh + 7u) / 8u), dim3(256), 0, stream, src, sw, sh, dst, dw, dh, format, srgbLut);
    return hipGetLastError();
}
extern "C" hipError_t incr_launch_mip_adopt_vmf(const uint32_t* vmfN, const uint32_t* vmfMR, uint32_t* dstN, uint32_t* dstMR, uint64_t first, uint64_t count, hipStream_t stream) {
    if (!count) return hipSuccess;
    hipLaunchKernelGGL(incr::mip_adopt_vmf_kernel, dim3((uint32_t)((count + 255u) / 256u)), dim3(256), 0, stream, vmfN, vmfMR, dstN, dstMR, first, count);
    return hipGetLastError();
}
extern "C" hipError_t incr_launch_bc_decode(const uint8_t* blocks, uint32_t w, uint32_t h, uint32_t format, uint32_t* dst, hipStream_t stream) {
    hipLaunchKernelGGL(incr::bc_decode_kernel, dim3((w + 31u) / 32u, (h + 7u) / 8u), dim3(256), 0, stream, blocks, w, h, format, dst);
    return hipGetLastError();
}
