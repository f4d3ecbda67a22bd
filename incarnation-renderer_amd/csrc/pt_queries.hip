// pt_queries.hip -- the small kernels around the path tracer: incr_trace_rays (batch of inline ray queries, the
// RayQuery / TraceRay stand-in exposed for tests and tools), TonemapPass and the shard un-interleave.
// The integrator itself is pathtrace_wf.hip.  (Round 1 also kept a one-launch megakernel and a persistent-wave variant
// of the integrator here; both were 2x slower than the wavefront kernels and were removed in round 2.)
#include "pt_traverse.h"
#include "ibl_device.h"
#include <algorithm>

namespace incr {

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

// Self-test of the hardware small-float codec (pt_device.h INCR_FAST_SMALL_FLOAT) against the generic bit-level functions, over EVERY
// float32 pattern (encode) and every storage pattern (decode).  out[0..3] = mismatches of f32 -> f16, f32 -> R11 / R10 channels,
// f16 -> f32, R11G11B10 -> f32 x 3 (two NaNs count as equal whatever their payload).
__global__ __launch_bounds__(256) void codec_selftest_kernel(unsigned long long* __restrict__ out) {
    unsigned long long bad[4] = {0, 0, 0, 0};
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t u = (uint32_t)i;
        const float f = __uint_as_float(u);
        if (f32_to_f16_bits(f, 0) != f32_to_f16_bits_generic(f, 0)) bad[0]++;
        if (pack_r11g11b10({f, f, f}, 0) != pack_r11g11b10_generic({f, f, f}, 0)) bad[1]++;
        if (u < 0x10000u) {
            const float a = f16_bits_to_f32(u), b = decode_small_float(u & 0x7fffu, 10, (u >> 15) & 1u);
            if (!(__float_as_uint(a) == __float_as_uint(b) || (a != a && b != b))) bad[2]++;
        }
        const f3 a = unpack_r11g11b10(u), b = unpack_r11g11b10_generic(u);
        const bool same = (__float_as_uint(a.x) == __float_as_uint(b.x) || (a.x != a.x && b.x != b.x)) && (__float_as_uint(a.y) == __float_as_uint(b.y) || (a.y != a.y && b.y != b.y)) &&
                          (__float_as_uint(a.z) == __float_as_uint(b.z) || (a.z != a.z && b.z != b.z));
        if (!same) bad[3]++;
    }
    for (int k = 0; k < 4; ++k) if (bad[k]) atomicAdd(out + k, bad[k]);
}

// Self-test of rcp_cr (pt_device.h: the reciprocal behind every pinned shader division a * rcp(b)) against the IEEE quotient 1.f / x over EVERY
// float32 pattern.  out[0] = patterns with x and 1 / x both NORMAL numbers (2^-126 <= |x| <= 2^126) where the two differ, out[1] = of those, the
// ones further than one ulp apart, out[2] = differing patterns outside that domain (denormal x or denormal quotient: v_rcp_f32 flushes them;
// no shader denominator lives there), two NaNs counting as equal.
__global__ __launch_bounds__(256) void rcp_selftest_kernel(unsigned long long* __restrict__ out) {
    unsigned long long bad = 0, far = 0, outside = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += (uint64_t)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((uint32_t)i);
        const float a = rcp_cr(x), b = 1.f / x;
        const uint32_t ua = __float_as_uint(a), ub = __float_as_uint(b);
        if (ua == ub || (a != a && b != b)) continue;
        const uint32_t e = ((uint32_t)i >> 23) & 0xffu;
        const bool domain = e >= 1u && (e < 253u || (e == 253u && ((uint32_t)i & 0x7fffffu) == 0u));      // |x| in [2^-126, 2^126]
        if (!domain) { outside++; continue; }
        bad++;
        const uint32_t d = ua > ub ? ua - ub : ub - ua;
        if (d > 1u) far++;
    }
    if (bad) atomicAdd(out, bad);
    if (far) atomicAdd(out + 1, far);
    if (outside) atomicAdd(out + 2, outside);
}

// TonemapPass: shaders/postprocess/tonemap.hlsl:30-59 with the R8G8B8A8_UNORM_SRGB render-target encode (tonemap_texel, pt_device.h).
// Streaming: 16 B in, 16 B out per lane (four texels) -- with one texel per lane the pass ran at 0.59 TB/s.
__global__ __launch_bounds__(256) void tonemap_kernel(const uint32_t* __restrict__ hdr, uint32_t* __restrict__ ldr, size_t n, int ev100, int viewmode, const float* __restrict__ srgbSteps) {
    __shared__ float s_steps[kSrgbSteps];
    for (uint32_t i = threadIdx.x; i < kSrgbSteps; i += blockDim.x) s_steps[i] = srgbSteps[i];
    __syncthreads();
    const float e = Exposure(ev100);
    const size_t quads = n / 4;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += (size_t)gridDim.x * blockDim.x) {
        const uint4 h = ((const uint4*)hdr)[q];
        ((uint4*)ldr)[q] = make_uint4(tonemap_texel(s_steps, h.x, e, viewmode), tonemap_texel(s_steps, h.y, e, viewmode), tonemap_texel(s_steps, h.z, e, viewmode), tonemap_texel(s_steps, h.w, e, viewmode));
    }
    for (size_t i = quads * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) ldr[i] = tonemap_texel(s_steps, hdr[i], e, viewmode);
}

__global__ __launch_bounds__(256) void tonemap_scalar_kernel(const uint32_t* __restrict__ hdr, uint32_t* __restrict__ ldr, size_t n, int ev100, int viewmode, const float* __restrict__ srgbSteps) {
    __shared__ float s_steps[kSrgbSteps];
    for (uint32_t i = threadIdx.x; i < kSrgbSteps; i += blockDim.x) s_steps[i] = srgbSteps[i];
    __syncthreads();
    const float e = Exposure(ev100);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) ldr[i] = tonemap_texel(s_steps, hdr[i], e, viewmode);
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
extern "C" hipError_t incr_launch_trace_rays(const incr::PtParams* p, uint32_t n, const float* origins, const float* dirs, float tmin, float tmax,
                                             int anyHit, uint32_t* ids, float* tuv, hipStream_t stream) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(incr::trace_rays_kernel, dim3((n + incr::kBlock - 1) / incr::kBlock), dim3(incr::kBlock), 0, stream, *p, n, origins, dirs, tmin, tmax, anyHit, ids, tuv);
    return hipGetLastError();
}
extern "C" hipError_t incr_launch_tonemap(const uint32_t* hdr, uint32_t* ldr, size_t n, int ev100, int viewmode, const float* srgbSteps, hipStream_t stream) {
    if (!n) return hipSuccess;
    // surfaces come from hipMalloc (256-B aligned) or are row offsets of one: rows*res_x*4 B keeps uint4 alignment when res_x % 4 == 0;
    // anything else takes the scalar tail loop only
    const bool wide = (((uintptr_t)hdr | (uintptr_t)ldr) & 15u) == 0;
    if (!wide) { const unsigned b1 = (unsigned)std::min<size_t>((n + 255) / 256, 8192); hipLaunchKernelGGL(incr::tonemap_scalar_kernel, dim3(b1), dim3(256), 0, stream, hdr, ldr, n, ev100, viewmode, srgbSteps); return hipGetLastError(); }
    const unsigned blocks = (unsigned)std::min<size_t>((n / 4 + 255) / 256 + 1, 8192);
    hipLaunchKernelGGL(incr::tonemap_kernel, dim3(blocks), dim3(256), 0, stream, hdr, ldr, n, ev100, viewmode, srgbSteps);
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

extern "C" hipError_t incr_launch_rcp_selftest(unsigned long long* out2, hipStream_t stream) {
    hipLaunchKernelGGL(incr::rcp_selftest_kernel, dim3(256 * 64), dim3(256), 0, stream, out2);
    return hipGetLastError();
}
extern "C" hipError_t incr_launch_codec_selftest(unsigned long long* out4, hipStream_t stream) {
    hipLaunchKernelGGL(incr::codec_selftest_kernel, dim3(256 * 64), dim3(256), 0, stream, out4);
    return hipGetLastError();
}
