// Issue cost of candidate instructions for the BVH slab test on gfx950, in cycles per wave64 instruction: one workgroup of W waves per
// SIMD-set runs N back-to-back independent instances of the instruction between two s_memtime stamps.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/issue_rate tools/microbench/issue_rate.hip && gpurun_out/issue_rate
// Round 5: 1 / 2 / 4 waves per SIMD as one workgroup on one CU (as before), and 4 / 6 / 7 / 8 waves per SIMD with EVERY CU loaded: W workgroups of 256 threads per CU,
// pinned to exactly W per CU by their LDS allocation (160 KB / W each) -- the occupancy the traversal kernels run at (7) -- plus the traversal loop's own mix
// (cvt_ubyte + fma + min / max + cndmask, with and without a global_load_dwordx4 every 48 instructions).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
template <int OP> __global__ void k(uint64_t* out, float* sink, const float4* gsrc) {
    extern __shared__ float s_pin[];          // only pins the number of workgroups per CU
    if (gsrc == (const float4*)1) s_pin[threadIdx.x] = 0.f;
    float a = threadIdx.x * 0.5f, b = 1.0001f, c = 0.3f, d = 0.7f, e = 0.1f;
    uint32_t u = threadIdx.x * 2654435761u, v = 0x01020304u, w = 7u, x = 9u;
    const uint64_t cond = 0x5555aaaa5555aaaaull ^ (uint64_t)(uintptr_t)out;
    __syncthreads();
    __builtin_amdgcn_s_waitcnt(0);
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 64; ++i) {
        if (OP == 0) { REP64(asm volatile("v_fma_f32 %0, %1, %2, %0\n v_fma_f32 %3, %1, %2, %3\n v_fma_f32 %4, %1, %2, %4\n v_fma_f32 %5, %1, %2, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(d));) }
        if (OP == 1) { REP64(asm volatile("v_cvt_f32_ubyte0 %0, %4\n v_cvt_f32_ubyte1 %1, %4\n v_cvt_f32_ubyte2 %2, %4\n v_cvt_f32_ubyte3 %3, %4" : "=v"(a), "=v"(c), "=v"(d), "=v"(e) : "v"(v));) }
        if (OP == 2) { REP64(asm volatile("v_dot2_f32_f16 %0, %4, %5, %0\n v_dot2_f32_f16 %1, %4, %5, %1\n v_dot2_f32_f16 %2, %4, %5, %2\n v_dot2_f32_f16 %3, %4, %5, %3" : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(v), "v"(w));) }
        if (OP == 3) { REP64(asm volatile("v_dot2_i32_i16 %0, %4, %5, %0\n v_dot2_i32_i16 %1, %4, %5, %1\n v_dot2_i32_i16 %2, %4, %5, %2\n v_dot2_i32_i16 %3, %4, %5, %3" : "+v"(u), "+v"(x), "+v"(w), "+v"(v) : "v"(v), "v"(w));) }
        if (OP == 4) { REP64(asm volatile("v_pk_mad_i16 %0, %4, %5, %0\n v_pk_mad_i16 %1, %4, %5, %1\n v_pk_mad_i16 %2, %4, %5, %2\n v_pk_mad_i16 %3, %4, %5, %3" : "+v"(u), "+v"(x), "+v"(w), "+v"(v) : "v"(v), "v"(w));) }
        if (OP == 5) { REP64(asm volatile("v_perm_b32 %0, %4, %5, %6\n v_perm_b32 %1, %4, %5, %6\n v_perm_b32 %2, %4, %5, %6\n v_perm_b32 %3, %4, %5, %6" : "=v"(u), "=v"(x), "=v"(w), "=v"(v) : "v"(v), "v"(w), "v"(x));) }
        if (OP == 6) { REP64(asm volatile("v_fma_mix_f32 %0, %4, %5, %0 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %4, %5, %1 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %2, %4, %5, %2 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %4, %5, %3 op_sel_hi:[1,0,0]" : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(v), "v"(b));) }
        if (OP == 7) { REP64(asm volatile("v_max3_f32 %0, %0, %4, %5\n v_max3_f32 %1, %1, %4, %5\n v_max3_f32 %2, %2, %4, %5\n v_max3_f32 %3, %3, %4, %5" : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b), "v"(b));) }
        if (OP == 8) { REP64(asm volatile("v_pk_fma_f32 %0, %2, %3, %0\n v_pk_fma_f32 %1, %2, %3, %1" : "+v"(*(double*)&u), "+v"(*(double*)&w) : "v"(*(double*)&a), "v"(*(double*)&c));) }
        if (OP == 9) { REP64(asm volatile("v_pk_max_i16 %0, %0, %4\n v_pk_min_i16 %1, %1, %4\n v_pk_max_i16 %2, %2, %4\n v_pk_min_i16 %3, %3, %4" : "+v"(u), "+v"(x), "+v"(w), "+v"(v) : "v"(v));) }
        if (OP == 10) { REP64(asm volatile("v_cndmask_b32 %0, %4, %5, %6\n v_cndmask_b32 %1, %4, %5, %6\n v_cndmask_b32 %2, %4, %5, %6\n v_cndmask_b32 %3, %4, %5, %6" : "=v"(u), "=v"(x), "=v"(w), "=v"(v) : "v"(a), "v"(b), "s"(cond));) }
        if (OP == 11) { REP64(asm volatile("v_cmp_lt_f32 vcc, %4, %5\n v_cndmask_b32 %0, %4, %5, vcc\n v_cmp_gt_f32 vcc, %4, %5\n v_cndmask_b32 %1, %4, %5, vcc" : "=v"(u), "=v"(x), "=v"(w), "=v"(v) : "v"(a), "v"(b) : "vcc");) }
        if (OP == 12) { REP64(asm volatile("v_cmp_lt_f32 %6, %4, %5\n v_cmp_gt_f32 %7, %4, %5\n v_cmp_le_f32 %6, %4, %5\n v_cmp_ge_f32 %7, %4, %5" : "=v"(u), "=v"(x), "=v"(w), "=v"(v) : "v"(a), "v"(b), "s"(cond), "s"(cond));) }
        if (OP == 13) { REP64(asm volatile("v_med3_f32 %0, %0, %4, %5\n v_min3_f32 %1, %1, %4, %5\n v_max_f32 %2, %2, %4\n v_min_f32 %3, %3, %4" : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b), "v"(b));) }
        if (OP == 14) { REP64(asm volatile("v_lshlrev_b32 %0, 3, %4\n v_and_b32 %1, %4, %5\n v_bfe_u32 %2, %4, 8, 8\n v_add_u32 %3, %4, %5" : "=v"(u), "=v"(x), "=v"(w), "=v"(v) : "v"(a), "v"(b));) }
        // round 4: the integer instructions of the traversal loop's bookkeeping (which of them are quarter-rate?)
        if (OP == 15) { REP64(asm volatile("v_mul_lo_u32 %0, %4, %5\n v_mul_lo_u32 %1, %4, %5\n v_mul_lo_u32 %2, %4, %5\n v_mul_lo_u32 %3, %4, %5" : "=v"(u), "=v"(x), "=v"(w), "=v"(v) : "v"(a), "v"(b));) }
        if (OP == 16) { REP64(asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1" : "+v"(*(double*)&u), "+v"(*(double*)&w) : "v"(a), "v"(b) : "vcc");) }
        if (OP == 17) { REP64(asm volatile("v_ffbl_b32 %0, %4\n v_ffbh_u32 %1, %4\n v_ffbl_b32 %2, %5\n v_ffbh_u32 %3, %5" : "=v"(u), "=v"(x), "=v"(w), "=v"(v) : "v"(a), "v"(b));) }
        if (OP == 18) { REP64(asm volatile("v_lshl_add_u32 %0, %4, 2, %5\n v_lshl_or_b32 %1, %4, 2, %5\n v_and_or_b32 %2, %4, %5, %4\n v_med3_u32 %3, %4, %5, %4" : "=v"(u), "=v"(x), "=v"(w), "=v"(v) : "v"(a), "v"(b));) }
        if (OP == 19) { REP64(asm volatile("v_rcp_f32 %0, %4\n v_rcp_f32 %1, %5\n v_sqrt_f32 %2, %4\n v_rsq_f32 %3, %5" : "=v"(a), "=v"(c), "=v"(d), "=v"(e) : "v"(b), "v"(b));) }
        if (OP == 20) { REP64(asm volatile("v_lshlrev_b64 %0, 5, %2\n v_lshlrev_b64 %1, 4, %2" : "=v"(*(double*)&u), "=v"(*(double*)&w) : "v"(*(double*)&a));) }
        if (OP == 21) { REP64(asm volatile("v_bfe_i32 %0, %4, 3, 1\n v_bcnt_u32_b32 %1, %4, %5\n v_mbcnt_lo_u32_b32 %2, %4, %5\n v_cvt_f32_u32 %3, %4" : "=v"(u), "=v"(x), "=v"(w), "=v"(v) : "v"(a), "v"(b));) }
        if (OP == 22) { REP64(asm volatile("v_mul_hi_u32 %0, %4, %5\n v_mul_u32_u24 %1, %4, %5\n v_mad_u32_u24 %2, %4, %5, %4\n v_mul_hi_u32 %3, %4, %5" : "=v"(u), "=v"(x), "=v"(w), "=v"(v) : "v"(a), "v"(b));) }
        if (OP == 23) { REP64(asm volatile("v_mul_u32_u24 %0, %4, %5\n v_mad_u32_u24 %1, %4, %5, %4\n v_mul_u32_u24 %2, %4, %5\n v_mad_u32_u24 %3, %4, %5, %4" : "=v"(u), "=v"(x), "=v"(w), "=v"(v) : "v"(a), "v"(b));) }
        if (OP == 24) { REP64(asm volatile("v_cvt_f32_ubyte0 %0, %4\n v_fma_f32 %1, %0, %5, %1\n v_cvt_f32_ubyte1 %2, %4\n v_fma_f32 %3, %2, %5, %3\n v_max_f32 %1, %1, %3\n v_min_f32 %3, %3, %0\n v_cndmask_b32 %0, %1, %3, %6\n v_fma_f32 %2, %0, %5, %2" : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(v), "v"(b), "s"(cond));) }
        if (OP == 25) { float4 ld; const float4* gp = gsrc + (threadIdx.x * 37u + (uint32_t)i * 64u) % 4096u;
            REP16(asm volatile("v_cvt_f32_ubyte0 %0, %5\n v_fma_f32 %1, %0, %6, %1\n v_cvt_f32_ubyte1 %2, %5\n v_fma_f32 %3, %2, %6, %3\n v_max_f32 %1, %1, %3\n v_min_f32 %3, %3, %0\n v_cndmask_b32 %0, %1, %3, %7\n v_fma_f32 %2, %0, %6, %2\n"
                               "v_cvt_f32_ubyte0 %0, %5\n v_fma_f32 %1, %0, %6, %1\n v_cvt_f32_ubyte1 %2, %5\n v_fma_f32 %3, %2, %6, %3\n v_max_f32 %1, %1, %3\n v_min_f32 %3, %3, %0\n v_cndmask_b32 %0, %1, %3, %7\n v_fma_f32 %2, %0, %6, %2\n"
                               "v_cvt_f32_ubyte0 %0, %5\n v_fma_f32 %1, %0, %6, %1\n v_cvt_f32_ubyte1 %2, %5\n v_fma_f32 %3, %2, %6, %3\n v_max_f32 %1, %1, %3\n v_min_f32 %3, %3, %0\n v_cndmask_b32 %0, %1, %3, %7\n v_fma_f32 %2, %0, %6, %2\n"
                               "v_cvt_f32_ubyte0 %0, %5\n v_fma_f32 %1, %0, %6, %1\n v_cvt_f32_ubyte1 %2, %5\n v_fma_f32 %3, %2, %6, %3\n v_max_f32 %1, %1, %3\n v_min_f32 %3, %3, %0\n v_cndmask_b32 %0, %1, %3, %7\n v_fma_f32 %2, %0, %6, %2\n"
                               "v_cvt_f32_ubyte0 %0, %5\n v_fma_f32 %1, %0, %6, %1\n v_cvt_f32_ubyte1 %2, %5\n v_fma_f32 %3, %2, %6, %3\n v_max_f32 %1, %1, %3\n v_min_f32 %3, %3, %0\n v_cndmask_b32 %0, %1, %3, %7\n v_fma_f32 %2, %0, %6, %2\n"
                               "v_cvt_f32_ubyte0 %0, %5\n v_fma_f32 %1, %0, %6, %1\n v_cvt_f32_ubyte1 %2, %5\n v_fma_f32 %3, %2, %6, %3\n v_max_f32 %1, %1, %3\n v_min_f32 %3, %3, %0\n v_cndmask_b32 %0, %1, %3, %7\n v_fma_f32 %2, %0, %6, %2\n"
                               "global_load_dwordx4 %4, %8, off\n s_waitcnt vmcnt(3)" : "+v"(a), "+v"(c), "+v"(d), "+v"(e), "=v"(ld) : "v"(v), "v"(b), "s"(cond), "v"(gp));)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
    sink[threadIdx.x + blockIdx.x * blockDim.x] = a + c + d + e + (float)(u + x + w + v);
}
template <int OP> static void run(const char* name, double perIter) {
    uint64_t* out; float* sink;
    hipMalloc(&out, 8 * 65536); hipMalloc(&sink, 4 * 1024 * 1024 * 2);
    float4* gsrc; hipMalloc(&gsrc, 16 * 8192); hipMemset(gsrc, 0, 16 * 8192);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    hipFuncSetAttribute((const void*)k<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int mode = 0; mode < 7; ++mode) {
        const int wavesPerSimd = mode < 3 ? (1 << mode) : (mode == 3 ? 4 : mode == 4 ? 6 : mode == 5 ? 7 : 8);
        const bool chip = mode >= 3;                      // every CU loaded with wavesPerSimd workgroups of 256 threads, pinned by LDS
        const int threads = chip ? 256 : 256 * wavesPerSimd;          // else one workgroup on one CU: wavesPerSimd waves on each of its 4 SIMDs
        const int blocks = chip ? prop.multiProcessorCount * wavesPerSimd : 1;
        const size_t lds = chip ? (size_t)(160 * 1024 / wavesPerSimd) / 1024 * 1024 - (wavesPerSimd == 1 ? 0 : 0) : 0;
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), lds, 0, out, sink, gsrc);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), lds, 0, out, sink, gsrc);
        hipDeviceSynchronize();
        static uint64_t h[65536]; const int nw = blocks * threads / 64; hipMemcpy(h, out, 8 * (size_t)nw, hipMemcpyDeviceToHost);
        double sum = 0; for (int i = 0; i < nw; ++i) sum += (double)h[i];
        const double instr = 64.0 * 64 * perIter;
        printf("%-22s %d wave(s)/SIMD%s: %.2f cycles per instruction per wave  (=> %.2f per SIMD slot)\n", name, wavesPerSimd, chip ? ", every CU loaded" : "", sum / nw / instr, sum / nw / instr / wavesPerSimd);
    }
    hipFree(gsrc);
    hipFree(out); hipFree(sink);
}
int main() {
    run<0>("v_fma_f32", 4); run<1>("v_cvt_f32_ubyteN", 4); run<2>("v_dot2_f32_f16", 4); run<3>("v_dot2_i32_i16", 4); run<4>("v_pk_mad_i16", 4);
    run<5>("v_perm_b32", 4); run<6>("v_fma_mix_f32", 4); run<7>("v_max3_f32", 4); run<8>("v_pk_fma_f32", 2); run<9>("v_pk_max/min_i16", 4); run<10>("v_cndmask_b32 (sgpr)", 4); run<11>("v_cmp+v_cndmask vcc", 4); run<12>("v_cmp_f32 -> sgpr", 4); run<13>("med3/min3/max/min", 4); run<14>("shift/and/bfe/add", 4);
    run<15>("v_mul_lo_u32", 4); run<16>("v_mad_u64_u32", 2); run<17>("v_ffbl/v_ffbh", 4); run<18>("lshl_add/lshl_or/and_or/med3_u32", 4); run<19>("rcp/rcp/sqrt/rsq", 4); run<20>("v_lshlrev_b64", 2);
    run<21>("bfe_i32/bcnt/mbcnt/cvt_f32_u32", 4); run<22>("mul_hi/mul_u24/mad_u24/mul_hi", 4); run<23>("mul_u32_u24/mad_u32_u24", 4);
    run<24>("traversal mix (cvt fma cvt fma max min cndmask fma)", 8); run<25>("the mix + a global_load_dwordx4 every 48", 49.0 * 16.0 / 64.0);
    return 0;
}
