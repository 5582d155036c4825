// Round-6 micro-probes (standalone: hipcc --offload-arch=gfx950 -O3 r06_probes.hip -o r06_probes).
//   1. VALU issue cost per wave-instruction for the instruction classes of the attention soft-max
//      (v_fma_f32, v_exp_f32, v_cvt_pk_bf16_f32, v_bfe_i32 + v_and, v_mul_lo_u32, v_mul_u32_u24, v_cndmask) at 1 / 2 / 4
//      waves per SIMD: decides whether "N VALU per score" is priced at 2 or 4 cycles per wave-instruction.
//   2. fp32 global atomic add throughput in the access pattern a key-stationary attention backward would use for dQ
//      (16 lanes x 4 B contiguous per row, 4 rows per wave-instruction; `nshare` workgroups adding into the same tile):
//      decides whether a one-kernel backward with atomically accumulated dQ can be considered at all.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int KIND>
__global__ void valu_kernel(float* out, unsigned long long* cyc, int iters) {
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
  unsigned u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, u4 = u0 + 4, u5 = u0 + 5, u6 = u0 + 6, u7 = u0 + 7;
  const float c = 1.0001f, e = 0.999f;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
    if constexpr (KIND == 0) {        // v_fma_f32, 8 independent chains x 8
#define S(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a##k) : "v"(c), "v"(e));
      REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S)
#undef S
    } else if constexpr (KIND == 1) { // v_exp_f32
#define S(k) asm volatile("v_exp_f32 %0, %0" : "+v"(a##k));
      REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S)
#undef S
    } else if constexpr (KIND == 2) { // v_cvt_pk_bf16_f32
#define S(k) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a##k) : "v"(c));
      REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S)
#undef S
    } else if constexpr (KIND == 3) { // v_bfe_i32 + v_and_b32 pair (counted as 2 instructions)
#define S(k) asm volatile("v_bfe_i32 %1, %1, 3, 1\n\tv_and_b32 %0, %0, %1" : "+v"(a##k), "+v"(u##k));
      REP8(S) REP8(S) REP8(S) REP8(S)
#undef S
    } else if constexpr (KIND == 4) { // v_mul_lo_u32
#define S(k) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u##k) : "v"(u0));
      REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S)
#undef S
    } else if constexpr (KIND == 5) { // v_mul_u32_u24
#define S(k) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(u##k) : "v"(u0));
      REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S)
#undef S
    } else if constexpr (KIND == 6) { // v_cmp + v_cndmask pair (2 instructions)
#define S(k) asm volatile("v_cmp_ge_u32 vcc, %1, %2\n\tv_cndmask_b32 %0, 0, %0, vcc" : "+v"(a##k) : "v"(u##k), "v"(u0) : "vcc");
      REP8(S) REP8(S) REP8(S) REP8(S)
#undef S
    } else if constexpr (KIND == 7) { // v_xor / v_lshlrev pair (2 instructions): the xorshift step
#define S(k) asm volatile("v_lshlrev_b32 %1, 13, %0\n\tv_xor_b32 %0, %0, %1" : "+v"(u##k), "+v"(u7));
      REP8(S) REP8(S) REP8(S) REP8(S)
#undef S
    } else if constexpr (KIND == 8) { // v_mul_f32
#define S(k) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a##k) : "v"(c));
      REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S)
#undef S
    } else if constexpr (KIND == 9) { // dependent v_fma chain (latency)
#define S(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(c), "v"(e));
      REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S)
#undef S
    } else if constexpr (KIND == 10) { // v_permlane32_swap
#define S(k) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(u##k), "+v"(u7));
      REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S)
#undef S
    } else if constexpr (KIND == 11) { // v_pk_mul_f32 (2 floats per lane)
      typedef float f2 __attribute__((ext_vector_type(2)));
      f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, cc = {c, e};
#define S(k) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p##k) : "v"(cc));
      S(0) S(1) S(2) S(3) S(0) S(1) S(2) S(3) S(0) S(1) S(2) S(3) S(0) S(1) S(2) S(3)
      S(0) S(1) S(2) S(3) S(0) S(1) S(2) S(3) S(0) S(1) S(2) S(3) S(0) S(1) S(2) S(3)
      S(0) S(1) S(2) S(3) S(0) S(1) S(2) S(3) S(0) S(1) S(2) S(3) S(0) S(1) S(2) S(3)
      S(0) S(1) S(2) S(3) S(0) S(1) S(2) S(3) S(0) S(1) S(2) S(3) S(0) S(1) S(2) S(3)
#undef S
      a0 = p0[0] + p1[0] + p2[0] + p3[0];
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(u0 ^ u1 ^ u2 ^ u3 ^ u4 ^ u5 ^ u6 ^ u7);
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run_valu(const char* name, int per_iter) {
  float* out; unsigned long long* cyc;
  CK(hipMalloc(&out, 256 * 1024 * sizeof(float)));
  CK(hipMalloc(&cyc, 256 * sizeof(unsigned long long)));
  const int iters = 2000;
  printf("%-34s", name);
  for (int waves_per_simd : {1, 2, 4}) {
    const int threads = 256 * waves_per_simd;     // one workgroup per CU (grid 256), waves spread over the 4 SIMDs
    hipLaunchKernelGGL(valu_kernel<KIND>, dim3(256), dim3(threads), 0, 0, out, cyc, 10);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(valu_kernel<KIND>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(256);
    CK(hipMemcpy(h.data(), cyc, 256 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double mean = 0; for (auto v : h) mean += (double)v; mean /= 256.0;
    // s_memtime / readcyclecounter ticks at a constant 100 MHz on gfx9 (not shader clocks): report wall ns per wave-instruction per SIMD too
    const double inst_per_simd = (double)iters * per_iter * waves_per_simd;
    printf("  %dw/SIMD: %.2f ns/inst/SIMD (ticks %.0f)", waves_per_simd, ms * 1e6 / inst_per_simd, mean);
  }
  printf("\n");
  CK(hipFree(out)); CK(hipFree(cyc));
}

// ---- atomics: grid = ntile * nshare workgroups of 256 threads; workgroup (t, s) adds `reps` times a 64 x 64 fp32 tile into
// tile t of dst (row pitch 64 floats): wave w rows 16 w .. 16 w + 15; per instruction: lane = 16 g + c -> row 4 i + g, columns
// 16 j + c  (16 lanes x 4 B contiguous = one 64-B segment per row, 4 rows per instruction)
template <int MODE>   // 0 fp32 atomic add, 1 plain store (reference), 2 packed bf16 atomic add
__global__ void atomic_kernel(float* dst, int nshare, int reps) {
  const int t = blockIdx.x / nshare;
  float* tile = dst + (size_t)t * 64 * 64;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int g = lane >> 4, c = lane & 15;
  for (int rp = 0; rp < reps; ++rp) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float* p = tile + (size_t)(16 * w + 4 * i + g) * 64 + 16 * j + c;
        if (MODE == 0) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (MODE == 1) __builtin_nontemporal_store(1.0f + rp, p);
      }
  }
}

template <int MODE>
void run_atomic(const char* name, int ntile, int nshare) {
  float* dst;
  const size_t bytes = (size_t)ntile * 64 * 64 * 4;
  CK(hipMalloc(&dst, bytes));
  CK(hipMemset(dst, 0, bytes));
  const int reps = 8;
  hipLaunchKernelGGL(atomic_kernel<MODE>, dim3(ntile * nshare), dim3(256), 0, 0, dst, nshare, 1);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(atomic_kernel<MODE>, dim3(ntile * nshare), dim3(256), 0, 0, dst, nshare, reps);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double nops = (double)ntile * nshare * reps * 4096.0;
  printf("%-26s tiles %6d (%.0f MB) x share %2d: %.3f ms  %.1f G elem/s  %.2f TB/s of fp32\n", name, ntile, bytes / 1e6, nshare, ms,
         nops / ms / 1e6, nops * 4 / ms / 1e9);
  CK(hipFree(dst));
}

int main() {
  printf("== VALU issue cost (ns per wave-instruction per SIMD; at 2.4 GHz 1 cycle = 0.417 ns)\n");
  run_valu<0>("v_fma_f32 (8 chains)", 64);
  run_valu<8>("v_mul_f32", 64);
  run_valu<9>("v_fma_f32 dependent chain", 64);
  run_valu<1>("v_exp_f32", 64);
  run_valu<2>("v_cvt_pk_bf16_f32", 64);
  run_valu<3>("v_bfe_i32 + v_and (2 inst)", 64);
  run_valu<4>("v_mul_lo_u32", 64);
  run_valu<5>("v_mul_u32_u24", 64);
  run_valu<6>("v_cmp + v_cndmask (2 inst)", 64);
  run_valu<7>("v_lshlrev + v_xor (2 inst)", 64);
  run_valu<10>("v_permlane32_swap", 64);
  run_valu<11>("v_pk_mul_f32", 64);
  printf("== fp32 atomic add, 64-B segments (dQ accumulation pattern)\n");
  for (int nshare : {1, 4, 13}) {
    run_atomic<0>("atomic add f32", 1024 * 13 / 4, nshare);     // 53 MB region: beyond L2, inside MALL
    run_atomic<0>("atomic add f32", 1024 * 13, nshare);         // 218 MB region (B = 128, H = 8, T = 800 of dQ)
  }
  run_atomic<1>("plain nt store", 1024 * 13, 1);
  run_atomic<1>("plain nt store", 1024 * 13, 13);
  return 0;
}
