// common.h -- shared device helpers for libnsp_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>
#include "../../include/nsp_hip.h"

// "these values exist in registers HERE": an empty asm that makes its operands opaque at this point of the program,
// so that the compiler can neither sink the arithmetic that produced them below it nor keep their inputs alive past
// it (CDNA guide 5.7 item 3).  No instruction is emitted.
#ifdef NSP_HOST_EMULATION
#define NSP_PIN4(a, b, c, d) ((void)0)
#else
#define NSP_PIN4(a, b, c, d) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
#endif

#define NSP_LAUNCH_CHECK()                         \
  do {                                             \
    hipError_t e__ = hipGetLastError();            \
    if (e__ != hipSuccess) return (int)e__;        \
  } while (0)

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;

static inline int nsp_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// v_rcp_f32 / v_exp_f32 based: an IEEE divide costs ~10 VALU ops and doubled the time of
// activation-carrying GEMM epilogues; 1-ulp reciprocal error is far inside every tolerance here
__device__ __forceinline__ float nsp_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float nsp_sigmoid(float x) { return nsp_rcp(1.f + __expf(-x)); }
__device__ __forceinline__ float nsp_tanh(float x) {
  // tanh(x) = 1 - 2/(exp(2x)+1); saturates correctly for |x| large (exp -> inf or 0)
  const float e = __expf(2.f * x);
  return 1.f - 2.f * nsp_rcp(e + 1.f);
}

__device__ __forceinline__ float nsp_act(float v, int act) {
  switch (act) {
    case NSP_ACT_RELU: return v > 0.f ? v : 0.f;
    case NSP_ACT_SWISH: return v * nsp_sigmoid(v);
    case NSP_ACT_TANH: return nsp_tanh(v);
    case NSP_ACT_GELU: return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
    case NSP_ACT_GELU_TANH: {
      float u = 0.7978845608028654f * (v + 0.044715f * v * v * v);
      return 0.5f * v * (1.f + nsp_tanh(u));
    }
    default: return v;
  }
}

// derivative of act evaluated at the PRE-activation value x
__device__ __forceinline__ float nsp_dact(float x, int act) {
  switch (act) {
    case NSP_ACT_RELU: return x > 0.f ? 1.f : 0.f;
    case NSP_ACT_SWISH: {
      float s = nsp_sigmoid(x);
      return s * (1.f + x * (1.f - s));
    }
    case NSP_ACT_TANH: {
      float t = nsp_tanh(x);
      return 1.f - t * t;
    }
    case NSP_ACT_GELU: {
      float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752f));
      float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
      return cdf + x * pdf;
    }
    case NSP_ACT_GELU_TANH: {
      float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
      float t = nsp_tanh(u);
      float du = 0.7978845608028654f * (1.f + 3.f * 0.044715f * x * x);
      return 0.5f * (1.f + t) + 0.5f * x * (1.f - t * t) * du;
    }
    case NSP_ACT_TANH_OUT: return 1.f - x * x;   // x is the activation OUTPUT here
    default: return 1.f;
  }
}

// Counter-based RNG for dropout masks: a keep decision is a pure function of
// (seed, offset + element index), so backward regenerates the forward mask.
__device__ __forceinline__ uint32_t nsp_hash_u32(unsigned long long seed, unsigned long long idx) {
  // 32-bit murmur3-style finaliser over (idx_lo, idx_hi ^ seed): three 32-bit multiplies
  // (the 64-bit splitmix used before cost ~4x more VALU in the GEMM epilogues)
  uint32_t x = (uint32_t)idx ^ ((uint32_t)seed * 0x9E3779B9u);
  uint32_t hi = (uint32_t)(idx >> 32) ^ (uint32_t)(seed >> 32);
  x ^= hi * 0x85EBCA6Bu + 0x632BE5ABu;
  x ^= x >> 16; x *= 0x85EBCA6Bu;
  x ^= x >> 13; x *= 0xC2B2AE35u;
  x ^= x >> 16;
  return x;
}
// One 32-bit mix decides TWO adjacent elements (16 bits each): element idx uses the low / high half
// of hash(seed, idx >> 1).  The per-element hash was ~45 % of the VALU work of a dropout-carrying
// GEMM epilogue; the drop probability is quantised to 1/65536.
__device__ __forceinline__ float nsp_keep_scale(unsigned long long seed, unsigned long long idx,
                                                float p) {
  // returns 0 (dropped) or 1/(1-p)
  const uint32_t h = nsp_hash_u32(seed, idx >> 1);
  const uint32_t bits = (idx & 1ull) ? (h >> 16) : (h & 0xFFFFu);
  return bits < (uint32_t)(p * 65536.f) ? 0.f : nsp_rcp(1.f - p);
}
// the same function for 4 consecutive elements base .. base+3 (two mixes when base is even)
__device__ __forceinline__ void nsp_keep_scale4(unsigned long long seed, unsigned long long base, float p,
                                                float (&k)[4]) {
  if ((base & 1ull) == 0ull) {
    const uint32_t thr = (uint32_t)(p * 65536.f);
    const float inv = nsp_rcp(1.f - p);
    const uint32_t h0 = nsp_hash_u32(seed, base >> 1);
    const uint32_t h1 = nsp_hash_u32(seed, (base >> 1) + 1ull);
    k[0] = (h0 & 0xFFFFu) < thr ? 0.f : inv;
    k[1] = (h0 >> 16) < thr ? 0.f : inv;
    k[2] = (h1 & 0xFFFFu) < thr ? 0.f : inv;
    k[3] = (h1 >> 16) < thr ? 0.f : inv;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) k[e] = nsp_keep_scale(seed, base + (unsigned long long)e, p);
  }
}

__device__ __forceinline__ float wave_reduce_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_reduce_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// block-wide reductions for blockDim.x <= 1024 (multiple of 64); `sh` holds >= 16 floats
__device__ __forceinline__ float block_reduce_sum(float v, float* sh) {
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_reduce_sum(v);
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < nw; ++i) r += sh[i];
  return r;
}
__device__ __forceinline__ float block_reduce_max(float v, float* sh) {
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_reduce_max(v);
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  float r = -FLT_MAX;
  for (int i = 0; i < nw; ++i) r = fmaxf(r, sh[i]);
  return r;
}

__device__ __forceinline__ float nsp_logaddexp(float a, float b) {
  float m = fmaxf(a, b);
  if (m == -INFINITY) return -INFINITY;
  return m + log1pf(__expf(-fabsf(a - b)));
}
