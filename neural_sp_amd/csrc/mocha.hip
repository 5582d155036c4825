// mocha.hip -- the scans of monotonic (chunkwise) attention training, one wave per row.
//
// Reference: neural_sp/models/modules/mocha/hma_train.py:12-67 (parallel_monotonic_attention: p_choose ->
// exclusive cumulative product in log space -> alpha recurrence) and mocha_train.py:13-83
// (soft_chunkwise_attention: clamped exp, moving sums over the chunk window, beta).  The reference builds both from
// ~20 small tensor ops per decoder step (sigmoid, log, cumsum, exp, clamp, pad, conv1d with a ones filter ...);
// round 2 restated them the same way with torch ops.  Here each is ONE forward and ONE backward kernel over
// `rows` independent rows of `klen` encoder frames (rows = batch x heads (x target positions) -- the LSTM / MoChA
// decoder calls them once per output step, the monotonic multi-head attention of the streaming Transformer
// decoders once per target position resp. once for all positions): a wave owns a row, a lane a contiguous
// run of ceil(klen / 64) frames; prefix sums are lane-local sums + one wave scan, reverse cumulative sums are
// "total - prefix"; the chunk windows (w <= 64 frames) are summed directly from LDS like the reference's ones
// filter (differences of fp32 prefix sums lose the small windows next to large ones), MILk (w = -1: the window
// is the whole prefix / suffix) uses the scans, as the reference's cumsum does.  fp32 throughout.
// These rows are a few hundred floats: the kernels are latency-, not bandwidth-bound; what they buy is ~40 launches
// per decoder step folded into 4, and no ATen / MIOpen kernel on the MoChA path.
#include "common.h"

namespace {

__device__ __forceinline__ float wave_incl_scan(float v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const float o = __shfl_up(v, d, 64);
    if (lane >= d) v += o;
  }
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, 64));
  return v;
}
__device__ __forceinline__ float wave_total(float incl) { return __shfl(incl, 63, 64); }
__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

// ---- alpha_j = p_j c_j sum_{k<=j} aw_prev_k / den_k,  p = (1 - lam) sigmoid(e),  c = exclusive cumprod of
// clamp(1 - p, eps, 1) (in log space),  den = clamp(c, eps, 1) or 1 (no_denom)
__global__ __launch_bounds__(64) void mono_alpha_fwd_kernel(const float* __restrict__ e, const float* __restrict__ aw,
                                                            float* __restrict__ alpha, float* __restrict__ pch,
                                                            float* __restrict__ cprod, int klen, float eps,
                                                            int no_denom, float lam) {
  const int lane = threadIdx.x;
  const long long base = (long long)blockIdx.x * klen;
  e += base; aw += base; alpha += base; pch += base; cprod += base;
  const int cpl = (klen + 63) / 64, k0 = min(klen, lane * cpl), k1 = min(klen, k0 + cpl);
  float ls = 0.f;
  for (int k = k0; k < k1; ++k) {
    const float p = (1.f - lam) * nsp_sigmoid(e[k]);
    pch[k] = p;
    ls += logf(clampf(1.f - p, eps, 1.f));
  }
  float L = wave_incl_scan(ls, lane) - ls;
  float xs = 0.f;
  for (int k = k0; k < k1; ++k) {
    const float c = expf(L);
    cprod[k] = c;
    L += logf(clampf(1.f - pch[k], eps, 1.f));
    xs += aw[k] / (no_denom ? 1.f : clampf(c, eps, 1.f));
  }
  float S = wave_incl_scan(xs, lane) - xs;
  for (int k = k0; k < k1; ++k) {
    const float c = cprod[k];
    S += aw[k] / (no_denom ? 1.f : clampf(c, eps, 1.f));
    alpha[k] = pch[k] * c * S;
  }
}

// sum over the lanes ABOVE this one (reverse exclusive scan of lane totals)
__device__ __forceinline__ float wave_rev_excl_scan(float v, int lane) {
  float t = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const float o = __shfl_down(t, d, 64);
    if (lane + d < 64) t += o;
  }
  return t - v;
}

// backward of the above: d e, d aw_prev from d alpha (clamp gradients as torch.clamp: passed inside [min, max]).
// The reverse cumulative sums (gradient of cumsum) are REAL suffix sums, walked from the end of the row: c decays
// geometrically along a row, so "total - prefix" cancels catastrophically exactly where 1 / den is largest
// (first version: 3.6 % error on d e for 50-frame rows).
__global__ __launch_bounds__(64) void mono_alpha_bwd_kernel(const float* __restrict__ dalpha, const float* __restrict__ pch,
                                                            const float* __restrict__ cprod, const float* __restrict__ aw,
                                                            float* __restrict__ de, float* __restrict__ daw, int klen,
                                                            float eps, int no_denom, float lam) {
  extern __shared__ float sm[];
  float* SS = sm;               // S_k = sum_{j <= k} aw_j / den_j
  float* DL = sm + klen;        // dL_k = dc_k c_k
  float* DP = sm + 2 * klen;    // the direct part of d p_k
  const int lane = threadIdx.x;
  const long long base = (long long)blockIdx.x * klen;
  dalpha += base; pch += base; cprod += base; aw += base; de += base; daw += base;
  const int cpl = (klen + 63) / 64, k0 = min(klen, lane * cpl), k1 = min(klen, k0 + cpl);
  float xs = 0.f, gs = 0.f;
  for (int k = k0; k < k1; ++k) {
    const float c = cprod[k];
    xs += aw[k] / (no_denom ? 1.f : clampf(c, eps, 1.f));
    gs += dalpha[k] * pch[k] * c;
  }
  float S = wave_incl_scan(xs, lane) - xs;
  for (int k = k0; k < k1; ++k) {
    S += aw[k] / (no_denom ? 1.f : clampf(cprod[k], eps, 1.f));
    SS[k] = S;
  }
  // R_k = sum_{j >= k} dalpha_j p_j c_j, from the end
  float R = wave_rev_excl_scan(gs, lane), lsum = 0.f;
  for (int k = k1 - 1; k >= k0; --k) {
    const float c = cprod[k], p = pch[k], a = aw[k], da_ = dalpha[k];
    const float den = no_denom ? 1.f : clampf(c, eps, 1.f);
    R += da_ * p * c;
    daw[k] = R / den;
    float dc = da_ * p * SS[k];
    if (!no_denom && c >= eps && c <= 1.f) dc -= R * a / (den * den);
    DL[k] = dc * c;
    DP[k] = da_ * c * SS[k];
    lsum += dc * c;
  }
  // G_k = sum_{j > k} dL_j, from the end
  float G = wave_rev_excl_scan(lsum, lane);
  for (int k = k1 - 1; k >= k0; --k) {
    const float p = pch[k], y = 1.f - p;
    float dp = DP[k];
    if (y >= eps && y <= 1.f) dp -= G / y;
    G += DL[k];
    const float sg = p / (1.f - lam);
    de[k] = dp * (1.f - lam) * sg * (1.f - sg);
  }
}

// ---- beta_i = ex_i sum_{j=i}^{i+w-1} alpha_j sf / den_j,  ex = max(exp(u - max u), 1e-5),  den_j = sum_{k=j-w+1}^{j} ex_k
// (w = -1, MILk: den_j = sum_{k<=j} ex_k, the outer sum runs to the end of the row)
__global__ __launch_bounds__(64) void chunk_beta_fwd_kernel(const float* __restrict__ u, const float* __restrict__ alpha,
                                                            float* __restrict__ beta, int klen, int w, float sf) {
  extern __shared__ float sm[];
  float* EX = sm;
  float* TT = sm + klen;
  const int lane = threadIdx.x;
  const long long base = (long long)blockIdx.x * klen;
  u += base; alpha += base; beta += base;
  const int cpl = (klen + 63) / 64, k0 = min(klen, lane * cpl), k1 = min(klen, k0 + cpl);
  float m = -FLT_MAX;
  for (int k = k0; k < k1; ++k) m = fmaxf(m, u[k]);
  m = wave_max(m);
  float es = 0.f;
  for (int k = k0; k < k1; ++k) {
    const float ex = fmaxf(expf(u[k] - m), 1e-5f);
    EX[k] = ex;
    es += ex;
  }
  __syncthreads();
  if (w > 0) {
    for (int j = k0; j < k1; ++j) {
      float den = 0.f;
      for (int k = max(0, j - w + 1); k <= j; ++k) den += EX[k];
      TT[j] = alpha[j] * sf / den;
    }
    __syncthreads();
    for (int i = k0; i < k1; ++i) {
      float s = 0.f;
      const int hi = min(klen - 1, i + w - 1);
      for (int j = i; j <= hi; ++j) s += TT[j];
      beta[i] = EX[i] * s;
    }
  } else {
    float run = wave_incl_scan(es, lane) - es, ts = 0.f;
    for (int j = k0; j < k1; ++j) {
      run += EX[j];
      const float t = alpha[j] * sf / run;
      TT[j] = t;
      ts += t;
    }
    float suf = wave_rev_excl_scan(ts, lane);            // sum_{j >= i} t_j, walked from the end of the row
    for (int i = k1 - 1; i >= k0; --i) {
      suf += TT[i];
      beta[i] = EX[i] * suf;
    }
  }
}

__global__ __launch_bounds__(64) void chunk_beta_bwd_kernel(const float* __restrict__ dbeta, const float* __restrict__ u,
                                                            const float* __restrict__ alpha, float* __restrict__ du,
                                                            float* __restrict__ dalpha, int klen, int w, float sf) {
  extern __shared__ float sm[];
  float* EX = sm;
  float* DEN = sm + klen;
  float* TT = sm + 2 * klen;
  float* G = sm + 3 * klen;     // dbeta ex, later d u before the max-shift term
  float* DD = sm + 4 * klen;    // d den
  const int lane = threadIdx.x;
  const long long base = (long long)blockIdx.x * klen;
  dbeta += base; u += base; alpha += base; du += base; dalpha += base;
  const int cpl = (klen + 63) / 64, k0 = min(klen, lane * cpl), k1 = min(klen, k0 + cpl);
  float m = -FLT_MAX;
  for (int k = k0; k < k1; ++k) m = fmaxf(m, u[k]);
  m = wave_max(m);
  int amax = klen;              // first index of the maximum (the element torch.max's gradient goes to)
  float es = 0.f, gsum = 0.f;
  for (int k = k0; k < k1; ++k) {
    const float ex = fmaxf(expf(u[k] - m), 1e-5f);
    EX[k] = ex;
    es += ex;
    const float g = dbeta[k] * ex;
    G[k] = g;
    gsum += g;
    if (u[k] == m && amax == klen) amax = k;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) amax = min(amax, __shfl_xor(amax, d, 64));
  __syncthreads();
  if (w > 0) {
    for (int j = k0; j < k1; ++j) {
      float den = 0.f;
      for (int k = max(0, j - w + 1); k <= j; ++k) den += EX[k];
      DEN[j] = den;
      TT[j] = alpha[j] * sf / den;
    }
    __syncthreads();
    for (int j = k0; j < k1; ++j) {
      float dt = 0.f;
      for (int i = max(0, j - w + 1); i <= j; ++i) dt += G[i];
      dalpha[j] = dt * sf / DEN[j];
      DD[j] = -dt * TT[j] / DEN[j];
    }
    __syncthreads();
    float acc = 0.f;
    for (int k = k0; k < k1; ++k) {
      const int hi = min(klen - 1, k + w - 1);
      float M = 0.f, dd = 0.f;
      for (int j = k; j <= hi; ++j) { M += TT[j]; dd += DD[j]; }
      const float raw = expf(u[k] - m);
      const float dup = raw >= 1e-5f ? (dbeta[k] * M + dd) * raw : 0.f;
      acc += dup;
      EX[k] = dup;              // (EX is dead by now: this loop reads only TT / DD, and each lane its own k)
    }
    const float tot = wave_total(wave_incl_scan(acc, lane));
    for (int k = k0; k < k1; ++k) du[k] = EX[k] - (k == amax ? tot : 0.f);
  } else {
    // MILk: den_j = prefix(ex); M_k = suffix(t); dt_j = prefix(g); d ex_k += suffix(d den)
    float run = wave_incl_scan(es, lane) - es, ts = 0.f;
    for (int j = k0; j < k1; ++j) {
      run += EX[j];
      DEN[j] = run;
      const float t = alpha[j] * sf / run;
      TT[j] = t;
      ts += t;
    }
    float gp = wave_incl_scan(gsum, lane) - gsum, ds = 0.f;
    for (int j = k0; j < k1; ++j) {
      gp += G[j];                                   // dt_j = sum_{i <= j} g_i
      dalpha[j] = gp * sf / DEN[j];
      const float dd = -gp * TT[j] / DEN[j];
      DD[j] = dd;
      ds += dd;
    }
    float M = wave_rev_excl_scan(ts, lane), sd = wave_rev_excl_scan(ds, lane), acc = 0.f;
    for (int k = k1 - 1; k >= k0; --k) {            // suffix sums over j >= k
      M += TT[k];
      sd += DD[k];
      const float raw = expf(u[k] - m);
      const float dup = raw >= 1e-5f ? (dbeta[k] * M + sd) * raw : 0.f;
      acc += dup;
      EX[k] = dup;
    }
    const float tot = wave_total(wave_incl_scan(acc, lane));
    for (int k = k0; k < k1; ++k) du[k] = EX[k] - (k == amax ? tot : 0.f);
  }
}

}  // namespace

extern "C" int nsp_mono_alpha_fwd(const float* e, const float* aw_prev, float* alpha, float* p_choose, float* cprod,
                                  int rows, int klen, float eps, int no_denom, float stableemit, void* stream) {
  if (rows <= 0 || klen <= 0) return NSP_OK;
  if (!e || !aw_prev || !alpha || !p_choose || !cprod || stableemit >= 1.f) return NSP_EINVAL;
  hipLaunchKernelGGL(mono_alpha_fwd_kernel, dim3(rows), dim3(64), 0, (hipStream_t)stream, e, aw_prev, alpha, p_choose,
                     cprod, klen, eps, no_denom, stableemit);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

extern "C" int nsp_mono_alpha_bwd(const float* d_alpha, const float* p_choose, const float* cprod, const float* aw_prev,
                                  float* d_e, float* d_aw_prev, int rows, int klen, float eps, int no_denom,
                                  float stableemit, void* stream) {
  if (rows <= 0 || klen <= 0) return NSP_OK;
  if (!d_alpha || !p_choose || !cprod || !aw_prev || !d_e || !d_aw_prev || stableemit >= 1.f) return NSP_EINVAL;
  if (klen > 8192) return NSP_EUNSUPPORTED;
  // the opt-in is per DEVICE and can fail (a device with less LDS than 3 x 4 B per key): set it for the bytes this
  // launch needs, on whatever device is current, and report instead of failing at launch with a generic error
  if (3 * klen * sizeof(float) > 65536 &&
      hipFuncSetAttribute((const void*)mono_alpha_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(3 * klen * sizeof(float))) != hipSuccess)
    return NSP_EUNSUPPORTED;
  hipLaunchKernelGGL(mono_alpha_bwd_kernel, dim3(rows), dim3(64), 3 * klen * sizeof(float), (hipStream_t)stream, d_alpha,
                     p_choose, cprod, aw_prev, d_e, d_aw_prev, klen, eps, no_denom, stableemit);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

extern "C" int nsp_chunk_beta_fwd(const float* u, const float* alpha, float* beta, int rows, int klen, int w, float sf,
                                  void* stream) {
  if (rows <= 0 || klen <= 0) return NSP_OK;
  if (!u || !alpha || !beta || w == 0 || w < -1) return NSP_EINVAL;
  if (w > 64 || klen > 8192) return NSP_EUNSUPPORTED;
  if (2 * klen * sizeof(float) > 65536 &&
      hipFuncSetAttribute((const void*)chunk_beta_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * klen * sizeof(float))) != hipSuccess)
    return NSP_EUNSUPPORTED;
  hipLaunchKernelGGL(chunk_beta_fwd_kernel, dim3(rows), dim3(64), 2 * klen * sizeof(float), (hipStream_t)stream, u, alpha,
                     beta, klen, w, sf);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

extern "C" int nsp_chunk_beta_bwd(const float* d_beta, const float* u, const float* alpha, float* d_u, float* d_alpha,
                                  int rows, int klen, int w, float sf, void* stream) {
  if (rows <= 0 || klen <= 0) return NSP_OK;
  if (!d_beta || !u || !alpha || !d_u || !d_alpha || w == 0 || w < -1) return NSP_EINVAL;
  if (w > 64 || klen > 8192) return NSP_EUNSUPPORTED;
  if (5 * klen * sizeof(float) > 65536 &&
      hipFuncSetAttribute((const void*)chunk_beta_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(5 * klen * sizeof(float))) != hipSuccess)
    return NSP_EUNSUPPORTED;
  hipLaunchKernelGGL(chunk_beta_bwd_kernel, dim3(rows), dim3(64), 5 * klen * sizeof(float), (hipStream_t)stream, d_beta, u,
                     alpha, d_u, d_alpha, klen, w, sf);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}
