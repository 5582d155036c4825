// norm_subsample.hip -- encoder variants beside the benchmarked configuration:
//
//  * the time subsamplers other than max-pool (reference encoders/subsampling.py:13-160,212-246):
//      drop / add / mean_pool = strided window SUMS  y[b,to,:] = s(to) * sum_j x[b, to*stride + j - pad, :]
//      concat / conv1d        = strided window GATHERS (im2col over time) feeding the MFMA GEMM
//  * BatchNorm1d / GroupNorm(groups of 2 channels) + Swish of the Conformer convolution module on the
//    flattened `[B*T, C]` rows (reference modules/conformer_convolution.py:58-66,119-124).
//
// Everything here is HBM-bound streaming over channels-last rows: a lane owns 4 adjacent channels
// (16 B), a wave a 1-KiB row segment.  Column reductions (batch statistics, gamma/beta gradients) are
// two-level and deterministic: row slabs -> per-slab partial sums `[S][2][C]` -> one pass over the slabs;
// no atomics.  No inline asm / gfx950 builtins in this file, so tests/hipemu can also run it on host
// threads (`-m "not gpu"` tests; there was no GPU time left in the round that added it).
#include "common.h"

namespace {

inline int ew_grid(long long n) {
  long long g = (n + 255) / 256;
  if (g > 256 * 16) g = 256 * 16;
  if (g < 1) g = 1;
  return (int)g;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 f4(float v) { return make_float4(v, v, v, v); }

// ---------------------------------------------------------------------------------------------
// window sums over time.  AvgPool1d(ceil_mode, padding 0) divides a clipped window by the number of
// frames it really covers (count_include_pad only counts *padding*, and there is none).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float window_scale(int to, int T, int k, int stride, int pad, int mean) {
  if (!mean) return 1.f;
  int t0 = to * stride - pad, t1 = t0 + k;
  if (t0 < 0) t0 = 0;
  if (t1 > T) t1 = T;
  const int cnt = t1 - t0;
  return cnt > 0 ? 1.f / (float)cnt : 0.f;
}

__global__ __launch_bounds__(256) void window_sum_fwd_kernel(const float* __restrict__ x,
                                                             float* __restrict__ y, int B, int T, int To,
                                                             int C, int k, int stride, int pad, int mean) {
  const int C4 = C >> 2;
  const long long total = (long long)B * To * C4;
  const long long gstride = (long long)gridDim.x * blockDim.x;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gstride) {
    const int c4 = (int)(idx % C4);
    const int to = (int)((idx / C4) % To);
    const long long b = idx / ((long long)C4 * To);
    float4 acc = f4(0.f);
    for (int j = 0; j < k; ++j) {
      const int t = to * stride + j - pad;
      if (t < 0 || t >= T) continue;
      const float4 v = ld4(x + ((b * T + t) * C) + 4 * c4);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    const float s = window_scale(to, T, k, stride, pad, mean);
    st4(y + 4 * idx, make_float4(acc.x * s, acc.y * s, acc.z * s, acc.w * s));
  }
}

// one thread per INPUT quad: gathers from every window that covers frame t
__global__ __launch_bounds__(256) void window_sum_bwd_kernel(const float* __restrict__ dy,
                                                             float* __restrict__ dx, int B, int T, int To,
                                                             int C, int k, int stride, int pad, int mean) {
  const int C4 = C >> 2;
  const long long total = (long long)B * T * C4;
  const long long gstride = (long long)gridDim.x * blockDim.x;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gstride) {
    const int c4 = (int)(idx % C4);
    const int t = (int)((idx / C4) % T);
    const long long b = idx / ((long long)C4 * T);
    float4 acc = f4(0.f);
    for (int j = 0; j < k; ++j) {
      const int num = t + pad - j;
      if (num < 0 || num % stride != 0) continue;
      const int to = num / stride;
      if (to >= To) continue;
      const float s = window_scale(to, T, k, stride, pad, mean);
      const float4 g = ld4(dy + ((b * To + to) * C) + 4 * c4);
      acc.x += g.x * s; acc.y += g.y * s; acc.z += g.z * s; acc.w += g.w * s;
    }
    st4(dx + 4 * idx, acc);
  }
}

// window gather: y[b,to, j*C + c] = x[b, to*stride + j - pad, c] (0 outside [0,T))
__global__ __launch_bounds__(256) void window_gather_fwd_kernel(const float* __restrict__ x,
                                                                float* __restrict__ y, int B, int T,
                                                                int To, int C, int k, int stride, int pad) {
  const int C4 = C >> 2;
  const long long total = (long long)B * To * k * C4;
  const long long gstride = (long long)gridDim.x * blockDim.x;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gstride) {
    const int c4 = (int)(idx % C4);
    const int j = (int)((idx / C4) % k);
    const int to = (int)((idx / ((long long)C4 * k)) % To);
    const long long b = idx / ((long long)C4 * k * To);
    const int t = to * stride + j - pad;
    float4 v = f4(0.f);
    if (t >= 0 && t < T) v = ld4(x + ((b * T + t) * C) + 4 * c4);
    st4(y + 4 * idx, v);
  }
}

__global__ __launch_bounds__(256) void window_gather_bwd_kernel(const float* __restrict__ dy,
                                                                float* __restrict__ dx, int B, int T,
                                                                int To, int C, int k, int stride, int pad) {
  const int C4 = C >> 2;
  const long long total = (long long)B * T * C4;
  const long long gstride = (long long)gridDim.x * blockDim.x;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gstride) {
    const int c4 = (int)(idx % C4);
    const int t = (int)((idx / C4) % T);
    const long long b = idx / ((long long)C4 * T);
    float4 acc = f4(0.f);
    for (int j = 0; j < k; ++j) {
      const int num = t + pad - j;
      if (num < 0 || num % stride != 0) continue;
      const int to = num / stride;
      if (to >= To) continue;
      const float4 g = ld4(dy + (((b * To + to) * k + j) * (long long)C) + 4 * c4);
      acc.x += g.x; acc.y += g.y; acc.z += g.z; acc.w += g.w;
    }
    st4(dx + 4 * idx, acc);
  }
}

// ---------------------------------------------------------------------------------------------
// im2col of a 3x3 / pad 1 / stride 1 convolution on channels-last x [B,T,F,Ci] for input-channel counts the MFMA
// conv kernels do not take (conv_in_channel = 3: static + delta + delta-delta features of the TIMIT / WSJ recipes,
// conv.py:167-175): cols[(b,t,f), ci*9 + kh*3 + kw] = x[b, t+kh-1, f+kw-1, ci] (0 outside), columns 9*Ci..Kp-1 = 0,
// which is the column order of nn.Conv2d's weight viewed as [Co, Ci*9]; the convolution is then one GEMM.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void im2col3x3_kernel(const float* __restrict__ x, float* __restrict__ cols,
                                                        int B, int T, int F, int Ci, int Kp) {
  const long long total = (long long)B * T * F * Kp;
  const long long gstride = (long long)gridDim.x * blockDim.x;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gstride) {
    const int col = (int)(idx % Kp);
    const long long pix = idx / Kp;
    const int f = (int)(pix % F);
    const int t = (int)((pix / F) % T);
    const long long b = pix / ((long long)F * T);
    float v = 0.f;
    if (col < 9 * Ci) {
      const int ci = col / 9, kh = (col % 9) / 3, kw = col % 3;
      const int tt = t + kh - 1, ff = f + kw - 1;
      if (tt >= 0 && tt < T && ff >= 0 && ff < F) v = x[((b * T + tt) * F + ff) * Ci + ci];
    }
    cols[idx] = v;
  }
}

// ---------------------------------------------------------------------------------------------
// What pack_padded_sequence / pad_packed_sequence do around a (B)LSTM (encoders/rnn.py:534-541), as data movement:
//   y[b,t,:] = t < len_b ? x[b, flip ? len_b-1-t : t, :] : 0        (rows of x / y may be strided: x_ld, y_ld)
// flip = 0 zeroes the frames past each utterance's end (what the padded output of a packed LSTM holds);
// flip = 1 additionally reverses every utterance inside its OWN length, so a left-to-right LSTM over the result
// is the backward direction of a packed bidirectional LSTM.  The op is its own adjoint.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void time_flip_mask_kernel(const float* __restrict__ x, long long x_ld,
                                                             float* __restrict__ y, long long y_ld,
                                                             const int* __restrict__ lens, int B, int T,
                                                             int C, int flip) {
  const int C4 = C >> 2;
  const long long total = (long long)B * T * C4;
  const long long gstride = (long long)gridDim.x * blockDim.x;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gstride) {
    const int c4 = (int)(idx % C4);
    const int t = (int)((idx / C4) % T);
    const long long b = idx / ((long long)C4 * T);
    int len = lens[b];
    if (len > T) len = T;
    float4 v = f4(0.f);
    if (t < len) v = ld4(x + (b * T + (flip ? len - 1 - t : t)) * x_ld + 4 * c4);
    st4(y + (b * T + t) * y_ld + 4 * c4, v);
  }
}

// ---------------------------------------------------------------------------------------------
// per-column parameters of the normalisations, one float4 of channels per lane
// ---------------------------------------------------------------------------------------------
struct ColParams {
  float4 mean, rstd, gamma, beta;
};

__device__ __forceinline__ float4 rstd_of(float4 scale, int scale_is_var, float eps) {
  if (!scale_is_var) return scale;
  return make_float4(1.f / sqrtf(scale.x + eps), 1.f / sqrtf(scale.y + eps), 1.f / sqrtf(scale.z + eps),
                     1.f / sqrtf(scale.w + eps));
}

__device__ __forceinline__ ColParams load_bn_params(const float* mean, const float* scale, int scale_is_var,
                                                    float eps, const float* gamma, const float* beta,
                                                    int c4) {
  ColParams p;
  p.mean = ld4(mean + 4 * c4);
  p.rstd = rstd_of(ld4(scale + 4 * c4), scale_is_var, eps);
  p.gamma = ld4(gamma + 4 * c4);
  p.beta = ld4(beta + 4 * c4);
  return p;
}

// GroupNorm over a group of two channels (a, b): mean (a+b)/2, biased variance ((a-b)/2)^2
__device__ __forceinline__ void gn2_hat(float a, float b, float eps, float& ha, float& hb, float& rstd) {
  const float d = 0.5f * (a - b);
  rstd = 1.f / sqrtf(d * d + eps);
  ha = d * rstd;
  hb = -ha;
}

// ---------------------------------------------------------------------------------------------
// two column sums over the rows of one slab.  block = 64 lanes (channel quads) x 4 row groups.
//   MODE 0  moments about a per-column shift p0:     s1 += x - p0          s2 += (x - p0)^2
//   MODE 1  BatchNorm backward:  xh = (x-mean)*rstd, dz = dy * act'(gamma*xh+beta):  s1 += dz, s2 += dz*xh
//   MODE 2  GroupNorm(2) backward: the same with xh from the channel pair
// part: [S][2][C]
// ---------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void col_reduce2_kernel(const float* __restrict__ x,
                                                          const float* __restrict__ dy,
                                                          const float* __restrict__ p_mean,
                                                          const float* __restrict__ p_scale,
                                                          int scale_is_var, float eps,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, int act,
                                                          float* __restrict__ part, long long M, int C,
                                                          int rows_per_slab) {
  __shared__ float4 sh[2][4][64];
  const int C4 = C >> 2;
  const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int c4 = blockIdx.x * 64 + lane;
  const bool active = c4 < C4;
  const long long r0 = (long long)blockIdx.y * rows_per_slab;
  long long r1 = r0 + rows_per_slab;
  if (r1 > M) r1 = M;
  float4 s1 = f4(0.f), s2 = f4(0.f);
  if (active) {
    ColParams p;
    p.mean = f4(0.f); p.rstd = f4(1.f); p.gamma = f4(1.f); p.beta = f4(0.f);
    if (MODE == 0) p.mean = ld4(p_mean + 4 * c4);
    if (MODE == 1) p = load_bn_params(p_mean, p_scale, scale_is_var, eps, gamma, beta, c4);
    if (MODE == 2) { p.gamma = ld4(gamma + 4 * c4); p.beta = ld4(beta + 4 * c4); }
    for (long long r = r0 + rg; r < r1; r += 4) {
      const float4 v = ld4(x + r * C + 4 * c4);
      if (MODE == 0) {
        const float a = v.x - p.mean.x, b = v.y - p.mean.y, c = v.z - p.mean.z, d = v.w - p.mean.w;
        s1.x += a; s1.y += b; s1.z += c; s1.w += d;
        s2.x += a * a; s2.y += b * b; s2.z += c * c; s2.w += d * d;
      } else {
        const float4 g = ld4(dy + r * C + 4 * c4);
        float4 h;
        if (MODE == 1) {
          h = make_float4((v.x - p.mean.x) * p.rstd.x, (v.y - p.mean.y) * p.rstd.y,
                          (v.z - p.mean.z) * p.rstd.z, (v.w - p.mean.w) * p.rstd.w);
        } else {
          float r_;
          gn2_hat(v.x, v.y, eps, h.x, h.y, r_);
          gn2_hat(v.z, v.w, eps, h.z, h.w, r_);
        }
        const float dzx = g.x * nsp_dact(p.gamma.x * h.x + p.beta.x, act);
        const float dzy = g.y * nsp_dact(p.gamma.y * h.y + p.beta.y, act);
        const float dzz = g.z * nsp_dact(p.gamma.z * h.z + p.beta.z, act);
        const float dzw = g.w * nsp_dact(p.gamma.w * h.w + p.beta.w, act);
        s1.x += dzx; s1.y += dzy; s1.z += dzz; s1.w += dzw;
        s2.x += dzx * h.x; s2.y += dzy * h.y; s2.z += dzz * h.z; s2.w += dzw * h.w;
      }
    }
  }
  sh[0][rg][lane] = s1;
  sh[1][rg][lane] = s2;
  __syncthreads();
  if (rg < 2 && active) {  // row group 0 folds the s1 sums, row group 1 the s2 sums
    float4 a = sh[rg][0][lane];
    for (int q = 1; q < 4; ++q) {
      const float4 b = sh[rg][q][lane];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    st4(part + ((long long)blockIdx.y * 2 + rg) * C + 4 * c4, a);
  }
}

// one thread per channel: fold the S slabs
__device__ __forceinline__ void fold_slabs(const float* part, int S, int C, int c, float& s1, float& s2) {
  s1 = 0.f;
  s2 = 0.f;
  for (int s = 0; s < S; ++s) {
    s1 += part[((long long)s * 2 + 0) * C + c];
    s2 += part[((long long)s * 2 + 1) * C + c];
  }
}

// BatchNorm1d training statistics from shifted moments (shift = row 0 of x: a sample of the data keeps
// E[d^2] - E[d]^2 free of cancellation) + the running-statistics update of nn.BatchNorm1d
// (momentum m: running = (1-m) running + m batch, variance unbiased).
__global__ __launch_bounds__(256) void bn_stats_finalize_kernel(const float* __restrict__ part, int S,
                                                                const float* __restrict__ shift,
                                                                long long M, int C, float eps,
                                                                float momentum, float* __restrict__ mean,
                                                                float* __restrict__ rstd,
                                                                float* running_mean, float* running_var,
                                                                long long* num_batches_tracked) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) {
    float s1, s2;
    fold_slabs(part, S, C, c, s1, s2);
    const float inv = 1.f / (float)M;
    const float m1 = s1 * inv;
    float var = s2 * inv - m1 * m1;
    if (var < 0.f) var = 0.f;
    const float mu = shift[c] + m1;
    mean[c] = mu;
    rstd[c] = 1.f / sqrtf(var + eps);
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
    if (running_var) {
      const float unb = M > 1 ? var * ((float)M / (float)(M - 1)) : var;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * unb;
    }
  }
  if (c == 0 && num_batches_tracked) num_batches_tracked[0] += 1;
}

__global__ __launch_bounds__(256) void col_grads_finalize_kernel(const float* __restrict__ part, int S,
                                                                 int C, float* __restrict__ dgamma,
                                                                 float* __restrict__ dbeta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) {
    float s1, s2;
    fold_slabs(part, S, C, c, s1, s2);
    dbeta[c] = s1;
    dgamma[c] = s2;
  }
}

// y = act(gamma * (x - mean) * rstd + beta)
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const float* __restrict__ x,
                                                         const float* __restrict__ mean,
                                                         const float* __restrict__ scale,
                                                         int scale_is_var, float eps,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, int act,
                                                         float* __restrict__ y, long long M, int C) {
  const int C4 = C >> 2;
  const long long total = M * C4;
  const long long gstride = (long long)gridDim.x * blockDim.x;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gstride) {
    const int c4 = (int)(idx % C4);
    const ColParams p = load_bn_params(mean, scale, scale_is_var, eps, gamma, beta, c4);
    const float4 v = ld4(x + 4 * idx);
    float4 o;
    o.x = nsp_act(p.gamma.x * ((v.x - p.mean.x) * p.rstd.x) + p.beta.x, act);
    o.y = nsp_act(p.gamma.y * ((v.y - p.mean.y) * p.rstd.y) + p.beta.y, act);
    o.z = nsp_act(p.gamma.z * ((v.z - p.mean.z) * p.rstd.z) + p.beta.z, act);
    o.w = nsp_act(p.gamma.w * ((v.w - p.mean.w) * p.rstd.w) + p.beta.w, act);
    st4(y + 4 * idx, o);
  }
}

// training: dx = gamma * rstd * (dz - (dbeta + xh * dgamma) / M);  eval (fixed statistics): gamma*rstd*dz
__global__ __launch_bounds__(256) void bn_act_bwd_apply_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ mean,
    const float* __restrict__ scale, int scale_is_var, float eps, const float* __restrict__ gamma,
    const float* __restrict__ beta, int act, const float* __restrict__ dgamma,
    const float* __restrict__ dbeta, int training, float* __restrict__ dx, long long M, int C) {
  const int C4 = C >> 2;
  const long long total = M * C4;
  const long long gstride = (long long)gridDim.x * blockDim.x;
  const float invM = training ? 1.f / (float)M : 0.f;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gstride) {
    const int c4 = (int)(idx % C4);
    const ColParams p = load_bn_params(mean, scale, scale_is_var, eps, gamma, beta, c4);
    const float4 dg = ld4(dgamma + 4 * c4), db = ld4(dbeta + 4 * c4);
    const float4 v = ld4(x + 4 * idx), g = ld4(dy + 4 * idx);
    float4 o;
#define NSP_BN_BWD(F)                                                     \
  {                                                                       \
    const float h = (v.F - p.mean.F) * p.rstd.F;                          \
    const float dz = g.F * nsp_dact(p.gamma.F * h + p.beta.F, act);       \
    o.F = p.gamma.F * p.rstd.F * (dz - (db.F + h * dg.F) * invM);         \
  }
    NSP_BN_BWD(x) NSP_BN_BWD(y) NSP_BN_BWD(z) NSP_BN_BWD(w)
#undef NSP_BN_BWD
    st4(dx + 4 * idx, o);
  }
}

// GroupNorm(groups of two adjacent channels) + activation
__global__ __launch_bounds__(256) void gn2_act_fwd_kernel(const float* __restrict__ x,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps,
                                                          int act, float* __restrict__ y, long long M,
                                                          int C) {
  const int C4 = C >> 2;
  const long long total = M * C4;
  const long long gstride = (long long)gridDim.x * blockDim.x;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gstride) {
    const int c4 = (int)(idx % C4);
    const float4 ga = ld4(gamma + 4 * c4), be = ld4(beta + 4 * c4);
    const float4 v = ld4(x + 4 * idx);
    float4 h;
    float r_;
    gn2_hat(v.x, v.y, eps, h.x, h.y, r_);
    gn2_hat(v.z, v.w, eps, h.z, h.w, r_);
    st4(y + 4 * idx, make_float4(nsp_act(ga.x * h.x + be.x, act), nsp_act(ga.y * h.y + be.y, act),
                                 nsp_act(ga.z * h.z + be.z, act), nsp_act(ga.w * h.w + be.w, act)));
  }
}

// per pair, n = 2:  dx_i = rstd * (g_i - mean(g) - xh_i * mean(g * xh)),  g_i = dz_i * gamma_i
__device__ __forceinline__ void gn2_bwd_pair(float a, float b, float dya, float dyb, float ga, float gb,
                                             float ba, float bb, float eps, int act, float& dxa,
                                             float& dxb) {
  float ha, hb, rstd;
  gn2_hat(a, b, eps, ha, hb, rstd);
  const float g0 = dya * nsp_dact(ga * ha + ba, act) * ga;
  const float g1 = dyb * nsp_dact(gb * hb + bb, act) * gb;
  const float mg = 0.5f * (g0 + g1);
  const float mgh = 0.5f * (g0 * ha + g1 * hb);
  dxa = rstd * (g0 - mg - ha * mgh);
  dxb = rstd * (g1 - mg - hb * mgh);
}

__global__ __launch_bounds__(256) void gn2_act_bwd_kernel(const float* __restrict__ x,
                                                          const float* __restrict__ dy,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps,
                                                          int act, float* __restrict__ dx, long long M,
                                                          int C) {
  const int C4 = C >> 2;
  const long long total = M * C4;
  const long long gstride = (long long)gridDim.x * blockDim.x;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gstride) {
    const int c4 = (int)(idx % C4);
    const float4 ga = ld4(gamma + 4 * c4), be = ld4(beta + 4 * c4);
    const float4 v = ld4(x + 4 * idx), g = ld4(dy + 4 * idx);
    float4 o;
    gn2_bwd_pair(v.x, v.y, g.x, g.y, ga.x, ga.y, be.x, be.y, eps, act, o.x, o.y);
    gn2_bwd_pair(v.z, v.w, g.z, g.w, ga.z, ga.w, be.z, be.w, eps, act, o.z, o.w);
    st4(dx + 4 * idx, o);
  }
}

inline int rows_per_slab_of(long long M, int S) { return (int)((M + S - 1) / S); }

}  // namespace

// number of row slabs the column reductions use for M rows (the caller sizes `part` = S*2*C floats)
extern "C" int nsp_col_reduce_slabs(long long M) {
  long long s = (M + 63) / 64;
  if (s > 256) s = 256;
  if (s < 1) s = 1;
  return (int)s;
}

static int window_args_ok(int B, int T, int To, int C, int k, int stride, int pad) {
  return B >= 0 && T >= 1 && To >= 0 && C >= 4 && C % 4 == 0 && k >= 1 && stride >= 1 && pad >= 0;
}

extern "C" int nsp_time_window_sum_fwd(const float* x, float* y, int B, int T, int To, int C, int k,
                                       int stride, int pad, int mean, void* stream) {
  if (!window_args_ok(B, T, To, C, k, stride, pad)) return NSP_EUNSUPPORTED;
  if ((long long)B * To == 0) return NSP_OK;
  hipLaunchKernelGGL(window_sum_fwd_kernel, dim3(ew_grid((long long)B * To * (C / 4))), dim3(256), 0,
                     (hipStream_t)stream, x, y, B, T, To, C, k, stride, pad, mean);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

extern "C" int nsp_time_window_sum_bwd(const float* dy, float* dx, int B, int T, int To, int C, int k,
                                       int stride, int pad, int mean, void* stream) {
  if (!window_args_ok(B, T, To, C, k, stride, pad)) return NSP_EUNSUPPORTED;
  if (B == 0) return NSP_OK;
  hipLaunchKernelGGL(window_sum_bwd_kernel, dim3(ew_grid((long long)B * T * (C / 4))), dim3(256), 0,
                     (hipStream_t)stream, dy, dx, B, T, To, C, k, stride, pad, mean);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

extern "C" int nsp_time_window_gather_fwd(const float* x, float* y, int B, int T, int To, int C, int k,
                                          int stride, int pad, void* stream) {
  if (!window_args_ok(B, T, To, C, k, stride, pad)) return NSP_EUNSUPPORTED;
  if ((long long)B * To == 0) return NSP_OK;
  hipLaunchKernelGGL(window_gather_fwd_kernel, dim3(ew_grid((long long)B * To * k * (C / 4))), dim3(256),
                     0, (hipStream_t)stream, x, y, B, T, To, C, k, stride, pad);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

extern "C" int nsp_time_window_gather_bwd(const float* dy, float* dx, int B, int T, int To, int C, int k,
                                          int stride, int pad, void* stream) {
  if (!window_args_ok(B, T, To, C, k, stride, pad)) return NSP_EUNSUPPORTED;
  if (B == 0) return NSP_OK;
  hipLaunchKernelGGL(window_gather_bwd_kernel, dim3(ew_grid((long long)B * T * (C / 4))), dim3(256), 0,
                     (hipStream_t)stream, dy, dx, B, T, To, C, k, stride, pad);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

extern "C" int nsp_im2col3x3(const float* x, float* cols, int B, int T, int F, int Ci, int Kp, void* stream) {
  if (B < 0 || T < 1 || F < 1 || Ci < 1 || Kp < 9 * Ci) return NSP_EUNSUPPORTED;
  if (B == 0) return NSP_OK;
  hipLaunchKernelGGL(im2col3x3_kernel, dim3(ew_grid((long long)B * T * F * Kp)), dim3(256), 0, (hipStream_t)stream,
                     x, cols, B, T, F, Ci, Kp);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

extern "C" int nsp_time_flip_mask(const float* x, long long x_ld, float* y, long long y_ld, const int* lens,
                                  int B, int T, int C, int flip, void* stream) {
  if (B < 0 || T < 1 || C < 4 || C % 4 || x_ld % 4 || y_ld % 4 || x_ld < C || y_ld < C) return NSP_EUNSUPPORTED;
  if (B == 0) return NSP_OK;
  hipLaunchKernelGGL(time_flip_mask_kernel, dim3(ew_grid((long long)B * T * (C / 4))), dim3(256), 0,
                     (hipStream_t)stream, x, x_ld, y, y_ld, lens, B, T, C, flip);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

extern "C" int nsp_bn_stats(const float* x, long long M, int C, float eps, float momentum, float* part,
                            float* mean, float* rstd, float* running_mean, float* running_var,
                            long long* num_batches_tracked, void* stream) {
  if (M < 1 || C < 4 || C % 4) return NSP_EUNSUPPORTED;
  const int S = nsp_col_reduce_slabs(M);
  const int C4 = C / 4;
  // shift = row 0 of x
  hipLaunchKernelGGL((col_reduce2_kernel<0>), dim3(nsp_cdiv(C4, 64), S), dim3(256), 0, (hipStream_t)stream,
                     x, (const float*)nullptr, x, (const float*)nullptr, 0, eps, (const float*)nullptr,
                     (const float*)nullptr, 0, part, M, C, rows_per_slab_of(M, S));
  NSP_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(nsp_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)part, S, x, M, C, eps, momentum, mean, rstd, running_mean, running_var,
                     num_batches_tracked);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

extern "C" int nsp_bn_act_fwd(const float* x, const float* mean, const float* scale, int scale_is_var,
                              float eps, const float* gamma, const float* beta, int act, float* y,
                              long long M, int C, void* stream) {
  if (M < 0 || C < 4 || C % 4) return NSP_EUNSUPPORTED;
  if (M == 0) return NSP_OK;
  hipLaunchKernelGGL(bn_act_fwd_kernel, dim3(ew_grid(M * (C / 4))), dim3(256), 0, (hipStream_t)stream, x,
                     mean, scale, scale_is_var, eps, gamma, beta, act, y, M, C);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

extern "C" int nsp_bn_act_bwd(const float* x, const float* dy, const float* mean, const float* scale,
                              int scale_is_var, float eps, const float* gamma, const float* beta, int act,
                              int training, float* part, float* dgamma, float* dbeta, float* dx,
                              long long M, int C, void* stream) {
  if (M < 1 || C < 4 || C % 4) return NSP_EUNSUPPORTED;
  const int S = nsp_col_reduce_slabs(M);
  const int C4 = C / 4;
  hipLaunchKernelGGL((col_reduce2_kernel<1>), dim3(nsp_cdiv(C4, 64), S), dim3(256), 0, (hipStream_t)stream,
                     x, dy, mean, scale, scale_is_var, eps, gamma, beta, act, part, M, C,
                     rows_per_slab_of(M, S));
  NSP_LAUNCH_CHECK();
  hipLaunchKernelGGL(col_grads_finalize_kernel, dim3(nsp_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)part, S, C, dgamma, dbeta);
  NSP_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_act_bwd_apply_kernel, dim3(ew_grid(M * C4)), dim3(256), 0, (hipStream_t)stream, x,
                     dy, mean, scale, scale_is_var, eps, gamma, beta, act, (const float*)dgamma,
                     (const float*)dbeta, training, dx, M, C);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

extern "C" int nsp_gn2_act_fwd(const float* x, const float* gamma, const float* beta, float eps, int act,
                               float* y, long long M, int C, void* stream) {
  if (M < 0 || C < 4 || C % 4) return NSP_EUNSUPPORTED;
  if (M == 0) return NSP_OK;
  hipLaunchKernelGGL(gn2_act_fwd_kernel, dim3(ew_grid(M * (C / 4))), dim3(256), 0, (hipStream_t)stream, x,
                     gamma, beta, eps, act, y, M, C);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

extern "C" int nsp_gn2_act_bwd(const float* x, const float* dy, const float* gamma, const float* beta,
                               float eps, int act, float* part, float* dgamma, float* dbeta, float* dx,
                               long long M, int C, void* stream) {
  if (M < 1 || C < 4 || C % 4) return NSP_EUNSUPPORTED;
  const int S = nsp_col_reduce_slabs(M);
  const int C4 = C / 4;
  hipLaunchKernelGGL((col_reduce2_kernel<2>), dim3(nsp_cdiv(C4, 64), S), dim3(256), 0, (hipStream_t)stream,
                     x, dy, (const float*)nullptr, (const float*)nullptr, 0, eps, gamma, beta, act, part,
                     M, C, rows_per_slab_of(M, S));
  NSP_LAUNCH_CHECK();
  hipLaunchKernelGGL(col_grads_finalize_kernel, dim3(nsp_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)part, S, C, dgamma, dbeta);
  NSP_LAUNCH_CHECK();
  hipLaunchKernelGGL(gn2_act_bwd_kernel, dim3(ew_grid(M * C4)), dim3(256), 0, (hipStream_t)stream, x, dy,
                     gamma, beta, eps, act, dx, M, C);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}
