// layernorm.hip -- row LayerNorm forward/backward, one wave64 per row, the row
// held in registers (two-pass mean/variance in fp32, matching ATen's numerics
// at eps=1e-12).  HBM-bound: forward reads x once and
// writes y once; backward reads dy, x (+y_pre) once and writes dx once, with
// dgamma/dbeta accumulated per wave in registers over a grid-stride loop of
// rows and flushed with one atomic per column per block.
#include "common.h"

namespace {

template <int VPL>  // float4 per lane; covers d <= VPL*256
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x,
                                                     const float* __restrict__ gamma,
                                                     const float* __restrict__ beta,
                                                     float* __restrict__ y, float* __restrict__ mean,
                                                     float* __restrict__ rstd, int rows, int d,
                                                     float eps, int act, float* __restrict__ y_pre,
                                                     __bf16* __restrict__ y16) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int d4 = d >> 2;
  const float inv_d = 1.f / (float)d;
  for (long long row = (long long)blockIdx.x * 4 + w; row < rows; row += (long long)gridDim.x * 4) {
    const float4* xr = reinterpret_cast<const float4*>(x + row * d);
    float4 v[VPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      int c = lane + i * 64;
      v[i] = c < d4 ? xr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
      s += v[i].x + v[i].y + v[i].z + v[i].w;
    }
    const float mu = wave_reduce_sum(s) * inv_d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      int c = lane + i * 64;
      if (c < d4) {
        float a = v[i].x - mu, b = v[i].y - mu, cc = v[i].z - mu, dd = v[i].w - mu;
        q += a * a + b * b + cc * cc + dd * dd;
      }
    }
    const float var = wave_reduce_sum(q) * inv_d;
    const float rs = 1.f / sqrtf(var + eps);
    if (lane == 0) {
      mean[row] = mu;
      rstd[row] = rs;
    }
    float4* yr = y ? reinterpret_cast<float4*>(y + row * d) : nullptr;
    float4* ypr = y_pre ? reinterpret_cast<float4*>(y_pre + row * d) : nullptr;
    bf16x4* y16r = y16 ? reinterpret_cast<bf16x4*>(y16 + row * d) : nullptr;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      int c = lane + i * 64;
      if (c < d4) {
        float4 g = reinterpret_cast<const float4*>(gamma)[c];
        float4 b = reinterpret_cast<const float4*>(beta)[c];
        float4 o = make_float4((v[i].x - mu) * rs * g.x + b.x, (v[i].y - mu) * rs * g.y + b.y,
                               (v[i].z - mu) * rs * g.z + b.z, (v[i].w - mu) * rs * g.w + b.w);
        if (ypr) ypr[c] = o;
        if (act != NSP_ACT_NONE)
          o = make_float4(nsp_act(o.x, act), nsp_act(o.y, act), nsp_act(o.z, act), nsp_act(o.w, act));
        if (yr) yr[c] = o;
        if (y16r) {
          bf16x4 h;
          h[0] = (__bf16)o.x; h[1] = (__bf16)o.y; h[2] = (__bf16)o.z; h[3] = (__bf16)o.w;
          y16r[c] = h;
        }
      }
    }
  }
}

// PREP (round 4): dx is the gradient of the residual stream at the point between two sub-blocks, and the sub-block in
// front of it needs exactly one thing from it for its last linear layer's backward: g = alpha * dx * dropout_mask as the
// bf16 operand of the gradient GEMMs, plus its column sums (bias gradient).  That was a separate pass over dx
// (grad_prep_colsum: 4 B read + 2 B written per element, 52 launches per step); here it is two more bytes written while
// dx is still in registers.
template <int VPL, bool PREP = false>
__global__ __launch_bounds__(256) void ln_bwd_kernel(
    const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ gamma,
    const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ y_pre,
    const float* __restrict__ dres, float* __restrict__ dx, float* __restrict__ dgamma,
    float* __restrict__ dbeta, int rows, int d, int act, const float* __restrict__ beta_re,
    __bf16* __restrict__ g16 = nullptr, float* __restrict__ gsum = nullptr, float g_alpha = 1.f, float g_p = 0.f,
    unsigned long long g_seed = 0ull, unsigned long long g_offset = 0ull) {
  extern __shared__ __attribute__((aligned(16))) float sh[];  // [2][4][d]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int d4 = d >> 2;
  const float inv_d = 1.f / (float)d;
  float4 dg[VPL], db[VPL], gm[VPL], bt[VPL], gs[VPL];
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    gs[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    dg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    db[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    int c = lane + i * 64;
    gm[i] = c < d4 ? reinterpret_cast<const float4*>(gamma)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    // beta_re: the pre-activation y_pre = xhat gamma + beta is RECOMPUTED from what this kernel reads anyway
    // (nsp_layernorm_bwd_recompute: forward then stores neither the fp32 output nor the pre-activation)
    bt[i] = (beta_re && c < d4) ? reinterpret_cast<const float4*>(beta_re)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // software pipeline over the wave's rows: the loads of row r+1 are in flight while row r is
  // reduced and stored (a wave's rows were strictly serial before: load -> 2 wave reductions ->
  // store, ~3.4 TB/s with 6 waves per CU)
  const long long rstride = (long long)gridDim.x * 4;
  long long row = (long long)blockIdx.x * 4 + w;
  float4 nx[VPL], ng[VPL], np[VPL], nr[VPL];
  float nmu = 0.f, nrs = 0.f;
  auto fetch = [&](long long r) {
    nmu = mean[r];
    nrs = rstd[r];
    const float4* xr = reinterpret_cast<const float4*>(x + r * d);
    const float4* gr = reinterpret_cast<const float4*>(dy + r * d);
    const float4* pr = (act != NSP_ACT_NONE && y_pre) ? reinterpret_cast<const float4*>(y_pre + r * d) : nullptr;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = lane + i * 64;
      if (c < d4) {
        nx[i] = xr[c];
        ng[i] = gr[c];
        if (pr) np[i] = pr[c];
        if (dres) nr[i] = reinterpret_cast<const float4*>(dres + r * d)[c];
      }
    }
  };
  if (row < rows) fetch(row);
  for (; row < rows; row += rstride) {
    const float mu = nmu, rs = nrs;
    float4 xv_[VPL], gv_[VPL], pv_[VPL], rv_[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) { xv_[i] = nx[i]; gv_[i] = ng[i]; pv_[i] = np[i]; rv_[i] = nr[i]; }
    if (row + rstride < rows) fetch(row + rstride);
    float4 xh[VPL], g[VPL];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      int c = lane + i * 64;
      if (c < d4) {
        float4 xv = xv_[i];
        float4 gv = gv_[i];
        xh[i] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
        if (act != NSP_ACT_NONE) {
          float4 p;
          if (y_pre) p = pv_[i];
          else p = make_float4(fmaf(xh[i].x, gm[i].x, bt[i].x), fmaf(xh[i].y, gm[i].y, bt[i].y),
                               fmaf(xh[i].z, gm[i].z, bt[i].z), fmaf(xh[i].w, gm[i].w, bt[i].w));
          gv.x *= nsp_dact(p.x, act); gv.y *= nsp_dact(p.y, act);
          gv.z *= nsp_dact(p.z, act); gv.w *= nsp_dact(p.w, act);
        }
        db[i].x += gv.x; db[i].y += gv.y; db[i].z += gv.z; db[i].w += gv.w;
        dg[i].x += gv.x * xh[i].x; dg[i].y += gv.y * xh[i].y;
        dg[i].z += gv.z * xh[i].z; dg[i].w += gv.w * xh[i].w;
        g[i] = make_float4(gv.x * gm[i].x, gv.y * gm[i].y, gv.z * gm[i].z, gv.w * gm[i].w);
        s1 += g[i].x + g[i].y + g[i].z + g[i].w;
        s2 += g[i].x * xh[i].x + g[i].y * xh[i].y + g[i].z * xh[i].z + g[i].w * xh[i].w;
      } else {
        xh[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        g[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    s1 = wave_reduce_sum(s1) * inv_d;
    s2 = wave_reduce_sum(s2) * inv_d;
    float4* dxr = reinterpret_cast<float4*>(dx + row * d);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      int c = lane + i * 64;
      if (c < d4) {
        float4 o = make_float4(rs * (g[i].x - s1 - xh[i].x * s2), rs * (g[i].y - s1 - xh[i].y * s2),
                               rs * (g[i].z - s1 - xh[i].z * s2), rs * (g[i].w - s1 - xh[i].w * s2));
        if (dres) { o.x += rv_[i].x; o.y += rv_[i].y; o.z += rv_[i].z; o.w += rv_[i].w; }
        dxr[c] = o;
        if constexpr (PREP) {
          float kp[4];
          nsp_keep_scale4(g_seed, g_offset + (unsigned long long)(row * d + 4ll * c), g_p, kp);   // (p = 0: keeps everything, scale 1)
          const float q0 = o.x * g_alpha * kp[0], q1 = o.y * g_alpha * kp[1], q2 = o.z * g_alpha * kp[2], q3 = o.w * g_alpha * kp[3];
          bf16x4 h;
          h[0] = (__bf16)q0; h[1] = (__bf16)q1; h[2] = (__bf16)q2; h[3] = (__bf16)q3;
          reinterpret_cast<bf16x4*>(g16 + row * d)[c] = h;
          gs[i].x += q0; gs[i].y += q1; gs[i].z += q2; gs[i].w += q3;
        }
      }
    }
  }
  // combine the 4 waves' partial dgamma/dbeta through LDS, one atomic per column
  float4* shg = reinterpret_cast<float4*>(sh);            // [4][d4]
  float4* shb = reinterpret_cast<float4*>(sh) + 4 * d4;   // [4][d4]
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    int c = lane + i * 64;
    if (c < d4) {
      shg[w * d4 + c] = dg[i];
      shb[w * d4 + c] = db[i];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < d; c += blockDim.x) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int ww = 0; ww < 4; ++ww) {
      a += sh[ww * d + c];
      b += sh[4 * d + ww * d + c];
    }
    unsafeAtomicAdd(dgamma + c, a);
    unsafeAtomicAdd(dbeta + c, b);
  }
  if constexpr (PREP) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      int c = lane + i * 64;
      if (c < d4) shg[w * d4 + c] = gs[i];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < d; c += blockDim.x) {
      float a = 0.f;
#pragma unroll
      for (int ww = 0; ww < 4; ++ww) a += sh[ww * d + c];
      unsafeAtomicAdd(gsum + c, a);
    }
  }
}

// ---- Round 6: the LayerNorm pair at a block boundary.  A Conformer block ends with y = LN_5(x) and the next one starts
// with LN_1(y) in front of its first feed-forward module (conformer_block.py:176-180 / :132-133): two kernels per
// direction that handed y (forward) and the gradient of y (backward) through HBM.  Forward: y is normalised a second
// time while it is still in registers -- x read once, y (fp32, the residual stream) and the bf16 image of LN_1(y)
// written -- 10 instead of 14 bytes per element.  Backward: y is recomputed from x (it is not even read), the gradient
// of y = LN_1's backward + the residual gradient stays in registers and goes straight into LN_5's backward: 18 (+ 2
// for the prepared image) instead of 30 bytes per element.  Same arithmetic, statement for statement, as the two kernels.
template <int VPL>
__global__ __launch_bounds__(256) void ln_pair_fwd_kernel(const float* __restrict__ x, const float* __restrict__ ga,
                                                          const float* __restrict__ ba, float eps_a,
                                                          const float* __restrict__ gb, const float* __restrict__ bb,
                                                          float eps_b, float* __restrict__ y, __bf16* __restrict__ z16,
                                                          float* __restrict__ mean_a, float* __restrict__ rstd_a,
                                                          float* __restrict__ mean_b, float* __restrict__ rstd_b,
                                                          int rows, int d) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int d4 = d >> 2;
  const float inv_d = 1.f / (float)d;
  for (long long row = (long long)blockIdx.x * 4 + w; row < rows; row += (long long)gridDim.x * 4) {
    const float4* xr = reinterpret_cast<const float4*>(x + row * d);
    float4 v[VPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = lane + i * 64;
      v[i] = c < d4 ? xr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
      s += v[i].x + v[i].y + v[i].z + v[i].w;
    }
    const float mu = wave_reduce_sum(s) * inv_d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = lane + i * 64;
      if (c < d4) {
        const float a = v[i].x - mu, b = v[i].y - mu, cc = v[i].z - mu, dd = v[i].w - mu;
        q += a * a + b * b + cc * cc + dd * dd;
      }
    }
    const float rs = 1.f / sqrtf(wave_reduce_sum(q) * inv_d + eps_a);
    // y = LN_a(x), kept in v
    float s2 = 0.f;
    float4* yr = reinterpret_cast<float4*>(y + row * d);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = lane + i * 64;
      if (c < d4) {
        const float4 g = reinterpret_cast<const float4*>(ga)[c];
        const float4 b = reinterpret_cast<const float4*>(ba)[c];
        v[i] = make_float4((v[i].x - mu) * rs * g.x + b.x, (v[i].y - mu) * rs * g.y + b.y,
                           (v[i].z - mu) * rs * g.z + b.z, (v[i].w - mu) * rs * g.w + b.w);
        yr[c] = v[i];
        s2 += v[i].x + v[i].y + v[i].z + v[i].w;
      }
    }
    const float mu2 = wave_reduce_sum(s2) * inv_d;
    float q2 = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = lane + i * 64;
      if (c < d4) {
        const float a = v[i].x - mu2, b = v[i].y - mu2, cc = v[i].z - mu2, dd = v[i].w - mu2;
        q2 += a * a + b * b + cc * cc + dd * dd;
      }
    }
    const float rs2 = 1.f / sqrtf(wave_reduce_sum(q2) * inv_d + eps_b);
    if (lane == 0) { mean_a[row] = mu; rstd_a[row] = rs; mean_b[row] = mu2; rstd_b[row] = rs2; }
    bf16x4* zr = reinterpret_cast<bf16x4*>(z16 + row * d);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = lane + i * 64;
      if (c < d4) {
        const float4 g = reinterpret_cast<const float4*>(gb)[c];
        const float4 b = reinterpret_cast<const float4*>(bb)[c];
        bf16x4 h;
        h[0] = (__bf16)((v[i].x - mu2) * rs2 * g.x + b.x); h[1] = (__bf16)((v[i].y - mu2) * rs2 * g.y + b.y);
        h[2] = (__bf16)((v[i].z - mu2) * rs2 * g.z + b.z); h[3] = (__bf16)((v[i].w - mu2) * rs2 * g.w + b.w);
        zr[c] = h;
      }
    }
  }
}

// dz = gradient of LN_b(y) (from the feed-forward module's data-gradient GEMM), dres = gradient of y through the residual
// path; dx = LN_a's backward of (LN_b's backward of dz + dres); PREP: also the prepared bf16 image of dx (see ln_bwd_kernel)
template <int VPL, bool PREP>
__global__ __launch_bounds__(256) void ln_pair_bwd_kernel(
    const float* __restrict__ dz, const float* __restrict__ dres, const float* __restrict__ x,
    const float* __restrict__ ga, const float* __restrict__ ba, const float* __restrict__ mean_a, const float* __restrict__ rstd_a,
    const float* __restrict__ gb, const float* __restrict__ mean_b, const float* __restrict__ rstd_b,
    float* __restrict__ dx, float* __restrict__ dga, float* __restrict__ dba, float* __restrict__ dgb, float* __restrict__ dbb,
    int rows, int d, __bf16* __restrict__ g16, float* __restrict__ gsum, float g_alpha, float g_p,
    unsigned long long g_seed, unsigned long long g_offset) {
  extern __shared__ __attribute__((aligned(16))) float sh[];  // [4 waves][d] x 2, used twice
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int d4 = d >> 2;
  const float inv_d = 1.f / (float)d;
  float4 dGa[VPL], dBa[VPL], dGb[VPL], dBb[VPL], gs[VPL], gma[VPL], bta[VPL], gmb[VPL];
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    dGa[i] = z; dBa[i] = z; dGb[i] = z; dBb[i] = z; gs[i] = z;
    const int c = lane + i * 64;
    gma[i] = c < d4 ? reinterpret_cast<const float4*>(ga)[c] : z;
    bta[i] = c < d4 ? reinterpret_cast<const float4*>(ba)[c] : z;
    gmb[i] = c < d4 ? reinterpret_cast<const float4*>(gb)[c] : z;
  }
  const long long rstride = (long long)gridDim.x * 4;
  long long row = (long long)blockIdx.x * 4 + w;
  float4 nx[VPL], ng[VPL], nr[VPL];
  float nmua = 0.f, nrsa = 0.f, nmub = 0.f, nrsb = 0.f;
  auto fetch = [&](long long r) {
    nmua = mean_a[r]; nrsa = rstd_a[r]; nmub = mean_b[r]; nrsb = rstd_b[r];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = lane + i * 64;
      if (c < d4) {
        nx[i] = reinterpret_cast<const float4*>(x + r * d)[c];
        ng[i] = reinterpret_cast<const float4*>(dz + r * d)[c];
        nr[i] = reinterpret_cast<const float4*>(dres + r * d)[c];
      }
    }
  };
  if (row < rows) fetch(row);
  for (; row < rows; row += rstride) {
    const float mua = nmua, rsa = nrsa, mub = nmub, rsb = nrsb;
    float4 xv[VPL], gv[VPL], rv[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) { xv[i] = nx[i]; gv[i] = ng[i]; rv[i] = nr[i]; }
    if (row + rstride < rows) fetch(row + rstride);
    // ---- LN_b backward on dz: xhat_b from y = xhat_a gamma_a + beta_a (recomputed)
    float4 xa[VPL], xb[VPL], g[VPL];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = lane + i * 64;
      if (c < d4) {
        xa[i] = make_float4((xv[i].x - mua) * rsa, (xv[i].y - mua) * rsa, (xv[i].z - mua) * rsa, (xv[i].w - mua) * rsa);
        const float4 yv = make_float4((xv[i].x - mua) * rsa * gma[i].x + bta[i].x, (xv[i].y - mua) * rsa * gma[i].y + bta[i].y,
                                      (xv[i].z - mua) * rsa * gma[i].z + bta[i].z, (xv[i].w - mua) * rsa * gma[i].w + bta[i].w);
        xb[i] = make_float4((yv.x - mub) * rsb, (yv.y - mub) * rsb, (yv.z - mub) * rsb, (yv.w - mub) * rsb);
        dBb[i].x += gv[i].x; dBb[i].y += gv[i].y; dBb[i].z += gv[i].z; dBb[i].w += gv[i].w;
        dGb[i].x += gv[i].x * xb[i].x; dGb[i].y += gv[i].y * xb[i].y; dGb[i].z += gv[i].z * xb[i].z; dGb[i].w += gv[i].w * xb[i].w;
        g[i] = make_float4(gv[i].x * gmb[i].x, gv[i].y * gmb[i].y, gv[i].z * gmb[i].z, gv[i].w * gmb[i].w);
        s1 += g[i].x + g[i].y + g[i].z + g[i].w;
        s2 += g[i].x * xb[i].x + g[i].y * xb[i].y + g[i].z * xb[i].z + g[i].w * xb[i].w;
      } else {
        xa[i] = xb[i] = g[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    s1 = wave_reduce_sum(s1) * inv_d;
    s2 = wave_reduce_sum(s2) * inv_d;
    // ---- gradient of y, then LN_a backward on it
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = lane + i * 64;
      if (c < d4) {
        float4 o = make_float4(rsb * (g[i].x - s1 - xb[i].x * s2), rsb * (g[i].y - s1 - xb[i].y * s2),
                               rsb * (g[i].z - s1 - xb[i].z * s2), rsb * (g[i].w - s1 - xb[i].w * s2));
        o.x += rv[i].x; o.y += rv[i].y; o.z += rv[i].z; o.w += rv[i].w;
        dBa[i].x += o.x; dBa[i].y += o.y; dBa[i].z += o.z; dBa[i].w += o.w;
        dGa[i].x += o.x * xa[i].x; dGa[i].y += o.y * xa[i].y; dGa[i].z += o.z * xa[i].z; dGa[i].w += o.w * xa[i].w;
        g[i] = make_float4(o.x * gma[i].x, o.y * gma[i].y, o.z * gma[i].z, o.w * gma[i].w);
        t1 += g[i].x + g[i].y + g[i].z + g[i].w;
        t2 += g[i].x * xa[i].x + g[i].y * xa[i].y + g[i].z * xa[i].z + g[i].w * xa[i].w;
      }
    }
    t1 = wave_reduce_sum(t1) * inv_d;
    t2 = wave_reduce_sum(t2) * inv_d;
    float4* dxr = reinterpret_cast<float4*>(dx + row * d);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = lane + i * 64;
      if (c < d4) {
        const float4 o = make_float4(rsa * (g[i].x - t1 - xa[i].x * t2), rsa * (g[i].y - t1 - xa[i].y * t2),
                                     rsa * (g[i].z - t1 - xa[i].z * t2), rsa * (g[i].w - t1 - xa[i].w * t2));
        dxr[c] = o;
        if constexpr (PREP) {
          float kp[4];
          nsp_keep_scale4(g_seed, g_offset + (unsigned long long)(row * d + 4ll * c), g_p, kp);
          const float q0 = o.x * g_alpha * kp[0], q1 = o.y * g_alpha * kp[1], q2 = o.z * g_alpha * kp[2], q3 = o.w * g_alpha * kp[3];
          bf16x4 h;
          h[0] = (__bf16)q0; h[1] = (__bf16)q1; h[2] = (__bf16)q2; h[3] = (__bf16)q3;
          reinterpret_cast<bf16x4*>(g16 + row * d)[c] = h;
          gs[i].x += q0; gs[i].y += q1; gs[i].z += q2; gs[i].w += q3;
        }
      }
    }
  }
  // column sums: the 4 waves' partials through LDS, one atomic per column; (dgamma, dbeta) of LN_a, then of LN_b, then gsum
  float4* sh0 = reinterpret_cast<float4*>(sh);            // [4][d4]
  float4* sh1 = reinterpret_cast<float4*>(sh) + 4 * d4;   // [4][d4]
  auto flush2 = [&](const float4 (&p0)[VPL], const float4 (&p1)[VPL], float* o0, float* o1) {
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = lane + i * 64;
      if (c < d4) { sh0[w * d4 + c] = p0[i]; sh1[w * d4 + c] = p1[i]; }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < d; c += blockDim.x) {
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int ww = 0; ww < 4; ++ww) { a += sh[ww * d + c]; b += sh[4 * d + ww * d + c]; }
      unsafeAtomicAdd(o0 + c, a);
      if (o1) unsafeAtomicAdd(o1 + c, b);
    }
    __syncthreads();
  };
  flush2(dGa, dBa, dga, dba);
  flush2(dGb, dBb, dgb, dbb);
  if constexpr (PREP) flush2(gs, gs, gsum, nullptr);
}

}  // namespace

extern "C" int nsp_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y,
                                 float* mean, float* rstd, int rows, int d, float eps, int act,
                                 float* y_pre, void* y16v, void* stream) {
  if (d % 4 || d > 2048 || rows <= 0) return NSP_EUNSUPPORTED;
  if (!y && !y16v) return NSP_EINVAL;
  __bf16* y16 = reinterpret_cast<__bf16*>(y16v);
  hipStream_t st = (hipStream_t)stream;
  int grid = nsp_cdiv(rows, 4);
  if (grid > 4096) grid = 4096;
  const int vpl = nsp_cdiv(d, 256);
#define LN_FWD(V) hipLaunchKernelGGL((ln_fwd_kernel<V>), dim3(grid), dim3(256), 0, st, x, gamma, beta, y, mean, rstd, rows, d, eps, act, y_pre, y16)
  if (vpl <= 1) LN_FWD(1);
  else if (vpl <= 2) LN_FWD(2);
  else if (vpl <= 4) LN_FWD(4);
  else LN_FWD(8);
#undef LN_FWD
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

static int ln_bwd_launch(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                         const float* y_pre, const float* beta_re, const float* dres, float* dx, float* dgamma,
                         float* dbeta, int rows, int d, int act, void* stream, void* g16 = nullptr, float* gsum = nullptr,
                         float g_alpha = 1.f, float g_p = 0.f, unsigned long long g_seed = 0ull,
                         unsigned long long g_offset = 0ull);

extern "C" int nsp_layernorm_bwd(const float* dy, const float* x, const float* gamma,
                                 const float* mean, const float* rstd, const float* y_pre,
                                 const float* dres, float* dx, float* dgamma, float* dbeta, int rows,
                                 int d, int act, void* stream) {
  if (act != NSP_ACT_NONE && !y_pre) return NSP_EINVAL;
  return ln_bwd_launch(dy, x, gamma, mean, rstd, y_pre, nullptr, dres, dx, dgamma, dbeta, rows, d, act, stream);
}

extern "C" int nsp_layernorm_bwd_recompute(const float* dy, const float* x, const float* gamma, const float* beta,
                                           const float* mean, const float* rstd, const float* dres, float* dx,
                                           float* dgamma, float* dbeta, int rows, int d, int act, void* stream) {
  if (!beta) return NSP_EINVAL;
  return ln_bwd_launch(dy, x, gamma, mean, rstd, nullptr, beta, dres, dx, dgamma, dbeta, rows, d, act, stream);
}

extern "C" int nsp_layernorm_bwd_prep(const float* dy, const float* x, const float* gamma, const float* mean,
                                      const float* rstd, const float* dres, float* dx, float* dgamma, float* dbeta,
                                      void* g16, float* gsum, float g_alpha, float g_p, unsigned long long g_seed,
                                      unsigned long long g_offset, int rows, int d, void* stream) {
  if (!g16 || !gsum || d % 8 || (reinterpret_cast<uintptr_t>(g16) & 7)) return NSP_EINVAL;
  if (g_p < 0.f || g_p >= 1.f) return NSP_EINVAL;
  return ln_bwd_launch(dy, x, gamma, mean, rstd, nullptr, nullptr, dres, dx, dgamma, dbeta, rows, d, NSP_ACT_NONE, stream,
                       g16, gsum, g_alpha, g_p, g_seed, g_offset);
}

static int ln_bwd_launch(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                         const float* y_pre, const float* beta_re, const float* dres, float* dx, float* dgamma,
                         float* dbeta, int rows, int d, int act, void* stream, void* g16, float* gsum, float g_alpha,
                         float g_p, unsigned long long g_seed, unsigned long long g_offset) {
  if (d % 4 || d > 2048 || rows <= 0) return NSP_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  int grid = nsp_cdiv(rows, 4 * 8);  // >= 8 rows per wave to amortise the column atomics
  if (grid > 1024) grid = 1024;
  if (grid < 1) grid = 1;
  const size_t shmem = sizeof(float) * 8 * d;
  const int vpl = nsp_cdiv(d, 256);
#define LN_BWD(V) hipLaunchKernelGGL((ln_bwd_kernel<V>), dim3(grid), dim3(256), shmem, st, dy, x, gamma, mean, rstd, y_pre, dres, dx, dgamma, dbeta, rows, d, act, beta_re)
#define LN_BWDP(V) hipLaunchKernelGGL((ln_bwd_kernel<V, true>), dim3(grid), dim3(256), shmem, st, dy, x, gamma, mean, rstd, y_pre, dres, dx, dgamma, dbeta, rows, d, act, beta_re, reinterpret_cast<__bf16*>(g16), gsum, g_alpha, g_p, g_seed, g_offset)
  if (g16) {
    if (vpl <= 1) LN_BWDP(1);
    else if (vpl <= 2) LN_BWDP(2);
    else if (vpl <= 4) LN_BWDP(4);
    else LN_BWDP(8);
  } else if (vpl <= 1) LN_BWD(1);
  else if (vpl <= 2) LN_BWD(2);
  else if (vpl <= 4) LN_BWD(4);
  else LN_BWD(8);
#undef LN_BWD
#undef LN_BWDP
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

// ---- LayerNorm pair at a block boundary (see ln_pair_fwd_kernel): y = LN_a(x) fp32, z16 = bf16(LN_b(y))
extern "C" int nsp_layernorm_pair_fwd(const float* x, const float* gamma_a, const float* beta_a, float eps_a,
                                      const float* gamma_b, const float* beta_b, float eps_b, float* y, void* z16,
                                      float* mean_a, float* rstd_a, float* mean_b, float* rstd_b, int rows, int d,
                                      void* stream) {
  if (d % 4 || d > 2048 || rows <= 0 || (reinterpret_cast<uintptr_t>(z16) & 7)) return NSP_EUNSUPPORTED;
  if (!y || !z16) return NSP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  int grid = nsp_cdiv(rows, 4);
  if (grid > 4096) grid = 4096;
  const int vpl = nsp_cdiv(d, 256);
#define LNP(V) hipLaunchKernelGGL((ln_pair_fwd_kernel<V>), dim3(grid), dim3(256), 0, st, x, gamma_a, beta_a, eps_a, gamma_b, beta_b, eps_b, y, reinterpret_cast<__bf16*>(z16), mean_a, rstd_a, mean_b, rstd_b, rows, d)
  if (vpl <= 1) LNP(1);
  else if (vpl <= 2) LNP(2);
  else if (vpl <= 4) LNP(4);
  else LNP(8);
#undef LNP
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

// dgamma / dbeta (both norms) and gsum are ACCUMULATED (caller-zeroed); g16 / gsum optional (both or neither: the prepared
// image of dx as in nsp_layernorm_bwd_prep)
extern "C" int nsp_layernorm_pair_bwd(const float* dz, const float* dres, const float* x, const float* gamma_a,
                                      const float* beta_a, const float* mean_a, const float* rstd_a, const float* gamma_b,
                                      const float* mean_b, const float* rstd_b, float* dx, float* dgamma_a, float* dbeta_a,
                                      float* dgamma_b, float* dbeta_b, void* g16, float* gsum, float g_alpha, float g_p,
                                      unsigned long long g_seed, unsigned long long g_offset, int rows, int d, void* stream) {
  if (d % 4 || d > 2048 || rows <= 0) return NSP_EUNSUPPORTED;
  if (!dz || !dres || (g16 != nullptr) != (gsum != nullptr)) return NSP_EINVAL;
  if (g16 && (d % 8 || (reinterpret_cast<uintptr_t>(g16) & 7) || g_p < 0.f || g_p >= 1.f)) return NSP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  int grid = nsp_cdiv(rows, 4 * 8);
  if (grid > 1024) grid = 1024;
  if (grid < 1) grid = 1;
  const size_t shmem = sizeof(float) * 8 * d;
  const int vpl = nsp_cdiv(d, 256);
#define LNPB(V, P) hipLaunchKernelGGL((ln_pair_bwd_kernel<V, P>), dim3(grid), dim3(256), shmem, st, dz, dres, x, gamma_a, beta_a, mean_a, rstd_a, gamma_b, mean_b, rstd_b, dx, dgamma_a, dbeta_a, dgamma_b, dbeta_b, rows, d, reinterpret_cast<__bf16*>(g16), gsum, g_alpha, g_p, g_seed, g_offset)
  if (g16) {
    if (vpl <= 1) LNPB(1, true); else if (vpl <= 2) LNPB(2, true); else if (vpl <= 4) LNPB(4, true); else LNPB(8, true);
  } else {
    if (vpl <= 1) LNPB(1, false); else if (vpl <= 2) LNPB(2, false); else if (vpl <= 4) LNPB(4, false); else LNPB(8, false);
  }
#undef LNPB
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}
