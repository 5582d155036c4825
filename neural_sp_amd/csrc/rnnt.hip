// rnnt.hip -- RNN-Transducer loss on CDNA4.
//
// The reference (rnn_transducer.py:242-256) materialises log_softmax over the
// joint tensor [B,T,U+1,V] and hands it to an external CUDA library
// (warp_rnnt 0.3: rnnt_loss(average_frames=False, reduction='mean',
// gather=False)).  That arithmetic is restated here from Graves (2012):
//   alpha(0,0)=0
//   alpha(t,u)=logaddexp(alpha(t-1,u)+lp_blank(t-1,u), alpha(t,u-1)+lp_label(t,u-1))
//   nll = -(alpha(T-1,U) + lp_blank(T-1,U))
// Only two numbers per lattice node matter, so the path is split in three:
//   1. rnnt_logsoftmax_gather: ONE pass over the logits -> lse, lp_blank,
//      lp_label per node (the dense log-prob tensor is never written);
//   2. rnnt_lattice: alpha and beta anti-diagonal wavefronts, one workgroup
//      per utterance (alpha on the first half, beta on the second half, the
//      previous diagonal in LDS), then node occupancies
//      g_blank/g_label = d nll / d lp_*;
//   3. rnnt_grad_logits: in place, logits <- d loss / d logits
//      = w * ( -(g_b+g_l) * softmax + g_b*[v=blank] + g_l*[v=label] ).
// plus the joint tanh(e_t + g_u) forward and its two backward reductions.
#include "common.h"

namespace {

template <bool VEC>  // VEC: V % 4 == 0 and V <= 1024 -> the row lives in 4 float4 registers per lane, one pass
__global__ __launch_bounds__(256) void rnnt_lsm_gather_kernel(
    const float* __restrict__ logits, const int* __restrict__ labels, const int* __restrict__ elens,
    const int* __restrict__ ylens, float* __restrict__ lse, float* __restrict__ lp_blank,
    float* __restrict__ lp_label, int B, int T, int U1, int V, int blank) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long long nrows = (long long)B * T * U1;
  const int U = U1 - 1;
  for (long long row = (long long)blockIdx.x * 4 + w; row < nrows; row += (long long)gridDim.x * 4) {
    const int u = (int)(row % U1);
    const int t = (int)((row / U1) % T);
    const int b = (int)(row / ((long long)U1 * T));
    if (t >= elens[b] || u > ylens[b]) {
      if (lane == 0) { lse[row] = 0.f; lp_blank[row] = -INFINITY; lp_label[row] = -INFINITY; }
      continue;
    }
    const float* xr = logits + row * V;
    float mx = -FLT_MAX, s = 0.f;
    if (VEC) {
      float4 x[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int v = 4 * (lane + 64 * k);
        x[k] = v < V ? *reinterpret_cast<const float4*>(xr + v) : make_float4(-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX);
        mx = fmaxf(mx, fmaxf(fmaxf(x[k].x, x[k].y), fmaxf(x[k].z, x[k].w)));
      }
      mx = wave_reduce_max(mx);
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (4 * (lane + 64 * k) < V)
          s += __expf(x[k].x - mx) + __expf(x[k].y - mx) + __expf(x[k].z - mx) + __expf(x[k].w - mx);
    } else {
      for (int v = lane; v < V; v += 64) mx = fmaxf(mx, xr[v]);
      mx = wave_reduce_max(mx);
      for (int v = lane; v < V; v += 64) s += __expf(xr[v] - mx);
    }
    s = wave_reduce_sum(s);
    if (lane == 0) {
      const float ls = mx + logf(s);
      lse[row] = ls;
      lp_blank[row] = xr[blank] - ls;
      lp_label[row] = (u < ylens[b] && u < U) ? xr[labels[(long long)b * U + u]] - ls : -INFINITY;
    }
  }
}

__global__ __launch_bounds__(1024) void rnnt_lattice_kernel(
    const float* __restrict__ lp_blank, const float* __restrict__ lp_label,
    const int* __restrict__ elens, const int* __restrict__ ylens, float* __restrict__ alpha,
    float* __restrict__ beta, float* __restrict__ nll, float* __restrict__ g_blank,
    float* __restrict__ g_label, int B, int T, int U1) {
  extern __shared__ __attribute__((aligned(16))) float sh[];  // a[2][U1], b[2][U1]
  float* abuf = sh;
  float* bbuf = sh + 2 * U1;
  const int b = blockIdx.x;
  const int half = blockDim.x >> 1;
  const bool is_beta = threadIdx.x >= half;
  const int tid = is_beta ? threadIdx.x - half : threadIdx.x;
  const int Tb = min(elens[b], T);
  const int Ub = min(ylens[b], U1 - 1);  // number of labels
  const long long base = (long long)b * T * U1;
  const float* lb = lp_blank + base;
  const float* ll = lp_label + base;
  float* al = alpha + base;
  float* be = beta + base;
  if (Tb <= 0) {
    if (threadIdx.x == 0) nll[b] = INFINITY;
    return;
  }
  const int ndiag = Tb + Ub;  // diagonals d = t+u, 0 .. Tb-1+Ub
  for (int d = 0; d < ndiag; ++d) {
    const int cur = d & 1, prv = cur ^ 1;
    if (!is_beta) {
      for (int u = tid; u <= Ub; u += half) {
        const int t = d - u;
        if (t < 0 || t >= Tb) continue;
        float a;
        if (t == 0 && u == 0) {
          a = 0.f;
        } else {
          float x = -INFINITY, y = -INFINITY;
          if (t > 0) x = abuf[prv * U1 + u] + lb[(long long)(t - 1) * U1 + u];
          if (u > 0) y = abuf[prv * U1 + u - 1] + ll[(long long)t * U1 + u - 1];
          a = nsp_logaddexp(x, y);
        }
        abuf[cur * U1 + u] = a;
        al[(long long)t * U1 + u] = a;
      }
    } else {
      // mirrored diagonal: t = Tb-1 - (d - (Ub-u))
      for (int u = tid; u <= Ub; u += half) {
        const int t = Tb - 1 - (d - (Ub - u));
        if (t < 0 || t >= Tb) continue;
        float v;
        if (t == Tb - 1 && u == Ub) {
          v = lb[(long long)t * U1 + u];
        } else {
          float x = -INFINITY, y = -INFINITY;
          if (t + 1 < Tb) x = bbuf[prv * U1 + u] + lb[(long long)t * U1 + u];
          if (u < Ub) y = bbuf[prv * U1 + u + 1] + ll[(long long)t * U1 + u];
          v = nsp_logaddexp(x, y);
        }
        bbuf[cur * U1 + u] = v;
        be[(long long)t * U1 + u] = v;
      }
    }
    __syncthreads();
  }
  // beta(0,0) was produced on the last diagonal by the beta half
  __shared__ float s_nll;
  if (threadIdx.x == half) {
    const float lpz = bbuf[((ndiag - 1) & 1) * U1 + 0];
    s_nll = -lpz;
    nll[b] = -lpz;
  }
  __syncthreads();
  const float nl = s_nll;
  const bool bad = isinf(nl) || isnan(nl);
  // occupancies (visible through global memory after the barrier within this workgroup)
  __threadfence_block();
  for (int i = threadIdx.x; i < T * U1; i += blockDim.x) {
    const int t = i / U1, u = i % U1;
    float gb = 0.f, gl = 0.f;
    if (!bad && t < Tb && u <= Ub) {
      const float a = al[i];
      if (t + 1 < Tb) gb = -expf(a + lb[i] + be[i + U1] + nl);
      else if (u == Ub) gb = -expf(a + lb[i] + nl);
      if (u < Ub) gl = -expf(a + ll[i] + be[i + 1] + nl);
    }
    g_blank[base + i] = gb;
    g_label[base + i] = gl;
  }
}

// Vectorised variant for the bf16 image (V % 4 == 0, ld16 % 4 == 0, ld16 <= 1024): lane owns the
// 4-column groups 4*(lane + 64k), k < 4 -> float4 loads, 8-byte bf16x4 stores, 16 register column
// sums.  (The scalar version issued 4-byte loads and 2-byte stores: 2.6 TB/s.)
__global__ __launch_bounds__(256) void rnnt_grad_logits_vec_kernel(
    const float* __restrict__ logits, const float* __restrict__ lse, const int* __restrict__ labels,
    const float* __restrict__ g_blank, const float* __restrict__ g_label,
    const int* __restrict__ elens, const int* __restrict__ ylens, float wscale_host,
    const float* __restrict__ wscale_dev, int B, int T, int U1, int V, int blank,
    __bf16* __restrict__ out16, int ld16, float* __restrict__ dbias) {
  const float wscale = wscale_dev ? wscale_host * wscale_dev[0] : wscale_host;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long long nrows = (long long)B * T * U1;
  const int U = U1 - 1;
  float csum[4][4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int e = 0; e < 4; ++e) csum[k][e] = 0.f;
  for (long long row = (long long)blockIdx.x * 4 + w; row < nrows; row += (long long)gridDim.x * 4) {
    const int u = (int)(row % U1);
    const int t = (int)((row / U1) % T);
    const int b = (int)(row / ((long long)U1 * T));
    const float* xr = logits + row * V;
    __bf16* o16 = out16 + row * ld16;
    if (t >= elens[b] || u > ylens[b]) {
      bf16x4 z;
#pragma unroll
      for (int e = 0; e < 4; ++e) z[e] = (__bf16)0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int v = 4 * (lane + 64 * k);
        if (v < ld16) *reinterpret_cast<bf16x4*>(o16 + v) = z;
      }
      continue;
    }
    const float ls = lse[row];
    const float gb = g_blank[row], gl = g_label[row];
    const int lab = (u < ylens[b] && u < U) ? labels[(long long)b * U + u] : -1;
    const float gsum = gb + gl;
    float4 x[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int v = 4 * (lane + 64 * k);
      if (v < V) x[k] = *reinterpret_cast<const float4*>(xr + v);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int v = 4 * (lane + 64 * k);
      if (v < ld16) {
        const float xv[4] = {x[k].x, x[k].y, x[k].z, x[k].w};
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float g = 0.f;
          if (v + e < V) {
            g = -gsum * __expf(xv[e] - ls);
            if (v + e == blank) g += gb;
            if (v + e == lab) g += gl;
            g *= wscale;
          }
          o[e] = (__bf16)g;
          csum[k][e] += g;
        }
        *reinterpret_cast<bf16x4*>(o16 + v) = o;
      }
    }
  }
  if (dbias) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int v = 4 * (lane + 64 * k) + e;
        if (v < V && csum[k][e] != 0.f) unsafeAtomicAdd(dbias + v, csum[k][e]);
      }
  }
}

template <int NK>  // lane owns columns lane + 64*k, k < NK (NK*64 >= ncol); NK == 0: no column sums
__global__ __launch_bounds__(256) void rnnt_grad_logits_kernel(
    float* __restrict__ logits, const float* __restrict__ lse, const int* __restrict__ labels,
    const float* __restrict__ g_blank, const float* __restrict__ g_label,
    const int* __restrict__ elens, const int* __restrict__ ylens, float wscale_host,
    const float* __restrict__ wscale_dev, int B, int T, int U1, int V, int blank,
    __bf16* __restrict__ out16, int ld16, float* __restrict__ dbias) {
  // One wave per lattice node (row of V logits).  With NK > 0 every lane also keeps the running
  // column sums of its NK columns in registers (= gradient of the output bias) and flushes them
  // with one atomic per column per wave: no separate pass over the 1.3 GB gradient image.
  const float wscale = wscale_dev ? wscale_host * wscale_dev[0] : wscale_host;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long long nrows = (long long)B * T * U1;
  const int U = U1 - 1;
  const int ncol = out16 ? ld16 : V;
  float csum[NK > 0 ? NK : 1];
#pragma unroll
  for (int k = 0; k < (NK > 0 ? NK : 1); ++k) csum[k] = 0.f;
  for (long long row = (long long)blockIdx.x * 4 + w; row < nrows; row += (long long)gridDim.x * 4) {
    const int u = (int)(row % U1);
    const int t = (int)((row / U1) % T);
    const int b = (int)(row / ((long long)U1 * T));
    float* xr = logits + row * V;
    __bf16* o16 = out16 ? out16 + row * ld16 : nullptr;
    if (t >= elens[b] || u > ylens[b]) {
      if (o16) { for (int v = lane; v < ld16; v += 64) o16[v] = (__bf16)0.f; }
      else { for (int v = lane; v < V; v += 64) xr[v] = 0.f; }
      continue;
    }
    const float ls = lse[row];
    const float gb = g_blank[row], gl = g_label[row];
    const int lab = (u < ylens[b] && u < U) ? labels[(long long)b * U + u] : -1;
    const float gsum = gb + gl;
    if (NK > 0) {
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        const int v = lane + 64 * k;
        if (v < ncol) {
          float g = 0.f;
          if (v < V) {
            g = -gsum * __expf(xr[v] - ls);
            if (v == blank) g += gb;
            if (v == lab) g += gl;
            g *= wscale;
          }
          if (o16) o16[v] = (__bf16)g;
          else xr[v] = g;
          csum[k] += g;
        }
      }
    } else {
      for (int v = lane; v < ncol; v += 64) {
        float g = 0.f;
        if (v < V) {
          g = -gsum * __expf(xr[v] - ls);
          if (v == blank) g += gb;
          if (v == lab) g += gl;
          g *= wscale;
        }
        if (o16) o16[v] = (__bf16)g;   // bf16 image (pitch ld16, zero padded) for the MFMA GEMMs
        else xr[v] = g;
      }
    }
  }
  if (NK > 0 && dbias) {
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const int v = lane + 64 * k;
      if (v < V && csum[k] != 0.f) unsafeAtomicAdd(dbias + v, csum[k]);
    }
  }
}

// h[b,t,u,:] = tanh(e[b,t,:] + g[b,u,:])
__global__ __launch_bounds__(256) void joint_tanh_fwd_kernel(const float* __restrict__ e,
                                                             const float* __restrict__ g,
                                                             float* __restrict__ h, int B, int T,
                                                             int U1, int J, __bf16* __restrict__ h16) {
  const int J4 = J >> 2;
  const long long total = (long long)B * T * U1 * J4;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
    const int j4 = (int)(idx % J4);
    const int u = (int)((idx / J4) % U1);
    const int t = (int)((idx / ((long long)J4 * U1)) % T);
    const long long b = idx / ((long long)J4 * U1 * T);
    const float4 ev = reinterpret_cast<const float4*>(e + (b * T + t) * J)[j4];
    const float4 gv = reinterpret_cast<const float4*>(g + (b * U1 + u) * J)[j4];
    const float4 o = make_float4(nsp_tanh(ev.x + gv.x), nsp_tanh(ev.y + gv.y), nsp_tanh(ev.z + gv.z), nsp_tanh(ev.w + gv.w));
    if (h) reinterpret_cast<float4*>(h)[idx] = o;
    if (h16) {
      bf16x4 q;
      q[0] = (__bf16)o.x; q[1] = (__bf16)o.y; q[2] = (__bf16)o.z; q[3] = (__bf16)o.w;
      reinterpret_cast<bf16x4*>(h16)[idx] = q;
    }
  }
}

// bf16 image only, J % 8 == 0: one workgroup per (b,t); a thread keeps its 8 columns of e[b,t,:] in
// registers and walks u (16-byte stores, no index divisions: the generic kernel above spent its
// time in 64-bit div/mod and 8-byte stores: 1.9 TB/s on a pure write stream)
__global__ __launch_bounds__(256) void joint_tanh_fwd16_kernel(const float* __restrict__ e,
                                                               const float* __restrict__ g, int T, int U1,
                                                               int J, __bf16* __restrict__ h16) {
  const int J8 = J >> 3;
  const long long bt = blockIdx.x;
  const long long b = bt / T;
  const int rows_per_pass = blockDim.x / J8;           // u rows handled per pass
  const int c = threadIdx.x % J8, ur = threadIdx.x / J8;
  if (ur >= rows_per_pass) return;
  const float4 e0 = reinterpret_cast<const float4*>(e + bt * J)[2 * c];
  const float4 e1 = reinterpret_cast<const float4*>(e + bt * J)[2 * c + 1];
  for (int u = ur; u < U1; u += rows_per_pass) {
    const float4* gp = reinterpret_cast<const float4*>(g + (b * U1 + u) * J) + 2 * c;
    const float4 g0 = gp[0], g1 = gp[1];
    bf16x8 q;
    q[0] = (__bf16)nsp_tanh(e0.x + g0.x); q[1] = (__bf16)nsp_tanh(e0.y + g0.y);
    q[2] = (__bf16)nsp_tanh(e0.z + g0.z); q[3] = (__bf16)nsp_tanh(e0.w + g0.w);
    q[4] = (__bf16)nsp_tanh(e1.x + g1.x); q[5] = (__bf16)nsp_tanh(e1.y + g1.y);
    q[6] = (__bf16)nsp_tanh(e1.z + g1.z); q[7] = (__bf16)nsp_tanh(e1.w + g1.w);
    *reinterpret_cast<bf16x8*>(h16 + ((bt * U1 + u) * (long long)J) + 8 * c) = q;
  }
}

// dz = dh * (1 - h^2) written in place over dh; de[b,t,:] = sum_u dz
__global__ __launch_bounds__(256) void joint_tanh_bwd_de_kernel(const float* __restrict__ h,
                                                                const __bf16* __restrict__ h16,
                                                                float* __restrict__ dh,
                                                                float* __restrict__ de, int B, int T,
                                                                int U1, int J) {
  const int J4 = J >> 2;
  const long long bt = blockIdx.x;  // one block per (b,t)
  for (int j4 = threadIdx.x; j4 < J4; j4 += blockDim.x) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int u = 0; u < U1; ++u) {
      const long long o = (bt * U1 + u) * J4 + j4;
      float4 hv;
      if (h16) {
        const bf16x4 q = reinterpret_cast<const bf16x4*>(h16)[o];
        hv = make_float4((float)q[0], (float)q[1], (float)q[2], (float)q[3]);
      } else {
        hv = reinterpret_cast<const float4*>(h)[o];
      }
      float4 dv = reinterpret_cast<float4*>(dh)[o];
      dv.x *= (1.f - hv.x * hv.x); dv.y *= (1.f - hv.y * hv.y);
      dv.z *= (1.f - hv.z * hv.z); dv.w *= (1.f - hv.w * hv.w);
      reinterpret_cast<float4*>(dh)[o] = dv;
      acc.x += dv.x; acc.y += dv.y; acc.z += dv.z; acc.w += dv.w;
    }
    reinterpret_cast<float4*>(de)[bt * J4 + j4] = acc;
  }
}

// dg[b,u,:] = sum_t dz[b,t,u,:]
__global__ __launch_bounds__(256) void joint_tanh_bwd_dg_kernel(const float* __restrict__ dz,
                                                                float* __restrict__ dg, int B, int T,
                                                                int U1, int J) {
  const int J4 = J >> 2;
  const long long bu = blockIdx.x;  // one block per (b,u)
  const long long b = bu / U1;
  const int u = (int)(bu % U1);
  for (int j4 = threadIdx.x; j4 < J4; j4 += blockDim.x) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = 0; t < T; ++t) {
      const float4 dv = reinterpret_cast<const float4*>(dz)[((b * T + t) * U1 + u) * J4 + j4];
      acc.x += dv.x; acc.y += dv.y; acc.z += dv.z; acc.w += dv.w;
    }
    reinterpret_cast<float4*>(dg)[bu * J4 + j4] = acc;
  }
}

// One pass over dz (bf16 [B,T,U1,J]) for both joint-input gradients.  grid: (J/32, B); block 256:
// thread = (4 columns of the workgroup's 32-column slice, one of 32 u-lanes).  A thread meets the
// same (u, columns) for every t, so sum_t lives in registers (<= 8 x 4 fp32 for U1 <= 256, LDS
// spill beyond); sum_u is reduced across the u-lanes once per t.
template <int KU>
__global__ __launch_bounds__(256) void joint_dz_reduce_kernel(const __bf16* __restrict__ dz,
                                                              float* __restrict__ de, float* __restrict__ dg,
                                                              int B, int T, int U1, int J, int Tc) {
  __shared__ float red[4][8][4];
  const int cg = threadIdx.x & 7, ul = threadIdx.x >> 3;
  const int wave = threadIdx.x >> 6;
  const int c0 = blockIdx.x * 32 + cg * 4;
  const long long b = blockIdx.y;
  float4 accg[KU];
#pragma unroll
  for (int k = 0; k < KU; ++k) accg[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  // blockIdx.z owns the time steps [z*Tc, (z+1)*Tc): its sum_t goes to slab z of dg (summed by
  // nsp_splitk_reduce afterwards); 8 workgroups per CU hide each other's barriers and latencies
  const int t_end = min(T, (int)(blockIdx.z + 1) * Tc);
  dg += (long long)blockIdx.z * B * U1 * J;
  for (int t = blockIdx.z * Tc; t < t_end; ++t) {
    const __bf16* base = dz + ((b * T + t) * U1) * J + c0;
    bf16x4 v[KU];
#pragma unroll
    for (int k = 0; k < KU; ++k) {
      const int u = ul + 32 * k;
      if (u < U1) v[k] = *reinterpret_cast<const bf16x4*>(base + (long long)u * J);
    }
    float4 acce = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < KU; ++k) {
      const int u = ul + 32 * k;
      if (u < U1) {
        const float x0 = (float)v[k][0], x1 = (float)v[k][1], x2 = (float)v[k][2], x3 = (float)v[k][3];
        accg[k].x += x0; accg[k].y += x1; accg[k].z += x2; accg[k].w += x3;
        acce.x += x0; acce.y += x1; acce.z += x2; acce.w += x3;
      }
    }
    // u-lanes of one wave: lane bits 3..5
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) {
      acce.x += __shfl_xor(acce.x, o, 64); acce.y += __shfl_xor(acce.y, o, 64);
      acce.z += __shfl_xor(acce.z, o, 64); acce.w += __shfl_xor(acce.w, o, 64);
    }
    __syncthreads();
    if ((threadIdx.x & 63) < 8) {
      red[wave][cg][0] = acce.x; red[wave][cg][1] = acce.y; red[wave][cg][2] = acce.z; red[wave][cg][3] = acce.w;
    }
    __syncthreads();
    if (threadIdx.x < 32) {
      const int cc = threadIdx.x >> 2, e = threadIdx.x & 3;
      de[(b * T + t) * J + blockIdx.x * 32 + cc * 4 + e] = red[0][cc][e] + red[1][cc][e] + red[2][cc][e] + red[3][cc][e];
    }
  }
#pragma unroll
  for (int k = 0; k < KU; ++k) {
    const int u = ul + 32 * k;
    if (u < U1) *reinterpret_cast<float4*>(dg + (b * U1 + u) * J + c0) = accg[k];
  }
}

}  // namespace

extern "C" int nsp_rnnt_joint_dz_reduce(const void* dz16, float* de, float* dg_slabs, int nslab, int B, int T,
                                        int U1, int J, void* stream) {
  if (J % 32 || U1 > 512 || B <= 0 || T <= 0 || U1 <= 0 || nslab < 1) return NSP_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const __bf16* z = reinterpret_cast<const __bf16*>(dz16);
  const int Tc = nsp_cdiv(T, nslab);
  dim3 grid(J / 32, B, nslab);
  const int ku = nsp_cdiv(U1, 32);
  if (ku <= 4) hipLaunchKernelGGL((joint_dz_reduce_kernel<4>), grid, dim3(256), 0, st, z, de, dg_slabs, B, T, U1, J, Tc);
  else if (ku <= 8) hipLaunchKernelGGL((joint_dz_reduce_kernel<8>), grid, dim3(256), 0, st, z, de, dg_slabs, B, T, U1, J, Tc);
  else hipLaunchKernelGGL((joint_dz_reduce_kernel<16>), grid, dim3(256), 0, st, z, de, dg_slabs, B, T, U1, J, Tc);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

extern "C" int nsp_rnnt_logsoftmax_gather(const float* logits, const int* labels, const int* elens,
                                          const int* ylens, float* lse, float* lp_blank,
                                          float* lp_label, int B, int T, int U1, int V, int blank,
                                          void* stream) {
  if (B <= 0 || T <= 0 || U1 <= 0 || V <= 1) return NSP_EINVAL;
  int grid = nsp_cdiv((long long)B * T * U1, 4);
  if (grid > 256 * 32) grid = 256 * 32;
  if (V % 4 == 0 && V <= 1024)
    hipLaunchKernelGGL((rnnt_lsm_gather_kernel<true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, logits,
                       labels, elens, ylens, lse, lp_blank, lp_label, B, T, U1, V, blank);
  else
    hipLaunchKernelGGL((rnnt_lsm_gather_kernel<false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, logits,
                       labels, elens, ylens, lse, lp_blank, lp_label, B, T, U1, V, blank);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

extern "C" int nsp_rnnt_lattice(const float* lp_blank, const float* lp_label, const int* elens,
                                const int* ylens, float* alpha, float* beta, float* nll,
                                float* g_blank, float* g_label, int B, int T, int U1, void* stream) {
  if (B <= 0 || T <= 0 || U1 <= 0) return NSP_EINVAL;
  const size_t sh = sizeof(float) * 4 * U1;
  if (sh > 150 * 1024) return NSP_EUNSUPPORTED;
  int half = ((U1 + 63) / 64) * 64;
  if (half > 512) half = 512;
  if (sh > 64 * 1024)
    hipFuncSetAttribute((const void*)rnnt_lattice_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
  hipLaunchKernelGGL(rnnt_lattice_kernel, dim3(B), dim3(2 * half), sh, (hipStream_t)stream, lp_blank,
                     lp_label, elens, ylens, alpha, beta, nll, g_blank, g_label, B, T, U1);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

extern "C" int nsp_rnnt_grad_logits(float* logits, const float* lse, const int* labels,
                                    const float* g_blank, const float* g_label, const int* elens,
                                    const int* ylens, float wscale, const float* wscale_dev, int B,
                                    int T, int U1, int V, int blank, void* out16, int ld16,
                                    float* dbias, void* stream) {
  if (out16 && ld16 < V) return NSP_EINVAL;
  int grid = nsp_cdiv((long long)B * T * U1, 4);
  const int ncol = out16 ? ld16 : V;
  const bool sums = dbias && ncol <= 16 * 64;
  const int cap = sums ? 256 * 4 : 256 * 32;  // fewer, longer-lived waves when column sums are kept
  if (grid > cap) grid = cap;
  __bf16* o16 = reinterpret_cast<__bf16*>(out16);
  hipStream_t st = (hipStream_t)stream;
  if (out16 && V % 4 == 0 && ld16 % 4 == 0 && ld16 <= 1024) {
    if (grid > 256 * 4) grid = 256 * 4;
    hipLaunchKernelGGL(rnnt_grad_logits_vec_kernel, dim3(grid), dim3(256), 0, st, logits, lse, labels, g_blank,
                       g_label, elens, ylens, wscale, wscale_dev, B, T, U1, V, blank, o16, ld16, dbias);
  } else if (sums)
    hipLaunchKernelGGL((rnnt_grad_logits_kernel<16>), dim3(grid), dim3(256), 0, st, logits, lse, labels,
                       g_blank, g_label, elens, ylens, wscale, wscale_dev, B, T, U1, V, blank, o16, ld16, dbias);
  else
    hipLaunchKernelGGL((rnnt_grad_logits_kernel<0>), dim3(grid), dim3(256), 0, st, logits, lse, labels,
                       g_blank, g_label, elens, ylens, wscale, wscale_dev, B, T, U1, V, blank, o16, ld16,
                       (float*)nullptr);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

extern "C" int nsp_rnnt_joint_tanh_fwd(const float* e, const float* g, float* h, void* h16, int B,
                                       int T, int U1, int J, void* stream) {
  if (J % 4) return NSP_EUNSUPPORTED;
  if (!h && !h16) return NSP_EINVAL;
  if (!h && h16 && J % 8 == 0 && J / 8 <= 256 && (reinterpret_cast<uintptr_t>(h16) & 15) == 0) {
    hipLaunchKernelGGL(joint_tanh_fwd16_kernel, dim3(B * T), dim3(256), 0, (hipStream_t)stream, e, g, T, U1, J,
                       reinterpret_cast<__bf16*>(h16));
    NSP_LAUNCH_CHECK();
    return NSP_OK;
  }
  long long n = (long long)B * T * U1 * (J / 4);
  long long gr = (n + 255) / 256;
  if (gr > 256 * 32) gr = 256 * 32;
  hipLaunchKernelGGL(joint_tanh_fwd_kernel, dim3((int)gr), dim3(256), 0, (hipStream_t)stream, e, g, h,
                     B, T, U1, J, reinterpret_cast<__bf16*>(h16));
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

extern "C" int nsp_rnnt_joint_tanh_bwd(const float* h, const void* h16, float* dh, float* de,
                                       float* dg, int B, int T, int U1, int J, void* stream) {
  if (J % 4) return NSP_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(joint_tanh_bwd_de_kernel, dim3(B * T), dim3(128), 0, st, h,
                     reinterpret_cast<const __bf16*>(h16), dh, de, B, T, U1, J);
  hipLaunchKernelGGL(joint_tanh_bwd_dg_kernel, dim3(B * U1), dim3(128), 0, st, dh, dg, B, T, U1, J);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}
