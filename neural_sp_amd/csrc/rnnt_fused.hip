// rnnt_fused.hip -- RNN-Transducer joint + loss without the [B,T,U+1,V] tensor (bf16 throughput mode).
//
// The reference materialises logits = output(tanh(w_enc(e)[:, :, None] + w_dec(g)[:, None])) and its
// log_softmax, both [B,T,U+1,V] fp32 (rnn_transducer.py:239-242, 262-276): 160 MB per utterance at
// T''=200, U=200, V=1000.  The lattice needs three numbers per node (lse, logit_blank, logit_label),
// its gradient is rank-structured:  d loss / d logits = -(g_b + g_l) softmax + g_b [blank] + g_l [label].
// So:
//   forward : h = tanh(e_t + g_u) (bf16, J wide) -> logit GEMM whose epilogue keeps per-row
//             (max, sum exp) partials and the two gathered logits (NSP_EPI_RNNT_LSE, gemm_bf16.hip);
//             merge -> lattice (alpha/beta wavefronts, occupancies);
//   backward: the SAME GEMM again, its epilogue turning the recomputed logit tile into the bf16
//             gradient image (NSP_EPI_RNNT_DLOGITS) that the weight-gradient / data-gradient GEMMs
//             consume, then one pass over dz for both joint-input gradients.
// Lattice nodes are COMPACTED: utterance b owns rows roff[b] .. roff[b+1] as a dense [T_b][U_b+1]
// grid; the ~30 % padded nodes of a [B, T_max, U_max+1] layout are never computed.
#include "common.h"

int nsp_gemm_bf16_launch(const nsp_gemm_params& p, hipStream_t st);  // gemm_bf16.hip

namespace {

// one workgroup per (b,t) with t < T_b; a thread keeps its 8 columns of e[b,t,:] in registers and walks u
__global__ __launch_bounds__(256) void joint_tanh_compact_kernel(
    const float* __restrict__ e, const float* __restrict__ g, const int* __restrict__ labels,
    const int* __restrict__ elens, const int* __restrict__ ylens, const long long* __restrict__ roff,
    __bf16* __restrict__ h16, int* __restrict__ lab, int T, int U1, int J) {
  const int J8 = J >> 3;
  const int b = blockIdx.x / T, t = blockIdx.x % T;
  const int Tb = min(elens[b], T), Ub = min(ylens[b], U1 - 1);
  if (t >= Tb) return;
  const int rows_per_pass = blockDim.x / J8;
  const int c = threadIdx.x % J8, ur = threadIdx.x / J8;
  if (ur >= rows_per_pass) return;
  const long long bt = (long long)b * T + t;
  const float4 e0 = reinterpret_cast<const float4*>(e + bt * J)[2 * c];
  const float4 e1 = reinterpret_cast<const float4*>(e + bt * J)[2 * c + 1];
  const long long row0 = roff[b] + (long long)t * (Ub + 1);
  for (int u = ur; u <= Ub; u += rows_per_pass) {
    const float4* gp = reinterpret_cast<const float4*>(g + ((long long)b * U1 + u) * J) + 2 * c;
    const float4 g0 = gp[0], g1 = gp[1];
    bf16x8 q;
    q[0] = (__bf16)nsp_tanh(e0.x + g0.x); q[1] = (__bf16)nsp_tanh(e0.y + g0.y);
    q[2] = (__bf16)nsp_tanh(e0.z + g0.z); q[3] = (__bf16)nsp_tanh(e0.w + g0.w);
    q[4] = (__bf16)nsp_tanh(e1.x + g1.x); q[5] = (__bf16)nsp_tanh(e1.y + g1.y);
    q[6] = (__bf16)nsp_tanh(e1.z + g1.z); q[7] = (__bf16)nsp_tanh(e1.w + g1.w);
    // (non-temporal: the 3.7-GB image is next read by the joint kernel long after it has left every L2)
    __builtin_nontemporal_store(q, reinterpret_cast<bf16x8*>(h16 + (row0 + u) * J + 8 * c));
    if (c == 0) lab[row0 + u] = u < Ub ? labels[(long long)b * (U1 - 1) + u] : -1;
  }
}

// 16 lanes per lattice node: lane l holds the (max, sum) partials l, l + 16, ... of its node, so that a wave
// reads 4 nodes x npart partials as one contiguous run (one thread per node read 8 B at a 128-B stride, twice:
// 0.37 TB/s), then the 16 (max, sum) pairs are merged with xor-shuffles.
__global__ __launch_bounds__(256) void lse_merge_kernel(const float* __restrict__ part, int npart,
                                                        float* __restrict__ lse, float* __restrict__ rb,
                                                        float* __restrict__ rl, const int* __restrict__ lab,
                                                        long long M) {
  const int lane = threadIdx.x & 63, sub = lane & 15, grp = lane >> 4;
  const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
  for (long long m0 = wave * 4; m0 < M; m0 += nwaves * 4) {
    const long long m = m0 + grp;
    const bool ok = m < M;
    const float2* pp = reinterpret_cast<const float2*>(part) + (ok ? m : M - 1) * npart;
    float mx = -FLT_MAX, s = 0.f;
    for (int i = sub; i < npart; i += 16) {
      const float2 p = pp[i];
      const float nm = fmaxf(mx, p.x);
      s = s * __expf(mx - nm) + p.y * __expf(p.x - nm);
      mx = nm;
    }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
      const float om = __shfl_xor(mx, o, 64), os = __shfl_xor(s, o, 64);
      const float nm = fmaxf(mx, om);
      s = s * __expf(mx - nm) + os * __expf(om - nm);
      mx = nm;
    }
    if (ok && sub == 0) {
      const float ls = mx + logf(s);
      lse[m] = ls;
      rb[m] = rb[m] - ls;
      rl[m] = lab[m] >= 0 ? rl[m] - ls : -INFINITY;
    }
  }
}

// alpha on the first half of the workgroup, beta on the second, previous anti-diagonal in LDS
// (same recursion as rnnt_lattice_kernel in rnnt.hip; here the lattice of utterance b is the dense
// [T_b][U_b+1] grid at row offset roff[b])
__global__ __launch_bounds__(1024) void rnnt_lattice_compact_kernel(
    const float* __restrict__ lp_blank, const float* __restrict__ lp_label, const int* __restrict__ elens,
    const int* __restrict__ ylens, const long long* __restrict__ roff, float* __restrict__ alpha,
    float* __restrict__ beta, float* __restrict__ nll, float* __restrict__ g_blank,
    float* __restrict__ g_label, int U1max) {
  extern __shared__ __attribute__((aligned(16))) float sh[];  // a[2][U1max], b[2][U1max]
  float* abuf = sh;
  float* bbuf = sh + 2 * U1max;
  const int b = blockIdx.x;
  const int half = blockDim.x >> 1;
  const bool is_beta = threadIdx.x >= half;
  const int tid = is_beta ? threadIdx.x - half : threadIdx.x;
  const long long base = roff[b];
  const int Ub = min(ylens[b], U1max - 1);
  const int U1 = Ub + 1;
  const int Tb = (int)((roff[b + 1] - base) / U1);   // == min(elens[b], T)
  (void)elens;
  const float* lb = lp_blank + base;
  const float* ll = lp_label + base;
  float* al = alpha + base;
  float* be = beta + base;
  if (Tb <= 0) {
    if (threadIdx.x == 0) nll[b] = INFINITY;
    return;
  }
  const int ndiag = Tb + Ub;
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
          if (t > 0) x = abuf[prv * U1max + u] + lb[(long long)(t - 1) * U1 + u];
          if (u > 0) y = abuf[prv * U1max + u - 1] + ll[(long long)t * U1 + u - 1];
          a = nsp_logaddexp(x, y);
        }
        abuf[cur * U1max + u] = a;
        al[(long long)t * U1 + u] = a;
      }
    } else {
      for (int u = tid; u <= Ub; u += half) {
        const int t = Tb - 1 - (d - (Ub - u));
        if (t < 0 || t >= Tb) continue;
        float v;
        if (t == Tb - 1 && u == Ub) {
          v = lb[(long long)t * U1 + u];
        } else {
          float x = -INFINITY, y = -INFINITY;
          if (t + 1 < Tb) x = bbuf[prv * U1max + u] + lb[(long long)t * U1 + u];
          if (u < Ub) y = bbuf[prv * U1max + u + 1] + ll[(long long)t * U1 + u];
          v = nsp_logaddexp(x, y);
        }
        bbuf[cur * U1max + u] = v;
        be[(long long)t * U1 + u] = v;
      }
    }
    __syncthreads();
  }
  __shared__ float s_nll;
  if (threadIdx.x == half) {
    const float lpz = bbuf[((ndiag - 1) & 1) * U1max + 0];
    s_nll = -lpz;
    nll[b] = -lpz;
  }
  __syncthreads();
  const float nl = s_nll;
  const bool bad = isinf(nl) || isnan(nl);
  __threadfence_block();
  const int nn = Tb * U1;
  for (int i = threadIdx.x; i < nn; i += blockDim.x) {
    const int t = i / U1, u = i - t * U1;
    float gb = 0.f, gl = 0.f;
    if (!bad) {
      const float a = al[i];
      if (t + 1 < Tb) gb = -expf(a + lb[i] + be[i + U1] + nl);
      else if (u == Ub) gb = -expf(a + lb[i] + nl);
      if (u < Ub) gl = -expf(a + ll[i] + be[i + 1] + nl);
    }
    g_blank[base + i] = gb;
    g_label[base + i] = gl;
  }
}

// One pass over dz (bf16 [M,J], compact rows) for both joint-input gradients.  grid: (J/64, B, nslab);
// block 256: thread = (8 columns of the workgroup's 64-column slice, one of 32 u-lanes).  A thread meets
// the same (u, columns) for every t, so sum_t lives in registers; sum_u is reduced across the u-lanes
// once per t.  Padded positions (t >= T_b, u > U_b) are written as zeros.
// Round 6: 16-B loads and 128 contiguous bytes per row and workgroup (was 8 B per lane in 64-B segments: half of every
// line fetched by each of two workgroups), and the cross-wave part of sum_u batched over TB frames per pair of barriers
// (was two barriers per frame): 1.26 ms per step at 2.9 TB/s before.
template <int KU>
__global__ __launch_bounds__(256) void joint_dz_reduce_compact_kernel(
    const __bf16* __restrict__ dz, const int* __restrict__ elens, const int* __restrict__ ylens,
    const long long* __restrict__ roff, float* __restrict__ de, float* __restrict__ dg, int B, int T, int U1,
    int J, int Tc) {
  constexpr int TB = 8;
  __shared__ float red[TB][4][64];
  const int cg = threadIdx.x & 7, ul = threadIdx.x >> 3;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c0 = blockIdx.x * 64 + cg * 8;
  const long long b = blockIdx.y;
  const int Ub = min(ylens[b], U1 - 1), U1b = Ub + 1;
  const int Tb = min(elens[b], T);
  const long long base = roff[b];
  float accg[KU][8];
#pragma unroll
  for (int k = 0; k < KU; ++k)
#pragma unroll
    for (int e = 0; e < 8; ++e) accg[k][e] = 0.f;
  const int t_beg = blockIdx.z * Tc, t_end = min(T, (int)(blockIdx.z + 1) * Tc);
  dg += (long long)blockIdx.z * B * U1 * J;
  for (int tb = t_beg; tb < t_end; tb += TB) {
#pragma unroll 1
    for (int tt = 0; tt < TB; ++tt) {
      const int t = tb + tt;
      float acce[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (t < Tb && t < t_end) {
        const __bf16* rowp = dz + (base + (long long)t * U1b) * J + c0;
        bf16x8 v[KU];
#pragma unroll
        for (int k = 0; k < KU; ++k) {
          const int u = ul + 32 * k;
          v[k] = *reinterpret_cast<const bf16x8*>(rowp + (long long)min(u, U1b - 1) * J);      // (clamped: unconditional loads)
        }
#pragma unroll
        for (int k = 0; k < KU; ++k) {
          const int u = ul + 32 * k;
          if (u < U1b) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float xv = (float)v[k][e]; accg[k][e] += xv; acce[e] += xv; }
          }
        }
#pragma unroll
        for (int o = 8; o < 64; o <<= 1)
#pragma unroll
          for (int e = 0; e < 8; ++e) acce[e] += __shfl_xor(acce[e], o, 64);
      }
      if (lane < 8) {
        *reinterpret_cast<float4*>(&red[tt][wave][cg * 8]) = make_float4(acce[0], acce[1], acce[2], acce[3]);
        *reinterpret_cast<float4*>(&red[tt][wave][cg * 8 + 4]) = make_float4(acce[4], acce[5], acce[6], acce[7]);
      }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < TB * 64; idx += 256) {
      const int tt = idx >> 6, c = idx & 63;
      const int t = tb + tt;
      if (t < t_end) de[(b * T + t) * J + blockIdx.x * 64 + c] = red[tt][0][c] + red[tt][1][c] + red[tt][2][c] + red[tt][3][c];
    }
    __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < KU; ++k) {
    const int u = ul + 32 * k;
    if (u < U1) {
      float* o = dg + (b * U1 + u) * J + c0;
      *reinterpret_cast<float4*>(o) = make_float4(accg[k][0], accg[k][1], accg[k][2], accg[k][3]);
      *reinterpret_cast<float4*>(o + 4) = make_float4(accg[k][4], accg[k][5], accg[k][6], accg[k][7]);
    }
  }
}

// (the 32-column form, kept for joint widths that are not a multiple of 64)
// One pass over dz (bf16 [M,J], compact rows) for both joint-input gradients.  grid: (J/32, B, nslab);
// block 256: thread = (4 columns of the workgroup's 32-column slice, one of 32 u-lanes).  A thread meets
// the same (u, columns) for every t, so sum_t lives in registers; sum_u is reduced across the u-lanes
// once per t.  Padded positions (t >= T_b, u > U_b) are written as zeros.
template <int KU>
__global__ __launch_bounds__(256) void joint_dz_reduce_compact32_kernel(
    const __bf16* __restrict__ dz, const int* __restrict__ elens, const int* __restrict__ ylens,
    const long long* __restrict__ roff, float* __restrict__ de, float* __restrict__ dg, int B, int T, int U1,
    int J, int Tc) {
  __shared__ float red[4][8][4];
  const int cg = threadIdx.x & 7, ul = threadIdx.x >> 3;
  const int wave = threadIdx.x >> 6;
  const int c0 = blockIdx.x * 32 + cg * 4;
  const long long b = blockIdx.y;
  const int Ub = min(ylens[b], U1 - 1), U1b = Ub + 1;
  const int Tb = min(elens[b], T);
  const long long base = roff[b];
  float4 accg[KU];
#pragma unroll
  for (int k = 0; k < KU; ++k) accg[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int t_beg = blockIdx.z * Tc, t_end = min(T, (int)(blockIdx.z + 1) * Tc);
  dg += (long long)blockIdx.z * B * U1 * J;
  for (int t = t_beg; t < t_end; ++t) {
    float4 acce = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t < Tb) {
      const __bf16* rowp = dz + (base + (long long)t * U1b) * J + c0;
      bf16x4 v[KU];
#pragma unroll
      for (int k = 0; k < KU; ++k) {
        const int u = ul + 32 * k;
        if (u < U1b) v[k] = *reinterpret_cast<const bf16x4*>(rowp + (long long)u * J);
      }
#pragma unroll
      for (int k = 0; k < KU; ++k) {
        const int u = ul + 32 * k;
        if (u < U1b) {
          const float x0 = (float)v[k][0], x1 = (float)v[k][1], x2 = (float)v[k][2], x3 = (float)v[k][3];
          accg[k].x += x0; accg[k].y += x1; accg[k].z += x2; accg[k].w += x3;
          acce.x += x0; acce.y += x1; acce.z += x2; acce.w += x3;
        }
      }
#pragma unroll
      for (int o = 8; o < 64; o <<= 1) {
        acce.x += __shfl_xor(acce.x, o, 64); acce.y += __shfl_xor(acce.y, o, 64);
        acce.z += __shfl_xor(acce.z, o, 64); acce.w += __shfl_xor(acce.w, o, 64);
      }
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

extern "C" int nsp_rnnt_joint_tanh_compact(const float* e, const float* g, const int* labels, const int* elens,
                                           const int* ylens, const long long* roff, void* h16, int* lab, int B,
                                           int T, int U1, int J, void* stream) {
  if (J % 8 || J / 8 > 256 || (reinterpret_cast<uintptr_t>(h16) & 15)) return NSP_EUNSUPPORTED;
  if (B <= 0 || T <= 0 || U1 <= 0) return NSP_EINVAL;
  hipLaunchKernelGGL(joint_tanh_compact_kernel, dim3(B * T), dim3(256), 0, (hipStream_t)stream, e, g, labels, elens,
                     ylens, roff, reinterpret_cast<__bf16*>(h16), lab, T, U1, J);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

// one 16-B record per lattice node for the DLOGITS epilogue: {lse, g_blank * s, g_label * s, bits(label)}
__global__ __launch_bounds__(256) void rnnt_pack_rows_kernel(const float* __restrict__ lse, const float* __restrict__ gb,
                                                             const float* __restrict__ gl, const int* __restrict__ lab,
                                                             float scale, const float* __restrict__ scale_dev,
                                                             float4* __restrict__ rec, long long M) {
  const float s = scale * (scale_dev ? scale_dev[0] : 1.f);
  for (long long m = (long long)blockIdx.x * 256 + threadIdx.x; m < M; m += (long long)gridDim.x * 256)
    rec[m] = make_float4(lse[m], gb[m] * s, gl[m] * s, __int_as_float(lab[m]));
}

extern "C" int nsp_rnnt_joint_gemm(int epi_mode, const void* h16, const void* w16, const float* bias, long long M,
                                   int V, int Vp, int J, int blank, const int* lab, float* f0, float* f1,
                                   float* f2, float* f3, void* d16, float scale, const float* scale_dev,
                                   float* rec, void* stream) {
  if (M <= 0) return NSP_OK;
  if (M > 0x7fffffffLL || Vp % 64 || V > Vp || V < 1 || J % 8 || !h16 || !w16 || !lab || !f0 || !f1 || !f2)
    return NSP_EINVAL;
  if (epi_mode != NSP_EPI_RNNT_LSE && epi_mode != NSP_EPI_RNNT_DLOGITS) return NSP_EINVAL;
  if (epi_mode == NSP_EPI_RNNT_DLOGITS && (!d16 || !rec || (reinterpret_cast<uintptr_t>(rec) & 15))) return NSP_EINVAL;
  if (epi_mode == NSP_EPI_RNNT_DLOGITS) {
    long long g = (M + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    hipLaunchKernelGGL(rnnt_pack_rows_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, f0, f1, f2, lab, scale,
                       scale_dev, reinterpret_cast<float4*>(rec), M);
    NSP_LAUNCH_CHECK();
    f0 = rec;
  }
  nsp_gemm_params p;
  p.M = (int)M; p.N = Vp; p.K = J;
  p.A = h16; p.a_rs = J; p.a_cs = 1;
  p.B = w16; p.b_ks = 1; p.b_ns = J;
  p.C = d16; p.ldc = Vp;
  p.batch1 = p.batch2 = 1;
  p.a_b1 = p.a_b2 = p.b_b1 = p.b_b2 = p.c_b1 = p.c_b2 = 0;
  p.bias = bias; p.act = NSP_ACT_NONE; p.pre_out = nullptr; p.dact_src = nullptr; p.dact = NSP_ACT_NONE;
  p.res = nullptr; p.alpha = 1.f; p.splitk = 1; p.mode = NSP_COMPUTE_BF16; p.dropout_p = 0.f;
  p.seed = p.offset = 0ull;
  p.a_dtype = p.b_dtype = NSP_DT_BF16; p.c_dtype = NSP_DT_BF16; p.pre_dtype = p.dact_dtype = NSP_DT_F32;
  p.c_ss = 0;
  p.epi_mode = epi_mode; p.epi_ncols = V; p.epi_blank = blank; p.epi_lab = lab;
  p.epi_f0 = f0; p.epi_f1 = f1; p.epi_f2 = f2; p.epi_f3 = f3;
  p.epi_scale_dev = scale_dev; p.epi_scale = scale;
  return nsp_gemm_bf16_launch(p, (hipStream_t)stream);
}

extern "C" int nsp_rnnt_lse_merge(const float* part, int npart, float* lse, float* rb, float* rl, const int* lab,
                                  long long M, void* stream) {
  if (M <= 0) return NSP_OK;
  if (npart < 1) return NSP_EINVAL;
  long long g = (M + 15) / 16;             // 16 nodes per 256-thread workgroup per round
  if (g > 256 * 16) g = 256 * 16;
  hipLaunchKernelGGL(lse_merge_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, part, npart, lse, rb, rl, lab, M);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

extern "C" int nsp_rnnt_lattice_compact(const float* lp_blank, const float* lp_label, const int* elens,
                                        const int* ylens, const long long* roff, float* alpha, float* beta,
                                        float* nll, float* g_blank, float* g_label, int B, int U1max,
                                        void* stream) {
  if (B <= 0 || U1max <= 0) return NSP_EINVAL;
  const size_t sh = sizeof(float) * 4 * U1max;
  if (sh > 150 * 1024) return NSP_EUNSUPPORTED;
  int half = ((U1max + 63) / 64) * 64;
  if (half > 512) half = 512;
  if (sh > 64 * 1024)
    hipFuncSetAttribute((const void*)rnnt_lattice_compact_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
  hipLaunchKernelGGL(rnnt_lattice_compact_kernel, dim3(B), dim3(2 * half), sh, (hipStream_t)stream, lp_blank, lp_label,
                     elens, ylens, roff, alpha, beta, nll, g_blank, g_label, U1max);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

extern "C" int nsp_rnnt_joint_dz_reduce_compact(const void* dz16, const int* elens, const int* ylens,
                                                const long long* roff, float* de, float* dg_slabs, int nslab,
                                                int B, int T, int U1, int J, void* stream) {
  if (J % 32 || U1 > 512 || B <= 0 || T <= 0 || U1 <= 0 || nslab < 1) return NSP_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const __bf16* z = reinterpret_cast<const __bf16*>(dz16);
  const int Tc = nsp_cdiv(T, nslab);
  const int ku = nsp_cdiv(U1, 32);
  if (J % 64 || (reinterpret_cast<uintptr_t>(dz16) & 15)) {
    dim3 grid32(J / 32, B, nslab);
    if (ku <= 4)
      hipLaunchKernelGGL((joint_dz_reduce_compact32_kernel<4>), grid32, dim3(256), 0, st, z, elens, ylens, roff, de, dg_slabs, B, T, U1, J, Tc);
    else if (ku <= 8)
      hipLaunchKernelGGL((joint_dz_reduce_compact32_kernel<8>), grid32, dim3(256), 0, st, z, elens, ylens, roff, de, dg_slabs, B, T, U1, J, Tc);
    else
      hipLaunchKernelGGL((joint_dz_reduce_compact32_kernel<16>), grid32, dim3(256), 0, st, z, elens, ylens, roff, de, dg_slabs, B, T, U1, J, Tc);
    NSP_LAUNCH_CHECK();
    return NSP_OK;
  }
  dim3 grid(J / 64, B, nslab);
  if (ku <= 4)
    hipLaunchKernelGGL((joint_dz_reduce_compact_kernel<4>), grid, dim3(256), 0, st, z, elens, ylens, roff, de, dg_slabs, B, T, U1, J, Tc);
  else if (ku <= 8)
    hipLaunchKernelGGL((joint_dz_reduce_compact_kernel<8>), grid, dim3(256), 0, st, z, elens, ylens, roff, de, dg_slabs, B, T, U1, J, Tc);
  else
    hipLaunchKernelGGL((joint_dz_reduce_compact_kernel<16>), grid, dim3(256), 0, st, z, elens, ylens, roff, de, dg_slabs, B, T, U1, J, Tc);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

// ---- NODE-STATIONARY joint GEMM (round 4): one workgroup = 256 lattice nodes x the WHOLE padded vocabulary.
// The logit GEMMs of the joint are M = 3.6 M nodes x N = 1024 x K = 512 with ~12 VALU per logit in their epilogues;
// as tiles of the general GEMM kernels they ran at 570-640 TFLOP/s (a K = 512 tile is two thirds epilogue; the
// per-64-column (max, sum) partials went to HBM and through a merge kernel).  Here the shapes are used:
//   * K <= 512: a wave's 32 nodes x K operand fragments live in REGISTERS for the whole vocabulary sweep (2 x 16
//     k-steps x 4 VGPRs = 128 at K = 512): h is read from HBM exactly once, never staged in LDS;
//   * the vocabulary is swept in SLICES of 64 columns; a slice of W_out is 64 rows x K bf16 = one contiguous 64 KB,
//     LDS-DMA'd into a two-slice ring shared by the 8 waves (16-B chunk index XOR-ed with row & 15 on the source
//     address: the 16 rows of a fragment read land in 16 different 16-B slots of the 256-B bank row; the whole matrix
//     is 1 MB and stays in every XCD's L2), one barrier per slice;
//   * the products are formed as mfma(W fragment, h fragment): lane & 15 = NODE, registers = 4 adjacent vocabulary
//     columns -- a lane sees 16 columns of its node per slice, so the soft-max statistics are per-lane running
//     (max, sum) pairs merged across the four lane groups ONCE, after the last slice: no cross-lane traffic, no
//     partials, no merge kernel; the blank / label logits are picked up on the way (LSE mode);
//   * DLOGITS mode: the slice's gradients are formed in registers from the per-node scalars (lse, g_blank, g_label,
//     label -- loaded once per workgroup, no packed records), transposed through a 2-KB per-wave slab so that
//     every store instruction writes 8 full 128-B lines of the bf16 image, and their column sums (output-bias
//     gradient) are reduced over the node index with xor-shuffles, across the 8 waves through LDS, and written as
//     ONE slab row per workgroup.
namespace {

__device__ __forceinline__ float jr_xmax4(float v) {      // over the four lanes that share lane & 15 (see flash_attn.hip)
  const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float jr_xsum4(float v) {
  const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// LDS reads the compiler does not see (dst is named in the s_waitcnt statement that follows, CDNA guide 5.7): with the
// slice DMA in flight hipcc puts an s_waitcnt vmcnt(0) in front of the first compiler-visible LDS read behind it (it
// cannot prove that the DMA writes elsewhere) -- for the wave group that starts its epilogue right after the DMA was
// issued that is one DMA round trip per slice (ISA audit).
template <class T>
__device__ __forceinline__ void jr_lds_read16(T& dst, const void* p) {
#ifdef NSP_HOST_EMULATION
  dst = *reinterpret_cast<const T*>(p);
#else
  typedef __attribute__((address_space(3))) unsigned char lds_uchar;
  const unsigned a = (unsigned)(uintptr_t)((lds_uchar*)const_cast<void*>(p));
  asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(a) : "memory");
#endif
}
__device__ __forceinline__ void jr_lds_read4(float& dst, const void* p) {
#ifdef NSP_HOST_EMULATION
  dst = *reinterpret_cast<const float*>(p);
#else
  typedef __attribute__((address_space(3))) unsigned char lds_uchar;
  const unsigned a = (unsigned)(uintptr_t)((lds_uchar*)const_cast<void*>(p));
  asm volatile("ds_read_b32 %0, %1" : "=v"(dst) : "v"(a) : "memory");
#endif
}

template <int NKS, bool LSE>
__global__ __launch_bounds__(512, 2) void rnnt_joint_rows_kernel(
    const __bf16* __restrict__ h16, const __bf16* __restrict__ w16, const float* __restrict__ bias, int M, int V, int Vp,
    int blank, const int* __restrict__ lab, float* __restrict__ lse, float* __restrict__ f1, float* __restrict__ f2,
    float* __restrict__ dbslabs, __bf16* __restrict__ d16, float scale, const float* __restrict__ scale_dev, int dbg) {
  constexpr int K = NKS * 32, PITCH = K * 2, SLICE = 64 * PITCH;
  constexpr int LPR = K / 8;                 // 16-B chunks (= DMA lanes) per row of W
  constexpr int PPW = SLICE / 1024 / 8;      // 1-KB DMA pieces per wave and slice
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];   // 2 slices | 8 x 2 KB staging | 2 x 8 x 64 column sums | Vp bias
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int mrow = blockIdx.x * 256 + wave * 32;        // this wave's first node
  const int nslice = Vp >> 6;
  // ---- this lane's operand fragments of its two 16-node blocks: node mrow + 16 mi + r, k = 32 ks + 8 g .. + 7
  bf16x8 afr[2][NKS];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const __bf16* ap = h16 + (long long)min(mrow + mi * 16 + r, M - 1) * K + g * 8;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) afr[mi][ks] = *reinterpret_cast<const bf16x8*>(ap + ks * 32);
  }
  // ---- slice DMA: piece p = 1 KB = 64 / LPR rows; lane (row, slot) fetches source chunk slot ^ (row & 15)
  const int prow = lane / LPR, pslot = lane % LPR;
  auto issue = [&](int s) {
    unsigned char* dst = smem + (s & 1) * SLICE;
    const char* sbase = reinterpret_cast<const char*>(w16) + (long long)s * SLICE;      // (uniform: the slice is contiguous)
    int pr = prow, ps = pslot;
#ifndef NSP_HOST_EMULATION
    // the piece offsets are recomputed per slice (2 VALU each): hoisted out of the slice loop they are 8 more live
    // registers next to 128 of operand fragments -- the DLOGITS variant spilled, and a scratch reload in front of the
    // DMA issue is an s_waitcnt vmcnt(0) that waits for the previous slice's stores
    asm volatile("" : "+v"(pr), "+v"(ps));
#endif
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int p = wave * PPW + i;
      const int row = p * (64 / LPR) + pr;
      const unsigned off = (unsigned)(row * PITCH + ((ps ^ (row & 15)) << 4));
      __builtin_amdgcn_global_load_lds((glb_void*)(sbase + off), (lds_void*)(dst + p * 1024), 16, 0, 0);
    }
  };
  // fragment read of k-step ks, column block ni: row 16 ni + r, chunk (4 ks + g) ^ r =
  //   ((ks >> 2) << 8) + (((ks & 3) ^ (r >> 2)) << 6) + ((g ^ (r & 3)) << 4): four lane bases, the rest immediates
  int fbase[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) fbase[j] = r * PITCH + ((j ^ (r >> 2)) << 6) + ((g ^ (r & 3)) << 4);
  // per-node scalars
  int labv[2];
  float nls[2], ngb[2], ngl[2];
  const float sc = LSE ? 1.f : scale * (scale_dev ? scale_dev[0] : 1.f);
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int mc = min(mrow + mi * 16 + r, M - 1);
    labv[mi] = lab[mc];
    nls[mi] = LSE ? 0.f : lse[mc];
    const bool rowok = mrow + mi * 16 + r < M;           // nodes beyond M: zero gradients -> zero image rows, zero sums
    ngb[mi] = (LSE || !rowok) ? 0.f : f1[mc] * sc;
    ngl[mi] = (LSE || !rowok) ? 0.f : f2[mc] * sc;
  }
  // The operand fragments and node scalars have LANDED before the slice loop, and the compiler knows it: left alone it
  // sank the loads behind the first barrier and waited for them at their first use INSIDE the loop -- an
  // s_waitcnt vmcnt(0) at the top of every multiply phase, i.e. a wait for the slice DMA issued a few instructions
  // earlier (ISA audit: one DMA round trip per slice).  (A builtin wait is one the compiler's scoreboard sees.)
  __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0)
#ifndef NSP_HOST_EMULATION
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) asm volatile("" : "+v"(afr[mi][ks]));
    asm volatile("" : "+v"(labv[mi]), "+v"(nls[mi]), "+v"(ngb[mi]), "+v"(ngl[mi]));
  }
#endif
  float mx[2] = {-FLT_MAX, -FLT_MAX}, sm[2] = {0.f, 0.f}, lb[2] = {0.f, 0.f}, ll[2] = {-FLT_MAX, -FLT_MAX};
  unsigned char* stg = smem + 2 * SLICE + wave * 2048;
  float* scr = reinterpret_cast<float*>(smem + 2 * SLICE + 8 * 2048);
  // the output bias lives in LDS: a global load in the epilogue would sit behind the slice DMA in the in-order vmcnt
  // queue, and group 1 runs its epilogue right after the DMA has been issued
  float* bias_s = scr + 2 * 8 * 64;
  for (int i = tid; i < Vp; i += 512) bias_s[i] = bias ? bias[i] : 0.f;
  // DLOGITS image of this workgroup's 256 nodes through a buffer descriptor that ENDS at the last node: rows beyond M are
  // dropped by the range check, so the epilogue's four stores are unconditional and the wait below can count them
  const long long rows_here = min(256ll, (long long)M - (long long)blockIdx.x * 256);
  const __amdgpu_buffer_rsrc_t rd16 = __builtin_amdgcn_make_buffer_rsrc(
      LSE ? nullptr : reinterpret_cast<void*>(d16 + (long long)blockIdx.x * 256 * Vp), 0,
      (LSE || (dbg & 1)) ? 0u : (unsigned)(rows_here * Vp * 2), 0x00020000);      // (dbg bit 0, NSP_RNNT_ROWS_DEBUG: every image store out of range -- timing experiments)
  // ---- schedule.  The 8 waves form two GROUPS (waves 0-3 / 4-7: the two waves of every SIMD) that run HALF A SLICE
  // apart: in every half-step one group multiplies a slice (128 MFMAs per wave, s_setprio 1) while the other runs the
  // epilogue of the slice it multiplied before (soft-max arithmetic / gradient image: VALU, LDS, stores) -- with both
  // waves of a SIMD in the same phase (first version: one barrier per slice, profiles/r04w_...) the matrix pipe idled
  // through every epilogue.  Half-step hs: step = hs >> 1; group gq multiplies slice `step` when (hs & 1) == gq, else it
  // finishes slice step (group 0) resp. step - 1 (group 1).  The DMA of slice step + 1 is issued at the start of a step
  // into the buffer both groups left before the barrier that ended the previous step, and waited for (vmcnt(0): also
  // this wave's stores of the step) before the barrier that ends the step.
  const int gq = wave >> 2;
  f32x4 acc[2][4];
  auto mfma_slice = [&](int s) {
    const unsigned char* buf = smem + (s & 1) * SLICE;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      bf16x8 bfr[4];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
        bfr[ni] = *reinterpret_cast<const bf16x8*>(buf + fbase[ks & 3] + (ks >> 2) * 256 + ni * 16 * PITCH);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ni], afr[mi][ks], acc[mi][ni], 0, 0, 0);
    }
  };
  auto epi_slice = [&](int s) {
    // ---- the slice's 2 x 16 logits of this lane: node (mi, r), columns s * 64 + 16 ni + 4 g + e
    const int c0 = s * 64 + 4 * g;
    f32x4 bq[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) jr_lds_read16(bq[ni], bias_s + c0 + ni * 16);
#ifndef NSP_HOST_EMULATION
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bq[0]), "+v"(bq[1]), "+v"(bq[2]), "+v"(bq[3]));
#endif
    float4 b4[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) b4[ni] = make_float4(bq[ni][0], bq[ni][1], bq[ni][2], bq[ni][3]);
    const bool ragged = s * 64 + 64 > V;                   // (uniform) some columns are vocabulary padding
    const bool has_blank = (blank >> 6) == s;
    float cs[4][4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int e = 0; e < 4; ++e) cs[ni][e] = 0.f;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      float v[4][4];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        v[ni][0] = acc[mi][ni][0] + b4[ni].x; v[ni][1] = acc[mi][ni][1] + b4[ni].y;
        v[ni][2] = acc[mi][ni][2] + b4[ni].z; v[ni][3] = acc[mi][ni][3] + b4[ni].w;
      }
      const int rel = labv[mi] - c0;                        // label column relative to this lane's first column
      if constexpr (LSE) {
        if (has_blank) {
#pragma unroll
          for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int e = 0; e < 4; ++e) if (c0 + ni * 16 + e == blank) lb[mi] = v[ni][e];
        }
        {
          // the label column, if it is one of this lane's 16: selects only (as `if (rel == ..) ll = v` hipcc built a
          // tree of divergent branches around every slice's epilogue)
          const int re = rel & 15, rn = rel >> 4;
          float t4[4];
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) {
            const float a01 = re == 1 ? v[ni][1] : v[ni][0], a23 = re == 3 ? v[ni][3] : v[ni][2];
            t4[ni] = re >= 2 ? a23 : a01;
          }
          const float c01 = rn == 1 ? t4[1] : t4[0], c23 = rn == 3 ? t4[3] : t4[2];
          const float cand = rn >= 2 ? c23 : c01;
          ll[mi] = ((unsigned)rel < 64u && re < 4) ? cand : ll[mi];
        }
        if (ragged) {
#pragma unroll
          for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int e = 0; e < 4; ++e) if (c0 + ni * 16 + e >= V) v[ni][e] = -FLT_MAX;
        }
        float ml = v[0][0];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
          for (int e = 0; e < 4; ++e) ml = fmaxf(ml, v[ni][e]);
        const float mn = fmaxf(mx[mi], ml);
        float t = sm[mi] * __expf(mx[mi] - mn);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float ex = __expf(v[ni][e] - mn);
            t += (ragged && c0 + ni * 16 + e >= V) ? 0.f : ex;
          }
        sm[mi] = t;
        mx[mi] = mn;
      } else {
        const float gs = ngb[mi] + ngl[mi];
        const int relh = ((unsigned)rel < 64u && (rel & 15) < 4) ? rel : -1;     // this lane holds the label column
        float gq[4][4];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float t = -gs * __expf(v[ni][e] - nls[mi]);
            gq[ni][e] = t + (relh == ni * 16 + e ? ngl[mi] : 0.f);       // (select + add: no branches)
          }
        if (has_blank) {
#pragma unroll
          for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int e = 0; e < 4; ++e) if (c0 + ni * 16 + e == blank) gq[ni][e] += ngb[mi];
        }
        if (ragged) {
#pragma unroll
          for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int e = 0; e < 4; ++e) if (c0 + ni * 16 + e >= V) gq[ni][e] = 0.f;
        }
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
          for (int e = 0; e < 4; ++e) cs[ni][e] += gq[ni][e];
        // transpose through the wave's slab: [16 nodes][64 columns] bf16, 16-B chunk index XOR-ed with node & 7
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          bf16x4 o;
          o[0] = (__bf16)gq[ni][0]; o[1] = (__bf16)gq[ni][1]; o[2] = (__bf16)gq[ni][2]; o[3] = (__bf16)gq[ni][3];
          *reinterpret_cast<bf16x4*>(stg + r * 128 + (((ni * 2 + (g >> 1)) ^ (r & 7)) << 4) + (g & 1) * 8) = o;
        }
        // (the slab writes above are compiler-visible ds_writes; the asm reads below are ordered behind them by the
        // wait: LDS operations of a wave complete in order)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_;
        u32x4_ q2[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int id = i * 64 + lane, row = id >> 3, ch = id & 7;
          jr_lds_read16(q2[i], stg + row * 128 + ((ch ^ (row & 7)) << 4));
        }
#ifndef NSP_HOST_EMULATION
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q2[0]), "+v"(q2[1]));
#endif
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int id = i * 64 + lane, row = id >> 3, ch = id & 7;
          const unsigned off = (unsigned)((wave * 32 + mi * 16 + row) * Vp + s * 64 + ch * 8) * 2u;
          __builtin_amdgcn_raw_buffer_store_b128(q2[i], rd16, off, 0, 2);      // nt: the 7.4-GB image streams past L2
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    if (!LSE && dbslabs) {
      // column sums over the wave's 32 nodes: over the node index (lane & 15) with xor-shuffles, lanes r = 0 publish
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          // 16-lane row sum in four DPP adds (quad swaps, half-row mirror, row mirror): every lane of the row ends up with
          // the total; as xor-shuffles these were 64 ds_bpermute per slice and wave on an LDS that the fragment reads
          // of the multiplying group already saturate
          float t = cs[ni][e];
          t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0xB1, 0xF, 0xF, true));
          t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x4E, 0xF, 0xF, true));
          t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x141, 0xF, 0xF, true));
          t += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x140, 0xF, 0xF, true));
          cs[ni][e] = t;
        }
      if (r == 0) {
        float* q = scr + (s & 1) * 512 + wave * 64 + 4 * g;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          const f32x4 o = {cs[ni][0], cs[ni][1], cs[ni][2], cs[ni][3]};
#ifdef NSP_HOST_EMULATION
          *reinterpret_cast<f32x4*>(q + ni * 16) = o;
#else
          // (asm: a compiler-visible LDS write behind the slice DMA draws an s_waitcnt vmcnt(0), see jr_lds_read16)
          typedef __attribute__((address_space(3))) unsigned char lds_uchar;
          const unsigned a = (unsigned)(uintptr_t)((lds_uchar*)reinterpret_cast<unsigned char*>(q + ni * 16));
          asm volatile("ds_write_b128 %0, %1" :: "v"(a), "v"(o) : "memory");
#endif
        }
      }
    }
  };
  issue(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#pragma unroll 1
  for (int hs = 0; hs < 2 * nslice + 2; ++hs) {
    const int step = hs >> 1, half = hs & 1;
    if (!LSE && half == 0 && step >= 2 && tid < 64 && dbslabs) {
      // column sums of slice step - 2 (both groups have published them; the slots are rewritten in the second half of
      // this step): 8 waves -> one slab row.  Issued BEFORE the DMA so that the counted wait below need not know it
      const float* q = scr + (step & 1) * 512 + tid;
      float t8[8];
#pragma unroll
      for (int w = 0; w < 8; ++w) jr_lds_read4(t8[w], q + w * 64);
#ifndef NSP_HOST_EMULATION
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t8[0]), "+v"(t8[1]), "+v"(t8[2]), "+v"(t8[3]), "+v"(t8[4]), "+v"(t8[5]), "+v"(t8[6]), "+v"(t8[7]));
#endif
      dbslabs[(long long)blockIdx.x * Vp + (step - 2) * 64 + tid] = ((t8[0] + t8[1]) + (t8[2] + t8[3])) + ((t8[4] + t8[5]) + (t8[6] + t8[7]));
    }
    if (half == 0 && step + 1 < nslice) issue(step + 1);
    if (half == gq) {
      if (step < nslice) {
        __builtin_amdgcn_s_setprio(1);
        mfma_slice(step);
        __builtin_amdgcn_s_setprio(0);
      }
    } else {
      const int se = gq == 0 ? step : step - 1;
      if (se >= 0 && se < nslice) epi_slice(se);
    }
    // (raw barrier: __syncthreads would also drain the DMA issued at the start of the step in its first half)
    // end of a step: the next slice has landed.  DLOGITS: each wave has issued exactly four image stores after the
    // DMA (group 0 a moment ago) -- they may stay in flight
    if (half == 1) {
      if (LSE || (gq == 1 && step == 0)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (group 1 has no epilogue in step 0)
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  }
  if (!LSE && tid < 64 && dbslabs) {      // the last slice's column sums (the loop's last step took slice nslice - 2)
    const float* q = scr + ((nslice - 1) & 1) * 512 + tid;
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) t += q[w * 64];
    dbslabs[(long long)blockIdx.x * Vp + (nslice - 1) * 64 + tid] = t;
  }
  if constexpr (LSE) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const float ma = jr_xmax4(mx[mi]);
      const float sa = jr_xsum4(sm[mi] * __expf(mx[mi] - ma));
      const float blk = jr_xsum4(lb[mi]);                   // (exactly one of the four lanes holds the blank column)
      const float lbl = jr_xmax4(ll[mi]);
      const int m = mrow + mi * 16 + r;
      if (g == 0 && m < M) {
        const float ls = ma + logf(sa);
        lse[m] = ls;
        f1[m] = blk - ls;
        f2[m] = labv[mi] >= 0 ? lbl - ls : -INFINITY;
      }
    }
  }
}

}  // namespace

extern "C" int nsp_rnnt_joint_rows(int epi_mode, const void* h16, const void* w16, const float* bias, long long M, int V,
                                   int Vp, int J, int blank, const int* lab, float* lse, float* f1, float* f2,
                                   float* dbslabs, void* d16, float scale, const float* scale_dev, void* stream) {
  if (M <= 0) return NSP_OK;
  if (M > 0x7fffffffLL - 512 || Vp % 64 || V > Vp || V < 1 || !h16 || !w16 || !lab || !lse || !f1 || !f2 || blank < 0 || blank >= V)
    return NSP_EINVAL;
  if (epi_mode != NSP_EPI_RNNT_LSE && epi_mode != NSP_EPI_RNNT_DLOGITS) return NSP_EINVAL;
  if (epi_mode == NSP_EPI_RNNT_DLOGITS && !d16) return NSP_EINVAL;
  if (J != 128 && J != 256 && J != 512) return NSP_EUNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(h16) | reinterpret_cast<uintptr_t>(w16) | reinterpret_cast<uintptr_t>(d16) |
       reinterpret_cast<uintptr_t>(bias)) & 15) return NSP_EUNSUPPORTED;
  const dim3 grid((unsigned)((M + 255) / 256));
  const size_t lds = 2 * 64 * (size_t)J * 2 + 8 * 2048 + 2 * 8 * 64 * 4 + (size_t)Vp * 4;     // ring | staging | column sums | bias
  if (lds > 163840) return NSP_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
#define NSP_JR(NKS, LSE_)                                                                                              \
  do {                                                                                                                 \
    /* opt in to the whole 160 KB once per (kernel, device): `lds` depends on the vocabulary, so a later call with a   \
       larger one (a second decoder, a sub-task head) must not find a smaller limit from the first call */            \
    static unsigned long long optin_devs = 0;                                                                          \
    int dev_ = 0;                                                                                                      \
    if (hipGetDevice(&dev_) != hipSuccess) dev_ = -1;           /* (unknown device: opt in on every call) */          \
    if (dev_ < 0 || dev_ >= 64 || !((optin_devs >> dev_) & 1ull)) {                                                     \
      if (hipFuncSetAttribute((const void*)rnnt_joint_rows_kernel<NKS, LSE_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                              163840) != hipSuccess) return NSP_EUNSUPPORTED;                                          \
      if (dev_ >= 0 && dev_ < 64) optin_devs |= 1ull << dev_;                                                           \
    }                                                                                                                  \
    hipLaunchKernelGGL((rnnt_joint_rows_kernel<NKS, LSE_>), grid, dim3(512), lds, st,                                  \
                       reinterpret_cast<const __bf16*>(h16), reinterpret_cast<const __bf16*>(w16), bias, (int)M, V, Vp, \
                       blank, lab, lse, f1, f2, dbslabs, reinterpret_cast<__bf16*>(d16), scale, scale_dev, dbg);       \
  } while (0)
  const bool l = epi_mode == NSP_EPI_RNNT_LSE;
  {
    static int gdbg = -1;      // NSP_GEMM_DEBUG=1: the launch in the GEMM launcher's line format (tools/pmc_traffic.py reads algbytes)
    if (gdbg < 0) { const char* e = getenv("NSP_GEMM_DEBUG"); gdbg = e ? atoi(e) : 0; }
    if (gdbg) {
      // algorithmic HBM bytes: h and W_out once; LSE: label in, three floats out per node; DLOGITS: label + three floats in
      // per node, the bf16 image and the bias-gradient slab rows out
      long long bytes = 2ll * M * J + 2ll * Vp * J + 16ll * M;
      if (!l) bytes += 2ll * M * Vp + (dbslabs ? 4ll * ((M + 255) / 256) * Vp : 0);
      fprintf(stderr, "[nsp_gemm_bf16] M %lld N %d K %d node-stationary joint kernel epi %d algbytes %lld\n", M, Vp, J, epi_mode, bytes);
    }
  }
  const char* edbg = getenv("NSP_RNNT_ROWS_DEBUG");
  const int dbg = edbg ? atoi(edbg) : 0;
  if (J == 512) { if (l) NSP_JR(16, true); else NSP_JR(16, false); }
  else if (J == 256) { if (l) NSP_JR(8, true); else NSP_JR(8, false); }
  else { if (l) NSP_JR(4, true); else NSP_JR(4, false); }
#undef NSP_JR
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}
