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
    *reinterpret_cast<bf16x8*>(h16 + (row0 + u) * J + 8 * c) = q;
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

// One pass over dz (bf16 [M,J], compact rows) for both joint-input gradients.  grid: (J/32, B, nslab);
// block 256: thread = (4 columns of the workgroup's 32-column slice, one of 32 u-lanes).  A thread meets
// the same (u, columns) for every t, so sum_t lives in registers; sum_u is reduced across the u-lanes
// once per t.  Padded positions (t >= T_b, u > U_b) are written as zeros.
template <int KU>
__global__ __launch_bounds__(256) void joint_dz_reduce_compact_kernel(
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
  dim3 grid(J / 32, B, nslab);
  const int ku = nsp_cdiv(U1, 32);
  if (ku <= 4)
    hipLaunchKernelGGL((joint_dz_reduce_compact_kernel<4>), grid, dim3(256), 0, st, z, elens, ylens, roff, de, dg_slabs, B, T, U1, J, Tc);
  else if (ku <= 8)
    hipLaunchKernelGGL((joint_dz_reduce_compact_kernel<8>), grid, dim3(256), 0, st, z, elens, ylens, roff, de, dg_slabs, B, T, U1, J, Tc);
  else
    hipLaunchKernelGGL((joint_dz_reduce_compact_kernel<16>), grid, dim3(256), 0, st, z, elens, ylens, roff, de, dg_slabs, B, T, U1, J, Tc);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}
