// attention.hip -- masked softmax over attention scores with the relative-
// position (BD) term folded in, forward and backward.  One wave64 per score row
// (b,h,i); the row is staged in LDS so HBM sees one read + one write per
// element.  Mask predicates (padding / causal+lookahead / chunkwise) are
// evaluated in-kernel from klens: the reference's [B,T,T] mask tensor and its
// H-fold repeat (relative_multihead_attention.py:164-166) are never built.
//
// Reference semantics reproduced exactly:
//   e(i,j) = (AC(i,j) + QP(i, min(|i-j|, clamp))) / sqrt(d_k)
//   masked e := -FLT_MAX (finite!), softmax over ALL Tk keys; a fully masked
//   row therefore becomes uniform 1/Tk, as in the reference.
#include "common.h"

namespace {

__device__ __forceinline__ bool key_visible(const nsp_attn_mask_params& p, int klen, int i, int j) {
  bool ok = j < klen;
  if (p.causal) ok = ok && (j <= i + p.lookahead);
  if (p.chunk_nc > 0) {
    int c0 = (i / p.chunk_nc) * p.chunk_nc;
    int lo = c0 - p.chunk_nl;
    if (lo < 0) lo = 0;
    ok = ok && (j >= lo) && (j < c0 + p.chunk_nc);
  }
  return ok;
}

__device__ __forceinline__ int rel_index(int i, int j, int clamp) {
  int r = i > j ? i - j : j - i;
  if (clamp > 0 && r > clamp) r = clamp;
  return r;
}

__global__ __launch_bounds__(256) void attn_softmax_fwd_kernel(const float* __restrict__ S,
                                                               const float* __restrict__ QP,
                                                               void* __restrict__ Pout,
                                                               void* __restrict__ Pdrop,
                                                               const nsp_attn_mask_params p) {
  extern __shared__ __attribute__((aligned(16))) float rowbuf[];  // [4][Tk]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long long nrows = (long long)p.B * p.H * p.Tq;
  float* buf = rowbuf + (long long)w * p.Tk;
  const int tkp = p.p_bf16 ? p.tk_pitch : p.Tk;
  for (long long row = (long long)blockIdx.x * 4 + w; row < nrows; row += (long long)gridDim.x * 4) {
    const int i = (int)(row % p.Tq);
    const int h = (int)((row / p.Tq) % p.H);
    const int b = (int)(row / ((long long)p.Tq * p.H));
    const int klen = p.klens ? p.klens[b] : p.Tk;
    const float* s = S + row * p.Tk;
    const float* qp = QP ? QP + (((long long)b * p.Tq + i) * p.H + h) * p.r_pitch : nullptr;
    float mx = -FLT_MAX;
    for (int j = lane; j < p.Tk; j += 64) {
      float e = s[j];
      if (qp) e += qp[rel_index(i, j, p.clamp)];
      e *= p.scale;
      if (!key_visible(p, klen, i, j)) e = -FLT_MAX;
      buf[j] = e;
      mx = fmaxf(mx, e);
    }
    mx = wave_reduce_max(mx);
    float sum = 0.f;
    for (int j = lane; j < p.Tk; j += 64) {
      float ex = __expf(buf[j] - mx);
      buf[j] = ex;
      sum += ex;
    }
    sum = wave_reduce_sum(sum);
    const float inv = 1.f / sum;
    for (int j = lane; j < tkp; j += 64) {
      const float pr = j < p.Tk ? buf[j] * inv : 0.f;  // pad columns of the bf16 image are zero
      float pd = pr;
      if (Pdrop && j < p.Tk)
        pd = pr * nsp_keep_scale(p.seed, p.offset + (unsigned long long)(row * p.Tk + j), p.dropout_p);
      if (p.p_bf16) {
        reinterpret_cast<__bf16*>(Pout)[row * tkp + j] = (__bf16)pr;
        if (Pdrop) reinterpret_cast<__bf16*>(Pdrop)[row * tkp + j] = (__bf16)pd;
      } else {
        reinterpret_cast<float*>(Pout)[row * tkp + j] = pr;
        if (Pdrop) reinterpret_cast<float*>(Pdrop)[row * tkp + j] = pd;
      }
    }
  }
}

__global__ __launch_bounds__(256) void attn_softmax_bwd_kernel(const void* __restrict__ P,
                                                               const float* __restrict__ dP,
                                                               void* __restrict__ dS,
                                                               float* __restrict__ dQP,
                                                               const nsp_attn_mask_params p) {
  extern __shared__ __attribute__((aligned(16))) float rowbuf[];  // [4][Tk]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long long nrows = (long long)p.B * p.H * p.Tq;
  float* buf = rowbuf + (long long)w * p.Tk;
  const int tkp = p.p_bf16 ? p.tk_pitch : p.Tk;
  for (long long row = (long long)blockIdx.x * 4 + w; row < nrows; row += (long long)gridDim.x * 4) {
    const int i = (int)(row % p.Tq);
    const int h = (int)((row / p.Tq) % p.H);
    const int b = (int)(row / ((long long)p.Tq * p.H));
    const int klen = p.klens ? p.klens[b] : p.Tk;
    const float* g = dP + row * p.Tk;
    float t = 0.f;
    for (int j = lane; j < p.Tk; j += 64) {
      float gj = g[j];
      if (p.dropout_p > 0.f)
        gj *= nsp_keep_scale(p.seed, p.offset + (unsigned long long)(row * p.Tk + j), p.dropout_p);
      buf[j] = gj;
      const float pj = p.p_bf16 ? (float)reinterpret_cast<const __bf16*>(P)[row * tkp + j]
                                : reinterpret_cast<const float*>(P)[row * tkp + j];
      t += pj * gj;
    }
    t = wave_reduce_sum(t);
    float far = 0.f;  // sum of dS over |i-j| >= clamp
    for (int j = lane; j < tkp; j += 64) {
      float ds = 0.f;
      if (j < p.Tk) {
        const float pj = p.p_bf16 ? (float)reinterpret_cast<const __bf16*>(P)[row * tkp + j]
                                  : reinterpret_cast<const float*>(P)[row * tkp + j];
        ds = pj * (buf[j] - t) * p.scale;
        if (!key_visible(p, klen, i, j)) ds = 0.f;  // masked_fill_ blocks the gradient
        buf[j] = ds;
        if (p.clamp > 0) {
          int r = i > j ? i - j : j - i;
          if (r >= p.clamp) far += ds;
        }
      }
      if (p.p_bf16) reinterpret_cast<__bf16*>(dS)[row * tkp + j] = (__bf16)ds;
      else reinterpret_cast<float*>(dS)[row * tkp + j] = ds;
    }
    if (dQP) {
      float* dq = dQP + (((long long)b * p.Tq + i) * p.H + h) * p.r_pitch;
      __builtin_amdgcn_wave_barrier();
      if (p.clamp > 0) far = wave_reduce_sum(far);
      for (int r = lane; r < p.r_pitch; r += 64) {
        float v = 0.f;
        if (r < p.R) {
          if (p.clamp > 0 && r == p.clamp) {
            v = far;
          } else {
            if (i - r >= 0 && i - r < p.Tk) v += buf[i - r];
            if (r > 0 && i + r < p.Tk) v += buf[i + r];
          }
        }
        dq[r] = v;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace

extern "C" int nsp_attn_softmax_fwd(const float* S, const float* QP, void* Pout, void* Pdrop,
                                    const nsp_attn_mask_params* pp, void* stream) {
  if (!pp || !S || !Pout) return NSP_EINVAL;
  nsp_attn_mask_params p = *pp;
  if (p.r_pitch < p.R) p.r_pitch = p.R;
  if (p.p_bf16 && p.tk_pitch < p.Tk) return NSP_EINVAL;
  // the gather index min(|i-j|, clamp) never exceeds min(clamp, Tk-1)
  if (QP && p.R < (p.clamp > 0 && p.clamp + 1 < p.Tk ? p.clamp + 1 : p.Tk)) return NSP_EINVAL;
  if (p.dropout_p > 0.f && !Pdrop) return NSP_EINVAL;
  const size_t shmem = sizeof(float) * 4 * (size_t)p.Tk;
  if (shmem > 150 * 1024) return NSP_EUNSUPPORTED;
  long long nrows = (long long)p.B * p.H * p.Tq;
  int grid = nsp_cdiv(nrows, 4);
  if (grid > 256 * 32) grid = 256 * 32;
  if (shmem > 64 * 1024)
    hipFuncSetAttribute((const void*)attn_softmax_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
  hipLaunchKernelGGL(attn_softmax_fwd_kernel, dim3(grid), dim3(256), shmem, (hipStream_t)stream, S, QP,
                     Pout, p.dropout_p > 0.f ? Pdrop : nullptr, p);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

extern "C" int nsp_attn_softmax_bwd(const void* P, const float* dP, void* dS, float* dQP,
                                    const nsp_attn_mask_params* pp, void* stream) {
  if (!pp || !P || !dP || !dS) return NSP_EINVAL;
  nsp_attn_mask_params p = *pp;
  if (p.r_pitch < p.R) p.r_pitch = p.R;
  if (p.p_bf16 && p.tk_pitch < p.Tk) return NSP_EINVAL;
  const size_t shmem = sizeof(float) * 4 * (size_t)p.Tk;
  if (shmem > 150 * 1024) return NSP_EUNSUPPORTED;
  long long nrows = (long long)p.B * p.H * p.Tq;
  int grid = nsp_cdiv(nrows, 4);
  if (grid > 256 * 32) grid = 256 * 32;
  if (shmem > 64 * 1024)
    hipFuncSetAttribute((const void*)attn_softmax_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
  hipLaunchKernelGGL(attn_softmax_bwd_kernel, dim3(grid), dim3(256), shmem, (hipStream_t)stream, P, dP,
                     dS, dQP, p);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}
