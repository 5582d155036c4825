// ctc.hip -- CTC loss on CDNA4: fused log-softmax normaliser + alpha/beta
// lattices + gradient w.r.t. the logits, and the label-smoothing KL term.
//
// Reference semantics (ctc.py:139-150): nn.CTCLoss(reduction='sum',
// zero_infinity=True)(logits.transpose(0,1).log_softmax(2), ys, elens, ylens)/B
// i.e. per-utterance NLL summed over the batch; an infinite NLL contributes 0
// loss and 0 gradient.  The log-softmax is never materialised: a row kernel
// produces lse[b,t]; lattice cells read logit[b,t,l'_s] - lse[b,t] directly.
//
// Lattice kernel: one workgroup per utterance; alpha runs on the first half of
// the workgroup and beta on the second half concurrently, the previous time
// step lives in LDS (double-buffered), one barrier per frame.  All recursions
// are fp32 log-space; alpha/beta are written to the caller's workspace for the
// gradient kernel.  Bandwidth/latency-bound (T'' serial steps).
#include "common.h"

namespace {

// lse[row] = logsumexp(x[row, :V]); one wave per row; rows with t >= elens[b] are skipped
__global__ __launch_bounds__(256) void row_lse_kernel(const float* __restrict__ x,
                                                      const int* __restrict__ elens,
                                                      float* __restrict__ lse, int B, int T, int V) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long long nrows = (long long)B * T;
  for (long long row = (long long)blockIdx.x * 4 + w; row < nrows; row += (long long)gridDim.x * 4) {
    const int t = (int)(row % T);
    const int b = (int)(row / T);
    if (elens && t >= elens[b]) {
      if (lane == 0) lse[row] = 0.f;
      continue;
    }
    const float* xr = x + row * V;
    float mx = -FLT_MAX;
    for (int v = lane; v < V; v += 64) mx = fmaxf(mx, xr[v]);
    mx = wave_reduce_max(mx);
    float s = 0.f;
    for (int v = lane; v < V; v += 64) s += expf(xr[v] - mx);
    s = wave_reduce_sum(s);
    if (lane == 0) lse[row] = mx + logf(s);
  }
}

__device__ __forceinline__ int ctc_label(const int* __restrict__ lab, int s, int blank) {
  return (s & 1) ? lab[s >> 1] : blank;
}

__global__ __launch_bounds__(512) void ctc_alpha_beta_kernel(
    const float* __restrict__ logits, const float* __restrict__ lse, const int* __restrict__ labels,
    const int* __restrict__ elens, const int* __restrict__ ylens, float* __restrict__ alpha,
    float* __restrict__ beta, float* __restrict__ nll, int B, int T, int V, int Lmax, int blank) {
  extern __shared__ __attribute__((aligned(16))) float sh[];  // a[2][Smax], b[2][Smax]
  const int Smax = 2 * Lmax + 1;
  float* abuf = sh;
  float* bbuf = sh + 2 * Smax;
  const int b = blockIdx.x;
  const int half = blockDim.x >> 1;
  const bool is_beta = threadIdx.x >= half;
  const int tid = is_beta ? threadIdx.x - half : threadIdx.x;
  const int Tb = min(elens[b], T);
  const int Lb = min(ylens[b], Lmax);
  const int Sb = 2 * Lb + 1;
  const int* lab = labels + (long long)b * Lmax;
  const float* lg = logits + (long long)b * T * V;
  const float* ls = lse + (long long)b * T;
  float* al = alpha + (long long)b * T * Smax;
  float* be = beta + (long long)b * T * Smax;

  if (Tb <= 0) {
    if (threadIdx.x == 0) nll[b] = (Lb == 0) ? 0.f : INFINITY;
    return;
  }
  for (int step = 0; step < Tb; ++step) {
    const int cur = step & 1, prv = cur ^ 1;
    if (!is_beta) {
      const int t = step;
      for (int s = tid; s < Sb; s += half) {
        const int l = ctc_label(lab, s, blank);
        const float lp = lg[(long long)t * V + l] - ls[t];
        float a;
        if (t == 0) {
          a = (s <= 1) ? lp : -INFINITY;
        } else {
          a = abuf[prv * Smax + s];
          if (s >= 1) a = nsp_logaddexp(a, abuf[prv * Smax + s - 1]);
          if (s >= 2 && l != blank && l != ctc_label(lab, s - 2, blank))
            a = nsp_logaddexp(a, abuf[prv * Smax + s - 2]);
          a += lp;
        }
        abuf[cur * Smax + s] = a;
        al[(long long)t * Smax + s] = a;
      }
    } else {
      const int t = Tb - 1 - step;
      for (int s = tid; s < Sb; s += half) {
        const int l = ctc_label(lab, s, blank);
        const float lp = lg[(long long)t * V + l] - ls[t];
        float v;
        if (step == 0) {
          v = (s >= Sb - 2) ? lp : -INFINITY;
        } else {
          v = bbuf[prv * Smax + s];
          if (s + 1 < Sb) v = nsp_logaddexp(v, bbuf[prv * Smax + s + 1]);
          if (s + 2 < Sb && l != blank && l != ctc_label(lab, s + 2, blank))
            v = nsp_logaddexp(v, bbuf[prv * Smax + s + 2]);
          v += lp;
        }
        bbuf[cur * Smax + s] = v;
        be[(long long)t * Smax + s] = v;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const int last = (Tb - 1) & 1;
    float ll = abuf[last * Smax + Sb - 1];
    if (Sb >= 2) ll = nsp_logaddexp(ll, abuf[last * Smax + Sb - 2]);
    nll[b] = -ll;  // +inf when no valid alignment exists
  }
}

// grad[b,t,v] = gscale * (softmax(b,t,v) - occupancy(b,t,v)); 0 for t >= elens[b]
// or when nll[b] is infinite (zero_infinity).  One workgroup per (b,t) row,
// the row is assembled in LDS.
__global__ __launch_bounds__(256) void ctc_grad_kernel(
    const float* __restrict__ logits, const float* __restrict__ lse, const int* __restrict__ labels,
    const int* __restrict__ elens, const int* __restrict__ ylens, const float* __restrict__ alpha,
    const float* __restrict__ beta, const float* __restrict__ nll, float* __restrict__ grad,
    float gscale, int B, int T, int V, int Lmax, int blank) {
  extern __shared__ __attribute__((aligned(16))) float row[];  // [V]
  const long long r = blockIdx.x;
  const int t = (int)(r % T);
  const int b = (int)(r / T);
  const int Smax = 2 * Lmax + 1;
  float* gr = grad + r * V;
  const float nl = nll[b];
  const int Tb = min(elens[b], T);
  if (t >= Tb || isinf(nl) || isnan(nl)) {
    for (int v = threadIdx.x; v < V; v += blockDim.x) gr[v] = 0.f;
    return;
  }
  const float* lg = logits + r * V;
  const float ls = lse[r];
  for (int v = threadIdx.x; v < V; v += blockDim.x) row[v] = expf(lg[v] - ls);
  __syncthreads();
  const int Lb = min(ylens[b], Lmax);
  const int Sb = 2 * Lb + 1;
  const int* lab = labels + (long long)b * Lmax;
  const float* al = alpha + ((long long)b * T + t) * Smax;
  const float* be = beta + ((long long)b * T + t) * Smax;
  for (int s = threadIdx.x; s < Sb; s += blockDim.x) {
    const int l = ctc_label(lab, s, blank);
    const float lp = lg[l] - ls;
    const float occ = expf(al[s] + be[s] - lp + nl);
    atomicAdd(&row[l], -occ);
  }
  __syncthreads();
  for (int v = threadIdx.x; v < V; v += blockDim.x) gr[v] = gscale * row[v];
}

// label-smoothing KL: one wave per (b,t) row
__global__ __launch_bounds__(256) void ctc_kldiv_kernel(const float* __restrict__ logits,
                                                        const int* __restrict__ elens,
                                                        float* __restrict__ kl_sum,
                                                        float* __restrict__ grad, float gscale,
                                                        int accumulate, int B, int T, int V) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long long nrows = (long long)B * T;
  const float c = logf(1.f / (float)(V - 1));
  float local = 0.f;
  for (long long row = (long long)blockIdx.x * 4 + w; row < nrows; row += (long long)gridDim.x * 4) {
    const int t = (int)(row % T);
    const int b = (int)(row / T);
    float* gr = grad ? grad + row * V : nullptr;
    if (t >= elens[b]) {
      if (gr && !accumulate)
        for (int v = lane; v < V; v += 64) gr[v] = 0.f;
      continue;
    }
    const float* xr = logits + row * V;
    float mx = -FLT_MAX;
    for (int v = lane; v < V; v += 64) mx = fmaxf(mx, xr[v]);
    mx = wave_reduce_max(mx);
    float s = 0.f;
    for (int v = lane; v < V; v += 64) s += expf(xr[v] - mx);
    s = wave_reduce_sum(s);
    const float ls = mx + logf(s);
    float f = 0.f;
    for (int v = lane; v < V; v += 64) {
      const float lp = xr[v] - ls;
      f += expf(lp) * (lp - c);
    }
    f = wave_reduce_sum(f);
    local += f;
    if (gr) {
      for (int v = lane; v < V; v += 64) {
        const float lp = xr[v] - ls;
        const float g = gscale * expf(lp) * ((lp - c) - f);
        gr[v] = accumulate ? gr[v] + g : g;
      }
    }
  }
  if (lane == 0 && local != 0.f) atomicAdd(kl_sum, local);
}

// CTC forced alignment (reference ctc.py:657-753).  One workgroup per utterance:
//   pass A: fb[t][s] = alpha_t(s)                      (emission at t included)
//   pass B: fb[t][s] += btilde_t(s), btilde = beta without the emission at t
//           -> fb = log posterior numerator alpha*beta/y, as accumulated by the
//           reference's two `cum_log_prob +=` passes (:547-561, :685-699)
//   pass C: greedy left-to-right walk: from the previously chosen lattice state s
//           only {s, s+1, s+2 (if a different non-blank label)} are reachable
//           (:708-725); take the arg-max of fb over them (lowest index on ties),
//           then emit the leftmost frame of every label run (:727-748) and the
//           last frame for <eos>.
__global__ __launch_bounds__(256) void ctc_align_kernel(
    const float* __restrict__ logits, const float* __restrict__ lse, const int* __restrict__ labels,
    const int* __restrict__ elens, const int* __restrict__ ylens, float* __restrict__ fb,
    int* __restrict__ best, int* __restrict__ trig, int B, int T, int V, int Lmax, int blank) {
  extern __shared__ __attribute__((aligned(16))) float sh[];  // [2][Smax]
  const int Smax = 2 * Lmax + 1;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int Tb = min(elens[b], T);
  const int Lb = min(ylens[b], Lmax);
  const int Sb = 2 * Lb + 1;
  const int* lab = labels + (long long)b * Lmax;
  const float* lg = logits + (long long)b * T * V;
  const float* ls = lse + (long long)b * T;
  float* f = fb + (long long)b * T * Smax;
  int* bp = best + (long long)b * T;
  int* tp = trig + (long long)b * (Lmax + 1);
  for (int i = tid; i < Lmax + 1; i += blockDim.x) tp[i] = 0;
  if (Tb <= 0) return;
  // pass A
  for (int t = 0; t < Tb; ++t) {
    const int cur = t & 1, prv = cur ^ 1;
    for (int s = tid; s < Sb; s += blockDim.x) {
      const int l = ctc_label(lab, s, blank);
      const float lp = lg[(long long)t * V + l] - ls[t];
      float a;
      if (t == 0) {
        a = (s <= 1) ? lp : -INFINITY;
      } else {
        a = sh[prv * Smax + s];
        if (s >= 1) a = nsp_logaddexp(a, sh[prv * Smax + s - 1]);
        if (s >= 2 && l != blank && l != ctc_label(lab, s - 2, blank))
          a = nsp_logaddexp(a, sh[prv * Smax + s - 2]);
        a += lp;
      }
      sh[cur * Smax + s] = a;
      f[(long long)t * Smax + s] = a;
    }
    __syncthreads();
  }
  // pass B
  for (int step = 0; step < Tb; ++step) {
    const int t = Tb - 1 - step;
    const int cur = step & 1, prv = cur ^ 1;
    for (int s = tid; s < Sb; s += blockDim.x) {
      const int l = ctc_label(lab, s, blank);
      float bt;
      if (step == 0) {
        bt = (s >= Sb - 2) ? 0.f : -INFINITY;
      } else {
        bt = sh[prv * Smax + s];
        if (s + 1 < Sb) bt = nsp_logaddexp(bt, sh[prv * Smax + s + 1]);
        if (s + 2 < Sb && l != blank && l != ctc_label(lab, s + 2, blank))
          bt = nsp_logaddexp(bt, sh[prv * Smax + s + 2]);
      }
      f[(long long)t * Smax + s] += bt;
      sh[cur * Smax + s] = bt + (lg[(long long)t * V + l] - ls[t]);
    }
    __syncthreads();
  }
  // pass C (sequential)
  if (tid == 0) {
    int sprev = -1;  // virtual start: reachable {0, 1}
    for (int t = 0; t < Tb; ++t) {
      int lo = sprev < 0 ? 0 : sprev;
      int hi = sprev < 0 ? 1 : sprev + 2;
      int arg = lo;
      float bestv = -INFINITY;
      bool have = false;
      for (int s = lo; s <= hi && s < Sb; ++s) {
        if (sprev >= 0 && s == sprev + 2) {
          const int l = ctc_label(lab, s, blank);
          if (l == blank || l == ctc_label(lab, sprev, blank)) continue;
        }
        const float v = f[(long long)t * Smax + s];
        if (!have || v > bestv) { bestv = v; arg = s; have = true; }
      }
      bp[t] = ctc_label(lab, arg, blank);
      sprev = arg;
    }
    tp[Lb] = Tb - 1;  // <eos> boundary (:732)
    int n = 0;
    for (int t = 0; t < Tb; ++t) {
      const int tok = bp[t];
      if (tok == blank) continue;
      if (t > 0 && tok == bp[t - 1]) continue;
      if (n < Lmax + 1) tp[n] = t;
      ++n;
    }
  }
}

}  // namespace

extern "C" long long nsp_ctc_align_workspace_bytes(int B, int T, int Lmax) {
  const long long Smax = 2LL * (Lmax < 1 ? 1 : Lmax) + 1;
  return sizeof(float) * ((long long)B * T * Smax + (long long)B * T) + sizeof(int) * (long long)B * T;
}

extern "C" int nsp_ctc_forced_align(const float* logits, const int* labels, const int* elens,
                                    const int* ylens, int* trigger_points, void* workspace, int B,
                                    int T, int V, int Lmax, int blank, void* stream) {
  if (B <= 0 || T <= 0 || V <= 1 || !workspace) return NSP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int Lm = Lmax < 1 ? 1 : Lmax;
  const long long Smax = 2LL * Lm + 1;
  float* fb = (float*)workspace;
  float* lse = fb + (long long)B * T * Smax;
  int* best = (int*)(lse + (long long)B * T);
  int grid = nsp_cdiv((long long)B * T, 4);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(row_lse_kernel, dim3(grid), dim3(256), 0, st, logits, elens, lse, B, T, V);
  const size_t sh = sizeof(float) * 2 * Smax;
  if (sh > 150 * 1024) return NSP_EUNSUPPORTED;
  hipLaunchKernelGGL(ctc_align_kernel, dim3(B), dim3(256), sh, st, logits, lse, labels, elens, ylens,
                     fb, best, trigger_points, B, T, V, Lm, blank);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

extern "C" long long nsp_ctc_workspace_bytes(int B, int T, int Lmax) {
  const long long Smax = 2LL * Lmax + 1;
  return sizeof(float) * ((long long)B * T * Smax * 2 + (long long)B * T);
}

extern "C" int nsp_ctc_loss_fwd_bwd(const float* logits, const int* labels, const int* elens,
                                    const int* ylens, float* nll, float* grad, float gscale,
                                    void* workspace, int B, int T, int V, int Lmax, int blank,
                                    void* stream) {
  if (B <= 0 || T <= 0 || V <= 1 || Lmax < 0 || !workspace) return NSP_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int Lm = Lmax < 1 ? 1 : Lmax;
  const long long Smax = 2LL * Lm + 1;
  float* alpha = (float*)workspace;
  float* beta = alpha + (long long)B * T * Smax;
  float* lse = beta + (long long)B * T * Smax;
  int grid = nsp_cdiv((long long)B * T, 4);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(row_lse_kernel, dim3(grid), dim3(256), 0, st, logits, elens, lse, B, T, V);
  const size_t sh = sizeof(float) * 4 * Smax;
  if (sh > 150 * 1024) return NSP_EUNSUPPORTED;
  if (sh > 64 * 1024)
    hipFuncSetAttribute((const void*)ctc_alpha_beta_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
  hipLaunchKernelGGL(ctc_alpha_beta_kernel, dim3(B), dim3(512), sh, st, logits, lse, labels, elens,
                     ylens, alpha, beta, nll, B, T, V, Lm, blank);
  if (grad) {
    const size_t shg = sizeof(float) * V;
    if (shg > 150 * 1024) return NSP_EUNSUPPORTED;
    if (shg > 64 * 1024)
      hipFuncSetAttribute((const void*)ctc_grad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shg);
    hipLaunchKernelGGL(ctc_grad_kernel, dim3(B * T), dim3(256), shg, st, logits, lse, labels, elens,
                       ylens, alpha, beta, nll, grad, gscale, B, T, V, Lm, blank);
  }
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

extern "C" int nsp_ctc_kldiv_fwd_bwd(const float* logits, const int* elens, float* kl_sum,
                                     float* grad, float gscale, int accumulate, int B, int T, int V,
                                     void* stream) {
  if (B <= 0 || T <= 0 || V <= 1) return NSP_EINVAL;
  int grid = nsp_cdiv((long long)B * T, 4);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(ctc_kldiv_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, logits, elens,
                     kl_sum, grad, gscale, accumulate, B, T, V);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}
