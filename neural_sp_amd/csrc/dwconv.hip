// dwconv.hip -- depthwise Conv1d over time on channels-last [B,T,C]
// (Conformer convolution module, reference conformer_convolution.py:47-53,
// 111-113) and MaxPool1d time subsampling (subsampling.py:188-209).
//
// Channels-last keeps the [B,T,C] activation layout of the rest of the block:
// the reference's two transposes [B,T,C]<->[B,C,T] (:106,:115) disappear.
// Each lane owns 4 adjacent channels (16 B), a wave covers 256 channels, so
// every global access is a fully coalesced 1 KiB row segment; the k-tap
// re-reads of x hit L1/L2 (HBM sees x once).  HBM-bound.
#include "common.h"

namespace {

// y[b,t,c] = bias[c] + sum_j w[jj][c] * x[b, t + j - pad, c],  jj = flip ? k-1-j : j
// wt is tap-major: [k][C]
__global__ __launch_bounds__(256) void dwconv_fwd_kernel(const float* __restrict__ x,
                                                         const float* __restrict__ wt,
                                                         const float* __restrict__ bias,
                                                         float* __restrict__ y, int B, int T, int C,
                                                         int k, int pad, int flip) {
  const int C4 = C >> 2;
  const long long total = (long long)B * T * C4;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
    const int c4 = (int)(idx % C4);
    const int t = (int)((idx / C4) % T);
    const int b = (int)(idx / ((long long)C4 * T));
    float4 acc = bias ? reinterpret_cast<const float4*>(bias)[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float* xb = x + (long long)b * T * C;
    for (int j = 0; j < k; ++j) {
      const int ts = t + j - pad;
      if (ts < 0 || ts >= T) continue;
      const float4 xv = reinterpret_cast<const float4*>(xb + (long long)ts * C)[c4];
      const float4 wv = reinterpret_cast<const float4*>(wt + (long long)(flip ? k - 1 - j : j) * C)[c4];
      acc.x += wv.x * xv.x; acc.y += wv.y * xv.y; acc.z += wv.z * xv.z; acc.w += wv.w * xv.w;
    }
    reinterpret_cast<float4*>(y)[idx] = acc;
  }
}

// Sliding-window variant for the kernel sizes the recipes use (K = 15, 7; pad = (K-1)/2): a thread owns
// 4 channels and TT = 8 consecutive frames of one utterance, loads the TT + K - 1 input rows of its
// window ONCE (all requests in flight together) and keeps them and the K taps in registers.  The
// kernel above issues K loads per output row (15x the L1/L2 requests of this one for K = 15; measured
// 2.3 TB/s of algorithmic traffic at 51200 x 512, against ~4.5 TB/s for the streaming kernels).
template <int K, int TT>
__global__ __launch_bounds__(256) void dwconv_fwd_sw_kernel(const float* __restrict__ x, const float* __restrict__ wt,
                                                            const float* __restrict__ bias, float* __restrict__ y,
                                                            int B, int T, int C, int flip) {
  constexpr int PAD = (K - 1) / 2, W = TT + K - 1;
  const int C4 = C >> 2;
  const int truns = (T + TT - 1) / TT;
  const long long total = (long long)B * truns * C4;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int c4 = (int)(idx % C4);
  const int tr = (int)((idx / C4) % truns);
  const int b = (int)(idx / ((long long)C4 * truns));
  const int t0 = tr * TT;
  const float4* xb = reinterpret_cast<const float4*>(x + (long long)b * T * C) + c4;
  float4 win[W];
#pragma unroll
  for (int i = 0; i < W; ++i) {
    const int ts = t0 + i - PAD;
    win[i] = (ts >= 0 && ts < T) ? xb[(long long)ts * C4] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float4 w[K];
#pragma unroll
  for (int j = 0; j < K; ++j) w[j] = reinterpret_cast<const float4*>(wt + (long long)(flip ? K - 1 - j : j) * C)[c4];
  const float4 bv = bias ? reinterpret_cast<const float4*>(bias)[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
  float4* yb = reinterpret_cast<float4*>(y + (long long)b * T * C) + c4;
#pragma unroll
  for (int i = 0; i < TT; ++i) {
    if (t0 + i >= T) break;
    float4 acc = bv;
#pragma unroll
    for (int j = 0; j < K; ++j) {
      acc.x = fmaf(w[j].x, win[i + j].x, acc.x); acc.y = fmaf(w[j].y, win[i + j].y, acc.y);
      acc.z = fmaf(w[j].z, win[i + j].z, acc.z); acc.w = fmaf(w[j].w, win[i + j].w, acc.w);
    }
    yb[(long long)(t0 + i) * C4] = acc;
  }
}

// ---- Round 6: GLU folded into the depthwise conv (conformer_convolution.py:110-113: glu -> depthwise_conv).  In
// throughput mode the first pointwise conv leaves its [rows, 2C] output as a bf16 image (nsp_glu_fwd_b16's input).  The
// GLU used to turn it into a fp32 [rows, C] tensor that the depthwise conv read back (and kept for its weight gradient),
// and backward moved the fp32 gradient of that tensor from the conv's data-gradient pass to the GLU's backward pass: 16
// bytes per element of fp32 hand-overs and two launches per direction.  Here the conv kernels apply the GLU to the bf16
// image as they load it (same arithmetic: (float)a * sigmoid((float)b)), and the data-gradient kernel finishes with the
// GLU's backward: it writes d(image) as the bf16 operand of the pointwise conv's gradient GEMMs and the column sums of it
// (that conv's bias gradient) as slabs, exactly what nsp_glu_bwd_b16 produced.
__device__ __forceinline__ float4 glu_load4(const __bf16* __restrict__ row, int C, int tc) {
  const bf16x4 a = reinterpret_cast<const bf16x4*>(row)[tc];
  const bf16x4 b = reinterpret_cast<const bf16x4*>(row + C)[tc];
  return make_float4((float)a[0] * nsp_sigmoid((float)b[0]), (float)a[1] * nsp_sigmoid((float)b[1]),
                     (float)a[2] * nsp_sigmoid((float)b[2]), (float)a[3] * nsp_sigmoid((float)b[3]));
}

// forward: y = bias + sum_j w[j] glu(h2)[t + j - PAD]; thread = 4 channels x TT frames (the sliding window of
// dwconv_fwd_sw_kernel, its rows made from the bf16 image)
template <int K, int TT>
__global__ __launch_bounds__(256) void dwconv_glu_fwd_sw_kernel(const __bf16* __restrict__ h2, const float* __restrict__ wt,
                                                                const float* __restrict__ bias, float* __restrict__ y,
                                                                int B, int T, int C) {
  constexpr int PAD = (K - 1) / 2, W = TT + K - 1;
  const int C4 = C >> 2;
  const int truns = (T + TT - 1) / TT;
  const long long total = (long long)B * truns * C4;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int c4 = (int)(idx % C4);
  const int tr = (int)((idx / C4) % truns);
  const int b = (int)(idx / ((long long)C4 * truns));
  const int t0 = tr * TT;
  const __bf16* hb = h2 + (long long)b * T * 2 * C;
  float4 win[W];
#pragma unroll
  for (int i = 0; i < W; ++i) {
    const int ts = t0 + i - PAD;
    // (clamped address + select: the loads stay unconditional)
    const float4 v = glu_load4(hb + (long long)min(max(ts, 0), T - 1) * 2 * C, C, c4);
    const bool ok = ts >= 0 && ts < T;
    win[i] = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
  }
  float4 w[K];
#pragma unroll
  for (int j = 0; j < K; ++j) w[j] = reinterpret_cast<const float4*>(wt + (long long)j * C)[c4];
  const float4 bv = bias ? reinterpret_cast<const float4*>(bias)[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
  float4* yb = reinterpret_cast<float4*>(y + (long long)b * T * C) + c4;
#pragma unroll
  for (int i = 0; i < TT; ++i) {
    if (t0 + i >= T) break;
    float4 acc = bv;
#pragma unroll
    for (int j = 0; j < K; ++j) {
      acc.x = fmaf(w[j].x, win[i + j].x, acc.x); acc.y = fmaf(w[j].y, win[i + j].y, acc.y);
      acc.z = fmaf(w[j].z, win[i + j].z, acc.z); acc.w = fmaf(w[j].w, win[i + j].w, acc.w);
    }
    yb[(long long)(t0 + i) * C4] = acc;
  }
}

// backward (data): d glu = sum_j w[K-1-j] dy[t + j - PAD] (the forward kernel on dy with flipped taps), then the GLU's
// backward on the bf16 image: g[., c] = d glu * sig(b), g[., C + c] = d glu * a * sig(b) (1 - sig(b)), written as bf16,
// and their column sums.  Block = (C / 4 channel quads) x (256 / (C / 4) runs of TT frames), `rloop` runs per thread;
// one slab row [2C] per block, every row written.  grid: (ceil(truns / (nrl * rloop)), B).
template <int K, int TT>
__global__ __launch_bounds__(256) void dwconv_glu_bwd_sw_kernel(const __bf16* __restrict__ h2, const float* __restrict__ dy,
                                                                const float* __restrict__ wt, __bf16* __restrict__ g16,
                                                                float* __restrict__ colsum, int B, int T, int C, int rloop) {
  constexpr int PAD = (K - 1) / 2, W = TT + K - 1;
  __shared__ float red[256][8];
  const int C4 = C >> 2;
  const int tc = threadIdx.x % C4, rl = threadIdx.x / C4, nrl = 256 / C4;
  const int truns = (T + TT - 1) / TT;
  const int b = blockIdx.y;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float4 w[K];
#pragma unroll
  for (int j = 0; j < K; ++j) w[j] = reinterpret_cast<const float4*>(wt + (long long)(K - 1 - j) * C)[tc];
  const float4* db4 = reinterpret_cast<const float4*>(dy + (long long)b * T * C) + tc;
  const __bf16* hb = h2 + (long long)b * T * 2 * C;
  __bf16* gb = g16 + (long long)b * T * 2 * C;
  for (int it = 0; it < rloop; ++it) {
    const int tr = (blockIdx.x * rloop + it) * nrl + rl;
    if (tr >= truns) break;
    const int t0 = tr * TT;
    float4 win[W];
#pragma unroll
    for (int i = 0; i < W; ++i) {
      const int ts = t0 + i - PAD;
      const float4 v = db4[(long long)min(max(ts, 0), T - 1) * C4];
      const bool ok = ts >= 0 && ts < T;
      win[i] = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
    }
#pragma unroll
    for (int i = 0; i < TT; ++i) {
      if (t0 + i >= T) break;
      float4 dg = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int j = 0; j < K; ++j) {
        dg.x = fmaf(w[j].x, win[i + j].x, dg.x); dg.y = fmaf(w[j].y, win[i + j].y, dg.y);
        dg.z = fmaf(w[j].z, win[i + j].z, dg.z); dg.w = fmaf(w[j].w, win[i + j].w, dg.w);
      }
      const long long rb = (long long)(t0 + i) * 2 * C;
      const bf16x4 a = reinterpret_cast<const bf16x4*>(hb + rb)[tc];
      const bf16x4 bb = reinterpret_cast<const bf16x4*>(hb + rb + C)[tc];
      const float g[4] = {dg.x, dg.y, dg.z, dg.w};
      bf16x4 da, dbv;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float sg = nsp_sigmoid((float)bb[e]);
        const float va = g[e] * sg, vb = g[e] * (float)a[e] * sg * (1.f - sg);
        da[e] = (__bf16)va; dbv[e] = (__bf16)vb;
        acc[e] += va; acc[4 + e] += vb;
      }
      reinterpret_cast<bf16x4*>(gb + rb)[tc] = da;
      reinterpret_cast<bf16x4*>(gb + rb + C)[tc] = dbv;
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[threadIdx.x][e] = acc[e];
  __syncthreads();
  if (rl == 0) {
    float sum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < nrl; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) sum[e] += red[j * C4 + tc][e];
    float* cs = colsum + ((long long)blockIdx.y * gridDim.x + blockIdx.x) * 2 * C;
    *reinterpret_cast<float4*>(cs + tc * 4) = make_float4(sum[0], sum[1], sum[2], sum[3]);
    *reinterpret_cast<float4*>(cs + C + tc * 4) = make_float4(sum[4], sum[5], sum[6], sum[7]);
  }
}

// dwt[j][c] += sum_{b,t} dy[b,t,c] * x[b, t + j - pad, c] ; dbias[c] += sum dy
// grid: (ceil(C4/64), nchunks); block 256 = 4 waves striding over the rows of a chunk
__global__ __launch_bounds__(256) void dwconv_wgrad_kernel(const float* __restrict__ x,
                                                           const float* __restrict__ dy,
                                                           float* __restrict__ dwt,
                                                           float* __restrict__ dbias, int B, int T,
                                                           int C, int k, int pad, int rows_per_chunk) {
  extern __shared__ __attribute__((aligned(16))) float4 sh4[];  // [4 waves][(k+1)][64 lanes]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int C4 = C >> 2;
  const int c4 = blockIdx.x * 64 + lane;
  const long long r0 = (long long)blockIdx.y * rows_per_chunk;
  const long long r1 = min((long long)B * T, r0 + rows_per_chunk);
  const bool active = c4 < C4;
  // accumulate taps in chunks of 8 to bound registers for large kernels
  for (int j0 = 0; j0 < k; j0 += 8) {
    float4 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 accb = make_float4(0.f, 0.f, 0.f, 0.f);
    if (active) {
      for (long long r = r0 + w; r < r1; r += 4) {
        const int t = (int)(r % T);
        const long long b = r / T;
        const float4 g = reinterpret_cast<const float4*>(dy + r * C)[c4];
        if (j0 == 0) { accb.x += g.x; accb.y += g.y; accb.z += g.z; accb.w += g.w; }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int ts = t + j0 + j - pad;
          if (j0 + j < k && ts >= 0 && ts < T) {
            const float4 xv = reinterpret_cast<const float4*>(x + (b * T + ts) * C)[c4];
            acc[j].x += g.x * xv.x; acc[j].y += g.y * xv.y; acc[j].z += g.z * xv.z; acc[j].w += g.w * xv.w;
          }
        }
      }
    }
    // combine the 4 waves
#pragma unroll
    for (int j = 0; j < 8; ++j) sh4[(w * 9 + j) * 64 + lane] = acc[j];
    sh4[(w * 9 + 8) * 64 + lane] = accb;
    __syncthreads();
    if (w == 0 && active) {
#pragma unroll
      for (int j = 0; j < 9; ++j) {
        float4 s = sh4[(0 * 9 + j) * 64 + lane];
#pragma unroll
        for (int ww = 1; ww < 4; ++ww) {
          float4 o = sh4[(ww * 9 + j) * 64 + lane];
          s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
        }
        float* dst = nullptr;
        if (j < 8) {
          if (j0 + j < k) dst = dwt + (long long)(j0 + j) * C + c4 * 4;
        } else if (j0 == 0 && dbias) {
          dst = dbias + c4 * 4;
        }
        if (dst) {
          unsafeAtomicAdd(dst + 0, s.x); unsafeAtomicAdd(dst + 1, s.y);
          unsafeAtomicAdd(dst + 2, s.z); unsafeAtomicAdd(dst + 3, s.w);
        }
      }
    }
    __syncthreads();
  }
}

// MaxPool1d(kernel=stride=factor, ceil_mode=True) over time; argmax saved as the
// source time index for the backward scatter.
__global__ void maxpool1d_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                     int* __restrict__ argmax, int B, int T, int To, int C,
                                     int factor) {
  const int C4 = C >> 2;
  const long long total = (long long)B * To * C4;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
    const int c4 = (int)(idx % C4);
    const int to = (int)((idx / C4) % To);
    const long long b = idx / ((long long)C4 * To);
    float4 best = make_float4(-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX);
    int4 bi = make_int4(to * factor, to * factor, to * factor, to * factor);
    for (int f = 0; f < factor; ++f) {
      const int t = to * factor + f;
      if (t >= T) break;
      const float4 v = reinterpret_cast<const float4*>(x + (b * T + t) * C)[c4];
      if (v.x > best.x || f == 0) { if (v.x > best.x || f == 0) { best.x = v.x; bi.x = t; } }
      if (v.y > best.y || f == 0) { best.y = v.y; bi.y = t; }
      if (v.z > best.z || f == 0) { best.z = v.z; bi.z = t; }
      if (v.w > best.w || f == 0) { best.w = v.w; bi.w = t; }
    }
    reinterpret_cast<float4*>(y)[idx] = best;
    reinterpret_cast<int4*>(argmax)[idx] = bi;
  }
}

__global__ void maxpool1d_bwd_kernel(const float* __restrict__ dy, const int* __restrict__ argmax,
                                     float* __restrict__ dx, int B, int T, int To, int C,
                                     int factor) {
  // one thread per INPUT element quad: gather from its pooling window's output
  const int C4 = C >> 2;
  const long long total = (long long)B * T * C4;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
    const int c4 = (int)(idx % C4);
    const int t = (int)((idx / C4) % T);
    const long long b = idx / ((long long)C4 * T);
    const int to = t / factor;
    const long long o = (b * To + to) * C4 + c4;
    const float4 g = reinterpret_cast<const float4*>(dy)[o];
    const int4 a = reinterpret_cast<const int4*>(argmax)[o];
    reinterpret_cast<float4*>(dx)[idx] = make_float4(a.x == t ? g.x : 0.f, a.y == t ? g.y : 0.f,
                                                     a.z == t ? g.z : 0.f, a.w == t ? g.w : 0.f);
  }
}

inline int ew_grid(long long n) {
  long long g = (n + 255) / 256;
  if (g > 256 * 16) g = 256 * 16;
  if (g < 1) g = 1;
  return (int)g;
}

// LDS-tiled weight gradient for k <= 15.  grid: (ceil(C4/64), tsplit, B); block 256:
// thread = (one float4 of channels, one of 4 tap groups of 4 slots: taps 4jg..4jg+3, slot k = bias).
// Per 24-row chunk the x rows (with their k-1 halo rows; zero outside the utterance) and the dy
// rows are staged once; every (row, tap) product then reads LDS only.  (The register version
// issues 1 + k global loads per row and is TA-bound: 80 us for 52 MB.)
constexpr int DW_RC = 24;
template <bool GLU>      // GLU: x is the bf16 [rows, 2C] image of the first pointwise conv and the conv's input is glu(x) (round 6)
__global__ __launch_bounds__(256) void dwconv_wgrad_lds_kernel(const float* __restrict__ x,
                                                               const float* __restrict__ dy,
                                                               float* __restrict__ part, int B, int T, int C,
                                                               int k, int pad, int rows_per_wg) {
  extern __shared__ __attribute__((aligned(16))) float4 dsh[];
  float4* xs = dsh;                              // [(DW_RC + k - 1)][64]
  float4* dys = dsh + (DW_RC + k - 1) * 64;      // [DW_RC][64]
  const int lane = threadIdx.x & 63, jg = threadIdx.x >> 6;
  const int C4 = C >> 2;
  const int c4 = blockIdx.x * 64 + lane;
  const bool active = c4 < C4;
  const long long b = blockIdx.z;
  const int ts = blockIdx.y * rows_per_wg;
  const int te = min(T, ts + rows_per_wg);
  const float4* x4 = reinterpret_cast<const float4*>(x) + b * T * C4;
  const __bf16* h2 = reinterpret_cast<const __bf16*>(x) + b * T * 2 * C;
  const float4* d4 = reinterpret_cast<const float4*>(dy) + b * T * C4;
  float4 acc[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) acc[s] = make_float4(0.f, 0.f, 0.f, 0.f);
  // The rows of chunk c+1 are fetched into registers while chunk c is multiplied out of LDS (the first version
  // ran load -> barrier -> multiply -> barrier with two workgroups per CU: 2.5 TB/s).  Loads are unconditional
  // with clamped row / channel indices and zeroed by a select when they are committed to LDS.
  constexpr int NX = (DW_RC + 14 + 3) / 4, ND = DW_RC / 4;     // x / dy rows per thread and chunk (k <= 15)
  float4 px[NX], pd[ND];
  const int c4c = min(c4, C4 - 1);
  int tf = ts;                 // next chunk to fetch
  int tl = ts, nl = 0;         // chunk resident in LDS (nl rows; 0 = none yet)
  while (true) {
    const bool fetched = tf < te;
    if (fetched) {
#pragma unroll
      for (int q = 0; q < NX; ++q) {
        const int tt = min(max(tf - pad + jg + 4 * q, 0), T - 1);
        if constexpr (GLU) px[q] = glu_load4(h2 + (long long)tt * 2 * C, C, c4c);
        else px[q] = x4[(long long)tt * C4 + c4c];
      }
#pragma unroll
      for (int q = 0; q < ND; ++q) pd[q] = d4[(long long)min(tf + jg + 4 * q, T - 1) * C4 + c4c];
    }
    for (int i = 0; i < nl; ++i) {
      const float4 g = dys[i * 64 + lane];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int j = jg * 4 + s;
        if (j < k) {
          const float4 xv = xs[(i + j) * 64 + lane];
          acc[s].x += g.x * xv.x; acc[s].y += g.y * xv.y; acc[s].z += g.z * xv.z; acc[s].w += g.w * xv.w;
        } else if (j == k) {
          acc[s].x += g.x; acc[s].y += g.y; acc[s].z += g.z; acc[s].w += g.w;
        }
      }
    }
    if (!fetched) break;
    __syncthreads();             // the resident chunk has been multiplied by every wave
    tl = tf;
    nl = min(DW_RC, te - tf);
#pragma unroll
    for (int q = 0; q < NX; ++q) {
      const int i = jg + 4 * q, tt = tl - pad + i;
      // (component-wise: `cond ? px[q] : zero` on float4 objects makes hipcc select between two ADDRESSES and
      // spill the whole prefetch array to scratch)
      const bool okx = active && tt >= 0 && tt < T;
      if (i < nl + k - 1)
        xs[i * 64 + lane] = make_float4(okx ? px[q].x : 0.f, okx ? px[q].y : 0.f, okx ? px[q].z : 0.f, okx ? px[q].w : 0.f);
    }
#pragma unroll
    for (int q = 0; q < ND; ++q) {
      const int i = jg + 4 * q;
      if (i < nl)
        dys[i * 64 + lane] = make_float4(active ? pd[q].x : 0.f, active ? pd[q].y : 0.f, active ? pd[q].z : 0.f, active ? pd[q].w : 0.f);
    }
    __syncthreads();
    tf += DW_RC;
  }
  if (active) {
    float* slab = part + ((long long)blockIdx.z * gridDim.y + blockIdx.y) * (long long)(k + 1) * C;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int j = jg * 4 + s;
      if (j <= k) *reinterpret_cast<float4*>(slab + (long long)j * C + c4 * 4) = acc[s];
    }
  }
}

// ---- Round 6: streaming weight gradient for K = 15 / 7 (pad = (K-1)/2).  dw[j][c] = sum_t dy[t][c] x[t + j - PAD][c]:
// a thread owns 4 channels and one time range of one utterance and walks it ONCE with the K input rows around the current
// frame in registers (a rotating window, unrolled K-fold so that every index is static): per frame one dy row and one new
// x row are loaded, K fused multiply-adds per channel.  No LDS, no barriers; every element is read exactly once (+ the
// K - 1 halo rows per range).  The LDS-tiled kernel above stages 24-row chunks between two barriers and reached 40 % of
// the copy rate at the step's shapes (138 us per call against 54 us of bytes).  x is either fp32 [B*T, C] or -- GLU --
// the bf16 [B*T, 2C] image of the first pointwise conv, gated as it is loaded.  grid: (C/4/64 channel groups of a wave,
// tsplit, B) as the tiled kernel; block = 64 threads x up to 4 sub-ranges (threadIdx.y) reduced through LDS at the end.
template <int K, bool GLU>
__global__ __launch_bounds__(256) void dwconv_wgrad_stream_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                  float* __restrict__ part, int B, int T, int C, int rows_per_wg) {
  constexpr int PAD = (K - 1) / 2;
  __shared__ float4 red[K + 1][64];            // the four waves add their sums into it one after the other (16 KB, not 64)
  const int lane = threadIdx.x & 63, sub = threadIdx.x >> 6;
  const int C4 = C >> 2;
  const int c4 = blockIdx.x * 64 + lane;
  const int c4c = min(c4, C4 - 1);
  const long long b = blockIdx.z;
  const int ts = blockIdx.y * rows_per_wg, te = min(T, ts + rows_per_wg);
  // four sub-ranges of the workgroup's range, one per wave
  const int span = (te - ts + 3) / 4;
  const int t0 = ts + sub * span, t1 = min(te, t0 + span);
  const float4* x4 = reinterpret_cast<const float4*>(x) + b * T * C4 + c4c;
  const __bf16* h2 = reinterpret_cast<const __bf16*>(x) + b * T * 2 * C;
  const float4* d4 = reinterpret_cast<const float4*>(dy) + b * T * C4 + c4c;
  auto xrow = [&](int t) -> float4 {            // input row t of this thread's channels (zero outside the utterance)
    const int tc = min(max(t, 0), T - 1);
    float4 v;
    if constexpr (GLU) v = glu_load4(h2 + (long long)tc * 2 * C, C, c4c);
    else v = x4[(long long)tc * C4];
    const bool ok = t >= 0 && t < T;
    return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
  };
  float4 acc[K + 1];
#pragma unroll
  for (int j = 0; j <= K; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (t0 < t1) {
    // window w[j] = x[t + j - PAD]; slot (j + phase) % K rotates, phase static inside the K-fold unrolled body
    float4 w[K];
#pragma unroll
    for (int j = 0; j < K - 1; ++j) w[j + 1] = xrow(t0 + j - PAD);          // rows t0 - PAD .. t0 + PAD - 1 sit in slots 1 .. K-1
    for (int tb = t0; tb < t1; tb += K) {
#pragma unroll
      for (int ph = 0; ph < K; ++ph) {
        const int t = tb + ph;
        // entering frame t: the oldest row leaves, row t + PAD enters: slots are indexed (j + ph + 1) % K for tap j
        w[ph % K] = xrow(t + PAD);
        const float4 g = d4[(long long)min(t, T - 1) * C4];
        const bool live = t < t1;
        const float4 gg = make_float4(live ? g.x : 0.f, live ? g.y : 0.f, live ? g.z : 0.f, live ? g.w : 0.f);
#pragma unroll
        for (int j = 0; j < K; ++j) {
          const float4 xv = w[(j + ph + 1) % K];
          acc[j].x = fmaf(gg.x, xv.x, acc[j].x); acc[j].y = fmaf(gg.y, xv.y, acc[j].y);
          acc[j].z = fmaf(gg.z, xv.z, acc[j].z); acc[j].w = fmaf(gg.w, xv.w, acc[j].w);
        }
        acc[K].x += gg.x; acc[K].y += gg.y; acc[K].z += gg.z; acc[K].w += gg.w;
      }
    }
  }
  for (int q = 0; q < 4; ++q) {
    if (sub == q) {
#pragma unroll
      for (int j = 0; j <= K; ++j) {
        float4 v = acc[j];
        if (q > 0) {
          const float4 o = red[j][lane];
          v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        }
        red[j][lane] = v;
      }
    }
    __syncthreads();
  }
  if (c4 < C4) {
    float* slab = part + ((long long)blockIdx.z * gridDim.y + blockIdx.y) * (long long)(K + 1) * C;
    for (int j = sub; j <= K; j += 4) *reinterpret_cast<float4*>(slab + (long long)j * C + c4 * 4) = red[j][lane];
  }
}

}  // namespace

extern "C" int nsp_dwconv1d_wgrad_slabs(const float* x, const float* dy, float* part, int tsplit, int B,
                                        int T, int C, int k, int pad, void* stream) {
  if (C % 4 || k < 1 || k > 15 || tsplit < 1) return NSP_EUNSUPPORTED;
  const int rows_per_wg = nsp_cdiv(T, tsplit);
  if ((k == 15 || k == 7) && pad == (k - 1) / 2) {
    const dim3 grid(nsp_cdiv(C / 4, 64), tsplit, B);
    if (k == 15) hipLaunchKernelGGL((dwconv_wgrad_stream_kernel<15, false>), grid, dim3(256), 0, (hipStream_t)stream, x, dy, part, B, T, C, rows_per_wg);
    else hipLaunchKernelGGL((dwconv_wgrad_stream_kernel<7, false>), grid, dim3(256), 0, (hipStream_t)stream, x, dy, part, B, T, C, rows_per_wg);
    NSP_LAUNCH_CHECK();
    return NSP_OK;
  }
  const size_t shmem = sizeof(float4) * 64 * (size_t)(2 * DW_RC + k - 1);
  hipLaunchKernelGGL(dwconv_wgrad_lds_kernel<false>, dim3(nsp_cdiv(C / 4, 64), tsplit, B), dim3(256), shmem,
                     (hipStream_t)stream, x, dy, part, B, T, C, k, pad, rows_per_wg);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

// ---- GLU folded into the depthwise conv (bf16 image h2 [B*T, 2C] of the first pointwise conv): see the kernels
static bool dwglu_ok(const void* h2, int C, int k, int pad) {
  return C % 4 == 0 && C / 4 <= 256 && 256 % (C / 4) == 0 && (k == 15 || k == 7) && pad == (k - 1) / 2 &&
         (reinterpret_cast<uintptr_t>(h2) & 7) == 0;
}
static int dwglu_rloop(int B, int T, int C) {
  const int nrl = 256 / (C / 4), truns = (T + 7) / 8;
  int r = nsp_cdiv((long long)B * truns, (long long)nrl * 3072);
  return r < 1 ? 1 : (r > 8 ? 8 : r);
}
extern "C" int nsp_dwconv1d_glu_bwd_slabs(int B, int T, int C) {
  if (C % 4 || C / 4 > 256 || 256 % (C / 4)) return 0;
  const int nrl = 256 / (C / 4), truns = (T + 7) / 8;
  return B * nsp_cdiv(truns, nrl * dwglu_rloop(B, T, C));
}
extern "C" int nsp_dwconv1d_glu_fwd(const void* h2, const float* wt, const float* bias, float* y, int B, int T, int C,
                                    int k, int pad, void* stream) {
  if (!dwglu_ok(h2, C, k, pad)) return NSP_EUNSUPPORTED;
  const long long threads = (long long)B * ((T + 7) / 8) * (C / 4);
  const dim3 grid((unsigned)((threads + 255) / 256));
  const __bf16* h = reinterpret_cast<const __bf16*>(h2);
  if (k == 15)
    hipLaunchKernelGGL((dwconv_glu_fwd_sw_kernel<15, 8>), grid, dim3(256), 0, (hipStream_t)stream, h, wt, bias, y, B, T, C);
  else
    hipLaunchKernelGGL((dwconv_glu_fwd_sw_kernel<7, 8>), grid, dim3(256), 0, (hipStream_t)stream, h, wt, bias, y, B, T, C);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}
extern "C" int nsp_dwconv1d_glu_bwd(const void* h2, const float* dy, const float* wt, void* g16, float* colsum_slabs,
                                    int B, int T, int C, int k, int pad, void* stream) {
  if (!dwglu_ok(h2, C, k, pad) || !colsum_slabs || (reinterpret_cast<uintptr_t>(g16) & 7)) return NSP_EUNSUPPORTED;
  const int nrl = 256 / (C / 4), truns = (T + 7) / 8, rloop = dwglu_rloop(B, T, C);
  const dim3 grid(nsp_cdiv(truns, nrl * rloop), B);
  const __bf16* h = reinterpret_cast<const __bf16*>(h2);
  __bf16* g = reinterpret_cast<__bf16*>(g16);
  if (k == 15)
    hipLaunchKernelGGL((dwconv_glu_bwd_sw_kernel<15, 8>), grid, dim3(256), 0, (hipStream_t)stream, h, dy, wt, g, colsum_slabs, B, T, C, rloop);
  else
    hipLaunchKernelGGL((dwconv_glu_bwd_sw_kernel<7, 8>), grid, dim3(256), 0, (hipStream_t)stream, h, dy, wt, g, colsum_slabs, B, T, C, rloop);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}
extern "C" int nsp_dwconv1d_glu_wgrad_slabs(const void* h2, const float* dy, float* part, int tsplit, int B,
                                            int T, int C, int k, int pad, void* stream) {
  if (C % 4 || k < 1 || k > 15 || tsplit < 1 || (reinterpret_cast<uintptr_t>(h2) & 7)) return NSP_EUNSUPPORTED;
  const int rows_per_wg = nsp_cdiv(T, tsplit);
  if ((k == 15 || k == 7) && pad == (k - 1) / 2) {
    const dim3 grid(nsp_cdiv(C / 4, 64), tsplit, B);
    const float* xf = reinterpret_cast<const float*>(h2);
    if (k == 15) hipLaunchKernelGGL((dwconv_wgrad_stream_kernel<15, true>), grid, dim3(256), 0, (hipStream_t)stream, xf, dy, part, B, T, C, rows_per_wg);
    else hipLaunchKernelGGL((dwconv_wgrad_stream_kernel<7, true>), grid, dim3(256), 0, (hipStream_t)stream, xf, dy, part, B, T, C, rows_per_wg);
    NSP_LAUNCH_CHECK();
    return NSP_OK;
  }
  const size_t shmem = sizeof(float4) * 64 * (size_t)(2 * DW_RC + k - 1);
  hipLaunchKernelGGL(dwconv_wgrad_lds_kernel<true>, dim3(nsp_cdiv(C / 4, 64), tsplit, B), dim3(256), shmem,
                     (hipStream_t)stream, reinterpret_cast<const float*>(h2), dy, part, B, T, C, k, pad, rows_per_wg);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

extern "C" int nsp_dwconv1d_fwd(const float* x, const float* wt, const float* bias, float* y, int B,
                                int T, int C, int k, int pad, int flip, void* stream) {
  if (C % 4 || k < 1) return NSP_EUNSUPPORTED;
  if ((k == 15 || k == 7) && pad == (k - 1) / 2) {
    const long long threads = (long long)B * ((T + 7) / 8) * (C / 4);
    const dim3 grid((unsigned)((threads + 255) / 256));
    if (k == 15)
      hipLaunchKernelGGL((dwconv_fwd_sw_kernel<15, 8>), grid, dim3(256), 0, (hipStream_t)stream, x, wt, bias, y, B, T, C, flip);
    else
      hipLaunchKernelGGL((dwconv_fwd_sw_kernel<7, 8>), grid, dim3(256), 0, (hipStream_t)stream, x, wt, bias, y, B, T, C, flip);
    NSP_LAUNCH_CHECK();
    return NSP_OK;
  }
  hipLaunchKernelGGL(dwconv_fwd_kernel, dim3(ew_grid((long long)B * T * (C / 4))), dim3(256), 0,
                     (hipStream_t)stream, x, wt, bias, y, B, T, C, k, pad, flip);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

extern "C" int nsp_dwconv1d_wgrad(const float* x, const float* dy, float* dwt, float* dbias, int B,
                                  int T, int C, int k, int pad, void* stream) {
  if (C % 4 || k < 1) return NSP_EUNSUPPORTED;
  const int gx = nsp_cdiv(C / 4, 64);
  const long long rows = (long long)B * T;
  int chunks = 1024 / gx;
  if (chunks < 1) chunks = 1;
  int rpc = nsp_cdiv(rows, chunks);
  if (rpc < 64) rpc = 64;
  chunks = nsp_cdiv(rows, rpc);
  const size_t shmem = sizeof(float4) * 4 * 9 * 64;
  hipLaunchKernelGGL(dwconv_wgrad_kernel, dim3(gx, chunks), dim3(256), shmem, (hipStream_t)stream, x,
                     dy, dwt, dbias, B, T, C, k, pad, rpc);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

extern "C" int nsp_maxpool1d_fwd(const float* x, float* y, int* argmax, int B, int T, int C,
                                 int factor, void* stream) {
  if (C % 4 || factor < 1) return NSP_EUNSUPPORTED;
  const int To = (T + factor - 1) / factor;
  hipLaunchKernelGGL(maxpool1d_fwd_kernel, dim3(ew_grid((long long)B * To * (C / 4))), dim3(256), 0,
                     (hipStream_t)stream, x, y, argmax, B, T, To, C, factor);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

extern "C" int nsp_maxpool1d_bwd(const float* dy, const int* argmax, float* dx, int B, int T, int C,
                                 int factor, void* stream) {
  if (C % 4 || factor < 1) return NSP_EUNSUPPORTED;
  const int To = (T + factor - 1) / factor;
  hipLaunchKernelGGL(maxpool1d_bwd_kernel, dim3(ew_grid((long long)B * T * (C / 4))), dim3(256), 0,
                     (hipStream_t)stream, dy, argmax, dx, B, T, To, C, factor);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}
