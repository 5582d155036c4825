// elementwise.hip -- HBM-bound helpers: axpby, activation fwd/bwd, column sums
// (bias gradients), GLU, ReLU backward, SpecAugment band zeroing, batch padding.
// All kernels are grid-stride with 16 B/lane accesses where alignment allows.
#include "common.h"

namespace {

constexpr int EW_THREADS = 256;
inline int ew_grid(long long n_vec) {
  long long g = (n_vec + EW_THREADS - 1) / EW_THREADS;
  if (g > 256 * 16) g = 256 * 16;
  if (g < 1) g = 1;
  return (int)g;
}

__global__ void axpby_kernel(const float* __restrict__ x, const float* __restrict__ z,
                             float* __restrict__ y, float alpha, float beta, long long n) {
  long long n4 = n >> 2;
  long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 a = reinterpret_cast<const float4*>(x)[i];
    float4 r = make_float4(alpha * a.x, alpha * a.y, alpha * a.z, alpha * a.w);
    if (z) {
      float4 b = reinterpret_cast<const float4*>(z)[i];
      r.x += beta * b.x; r.y += beta * b.y; r.z += beta * b.z; r.w += beta * b.w;
    }
    reinterpret_cast<float4*>(y)[i] = r;
  }
  for (long long i = (n4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    y[i] = alpha * x[i] + (z ? beta * z[i] : 0.f);
}

__global__ void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int act,
                               long long n) {
  long long n4 = n >> 2;
  long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 a = reinterpret_cast<const float4*>(x)[i];
    reinterpret_cast<float4*>(y)[i] =
        make_float4(nsp_act(a.x, act), nsp_act(a.y, act), nsp_act(a.z, act), nsp_act(a.w, act));
  }
  for (long long i = (n4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    y[i] = nsp_act(x[i], act);
}

__global__ void dact_mul_kernel(const float* __restrict__ dy, const float* __restrict__ pre,
                                float* __restrict__ out, int act, float alpha, long long n) {
  long long n4 = n >> 2;
  long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 g = reinterpret_cast<const float4*>(dy)[i];
    float4 p = reinterpret_cast<const float4*>(pre)[i];
    reinterpret_cast<float4*>(out)[i] =
        make_float4(alpha * g.x * nsp_dact(p.x, act), alpha * g.y * nsp_dact(p.y, act),
                    alpha * g.z * nsp_dact(p.z, act), alpha * g.w * nsp_dact(p.w, act));
  }
  for (long long i = (n4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = alpha * dy[i] * nsp_dact(pre[i], act);
}

// y = alpha * x * keep(seed, offset+i)/(1-p): forward dropout and, applied to dy, its backward
__global__ void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, float p,
                               float alpha, unsigned long long seed, unsigned long long offset,
                               long long n) {
  long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    y[i] = alpha * x[i] * nsp_keep_scale(seed, offset + (unsigned long long)i, p);
}

// g = alpha * dy * keep(seed, offset+i)/(1-p) * act'(pre): the one pass that turns the incoming
// gradient of a Linear's epilogue into the operand of its dgrad/wgrad GEMMs (fp32 or bf16 out)
__global__ void grad_prep_kernel(const float* __restrict__ dy, const void* __restrict__ pre,
                                 int pre_bf16, void* __restrict__ out, int out_bf16, int act,
                                 float alpha, float p, unsigned long long seed,
                                 unsigned long long offset, long long n) {
  const long long n4 = n >> 2;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 g4 = reinterpret_cast<const float4*>(dy)[i];
    float g[4] = {g4.x * alpha, g4.y * alpha, g4.z * alpha, g4.w * alpha};
    if (p > 0.f) {
      float kp[4];
      nsp_keep_scale4(seed, offset + (unsigned long long)(i * 4), p, kp);
#pragma unroll
      for (int e = 0; e < 4; ++e) g[e] *= kp[e];
    }
    if (pre) {
      float q[4];
      if (pre_bf16) {
        const bf16x4 h = reinterpret_cast<const bf16x4*>(pre)[i];
        q[0] = (float)h[0]; q[1] = (float)h[1]; q[2] = (float)h[2]; q[3] = (float)h[3];
      } else {
        const float4 f = reinterpret_cast<const float4*>(pre)[i];
        q[0] = f.x; q[1] = f.y; q[2] = f.z; q[3] = f.w;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) g[e] *= nsp_dact(q[e], act);
    }
    if (out_bf16) {
      bf16x4 h;
      h[0] = (__bf16)g[0]; h[1] = (__bf16)g[1]; h[2] = (__bf16)g[2]; h[3] = (__bf16)g[3];
      reinterpret_cast<bf16x4*>(out)[i] = h;
    } else {
      reinterpret_cast<float4*>(out)[i] = make_float4(g[0], g[1], g[2], g[3]);
    }
  }
}

// grad_prep with the column sums of its OUTPUT (the bias gradient of the Linear whose gradient this is)
// accumulated on the way: a block owns a band of rows; thread t owns the float4 column groups
// t % C4, (+256, ...) for rows r0 + t / C4 (+ 256 / C4, ...), keeps their sums in registers and flushes
// them once (LDS reduce over the row lanes, then ONE slab row per block: colsum [gridDim.x, cols] -- with
// one atomic per column per block, 1024 blocks x 512 columns of contended L2 atomics made the fused
// kernel slower than grad_prep + colsum).  cols % 4 == 0, cols <= 4096.  The caller sums the slab rows
// (a [<=1024, cols] fp32 matrix) instead of passing over the [rows, cols] bf16 image again.
template <int NG>   // column groups per thread = ceil(C4 / 256)
__global__ __launch_bounds__(256) void grad_prep_colsum_kernel(
    const float* __restrict__ dy, const void* __restrict__ pre, int pre_bf16, void* __restrict__ out, int out_bf16,
    int act, float alpha, float p, unsigned long long seed, unsigned long long offset, int rows, int cols,
    int rows_per_block, float* __restrict__ colsum) {
  __shared__ float red[256][4];
  const int C4 = cols >> 2;
  const int lanes_c = C4 < 256 ? C4 : 256;             // threads across the columns
  const int rl = threadIdx.x / lanes_c, nrl = 256 / lanes_c;   // row lane, row lanes per block
  const int tc = threadIdx.x % lanes_c;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  float acc[NG][4];
#pragma unroll
  for (int k = 0; k < NG; ++k) acc[k][0] = acc[k][1] = acc[k][2] = acc[k][3] = 0.f;
  if (rl < nrl) {
    for (int r = r0 + rl; r < r1; r += nrl) {
#pragma unroll
      for (int k = 0; k < NG; ++k) {
        const int c4 = tc + k * 256;
        if (c4 < C4) {
          const long long i = (long long)r * C4 + c4;
          const float4 g4 = reinterpret_cast<const float4*>(dy)[i];
          float g[4] = {g4.x * alpha, g4.y * alpha, g4.z * alpha, g4.w * alpha};
          if (p > 0.f) {
            float kp[4];
            nsp_keep_scale4(seed, offset + (unsigned long long)(i * 4), p, kp);
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] *= kp[e];
          }
          if (pre) {
            float q[4];
            if (pre_bf16) {
              const bf16x4 h = reinterpret_cast<const bf16x4*>(pre)[i];
              q[0] = (float)h[0]; q[1] = (float)h[1]; q[2] = (float)h[2]; q[3] = (float)h[3];
            } else {
              const float4 f = reinterpret_cast<const float4*>(pre)[i];
              q[0] = f.x; q[1] = f.y; q[2] = f.z; q[3] = f.w;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] *= nsp_dact(q[e], act);
          }
          if (out_bf16) {
            bf16x4 h;
            h[0] = (__bf16)g[0]; h[1] = (__bf16)g[1]; h[2] = (__bf16)g[2]; h[3] = (__bf16)g[3];
            reinterpret_cast<bf16x4*>(out)[i] = h;
          } else {
            reinterpret_cast<float4*>(out)[i] = make_float4(g[0], g[1], g[2], g[3]);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[k][e] += g[e];
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < NG; ++k) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) red[threadIdx.x][e] = acc[k][e];
    __syncthreads();
    if (rl == 0) {
      const int c4 = tc + k * 256;
      if (c4 < C4) {
        float sum[4] = {0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < nrl; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) sum[e] += red[j * lanes_c + tc][e];
        *reinterpret_cast<float4*>(colsum + (long long)blockIdx.x * cols + c4 * 4) =
            make_float4(sum[0], sum[1], sum[2], sum[3]);
      }
    }
  }
}

// Round 6: the slabs of a weight gradient (16-62 splits x 1-4 MB) summed by FOUR waves per 64 float4 columns, each wave
// a contiguous quarter of the splits with 8 loads in flight, partials combined through LDS in wave order.  The one-
// thread-per-column kernel below put 256 workgroups on the chip for a 512 x 512 weight (one per CU, 32 KB in flight per
// CU) and ran at 2.7 TB/s: 23 us x 119 launches per step.  Deterministic (fixed order), not the same rounding as the
// sequential sum.
__global__ __launch_bounds__(256) void splitk_reduce4_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                             int splits, long long n) {
  typedef __attribute__((ext_vector_type(4))) float v4f;
  __shared__ v4f red[3][64];
  const int g = threadIdx.x >> 6, c = threadIdx.x & 63;
  const long long i = (long long)blockIdx.x * 64 + c;      // float4 column (the launcher: n / 4 is a multiple of 64)
  const int per = (splits + 3) >> 2;
  const int s0 = g * per, s1 = min(splits, s0 + per);
  v4f a = {0.f, 0.f, 0.f, 0.f};
  int s = s0;
  for (; s + 8 <= s1; s += 8) {
    v4f b[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) b[q] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(part + (long long)(s + q) * n) + i);
#pragma unroll
    for (int q = 0; q < 8; ++q) a += b[q];
  }
  for (; s < s1; ++s) a += __builtin_nontemporal_load(reinterpret_cast<const v4f*>(part + (long long)s * n) + i);
  if (g > 0) red[g - 1][c] = a;
  __syncthreads();
  if (g == 0) {
#pragma unroll
    for (int q = 0; q < 3; ++q) a += red[q][c];
    reinterpret_cast<v4f*>(out)[i] = a;
  }
}

__global__ void splitk_reduce_kernel(const float* __restrict__ part, float* __restrict__ out,
                                     int splits, long long n) {
  const long long n4 = n >> 2;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    // 8 slab loads in flight per thread (the rolled loop issued them one dependent add apart); the order of
    // the additions is still s = 0, 1, 2, ...: bit-identical results
    float4 a = reinterpret_cast<const float4*>(part)[i];
    int s = 1;
    for (; s + 8 <= splits; s += 8) {
      float4 b[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) b[q] = reinterpret_cast<const float4*>(part + (long long)(s + q) * n)[i];
#pragma unroll
      for (int q = 0; q < 8; ++q) { a.x += b[q].x; a.y += b[q].y; a.z += b[q].z; a.w += b[q].w; }
    }
    for (; s < splits; ++s) {
      const float4 b = reinterpret_cast<const float4*>(part + (long long)s * n)[i];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    reinterpret_cast<float4*>(out)[i] = a;
  }
}

// relu backward from the OUTPUT y (y > 0)
__global__ void relu_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                float* __restrict__ dx, long long n) {
  long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    dx[i] = y[i] > 0.f ? dy[i] : 0.f;
}

// column sums: each block reduces a slab of rows for 64 columns (one wave-width),
// 4 waves stride over rows; partials are combined in LDS then atomically added.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x,
                                                     float* __restrict__ out, int rows, int cols,
                                                     long long ld, int rows_per_block) {
  __shared__ float sh[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int r0 = blockIdx.y * rows_per_block;
  const int r1 = min(rows, r0 + rows_per_block);
  float s = 0.f;
  if (c < cols)
    for (int r = r0 + w; r < r1; r += 4) s += (float)x[(long long)r * ld + c];
  sh[w][lane] = s;
  __syncthreads();
  if (w == 0 && c < cols) unsafeAtomicAdd(out + c, sh[0][lane] + sh[1][lane] + sh[2][lane] + sh[3][lane]);
}

// GLU on channels-last rows: x [rows, 2C] -> y [rows, C] = x[:, :C] * sigmoid(x[:, C:])
__global__ void glu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long long rows,
                               int C) {
  const int C4 = C >> 2;
  long long total = rows * C4;
  long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    long long r = i / C4;
    int c = (int)(i % C4);
    float4 a = reinterpret_cast<const float4*>(x + r * 2 * C)[c];
    float4 b = reinterpret_cast<const float4*>(x + r * 2 * C + C)[c];
    reinterpret_cast<float4*>(y + r * C)[c] =
        make_float4(a.x * nsp_sigmoid(b.x), a.y * nsp_sigmoid(b.y), a.z * nsp_sigmoid(b.z),
                    a.w * nsp_sigmoid(b.w));
  }
}

__global__ void glu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                               float* __restrict__ dx, long long rows, int C) {
  const int C4 = C >> 2;
  long long total = rows * C4;
  long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    long long r = i / C4;
    int c = (int)(i % C4);
    float4 a = reinterpret_cast<const float4*>(x + r * 2 * C)[c];
    float4 b = reinterpret_cast<const float4*>(x + r * 2 * C + C)[c];
    float4 g = reinterpret_cast<const float4*>(dy + r * C)[c];
    float4 s = make_float4(nsp_sigmoid(b.x), nsp_sigmoid(b.y), nsp_sigmoid(b.z), nsp_sigmoid(b.w));
    reinterpret_cast<float4*>(dx + r * 2 * C)[c] = make_float4(g.x * s.x, g.y * s.y, g.z * s.z, g.w * s.w);
    reinterpret_cast<float4*>(dx + r * 2 * C + C)[c] =
        make_float4(g.x * a.x * s.x * (1.f - s.x), g.y * a.y * s.y * (1.f - s.y),
                    g.z * a.z * s.z * (1.f - s.z), g.w * a.w * s.w * (1.f - s.w));
  }
}

// ---- GLU on a bf16 image (throughput mode, round 4).  The pointwise conv that feeds the GLU writes its [rows, 2C]
// output as bf16 from the GEMM epilogue; forward reads that image (2 KB per frame instead of 4 KB of fp32) and the
// SAME image is what backward keeps.  Backward writes d x directly as the bf16 operand of the two gradient GEMMs and
// accumulates its column sums (the pointwise conv's bias gradient) on the way -- one slab row per block, as
// grad_prep_colsum_kernel -- where the fp32 path ran glu_bwd (fp32 in / out) and then grad_prep_colsum over its result:
// 16 KB -> 6 KB per frame at C = 512.
__global__ void glu_fwd_b16_kernel(const __bf16* __restrict__ x, float* __restrict__ y, long long rows, int C) {
  const int C8 = C >> 3;
  const long long total = rows * C8;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long long r = i / C8;
    const int c = (int)(i % C8);
    const bf16x8 a = reinterpret_cast<const bf16x8*>(x + r * 2 * C)[c];
    const bf16x8 b = reinterpret_cast<const bf16x8*>(x + r * 2 * C + C)[c];
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (float)a[e] * nsp_sigmoid((float)b[e]);
    float4* yp = reinterpret_cast<float4*>(y + r * C) + 2 * c;
    yp[0] = make_float4(o[0], o[1], o[2], o[3]);
    yp[1] = make_float4(o[4], o[5], o[6], o[7]);
  }
}

// block = a band of rows; thread (row lane rl, column group tc of 4 channels) -- C / 4 <= 256 column groups
__global__ __launch_bounds__(256) void glu_bwd_b16_colsum_kernel(
    const __bf16* __restrict__ x, const float* __restrict__ dy, __bf16* __restrict__ dx, int rows, int C,
    int rows_per_block, float* __restrict__ colsum) {
  __shared__ float red[256][8];
  const int C4 = C >> 2;
  const int rl = threadIdx.x / C4, nrl = 256 / C4;
  const int tc = threadIdx.x % C4;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (rl < nrl) {
    for (int r = r0 + rl; r < r1; r += nrl) {
      const long long rb = (long long)r * 2 * C;
      const bf16x4 a = reinterpret_cast<const bf16x4*>(x + rb)[tc];
      const bf16x4 b = reinterpret_cast<const bf16x4*>(x + rb + C)[tc];
      const float4 g4 = reinterpret_cast<const float4*>(dy + (long long)r * C)[tc];
      const float g[4] = {g4.x, g4.y, g4.z, g4.w};
      bf16x4 da, db;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float sg = nsp_sigmoid((float)b[e]);
        const float va = g[e] * sg, vb = g[e] * (float)a[e] * sg * (1.f - sg);
        da[e] = (__bf16)va; db[e] = (__bf16)vb;
        acc[e] += va; acc[4 + e] += vb;
      }
      reinterpret_cast<bf16x4*>(dx + rb)[tc] = da;
      reinterpret_cast<bf16x4*>(dx + rb + C)[tc] = db;
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
    float* cs = colsum + (long long)blockIdx.x * 2 * C;
    *reinterpret_cast<float4*>(cs + tc * 4) = make_float4(sum[0], sum[1], sum[2], sum[3]);
    *reinterpret_cast<float4*>(cs + C + tc * 4) = make_float4(sum[4], sum[5], sum[6], sum[7]);
  }
}

// y[i] = alpha*x[i] + z[i % period]  (sinusoidal table broadcast over the batch,
// positional_embedding.py:85-88)
__global__ void scale_add_bcast_kernel(const float* __restrict__ x, const float* __restrict__ z,
                                       float* __restrict__ y, float alpha, long long n,
                                       long long period) {
  long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    y[i] = alpha * x[i] + z[i % period];
}

// Transformer-XL style table (positional_embedding.py:131-138):
// out[j][i] = sin(-(j+1) * inv_freq[i]) for i < d/2, cos(...) for the second half
__global__ void xl_pos_table_kernel(const float* __restrict__ inv_freq, float* __restrict__ out,
                                    int L, int d) {
  const int half = d >> 1;
  const long long total = (long long)L * d;
  long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c = (int)(i % d);
    const int j = (int)(i / d);
    const float ang = -(float)(j + 1) * inv_freq[c < half ? c : c - half];
    out[i] = c < half ? sinf(ang) : cosf(ang);
  }
}

__global__ void specaug_kernel(float* __restrict__ x, int B, int T, int F, const int* fb, int nf,
                               const int* tb, int nt) {
  // fb/tb are tiny device arrays of [start,end) pairs
  long long total = (long long)B * T * F;
  long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    int f = (int)(i % F);
    int t = (int)((i / F) % T);
    bool z = false;
    for (int k = 0; k < nf; ++k) z |= (f >= fb[2 * k] && f < fb[2 * k + 1]);
    for (int k = 0; k < nt; ++k) z |= (t >= tb[2 * k] && t < tb[2 * k + 1]);
    if (z) x[i] = 0.f;
  }
}

__global__ void pad_batch_kernel(const float* __restrict__ packed, const long long* __restrict__ offs,
                                 const int* __restrict__ lens, float* __restrict__ out, int B,
                                 int Tmax, int F, float pad) {
  long long total = (long long)B * Tmax * F;
  long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    int f = (int)(i % F);
    int t = (int)((i / F) % Tmax);
    int b = (int)(i / ((long long)F * Tmax));
    out[i] = t < lens[b] ? packed[offs[b] + (long long)t * F + f] : pad;
  }
}

}  // namespace

extern "C" int nsp_axpby(const float* x, const float* z, float* y, float alpha, float beta,
                         long long n, void* stream) {
  if (n <= 0) return NSP_OK;
  hipLaunchKernelGGL(axpby_kernel, dim3(ew_grid(n / 4 + 1)), dim3(EW_THREADS), 0, (hipStream_t)stream,
                     x, z, y, alpha, beta, n);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

extern "C" int nsp_act_fwd(const float* x, float* y, int act, long long n, void* stream) {
  if (n <= 0) return NSP_OK;
  hipLaunchKernelGGL(act_fwd_kernel, dim3(ew_grid(n / 4 + 1)), dim3(EW_THREADS), 0, (hipStream_t)stream,
                     x, y, act, n);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

extern "C" int nsp_dact_mul(const float* dy, const float* pre, float* out, int act, float alpha,
                            long long n, void* stream) {
  if (n <= 0) return NSP_OK;
  hipLaunchKernelGGL(dact_mul_kernel, dim3(ew_grid(n / 4 + 1)), dim3(EW_THREADS), 0,
                     (hipStream_t)stream, dy, pre, out, act, alpha, n);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

extern "C" int nsp_dropout(const float* x, float* y, float p, float alpha, unsigned long long seed,
                           unsigned long long offset, long long n, void* stream) {
  if (n <= 0) return NSP_OK;
  if (p < 0.f || p >= 1.f) return NSP_EINVAL;
  hipLaunchKernelGGL(dropout_kernel, dim3(ew_grid(n)), dim3(EW_THREADS), 0, (hipStream_t)stream, x, y,
                     p, alpha, seed, offset, n);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

extern "C" int nsp_splitk_reduce(const float* part, float* out, int splits, long long n, void* stream) {
  if (n <= 0 || splits < 1) return NSP_OK;
  if (n % 4) return NSP_EUNSUPPORTED;
  if (n % 256 == 0 && splits >= 8 && n / 256 <= 0x7FFFFFFFll)
    hipLaunchKernelGGL(splitk_reduce4_kernel, dim3((unsigned)(n / 256)), dim3(256), 0, (hipStream_t)stream, part, out, splits, n);
  else
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(ew_grid(n / 4)), dim3(EW_THREADS), 0,
                       (hipStream_t)stream, part, out, splits, n);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

extern "C" int nsp_relu_bwd(const float* y, const float* dy, float* dx, long long n, void* stream) {
  if (n <= 0) return NSP_OK;
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(ew_grid(n)), dim3(EW_THREADS), 0, (hipStream_t)stream, y,
                     dy, dx, n);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

extern "C" int nsp_colsum(const float* x, float* out, int rows, int cols, long long ld,
                          int accumulate, void* stream) {
  if (rows <= 0 || cols <= 0) return NSP_OK;
  hipStream_t st = (hipStream_t)stream;
  if (!accumulate) {
    hipError_t e = hipMemsetAsync(out, 0, sizeof(float) * cols, st);
    if (e != hipSuccess) return (int)e;
  }
  int gx = nsp_cdiv(cols, 64);
  int gy = 2048 / gx;
  if (gy < 1) gy = 1;
  int rpb = nsp_cdiv(rows, gy);
  if (rpb < 32) rpb = 32;
  gy = nsp_cdiv(rows, rpb);
  hipLaunchKernelGGL(colsum_kernel<float>, dim3(gx, gy), dim3(256), 0, st, x, out, rows, cols, ld, rpb);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

extern "C" int nsp_colsum_bf16(const void* x, float* out, int rows, int cols, long long ld,
                               int accumulate, void* stream) {
  if (rows <= 0 || cols <= 0) return NSP_OK;
  hipStream_t st = (hipStream_t)stream;
  if (!accumulate) {
    hipError_t e = hipMemsetAsync(out, 0, sizeof(float) * cols, st);
    if (e != hipSuccess) return (int)e;
  }
  int gx = nsp_cdiv(cols, 64);
  int gy = 2048 / gx;
  if (gy < 1) gy = 1;
  int rpb = nsp_cdiv(rows, gy);
  if (rpb < 32) rpb = 32;
  gy = nsp_cdiv(rows, rpb);
  hipLaunchKernelGGL(colsum_kernel<__bf16>, dim3(gx, gy), dim3(256), 0, st,
                     reinterpret_cast<const __bf16*>(x), out, rows, cols, ld, rpb);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

extern "C" int nsp_grad_prep(const float* dy, const void* pre, int pre_bf16, void* out, int out_bf16, int act,
                             float alpha, float p, unsigned long long seed, unsigned long long offset,
                             long long n, int cols, float* colsum, void* stream) {
  if (n <= 0) return NSP_OK;
  if (n % 4) return NSP_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (colsum) {
    // fused bias gradient: colsum = slabs [nsp_grad_prep_slabs(rows), cols] fp32, every row written
    if (cols <= 0 || cols % 4 || cols > 4096 || n % cols) return NSP_EUNSUPPORTED;
    const int rows = (int)(n / cols);
    int rpb = nsp_cdiv(rows, 1024);
    if (rpb < 8) rpb = 8;
    const int grid = nsp_cdiv(rows, rpb);   // == nsp_grad_prep_slabs(rows)
    const int ng = nsp_cdiv(cols / 4, 256);
#define GPCS(NG)                                                                                                  \
    hipLaunchKernelGGL((grad_prep_colsum_kernel<NG>), dim3(grid), dim3(256), 0, st, dy, pre, pre_bf16, out, out_bf16, \
                       act, alpha, p, seed, offset, rows, cols, rpb, colsum)
    if (ng <= 1) GPCS(1); else if (ng == 2) GPCS(2); else GPCS(4);
#undef GPCS
    NSP_LAUNCH_CHECK();
    return NSP_OK;
  }
  hipLaunchKernelGGL(grad_prep_kernel, dim3(ew_grid(n / 4)), dim3(EW_THREADS), 0, st,
                     dy, pre, pre_bf16, out, out_bf16, act, alpha, p, seed, offset, n);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

extern "C" int nsp_grad_prep_slabs(int rows) {
  int rpb = nsp_cdiv(rows, 1024);
  if (rpb < 8) rpb = 8;
  return nsp_cdiv(rows, rpb);
}

extern "C" int nsp_glu_fwd_b16(const void* x16, float* y, long long rows, int C, void* stream) {
  if (rows <= 0) return NSP_OK;
  if (C % 8 || (reinterpret_cast<uintptr_t>(x16) & 15) || (reinterpret_cast<uintptr_t>(y) & 15)) return NSP_EUNSUPPORTED;
  hipLaunchKernelGGL(glu_fwd_b16_kernel, dim3(ew_grid(rows * (C / 8))), dim3(EW_THREADS), 0, (hipStream_t)stream,
                     reinterpret_cast<const __bf16*>(x16), y, rows, C);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

extern "C" int nsp_glu_bwd_b16(const void* x16, const float* dy, void* dx16, float* colsum_slabs, int rows, int C,
                               void* stream) {
  if (rows <= 0) return NSP_OK;
  if (C % 4 || C / 4 > 256 || 256 % (C / 4) || !colsum_slabs) return NSP_EUNSUPPORTED;
  int rpb = nsp_cdiv(rows, 1024);
  if (rpb < 8) rpb = 8;
  hipLaunchKernelGGL(glu_bwd_b16_colsum_kernel, dim3(nsp_cdiv(rows, rpb)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const __bf16*>(x16), dy, reinterpret_cast<__bf16*>(dx16), rows, C, rpb, colsum_slabs);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

extern "C" int nsp_glu_fwd(const float* x, float* y, long long rows, int C, void* stream) {
  if (C % 4) return NSP_EUNSUPPORTED;
  hipLaunchKernelGGL(glu_fwd_kernel, dim3(ew_grid(rows * (C / 4))), dim3(EW_THREADS), 0,
                     (hipStream_t)stream, x, y, rows, C);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

extern "C" int nsp_glu_bwd(const float* x, const float* dy, float* dx, long long rows, int C,
                           void* stream) {
  if (C % 4) return NSP_EUNSUPPORTED;
  hipLaunchKernelGGL(glu_bwd_kernel, dim3(ew_grid(rows * (C / 4))), dim3(EW_THREADS), 0,
                     (hipStream_t)stream, x, dy, dx, rows, C);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

extern "C" int nsp_scale_add_bcast(const float* x, const float* z, float* y, float alpha,
                                   long long n, long long period, void* stream) {
  if (n <= 0 || period <= 0) return NSP_EINVAL;
  hipLaunchKernelGGL(scale_add_bcast_kernel, dim3(ew_grid(n)), dim3(EW_THREADS), 0,
                     (hipStream_t)stream, x, z, y, alpha, n, period);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

extern "C" int nsp_xl_pos_table(const float* inv_freq, float* out, int L, int d, void* stream) {
  if (L <= 0 || d <= 0 || (d & 1)) return NSP_EINVAL;
  hipLaunchKernelGGL(xl_pos_table_kernel, dim3(ew_grid((long long)L * d)), dim3(EW_THREADS), 0,
                     (hipStream_t)stream, inv_freq, out, L, d);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

extern "C" int nsp_specaug_apply(float* x, int B, int T, int F, const int* freq_bands, int n_freq,
                                 const int* time_bands, int n_time, void* stream) {
  // bands arrive as DEVICE int32 arrays of [start,end) pairs (the host draws them)
  if (n_freq + n_time == 0) return NSP_OK;
  hipLaunchKernelGGL(specaug_kernel, dim3(ew_grid((long long)B * T * F)), dim3(EW_THREADS), 0,
                     (hipStream_t)stream, x, B, T, F, freq_bands, n_freq, time_bands, n_time);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

extern "C" int nsp_pad_batch(const float* packed, const long long* offsets, const int* lens,
                             float* out, int B, int Tmax, int F, float pad_value, void* stream) {
  hipLaunchKernelGGL(pad_batch_kernel, dim3(ew_grid((long long)B * Tmax * F)), dim3(EW_THREADS), 0,
                     (hipStream_t)stream, packed, offsets, lens, out, B, Tmax, F, pad_value);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

// ---- one launch that refreshes EVERY bf16 weight shadow of the model (plain casts, transposed casts, pieces of stacked /
// concatenated shadows).  After an optimizer step all ~190 shadows of Conformer-L + RNN-T are stale; rebuilding them one
// by one cost ~190 launches (cast_bf16 x 90, transposing copies x 49, torch.cat ...) = ~3 ms of HOST time per step, which
// is what bounds the step at 16 utterances per GPU (26 ms of enqueue time for ~1500 launches, tools/b16_probe.py).
// table: n rows of 8 x int64 {src (fp32), dst (bf16), rows, cols, src_ld, dst_ld, transpose, first tile}; dst(r, c) =
// transpose ? src[c * src_ld + r] : src[r * src_ld + c] for r < rows, c < cols; 64 x 64 tiles, one per workgroup.
namespace {
__global__ __launch_bounds__(256) void shadow_refresh_kernel(const long long* __restrict__ table, int n) {
  __shared__ float tile[64][65];
  // entry of this workgroup: the last one whose first tile is <= blockIdx.x
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[mid * 8 + 7] <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const long long* e = table + lo * 8;
  const float* src = reinterpret_cast<const float*>(e[0]);
  __bf16* dst = reinterpret_cast<__bf16*>(e[1]);
  const int rows = (int)e[2], cols = (int)e[3];
  const long long src_ld = e[4], dst_ld = e[5];
  const bool tr = e[6] != 0;
  const int t = blockIdx.x - (int)e[7];
  const int tiles_c = (cols + 63) >> 6;
  const int r0 = (t / tiles_c) * 64, c0 = (t % tiles_c) * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  if (!tr) {
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
      const int r = r0 + i * 4 + ty, c = c0 + tx;
      if (r < rows && c < cols) dst[(long long)r * dst_ld + c] = (__bf16)src[(long long)r * src_ld + c];
    }
  } else {
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {        // read along the source's contiguous index (= dst row index r)
      const int c = c0 + i * 4 + ty, r = r0 + tx;
      tile[i * 4 + ty][tx] = (r < rows && c < cols) ? src[(long long)c * src_ld + r] : 0.f;
    }
    __syncthreads();
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
      const int r = r0 + i * 4 + ty, c = c0 + tx;
      if (r < rows && c < cols) dst[(long long)r * dst_ld + c] = (__bf16)tile[tx][i * 4 + ty];
    }
  }
}
}  // namespace

extern "C" int nsp_shadow_refresh(const long long* table, int n_entries, int total_tiles, void* stream) {
  if (n_entries <= 0 || total_tiles <= 0) return NSP_OK;
  if (!table) return NSP_EINVAL;
  hipLaunchKernelGGL(shadow_refresh_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, table, n_entries);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}


// ---- test hook: workgroups that only occupy compute units (see include/nsp_hip.h)
namespace {
__global__ __launch_bounds__(1024) void occupy_kernel(long long cycles) {
  __shared__ unsigned char hold[65536];      // + 64 KB of LDS per workgroup: at most two of them share a CU
  const long long t0 = (long long)__builtin_readcyclecounter();
  while ((long long)__builtin_readcyclecounter() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
  if (cycles < 0) hold[threadIdx.x] = 1;     // (keeps the array)
}
}  // namespace
extern "C" int nsp_debug_occupy(int n_wg, long long cycles, void* stream) {
  if (n_wg <= 0) return NSP_OK;
  hipLaunchKernelGGL(occupy_kernel, dim3(n_wg), dim3(1024), 0, (hipStream_t)stream, cycles);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

// ---- test hook: a hostile neighbour (see include/nsp_hip.h).  Short-lived workgroups that overwrite their LDS allocation,
// ~96 VGPRs and ~96 AGPRs with `pattern` and exit: whatever kernel gets the CU slot next finds that pattern in every LDS
// byte / register it reads before writing.  `rounds` LDS sweeps per workgroup also load the CU's LDS pipe.
namespace {
__global__ __launch_bounds__(256) void scribble_kernel(unsigned int pattern, int lds_dwords, int rounds) {
  extern __shared__ __attribute__((aligned(16))) unsigned char scr[];
  unsigned int* w = reinterpret_cast<unsigned int*>(scr);
  for (int r = 0; r < rounds; ++r) {
    for (int i = threadIdx.x; i < lds_dwords; i += 256) w[i] = pattern + (r == rounds ? 1u : 0u);
    __syncthreads();
  }
#ifndef NSP_HOST_EMULATION
  unsigned int v[96], a[96];
#pragma unroll
  for (int i = 0; i < 96; ++i) asm volatile("v_mov_b32 %0, %1" : "=v"(v[i]) : "v"(pattern));
#pragma unroll
  for (int i = 0; i < 96; ++i) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(a[i]) : "v"(pattern));
#pragma unroll
  for (int i = 0; i < 96; ++i) asm volatile("" :: "v"(v[i]), "a"(a[i]));
#endif
}
}  // namespace
extern "C" int nsp_debug_scribble(int n_wg, int lds_bytes, unsigned int pattern, int rounds, void* stream) {
  if (n_wg <= 0) return NSP_OK;
  if (lds_bytes < 0 || lds_bytes > 65536 || (lds_bytes & 3)) return NSP_EINVAL;
  hipLaunchKernelGGL(scribble_kernel, dim3(n_wg), dim3(256), (size_t)lds_bytes, (hipStream_t)stream, pattern,
                     lds_bytes / 4, rounds);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}
