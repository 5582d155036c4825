// xent.hip -- label-smoothed cross entropy of the attention decoders (criterion.py:45-86
// `cross_entropy_lsm`, called at decoders/transformer.py:442 and las.py:735) fused with its gradient
// and the teacher-forcing accuracy (torch_utils.py:129-145 `compute_accuracy`).
//
// One wave per target position (row of V logits; V = 10k for the LibriSpeech recipes): pass 1 = row
// max + arg-max, pass 2 (the row is L2 / L1 resident by then) = sum exp, sum of logits -> per-row
//   loss_row = -[(1-e) (x_y - lse) + e/(V-1) (sum_v x_v - x_y - (V-1) lse)]      (e = lsm_prob)
//   grad[v]  = scale * (softmax_v - target_v),  target = (1-e) at y, e/(V-1) elsewhere
// rows whose label is `ignore_index` contribute nothing (loss 0, grad 0).  The reference builds the
// dense [B*L, V] target distribution, a log_softmax copy and their product: four V-wide passes.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void xe_lsm_kernel(const float* __restrict__ logits, const int* __restrict__ ys,
                                                     float* __restrict__ loss_rows, int* __restrict__ correct,
                                                     float* __restrict__ grad, long long rows, int V, int ignore_index,
                                                     float lsm, float scale) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const float off_t = lsm / (float)(V - 1), on_t = 1.f - lsm;
  for (long long row = (long long)blockIdx.x * 4 + w; row < rows; row += (long long)gridDim.x * 4) {
    const float* x = logits + row * V;
    float* g = grad ? grad + row * V : nullptr;
    const int y = ys[row];
    if (y == ignore_index) {
      if (g) for (int v = lane; v < V; v += 64) g[v] = 0.f;
      if (lane == 0) { loss_rows[row] = 0.f; correct[row] = 0; }
      continue;
    }
    float mx = -FLT_MAX;
    int am = 0x7fffffff;
    for (int v = lane; v < V; v += 64) {
      const float t = x[v];
      if (t > mx || (t == mx && v < am)) { mx = t; am = v; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float om = __shfl_xor(mx, o);
      const int oa = __shfl_xor(am, o);
      if (om > mx || (om == mx && oa < am)) { mx = om; am = oa; }
    }
    float se = 0.f, sx = 0.f;
    for (int v = lane; v < V; v += 64) {
      const float t = x[v];
      se += __expf(t - mx);
      sx += t;
    }
    se = wave_reduce_sum(se);
    sx = wave_reduce_sum(sx);
    const float lse = mx + logf(se);
    const float xy = x[y];
    if (lane == 0) {
      loss_rows[row] = -(on_t * (xy - lse) + off_t * (sx - xy - (float)(V - 1) * lse));
      correct[row] = am == y ? 1 : 0;
    }
    if (g) {
      for (int v = lane; v < V; v += 64) {
        const float pv = __expf(x[v] - lse);
        g[v] = scale * (pv - (v == y ? on_t : off_t));
      }
    }
  }
}

}  // namespace

extern "C" int nsp_xe_lsm_fwd_bwd(const float* logits, const int* ys, float* loss_rows, int* correct, float* grad,
                                  long long rows, int V, int ignore_index, float lsm_prob, float grad_scale,
                                  void* stream) {
  if (rows <= 0) return NSP_OK;
  if (V < 2 || !logits || !ys || !loss_rows || !correct) return NSP_EINVAL;
  long long g = (rows + 3) / 4;
  if (g > 256 * 32) g = 256 * 32;
  hipLaunchKernelGGL(xe_lsm_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, logits, ys, loss_rows, correct,
                     grad, rows, V, ignore_index, lsm_prob, grad_scale);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}
