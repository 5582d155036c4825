// decode.hip -- greedy decoding helpers for validate() (train.py:341 -> evaluators -> Speech2Text.decode):
//   * row arg-max of a [rows, V] score matrix (CTC best path, ctc.py:229-230; RNN-T per-frame
//     1-best, rnn_transducer.py:363-364): one wave per row, first index wins ties like torch.argmax;
//   * one masked LSTM cell update for a batch of prediction-network states
//     (rnn_transducer.py:369-370: the state advances only where a non-blank label was emitted).
// Both are tiny and latency-bound; they exist so that the decode loop has no host synchronisation
// per frame (the reference calls .item() once per frame and utterance).
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ x, int* __restrict__ out,
                                                          long long rows, int cols, long long ld) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * ld;
  float best = -FLT_MAX;
  int bi = 0x7fffffff;
  for (int c = lane; c < cols; c += 64) {
    const float v = xr[c];
    // NaN never wins unless the whole row is NaN (then index 0, below)
    if (v > best || (v == best && c < bi)) { best = v; bi = c; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o);
    const int oi = __shfl_xor(bi, o);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if (lane == 0) out[row] = bi == 0x7fffffff ? 0 : bi;
}

// gates: pre-activations [B,4H] in PyTorch order (i,f,g,o), already = x W_ih^T + b_ih + h W_hh^T + b_hh
__global__ void lstm_cell_step_kernel(const float* __restrict__ gates, const float* __restrict__ h_prev,
                                      const float* __restrict__ c_prev, const int* __restrict__ update,
                                      float* __restrict__ h_out, float* __restrict__ c_out, int B, int H) {
  const long long n = (long long)B * H;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int b = (int)(i / H), u = (int)(i % H);
    if (update && update[b] == 0) {
      h_out[i] = h_prev[i];
      c_out[i] = c_prev[i];
      continue;
    }
    const float* g = gates + (long long)b * 4 * H;
    const float ig = nsp_sigmoid(g[u]), fg = nsp_sigmoid(g[H + u]);
    const float gg = nsp_tanh(g[2 * H + u]), og = nsp_sigmoid(g[3 * H + u]);
    const float c = fg * c_prev[i] + ig * gg;
    c_out[i] = c;
    h_out[i] = og * nsp_tanh(c);
  }
}

}  // namespace

extern "C" int nsp_argmax_rows(const float* x, int* out, long long rows, int cols, long long ld, void* stream) {
  if (rows <= 0) return NSP_OK;
  if (cols <= 0 || ld < cols) return NSP_EINVAL;
  hipLaunchKernelGGL(argmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, out,
                     rows, cols, ld);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

extern "C" int nsp_lstm_cell_step(const float* gates, const float* h_prev, const float* c_prev, const int* update,
                                  float* h_out, float* c_out, int B, int H, void* stream) {
  if (B <= 0 || H <= 0) return NSP_OK;
  long long n = (long long)B * H;
  int g = (int)((n + 255) / 256);
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(lstm_cell_step_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, gates, h_prev, c_prev, update,
                     h_out, c_out, B, H);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}
