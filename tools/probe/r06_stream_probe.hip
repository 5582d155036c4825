// Round-6 probe: what bounds the K / V (Q / dO) tile stream of the attention kernels -- DMA latency x bytes in flight
// (Little's law) or an L2 -> LDS bandwidth ceiling?  (hipcc --offload-arch=gfx950 -O3 r06_stream_probe.hip -o r06_stream_probe)
//
// Grid = NBH x 13 workgroups of 256 threads, exactly the attention kernels' decomposition at T = 800: workgroup (bh, j)
// streams ALL 13 tiles (64 rows x 128 B from each of two arrays with a 3072-B row pitch, i.e. the K | V column blocks of
// qkv[B*T, 3*512] bf16) of utterance-head bh through an LDS ring of DEPTH slots, `ahead` = DEPTH - 1 tiles requested
// before the one being consumed; "consume" = every wave reads the tile once from LDS (ds_read_b128) and folds it into a
// checksum.  Workgroups per CU are set by padding the dynamic LDS allocation.  MODE 0: global_load_lds_dwordx4 (LDS-DMA),
// MODE 1: global_load_dwordx4 -> registers -> ds_write_b128 (register staged), MODE 2: no LDS at all, registers only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr int T = 800, NT = 13, PITCH = 3072;   // bytes per row of the [B*T, 3 d] bf16 matrix

template <int MODE, int DEPTH>
__global__ __launch_bounds__(256) void stream_kernel(const unsigned char* __restrict__ src, unsigned* __restrict__ out, int H) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ring[];   // DEPTH x 16 KB (+ padding that sets the occupancy)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = blockIdx.x % H, b = blockIdx.y;
  const unsigned char* base = src + (size_t)b * T * PITCH + 1024 + h * 128;    // K block of head h
  auto issue = [&](int tile, int slot) {
    unsigned char* dst = ring + slot * 16384;
#pragma unroll
    for (int a = 0; a < 2; ++a)          // two arrays (K, V): column blocks 1024 B apart
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ii = wave * 2 + i;
        const int row = min(tile * 64 + ii * 8 + (lane >> 3), T - 1), c = lane & 7;
        const unsigned char* g = base + (size_t)row * PITCH + a * 1024 + ((c ^ (row & 7)) << 4);
        if (MODE == 0) __builtin_amdgcn_global_load_lds((glb_void*)g, (lds_void*)(dst + a * 8192 + ii * 1024), 16, 0, 0);
      }
  };
  unsigned acc = 0;
  if (MODE == 0) {
#pragma unroll
    for (int t = 0; t < DEPTH - 1; ++t) issue(t, t);
    for (int t = 0; t < NT; ++t) {
      if (t + DEPTH - 1 < NT) issue(t + DEPTH - 1, (t + DEPTH - 1) % DEPTH);
      // wait for tile t: everything but the (DEPTH - 1) younger tiles' 4 loads each (fewer at the tail)
      const int younger = min(DEPTH - 1, NT - 1 - t) * 4;
      if (younger >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else if (younger >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (younger >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      const unsigned char* cur = ring + (t % DEPTH) * 16384;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(cur + ((i * 64 + lane) << 4) + (wave & 1) * 4096 * 0);
        acc += v[0] ^ v[1] ^ v[2] ^ v[3];
      }
      __syncthreads();
    }
  } else {
    // register staged: one tile ahead in registers (4 x 16 B per thread), written to LDS after the barrier
    u32x4 st[4];
    auto load = [&](int tile) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int ii = wave * 2 + i;
          const int row = min(tile * 64 + ii * 8 + (lane >> 3), T - 1), c = lane & 7;
          st[a * 2 + i] = *reinterpret_cast<const u32x4*>(base + (size_t)row * PITCH + a * 1024 + (c << 4));
        }
    };
    load(0);
    for (int t = 0; t < NT; ++t) {
      unsigned char* cur = ring + (t & 1) * 16384;
      if (MODE == 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<u32x4*>(cur + (q >> 1) * 8192 + (wave * 2 + (q & 1)) * 1024 + (lane << 4)) = st[q];
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) acc += st[q][0] ^ st[q][1] ^ st[q][2] ^ st[q][3];
      }
      if (t + 1 < NT) load(t + 1);
      if (MODE == 1) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const u32x4 v = *reinterpret_cast<const u32x4*>(cur + ((i * 64 + lane) << 4));
          acc += v[0] ^ v[1] ^ v[2] ^ v[3];
        }
      }
    }
  }
  out[(blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x] = acc;
}

template <int MODE, int DEPTH>
void run(const unsigned char* src, unsigned* out, int B, int H, int wg_per_cu, const char* name) {
  // LDS per workgroup so that exactly wg_per_cu fit into 160 KB (never below what the ring needs)
  size_t lds = 160 * 1024 / wg_per_cu;
  lds = lds / 1024 * 1024;
  const size_t need = MODE == 2 ? 0 : (MODE == 0 ? DEPTH : 2) * 16384;
  if (lds < need) { printf("%-22s depth %d wg/CU %d: ring does not fit\n", name, DEPTH, wg_per_cu); return; }
  CK(hipFuncSetAttribute((const void*)stream_kernel<MODE, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  dim3 grid(NT * H, B);
  hipLaunchKernelGGL((stream_kernel<MODE, DEPTH>), grid, dim3(256), lds, 0, src, out, H);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((stream_kernel<MODE, DEPTH>), grid, dim3(256), lds, 0, src, out, H);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
  const double bytes = (double)B * H * NT * NT * 16384.0;
  printf("%-22s depth %d  wg/CU %d (%3zu KB in flight per CU): %8.1f us  %6.2f TB/s into the CUs  %6.1f GB/s per CU\n", name, DEPTH, wg_per_cu,
         (size_t)(MODE == 0 ? DEPTH - 1 : 1) * 16 * wg_per_cu, ms * 1e3, bytes / ms / 1e9, bytes / ms / 1e6 / 256);
}

int main() {
  const int B = 128, H = 8;
  unsigned char* src; unsigned* out;
  const size_t bytes = (size_t)B * T * PITCH;
  CK(hipMalloc(&src, bytes));
  CK(hipMemset(src, 1, bytes));
  CK(hipMalloc(&out, (size_t)B * H * NT * 256 * 4));
  printf("== K / V tile stream at the attention kernels' decomposition (B = 128, H = 8, T = 800: 2.8 GB through the CUs per launch)\n");
  for (int w : {1, 2, 3, 4, 5}) run<0, 2>(src, out, B, H, w, "LDS-DMA");
  for (int w : {1, 2, 3}) run<0, 3>(src, out, B, H, w, "LDS-DMA");
  for (int w : {1, 2}) run<0, 4>(src, out, B, H, w, "LDS-DMA");
  for (int w : {1, 2, 3, 4, 5}) run<1, 2>(src, out, B, H, w, "register staged");
  for (int w : {1, 2, 3, 4, 5, 8}) run<2, 2>(src, out, B, H, w, "registers only");
  return 0;
}
