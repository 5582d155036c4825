// Probe ds_read_b64_tr_b16 lane mapping and global_load_lds behaviour on gfx950.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;

extern "C" __global__ void tr_probe(unsigned short* out, int mode) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = (unsigned short)i;
  __syncthreads();
  int lane = threadIdx.x;
  unsigned addr;
  if (mode == 0) addr = lane * 8;                       // lane reads elems 4l..4l+3 (plain layout)
  else if (mode == 1) addr = (lane & 15) * 2 + (lane >> 4) * 128;  // guide formula guess: col (l&15), group offset 64 elems
  else addr = (lane & 15) * 64 * 2 + (lane >> 4) * 8;    // row-per-lane, pitch 64 elems
  unsigned base = (unsigned)(uintptr_t)lds;  // LDS address (low 32 bits of the shared pointer)
  addr += base;
  u16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = v[j];
}

// global_load_lds 16B: each lane supplies a global address; LDS dest = base + lane*16 ?
extern "C" __global__ void glds_probe(const unsigned* src, unsigned* out) {
  __shared__ __attribute__((aligned(16))) unsigned lds[1024];
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = 0xdeadbeef;
  __syncthreads();
  int lane = threadIdx.x;
  // lane l loads 16 B from src + (63-l)*4 dwords (reversed) so we can see where it lands
  const unsigned* g = src + (63 - lane) * 4;
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) out[i] = lds[i];
}
