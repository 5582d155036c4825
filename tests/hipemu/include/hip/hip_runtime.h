// hip/hip_runtime.h -- HOST EMULATION of the HIP kernel language.  TEST INFRASTRUCTURE ONLY.
//
// There is no GPU in the build container, so a kernel written without one cannot be run before the
// round-end GPU job.  This shim lets the *same* .hip source be compiled by the host clang++ (x86) and
// executed on CPU threads, so that `-m "not gpu"` tests can check its indexing, reductions and
// arithmetic against torch.  It is put first on the include path by tests/hipemu/build_emu.py; the
// product build (neural_sp_amd/_lib.py, hipcc --offload-arch=gfx950) never sees this directory, and
// nothing under neural_sp_amd/ loads the library it produces.
//
// Model: one launch = one host thread per WAVE; the 64 lanes of a wave are ucontext fibers scheduled round-robin by
// that thread and switched only at collectives, so wave-level operations (shuffles, MFMA, wave barriers) cost a
// few user-space context switches and no system call.  Each wave thread walks the grid's blocks in order (blocks are
// serialised, so `__shared__` == function-local static storage is private to the running block); __syncthreads()
// = every lane of the wave arrives, the last one joins a pthread barrier over the block's waves, then all are
// released.  Lanes that have returned no longer take part in collectives (as on the hardware); a whole wave must not
// exit while other waves of its block still wait at a __syncthreads().
// Dynamic shared memory: `extern __shared__ T name[];` is rewritten by tests/hipemu/build_emu.py into a pointer to a
// per-launch buffer (HIPEMU_DYN_SHARED).  Emulated as wave collectives: the fp32 and bf16 MFMA builtins and the
// transposed LDS read (ds_read_b64_tr_b16); raw buffer loads / stores keep their range check; LDS-DMA
// (global_load_lds) lands at the s_waitcnt vmcnt(n) that retires it (build_emu.py turns that asm into hipemu_waitcnt_vm(n));
// lgkmcnt waits and scheduling barriers are no-ops.
// Not emulated: other inline asm, streams, cooperative launches (the occupancy query fails, launchers fall back).
#pragma once
#define NSP_HOST_EMULATION 1   /* csrc/common.h: register-pinning asm statements become no-ops */
#include <float.h>
#include <math.h>
#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <ucontext.h>
// Lane switches: glibc's swapcontext saves / restores the signal mask with a system call on EVERY switch (two per lane
// per collective); on x86-64 a 14-instruction switch of the callee-saved registers and the stack pointer
// (hipemu_switch, emu_runtime.cpp) replaces it -- 3-4x faster emulation.  -DHIPEMU_UCONTEXT keeps ucontext (other
// hosts, and the AddressSanitizer build, whose runtime knows swapcontext).
#if !defined(__x86_64__) && !defined(HIPEMU_UCONTEXT)
#define HIPEMU_UCONTEXT 1
#endif
#ifndef HIPEMU_UCONTEXT
extern "C" void hipemu_switch(void** save_sp, void* load_sp);
#endif

#include <thread>
#include <type_traits>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
typedef void* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0 };
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) {
  memset(p, v, n);
  return hipSuccess;
}
// persistent (grid-barrier) kernels need all blocks resident at once; the emulator serialises blocks, so the
// occupancy query fails and the launchers refuse the persistent path (callers fall back to per-step kernels)
enum { hipDeviceAttributeMultiprocessorCount = 63 };
static inline hipError_t hipGetDevice(int*) { return 100; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n); return *p ? hipSuccess : 2; }   // (the GEMM's stream-K workspace)
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipDeviceGetAttribute(int*, int, int) { return 100; }
template <class K> static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int*, K, int, size_t) { return 100; }
#define __HIP_MEMORY_SCOPE_AGENT 0
#define __HIP_MEMORY_SCOPE_SYSTEM 0
#define __hip_atomic_fetch_add(p, v, order, scope) __atomic_fetch_add((p), (v), (order))
#define __hip_atomic_load(p, order, scope) __atomic_load_n((p), (order))
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n((p), (v), (order))
#define __builtin_amdgcn_s_sleep(n) sched_yield()
#define __builtin_amdgcn_fence(order, scope) __atomic_thread_fence(__ATOMIC_SEQ_CST)
template <class T> static inline T min(T a, T b) { return b < a ? b : a; }
template <class T> static inline T max(T a, T b) { return a < b ? b : a; }
static inline long long min(long long a, int b) { return a < b ? a : b; }
static inline long long min(int a, long long b) { return a < b ? a : b; }
static inline long long max(long long a, int b) { return a > b ? a : b; }
static inline long long max(int a, long long b) { return a > b ? a : b; }

struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(16) int4 { int x, y, z, w; };
struct int2 { int x, y; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
struct alignas(16) uint4 { unsigned x, y, z, w; };
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
static inline unsigned __umul24(unsigned a, unsigned b) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu); }

#define __expf(x) expf(x)
#define __logf(x) logf(x)
#define __fdividef(a, b) ((a) / (b))
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }

namespace hipemu {
struct Launch;
struct Wave {              // one per host thread
  Launch* L;
  int index, n;            // wave number inside the block, lanes in it
  int cur, alive, arrived; // running lane; lanes not yet returned; lanes waiting at the current collective
  unsigned gen;            // collective generation (a waiting lane resumes when it changes)
#ifdef HIPEMU_UCONTEXT
  ucontext_t sched;
  ucontext_t ctx[64];
#else
  void* sched;             // saved stack pointers (hipemu_switch)
  void* ctx[64];
#endif
  bool done[64];
  char* stacks;            // 64 fiber stacks
  void (*invoke)(void*);
  void* closure;
  uint64_t slot[64];
  alignas(16) unsigned char frag[64][32];   // MFMA operand fragments: A (16 B) | B (16 B) per lane
  // LDS-DMA (global_load_lds) in flight, per lane, oldest first: the LDS write lands only when a s_waitcnt vmcnt(n)
  // retires it (or at a __syncthreads, whose fence waits for everything) -- the latest moment the hardware allows
  struct Dma { char* dst; int size; unsigned char data[16]; };
  std::vector<Dma> dma[64];
};
struct Launch {
  dim3 grid, block;
  pthread_barrier_t bar;   // over the block's waves
  std::vector<double> dyn; // dynamic shared memory of the running block (8-byte storage: any alignment <= 16 via offset)
};
extern thread_local Launch* cur;
extern thread_local Wave* wave;
extern thread_local int tid_flat;
static const size_t kFiberStack = 256 * 1024;
static inline void yield_lane() {
  Wave* w = wave;
#ifdef HIPEMU_UCONTEXT
  swapcontext(&w->ctx[w->cur], &w->sched);
#else
  hipemu_switch(&w->ctx[w->cur], w->sched);
#endif
}
// every live lane of the wave arrives; `at_release` (may be null) runs once, on the last arriver, before the release
template <class F>
static inline void wave_collective(F at_release) {
  Wave* w = wave;
  const unsigned g = w->gen;
  if (++w->arrived >= w->alive) {
    at_release();
    w->arrived = 0;
    ++w->gen;
  } else {
    while (w->gen == g) yield_lane();
  }
}
static inline void wave_barrier() { wave_collective([] {}); }
}  // namespace hipemu

extern thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;

static inline void __threadfence_block() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline float __int_as_float(int v) { float f; memcpy(&f, &v, 4); return f; }
static inline int __float_as_int(float f) { int v; memcpy(&v, &f, 4); return v; }
static inline void hipemu_waitcnt_vm(int n) {     // s_waitcnt vmcnt(n): all but the n most recent LDS-DMA loads of this lane land
  std::vector<hipemu::Wave::Dma>& q = hipemu::wave->dma[hipemu::tid_flat & 63];
  const int retire = (int)q.size() - n;
  if (retire <= 0) return;
  for (int i = 0; i < retire; ++i) if (q[i].size) memcpy(q[i].dst, q[i].data, (size_t)q[i].size);
  q.erase(q.begin(), q.begin() + retire);
}
static inline void hipemu_raw_barrier() {         // s_barrier alone: no memory wait
  hipemu::wave_collective([] { pthread_barrier_wait(&hipemu::cur->bar); });
}
static inline void __syncthreads() {              // HIP's __syncthreads = s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier
  hipemu_waitcnt_vm(0);
  hipemu_raw_barrier();
}
// 16-byte aligned base of the launch's dynamic shared memory
#define HIPEMU_DYN_SHARED \
  (reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(hipemu::cur->dyn.data()) + 15) & ~uintptr_t(15)))

template <class T>
static inline T hipemu_shfl(T v, int src_of_lane_fn(int, int), int arg) {
  static_assert(sizeof(T) <= 8, "shuffle of <= 8-byte values only");
  hipemu::Wave& w = *hipemu::wave;
  const int lane = hipemu::tid_flat & 63;
  uint64_t raw = 0;
  memcpy(&raw, &v, sizeof(T));
  w.slot[lane] = raw;
  hipemu::wave_barrier();
  int src = src_of_lane_fn(lane, arg);
  if (src < 0 || src >= w.n) src = lane;  // inactive source lane: own value
  raw = w.slot[src];
  hipemu::wave_barrier();
  T r;
  memcpy(&r, &raw, sizeof(T));
  return r;
}
static inline int hipemu_src_xor(int lane, int m) { return lane ^ m; }
static inline int hipemu_src_down(int lane, int d) { return lane + d; }
static inline int hipemu_src_up(int lane, int d) { return lane - d; }
static inline int hipemu_src_idx(int, int i) { return i; }
template <class T> static inline T __shfl_xor(T v, int m, int = 64) { return hipemu_shfl(v, hipemu_src_xor, m); }
template <class T> static inline T __shfl_down(T v, int d, int = 64) { return hipemu_shfl(v, hipemu_src_down, d); }
template <class T> static inline T __shfl_up(T v, int d, int = 64) { return hipemu_shfl(v, hipemu_src_up, d); }
template <class T> static inline T __shfl(T v, int i, int = 64) { return hipemu_shfl(v, hipemu_src_idx, i); }
// v_permlane16_swap / v_permlane32_swap (vdst, src): odd 16-lane rows (upper 32-lane half) of vdst <-> even rows (lower
// half) of src; returns {vdst', src'}
struct hipemu_swap_pair { unsigned v[2]; unsigned operator[](int i) const { return v[i]; } };
static inline int hipemu_src_same(int lane, int) { return lane; }
static inline hipemu_swap_pair hipemu_permlane_swap(unsigned a, unsigned b, int width) {
  const int lane = hipemu::tid_flat & 63;
  const bool upper = (lane / width) & 1;
  // lane in an upper block of vdst receives src's lane - width; lane in a lower block of src receives vdst's lane + width
  const unsigned b_from_lo = hipemu_shfl(b, hipemu_src_up, width);     // value of b at lane - width
  const unsigned a_from_hi = hipemu_shfl(a, hipemu_src_down, width);   // value of a at lane + width
  hipemu_swap_pair r;
  r.v[0] = upper ? b_from_lo : a;
  r.v[1] = upper ? b : a_from_hi;
  return r;
}
static inline hipemu_swap_pair __builtin_amdgcn_permlane16_swap(unsigned a, unsigned b, bool, bool) { return hipemu_permlane_swap(a, b, 16); }
static inline hipemu_swap_pair __builtin_amdgcn_permlane32_swap(unsigned a, unsigned b, bool, bool) { return hipemu_permlane_swap(a, b, 32); }
// DPP quad_perm (dpp_ctrl < 0x100: lane l reads lane (l & ~3) | ((ctrl >> 2 (l & 3)) & 3)); all rows / banks enabled
// + row_mirror (0x140: lane l reads lane 15 - l of its 16-lane row) and row_half_mirror (0x141: 7 - l within 8 lanes)
static inline int hipemu_src_quad(int lane, int ctrl) {
  if (ctrl == 0x140) return (lane & ~15) | (15 - (lane & 15));
  if (ctrl == 0x141) return (lane & ~7) | (7 - (lane & 7));
  return (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3);
}
static inline int __builtin_amdgcn_update_dpp(int /*old*/, int src, int ctrl, int /*row_mask*/, int /*bank_mask*/, bool /*bound_ctrl*/) {
  return hipemu_shfl(src, hipemu_src_quad, ctrl);
}

// ---- MFMA as a wave collective (register layouts: cdna_hip_programming.md, "Fragment layout") ----
//   v_mfma_f32_16x16x4_f32 : lane l holds A[l&15][k=l>>4], B[k=l>>4][l&15]; D/C reg r: row (l>>4)*4+r, col l&15
//   v_mfma_f32_16x16x32_bf16: lane l holds A[l&15][8*(l>>4)+e], B[8*(l>>4)+e][l&15], e = 0..7; same C/D map
// products are accumulated in k order with fmaf (the f32 form is documented as an exact fmaf chain).
typedef float hipemu_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 hipemu_bf16x8 __attribute__((ext_vector_type(8)));
static inline void hipemu_wave_barrier() { hipemu::wave_barrier(); }
static inline hipemu_f32x4 hipemu_mfma_f32_16x16x4f32(float a, float b, hipemu_f32x4 c, int, int, int) {
  hipemu::Wave& w = *hipemu::wave;
  const int l = hipemu::tid_flat & 63;
  memcpy(w.frag[l], &a, 4);
  memcpy(w.frag[l] + 16, &b, 4);
  hipemu::wave_barrier();
  const int col = l & 15;
  for (int r = 0; r < 4; ++r) {
    const int row = (l >> 4) * 4 + r;
    float acc = c[r];
    for (int k = 0; k < 4; ++k) {
      float av, bv;
      memcpy(&av, w.frag[k * 16 + row], 4);
      memcpy(&bv, w.frag[k * 16 + col] + 16, 4);
      acc = fmaf(av, bv, acc);
    }
    c[r] = acc;
  }
  hipemu::wave_barrier();
  return c;
}
static inline hipemu_f32x4 hipemu_mfma_f32_16x16x32_bf16(hipemu_bf16x8 a, hipemu_bf16x8 b, hipemu_f32x4 c, int, int, int) {
  hipemu::Wave& w = *hipemu::wave;
  const int l = hipemu::tid_flat & 63;
  memcpy(w.frag[l], &a, 16);
  memcpy(w.frag[l] + 16, &b, 16);
  hipemu::wave_barrier();
  const int col = l & 15;
  for (int r = 0; r < 4; ++r) {
    const int row = (l >> 4) * 4 + r;
    float acc = c[r];
    for (int g = 0; g < 4; ++g) {
      hipemu_bf16x8 av, bv;
      memcpy(&av, w.frag[g * 16 + row], 16);
      memcpy(&bv, w.frag[g * 16 + col] + 16, 16);
      for (int e = 0; e < 8; ++e) acc = fmaf((float)av[e], (float)bv[e], acc);
    }
    c[r] = acc;
  }
  hipemu::wave_barrier();
  return c;
}
#define __builtin_amdgcn_mfma_f32_16x16x4f32 hipemu_mfma_f32_16x16x4f32
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16 hipemu_mfma_f32_16x16x32_bf16
#define __builtin_amdgcn_wave_barrier hipemu_wave_barrier

// ---- raw buffer access (stride 0): every dword is range-checked against num_records, out-of-range loads give 0,
// out-of-range stores are dropped -- the property conv2d.hip's predicate-free tile loop relies on ----
struct hipemu_rsrc { char* base; unsigned num; };
typedef hipemu_rsrc __amdgpu_buffer_rsrc_t;
typedef unsigned hipemu_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned hipemu_u32x2 __attribute__((ext_vector_type(2)));
template <class P> static inline hipemu_rsrc hipemu_make_rsrc(P* p, short, unsigned num, unsigned) {
  return hipemu_rsrc{reinterpret_cast<char*>(const_cast<typename std::remove_const<P>::type*>(p)), num};
}
template <int N, class V> static inline V hipemu_buf_load(hipemu_rsrc r, unsigned voff, unsigned soff) {
  V v;
  for (int i = 0; i < N; ++i) {
    unsigned w = 0;
    const unsigned long long o = (unsigned long long)voff + soff + 4ull * i;
    if (o + 4 <= r.num) memcpy(&w, r.base + o, 4);
    v[i] = w;
  }
  return v;
}
static inline hipemu_u32x4 hipemu_buf_load_b128(hipemu_rsrc r, unsigned v, unsigned s, int) { return hipemu_buf_load<4, hipemu_u32x4>(r, v, s); }
static inline hipemu_u32x2 hipemu_buf_load_b64(hipemu_rsrc r, unsigned v, unsigned s, int) { return hipemu_buf_load<2, hipemu_u32x2>(r, v, s); }
// a buffer store occupies a slot of the lane's in-order vmcnt queue like any other vector-memory operation: kernels
// that leave stores in flight behind an LDS-DMA load (s_waitcnt vmcnt(n > 0), rnnt_joint_rows_kernel) count them
static inline void hipemu_vm_marker() {
  hipemu::Wave::Dma d;
  d.dst = nullptr;
  d.size = 0;
  hipemu::wave->dma[hipemu::tid_flat & 63].push_back(d);
}
static inline void hipemu_buf_store_b64(hipemu_u32x2 d, hipemu_rsrc r, unsigned voff, unsigned soff, int) {
  hipemu_vm_marker();
  for (int i = 0; i < 2; ++i) {
    const unsigned long long o = (unsigned long long)voff + soff + 4ull * i;
    unsigned w = d[i];
    if (o + 4 <= r.num) memcpy(r.base + o, &w, 4);
  }
}
static inline void hipemu_buf_store_b128(hipemu_u32x4 d, hipemu_rsrc r, unsigned voff, unsigned soff, int) {
  hipemu_vm_marker();
  for (int i = 0; i < 4; ++i) {
    const unsigned long long o = (unsigned long long)voff + soff + 4ull * i;
    unsigned w = d[i];
    if (o + 4 <= r.num) memcpy(r.base + o, &w, 4);
  }
}
#define __builtin_amdgcn_raw_buffer_store_b128 hipemu_buf_store_b128
#define __builtin_amdgcn_make_buffer_rsrc hipemu_make_rsrc
#define __builtin_amdgcn_raw_buffer_load_b128 hipemu_buf_load_b128
#define __builtin_amdgcn_raw_buffer_load_b64 hipemu_buf_load_b64
#define __builtin_amdgcn_raw_buffer_store_b64 hipemu_buf_store_b64
#define __builtin_amdgcn_s_barrier hipemu_raw_barrier
// ds_read_b64_tr_b16 as a wave collective (lane mapping as probed on gfx950, see the header of gemm_bf16.hip): within
// a 16-lane group, lane 4a+b supplies the address of row a, columns 4b..4b+3 of a 4x16 bf16 block; lane i receives
// column i (rows 0..3)
typedef __bf16 hipemu_bf16x4 __attribute__((ext_vector_type(4)));
template <class P> static inline hipemu_bf16x4 hipemu_ds_read_tr16(P p) {
  hipemu::Wave& w = *hipemu::wave;
  const int l = hipemu::tid_flat & 63;
  w.slot[l] = (uint64_t)reinterpret_cast<uintptr_t>(p);
  hipemu::wave_barrier();
  hipemu_bf16x4 r;
  const int g = l & ~15, i = l & 15;
  for (int a = 0; a < 4; ++a) {
    const __bf16* src = reinterpret_cast<const __bf16*>((uintptr_t)w.slot[g + 4 * a + (i >> 2)]);
    r[a] = src[i & 3];
  }
  hipemu::wave_barrier();
  return r;
}
// global_load_lds (16 B per lane): LDS destination = the wave-uniform base + 16 * lane.  The data is captured at issue
// (kernel inputs do not change during a launch) and written to LDS when a s_waitcnt retires it (hipemu_waitcnt_vm):
// a consumer that reads the tile without the wait (+ barrier) sees stale LDS, as it may on the hardware
// HIPEMU_DMA_EAGER=1: the opposite extreme -- the data lands AT ISSUE, the earliest moment the hardware allows: a kernel
// that re-arms an LDS region which another wave may still be reading (write-after-read) then reads the new tile early
static inline bool hipemu_dma_eager() {
  static const bool eager = getenv("HIPEMU_DMA_EAGER") && atoi(getenv("HIPEMU_DMA_EAGER")) != 0;
  return eager;
}
template <class G, class L> static inline void hipemu_global_load_lds(G* g, L* lds, int size, int, int) {
  const int lane = hipemu::tid_flat & 63;
  if (hipemu_dma_eager()) {
    memcpy(reinterpret_cast<char*>(lds) + (size_t)size * lane, reinterpret_cast<const char*>(g), (size_t)size);
    return;
  }
  hipemu::Wave::Dma d;
  d.dst = reinterpret_cast<char*>(lds) + (size_t)size * lane;
  d.size = size;
  memcpy(d.data, reinterpret_cast<const char*>(g), (size_t)size);
  hipemu::wave->dma[lane].push_back(d);
}
#define __builtin_amdgcn_global_load_lds hipemu_global_load_lds
// buffer_load_dwordx4 ... lds (raw buffer, 16 B per lane): every dword is range-checked, out-of-range dwords land as 0
template <class L> static inline void hipemu_raw_buffer_load_lds(hipemu_rsrc r, L* lds, int size, int voff, int soff, int off, int) {
  const int lane = hipemu::tid_flat & 63;
  hipemu::Wave::Dma d;
  d.dst = reinterpret_cast<char*>(lds) + (size_t)size * lane;
  d.size = size;
  for (int i = 0; i < size / 4; ++i) {
    unsigned w = 0;
    const unsigned long long o = (unsigned long long)(unsigned)voff + (unsigned)soff + (unsigned)off + 4ull * i;
    if (o + 4 <= r.num) memcpy(&w, r.base + o, 4);
    memcpy(d.data + 4 * i, &w, 4);
  }
  if (hipemu_dma_eager()) { memcpy(d.dst, d.data, (size_t)size); return; }
  hipemu::wave->dma[lane].push_back(d);
}
#define __builtin_amdgcn_raw_ptr_buffer_load_lds hipemu_raw_buffer_load_lds
#define __builtin_amdgcn_sched_barrier(m) ((void)0)
#define __builtin_amdgcn_readfirstlane(x) (x)              /* only ever applied to wave-uniform values */
#define __builtin_amdgcn_s_setprio(p) ((void)0)        /* scheduling hint only */
/* s_waitcnt as a builtin (gfx9 encoding: vmcnt = simm16[3:0] | simm16[15:14] << 4) */
#define __builtin_amdgcn_s_waitcnt(imm) hipemu_waitcnt_vm(((imm) & 15) | ((((imm) >> 14) & 3) << 4))
#define __builtin_amdgcn_sched_group_barrier(m, n, id) ((void)0)
#define __builtin_amdgcn_exp2f(x) exp2f(x)
#define __builtin_amdgcn_ds_read_tr16_b64_v4bf16 hipemu_ds_read_tr16
static inline float __uint_as_float(unsigned v) { float f; memcpy(&f, &v, 4); return f; }
static inline unsigned __float_as_uint(float f) { unsigned v; memcpy(&v, &f, 4); return v; }
struct uint2 { unsigned x, y; };
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }

static inline float atomicAdd(float* p, float v);
static inline float unsafeAtomicAdd(float* p, float v) { return atomicAdd(p, v); }
static inline float atomicAdd(float* p, float v) {
  uint32_t* ip = reinterpret_cast<uint32_t*>(p);
  uint32_t old = __atomic_load_n(ip, __ATOMIC_RELAXED), want;
  float f;
  do {
    memcpy(&f, &old, 4);
    f += v;
    memcpy(&want, &f, 4);
  } while (!__atomic_compare_exchange_n(ip, &old, want, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
  memcpy(&f, &old, 4);
  return f;
}

namespace hipemu {
static inline void fiber_entry() {
  Wave* w = wave;
  w->invoke(w->closure);
  w->dma[w->cur].clear();   // LDS dies with the block
  w->done[w->cur] = true;   // returns to uc_link = the wave's scheduler
#ifndef HIPEMU_UCONTEXT
  hipemu_switch(&w->ctx[w->cur], w->sched);   // never resumed
  abort();
#endif
}
template <class F>
static inline void launch(dim3 grid, dim3 block, size_t shmem, F body) {
  Launch L;
  L.grid = grid;
  L.block = block;
  L.dyn.assign(shmem / 8 + 4, 0.0);
  const int nthr = (int)(block.x * block.y * block.z);
  const int nw = (nthr + 63) / 64;
  pthread_barrier_init(&L.bar, nullptr, nw);
  std::vector<std::thread> th;
  th.reserve(nw);
  for (int wi = 0; wi < nw; ++wi) {
    th.emplace_back([&L, wi, nw, nthr, grid, block, &body]() {
      Wave* w = new Wave();
      w->L = &L;
      w->index = wi;
      w->n = (wi == nw - 1) ? nthr - 64 * wi : 64;
      w->stacks = (char*)mmap(nullptr, 64 * kFiberStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
      w->invoke = [](void* c) { (*static_cast<F*>(c))(); };
      w->closure = (void*)&body;
      cur = &L;
      wave = w;
      blockDim = block;
      gridDim = grid;
      for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
          for (unsigned bx = 0; bx < grid.x; ++bx) {
            blockIdx = dim3(bx, by, bz);
            w->alive = w->n;
            w->arrived = 0;
            for (int l = 0; l < w->n; ++l) {
              w->done[l] = false;
#ifdef HIPEMU_UCONTEXT
              getcontext(&w->ctx[l]);
              w->ctx[l].uc_stack.ss_sp = w->stacks + (size_t)l * kFiberStack;
              w->ctx[l].uc_stack.ss_size = kFiberStack;
              w->ctx[l].uc_link = &w->sched;
              makecontext(&w->ctx[l], (void (*)())fiber_entry, 0);
#else
              {   // initial frame: six callee-saved registers, the entry point as return address, a null caller
                void** top = reinterpret_cast<void**>(w->stacks + (size_t)(l + 1) * kFiberStack);   // 16-byte aligned
                top[-1] = nullptr;
                top[-2] = reinterpret_cast<void*>(&fiber_entry);
                for (int r = 3; r <= 8; ++r) top[-r] = nullptr;
                w->ctx[l] = top - 8;
              }
#endif
            }
            while (w->alive > 0) {
              for (int l = 0; l < w->n; ++l) {
                if (w->done[l]) continue;
                w->cur = l;
                const int t = wi * 64 + l;
                tid_flat = t;
                threadIdx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
#ifdef HIPEMU_UCONTEXT
                swapcontext(&w->sched, &w->ctx[l]);
#else
                hipemu_switch(&w->sched, w->ctx[l]);
#endif
                if (w->done[l]) {
                  // a lane that returned no longer takes part in collectives: release the others if they only
                  // waited for it (wave-level collectives only; see the header comment for __syncthreads)
                  if (--w->alive > 0 && w->arrived >= w->alive) {
                    w->arrived = 0;
                    ++w->gen;
                  }
                }
              }
            }
            pthread_barrier_wait(&L.bar);  // a block ends before the next one reuses its statics
          }
      munmap(w->stacks, 64 * kFiberStack);
      delete w;
    });
  }
  for (auto& t : th) t.join();
  pthread_barrier_destroy(&L.bar);
}
}  // namespace hipemu

#define HIPEMU_KERNEL_NAME(...) __VA_ARGS__
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...)                     \
  do {                                                                                  \
    (void)(stream);                                                                     \
    hipemu::launch((grid), (block), (size_t)(shmem), [&]() { HIPEMU_KERNEL_NAME(kernel)(__VA_ARGS__); }); \
  } while (0)
