// gemm_bf16.hip -- the throughput GEMM: bf16 operands in HBM, fp32 accumulate on
// v_mfma_f32_16x16x32_bf16, fused epilogue, fp32 or bf16 output.
//
// Why a second kernel: the fp32-operand kernel (gemm.hip) moves 4 B per operand
// element HBM/L2 -> LDS and converts on the fly; at a 128x128 tile that is
// 32 B per KFLOP, i.e. more L2 bandwidth than the chip has at MFMA rates (it
// measures ~120-250 TFLOP/s).  With bf16 shadow copies of weights / activations
// the same tile needs 16 B/KFLOP and no VALU conversion.
//
// Tile 128x128x64, 4 waves (2x2), wave tile 64x64 = 4x4 MFMA fragments, 32 MFMAs
// per wave per k-tile.  Global loads are 16 B/lane straight into registers for
// tile t+1 while tile t is multiplied (register software pipeline), then 16-B
// LDS stores.  Two LDS images:
//   KC operand (reduction index contiguous in memory): [128 rows][64 k], pitch
//      144 B (128 + 16 pad => 16 rows x ds_read_b128 hit 16 distinct 4-bank slots).
//   RC operand (row index contiguous, e.g. both operands of a weight gradient):
//      kept k-major as in memory, [64 k][128 rows], pitch 288 B; the MFMA fragment
//      (row r, 8 consecutive k) is assembled by two ds_read_b64_tr_b16 transpose
//      reads -- no register transposes, no strided LDS traffic.
//      Lane mapping of ds_read_b64_tr_b16 (probed on gfx950, tools/probe): within a
//      16-lane group, lane 4a+b supplies the address of matrix row a, columns
//      4b..4b+3; lane i receives column i of the 4x16 block.
#include "common.h"
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <type_traits>
#include <vector>
#ifndef NSP_GEMM_8P_AB
#define NSP_GEMM_8P_AB 0   // development (-DNSP_GEMM_8P_AB=1): also compile the direct-epilogue twins (NSP_GEMM_8P_VAR=0 / 8) and the main-loop ablations (20 / 36)
#endif

namespace {

constexpr int BM = 128, BN = 128, BK = 64, NTHREADS = 256;
// LDS images of the register-staged kernels.  KC tiles ([128 rows][64 k]): 128-B rows + 32 B of padding --
// with 16 B (round 1) a ds_read_b128 fragment took 8 LDS cycles instead of 4.  RC tiles ([64 k][128 cols]):
// unpadded 256-B rows with the 16-B chunk index XOR-swizzled by rr_swz(k) exactly like the LDS-DMA
// weight-gradient kernel (padding cannot fix the transposed reads: 4 cycles instead of 2 at any pitch;
// measured 33 % bank-conflict cycles on gemm_bf16_kernel<false,false>).
constexpr int PITCH_KC = 160;
constexpr int PITCH_RC = 256;
constexpr int TILE_BYTES = 128 * PITCH_KC;  // 20480 >= 64 * PITCH_RC
__device__ __forceinline__ int rr_swz(int krow) { return ((krow & 3) | (((krow >> 3) & 1) << 2)) << 1; }

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <bool KC>
struct Bf16TileLoader {
  u32x4 r[4];
  // element (row, k): KC: base[row*ld + k] ; RC: base[k*ld + row]
  __device__ __forceinline__ void load(const __bf16* __restrict__ base, long long ld, int row0, int R,
                                       int k0, int Kend) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + i * NTHREADS;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (KC) {
        const int row = idx >> 3, c = idx & 7;
        const int gr = row0 + row, gk = k0 + c * 8;
        if (gr < R && gk < Kend) v = *reinterpret_cast<const u32x4*>(base + (long long)gr * ld + gk);
      } else {
        const int kk = idx >> 4, cb = idx & 15;
        const int gk = k0 + kk, gr = row0 + cb * 8;
        if (gk < Kend && gr < R) v = *reinterpret_cast<const u32x4*>(base + (long long)gk * ld + gr);
      }
      r[i] = v;
    }
  }
  __device__ __forceinline__ void store(unsigned char* lds) const {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + i * NTHREADS;
      if (KC) {
        const int row = idx >> 3, c = idx & 7;
        *reinterpret_cast<u32x4*>(lds + row * PITCH_KC + c * 16) = r[i];
      } else {
        const int kk = idx >> 4, cb = idx & 15;
        *reinterpret_cast<u32x4*>(lds + kk * PITCH_RC + ((cb ^ rr_swz(kk)) << 4)) = r[i];
      }
    }
  }
};

// MFMA operand fragment for 16 rows starting at `rbase`, k-step s (32 k), lane (r, g)
template <bool KC>
__device__ __forceinline__ bf16x8 read_frag(const unsigned char* lds, int rbase, int s, int r, int g) {
  if (KC) {
    return *reinterpret_cast<const bf16x8*>(lds + (rbase + r) * PITCH_KC + (s * 4 + g) * 16);
  } else {
    const int a = r >> 2, b = r & 3;
    const int k0 = s * 32 + g * 8 + a, col = rbase + b * 4;      // rows k0 and k0 + 4 share their swizzle
    const unsigned char* p = lds + k0 * PITCH_RC + (((col >> 3) ^ rr_swz(k0)) << 4) + ((col & 7) << 1);
    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p + 4 * PITCH_RC));
    bf16x8 o;
    o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3];
    o[4] = hi[0]; o[5] = hi[1]; o[6] = hi[2]; o[7] = hi[3];
    return o;
  }
}

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int NX = 8;
  if (nwg < NX * 2) return bid;
  int q = nwg / NX, rem = nwg % NX;
  int xcd = bid % NX, slot = bid / NX;
  int base = xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q;
  return base + slot;
}

// (tile, split, batch) of a workgroup.  Split-K launches of ONE problem use a flat grid (gridDim.z == 1):
// workgroup -> v = xcd_remap(id) -> split = v / ntiles, tile = v % ntiles, so that an XCD (id % 8) works
// on whole reduction splits: all tiles of a split read the same k-rows of both operands, and with the
// tiles of one split spread over eight private L2s (the old z-major grid) every operand panel was
// fetched from HBM / Infinity Cache once per XCD -- measured in round 2: the weight-gradient kernels
// read 1.76x (2048 x 512 outputs) to 5x (512 x 512 outputs) their operand bytes.
#ifndef NSP_GEMM_TRACE
#define NSP_GEMM_TRACE 0   // development (tools/gemm_wg_trace.py): per-workgroup timestamps of gemm_bf16_kk_glds_kernel<0>
#endif
#if NSP_GEMM_TRACE
__device__ unsigned long long nsp_gemm_trace_buf[4 * 16384];
#define NSP_TRACE_MARK(i) do { if (threadIdx.x == 0 && blockIdx.x < 16384) nsp_gemm_trace_buf[4 * blockIdx.x + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define NSP_TRACE_MARK(i) do { } while (0)
#endif
struct TileCoord { int tile, split, z1, z2; };
__device__ __forceinline__ TileCoord tile_coord(const nsp_gemm_params& p, int ntiles) {
  TileCoord c;
  if (gridDim.z == 1 && p.splitk > 1) {
    const int v = xcd_remap(blockIdx.x, ntiles * p.splitk);
    c.split = v / ntiles;
    c.tile = v - c.split * ntiles;
    c.z1 = c.z2 = 0;
    return c;
  }
  c.tile = xcd_remap(blockIdx.x, ntiles);
  int z = blockIdx.z;
  c.split = z % p.splitk;
  z /= p.splitk;
  c.z2 = z % p.batch2;
  c.z1 = z / p.batch2;
  return c;
}

// Epilogue cache policy.  The outputs of the step's GEMMs are 100-800 MB images that nobody re-reads before they have
// left the 4-MB L2 of their XCD anyway; written with plain stores they push the weight / activation panels that the
// OTHER tiles of the launch are about to re-read out of it (PMC, round 3: gemm_bf16_kk_glds_kernel<0> read 1.5x its
// algorithmic bytes).  NSP_EPI_STORE: 0 = plain stores, 1 = non-temporal (default), 2 = fp32 images with sc1 (write-
// through) and bf16 images non-temporal; NSP_EPI_SIDE_NT: the residual / act' side operand (read exactly once) with
// non-temporal loads.  Measured interleaved at M = 102400 (profiles/r03q*_store_policy_ab.log): non-temporal stores
// x1.15 on the FFN first linear and the stacked QKV projection (bf16 images), x1.48 on the fp32-output GEMMs without a
// side operand (pointwise conv 1, d x d data gradients), neutral on the K >= 1536 shapes; sc1 buys nothing over plain;
// non-temporal side loads +4-5 % on the residual epilogues.  Full step: 130.2 -> 128.6 ms.
#ifndef NSP_EPI_STORE
#define NSP_EPI_STORE 1
#endif
#ifndef NSP_EPI_ABLATE
#define NSP_EPI_ABLATE 0    // development: bit 0 no stores, bit 1 no epilogue arithmetic, bit 2 no LDS staging (timing only)
#endif
#ifndef NSP_EPI_SIDE_NT
#define NSP_EPI_SIDE_NT 1
#endif
__device__ __forceinline__ void store4(void* base, int dtype, long long off, const float* v, int nv,
                                       bool vec) {
#if NSP_EPI_ABLATE & 1      // (ablation: values computed, nothing stored)
  asm volatile("" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
  return;
#endif
  if (dtype == NSP_DT_BF16) {
    __bf16* o = reinterpret_cast<__bf16*>(base) + off;
    if (vec) {
      bf16x4 h;
      h[0] = (__bf16)v[0]; h[1] = (__bf16)v[1]; h[2] = (__bf16)v[2]; h[3] = (__bf16)v[3];
#if NSP_EPI_STORE >= 1
      __builtin_nontemporal_store(h, reinterpret_cast<bf16x4*>(o));
#else
      *reinterpret_cast<bf16x4*>(o) = h;
#endif
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) if (e < nv) o[e] = (__bf16)v[e];
    }
  } else {
    float* o = reinterpret_cast<float*>(base) + off;
    if (vec) {
#if NSP_EPI_STORE == 1
      typedef __attribute__((ext_vector_type(4))) float f32x4_;
      f32x4_ q = {v[0], v[1], v[2], v[3]};
      __builtin_nontemporal_store(q, reinterpret_cast<f32x4_*>(o));
#elif NSP_EPI_STORE == 2
      typedef __attribute__((ext_vector_type(4))) float f32x4_;
      f32x4_ q = {v[0], v[1], v[2], v[3]};
      asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(o), "v"(q) : "memory");
#else
      *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
#endif
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) if (e < nv) o[e] = v[e];
    }
  }
}

__device__ __forceinline__ void load4(const void* base, int dtype, long long off, float* v, int nv,
                                      bool vec) {
  if (dtype == NSP_DT_BF16) {
    const __bf16* o = reinterpret_cast<const __bf16*>(base) + off;
    if (vec) {
      bf16x4 h = *reinterpret_cast<const bf16x4*>(o);
      v[0] = (float)h[0]; v[1] = (float)h[1]; v[2] = (float)h[2]; v[3] = (float)h[3];
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) if (e < nv) v[e] = (float)o[e];
    }
  } else {
    const float* o = reinterpret_cast<const float*>(base) + off;
    if (vec) {
      float4 f = *reinterpret_cast<const float4*>(o);
      v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) if (e < nv) v[e] = o[e];
    }
  }
}

// SWZ: the per-wave staging slab is 16 rows x 64 floats UNPADDED (4 KB) with the 16-B chunk index XOR-ed with
// (row & 7) instead of 68-float rows (the 8-wave kernel: 8 x 4 KB next to a 128-KB ring = the whole 160-KB LDS);
// conflict-free for the ds_write_b128 of a fragment (8 consecutive rows, one chunk) and for the row-major read-back.
template <bool SWZ>
__device__ __forceinline__ int stage_idx(int row, int chunk) {
  return SWZ ? row * 64 + ((chunk ^ (row & 7)) << 2) : row * 68 + (chunk << 2);
}
// ---- RNN-T joint epilogues (nsp_gemm_params::epi_mode, nsp_rnnt_joint_gemm): the tile holds logits
// of 16*MI x 64 (rows = compacted lattice nodes, cols = vocabulary) per wave.  Same LDS staging as
// the standard epilogue: after the read-back the 16 lanes (lane & 15) of one row group hold the 64
// columns of one row, so row statistics are 4 xor-shuffles.
//   LSE     : per row (max, sum exp) of this 64-column block + the raw logits at the blank / label
//             columns.  Nothing of the [M, V] logit matrix reaches HBM (128 B of partials per row
//             instead of 4 KB of fp32 logits).
//   DLOGITS : d loss / d logits from the recomputed tile, written as the bf16 operand image of the
//             two gradient GEMMs; column sums (output-bias gradient) per 64-row block, no atomics.
// Both are straight-line per row group (mode is a template parameter) and keep the per-row scalars
// (label; log-sum-exp and the two lattice gradients) of ONE row block in registers, refilled for the next
// block right after use with clamped, unconditional loads -- see gemm_epilogue_fast for why (in-order
// vmcnt: a load issued behind a store cannot be waited for without waiting for the store).
template <int MI, bool LSE, bool SWZ = false>
__device__ __forceinline__ void rnnt_epilogue_core(const nsp_gemm_params& p, f32x4 (&acc)[MI][4], float* stage,
                                                   int mrow0, int nbase, int lane) {
  // stage: this wave's private slab; mrow0 / nbase: first row / column of the wave's 16 MI x 64 logit tile
  const int fr = lane & 15, fg = lane >> 4;
  const int er = lane >> 4, ec = (lane & 15) * 4;
  const int n = nbase + ec;                          // first of this lane's 4 columns (n + 3 < N: N % 64 == 0)
  const int npart = p.N >> 6, pidx = nbase >> 6;
  const bool colok = n < p.N;
  float b4[4] = {0.f, 0.f, 0.f, 0.f};
  if (p.bias && colok) {
    const float4 b = *reinterpret_cast<const float4*>(p.bias + n);
    b4[0] = b.x; b4[1] = b.y; b4[2] = b.z; b4[3] = b.w;
  }
  __bf16* cbase = reinterpret_cast<__bf16*>(p.C) + (long long)(mrow0 + er) * p.ldc + n;   // DLOGITS output, row er of block 0
  const long long ldc4 = 4ll * p.ldc;
  // per-row scalars of the current row block: LSE needs the label; DLOGITS one 16-B record per row
  // {lse, g_blank * scale, g_label * scale, bits(label)} (epi_f0 as float4 [M], packed by nsp_rnnt_joint_gemm)
  float4 rec[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int mcl = min(mrow0 + er + 4 * j, p.M - 1);
    if (LSE) rec[j] = make_float4(0.f, 0.f, 0.f, __int_as_float(p.epi_lab[mcl]));
    else rec[j] = reinterpret_cast<const float4*>(p.epi_f0)[mcl];
  }
  float csum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
#if !(NSP_EPI_ABLATE & 4)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
      *reinterpret_cast<float4*>(stage + stage_idx<SWZ>(fr, ni * 4 + fg)) =
          make_float4(acc[mi][ni][0], acc[mi][ni][1], acc[mi][ni][2], acc[mi][ni][3]);
#endif
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      __builtin_amdgcn_sched_barrier(0);
      const int row = er + 4 * j;
      const float4 a4 = *reinterpret_cast<const float4*>(stage + stage_idx<SWZ>(row, lane & 15));
      const int m = mrow0 + mi * 16 + row;
      const bool rowok = m < p.M && colok;
      const int lab = __float_as_int(rec[j].w);
      const float ls = rec[j].x, gb = rec[j].y, gl = rec[j].z;
#ifndef NSP_HOST_EMULATION
      // (persistent caller: the per-row record is consumed on every path, see gemm_epilogue_fast)
      if (SWZ) asm volatile("" :: "v"(lab), "v"(ls), "v"(gb), "v"(gl));
#endif
      if (mi + 1 < MI) {
        const int mcl = min(m + 16, p.M - 1);
        if (LSE) rec[j].w = __int_as_float(p.epi_lab[mcl]);
        else rec[j] = reinterpret_cast<const float4*>(p.epi_f0)[mcl];
      }
      float v[4] = {a4.x + b4[0], a4.y + b4[1], a4.z + b4[2], a4.w + b4[3]};
      if (LSE) {
        float mx = -FLT_MAX;
#pragma unroll
        for (int e = 0; e < 4; ++e) if (n + e < p.epi_ncols) mx = fmaxf(mx, v[e]);
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        float sm = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) if (n + e < p.epi_ncols) sm += __expf(v[e] - mx);
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) sm += __shfl_xor(sm, o, 64);
        if (rowok) {
          if ((lane & 15) == 0)
            *reinterpret_cast<float2*>(p.epi_f0 + ((long long)m * npart + pidx) * 2) = make_float2(mx, sm);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (n + e == p.epi_blank) p.epi_f1[m] = v[e];
            if (n + e == lab) p.epi_f2[m] = v[e];
          }
        }
      } else {
        const float gs = gb + gl;
        float g[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = -gs * __expf(v[e] - ls);
          if (n + e == p.epi_blank) t += gb;
          if (n + e == lab) t += gl;
          g[e] = (rowok && n + e < p.epi_ncols) ? t : 0.f;
        }
        bf16x4 o;
        o[0] = (__bf16)g[0]; o[1] = (__bf16)g[1]; o[2] = (__bf16)g[2]; o[3] = (__bf16)g[3];
#pragma unroll
        for (int e = 0; e < 4; ++e) csum[e] += g[e];
        if (rowok) __builtin_nontemporal_store(o, reinterpret_cast<bf16x4*>(cbase + (long long)(mi * 4 + j) * ldc4));   // 7.6 GB image: streams past L2 (see store4)
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (!LSE && p.epi_f3) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      csum[e] += __shfl_xor(csum[e], 16, 64);
      csum[e] += __shfl_xor(csum[e], 32, 64);
    }
    if (lane < 16 && colok) {
      const long long slab = (long long)mrow0 / (16 * MI);   // one slab per 16*MI-row block
      *reinterpret_cast<float4*>(p.epi_f3 + slab * p.N + n) = make_float4(csum[0], csum[1], csum[2], csum[3]);
    }
  }
}

template <int MI, bool LSE>
__device__ __forceinline__ void rnnt_epilogue_mode(const nsp_gemm_params& p, f32x4 (&acc)[MI][4], unsigned char* smem,
                                                   int m0, int n0, int wm, int wn, int lane, int wave) {
  rnnt_epilogue_core<MI, LSE, false>(p, acc, reinterpret_cast<float*>(smem) + wave * (16 * 68), m0 + wm * (16 * MI),
                                     n0 + wn * 64, lane);
}

template <int MI>
__device__ __forceinline__ void rnnt_epilogue(const nsp_gemm_params& p, f32x4 (&acc)[MI][4], unsigned char* smem,
                                              int m0, int n0, int wm, int wn, int lane, int wave) {
  if (p.epi_mode == NSP_EPI_RNNT_LSE) rnnt_epilogue_mode<MI, true>(p, acc, smem, m0, n0, wm, wn, lane, wave);
  else rnnt_epilogue_mode<MI, false>(p, acc, smem, m0, n0, wm, wn, lane, wave);
}

// ---- fast standard epilogue: same staging and arithmetic as gemm_epilogue below, for the common case
// (vector-aligned output, N % 4 == 0, no atomics, at most ONE side operand: residual or act' source).
// gfx9 retires vector-memory operations in issue order, so a load that is issued after a store cannot
// be waited for without also waiting for that store to reach L2 -- with a bias / residual / act' load
// inside every row group the generic loop serialised 16 store round trips per tile.  Here the bias is
// loaded once, and the side operand of row block mi+1 is requested BEFORE the stores of block mi are
// issued, so a wait only ever covers stores that are a whole block old.
// That only works when the row loop is straight-line code: with the configuration tested at run time
// (activation switch, dtype branches) hipcc's wait insertion falls back to vmcnt(0) in front of the
// branchy region, i.e. it waits for the prefetch it has just issued AND for every older store
// (measured, 51200 x 2048 x 512 back to back: bf16 out 158 us; + act' source 300 us whether the source
// is bf16 or fp32 and whether act' is relu or swish; + residual 227 us).  The configurations the
// training step uses are therefore compiled as specialisations (EpiSpec); everything else takes the
// run-time version (EpiRuntime), which is correct but serialises as described.
struct EpiRuntime { static constexpr bool kStatic = false; };
template <int ACT_, int DACT_, bool C16_, bool PRE16_, bool RES_, bool DROP_>
struct EpiSpec {
  static constexpr bool kStatic = true;
  static constexpr int ACT = ACT_, DACT = DACT_;   // DACT > 0: bf16 act' source
  static constexpr bool C16 = C16_, PRE16 = PRE16_, RES = RES_, DROP = DROP_;
};

typedef __attribute__((ext_vector_type(4))) unsigned int cu32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int cu32x2_t;
#ifndef NSP_EPI_BUF
#define NSP_EPI_BUF 1         // 0: the static specialisations with predicated global stores / clamped global side loads again
#endif
#ifndef NSP_EPI_RG_GROUP_8P
#define NSP_EPI_RG_GROUP_8P 1   // row groups the scheduler may interleave (8-phase kernel / 128 x 128 kernels)
#endif
#ifndef NSP_EPI_RG_GROUP_128
#define NSP_EPI_RG_GROUP_128 1
#endif
// the specialisations that have a stream-K twin of the 8-phase kernel: the fp32-output epilogues of the step's N = 512
// products (plain data gradients; bias / dropout / residual of the FFN second linear and the attention output)
template <class S>
__host__ __device__ constexpr bool epi_spec_streamk_twin() {
  if constexpr (S::kStatic) return !S::C16 && !S::PRE16 && S::ACT == 0 && S::DACT == 0;
  else return false;
}
template <class S>
__host__ __device__ constexpr bool epi_spec_has_side() {
  if constexpr (S::kStatic) return S::DACT != NSP_ACT_NONE || S::RES;
  else return true;
}
template <class S>
__host__ __device__ constexpr bool epi_spec_slabs() {
  if constexpr (S::kStatic) return S::DACT != NSP_ACT_NONE;
  else return true;
}
template <int MI, class S, bool SWZ = false>
__device__ __forceinline__ void gemm_epilogue_fast(const nsp_gemm_params& p, f32x4 (&acc)[MI][4], float* stage,
                                                   int mrow0, int n, int lane, long long coff,
                                                   const float4* bias_pre = nullptr) {
  const int fr = lane & 15, fg = lane >> 4;
  const int er = lane >> 4, ec = (lane & 15) * 4;
  const bool colok = n < p.N;                       // N % 4 == 0: all four columns or none
  bool has_res, has_dact, side16, has_pre, drop;
  int act, dact, c_dt, pre_dt;
  if constexpr (S::kStatic) {
    has_res = S::RES; has_dact = S::DACT != NSP_ACT_NONE; side16 = has_dact; has_pre = S::PRE16; drop = S::DROP;
    act = S::ACT; dact = S::DACT; c_dt = S::C16 ? NSP_DT_BF16 : NSP_DT_F32; pre_dt = NSP_DT_BF16;
  } else {
    has_res = p.res != nullptr; has_dact = p.dact_src != nullptr; side16 = !has_res && p.dact_dtype == NSP_DT_BF16;
    has_pre = p.pre_out != nullptr; drop = p.dropout_p > 0.f;
    act = p.act; dact = p.dact; c_dt = p.c_dtype; pre_dt = p.pre_dtype;
  }
  const bool has_side = has_res || has_dact;
  const char* side = has_res ? reinterpret_cast<const char*>(p.res) : reinterpret_cast<const char*>(p.dact_src);
  const long long off0 = coff + (long long)(mrow0 + er) * p.ldc + n;
  const long long ldc4 = 4ll * p.ldc;
  // W32 (round 6; the static specialisations, ldc < 2^22 -- epi_spec_visit): every address of the wave's tile is a
  // SCALAR origin (SGPR pair, SALU arithmetic per row group) + a 32-bit lane offset that is computed once.  The 64-bit
  // per-lane forms below cost ~12 VALU per row group (v_mad_u64_u32 + two v_mul_lo_u32 for row * ldc alone) and the
  // dropout counter another ~16 on top of its two hashes; these epilogues are VALU-bound.
  constexpr bool W32 = S::kStatic;
  const int mrow_s = W32 ? __builtin_amdgcn_readfirstlane(mrow0) : mrow0;
  const int nbase_s = W32 ? __builtin_amdgcn_readfirstlane(n - (lane & 15) * 4) : n;
  const long long sorg = coff + (long long)mrow_s * p.ldc + nbase_s;          // element offset of the tile's origin (uniform)
  const unsigned loff = (unsigned)__umul24(er, (int)p.ldc) + (unsigned)((lane & 15) * 4);   // this lane inside row group 0
  // side operand (clamped, never predicated -- see `request`): its own origin, clamped into the matrix, so that the
  // lane offsets of a wave whose rows / columns lie beyond the edge stay non-negative
  const int morg = min(mrow_s, p.M - 1), norg = min(nbase_s, p.N - 4);
  const long long sorg_side = coff + (long long)morg * p.ldc + norg;
  const int rows_left = p.M - 1 - morg;                                       // last valid row relative to that origin
  // BUF (static specialisations): stores and side loads are BUFFER instructions whose descriptor starts at the tile's
  // origin and ends with the matrix -- rows >= M are out of range by construction, lanes with columns >= N carry an
  // out-of-range offset; the row group's offset rides in the scalar offset field.  No predicated blocks (each was a
  // basic block of its own: s_and_saveexec + branch, and nothing could be scheduled across it), no address VALU at all.
  // (the 128 x 128 kernel compiles thirteen specialisations into one kernel at 128 VGPRs: with three descriptors live in
  // the side-operand variants it spilled 178 registers and the FFN data gradient went from 352 to 488 us -- there the
  // side-operand variants keep global loads / stores, profiles/r06_gemm_epilogue.log)
  // STORES: the row group's offset is ADDED TO THE LANE OFFSET, the scalar-offset field stays 0.  With the offset in an SGPR
  // (first version of this path) some builds of the 8-phase kernel stored a wrong first dword in the last four of every 16
  // lanes of one row group -- the integer that the NEXT instruction, a VALU write of the store's first data register (the
  // next row group's LDS address), had just produced: on gfx950 a 128-bit buffer store still reads its data registers in
  // the cycle after issue, and hipcc's hazard recogniser leaves out the wait state when soffset is a register (its rule:
  // the hazard exists "only if the instruction is not using a register in the soffset field").  Whether a build showed
  // it depended on the register allocation, i.e. on unrelated code (found while a stream-K schedule was being added to the
  // kernel: profiles/r06_gemm_epilogue.log).  An out-of-range lane (OOB = 2^31) stays out of range: the sum is below
  // 2^32 and the descriptors are at most 2^30 bytes long.  (The side-operand LOADS keep the scalar offset: no such hazard.)
  constexpr bool BUF = W32 && NSP_EPI_BUF && (SWZ || !(S::kStatic && epi_spec_has_side<S>()));
  constexpr unsigned OOB = 0x80000000u;
  constexpr int AUXS = NSP_EPI_STORE >= 1 ? 2 : 0, AUXL = NSP_EPI_SIDE_NT ? 2 : 0;
  const long long rem_el = (long long)p.M * p.ldc - ((long long)mrow_s * p.ldc + nbase_s);   // elements from the origin to the end
  auto mkrsrc = [&](const void* ptr, int esz) {
    const long long b = rem_el * esz;
    const unsigned nrec = b <= 0 ? 0u : (b > 0x40000000ll ? 0x40000000u : (unsigned)b);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(ptr)) + sorg * esz, 0, nrec, 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t rc = mkrsrc(p.C, c_dt == NSP_DT_BF16 ? 2 : 4);
  const __amdgpu_buffer_rsrc_t rpre = mkrsrc(has_pre ? p.pre_out : p.C, 2);
  const __amdgpu_buffer_rsrc_t rside = mkrsrc(has_side ? reinterpret_cast<const void*>(side) : reinterpret_cast<const void*>(p.C), side16 ? 2 : 4);
  const unsigned lo2 = colok ? loff * 2u : OOB, lo4 = colok ? loff * 4u : OOB;
  float b4[4] = {0.f, 0.f, 0.f, 0.f};
  const bool bias_here = bias_pre == nullptr && p.bias && colok;
  float4 bld = make_float4(0.f, 0.f, 0.f, 0.f);
  if (bias_pre) bld = *bias_pre;                   // (requested by the kernel in front of its main loop)
  else if (bias_here) bld = *reinterpret_cast<const float4*>(p.bias + n);
  // The bias must have LANDED before the row loop, and the compiler must know it.  The row groups below are
  // predicated blocks (only rows < M store); hipcc's wait insertion does not carry "this load was waited
  // for" out of a conditional block, so it re-waited for the bias in EVERY row group -- as s_waitcnt
  // vmcnt(0), which on gfx9 (one in-order counter for loads and stores) also waits for every store of the
  // previous row groups: 16 store round trips per tile in the epilogues that have a bias but no side
  // operand (FFN first linear, pointwise conv 1, every plain Linear).  The builtin wait below is a wait the
  // compiler's scoreboard sees: afterwards only lgkmcnt waits remain inside the loop (round 3, .s audit).
  const uint32_t keep_thr = (uint32_t)(p.dropout_p * 65536.f);
  const float keep_inv = nsp_rcp(1.f - p.dropout_p);
  // side operand of the CURRENT row block, one 16-B (fp32) / 8-B (bf16) chunk per row group; the chunk of
  // row group j of block mi+1 is requested into the same registers right after block mi consumed its copy
  // and BEFORE block mi stores row group j: a wait for it covers only stores that are a block old
  uint4 raw[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) raw[j] = make_uint4(0u, 0u, 0u, 0u);
  // (every lane requests: rows / columns beyond the edge are clamped to the last valid chunk instead of
  // being predicated off -- a conditional load makes the buffer a phi of old and new value, and the copy
  // that resolves it waits for the load in the row group that issued it)
  const int ncl = min(n, p.N - 4);
  auto request = [&](int mi, int j, uint4& buf) {
    if constexpr (BUF) {
      const unsigned srq = (unsigned)(mi * 16 + 4 * j) * (unsigned)p.ldc;
      if (side16) {
        const cu32x2_t h = __builtin_amdgcn_raw_buffer_load_b64(rside, lo2, srq * 2u, AUXL);
        buf.x = h[0]; buf.y = h[1];
      } else {
        const cu32x4_t q = __builtin_amdgcn_raw_buffer_load_b128(rside, lo4, srq * 4u, AUXL);
        buf = make_uint4(q[0], q[1], q[2], q[3]);
      }
      return;
    }
    const char* src2;         // the chunk's address for a 2-byte / 4-byte side operand
    const char* src4;
    if constexpr (W32) {
      const unsigned rrel = (unsigned)min(mi * 16 + 4 * j + er, rows_left);
      const unsigned el = (unsigned)__umul24((int)rrel, (int)p.ldc) + (unsigned)(ncl - norg);
      src2 = side + sorg_side * 2 + (size_t)(el * 2u);
      src4 = side + sorg_side * 4 + (size_t)(el * 4u);
    } else {
      const int mcl = min(mrow0 + mi * 16 + er + 4 * j, p.M - 1);
      const long long off = coff + (long long)mcl * p.ldc + ncl;
      src2 = side + off * 2;
      src4 = side + off * 4;
    }
    if (side16) {
#if NSP_EPI_SIDE_NT
      typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_;
      const u32x2_ h = __builtin_nontemporal_load(reinterpret_cast<const u32x2_*>(src2));
      buf.x = h[0]; buf.y = h[1];
#else
      const uint2 h = *reinterpret_cast<const uint2*>(src2);
      buf.x = h.x; buf.y = h.y;
#endif
    } else {
#if NSP_EPI_SIDE_NT
      const u32x4 q = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src4));
      buf = make_uint4(q[0], q[1], q[2], q[3]);
#else
      buf = *reinterpret_cast<const uint4*>(src4);
#endif
    }
  };
  if (has_side) {
#pragma unroll
    for (int j = 0; j < 4; ++j) request(0, j, raw[j]);
  }
  // ONE wait for everything requested above (round 6): the bias, the side operand of row block 0 and -- in the
  // persistent 8-phase kernel -- the LDS-DMA units of the next tile that are still in flight (they must have landed
  // before the first store is issued: the next tile's counted waits then never have a store among the operations they
  // leave in flight).  Before, these were up to three round trips one after the other in front of every tile's
  // first row group (drain, then the bias, then the side operand).
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt / lgkmcnt untouched
  b4[0] = bld.x; b4[1] = bld.y; b4[2] = bld.z; b4[3] = bld.w;
  // column-sum slabs exist only beside an act' source in the step (the FFN data gradient); the static specialisations
  // without one do not carry the four accumulations per row group (epi_spec_visit sends such a request to the run-time version)
  constexpr bool kSlabs = epi_spec_slabs<S>();
  float csum[4] = {0.f, 0.f, 0.f, 0.f};
  // one row group (rows er + 4j of block mi, this lane's four columns): staged accumulators -> stores
  auto row_group = [&](int mi, int j) {
    float v[4];
    const int row = er + 4 * j;
#if NSP_EPI_ABLATE & 4      // (ablation: no LDS staging -- the lane's own fragment values stand in for the transposed ones)
    const float4 a4 = make_float4(acc[mi][j][0], acc[mi][j][1], acc[mi][j][2], acc[mi][j][3]);
#else
    const float4 a4 = *reinterpret_cast<const float4*>(stage + stage_idx<SWZ>(row, lane & 15));
#endif
    const int m = mrow0 + mi * 16 + row;
    const bool ok = m < p.M && colok;              // only the stores are predicated
    const long long off = off0 + (long long)(mi * 4 + j) * ldc4;
    const uint4 sd = raw[j];
#ifndef NSP_HOST_EMULATION
    // the side operand is CONSUMED on every path: hipcc sinks its only uses into the predicated store block, and a
    // wave whose rows are all beyond M would otherwise leave the load on the scoreboard -- in a persistent kernel that
    // becomes an s_waitcnt vmcnt(0) (= every store of the epilogue, gfx9 retires in order) at the top of the next tile
    if (SWZ && has_side) asm volatile("" :: "v"(sd.x), "v"(sd.y), "v"(sd.z), "v"(sd.w));
#endif
    if (has_side && mi + 1 < MI) request(mi + 1, j, raw[j]);
    v[0] = a4.x + b4[0]; v[1] = a4.y + b4[1]; v[2] = a4.z + b4[2]; v[3] = a4.w + b4[3];
#if NSP_EPI_ABLATE & 2      // (ablation: no activation / act' / dropout arithmetic -- wrong results, timing only)
    if (ok) {
      if (has_pre) store4(reinterpret_cast<char*>(p.pre_out) + (sorg + (long long)(mi * 16 + 4 * j) * p.ldc) * 2 + (size_t)(loff * 2u), pre_dt, 0, v, 4, true);
      store4(reinterpret_cast<char*>(p.C) + (sorg + (long long)(mi * 16 + 4 * j) * p.ldc) * (c_dt == NSP_DT_BF16 ? 2 : 4) + (size_t)(loff * (c_dt == NSP_DT_BF16 ? 2u : 4u)), c_dt, 0, v, 4, true);
    }
    return;
#endif
    // W32: scalar element offset of this row group's first row + the lane's 32-bit byte offset
    const long long srg = sorg + (long long)(mi * 16 + 4 * j) * p.ldc;
    const unsigned srb = (unsigned)(mi * 16 + 4 * j) * (unsigned)p.ldc;     // (BUF) the same relative to the descriptor's base
    if constexpr (BUF) {
      if (has_pre) {
        bf16x4 h;
        h[0] = (__bf16)v[0]; h[1] = (__bf16)v[1]; h[2] = (__bf16)v[2]; h[3] = (__bf16)v[3];
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(cu32x2_t, h), rpre, lo2 + srb * 2u, 0, AUXS);
      }
    } else if (has_pre && ok) {
      if constexpr (W32) store4(reinterpret_cast<char*>(p.pre_out) + srg * 2 + (size_t)(loff * 2u), pre_dt, 0, v, 4, true);
      else store4(p.pre_out, pre_dt, off, v, 4, true);
    }
    if (act != NSP_ACT_NONE) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = nsp_act(v[e], act);
    }
    if (has_dact) {
      float d[4];
      if (side16) {
        d[0] = __uint_as_float(sd.x << 16); d[1] = __uint_as_float(sd.x & 0xFFFF0000u);
        d[2] = __uint_as_float(sd.y << 16); d[3] = __uint_as_float(sd.y & 0xFFFF0000u);
      } else {
        d[0] = __uint_as_float(sd.x); d[1] = __uint_as_float(sd.y);
        d[2] = __uint_as_float(sd.z); d[3] = __uint_as_float(sd.w);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= nsp_dact(d[e], dact);
    }
    if (!(W32 && drop)) {       // (W32 with dropout: alpha rides in the keep scale)
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= p.alpha;
    }
    if (drop) {
      float kp[4];
      if constexpr (S::kStatic) {
        // (p.offset + off) is even here (the dispatcher checks p.offset; ldc % 4 == 0): the two-mixes-per-four form of
        // nsp_keep_scale4 without its per-lane parity branch.  The counter (p.offset + off) >> 1 = SCALAR part (this row
        // group's origin) + lane part (loff >> 1), both halves exact because both terms are even; nsp_hash_u32 starts
        // with x = lo ^ K(hi), K = seed mix ^ ((hi ^ seed_hi) * c1 + c2): K for hi and hi + 1 are scalars, the carry of
        // the 32-bit add picks one; the second counter is the first + 1 with the first even, i.e. x ^ 1, same hi.
        // Bit-identical masks to nsp_keep_scale4 (tests/test_gemm_epilogues_gpu.py compares with the oracle's masks).
        const unsigned long long cnt = (p.offset + (unsigned long long)srg) >> 1;
        const uint32_t clo = (uint32_t)cnt, chi = (uint32_t)(cnt >> 32);
        const uint32_t smix = (uint32_t)p.seed * 0x9E3779B9u, shi = (uint32_t)(p.seed >> 32);
        const uint32_t k0 = smix ^ ((chi ^ shi) * 0x85EBCA6Bu + 0x632BE5ABu);
        uint32_t k1 = smix ^ (((chi + 1u) ^ shi) * 0x85EBCA6Bu + 0x632BE5ABu);
#ifndef NSP_HOST_EMULATION
        asm volatile("" : "+s"(k1));   // (opaque: otherwise the select below is sunk into a per-lane hi word + v_mul_lo_u32 again)
#endif
        const uint32_t lo = clo + (loff >> 1);
        uint32_t h0 = lo ^ (lo < clo ? k1 : k0);
        uint32_t h1 = h0 ^ 1u;
        h0 ^= h0 >> 16; h0 *= 0x85EBCA6Bu; h0 ^= h0 >> 13; h0 *= 0xC2B2AE35u; h0 ^= h0 >> 16;
        h1 ^= h1 >> 16; h1 *= 0x85EBCA6Bu; h1 ^= h1 >> 13; h1 *= 0xC2B2AE35u; h1 ^= h1 >> 16;
        const float kscale = keep_inv * p.alpha;
        kp[0] = (h0 & 0xFFFFu) < keep_thr ? 0.f : kscale;
        kp[1] = (h0 >> 16) < keep_thr ? 0.f : kscale;
        kp[2] = (h1 & 0xFFFFu) < keep_thr ? 0.f : kscale;
        kp[3] = (h1 >> 16) < keep_thr ? 0.f : kscale;
      } else {
        nsp_keep_scale4(p.seed, p.offset + (unsigned long long)off, p.dropout_p, kp);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= kp[e];
    }
    if (has_res) {
      v[0] += __uint_as_float(sd.x); v[1] += __uint_as_float(sd.y);
      v[2] += __uint_as_float(sd.z); v[3] += __uint_as_float(sd.w);
    }
    if constexpr (BUF) {
      if (c_dt == NSP_DT_BF16) {
        bf16x4 h;
        h[0] = (__bf16)v[0]; h[1] = (__bf16)v[1]; h[2] = (__bf16)v[2]; h[3] = (__bf16)v[3];
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(cu32x2_t, h), rc, lo2 + srb * 2u, 0, AUXS);
      } else {
        const f32x4 q = {v[0], v[1], v[2], v[3]};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(cu32x4_t, q), rc, lo4 + srb * 4u, 0, AUXS);
      }
      if constexpr (kSlabs) {
#pragma unroll
        for (int e = 0; e < 4; ++e) csum[e] += ok ? v[e] : 0.f;
      }
    } else if (ok) {
      if constexpr (W32) {
        if (c_dt == NSP_DT_BF16) store4(reinterpret_cast<char*>(p.C) + srg * 2 + (size_t)(loff * 2u), c_dt, 0, v, 4, true);
        else store4(reinterpret_cast<char*>(p.C) + srg * 4 + (size_t)(loff * 4u), c_dt, 0, v, 4, true);
      } else {
        store4(p.C, c_dt, off, v, 4, true);
      }
      if constexpr (kSlabs) {
#pragma unroll
        for (int e = 0; e < 4; ++e) csum[e] += v[e];
      }
    }
  };
  // MEASURED AND REMOVED (round 3): finishing row groups in pairs with a DPP lane swap so that every bf16 image
  // is written with dwordx4 instead of dwordx2 stores (CDNA guide T21) changed nothing (x0.98-1.00 on all ten
  // GEMM configurations of the step, profiles/r03f_gemm_step_ab.log): this epilogue is not store-issue-bound.
  // 102400 x 2048 x 512 with two bf16 images = 840 MB written in the ~265 us the epilogue adds to the main loop =
  // 3.2 TB/s of pure writes, i.e. the chip-wide write phase runs near what HBM3E sustains for writes (a copy's
  // 6.3 TB/s is half reads) while the MFMA pipes idle; the lever is overlapping that phase with other tiles'
  // main loops, not the store width.
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
      *reinterpret_cast<float4*>(stage + stage_idx<SWZ>(fr, ni * 4 + fg)) =
          make_float4(acc[mi][ni][0], acc[mi][ni][1], acc[mi][ni][2], acc[mi][ni][3]);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // one row group at a time (interleaving four of them costs ~24 VGPRs) unless the build says otherwise
      if (!BUF || j % (SWZ ? NSP_EPI_RG_GROUP_8P : NSP_EPI_RG_GROUP_128) == 0)
        __builtin_amdgcn_sched_barrier(0);
      row_group(mi, j);
    }
    __builtin_amdgcn_wave_barrier();
    // column-sum slabs: one slab row per block of FB row blocks (64 rows, or the whole wave tile when it is shorter)
    constexpr int FB = MI < 4 ? MI : 4;
    if ((mi % FB) == FB - 1 && kSlabs && p.epi_f3) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        csum[e] += __shfl_xor(csum[e], 16, 64);
        csum[e] += __shfl_xor(csum[e], 32, 64);
      }
      if (lane < 16 && colok)
        *reinterpret_cast<float4*>(p.epi_f3 + (long long)((mrow0 + (mi - (FB - 1)) * 16) / (16 * FB)) * p.N + n) =
            make_float4(csum[0], csum[1], csum[2], csum[3]);
#pragma unroll
      for (int e = 0; e < 4; ++e) csum[e] = 0.f;
    }
  }
  // The run-time version can leave a (conditional, never consumed) side-operand load on the compiler's scoreboard.  In a
  // PERSISTENT caller all epilogue variants merge at the tile loop's back edge, and the merged state made hipcc put an
  // s_waitcnt vmcnt(0) -- which on gfx9 also waits for every store of the epilogue -- in front of the next tile's
  // first register write.  Resolving it here keeps that wait on the rare path.
  if constexpr (!S::kStatic && SWZ) __builtin_amdgcn_s_waitcnt(0x0F70);
}

// picks the specialisation for the epilogues of the training step (bf16 mode); SPECIALISE = false keeps a
// kernel on the run-time version only (compile time / code size of the kernels that rarely see big grids)
// what the static specialisations assume beyond their EpiSpec (gemm_epilogue_fast, W32): 32-bit lane offsets inside a
// wave's tile (at most 128 rows x ldc elements x 4 B), column-sum slabs only beside an act' source
__host__ __device__ __forceinline__ bool epi_spec_w32_ok(const nsp_gemm_params& p) {
  return p.ldc < (1ll << 22) && !(p.epi_f3 && !p.dact_src);
}
template <int MI, bool SPECIALISE, bool SWZ = false>
__device__ __forceinline__ void gemm_epilogue_fast_dispatch(const nsp_gemm_params& p, f32x4 (&acc)[MI][4], float* stage,
                                                            int mrow0, int nbase, int lane, long long coff,
                                                            const float4* bias_pre = nullptr) {
  const int n = nbase + (lane & 15) * 4;
#define NSP_EPI(...) do { gemm_epilogue_fast<MI, EpiSpec<__VA_ARGS__>, SWZ>(p, acc, stage, mrow0, n, lane, coff, bias_pre); return; } while (0)
  if constexpr (SPECIALISE) {
    const bool c16 = p.c_dtype == NSP_DT_BF16;
    const bool drop = p.dropout_p > 0.f;
    const bool even = (p.offset & 1ull) == 0ull;
    if ((even || !drop) && epi_spec_w32_ok(p)) {
      if (p.pre_out && p.pre_dtype == NSP_DT_BF16 && c16 && !p.res && !p.dact_src) {           // FFN first linear
        if (p.act == NSP_ACT_SWISH) { if (drop) NSP_EPI(NSP_ACT_SWISH, 0, true, true, false, true); NSP_EPI(NSP_ACT_SWISH, 0, true, true, false, false); }
        if (p.act == NSP_ACT_RELU) { if (drop) NSP_EPI(NSP_ACT_RELU, 0, true, true, false, true); NSP_EPI(NSP_ACT_RELU, 0, true, true, false, false); }
      } else if (p.dact_src && p.dact_dtype == NSP_DT_BF16 && c16 && !p.res && !p.pre_out && p.act == NSP_ACT_NONE) {
        if (p.dact == NSP_ACT_SWISH) { if (drop) NSP_EPI(0, NSP_ACT_SWISH, true, false, false, true); NSP_EPI(0, NSP_ACT_SWISH, true, false, false, false); }
        if (p.dact == NSP_ACT_RELU) { if (drop) NSP_EPI(0, NSP_ACT_RELU, true, false, false, true); NSP_EPI(0, NSP_ACT_RELU, true, false, false, false); }
        if (p.dact == NSP_ACT_TANH_OUT && !drop) NSP_EPI(0, NSP_ACT_TANH_OUT, true, false, false, false);
      } else if (p.res && !c16 && !p.pre_out && !p.dact_src && p.act == NSP_ACT_NONE) {        // residual branches
        if (drop) NSP_EPI(0, 0, false, false, true, true);
        NSP_EPI(0, 0, false, false, true, false);
      } else if (!p.res && !p.pre_out && !p.dact_src && p.act == NSP_ACT_NONE && !drop) {      // plain (+ bias)
        if (c16) NSP_EPI(0, 0, true, false, false, false);
        NSP_EPI(0, 0, false, false, false, false);
      }
    }
  }
#undef NSP_EPI
  gemm_epilogue_fast<MI, EpiRuntime, SWZ>(p, acc, stage, mrow0, n, lane, coff, bias_pre);
}

// ---- shared epilogue (see the comment inside): acc[mi][ni] -> global with full-line accesses
// GENERIC = false: the kernel is only launched when the fast path applies (the launcher's `fast_epi`); compiling both paths into one kernel costs 24 VGPRs = one workgroup per CU.
// SPEC: compile the EpiSpec specialisations of the fast path into this kernel (default: only where GENERIC is off).
template <int MI, bool GENERIC = true, bool SPEC = !GENERIC>  // MI 16-row fragments per wave along M (wave tile = 16*MI x 64)
__device__ __forceinline__ void gemm_epilogue(const nsp_gemm_params& p, f32x4 (&acc)[MI][4],
                                              unsigned char* smem, int m0, int n0, int wm, int wn,
                                              int lane, int wave, long long coff, int c_vec,
                                              const float4* bias_pre = nullptr) {
  if constexpr (GENERIC) {
    if (p.epi_mode != NSP_EPI_NONE) {
      rnnt_epilogue<MI>(p, acc, smem, m0, n0, wm, wn, lane, wave);
      return;
    }
  }
  const int fr = lane & 15, fg = lane >> 4;
  // ---- epilogue.  The MFMA leaves lane (fr, fg) with C[m0+..+fr][n .. n+3]: storing that
  // directly makes every store instruction touch 16 different rows with 16..64 B each
  // (partial cache lines).  Each wave therefore stages 16 rows x 64 cols through its private
  // LDS slab and re-reads it row-major, so that 16 consecutive lanes cover 256 contiguous
  // bytes of one output row: all epilogue loads (bias, residual, act' source) and stores are
  // full-line, 16 B per lane.
  const bool atomic = p.splitk > 1 && p.c_ss == 0;
  constexpr int SP = 68;  // floats per staged row (64 + 4 pad)
  float* stage = reinterpret_cast<float*>(smem) + wave * (16 * SP);
  if (!GENERIC || (c_vec && (p.N & 3) == 0 && !atomic && !(p.res && p.dact_src))) {
    gemm_epilogue_fast_dispatch<MI, SPEC>(p, acc, stage, m0 + wm * (16 * MI), n0 + wn * 64, lane, coff, bias_pre);
    return;
  }
  if constexpr (!GENERIC) return;
  const int er = lane >> 4, ec = (lane & 15) * 4;  // read-back: row er + 4*j, cols ec..ec+3
  // optional: column sums of the STORED values (the bias gradient of the layer below, when C is that
  // layer's d(pre-activation)): per wave 16*MI rows x 64 cols -> one slab row, no atomics
  float csum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
      *reinterpret_cast<float4*>(stage + fr * SP + ni * 16 + fg * 4) =
          make_float4(acc[mi][ni][0], acc[mi][ni][1], acc[mi][ni][2], acc[mi][ni][3]);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = er + 4 * j;
      const float4 a4 = *reinterpret_cast<const float4*>(stage + row * SP + ec);
      const int m = m0 + wm * (16 * MI) + mi * 16 + row;
      const int n = n0 + wn * 64 + ec;
      if (m >= p.M || n >= p.N) continue;
      const long long off = coff + (long long)m * p.ldc + n;
      float v[4] = {a4.x, a4.y, a4.z, a4.w};
      const int nv = min(4, p.N - n);
      if (atomic) {
        float* c = reinterpret_cast<float*>(p.C) + off;
#pragma unroll
        for (int e = 0; e < 4; ++e) if (e < nv) unsafeAtomicAdd(c + e, v[e] * p.alpha);
        continue;
      }
      const bool vec = c_vec && nv == 4;
      if (p.bias) {
        float b4[4] = {0.f, 0.f, 0.f, 0.f};
        load4(p.bias, NSP_DT_F32, n, b4, nv, vec);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += b4[e];
      }
      if (p.pre_out) store4(p.pre_out, p.pre_dtype, off, v, nv, vec);
      if (p.act != NSP_ACT_NONE) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = nsp_act(v[e], p.act);
      }
      if (p.dact_src) {
        float d[4] = {0.f, 0.f, 0.f, 0.f};
        load4(p.dact_src, p.dact_dtype, off, d, nv, vec);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= nsp_dact(d[e], p.dact);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= p.alpha;
      if (p.dropout_p > 0.f) {
        float kp[4];
        nsp_keep_scale4(p.seed, p.offset + (unsigned long long)off, p.dropout_p, kp);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= kp[e];
      }
      if (p.res) {
        float r4[4] = {0.f, 0.f, 0.f, 0.f};
        load4(p.res, NSP_DT_F32, off, r4, nv, vec);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += r4[e];
      }
      store4(p.C, p.c_dtype, off, v, nv, vec);
#pragma unroll
      for (int e = 0; e < 4; ++e) if (e < nv) csum[e] += v[e];
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (p.epi_f3) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      csum[e] += __shfl_xor(csum[e], 16, 64);
      csum[e] += __shfl_xor(csum[e], 32, 64);
    }
    const int n = n0 + wn * 64 + ec;
    if (lane < 16 && n < p.N) {
      float* dst = p.epi_f3 + (long long)((m0 + wm * (16 * MI)) / (16 * MI)) * p.N + n;
#pragma unroll
      for (int e = 0; e < 4; ++e) if (n + e < p.N) dst[e] = csum[e];
    }
  }
}

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(NTHREADS) void gemm_bf16_kernel(const nsp_gemm_params p, int tiles_m,
                                                             int tiles_n, int c_vec) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * TILE_BYTES];
  unsigned char* smA = smem;
  unsigned char* smB = smem + TILE_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const TileCoord tc = tile_coord(p, tiles_m * tiles_n);
  const int tile = tc.tile, split = tc.split, z1 = tc.z1, z2 = tc.z2;
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const __bf16* A = reinterpret_cast<const __bf16*>(p.A) + z1 * p.a_b1 + z2 * p.a_b2;
  const __bf16* B = reinterpret_cast<const __bf16*>(p.B) + z1 * p.b_b1 + z2 * p.b_b2;
  const long long coff = z1 * p.c_b1 + z2 * p.c_b2 + (p.c_ss ? (long long)split * p.c_ss : 0);
  const long long lda = A_KC ? p.a_rs : p.a_cs;
  const long long ldb = B_KC ? p.b_ns : p.b_ks;

  int kbeg = 0, kend = p.K;
  if (p.splitk > 1) {
    int nkt = (p.K + BK - 1) / BK;
    int per = (nkt + p.splitk - 1) / p.splitk;
    kbeg = split * per * BK;
    kend = min(p.K, (split + 1) * per * BK);
    if (kbeg >= kend) {
      if (!p.c_ss) return;   // atomic accumulation: nothing to add
      kend = kbeg;           // slab mode: an empty split still has to write its zeros
    }
  }

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  Bf16TileLoader<A_KC> la;
  Bf16TileLoader<B_KC> lb;
  la.load(A, lda, m0, p.M, kbeg, kend);
  lb.load(B, ldb, n0, p.N, kbeg, kend);
  const int fr = lane & 15, fg = lane >> 4;

  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    la.store(smA);
    lb.store(smB);
    __syncthreads();
    if (k0 + BK < kend) {
      la.load(A, lda, m0, p.M, k0 + BK, kend);
      lb.load(B, ldb, n0, p.N, k0 + BK, kend);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 af[4], bf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        af[i] = read_frag<A_KC>(smA, wm * 64 + i * 16, s, fr, fg);
        bf[i] = read_frag<B_KC>(smB, wn * 64 + i * 16, s, fr, fg);
      }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[ni], af[mi], acc[mi][ni], 0, 0, 0);
    }
    __syncthreads();
  }

  gemm_epilogue<4>(p, acc, smem, m0, n0, wm, wn, lane, wave, coff, c_vec);
}

// ---- KC x KC with direct-to-LDS loads (global_load_lds_dwordx4): no VGPR staging, no
// ds_write pass (the 8 ds_write_b128 per thread per k-tile of the register-staged loader cost
// about as many LDS cycles as the MFMA work they feed).  The DMA writes lane-linear, so the LDS
// image cannot be padded: rows are exactly 128 B ([128 rows][64 k]) and the 16-B chunk index
// is XOR-swizzled with (row & 7) on the SOURCE address and on the fragment read, which spreads
// the 16 rows of a ds_read_b128 group over all banks.  Requires K % 64 == 0.
// EPI: 0 = standard epilogue (fast path only, see gemm_epilogue), 1 / 2 = the RNN-T LSE / DLOGITS epilogues.
// Three kernels rather than one with a run-time switch: each epilogue then gets the 128-register
// budget of 4 workgroups per CU to itself (one kernel holding all of them spilled into the row loop).
template <int EPI>
__global__ __launch_bounds__(NTHREADS, 4) void gemm_bf16_kk_glds_kernel(const nsp_gemm_params p,
                                                                     int tiles_m, int tiles_n,
                                                                     int c_vec) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 128 * 128];   // A | B image, 16 KB each (4 workgroups = 128 KB per CU)
  unsigned char* smA = smem;
  unsigned char* smB = smem + 128 * 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const TileCoord tc = tile_coord(p, tiles_m * tiles_n);
  const int tile = tc.tile, split = tc.split, z1 = tc.z1, z2 = tc.z2;
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const __bf16* A = reinterpret_cast<const __bf16*>(p.A) + z1 * p.a_b1 + z2 * p.a_b2;
  const __bf16* B = reinterpret_cast<const __bf16*>(p.B) + z1 * p.b_b1 + z2 * p.b_b2;
  const long long coff = z1 * p.c_b1 + z2 * p.c_b2 + (p.c_ss ? (long long)split * p.c_ss : 0);
  int kbeg = 0, kend = p.K;
  if (p.splitk > 1) {
    int nkt = p.K / BK;
    int per = (nkt + p.splitk - 1) / p.splitk;
    kbeg = split * per * BK;
    kend = min(p.K, (split + 1) * per * BK);
    if (kbeg >= kend) {
      if (!p.c_ss) return;   // atomic accumulation: nothing to add
      kend = kbeg;           // slab mode: an empty split still has to write its zeros
    }
  }
#if defined(NSP_GLDS_STAGGER) && !defined(NSP_HOST_EMULATION)
  // (experiment) the four co-resident workgroups of a CU start together and run their main loops and epilogues in step;
  // delay the first generation by (wave slot on its SIMD) x NSP_GLDS_STAGGER x 8128 clocks
  if (blockIdx.x < 1024) {
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    const int slot = hwid & 3;
    for (int i = 0; i < slot * NSP_GLDS_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
  }
#endif
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // per-lane source rows: wave w, instruction i covers tile rows (w*4+i)*8 .. +7
  const int lrow = lane >> 3, lpos = lane & 7;
  const __bf16* asrc[4];
  const __bf16* bsrc[4];
  bool aok[4], bok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 8 + lrow;
    const int chunk = lpos ^ (row & 7);
    aok[i] = (m0 + row) < p.M;
    bok[i] = (n0 + row) < p.N;
    asrc[i] = A + (long long)min(m0 + row, p.M - 1) * p.a_rs + chunk * 8;
    bsrc[i] = B + (long long)min(n0 + row, p.N - 1) * p.b_ns + chunk * 8;
  }
  const int fr = lane & 15, fg = lane >> 4;
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;

  // the epilogue's bias chunk, requested HERE (round 6): it lands during the first k-tile instead of costing every
  // workgroup a round trip between its last MFMA and its first store (4 VGPRs through the main loop)
  float4 bias_pre = make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (EPI == 0) {
    const int nb = n0 + wn * 64 + (lane & 15) * 4;
    if (p.bias && nb < p.N) bias_pre = *reinterpret_cast<const float4*>(p.bias + nb);
  }
  if (EPI == 0) NSP_TRACE_MARK(0);
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      // rows beyond M / N keep whatever the slot held: they only feed outputs that are never stored
      if (aok[i])
        __builtin_amdgcn_global_load_lds((glb_void*)(asrc[i] + k0), (lds_void*)(smA + (wave * 4 + i) * 1024), 16, 0, 0);
      if (bok[i])
        __builtin_amdgcn_global_load_lds((glb_void*)(bsrc[i] + k0), (lds_void*)(smB + (wave * 4 + i) * 1024), 16, 0, 0);
    }
    __syncthreads();
    if (EPI == 0 && k0 == kbeg) NSP_TRACE_MARK(1);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 af[4], bf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ra = wm * 64 + i * 16 + fr, rb = wn * 64 + i * 16 + fr;
        af[i] = *reinterpret_cast<const bf16x8*>(smA + ra * 128 + (((s * 4 + fg) ^ (ra & 7)) << 4));
        bf[i] = *reinterpret_cast<const bf16x8*>(smB + rb * 128 + (((s * 4 + fg) ^ (rb & 7)) << 4));
      }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[ni], af[mi], acc[mi][ni], 0, 0, 0);
    }
    __syncthreads();
  }
  if (EPI == 0) NSP_TRACE_MARK(2);
  if constexpr (EPI == 1) rnnt_epilogue_mode<4, true>(p, acc, smem, m0, n0, wm, wn, lane, wave);
  else if constexpr (EPI == 2) rnnt_epilogue_mode<4, false>(p, acc, smem, m0, n0, wm, wn, lane, wave);
  else gemm_epilogue<4, false>(p, acc, smem, m0, n0, wm, wn, lane, wave, coff, c_vec, &bias_pre);
#if NSP_GEMM_TRACE
  if (EPI == 0) {
    __builtin_amdgcn_s_waitcnt(0x0F70);   // the stores have left (vmcnt(0)) before the last mark
    NSP_TRACE_MARK(3);
  }
#endif
}

// ---- the same KC x KC tile with an NS-stage LDS ring.  The single-stage kernel above hides the
// global->LDS latency only through occupancy (4 workgroups per CU); a grid that puts at most one
// or two workgroups on a CU (M = 3200 / 6400 encoder rows: 100..400 tiles for 256 CUs) then runs
// load -> wait -> 32 MFMAs -> load ...: ~12 % of a CU's peak.  Here NS-1 k-tiles are in flight
// while one is consumed: each wave waits for ITS loads of tile kt with a counted s_waitcnt
// (vmcnt(8 x tiles issued since), never the compiler's vmcnt(0) that __syncthreads attaches when
// LDS-DMA is outstanding), a raw s_barrier makes every wave's part visible and proves the stage
// consumed in the previous iteration free, and only then is that stage re-armed.
// MI = 2 halves the tile along M (64 x 128; wave tile 32 x 64) to double the workgroup count of
// grids that would otherwise leave more than half of the CUs idle.
template <int NS, int MI>
__global__ __launch_bounds__(NTHREADS) void gemm_bf16_kk_ring_kernel(const nsp_gemm_params p, int tiles_m,
                                                                     int tiles_n, int c_vec) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ring[];  // NS x (A tile | B tile 16 KB)
  constexpr int BM_ = 32 * MI;            // rows of the A tile
  constexpr int A_BYTES = BM_ * 128;      // 16 KB (MI = 4) or 8 KB (MI = 2)
  constexpr int STAGE = A_BYTES + 16384;
  constexpr int NA = MI;                  // A-tile DMA instructions per wave per k-tile (8 rows each)
  constexpr int NLOAD = NA + 4;           // loads per lane per k-tile
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const TileCoord tc = tile_coord(p, tiles_m * tiles_n);
  const int tile = tc.tile, split = tc.split, z1 = tc.z1, z2 = tc.z2;
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  const int m0 = tm * BM_, n0 = tn * BN;
  const __bf16* A = reinterpret_cast<const __bf16*>(p.A) + z1 * p.a_b1 + z2 * p.a_b2;
  const __bf16* B = reinterpret_cast<const __bf16*>(p.B) + z1 * p.b_b1 + z2 * p.b_b2;
  const long long coff = z1 * p.c_b1 + z2 * p.c_b2 + (p.c_ss ? (long long)split * p.c_ss : 0);
  int kbeg = 0, kend = p.K;
  if (p.splitk > 1) {
    int nkt_all = p.K / BK;
    int per = (nkt_all + p.splitk - 1) / p.splitk;
    kbeg = split * per * BK;
    kend = min(p.K, (split + 1) * per * BK);
    if (kbeg >= kend) {
      if (!p.c_ss) return;   // atomic accumulation: nothing to add
      kend = kbeg;           // slab mode: an empty split still has to write its zeros
    }
  }
  f32x4 acc[MI][4];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // every lane issues exactly NLOAD LDS-DMA loads per k-tile (rows beyond M / N re-read the last
  // valid row: they only feed outputs that are never stored), so the vmcnt arithmetic is uniform
  const int lrow = lane >> 3, lpos = lane & 7;
  const __bf16* asrc[NA];
  const __bf16* bsrc[4];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int row = (wave * NA + i) * 8 + lrow;
    asrc[i] = A + (long long)min(m0 + row, p.M - 1) * p.a_rs + (lpos ^ (row & 7)) * 8 + kbeg;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 8 + lrow;
    bsrc[i] = B + (long long)min(n0 + row, p.N - 1) * p.b_ns + (lpos ^ (row & 7)) * 8 + kbeg;
  }
  const int fr = lane & 15, fg = lane >> 4;
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;
  const int nkt = (kend - kbeg) / BK;
  auto issue = [&](int kt) {
    unsigned char* sa = ring + (kt % NS) * STAGE;
    const int k0 = kt * BK;
#pragma unroll
    for (int i = 0; i < NA; ++i)
      __builtin_amdgcn_global_load_lds((glb_void*)(asrc[i] + k0), (lds_void*)(sa + (wave * NA + i) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((glb_void*)(bsrc[i] + k0), (lds_void*)(sa + A_BYTES + (wave * 4 + i) * 1024), 16, 0, 0);
  };
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nkt) issue(s);
  for (int kt = 0; kt < nkt; ++kt) {
    const int ahead = min(NS - 2, nkt - 1 - kt);  // k-tiles issued after tile kt so far
    if (NS >= 4 && ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NLOAD) : "memory");
    else if (NS >= 3 && ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLOAD) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kt + NS - 1 < nkt) issue(kt + NS - 1);
    const unsigned char* smA = ring + (kt % NS) * STAGE;
    const unsigned char* smB = smA + A_BYTES;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 af[MI], bf[4];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int ra = wm * (16 * MI) + i * 16 + fr;
        af[i] = *reinterpret_cast<const bf16x8*>(smA + ra * 128 + (((s * 4 + fg) ^ (ra & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int rb = wn * 64 + i * 16 + fr;
        bf[i] = *reinterpret_cast<const bf16x8*>(smB + rb * 128 + (((s * 4 + fg) ^ (rb & 7)) << 4));
      }
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[ni], af[mi], acc[mi][ni], 0, 0, 0);
    }
  }
  __syncthreads();  // the epilogue reuses the ring as its staging area
  // (the 2-stage / full-height ring is what mid-size grids of the step run on: it gets the specialisations too)
  gemm_epilogue<MI, true, (NS == 2 && MI == 4)>(p, acc, ring, m0, n0, wm, wn, lane, wave, coff, c_vec);
}

// ---- DIRECT epilogue (round 4): no LDS.  The MFMA leaves lane (fr, fg) with C[row fr][4 consecutive columns] of every
// 16 x 16 fragment; a store instruction of one fragment therefore covers 16 rows x 64 B (fp32) / 32 B (bf16).  The staged
// epilogue above exists to turn that into full 256-B row segments -- at the price of a ds_write / ds_read round trip and a
// serial chain per 16-row slab, which a kernel with ONE workgroup per CU (nothing else running during its epilogue)
// pays in full: ~4 us per 256 x 256 tile even for a plain bf16 store.  Here every fragment is an independent chain
// (bias -> act -> dropout -> side operand -> store); the four fragments of a row block are written back to back, so the
// XCD's L2 sees the four 64-B pieces of a 256-B row segment within a few hundred cycles and merges them.
// All memory operations are BUFFER instructions with the range check as the predicate (rows >= M, columns >= N get an
// out-of-range offset: loads return 0, stores are dropped): no branches, so hipcc's counted waits survive.
// Static specialisations only; requires N % 4 == 0 and M * ldc * 4 < 2^31 (32-bit byte offsets).
template <class S, int AUX_NT = 2>               // AUX_NT: gfx940+ cache policy of the stores / side loads (2 = nt, 0 = default)
__device__ __forceinline__ void gemm_epilogue_direct(const nsp_gemm_params& p, f32x4 (&acc)[4][4], int mrow0, int ncol,
                                                     int lane) {
  constexpr bool has_dact = S::DACT != NSP_ACT_NONE, has_res = S::RES, has_side = has_dact || has_res;
  constexpr unsigned OOB = 0x80000000u;
  const int fr = lane & 15, fg = lane >> 4;
  // Opaque copies of the scalars everything below is derived from: the persistent caller's tile loop encloses thirteen
  // specialisations of this function, and without the pins hipcc hoists each one's loop-invariant values (buffer
  // descriptors, thresholds, byte counts) in front of the MAIN loop, which then spills (ISA audit, round 4).
  int M_ = p.M, ldc_ = (int)p.ldc;
  float drop_p = p.dropout_p;
#ifndef NSP_HOST_EMULATION
  asm volatile("" : "+s"(M_), "+s"(ldc_), "+s"(drop_p));
#endif
  const long long celems = (long long)M_ * ldc_;
  const unsigned c_bytes = (unsigned)(celems * (S::C16 ? 2 : 4));
  const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, c_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rpre =
      __builtin_amdgcn_make_buffer_rsrc(S::PRE16 ? p.pre_out : p.C, 0, (unsigned)(celems * 2), 0x00020000);
  const void* side_ptr = has_res ? reinterpret_cast<const void*>(p.res) : (has_dact ? p.dact_src : p.C);
  const __amdgpu_buffer_rsrc_t rside = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(side_ptr), 0, (unsigned)(celems * (has_res ? 4 : 2)), 0x00020000);
  const int n0 = ncol + fg * 4;
  constexpr bool HAS_BIAS = !has_dact;     // data gradients carry no bias (the launcher checks): 16 registers less
  float b4[4][4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    const int n = min(n0 + ni * 16, p.N - 4);
    float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
    if (HAS_BIAS && p.bias) b = *reinterpret_cast<const float4*>(p.bias + n);
    b4[ni][0] = b.x; b4[ni][1] = b.y; b4[ni][2] = b.z; b4[ni][3] = b.w;
  }
  const uint32_t keep_thr = (uint32_t)(drop_p * 65536.f);
  const float keep_inv = nsp_rcp(1.f - drop_p);
  // element offset of fragment (mi, ni) = eoff0 + mi * 16 * ldc + ni * 16, or out of range
  const unsigned ldc = (unsigned)ldc_;
  auto elem_off = [&](int mi, int ni) -> unsigned {
    const int m = mrow0 + mi * 16 + fr, n = n0 + ni * 16;
    return (m < M_ && n < p.N) ? (unsigned)m * ldc + (unsigned)n : OOB;
  };
  // side operand of the CURRENT row block, one 16-B (fp32 residual) / 8-B (bf16 act' source) chunk per fragment; the
  // chunk of fragment (mi + 1, ni) is requested into the same registers right after fragment (mi, ni) consumed its
  // copy and BEFORE that fragment's store (in-order vmcnt: a wait for it then only covers stores a whole block old)
  cu32x4_t side[4];
  auto request = [&](int mi, int ni) {
    const unsigned eo = elem_off(mi, ni);
    if (has_res) {
      side[ni] = __builtin_amdgcn_raw_buffer_load_b128(rside, eo == OOB ? OOB : eo * 4u, 0, AUX_NT);
    } else {
      const cu32x2_t h = __builtin_amdgcn_raw_buffer_load_b64(rside, eo == OOB ? OOB : eo * 2u, 0, AUX_NT);
      side[ni][0] = h[0]; side[ni][1] = h[1];
    }
  };
  if (has_side) {
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) request(0, ni);
  }
  constexpr bool SLABS = has_dact;     // column-sum slabs come with d(pre-activation) outputs only (the caller checks)
  float cs[4][4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni)
#pragma unroll
    for (int e = 0; e < 4; ++e) cs[ni][e] = 0.f;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const unsigned eo = elem_off(mi, ni);
      const bool ok = eo != OOB;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = HAS_BIAS ? acc[mi][ni][e] + b4[ni][e] : acc[mi][ni][e];
      const cu32x4_t sd = side[ni];
      if (has_side && mi + 1 < 4) request(mi + 1, ni);
      if (S::PRE16) {
        bf16x4 h;
        h[0] = (__bf16)v[0]; h[1] = (__bf16)v[1]; h[2] = (__bf16)v[2]; h[3] = (__bf16)v[3];
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(cu32x2_t, h), rpre, ok ? eo * 2u : OOB, 0, AUX_NT);
      }
      if (S::ACT != NSP_ACT_NONE) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = nsp_act(v[e], S::ACT);
      }
      if (has_dact) {
        float d[4];
        d[0] = __uint_as_float(sd[0] << 16); d[1] = __uint_as_float(sd[0] & 0xFFFF0000u);
        d[2] = __uint_as_float(sd[1] << 16); d[3] = __uint_as_float(sd[1] & 0xFFFF0000u);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= nsp_dact(d[e], S::DACT);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= p.alpha;
      if (S::DROP) {
        // (p.offset + element offset) is even (the dispatcher checks p.offset; n is a multiple of 4): two mixes per four
        const unsigned long long base = (p.offset + (unsigned long long)(ok ? eo : 0u)) >> 1;
        const uint32_t h0 = nsp_hash_u32(p.seed, base), h1 = nsp_hash_u32(p.seed, base + 1ull);
        v[0] *= (h0 & 0xFFFFu) < keep_thr ? 0.f : keep_inv;
        v[1] *= (h0 >> 16) < keep_thr ? 0.f : keep_inv;
        v[2] *= (h1 & 0xFFFFu) < keep_thr ? 0.f : keep_inv;
        v[3] *= (h1 >> 16) < keep_thr ? 0.f : keep_inv;
      }
      if (has_res) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += __uint_as_float(sd[e]);
      }
      if (S::C16) {
        bf16x4 h;
        h[0] = (__bf16)v[0]; h[1] = (__bf16)v[1]; h[2] = (__bf16)v[2]; h[3] = (__bf16)v[3];
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(cu32x2_t, h), rc, ok ? eo * 2u : OOB, 0, AUX_NT);
      } else {
        cu32x4_t q;
        q[0] = __float_as_uint(v[0]); q[1] = __float_as_uint(v[1]); q[2] = __float_as_uint(v[2]); q[3] = __float_as_uint(v[3]);
        __builtin_amdgcn_raw_buffer_store_b128(q, rc, ok ? eo * 4u : OOB, 0, AUX_NT);
      }
      if (SLABS) {
#pragma unroll
        for (int e = 0; e < 4; ++e) cs[ni][e] += ok ? v[e] : 0.f;
      }
      __builtin_amdgcn_sched_barrier(0);   // one fragment at a time: letting the scheduler interleave sixteen chains spills
    }
  }
  if (SLABS && p.epi_f3) {
    // column sums of the stored values over this wave's 64 rows: 4 mi in registers, 16 rows across the lanes of a
    // 16-lane row group (xor 1, 2, 4, 8), one slab row per 64-row block
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float x = cs[ni][e];
        x += __shfl_xor(x, 1, 64);
        x += __shfl_xor(x, 2, 64);
        x += __shfl_xor(x, 4, 64);
        x += __shfl_xor(x, 8, 64);
        cs[ni][e] = x;
      }
    if (fr == 0) {
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const int n = n0 + ni * 16;
        if (n < p.N)
          *reinterpret_cast<float4*>(p.epi_f3 + (long long)(mrow0 >> 6) * p.N + n) =
              make_float4(cs[ni][0], cs[ni][1], cs[ni][2], cs[ni][3]);
      }
    }
  }
}

// calls f(EpiSpec<...>{}) for the specialisation that matches the call's epilogue (the same table as
// gemm_epilogue_fast_dispatch) and returns true, or returns false
template <class F>
__host__ __device__ __forceinline__ bool epi_spec_visit(const nsp_gemm_params& p, F&& f) {
  const bool c16 = p.c_dtype == NSP_DT_BF16;
  const bool drop = p.dropout_p > 0.f;
  const bool even = (p.offset & 1ull) == 0ull;
  if (!(even || !drop) || !epi_spec_w32_ok(p)) return false;
#define NSP_VISIT(...) do { f(EpiSpec<__VA_ARGS__>{}); return true; } while (0)
  if (p.pre_out && p.pre_dtype == NSP_DT_BF16 && c16 && !p.res && !p.dact_src) {           // FFN first linear
    if (p.act == NSP_ACT_SWISH) { if (drop) NSP_VISIT(NSP_ACT_SWISH, 0, true, true, false, true); NSP_VISIT(NSP_ACT_SWISH, 0, true, true, false, false); }
    if (p.act == NSP_ACT_RELU) { if (drop) NSP_VISIT(NSP_ACT_RELU, 0, true, true, false, true); NSP_VISIT(NSP_ACT_RELU, 0, true, true, false, false); }
  } else if (p.dact_src && p.dact_dtype == NSP_DT_BF16 && c16 && !p.res && !p.pre_out && p.act == NSP_ACT_NONE) {
    if (p.dact == NSP_ACT_SWISH) { if (drop) NSP_VISIT(0, NSP_ACT_SWISH, true, false, false, true); NSP_VISIT(0, NSP_ACT_SWISH, true, false, false, false); }
    if (p.dact == NSP_ACT_RELU) { if (drop) NSP_VISIT(0, NSP_ACT_RELU, true, false, false, true); NSP_VISIT(0, NSP_ACT_RELU, true, false, false, false); }
    if (p.dact == NSP_ACT_TANH_OUT && !drop) NSP_VISIT(0, NSP_ACT_TANH_OUT, true, false, false, false);
  } else if (p.res && !c16 && !p.pre_out && !p.dact_src && p.act == NSP_ACT_NONE) {        // residual branches
    if (drop) NSP_VISIT(0, 0, false, false, true, true);
    NSP_VISIT(0, 0, false, false, true, false);
  } else if (!p.res && !p.pre_out && !p.dact_src && p.act == NSP_ACT_NONE && !drop) {      // plain (+ bias)
    if (c16) NSP_VISIT(0, 0, true, false, false, false);
    NSP_VISIT(0, 0, false, false, false, false);
  }
#undef NSP_VISIT
  return false;
}

// ---- KC x KC, 256 x 256 x 64 tiles, 8 waves, PHASE-INTERLEAVED main loop (round 4).
// What the per-workgroup trace of the 128 x 128 kernel said (profiles/r03am): a k-tile costs a workgroup ~3.1 us for
// 0.23 us of MFMA work, because it waits for one LDS-DMA round trip per k-tile; the first 256 x 256 kernel above has the
// same one-tile-in-flight structure (vmcnt(0) + barrier per k-tile).  This kernel never drains the DMA queue inside a
// tile:
//   * 8 waves = 2 (M) x 4 (N), wave tile 128 x 64 = four 32-row QUADRANTS (2 x 4 MFMA fragments each); a k-tile is four
//     PHASES, phase q = {ds_read the A rows of quadrant q (+ the wave's 64 B rows in phase 0), issue ONE 16-KB load
//     unit (2 LDS-DMA instructions per lane), s_barrier, 16 MFMAs, s_barrier};
//   * the two waves of a SIMD (wave w and w + 4: row halves 0 / 1) run ONE BARRIER APART, so one of them is always in
//     its MFMA segment (s_setprio 1) while the other issues its LDS reads and DMA requests;
//   * load units of a k-tile: 0 / 1 = B rows 0-127 / 128-255, 2 = the A rows of quadrants 0-1 of both row halves,
//     3 = those of quadrants 2-3.  The unit stream runs SIX units (1.5 k-tiles) ahead of the phase that is being
//     multiplied and keeps running across the tile boundary (persistent workgroup, two 64-KB k-tile buffers).
//     Six is the largest distance the two buffers allow: the later wave's reads of phase Q are only known to be
//     complete behind the barrier that ends ITS MFMA segment of phase Q, so a region last read in phase Q may be
//     re-armed from phase Q + 2 on; units 0 and 3 meet that bound exactly (GEMM section).
//   * counted waits only: behind the issue of phase 4t+1 vmcnt(8) (unit 3 of k-tile t landed, four newer units in
//     flight), behind phase 4t+3 vmcnt(6) (units 0-2 of k-tile t+1).  A wave's wait is followed by the barrier that
//     closes its read segment, so every reader of the NEXT phase has a barrier between all eight waves' waits and its
//     ds_read.
//   * epilogue: both wave groups re-align (the early one waits one barrier), drain their own DMA queue (the next
//     tile's first six units were requested during the last six phases and keep landing meanwhile) and run the
//     standard fused epilogue through a private 4-KB staging slab per wave (XOR-swizzled, stage_idx<true>) -- the ring
//     is never used for staging, both k-tile buffers stay armed.
// Requires K % 128 == 0 (two k-tiles per loop iteration: buffer indices are compile-time), one problem, no split-K,
// the fast epilogue's conditions, operand extents below 2^32 elements.  LDS 128 KB + 32 KB = all of the CU's 160 KB.
// ---- weight-gradient (RR) operand images of the 8-phase kernel: sub-images [64 k][64 columns], 128-B rows.  A transposed
// read (ds_read_b64_tr_b16) of a 32-lane half touches 8 k-rows (k & 3, k bit 3) x 32 B; two 128-B rows share a 256-B bank
// window, so the 32-B slot is (k & 1) * 4 + ((column slot) ^ (k bit 1 | k bit 3 << 1)): eight distinct slots, conflict-free.
__device__ __forceinline__ int rr8_swz(int krow) { return (((krow >> 1) & 1) | (((krow >> 3) & 1) << 1)) << 1; }
// The transposed reads are issued from INLINE ASM.  Behind the builtin (__builtin_amdgcn_ds_read_tr16_b64) hipcc's wait
// insertion assumes the read may alias every LDS-DMA load in flight and puts s_waitcnt vmcnt(0) in front of it -- the
// ISA of the round-1..3 weight-gradient ring kernels shows exactly that wait ahead of their first transposed read, i.e.
// their "2-stage rings" never had a load in flight across a read.  The compiler does not count an asm load (CDNA guide
// 5.7 item 1), so the two halves of a fragment are written by two statements and every destination is named "+v" in the
// wait statement in front of its first consumer (form (ii) there).
struct RR8Frag { bf16x4 lo, hi; };
__device__ __forceinline__ void rr8_read(RR8Frag& f, const unsigned char* addr) {
#ifdef NSP_HOST_EMULATION
  typedef bf16x4 lds_bf16x4;
  f.lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(addr));
  f.hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(addr + 4 * 128));
#else
  typedef __attribute__((address_space(3))) unsigned char lds_uchar;
  const unsigned a = (unsigned)(uintptr_t)((lds_uchar*)addr);
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.lo) : "v"(a) : "memory");
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:512" : "=v"(f.hi) : "v"(a) : "memory");
#endif
}
__device__ __forceinline__ bf16x8 rr8_join(const RR8Frag& f) {
  return __builtin_shufflevector(f.lo, f.hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __attribute__((aligned(256))) unsigned int nsp_zero_line[64];   // 256 B of zeros (static storage)

// One kernel per epilogue specialisation S (EpiSpec<...>: the direct epilogue; EpiRuntime: everything else through the
// staged run-time version): with all of them behind a run-time switch inside ONE persistent kernel, hipcc hoisted each
// variant's tile-invariant values in front of the main loop and spilled ~800 registers (each variant alone: 226-232,
// no scratch).  The launcher picks S with the table of epi_spec_visit.
// VAR (development A/B, NSP_GEMM_8P_VAR): bit 0 = no s_setprio around the MFMA segments, bit 1 = static priority 1 for the
// second (later dispatched) wave half instead, bit 2 = staged (LDS) epilogue for S instead of the direct one.
template <class S, int VAR = 0, bool RR = false, bool SK = false>
__global__ __launch_bounds__(512, 2) void gemm_bf16_kk8p_kernel(const nsp_gemm_params p, int tiles_m, int tiles_n,
                                                                int c_vec, unsigned char* skws = nullptr,
                                                                unsigned sk_epoch = 0u) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ring[];  // 2 x (A 32 KB | B 32 KB) | 8 x 4 KB staging
  constexpr int BUF = 65536, A_BYTES = 32768, STAGING = 131072;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: LDS piece bases (M0) and row halves stay in SGPRs
  const int wr = wave >> 2, wc = wave & 3;
  const int ntiles = tiles_m * tiles_n;
  // KK: persistent over the tile list.  RR (weight gradients): ONE (tile, reduction split) per workgroup on the flat
  // split-K grid of tile_coord; the split's k-tiles [ktbeg, ktbeg + 2 nit) -- an even count, k-rows beyond p.K read zeros
  int nit = p.K >> 7;                          // loop iterations of two k-tiles
  int ktbeg = 0, rr_tile = 0;
  long long coff = 0;
  if (RR) {
    const TileCoord tc = tile_coord(p, ntiles);
    rr_tile = tc.tile;
    const int nkt_pad = ((p.K + 127) >> 7) << 1;
    const int per = (((nkt_pad + p.splitk - 1) / p.splitk) + 1) & ~1;
    ktbeg = min(tc.split * per, nkt_pad);
    nit = (min(ktbeg + per, nkt_pad) - ktbeg) >> 1;
    coff = p.c_ss ? (long long)tc.split * p.c_ss : 0;
  }
  const int nit_full = nit;                    // iterations of a whole tile (KK) / of this split (RR)
  const int xq = blockIdx.x >> 3, xx = blockIdx.x & 7, per_xcd = gridDim.x >> 3;
  // (a staggered start of the XCDs' workgroups -- NSP_GEMM_8P_STAGGER, round 4 -- measured nothing and was removed:
  // profiles/r04v_gemm_8p_stagger_negative.log)
  // Tile order (KK).  Narrow outputs (< 16 tile columns: every GEMM of the training step): the 32 workgroups of an XCD
  // take 32 CONSECUTIVE tiles of the n-fastest list -- a few A panels x all their column tiles, so every A panel crosses
  // the fabric once and the whole weight matrix stays in the XCD's L2.  Wide outputs (N >= 4096: the 8192^3 square the
  // roofline quotes beside the library GEMM): 32 consecutive tiles are ONE A panel against 32 different B panels, and
  // every B panel is fetched again for every row of tiles -- measured 5.2 TB/s of fabric traffic, the kernel's limit
  // there.  The XCD then takes a 4 x 8 SUPER-TILE (4 A panels + 8 B panels for 32 tiles: 12 panel loads instead of 33).
  const bool wide = !RR && tiles_n >= 16 && per_xcd == 32;
  const int sn_cnt = (tiles_n + 7) >> 3, s_cnt = ((tiles_m + 3) >> 2) * sn_cnt;
  auto tile_of = [&](int step) {               // KK: gridDim.x % 8 == 0 (launcher); returns ntiles when the list is exhausted
    if constexpr (RR) return step == 0 ? rr_tile : ntiles;
    if (!wide) return min(per_xcd * (8 * step + xx) + xq, ntiles);
    const int sidx = 8 * step + xx;
    if (sidx >= s_cnt) return ntiles;
    const int sm = sidx / sn_cnt, sn = sidx - sm * sn_cnt;
    const int tm_ = sm * 4 + (xq >> 3), tn_ = sn * 8 + (xq & 7);
    return (tm_ < tiles_m && tn_ < tiles_n) ? tm_ * tiles_n + tn_ : -1;     // -1: a hole of a ragged super-tile, skipped
  };
  auto next_step = [&](int from) {             // first step >= from whose tile exists (or the end of the list)
    int st_ = from;
    if constexpr (!RR) {
      while (tile_of(st_) < 0) ++st_;
    }
    return st_;
  };
  // ---- STREAM-K schedule (round 6, KK, skws != nullptr).  The tile list of a narrow output rarely fills its last
  // round (M = 102 400, N = 512: 800 tiles on 256 workgroups = 3.125 rounds, the fourth at 12 % occupancy = 22 % of the
  // launch).  Here an XCD owns ntiles / 8 consecutive tiles and its workgroups equal shares of those tiles' LOOP
  // ITERATIONS (two k-tiles each): workgroup q takes iterations [q W, (q + 1) W) of the XCD's Tx * U.  With W >= U
  // (the launcher checks) a tile is shared by at most two neighbours q, q + 1.  A workgroup runs, in this order:
  //   1. GIVE  -- the head [0, xb) of the tile its range ends in: accumulators stored raw (fragment layout, write-
  //      through) to its 256-KB workspace slot, then its flag = this launch's epoch;
  //   2. its whole tiles;
  //   3. TAKE  -- the tail [ya, U) of the tile its range starts in: accumulators START from the left neighbour's
  //      partial (produced first thing over there, so the flag is up long before) and end in the normal epilogue.
  //      (TAKE second instead of last measured 241 vs 217 us on 102 400 x 512 x 2048: the poll waited for the neighbour.)
  // A producer has the lower blockIdx of the pair (same XCD, q - 1), so it is never dispatched after its consumer.
  struct Item { int tile, kb, nit, kind; };            // kind 0 whole tile, 1 give, 2 take
  // (SK is a TEMPLATE parameter: with the schedule behind a run-time flag in the one kernel, builds of it returned wrong
  // 4 x 4 blocks from the plain tile-list schedule on the device -- a different accumulator allocation each time the dead
  // code changed, never on the emulator, never with the flag a compile-time false; profiles/r06_gemm_epilogue.log.  The
  // tile-list kernels are therefore the same code as before, and the stream-K twins exist for the specialisations that
  // the step's N = 512 products use.)
  constexpr bool sk = SK && !RR;
  int sk_ta = 0, sk_tb = 0, sk_ya = 0, sk_xb = 0, sk_n = 0, sk_full0 = 0;
  const int sk_gidx = xx * per_xcd + xq;
  if (sk) {
    const int Tx = ntiles >> 3, U = nit_full;
    const int total = Tx * U, W = (total + per_xcd - 1) / per_xcd;
    const int r0 = min(xq * W, total), r1 = min(r0 + W, total);
    if (r0 >= r1) return;
    sk_ta = r0 / U; sk_ya = r0 - sk_ta * U;
    sk_tb = r1 / U; sk_xb = r1 - sk_tb * U;
    sk_full0 = sk_ta + (sk_ya > 0 ? 1 : 0);
    sk_n = (sk_xb > 0 ? 1 : 0) + (sk_ya > 0 ? 1 : 0) + max(sk_tb - sk_full0, 0);
  }
  // local tile index of the XCD -> tile: COLUMN-major inside the XCD's row panels (all its tiles of column 0, then column 1, ...),
  // so that the tiles sharing an A panel are in flight on DIFFERENT workgroups at the same time (one fetch from HBM, the
  // other hits L2) -- with the n-fastest order both were consecutive items of one workgroup, the panel had left L2 in
  // between, and inside the step (cold operands) the schedule lost what it gained alone
  const int sk_rpx = sk ? (ntiles >> 3) / tiles_n : 1;       // row panels per XCD (the launcher: a whole number)
  auto sk_tile = [&](int l) { const int n_ = l / sk_rpx; return (xx * sk_rpx + (l - n_ * sk_rpx)) * tiles_n + n_; };
  auto item_of = [&](int step) -> Item {
    if (!sk) return Item{tile_of(step), 0, nit_full, 0};
    if (step >= sk_n) return Item{ntiles, 0, nit_full, 0};
    int i = step;
    if (sk_xb > 0) { if (i == 0) return Item{sk_tile(sk_tb), 0, sk_xb, 1}; --i; }
    if (sk_ya > 0 && step == sk_n - 1) return Item{sk_tile(sk_ta), sk_ya, nit_full - sk_ya, 2};     // (LAST: the neighbour's partial is long there)
    return Item{sk_tile(sk_full0 + i), 0, nit_full, 0};
  };
  const char* A = reinterpret_cast<const char*>(p.A);
  const char* B = reinterpret_cast<const char*>(p.B);
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;
  const int fr = lane & 15, fg = lane >> 4;
  // ---- DMA sources of the tile the unit stream is in.  Piece pc = wave * 2 + i of a unit = 8 rows x 128 B; lane
  // (lrow, lpos) fetches the 16-B chunk lpos ^ lrow of its row (source-side swizzle: the DMA writes lane-linear)
  const int lrow = lane >> 3, lpos = lane & 7;
  const unsigned sw = (unsigned)((lpos ^ lrow) * 16);
  unsigned aoff[2][2], boff[2][2];           // [unit 2 / 3 resp. 0 / 1][piece]: BYTE offsets from A / B (scalar base + 32-bit lane offset)
  // RR images: a load unit = two 8-KB sub-images [64 k][64 columns] (128-B rows; unit 2 / 3: quadrant pair 0-1 / 2-3 of
  // row half 0 and 1; unit 0 / 1: the B columns of waves wc = 0, 1 / 2, 3), piece pc = wave * 2 + i = 8 k-rows of
  // sub-image pc >> 3; the 16-B chunk index is XOR-ed with rr8_swz(k) on the source address and on the transposed read
  const char* Atile = A;                     // (KK) first row of the A panel of the tile the unit stream is in
  auto set_src = [&](int tile) {
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    if (RR) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int pc = wave * 2 + i, krow = (pc & 7) * 8 + lrow;
          const int chunk = (lpos ^ rr8_swz(krow)) * 8;
          // columns beyond M / N re-read the tile's first chunk: they only feed outputs that are never stored
          int ma = tm * 256 + (pc >> 3) * 128 + u * 64 + chunk;
          int nb = tn * 256 + (u * 2 + (pc >> 3)) * 64 + chunk;
          if (ma >= p.M) ma = tm * 256;
          if (nb >= p.N) nb = tn * 256;
          aoff[u][i] = (unsigned)krow * (unsigned)(2 * p.a_cs) + (unsigned)(2 * ma);
          boff[u][i] = (unsigned)krow * (unsigned)(2 * p.b_ks) + (unsigned)(2 * nb);
        }
      return;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r = (wave * 2 + i) * 8 + lrow;                       // 0..127 inside the unit
        const int arow = (r & 63) + ((r >> 6) << 7) + u * 64;          // quadrants 2u, 2u+1 of row half r >> 6
        aoff[u][i] = (unsigned)min(arow, p.M - 1 - tm * 256) * (unsigned)(2 * p.a_rs) + sw;    // relative to the tile's first row
        boff[u][i] = (unsigned)min(tn * 256 + u * 128 + r, p.N - 1) * (unsigned)(2 * p.b_ns) + sw;
      }
    // the scalar base moves with the tile: the 32-bit lane offsets then span 256 rows, whatever M is (the RNN-T joint's
    // data gradient reads a 3.6 M x 1024 bf16 operand: 7.4 GB)
    Atile = A + (long long)tm * 256 * p.a_rs * 2;
  };
  // (RR) descriptors rebased to the split's first k-row, so that 32-bit offsets only have to span ONE split (the RNN-T
  // output layer reduces over 3.6 M rows x 2 KB); they end at the last k-row of the problem (clipped to 4 GB - 1: the
  // clip can only bite in a split that does not reach the end anyway)
  const long long rows_left = RR ? (long long)p.K - (long long)ktbeg * BK : 0;
  const long long ext_a = rows_left > 0 ? rows_left * p.a_cs * 2 : 0, ext_b = rows_left > 0 ? rows_left * p.b_ks * 2 : 0;
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(A) + (RR ? (long long)ktbeg * BK * p.a_cs * 2 : 0), 0, (unsigned)min(ext_a, 0xFFFFFFFFll), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(B) + (RR ? (long long)ktbeg * BK * p.b_ks * 2 : 0), 0, (unsigned)min(ext_b, 0xFFFFFFFFll), 0x00020000);
  // unit j (0..3) of k-tile `kt` of the stream's tile into k-tile buffer `buf`
  auto issue = [&](int j, int buf, int kt) {
    unsigned char* base = ring + buf * BUF;
    if (RR) {
      // BUFFER LDS-DMA: the descriptors end at the last k-row, so the k-rows of a ragged (or padded) last k-tile are
      // out of range and land as ZEROS -- no tail logic; the k-tile offset rides in the scalar offset field
      const int ska = kt * (2 * BK) * (int)p.a_cs, skb = kt * (2 * BK) * (int)p.b_ks;   // relative to the split's first k-row
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int pc = wave * 2 + i;
        const int sub = j < 2 ? 4 + j * 2 + (pc >> 3) : (j - 2) * 2 + (pc >> 3);   // A sub-images 0..3, B sub-images 4..7
        unsigned char* dst = base + sub * 8192 + (pc & 7) * 1024;
        if (j < 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void*)dst, 16, (int)boff[j][i], skb, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_void*)dst, 16, (int)aoff[j - 2][i], ska, 0, 0);
      }
      return;
    }
    const unsigned ko = (unsigned)(kt * (2 * BK));     // added to the 32-bit lane offset: keeps the scalar-base + voffset form
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r0 = (wave * 2 + i) * 8;
      if (j < 2) {
        __builtin_amdgcn_global_load_lds((glb_void*)(B + (boff[j][i] + ko)),
                                         (lds_void*)(base + A_BYTES + (j * 128 + r0) * 128), 16, 0, 0);
      } else {
        const int arow0 = (r0 & 63) + ((r0 >> 6) << 7) + (j - 2) * 64;
        __builtin_amdgcn_global_load_lds((glb_void*)(Atile + (aoff[j - 2][i] + ko)), (lds_void*)(base + arow0 * 128), 16, 0, 0);
      }
    }
  };
  // ---- fragment addresses (bytes inside a k-tile buffer): row (.. + fr), 16-B chunk (4 s + fg) ^ (row & 7); the rows
  // of one lane differ by multiples of 16, so (row & 7) = fr & 7 throughout, s = 1 is the s = 0 address ^ 64, and everything else is an immediate offset
  const int chunk0 = (fg ^ (fr & 7)) << 4;
  const int a_addr[2] = {(wr * 128 + fr) * 128 + chunk0, (wr * 128 + fr) * 128 + (chunk0 ^ 64)};   // [s]; + immediates
  const int b_addr[2] = {A_BYTES + (wc * 64 + fr) * 128 + chunk0, A_BYTES + (wc * 64 + fr) * 128 + (chunk0 ^ 64)};
  // RR: lane (a = fr >> 2, b = fr & 3, g = fg) of a transposed read addresses k-row 32 s + 8 g + a (and + 4), columns
  // cbase + 4 b .. + 3 of its sub-image; rr8_swz of that row does not depend on s
  const int rr_k = fg * 8 + (fr >> 2);
  const int rr_base = rr_k * 128 + ((fr & 1) << 3);
  const int rr_c = ((fr >> 1) & 1) ^ rr8_swz(rr_k);           // chunk index of columns cbase = 0
  float* stage = reinterpret_cast<float*>(ring + STAGING) + wave * 1024;

  int step = sk ? 0 : next_step(0);
  Item cur = item_of(step);
  int tile = cur.tile;
  if (tile >= ntiles) return;
  if ((VAR & 2) && wr == 1) __builtin_amdgcn_s_setprio(1);
  set_src(tile);
  // prologue: units 0..5 of the first tile (k-tile 0 -> buffer 0, units 0-1 of k-tile 1 -> buffer 1)
  if (!RR || nit > 0) {                                // (RR: a split without k-tiles only writes its zero slab)
#pragma unroll
    for (int u = 0; u < 6; ++u) issue(u & 3, u >> 2, 2 * cur.kb + (u >> 2));
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // units 0-2: everything phase 0 reads
  }
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  bool first = true;
  typedef __attribute__((address_space(1))) unsigned int gu32_t;
  while (true) {
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int step_n = sk ? step + 1 : next_step(step + 1);
    const Item nxt = item_of(step_n);
    const int next_tile = nxt.tile;
    const bool has_next = next_tile < ntiles;
    if (!RR) nit = cur.nit;
    f32x4 acc[8][4];
    if (sk && cur.kind == 2) {
      // TAKE: start from the left neighbour's partial.  Every wave polls for itself (no workgroup barrier: the two wave
      // halves' barrier phase must not move); the 32 loads are younger than the DMA units in flight, so the first
      // counted waits of this item over-wait (in-order retirement) -- safe.
      gu32_t* flag = (gu32_t*)(reinterpret_cast<unsigned int*>(skws) + (sk_gidx - 1));
      if (lane == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != sk_epoch) {
          __builtin_amdgcn_s_sleep(2);
#ifndef NSP_HOST_EMULATION
          if (++spins > (1u << 26)) __builtin_trap();     // seconds: the producer never ran -- fail the launch, do not return a wrong tile
#endif
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      const unsigned char* slot = skws + 4096 + (size_t)(sk_gidx - 1) * 262144 + wave * 32768;
      int tl = lane;                 // (opaque per item: otherwise 16 lane addresses are hoisted in front of the tile loop and spilled)
#ifndef NSP_HOST_EMULATION
      asm volatile("" : "+v"(tl));
#endif
#ifdef NSP_HOST_EMULATION
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const f32x4* rowp = reinterpret_cast<const f32x4*>(slot + i * 4096);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = rowp[j * 64 + tl];
      }
#else
      // asm loads + one named wait: behind compiler-visible loads hipcc put an s_waitcnt vmcnt(0) in front of EVERY item's
      // main loop (the accumulators are a phi of "loaded" and "zero"), i.e. every tile waited for the previous epilogue's stores
      const unsigned toff = (unsigned)tl * 16u;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const unsigned char* rowbase = slot + i * 4096;
        // (s_nop 4 first: the scalar base may have just been restored from a spill lane by v_readlane -- VALU writes SGPR ->
        // VMEM reads it needs 5 wait states, and hipcc cannot see the VMEM instruction inside the asm)
        asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %4, %5 nt\n\tglobal_load_dwordx4 %1, %4, %5 offset:1024 nt\n\t"
                     "global_load_dwordx4 %2, %4, %5 offset:2048 nt\n\tglobal_load_dwordx4 %3, %4, %5 offset:3072 nt"
                     : "=&v"(acc[i][0]), "=&v"(acc[i][1]), "=&v"(acc[i][2]), "=&v"(acc[i][3]) : "v"(toff), "s"(rowbase) : "memory");
      }
#pragma unroll
      for (int i = 0; i < 8; ++i)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(acc[i][0]), "+v"(acc[i][1]), "+v"(acc[i][2]), "+v"(acc[i][3]) :: "memory");
#endif
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (wr == 1) {                                     // the second row half runs one barrier behind the first
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
#pragma unroll 1   // (also keeps the unroller from PEELING iteration 0 -- a second copy of the eight phases whose register
                    // allocation spilled the fragment addresses and reloaded them behind s_waitcnt vmcnt(0))
    for (int it = 0; it < nit; ++it) {
      const bool last = it == nit - 1;
      bf16x8 bfr[2][4], afr[2][2];
      RR8Frag rb[2][4], ra[2][2];      // RR: the asm-loaded halves of the same fragments
#pragma unroll
      for (int ph = 0; ph < 8; ++ph) {
        const int buf = ph >> 2, q = ph & 3;
        const unsigned char* kb = ring + buf * BUF;
        // ---- read segment: fragments of this phase, one load unit, the counted wait
        if (RR) {
          if (q == 0) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
              for (int ni = 0; ni < 4; ++ni)
                rr8_read(rb[s][ni], kb + (4 + wc) * 8192 + s * 4096 + rr_base + ((rr_c ^ (ni * 2)) << 4));
          }
#pragma unroll
          for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
              rr8_read(ra[s][mi], kb + ((q >> 1) * 2 + wr) * 8192 + s * 4096 + rr_base + ((rr_c ^ ((q & 1) * 4 + mi * 2)) << 4));
        } else {
          if (q == 0) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
              for (int ni = 0; ni < 4; ++ni)
                bfr[s][ni] = *reinterpret_cast<const bf16x8*>(kb + b_addr[s] + ni * 2048);
          }
#pragma unroll
          for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
              afr[s][mi] = *reinterpret_cast<const bf16x8*>(kb + a_addr[s] + (q * 32 + mi * 16) * 128);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!(VAR & 32)) {      // (ablation bit 5: no load units at all -- wrong results, timing only)
          // the stream is six units ahead: unit (ph + 6) & 3 of k-tile 2 it + (ph + 6) / 4 -- or, from phase 2 of the
          // last iteration on, of the NEXT tile's k-tiles 0 / 1
          const int j = (ph + 6) & 3, ahead = (ph + 6) >> 2, ibuf = ahead & 1;
          if (!last || ph < 2) {
            issue(j, ibuf, 2 * (cur.kb + it) + ahead);
          } else if (has_next) {
            if (ph == 2) set_src(next_tile);
            issue(j, ibuf, 2 * nxt.kb + ahead - 2);
          }
        }
        if (VAR & (16 | 32)) {  // (ablation bit 4: no counted waits -- wrong results, timing only)
        } else if (ph == 1) {
          if (first || it > 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else if (ph == 5) {
          if (!last || has_next) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (ph == 3) {
          if (!last || has_next) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        } else if (ph == 7) {
          if (!last || has_next) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // ---- multiply segment: quadrant q x this k-tile
        if (RR) {
#ifndef NSP_HOST_EMULATION
          // the asm reads of this phase have landed; every destination is named so that no consumer moves above the wait
          if (q == 0) {
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(rb[0][0].lo), "+v"(rb[0][0].hi), "+v"(rb[0][1].lo), "+v"(rb[0][1].hi), "+v"(rb[0][2].lo), "+v"(rb[0][2].hi),
                           "+v"(rb[0][3].lo), "+v"(rb[0][3].hi), "+v"(rb[1][0].lo), "+v"(rb[1][0].hi), "+v"(rb[1][1].lo), "+v"(rb[1][1].hi));
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(rb[1][2].lo), "+v"(rb[1][2].hi), "+v"(rb[1][3].lo), "+v"(rb[1][3].hi));
          }
          asm volatile("s_waitcnt lgkmcnt(0)"
                       : "+v"(ra[0][0].lo), "+v"(ra[0][0].hi), "+v"(ra[0][1].lo), "+v"(ra[0][1].hi), "+v"(ra[1][0].lo), "+v"(ra[1][0].hi),
                         "+v"(ra[1][1].lo), "+v"(ra[1][1].hi));
          __builtin_amdgcn_sched_barrier(0);
#endif
          if (q == 0) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
              for (int ni = 0; ni < 4; ++ni) bfr[s][ni] = rr8_join(rb[s][ni]);
          }
#pragma unroll
          for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) afr[s][mi] = rr8_join(ra[s][mi]);
        }
        if (!(VAR & 1)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
              acc[q * 2 + mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[s][ni], afr[s][mi], acc[q * 2 + mi][ni], 0, 0, 0);
        if (!(VAR & 1)) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (wr == 0) {                                     // re-align: both halves run their epilogues at the same time
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    // this wave's outstanding DMA (units 3-5 of the next tile) lands before the first store is issued: the counted
    // waits of the next tile then never have a store among the operations they leave in flight
    if (sk && cur.kind == 1) {
      // GIVE: the raw accumulators to this workgroup's slot (16 B per lane and fragment, write-through), then the flag.
      // ONE lane-offset register and a scalar base per row of fragments: 32 per-lane 64-bit addresses spilled the kernel.
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the next item's units have landed: no DMA among the stores
      unsigned char* slot = skws + 4096 + (size_t)sk_gidx * 262144 + wave * 32768;
      int gl = lane;
#ifndef NSP_HOST_EMULATION
      asm volatile("" : "+v"(gl));
#endif
      const unsigned voff = (unsigned)gl * 16u;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        unsigned char* rowbase = slot + i * 4096;
#ifdef NSP_HOST_EMULATION
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(rowbase + j * 1024 + voff) = acc[i][j];
#else
        asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %5 sc1\n\tglobal_store_dwordx4 %0, %2, %5 offset:1024 sc1\n\t"
                     "global_store_dwordx4 %0, %3, %5 offset:2048 sc1\n\tglobal_store_dwordx4 %0, %4, %5 offset:3072 sc1\n\ts_nop 1"
                     :: "v"(voff), "v"(acc[i][0]), "v"(acc[i][1]), "v"(acc[i][2]), "v"(acc[i][3]), "s"(rowbase) : "memory");
#endif
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (tid == 0) __hip_atomic_store((gu32_t*)(reinterpret_cast<unsigned int*>(skws) + sk_gidx), sk_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
    // (the wait itself sits inside gemm_epilogue_fast, behind the requests for the bias and the first side-operand chunks)
    const int mrow = tm * 256 + wr * 128, ncol = tn * 256 + wc * 64;
    // everything the epilogue derives from the lane index is derived HERE, behind an opaque copy: otherwise the
    // compiler hoists its address arithmetic out of the tile loop and the main loop spills (ISA audit, round 4)
    int elane = lane;
#ifndef NSP_HOST_EMULATION
    asm volatile("" : "+v"(elane));
#endif
    if constexpr (!S::kStatic || (VAR & 4)) {
      // ONE call over the wave's eight row blocks (round 6; two calls of four re-loaded the bias and started their
      // side-operand chain behind a vmcnt(0) that also waited for all 32 stores of the first half)
      if constexpr (S::kStatic) {
        gemm_epilogue_fast<8, S, true>(p, acc, stage, mrow, ncol + (elane & 15) * 4, elane, coff);
      } else {   // (the run-time version over eight blocks is not unrolled by hipcc: dynamic accumulator indexing, scratch)
        gemm_epilogue_fast<4, S, true>(p, reinterpret_cast<f32x4(&)[4][4]>(acc[0]), stage, mrow, ncol + (elane & 15) * 4, elane, coff);
        gemm_epilogue_fast<4, S, true>(p, reinterpret_cast<f32x4(&)[4][4]>(acc[4]), stage, mrow + 64, ncol + (elane & 15) * 4, elane, coff);
      }
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      gemm_epilogue_direct<S, (VAR & 8) ? 0 : 2>(p, reinterpret_cast<f32x4(&)[4][4]>(acc[0]), mrow, ncol, elane);
      gemm_epilogue_direct<S, (VAR & 8) ? 0 : 2>(p, reinterpret_cast<f32x4(&)[4][4]>(acc[4]), mrow + 64, ncol, elane);
    }
    }
    if (!has_next) break;
    step = step_n;
    cur = nxt;
    tile = next_tile;
    first = false;
  }
}

// (Round 4, measured and REMOVED -- kernel text kept in profiles/r04s_gemm_kk2w_kernel_source.hip.txt, numbers in
// profiles/r04s_gemm_2w_256x128_two_workgroups_negative.log: the 8-phase kernel's wave tile (128 x 64) on 4 waves with a
// three-stage ring of 24-KB k-tiles, 72 KB, so that TWO workgroups share a CU and one can run its epilogue while the
// other multiplies.  Correct (race screen clean), but 0.85-1.0x the 128 x 128 kernel at every shape of the step, 0.74x
// the 8-phase kernel at K = 2048: two stages ahead of a 0.45-us k-step is 0.9 us of prefetch against a ~2-us DMA round
// trip -- with 2 x 48 KB in flight per CU at 1.5x the 8-phase kernel's bytes per flop it is latency-bound at ~0.8
// PFLOP/s before any epilogue overlap can matter.)

// ---- RC x RC (both operands contiguous along their OUTPUT index, reduction index strided: the
// weight gradients dW = dY^T X) on the same LDS-DMA ring.  A stage holds the k-major images
// [64 k][128 m] and [64 k][128 n] (256-B rows, unpadded: the DMA writes lane-linear); MFMA operands
// are formed with ds_read_b64_tr_b16 (4 k-rows x 16 columns -> 4 k-values per lane).  The 32-B
// column pair a transposed read touches in row k is XOR-swizzled with (k & 3) | ((k >> 3) & 1) << 2
// (on the source address of the DMA and on the read), which spreads the 4 rows of one 16-lane group
// and the two row groups of a 32-lane pass over all 64 banks.  Requires K % 64 == 0.
__device__ __forceinline__ bf16x8 rr_frag(const unsigned char* tile, int cbase, int s, int r, int g) {
  const int a = r >> 2, b = r & 3;
  const int k0 = s * 32 + g * 8 + a;          // rows k0 and k0 + 4 share (k & 3) and bit 3
  const int col = cbase + b * 4;
  const int off = (((col >> 3) ^ rr_swz(k0)) << 4) + ((col & 7) << 1);
  typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(tile + k0 * 256 + off));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(tile + (k0 + 4) * 256 + off));
  bf16x8 o;
  o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3];
  o[4] = hi[0]; o[5] = hi[1]; o[6] = hi[2]; o[7] = hi[3];
  return o;
}

template <int NS>
__global__ __launch_bounds__(NTHREADS) void gemm_bf16_rr_ring_kernel(const nsp_gemm_params p, int tiles_m,
                                                                     int tiles_n, int c_vec) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ring[];  // NS x (A image 16 KB | B image 16 KB)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const TileCoord tc = tile_coord(p, tiles_m * tiles_n);
  const int tile = tc.tile, split = tc.split, z1 = tc.z1, z2 = tc.z2;
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const __bf16* A = reinterpret_cast<const __bf16*>(p.A) + z1 * p.a_b1 + z2 * p.a_b2;
  const __bf16* B = reinterpret_cast<const __bf16*>(p.B) + z1 * p.b_b1 + z2 * p.b_b2;
  const long long coff = z1 * p.c_b1 + z2 * p.c_b2 + (p.c_ss ? (long long)split * p.c_ss : 0);
  const long long lda = p.a_cs, ldb = p.b_ks;   // row pitch of the k-major operands
  int kbeg = 0, kend = p.K;
  if (p.splitk > 1) {
    int nkt_all = p.K / BK;
    int per = (nkt_all + p.splitk - 1) / p.splitk;
    kbeg = split * per * BK;
    kend = min(p.K, (split + 1) * per * BK);
    if (kbeg >= kend) {
      if (!p.c_ss) return;
      kend = kbeg;
    }
  }
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // DMA instruction i of wave w covers k-rows 16w + 4i .. +3 of a tile: lane = (k-row & 3, chunk)
  const int lk = lane >> 4, lp = lane & 15;
  const __bf16* asrc[4];
  const __bf16* bsrc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int krow = wave * 16 + i * 4 + lk;
    const int csrc = lp ^ rr_swz(krow);
    // chunks beyond M / N re-read chunk 0 of the row: they only feed outputs that are never stored
    const int ma = (m0 + csrc * 8 < p.M) ? m0 + csrc * 8 : m0;
    const int nb = (n0 + csrc * 8 < p.N) ? n0 + csrc * 8 : n0;
    asrc[i] = A + (long long)(kbeg + krow) * lda + ma;
    bsrc[i] = B + (long long)(kbeg + krow) * ldb + nb;
  }
  const int fr = lane & 15, fg = lane >> 4;
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;
  const int nkt = (kend - kbeg) / BK;
  auto issue = [&](int kt) {
    unsigned char* sa = ring + (kt % NS) * 32768 + wave * 4096;
    const long long ka = (long long)kt * BK * lda, kb = (long long)kt * BK * ldb;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_global_load_lds((glb_void*)(asrc[i] + ka), (lds_void*)(sa + i * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_void*)(bsrc[i] + kb), (lds_void*)(sa + 16384 + i * 1024), 16, 0, 0);
    }
  };
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nkt) issue(s);
  for (int kt = 0; kt < nkt; ++kt) {
    const int ahead = min(NS - 2, nkt - 1 - kt);
    if (NS >= 4 && ahead >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (NS >= 3 && ahead == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kt + NS - 1 < nkt) issue(kt + NS - 1);
    const unsigned char* smA = ring + (kt % NS) * 32768;
    const unsigned char* smB = smA + 16384;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 af[4], bf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        af[i] = rr_frag(smA, wm * 64 + i * 16, s, fr, fg);
        bf[i] = rr_frag(smB, wn * 64 + i * 16, s, fr, fg);
      }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[ni], af[mi], acc[mi][ni], 0, 0, 0);
    }
  }
  __syncthreads();  // the epilogue reuses the ring as its staging area
  gemm_epilogue<4>(p, acc, ring, m0, n0, wm, wn, lane, wave, coff, c_vec);
}

__global__ void cast_bf16_kernel(const float* __restrict__ x, __bf16* __restrict__ out, long long rows,
                                 int cols, long long ld_in, long long ld_out, int vec_in) {
  const long long chunks_per_row = (cols + 7) >> 3;  // pad only inside the last 8-wide chunk
  const long long total = rows * chunks_per_row;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long long r = i / chunks_per_row;
    const int c = (int)(i % chunks_per_row) * 8;
    float v[8];
    const float* src = x + r * ld_in + c;
    if (vec_in && c + 8 <= cols) {
      float4 a = reinterpret_cast<const float4*>(src)[0], b = reinterpret_cast<const float4*>(src)[1];
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (c + e < cols) ? src[e] : 0.f;
    }
    bf16x8 h;
#pragma unroll
    for (int e = 0; e < 8; ++e) h[e] = (__bf16)v[e];
    *reinterpret_cast<bf16x8*>(out + r * ld_out + c) = h;
  }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// weight gradients the 8-phase kernel does not take (M or N <= 128, short reductions) run on the 2-stage LDS-DMA ring from
// this many output tiles on, below it on the register-staged kernel
constexpr int RR_RING_MIN_TILES = 24;

// 8-phase weight-gradient kernel (gemm_bf16_kk8p_kernel<.., RR = true>): NSP_GEMM_RR8P = 0 switches it off (read on every call)
inline bool rr8p_shape_ok(long long M, long long N, long long K, long long lda, long long ldb, int splitk) {
  const char* e = getenv("NSP_GEMM_RR8P");
  if (e && atoi(e) == 0) return false;
  // 32-bit byte offsets inside ONE reduction split (+ the lane's 64 k-rows)
  const long long per_rows = ((K + 127) / 128 * 2 + splitk - 1) / splitk * 64 + 256;
  return M % 8 == 0 && N % 8 == 0 && M > 128 && N > 128 && K >= 4 * BK && per_rows * lda * 2 < (1ll << 31) &&
         per_rows * ldb * 2 < (1ll << 31);
}

}  // namespace

// called from nsp_gemm (gemm.hip) when both operands are bf16
// Stream-K workspace of the 8-phase kernel: per stream, 4 KB of flags (one per workgroup, compared with the launch's
// epoch -- never reset) + 256 slots of 256 KB for the raw accumulators of a shared tile's head.  Launches on one stream
// are ordered, so one workspace per stream is enough; the memory is allocated once and kept.
static unsigned char* streamk_workspace(hipStream_t st, unsigned* epoch) {
  struct Ws { hipStream_t st; int dev; unsigned char* p; unsigned epoch; };
  static std::mutex mu;
  static std::vector<Ws> all;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lk(mu);
  for (auto& w : all)
    if (w.st == st && w.dev == dev) { *epoch = ++w.epoch; return w.p; }
  unsigned char* ptr = nullptr;
  const size_t bytes = 4096 + 256ull * 262144;
  if (hipMalloc(reinterpret_cast<void**>(&ptr), bytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  if (hipMemsetAsync(ptr, 0, 4096, st) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(ptr); return nullptr; }
  all.push_back(Ws{st, dev, ptr, 1u});
  *epoch = 1u;
  return ptr;
}

int nsp_gemm_bf16_launch(const nsp_gemm_params& p, hipStream_t st) {
  const bool a_kc = p.a_cs == 1, b_kc = p.b_ks == 1;
  {
    static int dbg = -1;       // NSP_GEMM_DEBUG=1: one line per launch on stderr (which shapes reach which kernel family)
    if (dbg < 0) { const char* e = getenv("NSP_GEMM_DEBUG"); dbg = e ? atoi(e) : 0; }
    if (dbg) {
      // algorithmic HBM bytes of the launch: both operands once, every output image once, every side operand once
      const long long nb = (long long)p.batch1 * p.batch2, mn = (long long)p.M * p.N;
      const long long csz_ = p.c_dtype == NSP_DT_BF16 ? 2 : 4;
      long long bytes = nb * (2ll * p.M * p.K + 2ll * p.N * p.K);
      if (p.epi_mode == NSP_EPI_RNNT_LSE) bytes += (long long)p.M * (p.N / 64) * 8 + 8ll * p.M;
      else bytes += nb * mn * csz_ * (p.c_ss ? p.splitk : 1);
      if (p.pre_out) bytes += mn * (p.pre_dtype == NSP_DT_BF16 ? 2 : 4);
      if (p.dact_src) bytes += mn * (p.dact_dtype == NSP_DT_BF16 ? 2 : 4);
      if (p.res) bytes += mn * 4;
      fprintf(stderr, "[nsp_gemm_bf16] M %d N %d K %d a_kc %d b_kc %d splitk %d batch %dx%d epi %d c16 %d pre %d dact %d res %d drop %g algbytes %lld\n", p.M, p.N, p.K,
              (int)a_kc, (int)b_kc, p.splitk, p.batch1, p.batch2, p.epi_mode, (int)(p.c_dtype == NSP_DT_BF16), (int)(p.pre_out != nullptr),
              p.dact_src ? p.dact : 0, (int)(p.res != nullptr), (double)p.dropout_p, bytes);
    }
  }
  if (!a_kc && p.a_rs != 1) return NSP_EINVAL;
  if (!b_kc && p.b_ns != 1) return NSP_EINVAL;
  const long long lda = a_kc ? p.a_rs : p.a_cs, ldb = b_kc ? p.b_ns : p.b_ks;
  if (lda % 8 || ldb % 8 || p.a_b1 % 8 || p.a_b2 % 8 || p.b_b1 % 8 || p.b_b2 % 8) return NSP_EINVAL;
  if (!aligned16(p.A) || !aligned16(p.B)) return NSP_EINVAL;
  // every 16-B chunk along the contiguous index must be readable: either the extent is a
  // multiple of 8 or the leading dimension is padded to one (pad values must be finite; the
  // other operand's zero fill of the reduction tail / the epilogue's bounds make them inert)
  const int a_ext = a_kc ? p.K : p.M, b_ext = b_kc ? p.K : p.N;
  if ((a_ext % 8) && lda < ((a_ext + 7) / 8) * 8) return NSP_EINVAL;
  if ((b_ext % 8) && ldb < ((b_ext + 7) / 8) * 8) return NSP_EINVAL;
  int tiles_m = nsp_cdiv(p.M, BM);
  const int tiles_n = nsp_cdiv(p.N, BN);
  const int csz = p.c_dtype == NSP_DT_BF16 ? 2 : 4;
  int c_vec = (reinterpret_cast<uintptr_t>(p.C) % (4 * csz) == 0) && p.ldc % 4 == 0 && p.c_b1 % 4 == 0 &&
              p.c_b2 % 4 == 0;
  if (p.pre_out && reinterpret_cast<uintptr_t>(p.pre_out) % 16) c_vec = 0;
  if (p.dact_src && reinterpret_cast<uintptr_t>(p.dact_src) % 16) c_vec = 0;
  if (p.res && !aligned16(p.res)) c_vec = 0;
  if (p.bias && !aligned16(p.bias)) c_vec = 0;
  // split-K of one problem: flat grid, whole splits per XCD (tile_coord)
  const int flat = (p.batch1 * p.batch2 == 1 && p.splitk > 1) ? p.splitk : 1;
  dim3 grid(tiles_m * tiles_n * flat, 1, flat > 1 ? 1 : p.batch1 * p.batch2 * p.splitk), block(NTHREADS);
  const bool fast_epi = c_vec && p.N % 4 == 0 && !(p.splitk > 1 && p.c_ss == 0) && !(p.res && p.dact_src);
  if (a_kc && b_kc && p.K % BK == 0 && p.K >= BK) {
    // how many workgroups would share a CU decides how the load latency gets hidden
    constexpr long long ring2_max = 640;
    static bool ring_attr = false;
    if (!ring_attr) {
      ring_attr = true;
      (void)hipFuncSetAttribute((const void*)gemm_bf16_kk_ring_kernel<4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768);
      (void)hipFuncSetAttribute((const void*)gemm_bf16_kk_ring_kernel<2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 32768);
      (void)hipFuncSetAttribute((const void*)gemm_bf16_kk_ring_kernel<4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 24576);
    }
    const long long wgs = (long long)grid.x * grid.z;
    const int nkt = p.K / BK / p.splitk;
    const long long t256 = (long long)nsp_cdiv(p.M, 256) * nsp_cdiv(p.N, 256);
    // 256 x 256 phase-interleaved persistent kernel (round 4): the default for one problem with the fast epilogue
    // and enough 256-tiles (NSP_GEMM_8P=0 switches it off, NSP_GEMM_8P_MIN_TILES moves the threshold; both read on
    // every call so that tests can flip them inside one process)
    {
      const char* e8 = getenv("NSP_GEMM_8P");
      const char* e8m = getenv("NSP_GEMM_8P_MIN_TILES");
      const int on8 = e8 ? atoi(e8) : 1;
      const long long min8 = e8m ? atoll(e8m) : 96;
      // Where it wins (interleaved A/B at the step's shapes, profiles/r04c_gemm_8p_bench.log): every long reduction
      // (K >= 1024: x1.12-1.24 at 102400 rows, x1.15-1.22 at 25600), and K = 512 only with a light epilogue on a grid
      // that fills its last round of 256 workgroups (stacked QKV x1.07, the FFN data gradient x1.10 at 102400 rows).
      // With K = 512 the exposed epilogue of ONE workgroup per CU costs what the faster main loop gains; the FFN
      // first linear (two bf16 images, Swish, dropout: VALU-bound epilogue) and the thin N <= 1024 outputs stay on the
      // 128 x 128 kernel, whose four workgroups per CU overlap their epilogues.  NSP_GEMM_8P = 2 forces it everywhere.
      const long long rounds8 = (t256 + 255) / 256;
      const bool fills = t256 * 10 >= rounds8 * 256 * 9;
      const bool want8 = on8 >= 2 || p.K >= 1024 || (fills && !p.pre_out && p.N >= 1536);
      if (on8 && want8 && p.epi_mode == NSP_EPI_NONE && fast_epi && p.batch1 * p.batch2 == 1 && p.splitk == 1 && p.K % 128 == 0 &&
          t256 >= min8 && 256ll * p.a_rs + p.K < (1ll << 31) && (long long)p.N * p.b_ns + p.K < (1ll << 31)) {
        const int tm256 = nsp_cdiv(p.M, 256), tn256 = nsp_cdiv(p.N, 256);
        int g8 = (int)(t256 >= 256 ? 256 : (t256 + 7) / 8 * 8);
        const char* e8g = getenv("NSP_GEMM_8P_GRID");   // tests: fewer workgroups, i.e. several tiles per workgroup on small problems
        if (e8g && atoi(e8g) >= 8 && atoi(e8g) < g8) g8 = atoi(e8g) / 8 * 8;
        const char* e8v = getenv("NSP_GEMM_8P_VAR");
        const int var8 = e8v ? atoi(e8v) : 4;   // default: the staged epilogue (the direct one measured 0.4-0.9x, see gemm_epilogue_direct)
        // stream-K (see the kernel): a full grid whose last round is less than 3/4 full, tile count a multiple of 8 (whole
        // tiles per XCD), at least one tile of iterations per workgroup and narrow output (the n-fastest tile list).
        // ON by default where it should pay.  Measured (profiles/r06_gemm_epilogue.log): alone, back to back, the step's N = 512
        // products gain 3-8 % (FFN input gradient 217 -> 206 us at 102 400 rows, 109 -> 101 at 51 200); inside the step, with the
        // XCD's tiles in column-major order, five alternating pairs in one call: 93.96 -> 93.59 ms (every pair), kernel time
        // under rocprofv3 671.6 -> 667.2 ms per 6 steps.  (With the n-fastest order the two tiles sharing an A panel were
        // consecutive items of one workgroup and the step LOST 0.6 ms.)  NSP_GEMM_8P_STREAMK = 0 switches it off, 2 forces it
        // wherever it is legal (tests); read on every call.
        unsigned char* skws = nullptr;
        unsigned sk_epoch = 0;
        {
          const char* esk = getenv("NSP_GEMM_8P_STREAMK");
          const int sk_on = esk ? atoi(esk) : 1;
          const long long rem8 = t256 % g8;
          const bool legal = t256 % 8 == 0 && tn256 < 16 && t256 / 8 >= g8 / 8 &&   // (>= one tile of iterations per workgroup)
                             (t256 / 8) % tn256 == 0;                                  // (whole row panels per XCD)
          const bool pays = g8 == 256 && rem8 != 0 && rem8 * 4 < 3 * g8;
          const bool twin = p.c_dtype == NSP_DT_F32 && !p.pre_out && !p.dact_src && p.act == NSP_ACT_NONE;   // epi_spec_streamk_twin
          if (sk_on && legal && twin && (pays || sk_on >= 2)) {
            skws = streamk_workspace(st, &sk_epoch);
          }
        }
        auto launch8 = [&](auto spec, auto var) {
          using S8 = decltype(spec);
          constexpr int V8 = decltype(var)::value;
          static bool attr = false;     // (one flag per instantiation of this lambda's call operator)
          if (!attr) {
            (void)hipFuncSetAttribute((const void*)gemm_bf16_kk8p_kernel<S8, V8>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
            attr = true;
          }
          constexpr bool has_twin = S8::kStatic && V8 == 4 && epi_spec_streamk_twin<S8>();
          if constexpr (has_twin) {
            if (skws) {
              static bool attr_sk = false;
              if (!attr_sk) {
                (void)hipFuncSetAttribute((const void*)gemm_bf16_kk8p_kernel<S8, V8, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
                attr_sk = true;
              }
              hipLaunchKernelGGL((gemm_bf16_kk8p_kernel<S8, V8, false, true>), dim3(g8), dim3(512), 163840, st, p, tm256, tn256, c_vec & 255, skws, sk_epoch);
              return;
            }
          }
          hipLaunchKernelGGL((gemm_bf16_kk8p_kernel<S8, V8>), dim3(g8), dim3(512), 163840, st, p, tm256, tn256, c_vec & 255, nullptr, 0u);
        };
        // the direct epilogue's own conditions (32-bit byte offsets into C; column-sum slabs only with an act' source)
        const bool direct_ok = (long long)p.M * p.ldc < (1ll << 29) && !(p.epi_f3 && !p.dact_src) && !(p.bias && p.dact_src);
        bool done = false;
#if NSP_GEMM_8P_AB
        if (direct_ok && var8 == 0) done = epi_spec_visit(p, [&](auto spec) { launch8(spec, std::integral_constant<int, 0>{}); });
        else if (direct_ok && var8 == 8) done = epi_spec_visit(p, [&](auto spec) { launch8(spec, std::integral_constant<int, 8>{}); });
        else if ((var8 == 20 || var8 == 36) && !p.res && !p.pre_out && !p.dact_src && p.act == NSP_ACT_NONE && p.dropout_p == 0.f) {
          // main-loop ablations (timing only, results are wrong), plain epilogues only
          const bool c16 = p.c_dtype == NSP_DT_BF16;
          if (var8 == 20 && c16) launch8(EpiSpec<0, 0, true, false, false, false>{}, std::integral_constant<int, 20>{});
          else if (var8 == 20) launch8(EpiSpec<0, 0, false, false, false, false>{}, std::integral_constant<int, 20>{});
          else if (c16) launch8(EpiSpec<0, 0, true, false, false, false>{}, std::integral_constant<int, 36>{});
          else launch8(EpiSpec<0, 0, false, false, false, false>{}, std::integral_constant<int, 36>{});
          done = true;
        }
#else
        (void)direct_ok; (void)var8;
#endif
        if (!done) done = epi_spec_visit(p, [&](auto spec) { launch8(spec, std::integral_constant<int, 4>{}); });
        if (!done) launch8(EpiRuntime{}, std::integral_constant<int, 0>{});
        NSP_LAUNCH_CHECK();
        return NSP_OK;
      }
    }
    if (wgs < 192 && p.M > 64 && nkt >= 4 && p.epi_mode == NSP_EPI_NONE) {
      tiles_m = nsp_cdiv(p.M, 64);
      grid.x = tiles_m * tiles_n * flat;
      hipLaunchKernelGGL((gemm_bf16_kk_ring_kernel<4, 2>), grid, block, 4 * 24576, st, p, tiles_m, tiles_n, c_vec);
    } else if (wgs <= 288 && nkt >= 4)
      hipLaunchKernelGGL((gemm_bf16_kk_ring_kernel<4, 4>), grid, block, 4 * 32768, st, p, tiles_m, tiles_n, c_vec);
    else if (wgs <= ring2_max && nkt >= 2)
      hipLaunchKernelGGL((gemm_bf16_kk_ring_kernel<2, 4>), grid, block, 2 * 32768, st, p, tiles_m, tiles_n, c_vec);
    else if (p.epi_mode == NSP_EPI_RNNT_LSE)
      hipLaunchKernelGGL(gemm_bf16_kk_glds_kernel<1>, grid, block, 0, st, p, tiles_m, tiles_n, c_vec);
    else if (p.epi_mode == NSP_EPI_RNNT_DLOGITS)
      hipLaunchKernelGGL(gemm_bf16_kk_glds_kernel<2>, grid, block, 0, st, p, tiles_m, tiles_n, c_vec);
    else if (fast_epi)
      hipLaunchKernelGGL(gemm_bf16_kk_glds_kernel<0>, grid, block, 0, st, p, tiles_m, tiles_n, c_vec);
    else   // odd widths / unaligned outputs / atomic split-K on a large grid: the generic epilogue lives in the ring kernels
      hipLaunchKernelGGL((gemm_bf16_kk_ring_kernel<2, 4>), grid, block, 2 * 32768, st, p, tiles_m, tiles_n, c_vec);
  }
  else if (!a_kc && !b_kc && p.batch1 * p.batch2 == 1 && fast_epi && (p.splitk == 1 || p.c_ss) && p.epi_mode == NSP_EPI_NONE &&
           rr8p_shape_ok(p.M, p.N, p.K, lda, ldb, p.splitk)) {
    const int tm256 = nsp_cdiv(p.M, 256), tn256 = nsp_cdiv(p.N, 256);
    const dim3 g(tm256 * tn256 * p.splitk);
    const bool plain = !p.bias && !p.res && !p.pre_out && !p.dact_src && p.act == NSP_ACT_NONE && p.dropout_p == 0.f &&
                       p.c_dtype == NSP_DT_F32 && epi_spec_w32_ok(p);
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute((const void*)gemm_bf16_kk8p_kernel<EpiSpec<0, 0, false, false, false, false>, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
      (void)hipFuncSetAttribute((const void*)gemm_bf16_kk8p_kernel<EpiRuntime, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
      attr = true;
    }
    if (plain) hipLaunchKernelGGL((gemm_bf16_kk8p_kernel<EpiSpec<0, 0, false, false, false, false>, 4, true>), g, dim3(512), 163840, st, p, tm256, tn256, c_vec);
    else hipLaunchKernelGGL((gemm_bf16_kk8p_kernel<EpiRuntime, 0, true>), g, dim3(512), 163840, st, p, tm256, tn256, c_vec);
  }
  else if (!a_kc && !b_kc && p.K % BK == 0 && p.K >= 2 * BK && p.M % 8 == 0 && p.N % 8 == 0 &&
           tiles_m * tiles_n >= RR_RING_MIN_TILES) {
    // the weight gradients the 8-phase kernel does not take: LDS-DMA ring with swizzled transposed reads, 2 stages, two
    // workgroups per CU (the 1-stage / 3-stage variants and the 256 x 256 one-barrier kernel of rounds 1-3 measured equal
    // or slower at every shape of the step and were removed in round 5)
    static bool rr_attr = false;
    if (!rr_attr) {
      (void)hipFuncSetAttribute((const void*)gemm_bf16_rr_ring_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 32768);
      rr_attr = true;
    }
    hipLaunchKernelGGL((gemm_bf16_rr_ring_kernel<2>), grid, block, 2 * 32768, st, p, tiles_m, tiles_n, c_vec);
  }
  else if (a_kc && b_kc) hipLaunchKernelGGL((gemm_bf16_kernel<true, true>), grid, block, 0, st, p, tiles_m, tiles_n, c_vec);
  else if (a_kc && !b_kc) hipLaunchKernelGGL((gemm_bf16_kernel<true, false>), grid, block, 0, st, p, tiles_m, tiles_n, c_vec);
  else if (!a_kc && b_kc) hipLaunchKernelGGL((gemm_bf16_kernel<false, true>), grid, block, 0, st, p, tiles_m, tiles_n, c_vec);
  else hipLaunchKernelGGL((gemm_bf16_kernel<false, false>), grid, block, 0, st, p, tiles_m, tiles_n, c_vec);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

// reduction splits of a weight gradient dW[N, K] = dY[rows, N]^T X[rows, K] in bf16 mode, or 0 when the caller's
// own rule applies (the 128 x 128 kernels).  256 x 256 tiles x splits ~ 256 workgroups = one per CU.
extern "C" int nsp_wgrad_splitk(long long N, long long K, long long rows) {
  {
    // 8-phase kernel: 256 x 256 tiles x splits ~ one workgroup per CU, an EVEN number of k-tiles per split (>= 4)
    const long long tiles = (long long)nsp_cdiv((int)N, 256) * nsp_cdiv((int)K, 256);
    const long long nkt_pad = ((rows + 127) >> 7) << 1;
    long long sk = 256 / tiles;
    if (sk > nkt_pad / 12) sk = nkt_pad / 12;   // >= 12 k-tiles per split: the pipeline's fill and the fp32 slab (1 MB per
    if (sk < 1) sk = 1;                         // 256 x 256 tile and split, written and reduced again) want amortising
    const long long per = (((nkt_pad + sk - 1) / sk) + 1) & ~1ll;
    const int plan = (int)((nkt_pad + per - 1) / per);      // no empty splits
    if (rr8p_shape_ok(N, K, rows, N, K, plan)) return plan;
  }
  return 0;
}

extern "C" int nsp_cast_bf16(const float* x, void* out, long long rows, int cols, long long ld_in,
                             long long ld_out, void* stream) {
  if (rows <= 0 || cols <= 0) return NSP_OK;
  if (ld_out % 8 || ld_out < cols || (reinterpret_cast<uintptr_t>(out) & 15)) return NSP_EINVAL;
  const int vec_in = (reinterpret_cast<uintptr_t>(x) % 16 == 0) && ld_in % 4 == 0;
  long long n = rows * ((cols + 7) / 8);
  long long g = (n + 255) / 256;
  if (g > 256 * 16) g = 256 * 16;
  hipLaunchKernelGGL(cast_bf16_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, x,
                     reinterpret_cast<__bf16*>(out), rows, cols, ld_in, ld_out, vec_in);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

#if NSP_GEMM_TRACE
extern "C" int nsp_gemm_trace_read(unsigned long long* out, int n_wg) {
  (void)hipDeviceSynchronize();
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(nsp_gemm_trace_buf), sizeof(unsigned long long) * 4 * (size_t)n_wg) == hipSuccess ? 0 : -1;
}
#endif
