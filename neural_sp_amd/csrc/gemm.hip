// gemm.hip -- batched strided GEMM with fused epilogue on CDNA4 MFMA.
//
// One kernel template serves every dense contraction of the Speech2Text hot
// path (reference call sites listed in include/nsp_hip.h).  Operands live in
// HBM as fp32; a 128x128 output tile is computed by 4 waves (2x2), each wave
// owning a 64x64 sub-tile = 4x4 MFMA 16x16 fragments.
//
//   MODE 0 (NSP_COMPUTE_BF16): operands are rounded to bf16 (RNE) on the way
//       into LDS and fed to v_mfma_f32_16x16x32_bf16, fp32 accumulate.  BK=32.
//   MODE 1 (NSP_COMPUTE_F32):  operands stay fp32, v_mfma_f32_16x16x4_f32
//       (bit-exact fp32 fma chain).  BK=16.  This is the parity mode.
//
// LDS image of an operand tile: [128 rows][BK] with the reduction index
// contiguous and an 80-byte row pitch (64 B payload + 16 B pad; keeps the
// 16-B ds_read_b128 fragment reads aligned and spreads rows over banks).
// Both A (m,k) and B (k,n) may have the reduction index either contiguous
// (KC) or strided with the row/col index contiguous (RC); the RC loader
// transposes 4x4 blocks in registers so that global reads stay 16 B/lane
// coalesced.  Global->register loads of tile t+1 are issued before the MFMA
// work on tile t.
//
// The MFMA is issued as D^T = B_frag x A_frag so that each lane ends up with 4
// consecutive n for one m: the epilogue then uses float4 loads/stores.
#include "common.h"
#include <string.h>

namespace {

constexpr int BM = 128, BN = 128, NTHREADS = 256;
constexpr int PITCH = 80;  // bytes per LDS row

template <int MODE> struct ModeCfg;
template <> struct ModeCfg<0> { static constexpr int BK = 32; static constexpr int NL = 4; };
template <> struct ModeCfg<1> { static constexpr int BK = 16; static constexpr int NL = 2; };

// ---- tile loader: 128 rows x BK reduction elements ------------------------
// element (r, k) lives at base + r*rs + k*ks ; KC: ks == 1 ; RC: rs == 1
template <int MODE, bool KC>
struct TileLoader {
  static constexpr int BK = ModeCfg<MODE>::BK;
  static constexpr int NL = ModeCfg<MODE>::NL;
  float4 r[NL];

  __device__ __forceinline__ void load(const float* __restrict__ base, long long rs, long long ks,
                                       int row0, int R, int k0, int Kend, bool vec_ok) {
    const int tid = threadIdx.x;
    if (KC) {
      constexpr int CPR = BK / 4;  // float4 chunks per row
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        int idx = tid + i * NTHREADS;
        int row = idx / CPR, c = idx % CPR;
        int gr = row0 + row, gk = k0 + c * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gr < R && gk < Kend) {
          const float* ptr = base + (long long)gr * rs + (long long)gk * ks;
          if (vec_ok && gk + 3 < Kend) {
            v = *reinterpret_cast<const float4*>(ptr);
          } else {
            v.x = ptr[0];
            if (gk + 1 < Kend) v.y = ptr[ks];
            if (gk + 2 < Kend) v.z = ptr[2 * ks];
            if (gk + 3 < Kend) v.w = ptr[3 * ks];
          }
        }
        r[i] = v;
      }
    } else {
      // thread owns rows rq*4..rq*4+3 and reduction indices kq*NL..kq*NL+NL-1
      const int rq = (tid & 7) + 8 * (tid >> 6);
      const int kq = (tid >> 3) & 7;
      const int gr = row0 + rq * 4;
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        int gk = k0 + kq * NL + i;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gk < Kend && gr < R) {
          const float* ptr = base + (long long)gk * ks + (long long)gr * rs;
          if (vec_ok && gr + 3 < R) {
            v = *reinterpret_cast<const float4*>(ptr);
          } else {
            v.x = ptr[0];
            if (gr + 1 < R) v.y = ptr[rs];
            if (gr + 2 < R) v.z = ptr[2 * rs];
            if (gr + 3 < R) v.w = ptr[3 * rs];
          }
        }
        r[i] = v;
      }
    }
  }

  __device__ __forceinline__ void store(unsigned char* lds) const {
    const int tid = threadIdx.x;
    if (KC) {
      constexpr int CPR = BK / 4;
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        int idx = tid + i * NTHREADS;
        int row = idx / CPR, c = idx % CPR;
        if (MODE == 0) {
          bf16x4 h;
          h[0] = (__bf16)r[i].x; h[1] = (__bf16)r[i].y; h[2] = (__bf16)r[i].z; h[3] = (__bf16)r[i].w;
          *reinterpret_cast<bf16x4*>(lds + row * PITCH + c * 8) = h;
        } else {
          *reinterpret_cast<float4*>(lds + row * PITCH + c * 16) = r[i];
        }
      }
    } else {
      const int rq = (tid & 7) + 8 * (tid >> 6);
      const int kq = (tid >> 3) & 7;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int row = rq * 4 + j;
        if (MODE == 0) {
          bf16x4 h;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float e = j == 0 ? r[i].x : j == 1 ? r[i].y : j == 2 ? r[i].z : r[i].w;
            h[i] = (__bf16)e;
          }
          *reinterpret_cast<bf16x4*>(lds + row * PITCH + kq * 8) = h;
        } else {
          float2 f;
          f.x = j == 0 ? r[0].x : j == 1 ? r[0].y : j == 2 ? r[0].z : r[0].w;
          f.y = j == 0 ? r[1].x : j == 1 ? r[1].y : j == 2 ? r[1].z : r[1].w;
          *reinterpret_cast<float2*>(lds + row * PITCH + kq * 8) = f;
        }
      }
    }
  }
};

// bijective XCD-aware remap (blocks b, b+8, ... share an XCD/L2): make the
// tiles that share an A panel land on one XCD.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int NX = 8;
  if (nwg < NX * 2) return bid;
  int q = nwg / NX, rem = nwg % NX;
  int xcd = bid % NX, slot = bid / NX;
  int base = xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q;
  return base + slot;
}

template <int MODE, bool A_KC, bool B_KC>
__global__ __launch_bounds__(NTHREADS) void gemm_kernel(const nsp_gemm_params p, int tiles_m,
                                                        int tiles_n, int a_vec, int b_vec,
                                                        int c_vec) {
  constexpr int BK = ModeCfg<MODE>::BK;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BM * PITCH];
  unsigned char* smA = smem;
  unsigned char* smB = smem + BM * PITCH;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  const int tile = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  // batch / split-k decomposition of blockIdx.z
  int z = blockIdx.z;
  const int split = z % p.splitk;
  z /= p.splitk;
  const int z2 = z % p.batch2, z1 = z / p.batch2;
  const float* A = reinterpret_cast<const float*>(p.A) + z1 * p.a_b1 + z2 * p.a_b2;
  const float* B = reinterpret_cast<const float*>(p.B) + z1 * p.b_b1 + z2 * p.b_b2;
  float* const Cp = reinterpret_cast<float*>(p.C);
  float* const preo = reinterpret_cast<float*>(p.pre_out);
  const float* const dsrc = reinterpret_cast<const float*>(p.dact_src);
  const long long coff = z1 * p.c_b1 + z2 * p.c_b2 + (p.c_ss ? (long long)split * p.c_ss : 0);

  // reduction range of this split (multiples of BK)
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

  TileLoader<MODE, A_KC> la;
  TileLoader<MODE, B_KC> lb;
  // A: row = m ; B: row = n
  la.load(A, p.a_rs, p.a_cs, m0, p.M, kbeg, kend, a_vec);
  lb.load(B, p.b_ns, p.b_ks, n0, p.N, kbeg, kend, b_vec);

  const int frow = lane & 15, fk = lane >> 4;

  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    la.store(smA);
    lb.store(smB);
    __syncthreads();
    if (k0 + BK < kend) {
      la.load(A, p.a_rs, p.a_cs, m0, p.M, k0 + BK, kend, a_vec);
      lb.load(B, p.b_ns, p.b_ks, n0, p.N, k0 + BK, kend, b_vec);
    }
    if (MODE == 0) {
      bf16x8 af[4], bf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        af[i] = *reinterpret_cast<const bf16x8*>(smA + (wm * 64 + i * 16 + frow) * PITCH + fk * 16);
        bf[i] = *reinterpret_cast<const bf16x8*>(smB + (wn * 64 + i * 16 + frow) * PITCH + fk * 16);
      }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[ni], af[mi], acc[mi][ni], 0, 0, 0);
    } else {
#pragma unroll
      for (int s = 0; s < BK / 4; ++s) {
        float af[4], bf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          af[i] = *reinterpret_cast<const float*>(smA + (wm * 64 + i * 16 + frow) * PITCH + (s * 4 + fk) * 4);
          bf[i] = *reinterpret_cast<const float*>(smB + (wn * 64 + i * 16 + frow) * PITCH + (s * 4 + fk) * 4);
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[ni], af[mi], acc[mi][ni], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  // ---- epilogue: lane holds C[m][n..n+3], m = m0+wm*64+mi*16+(lane&15),
  //      n = n0+wn*64+ni*16+(lane>>4)*4
  const bool atomic = p.splitk > 1 && p.c_ss == 0;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const int m = m0 + wm * 64 + mi * 16 + frow;
    if (m >= p.M) continue;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int n = n0 + wn * 64 + ni * 16 + fk * 4;
      if (n >= p.N) continue;
      const long long off = coff + (long long)m * p.ldc + n;
      float v[4] = {acc[mi][ni][0], acc[mi][ni][1], acc[mi][ni][2], acc[mi][ni][3]};
      const int nv = min(4, p.N - n);
      if (atomic) {
        _Pragma("unroll") for (int e = 0; e < 4; ++e) if (e < nv) unsafeAtomicAdd(Cp + off + e, v[e] * p.alpha);
        continue;
      }
      const bool vec = c_vec && nv == 4;
      if (p.bias) {
        if (vec) {
          float4 b4 = *reinterpret_cast<const float4*>(p.bias + n);
          v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
        } else {
          _Pragma("unroll") for (int e = 0; e < 4; ++e) if (e < nv) v[e] += p.bias[n + e];
        }
      }
      if (p.pre_out) {
        if (vec) *reinterpret_cast<float4*>(preo + off) = make_float4(v[0], v[1], v[2], v[3]);
        else { _Pragma("unroll") for (int e = 0; e < 4; ++e) if (e < nv) preo[off + e] = v[e]; }
      }
      if (p.act != NSP_ACT_NONE) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = nsp_act(v[e], p.act);
      }
      if (p.dact_src) {
        float d[4] = {0.f, 0.f, 0.f, 0.f};
        if (vec) {
          float4 d4 = *reinterpret_cast<const float4*>(dsrc + off);
          d[0] = d4.x; d[1] = d4.y; d[2] = d4.z; d[3] = d4.w;
        } else {
          _Pragma("unroll") for (int e = 0; e < 4; ++e) if (e < nv) d[e] = dsrc[off + e];
        }
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
        if (vec) {
          float4 r4 = *reinterpret_cast<const float4*>(p.res + off);
          v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
        } else {
          _Pragma("unroll") for (int e = 0; e < 4; ++e) if (e < nv) v[e] += p.res[off + e];
        }
      }
      if (vec) *reinterpret_cast<float4*>(Cp + off) = make_float4(v[0], v[1], v[2], v[3]);
      else { _Pragma("unroll") for (int e = 0; e < 4; ++e) if (e < nv) Cp[off + e] = v[e]; }
    }
  }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <int MODE>
int launch_mode(const nsp_gemm_params& p, hipStream_t st) {
  const int tiles_m = nsp_cdiv(p.M, BM), tiles_n = nsp_cdiv(p.N, BN);
  const bool a_kc = (p.a_cs == 1) || (p.a_rs != 1);  // default to KC when neither is unit
  const bool b_kc = (p.b_ks == 1) || (p.b_ns != 1);
  // vector (float4) global access is legal when the contiguous index has unit
  // stride and base / leading dim / batch strides keep 16-B alignment.
  auto vec_ok = [&](const void* base, long long unit, long long ld, long long s1, long long s2) {
    return unit == 1 && aligned16(base) && ld % 4 == 0 && s1 % 4 == 0 && s2 % 4 == 0;
  };
  const int a_vec = a_kc ? vec_ok(p.A, p.a_cs, p.a_rs, p.a_b1, p.a_b2) : vec_ok(p.A, p.a_rs, p.a_cs, p.a_b1, p.a_b2);
  const int b_vec = b_kc ? vec_ok(p.B, p.b_ks, p.b_ns, p.b_b1, p.b_b2) : vec_ok(p.B, p.b_ns, p.b_ks, p.b_b1, p.b_b2);
  int c_vec = aligned16(p.C) && p.ldc % 4 == 0 && p.c_b1 % 4 == 0 && p.c_b2 % 4 == 0;
  if (p.bias && !aligned16(p.bias)) c_vec = 0;
  if (p.pre_out && !aligned16(p.pre_out)) c_vec = 0;
  if (p.dact_src && !aligned16(p.dact_src)) c_vec = 0;
  if (p.res && !aligned16(p.res)) c_vec = 0;
  dim3 grid(tiles_m * tiles_n, 1, p.batch1 * p.batch2 * p.splitk), block(NTHREADS);
  if (a_kc && b_kc)
    hipLaunchKernelGGL((gemm_kernel<MODE, true, true>), grid, block, 0, st, p, tiles_m, tiles_n, a_vec, b_vec, c_vec);
  else if (a_kc && !b_kc)
    hipLaunchKernelGGL((gemm_kernel<MODE, true, false>), grid, block, 0, st, p, tiles_m, tiles_n, a_vec, b_vec, c_vec);
  else if (!a_kc && b_kc)
    hipLaunchKernelGGL((gemm_kernel<MODE, false, true>), grid, block, 0, st, p, tiles_m, tiles_n, a_vec, b_vec, c_vec);
  else
    hipLaunchKernelGGL((gemm_kernel<MODE, false, false>), grid, block, 0, st, p, tiles_m, tiles_n, a_vec, b_vec, c_vec);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

}  // namespace

int nsp_gemm_bf16_launch(const nsp_gemm_params& p, hipStream_t st);  // gemm_bf16.hip

extern "C" int nsp_gemm(const nsp_gemm_params* pp, void* stream) {
  if (!pp) return NSP_EINVAL;
  nsp_gemm_params p = *pp;
  if (p.M <= 0 || p.N <= 0 || p.K < 0 || !p.A || !p.B || !p.C) return NSP_EINVAL;
  if (p.batch1 < 1) p.batch1 = 1;
  if (p.batch2 < 1) p.batch2 = 1;
  if (p.splitk < 1) p.splitk = 1;
  if ((long long)p.batch1 * p.batch2 * p.splitk > 65535) return NSP_EINVAL;
  if (p.splitk > 1 && (p.bias || p.pre_out || p.dact_src || p.res || p.act != NSP_ACT_NONE || p.dropout_p > 0.f))
    return NSP_EINVAL;
  // a non-unit/non-unit operand is handled by the KC loader's scalar path
  hipStream_t st = (hipStream_t)stream;
  if (p.a_dtype != p.b_dtype) return NSP_EINVAL;
  if (p.a_dtype == NSP_DT_BF16) {
    if (p.mode != NSP_COMPUTE_BF16) return NSP_EINVAL;
    if (p.splitk > 1 && p.c_dtype != NSP_DT_F32) return NSP_EINVAL;
    return nsp_gemm_bf16_launch(p, st);
  }
  if (p.c_dtype != NSP_DT_F32 || p.pre_dtype != NSP_DT_F32 || p.dact_dtype != NSP_DT_F32) return NSP_EINVAL;
  if (p.mode == NSP_COMPUTE_BF16) return launch_mode<0>(p, st);
  if (p.mode == NSP_COMPUTE_F32) return launch_mode<1>(p, st);
  return NSP_EINVAL;
}

extern "C" int nsp_version(void) { return 100; }

// Positional-argument twin of nsp_gemm for FFI layers where filling a struct field by field is
// the dominant host cost (ctypes): same semantics, same validation.
extern "C" int nsp_gemm_flat(int M, int N, int K, const void* A, long long a_rs, long long a_cs,
                             const void* B, long long b_ks, long long b_ns, void* C, long long ldc,
                             int batch1, int batch2, long long a_b1, long long a_b2, long long b_b1,
                             long long b_b2, long long c_b1, long long c_b2, const float* bias, int act,
                             void* pre_out, const void* dact_src, int dact, const float* res,
                             float alpha, int splitk, int mode, float dropout_p,
                             unsigned long long seed, unsigned long long offset, int a_dtype,
                             int b_dtype, int c_dtype, int pre_dtype, int dact_dtype, long long c_ss,
                             float* colsum_slabs, void* stream) {
  nsp_gemm_params p;
  p.M = M; p.N = N; p.K = K;
  p.A = A; p.a_rs = a_rs; p.a_cs = a_cs;
  p.B = B; p.b_ks = b_ks; p.b_ns = b_ns;
  p.C = C; p.ldc = ldc;
  p.batch1 = batch1; p.batch2 = batch2;
  p.a_b1 = a_b1; p.a_b2 = a_b2; p.b_b1 = b_b1; p.b_b2 = b_b2; p.c_b1 = c_b1; p.c_b2 = c_b2;
  p.bias = bias; p.act = act; p.pre_out = pre_out; p.dact_src = dact_src; p.dact = dact;
  p.res = res; p.alpha = alpha; p.splitk = splitk; p.mode = mode; p.dropout_p = dropout_p;
  p.seed = seed; p.offset = offset;
  p.a_dtype = a_dtype; p.b_dtype = b_dtype; p.c_dtype = c_dtype; p.pre_dtype = pre_dtype;
  p.dact_dtype = dact_dtype; p.c_ss = c_ss;
  p.epi_mode = NSP_EPI_NONE; p.epi_ncols = 0; p.epi_blank = 0; p.epi_lab = nullptr;
  p.epi_f0 = p.epi_f1 = p.epi_f2 = nullptr; p.epi_f3 = colsum_slabs; p.epi_scale_dev = nullptr; p.epi_scale = 1.f;
  if (colsum_slabs && (a_dtype != NSP_DT_BF16 || splitk > 1)) return NSP_EUNSUPPORTED;
  return nsp_gemm(&p, stream);
}

// The same call with its 38 arguments packed into 38 little-endian 8-byte slots (int64 / pointer / double) in the order of
// nsp_gemm_flat's parameter list.  Round 6: ctypes converts arguments one by one (~0.2 us each); a training step at 16
// utterances per GPU makes ~360 GEMM calls and is bound by the host, so the 39-argument call was ~3 ms of every such step --
// struct.pack + a two-argument call is ~2 us.
extern "C" int nsp_gemm_packed(const void* packed, void* stream) {
  if (!packed) return NSP_EINVAL;
  long long q[38];
  memcpy(q, packed, sizeof(q));
  auto ptr = [&](int i) { return reinterpret_cast<void*>(static_cast<uintptr_t>(q[i])); };
  auto dbl = [&](int i) { double v; memcpy(&v, &q[i], sizeof(v)); return v; };
  return nsp_gemm_flat((int)q[0], (int)q[1], (int)q[2], ptr(3), q[4], q[5], ptr(6), q[7], q[8], ptr(9), q[10],
                       (int)q[11], (int)q[12], q[13], q[14], q[15], q[16], q[17], q[18], reinterpret_cast<const float*>(ptr(19)),
                       (int)q[20], ptr(21), ptr(22), (int)q[23], reinterpret_cast<const float*>(ptr(24)), (float)dbl(25),
                       (int)q[26], (int)q[27], (float)dbl(28), (unsigned long long)q[29], (unsigned long long)q[30],
                       (int)q[31], (int)q[32], (int)q[33], (int)q[34], (int)q[35], q[36], reinterpret_cast<float*>(ptr(37)), stream);
}
