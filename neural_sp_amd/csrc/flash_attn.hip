// flash_attn.hip -- fused relative-position self-attention for d_k = 64 (bf16 MFMA).
//
// Replaces the score / softmax / context pipeline of
// relative_multihead_attention.py:179-215 (and multihead_attention.py:124-153 when there is
// no position term) without ever writing the [B,H,T,T] score or probability tensors:
//
//   forward : per (64-query tile, head, utterance) workgroup, 4 waves x 16 queries; K/V tiles
//             of 64 keys staged in LDS; S^T = K Q^T on MFMA with the QUERY index on lane&15, so
//             the row statistics (online max / sum) need only two cross-lane shuffles and the
//             probabilities are already laid out as the MFMA operand of P V (no LDS round trip);
//             V is consumed through ds_read_b64_tr_b16 transpose reads.
//             e(i,j) = (q_i.k_j + QP[i, min(|i-j|, clamp)]) * scale with the reference's mask
//             semantics (masked = -FLT_MAX, finite) and counter-based dropout.
//             Saves the (integer-valued, log2-domain) row max and 1/sum (LSE[0], LSE[1], each [B,H,T]) for backward.
//   backward: per (64-key tile, head, utterance) workgroup looping over query tiles: recomputes
//             P from LSE, dP^T = V dO^T, dS = P (dP - D) scale; dV += P_drop^T dO and
//             dK += dS^T Q over the workgroup's keys (accumulated in registers, the P / dS
//             tiles transposed through LDS with transpose reads); a second, query-parallel
//             kernel recomputes dS and produces dQ = dS K and the relative-table gradient dQP
//             with no global atomics.
// The 11-entry (clamp_len = 10) position table per query lives in LDS.
// LSE[0] holds the CEILING of the row maximum in the LOG2 domain (logit * log2 e), LSE[1] the reciprocal row sum.
//
// Round 3 (measured at B = 64, H = 8, T = 800, key lengths 600..800; profiles/r03i_flash_bench.log):
// forward 327 -> 228 us (368 TFLOP/s; with dropout 368 -> 263 us), backward 1182 -> 1113 us.
//   * key tiles beyond an utterance's length are skipped (their probabilities are exactly 0), the three tiles
//     around the diagonal take a straight-line position lookup instead of the general masked path;
//   * ONE bf16 probability operand instead of the hi + lo pair: the running maximum is integer-valued, so every
//     rescale is a power of two and backward reproduces forward's P bit for bit (see the forward kernel);
//   * K / V (dK/dV kernel: Q / dO) tiles by LDS-DMA into unpadded swizzled images: no staging registers, no
//     ds_write pass (ablation: prefetch + restage was 32 % of the forward, profiles/r03h_flash_fwd_ablation.log);
//   * cross-lane max / sum with v_permlane16/32_swap instead of ds_bpermute, the row sum combined once at the end.
// What remains is VALU issue: ~13 (forward) to ~25 (backward with dropout) vector instructions per score against
// 256 MFMA flops -- the ablation's "everything removed but fragment reads, max, rescale and conversions" build
// still takes 42 % of the forward.
#include "common.h"
#include <type_traits>

namespace {

constexpr int DK = 64;
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

// 16-B LDS read the compiler does not see (see the dK/dV kernel); T = f32x4 or u32x4
template <class T>
__device__ __forceinline__ void fa_lds_read4(T& dst, const void* p) {
#ifdef NSP_HOST_EMULATION
  dst = *reinterpret_cast<const T*>(p);
#else
  typedef __attribute__((address_space(3))) unsigned char lds_uchar;
  const unsigned a = (unsigned)(uintptr_t)((lds_uchar*)const_cast<void*>(p));
  asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(a) : "memory");
#endif
}

__device__ __forceinline__ bool fa_visible(const nsp_attn_mask_params& p, int klen, int i, int j) {
  bool ok = j < klen;
  if (p.causal) ok = ok && (j <= i + p.lookahead);
  if (p.chunk_nc > 0) {
    int c0 = (i / p.chunk_nc) * p.chunk_nc;
    int lo = c0 - p.chunk_nl;
    if (lo < 0) lo = 0;
    ok = ok && (j >= lo) && (j < c0 + p.chunk_nc);
  }
  return ok;
}

typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
__device__ __forceinline__ void fa_lds_read2(u32x2& dst, const void* p) {
#ifdef NSP_HOST_EMULATION
  dst = *reinterpret_cast<const u32x2*>(p);
#else
  typedef __attribute__((address_space(3))) unsigned char lds_uchar;
  const unsigned a = (unsigned)(uintptr_t)((lds_uchar*)const_cast<void*>(p));
  asm volatile("ds_read_b64 %0, %1" : "=v"(dst) : "v"(a) : "memory");
#endif
}

// ---- per-element work of a (query tile, key tile) pair.  With d_k = 64 the softmax arithmetic,
// not the MFMA, bounds these kernels (16 logits per lane per tile against 16 MFMAs per wave), so
// everything that is uniform over a tile is decided once per tile:
//   plain : no causal / chunk mask and the key tile lies inside min(klen, T) -> no predicates;
//   far   : every pair is >= clamp apart -> the relative term is the per-query constant QP[i][clamp]
//           (all but the 1-3 tiles around the diagonal).
// Logits are kept in the log2 domain (scale * log2(e) folded into one FMA; v_exp_f32 is exp2), and
// the saved row maximum LSE[0] is in that domain too.  Dropout: one 32-bit mix per PAIR of adjacent
// keys on top of a per-row hash, 16 bits per decision.
constexpr float LOG2E = 1.4426950408889634f;

struct FaTile { bool plain, far; };
__device__ __forceinline__ FaTile fa_tile(const nsp_attn_mask_params& p, bool has_qp, int q0, int k0, int T, int klen,
                                          int nq = 64) {
  // queries q0 .. q0 + nq - 1 against keys k0 .. k0 + 63
  FaTile t;
  t.plain = !p.causal && p.chunk_nc == 0 && (k0 + 63 < min(klen, T));
  t.far = has_qp && p.clamp > 0 && (k0 - (q0 + nq - 1) >= p.clamp || q0 - (k0 + 63) >= p.clamp);
  return t;
}

// ev = log2-domain logits of the lane's pairs (query qi, key k0 + 16 kf + 4 g + e); returns the
// visibility bits (bit 4 kf + e) -- all ones on plain tiles.  qrow = this query's row of the
// relative table pre-multiplied by scale*log2(e), or nullptr.
__device__ __forceinline__ unsigned fa_logits(const f32x4 (&s_acc)[4], float (&ev)[4][4], const nsp_attn_mask_params& p,
                                              const float* qrow, float sl2, int qi, int k0, int g, int klen, FaTile tl) {
  if (tl.plain && (qrow == nullptr || tl.far)) {
    const float add = qrow ? qrow[p.clamp] : 0.f;
#pragma unroll
    for (int kf = 0; kf < 4; ++kf)
#pragma unroll
      for (int e = 0; e < 4; ++e) ev[kf][e] = fmaf(s_acc[kf][e], sl2, add);
    return 0xFFFFu;
  }
  unsigned vis = 0u;
#pragma unroll
  for (int kf = 0; kf < 4; ++kf)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int key = k0 + kf * 16 + 4 * g + e;
      float v = s_acc[kf][e] * sl2;
      if (qrow) {
        int rel = qi > key ? qi - key : key - qi;
        if (p.clamp > 0 && rel > p.clamp) rel = p.clamp;
        v += qrow[rel];
      }
      const bool ok = tl.plain || fa_visible(p, klen, qi, key);
      if (!ok) v = -FLT_MAX;
      vis |= (ok ? 1u : 0u) << (kf * 4 + e);
      ev[kf][e] = v;
    }
  return vis;
}

// NEAR tile: no mask predicate (plain) but inside the clamp band, so the relative term differs from key to key:
// straight-line |i - j| -> min(., clamp) -> one LDS word per score, no branches.  (Round 2 sent these tiles through
// fa_logits with its per-score visibility / causal / chunk predicates: with 64-query and 64-key tiles the three
// tiles around the diagonal are "near", and together with the key tiles beyond an utterance's length -- now skipped
// altogether -- ~45 % of all tiles ran that branchy path at several times the cost of a uniform tile.)
__device__ __forceinline__ void fa_logits_near(const f32x4 (&s_acc)[4], float (&ev)[4][4], const float* qrow, float sl2,
                                               int qi, int k0, int g, int clamp) {
#pragma unroll
  for (int kf = 0; kf < 4; ++kf)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int dlt = qi - (k0 + kf * 16 + 4 * g + e);
      const int rel = min(dlt < 0 ? -dlt : dlt, clamp);
      ev[kf][e] = fmaf(s_acc[kf][e], sl2, qrow[rel]);
    }
}

// max / sum over the four lanes that share lane & 15 (xor 16, xor 32): v_permlane16_swap / v_permlane32_swap exchange
// 16- resp. 32-lane halves between two registers in one VALU instruction each -- the ds_bpermute behind
// __shfl_xor is an LDS-crossbar round trip (~100+ cycles of latency on the softmax's critical path, x4 per tile).
__device__ __forceinline__ float fa_xmax4(float v) {
  const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float fa_xsum4(float v) {
  const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// number of key tiles that can contribute: keys >= klen are masked with -FLT_MAX in the reference
// (relative_multihead_attention.py:203-204), and with at least one visible key in the row exp(-FLT_MAX - max) is
// exactly 0 -- whole tiles beyond the utterance's length add nothing to any sum.  Only without causal / chunk
// masks (with them a row can be fully masked, where the reference's softmax is uniform over ALL keys).
__device__ __forceinline__ int fa_key_tiles(const nsp_attn_mask_params& p, int T, int klen) {
  const int nkt = (T + 63) / 64;
  if (p.causal || p.chunk_nc > 0 || klen < 1) return nkt;
  return min(nkt, (min(klen, T) + 63) / 64);
}

// Dropout decisions.  Round 6: the FORWARD draws them and stores them, one 16-bit word per lane and (16-query block, 64-key
// tile) pair -- bit 15 - (4 kf + e) = keep (query 16 qblk + r, key k0 + 16 kf + 4 g + e), lane = r + 16 g, i.e. exactly the
// forward's / dQ kernel's score registers; keepbits is u16 [B][H][key tile][16-query block][64 lanes] (128 B per pair, 89 MB
// at B = 128, H = 8, T = 800) -- and both backward kernels READ them (2 VALU per score: sign-extended bit field + AND) instead
// of re-drawing them (~14 VALU per score in each of the two kernels: the hash was more than half of their vector
// instructions).  The draw itself is unchanged: one 32-bit value per PAIR of adjacent keys (16 bits per decision) from
// full-rate integer ops only -- a 24-bit multiply-add of the pair index folded into the per-row hash, one xorshift round and a
// second 24-bit multiply-add -- so the decisions (and therefore every result) equal those of the rounds before bit for bit.
// bits = 2 bits + c in ONE instruction: the compare's lane mask is the carry-in of v_addc_co_u32 (written as
// `bits + bits + (c ? 1 : 0)` hipcc built the word with a v_cndmask and 16-bit shift / or per decision: + 2 VALU per score)
__device__ __forceinline__ void fa_shift_in(unsigned& bits, bool c) {
#ifdef NSP_HOST_EMULATION
  bits = bits + bits + (c ? 1u : 0u);
#else
  const unsigned long long m = __builtin_amdgcn_ballot_w64(c);
  asm("v_addc_co_u32 %0, vcc, %0, %0, %1" : "+v"(bits) : "s"(m) : "vcc");
#endif
}
// kp = inv_keep or 0 per score (what the forward multiplies with), return = the 16 decisions as a word
__device__ __forceinline__ unsigned fa_keep_bits(float (&kp)[4][4], unsigned rowhash, int k0, int g, unsigned thr16, float inv_keep) {
  unsigned bits = 0u;
#pragma unroll
  for (int kf = 0; kf < 4; ++kf)
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      const unsigned pair = (unsigned)(k0 + kf * 16 + 4 * g + 2 * pr) >> 1;
      unsigned y = rowhash + __umul24(pair, 0x9E3779u) + (pair << 11);
      y ^= y << 13; y ^= y >> 17; y ^= y << 5;
      y = __umul24(y >> 8, 0x85EBCBu) ^ y;
      const bool c0 = (y & 0xFFFFu) >= thr16, c1 = (y >> 16) >= thr16;
      kp[kf][2 * pr] = c0 ? inv_keep : 0.f;
      kp[kf][2 * pr + 1] = c1 ? inv_keep : 0.f;
      fa_shift_in(bits, c0);
      fa_shift_in(bits, c1);
    }
  return bits;
}
// v if the decision with index idx (= 4 kf + e) of `bits` says keep, else +0
__device__ __forceinline__ float fa_keep_apply(float v, unsigned bits, int idx) {
  const int m = ((int)(bits << (16 + idx))) >> 31;
  return __uint_as_float(__float_as_uint(v) & (unsigned)m);
}
// the word of (key tile kt, 16-query block qblk) of head (b, h) starts at this u16 index (+ lane)
__device__ __forceinline__ long long fa_keep_index(const nsp_attn_mask_params& p, int b, int h, int kt, int qblk) {
  const int nkt = (p.Tq + 63) >> 6;
  return ((((long long)b * p.H + h) * nkt + kt) * (4 * nkt) + qblk) * 64;
}
__device__ __forceinline__ unsigned fa_rowhash(const nsp_attn_mask_params& p, int b, int h, int T, int qi) {
  return nsp_hash_u32(p.seed, p.offset + (unsigned long long)(((long long)b * p.H + h) * T + qi));
}

// ---- K / V tiles staged by LDS-DMA (global_load_lds_dwordx4) in the forward and dq kernels.  The ablation of the
// register-staged forward (profiles/r03h_flash_fwd_ablation.log: T = 800, B = 64) put 32 % of the kernel into the
// next tile's prefetch + restage (4 predicated 16-B loads, 4 ds_write_b128 and their address arithmetic per
// thread and key tile, 16 staging VGPRs) and 24 % behind the barrier that follows it.  The DMA writes a wave's 64
// x 16 B lane-linear, so rows are unpadded 128 B ([64 keys][64 d_k] bf16 = 8 KB per tile) and the bank spreading
// moves into an XOR of the 16-B chunk index on the SOURCE address, mirrored by the reads (CDNA guide rule 21):
//   chunk ^= row & 7 (the GEMM tiles' swizzle): conflict-free both for the ds_read_b128 row fragments (16 rows at one
//   chunk) and for the ds_read_b64_tr_b16 transposed fragments (a 32-lane pass = 8 rows x 32 B -> 8 distinct 32-B
//   slots of the 256-B bank row) -- tools/lds_bank_sim.py: 4 and 2 LDS cycles, the ideal -- so ONE image serves
//   tiles that are read both ways (K in the dq kernel, Q / dO in the dK/dV kernel).
// Rows beyond T re-read row T - 1 (finite values): their scores are masked as tile padding and their
// probabilities are 0, so no zero fill is needed.
constexpr int KD = 128;   // DMA tile row pitch (bytes)
typedef __attribute__((address_space(3))) void fa_lds_void;
typedef const __attribute__((address_space(1))) void fa_glb_void;
// Addresses (round 6): a workgroup-uniform 64-bit base (the utterance's first row: src + brow0 * ld, formed once) plus a
// 32-bit byte offset per lane -- one 24-bit multiply-add per piece.  (As `src + (brow0 + row) * ld` per lane hipcc issued a
// 64-bit multiply -- two v_mul_lo_u32 and a v_mad_u64_u32 -- per piece and tile: 12 of the ~25 slow-rate integer
// instructions of every iteration of the three kernels.)  An utterance's rows span < 2^31 bytes: T * ld * 2.
__device__ __forceinline__ void tile_dma(unsigned char* tile, const __bf16* __restrict__ src, long long ld, long long brow0,
                                         int row0, int T, int wave, int lane) {
  const unsigned char* base = reinterpret_cast<const unsigned char*>(src + brow0 * ld);     // uniform
  const unsigned ldb = (unsigned)ld * 2u;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ii = wave * 2 + i;                    // 1-KB piece: rows 8 ii .. 8 ii + 7
    const int row = ii * 8 + (lane >> 3), c = lane & 7;
    const int sc = c ^ (row & 7);
    const unsigned off = __umul24((unsigned)min(row0 + row, T - 1), ldb) + (unsigned)(sc * 16);
    __builtin_amdgcn_global_load_lds((fa_glb_void*)(base + off), (fa_lds_void*)(tile + ii * 1024), 16, 0, 0);
  }
}
__device__ __forceinline__ bf16x8 frag_kc_dma(const unsigned char* tile, int rbase, int s, int r, int g) {
  const int row = rbase + r;
  return *reinterpret_cast<const bf16x8*>(tile + row * KD + (((s * 4 + g) ^ (row & 7)) << 4));
}
__device__ __forceinline__ bf16x8 frag_tr_dma(const unsigned char* tile, int cbase, int k0, int k1, int r) {
  const int a = r >> 2, b = r & 3;
  const int ch = (cbase >> 3) + (b >> 1), sub = (b & 1) * 8;
  const int r0 = k0 + a, r1 = k1 + a;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(tile + r0 * KD + ((ch ^ (r0 & 7)) << 4) + sub));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(tile + r1 * KD + ((ch ^ (r1 & 7)) << 4) + sub));
  bf16x8 o;
  o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3];
  o[4] = hi[0]; o[5] = hi[1]; o[6] = hi[2]; o[7] = hi[3];
  return o;
}

// The same transposed fragment as two INLINE-ASM reads (round 4).  __builtin_amdgcn_ds_read_tr16_b64 is modelled as
// reading memory an LDS-DMA in flight may write, so hipcc puts an s_waitcnt vmcnt(0) in front of the first one: in the
// forward and dQ kernels that drained the NEXT tile's K / V DMA in the middle of the current tile (before P V resp.
// dS K) instead of at the tile's end.  The compiler does not count an asm load: the destinations are named in the
// s_waitcnt statement that follows the batch (CDNA guide 5.7).
struct FaTr { bf16x4 lo, hi; };
__device__ __forceinline__ void fa_tr_issue(FaTr& f, const unsigned char* tile, int cbase, int k0, int k1, int r) {
  const int a = r >> 2, b = r & 3;
  const int ch = (cbase >> 3) + (b >> 1), sub = (b & 1) * 8;
  const int r0 = k0 + a, r1 = k1 + a;
  const unsigned char* p0 = tile + r0 * KD + ((ch ^ (r0 & 7)) << 4) + sub;
  const unsigned char* p1 = tile + r1 * KD + ((ch ^ (r1 & 7)) << 4) + sub;
#ifdef NSP_HOST_EMULATION
  f.lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)p0);
  f.hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)p1);
#else
  typedef __attribute__((address_space(3))) unsigned char lds_uchar;
  const unsigned a0 = (unsigned)(uintptr_t)((lds_uchar*)const_cast<unsigned char*>(p0));
  const unsigned a1 = (unsigned)(uintptr_t)((lds_uchar*)const_cast<unsigned char*>(p1));
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.lo) : "v"(a0) : "memory");
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.hi) : "v"(a1) : "memory");
#endif
}
__device__ __forceinline__ bf16x8 fa_tr_join(const FaTr& f) {
  return __builtin_shufflevector(f.lo, f.hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
// (all eight fragments of a 64 x 64 tile: issue, then wait for the first / second four)
__device__ __forceinline__ void fa_tr_wait(FaTr (&t)[8], int half) {
#ifndef NSP_HOST_EMULATION
  if (half == 0)
    asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(t[0].lo), "+v"(t[0].hi), "+v"(t[1].lo), "+v"(t[1].hi), "+v"(t[2].lo), "+v"(t[2].hi), "+v"(t[3].lo), "+v"(t[3].hi));
  else
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t[4].lo), "+v"(t[4].hi), "+v"(t[5].lo), "+v"(t[5].hi), "+v"(t[6].lo), "+v"(t[6].hi), "+v"(t[7].lo), "+v"(t[7].hi));
#else
  (void)t; (void)half;
#endif
}

// forward: one workgroup per (128-query tile, head, utterance); 4 waves x 32 queries (two 16-query
// fragments share every K / V fragment read); K / V tiles double-buffered in LDS with the next tile's
// global loads in flight during the current tile's arithmetic -> ONE barrier per key tile.
// grid.x = 8-way interleave of heads and query tiles (id % H = head): with H = 8 every head lives on one
// XCD, so the q-tiles of one (utterance, head) re-read its K / V from that XCD's L2.
template <int NQ>   // 16-query fragments per wave: workgroup tile = 64 * NQ queries
__global__ __launch_bounds__(256, 4) void flash_fwd_kernel(const __bf16* __restrict__ qkv, int d,
                                                        const float* __restrict__ QP,
                                                        __bf16* __restrict__ O, float* __restrict__ O32,
                                                        float* __restrict__ LSE,
                                                        unsigned short* __restrict__ keepbits,
                                                        const nsp_attn_mask_params p) {
  __shared__ __attribute__((aligned(16))) unsigned char KV[4 * 64 * KD + 64 * NQ * 17 * 4];   // ONE LDS object (a second one makes
  unsigned char (*Ks)[64 * KD] = reinterpret_cast<unsigned char (*)[64 * KD]>(KV);             //  hipcc wait vmcnt(0) per LDS read
  unsigned char (*Vs)[64 * KD] = reinterpret_cast<unsigned char (*)[64 * KD]>(KV + 2 * 64 * KD); //  while a DMA is in flight)
  float (*QPs)[17] = reinterpret_cast<float (*)[17]>(KV + 4 * 64 * KD);
  const int T = p.Tq;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int h = blockIdx.x % p.H, q0 = (blockIdx.x / p.H) * (64 * NQ), b = blockIdx.y;
  const long long ld3 = 3LL * d;
  const long long brow0 = (long long)b * T;
  const int klen = p.klens ? p.klens[b] : T;
  const float sl2 = p.scale * LOG2E;
  const __bf16* kbase = qkv + d + h * DK;
  const __bf16* vbase = qkv + 2 * d + h * DK;
  // NOTE the order: Q / QP loads are issued BEFORE tile 0's DMA.  vmcnt retires in issue order, so the wait in
  // front of the first barrier then covers them too and the k-loop is entered with nothing pending.
  int qi[NQ];
  bf16x8 Qf[NQ][2];
#pragma unroll
  for (int f = 0; f < NQ; ++f) {
    qi[f] = q0 + wave * (16 * NQ) + f * 16 + r;
    const __bf16* qp_ = qkv + (brow0 + min(qi[f], T - 1)) * ld3 + h * DK;
    Qf[f][0] = *reinterpret_cast<const bf16x8*>(qp_ + g * 8);
    Qf[f][1] = *reinterpret_cast<const bf16x8*>(qp_ + 32 + g * 8);
  }
  if (QP) {
    for (int idx = threadIdx.x; idx < 64 * NQ * p.r_pitch; idx += 256) {
      const int ql = idx / p.r_pitch, rr = idx % p.r_pitch;
      const int q = min(q0 + ql, T - 1);
      QPs[ql][rr] = QP[((brow0 + q) * p.H + h) * p.r_pitch + rr] * sl2;
    }
  }
  const bool drop = p.dropout_p > 0.f;
  const unsigned thr16 = (unsigned)(p.dropout_p * 65536.f);
  const float inv_keep = drop ? nsp_rcp(1.f - p.dropout_p) : 1.f;
  unsigned rowhash[NQ];
#pragma unroll
  for (int f = 0; f < NQ; ++f) rowhash[f] = drop ? fa_rowhash(p, b, h, T, qi[f]) : 0u;
  f32x4 o_acc[NQ][4];
#pragma unroll
  for (int f = 0; f < NQ; ++f)
#pragma unroll
    for (int i = 0; i < 4; ++i) o_acc[f][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run[NQ], l_run[NQ];
#pragma unroll
  for (int f = 0; f < NQ; ++f) { m_run[f] = -INFINITY; l_run[f] = 0.f; }
  tile_dma(Ks[0], kbase, ld3, brow0, 0, T, wave, lane);
  tile_dma(Vs[0], vbase, ld3, brow0, 0, T, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int nkt = fa_key_tiles(p, T, klen);
  for (int kt = 0; kt < nkt; ++kt) {
    const unsigned char* Kc = Ks[kt & 1];
    const unsigned char* Vc = Vs[kt & 1];
    if (kt + 1 < nkt) {     // the other buffer was last read in iteration kt - 1, behind that iteration's barrier
      tile_dma(Ks[(kt + 1) & 1], kbase, ld3, brow0, (kt + 1) * 64, T, wave, lane);
      tile_dma(Vs[(kt + 1) & 1], vbase, ld3, brow0, (kt + 1) * 64, T, wave, lane);
    }
    f32x4 s_acc[NQ][4];
#pragma unroll
    for (int kf = 0; kf < 4; ++kf) {
#pragma unroll
      for (int f = 0; f < NQ; ++f) s_acc[f][kf] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 kfrag = frag_kc_dma(Kc, kf * 16, s, r, g);
#pragma unroll
        for (int f = 0; f < NQ; ++f)
          s_acc[f][kf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfrag, Qf[f][s], s_acc[f][kf], 0, 0, 0);
      }
    }
    bf16x8 Pf[NQ][2];
#pragma unroll
    for (int f = 0; f < NQ; ++f) {
      // lane: query qi[f], keys kt*64 + kf*16 + 4g + e
      const int q0f = q0 + wave * (16 * NQ) + f * 16;
      const FaTile tl = fa_tile(p, QP != nullptr, q0f, kt * 64, T, klen, 16);
      const float* qrow = QP ? QPs[wave * (16 * NQ) + f * 16 + r] : nullptr;
      // uniform tile (no mask predicate, relative term = the per-query constant QP[i][clamp]): the logit is
      // fma(s, sl2, addc), so max / exp2 work on the raw MFMA output and addc - m folds into ONE fma per score
      const bool uni = tl.plain && (qrow == nullptr || tl.far);
      const float addc = (uni && qrow) ? qrow[p.clamp] : 0.f;
      float ev[4][4];
      float mx = -INFINITY;
      if (uni) {
#pragma unroll
        for (int kf = 0; kf < 4; ++kf)
          mx = fmaxf(mx, fmaxf(fmaxf(s_acc[f][kf][0], s_acc[f][kf][1]), fmaxf(s_acc[f][kf][2], s_acc[f][kf][3])));
        mx = fmaf(mx, sl2, addc);           // sl2 > 0: max commutes with the affine map
      } else if (tl.plain) {
        fa_logits_near(s_acc[f], ev, qrow, sl2, qi[f], kt * 64, g, p.clamp);
#pragma unroll
        for (int kf = 0; kf < 4; ++kf)
#pragma unroll
          for (int e = 0; e < 4; ++e) mx = fmaxf(mx, ev[kf][e]);
      } else {
        fa_logits(s_acc[f], ev, p, qrow, sl2, qi[f], kt * 64, g, klen, tl);
        if (!tl.plain && kt * 64 + 63 >= T) {
#pragma unroll
          for (int kf = 0; kf < 4; ++kf)
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (kt * 64 + kf * 16 + 4 * g + e >= T) ev[kf][e] = -INFINITY;   // tile padding: not in the softmax
        }
#pragma unroll
        for (int kf = 0; kf < 4; ++kf)
#pragma unroll
          for (int e = 0; e < 4; ++e) mx = fmaxf(mx, ev[kf][e]);
      }
      // The running maximum is kept INTEGER-valued (log2 domain: ceil of the true maximum), so every rescale
      // factor alpha = 2^(m_old - m_new) is a power of two.  Why: backward's D_i = dO_i . O_i must equal
      // sum_j Pd_ij dP_ij of the probabilities backward RECOMPUTES far better than bf16 precision -- dS =
      // P (dP - D) sums to zero over keys only then, and any residue multiplies the component common to all
      // keys / queries (large once biases are non-zero), which the softmax's shift invariance removes from the
      // true gradient (measured in round 2: w_query / w_key gradients of the upper Conformer-L blocks at cosine
      // 0.45 with a single bf16 P against fp32-recomputed P).  Round 2 fixed it with a bf16 hi + lo pair (two
      // P V MFMAs per fragment, ~4 extra VALU per score: 18 % of this kernel, profiles/r03h).  With power-of-
      // two rescales the value that enters P V, bf16(exp2(t - m_tile) keep) * 2^(m_tile - m_final), equals
      // bf16(exp2(t - m_final) keep) BIT FOR BIT (rounding commutes with a power-of-two scale), which is what
      // backward forms from the saved m_final: O is then exactly the sum backward pairs with dP, with ONE bf16
      // probability operand.  (exp2's argument differs by one fp32 rounding of t - m between the two:
      // ~1e-6 relative, a bf16 flip once in ~3000 elements -- below fp32 accumulation noise in D.)
      mx = fa_xmax4(mx);
      const float m_new = fmaxf(m_run[f], ceilf(mx));
      const float alpha = __builtin_amdgcn_exp2f(m_run[f] - m_new);   // m_run = -inf on the first tile -> 0
      const float c0 = addc - m_new;
      float rs = 0.f;
      // compiled once per (uniform tile?, dropout?) and chosen once per fragment: with the two wave-uniform flags tested
      // per score hipcc evaluated BOTH exponentials of `uni ? exp2(fma) : exp2(ev - m)` and selected (ISA audit, round 4)
      auto probs = [&](auto uni_, auto drop_) {
        constexpr bool UNI = decltype(uni_)::value, DROP = decltype(drop_)::value;
        float kp[4][4];
        if constexpr (DROP) {
          const unsigned kb = fa_keep_bits(kp, rowhash[f], kt * 64, g, thr16, inv_keep);
          keepbits[fa_keep_index(p, b, h, kt, (q0 >> 4) + wave * NQ + f) + lane] = (unsigned short)kb;   // for backward
        }
#pragma unroll
        for (int kf = 0; kf < 4; ++kf)
#pragma unroll
          for (int e2 = 0; e2 < 4; e2 += 2) {
            f32x2 pr;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              const int e = e2 + q;
              float v;
              if constexpr (UNI) v = __builtin_amdgcn_exp2f(fmaf(s_acc[f][kf][e], sl2, c0));
              else v = __builtin_amdgcn_exp2f(ev[kf][e] - m_new);
              rs += v;
              if constexpr (DROP) v *= kp[kf][e];
              pr[q] = v;
            }
            const bf16x2 hi = __builtin_convertvector(pr, bf16x2);
            Pf[f][kf >> 1][(kf & 1) * 4 + e2] = hi[0];
            Pf[f][kf >> 1][(kf & 1) * 4 + e2 + 1] = hi[1];
          }
      };
      if (uni) { if (drop) probs(std::true_type{}, std::true_type{}); else probs(std::true_type{}, std::false_type{}); }
      else { if (drop) probs(std::false_type{}, std::true_type{}); else probs(std::false_type{}, std::false_type{}); }
      // l_run is this LANE's partial row sum (its 16 keys per tile); alpha is common to the four lanes of a
      // query, so the partials are combined once, after the last tile
      l_run[f] = l_run[f] * alpha + rs;
      m_run[f] = m_new;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        o_acc[f][i][0] *= alpha; o_acc[f][i][1] *= alpha; o_acc[f][i][2] *= alpha; o_acc[f][i][3] *= alpha;
      }
    }
    // O^T[dd][query] += V^T P^T : X = V^T fragment (rows dd), Y = P fragment (rows queries)
    {
      FaTr vt[8];                                     // [ddf * 2 + s]
#pragma unroll
      for (int i = 0; i < 8; ++i) fa_tr_issue(vt[i], Vc, (i >> 1) * 16, 32 * (i & 1) + 4 * g, 32 * (i & 1) + 16 + 4 * g, r);
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        fa_tr_wait(vt, hf);
#pragma unroll
        for (int i = hf * 4; i < hf * 4 + 4; ++i) {
          const bf16x8 vT = fa_tr_join(vt[i]);
#pragma unroll
          for (int f = 0; f < NQ; ++f)
            o_acc[f][i >> 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vT, Pf[f][i & 1], o_acc[f][i >> 1], 0, 0, 0);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of tile kt + 1 have landed ...
    __syncthreads();                                     // ... everybody's have; tile kt's buffers are free
  }
#pragma unroll
  for (int f = 0; f < NQ; ++f) {
    const float l_row = fa_xsum4(l_run[f]);
    if (qi[f] < T) {
      const float inv = nsp_rcp(l_row);
      __bf16* op = O + (brow0 + qi[f]) * d + h * DK;
      float* op32 = O32 ? O32 + (brow0 + qi[f]) * d + h * DK : nullptr;
#pragma unroll
      for (int ddf = 0; ddf < 4; ++ddf) {
        const float4 o = make_float4(o_acc[f][ddf][0] * inv, o_acc[f][ddf][1] * inv, o_acc[f][ddf][2] * inv,
                                     o_acc[f][ddf][3] * inv);
        bf16x4 o4;
        o4[0] = (__bf16)o.x; o4[1] = (__bf16)o.y; o4[2] = (__bf16)o.z; o4[3] = (__bf16)o.w;
        *reinterpret_cast<bf16x4*>(op + ddf * 16 + 4 * g) = o4;
        if (op32) *reinterpret_cast<float4*>(op32 + ddf * 16 + 4 * g) = o;   // backward's D reads this one
      }
      if (g == 0) {
        // row max and 1/sum are kept SEPARATELY: for a fully masked row max = -FLT_MAX and
        // max + log(sum) is not representable (the log is absorbed), which would turn the
        // reference's uniform 1/T probabilities into 1 in the backward recomputation
        const long long ri = ((long long)b * p.H + h) * T + qi[f];
        LSE[ri] = m_run[f];
        LSE[(long long)p.B * p.H * T + ri] = inv;
      }
    }
  }
}

// ---- backward, part 1: dK / dV.  One workgroup per 64-key tile, looping over 64-query tiles.
// Wave w owns keys k0 + 16 w .. + 15 for the WHOLE pipeline: it computes S and dP as
// S[query][key] = mfma(X = Q fragment, Y = K fragment), so a lane holds ONE key (lane & 15) and four
// queries (4 g + e) of each of the four 16-query blocks.  Two blocks side by side are exactly the
// 8 k-values lane group g feeds into the key-major products dV += P^T dO and dK += dS^T Q -- the
// reduction index may be enumerated in any order as long as both operands agree, and the dO^T / Q^T
// operands are transpose-read from LDS with that same enumeration (rows 32 s + 4 g .. +3 and
// 32 s + 16 + 4 g .. +3).  P and dS therefore never leave registers (the previous version wrote both
// as 64 x 64 bf16 tiles to LDS and transposed them back: two extra barriers' worth of LDS traffic per
// query tile and 36 KB of LDS).  The wave's K / V fragments are 16 registers; Q / dO tiles and the
// per-query statistics (row max, 1/sum, D, dropout row hash, far-diagonal position score) of the NEXT
// query tile are fetched while the current one is processed: one barrier per query tile.
// H2 (what is launched; the one-pass form <false> has no switch any more): the 64-query tile is processed as two halves of
// 32 queries (S / dP / P / dS of one half live at a time) under __launch_bounds__(256, 3), i.e. 168 VGPRs (24 B of
// scratch) and a third wave per SIMD for a kernel that at 255 VGPRs waits more than it issues.  Same
// arithmetic in the same order per accumulator: dK / dV are BIT-identical to the one-pass form (checked on the
// emulator for uniform / near / masked tiles with and without dropout).  Measured (profiles/r04zn_...): backward
// T = 800 845 -> 816 us, T = 400 302 -> 280 us at B = 64.
template <bool H2>
__global__ __launch_bounds__(256, H2 ? 3 : 2) void flash_bwd_dkv_kernel(
    const __bf16* __restrict__ qkv, int d, const float* __restrict__ QP, const __bf16* __restrict__ dO,
    const float* __restrict__ LSE, const float* __restrict__ Drow, __bf16* __restrict__ dqkv,
    const unsigned short* __restrict__ keepbits, const nsp_attn_mask_params p) {
  // one LDS object (see the forward kernel): Q | dO tiles (DMA images), position rows, per-query statistics, dropout words
  __shared__ __attribute__((aligned(16))) unsigned char SM[4 * 64 * KD + 2 * 64 * 16 * 4 + 4 * 2 * 64 * 4 + 2 * 128 * 4];
  unsigned char (*Qs)[64 * KD] = reinterpret_cast<unsigned char (*)[64 * KD]>(SM);
  unsigned char (*dOs)[64 * KD] = reinterpret_cast<unsigned char (*)[64 * KD]>(SM + 2 * 64 * KD);
  float (*QPs)[64][16] = reinterpret_cast<float (*)[64][16]>(SM + 4 * 64 * KD);
  float (*st_c0)[64] = reinterpret_cast<float (*)[64]>(SM + 4 * 64 * KD + 2 * 64 * 16 * 4);   // far score * sl2 - row max (uniform tiles)
  float (*st_max)[64] = st_c0 + 2;       // row max (log2 domain)
  float (*st_inv)[64] = st_c0 + 4;       // 1 / row sum (0 for rows >= T)
  float (*st_d)[64] = st_c0 + 6;         // D_i = dO_i . O_i
  // the forward's dropout decisions of the current 64-query tile against this workgroup's key tile: 4 blocks x 64 lanes x u16
  unsigned (*st_keep)[128] = reinterpret_cast<unsigned (*)[128]>(st_c0 + 8);
  const int T = p.Tq;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int h = blockIdx.x % p.H, k0 = (blockIdx.x / p.H) * 64, b = blockIdx.y;
  const long long ld3 = 3LL * d;
  const long long brow0 = (long long)b * T;
  const long long nrow = (long long)p.B * p.H * T;
  const int klen = p.klens ? p.klens[b] : T;
  const float sl2 = p.scale * LOG2E;
  const bool drop = p.dropout_p > 0.f;
  const float inv_keep = drop ? nsp_rcp(1.f - p.dropout_p) : 1.f;
  const int rp = p.r_pitch;
  if (!p.causal && p.chunk_nc == 0 && klen >= 1 && k0 >= klen) {
    // every probability of these keys is exactly 0 (see fa_key_tiles): dK = dV = 0
    for (int idx = threadIdx.x; idx < 64 * 16; idx += 256) {
      const int kk = k0 + (idx >> 4), c = (idx & 15) * 4;
      if (kk < T) {
        const long long rowoff = (brow0 + kk) * ld3 + h * DK + c;
        const bf16x4 z = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
        *reinterpret_cast<bf16x4*>(dqkv + rowoff + d) = z;
        *reinterpret_cast<bf16x4*>(dqkv + rowoff + 2 * d) = z;
      }
    }
    return;
  }
  const int key = k0 + wave * 16 + r;                 // this lane's key
  // K / V fragments of the wave's 16 keys (Y operands: row = key r, k-chunk (s, g))
  bf16x8 Kf[2], Vf[2];
  {
    const __bf16* kp_ = qkv + (brow0 + min(key, T - 1)) * ld3 + d + h * DK;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      bf16x8 kv = *reinterpret_cast<const bf16x8*>(kp_ + s2 * 32 + g * 8);
      bf16x8 vv = *reinterpret_cast<const bf16x8*>(kp_ + d + s2 * 32 + g * 8);
      if (key >= T) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { kv[e] = (__bf16)0.f; vv[e] = (__bf16)0.f; }
      }
      Kf[s2] = kv;
      Vf[s2] = vv;
    }
  }
  const __bf16* qbase = qkv + h * DK;
  const __bf16* dobase = dO + h * DK;
  // per-tile staging registers: Q / dO tiles, one float4 of the position table, and (threads 0..63) the
  // statistics of query q0 + tid
  // RAW loaded values only: every use of them (scaling, selects) happens in *_store at the END of the iteration, so the
  // loads -- issued BEFORE the next tile's DMA -- are not waited for until then
  struct Stat { float4 s4; unsigned keep; int qi; };
  // STRAIGHT-LINE loads (round 4, ISA audit): with the loads inside conditional expressions (`qi < T ? LSE[..] : 0`,
  // `c < rp ? src[c] * sl2 : 0`) hipcc put every one of them into its own branch with an s_waitcnt vmcnt(0) behind it --
  // up to eight SERIALISED global round trips at the top of every query-tile iteration of this kernel.  All addresses
  // are clamped to valid ones and everything is requested back to back.
  // dropout words of (key tile k0 / 64, query blocks q0 / 16 .. + 3): 512 contiguous bytes, one dword per thread 0..127
  // (see fa_keep_bits; a valid address also without dropout: straight-line load)
  const unsigned* kb32 = drop ? reinterpret_cast<const unsigned*>(keepbits + fa_keep_index(p, b, h, k0 >> 6, 0)) + (threadIdx.x & 127)
                              : reinterpret_cast<const unsigned*>(LSE);
  const int kb_step = drop ? 32 : 0;             // dwords per 16-query block
  // per-query statistics: ONE 16-B record {c0, row max, A, A D} written by the dQ kernel (which runs first), a
  // workgroup-uniform base + a 32-bit lane offset.  A = scale / l and A D: dS = A Pd dP - (A D) p~, and the dV operand is
  // bf16(A Pd) -- the same mantissa as bf16(Pd / l) for the power-of-two scale of d_k = 64; dV is multiplied by 1 / scale
  // once at the end (round 6: two multiplies per score fewer).  Rows beyond T contribute nothing: A = A D = 0.
  const float4* st4 = reinterpret_cast<const float4*>(Drow) + ((long long)b * p.H + h) * T;
  auto stat_load = [&](int q0) {
    Stat s_;
    s_.qi = q0 + (int)(threadIdx.x & 63);
    s_.keep = kb32[(q0 >> 4) * kb_step];
    s_.s4 = st4[min(s_.qi, T - 1)];
    return s_;
  };
  auto stat_store = [&](int buf, const Stat& s_) {
    if (threadIdx.x < 64) {
      st_c0[buf][threadIdx.x] = s_.s4.x;
      st_max[buf][threadIdx.x] = s_.s4.y;
      st_inv[buf][threadIdx.x] = s_.qi < T ? s_.s4.z : 0.f;
      st_d[buf][threadIdx.x] = s_.qi < T ? s_.s4.w : 0.f;
    }
    if (threadIdx.x < 128) st_keep[buf][threadIdx.x] = s_.keep;
  };
  auto qp_load = [&](int q0) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (QP) {
      const int ql = threadIdx.x >> 2, c4 = (threadIdx.x & 3) * 4;
      const int q = min(q0 + ql, T - 1);
      // one 16-B load (r_pitch is a multiple of 4: rows are 16-B aligned; columns >= r_pitch are zeroed in qp_store);
      // uniform base + 32-bit lane offset (an utterance's table is T * H * r_pitch floats)
      const float* src = QP + (brow0 * p.H + h) * rp;
      v = *reinterpret_cast<const float4*>(src + (__umul24((unsigned)q, (unsigned)(p.H * rp)) + (unsigned)min(c4, rp - 4)));
    }
    return v;
  };
  auto qp_store = [&](int buf, const float4& a) {
    const int c4 = (threadIdx.x & 3) * 4;
    float4 v;
    v.x = c4 + 0 < rp ? a.x * sl2 : 0.f;
    v.y = c4 + 1 < rp ? a.y * sl2 : 0.f;
    v.z = c4 + 2 < rp ? a.z * sl2 : 0.f;
    v.w = c4 + 3 < rp ? a.w * sl2 : 0.f;
    *reinterpret_cast<float4*>(&QPs[buf][threadIdx.x >> 2][c4]) = v;
  };
  float4 qpr = qp_load(0);
  Stat str = stat_load(0);
  tile_dma(Qs[0], qbase, ld3, brow0, 0, T, wave, lane);
  tile_dma(dOs[0], dobase, d, brow0, 0, T, wave, lane);
  f32x4 dk_acc[4], dv_acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { dk_acc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; dv_acc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  qp_store(0, qpr);
  stat_store(0, str);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // dropout: lane (key r, queries 4 g + e of block qq) finds its decision in the forward's lane (query 4 g + e, key group
  // r >> 2) word, bit 15 - (4 wave + (r & 3)): the four words e = 0..3 are 8 contiguous bytes of the block's 128; shift that
  // moves the bit of the even (low half-word) / odd (high) query to bit 31
  const int keep_off = 8 * g + 32 * (r >> 2);                   // byte offset inside a 16-query block's words
  const int keep_sh_odd = wave * 4 + (r & 3), keep_sh_even = 16 + keep_sh_odd;
  const int nqt = (T + 63) / 64;
  for (int qt = 0; qt < nqt; ++qt) {
    const int q0 = qt * 64;
    const int cur = qt & 1;
    // the position rows of the next query tile are read only by the 1-3 tiles around the diagonal and by masked tiles: a
    // uniform tile (the other ~10 of 13) takes its one value per query from the statistics (round 6: the table rows were
    // fetched and staged for every tile -- 0.7 GB of L2 reads per call at T = 800, B = 128)
    bool qp_next = false;
    if (qt + 1 < nqt) {
      const FaTile tn = fa_tile(p, QP != nullptr, q0 + 64, k0, T, klen);
      qp_next = QP != nullptr && !(tn.plain && tn.far);
      // (the register loads first: a wait for them then does not have to cover the DMA issued after them)
      if (qp_next) qpr = qp_load(q0 + 64);
      str = stat_load(q0 + 64);
      tile_dma(Qs[cur ^ 1], qbase, ld3, brow0, q0 + 64, T, wave, lane);
      tile_dma(dOs[cur ^ 1], dobase, d, brow0, q0 + 64, T, wave, lane);
    }
    const FaTile tl = fa_tile(p, QP != nullptr, q0, k0, T, klen);
    const bool uni = tl.plain && (QP == nullptr || tl.far);
    constexpr int NQB = H2 ? 2 : 4;          // query blocks (of 16) processed together
    auto half_tile = [&](const int hq) {
    const int qb0 = hq * NQB;
    // S[query][key], dP[query][key] for NQB query blocks: lane = key r, queries (qb0 + qb)*16 + 4g + e
    f32x4 s_acc[NQB], dp_acc[NQB];
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
      s_acc[qb] = f32x4{0.f, 0.f, 0.f, 0.f};
      dp_acc[qb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        s_acc[qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_kc_dma(Qs[cur], (qb0 + qb) * 16, s2, r, g), Kf[s2], s_acc[qb], 0, 0, 0);
        dp_acc[qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_kc_dma(dOs[cur], (qb0 + qb) * 16, s2, r, g), Vf[s2], dp_acc[qb], 0, 0, 0);
      }
    }
    bf16x8 Pt[NQB / 2], dSt[NQB / 2];     // X operands: rows = keys, k = [block 2s: queries 4g..4g+3 | block 2s+1: same]
    // The tile class (uniform / near the diagonal / masked) and dropout are wave-uniform but known only at run time:
    // tested per element inside the unrolled loops they became three scalar branches around EVERY score (ISA audit,
    // round 4: ~50 branches per tile, basic blocks of a dozen instructions, no scheduling across scores).  The element
    // loop is compiled once per class (KIND 0 uniform, 1 near, 2 general; DROP) and the class is chosen once per tile.
    auto tile_scores = [&](auto kind_, auto drop_) {
      constexpr int KIND = decltype(kind_)::value;
      constexpr bool DROP = decltype(drop_)::value;
#pragma unroll
      for (int qb = 0; qb < NQB; ++qb) {
        const int qq = qb0 + qb;             // query block inside the 64-query tile
        // The statistics rows are read through INLINE ASM: in front of a compiler-visible read of these arrays hipcc
        // put an s_waitcnt vmcnt(0) (ISA audit, round 4) -- i.e. the next query tile's Q / dO DMA, issued a few dozen
        // instructions earlier, was drained right here, before the soft-max arithmetic it is supposed to hide behind.
        // (The compiler does not count asm loads: the destinations are named in the wait statement, CDNA guide 5.7.)
        f32x4 c0v, mxv, inv, ddv;
        u32x2 kw = {0u, 0u};
        fa_lds_read4(c0v, &st_c0[cur][qq * 16 + 4 * g]);
        fa_lds_read4(mxv, &st_max[cur][qq * 16 + 4 * g]);
        fa_lds_read4(inv, &st_inv[cur][qq * 16 + 4 * g]);
        fa_lds_read4(ddv, &st_d[cur][qq * 16 + 4 * g]);
        if (DROP) fa_lds_read2(kw, reinterpret_cast<const unsigned char*>(&st_keep[cur][qq * 32]) + keep_off);
#ifndef NSP_HOST_EMULATION
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(c0v), "+v"(mxv), "+v"(inv), "+v"(ddv), "+v"(kw));
#endif
        const float c0a[4] = {c0v[0], c0v[1], c0v[2], c0v[3]}, mxa[4] = {mxv[0], mxv[1], mxv[2], mxv[3]};
        const float ina[4] = {inv[0], inv[1], inv[2], inv[3]}, dda[4] = {ddv[0], ddv[1], ddv[2], ddv[3]};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int ql = qq * 16 + 4 * g + e;
          float ex;
          bool vis = true;
          if constexpr (KIND == 0) {
            ex = __builtin_amdgcn_exp2f(fmaf(s_acc[qb][e], sl2, c0a[e]));
          } else if constexpr (KIND == 1) {          // near the diagonal, no mask predicate (fa_logits_near)
            const int dlt = q0 + ql - key;
            const int rel = min(dlt < 0 ? -dlt : dlt, p.clamp);
            ex = __builtin_amdgcn_exp2f(fmaf(s_acc[qb][e], sl2, QPs[cur][ql][rel]) - mxa[e]);
          } else {
            const int qi = q0 + ql;
            float v = s_acc[qb][e] * sl2;
            if (QP) {
              int rel = qi > key ? qi - key : key - qi;
              if (p.clamp > 0 && rel > p.clamp) rel = p.clamp;
              v += QPs[cur][ql][rel];
            }
            vis = tl.plain || fa_visible(p, klen, qi, key);
            if (!vis) v = -FLT_MAX;
            ex = __builtin_amdgcn_exp2f(v - mxa[e]);
            if (key >= T) ex = 0.f;                          // tile padding: not in the softmax
          }
          float exk = ex;
          if constexpr (DROP) {
            const int m = ((int)(kw[e >> 1] << ((e & 1) ? keep_sh_odd : keep_sh_even))) >> 31;
            exk = __uint_as_float(__float_as_uint(ex * inv_keep) & (unsigned)m);
          }
          // pdr = the value forward fed into P V, bit for bit (integer row max: see the forward kernel); the first
          // term pairs it with dP so that sum_j of it equals D_i = dO_i . O_i, the second uses the fp32
          // probability that sums to one with the saved 1 / l
          const float pdr = (float)(__bf16)exk;
          const float pa = pdr * ina[e];                  // (ina = scale / l, dda = D scale / l: see stat_store)
          float ds = fmaf(pa, dp_acc[qb][e], -ex * dda[e]);
          if (KIND == 2 && !vis) ds = 0.f;
          Pt[qb >> 1][(qb & 1) * 4 + e] = (__bf16)pa;
          dSt[qb >> 1][(qb & 1) * 4 + e] = (__bf16)ds;
        }
      }
    };
    {
      typedef std::integral_constant<int, 0> K0;
      typedef std::integral_constant<int, 1> K1;
      typedef std::integral_constant<int, 2> K2;
      if (uni) { if (drop) tile_scores(K0{}, std::true_type{}); else tile_scores(K0{}, std::false_type{}); }
      else if (tl.plain) { if (drop) tile_scores(K1{}, std::true_type{}); else tile_scores(K1{}, std::false_type{}); }
      else { if (drop) tile_scores(K2{}, std::true_type{}); else tile_scores(K2{}, std::false_type{}); }
    }
    // dV[key][dd] += sum_q Pd[q][key] dO[q][dd] ; dK[key][dk'] += sum_q dS[q][key] Q[q][dk']
#pragma unroll
    for (int sl = 0; sl < NQB / 2; ++sl) {
      const int s2 = hq * (NQB / 2) + sl;      // 32-query k-step inside the tile
      // all eight transposed fragments of the 32-query k-step requested back to back through inline asm (the builtin
      // drew an s_waitcnt vmcnt(0) -- the NEXT tile's Q / dO DMA drained in the middle of this tile -- and hipcc
      // serialised read -> wait -> MFMA per fragment), then two counted waits
      FaTr tf[8];                                     // [df * 2 + {dO, Q}]
#pragma unroll
      for (int df = 0; df < 4; ++df) {
        fa_tr_issue(tf[df * 2], dOs[cur], df * 16, 32 * s2 + 4 * g, 32 * s2 + 16 + 4 * g, r);
        fa_tr_issue(tf[df * 2 + 1], Qs[cur], df * 16, 32 * s2 + 4 * g, 32 * s2 + 16 + 4 * g, r);
      }
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        fa_tr_wait(tf, hf);
#pragma unroll
        for (int df = hf * 2; df < hf * 2 + 2; ++df) {
          dv_acc[df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Pt[sl], fa_tr_join(tf[df * 2]), dv_acc[df], 0, 0, 0);
          dk_acc[df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dSt[sl], fa_tr_join(tf[df * 2 + 1]), dk_acc[df], 0, 0, 0);
        }
      }
    }
    };
    if constexpr (H2) {
#pragma unroll 1
      for (int hq = 0; hq < 2; ++hq) half_tile(hq);
    } else {
      half_tile(0);
    }
    if (qt + 1 < nqt) {
      if (qp_next) qp_store(cur ^ 1, qpr);
      stat_store(cur ^ 1, str);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // next tile's buffers visible; this tile's free
  }
  // D[i = key (4g+e)][j = lane&15 = channel within fragment df]
  const float inv_scale = 1.f / p.scale;         // the dV operand carried the factor `scale` (see stat_store)
#pragma unroll
  for (int df = 0; df < 4; ++df)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int ko = k0 + wave * 16 + 4 * g + e;
      if (ko < T) {
        const long long rowoff = (brow0 + ko) * ld3 + h * DK + df * 16 + r;
        dqkv[rowoff + d] = (__bf16)dk_acc[df][e];
        dqkv[rowoff + 2 * d] = (__bf16)(dv_acc[df][e] * inv_scale);
      }
    }
}

// ---- backward, part 2: dQ and the relative-table gradient dQP.  One workgroup per 64-query
// tile looping over key tiles (the forward's structure: double-buffered K / V tiles, next tile's loads
// in flight during the current tile, one barrier per tile): every output element is owned by exactly
// one lane, so there is not a single global atomic (the first version accumulated dQ with fp32
// atomics from the key-parallel kernel: 88 M atomics per call at T = 800 made it 4x slower).
__global__ __launch_bounds__(256, 3) void flash_bwd_dq_kernel(
    const __bf16* __restrict__ qkv, int d, const float* __restrict__ QP, const __bf16* __restrict__ dO,
    const float* __restrict__ O32, const float* __restrict__ LSE, float* __restrict__ Drow, float* __restrict__ dq32,
    float* __restrict__ dQP, const unsigned short* __restrict__ keepbits, const __bf16* __restrict__ pos16,
    __bf16* __restrict__ dq16, const nsp_attn_mask_params p) {
  __shared__ __attribute__((aligned(16))) unsigned char KV[4 * 64 * KD + 2 * 64 * 17 * 4 + 16 * 64 * 4];   // one LDS object (see the forward kernel)
  unsigned char (*Ks)[64 * KD] = reinterpret_cast<unsigned char (*)[64 * KD]>(KV);
  unsigned char (*Vs)[64 * KD] = reinterpret_cast<unsigned char (*)[64 * KD]>(KV + 2 * 64 * KD);
  float (*QPs)[17] = reinterpret_cast<float (*)[17]>(KV + 4 * 64 * KD);
  float (*dQPs)[17] = reinterpret_cast<float (*)[17]>(KV + 4 * 64 * KD + 64 * 17 * 4);
  float (*POSs)[64] = reinterpret_cast<float (*)[64]>(KV + 4 * 64 * KD + 2 * 64 * 17 * 4);   // this head's projected position table
  const int T = p.Tq;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int h = blockIdx.x % p.H, q0 = (blockIdx.x / p.H) * 64, b = blockIdx.y;
  const long long ld3 = 3LL * d;
  const long long brow0 = (long long)b * T;
  const long long nrow = (long long)p.B * p.H * T;
  const int klen = p.klens ? p.klens[b] : T;
  const int ql = wave * 16 + r;
  if (pos16) {      // rows rel < R of pos16[R.., d] (bf16), columns of head h; read after the k-loop's barriers
    for (int idx = threadIdx.x; idx < 16 * 64; idx += 256) {
      const int rr = idx >> 6, c = idx & 63;
      POSs[rr][c] = rr < p.R ? (float)pos16[(long long)rr * d + h * DK + c] : 0.f;
    }
  }
  const int qi = q0 + ql;
  const int qc = min(qi, T - 1);
  // everything the loop reads from global memory besides the K / V tiles is fetched BEFORE tile 0
  // (see the forward kernel's note on vmcnt order)
  const __bf16* qp_ = qkv + (brow0 + qc) * ld3 + h * DK;
  const __bf16* dop = dO + (brow0 + qc) * d + h * DK;
  bf16x8 Qf[2], dOf[2];
  Qf[0] = *reinterpret_cast<const bf16x8*>(qp_ + g * 8);
  Qf[1] = *reinterpret_cast<const bf16x8*>(qp_ + 32 + g * 8);
  dOf[0] = *reinterpret_cast<const bf16x8*>(dop + g * 8);
  dOf[1] = *reinterpret_cast<const bf16x8*>(dop + 32 + g * 8);
  const float sl2 = p.scale * LOG2E;
  if (QP) {
    for (int idx = threadIdx.x; idx < 64 * p.r_pitch; idx += 256) {
      const int l2 = idx / p.r_pitch, rr = idx % p.r_pitch;
      const int q = min(q0 + l2, T - 1);
      QPs[l2][rr] = QP[((brow0 + q) * p.H + h) * p.r_pitch + rr] * sl2;
      dQPs[l2][rr] = 0.f;
    }
  }
  const long long ri = ((long long)b * p.H + h) * T + qc;
  const float rmax = LSE[ri];
  const float rinv = qi < T ? LSE[nrow + ri] : 0.f;
  // D_i = dO_i . O_i of this lane's query, formed HERE from the dO fragment the lane holds anyway and the matching 16
  // values of the forward's fp32 output (round 4: was a separate pass over dO and O, flash_dot_kernel, 1.0 ms per step
  // and 12 launches), and handed to the dK/dV kernel -- which runs after this one -- through Drow
  float dsum;
  {
    const float* op = O32 + (brow0 + qc) * d + h * DK;
    float t = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const float4 o0 = *reinterpret_cast<const float4*>(op + s2 * 32 + g * 8);
      const float4 o1 = *reinterpret_cast<const float4*>(op + s2 * 32 + g * 8 + 4);
      t = fmaf((float)dOf[s2][0], o0.x, t); t = fmaf((float)dOf[s2][1], o0.y, t);
      t = fmaf((float)dOf[s2][2], o0.z, t); t = fmaf((float)dOf[s2][3], o0.w, t);
      t = fmaf((float)dOf[s2][4], o1.x, t); t = fmaf((float)dOf[s2][5], o1.y, t);
      t = fmaf((float)dOf[s2][6], o1.z, t); t = fmaf((float)dOf[s2][7], o1.w, t);
    }
    dsum = fa_xsum4(t);
    // Round 6: everything the dK/dV kernel needs per query in ONE 16-B record {c0, row max, A, A D} -- c0 = position score
    // at the clamp distance (log2 domain) - row max (the uniform tiles' exponent offset), A = scale / l -- instead of four
    // scalar loads from three arrays per query and tile (that kernel's per-tile prologue was a third of a wave's time)
    if (g == 0 && qi < T) {
      const float farv = (QP && p.clamp > 0) ? QP[((brow0 + qc) * p.H + h) * p.r_pitch + p.clamp] * sl2 : 0.f;
      const float a_ = rinv * p.scale;
      reinterpret_cast<float4*>(Drow)[ri] = make_float4(farv - rmax, rmax, a_, a_ * dsum);
    }
  }
  const bool drop = p.dropout_p > 0.f;
  const float inv_keep = drop ? nsp_rcp(1.f - p.dropout_p) : 1.f;
  // the forward's dropout decisions of this wave's 16 queries (see fa_keep_bits): one 16-bit word per lane and key tile,
  // requested one tile ahead -- and BEFORE that tile's DMA, so that the wait for it never has to cover the DMA
  // (straight-line load from a valid address whether or not there is dropout: a conditional load becomes a branch with
  // its own s_waitcnt vmcnt(0))
  const unsigned short* kbp = drop ? keepbits + fa_keep_index(p, b, h, 0, (q0 >> 4) + wave) + lane
                                   : reinterpret_cast<const unsigned short*>(LSE);
  const long long kb_stride = drop ? 4LL * ((T + 63) >> 6) * 64 : 0;      // u16 elements between consecutive key tiles
  unsigned kb_next = *kbp;
  const __bf16* kbase = qkv + d + h * DK;
  const __bf16* vbase = qkv + 2 * d + h * DK;
  tile_dma(Ks[0], kbase, ld3, brow0, 0, T, wave, lane);
  tile_dma(Vs[0], vbase, ld3, brow0, 0, T, wave, lane);
  f32x4 dq_acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) dq_acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float far = 0.f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int nkt = fa_key_tiles(p, T, klen);
  for (int kt = 0; kt < nkt; ++kt) {
    const unsigned char* Kc = Ks[kt & 1];
    const unsigned char* Vc = Vs[kt & 1];
    const unsigned kb = kb_next;
    if (kt + 1 < nkt) {
      kb_next = kbp[(long long)(kt + 1) * kb_stride];
      tile_dma(Ks[(kt + 1) & 1], kbase, ld3, brow0, (kt + 1) * 64, T, wave, lane);
      tile_dma(Vs[(kt + 1) & 1], vbase, ld3, brow0, (kt + 1) * 64, T, wave, lane);
    }
    f32x4 s_acc[4], dp_acc[4];
#pragma unroll
    for (int kf = 0; kf < 4; ++kf) {
      s_acc[kf] = f32x4{0.f, 0.f, 0.f, 0.f};
      dp_acc[kf] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        s_acc[kf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_kc_dma(Kc, kf * 16, s, r, g), Qf[s], s_acc[kf], 0, 0, 0);
        dp_acc[kf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_kc_dma(Vc, kf * 16, s, r, g), dOf[s], dp_acc[kf], 0, 0, 0);
      }
    }
    bf16x8 dSf[2];
    const FaTile tl = fa_tile(p, QP != nullptr, q0, kt * 64, T, klen);
    const float* qrow = QP ? QPs[ql] : nullptr;
    const bool uni = tl.plain && (qrow == nullptr || tl.far);   // see the forward kernel
    const float c0 = ((uni && qrow) ? qrow[p.clamp] : 0.f) - rmax;
    const float rs_ = rinv * p.scale;
    // the element loop is compiled per tile class (KIND 0 uniform: no mask, relative term constant; 1 near the diagonal;
    // 2 masked) x dropout and chosen once per tile -- see flash_bwd_dkv_kernel (per-score tests of the wave-uniform
    // flags evaluated both exponentials and put branches around every score)
    auto tile_scores = [&](auto kind_, auto drop_) {
      constexpr int KIND = decltype(kind_)::value;
      constexpr bool DROP = decltype(drop_)::value;
      float ev[4][4];
      unsigned vis = 0xFFFFu;
      if constexpr (KIND == 1) fa_logits_near(s_acc, ev, qrow, sl2, qi, kt * 64, g, p.clamp);
      if constexpr (KIND == 2) vis = fa_logits(s_acc, ev, p, qrow, sl2, qi, kt * 64, g, klen, tl);
#pragma unroll
      for (int kf = 0; kf < 4; ++kf)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int key = kt * 64 + kf * 16 + 4 * g + e;
          float ex;
          if constexpr (KIND == 0) ex = __builtin_amdgcn_exp2f(fmaf(s_acc[kf][e], sl2, c0));
          else ex = __builtin_amdgcn_exp2f(ev[kf][e] - rmax);
          if (KIND == 2 && !tl.plain && key >= T) ex = 0.f;
          // forward's P V operand, bit for bit (see flash_bwd_dkv_kernel)
          const float pdr = (float)(__bf16)(DROP ? fa_keep_apply(ex * inv_keep, kb, kf * 4 + e) : ex);
          float ds = rs_ * fmaf(pdr, dp_acc[kf][e], -ex * dsum);
          if (KIND == 2 && !tl.plain && !((vis >> (kf * 4 + e)) & 1u)) ds = 0.f;
          dSf[kf >> 1][(kf & 1) * 4 + e] = (__bf16)ds;
          if constexpr (KIND == 0) {
            far += ds;                         // (only read when there is a relative term: a uniform tile is a far one then)
          } else {
            if (QP) {
              if (tl.far) {
                far += ds;
              } else if (ds != 0.f) {
                int rel = qi > key ? qi - key : key - qi;
                if (p.clamp > 0 && rel > p.clamp) rel = p.clamp;
                if (p.clamp > 0 && rel == p.clamp) far += ds;
                else atomicAdd(&dQPs[ql][rel], ds);  // LDS, near-diagonal elements only
              }
            }
          }
        }
    };
    {
      typedef std::integral_constant<int, 0> K0;
      typedef std::integral_constant<int, 1> K1;
      typedef std::integral_constant<int, 2> K2;
      if (uni) { if (drop) tile_scores(K0{}, std::true_type{}); else tile_scores(K0{}, std::false_type{}); }
      else if (tl.plain) { if (drop) tile_scores(K1{}, std::true_type{}); else tile_scores(K1{}, std::false_type{}); }
      else { if (drop) tile_scores(K2{}, std::true_type{}); else tile_scores(K2{}, std::false_type{}); }
    }
    // dQ^T[dk'][query] += K^T dS^T : X = K^T fragment (rows dk', k = keys), Y = dS fragment
    {
      FaTr kt_[8];                                    // [df * 2 + s]
#pragma unroll
      for (int i = 0; i < 8; ++i) fa_tr_issue(kt_[i], Kc, (i >> 1) * 16, 32 * (i & 1) + 4 * g, 32 * (i & 1) + 16 + 4 * g, r);
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        fa_tr_wait(kt_, hf);
#pragma unroll
        for (int i = hf * 4; i < hf * 4 + 4; ++i)
          dq_acc[i >> 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa_tr_join(kt_[i]), dSf[i & 1], dq_acc[i >> 1], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (dq32 && qi < T) {
    float* dqp = dq32 + (brow0 + qi) * d + h * DK;
#pragma unroll
    for (int df = 0; df < 4; ++df)
      *reinterpret_cast<float4*>(dqp + df * 16 + 4 * g) =
          make_float4(dq_acc[df][0], dq_acc[df][1], dq_acc[df][2], dq_acc[df][3]);
  }
  if (QP) {
    if (p.clamp > 0) {
      far += __shfl_xor(far, 16, 64);
      far += __shfl_xor(far, 32, 64);
      if (g == 0) atomicAdd(&dQPs[ql][p.clamp], far);
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 64 * p.r_pitch; idx += 256) {
      const int l2 = idx / p.r_pitch, rr = idx % p.r_pitch;
      if (q0 + l2 < T) dQP[((brow0 + q0 + l2) * p.H + h) * p.r_pitch + rr] = dQPs[l2][rr];
    }
  }
  if (dq16) {
    // Round 6: the query gradient leaves this kernel FINISHED -- the position term's share dQP . pos (the reference's
    // BD = q pos^T, relative_multihead_attention.py:188-193; was a K = 16 GEMM that read the fp32 dQ back, accumulated
    // into it, and a cast pass: 18 bytes per element of fp32 hand-overs) is 11 multiply-adds per output here, on the table
    // gradient this workgroup has just finished in LDS -- and as the bf16 operand of the QKV gradient GEMMs, in column
    // block 0 of dqkv.
    if (pos16 && QP) {
      for (int rr = 0; rr < p.R; ++rr) {
        const float w = dQPs[ql][rr];
#pragma unroll
        for (int df = 0; df < 4; ++df) {
          const float4 pv = *reinterpret_cast<const float4*>(&POSs[rr][df * 16 + 4 * g]);
          dq_acc[df][0] = fmaf(w, pv.x, dq_acc[df][0]); dq_acc[df][1] = fmaf(w, pv.y, dq_acc[df][1]);
          dq_acc[df][2] = fmaf(w, pv.z, dq_acc[df][2]); dq_acc[df][3] = fmaf(w, pv.w, dq_acc[df][3]);
        }
      }
    }
    if (qi < T) {
      __bf16* dqp = dq16 + (brow0 + qi) * ld3 + h * DK;
#pragma unroll
      for (int df = 0; df < 4; ++df) {
        bf16x4 o4;
        o4[0] = (__bf16)dq_acc[df][0]; o4[1] = (__bf16)dq_acc[df][1]; o4[2] = (__bf16)dq_acc[df][2]; o4[3] = (__bf16)dq_acc[df][3];
        *reinterpret_cast<bf16x4*>(dqp + df * 16 + 4 * g) = o4;
      }
    }
  }
}

}  // namespace

extern "C" long long nsp_flash_attn_keepbits_bytes(int B, int H, int T) {
  const long long nkt = (T + 63) / 64;
  return (long long)B * H * nkt * (4 * nkt) * 128;
}

extern "C" int nsp_flash_attn_fwd(const void* qkv, int d, const float* QP, void* O, float* O32, float* LSE, void* keepbits,
                                  const nsp_attn_mask_params* pp, void* stream) {
  if (!pp || !qkv || !O || !LSE) return NSP_EINVAL;
  nsp_attn_mask_params p = *pp;
  if (p.dropout_p > 0.f && !keepbits) return NSP_EINVAL;
  if (p.Tq != p.Tk || d != p.H * DK) return NSP_EUNSUPPORTED;
  if (QP && !(p.clamp > 0 && p.R <= 16 && p.r_pitch <= 16 && p.R >= (p.clamp + 1 < p.Tk ? p.clamp + 1 : p.Tk)))
    return NSP_EUNSUPPORTED;
  if (p.r_pitch < p.R) p.r_pitch = p.R;
  // 16 queries per wave (flash_fwd_kernel<1>): measured faster than 32 (occupancy 3 vs 2; round 3: 228 vs 311 us at T = 800,
  // B = 64; the <2> instantiation and its switch were removed in round 5).  Forcing 4 waves per SIMD -- 128 VGPRs, 8 dwords
  // spilled -- measured equal to the 137-VGPR / 3-wave build.
  {
    dim3 grid(((p.Tq + 63) / 64) * p.H, p.B);
    hipLaunchKernelGGL(flash_fwd_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const __bf16*>(qkv), d, QP, reinterpret_cast<__bf16*>(O), O32, LSE,
                       reinterpret_cast<unsigned short*>(keepbits), p);
  }
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

// dqkv receives dK at column block d and dV at 2d (bf16); dQP is plainly written (no zero-init needed); the query gradient
// either as dq32 [B*T, d] fp32 WITHOUT the position term's share (pos16 must be NULL), or -- dq32 == NULL -- finished, as
// bf16 in column block 0 of dqkv: dS K plus, when pos16 (the projected position table [>= R, d] bf16) is given, dQP . pos16.
// D is scratch [B,H,T,4] (the dQ kernel leaves a 16-B record per query for the dK/dV kernel); keepbits = what the forward call with the same parameters wrote
// (required iff dropout_p > 0).
extern "C" int nsp_flash_attn_bwd(const void* qkv, int d, const float* QP, const void* dO, const float* O32,
                                  const float* LSE, const void* keepbits, float* D, void* dqkv, float* dq32, float* dQP,
                                  const void* pos16, const nsp_attn_mask_params* pp, void* stream) {
  if (!pp || !qkv || !dO || !O32 || !LSE || !D || !dqkv) return NSP_EINVAL;
  nsp_attn_mask_params p = *pp;
  if (p.dropout_p > 0.f && !keepbits) return NSP_EINVAL;
  if (dq32 && pos16) return NSP_EINVAL;          // the position term is only folded into the finished (bf16) query gradient
  if (p.Tq != p.Tk || d != p.H * DK) return NSP_EUNSUPPORTED;
  if (QP && !(p.clamp > 0 && p.R <= 16 && p.r_pitch <= 16)) return NSP_EUNSUPPORTED;
  if (QP && !dQP) return NSP_EINVAL;
  if (p.r_pitch < p.R) p.r_pitch = p.R;
  if (QP && (p.r_pitch % 4 != 0 || p.r_pitch < 4)) return NSP_EUNSUPPORTED;     // (the dK/dV kernel loads table rows as float4)
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(((p.Tq + 63) / 64) * p.H, p.B);
  // the dQ kernel first: it forms D = dO . O for its queries and leaves it in D for the dK/dV kernel
  hipLaunchKernelGGL(flash_bwd_dq_kernel, grid, dim3(256), 0, st, reinterpret_cast<const __bf16*>(qkv), d, QP,
                     reinterpret_cast<const __bf16*>(dO), O32, LSE, D, dq32, dQP,
                     reinterpret_cast<const unsigned short*>(keepbits), reinterpret_cast<const __bf16*>(pos16),
                     dq32 ? nullptr : reinterpret_cast<__bf16*>(dqkv), p);
  // the 64-query tile as two halves of 32 (168 VGPRs, a third wave per SIMD; the whole-tile form -- 255 VGPRs -- and its
  // switch were removed in round 5: bit-identical results, 3-7 % slower, profiles/r04zn_flash_dkv_two_halves_ab.log)
  hipLaunchKernelGGL(flash_bwd_dkv_kernel<true>, grid, dim3(256), 0, st, reinterpret_cast<const __bf16*>(qkv), d,
                     QP, reinterpret_cast<const __bf16*>(dO), LSE, D, reinterpret_cast<__bf16*>(dqkv),
                     reinterpret_cast<const unsigned short*>(keepbits), p);
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}
