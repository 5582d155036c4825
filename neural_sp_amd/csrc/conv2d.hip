// conv2d.hip -- VGG-style Conv2d frontend (reference conv.py:289-396) on
// channels-last [B,T,F,C] activations.
//
//   * 3x3 / pad 1 / stride 1, C_in = C_out = 32: MFMA kernel.  A block owns an
//     8(time) x 16(freq) output tile; its 10x18 input halo is staged ONCE in LDS
//     (bf16 in NSP_COMPUTE_BF16, fp32 in NSP_COMPUTE_F32) and the 9 taps are read
//     as shifted MFMA A-fragments straight out of that tile -- no im2col copy
//     ever exists, HBM sees each input pixel ~1.4x.  The 32x288 filter bank
//     lives in registers (bf16) or LDS (fp32).  The same kernel computes the
//     data gradient with the tap-flipped, channel-transposed filter bank.
//   * C_in = 1 (first layer): direct kernel, HBM-bound on the 32-channel write.
//   * weight gradients: register-tiled SIMT kernels (persistent blocks, one
//     atomic flush per block).
//   * MaxPool2d(ceil_mode=True) with argmax for the backward gather; can emit the
//     reference's [B,T',C*F'] feature order (conv.py:189) directly.
// Filter layout everywhere: w[co][3][3][ci] (tap-major, ci innermost).
#include "common.h"

namespace {

constexpr int TT = 8, TF = 16, HT = TT + 2, HF = TF + 2, CH = 32;

template <int MODE> struct ConvCfg;
// 32 bf16 + 32 B pad: with 16 B of padding the ds_read_b128 pixel-per-lane fragments took 8 LDS cycles instead
// of 4 (measured: 49.5 % of the LDS cycles of the conv kernels were bank-conflict cycles)
template <> struct ConvCfg<0> { static constexpr int PIX_PITCH = 96; };
template <> struct ConvCfg<1> { static constexpr int PIX_PITCH = 144; };   // 32 fp32 + 16 B pad
constexpr int W_PITCH = 292;  // floats per co row of the fp32 LDS filter bank (288 + 4 pad)

// Feature maps [B,T,F,32] are stored as bf16 in the throughput mode (TIO = __bf16): every consumer rounds
// them to bf16 for its MFMA anyway, so the stored rounding is numerically free, and the four
// full-resolution maps of the front-end (0.9 GB each in fp32 at batch 64) are what its kernels stream.
template <typename T> __device__ __forceinline__ float4 ld4(const T* p);
template <> __device__ __forceinline__ float4 ld4<float>(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <> __device__ __forceinline__ float4 ld4<__bf16>(const __bf16* p) {
  const bf16x4 h = *reinterpret_cast<const bf16x4*>(p);
  return make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
}
template <typename T> __device__ __forceinline__ void st4(T* p, const float4& v);
template <> __device__ __forceinline__ void st4<float>(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }
template <> __device__ __forceinline__ void st4<__bf16>(__bf16* p, const float4& v) {
  bf16x4 h;
  h[0] = (__bf16)v.x; h[1] = (__bf16)v.y; h[2] = (__bf16)v.z; h[3] = (__bf16)v.w;
  *reinterpret_cast<bf16x4*>(p) = h;
}

template <int MODE, typename TIO>
__global__ __launch_bounds__(256) void conv3x3_c32_kernel(const TIO* __restrict__ x,
                                                          const float* __restrict__ w,
                                                          const float* __restrict__ bias,
                                                          TIO* __restrict__ y, int B, int T, int F,
                                                          int relu, const TIO* __restrict__ mask_src,
                                                          int tiles_f, int tiles_t) {
  // Persistent workgroups: the 32x288 filter bank (36 x 16 B per lane) is fetched ONCE per
  // workgroup and kept in registers (bf16 mode) / LDS (fp32 mode) while the workgroup walks
  // over output tiles; re-fetching it per 128-pixel tile cost 6x the input traffic.
  // mask_src (optional): out = mask_src > 0 ? out : 0 -- the ReLU backward of the PREVIOUS
  // layer fused into this layer's data-gradient pass.
  constexpr int PP = ConvCfg<MODE>::PIX_PITCH;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* xs = smem;                                          // [HT*HF][PP]
  float* ws = reinterpret_cast<float*>(smem + HT * HF * PP);          // MODE 1 only: [32][W_PITCH]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;

  bf16x8 bfrag[9][2];
  if (MODE == 0) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const float* wp = w + ((long long)(nb * 16 + r) * 9 + tap) * CH + g * 8;
        const float4 lo = reinterpret_cast<const float4*>(wp)[0];
        const float4 hi = reinterpret_cast<const float4*>(wp)[1];
        bf16x8 h;
        h[0] = (__bf16)lo.x; h[1] = (__bf16)lo.y; h[2] = (__bf16)lo.z; h[3] = (__bf16)lo.w;
        h[4] = (__bf16)hi.x; h[5] = (__bf16)hi.y; h[6] = (__bf16)hi.z; h[7] = (__bf16)hi.w;
        bfrag[tap][nb] = h;
      }
  } else {
    for (int idx = tid; idx < CH * 288 / 4; idx += 256) {
      const int co = idx / 72, q = idx % 72;
      *reinterpret_cast<float4*>(ws + co * W_PITCH + q * 4) =
          reinterpret_cast<const float4*>(w + (long long)co * 288)[q];
    }
  }
  float4 bb[2];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
    bb[nb] = bias ? reinterpret_cast<const float4*>(bias + nb * 16 + g * 4)[0] : make_float4(0.f, 0.f, 0.f, 0.f);
  const long long ntiles = (long long)B * tiles_t * tiles_f;
  // bf16 mode: the halo of the NEXT tile is fetched into registers before the MFMAs of the current
  // one and written to the other LDS buffer after its epilogue: one barrier per tile and the
  // global-load latency hides behind the tile's compute + stores (the single-buffered version
  // ran load -> barrier -> 36 MFMAs -> store -> barrier: 2.6 TB/s, neither HBM- nor MFMA-bound)
  constexpr bool IO16 = sizeof(TIO) == 2;
  // fp32 maps: 8 float4 per pixel, converted to bf16 on commit; bf16 maps: 4 x 16-B chunks per pixel copied
  // verbatim (the first bf16 version went through 8-byte loads + unpack + repack: 1.5x SLOWER than fp32 maps)
  constexpr int CPP = IO16 ? 4 : 8;                        // chunks per pixel
  constexpr int NST = (HT * HF * CPP + 255) / 256;         // chunks per thread per halo tile
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
  float4 pre[IO16 ? 1 : NST];
  u32x4_t pre16[IO16 ? NST : 1];
  // Every lane loads (out-of-range pixels / chunk slots with clamped addresses) and the zero padding is
  // applied at commit time from a bit mask: predicated loads become branches, behind which hipcc's wait
  // insertion only knows vmcnt(0) -- and with in-order retirement that also waits for the tile's stores.
  unsigned pre_ok = 0u;
  auto halo_fetch = [&](long long tl) {
    const int tf = (int)(tl % tiles_f);
    const int tt = (int)((tl / tiles_f) % tiles_t);
    const long long b = tl / ((long long)tiles_f * tiles_t);
    pre_ok = 0u;
#pragma unroll
    for (int i = 0; i < NST; ++i) {
      const int idx = min(tid + i * 256, HT * HF * CPP - 1);
      const int cc = idx % CPP, pix = idx / CPP;
      const int ht = pix / HF, hf = pix % HF;
      const int t = tt * TT + ht - 1, f = tf * TF + hf - 1;
      const bool ok = t >= 0 && t < T && f >= 0 && f < F;
      pre_ok |= (ok ? 1u : 0u) << i;
      const int tc = min(max(t, 0), T - 1), fc = min(max(f, 0), F - 1);
      if constexpr (IO16) {
        pre16[i] = *reinterpret_cast<const u32x4_t*>(x + ((b * T + tc) * F + fc) * CH + cc * 8);
      } else {
        pre[i] = ld4<TIO>(x + ((b * T + tc) * F + fc) * CH + cc * 4);
      }
    }
  };
  auto halo_commit = [&](unsigned char* dst) {
#pragma unroll
    for (int i = 0; i < NST; ++i) {
      const int idx = tid + i * 256;
      const bool ok = (pre_ok >> i) & 1u;
      if (idx < HT * HF * CPP) {
        const int cc = idx % CPP, pix = idx / CPP;
        if constexpr (IO16) {
          *reinterpret_cast<u32x4_t*>(dst + pix * PP + cc * 16) = ok ? pre16[i] : u32x4_t{0u, 0u, 0u, 0u};
        } else {
          bf16x4 h;
          h[0] = (__bf16)(ok ? pre[i].x : 0.f); h[1] = (__bf16)(ok ? pre[i].y : 0.f);
          h[2] = (__bf16)(ok ? pre[i].z : 0.f); h[3] = (__bf16)(ok ? pre[i].w : 0.f);
          *reinterpret_cast<bf16x4*>(dst + pix * PP + cc * 8) = h;
        }
      }
    }
  };
  int cur = 0;
  if (MODE == 0 && (long long)blockIdx.x < ntiles) {
    halo_fetch(blockIdx.x);
    halo_commit(smem);
    __syncthreads();
  }
  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int tf = (int)(tile % tiles_f);
    const int tt = (int)((tile / tiles_f) % tiles_t);
    const long long b = tile / ((long long)tiles_f * tiles_t);
    const int f0 = tf * TF, t0 = tt * TT;
    const bool has_next = tile + gridDim.x < ntiles;
    if (MODE == 0) {
      xs = smem + cur * (HT * HF * PP);
      if (has_next) halo_fetch(tile + gridDim.x);
    } else {
      __syncthreads();  // previous tile's fragment reads are done
      // ---- stage the halo tile: HT*HF pixels x 8 float4
      for (int idx = tid; idx < HT * HF * 8; idx += 256) {
        const int c4 = idx & 7, pix = idx >> 3;
        const int ht = pix / HF, hf = pix % HF;
        const int t = t0 + ht - 1, f = f0 + hf - 1;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t >= 0 && t < T && f >= 0 && f < F)
          v = ld4<TIO>(x + ((b * T + t) * F + f) * CH + c4 * 4);
        *reinterpret_cast<float4*>(xs + pix * PP + c4 * 16) = v;
      }
      __syncthreads();
    }
    f32x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) acc[a][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (MODE == 0) {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int dt = tap / 3, df = tap % 3;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          const int pix = (wave * 2 + a + dt) * HF + r + df;
          const bf16x8 af = *reinterpret_cast<const bf16x8*>(xs + pix * PP + g * 16);
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
            acc[a][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfrag[tap][nb], af, acc[a][nb], 0, 0, 0);
        }
      }
    } else {
      for (int tap = 0; tap < 9; ++tap) {
        const int dt = tap / 3, df = tap % 3;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          float af[2], bf[2];
#pragma unroll
          for (int a = 0; a < 2; ++a) {
            const int pix = (wave * 2 + a + dt) * HF + r + df;
            af[a] = *reinterpret_cast<const float*>(xs + pix * PP + (s * 4 + g) * 4);
          }
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) bf[nb] = ws[(nb * 16 + r) * W_PITCH + tap * CH + s * 4 + g];
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
              acc[a][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[nb], af[a], acc[a][nb], 0, 0, 0);
        }
      }
    }
    // ---- epilogue: lane holds pixel (t0+2*wave+a, f0+r), co = nb*16 + g*4 .. +3.
    // All mask loads are issued before the first store (and the bias lives in registers, loaded once per
    // workgroup): gfx9 retires vector-memory operations in order, so a load issued behind a store can only
    // be waited for together with that store -- load/wait/store per output chunk exposed three store
    // round trips per 128-pixel tile.
    const int f = f0 + r;
    float4 mk[2][2];
    if (mask_src) {
      // (clamped, unconditional: a per-lane predicate around the load makes hipcc consume it inside the branch)
      const int fc = min(f, F - 1);
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int tc = min(t0 + wave * 2 + a, T - 1);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) mk[a][nb] = ld4<TIO>(mask_src + ((b * T + tc) * F + fc) * CH + nb * 16 + g * 4);
      }
    } else {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) mk[a][nb] = make_float4(1.f, 1.f, 1.f, 1.f);
    }
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int t = t0 + wave * 2 + a;
      if (t >= T || f >= F) continue;
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const int co = nb * 16 + g * 4;
        const long long off = ((b * T + t) * F + f) * CH + co;
        float4 v = make_float4(acc[a][nb][0] + bb[nb].x, acc[a][nb][1] + bb[nb].y, acc[a][nb][2] + bb[nb].z,
                               acc[a][nb][3] + bb[nb].w);
        if (relu) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        const float4 m = mk[a][nb];
        v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f;
        v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
        st4<TIO>(y + off, v);
      }
    }
    if (MODE == 0) {
      if (has_next) halo_commit(smem + (cur ^ 1) * (HT * HF * PP));
      __syncthreads();  // next buffer complete, every wave done reading the current one
      cur ^= 1;
    }
  }
}

// ---- bf16 feature maps, throughput mode: the kernel above restated with buffer addressing and a two-tile
// prefetch.  Why: (1) gfx9 retires vector-memory operations in issue order, so waiting for a load that
// was issued after a store also waits for that store to reach L2; the kernel above fetched the next halo
// and the ReLU-mask chunks behind the previous tile's stores and exposed a store round trip per
// 128-pixel tile (~5 us of work).  Here the halo of tile i+2 and the mask chunks of tile i+1 are
// requested at the top of iteration i, i.e. BEFORE the stores of tile i, and are consumed in iteration
// i+1: every wait covers only loads that are older than the last stores.  (2) that only holds if hipcc can
// count: a predicated load or store is a branch, and behind a branch its wait insertion assumes the
// minimum, which degenerates to vmcnt(0).  Buffer instructions return 0 / drop the write for offsets
// beyond num_records, so edge pixels, the zero padding of the halo and tiles past the end are handled by
// an out-of-range offset instead of a predicate and the tile loop is straight-line code.
// Requires B*T*F*64 bytes < 4 GiB per launch (the launcher cuts the batch otherwise).
typedef __attribute__((ext_vector_type(4))) unsigned int cu32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int cu32x2;

template <bool HAS_MASK>
__global__ __launch_bounds__(256) void conv3x3_c32_b16_kernel(const __bf16* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ bias, __bf16* __restrict__ y,
                                                              int B, int T, int F, int relu,
                                                              const __bf16* __restrict__ mask_src, int tiles_f,
                                                              int tiles_t, unsigned map_bytes) {
  constexpr int PP = ConvCfg<0>::PIX_PITCH;
  constexpr int NCH = HT * HF * 4;                 // 16-B chunks of one halo tile (4 per pixel)
  constexpr int NST = (NCH + 255) / 256;
  constexpr unsigned OOB = 0xFFFFFFFFu;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(x), 0, map_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(y, 0, map_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rm =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(HAS_MASK ? mask_src : x), 0, map_bytes, 0x00020000);

  bf16x8 bfrag[9][2];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const float* wp = w + ((long long)(nb * 16 + r) * 9 + tap) * CH + g * 8;
      const float4 lo = reinterpret_cast<const float4*>(wp)[0];
      const float4 hi = reinterpret_cast<const float4*>(wp)[1];
      bf16x8 h;
      h[0] = (__bf16)lo.x; h[1] = (__bf16)lo.y; h[2] = (__bf16)lo.z; h[3] = (__bf16)lo.w;
      h[4] = (__bf16)hi.x; h[5] = (__bf16)hi.y; h[6] = (__bf16)hi.z; h[7] = (__bf16)hi.w;
      bfrag[tap][nb] = h;
    }
  float4 bb[2];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
    bb[nb] = bias ? reinterpret_cast<const float4*>(bias + nb * 16 + g * 4)[0] : make_float4(0.f, 0.f, 0.f, 0.f);
  const long long ntiles = (long long)B * tiles_t * tiles_f;
  const long long G = gridDim.x;

  // halo chunk slots of this thread (tile independent): LDS byte offset and (dt, df, channel chunk)
  int h_ht[NST], h_hf[NST], h_cc[NST];
#pragma unroll
  for (int i = 0; i < NST; ++i) {
    const int idx = tid + i * 256;
    const int pix = idx / 4;
    h_cc[i] = idx % 4; h_ht[i] = pix / HF; h_hf[i] = pix % HF;
  }
  auto fetch_halo = [&](long long tl, cu32x4 (&h)[NST]) {
    const bool live = tl < ntiles;
    const int tf = (int)(tl % tiles_f);
    const int tt = (int)((tl / tiles_f) % tiles_t);
    const long long b = tl / ((long long)tiles_f * tiles_t);
#pragma unroll
    for (int i = 0; i < NST; ++i) {
      const int t = tt * TT + h_ht[i] - 1, f = tf * TF + h_hf[i] - 1;
      const bool ok = live && tid + i * 256 < NCH && t >= 0 && t < T && f >= 0 && f < F;
      const unsigned off = ok ? (unsigned)((((b * T + t) * F + f) * CH + h_cc[i] * 8) * 2) : OOB;
      h[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0);
    }
  };
  // output / mask chunk offsets of this lane in tile tl: pixel (t0 + 2*wave + a, f0 + r), channels nb*16 + g*4 .. +3
  auto out_off = [&](long long tl, int a, int nb) -> unsigned {
    const int tf = (int)(tl % tiles_f);
    const int tt = (int)((tl / tiles_f) % tiles_t);
    const long long b = tl / ((long long)tiles_f * tiles_t);
    const int t = tt * TT + wave * 2 + a, f = tf * TF + r;
    const bool ok = tl < ntiles && t < T && f < F;
    return ok ? (unsigned)((((b * T + t) * F + f) * CH + nb * 16 + g * 4) * 2) : OOB;
  };
  // the same rows as 16-B chunks: lane = (pixel f0 + (lane >> 2), chunk lane & 3) of row t0 + 2 * wave + a
  auto row_off = [&](long long tl, int a) -> unsigned {
    const int tf = (int)(tl % tiles_f);
    const int tt = (int)((tl / tiles_f) % tiles_t);
    const long long b = tl / ((long long)tiles_f * tiles_t);
    const int t = tt * TT + wave * 2 + a, f = tf * TF + (lane >> 2);
    const bool ok = tl < ntiles && t < T && f < F;
    return ok ? (unsigned)((((b * T + t) * F + f) * CH) * 2 + (lane & 3) * 16) : OOB;
  };
  auto fetch_mask = [&](long long tl, cu32x2 (&m)[2][2]) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        if constexpr (HAS_MASK) m[a][nb] = __builtin_amdgcn_raw_buffer_load_b64(rm, out_off(tl, a, nb), 0, 0);
        else m[a][nb] = cu32x2{0u, 0u};
      }
  };
  auto commit = [&](const cu32x4 (&h)[NST], unsigned char* dst) {
#pragma unroll
    for (int i = 0; i < NST; ++i)
      if (tid + i * 256 < NCH)
        *reinterpret_cast<cu32x4*>(dst + (h_ht[i] * HF + h_hf[i]) * PP + h_cc[i] * 16) = h[i];
  };
  auto lds_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (not __syncthreads: its fence waits vmcnt(0) = for the stores)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  int cur = 0;
  // one tile: MFMAs on LDS[cur], stage the next tile's halo (C.h) into LDS[cur^1], epilogue with C.m
  auto body = [&](long long tile, const cu32x4 (&ch)[NST], const cu32x2 (&cm)[2][2]) {
    const unsigned char* xs = smem + cur * (HT * HF * PP);
    f32x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) acc[a][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dt = tap / 3, df = tap % 3;
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int pix = (wave * 2 + a + dt) * HF + r + df;
        const bf16x8 af = *reinterpret_cast<const bf16x8*>(xs + pix * PP + g * 16);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
          acc[a][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfrag[tap][nb], af, acc[a][nb], 0, 0, 0);
      }
    }
    commit(ch, smem + (cur ^ 1) * (HT * HF * PP));
    // Round 6: the outputs leave through a per-wave LDS slab so that every store instruction writes ONE row of 16 pixels =
    // 1 KB contiguous (16 B per lane); the MFMA layout gives a lane 4 channels (8 B) of a pixel, i.e. 32-B pieces at a
    // 64-B stride per instruction -- partial lines, the same pattern that made the GEMMs' direct epilogue lose to the staged
    // one.  Slab: [16 pixels][80 B] (pitch 80: 16-B aligned rows, 2-way on the 8-B writes).
    unsigned char* slab = smem + 2 * (HT * HF * PP) + wave * (16 * 80);
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        float v[4] = {acc[a][nb][0] + bb[nb].x, acc[a][nb][1] + bb[nb].y, acc[a][nb][2] + bb[nb].z,
                      acc[a][nb][3] + bb[nb].w};
        if (relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if constexpr (HAS_MASK) {
          const cu32x2 m = cm[a][nb];
          // bf16 > 0  <=>  sign bit clear and not (+-)zero
          const unsigned mv[4] = {m.x << 16, m.x & 0xFFFF0000u, m.y << 16, m.y & 0xFFFF0000u};
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = __uint_as_float(mv[e]) > 0.f ? v[e] : 0.f;
        }
        bf16x4 h;
        h[0] = (__bf16)v[0]; h[1] = (__bf16)v[1]; h[2] = (__bf16)v[2]; h[3] = (__bf16)v[3];
        *reinterpret_cast<bf16x4*>(slab + r * 80 + nb * 32 + g * 8) = h;
      }
      // (one wave, in-order LDS queue: the reads below see the writes above; the slab is private to the wave)
      __builtin_amdgcn_wave_barrier();
      const cu32x4 o = *reinterpret_cast<const cu32x4*>(slab + (lane >> 2) * 80 + (lane & 3) * 16);
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_raw_buffer_store_b128(o, ry, row_off(tile, a), 0, 0);
    }
    lds_barrier();   // next halo complete in LDS[cur^1], every wave done reading LDS[cur]
    cur ^= 1;
  };

  cu32x4 ha[NST], hb[NST];
  cu32x2 ma[2][2], mb[2][2];
  long long tile = blockIdx.x;
  fetch_halo(tile, ha);
  commit(ha, smem);
  fetch_halo(tile + G, ha);
  fetch_mask(tile, ma);
  lds_barrier();
  while (tile < ntiles) {
    fetch_halo(tile + 2 * G, hb);
    fetch_mask(tile + G, mb);
    body(tile, ha, ma);
    tile += G;
    if (tile >= ntiles) break;
    fetch_halo(tile + 2 * G, ha);
    fetch_mask(tile + G, ma);
    body(tile, hb, mb);
    tile += G;
  }
}

// ---- first layer: C_in = 1 -> 32, direct.  A workgroup owns `rows_per_block` consecutive (b,t)
// rows: their input rows (+1 halo row each side, zero halo columns) are staged in LDS once; thread =
// (4 output channels with their 9x4 filter taps in registers, one of 32 pixel lanes); the only
// global stream is the 128 B/pixel output, written as float4.  (The first version re-derived
// (b,t,f) with 64-bit divisions per element and issued 9 predicated global loads: 1.4 TB/s.)
template <typename TY>
__global__ __launch_bounds__(256) void conv3x3_c1_kernel(const float* __restrict__ x,
                                                         const float* __restrict__ w,
                                                         const float* __restrict__ bias,
                                                         TY* __restrict__ y, int B, int T, int F,
                                                         int relu, int rows_per_block) {
  extern __shared__ __attribute__((aligned(16))) float xs1[];  // [(rows_per_block + 2)][F + 2]
  const int cg = threadIdx.x & 7, pl = threadIdx.x >> 3;
  const int BT = B * T, FP = F + 2;
  const int r0 = blockIdx.x * rows_per_block;
  const int nr = min(BT, r0 + rows_per_block) - r0;
  for (int idx = threadIdx.x; idx < (nr + 2) * FP; idx += blockDim.x) {
    const int gr = r0 + idx / FP - 1, fc = idx % FP - 1;
    xs1[idx] = (gr >= 0 && gr < BT && fc >= 0 && fc < F) ? x[(long long)gr * F + fc] : 0.f;
  }
  float wr[9][4];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int e = 0; e < 4; ++e) wr[tap][e] = w[(cg * 4 + e) * 9 + tap];
  const float4 b4 = bias ? reinterpret_cast<const float4*>(bias)[cg] : make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  const int t0 = r0 % T;
  int lr = pl / F, f = pl % F;
  TY* yb = y + (long long)r0 * F * CH + cg * 4;
#pragma unroll 2
  for (int q = pl; q < nr * F; q += 32) {
    int t = t0 + lr;
    if (t >= T) t -= T;
    // a halo row that belongs to the neighbouring utterance (or lies outside) counts as zero
    const float m_up = t > 0 ? 1.f : 0.f, m_dn = t < T - 1 ? 1.f : 0.f;
    const float* xc = xs1 + (lr + 1) * FP + (f + 1);
    float4 acc = b4;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dt = tap / 3 - 1, df = tap % 3 - 1;
      float xv = xc[dt * FP + df];
      if (dt < 0) xv *= m_up;
      if (dt > 0) xv *= m_dn;
      acc.x += xv * wr[tap][0]; acc.y += xv * wr[tap][1]; acc.z += xv * wr[tap][2]; acc.w += xv * wr[tap][3];
    }
    if (relu) {
      acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f);
    }
    st4<TY>(yb + (long long)q * CH, acc);
    f += 32;
    while (f >= F) { f -= F; ++lr; }
  }
}

// ---- weight gradient, C_in = 1: dw[co][tap] += sum_p dy[p][co] x[p+tap]; dbias[co] += sum dy
// A workgroup owns `rows_per_block` consecutive (b,t) rows: their x rows (+1 halo row each side,
// zero halo columns) are staged in LDS once, so the only global stream is dy, read as float4 by
// thread = (4 output channels, one of 32 pixel lanes).  (The first version re-derived (b,t,f)
// with 64-bit divisions and issued 9 predicated global loads per pixel: 0.3 TB/s.)
template <typename TD>
__global__ __launch_bounds__(256) void conv3x3_c1_wgrad_kernel(const float* __restrict__ x,
                                                               const TD* __restrict__ dy,
                                                               float* __restrict__ dw,
                                                               float* __restrict__ dbias, int B, int T,
                                                               int F, int rows_per_block) {
  extern __shared__ __attribute__((aligned(16))) float xs[];  // [(rows_per_block + 2)][F + 2]
  __shared__ float sh[32][10][CH + 1];
  const int cg = threadIdx.x & 7, pl = threadIdx.x >> 3;
  const int BT = B * T, FP = F + 2;
  const int r0 = blockIdx.x * rows_per_block;
  const int nr = min(BT, r0 + rows_per_block) - r0;
  for (int idx = threadIdx.x; idx < (nr + 2) * FP; idx += blockDim.x) {
    const int gr = r0 + idx / FP - 1, fc = idx % FP - 1;
    xs[idx] = (gr >= 0 && gr < BT && fc >= 0 && fc < F) ? x[(long long)gr * F + fc] : 0.f;
  }
  __syncthreads();
  float acc[10][4];
#pragma unroll
  for (int i = 0; i < 10; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;
  const int t0 = r0 % T;
  int lr = pl / F, f = pl % F;
  const TD* dyb = dy + (long long)r0 * F * CH + cg * 4;
#pragma unroll 2
  for (int q = pl; q < nr * F; q += 32) {
    const float4 g = ld4<TD>(dyb + (long long)q * CH);
    int t = t0 + lr;
    if (t >= T) t -= T;
    // a halo row that belongs to the neighbouring utterance (or lies outside) counts as zero
    const float m_up = t > 0 ? 1.f : 0.f, m_dn = t < T - 1 ? 1.f : 0.f;
    const float* xc = xs + (lr + 1) * FP + (f + 1);
    acc[9][0] += g.x; acc[9][1] += g.y; acc[9][2] += g.z; acc[9][3] += g.w;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dt = tap / 3 - 1, df = tap % 3 - 1;
      float xv = xc[dt * FP + df];
      if (dt < 0) xv *= m_up;
      if (dt > 0) xv *= m_dn;
      acc[tap][0] += g.x * xv; acc[tap][1] += g.y * xv; acc[tap][2] += g.z * xv; acc[tap][3] += g.w * xv;
    }
    f += 32;
    while (f >= F) { f -= F; ++lr; }
  }
#pragma unroll
  for (int i = 0; i < 10; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) sh[pl][i][cg * 4 + e] = acc[i][e];
  __syncthreads();
  for (int i = threadIdx.x; i < 10 * CH; i += blockDim.x) {
    const int tap = i / CH, c = i % CH;
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 32; ++q) s += sh[q][tap][c];
    if (tap < 9) unsafeAtomicAdd(dw + c * 9 + tap, s);
    else if (dbias) unsafeAtomicAdd(dbias + c, s);
  }
}

// ---- weight gradient, 32 -> 32: thread = (ci, 4 co), 9x4 accumulators, persistent over tiles
__global__ __launch_bounds__(256) void conv3x3_c32_wgrad_kernel(const float* __restrict__ x,
                                                                const float* __restrict__ dy,
                                                                float* __restrict__ dw,
                                                                float* __restrict__ dbias, int B, int T,
                                                                int F, int tiles_f, int tiles_t) {
  __shared__ __attribute__((aligned(16))) float xs[HT * HF][CH];
  __shared__ __attribute__((aligned(16))) float ds[TT * TF][CH];
  const int tid = threadIdx.x;
  const int ci = tid & 31, cog = tid >> 5;
  float acc[9][4];
#pragma unroll
  for (int i = 0; i < 9; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  float accb[4] = {0.f, 0.f, 0.f, 0.f};
  const long long ntiles = (long long)B * tiles_t * tiles_f;
  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int tf = (int)(tile % tiles_f);
    const int tt = (int)((tile / tiles_f) % tiles_t);
    const long long b = tile / ((long long)tiles_f * tiles_t);
    const int f0 = tf * TF, t0 = tt * TT;
    __syncthreads();
    for (int idx = tid; idx < HT * HF * 8; idx += 256) {
      const int c4 = idx & 7, pix = idx >> 3;
      const int t = t0 + pix / HF - 1, f = f0 + pix % HF - 1;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (t >= 0 && t < T && f >= 0 && f < F)
        v = reinterpret_cast<const float4*>(x + ((b * T + t) * F + f) * CH)[c4];
      *reinterpret_cast<float4*>(&xs[pix][c4 * 4]) = v;
    }
    for (int idx = tid; idx < TT * TF * 8; idx += 256) {
      const int c4 = idx & 7, pix = idx >> 3;
      const int t = t0 + pix / TF, f = f0 + pix % TF;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (t < T && f < F) v = reinterpret_cast<const float4*>(dy + ((b * T + t) * F + f) * CH)[c4];
      *reinterpret_cast<float4*>(&ds[pix][c4 * 4]) = v;
    }
    __syncthreads();
    for (int p = 0; p < TT * TF; ++p) {
      const int pt = p / TF, pf = p % TF;
      const float4 g = *reinterpret_cast<const float4*>(&ds[p][cog * 4]);
      if (ci == 0) { accb[0] += g.x; accb[1] += g.y; accb[2] += g.z; accb[3] += g.w; }
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const float xv = xs[(pt + tap / 3) * HF + pf + tap % 3][ci];
        acc[tap][0] += g.x * xv; acc[tap][1] += g.y * xv;
        acc[tap][2] += g.z * xv; acc[tap][3] += g.w * xv;
      }
    }
  }
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      unsafeAtomicAdd(dw + ((long long)(cog * 4 + j) * 9 + tap) * CH + ci, acc[tap][j]);
  if (ci == 0 && dbias)
#pragma unroll
    for (int j = 0; j < 4; ++j) unsafeAtomicAdd(dbias + cog * 4 + j, accb[j]);
}

// ---- weight gradient, 32 -> 32, bf16 MFMA: dw[co][tap][ci] = sum_p dy[p][co] * x[p+tap][ci].
// The reduction index of the MFMA is the PIXEL, which is the strided index of both
// channels-last operands; ds_read_b64_tr_b16 turns 4 consecutive pixels x 16 channels of the LDS
// tiles into the 4 k-values a lane needs (lane mapping in gemm_bf16.hip).  Wave w owns the 32
// pixels of tile rows 2w, 2w+1 (one MFMA k-step) for all 9 taps: 36 MFMAs per wave per tile, 36
// accumulator fragments kept across the persistent tile loop, one cross-wave LDS reduction and
// one atomic flush per workgroup at the end.
template <typename TIO>
__global__ __launch_bounds__(256) void conv3x3_c32_wgrad_mfma_kernel(
    const TIO* __restrict__ x, const TIO* __restrict__ dy, float* __restrict__ dw,
    float* __restrict__ dbias, int B, int T, int F, int tiles_f, int tiles_t) {
  constexpr int PP = 80;  // bytes per pixel: 32 bf16 + 16 B pad
  constexpr int XS_B = HT * HF * PP, DS_B = TT * TF * PP;
  // two (x halo, dy tile) buffer pairs: the next tile is fetched into registers before the MFMAs
  // of the current one and committed to the other pair afterwards (one barrier per tile); the
  // final cross-wave reduction reuses the same memory
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (XS_B + DS_B)];
  float (*red)[16][68] = reinterpret_cast<float (*)[16][68]>(smem);   // [4][16][68] after the tile loop
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4, a4 = r >> 2, b4 = r & 3;
  typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
  f32x4 acc[9][2][2];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);  // this thread always stages channel quad tid&7
  const long long ntiles = (long long)B * tiles_t * tiles_f;
  constexpr bool IO16 = sizeof(TIO) == 2;
  constexpr int CPP = IO16 ? 4 : 8;     // chunks per pixel: 4 x 16 B (bf16 maps, copied verbatim) or 8 float4
  constexpr int NX = (HT * HF * CPP + 255) / 256, ND = (TT * TF * CPP) / 256;
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
  float4 px[IO16 ? 1 : NX], pd[IO16 ? 1 : ND];
  u32x4_t px16[IO16 ? NX : 1], pd16[IO16 ? ND : 1];
  auto fetch = [&](long long tl) {
    const int tf = (int)(tl % tiles_f);
    const int tt = (int)((tl / tiles_f) % tiles_t);
    const long long b = tl / ((long long)tiles_f * tiles_t);
    const int f0 = tf * TF, t0 = tt * TT;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int idx = tid + i * 256;
      const int cc = idx % CPP, pix = idx / CPP;
      const int t = t0 + pix / HF - 1, f = f0 + pix % HF - 1;
      const bool ok = idx < HT * HF * CPP && t >= 0 && t < T && f >= 0 && f < F;
      if constexpr (IO16) {
        px16[i] = u32x4_t{0u, 0u, 0u, 0u};
        if (ok) px16[i] = *reinterpret_cast<const u32x4_t*>(x + ((b * T + t) * F + f) * CH + cc * 8);
      } else {
        px[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) px[i] = ld4<TIO>(x + ((b * T + t) * F + f) * CH + cc * 4);
      }
    }
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      const int idx = tid + i * 256;
      const int cc = idx % CPP, pix = idx / CPP;
      const int t = t0 + pix / TF, f = f0 + pix % TF;
      const bool ok = t < T && f < F;
      if constexpr (IO16) {
        pd16[i] = u32x4_t{0u, 0u, 0u, 0u};
        if (ok) pd16[i] = *reinterpret_cast<const u32x4_t*>(dy + ((b * T + t) * F + f) * CH + cc * 8);
      } else {
        pd[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) pd[i] = ld4<TIO>(dy + ((b * T + t) * F + f) * CH + cc * 4);
      }
    }
  };
  // bias sums: a thread always stages the same channel group (cc = tid % CPP): 4 channels (fp32 maps, in
  // bsum) or 8 channels (bf16 maps, in bsum + bsum_hi)
  float4 bsum_hi = make_float4(0.f, 0.f, 0.f, 0.f);
  auto commit = [&](unsigned char* xs_, unsigned char* ds_) {
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int idx = tid + i * 256;
      if (idx < HT * HF * CPP) {
        if constexpr (IO16) {
          *reinterpret_cast<u32x4_t*>(xs_ + (idx / CPP) * PP + (idx % CPP) * 16) = px16[i];
        } else {
          bf16x4 h;
          h[0] = (__bf16)px[i].x; h[1] = (__bf16)px[i].y; h[2] = (__bf16)px[i].z; h[3] = (__bf16)px[i].w;
          *reinterpret_cast<bf16x4*>(xs_ + (idx >> 3) * PP + (idx & 7) * 8) = h;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      const int idx = tid + i * 256;
      if constexpr (IO16) {
        const u32x4_t q = pd16[i];
        bsum.x += __uint_as_float(q[0] << 16); bsum.y += __uint_as_float(q[0] & 0xffff0000u);
        bsum.z += __uint_as_float(q[1] << 16); bsum.w += __uint_as_float(q[1] & 0xffff0000u);
        bsum_hi.x += __uint_as_float(q[2] << 16); bsum_hi.y += __uint_as_float(q[2] & 0xffff0000u);
        bsum_hi.z += __uint_as_float(q[3] << 16); bsum_hi.w += __uint_as_float(q[3] & 0xffff0000u);
        *reinterpret_cast<u32x4_t*>(ds_ + (idx / CPP) * PP + (idx % CPP) * 16) = q;
      } else {
        bsum.x += pd[i].x; bsum.y += pd[i].y; bsum.z += pd[i].z; bsum.w += pd[i].w;
        bf16x4 h;
        h[0] = (__bf16)pd[i].x; h[1] = (__bf16)pd[i].y; h[2] = (__bf16)pd[i].z; h[3] = (__bf16)pd[i].w;
        *reinterpret_cast<bf16x4*>(ds_ + (idx >> 3) * PP + (idx & 7) * 8) = h;
      }
    }
  };
  int cur = 0;
  if ((long long)blockIdx.x < ntiles) {
    fetch(blockIdx.x);
    commit(smem, smem + XS_B);
  }
  __syncthreads();
  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const unsigned char* xs = smem + cur * (XS_B + DS_B);
    const unsigned char* ds = xs + XS_B;
    const bool has_next = tile + gridDim.x < ntiles;
    if (has_next) fetch(tile + gridDim.x);
    // this wave's 32 pixels: tile-local pixel k = 32*wave + 8g + a4 (+4)
    const int kloc = 8 * g + a4;                 // 0..31 within the wave's two tile rows
    const int trow = 2 * wave + (kloc >> 4), fcol = kloc & 15;
    bf16x8 dfrag[2];
#pragma unroll
    for (int cf = 0; cf < 2; ++cf) {
      const unsigned char* p = ds + (trow * TF + fcol) * PP + (cf * 16 + b4 * 4) * 2;
      const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)p);
      const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p + 4 * PP));
      bf16x8 o;
      o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3];
      o[4] = hi[0]; o[5] = hi[1]; o[6] = hi[2]; o[7] = hi[3];
      dfrag[cf] = o;
    }
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dt = tap / 3, df = tap % 3;
#pragma unroll
      for (int cif = 0; cif < 2; ++cif) {
        const unsigned char* p = xs + ((trow + dt) * HF + fcol + df) * PP + (cif * 16 + b4 * 4) * 2;
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)p);
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(p + 4 * PP));
        bf16x8 xf;
        xf[0] = lo[0]; xf[1] = lo[1]; xf[2] = lo[2]; xf[3] = lo[3];
        xf[4] = hi[0]; xf[5] = hi[1]; xf[6] = hi[2]; xf[7] = hi[3];
#pragma unroll
        for (int cf = 0; cf < 2; ++cf)
          acc[tap][cf][cif] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dfrag[cf], xf, acc[tap][cf][cif], 0, 0, 0);
      }
    }
    if (has_next) {
      unsigned char* nx = smem + (cur ^ 1) * (XS_B + DS_B);
      commit(nx, nx + XS_B);
    }
    __syncthreads();  // next pair complete, every wave done with the current one
    cur ^= 1;
  }
  // ---- reduce the 4 waves' fragments and flush: lane holds D[co = cf*16 + 4g + e][ci = cif*16 + r]
  for (int tap = 0; tap < 9; ++tap) {
    for (int cf = 0; cf < 2; ++cf) {
      __syncthreads();
#pragma unroll
      for (int cif = 0; cif < 2; ++cif)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[wave][4 * g + e][cif * 16 + r] = acc[tap][cf][cif][e];
      __syncthreads();
      for (int i = tid; i < 16 * 32; i += 256) {
        const int co = i >> 5, ci = i & 31;
        const float v = red[0][co][ci] + red[1][co][ci] + red[2][co][ci] + red[3][co][ci];
        unsafeAtomicAdd(dw + ((long long)(cf * 16 + co) * 9 + tap) * CH + ci, v);
      }
    }
  }
  if (dbias) {
    __syncthreads();
    float* rb = &red[0][0][0];  // [256][8]
    rb[tid * 8 + 0] = bsum.x; rb[tid * 8 + 1] = bsum.y; rb[tid * 8 + 2] = bsum.z; rb[tid * 8 + 3] = bsum.w;
    rb[tid * 8 + 4] = bsum_hi.x; rb[tid * 8 + 5] = bsum_hi.y; rb[tid * 8 + 6] = bsum_hi.z; rb[tid * 8 + 7] = bsum_hi.w;
    __syncthreads();
    if (tid < 32) {
      // channel tid: fp32 maps -> staged by threads with tid % 8 == tid / 4, element tid % 4;
      //              bf16 maps -> threads with tid % 4 == tid / 8, element tid % 8
      const int grp = IO16 ? (tid >> 3) : (tid >> 2), e = IO16 ? (tid & 7) : (tid & 3);
      float s2 = 0.f;
      for (int k = grp; k < 256; k += CPP) s2 += rb[k * 8 + e];
      unsafeAtomicAdd(dbias + tid, s2);
    }
  }
}

// ---- MaxPool2d(kernel=stride=(pt,pf), ceil_mode) on [B,T,F,C]
// One thread owns 16 B of input channels (8 bf16 / 4 fp32) of one output pixel; the arg-max is kept as ONE
// BYTE per element (position inside the window, dt * pf + df: the int32 absolute index of the first version
// was two thirds of the bytes this kernel wrote and a fifth of what backward read).
template <typename T> struct Vec16;
template <> struct Vec16<float> {
  static constexpr int N = 4;
  static __device__ __forceinline__ void load(const float* p, float (&v)[4]) {
    const float4 f = *reinterpret_cast<const float4*>(p);
    v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
  }
};
template <> struct Vec16<__bf16> {
  static constexpr int N = 8;
  static __device__ __forceinline__ void load(const __bf16* p, float (&v)[8]) {
    const bf16x8 h = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (float)h[e];
  }
};
template <typename T, int N> __device__ __forceinline__ void store_vec(T* p, const float (&v)[N]) {
  if constexpr (sizeof(T) == 2) {
    if constexpr (N == 8) {
      bf16x8 h;
#pragma unroll
      for (int e = 0; e < 8; ++e) h[e] = (__bf16)v[e];
      *reinterpret_cast<bf16x8*>(p) = h;
    } else {
      bf16x4 h;
#pragma unroll
      for (int e = 0; e < 4; ++e) h[e] = (__bf16)v[e];
      *reinterpret_cast<bf16x4*>(p) = h;
    }
  } else {
#pragma unroll
    for (int q = 0; q < N / 4; ++q)
      reinterpret_cast<float4*>(p)[q] = make_float4(v[q * 4], v[q * 4 + 1], v[q * 4 + 2], v[q * 4 + 3]);
  }
}
template <typename T, int N> __device__ __forceinline__ void load_vec(const T* p, float (&v)[N]) {
  if constexpr (sizeof(T) == 2) {
    if constexpr (N == 8) {
      const bf16x8 h = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (float)h[e];
    } else {
      const bf16x4 h = *reinterpret_cast<const bf16x4*>(p);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (float)h[e];
    }
  } else {
#pragma unroll
    for (int q = 0; q < N / 4; ++q) {
      const float4 f = reinterpret_cast<const float4*>(p)[q];
      v[q * 4] = f.x; v[q * 4 + 1] = f.y; v[q * 4 + 2] = f.z; v[q * 4 + 3] = f.w;
    }
  }
}

template <typename TX, typename TY>
__global__ __launch_bounds__(256) void maxpool2d_fwd_kernel(const TX* __restrict__ x, TY* __restrict__ y,
                                                            unsigned char* __restrict__ argmax, int B, int T, int F, int C,
                                                            int To, int Fo, int pt, int pf, int to_btcf) {
  // one workgroup per output row (b, to): 32-bit index arithmetic only (the flat 64-bit idx / % chains of
  // the first version were ~600 VALU instructions per 16 B moved -- the kernel was division-bound)
  constexpr int V = Vec16<TX>::N;
  const int CV = C / V;
  const int to = blockIdx.x % To;
  const long long b = blockIdx.x / To;
  const int row_elems = Fo * CV;
  for (int i = threadIdx.x; i < row_elems; i += blockDim.x) {
    const int fo = i / CV, cv = i - fo * CV;
    const long long idx = ((b * To + to) * Fo + fo) * CV + cv;
    float best[V];
    unsigned char bi[V];
#pragma unroll
    for (int e = 0; e < V; ++e) { best[e] = -FLT_MAX; bi[e] = 0; }
    bool first = true;
    for (int dt = 0; dt < pt; ++dt) {
      const int t = to * pt + dt;
      if (t >= T) break;
      for (int df = 0; df < pf; ++df) {
        const int f = fo * pf + df;
        if (f >= F) break;
        float vv[V];
        Vec16<TX>::load(x + ((b * T + t) * F + f) * C + cv * V, vv);
#pragma unroll
        for (int e = 0; e < V; ++e)
          if (first || vv[e] > best[e]) { best[e] = vv[e]; bi[e] = (unsigned char)(dt * pf + df); }
        first = false;
      }
    }
    if (to_btcf) {
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const long long o = ((b * To + to) * C + cv * V + e) * Fo + fo;
        y[o] = (TY)best[e];
        argmax[o] = bi[e];
      }
    } else {
      store_vec<TY, V>(y + idx * V, best);
      if constexpr (V == 8) {
        uint2 pk;
        pk.x = bi[0] | (bi[1] << 8) | (bi[2] << 16) | ((unsigned)bi[3] << 24);
        pk.y = bi[4] | (bi[5] << 8) | (bi[6] << 16) | ((unsigned)bi[7] << 24);
        *reinterpret_cast<uint2*>(argmax + idx * 8) = pk;
      } else {
        *reinterpret_cast<unsigned*>(argmax + idx * 4) = bi[0] | (bi[1] << 8) | (bi[2] << 16) | ((unsigned)bi[3] << 24);
      }
    }
  }
}

// TD: gradient type; TX: type of dx and relu_src.  One thread = 16 B of dx channels of one input pixel.
template <typename TD, typename TX>
__global__ __launch_bounds__(256) void maxpool2d_bwd_kernel(const TD* __restrict__ dy, const unsigned char* __restrict__ argmax,
                                                            TX* __restrict__ dx, int B, int T, int F, int C, int To,
                                                            int Fo, int pt, int pf, int from_btcf,
                                                            const TX* __restrict__ relu_src) {
  constexpr int V = Vec16<TX>::N;
  const int CV = C / V;
  const int t = blockIdx.x % T;            // one workgroup per input row (b, t), 32-bit arithmetic inside
  const long long b = blockIdx.x / T;
  const int row_elems = F * CV;
  for (int i = threadIdx.x; i < row_elems; i += blockDim.x) {
    const int f = i / CV, cv = i - f * CV;
    const long long idx = ((b * T + t) * F + f) * CV + cv;
    const int to = t / pt, fo = f / pf;
    const unsigned self = (unsigned)((t - to * pt) * pf + (f - fo * pf));
    float o[V];
    if (from_btcf) {
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const long long oi = ((b * To + to) * C + cv * V + e) * Fo + fo;
        o[e] = argmax[oi] == self ? (float)dy[oi] : 0.f;
      }
    } else {
      const long long oi = ((b * To + to) * Fo + fo) * C + cv * V;
      float g[V];
      load_vec<TD, V>(dy + oi, g);
      unsigned code[2];
      if constexpr (V == 8) {
        const uint2 pk = *reinterpret_cast<const uint2*>(argmax + oi);
        code[0] = pk.x; code[1] = pk.y;
      } else {
        code[0] = *reinterpret_cast<const unsigned*>(argmax + oi); code[1] = 0u;
      }
#pragma unroll
      for (int e = 0; e < V; ++e) o[e] = ((code[e >> 2] >> ((e & 3) * 8)) & 0xFFu) == self ? g[e] : 0.f;
    }
    if (relu_src) {  // fused ReLU backward of the layer that fed the pool (its output is relu_src)
      float m[V];
      Vec16<TX>::load(relu_src + idx * V, m);
#pragma unroll
      for (int e = 0; e < V; ++e) o[e] = m[e] > 0.f ? o[e] : 0.f;
    }
    store_vec<TX, V>(dx + idx * V, o);
  }
}

// ---- 2 x 2 windows, channels-last output (the two pools of the front-end at full map size): the generic
// kernels above run one dependent load -> compare -> (load ...) -> store chain per thread and per
// short-lived workgroup; measured 1.9 TB/s.  Here a thread owns UNR elements, issues ALL of their loads
// first (clamped, unconditional: an edge row / column is simply read twice, which cannot change a
// first-maximum) and only then compares and stores; workgroups cover RPB rows.
template <typename TX, typename TY, int UNR>
__global__ __launch_bounds__(256) void maxpool2d22_fwd_kernel(const TX* __restrict__ x, TY* __restrict__ y,
                                                              unsigned char* __restrict__ argmax, int BTo, int T, int F,
                                                              int To, int Fo, int cv_shift, int rpb) {
  constexpr int V = Vec16<TX>::N;
  const int CV = 1 << cv_shift, C = CV * V;
  const int row_elems = Fo * CV, nel = rpb * row_elems;
  const int g0 = blockIdx.x * rpb;
  for (int e0 = threadIdx.x; e0 < nel; e0 += 256 * UNR) {
    float v[UNR][4][V];
    long long oidx[UNR];
    bool ok[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int e = e0 + u * 256;
      const int row = e / row_elems, j = e - row * row_elems;
      const int gq = g0 + row;
      ok[u] = e < nel && gq < BTo;
      const int gcl = min(gq, BTo - 1);
      const int b = gcl / To, to = gcl - b * To;
      const int fo = j >> cv_shift, cv = j & (CV - 1);
      oidx[u] = (long long)gcl * row_elems + j;
      const int t0 = 2 * to, t1 = min(t0 + 1, T - 1), f0 = 2 * fo, f1 = min(f0 + 1, F - 1);
      const TX* xb = x + (long long)b * T * F * C + cv * V;
      Vec16<TX>::load(xb + ((long long)t0 * F + f0) * C, v[u][0]);
      Vec16<TX>::load(xb + ((long long)t0 * F + f1) * C, v[u][1]);
      Vec16<TX>::load(xb + ((long long)t1 * F + f0) * C, v[u][2]);
      Vec16<TX>::load(xb + ((long long)t1 * F + f1) * C, v[u][3]);
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      float best[V];
      unsigned code[2] = {0u, 0u};
#pragma unroll
      for (int q = 0; q < V; ++q) {
        float bv = v[u][0][q];
        unsigned bi = 0u;
#pragma unroll
        for (int w = 1; w < 4; ++w)
          if (v[u][w][q] > bv) { bv = v[u][w][q]; bi = (unsigned)w; }
        best[q] = bv;
        code[q >> 2] |= bi << ((q & 3) * 8);
      }
      if (ok[u]) {
        store_vec<TY, V>(y + oidx[u] * V, best);
        if constexpr (V == 8) *reinterpret_cast<uint2*>(argmax + oidx[u] * 8) = make_uint2(code[0], code[1]);
        else *reinterpret_cast<unsigned*>(argmax + oidx[u] * 4) = code[0];
      }
    }
  }
}

template <typename TD, typename TX, int UNR>
__global__ __launch_bounds__(256) void maxpool2d22_bwd_kernel(const TD* __restrict__ dy, const unsigned char* __restrict__ argmax,
                                                              TX* __restrict__ dx, int BT, int T, int F, int To, int Fo,
                                                              int cv_shift, int rpb, const TX* __restrict__ relu_src) {
  constexpr int V = Vec16<TX>::N;
  const int CV = 1 << cv_shift, C = CV * V;
  const int row_elems = F * CV, nel = rpb * row_elems;
  const int g0 = blockIdx.x * rpb;
  for (int e0 = threadIdx.x; e0 < nel; e0 += 256 * UNR) {
    float g[UNR][V], m[UNR][V];
    unsigned code[UNR][2], self[UNR];
    long long idx[UNR];
    bool ok[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int e = e0 + u * 256;
      const int row = e / row_elems, j = e - row * row_elems;
      const int gq = g0 + row;
      ok[u] = e < nel && gq < BT;
      const int gcl = min(gq, BT - 1);
      const int b = gcl / T, t = gcl - b * T;
      const int f = j >> cv_shift, cv = j & (CV - 1);
      idx[u] = (long long)gcl * row_elems + j;
      self[u] = (unsigned)(((t & 1) << 1) | (f & 1));
      const long long oi = (((long long)b * To + (t >> 1)) * Fo + (f >> 1)) * C + cv * V;
      load_vec<TD, V>(dy + oi, g[u]);
      if constexpr (V == 8) {
        const uint2 pk = *reinterpret_cast<const uint2*>(argmax + oi);
        code[u][0] = pk.x; code[u][1] = pk.y;
      } else {
        code[u][0] = *reinterpret_cast<const unsigned*>(argmax + oi); code[u][1] = 0u;
      }
      if (relu_src) Vec16<TX>::load(relu_src + idx[u] * V, m[u]);
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      float o[V];
#pragma unroll
      for (int q = 0; q < V; ++q) {
        o[q] = ((code[u][q >> 2] >> ((q & 3) * 8)) & 0xFFu) == self[u] ? g[u][q] : 0.f;
        if (relu_src) o[q] = m[u][q] > 0.f ? o[q] : 0.f;
      }
      if (ok[u]) store_vec<TX, V>(dx + idx[u] * V, o);
    }
  }
}

// ---- 2 x 2 windows with the [B,T',C,F'] output order of the LAST pool (conv.py:189 flattens it to C*F'):
// the transposition goes through LDS, so that global traffic is 16-B coalesced on both sides (the generic
// kernels gather / scatter single bf16 values per channel: 16 two-byte accesses per thread).
// One workgroup = one output row (b, to): C*Fo values (<= 4096).
template <typename TX, typename TY>
__global__ __launch_bounds__(256) void maxpool2d22_btcf_fwd_kernel(const TX* __restrict__ x, TY* __restrict__ y,
                                                                   unsigned char* __restrict__ argmax, int T, int F, int To,
                                                                   int Fo, int cv_shift) {
  constexpr int V = Vec16<TX>::N;
  const int CV = 1 << cv_shift, C = CV * V;
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
  TY* sy = reinterpret_cast<TY*>(sm);                       // [C][Fo]
  unsigned char* sa = sm + (size_t)C * Fo * sizeof(TY);    // [C][Fo]
  const int to = blockIdx.x % To;
  const long long b = blockIdx.x / To;
  const int t0 = 2 * to, t1 = min(t0 + 1, T - 1);
  const TX* xb = x + b * T * F * C;
  for (int i = threadIdx.x; i < Fo * CV; i += 256) {
    const int fo = i >> cv_shift, cv = i & (CV - 1);
    const int f0 = 2 * fo, f1 = min(f0 + 1, F - 1);
    float v[4][V];
    Vec16<TX>::load(xb + ((long long)t0 * F + f0) * C + cv * V, v[0]);
    Vec16<TX>::load(xb + ((long long)t0 * F + f1) * C + cv * V, v[1]);
    Vec16<TX>::load(xb + ((long long)t1 * F + f0) * C + cv * V, v[2]);
    Vec16<TX>::load(xb + ((long long)t1 * F + f1) * C + cv * V, v[3]);
#pragma unroll
    for (int q = 0; q < V; ++q) {
      float bv = v[0][q];
      unsigned bi = 0u;
#pragma unroll
      for (int w = 1; w < 4; ++w)
        if (v[w][q] > bv) { bv = v[w][q]; bi = (unsigned)w; }
      sy[(cv * V + q) * Fo + fo] = (TY)bv;
      sa[(cv * V + q) * Fo + fo] = (unsigned char)bi;
    }
  }
  __syncthreads();
  const long long o0 = (b * To + to) * C * Fo;              // the row is contiguous in [B,T',C,F']
  const int n = C * Fo;
  for (int i = threadIdx.x; i < n; i += 256) { y[o0 + i] = sy[i]; argmax[o0 + i] = sa[i]; }
}

template <typename TD, typename TX>
__global__ __launch_bounds__(256) void maxpool2d22_btcf_bwd_kernel(const TD* __restrict__ dy, const unsigned char* __restrict__ argmax,
                                                                   TX* __restrict__ dx, int T, int F, int To, int Fo,
                                                                   int cv_shift, const TX* __restrict__ relu_src) {
  constexpr int V = Vec16<TX>::N;
  const int CV = 1 << cv_shift, C = CV * V;
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
  TD* sg = reinterpret_cast<TD*>(sm);                       // [C][Fo]
  unsigned char* sa = sm + (size_t)C * Fo * sizeof(TD);
  const int to = blockIdx.x % To;
  const long long b = blockIdx.x / To;
  const long long o0 = (b * To + to) * C * Fo;
  const int n = C * Fo;
  for (int i = threadIdx.x; i < n; i += 256) { sg[i] = dy[o0 + i]; sa[i] = argmax[o0 + i]; }
  __syncthreads();
  const int rows = min(2, T - 2 * to);
  const int row_elems = F * CV;
  for (int i = threadIdx.x; i < rows * row_elems; i += 256) {
    const int dt = i >= row_elems ? 1 : 0;
    const int j = i - dt * row_elems;
    const int f = j >> cv_shift, cv = j & (CV - 1);
    const int fo = f >> 1;
    const unsigned self = (unsigned)((dt << 1) | (f & 1));
    const long long idx = ((b * T + 2 * to + dt) * F + f) * CV + cv;
    float m[V];
    if (relu_src) Vec16<TX>::load(relu_src + idx * V, m);
    float o[V];
#pragma unroll
    for (int q = 0; q < V; ++q) {
      const int k = (cv * V + q) * Fo + fo;
      o[q] = sa[k] == self ? (float)sg[k] : 0.f;
      if (relu_src) o[q] = m[q] > 0.f ? o[q] : 0.f;
    }
    store_vec<TX, V>(dx + idx * V, o);
  }
}

inline int ew_grid(long long n) {
  long long g = (n + 255) / 256;
  if (g > 256 * 16) g = 256 * 16;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

// io_dtype (NSP_DT_F32 / NSP_DT_BF16): element type of the 32-channel feature maps x (C_in = 32), y and
// mask_src; the single-channel input of the first layer is always fp32.  bf16 maps need NSP_COMPUTE_BF16.
extern "C" int nsp_conv2d3x3_fwd(const void* x, const float* w, const float* bias, void* y, int B,
                                 int T, int F, int Ci, int Co, int relu, const void* mask_src,
                                 int mode, int io_dtype, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (Co != CH) return NSP_EUNSUPPORTED;
  const bool io16 = io_dtype == NSP_DT_BF16;
  if (io16 && mode != NSP_COMPUTE_BF16) return NSP_EINVAL;
  if (Ci == 1) {
    if (mask_src) return NSP_EUNSUPPORTED;
    const int BT = B * T;
    int rpb = (BT + 2047) / 2048;                        // ~2048 workgroups ...
    const int cap = (int)((20 * 1024) / (sizeof(float) * (F + 2))) - 2;  // ... within 20 KB of staged rows
    if (rpb > cap) rpb = cap;
    if (rpb > T) rpb = T;
    if (rpb < 1) return NSP_EUNSUPPORTED;
    const size_t shmem = sizeof(float) * (size_t)(rpb + 2) * (F + 2);
    if (io16)
      hipLaunchKernelGGL(conv3x3_c1_kernel<__bf16>, dim3(nsp_cdiv(BT, rpb)), dim3(256), shmem, st, (const float*)x, w,
                         bias, (__bf16*)y, B, T, F, relu, rpb);
    else
      hipLaunchKernelGGL(conv3x3_c1_kernel<float>, dim3(nsp_cdiv(BT, rpb)), dim3(256), shmem, st, (const float*)x, w,
                         bias, (float*)y, B, T, F, relu, rpb);
  } else if (Ci == CH) {
    const int tiles_f = nsp_cdiv(F, TF), tiles_t = nsp_cdiv(T, TT);
    const long long ntiles = (long long)B * tiles_f * tiles_t;
    const int grid = (int)(ntiles < 1024 ? ntiles : 1024);  // 4 persistent workgroups per CU
    if (mode == NSP_COMPUTE_BF16) {
      const size_t sh = 2 * HT * HF * ConvCfg<0>::PIX_PITCH + 4 * 16 * 80;   // double-buffered halo + the waves' output slabs (bf16-map kernel)
      if (io16) {
        // buffer-addressed kernel: 32-bit byte offsets -> at most 4 GiB of map per launch (cut over the batch)
        const long long per_utt = (long long)T * F * CH * 2;
        if (per_utt >= 0xFFFFFFFFLL) return NSP_EUNSUPPORTED;
        const int bmax = (int)(0xFFFFFFF0LL / per_utt);
        for (int b0 = 0; b0 < B; b0 += bmax) {
          const int bn = B - b0 < bmax ? B - b0 : bmax;
          const long long nt = (long long)bn * tiles_f * tiles_t;
          const int gr = (int)(nt < 1024 ? nt : 1024);
          const __bf16* xb = (const __bf16*)x + (long long)b0 * T * F * CH;
          __bf16* yb = (__bf16*)y + (long long)b0 * T * F * CH;
          const __bf16* mb = mask_src ? (const __bf16*)mask_src + (long long)b0 * T * F * CH : nullptr;
          const unsigned bytes = (unsigned)(per_utt * bn);
          if (mask_src)
            hipLaunchKernelGGL((conv3x3_c32_b16_kernel<true>), dim3(gr), dim3(256), sh, st, xb, w, bias, yb, bn, T, F, relu,
                               mb, tiles_f, tiles_t, bytes);
          else
            hipLaunchKernelGGL((conv3x3_c32_b16_kernel<false>), dim3(gr), dim3(256), sh, st, xb, w, bias, yb, bn, T, F, relu,
                               mb, tiles_f, tiles_t, bytes);
        }
      } else
        hipLaunchKernelGGL((conv3x3_c32_kernel<0, float>), dim3(grid), dim3(256), sh, st, (const float*)x, w, bias,
                           (float*)y, B, T, F, relu, (const float*)mask_src, tiles_f, tiles_t);
    } else {
      const size_t sh = HT * HF * ConvCfg<1>::PIX_PITCH + sizeof(float) * CH * W_PITCH;
      hipFuncSetAttribute((const void*)conv3x3_c32_kernel<1, float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
      hipLaunchKernelGGL((conv3x3_c32_kernel<1, float>), dim3(grid), dim3(256), sh, st, (const float*)x, w, bias,
                         (float*)y, B, T, F, relu, (const float*)mask_src, tiles_f, tiles_t);
    }
  } else {
    return NSP_EUNSUPPORTED;
  }
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

// dw / dbias must be zeroed by the caller (atomic accumulation); io_dtype: type of dy and (C_in = 32) of x
extern "C" int nsp_conv2d3x3_wgrad(const void* x, const void* dy, float* dw, float* dbias, int B,
                                   int T, int F, int Ci, int Co, int mode, int io_dtype, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (Co != CH) return NSP_EUNSUPPORTED;
  const bool io16 = io_dtype == NSP_DT_BF16;
  if (io16 && mode != NSP_COMPUTE_BF16) return NSP_EINVAL;
  if (Ci == 1) {
    const int BT = B * T;
    int rpb = (BT + 1023) / 1024;                       // ~1024 workgroups ...
    const int cap = (int)((20 * 1024) / (sizeof(float) * (F + 2))) - 2;  // ... within 20 KB of staged rows
    if (rpb > cap) rpb = cap;
    if (rpb > T) rpb = T;
    if (rpb < 1) return NSP_EUNSUPPORTED;
    const size_t shmem = sizeof(float) * (size_t)(rpb + 2) * (F + 2);
    if (io16)
      hipLaunchKernelGGL(conv3x3_c1_wgrad_kernel<__bf16>, dim3(nsp_cdiv(BT, rpb)), dim3(256), shmem, st, (const float*)x,
                         (const __bf16*)dy, dw, dbias, B, T, F, rpb);
    else
      hipLaunchKernelGGL(conv3x3_c1_wgrad_kernel<float>, dim3(nsp_cdiv(BT, rpb)), dim3(256), shmem, st, (const float*)x,
                         (const float*)dy, dw, dbias, B, T, F, rpb);
  } else if (Ci == CH) {
    const int tiles_f = nsp_cdiv(F, TF), tiles_t = nsp_cdiv(T, TT);
    long long ntiles = (long long)B * tiles_f * tiles_t;
    if (mode == NSP_COMPUTE_BF16) {
      int blocks = ntiles < 512 ? (int)ntiles : 512;
      if (io16)
        hipLaunchKernelGGL(conv3x3_c32_wgrad_mfma_kernel<__bf16>, dim3(blocks), dim3(256), 0, st, (const __bf16*)x,
                           (const __bf16*)dy, dw, dbias, B, T, F, tiles_f, tiles_t);
      else
        hipLaunchKernelGGL(conv3x3_c32_wgrad_mfma_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)x,
                           (const float*)dy, dw, dbias, B, T, F, tiles_f, tiles_t);
    } else {
      int blocks = ntiles < 1024 ? (int)ntiles : 1024;
      hipLaunchKernelGGL(conv3x3_c32_wgrad_kernel, dim3(blocks), dim3(256), 0, st, (const float*)x, (const float*)dy, dw,
                         dbias, B, T, F, tiles_f, tiles_t);
    }
  } else {
    return NSP_EUNSUPPORTED;
  }
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

extern "C" int nsp_maxpool2d_fwd(const void* x, void* y, unsigned char* argmax, int B, int T, int F, int C,
                                 int pt, int pf, int to_btcf, int x_dtype, int y_dtype, void* stream) {
  if (C % 4) return NSP_EUNSUPPORTED;
  const int To = (T + pt - 1) / pt, Fo = (F + pf - 1) / pf;
  const int vec = x_dtype == NSP_DT_BF16 ? 8 : 4;
  if (C % vec || pt * pf > 256 || pt < 1 || pf < 1) return NSP_EUNSUPPORTED;
  if ((long long)B * To > 0x7fffffffLL) return NSP_EUNSUPPORTED;
  const int cvn = C / vec;
  if (pt == 2 && pf == 2 && !to_btcf && (cvn & (cvn - 1)) == 0 && x_dtype == y_dtype) {
    int sh = 0;
    while ((1 << sh) < cvn) ++sh;
    const int row_elems = Fo * cvn;
    int rpb = 2048 / row_elems;          // ~2048 elements = 4 per thread x 2 rounds of UNR = 2 ... per workgroup
    if (rpb < 1) rpb = 1;
    const dim3 g2((B * To + rpb - 1) / rpb);
    hipStream_t s2 = (hipStream_t)stream;
    if (x_dtype == NSP_DT_BF16)
      hipLaunchKernelGGL((maxpool2d22_fwd_kernel<__bf16, __bf16, 2>), g2, dim3(256), 0, s2, (const __bf16*)x, (__bf16*)y, argmax,
                         B * To, T, F, To, Fo, sh, rpb);
    else
      hipLaunchKernelGGL((maxpool2d22_fwd_kernel<float, float, 2>), g2, dim3(256), 0, s2, (const float*)x, (float*)y, argmax,
                         B * To, T, F, To, Fo, sh, rpb);
    NSP_LAUNCH_CHECK();
    return NSP_OK;
  }
  if (pt == 2 && pf == 2 && to_btcf && (cvn & (cvn - 1)) == 0 && x_dtype == y_dtype && C * Fo <= 8192) {
    int sh = 0;
    while ((1 << sh) < cvn) ++sh;
    hipStream_t s2 = (hipStream_t)stream;
    if (x_dtype == NSP_DT_BF16)
      hipLaunchKernelGGL((maxpool2d22_btcf_fwd_kernel<__bf16, __bf16>), dim3(B * To), dim3(256), (size_t)C * Fo * 3, s2,
                         (const __bf16*)x, (__bf16*)y, argmax, T, F, To, Fo, sh);
    else
      hipLaunchKernelGGL((maxpool2d22_btcf_fwd_kernel<float, float>), dim3(B * To), dim3(256), (size_t)C * Fo * 5, s2,
                         (const float*)x, (float*)y, argmax, T, F, To, Fo, sh);
    NSP_LAUNCH_CHECK();
    return NSP_OK;
  }
  const dim3 grid(B * To);
  const dim3 blk(Fo * (C / vec) > 128 ? 256 : 128);
  hipStream_t st = (hipStream_t)stream;
#define MPF(TX, TY) hipLaunchKernelGGL((maxpool2d_fwd_kernel<TX, TY>), grid, blk, 0, st, (const TX*)x, (TY*)y, \
                                       argmax, B, T, F, C, To, Fo, pt, pf, to_btcf)
  if (x_dtype == NSP_DT_BF16 && y_dtype == NSP_DT_BF16) MPF(__bf16, __bf16);
  else if (x_dtype == NSP_DT_BF16) MPF(__bf16, float);
  else if (y_dtype == NSP_DT_BF16) MPF(float, __bf16);
  else MPF(float, float);
#undef MPF
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}

// dy_dtype: type of the incoming gradient; dx_dtype: type of dx AND of relu_src
extern "C" int nsp_maxpool2d_bwd(const void* dy, const unsigned char* argmax, void* dx, int B, int T, int F,
                                 int C, int pt, int pf, int from_btcf, const void* relu_src,
                                 int dy_dtype, int dx_dtype, void* stream) {
  if (C % 4) return NSP_EUNSUPPORTED;
  const int To = (T + pt - 1) / pt, Fo = (F + pf - 1) / pf;
  const int vec = dx_dtype == NSP_DT_BF16 ? 8 : 4;
  if (C % vec || pt * pf > 256 || pt < 1 || pf < 1) return NSP_EUNSUPPORTED;
  if ((long long)B * T > 0x7fffffffLL) return NSP_EUNSUPPORTED;
  const int cvn = C / vec;
  if (pt == 2 && pf == 2 && !from_btcf && (cvn & (cvn - 1)) == 0 && dy_dtype == dx_dtype) {
    int sh = 0;
    while ((1 << sh) < cvn) ++sh;
    const int row_elems = F * cvn;
    int rpb = 4096 / row_elems;
    if (rpb < 1) rpb = 1;
    const dim3 g2((B * T + rpb - 1) / rpb);
    hipStream_t s2 = (hipStream_t)stream;
    if (dx_dtype == NSP_DT_BF16)
      hipLaunchKernelGGL((maxpool2d22_bwd_kernel<__bf16, __bf16, 4>), g2, dim3(256), 0, s2, (const __bf16*)dy, argmax, (__bf16*)dx,
                         B * T, T, F, To, Fo, sh, rpb, (const __bf16*)relu_src);
    else
      hipLaunchKernelGGL((maxpool2d22_bwd_kernel<float, float, 4>), g2, dim3(256), 0, s2, (const float*)dy, argmax, (float*)dx,
                         B * T, T, F, To, Fo, sh, rpb, (const float*)relu_src);
    NSP_LAUNCH_CHECK();
    return NSP_OK;
  }
  if (pt == 2 && pf == 2 && from_btcf && (cvn & (cvn - 1)) == 0 && dy_dtype == dx_dtype && C * Fo <= 8192) {
    int sh = 0;
    while ((1 << sh) < cvn) ++sh;
    hipStream_t s2 = (hipStream_t)stream;
    if (dx_dtype == NSP_DT_BF16)
      hipLaunchKernelGGL((maxpool2d22_btcf_bwd_kernel<__bf16, __bf16>), dim3(B * To), dim3(256), (size_t)C * Fo * 3, s2,
                         (const __bf16*)dy, argmax, (__bf16*)dx, T, F, To, Fo, sh, (const __bf16*)relu_src);
    else
      hipLaunchKernelGGL((maxpool2d22_btcf_bwd_kernel<float, float>), dim3(B * To), dim3(256), (size_t)C * Fo * 5, s2,
                         (const float*)dy, argmax, (float*)dx, T, F, To, Fo, sh, (const float*)relu_src);
    NSP_LAUNCH_CHECK();
    return NSP_OK;
  }
  const dim3 grid(B * T);
  const dim3 blk(F * (C / vec) > 128 ? 256 : 128);
  hipStream_t st = (hipStream_t)stream;
#define MPB(TD, TX) hipLaunchKernelGGL((maxpool2d_bwd_kernel<TD, TX>), grid, blk, 0, st, (const TD*)dy, argmax, \
                                       (TX*)dx, B, T, F, C, To, Fo, pt, pf, from_btcf, (const TX*)relu_src)
  if (dy_dtype == NSP_DT_BF16 && dx_dtype == NSP_DT_BF16) MPB(__bf16, __bf16);
  else if (dy_dtype == NSP_DT_BF16) MPB(__bf16, float);
  else if (dx_dtype == NSP_DT_BF16) MPB(float, __bf16);
  else MPB(float, float);
#undef MPB
  NSP_LAUNCH_CHECK();
  return NSP_OK;
}
