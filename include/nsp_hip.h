/*
 * nsp_hip.h -- C ABI of libnsp_hip.so: the MI355X (gfx950) kernels behind the
 * neural_sp Speech2Text training hot path.
 *
 * The reference (hirofumi0810/neural_sp) has no FFI of its own: the hot path is
 * plain torch ops plus three un-vendored loss libraries.  Every entry point
 * below names the reference call site(s) it replaces (paths relative to the
 * reference root).  The Python host side (neural_sp_amd/ops.py) binds these with
 * ctypes; INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success, a hipError_t (>0) on a HIP failure,
 *     or a negative NSP_E* code for invalid arguments;
 *   - pointers are device pointers unless a comment says "host";
 *   - `stream` is a hipStream_t passed as void*; nothing here synchronises,
 *     allocates or frees: workspaces are passed in by the caller (one exception, round 6: nsp_gemm* keeps a 64-MB
 *     stream-K exchange workspace per stream, hipMalloc'ed on the first launch that uses it; NSP_GEMM_8P_STREAMK=0 avoids it);
 *   - tensors are dense fp32 row-major unless stated; lengths are int32;
 *   - `mode`: NSP_COMPUTE_BF16 = bf16 MFMA operands / fp32 accumulate,
 *             NSP_COMPUTE_F32  = exact fp32 MFMA (v_mfma_f32_16x16x4_f32).
 */
#ifndef NSP_HIP_H
#define NSP_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define NSP_OK 0
#define NSP_EINVAL (-1)
#define NSP_EUNSUPPORTED (-2)

#define NSP_COMPUTE_BF16 0
#define NSP_COMPUTE_F32 1

#define NSP_DT_F32 0
#define NSP_DT_BF16 1

/* activations understood by the GEMM epilogue and the elementwise kernels */
#define NSP_ACT_NONE 0
#define NSP_ACT_RELU 1
#define NSP_ACT_SWISH 2
#define NSP_ACT_TANH 3
#define NSP_ACT_GELU 4 /* 0.5x(1+erf(x/sqrt2)), reference modules/gelu.py gelu_accurate=tanh form is 5 */
#define NSP_ACT_GELU_TANH 5
#define NSP_ACT_TANH_OUT 6 /* derivative only (dact): source holds y = tanh(x), factor 1 - y^2 */

int nsp_version(void);

/* ------------------------------------------------------------------------ *
 * Batched strided GEMM with fused epilogue.                                *
 * Replaces every nn.Linear / 1x1 Conv1d / einsum contraction on the path:  *
 *   positionwise_feed_forward.py:89, relative_multihead_attention.py:169-  *
 *   193,215,217, multihead_attention.py:124-153, conformer_convolution.py: *
 *   108,127, conv.py:192-193 (bridge), ctc.py:124, rnn_transducer.py:272-  *
 *   275, and their autograd backward (dgrad / wgrad).                      *
 *                                                                          *
 *   C[z][m][n] = epi( sum_k A[z](m,k) * B[z](k,n) )                        *
 *   A(m,k) at A + z1*a_b1 + z2*a_b2 + m*a_rs + k*a_cs                       *
 *   B(k,n) at B + z1*b_b1 + z2*b_b2 + k*b_ks + n*b_ns                       *
 *   C(m,n) at C + z1*c_b1 + z2*c_b2 + m*ldc + n                             *
 *   epi(v): v += bias[n]; if(pre_out) pre_out(m,n)=v; v = act(v);          *
 *           if(dact_src) v *= act'(dact_src(m,n)) (dact selects act');     *
 *           v *= alpha; if(res) v += res(m,n);                             *
 *           if(mask_keep) v *= mask(m,n) (0 or 1/(1-p) dropout scale)      *
 *   splitk>1: partial sums are atomically added into C (caller zeroes C;   *
 *             only alpha is applied).                                      *
 * ------------------------------------------------------------------------ */
typedef struct {
  int M, N, K;
  const void* A; long long a_rs, a_cs;   /* float* or bf16 (uint16) storage, see a_dtype */
  const void* B; long long b_ks, b_ns;
  void* C; long long ldc;
  int batch1, batch2;
  long long a_b1, a_b2, b_b1, b_b2, c_b1, c_b2;
  const float* bias;      /* [N] or NULL */
  int act;                /* NSP_ACT_* applied after bias */
  void* pre_out;          /* optional: value before activation, layout of C */
  const void* dact_src;   /* optional: multiply by act'(dact_src), layout of C */
  int dact;               /* NSP_ACT_* of the derivative */
  const float* res;       /* optional residual, layout of C (may alias C) */
  float alpha;
  int splitk;             /* >=1 */
  int mode;               /* NSP_COMPUTE_* */
  float dropout_p;        /* 0 => off; else philox-style keep mask regenerated from seed */
  unsigned long long seed, offset;
  /* element types: NSP_DT_F32 (0) or NSP_DT_BF16 (1).  A and B must agree.  With bf16
   * operands (NSP_COMPUTE_BF16 only) the strides are in bf16 elements, the contiguous
   * index must have unit stride, leading dims / K (KC) resp. the row extent (RC) must be
   * multiples of 8 and bases 16-B aligned. */
  int a_dtype, b_dtype, c_dtype, pre_dtype, dact_dtype;
  /* split-K without atomics: if c_ss != 0 split s stores its alpha-scaled partial tile at
   * C + s*c_ss (fp32); the caller sums the splitk slabs with nsp_splitk_reduce. */
  long long c_ss;
  /* RNN-T joint epilogues: set only by nsp_rnnt_joint_gemm (NSP_EPI_NONE = the epilogue above).
   * N is the PADDED vocabulary width (multiple of 64), epi_ncols the real one.
   *   NSP_EPI_RNNT_LSE    : nothing is stored to C.  Per row and 64-column block: (max, sum exp(.-max))
   *                         -> epi_f0 [M, N/64, 2]; logit at column epi_blank -> epi_f1 [M]; logit at
   *                         column epi_lab[m] (>= 0) -> epi_f2 [M].
   *   NSP_EPI_RNNT_DLOGITS: C (bf16 [M, ldc = N]) <- -(gb+gl) exp(logit - lse) + gb [n=blank]
   *                         + gl [n=lab] with one record per row {lse, gb * scale, gl * scale, bits(lab)}
   *                         in epi_f0 (float4 [M]; nsp_rnnt_joint_gemm packs it from its f0/f1/f2), zeros
   *                         in the pad columns; column sums of every 64-row block -> epi_f3
   *                         [ceil(M/128)*2, N] (output-bias gradient slabs, no atomics).
   *   scale = epi_scale * (epi_scale_dev ? *epi_scale_dev : 1).                                  */
  /* epi_mode == NSP_EPI_NONE and epi_f3 != NULL: column-sum slabs of the stored values (see nsp_gemm_flat) */
  int epi_mode, epi_ncols, epi_blank;
  const int* epi_lab;
  float* epi_f0; float* epi_f1; float* epi_f2; float* epi_f3;
  const float* epi_scale_dev;
  float epi_scale;
} nsp_gemm_params;
#define NSP_EPI_NONE 0
#define NSP_EPI_RNNT_LSE 1
#define NSP_EPI_RNNT_DLOGITS 2

/* reduction splits for the weight gradient dW[N, K] = dY[rows, N]^T X[rows, K] with bf16 operands (slab mode, c_ss),
 * sized for the 256 x 256 kernel (tiles x splits ~ one workgroup per CU); 0 = shape not eligible, caller's rule */
int nsp_wgrad_splitk(long long N, long long K, long long rows);
/* refresh many bf16 weight shadows in ONE launch.  table (device): n_entries rows of 8 x int64
 * {src fp32 pointer, dst bf16 pointer, rows, cols, src_ld, dst_ld, transpose, first tile}: dst(r,c) = transpose ?
 * src[c*src_ld + r] : src[r*src_ld + c] for r < rows, c < cols; entries sorted by first tile, an entry owns
 * ceil(rows/64) * ceil(cols/64) tiles; total_tiles = their sum.  Pad elements of dst are not touched. */
int nsp_shadow_refresh(const long long* table, int n_entries, int total_tiles, void* stream);
/* out[i] = sum_s part[s*n + i] (i < n): the deterministic reduction of split-K slabs */
int nsp_splitk_reduce(const float* part, float* out, int splits, long long n, void* stream);
/* TEST HOOK (tests/test_kernels_conv_loss_gpu.py): n_wg workgroups of 1024 threads that do nothing but hold their CU for
 * `cycles` shader cycles -- what a resident collective (an RCCL channel set) or any long kernel of another stream does to
 * the grid-barrier LSTM: it has to become resident beside them, later but correctly. */
int nsp_debug_occupy(int n_wg, long long cycles, void* stream);
/* TEST HOOK (tests/test_hostile_neighbour_gpu.py): n_wg short-lived workgroups of 256 threads that fill lds_bytes (<= 64 KB,
 * multiple of 4) of LDS `rounds` times and ~96 VGPRs + ~96 AGPRs per lane with `pattern`, then exit.  Run on another stream
 * (or by another process) beside the step, they are what a second rank on the same device, an RCCL kernel or any other
 * tenant is to the step's kernels: every LDS byte and register a kernel reads before writing then holds `pattern` (e.g. a
 * NaN) instead of what the previous kernel of the same stream left there, and LDS / issue slots are contended. */
int nsp_debug_scribble(int n_wg, int lds_bytes, unsigned int pattern, int rounds, void* stream);

int nsp_gemm(const nsp_gemm_params* p, void* stream);
/* same call with the struct fields as positional arguments (cheaper to marshal from ctypes) */
int nsp_gemm_flat(int M, int N, int K, const void* A, long long a_rs, long long a_cs,
                  const void* B, long long b_ks, long long b_ns, void* C, long long ldc,
                  int batch1, int batch2, long long a_b1, long long a_b2, long long b_b1,
                  long long b_b2, long long c_b1, long long c_b2, const float* bias, int act,
                  void* pre_out, const void* dact_src, int dact, const float* res,
                  float alpha, int splitk, int mode, float dropout_p,
                  unsigned long long seed, unsigned long long offset, int a_dtype,
                  int b_dtype, int c_dtype, int pre_dtype, int dact_dtype, long long c_ss,
                  float* colsum_slabs, void* stream);
/* the same call, its 38 arguments (everything but `stream`, in this order) packed into 38 8-byte little-endian slots:
 * integers and pointers as int64, alpha and dropout_p as double -- one struct.pack on the Python side */
int nsp_gemm_packed(const void* packed, void* stream);
/* colsum_slabs (bf16 operands, splitk == 1; may be NULL): fp32 [4 * ceil(M/128), N] (the tile grid overhangs M),
 * ZERO-INITIALISED by the caller.
 * Row (m / R) receives the column sums of the stored values of rows m .. m+R-1 for every R-row block a wave
 * owns (R = 64, or 32 on the 64-row-tile variant; untouched rows stay zero): summing the rows gives
 * sum_m C(m, n) -- the bias gradient when C is a d(pre-activation) -- without a separate pass over C. */

/* ------------------------------------------------------------------------ *
 * LayerNorm over the last dim (eps inside sqrt, biased variance).          *
 * Replaces nn.LayerNorm at conformer_block.py:132,140,160,177,180,         *
 * transformer_block.py:109,132, transformer.py:600,                        *
 * conformer_convolution.py:64,119.                                         *
 *   y = (x-mean)*rstd*gamma + beta ; optional fused activation on y.       *
 *   mean/rstd [rows] are saved for backward.  y16 (optional) receives a    *
 *   bf16 copy of y for the consuming MFMA GEMM; y itself may be NULL then. *
 * ------------------------------------------------------------------------ */
int nsp_layernorm_fwd(const float* x, const float* gamma, const float* beta,
                      float* y, float* mean, float* rstd,
                      int rows, int d, float eps, int act, float* y_pre,
                      void* y16, void* stream);
/* dx (may alias dy), dgamma/dbeta accumulated via atomics into zeroed buffers.
 * If act != NONE, y_pre is the pre-activation LN output saved by fwd.
 * dres (optional, [rows,d]) is added to dx: the gradient that reaches x through the residual
 * connection around the normalised branch (x feeds both LN and "+ x"), so that the two are summed
 * in this pass instead of by a separate elementwise kernel. */
int nsp_layernorm_bwd(const float* dy, const float* x, const float* gamma,
                      const float* mean, const float* rstd, const float* y_pre,
                      const float* dres, float* dx, float* dgamma, float* dbeta,
                      int rows, int d, int act, void* stream);
/* the same with the pre-activation RECOMPUTED as xhat * gamma + beta from the operands the pass reads anyway: forward
 * then needs to store neither y_pre nor (when only a GEMM consumes the bf16 image) the fp32 output
 * (conformer_convolution.py:119-124 in throughput mode) */
/* act = none, plus the "prepared" image of dx for the backward of the Linear whose output x is: g16 bf16 [rows, d] =
 * g_alpha * dx * dropout_keep(g_seed, g_offset + element index, g_p) (the mask that Linear's epilogue applied in
 * forward) and its column sums ACCUMULATED into gsum [d] (fp32, zeroed by the caller).  d % 8 == 0. */
int nsp_layernorm_bwd_prep(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                           const float* dres, float* dx, float* dgamma, float* dbeta, void* g16, float* gsum,
                           float g_alpha, float g_p, unsigned long long g_seed, unsigned long long g_offset,
                           int rows, int d, void* stream);
/* LayerNorm pair at a Conformer block boundary (conformer_block.py:176-180 + :132-133 of the next block): y = LN_a(x)
 * (fp32: the residual stream) and z16 = bf16(LN_b(y)) (the operand of the next block's first feed-forward GEMM) in one
 * pass; backward: dx = LN_a'(LN_b'(dz) + dres) with y recomputed from x; the four parameter gradients and gsum are
 * ACCUMULATED into caller-zeroed buffers; g16 / gsum as in nsp_layernorm_bwd_prep (both or neither). */
int nsp_layernorm_pair_fwd(const float* x, const float* gamma_a, const float* beta_a, float eps_a,
                           const float* gamma_b, const float* beta_b, float eps_b, float* y, void* z16,
                           float* mean_a, float* rstd_a, float* mean_b, float* rstd_b, int rows, int d, void* stream);
int nsp_layernorm_pair_bwd(const float* dz, const float* dres, const float* x, const float* gamma_a,
                           const float* beta_a, const float* mean_a, const float* rstd_a, const float* gamma_b,
                           const float* mean_b, const float* rstd_b, float* dx, float* dgamma_a, float* dbeta_a,
                           float* dgamma_b, float* dbeta_b, void* g16, float* gsum, float g_alpha, float g_p,
                           unsigned long long g_seed, unsigned long long g_offset, int rows, int d, void* stream);
int nsp_layernorm_bwd_recompute(const float* dy, const float* x, const float* gamma, const float* beta,
                                const float* mean, const float* rstd, const float* dres, float* dx,
                                float* dgamma, float* dbeta, int rows, int d, int act, void* stream);

/* ------------------------------------------------------------------------ *
 * Elementwise helpers (vectorised, grid-stride).                           *
 * ------------------------------------------------------------------------ */
/* fp32 -> bf16 (round-to-nearest-even) copy of a [rows, cols] matrix (row stride ld_in) into a
 * bf16 image with row pitch ld_out >= roundup8(cols); columns cols..roundup8(cols)-1 are
 * zero filled (columns beyond that are not touched, so the destination may be a column
 * block of a wider matrix): the bf16 shadow copies of weights and activations that feed the
 * MFMA GEMM.  out is uint16 storage, 16-B aligned. */
int nsp_cast_bf16(const float* x, void* out, long long rows, int cols, long long ld_in,
                  long long ld_out, void* stream);
/* y = alpha*x + beta*z (z may be NULL) */
int nsp_axpby(const float* x, const float* z, float* y, float alpha, float beta,
              long long n, void* stream);
/* y = act(x) ; out = alpha * dy * act'(pre) */
int nsp_act_fwd(const float* x, float* y, int act, long long n, void* stream);
int nsp_dact_mul(const float* dy, const float* pre, float* out, int act, float alpha,
                 long long n, void* stream);
/* y[i] = alpha * x[i] * keep(seed, offset+i) / (1-p): dropout forward; applied to dy with
 * the same (seed, offset) it is the backward.  Same counter-based mask as the GEMM /
 * softmax epilogues (nn.Dropout sites: conformer_block.py:134,154,171,179 ...). */
int nsp_dropout(const float* x, float* y, float p, float alpha, unsigned long long seed,
                unsigned long long offset, long long n, void* stream);
/* y[i] = alpha*x[i] + z[i % period]: additive sinusoidal encoding broadcast over the batch
 * (positional_embedding.py:80-88) */
int nsp_scale_add_bcast(const float* x, const float* z, float* y, float alpha,
                        long long n, long long period, void* stream);
/* relative position table out[L][d] = [sin(-(j+1)w_i), cos(-(j+1)w_i)]
 * (positional_embedding.py:131-138) */
int nsp_xl_pos_table(const float* inv_freq, float* out, int L, int d, void* stream);
/* column sum: out[n] (+)= sum_m x[m*ld + n]   (bias gradients) */
int nsp_colsum(const float* x, float* out, int rows, int cols, long long ld,
               int accumulate, void* stream);
int nsp_colsum_bf16(const void* x /*bf16*/, float* out, int rows, int cols, long long ld,
                    int accumulate, void* stream);
/* out = alpha * dy * dropout_keep(seed, offset+i)/(1-p) * act'(pre): turns the gradient that
 * arrives at a Linear's fused epilogue into the dgrad/wgrad operand in ONE pass
 * (pre may be NULL; pre/out are fp32 or bf16; n % 4 == 0) */
/* colsum (may be NULL; needs cols % 4 == 0, cols <= 4096): slabs fp32 [nsp_grad_prep_slabs(n / cols), cols];
 * slab row k receives the column sums of `out` over the k-th band of rows, so the bias gradient is the sum
 * of the slab rows -- no second pass over the [rows, cols] image, no atomics. */
int nsp_grad_prep_slabs(int rows);
int nsp_grad_prep(const float* dy, const void* pre, int pre_bf16, void* out, int out_bf16,
                  int act, float alpha, float p, unsigned long long seed,
                  unsigned long long offset, long long n, int cols, float* colsum, void* stream);
/* GLU over the channel dim of [rows, 2C] -> [rows, C]: a*sigmoid(b)
 * (conformer_convolution.py:109, F.glu(dim=1) on [B,2C,T]) */
int nsp_glu_fwd(const float* x, float* y, long long rows, int C, void* stream);
int nsp_glu_bwd(const float* x, const float* dy, float* dx, long long rows, int C,
                void* stream);
/* GLU on the bf16 image the pointwise conv's GEMM epilogue writes (throughput mode; conformer_convolution.py:107-112):
 * forward y fp32 [rows, C] from x16 bf16 [rows, 2C]; backward dx16 bf16 [rows, 2C] (the operand of the gradient GEMMs) and
 * its column sums as slabs [nsp_grad_prep_slabs(rows), 2C] fp32, every row written (bias gradient).  C % 8 == 0;
 * backward: C / 4 a divisor of 256. */
int nsp_glu_fwd_b16(const void* x16, float* y, long long rows, int C, void* stream);
int nsp_glu_bwd_b16(const void* x16, const float* dy, void* dx16, float* colsum_slabs, int rows, int C, void* stream);

/* ------------------------------------------------------------------------ *
 * Attention score softmax with relative-position term and in-kernel masks. *
 * Replaces relative_multihead_attention.py:112-144 (_rel_shift gather),    *
 * :196-206 (scale, masked_fill(NEG_INF=fp32 min), softmax) and             *
 * multihead_attention.py:137-142; mask semantics of transformer.py:633-686 *
 * (make_san_mask / causal / make_chunkwise_san_mask) are evaluated from    *
 * klens instead of reading a [B,T,T] mask tensor.                          *
 *   S  [B,H,Tq,Tk] content scores (AC), overwritten in place by softmax    *
 *   QP [B,Tq,H,R]  q . pos table (BD before shift), NULL for plain MHA     *
 *   e(i,j) = (S + QP[i, min(|i-j|,clamp) ]) * scale                        *
 *   key j is visible to query i iff j < klens[b]                           *
 *        and (!causal || j <= i + lookahead)                               *
 *        and (chunk_nc==0 || (j >= max(0,c0-chunk_nl) && j < c0+chunk_nc)  *
 *                             with c0 = (i/chunk_nc)*chunk_nc)             *
 *   masked scores are set to -FLT_MAX (not -inf), exactly as the reference.*
 * ------------------------------------------------------------------------ */
typedef struct {
  int B, H, Tq, Tk, R;     /* R = rows of the position table (clamp+1 or Tk) */
  int clamp;               /* <=0: no clamp */
  float scale;             /* 1/sqrt(d_k) */
  const int* klens;        /* [B] device int32 or NULL (no padding mask) */
  int causal, lookahead;
  int chunk_nl, chunk_nc;
  float dropout_p; unsigned long long seed, offset;
  int p_bf16;              /* probabilities / dS are bf16 images with row pitch tk_pitch (zero padded) */
  int tk_pitch;            /* >= Tk, multiple of 8 when p_bf16 */
  int r_pitch;             /* row pitch of QP / dQP (>= R; pad columns of dQP are written as 0) */
} nsp_attn_mask_params;

/* S: content scores fp32 [B,H,Tq,Tk] (read only).  Pout receives softmax(e) (fp32 pitch Tk, may
 * alias S; or bf16 pitch tk_pitch), Pdrop (required iff dropout_p > 0) the dropped copy. */
int nsp_attn_softmax_fwd(const float* S, const float* QP, void* Pout, void* Pdrop,
                         const nsp_attn_mask_params* p, void* stream);
/* dS (same dtype/pitch as P; may alias dP when fp32) = P*(dP - sum_j P*dP) * scale, 0 on masked
 * keys; dQP[b,i,h,r] = sum of dS(i,j) with rel(i,j)==r. */
int nsp_attn_softmax_bwd(const void* P, const float* dP, void* dS, float* dQP,
                         const nsp_attn_mask_params* p, void* stream);

/* ------------------------------------------------------------------------ *
 * Fused self-attention (flash style), d_k = 64, bf16 MFMA: the whole of      *
 * relative_multihead_attention.py:179-215 without materialising [B,H,T,T].  *
 *   qkv  bf16 [B*T, 3d]: q | k | v column blocks, head h at +h*64           *
 *   QP   fp32 [B,T,H,r_pitch] position scores (NULL for plain MHA); needs    *
 *        clamp > 0 and R <= 16                                              *
 *   O    bf16 [B*T, d] context (operand of the output projection);          *
 *   O32  fp32 [B*T, d] the same context un-rounded (may be NULL in inference):  *
 *        backward forms D_i = dO_i . O_i from it; the running row maximum is kept  *
 *        integer-valued (log2 domain) so that the bf16 probabilities of P V are    *
 *        reproduced bit for bit by backward (softmax shift invariance of dS, see   *
 *        flash_attn.hip)                                                           *
 *   LSE  fp32 [2,B,H,T]: ceil(row max) (log2 domain) and 1/row-sum of exp2(. - it) *
 *   keepbits (required iff dropout_p > 0; nsp_flash_attn_keepbits_bytes(B,H,T)     *
 *        bytes): the dropout decisions the forward drew (the reference's            *
 *        nn.Dropout on the attention weights, relative_multihead_attention.py:206), *
 *        one bit per (query, key) pair in the forward's register layout; backward   *
 *        reads them instead of drawing them again                                   *
 * Backward: dqkv bf16 [B*T,3d] receives dK (block d) and dV (block 2d);      *
 * dq32 fp32 [B*T,d] and dQP [B,T,H,r_pitch] are written (no zero-init);      *
 * D is scratch [B,H,T,4] fp32.  Masks / dropout as in nsp_attn_softmax_*.   *
 * ------------------------------------------------------------------------ */
long long nsp_flash_attn_keepbits_bytes(int B, int H, int T);
int nsp_flash_attn_fwd(const void* qkv, int d, const float* QP, void* O, float* O32, float* LSE, void* keepbits,
                       const nsp_attn_mask_params* p, void* stream);
/* query gradient: dq32 fp32 [B*T,d] (position term's share dQP . pos NOT included; pos16 NULL), or with dq32 == NULL     *
 * finished and rounded, in column block 0 of dqkv: dS k + dQP . pos16 (pos16 = projected position table [>=R, d] bf16, or *
 * NULL for plain MHA)                                                                                                    */
int nsp_flash_attn_bwd(const void* qkv, int d, const float* QP, const void* dO, const float* O32,
                       const float* LSE, const void* keepbits, float* D, void* dqkv, float* dq32, float* dQP,
                       const void* pos16, const nsp_attn_mask_params* p, void* stream);

/* ------------------------------------------------------------------------ *
 * Conformer convolution module core: depthwise Conv1d over time on         *
 * channels-last [B,T,C] (conformer_convolution.py:111-113; groups=C,       *
 * padding (k-1)/2, or causal: left pad k-1).                               *
 * ------------------------------------------------------------------------ */
/* y[b,t,c] = bias[c] + sum_j wt[jj][c] * x[b, t+j-pad, c], jj = flip ? k-1-j : j.
 * wt is tap-major [k][C] (the host transposes the reference's [C,1,k] weight).
 * The data gradient is the same kernel on dy with flip=1, pad = k-1-pad. */
int nsp_dwconv1d_fwd(const float* x, const float* wt /*[k,C]*/, const float* bias,
                     float* y, int B, int T, int C, int k, int pad, int flip, void* stream);
/* dwt[k][C], dbias[C] are accumulated atomically into caller-zeroed buffers */
int nsp_dwconv1d_wgrad(const float* x, const float* dy, float* dwt, float* dbias,
                       int B, int T, int C, int k, int pad, void* stream);
/* k <= 15: LDS-tiled variant without atomics.  Utterance b's time axis is cut into `tsplit`
 * ranges; workgroup (channel block, range, b) writes its partial [(k+1)][C] (taps, then the bias
 * row) to slab b*tsplit + range of `part` (B*tsplit slabs, all written); sum them with
 * nsp_splitk_reduce. */
int nsp_dwconv1d_wgrad_slabs(const float* x, const float* dy, float* part, int tsplit,
                             int B, int T, int C, int k, int pad, void* stream);
/* GLU folded into the depthwise conv (conformer_convolution.py:110-113) for the bf16 image h2 [B*T, 2C] that the first
 * pointwise conv's GEMM wrote: y = dwconv(glu(h2)); backward writes d h2 as bf16 (operand of the pointwise conv's gradient
 * GEMMs) and its column sums as slabs [nsp_dwconv1d_glu_bwd_slabs(B,T,C), 2C] fp32 (every row written); the weight gradient
 * as nsp_dwconv1d_wgrad_slabs with x = glu(h2).  k in {7, 15}, pad = (k-1)/2, C/4 a divisor of 256 (else NSP_EUNSUPPORTED). */
int nsp_dwconv1d_glu_bwd_slabs(int B, int T, int C);
int nsp_dwconv1d_glu_fwd(const void* h2, const float* wt, const float* bias, float* y, int B, int T, int C,
                         int k, int pad, void* stream);
int nsp_dwconv1d_glu_bwd(const void* h2, const float* dy, const float* wt, void* g16, float* colsum_slabs,
                         int B, int T, int C, int k, int pad, void* stream);
int nsp_dwconv1d_glu_wgrad_slabs(const void* h2, const float* dy, float* part, int tsplit,
                                 int B, int T, int C, int k, int pad, void* stream);

/* ------------------------------------------------------------------------ *
 * VGG-style Conv2d frontend on channels-last [B,T,F,C] (conv.py:289-396).  *
 * 3x3, padding 1, stride 1, fused bias + ReLU; MaxPool2d(ceil_mode).       *
 * ------------------------------------------------------------------------ */
/* mask_src (optional, layout of y): y = mask_src > 0 ? y : 0 -- lets the data-gradient call
 * apply the ReLU backward of the layer below in its epilogue. */
/* io_dtype (NSP_DT_F32 / NSP_DT_BF16): storage type of the 32-channel feature maps (x when Ci = 32, y,
 * mask_src); bf16 maps need NSP_COMPUTE_BF16 (every consumer rounds them to bf16 for its MFMA anyway, so
 * the stored rounding is free and halves the front-end's HBM traffic).  The Ci = 1 input is always fp32. */
int nsp_conv2d3x3_fwd(const void* x, const float* w /*[Co,3,3,Ci]*/, const float* bias,
                      void* y, int B, int T, int F, int Ci, int Co, int relu,
                      const void* mask_src, int mode, int io_dtype, void* stream);
/* dw [Co,3,3,Ci] and dbias [Co] accumulated atomically into caller-zeroed
 * buffers; dy must already be masked by the ReLU (nsp_relu_bwd).  The data
 * gradient is nsp_conv2d3x3_fwd on dy with the tap-flipped, channel-transposed
 * filter bank (built by the host, 9K floats). */
int nsp_conv2d3x3_wgrad(const void* x, const void* dy, float* dw, float* dbias,
                        int B, int T, int F, int Ci, int Co, int mode, int io_dtype, void* stream);
int nsp_relu_bwd(const float* y, const float* dy, float* dx, long long n, void* stream);
/* MaxPool2d(kernel=stride=(pt,pf), ceil_mode=True) on [B,T,F,C]; if
 * to_btcf != 0 the output is written as [B,T',C,F'] (= the reference's
 * transpose(2,1).view(B,T',C*F'), conv.py:189) */
/* argmax: ONE BYTE per output element = position of the maximum inside its window, dt * pf + df
 * (first maximum in scan order, like torch); pt * pf <= 256; C % 8 == 0 for bf16 inputs (C % 4 for fp32) */
int nsp_maxpool2d_fwd(const void* x, void* y, unsigned char* argmax, int B, int T, int F, int C,
                      int pt, int pf, int to_btcf, int x_dtype, int y_dtype, void* stream);
/* relu_src (optional, layout of dx): dx = relu_src > 0 ? dx : 0 (ReLU backward of the conv that
 * fed the pool, fused) */
/* dy_dtype: type of dy; dx_dtype: type of dx and relu_src */
int nsp_maxpool2d_bwd(const void* dy, const unsigned char* argmax, void* dx, int B, int T, int F,
                      int C, int pt, int pf, int from_btcf, const void* relu_src,
                      int dy_dtype, int dx_dtype, void* stream);
/* MaxPool1d(k=s=factor, ceil_mode=True) over time of [B,T,C]
 * (subsampling.py:188-209) */
int nsp_maxpool1d_fwd(const float* x, float* y, int* argmax, int B, int T, int C, int factor,
                      void* stream);
int nsp_maxpool1d_bwd(const float* dy, const int* argmax, float* dx, int B, int T, int C,
                      int factor, void* stream);

/* ------------------------------------------------------------------------ *
 * The other time subsamplers of encoders/subsampling.py on [B,T,C]         *
 * (C % 4 == 0), To output frames, window of k frames starting at           *
 * to*stride - pad (frames outside [0,T) count as zero):                    *
 *   window SUM     y[b,to,c]       = s(to) * sum_j x[b, to*stride+j-pad, c] *
 *     DropSubsampler (:97-128)     k=1, stride=f, To=ceil(T/f)              *
 *     AddSubsampler (:131-172)     k=2, stride=2, To=ceil(T/2)              *
 *     MeanPoolSubsampler (:212-246, AvgPool1d ceil_mode) k=stride=f, mean=1: *
 *       s(to) = 1 / #frames of the window inside [0,T)                      *
 *   window GATHER  y[b,to,j*C+c]   = x[b, to*stride+j-pad, c]  (im2col)      *
 *     ConcatSubsampler (:13-52)    k=stride=f, pad=0, To=T/f, then Linear    *
 *     Conv1dSubsampler (:55-94)    k=3, stride=f, pad=1, To=(T-1)/f+1, then  *
 *       the GEMM with the [Co, k*Ci] view of the Conv1d weight              *
 * The *_bwd entry points are the exact adjoints (one thread per input      *
 * element gathers from every window that covers it: no atomics).           *
 * ------------------------------------------------------------------------ */
int nsp_time_window_sum_fwd(const float* x, float* y, int B, int T, int To, int C, int k, int stride,
                            int pad, int mean, void* stream);
int nsp_time_window_sum_bwd(const float* dy, float* dx, int B, int T, int To, int C, int k, int stride,
                            int pad, int mean, void* stream);
int nsp_time_window_gather_fwd(const float* x, float* y, int B, int T, int To, int C, int k, int stride,
                               int pad, void* stream);
int nsp_time_window_gather_bwd(const float* dy, float* dx, int B, int T, int To, int C, int k,
                               int stride, int pad, void* stream);

/* im2col of a 3x3 / pad 1 / stride 1 convolution on channels-last x [B,T,F,Ci], for input-channel counts the MFMA
 * conv kernels do not take (conv_in_channel = 3, conv.py:167-175): cols [B*T*F, Kp], Kp >= 9*Ci,
 * cols[(b,t,f), ci*9 + kh*3 + kw] = x[b, t+kh-1, f+kw-1, ci], zero outside the map and in columns 9*Ci..Kp-1
 * (= the column order of nn.Conv2d's weight viewed as [Co, Ci*9]); the convolution is then one GEMM. */
int nsp_im2col3x3(const float* x, float* cols, int B, int T, int F, int Ci, int Kp, void* stream);

/* pack_padded_sequence / pad_packed_sequence around the (B)LSTM layers of    *
 * encoders/rnn.py:534-541, as data movement on [B,T,C] (C % 4 == 0; rows may *
 * be strided: x_ld / y_ld floats per frame, multiples of 4):                *
 *   y[b,t,:] = t < lens[b] ? x[b, flip ? lens[b]-1-t : t, :] : 0            *
 * flip = 0: zero the frames past each utterance's end; flip = 1: also       *
 * reverse every utterance inside its own length (a left-to-right LSTM over  *
 * the result is the backward direction of a packed BLSTM).  Self-adjoint.   */
int nsp_time_flip_mask(const float* x, long long x_ld, float* y, long long y_ld, const int* lens,
                       int B, int T, int C, int flip, void* stream);

/* ------------------------------------------------------------------------ *
 * BatchNorm1d / GroupNorm + activation of the Conformer convolution module *
 * on the flattened rows x [M = B*T, C] (conformer_convolution.py:58-66,    *
 * 119-124: `norm(xs.view(B*T, C, 1))` then Swish), C % 4 == 0.             *
 * Column reductions run over nsp_col_reduce_slabs(M) row slabs;            *
 * `part` is a caller-provided workspace of slabs*2*C floats.               *
 * nsp_bn_stats: batch mean / 1/sqrt(biased var + eps) per channel (shifted *
 *   moments, shift = row 0) and, when the pointers are non-NULL, the       *
 *   running_mean / running_var (unbiased) / num_batches_tracked update of  *
 *   nn.BatchNorm1d(momentum).                                              *
 * nsp_bn_act_fwd: y = act(gamma * (x - mean) * rstd + beta); `scale` is    *
 *   rstd (training: from nsp_bn_stats) or, with scale_is_var != 0, a       *
 *   variance (eval: running_var) from which rstd = 1/sqrt(var + eps).      *
 * nsp_bn_act_bwd: dz = dy*act'(z); dbeta = sum dz; dgamma = sum dz*xhat;   *
 *   dx = gamma*rstd*(dz - (dbeta + xhat*dgamma)/M) when training != 0       *
 *   (batch statistics), gamma*rstd*dz otherwise.                           *
 * nsp_gn2_*: GroupNorm with groups of TWO adjacent channels -- what        *
 *   `nn.GroupNorm(max(1, d_model // 2), d_model)` builds for every even    *
 *   d_model (:61-63); statistics per row and pair, nothing saved.          *
 * ------------------------------------------------------------------------ */
int nsp_col_reduce_slabs(long long M);
int nsp_bn_stats(const float* x, long long M, int C, float eps, float momentum, float* part,
                 float* mean, float* rstd, float* running_mean, float* running_var,
                 long long* num_batches_tracked, void* stream);
int nsp_bn_act_fwd(const float* x, const float* mean, const float* scale, int scale_is_var, float eps,
                   const float* gamma, const float* beta, int act, float* y, long long M, int C,
                   void* stream);
int nsp_bn_act_bwd(const float* x, const float* dy, const float* mean, const float* scale,
                   int scale_is_var, float eps, const float* gamma, const float* beta, int act,
                   int training, float* part, float* dgamma, float* dbeta, float* dx, long long M, int C,
                   void* stream);
int nsp_gn2_act_fwd(const float* x, const float* gamma, const float* beta, float eps, int act, float* y,
                    long long M, int C, void* stream);
int nsp_gn2_act_bwd(const float* x, const float* dy, const float* gamma, const float* beta, float eps,
                    int act, float* part, float* dgamma, float* dbeta, float* dx, long long M, int C,
                    void* stream);

/* ------------------------------------------------------------------------ *
 * CTC: fused log-softmax + alpha/beta lattice + gradient w.r.t. logits.    *
 * Replaces ctc.py:139-150 (nn.CTCLoss(reduction='sum', zero_infinity=True) *
 * on logits.log_softmax(2)) and criterion.py:110-127 (kldiv_lsm_ctc).      *
 *   logits [B,T,V]; labels [B,Lmax] int32 (padded); elens/ylens [B] int32  *
 *   nll[b] = -log p(y_b | x_b) (0 if infinite); grad [B,T,V] = d sum_b     *
 *   nll_b / d logits (zero for t >= elens[b]).                             *
 *   workspace: alpha/beta fp32 [B,T,2*Lmax+1] each + lse [B,T].            *
 * ------------------------------------------------------------------------ */
long long nsp_ctc_workspace_bytes(int B, int T, int Lmax);
int nsp_ctc_loss_fwd_bwd(const float* logits, const int* labels, const int* elens,
                         const int* ylens, float* nll, float* grad, float gscale,
                         void* workspace, int B, int T, int V, int Lmax, int blank,
                         void* stream);
/* label-smoothing KL term and its gradient (criterion.py:110-127):
 * kl_sum[0] += sum_{b,t<elens_b,v} p (log p - log(1/(V-1))); grad (+)= gscale * d kl_sum */
int nsp_ctc_kldiv_fwd_bwd(const float* logits, const int* elens, float* kl_sum,
                          float* grad, float gscale, int accumulate,
                          int B, int T, int V, void* stream);
/* CTC forced alignment (ctc.py:628-753): trigger_points [B,Lmax+1] int32 */
long long nsp_ctc_align_workspace_bytes(int B, int T, int Lmax);
int nsp_ctc_forced_align(const float* logits, const int* labels, const int* elens,
                         const int* ylens, int* trigger_points, void* workspace,
                         int B, int T, int V, int Lmax, int blank, void* stream);

/* ------------------------------------------------------------------------ *
 * RNN-T: joint log-softmax gather + alpha/beta lattice loss + gradients.   *
 * Replaces rnn_transducer.py:242-256 (log_softmax over [B,T,U+1,V] +       *
 * warp_rnnt.rnnt_loss(average_frames=False, reduction='mean',gather=False) *
 * / warprnnt_pytorch.RNNTLoss()).                                          *
 *   nsp_rnnt_logsoftmax_gather: logits [B,T,U1,V] -> lse [B,T,U1],         *
 *        lp_blank, lp_label [B,T,U1] (label of (t,u) is labels[b,u])       *
 *   nsp_rnnt_lattice: alpha/beta recursions, nll[b], lattice occupancies   *
 *        g_blank/g_label [B,T,U1] = d nll_b / d lp_*                       *
 *   nsp_rnnt_grad_logits: in place logits <- d(sum_b w_b nll_b)/d logits   *
 * ------------------------------------------------------------------------ */
int nsp_rnnt_logsoftmax_gather(const float* logits, const int* labels, const int* elens,
                               const int* ylens, float* lse, float* lp_blank, float* lp_label,
                               int B, int T, int U1, int V, int blank, void* stream);
int nsp_rnnt_lattice(const float* lp_blank, const float* lp_label, const int* elens,
                     const int* ylens, float* alpha, float* beta, float* nll,
                     float* g_blank, float* g_label, int B, int T, int U1, void* stream);
/* out16 == NULL: logits are overwritten in place with the gradient (fp32); otherwise the
 * gradient is written as a bf16 image [rows, ld16] (ld16 >= V, zero padded) and logits are
 * left untouched. */
int nsp_rnnt_grad_logits(float* logits, const float* lse, const int* labels,
                         const float* g_blank, const float* g_label, const int* elens,
                         const int* ylens, float wscale, const float* wscale_dev /*optional device
                         scalar multiplied into wscale: the upstream gradient, no host sync*/,
                         int B, int T, int U1, int V, int blank, void* out16, int ld16,
                         float* dbias /*optional [V], zeroed by the caller: column sums of the
                         gradient = output-bias gradient, accumulated in the same pass*/,
                         void* stream);
/* joint pre-activation: h[b,t,u,:] = tanh(e[b,t,:] + g[b,u,:]) and its backward
 * reductions (rnn_transducer.py:272-274) */
/* h (fp32) and/or h16 (bf16) receive the activation */
int nsp_rnnt_joint_tanh_fwd(const float* e, const float* g, float* h, void* h16,
                            int B, int T, int U1, int J, void* stream);
/* dh is overwritten with dz = dh*(1-h^2); de[b,t,:] = sum_u dz, dg[b,u,:] = sum_t dz */
int nsp_rnnt_joint_tanh_bwd(const float* h, const void* h16, float* dh, float* de, float* dg,
                            int B, int T, int U1, int J, void* stream);
/* bf16 path: dz = dh * (1 - h^2) already formed (bf16 [B,T,U1,J]) by the data-gradient GEMM's
 * epilogue (dact = NSP_ACT_TANH_OUT); one pass over it yields de[b,t,:] = sum_u dz (written) and
 * `nslab` partial sums over time of dg[b,u,:] = sum_t dz (slab z = steps [z*ceil(T/nslab), ...),
 * each [B,U1,J] fp32; nslab slabs are always written; sum them with nsp_splitk_reduce).
 * J % 32 == 0, U1 <= 512. */
/* ---- fused / compacted RNN-T joint (bf16 throughput mode): the [B,T,U+1,V] logit tensor is never  *
 * written (rnn_transducer.py:239-242,262-276 materialise it twice: logits and log_softmax).         *
 * Lattice nodes are stored COMPACTED: utterance b owns rows roff[b] .. roff[b+1] laid out           *
 * [T_b][U_b+1] (t-major); padded nodes do not exist.  M = roff[B].                                   *
 *   nsp_rnnt_joint_tanh_compact: h16[row] = bf16 tanh(e[b,t,:] + g[b,u,:]) (J % 8 == 0);             *
 *        lab[row] = labels[b,u] for u < U_b, -1 at u = U_b.  e [B,T,J], g [B,U1,J] fp32.             *
 *   nsp_rnnt_joint_gemm: logits = h16 W^T + b on the bf16 MFMA GEMM with the NSP_EPI_RNNT_* epilogues *
 *        (mode LSE in forward; mode DLOGITS recomputes the logit tiles in backward and emits the     *
 *        bf16 gradient image d16 [M,Vp] + bias-gradient slabs).  w16 bf16 [Vp,J] zero-padded rows,   *
 *        bias [Vp] fp32; f0..f3 as documented at nsp_gemm_params.                                    *
 *   nsp_rnnt_lse_merge: partials -> lse [M], lp_blank = raw_b - lse, lp_label = raw_l - lse (-inf at  *
 *        lab < 0), in place over raw_b / raw_l.                                                      *
 *   nsp_rnnt_lattice_compact: alpha / beta / nll / occupancies on the compact layout.                *
 *   nsp_rnnt_joint_dz_reduce_compact: de[b,t,:] = sum_u dz, dg slabs [nslab][B,U1,J] = sum_t dz      *
 *        (zeros at padded t / u), dz bf16 [M,J].                                                     */
int nsp_rnnt_joint_tanh_compact(const float* e, const float* g, const int* labels /*[B,U1-1]*/,
                                const int* elens, const int* ylens, const long long* roff /*[B+1]*/,
                                void* h16, int* lab, int B, int T, int U1, int J, void* stream);
int nsp_rnnt_joint_gemm(int epi_mode, const void* h16, const void* w16, const float* bias, long long M,
                        int V, int Vp, int J, int blank, const int* lab, float* f0, float* f1, float* f2,
                        float* f3, void* d16, float scale, const float* scale_dev,
                        float* rec /* DLOGITS: workspace [M,4] fp32, 16-B aligned; LSE: NULL */, void* stream);
int nsp_rnnt_lse_merge(const float* part, int npart, float* lse, float* raw_b_to_lpb, float* raw_l_to_lpl,
                       const int* lab, long long M, void* stream);
/* The same two passes, NODE-STATIONARY (J = 128 / 256 / 512; anything else: NSP_EUNSUPPORTED -> use nsp_rnnt_joint_gemm):
 * one workgroup owns 256 lattice nodes for the WHOLE padded vocabulary, so nothing per 64-column block exists.
 *   NSP_EPI_RNNT_LSE    : lse [M], f1 [M] = logit(blank) - lse, f2 [M] = logit(lab[m]) - lse (-inf where lab[m] < 0) are
 *                         WRITTEN (what nsp_rnnt_joint_gemm + nsp_rnnt_lse_merge leave behind); dbslabs / d16 unused.
 *   NSP_EPI_RNNT_DLOGITS: lse, f1 = g_blank, f2 = g_label [M] are READ (scaled by scale * scale_dev[0]); d16 bf16 [M, Vp]
 *                         receives the gradient image; dbslabs (optional) fp32 [ceil(M / 256), Vp] its column sums per
 *                         workgroup, every entry written (rnn_transducer.py:239-242,262-276). */
int nsp_rnnt_joint_rows(int epi_mode, const void* h16, const void* w16, const float* bias, long long M, int V, int Vp,
                        int J, int blank, const int* lab, float* lse, float* f1, float* f2, float* dbslabs, void* d16,
                        float scale, const float* scale_dev, void* stream);
int nsp_rnnt_lattice_compact(const float* lp_blank, const float* lp_label, const int* elens, const int* ylens,
                             const long long* roff, float* alpha, float* beta, float* nll, float* g_blank,
                             float* g_label, int B, int U1max, void* stream);
int nsp_rnnt_joint_dz_reduce_compact(const void* dz16, const int* elens, const int* ylens, const long long* roff,
                                     float* de, float* dg_slabs, int nslab, int B, int T, int U1, int J,
                                     void* stream);
int nsp_rnnt_joint_dz_reduce(const void* dz16, float* de, float* dg_slabs, int nslab, int B, int T,
                             int U1, int J, void* stream);

/* ------------------------------------------------------------------------ *
 * LSTM recurrence of the RNN-T prediction network (rnn_transducer.py:101-111,  *
 * 278-311: nn.LSTM(in, H, 1 layer, batch_first), zero initial state).         *
 * gi = x W_ih^T + b_ih + b_hh for all steps is a GEMM done by the caller;      *
 * nsp_lstm_fwd runs the L sequential steps h_{t-1} W_hh^T + cell update and    *
 * saves the activated gates (i,f,g,o) and cell states; nsp_lstm_bwd walks back *
 * and emits dgates [B,L,4H], from which dx, dW_ih, dW_hh, db are GEMMs/colsums.*
 * Whh / WhhT / *shadow are bf16 in NSP_COMPUTE_BF16 and fp32 in NSP_COMPUTE_F32*
 * ------------------------------------------------------------------------ */
int nsp_lstm_fwd(const float* gi, const void* Whh /*[4H,H]*/, float* y /*[B,L,H]*/,
                 void* yshadow /*bf16 [B,L,H]*/, float* c_all /*[B,L,H]*/,
                 float* gates /*[B,L,4H]*/, int B, int L, int H, int mode, void* stream);
int nsp_lstm_bwd(const float* dy /*[B,L,H]*/, const void* WhhT /*[H,4H]*/, const float* c_all,
                 const float* gates, float* dgates /*[B,L,4H]*/, void* dgshadow /*bf16*/,
                 float* dc /*[B,H] scratch*/, int B, int L, int H, int mode, void* stream);
/* The same step kernels over the steps [t_begin, t_end) only -- an LSTM with a given initial state and a gradient
 * w.r.t. its final state (forward direction of the latency-controlled BLSTM, encoders/rnn.py:466-475, whose state is
 * carried from chunk to chunk): the caller lays the sequence out in rows 1..n of buffers with L = n + 2 rows, puts
 * (h0, c0) into row 0 of y / yshadow / c_all and runs [1, n+1); backward: dy[:, n] carries the final-h gradient,
 * `dc` the final-c gradient, dgates[:, n+1] = 0, steps n..1; afterwards `dc` is d/dc0 and dgates[:, 1] W_hh d/dh0. */
int nsp_lstm_fwd_range(const float* gi, const void* Whh, float* y, void* yshadow, float* c_all, float* gates,
                       int B, int L, int H, int mode, int t_begin, int t_end, void* stream);
int nsp_lstm_bwd_range(const float* dy, const void* WhhT, const float* c_all, const float* gates, float* dgates,
                       void* dgshadow, float* dc, int B, int L, int H, int mode, int t_begin, int t_end,
                       void* stream);

/* ------------------------------------------------------------------------ *
 * The whole LSTM stack of the prediction network as ONE wavefront over        *
 * (layer, time): stage s runs step t = s - l of every layer l in one launch, *
 * so n layers of L steps cost L + n - 1 dependent launches instead of n * L  *
 * (rnn_transducer.py:278-311 runs the layers one after the other).  Layers   *
 * l >= 1 fold their input projection into the step (K = 2H: [W_ih | W_hh]);  *
 * the inter-layer dropout (rnn_transducer.py:303) is applied where layer l   *
 * publishes its output for layer l+1 (counter-based mask from seed/offset).  *
 * bf16 operands only (NSP_COMPUTE_BF16); H % 64 == 0; nl <= NSP_LSTM_MAX_LAYERS.
 * Forward fields: gi0 [B,L,4H] = x W_ih0^T + b (GEMM by the caller);         *
 *   w[0] = W_hh0 bf16 [4H,H]; w[l>=1] = [W_ih_l | W_hh_l] bf16 [4H,2H];      *
 *   bias[l>=1] = b_ih_l + b_hh_l fp32 [4H] (bias[0] unused);                 *
 *   y_top fp32 [B,L,H] (output of the last layer, before any dropout);       *
 *   hp16[l] bf16 [B,L,H]: h shifted by one step (hp16[b,t] = h_{t-1}, 0 at t=0):
 *     the recurrent operand and, in backward, the W_hh weight-gradient operand;
 *   yd16[l] bf16 [B,L,H], l < nl-1: dropout(h_l) = layer l+1's input;        *
 *   c_all[l] fp32 [B,L,H], gates[l] fp32 [B,L,4H] (activated i,f,g,o).       *
 * Backward fields: dy_top fp32 [B,L,H]; w[nl-1] = W_hh^T bf16 [H,4H];         *
 *   w[l<nl-1] = [W_ih_{l+1}^T | W_hh_l^T] bf16 [H,8H]; dg16[l] bf16 [B,L,4H]  *
 *   receives d(pre-activation gates); dc[l] fp32 [B,H] scratch.              *
 * ------------------------------------------------------------------------ */
#define NSP_LSTM_MAX_LAYERS 4
typedef struct {
  int nl, B, L, H;
  float dropout_p;
  int reserved;
  const float* gi0;
  const float* dy_top;
  float* y_top;
  const void* w[NSP_LSTM_MAX_LAYERS];
  const float* bias[NSP_LSTM_MAX_LAYERS];
  void* hp16[NSP_LSTM_MAX_LAYERS];
  void* yd16[NSP_LSTM_MAX_LAYERS];
  float* c_all[NSP_LSTM_MAX_LAYERS];
  float* gates[NSP_LSTM_MAX_LAYERS];
  void* dg16[NSP_LSTM_MAX_LAYERS];
  float* dc[NSP_LSTM_MAX_LAYERS];
  unsigned long long seed[NSP_LSTM_MAX_LAYERS];
  unsigned long long offset[NSP_LSTM_MAX_LAYERS];
  void* xchg[NSP_LSTM_MAX_LAYERS];   /* persistent launches only: per-layer exchange scratch, see below */
} nsp_lstm_stack_params;
int nsp_lstm_stack_fwd(const nsp_lstm_stack_params* p, void* stream);
int nsp_lstm_stack_bwd(const nsp_lstm_stack_params* p, void* stream);
/* The same recurrences as ONE persistent launch each: every workgroup keeps its slice of the   *
 * weights in registers for all stages (the per-stage launches above re-read 8-16 MB of weights *
 * from beyond L2 after every launch boundary), stages are separated by a grid barrier (one      *
 * monotonic device counter; h / dgates handed over with write-through stores + agent acquire). *
 * Requires B <= 64 (batch blocks of 16 are looped inside a stage), H % 256 == 0, H <= 1024,     *
 * nl * H/16 co-resident workgroups (<= 256).                                                   *
 * p->xchg[l] = scratch for the fragment-major hand-over between stages (csrc/lstm.hip):        *
 *   forward 2 * L * 64 * H bf16 elements per layer, backward L * 64 * 4H.                        *
 * `sync` = 2 zero-initialised 32-bit words owned by this call (barrier counter, abort flag):   *
 * spins are bounded; on expiry the flag is set and y_top[0] / dg16[0][0] become NaN instead of *
 * the call hanging.  Returns NSP_EUNSUPPORTED when the shape does not qualify.                 */
int nsp_lstm_stack_fwd_persistent(const nsp_lstm_stack_params* p, unsigned int* sync, void* stream);
int nsp_lstm_stack_bwd_persistent(const nsp_lstm_stack_params* p, unsigned int* sync, void* stream);

/* ------------------------------------------------------------------------ *
 * SpecAugment band zeroing in place on [B,T,F] (spec_augment.py:112-140).  *
 * bands are small device arrays of [start,end) pairs (drawn on the host     *
 * with the reference's np.random stream, then copied).                      *
 * ------------------------------------------------------------------------ */
int nsp_specaug_apply(float* x, int B, int T, int F,
                      const int* freq_bands /*device [n_freq][2]*/, int n_freq,
                      const int* time_bands /*device [n_time][2]*/, int n_time, void* stream);

/* pad a ragged batch: src is one packed device buffer of sum(T_b)*F floats
 * (torch_utils.py:56 pad_list + speech2text.py:397) */
int nsp_pad_batch(const float* packed, const long long* offsets /*device [B]*/,
                  const int* lens /*device [B]*/, float* out, int B, int Tmax, int F,
                  float pad_value, void* stream);

/* ------------------------------------------------------------------------ *
 * Label-smoothed cross entropy of the attention decoders + gradient +      *
 * teacher-forcing accuracy in one kernel (criterion.py:45-86                *
 * cross_entropy_lsm, decoders/transformer.py:442; torch_utils.py:129-145).  *
 *   logits fp32 [rows, V]; ys int32 [rows] (ignore_index rows contribute 0);*
 *   loss_rows[r] = -sum_v target_v log_softmax(logits_r)_v with             *
 *   target = 1 - lsm at ys[r], lsm / (V-1) elsewhere; correct[r] = argmax == *
 *   ys[r]; grad (may be NULL) = grad_scale * (softmax - target).            *
 * ------------------------------------------------------------------------ */
int nsp_xe_lsm_fwd_bwd(const float* logits, const int* ys, float* loss_rows, int* correct, float* grad,
                       long long rows, int V, int ignore_index, float lsm_prob, float grad_scale, void* stream);

/* ------------------------------------------------------------------------ *
 * Greedy decoding helpers (validate(): train.py:341 -> evaluators ->        *
 * Speech2Text.decode, speech2text.py:709-800).                             *
 * nsp_argmax_rows: out[r] = argmax_c x[r*ld + c] (first index on ties) --   *
 *   the best path of ctc.py:229-230 and the per-frame 1-best of             *
 *   rnn_transducer.py:363-364, without a host sync per frame.               *
 * nsp_lstm_cell_step: one LSTM cell update from pre-activation gates        *
 *   [B,4H] (i,f,g,o) and the previous state; rows with update[b] == 0 keep   *
 *   their state (rnn_transducer.py:367-370: the prediction network advances *
 *   only where a non-blank label was emitted).  update may be NULL.         *
 * ------------------------------------------------------------------------ */
int nsp_argmax_rows(const float* x, int* out, long long rows, int cols, long long ld, void* stream);
int nsp_lstm_cell_step(const float* gates, const float* h_prev, const float* c_prev,
                       const int* update /*device [B] or NULL*/, float* h_out, float* c_out,
                       int B, int H, void* stream);

/* ------------------------------------------------------------------------ *
 * Monotonic (chunkwise) attention training scans (MoChA / MMA), fp32, `rows`  *
 * independent rows of klen encoder frames, one wave per row (mocha.hip).      *
 * hma_train.py:12-67: p = (1 - stableemit) sigmoid(e); c = exclusive cumprod  *
 *   of clamp(1 - p, eps, 1) in log space; alpha = p c cumsum(aw_prev / den),   *
 *   den = clamp(c, eps, 1) (1 when no_denom).  p_choose / cprod are saved for  *
 *   the backward, which returns d e and d aw_prev.                              *
 * mocha_train.py:13-83: ex = max(exp(u - max u), 1e-5); den_j = sum of ex over  *
 *   the w frames ending at j; beta_i = ex_i sum_{j=i}^{i+w-1} alpha_j sf / den_j *
 *   (w = -1: MILk, den = prefix sum, outer sum to the end of the row; w <= 64). *
 * ------------------------------------------------------------------------ */
int nsp_mono_alpha_fwd(const float* e, const float* aw_prev, float* alpha, float* p_choose, float* cprod,
                       int rows, int klen, float eps, int no_denom, float stableemit, void* stream);
int nsp_mono_alpha_bwd(const float* d_alpha, const float* p_choose, const float* cprod, const float* aw_prev,
                       float* d_e, float* d_aw_prev, int rows, int klen, float eps, int no_denom,
                       float stableemit, void* stream);
int nsp_chunk_beta_fwd(const float* u, const float* alpha, float* beta, int rows, int klen, int w, float sf,
                       void* stream);
int nsp_chunk_beta_bwd(const float* d_beta, const float* u, const float* alpha, float* d_u, float* d_alpha,
                       int rows, int klen, int w, float sf, void* stream);

/* ------------------------------------------------------------------------ *
 * Per-step pieces of the attention-based LSTM decoder (decoder_step.hip),   *
 * fp32.  las.py:667-776, modules/attention.py:148-176, monotonic_energy.py:  *
 * 137-146, chunk_energy.py.                                                 *
 *  add_energy: e[b,t] = sum_a v[a] act(K[b,t,a] + Q[b,a] (+ C[b,t,a]));     *
 *    act = NSP_ACT_TANH | NSP_ACT_RELU; C may be NULL.  bwd: dtmp [B,T,A] =  *
 *    de v act' (gradient of K and of C), dQ [B,A] = sum_t dtmp, dv_part      *
 *    [B,A] = sum_t de act(tmp) (dv = its sum over b).                        *
 *  row_softmax: aw = softmax(sharp * e) over frames with mask != 0 (mask    *
 *    uint8 [rows,T] or NULL); masked frames get no gradient.                 *
 *  lstm_cell: gates [B,4H] pre-activation, PyTorch order (i,f,g,o).          *
 * ------------------------------------------------------------------------ */
int nsp_add_energy_fwd(const float* K, const float* Q, const float* C, const float* v, float* e, int B, int T,
                       int A, int act, void* stream);
int nsp_add_energy_bwd(const float* de, const float* K, const float* Q, const float* C, const float* v,
                       float* dtmp, float* dQ, float* dv_part, int B, int T, int A, int act, void* stream);
int nsp_row_softmax_fwd(const float* e, const unsigned char* mask, float* aw, int rows, int T, float sharp,
                        void* stream);
int nsp_row_softmax_bwd(const float* aw, const float* daw, const unsigned char* mask, float* de, int rows, int T,
                        float sharp, void* stream);
int nsp_lstm_cell_fwd(const float* gates, const float* c_prev, float* h, float* c, int B, int H, void* stream);
int nsp_lstm_cell_bwd(const float* dh, const float* dc_next, const float* gates, const float* c_prev, const float* c,
                      float* dgates, float* dc_prev, int B, int H, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NSP_HIP_H */
