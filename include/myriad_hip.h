/* libmyriad_hip.so -- C ABI of the MI355X (gfx950) kernels behind the Myriad / MiniGPT-4 hot path.
 *
 * The reference (tzjtatata/Myriad) has no FFI of its own: every op below replaces a stock PyTorch op sequence
 * inside the reference's nn.Module sub-blocks (file:line cited per entry, paths relative to the reference
 * checkout).  The drop-in boundary one level up is the registered model class (myriad_amd.Myriad / MiniGPT4).
 *
 * Conventions: plain device pointers owned by the caller, explicit dims / leading
 * dimensions in ELEMENTS, explicit hipStream_t, no hidden allocation (workspaces are passed in).  Library state is what
 * the caller registers: the split-K scratch -- in an opaque mh_ctx (mh_ctx_set_workspace + mh_ctx_make_current) or in the
 * process-default record (mh_set_workspace / mh_set_stream_workspace); one scratch per stream that may split K: two streams
 * must not share one -- the launch profiler's event pool (mh_prof_*, off unless started) and the option table below
 * (mh_set_option / mh_get_option: on/off switches between equivalent code paths, each defaulting to the measured-best setting).
 * The shipped library exports nothing else: the timing probes and sweep switches of tools/ (mhdbg_*) are compiled only into
 * libmyriad_hip_dbg.so (-DMH_DEBUG_HOOKS).  Calls on different streams with their own scratch are independent.
 * Every function returns 0 on success or a negative MH_ERR_* code and never throws.
 * bf16 tensors are raw uint16 bit patterns; "f32" means IEEE float.
 */
#ifndef MYRIAD_HIP_H
#define MYRIAD_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* mh_stream_t; /* == hipStream_t */

#define MH_OK 0
#define MH_ERR_ARG (-1)
#define MH_ERR_LAUNCH (-2)
#define MH_ERR_UNSUPPORTED (-3)

/* mh_gemm_bf16_nt flags */
#define MH_GEMM_OUT_F32 1  /* C is f32 (else bf16) */
#define MH_GEMM_GELU 2     /* erf-GELU after bias, before residual */
#define MH_GEMM_REGSTAGE 4 /* register-staged LDS fill instead of LDS-DMA (A/B testing) */

/* K1/K2  C[M,N] = alpha * A[M,K] . B[N,K]^T (+bias[N]) (GELU) (+residual[M,N] f32).  A,B bf16, K % 64 == 0.
 * Replaces every nn.Linear on the path: eva_vit.py:124,146,55-59; Qformer.py:127-133,281,352,367;
 * myriad.py:263 (llama_proj); modeling_llama.py:134-136,159-162,604; and their autograd dgrad/wgrad. */
int mh_gemm_bf16_nt(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                    const float* bias, const float* residual, int ldr, int flags, float alpha, mh_stream_t s);

/* Which kernel mh_gemm_bf16_nt will launch for this shape and with how many K splits (splits > 1 adds one
 * fixed-order reduce launch): *kernel = 0 weight-streaming gemv (M <= 16), 1 = 128x128x64 tile (gemm_nt_kernel),
 * 2 = 256x256x64 tile, eight waves, hand-scheduled K loop (gemm_x8_kernel; gemm_256_kernel, the 256x256x32 kernel of
 * rounds 1-3, only for operands whose byte offsets pass 2 GiB or with option gemm256_impl = 0), 3 = gemm_nt_kernel with a
 * 128x64 tile (grids that leave most CUs empty), 4 / 5 = gemm_nt_kernel with a 160x128 / 160x96 tile and a 4-deep ring
 * (128 < M <= 320: the batch-1 step's 148 / 257 rows as one / two row tiles, the weight streamed once), 6 = gemm_nt_kernel
 * with a 64x64 tile and an 8-deep ring (one-round grids with K <= 3072).
 * For profilers and benchmarks that attribute time per kernel. */
int mh_gemm_plan(int M, int N, int K, int flags, int* kernel, int* splits);

/* Launch profiler (`roofline.achieved` is per launch of the dominant kernel): between mh_prof_start and
 * mh_prof_stop every launch of gemm_x8_kernel / gemm_256_kernel / gemm_nt_kernel is bracketed by two HIP events on the stream it is launched
 * on -- the kernel alone, also when it is the partial-product launch of a split-K op.  Launches inside a stream capture are
 * skipped.  mh_prof_stop waits for the device and writes, per recorded launch i, meta[6i..6i+5] = (kernel id as
 * mh_gemm_plan numbers them, M, N, K, K splits, flags) and ms[i]; returns the number of records (<= cap).  The events
 * (2 per record) are created once and reused.  One host thread.  Both calls launch a one-wave no-op kernel
 * (mh_prof_marker_kernel) on `s`, so that a rocprofv3 kernel trace of the same run shows where the profiled region lies. */
int mh_prof_start(int capacity, mh_stream_t s);
double mh_prof_overhead_ms(void); /* median of 33 EMPTY event pairs timed by the last mh_prof_start: what a pair adds to a launch */
int mh_prof_stop(int* meta, float* ms, int cap, mh_stream_t s);

/* Linear + residual add + the RMSNorm that consumes the sum (modeling_llama.py:281-293 then :66-74 of the next block):
 * H[M,N] = A.B^T + residual (f32, the new residual stream), Y[M,N] = bf16(norm_w * H * rsqrt(mean(H^2) + eps)).
 * When the GEMM is split along K the slab reduction, the residual add and the norm run as ONE kernel; otherwise this is
 * mh_gemm_bf16_nt followed by mh_rmsnorm_fwd.  Both forms give bit-identical H and Y. */
int mh_gemm_residual_rmsnorm(const void* A, int lda, const void* B, int ldb, float* H, int ldh, const float* residual,
                             int ldr, const float* norm_w, float eps, void* Y, long ldy, int M, int N, int K,
                             mh_stream_t s);

/* Same for the pre-LN ViT blocks (eva_vit.py:173-180; ImageBind transformer.py:160-163): H = A.B^T + bias + residual,
 * Y = bf16(LayerNorm(H) * norm_w + norm_b), dense [M,N] outputs. */
int mh_gemm_residual_layernorm(const void* A, int lda, const void* B, int ldb, float* H, int ldh, const float* bias,
                               const float* residual, int ldr, const float* norm_w, const float* norm_b, float eps, void* Y,
                               int M, int N, int K, mh_stream_t s);
/* dY = A.B^T (a dgrad Linear) followed by the RMSNorm backward that consumes it: dx = d rmsnorm(x; w)(dY) + dres, written
 * as f32 (dx) and/or bf16 (dx_bf16).  dy_buf: [M, N] f32 scratch (used when K is not split).  Same bits as
 * mh_gemm_bf16_nt(..., MH_GEMM_OUT_F32) + mh_rmsnorm_bwd; one launch and one pass over dY less when K is split. */
int mh_gemm_rmsnorm_bwd(const void* A, int lda, const void* B, int ldb, float* dy_buf, const float* x, const float* w,
                        const float* dres, float* dx, void* dx_bf16, int M, int N, int K, float eps, mh_stream_t s);

/* Scratch for the automatic split-K path of mh_gemm_bf16_nt (used for shapes whose tile count under-fills the
 * 256 CUs).  The caller owns the buffer; pass NULL to disable.  Not needed for correctness. */
int mh_set_workspace(void* ptr, long bytes);
/* Further scratches (each >= the first one's size) for launches on other streams (up to 4), so that a frozen forward or a
 * leaf backward can run beside the main stream without sharing split-K slabs with it.  ptr = NULL unregisters. */
int mh_set_stream_workspace(mh_stream_t stream, void* ptr, long bytes);
/* split-K variant (f32 out, no epilogue) for skinny outputs with a long reduction (conv-stem wgrad):
 * ws holds mh_gemm_splitk_ws_floats(M,N,splits) floats; fixed-order reduction -> deterministic. */
long mh_gemm_splitk_ws_floats(int M, int N, int splits);
int mh_gemm_bf16_nt_splitk(const void* A, int lda, const void* B, int ldb, float* C, int ldc, int M, int N, int K,
                           int splits, float* ws, mh_stream_t s);

/* K3/K4/K5 fused attention.  q/k/v/o token-major [B,S,ld] bf16, head h at columns [h*D,(h+1)*D); D in
 * {<=64, <=96 (88), <=128}; lse [B,H,Sq] f32; bias optional additive [H,Sq,Sk] f32 (eva_vit.py:131-140);
 * kv_len optional [B] valid-key counts (right padding, modeling_llama.py:43-54); causal aligns the last query
 * with the last key (KV-cache decode: Sq=1).  Replaces eva_vit.py:125-145, Qformer.py:195-262,
 * modeling_llama.py:197-222 and their autograd. */
int mh_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const float* bias,
                const int* kv_len, int B, int H, int Sq, int Sk, int D, long q_bs, int ldq, long k_bs, int ldk,
                long v_bs, int ldv, long o_bs, int ldo, float scale, int causal, mh_stream_t s);
int mh_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                float* delta_ws /* [B,H,Sq] f32 scratch */, void* dq, void* dk, void* dv, const float* bias,
                const int* kv_len, int B, int H, int Sq, int Sk, int D, long q_bs, int ldq, long k_bs, int ldk,
                long v_bs, int ldv, long o_bs, int ldo, long do_bs, int lddo, long dq_bs, int lddq, long dk_bs,
                int lddk, long dv_bs, int lddv, float scale, int causal, mh_stream_t s);

/* K5 + K8 fused, training path: LLaMA causal self-attention with the rotary embedding applied on load
   (modeling_llama.py:109-123 apply_rotary_pos_emb, :168-231 LlamaAttention.forward).  qkv = [q | k | v] [B, S, ld] bf16,
   PRE-rotary, head h at columns h*D..; one workgroup per (batch, head) holds the whole sequence (S <= 160, D = 128:
   mh_attn_rope_supported; otherwise MH_ERR_UNSUPPORTED and the caller uses mh_rope_inplace + mh_attn_fwd/bwd).
   pos [B*S] position ids, cos/sin [max_pos, D/2] f32, kv_len optional [B] (right padding).  bwd writes
   dqkv = [dq | dk | dv] with dq, dk already un-rotated.  mh_gemm_attn_rope_bwd = the o_proj dgrad GEMM
   dO = A[M,K].Bw[N,K]^T followed by that backward, with the split-K partial sums of dO added up inside the attention
   kernel (do_buf [M, N] bf16 is scratch for the unsplit case). */
int mh_attn_rope_supported(int S, int D);
int mh_attn_rope_fwd(const void* qkv, int ld, void* o, int ldo, float* lse, const int* pos, const float* cos_tab,
                     const float* sin_tab, const int* kv_len, int B, int H, int S, int D, float scale, mh_stream_t s);
int mh_attn_rope_bwd(const void* qkv, int ld, const void* o, int ldo, const void* dout, int ldd, const float* lse,
                     void* dqkv, const int* pos, const float* cos_tab, const float* sin_tab, const int* kv_len,
                     int B, int H, int S, int D, float scale, mh_stream_t s);
int mh_gemm_attn_rope_bwd(const void* A, int lda, const void* Bw, int ldb, void* do_buf, int K, const void* qkv,
                          int ld, const void* o, int ldo, const float* lse, void* dqkv, const int* pos,
                          const float* cos_tab, const float* sin_tab, const int* kv_len, int B, int H, int S,
                          int D, float scale, mh_stream_t s);

/* K7 RMSNorm (modeling_llama.py:66-74) f32 in -> bf16 out; bwd is dgrad-only (+ optional residual-grad add,
 * optional bf16 copy of dx for the next dgrad GEMM). */
int mh_rmsnorm_fwd(const float* x, const float* w, void* y_bf16, long ldy, int M, int D, float eps, mh_stream_t s);
int mh_rmsnorm_bwd(const float* dy, const float* x, const float* w, const float* dres, float* dx, void* dx_bf16,
                   int M, int D, float eps, mh_stream_t s);
/* K6 LayerNorm (eva_vit.py:175-176; blip2.py:119-125; Qformer.py:106,288,374) f32 in -> bf16 and/or f32 out. */
int mh_layernorm_fwd(const float* x, const float* w, const float* b, void* y_bf16, float* y_f32, int M, int D,
                     float eps, mh_stream_t s);
int mh_layernorm_bwd(const float* dy, const float* x, const float* w, const float* dres, float* dx, void* dx_bf16,
                     int M, int D, float eps, mh_stream_t s);

/* K8 rotary, rotate-half form gathered by position id (modeling_llama.py:109-123); in place on heads
 * [col0, col0 + n_heads*head_dim) of a [n_tok, ld] bf16 buffer; tables [max_pos, head_dim/2] f32; sign=-1 = bwd. */
int mh_rope_inplace(void* x, int ld, int col0, int n_tok, int n_heads, int head_dim, const int* pos,
                    const float* cos_tab, const float* sin_tab, float sign, mh_stream_t s);
/* K9 SiLU-gated MLP elementwise (modeling_llama.py:139-140): gu = [M, 2I] = [gate | up] bf16. */
int mh_silu_mul_fwd(const void* gu, void* h, int M, int I, mh_stream_t s);
int mh_silu_mul_bwd(const void* dh, const void* gu, void* dgu, int M, int I, mh_stream_t s);
/* the same with gate / up interleaved in blocks of `blk` columns ([g 0..blk-1 | u 0..blk-1 | g blk.. ]; blk = 0: the halves
   layout above), and the SiLU-gated MLP fused into the GEMMs around it (modeling_llama.py:139-140): gu[M, 2I] = X Wgu^T with
   Wgu's rows interleaved in blocks of 128, act[M, I] = silu(g) * u from the gate|up GEMM's epilogue; dgu = silu_mul_bwd(dH
   WdT^T, gu) from the down projection's dgrad epilogue (dact_buf [M, I] bf16: scratch when the policy does not run the fused
   kernel).  Same bits as GEMM + mh_silu_mul_*_blk. */
int mh_silu_mul_fwd_blk(const void* gu, void* h, int M, int I, int blk, mh_stream_t s);
int mh_silu_mul_bwd_blk(const void* dh, const void* gu, void* dgu, int M, int I, int blk, mh_stream_t s);
int mh_gemm_swiglu_fwd(const void* X, int ldx, const void* Wgu, int ldw, void* gu, int ldgu, void* act, int ldact, int M,
                       int I, int K, mh_stream_t s);
int mh_gemm_swiglu_bwd(const void* dH, int lddh, const void* WdT, int ldw, const void* gu, int ldgu, void* dgu, int lddgu,
                       void* dact_buf, int M, int I, int K, mh_stream_t s);
/* erf-GELU on bf16 (Qformer.py:352-356 via ACT2FN["gelu"]) */
int mh_gelu_fwd(const void* x, void* y, long n, mh_stream_t s);
int mh_gelu_bwd(const void* dy, const void* x, void* dx, long n, mh_stream_t s);
/* the erf-GELU MLP (Qformer.py:481-484: intermediate_query -> gelu -> output_query; eva_vit.py:54-61) with the elementwise half in
   the epilogue of the GEMM next to it: pre[M, N] = X W^T + bias (bf16, kept for the backward) and act = gelu(pre) in one launch;
   dpre = bf16(dY WT^T) * gelu'(pre) in one launch.  Fused when the policy runs an unsplit tile kernel, otherwise GEMM + mh_gelu_*
   (dact_buf [M, N] bf16: scratch for that case; dense rows required then).  Same bits as the separate launches. */
int mh_gemm_gelu_fwd(const void* X, int ldx, const void* W, int ldw, const float* bias, void* pre, int ldpre, void* act, int ldact,
                     int M, int N, int K, mh_stream_t s);
int mh_gemm_gelu_bwd(const void* dY, int lddy, const void* WT, int ldw, const void* pre, int ldpre, void* dpre, int lddpre,
                     void* dact_buf, int M, int N, int K, mh_stream_t s);

/* counter-based dropout for PEFT lora_dropout (myriad.py:171-178): mask = f(seed, flat index); bwd regenerates it */
int mh_dropout_bf16(const void* x, long ldx, void* y, long ldy, long rows, int cols, float p, unsigned long long seed,
                    mh_stream_t s);
int mh_dropout_add_f32(const float* dy, long lddy, float* acc, long ldacc, long rows, int cols, float p,
                       unsigned long long seed, mh_stream_t s);

/* K10 (PEFT form) LoRA on q_proj/v_proj, y = W x + (alpha/r) B (A dropout(x)) (myriad.py:170-180).  The UP projection
 * rides the qkv GEMM as a 64-column K border; these kernels do the skinny parts.  A = [R2, D] f32 (A_q rows then A_v
 * rows, R2 = 2r in {16,32}); border / dborder = columns [D, D+R2) of the bordered operand / its gradient. */
int mh_lora_down(const void* x, long ldx, const float* A, void* border, long ldo, int M, int D, int R2, float s, float p,
                 unsigned long long seed, mh_stream_t st);
int mh_lora_dx(const float* dx_ext, long ld, const float* A, float* out, int M, int D, int R2, float s, float p,
               unsigned long long seed, mh_stream_t st);
/* Head of a single-token decode step with LoRA attached (M <= 2 rows, no dropout): x_ext[:, :D] = bf16(rmsnorm(h; w, eps)) and
 * x_ext[:, D:D+64] = its LoRA border, one launch with the bits of mh_rmsnorm_fwd + mh_lora_down (LlamaRMSNorm,
 * modeling_llama.py:66-74, in front of the peft q_proj / v_proj).  MH_ERR_UNSUPPORTED for M > 2 or D > 4096. */
int mh_rmsnorm_lora_down(const float* h, long ldh, const float* norm_w, float eps, const float* A, void* x_ext, long ldx, int M,
                         int D, int R2, float s, mh_stream_t st);
/* The qkv dgrad with the LoRA border, [M, D+64] = dqkv . [W_qkv | B_ext]^T-layout operand (myriad.py:170-180 under autograd),
 * and the LoRA dx correction that reads it, in one call: when mh_gemm_plan(M, D+64, K) splits K the correction kernel sums the
 * fp32 partial slabs itself (no reduce launch, same bits) and writes the summed border d(s*t) [M, 64] to border_out; otherwise
 * the product goes through dx_ext_buf [M, D+64] (which then holds the border) and border_out is left alone. */
int mh_gemm_lora_dx(const void* A, int lda, const void* Bw, int ldb, float* dx_ext_buf, const float* loraA, float* dxn,
                    float* border_out, int M, int D, int K, int R2, float s, float p, unsigned long long seed, mh_stream_t stream);
/* mh_gemm_lora_dx followed by the backward of the RMSNorm whose output the qkv projection read (the layer's input norm,
 * modeling_llama.py:66-74, 257-259 under autograd): dx = d rmsnorm(x; w)(dxn) + dres as f32 (dx) and / or bf16 (dx_bf16).
 * With D <= 4096 and R2 = 16 the LoRA correction and the norm backward are ONE kernel that also sums the dgrad's split-K
 * slabs (the [M, D] f32 dxn never exists); otherwise, or with option lora_norm_fused = 0, mh_lora_dx and mh_rmsnorm_bwd run
 * back to back through dxn_buf [M, D] f32.  Same bits either way.  dx_ext_buf / border_out as in mh_gemm_lora_dx. */
int mh_gemm_lora_rmsnorm_bwd(const void* A, int lda, const void* Bw, int ldb, float* dx_ext_buf, const float* loraA,
                             float* dxn_buf, float* border_out, const float* x, const float* w, const float* dres, float* dx,
                             void* dx_bf16, int M, int D, int K, int R2, float s, float p, unsigned long long seed, float eps,
                             mh_stream_t stream);
long mh_lora_wgrad_ws_floats(int D, int R2);
int mh_lora_wgrad(const void* x, long ldx, const float* dx_ext, long ldg, const void* dq, const void* dv, long ldq,
                  const void* border, long ldb, float* dA, float* dBq, float* dBv, float* ws, int M, int D, int R2,
                  float s, float p, unsigned long long seed, mh_stream_t st);
int mh_lora_refresh_border(const float* Bq, const float* Bv, void* ext, long ld_ext, void* extT, long ld_extT, int W,
                           int D, int r, mh_stream_t st);
/* The same for every layer in one launch: table = device array of n_layers x {B_q, B_v, W_ext, W_ext^T or NULL} pointers
 * (4 x 8 bytes per layer), all layers sharing the two row strides. */
int mh_lora_refresh_borders(const void* table, int n_layers, long ld_ext, long ld_extT, int W, int D, int r, mh_stream_t stream);

/* K10 rank-r adaptor y = x + (x A^T) B^T (networks.py:81-93), f32, r in {1,2,4,8}. */
int mh_lowrank_fwd(const float* x, const float* A, const float* Bm, float* y, float* t, int M, int D, int R,
                   mh_stream_t s);
long mh_lowrank_bwd_ws_floats(int M, int D, int R);
int mh_lowrank_bwd(const float* dy, const float* x, const float* t, const float* A, const float* Bm, float* dA,
                   float* dB, float* dx, float* ws, int M, int D, int R, mh_stream_t s);

/* K11 clamp-CE (modeling_llama.py:718-728) per-row loss + fused d(logits) (bf16, zero-padded to ldd). */
int mh_clamp_ce(const float* logits, long ldl, const long* labels, float* row_loss, void* dlogits_bf16, long ldd,
                int R, int V, float grad_scale, mh_stream_t s);
int mh_sum_f32(const float* x, float* out, long n, float scale, mh_stream_t s);
/* greedy step (evaluation_aqa_dataset.py:289-301 top_p=0.01 == arg-max; min_length ban of EOS) */
int mh_argmax_rows(const float* logits, long ldl, long* out, float* margin, int R, int V, int ban_id, mh_stream_t s);
/* the same plus p_max[row] = softmax(logits * inv_temp)[argmax] (banned id excluded): tells whether HF's
   `do_sample=True, top_p=p` is the arg-max (p_max >= p keeps exactly one token) -- generation_kwargs of the eval script */
int mh_argmax_pmax_rows(const float* logits, long ldl, long* out, float* margin, float* pmax, int R, int V, int ban_id,
                        float inv_temp, mh_stream_t s);
/* end of a decode step on the device: rec[3][R] f32 = (ids, margins, p_max) of this step in one small record, next_ids = ids
   (the next step's input: myriad.py:433-454's feed-back of the generated token), *step += 1 -- the token step needs no host
   input and the host fetches a step with one copy */
int mh_decode_record(const long* nxt, const float* margin, const float* pmax, float* rec, long* next_ids, int* step, int R,
                     mh_stream_t s);
/* the same plus pos[r] += 1 and kvlen[r] += 1 (the decode step's device-resident rotary position and valid-key count, one int per
   row): the step's three bookkeeping launches as one */
int mh_decode_advance(const long* nxt, const float* margin, const float* pmax, float* rec, long* next_ids, int* step, int* pos,
                      int* kvlen, int R, mh_stream_t s);

/* K12 conv stacks of VEInstructorV2 / VETokenizer (networks.py:98-127,159-189) as im2col + mh_gemm_bf16_nt. */
int mh_im2col_nhwc(const void* x, void* col, int B, int H, int W, int C, int kh, int kw, int pad, int Kpad,
                   mh_stream_t s);
int mh_col2im_nhwc(const void* dcol, float* dx, int B, int H, int W, int C, int kh, int kw, int pad, int Kpad,
                   mh_stream_t s);
int mh_relu_maxpool2_fwd(const void* y, int y_is_f32, long ldy, void* p, int B, int H, int W, int C, mh_stream_t s);
int mh_relu_maxpool2_bwd(const float* dp, const void* y, int y_is_f32, long ldy, void* dy, long lddy, int B, int H,
                         int W, int C, mh_stream_t s);
int mh_conv_pack_weight(const float* W, const float* bias, void* Wp, int Cout, int K, int Kpad, mh_stream_t s);
int mh_conv_unpack_grad(const float* dWp, float* dW, float* db, int Cout, int K, int Kpad, mh_stream_t s);

/* K13 assembly of inputs_embeds (myriad.py:354-375,395-421) and generic data movement. */
int mh_embed_gather(const void* table_bf16, const long* ids, const int* dst_rows, float* out, long n, int D, long ldo,
                    mh_stream_t s);
int mh_copy2d_f32(const float* src, long lds, float* dst, long ldd, long rows, int cols, int accumulate,
                  mh_stream_t s);
int mh_copy3d_f32(const float* src, long src_bstride, long lds, float* dst, long dst_bstride, long ldd, int nb,
                  long rows, int cols, int accumulate, mh_stream_t s);
int mh_gather_rows_f32_to_bf16(const float* src, long lds, const int* rows, void* dst, long n, int D, mh_stream_t s);
int mh_gather_rows_f32(const float* src, long lds, const int* rows, float* dst, long n, int D, mh_stream_t s);
/* The same for rows of any 2- or 4-byte element type (16-byte units: D and lds multiples of 16 / elem_bytes), and its inverse with
 * zero fill: dst[m, :] = inv[m] >= 0 ? src[inv[m], :] : 0 for every row m < M of dst (one launch instead of a fill + a scatter).
 * llama.py runs the LAST decoder layer's o_proj / MLP (modeling_llama.py:281-293) on the label-bearing rows only and expands the
 * row gradients back with these. */
int mh_gather_rows(const void* src, long lds, const int* rows, void* dst, long n, int D, int elem_bytes, mh_stream_t s);
int mh_expand_rows(const void* src, const int* inv, void* dst, long ldd, long M, int D, int elem_bytes, mh_stream_t s);
int mh_copy3d_bf16(const void* src, long src_bstride, long lds, void* dst, long dst_bstride, long ldd, int nb,
                   long rows, int cols, mh_stream_t s);
/* KV-cache append at a device-resident position + device counter bump: lets one decode step be captured in a
 * hipGraph and replayed per token (modeling_llama.py:190-195 concatenates; the cache is written in place). */
int mh_kv_append_bf16(const void* src, long ld_src, void* cache, long cache_bstride, long ld_cache, const int* pos_dev,
                      int B, int cols, mh_stream_t s);
/* One decode token per batch row: rotary on q (in place) and on k, then k|v written into cache row pos_dev[0]
 * (modeling_llama.py:186-195: apply_rotary_pos_emb + cat with past_key_value) -- rope + append in one launch.
 * qkv [B, ld] bf16 = [q | k | v] with W = n_heads*head_dim columns each; cache row layout [k | v]. */
int mh_rope_kv_append(void* qkv, long ld, int n_heads, int head_dim, const int* pos, const float* cos_tab,
                      const float* sin_tab, void* cache, long cache_bstride, long ld_cache, const int* pos_dev, int B,
                      mh_stream_t s);
int mh_add_i32(int* x, int n, int delta, mh_stream_t s);
/* Decode weight layout (modeling_llama.py:184-231 with the KV cache: every token multiplies <= 16 rows by every frozen
 * Linear).  mh_gemv_pack writes a stream-ordered copy of W [N, K] bf16 (mh_gemv_pack_elems(N, K) elements, -1 on bad
 * dims): the 16 B lane (lr, lg) of wave w reads at step t sit at ((block*NW + w)*per + t)*2 KiB + h*1 KiB + lane*16, so
 * the skinny-M kernel reads each KiB contiguously.  mh_gemv_packed = mh_gemm_bf16_nt for M <= 16 on that copy, bit-identical. */
long mh_gemv_pack_elems(int N, int K);
int mh_gemv_pack(const void* W, int ldb, int N, int K, void* out, mh_stream_t s);
int mh_gemv_packed(const void* A, int lda, const void* P, void* C, int ldc, int M, int N, int K, const float* bias,
                   const float* residual, int ldr, int out_f32, float alpha, mh_stream_t s);
/* The decode step's Linear with the producer of its operand fused in (bit-identical to the two-launch forms): the RMSNorm
 * in front of q/k/v, gate|up and lm_head (modeling_llama.py:66-74; H [M, K] f32) and the SiLU gate in front of the down
 * projection (:139-140; gu [M, 2K] bf16, gate / up interleaved in blocks of 128).  Every workgroup rebuilds the <= 16
 * operand rows in LDS; MH_ERR_UNSUPPORTED when M * K * 2 bytes exceed 64 KiB (use the two-launch form). */
int mh_gemv_packed_rmsnorm(const float* H, long ldh, const float* norm_w, float eps, const void* P, void* C, int ldc, int M,
                           int N, int K, const float* bias, const float* residual, int ldr, int out_f32, float alpha,
                           mh_stream_t s);
int mh_gemv_packed_silu(const void* gu, long ldgu, const void* P, void* C, int ldc, int M, int N, int K, const float* bias,
                        const float* residual, int ldr, int out_f32, float alpha, mh_stream_t s);
/* One decode token of attention (modeling_llama.py:186-222 with the KV cache): rotary on q / k, k | v appended at cache row
 * pos_dev[0], the one query against kv_len[b] keys -- mh_rope_kv_append + mh_attn_fwd(Sq = 1) in one launch, same bits.
 * qkv [B, ld_qkv] bf16 = [q | k | v] (q rotated in place), cache [B][T_cap][2 H D] rows [k | v], out [B, H D] bf16. */
int mh_attn_decode_rope(void* qkv, long ld_qkv, void* cache, long cache_bstride, long ld_cache, const int* pos,
                        const int* pos_dev, const int* kv_len, const float* cos_tab, const float* sin_tab, void* out, long ldo,
                        int B, int H, int D, int T_cap, float scale, mh_stream_t s);
/* K14 patch embedding operand (eva_vit.py:196-204): NCHW f32 image -> [B*np, Kpad] bf16 in (c,iy,ix) order */
int mh_patchify_nchw(const float* img, void* out, int B, int C, int H, int W, int P, int Kpad, mh_stream_t s);
int mh_scatter_rows_f32(const float* src, const int* rows, float* dst, long ldd, long n, int D, int accumulate,
                        mh_stream_t s);
int mh_cast_f32_to_bf16(const float* x, void* y, long n, mh_stream_t s);
int mh_cast_bf16_to_f32(const void* x, float* y, long n, mh_stream_t s);
int mh_transpose_to_bf16(const void* in, int in_is_f32, long ldi, void* out, long ldo, int R, int C, mh_stream_t s);
int mh_colsum_f32(const float* in, long ld, float* out, long R, int C, mh_stream_t s);
int mh_scale_f32(float* x, float a, long n, mh_stream_t s);

/* K15 AdamW (runner_base.py:104-139) on a flat f32 buffer with optional bf16 shadow; step is 1-based. */
int mh_adamw_step(float* p, const float* g, float* m, float* v, void* shadow_bf16, long n, double lr, double beta1,
                  double beta2, double eps, double weight_decay, int step, double grad_scale, mh_stream_t s);

/* K18 data path: NSA / CutPaste self-supervised anomaly augmentation on uint8 crops resident in HBM --
   minigpt4/datasets/self_sup_tasks.py:254-268 (the 'swap' and 'uniform' blends) and :97-113 (the label).  The random patch
   geometry is sampled on the host exactly as the reference samples it (myriad_amd/self_sup.py); ops is a device array of
   56-byte records {int image, y0, x0, h, w, sy, sx, mode; long mask_off; double factor; long patch_off} -- mode & 3: 0 swap,
   1 uniform, 2 union mask only (a Poisson patch); mode & 4: the source pixels are the resampled patch [h][w][3] at patch_off
   of `patches` (may be NULL when no record sets the bit) -- masks the byte pool the records index, hs / ws HOST copies of
   h, w.  out starts as a copy of dest; union_mask [B,H,W] u8 zeroed.
   mh_patch_label: mode 0 binary, 1 continuous (factor [B] f64), 2 intensity, 3 logistic-intensity (k, x0). */
int mh_patch_blend_u8(void* out, const void* src, const void* masks, const void* ops, const int* hs, const int* ws,
                      int n_ops, int B, int H, int W, void* union_mask, const void* patches, mh_stream_t s);
int mh_patch_label(const void* dest, const void* out, const void* union_mask, void* sums_ws, void* lm_ws, float* label,
                   const double* factor, int B, int H, int W, int mode, int tol, double k, double x0, mh_stream_t s);
/* The two OpenCV steps of the shipped augmentation recipe (self_sup_tasks.py:213-227 cv2.resize, :269-288 cv2.seamlessClone
   with NORMAL_CLONE; anomaly_detection.py:118-141), from their published algorithms -- PARITY UNPINNED, OpenCV is absent.
   mh_patch_resize_u8: 8-bit INTER_LINEAR of the box (sy, sx, sh, sw) of src[image] to out [h][w][3]; xi / yi [w] / [h] left
   source indices, xw / yw [w][2] / [h][2] 11-bit weights (device ints; myriad_amd/self_sup.linear_resize_tables); an exact
   2 x 2 decimation is the rounded mean of 4 (tables may be NULL).
   mh_patch_poisson_u8: gradient-domain cloning of patch [hp][wp][3] into out[image] in place.  pms [hp][wp] the clone mask
   (border cleared), eroded [h][w] its ROI after erode(3x3) x 3, ROI h x w at (y0s, x0s) of the patch and (dy0, dx0) of the
   image; Sh [h-2][h-2], Sw [w-2][w-2] sine matrices, cy / cx their 2 cos terms (f64); ws mh_patch_poisson_ws_doubles(h, w)
   doubles.  float64 throughout, interior = floor(clamp(u, 0, 255) + 1e-6).  mixed = 0: cv2.NORMAL_CLONE; 1: cv2.MIXED_CLONE
   (self_sup_tasks.py:22,47,267: the patch's gradient pair is kept where |Px - Py| > |Dx - Dy|, else the destination's). */
int mh_patch_resize_u8(const void* src, int image, int H, int W, int sy, int sx, int sh, int sw, const int* xi, const int* xw,
                       const int* yi, const int* yw, void* out, int h, int w, mh_stream_t s);
long mh_patch_poisson_ws_doubles(int h, int w);
int mh_patch_poisson_u8(void* out, int image, int H, int W, const void* patch, int hp, int wp, const void* pms,
                        const void* eroded, int y0s, int x0s, int dy0, int dx0, int h, int w, const double* Sh, const double* cy,
                        const double* Sw, const double* cx, double* ws, int mixed, mh_stream_t s);

/* Gated AdamW: torch.optim.AdamW skips parameters whose .grad is None -- a module no rank used this step (random prompt
   stage, myriad.py:378; DDP find_unused_parameters, runner_base.py:96-98) keeps its parameters, moments and step count.
   used/steps are DEVICE scalars (use count summed over ranks by the gradient all-reduce; updates applied so far):
   mh_adamw_gated updates one contiguous range iff *used > 0 with step = *steps + 1 (bias corrections in double);
   mh_adamw_bump(used[n], steps[n]) then advances the counters of the used modules. */
int mh_adamw_gated(float* p, const float* g, float* m, float* v, long n, double lr, double beta1, double beta2, double eps,
                   double weight_decay, double grad_scale, const float* used, const int* steps, mh_stream_t s);
int mh_adamw_bump(const float* used, int* steps, int n, mh_stream_t s);

/* K16 anomaly-map heads of the vision expert (adrefexpert_v2.py:243-301).  All f32.
 * l2norm_rows: y = x / max(||x||, eps) (bf16 and/or f32 out) -- operands of the cosine similarities (:259, :283);
 * pair_logits: out[row] = scale * [<p,t0>, <p,t1>] / ||p|| against the per-sample [normal, abnormal] text pair (:283-284);
 * zs_accumulate: one tap of the zero-shot branch: mask_acc[B,h,h] += w*softmax(pair)[1], map_acc[B,S,S] += w*softmax(
 *   bilinear(align_corners=True) upsampled pair)[1] (:285-296);
 * rowmax_skip: acc[row] += w * max over columns c with c % period != 0 (best reference patch, class-token columns
 *   skipped; :260-261);  bilinear_ac: align_corners=True resize, optionally 1 - x (:265-268). */
int mh_l2norm_rows(const float* x, long ldx, void* y_bf16, float* y_f32, long ldy, int M, int D, float eps, mh_stream_t s);
int mh_pair_logits(const float* p, long ldp, const float* text, float* out, long rows, int rows_per_batch, int C, float scale,
                   mh_stream_t s);
int mh_zs_accumulate(const float* logits, float* mask_acc, float* map_acc, int B, int h, int S, float w, mh_stream_t s);
int mh_rowmax_skip(const float* scores, long lds, float* acc, long rows, int cols, int period, float w, mh_stream_t s);
int mh_bilinear_ac(const float* in, float* out, int B, int h, int w, int H, int W, int one_minus, mh_stream_t s);

/* ---- opaque context: library state the caller owns explicitly ------------------------------------------------
 * An mh_ctx holds (1) a split-K scratch record -- mh_ctx_set_workspace(ctx, NULL, ptr, bytes) the main scratch,
 * (ctx, stream, ptr, bytes) that of one more stream that may split K concurrently; library calls use the record of the context
 * made current by mh_ctx_make_current (NULL = the process default that mh_set_workspace fills) -- and (2) the gradient
 * exchange of the data-parallel step (runner_base.py:94-98): an RCCL communicator, a side HIP stream and two events.
 * mh_ctx_comm_id writes the 128-byte id rank 0 generates; every rank passes it to mh_ctx_comm_init(ctx, id, rank, world).
 * mh_allreduce_start(ctx, buf, n, producer): in-place sum of buf[0..n) f32 over the ranks on the context's side stream, ordered
 * after what `producer` has queued; returns at once.  mh_allreduce_wait(ctx, consumer): `consumer` waits on the device for it.
 * RCCL is dlopen'ed by the comm calls only (MH_ERR_UNSUPPORTED if it cannot be loaded); world == 1 needs no communicator. */
typedef struct mh_ctx mh_ctx;
int mh_ctx_create(mh_ctx** out);
int mh_ctx_destroy(mh_ctx* ctx);
int mh_ctx_set_workspace(mh_ctx* ctx, mh_stream_t stream, void* ptr, long bytes);
int mh_ctx_make_current(mh_ctx* ctx);
int mh_ctx_comm_id(void* id128);
int mh_ctx_comm_init(mh_ctx* ctx, const void* id128, int rank, int world);
int mh_ctx_world(const mh_ctx* ctx);
int mh_allreduce_start(mh_ctx* ctx, float* buf, long n, mh_stream_t producer);
/* Round 4: wire type per call (MH_DT_F32 / MH_DT_BF16 elements) and the two halves of the sharded exchange
 * (runner.DataParallel mode 'rs_ag': reduce-scatter the gradients, AdamW on the own 1/world shard, all-gather the parameters).
 * mh_reduce_scatter_start: recv[0..n_per_rank) = sum over ranks of their send[rank * n_per_rank ..); send holds world *
 * n_per_rank elements.  mh_allgather_start: recv[r * n_per_rank ..) = rank r's send[0..n_per_rank).  Verbs started back to
 * back queue on the context's side stream; one mh_allreduce_wait covers everything started so far. */
#define MH_DT_F32 0
#define MH_DT_BF16 1
int mh_allreduce_start_dt(mh_ctx* ctx, void* buf, long n, int dt, mh_stream_t producer);
int mh_reduce_scatter_start(mh_ctx* ctx, const void* send, void* recv, long n_per_rank, int dt, mh_stream_t producer);
int mh_allgather_start(mh_ctx* ctx, const void* send, void* recv, long n_per_rank, int dt, mh_stream_t producer);
int mh_allreduce_wait(mh_ctx* ctx, mh_stream_t consumer);

/* library identity */
const char* mh_version(void);
int mh_target_arch(void); /* 950 */

/* Options: process-wide on/off switches between code paths that compute the same thing (the fused and the separate form of
 * an op are bit-identical; tests flip them to prove it) or between stated precisions.  Each starts from its MYRIAD_<NAME>
 * environment variable when set ("0" = off), else from the default:
 *   slab_bf16 (1)       bf16 split-K slabs of the 256x256 kernel; 0 = fp32 slabs
 *   gemm_skinny (1)     160-row tiles for 128 < M <= 320
 *   swiglu_fused (1)    SiLU gate in the epilogues of the gate|up forward / down dgrad GEMMs
 *   gelu_fused (1)      erf-GELU in the epilogues of the Q-Former MLP GEMMs
 *   attn_bwd_split (1)  two workgroups per (batch, head) in mh_attn_rope_bwd when B*H <= 128
 *   gemm_zero_pad (1)   rows past M / N of a 256x256 tile read as zeros; 0 = as copies of the last row
 *   gemm256_impl (1)    plan kernel 2 = gemm_x8_kernel; 0 = gemm_256_kernel
 *   lora_norm_fused (1) LoRA dx + input-norm backward as one kernel (mh_gemm_lora_rmsnorm_bwd); 0 = the two launches
 *   attn_full (1)       mh_attn_fwd without mask / bias and with Sk <= 288, head dim in (32, 96]: one workgroup stages the whole
 *                       K and V of a (batch, head) (attn_full.hip) instead of 64x64 tiles; 0 = the tiled kernel
 *   lora_wgrad_mfma (1) mh_lora_wgrad at r = 8, D % 128 == 0: the sums over token rows as MFMA products (per-row scalars as a bf16 head +
 *                       bf16 remainder, fp32 accumulation); 0 = the thread-per-column fp32 kernel
 *   gemm_skip_pad (1)   256x256 kernel: a wave whose rows lie past M but for at most two 16-row fragments issues no MFMAs for the others
 *                       (same results; the chip's clock under this loop is set by the matrix pipes' power); 0 = every wave runs the full loop
 *   gemm_split_xcd (1)  256x256 kernel, K-split launches: the workgroups of one XCD (round-robin dispatch, private L2) own ONE K split of a
 *                       band of tile columns, so each L2 streams its A panels' K range once; 0 = every XCD runs all splits of a tile
 *                       block and reads the whole A matrix (same results, the mapping only moves workgroups)
 * mh_set_option returns the previous value (0 / 1) or MH_ERR_ARG (unknown name, value not 0 / 1); mh_get_option the current
 * value or MH_ERR_ARG.  Not thread-safe against concurrent launches. */
int mh_set_option(const char* name, int value);
int mh_get_option(const char* name);
/* self-check of the hardware fp32 -> bf16 rounding (v_cvt_pk_bf16_f32) against the integer round-to-nearest-even form:
   hw / hw_pk / sw [n] bf16 = scalar conversion, packed conversion, integer form of x[n] (n even) */
int mh_bf16_round_check(const float* x, void* hw, void* hw_pk, void* sw, long n, mh_stream_t s);

/* K17 image front-end (data path, SURVEY 8 f-2 image side): what datasets/datasets/anomaly_detection.py:118-122,246 and
 * processors/blip_processors.py:21-29,120-147,189-203 do on the CPU with torchvision + Pillow -- Resize(BICUBIC) ->
 * CenterCrop -> ToTensor -> Normalize -- for one uint8 RGB image resident in HBM, bit-exact with Pillow's 8-bit resampler.
 * kh/bh, kv/bv: 22-bit fixed-point weights [res][ksz] and (first, count) bounds [res][2] of the full resize per axis (device
 * ints, built by the host exactly as Resample.c builds them); (crop_y0, crop_x0, out_h, out_w) the kept window; (y0, rows)
 * the input rows that window taps; tmp rows*out_w*3 bytes; lut [3][256] float = ((v/255) - mean) / std in float32.
 * out [3][out_h][out_w] f32 and/or u8_out [out_h][out_w][3] (the uint8 crop the NSA augmentation edits). */
int mh_image_resize_crop_norm(const void* img, int H, int W, long row_stride, const int* kh, const int* bh, int ksz_h,
                              const int* kv, const int* bv, int ksz_v, int crop_y0, int crop_x0, int out_h, int out_w, int y0,
                              int rows, void* tmp, const float* lut, float* out, void* u8_out, mh_stream_t s);
int mh_image_u8_normalize(const void* u8_hwc, long n_pix, const float* lut, float* out, mh_stream_t s);

#ifdef __cplusplus
}
#endif
#endif
