/*
 * autoprog_hip.h -- C ABI of libautoprog_hip.so: the MI355X (gfx950) kernels behind the
 * AutoProg VOLO/DeiT training hot path.
 *
 * The reference (changlin31/AutoProg) has no FFI/plugin ABI: its hot path is Python
 * nn.Modules calling ATen ops.  This header is therefore
 * build-defined (row B2); each entry point names the reference call site whose device
 * arithmetic it replaces (file:line relative to the reference tree).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer borrowed for the call; nothing is allocated,
 *     freed or synchronised inside; work is enqueued on `stream` (a hipStream_t)
 *   - bf16 tensors are passed as uint16_t*, row-major, leading dimension in ELEMENTS and a
 *     multiple of 8 (16-byte rows); fp32 statistics / parameters / gradients are float*
 *   - return 0 on success, a negative AP_ERR_* otherwise (no exceptions cross the ABI)
 *   - entry points are re-entrant and thread-compatible; the only process-wide state is a handful of tuning switches
 *     (AP_* environment variables, each read once into a function-local static) and one-time hipFuncSetAttribute calls
 */
#ifndef AUTOPROG_HIP_H
#define AUTOPROG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* ap_stream_t;          /* hipStream_t */
typedef uint16_t ap_bf16;

enum {
    AP_OK = 0,
    AP_ERR_SHAPE = -1,        /* a size/stride violates the documented constraints */
    AP_ERR_UNSUPPORTED = -2,  /* valid request outside what the kernels implement  */
    AP_ERR_LAUNCH = -3,       /* hipGetLastError() != hipSuccess after a launch    */
    AP_ERR_NULL = -4          /* a required pointer is NULL                         */
};

int ap_abi_version(void);
const char* ap_error_string(int code);

/* ---- calibration probes of the device a measurement runs on (no reference call site: bench.py `calibration`, next to the timing
 * loop of main_prog.py:1007-1008,1061-1064).  ap_calib_copy: dst[0:bytes) = src[0:bytes) with 16 bytes per lane (bytes % 16 == 0);
 * ap_calib_mfma: 256 workgroups x 4 waves, each `iters` trips of 16 independent v_mfma_f32_16x16x32_bf16 on operands read once from
 * seed (>= 2048 bf16) -- 256 * 4 * iters * 16 * 16384 FLOP; sink receives nothing unless a sum hits a sentinel (keeps the loop alive) */
int ap_calib_copy(const void* src, void* dst, int64_t bytes, ap_stream_t stream);
int ap_calib_mfma(const ap_bf16* seed, float* sink, int iters, ap_stream_t stream);

/* ---- precision plumbing (apex O1 casts, main_prog.py:491: fp32 master -> 16-bit) ------- */
int ap_cast_f32_bf16(const float* src, ap_bf16* dst, int64_t n, ap_stream_t stream);
int ap_cast_bf16_f32(const ap_bf16* src, float* dst, int64_t n, ap_stream_t stream);
/* dst[c*ld_dst + r] = bf16(src[r*cols + c]); columns rows..ld_dst-1 of dst are zeroed */
int ap_cast_transpose_f32_bf16(const float* src, ap_bf16* dst, int rows, int cols, int ld_dst, ap_stream_t stream);

/* ---- per-step input resize of progressive / supernet training (main_prog.py:973-974,1908-1910:
 * F.interpolate(input, size=(r,r), mode='bilinear', align_corners=False)) fused with the stem's NCHW fp32 -> NHWC bf16 conversion:
 * x [B,C,Hi,Wi] fp32 -> y [B,Ho,Wo,C] bf16.  Ho = Hi, Wo = Wi is the plain layout change + cast. */
int ap_resize_bilinear_nhwc(const float* x, ap_bf16* y, int B, int C, int Hi, int Wi, int Ho, int Wo, ap_stream_t stream);

/* ---- DropPath masks of a whole forward pass (timm DropPath behind models/volo.py:128,218: mask = floor(keep + U), y = x * mask / keep):
 * uniform [sites,B] in [0,1), keep [sites] -> factor [sites,B] = mask/keep, mask [sites,B] in {0,1}, and (tokens > 0) token_mask
 * [sites, token_row] bf16 with token_mask[s, b*tokens + t] = mask[s,b], zero padding up to token_row (a multiple of 8 >= B*tokens) */
int ap_droppath_masks(const float* uniform, const float* keep, float* factor, float* mask, ap_bf16* token_mask, int sites, int B,
                      int tokens, int token_row, ap_stream_t stream);

/* ---- LayerNorm (nn.LayerNorm: models/volo.py:122,131,213,221,290,297,550) ------------- */
int ap_layernorm_fwd(const ap_bf16* x, const float* gamma, const float* beta, ap_bf16* y,
                     float* mean, float* rstd, int64_t rows, int C, float eps, ap_stream_t stream);
/* the same with a second copy of the output as OCP e4m3 bytes for an fp8 GEMM (configs[4]): y8 = sat(y * q_scale[0]),
 * q_amax[0] = max(q_amax[0], max |y|) (nullable) -- what ap_quantize_fp8 would produce from y, without its pass over y */
int ap_layernorm_fwd_fp8(const ap_bf16* x, const float* gamma, const float* beta, ap_bf16* y, unsigned char* y8, const float* q_scale,
                         float* q_amax, float* mean, float* rstd, int64_t rows, int C, float eps, ap_stream_t stream);
/* dx = dres + d(LN)/dx ; dgamma/dbeta are ACCUMULATED (+=).  `workspace` (device, caller-owned,
 * >= ap_layernorm_bwd_workspace() bytes) holds per-workgroup partial sums for the deterministic
 * two-pass column reduction. */
size_t ap_layernorm_bwd_workspace(int64_t rows, int C);
int ap_layernorm_bwd(const ap_bf16* dy, const ap_bf16* x, const float* gamma, const float* mean,
                     const float* rstd, const ap_bf16* dres /*nullable*/, ap_bf16* dx,
                     float* dgamma, float* dbeta, int64_t rows, int C,
                     void* workspace, size_t ws_bytes, ap_stream_t stream);
/* the same without the dgamma / dbeta reduction: the per-workgroup partial rows stay in `workspace` (which must live until the
 * batched reduction ran), *n_partial receives their count; ap_layernorm_bwd_reduce_batched then reduces up to AP_LN_MAX_BATCH
 * LayerNorms (all of one block) in ONE launch: dgamma / dbeta += column sums of the partial rows */
#define AP_LN_MAX_BATCH 12
typedef struct ap_ln_reduce { const float* partial; int n_partial; int C; float* dgamma; float* dbeta; } ap_ln_reduce;
int ap_layernorm_bwd_partial(const ap_bf16* dy, const ap_bf16* x, const float* gamma, const float* mean, const float* rstd,
                             const ap_bf16* dres, ap_bf16* dx, int64_t rows, int C, void* workspace, size_t ws_bytes, int* n_partial,
                             ap_stream_t stream);
/* (ABI version 6) the same for the LayerNorm whose output also feeds a 2 x 2 ceil-mode average pool (OutlookAttention, models/volo.py:75,87):
 * rows = B*H*W tokens of a [B,H,W,C] grid, and the incoming gradient of token (b, y, x) is dy + pool_grad[b, y/2, x/2] / count (count = pixels
 * under the pooled cell) -- the pool's backward rides in the kernel instead of a pass of its own over dy (ap_avgpool2_bwd_acc).
 * AP_ERR_UNSUPPORTED where the pipelined one-chunk-per-lane kernel does not apply (C > 512): call ap_avgpool2_bwd_acc + the plain form. */
int ap_layernorm_bwd_partial_pool(const ap_bf16* dy, const ap_bf16* pool_grad, int B, int H, int W, const ap_bf16* x, const float* gamma,
                                  const float* mean, const float* rstd, const ap_bf16* dres, ap_bf16* dx, int C, void* workspace, size_t ws_bytes,
                                  int* n_partial, ap_stream_t stream);
int ap_layernorm_bwd_reduce_batched(const ap_ln_reduce* items, int count, ap_stream_t stream);

/* ---- Linear layers (nn.Linear: models/volo.py:67,68,71,156,158,180,182,253,256,258,547,553)
 * C[M,N] = epilogue( A[M,K] . B[N,K]^T )   bf16 in, fp32 MFMA accumulate, bf16 out
 * epilogue order: +bias[n]; GELU (optionally also storing the pre-activation); * gelu'(h[m,n]);
 * * row_scale[m / rows_per_scale] (DropPath, timm); + residual[m,n]                         */
/* Patch addressing: a k x k / stride k convolution on an NHWC feature map [B,H,W,C] (PatchEmbed.proj, Downsample:
 * models/volo.py:368-372,383-396) is a GEMM whose row m = (b, i, j) and column kk = (dy, dx, c) live at element offset
 *   (m / group) * group_stride + (m % group) * row_stride + (kk / kseg) * kseg_stride + (kk % kseg)
 * with group = W/k output columns, group_stride = k*W*C, row_stride = k*C, kseg = k*C, kseg_stride = W*C (H % k == 0). */
typedef struct ap_patch_map { int group, group_stride, row_stride, kseg, kseg_stride; } ap_patch_map;
/* side 1: C[M, ld] = A_patches[M, K] . B[N, K]^T + bias   (forward; K % 64 == 0, kseg % 64 == 0)
 * side 2: C_patches[M, N] = A[M, ld >= K] . B[N, K]^T        (input gradient written in place of the feature map; bias NULL) */
int ap_gemm_nt_patch(const ap_bf16* A, const ap_bf16* B, int ldb, ap_bf16* C, int ld, int M, int N, int K,
                     const float* bias, const ap_patch_map* map, int side, ap_stream_t stream);
/* (ABI version 6) side = 1 with a_bn: the A rows are relu(bn(.)) of the 64-channel NHWC map that is read (the activation between the last
 * stem convolution and PatchEmbed.proj, models/volo.py:364-372, is never materialised); a_bn NULL: ap_gemm_nt_patch */
struct ap_bn_input;
int ap_gemm_nt_patch_bn(const ap_bf16* A, const struct ap_bn_input* a_bn, const ap_bf16* B, int ldb, ap_bf16* C, int ld, int M, int N, int K,
                        const float* bias, const ap_patch_map* map, int side, ap_stream_t stream);

typedef struct ap_gemm_epilogue {
    const float* bias;          /* [N] or NULL */
    int gelu;                   /* 1: out = gelu_erf(v) (models/volo.py:157); 2: the same, and preact_out receives gelu'(v) instead of v;
                                 * 3 (ABI version 6): as 2, but preact_out is an UNSIGNED CHAR [M, ldc] tensor of 8-bit fixed-point codes
                                 * code = clamp(rint(202 * gelu'(v)) + 26, 0, 255), i.e. gelu' = (code - 26) / 202 on [-0.1287, 1.1337] (gelu' lives in
                                 * [-0.1290, 1.1290]; 0, 1/2 and 1 are exact codes): half the bytes of the bf16 derivative, |error| <= 1/404 */
    ap_bf16* preact_out;        /* with gelu: also store v (the pre-activation; gelu = 2: its activation derivative) here, ld = ldc */
    const ap_bf16* dgelu_of;    /* out = v * gelu'(dgelu_of[m,n]) (backward of the above), ld = ldc */
    const float* row_scale;     /* [ceil(M/rows_per_scale)] or NULL */
    int rows_per_scale;
    const ap_bf16* residual;    /* [M,N] with leading dimension ldr, or NULL */
    int ldr;
    const ap_bf16* mul_by;      /* out = v * mul_by[m,n] (ld = ldc), applied where dgelu_of is: the backward of gelu = 2, whose forward stored
                                 * gelu'(h) -- the only thing the backward needs of h (autograd of models/volo.py:157) -- so that it is a
                                 * multiplication instead of ~18 instructions per element; not together with dgelu_of.  (ABI version 3) */
    const unsigned char* mul_by8; /* (ABI version 6) as mul_by with the 8-bit codes a gelu = 3 forward stored: out = v * (mul_by8[m,n] - 26) / 202, ld = ldc
                                 * bytes; not together with mul_by / dgelu_of */
    unsigned char* q8_out;      /* ap_gemm_nt_fp8 with gelu (ABI version 4), ap_gemm_nt with mul_by8 and nothing else but row_scale (version 6:   */
                                /* the input gradient of fc2, operand of fc1's fp8 input-gradient product; AP_ERR_UNSUPPORTED for any other launch */
                                /* or one the 8-phase kernel does not take): the output a second time as OCP e4m3 bytes [M, ldc] --             */
    const float* q8_scale;      /* q8_out = sat(out * q8_scale[0]), q8_amax[0] = max(q8_amax[0], max |out|) (nullable) -- the operand of the */
    float* q8_amax;             /* fp8 GEMM that consumes this activation, without a quantisation pass.  Launches of the 8-phase kernel only */
                                /* (M >= 4096, K % 128 == 0, N as for ap_gemm_nt): AP_ERR_UNSUPPORTED otherwise -- also from ap_gemm_nt, which  */
                                /* since ABI version 6 takes q8_out on its mul_by8 launches and refuses it on any launch whose kernel cannot     */
                                /* write it (a forced tile variant included)                                                                     */
} ap_gemm_epilogue;
int ap_gemm_nt(const ap_bf16* A, int lda, const ap_bf16* B, int ldb, ap_bf16* C, int ldc,
               int M, int N, int K, const ap_gemm_epilogue* epi, ap_stream_t stream);
/* ---- (ABI version 7) the MLP of a block in ONE launch: Mlp.forward, models/volo.py:147-167 (fc1 -> GELU -> fc2) with the residual add and
 * DropPath scale of its call sites (:143 Outlooker, :233 Transformer), and -- the same kernel, backward = 1 -- the two input-gradient products
 * of its backward pass.  The hidden activation never makes the trip to memory and back between the two products; it still LEAVES (the weight
 * gradients read it).  Rounding points and K order are those of the two ap_gemm_nt launches it replaces: results are bit-identical to
 *     forward : a = ap_gemm_nt(x, wa, bias1, gelu = 3 -> codes, row_scale_hidden);  out = ap_gemm_nt(a, wb, bias2, row_scale_out, residual)
 *     backward: dh = ap_gemm_nt(x = dL/dout, wa = fc2.weight^T copy, mul_by8 = codes, row_scale_hidden);  out = ap_gemm_nt(dh, wb = fc1.weight^T copy)
 * Built for c = 384, hidden = 3 c, m % 128 == 0 (VOLO-D1's transformer stages at any batch of 128-row blocks): AP_ERR_UNSUPPORTED otherwise,
 * and for a forward launch while the GELU table cannot be had (AP_GELU_TABLE=0; a stream capture in front of its first build) -- the caller
 * then issues the two launches. */
typedef struct ap_mlp_fused_args {
    const ap_bf16* x; int ldx;               /* [m, c] */
    const ap_bf16* wa; int ldwa;             /* [hidden, c] */
    const ap_bf16* wb; int ldwb;             /* [c, hidden] */
    ap_bf16* out; int ldo;                   /* [m, c] */
    ap_bf16* hidden_out; int ldh;            /* [m, hidden]: forward gelu(h) * row_scale_hidden, backward dL/dh */
    unsigned char* codes;                    /* [m, hidden] bytes, row stride ldh: 8-bit gelu' codes (ap_gemm_epilogue.gelu = 3) -- written by the
                                              * forward, read by the backward */
    const float* bias1; const float* bias2;  /* forward: [hidden], [c] (nullable); backward: NULL */
    const float* row_scale_hidden;           /* [ceil(m / rows_per_scale)] or NULL: forward the 0/1 DropPath mask on a, backward mask / keep on dL/dh */
    const float* row_scale_out;              /* forward: mask / keep on the branch output (before the residual); backward: NULL */
    int rows_per_scale;
    const ap_bf16* residual; int ldr;        /* forward: [m, c] or NULL; backward: NULL */
    int m, c, hidden;
    int backward;
    /* forward, optional: the LayerNorm in front of fc1 (Transformer.forward, models/volo.py:233 `self.mlp(self.norm2(x))`) inside the same
     * launch.  ln_in non-NULL: x is ignored, the rows are read from ln_in [m, c], normalised with ln_gamma / ln_beta / ln_eps (bit-identical to
     * ap_layernorm_fwd), written to ln_out [m, c] (the fc1 weight gradient reads them) and their statistics to ln_mean / ln_rstd [m] */
    const ap_bf16* ln_in; int ld_ln;
    ap_bf16* ln_out; int ld_lno;
    const float* ln_gamma; const float* ln_beta; float ln_eps;
    float* ln_mean; float* ln_rstd;
} ap_mlp_fused_args;
int ap_mlp_fused(const ap_mlp_fused_args* args, ap_stream_t stream);

/* ---- fp8 forward GEMM (BASELINE configs[4] "mixed MFMA fp8 GEMM"): OCP e4m3 operands, fp32 accumulation, bf16 output.
 * y = sat(x * scale[0]) -> e4m3, n % 16 == 0; amax (nullable): amax[0] = max(amax[0], max |x|) for the next step's scale */
int ap_quantize_fp8(const ap_bf16* x, unsigned char* y, int64_t n, const float* scale, float* amax, ap_stream_t stream);
/* the same for several tensors in ONE launch (the Linear weights of a model, once per optimizer step): `jobs_device` is a DEVICE array;
 * job j is scaled by scales[slot] and raises amax[slot] (n % 16 == 0) */
typedef struct ap_fp8_job { const ap_bf16* x; unsigned char* y; int64_t n; int slot; int pad_; } ap_fp8_job;
int ap_quantize_fp8_multi(const ap_fp8_job* jobs_device, int njobs, const float* scales, float* amax, ap_stream_t stream);

/* ---- test aid: fill all 160 KB of LDS of every CU with `pattern` (one workgroup per CU; scratch2: 8 bytes of device memory).  A kernel
 * that reads LDS it never wrote (padded rows of a tile) then sees NaNs instead of the previous kernel's leftovers: tests/test_gpu_kernels.py */
int ap_debug_poison_lds(unsigned pattern, unsigned* scratch2, ap_stream_t stream);
/* C = epi(dq_a[0] * dq_b[0] * A8 . B8^T): A8 [M,K], B8 [N,K] e4m3 bytes (K, lda, ldb multiples of 16), dq_* device scalars (1/scale);
 * same epilogue as ap_gemm_nt */
int ap_gemm_nt_fp8(const unsigned char* A, int lda, const unsigned char* B, int ldb, ap_bf16* C, int ldc, int M, int N, int K,
                   const float* dq_a, const float* dq_b, const struct ap_gemm_epilogue* epi, ap_stream_t stream);
/* weight gradient: C[N1,N2] += A[M,N1]^T . B[M,N2]   (fp32 accumulate into C, atomics);
 * optional fused bias gradient: colsum_A[n] += sum_m A[m,n] (NULL to skip) */
int ap_gemm_tn_acc(const ap_bf16* A, int lda, const ap_bf16* B, int ldb, float* C, int ldc,
                   int M, int N1, int N2, float* colsum_A, ap_stream_t stream);
/* the same for up to AP_TN_MAX_GROUP problems in ONE launch (all Linear layers of a block, or of several blocks: the
 * reference's autograd issues one addmm per layer, models/volo.py:67-71,156-158,180-182): the launch's workgroups are shared
 * between the problems, so each is split over fewer token ranges and adds fewer fp32 partial tiles atomically -- with the
 * weight gradients of ~6 transformer blocks in one launch no problem is split at all and the partial tiles are plain
 * read-add-stores (the atomics of a block's own launch were 21-27 us of its ~100).  The deterministic mode takes at most 8. */
#define AP_TN_MAX_GROUP 32
typedef struct ap_tn_problem {
    const ap_bf16* A; int lda;      /* [M,N1] */
    const ap_bf16* B; int ldb;      /* [M,N2] */
    float* C; int ldc;              /* [N1,N2] += A^T . B */
    int M, N1, N2;
    float alpha;                    /* C += alpha * A^T . B (0 is read as 1: zero-initialised structs keep the plain product) */
    float* colsum_A;                /* [N1] += colsum_scale * sum_m w[m] * A[m,n], or NULL */
    const ap_bf16* colsum_weight;   /* per-token weights w (bf16, ceil(M/8)*8 elements readable, 16-byte aligned), NULL = ones: the DropPath keep mask of the
                                     * bias gradient d/db of x + (mask/keep) * (y W^T + b), models/volo.py:230-234 */
    float colsum_scale;             /* used with colsum_weight only (1/keep) */
    const struct ap_patch_map* b_patch;   /* non-NULL: the rows of B are patches of an NHWC feature map (ldb ignored) -- the weight
                                           * gradient of a k x k / stride k convolution without a gathered copy of its input */
    const struct ap_bn_input* b_bn;       /* (ABI version 6, with b_patch, 64-channel maps) non-NULL: the B rows are relu(bn(.)) of the map that is read
                                           * -- PatchEmbed.proj's weight gradient on the PRE-BatchNorm output of the last stem convolution */
} ap_tn_problem;
/* `workspace` NULL: partial tiles of the token splits are added with fp32 atomics (results vary in the last bits from run to run).
 * `workspace` of >= ap_gemm_tn_grouped_workspace() bytes: DETERMINISTIC -- every split stores its partial tile and a second kernel adds
 * them in split order (bitwise reproducible; the mode behind AP_DETERMINISTIC=1 of the Python layer). */
size_t ap_gemm_tn_grouped_workspace(const ap_tn_problem* problems, int count);
int ap_gemm_tn_acc_grouped(const ap_tn_problem* problems, int count, void* workspace, size_t ws_bytes, ap_stream_t stream);
/* the same launch also performs up to AP_LN_MAX_BATCH deferred LayerNorm dgamma / dbeta reductions (ap_layernorm_bwd_partial) on
 * workgroups of its own: the weight gradients and the LayerNorm parameter gradients of one block in ONE launch */
int ap_gemm_tn_acc_grouped_ln(const ap_tn_problem* problems, int count, const ap_ln_reduce* ln_items, int ln_count,
                              void* workspace, size_t ws_bytes, ap_stream_t stream);
/* bias gradient: out[n] += sum_m A[m,n] */
int ap_colsum_acc(const ap_bf16* A, int lda, float* out, int M, int N, ap_stream_t stream);

/* ---- Outlook attention core (models/volo.py:83-98: unfold, softmax, attn@v, fold) -------
 * v [B,H,W,C], logits [B*h*w, ldl] with channel = head*81 + p*9 + q (kernel 3, pad 1, stride 2),
 * y [B,H,W,C].  C = heads*hd.                                                               */
int ap_outlook_fwd(const ap_bf16* v, const ap_bf16* logits, int ldl, ap_bf16* y,
                   int B, int H, int W, int heads, int hd, float scale, ap_stream_t stream);
int ap_outlook_bwd(const ap_bf16* v, const ap_bf16* logits, int ldl, const ap_bf16* dy,
                   ap_bf16* dv, ap_bf16* dlogits, int B, int H, int W, int heads, int hd,
                   float scale, ap_stream_t stream);
/* AvgPool2d(2,2,ceil_mode=True) on NHWC tokens (models/volo.py:75,87) */
int ap_avgpool2_fwd(const ap_bf16* x, ap_bf16* y, int B, int H, int W, int C, ap_stream_t stream);
/* dx[b,y,x,:] += dpooled[b,y/2,x/2,:] / count */
int ap_avgpool2_bwd_acc(const ap_bf16* dpooled, ap_bf16* dx, int B, int H, int W, int C, ap_stream_t stream);

/* ---- Multi-head self-attention core (models/volo.py:188-197) ---------------------------
 * qkv [B,N,3C] packed (channel = which*C + head*hd + d), out [B,N,C], lse [B,heads,N].
 * head_dim 32 / 48 / 64 (VOLO-D1..D3 and DeiT: 32 / 64; VOLO-D4/D5: 48, models/volo.py:776-821), any N:
 * N <= 256 with head_dim 32 / 64 runs LDS-resident kernels, everything else key/query-blocked ones.
 * The backward of the blocked path needs ap_mhsa_bwd_workspace() bytes of device scratch (0 for the
 * resident path: `workspace` may then be NULL).                                              */
/* out_row_scale (nullable, fp32 [B]): out[b] is stored multiplied by it.  Used with the 0/1 DropPath keep mask of the projection
 * that follows: rows of dropped samples are zeros, so the projection's weight gradient dx^T out needs no masked copy of dx
 * (models/volo.py:230-234).  The backward needs nothing extra: a dropped sample arrives with dout = 0. */
int ap_mhsa_fwd(const ap_bf16* qkv, ap_bf16* out, float* lse, int B, int N, int heads, int hd,
                float scale, const float* out_row_scale, ap_stream_t stream);
/* the same with the output a second time as OCP e4m3 bytes (out8 = sat(out * q_scale[0]), q_amax[0] raised to max |out|, nullable): the
 * operand of an fp8 output projection (configs[4]).  The key/query-blocked kernel only (N > 256 or head_dim 48): AP_ERR_UNSUPPORTED else */
int ap_mhsa_fwd_fp8(const ap_bf16* qkv, ap_bf16* out, unsigned char* out8, const float* q_scale, float* q_amax, float* lse, int B, int N,
                    int heads, int hd, float scale, const float* out_row_scale, ap_stream_t stream);
size_t ap_mhsa_bwd_workspace(int B, int N, int heads, int hd);
int ap_mhsa_bwd(const ap_bf16* qkv, const ap_bf16* out, const ap_bf16* dout, const float* lse,
                ap_bf16* dqkv, int B, int N, int heads, int hd, float scale,
                void* workspace, size_t ws_bytes, ap_stream_t stream);

/* ---- Class attention core (models/volo.py:264-274): one query per image ----------------
 * q [B,C] (un-scaled), kv [B,N,2C] (channel = which*C + head*hd + d), out [B,C], probs [B,heads,N]; head_dim 32 / 48 / 64.
 * Split layout (kv_cls != NULL): key 0 -- the class token -- is row b of kv_cls [B,2C] and keys 1..N-1 are the N-1 token rows of
 * kv [B,N-1,2C]: the reference concatenates the class token with the tokens before every class block (models/volo.py:304-308),
 * here that copy never happens.  dkv_cls must be given exactly when kv_cls is. */
int ap_class_attn_fwd(const ap_bf16* q, const ap_bf16* kv, const ap_bf16* kv_cls /*nullable*/, ap_bf16* out, float* probs,
                      int B, int N, int heads, int hd, float scale, ap_stream_t stream);
int ap_class_attn_bwd(const ap_bf16* q, const ap_bf16* kv, const ap_bf16* kv_cls /*nullable*/, const float* probs, const ap_bf16* dout,
                      ap_bf16* dq, ap_bf16* dkv, ap_bf16* dkv_cls /*nullable*/, int B, int N, int heads, int hd, float scale,
                      ap_stream_t stream);

/* ---- Mix-token region swap (models/volo.py:654-658, 685-689) ----------------------------
 * y = x except y[b, r0:r1, c0:c1, :] = x[B-1-b, r0:r1, c0:c1, :]  (x: [B,H,W,C])            */
int ap_mix_token_swap(const ap_bf16* x, ap_bf16* y, int B, int H, int W, int C,
                      int r0, int r1, int c0, int c1, ap_stream_t stream);
/* (ABI version 6) the same with the box in DEVICE memory: box_dev = int[4] {r0, r1, c0, c1} on the token-label grid, multiplied by `scale`
 * (2 for the token swap in front of the stages, 1 for the aux-logit swap: models/volo.py:654-658, 685-689).  Part of the graph-replayable
 * step: the box of the step is written into a device buffer in front of the replay instead of being a launch argument. */
int ap_mix_token_swap_dev(const ap_bf16* x, ap_bf16* y, int B, int H, int W, int C, const int* box_dev, int scale, ap_stream_t stream);

/* ---- Dense soft-target cross entropy (loss/cross_entropy.py:35-36, 147-156) -------------
 * logits [M,ldx] (C valid classes); target element (row,c) =
 *   target[(row / rows_per_batch)*t_sb + c*t_sc + (row % rows_per_batch)*t_sn]   (fp32)
 * row_loss[row] = -sum_c t*log_softmax(x);  dlogits = grad_scale*(softmax*sum_c t - t)
 * (columns C..ldx-1 of dlogits are zeroed).
 * mix_batches = B > 0: the target of batch b is mix_lam * t[b] + (1 - mix_lam) * t[B-1-b] (the mix-token image label,
 * loss/cross_entropy.py:151-152) -- no mixed copy of the target is materialised; 0: plain.       */
int ap_soft_ce_fwd_bwd(const ap_bf16* logits, int ldx, const float* target, int64_t t_sb,
                       int64_t t_sc, int64_t t_sn, int rows_per_batch, float* row_loss,
                       ap_bf16* dlogits, float grad_scale, int64_t M, int C,
                       float mix_lam, int mix_batches, ap_stream_t stream);
/* The same loss on the token-label target in its SOURCE form: the reference builds the dense [B,C,2+N] tensor on the GPU every
 * step from top-K (class, score) label maps plus label smoothing (main_prog.py:994-1004 -> tlt create_token_label_target) and
 * feeds it to loss/cross_entropy.py:136-156.  Here the target of logits row r = (b, n), b = r / rows_per_batch, is
 *   t[c] = (1 - smoothing) * sum_k [idx[o + k] == c] * val[o + k] + smoothing / C,   o = b * p_sb + n * p_sn,   k < K <= 16
 * (repeated indices accumulate, indices outside [0, C) contribute nothing); loss and gradient as ap_soft_ce_fwd_bwd.  No dense
 * target exists: 101 MB less to read per step at B = 128.
 * mix_batches = B > 0 (ABI version 5; as in ap_soft_ce_fwd_bwd): the mix-token class target of loss/cross_entropy.py:150-152,
 * t = mix_lam * t[b] + (1 - mix_lam) * t[B-1-b] -- row (b, n) takes the K pairs of its own slot weighted mix_lam and the K pairs of the
 * same slot of image B-1-b weighted 1 - mix_lam (2 K <= 16; M == B * rows_per_batch).  0: no mixing (mix_lam ignored). */
int ap_soft_ce_sparse_fwd_bwd(const ap_bf16* logits, int ldx, const int* idx, const float* val, int K, int64_t p_sb, int64_t p_sn,
                              int rows_per_batch, float smoothing, float* row_loss, ap_bf16* dlogits, float grad_scale,
                              int64_t M, int C, float mix_lam, int mix_batches, ap_stream_t stream);
/* (ABI version 6) both losses with the mix-token lam in DEVICE memory (mix_lam_dev non-NULL overrides mix_lam; pass mix_batches = B
 * always: lam = 1 gives lam t + 0 t' = t exactly) -- the graph-replayable step */
int ap_soft_ce_fwd_bwd_dev(const ap_bf16* logits, int ldx, const float* target, int64_t t_sb, int64_t t_sc, int64_t t_sn, int rows_per_batch,
                           float* row_loss, ap_bf16* dlogits, float grad_scale, int64_t M, int C, float mix_lam, int mix_batches,
                           const float* mix_lam_dev, ap_stream_t stream);
int ap_soft_ce_sparse_fwd_bwd_dev(const ap_bf16* logits, int ldx, const int* idx, const float* val, int K, int64_t p_sb, int64_t p_sn,
                                  int rows_per_batch, float smoothing, float* row_loss, ap_bf16* dlogits, float grad_scale, int64_t M, int C,
                                  float mix_lam, int mix_batches, const float* mix_lam_dev, ap_stream_t stream);
/* out[0] = wa * sum(a[0:na]) + wb * sum(b[0:nb]): cls_weight * mean(row losses) + dense_weight * mean(row losses)
 * of the token-label loss (loss/cross_entropy.py:154-156) in one launch */
int ap_loss_combine(const float* a, int64_t na, float wa, const float* b, int64_t nb, float wb, float* out, ap_stream_t stream);

/* ---- small fused elementwise helpers ---------------------------------------------------- */
/* y[m,:] = x[m,:] * scale[m / rows_per_scale] */
int ap_row_scale(const ap_bf16* x, const float* scale, ap_bf16* y, int64_t M, int C,
                 int rows_per_scale, ap_stream_t stream);
/* y = a + b (b broadcast over the leading `reps` copies when b_elems < n) */
int ap_add_bcast(const ap_bf16* a, const ap_bf16* b, ap_bf16* y, int64_t n, int64_t b_elems, ap_stream_t stream);
/* out[i] += sum over reps of x[r*n + i]  (fp32 accumulate; gradient of a broadcast add) */
int ap_sum_reps_acc(const ap_bf16* x, float* out, int64_t n, int reps, ap_stream_t stream);
/* out[oy, ox, c] (+)= sum_iy wy[oy*hi + iy] * sum_ix wx[ox*wi + ix] * in[(iy*wi + ix)*C + c]   (fp32 NHWC grids, dense tap matrices wy [ho, hi],
 * wx [wo, wi]): VOLO.interpolate_pos_encoding (models/volo.py:580-596 -- F.interpolate(pos_embed, scale_factor, mode="bicubic") on every
 * forward whose token grid differs from the embedding's) with the bicubic taps of that call as the matrices; with the transposed
 * matrices and accumulate = 1 its backward, added into the embedding's gradient.  hi, wi <= 64 (AP_ERR_UNSUPPORTED beyond).  (ABI version 5) */
int ap_resample_grid(const float* in, int hi, int wi, const float* wy, const float* wx, float* out, int ho, int wo, int C, int accumulate,
                     ap_stream_t stream);

/* ---- fused BatchNorm2d + ReLU of the conv stem on NHWC bf16 rows [T = B*H*W, C] (models/volo.py:355-367;
 * SURVEY.md row N3).  training != 0: batch statistics (biased variance), running stats updated with
 * `momentum` (unbiased variance) as nn.BatchNorm2d; mean/rstd are outputs saved for backward.
 * training == 0: mean/rstd are INPUTS (running_mean, 1/sqrt(running_var+eps)).  C/8 must be a power of two. */
size_t ap_bn_relu_workspace(int64_t T, int C);
int ap_bn_relu_fwd(const ap_bf16* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                   int training, float momentum, float eps, ap_bf16* y, float* mean, float* rstd, int64_t T, int C,
                   void* workspace, size_t ws_bytes, ap_stream_t stream);
/* training-mode ap_bn_relu_fwd whose batch statistics come from `partial`: n_partial rows [2][C] of per-channel sums and sums
 * of squares of x, produced by the kernel that wrote x (ap_conv3x3_c64 with stats != NULL) -- saves the pass over x.
 * y NULL: statistics only (mean, rstd, running statistics); the consumer applies the normalisation itself (ap_conv3x3_c64_bn) */
int ap_bn_relu_fwd_partials(const ap_bf16* x, const float* partial, int n_partial, const float* gamma, const float* beta,
                            float* running_mean, float* running_var, float momentum, float eps, ap_bf16* y, float* mean, float* rstd,
                            int64_t T, int C, ap_stream_t stream);
/* dx = d(relu(bn(x)))/dx . dy ; dgamma/dbeta accumulated (+=) */
int ap_bn_relu_bwd(const ap_bf16* dy, const ap_bf16* x, const float* gamma, const float* beta, const float* mean,
                   const float* rstd, ap_bf16* dx, float* dgamma, float* dbeta, int64_t T, int C,
                   void* workspace, size_t ws_bytes, ap_stream_t stream);

/* (ABI version 6) ap_bn_relu_bwd that also writes act = relu(bn(x)) ([T, C] bf16, the arithmetic of ap_bn_relu_fwd): for a forward that
 * applied the BatchNorm inside its consumer (ap_gemm_nt_patch_bn) and never stored the activation -- the weight gradient of that
 * consumer then reads a plain tensor */
int ap_bn_relu_bwd_act(const ap_bf16* dy, const ap_bf16* x, const float* gamma, const float* beta, const float* mean,
                       const float* rstd, ap_bf16* dx, ap_bf16* act, float* dgamma, float* dbeta, int64_t T, int C,
                       void* workspace, size_t ws_bytes, ap_stream_t stream);
/* (ABI version 6) ap_bn_relu_bwd whose first pass has already happened: `partial` holds n_partial rows [2][C] of per-channel partial sums
 * (sum dz | sum dz * xhat, dz = dy where the ReLU passed) -- written by ap_conv3x3_c64_bwd_stats, the input-gradient convolution that
 * produced dy -- so only the finalize (dgamma / dbeta +=) and the dx pass run.  Replaces the backward of nn.BatchNorm2d + nn.ReLU at
 * reference models/volo.py:356-366 (autograd), one pass over dy and x shorter. */
int ap_bn_relu_bwd_partials(const ap_bf16* dy, const ap_bf16* x, const float* gamma, const float* beta, const float* mean,
                            const float* rstd, const float* partial, int n_partial, ap_bf16* dx, float* dgamma, float* dbeta,
                            int64_t T, int C, void* workspace, size_t ws_bytes, ap_stream_t stream);

/* ---- stem 3x3 convolutions in HIP (reference models/volo.py:355-367: the two
 * nn.Conv2d(hidden, hidden, 3, 1, 1, bias=False) of PatchEmbed at hidden = 64).  NHWC bf16 feature maps. */
/* fp32 OIHW [64][64][3][3] -> the two bf16 operand layouts of ap_conv3x3_c64: w_fwd [tap][co][ci] and
 * w_bwd [8 - tap][ci][co] (input gradient = the same convolution on flipped, transposed weights); 9*64*64 elements each */
int ap_conv3x3_c64_pack(const float* w_oihw, ap_bf16* w_fwd, ap_bf16* w_bwd, ap_stream_t stream);
/* y[B,H,W,64] = conv3x3(x[B,H,W,64], stride 1, zero pad 1) with packed weights (w_fwd: forward; w_bwd with x = dy: dx).
 * stats (nullable): float[ap_conv3x3_c64_stat_rows(B,H,W)][2][64]; every workgroup stores the per-channel sum and sum of
 * squares of its bf16 outputs to its row (the partial batch statistics of the BatchNorm that follows: ap_bn_relu_fwd_partials) */
int ap_conv3x3_c64_stat_rows(int B, int H, int W);
int ap_conv3x3_c64(const ap_bf16* x, const ap_bf16* w_packed, ap_bf16* y, int B, int H, int W, float* stats, ap_stream_t stream);
/* the same on the PRE-BatchNorm output of the previous stem convolution: the kernel applies relu((x - mean) * rstd * gamma + beta) (the
 * arithmetic of ap_bn_relu_fwd, rounded to bf16) to its input while staging it, pixels outside the image stay zero -- the activation
 * between the two convolutions (models/volo.py:358-359, 361-362) is never materialised.  bn_in NULL: plain ap_conv3x3_c64. */
typedef struct ap_bn_input { const float* mean; const float* rstd; const float* gamma; const float* beta; } ap_bn_input;   /* [64] each */
int ap_conv3x3_c64_bn(const ap_bf16* x, const ap_bn_input* bn_in, const ap_bf16* w_packed, ap_bf16* y, int B, int H, int W, float* stats,
                      ap_stream_t stream);

/* dw_oihw[64][64][3][3] (fp32) += weight gradient of ap_conv3x3_c64: x the layer input, dy the output gradient (both
 * [B,H,W,64] NHWC bf16).  Per-workgroup partial sums go to `workspace` and are added in a fixed order: no atomics. */
size_t ap_conv3x3_c64_wgrad_workspace(int B, int H, int W);
int ap_conv3x3_c64_wgrad(const ap_bf16* x, const ap_bf16* dy, float* dw_oihw, int B, int H, int W, void* workspace, size_t ws_bytes,
                         ap_stream_t stream);
/* the same for a layer whose input was relu(bn(x)) of ap_conv3x3_c64_bn: x is the pre-BatchNorm tensor */
int ap_conv3x3_c64_wgrad_bn(const ap_bf16* x, const ap_bn_input* bn_in, const ap_bf16* dy, float* dw_oihw, int B, int H, int W, void* workspace,
                            size_t ws_bytes, ap_stream_t stream);

/* (ABI version 6) the INPUT-GRADIENT convolution of a stem layer (dz on w_bwd of ap_conv3x3_c64_pack -> da, as ap_conv3x3_c64) whose
 * epilogue also runs the first pass of the backward of the BatchNorm + ReLU BELOW the layer (a = relu(bn(z_below)), reference
 * models/volo.py:356-366): stats = float[ap_conv3x3_c64_stat_rows(B,H,W)][2][64], per-workgroup partial sums (sum dzb | sum dzb * xhat,
 * dzb = the bf16-rounded da where bn(z_below) > 0), the `partial` argument of ap_bn_relu_bwd_partials */
int ap_conv3x3_c64_bwd_stats(const ap_bf16* dz, const ap_bf16* w_packed_bwd, ap_bf16* da, int B, int H, int W, const ap_bf16* z_below,
                             const ap_bn_input* bn_below, float* stats, ap_stream_t stream);

/* ---- first stem convolution in HIP: 7x7 / stride 2 / pad 3, 3 -> 64, no bias (models/volo.py:355-357) on the space-to-depth input
 * xs[B, H, W, 16] (bf16; H, W = half the image size; channel (sy*2+sx)*3+c = pixel (2Y+sy, 2X+sx) channel c, 12..15 zero) */
/* fp32 [B,3,Hi,Wi] -> xs [B,Ho/2,Wo/2,16]: the bilinear resize of ap_resize_bilinear_nhwc written in that layout (Ho, Wo even) */
int ap_resize_bilinear_s2d16(const float* x, ap_bf16* xs, int B, int Hi, int Wi, int Ho, int Wo, ap_stream_t stream);
/* fp32 OIHW [64][3][7][7] -> packed bf16 operand (16*64*16 elements) */
int ap_conv7_pack(const float* w_oihw, ap_bf16* w_packed, ap_stream_t stream);
/* y[B,H,W,64] = conv7x7/s2(image) ; stats as for ap_conv3x3_c64: float[ap_conv7_s2d_stat_rows(B,H,W)][2][64] or NULL */
int ap_conv7_s2d_stat_rows(int B, int H, int W);
int ap_conv7_s2d(const ap_bf16* xs, const ap_bf16* w_packed, ap_bf16* y, int B, int H, int W, float* stats, ap_stream_t stream);
/* dw_oihw[64][3][7][7] (fp32) += weight gradient; dz = output gradient [B,H,W,64]; per-workgroup slabs in `workspace`, ordered reduction */
size_t ap_conv7_s2d_wgrad_workspace(int B, int H, int W);
int ap_conv7_s2d_wgrad(const ap_bf16* xs, const ap_bf16* dz, float* dw_oihw, int B, int H, int W, void* workspace, size_t ws_bytes,
                       ap_stream_t stream);
/* (ABI version 6) the same two with an explicit pixel stride (elements) of y / dz: 64 output channels written into / read from a wider
 * NHWC tensor.  The 128-wide stem of VOLO-D4 / D5 (models/volo.py:799-821: nn.Conv2d(3, 128, 7, 2, 3)) is two such launches on the
 * channel halves: weights w_oihw + h * 64 * 147, y + 64 h with ldy = 128 (stats per half: [rows][2][64]) */
int ap_conv7_s2d_ld(const ap_bf16* xs, const ap_bf16* w_packed, ap_bf16* y, int ldy, int B, int H, int W, float* stats, ap_stream_t stream);
int ap_conv7_s2d_wgrad_ld(const ap_bf16* xs, const ap_bf16* dz, int lddz, float* dw_oihw, int B, int H, int W, void* workspace, size_t ws_bytes,
                          ap_stream_t stream);

/* ---- (ABI version 6) 3x3 / stride 1 / pad 1 convolution at 128 channels in HIP: the stem of VOLO-D4 / D5 (stem_hidden_dim = 128,
 * models/volo.py:355-367,803,818; BASELINE configs[4]) -- csrc/conv128.hip: the input patch in LDS, the weights streamed in 18
 * (tap, 64-input-channel) slabs.  Contracts as the 64-channel family above.
 * pack: fp32 OIHW [128][128][3][3] -> w_fwd / w_bwd, 9 * 128 * 128 bf16 each (slab layouts of the kernel) */
int ap_conv3x3_c128_pack(const float* w_oihw, ap_bf16* w_fwd, ap_bf16* w_bwd, ap_stream_t stream);
/* y[B,H,W,128] = conv3x3(x[B,H,W,128]) (w_fwd: forward; w_bwd with x = dy: the input gradient); stats (nullable):
 * float[ap_conv3x3_c128_stat_rows(B,H,W)][2][128] partial sums / sums of squares of the bf16-rounded outputs (ap_bn_relu_fwd_partials) */
int ap_conv3x3_c128_stat_rows(int B, int H, int W);
int ap_conv3x3_c128(const ap_bf16* x, const ap_bf16* w_packed, ap_bf16* y, int B, int H, int W, float* stats, ap_stream_t stream);
/* dw_oihw[128][128][3][3] (fp32) += weight gradient: the 64-channel kernel on the four (output half, input half) quadrants */
size_t ap_conv3x3_c128_wgrad_workspace(int B, int H, int W);
int ap_conv3x3_c128_wgrad(const ap_bf16* x, const ap_bf16* dy, float* dw_oihw, int B, int H, int W, void* workspace, size_t ws_bytes,
                          ap_stream_t stream);

/* ---- fused optimizer step: AdamW (torch.optim.AdamW semantics, main_prog.py:484)
 * + n_ema <= 4 ModelEmaV2 updates (main_prog.py:1030-1033) over one flat fp32 slab of n parameters
 * (n % 4 == 0).  wd_mask[i] != 0 selects decoupled weight decay for element i; `ema` / `ema_decay`
 * are HOST arrays of n_ema device pointers / decays; `step` is the 1-based update count. */
/* sum of squares of a flat fp32 slab (16-byte aligned): out[0] = sum x[i]^2, deterministic (1024 ordered partials in `workspace` of
 * ap_sumsq_workspace() bytes).  The global gradient norm of torch.nn.utils.clip_grad_norm_ (timm dispatch_clip_grad 'norm',
 * prog/scaler.py:60-68, main_prog.py:129-132,1019-1027) is one pass over the gradient slab.  (ABI version 6) */
size_t ap_sumsq_workspace(void);
int ap_sumsq_f32(const float* x, int64_t n, float* out, void* workspace, size_t ws_bytes, ap_stream_t stream);
/* gradient clipping is folded into the update (ABI version 6): gnorm_sq (nullable DEVICE scalar = ap_sumsq_f32 of g) with max_norm > 0
 * scales the gradient by min(1, max_norm / (grad_scale * sqrt(gnorm_sq[0]) + 1e-6)) -- clip_grad_norm_ on the MEAN gradient, read on
 * the device (no host round trip); clip_value > 0 clamps every element of the scaled gradient to [-clip_value, clip_value]
 * (clip_grad_value_); 0 / NULL: no clipping */
int ap_adamw_ema_step(float* p, const float* g, float* m, float* v, const unsigned char* wd_mask, int64_t n,
                      float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                      float grad_scale /* g is multiplied by this first: 1/world_size turns the all-reduced SUM into the mean */,
                      const float* gnorm_sq, float max_norm, float clip_value,
                      const float* step_scalars_dev /* nullable DEVICE float[3] {lr, 1 - beta1^step, sqrt(1 - beta2^step)}: overrides lr / step
                                                     * (the graph-replayable step: the scheduler's lr and the step count change per replay) */,
                      float* const* ema, const float* ema_decay, int n_ema,
                      ap_bf16* p_bf16 /* nullable: bf16 copy of the updated parameters, same offsets */,
                      ap_stream_t stream);
/* transpose `count` bf16 matrices of one slab in one launch: desc_dev = device array of
 * {int64 src_off, int64 dst_off, int rows, int cols, int ld_dst, int first_tile} (32x32 tiles, first_tile
 * = running tile prefix); dst[dst_off + c*ld_dst + r] = src[src_off + r*cols + c], pad columns zeroed */
int ap_batched_transpose_bf16(const ap_bf16* src, ap_bf16* dst, const void* desc_dev, int count, int total_tiles,
                              ap_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* AUTOPROG_HIP_H */
