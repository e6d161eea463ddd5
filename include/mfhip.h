/*
 * mfhip.h — C ABI of libmfhip.so, the MI355X (gfx950) kernel library behind the
 * MirrorFusion (SD1.5 + BrushNet) denoising hot path.
 *
 * The reference (val-iisc/Reflecting-Reality, MirrorFusion/src/diffusers) is 100 % Python and has
 * no native operator boundary; every arithmetic step is an ATen call.  Each entry point below
 * therefore cites the reference *call site* it replaces (paths relative to
 * MirrorFusion/src/diffusers/).  All entry points:
 *   - are extern "C", take plain pointers / sizes / a hipStream_t (passed as void*),
 *   - never allocate, free, synchronise or retain caller memory (graph-capture safe),
 *   - return 0 on success, a negative MF_E* code otherwise; mf_last_error() gives the text.
 * Device buffers are owned by the caller (PyTorch-ROCm allocates them).
 *
 * Layout conventions: activations are NHWC ("pixel-major": [batch][y][x][channel]) which is
 * bit-identical to the [batch][tokens][channels] layout of the transformer blocks, so no
 * transposes exist between conv and attention.  Weights are [N][K] with K contiguous
 * (conv: [Cout][ky][kx][Cin]).  dtype codes: MF_F32 = 0, MF_BF16 = 1.
 */
#ifndef MFHIP_H
#define MFHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MF_F32 0
#define MF_BF16 1
/* compute-only codes for mf_gemm_desc.dtype ("split" precision: the mode that meets the reference's fp32 results to
 * 1e-3 on the latents AND runs on the 16-bit matrix pipe).  Operands are fp32 in memory; every value x is split into
 * two 16-bit halves x = hi + lo (hi = x rounded toward zero to 11 / 8 significant bits, lo = x - hi, exact in fp32,
 * rounded to 16 bits) and a product is three MFMAs: hi*hi + hi*lo + lo*hi, fp32 accumulate.  MF_F16X3: fp16 halves
 * (22 significant bits, |x| < 65504); MF_BF16X3: bf16 halves (16 bits, fp32 range). */
#define MF_F16X3 2
#define MF_BF16X3 3
/* OCP fp8 e4m3 operands (1 byte each, a_dtype == MF_FP8 too) on the block-scaled matrix instruction
 * v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales (2x the bf16 MFMA rate); the quantisation scales are
 * per A row / per W row fp32 vectors applied in the epilogue (mf_gemm_desc.a_scale / w_scale) */
#define MF_FP8 4
/* fp32 operands in memory (a_dtype == MF_F32, raw fp32 W), each rounded to bf16 (nearest-even) on its way into ONE bf16
 * MFMA, fp32 accumulate: the arithmetic of the reference's --mixed_precision=bf16 autocast (train_brushnet_mirror.py:567,
 * 1127-1131) with the activations still stored in fp32.  Training compute code (mf_gemm_conv, mf_conv_wgrad). */
#define MF_BF16X1 5
/* fp16 operands and activation storage (torch.float16: the default of the reference's inference script,
 * examples/brushnet/test_brushnet.py:122-126), one v_mfma_f32_*_f16 per product at the bf16 rate, fp32 accumulate / statistics.
 * Storage AND compute code: the byte layout of every tensor equals the MF_BF16 one.  11 significant bits against bf16's 8, range
 * |x| < 65504 (a value beyond it becomes inf, as in the reference's fp16 run). */
#define MF_F16 6

#define MF_OK 0
#define MF_EINVAL (-1)   /* bad argument / unsupported shape */
#define MF_ELAUNCH (-2)  /* hipLaunch failed */
#define MF_EALIGN (-3)   /* pointer / stride alignment not met */

#define MF_ACT_NONE 0
#define MF_ACT_SILU 1
/* GEGLU fused into the GEMM: weight rows interleaved [4 value rows | 4 gate rows]; writes n/2 channels */
#define MF_ACT_GEGLU4 2

/* ABI version, bumped on any struct change; checked by the Python host at load time. */
#define MF_ABI_VERSION 20
int mf_abi_version(void);
const char* mf_last_error(void);
/* sizeof() of the descriptor structs, so a foreign-language binding can verify its layout */
int mf_sizeof_gemm_desc(void);
int mf_sizeof_groupnorm_desc(void);

/* --------------------------------------------------------------------------------------------
 * mf_gemm_conv — implicit-GEMM convolution / linear / strided-batched NT GEMM on MFMA.
 *
 *   out[z][m][n] = alpha * ( sum_k A[z][m][k] * W[z][n][k] + bias + temb[b(m)][n] )
 *                  + res0[m][n] + res1[m][n]              (then optional activation)
 *
 * A is gathered on the fly from up to two NHWC tensors concatenated along channels
 * (torch.cat([h, skip], 1) is never materialised), optionally through a nearest-2x upsample,
 * with zero padding; k = (ky*kw + kx)*(c0+c1) + c.
 *
 * Replaces: LoRACompatibleConv/F.conv2d (models/lora.py:293-366; call sites resnet.py:278,294,
 * 320-327, downsampling.py:118-120,134-154, upsampling.py:170-191, brushnet.py:221,325-447,
 * transformer_2d.py:162,225, unet_2d_condition.py conv_in/conv_out), LoRACompatibleLinear/F.linear
 * (lora.py:368-451; attention_processor.py:190-205, activations.py:92, attention.py:663,
 * resnet.py:282, embeddings.py:225-237), torch.cat (unet_2d_blocks.py:2586,2728), the injection
 * adds (unet_2d_blocks.py:1389,1398,1484,1493,2627,2635,2752,2761; unet_2d_condition.py:1218,1289)
 * and, in batched form, the QK^T / PV products of attention_processor.py:577-622.
 * ------------------------------------------------------------------------------------------ */
typedef struct mf_gemm_desc {
    /* compute dtype: MF_BF16 (bf16 MFMA 32x32x16, fp32 accumulate), MF_F32 (fp32 MFMA 32x32x2) or a split code
     * MF_F16X3 / MF_BF16X3 (fp32 operands, three 16-bit MFMAs per product; a_dtype must be MF_F32), MF_BF16X1 (fp32
     * operands, one bf16 MFMA per product; raw W only) */
    int32_t dtype;
    /* A operand */
    const void* a0;       /* segment 0, NHWC */
    const void* a1;       /* segment 1 or NULL */
    int32_t c0, c1;       /* channels per segment (c1 = 0 when a1 is NULL) */
    int64_t lda0, lda1;   /* pixel stride of each segment, in elements (>= c) */
    int32_t a_dtype;      /* storage dtype of a0/a1: == dtype, or MF_F32 with dtype == MF_BF16 */
    int32_t batch, h_in, w_in;  /* input geometry (before upsample) */
    int32_t h_out, w_out;
    int32_t kh, kw, stride, pad_t, pad_l;
    int32_t upsample;     /* 1: nearest-neighbour 2x before the convolution */
    /* W operand, [n][k] row-major, row stride ldw elements, dtype == dtype (fp32 for the split codes).
     * w_split = 1 (split codes only): W was split ahead of time — per block of 32 k the row holds the 32 high halves
     * then the 32 low halves (16-bit each: the same 128 bytes as 32 floats), rows zero-padded to a multiple of 32 k,
     * ldw counted in 4-byte units.  w_split = 0: raw fp32, split on the fly (activation x activation products). */
    const void* w;
    int64_t ldw;
    int32_t w_split;
    int32_t n;            /* output channels */
    /* strided batching (attention): blockIdx.z = z, offset = (z / zdiv)*zs_o + (z % zdiv)*zs_i */
    int32_t nz, zdiv;
    int64_t a_zs_o, a_zs_i, w_zs_o, w_zs_i, o_zs_o, o_zs_i;
    /* epilogue */
    const float* bias;    /* [n] (bias_mode 0) or [m] (bias_mode 1) or NULL */
    int32_t bias_mode;
    const float* temb;    /* [batch][ld_temb] fp32 or NULL; row = m / (h_out*w_out) */
    int64_t ld_temb;
    /* out = alpha * (a_scale[m] * w_scale[n] * acc + bias + temb) + ... : dequantisation of fp8 operands (either may be
     * NULL = 1; usable with any dtype).  Batched calls: vector z starts at a_scale + (z / zdiv) * a_scale_zs (w alike) */
    const float* a_scale; const float* w_scale;
    int64_t a_scale_zs, w_scale_zs;
    const void* res0; int32_t res0_dtype; int64_t ld_res0;
    const void* res1; int32_t res1_dtype; int64_t ld_res1;
    /* 0 (or >= M): res1 has one row per output row.  0 < res1_rows < M (must divide M): res1 has res1_rows rows and output
     * row m adds row m % res1_rows — one residual shared by batch replicas (the BrushNet residual of both halves of a
     * classifier-free-guidance batch: pipeline_brushnet.py:1256-1296 feeds an attention-free BrushNet the same inputs twice) */
    int32_t res1_rows;
    float alpha;
    int32_t act;
    void* out; int32_t out_dtype; int64_t ldc;
    /* split-K: 0 = library heuristic (uses `ws` when it is large enough), 1 = off, >1 = forced;
     * `ws` is a caller-owned scratch of `ws_floats` floats (splitk*nz*M*N are needed) */
    int32_t splitk;
    float* ws;
    int64_t ws_floats;
    /* tile selection: 0 = library heuristic, else an index into the instantiated tile table (mf_gemm_num_tiles).
     * 1-15: implicit-GEMM tiles (any call); 16-19: 3x3 / stride-1 kernels with the input patch resident in LDS (bf16,
     * image sizes that tile by 8x16 / 16x16); 20-24: implicit GEMM with dx-tap reuse of the A window (3x3 / stride 1,
     * W divides or is divided by BM); 25-30: 16x16x32-MFMA forms (bf16); 31-36: deeper LDS rings; 37-40, 47: warp-specialised
     * dx-reuse convs (bf16: extra waves that only stage operands); 41-46, 48: the warp-specialised form of the plain ring
     * (bf16, any call; the only tiles that serve ln_colsum / vt_out).  A tile that does not apply to the call returns MF_EINVAL (the host autotuner skips it); nothing
     * is silently rerouted. */
    int32_t tile;
    /* LayerNorm folded into the GEMM (attention.py:203,233,261 -> the Linear that follows): ln_colsum[n] = sum_k W[n][k] of the
     * weight handed in, which the caller has already multiplied by the norm's gamma (and whose bias already holds W.beta):
     *   out = rstd[m] * (acc[m][n] - mean[m] * ln_colsum[n]) + bias ...  with mean / rstd of A row m over all K (eps = ln_eps),
     * gathered inside the kernel while the A tiles pass through LDS — the normalised tensor is never written.  bf16 1x1 GEMMs
     * with one A segment and no split-K; served by the warp-specialised ring tiles (41-46, 48).  NULL = off. */
    const float* ln_colsum; float ln_eps;
    /* Transposed output columns (a fused q | k | v projection whose V third attention wants as V^T): columns n >= vt_n0 are not
     * written to `out` but as vt_out[image][n - vt_n0][token] (bf16, row stride vt_ld elements), image = m / vt_tokens,
     * token = m % vt_tokens.  vt_n0 must be a multiple of 640; same tiles as ln_colsum.  NULL = off. */
    void* vt_out; int32_t vt_n0, vt_tokens; int64_t vt_ld;
    /* In-launch split-K combine: sk_tickets = `sk_ticket_cap` 32-bit arrival counters owned by the caller, ALL ZERO when the call
     * is issued (the kernel leaves them zero again; one buffer per stream, like `ws`).  With split-K > 1 and at most
     * sk_ticket_cap output tiles, the K-slice block that arrives last at its tile's counter sums the tile's fp32 slabs in slice
     * order (bit-reproducible) and runs the epilogue itself: the separate reduce launch and the kernel boundary behind the
     * dirty partials disappear.  Built into the tiles small-M / deep-K calls use (1, 2, 3, 6 and the warp-specialised rings 41,
     * 43, 44, 48 in bf16; 1, 2, 3, 6, 41, 44 in MF_F16X3); any other tile, or NULL, keeps the reduce launch — same result. */
    uint32_t* sk_tickets; int32_t sk_ticket_cap;
    /* GroupNorm statistics from the producer (round 6; resnet.py:337-338,381,393, transformer_2d.py:158: every GroupNorm of the
     * path reads the output of one of these launches).  gn_part != NULL: besides `out`, the call leaves per-channel partial sums
     * of the FINAL output values (after bias / temb / residuals / activation; fp32, taken before the storage rounding when the
     * epilogue produces them) in gn_part as float pairs
     *     gn_part[(m / R) * n_channels + n] = (sum, sum of squares) over output rows [R * (m / R), R * (m / R) + R) of channel n,
     * and writes R (> 0, a divisor of h_out * w_out, so a block of rows lies inside one image) to *gn_part_rows (a HOST int).
     * mf_groupnorm takes (gn_part, R) as part0 / part1 and skips its statistics pass over the tensor.  Produced inside the GEMM's
     * epilogue (R = the tile's rows; fixed summation order, bit-reproducible) when the launch has no split-K reduce and the tile
     * is an implicit-GEMM tile with h_out * w_out a multiple of its rows; otherwise by one extra column-sum launch over the
     * stored output (R = 128, 64 or 32) — same contract.  Needs n % 8 == 0, nz == 1, h_out * w_out % 32 == 0, no GEGLU, no
     * vt_out, and gn_part_floats >= 2 * n * (M / 32) (enough for the smallest R). */
    float* gn_part; int64_t gn_part_floats; int32_t* gn_part_rows;
    /* gn_groups > 0 (with gn_part): the consumer is a GroupNorm over exactly this tensor with gn_groups groups (the common case: every
     * norm of the path but the decoder's concatenated ones).  When the epilogue produces the sums and the tile's columns hold whole
     * groups (n / gn_groups divides the tile's column count: the 160-column tiles for 320 / 640 / 1280 channels in 32 groups), it also
     * leaves PER-GROUP sums behind the per-channel ones,
     *     gn_part[2 * n * (M / R) + 2 * ((m / R) * gn_groups + g)] = (sum, sum of squares) over rows [R (m / R), +R) of group g,
     * and writes 1 to *gn_grouped (a HOST int, else 0): mf_groupnorm then needs no finalize launch at all (grp0 below). */
    int32_t gn_groups; int32_t* gn_grouped;
    /* Split-K reduce left to the consumer (round 6, ABI 20).  The 8 x 8 / 16 x 16 levels run their 3x3 convs with split-K (M = 512 / 2048
     * rows cannot fill 256 CUs otherwise); the first conv of a ResnetBlock2D feeds nothing but norm2 (resnet.py:381-393), so the launch that
     * sums the K slices can be the GroupNorm itself.  defer_reduce != 0: when this call ends with split-K slabs in `ws` and an epilogue of
     * bias (per column) / temb / alpha only — no residual, activation, scales, folded LayerNorm, vt_out, gn_part; nz == 1; the 8-channel
     * vector form — the reduce launch is SKIPPED, `out` is NOT written and *deferred_splits (a HOST int) receives the number of slabs
     * [split][M][n] fp32 in ws; mf_groupnorm takes them as sk_ws (below) and must be the next user of `ws` on this stream.  Otherwise
     * *deferred_splits = 0 and the call is complete as always. */
    int32_t defer_reduce; int32_t* deferred_splits;
} mf_gemm_desc;

int mf_gemm_conv(const mf_gemm_desc* d, void* stream);
/* number of instantiated tile configurations and their (BM, BN) */
int mf_gemm_num_tiles(void);
int mf_gemm_tile_shape(int tile, int* bm, int* bn);
/* bumped whenever tile indices are renumbered or change meaning (appending tiles keeps it): a host-side cache of tuned
 * tile indices is only valid for the version it was recorded under */
int mf_gemm_tile_table_version(void);

/* --------------------------------------------------------------------------------------------
 * mf_groupnorm — GroupNorm over NHWC (optionally over the channel-concat of two tensors),
 * fused with SiLU.  Replaces torch.nn.GroupNorm + SiLU (resnet.py:337-338,381,393;
 * transformer_2d.py:158,338; unet_2d_condition.py:1336-1338; vae.py; attention_processor.py:1244).
 * `ws` is a scratch buffer of mf_groupnorm_ws_floats(batch, groups, c0 + c1) floats.
 * ------------------------------------------------------------------------------------------ */
typedef struct mf_groupnorm_desc {
    const void* x0; const void* x1;   /* NHWC segments; x1 may be NULL */
    int32_t c0, c1;
    int32_t in_dtype;                 /* MF_F32 / MF_BF16 */
    int32_t batch, hw;                /* hw = H*W */
    int32_t groups;
    float eps;
    const float* gamma; const float* beta;   /* [c0+c1] fp32 */
    int32_t silu;                     /* 1: y = silu(gn(x)) */
    void* out; int32_t out_dtype;     /* [batch][hw][c0+c1] */
    float* ws;
    float* stats_out;                 /* nullable [batch][groups][2]: (mean, rstd) of every group, for mf_groupnorm_bwd's stats_in */
    /* Statistics handed over by the producer(s) of x0 / x1 (mf_gemm_desc.gn_part and the R it reported): per-channel (sum, sum of
     * squares) of every block of part_rows rows.  When every present segment has its partials (and hw > 256: below that the
     * one-launch form keeps the rows in registers anyway) the statistics pass over the tensor is replaced by a small finalize
     * launch over the partials; otherwise they are ignored.  NULL = compute the statistics from the tensor. */
    const float* part0; int32_t part0_rows;
    const float* part1; int32_t part1_rows;
    /* Per-GROUP sums of x0 from its producer (mf_gemm_desc.gn_groups == groups, *gn_grouped == 1): float pairs [batch * hw / grp0_rows]
     * [groups].  One segment only (x1 == NULL), at most 64 row blocks per image: every block of the apply pass combines its image's
     * blocks itself (fixed order, double) and the normalisation is ONE launch. */
    const float* grp0; int32_t grp0_rows;
    /* The input as a deferred split-K reduce (mf_gemm_desc.defer_reduce): x = round(((sum over sk_splits slabs of sk_ws[split][row][c])
     * + sk_bias[c] + sk_temb[image * sk_ld_temb + c]) * sk_alpha), summed in slab order and rounded to in_dtype — bit for bit what the
     * reduce launch would have stored — instead of reading x0 (ignored; c1 must be 0).  Only where the one-launch form applies
     * (hw <= 256, channels % 8 == 0: the levels that split K); refused otherwise.  sk_bias / sk_temb nullable. */
    const float* sk_ws; int32_t sk_splits;
    const float* sk_bias; const float* sk_temb; int64_t sk_ld_temb; float sk_alpha;
} mf_groupnorm_desc;
int mf_groupnorm(const mf_groupnorm_desc* d, void* stream);
int64_t mf_groupnorm_ws_floats(int32_t batch, int32_t groups, int32_t channels);

/* LayerNorm over the last dim of [rows][c]; replaces nn.LayerNorm (attention.py:203,233,261). */
int mf_layernorm(const void* x, int32_t in_dtype, void* out, int32_t out_dtype, const float* gamma,
                 const float* beta, int64_t rows, int32_t c, float eps, void* stream);

/* Row softmax in place over scores[rows][ld] (fp32), valid length `cols`; columns in
 * [cols, ld) are written as 0 so the buffer can feed a K-padded GEMM.  Writes P in `out`
 * ([rows][ld], out_dtype).  Replaces attention_probs.softmax(dim=-1) (attention_processor.py:616). */
int mf_softmax_rows(const float* scores, void* out, int32_t out_dtype, int64_t rows, int32_t cols,
                    int32_t ld, void* stream);

/* Fused flash-style attention (bf16): out[b][s][h*d + :] = softmax(q k^T * scale) v.
 * q: [B][Sq][ldq], k: [B][Skv][ldk], vt: V^T as [B][heads*d][ldvt] (keys contiguous),
 * replaces F.scaled_dot_product_attention (attention_processor.py:1266-1268). */
int mf_attention_bf16(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* vt, int64_t ldvt,
                      void* out, int64_t ldo, int32_t batch, int32_t heads, int32_t sq, int32_t skv,
                      int32_t head_dim, float scale, void* stream);
/* mf_attention_bf16 on fp16 operands (MF_F16 storage: q, k, vt and out are fp16; the f16 MFMA forms, fp32 softmax and accumulation):
 * F.scaled_dot_product_attention of the reference's default torch_dtype=torch.float16 run (examples/brushnet/test_brushnet.py:122-126,
 * models/attention_processor.py:1259-1268).  Same layouts, head dims and constraints. */
int mf_attention_f16(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* vt, int64_t ldvt,
                      void* out, int64_t ldo, int32_t batch, int32_t heads, int32_t sq, int32_t skv,
                      int32_t head_dim, float scale, void* stream);

/* The same kernel in split precision (the parity mode's attention): every operand is given as two fp16 planes
 * (hi, lo) of identical layout with hi + lo = the fp32 value to 22 bits (mf_split_halves makes them), each product is
 * three fp16 MFMAs, P is split in registers, out is fp32 [B][Sq][ldo].  head_dim 8 / 40 / 64 / 80. */
int mf_attention_f16x3(const void* q_hi, const void* q_lo, int64_t ldq, const void* k_hi, const void* k_lo, int64_t ldk,
                       const void* vt_hi, const void* vt_lo, int64_t ldvt, float* out, int64_t ldo, int32_t batch,
                       int32_t heads, int32_t sq, int32_t skv, int32_t head_dim, float scale, void* stream);
/* The same with the row statistics the backward pass needs: lse[b][head][i] = log2 of row i's softmax denominator in the exp2
 * domain (max + log2 sum, with scale * log2 e folded in), fp32 [B][heads][Sq]; lse may be NULL. */
int mf_attention_f16x3_lse(const void* q_hi, const void* q_lo, int64_t ldq, const void* k_hi, const void* k_lo, int64_t ldk,
                           const void* vt_hi, const void* vt_lo, int64_t ldvt, float* out, int64_t ldo, float* lse, int32_t batch,
                           int32_t heads, int32_t sq, int32_t skv, int32_t head_dim, float scale, void* stream);

/* Flash-style attention BACKWARD in split precision (the training step's backward through
 * F.scaled_dot_product_attention, attention_processor.py:1266-1268 under train_brushnet_mirror.py:1459): dQ, dK, dV from Q, K, V,
 * dO, the forward's lse and dd[b][head][i] = sum_c dO[b][i][head*d + c] * O[b][i][head*d + c] (mf_rowdot_heads).  P is recomputed
 * tile by tile; nothing of size Sq x Skv is written; no atomics (dK / dV and dQ are two passes of one kernel), so the result is
 * bit-reproducible.  Every operand is given as two fp16 planes (mf_split_halves) — row-major [B][S][ld] for Q, K, V, dO and
 * transposed [B][heads*d][ldt] (ldt >= S, a multiple of 8, pad columns finite) for Q, K, dO.  head_dim 8 / 40 (the 4096-token
 * layers of the path; head dims 64 / 80 exceed the LDS of the double-buffered dK / dV pass and keep the unfused backward). */
typedef struct mf_attn_bwd_desc {
    const void* q_hi; const void* q_lo; int64_t ldq;
    const void* k_hi; const void* k_lo; int64_t ldk;
    const void* v_hi; const void* v_lo; int64_t ldv;
    const void* do_hi; const void* do_lo; int64_t lddo;
    const void* qt_hi; const void* qt_lo; int64_t ldqt;
    const void* kt_hi; const void* kt_lo; int64_t ldkt;
    const void* dot_hi; const void* dot_lo; int64_t lddot;
    const float* lse; const float* dd;
    void* dq; void* dk; void* dv; int64_t ldo;         /* [B][S][ldo], fp32 — or bf16 with out_dtype = MF_BF16 (mf_attention_bwd_bf16 only) */
    int32_t batch, heads, sq, skv, head_dim;
    float scale;
    int32_t out_dtype;                                 /* 0 / MF_F32: fp32 gradients; MF_BF16: bf16 (the operands of the q / k / v projections'
                                                          weight and data gradients in the MF_BF16X1 mode, which rounds them anyway) */
} mf_attn_bwd_desc;
int mf_sizeof_attn_bwd_desc(void);
int mf_attention_bwd_f16x3(const mf_attn_bwd_desc* d, void* stream);
/* The same backward on ONE bf16 plane per operand (the *_hi pointers; *_lo are ignored) and one bf16 MFMA per product, P and dS
 * rounded to bf16: the attention backward of the MF_BF16X1 training mode on pre-rounded operands (what the reference's bf16 autocast
 * computes in F.scaled_dot_product_attention's backward, attention_processor.py:1266-1268).  head_dim 8 / 40 / 80. */
int mf_attention_bwd_bf16(const mf_attn_bwd_desc* d, void* stream);
/* mf_attention_bf16 that also writes lse[b][head][q] = log2 of the row's softmax denominator in the exp2 domain (m + log2 l), the row
 * statistic the flash backward recomputes P from.  lse may be NULL. */
int mf_attention_bf16_lse(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* vt, int64_t ldvt, void* out, int64_t ldo,
                          float* lse, int32_t batch, int32_t heads, int32_t sq, int32_t skv, int32_t head_dim, float scale, void* stream);
/* mf_rowdot_heads with a bf16 second operand (the bf16 forward's output O) */
int mf_rowdot_heads_bf16(const float* a, const void* b, float* out, int32_t batch, int32_t sq, int32_t heads, int32_t head_dim, int64_t ld,
                         void* stream);
/* mf_rowdot_heads_bf16 on contiguous [batch * sq][heads * head_dim] rows that ALSO writes a16 = bf16(a): the flash backward of the
 * MF_BF16X1 mode needs D = rowdot(dO, O) and the rounded dO, both from one read of dO.  head_dim % 8 == 0, heads * head_dim <= 2048. */
int mf_rowdot_heads_cast(const float* a, const void* b16, void* a16, float* out, int32_t batch, int32_t sq, int32_t heads, int32_t head_dim,
                         void* stream);
/* out[b][head][i] = sum_c a[b][i][head*d + c] * b[b][i][head*d + c]  (fp32 [B][S][ld] inputs) */
int mf_rowdot_heads(const float* a, const float* b, float* out, int32_t batch, int32_t sq, int32_t heads, int32_t head_dim, int64_t ld,
                    void* stream);

/* x (n fp32, n % 4 == 0) -> fp16 planes hi = x rounded toward zero, lo = (x - hi) rounded toward zero */
int mf_split_halves(const float* x, void* hi, void* lo, int64_t n, void* stream);
/* Range guard of MF_F16X3.  The fp16 halves hold |x| <= 65504 only and v_cvt_pkrtz_f16_f32 SATURATES above that (no inf, no
 * NaN: nothing downstream would notice), so every kernel that splits operands — mf_gemm_conv and mf_conv_wgrad with
 * MF_F16X3, mf_split_halves — tracks max |operand| and raises a sticky device flag when it leaves the range.  This call
 * copies the flags to *raised (bit 0: mf_gemm_conv, bit 1: mf_split_halves / mf_attention_f16x3 operands, bit 2:
 * mf_conv_wgrad; 0 = every split since the last reset was exact to 22 bits) and clears them when `reset`.  Diagnostic entry:
 * the ONLY one that synchronises `stream`; never call it inside a hipGraph capture.  A caller that sees a flag re-runs the
 * affected work in MF_BF16X3 (fp32 range, 16 bits) or MF_F32. */
int mf_split_overflow(int32_t reset, int32_t* raised, void* stream);

/* Per-row dynamic fp8 (e4m3) quantisation of [rows][c] (c % 8 == 0, c <= 8192), optionally fused behind a LayerNorm
 * (gamma / beta non-NULL: y = LN(x) first, attention.py:203,233,261): out_q[r][j] = fp8(y[r][j] / scale[r]),
 * scale[r] = max_j |y[r][j]| / 448.  The pair (out_q, scale) is an fp8 GEMM's A operand + a_scale. */
int mf_quantize_rows_fp8(const void* x, int32_t in_dtype, void* out_q, float* scale, int64_t rows, int32_t c,
                         const float* gamma, const float* beta, float eps, void* stream);

/* ---- elementwise / layout -------------------------------------------------------------- */
/* NCHW fp32 -> NHWC (dtype), channels zero-padded to c_pad; two sources concatenated along C
 * (brushnet.py:810 torch.concat([sample, brushnet_cond], 1)); src1 may be NULL. */
int mf_pack_nhwc(const float* src0, int32_t c0, const float* src1, int32_t c1, void* dst, int32_t dst_dtype,
                 int32_t c_pad, int32_t batch, int32_t hw, void* stream);
/* NHWC (dtype, channel stride ld) -> NCHW fp32, first c channels */
int mf_unpack_nchw(const void* src, int32_t src_dtype, int64_t ld, float* dst, int32_t c, int32_t batch,
                   int32_t hw, void* stream);
/* out = a + b (elementwise over n elements, dtypes independent) (unet_2d_condition.py:1218) */
int mf_add(const void* a, int32_t a_dtype, const void* b, int32_t b_dtype, void* out, int32_t out_dtype,
           int64_t n, void* stream);
/* out[i] = bf16(x[i]), nearest-even, n elements (x and out 16-byte aligned): the operand copies of the MF_BF16X1 training mode —
 * the reference's autocast rounds every conv / linear operand to bf16 (train_brushnet_mirror.py:567, 1127-1131); rounding them ONCE
 * into a copy lets the step's forward and data-gradient GEMMs run on the LDS-DMA bf16 kernels of the inference path. */
int mf_cast_bf16(const float* x, void* out, int64_t n, void* stream);
/* GEGLU: out[r][j] = h[r][j] * gelu_erf(h[r][c + j]) for h = [rows][2c] (activations.py:100-103) */
int mf_geglu(const void* h, int32_t in_dtype, void* out, int32_t out_dtype, int64_t rows, int32_t c,
             void* stream);
/* sinusoidal timestep embedding, flip_sin_to_cos / freq_shift as embeddings.py:27-67; out [n][dim] fp32 */
int mf_timestep_embedding(const float* t, float* out, int32_t n, int32_t dim, int32_t flip_sin_to_cos,
                          float freq_shift, void* stream);
/* SiLU on fp32 vector (resnet.py:372 nonlinearity(temb)) */
int mf_silu_f32(const float* x, float* out, int64_t n, void* stream);

/* Fused classifier-free guidance + DDIM step (pipeline_brushnet.py:1310-1315,
 * scheduling_ddim.py:404-450, eta = 0):
 *   e = eu + g*(ec - eu)                       (g < 0: e = eu, no guidance)
 *   pred_type 0 (epsilon):      x0 = (x - sqrt_1m_at*e)/sqrt_at ; eps = e
 *   pred_type 1 (v_prediction): x0 = sqrt_at*x - sqrt_1m_at*e   ; eps = sqrt_at*e + sqrt_1m_at*x
 *   clip > 0: x0 = clamp(x0, -clip, clip)      (clip_sample, :427-430)
 *   x_prev = sqrt_ap*x0 + dir_coef*eps
 * eps_u/eps_c: the two halves of the UNet output in NCHW fp32; eps_out (nullable) receives e. */
int mf_cfg_ddim_step(const float* eps_u, const float* eps_c, float g, const float* x, float* x_prev,
                     float sqrt_at, float sqrt_1m_at, float sqrt_ap, float dir_coef, int32_t pred_type,
                     float clip, float* eps_out, int64_t n, void* stream);
/* Same update with the four coefficients {sqrt_at, sqrt_1m_at, sqrt_ap, dir_coef} read from DEVICE memory, so a
 * captured hipGraph of the whole denoise step can be replayed for every timestep (the host only refreshes the
 * coefficient / timestep buffers between replays).  x_prev may alias x (in-place). */
int mf_cfg_ddim_step_dev(const float* eps_u, const float* eps_c, float g, const float* x, float* x_prev,
                         const float* coef4, int32_t pred_type, float clip, int64_t n, void* stream);
/* CFG combine only (PNDM keeps its own history on the host side): eps = eu + g*(ec-eu) */
int mf_cfg_combine(const float* eps_u, const float* eps_c, float g, float* eps, int64_t n, void* stream);
/* generic y = sum_i c[i]*x[i] (i < nin <= 6) — PNDM/PLMS linear multistep and _get_prev_sample
 * (scheduling_pndm.py:370-382,436-446) */
int mf_axpby_n(const float* const* xs, const float* coefs, int32_t nin, float* y, int64_t n, void* stream);
/* training loss (examples/brushnet/train_brushnet_mirror.py:1433-1449): per_sample[r] = mean_i (pred[r][i] -
 * target[r][i])^2 * (weights ? weights[r] : 1) and loss[0] = mean_r per_sample[r]; fp32 in, double accumulation */
int mf_mse_loss(const float* pred, const float* target, const float* weights, float* per_sample, float* loss,
                int32_t rows, int64_t n, void* stream);
/* DiagonalGaussianDistribution.sample * scaling (vae.py:769-791, pipeline_brushnet.py:1188):
 * moments NHWC [b][hw][ld] (mean = ch 0..c-1, logvar = ch c..2c-1) -> z NCHW fp32 [b][c][hw] */
int mf_vae_sample(const void* moments, int32_t m_dtype, int64_t ld, const float* noise, float* z, int32_t c,
                  int32_t batch, int32_t hw, float scaling, void* stream);
/* nearest-neighbour resize of NCHW fp32 planes (F.interpolate default, pipeline_brushnet.py:1189-1200) */
int mf_nearest_resize(const float* src, float* dst, int32_t planes, int32_t h_in, int32_t w_in, int32_t h_out,
                      int32_t w_out, void* stream);

/* ---- image front-end on the device (image_processor.py:446-610, pipeline_brushnet.py:1139,1196-1215,
 * examples/brushnet/dataset/dataset.py:98-145); fp32 NCHW planes, no host synchronisation ------------------- */
/* out2[0] = min, out2[1] = max of x[0..n) (only where mask[i] > 0 when mask is given); ws: mf_minmax_ws_floats() floats */
int64_t mf_minmax_ws_floats(void);
int mf_minmax(const float* x, const float* mask, int64_t n, float* out2, float* ws, void* stream);
/* VaeImageProcessor.preprocess's normalisation: y = 2x - 1 if minmax[0] >= 0 (device value from mf_minmax) else y = x */
int mf_image_normalize(const float* x, float* y, int64_t n, const float* minmax, void* stream);
/* out[b][0][p] = (sum_c mask[b][c][p] < 0) ? 1 : 0: 1 = keep, 0 = hole (pipeline_brushnet.py:1139) */
int mf_mask_keep(const float* mask, float* out, int32_t batch, int32_t channels, int64_t hw, void* stream);
/* torch.cat along channels of up to 8 NCHW sources into out [batch][sum channels][hw]; source i has batches[i] images and
 * is repeated (image b % batches[i]) when that is smaller than batch */
int mf_concat_channels(const float* const* srcs, const int32_t* channels, const int32_t* batches, int32_t nsrc, float* out,
                       int32_t batch, int64_t hw, void* stream);
/* VaeImageProcessor.postprocess: clamp(x / 2 + 0.5, 0, 1) (denormalize) to fp32 NCHW (out_f32) and / or round(. * 255) to
 * uint8 NHWC (out_u8, what PIL.Image.fromarray takes); either output may be NULL */
int mf_postprocess(const float* x, float* out_f32, void* out_u8, int32_t batch, int32_t channels, int64_t hw, int32_t denormalize,
                   void* stream);
/* apply_transforms_depth(normalization_method="max_scene_depth"): scene = (mask ? max over mask > 0 of depth : max_scene_depth
 * given) + (mask ? delta : 0); out = clip(depth, 0, scene) / scene, mapped to [-1, 1] when signed_range;
 * ws: mf_minmax_ws_floats() + 2 floats */
int mf_depth_normalize(const float* depth, const float* mask, float* out, int64_t n, float max_scene_depth, float delta,
                       int32_t signed_range, float* ws, void* stream);

/* apply_transforms_depth(normalization_method="percentile") (dataset.py:115-127): the order statistics np.percentile
 * interpolates between, by a two-level radix select (no sort; integer atomics, order-independent).  mf_select_ranks: the nr <= 4
 * order statistics x_(ranks[i]) (0-based ranks in DEVICE memory) -> vals[i] (device); ws: mf_select_ws_bytes() bytes.
 * mf_depth_percentile_normalize: ranks4 = {floor, ceil of 0.02 (n - 1); floor, ceil of 0.98 (n - 1)}, t_lo / t_hi the
 * fractional parts; d2 / d98 interpolated like numpy's 'linear' method, out = (clip(d, d2, d98) - d2) / (d98 - d2), mapped to
 * [-1, 1] when signed_range; vals4: 4 floats of device scratch. */
int64_t mf_select_ws_bytes(void);
int mf_select_ranks(const float* x, int64_t n, const int64_t* ranks, int32_t nr, float* vals, void* ws, void* stream);
int mf_depth_percentile_normalize(const float* depth, float* out, int64_t n, const int64_t* ranks4, float t_lo, float t_hi,
                                  int32_t signed_range, float* vals4, void* ws, void* stream);
/* torchvision Resize(interpolation=BICUBIC, antialias=False) + CenterCrop (+ Normalize) of dataset.py:150-164,184-192 on fp32 planes:
 * PyTorch's plain bicubic kernel (align_corners = False, A = -0.75, no antialiasing: torchvision < 0.17's default for tensors)
 * evaluated inside the crop window only:
 * dst[p][y][x] = a * bicubic(src[p] resized to h_res x w_res)[y + crop_top][x + crop_left] + b */
int mf_bicubic_resize_crop(const float* src, float* dst, int32_t planes, int32_t h_in, int32_t w_in, int32_t h_res, int32_t w_res,
                           int32_t crop_top, int32_t crop_left, int32_t h_out, int32_t w_out, float a, float b, void* stream);
/* The same with PyTorch's ANTIALIASED bicubic (F.interpolate(mode="bicubic", align_corners=False, antialias=True): ATen's
 * _upsample_bicubic2d_aa, Keys kernel a = -0.5 stretched by max(scale, 1), weights normalised per output pixel, width pass then
 * height pass) — what torchvision 0.18 (the reference's pinned version) runs for transforms.Resize(BICUBIC) on a tensor at every
 * scale (dataset.py:150-164,183-192: antialias defaults to True there). */
int mf_bicubic_aa_resize_crop(const float* src, float* dst, int32_t planes, int32_t h_in, int32_t w_in, int32_t h_res, int32_t w_res,
                              int32_t crop_top, int32_t crop_left, int32_t h_out, int32_t w_out, float a, float b, void* stream);
/* y[c][p] = a * x[p][c] + b: HWC -> CHW with Normalize([0.5], [0.5]) (apply_transforms_normals, dataset.py:184-192) */
int mf_hwc_to_chw_affine(const float* x, float* y, int64_t hw, int32_t channels, float a, float b, void* stream);

/* ============================================================================================
 * Training: the backward pass and the optimizer of examples/brushnet/train_brushnet_mirror.py:1459-1466
 * (accelerator.backward -> ATen autograd in the reference; clip_grad_norm_ :1463; torch.optim.AdamW :1188-1200).
 * All tensors fp32.  Data gradients of convolutions / linears need no entry of their own: they are mf_gemm_conv on the
 * transposed (tap-flipped) weight that mf_transpose lays out; strided / upsampling convs add mf_zero_insert2x /
 * mf_sumpool2x2.  Every reduction has a fixed order: gradients are bit-reproducible.
 * ========================================================================================== */

/* Weight gradient of mf_gemm_conv's convolution (same geometry fields): dw[n][(ky*kw + kx)*(c0+c1) + c] (+)= sum over
 * output pixels m of dy[m][n] * A[pixel(m, ky, kx)][c].  dtype MF_F32 (fp32 MFMA) or MF_F16X3 (split precision).
 * Replaces autograd's conv2d / linear weight gradients (torch/nn/grad.py conv2d_weight; linear: dy^T x). */
typedef struct mf_wgrad_desc {
    /* MF_F32 / MF_F16X3 / MF_BF16X1: fp32 a0 / a1 / dy (rounded or split while staged).  MF_BF16: a0 / a1 / dy are bf16 tensors
     * (declared float* for the layout's sake; strides in elements; channel counts multiples of 8) — the pre-rounded operand copies of
     * the bf16x1 training mode: half the operand bytes, no conversion in the staging loop, the same products as MF_BF16X1. */
    int32_t dtype;
    const float* a0; const float* a1;   /* the forward input, NHWC, one or two channel segments */
    int32_t c0, c1;
    int64_t lda0, lda1;
    int32_t batch, h_in, w_in, h_out, w_out, kh, kw, stride, pad_t, pad_l, upsample;
    const float* dy; int64_t lddy;      /* gradient of the conv output, [batch*h_out*w_out][n] */
    int32_t n;
    float* dw; int64_t lddw;            /* [n][kh*kw*(c0+c1)] */
    int32_t accumulate;                 /* 1: dw += ... */
    int32_t splitm;                     /* slabs over the pixel dimension: 0 = heuristic */
    float* ws; int64_t ws_floats;       /* splitm * n * kh*kw*(c0+c1) floats (mf_conv_wgrad_ws_floats) */
} mf_wgrad_desc;
int mf_sizeof_wgrad_desc(void);
int64_t mf_conv_wgrad_ws_floats(const mf_wgrad_desc* d);
int mf_conv_wgrad(const mf_wgrad_desc* d, void* stream);
/* Developer / test switch: 0 routes bf16-input weight gradients to the register-staged kernel instead of the LDS-DMA one (the two are
 * bit-identical; tests compare them).  Process-wide, not thread-safe against concurrent mf_conv_wgrad calls. */
void mf_debug_set_wgrad_dma(int on);

/* base[offs[i] .. offs[i] + lens[i]) = 0 for i < count, one launch (offs / lens: DEVICE arrays of int64, element units).  The training
 * step clears only the small accumulated parameters of the gradient arena this way; conv / linear weight gradients are written by their
 * first mf_conv_wgrad of the step (accumulate = 0) — optimizer.zero_grad() of train_brushnet_mirror.py:1466 without the 2.5 GB memset. */
int mf_zero_ranges(float* base, const int64_t* offs, const int64_t* lens, int32_t count, void* stream);

/* fp32 weight [rows][k] (row stride ldw) -> mf_gemm_desc's w_split layout for MF_F16X3 / MF_BF16X3: out = 16-bit
 * [rows][2 * kp], kp = round_up(k, 32), per 32 k the 32 high halves then the 32 low halves (hi = RNE(w), lo = RNE(w - hi)),
 * zero padded.  The device form of what the inference weights get once on the host: training re-splits the weights the
 * optimizer just changed (models.py / autograd.py), the frozen network's once.  An |w| > 65504 under MF_F16X3 raises the
 * flag mf_split_overflow() reports. */
int mf_split_pack(const float* w, int64_t ldw, void* out, int64_t rows, int32_t k, int32_t dtype, void* stream);

/* y[z][c][r] = x[z][r][c] for nz matrices: element strides ldx / ldy between rows, zsx / zsy between matrices (signed:
 * a negative zsy with y pointing at the last matrix writes the batch in reverse — the tap flip of a dgrad weight) */
int mf_transpose(const float* x, float* y, int32_t nz, int32_t rows, int32_t cols, int64_t ldx, int64_t ldy, int64_t zsx,
                 int64_t zsy, void* stream);
/* The same with a bf16 (nearest-even) result, strides in elements: the transposed, tap-flipped weight of the data-gradient GEMM in the
 * bf16x1 training mode is rounded while it is laid out (one launch instead of mf_transpose + mf_cast_bf16). */
int mf_transpose_bf16(const float* x, void* y, int32_t nz, int32_t rows, int32_t cols, int64_t ldx, int64_t ldy, int64_t zsx,
                      int64_t zsy, void* stream);

/* out[s][j] (+)= sum over the rows of segment s (rows_per_seg consecutive rows) of x[row][j], j < n: bias gradients
 * (one segment), the time-embedding gradient of a resnet (one segment per image), dgamma / dbeta partials */
/* mf_transpose on bf16 in and out (16-byte accesses both ways): cols, ldx, ldy, zsx, zsy multiples of 8, ldy >= rows rounded up to 8
 * (pad columns are written as zeros).  Q^T / K^T / V^T of the MF_BF16X1 attention. */
int mf_transpose_bf16_bf16(const void* x, void* y, int32_t nz, int32_t rows, int32_t cols, int64_t ldx, int64_t ldy, int64_t zsx,
                           int64_t zsy, void* stream);

int64_t mf_colsum_ws_floats(int32_t segs, int64_t rows_per_seg, int32_t n);
int mf_colsum(const float* x, int64_t ldx, float* out, int64_t ldo, int32_t segs, int64_t rows_per_seg, int32_t n,
              int32_t accumulate, float* ws, void* stream);

/* mf_cast_bf16 of a contiguous [segs * rows_per_seg][n] gradient (n % 8 == 0) AND its column sums from the same read: seg_out[s][j]
 * (+)= the sum over segment s (nullable: the time-embedding gradient of a resnet, resnet.py:369-381), tot_out[j] (+)= the sum over all
 * rows (nullable: the bias gradient).  Replaces mf_cast_bf16 + up to two mf_colsum passes in the bf16x1 training mode. */
int64_t mf_cast_bf16_colsum_ws_floats(int32_t segs, int64_t rows_per_seg, int32_t n);
int mf_cast_bf16_colsum(const float* x, void* out16, int32_t segs, int64_t rows_per_seg, int32_t n, float* seg_out, int64_t ldo,
                        int32_t seg_accumulate, float* tot_out, int32_t tot_accumulate, float* ws, void* stream);

/* Backward of mf_groupnorm (GroupNorm + optional SiLU over one or two NHWC segments; native_group_norm_backward +
 * silu_backward in the reference's autograd): dx per segment, per-image dgamma / dbeta partials [batch][c0+c1]
 * (nullable; sum them with mf_colsum). */
typedef struct mf_groupnorm_bwd_desc {
    const float* x0; const float* x1; int32_t c0, c1;
    const float* dy;
    const float* gamma; const float* beta;
    float* dx0; float* dx1;
    float* dgamma_part; float* dbeta_part;
    int32_t batch, hw, groups, silu;
    float eps;
    float* ws;      /* mf_groupnorm_bwd_ws_floats() floats, 16-byte aligned; null = the one-block-per-group kernel only */
    float* dgamma_acc; float* dbeta_acc;   /* [c0 + c1], nullable, INSTEAD of the partials: the gradients summed over the batch
                                            * are ADDED here (needs mf_groupnorm_bwd_streams(...) != 0 and ws) */
    const float* add0; const float* add1;  /* nullable, laid out like dx0 / dx1: dx = gradient + add (the gradient a residual
                                            * connection already left for the same tensor: no separate accumulation pass) */
    const float* stats_in;                 /* nullable [batch][groups][2]: the (mean, rstd) mf_groupnorm wrote to stats_out for the same
                                            * input — the streaming form then skips its statistics pass (one read of x less) */
} mf_groupnorm_bwd_desc;
int mf_sizeof_groupnorm_bwd_desc(void);
int mf_groupnorm_bwd_streams(int32_t batch, int32_t hw, int32_t c0, int32_t c1);   /* 1: this shape runs the streaming form */
int64_t mf_groupnorm_bwd_ws_floats(int32_t batch, int32_t hw, int32_t channels, int32_t groups);
int mf_groupnorm_bwd(const mf_groupnorm_bwd_desc* d, void* stream);

/* LayerNorm backward over [rows][c] (c <= 2048): dx, and per-block dgamma / dbeta partials
 * [mf_layernorm_bwd_parts(rows)][c] (nullable) */
int64_t mf_layernorm_bwd_parts(int64_t rows);
int mf_layernorm_bwd(const float* x, const float* dy, const float* gamma, float* dx, float* dgamma_part, float* dbeta_part,
                     int64_t rows, int32_t c, float eps, const float* add /* nullable [rows][c]: dx = gradient + add */, void* stream);
/* ds = scale * p * (dp - sum_j dp*p) per row of [rows][ld] (valid length cols, pad written 0): softmax backward */
int mf_softmax_bwd(const float* p, const float* dp, float* ds, int64_t rows, int32_t cols, int32_t ld, float scale, void* stream);
int mf_silu_bwd(const float* x, const float* dy, float* dx, int64_t n, void* stream);
/* h = [a | g] ([rows][2c]), out = a * gelu_erf(g): dh from dout ([rows][c]) */
int mf_geglu_bwd(const float* h, const float* dout, float* dh, int64_t rows, int32_t c, void* stream);
/* mf_geglu_bwd on a bf16 pre-activation h16 [rows][2c] (the MF_BF16X1 mode keeps FeedForward's hidden tensor in bf16, as the reference's
 * autocast does: attention.py:617-675 under train_brushnet_mirror.py:902-907,1127-1131): dh16 bf16 [rows][2c]; bias_grad (nullable, fp32 [2c],
 * ADDED to) receives the column sums of dh16 — ff.net.0.proj's bias gradient — from the same pass (ws: mf_geglu_bwd_bf16_ws_floats). */
int64_t mf_geglu_bwd_bf16_ws_floats(int64_t rows, int32_t c);
int mf_geglu_bwd_bf16(const void* h16, const float* dout, void* dh16, int64_t rows, int32_t c, float* bias_grad, float* ws, void* stream);
/* y[b][2h][2w][c]: x at the even positions, zeros elsewhere (a stride-2 conv's data gradient as a stride-1 conv) */
int mf_zero_insert2x(const float* x, float* y, int32_t batch, int32_t h, int32_t w, int32_t c, void* stream);
/* y[b][h][w][c] = sum of the 2x2 block of x[b][2h][2w][c]: backward of the nearest-2x upsample (upsampling.py:170-178) */
int mf_sumpool2x2(const float* x, float* y, int32_t batch, int32_t h, int32_t w, int32_t c, void* stream);
/* gradient of mf_mse_loss's loss[0] with respect to pred */
int mf_mse_grad(const float* pred, const float* target, const float* weights, float* dpred, int32_t rows, int64_t n, void* stream);

/* sum of squares of a flat gradient arena into out[0] (double, device), two fixed-order stages */
int64_t mf_sumsq_ws_doubles(void);
int mf_sumsq(const float* x, int64_t n, double* out, int32_t accumulate, double* ws, void* stream);
/* torch.nn.utils.clip_grad_norm_ (train_brushnet_mirror.py:1463) for an arena that holds gradients x S (S = 1 / unscale, the
 * loss scale of the split-precision backward; unscale = 1 without one): norm = sqrt(sumsq) * unscale -> norm_out[0]
 * (nullable); coef[0] = min(1, max_norm / (norm + 1e-6)) * unscale stays on the device (mf_adamw multiplies the stored
 * gradients with it: no host synchronisation in the step) */
int mf_clip_coef(const double* sumsq, float max_norm, float unscale, float* coef, float* norm_out, void* stream);
/* torch.optim.AdamW (train_brushnet_mirror.py:1188-1200; no amsgrad) over flat arenas of n fp32: the gradient is read as
 * g * grad_scale[0] (grad_scale nullable) */
int mf_adamw(float* w, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
             float weight_decay, int32_t step, const float* grad_scale, void* stream);

/* ---- step programs: a whole forward pass behind ONE entry (mf_brushnet_forward / mf_unet_forward /
 * mf_denoise_step_fused) -----------------------------------------------------------------------------------------------------
 * The sequencing of a pass — which of the entries above, on which buffers, with which tiles — is written down ONCE by the Python
 * side (reflecting_reality_amd/program.py: one eager pass of BrushNetModel.forward, models/brushnet.py:693-936;
 * UNet2DConditionModel.forward, models/unets/unet_2d_condition.py:1037-1311; or the loop body of
 * pipelines/brushnet/pipeline_brushnet.py:1250-1332) into a program file; a host with no Python loads the file, gives every buffer
 * memory, and replays it.  A program is specialised like a hipGraph (shapes, precision, tiles, scalar arguments are the recorded
 * ones) and, like one, is replayed on any stream and may itself be captured into a hipGraph by the host.
 *
 * File: a header of header_bytes (everything mf_program_load parses: magic "MFPROG1", ABI version, buffer table, calls) followed
 * by the bytes of the constant and io buffers at the file offsets mf_program_buffer_info reports.  Buffer kinds:
 *   MF_PROGRAM_CONST      weights in their device layouts, prompt K / V^T, tables, scratch: upload the file's bytes once
 *   MF_PROGRAM_WORKSPACE  intermediates: any device memory of `bytes` (contents undefined between runs, data_offset -1)
 *   MF_PROGRAM_IO         named inputs / outputs ("latents", "coef4", ...): the host's own buffers; the file holds the recorded
 *                         pass's values, so that replaying the file as it is reproduces the recorded pass bit for bit
 * Every buffer must be bound (16-byte aligned device memory of at least `bytes`) before mf_program_run; the library keeps the
 * pointers, never the memory.
 *
 * Streams: a program recorded inside a hipGraph capture (the denoise step: BrushNet on a side stream under the UNet's encoder,
 * pipeline.py::_denoise_graph) carries that capture's forks and joins.  Stream 0 is the `stream` argument of mf_program_run; the
 * others, and the events between them, are created by the first run on the device current then and destroyed by
 * mf_program_destroy — the only device objects this library ever owns.  Every side stream forks from stream 0 and has joined it
 * again when the run's last call is enqueued, so runs on one stream are ordered like single launches and the whole run can be
 * captured (hipStreamBeginCapture on `stream`). */
#define MF_PROGRAM_CONST 0
#define MF_PROGRAM_WORKSPACE 1
#define MF_PROGRAM_IO 2
typedef struct mf_program mf_program;
/* blob: the first header_bytes of the file (bytes 24..31 of the file, little endian); the library copies what it needs */
int mf_program_load(const void* blob, int64_t bytes, mf_program** out);
void mf_program_destroy(mf_program* p);
int32_t mf_program_num_buffers(const mf_program* p);
int mf_program_buffer_info(const mf_program* p, int32_t index, int32_t* kind, int64_t* bytes, int64_t* data_offset, const char** name);
int32_t mf_program_find_buffer(const mf_program* p, const char* name);      /* -1: no buffer of that name */
int mf_program_bind(mf_program* p, int32_t index, void* device_ptr);
int32_t mf_program_num_calls(const mf_program* p);
/* the free-form description the exporter stored (JSON: what was recorded, shapes, precision, steps of the tables); never NULL */
const char* mf_program_meta(const mf_program* p);
int mf_program_run(mf_program* p, void* stream);
/* The loop body of pipeline_brushnet.py:1250-1332 — latent doubling, BrushNet, UNet with the 28 injected residuals, classifier-free
 * guidance, DDIM update — as one call: binds the io buffers "latents" (NCHW fp32, updated IN PLACE), "coef4" (the step's
 * {sqrt_at, sqrt_1m_at, sqrt_ap, dir_coef}, mf_cfg_ddim_step_dev), "temb_unet" / "temb_brushnet" (the step's rows of the two
 * time-embedding tables) and runs the program.  A NULL argument keeps the buffer's current binding.  A step exported under a
 * multistep scheduler (PNDM, UniPC: host-side state between steps, scheduling_pndm.py:321-390) has no "coef4": it ends with the guided
 * noise prediction e = eu + g (ec - eu) in the io buffer "eps" (pipeline_brushnet.py:1310-1312) and the host's scheduler steps from it. */
int mf_denoise_step_fused(mf_program* step, void* latents, const void* coef4, const void* temb_unet, const void* temb_brushnet, void* stream);
/* UNet2DConditionModel.forward (unet_2d_condition.py:1037-1311) with BrushNet's residuals injected (:1172-1177, 1218, 1262-1284):
 * io buffers "sample", "temb", "residual.<i>" (i < n_residuals, the order of brushnet.py:896-936: down, mid, up), "eps" (the noise
 * prediction).  The prompt enters through the program's constants (the cross-attention K / V^T bound when it was recorded). */
int mf_unet_forward(mf_program* unet, const void* sample, const void* temb, const void* const* residuals_in, int32_t n_residuals,
                    void* eps_out, void* stream);
/* BrushNetModel.forward (brushnet.py:693-936): io buffers "sample", "temb", "cond" (the 5-channel conditioning latents of
 * pipeline_brushnet.py:1186-1215) and "residual.<i>" (written). */
int mf_brushnet_forward(mf_program* brushnet, const void* sample, const void* temb, const void* cond, void* const* residuals_out,
                        int32_t n_residuals, void* stream);
/* AutoencoderKL.decode (models/autoencoders/autoencoder_kl.py:294-318, vae.py:285-351; pipeline_brushnet.py:1342 hands it latents /
 * scaling_factor): io buffers "z" (NCHW fp32 latents) and "image" (NCHW fp32, written). */
int mf_vae_decode(mf_program* vae_decoder, const void* z, void* image_out, void* stream);
/* AutoencoderKL.encode up to the posterior's moments (autoencoder_kl.py:256-291, vae.py:137-167 + quant_conv): io buffers "image"
 * (NCHW fp32) and "moments" (NHWC fp32 [b][h/8][w/8][2 * latent_channels]: mean | logvar, what mf_vae_sample reads; written). */
int mf_vae_encode_moments(mf_program* vae_encoder, const void* image, void* moments_out, void* stream);
/* the device copies / fills a recorded pass contains (torch made them between the launches): hipMemcpy2DAsync / hipMemsetAsync */
int mf_memcpy2d(void* dst, int64_t dpitch, const void* src, int64_t spitch, int64_t width_bytes, int64_t height, void* stream);
int mf_memset(void* dst, int32_t value, int64_t bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MFHIP_H */
