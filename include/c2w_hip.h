/* c2w_hip.h -- C ABI of libc2w_hip.so, the MI355X (gfx950) kernel library under
 * climate2weather_amd.
 *
 * The reference (schmidtjonathan/Climate2Weather) is pure Python and has no FFI:
 * its hot path dispatches torch ops.  Each entry point below names
 * the reference call sites whose arithmetic it replaces.  Conventions:
 *   - caller owns every buffer (device pointers from the caller's allocator); nothing is
 *     allocated, freed or synchronised inside; work is enqueued on `stream` (a hipStream_t);
 *   - activations are NHWC: [B][H][W][ld] with `ld` the channel stride in elements;
 *   - `dtype` selects the storage/MFMA-operand type: C2W_DTYPE_F32 (exact fp32 matrix-core
 *     path, the <=1e-4 parity mode), C2W_DTYPE_BF16 (throughput mode) or C2W_DTYPE_F16 (IEEE half, the type the
 *     reference's Fabric "16-mixed" autocast computes in, train.py:98: same kernels and layouts as bf16 with the f16
 *     matrix-core opcode; its narrow range needs the loss scale of c2w_grad_scaler_* in training); accumulation and all
 *     pointwise math are fp32 in all three;
 *   - return value: 0 on success, a positive hipError_t, or a negative C2W_ERR_* code.
 *     Nothing throws across this boundary.
 */
#ifndef C2W_HIP_H
#define C2W_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { C2W_DTYPE_F32 = 0, C2W_DTYPE_BF16 = 1, C2W_DTYPE_F16 = 2 };
enum { C2W_ERR_BAD_ARG = -1, C2W_ERR_BAD_SHAPE = -2, C2W_ERR_UNSUPPORTED = -3 };

/* geometry of the implicit GEMM */
enum {
    C2W_CONV_1X1 = 0, /* Linear / Conv1d(k=1): model/nn.py:45,47,149; model/score.py:56-57 */
    C2W_CONV_S1 = 1,  /* Conv2d 3x3 stride 1 pad 1: model/nn.py:155,157,193-194 (and its dgrad with flipped weights) */
    C2W_CONV_S2 = 2,  /* Conv2d 3x3 stride 2 pad 1: model/nn.py:169-174 */
    C2W_CONV_UP = 3,  /* Upsample(nearest,x2) -> Conv2d 3x3: model/nn.py:184-189, upsample folded into the gather */
    C2W_CONV_TS2 = 4  /* input-gradient of C2W_CONV_S2 (x := dy, y := dx) */
};
/* C2W_ACT_SILU_PAIR (training): with a = the conv result as stored, y = silu(a) and y2 = silu'(a) -- the activation the next
 * conv reads and the factor the backward pass multiplies by (model/nn.py:156 forward/backward), so the pre-activation itself
 * is never written and the backward epilogue needs no transcendental. */
/* C2W_ACT_RELU / C2W_ACT_RELU_PAIR: the same for torch.nn.ReLU, the default `activation` of the reference's UNet (model/nn.py:118):
 * y = max(a, 0), y2 = (a > 0) -- the backward pass multiplies by y2 with C2W_MUL_PLAIN. */
enum { C2W_ACT_NONE = 0, C2W_ACT_SILU = 1, C2W_ACT_SILU_PAIR = 2, C2W_ACT_RELU = 3, C2W_ACT_RELU_PAIR = 4 };
enum { C2W_MUL_PLAIN = 0, C2W_MUL_DSILU = 1 };
enum { C2W_CONV_POOL2 = 1, C2W_CONV_WPACKED = 2, C2W_CONV_NO_Y = 4 }; /* C2wConvArgs.flags (bit set) */

/* y[q][co] = act( sum_{tap,ci} w[co][tap][ci] * x[src(q,tap)][ci] + bias[co] ) (* mul' ) (+ res)
 * q runs over the B*Hout*Wout output pixels in NHWC raster order. */
typedef struct C2wConvArgs {
    const void* x;     /* [B][Hin][Win][Cin]  (Cin = channel stride; multiple of 32 (f32) / 64 (bf16)) */
    const void* w;     /* [wrows][taps][Cin], taps = 1 or 9 (kh*3+kw); rows >= wrows read as zero */
    const float* bias; /* [wrows] fp32 or NULL */
    const void* res;   /* [B*Hout*Wout][ldy] added last, or NULL   (residual / skip: model/nn.py:28,238) */
    const void* mul;   /* [B*Hout*Wout][ldy] or NULL: y *= mul (C2W_MUL_PLAIN) or y *= silu'(mul) (C2W_MUL_DSILU) */
    void* y;           /* [B*Hout*Wout][ldy] */
    void* y2;          /* optional second output [B*Hout*Wout][ldy]: silu(y) (training keeps the pre-activation too) */
    int32_t B, Hin, Win, Cin;
    int32_t Hout, Wout, Cout, ldy;
    int32_t wrows;
    int32_t mode;    /* C2W_CONV_* */
    int32_t act;     /* C2W_ACT_* */
    int32_t mulmode; /* C2W_MUL_* */
    /* Optional fused LayerNorm backward (see c2w_conv_lnbwd_supported): with ln_x != NULL the conv result g is not stored;
     * instead y = res + dLN(g; ln_x + ln_m[b]) and ln_dm[b][c] += sum over the image's pixels of the LN part -- exactly
     * c2w_ln_backward(dy = g, x = ln_x, m = ln_m, dres = res, dx = y, dm = ln_dm) applied to the conv's output tile while
     * it is still on chip (model/nn.py:28,154 backward: the input gradient of conv1 feeds LN's backward directly). */
    const void* ln_x;   /* [B*Hout*Wout][ldy] LN input rows (the block input), or NULL = no fusion */
    const float* ln_m;  /* [B][ln_ldm] fp32 modulation rows added to ln_x before the norm, or NULL */
    float* ln_dm;       /* [B][ln_ldm] fp32 modulation gradient, accumulated (atomics), or NULL */
    int32_t ln_ldm;
    int32_t ln_unbiased;
    float ln_eps;
    int32_t flags;      /* C2W_CONV_POOL2: y is [B][Hout/2][Wout/2][ldy] and receives the 2x2 SUMS of the result -- the adjoint of
                         * Upsample(nearest, x2) (model/nn.py:184) applied to the input gradient of the conv behind it, on chip: the
                         * full-resolution gradient is never written (see c2w_conv_pool2_supported) */
                        /* C2W_CONV_WPACKED: w is the stage-major copy c2w_pack_conv_weights_batched makes ([tap][Cin / 32][rows padded to 128][32 ci],
                         * 8 KiB per K stage of the 16x16-tile kernel in one piece); accepted only where c2w_conv_wpacked_supported says so */
    /* Optional fused LayerNorm FORWARD of the consumer (see c2w_conv_lnfwd_supported): with lnf_y != NULL the kernel also
     * writes lnf_y = LN_C(y + lnf_m[b]) -- c2w_ln_forward(x = y as stored, m = lnf_m, ldm = ln_ldm, eps = ln_eps,
     * unbiased = ln_unbiased) -- i.e. the next residual block's normalised input (model/nn.py:28,154) leaves the conv that
     * produced the block input, without a separate pass over it. */
    void* lnf_y;        /* [B*Hout*Wout][ldy] or NULL */
    const float* lnf_m; /* [B][ln_ldm] fp32 modulation rows of the CONSUMER block, or NULL (plain LayerNorm) */
    /* Optional promise about channel padding: 0, or the number of leading input channels that can be non-zero -- channels kvalid..Cin-1
     * of x (or of w) are ALL ZERO (the network-input conv at C = 65 reads rows padded to 128 channels, model/nn.py:193; the input
     * gradient of the output conv reads a gradient whose padding channels are zero, :194).  Kernels may skip the multiplications
     * the promise makes void (conv_patch_t3_kernel: whole 32-channel half chunks); results are unchanged. */
    int32_t kvalid;
    /* Optional per-pixel statistics of the fused LayerNorms (16-bit launches; both NULL: rounds 1-4 behaviour):
     *   lnf_rstd (with lnf_y): the forward also writes 1/sqrt(var + eps) of every pixel row it normalises, [B*Hout*Wout] fp32;
     *   ln_rstd (with ln_x): the backward is handed what the forward kept -- ln_x then holds the NORMALISED rows (the lnf_y the
     *   forward wrote = the conv input of model/nn.py:155, which training keeps anyway) and ln_rstd their 1/sigma; ln_m is not
     *   read.  dLN = rstd * (g - mean(g) - xhat * sum(g * xhat) / den) needs neither the mean nor the variance again: two of the
     *   four 16-lane reductions per pixel row and the modulation add leave the epilogue (model/nn.py:28,154 backward). */
    float* lnf_rstd;
    const float* ln_rstd;
    /* Optional, with lnf_y (round 6; see c2w_conv_lnfwd_chain_supported) -- a chain of residual blocks (model/nn.py:27-28,146-159) whose
     * intermediate outputs are never written: block k's second conv reads its residual x_k, adds its result, and emits the next
     * block's normalised input h_{k+1} = LN(x_{k+1} + m_{k+1}) together with that LayerNorm's per-pixel mean and 1/sigma; x_{k+1} itself
     * is read by nothing but block k+1's residual add, which can rebuild it from what was emitted:  x = h / rstd + mean - m.
     *   lnf_mean:  [B*Hout*Wout] fp32, the mean of every pixel row this launch normalises (next to lnf_rstd), or NULL;
     *   res_rstd, res_mean, res_m (all or none): `res` holds NORMALISED rows h = LN(x + res_m[b]) (an earlier launch's lnf_y) and these
     *     are that LayerNorm's lnf_rstd / lnf_mean and its modulation rows (stride ln_ldm, NULL = none): the residual added is
     *     h / res_rstd + res_mean - res_m[b];
     *   flags & C2W_CONV_NO_Y: y is not written (pass any valid pointer); lnf_y then normalises the fp32 sum instead of its 16-bit
     *     rounding.  537 MB less per launch at 128 channels x 128^2 x 128 windows. */
    float* lnf_mean;
    const float* res_rstd;
    const float* res_mean;
    const float* res_m;
    /* Optional fused training loss (round 6; see c2w_conv_loss_supported): with loss_sum != NULL the conv result itself is not
     * stored.  Instead, with eps = the noise rows loss_eps ([B*Hout*Wout][loss_lde] IEEE half, NHWC like y; written by
     * c2w_nchw_to_nhwc_noise_rows next to the noised input they were mixed into) and a = the result rounded to the storage type,
     * loss_sum += sum (a - eps)^2 over the loss_C real channels and y = (a - eps) * loss_gscale [* loss_scaler[0]] (channels >=
     * loss_C: zero) -- c2w_mse_loss_grad applied to the network-output conv's tile while it is still on chip
     * (src/thor/pipelines.py:35, training_loop.py:376-377: the loss and the gradient the backward pass starts from; the prediction
     * is used by nothing else in a training step). */
    float* loss_sum;          /* one fp32, accumulated (atomics), or NULL = no fusion */
    const float* loss_scaler; /* device-resident loss scale (fp16 training, c2w_grad_scaler_*) or NULL */
    const void* loss_eps;     /* half rows, channels loss_C .. loss_lde-1 zero */
    float loss_gscale;
    int32_t loss_C;
    int32_t loss_lde;         /* channel stride of loss_eps: a multiple of 8, loss_C <= loss_lde <= 128 */
    /* Optional split-K (round 6; c2w_conv_splitk_plan): splitk > 1 deals every output tile's K chunks to that many workgroups; their fp32
     * partial tiles go through the caller's scratch splitk_ws (splitk_ws_bytes >= what the plan asked for; contents undefined before and
     * after; one buffer per stream, like c2w_conv_wgrad's workspace) and a second launch on the same stream adds them in a fixed order
     * and applies bias / activation / mul / res.  For launches of fewer workgroups than the chip has CUs (the deep levels of a
     * sampler step on a short trajectory, exp/configs/000_on-model-eval/s16_t6.yml: 37 windows): such a launch costs its chain of K
     * stages whatever it carries.  Results are bit-reproducible; they differ from the unsplit launch in summation order only. */
    float* splitk_ws;
    unsigned long long splitk_ws_bytes;
    int32_t splitk;           /* 0 / 1: no split; else exactly c2w_conv_splitk_plan's answer for these arguments */
} C2wConvArgs;

/* 1 when c2w_conv_forward / c2w_conv_wgrad run this geometry on the halo-patch kernels (3x3 stride-1, image tiled exactly
 * by 8 x 16-pixel tiles), 0 when it takes the general gather kernels. */
int c2w_conv_patch_supported(const C2wConvArgs* args, int dtype);
/* 1 when c2w_conv_forward can store the 2x2-pooled result (flags & C2W_CONV_POOL2): halo-patch geometry, stride 1, no
 * res / mul / y2 / activation / LayerNorm fusion.  Otherwise callers run the conv and c2w_sumpool2. */
int c2w_conv_pool2_supported(const C2wConvArgs* args, int dtype);
/* 1 when c2w_conv_forward can also emit the consumer's LayerNorm (lnf_y): same shape conditions as the backward fusion,
 * residual allowed. */
int c2w_conv_lnfwd_supported(const C2wConvArgs* args, int dtype);
/* 1 when c2w_conv_forward can run args with the fused LayerNorm backward (bf16, Cout == ldy == 128, 3x3 stride-1 on an
 * image the halo-patch kernel tiles, no mul / act / y2), else 0.  Callers fall back to conv + c2w_ln_backward. */
int c2w_conv_lnbwd_supported(const C2wConvArgs* args, int dtype);
/* 1 when c2w_conv_forward takes lnf_mean / res_rstd + res_mean + res_m / C2W_CONV_NO_Y for this geometry (c2w_conv_lnfwd_supported's
 * conditions on the 16x16-tile kernel).  Elsewhere those fields make the call fail with C2W_ERR_BAD_SHAPE. */
int c2w_conv_lnfwd_chain_supported(const C2wConvArgs* args, int dtype);
/* 1 when c2w_conv_forward can run args with the fused training loss (loss_sum / loss_seed / loss_gscale / loss_C): 16-bit, 3x3
 * stride-1 on the 16x16-tile kernel, at most 80 weight rows in rows of 128 channels (the network-output conv, model/nn.py:194),
 * loss_C <= wrows, no res / mul / act / y2 / LayerNorm fusion.  Otherwise callers run the conv and c2w_mse_loss_grad[_noise]. */
int c2w_conv_loss_supported(const C2wConvArgs* args, int dtype);
/* Workgroups per output tile c2w_conv_forward can deal the K chunks of these arguments to (1: no split -- geometry outside the 8x16-tile
 * kernels, fused epilogues other than bias / activation / mul / res, or a launch that already has a workgroup per CU), and through
 * *ws_bytes (may be NULL) the scratch it then needs.  A pure function of the arguments' geometry, epilogue fields and the dtype. */
int c2w_conv_splitk_plan(const C2wConvArgs* args, int dtype, unsigned long long* ws_bytes);

/* Which kernel family c2w_conv_forward (naive == 0) / c2w_conv_wgrad run these arguments on -- a pure function of the geometry,
 * the dtype and the fusion fields; the parity tests assert with it that a case reaches the kernel it is meant to cover.
 * GATHER: conv_igemm / wgrad gather kernels; PATCH_8X16: conv_patch_half_kernel / wgrad_patch_kernel; PATCH_16X16:
 * conv_patch_t3_kernel<16> (16-bit launches of >= 512 workgroups of that tile, C2W_CONV_T3_MIN_WGS); PATCH_PAIR: 8-pixel-wide images, two per tile; PATCH_TS2: the
 * stride-2 input gradient per output-parity class; PATCH_S2: the stride-2 FORWARD on the parity planes of the halo patch (16-bit; output
 * tiled by 8x16 pixels, or 8 pixels wide with two images per tile; from four K chunks on or up to 2048 workgroups). */
enum { C2W_KERNEL_GATHER = 0, C2W_KERNEL_PATCH_8X16 = 1, C2W_KERNEL_PATCH_16X16 = 2, C2W_KERNEL_PATCH_PAIR = 3, C2W_KERNEL_PATCH_TS2 = 4, C2W_KERNEL_PATCH_S2 = 5 };
int c2w_conv_dispatch(const C2wConvArgs* args, int dtype);
int c2w_conv_wgrad_dispatch(const C2wConvArgs* args, int dtype);

/* naive == 0: product path (halo-patch MFMA kernel for 3x3 stride-1 on 16x16-tileable images, general gather MFMA
 * kernel otherwise); naive == 2: force the gather MFMA kernel; naive == 1: one-thread-per-output direct convolution
 * with identical semantics (debug cross-check only). */
int c2w_conv_forward(const C2wConvArgs* args, int dtype, int naive, void* stream);

/* dW[co][tap][ci] (fp32) += sum_q dY[q][co] * x[src(q,tap)][ci]  -- weight gradient of the same geometry.
 * Pass the forward call's block with y := dY; w/bias/res/mul/act are ignored.  The reduction over pixels is split across
 * workgroups: zero dw (or leave the value to accumulate onto) beforehand.
 * dbias (may be NULL): [Cout] fp32, += sum_q dY[q][co] (the bias gradient, from the same pass over dY).
 * workspace / workspace_bytes: caller-owned device scratch (16-byte aligned) for the split's partial sums, handed over PER CALL --
 * the library keeps no pointer, so the entry point is re-entrant: with it the workgroups store their partial tiles and a second
 * launch on the same stream reduces them in a fixed order (75 MB of coalesced stores + reads per launch instead of 75 MB of fp32
 * atomics; dw is then bit-reproducible run to run -- dbias is NOT: every split workgroup adds its column sums with fp32 atomics); NULL, or fewer bytes than c2w_conv_wgrad_workspace_bytes() asks for: fp32 atomics.
 * Contents are undefined before and after the call.  One buffer must not be handed to launches that may run concurrently
 * (different streams without an ordering between them): give each stream its own.
 * Replaces autograd's weight/bias backward of every Conv2d/Conv1d/Linear cited above. */
int c2w_conv_wgrad(const C2wConvArgs* args, float* dw, float* dbias, void* workspace, unsigned long long workspace_bytes, int dtype,
                   void* stream);
/* Scratch bytes c2w_conv_wgrad uses for this geometry and dtype (0: no split), or a negative C2W_ERR_* status. */
long long c2w_conv_wgrad_workspace_bytes(const C2wConvArgs* args, int dtype);

/* The weight gradients of n layers that share ONE geometry (`args`: the forward block of any of them; x / y are ignored) as one launch:
 * items[i] = {x, dy, dw, dbias} of layer i, each with the meaning c2w_conv_wgrad gives them (dw, dbias accumulated into).  The
 * residual-block convs of a UNet level (model/nn.py:146-159: 6 or 12 layers of one shape) have their output gradients one after the
 * other during the backward pass and independent weight gradients; a launch per layer must split its pixel reduction over the whole
 * chip (at 8x8: 8 K tiles per workgroup, each followed by 295 KB of partial sums), together the layers fill it with a fraction of
 * the splits.  `items` is a HOST array, copied into the kernel arguments (nothing is kept).  The dw results are deterministic (dbias: fp32 atomics across the split workgroups, last-bit order noise) (a fixed
 * reduction order) but differ in rounding from n single calls (another split of the same sum).
 * c2w_conv_wgrad_grouped_supported: 1 when the n layers run as one launch (2 <= n <= 16; the halo-patch geometries on
 * wgrad_patch_group_kernel, 1x1 layers -- the attention level's qkv / proj_out, model/nn.py:45,47 -- on wgrad_group_kernel); otherwise
 * the call returns C2W_ERR_UNSUPPORTED and the caller issues n c2w_conv_wgrad calls.  workspace must hold
 * c2w_conv_wgrad_grouped_workspace_bytes(args, n, dtype) bytes (0 when the plan does not split). */
typedef struct C2wWgradItem {
    const void* x;
    const void* dy;
    float* dw;
    float* dbias; /* may be NULL */
} C2wWgradItem;
int c2w_conv_wgrad_grouped_supported(const C2wConvArgs* args, int n, int dtype);
long long c2w_conv_wgrad_grouped_workspace_bytes(const C2wConvArgs* args, int n, int dtype);
int c2w_conv_wgrad_grouped(const C2wConvArgs* args, const C2wWgradItem* items, int n, void* workspace, unsigned long long workspace_bytes,
                           int dtype, void* stream);

/* y = LN_C(x + m[b]): parameter-free channel LayerNorm (zuko.nn.LayerNorm at model/nn.py:44,154,183) fused with
 * the time-modulation add of model/nn.py:28.  x,y: [npix][C]; m: fp32 rows of C (row b = pixel / HW, stride ldm;
 * ldm == 0 -> one row shared by every pixel; m == NULL -> no add).  unbiased selects the N-1 variance. */
int c2w_ln_forward(const void* x, const float* m, void* y, long long npix, int HW, int C, int ldm, float eps,
                   int unbiased, int dtype, void* stream);
/* dx = dres + dLN(dy; x+m);  dm[b][c] += sum over the image's pixels of the LN part (fp32 atomics; dm may be NULL). */
int c2w_ln_backward(const void* dy, const void* x, const float* m, const void* dres, void* dx, float* dm,
                    long long npix, int HW, int C, int ldm, float eps, int unbiased, int dtype, void* stream);

/* out[c] += sum_rows a[row][c]   (bias gradients; fp32 atomics) */
int c2w_colsum(const void* a, float* out, long long rows, int C, int lda, int dtype, void* stream);
/* y = silu(x) ; dx = dy * silu'(x)   (the activation train.py:171 passes; model/nn.py:156, model/score.py:63,67) */
int c2w_silu(const void* x, void* y, long long n, int dtype, void* stream);
int c2w_silu_backward(const void* x, const void* dy, void* dx, long long n, int dtype, void* stream);
/* adjoint of Upsample(nearest, x2) (model/nn.py:184): dx[b][h][w] = sum of the 2x2 block of g ([B][2H][2W][C]) */
int c2w_sumpool2(const void* g, void* dx, int B, int H, int W, int C, int dtype, void* stream);

/* Upsample(nearest, x2) (model/nn.py:184) on NHWC rows: y[b][2h+i][2w+j] = x[b][h][w] */
int c2w_upsample2(const void* x, void* y, int B, int H, int W, int C, int dtype, void* stream);
/* NCHW fp32 <-> NHWC (padded to ldc channels).  With eps != NULL the forward noise process of
 * src/thor/pipelines.py:22-25 is fused: y = mu[b] x + sigma[b] eps, musig = {mu_0, sigma_0, mu_1, ...}. */
int c2w_nchw_to_nhwc(const float* x, const float* eps, const float* musig, void* y, int B, int C, int HW, int ldc,
                     int dtype, void* stream);
int c2w_nhwc_to_nchw(const void* y, float* out, int B, int C, int HW, int ldc, int dtype, void* stream);
/* c2w_nchw_to_nhwc_noise (below) that also KEEPS the noise it mixed in: erows [B*HW][lde] IEEE half (lde = a multiple of 8 >= C,
 * channels C .. lde-1 zero) receives eps rounded to half precision, and y = mu x + sigma * (that rounded value) -- the step's noise IS
 * the rounded stream, so the loss tail (C2wConvArgs.loss_eps) reads back exactly what the input was noised with instead of running
 * Philox + Box-Muller a second time (0.2 ms of VALU work per 128 windows).  img_off (may be NULL): as in c2w_windows_to_nhwc_noise.
 * HW % 4 == 0.  C2W_ERR_UNSUPPORTED: shape outside the kernel (callers fall back to the *_noise pair). */
int c2w_nchw_to_nhwc_noise_rows(const float* x, const long long* img_off, unsigned long long seed, const float* musig, void* y, void* erows,
                                int B, int C, int HW, int ldc, int lde, int dtype, void* stream);
/* loss tail (src/thor/pipelines.py:35, training_loop.py:377): loss_sum += sum (y-eps)^2 ; dy = (y-eps) * gscale */
int c2w_mse_loss_grad(const void* y, const float* eps, void* dy, float* loss_sum, int B, int C, int HW, int ldc,
                      float gscale, int dtype, void* stream);
/* same with dy additionally multiplied by the device-resident loss scale scaler_state[0] (NULL = 1): fp16 training */
int c2w_mse_loss_grad_scaled(const void* y, const float* eps, void* dy, float* loss_sum, int B, int C, int HW, int ldc,
                             float gscale, const float* scaler_state, int dtype, void* stream);
/* Regenerated noise.  The training step's eps = randn_like(x) (src/thor/pipelines.py:22-25) is consumed twice -- by the forward
 * process x_t = mu x + sigma eps and by the loss (eps_pred - eps)^2 -- and nowhere else.  A counter-based generator (Philox4x32-10 on
 * the index of a block of four consecutive elements, Box-Muller) lets both kernels compute eps[e] instead of reading it:
 *   c2w_philox_normal        out[e] = N(0,1) stream of `seed`                                 (tests; fallback for shapes the fused
 *                                                                                             kernels do not take)
 *   c2w_nchw_to_nhwc_noise   c2w_nchw_to_nhwc with eps := that stream (element index = NCHW linear index)
 *   c2w_mse_loss_grad_noise  c2w_mse_loss_grad_scaled with eps := that stream
 * Same seed => bit-identical eps in all three.  C2W_ERR_UNSUPPORTED: channel rows too wide for the LDS tile (use the fallback). */
int c2w_philox_normal(float* out, long long n, unsigned long long seed, void* stream);
int c2w_nchw_to_nhwc_noise(const float* x, unsigned long long seed, const float* musig, void* y, int B, int C, int HW, int ldc,
                           int dtype, void* stream);
/* The same with the B images picked out of a dataset array: image b is the C*HW contiguous floats at data + img_off[b] (a window of
 * consecutive frames: dataset.py:114-126; img_off[b] a multiple of 4).  The noise stream is addressed with the dense (B,C,H,W) index,
 * so the result equals c2w_nchw_to_nhwc_noise on the gathered batch bit for bit -- the batch tensor itself is never written. */
int c2w_windows_to_nhwc_noise(const float* data, const long long* img_off, unsigned long long seed, const float* musig, void* y, int B, int C,
                              int HW, int ldc, int dtype, void* stream);
int c2w_mse_loss_grad_noise(const void* y, unsigned long long seed, void* dy, float* loss_sum, int B, int C, int HW, int ldc,
                            float gscale, const float* scaler_state, int dtype, void* stream);
/* The UNREDUCED loss tensor src/thor/pipelines.py:35 returns, `(eps_pred - eps) ** 2` (the caller takes .mean(), training_loop.py:377),
 * for the module path: out[b][c][px] (NCHW fp32) = (y[b][px][c] - eps[b][c][px])^2 from the network's NHWC output rows; eps read
 * from memory (c2w_sq_err) or regenerated from the step's stream (c2w_sq_err_noise); loss_sum (optional) += the sum of `out`, so
 * the caller's mean costs no second pass.  C2W_ERR_UNSUPPORTED: HW % 4 != 0 or rows too wide for the LDS tile (caller converts the
 * layout and uses tensor arithmetic). */
int c2w_sq_err(const void* y, const float* eps, float* out, float* loss_sum, int B, int C, int HW, int ldc, int dtype, void* stream);
int c2w_sq_err_noise(const void* y, unsigned long long seed, float* out, float* loss_sum, int B, int C, int HW, int ldc, int dtype,
                     void* stream);
/* The network's output convolution (model/nn.py:194: 3x3, stride 1, zero padding) restricted to what the sampler's fold keeps
 * (src/thor/score.py:76-88: of a window's w * F output channels only the centre frame's F, all of them only for the first / last
 * window of a trajectory): rows r0 .. r0 + nr - 1 (nr <= 16) of the [wrows][9][Cin] weight matrix `w` over the NHWC rows `x`
 * ([B][H][W][Cin], 16-bit), bias added, rounded through the compute type like the full convolution's output, stored as fp32 planes:
 *     out[b * ostride + c * H * W + pix],  c < nr
 * -- with out = eps + (i0 + k) * F * H * W, ostride = F * H * W, r0 = k * F, nr = F the centre frames of windows i0 .. i0 + B - 1
 * land where fold() puts them, and no pass over the 128-channel output rows remains.  c2w_conv_center_supported: bf16 / fp16,
 * H % 8 == 0, W % 16 == 0, Cin in {64, 128}. */
int c2w_conv_center_supported(int H, int W, int Cin, int nr, int dtype);
int c2w_conv_center(const void* x, const void* w, const float* bias, float* out, int B, int H, int W, int Cin, int wrows, int r0, int nr,
                    long long ostride, int dtype, void* stream);
/* One-row fp32 Linear: y[r] = act(bias[r] + W[r][0..K) . x), W rows `ldk` floats apart; act = C2W_ACT_NONE / SILU / RELU.  The time
 * embedding MLP and the modulation projections (model/score.py:56-57,62-67; model/nn.py:149) when t is one value for the whole batch
 * (every network call of the sampler). */
int c2w_gemv_f32(const float* x, const float* W, const float* bias, float* y, int rows, int K, int ldk, int act, void* stream);
/* training_loop.py:385 (`loss.detach().item()` after optimizer.step()): the fp32 device scalar `src` is copied into host_slot[0]
 * (its bits) and host_slot[1] = seq is stored after it, release at system scope.  host_slot: two ints of pinned, device-visible host
 * memory; the host polls host_slot[1] for `seq` instead of synchronising the stream the value was produced on. */
int c2w_publish_scalar(const float* src, int* host_slot, int seq, void* stream);
/* model/score.py:14-34 */
int c2w_timestep_embedding(const float* t, float* out, int n, int dim, float max_period, void* stream);
/* src/thor/pipelines.py:13-20: musig[i] = {mu(t_i), sigma(t_i)} */
int c2w_mu_sigma(const float* t, float* musig, int n, float eta, void* stream);
int c2w_cast_f32(const float* in, void* out, long long n, int dtype, void* stream);
/* w[r][tap][k] (fp32, k-stride ldk) -> out[k][tap'][r] (dtype, r-stride ldr), taps reversed when flip: the operand of the
 * input-gradient convolution (dgrad = the same implicit GEMM over dy with these weights). */
int c2w_weight_transpose(const float* w, void* out, int R, int NT, int K, int ldk, int ldr, int flip, int dtype,
                         void* stream);
/* the same for nconv weight matrices in one launch: desc (device memory) holds 8 int64 per matrix
 * {w_off, out_off, R, NT, K, ldk, ldr, flip}, offsets in elements into flat / out */
int c2w_weight_transpose_batched(const float* flat, void* out, const long long* desc, int nconv, int dtype, void* stream);
/* Stage-major copies of 16-bit 3x3 conv weights for the 16x16-tile kernel (C2W_CONV_WPACKED), n matrices in one launch:
 * desc[j] = {src_off, dst_off, rows, cin} (offsets in ELEMENTS into src / dst; src matrix [rows][9][cin], cin a multiple of 32):
 *   dst[((tap * (cin / 32) + h) * rows_pad + r) * 32 + ((q ^ swz(r)) * 8) + e] = src[r][tap][h * 32 + q * 8 + e],  rows_pad = rows
 *   rounded up to 128, zero for r >= rows, swz(r) = (4 - ((r >> 2) & 3)) & 3 (the kernel's LDS swizzle, baked in).
 * The kernel then fetches the 128 rows x 64 B of a (tap, 32-channel half) as ONE contiguous 8 KiB block instead of 128 row pieces
 * 9 * cin * 2 bytes apart: 4.7 % of the dominant launch (profiles/r03_experiments.md section 11).  Size of a packed matrix:
 * 9 * cin * rows_pad elements. */
int c2w_pack_conv_weights_batched(const void* src, void* dst, const long long* desc, int n, int dtype, void* stream);
/* 1 when c2w_conv_forward takes args with C2W_CONV_WPACKED set (the launch goes to the 16x16-tile halo-patch kernel), else 0. */
int c2w_conv_wpacked_supported(const C2wConvArgs* args, int dtype);
/* fused torch.optim.AdamW step (train.py:176-181) + EMA (src/thor/ema.py:23-27) + bf16 shadow refresh over a flat
 * parameter buffer; ema / shadow_bf16 may be NULL; g is multiplied by grad_scale first. */
int c2w_adamw_ema(float* p, const float* g, float* m, float* v, float* ema, void* shadow_bf16, long long n, float lr,
                  float beta1, float beta2, float eps, float weight_decay, int step, float ema_rate, float grad_scale,
                  void* stream);

/* The same step with a 16-bit weight shadow of either format (shadow_dtype = C2W_DTYPE_BF16 / C2W_DTYPE_F16) and, when
 * scaler_state != NULL, under the dynamic loss scale: g is divided by scaler_state[0]; if scaler_state[2] (found_inf) is
 * set the step changes nothing but the EMA; the Adam bias corrections use scaler_state[3] + 1 (steps actually taken)
 * instead of `step`. */
int c2w_adamw_ema_scaled(float* p, const float* g, float* m, float* v, float* ema, void* shadow, int shadow_dtype,
                         long long n, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                         float ema_rate, float grad_scale, const float* scaler_state, void* stream);

/* Dynamic loss scale for fp16 training, resident on the device: torch.cuda.amp.GradScaler's rule, which Fabric's
 * precision="16-mixed" (train.py:98) wraps around fabric.backward / optimizer.step (training_loop.py:378,383).
 * state: 4 floats in device memory {scale, growth tracker, found_inf, optimizer steps taken}.  One step =
 *   loss gradient x state[0] (c2w_mse_loss_grad_scaled) -> backward -> [all-reduce] -> c2w_grad_scaler_check (found_inf = any
 *   non-finite gradient; identical on every rank after the all-reduce) -> c2w_adamw_ema_scaled -> c2w_grad_scaler_update
 *   (overflow: scale *= backoff, tracker = 0; else steps += 1, tracker += 1, and scale *= growth every growth_interval
 *   clean steps).  No host synchronisation anywhere; the host may read the state whenever it likes. */
int c2w_grad_scaler_init(float* state, float init_scale, void* stream);
int c2w_grad_scaler_check(const float* g, long long n, float* state, void* stream);
int c2w_grad_scaler_update(float* state, float growth, float backoff, int growth_interval, void* stream);

/* p_ema = rate * p_ema + (1 - rate) * p over a flat buffer (src/thor/ema.py:23-27) */
int c2w_ema_update(float* ema, const float* p, long long n, float rate, void* stream);

/* QKVAttention, one head (model/nn.py:62-85).  qkv: [B][T][3C] (q|k|v), o: [B][T][C], lse: [B][T] fp32 (may be NULL
 * for inference).  backward recomputes probabilities from lse; delta_ws is a [B][T] fp32 scratch. */
int c2w_attention_forward(const void* qkv, void* o, float* lse, int B, int T, int C, int dtype, void* stream);
int c2w_attention_backward(const void* qkv, const void* o, const void* d_o, const float* lse, float* delta_ws, void* dqkv,
                           int B, int T, int C, int dtype, void* stream);

/* ---- device-resident sampler (replaces the CPU-resident state + per-batch PCIe round trips of
 * src/thor/score.py:156-185 and the elementwise updates of src/thor/pipelines.py:41-91) ---- */
/* unfold, src/thor/score.py:68-74: windows i0..i0+nw-1 of x[L][F][HW] (fp32) -> NHWC rows [nw*HW][ldc], channel = tau*F + c */
int c2w_window_gather(const float* x, void* y, int nw, int F, int HW, int k, int i0, int ldc, int dtype, void* stream);
/* fold, src/thor/score.py:76-88: centre frame of every window (+ leading k of window 0, trailing k of the last) -> eps[L][F][HW] */
int c2w_window_scatter(const void* y, float* eps, int nw, int F, int HW, int k, int i0, int nwin_total, int ldc,
                       int dtype, void* stream);
/* predictor, src/thor/pipelines.py:41-46: x = a*x + b*eps with a = mu'/mu, b = sigma' - mu' sigma/mu; non-finite -> *nan_flag |= 1 */
int c2w_sampler_predict(float* x, const float* eps, int* nan_flag, long long n, float a, float b, void* stream);
/* out[0] += sum v^2 */
int c2w_sumsq(const float* v, float* out, long long n, void* stream);
/* corrector, src/thor/pipelines.py:81-88: delta = tau/(sumsq[0]/n); x -= (delta eps + sqrt(2 delta) z) sigma_next */
int c2w_sampler_correct(float* x, const float* eps, const float* z, const float* sumsq, int* nan_flag, long long n,
                        float tau, float sigma_next, void* stream);
/* likelihood guidance for A = AvgPool2d(s_step) o x[::t_step] (exp/downscaling.py:129-132) with exact_grad=False
 * (src/thor/score.py:24-57): eps -= sigma/mu * A^T((y - A((x - sigma eps)/mu)) / (std_c^2 + gamma (sigma/mu)^2)), in place */
int c2w_guidance(const float* x, float* eps, const float* yobs, const float* stdv, int nobs, int F, int H, int W,
                 int s_step, int t_step, float mu, float sigma, float gamma, void* stream);
/* the same with one gamma per variable, gammav[F] in device memory: what exp/downscaling.py:228-233 hands condition_on for a
 * list-valued `likelihood_gamma` (a (1, C, 1, 1) tensor that src/thor/score.py:55 broadcasts against err) */
int c2w_guidance_per_variable(const float* x, float* eps, const float* yobs, const float* stdv, const float* gammav, int nobs,
                              int F, int H, int W, int s_step, int t_step, float mu, float sigma, void* stream);
/* the measurement operator itself: y[o][c][ph][pw] = mean of the s x s cell of x[o*t_step][c] (exp/downscaling.py:129-132) */
int c2w_pool_stride(const float* x, float* y, int nobs, int F, int H, int W, int s_step, int t_step, void* stream);
/* per-variable affine map over (planes = L*F) planes of HW values: y = x * scale[c] + shift[c], c = plane % F -- the quantile
 * normalisation / de-normalisation of data/pipeline.py:183-244 either side of the sampler (x == y allowed) */
int c2w_affine_channels(const float* x, float* y, const float* scale, const float* shift, long long planes, int F, int HW,
                        void* stream);

/* library identity: returns the gfx target string the kernels were compiled for ("gfx950") */
/* Run-time knobs (dispatch overrides for tests and A/B measurements: C2W_FORCE_GATHER, C2W_CONV_T3, C2W_CONV_PAIR, C2W_CONV_TS2_PATCH,
 * C2W_NO_UP_PATCH, C2W_NO_POOL2, C2W_NO_LN_FUSION, C2W_NO_LNF, C2W_WGRAD_ATOMICS, C2W_ATTN_VALU; csrc/knobs.h) are
 * read from the environment once, when the library is loaded; this re-reads them. */
void c2w_knobs_reload(void);

const char* c2w_target(void);
/* provenance: hex sha256 over the sources (csrc/, include/c2w_hip.h, compile flags) this library was built from; the Python loader
 * refuses a library whose digest differs from the source tree next to it (climate2weather_amd/build.py) */
const char* c2w_sources_sha256(void);

#ifdef __cplusplus
}
#endif
#endif /* C2W_HIP_H */
