/* stx_hip.h -- C-ABI of libstx_hip.so: the MI355X (gfx950) cost-volume hot path of
 * xxxupeng/stereo_toolbox (PSMNet / GwcNet / ACVNet).
 *
 * The reference is 100 % Python/PyTorch and has no FFI of its own; each entry point below replaces a
 * *torch-op sequence* of the reference (file:line relative to /root/reference/stereo_toolbox).  This
 * is the boundary a maintainer binds (ctypes stub in INTEGRATION.md; stereo_toolbox_amd/_capi.py is
 * the binding this repo ships).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to dense fp32 data unless noted; `stream` is a hipStream_t
 *     (NULL = default stream); calls are asynchronous on that stream, never synchronise, never
 *     allocate; all scratch memory is passed in by the caller;
 *   - 2-D features are NCHW [B][C][H][W]; 3-D activations are channels-last [B][D][H][W][C];
 *   - return value 0 = OK; 1 = invalid argument; 2 = launch failure; stx_last_error() gives the
 *     message of the last failure on the calling thread;
 *   - re-entrant: no mutable state beyond the tuning switches below, safe from several host threads / one process
 *     per GPU.
 */
#ifndef STX_HIP_H
#define STX_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

const char* stx_last_error(void);
const char* stx_build_info(void);
/* Tuning / A-B switches (names = their environment variables, e.g. "STX_MARCH_BS"; list and defaults: StxTune in
 * csrc/stx_common.h).  The environment is read ONCE, when the library is loaded; stx_set_tuning changes a switch for the
 * calls that follow in this process (tests, tools/kernel_bench.py).  None changes a result beyond fp32 rounding.
 * stx_get_tuning: value, or -1 for an unknown name;  stx_set_tuning: 0, or 1 for an unknown name. */
int stx_get_tuning(const char* name);
int stx_set_tuning(const char* name, int value);

/* ---- cost-volume builders ------------------------------------------------------------------
 * build_gwc_volume / groupwise_correlation  models/GwcNet/submodule.py:44-63 (dup ACVNet/submodule.py:209-238)
 * build_concat_volume                       models/GwcNet/submodule.py:30-41, PSMNet/stackhourglass.py:111-120
 *                                           (mask_left=1), ACVNet/submodule.py:180-191 (mask_left=0)
 * torch.cat((gwc, concat), 1)               models/GwcNet/gwcnet.py:180 (fused)
 * softmax(att, dim=2) * concat_volume       models/ACVNet/acv.py:196 (`scale` = softmax probabilities [B][D][H][W], or NULL)
 * vol: [B][D][H][W][G + 2*Cc].  Lg/Rg: [B][Cg][H][W] (NULL when G == 0); Lc/Rc: [B][Cc][H][W] (NULL when Cc == 0).
 * Requires Cg % G == 0 (reference assert submodule.py:46), Cg/G in {4,8,12,16} (and 20 / 28 with 4..16 groups: FoundationStereo's
 * 160 / 224-channel maps in 8 groups), G % 4 == 0, Cc % 4 == 0. */
int stx_cost_volume_fwd(const float* Lg, const float* Rg, int Cg, int G, const float* Lc, const float* Rc, int Cc,
                        const float* scale, float* vol, int B, int H, int W, int D, int mask_left, void* stream);
/* Per-pixel, per-group L2 normalisation of an NCHW map, the pre-pass of FoundationStereo's normalised group-wise correlation
 * (models/FoundationStereo/submodule.py:388-397: F.normalize(fea.float(), dim=2) on [B,G,C/G,H,W], eps 1e-12, group SUM):
 * y[b][c][p] = out_scale * x[b][c][p] / max(||x[b][group of c][p]||_2, 1e-12).  The normalised volume (:399-413) is
 * stx_cost_volume_fwd of the two normalised maps with out_scale = C/G on the left one (group mean x C/G = group sum).
 * x, y, gy, gx: [B][C][HW]; C % G == 0 (reference assert :390). */
int stx_group_normalize_fwd(const float* x, float* y, int B, int C, int G, int HW, float out_scale, void* stream);
int stx_group_normalize_bwd(const float* x, const float* gy, float* gx, int B, int C, int G, int HW, float out_scale, void* stream);
/* autograd backward of the above (no `scale`): gvol [B][D][H][W][G+2Cc] -> gLg,gRg [B][Cg][H][W], gLc,gRc [B][Cc][H][W] */
int stx_cost_volume_bwd(const float* gvol, const float* Lg, const float* Rg, int Cg, int G, int Cc, float* gLg,
                        float* gRg, float* gLc, float* gRc, int B, int H, int W, int D, int mask_left, void* stream);

/* ---- regression head and disparity estimators ---------------------------------------------------
 * F.upsample(trilinear) -> squeeze -> F.softmax(dim=1) -> disparity_regression, fused:
 *   models/GwcNet/gwcnet.py:197-224, models/PSMNet/stackhourglass.py:139-153, models/ACVNet/acv.py:206-251.
 * cost [B][Dc][Hc][Wc] -> disp [B][H][W]; stats [B][H][W][2] (per-pixel max and sum-exp, kept for backward; may be NULL). */
int stx_head_fwd(const float* cost, float* disp, float* stats, int B, int Dc, int Hc, int Wc, int D, int H, int W,
                 void* stream);
/* backward (deterministic, two passes); workspace: stx_head_bwd_workspace_floats(B, Dc, H, W) floats */
long long stx_head_bwd_workspace_floats(int B, int Dc, int H, int W);
int stx_head_bwd(const float* gdisp, const float* cost, const float* disp, const float* stats, float* gcost,
                 float* workspace, int B, int Dc, int Hc, int Wc, int D, int H, int W, void* stream);
/* the same two with an explicit interpolation rule: align_corners != 0 is F.upsample(..., align_corners=True), the
 * head of the PCWNet / CFNet family (models/PCWNet/pcwnet.py:446-470); align_corners == 0 equals the entries above */
int stx_head_fwd2(const float* cost, float* disp, float* stats, int B, int Dc, int Hc, int Wc, int D, int H, int W,
                  int align_corners, void* stream);
int stx_head_bwd2(const float* gdisp, const float* cost, const float* disp, const float* stats, float* gcost,
                  float* workspace, int B, int Dc, int Hc, int Wc, int D, int H, int W, int align_corners, void* stream);
/* disparity_regression (GwcNet/submodule.py:23-27), disparityregression (PSMNet/submodule.py:46-54),
 * softargmax_disparity_estimator (disparity_estimators/__init__.py:7-10): out[b][hw] = sum_d d * x[b][d][hw] */
int stx_softargmax_fwd(const float* x, float* out, int B, int D, int HW, void* stream);
/* argmax_disparity_estimator (disparity_estimators/__init__.py:13-15): out int64 [B][HW], first maximum */
int stx_argmax_fwd(const float* x, long long* out, int B, int D, int HW, void* stream);
/* unimodal_disparity_estimator (disparity_estimators/unimodal_disparity_estimator.py:4-25) and
 * dominant_modal_disparity_estimator (disparity_estimators/dominant_modal_disparity_estimator.py:35-54):
 * x [B][D][HW] probability volume (D = maxdisp) -> out [B][HW], the re-normalised expectation of d over the mode that
 * contains the arg-max / over the heavier of the two main modes of the 5-tap-blurred volume. 0/0 -> NaN as in the
 * reference. */
int stx_unimodal_fwd(const float* x, float* out, int B, int D, int HW, void* stream);
int stx_dominant_modal_fwd(const float* x, float* out, int B, int D, int HW, void* stream);
/* The same two estimators with their backward pass (kind 0: unimodal, 1: dominant-modal).  The reference functions are
 * differentiable w.r.t. x inside the mode mask, which is a constant of the graph (`x * mask.data`,
 * unimodal_disparity_estimator.py:20; boolean masks, dominant_modal_disparity_estimator.py:45-49; twin logic in
 * loss_functions/split_mode.py:9-35): d out / d x_k = m_k (k - out) / S.  stx_modal_fwd optionally writes
 * aux [B][5][HW] = (lo, hi, xlo, xhi, S): support [lo, hi] minus [xlo, xhi] and its probability mass S;
 * stx_modal_bwd turns g [B][HW], out and aux into gx [B][D][HW] (zero outside the support). */
int stx_modal_fwd(const float* x, float* out, float* aux, int B, int D, int HW, int kind, void* stream);
int stx_modal_bwd(const float* g, const float* out, const float* aux, float* gx, int B, int D, int HW, void* stream);
/* split_mode(x, maxdisp) -> (mode, mask)  (loss_functions/split_mode.py:9-35): the modal estimators' support mask of
 * the RAW volume x [B][D][HW] (arg-max, edges of its mode, symmetrised when the arg-max sits >= 3 bins off centre):
 * mask [B][D][HW] bytes (0 / 1 = the storage of a torch.bool tensor), mode = x * mask (fp32, same shape). */
int stx_split_mode(const float* x, float* mode, unsigned char* mask, int B, int D, int HW, void* stream);
/* F.softmax over the disparity axis of [B][D][HW] (ACVNet attention weights, acv.py:196) */
int stx_softmax_d_fwd(const float* x, float* y, int B, int D, int HW, void* stream);

/* ---- Conv3d / ConvTranspose3d aggregation (fp32 MFMA implicit GEMM) ---------------------------------
 * nn.Conv3d(k=3,p=1,s=1|2, bias=False), nn.Conv3d(k=1), nn.ConvTranspose3d(k=3,s=2,p=1,op=1, bias=False) of
 * convbn_3d / hourglass / dres / classif: models/GwcNet/gwcnet.py:68-153, models/GwcNet/submodule.py:17-20,
 * models/PSMNet/stackhourglass.py:10-84, models/ACVNet/acv.py:56-144.
 * Weights are first re-laid-out on the device into MFMA operand order (w: torch layout [A][B][T], T = k^3):
 *   mode 0: conv forward (w = [Cout][Cin][T]) and ConvTranspose dgrad;  1: stride-1 conv dgrad;
 *   mode 2: ConvTranspose forward (w = [Cin][Cout][T]) and stride-2 conv dgrad. */
long long stx_conv3d_packed_floats(int K, int N, int T);
int stx_conv3d_pack_weight(const float* w, float* wp, int A, int B, int T, int mode, void* stream);
/* out = act(conv(x) * scale[c] + bias[c] + residual); `relu` is the activation code: 0 none, 1 ReLU, 2 Mish, 3 LeakyReLU(0.01)
 * (x * tanh(softplus(x)), models/PCWNet/submodule.py:11-18); scale/bias/residual may be NULL; if `stats` != NULL the
 * per-workgroup (sum, sum of squares) of the RAW conv output are written to stats[rows][2][Cout] for train-mode
 * BatchNorm: rows = stx_conv3d_fwd_stat_rows(same shape arguments) -- every one of these rows is written, nothing beyond
 * them is touched (no zero-fill pass: hand exactly `rows` rows to stx_bn_finalize).  Cin % 8 == 0, Cout <= 128.
 * stx_conv3d_fwd_blocks(Do,Ho,Wo) * B is an upper bound of `rows` for any channel configuration. */
int stx_conv3d_fwd_blocks(int Do, int Ho, int Wo);
long long stx_conv3d_fwd_stat_rows(int B, int Di, int Hi, int Wi, int Cin, int Cout, int ks, int stride);
int stx_conv3d_fwd(const float* x, const float* wp, float* out, const float* scale, const float* bias,
                   const float* residual, float* stats, int B, int Di, int Hi, int Wi, int Cin, int Cout, int ks,
                   int stride, int relu, void* stream);
int stx_deconv3d_fwd_blocks(int Di, int Hi, int Wi);
int stx_deconv3d_fwd(const float* x, const float* wp, float* out, const float* scale, const float* bias,
                     const float* residual, float* stats, int B, int Di, int Hi, int Wi, int Cin, int Cout, int Do,
                     int Ho, int Wo, int relu, void* stream);
/* weight gradient dW[cc][cf][T] = sum_o fine[S*o+tap-pad][cf] * coarse[o][cc]
 *   conv: fine = layer input, coarse = grad of output;  ConvTranspose: fine = grad of output, coarse = layer input.
 * workspace: stx_conv3d_wgrad_workspace_floats(...) floats. CF % 32 == 0, CC % 32 == 0. */
long long stx_conv3d_wgrad_workspace_floats(int B, int Dc, int Hc, int Wc, int CF, int CC, int ks, int stride);
int stx_conv3d_wgrad(const float* fine, const float* coarse, float* dw, float* workspace, int B, int Df, int Hf, int Wf,
                     int CF, int Dc, int Hc, int Wc, int CC, int ks, int stride, void* stream);

/* The same weight gradient for a 3x3x3 stride-1 `convbn_3d` (+ ReLU) block in training (reference models/GwcNet/submodule.py:17-20
 * under autograd), taking the gradient gy BEHIND the BatchNorm / activation: the gradient of the raw convolution output
 *   dz = gamma invstd (gy' - sum_g / n - xhat sum_gx / n),  gy' = gy masked by the activation (sign of fmaf(z, scale, shift)),
 *   xhat = (z - mean) invstd,  sums = [2][CC] = rows 0 and 1 of stx_bn_bwd_reduce2's sums,  inv_n = 1 / (B D H W),
 * is formed inside the kernel and also written to `dz` (for the data-gradient launch): the stx_bn_bwd_apply2 pass of the block is
 * not needed.  act: 0 none, 1 ReLU.  gamma may be NULL (ones).  Only shapes stx_conv3d_wgrad_bn_supported accepts; workspace as
 * stx_conv3d_wgrad_workspace_floats(B, D, H, W, CF, CC, 3, 1). */
int stx_conv3d_wgrad_bn_supported(int B, int D, int H, int W, int CF, int CC);
int stx_conv3d_wgrad_bn(const float* x, const float* gy, const float* z, const float* scale, const float* shift, const float* mean,
                        const float* invstd, const float* gamma, const float* sums, float inv_n, int act, float* dz, float* dw,
                        float* workspace, int B, int D, int H, int W, int CF, int CC, void* stream);

/* 3x3 stride-1 Conv2d (padding 1, no bias), channels-last, of the 2-D feature CNN's BasicBlocks with 32 / 64 channels: forward
 * and data gradient (reference models/GwcNet/gwcnet.py:12-42 `convbn(in, out, 3, 1, pad, 1)` -> nn.Conv2d at 1/2 and 1/4
 * resolution; PSMNet/submodule.py:57-97; ACVNet/acv.py:15-40).  x [B][H][W][Cin], out [B][H][W][Cout], fp32.
 * w = the layer's parameter [Co_w][Ci_w][3][3] in channels_last storage = dense [Co_w][3][3][Ci_w], read by the kernel as it is
 * (no packing step).  dgrad = 0: out = conv(x, w), Cin = Ci_w, Cout = Co_w.  dgrad = 1: the input gradient of that layer,
 * x = the output gradient, Cin = Co_w, Cout = Ci_w (flipped taps, transposed channels).
 * stats (may be NULL): [stx_conv2d_stat_rows(groups)][2][Cout] per-workgroup sum / sum of squares of the raw output, the batch
 *   split into `groups` equal parts (views) with their own rows: rows [g * rows / groups, (g + 1) * rows / groups) belong to part g
 *   -> stx_bn_finalize_groups.  Shapes: stx_conv2d_supported (Cin 32 or 64, Cout % 16 == 0). */
int stx_conv2d_supported(int Cin, int Cout);
long long stx_conv2d_stat_rows(int groups);
int stx_conv2d_fwd(const float* x, const float* w, float* out, float* stats, int B, int H, int W, int Cin, int Cout, int dgrad,
                   int groups, void* stream);

/* Classifier tail Conv3d(Cin, 1, k=3, p=1, bias=False) (GwcNet/gwcnet.py:139-153, PSMNet/stackhourglass.py:74-84):
 * N = 1 is not GEMM-shaped, so it gets VALU kernels. w: torch layout [1][Cin][27]; out/residual/gy: [B][D][H][W].
 * Cin % 16 == 0 (wgrad: Cin <= 64). dgrad: gx [B][D][H][W][Cin] = autograd input gradient of the same layer
 * (Cin % 4 == 0, Cin <= 64); a streaming kernel: reads the 1-channel gy, writes Cin channels. */
int stx_conv3d_c1_fwd(const float* x, const float* w, const float* residual, float* out, int B, int D, int H, int W,
                      int Cin, void* stream);
long long stx_conv3d_c1_wgrad_workspace_floats(int Cin);
int stx_conv3d_c1_wgrad(const float* x, const float* gy, float* dw, float* workspace, int B, int D, int H, int W,
                        int Cin, void* stream);
int stx_conv3d_c1_dgrad(const float* gy, const float* w, float* gx, int B, int D, int H, int W, int Cin, void* stream);

/* Mish activation y = x * tanh(softplus(x)) (models/PCWNet/submodule.py:11-18,178-190): n floats, n % 4 == 0, in place allowed;
 * backward gx = gy * mish'(x) with x the activation input. */
int stx_mish_fwd(const float* x, float* y, long long n, void* stream);
int stx_mish_bwd(const float* gy, const float* x, float* gx, long long n, void* stream);
/* IGEV-family cost aggregation (models/IGEVStereo/igev_stereo.py:23-100).
 * stx_depth_to_space: the interleave of ConvTranspose3d(k=4, s=2, p=1)'s eight output-parity classes (igev_stereo.py:44-51):
 * the transposed convolution runs as ONE 3x3x3 stride-1 convolution with 8*C class-major output channels on stx_conv3d_fwd
 * (each class: its 8 taps in a zero-filled 27-tap set); y [B][D][H][W][8C] -> out [B][2D][2H][2W][C],
 * out[2d+pd][2h+ph][2w+pw][c] = y[d][h][w][(4pd+2ph+pw)C + c]; inverse != 0 maps the other way (its backward).  C % 4 == 0. */
int stx_depth_to_space(const float* y, float* out, int B, int D, int H, int W, int C, int inverse, void* stream);
/* FeatureAtt (models/IGEVStereo/submodule.py:228-241): out = cv * sigmoid(att), the gate att [B][HW][C] broadcast over the
 * disparity axis of cv / out [B][D][HW][C]; backward: gcv = g * sigmoid(att), gatt = sum_d g * cv * s (1 - s) (either may be
 * NULL).  C % 4 == 0. */
int stx_gate_fwd(const float* cv, const float* att, float* out, int B, int D, long long HW, int C, void* stream);
int stx_gate_bwd(const float* g, const float* cv, const float* att, float* gcv, float* gatt, int B, int D, long long HW,
                 int C, void* stream);

/* ---- channel concatenation of channels-last activations --------------------------------------------------
 * Replaces `torch.cat((l2, l3, l4), dim=1)` of the feature extractors (reference models/GwcNet/gwcnet.py:59,
 * models/ACVNet/acv.py:48) for dense [nvox][C_k] (NHWC / NDHWC) tensors: out[v] = in0[v] | in1[v] | in2[v] | in3[v]; unused
 * parts: NULL with C = 0; every C_k a multiple of 4.  stx_split_channels is the inverse (the concatenation's backward); a
 * part with C_k > 0 must have a destination. */
int stx_concat_channels(const float* in0, const float* in1, const float* in2, const float* in3, int C0, int C1, int C2, int C3,
                        float* out, long long nvox, void* stream);
int stx_split_channels(const float* in, float* out0, float* out1, float* out2, float* out3, int C0, int C1, int C2, int C3,
                       long long nvox, void* stream);
/* Batched 2-D transpose out[n][c][r] = in[n][r][c]: the channels-last <-> channel-major re-layout of the feature maps in front
 * of the cost-volume builders (their kernels take NCHW rows, the 2-D CNN produces NHWC; `Tensor.contiguous()` in the reference's
 * terms).  rows and cols multiples of 4. */
int stx_transpose(const float* in, float* out, int N, int rows, int cols, void* stream);

/* ---- ACVNet extras (models/ACVNet/acv.py) --------------------------------------------------------------
 * Depth-wise nn.Conv3d(C, C, (1,3,3), groups=C, dilation=d, padding=(0,d,d)) (acv.py:109-112,183-187) on a channels-last
 * volume; `dil` = int[C/4] dilation per channel quad (device pointer); w = [C][9]; flip=1 mirrors the taps (input gradient). */
int stx_dwconv_hw_fwd(const float* x, const float* w, const int* dil, float* out, int B, int D, int H, int W, int C,
                      int flip, void* stream);
long long stx_dwconv_hw_wgrad_workspace_floats(int C);
int stx_dwconv_hw_wgrad(const float* x, const float* gy, const int* dil, float* dw, float* workspace, int B, int D,
                        int H, int W, int C, void* stream);
/* gradient of `softmax(att, dim=2) * concat_volume` (acv.py:196) w.r.t. the probabilities:
 * gscale[b][d][h][w] = sum_c gvol[b][d][h][w][c] * concat[b][d][h][w][c]; gvol has 2*Cc channels */
int stx_cost_volume_scale_bwd(const float* gvol, const float* Lc, const float* Rc, float* gscale, int B, int Cc, int H,
                              int W, int D, int mask_left, void* stream);
/* out[v][c] = x[v][c] * s[v] over [nvox][C] */
int stx_scale_channels(const float* x, const float* s, float* out, long long nvox, int C, void* stream);
/* Backward of the attention concat volume vol = prob * concat(L, R shifted) (models/ACVNet/acv.py:196 with
 * ACVNet/submodule.py:180-191) in one pass over gvol [B][D][H][W][2 Cc]: gL, gR [B][Cc][H][W], gprob [B][D][H][W].
 * D * W * 4 bytes must fit in 150 KiB of LDS (else: bad-argument error; use stx_cost_volume_scale_bwd +
 * stx_scale_channels + stx_cost_volume_bwd). */
int stx_ac_volume_bwd(const float* gvol, const float* Lc, const float* Rc, const float* prob, float* gL, float* gR,
                      float* gprob, int B, int Cc, int H, int W, int D, int mask_left, void* stream);

/* ---- train-mode BatchNorm3d (+ReLU / residual) around the convolutions ---------------------------------
 * nn.BatchNorm3d of convbn_3d (models/GwcNet/submodule.py:17-20) in train() mode and the adds/ReLUs that follow it
 * (GwcNet/gwcnet.py:96-103,185; PSMNet/stackhourglass.py:31-48). */
int stx_bn_reduce_blocks(void);
/* Batch statistics of a channels-last activation z [nvox][C] whose producer has no fused epilogue -- the nn.BatchNorm2d
 * layers behind the MIOpen convolutions of the 2-D feature CNN (models/GwcNet/gwcnet.py:12-65 `convbn`, BasicBlock):
 * partials [stx_bn_stats_rows(nvox, C)][2][C] = per-workgroup (sum z, sum z^2), the row format stx_bn_finalize takes.
 * C: multiple of 4 with C/4 dividing 256.
 * GROUPS (here and in stx_bn_apply / stx_bn_bwd_reduce2 / stx_bn_bwd_apply2): the tensors are `groups` consecutive slabs of
 * `nvox` voxels, each with its OWN statistics -- the per-group vectors (partials, scale, shift, mean, invstd, sums) are
 * [groups][...], gamma is shared.  The 2-D CNN sends the left and the right view through its convolutions as one batch
 * while every BatchNorm keeps the per-view statistics of the reference's two extractor calls (gwcnet.py:172-173). */
int stx_bn_stats_rows(long long nvox, int C);
int stx_bn_stats(const float* z, float* partials, long long nvox, int C, int groups, void* stream);
/* partials [nrows][2][C] (from the conv epilogue) -> scale = gamma*invstd, shift = beta - mean*scale, mean, invstd;
 * running_mean/var (may be NULL) updated with `momentum` and the unbiased variance, like torch. */
int stx_bn_finalize(const float* partials, int nrows, int C, double count, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, float momentum, float eps, float* scale, float* shift,
                    float* mean, float* invstd, void* stream);
/* The same for `groups` slabs of a batch with their own statistics in ONE launch: partials [groups][nrows][2][C],
 * count = voxels per slab, out [4][groups][C] = scale, shift, mean, invstd.  The running statistics are updated slab
 * after slab, in order -- bit-identical to `groups` calls of stx_bn_finalize (the two views of the 2-D feature CNN:
 * reference models/GwcNet/gwcnet.py:172-173 runs the extractor once per view). */
int stx_bn_finalize_groups(const float* partials, int nrows, int C, double count, const float* gamma, const float* beta,
                           float* running_mean, float* running_var, float momentum, float eps, float* out, int groups,
                           void* stream);
/* out = act(z1*scale1+shift1 [+ z2*scale2+shift2 | + z2 when scale2 == NULL]) over [nvox][C]; `relu` = activation code
 * (0 none, 1 ReLU, 2 Mish, 3 LeakyReLU(0.01)) here and in the backward passes below; Mish is differentiated at the pre-activation value,
 * which stx_bn_bwd_reduce2 / _apply2 recompute from z and the scale / shift vectors (y = NULL) */
int stx_bn_apply(const float* z1, const float* scale1, const float* shift1, const float* z2, const float* scale2,
                 const float* shift2, float* out, long long nvox, int C, int relu, int groups, void* stream);
/* sums[3][C] = sum g, sum g*xhat1, sum g*xhat2 with g = gy*[y>0]; partials: scratch of stx_bn_reduce_blocks()*3*C floats */
int stx_bn_bwd_reduce(const float* gy, const float* y, const float* z1, const float* mean1, const float* invstd1,
                      const float* z2, const float* mean2, const float* invstd2, float* partials, float* sums,
                      long long nvox, int C, int relu, void* stream);
/* dz_k = gamma_k*invstd_k*(g - sums[0]/N - xhat_k*sums[k]/N); gout (may be NULL) receives g for a plain residual branch */
int stx_bn_bwd_apply(const float* gy, const float* y, const float* z1, const float* mean1, const float* invstd1,
                     const float* gamma1, const float* z2, const float* mean2, const float* invstd2,
                     const float* gamma2, const float* sums, float* dz1, float* dz2, float* gout, long long nvox, int C,
                     int relu, void* stream);
/* The same two passes taking the ReLU mask from the forward pass's per-channel scale / shift instead of the activated
 * output (y may be NULL): sign(y) = sign(fmaf(z1, scale1, shift1) [+ fmaf(z2, scale2, shift2)]) is recomputed from
 * operands these passes read anyway -- one volume-sized read less per pass.  (A block with a plain residual still passes y.)
 * With groups > 1 `sums` holds groups + 1 slabs of [3][C]: the per-group sums, then their total over the groups (the gradients
 * of the gamma / beta the groups share); stx_bn_bwd_apply2 reads the first `groups` slabs. */
int stx_bn_bwd_reduce2(const float* gy, const float* y, const float* z1, const float* mean1, const float* invstd1,
                       const float* z2, const float* mean2, const float* invstd2, const float* scale1,
                       const float* shift1, const float* scale2, const float* shift2, float* partials, float* sums,
                       long long nvox, int C, int relu, int groups, void* stream);
int stx_bn_bwd_apply2(const float* gy, const float* y, const float* z1, const float* mean1, const float* invstd1,
                      const float* gamma1, const float* z2, const float* mean2, const float* invstd2,
                      const float* gamma2, const float* scale1, const float* shift1, const float* scale2,
                      const float* shift2, const float* sums, float* dz1, float* dz2, float* gout, long long nvox,
                      int C, int relu, int groups, void* stream);

/* ---- Evaluation-path input step on the device --------------------------------------------------------
 * pad_to_2x (datasets/data_augmentation/__init__.py:57-80: zero padding on top and to the right, to multiples of 96)
 * + get_transform (datasets/utils.py:62-69: torchvision ToTensor and Normalize) of one view, fused:
 *   img uint8 [B][H][W][3] -> out fp32 [B][3][Hp][Wp],  out = ((inside ? img : 0) / 255 - mean[c]) / std[c],
 * top = Hp - H rows of padding above the image, Wp - W columns to its right; mean3 / std3 are HOST arrays of 3 floats.
 * Bit-identical to the reference arithmetic (IEEE fp32 division by 255, subtraction, division by std). */
int stx_pad_normalize_u8(const unsigned char* img, float* out, int B, int H, int W, int Hp, int Wp, int top,
                         const float* mean3, const float* std3, void* stream);

/* ---- Sampled (cascade) cost volume of the CFNet family ---------------------------------------------------
 * One call replaces, per cascade stage (models/CFNet/cfnet.py:553-566 / 584-597),
 *   SpatialTransformer (models/CFNet/submodule.py:306-350) x 2, groupwise_correlation_4D (submodule.py:163-169),
 *   cost_volume_generator (cfnet.py:470-497) x 2 and torch.cat((gwc_volume, concat_volume, disparity_samples), dim=1):
 *   vol[b][s][h][w][:] = ( mean_c Lg[g*cpg+c][h][w] * Rg[g*cpg+c][h][x]   g < G
 *                        | Lc[c][h][w] | Rc[c][h][x]                        c < Cc
 *                        | samples[b][s][h][w] | zero pad up to CTp ),     x = w - samples[b][s][h][w]
 * with the reference's boundary rule: the gather index is clamped into the row and every right-feature term is
 * zeroed where the un-clamped x leaves [0, W-1].  Features NCHW fp32, samples [B][S][H][W] (integer-valued floats),
 * vol NDHWC [B][S][H][W][CTp], CTp >= G + 2*Cc + 1 and a multiple of 4 (the consuming convolution wants 8).
 * _bwd: gradients w.r.t. the four feature maps (the hypotheses carry none); gRg / gRc are zeroed and then accumulated
 * with float atomics, like the reference's gather backward. */
int stx_sampled_volume_fwd(const float* Lg, const float* Rg, int Cg, int G, const float* Lc, const float* Rc, int Cc,
                           const float* samples, float* vol, int B, int H, int W, int S, int CTp, void* stream);
int stx_sampled_volume_bwd(const float* gvol, const float* Lg, const float* Rg, int Cg, int G, int Cc,
                           const float* samples, float* gLg, float* gRg, float* gLc, float* gRc, int B, int H, int W,
                           int S, int CTp, void* stream);

/* ---- refine2d.hip: 2-D helpers of the PCWNet / CFNet family -------------------------------------
 * warp(x, disp) (models/PCWNet/submodule.py:137-176): x [B][C][H][W] (right-view features), disp [B][1][H][W] ->
 * out [B][C][H][W] = grid_sample(x, grid(w - disp, h)) (bilinear, zeros outside, the reference's (W-1)/(H-1) grid under
 * align_corners=False) * (in-image weight sum >= 0.999).  Backward: gx (cleared inside, float atomics like torch's own
 * grid-sampler backward; may be NULL) and gdisp [B][1][H][W] (may be NULL); the validity mask is a constant. */
int stx_warp_fwd(const float* x, const float* disp, float* out, int B, int C, int H, int W, void* stream);
int stx_warp_bwd(const float* gout, const float* x, const float* disp, float* gx, float* gdisp, int B, int C, int H, int W,
                 void* stream);
/* build_corrleation_volume(ref, tgt, maxdisp, groups) (models/PCWNet/submodule.py:121-135, dup CFNet/submodule.py:181-195):
 * ref, tgt [B][C][H][W] -> vol [B][groups][2*maxdisp+1][H][W]; slice maxdisp+i, i >= 0: mean over the group's channels of
 * ref[w] * tgt[w-i] at w >= i, 0 elsewhere; slice maxdisp-n, n >= 1 (literal semantics of the reference's `[..., :-i]` with
 * negative i): the FIRST n columns of ref against the LAST n columns of tgt, 0 elsewhere.  Every element of vol is written.
 * maxdisp <= min(48, W).  Backward: gref / gtgt [B][C][H][W] fully written (either may be NULL), deterministic. */
int stx_corr_volume_fwd(const float* ref, const float* tgt, float* vol, int B, int C, int H, int W, int maxdisp, int groups,
                        void* stream);
int stx_corr_volume_bwd(const float* gvol, const float* ref, const float* tgt, float* gref, float* gtgt, int B, int C, int H,
                        int W, int maxdisp, int groups, void* stream);
/* disparity_variance(x, maxdisp, disparity) (models/CFNet/submodule.py:128-134; samples == NULL): out[b][i] =
 * sum_d x[b][d][i] * (d - disp[b][i])^2, and disparity_variance_confidence(x, samples, disparity) (:136-140; samples
 * [B][D][HW]): sum_d x * (disp - samples)^2.  x [B][D][HW], disp / out [B][HW].  Backward: gx, gdisp, gsamples (each may
 * be NULL). */
int stx_disparity_variance_fwd(const float* x, const float* disp, const float* samples, float* out, int B, int D, int HW,
                               void* stream);
int stx_disparity_variance_bwd(const float* g, const float* x, const float* disp, const float* samples, float* gx, float* gdisp,
                               float* gsamples, int B, int D, int HW, void* stream);

#ifdef __cplusplus
}
#endif
#endif
