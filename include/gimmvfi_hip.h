/* gimmvfi_hip.h -- C ABI of libgimmvfi_hip.so (MI355X / gfx950 HIP kernels for the
 * GIMM-VFI-R / GIMM-VFI-F inference path: RAFT or FlowFormer flow estimator, GIMM motion INR,
 * softmax splatting, frame synthesis).
 *
 * Plain pointers and sizes only: every pointer is a DEVICE pointer, every entry point
 * enqueues on the hipStream_t passed last (cast to void*; 0 = default stream) and returns
 * the hipError_t of the launch (0 = success).  No torch types cross this boundary.
 *
 * Data layout ("activation tensor"): NHWC, channel-contiguous, with an explicit pixel
 * pitch `ld` (elements between consecutive pixels) so that a kernel can read or write a
 * channel slice of a wider (concatenation) buffer.  Element type `dtype` is
 * GVFI_F32 (float) or GVFI_BF16 (raw bfloat16 bits); geometry (flows, coordinates,
 * correlation volumes, splat accumulators, images for warping) is always float.
 *
 * Each entry point cites the reference code (GSeanCDAT/GIMM-VFI, paths relative to
 * src/models/generalizable_INR/) it replaces.
 */
#ifndef GIMMVFI_HIP_H
#define GIMMVFI_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

/* element type of the activation / weight tensors of a call ("dtype").  GVFI_F16 (IEEE half, MFMA
 * v_mfma_f32_32x32x16_f16 at the bf16 rate, float accumulation, conversions saturate at +-65504) exists for the
 * recurrence of the FlowFormer flow estimator: 11 significand bits instead of 8 -- bf16 operand rounding inside the
 * un-trained 32-iteration update block costs GIMM-VFI-F its reference fidelity at 2K / 4K, half precision does not
 * (DESIGN.md section 9).  Supported by the LDS-DMA convolution (4-wave tiles + weights-direct variant), the generic
 * convolution and every element-wise / gather kernel; the halo-staged 3x3, patch, fused-INR and MFMA-attention kernels
 * are bf16 / float only and report "not eligible" for it. */
enum { GVFI_F32 = 0, GVFI_BF16 = 1, GVFI_F16 = 2 };
enum {
    GVFI_ACT_NONE = 0,
    GVFI_ACT_RELU = 1,
    GVFI_ACT_LRELU = 2,   /* LeakyReLU(0.1) */
    GVFI_ACT_PRELU = 3,   /* per-channel slope */
    GVFI_ACT_SIGMOID = 4,
    GVFI_ACT_TANH = 5,
    GVFI_ACT_SIN = 6,
    GVFI_ACT_GELU = 7     /* exact (erf) GELU: nn.GELU() of the FlowFormer MLPs / token encoders */
};
enum { GVFI_PAD_ZEROS = 0, GVFI_PAD_REFLECT = 1 };
enum {
    GVFI_EPI_STD = 0,    /* y = act2(act1(acc + bias) + res) * out_scale                       */
    GVFI_EPI_GRU_ZR = 1, /* cout <  Cout/2: y  = sigmoid(acc+bias+res)        (z gate)          *
                          * cout >= Cout/2: y2 = sigmoid(acc+bias+res) * aux0 (r * h)           */
    GVFI_EPI_GRU_Q = 2   /* y = (1 - aux1) * aux0 + aux1 * tanh(acc+bias+res) (h update)        *
                          * res (optional, [pixel][Cout]): pre-activation term evaluated outside *
                          * the recurrence -- the gate convolution's share of RAFT's constant    *
                          * context features (raft/update.py:58-73, raft/raft.py:134-136)        */
};

/* Implicit-GEMM convolution on the matrix cores (MFMA), NHWC, fused epilogue.
 * Replaces every nn.Conv2d (+ bias + activation + residual + GRU gating) on the path:
 * raft/extractor.py:6-58,122-220; raft/update.py:6-148; modules/fi_components.py:17-54,
 * 97-222,229-340; gimmvfi_r.py:51-64,84-109; and, with per-group weights, the all-pairs
 * correlation GEMM raft/corr.py:167-175 and the INR linear layers modules/hyponet.py:95-143. */
typedef struct {
    int dtype;              /* element type of x0, x1, w (and of y/res/aux unless flagged f32) */
    /* input = channel concatenation of up to two NHWC sources (c1 may be 0).             */
    const void* x0; int ld0; int c0;   /* c0, c1: multiples of the 16-byte vector (4 f32 / 8 bf16) */
    const void* x1; int ld1; int c1;
    int N, H, W;            /* input images, height, width                                 */
    /* weights packed [groups][Cout][KH][KW][c0+c1] in `dtype`; w_group_stride in elements */
    const void* w; long long w_group_stride; int groups;   /* images per group = N/groups  */
    const float* bias;      /* [Cout] float or NULL (shared by all groups)                 */
    int Cout, KH, KW, stride, pad_h, pad_w, pad_mode;
    int Ho, Wo;
    int epi_mode;
    int act1; const float* slope1;
    const void* res; int ldr; int res_f32;
    int act2; const float* slope2;
    float out_scale;
    void* y; int ldy; int y_f32;
    void* y2; int ldy2;                 /* GRU_ZR: r*h destination                          */
    const void* aux0; int lda0;         /* GRU: h                                           */
    const void* aux1; int lda1;         /* GRU_Q: z                                         */
    float* stats;                       /* optional [N][Cout][2] x 8 bytes: per (image, channel) sum and sum of squares of the
                                         * stored outputs as 64-bit fixed point (Q24 / Q20: integer atomics, the result does not
                                         * depend on the workgroups' order -- round 6; zeroed by the caller), accumulated by the
                                         * LDS-DMA kernel's bf16 store loop (InstanceNorm statistics of raft/extractor.py:26-58
                                         * fused into the producing convolution); needs Ho*Wo % BM == 0, see gvfi_conv2d_stats_ok */
    int tile_hint;                      /* 0 = auto, else BN | BM << 10 | NS << 20: Cout tile width 32/64/128/256, (LDS-DMA
                                         * kernel) pixel tile height 64/128 (BN = 128) or 128/256 (BN = 32), and ring depth
                                         * NS = 2..4 of its 128-byte K chunks (4-wave tiles) */
    int w_layout;                       /* 0: [Cout][KH][KW][Cin];  1 (LDS-DMA kernel only): K-chunk major,  *
                                         * [K/64][Cout][64] with the 16-byte groups of a row XOR-swizzled by *
                                         * (cout>>1)&7 -- the exact LDS image, so the weight tile DMA is one  *
                                         * contiguous copy (K chunk = 128 bytes).  Chunk index =               *
                                         * channel_chunk * KH*KW + tap (the kernel walks the taps innermost);  *
                                         * 2 (LDS-DMA kernel, bf16, "weights direct"): MFMA-fragment order      *
                                         * [K/64][ceil(Cout/32)][4 k-steps][64 lanes][8]: lane l of a fragment  *
                                         * holds W[32*nb + (l&31)][64*chunk + 16*kk + 8*(l>>5) .. +8] (zeros    *
                                         * beyond Cout) -- one coalesced 1 KiB load per MFMA operand, no LDS;   *
                                         * same chunk order as layout 1                                          */
    int algo;                           /* bits 0..3: 0 = auto, 1 = generic register-staged kernel,    *
                                         * 2 = LDS-DMA kernel (needs c0,c1 % 64 bf16 / 32 f32),   *
                                         * 3 = patch kernel (few channels, see gvfi_conv2d_patch); *
                                         * 4 = halo-staged 3x3 kernel (gvfi_conv2d_p3x3);           *
                                         * 5 = its mid-channel sibling (gvfi_conv2d_p3x3s);         *
                                         * 7 = column kernel of the 7x7 few-channel layers          *
                                         * (gvfi_conv2d_col7);                                       *
                                         * bit 4 "pad16": the caller owns the channel padding of   *
                                         * y and res up to the next 16-byte boundary -- a ragged   *
                                         * last channel group may be accessed in whole 16-byte     *
                                         * units, y's pad channels receive zeros (patch kernel);   *
                                         * bit 6 (column kernel, Cout 3, float y): the result is   *
                                         * finalised to clamp((y + 1) / 2, 0, 1) and stored PLANAR *
                                         * (N, 3, Ho, Wo) float at y2; y is not written           *
                                         * (gvfi_finalize_image folded into the last layer);       *
                                         * bit 5 / 7: A/B switches (8-wave tile: DMA issue spread *
                                         * over the MFMA groups; 64-byte K chunks), bits 8..:      *
                                         * profiling switches (skip phases, s_memtime stamps)     */
    int state_f32;                      /* GRU epilogues, bf16 mode: the recurrent state is FLOAT -- the hidden state is *
                                         * an accumulator over 20-32 iterations; rounding it to bf16 every iteration is   *
                                         * the one place where bf16 error accumulates instead of only rounding MFMA       *
                                         * operands.  GRU_ZR: aux0 (h) and y (z) are float, y2 (r*h, an operand) stays    *
                                         * bf16.  GRU_Q: aux0 (h), aux1 (z) float; y = new h in bf16 (operand copy for the *
                                         * next convolutions), y2 = new h in float (the state).                           */
} gvfi_conv_params;

int gvfi_conv2d(const gvfi_conv_params* p, void* stream);
/* which kernel gvfi_conv2d would launch: plan[5] = {algo (1 generic, 2 LDS-DMA), BM, BN, K-chunk bytes,
 * LDS stages} */
int gvfi_conv2d_plan(const gvfi_conv_params* p, int* plan);
int gvfi_conv2d_glds_plan(const gvfi_conv_params* p, int* plan);
/* 1 when gvfi_conv2d would accumulate p->stats for this problem (else the caller runs gvfi_instnorm_stats) */
int gvfi_conv2d_stats_ok(const gvfi_conv_params* p);
/* the two kernels behind gvfi_conv2d (exposed for A/B measurements) */
int gvfi_conv2d_glds_eligible(const gvfi_conv_params* p);
int gvfi_conv2d_glds(const gvfi_conv_params* p, void* stream);
/* Two INDEPENDENT convolutions as ONE launch (csrc/conv_igemm_glds.hip, conv_igemm_glds_pair_kernel): the two branches of the
 * flow estimators' motion encoder -- convc1 || convf1 and convc2 || convf2 of BasicMotionEncoder (raft/update.py:94-112;
 * FlowFormer gru.py:96-116) read different inputs and meet only in `conv`.  In the 20 / 32-iteration recurrences every launch is
 * a latency chain of its own (prologue + K loop + epilogue of one workgroup), so the small flow-branch layers cost a launch
 * each on the critical path; here their workgroups are the tail of the correlation-branch layer's grid.  Both problems must
 * be ones the weights-direct 64 x 128 variant takes (w_layout 2, same 16-bit dtype, groups <= 1, no stats): 0 = launched,
 * -2 = not such a pair (launch them one by one: the pair kernel runs the SAME body -- results are bit-identical). */
int gvfi_conv2d_pair(const gvfi_conv_params* a, const gvfi_conv_params* b, void* stream);
/* "Patch" convolution (csrc/conv_patch.hip) for few-channel layers at full resolution that the LDS-DMA kernel cannot
 * take: one source of <= 64 bf16 / 32 f32 channels (multiple of a 16-byte group), <= 64 output channels, filters up to
 * 7x7, stride 1 or 2, zero or reflect padding, standard epilogue -- the combination block of multi_flow_combine
 * (gimmvfi_r.py:60-64,305-308), the encoder stems (raft/extractor.py:137), the reflect-padded motion-encoder layers
 * (gimmvfi_r.py:84-109).  The 8 x 64-pixel output block's input halo and all weights are staged once in LDS.
 * gvfi_conv2d routes here by itself (gvfi_conv2d_patch_eligible == 1 and the LDS-DMA kernel not eligible) or with
 * algo = 3. */
int gvfi_conv2d_patch_eligible(const gvfi_conv_params* p);
int gvfi_conv2d_patch(const gvfi_conv_params* p, void* stream);
/* 7x7 stride-1 zero-padded (pad 3) bf16 convolution over <= 32 input and <= 32 output channels with the "pad16" contract
 * (algo bit 4), activation none / ReLU / LeakyReLU / PReLU, optional FLOAT residual into a float y: the combination block
 * of multi_flow_combine, 9 -> 18 -> 3 (gimmvfi_r.py:60-64,305-308) (csrc/conv_col7.hip).  Lanes run down a column of
 * pixels, so one LDS fragment of an input column feeds the MFMAs of all seven horizontal taps; v_mfma_f32_16x16x32_bf16
 * with the weights as the 16-row operand (Cout padded to 16, not 32).  gvfi_conv2d routes here when
 * gvfi_conv2d_col7_eligible == 1 (algo 0; >= 65536 output pixels, image >= 32 x 32) or with algo = 7 (eligible == 2:
 * runnable); algo = 3 keeps the patch kernel.  fp32 accumulation order differs from the patch kernel (not bit-identical). */
int gvfi_conv2d_col7_eligible(const gvfi_conv_params* p);
int gvfi_conv2d_col7(const gvfi_conv_params* p, void* stream);
/* 3x3 stride-1 zero-padded bf16 convolution with Cout % 256 == 0, channel counts % 64 == 0, w_layout 1 and >= 65536
 * output pixels -- the ResBlocks of NewMultiFlowDecoder (fi_components.py:97-154,279-340), the path's FLOP-dominant
 * layers (csrc/conv_p3x3.hip): 16 x 16-pixel output tile whose 18 x 18 input halo is staged once per 64-channel chunk
 * instead of once per tap.  Results are bit-identical to the LDS-DMA kernel's (same K order, same epilogue arithmetic).
 * gvfi_conv2d routes here by itself when gvfi_conv2d_p3x3_eligible == 1 (algo 0) or with algo = 4 (eligible == 2:
 * runnable, but fewer than 65536 output pixels); algo = 2 keeps the LDS-DMA kernel.  Two launch forms, same results: with
 * Cout == 256, out_scale == 1 and at least two tiles per compute unit ONE PERSISTENT workgroup per CU walks its tiles as one
 * stream of channel chunks (the LDS-DMA ring never drains between tiles, wave-private epilogue, stores retire under the next
 * tile's K loop: round 6, +4-6 % on the hot layer); else one workgroup per tile.  algo bits 13, 14 are A/B switches: 1 = tile
 * per workgroup with the workgroup-wide epilogue (the round-2 kernel), 2 = tile per workgroup, 3 = stream where it applies. */
int gvfi_conv2d_p3x3_eligible(const gvfi_conv_params* p);
int gvfi_conv2d_p3x3(const gvfi_conv_params* p, void* stream);
/* the launch form gvfi_conv2d_p3x3 takes for this problem on the current device: 1 / 2 = one workgroup per tile (workgroup-wide /
 * wave-private epilogue), 3 = persistent stream kernel; 0 = not this kernel's problem */
int gvfi_conv2d_p3x3_form(const gvfi_conv_params* p);
/* The same scheme for the mid-channel full-resolution layers (csrc/conv_p3x3s.hip): 3x3 stride-1 zero-padded bf16, ONE
 * source of exactly 32 or 64 channels, Cout <= 64 (multiple of 8), plain weight image (w_layout 0), >= 65536 output
 * pixels -- conv2 / conv4 of the decoder ResBlocks (fi_components.py:107-133), the CNN encoder's 32 -> 32 layers, the
 * 32 <-> 64 transitions.  One patch buffer + two small weight stages: 2-5 workgroups per CU.  Bit-identical to the LDS-DMA
 * kernel.  gvfi_conv2d routes here when gvfi_conv2d_p3x3s_eligible == 1 (algo 0) or with algo = 5. */
int gvfi_conv2d_p3x3s_eligible(const gvfi_conv_params* p);
int gvfi_conv2d_p3x3s(const gvfi_conv_params* p, void* stream);

/* ---- input preparation (gimmvfi_r.py:230-231,329-337,349; raft/raft.py:111-112) ------ */
/* bilinear resize of float planes, align_corners=False, rscale = (float)(1.0/scale_factor)
 * as torch computes it for an explicit scale_factor (fi_utils.py:67-70) */
int gvfi_resize_planes_f32(const float* src, float* dst, int planes, int H, int W, int Ho, int Wo,
                           float rscale, void* stream);
/* img_xs (B,3,2,H,W) in [0,1] -> normalised 2x-1 images, image order [frame0 of b=0..B-1, frame1 ...]:
 * act [2B,H,W,8] in dtype (channels 3..7 = 0) for convolutions and img4 [2B,H,W,4] float for warps */
int gvfi_prep_images(const float* img_xs, void* act, float* img4, int B, int H, int W, int dtype, void* stream);

/* ---- InstanceNorm2d (affine=False, eps=1e-5) of raft fnet, raft/extractor.py:26-30,50-58 */
int gvfi_instnorm_stats(const void* x, int ld, int C, int N, int HW, float* stats /*[N][C][2] x 8 bytes (64-bit fixed point, see
                        gvfi_conv_params.stats), zeroed*/,
                        int dtype, void* stream);
/* out = relu?( (x-mean)*rstd );  if res: out = relu(res + out) */
int gvfi_instnorm_apply(const void* x, int ld, int C, int N, int HW, const float* stats, int relu,
                        const void* res, int ldr, void* out, int ldo, int dtype, void* stream);

/* ---- correlation pyramid + lookup (raft/corr.py:23-93,127-165; raft/utils/utils.py:66-80) */
int gvfi_avgpool2_f32(const float* src, float* dst, long long maps, int h, int w, void* stream);
/* src_N (also gvfi_warp_nhwc / gvfi_warp_blend / gvfi_combine_warps_up; gvfi_copy_channels: src_npix): 0 = the source
 * holds N images; 0 < src_N <= N = the source holds src_N images and output image n reads source image n % src_N -- the
 * T timesteps of a pair run through frame synthesis as ONE batch [t][b] against the pair's t-independent tensors
 * (correlation pyramids, context features, images), the loop of gimmvfi_r.py:376-396 without replicating them */
int gvfi_corr_lookup(const float* l0, const float* l1, const float* l2, const float* l3,
                     const float* coords /*[N,h,w,2] (x,y)*/, void* out, int ldo, int dtype,
                     int N, int src_N, int h, int w, int h2, int w2, int radius, void* stream);
/* the same look-up with the windows staged through LDS (one workgroup per 8 queries; identical results); which of the two
 * the engine launches is decided by measurement (GVFI_LOOKUP_LDS, profiles/r3_lookup_ab.txt) */
int gvfi_corr_lookup_lds(const float* l0, const float* l1, const float* l2, const float* l3,
                     const float* coords /*[N,h,w,2] (x,y)*/, void* out, int ldo, int dtype,
                     int N, int src_N, int h, int w, int h2, int w2, int radius, void* stream);

/* patch matrix of a stride-1 zero-padded KHxKW convolution over c (tiny) channels: out[n,oy,ox, (kh*KW+kw)*c + ch],
 * zero-filled up to ldo (a whole K chunk); turns raft/update.py:100,107 (convf1, 2 -> 128, 7x7) and
 * modules/fi_components.py:177 (4 -> 128, 7x7) into 1x1 convolutions for gvfi_conv2d */
int gvfi_im2col(const void* x, int ld, int c, int N, int H, int W, int KH, int KW, int pad_h, int pad_w,
                void* out, int ldo, int dtype, void* stream);

/* ---- volume-free correlation lookup: the reference's native module alt_cuda_corr ------------------------------
 * (flowformer/alt_cuda_corr/correlation.cpp:19-54 `forward`, correlation_kernel.cu:18-119; caller raft/corr.py:96-124)
 * fmap1 [B,H1,W1,C], fmap2 [B,H2,W2,C] (dtype GVFI_F32 like the reference op, or GVFI_BF16), coords [B,N,H1,W1,2] float
 * (x, y) in fmap2 pixels -> corr [B,N,(2r+1)^2,H1,W1] float, channel = dy_index + (2r+1)*dx_index, bilinear in the
 * dot products, zeros outside fmap2, NOT divided by sqrt(C).  Every output element is written (no zero-fill needed).
 * radius <= 5, C <= 512. */
int gvfi_alt_corr_forward(const void* fmap1, const void* fmap2, const float* coords, float* corr, int B, int N,
                          int H1, int W1, int H2, int W2, int C, int radius, int dtype, void* stream);
/* 2x2 average pooling of an NHWC map [N,H,W,C] -> [N,H/2,W/2,C] (fmap2 pyramid of raft/corr.py:101-105) */
int gvfi_avgpool2_nhwc(const void* src, void* dst, int N, int H, int W, int C, int dtype, void* stream);

/* ---- RAFT glue (raft/raft.py:77-97,139-161) ------------------------------------------- */
int gvfi_coords_init(float* coords, int N, int h, int w, void* stream);
/* flow = coords1 - grid -> dst0[.,0:2] (+ zero pad to pad0 channels) and dst1[.,0:2] */
int gvfi_flow_pack(const float* coords1, void* dst0, int ld0, int pad0, void* dst1, int ld1,
                   int N, int h, int w, int dtype, void* stream);
/* second half of a "tap split" convolution (raft/update.py:6-14 FlowHead.conv2, 256 -> 2, 3x3): P[n,y,x, tap*C + c]
 * (float, pitch ldp) holds the per-tap partial sums of a 1x1 convolution with the KH*KW*C re-arranged filters;
 * out[n,y,x,c] = res + bias[c] + sum over taps of P at the tap's source pixel (zero padding).  res may alias out. */
int gvfi_tap_sum(const float* P, int ldp, int C, int KH, int KW, const float* bias, const float* res, int ldr,
                 float* out, int ldo, int N, int H, int W, void* stream);
/* the seam between two update iterations as ONE launch: coords_out = coords + tap sum of the 3x3x2 partial sums P (a second
 * tensor: workgroups read their neighbours' pixels; with P == NULL there is no update and coords_out is not written),
 * flow = coords1 - coords0 -> fl [N,h,w,ldf] (2 channels + zeros up to padf) and xb (2 channels at pitch ldx, may be
 * NULL), and the 7x7x2 im2col of that flow -> col [N,h,w,ldc] (98 entries, K order (kh, kw, c), zero padded).  Bit-identical
 * to gvfi_tap_sum + gvfi_flow_pack + gvfi_im2col in sequence (raft/raft.py:150,158-159; raft/update.py:6-14,100). */
int gvfi_flow_step(const float* P, int ldp, const float* bias, const float* coords, float* coords_out, void* fl, int ldf,
                   int padf, void* xb, int ldx, void* col, int ldc, int N, int h, int w, int dtype, void* stream);
int gvfi_convex_upsample(const float* coords1, const void* mask, int ldm, int mask_f32, float* flow_up,
                         int N, int h, int w, int dtype, void* stream);

/* ---- flow normalisation (modules/fi_utils.py:52-64) ----------------------------------- */
/* scaler[b] = max |f01[b]|,|f10[b]| ; scaler must be zeroed by the caller */
int gvfi_flow_absmax(const float* f01, const float* f10, float* scaler, int B, int HW, void* stream);
/* nf[0:B] = (f01/s+1)/2, nf[B:2B] = (-f10/s+1)/2 -> act [2B,H,W,pad] dtype and nflow float (B,2,2,H,W) */
int gvfi_flow_normalize(const float* f01, const float* f10, const float* scaler, void* act, int ld, int pad,
                        float* nflow_out, int B, int H, int W, int dtype, void* stream);
int gvfi_flow_unnormalize(const float* ninr /*[B,H,W,2]*/, const float* scaler, float* flow_t /*[B,H,W,2]*/,
                          float* ninr_nchw /* (B,2,1,H,W) or NULL */, int B, int HW, void* stream);

/* ---- splatting (gimmvfi_r.py:444-492; modules/softsplat.py:286-352,371-421) ------------ */
int gvfi_splat_weights(const float* f01, const float* f10, const float* gfilt9, float alpha_v, float alpha_fe,
                       float* z0, float* z1, int B, int H, int W, void* stream);
/* acc[B,H,W,C+1] (float, zeroed) += splat of [lat*Z, Z] by flow*tscale[b] */
int gvfi_softsplat_accum(const void* lat, int ldl, int C, const float* flow, const float* z, const float* t,
                         int one_minus_t, float* acc, int B, int H, int W, int dtype, void* stream);
int gvfi_softsplat_normalize(const float* acc, int C, void* dst, int ldd, long long npix, int dtype, void* stream);
/* the same softmax splat + "linear-zeroeps" normalisation as a DETERMINISTIC gather, both directions in one launch
 * (softsplat.py:286-352, 371-421; gimmvfi_r.py:171-193): gvfi_softsplat_lists enters every source pixel into the list
 * of its target cell (floor(x + t_d * flow), one integer exchange per pixel); gvfi_softsplat_gather walks, per target
 * pixel, the four cells whose sources reach it and adds their contributions in ascending source index -- no float
 * atomics, a result independent of scheduling.  head: int [2][B][H+1][W+1] set to -1 by the caller; next: int [2][B][H][W]
 * (scratch).  lat [B,H,W,ldl]: direction d's 16 latent channels at [16 d, 16 d + 16); dst likewise; t [B]: direction 0
 * uses t, direction 1 uses 1 - t. */
int gvfi_softsplat_lists(const float* f01, const float* f10, const float* t, int* head, int* next, int B, int H, int W,
                         void* stream);
int gvfi_softsplat_gather(const void* lat, int ldl, const float* f01, const float* f10, const float* z0, const float* z1,
                          const float* t, const int* head, const int* next, void* dst, int ldd, int B, int H, int W,
                          int dtype, void* stream);
/* The reference's native op with its own contract -- softsplat_func.forward / CuPy kernel `softsplat_out`
 * (modules/softsplat.py:358-446): tenIn (N,C,H,W) f32, tenFlow (N,2,H,W) f32, tenOut (N,C,H,W) f32 that the CALLER has
 * zero-initialised (softsplat.py:362-364), accumulated with float atomics on `stream`. */
int gvfi_softsplat_out_nchw(const float* tenIn, const float* tenFlow, float* tenOut, int N, int C, int H, int W,
                            void* stream);

/* ---- INR (modules/hyponet.py:71-146, modules/coord_sampler.py:15-43) -------------------- */
/* dst[.,0:C]=latent, dst[.,C:C+3]=coord(t,y,x), rest 0 up to pad */
int gvfi_inr_pack(const void* lat, int ldl, int C, const float* coord, void* dst, int ldd, int pad,
                  long long npix, int dtype, void* stream);
/* fused hypo-network (modules/hyponet.py:71-146 with the GIMM-VFI-R config: 5 layers, 32 latent + 3 coordinates ->
 * 128 -> 128 -> 128 -> 128 -> 2, sin activations, F.normalize'd weights and output_bias folded by the caller):
 * out[npix,2] (float) from lat[npix, ldl] (first 32 channels, bf16) and coord[npix,3] (float).  bf16 mode only
 * (returns -2 for GVFI_F32: run the layers through gvfi_conv2d).  wfrag / bias are the images built by
 * gvfi_inr_mlp_pack (host function, no GPU work) from row-major [out][in] float weights w[0..4] and biases b[0..4];
 * their sizes come from gvfi_inr_mlp_pack_sizes. */
int gvfi_inr_mlp_pack_sizes(int* wfrag_bytes, int* bias_floats);
int gvfi_inr_mlp_pack(const float* const* w, const float* const* b, void* wfrag_bf16, float* bias);
int gvfi_inr_mlp(const void* lat, int ldl, const float* coord, const void* wfrag, const float* bias,
                 float* out, long long npix, int dtype, void* stream);

/* ---- generic NHWC glue (modules/fi_utils.py:19-49,67-70; F.pixel_shuffle) --------------- */
/* bilinear resize (align_corners=False, rscale = 1/scale_factor), dst = mul * resize(src) */
int gvfi_resize_nhwc(const void* src, int lds, int src_f32, void* dst, int ldd, int dst_f32, int C,
                     int N, int H, int W, int Ho, int Wo, float rscale, float mul, int dtype, void* stream);
/* backward warp, border padding, align_corners=True; flow float [N,H,W,ldf] at channel foff, scaled by fmul */
int gvfi_warp_nhwc(const void* src, int lds, int src_f32, const float* flow, int ldf, float fmul,
                   void* dst, int ldd, int dst_f32, int C, int N, int src_N, int H, int W, int dtype, void* stream);
int gvfi_pixel_shuffle2(const void* src, int lds, void* dst, int ldd, int Cout, int N, int H, int W,
                        int dtype, void* stream);
/* dst[.,0:C] = mul*src[.,0:C] (+ add[.,0:C]) with independent dtypes/pitches */
int gvfi_copy_channels(const void* src, int lds, int src_f32, const void* add, int lda, int add_f32,
                       void* dst, int ldd, int dst_f32, int C, float mul, long long npix, long long src_npix, int dtype,
                       void* stream);
/* per-sample time scaling of a flow field: dst0 = -t*f, dst1 = (1-t)*f   (gimmvfi_r.py:239-240) */
int gvfi_flow_split_t(const float* flow_t, const float* t, float* ft0, float* ft1, int B, int HW, void* stream);
/* lookup coordinates of gimmvfi_r.py:494-507: c0 = grid + fl1/(1-t), c1 = grid + fl0/t */
int gvfi_lookup_coords(const float* fl0, const float* fl1, const float* t, float* c0, float* c1,
                       int B, int h, int w, void* stream);

/* ---- GIMM-VFI-F: FlowFormer flow estimator glue (flowformer/core/FlowFormer/LatentCostFormer/*; the Twins-SVT
 * backbone is timm's twins_svt_large, vendored classes twins.py:814-983,1028-1150).  Token tensors are row matrices
 * [rows][ld] in `dtype`; the linear layers, patch / sub-sampling convolutions, cost volume and the GMA contractions go
 * through gvfi_conv2d. ------------------------------------------------------------------------------------------- */
/* nn.LayerNorm(C): y = (x - mean) * rsqrt(var + eps) * gamma + beta   (twins.py:1146,1169; encoder.py:65,236-237);
 * x is float when x_f32 (the residual streams of the transformer blocks are kept in float), y is in dtype */
int gvfi_layernorm(const void* x, int ldx, int x_f32, const float* gamma, const float* beta, float eps, void* y,
                   int ldy, long long rows, int C, int dtype, void* stream);
/* PEG (twins.py:1100-1119): y = x + depthwise3x3(x) + bias, NHWC, w float [9][C]; io_f32: x, y float */
int gvfi_dwconv3x3_res(const void* x, int ldx, const float* w, const float* bias, void* y, int ldy, int io_f32,
                       int N, int H, int W, int C, int dtype, void* stream);
/* LinearPositionEmbeddingSine (attention.py:170-182): out[row, 0:dim] (+)= enc(scale * coords[row % period] + offset),
 * coords float [period][2] (x, y) */
int gvfi_pos_embed(const float* coords, long long period, float scale, float offset, int dim, void* out, int ldo,
                   long long rows, int accumulate, int dtype, void* stream);
/* first cost-map convolution (encoder.py:39-41,70-75): Conv2d(1,16,6,s2,p2)+ReLU over float maps [maps][H][W]
 * (zero-extended right/bottom to Ho*2 x Wo*2) -> out [maps][Ho][Wo][ldo >= 16]; w float [36][16] */
int gvfi_cost_embed1(const float* vol, const float* w, const float* bias, void* out, int ldo, long long maps,
                     int H, int W, int Ho, int Wo, int dtype, void* stream);
/* MemoryDecoder.encode_flow_token (decoder.py:237-255): (2r+1)^2 bilinear taps of cost map q around coords[q] */
int gvfi_cost_lookup(const float* maps, const float* coords, void* out, int ldo, long long Q, int h, int w,
                     int radius, int dtype, void* stream);
/* locally-grouped attention over ws x ws windows of an H x W token grid (twins.py:814-867, 331-427); kpad/vpad float
 * [ws*ws][heads*head_dim]: key / value of the window positions outside the grid; head_dim 8, 16 or 32.  (bf16, head_dim
 * 16 / 32: the MFMA path of attn_mfma.hip rounds these float tables to bf16 like every other key / value -- they are
 * MFMA operands there; the scalar kernels of the float mode use them as given) */
int gvfi_attn_window(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const float* kpad,
                     const float* vpad, void* out, int ldo, int n_img, int H, int W, int ws, int heads,
                     int head_dim, float scale, int dtype, void* stream);
/* attention of NQ queries against M keys per group (g1 < G1, g0 < G0); rows: query g1*qb1 + g0*qb0 + i*qs,
 * key/value g1*kb1 + g0*kb0 + j*ks, output g1*ob1 + g0*ob0 + i*os (twins.py:870-925, 430-546; attention.py:10-66;
 * encoder.py:214-346; decoder.py:35-120) */
int gvfi_attn_global(const void* q, int ldq, long long qb1, long long qb0, long long qs, const void* k, int ldk,
                     const void* v, int ldv, long long kb1, long long kb0, long long ks, void* out, int ldo,
                     long long ob1, long long ob0, long long os, long long G1, int G0, int NQ, int M, int heads,
                     int head_dim, float scale, int dtype, void* stream);
/* MFMA forms of the two attentions (csrc/attn_mfma.hip; bf16 -- IEEE half: the _f16 entry points below --, head_dim 16 or 32): one wave per (group, head) keeps the key
 * fragments and the transposed value fragments in registers and walks blocks of 32 queries -- S^T = K Q^T, soft-max in
 * registers, O^T = V^T P^T, no LDS.  gvfi_attn_global (M in 9..128, NQ >= 16) and gvfi_attn_window (ws == 7) route here by
 * themselves when gvfi_attn_mfma_ok says 1 (environment GVFI_ATTN_MFMA=0 keeps the scalar kernels); -3 = a pointer or pitch
 * the 16-byte loads cannot take. */
int gvfi_attn_mfma_ok(int window, int M, int NQ, int head_dim, int ws, int dtype);
int gvfi_attn_global_mfma(const void* q, int ldq, long long qb1, long long qb0, long long qs, const void* k, int ldk,
                          const void* v, int ldv, long long kb1, long long kb0, long long ks, void* out, int ldo,
                          long long ob1, long long ob0, long long os, long long G1, int G0, int NQ, int M, int heads,
                          int head_dim, float scale, void* stream);
int gvfi_attn_window_mfma(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const float* kpad,
                          const float* vpad, void* out, int ldo, int n_img, int H, int W, int ws, int heads,
                          int head_dim, float scale, void* stream);
/* the same two kernels on IEEE-half operands (v_mfma_f32_32x32x16_f16; q / k / v / out are GVFI_F16 tensors): the Twins encoders
 * of GIMM-VFI-F when the precision policy runs the stage "enc" in half ("enc:f16", round 5: their bf16 operand rounding costs
 * the two hardest reference fixtures 2 - 2.5 dB, profiles/r5_f_policy_enc_cost.txt).  gvfi_attn_global / gvfi_attn_window route
 * here for dtype GVFI_F16. */
int gvfi_attn_global_mfma_f16(const void* q, int ldq, long long qb1, long long qb0, long long qs, const void* k, int ldk,
                              const void* v, int ldv, long long kb1, long long kb0, long long ks, void* out, int ldo,
                              long long ob1, long long ob0, long long os, long long G1, int G0, int NQ, int M, int heads,
                              int head_dim, float scale, void* stream);
int gvfi_attn_window_mfma_f16(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const float* kpad,
                              const float* vpad, void* out, int ldo, int n_img, int H, int W, int ws, int heads,
                              int head_dim, float scale, void* stream);
/* out = [x | ctx[cimg(im)]] (+ positional code of the window position (enc_mode 1) or grid position (2)) for the
 * context-aware attention of the cost encoder (twins.py:366-395, 465-493); cimg reproduces the reference's
 * context.repeat() tiling over (batch, latent token): nb = pairs per direction, K = latent tokens */
int gvfi_ff_xqk(const void* x, int ldx, int Cx, const void* ctx, int ldc, int Cc, void* out, int ldo, int n_img,
                int H, int W, int K, int nb, int enc_mode, int ws, const float* enc_table, int dtype, void* stream);
/* enc_table of gvfi_ff_xqk (optional, NULL = evaluate the code per element): float [ws*ws (enc_mode 1) or H*W (2)][Ct]
 * = LinearPositionEmbeddingSine (attention.py:170-182) of the positions gvfi_ff_xqk adds, once per forward */
int gvfi_ff_pos_table(float* table, int H, int W, int Ct, int enc_mode, int ws, void* stream);
/* out[row, 0:C] = table[(row / P) % K]  (the learned latent tokens broadcast over the cost maps, encoder.py:420) */
int gvfi_tile_rows(const float* table, void* out, int ldo, int out_f32, long long rows, int P, int K, int C,
                   int dtype, void* stream);
/* row soft-max of the GMA similarity (gma.py:70-74): x float [rows][n] -> y [rows][ldy] in dtype, pad columns zeroed */
int gvfi_softmax_rows(const float* x, int n, void* y, int ldy, long long rows, int dtype, void* stream);

/* ---- frame synthesis glue (modules/fi_components.py:57-94,255-340; gimmvfi_r.py:213-220,305-308) */
/* out = clamp((sigmoid(mask)*warp(img0,f0) + (1-sigmoid(mask))*warp(img1,f1) + 1)/2, 0, 1), NCHW float */
int gvfi_warp_blend(const float* img4_0, const float* img4_1, const float* f0, const float* f1,
                    const float* mask, float* out_nchw, int B, int src_B, int H, int W, void* stream);
/* multi_flow_combine front half: 3 candidate blends + residual -> act [B,H,W,pad>=9] dtype, mean [B,H,W,4] float.
 * dec = final decoder output float [B,H,W,24] = [flow0(6) flow1(6) mask(3, post-sigmoid) res(9)] */
int gvfi_combine_warps(const float* img4_0, const float* img4_1, const float* dec, int ldd, void* act, int lda,
                       int pad, float* mean4, int B, int H, int W, int dtype, void* stream);
/* the same with the decoder output given at the working resolution [B,H,W,24] (DS_SCALE < 1, gimmvfi_r.py:294-303):
 * bilinear up-sampling to (Hf,Wf) (align_corners=False; flows x Hf/H) fused in, plus the planar (B,3,2,Hf,Wf) float
 * copies of the up-sampled flows that forward() returns as flowt0_pred[0] / flowt1_pred[0] (NULL = not wanted).
 * H == Hf is allowed (plain multi_flow_combine + the planar flows). */
int gvfi_combine_warps_up(const float* img4_0, const float* img4_1, const float* dec, int ldd, int H, int W, void* act,
                          int lda, int pad, float* mean4, float* flow0_planar, float* flow1_planar, int B, int src_B,
                          int Hf, int Wf, int dtype, void* stream);
/* final decoder head fix-up (fi_components.py:331-340): dec[.,0:6]+=flow0 x3, [6:12]+=flow1 x3,
 * [12:15] = sigmoid(dec + mask) ; flows float [.,2], mask float [.,1] */
int gvfi_decoder_head(float* dec, int ldd, const float* flow0, const float* flow1, const float* mask,
                      long long npix, void* stream);
/* imgt = clamp((x+1)/2, 0, 1): float NHWC (ld, 3 ch) -> float NCHW (B,3,H,W) */
int gvfi_finalize_image(const float* x, int ld, float* out_nchw, int B, int H, int W, void* stream);
/* float NHWC [N,H,W,ld] channels [0,C) -> float NCHW (N,C,H,W) */
int gvfi_nhwc_to_nchw_f32(const float* src, int ld, float* dst, int C, int N, int H, int W, void* stream);

/* output frame (B,3,H,W) float in [0,1] -> uint8 [B,H,W,3] (truncation of x*255, src/video_Nx.py:192-196);
 * the unit that is gathered to rank 0 over RCCL in multi-GPU runs */
int gvfi_frames_to_u8(const float* src_nchw, unsigned char* dst_nhwc, int B, int H, int W, void* stream);
/* the CLI's side-by-side output frames composed on the device (src/video_Nx.py:139-151,198-216): frames = the padded float
 * (pairs + 1, 3, Hp, Wp) input frames of a block of consecutive pairs (un-padded picture at (pad_top, pad_left), H0 x W0),
 * pred_u8 = its interpolated frames [pairs, N-1, H0, W0, 3] RGB; out [pairs * N + lead, H0, 2 * W0, 3] BGR uint8: an
 * optional leading [orig 0 | orig 0], then per pair N-1 x [orig j | interpolated i] and [orig j+1 | orig j+1].  Originals
 * are (float * 255) truncated, as the reference's astype(np.uint8). */
int gvfi_compose_sbs_u8(const float* frames, int n_frames, int Hp, int Wp, int pad_top, int pad_left,
                        const unsigned char* pred_u8, int pairs, int N, int lead, unsigned char* out, int H0, int W0,
                        void* stream);
/* flow colour coding of the CLI's flow.mp4 (reference src/utils/flow_viz.py:20-136 flow_to_image, src/video_Nx.py:199-207):
 * flow = n_img planar (u, v) float images img_stride floats apart, wheel = the 55 x 3 Middlebury wheel, radmax_zeroed =
 * n_img zeroed words of scratch (per-image max radius), out = [n_img][h][w][3] uint8 (BGR when bgr != 0) */
int gvfi_flow_to_image(const float* flow, long long img_stride, int n_img, int h, int w, const float* wheel,
                       unsigned* radmax_zeroed, unsigned char* out, int bgr, void* stream);

/* library identity / self-check */
const char* gvfi_version(void);
int gvfi_device_ok(void);   /* 1 if a gfx950 device is present and usable */

/* Per-token chain of three small linear layers (<= 128 -> 64 -> 64 -> 64 features) with a LayerNorm after layer
 * `ln_after`, an optional LinearPositionEmbeddingSine of `coords` added behind it, GELUs and residuals: the two halves of
 * the flow-token path of FlowFormer's MemoryDecoder iteration around its cross-attention (decoder.py:237-255 flow-token
 * encoder + decoder.py:84-120 CrossAttentionLayer norm1 / q;  proj / norm2 / ffn) in ONE launch each instead of 5 + 4
 * (csrc/token_chain.hip; GVFI_BF16 / GVFI_F16 only).  Row r of the first layer's input = [in0[r][0:k0a] | in1[r][0:128-k0a]].
 *   y0 = act0(W0 x + b0) (+ res0[r]);  [ln_after == 0: LN + position code]
 *   y1 = act1(W1 y0' + b1) -> out1 (optional);  [ln_after == 1: LN + position code]
 *   y2 = W2 y1' + b2 (+ y0 when res2_from0) -> out2.
 * Every intermediate is rounded to the activation type where the unfused sequence stores a tensor.  wfrag: the three weight
 * matrices in MFMA order, fragment (layer, 32-row block mb, k-step kk) = 64 lanes x 8 values, lane l holds
 * W[32 mb + (l & 31)][16 kk + 8 (l >> 5) .. +8]; layer 0 has 2 x 8 fragments (K = 128), layers 1 / 2 have 2 x 4. */
typedef struct {
    const void* in0; int ld0;
    const void* in1; int ld1;
    int k0a;
    const void* wfrag;
    const float* bias;                  /* [3][64] */
    const float* ln_g; const float* ln_b; float eps; int ln_after;
    const float* coords; long long period;      /* position code of coords[(r % period)] (x, y), or NULL */
    int act0, act1;                     /* GVFI_ACT_NONE or GVFI_ACT_GELU */
    const void* res0; int ldr0;
    int res2_from0;
    void* out1; int ldo1;
    void* out2; int ldo2;
    long long rows; int dtype;
} gvfi_token_chain_params;
int gvfi_token_chain(const gvfi_token_chain_params* p, void* stream);

#ifdef __cplusplus
}
#endif
#endif
