/* EXPERIMENTS, not part of libgimmvfi_hip.so: three kernels that were built, proven bit-identical to the launches they replace
 * and measured SLOWER or neutral end to end (profiles/r4_lin_kernel_ab.txt, r4_tokpath_ab_v2.txt, r5_gru_fused_ab_v2.txt; the
 * round-6 recurrence experiment profiles/r6_bm128_merged_ab.txt closed the topic).  They left the product library in round 6;
 * the sources are kept buildable (tools/experiments/build.sh -> tools/experiments/libgimmvfi_experiments.so) for anyone who
 * wants to repeat the A/Bs.  Nothing in gimm-vfi_amd/, tests/ or bench.py uses them. */
#ifndef GIMMVFI_EXPERIMENTS_H
#define GIMMVFI_EXPERIMENTS_H
#include "gimmvfi_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Linear layers (1x1, stride 1) on very many rows with K = c0 + c1 in {128, 192, 256, 512} and Cout <= 512 (K = 128),
 * <= 256 (K = 192 / 256), <= 128 (K = 512) -- the transformer linears of GIMM-VFI-F's Twins encoders / latent cost encoder
 * (twins.py:331-546, encoder.py:214-346): weights (w_layout 2, the fragment-ordered image) resident in registers for a
 * persistent loop over the rows, operands straight from the rows, bias + none / ReLU / GELU + optional residual (16-bit or
 * float), 16-bit or float output.  bf16 / IEEE half.  Selected with algo = 8; gvfi_conv2d_lin_eligible: 1 = worth routing
 * (>= 65536 rows), 2 = runnable, 0 = not this kernel's problem. */
int gvfi_conv2d_lin_eligible(const gvfi_conv_params* p);
int gvfi_conv2d_lin(const gvfi_conv_params* p, void* stream);

/* The whole flow-token path of one MemoryDecoder iteration as ONE launch (csrc/token_path.hip; decoder.py:237-255 look-up +
 * flow_token_encoder, :35-120 CrossAttentionLayer): gvfi_cost_lookup (radius 4) -> gvfi_token_chain `a` (GELU linear, linear =
 * query, LayerNorm + position code of `coords`, linear = q) -> the one-query attention of gvfi_attn_global over the K latent
 * tokens of the token's cost map (8 heads of 8, key | value rows of 128 features at kv[(img * K + j) * P + p]) ->
 * gvfi_token_chain `c` ([attention | query] linear + query, LayerNorm, GELU linear, linear + x).  Bit-identical to those four
 * launches.  Of `a` / `c` only wfrag, bias, ln_g, ln_b, eps, ln_after, act0, act1, res2_from0 are read (the tensors in between
 * never leave the chip).  rows = images * P tokens; maps float [rows][h*w]; coords float [rows][2]; taps_out [rows][ldt] receives
 * the 81 taps (cost_forward), out [rows][ldo] the 64 result features (cost_global), both in dtype (GVFI_BF16 / GVFI_F16). */
typedef struct {
    gvfi_token_chain_params a, c;
    const float* maps; const float* coords; int h, w, radius;
    void* taps_out; int ldt;
    const void* kv; int ldkv; int K; long long P; float scale;
    void* out; int ldo;
    long long rows; int dtype;
} gvfi_token_path_params;
int gvfi_token_path(const gvfi_token_path_params* p, void* stream);

/* One half of the SepConvGRU of the flow estimators' update block as ONE launch (csrc/gru_fused.hip; raft/update.py:58-73,
 * FlowFormer gru.py:130-160): z, r = sigmoid(conv_z / conv_r([h | x])), q = tanh(conv_q([r * h | x])), h' = (1 - z) h + z q with
 * 1 x 5 (vertical = 0) or 5 x 1 (vertical = 1) filters.  Replaces the gvfi_conv2d pair (GVFI_EPI_GRU_ZR, GVFI_EPI_GRU_Q) of one
 * half: a workgroup owns whole lines of the image (one row of W <= 64 pixels / two columns of H <= 32), stages their [h | x]
 * once in LDS and runs the three contractions without barriers; z never leaves registers, r * h never leaves LDS.  Bit-identical
 * to the two launches.  h [N,H,W,ldh] (128 channels), x [N,H,W,ldx] (cx = 128 or 256 channels: RAFT's [motion | flow] /
 * FlowFormer's [motion | flow | aggregated motion]), wzr / wq = the fragment-ordered weight images (w_layout 2 of gvfi_conv2d)
 * of the 256- / 128-output gate convolutions over [h | x], bzr [256] / bq [128] float biases (may be null), ctx_zr [N,H,W,ld_czr]
 * / ctx_q float pre-activation terms (the context share of the gate convolutions, evaluated once per forward; may be null), out
 * [N,H,W,ldo] = h'.  dtype GVFI_BF16 / GVFI_F16.  gvfi_gru_half_ok: 1 = this geometry is taken. */
typedef struct {
    int dtype;
    const void* h; int ldh;
    const void* x; int ldx; int cx;
    const void* wzr; const void* wq;
    const float* bzr; const float* bq;
    const float* ctx_zr; int ld_czr;
    const float* ctx_q; int ld_cq;
    void* out; int ldo;
    int N, H, W, vertical;
} gvfi_gru_params;
int gvfi_gru_half_ok(const gvfi_gru_params* p);
int gvfi_gru_half(const gvfi_gru_params* p, void* stream);

#ifdef __cplusplus
}
#endif
#endif
