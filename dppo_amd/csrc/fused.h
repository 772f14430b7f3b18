// Fused row-tile MLP kernels for the big-batch side of the path (log-prob precompute, PPO update).
//
// Same engine as the sampler (sampler.hip), widened to MR row sub-tiles of 16 rows per workgroup: a persistent
// 512-thread workgroup walks 16*MR-row tiles; the whole residual MLP runs with the activations resident in LDS
// (XOR-swizzled [rows][H] images) and the residual stream in registers, the weights arrive as per-wave
// fragment streams from L2 through a register ring that keeps prefetching across layers AND across tiles.
// Only what the weight-gradient GEMMs need is written to HBM (training: act(h_b), act(z1_b), elem(h_nb);
// inference: nothing but the network output).  The layer-by-layer gemm_nt path remains as the general fallback
// and as an independent cross-check (tuning knob 1).
//
//  fused_forward : in [M][ld_in] -> out [M][ldout] (+ stored activations)
//  fused_backward: d_out [M][Kpo] + stored activations -> dh_b (b = nb..0), dz1_b, per-tile column sums
#pragma once
#include "common.h"
#include "sampler.h"

namespace dppo {

struct FusedGeom {  // backward stream geometry (the forward one is SamplerGeom)
  int KSB0;         // k-step positions of the dh = d_out . Wout layer (Kpo padded to PD*KB)
  int KpB0;         // its padded K in elements
  int total_pos;    // KSB0 + 2*nb*KSH
  size_t frags_per_wave;
};
template <class P>
FusedGeom fused_geom(const dppo_net_desc& d);

struct FusedFwdArgs {
  const u32x4* wstream;  // forward hidden-layer stream of the net (same image the sampler uses)
  const u32x4* ostream;  // out-layer fragments, [ks][to][lane]
  const float* params;
  int bias_off[2 + 2 * MAX_BLOCKS];
  int ln_off[4 * MAX_BLOCKS];  // per block: norm1.weight, norm1.bias, norm2.weight, norm2.bias (use_ln only)
  int use_ln;
  float* ln_stats;  // training + use_ln: [nb][2 norms][M][2] (mean, rstd) of every LayerNorm input row, else null
  const void* in;  // [M][ld_in] elem ; columns >= ld_in are zero
  int ld_in;
  int M, Kp0, nb, act, out_dim;
  int in_valid;  // true input width (algorithmic FLOP accounting only)
  int consts_lds;  // set by the launcher: bit 0 = biases staged in LDS, bit 1 = out-layer fragments staged in LDS
  // training-mode stores, each [M][H] elem (null = skip)
  void* a1[MAX_BLOCKS];        // act(h_b)
  void* a2[MAX_BLOCKS];        // act(z1_b)
  void* z1[MAX_BLOCKS];        // z1_b          (needed for Mish')
  void* hpre[MAX_BLOCKS + 1];  // elem(h_b)     (needed for Mish'; hpre[nb] = hE is always the out-layer input)
  float* out;  // [M][ldout]
  int ldout;
  // merged top (one-block networks, fused_can_merge()): the block's second layer is folded into the out layer --
  // out = (Wout W0) x + (Wout W2) act(z1) + cbias2; hpre[nb] is then neither formed nor stored (fused_forward_merged_kernel)
  int merge_top, ks0v;      // ks0v: k-steps of the input that hold data, ceil(in_dim / KB)
  const u32x4* ostream0;    // fragments of Wout W0, [ks][to][lane]
  const u32x4* ostream2;    // fragments of Wout W2, [ks][to][lane]
  const float* cbias2;      // [out_dim] bout + Wout (b0 + b2)
  // K-major fragment stores (gemm.h, GemmTNFrag; merged kernel, bf16, training): act(h_0) and act(z1) -- read by nothing
  // but the weight-gradient GEMMs -- are written as the MFMA operand fragments those want instead of row-major; a1[0] /
  // a2[0] are then not written.  [k-step][H / 16][64 lanes] u32x4 each, k-steps of 32 rows, whole tiles.
  u32x4* a1f;
  u32x4* a2f;
};

struct FusedBwdArgs {
  const u32x4* bstream;  // backward stream: [dh layer (Wout^T)][top block: (Wout . W2)^T as short as the dh layer, W1^T]
                         // [lower blocks, top down: W2^T, W1^T].  d loss / d h_nb = d_out . Wout has rank <= out_dim,
                         // so the top block's dh . W2 is d_out . (Wout . W2): a K = out_dim layer, not a K = H one.
  const void* d_out;     // [M][Kpo] elem, zero padded
  int ld_dout;
  int M, KpB0, nb, act;
  int out_valid;  // true output width of the network (algorithmic FLOP accounting only)
  // LayerNorm blocks: parameters (gamma/beta), saved row statistics; m1/m0 then hold the PRE-LayerNorm tensors z1_b / h_b
  const float* params;
  int ln_off[4 * MAX_BLOCKS];
  int use_ln;
  const float* ln_stats;  // [nb][2][M][2] (mean, rstd) from the forward
  // derivative sources, [M][H] elem: ReLU uses the activated copies (a > 0), Mish the pre-activations
  const void* m1[MAX_BLOCKS];  // for act'(z1_b): a2_b (ReLU) or z1_b (Mish)
  const void* m0[MAX_BLOCKS];  // for act'(h_b) : a1_b (ReLU) or hpre_b (Mish)
  // outputs, [M][H] elem
  void* dh[MAX_BLOCKS + 1];  // dh[b] = d loss / d h_b  (dh[nb] = d_out . Wout)
  void* dz1[MAX_BLOCKS];
  // per-tile column sums: [slots][tiles][H] f32.  Slots 0..nb: dh[nb..0]; nb+1..2nb: dz1[nb-1..0] (bias gradients);
  // with LayerNorm 4 more per block, top block first: d gamma1, d beta1, d gamma2, d beta2; then, when dout_slot >= 0,
  // one slot with the column sums of d_out itself (the out-layer bias gradient; first KpB0 columns of the slot)
  float* colsum;
  int dout_slot;
  int dbg;  // timing experiments only (tuning knob 8): bit 1 = do not fetch the derivative sources (wrong results)
  int one_block;  // fused_backward_one_kernel (fused_bwd_one_block(), nb == 1): dh[1] and column-sum slot 0 are not produced
  // K-major fragment stores (gemm.h, GemmTNFrag; one-block kernel, bf16): dz1 and dh_0 as fragments instead of row-major
  // (dz1[0] / dh[0] are then not written), plus fragment copies of the two small operands the GEMMs contract them with: the
  // d_out tile (first dof_nt feature tiles of it) and the network's input rows x ([M][ld_x] elem, x_nt = ld_x / 16 tiles)
  u32x4* dz1f;
  u32x4* dh0f;
  u32x4* doutf;
  int dof_nt;
  const void* x;
  int ld_x;
  u32x4* xf;
  // In-kernel first-layer weight gradient (one-block kernel, bf16; fused_dw0_shape()): dh_0 is never stored.  Every persistent
  // workgroup keeps dW0_part[H][32] = sum over ITS tiles of bf16(dh_0)^T . x' in LDS and writes it once, at the end, to
  // dw0_slab[blockIdx.x][H][32] (f32); the caller reduces the gridDim.x slabs.  x' = 32 columns of the input rows xc
  // ([M][ld_xc] elem, ld_xc >= 64): column c of x' is column c of the row below xc_af and column c + xc_skip from there on.
  // dw0_round: the column sums of dh_0 (slot 1, the first layer's bias gradient) are taken of the ROUNDED values, the ones the
  // product saw (a caller that leaves a one-hot column out of the 32 rebuilds its sums from them).
  float* dw0_slab;
  const void* xc;
  int ld_xc, xc_af, xc_skip, dw0_round;
};
// workgroups (= slabs) the one-block backward will launch for M rows, and whether its in-kernel dW0 covers the network
template <class P>
int fused_bwd_one_grid(const dppo_net_desc& d, int64_t M);
bool fused_dw0_shape(const dppo_net_desc& d);

struct LossArgs;
// loss != null (fused_loss_shape(), merged forward, training): the policy half of the PPO loss runs in the kernel's epilogue
// (loss_dev.h): a.out is not written, loss->d_eps and loss->partial are
template <class P>
int launch_fused_forward(const dppo_net_desc& d, const FusedFwdArgs& a, hipStream_t s, const LossArgs* loss = nullptr);   // <0: shape not covered
bool fused_loss_shape(const dppo_net_desc& d);
template <class P>
int launch_fused_backward(const dppo_net_desc& d, const FusedBwdArgs& a, hipStream_t s);
void set_fused_short_tiles(int v);  // tuning knob 7
void set_fused_merge_fwd(int v);    // tuning knob 22
template <class P>
bool fused_can_merge(const dppo_net_desc& d);  // the forward of this network runs merged (and the backward must rebuild dWout)
template <class P>
int fused_rows_per_tile(const dppo_net_desc& d, bool one_block = false);  // rows per tile of the BACKWARD kernel (sizes the per-tile column sums), 0 if not covered
template <class P>
bool fused_bwd_one_block(const dppo_net_desc& d);  // shape covered by fused_backward_one_kernel (the caller adds: low-rank dW2 on)
void set_fused_bwd_one(int v);  // tuning knob 23
bool fused_frag_shape(const dppo_net_desc& d);  // the fragment-output kernel variants (FusedFwdArgs::a1f, FusedBwdArgs::dz1f) exist for it
void set_fused_compact(int v);  // tuning knob 25

// Fragment packing of a whole stream in ONE launch: layer l occupies positions [pos0, pos0 + KS); element
// (feature f, contraction index k) of its weight matrix is W[f*rs + k*cs] (rs = ld, cs = 1 for W; rs = 1, cs = ld for W^T)
struct PackLayer {  // 40 bytes: two networks' worth of layers must fit one kernel-argument block (pack_nets_kernel)
  const float* W;
  u32x4* stream;  // destination stream and its length in k-step positions (forward and backward streams differ)
  int rs, cs;
  int in_valid, KS, pos0;
  int total_pos;
};
inline PackLayer pack_layer(const float* W, int rs, int cs, int in_valid, int KS, int pos0, u32x4* stream, int total_pos) {
  PackLayer L;
  L.W = W, L.stream = stream, L.rs = rs, L.cs = cs, L.in_valid = in_valid, L.KS = KS, L.pos0 = pos0, L.total_pos = total_pos;
  return L;
}
struct PackStream {  // every layer of both streams of a network: one launch
  PackLayer layer[2 * (1 + 2 * MAX_BLOCKS)];
  int n_layers, TPW;
};
template <class P>
void launch_pack_stream(const PackStream& d, hipStream_t s);

// out[slot][c] = sum_t in[slot][t][c]  for c < n, one launch for all slots (bias gradients from per-tile column sums)
struct SlotOuts {
  float* out[6 * MAX_BLOCKS + 2];
  int n[6 * MAX_BLOCKS + 2];  // columns of each slot that are written (<= the slot width)
  int n_slots;
};
void launch_reduce_slots(const float* in, int tiles, int n, const SlotOuts& o, hipStream_t s);

}  // namespace dppo
