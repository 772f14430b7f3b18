// C ABI of libdppo_hip.so (include/dppo_hip.h): argument checks, layouts, workspace carving and the
// launch sequences.  No allocation, no synchronisation, no global mutable state.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "fused.h"
#include "gemm.h"
#include "ppo.h"
#include "gaussian.h"
#include "gmm.h"
#include "unet.h"
#include "sampler.h"

using namespace dppo;

static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
namespace dppo {
void set_vis_mfma_attn(int v);  // vision.hip
}
namespace dppo {  // the same two helpers for the other translation units that export entry points (unet.hip)
int api_fail(int code, const char* msg) { return fail(code, "%s", msg); }
int api_check_launch() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail((int)e, "HIP launch failed: %s", hipGetErrorString(e));
  return 0;
}
}  // namespace dppo
// A fused kernel refused a shape that fused_ok() admitted: the packed image then holds no layered operands to fall back
// on, so the call must fail (reported by the check_launch() that ends every entry point).
static int g_fused_fault = 0;
static int check_launch() {
  if (g_fused_fault) {
    const int rc = g_fused_fault;
    g_fused_fault = 0;
    return fail(-1, "fused MLP kernel refused the launch (code %d): shape admitted by the packer but not by the kernel", rc);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail((int)e, "HIP launch failed: %s", hipGetErrorString(e));
  return 0;
}

// ------------------------------------------------------------------------------------------------
// layouts
// ------------------------------------------------------------------------------------------------
struct ParamLayout {  // float offsets into the flat fp32 parameter / gradient buffer (state-dict order)
  int64_t te1_w, te1_b, te2_w, te2_b, c1w, c1b, c2w, c2b, W0, b0, l1w[MAX_BLOCKS], l1b[MAX_BLOCKS], l2w[MAX_BLOCKS], l2b[MAX_BLOCKS],
      n1w[MAX_BLOCKS], n1b[MAX_BLOCKS], n2w[MAX_BLOCKS], n2b[MAX_BLOCKS], Wout, bout, total;
};
static ParamLayout param_layout(const dppo_net_desc& d) {
  ParamLayout L;
  memset(&L, 0, sizeof(L));
  int64_t o = 0;
  const int td = d.time_dim, H = d.hidden;
  if (d.kind == 0) {
    L.te1_w = o, o += 2 * td * td;
    L.te1_b = o, o += 2 * td;
    L.te2_w = o, o += td * 2 * td;
    L.te2_b = o, o += td;
    if (d.cond_hidden > 0) {
      L.c1w = o, o += (int64_t)d.cond_hidden * d.cond_dim;
      L.c1b = o, o += d.cond_hidden;
      L.c2w = o, o += (int64_t)d.cond_out * d.cond_hidden;
      L.c2b = o, o += d.cond_out;
    }
  }
  L.W0 = o, o += (int64_t)H * d.in_dim;
  L.b0 = o, o += H;
  for (int b = 0; b < d.n_blocks; ++b) {
    L.l1w[b] = o, o += (int64_t)H * H;
    L.l1b[b] = o, o += H;
    if (d.plain) continue;  // a plain MLP's hidden layers are single Linear(H, H) modules (mlp.py:46-74)
    L.l2w[b] = o, o += (int64_t)H * H;
    L.l2b[b] = o, o += H;
    if (d.use_layernorm) {
      L.n1w[b] = o, o += H;
      L.n1b[b] = o, o += H;
      L.n2w[b] = o, o += H;
      L.n2b[b] = o, o += H;
    }
  }
  L.Wout = o, o += (int64_t)d.out_dim * H;
  L.bout = o, o += d.out_dim;
  L.total = o;
  return L;
}

struct PackLayout {  // byte offsets into the packed image
  // every offset depends on (net, prec) only; the time table sits last so that only `total` grows with n_time
  size_t W0, W1[MAX_BLOCKS], W2[MAX_BLOCKS], Wout, W1T[MAX_BLOCKS], W2T[MAX_BLOCKS], WoutT, W0tT, sstream, ostream,
      bstream, wcomp, ostream2, cbias, w0comp, ostream0, cbias2, Wc1, Wc2, Wc2T, W0eT, temb, total;
  int Kp0, Kpo, tdp;
  int Kp0s, CNT0;  // one-block networks: padded K of the first-layer composite Wout . W0 and its out-stream k-steps per wave
  int Kpc, C1p, Ep;  // cond_mlp: padded K of the encoder layers (cond, hidden) and padded encoder width
};
static size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
// Width of a network's input rows x (and of the row-major first-layer operand): in_dim padded to 64 -- and, for a denoiser,
// with at least 24 spare columns behind in_dim: the one-hot of the row's denoising step lives there (temb_onehot_col), and
// without room for it the time-embedding gradient costs a gemm_nt over all rows, a segmented sum and two more launches
// (halfcheetah: in_dim 57, 7 spare of 64: +40 us per update until the rows became 128 wide).
static int row_kp0(const dppo_net_desc& d) {
  int k = round_up(d.in_dim, 64);
  if (d.kind == 0 && k - d.in_dim < 24) k += 64;
  return k;
}
template <class P>
static PackLayout pack_layout(const dppo_net_desc& d, int n_time) {
  PackLayout L;
  memset(&L, 0, sizeof(L));
  const size_t ES = P::ESIZE;
  const int H = d.hidden;
  L.Kp0 = row_kp0(d);
  L.Kpo = round_up(d.out_dim, 64);
  L.tdp = round_up(d.time_dim > 0 ? d.time_dim : 1, 16);
  size_t o = 0;
  L.W0 = o, o = al256(o + (size_t)H * L.Kp0 * ES);
  for (int b = 0; b < d.n_blocks; ++b) {
    L.W1[b] = o, o = al256(o + (size_t)H * H * ES);
    L.W2[b] = o, o = al256(o + (size_t)H * H * ES);
    L.W1T[b] = o, o = al256(o + (size_t)H * H * ES);
    L.W2T[b] = o, o = al256(o + (size_t)H * H * ES);
  }
  L.Wout = o, o = al256(o + (size_t)d.out_dim * H * ES);
  L.WoutT = o, o = al256(o + (size_t)H * L.Kpo * ES);
  if (d.kind == 0) L.W0tT = o, o = al256(o + (size_t)d.time_dim * H * ES);
  if (!d.plain && d.out_dim <= 128) {  // per-wave fragment streams: forward (sampler + fused forward), out layer, backward
    const SamplerGeom g = sampler_geom<P>(d);
    const FusedGeom fg = fused_geom<P>(d);
    L.sstream = o, o = al256(o + (size_t)SAMPLER_WAVES * g.hidden_frags_per_wave * 64 * 16);
    L.ostream = o, o = al256(o + (size_t)SAMPLER_WAVES * g.out_frags_per_wave * 64 * 16);
    L.bstream = o, o = al256(o + (size_t)SAMPLER_WAVES * fg.frags_per_wave * 64 * 16);
    L.wcomp = o, o = al256(o + (size_t)d.out_dim * H * 4);  // fp32 Wout . W2 of the top block (source of its composite layer)
    // merged out layer (inference): out = Wout . h_in + (Wout . W2) . act(z1) + cbias, cbias = bout + Wout . b2
    L.ostream2 = o, o = al256(o + (size_t)SAMPLER_WAVES * g.out_frags_per_wave * 64 * 16);
    L.cbias = o, o = al256(o + (size_t)round_up(d.out_dim, 16) * 4);
    if (d.n_blocks == 1) {  // the fused forward's merged out layer (FusedFwdArgs::merge_top)
      L.Kp0s = g.Kp0, L.CNT0 = (g.KS0 + SAMPLER_WAVES - 1) / SAMPLER_WAVES;
      L.w0comp = o, o = al256(o + (size_t)d.out_dim * L.Kp0s * 4);
      L.ostream0 = o, o = al256(o + (size_t)SAMPLER_WAVES * L.CNT0 * g.OT * 64 * 16);
      L.cbias2 = o, o = al256(o + (size_t)round_up(d.out_dim, 16) * 4);
    }
  }
  L.Kpc = round_up(d.cond_dim > 0 ? d.cond_dim : 1, 64);
  if (d.cond_hidden > 0) {  // observation encoder: row-major GEMM operands (small), W2^T and the encoder columns of W0
    L.C1p = round_up(d.cond_hidden, 64), L.Ep = round_up(d.cond_out, 64);
    L.Wc1 = o, o = al256(o + (size_t)d.cond_hidden * L.Kpc * ES);
    L.Wc2 = o, o = al256(o + (size_t)d.cond_out * L.C1p * ES);
    L.Wc2T = o, o = al256(o + (size_t)d.cond_hidden * L.Ep * ES);
    L.W0eT = o, o = al256(o + (size_t)d.cond_out * H * ES);
  }
  L.temb = o, o = al256(o + (size_t)n_time * (d.time_dim > 0 ? d.time_dim : 0) * 4);
  L.total = o;
  return L;
}

static int check_net(const dppo_net_desc* d) {
  if (!d) return fail(-1, "null net descriptor");
  if (d->kind != 0 && d->kind != 1) return fail(-1, "net.kind must be 0 (actor) or 1 (critic)");
  if (d->plain != 0 && d->plain != 1) return fail(-1, "plain must be 0 or 1");
  if (d->plain) {  // non-residual MLP trunk (model/common/mlp.py:27-81): layered GEMM path only
    if (d->hidden < 64 || d->hidden % 64) return fail(-1, "plain MLP: hidden=%d must be a positive multiple of 64", d->hidden);
    if (d->n_blocks < 1) return fail(-1, "plain MLP: at least two hidden widths (n_blocks = hidden-to-hidden layers >= 1)");
    if (d->use_layernorm || d->cond_hidden || d->cond_out) return fail(-1, "plain MLP: LayerNorm / cond_mlp are not built");
  } else if (d->hidden < 128 || d->hidden % 128) return fail(-1, "hidden=%d must be a positive multiple of 128", d->hidden);
  if (d->n_blocks < 0 || d->n_blocks > MAX_BLOCKS) return fail(-1, "n_blocks=%d out of [0,%d]", d->n_blocks, MAX_BLOCKS);
  if (d->act != DPPO_ACT_RELU && d->act != DPPO_ACT_MISH) return fail(-1, "activation %d unsupported", d->act);
  // (out_dim > 128 -- the mixture-of-Gaussians head, Ta*Da*num_modes outputs -- runs on the layered GEMM path only: no
  // fragment streams are packed, and such a descriptor is refused by the K-step sampler)
  if (d->out_dim < 1 || d->out_dim > 1024) return fail(-1, "out_dim=%d out of [1,1024]", d->out_dim);
  if (d->out_dim > 128 && d->kind == 0) return fail(-1, "a denoiser's out_dim = Ta*Da must be <= 128");
  if (d->kind == 0) {
    if (d->time_dim < 4 || d->time_dim % 2) return fail(-1, "time_dim=%d must be even and >= 4", d->time_dim);
    if (d->act_flat != d->out_dim) return fail(-1, "actor out_dim must equal act_flat");
    if ((d->cond_hidden > 0) != (d->cond_out > 0)) return fail(-1, "cond_hidden / cond_out must both be set or both 0");
    if (d->cond_hidden < 0 || d->cond_hidden > 1024 || d->cond_out > 256) return fail(-1, "cond_mlp dims out of range");
    if (d->cond_out > 0 && (d->act_flat + d->time_dim) % 4) return fail(-1, "cond_mlp needs Ta*Da + time_dim to be a multiple of 4");
    if (d->in_dim != d->act_flat + d->time_dim + (d->cond_out > 0 ? d->cond_out : d->cond_dim))
      return fail(-1, "actor in_dim mismatch");
  } else {
    if (d->in_dim != d->cond_dim) return fail(-1, "critic in_dim must equal cond_dim");
    if (d->cond_hidden || d->cond_out) return fail(-1, "critic has no cond_mlp");
  }
  if (d->in_dim < 1 || d->in_dim > 1024) return fail(-1, "in_dim=%d out of [1,1024]", d->in_dim);
  if (d->use_layernorm != 0 && d->use_layernorm != 1) return fail(-1, "use_layernorm must be 0 or 1");
  if (d->use_layernorm && !(d->hidden == 256 || d->hidden == 512 || d->hidden == 1024))
    return fail(-1, "LayerNorm blocks need hidden in {256, 512, 1024} (fused kernels only), got %d", d->hidden);
  return 0;
}
static int check_prec(int prec) {
  if (prec != DPPO_PREC_F32 && prec != DPPO_PREC_BF16) return fail(-1, "prec must be DPPO_PREC_F32 or DPPO_PREC_BF16");
  return 0;
}
#define DPPO_DISPATCH(prec, CALL)        \
  ((prec) == DPPO_PREC_F32 ? CALL(F32) : CALL(BF16))

static int g_use_fused = 1;  // tuning knob 1: 1 = fused row-tile kernels where the shape is covered, 0 = layered GEMMs

template <class P>
static bool fused_ok(const dppo_net_desc& d) {
  // the input tile (width round_up(in_dim, .)) and the d_out tile share an H-wide LDS image with the activations
  if (d.plain) return false;
  return (g_use_fused || d.use_layernorm) && fused_rows_per_tile<P>(d) > 0 && d.out_dim <= 128 && d.hidden <= 1024 &&
         sampler_geom<P>(d).Kp0 <= d.hidden && fused_geom<P>(d).KpB0 <= d.hidden;
}

// ------------------------------------------------------------------------------------------------
// pack
// ------------------------------------------------------------------------------------------------
static int g_pack_one = 1;  // tuning knob 13: one launch per network for all of its kernel-ready images
template <class P>
static int pack_impl(const dppo_net_desc& d, int n_time, const float* prm, char* pk, hipStream_t s,
                     PackNet* defer_pn = nullptr, ComposeJob* defer_cj = nullptr, bool* deferred = nullptr) {
  // defer_pn / defer_cj: the caller launches the one-launch pack (and the composite) itself, batched with another
  // network's (dppo_pack_nets); *deferred tells whether this network's could be deferred
  const ParamLayout pl = param_layout(d);
  const PackLayout L = pack_layout<P>(d, n_time);
  const int H = d.hidden;
  if (deferred) *deferred = false;
  if (d.plain) {  // row-major operands of the layered path only
    if (d.kind == 0) {
      launch_time_table(prm + pl.te1_w, prm + pl.te1_b, prm + pl.te2_w, prm + pl.te2_b, d.time_dim, n_time, (float*)(pk + L.temb), s);
      launch_transpose_cast<P>(prm + pl.W0, H, d.time_dim, d.in_dim, d.act_flat, pk + L.W0tT, H, s);
    }
    launch_cast_pad<P>(prm + pl.W0, H, d.in_dim, d.in_dim, pk + L.W0, L.Kp0, s);
    for (int b = 0; b < d.n_blocks; ++b) {
      launch_cast_pad<P>(prm + pl.l1w[b], H, H, H, pk + L.W1[b], H, s);
      launch_transpose_cast<P>(prm + pl.l1w[b], H, H, H, 0, pk + L.W1T[b], H, s);
    }
    launch_cast_pad<P>(prm + pl.Wout, d.out_dim, H, H, pk + L.Wout, H, s);
    launch_transpose_cast<P>(prm + pl.Wout, d.out_dim, H, H, 0, pk + L.WoutT, L.Kpo, s);
    return check_launch();
  }
  const bool one_launch = fused_ok<P>(d) && g_pack_one && pack_net_supports(d.time_dim);
  if (d.kind == 0 && !one_launch)
    launch_time_table(prm + pl.te1_w, prm + pl.te1_b, prm + pl.te2_w, prm + pl.te2_b, d.time_dim, n_time,
                      (float*)(pk + L.temb), s);
  if (!fused_ok<P>(d)) {  // row-major operand copies of the layer-by-layer gemm_nt path (unused by the fused kernels)
    launch_cast_pad<P>(prm + pl.W0, H, d.in_dim, d.in_dim, pk + L.W0, L.Kp0, s);
    for (int b = 0; b < d.n_blocks; ++b) {
      launch_cast_pad<P>(prm + pl.l1w[b], H, H, H, pk + L.W1[b], H, s);
      launch_cast_pad<P>(prm + pl.l2w[b], H, H, H, pk + L.W2[b], H, s);
      launch_transpose_cast<P>(prm + pl.l1w[b], H, H, H, 0, pk + L.W1T[b], H, s);
      launch_transpose_cast<P>(prm + pl.l2w[b], H, H, H, 0, pk + L.W2T[b], H, s);
    }
    launch_cast_pad<P>(prm + pl.Wout, d.out_dim, H, H, pk + L.Wout, H, s);
    // WoutT[h][o] = Wout[o][h] : src rows = out_dim, cols = H  -> dst [H][Kpo]
    launch_transpose_cast<P>(prm + pl.Wout, d.out_dim, H, H, 0, pk + L.WoutT, L.Kpo, s);
  }
  // W0tT[j][h] = W0[h][act_flat + j]
  if (d.kind == 0 && !one_launch)
    launch_transpose_cast<P>(prm + pl.W0, H, d.time_dim, d.in_dim, d.act_flat, pk + L.W0tT, H, s);
  if (d.cond_hidden > 0) {
    launch_cast_pad<P>(prm + pl.c1w, d.cond_hidden, d.cond_dim, d.cond_dim, pk + L.Wc1, L.Kpc, s);
    launch_cast_pad<P>(prm + pl.c2w, d.cond_out, d.cond_hidden, d.cond_hidden, pk + L.Wc2, L.C1p, s);
    // Wc2T[c][e] = Wc2[e][c] -> [cond_hidden][Ep] ; W0eT[e][h] = W0[h][act_flat + td + e] -> [cond_out][H]
    launch_transpose_cast<P>(prm + pl.c2w, d.cond_out, d.cond_hidden, d.cond_hidden, 0, pk + L.Wc2T, L.Ep, s);
    launch_transpose_cast<P>(prm + pl.W0, H, d.cond_out, d.in_dim, d.act_flat + d.time_dim, pk + L.W0eT, H, s);
  }
  if (d.out_dim > 128) return check_launch();  // layered path only (fused_ok() is false: the row-major operands above are it)
  {
    const SamplerGeom g = sampler_geom<P>(d);
    const FusedGeom fg = fused_geom<P>(d);
    // forward stream [L0][b: l1, l2]... and backward stream, top down: [dh = d_out . Wout][b = nb-1..0: W2^T, W1^T]
    // ("feature f, contraction index k" of a transposed layer is W[k][f] = W[k*H + f]); all in one launch
    PackNet pn;
    memset(&pn, 0, sizeof(pn));
    PackStream& ps = pn.ps;
    ps.TPW = g.TPW;
    u32x4* fwd = (u32x4*)(pk + L.sstream);
    u32x4* bwd = (u32x4*)(pk + L.bstream);
    ps.layer[ps.n_layers++] = pack_layer(prm + pl.W0, d.in_dim, 1, d.in_dim, g.KS0, 0, fwd, g.total_pos);
    for (int b = 0; b < d.n_blocks; ++b) {
      ps.layer[ps.n_layers++] = pack_layer(prm + pl.l1w[b], H, 1, H, g.KSH, g.KS0 + 2 * b * g.KSH, fwd, g.total_pos);
      ps.layer[ps.n_layers++] = pack_layer(prm + pl.l2w[b], H, 1, H, g.KSH, g.KS0 + (2 * b + 1) * g.KSH, fwd, g.total_pos);
    }
    ps.layer[ps.n_layers++] = pack_layer(prm + pl.Wout, 1, H, d.out_dim, fg.KSB0, 0, bwd, fg.total_pos);
    for (int b = d.n_blocks - 1, pos = fg.KSB0; b >= 0; --b) {
      if (b == d.n_blocks - 1) {  // top block: the composite (Wout . W2)^T, laid out like the Wout^T layer
        float* wc = (float*)(pk + L.wcomp);
        ComposeJobs cq;
        cq.n = 1;
        memset(&cq.j[0], 0, sizeof(cq.j[0]));
        cq.j[0].Wout = prm + pl.Wout, cq.j[0].W2 = prm + pl.l2w[b], cq.j[0].b2 = prm + pl.l2b[b], cq.j[0].bout = prm + pl.bout;
        cq.j[0].Wc = wc, cq.j[0].cbias = (float*)(pk + L.cbias), cq.j[0].H = H, cq.j[0].out_dim = d.out_dim;
        if (d.n_blocks == 1) {
          cq.j[0].W0 = prm + pl.W0, cq.j[0].b0 = prm + pl.b0, cq.j[0].W0c = (float*)(pk + L.w0comp);
          cq.j[0].cbias2 = (float*)(pk + L.cbias2), cq.j[0].in_dim = d.in_dim, cq.j[0].Kp0s = L.Kp0s;
        }
        if (one_launch && defer_cj != nullptr)
          *defer_cj = cq.j[0];
        else
          launch_compose(cq, s);
        ps.layer[ps.n_layers++] = pack_layer(wc, 1, H, d.out_dim, fg.KSB0, pos, bwd, fg.total_pos);
        pos += fg.KSB0;
      } else {
        ps.layer[ps.n_layers++] = pack_layer(prm + pl.l2w[b], 1, H, H, g.KSH, pos, bwd, fg.total_pos);
        pos += g.KSH;
      }
      ps.layer[ps.n_layers++] = pack_layer(prm + pl.l1w[b], 1, H, H, g.KSH, pos, bwd, fg.total_pos);
      pos += g.KSH;
    }
    if (one_launch) {  // + out-layer stream, time-embedding table and W0tT, all in the same launch
      int maxks = 0;
      for (int l = 0; l < ps.n_layers; ++l) maxks = ps.layer[l].KS > maxks ? ps.layer[l].KS : maxks;
      pn.ps_x = SAMPLER_WAVES * maxks * ps.TPW;
      pn.Wout = prm + pl.Wout, pn.out_dim = d.out_dim, pn.H = H, pn.OT = g.OT, pn.CNT = g.CNT;
      pn.ostream = (u32x4*)(pk + L.ostream);
      if (d.n_blocks >= 1) pn.Wc = (const float*)(pk + L.wcomp), pn.ostream2 = (u32x4*)(pk + L.ostream2);
      if (d.n_blocks == 1)
        pn.W0c = (const float*)(pk + L.w0comp), pn.ostream0 = (u32x4*)(pk + L.ostream0), pn.Kp0s = L.Kp0s, pn.CNT0 = L.CNT0;
      if (d.kind == 0) {
        pn.te_w1 = prm + pl.te1_w, pn.te_b1 = prm + pl.te1_b, pn.te_w2 = prm + pl.te2_w, pn.te_b2 = prm + pl.te2_b;
        pn.td = d.time_dim, pn.n_time = n_time, pn.temb = (float*)(pk + L.temb);
        pn.tsrc = prm + pl.W0, pn.t_rows = H, pn.t_cols = d.time_dim, pn.t_lds = d.in_dim, pn.t_coff = d.act_flat;
        pn.t_ldd = H, pn.tdst = pk + L.W0tT;
      }
      if (defer_pn != nullptr && (d.n_blocks == 0 || defer_cj != nullptr)) {
        *defer_pn = pn;
        if (deferred) *deferred = true;
      } else {
        launch_pack_net<P>(pn, s);
      }
    } else {
      launch_pack_stream<P>(ps, s);
      launch_pack_out<P>(prm + pl.Wout, d.out_dim, H, g.OT, g.CNT, (u32x4*)(pk + L.ostream), s);
      if (d.n_blocks >= 1)
        launch_pack_out<P>((const float*)(pk + L.wcomp), d.out_dim, H, g.OT, g.CNT, (u32x4*)(pk + L.ostream2), s);
      if (d.n_blocks == 1)
        launch_pack_out<P>((const float*)(pk + L.w0comp), d.out_dim, L.Kp0s, g.OT, L.CNT0, (u32x4*)(pk + L.ostream0), s);
    }
  }
  return check_launch();
}

// ------------------------------------------------------------------------------------------------
// MLP forward / backward drivers on the GEMM path
// ------------------------------------------------------------------------------------------------
struct Carver {
  char* base;
  size_t off, cap;
  void* take(size_t bytes) {
    off = al256(off);
    void* p = base ? base + off : nullptr;
    off += bytes;
    return p;
  }
};

template <class P>
struct MlpBufs {  // activations of one network for M rows
  void* in;                    // [M][Kp0] elem
  float* h[MAX_BLOCKS + 1];    // [M][H] f32 residual stream (h[0] = layer-0 output)
  void* a1[MAX_BLOCKS];        // act(h[b])
  void* z1[MAX_BLOCKS];        // l1 pre-activation (elem)
  void* a2[MAX_BLOCKS];        // act(z1)
  void* hpre[MAX_BLOCKS + 1];  // elem(h[b]); hpre[nb] = hE (out-layer input)
  void* hE;                    // elem(h[nb])
  float* out;                  // [M][ldout] f32
  int ldout;
  // backward
  void* d_out;  // [M][Kpo] elem
  void* dh;     // [M][H] elem
  void* dz1;    // [M][H] elem
  float* dtemb; // [M][tdp] f32 (actor)
  // cond_mlp (observation encoder) activations / gradients
  void* cin;     // [M][Kpc] elem : [obs | 0]
  void* ca;      // [M][C1p] act(z_c)
  void* cz;      // [M][C1p] z_c (Mish')
  void* d_enc;   // [M][Ep]
  void* d_cz;    // [M][C1p]
  void* dh_all[MAX_BLOCKS + 1];  // fused backward: dh[b] = d loss / d h_b, elem [M][H]
  void* dz1_all[MAX_BLOCKS];
  float* ln_stats;               // [nb][2][M][2] LayerNorm row statistics (training)
  const void* dh0_final;         // where the last backward left d loss / d h_0
  void* w0T;                     // [cond_dim][H] elem: layer 0's observation columns, transposed (obs_grad)
  float* dobs;                   // [M][round_up(cond_dim, 16)]: obs_grad's GEMM output (the epilogue stores 16-column groups)
  float* tile_colsum;            // [(2nb+1)][tiles][H]
  int tiles;
  float* slab;  // split-M partial weight gradients: a pool, one sub-slab per pending weight-gradient GEMM
  size_t slab_used;   // floats handed out since the last flush
  SlabJobs slab_jobs; // reductions pending on the pool (flush_slabs)
  GemmTNGroup tn_group;  // weight-gradient GEMMs pending on the pool: launched together by flush_slabs
  // K-major fragment mode (gemm.h GemmTNFrag; frag_ok()): the four tensors only the weight-gradient GEMMs read are written
  // as MFMA operand fragments by the one-block fused kernels -- into the SAME buffers a1[0], a2[0], dz1_all[0], dh_all[0]
  // (carved with rows rounded up to whole tiles) -- and the two small operands get fragment copies
  // Folded tail (gemm.h: GemmTN::red_cnt, GemmTNExtra; knob 35): the slab reductions happen inside the GEMM launch (last
  // workgroup to arrive at an output tile) and the bias sums / loss statistics ride in it, so no reduction launch follows
  bool fold;          // set by the caller between carve and backward: red_cnt was zeroed in this call (row builder)
  unsigned* red_cnt;  // [RED_CNT] tile arrival counters, right behind post_counter (zeroed with it, left zero)
  bool dw0;         // set by the caller between carve and the backward (dw0_ok()): the backward keeps dW0 on chip, dh_0 is never stored
  // riders of the next weight-gradient GEMM launch (knob 40; filled by mlp_backward, consumed by flush_slabs)
  bool ride;
  TailReduce ride_t;     // slab jobs (the backward kernel's dW0 slabs), bias-sum slots, loss statistics
  PostReduce ride_q;     // the time-embedding part (G == null: none)
  bool allow_frag;  // set by the caller between carve and forward: the whole pass (forward, backward, GEMMs) may run in it
  bool frag;        // decided by the forward (allow_frag && merged), obeyed by the backward
  u32x4* doutf;     // [ks][dof_nt][64]
  u32x4* xf;        // [ks][Kp0 / 16][64]
  int dof_nt;
  int64_t mpad;     // rows rounded up to the fused kernels' tile height
  GemmTNFragGroup tnf_group;
  // side streams to join into the flushing stream right behind the GEMM launch: the barrier packets (~10 us each even when
  // the event fired long ago) are then processed while the GEMMs run instead of at the end of the call
  hipStream_t join_s[2];
  int join_idx[2], n_join;
  float* part;  // column-sum / segment-sum partials
  float* lowrank;  // [out_dim][H] T = d_out^T . act(z1) of the top block (see lowrank_dw_kernel)
  float* lowrank_u;  // [out_dim][Kp0] U = d_out^T . x (merged-top networks: dWout is rebuilt from U and T, see PostReduce)
  bool merged;       // this network's forward ran merged (fused_can_merge): h_nb was never stored
  bool post_zeroed;      // the caller's row builder zeroed post_counter in this call
  double* post_counter;  // 8 zeroed bytes: arrival counter of post_reduce_kernel (zeroed by the row builder, left zero)
  size_t slab_floats, part_floats;
};

constexpr int REDUCE_BLOCKS = 256;
constexpr int RED_CNT = 192;  // tile counters of the folded slab reduction: with post_counter 97 doubles, one row-builder zero array

static bool lowrank_top(const dppo_net_desc& d, int64_t M);
// the one-block backward kernel writes no dh_nb tensor: it needs the low-rank dW2
template <class P>
static bool bwd_one(const dppo_net_desc& d, int64_t M) {
  return fused_bwd_one_block<P>(d) && d.out_dim <= 128 && lowrank_top(d, M);
}

static int g_dbg = 0;  // tuning knob 8: timing experiments on the fused kernels (results are wrong while it is set)
template <class P>
static void carve_mlp(Carver& c, const dppo_net_desc& d, int64_t M, bool keep, bool bwd, MlpBufs<P>& B) {
  const size_t ES = P::ESIZE;
  const int H = d.hidden, nb = d.n_blocks;
  const int Kp0 = row_kp0(d), Kpo = round_up(d.out_dim, 64);
  memset(&B, 0, sizeof(B));
  B.in = c.take((size_t)M * Kp0 * ES);
  const int nh = keep ? nb + 1 : (nb > 0 ? 2 : 1);
  float* hbuf[MAX_BLOCKS + 1];
  for (int i = 0; i < nh; ++i) hbuf[i] = (float*)c.take((size_t)M * H * 4);
  for (int b = 0; b <= nb; ++b) B.h[b] = keep ? hbuf[b] : hbuf[b & 1];
  const size_t Mp = (size_t)round_up((int)M, 128);  // (fragment mode writes whole tiles of up to 128 rows)
  void* a1s = nullptr;
  void* a2s = nullptr;
  for (int b = 0; b < nb; ++b) {
    if (keep || b == 0) {
      a1s = c.take(Mp * H * ES);
      a2s = c.take(Mp * H * ES);
    }
    B.a1[b] = a1s;
    B.a2[b] = a2s;
    B.z1[b] = keep ? c.take((size_t)M * H * ES) : nullptr;
  }
  if (d.cond_hidden > 0) {
    const int Kpc = round_up(d.cond_dim, 64), C1p = round_up(d.cond_hidden, 64), Ep = round_up(d.cond_out, 64);
    B.cin = c.take((size_t)M * Kpc * ES);
    B.ca = c.take((size_t)M * C1p * ES);
    B.cz = keep ? c.take((size_t)M * C1p * ES) : nullptr;
    if (bwd) {
      B.d_enc = c.take((size_t)M * Ep * ES);
      B.d_cz = c.take((size_t)M * C1p * ES);
    }
  }
  if (d.use_layernorm && keep) B.ln_stats = (float*)c.take((size_t)nb * 2 * M * 2 * 4);
  B.hE = c.take((size_t)M * H * ES);
  for (int b = 0; b <= nb; ++b) B.hpre[b] = b == nb ? B.hE : (keep ? c.take((size_t)M * H * ES) : nullptr);
  B.ldout = round_up(d.out_dim, 16);
  B.out = (float*)c.take((size_t)M * B.ldout * 4);
  if (bwd) {
    B.d_out = c.take((size_t)M * Kpo * ES);
    B.dh = c.take((size_t)M * H * ES);
    B.dz1 = c.take(Mp * H * ES);
    B.dh_all[nb] = B.dh;
    for (int b = 0; b < nb; ++b) {
      B.dh_all[b] = c.take(Mp * H * ES);
      B.dz1_all[b] = b == 0 ? B.dz1 : c.take((size_t)M * H * ES);
    }
    B.doutf = (u32x4*)c.take(Mp * 64 * 2);  // up to four feature tiles of d_out
    B.xf = (u32x4*)c.take(Mp * Kp0 * 2);
    const int mt = d.plain || d.out_dim > 128 ? 0 : fused_rows_per_tile<P>(d, bwd_one<P>(d, M));
    B.tiles = mt > 0 ? (int)((M + mt - 1) / mt) : 0;
    B.tile_colsum = (float*)c.take((size_t)(2 * nb + 2 + (d.use_layernorm ? 4 * nb : 0)) * (B.tiles > 0 ? B.tiles : 1) * H * 4);
    const int tdp = round_up(d.time_dim > 0 ? d.time_dim : 1, 16);
    B.dtemb = d.kind == 0 ? (float*)c.take((size_t)M * tdp * 4) : nullptr;
    // largest slab: H x max(H, Kp0) with up to 64 splits of a <=4-tile output, or 16+ splits of H x H
    const size_t tiles_hh = (size_t)((H + 127) / 128) * ((H + 127) / 128);
    size_t splits_hh = (512 + tiles_hh - 1) / tiles_hh;
    const size_t alt = (size_t)128 * H * (size_t)(Kp0 > 128 ? Kp0 : 128);  // thin outputs, up to 128 splits
    // a pool for all of one backward's GEMMs (2 per block + first and out layer), reduced together at the end
    B.slab_floats = (size_t)2 * (nb > 0 ? nb : 1) * splits_hh * (size_t)H * H + 2 * alt;
    B.slab = (float*)c.take(B.slab_floats * 4);
    B.slab_used = 0, B.slab_jobs.n = 0, B.tn_group.n = 0, B.n_join = 0;
    B.part_floats = (size_t)REDUCE_BLOCKS * (H > 1024 ? H : 1024);
    B.part = (float*)c.take(B.part_floats * 4);
    B.lowrank = (float*)c.take((size_t)round_up(d.out_dim, 16) * H * 4);
    B.lowrank_u = (float*)c.take((size_t)round_up(d.out_dim, 16) * Kp0 * 4);
    B.post_counter = (double*)c.take(8 + RED_CNT * 4);
    B.red_cnt = (unsigned*)(B.post_counter + 1);
    B.post_zeroed = false;
    B.w0T = c.take((size_t)round_up(d.cond_dim > 0 ? d.cond_dim : 1, 16) * H * ES);
    B.dobs = (float*)c.take((size_t)M * round_up(d.cond_dim > 0 ? d.cond_dim : 1, 16) * 4);
  }
}

static void fill_bias_off(const dppo_net_desc& d, const ParamLayout& pl, int* off) {
  off[0] = (int)pl.b0;
  for (int b = 0; b < d.n_blocks; ++b) off[1 + 2 * b] = (int)pl.l1b[b], off[2 + 2 * b] = (int)pl.l2b[b];
  off[1 + 2 * d.n_blocks] = (int)pl.bout;
}
static void fill_ln_off(const dppo_net_desc& d, const ParamLayout& pl, int* off) {
  for (int b = 0; b < d.n_blocks; ++b)
    off[4 * b] = (int)pl.n1w[b], off[4 * b + 1] = (int)pl.n1b[b], off[4 * b + 2] = (int)pl.n2w[b], off[4 * b + 3] = (int)pl.n2b[b];
}

// cond_mlp (mlp_diffusion.py:201-207,240-241): enc = Linear2(act(Linear1(obs))), written as elem straight into the
// encoder columns [act_flat + time_dim, +cond_out) of the trunk's input rows `in` (ld Kp0).  cin: [M][Kpc] elem obs.
template <class P>
static void cond_encode(const dppo_net_desc& d, const float* prm, const char* pk, const PackLayout& L, int64_t M,
                        const void* cin, MlpBufs<P>& B, void* in, float* enc_f32, int ld_enc, bool keep, hipStream_t s) {
  const ParamLayout pl = param_layout(d);
  GemmNT g;
  memset(&g, 0, sizeof(g));
  g.M = (int)M, g.N = d.cond_hidden, g.Kp = L.Kpc, g.ldx = L.Kpc, g.ldw = L.Kpc, g.ldo = L.C1p, g.act = d.act;
  g.X = cin, g.W = pk + L.Wc1, g.bias = prm + pl.c1b, g.out_act = B.ca, g.out_pre = keep ? B.cz : nullptr;
  launch_gemm_nt<P>(g, s);
  if (L.C1p > round_up(d.cond_hidden, 16)) launch_zero_cols<P>(B.ca, (int)M, round_up(d.cond_hidden, 16), L.C1p, L.C1p, s);
  memset(&g, 0, sizeof(g));
  g.M = (int)M, g.N = d.cond_out, g.Kp = L.C1p, g.ldx = L.C1p, g.ldw = L.C1p;
  g.X = B.ca, g.W = pk + L.Wc2, g.bias = prm + pl.c2b;
  if (in != nullptr) g.out_pre = (char*)in + (size_t)(d.act_flat + d.time_dim) * P::ESIZE, g.ldo = L.Kp0;
  if (enc_f32 != nullptr) g.out_f32 = enc_f32, g.ldo32 = ld_enc;
  launch_gemm_nt<P>(g, s);
}

template <class P>
static void mlp_forward(const dppo_net_desc& d, const float* prm, const char* pk, const PackLayout& L, int64_t M,
                        MlpBufs<P>& B, bool keep, hipStream_t s, const LossArgs* fuse_loss = nullptr) {
  // fuse_loss: the policy half of the PPO loss in the forward kernel's epilogue (fuse_loss_ok() held: the merged fused kernel runs)
  const ParamLayout pl = param_layout(d);
  const int H = d.hidden, nb = d.n_blocks;
  if (fused_ok<P>(d)) {
    FusedFwdArgs f;
    memset(&f, 0, sizeof(f));
    f.wstream = (const u32x4*)(pk + L.sstream), f.ostream = (const u32x4*)(pk + L.ostream), f.params = prm;
    fill_bias_off(d, pl, f.bias_off);
    f.in = B.in, f.ld_in = L.Kp0, f.M = (int)M, f.Kp0 = sampler_geom<P>(d).Kp0, f.nb = nb, f.act = d.act;
    f.out_dim = d.out_dim, f.out = B.out, f.ldout = B.ldout, f.in_valid = d.in_dim;
    f.use_ln = d.use_layernorm;
    if (d.use_layernorm) fill_ln_off(d, pl, f.ln_off);
    if (keep) {
      for (int b = 0; b < nb; ++b) {
        f.a1[b] = B.a1[b], f.a2[b] = B.a2[b];
        // derivative sources: Mish' itself, ReLU sign words, or (LayerNorm) the pre-LayerNorm tensors
        f.z1[b] = B.z1[b], f.hpre[b] = B.hpre[b];
      }
      f.hpre[nb] = B.hE;
      f.ln_stats = d.use_layernorm ? B.ln_stats : nullptr;
    }
    B.merged = fused_can_merge<P>(d);
    if (B.merged) {  // the block's second layer folded into the out layer: h_nb is never formed (fused_forward_merged_kernel)
      f.merge_top = 1, f.ks0v = (d.in_dim + P::KB - 1) / P::KB, f.hpre[nb] = nullptr;
      if (g_dbg & 64) f.hpre[0] = nullptr;   // timing experiments: the forward does not store act'(h_0) ...
      if (g_dbg & 128) f.a2[0] = nullptr;    // ... / act(z1)
      if (g_dbg & 256) f.z1[0] = nullptr;    // ... / act'(z1)
      f.ostream0 = (const u32x4*)(pk + L.ostream0), f.ostream2 = (const u32x4*)(pk + L.ostream2);
      f.cbias2 = (const float*)(pk + L.cbias2);
    }
    B.frag = keep && B.allow_frag && B.merged;
    if (B.frag) {  // act(h_0), act(z1) as K-major fragments, in place of the row-major tensors (frag_ok())
      const int mt = fused_rows_per_tile<P>(d, true);
      B.mpad = (M + mt - 1) / mt * mt;
      f.a1f = (u32x4*)B.a1[0], f.a2f = (u32x4*)B.a2[0];
    }
    if (fuse_loss != nullptr && (!B.merged || B.frag)) {
      g_fused_fault = -8;  // (fuse_loss_ok() and this function disagree)
      return;
    }
    g_fused_fault = launch_fused_forward<P>(d, f, s, fuse_loss);
    return;
  }
  if (fuse_loss != nullptr) {
    g_fused_fault = -8;
    return;
  }
  if (d.plain) {  // x -> act(W0 x) -> act(W_b .) ... -> Wout .   (pre-activations kept for the backward: hpre[0], z1[b])
    GemmNT q;
    memset(&q, 0, sizeof(q));
    q.M = (int)M, q.X = B.in, q.ldx = L.Kp0, q.W = pk + L.W0, q.ldw = L.Kp0, q.Kp = L.Kp0, q.N = H, q.bias = prm + pl.b0;
    q.ldo = H, q.act = d.act, q.out_act = B.a1[0], q.out_pre = keep ? B.hpre[0] : nullptr;
    launch_gemm_nt<P>(q, s);
    for (int b = 0; b < nb; ++b) {
      memset(&q, 0, sizeof(q));
      q.M = (int)M, q.N = H, q.Kp = H, q.ldx = H, q.ldw = H, q.ldo = H, q.act = d.act;
      q.X = b == 0 ? B.a1[0] : B.a2[b - 1], q.W = pk + L.W1[b], q.bias = prm + pl.l1b[b];
      q.out_pre = keep ? B.z1[b] : nullptr, q.out_act = B.a2[b];
      launch_gemm_nt<P>(q, s);
    }
    memset(&q, 0, sizeof(q));
    q.M = (int)M, q.N = d.out_dim, q.Kp = H, q.ldx = H, q.ldw = H;
    q.X = B.a2[nb - 1], q.W = pk + L.Wout, q.bias = prm + pl.bout, q.out_f32 = B.out, q.ldo32 = B.ldout;
    launch_gemm_nt<P>(q, s);
    return;
  }
  GemmNT g;
  memset(&g, 0, sizeof(g));
  g.M = (int)M;
  // layer 0
  g.X = B.in, g.ldx = L.Kp0, g.W = pk + L.W0, g.ldw = L.Kp0, g.Kp = L.Kp0, g.N = H, g.bias = prm + pl.b0;
  g.out_f32 = B.h[0], g.ldo32 = H, g.ldo = H, g.act = d.act;
  g.out_act = nb > 0 ? B.a1[0] : nullptr;
  g.out_pre = nb > 0 ? nullptr : B.hE;
  launch_gemm_nt<P>(g, s);
  for (int b = 0; b < nb; ++b) {
    memset(&g, 0, sizeof(g));
    g.M = (int)M, g.N = H, g.Kp = H, g.ldx = H, g.ldw = H, g.ldo = H, g.ldo32 = H, g.act = d.act;
    g.X = B.a1[b], g.W = pk + L.W1[b], g.bias = prm + pl.l1b[b];
    g.out_pre = keep ? B.z1[b] : nullptr, g.out_act = B.a2[b];
    launch_gemm_nt<P>(g, s);
    g.X = B.a2[b], g.W = pk + L.W2[b], g.bias = prm + pl.l2b[b];
    g.res = B.h[b], g.ldres = H;
    g.out_f32 = B.h[b + 1];
    g.out_act = b + 1 < nb ? B.a1[b + 1] : nullptr;
    g.out_pre = b + 1 < nb ? nullptr : B.hE;
    launch_gemm_nt<P>(g, s);
  }
  memset(&g, 0, sizeof(g));
  g.M = (int)M, g.N = d.out_dim, g.Kp = H, g.ldx = H, g.ldw = H;
  g.X = B.hE, g.W = pk + L.Wout, g.bias = prm + pl.bout, g.out_f32 = B.out, g.ldo32 = B.ldout;
  launch_gemm_nt<P>(g, s);
}

// ---- side streams ----------------------------------------------------------------------------------------
// Independent parts of one call run on side streams so that their kernels fill each other's gaps (the persistent
// row-tile kernels leave CUs idle in their last round of tiles; the small reductions are latency-bound).  Fork / join
// through events: legal under stream capture too.  Tuning knob 2 turns it off (one stream, for per-kernel timing).
//   side 0: the critic pipeline of a PPO update;  side 1: the bias / time-embedding gradient tail of the actor
static int g_overlap = 1;
static int g_gate_critic = 0;  // tuning knob 10: measured 77 -> 74 M samples/s when on (letting the critic run ahead alone is better)
static int g_side_low_priority = 0;  // read when a side stream is first created
struct SideStream {
  hipStream_t s = nullptr;
  hipEvent_t fork = nullptr, join = nullptr, gate = nullptr;
  bool ok = false;
};
static SideStream* side_stream(int idx) {
  static SideStream tab[16][3];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  SideStream& t = tab[dev][idx];
  if (!t.ok) {
    // tuning knob 9 = 1 creates the side streams at the lowest priority, so that the caller's stream (the actor pipeline,
    // the longer one) would never wait for CUs.  Measured: 77 -> 46 M samples/s -- a low-priority queue is starved far
    // beyond the intent, so the default is equal priority and first come, first served.
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    if (g_side_low_priority) {
      if (hipStreamCreateWithPriority(&t.s, hipStreamNonBlocking, least) != hipSuccess) return nullptr;
    } else if (hipStreamCreateWithFlags(&t.s, hipStreamNonBlocking) != hipSuccess) {
      return nullptr;
    }
    if (hipEventCreateWithFlags(&t.fork, hipEventDisableTiming) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&t.join, hipEventDisableTiming) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&t.gate, hipEventDisableTiming) != hipSuccess) return nullptr;
    t.ok = true;
  }
  return &t;
}
static hipStream_t fork_side(hipStream_t main, int idx = 0) {  // returns the stream the independent part should use
  // g_overlap: 0 = none, 1 = every side stream, other values = bit mask over the side-stream indices, shifted by one
  // (e.g. 2 = the critic pipeline's only)
  const int mask = g_overlap == 1 ? 7 : g_overlap >> 1;
  SideStream* t = ((mask >> idx) & 1) ? side_stream(idx) : nullptr;
  if (t == nullptr) return main;
  (void)hipEventRecord(t->fork, main);
  (void)hipStreamWaitEvent(t->s, t->fork, 0);
  return t->s;
}
// the side stream's next kernel does not start before everything enqueued on `main` so far has finished
static void gate_side(hipStream_t main, hipStream_t sidestream, int idx = 0) {
  if (sidestream == main) return;
  SideStream* t = side_stream(idx);
  (void)hipEventRecord(t->gate, main);
  (void)hipStreamWaitEvent(sidestream, t->gate, 0);
}
static void join_side(hipStream_t main, hipStream_t sidestream, int idx = 0) {
  if (sidestream == main) return;
  SideStream* t = side_stream(idx);
  (void)hipEventRecord(t->join, sidestream);
  (void)hipStreamWaitEvent(main, t->join, 0);
}

static int g_post_one = 1;         // tuning knob 18: low-rank dW2 + time-embedding gradient in one launch after the slab reduce
static int g_merge_top = 1;        // tuning knob 17: sampler merges the top block's second layer into the out layer
static int g_lowrank_top = 1;      // tuning knob 16: top block's dW2 from the rank-out_dim factorisation (no H x H GEMM, no dh store)
static int g_early_join = 1;       // tuning knob 14: side streams joined right behind the weight-gradient GEMM launch
static int g_tn_group = 1;         // tuning knob 12: one launch for all weight-gradient GEMMs of a backward pass
static int g_tn_target = 256;      // tuning knob 3: workgroups a weight-gradient GEMM aims for (tiles x row splits)
static int g_tn_max_splits = 128;  // tuning knob 4: cap on its row splits (each split costs one fp32 slab of the output)
// (the out_dim-deep product costs H^2 out_dim scalar MACs against the M H^2 MFMA MACs it saves: on for M >= 100 out_dim;
// measured a loss at out_dim = 112, M = 10,000 -- ratio 89 -- and wins at out_dim 56, M = 7,500 -- ratio 134, BASELINE
// configs[2], where it also admits the one-block backward; tuning knob 30 moves the ratio)
static int g_lowrank_ratio = 100;
static bool lowrank_top(const dppo_net_desc& d, int64_t M) {
  return g_lowrank_top && d.n_blocks >= 1 && M >= (int64_t)g_lowrank_ratio * d.out_dim;
}
// tuning knob 31: K-major fragment operands for the weight-gradient GEMMs of one-block bf16 networks.  OFF by default: parity
// green (tests/test_hip_parity.py, knob 31 cases) but not faster yet -- 114-118 us for the step's two launches against 107 with
// the transposing kernel (DESIGN.md section 13 has the knock-out measurements: 43 us skeleton + 52 us HBM stream + 23 us ring,
// adding up instead of overlapping)
static int g_frag = 0;
// tuning knob 35: slab reductions folded into the grouped weight-gradient GEMM launch (no tail_reduce launch).  OFF by default:
// correct (160 parity tests with it on) but slower -- the launch goes from 107 to 200 us per step (both networks): a tile's LAST
// workgroup sums splits x 64 KB alone (1 MB for an H x H tile, 16 such workgroups at the very end of the launch, each at one
// CU's ~25 GB/s), where tail_reduce_kernel puts the whole chip on the same 22 MB for 17 us.  (With release / acquire fences
// instead of sc1 stores it was 589 us: buffer_wbl2 flushes the XCD's whole L2 once per wave.)
static int g_fold = 0;
static int g_mom_rider = 0;  // tuning knob 36: the minibatch's advantage moments as riders of the row builder (1: always, 2: never) or a launch
                              // of their own.  0 (default): riders for minibatches of at most MOM_RIDER_MAX_N samples -- at N = 50,000 the
                              // launch already hides under the critic's forward on the side stream (the riders save 3.5 us only in the
                              // serial order, profiles/r03_moments_rider_ab.txt); at BASELINE configs[2]'s 7,500 the step is a chain of
                              // launches and one fewer is 0.186 -> 0.180 ms
constexpr int64_t MOM_RIDER_MAX_N = 16384;
// Time-embedding gradient through the first layer's weight-gradient GEMM: with a one-hot of the row's denoising step k in
// the K padding of the input rows, dW0's extra columns are S[h][k] = sum over the rows of step k of dh0[row][h], and
// d loss / d temb[k] = W0[:, temb columns]^T S[:, k] -- no second pass over dh0, no segmented reduction (tuning knob 11).
static int g_temb_onehot = 1;
template <class P>
static int temb_onehot_col(const dppo_net_desc& d, const PackLayout& L, int Kft, const MlpBufs<P>& B) {
  if (!g_temb_onehot || d.kind != 0 || !fused_ok<P>(d) || Kft < 1) return -1;
  if (d.cond_out > 0 && d.cond_out % 16) return -1;  // the encoder's epilogue zero-fills up to a multiple of 16 columns
  if (L.Kp0 - d.in_dim < Kft || (size_t)(d.hidden + d.time_dim) * Kft > B.part_floats) return -1;
  return d.in_dim;
}
// May this network's whole training pass run in fragment mode?  Everything the forward, the backward and the GEMM group
// will decide later must already hold: merged forward, one-block backward (hence the low-rank dW2), grouped GEMM launch, the
// time-embedding gradient through the one-hot columns (the row-major dh_0 has no other reader then), no cond_mlp encoder
// (its backward reads dh_0 row-major), the caller wants no d loss / d observation, and the backward's LDS holds the extras.
template <class P>
static bool frag_ok(const dppo_net_desc& d, int64_t M, const PackLayout& L, int Kft, const MlpBufs<P>& B, bool wants_dobs) {
  if (!g_frag || P::ESIZE != 2 || wants_dobs || d.cond_hidden > 0 || !fused_frag_shape(d) || !g_tn_group) return false;
  if (!fused_ok<P>(d) || B.tiles <= 0 || !fused_can_merge<P>(d) || !bwd_one<P>(d, M)) return false;
  if (d.kind == 0 && temb_onehot_col<P>(d, L, Kft, B) < 0) return false;
  const int mt = fused_rows_per_tile<P>(d, true);
  const FusedGeom fg = fused_geom<P>(d);
  if (L.Kp0 % 16 || L.Kp0 > 128) return false;
  return (size_t)mt * ((size_t)fg.KpB0 * 2 + (size_t)L.Kp0 * 2) + (size_t)SAMPLER_WAVES * 32 * 32 * (d.hidden / 128) <=
         (size_t)mt * d.hidden * 2;
}
// In-kernel first-layer weight gradient (tuning knob 37; fused.h, FusedBwdArgs::dw0_slab).  dh_0 -- a third of the bytes the
// weight-gradient GEMMs read, for a 512 x 64 output -- is then neither stored nor read back: the one-block backward multiplies each
// tile of it with the tile's input rows while both are on chip and keeps the partial product in LDS.  The 64 KB that has room for
// is [H][32]: the network's informative input columns must fit 32.  A denoiser's rows are [x_k | temb(t_k) | obs | one-hot(k)]:
// the temb columns are a function of k alone (their gradient is rebuilt from the one-hot sums, PostReduce::dW0t; the kernel
// skips them when it reads the row tile: act_flat and time_dim must be multiples of 4), and of the Kft one-hot columns the last
// may be left out (every row carries exactly one: PostReduce::S_rest, with the bias gradient taken of the rounded dh_0) --
// hopper: 12 + 11 + 9 = 32.
// Needs the one-block backward on its compact walk, the one-hot time columns, post_reduce in one launch, nobody asking for
// d loss / d observation, no cond_mlp, and a slab per workgroup in the pool.
static int g_dw0 = 1;
// Tuning knob 38: with the in-kernel dW0 everything the time-embedding gradient needs -- the one-hot sums, the first layer's bias
// gradient, then G = W0_temb^T . S and the time MLP's backward (a chain of dependent steps: 11 of post_reduce's 16 us) -- comes out
// of the backward KERNEL, not out of the weight-gradient GEMMs: that reduction and that part of post_reduce run on a side stream
// under the GEMM launch, with the bias sums and the loss statistics, and only the GEMMs' own slabs, the low-rank dW2 and dWout
// stay behind the GEMMs on the caller's stream.
static int g_side_tail = 1;
// Tuning knob 40: ... and instead of a side stream (whose fork is an event record between the backward kernel and the GEMM launch
// on the caller's stream: ~8 us of idle time there, profiles/r03_dw0_ab.txt) the same work RIDES in the actor's GEMM launch as extra
// workgroups in front of the GEMM tiles (gemm.h, GemmTNExtra): producers (slab reductions, bias sums, loss statistics) first, then
// the time-embedding blocks, which wait for the producers on an arrival counter (post_blocks.h).  No stream, no event, no launch.
// OFF by default: parity green on the first run, but the actor's GEMM launch goes from 68 to 94 us (serial) and the step from 0.350
// to 0.381 ms -- the riders' dependent chain (reduce, G, the time MLP's backward: latency-bound steps) runs under the GEMM's memory
// load, every step of it several times slower than alone, and the launch cannot end before it does; on its own stream the same chain
// is as slow but nobody waits for it (profiles/r03_dw0_ab.txt).
static int g_tail_riders = 0;
// Tuning knob 41: with knob 38, what is left behind the actor's GEMMs -- their slab reductions, then the low-rank dW2 / dWout / db2,
// which read only the two THIN products -- is ONE launch (tail_post_kernel: the thin products' reduce blocks first, the dependent
// blocks poll their arrival, the big slabs are reduced beside them) instead of two.
static int g_tail_post = 1;
// Tuning knob 39: the policy half of the PPO loss in the epilogue of the actor's fused forward (loss_dev.h; fused.hip, LOSSF): no
// loss launch between the actor's forward and backward, the forward's eps tile never goes to HBM.  bf16 one-block actor on the
// merged forward with 64-row tiles (hidden 512), a head of at most 16 outputs whose reward-horizon part is a multiple of 4 wide,
// the two pipelines on two streams (the value half stays a launch on the critic's), the row builder's loss table, no riders.
// OFF by default: parity green (ratio == 1 bit for bit, tests/test_hip_parity.py knob 39), step time EQUAL to the separate launch's.
// At first it was 16 us slower: the variant allocated 235 VGPRs where the plain forward has 223, two waves of it left a SIMD 32 free
// registers instead of 64, and the critic's value-loss launch could no longer slip a wave in beside the actor's persistent
// workgroups (13 -> 80 us: its whole pipeline started that much later).  Capped at 224 (`amdgpu_num_vgpr(112)`: hipcc doubles the
// request on the unified register file; eight cold values go to scratch) that is gone -- and what remains is a wash: the forward's
// epilogue (one wave walks the tile's 64 samples, seven wait) costs what the launch cost (profiles/r03_fused_loss_ab.txt).
static int g_fuse_loss = 0;
template <class P>
static bool fuse_loss_ok(const dppo_net_desc& d, const LossArgs& la, const MlpBufs<P>& B, bool two_streams, bool mom_rider) {
  if (!g_fuse_loss || P::ESIZE != 2 || !two_streams || mom_rider || B.allow_frag || !fused_ok<P>(d) || !fused_loss_shape(d)) return false;
  const dppo_ppo_cfg& pc = la.pcfg;
  const int rh = pc.reward_horizon < pc.horizon_steps ? pc.reward_horizon : pc.horizon_steps, cnt = rh * pc.action_dim;
  if (la.tab == nullptr || pc.ft_denoising_steps > 64 || cnt > 16 || cnt <= 0) return false;
  return ((la.AF | cnt) & 3) == 0 && la.AF <= 16 && la.ldde % 8 == 0 && la.ldde >= 16;
}
static int dw0_cols(const dppo_net_desc& d) { return d.kind == 0 ? d.act_flat + d.cond_dim : d.in_dim; }
static int dw0_nhot(const dppo_net_desc& d, int Kft) {  // one-hot columns that fit behind the data columns
  const int room = 32 - dw0_cols(d);
  return d.kind != 0 ? 0 : (room >= Kft ? Kft : room);
}
template <class P>
static bool dw0_ok(const dppo_net_desc& d, int64_t M, const PackLayout& L, int Kft, const MlpBufs<P>& B, bool wants_dobs) {
  if (!g_dw0 || P::ESIZE != 2 || wants_dobs || d.cond_hidden > 0 || !g_tn_group || !g_post_one) return false;
  if (!fused_ok<P>(d) || B.tiles <= 0 || !bwd_one<P>(d, M) || !fused_dw0_shape(d) || L.Kp0 < 64) return false;
  if (d.kind == 0) {
    if (temb_onehot_col<P>(d, L, Kft, B) < 0 || !B.post_zeroed || d.in_dim != d.act_flat + d.time_dim + d.cond_dim) return false;
    if (d.act_flat % 4 || d.time_dim % 4 || 32 + d.time_dim > 64) return false;
    if (dw0_nhot(d, Kft) < Kft - 1) return false;
  } else if (d.in_dim > 32) {
    return false;
  }
  return (size_t)fused_bwd_one_grid<P>(d, M) * d.hidden * 32 <= B.slab_floats / 2;
}
template <class P>
static void flush_slabs(MlpBufs<P>& B, hipStream_t s, const SlotOuts* slots = nullptr, int slot_width = 0,
                        const LossArgs* fin = nullptr, bool defer_reduce = false) {
  // defer_reduce: launch the GEMMs (and join the side streams) but leave the slab jobs to the caller (tail_post_kernel)
  // slots / fin: the fused backward's per-tile column sums and the loss statistics ride in the reduction launch (see
  // tail_reduce_kernel) -- or, with the folded tail, in the GEMM launch itself, and there is no reduction launch
  bool folded = false;
  if (B.tn_group.n > 0) {
    GemmTNGroup& gr = B.tn_group;
    for (int i = 1; i < gr.n; ++i)  // longest row ranges first (insertion sort: the short jobs fill the last round)
      for (int k = i; k > 0 && gr.j[k].rows_per_split > gr.j[k - 1].rows_per_split; --k) {
        const GemmTN t = gr.j[k];
        gr.j[k] = gr.j[k - 1], gr.j[k - 1] = t;
      }
    gr.base[0] = 0;
    int tiles_total = 0;
    for (int i = 0; i < gr.n; ++i) {
      const int tiles = ((gr.j[i].N1 + 127) / 128) * ((gr.j[i].N2 + 127) / 128);
      gr.base[i + 1] = gr.base[i] + gr.j[i].splits * tiles;
      tiles_total += tiles;
    }
    memset(&gr.ex, 0, sizeof(gr.ex));
    // Folded tail: possible when this flush reduces exactly this group's slabs (no GEMM was launched outside the group since
    // the last flush) and the caller's row builder zeroed the tile counters
    int n_group_slabs = 0;
    for (int i = 0; i < gr.n; ++i) n_group_slabs += gr.j[i].red_n2a >= 0 ? 2 : 1;
    folded = g_fold && B.fold && (slots != nullptr || fin != nullptr) && tiles_total <= RED_CNT &&
             n_group_slabs == B.slab_jobs.n && !(g_dbg & 28) && (slots == nullptr || slots->n_slots <= TN_MAX_SLOTS);
    if (folded) {
      int cnt0 = 0;
      for (int i = 0; i < gr.n; ++i) {
        gr.j[i].red_cnt = B.red_cnt + cnt0;
        cnt0 += ((gr.j[i].N1 + 127) / 128) * ((gr.j[i].N2 + 127) / 128);
      }
      GemmTNExtra& e = gr.ex;
      if (slots != nullptr && slots->n_slots > 0) {
        e.colsum = B.tile_colsum, e.tiles = B.tiles, e.width = slot_width, e.n_slots = slots->n_slots;
        e.slot_bx = (slot_width + 15) / 16, e.n_slot_blocks = e.n_slots * e.slot_bx;
        for (int i = 0; i < slots->n_slots; ++i) e.slot_out[i] = slots->out[i], e.slot_n[i] = slots->n[i];
      }
      if (fin != nullptr && fin->N > 0) {
        e.fin_partial = fin->partial, e.fin_blocks = loss_blocks(fin->N), e.fin_moments = fin->moments, e.fin_stats = fin->stats;
        e.fin_part = fin->part, e.fin_n_count = fin->n_count;
      }
      e.n_blocks = (e.n_slot_blocks + (e.fin_stats != nullptr ? 1 : 0) + 7) / 8 * 8;  // (keeps the GEMM tiles' block id = XCD map)
    } else {
      for (int i = 0; i < gr.n; ++i) gr.j[i].red_cnt = nullptr;
      if (B.ride) {  // the backward kernel's own reductions and the time-embedding gradient as riders (knob 40)
        GemmTNExtra& e = gr.ex;
        const TailReduce& t = B.ride_t;
        e.colsum = t.colsum, e.tiles = t.tiles, e.width = t.width, e.n_slots = t.slots.n_slots;
        e.slot_bx = (t.width + 15) / 16, e.n_slot_blocks = e.n_slots * e.slot_bx;
        for (int i = 0; i < e.n_slots; ++i) e.slot_out[i] = t.slots.out[i], e.slot_n[i] = t.slots.n[i];
        if (t.fin_stats != nullptr) {
          e.fin_partial = t.fin_partial, e.fin_blocks = t.fin_blocks, e.fin_moments = t.fin_moments, e.fin_stats = t.fin_stats;
          e.fin_part = t.fin_part, e.fin_n_count = t.fin_n_count;
        }
        int n_prod = e.n_slot_blocks + 1;
        e.n_rjobs = t.jobs.n;
        for (int i = 0; i < t.jobs.n; ++i) {
          e.rjob[i] = t.jobs.j[i];
          e.rjob_blocks[i] = (int)(((size_t)t.jobs.j[i].rows * t.jobs.j[i].cols + 15) / 16);  // (16 elements per block: slab_job_rider)
          n_prod += e.rjob_blocks[i];
        }
        e.post = B.ride_q;
        int n_cons = 0;
        if (e.post.G != nullptr) {
          e.arrive_cnt = B.red_cnt + RED_CNT - 1;  // (zeroed by the row builder with the post-reduce counter: B.post_zeroed)
          e.post.wait_cnt = e.arrive_cnt, e.post.wait_need = n_prod;
          e.post.n_temb = (e.post.Kft * e.post.td + 3) / 4, e.post.n_dw0t = (e.post.H * e.post.td + 255) / 256;
          e.post.n_lowrank = e.post.n_wout = 0;
          n_cons = e.post.n_temb + e.post.n_dw0t;
        }
        e.n_blocks = (n_prod + n_cons + 7) / 8 * 8;
      }
    }
    B.ride = false;
    launch_gemm_tn_group<P>(gr, s);
    gr.n = 0;
  }
  if (B.tnf_group.n > 0) {  // fragment-operand jobs (weight_grad_frag): same order rule, same single launch
    GemmTNFragGroup& gr = B.tnf_group;
    for (int i = 1; i < gr.n; ++i)
      for (int k = i; k > 0 && (int64_t)gr.j[k].ks_per_split * gr.j[k].tb > (int64_t)gr.j[k - 1].ks_per_split * gr.j[k - 1].tb; --k) {
        const GemmTNFrag t = gr.j[k];
        gr.j[k] = gr.j[k - 1], gr.j[k - 1] = t;
      }
    gr.base[0] = 0;
    for (int i = 0; i < gr.n; ++i) gr.base[i + 1] = gr.base[i] + gemm_tn_frag_blocks(gr.j[i]);
    launch_gemm_tn_frag_group(gr, B.mpad, s);
    gr.n = 0;
  }
  // One wait on the caller's stream however many side streams there are: the earlier ones are joined into the LAST one (their
  // waits cost nothing there) and only that one into s -- a satisfied event wait is still a barrier packet between the GEMM
  // launch and the reduction behind it, ~6 us each on the update's critical path (profiles/r03_dw0_ab.txt, knob 38).
  for (int i = 0; i + 1 < B.n_join; ++i) join_side(B.join_s[B.n_join - 1], B.join_s[i], B.join_idx[i]);
  if (B.n_join > 0) join_side(s, B.join_s[B.n_join - 1], B.join_idx[B.n_join - 1]);
  B.n_join = 0;
  if (defer_reduce && !folded) return;
  if (folded) {
    // everything a reduction launch would do has been done by the GEMM launch
  } else if (slots != nullptr || fin != nullptr) {
    TailReduce t;
    memset(&t, 0, sizeof(t));
    t.jobs = B.slab_jobs;
    if (slots != nullptr) t.colsum = B.tile_colsum, t.tiles = B.tiles, t.width = slot_width, t.slots = *slots;
    if (g_dbg & 4) t.jobs.n = 0;          // timing experiments (knob 8; results are wrong while set): no slab reductions,
    if (g_dbg & 8) t.slots.n_slots = 0;   // no bias sums,
    if (g_dbg & 16) fin = nullptr;        // no loss statistics
    launch_tail_reduce(t, fin, s);
  } else {
    launch_slab_reduce_batch(B.slab_jobs, s);
  }
  B.slab_jobs.n = 0, B.slab_used = 0;
}
// gw[N1][N2] (ld ldgw) = A[M][N1]^T . B[M][N2].  A thin N1 (the out layer) is computed transposed, B^T . A, so that the
// 512 x 64 block shape of gemm_tn covers it in one output tile; the slab reduce transposes back.
template <class P>
static void weight_grad(const void* A, int lda, int N1, const void* Bm, int ldb, int N2, int64_t M, MlpBufs<P>& B,
                        float* gw, int ldgw, hipStream_t s, bool defer = false, int n2a = -1, float* gw2 = nullptr,
                        int ldgw2 = 0) {  // n2a >= 0: columns [n2a, N2) of the result go to gw2 instead (never with a thin N1)
  const bool swap = N1 <= 64 && N2 > 64;
  if (swap) {
    const void* tp = A;
    A = Bm, Bm = tp;
    int ti = lda;
    lda = ldb, ldb = ti;
    ti = N1, N1 = N2, N2 = ti;
  }
  const bool group = defer && g_tn_group;  // launched with the caller's other GEMMs, 128 x 128 tiles throughout
  const bool thin = !group && gemm_tn_thin(N1, N2);
  const size_t N1s = (size_t)round_up(N1, 4);  // (the folded reduction's transposed slabs pad N1 to whole 16-byte pieces)
  const size_t tiles = thin ? (size_t)((N1 + 511) / 512) : (size_t)((N1 + 127) / 128) * ((N2 + 127) / 128);
  int64_t splits = (g_tn_target + tiles - 1) / tiles;
  const int64_t max_splits = (M + 63) / 64;
  if (splits > max_splits) splits = max_splits;
  if (splits > g_tn_max_splits) splits = g_tn_max_splits;
  if ((size_t)splits * N1s * N2 > B.slab_floats - B.slab_used || B.slab_jobs.n + 2 > MAX_SLAB_JOBS ||
      B.tn_group.n >= MAX_TN_JOBS)
    flush_slabs(B, s);
  while (splits > 1 && (size_t)splits * N1s * N2 > B.slab_floats) --splits;
  if (splits >= 8) splits = splits / 8 * 8;  // a multiple of the XCD count keeps one split's tiles on one XCD
  int64_t rps = (M + splits - 1) / splits;
  rps = (rps + 63) / 64 * 64;
  const int64_t need = (M + rps - 1) / rps;
  if (need < splits) splits = need >= 8 ? (need + 7) / 8 * 8 : need;  // surplus splits see no rows and store zeros
  GemmTN t;
  memset(&t, 0, sizeof(t));
  t.A = A, t.B = Bm, t.M = (int)M, t.N1 = N1, t.N2 = N2, t.lda = lda, t.ldb = ldb;
  float* sub = B.slab + B.slab_used;
  B.slab_used += (size_t)splits * N1s * N2;
  t.slab = sub, t.ldc = N2, t.splits = (int)splits, t.rows_per_split = (int)rps;
  // (the folded reduction of flush_slabs(): where this product goes -- exactly what the SlabJob(s) below say)
  t.red_out = gw, t.red_ldo = ldgw, t.red_transpose = swap ? 1 : 0, t.red_n2a = (n2a >= 0 && !swap) ? n2a : -1;
  t.red_out2 = gw2, t.red_ldo2 = ldgw2;
  if (group)
    B.tn_group.j[B.tn_group.n++] = t;
  else
    launch_gemm_tn<P>(t, s);
  SlabJob& j = B.slab_jobs.j[B.slab_jobs.n++];
  j.slab = sub, j.out = gw, j.splits = (int)splits, j.rows = N1, j.cols = N2, j.lds = N2, j.ldo = ldgw, j.transpose = swap ? 1 : 0;
  j.c0 = 0, j.wide = 0;
  if (n2a >= 0 && !swap) {
    j.cols = n2a;
    SlabJob& j2 = B.slab_jobs.j[B.slab_jobs.n++];
    j2 = j;
    j2.out = gw2, j2.ldo = ldgw2, j2.c0 = n2a, j2.cols = N2 - n2a;
  }
  if (!defer) flush_slabs(B, s);  // deferred: the caller flushes once after its last GEMM (same stream)
}

// The same contraction from K-major fragment operands (gemm.h, GemmTNFrag): gw[N1][N2] = FA^T . FB over the batch rows,
// FA / FB with nta / ntb feature tiles per k-step; always deferred to the caller's flush.  transpose: gw receives the
// transposed result (the thin side of a thin product is passed as B, the wide one as A: a wave's tile is 64 x 16 tb).
template <class P>
static void weight_grad_frag(const u32x4* FA, int nta, int N1, const u32x4* FB, int ntb, int N2, MlpBufs<P>& B, float* gw,
                             int ldgw, hipStream_t s, bool transpose = false, int n2a = -1, float* gw2 = nullptr,
                             int ldgw2 = 0) {
  GemmTNFrag t;
  memset(&t, 0, sizeof(t));
  t.A = FA, t.B = FB, t.nta = nta, t.ntb = ntb, t.N1 = N1, t.N2 = N2;
  if (!gemm_tn_frag_prepare(t)) {  // (frag_ok() admits only shapes the kernel covers: H <= 512, Kp0 <= 128)
    g_fused_fault = -6;
    return;
  }
  const int nba = (N1 + 63) / 64, nbb = (N2 + 16 * t.tb - 1) / (16 * t.tb);
  const int64_t wg_tiles = (int64_t)((nba + t.wga - 1) / t.wga) * ((nbb + t.wgb - 1) / t.wgb);
  t.ks_total = (int)(B.mpad / 32);
  // workgroups a job aims for: the chip's CUs for the H x H class, half that for the thin ones (their waves carry a
  // quarter of the MFMAs per k-step and each split costs a slab)
  const int64_t target = (int64_t)N1 * N2 >= 128 * 128 ? g_tn_target : g_tn_target / 2;
  int64_t splits = (target + wg_tiles - 1) / wg_tiles;
  const int64_t max_splits = t.ks_total >= 2 ? t.ks_total / 2 : 1;
  if (splits > max_splits) splits = max_splits;
  if (splits > g_tn_max_splits) splits = g_tn_max_splits;
  if (splits < 1) splits = 1;
  if ((size_t)splits * N1 * N2 > B.slab_floats - B.slab_used || B.slab_jobs.n + 2 > MAX_SLAB_JOBS ||
      B.tnf_group.n >= MAX_TN_JOBS)
    flush_slabs(B, s);
  while (splits > 1 && (size_t)splits * N1 * N2 > B.slab_floats) --splits;
  if (splits >= 8) splits = splits / 8 * 8;  // a multiple of the XCD count keeps one split's tiles on one XCD
  int64_t kps = (t.ks_total + splits - 1) / splits;
  const int64_t need = (t.ks_total + kps - 1) / kps;
  if (need < splits) splits = need >= 8 ? (need + 7) / 8 * 8 : need;  // surplus splits see no k-steps and store zeros
  float* sub = B.slab + B.slab_used;
  B.slab_used += (size_t)splits * N1 * N2;
  t.slab = sub, t.ldc = N2, t.splits = (int)splits, t.ks_per_split = (int)kps;
  B.tnf_group.j[B.tnf_group.n++] = t;
  SlabJob& j = B.slab_jobs.j[B.slab_jobs.n++];
  j.slab = sub, j.out = gw, j.splits = (int)splits, j.rows = N1, j.cols = N2, j.lds = N2, j.ldo = ldgw, j.transpose = transpose ? 1 : 0;
  j.c0 = 0, j.wide = 0;
  if (n2a >= 0 && !transpose) {
    j.cols = n2a;
    SlabJob& j2 = B.slab_jobs.j[B.slab_jobs.n++];
    j2 = j;
    j2.out = gw2, j2.ldo = ldgw2, j2.c0 = n2a, j2.cols = N2 - n2a;
  }
}

// d temb = dh0 . W0[:, temb columns]; summed per fine-tuned step; back through the tiny time MLP
template <class P>
static void time_embedding_grad(const dppo_net_desc& d, const float* prm, const char* pk, const PackLayout& L, int64_t M,
                                MlpBufs<P>& B, const void* dh0, float* grad, const int32_t* krow, const dppo_step* ksteps,
                                int Kft, hipStream_t s) {
  const ParamLayout pl = param_layout(d);
  const int H = d.hidden, td = d.time_dim;
  GemmNT g;
  memset(&g, 0, sizeof(g));
  g.M = (int)M, g.N = td, g.Kp = H, g.ldx = H, g.ldw = H;
  g.X = dh0, g.W = pk + L.W0tT, g.out_f32 = B.dtemb, g.ldo32 = L.tdp;
  launch_gemm_nt<P>(g, s);
  int sblocks = (int)(B.part_floats / ((size_t)Kft * td)) - 1;  // partial[sblocks][Kft*td] + G[Kft*td] must fit
  sblocks = sblocks > REDUCE_BLOCKS ? REDUCE_BLOCKS : (sblocks < 1 ? 1 : sblocks);
  launch_temb_segsum(B.dtemb, L.tdp, krow, M, Kft, td, B.part, sblocks, s);
  float* G = B.part + (size_t)sblocks * Kft * td;
  launch_slab_reduce(B.part, sblocks, (size_t)Kft * td, G, 1.f, s);
  launch_time_backward(prm + pl.te1_w, prm + pl.te1_b, prm + pl.te2_w, G, ksteps, Kft, td, grad + pl.te1_w,
                       grad + pl.te1_b, grad + pl.te2_w, grad + pl.te2_b, s);
}

// gradients of the encoder from dh0 (= d loss / d h_0 of the trunk)
template <class P>
static void cond_backward(const dppo_net_desc& d, const float* prm, const char* pk, const PackLayout& L, int64_t M,
                          MlpBufs<P>& B, const void* dh0, const void* cin, float* grad, hipStream_t s) {
  const ParamLayout pl = param_layout(d);
  const int H = d.hidden;
  GemmNT g;
  memset(&g, 0, sizeof(g));  // d_enc = dh0 . W0[:, enc columns]
  g.M = (int)M, g.N = d.cond_out, g.Kp = H, g.ldx = H, g.ldw = H, g.ldo = L.Ep;
  g.X = dh0, g.W = pk + L.W0eT, g.out_pre = B.d_enc;
  launch_gemm_nt<P>(g, s);
  if (L.Ep > round_up(d.cond_out, 16))  // columns the epilogue does not store must be zero: they are the next GEMM's K padding
    launch_zero_cols<P>(B.d_enc, (int)M, round_up(d.cond_out, 16), L.Ep, L.Ep, s);
  weight_grad<P>(B.d_enc, L.Ep, d.cond_out, B.ca, L.C1p, d.cond_hidden, M, B, grad + pl.c2w, d.cond_hidden, s);
  launch_colsum<P>(B.d_enc, (int)M, d.cond_out, L.Ep, B.part, REDUCE_BLOCKS, grad + pl.c2b, 1.f, s);
  memset(&g, 0, sizeof(g));  // d_zc = (d_enc . Wc2) * act'(z_c)
  g.M = (int)M, g.N = d.cond_hidden, g.Kp = L.Ep, g.ldx = L.Ep, g.ldw = L.Ep, g.ldo = L.C1p;
  g.X = B.d_enc, g.W = pk + L.Wc2T, g.out_pre = B.d_cz, g.dsrc_kind = 2, g.dsrc_ld = L.C1p, g.dact = d.act;
  g.dsrc = d.act == DPPO_ACT_RELU ? B.ca : B.cz;
  launch_gemm_nt<P>(g, s);
  weight_grad<P>(B.d_cz, L.C1p, d.cond_hidden, cin, L.Kpc, d.cond_dim, M, B, grad + pl.c1w, d.cond_dim, s);
  launch_colsum<P>(B.d_cz, (int)M, d.cond_hidden, L.C1p, B.part, REDUCE_BLOCKS, grad + pl.c1b, 1.f, s);
}

// d_out (B.d_out, [M][Kpo] elem) -> gradients of every parameter of the network into `grad`
template <class P>
static void mlp_backward(const dppo_net_desc& d, const float* prm, const char* pk, const PackLayout& L, int64_t M,
                         MlpBufs<P>& B, float* grad, const int32_t* krow, const dppo_step* ksteps, int Kft,
                         hipStream_t s, bool bout_done = false, int aux_idx = 1, const LossArgs* fin = nullptr) {
  // fin: the loss's statistics are finalised off the critical path (on the tail stream, or after the GEMMs)
  const ParamLayout pl = param_layout(d);
  const int H = d.hidden, nb = d.n_blocks;
  if (fused_ok<P>(d) && B.tiles > 0) {
    // one kernel produces every data gradient (dh[nb..0], dz1[..]) and their per-tile column sums
    FusedBwdArgs f;
    memset(&f, 0, sizeof(f));
    const FusedGeom fg = fused_geom<P>(d);
    f.bstream = (const u32x4*)(pk + L.bstream), f.d_out = B.d_out, f.ld_dout = L.Kpo, f.M = (int)M, f.KpB0 = fg.KpB0;
    f.nb = nb, f.act = d.act, f.colsum = B.tile_colsum, f.out_valid = d.out_dim;
    f.dout_slot = bout_done ? -1 : 2 * nb + 1 + (d.use_layernorm ? 4 * nb : 0);
    const bool post = false;  // the forward left the derivative sources in z1 / hpre (see emit())
    f.params = prm, f.use_ln = d.use_layernorm, f.ln_stats = B.ln_stats;
    if (d.use_layernorm) fill_ln_off(d, pl, f.ln_off);
    for (int b = 0; b < nb; ++b) {
      f.m1[b] = post ? B.a2[b] : B.z1[b];
      f.m0[b] = post ? B.a1[b] : B.hpre[b];
      f.dz1[b] = B.dz1_all[b];
    }
    for (int b = 0; b <= nb; ++b) f.dh[b] = B.dh_all[b];
    const bool lowrank = lowrank_top(d, M);
    if (lowrank) f.dh[nb] = nullptr;  // only dW2 of the top block read it: see lowrank_dw_kernel
    const bool one = bwd_one<P>(d, M);  // (what carve_mlp sized the per-tile column sums for)
    f.one_block = one ? 1 : 0;
    if (g_dbg & 1)  // timing experiment: no gradient stores (the weight-gradient GEMMs then read stale buffers)
      for (int b = 0; b <= nb; ++b) f.dh[b] = nullptr, f.dz1[b < nb ? b : 0] = nullptr;
    if (g_dbg & 32) f.dh[0] = nullptr;  // timing experiment: dh_0 is not stored (bound of the in-kernel dW0)
    if (g_dbg & 2)  // timing experiment: no derivative-source fetch
      for (int b = 0; b < nb; ++b) f.m1[b] = f.m0[b] = nullptr;
    const bool frag = B.frag;
    if (frag) {  // (frag_ok() held when the forward ran: one-block kernel, one-hot time columns, ...)
      if (!one || !B.merged || !lowrank) {
        g_fused_fault = -5;
        return;
      }
      B.dof_nt = d.out_dim <= 16 ? 1 : (d.out_dim <= 32 ? 2 : 4);
      f.dz1f = (u32x4*)B.dz1_all[0], f.dh0f = (u32x4*)B.dh_all[0], f.doutf = B.doutf, f.dof_nt = B.dof_nt;
      f.x = B.in, f.ld_x = L.Kp0, f.xf = B.xf;
    }
    // in-kernel dW0 (dw0_ok() held when the rows were built): a slab per workgroup out of the pool, reduced with the GEMMs' slabs
    const bool dw0 = B.dw0 && one && !frag && !(g_dbg & 33);
    const int dw0_nh = dw0 ? dw0_nhot(d, Kft) : 0;
    float* dw0_slab = nullptr;
    int dw0_grid = 0;
    if (B.dw0 && !dw0) {
      g_fused_fault = -7;  // (the compact rows were built for a pass that cannot use them: dw0_ok() and this function disagree)
      return;
    }
    if (dw0) {
      dw0_grid = fused_bwd_one_grid<P>(d, M);
      if ((size_t)dw0_grid * H * 32 > B.slab_floats - B.slab_used) flush_slabs(B, s);
      dw0_slab = B.slab + B.slab_used;
      B.slab_used += (size_t)dw0_grid * H * 32;
      f.dh[0] = nullptr, f.dw0_slab = dw0_slab, f.xc = B.in, f.ld_xc = L.Kp0;
      f.xc_af = d.kind == 0 ? d.act_flat : 64, f.xc_skip = d.kind == 0 ? d.time_dim : 0;
      f.dw0_round = d.kind == 0 && dw0_nh < Kft ? 1 : 0;  // (the last one-hot column is rebuilt from the bias gradient)
    }
    g_fused_fault = launch_fused_backward<P>(d, f, s);
    if (g_fused_fault != 0) return;
    {
      // bias gradients = column sums, reduced over tiles in one launch; slot order: dh[nb..0], then dz1[nb-1..0]
      SlotOuts so;
      memset(&so, 0, sizeof(so));
      so.n_slots = 2 * nb + 1;
      for (int b = nb - 1; b >= 0; --b) {
        so.out[nb - (b + 1)] = grad + pl.l2b[b];
        so.out[(nb + 1) + (nb - 1 - b)] = grad + pl.l1b[b];
      }
      so.out[nb] = grad + pl.b0;
      if (d.use_layernorm) {  // 4 more slots per block, top block first: d gamma1, d beta1, d gamma2, d beta2
        for (int b = nb - 1; b >= 0; --b) {
          const int ls = (2 * nb + 1) + 4 * (nb - 1 - b);
          so.out[ls] = grad + pl.n1w[b], so.out[ls + 1] = grad + pl.n1b[b];
          so.out[ls + 2] = grad + pl.n2w[b], so.out[ls + 3] = grad + pl.n2b[b];
        }
        so.n_slots += 4 * nb;
      }
      for (int i = 0; i < so.n_slots; ++i) so.n[i] = H;
      if (one) so.n[0] = 0;  // colsum(dh_nb) is not formed by the one-block kernel: db2 comes from PostReduce::db2
      // in-kernel dW0 without room for the last one-hot column: the bias gradient is the column sums of the rounded dh_0
      const float* s_rest = dw0 && f.dw0_round ? grad + pl.b0 : nullptr;
      if (!bout_done) so.out[so.n_slots] = grad + pl.bout, so.n[so.n_slots] = d.out_dim, ++so.n_slots;  // the d_out slot
      // The bias sums and the loss statistics ride in the slab-reduction launch behind the GEMMs (tail_reduce_kernel).
      // Only the rare time-embedding gradient WITHOUT the one-hot columns (a gemm_nt + segmented sum over all rows, see
      // temb_onehot_col) still runs beside the GEMMs on a side stream (it shares B.part with nothing on s).
      const int oh = d.kind == 0 ? temb_onehot_col<P>(d, L, Kft, B) : -1;  // must match what the row builder was told
      const bool need_aux = d.kind == 0 && oh < 0;
      hipStream_t aux = aux_idx >= 0 && need_aux ? fork_side(s, aux_idx) : s;
      if (need_aux) time_embedding_grad<P>(d, prm, pk, L, M, B, B.dh_all[0], grad, krow, ksteps, Kft, aux);
      // merged top (the forward never formed h_nb): dWout = d_out^T . h_nb is rebuilt behind the slab reduce from
      // U = d_out^T . x and T = d_out^T . act(z1) (PostReduce::U); T is then needed whether or not dW2 uses it
      const bool merged = B.merged;
      const bool side_tail = dw0 && d.kind == 0 && oh >= 0 && aux_idx >= 0 && !need_aux && g_side_tail && g_early_join && g_post_one &&
                             B.post_zeroed && !(g_dbg & 28);
      TailReduce side_t;
      memset(&side_t, 0, sizeof(side_t));
      if (frag) {  // the same four products from fragment operands (one block, merged, low-rank: see frag_ok())
        const int ntx = L.Kp0 / 16, nth = H / 16;
        // U^T = x^T . d_out  [in_dim][out_dim] -> lowrank_u [out_dim][Kp0]
        weight_grad_frag<P>(B.xf, ntx, d.in_dim, B.doutf, B.dof_nt, d.out_dim, B, B.lowrank_u, L.Kp0, s, true);
        // T^T = act(z1)^T . d_out  [H][out_dim] -> lowrank [out_dim][H]
        weight_grad_frag<P>((const u32x4*)B.a2[0], nth, H, B.doutf, B.dof_nt, d.out_dim, B, B.lowrank, H, s, true);
        weight_grad_frag<P>((const u32x4*)B.dz1_all[0], nth, H, (const u32x4*)B.a1[0], nth, H, B, grad + pl.l1w[0], H, s);
        if (oh >= 0)
          weight_grad_frag<P>((const u32x4*)B.dh_all[0], nth, H, B.xf, ntx, d.in_dim + Kft, B, grad + pl.W0, d.in_dim, s, false,
                              d.in_dim, B.part, Kft);
        else
          weight_grad_frag<P>((const u32x4*)B.dh_all[0], nth, H, B.xf, ntx, d.in_dim, B, grad + pl.W0, d.in_dim, s);
      } else {
      if (merged)
        weight_grad<P>(B.d_out, L.Kpo, d.out_dim, B.in, L.Kp0, d.in_dim, M, B, B.lowrank_u, L.Kp0, s, true);
      else
        weight_grad<P>(B.d_out, L.Kpo, d.out_dim, B.hE, H, H, M, B, grad + pl.Wout, H, s, true);
      for (int b = nb - 1; b >= 0; --b) {
        if ((lowrank || merged) && b == nb - 1)  // T = d_out^T . act(z1), [out_dim][H]; dW2 = Wout^T . T after the slab reduce
          weight_grad<P>(B.d_out, L.Kpo, d.out_dim, B.a2[b], H, H, M, B, B.lowrank, H, s, true);
        if (!(lowrank && b == nb - 1))
          weight_grad<P>(B.dh_all[b + 1], H, H, B.a2[b], H, H, M, B, grad + pl.l2w[b], H, s, true);
        weight_grad<P>(B.dz1_all[b], H, H, B.a1[b], H, H, M, B, grad + pl.l1w[b], H, s, true);
      }
      if (dw0) {  // the kernel's per-workgroup partials [grid][H][32]: data columns -> dW0, one-hot columns -> S[h][k] (B.part)
        auto job = [&](int c0, int cols, float* out, int ldo) {
          SlabJob& j = side_tail ? side_t.jobs.j[side_t.jobs.n++] : B.slab_jobs.j[B.slab_jobs.n++];
          j.slab = dw0_slab, j.out = out, j.splits = dw0_grid, j.rows = H, j.cols = cols, j.lds = 32, j.ldo = ldo, j.transpose = 0;
          j.c0 = c0, j.wide = dw0_grid >= 32 ? 1 : 0;  // (a slab per workgroup: slab_job_block_wide)
        };
        if (d.kind == 0) {
          job(0, d.act_flat, grad + pl.W0, d.in_dim);
          job(d.act_flat, d.cond_dim, grad + pl.W0 + d.act_flat + d.time_dim, d.in_dim);
          job(d.act_flat + d.cond_dim, dw0_nh, B.part, Kft);
        } else {
          job(0, d.in_dim, grad + pl.W0, d.in_dim);
        }
      } else if (g_dbg & 32) {  // timing experiment: no dW0 product in the group
      } else if (oh >= 0)  // + Kft one-hot columns: their block of the result is S[h][k] (B.part), see temb_onehot_col()
        weight_grad<P>(B.dh_all[0], H, H, B.in, L.Kp0, d.in_dim + Kft, M, B, grad + pl.W0, d.in_dim, s, true, d.in_dim, B.part,
                       Kft);
      else
        weight_grad<P>(B.dh_all[0], H, H, B.in, L.Kp0, d.in_dim, M, B, grad + pl.W0, d.in_dim, s, true);
      }
      // ... as riders of the GEMM launch (knob 40): needs the grouped launch to exist and take them (three wide jobs, the time MLP's
      // LDS within the GEMM's own, no folded reduction), else the side stream
      const bool riders = side_tail && g_tail_riders && !g_fold && B.tn_group.n > 0 && side_t.jobs.n <= 3 && so.n_slots <= TN_MAX_SLOTS &&
                          time_backward_lds_bytes(Kft, d.time_dim) <= 32 * 1024;
      if (side_tail) {
        // what the backward kernel alone feeds: its dW0 slabs, the bias sums, the loss statistics, then the time-embedding
        // gradient -- queued on the side stream here, BEHIND the kernel and BESIDE the GEMM launch that follows on s
        hipStream_t st = riders ? s : fork_side(s, aux_idx);
        side_t.colsum = B.tile_colsum, side_t.tiles = B.tiles, side_t.width = H, side_t.slots = so;
        if (riders) {
          if (fin != nullptr && fin->N > 0) {
            side_t.fin_partial = fin->partial, side_t.fin_blocks = loss_blocks(fin->N), side_t.fin_moments = fin->moments;
            side_t.fin_stats = fin->stats, side_t.fin_part = fin->part, side_t.fin_n_count = fin->n_count;
          }
        } else {
          launch_tail_reduce(side_t, fin, st);
        }
        PostReduce qs;
        memset(&qs, 0, sizeof(qs));
        qs.H = H, qs.out_dim = d.out_dim;
        qs.S = B.part, qs.W0 = prm + pl.W0, qs.ldw0 = d.in_dim, qs.AF = d.act_flat, qs.Kft = Kft, qs.td = d.time_dim;
        qs.G = B.part + (size_t)H * Kft, qs.w1 = prm + pl.te1_w, qs.b1 = prm + pl.te1_b, qs.w2 = prm + pl.te2_w;
        qs.ksteps = ksteps, qs.gw1 = grad + pl.te1_w, qs.gb1 = grad + pl.te1_b, qs.gw2 = grad + pl.te2_w, qs.gb2 = grad + pl.te2_b;
        qs.S_rest = s_rest, qs.dW0t = grad + pl.W0, qs.temb = (const float*)(pk + L.temb), qs.temb_bf16 = P::ESIZE == 2 ? 1 : 0;
        qs.counter = (unsigned*)B.post_counter;
        if (riders) {
          B.ride = true, B.ride_t = side_t, B.ride_q = qs;  // (flush_slabs hands them to the group launch)
        } else {
          launch_post_reduce(qs, st);
          B.join_s[B.n_join] = st, B.join_idx[B.n_join++] = aux_idx;  // (joined right behind the GEMM launch: flush_slabs)
        }
      }
      if (aux != s && g_early_join) B.join_s[B.n_join] = aux, B.join_idx[B.n_join++] = aux_idx;
      const bool tail_post = side_tail && g_tail_post && !g_fold && (lowrank || merged || one);
      if (tail_post)
        flush_slabs(B, s, nullptr, 0, nullptr, true);  // the GEMMs; their slabs are reduced with the post-reduce parts below
      else if (side_tail)
        flush_slabs(B, s);  // the GEMMs' own slabs
      else
        flush_slabs(B, s, &so, H, fin);  // every slab of this backward, its bias sums and the loss statistics: one launch
      if (aux != s && !g_early_join) join_side(s, aux, aux_idx);
      PostReduce q;
      memset(&q, 0, sizeof(q));
      q.H = H, q.out_dim = d.out_dim, q.T = B.lowrank;
      if (merged) {  // (cs = column sums of d_out = the out-layer bias gradient, reduced on the aux stream joined above)
        q.U = B.lowrank_u, q.ldu = L.Kp0, q.in_dim = d.in_dim, q.W0 = prm + pl.W0, q.ldw0 = d.in_dim, q.W2 = prm + pl.l2w[nb - 1];
        q.b0 = prm + pl.b0, q.b2 = prm + pl.l2b[nb - 1], q.cs = grad + pl.bout, q.dWout = grad + pl.Wout;
      }
      if (one) q.db2 = grad + pl.l2b[0], q.Wout_b = prm + pl.Wout, q.cs = grad + pl.bout;
      // everything behind the reduction in one launch (knob 18); its arrival counter (zeroed by the row builder) is only
      // needed by the time-embedding part
      if (g_post_one && (oh < 0 || B.post_zeroed) && (lowrank || oh >= 0 || merged || one)) {
        if (lowrank) q.Wout = prm + pl.Wout, q.dW = grad + pl.l2w[nb - 1];
        if (oh >= 0 && !side_tail) {
          q.S = B.part, q.W0 = prm + pl.W0, q.ldw0 = d.in_dim, q.AF = d.act_flat, q.Kft = Kft, q.td = d.time_dim;
          q.G = B.part + (size_t)H * Kft, q.w1 = prm + pl.te1_w, q.b1 = prm + pl.te1_b, q.w2 = prm + pl.te2_w;
          q.ksteps = ksteps, q.gw1 = grad + pl.te1_w, q.gb1 = grad + pl.te1_b, q.gw2 = grad + pl.te2_w;
          q.gb2 = grad + pl.te2_b;
          if (dw0)  // the time-embedding columns of dW0 from the one-hot sums, the last of which may have to be rebuilt
            q.S_rest = s_rest, q.dW0t = grad + pl.W0, q.temb = (const float*)(pk + L.temb), q.temb_bf16 = P::ESIZE == 2 ? 1 : 0;
        }
        q.counter = (unsigned*)B.post_counter;
        if (tail_post) {  // (no time-embedding part here: it went to the side stream)
          TailPost tp;
          memset(&tp, 0, sizeof(tp));
          for (int pass = 0; pass < 2; ++pass)  // the thin products' jobs first
            for (int i = 0; i < B.slab_jobs.n; ++i) {
              const SlabJob& j = B.slab_jobs.j[i];
              const bool thin = j.out == B.lowrank || j.out == B.lowrank_u;
              if (thin == (pass == 0)) tp.jobs.j[tp.jobs.n++] = j;
              if (thin && pass == 0) ++tp.n_first_jobs;
            }
          tp.q = q;
          tp.q.wait_cnt = B.red_cnt + RED_CNT - 2;  // (zeroed by the row builder: B.post_zeroed)
          launch_tail_post(tp, s);
          B.slab_jobs.n = 0, B.slab_used = 0;
        } else {
          launch_post_reduce(q, s);
        }
      } else {
        launch_wout_grad(q, s);  // (merged top and / or one-block backward; nothing otherwise)
        if (lowrank) launch_lowrank_dw(prm + pl.Wout, B.lowrank, d.out_dim, H, grad + pl.l2w[nb - 1], s);
        if (oh >= 0)
          launch_time_backward_from_sums(prm + pl.te1_w, prm + pl.te1_b, prm + pl.te2_w, B.part, prm + pl.W0, d.in_dim,
                                         d.act_flat, H, B.part + (size_t)H * Kft, ksteps, Kft, d.time_dim, grad + pl.te1_w,
                                         grad + pl.te1_b, grad + pl.te2_w, grad + pl.te2_b, s);
      }
      B.dh0_final = frag || dw0 ? nullptr : B.dh_all[0];  // (fragment mode / in-kernel dW0: no row-major dh_0 exists; frag_ok() / dw0_ok() made sure nobody asks)
      return;
    }
  }
  if (d.plain) {
    if (fin) launch_loss_finalize(*fin, s);
    weight_grad<P>(B.d_out, L.Kpo, d.out_dim, B.a2[nb - 1], H, H, M, B, grad + pl.Wout, H, s);
    launch_colsum<P>(B.d_out, (int)M, d.out_dim, L.Kpo, B.part, REDUCE_BLOCKS, grad + pl.bout, 1.f, s);
    GemmNT q;  // dz = (upstream . W) * act'(z): starts from d_out . Wout at the last hidden layer's pre-activation
    memset(&q, 0, sizeof(q));
    q.M = (int)M, q.N = H, q.Kp = L.Kpo, q.ldx = L.Kpo, q.ldw = L.Kpo, q.ldo = H;
    q.X = B.d_out, q.W = pk + L.WoutT, q.dsrc = B.z1[nb - 1], q.dsrc_kind = 2, q.dsrc_ld = H, q.dact = d.act, q.out_pre = B.dh;
    launch_gemm_nt<P>(q, s);
    void* dz = B.dh;
    void* other = B.dz1;
    for (int b = nb - 1; b >= 0; --b) {
      weight_grad<P>(dz, H, H, b == 0 ? B.a1[0] : B.a2[b - 1], H, H, M, B, grad + pl.l1w[b], H, s);
      launch_colsum<P>(dz, (int)M, H, H, B.part, REDUCE_BLOCKS, grad + pl.l1b[b], 1.f, s);
      memset(&q, 0, sizeof(q));
      q.M = (int)M, q.N = H, q.Kp = H, q.ldx = H, q.ldw = H, q.ldo = H;
      q.X = dz, q.W = pk + L.W1T[b], q.dsrc = b == 0 ? B.hpre[0] : B.z1[b - 1], q.dsrc_kind = 2, q.dsrc_ld = H, q.dact = d.act;
      q.out_pre = other;
      launch_gemm_nt<P>(q, s);
      void* t = dz;
      dz = other, other = t;
    }
    weight_grad<P>(dz, H, H, B.in, L.Kp0, d.in_dim, M, B, grad + pl.W0, d.in_dim, s);
    launch_colsum<P>(dz, (int)M, H, H, B.part, REDUCE_BLOCKS, grad + pl.b0, 1.f, s);
    if (d.kind == 0) time_embedding_grad<P>(d, prm, pk, L, M, B, dz, grad, krow, ksteps, Kft, s);
    B.dh0_final = dz;
    return;
  }
  if (fin) launch_loss_finalize(*fin, s);
  // output layer parameters
  weight_grad<P>(B.d_out, L.Kpo, d.out_dim, B.hE, H, H, M, B, grad + pl.Wout, H, s);
  launch_colsum<P>(B.d_out, (int)M, d.out_dim, L.Kpo, B.part, REDUCE_BLOCKS, grad + pl.bout, 1.f, s);
  // dh = d_out . Wout
  GemmNT g;
  memset(&g, 0, sizeof(g));
  g.M = (int)M, g.N = H, g.Kp = L.Kpo, g.ldx = L.Kpo, g.ldw = L.Kpo, g.ldo = H;
  g.X = B.d_out, g.W = pk + L.WoutT, g.out_pre = B.dh;
  launch_gemm_nt<P>(g, s);
  for (int b = nb - 1; b >= 0; --b) {
    // l2: dz2 = dh
    weight_grad<P>(B.dh, H, H, B.a2[b], H, H, M, B, grad + pl.l2w[b], H, s);
    launch_colsum<P>(B.dh, (int)M, H, H, B.part, REDUCE_BLOCKS, grad + pl.l2b[b], 1.f, s);
    // dz1 = (dh . W2) * act'(z1)
    memset(&g, 0, sizeof(g));
    g.M = (int)M, g.N = H, g.Kp = H, g.ldx = H, g.ldw = H, g.ldo = H;
    g.X = B.dh, g.W = pk + L.W2T[b], g.dsrc = B.z1[b], g.dsrc_kind = 2, g.dsrc_ld = H, g.dact = d.act;
    g.out_pre = B.dz1;
    launch_gemm_nt<P>(g, s);
    weight_grad<P>(B.dz1, H, H, B.a1[b], H, H, M, B, grad + pl.l1w[b], H, s);
    launch_colsum<P>(B.dz1, (int)M, H, H, B.part, REDUCE_BLOCKS, grad + pl.l1b[b], 1.f, s);
    // dh <- dh + (dz1 . W1) * act'(h[b])      (in place: each element is read then written by one lane)
    memset(&g, 0, sizeof(g));
    g.M = (int)M, g.N = H, g.Kp = H, g.ldx = H, g.ldw = H, g.ldo = H;
    g.X = B.dz1, g.W = pk + L.W1T[b], g.dsrc = B.h[b], g.dsrc_kind = 1, g.dsrc_ld = H, g.dact = d.act;
    g.add = B.dh, g.ldadd = H, g.out_pre = B.dh;
    launch_gemm_nt<P>(g, s);
  }
  // layer 0: dW0[h][c] = sum_m dh[m][h] in[m][c]
  weight_grad<P>(B.dh, H, H, B.in, L.Kp0, d.in_dim, M, B, grad + pl.W0, d.in_dim, s);
  launch_colsum<P>(B.dh, (int)M, H, H, B.part, REDUCE_BLOCKS, grad + pl.b0, 1.f, s);
  if (d.kind == 0) time_embedding_grad<P>(d, prm, pk, L, M, B, B.dh, grad, krow, ksteps, Kft, s);
  B.dh0_final = B.dh;
}

// d loss / d observation of a trunk WITHOUT an observation encoder: d_obs[m][c] = sum_h dh0[m][h] W0[h][col0 + c], where the
// observation sits in layer 0's input at columns col0 = act_flat + time_dim (actor: cat[x, temb, obs]) or 0 (critic).  What a
// visual encoder in front of the trunk (vision.hip) back-propagates from.  One transposing pack of cond_dim x H weights
// + one gemm_nt.
template <class P>
static void obs_grad(const dppo_net_desc& d, const float* prm, int64_t M, MlpBufs<P>& B, float* d_obs, hipStream_t s) {
  const ParamLayout pl = param_layout(d);
  const int H = d.hidden, col0 = d.kind == 0 ? d.act_flat + d.time_dim : 0;
  launch_transpose_cast<P>(prm + pl.W0, H, d.cond_dim, d.in_dim, col0, B.w0T, H, s);
  GemmNT g;
  memset(&g, 0, sizeof(g));
  g.M = (int)M, g.N = d.cond_dim, g.Kp = H, g.ldx = H, g.ldw = H, g.X = B.dh0_final, g.W = B.w0T;
  g.out_f32 = B.dobs, g.ldo32 = round_up(d.cond_dim, 16);
  launch_gemm_nt<P>(g, s);
  launch_copy_cols(B.dobs, g.ldo32, 0, d.cond_dim, M, d_obs, s);
}

// ------------------------------------------------------------------------------------------------
// exported functions
// ------------------------------------------------------------------------------------------------
// Exported functions take C linkage from their declarations in include/dppo_hip.h.

int dppo_version(void) { return 1; }
const char* dppo_last_error(void) { return g_err; }

int64_t dppo_net_param_count(const dppo_net_desc* net) {
  if (check_net(net)) return -1;
  return param_layout(*net).total;
}

int64_t dppo_packed_bytes(const dppo_net_desc* net, int prec, int n_time) {
  if (check_net(net) || check_prec(prec)) return -1;
  if (n_time < 0) return fail(-1, "n_time < 0");
#define CALL(P) (int64_t) pack_layout<P>(*net, n_time).total
  return DPPO_DISPATCH(prec, CALL);
#undef CALL
}

template <class P>
static int pack_two_impl(const dppo_net_desc& d0, int t0, const float* p0, char* k0, const dppo_net_desc& d1, int t1,
                         const float* p1, char* k1, hipStream_t s) {
  PackNets q;
  ComposeJobs cq;
  memset(&q, 0, sizeof(q));
  memset(&cq, 0, sizeof(cq));
  bool def0 = false, def1 = false;
  ComposeJob c0, c1;
  memset(&c0, 0, sizeof(c0));
  memset(&c1, 0, sizeof(c1));
  if (int e = pack_impl<P>(d0, t0, p0, k0, s, &q.n[0], &c0, &def0)) return e;
  if (int e = pack_impl<P>(d1, t1, p1, k1, s, &q.n[1], &c1, &def1)) return e;
  if (def0 && c0.Wout) cq.j[cq.n++] = c0;
  if (def1 && c1.Wout) cq.j[cq.n++] = c1;
  launch_compose(cq, s);  // both composites in one launch, then both networks' images in one launch
  if (def0 && def1)
    launch_pack_nets<P>(q, s);
  else if (def0)
    launch_pack_net<P>(q.n[0], s);
  else if (def1)
    launch_pack_net<P>(q.n[1], s);
  return check_launch();
}

int dppo_pack_nets(const dppo_net_desc* net0, int n_time0, const float* params0, void* packed0,
                   const dppo_net_desc* net1, int n_time1, const float* params1, void* packed1, int prec,
                   dppo_stream_t stream) {
  if (int e = check_net(net0)) return e;
  if (int e = check_net(net1)) return e;
  if (int e = check_prec(prec)) return e;
  if (!params0 || !packed0 || !params1 || !packed1) return fail(-1, "null pointer");
  if ((net0->kind == 0 && n_time0 < 1) || (net1->kind == 0 && n_time1 < 1)) return fail(-1, "actor needs n_time >= 1");
#define CALL(P)                                                                                                  \
  pack_two_impl<P>(*net0, n_time0, params0, (char*)packed0, *net1, n_time1, params1, (char*)packed1, (hipStream_t)stream)
  return DPPO_DISPATCH(prec, CALL);
#undef CALL
}

int dppo_pack_net(const dppo_net_desc* net, int prec, int n_time, const float* params, void* packed,
                  dppo_stream_t stream) {
  if (int e = check_net(net)) return e;
  if (int e = check_prec(prec)) return e;
  if (!params || !packed) return fail(-1, "null pointer");
  if (net->kind == 0 && n_time < 1) return fail(-1, "actor needs n_time >= 1");
#define CALL(P) pack_impl<P>(*net, n_time, params, (char*)packed, (hipStream_t)stream)
  return DPPO_DISPATCH(prec, CALL);
#undef CALL
}

// ---- forwards --------------------------------------------------------------------------------------
template <class P>
static int64_t fwd_ws_bytes(const dppo_net_desc& d, int64_t rows) {
  Carver c{nullptr, 0, 0};
  MlpBufs<P> B;
  carve_mlp<P>(c, d, rows, false, false, B);
  return (int64_t)al256(c.off);
}
int64_t dppo_mlp_forward_workspace_bytes(const dppo_net_desc* net, int prec, int64_t rows) {
  if (check_net(net) || check_prec(prec)) return -1;
  if (rows < 0 || rows > 0x7fffffff) return fail(-1, "rows out of range");
#define CALL(P) fwd_ws_bytes<P>(*net, rows)
  return DPPO_DISPATCH(prec, CALL);
#undef CALL
}

template <class P>
static int net_forward_impl(const dppo_net_desc& d, const float* prm, const char* pk, const float* x,
                            const int64_t* t, const float* state, int64_t M, float* out, void* ws, int64_t wsb,
                            hipStream_t s) {
  Carver c{(char*)ws, 0, (size_t)wsb};
  MlpBufs<P> B;
  carve_mlp<P>(c, d, M, false, false, B);
  if ((int64_t)c.off > wsb) return fail(-1, "workspace too small: need %zu bytes, got %lld", c.off, (long long)wsb);
  const PackLayout L = pack_layout<P>(d, 0);
  if (d.cond_hidden > 0) {  // [x | temb | 0] rows, then the encoder writes its columns
    launch_build_direct<P>(x, t, nullptr, (const float*)(pk + L.temb), d.act_flat, d.time_dim, 0, M, B.in, L.Kp0, s);
    launch_build_direct<P>(nullptr, nullptr, state, nullptr, 0, 0, d.cond_dim, M, B.cin, L.Kpc, s);
    cond_encode<P>(d, prm, pk, L, M, B.cin, B, B.in, nullptr, 0, false, s);
  } else {
    launch_build_direct<P>(x, t, state, (const float*)(pk + L.temb), d.act_flat, d.time_dim, d.cond_dim, M, B.in,
                           L.Kp0, s);
  }
  mlp_forward<P>(d, prm, pk, L, M, B, false, s);
  launch_slab_reduce_2d(B.out, 1, (int)M, d.out_dim, B.ldout, out, d.out_dim, 1.f, s);
  return check_launch();
}

int dppo_actor_forward(const dppo_net_desc* net, int prec, const float* params, const void* packed, const float* x,
                       const int64_t* t, const float* state, int64_t rows, float* eps, void* workspace,
                       int64_t workspace_bytes, dppo_stream_t stream) {
  if (int e = check_net(net)) return e;
  if (int e = check_prec(prec)) return e;
  if (net->kind != 0) return fail(-1, "dppo_actor_forward needs an actor descriptor");
  if (!params || !packed || !x || !t || !state || !eps || !workspace) return fail(-1, "null pointer");
  if (rows <= 0 || rows > 0x7fffffff) return fail(-1, "rows out of range");
#define CALL(P) \
  net_forward_impl<P>(*net, params, (const char*)packed, x, t, state, rows, eps, workspace, workspace_bytes, (hipStream_t)stream)
  return DPPO_DISPATCH(prec, CALL);
#undef CALL
}

int dppo_critic_forward(const dppo_net_desc* net, int prec, const float* params, const void* packed,
                        const float* state, int64_t rows, float* values, void* workspace, int64_t workspace_bytes,
                        dppo_stream_t stream) {
  if (int e = check_net(net)) return e;
  if (int e = check_prec(prec)) return e;
  if (net->kind != 1) return fail(-1, "dppo_critic_forward needs a critic descriptor");
  if (!params || !packed || !state || !values || !workspace) return fail(-1, "null pointer");
  if (rows <= 0 || rows > 0x7fffffff) return fail(-1, "rows out of range");
#define CALL(P)                                                                                                    \
  net_forward_impl<P>(*net, params, (const char*)packed, nullptr, nullptr, state, rows, values, workspace, workspace_bytes, \
                      (hipStream_t)stream)
  return DPPO_DISPATCH(prec, CALL);
#undef CALL
}

// ---- sampler -----------------------------------------------------------------------------------------
template <class P>
static bool sample_split(const dppo_net_desc& d, int64_t B) {
  return sampler_split_ok(d, P::ESIZE == 2, B, g_merge_top && d.n_blocks >= 1);
}
template <class P>
static size_t sample_carve(Carver& c, const dppo_net_desc& d, int64_t B, MlpBufs<P>& Bz, float*& enc, void** xch = nullptr) {
  memset(&Bz, 0, sizeof(Bz));
  enc = nullptr;
  if (sample_split<P>(d, B)) {  // the split sampler's exchange block: first, so that its memset starts at the allocation
    void* x = c.take(sampler_split_xch_bytes(d, B));
    if (xch) *xch = x;
  }
  if (d.cond_hidden <= 0) return al256(c.off);
  const size_t ES = P::ESIZE;
  Bz.cin = c.take((size_t)B * round_up(d.cond_dim, 64) * ES);
  Bz.ca = c.take((size_t)B * round_up(d.cond_hidden, 64) * ES);
  enc = (float*)c.take((size_t)2 * B * round_up(d.cond_out, 16) * 4);
  return al256(c.off);
}
template <class P>
static size_t sample_ws(const dppo_net_desc& d, int64_t B) {
  Carver c{nullptr, 0, 0};
  MlpBufs<P> Bz;
  float* enc;
  return sample_carve<P>(c, d, B, Bz, enc);
}

template <class P>
static int sample_impl(const dppo_net_desc& d, const float* pb, const char* kb, const float* pf, const char* kf,
                       const dppo_diffusion_cfg& cfg, const dppo_step* sched, int n_steps, const float* obs,
                       const float* noise, int64_t B, float* traj, float* chains, int chain_len, int init_slot, void* ws,
                       int64_t wsb, hipStream_t s) {
  const SamplerGeom g = sampler_geom<P>(d);
  const PackLayout L = pack_layout<P>(d, 0);
  const ParamLayout pl = param_layout(d);
  SampleArgs a;
  memset(&a, 0, sizeof(a));
  a.wstream[0] = (const u32x4*)(kb + L.sstream), a.wstream[1] = (const u32x4*)(kf + L.sstream);
  a.ostream[0] = (const u32x4*)(kb + L.ostream), a.ostream[1] = (const u32x4*)(kf + L.ostream);
  a.ostream2[0] = (const u32x4*)(kb + L.ostream2), a.ostream2[1] = (const u32x4*)(kf + L.ostream2);
  a.cbias[0] = (const float*)(kb + L.cbias), a.cbias[1] = (const float*)(kf + L.cbias);
  a.merge_top = g_merge_top && d.n_blocks >= 1 ? 1 : 0;
  a.params[0] = pb, a.params[1] = pf;
  a.temb[0] = (const float*)(kb + L.temb), a.temb[1] = (const float*)(kf + L.temb);
  fill_bias_off(d, pl, a.bias_off);
  a.use_ln = d.use_layernorm;
  if (d.use_layernorm) fill_ln_off(d, pl, a.ln_off);
  a.obs[0] = a.obs[1] = obs, a.cond = d.cond_dim, a.ld_obs = d.cond_dim;
  void* xch = nullptr;
  const bool split = sample_split<P>(d, B);
  if (d.cond_hidden > 0 || split) {
    MlpBufs<P> Bz;
    float* enc = nullptr;
    if (!ws || (int64_t)sample_ws<P>(d, B) > wsb)
      return fail(-1, "sampler workspace too small: need %zu bytes (dppo_sample_chain_workspace_bytes)", sample_ws<P>(d, B));
    Carver c{(char*)ws, 0, (size_t)wsb};
    sample_carve<P>(c, d, B, Bz, enc, &xch);
    if (d.cond_hidden > 0) {  // per-network encoded observation, computed once per call (constant over the K steps)
      const int lde = round_up(d.cond_out, 16);
      launch_build_direct<P>(nullptr, nullptr, obs, nullptr, 0, 0, d.cond_dim, B, Bz.cin, L.Kpc, s);
      cond_encode<P>(d, pb, kb, L, B, Bz.cin, Bz, nullptr, enc, lde, false, s);
      cond_encode<P>(d, pf, kf, L, B, Bz.cin, Bz, nullptr, enc + (size_t)B * lde, lde, false, s);
      a.obs[0] = enc, a.obs[1] = enc + (size_t)B * lde, a.cond = d.cond_out, a.ld_obs = lde;
    }
  }
  a.noise = noise, a.seed_lo = cfg.seed_lo, a.seed_hi = cfg.seed_hi, a.traj = traj, a.chains = chains, a.sched = sched;
  a.B = (int)B, a.AF = d.act_flat, a.td = d.time_dim, a.Kp0 = g.Kp0, a.nb = d.n_blocks;
  a.n_steps = n_steps, a.chain_len = chain_len, a.init_slot = init_slot, a.act = d.act, a.use_ddim = cfg.use_ddim;
  a.has_dclip = cfg.has_denoised_clip, a.has_eclip = cfg.has_eps_clip, a.has_fclip = cfg.has_final_clip;
  a.dclip = cfg.denoised_clip, a.eclip = cfg.eps_clip, a.rclip = cfg.randn_clip, a.fclip = cfg.final_clip;
  if (split) {
    const int rs = launch_sample_chain_split(g, a, xch, sampler_split_xch_bytes(d, B), s);
    if (rs == 0) return check_launch();
    if (rs != -1) return fail(-1, "split sampler: launch failed (%d)", rs);
  }
  const int rc = launch_sample_chain<P>(g, a, s);
  if (rc == -1) return fail(-1, "sampler: hidden=%d / out_dim=%d not instantiated (hidden in {256,512,768,1024}, out_dim <= 128)", d.hidden, d.out_dim);
  if (rc == -2) return fail(-1, "sampler: LDS image exceeds 160 KiB for hidden=%d at this precision", d.hidden);
  return check_launch();
}

// ---- plain (non-residual) trunks: the K-step loop on the host, one layered forward + one posterior kernel per step -------
template <class P>
static size_t carve_plain_sample(Carver& c, const dppo_net_desc& d, int64_t B, MlpBufs<P>& Bz, float*& x) {
  carve_mlp<P>(c, d, B, false, false, Bz);
  x = (float*)c.take((size_t)B * d.act_flat * 4);
  return al256(c.off);
}
int64_t dppo_plain_sample_workspace_bytes(const dppo_net_desc* actor, int prec, int64_t B) {
  if (check_net(actor) || check_prec(prec)) return -1;
  if (actor->kind != 0 || !actor->plain) return fail(-1, "dppo_plain_sample_chain needs a plain actor descriptor");
  if (B < 1 || B > 0x7fffffff) return fail(-1, "B out of range");
  Carver c{nullptr, 0, 0};
  float* x;
  if (prec == DPPO_PREC_F32) {
    MlpBufs<F32> Bz;
    return (int64_t)carve_plain_sample<F32>(c, *actor, B, Bz, x);
  }
  MlpBufs<BF16> Bz;
  return (int64_t)carve_plain_sample<BF16>(c, *actor, B, Bz, x);
}
template <class P>
static int plain_sample_impl(const dppo_net_desc& d, const float* pb, const char* kb, const float* pf, const char* kf,
                             const dppo_diffusion_cfg& cfg, const dppo_step* sched, int n_steps, const float* obs,
                             const float* noise, int64_t B, float* traj, float* chains, int chain_len, int init_slot, void* ws,
                             int64_t wsb, hipStream_t s) {
  Carver c{(char*)ws, 0, (size_t)wsb};
  MlpBufs<P> Bz;
  float* x;
  const size_t need = carve_plain_sample<P>(c, d, B, Bz, x);
  if ((int64_t)need > wsb) return fail(-1, "workspace too small: need %zu bytes, got %lld", need, (long long)wsb);
  const PackLayout L = pack_layout<P>(d, 0);
  const int AF = d.act_flat;
  const int64_t n = B * AF;
  launch_chain_init(noise, cfg.seed_lo, cfg.seed_hi, n, AF, x, chains, chain_len, init_slot, s);
  for (int i = 0; i < n_steps; ++i) {
    const dppo_step& st = sched[i];
    const float* prm = st.net ? pf : pb;
    const char* pk = st.net ? kf : kb;
    // rows [x | temb(t) | obs]: the table pointer is offset to row t, so every row reads its row 0
    launch_build_direct<P>(x, nullptr, obs, (const float*)(pk + L.temb) + (size_t)st.t * d.time_dim, AF, d.time_dim, d.cond_dim, B,
                           Bz.in, L.Kp0, s);
    mlp_forward<P>(d, prm, pk, L, B, Bz, false, s);
    launch_chain_step(cfg, st, x, Bz.out, Bz.ldout, noise, (size_t)(i + 1) * n, n, AF, chain_len, i + 1 == n_steps, chains, traj, s);
  }
  return check_launch();
}
int dppo_plain_sample_chain(const dppo_net_desc* actor, int prec, const float* params_base, const void* packed_base,
                            const float* params_ft, const void* packed_ft, const dppo_diffusion_cfg* cfg,
                            const dppo_step* sched_host, int n_steps, const float* obs, const float* noise, int64_t B,
                            float* traj, float* chains, int chain_len, int init_slot, void* workspace,
                            int64_t workspace_bytes, dppo_stream_t stream) {
  if (int e = check_net(actor)) return e;
  if (int e = check_prec(prec)) return e;
  if (actor->kind != 0 || !actor->plain) return fail(-1, "dppo_plain_sample_chain needs a plain actor descriptor");
  if (!params_base || !packed_base || !params_ft || !packed_ft || !cfg || !sched_host || !obs || !traj || !workspace)
    return fail(-1, "null pointer");
  if (B < 1 || B > 0x7fffffff || n_steps < 1) return fail(-1, "B / n_steps out of range");
  if (chains != nullptr && chain_len < 1) return fail(-1, "chain_len must be >= 1 when chains are requested");
#define CALL(P)                                                                                                             \
  plain_sample_impl<P>(*actor, params_base, (const char*)packed_base, params_ft, (const char*)packed_ft, *cfg, sched_host, n_steps, \
                       obs, noise, B, traj, chains, chain_len, init_slot, workspace, workspace_bytes, (hipStream_t)stream)
  return DPPO_DISPATCH(prec, CALL);
#undef CALL
}

int64_t dppo_sample_chain_workspace_bytes(const dppo_net_desc* actor, int prec, int64_t B) {
  if (check_net(actor) || check_prec(prec)) return -1;
  if (actor->plain) return fail(-1, "plain MLP trunks sample through dppo_plain_sample_chain (dppo_plain_sample_workspace_bytes)");
  if (B < 1) return fail(-1, "B out of range");
  return prec == DPPO_PREC_F32 ? (int64_t)sample_ws<F32>(*actor, B) : (int64_t)sample_ws<BF16>(*actor, B);
}

int64_t dppo_sample_chain_exchange_bytes(const dppo_net_desc* actor, int prec, int64_t B) {
  if (check_net(actor) || check_prec(prec)) return -1;
  if (actor->plain || B < 1) return 0;
  const bool split = prec == DPPO_PREC_F32 ? sample_split<F32>(*actor, B) : sample_split<BF16>(*actor, B);
  return split ? (int64_t)sampler_split_xch_bytes(*actor, B) : 0;
}

int dppo_sample_chain(const dppo_net_desc* actor, int prec, const float* params_base, const void* packed_base,
                      const float* params_ft, const void* packed_ft, const dppo_diffusion_cfg* cfg,
                      const dppo_step* sched, int n_steps, const float* obs, const float* noise, int64_t B, float* traj,
                      float* chains, int chain_len, int init_slot, void* workspace, int64_t workspace_bytes,
                      dppo_stream_t stream) {
  if (int e = check_net(actor)) return e;
  if (int e = check_prec(prec)) return e;
  if (actor->plain) return fail(-1, "plain MLP trunks sample through dppo_plain_sample_chain (host-looped steps)");
  if (actor->kind != 0) return fail(-1, "dppo_sample_chain needs an actor descriptor");
  if (!params_base || !packed_base || !params_ft || !packed_ft || !cfg || !sched || !obs || !traj)  // noise may be NULL
    return fail(-1, "null pointer");
  if (n_steps < 1) return fail(-1, "n_steps must be >= 1");
  if (B < 1 || B > (1 << 24)) return fail(-1, "B out of range");
  if (chain_len < 0 || (chain_len > 0 && !chains)) return fail(-1, "chains buffer missing");
  if (init_slot >= chain_len) return fail(-1, "init_slot outside the chain");
#define CALL(P)                                                                                                   \
  sample_impl<P>(*actor, params_base, (const char*)packed_base, params_ft, (const char*)packed_ft, *cfg, sched, n_steps, obs, \
                 noise, B, traj, chains, chain_len, init_slot, workspace, workspace_bytes, (hipStream_t)stream)
  return DPPO_DISPATCH(prec, CALL);
#undef CALL
}

// ---- chain log-probs ---------------------------------------------------------------------------------
template <class P>
static int64_t logprob_ws_bytes(const dppo_net_desc& d, int64_t rows) {
  Carver c{nullptr, 0, 0};
  MlpBufs<P> B;
  carve_mlp<P>(c, d, rows, false, false, B);
  c.take((size_t)rows * 4);
  c.take((size_t)rows * 4);
  return (int64_t)al256(c.off);
}
int64_t dppo_chain_logprob_workspace_bytes(const dppo_net_desc* actor, int prec, int64_t B, int Kft) {
  if (check_net(actor) || check_prec(prec)) return -1;
  if (B < 0 || Kft < 1 || B * Kft > 0x7fffffff) return fail(-1, "B*Kft out of range");
#define CALL(P) logprob_ws_bytes<P>(*actor, B* Kft)
  return DPPO_DISPATCH(prec, CALL);
#undef CALL
}

template <class P>
static int logprob_impl(const dppo_net_desc& d, const float* prm, const char* pk, const dppo_diffusion_cfg& cfg,
                        const dppo_step* ksteps, int Kft, const float* obs, const float* chains, int64_t Bn, float* logp,
                        void* ws, int64_t wsb, hipStream_t s) {
  const int64_t M = Bn * Kft;
  Carver c{(char*)ws, 0, (size_t)wsb};
  MlpBufs<P> B;
  carve_mlp<P>(c, d, M, false, false, B);
  int32_t* brow = (int32_t*)c.take((size_t)M * 4);
  int32_t* krow = (int32_t*)c.take((size_t)M * 4);
  if ((int64_t)c.off > wsb) return fail(-1, "workspace too small: need %zu bytes, got %lld", c.off, (long long)wsb);
  const PackLayout L = pack_layout<P>(d, 0);
  BuildRows br;
  memset(&br, 0, sizeof(br));
  br.chains = chains, br.obs = obs, br.temb = (const float*)(pk + L.temb), br.ksteps = ksteps;
  br.Kft = Kft, br.AF = d.act_flat, br.td = d.time_dim, br.cond = d.cond_dim, br.M = M, br.obs_in_a = 1;
  br.inA = B.in, br.KpA = L.Kp0, br.brow = brow, br.krow = krow, br.onehot0 = -1;
  if (d.cond_hidden > 0) br.obs_in_a = 0, br.inC = B.cin, br.KpC = L.Kpc;
  launch_build_rows<P>(br, s);
  if (d.cond_hidden > 0) cond_encode<P>(d, prm, pk, L, M, B.cin, B, B.in, nullptr, 0, false, s);
  mlp_forward<P>(d, prm, pk, L, M, B, false, s);
  LogprobArgs la;
  la.eps = B.out, la.lde = B.ldout, la.chains = chains, la.ksteps = ksteps, la.cfg = cfg, la.Kft = Kft;
  la.AF = d.act_flat, la.M = M, la.logp = logp;
  launch_logprob(la, s);
  return check_launch();
}

int dppo_chain_logprob(const dppo_net_desc* actor, int prec, const float* params, const void* packed,
                       const dppo_diffusion_cfg* cfg, const dppo_step* ksteps, int Kft, const float* obs,
                       const float* chains, int64_t B, float* logprobs, void* workspace, int64_t workspace_bytes,
                       dppo_stream_t stream) {
  if (int e = check_net(actor)) return e;
  if (int e = check_prec(prec)) return e;
  if (actor->kind != 0) return fail(-1, "dppo_chain_logprob needs an actor descriptor");
  if (!params || !packed || !cfg || !ksteps || !obs || !chains || !logprobs || !workspace) return fail(-1, "null pointer");
  if (B < 1 || Kft < 1 || B * Kft > 0x7fffffff) return fail(-1, "B*Kft out of range");
#define CALL(P)                                                                                                 \
  logprob_impl<P>(*actor, params, (const char*)packed, *cfg, ksteps, Kft, obs, chains, B, logprobs, workspace, workspace_bytes, \
                  (hipStream_t)stream)
  return DPPO_DISPATCH(prec, CALL);
#undef CALL
}

// ---- behaviour-cloning term ----------------------------------------------------------------------------
template <class P>
static size_t carve_bc(Carver& c, const dppo_net_desc& d, int64_t M, MlpBufs<P>& B, int32_t*& brow, int32_t*& krow,
                       double** partial = nullptr) {
  carve_mlp<P>(c, d, M, true, true, B);
  brow = (int32_t*)c.take((size_t)M * 4);
  krow = (int32_t*)c.take((size_t)M * 4);
  double* pp = (double*)c.take((size_t)(bc_loss_blocks(M, pack_layout<P>(d, 0).Kpo) + 1) * sizeof(double));
  if (partial) *partial = pp;
  return al256(c.off);
}
int64_t dppo_bc_loss_workspace_bytes(const dppo_net_desc* actor, int prec, int64_t B, int Kft) {
  if (check_net(actor) || check_prec(prec)) return -1;
  if (B < 1 || Kft < 1 || B * Kft > 0x7fffffff) return fail(-1, "B*Kft out of range");
  if (Kft > 1024 || (int64_t)Kft * actor->time_dim > 65536 ||
      time_backward_lds_bytes(Kft, actor->time_dim) > 156 * 1024)
    return fail(-1, "Kft * time_dim = %d too large for the time-embedding backward (LDS)", Kft * actor->time_dim);
  Carver c{nullptr, 0, 0};
  int32_t *br, *kr;
  if (prec == DPPO_PREC_F32) {
    MlpBufs<F32> W;
    return (int64_t)carve_bc<F32>(c, *actor, B * Kft, W, br, kr);
  }
  MlpBufs<BF16> W;
  return (int64_t)carve_bc<BF16>(c, *actor, B * Kft, W, br, kr);
}
template <class P>
static int bc_impl(const dppo_net_desc& d, const float* prm, const char* pk, const dppo_diffusion_cfg& cfg,
                   const dppo_step* ksteps, int Kft, const float* obs, const float* chains, int64_t Bn, float* grad,
                   double* loss, void* ws, int64_t wsb, hipStream_t s) {
  const int64_t M = Bn * Kft;
  Carver c{(char*)ws, 0, (size_t)wsb};
  MlpBufs<P> B;
  int32_t *brow, *krow;
  double* partial;
  const size_t need = carve_bc<P>(c, d, M, B, brow, krow, &partial);
  if ((int64_t)need > wsb) return fail(-1, "workspace too small: need %zu bytes, got %lld", need, (long long)wsb);
  const PackLayout L = pack_layout<P>(d, 0);
  BuildRows br;
  memset(&br, 0, sizeof(br));
  br.chains = chains, br.obs = obs, br.temb = (const float*)(pk + L.temb), br.ksteps = ksteps;
  br.Kft = Kft, br.AF = d.act_flat, br.td = d.time_dim, br.cond = d.cond_dim, br.M = M, br.obs_in_a = 1;
  br.inA = B.in, br.KpA = L.Kp0, br.brow = brow, br.krow = krow, br.onehot0 = temb_onehot_col<P>(d, L, Kft, B);
  if (d.cond_hidden > 0) br.obs_in_a = 0, br.inC = B.cin, br.KpC = L.Kpc;
  launch_build_rows<P>(br, s);
  if (d.cond_hidden > 0) cond_encode<P>(d, prm, pk, L, M, B.cin, B, B.in, nullptr, 0, true, s);
  mlp_forward<P>(d, prm, pk, L, M, B, true, s);
  BcArgs ba;
  ba.eps = B.out, ba.lde = B.ldout, ba.chains = chains, ba.ksteps = ksteps, ba.cfg = cfg, ba.Kft = Kft;
  ba.AF = d.act_flat, ba.M = M, ba.d_eps = B.d_out, ba.ldde = L.Kpo, ba.loss = loss, ba.partial = partial;
  launch_bc_loss<P>(ba, s);
  mlp_backward<P>(d, prm, pk, L, M, B, grad, krow, ksteps, Kft, s, false);
  if (d.cond_hidden > 0) cond_backward<P>(d, prm, pk, L, M, B, B.dh0_final, B.cin, grad, s);
  return check_launch();
}
int dppo_bc_loss_fwd_bwd(const dppo_net_desc* actor, int prec, const float* params, const void* packed,
                         const dppo_diffusion_cfg* cfg, const dppo_step* ksteps, int Kft, const float* obs,
                         const float* chains, int64_t B, float* grad, double* loss, void* workspace,
                         int64_t workspace_bytes, dppo_stream_t stream) {
  if (int e = check_net(actor)) return e;
  if (int e = check_prec(prec)) return e;
  if (actor->kind != 0) return fail(-1, "dppo_bc_loss_fwd_bwd needs an actor descriptor");
  if (!params || !packed || !cfg || !ksteps || !obs || !chains || !grad || !loss || !workspace) return fail(-1, "null pointer");
  if (B < 1 || Kft < 1 || B * Kft > 0x7fffffff) return fail(-1, "B*Kft out of range");
  if (Kft > 1024) return fail(-1, "Kft out of range");
  if ((int64_t)Kft * actor->time_dim > 65536 || time_backward_lds_bytes(Kft, actor->time_dim) > 156 * 1024)
    return fail(-1, "Kft * time_dim = %d too large for the time-embedding backward (LDS)", Kft * actor->time_dim);
#define CALL(P)                                                                                                       \
  bc_impl<P>(*actor, params, (const char*)packed, *cfg, ksteps, Kft, obs, chains, B, grad, loss, workspace, workspace_bytes, \
             (hipStream_t)stream)
  return DPPO_DISPATCH(prec, CALL);
#undef CALL
}
int dppo_axpy(float* y, const float* x, double alpha, int64_t n, dppo_stream_t stream) {
  if (!y || !x || n < 0) return fail(-1, "bad argument");
  launch_axpy(y, x, (float)alpha, n, (hipStream_t)stream);
  return check_launch();
}

// ---- supervised denoising loss (pre-training) ------------------------------------------------------------
int64_t dppo_denoise_mse_workspace_bytes(const dppo_net_desc* actor, int prec, int64_t N) {
  return dppo_bc_loss_workspace_bytes(actor, prec, N, 1);
}
template <class P>
static int mse_impl(const dppo_net_desc& d, const float* prm, const char* pk, const dppo_step* tsteps, int n_time,
                    const float* obs, const float* pairs, const int64_t* kinds, int64_t M, float* grad, double* loss,
                    void* ws, int64_t wsb, hipStream_t s, float* d_obs = nullptr) {
  Carver c{(char*)ws, 0, (size_t)wsb};
  MlpBufs<P> B;
  int32_t *brow, *krow;
  double* partial;
  const size_t need = carve_bc<P>(c, d, M, B, brow, krow, &partial);
  if ((int64_t)need > wsb) return fail(-1, "workspace too small: need %zu bytes, got %lld", need, (long long)wsb);
  const PackLayout L = pack_layout<P>(d, 0);
  BuildRows br;  // gathered mode: row n = (pairs[n][0], temb(tsteps[kinds[n]].t), obs[n])
  memset(&br, 0, sizeof(br));
  br.kinds = kinds, br.chains = pairs, br.obs = obs, br.temb = (const float*)(pk + L.temb), br.ksteps = tsteps;
  br.Kft = n_time, br.AF = d.act_flat, br.td = d.time_dim, br.cond = d.cond_dim, br.M = M, br.obs_in_a = 1;
  br.inA = B.in, br.KpA = L.Kp0, br.brow = brow, br.krow = krow, br.onehot0 = temb_onehot_col<P>(d, L, n_time, B);
  if (d.cond_hidden > 0) br.obs_in_a = 0, br.inC = B.cin, br.KpC = L.Kpc;
  launch_build_rows<P>(br, s);
  if (d.cond_hidden > 0) cond_encode<P>(d, prm, pk, L, M, B.cin, B, B.in, nullptr, 0, true, s);
  mlp_forward<P>(d, prm, pk, L, M, B, true, s);
  MseArgs ma;
  ma.eps = B.out, ma.lde = B.ldout, ma.pairs = pairs, ma.AF = d.act_flat, ma.M = M, ma.d_eps = B.d_out, ma.ldde = L.Kpo;
  ma.loss = loss, ma.partial = partial;
  launch_mse_loss<P>(ma, s);
  mlp_backward<P>(d, prm, pk, L, M, B, grad, krow, tsteps, n_time, s, false);
  if (d.cond_hidden > 0) cond_backward<P>(d, prm, pk, L, M, B, B.dh0_final, B.cin, grad, s);
  if (d_obs) obs_grad<P>(d, prm, M, B, d_obs, s);
  return check_launch();
}
static int mse_entry(const dppo_net_desc* actor, int prec, const float* params, const void* packed,
                             const dppo_step* tsteps, int n_time, const float* obs, const float* pairs,
                             const int64_t* kinds, int64_t N, float* grad, double* loss, void* workspace,
                             int64_t workspace_bytes, dppo_stream_t stream, float* d_obs) {
  if (int e = check_net(actor)) return e;
  if (int e = check_prec(prec)) return e;
  if (actor->kind != 0) return fail(-1, "dppo_denoise_mse_fwd_bwd needs an actor descriptor");
  if (d_obs && actor->cond_hidden > 0) return fail(-1, "d_obs with a cond_mlp actor is not built");
  if (!params || !packed || !tsteps || !obs || !pairs || !kinds || !grad || !loss || !workspace)
    return fail(-1, "null pointer");
  if (N < 1 || N > 0x7fffffff || n_time < 1 || n_time > 1024) return fail(-1, "N / n_time out of range");
  if (time_backward_lds_bytes(n_time, actor->time_dim) > 156 * 1024)
    return fail(-1, "n_time * time_dim = %d too large for the time-embedding backward (LDS)", n_time * actor->time_dim);
#define CALL(P)                                                                                                        \
  mse_impl<P>(*actor, params, (const char*)packed, tsteps, n_time, obs, pairs, kinds, N, grad, loss, workspace, \
              workspace_bytes, (hipStream_t)stream, d_obs)
  return DPPO_DISPATCH(prec, CALL);
#undef CALL
}
int dppo_denoise_mse_fwd_bwd(const dppo_net_desc* actor, int prec, const float* params, const void* packed,
                             const dppo_step* tsteps, int n_time, const float* obs, const float* pairs,
                             const int64_t* kinds, int64_t N, float* grad, double* loss, void* workspace,
                             int64_t workspace_bytes, dppo_stream_t stream) {
  return mse_entry(actor, prec, params, packed, tsteps, n_time, obs, pairs, kinds, N, grad, loss, workspace, workspace_bytes,
                   stream, nullptr);
}
int dppo_denoise_mse_fwd_bwd_obs(const dppo_net_desc* actor, int prec, const float* params, const void* packed,
                                 const dppo_step* tsteps, int n_time, const float* obs, const float* pairs,
                                 const int64_t* kinds, int64_t N, float* grad, double* loss, void* workspace,
                                 int64_t workspace_bytes, dppo_stream_t stream, float* d_obs) {
  if (!d_obs) return fail(-1, "null pointer");
  return mse_entry(actor, prec, params, packed, tsteps, n_time, obs, pairs, kinds, N, grad, loss, workspace, workspace_bytes,
                   stream, d_obs);
}

// ---- GAE ---------------------------------------------------------------------------------------------
int dppo_gae(const double* reward, const float* values, const float* terminated, const float* last_values, int n_steps,
             int n_envs, double gamma, double gae_lambda, double reward_scale_const, double* adv64, double* ret64,
             float* adv32, float* ret32, dppo_stream_t stream) {
  if (!reward || !values || !terminated || !last_values) return fail(-1, "null pointer");
  if (n_steps < 1 || n_envs < 1) return fail(-1, "empty rollout");
  launch_gae(reward, values, terminated, last_values, n_steps, n_envs, gamma, gae_lambda, reward_scale_const, adv64,
             ret64, adv32, ret32, (hipStream_t)stream);
  return check_launch();
}

// ---- PPO update --------------------------------------------------------------------------------------
template <class P>
struct PpoWs {
  MlpBufs<P> A, C;
  int32_t *brow, *krow;
  int32_t* brow_c;  // the critic pipeline's own copy of brow (it builds its rows on its own stream)
  double* moments;
  float* loss_tab;  // [2 Kft], Kft <= 1024
  double* loss_partial;
  double* loss_partial_v;  // the value half's, when it runs on the critic's stream
};
template <class P>
static size_t carve_ppo(Carver& c, const dppo_net_desc& a, const dppo_net_desc& cr, int64_t N, PpoWs<P>& W) {
  W.moments = (double*)c.take((8 + 2 * ADV_MOMENT_BLOCKS) * sizeof(double));
  W.loss_tab = (float*)c.take(2 * 1024 * sizeof(float));
  W.loss_partial = (double*)c.take((size_t)loss_blocks(N) * 8 * sizeof(double));
  W.loss_partial_v = (double*)c.take((size_t)loss_blocks(N) * 8 * sizeof(double));
  W.brow = (int32_t*)c.take((size_t)N * 4);
  W.brow_c = (int32_t*)c.take((size_t)N * 4);
  W.krow = (int32_t*)c.take((size_t)N * 4);
  carve_mlp<P>(c, a, N, true, true, W.A);
  carve_mlp<P>(c, cr, N, true, true, W.C);
  return al256(c.off);
}
int64_t dppo_ppo_workspace_bytes(const dppo_net_desc* actor, const dppo_net_desc* critic, int prec, int64_t N) {
  if (check_net(actor) || check_net(critic) || check_prec(prec)) return -1;
  if (N < 2 || N > 0x7fffffff) return fail(-1, "N out of range");
  Carver c{nullptr, 0, 0};
  if (prec == DPPO_PREC_F32) {
    PpoWs<F32> W;
    return (int64_t)carve_ppo<F32>(c, *actor, *critic, N, W);
  }
  PpoWs<BF16> W;
  return (int64_t)carve_ppo<BF16>(c, *actor, *critic, N, W);
}

template <class P>
static int ppo_impl(const dppo_net_desc& a, const dppo_net_desc& cr, const float* ap, const char* ak, const float* cp,
                    const char* ck, const dppo_diffusion_cfg& dcfg, const dppo_ppo_cfg& pcfg, const dppo_step* ksteps,
                    const float* obs_k, const float* chains_k, const float* returns_k, const float* values_k,
                    const float* adv_k, const float* logprobs_k, const int64_t* inds, const int64_t* kinds, int64_t N,
                    const double* gmom, float* agrad, float* cgrad, double* stats, void* ws, int64_t wsb, hipStream_t s,
                    const dppo_obs_io* oio = nullptr, const dppo_dp_hook* hook = nullptr) {
  Carver c{(char*)ws, 0, (size_t)wsb};
  PpoWs<P> W;
  const size_t need = carve_ppo<P>(c, a, cr, N, W);
  if ((int64_t)need > wsb) return fail(-1, "workspace too small: need %zu bytes, got %lld", need, (long long)wsb);
  const PackLayout LA = pack_layout<P>(a, 0), LC = pack_layout<P>(cr, 0);
  const int Kft = pcfg.ft_denoising_steps;
  W.A.allow_frag = frag_ok<P>(a, N, LA, Kft, W.A, oio && oio->d_obs_actor);
  W.C.allow_frag = frag_ok<P>(cr, N, LC, Kft, W.C, oio && oio->d_obs_critic);
  // The critic pipeline (rows -> forward -> value loss -> backward -> weight gradients) and the actor pipeline (rows ->
  // advantage moments -> forward -> policy loss -> ...) share nothing but the call's inputs: one fork at entry, one join
  // at the end.  A cross-stream event hop costs 10-17 us of device idle time; at entry it hides behind the actor's row
  // builder, and the loss is evaluated as two launches rather than joining the streams in the middle of the call.
  // (With a cond_mlp the actor's encoder reads the critic's observation rows: the row builder then stays one launch.)
  const bool own_rows = a.cond_hidden == 0;
  hipStream_t s2 = own_rows ? fork_side(s) : s;
  BuildRows br;
  memset(&br, 0, sizeof(br));
  br.zero_b = W.moments, br.n_zero_b = 32;  // zeroed by the row builder (every statistic has one owner launch that writes it)
  // post_reduce_kernel's arrival counter and, behind it, the tile counters of the folded slab reduction (both networks')
  br.zero_a = W.A.post_counter, br.n_zero_a = 1 + RED_CNT / 2, W.A.post_zeroed = true, W.A.fold = true;
  br.zero_c = W.C.post_counter, br.n_zero_c = 1 + RED_CNT / 2, W.C.fold = true;
  W.A.dw0 = !W.A.allow_frag && dw0_ok<P>(a, N, LA, Kft, W.A, oio && oio->d_obs_actor);
  W.C.dw0 = !W.C.allow_frag && dw0_ok<P>(cr, N, LC, Kft, W.C, oio && oio->d_obs_critic);
  if (Kft <= 1024) br.loss_tab = W.loss_tab, br.pcfg = pcfg;
  br.inds = inds, br.kinds = kinds, br.chains = chains_k, br.obs = obs_k, br.temb = (const float*)(ak + LA.temb);
  br.ksteps = ksteps;
  br.Kft = Kft, br.AF = a.act_flat, br.td = a.time_dim, br.cond = a.cond_dim, br.M = N;
  br.inA = W.A.in, br.KpA = LA.Kp0, br.inC = W.C.in, br.KpC = LC.Kp0, br.brow = W.brow, br.krow = W.krow;
  br.obs_in_a = a.cond_hidden > 0 ? 0 : 1;  // with cond_mlp the encoder fills the state columns (from the critic's obs rows)
  br.onehot0 = temb_onehot_col<P>(a, LA, Kft, W.A);
  // advantage moments as riders of the actor's row builder (knob 36): no launch between the rows and the actor's forward
  const bool mom_rider = gmom == nullptr && (g_mom_rider == 1 || (g_mom_rider == 0 && N <= MOM_RIDER_MAX_N));
  if (mom_rider) br.mom_adv = adv_k, br.mom_out = W.moments, br.n_zero_b = 8;  // (block 0 must not zero the riders' slots, [8, ...))
  const float* obs_c = oio && oio->obs_critic ? oio->obs_critic : nullptr;  // the critic's own observation rows (pixel nets)
  const bool split = s2 != s || obs_c != nullptr;
  if (split) {
    BuildRows bc = br;  // critic rows only, on the critic's stream
    bc.zero_a = bc.zero_b = nullptr, bc.n_zero_a = bc.n_zero_b = 0, bc.loss_tab = nullptr, bc.mom_adv = nullptr;
    br.zero_c = nullptr, br.n_zero_c = 0;  // (the critic's row builder zeroes the critic's counters: its own stream's order)
    bc.inA = nullptr, bc.brow = W.brow_c, bc.krow = nullptr;
    if (obs_c) bc.obs = obs_c;
    bc.cond = cr.cond_dim;
    launch_build_rows<P>(bc, s2);
    br.inC = nullptr;
  }
  launch_build_rows<P>(br, s);
  if (a.cond_hidden > 0) cond_encode<P>(a, ap, ak, LA, N, W.C.in, W.A, W.A.in, nullptr, 0, true, s);
  if (gmom == nullptr && !mom_rider) launch_adv_moments(adv_k, W.brow, N, W.moments, s);
  if (!own_rows) s2 = fork_side(s);
  LossArgs la;
  memset(&la, 0, sizeof(la));
  la.eps = W.A.out, la.lde = W.A.ldout, la.vnew = W.C.out, la.ldv = W.C.ldout, la.brow = W.brow, la.krow = W.krow;
  la.gathered = kinds != nullptr;
  la.chains = chains_k, la.logprobs_k = logprobs_k, la.returns_k = returns_k, la.values_k = values_k, la.adv_k = adv_k;
  la.ksteps = ksteps, la.dcfg = dcfg, la.pcfg = pcfg, la.AF = a.act_flat, la.N = N;
  la.moments = gmom ? gmom : W.moments;
  if (mom_rider) la.mom_blocks = ADV_RIDER_BLOCKS, la.moments_out = W.moments;
  la.tab = Kft <= 1024 ? W.loss_tab : nullptr;
  la.d_eps = W.A.d_out, la.ldde = LA.Kpo, la.d_v = W.C.d_out, la.lddv = LC.Kpo, la.stats = stats;
  const bool fuse_bout = false;  // out-layer bias gradients come from the fused backward's d_out column sums
  const bool two_streams = s2 != s;
  // critic half.  (Knob 10 gates its first persistent kernel so that it becomes eligible together with the actor's instead
  // of starting alone and taking every CU first; measured worse, off.)
  const bool actor_first = split && g_gate_critic == 2;  // experiment: the critic's forward waits for the actor's
  // the policy half of the loss in the actor forward's epilogue (knob 39): its arguments as the policy launch would get them
  LossArgs lpol = la;
  lpol.part = 1, lpol.partial = W.loss_partial;
  const bool fuse_loss = fuse_loss_ok<P>(a, lpol, W.A, two_streams, mom_rider);
  if (actor_first) {
    mlp_forward<P>(a, ap, ak, LA, N, W.A, true, s, fuse_loss ? &lpol : nullptr);
    gate_side(s, s2);
  } else if (split && g_gate_critic) {
    gate_side(s, s2);
  }
  mlp_forward<P>(cr, cp, ck, LC, N, W.C, true, s2);
  if (two_streams) {
    la.part = 2, la.partial = W.loss_partial_v;
    if (split) la.brow = W.brow_c;
    if (gmom == nullptr) la.n_count = (double)N;  // = what adv_moments leaves in moments[2], without waiting for it
    launch_ppo_loss<P>(la, s2);
    const LossArgs lv = la;
    la.brow = W.brow, la.n_count = 0;
    // (no tail stream of its own: a fork from a forked stream crashes hipGraph capture on ROCm 7.0 at capture end, and
    // the critic's tail is one 10-us reduction)
    mlp_backward<P>(cr, cp, ck, LC, N, W.C, cgrad, nullptr, nullptr, 0, s2, fuse_bout, -1, &lv);
    if (oio && oio->d_obs_critic) obs_grad<P>(cr, cp, N, W.C, oio->d_obs_critic, s2);
    // data parallel: everything that writes critic_grad is enqueued on s2 -- the caller queues the critic slice's
    // all-reduce behind it THERE, so that it runs while the actor's forward / backward still occupy the main stream
    if (hook && hook->critic_grads_enqueued) hook->critic_grads_enqueued(hook->user, (dppo_stream_t)s2);
  }
  // actor half
  if (!actor_first) mlp_forward<P>(a, ap, ak, LA, N, W.A, true, s, fuse_loss ? &lpol : nullptr);
  la.part = two_streams ? 1 : 3, la.partial = W.loss_partial;
  if (!fuse_loss) launch_ppo_loss<P>(la, s);
  if (!two_streams) {
    mlp_backward<P>(cr, cp, ck, LC, N, W.C, cgrad, nullptr, nullptr, 0, s, fuse_bout, -1);
    if (oio && oio->d_obs_critic) obs_grad<P>(cr, cp, N, W.C, oio->d_obs_critic, s);
    if (hook && hook->critic_grads_enqueued) hook->critic_grads_enqueued(hook->user, (dppo_stream_t)s);
  }
  if (two_streams && g_early_join) W.A.join_s[W.A.n_join] = s2, W.A.join_idx[W.A.n_join++] = 0;
  mlp_backward<P>(a, ap, ak, LA, N, W.A, agrad, W.krow, ksteps, Kft, s, fuse_bout, 1, &la);
  if (a.cond_hidden > 0) cond_backward<P>(a, ap, ak, LA, N, W.A, W.A.dh0_final, W.C.in, agrad, s);
  if (oio && oio->d_obs_actor) obs_grad<P>(a, ap, N, W.A, oio->d_obs_actor, s);
  if (!(two_streams && g_early_join) || W.A.n_join > 0) join_side(s, s2);  // (n_join > 0: the backward never flushed)
  W.A.n_join = 0;
  return check_launch();
}

static bool hipStreamIsCapturing_safe(hipStream_t s) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  return hipStreamIsCapturing(s, &st) == hipSuccess && st != hipStreamCaptureStatusNone;
}
static int check_obs_io(const dppo_obs_io* io, const int64_t* kinds, bool actor_has_encoder) {
  if (!io) return 0;
  if (kinds == nullptr) return fail(-1, "the _obs entries take pre-gathered samples (kinds), one observation row per sample");
  if (actor_has_encoder && (io->obs_critic || io->d_obs_actor)) return fail(-1, "d_obs / obs_critic with a cond_mlp actor is not built");
  return 0;
}
static int ppo_entry(const dppo_net_desc* actor, const dppo_net_desc* critic, int prec, const float* actor_params,
                          const void* actor_packed, const float* critic_params, const void* critic_packed,
                          const dppo_diffusion_cfg* dcfg, const dppo_ppo_cfg* pcfg, const dppo_step* ksteps,
                          const float* obs_k, const float* chains_k, const float* returns_k, const float* values_k,
                          const float* adv_k, const float* logprobs_k, const int64_t* inds, const int64_t* kinds,
                          int64_t N, const double* global_moments, float* actor_grad, float* critic_grad,
                          double* stats, void* workspace, int64_t workspace_bytes, dppo_stream_t stream, const dppo_obs_io* io,
                          const dppo_dp_hook* hook = nullptr) {
  if (int e = check_net(actor)) return e;
  if (int e = check_net(critic)) return e;
  if (int e = check_prec(prec)) return e;
  if (actor->kind != 0 || critic->kind != 1 || critic->out_dim != 1)
    return fail(-1, "descriptor kinds must be (actor, critic with out_dim 1)");
  if (actor->cond_dim != critic->cond_dim && !(io && io->obs_critic))
    return fail(-1, "actor and critic observe different cond_dim");
  if (!actor_params || !actor_packed || !critic_params || !critic_packed || !dcfg || !pcfg || !ksteps || !obs_k ||
      !chains_k || !returns_k || !values_k || !adv_k || !logprobs_k || !actor_grad || !critic_grad || !stats ||
      !workspace)
    return fail(-1, "null pointer");
  if (int e = check_obs_io(io, kinds, actor->cond_hidden > 0)) return e;
  if ((inds == nullptr) == (kinds == nullptr)) return fail(-1, "pass exactly one of inds (rollout mode) / kinds (gathered mode)");
  if (N < 2 || N > 0x7fffffff) return fail(-1, "N out of range");
  if (pcfg->ft_denoising_steps < 1 || pcfg->ft_denoising_steps > 1024) return fail(-1, "Kft out of range");
  if ((int64_t)pcfg->ft_denoising_steps * actor->time_dim > 65536 ||
      time_backward_lds_bytes(pcfg->ft_denoising_steps, actor->time_dim) > 156 * 1024)
    return fail(-1, "Kft * time_dim too large");
  if (pcfg->horizon_steps * pcfg->action_dim != actor->act_flat) return fail(-1, "Ta*Da != act_flat");
  if (pcfg->reward_horizon < 1) return fail(-1, "reward_horizon must be >= 1");
#define CALL(P)                                                                                                        \
  ppo_impl<P>(*actor, *critic, actor_params, (const char*)actor_packed, critic_params, (const char*)critic_packed, *dcfg,     \
              *pcfg, ksteps, obs_k, chains_k, returns_k, values_k, adv_k, logprobs_k, inds, kinds, N, global_moments, actor_grad, critic_grad, stats, \
              workspace, workspace_bytes, (hipStream_t)stream, io, hook)
  return DPPO_DISPATCH(prec, CALL);
#undef CALL
}
int dppo_ppo_loss_fwd_bwd_dp(const dppo_net_desc* actor, const dppo_net_desc* critic, int prec, const float* actor_params,
                             const void* actor_packed, const float* critic_params, const void* critic_packed,
                             const dppo_diffusion_cfg* dcfg, const dppo_ppo_cfg* pcfg, const dppo_step* ksteps,
                             const float* obs_k, const float* chains_k, const float* returns_k, const float* values_k,
                             const float* adv_k, const float* logprobs_k, const int64_t* inds, const int64_t* kinds,
                             int64_t N, const double* global_moments, float* actor_grad, float* critic_grad,
                             double* stats, void* workspace, int64_t workspace_bytes, dppo_stream_t stream,
                             const dppo_dp_hook* hook) {
  if (hipStreamIsCapturing_safe((hipStream_t)stream) && hook && hook->critic_grads_enqueued)
    return fail(-1, "dppo_ppo_loss_fwd_bwd_dp: a collective cannot be queued from inside a stream capture");
  return ppo_entry(actor, critic, prec, actor_params, actor_packed, critic_params, critic_packed, dcfg, pcfg, ksteps, obs_k,
                   chains_k, returns_k, values_k, adv_k, logprobs_k, inds, kinds, N, global_moments, actor_grad, critic_grad,
                   stats, workspace, workspace_bytes, stream, nullptr, hook);
}
int dppo_ppo_loss_fwd_bwd(const dppo_net_desc* actor, const dppo_net_desc* critic, int prec, const float* actor_params,
                          const void* actor_packed, const float* critic_params, const void* critic_packed,
                          const dppo_diffusion_cfg* dcfg, const dppo_ppo_cfg* pcfg, const dppo_step* ksteps,
                          const float* obs_k, const float* chains_k, const float* returns_k, const float* values_k,
                          const float* adv_k, const float* logprobs_k, const int64_t* inds, const int64_t* kinds,
                          int64_t N, const double* global_moments, float* actor_grad, float* critic_grad,
                          double* stats, void* workspace, int64_t workspace_bytes, dppo_stream_t stream) {
  return ppo_entry(actor, critic, prec, actor_params, actor_packed, critic_params, critic_packed, dcfg, pcfg, ksteps, obs_k,
                   chains_k, returns_k, values_k, adv_k, logprobs_k, inds, kinds, N, global_moments, actor_grad, critic_grad,
                   stats, workspace, workspace_bytes, stream, nullptr);
}
int dppo_ppo_loss_fwd_bwd_obs(const dppo_net_desc* actor, const dppo_net_desc* critic, int prec, const float* actor_params,
                              const void* actor_packed, const float* critic_params, const void* critic_packed,
                              const dppo_diffusion_cfg* dcfg, const dppo_ppo_cfg* pcfg, const dppo_step* ksteps,
                              const float* obs_k, const float* chains_k, const float* returns_k, const float* values_k,
                              const float* adv_k, const float* logprobs_k, const int64_t* kinds, int64_t N,
                              const double* global_moments, float* actor_grad, float* critic_grad, double* stats,
                              void* workspace, int64_t workspace_bytes, dppo_stream_t stream, const dppo_obs_io* io) {
  if (!io) return fail(-1, "null pointer");
  return ppo_entry(actor, critic, prec, actor_params, actor_packed, critic_params, critic_packed, dcfg, pcfg, ksteps, obs_k,
                   chains_k, returns_k, values_k, adv_k, logprobs_k, nullptr, kinds, N, global_moments, actor_grad, critic_grad,
                   stats, workspace, workspace_bytes, stream, io);
}

// ---- Gaussian-policy PPO (gaussian.hip) ----------------------------------------------------------------------
static int check_gauss(const dppo_net_desc* actor, const dppo_gaussian_cfg* cfg, const float* logvar) {
  if (int e = check_net(actor)) return e;
  if (actor->kind != 1) return fail(-1, "the Gaussian actor is a kind-1 (observation trunk) descriptor");
  if (!cfg) return fail(-1, "null cfg");
  if (cfg->horizon_steps * cfg->action_dim != actor->out_dim) return fail(-1, "Ta*Da != actor out_dim");
  if (cfg->std_mode != 0 && cfg->std_mode != 1) return fail(-1, "std_mode must be 0 (fixed) or 1 (learned per dimension)");
  if (cfg->std_mode == 1 && !logvar) return fail(-1, "std_mode 1 needs logvar");
  if (cfg->std_mode == 0 && !(cfg->fixed_std > 0)) return fail(-1, "fixed_std must be positive");
  return 0;
}
template <class P>
static size_t carve_gauss(Carver& c, const dppo_net_desc& a, const dppo_net_desc* cr, int64_t N, bool train, MlpBufs<P>& A,
                          MlpBufs<P>& Cb, double*& moments, double*& scratch, double*& partial) {
  moments = (double*)c.take(4 * sizeof(double));
  scratch = (double*)c.take(2 * 64 * sizeof(double));
  partial = (double*)c.take((size_t)gauss_blocks(N) * (8 + a.out_dim) * sizeof(double));
  carve_mlp<P>(c, a, N, train, train, A);
  if (cr) carve_mlp<P>(c, *cr, N, train, train, Cb);
  return al256(c.off);
}
int64_t dppo_gaussian_workspace_bytes(const dppo_net_desc* actor, const dppo_net_desc* critic, int prec, int64_t N) {
  if (check_net(actor) || (critic && check_net(critic)) || check_prec(prec)) return -1;
  if (N < 1 || N > 0x7fffffff) return fail(-1, "N out of range");
  Carver c{nullptr, 0, 0};
  double *m, *sc, *pa;
  if (prec == DPPO_PREC_F32) {
    MlpBufs<F32> A, Cb;
    return (int64_t)carve_gauss<F32>(c, *actor, critic, N, critic != nullptr, A, Cb, m, sc, pa);
  }
  MlpBufs<BF16> A, Cb;
  return (int64_t)carve_gauss<BF16>(c, *actor, critic, N, critic != nullptr, A, Cb, m, sc, pa);
}
template <class P>
static int gauss_infer_impl(const dppo_net_desc& d, const float* prm, const char* pk, const dppo_gaussian_cfg& cfg,
                            const float* logvar, const float* obs, const float* noise, const float* actions, int64_t N,
                            float* out_actions, float* out_mean, float* out_logp, void* ws, int64_t wsb, hipStream_t s) {
  Carver c{(char*)ws, 0, (size_t)wsb};
  MlpBufs<P> A, Cb;
  double *m, *sc, *pa;
  const size_t need = carve_gauss<P>(c, d, nullptr, N, false, A, Cb, m, sc, pa);
  if (!ws || (int64_t)need > wsb) return fail(-1, "workspace too small: need %zu bytes, got %lld", need, (long long)wsb);
  const PackLayout L = pack_layout<P>(d, 0);
  launch_build_direct<P>(nullptr, nullptr, obs, nullptr, 0, 0, d.cond_dim, N, A.in, L.Kp0, s);
  mlp_forward<P>(d, prm, pk, L, N, A, false, s);
  GaussArgs g;
  memset(&g, 0, sizeof(g));
  g.cfg = cfg, g.mean_pre = A.out, g.ldm = A.ldout, g.logvar = logvar, g.N = N, g.AF = d.out_dim;
  if (out_actions) {
    g.noise = noise, g.out_actions = out_actions, g.out_mean = out_mean;
    launch_gauss_sample(g, s);
  } else {
    g.actions = actions, g.out_logp = out_logp;
    launch_gauss_logprob(g, s);
  }
  return check_launch();
}
int dppo_gaussian_sample(const dppo_net_desc* actor, int prec, const float* params, const void* packed,
                         const dppo_gaussian_cfg* cfg, const float* logvar, const float* obs, const float* noise,
                         int64_t B, float* actions, float* mean_out, void* workspace, int64_t workspace_bytes,
                         dppo_stream_t stream) {
  if (int e = check_gauss(actor, cfg, logvar)) return e;
  if (int e = check_prec(prec)) return e;
  if (!params || !packed || !obs || !actions) return fail(-1, "null pointer");
  if (B < 1 || B > 0x7fffffff) return fail(-1, "B out of range");
#define CALL(P)                                                                                                        \
  gauss_infer_impl<P>(*actor, params, (const char*)packed, *cfg, logvar, obs, noise, nullptr, B, actions, mean_out, nullptr, \
                      workspace, workspace_bytes, (hipStream_t)stream)
  return DPPO_DISPATCH(prec, CALL);
#undef CALL
}
int dppo_gaussian_logprob(const dppo_net_desc* actor, int prec, const float* params, const void* packed,
                          const dppo_gaussian_cfg* cfg, const float* logvar, const float* obs, const float* actions,
                          int64_t N, float* logp, void* workspace, int64_t workspace_bytes, dppo_stream_t stream) {
  if (int e = check_gauss(actor, cfg, logvar)) return e;
  if (int e = check_prec(prec)) return e;
  if (!params || !packed || !obs || !actions || !logp) return fail(-1, "null pointer");
  if (N < 1 || N > 0x7fffffff) return fail(-1, "N out of range");
#define CALL(P)                                                                                                        \
  gauss_infer_impl<P>(*actor, params, (const char*)packed, *cfg, logvar, obs, nullptr, actions, N, nullptr, nullptr, logp, \
                      workspace, workspace_bytes, (hipStream_t)stream)
  return DPPO_DISPATCH(prec, CALL);
#undef CALL
}
template <class P>
static int gauss_ppo_impl(const dppo_net_desc& a, const dppo_net_desc& cr, const float* ap, const char* ak, const float* cp,
                          const char* ck, const dppo_gaussian_cfg& cfg, const float* logvar, const float* obs,
                          const float* actions, const float* returns, const float* oldvalues, const float* adv,
                          const float* oldlogp, int64_t N, const double* gmom, float* agrad, float* cgrad, float* lvgrad,
                          double* stats, void* ws, int64_t wsb, hipStream_t s, const dppo_obs_io* oio = nullptr) {
  Carver c{(char*)ws, 0, (size_t)wsb};
  MlpBufs<P> A, Cb;
  double *moments, *scratch, *partial;
  const size_t need = carve_gauss<P>(c, a, &cr, N, true, A, Cb, moments, scratch, partial);
  if ((int64_t)need > wsb) return fail(-1, "workspace too small: need %zu bytes, got %lld", need, (long long)wsb);
  const PackLayout LA = pack_layout<P>(a, 0), LC = pack_layout<P>(cr, 0);
  // the critic pipeline on the side stream beside the actor's forward (one fork here, one join at the end)
  hipStream_t s2 = fork_side(s);
  launch_build_direct<P>(nullptr, nullptr, oio && oio->obs_critic ? oio->obs_critic : obs, nullptr, 0, 0, cr.cond_dim, N, Cb.in,
                         LC.Kp0, s2);
  mlp_forward<P>(cr, cp, ck, LC, N, Cb, true, s2);
  launch_build_direct<P>(nullptr, nullptr, obs, nullptr, 0, 0, a.cond_dim, N, A.in, LA.Kp0, s);
  if (gmom == nullptr) launch_gauss_moments(adv, N, moments, scratch, s);
  mlp_forward<P>(a, ap, ak, LA, N, A, true, s);
  if (s2 != s) join_side(s, s2);  // the loss reads both outputs
  GaussArgs g;
  memset(&g, 0, sizeof(g));
  g.cfg = cfg, g.mean_pre = A.out, g.ldm = A.ldout, g.logvar = logvar, g.actions = actions, g.N = N, g.AF = a.out_dim;
  g.vnew = Cb.out, g.ldv = Cb.ldout, g.returns = returns, g.oldvalues = oldvalues, g.adv = adv, g.oldlogp = oldlogp;
  g.moments = gmom ? gmom : moments, g.d_mean = A.d_out, g.lddm = LA.Kpo, g.d_v = Cb.d_out, g.lddv = LC.Kpo;
  g.partial = partial, g.stats = stats, g.logvar_grad = lvgrad;
  launch_gauss_loss<P>(g, s);
  s2 = fork_side(s);
  mlp_backward<P>(cr, cp, ck, LC, N, Cb, cgrad, nullptr, nullptr, 0, s2, false, -1);
  if (oio && oio->d_obs_critic) obs_grad<P>(cr, cp, N, Cb, oio->d_obs_critic, s2);
  mlp_backward<P>(a, ap, ak, LA, N, A, agrad, nullptr, nullptr, 0, s, false, 1);
  if (oio && oio->d_obs_actor) obs_grad<P>(a, ap, N, A, oio->d_obs_actor, s);
  if (s2 != s) join_side(s, s2);
  return check_launch();
}
static int gauss_ppo_entry(const dppo_net_desc* actor, const dppo_net_desc* critic, int prec,
                                   const float* actor_params, const void* actor_packed, const float* critic_params,
                                   const void* critic_packed, const dppo_gaussian_cfg* cfg, const float* logvar,
                                   const float* obs, const float* actions, const float* returns,
                                   const float* oldvalues, const float* adv, const float* oldlogp, int64_t N,
                                   const double* global_moments, float* actor_grad, float* critic_grad,
                                   float* logvar_grad, double* stats, void* workspace, int64_t workspace_bytes,
                                   dppo_stream_t stream, const dppo_obs_io* io) {
  if (int e = check_gauss(actor, cfg, logvar)) return e;
  if (int e = check_net(critic)) return e;
  if (int e = check_prec(prec)) return e;
  if (critic->kind != 1 || critic->out_dim != 1) return fail(-1, "critic descriptor must be kind 1 with out_dim 1");
  if (actor->cond_dim != critic->cond_dim && !(io && io->obs_critic))
    return fail(-1, "actor and critic observe different cond_dim");
  if (!actor_params || !actor_packed || !critic_params || !critic_packed || !obs || !actions || !returns || !oldvalues ||
      !adv || !oldlogp || !actor_grad || !critic_grad || !stats || !workspace)
    return fail(-1, "null pointer");
  if (cfg->std_mode == 1 && !logvar_grad) return fail(-1, "std_mode 1 needs logvar_grad");
  if (N < 2 || N > 0x7fffffff) return fail(-1, "N out of range");
#define CALL(P)                                                                                                          \
  gauss_ppo_impl<P>(*actor, *critic, actor_params, (const char*)actor_packed, critic_params, (const char*)critic_packed, *cfg, \
                    logvar, obs, actions, returns, oldvalues, adv, oldlogp, N, global_moments, actor_grad, critic_grad,       \
                    logvar_grad, stats, workspace, workspace_bytes, (hipStream_t)stream, io)
  return DPPO_DISPATCH(prec, CALL);
#undef CALL
}
int dppo_gaussian_ppo_loss_fwd_bwd(const dppo_net_desc* actor, const dppo_net_desc* critic, int prec,
                                   const float* actor_params, const void* actor_packed, const float* critic_params,
                                   const void* critic_packed, const dppo_gaussian_cfg* cfg, const float* logvar,
                                   const float* obs, const float* actions, const float* returns,
                                   const float* oldvalues, const float* adv, const float* oldlogp, int64_t N,
                                   const double* global_moments, float* actor_grad, float* critic_grad,
                                   float* logvar_grad, double* stats, void* workspace, int64_t workspace_bytes,
                                   dppo_stream_t stream) {
  return gauss_ppo_entry(actor, critic, prec, actor_params, actor_packed, critic_params, critic_packed, cfg, logvar, obs, actions,
                         returns, oldvalues, adv, oldlogp, N, global_moments, actor_grad, critic_grad, logvar_grad, stats,
                         workspace, workspace_bytes, stream, nullptr);
}
int dppo_gaussian_ppo_loss_fwd_bwd_obs(const dppo_net_desc* actor, const dppo_net_desc* critic, int prec,
                                       const float* actor_params, const void* actor_packed, const float* critic_params,
                                       const void* critic_packed, const dppo_gaussian_cfg* cfg, const float* logvar,
                                       const float* obs, const float* actions, const float* returns,
                                       const float* oldvalues, const float* adv, const float* oldlogp, int64_t N,
                                       const double* global_moments, float* actor_grad, float* critic_grad,
                                       float* logvar_grad, double* stats, void* workspace, int64_t workspace_bytes,
                                       dppo_stream_t stream, const dppo_obs_io* io) {
  if (!io) return fail(-1, "null pointer");
  return gauss_ppo_entry(actor, critic, prec, actor_params, actor_packed, critic_params, critic_packed, cfg, logvar, obs, actions,
                         returns, oldvalues, adv, oldlogp, N, global_moments, actor_grad, critic_grad, logvar_grad, stats,
                         workspace, workspace_bytes, stream, io);
}

// ---- mixture-of-Gaussians policy PPO (gmm.hip) ---------------------------------------------------------------------------
static int check_gmm(const dppo_net_desc* mean, const dppo_net_desc* wts, const dppo_gmm_cfg* cfg, const float* logvar) {
  if (int e = check_net(mean)) return e;
  if (int e = check_net(wts)) return e;
  if (!cfg) return fail(-1, "null cfg");
  if (mean->kind != 1 || wts->kind != 1) return fail(-1, "GMM trunks are kind-1 descriptors on the observation");
  if (cfg->num_modes < 1 || cfg->num_modes > GMM_MAX_MODES) return fail(-1, "num_modes out of [1, %d]", GMM_MAX_MODES);
  if (cfg->horizon_steps < 1 || cfg->action_dim < 1) return fail(-1, "bad Ta / Da");
  if (mean->out_dim != cfg->num_modes * cfg->horizon_steps * cfg->action_dim) return fail(-1, "mean trunk out_dim != num_modes * Ta * Da");
  if (wts->out_dim != cfg->num_modes) return fail(-1, "weights trunk out_dim != num_modes");
  if (mean->cond_dim != wts->cond_dim) return fail(-1, "the two trunks observe different cond_dim");
  if (cfg->std_mode != 0 && cfg->std_mode != 1) return fail(-1, "std_mode must be 0 or 1");
  if (cfg->std_mode == 1 && !logvar) return fail(-1, "std_mode 1 needs logvar");
  return 0;
}
template <class P>
static size_t carve_gmm(Carver& c, const dppo_net_desc& am, const dppo_net_desc& aw, const dppo_net_desc* cr, int64_t N, int K,
                        bool train, MlpBufs<P>& Am, MlpBufs<P>& Aw, MlpBufs<P>& Cb, double*& moments, double*& scratch,
                        double*& partial) {
  moments = (double*)c.take(4 * sizeof(double));
  scratch = (double*)c.take(2 * 64 * sizeof(double));
  partial = (double*)c.take((size_t)gmm_blocks(N) * (8 + K) * sizeof(double));
  carve_mlp<P>(c, am, N, train, train, Am);
  carve_mlp<P>(c, aw, N, train, train, Aw);
  if (cr) carve_mlp<P>(c, *cr, N, train, train, Cb);
  return al256(c.off);
}
int64_t dppo_gmm_workspace_bytes(const dppo_net_desc* mean, const dppo_net_desc* weights, const dppo_net_desc* critic, int prec,
                                 int64_t N) {
  if (check_net(mean) || check_net(weights) || (critic && check_net(critic)) || check_prec(prec)) return -1;
  if (N < 1 || N > 0x7fffffff) return fail(-1, "N out of range");
  Carver c{nullptr, 0, 0};
  double *m, *sc, *pa;
  const int K = GMM_MAX_MODES * 64;
  if (prec == DPPO_PREC_F32) {
    MlpBufs<F32> Am, Aw, Cb;
    return (int64_t)carve_gmm<F32>(c, *mean, *weights, critic, N, K, critic != nullptr, Am, Aw, Cb, m, sc, pa);
  }
  MlpBufs<BF16> Am, Aw, Cb;
  return (int64_t)carve_gmm<BF16>(c, *mean, *weights, critic, N, K, critic != nullptr, Am, Aw, Cb, m, sc, pa);
}
template <class P>
static int gmm_infer_impl(const dppo_net_desc& am, const dppo_net_desc& aw, const float* mp, const char* mk, const float* wp,
                          const char* wk, const dppo_gmm_cfg& cfg, const float* logvar, const float* obs, const int64_t* modes,
                          const float* noise, const float* actions, int64_t N, float* out_actions, float* out_logp, void* ws,
                          int64_t wsb, hipStream_t s) {
  Carver c{(char*)ws, 0, (size_t)wsb};
  MlpBufs<P> Am, Aw, Cb;
  double *moments, *scratch, *partial;
  const size_t need = carve_gmm<P>(c, am, aw, nullptr, N, GMM_MAX_MODES * 64, false, Am, Aw, Cb, moments, scratch, partial);
  if ((int64_t)need > wsb) return fail(-1, "workspace too small: need %zu bytes, got %lld", need, (long long)wsb);
  const PackLayout LM = pack_layout<P>(am, 0), LW = pack_layout<P>(aw, 0);
  launch_build_direct<P>(nullptr, nullptr, obs, nullptr, 0, 0, am.cond_dim, N, Am.in, LM.Kp0, s);
  launch_build_direct<P>(nullptr, nullptr, obs, nullptr, 0, 0, aw.cond_dim, N, Aw.in, LW.Kp0, s);
  mlp_forward<P>(am, mp, mk, LM, N, Am, false, s);
  mlp_forward<P>(aw, wp, wk, LW, N, Aw, false, s);
  GmmArgs g;
  memset(&g, 0, sizeof(g));
  g.cfg = cfg, g.mean_pre = Am.out, g.ldm = Am.ldout, g.logits = Aw.out, g.ldl = Aw.ldout, g.logvar = logvar, g.N = N;
  g.AF = cfg.horizon_steps * cfg.action_dim, g.modes_in = modes, g.noise = noise, g.actions = actions;
  g.out_actions = out_actions, g.out_logp = out_logp;
  if (out_actions) launch_gmm_sample(g, s);
  else launch_gmm_logprob(g, s);
  return check_launch();
}
int dppo_gmm_sample(const dppo_net_desc* mean, const dppo_net_desc* weights, int prec, const float* mean_params,
                    const void* mean_packed, const float* weights_params, const void* weights_packed, const dppo_gmm_cfg* cfg,
                    const float* logvar, const float* obs, const int64_t* modes, const float* noise, int64_t B, float* actions,
                    void* workspace, int64_t workspace_bytes, dppo_stream_t stream) {
  if (int e = check_gmm(mean, weights, cfg, logvar)) return e;
  if (int e = check_prec(prec)) return e;
  if (!mean_params || !mean_packed || !weights_params || !weights_packed || !obs || !actions || !workspace) return fail(-1, "null pointer");
  if (B < 1 || B > 0x7fffffff) return fail(-1, "B out of range");
#define CALL(P)                                                                                                                \
  gmm_infer_impl<P>(*mean, *weights, mean_params, (const char*)mean_packed, weights_params, (const char*)weights_packed, *cfg, logvar, \
                    obs, modes, noise, nullptr, B, actions, nullptr, workspace, workspace_bytes, (hipStream_t)stream)
  return DPPO_DISPATCH(prec, CALL);
#undef CALL
}
int dppo_gmm_logprob(const dppo_net_desc* mean, const dppo_net_desc* weights, int prec, const float* mean_params,
                     const void* mean_packed, const float* weights_params, const void* weights_packed, const dppo_gmm_cfg* cfg,
                     const float* logvar, const float* obs, const float* actions, int64_t N, float* logp, void* workspace,
                     int64_t workspace_bytes, dppo_stream_t stream) {
  if (int e = check_gmm(mean, weights, cfg, logvar)) return e;
  if (int e = check_prec(prec)) return e;
  if (!mean_params || !mean_packed || !weights_params || !weights_packed || !obs || !actions || !logp || !workspace)
    return fail(-1, "null pointer");
  if (N < 1 || N > 0x7fffffff) return fail(-1, "N out of range");
#define CALL(P)                                                                                                                \
  gmm_infer_impl<P>(*mean, *weights, mean_params, (const char*)mean_packed, weights_params, (const char*)weights_packed, *cfg, logvar, \
                    obs, nullptr, nullptr, actions, N, nullptr, logp, workspace, workspace_bytes, (hipStream_t)stream)
  return DPPO_DISPATCH(prec, CALL);
#undef CALL
}
template <class P>
static int gmm_ppo_impl(const dppo_net_desc& am, const dppo_net_desc& aw, const dppo_net_desc& cr, const float* mp, const char* mk,
                        const float* wp, const char* wk, const float* cp, const char* ck, const dppo_gmm_cfg& cfg,
                        const float* logvar, const float* obs, const float* actions, const float* returns, const float* oldvalues,
                        const float* adv, const float* oldlogp, int64_t N, const double* gmom, float* mgrad, float* wgrad,
                        float* cgrad, float* lvgrad, double* stats, void* ws, int64_t wsb, hipStream_t s) {
  Carver c{(char*)ws, 0, (size_t)wsb};
  MlpBufs<P> Am, Aw, Cb;
  double *moments, *scratch, *partial;
  const size_t need = carve_gmm<P>(c, am, aw, &cr, N, GMM_MAX_MODES * 64, true, Am, Aw, Cb, moments, scratch, partial);
  if ((int64_t)need > wsb) return fail(-1, "workspace too small: need %zu bytes, got %lld", need, (long long)wsb);
  const PackLayout LM = pack_layout<P>(am, 0), LW = pack_layout<P>(aw, 0), LC = pack_layout<P>(cr, 0);
  hipStream_t s2 = fork_side(s);  // the critic pipeline beside the two actor trunks
  launch_build_direct<P>(nullptr, nullptr, obs, nullptr, 0, 0, cr.cond_dim, N, Cb.in, LC.Kp0, s2);
  mlp_forward<P>(cr, cp, ck, LC, N, Cb, true, s2);
  launch_build_direct<P>(nullptr, nullptr, obs, nullptr, 0, 0, am.cond_dim, N, Am.in, LM.Kp0, s);
  launch_build_direct<P>(nullptr, nullptr, obs, nullptr, 0, 0, aw.cond_dim, N, Aw.in, LW.Kp0, s);
  if (gmom == nullptr) launch_gauss_moments(adv, N, moments, scratch, s);
  mlp_forward<P>(am, mp, mk, LM, N, Am, true, s);
  mlp_forward<P>(aw, wp, wk, LW, N, Aw, true, s);
  if (s2 != s) join_side(s, s2);
  GmmArgs g;
  memset(&g, 0, sizeof(g));
  g.cfg = cfg, g.mean_pre = Am.out, g.ldm = Am.ldout, g.logits = Aw.out, g.ldl = Aw.ldout, g.logvar = logvar, g.actions = actions;
  g.N = N, g.AF = cfg.horizon_steps * cfg.action_dim, g.vnew = Cb.out, g.ldv = Cb.ldout, g.returns = returns;
  g.oldvalues = oldvalues, g.adv = adv, g.oldlogp = oldlogp, g.moments = gmom ? gmom : moments;
  g.d_mean = Am.d_out, g.lddm = LM.Kpo, g.d_logits = Aw.d_out, g.lddl = LW.Kpo, g.d_v = Cb.d_out, g.lddv = LC.Kpo;
  g.partial = partial, g.stats = stats, g.logvar_grad = lvgrad;
  launch_gmm_loss<P>(g, s);
  s2 = fork_side(s);
  mlp_backward<P>(cr, cp, ck, LC, N, Cb, cgrad, nullptr, nullptr, 0, s2, false, -1);
  mlp_backward<P>(am, mp, mk, LM, N, Am, mgrad, nullptr, nullptr, 0, s, false, 1);
  mlp_backward<P>(aw, wp, wk, LW, N, Aw, wgrad, nullptr, nullptr, 0, s, false, 1);
  if (s2 != s) join_side(s, s2);
  return check_launch();
}
int dppo_gmm_ppo_loss_fwd_bwd(const dppo_net_desc* mean, const dppo_net_desc* weights, const dppo_net_desc* critic, int prec,
                              const float* mean_params, const void* mean_packed, const float* weights_params,
                              const void* weights_packed, const float* critic_params, const void* critic_packed,
                              const dppo_gmm_cfg* cfg, const float* logvar, const float* obs, const float* actions,
                              const float* returns, const float* oldvalues, const float* adv, const float* oldlogp, int64_t N,
                              const double* global_moments, float* mean_grad, float* weights_grad, float* critic_grad,
                              float* logvar_grad, double* stats, void* workspace, int64_t workspace_bytes, dppo_stream_t stream) {
  if (int e = check_gmm(mean, weights, cfg, logvar)) return e;
  if (int e = check_net(critic)) return e;
  if (int e = check_prec(prec)) return e;
  if (critic->kind != 1 || critic->out_dim != 1) return fail(-1, "critic descriptor must be kind 1 with out_dim 1");
  if (mean->cond_dim != critic->cond_dim) return fail(-1, "actor and critic observe different cond_dim");
  if (!mean_params || !mean_packed || !weights_params || !weights_packed || !critic_params || !critic_packed || !obs || !actions ||
      !returns || !oldvalues || !adv || !oldlogp || !mean_grad || !weights_grad || !critic_grad || !stats || !workspace)
    return fail(-1, "null pointer");
  if (cfg->std_mode == 1 && !logvar_grad) return fail(-1, "std_mode 1 needs logvar_grad");
  if (N < 2 || N > 0x7fffffff) return fail(-1, "N out of range");
#define CALL(P)                                                                                                                  \
  gmm_ppo_impl<P>(*mean, *weights, *critic, mean_params, (const char*)mean_packed, weights_params, (const char*)weights_packed,        \
                  critic_params, (const char*)critic_packed, *cfg, logvar, obs, actions, returns, oldvalues, adv, oldlogp, N,          \
                  global_moments, mean_grad, weights_grad, critic_grad, logvar_grad, stats, workspace, workspace_bytes, (hipStream_t)stream)
  return DPPO_DISPATCH(prec, CALL);
#undef CALL
}

// ---- conv denoiser: PPO update and supervised loss (unet.hip does the network, this file the loss and the critic) -------
template <class P>
struct UnetPpoWs {
  double *moments, *loss_partial;
  float* loss_tab;
  int32_t *brow, *krow;
  MlpBufs<P> C;
  void* d_eps;
  void* uws;
  size_t ubytes;
  int ldde;
};
template <class P>
static size_t carve_unet_ppo(Carver& c, const dppo_unet_desc& u, const dppo_net_desc* cr, int64_t N, UnetPpoWs<P>& W) {
  W.moments = (double*)c.take((8 + 2 * ADV_MOMENT_BLOCKS) * sizeof(double));
  W.loss_tab = (float*)c.take(2 * 1024 * sizeof(float));
  W.ldde = round_up(u.horizon_steps * u.action_dim, 64);
  const int64_t lb = loss_blocks(N), mb = bc_loss_blocks(N, W.ldde) + 1;
  W.loss_partial = (double*)c.take((size_t)(lb * 8 > mb ? lb * 8 : mb) * sizeof(double));
  W.brow = (int32_t*)c.take((size_t)N * 4);
  W.krow = (int32_t*)c.take((size_t)N * 4);
  if (cr) carve_mlp<P>(c, *cr, N, true, true, W.C);
  W.d_eps = c.take((size_t)N * W.ldde * P::ESIZE);
  W.ubytes = unet_trainer_bytes<P>(u, N);
  W.uws = c.take(W.ubytes);
  return al256(c.off);
}
int64_t dppo_unet_ppo_workspace_bytes(const dppo_unet_desc* actor, const dppo_net_desc* critic, int prec, int64_t N) {
  if (unet_check_desc(actor) || check_net(critic) || check_prec(prec)) return -1;
  if (N < 2 || N > (1 << 24)) return fail(-1, "N out of range");
  Carver c{nullptr, 0, 0};
  if (prec == DPPO_PREC_F32) {
    UnetPpoWs<F32> W;
    return (int64_t)carve_unet_ppo<F32>(c, *actor, critic, N, W);
  }
  UnetPpoWs<BF16> W;
  return (int64_t)carve_unet_ppo<BF16>(c, *actor, critic, N, W);
}
template <class P>
static int unet_ppo_impl(const dppo_unet_desc& u, const dppo_net_desc& cr, const float* ap, const char* ak, const float* cp,
                         const char* ck, const dppo_diffusion_cfg& dcfg, const dppo_ppo_cfg& pcfg, const dppo_step* ksteps,
                         const float* obs_k, const float* chains_k, const float* returns_k, const float* values_k,
                         const float* adv_k, const float* logprobs_k, const int64_t* inds, const int64_t* kinds, int64_t N,
                         const double* gmom, float* agrad, float* cgrad, double* stats, void* ws, int64_t wsb, hipStream_t s,
                         const dppo_obs_io* oio = nullptr) {
  Carver c{(char*)ws, 0, (size_t)wsb};
  UnetPpoWs<P> W;
  const size_t need = carve_unet_ppo<P>(c, u, &cr, N, W);
  if ((int64_t)need > wsb) return fail(-1, "workspace too small: need %zu bytes, got %lld", need, (long long)wsb);
  const PackLayout LC = pack_layout<P>(cr, 0);
  const int Kft = pcfg.ft_denoising_steps, AF = u.horizon_steps * u.action_dim;
  BuildRows br;  // critic rows + the zeroing / loss table side jobs; the conv actor builds its own input images
  memset(&br, 0, sizeof(br));
  br.zero_b = W.moments, br.n_zero_b = 32;
  if (Kft <= 1024) br.loss_tab = W.loss_tab, br.pcfg = pcfg;
  br.inds = inds, br.kinds = kinds, br.chains = chains_k, br.ksteps = ksteps, br.Kft = Kft, br.AF = AF;
  br.obs = oio && oio->obs_critic ? oio->obs_critic : obs_k;  // (pixel nets: the critic encodes the images itself)
  br.cond = cr.cond_dim, br.M = N, br.inC = W.C.in, br.KpC = LC.Kp0, br.brow = W.brow, br.onehot0 = -1;
  launch_build_rows<P>(br, s);
  launch_unet_index(inds, kinds, Kft, N, W.brow, W.krow, s);
  if (gmom == nullptr) launch_adv_moments(adv_k, W.brow, N, W.moments, s);
  // critic pipeline beside the actor's forward
  hipStream_t s2 = fork_side(s);
  mlp_forward<P>(cr, cp, ck, LC, N, W.C, true, s2);
  UnetTrainer<P>* T = unet_trainer_new<P>(u, ap, ak, N, W.uws, W.ubytes, s);
  UnetTrainIO io;
  memset(&io, 0, sizeof(io));
  io.chains = chains_k, io.obs = obs_k, io.brow = W.brow, io.krow = W.krow, io.ksteps = ksteps, io.Kft = Kft;
  io.gathered = kinds != nullptr;
  const float* eps = unet_trainer_forward<P>(T, io);
  if (s2 != s) join_side(s, s2);
  LossArgs la;
  memset(&la, 0, sizeof(la));
  la.eps = eps, la.lde = AF, la.vnew = W.C.out, la.ldv = W.C.ldout, la.brow = W.brow, la.krow = W.krow;
  la.gathered = kinds != nullptr;
  la.chains = chains_k, la.logprobs_k = logprobs_k, la.returns_k = returns_k, la.values_k = values_k, la.adv_k = adv_k;
  la.ksteps = ksteps, la.dcfg = dcfg, la.pcfg = pcfg, la.AF = AF, la.N = N;
  la.moments = gmom ? gmom : W.moments;
  la.tab = Kft <= 1024 ? W.loss_tab : nullptr;
  la.d_eps = W.d_eps, la.ldde = W.ldde, la.d_v = W.C.d_out, la.lddv = LC.Kpo, la.stats = stats;
  la.part = 3, la.partial = W.loss_partial;
  launch_ppo_loss<P>(la, s);
  s2 = fork_side(s);
  mlp_backward<P>(cr, cp, ck, LC, N, W.C, cgrad, nullptr, nullptr, 0, s2, false, -1, &la);
  if (oio && oio->d_obs_critic) obs_grad<P>(cr, cp, N, W.C, oio->d_obs_critic, s2);
  unet_trainer_backward<P>(T, W.d_eps, W.ldde, agrad, oio ? oio->d_obs_actor : nullptr);
  unet_trainer_free<P>(T);
  if (s2 != s) join_side(s, s2);
  return check_launch();
}
static int unet_ppo_entry(const dppo_unet_desc* actor, const dppo_net_desc* critic, int prec, const float* actor_params,
                               const void* actor_packed, const float* critic_params, const void* critic_packed,
                               const dppo_diffusion_cfg* dcfg, const dppo_ppo_cfg* pcfg, const dppo_step* ksteps,
                               const float* obs_k, const float* chains_k, const float* returns_k, const float* values_k,
                               const float* adv_k, const float* logprobs_k, const int64_t* inds, const int64_t* kinds,
                               int64_t N, const double* global_moments, float* actor_grad, float* critic_grad,
                               double* stats, void* workspace, int64_t workspace_bytes, dppo_stream_t stream,
                               const dppo_obs_io* io) {
  if (int e = unet_check_desc(actor)) return e;
  if (int e = check_net(critic)) return e;
  if (int e = check_prec(prec)) return e;
  if (critic->kind != 1 || critic->out_dim != 1) return fail(-1, "critic descriptor must be kind 1 with out_dim 1");
  if (actor->cond_dim != critic->cond_dim && !(io && io->obs_critic))
    return fail(-1, "actor and critic observe different cond_dim");
  if (!actor_params || !actor_packed || !critic_params || !critic_packed || !dcfg || !pcfg || !ksteps || !obs_k ||
      !chains_k || !returns_k || !values_k || !adv_k || !logprobs_k || !actor_grad || !critic_grad || !stats || !workspace)
    return fail(-1, "null pointer");
  if (int e = check_obs_io(io, kinds, false)) return e;
  if ((inds == nullptr) == (kinds == nullptr)) return fail(-1, "pass exactly one of inds (rollout mode) / kinds (gathered mode)");
  if (N < 2 || N > (1 << 24)) return fail(-1, "N out of range");
  if (pcfg->ft_denoising_steps < 1 || pcfg->ft_denoising_steps > 1024) return fail(-1, "Kft out of range");
  if ((size_t)pcfg->ft_denoising_steps * 9 * actor->time_dim * 4 > 64 * 1024) return fail(-1, "Kft * time_dim too large (LDS of the time MLP's backward)");
  if (pcfg->horizon_steps != actor->horizon_steps || pcfg->action_dim != actor->action_dim) return fail(-1, "Ta / Da mismatch");
  if (pcfg->reward_horizon < 1) return fail(-1, "reward_horizon must be >= 1");
#define CALL(P)                                                                                                          \
  unet_ppo_impl<P>(*actor, *critic, actor_params, (const char*)actor_packed, critic_params, (const char*)critic_packed, *dcfg, \
                   *pcfg, ksteps, obs_k, chains_k, returns_k, values_k, adv_k, logprobs_k, inds, kinds, N, global_moments,     \
                   actor_grad, critic_grad, stats, workspace, workspace_bytes, (hipStream_t)stream, io)
  return DPPO_DISPATCH(prec, CALL);
#undef CALL
}
int dppo_unet_ppo_loss_fwd_bwd(const dppo_unet_desc* actor, const dppo_net_desc* critic, int prec, const float* actor_params,
                               const void* actor_packed, const float* critic_params, const void* critic_packed,
                               const dppo_diffusion_cfg* dcfg, const dppo_ppo_cfg* pcfg, const dppo_step* ksteps,
                               const float* obs_k, const float* chains_k, const float* returns_k, const float* values_k,
                               const float* adv_k, const float* logprobs_k, const int64_t* inds, const int64_t* kinds,
                               int64_t N, const double* global_moments, float* actor_grad, float* critic_grad,
                               double* stats, void* workspace, int64_t workspace_bytes, dppo_stream_t stream) {
  return unet_ppo_entry(actor, critic, prec, actor_params, actor_packed, critic_params, critic_packed, dcfg, pcfg, ksteps, obs_k,
                        chains_k, returns_k, values_k, adv_k, logprobs_k, inds, kinds, N, global_moments, actor_grad,
                        critic_grad, stats, workspace, workspace_bytes, stream, nullptr);
}
int dppo_unet_ppo_loss_fwd_bwd_obs(const dppo_unet_desc* actor, const dppo_net_desc* critic, int prec, const float* actor_params,
                                   const void* actor_packed, const float* critic_params, const void* critic_packed,
                                   const dppo_diffusion_cfg* dcfg, const dppo_ppo_cfg* pcfg, const dppo_step* ksteps,
                                   const float* obs_k, const float* chains_k, const float* returns_k, const float* values_k,
                                   const float* adv_k, const float* logprobs_k, const int64_t* kinds, int64_t N,
                                   const double* global_moments, float* actor_grad, float* critic_grad, double* stats,
                                   void* workspace, int64_t workspace_bytes, dppo_stream_t stream, const dppo_obs_io* io) {
  if (!io) return fail(-1, "null pointer");
  return unet_ppo_entry(actor, critic, prec, actor_params, actor_packed, critic_params, critic_packed, dcfg, pcfg, ksteps, obs_k,
                        chains_k, returns_k, values_k, adv_k, logprobs_k, nullptr, kinds, N, global_moments, actor_grad,
                        critic_grad, stats, workspace, workspace_bytes, stream, io);
}
int64_t dppo_unet_denoise_mse_workspace_bytes(const dppo_unet_desc* net, int prec, int64_t N) {
  if (unet_check_desc(net) || check_prec(prec)) return -1;
  if (N < 1 || N > (1 << 24)) return fail(-1, "N out of range");
  Carver c{nullptr, 0, 0};
  if (prec == DPPO_PREC_F32) {
    UnetPpoWs<F32> W;
    return (int64_t)carve_unet_ppo<F32>(c, *net, nullptr, N, W);
  }
  UnetPpoWs<BF16> W;
  return (int64_t)carve_unet_ppo<BF16>(c, *net, nullptr, N, W);
}
template <class P>
static int unet_mse_impl(const dppo_unet_desc& u, const float* prm, const char* pk, const dppo_step* tsteps, int n_time,
                         const float* obs, const float* pairs, const int64_t* kinds, int64_t N, float* grad, double* loss,
                         void* ws, int64_t wsb, hipStream_t s, float* d_obs = nullptr) {
  Carver c{(char*)ws, 0, (size_t)wsb};
  UnetPpoWs<P> W;
  const size_t need = carve_unet_ppo<P>(c, u, nullptr, N, W);
  if ((int64_t)need > wsb) return fail(-1, "workspace too small: need %zu bytes, got %lld", need, (long long)wsb);
  const int AF = u.horizon_steps * u.action_dim;
  launch_unet_index(nullptr, kinds, n_time, N, W.brow, W.krow, s);
  UnetTrainer<P>* T = unet_trainer_new<P>(u, prm, pk, N, W.uws, W.ubytes, s);
  UnetTrainIO io;
  memset(&io, 0, sizeof(io));
  io.chains = pairs, io.obs = obs, io.brow = W.brow, io.krow = W.krow, io.ksteps = tsteps, io.Kft = n_time, io.gathered = 1;
  const float* eps = unet_trainer_forward<P>(T, io);
  MseArgs ma;
  ma.eps = eps, ma.lde = AF, ma.pairs = pairs, ma.AF = AF, ma.M = N, ma.d_eps = W.d_eps, ma.ldde = W.ldde, ma.loss = loss;
  ma.partial = W.loss_partial;
  launch_mse_loss<P>(ma, s);
  unet_trainer_backward<P>(T, W.d_eps, W.ldde, grad, d_obs);
  unet_trainer_free<P>(T);
  return check_launch();
}
static int unet_mse_entry(const dppo_unet_desc* net, int prec, const float* params, const void* packed,
                                  const dppo_step* tsteps, int n_time, const float* obs, const float* pairs,
                                  const int64_t* kinds, int64_t N, float* grad, double* loss, void* workspace,
                                  int64_t workspace_bytes, dppo_stream_t stream, float* d_obs) {
  if (int e = unet_check_desc(net)) return e;
  if (int e = check_prec(prec)) return e;
  if (!params || !packed || !tsteps || !obs || !pairs || !kinds || !grad || !loss || !workspace) return fail(-1, "null pointer");
  if (N < 1 || N > (1 << 24) || n_time < 1 || n_time > 1024) return fail(-1, "N / n_time out of range");
  if ((size_t)n_time * 9 * net->time_dim * 4 > 64 * 1024) return fail(-1, "n_time * time_dim too large (LDS of the time MLP's backward)");
#define CALL(P)                                                                                                          \
  unet_mse_impl<P>(*net, params, (const char*)packed, tsteps, n_time, obs, pairs, kinds, N, grad, loss, workspace, workspace_bytes, \
                   (hipStream_t)stream, d_obs)
  return DPPO_DISPATCH(prec, CALL);
#undef CALL
}
int dppo_unet_denoise_mse_fwd_bwd(const dppo_unet_desc* net, int prec, const float* params, const void* packed,
                                  const dppo_step* tsteps, int n_time, const float* obs, const float* pairs,
                                  const int64_t* kinds, int64_t N, float* grad, double* loss, void* workspace,
                                  int64_t workspace_bytes, dppo_stream_t stream) {
  return unet_mse_entry(net, prec, params, packed, tsteps, n_time, obs, pairs, kinds, N, grad, loss, workspace, workspace_bytes,
                        stream, nullptr);
}
int dppo_unet_denoise_mse_fwd_bwd_obs(const dppo_unet_desc* net, int prec, const float* params, const void* packed,
                                      const dppo_step* tsteps, int n_time, const float* obs, const float* pairs,
                                      const int64_t* kinds, int64_t N, float* grad, double* loss, void* workspace,
                                      int64_t workspace_bytes, dppo_stream_t stream, float* d_obs) {
  if (!d_obs) return fail(-1, "null pointer");
  return unet_mse_entry(net, prec, params, packed, tsteps, n_time, obs, pairs, kinds, N, grad, loss, workspace, workspace_bytes,
                        stream, d_obs);
}

// ---- optimiser ----------------------------------------------------------------------------------------
int dppo_grad_sq_norm(const float* grad, int64_t n, double* scratch, double* out, dppo_stream_t stream) {
  if (!grad || !scratch || !out || n < 1) return fail(-1, "bad argument");
  launch_sq_norm(grad, n, scratch, out, (hipStream_t)stream);
  return check_launch();
}

int dppo_adamw_step(float* params, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, int step, double lr,
                    double beta1, double beta2, double eps, double weight_decay, const double* sq_norm, double max_norm,
                    dppo_stream_t stream) {
  if (!params || !grad || !exp_avg || !exp_avg_sq || n < 1 || step < 1) return fail(-1, "bad argument");
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  launch_adamw(params, grad, exp_avg, exp_avg_sq, n, (float)(1.0 - lr * weight_decay), (float)(1.0 - beta1), (float)beta2,
               (float)(1.0 - beta2), (float)(lr / bc1), (float)sqrt(bc2), (float)eps, sq_norm, (float)max_norm,
               (hipStream_t)stream);
  return check_launch();
}


int dppo_adamw_step_dev(float* params, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, int32_t* step_dev,
                        const float* lr_dev, double beta1, double beta2, double eps, double weight_decay,
                        const double* sq_norm, double max_norm, dppo_stream_t stream) {
  if (!params || !grad || !exp_avg || !exp_avg_sq || !step_dev || !lr_dev || n < 1) return fail(-1, "bad argument");
  launch_adamw_dev(params, grad, exp_avg, exp_avg_sq, n, step_dev, lr_dev, beta1, beta2, (float)eps, weight_decay, sq_norm,
                   (float)max_norm, (hipStream_t)stream);
  return check_launch();
}

int dppo_adamw_step_multi(const dppo_adamw_slot* slots, int n_slots, dppo_stream_t stream) {
  if (!slots || n_slots < 1 || n_slots > 4) return fail(-1, "1..4 slots");
  AdamwSlots a;
  memset(&a, 0, sizeof(a));
  for (int i = 0; i < n_slots; ++i) {
    const dppo_adamw_slot& q = slots[i];
    if (!q.params || !q.grad || !q.exp_avg || !q.exp_avg_sq || !q.step_dev || !q.lr_dev || q.n < 1)
      return fail(-1, "bad argument in slot %d", i);
    AdamwSlot& t = a.s[i];
    t.p = q.params, t.g = q.grad, t.m = q.exp_avg, t.v = q.exp_avg_sq, t.n = q.n, t.step_dev = q.step_dev;
    t.lr_dev = q.lr_dev, t.beta1 = q.beta1, t.beta2 = q.beta2, t.weight_decay = q.weight_decay, t.eps = (float)q.eps;
    t.max_norm = (float)q.max_norm, t.sq_norm = q.sq_norm;
  }
  a.n = n_slots;
  launch_adamw_multi(a, (hipStream_t)stream);
  return check_launch();
}

int dppo_stats_split(const double* stats, float* hi_lo, dppo_stream_t stream) {
  if (!stats || !hi_lo) return fail(-1, "null pointer");
  launch_stats_split(stats, hi_lo, DPPO_STAT_COUNT, (hipStream_t)stream);
  return check_launch();
}
int dppo_stats_merge(const float* hi_lo, double* stats, int world, dppo_stream_t stream) {
  return dppo_stats_merge_n(hi_lo, stats, world, 2, stream);
}
int dppo_stats_merge_n(const float* hi_lo, double* stats, int world, int n_avg, dppo_stream_t stream) {
  if (!stats || !hi_lo || world < 1 || n_avg < 0 || DPPO_STAT_ADV_MEAN + n_avg > DPPO_STAT_COUNT) return fail(-1, "bad argument");
  launch_stats_merge(hi_lo, stats, DPPO_STAT_COUNT, DPPO_STAT_ADV_MEAN, n_avg, 1.0 / world, (hipStream_t)stream);
  return check_launch();
}

// ---- measurement hook ----------------------------------------------------------------------------------
int dppo_probe_arm(int kernel_id, int max_launches) {
  if (probe_arm(kernel_id, max_launches)) return fail(-1, "probe already armed or bad size");
  return 0;
}
int dppo_probe_collect(double* total_ms_host, int* launches_host, double* flops_host) {
  if (!total_ms_host || !launches_host || !flops_host) return fail(-1, "null pointer");
  if (probe_collect(total_ms_host, launches_host, flops_host)) return fail(-1, "probe not armed / event error");
  return 0;
}
int dppo_probe_collect_bytes(double* total_ms_host, int* launches_host, double* flops_host, double* bytes_host) {
  if (!total_ms_host || !launches_host || !flops_host || !bytes_host) return fail(-1, "null pointer");
  if (probe_collect(total_ms_host, launches_host, flops_host, bytes_host)) return fail(-1, "probe not armed / event error");
  return 0;
}

// ---- tuning / micro-benchmark hooks -------------------------------------------------------------------
int dppo_tune_set(int knob, int value) {
  if (knob == 0) {
    set_gemm_nt_variant(value);
    return 0;
  }
  if (knob == 1) {
    g_use_fused = value;
    return 0;
  }
  if (knob == 2) {
    g_overlap = value;
    return 0;
  }
  if (knob == 3 && value >= 1) {
    g_tn_target = value;
    return 0;
  }
  if (knob == 4 && value >= 1) {
    g_tn_max_splits = value;
    return 0;
  }
  if (knob == 5) {
    set_gemm_tn_variant(value);
    return 0;
  }
  if (knob == 6) {
    set_gemm_tn_thin(value);
    return 0;
  }
  if (knob == 7) {
    set_fused_short_tiles(value);
    return 0;
  }
  if (knob == 8) {
    g_dbg = value;
    return 0;
  }
  if (knob == 9) {
    g_side_low_priority = value;
    return 0;
  }
  if (knob == 10) {
    g_gate_critic = value;
    return 0;
  }
  if (knob == 11) {
    g_temb_onehot = value;
    return 0;
  }
  if (knob == 12) {
    g_tn_group = value;
    return 0;
  }
  if (knob == 13) {
    g_pack_one = value;
    return 0;
  }
  if (knob == 14) {
    g_early_join = value;
    return 0;
  }
  if (knob == 15) {
    set_sampler_l0_lds(value);
    return 0;
  }
  if (knob == 16) {
    g_lowrank_top = value;
    return 0;
  }
  if (knob == 17) {
    g_merge_top = value;
    return 0;
  }
  if (knob == 18) {
    g_post_one = value;
    return 0;
  }
  if (knob == 27) {  // sampler: one 16-row tile over eight workgroups for small env batches (1, default) or one (0)
    set_sampler_split(value);
    return 0;
  }
  if (knob == 28) {  // split sampler: 64-cycle sleep periods between a member's exchange store and its first sweep (default 4)
    set_sampler_split_pre_sweep(value);
    return 0;
  }
  if (knob == 31) {  // weight-gradient GEMMs of one-block bf16 networks from K-major fragment operands (1, default) or row-major (0)
    g_frag = value;
    return 0;
  }
  if (knob == 32) {  // fragment GEMM: 0 (default) LDS-ring kernel; 2..4 register-only kernel with that lookahead
    set_gemm_tn_frag_depth(value);
    return 0;
  }
  if (knob == 36) {  // advantage moments: partial sums by the row builder's last blocks, added up by the loss kernel (1, default) or adv_moments_kernel (0)
    g_mom_rider = value;
    return 0;
  }
  if (knob == 37) {  // in-kernel first-layer weight gradient of the one-block backward (1, default) or dh_0 stored and a GEMM of its own (0)
    g_dw0 = value;
    return 0;
  }
  if (knob == 38) {  // with knob 37: the reductions the backward kernel alone feeds and the time-embedding gradient on a side stream under the GEMMs (1, default)
    g_side_tail = value;
    return 0;
  }
  if (knob == 40) {  // with knob 38: that work as riders of the actor's weight-gradient GEMM launch (1) or on a side stream (0, default)
    g_tail_riders = value;
    return 0;
  }
  if (knob == 41) {  // with knob 38: the GEMMs' slab reductions and the post-reduce parts behind them in one launch (1, default) or two (0)
    g_tail_post = value;
    return 0;
  }
  if (knob == 39) {  // the policy half of the PPO loss in the epilogue of the actor's fused forward (1) or a launch of its own (0, default)
    g_fuse_loss = value;
    return 0;
  }
  if (knob == 35) {  // the reductions behind the weight-gradient GEMMs inside their launch (1, default) or as a launch of their own (0)
    g_fold = value;
    return 0;
  }
  if (knob == 34) {  // fragment GEMM, timing experiments (results are wrong while set): 1 no MFMAs, 2 no ring loads, 4 no prefetch
    set_gemm_tn_frag_dbg(value);
    return 0;
  }
  if (knob == 33) {  // fragment GEMM: k-steps its L2 prefetch runs ahead of the ring's own loads (default 12; 0: none ahead)
    set_gemm_tn_frag_pfd(value);
    return 0;
  }
  if (knob == 30 && value >= 1) {  // low-rank dW2 (knob 16) and with it the one-block backward: on for M >= value x out_dim (default 100)
    g_lowrank_ratio = value;
    return 0;
  }
  if (knob == 29) {  // split sampler: sweeps a member waits before giving up (tests force a time-out with 1; <= 0: default 2^20)
    set_sampler_split_spin_limit(value);
    return 0;
  }
  if (knob == 26) {  // grouped weight-gradient GEMM: LDS stages (1, default: three workgroups per CU; 2)
    set_gemm_tn_nbuf(value);
    return 0;
  }
  if (knob == 25) {  // one-block kernels: short layers walked without their padding k-steps (1, default) or padded (0)
    set_fused_compact(value);
    return 0;
  }
  if (knob == 23) {  // fused backward of one-block networks: the specialised kernel (1, default) or the general one (0)
    set_fused_bwd_one(value);
    return 0;
  }
  if (knob == 22) {  // fused forward of one-block networks: second layer merged into the out layer (1, default) or not (0);
    set_fused_merge_fwd(value);  // takes effect at the next pack (the images of both forms are always packed)
    return 0;
  }
  if (knob == 21) {  // gemm_nt: small tiles for small problems (1, default) or the 128 x 128 / 64 x 128 / 16 x 256 shapes only (0)
    set_gemm_nt_small(value);
    return 0;
  }
  if (knob == 20) {  // visual encoder: attention on the matrix cores (1, default) or the scalar kernels (0)
    dppo::set_vis_mfma_attn(value);
    return 0;
  }
  return fail(-1, "unknown tuning knob %d", knob);
}

int dppo_gemm_nt_raw(int prec, const void* X, const void* W, const float* bias, int64_t M, int N, int Kp, float* out_f32,
                     void* out_elem, int ldo, int act, dppo_stream_t stream) {
  if (int e = check_prec(prec)) return e;
  if (!X || !W || M < 1 || M > 0x7fffffff || N < 1) return fail(-1, "bad argument");
  const int es = prec == DPPO_PREC_F32 ? 4 : 2;
  if (Kp < 1 || (Kp * es) % 128) return fail(-1, "Kp must be a multiple of %d", 128 / es);
  if (ldo < ((N + 15) & ~15)) return fail(-1, "ldo too small");
  GemmNT g;
  memset(&g, 0, sizeof(g));
  g.X = X, g.W = W, g.bias = bias, g.M = (int)M, g.N = N, g.Kp = Kp, g.ldx = Kp, g.ldw = Kp;
  g.out_f32 = out_f32, g.ldo32 = ldo, g.out_act = out_elem, g.ldo = ldo, g.act = act;
  if (prec == DPPO_PREC_F32)
    launch_gemm_nt<F32>(g, (hipStream_t)stream);
  else
    launch_gemm_nt<BF16>(g, (hipStream_t)stream);
  return check_launch();
}

int dppo_gemm_tn_raw(int prec, const void* A, int lda, int N1, const void* B, int ldb, int N2, int64_t M,
                     int rows_per_split, float* slab, float* C, dppo_stream_t stream) {
  if (int e = check_prec(prec)) return e;
  if (!A || !B || !slab || !C || M < 1 || M > 0x7fffffff || N1 < 1 || N2 < 1) return fail(-1, "bad argument");
  if (rows_per_split < 64 || rows_per_split % 64) return fail(-1, "rows_per_split must be a multiple of 64");
  GemmTN t;
  memset(&t, 0, sizeof(t));
  t.A = A, t.B = B, t.M = (int)M, t.N1 = N1, t.N2 = N2, t.lda = lda, t.ldb = ldb, t.slab = slab, t.ldc = N2;
  t.rows_per_split = rows_per_split, t.splits = (int)((M + rows_per_split - 1) / rows_per_split);
  if (prec == DPPO_PREC_F32)
    launch_gemm_tn<F32>(t, (hipStream_t)stream);
  else
    launch_gemm_tn<BF16>(t, (hipStream_t)stream);
  launch_slab_reduce_2d(slab, t.splits, N1, N2, N2, C, N2, 1.f, (hipStream_t)stream);
  return check_launch();
}
