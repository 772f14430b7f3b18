// Conv denoiser (Unet1D) on gfx950: see the dppo_unet_* section of include/dppo_hip.h.  Compiled with -ffp-contract=off
// (the step kernel repeats the sampler's posterior arithmetic).
//
// Layout.  Activations are channel-LAST and time-PADDED: img[b][tp][c], tp = t + PAD (PAD = 2 zero rows on either side),
// c contiguous.  The im2col row of output (b, t) for a k-tap convolution is then the CONTIGUOUS window
// img[b][t + PAD - k/2 .. + k) -- so a convolution is gemm_nt over rows of stride C and depth k*C with no gather at all:
// X = img + (PAD - k/2) rows, ldx = C, Kp = k*C, one GEMM row per padded position (the 4 pad positions per sample compute
// garbage nobody reads: (T+4)/T of the useful work, at shapes where the launch, not the MFMA, is the cost).  The stride-2
// Downsample1d is the same with ldx = 2C; the ConvTranspose1d of Upsample1d is ONE GEMM with N = 2C (even | odd output
// phase) over the 3-row window (m-1, m, m+1).  Weights are packed once per optimiser step as [Cout][k][Cin_p] (Cin padded to
// 64: only the first conv, Cin = action_dim).  GroupNorm + activation + FiLM (or + residual) is one epilogue kernel per
// block half: one workgroup per sample, two-pass statistics in fp32, writes the next padded image (and zeroes its pads).
#include <string.h>

#include <vector>

#include "gemm.h"
#include "posterior.h"
#include "ppo.h"
#include "unet.h"

namespace dppo {
int api_fail(int code, const char* msg);
int api_check_launch();

namespace {

constexpr int PAD = 2;
inline int rup(int x, int m) { return (x + m - 1) / m * m; }
inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

// ---------------------------------------------------------------------------------------------------------------------
// parameter / packed-image layout: one walk in the reference's state-dict order (model/diffusion/unet.py:139-262)
// ---------------------------------------------------------------------------------------------------------------------
struct Lin {
  int64_t w, b;   // float offsets in the flat parameter buffer
  size_t pk;      // byte offset of the packed [out][Kp] operand
  size_t pkT;     // backward: W^T as [in_p][out] (data gradient); in_p = inT rows kept (all, or the time-embedding columns)
  int in, out, Kp, inT;
};
struct Conv {
  int64_t w, b;
  size_t pk;
  size_t pkT;  // backward (data gradient) operand, see pack_convT_bwd / pack_conv_bwd / pack_down_bwd
  int ci, co, ks, cip, Kp;  // cip: channel stride of the image it reads (each map padded to a multiple of 64 channels)
  bool transposed;
  int stride;  // 1, or 2 for Downsample1d
  int split;   // > 0: the input is a concat image [x (split, padded) | skip (split, padded)]: channel c >= split sits at
               // rup(split, 64) + c - split
};
// position of real input channel c in the image a conv reads
__host__ __device__ inline int chan_pos(int c, int split) { return (split > 0 && c >= split) ? ((split + 63) / 64 * 64) + c - split : c; }
struct Norm {
  int64_t g, b;
};
struct ResBlock {
  Conv c1, c2, res;
  Norm n1, n2;
  Lin enc[3];
  int n_enc;
  bool has_res;
  int ci, co, cc;
};
struct Layout {
  Lin t1, t2;
  std::vector<ResBlock> mid, down, up;  // down / up: two per level
  std::vector<Conv> downs, ups;
  Conv fin;
  Norm fin_n;
  Conv fin_out;
  int64_t n_params;
  size_t temb, pk_bytes;
  int Kg;  // padded width of the conditioning vector [time embedding | state]
  std::vector<int> dims;  // [action_dim, dim*m0, dim*m1, ...]
};

Layout make_layout(const dppo_unet_desc& d, int es, int n_time) {
  Layout L;
  int64_t o = 0;
  size_t pk = 0;
  const int cbd = d.time_dim + d.cond_dim;
  L.Kg = rup(cbd, 64);
  auto lin = [&](int in, int out) {
    Lin l;
    l.in = in, l.out = out, l.Kp = rup(in, 64);
    l.w = o, o += (int64_t)in * out;
    l.b = o, o += out;
    l.pk = pk, pk = al(pk + (size_t)out * l.Kp * es);
    l.inT = in, l.pkT = pk, pk = al(pk + (size_t)rup(in, 16) * rup(out, 64) * es);
    return l;
  };
  auto conv = [&](int ci, int co, int ks, bool tr = false, int split = 0) {
    Conv c;
    c.ci = ci, c.co = co, c.ks = ks, c.cip = split > 0 ? 2 * rup(split, 64) : rup(ci, 64), c.transposed = tr, c.split = split;
    c.w = o, o += (int64_t)ci * co * ks;
    c.b = o, o += co;
    c.stride = 1;
    if (!tr) {
      c.Kp = ks * c.cip;
      c.pk = pk, pk = al(pk + (size_t)co * c.Kp * es);
      // data gradient: a convolution of the padded dU image with the flipped kernel, [ci][ks * co]; the stride-2 conv's
      // is a two-phase GEMM [2 ci][2 co] (set by the caller below)
      c.pkT = pk, pk = al(pk + (size_t)2 * rup(ci, 16) * 3 * rup(co, 64) * es + (size_t)rup(ci, 16) * ks * rup(co, 64) * es);
    } else {  // ConvTranspose1d(C, C, 4, 2, 1) as one GEMM: N = 2 co (even | odd phase), K = 3 ci (window m-1, m, m+1)
      c.Kp = 3 * c.cip;
      c.pk = pk, pk = al(pk + (size_t)2 * co * c.Kp * es);
      c.pkT = pk, pk = al(pk + (size_t)rup(ci, 16) * 4 * rup(co, 64) * es);  // strided conv over dy: [ci][4 co_p]
    }
    return c;
  };
  auto norm = [&](int c) {
    Norm n;
    n.g = o, o += c;
    n.b = o, o += c;
    return n;
  };
  auto resblock = [&](int ci, int co, int split = 0) {
    ResBlock r;
    memset(&r, 0, sizeof(r));
    r.ci = ci, r.co = co, r.cc = d.cond_predict_scale ? 2 * co : co;
    r.c1 = conv(ci, co, d.kernel_size, false, split), r.n1 = norm(co);
    r.c2 = conv(co, co, d.kernel_size), r.n2 = norm(co);
    if (d.larger_encoder) {
      r.n_enc = 3;
      r.enc[0] = lin(cbd, r.cc), r.enc[1] = lin(r.cc, r.cc), r.enc[2] = lin(r.cc, r.cc);
    } else {
      r.n_enc = 1;
      r.enc[0] = lin(cbd, r.cc);
    }
    r.has_res = ci != co;
    if (r.has_res) r.res = conv(ci, co, 1, false, split);
    return r;
  };
  L.t1 = lin(d.time_dim, 4 * d.time_dim);
  L.t2 = lin(4 * d.time_dim, d.time_dim);
  L.dims.push_back(d.action_dim);
  for (int i = 0; i < d.n_levels; ++i) L.dims.push_back(d.dim * d.mults[i]);
  const int nl = d.n_levels, top = L.dims[nl];
  L.mid.push_back(resblock(top, top));
  L.mid.push_back(resblock(top, top));
  for (int i = 0; i < nl; ++i) {
    L.down.push_back(resblock(L.dims[i], L.dims[i + 1]));
    L.down.push_back(resblock(L.dims[i + 1], L.dims[i + 1]));
    if (i < nl - 1) {
      L.downs.push_back(conv(L.dims[i + 1], L.dims[i + 1], 3));
      L.downs.back().stride = 2;
    }
  }
  for (int j = 0; j < nl - 1; ++j) {  // (dim_in, dim_out) = reversed(in_out[1:])[j] = (dims[nl-1-j], dims[nl-j])
    const int din = L.dims[nl - 1 - j], dout = L.dims[nl - j];
    L.up.push_back(resblock(2 * dout, din, dout));
    L.up.push_back(resblock(din, din));
    L.ups.push_back(conv(din, din, 4, true));
  }
  L.fin = conv(d.dim, d.dim, d.kernel_size), L.fin_n = norm(d.dim);
  L.fin_out = conv(d.dim, d.action_dim, 1);
  L.n_params = o;
  L.temb = pk, pk = al(pk + (size_t)(n_time > 0 ? n_time : 1) * d.time_dim * 4);
  L.pk_bytes = pk;
  return L;
}

int check_desc(const dppo_unet_desc* d) {
  if (!d) return api_fail(-1, "null unet descriptor");
  if (d->n_levels < 1 || d->n_levels > 4) return api_fail(-1, "unet: n_levels out of [1,4]");
  if (d->dim < 8 || d->dim % 8) return api_fail(-1, "unet: dim must be a positive multiple of 8");
  for (int i = 0; i < d->n_levels; ++i)
    if (d->mults[i] < 1 || d->dim * d->mults[i] > 1024) return api_fail(-1, "unet: channel count out of range");
  if (d->kernel_size != 3 && d->kernel_size != 5) return api_fail(-1, "unet: kernel_size must be 3 or 5");
  if (d->n_groups < 1 || d->n_groups > 32) return api_fail(-1, "unet: n_groups out of [1,32]");
  for (int i = 0; i < d->n_levels; ++i)
    if ((d->dim * d->mults[i]) % d->n_groups) return api_fail(-1, "unet: channels not divisible by n_groups");
  if (d->action_dim < 1 || d->action_dim > 64) return api_fail(-1, "unet: action_dim out of [1,64]");
  if (d->time_dim < 4 || d->time_dim % 2 || d->time_dim > 128) return api_fail(-1, "unet: time_dim must be even, in [4,128]");
  if (d->cond_dim < 1 || d->cond_dim > 1024) return api_fail(-1, "unet: cond_dim out of range");
  if (d->horizon_steps < 1 || d->horizon_steps > 64) return api_fail(-1, "unet: horizon_steps out of [1,64]");
  if (d->horizon_steps % (1 << (d->n_levels - 1)))
    return api_fail(-1, "unet: horizon_steps must be divisible by 2^(n_levels-1) (Downsample1d / Upsample1d round trip)");
  if (d->act != DPPO_ACT_RELU && d->act != DPPO_ACT_MISH) return api_fail(-1, "unet: activation unsupported");
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// pack kernels
// ---------------------------------------------------------------------------------------------------------------------
template <class P>
__global__ void pack_conv_kernel(const float* w, int co, int ci, int ks, int cip, int split, typename P::elem_t* dst) {
  // dst[o][k * cip + chan_pos(c)] = w[o][c][k]; zero elsewhere (dst is cleared by the caller's memset)
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)co * ci * ks) return;
  const int k = (int)(i % ks), c = (int)((i / ks) % ci), o = (int)(i / ((size_t)ks * ci));
  dst[(size_t)o * ks * cip + (size_t)k * cip + chan_pos(c, split)] = P::from_f32(w[i]);
}
template <class P>
__global__ void pack_convT_kernel(const float* w, int ch, int chp, typename P::elem_t* dst) {
  // w[ci][co][4] (ConvTranspose1d, stride 2, padding 1): out[2m] = x[m] w[.,.,1] + x[m-1] w[.,.,3];
  // out[2m+1] = x[m+1] w[.,.,0] + x[m] w[.,.,2].  Window slots (m-1, m, m+1) -> dst[phase * ch + co][slot * chp + ci]
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t Kp = (size_t)3 * chp;
  if (i >= (size_t)2 * ch * Kp) return;
  const int row = (int)(i / Kp), r = (int)(i % Kp), slot = r / chp, ci = r % chp;
  const int phase = row / ch, co = row % ch;
  int k = -1;
  if (phase == 0) k = slot == 1 ? 1 : (slot == 0 ? 3 : -1);
  else k = slot == 2 ? 0 : (slot == 1 ? 2 : -1);
  dst[i] = P::from_f32(k >= 0 && ci < ch ? w[((size_t)ci * ch + co) * 4 + k] : 0.f);
}
// backward (data-gradient) operands ---------------------------------------------------------------------------------
template <class P>
__global__ void pack_conv_bwd_kernel(const float* w, int co, int ci, int ks, int cop, typename P::elem_t* dst) {
  // dX[b][t][ci] = sum_j sum_co dUimg[b][t - ks/2 + j][co] * w[co][ci][ks-1-j]  ->  dst[ci][j * cop + o] (real ci order)
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t K = (size_t)ks * cop;
  if (i >= (size_t)ci * K) return;
  const int c = (int)(i / K), r = (int)(i % K), j = r / cop, o = r % cop;
  dst[i] = P::from_f32(o < co ? w[((size_t)o * ci + c) * ks + (ks - 1 - j)] : 0.f);
}
template <class P>
__global__ void pack_down_bwd_kernel(const float* w, int ch, int chp, typename P::elem_t* dst) {
  // Downsample1d (k 3, stride 2, pad 1): dx[2m] = W1^T dy[m]; dx[2m+1] = W2^T dy[m] + W0^T dy[m+1]
  // window slots (dy[m], dy[m+1]) -> dst[phase * ch + ci][slot * chp + co]
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t K = (size_t)2 * chp;
  if (i >= (size_t)2 * ch * K) return;
  const int row = (int)(i / K), r = (int)(i % K), slot = r / chp, co = r % chp, phase = row / ch, ci = row % ch;
  int k = -1;
  if (phase == 0) k = slot == 0 ? 1 : -1;
  else k = slot == 0 ? 2 : 0;
  dst[i] = P::from_f32(k >= 0 && co < ch ? w[((size_t)co * ch + ci) * 3 + k] : 0.f);
}
template <class P>
__global__ void pack_convT_bwd_kernel(const float* w, int ch, int chp, typename P::elem_t* dst) {
  // Upsample1d: dx[s][ci] = sum_k sum_co dy[2s - 1 + k][co] w[ci][co][k]  ->  dst[ci][k * chp + co]
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t K = (size_t)4 * chp;
  if (i >= (size_t)ch * K) return;
  const int ci = (int)(i / K), r = (int)(i % K), k = r / chp, co = r % chp;
  dst[i] = P::from_f32(co < ch ? w[((size_t)ci * ch + co) * 4 + k] : 0.f);
}
__device__ __forceinline__ float sinus(int t, int j, int td) {
  const int half = td / 2;
  const float step = -logf(10000.f) / (float)(half - 1);
  const int jj = j < half ? j : j - half;
  const float ang = (float)t * expf((float)jj * step);
  return j < half ? sinf(ang) : cosf(ang);
}
// time_mlp: Linear(d, 4d) -> Mish -> Linear(4d, d) of the sinusoidal embedding, one block per diffusion time
__global__ void unet_time_table_kernel(const float* w1, const float* b1, const float* w2, const float* b2, int td,
                                       float* temb) {
  extern __shared__ float sh[];  // [td] + [4 td]
  float* e0 = sh;
  float* a1 = sh + td;
  const int t = blockIdx.x;
  for (int j = threadIdx.x; j < td; j += blockDim.x) e0[j] = sinus(t, j, td);
  __syncthreads();
  for (int o = threadIdx.x; o < 4 * td; o += blockDim.x) {
    float s = b1[o];
    for (int j = 0; j < td; ++j) s += w1[o * td + j] * e0[j];
    a1[o] = mish_f(s);
  }
  __syncthreads();
  for (int o = threadIdx.x; o < td; o += blockDim.x) {
    float s = b2[o];
    for (int j = 0; j < 4 * td; ++j) s += w2[o * 4 * td + j] * a1[j];
    temb[(size_t)t * td + o] = s;
  }
}

template <class P>
int pack_impl(const dppo_unet_desc& d, int n_time, const float* prm, char* pk, hipStream_t s) {
  typedef typename P::elem_t E;
  const Layout L = make_layout(d, P::ESIZE, n_time);
  auto pl = [&](const Lin& l) {
    launch_cast_pad<P>(prm + l.w, l.out, l.in, l.in, pk + l.pk, l.Kp, s);
    launch_transpose_cast<P>(prm + l.w, l.out, l.inT, l.in, 0, pk + l.pkT, rup(l.out, 64), s);  // [inT][out_p] = W^T
  };
  auto pc = [&](const Conv& c) {
    const int cop = rup(c.co, 64);
    if (!c.transposed) {
      launch_zero_bytes(pk + c.pk, (size_t)c.co * c.Kp * P::ESIZE, s);
      const size_t n = (size_t)c.co * c.ci * c.ks;
      hipLaunchKernelGGL((pack_conv_kernel<P>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, prm + c.w, c.co, c.ci,
                         c.ks, c.cip, c.split, (E*)(pk + c.pk));
      if (c.stride == 2) {
        const size_t nb = (size_t)2 * c.co * 2 * cop;
        hipLaunchKernelGGL((pack_down_bwd_kernel<P>), dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, s, prm + c.w, c.co,
                           cop, (E*)(pk + c.pkT));
      } else if (&c != &L.fin_out) {  // (the 1x1 output conv's data-gradient operand is packed below)
        const size_t nb = (size_t)c.ci * c.ks * cop;
        hipLaunchKernelGGL((pack_conv_bwd_kernel<P>), dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, s, prm + c.w, c.co,
                           c.ci, c.ks, cop, (E*)(pk + c.pkT));
      }
    } else {
      const size_t n = (size_t)2 * c.co * c.Kp;
      hipLaunchKernelGGL((pack_convT_kernel<P>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, prm + c.w, c.co, cop,
                         (E*)(pk + c.pk));
      const size_t nb = (size_t)c.co * 4 * cop;
      hipLaunchKernelGGL((pack_convT_bwd_kernel<P>), dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, s, prm + c.w, c.co,
                         cop, (E*)(pk + c.pkT));
    }
  };
  auto pr = [&](const ResBlock& r) {
    pc(r.c1), pc(r.c2);
    for (int i = 0; i < r.n_enc; ++i) pl(r.enc[i]);
    if (r.has_res) pc(r.res);
  };
  for (auto& r : L.mid) pr(r);
  for (auto& r : L.down) pr(r);
  for (auto& r : L.up) pr(r);
  for (auto& c : L.downs) pc(c);
  for (auto& c : L.ups) pc(c);
  pc(L.fin), pc(L.fin_out);
  // the 1x1 output conv's data-gradient operand: [dim][64] = Wout^T, zero beyond action_dim
  launch_transpose_cast<P>(prm + L.fin_out.w, d.action_dim, d.dim, d.dim, 0, pk + L.fin_out.pkT, 64, s);
  if (n_time > 0)
    hipLaunchKernelGGL(unet_time_table_kernel, dim3(n_time), dim3(64), 5 * d.time_dim * sizeof(float), s, prm + L.t1.w,
                       prm + L.t1.b, prm + L.t2.w, prm + L.t2.b, d.time_dim, (float*)(pk + L.temb));
  return api_check_launch();
}

// ---------------------------------------------------------------------------------------------------------------------
// forward: epilogue / data-movement kernels
// ---------------------------------------------------------------------------------------------------------------------
struct Img {  // channel-last, time-padded activation image [rows][T + 2 PAD][C] elem
  void* p;
  int T, C;
  int Tp() const { return T + 2 * PAD; }
};

// x f32 [rows][T][Da] -> img[rows][Tp][C] (channels >= Da and the pad rows zero)
template <class P>
__global__ void unet_input_kernel(const float* x, int T, int Da, typename P::elem_t* img, int C) {
  const int64_t b = blockIdx.x;
  const int Tp = T + 2 * PAD;
  for (int i = threadIdx.x; i < Tp * C; i += blockDim.x) {
    const int tp = i / C, c = i % C, t = tp - PAD;
    const float v = (t >= 0 && t < T && c < Da) ? x[(b * T + t) * Da + c] : 0.f;
    img[(size_t)b * Tp * C + i] = P::from_f32(v);
  }
}
// the same straight from a chain buffer: row n = (b = n / Kft, k = n % Kft) takes chains[b][k]; also t[n] = ksteps[k].t
template <class P>
__global__ void unet_chain_input_kernel(const float* chains, const dppo_step* ksteps, int Kft, int T, int Da,
                                        typename P::elem_t* img, int C, int64_t* tout) {
  const int64_t n = blockIdx.x, b = n / Kft;
  const int k = (int)(n % Kft), Tp = T + 2 * PAD, AF = T * Da;
  const float* x = chains + ((size_t)b * (Kft + 1) + k) * AF;
  for (int i = threadIdx.x; i < Tp * C; i += blockDim.x) {
    const int tp = i / C, c = i % C, t = tp - PAD;
    img[(size_t)n * Tp * C + i] = P::from_f32((t >= 0 && t < T && c < Da) ? x[t * Da + c] : 0.f);
  }
  if (threadIdx.x == 0) tout[n] = ksteps[k].t;
}
// conditioning rows for the log-prob evaluation: g[n] = [temb[t[n]] | state[n / Kft] | 0]
template <class P>
__global__ void unet_cond_rows_kernel(const float* temb, const int64_t* t, const float* state, int td, int cond, int rep,
                                      typename P::elem_t* g, int Kg, int act) {
  const int64_t n = blockIdx.x;
  const int tt = t ? (int)t[n] : 0;
  for (int c = threadIdx.x; c < Kg; c += blockDim.x) {
    float v = 0.f;
    if (c < td)
      v = temb[(size_t)tt * td + c];
    else if (c < td + cond)
      v = state[(size_t)(n / rep) * cond + (c - td)];
    if (act >= 0 && c < td + cond) v = act_f(act, v);  // the one-layer encoder applies the activation to the vector first
    g[(size_t)n * Kg + c] = P::from_f32(v);
  }
}

struct GnArgs {
  const float* src;  // conv output (+ bias) f32 [rows * Tps][lds]: row b * Tps + t, t < T valid
  int lds, Tps, T, C, G;
  const float *gamma, *beta;
  float eps;
  int act;
  int film;          // 0 none, 1 additive emb[b][c], 2 scale emb[b][c] / bias emb[b][C + c]
  const float* emb;
  int lde;
  int res;           // 0 none, 1 f32 res[b * Tps + t][c] (the 1x1 residual conv), 2 identity from an image
  const float* resf;
  int ldr;
  const void* resi;  // image [rows][Tp][ldri]
  int ldri;
  void* dst;         // image [rows][T + 2 PAD][ldd], written at channel offset coff; pad rows (all ldd channels) zeroed
  int ldd, coff, zero_pads;
  void* dst2;        // optional second destination (the skip connection's half of a concat image)
  int ldd2, coff2, zero_pads2;
  int Cw;            // width of the map in the image (C padded to a multiple of 64): channels [C, Cw) are written as zeros
};
// Conv1dBlock's GroupNorm + activation (modules.py:50-95), then FiLM (unet.py:105-113) or the block's skip sum (:116)
template <class P>
__global__ __launch_bounds__(256) void unet_gn_kernel(const GnArgs a) {
  typedef typename P::elem_t E;
  __shared__ float mean[32], rstd[32];
  const int64_t b = blockIdx.x;
  const int cg = a.C / a.G, cnt = cg * a.T;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const float* src = a.src + (size_t)b * a.Tps * a.lds;
  for (int g = w; g < a.G; g += 4) {  // a wave per group: two-pass mean / biased variance (torch.nn.GroupNorm)
    float s = 0.f;
    for (int i = lane; i < cnt; i += 64) s += src[(size_t)(i / cg) * a.lds + g * cg + (i % cg)];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float m = s / (float)cnt;
    float q = 0.f;
    for (int i = lane; i < cnt; i += 64) {
      const float dlt = src[(size_t)(i / cg) * a.lds + g * cg + (i % cg)] - m;
      q += dlt * dlt;
    }
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    if (lane == 0) mean[g] = m, rstd[g] = 1.f / sqrtf(q / (float)cnt + a.eps);
  }
  __syncthreads();
  const int Tp = a.T + 2 * PAD;
  E* dst = (E*)a.dst + (size_t)b * Tp * a.ldd;
  E* dst2 = a.dst2 ? (E*)a.dst2 + (size_t)b * Tp * a.ldd2 : nullptr;
  for (int i = threadIdx.x; i < a.T * a.Cw; i += 256) {
    const int t = i / a.Cw, c = i % a.Cw;
    float v = 0.f;
    if (c < a.C) {
      const int g = c / cg;
      v = (src[(size_t)t * a.lds + c] - mean[g]) * rstd[g] * a.gamma[c] + a.beta[c];
      v = act_f(a.act, v);
      if (a.film == 1)
        v = v + a.emb[(size_t)b * a.lde + c];
      else if (a.film == 2)
        v = a.emb[(size_t)b * a.lde + c] * v + a.emb[(size_t)b * a.lde + a.C + c];
      if (a.res == 1)
        v = v + a.resf[((size_t)b * a.Tps + t) * a.ldr + c];
      else if (a.res == 2)
        v = v + P::to_f32(((const E*)a.resi)[((size_t)b * Tp + t + PAD) * a.ldri + c]);
    }
    dst[(size_t)(t + PAD) * a.ldd + a.coff + c] = P::from_f32(v);
    if (dst2) dst2[(size_t)(t + PAD) * a.ldd2 + a.coff2 + c] = P::from_f32(v);
  }
  if (a.zero_pads)
    for (int i = threadIdx.x; i < 2 * PAD * a.ldd; i += 256) {
      const int r = i / a.ldd, c = i % a.ldd;
      dst[(size_t)(r < PAD ? r : a.T + r) * a.ldd + c] = P::from_f32(0.f);
    }
  if (dst2 && a.zero_pads2)
    for (int i = threadIdx.x; i < 2 * PAD * a.ldd2; i += 256) {
      const int r = i / a.ldd2, c = i % a.ldd2;
      dst2[(size_t)(r < PAD ? r : a.T + r) * a.ldd2 + c] = P::from_f32(0.f);
    }
}

// plain conv output -> image: src row b * Tps + m (m < Tsrc) holds `nsub` consecutive output positions of C channels
// (nsub 1: Downsample1d; 2: the even | odd phases of Upsample1d): dst[b][m * nsub + sub + PAD][coff + c]
template <class P>
__global__ __launch_bounds__(256) void unet_scatter_kernel(const float* src, int lds, int Tps, int Tsrc, int nsub, int C,
                                                           typename P::elem_t* dst, int ldd, int coff, int zero_pads) {
  const int64_t b = blockIdx.x;
  const int Tout = Tsrc * nsub, Tp = Tout + 2 * PAD, Cw = (C + 63) / 64 * 64;
  typename P::elem_t* d = dst + (size_t)b * Tp * ldd;
  for (int i = threadIdx.x; i < Tout * Cw; i += 256) {
    const int t = i / Cw, c = i % Cw, m = t / nsub, sub = t % nsub;
    d[(size_t)(t + PAD) * ldd + coff + c] = P::from_f32(c < C ? src[((size_t)b * Tps + m) * lds + sub * C + c] : 0.f);
  }
  if (zero_pads)
    for (int i = threadIdx.x; i < 2 * PAD * ldd; i += 256) {
      const int r = i / ldd, c = i % ldd;
      d[(size_t)(r < PAD ? r : Tout + r) * ldd + c] = P::from_f32(0.f);
    }
}
// the two phases of Upsample1d's transposed conv (+ its bias, which the GEMM left out: both phases share it)
template <class P>
__global__ __launch_bounds__(256) void unet_up_scatter_kernel(const float* src, int lds, int Tps, int Tsrc, int C, const float* bias,
                                              void* dstv, int ldd, int zero_pads) {
  typedef typename P::elem_t E;
  const int64_t b = blockIdx.x;
  const int Tout = 2 * Tsrc, Tp = Tout + 2 * PAD, Cw = (C + 63) / 64 * 64;
  E* dd = (E*)dstv + (size_t)b * Tp * ldd;
  for (int i = threadIdx.x; i < Tout * Cw; i += 256) {
    const int t = i / Cw, c = i % Cw, m = t / 2, sub = t % 2;
    dd[(size_t)(t + PAD) * ldd + c] = P::from_f32(c < C ? src[((size_t)b * Tps + m) * lds + sub * C + c] + bias[c] : 0.f);
  }
  if (zero_pads)
    for (int i = threadIdx.x; i < 2 * PAD * ldd; i += 256) {
      const int r = i / ldd, c = i % ldd;
      dd[(size_t)(r < PAD ? r : Tout + r) * ldd + c] = P::from_f32(0.f);
    }
}
// final 1x1 conv output rows (b * Tps + t) -> eps [rows][T][Da]
__global__ void unet_gather_kernel(const float* src, int lds, int Tps, int T, int Da, float* eps, int64_t rows) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * T * Da) return;
  const int64_t b = i / (T * Da);
  const int r = (int)(i % (T * Da)), t = r / Da, c = r % Da;
  eps[i] = src[((size_t)b * Tps + t) * lds + c];
}

// one denoising step of the sampling loop (diffusion_vpg.py:165-223, 279-311), same arithmetic as sample_chain_kernel's epilogue
struct StepArgs {
  dppo_diffusion_cfg cfg;
  dppo_step st;
  float* x;          // [B][AF] in / out
  const float* eps;  // [B][AF]
  const float* noise;
  size_t nz0;        // element offset of this step's draws in the (n_steps+1, B, AF) noise tensor / Philox counter space
  int64_t n;         // B * AF
  int AF, chain_len, last;
  float *chains, *traj;
  int lde;  // row stride of eps (0 = AF: dense)
};
__global__ void unet_step_kernel(const StepArgs a) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  const dppo_step& st = a.st;
  const float x = a.x[i];
  float eps = a.lde ? a.eps[(i / a.AF) * a.lde + i % a.AF] : a.eps[i], x0, mu;
  if (!a.cfg.use_ddim) {
    x0 = st.c0 * x - st.c1 * eps;
    if (a.cfg.has_denoised_clip) x0 = fminf(fmaxf(x0, -a.cfg.denoised_clip), a.cfg.denoised_clip);
    mu = st.c2 * x0 + st.c3 * x;
  } else {
    x0 = (x - st.c1 * eps) / st.c0;
    if (a.cfg.has_denoised_clip) {
      x0 = fminf(fmaxf(x0, -a.cfg.denoised_clip), a.cfg.denoised_clip);
      eps = (x - st.c0 * x0) / st.c1;
    }
    if (a.cfg.has_eps_clip) eps = fminf(fmaxf(eps, -a.cfg.eps_clip), a.cfg.eps_clip);
    mu = st.c2 * x0 + st.c3 * eps;
  }
  const size_t ni = a.nz0 + (size_t)i;
  float z = a.noise != nullptr ? a.noise[ni] : philox_normal(ni, a.cfg.seed_lo, a.cfg.seed_hi);
  z = fminf(fmaxf(z, -a.cfg.randn_clip), a.cfg.randn_clip);
  float xn = mu + st.std * z;
  if (st.final_clip) xn = fminf(fmaxf(xn, -a.cfg.final_clip), a.cfg.final_clip);
  a.x[i] = xn;
  const int64_t b = i / a.AF;
  const int j = (int)(i % a.AF);
  if (st.chain_slot >= 0 && a.chains != nullptr) a.chains[((size_t)b * a.chain_len + st.chain_slot) * a.AF + j] = xn;
  if (a.last) a.traj[i] = xn;
}
// x_K: the initial draw (not clipped, diffusion_vpg.py:271), stored in the chain when every step is fine-tuned
__global__ void unet_init_kernel(const float* noise, uint32_t k0, uint32_t k1, int64_t n, int AF, float* x, float* chains,
                                 int chain_len, int init_slot) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = noise != nullptr ? noise[i] : philox_normal((uint64_t)i, k0, k1);
  x[i] = v;
  if (init_slot >= 0 && chains != nullptr) chains[((size_t)(i / AF) * chain_len + init_slot) * AF + (i % AF)] = v;
}

// ---------------------------------------------------------------------------------------------------------------------
// forward: host orchestration
// ---------------------------------------------------------------------------------------------------------------------
template <class P>
struct Ws {
  void* in_img;   // [rows][T0p][64]
  void* g;        // [rows][Kg] conditioning vector (activated for the one-layer encoder)
  void *e1, *e2;  // [rows][ccmax] encoder hidden activations
  float* emb;     // [rows][ccmax] FiLM parameters
  float *conv, *res;  // conv / residual-conv outputs f32 [rows * T0p][nmax]
  void *bufA, *bufB, *bufC;  // images, sized for the largest level
  void* cat[4];   // per level: concat image [rows][Tp_l][2 C_l] (x | skip)
  float* x;       // sampler state [rows][AF]
  float* eps;     // [rows][AF]
  int64_t* tdev;  // [rows]
  size_t bytes;
};
template <class P>
void carve(const dppo_unet_desc& d, const Layout& L, int64_t rows, char* base, Ws<P>& W) {
  const size_t ES = P::ESIZE;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    off = al(off);
    void* p = base ? base + off : nullptr;
    off += bytes + 65536;  // slack: the last GEMM rows read a few image rows past the end of their buffer
    return p;
  };
  const int T0 = d.horizon_steps, T0p = T0 + 2 * PAD, nl = d.n_levels;
  int cmax = 64, ccmax = 64;
  for (int i = 1; i <= nl; ++i) cmax = rup(L.dims[i], 64) > cmax ? rup(L.dims[i], 64) : cmax;
  ccmax = 2 * cmax;
  size_t img_max = 0;
  for (int l = 0; l < nl; ++l) {
    const size_t b = (size_t)((T0 >> l) + 2 * PAD) * (size_t)(2 * rup(L.dims[l + 1], 64));
    img_max = b > img_max ? b : img_max;
  }
  W.in_img = take((size_t)rows * T0p * 64 * ES);
  W.g = take((size_t)rows * L.Kg * ES);
  W.e1 = take((size_t)rows * ccmax * ES);
  W.e2 = take((size_t)rows * ccmax * ES);
  W.emb = (float*)take((size_t)rows * ccmax * 4);
  W.conv = (float*)take((size_t)rows * T0p * ccmax * 4);
  W.res = (float*)take((size_t)rows * T0p * cmax * 4);
  W.bufA = take((size_t)rows * img_max * ES);
  W.bufB = take((size_t)rows * img_max * ES);
  W.bufC = take((size_t)rows * img_max * ES);
  for (int l = 0; l < 4; ++l)
    W.cat[l] = l < nl ? take((size_t)rows * ((T0 >> l) + 2 * PAD) * 2 * rup(L.dims[l + 1], 64) * ES) : nullptr;
  W.x = (float*)take((size_t)rows * T0 * d.action_dim * 4);
  W.eps = (float*)take((size_t)rows * T0 * d.action_dim * 4);
  W.tdev = (int64_t*)take((size_t)rows * 8);
  W.bytes = al(off);
}

template <class P>
struct Runner {
  const dppo_unet_desc& d;
  const Layout& L;
  const float* prm;
  const char* pk;
  Ws<P>& W;
  int64_t rows;
  hipStream_t s;
  // sampler: FiLM parameters of every block precomputed for all steps (they depend on (t_k, obs) only, not on x): film[b] is
  // block b's table in forward order, this step's rows start at film_row0; null = run the encoders in place
  float* const* film = nullptr;
  int64_t film_row0 = 0;
  int block_idx = 0;

  // GEMM over image rows: start row `r0` of the padded image, row stride `stride` images rows, K = Kp
  void gemm(const Img& in, int r0, int stride, const void* Wp, int N, int Kp, const float* bias, float* out, int ldo) {
    GemmNT g;
    memset(&g, 0, sizeof(g));
    g.X = (const char*)in.p + (size_t)r0 * in.C * P::ESIZE;
    g.ldx = stride * in.C, g.M = (int)(rows * in.Tp() / stride), g.N = N, g.Kp = Kp, g.W = Wp, g.ldw = Kp, g.bias = bias;
    g.out_f32 = out, g.ldo32 = ldo;
    launch_gemm_nt<P>(g, s);
  }
  void conv(const Conv& c, const Img& in, float* out, int ldo) {
    gemm(in, PAD - c.ks / 2, 1, pk + c.pk, c.co, c.Kp, prm + c.b, out, ldo);
  }
  // FiLM parameters of one block from the conditioning vector (unet.py:76-90,102): `n` rows of g -> out [n][rup(cc, 16)]
  void encoder(const ResBlock& r, const void* gsrc = nullptr, int64_t n = -1, float* out = nullptr, void* e1 = nullptr,
               void* e2 = nullptr) {
    GemmNT g;
    const void* x = gsrc ? gsrc : W.g;
    if (n < 0) n = rows;
    int K = L.Kg;
    for (int i = 0; i < r.n_enc; ++i) {
      const Lin& l = r.enc[i];
      memset(&g, 0, sizeof(g));
      g.X = x, g.ldx = K, g.M = (int)n, g.N = l.out, g.Kp = l.Kp, g.W = pk + l.pk, g.ldw = l.Kp, g.bias = prm + l.b;
      if (i + 1 < r.n_enc) {
        g.out_act = i == 0 ? (e1 ? e1 : W.e1) : (e2 ? e2 : W.e2), g.ldo = rup(l.out, 64), g.act = d.act;
        x = g.out_act, K = g.ldo;
      } else {
        g.out_f32 = out ? out : W.emb, g.ldo32 = rup(r.cc, 16);
      }
      launch_gemm_nt<P>(g, s);
      if (i + 1 < r.n_enc && rup(l.out, 16) < g.ldo)  // the next GEMM's K runs over the padded width
        launch_zero_cols<P>(g.out_act, (int)n, rup(l.out, 16), g.ldo, g.ldo, s);
    }
  }
  // ResidualBlock1D.forward (unet.py:100-118): in -> out image (channel offset coff of an image of width ldd), optionally
  // also into dst2 (the skip's half of a concat image)
  void resblock(const ResBlock& r, const Img& in, void* mid_img, void* out_img, int ldd, int coff, int zero_pads,
                void* dst2 = nullptr, int ldd2 = 0, int coff2 = 0, int zero2 = 0) {
    const int T = in.T, Tp = in.Tp(), ldc = rup(r.co, 16), Cw = rup(r.co, 64);
    const float* emb = W.emb;
    if (film) emb = film[block_idx] + (size_t)film_row0 * rup(r.cc, 16);
    else encoder(r);
    ++block_idx;
    conv(r.c1, in, W.conv, ldc);
    GnArgs a;
    memset(&a, 0, sizeof(a));
    a.src = W.conv, a.lds = ldc, a.Tps = Tp, a.T = T, a.C = r.co, a.Cw = Cw, a.G = d.n_groups, a.gamma = prm + r.n1.g;
    a.beta = prm + r.n1.b, a.eps = d.groupnorm_eps, a.act = d.act, a.film = d.cond_predict_scale ? 2 : 1, a.emb = emb;
    a.lde = rup(r.cc, 16), a.dst = mid_img, a.ldd = Cw, a.coff = 0, a.zero_pads = 1;
    hipLaunchKernelGGL((unet_gn_kernel<P>), dim3((unsigned)rows), dim3(256), 0, s, a);
    Img mid{mid_img, T, Cw};
    conv(r.c2, mid, W.conv, ldc);
    memset(&a, 0, sizeof(a));
    a.src = W.conv, a.lds = ldc, a.Tps = Tp, a.T = T, a.C = r.co, a.Cw = Cw, a.G = d.n_groups, a.gamma = prm + r.n2.g;
    a.beta = prm + r.n2.b, a.eps = d.groupnorm_eps, a.act = d.act;
    if (r.has_res) {
      gemm(in, PAD, 1, pk + r.res.pk, r.co, r.res.Kp, prm + r.res.b, W.res, ldc);
      a.res = 1, a.resf = W.res, a.ldr = ldc;
    } else {
      a.res = 2, a.resi = in.p, a.ldri = in.C;
    }
    a.dst = out_img, a.ldd = ldd, a.coff = coff, a.zero_pads = zero_pads;
    a.dst2 = dst2, a.ldd2 = ldd2, a.coff2 = coff2, a.zero_pads2 = zero2;
    hipLaunchKernelGGL((unet_gn_kernel<P>), dim3((unsigned)rows), dim3(256), 0, s, a);
  }

  // Unet1D.forward (unet.py:264-327) from the prepared input image + conditioning rows; eps -> out [rows][T][Da]
  // (every map sits in its image padded to a multiple of 64 channels, the padding written as zeros: Img.C is that stride)
  void forward(float* out) {
    const int nl = d.n_levels, T0 = d.horizon_steps;
    block_idx = 0;
    Img cur{W.in_img, T0, 64};
    void* bufs[3] = {W.bufA, W.bufB, W.bufC};
    auto pick = [&](const void* x0, const void* x1) {  // a scratch image that is neither of the two in use
      for (void* b : bufs)
        if (b != x0 && b != x1) return b;
      return (void*)nullptr;
    };
    for (int i = 0; i < nl; ++i) {
      const int C = L.dims[i + 1], Cw = rup(C, 64), T = T0 >> i;
      void* m = pick(cur.p, nullptr);
      void* o = pick(cur.p, m);
      resblock(L.down[2 * i], cur, m, o, Cw, 0, 1);
      Img a{o, T, Cw};
      m = pick(a.p, nullptr);
      o = pick(a.p, m);
      // second block of the level: its output is the skip connection -> also the upper half of the level's concat image
      const bool skip_used = i >= 1;  // up_modules has n_levels - 1 entries: the level-0 skip is never popped (:300-308)
      resblock(L.down[2 * i + 1], a, m, o, Cw, 0, 1, skip_used ? W.cat[i] : nullptr, 2 * Cw, Cw, 1);
      cur = Img{o, T, Cw};
      if (i < nl - 1) {  // Downsample1d: Conv1d(C, C, 3, stride 2, padding 1)
        const Conv& c = L.downs[i];
        gemm(cur, PAD - 1, 2, pk + c.pk, c.co, c.Kp, prm + c.b, W.conv, rup(C, 16));
        void* dn = pick(cur.p, nullptr);
        hipLaunchKernelGGL((unet_scatter_kernel<P>), dim3((unsigned)rows), dim3(256), 0, s, W.conv, rup(C, 16),
                           cur.Tp() / 2, T / 2, 1, C, (typename P::elem_t*)dn, Cw, 0, 1);
        cur = Img{dn, T / 2, Cw};
      }
    }
    for (int i = 0; i < 2; ++i) {
      void* m = pick(cur.p, nullptr);
      const bool to_cat = i == 1 && nl >= 2;  // the last mid block feeds cat(x, skip) of the first up level
      void* o = to_cat ? W.cat[nl - 1] : pick(cur.p, m);
      const int Cw = rup(L.dims[nl], 64);
      resblock(L.mid[i], cur, m, o, to_cat ? 2 * Cw : Cw, 0, to_cat ? 0 : 1);
      cur = Img{o, cur.T, to_cat ? 2 * Cw : Cw};
    }
    for (int j = 0; j < nl - 1; ++j) {
      const int din = L.dims[nl - 1 - j], dinw = rup(din, 64);
      void* m = pick(cur.p, nullptr);
      void* o = pick(cur.p, m);
      resblock(L.up[2 * j], cur, m, o, dinw, 0, 1);
      Img a{o, cur.T, dinw};
      m = pick(a.p, nullptr);
      o = pick(a.p, m);
      resblock(L.up[2 * j + 1], a, m, o, dinw, 0, 1);
      Img b2{o, cur.T, dinw};
      // Upsample1d: ConvTranspose1d(din, din, 4, 2, 1) as one GEMM (even | odd phase)
      const Conv& c = L.ups[j];
      gemm(b2, PAD - 1, 1, pk + c.pk, 2 * din, c.Kp, nullptr, W.conv, rup(2 * din, 16));
      const int lvl = nl - 2 - j;  // the level the upsampled map lands on
      const bool to_cat = lvl >= 1;  // another up level follows: write the lower half of that level's concat image
      void* up = to_cat ? W.cat[lvl] : pick(b2.p, nullptr);
      const int ldd = to_cat ? 2 * dinw : dinw;
      up_bias_scatter(c, b2, up, ldd, din, to_cat ? 0 : 1);
      cur = Img{up, b2.T * 2, ldd};
    }
    // final_conv: Conv1dBlock(dim, dim) + Conv1d(dim, action_dim, 1)
    {
      const int C = d.dim, Cw = rup(C, 64), T = cur.T, ldc = rup(C, 16);
      conv(L.fin, cur, W.conv, ldc);
      GnArgs a;
      memset(&a, 0, sizeof(a));
      void* o = pick(cur.p, nullptr);
      a.src = W.conv, a.lds = ldc, a.Tps = cur.Tp(), a.T = T, a.C = C, a.Cw = Cw, a.G = d.n_groups, a.gamma = prm + L.fin_n.g;
      a.beta = prm + L.fin_n.b, a.eps = d.groupnorm_eps, a.act = d.act, a.dst = o, a.ldd = Cw, a.zero_pads = 1;
      hipLaunchKernelGGL((unet_gn_kernel<P>), dim3((unsigned)rows), dim3(256), 0, s, a);
      Img f{o, T, Cw};
      gemm(f, PAD, 1, pk + L.fin_out.pk, d.action_dim, L.fin_out.Kp, prm + L.fin_out.b, W.res, 16 * ((d.action_dim + 15) / 16));
      const int64_t n = rows * T * d.action_dim;
      hipLaunchKernelGGL(unet_gather_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, W.res,
                         16 * ((d.action_dim + 15) / 16), f.Tp(), T, d.action_dim, out, rows);
    }
  }
  // bias of the transposed conv + scatter of the two phases (the GEMM above ran without bias: both phases share it)
  void up_bias_scatter(const Conv& c, const Img& src_img, void* dst, int ldd, int C, int zero_pads) {
    hipLaunchKernelGGL((unet_up_scatter_kernel<P>), dim3((unsigned)rows), dim3(256), 0, s, W.conv, rup(2 * C, 16),
                       src_img.Tp(), src_img.T, C, prm + c.b, dst, ldd, zero_pads);
  }
};


// =====================================================================================================================
// training: forward with a tape, backward to every parameter gradient
// =====================================================================================================================
// Gradients of activation images travel in the GEMMs' f32 output-row format: G[(b * Tp + t) * ld + coff + c], valid for
// t < T (the other rows hold garbage nobody reads).  A gradient handed to a GEMM as an OPERAND is first turned into a
// padded elem image with ZERO pads (gn_bwd / rows_to_img), which then serves both the data-gradient GEMM (the same
// contiguous-window trick as the forward, flipped kernel) and, shifted by PAD rows, the A operand of the weight gradient.
struct GradSrc {
  const float* p;
  int ld, coff, Tp;
};

template <class P>
__global__ __launch_bounds__(256) void unet_train_input_kernel(const UnetTrainIO io, int T, int Da, int cond, int td,
                                                               const float* temb, typename P::elem_t* img, int C,
                                                               typename P::elem_t* g, typename P::elem_t* graw, int Kg,
                                                               int act) {
  const int64_t n = blockIdx.x;
  const int b = io.brow[n], k = io.krow[n], AF = T * Da, Tp = T + 2 * PAD;
  const float* x = io.gathered ? io.chains + (size_t)b * 2 * AF : io.chains + ((size_t)b * (io.Kft + 1) + k) * AF;
  for (int i = threadIdx.x; i < Tp * C; i += 256) {
    const int tp = i / C, c = i % C, t = tp - PAD;
    img[(size_t)n * Tp * C + i] = P::from_f32((t >= 0 && t < T && c < Da) ? x[t * Da + c] : 0.f);
  }
  const int tt = io.ksteps[k].t;
  for (int c = threadIdx.x; c < Kg; c += 256) {
    float v = 0.f;
    if (c < td)
      v = temb[(size_t)tt * td + c];
    else if (c < td + cond)
      v = io.obs[(size_t)b * cond + (c - td)];
    graw[(size_t)n * Kg + c] = P::from_f32(v);
    if (act >= 0 && c < td + cond) v = act_f(act, v);
    g[(size_t)n * Kg + c] = P::from_f32(v);
  }
}

// d_eps elem [N][ldde] (column t * Da + c) -> padded image [N][Tp][64]
template <class P>
__global__ __launch_bounds__(256) void unet_deps_img_kernel(const typename P::elem_t* de, int ldde, int T, int Da,
                                                            typename P::elem_t* img) {
  const int64_t n = blockIdx.x;
  const int Tp = T + 2 * PAD;
  for (int i = threadIdx.x; i < Tp * 64; i += 256) {
    const int tp = i / 64, c = i % 64, t = tp - PAD;
    img[(size_t)n * Tp * 64 + i] = (t >= 0 && t < T && c < Da) ? de[(size_t)n * ldde + t * Da + c] : P::from_f32(0.f);
  }
}

struct GnBwdArgs {
  const float* u;  // conv output (pre-GroupNorm) rows [rows * Tps][lds]
  int lds, Tps, T, C, G;
  const float *gamma, *beta;
  float eps;
  int act;
  GradSrc up[2];   // upstream gradient(s) w.r.t. this half's OUTPUT (after FiLM for the first half)
  int n_up;
  int film;        // 0 none, 1 additive, 2 scale / bias
  const float* emb;
  int lde;
  float* demb;     // [rows][lde]: d loss / d FiLM parameters (film != 0)
  void* dU;        // out: padded elem image [rows][Tp][Cw], zero pads and zero channels [C, Cw)
  float* dgb;      // out: per-sample [rows][2 C]: d gamma | d beta contributions
  int Cw;
};
template <class P>
__global__ __launch_bounds__(256) void unet_gn_bwd_kernel(const GnBwdArgs a) {
  typedef typename P::elem_t E;
  __shared__ float mean[32], rstd[32], s1[32], s2[32];
  const int64_t b = blockIdx.x;
  const int cg = a.C / a.G, cnt = cg * a.T, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const float* u = a.u + (size_t)b * a.Tps * a.lds;
  auto upstream = [&](int t, int c) {
    float v = a.up[0].p[((size_t)b * a.up[0].Tp + t) * a.up[0].ld + a.up[0].coff + c];
    if (a.n_up > 1) v += a.up[1].p[((size_t)b * a.up[1].Tp + t) * a.up[1].ld + a.up[1].coff + c];
    return v;
  };
  // d loss / d z (z = GroupNorm output, the activation's input) of element (t, c), given the group statistics
  auto dz_of = [&](int t, int c, float m, float r, float& xhat) {
    xhat = (u[(size_t)t * a.lds + c] - m) * r;
    const float z = xhat * a.gamma[c] + a.beta[c];
    float dh = upstream(t, c);
    if (a.film == 2) dh *= a.emb[(size_t)b * a.lde + c];
    return dh * act_grad_f(a.act, z);
  };
  for (int g = w; g < a.G; g += 4) {
    float s = 0.f;
    for (int i = lane; i < cnt; i += 64) s += u[(size_t)(i / cg) * a.lds + g * cg + (i % cg)];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float m = s / (float)cnt;
    float q = 0.f;
    for (int i = lane; i < cnt; i += 64) {
      const float dlt = u[(size_t)(i / cg) * a.lds + g * cg + (i % cg)] - m;
      q += dlt * dlt;
    }
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float r = 1.f / sqrtf(q / (float)cnt + a.eps);
    float t1 = 0.f, t2 = 0.f;  // sum dxhat, sum dxhat * xhat over the group
    for (int i = lane; i < cnt; i += 64) {
      const int t = i / cg, c = g * cg + (i % cg);
      float xh;
      const float dxh = dz_of(t, c, m, r, xh) * a.gamma[c];
      t1 += dxh, t2 += dxh * xh;
    }
    for (int o = 32; o > 0; o >>= 1) t1 += __shfl_xor(t1, o), t2 += __shfl_xor(t2, o);
    if (lane == 0) mean[g] = m, rstd[g] = r, s1[g] = t1 / (float)cnt, s2[g] = t2 / (float)cnt;
  }
  __syncthreads();
  const int Tp = a.T + 2 * PAD;
  E* dU = (E*)a.dU + (size_t)b * Tp * a.Cw;
  for (int i = threadIdx.x; i < a.T * a.Cw; i += 256) {
    const int t = i / a.Cw, c = i % a.Cw;
    float v = 0.f;
    if (c < a.C) {
      const int g = c / cg;
      float xh;
      const float dxh = dz_of(t, c, mean[g], rstd[g], xh) * a.gamma[c];
      v = rstd[g] * (dxh - s1[g] - xh * s2[g]);
    }
    dU[(size_t)(t + PAD) * a.Cw + c] = P::from_f32(v);
  }
  for (int i = threadIdx.x; i < 2 * PAD * a.Cw; i += 256) {
    const int r = i / a.Cw, c = i % a.Cw;
    dU[(size_t)(r < PAD ? r : a.T + r) * a.Cw + c] = P::from_f32(0.f);
  }
  // per-channel sums over t: d gamma, d beta, and the FiLM parameter gradients
  for (int c = threadIdx.x; c < a.C; c += 256) {
    const int g = c / cg;
    float dg = 0.f, db = 0.f, dsc = 0.f, dbi = 0.f;
    for (int t = 0; t < a.T; ++t) {
      float xh;
      const float dz = dz_of(t, c, mean[g], rstd[g], xh);
      dg += dz * xh, db += dz;
      if (a.film != 0) {
        const float dout = upstream(t, c);
        dbi += dout;
        if (a.film == 2) dsc += dout * act_f(a.act, xh * a.gamma[c] + a.beta[c]);
      }
    }
    a.dgb[(size_t)b * 2 * a.C + c] = dg;
    a.dgb[(size_t)b * 2 * a.C + a.C + c] = db;
    if (a.film == 1) a.demb[(size_t)b * a.lde + c] = dbi;
    if (a.film == 2) a.demb[(size_t)b * a.lde + c] = dsc, a.demb[(size_t)b * a.lde + a.C + c] = dbi;
  }
}

// sum of up to two gradient sources -> padded elem image with zero pads (an operand for the residual conv's GEMMs)
template <class P>
__global__ __launch_bounds__(256) void unet_rows_to_img_kernel(GradSrc a0, GradSrc a1, int n_src, int T, int C,
                                                               typename P::elem_t* img) {
  const int64_t b = blockIdx.x;
  const int Tp = T + 2 * PAD, Cw = (C + 63) / 64 * 64;
  typename P::elem_t* d = img + (size_t)b * Tp * Cw;
  for (int i = threadIdx.x; i < Tp * Cw; i += 256) {
    const int tp = i / Cw, c = i % Cw, t = tp - PAD;
    float v = 0.f;
    if (t >= 0 && t < T && c < C) {
      v = a0.p[((size_t)b * a0.Tp + t) * a0.ld + a0.coff + c];
      if (n_src > 1) v += a1.p[((size_t)b * a1.Tp + t) * a1.ld + a1.coff + c];
    }
    d[i] = P::from_f32(v);
  }
}
// the same into the FORWARD GEMM's output-row format of a strided / phase layer (the A operand of its weight gradient):
//   mode 1 (Downsample1d): rows b * Tps + t', t' < T: dst[.][c] = G(b, t', c); zero rows beyond
//   mode 2 (Upsample1d)  : rows b * Tps + m, m < T / 2: dst[.][ph * C + c] = G(b, 2 m + ph, c)
template <class P>
__global__ __launch_bounds__(256) void unet_rows_fmt_kernel(GradSrc a0, GradSrc a1, int n_src, int T, int C, int mode, int Tps,
                                                            typename P::elem_t* dst) {
  const int64_t b = blockIdx.x;
  const int W = mode == 2 ? 2 * C : C;
  typename P::elem_t* d = dst + (size_t)b * Tps * W;
  for (int i = threadIdx.x; i < Tps * W; i += 256) {
    const int r = i / W, cc = i % W;
    int t = r, c = cc;
    if (mode == 2) t = 2 * r + cc / C, c = cc % C;
    float v = 0.f;
    if (t < T) {
      v = a0.p[((size_t)b * a0.Tp + t) * a0.ld + a0.coff + c];
      if (n_src > 1) v += a1.p[((size_t)b * a1.Tp + t) * a1.ld + a1.coff + c];
    }
    d[i] = P::from_f32(v);
  }
}
// two-phase data gradient of Downsample1d: src rows b * Tps + m hold [dx(2m) | dx(2m+1)] -> f32 rows of the big image
__global__ __launch_bounds__(256) void unet_phase_rows_kernel(const float* src, int lds, int Tps, int Tsmall, int C, float* dst,
                                                              int Tpd) {
  const int64_t b = blockIdx.x;
  for (int i = threadIdx.x; i < 2 * Tsmall * C; i += 256) {
    const int t = i / C, c = i % C, m = t / 2, ph = t % 2;
    dst[((size_t)b * Tpd + t) * C + c] = src[((size_t)b * Tps + m) * lds + ph * C + c];
  }
}
// deterministic column sums of an f32 matrix: out[c] (+)= sum_r A[r][c]
__global__ __launch_bounds__(256) void unet_colsum1_kernel(const float* A, int64_t M, int N, int lda, float* part, int blocks) {
  const int64_t per = (M + blocks - 1) / blocks, r0 = (int64_t)blockIdx.x * per, r1 = r0 + per < M ? r0 + per : M;
  for (int c = threadIdx.x; c < N; c += 256) {
    float s = 0.f;
    for (int64_t r = r0; r < r1; ++r) s += A[r * lda + c];
    part[(size_t)blockIdx.x * N + c] = s;
  }
}
__global__ __launch_bounds__(256) void unet_colsum2_kernel(const float* part, int blocks, int N, float* out, int accumulate) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= N) return;
  float s = 0.f;
  for (int b = 0; b < blocks; ++b) s += part[(size_t)b * N + c];
  out[c] = accumulate ? out[c] + s : s;
}
// bias gradient = column sums over the rows of a padded elem image (pads are zero)
template <class P>
__global__ __launch_bounds__(256) void unet_colsumE_kernel(const typename P::elem_t* A, int64_t M, int N, int lda, float* part,
                                                           int blocks) {
  const int64_t per = (M + blocks - 1) / blocks, r0 = (int64_t)blockIdx.x * per, r1 = r0 + per < M ? r0 + per : M;
  for (int c = threadIdx.x; c < N; c += 256) {
    float s = 0.f;
    for (int64_t r = r0; r < r1; ++r) s += P::to_f32(A[r * lda + c]);
    part[(size_t)blockIdx.x * N + c] = s;
  }
}
// packed weight gradient [N1][ld] -> the flat layout: conv w[co][ci][k] <- dWp[co][k * cip + ci]; linear: ks = 1
__global__ void unet_unpack_conv_kernel(const float* dWp, int ld, int co, int ci, int ks, int cip, int split, float* out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)co * ci * ks) return;
  const int k = (int)(i % ks), c = (int)((i / ks) % ci), o = (int)(i / ((size_t)ks * ci));
  out[i] = dWp[(size_t)o * ld + k * cip + chan_pos(c, split)];
}
// ConvTranspose1d w[ci][co][4] <- dW2[phase * C + co][slot * C + ci] (see pack_convT_kernel)
__global__ void unet_unpack_convT_kernel(const float* dW2, int ld, int ch, int chp, float* out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)ch * ch * 4) return;
  const int k = (int)(i % 4), co = (int)((i / 4) % ch), ci = (int)(i / ((size_t)4 * ch));
  const int phase = (k == 1 || k == 3) ? 0 : 1;
  const int slot = k == 1 ? 1 : (k == 3 ? 0 : (k == 0 ? 2 : 1));
  out[i] = dW2[(size_t)(phase * ch + co) * ld + slot * chp + ci];
}
// G[k][j] = sum over rows with krow == k of dg[row][j] (j < td): one block per k, fixed order
__global__ __launch_bounds__(256) void unet_temb_segsum_kernel(const float* dg, int ld, const int32_t* krow, int64_t N, int td,
                                                               float* G) {
  __shared__ float red[256];
  const int k = blockIdx.x;
  for (int j = 0; j < td; ++j) {
    float s = 0.f;
    for (int64_t n = threadIdx.x; n < N; n += 256)
      if (krow[n] == k) s += dg[n * ld + j];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) G[(size_t)k * td + j] = red[0];
    __syncthreads();
  }
}
// backward of time_mlp (Linear(d, 4d) -> Mish -> Linear(4d, d)) from G[k][d] = d loss / d temb[t_k]; one block
__global__ __launch_bounds__(256) void unet_time_bwd_kernel(const float* w1, const float* b1, const float* w2, const float* G,
                                                            const dppo_step* ksteps, int Kft, int td, float* gw1, float* gb1,
                                                            float* gw2, float* gb2) {
  extern __shared__ float sh[];  // per k: e0[td], z1[4td], dz1[4td]
  const int H = 4 * td, per = td + 2 * H, tid = threadIdx.x;
  for (int k = 0; k < Kft; ++k) {
    float* e0 = sh + k * per;
    for (int j = tid; j < td; j += 256) e0[j] = sinus(ksteps[k].t, j, td);
  }
  __syncthreads();
  for (int i = tid; i < Kft * H; i += 256) {
    const int k = i / H, o = i % H;
    const float* e0 = sh + k * per;
    float s = b1[o];
    for (int j = 0; j < td; ++j) s += w1[o * td + j] * e0[j];
    sh[k * per + td + o] = s;
  }
  __syncthreads();
  for (int i = tid; i < Kft * H; i += 256) {  // dz1 = (W2^T G[k]) * mish'(z1)
    const int k = i / H, o = i % H;
    float s = 0.f;
    for (int j = 0; j < td; ++j) s += w2[j * H + o] * G[k * td + j];
    sh[k * per + td + H + o] = s * mish_grad_f(sh[k * per + td + o]);
  }
  __syncthreads();
  for (int i = tid; i < td * H; i += 256) {  // gw2[j][o] = sum_k G[k][j] mish(z1[k][o])
    const int j = i / H, o = i % H;
    float s = 0.f;
    for (int k = 0; k < Kft; ++k) s += G[k * td + j] * mish_f(sh[k * per + td + o]);
    gw2[i] = s;
  }
  for (int j = tid; j < td; j += 256) {
    float s = 0.f;
    for (int k = 0; k < Kft; ++k) s += G[k * td + j];
    gb2[j] = s;
  }
  for (int i = tid; i < H * td; i += 256) {  // gw1[o][j] = sum_k dz1[k][o] e0[k][j]
    const int o = i / td, j = i % td;
    float s = 0.f;
    for (int k = 0; k < Kft; ++k) s += sh[k * per + td + H + o] * sh[k * per + j];
    gw1[i] = s;
  }
  for (int o = tid; o < H; o += 256) {
    float s = 0.f;
    for (int k = 0; k < Kft; ++k) s += sh[k * per + td + H + o];
    gb1[o] = s;
  }
}

// out[r][c] = src[r][col0 + c]
__global__ void unet_copy_cols_kernel(const float* src, int ld, int col0, int ncols, int64_t rows, float* out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < (size_t)rows * ncols) out[i] = src[(i / ncols) * ld + col0 + i % ncols];
}

}  // namespace

template <class P>
struct UnetTrainer {
  typedef typename P::elem_t E;
  dppo_unet_desc d;
  Layout L;
  const float* prm;
  const char* pk;
  int64_t rows;
  hipStream_t s;
  char* base;
  size_t off, cap;
  bool dry;
  UnetTrainIO io;
  // tape
  struct RBTape {
    const ResBlock* r;
    Img in;
    float *u1, *u2, *emb;
    Img mid;
    void *ez[2], *ea[2];  // encoder pre-activations / activations
    int T;
  };
  std::vector<RBTape> tape;  // forward order
  struct LvlTape {
    Img skip_in;  // input of Downsample1d / Upsample1d
  };
  Img in_img, down_in[4], up_in[4], fin_in, fin_mid;
  float* fin_u;
  void* g;      // conditioning rows [rows][Kg] (activated for the one-layer encoder)
  void* graw;   // the same before the activation
  float* eps;   // [rows][AF]
  float* dgacc; // [rows][Kg-ish] accumulated d loss / d g (time-embedding columns)
  // scratch shared by the backward
  float *tmpA, *tmpB, *tmpC, *slab, *dwp, *part, *dgb;
  void *imgA, *imgB;
  size_t tmp_floats, slab_floats;

  void* take(size_t bytes) {
    off = al(off);
    void* p = dry ? nullptr : base + off;
    off += bytes + 65536;
    return p;
  }
  Img new_img(int T, int C) { return Img{take((size_t)rows * (T + 2 * PAD) * C * P::ESIZE), T, C}; }
  float* new_f32(size_t n) { return (float*)take(n * 4); }

  // ------------------------------------------------------------------------------------------------ GEMM wrappers
  void gemm(const void* X, int ldx, int M, const void* Wp, int N, int Kp, const float* bias, float* out, int ldo,
            const float* res = nullptr, int ldres = 0) {
    if (dry) return;
    GemmNT g2;
    memset(&g2, 0, sizeof(g2));
    g2.X = X, g2.ldx = ldx, g2.M = M, g2.N = N, g2.Kp = Kp, g2.W = Wp, g2.ldw = Kp, g2.bias = bias, g2.out_f32 = out;
    g2.ldo32 = ldo, g2.res = res, g2.ldres = ldres;
    launch_gemm_nt<P>(g2, s);
  }
  const char* imgrow(const Img& im, int r0) const { return (const char*)im.p + (size_t)r0 * im.C * P::ESIZE; }
  // conv forward / data gradient over an image: rows = rows * Tp / stride
  void conv_gemm(const Img& in, int r0, int stride, const void* Wp, int N, int Kp, const float* bias, float* out, int ldo,
                 const float* res = nullptr, int ldres = 0) {
    gemm(dry ? nullptr : imgrow(in, r0), stride * in.C, (int)(rows * in.Tp() / stride), Wp, N, Kp, bias, out, ldo, res, ldres);
  }
  // dWp[N1][N2] = A^T . B over M rows (A [M][lda] elem, B rows may overlap: width N2 over stride ldb); -> dwp (f32 [N1][N2])
  void wgrad(const void* A, int lda, int N1, const void* Bm, int ldb, int N2, int64_t M) {
    if (dry) return;
    // small outputs (1-20 tiles of 128 x 128) over a few thousand rows: split the rows until ~64 workgroups exist (more
    // splits cost more in slab writes and the reduction than they gain: 256 workgroups measured 13 % slower on the update)
    const int64_t tiles = (int64_t)((N1 + 127) / 128) * ((N2 + 127) / 128);
    int64_t splits = (64 + tiles - 1) / tiles;
    if (splits > (M + 255) / 256) splits = (M + 255) / 256;
    if (splits > 16) splits = 16;
    if (splits < 1) splits = 1;
    while (splits > 1 && (size_t)splits * N1 * N2 > slab_floats) --splits;
    int64_t rps = ((M + splits - 1) / splits + 63) / 64 * 64;
    splits = (M + rps - 1) / rps;
    GemmTN t;
    memset(&t, 0, sizeof(t));
    t.A = A, t.B = Bm, t.M = (int)M, t.N1 = N1, t.N2 = N2, t.lda = lda, t.ldb = ldb, t.ncol_a = N1, t.ncol_b = N2;
    t.slab = slab, t.ldc = N2, t.splits = (int)splits, t.rows_per_split = (int)rps;
    launch_gemm_tn<P>(t, s);
    launch_slab_reduce_2d(slab, (int)splits, N1, N2, N2, dwp, N2, 1.f, s);
  }
  static int colsum_blocks(int64_t) { return 64; }
  void colsum_f32(const float* A, int64_t M, int N, int lda, float* out, int accumulate = 0) {
    if (dry) return;
    const int blocks = colsum_blocks(M);
    hipLaunchKernelGGL(unet_colsum1_kernel, dim3(blocks), dim3(256), 0, s, A, M, N, lda, part, blocks);
    hipLaunchKernelGGL(unet_colsum2_kernel, dim3((N + 255) / 256), dim3(256), 0, s, part, blocks, N, out, accumulate);
  }
  void colsum_img(const void* A, int64_t M, int N, int lda, float* out) {
    if (dry) return;
    const int blocks = colsum_blocks(M);
    hipLaunchKernelGGL((unet_colsumE_kernel<P>), dim3(blocks), dim3(256), 0, s, (const E*)A, M, N, lda, part, blocks);
    hipLaunchKernelGGL(unet_colsum2_kernel, dim3((N + 255) / 256), dim3(256), 0, s, part, blocks, N, out, 0);
  }

  // ------------------------------------------------------------------------------------------------ forward
  void encoder_fwd(RBTape& tp) {
    const ResBlock& r = *tp.r;
    const void* x = g;
    int K = L.Kg;
    tp.emb = new_f32((size_t)rows * rup(r.cc, 16));
    for (int i = 0; i < r.n_enc; ++i) {
      const Lin& l = r.enc[i];
      if (i + 1 < r.n_enc) {
        tp.ez[i] = take((size_t)rows * rup(l.out, 64) * P::ESIZE);
        tp.ea[i] = take((size_t)rows * rup(l.out, 64) * P::ESIZE);
      }
      if (!dry) {
        GemmNT g2;
        memset(&g2, 0, sizeof(g2));
        g2.X = x, g2.ldx = K, g2.M = (int)rows, g2.N = l.out, g2.Kp = l.Kp, g2.W = pk + l.pk, g2.ldw = l.Kp, g2.bias = prm + l.b;
        if (i + 1 < r.n_enc)
          g2.out_pre = tp.ez[i], g2.out_act = tp.ea[i], g2.ldo = rup(l.out, 64), g2.act = d.act;
        else
          g2.out_f32 = tp.emb, g2.ldo32 = rup(r.cc, 16);
        launch_gemm_nt<P>(g2, s);
        if (i + 1 < r.n_enc && rup(l.out, 16) < rup(l.out, 64)) {
          launch_zero_cols<P>(tp.ez[i], (int)rows, rup(l.out, 16), rup(l.out, 64), rup(l.out, 64), s);
          launch_zero_cols<P>(tp.ea[i], (int)rows, rup(l.out, 16), rup(l.out, 64), rup(l.out, 64), s);
        }
      }
      if (i + 1 < r.n_enc) x = tp.ea[i], K = rup(l.out, 64);
    }
  }
  // one block; `out` is an image of width ldd the block writes at channel offset coff (and optionally a second place)
  void resblock_fwd(const ResBlock& r, const Img& in, void* out, int ldd, int coff, int zero_pads, void* dst2 = nullptr,
                    int ldd2 = 0, int coff2 = 0, int zero2 = 0) {
    RBTape tp;
    memset(&tp, 0, sizeof(tp));
    tp.r = &r, tp.in = in, tp.T = in.T;
    const int T = in.T, Tp = in.Tp(), ldc = rup(r.co, 16), Cw = rup(r.co, 64);
    encoder_fwd(tp);
    tp.u1 = new_f32((size_t)rows * Tp * ldc);
    tp.u2 = new_f32((size_t)rows * Tp * ldc);
    tp.mid = new_img(T, Cw);
    conv_gemm(in, PAD - r.c1.ks / 2, 1, pk + r.c1.pk, r.co, r.c1.Kp, prm + r.c1.b, tp.u1, ldc);
    GnArgs a;
    memset(&a, 0, sizeof(a));
    a.src = tp.u1, a.lds = ldc, a.Tps = Tp, a.T = T, a.C = r.co, a.Cw = Cw, a.G = d.n_groups, a.gamma = prm + r.n1.g;
    a.beta = prm + r.n1.b, a.eps = d.groupnorm_eps, a.act = d.act, a.film = d.cond_predict_scale ? 2 : 1, a.emb = tp.emb;
    a.lde = rup(r.cc, 16), a.dst = tp.mid.p, a.ldd = Cw, a.zero_pads = 1;
    if (!dry) hipLaunchKernelGGL((unet_gn_kernel<P>), dim3((unsigned)rows), dim3(256), 0, s, a);
    conv_gemm(tp.mid, PAD - r.c2.ks / 2, 1, pk + r.c2.pk, r.co, r.c2.Kp, prm + r.c2.b, tp.u2, ldc);
    memset(&a, 0, sizeof(a));
    a.src = tp.u2, a.lds = ldc, a.Tps = Tp, a.T = T, a.C = r.co, a.Cw = Cw, a.G = d.n_groups, a.gamma = prm + r.n2.g;
    a.beta = prm + r.n2.b, a.eps = d.groupnorm_eps, a.act = d.act;
    if (r.has_res) {
      conv_gemm(in, PAD, 1, pk + r.res.pk, r.co, r.res.Kp, prm + r.res.b, tmpA, ldc);
      a.res = 1, a.resf = tmpA, a.ldr = ldc;
    } else {
      a.res = 2, a.resi = in.p, a.ldri = in.C;
    }
    a.dst = out, a.ldd = ldd, a.coff = coff, a.zero_pads = zero_pads, a.dst2 = dst2, a.ldd2 = ldd2, a.coff2 = coff2;
    a.zero_pads2 = zero2;
    if (!dry) hipLaunchKernelGGL((unet_gn_kernel<P>), dim3((unsigned)rows), dim3(256), 0, s, a);
    tape.push_back(tp);
  }
  Img cat[4];
  void forward() {
    const int nl = d.n_levels, T0 = d.horizon_steps;
    tape.clear();
    in_img = new_img(T0, 64);
    g = take((size_t)rows * L.Kg * P::ESIZE);
    graw = take((size_t)rows * L.Kg * P::ESIZE);
    eps = new_f32((size_t)rows * T0 * d.action_dim);
    if (!dry)
      hipLaunchKernelGGL((unet_train_input_kernel<P>), dim3((unsigned)rows), dim3(256), 0, s, io, T0, d.action_dim, d.cond_dim,
                         d.time_dim, (const float*)(pk + L.temb), (E*)in_img.p, 64, (E*)g, (E*)graw, L.Kg,
                         d.larger_encoder ? -1 : d.act);
    for (int l = 1; l < nl; ++l) cat[l] = new_img(T0 >> l, 2 * rup(L.dims[l + 1], 64));
    Img cur = in_img;
    for (int i = 0; i < nl; ++i) {
      const int C = L.dims[i + 1], Cw = rup(C, 64), T = T0 >> i;
      Img a = new_img(T, Cw);
      resblock_fwd(L.down[2 * i], cur, a.p, Cw, 0, 1);
      Img o = new_img(T, Cw);
      resblock_fwd(L.down[2 * i + 1], a, o.p, Cw, 0, 1, i >= 1 ? cat[i].p : nullptr, 2 * Cw, Cw, 1);
      cur = o;
      if (i < nl - 1) {
        const Conv& c = L.downs[i];
        down_in[i] = cur;
        conv_gemm(cur, PAD - 1, 2, pk + c.pk, c.co, c.Kp, prm + c.b, tmpA, rup(C, 16));
        Img dn = new_img(T / 2, Cw);
        if (!dry)
          hipLaunchKernelGGL((unet_scatter_kernel<P>), dim3((unsigned)rows), dim3(256), 0, s, tmpA, rup(C, 16), cur.Tp() / 2,
                             T / 2, 1, C, (E*)dn.p, Cw, 0, 1);
        cur = dn;
      }
    }
    for (int i = 0; i < 2; ++i) {
      const int Cw = rup(L.dims[nl], 64);
      const bool to_cat = i == 1 && nl >= 2;
      Img o = to_cat ? cat[nl - 1] : new_img(cur.T, Cw);
      resblock_fwd(L.mid[i], cur, o.p, to_cat ? 2 * Cw : Cw, 0, to_cat ? 0 : 1);
      cur = Img{o.p, cur.T, to_cat ? 2 * Cw : Cw};
    }
    for (int j = 0; j < nl - 1; ++j) {
      const int din = L.dims[nl - 1 - j], dinw = rup(din, 64);
      Img a = new_img(cur.T, dinw);
      resblock_fwd(L.up[2 * j], cur, a.p, dinw, 0, 1);
      Img b2 = new_img(cur.T, dinw);
      resblock_fwd(L.up[2 * j + 1], a, b2.p, dinw, 0, 1);
      const Conv& c = L.ups[j];
      up_in[j] = b2;
      conv_gemm(b2, PAD - 1, 1, pk + c.pk, 2 * din, c.Kp, nullptr, tmpA, rup(2 * din, 16));
      const int lvl = nl - 2 - j;
      const bool to_cat = lvl >= 1;
      Img up = to_cat ? cat[lvl] : new_img(b2.T * 2, dinw);
      const int ldd = to_cat ? 2 * dinw : dinw;
      if (!dry)
        hipLaunchKernelGGL((unet_up_scatter_kernel<P>), dim3((unsigned)rows), dim3(256), 0, s, tmpA, rup(2 * din, 16), b2.Tp(),
                           b2.T, din, prm + c.b, up.p, ldd, to_cat ? 0 : 1);
      cur = Img{up.p, b2.T * 2, ldd};
    }
    {
      const int C = d.dim, Cw = rup(C, 64), T = cur.T, ldc = rup(C, 16);
      fin_in = cur;
      fin_u = new_f32((size_t)rows * cur.Tp() * ldc);
      fin_mid = new_img(T, Cw);
      conv_gemm(cur, PAD - L.fin.ks / 2, 1, pk + L.fin.pk, C, L.fin.Kp, prm + L.fin.b, fin_u, ldc);
      GnArgs a;
      memset(&a, 0, sizeof(a));
      a.src = fin_u, a.lds = ldc, a.Tps = cur.Tp(), a.T = T, a.C = C, a.Cw = Cw, a.G = d.n_groups, a.gamma = prm + L.fin_n.g;
      a.beta = prm + L.fin_n.b, a.eps = d.groupnorm_eps, a.act = d.act, a.dst = fin_mid.p, a.ldd = Cw, a.zero_pads = 1;
      if (!dry) hipLaunchKernelGGL((unet_gn_kernel<P>), dim3((unsigned)rows), dim3(256), 0, s, a);
      const int ldo = 16 * ((d.action_dim + 15) / 16);
      conv_gemm(fin_mid, PAD, 1, pk + L.fin_out.pk, d.action_dim, L.fin_out.Kp, prm + L.fin_out.b, tmpA, ldo);
      const int64_t n = rows * T * d.action_dim;
      if (!dry)
        hipLaunchKernelGGL(unet_gather_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, tmpA, ldo, fin_mid.Tp(), T,
                           d.action_dim, eps, rows);
    }
  }

  // ------------------------------------------------------------------------------------------------ backward
  float* grad;
  // rows a weight-gradient GEMM may include: the window of its last row must end inside the image
  int64_t wrows(const Img& im, int r0, int stride, int win) const { return ((int64_t)rows * im.Tp() - r0 - win) / stride + 1; }
  void conv_wgrad(const Conv& c, const void* dUimg_or_rows, int lda, int N1, const Img& in, int r0, int stride, int win) {
    // A: gradient rows [.][N1] aligned with the forward GEMM's output rows; B: the forward's X operand
    wgrad(dUimg_or_rows, lda, N1, dry ? nullptr : imgrow(in, r0), stride * in.C, win * in.C, wrows(in, r0, stride, win));
  }
  void unpack_conv(const Conv& c, int ld) {
    if (dry) return;
    const size_t n = (size_t)c.co * c.ci * c.ks;
    hipLaunchKernelGGL(unet_unpack_conv_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dwp, ld, c.co, c.ci, c.ks,
                       c.cip, c.split, grad + c.w);
  }
  // Conv1dBlock backward: upstream gradient(s) of the block-half output -> dU image (returned in `dU`), parameter
  // gradients of the conv and the norm; FiLM gradients -> demb
  void convblock_bwd(const Conv& c, const Norm& nm, const Img& in, const float* u, int T, GradSrc* up, int n_up, int film,
                     const float* emb, int lde, float* demb, void* dU) {
    const int C = c.co, Cw = rup(C, 64), Tp = T + 2 * PAD, ldc = rup(C, 16);
    GnBwdArgs a;
    memset(&a, 0, sizeof(a));
    a.u = u, a.lds = ldc, a.Tps = Tp, a.T = T, a.C = C, a.Cw = Cw, a.G = d.n_groups, a.gamma = prm + nm.g, a.beta = prm + nm.b;
    a.eps = d.groupnorm_eps, a.act = d.act, a.n_up = n_up, a.up[0] = up[0];
    if (n_up > 1) a.up[1] = up[1];
    a.film = film, a.emb = emb, a.lde = lde, a.demb = demb, a.dU = dU, a.dgb = dgb;
    if (!dry) hipLaunchKernelGGL((unet_gn_bwd_kernel<P>), dim3((unsigned)rows), dim3(256), 0, s, a);
    colsum_f32(dgb, rows, 2 * C, 2 * C, grad + nm.g);  // d gamma | d beta are adjacent in the flat layout
    // conv parameters: dW = dU^T . Xwin (A = dU image shifted by PAD rows = the GEMM's output-row alignment), db = colsum
    Img dUi{dU, T, Cw};
    conv_wgrad(c, dry ? nullptr : imgrow(dUi, PAD), Cw, C, in, PAD - c.ks / 2, 1, c.ks);
    unpack_conv(c, c.ks * c.cip);
    colsum_img(dU, rows * Tp, C, Cw, grad + c.b);
  }
  void encoder_bwd(const RBTape& tp, float* demb) {
    // emb = L_last(...): gradients of the encoder's linears; d loss / d g accumulates into dgacc (time-embedding columns)
    const ResBlock& r = *tp.r;
    const int lde = rup(r.cc, 16);
    // demb f32 -> elem operand
    void* dcur = imgA;  // [rows][rup(cc,64)] elem
    const int ldq = rup(r.cc, 64);
    if (!dry) launch_cast_pad<P>(demb, (int)rows, r.cc, lde, dcur, ldq, s);
    for (int i = r.n_enc - 1; i >= 0; --i) {
      const Lin& l = r.enc[i];
      const void* xin = i == 0 ? g : tp.ea[i - 1];
      const int ldx = i == 0 ? L.Kg : rup(r.enc[i - 1].out, 64);
      wgrad(dcur, ldq, l.out, xin, ldx, l.in, rows);
      if (!dry) {
        const size_t n = (size_t)l.out * l.in;
        hipLaunchKernelGGL(unet_unpack_conv_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dwp, l.in, l.out, l.in,
                           1, l.in, 0, grad + l.w);
      }
      colsum_img(dcur, rows, l.out, ldq, grad + l.b);
      if (i > 0) {  // d (previous activation) = dcur . W, times act'(z_{i-1})
        if (!dry) {
          GemmNT g2;
          memset(&g2, 0, sizeof(g2));
          g2.X = dcur, g2.ldx = ldq, g2.M = (int)rows, g2.N = l.in, g2.Kp = rup(l.out, 64), g2.W = pk + l.pkT;
          g2.ldw = rup(l.out, 64), g2.dsrc = tp.ez[i - 1], g2.dsrc_kind = 2, g2.dsrc_ld = rup(r.enc[i - 1].out, 64);
          g2.dact = d.act, g2.out_pre = dcur == imgA ? imgB : imgA, g2.ldo = ldq;
          launch_gemm_nt<P>(g2, s);
          if (rup(l.in, 16) < ldq) launch_zero_cols<P>(g2.out_pre, (int)rows, rup(l.in, 16), ldq, ldq, s);
        }
        dcur = dcur == imgA ? imgB : imgA;
      } else {  // d g (time-embedding columns only; the state is data): accumulate over the blocks
        if (!dry) {
          GemmNT g2;
          memset(&g2, 0, sizeof(g2));
          // (with d_obs requested -- a visual encoder feeds the observation columns -- all of g's columns are kept)
          g2.X = dcur, g2.ldx = ldq, g2.M = (int)rows, g2.N = dobs_out ? d.time_dim + d.cond_dim : d.time_dim;
          g2.Kp = rup(l.out, 64), g2.W = pk + l.pkT;
          g2.ldw = rup(l.out, 64), g2.out_f32 = dgacc, g2.ldo32 = dg_ld();
          if (dg_started) g2.res = dgacc, g2.ldres = dg_ld();
          if (!d.larger_encoder)  // one-layer encoder = Linear(act(g)): the chain rule's act'(g) on the raw vector
            g2.dsrc = graw, g2.dsrc_kind = 2, g2.dsrc_ld = L.Kg, g2.dact = d.act;
          launch_gemm_nt<P>(g2, s);
        }
        dg_started = true;
      }
    }
  }
  bool dg_started;
  float* dobs_out = nullptr;  // [rows][cond] f32: d loss / d observation columns of g (optional output of backward())
  int dg_ld() const { return rup(d.time_dim + d.cond_dim, 16); }
  // ResidualBlock1D backward.  up / n_up: gradient of the block's output; returns the gradient of its input in `gin`
  // (f32 rows [rows * Tp][ci_ld]) unless `need_in` is false (the network input)
  void resblock_bwd(const RBTape& tp, GradSrc* up, int n_up, float* gin, bool need_in) {
    const ResBlock& r = *tp.r;
    const int T = tp.T, Tp = T + 2 * PAD, C = r.co, Cw = rup(C, 64);
    // second half: y = act(GN(u2)) + res
    void* dU2 = imgA;
    convblock_bwd(r.c2, r.n2, tp.mid, tp.u2, T, up, n_up, 0, nullptr, 0, nullptr, dU2);
    // d mid = conv2 data gradient (f32 rows, width C)
    Img dU2i{dU2, T, Cw};
    conv_gemm(dU2i, PAD - r.c2.ks / 2, 1, pk + r.c2.pkT, C, r.c2.ks * Cw, nullptr, tmpB, rup(C, 16));
    // first half: mid = FiLM(act(GN(u1)))
    GradSrc gm{tmpB, rup(C, 16), 0, Tp};
    float* demb = tmpC;
    void* dU1 = imgB;
    convblock_bwd(r.c1, r.n1, tp.in, tp.u1, T, &gm, 1, d.cond_predict_scale ? 2 : 1, tp.emb, rup(r.cc, 16), demb, dU1);
    // input gradient: conv1 data gradient + the skip path
    if (need_in) {
      Img dU1i{dU1, T, Cw};
      const int ldi = rup(r.ci, 16);
      if (r.has_res) {
        // dY image (sum of the upstream sources) feeds the 1x1 conv's data gradient, then conv1's adds to it
        if (!dry)
          hipLaunchKernelGGL((unet_rows_to_img_kernel<P>), dim3((unsigned)rows), dim3(256), 0, s, up[0], n_up > 1 ? up[1] : up[0],
                             n_up, T, C, (E*)imgA);
        Img dYi{imgA, T, Cw};
        conv_gemm(dYi, PAD, 1, pk + r.res.pkT, r.ci, Cw, nullptr, gin, ldi);
        conv_gemm(dU1i, PAD - r.c1.ks / 2, 1, pk + r.c1.pkT, r.ci, r.c1.ks * Cw, nullptr, gin, ldi, gin, ldi);
      } else {
        // identity skip: gin = conv1 data gradient + sum(up).  The GEMM's f32 addend takes one source; a second is added first
        if (n_up == 1 && up[0].coff == 0 && up[0].Tp == Tp) {
          conv_gemm(dU1i, PAD - r.c1.ks / 2, 1, pk + r.c1.pkT, r.ci, r.c1.ks * Cw, nullptr, gin, ldi, up[0].p, up[0].ld);
        } else {
          if (!dry)
            hipLaunchKernelGGL((unet_rows_to_img_kernel<P>), dim3((unsigned)rows), dim3(256), 0, s, up[0],
                               n_up > 1 ? up[1] : up[0], n_up, T, C, (E*)imgA);
          conv_gemm(dU1i, PAD - r.c1.ks / 2, 1, pk + r.c1.pkT, r.ci, r.c1.ks * Cw, nullptr, gin, ldi);
          if (!dry) add_img_rows(gin, ldi, imgA, T, C);
        }
      }
    }
    if (r.has_res) {  // parameters of the 1x1 residual conv: dW = dY^T . x, db = colsum(dY)
      if (!need_in && !dry)
        hipLaunchKernelGGL((unet_rows_to_img_kernel<P>), dim3((unsigned)rows), dim3(256), 0, s, up[0], n_up > 1 ? up[1] : up[0],
                           n_up, T, C, (E*)imgA);
      Img dYi{imgA, T, Cw};
      conv_wgrad(r.res, dry ? nullptr : imgrow(dYi, PAD), Cw, C, tp.in, PAD, 1, 1);
      unpack_conv(r.res, r.res.cip);
      colsum_img(imgA, rows * Tp, C, Cw, grad + r.res.b);
    }
    encoder_bwd(tp, demb);
  }
  // gin[(b, t)][c] += img[b][t + PAD][c]
  void add_img_rows(float* gin, int ld, const void* img, int T, int C);

  void backward(const void* d_eps, int ldde, float* grad_, float* dobs = nullptr) {
    grad = grad_;
    dobs_out = dobs;
    dg_started = false;
    const int nl = d.n_levels, T0 = d.horizon_steps, Tp0 = T0 + 2 * PAD;
    if (!dry) launch_zero_bytes(grad, (size_t)L.n_params * 4, s);
    // ---- final 1x1 conv
    Img dE = new_img(T0, 64);
    if (!dry)
      hipLaunchKernelGGL((unet_deps_img_kernel<P>), dim3((unsigned)rows), dim3(256), 0, s, (const E*)d_eps, ldde, T0,
                         d.action_dim, (E*)dE.p);
    {
      const Conv& c = L.fin_out;
      wgrad(dry ? nullptr : imgrow(dE, PAD), 64, d.action_dim, dry ? nullptr : imgrow(fin_mid, PAD), fin_mid.C, fin_mid.C,
            wrows(fin_mid, PAD, 1, 1));
      unpack_conv(c, c.cip);
      colsum_img(dE.p, rows * Tp0, d.action_dim, 64, grad + c.b);
    }
    // gradient buffers (f32 rows), two per level ping-pong + one for concat splits
    size_t gmax = 0;
    for (int l = 0; l < nl; ++l) {
      const size_t b = (size_t)((T0 >> l) + 2 * PAD) * rup(2 * rup(L.dims[l + 1], 64), 16);
      gmax = b > gmax ? b : gmax;
    }
    float* G[4];
    for (int i = 0; i < 4; ++i) G[i] = new_f32((size_t)rows * gmax);
    int gi = 0;
    auto nextG = [&]() { return G[(gi++) & 3]; };
    // d fin_mid = dE . Wout  (f32 rows, width dim)
    float* g_finmid = nextG();
    conv_gemm(dE, PAD, 1, pk + L.fin_out.pkT, d.dim, 64, nullptr, g_finmid, rup(d.dim, 16));
    // final Conv1dBlock
    GradSrc up0{g_finmid, rup(d.dim, 16), 0, Tp0};
    convblock_bwd(L.fin, L.fin_n, fin_in, fin_u, T0, &up0, 1, 0, nullptr, 0, nullptr, imgA);
    Img dUf{imgA, T0, rup(d.dim, 64)};
    float* gcur = nextG();
    int gcur_ld = rup(d.dim, 16);  // final_conv's input map has `dim` channels (real)
    conv_gemm(dUf, PAD - L.fin.ks / 2, 1, pk + L.fin.pkT, d.dim, L.fin.ks * rup(d.dim, 64), nullptr, gcur, gcur_ld);
    GradSrc cur{gcur, gcur_ld, 0, Tp0};  // gradient of the image feeding final_conv (width = its C)
    int ti = (int)tape.size() - 1;
    GradSrc skipg[4];
    memset(skipg, 0, sizeof(skipg));
    // ---- up path (reverse)
    for (int j = nl - 2; j >= 0; --j) {
      const int din = L.dims[nl - 1 - j], dinw = rup(din, 64);
      const int lvl = nl - 2 - j;  // the level the upsampled map landed on
      const Img& xin = up_in[j];   // Upsample1d's input (T small)
      const int Ts = xin.T, Tb = 2 * Ts;
      // `cur` is the gradient of the upsampled image: when it was written into a concat image its width is 2 din and the
      // upper half belongs to the skip (handled at the consumer below), the lower half is ours
      const Conv& c = L.ups[j];
      // parameters: forward form Y2[(b, m)][ph * C + co] = Xwin . W2^T
      void* dY2 = imgA;
      if (!dry)
        hipLaunchKernelGGL((unet_rows_fmt_kernel<P>), dim3((unsigned)rows), dim3(256), 0, s, cur, cur, 1, Tb, din, 2, xin.Tp(),
                           (E*)dY2);
      wgrad(dY2, 2 * din, 2 * din, dry ? nullptr : imgrow(xin, PAD - 1), dinw, 3 * dinw, wrows(xin, PAD - 1, 1, 3));
      if (!dry) {
        const size_t n = (size_t)din * din * 4;
        hipLaunchKernelGGL(unet_unpack_convT_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dwp, 3 * dinw, din, dinw,
                           grad + c.w);
      }
      // bias: both phases share it: column sums of the big-T gradient
      void* dYimg = imgB;
      if (!dry)
        hipLaunchKernelGGL((unet_rows_to_img_kernel<P>), dim3((unsigned)rows), dim3(256), 0, s, cur, cur, 1, Tb, din, (E*)dYimg);
      colsum_img(dYimg, rows * (Tb + 2 * PAD), din, dinw, grad + c.b);
      // data: strided conv over the dy image: rows r' = b * (Tpb / 2) + s
      Img dYi{dYimg, Tb, dinw};
      float* gx = nextG();
      conv_gemm(dYi, PAD - 1, 2, pk + c.pkT, din, 4 * dinw, nullptr, gx, rup(din, 16));
      GradSrc up1{gx, rup(din, 16), 0, dYi.Tp() / 2};
      // second block of the level
      float* g1 = nextG();
      resblock_bwd(tape[ti--], &up1, 1, g1, true);
      GradSrc up2{g1, rup(din, 16), 0, Ts + 2 * PAD};
      const int dout = L.dims[nl - j];
      // (dedicated, not from the rotating set: its upper half is the skip's gradient, consumed much later by the down path)
      float* g0 = new_f32((size_t)rows * (Ts + 2 * PAD) * rup(2 * dout, 16));
      resblock_bwd(tape[ti--], &up2, 1, g0, true);  // input: concat image, width 2 * dims[nl - j]
      cur = GradSrc{g0, rup(2 * dout, 16), 0, Ts + 2 * PAD};
      skipg[nl - 1 - j] = GradSrc{g0, rup(2 * dout, 16), dout, Ts + 2 * PAD};  // the skip's half (level nl-1-j)
      (void)lvl;
    }
    // ---- mid
    for (int i = 1; i >= 0; --i) {
      float* gq = nextG();
      resblock_bwd(tape[ti--], &cur, 1, gq, true);
      cur = GradSrc{gq, rup(L.dims[nl], 16), 0, (T0 >> (nl - 1)) + 2 * PAD};
    }
    // ---- down path (reverse)
    for (int i = nl - 1; i >= 0; --i) {
      const int C = L.dims[i + 1], T = T0 >> i;
      GradSrc ups[2];
      int n_up = 0;
      ups[n_up++] = cur;                       // from the next layer (mid, or this level's Downsample1d below)
      if (i >= 1 && nl >= 2) ups[n_up++] = skipg[i];  // from the concat consumer
      float* g1 = nextG();
      resblock_bwd(tape[ti--], ups, n_up, g1, true);
      GradSrc u2{g1, rup(C, 16), 0, T + 2 * PAD};
      float* g0 = nextG();
      const bool need_in = i > 0;
      resblock_bwd(tape[ti--], &u2, 1, g0, need_in);
      if (i > 0) {
        // Downsample1d of level i-1 produced this level's input: g0 is the gradient of its output (T small)
        const Conv& c = L.downs[i - 1];
        const Img& xin = down_in[i - 1];  // big-T image, C = dims[i]
        const int Cb = L.dims[i], Cbw = rup(Cb, 64), Tb = xin.T;
        GradSrc gy{g0, rup(Cb, 16), 0, T + 2 * PAD};
        // A operand in the forward GEMM's row format (b * Tpb / 2 + t'), and as an image for the data gradient
        void* dYr = imgA;
        if (!dry)
          hipLaunchKernelGGL((unet_rows_fmt_kernel<P>), dim3((unsigned)rows), dim3(256), 0, s, gy, gy, 1, T, Cb, 1, xin.Tp() / 2,
                             (E*)dYr);
        wgrad(dYr, Cb, Cb, dry ? nullptr : imgrow(xin, PAD - 1), 2 * Cbw, 3 * Cbw, wrows(xin, PAD - 1, 2, 3));
        unpack_conv(c, c.ks * c.cip);
        colsum_img(dYr, rows * (xin.Tp() / 2), Cb, Cb, grad + c.b);
        void* dYimg = imgB;
        if (!dry)
          hipLaunchKernelGGL((unet_rows_to_img_kernel<P>), dim3((unsigned)rows), dim3(256), 0, s, gy, gy, 1, T, Cb, (E*)dYimg);
        Img dYi{dYimg, T, Cbw};
        conv_gemm(dYi, PAD, 1, pk + c.pkT, 2 * Cb, 2 * Cbw, nullptr, tmpB, rup(2 * Cb, 16));
        float* gb = nextG();
        if (!dry)
          hipLaunchKernelGGL(unet_phase_rows_kernel, dim3((unsigned)rows), dim3(256), 0, s, tmpB, rup(2 * Cb, 16), dYi.Tp(), T, Cb,
                             gb, Tb + 2 * PAD);
        cur = GradSrc{gb, Cb, 0, Tb + 2 * PAD};
      }
    }
    // ---- time embedding: G[k][td] = segmented sum of d g over the rows of step k, then the time MLP's backward
    if (!dry) {
      float* Gk = part;  // [Kft][td]
      hipLaunchKernelGGL(unet_temb_segsum_kernel, dim3(io.Kft), dim3(256), 0, s, dgacc, dg_ld(), io.krow, rows,
                         d.time_dim, Gk);
      if (dobs_out) {
        const size_t n = (size_t)rows * d.cond_dim;
        hipLaunchKernelGGL(unet_copy_cols_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dgacc, dg_ld(), d.time_dim,
                           d.cond_dim, rows, dobs_out);
      }
      const size_t lds = (size_t)io.Kft * (d.time_dim + 8 * d.time_dim) * 4;
      hipLaunchKernelGGL(unet_time_bwd_kernel, dim3(1), dim3(256), lds, s, prm + L.t1.w, prm + L.t1.b, prm + L.t2.w, Gk, io.ksteps,
                         io.Kft, d.time_dim, grad + L.t1.w, grad + L.t1.b, grad + L.t2.w, grad + L.t2.b);
    }
  }

  void plan_scratch() {
    const int nl = d.n_levels, T0 = d.horizon_steps;
    int cmax = 64;
    for (int i = 1; i <= nl; ++i) cmax = rup(L.dims[i], 64) > cmax ? rup(L.dims[i], 64) : cmax;
    tmp_floats = (size_t)rows * (T0 + 2 * PAD) * 2 * cmax;
    tmpA = new_f32(tmp_floats), tmpB = new_f32(tmp_floats), tmpC = new_f32((size_t)rows * 2 * cmax);
    imgA = take(tmp_floats * P::ESIZE), imgB = take(tmp_floats * P::ESIZE);
    size_t wmax = (size_t)2 * cmax * 5 * 2 * cmax;  // largest packed weight: first up block's conv1 (2C -> C, 5 taps) etc.
    dwp = new_f32(wmax);
    slab_floats = wmax * 8;
    slab = new_f32(slab_floats);
    part = new_f32((size_t)64 * 4 * cmax + 4096);
    dgb = new_f32((size_t)rows * 2 * cmax);
    dgacc = new_f32((size_t)rows * dg_ld());
  }
};

template <class P>
__global__ __launch_bounds__(256) void unet_add_img_rows_kernel(float* gin, int ld, const typename P::elem_t* img, int T, int C) {
  const int64_t b = blockIdx.x;
  const int Tp = T + 2 * PAD, Cw = (C + 63) / 64 * 64;
  for (int i = threadIdx.x; i < T * C; i += 256) {
    const int t = i / C, c = i % C;
    gin[((size_t)b * Tp + t) * ld + c] += P::to_f32(img[((size_t)b * Tp + t + PAD) * Cw + c]);
  }
}
template <class P>
void UnetTrainer<P>::add_img_rows(float* gin, int ld, const void* img, int T, int C) {
  hipLaunchKernelGGL((unet_add_img_rows_kernel<P>), dim3((unsigned)rows), dim3(256), 0, s, gin, ld, (const E*)img, T, C);
}

template <class P>
static void trainer_init(UnetTrainer<P>& t, const dppo_unet_desc& d, int64_t N) {
  t.d = d;
  t.L = make_layout(d, P::ESIZE, 0);
  t.rows = N, t.off = 0, t.grad = nullptr;
  memset(&t.io, 0, sizeof(t.io));
  t.io.Kft = 1;
}
template <class P>
size_t unet_trainer_bytes(const dppo_unet_desc& d, int64_t N) {
  UnetTrainer<P> t;
  trainer_init(t, d, N);
  t.dry = true, t.base = nullptr, t.s = nullptr, t.prm = nullptr, t.pk = nullptr;
  t.plan_scratch();
  t.forward();
  t.backward(nullptr, 0, nullptr);
  return al(t.off);
}
template <class P>
UnetTrainer<P>* unet_trainer_new(const dppo_unet_desc& d, const float* prm, const char* pk, int64_t N, void* ws, size_t wsb,
                                 hipStream_t s) {
  UnetTrainer<P>* t = new UnetTrainer<P>();
  trainer_init(*t, d, N);
  t->dry = false, t->base = (char*)ws, t->cap = wsb, t->s = s, t->prm = prm, t->pk = pk;
  t->plan_scratch();
  return t;
}
template <class P>
float* unet_trainer_forward(UnetTrainer<P>* t, const UnetTrainIO& io) {
  t->io = io;
  t->forward();
  return t->eps;
}
template <class P>
void unet_trainer_backward(UnetTrainer<P>* t, const void* d_eps, int ldde, float* grad, float* d_obs) {
  t->backward(d_eps, ldde, grad, d_obs);
}
template <class P>
void unet_trainer_free(UnetTrainer<P>* t) {
  delete t;
}
int unet_check_desc(const dppo_unet_desc* d) { return check_desc(d); }
// the sampler's two elementwise kernels, for any denoiser whose K-step loop runs on the host (plain MLP trunks: api.hip)
void launch_chain_init(const float* noise, uint32_t k0, uint32_t k1, int64_t n, int AF, float* x, float* chains, int chain_len,
                       int init_slot, hipStream_t s) {
  hipLaunchKernelGGL(unet_init_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, noise, k0, k1, n, AF, x, chains,
                     chain_len, init_slot);
}
void launch_chain_step(const dppo_diffusion_cfg& cfg, const dppo_step& st, float* x, const float* eps, int lde, const float* noise,
                       size_t nz0, int64_t n, int AF, int chain_len, int last, float* chains, float* traj, hipStream_t s) {
  StepArgs a;
  memset(&a, 0, sizeof(a));
  a.cfg = cfg, a.st = st, a.x = x, a.eps = eps, a.lde = lde, a.noise = noise, a.nz0 = nz0, a.n = n, a.AF = AF;
  a.chain_len = chain_len, a.last = last, a.chains = chains, a.traj = traj;
  hipLaunchKernelGGL(unet_step_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
}
__global__ void unet_index_kernel(const int64_t* inds, const int64_t* kinds, int Kft, int64_t N, int32_t* brow, int32_t* krow) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  if (kinds != nullptr) {
    brow[n] = (int32_t)n, krow[n] = (int32_t)kinds[n];
  } else {
    const int64_t ind = inds ? inds[n] : n;
    brow[n] = (int32_t)(ind / Kft), krow[n] = (int32_t)(ind % Kft);
  }
}
void launch_unet_index(const int64_t* inds, const int64_t* kinds, int Kft, int64_t N, int32_t* brow, int32_t* krow,
                       hipStream_t s) {
  hipLaunchKernelGGL(unet_index_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, inds, kinds, Kft, N, brow, krow);
}
#define UNET_INST(P)                                                                                                    \
  template size_t unet_trainer_bytes<P>(const dppo_unet_desc&, int64_t);                                                \
  template UnetTrainer<P>* unet_trainer_new<P>(const dppo_unet_desc&, const float*, const char*, int64_t, void*, size_t, \
                                               hipStream_t);                                                            \
  template float* unet_trainer_forward<P>(UnetTrainer<P>*, const UnetTrainIO&);                                         \
  template void unet_trainer_backward<P>(UnetTrainer<P>*, const void*, int, float*, float*);                            \
  template void unet_trainer_free<P>(UnetTrainer<P>*);
UNET_INST(F32)
UNET_INST(BF16)

namespace {

template <class P>
int forward_impl(const dppo_unet_desc& d, const float* prm, const char* pk, const float* x, const int64_t* t,
                 const float* state, int64_t rows, float* eps, void* ws, int64_t wsb, hipStream_t s) {
  const Layout L = make_layout(d, P::ESIZE, 0);
  Ws<P> W;
  carve<P>(d, L, rows, (char*)ws, W);
  if ((int64_t)W.bytes > wsb) return api_fail(-1, "unet: workspace too small");
  hipLaunchKernelGGL((unet_input_kernel<P>), dim3((unsigned)rows), dim3(256), 0, s, x, d.horizon_steps, d.action_dim,
                     (typename P::elem_t*)W.in_img, 64);
  hipLaunchKernelGGL((unet_cond_rows_kernel<P>), dim3((unsigned)rows), dim3(64), 0, s, (const float*)(pk + L.temb), t, state,
                     d.time_dim, d.cond_dim, 1, (typename P::elem_t*)W.g, L.Kg, d.larger_encoder ? -1 : d.act);
  Runner<P> R{d, L, prm, pk, W, rows, s};
  R.forward(eps);
  return api_check_launch();
}

// the ResidualBlock1Ds in the order Runner::forward runs them
int forward_order(const Layout& L, int nl, const ResBlock** out) {
  int n = 0;
  for (int i = 0; i < nl; ++i) out[n++] = &L.down[2 * i], out[n++] = &L.down[2 * i + 1];
  out[n++] = &L.mid[0], out[n++] = &L.mid[1];
  for (int j = 0; j < nl - 1; ++j) out[n++] = &L.up[2 * j], out[n++] = &L.up[2 * j + 1];
  return n;
}
template <class P>
struct FilmPlan {
  void *g, *e1, *e2;  // [n_steps * B][Kg] conditioning rows; encoder hidden activations
  float* film[64];    // per block: [n_steps * B][rup(cc, 16)]
  size_t bytes;
};
// carve the FiLM tables of a sampling call out of `avail` bytes at `base`; false (and nothing used) if they do not fit
template <class P>
bool plan_film(const dppo_unet_desc& d, const Layout& L, int64_t B, int n_steps, char* base, int64_t avail, FilmPlan<P>& F) {
  const ResBlock* order[64];
  const int nb = forward_order(L, d.n_levels, order);
  const size_t R = (size_t)B * n_steps;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    off = al(off);
    void* p = base ? base + off : nullptr;
    off += bytes + 65536;
    return p;
  };
  int ccmax = 64;
  for (int b = 0; b < nb; ++b) ccmax = rup(order[b]->cc, 64) > ccmax ? rup(order[b]->cc, 64) : ccmax;
  F.g = take(R * L.Kg * P::ESIZE);
  F.e1 = take(R * ccmax * P::ESIZE);
  F.e2 = take(R * ccmax * P::ESIZE);
  for (int b = 0; b < nb; ++b) F.film[b] = (float*)take(R * rup(order[b]->cc, 16) * 4);
  F.bytes = al(off);
  return base != nullptr && avail >= (int64_t)F.bytes;
}

template <class P>
int sample_impl(const dppo_unet_desc& d, const float* pb, const char* kb, const float* pf, const char* kf,
                const dppo_diffusion_cfg& cfg, const dppo_step* sched, int n_steps, const float* obs, const float* noise,
                int64_t B, float* traj, float* chains, int chain_len, int init_slot, void* ws, int64_t wsb, hipStream_t s) {
  const Layout L = make_layout(d, P::ESIZE, 0);
  Ws<P> W;
  carve<P>(d, L, B, (char*)ws, W);
  if ((int64_t)W.bytes > wsb) return api_fail(-1, "unet: workspace too small");
  const int AF = d.horizon_steps * d.action_dim;
  const int64_t n = B * AF;
  const unsigned blocks = (unsigned)((n + 255) / 256);
  // FiLM tables for all steps, when the caller's workspace has the room (dppo_unet_sample_workspace_bytes): the encoders of
  // every block see (t_k, obs) only, so their 3 GEMMs x 8-12 blocks leave the per-step chain (~a third of its launches)
  FilmPlan<P> F;
  const bool pre = plan_film<P>(d, L, B, n_steps, (char*)ws + W.bytes, wsb - (int64_t)W.bytes, F);
  if (pre) {
    for (int i = 0; i < n_steps; ++i)
      hipLaunchKernelGGL((unet_cond_rows_kernel<P>), dim3((unsigned)B), dim3(64), 0, s,
                         (const float*)((sched[i].net ? kf : kb) + L.temb) + (size_t)sched[i].t * d.time_dim,
                         (const int64_t*)nullptr, obs, d.time_dim, d.cond_dim, 1,
                         (typename P::elem_t*)F.g + (size_t)i * B * L.Kg, L.Kg, d.larger_encoder ? -1 : d.act);
    // runs of consecutive steps on the same network share one encoder pass
    for (int a0 = 0; a0 < n_steps;) {
      int a1 = a0 + 1;
      while (a1 < n_steps && sched[a1].net == sched[a0].net) ++a1;
      Runner<P> R{d, L, sched[a0].net ? pf : pb, sched[a0].net ? kf : kb, W, B, s};
      const ResBlock* order[64];
      const int nbk = forward_order(L, d.n_levels, order);
      for (int b = 0; b < nbk; ++b)
        R.encoder(*order[b], (const char*)F.g + (size_t)a0 * B * L.Kg * P::ESIZE, (int64_t)(a1 - a0) * B,
                  F.film[b] + (size_t)a0 * B * rup(order[b]->cc, 16), F.e1, F.e2);
      a0 = a1;
    }
  }
  hipLaunchKernelGGL(unet_init_kernel, dim3(blocks), dim3(256), 0, s, noise, cfg.seed_lo, cfg.seed_hi, n, AF, W.x, chains,
                     chain_len, init_slot);
  for (int i = 0; i < n_steps; ++i) {
    const dppo_step& st = sched[i];
    const float* prm = st.net ? pf : pb;
    const char* pk = st.net ? kf : kb;
    hipLaunchKernelGGL((unet_input_kernel<P>), dim3((unsigned)B), dim3(256), 0, s, W.x, d.horizon_steps, d.action_dim,
                       (typename P::elem_t*)W.in_img, 64);
    if (!pre)
      hipLaunchKernelGGL((unet_cond_rows_kernel<P>), dim3((unsigned)B), dim3(64), 0, s,
                         (const float*)(pk + L.temb) + (size_t)st.t * d.time_dim, (const int64_t*)nullptr, obs, d.time_dim,
                         d.cond_dim, 1, (typename P::elem_t*)W.g, L.Kg, d.larger_encoder ? -1 : d.act);
    Runner<P> R{d, L, prm, pk, W, B, s};
    if (pre) R.film = F.film, R.film_row0 = (int64_t)i * B;
    R.forward(W.eps);
    StepArgs a;
    memset(&a, 0, sizeof(a));
    a.cfg = cfg, a.st = st, a.x = W.x, a.eps = W.eps, a.noise = noise, a.nz0 = (size_t)(i + 1) * n, a.n = n, a.AF = AF;
    a.chain_len = chain_len, a.last = i + 1 == n_steps, a.chains = chains, a.traj = traj;
    hipLaunchKernelGGL(unet_step_kernel, dim3(blocks), dim3(256), 0, s, a);
  }
  return api_check_launch();
}

template <class P>
int logprob_impl(const dppo_unet_desc& d, const float* prm, const char* pk, const dppo_diffusion_cfg& cfg,
                 const dppo_step* ksteps, int Kft, const float* obs, const float* chains, int64_t B, float* logp, void* ws,
                 int64_t wsb, hipStream_t s) {
  const Layout L = make_layout(d, P::ESIZE, 0);
  const int64_t rows = B * Kft;
  Ws<P> W;
  carve<P>(d, L, rows, (char*)ws, W);
  if ((int64_t)W.bytes > wsb) return api_fail(-1, "unet: workspace too small");
  hipLaunchKernelGGL((unet_chain_input_kernel<P>), dim3((unsigned)rows), dim3(256), 0, s, chains, ksteps, Kft,
                     d.horizon_steps, d.action_dim, (typename P::elem_t*)W.in_img, 64, W.tdev);
  hipLaunchKernelGGL((unet_cond_rows_kernel<P>), dim3((unsigned)rows), dim3(64), 0, s, (const float*)(pk + L.temb), W.tdev,
                     obs, d.time_dim, d.cond_dim, Kft, (typename P::elem_t*)W.g, L.Kg, d.larger_encoder ? -1 : d.act);
  Runner<P> R{d, L, prm, pk, W, rows, s};
  R.forward(W.eps);
  LogprobArgs la;
  memset(&la, 0, sizeof(la));
  la.eps = W.eps, la.lde = d.horizon_steps * d.action_dim, la.chains = chains, la.ksteps = ksteps, la.cfg = cfg;
  la.Kft = Kft, la.AF = d.horizon_steps * d.action_dim, la.M = rows, la.logp = logp;
  launch_logprob(la, s);
  return api_check_launch();
}

}  // namespace
}  // namespace dppo

using namespace dppo;
#define UNET_DISPATCH(prec, CALL) ((prec) == DPPO_PREC_F32 ? CALL(F32) : CALL(BF16))
static int check_prec_u(int prec) {
  if (prec != DPPO_PREC_F32 && prec != DPPO_PREC_BF16) return api_fail(-1, "prec must be DPPO_PREC_F32 or DPPO_PREC_BF16");
  return 0;
}

int64_t dppo_unet_param_count(const dppo_unet_desc* net) {
  if (check_desc(net)) return -1;
  return make_layout(*net, 4, 0).n_params;
}
int64_t dppo_unet_packed_bytes(const dppo_unet_desc* net, int prec, int n_time) {
  if (check_desc(net) || check_prec_u(prec)) return -1;
  if (n_time < 0 || n_time > 100000) return api_fail(-1, "n_time out of range");
  return (int64_t)make_layout(*net, prec == DPPO_PREC_F32 ? 4 : 2, n_time).pk_bytes;
}
int dppo_unet_pack(const dppo_unet_desc* net, int prec, int n_time, const float* params, void* packed,
                   dppo_stream_t stream) {
  if (int e = check_desc(net)) return e;
  if (int e = check_prec_u(prec)) return e;
  if (!params || !packed) return api_fail(-1, "null pointer");
#define CALL(P) pack_impl<P>(*net, n_time, params, (char*)packed, (hipStream_t)stream)
  return UNET_DISPATCH(prec, CALL);
#undef CALL
}
int64_t dppo_unet_workspace_bytes(const dppo_unet_desc* net, int prec, int64_t rows) {
  if (check_desc(net) || check_prec_u(prec)) return -1;
  if (rows < 1 || rows > (1 << 24)) return api_fail(-1, "rows out of range");
  if (prec == DPPO_PREC_F32) {
    const Layout L = make_layout(*net, 4, 0);
    Ws<F32> W;
    carve<F32>(*net, L, rows, nullptr, W);
    return (int64_t)W.bytes;
  }
  const Layout L = make_layout(*net, 2, 0);
  Ws<BF16> W;
  carve<BF16>(*net, L, rows, nullptr, W);
  return (int64_t)W.bytes;
}
int64_t dppo_unet_sample_workspace_bytes(const dppo_unet_desc* net, int prec, int64_t B, int n_steps) {
  const int64_t base = dppo_unet_workspace_bytes(net, prec, B);
  if (base < 0) return base;
  if (n_steps < 1 || n_steps > 100000) return api_fail(-1, "n_steps out of range");
  const bool f32 = prec == DPPO_PREC_F32;
  const Layout L = make_layout(*net, f32 ? 4 : 2, 0);
  if (f32) {
    FilmPlan<F32> F;
    plan_film<F32>(*net, L, B, n_steps, nullptr, 0, F);
    return base + (int64_t)F.bytes;
  }
  FilmPlan<BF16> F;
  plan_film<BF16>(*net, L, B, n_steps, nullptr, 0, F);
  return base + (int64_t)F.bytes;
}
int dppo_unet_forward(const dppo_unet_desc* net, int prec, const float* params, const void* packed, const float* x,
                      const int64_t* t, const float* state, int64_t rows, float* eps, void* workspace,
                      int64_t workspace_bytes, dppo_stream_t stream) {
  if (int e = check_desc(net)) return e;
  if (int e = check_prec_u(prec)) return e;
  if (!params || !packed || !x || !t || !state || !eps || !workspace) return api_fail(-1, "null pointer");
  if (rows < 1 || rows > (1 << 24)) return api_fail(-1, "rows out of range");
#define CALL(P) \
  forward_impl<P>(*net, params, (const char*)packed, x, t, state, rows, eps, workspace, workspace_bytes, (hipStream_t)stream)
  return UNET_DISPATCH(prec, CALL);
#undef CALL
}
int dppo_unet_sample_chain(const dppo_unet_desc* net, int prec, const float* params_base, const void* packed_base,
                           const float* params_ft, const void* packed_ft, const dppo_diffusion_cfg* cfg,
                           const dppo_step* sched_host, int n_steps, const float* obs, const float* noise, int64_t B,
                           float* traj, float* chains, int chain_len, int init_slot, void* workspace,
                           int64_t workspace_bytes, dppo_stream_t stream) {
  if (int e = check_desc(net)) return e;
  if (int e = check_prec_u(prec)) return e;
  if (!params_base || !packed_base || !params_ft || !packed_ft || !cfg || !sched_host || !obs || !traj || !workspace)
    return api_fail(-1, "null pointer");
  if (B < 1 || B > (1 << 24) || n_steps < 1) return api_fail(-1, "B / n_steps out of range");
  if (chains != nullptr && chain_len < 1) return api_fail(-1, "chain_len must be >= 1 when chains are requested");
#define CALL(P)                                                                                                     \
  sample_impl<P>(*net, params_base, (const char*)packed_base, params_ft, (const char*)packed_ft, *cfg, sched_host, n_steps, \
                 obs, noise, B, traj, chains, chain_len, init_slot, workspace, workspace_bytes, (hipStream_t)stream)
  return UNET_DISPATCH(prec, CALL);
#undef CALL
}
int dppo_unet_chain_logprob(const dppo_unet_desc* net, int prec, const float* params, const void* packed,
                            const dppo_diffusion_cfg* cfg, const dppo_step* ksteps, const dppo_step* ksteps_host, int Kft,
                            const float* obs, const float* chains, int64_t B, float* logp, void* workspace,
                            int64_t workspace_bytes, dppo_stream_t stream) {
  if (int e = check_desc(net)) return e;
  if (int e = check_prec_u(prec)) return e;
  if (!params || !packed || !cfg || !ksteps || !obs || !chains || !logp || !workspace) return api_fail(-1, "null pointer");
  if (B < 1 || Kft < 1 || B * Kft > (1 << 24)) return api_fail(-1, "B * Kft out of range");
  (void)ksteps_host;
#define CALL(P)                                                                                                      \
  logprob_impl<P>(*net, params, (const char*)packed, *cfg, ksteps, Kft, obs, chains, B, logp, workspace, workspace_bytes, \
                  (hipStream_t)stream)
  return UNET_DISPATCH(prec, CALL);
#undef CALL
}
