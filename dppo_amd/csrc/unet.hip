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
  int in, out, Kp;
};
struct Conv {
  int64_t w, b;
  size_t pk;
  int ci, co, ks, cip, Kp;  // cip: padded input channels (ld of the image it reads)
  bool transposed;
};
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
    return l;
  };
  auto conv = [&](int ci, int co, int ks, bool tr = false) {
    Conv c;
    c.ci = ci, c.co = co, c.ks = ks, c.cip = rup(ci, 64), c.transposed = tr;
    c.w = o, o += (int64_t)ci * co * ks;
    c.b = o, o += co;
    if (!tr) {
      c.Kp = ks * c.cip;
      c.pk = pk, pk = al(pk + (size_t)co * c.Kp * es);
    } else {  // ConvTranspose1d(C, C, 4, 2, 1) as one GEMM: N = 2 co (even | odd phase), K = 3 ci (window m-1, m, m+1)
      c.Kp = 3 * c.cip;
      c.pk = pk, pk = al(pk + (size_t)2 * co * c.Kp * es);
    }
    return c;
  };
  auto norm = [&](int c) {
    Norm n;
    n.g = o, o += c;
    n.b = o, o += c;
    return n;
  };
  auto resblock = [&](int ci, int co) {
    ResBlock r;
    memset(&r, 0, sizeof(r));
    r.ci = ci, r.co = co, r.cc = d.cond_predict_scale ? 2 * co : co;
    r.c1 = conv(ci, co, d.kernel_size), r.n1 = norm(co);
    r.c2 = conv(co, co, d.kernel_size), r.n2 = norm(co);
    if (d.larger_encoder) {
      r.n_enc = 3;
      r.enc[0] = lin(cbd, r.cc), r.enc[1] = lin(r.cc, r.cc), r.enc[2] = lin(r.cc, r.cc);
    } else {
      r.n_enc = 1;
      r.enc[0] = lin(cbd, r.cc);
    }
    r.has_res = ci != co;
    if (r.has_res) r.res = conv(ci, co, 1);
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
    if (i < nl - 1) L.downs.push_back(conv(L.dims[i + 1], L.dims[i + 1], 3));
  }
  for (int j = 0; j < nl - 1; ++j) {  // (dim_in, dim_out) = reversed(in_out[1:])[j] = (dims[nl-1-j], dims[nl-j])
    const int din = L.dims[nl - 1 - j], dout = L.dims[nl - j];
    L.up.push_back(resblock(2 * dout, din));
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
  if (d->dim < 64 || d->dim % 64) return api_fail(-1, "unet: dim must be a positive multiple of 64");
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
__global__ void pack_conv_kernel(const float* w, int co, int ci, int ks, int cip, typename P::elem_t* dst) {
  // dst[o][k * cip + c] = w[o][c][k]
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t Kp = (size_t)ks * cip;
  if (i >= (size_t)co * Kp) return;
  const int o = (int)(i / Kp), r = (int)(i % Kp), k = r / cip, c = r % cip;
  dst[i] = P::from_f32(c < ci ? w[((size_t)o * ci + c) * ks + k] : 0.f);
}
template <class P>
__global__ void pack_convT_kernel(const float* w, int ch, typename P::elem_t* dst) {
  // w[ci][co][4] (ConvTranspose1d, stride 2, padding 1): out[2m] = x[m] w[.,.,1] + x[m-1] w[.,.,3];
  // out[2m+1] = x[m+1] w[.,.,0] + x[m] w[.,.,2].  Window slots (m-1, m, m+1) -> dst[phase * ch + co][slot * ch + ci]
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t Kp = (size_t)3 * ch;
  if (i >= (size_t)2 * ch * Kp) return;
  const int row = (int)(i / Kp), r = (int)(i % Kp), slot = r / ch, ci = r % ch;
  const int phase = row / ch, co = row % ch;
  int k = -1;
  if (phase == 0) k = slot == 1 ? 1 : (slot == 0 ? 3 : -1);
  else k = slot == 2 ? 0 : (slot == 1 ? 2 : -1);
  dst[i] = P::from_f32(k >= 0 ? w[((size_t)ci * ch + co) * 4 + k] : 0.f);
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
  auto pl = [&](const Lin& l) { launch_cast_pad<P>(prm + l.w, l.out, l.in, l.in, pk + l.pk, l.Kp, s); };
  auto pc = [&](const Conv& c) {
    if (!c.transposed) {
      const size_t n = (size_t)c.co * c.Kp;
      hipLaunchKernelGGL((pack_conv_kernel<P>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, prm + c.w, c.co, c.ci,
                         c.ks, c.cip, (E*)(pk + c.pk));
    } else {
      const size_t n = (size_t)2 * c.co * c.Kp;
      hipLaunchKernelGGL((pack_convT_kernel<P>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, prm + c.w, c.co,
                         (E*)(pk + c.pk));
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
  for (int i = threadIdx.x; i < a.T * a.C; i += 256) {
    const int t = i / a.C, c = i % a.C, g = c / cg;
    float v = (src[(size_t)t * a.lds + c] - mean[g]) * rstd[g] * a.gamma[c] + a.beta[c];
    v = act_f(a.act, v);
    if (a.film == 1)
      v = v + a.emb[(size_t)b * a.lde + c];
    else if (a.film == 2)
      v = a.emb[(size_t)b * a.lde + c] * v + a.emb[(size_t)b * a.lde + a.C + c];
    if (a.res == 1)
      v = v + a.resf[((size_t)b * a.Tps + t) * a.ldr + c];
    else if (a.res == 2)
      v = v + P::to_f32(((const E*)a.resi)[((size_t)b * Tp + t + PAD) * a.ldri + c]);
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
  const int Tout = Tsrc * nsub, Tp = Tout + 2 * PAD;
  typename P::elem_t* d = dst + (size_t)b * Tp * ldd;
  for (int i = threadIdx.x; i < Tout * C; i += 256) {
    const int t = i / C, c = i % C, m = t / nsub, sub = t % nsub;
    d[(size_t)(t + PAD) * ldd + coff + c] = P::from_f32(src[((size_t)b * Tps + m) * lds + sub * C + c]);
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
  const int Tout = 2 * Tsrc, Tp = Tout + 2 * PAD;
  E* dd = (E*)dstv + (size_t)b * Tp * ldd;
  for (int i = threadIdx.x; i < Tout * C; i += 256) {
    const int t = i / C, c = i % C, m = t / 2, sub = t % 2;
    dd[(size_t)(t + PAD) * ldd + c] = P::from_f32(src[((size_t)b * Tps + m) * lds + sub * C + c] + bias[c]);
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
};
__global__ void unet_step_kernel(const StepArgs a) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  const dppo_step& st = a.st;
  const float x = a.x[i];
  float eps = a.eps[i], x0, mu;
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
  for (int i = 1; i <= nl; ++i) cmax = L.dims[i] > cmax ? L.dims[i] : cmax;
  ccmax = 2 * cmax;
  size_t img_max = 0;
  for (int l = 0; l < nl; ++l) {
    const size_t b = (size_t)((T0 >> l) + 2 * PAD) * (size_t)(2 * L.dims[l + 1]);
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
    W.cat[l] = l < nl ? take((size_t)rows * ((T0 >> l) + 2 * PAD) * 2 * L.dims[l + 1] * ES) : nullptr;
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
  void encoder(const ResBlock& r) {  // FiLM parameters of one block from the conditioning vector (unet.py:76-90,102)
    GemmNT g;
    const void* x = W.g;
    int K = L.Kg;
    for (int i = 0; i < r.n_enc; ++i) {
      const Lin& l = r.enc[i];
      memset(&g, 0, sizeof(g));
      g.X = x, g.ldx = K, g.M = (int)rows, g.N = l.out, g.Kp = l.Kp, g.W = pk + l.pk, g.ldw = l.Kp, g.bias = prm + l.b;
      if (i + 1 < r.n_enc) {
        g.out_act = i == 0 ? W.e1 : W.e2, g.ldo = rup(l.out, 64), g.act = d.act;
        x = g.out_act, K = g.ldo;
      } else {
        g.out_f32 = W.emb, g.ldo32 = rup(r.cc, 16);
      }
      launch_gemm_nt<P>(g, s);
    }
  }
  // ResidualBlock1D.forward (unet.py:100-118): in -> out image (channel offset coff of an image of width ldd), optionally
  // also into dst2 (the skip's half of a concat image)
  void resblock(const ResBlock& r, const Img& in, void* mid_img, void* out_img, int ldd, int coff, int zero_pads,
                void* dst2 = nullptr, int ldd2 = 0, int coff2 = 0, int zero2 = 0) {
    const int T = in.T, Tp = in.Tp(), ldc = rup(r.co, 16);
    encoder(r);
    conv(r.c1, in, W.conv, ldc);
    GnArgs a;
    memset(&a, 0, sizeof(a));
    a.src = W.conv, a.lds = ldc, a.Tps = Tp, a.T = T, a.C = r.co, a.G = d.n_groups, a.gamma = prm + r.n1.g;
    a.beta = prm + r.n1.b, a.eps = d.groupnorm_eps, a.act = d.act, a.film = d.cond_predict_scale ? 2 : 1, a.emb = W.emb;
    a.lde = rup(r.cc, 16), a.dst = mid_img, a.ldd = r.co, a.coff = 0, a.zero_pads = 1;
    hipLaunchKernelGGL((unet_gn_kernel<P>), dim3((unsigned)rows), dim3(256), 0, s, a);
    Img mid{mid_img, T, r.co};
    conv(r.c2, mid, W.conv, ldc);
    memset(&a, 0, sizeof(a));
    a.src = W.conv, a.lds = ldc, a.Tps = Tp, a.T = T, a.C = r.co, a.G = d.n_groups, a.gamma = prm + r.n2.g;
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
  void forward(float* out) {
    const int nl = d.n_levels, T0 = d.horizon_steps;
    Img cur{W.in_img, T0, 64};
    void* bufs[3] = {W.bufA, W.bufB, W.bufC};
    auto pick = [&](const void* x0, const void* x1) {  // a scratch image that is neither of the two in use
      for (void* b : bufs)
        if (b != x0 && b != x1) return b;
      return (void*)nullptr;
    };
    for (int i = 0; i < nl; ++i) {
      const int C = L.dims[i + 1], T = T0 >> i;
      void* m = pick(cur.p, nullptr);
      void* o = pick(cur.p, m);
      resblock(L.down[2 * i], cur, m, o, C, 0, 1);
      Img a{o, T, C};
      m = pick(a.p, nullptr);
      o = pick(a.p, m);
      // second block of the level: its output is the skip connection -> also the upper half of the level's concat image
      const bool skip_used = i >= 1;  // up_modules has n_levels - 1 entries: the level-0 skip is never popped (:300-308)
      resblock(L.down[2 * i + 1], a, m, o, C, 0, 1, skip_used ? W.cat[i] : nullptr, 2 * C, C, 1);
      cur = Img{o, T, C};
      if (i < nl - 1) {  // Downsample1d: Conv1d(C, C, 3, stride 2, padding 1)
        const Conv& c = L.downs[i];
        gemm(cur, PAD - 1, 2, pk + c.pk, c.co, c.Kp, prm + c.b, W.conv, rup(C, 16));
        void* dn = pick(cur.p, nullptr);
        hipLaunchKernelGGL((unet_scatter_kernel<P>), dim3((unsigned)rows), dim3(256), 0, s, W.conv, rup(C, 16),
                           cur.Tp() / 2, T / 2, 1, C, (typename P::elem_t*)dn, C, 0, 1);
        cur = Img{dn, T / 2, C};
      }
    }
    for (int i = 0; i < 2; ++i) {
      void* m = pick(cur.p, nullptr);
      const bool to_cat = i == 1 && nl >= 2;  // the last mid block feeds cat(x, skip) of the first up level
      void* o = to_cat ? W.cat[nl - 1] : pick(cur.p, m);
      const int C = L.dims[nl];
      resblock(L.mid[i], cur, m, o, to_cat ? 2 * C : C, 0, to_cat ? 0 : 1);
      cur = Img{o, cur.T, to_cat ? 2 * C : C};
    }
    for (int j = 0; j < nl - 1; ++j) {
      const int din = L.dims[nl - 1 - j];
      void* m = pick(cur.p, nullptr);
      void* o = pick(cur.p, m);
      resblock(L.up[2 * j], cur, m, o, din, 0, 1);
      Img a{o, cur.T, din};
      m = pick(a.p, nullptr);
      o = pick(a.p, m);
      resblock(L.up[2 * j + 1], a, m, o, din, 0, 1);
      Img b2{o, cur.T, din};
      // Upsample1d: ConvTranspose1d(din, din, 4, 2, 1) as one GEMM (even | odd phase)
      const Conv& c = L.ups[j];
      gemm(b2, PAD - 1, 1, pk + c.pk, 2 * din, c.Kp, nullptr, W.conv, rup(2 * din, 16));
      const int lvl = nl - 2 - j;  // the level the upsampled map lands on
      const bool to_cat = lvl >= 1;  // another up level follows: write the lower half of that level's concat image
      void* up = to_cat ? W.cat[lvl] : pick(b2.p, nullptr);
      const int ldd = to_cat ? 2 * din : din;
      up_bias_scatter(c, b2, up, ldd, din, to_cat ? 0 : 1);
      cur = Img{up, b2.T * 2, ldd};
    }
    // final_conv: Conv1dBlock(dim, dim) + Conv1d(dim, action_dim, 1)
    {
      const int C = d.dim, T = cur.T, ldc = rup(C, 16);
      conv(L.fin, cur, W.conv, ldc);
      GnArgs a;
      memset(&a, 0, sizeof(a));
      void* o = pick(cur.p, nullptr);
      a.src = W.conv, a.lds = ldc, a.Tps = cur.Tp(), a.T = T, a.C = C, a.G = d.n_groups, a.gamma = prm + L.fin_n.g;
      a.beta = prm + L.fin_n.b, a.eps = d.groupnorm_eps, a.act = d.act, a.dst = o, a.ldd = C, a.zero_pads = 1;
      hipLaunchKernelGGL((unet_gn_kernel<P>), dim3((unsigned)rows), dim3(256), 0, s, a);
      Img f{o, T, C};
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
  hipLaunchKernelGGL(unet_init_kernel, dim3(blocks), dim3(256), 0, s, noise, cfg.seed_lo, cfg.seed_hi, n, AF, W.x, chains,
                     chain_len, init_slot);
  for (int i = 0; i < n_steps; ++i) {
    const dppo_step& st = sched[i];
    const float* prm = st.net ? pf : pb;
    const char* pk = st.net ? kf : kb;
    hipLaunchKernelGGL((unet_input_kernel<P>), dim3((unsigned)B), dim3(256), 0, s, W.x, d.horizon_steps, d.action_dim,
                       (typename P::elem_t*)W.in_img, 64);
    hipLaunchKernelGGL((unet_cond_rows_kernel<P>), dim3((unsigned)B), dim3(64), 0, s,
                       (const float*)(pk + L.temb) + (size_t)st.t * d.time_dim, (const int64_t*)nullptr, obs, d.time_dim,
                       d.cond_dim, 1, (typename P::elem_t*)W.g, L.Kg, d.larger_encoder ? -1 : d.act);
    Runner<P> R{d, L, prm, pk, W, B, s};
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
