// Visual encoder (ViT patch encoder + SpatialEmb) on gfx950: see the dppo_vis_* section of include/dppo_hip.h.
//
// The reference's pixel networks (VisionDiffusionMLP, VisionUnet1D, ViTCritic) are "encoder -> cat[feat, state] -> the same
// trunk as the state-observation network".  The feature does not depend on the denoising step, so it is computed ONCE per
// observation here and handed to the existing sampler / log-prob / loss entry points as the observation vector; the loss
// entries return d loss / d observation and dppo_vis_backward carries it to the encoder's parameters.
//
// Every contraction is one of the library's two MFMA GEMMs (gemm.h):
//   Conv2d(k8, s4)   : im2col rows [img*P1][C*64] (8 contiguous pixels per (channel, ky): 16/32-byte copies) -> gemm_nt + ReLU
//   Conv2d(k3, s2)   : im2col rows [img*P][9*D] over the channel-last map (D-element runs)             -> gemm_nt
//   qkv / out / MLP  : gemm_nt on token rows [img*P][D]
//   SpatialEmb       : rows (img, d) = [feat[:, d] over the P patches | state], K padded to 64          -> gemm_nt
// and their backward (data gradients: gemm_nt on packed transposes; weight gradients: gemm_tn over the forward's operands).
// Around them: LayerNorm (one wave per row), attention (one workgroup per (image, head): K/V in LDS, fp32, online softmax;
// backward recomputes the probabilities from the saved log-sum-exp, one pass per query row for dQ and one per key row for
// dK/dV -- no atomics, bit-reproducible), exact-erf GELU, the SpatialEmb reduction, transposes through LDS.
#include <string.h>

#include "dppo_hip.h"
#include "gemm.h"

namespace dppo {
int api_fail(int code, const char* msg);
int api_check_launch();

namespace {

constexpr int MAX_DEPTH = 4;
constexpr float LN_EPS = 1e-5f;
inline int rup(int x, int m) { return (x + m - 1) / m * m; }
inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

// ---------------------------------------------------------------------------------------------------------------------
// layout: float offsets into the flat parameter buffer (reference state-dict order: a module's own Parameters precede its
// children -- pos_embed leads MinVit, weight leads SpatialEmb) and byte offsets into the packed image
// ---------------------------------------------------------------------------------------------------------------------
struct VLayer {
  int64_t ln1w, ln1b, qkvw, qkvb, ow, ob, ln2w, ln2b, f1w, f1b, f2w, f2b;
  size_t pqkv, po, pf1, pf2, pqkvT, poT, pf1T, pf2T;
};
struct VComp {
  int64_t w, pw, pb, lnw, lnb;
  size_t pp, ppT;
};
struct VLayout {
  int C, H, W, H1, W1, H2, W2, P1, P, D, nh, hd, depth, S, prop, nimg;
  int K1p, K2, Ksp, Pp;  // conv1 / conv2 / SpatialEmb contraction depths; Pp = patch columns kept of dXs (multiple of 64)
  int64_t pos, c1w, c1b, c2w, c2b, nw, nb, n_params;
  VLayer L[MAX_DEPTH];
  VComp cmp[2];
  size_t pc1, pc2, pc2T, pk_bytes;
};

int check_desc(const dppo_vis_desc* d) {
  if (!d) return api_fail(-1, "null descriptor");
  if (d->in_ch < 3 || d->in_ch > 24 || d->in_ch % 3) return api_fail(-1, "vis: in_ch must be 3 * img_cond_steps (3..24)");
  if (d->img_h < 16 || d->img_w < 16 || d->img_h > 512 || d->img_w > 512) return api_fail(-1, "vis: image size out of range");
  if (d->img_w % 4 || (d->img_h - 8) % 4 || (d->img_w - 8) % 4) return api_fail(-1, "vis: image sides must be 8 + 4 n");
  const int h1 = (d->img_h - 8) / 4 + 1, w1 = (d->img_w - 8) / 4 + 1;
  if ((h1 - 3) % 2 || (w1 - 3) % 2) return api_fail(-1, "vis: Conv2d(k3, s2) does not tile the first map (pos_embed would not fit)");
  if (d->embed_dim % 64 || d->embed_dim < 64 || d->embed_dim > 512) return api_fail(-1, "vis: embed_dim must be a multiple of 64 <= 512");
  if (d->num_heads < 1 || d->embed_dim % d->num_heads) return api_fail(-1, "vis: embed_dim % num_heads != 0");
  const int hd = d->embed_dim / d->num_heads;
  if (hd != 16 && hd != 32 && hd != 64) return api_fail(-1, "vis: head dimension must be 16, 32 or 64");
  if (d->depth < 1 || d->depth > MAX_DEPTH) return api_fail(-1, "vis: depth must be 1..4");
  if (d->embed_norm != 0) return api_fail(-1, "vis: embed_norm is not built (no shipped cfg sets it)");
  if (d->prop_dim < 0 || d->prop_dim > 1024) return api_fail(-1, "vis: prop_dim out of range");
  if (d->spatial_emb % 64 || d->spatial_emb < 64 || d->spatial_emb > 512) return api_fail(-1, "vis: spatial_emb must be a multiple of 64 <= 512");
  if (d->num_img < 1 || d->num_img > 2) return api_fail(-1, "vis: num_img must be 1 or 2");
  const int P = ((h1 - 3) / 2 + 1) * ((w1 - 3) / 2 + 1);
  if (P > 240) return api_fail(-1, "vis: more than 240 patches (the transposes and the attention keep one image's tokens in LDS)");
  return 0;
}

VLayout make_layout(const dppo_vis_desc& d, int es) {
  VLayout L;
  memset(&L, 0, sizeof(L));
  L.C = d.in_ch, L.H = d.img_h, L.W = d.img_w;
  L.H1 = (L.H - 8) / 4 + 1, L.W1 = (L.W - 8) / 4 + 1, L.H2 = (L.H1 - 3) / 2 + 1, L.W2 = (L.W1 - 3) / 2 + 1;
  L.P1 = L.H1 * L.W1, L.P = L.H2 * L.W2, L.D = d.embed_dim, L.nh = d.num_heads, L.hd = L.D / L.nh, L.depth = d.depth;
  L.S = d.spatial_emb, L.prop = d.prop_dim, L.nimg = d.num_img;
  L.K1p = 64 * L.C, L.K2 = 9 * L.D, L.Ksp = rup(L.P + L.prop, 64), L.Pp = rup(L.P, 64);
  const int D = L.D;
  int64_t o = 0;
  size_t pk = 0;
  auto prm = [&](int64_t n) { int64_t r = o; o += n; return r; };
  auto img = [&](size_t elems) { size_t r = pk; pk = al(pk + elems * es); return r; };
  L.pos = prm((int64_t)L.P * D);
  L.c1w = prm((int64_t)D * L.C * 64), L.c1b = prm(D);
  L.c2w = prm((int64_t)D * D * 9), L.c2b = prm(D);
  L.pc1 = img((size_t)D * L.K1p), L.pc2 = img((size_t)D * L.K2), L.pc2T = img((size_t)L.K2 * D);
  for (int l = 0; l < L.depth; ++l) {
    VLayer& v = L.L[l];
    v.ln1w = prm(D), v.ln1b = prm(D);
    v.qkvw = prm((int64_t)3 * D * D), v.qkvb = prm(3 * D);
    v.ow = prm((int64_t)D * D), v.ob = prm(D);
    v.ln2w = prm(D), v.ln2b = prm(D);
    v.f1w = prm((int64_t)4 * D * D), v.f1b = prm(4 * D);
    v.f2w = prm((int64_t)4 * D * D), v.f2b = prm(D);
    v.pqkv = img((size_t)3 * D * D), v.po = img((size_t)D * D), v.pf1 = img((size_t)4 * D * D), v.pf2 = img((size_t)4 * D * D);
    v.pqkvT = img((size_t)3 * D * D), v.poT = img((size_t)D * D), v.pf1T = img((size_t)4 * D * D), v.pf2T = img((size_t)4 * D * D);
  }
  L.nw = prm(D), L.nb = prm(D);
  for (int n = 0; n < L.nimg; ++n) {
    VComp& c = L.cmp[n];
    c.w = prm((int64_t)D * L.S);
    c.pw = prm((int64_t)L.S * (L.P + L.prop)), c.pb = prm(L.S);
    c.lnw = prm(L.S), c.lnb = prm(L.S);
    c.pp = img((size_t)L.S * L.Ksp), c.ppT = img((size_t)L.Pp * L.S);
  }
  L.n_params = o, L.pk_bytes = pk;
  return L;
}

// ---------------------------------------------------------------------------------------------------------------------
// weight packing: one launch over a job table
// ---------------------------------------------------------------------------------------------------------------------
struct PackJob {
  const float* src;
  void* dst;
  int rows, cols;  // destination extent (rows x cols elements, cols = row stride)
  int mode;        // 0: dst[r][c] = src[r * a + c], c < a ; 1: dst[r][c] = src[c * a + r], r < a, c < b (transpose)
                   // 2: conv2 dst[o][(kk) * D + i] = src[(o * D + i) * 9 + kk] ; 3: its transpose dst[kk * D + i][o]
  int a, b;
};
constexpr int MAX_PACK_JOBS = 48;
struct PackJobs {
  PackJob j[MAX_PACK_JOBS];
  int base[MAX_PACK_JOBS + 1];  // first block of job i
  int n;
};
template <class P>
__global__ __launch_bounds__(256) void vis_pack_kernel(const PackJobs jobs) {
  int ji = 0;
  while (ji + 1 < jobs.n && (int)blockIdx.x >= jobs.base[ji + 1]) ++ji;
  const PackJob& J = jobs.j[ji];
  const size_t i = (size_t)(blockIdx.x - jobs.base[ji]) * 256 + threadIdx.x;
  if (i >= (size_t)J.rows * J.cols) return;
  const int r = (int)(i / J.cols), c = (int)(i % J.cols);
  float v = 0.f;
  if (J.mode == 0) {
    if (c < J.a) v = J.src[(size_t)r * J.a + c];
  } else if (J.mode == 1) {
    if (r < J.a && c < J.b) v = J.src[(size_t)c * J.a + r];
  } else if (J.mode == 2) {
    const int D = J.a, kk = c / D, ci = c % D;
    v = J.src[((size_t)r * D + ci) * 9 + kk];
  } else {
    const int D = J.a, kk = r / D, ci = r % D;
    v = J.src[((size_t)c * D + ci) * 9 + kk];
  }
  ((typename P::elem_t*)J.dst)[i] = P::from_f32(v);
}

template <class P>
int pack_impl(const dppo_vis_desc& d, const float* prm, char* pk, hipStream_t s) {
  const VLayout L = make_layout(d, P::ESIZE);
  PackJobs jobs;
  memset(&jobs, 0, sizeof(jobs));
  int nb = 0;
  auto add = [&](const float* src, size_t dst, int rows, int cols, int mode, int a, int b) {
    PackJob& J = jobs.j[jobs.n];
    J.src = src, J.dst = pk + dst, J.rows = rows, J.cols = cols, J.mode = mode, J.a = a, J.b = b;
    jobs.base[jobs.n++] = nb;
    nb += (int)(((size_t)rows * cols + 255) / 256);
  };
  const int D = L.D;
  add(prm + L.c1w, L.pc1, D, L.K1p, 0, L.C * 64, 0);
  add(prm + L.c2w, L.pc2, D, L.K2, 2, D, 0);
  add(prm + L.c2w, L.pc2T, L.K2, D, 3, D, 0);
  for (int l = 0; l < L.depth; ++l) {
    const VLayer& v = L.L[l];
    add(prm + v.qkvw, v.pqkv, 3 * D, D, 0, D, 0);
    add(prm + v.ow, v.po, D, D, 0, D, 0);
    add(prm + v.f1w, v.pf1, 4 * D, D, 0, D, 0);
    add(prm + v.f2w, v.pf2, D, 4 * D, 0, 4 * D, 0);
    add(prm + v.qkvw, v.pqkvT, D, 3 * D, 1, D, 3 * D);   // src [3D][D] -> dst [D][3D]
    add(prm + v.ow, v.poT, D, D, 1, D, D);
    add(prm + v.f1w, v.pf1T, D, 4 * D, 1, D, 4 * D);      // src [4D][D] -> dst [D][4D]
    add(prm + v.f2w, v.pf2T, 4 * D, D, 1, 4 * D, D);      // src [D][4D] -> dst [4D][D]
  }
  for (int n = 0; n < L.nimg; ++n) {
    const VComp& c = L.cmp[n];
    add(prm + c.pw, c.pp, L.S, L.Ksp, 0, L.P + L.prop, 0);
    // dXs = dy . Wp: operand [patch p][s] = Wp[s][p]; src [S][P + prop] -> dst rows p < P (the rest zero)
    add(prm + c.pw, c.ppT, L.Pp, L.S, 1, L.P + L.prop, L.S);  // (rows p >= P: the state columns' transposes, never read)
  }
  jobs.base[jobs.n] = nb;
  hipLaunchKernelGGL((vis_pack_kernel<P>), dim3(nb), dim3(256), 0, s, jobs);
  return api_check_launch();
}

// ---------------------------------------------------------------------------------------------------------------------
// forward kernels
// ---------------------------------------------------------------------------------------------------------------------
// cols1[m][ci * 64 + ky * 8 + kx] = rgb[img][ci][4 oy + ky][4 ox + kx] / 255 - 0.5  (the k order of nn.Conv2d's flat weight).
// rgb is the reference's cond["rgb"]: (B, T, 3 * nimg, H, W); image index = n * B + b takes planes (t, 3 n + ch), ci = 3 t + ch.
template <class P, bool U8>
__global__ __launch_bounds__(256) void vis_im2col1_kernel(const void* rgb, int64_t B, int nimg, int C, int H, int W, int H1,
                                                          int W1, typename P::elem_t* cols, int K1p) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = B * nimg * H1 * W1 * C * 8;
  if (i >= total) return;
  const int ky = (int)(i % 8), ci = (int)((i / 8) % C);
  const int64_t m = i / (8 * C);
  const int ox = (int)(m % W1), oy = (int)((m / W1) % H1);
  const int64_t img = m / ((int64_t)W1 * H1);
  const int64_t b = img % B;
  const int n = (int)(img / B);
  const int t = ci / 3, ch = ci % 3;
  const int64_t plane = (b * (C / 3) + t) * (3 * nimg) + n * 3 + ch;
  const int64_t src = (plane * H + 4 * oy + ky) * W + 4 * ox;
  float v[8];
  if (U8) {
    const uint32_t* p = (const uint32_t*)((const uint8_t*)rgb + src);  // 4 ox: dword aligned (W % 4 == 0)
    const uint32_t a = p[0], c = p[1];
    for (int k = 0; k < 4; ++k) v[k] = (float)((a >> (8 * k)) & 255), v[4 + k] = (float)((c >> (8 * k)) & 255);
  } else {
    const float4* p = (const float4*)((const float*)rgb + src);
    const float4 a = p[0], c = p[1];
    v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = c.x, v[5] = c.y, v[6] = c.z, v[7] = c.w;
  }
  typename P::elem_t* o = cols + m * K1p + ci * 64 + ky * 8;
  for (int k = 0; k < 8; ++k) o[k] = P::from_f32(v[k] / 255.0f - 0.5f);
}

// cols2[m2][(ky * 3 + kx) * D + c] = a1[img][2 oy + ky][2 ox + kx][c]; one thread per 16-byte chunk
template <class P>
__global__ __launch_bounds__(256) void vis_im2col2_kernel(const typename P::elem_t* a1, int64_t NI, int H1, int W1, int H2, int W2,
                                                          int D, typename P::elem_t* cols) {
  constexpr int V = 16 / P::ESIZE;
  const int cpr = D / V;  // chunks per D-run
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = NI * H2 * W2 * 9 * cpr;
  if (i >= total) return;
  const int ch = (int)(i % cpr), kk = (int)((i / cpr) % 9);
  const int64_t m = i / ((int64_t)9 * cpr);
  const int ox = (int)(m % W2), oy = (int)((m / W2) % H2);
  const int64_t img = m / ((int64_t)W2 * H2);
  const int ky = kk / 3, kx = kk % 3;
  const int64_t src = ((img * H1 + 2 * oy + ky) * W1 + 2 * ox + kx) * D + ch * V;
  *(u32x4*)(cols + (m * 9 + kk) * D + ch * V) = *(const u32x4*)(a1 + src);
}

__device__ __forceinline__ float wave_sum(float v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// LayerNorm over rows of D (D = 64 NE, NE <= 8): one wave per row.  x = in (+ pos[m % P]) ; optionally x is stored back
// (xout, f32); out = elem((x - mean) rstd gamma + beta); stats[m] = (mean, rstd).
template <class P>
__global__ __launch_bounds__(256) void vis_ln_kernel(const float* in, const float* pos, int Pn, float* xout, const float* gamma,
                                                     const float* beta, int64_t M, int D, typename P::elem_t* out, float* stats) {
  const int lane = threadIdx.x & 63;
  const int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const int NE = D / 64;
  float x[8];
  float s = 0.f;
  for (int e = 0; e < NE; ++e) {
    const int c = lane + 64 * e;
    float v = in[m * D + c];
    if (pos) v += pos[(m % Pn) * D + c];
    x[e] = v, s += v;
  }
  const float mean = wave_sum(s) / D;
  float q = 0.f;
  for (int e = 0; e < NE; ++e) q += (x[e] - mean) * (x[e] - mean);
  const float rstd = rsqrtf(wave_sum(q) / D + LN_EPS);
  for (int e = 0; e < NE; ++e) {
    const int c = lane + 64 * e;
    if (xout) xout[m * D + c] = x[e];
    out[m * D + c] = P::from_f32((x[e] - mean) * rstd * gamma[c] + beta[c]);
  }
  if (lane == 0 && stats) stats[2 * m] = mean, stats[2 * m + 1] = rstd;
}

// Attention of one (image, head): qkv elem [img*T][3D], columns (k h d) as the reference's rearrange (vit.py:117-119).
// K and V of the head sit in LDS as fp32; thread i owns query row i: online softmax over the T keys.
template <class P, int HD>
__global__ __launch_bounds__(128) void vis_attn_fwd_kernel(const typename P::elem_t* qkv, int T, int D, int nh, float scale,
                                                           typename P::elem_t* att, float* lse) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Ks = sm;
  float* Vs = sm + (size_t)T * HD;
  const int h = blockIdx.x % nh;
  const int64_t img = blockIdx.x / nh;
  const typename P::elem_t* base = qkv + img * T * 3 * D + h * HD;
  for (int i = threadIdx.x; i < T * HD; i += 128) {
    const int j = i / HD, c = i % HD;
    Ks[i] = P::to_f32(base[(size_t)j * 3 * D + D + c]);
    Vs[i] = P::to_f32(base[(size_t)j * 3 * D + 2 * D + c]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < T; i += 128) {
    float q[HD], o[HD];
    for (int c = 0; c < HD; ++c) q[c] = P::to_f32(base[(size_t)i * 3 * D + c]) * scale, o[c] = 0.f;
    float mx = -INFINITY, l = 0.f;
    for (int j = 0; j < T; ++j) {
      float sc = 0.f;
      for (int c = 0; c < HD; ++c) sc += q[c] * Ks[j * HD + c];
      const float mn = fmaxf(mx, sc);
      const float corr = __expf(mx - mn), p = __expf(sc - mn);
      l = l * corr + p;
      for (int c = 0; c < HD; ++c) o[c] = o[c] * corr + p * Vs[j * HD + c];
      mx = mn;
    }
    const float inv = 1.f / l;
    typename P::elem_t* dst = att + (img * T + i) * D + h * HD;
    for (int c = 0; c < HD; ++c) dst[c] = P::from_f32(o[c] * inv);
    lse[((size_t)img * nh + h) * T + i] = mx + __logf(l);
  }
}

// ---- attention on the matrix cores (bf16, head dimension 32, T <= 128 tokens: the shipped ViT) ------------------------------
// One workgroup per (image, head), eight waves.  Scores are computed TRANSPOSED, S^T = K Q^T (A operand = a 16-key tile of K,
// B operand = the wave's 16-query tile of Q; head dim 32 = one k-step), so that lane (r, g) of the MFMA result holds, for query
// r, the keys 4g + e of each key tile: a query row lives in four lanes and the softmax reductions are two xor-shuffles.
// Those registers, rounded to bf16 and packed two key tiles at a time, ARE a B operand whose k-slot (g, s) is key
// 32 blk + 16 (s / 4) + 4 g + s % 4; the other operand (V^T, K^T, ...) is staged in LDS transposed with its columns in that
// order (perm_col), so P.V and dS.K need no shuffle or LDS round trip of the probabilities.  The backward makes two passes:
// per query tile (dQ) and per key tile (dK, dV: scores computed untransposed so the registers contract over queries).
constexpr int AT_T = 128, AT_LD = 136;
__device__ __forceinline__ int perm_col(int j) {
  const int w = j & 31;
  return (j & ~31) + ((w & 15) >> 2) * 8 + (w >> 4) * 4 + (w & 3);
}
__device__ __forceinline__ u32x4 pack8(const f32x4& a, const f32x4& b) {
  u32x4 o;
  o.x = (uint32_t)f2bf(a[0]) | ((uint32_t)f2bf(a[1]) << 16);
  o.y = (uint32_t)f2bf(a[2]) | ((uint32_t)f2bf(a[3]) << 16);
  o.z = (uint32_t)f2bf(b[0]) | ((uint32_t)f2bf(b[1]) << 16);
  o.w = (uint32_t)f2bf(b[2]) | ((uint32_t)f2bf(b[3]) << 16);
  return o;
}
__device__ __forceinline__ float xor4(float v, bool mx) {  // reduce over the four lanes (r, 0..3) that share a query row
  const float a = __shfl_xor(v, 16, 64);
  v = mx ? fmaxf(v, a) : v + a;
  const float b = __shfl_xor(v, 32, 64);
  return mx ? fmaxf(v, b) : v + b;
}
// dst[d][perm_col(j)] = src[j][d] for the 32 head columns, zero beyond T
__device__ __forceinline__ void stage_transposed(uint16_t* dst, const uint16_t* src, int ld, int T) {
  for (int i = threadIdx.x; i < AT_T * 32; i += blockDim.x) {
    const int j = i >> 5, d = i & 31;
    dst[d * AT_LD + perm_col(j)] = j < T ? src[(size_t)j * ld + d] : (uint16_t)0;
  }
}
__device__ __forceinline__ u32x4 row_frag(const uint16_t* src, int ld, int row, int T, int g) {
  return row < T ? *(const u32x4*)(src + (size_t)row * ld + 8 * g) : (u32x4){0, 0, 0, 0};
}

__global__ __launch_bounds__(512) void vis_attn_fwd_mfma_kernel(const uint16_t* qkv, int T, int D, int nh, float scale, uint16_t* att,
                                                                float* lse) {
  __shared__ __attribute__((aligned(16))) uint16_t VT[32 * AT_LD];
  const int h = blockIdx.x % nh;
  const int64_t img = blockIdx.x / nh;
  const uint16_t* base = qkv + img * T * 3 * D + h * 32;
  stage_transposed(VT, base + 2 * D, 3 * D, T);
  __syncthreads();
  const int lane = threadIdx.x & 63, qt = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
  if (16 * qt >= T) return;
  const int i = 16 * qt + r;
  const u32x4 qf = row_frag(base, 3 * D, i, T, g);
  f32x4 p[8];
  float mx = -INFINITY;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    p[t] = BF16::mma(row_frag(base + D, 3 * D, 16 * t + r, T, g), qf, (f32x4){0.f, 0.f, 0.f, 0.f});
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      p[t][e] = 16 * t + 4 * g + e < T ? p[t][e] * scale : -INFINITY;
      mx = fmaxf(mx, p[t][e]);
    }
  }
  mx = xor4(mx, true);
  float l = 0.f;
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      p[t][e] = __expf(p[t][e] - mx);
      l += p[t][e];
    }
  l = xor4(l, false);
  f32x4 o[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int blk = 0; blk < 4; ++blk) {
    const u32x4 pf = pack8(p[2 * blk], p[2 * blk + 1]);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
      o[dt] = BF16::mma(*(const u32x4*)(VT + (16 * dt + r) * AT_LD + 32 * blk + 8 * g), pf, o[dt]);
  }
  if (i < T) {
    const float inv = 1.f / l;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      u32x2 w;
      w.x = (uint32_t)f2bf(o[dt][0] * inv) | ((uint32_t)f2bf(o[dt][1] * inv) << 16);
      w.y = (uint32_t)f2bf(o[dt][2] * inv) | ((uint32_t)f2bf(o[dt][3] * inv) << 16);
      *(u32x2*)(att + (img * T + i) * D + h * 32 + 16 * dt + 4 * g) = w;
    }
    if (g == 0) lse[((size_t)img * nh + h) * T + i] = mx + __logf(l);
  }
}

__global__ __launch_bounds__(512) void vis_attn_bwd_mfma_kernel(const uint16_t* qkv, const uint16_t* att, const uint16_t* datt,
                                                                const float* lse, int T, int D, int nh, float scale, uint16_t* dqkv) {
  __shared__ __attribute__((aligned(16))) uint16_t KT[32 * AT_LD], QT[32 * AT_LD], GT[32 * AT_LD];
  __shared__ float Ls[AT_T], Ds[AT_T];
  const int h = blockIdx.x % nh;
  const int64_t img = blockIdx.x / nh;
  const uint16_t* base = qkv + img * T * 3 * D + h * 32;
  const uint16_t* gbase = datt + img * T * D + h * 32;
  const uint16_t* obase = att + img * T * D + h * 32;
  stage_transposed(QT, base, 3 * D, T);
  stage_transposed(KT, base + D, 3 * D, T);
  stage_transposed(GT, gbase, D, T);
  for (int i = threadIdx.x; i < AT_T; i += blockDim.x) {
    float dl = 0.f;
    if (i < T)
      for (int c = 0; c < 32; ++c) dl += bf2f(gbase[(size_t)i * D + c]) * bf2f(obase[(size_t)i * D + c]);
    Ds[i] = dl;
    Ls[i] = i < T ? lse[((size_t)img * nh + h) * T + i] : 1e30f;  // padded queries: exp(s - 1e30) = 0
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wt = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
  if (16 * wt >= T) return;
  uint16_t* dbase = dqkv + img * T * 3 * D + h * 32;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  {  // pass A: this wave's query tile -> dQ
    const int i = 16 * wt + r;
    const u32x4 qf = row_frag(base, 3 * D, i, T, g), gf = row_frag(gbase, D, i, T, g);
    const float li = Ls[i], di = Ds[i];
    f32x4 ds[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const f32x4 sa = BF16::mma(row_frag(base + D, 3 * D, 16 * t + r, T, g), qf, zero);
      const f32x4 pa = BF16::mma(row_frag(base + 2 * D, 3 * D, 16 * t + r, T, g), gf, zero);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        ds[t][e] = 16 * t + 4 * g + e < T ? __expf(sa[e] * scale - li) * (pa[e] - di) : 0.f;
    }
    f32x4 dq[2] = {zero, zero};
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
      const u32x4 bf = pack8(ds[2 * blk], ds[2 * blk + 1]);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
        dq[dt] = BF16::mma(*(const u32x4*)(KT + (16 * dt + r) * AT_LD + 32 * blk + 8 * g), bf, dq[dt]);
    }
    if (i < T)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        u32x2 w;
        w.x = (uint32_t)f2bf(dq[dt][0] * scale) | ((uint32_t)f2bf(dq[dt][1] * scale) << 16);
        w.y = (uint32_t)f2bf(dq[dt][2] * scale) | ((uint32_t)f2bf(dq[dt][3] * scale) << 16);
        *(u32x2*)(dbase + (size_t)i * 3 * D + 16 * dt + 4 * g) = w;
      }
  }
  {  // pass B: this wave's key tile -> dK, dV (scores untransposed: lane (r, g) holds key r, queries 4g + e of each tile)
    const int j = 16 * wt + r;
    const u32x4 kf = row_frag(base + D, 3 * D, j, T, g), vf = row_frag(base + 2 * D, 3 * D, j, T, g);
    f32x4 pp[8], ps[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const f32x4 sa = BF16::mma(row_frag(base, 3 * D, 16 * t + r, T, g), kf, zero);
      const f32x4 pa = BF16::mma(row_frag(gbase, D, 16 * t + r, T, g), vf, zero);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = 16 * t + 4 * g + e;
        pp[t][e] = __expf(sa[e] * scale - Ls[i]);
        ps[t][e] = pp[t][e] * (pa[e] - Ds[i]);
      }
    }
    f32x4 dk[2] = {zero, zero}, dv[2] = {zero, zero};
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
      const u32x4 pf = pack8(pp[2 * blk], pp[2 * blk + 1]), sf = pack8(ps[2 * blk], ps[2 * blk + 1]);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        dv[dt] = BF16::mma(*(const u32x4*)(GT + (16 * dt + r) * AT_LD + 32 * blk + 8 * g), pf, dv[dt]);
        dk[dt] = BF16::mma(*(const u32x4*)(QT + (16 * dt + r) * AT_LD + 32 * blk + 8 * g), sf, dk[dt]);
      }
    }
    if (j < T)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        u32x2 w;
        w.x = (uint32_t)f2bf(dk[dt][0] * scale) | ((uint32_t)f2bf(dk[dt][1] * scale) << 16);
        w.y = (uint32_t)f2bf(dk[dt][2] * scale) | ((uint32_t)f2bf(dk[dt][3] * scale) << 16);
        *(u32x2*)(dbase + (size_t)j * 3 * D + D + 16 * dt + 4 * g) = w;
        w.x = (uint32_t)f2bf(dv[dt][0]) | ((uint32_t)f2bf(dv[dt][1]) << 16);
        w.y = (uint32_t)f2bf(dv[dt][2]) | ((uint32_t)f2bf(dv[dt][3]) << 16);
        *(u32x2*)(dbase + (size_t)j * 3 * D + 2 * D + 16 * dt + 4 * g) = w;
      }
  }
}

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
  return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.39894228040143268f * __expf(-0.5f * x * x);
}
template <class P>
__global__ __launch_bounds__(256) void vis_gelu_kernel(const typename P::elem_t* z, typename P::elem_t* a, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) a[i] = P::from_f32(gelu_f(P::to_f32(z[i])));
}
template <class P>
__global__ __launch_bounds__(256) void vis_gelu_bwd_kernel(typename P::elem_t* da, const typename P::elem_t* z, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) da[i] = P::from_f32(P::to_f32(da[i]) * gelu_grad_f(P::to_f32(z[i])));
}

// Xs[(img * D + d)][c] = feats[img][c][d] (c < P) | state[b][c - P] (c < P + prop) | 0 : the rows SpatialEmb's Linear reads
// (modules.py:33-38).  One workgroup per (image, 64 channels): through LDS so both sides are coalesced.
template <class P>
__global__ __launch_bounds__(256) void vis_to_spatial_kernel(const typename P::elem_t* feats, const float* state, int64_t B, int Pn,
                                                             int D, int prop, int Ksp, typename P::elem_t* Xs) {
  extern __shared__ __attribute__((aligned(16))) float sm[];  // [Pn][65]
  const int dch = D / 64;
  const int d0 = (blockIdx.x % dch) * 64;
  const int64_t img = blockIdx.x / dch;
  for (int i = threadIdx.x; i < Pn * 64; i += 256) {
    const int p = i / 64, c = i % 64;
    sm[p * 65 + c] = P::to_f32(feats[(img * Pn + p) * D + d0 + c]);
  }
  __syncthreads();
  const int64_t b = img % B;
  for (int i = threadIdx.x; i < 64 * Ksp; i += 256) {
    const int dd = i / Ksp, c = i % Ksp;
    float v = 0.f;
    if (c < Pn) v = sm[c * 65 + dd];
    else if (c < Pn + prop) v = state[b * prop + c - Pn];
    Xs[(img * D + d0 + dd) * Ksp + c] = P::from_f32(v);
  }
}

// z[b][col0 + s] = sum_d weight[d][s] relu(LN(y[(img, d)])[s])  (modules.py:20-24,39-40); one workgroup per image, wave w
// takes rows d = w, w + 4, ...; stats[(img * D + d)] = (mean, rstd) of the row for the backward.
__global__ __launch_bounds__(256) void vis_spatial_kernel(const float* y, const float* weight, const float* gamma, const float* beta,
                                                          int D, int S, float* stats, float* obs, int ldobs, int col0, int64_t B,
                                                          int64_t img0) {
  __shared__ float red[4][512];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t img = img0 + blockIdx.x;
  const int NE = S / 64;
  float acc[8];
  for (int e = 0; e < NE; ++e) acc[e] = 0.f;
  for (int d = w; d < D; d += 4) {
    const float* row = y + (img * D + d) * S;
    float x[8], s = 0.f;
    for (int e = 0; e < NE; ++e) x[e] = row[lane + 64 * e], s += x[e];
    const float mean = wave_sum(s) / S;
    float q = 0.f;
    for (int e = 0; e < NE; ++e) q += (x[e] - mean) * (x[e] - mean);
    const float rstd = rsqrtf(wave_sum(q) / S + LN_EPS);
    for (int e = 0; e < NE; ++e) {
      const int c = lane + 64 * e;
      acc[e] += weight[d * S + c] * fmaxf((x[e] - mean) * rstd * gamma[c] + beta[c], 0.f);
    }
    if (lane == 0) stats[2 * (img * D + d)] = mean, stats[2 * (img * D + d) + 1] = rstd;
  }
  for (int e = 0; e < NE; ++e) red[w][lane + 64 * e] = acc[e];
  __syncthreads();
  for (int c = threadIdx.x; c < S; c += 256)
    obs[(img % B) * ldobs + col0 + c] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
}
__global__ void vis_copy_state_kernel(const float* state, int64_t B, int prop, float* obs, int ldobs, int col0) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < B * prop) obs[(i / prop) * ldobs + col0 + i % prop] = state[i];
}

// ---------------------------------------------------------------------------------------------------------------------
// backward kernels
// ---------------------------------------------------------------------------------------------------------------------
// SpatialEmb backward for channel row d over a chunk of the batch: dy (elem) for the GEMM backward, and the partial sums
// of d weight[d][:] and of the LayerNorm affine gradients.  grid (D, chunks); wave w takes images w, w + 4, ... of the chunk.
template <class P>
__global__ __launch_bounds__(256) void vis_spatial_bwd_kernel(const float* y, const float* stats, const float* weight,
                                                              const float* gamma, const float* beta, const float* dobs, int lddobs,
                                                              int col0, int D, int S, int64_t B, int64_t img0, int chunks,
                                                              typename P::elem_t* dy, float* wpart, float* gpart) {
  __shared__ float red[4][3][512];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int d = blockIdx.x, ck = blockIdx.y;
  const int64_t per = (B + chunks - 1) / chunks, b0 = ck * per, b1 = b0 + per < B ? b0 + per : B;
  const int NE = S / 64;
  float aw[8], ag[8], ab[8], wd[8], gm[8], bt[8];
  for (int e = 0; e < NE; ++e) {
    const int c = lane + 64 * e;
    aw[e] = ag[e] = ab[e] = 0.f, wd[e] = weight[d * S + c], gm[e] = gamma[c], bt[e] = beta[c];
  }
  for (int64_t b = b0 + w; b < b1; b += 4) {
    const int64_t r = (img0 + b) * D + d;
    const float mean = stats[2 * r], rstd = stats[2 * r + 1];
    float xh[8], gx[8], s1 = 0.f, s2 = 0.f;
    for (int e = 0; e < NE; ++e) {
      const int c = lane + 64 * e;
      xh[e] = (y[r * S + c] - mean) * rstd;
      const float pre = xh[e] * gm[e] + bt[e];
      const float dz = dobs[b * lddobs + col0 + c];
      aw[e] += dz * fmaxf(pre, 0.f);
      const float gp = pre > 0.f ? dz * wd[e] : 0.f;
      ag[e] += gp * xh[e], ab[e] += gp;
      gx[e] = gp * gm[e];
      s1 += gx[e], s2 += gx[e] * xh[e];
    }
    s1 = wave_sum(s1) / S, s2 = wave_sum(s2) / S;
    for (int e = 0; e < NE; ++e) dy[r * S + lane + 64 * e] = P::from_f32(rstd * (gx[e] - s1 - xh[e] * s2));
  }
  for (int e = 0; e < NE; ++e) red[w][0][lane + 64 * e] = aw[e], red[w][1][lane + 64 * e] = ag[e], red[w][2][lane + 64 * e] = ab[e];
  __syncthreads();
  for (int c = threadIdx.x; c < S; c += 256) {
    wpart[((size_t)ck * D + d) * S + c] = red[0][0][c] + red[1][0][c] + red[2][0][c] + red[3][0][c];
    gpart[((size_t)ck * D + d) * 2 * S + c] = red[0][1][c] + red[1][1][c] + red[2][1][c] + red[3][1][c];
    gpart[((size_t)ck * D + d) * 2 * S + S + c] = red[0][2][c] + red[1][2][c] + red[2][2][c] + red[3][2][c];
  }
}
// out[i] = sum_k part[k][i]  (fixed order): 16 columns x 16 row-lanes per workgroup, so K partial rows cost K / 16 dependent
// steps instead of K
__global__ __launch_bounds__(256) void vis_reduce_rows_kernel(const float* part, int K, size_t n, float* out, int accumulate) {
  __shared__ float red[16][17];
  const int c = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const size_t i = (size_t)blockIdx.x * 16 + c;
  float s = 0.f;
  if (i < n)
    for (int k = rl; k < K; k += 16) s += part[(size_t)k * n + i];
  red[rl][c] = s;
  __syncthreads();
  if (rl == 0 && i < n) {
    float t = 0.f;
    for (int k = 0; k < 16; ++k) t += red[k][c];
    out[i] = accumulate ? out[i] + t : t;
  }
}
// dfeats[img][p][d] (f32) = dXs[(img * D + d)][p]
template <class P>
__global__ __launch_bounds__(256) void vis_from_spatial_kernel(const typename P::elem_t* dXs, int Pp, int Pn, int D, float* dfeats,
                                                               int accumulate) {
  extern __shared__ __attribute__((aligned(16))) float sm[];  // [64][Pn + 1]
  const int dch = D / 64;
  const int d0 = (blockIdx.x % dch) * 64;
  const int64_t img = blockIdx.x / dch;
  const int ldp = Pn + 1;
  for (int i = threadIdx.x; i < 64 * Pn; i += 256) {
    const int dd = i / Pn, p = i % Pn;
    sm[dd * ldp + p] = P::to_f32(dXs[(img * D + d0 + dd) * Pp + p]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < Pn * 64; i += 256) {
    const int p = i / 64, c = i % 64;
    float* o = dfeats + (img * Pn + p) * D + d0 + c;
    *o = accumulate ? *o + sm[c * ldp + p] : sm[c * ldp + p];
  }
}

// LayerNorm backward, one wave per row: dx = dres + rstd (g - mean(g) - xhat mean(g xhat)), g = dout gamma.  Per-block
// partial sums of d gamma / d beta go to part[block][2 D] (rows are strided over a fixed grid).
template <class P>
__global__ __launch_bounds__(256) void vis_ln_bwd_kernel(const float* dout, const float* x, const float* stats, const float* gamma,
                                                         const float* dres, int64_t M, int D, float* dx, typename P::elem_t* dxe,
                                                         float* part) {
  __shared__ float red[4][2][512];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int NE = D / 64;
  float ag[8], ab[8], gm[8];
  for (int e = 0; e < NE; ++e) ag[e] = ab[e] = 0.f, gm[e] = gamma[lane + 64 * e];
  for (int64_t m = (int64_t)blockIdx.x * 4 + w; m < M; m += (int64_t)gridDim.x * 4) {
    const float mean = stats[2 * m], rstd = stats[2 * m + 1];
    float xh[8], g[8], s1 = 0.f, s2 = 0.f;
    for (int e = 0; e < NE; ++e) {
      const int c = lane + 64 * e;
      xh[e] = (x[m * D + c] - mean) * rstd;
      const float go = dout[m * D + c];
      ag[e] += go * xh[e], ab[e] += go;
      g[e] = go * gm[e];
      s1 += g[e], s2 += g[e] * xh[e];
    }
    s1 = wave_sum(s1) / D, s2 = wave_sum(s2) / D;
    for (int e = 0; e < NE; ++e) {
      const int c = lane + 64 * e;
      float v = rstd * (g[e] - s1 - xh[e] * s2);
      if (dres) v += dres[m * D + c];
      if (dx) dx[m * D + c] = v;
      if (dxe) dxe[m * D + c] = P::from_f32(v);
    }
  }
  for (int e = 0; e < NE; ++e) red[w][0][lane + 64 * e] = ag[e], red[w][1][lane + 64 * e] = ab[e];
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += 256) {
    part[(size_t)blockIdx.x * 2 * D + c] = red[0][0][c] + red[1][0][c] + red[2][0][c] + red[3][0][c];
    part[(size_t)blockIdx.x * 2 * D + D + c] = red[0][1][c] + red[1][1][c] + red[2][1][c] + red[3][1][c];
  }
}

// Attention backward of one (image, head).  Phase A (thread = query row i): dQ_i; phase B (thread = key row j): dK_j, dV_j.
// P_ij is recomputed from the saved log-sum-exp; Delta_i = dO_i . O_i.
template <class P, int HD>
__global__ __launch_bounds__(128) void vis_attn_bwd_kernel(const typename P::elem_t* qkv, const typename P::elem_t* att,
                                                           const float* datt, const float* lse, int T, int D, int nh, float scale,
                                                           typename P::elem_t* dqkv) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Ks = sm;
  float* Vs = Ks + (size_t)T * HD;
  float* Qs = Vs + (size_t)T * HD;
  float* Gs = Qs + (size_t)T * HD;  // dO
  float* Ls = Gs + (size_t)T * HD;
  float* Ds = Ls + T;
  const int h = blockIdx.x % nh;
  const int64_t img = blockIdx.x / nh;
  const typename P::elem_t* base = qkv + img * T * 3 * D + h * HD;
  for (int i = threadIdx.x; i < T * HD; i += 128) {
    const int j = i / HD, c = i % HD;
    Qs[i] = P::to_f32(base[(size_t)j * 3 * D + c]);
    Ks[i] = P::to_f32(base[(size_t)j * 3 * D + D + c]);
    Vs[i] = P::to_f32(base[(size_t)j * 3 * D + 2 * D + c]);
    Gs[i] = datt[(img * T + j) * D + h * HD + c];
  }
  for (int i = threadIdx.x; i < T; i += 128) {
    Ls[i] = lse[((size_t)img * nh + h) * T + i];
    float dl = 0.f;
    for (int c = 0; c < HD; ++c) dl += datt[(img * T + i) * D + h * HD + c] * P::to_f32(att[(img * T + i) * D + h * HD + c]);
    Ds[i] = dl;
  }
  __syncthreads();
  typename P::elem_t* dbase = dqkv + img * T * 3 * D + h * HD;
  for (int i = threadIdx.x; i < T; i += 128) {  // phase A
    float q[HD], g[HD], dq[HD];
    for (int c = 0; c < HD; ++c) q[c] = Qs[i * HD + c] * scale, g[c] = Gs[i * HD + c], dq[c] = 0.f;
    const float li = Ls[i], di = Ds[i];
    for (int j = 0; j < T; ++j) {
      float sc = 0.f, dp = 0.f;
      for (int c = 0; c < HD; ++c) sc += q[c] * Ks[j * HD + c], dp += g[c] * Vs[j * HD + c];
      const float ds = __expf(sc - li) * (dp - di);
      for (int c = 0; c < HD; ++c) dq[c] += ds * Ks[j * HD + c];
    }
    for (int c = 0; c < HD; ++c) dbase[(size_t)i * 3 * D + c] = P::from_f32(dq[c] * scale);
  }
  for (int j = threadIdx.x; j < T; j += 128) {  // phase B
    float k[HD], v[HD], dk[HD], dv[HD];
    for (int c = 0; c < HD; ++c) k[c] = Ks[j * HD + c] * scale, v[c] = Vs[j * HD + c], dk[c] = dv[c] = 0.f;
    for (int i = 0; i < T; ++i) {
      float sc = 0.f, dp = 0.f;
      for (int c = 0; c < HD; ++c) sc += Qs[i * HD + c] * k[c], dp += Gs[i * HD + c] * v[c];
      const float p = __expf(sc - Ls[i]);
      const float ds = p * (dp - Ds[i]);
      for (int c = 0; c < HD; ++c) dv[c] += p * Gs[i * HD + c], dk[c] += ds * Qs[i * HD + c];
    }
    for (int c = 0; c < HD; ++c) {
      dbase[(size_t)j * 3 * D + D + c] = P::from_f32(dk[c] * scale);
      dbase[(size_t)j * 3 * D + 2 * D + c] = P::from_f32(dv[c]);
    }
  }
}

// dz1[img][y][x][c] = relu'(a1) * sum over the (ky, kx) windows of conv2 that cover (y, x) of dcols2; 16-byte chunks
template <class P>
__global__ __launch_bounds__(256) void vis_col2im2_kernel(const typename P::elem_t* dcols, const typename P::elem_t* a1, int64_t NI,
                                                          int H1, int W1, int H2, int W2, int D, typename P::elem_t* dz1) {
  constexpr int V = 16 / P::ESIZE;
  const int cpr = D / V;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= NI * H1 * W1 * cpr) return;
  const int ch = (int)(i % cpr);
  const int64_t m1 = i / cpr;
  const int x = (int)(m1 % W1), y = (int)((m1 / W1) % H1);
  const int64_t img = m1 / ((int64_t)W1 * H1);
  float acc[V];
  for (int k = 0; k < V; ++k) acc[k] = 0.f;
  for (int ky = 0; ky < 3; ++ky) {
    const int ty = y - ky;
    if (ty < 0 || (ty & 1) || ty / 2 >= H2) continue;
    for (int kx = 0; kx < 3; ++kx) {
      const int tx = x - kx;
      if (tx < 0 || (tx & 1) || tx / 2 >= W2) continue;
      const int64_t m2 = (img * H2 + ty / 2) * W2 + tx / 2;
      const typename P::elem_t* s = dcols + (m2 * 9 + ky * 3 + kx) * D + ch * V;
      for (int k = 0; k < V; ++k) acc[k] += P::to_f32(s[k]);
    }
  }
  const typename P::elem_t* a = a1 + m1 * D + ch * V;
  typename P::elem_t* o = dz1 + m1 * D + ch * V;
  for (int k = 0; k < V; ++k) o[k] = P::from_f32(P::to_f32(a[k]) > 0.f ? acc[k] : 0.f);
}
// column sums, two stages (fixed order).  Stage 1: a workgroup covers CT * 8 columns (16-/32-byte loads, CT = min(N / 8, 32)
// threads across) with 256 / CT row-lanes over its chunk of the rows; part[row chunk][N].  N % 8 == 0.
template <class P, bool ELEM>
__global__ __launch_bounds__(256) void vis_colsum1_kernel(const void* A, int64_t M, int N, int lda, float* part, int CT) {
  __shared__ float red[256][9];
  const int ct = threadIdx.x % CT, rl = threadIdx.x / CT, RL = 256 / CT;
  const int c0 = (blockIdx.x * CT + ct) * 8;
  const int64_t per = (M + gridDim.y - 1) / gridDim.y, r0 = (int64_t)blockIdx.y * per, r1 = r0 + per < M ? r0 + per : M;
  float acc[8];
  for (int k = 0; k < 8; ++k) acc[k] = 0.f;
  if (c0 < N)
    for (int64_t r = r0 + rl; r < r1; r += RL) {
      if (ELEM) {
        const typename P::elem_t* p = (const typename P::elem_t*)A + r * lda + c0;
        if constexpr (P::ESIZE == 2) {
          const u32x4 v = *(const u32x4*)p;
          const uint32_t w[4] = {v.x, v.y, v.z, v.w};
          for (int k = 0; k < 4; ++k) acc[2 * k] += bf2f((uint16_t)(w[k] & 0xffff)), acc[2 * k + 1] += bf2f((uint16_t)(w[k] >> 16));
        } else {
          const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
          acc[0] += a.x, acc[1] += a.y, acc[2] += a.z, acc[3] += a.w, acc[4] += b.x, acc[5] += b.y, acc[6] += b.z, acc[7] += b.w;
        }
      } else {
        const float* p = (const float*)A + r * lda + c0;
        const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
        acc[0] += a.x, acc[1] += a.y, acc[2] += a.z, acc[3] += a.w, acc[4] += b.x, acc[5] += b.y, acc[6] += b.z, acc[7] += b.w;
      }
    }
  for (int k = 0; k < 8; ++k) red[threadIdx.x][k] = acc[k];
  __syncthreads();
  if (rl == 0 && c0 < N)
    for (int k = 0; k < 8; ++k) {
      float t = 0.f;
      for (int j = 0; j < RL; ++j) t += red[j * CT + ct][k];
      part[(size_t)blockIdx.y * N + c0 + k] = t;
    }
}
// out[r * ldo + c] = sum_s slab[s][r][c] for c < N2v; mode 1: conv2 unpack out[(o * D + i) * 9 + kk] <- [o][kk * D + i].
// 16 outputs x 16 split-lanes per workgroup (fixed order)
__global__ __launch_bounds__(256) void vis_slab_out_kernel(const float* slab, int splits, int N1, int N2, int N2v, float* out, int ldo,
                                                           int mode, int D) {
  __shared__ float red[16][17];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const size_t i = (size_t)blockIdx.x * 16 + cl, n = (size_t)N1 * N2v;
  const int r = (int)(i / N2v), c = (int)(i % N2v);
  float s = 0.f;
  if (i < n)
    for (int k = rl; k < splits; k += 16) s += slab[((size_t)k * N1 + r) * N2 + c];
  red[rl][cl] = s;
  __syncthreads();
  if (rl != 0 || i >= n) return;
  float t = 0.f;
  for (int k = 0; k < 16; ++k) t += red[k][cl];
  if (mode == 0) out[(size_t)r * ldo + c] = t;
  else out[((size_t)r * D + c % D) * 9 + c / D] = t;
}

// ---------------------------------------------------------------------------------------------------------------------
// runner: carves the workspace the same way for the forward and the backward call, so the tape is the workspace
// ---------------------------------------------------------------------------------------------------------------------
DevLatch g_attn_latch[2][3];
int g_vis_mfma_attn = 1;  // tuning knob 20 (A/B against the scalar attention kernels)

template <class P>
struct VisRunner {
  typedef typename P::elem_t E;
  dppo_vis_desc d;
  VLayout L;
  const float* prm;
  const char* pk;
  int64_t B, NI, M1, M2;
  hipStream_t s;
  char* base;
  size_t off;
  bool dry;
  // tape
  E *cols1, *a1, *cols2, *h1[MAX_DEPTH], *qkv[MAX_DEPTH], *att[MAX_DEPTH], *h2[MAX_DEPTH], *z[MAX_DEPTH], *ga[MAX_DEPTH], *feats, *Xs;
  float *x[2 * MAX_DEPTH + 1], *st[2 * MAX_DEPTH + 1], *lse[MAX_DEPTH], *y, *sst;
  // backward scratch
  E *dy, *dXs, *dxe, *dA, *dqkv, *dcols2, *dz1, *dattE;
  float *dfa, *dfb, *dh, *slab, *part;
  size_t slab_floats;

  void* take(size_t bytes) {
    off = al(off);
    void* p = dry ? nullptr : base + off;
    off += bytes;
    return p;
  }
  E* elems(size_t n) { return (E*)take(n * P::ESIZE); }
  float* f32s(size_t n) { return (float*)take(n * 4); }

  VisRunner(const dppo_vis_desc& dd, const float* prm_, const char* pk_, int64_t B_, void* ws, hipStream_t s_, bool train)
      : d(dd), L(make_layout(dd, P::ESIZE)), prm(prm_), pk(pk_), B(B_), s(s_), base((char*)ws), off(0), dry(ws == nullptr) {
    NI = B * L.nimg, M1 = NI * L.P1, M2 = NI * L.P;
    const int D = L.D;
    cols1 = elems((size_t)M1 * L.K1p), a1 = elems((size_t)M1 * D), cols2 = elems((size_t)M2 * L.K2);
    for (int i = 0; i <= 2 * L.depth; ++i) x[i] = f32s((size_t)M2 * D), st[i] = f32s((size_t)M2 * 2);
    for (int l = 0; l < L.depth; ++l) {
      h1[l] = elems((size_t)M2 * D), qkv[l] = elems((size_t)M2 * 3 * D), att[l] = elems((size_t)M2 * D);
      lse[l] = f32s((size_t)NI * L.nh * L.P);
      h2[l] = elems((size_t)M2 * D), z[l] = elems((size_t)M2 * 4 * D), ga[l] = elems((size_t)M2 * 4 * D);
    }
    feats = elems((size_t)M2 * D), Xs = elems((size_t)NI * D * L.Ksp), y = f32s((size_t)NI * D * L.S), sst = f32s((size_t)NI * D * 2);
    dy = dXs = dxe = dA = dqkv = dcols2 = dz1 = dattE = nullptr;
    dfa = dfb = dh = slab = part = nullptr;
    slab_floats = 0;
    if (train) {
      dy = elems((size_t)NI * D * L.S), dXs = elems((size_t)NI * D * L.Pp), dxe = elems((size_t)M2 * D);
      dA = elems((size_t)M2 * 4 * D), dqkv = elems((size_t)M2 * 3 * D), dcols2 = elems((size_t)M2 * L.K2), dz1 = elems((size_t)M1 * D);
      dfa = f32s((size_t)M2 * D), dfb = f32s((size_t)M2 * D), dh = f32s((size_t)M2 * D);
      dattE = elems((size_t)M2 * D);
      slab_floats = (size_t)64 * D * (L.K2 > L.K1p ? L.K2 : L.K1p);  // >= 128 splits of a D x 4D output, 64 of the D x 9D one
      slab = f32s(slab_floats);
      {  // partial sums: SpatialEmb (8 chunks of D x 3S), LayerNorm (1024 blocks x 2D), column sums (<= 512 chunks of 256 columns + N)
        size_t pf = (size_t)8 * D * 3 * L.S;
        const size_t ln = (size_t)1024 * 2 * D, cs = (size_t)768 * 256 + 2 * (size_t)L.P * D + 4 * (size_t)L.K2;
        pf = pf > ln ? pf : ln;
        part = f32s((pf > cs ? pf : cs) + 4096);
      }
    }
  }
  size_t bytes() const { return al(off); }

  void gemm(const void* X, int ldx, int64_t M, size_t Wp, int N, int Kp, const float* bias, float* o32, int ldo32, void* opre,
            void* oact, int ldo, int act, const float* res = nullptr, int ldres = 0) {
    GemmNT g;
    memset(&g, 0, sizeof(g));
    g.X = X, g.ldx = ldx, g.M = (int)M, g.N = N, g.Kp = Kp, g.W = pk + Wp, g.ldw = Kp, g.bias = bias;
    g.out_f32 = o32, g.ldo32 = ldo32, g.out_pre = opre, g.out_act = oact, g.ldo = ldo, g.act = act, g.res = res, g.ldres = ldres;
    launch_gemm_nt<P>(g, s);
  }
  void ln(const float* in, const float* pos, float* xout, int64_t gw, int64_t gb, E* out, float* stats) {
    hipLaunchKernelGGL((vis_ln_kernel<P>), dim3((unsigned)((M2 + 3) / 4)), dim3(256), 0, s, in, pos, L.P, xout, prm + gw, prm + gb,
                       M2, L.D, out, stats);
  }
  // bf16 with the shipped head geometry: attention on the matrix cores; otherwise (fp32 parity mode, other head sizes, more
  // than 128 tokens) the scalar kernels
  bool mfma_attn() const { return P::ESIZE == 2 && L.hd == 32 && L.P <= AT_T && g_vis_mfma_attn; }
  template <class K>
  void attn_attr(K kern, size_t lds, int which) {
    const int hi = L.hd == 16 ? 0 : (L.hd == 32 ? 1 : 2);
    if (lds > 64 * 1024 && g_attn_latch[which][hi].need()) {
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      g_attn_latch[which][hi].done();
    }
  }
  template <int HD>
  void attn_fwd(int l) {
    const size_t lds = (size_t)2 * L.P * HD * 4;
    attn_attr(vis_attn_fwd_kernel<P, HD>, lds, 0);
    hipLaunchKernelGGL((vis_attn_fwd_kernel<P, HD>), dim3((unsigned)(NI * L.nh)), dim3(128), lds, s, qkv[l], L.P, L.D, L.nh,
                       1.f / sqrtf((float)HD), att[l], lse[l]);
  }
  template <int HD>
  void attn_bwd(int l, const float* datt) {
    const size_t lds = ((size_t)4 * L.P * HD + 2 * L.P) * 4;
    attn_attr(vis_attn_bwd_kernel<P, HD>, lds, 1);
    hipLaunchKernelGGL((vis_attn_bwd_kernel<P, HD>), dim3((unsigned)(NI * L.nh)), dim3(128), lds, s, qkv[l], att[l], datt, lse[l],
                       L.P, L.D, L.nh, 1.f / sqrtf((float)HD), dqkv);
  }

  // obs (B, ldobs) f32 <- cat[feat (S * nimg), state (prop)]
  int forward(const void* rgb, int rgb_u8, const float* state, float* obs, int ldobs) {
    const int D = L.D;
    {
      const int64_t n = M1 * L.C * 8;
      if (rgb_u8)
        hipLaunchKernelGGL((vis_im2col1_kernel<P, true>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, rgb, B, L.nimg, L.C,
                           L.H, L.W, L.H1, L.W1, cols1, L.K1p);
      else
        hipLaunchKernelGGL((vis_im2col1_kernel<P, false>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, rgb, B, L.nimg, L.C,
                           L.H, L.W, L.H1, L.W1, cols1, L.K1p);
    }
    gemm(cols1, L.K1p, M1, L.pc1, D, L.K1p, prm + L.c1b, nullptr, 0, nullptr, a1, D, ACT_RELU);
    {
      const int64_t n = M2 * 9 * (D / (16 / P::ESIZE));
      hipLaunchKernelGGL((vis_im2col2_kernel<P>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a1, NI, L.H1, L.W1, L.H2, L.W2,
                         D, cols2);
    }
    gemm(cols2, L.K2, M2, L.pc2, D, L.K2, prm + L.c2b, x[0], D, nullptr, nullptr, 0, ACT_NONE);
    for (int l = 0; l < L.depth; ++l) {
      const VLayer& v = L.L[l];
      float *xi = x[2 * l], *xa = x[2 * l + 1], *xo = x[2 * l + 2];
      ln(xi, l == 0 ? prm + L.pos : nullptr, l == 0 ? xi : nullptr, v.ln1w, v.ln1b, h1[l], st[2 * l]);
      gemm(h1[l], D, M2, v.pqkv, 3 * D, D, prm + v.qkvb, nullptr, 0, qkv[l], nullptr, 3 * D, ACT_NONE);
      if (mfma_attn())
        hipLaunchKernelGGL(vis_attn_fwd_mfma_kernel, dim3((unsigned)(NI * L.nh)), dim3(512), 0, s, (const uint16_t*)qkv[l], L.P, D,
                           L.nh, 1.f / sqrtf(32.f), (uint16_t*)att[l], lse[l]);
      else if (L.hd == 16) attn_fwd<16>(l);
      else if (L.hd == 32) attn_fwd<32>(l);
      else attn_fwd<64>(l);
      gemm(att[l], D, M2, v.po, D, D, prm + v.ob, xa, D, nullptr, nullptr, 0, ACT_NONE, xi, D);
      ln(xa, nullptr, nullptr, v.ln2w, v.ln2b, h2[l], st[2 * l + 1]);
      gemm(h2[l], D, M2, v.pf1, 4 * D, D, prm + v.f1b, nullptr, 0, z[l], nullptr, 4 * D, ACT_NONE);
      {
        const size_t n = (size_t)M2 * 4 * D;
        hipLaunchKernelGGL((vis_gelu_kernel<P>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, z[l], ga[l], n);
      }
      gemm(ga[l], 4 * D, M2, v.pf2, D, 4 * D, prm + v.f2b, xo, D, nullptr, nullptr, 0, ACT_NONE, xa, D);
    }
    ln(x[2 * L.depth], nullptr, nullptr, L.nw, L.nb, feats, st[2 * L.depth]);
    hipLaunchKernelGGL((vis_to_spatial_kernel<P>), dim3((unsigned)(NI * (D / 64))), dim3(256), (size_t)L.P * 65 * 4, s, feats, state,
                       B, L.P, D, L.prop, L.Ksp, Xs);
    for (int n = 0; n < L.nimg; ++n) {
      const VComp& c = L.cmp[n];
      const size_t r0 = (size_t)n * B * D;
      gemm(Xs + r0 * L.Ksp, L.Ksp, B * D, c.pp, L.S, L.Ksp, prm + c.pb, y + r0 * L.S, L.S, nullptr, nullptr, 0, ACT_NONE);
      hipLaunchKernelGGL(vis_spatial_kernel, dim3((unsigned)B), dim3(256), 0, s, y, prm + c.w, prm + c.lnw, prm + c.lnb, D, L.S, sst,
                         obs, ldobs, n * L.S, B, (int64_t)n * B);
    }
    if (L.prop > 0)
      hipLaunchKernelGGL(vis_copy_state_kernel, dim3((unsigned)((B * L.prop + 255) / 256)), dim3(256), 0, s, state, B, L.prop, obs,
                         ldobs, L.nimg * L.S);
    return api_check_launch();
  }

  // ---- backward helpers
  // out[N1][ldo] (first N2v columns) = A^T . B over M rows
  void wgrad(const void* A, int lda, int N1, const void* Bm, int ldb, int N2, int N2v, int64_t M, float* out, int ldo, int mode = 0) {
    // the outputs are small (one to a few dozen 128 x 128 tiles) and M is large: split the rows until ~512 workgroups exist
    const int64_t tiles = (int64_t)((N1 + 127) / 128) * ((N2 + 127) / 128);
    int64_t splits = (512 + tiles - 1) / tiles;
    if (splits > (M + 255) / 256) splits = (M + 255) / 256;
    if (splits > 128) splits = 128;
    if (splits < 1) splits = 1;
    while (splits > 1 && (size_t)splits * N1 * N2 > slab_floats) --splits;
    int64_t rps = ((M + splits - 1) / splits + 63) / 64 * 64;
    splits = (M + rps - 1) / rps;
    GemmTN t;
    memset(&t, 0, sizeof(t));
    t.A = A, t.B = Bm, t.M = (int)M, t.N1 = N1, t.N2 = N2, t.lda = lda, t.ldb = ldb;
    t.slab = slab, t.ldc = N2, t.splits = (int)splits, t.rows_per_split = (int)rps;
    launch_gemm_tn<P>(t, s);
    const size_t n = (size_t)N1 * N2v;
    hipLaunchKernelGGL(vis_slab_out_kernel, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, s, slab, (int)splits, N1, N2, N2v, out,
                       ldo, mode, L.D);
  }
  void colsum(const void* A, bool elem, int64_t M, int N, int lda, float* out) {
    const int CT = N / 8 < 32 ? N / 8 : 32;
    const int cb = (N / 8 + CT - 1) / CT;
    int rb = (int)((M + 255) / 256);  // >= 256 rows per workgroup
    const int cap = (512 + cb - 1) / cb;
    rb = rb < 1 ? 1 : (rb > cap ? cap : rb);
    if (elem)
      hipLaunchKernelGGL((vis_colsum1_kernel<P, true>), dim3(cb, rb), dim3(256), 0, s, A, M, N, lda, part, CT);
    else
      hipLaunchKernelGGL((vis_colsum1_kernel<P, false>), dim3(cb, rb), dim3(256), 0, s, A, M, N, lda, part, CT);
    reduce_rows(part, rb, (size_t)N, out);
  }
  void reduce_rows(const float* src, int K, size_t n, float* out) {
    hipLaunchKernelGGL(vis_reduce_rows_kernel, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, s, src, K, n, out, 0);
  }
  // dx = dres + LNbwd(dout); d gamma / d beta -> grad
  void ln_bwd(const float* dout, const float* xin, const float* stats, int64_t gw, int64_t gb, const float* dres, float* dx, E* dxe_,
              float* grad) {
    const int blocks = M2 >= 16384 ? 1024 : 256;
    hipLaunchKernelGGL((vis_ln_bwd_kernel<P>), dim3(blocks), dim3(256), 0, s, dout, xin, stats, prm + gw, dres, M2, L.D, dx, dxe_,
                       part);
    // part [blocks][2 D] -> (d gamma | d beta): weight and bias of a LayerNorm are adjacent in the flat layout
    (void)gb;
    reduce_rows(part, blocks, (size_t)2 * L.D, grad + gw);
  }

  // dobs (B, lddobs): d loss / d obs; the first S * nimg columns are read.  grad: flat, every entry written.
  int backward(const float* dobs, int lddobs, float* grad) {
    const int D = L.D, S = L.S;
    const int chunks = 8;
    for (int n = 0; n < L.nimg; ++n) {
      const VComp& c = L.cmp[n];
      const size_t r0 = (size_t)n * B * D;
      float* wpart = part;
      float* gpart = part + (size_t)chunks * D * S;
      hipLaunchKernelGGL((vis_spatial_bwd_kernel<P>), dim3(D, chunks), dim3(256), 0, s, y, sst, prm + c.w, prm + c.lnw, prm + c.lnb,
                         dobs, lddobs, n * S, D, S, B, (int64_t)n * B, chunks, dy, wpart, gpart);
      reduce_rows(wpart, chunks, (size_t)D * S, grad + c.w);
      // LayerNorm affine gradients: rows (chunk, d) of [2 S] -> one [2 S] vector (gamma | beta adjacent in the flat layout)
      reduce_rows(gpart, chunks * D, (size_t)2 * S, grad + c.lnw);
      wgrad(dy + r0 * S, S, S, Xs + r0 * L.Ksp, L.Ksp, L.Ksp, L.P + L.prop, B * D, grad + c.pw, L.P + L.prop);
      colsum(dy + r0 * S, true, B * D, S, S, grad + c.pb);
      gemm(dy + r0 * S, S, B * D, c.ppT, L.Pp, S, nullptr, nullptr, 0, dXs + r0 * L.Pp, nullptr, L.Pp, ACT_NONE);
    }
    hipLaunchKernelGGL((vis_from_spatial_kernel<P>), dim3((unsigned)(NI * (D / 64))), dim3(256), (size_t)64 * (L.P + 1) * 4, s, dXs,
                       L.Pp, L.P, D, dfa, 0);
    // final LayerNorm
    float *dcur = dfb, *dalt = dfa;
    ln_bwd(dfa, x[2 * L.depth], st[2 * L.depth], L.nw, L.nb, nullptr, dcur, dxe, grad);
    for (int l = L.depth - 1; l >= 0; --l) {
      const VLayer& v = L.L[l];
      // x_out = x_a + f2(gelu(f1(LN2(x_a)))) : dcur = d x_out (f32), dxe = the same as elem
      wgrad(dxe, D, D, ga[l], 4 * D, 4 * D, 4 * D, M2, grad + v.f2w, 4 * D);
      colsum(dcur, false, M2, D, D, grad + v.f2b);
      gemm(dxe, D, M2, v.pf2T, 4 * D, D, nullptr, nullptr, 0, dA, nullptr, 4 * D, ACT_NONE);
      {
        const size_t n = (size_t)M2 * 4 * D;
        hipLaunchKernelGGL((vis_gelu_bwd_kernel<P>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dA, z[l], n);
      }
      wgrad(dA, 4 * D, 4 * D, h2[l], D, D, D, M2, grad + v.f1w, D);
      colsum(dA, true, M2, 4 * D, 4 * D, grad + v.f1b);
      gemm(dA, 4 * D, M2, v.pf1T, D, 4 * D, nullptr, dh, D, nullptr, nullptr, 0, ACT_NONE);
      ln_bwd(dh, x[2 * l + 1], st[2 * l + 1], v.ln2w, v.ln2b, dcur, dalt, dxe, grad);  // dalt = d x_a
      // x_a = x_i + out_proj(attn(qkv(LN1(x_i))))
      wgrad(dxe, D, D, att[l], D, D, D, M2, grad + v.ow, D);
      colsum(dalt, false, M2, D, D, grad + v.ob);
      if (mfma_attn()) {
        gemm(dxe, D, M2, v.poT, D, D, nullptr, nullptr, 0, dattE, nullptr, D, ACT_NONE);  // d att (bf16 operand)
        hipLaunchKernelGGL(vis_attn_bwd_mfma_kernel, dim3((unsigned)(NI * L.nh)), dim3(512), 0, s, (const uint16_t*)qkv[l],
                           (const uint16_t*)att[l], (const uint16_t*)dattE, lse[l], L.P, D, L.nh, 1.f / sqrtf(32.f), (uint16_t*)dqkv);
      } else {
        gemm(dxe, D, M2, v.poT, D, D, nullptr, dh, D, nullptr, nullptr, 0, ACT_NONE);  // d att (f32)
        if (L.hd == 16) attn_bwd<16>(l, dh);
        else if (L.hd == 32) attn_bwd<32>(l, dh);
        else attn_bwd<64>(l, dh);
      }
      wgrad(dqkv, 3 * D, 3 * D, h1[l], D, D, D, M2, grad + v.qkvw, D);
      colsum(dqkv, true, M2, 3 * D, 3 * D, grad + v.qkvb);
      gemm(dqkv, 3 * D, M2, v.pqkvT, D, 3 * D, nullptr, dh, D, nullptr, nullptr, 0, ACT_NONE);
      ln_bwd(dh, x[2 * l], st[2 * l], v.ln1w, v.ln1b, dalt, dcur, dxe, grad);  // dcur = d x_i
    }
    // x_0 = conv2(relu(conv1(img))) + pos
    colsum(dcur, false, NI, L.P * D, L.P * D, grad + L.pos);  // d pos[p][c] = sum over images
    wgrad(dxe, D, D, cols2, L.K2, L.K2, L.K2, M2, grad + L.c2w, 0, 1);
    colsum(dcur, false, M2, D, D, grad + L.c2b);
    gemm(dxe, D, M2, L.pc2T, L.K2, D, nullptr, nullptr, 0, dcols2, nullptr, L.K2, ACT_NONE);
    {
      const int64_t n = M1 * (D / (16 / P::ESIZE));
      hipLaunchKernelGGL((vis_col2im2_kernel<P>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dcols2, a1, NI, L.H1, L.W1,
                         L.H2, L.W2, D, dz1);
    }
    wgrad(dz1, D, D, cols1, L.K1p, L.K1p, L.C * 64, M1, grad + L.c1w, L.C * 64);
    colsum(dz1, true, M1, D, D, grad + L.c1b);
    return api_check_launch();
  }
};

int check_prec_v(int prec) {
  if (prec != DPPO_PREC_F32 && prec != DPPO_PREC_BF16) return api_fail(-1, "prec must be DPPO_PREC_F32 or DPPO_PREC_BF16");
  return 0;
}
#define VIS_DISPATCH(prec, CALL) ((prec) == DPPO_PREC_F32 ? CALL(F32) : CALL(BF16))

template <class P>
int64_t ws_bytes(const dppo_vis_desc& d, int64_t B, int train) {
  VisRunner<P> r(d, nullptr, nullptr, B, nullptr, nullptr, train != 0);
  return (int64_t)r.bytes();
}
template <class P>
int encode_impl(const dppo_vis_desc& d, const float* prm, const char* pk, const void* rgb, int u8, const float* state, int64_t B,
                float* obs, int ldobs, int train, void* ws, int64_t wsb, hipStream_t s) {
  VisRunner<P> r(d, prm, pk, B, ws, s, train != 0);
  if ((int64_t)r.bytes() > wsb) return api_fail(-1, "vis: workspace too small");
  return r.forward(rgb, u8, state, obs, ldobs);
}
template <class P>
int backward_impl(const dppo_vis_desc& d, const float* prm, const char* pk, const float* dobs, int lddobs, int64_t B, float* grad,
                  void* ws, int64_t wsb, hipStream_t s) {
  VisRunner<P> r(d, prm, pk, B, ws, s, true);
  if ((int64_t)r.bytes() > wsb) return api_fail(-1, "vis: workspace too small");
  return r.backward(dobs, lddobs, grad);
}

}  // namespace
void set_vis_mfma_attn(int v) { g_vis_mfma_attn = v; }
}  // namespace dppo

using namespace dppo;

int64_t dppo_vis_param_count(const dppo_vis_desc* d) {
  if (check_desc(d)) return -1;
  return make_layout(*d, 4).n_params;
}
int64_t dppo_vis_packed_bytes(const dppo_vis_desc* d, int prec) {
  if (check_desc(d) || check_prec_v(prec)) return -1;
  return (int64_t)make_layout(*d, prec == DPPO_PREC_F32 ? 4 : 2).pk_bytes;
}
int dppo_vis_pack(const dppo_vis_desc* d, int prec, const float* params, void* packed, dppo_stream_t stream) {
  if (int e = check_desc(d)) return e;
  if (int e = check_prec_v(prec)) return e;
  if (!params || !packed) return api_fail(-1, "null pointer");
#define CALL(P) pack_impl<P>(*d, params, (char*)packed, (hipStream_t)stream)
  return VIS_DISPATCH(prec, CALL);
#undef CALL
}
int64_t dppo_vis_workspace_bytes(const dppo_vis_desc* d, int prec, int64_t B, int train) {
  if (check_desc(d) || check_prec_v(prec)) return -1;
  if (B < 1 || B > 65536) return api_fail(-1, "vis: B out of range (1..65536 images per call)");
#define CALL(P) ws_bytes<P>(*d, B, train)
  return VIS_DISPATCH(prec, CALL);
#undef CALL
}
int dppo_vis_encode(const dppo_vis_desc* d, int prec, const float* params, const void* packed, const void* rgb, int rgb_u8,
                    const float* state, int64_t B, float* obs, int ld_obs, int train, void* workspace, int64_t workspace_bytes,
                    dppo_stream_t stream) {
  if (int e = check_desc(d)) return e;
  if (int e = check_prec_v(prec)) return e;
  if (!params || !packed || !rgb || !obs || !workspace || (d->prop_dim > 0 && !state)) return api_fail(-1, "null pointer");
  if (B < 1 || B > 65536) return api_fail(-1, "vis: B out of range (1..65536 images per call)");
  if (ld_obs < d->spatial_emb * d->num_img + d->prop_dim) return api_fail(-1, "vis: ld_obs < feat_dim + prop_dim");
#define CALL(P) \
  encode_impl<P>(*d, params, (const char*)packed, rgb, rgb_u8, state, B, obs, ld_obs, train, workspace, workspace_bytes, (hipStream_t)stream)
  return VIS_DISPATCH(prec, CALL);
#undef CALL
}
int dppo_vis_backward(const dppo_vis_desc* d, int prec, const float* params, const void* packed, const float* d_obs, int ld_dobs,
                      int64_t B, float* grad, void* workspace, int64_t workspace_bytes, dppo_stream_t stream) {
  if (int e = check_desc(d)) return e;
  if (int e = check_prec_v(prec)) return e;
  if (!params || !packed || !d_obs || !grad || !workspace) return api_fail(-1, "null pointer");
  if (B < 1 || B > 65536) return api_fail(-1, "vis: B out of range (1..65536 images per call)");
  if (ld_dobs < d->spatial_emb * d->num_img) return api_fail(-1, "vis: ld_dobs < feat_dim");
#define CALL(P) backward_impl<P>(*d, params, (const char*)packed, d_obs, ld_dobs, B, grad, workspace, workspace_bytes, (hipStream_t)stream)
  return VIS_DISPATCH(prec, CALL);
#undef CALL
}
