// See ppo.h.  gfx950 only.  Compiled with -ffp-contract=off: the loss / posterior arithmetic keeps
// the reference's op sequence.
#include "ppo.h"
#include "pack_dev.h"
#include "posterior.h"
#include "post_blocks.h"

namespace dppo {

// =================================================================================================
// packing
// =================================================================================================
template <class P>
__global__ void cast_pad_kernel(const float* src, int rows, int cols, int lds, typename P::elem_t* dst, int ldd) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)rows * ldd) return;
  const int r = (int)(i / ldd), c = (int)(i % ldd);
  dst[i] = P::from_f32(c < cols ? src[(size_t)r * lds + c] : 0.f);
}
template <class P>
void launch_cast_pad(const float* src, int rows, int cols, int lds, void* dst, int ldd, hipStream_t s) {
  const size_t n = (size_t)rows * ldd;
  hipLaunchKernelGGL((cast_pad_kernel<P>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, rows, cols, lds,
                     (typename P::elem_t*)dst, ldd);
}
template void launch_cast_pad<F32>(const float*, int, int, int, void*, int, hipStream_t);
template void launch_cast_pad<BF16>(const float*, int, int, int, void*, int, hipStream_t);

template <class P>
__global__ void transpose_cast_kernel(const float* src, int rows, int cols, int lds, int coff,
                                      typename P::elem_t* dst, int ldd) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)cols * ldd) return;
  const int c = (int)(i / ldd), r = (int)(i % ldd);
  dst[i] = P::from_f32(r < rows ? src[(size_t)r * lds + coff + c] : 0.f);
}
template <class P>
void launch_transpose_cast(const float* src, int rows, int cols, int lds, int coff, void* dst, int ldd,
                           hipStream_t s) {
  const size_t n = (size_t)cols * ldd;
  hipLaunchKernelGGL((transpose_cast_kernel<P>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, rows, cols,
                     lds, coff, (typename P::elem_t*)dst, ldd);
}
template void launch_transpose_cast<F32>(const float*, int, int, int, int, void*, int, hipStream_t);
template void launch_transpose_cast<BF16>(const float*, int, int, int, int, void*, int, hipStream_t);

// sinusoid -> Linear(td,2td) -> Mish -> Linear(2td,td); one block per diffusion time
__device__ __forceinline__ void time_table_block(const float* w1, const float* b1, const float* w2, const float* b2,
                                                 int td, float* temb, const int t, float* sh) {
  float* e0 = sh;  // [td] sinusoid, [2td] hidden
  float* a1 = sh + td;
  for (int j = threadIdx.x; j < td; j += blockDim.x) e0[j] = sinus_feat(t, j, td);
  __syncthreads();
  for (int o = threadIdx.x; o < 2 * td; o += blockDim.x) {
    float s = b1[o];
    for (int j = 0; j < td; ++j) s += w1[o * td + j] * e0[j];
    a1[o] = mish_f(s);
  }
  __syncthreads();
  for (int o = threadIdx.x; o < td; o += blockDim.x) {
    float s = b2[o];
    for (int j = 0; j < 2 * td; ++j) s += w2[o * 2 * td + j] * a1[j];
    temb[(size_t)t * td + o] = s;
  }
}
__global__ void time_table_kernel(const float* w1, const float* b1, const float* w2, const float* b2, int td,
                                  float* temb) {
  extern __shared__ float sh[];
  time_table_block(w1, b1, w2, b2, td, temb, blockIdx.x, sh);
}
void launch_time_table(const float* w1, const float* b1, const float* w2, const float* b2, int td, int n_time,
                       float* temb, hipStream_t s) {
  hipLaunchKernelGGL(time_table_kernel, dim3(n_time), dim3(64), 3 * td * sizeof(float), s, w1, b1, w2, b2, td, temb);
}

constexpr int PACK_TD_MAX = 128;  // time_dim bound of the one-launch packer (3 * td floats of LDS)
template <class P>
__device__ __forceinline__ void pack_net_block(const PackNet& n, int b, float* sh) {
  const int n_stream = n.ps_x * n.ps.n_layers;
  if (b < n_stream) {
    pack_stream_block<P>(n.ps.layer[b / n.ps_x], n.ps.TPW, b % n.ps_x);
    return;
  }
  b -= n_stream;
  const int n_out = SAMPLER_WAVES * n.CNT * n.OT;
  if (b < n_out) {
    pack_out_block<P>(n.Wout, n.out_dim, n.H, n.OT, n.CNT, n.ostream, b);
    return;
  }
  b -= n_out;
  if (n.Wc != nullptr) {  // the merged out layer's second half: (Wout . W2) on act(z1) of the top block
    if (b < n_out) {
      pack_out_block<P>(n.Wc, n.out_dim, n.H, n.OT, n.CNT, n.ostream2, b);
      return;
    }
    b -= n_out;
  }
  if (n.W0c != nullptr) {  // the fused forward's merged out layer, first half: (Wout . W0) on the input rows
    const int n_out0 = SAMPLER_WAVES * n.CNT0 * n.OT;
    if (b < n_out0) {
      pack_out_block<P>(n.W0c, n.out_dim, n.Kp0s, n.OT, n.CNT0, n.ostream0, b);
      return;
    }
    b -= n_out0;
  }
  if (b < n.n_time) {
    time_table_block(n.te_w1, n.te_b1, n.te_w2, n.te_b2, n.td, n.temb, b, sh);
    return;
  }
  b -= n.n_time;
  typename P::elem_t* dst = (typename P::elem_t*)n.tdst;
  const size_t i = (size_t)b * 64 + threadIdx.x;
  if (i >= (size_t)n.t_cols * n.t_ldd) return;
  const int c = (int)(i / n.t_ldd), r = (int)(i % n.t_ldd);
  dst[i] = P::from_f32(r < n.t_rows ? n.tsrc[(size_t)r * n.t_lds + n.t_coff + c] : 0.f);
}
static size_t pack_net_blocks(const PackNet& n) {
  const size_t tblocks = ((size_t)n.t_cols * n.t_ldd + 63) / 64;
  return (size_t)n.ps_x * n.ps.n_layers + (size_t)SAMPLER_WAVES * n.CNT * n.OT * (n.Wc ? 2 : 1) +
         (n.W0c ? (size_t)SAMPLER_WAVES * n.CNT0 * n.OT : 0) + n.n_time + tblocks;
}
template <class P>
__global__ __launch_bounds__(64) void pack_net_kernel(const PackNet n) {
  __shared__ float sh[3 * PACK_TD_MAX];
  pack_net_block<P>(n, blockIdx.x, sh);
}
// two networks (actor_ft and critic after an optimiser step) in one launch
template <class P>
__global__ __launch_bounds__(64) void pack_nets_kernel(const PackNets q) {
  __shared__ float sh[3 * PACK_TD_MAX];
  const int k = (int)blockIdx.x >= q.base1 ? 1 : 0;
  pack_net_block<P>(q.n[k], blockIdx.x - (k ? q.base1 : 0), sh);
}
template <class P>
void launch_pack_nets(PackNets& q, hipStream_t s) {
  q.base1 = (int)pack_net_blocks(q.n[0]);
  const size_t blocks = (size_t)q.base1 + pack_net_blocks(q.n[1]);
  hipLaunchKernelGGL((pack_nets_kernel<P>), dim3((unsigned)blocks), dim3(64), 0, s, q);
}
template void launch_pack_nets<F32>(PackNets&, hipStream_t);
template void launch_pack_nets<BF16>(PackNets&, hipStream_t);
template <class P>
void launch_pack_net(const PackNet& n, hipStream_t s) {
  hipLaunchKernelGGL((pack_net_kernel<P>), dim3((unsigned)pack_net_blocks(n)), dim3(64), 0, s, n);
}
template void launch_pack_net<F32>(const PackNet&, hipStream_t);
template void launch_pack_net<BF16>(const PackNet&, hipStream_t);
bool pack_net_supports(int time_dim) { return time_dim <= PACK_TD_MAX; }

// =================================================================================================
// row building
// =================================================================================================
// per-k constants of PPODiffusion.loss (only Kft distinct values exist): tab[k] = gamma_denoising^(Kft-k-1)
// (diffusion_ppo.py:138-144), tab[Kft + k] = clip range eps_k (:151-159)
__device__ __forceinline__ void loss_table_entry(const dppo_ppo_cfg& pc, int k, float* tab) {
  const int Kft = pc.ft_denoising_steps;
  tab[k] = (float)pow(pc.gamma_denoising, (double)(Kft - k - 1));
  float ek;
  if (Kft > 1) {
    const float t = (float)k / (float)(Kft - 1);
    const float num = expf((float)pc.clip_ploss_coef_rate * t) - 1.f;
    ek = (float)pc.clip_ploss_coef_base +
         (float)(pc.clip_ploss_coef - pc.clip_ploss_coef_base) * num / (float)(exp(pc.clip_ploss_coef_rate) - 1.0);
  } else {
    ek = (float)k / (float)(Kft - 1);
  }
  tab[Kft + k] = ek;
}

// One thread per 16-byte chunk of an output row (whole 128-byte rows per 8 lanes: coalesced stores), 256 threads per
// block; blocks stride over the chunks of inA first, then of inC.  Block 0 also zeroes the step's accumulators
// (zero_a, zero_b), which saves two memset packets and the pipeline bubble behind them.
template <class P>
__global__ __launch_bounds__(256) void build_rows_kernel(const BuildRows a) {
  constexpr int EPC = 16 / P::ESIZE;  // elements per chunk
  if (blockIdx.x == 0) {
    if ((int)threadIdx.x < a.n_zero_a) a.zero_a[threadIdx.x] = 0.0;
    if ((int)threadIdx.x >= 128 && (int)threadIdx.x - 128 < a.n_zero_b) a.zero_b[threadIdx.x - 128] = 0.0;
    if ((int)threadIdx.x < a.n_zero_c) a.zero_c[threadIdx.x] = 0.0;
    if (a.loss_tab != nullptr)
      for (int k = threadIdx.x; k < a.pcfg.ft_denoising_steps; k += blockDim.x) {
        loss_table_entry(a.pcfg, k, a.loss_tab);
        // log(std_k) as THIS translation unit rounds it (the loss and log-prob kernels live here): the forward kernel's fused policy
        // loss (fused.hip is compiled with contraction on, where the logf expansion ends in an fma instead of an add) reads it
        if (3 * a.pcfg.ft_denoising_steps <= 2048) a.loss_tab[2 * a.pcfg.ft_denoising_steps + k] = logf(a.ksteps[k].std);
      }
  }
  const int ca = a.inA != nullptr ? a.KpA / EPC : 0, cc = a.inC != nullptr ? a.KpC / EPC : 0;
  const int64_t total = a.M * (ca + cc);
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (int64_t)gridDim.x * blockDim.x) {
    const bool isA = q < a.M * ca;
    const int64_t qq = isA ? q : q - a.M * ca;
    const int cpr = isA ? ca : cc;
    // (32-bit divisions where the values fit -- they always do in practice: a 64-bit one is ~100 instructions, twice per thread)
    const int64_t n = (qq >> 31) == 0 ? (int64_t)((uint32_t)qq / (uint32_t)cpr) : qq / cpr;
    const int c0 = (int)(qq - n * cpr) * EPC;
    int64_t b;
    int k;
    if (a.kinds != nullptr) {
      b = n;
      k = (int)a.kinds[n];
    } else {
      const int64_t ind = a.inds ? a.inds[n] : n;
      b = (ind >> 31) == 0 ? (int64_t)((uint32_t)ind / (uint32_t)a.Kft) : ind / a.Kft;
      k = (int)(ind - b * a.Kft);
    }
    const float* ob = a.obs + (size_t)b * a.cond;
    float v[EPC];
    if (isA) {
      if (c0 == 0) {
        a.brow[n] = (int32_t)b;
        a.krow[n] = k;
      }
      const int t = a.ksteps[k].t;
      const float* xk = a.kinds != nullptr ? a.chains + (size_t)b * 2 * a.AF : a.chains + ((size_t)b * (a.Kft + 1) + k) * a.AF;
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        const int c = c0 + e;
        float x = 0.f;
        if (c < a.AF)
          x = xk[c];
        else if (c < a.AF + a.td)
          x = a.temb[(size_t)t * a.td + (c - a.AF)];
        else if (c < a.AF + a.td + a.cond && a.obs_in_a)
          x = ob[c - a.AF - a.td];
        else if (a.onehot0 >= 0 && c == a.onehot0 + k)
          x = 1.f;
        v[e] = x;
      }
    } else {
      if (ca == 0 && c0 == 0 && a.brow != nullptr) a.brow[n] = (int32_t)b;  // critic-only launch: its own row index
#pragma unroll
      for (int e = 0; e < EPC; ++e) v[e] = c0 + e < a.cond ? ob[c0 + e] : 0.f;
    }
    u32x4 o;
    if constexpr (P::ESIZE == 4) {
      o = (u32x4){__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (uint32_t)f2bf(v[2 * e]) | ((uint32_t)f2bf(v[2 * e + 1]) << 16);
    }
    char* dst = isA ? (char*)a.inA + ((size_t)n * a.KpA + c0) * P::ESIZE : (char*)a.inC + ((size_t)n * a.KpC + c0) * P::ESIZE;
    *(u32x4*)dst = o;
  }
  // ---- rider: partial advantage moments (see BuildRows::mom_adv).  The last blocks of the grid: they are dispatched last
  // and have the fewest row chunks left
  const int rb = (int)gridDim.x - 1 - (int)blockIdx.x;  // rider index
  if (a.mom_adv != nullptr && rb < ADV_RIDER_BLOCKS) {
    __shared__ double shm[2][4];
    const int nr = (int)gridDim.x < ADV_RIDER_BLOCKS ? (int)gridDim.x : ADV_RIDER_BLOCKS;  // riders that exist
    double s1 = 0, s2 = 0;
    for (int64_t n = (int64_t)rb * 256 + threadIdx.x; n < a.M; n += (int64_t)nr * 256) {
      const int64_t b = a.kinds != nullptr ? n : (a.inds ? a.inds[n] : n) / a.Kft;
      const double v = a.mom_adv[b];
      s1 += v;
      s2 += v * v;
    }
    for (int o = 32; o > 0; o >>= 1) s1 += __shfl_down(s1, o), s2 += __shfl_down(s2, o);
    if ((threadIdx.x & 63) == 0) shm[0][threadIdx.x >> 6] = s1, shm[1][threadIdx.x >> 6] = s2;
    __syncthreads();
    if (threadIdx.x == 0) {
      a.mom_out[8 + 2 * rb] = (shm[0][0] + shm[0][1]) + (shm[0][2] + shm[0][3]);
      a.mom_out[9 + 2 * rb] = (shm[1][0] + shm[1][1]) + (shm[1][2] + shm[1][3]);
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x < 64 && (int)threadIdx.x >= nr)  // (a grid of fewer blocks: zero the rest)
      a.mom_out[8 + 2 * threadIdx.x] = 0.0, a.mom_out[9 + 2 * threadIdx.x] = 0.0;
  }
}
template <class P>
void launch_build_rows(const BuildRows& a, hipStream_t s) {
  if (a.M <= 0) return;
  constexpr int EPC = 16 / P::ESIZE;
  const int64_t total = a.M * ((a.inA != nullptr ? a.KpA / EPC : 0) + (a.inC != nullptr ? a.KpC / EPC : 0));
  int64_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL((build_rows_kernel<P>), dim3((unsigned)blocks), dim3(256), 0, s, a);
}
template void launch_build_rows<F32>(const BuildRows&, hipStream_t);
template void launch_build_rows<BF16>(const BuildRows&, hipStream_t);

template <class P>
__global__ void build_direct_kernel(const float* x, const int64_t* t, const float* state, const float* temb, int AF,
                                    int td, int cond, int64_t M, typename P::elem_t* in, int Kp) {
  const int64_t n = blockIdx.x;
  if (n >= M) return;
  const int tt = t ? (int)t[n] : 0;
  for (int c = threadIdx.x; c < Kp; c += blockDim.x) {
    float v = 0.f;
    if (c < AF)
      v = x[(size_t)n * AF + c];
    else if (c < AF + td)
      v = temb[(size_t)tt * td + (c - AF)];
    else if (c < AF + td + cond)
      v = state[(size_t)n * cond + (c - AF - td)];
    in[(size_t)n * Kp + c] = P::from_f32(v);
  }
}
template <class P>
void launch_build_direct(const float* x, const int64_t* t, const float* state, const float* temb, int AF, int td,
                         int cond, int64_t M, void* in, int Kp, hipStream_t s) {
  if (M <= 0) return;
  hipLaunchKernelGGL((build_direct_kernel<P>), dim3((unsigned)M), dim3(64), 0, s, x, t, state, temb, AF, td, cond, M,
                     (typename P::elem_t*)in, Kp);
}
template void launch_build_direct<F32>(const float*, const int64_t*, const float*, const float*, int, int, int, int64_t,
                                       void*, int, hipStream_t);
template void launch_build_direct<BF16>(const float*, const int64_t*, const float*, const float*, int, int, int,
                                        int64_t, void*, int, hipStream_t);

__global__ void copy_cols_kernel(const float* src, int ld, int col0, int ncols, int64_t rows, float* out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < (size_t)rows * ncols) out[i] = src[(i / ncols) * ld + col0 + i % ncols];
}
void launch_copy_cols(const float* src, int ld, int col0, int ncols, int64_t rows, float* out, hipStream_t s) {
  const size_t n = (size_t)rows * ncols;
  if (n == 0) return;
  hipLaunchKernelGGL(copy_cols_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, ld, col0, ncols, rows, out);
}

template <class P>
__global__ void zero_cols_kernel(typename P::elem_t* X, int M, int c0, int c1, int ld) {
  const int w = c1 - c0;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)M * w) return;
  X[(i / w) * ld + c0 + (i % w)] = P::from_f32(0.f);
}
template <class P>
void launch_zero_cols(void* X, int M, int c0, int c1, int ld, hipStream_t s) {
  const size_t n = (size_t)M * (c1 - c0);
  if (n == 0) return;
  hipLaunchKernelGGL((zero_cols_kernel<P>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (typename P::elem_t*)X, M,
                     c0, c1, ld);
}
template void launch_zero_cols<F32>(void*, int, int, int, int, hipStream_t);
template void launch_zero_cols<BF16>(void*, int, int, int, int, hipStream_t);

#define DPPO_LOG_SQRT_2PI 0.91893853320467274178f

__global__ void logprob_kernel(const LogprobArgs a) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.M * a.AF) return;
  const int64_t n = i / a.AF;
  const int j = (int)(i - n * a.AF);
  const int64_t b = n / a.Kft;
  const int k = (int)(n - b * a.Kft);
  const dppo_step st = a.ksteps[k];
  const float* ch = a.chains + ((size_t)b * (a.Kft + 1) + k) * a.AF;
  const float x = ch[j], xn = ch[a.AF + j];
  float mu, dmu;
  posterior(a.cfg, st, x, a.eps[(size_t)n * a.lde + j], mu, dmu);
  const float var = st.std * st.std;
  const float d = xn - mu;
  a.logp[i] = -(d * d) / (2.f * var) - logf(st.std) - DPPO_LOG_SQRT_2PI;
}
void launch_logprob(const LogprobArgs& a, hipStream_t s) {
  const int64_t n = a.M * a.AF;
  if (n <= 0) return;
  hipLaunchKernelGGL(logprob_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
}

// =================================================================================================
// behaviour-cloning term
// =================================================================================================
__device__ __forceinline__ double block_sum(double v, double* sh);
// Deterministic grid sum: every block stores its part, the block that arrives last adds all of them in block order.
// partial: [gridDim.x] doubles followed by an 8-byte arrival counter (zero on entry).  Call from all 256 threads.
__device__ __forceinline__ void finish_loss_sum(double part, double* partial, double* out, double* sh) {
  __shared__ bool last;
  if (threadIdx.x == 0) {
    partial[blockIdx.x] = part;
    __threadfence();
    last = atomicAdd((unsigned long long*)(partial + gridDim.x), 1ull) == gridDim.x - 1;
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  double t = 0.0;
  for (unsigned b = threadIdx.x; b < gridDim.x; b += 256) t += __builtin_nontemporal_load(&partial[b]);
  t = block_sum(t, sh);
  if (threadIdx.x == 0) *out = t;
}

// one thread per (row, padded column); the row's d_eps is written whole (zero padding included)
template <class P>
__global__ __launch_bounds__(256) void bc_loss_kernel(const BcArgs a) {
  __shared__ double sh[4];
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  double part = 0.0;
  if (i < a.M * a.ldde) {
    const int64_t n = i / a.ldde;
    const int j = (int)(i - n * a.ldde);
    float g = 0.f;
    if (j < a.AF) {
      const int64_t b = n / a.Kft;
      const int k = (int)(n - b * a.Kft);
      const dppo_step st = a.ksteps[k];
      const float* ch = a.chains + ((size_t)b * (a.Kft + 1) + k) * a.AF;
      float mu, dmu;
      posterior(a.cfg, st, ch[j], a.eps[(size_t)n * a.lde + j], mu, dmu);
      const float var = st.std * st.std;
      const float d = ch[a.AF + j] - mu;
      const float lp = -(d * d) / (2.f * var) - logf(st.std) - DPPO_LOG_SQRT_2PI;
      const float scale = 1.f / ((float)a.M * (float)a.AF);
      part = -(double)fminf(fmaxf(lp, -5.f), 2.f) * (double)scale;
      if (lp >= -5.f && lp <= 2.f) g = -scale * (d / var) * dmu;  // d(-lp)/d eps = -(d / var) * d mu / d eps
    }
    ((typename P::elem_t*)a.d_eps)[i] = P::from_f32(g);
  }
  part = block_sum(part, sh);
  finish_loss_sum(part, a.partial, a.loss, sh);
}
// see common.h: zeroing is a kernel launch, never a memset node
__global__ __launch_bounds__(256) void zero_bytes_kernel(uint32_t* p, size_t n4) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) p[i] = 0u;
}
void launch_zero_bytes(void* p, size_t bytes, hipStream_t s) {
  const size_t n4 = bytes / 4;
  if (n4 == 0) return;
  const size_t blocks = (n4 + 255) / 256;
  hipLaunchKernelGGL(zero_bytes_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, s, (uint32_t*)p, n4);
}
int64_t bc_loss_blocks(int64_t M, int ldde) { return (M * ldde + 255) / 256; }
template <class P>
void launch_bc_loss(const BcArgs& a, hipStream_t s) {
  const int64_t n = a.M * a.ldde;
  if (n <= 0) return;
  const int64_t blocks = bc_loss_blocks(a.M, a.ldde);
  launch_zero_bytes(a.partial + blocks, 8, s);  // the arrival counter
  hipLaunchKernelGGL((bc_loss_kernel<P>), dim3((unsigned)blocks), dim3(256), 0, s, a);
}
template void launch_bc_loss<F32>(const BcArgs&, hipStream_t);
template void launch_bc_loss<BF16>(const BcArgs&, hipStream_t);

template <class P>
__global__ __launch_bounds__(256) void mse_loss_kernel(const MseArgs a) {
  __shared__ double sh[4];
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  double part = 0.0;
  if (i < a.M * a.ldde) {
    const int64_t n = i / a.ldde;
    const int j = (int)(i - n * a.ldde);
    float g = 0.f;
    if (j < a.AF) {
      const float d = a.eps[(size_t)n * a.lde + j] - a.pairs[((size_t)n * 2 + 1) * a.AF + j];
      const float scale = 1.f / ((float)a.M * (float)a.AF);
      part = (double)(d * d) * (double)scale;
      g = 2.f * d * scale;
    }
    ((typename P::elem_t*)a.d_eps)[i] = P::from_f32(g);
  }
  part = block_sum(part, sh);
  finish_loss_sum(part, a.partial, a.loss, sh);
}
template <class P>
void launch_mse_loss(const MseArgs& a, hipStream_t s) {
  const int64_t n = a.M * a.ldde;
  if (n <= 0) return;
  const int64_t blocks = bc_loss_blocks(a.M, a.ldde);
  launch_zero_bytes(a.partial + blocks, 8, s);
  hipLaunchKernelGGL((mse_loss_kernel<P>), dim3((unsigned)blocks), dim3(256), 0, s, a);
}
template void launch_mse_loss<F32>(const MseArgs&, hipStream_t);
template void launch_mse_loss<BF16>(const MseArgs&, hipStream_t);

__global__ void axpy_kernel(float* y, const float* x, float alpha, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] += alpha * x[i];
}
void launch_axpy(float* y, const float* x, float alpha, int64_t n, hipStream_t s) {
  if (n > 0) hipLaunchKernelGGL(axpy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, y, x, alpha, n);
}

// =================================================================================================
// fused PPO loss
// =================================================================================================
__device__ __forceinline__ double block_sum(double v, double* sh) {
  // 256 threads: wave shuffle then 4 partials through LDS
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// moments[0..2] = (sum, sum of squares, N) of the minibatch's advantages.  Each block leaves its partial sums in
// moments[8 + 2 * block ..]; the block that finishes last (counter in moments[3], zeroed by the row builder) adds them
// in block order: no floating-point atomics, so the result is the same in every run (an atomic sum changed the last
// bit of the std from run to run)
__global__ __launch_bounds__(256) void adv_moments_kernel(const float* adv_k, const int32_t* brow, int64_t N,
                                                          double* moments) {
  __shared__ double sh[4];
  __shared__ bool last;
  double s = 0, q = 0;
  for (int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x; n < N; n += (int64_t)gridDim.x * 256) {
    const double v = adv_k[brow[n]];
    s += v;
    q += v * v;
  }
  s = block_sum(s, sh);
  q = block_sum(q, sh);
  if (threadIdx.x == 0) {
    // write-through stores, drained, then ONE relaxed agent-scope add; the last block reads the partials with sc1 loads behind
    // its barrier: no __threadfence() on either side (~3.5 us each on gfx950; this launch sits in front of the actor's forward)
    __hip_atomic_store(&moments[8 + 2 * blockIdx.x], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&moments[9 + 2 * blockIdx.x], q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // (DPPO_HANDOVER_*: common.h -- the fence-free form is an ISA property of gfx942 / gfx950, other targets get release / acquire)
    DPPO_HANDOVER_DRAIN();
    last = __hip_atomic_fetch_add((unsigned long long*)&moments[3], 1ull, DPPO_HANDOVER_ARRIVE_ORDER, __HIP_MEMORY_SCOPE_AGENT) ==
           gridDim.x - 1;
  }
  __syncthreads();
  if (!last) return;
  DPPO_HANDOVER_ACQUIRE();
  s = threadIdx.x < gridDim.x ? __hip_atomic_load(&moments[8 + 2 * threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
  q = threadIdx.x < gridDim.x ? __hip_atomic_load(&moments[9 + 2 * threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
  s = block_sum(s, sh);
  q = block_sum(q, sh);
  if (threadIdx.x == 0) moments[0] = s, moments[1] = q, moments[2] = (double)N;
}
void launch_adv_moments(const float* adv_k, const int32_t* brow, int64_t N, double* moments, hipStream_t s) {
  const int blocks = (int)min((int64_t)ADV_MOMENT_BLOCKS, (N + 255) / 256);
  hipLaunchKernelGGL(adv_moments_kernel, dim3(blocks), dim3(256), 0, s, adv_k, brow, N, moments);
}

// One lane per sample, 64 samples per block.  The first version gave a sample 16 lanes (one element each) and had all 16
// redo the per-sample scalar math (normalisation, exp, clip schedule, value loss, five double-precision statistics): a wave
// then finished 4 samples per pass and the kernel was VALU-bound at 34 us for 50,000 samples (profiles/r01_h_bench_serial_
// kernel_stats.csv) while moving 10 MB.  Here a lane walks its sample's elements itself (16-byte loads where the row pitch
// allows), the scalar math runs once per sample, and the elements' contribution to d loss / d eps stays in registers between
// the log-prob sum and the store (up to NREG elements; beyond that the second pass recomputes it from L1-resident inputs).
// Both log-prob sums run over j in ascending order, so unchanged weights still give ratio == 1 exactly.
constexpr int LOSS_THREADS = 64;

// (NREG = 64: action chunks of 33-64 elements, register cap lifted -- such cfgs have small minibatches, a few hundred one-wave
// blocks on 256 CUs, so occupancy is no concern.  Wider chunks (transport's 112, furniture's 80 elements) take NREG = 0, whose
// two passes walk the elements with 16-byte loads: as scalar loads they were 80 us for 10,000 samples.)
template <class P, int NREG>
__global__ __launch_bounds__(LOSS_THREADS, NREG <= 32 ? 8 : 1) void ppo_loss_kernel(const LossArgs a) {
  typedef typename P::elem_t E;
  constexpr int EPC = 16 / P::ESIZE;  // elements per 16-byte chunk of the outputs
  // per-k constants (only Kft distinct values exist): denoising discount and clip range, built once per block in
  // the reference's precision recipe (double pow / exp, then fp32) instead of per lane
  extern __shared__ float tab[];  // [Kft] discount, [Kft] eps_k, [2] adv mean / std
  const dppo_ppo_cfg& pc = a.pcfg;
  const int Kft = pc.ft_denoising_steps, AF = a.AF, Da = pc.action_dim;
  const bool pol = (a.part & 1) != 0, val = (a.part & 2) != 0;
  if (pol) {
    if (a.tab != nullptr) {
      for (int k = threadIdx.x; k < 2 * Kft; k += LOSS_THREADS) tab[k] = a.tab[k];
    } else {
      for (int k = threadIdx.x; k < Kft; k += LOSS_THREADS) loss_table_entry(pc, k, tab);
    }
    if (a.mom_blocks > 0) {  // the row builder's partial sums, one per lane, added by a fixed shuffle tree
      double s1 = (int)threadIdx.x < a.mom_blocks ? a.moments[8 + 2 * threadIdx.x] : 0.0;
      double s2 = (int)threadIdx.x < a.mom_blocks ? a.moments[9 + 2 * threadIdx.x] : 0.0;
      for (int o = 32; o > 0; o >>= 1) s1 += __shfl_down(s1, o), s2 += __shfl_down(s2, o);
      if (threadIdx.x == 0) {
        const double Nm = (double)a.N, mean = s1 / Nm;
        const double varu = (s2 - Nm * mean * mean) / (Nm - 1.0);  // unbiased (torch.std)
        tab[2 * Kft] = (float)mean;
        tab[2 * Kft + 1] = (float)sqrt(varu > 0 ? varu : 0);
        if (blockIdx.x == 0) a.moments_out[0] = s1, a.moments_out[1] = s2, a.moments_out[2] = Nm;  // for loss_finalize
      }
    } else if (threadIdx.x == 0) {
      const double Nm = a.moments[2], mean = a.moments[0] / Nm;
      const double varu = (a.moments[1] - Nm * mean * mean) / (Nm - 1.0);  // unbiased (torch.std)
      tab[2 * Kft] = (float)mean;
      tab[2 * Kft + 1] = (float)sqrt(varu > 0 ? varu : 0);
    }
    __syncthreads();
  }
  const int rh = pc.reward_horizon < pc.horizon_steps ? pc.reward_horizon : pc.horizon_steps;
  const int cnt = rh * Da;
  double s_pg = 0, s_v = 0, s_kl = 0, s_cf = 0, s_ratio = 0;
  const double Nn = a.n_count > 0 ? a.n_count : (a.mom_blocks > 0 ? (double)a.N : a.moments[2]);  // samples in the (global) minibatch
  const int64_t n = (int64_t)blockIdx.x * LOSS_THREADS + threadIdx.x;
  if (n < a.N) {
    const int b = a.brow[n];
    if (pol) {
      const int k = a.krow[n];
      const dppo_step st = a.ksteps[k];
      const float* ch = a.gathered ? a.chains + (size_t)b * 2 * AF : a.chains + ((size_t)b * (Kft + 1) + k) * AF;
      const float* olp = a.gathered ? a.logprobs_k + (size_t)b * AF : a.logprobs_k + ((size_t)b * Kft + k) * AF;
      const float* ep = a.eps + (size_t)n * a.lde;
      const float var = st.std * st.std, lstd = logf(st.std);
      // 16-byte loads when every row of the four tensors starts on a 16-byte boundary
      const bool vec = ((AF | a.lde | cnt) & 3) == 0;
      // ---- new / old log-probs, clamped to [-5, 2], averaged over the first `rh` chunk steps (:93-102)
      float sum_new = 0.f, sum_old = 0.f;
      float gsrc[NREG > 0 ? NREG : 1];  // (d / var) * d mu / d eps where the clamp passes the gradient, else 0
      auto element = [&](float x, float xn, float e, float o, float& gs) {
        float mu, dmu;
        posterior(a.dcfg, st, x, e, mu, dmu);
        const float d = xn - mu;
        const float lp = -(d * d) / (2.f * var) - lstd - DPPO_LOG_SQRT_2PI;
        sum_new += fminf(fmaxf(lp, -5.f), 2.f);
        sum_old += fminf(fmaxf(o, -5.f), 2.f);
        gs = (lp >= -5.f && lp <= 2.f) ? (d / var) * dmu : 0.f;
      };
      if constexpr (NREG > 0) {
        if (vec) {
#pragma unroll
          for (int j0 = 0; j0 < NREG; j0 += 4) {
            if (j0 < cnt) {
              const float4 x = *(const float4*)(ch + j0), xn = *(const float4*)(ch + AF + j0);
              const float4 e = *(const float4*)(ep + j0), o = *(const float4*)(olp + j0);
              element(x.x, xn.x, e.x, o.x, gsrc[j0]);
              element(x.y, xn.y, e.y, o.y, gsrc[j0 + 1]);
              element(x.z, xn.z, e.z, o.z, gsrc[j0 + 2]);
              element(x.w, xn.w, e.w, o.w, gsrc[j0 + 3]);
            }
          }
        } else {
#pragma unroll
          for (int j = 0; j < NREG; ++j)
            if (j < cnt) element(ch[j], ch[AF + j], ep[j], olp[j], gsrc[j]);
        }
      } else {
        float gs;
        if (vec) {  // (same element order: j ascending)
          for (int j0 = 0; j0 < cnt; j0 += 4) {
            const float4 x = *(const float4*)(ch + j0), xn = *(const float4*)(ch + AF + j0);
            const float4 e = *(const float4*)(ep + j0), o = *(const float4*)(olp + j0);
            element(x.x, xn.x, e.x, o.x, gs);
            element(x.y, xn.y, e.y, o.y, gs);
            element(x.z, xn.z, e.z, o.z, gs);
            element(x.w, xn.w, e.w, o.w, gs);
          }
        } else {
          for (int j = 0; j < cnt; ++j) element(ch[j], ch[AF + j], ep[j], olp[j], gs);
        }
      }
      const float newlp = sum_new / (float)cnt, oldlp = sum_old / (float)cnt;
      // ---- advantage: normalise over the minibatch, quantile clip, denoising discount (:129-144)
      float adv = a.adv_k[b];
      if (pc.norm_adv) adv = (adv - tab[2 * Kft]) / (tab[2 * Kft + 1] + 1e-8f);
      if (pc.has_adv_clip) adv = fminf(fmaxf(adv, pc.adv_clip_lo), pc.adv_clip_hi);
      adv *= tab[k];
      // ---- ratio, per-step clip range (:147-159)
      const float logratio = newlp - oldlp;
      const float ratio = expf(logratio);
      const float eps_k = tab[Kft + k];
      // ---- clipped surrogate (:170-174) and d L / d ratio with torch.max / clamp sub-gradients
      const float lo = 1.f - eps_k, hi = 1.f + eps_k;
      const float rc = fminf(fmaxf(ratio, lo), hi);
      const float pg1 = -adv * ratio, pg2 = -adv * rc;
      const float w1 = pg1 > pg2 ? 1.f : (pg1 == pg2 ? 0.5f : 0.f);
      const float within = (ratio >= lo && ratio <= hi) ? 1.f : 0.f;
      const float dL_dratio = -adv * (w1 + (1.f - w1) * within);
      const float coef = dL_dratio * ratio / ((float)Nn * (float)cnt);  // d mean(L) / d lp_j (before clamp mask)
      s_kl = (double)((ratio - 1.f) - logratio);
      s_cf = fabsf(ratio - 1.f) > eps_k ? 1.0 : 0.0;
      s_ratio = ratio;
      s_pg = fmaxf(pg1, pg2);
      // ---- d loss / d eps, zero padded to the GEMM K width: whole 16-byte chunks, then a scalar remainder
      E* de = (E*)a.d_eps + (size_t)n * a.ldde;
      auto grad_at = [&](int j) -> float {
        if (j >= cnt) return 0.f;
        if constexpr (NREG > 0) {
          return coef * gsrc[j];  // (j is a compile-time constant after unrolling: the array stays in registers)
        } else {
          float mu, dmu;
          posterior(a.dcfg, st, ch[j], ep[j], mu, dmu);
          const float d = ch[AF + j] - mu;
          const float lp = -(d * d) / (2.f * var) - lstd - DPPO_LOG_SQRT_2PI;
          return (lp >= -5.f && lp <= 2.f) ? coef * ((d / var) * dmu) : 0.f;
        }
      };
      auto pack_store = [&](E* dst, const float (&v)[EPC]) {
        u32x4 o;
        if constexpr (P::ESIZE == 4) {
          o = (u32x4){__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) o[q] = (uint32_t)f2bf(v[2 * q]) | ((uint32_t)f2bf(v[2 * q + 1]) << 16);
        }
        *(u32x4*)dst = o;
      };
      const int nch = a.ldde / EPC;
      int c = 0;
      if constexpr (NREG > 0) {
#pragma unroll
        for (int cc = 0; cc < NREG / EPC; ++cc) {
          if (cc < nch) {
            float v[EPC];
#pragma unroll
            for (int q = 0; q < EPC; ++q) v[q] = grad_at(cc * EPC + q);
            pack_store(de + cc * EPC, v);
          }
        }
        c = NREG / EPC < nch ? NREG / EPC : nch;
        const u32x4 z = (u32x4){0, 0, 0, 0};  // cnt <= NREG: everything behind the register chunks is padding
        for (; c < nch; ++c) *(u32x4*)(de + c * EPC) = z;
      } else {
        auto grad_of = [&](float x, float xn, float e) -> float {
          float mu, dmu;
          posterior(a.dcfg, st, x, e, mu, dmu);
          const float d = xn - mu;
          const float lp = -(d * d) / (2.f * var) - lstd - DPPO_LOG_SQRT_2PI;
          return (lp >= -5.f && lp <= 2.f) ? coef * ((d / var) * dmu) : 0.f;
        };
        for (; c < nch; ++c) {
          float v[EPC];
          if (vec && (c + 1) * EPC <= cnt) {  // a whole chunk of real elements: 16-byte loads (the same arithmetic as grad_at)
#pragma unroll
            for (int q = 0; q < EPC; q += 4) {
              const int j0 = c * EPC + q;
              const float4 x = *(const float4*)(ch + j0), xn = *(const float4*)(ch + AF + j0), e = *(const float4*)(ep + j0);
              v[q] = grad_of(x.x, xn.x, e.x), v[q + 1] = grad_of(x.y, xn.y, e.y);
              v[q + 2] = grad_of(x.z, xn.z, e.z), v[q + 3] = grad_of(x.w, xn.w, e.w);
            }
          } else {
#pragma unroll
            for (int q = 0; q < EPC; ++q) v[q] = grad_at(c * EPC + q);
          }
          pack_store(de + c * EPC, v);
        }
      }
      for (int j = nch * EPC; j < a.ldde; ++j) {
        float g = 0.f;
        if constexpr (NREG == 0) g = grad_at(j);  // (NREG > 0 is only launched with ldde a multiple of EPC)
        de[j] = P::from_f32(g);
      }
    }
    if (val) {
      // ---- value loss (:177-189)
      const float v = a.vnew[(size_t)n * a.ldv];
      const float ret = a.returns_k[b];
      float dv, lv;
      if (pc.has_vclip) {
        const float ov = a.values_k[b];
        const float c = (float)pc.clip_vloss_coef;
        const float dlt = v - ov;
        const float vc = ov + fminf(fmaxf(dlt, -c), c);
        const float lu = (v - ret) * (v - ret), lc = (vc - ret) * (vc - ret);
        lv = 0.5f * fmaxf(lu, lc);
        const float inr = (dlt >= -c && dlt <= c) ? 1.f : 0.f;
        const float wu = lu > lc ? 1.f : (lu == lc ? 0.5f : 0.f);
        dv = wu * (v - ret) + (1.f - wu) * (vc - ret) * inr;
      } else {
        lv = 0.5f * ((v - ret) * (v - ret));
        dv = v - ret;
      }
      s_v = lv;
      // ---- d loss / d v: column 0 of a zero-padded row
      E* dvp = (E*)a.d_v + (size_t)n * a.lddv;
      const int nch = a.lddv / EPC;
      u32x4 o = (u32x4){0, 0, 0, 0};
      if constexpr (P::ESIZE == 4)
        o[0] = __float_as_uint(dv / (float)Nn);
      else
        o[0] = (uint32_t)f2bf(dv / (float)Nn);
      if (nch > 0) *(u32x4*)dvp = o;
      const u32x4 z = (u32x4){0, 0, 0, 0};
      for (int c = 1; c < nch; ++c) *(u32x4*)(dvp + c * EPC) = z;
      for (int j = nch * EPC; j < a.lddv; ++j) dvp[j] = P::from_f32(j == 0 ? dv / (float)Nn : 0.f);
    }
  }
  // per-block partial sums; loss_finalize_kernel adds them in block order (no atomics: reproducible, and thousands of
  // double atomics on five addresses serialise in L2)
  double v5[5] = {s_pg, s_v, s_kl, s_cf, s_ratio};
#pragma unroll
  for (int q = 0; q < 5; ++q)
    for (int o = 32; o > 0; o >>= 1) v5[q] += __shfl_down(v5[q], o);
  if (threadIdx.x == 0) {
    double* o = a.partial + (size_t)blockIdx.x * 8;
    o[DPPO_STAT_PG_LOSS] = v5[0], o[DPPO_STAT_V_LOSS] = v5[1], o[DPPO_STAT_APPROX_KL] = v5[2];
    o[DPPO_STAT_CLIPFRAC] = v5[3], o[DPPO_STAT_RATIO] = v5[4];
  }
}

// one block of 1024 threads: the five statistics, summed over the loss kernel's blocks in a fixed order => reproducible
__device__ __forceinline__ void loss_finalize_block(const double* partial, int blocks, const double* moments, double* stats,
                                                    int part, double n_count) {
  // all five statistics in ONE pass and one barrier: this block may ride in the reduction launch on the update's critical
  // path (tail_reduce_kernel), where five passes with two barriers each were 8 us of latency
  __shared__ double shd[16][5];
  const double Nn = n_count > 0 ? n_count : moments[2];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  double v[5] = {0, 0, 0, 0, 0};
  for (int b = tid; b < blocks; b += 1024) {
#pragma unroll
    for (int k = 0; k < 5; ++k) v[k] += partial[(size_t)b * 8 + k];
  }
#pragma unroll
  for (int k = 0; k < 5; ++k)
    for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_down(v[k], o);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 5; ++k) shd[w][k] = v[k];
  }
  __syncthreads();
  if (tid < 5 && ((tid == DPPO_STAT_V_LOSS ? 2 : 1) & part)) {  // the other half's launch owns the other entries
    double t = 0;
    for (int i = 0; i < 16; ++i) t += shd[i][tid];
    stats[tid] = t / Nn;  // each entry has exactly one owner launch
  }
  if (tid == 64 && (part & 1)) {  // the policy half owns these (its stream is the one the moments were pooled on)
    const double mean = moments[0] / Nn;
    const double varu = Nn > 1 ? (moments[1] - Nn * mean * mean) / (Nn - 1.0) : 0.0;
    stats[DPPO_STAT_ADV_MEAN] = mean;
    stats[DPPO_STAT_ADV_STD] = sqrt(varu > 0 ? varu : 0);
  }
}
__global__ __launch_bounds__(1024) void loss_finalize_kernel(const double* partial, int blocks, const double* moments,
                                                             double* stats, int part, double n_count) {
  loss_finalize_block(partial, blocks, moments, stats, part, n_count);
}

int loss_blocks(int64_t N) { return (int)((N + LOSS_THREADS - 1) / LOSS_THREADS); }

template <class P>
void launch_ppo_loss(const LossArgs& a, hipStream_t s) {
  if (a.N <= 0) return;
  const size_t lds = (size_t)(2 * a.pcfg.ft_denoising_steps + 2) * sizeof(float);
  const int blocks = loss_blocks(a.N);
  const int rh = a.pcfg.reward_horizon < a.pcfg.horizon_steps ? a.pcfg.reward_horizon : a.pcfg.horizon_steps;
  const int cnt = rh * a.pcfg.action_dim;
  constexpr int EPC = 16 / P::ESIZE;
  const bool regs = (a.part & 1) && a.ldde % EPC == 0;  // the register variants store whole chunks
  if (regs && cnt <= 16)
    hipLaunchKernelGGL((ppo_loss_kernel<P, 16>), dim3(blocks), dim3(LOSS_THREADS), lds, s, a);
  else if (regs && cnt <= 32)
    hipLaunchKernelGGL((ppo_loss_kernel<P, 32>), dim3(blocks), dim3(LOSS_THREADS), lds, s, a);
  else if (regs && cnt <= 64)
    hipLaunchKernelGGL((ppo_loss_kernel<P, 64>), dim3(blocks), dim3(LOSS_THREADS), lds, s, a);
  else
    hipLaunchKernelGGL((ppo_loss_kernel<P, 0>), dim3(blocks), dim3(LOSS_THREADS), lds, s, a);
}
// the statistics (and nothing the backward pass reads): any stream ordered after the loss kernel will do
void launch_loss_finalize(const LossArgs& a, hipStream_t s) {
  if (a.N <= 0) return;
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(1024), 0, s, a.partial, loss_blocks(a.N), a.moments, a.stats,
                     a.part, a.n_count);
}
template void launch_ppo_loss<F32>(const LossArgs&, hipStream_t);
template void launch_ppo_loss<BF16>(const LossArgs&, hipStream_t);

// =================================================================================================
// time-embedding backward
// =================================================================================================
// grid (blocks, Kft); 256 threads = 16 row-lanes x 16 column-lanes
__global__ __launch_bounds__(256) void temb_segsum_kernel(const float* dtemb, int ld, const int32_t* krow, int64_t M,
                                                          int Kft, int td, float* partial) {
  __shared__ float red[16][17];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int k = blockIdx.y;
  const int64_t per = (M + gridDim.x - 1) / gridDim.x;
  const int64_t m0 = (int64_t)blockIdx.x * per, m1 = m0 + per < M ? m0 + per : M;
  for (int c0 = 0; c0 < td; c0 += 16) {
    const int c = c0 + cl;
    float s = 0.f;
    if (c < td)
      for (int64_t m = m0 + rl; m < m1; m += 16)
        if (krow[m] == k) s += dtemb[(size_t)m * ld + c];
    red[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && c < td) {
      float t = 0.f;
      for (int i = 0; i < 16; ++i) t += red[i][cl];
      partial[((size_t)blockIdx.x * Kft + k) * td + c] = t;
    }
    __syncthreads();
  }
}
void launch_temb_segsum(const float* dtemb, int ld, const int32_t* krow, int64_t M, int Kft, int td, float* partial,
                        int blocks, hipStream_t s) {
  hipLaunchKernelGGL(temb_segsum_kernel, dim3(blocks, Kft), dim3(256), 0, s, dtemb, ld, krow, M, Kft, td, partial);
}

static size_t time_backward_lds(int Kft, int td) {  // see time_backward_block
  return (size_t)(Kft * 7 * td + 4 * td * td + 2 * td + Kft * td) * sizeof(float);
}
size_t time_backward_lds_bytes(int Kft, int td) { return time_backward_lds(Kft, td); }
template <class K>
static void raise_dyn_lds(K kern) {  // above 64 KB of dynamic LDS a kernel needs its cap raised (once); the cap counts
  // static __shared__ too (post_reduce_kernel has a few bytes): ask for a little less than the 160 KB a CU has
  if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) != hipSuccess)
    (void)hipGetLastError();  // not fatal: launches below 64 KB do not need it
}
__device__ __forceinline__ void time_backward_block(const float* w1, const float* b1, const float* w2, const float* G_in,
                                                    const dppo_step* ksteps, int Kft, int td, float* gw1, float* gb1,
                                                    float* gw2, float* gb2, float* sh) {
  time_backward_prepare(w1, b1, w2, ksteps, Kft, td, sh);
  time_backward_finish(G_in, Kft, td, gw1, gb1, gw2, gb2, sh);
}
__global__ __launch_bounds__(256) void time_backward_kernel(const float* w1, const float* b1, const float* w2,
                                                            const float* G_in, const dppo_step* ksteps, int Kft, int td,
                                                            float* gw1, float* gb1, float* gw2, float* gb2) {
  extern __shared__ float sh[];
  time_backward_block(w1, b1, w2, G_in, ksteps, Kft, td, gw1, gb1, gw2, gb2, sh);
}

// Everything that follows the slab reduction of a backward pass, in ONE launch: the low-rank dW2 (lowrank_dw_kernel's
// body), G = W0_temb^T . S (temb_from_sums_kernel's, one wave per output) and, in the block that finishes G last, the
// time MLP's backward -- three dependent-looking launches of 5-17 us that only shared the slab reduction as an input.
// dWout[o][h] of a merged-top network (see PostReduce::U) and db2[h] of a one-block backward (PostReduce::db2): one wave
// per output, lanes over the contraction
__device__ __forceinline__ void wout_grad_block(const PostReduce& q, int b) {
  const int lane = threadIdx.x & 63;
  int out = b * 4 + (threadIdx.x >> 6);
  const int n1 = q.U != nullptr ? q.out_dim * q.H : 0;
  if (out < n1) {
    const int o = out / q.H, h = out - o * q.H;
    float acc = 0.f;
    for (int j = lane; j < q.H; j += 64) acc += post_in(q, q.T + (size_t)o * q.H + j) * q.W2[(size_t)h * q.H + j];
    for (int c = lane; c < q.in_dim; c += 64) acc += post_in(q, q.U + (size_t)o * q.ldu + c) * q.W0[(size_t)h * q.ldw0 + c];
    for (int d = 32; d > 0; d >>= 1) acc += __shfl_down(acc, d);
    if (lane == 0) q.dWout[out] = acc + q.cs[o] * (q.b0[h] + q.b2[h]);
    return;
  }
  out -= n1;
  if (q.db2 != nullptr && out < q.H) {  // colsum(dh_1) = colsum(d_out) . Wout
    float acc = 0.f;
    for (int o = lane; o < q.out_dim; o += 64) acc += q.cs[o] * q.Wout_b[(size_t)o * q.H + out];
    for (int d = 32; d > 0; d >>= 1) acc += __shfl_down(acc, d);
    if (lane == 0) q.db2[out] = acc;
  }
}
// dW[i][j] = sum_o Wout[o][i] T[o][j] (the top block's second-layer weight gradient from T = d_out^T act(z1)): a block owns
// LOWRANK_RI rows i x 256 columns j; a thread loads T[o][j] once per o for its 8 rows (the first form, one output per thread,
// re-read T for every row: 470 MB from L2 and 40-60 us at H = 1024 with 28-112 outputs)
constexpr int LOWRANK_RI = 8;
static int lowrank_dw_blocks(int H) { return ((H + LOWRANK_RI - 1) / LOWRANK_RI) * ((H + 255) / 256); }
__device__ __forceinline__ void lowrank_dw_block(const float* Wout, const float* T, int out_dim, int H, float* dW, int b,
                                                 bool t_sc1 = false) {  // t_sc1: T was produced by other workgroups of this launch
  const int jb = (H + 255) / 256;
  const int i0 = (b / jb) * LOWRANK_RI, j = (b % jb) * 256 + threadIdx.x;
  if (j >= H) return;
  float acc[LOWRANK_RI];
#pragma unroll
  for (int u = 0; u < LOWRANK_RI; ++u) acc[u] = 0.f;
  for (int o = 0; o < out_dim; ++o) {
    const float t = t_sc1 ? __hip_atomic_load(T + (size_t)o * H + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : T[(size_t)o * H + j];
    const float* w = Wout + (size_t)o * H + i0;  // (wave-uniform: scalar loads)
#pragma unroll
    for (int u = 0; u < LOWRANK_RI; ++u) acc[u] += (i0 + u < H ? w[u] : 0.f) * t;
  }
#pragma unroll
  for (int u = 0; u < LOWRANK_RI; ++u)
    if (i0 + u < H) dW[(size_t)(i0 + u) * H + j] = acc[u];
}
static int wout_grad_blocks(const PostReduce& q) {
  return ((q.U != nullptr ? q.out_dim * q.H : 0) + (q.db2 != nullptr ? q.H : 0) + 3) / 4;
}
__global__ __launch_bounds__(256) void wout_grad_kernel(const PostReduce q) { wout_grad_block(q, blockIdx.x); }
void launch_wout_grad(const PostReduce& q, hipStream_t s) {
  const int blocks = wout_grad_blocks(q);
  if (blocks > 0) hipLaunchKernelGGL(wout_grad_kernel, dim3(blocks), dim3(256), 0, s, q);
}

__global__ __launch_bounds__(256) void post_reduce_kernel(const PostReduce q) {
  extern __shared__ float sh[];
  const int tid = threadIdx.x;
  if ((int)blockIdx.x >= q.n_lowrank + q.n_temb + q.n_wout) {
    dw0_temb_block(q, blockIdx.x - q.n_lowrank - q.n_temb - q.n_wout);
    return;
  }
  if ((int)blockIdx.x >= q.n_lowrank + q.n_temb) {
    wout_grad_block(q, blockIdx.x - q.n_lowrank - q.n_temb);
    return;
  }
  if ((int)blockIdx.x < q.n_lowrank) {
    lowrank_dw_block(q.Wout, q.T, q.out_dim, q.H, q.dW, blockIdx.x);
    return;
  }
  temb_g_block(q, blockIdx.x - q.n_lowrank, sh);
}
void launch_post_reduce(PostReduce& q, hipStream_t s) {
  q.n_lowrank = q.dW != nullptr ? lowrank_dw_blocks(q.H) : 0;
  q.n_temb = q.G != nullptr ? (q.Kft * q.td + 3) / 4 : 0;
  q.n_wout = wout_grad_blocks(q);
  q.n_dw0t = q.dW0t != nullptr && q.G != nullptr ? (q.H * q.td + 255) / 256 : 0;
  const int blocks = q.n_lowrank + q.n_temb + q.n_wout + q.n_dw0t;
  static DevLatch raised;
  if (raised.need()) raise_dyn_lds(post_reduce_kernel), raised.done();
  if (blocks > 0)
    hipLaunchKernelGGL(post_reduce_kernel, dim3(blocks), dim3(256), (q.G ? time_backward_lds(q.Kft, q.td) : sizeof(float)), s, q);
}
void launch_time_backward(const float* w1, const float* b1, const float* w2, const float* G, const dppo_step* ksteps,
                          int Kft, int td, float* gw1, float* gb1, float* gw2, float* gb2, hipStream_t s) {
  static DevLatch raised;
  if (raised.need()) raise_dyn_lds(time_backward_kernel), raised.done();
  hipLaunchKernelGGL(time_backward_kernel, dim3(1), dim3(256), time_backward_lds(Kft, td), s, w1, b1, w2, G,
                     ksteps, Kft, td, gw1, gb1, gw2, gb2);
}
// G[k][j] = sum_h W0[h*ldw0 + AF + j] * S[h*Kft + k]: one wave per output, lanes over h (a single block looping over h
// costs 512 dependent L2 latencies: measured 150 us on the critical path)
__global__ __launch_bounds__(64) void temb_from_sums_kernel(const float* S, const float* W0, int ldw0, int AF, int H, int Kft,
                                                            int td, float* G) {
  const int k = blockIdx.x / td, j = blockIdx.x % td;
  float acc = 0.f;
  for (int h = threadIdx.x; h < H; h += 64) acc += W0[(size_t)h * ldw0 + AF + j] * S[(size_t)h * Kft + k];
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
  if (threadIdx.x == 0) G[blockIdx.x] = acc;
}
void launch_time_backward_from_sums(const float* w1, const float* b1, const float* w2, const float* S, const float* W0,
                                    int ldw0, int AF, int H, float* G, const dppo_step* ksteps, int Kft, int td, float* gw1,
                                    float* gb1, float* gw2, float* gb2, hipStream_t s) {
  hipLaunchKernelGGL(temb_from_sums_kernel, dim3(Kft * td), dim3(64), 0, s, S, W0, ldw0, AF, H, Kft, td, G);
  launch_time_backward(w1, b1, w2, G, ksteps, Kft, td, gw1, gb1, gw2, gb2, s);
}

__global__ void slab_reduce_2d_kernel(const float* slab, int splits, int rows, int cols, int lds, float* out, int ldo,
                                      float scale, int transpose) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)rows * cols) return;
  const int r = (int)(i / cols), c = (int)(i % cols);
  float p[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int k = 0;
  for (; k + 8 <= splits; k += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) p[u] += slab[((size_t)(k + u) * rows + r) * lds + c];
  }
  for (; k < splits; ++k) p[k & 7] += slab[((size_t)k * rows + r) * lds + c];
  const float v = (((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]))) * scale;
  if (transpose)
    out[(size_t)c * ldo + r] = v;
  else
    out[(size_t)r * ldo + c] = v;
}
void launch_slab_reduce_2d(const float* slab, int splits, int rows, int cols, int lds, float* out, int ldo,
                           float scale, hipStream_t s, int transpose) {
  const size_t n = (size_t)rows * cols;
  if (n == 0) return;
  hipLaunchKernelGGL(slab_reduce_2d_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, slab, splits, rows,
                     cols, lds, out, ldo, scale, transpose);
}

template <bool WT = false>  // WT: results stored write-through (read by other workgroups of the same launch: tail_post_kernel)
__device__ __forceinline__ void slab_job_block(const SlabJob& J, int bx = -1, int nbx = 0) {  // bx >= 0: block bx of nbx
  if (J.wide) {  // (uniform over the block)
    slab_job_block_wide<WT>(J, bx, nbx);
    return;
  }
  const size_t b0 = bx >= 0 ? (size_t)bx : blockIdx.x, bn = bx >= 0 ? (size_t)nbx : gridDim.x;
  // (A 16-byte form of this loop -- a thread owning four consecutive columns, the same summation tree per element, 128 bytes
  // per lane in flight instead of 32 -- was SLOWER: 21.1 vs 17.0 us per launch on average, tools/tail_reduce_parts.sh.  The
  // slabs of one output element lie rows x lds x 4 bytes = 1 MB apart, so what bounds the loop is not the bytes in flight per
  // thread; a quarter of the threads with four times the bytes each just spreads the same requests over fewer CUs.)
  const size_t n = (size_t)J.rows * J.cols;
  for (size_t i = b0 * blockDim.x + threadIdx.x; i < n; i += bn * blockDim.x) {
    const int r = (int)(i / J.cols), c = (int)(i % J.cols);
    float p[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // same summation tree as slab_reduce_2d_kernel
    int k = 0;
    // (32 loads in flight per thread where there are that many slabs: the thin products are split 64 ways, and eight
    // batches of eight loads were eight memory latencies in a row -- the longest chain of the launch)
    for (; k + 32 <= J.splits; k += 32) {
      float t[32];
#pragma unroll
      for (int u = 0; u < 32; ++u) t[u] = J.slab[((size_t)(k + u) * J.rows + r) * J.lds + J.c0 + c];
#pragma unroll
      for (int u = 0; u < 32; ++u) p[u & 7] += t[u];  // (p[u] still receives k + u, k + u + 8, ... in this order)
    }
    for (; k + 8 <= J.splits; k += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) p[u] += J.slab[((size_t)(k + u) * J.rows + r) * J.lds + J.c0 + c];
    }
    for (; k < J.splits; ++k) p[k & 7] += J.slab[((size_t)k * J.rows + r) * J.lds + J.c0 + c];
    const float v = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
    float* dst = J.transpose ? J.out + (size_t)c * J.ldo + r : J.out + (size_t)r * J.ldo + c;
    if constexpr (WT)
      __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
      *dst = v;
  }
}
__global__ __launch_bounds__(256) void slab_reduce_batch_kernel(const SlabJobs jobs) { slab_job_block(jobs.j[blockIdx.y]); }

// The slab reductions of a backward pass's GEMMs AND what follows them (low-rank dW2, dWout, db2: post_reduce_kernel's parts that
// read only the two THIN products T and U) in one launch: blocks [0, n_first) reduce the thin products (write-through, one arrival
// each), the next q.n_lowrank + q.n_wout blocks wait for them (post_blocks.h) and run behind, the rest reduce the other slabs
// beside them.  One launch boundary less between the GEMMs and AdamW.
__global__ __launch_bounds__(256) void tail_post_kernel(const TailPost t) {
  int b = blockIdx.x;
  if (b < t.n_first) {  // producers: jobs [0, n_first_jobs)
    int j = 0;
    while (j + 1 < t.n_first_jobs && b >= t.job_blocks[j]) b -= t.job_blocks[j], ++j;
    slab_job_block<true>(t.jobs.j[j], b, t.job_blocks[j]);
    post_arrive(t.q.wait_cnt);
    return;
  }
  b -= t.n_first;
  if (b < t.q.n_lowrank + t.q.n_wout) {  // consumers
    post_wait(t.q);
    if (b < t.q.n_lowrank)
      lowrank_dw_block(t.q.Wout, t.q.T, t.q.out_dim, t.q.H, t.q.dW, b, true);
    else
      wout_grad_block(t.q, b - t.q.n_lowrank);
    return;
  }
  b -= t.q.n_lowrank + t.q.n_wout;
  int j = t.n_first_jobs;
  while (j + 1 < t.jobs.n && b >= t.job_blocks[j]) b -= t.job_blocks[j], ++j;
  if (j < t.jobs.n) slab_job_block<false>(t.jobs.j[j], b, t.job_blocks[j]);
}
void launch_tail_post(TailPost& t, hipStream_t s) {
  t.q.n_lowrank = t.q.dW != nullptr ? lowrank_dw_blocks(t.q.H) : 0;
  t.q.n_wout = wout_grad_blocks(t.q);
  t.q.n_temb = t.q.n_dw0t = 0;
  int total = 0;
  t.n_first = 0;
  for (int i = 0; i < t.jobs.n; ++i) {
    const size_t n = (size_t)t.jobs.j[i].rows * t.jobs.j[i].cols;
    t.job_blocks[i] = (int)((n + (t.jobs.j[i].wide ? 63 : 255)) / (t.jobs.j[i].wide ? 64 : 256));
    if (t.job_blocks[i] > 512) t.job_blocks[i] = 512;  // (grid-stride loops inside)
    if (i < t.n_first_jobs) t.n_first += t.job_blocks[i];
    total += t.job_blocks[i];
  }
  t.q.wait_need = t.n_first;
  total += t.q.n_lowrank + t.q.n_wout;
  if (total > 0) hipLaunchKernelGGL(tail_post_kernel, dim3(total), dim3(256), 0, s, t);
}

// Everything that only waits for a backward pass's weight-gradient GEMMs and data-gradient kernel, in ONE launch of
// 1024-thread blocks: the slab reductions (blockIdx.y = job), the bias gradients = per-tile column sums of the fused
// backward reduced over tiles (one y per slot, 64 columns per block: reduce_slots_kernel's body), and the loss statistics
// (one block).  They were three launches on two streams; the second stream's join was a barrier packet (~5 us) between the
// GEMMs and the reductions on the update's critical path.
__global__ __launch_bounds__(1024) void tail_reduce_kernel(const TailReduce t) {
  // y order: the loss statistics (one block, a chain of dependent loads), the slots (49 dependent loads per lane at C2),
  // then the slab jobs (8 independent loads per lane): workgroups are dispatched in index order, the long chains go first
  const int nfin = t.fin_stats != nullptr ? 1 : 0;
  const int y = blockIdx.y;
  if (y < nfin) {
    if (blockIdx.x == 0) loss_finalize_block(t.fin_partial, t.fin_blocks, t.fin_moments, t.fin_stats, t.fin_part, t.fin_n_count);
    return;
  }
  if (y < nfin + t.slots.n_slots) {
    // 16 columns x 64 tile-lanes per block: a lane adds tiles / 64 values in four independent chains (12 loads at C2's 782
    // tiles; with 64 columns x 16 tile-lanes it was 49, and this block was the longest dependent chain of the launch)
    __shared__ float red[64][17];
    const int slot = y - nfin, n = t.width;
    if ((int)blockIdx.x * 16 >= n) return;
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    const float* src = t.colsum + (size_t)slot * t.tiles * n;
    float p[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < n && t.slots.n[slot] > 0) {
      int r = rl;
      for (; r + 192 < t.tiles; r += 256) {
#pragma unroll
        for (int u = 0; u < 4; ++u) p[u] += src[(size_t)(r + 64 * u) * n + c];
      }
      for (; r < t.tiles; r += 64) p[0] += src[(size_t)r * n + c];
    }
    red[rl][cl] = (p[0] + p[1]) + (p[2] + p[3]);
    __syncthreads();
    if (rl < 4) {  // 64 partial sums per column: four lanes x 16, then three adds
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) s += red[rl * 16 + i][cl];
      red[rl][cl] = s;
    }
    __syncthreads();
    if (rl == 0 && c < t.slots.n[slot]) t.slots.out[slot][c] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
    return;
  }
  slab_job_block(t.jobs.j[y - nfin - t.slots.n_slots]);
}
void launch_tail_reduce(TailReduce& t, const LossArgs* fin, hipStream_t s) {
  size_t most = 0;
  for (int i = 0; i < t.jobs.n; ++i) {
    size_t n = (size_t)t.jobs.j[i].rows * t.jobs.j[i].cols;
    if (t.jobs.j[i].wide) n *= 16;  // (64 elements per 1024-thread block: slab_job_block_wide)
    most = n > most ? n : most;
  }
  const bool has_fin = fin != nullptr && fin->N > 0;
  t.fin_stats = nullptr;
  if (has_fin) {
    t.fin_partial = fin->partial, t.fin_blocks = loss_blocks(fin->N), t.fin_moments = fin->moments, t.fin_stats = fin->stats;
    t.fin_part = fin->part, t.fin_n_count = fin->n_count;
  }
  unsigned gx = (unsigned)((most + 1023) / 1024);
  const unsigned gs = t.slots.n_slots > 0 ? (unsigned)((t.width + 15) / 16) : 0;
  gx = gx > gs ? gx : gs;
  const unsigned gy = (unsigned)(t.jobs.n + t.slots.n_slots + (has_fin ? 1 : 0));
  if (gx == 0) gx = 1;
  if (gy > 0) hipLaunchKernelGGL(tail_reduce_kernel, dim3(gx, gy), dim3(1024), 0, s, t);
}
void launch_slab_reduce_batch(const SlabJobs& jobs, hipStream_t s) {
  if (jobs.n <= 0) return;
  size_t most = 0;
  for (int i = 0; i < jobs.n; ++i) {
    size_t n = (size_t)jobs.j[i].rows * jobs.j[i].cols;
    if (jobs.j[i].wide) n *= 4;  // (64 elements per 256-thread block)
    most = n > most ? n : most;
  }
  if (most == 0) return;
  hipLaunchKernelGGL(slab_reduce_batch_kernel, dim3((unsigned)((most + 255) / 256), jobs.n), dim3(256), 0, s, jobs);
}

// =================================================================================================
// GAE: one thread per env, float64 reverse scan (train_ppo_diffusion_agent.py:255-279)
// =================================================================================================
__global__ void gae_kernel(const double* reward, const float* values, const float* terminated,
                           const float* last_values, int S, int E, double gamma, double lam, double rconst,
                           double* adv64, double* ret64, float* adv32, float* ret32) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  double last = 0.0;
  double nextv = (double)last_values[e];
  for (int t = S - 1; t >= 0; --t) {
    const size_t i = (size_t)t * E + e;
    const double v = (double)values[i];
    const double nonterm = 1.0 - (double)terminated[i];
    const double delta = reward[i] * rconst + gamma * nextv * nonterm - v;
    last = delta + gamma * lam * nonterm * last;
    const double ret = last + v;
    if (adv64) adv64[i] = last;
    if (ret64) ret64[i] = ret;
    if (adv32) adv32[i] = (float)last;
    if (ret32) ret32[i] = (float)ret;
    nextv = v;
  }
}
void launch_gae(const double* reward, const float* values, const float* terminated, const float* last_values, int S,
                int E, double gamma, double lam, double rconst, double* adv64, double* ret64, float* adv32,
                float* ret32, hipStream_t s) {
  if (E <= 0) return;
  hipLaunchKernelGGL(gae_kernel, dim3((E + 63) / 64), dim3(64), 0, s, reward, values, terminated, last_values, S, E,
                     gamma, lam, rconst, adv64, ret64, adv32, ret32);
}

// =================================================================================================
// optimiser
// =================================================================================================
__global__ __launch_bounds__(256) void sq_norm_stage1(const float* g, int64_t n, double* scratch) {
  __shared__ double sh[4];
  double s = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const double v = g[i];
    s += v * v;
  }
  s = block_sum(s, sh);
  if (threadIdx.x == 0) scratch[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void sq_norm_stage2(const double* scratch, int nb, double* out) {
  __shared__ double sh[4];
  double s = 0;
  for (int i = threadIdx.x; i < nb; i += 256) s += scratch[i];
  s = block_sum(s, sh);
  if (threadIdx.x == 0) out[0] = s;
}
void launch_sq_norm(const float* g, int64_t n, double* scratch, double* out, hipStream_t s) {
  const int nb = (int)min((int64_t)1024, max((int64_t)1, (n + 4095) / 4096));
  hipLaunchKernelGGL(sq_norm_stage1, dim3(nb), dim3(256), 0, s, g, n, scratch);
  hipLaunchKernelGGL(sq_norm_stage2, dim3(1), dim3(256), 0, s, scratch, nb, out);
}

// torch.optim.AdamW single-tensor path: p *= 1 - lr*wd; m.lerp_(g, 1-b1); v = b2 v + (1-b2) g g;
// p -= step_size * m / (sqrt(v) / sqrt(bc2) + eps)
__global__ void adamw_kernel(float* p, const float* g, float* m, float* v, int64_t n, float lr_wd_mul,
                             float one_m_b1, float b2, float one_m_b2, float step_size, float bc2_sqrt, float eps,
                             const double* sq_norm, float max_norm) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float gi = g[i];
  if (sq_norm != nullptr) {
    const float total = (float)sqrt(sq_norm[0]);
    const float coef = fminf(max_norm / (total + 1e-6f), 1.0f);
    gi *= coef;
  }
  float pi = p[i] * lr_wd_mul;
  float mi = m[i];
  mi = mi + one_m_b1 * (gi - mi);
  float vi = v[i] * b2 + one_m_b2 * (gi * gi);
  const float denom = sqrtf(vi) / bc2_sqrt + eps;
  pi = pi - step_size * (mi / denom);
  p[i] = pi;
  m[i] = mi;
  v[i] = vi;
}
// The same with the step count and the learning rate read from device memory, so that a captured graph of the update
// replays correctly: step t = *step_dev + 1 (adamw_tick_kernel increments it after the launch), lr = *lr_dev.
__global__ __launch_bounds__(256) void adamw_dev_kernel(float* p, const float* g, float* m, float* v, int64_t n,
                                                        const int32_t* step_dev, const float* lr_dev, double beta1,
                                                        double beta2, float eps, double weight_decay, const double* sq_norm,
                                                        float max_norm) {
  __shared__ float c[4];
  if (threadIdx.x == 0) {
    const double t = (double)(step_dev[0] + 1), lr = (double)lr_dev[0];
    const double bc1 = 1.0 - pow(beta1, t), bc2 = 1.0 - pow(beta2, t);
    c[0] = (float)(1.0 - lr * weight_decay), c[1] = (float)(lr / bc1), c[2] = (float)sqrt(bc2);
  }
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float gi = g[i];
  if (sq_norm != nullptr) {
    const float total = (float)sqrt(sq_norm[0]);
    gi *= fminf(max_norm / (total + 1e-6f), 1.0f);
  }
  const float one_m_b1 = (float)(1.0 - beta1), b2 = (float)beta2, one_m_b2 = (float)(1.0 - beta2);
  float pi = p[i] * c[0];
  float mi = m[i];
  mi = mi + one_m_b1 * (gi - mi);
  const float vi = v[i] * b2 + one_m_b2 * (gi * gi);
  pi = pi - c[1] * (mi / (sqrtf(vi) / c[2] + eps));
  p[i] = pi, m[i] = mi, v[i] = vi;
}
__global__ void adamw_tick_kernel(int32_t* step_dev) { step_dev[0] += 1; }
void launch_adamw_dev(float* p, const float* g, float* m, float* v, int64_t n, int32_t* step_dev, const float* lr_dev,
                      double beta1, double beta2, float eps, double weight_decay, const double* sq_norm, float max_norm,
                      hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(adamw_dev_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, g, m, v, n, step_dev, lr_dev,
                     beta1, beta2, eps, weight_decay, sq_norm, max_norm);
  hipLaunchKernelGGL(adamw_tick_kernel, dim3(1), dim3(1), 0, s, step_dev);
}

// Wc[o][k] = sum_j Wout[o][j] * W2[j][k]  ([out_dim][H], fp32): the composite layer of the fused backward's top block.
// One block per (o, 64 columns k): sixteen waves split j, lanes are consecutive k (coalesced rows of W2); eight loads in
// flight per lane (a plain loop over j is one L2 latency per iteration: 128 of them cost 50 us on the optimiser tail).
template <int COMPOSE_OB>  // outputs per block: 4 for wide heads (operand reuse), 1 for narrow ones (more, shorter blocks)
__global__ __launch_bounds__(1024) void compose_wc_kernel(const ComposeJobs q) {
  __shared__ float red[16][64];
  const ComposeJob& jb = q.j[blockIdx.z];
  const float *Wout = jb.Wout, *W2 = jb.W2, *b2 = jb.b2, *bout = jb.bout;
  float *Wc = jb.Wc, *cbias = jb.cbias;
  const int H = jb.H, cols = (H + 63) / 64;
  const int cols0 = jb.W0 != nullptr ? (jb.Kp0s + 63) / 64 : 0;
  // blockIdx.y = a group of COMPOSE_OB outputs o: the 64-column slice of W2 (or W0) a block walks is multiplied into all of
  // them (one output per block re-read the slice for every o: 112 x at transport's head, 470 MB from L2, 46 us)
  const int o0 = blockIdx.y * COMPOSE_OB;
  if (o0 >= jb.out_dim || (int)blockIdx.x > cols + cols0) return;  // the grid is sized for the larger network
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if ((int)blockIdx.x == cols) {  // extra column: cbias[o] = bout[o] + Wout[o] . b2 (the merged out layer's constant)
    for (int oo = 0; oo < COMPOSE_OB && o0 + oo < jb.out_dim; ++oo) {  // and cbias2[o] = cbias[o] + Wout[o] . b0
      const int o = o0 + oo;
      float t = 0.f, t0 = 0.f;
      for (int j = threadIdx.x; j < H; j += 1024) {
        t += Wout[(size_t)o * H + j] * b2[j];
        if (jb.W0 != nullptr) t0 += Wout[(size_t)o * H + j] * jb.b0[j];
      }
      for (int d = 32; d > 0; d >>= 1) t += __shfl_down(t, d), t0 += __shfl_down(t0, d);
      __syncthreads();
      if (lane == 0) red[w][0] = t, red[w][1] = t0;
      __syncthreads();
      if (threadIdx.x == 0) {
        float u = bout[o], u0 = 0.f;
        for (int i = 0; i < 16; ++i) u += red[i][0], u0 += red[i][1];
        cbias[o] = u;
        if (jb.W0 != nullptr) jb.cbias2[o] = u + u0;
      }
    }
    return;
  }
  const bool first = (int)blockIdx.x > cols;  // W0c[o][k] = sum_j Wout[o][j] * W0[j][k] instead (W0 is [H][in_dim])
  const int k = (first ? (int)blockIdx.x - cols - 1 : (int)blockIdx.x) * 64 + lane;
  const float* Wr = first ? jb.W0 : W2;
  const int ldr = first ? jb.in_dim : H, kmax = first ? jb.in_dim : H;
  // same summation tree per output as before: eight interleaved partial sums over j = w, w + 16, ..., then the 16 waves
  float acc[COMPOSE_OB][8];
#pragma unroll
  for (int oo = 0; oo < COMPOSE_OB; ++oo)
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[oo][u] = 0.f;
  if (k < kmax) {
    int j = w;
    for (; j + 7 * 16 < H; j += 8 * 16) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float r = Wr[(size_t)(j + 16 * u) * ldr + k];
#pragma unroll
        for (int oo = 0; oo < COMPOSE_OB; ++oo)
          acc[oo][u] += Wout[(size_t)(o0 + oo < jb.out_dim ? o0 + oo : o0) * H + j + 16 * u] * r;
      }
    }
    for (; j < H; j += 16) {
      const float r = Wr[(size_t)j * ldr + k];
#pragma unroll
      for (int oo = 0; oo < COMPOSE_OB; ++oo) acc[oo][0] += Wout[(size_t)(o0 + oo < jb.out_dim ? o0 + oo : o0) * H + j] * r;
    }
  }
  for (int oo = 0; oo < COMPOSE_OB && o0 + oo < jb.out_dim; ++oo) {
    const int o = o0 + oo;
    __syncthreads();
    red[w][lane] = ((acc[oo][0] + acc[oo][1]) + (acc[oo][2] + acc[oo][3])) + ((acc[oo][4] + acc[oo][5]) + (acc[oo][6] + acc[oo][7]));
    __syncthreads();
    if (w == 0) {
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) t += red[i][lane];
      if (first) {
        if (k < jb.Kp0s) jb.W0c[(size_t)o * jb.Kp0s + k] = k < kmax ? t : 0.f;
      } else if (k < H) {
        Wc[(size_t)o * H + k] = t;
      }
    }
  }
}
void launch_compose(const ComposeJobs& q, hipStream_t s) {
  int gx = 0, rows = 0;
  for (int i = 0; i < q.n; ++i) {
    const int x = (q.j[i].H + 63) / 64 + 1 + (q.j[i].W0 != nullptr ? (q.j[i].Kp0s + 63) / 64 : 0);
    gx = x > gx ? x : gx;
    rows = q.j[i].out_dim > rows ? q.j[i].out_dim : rows;
  }
  if (q.n <= 0) return;
  if (rows > 32)
    hipLaunchKernelGGL(compose_wc_kernel<4>, dim3(gx, (rows + 3) / 4, q.n), dim3(1024), 0, s, q);
  else
    hipLaunchKernelGGL(compose_wc_kernel<1>, dim3(gx, rows, q.n), dim3(1024), 0, s, q);
}

// dW[i][j] = sum_o Wout[o][i] * T[o][j]: the top block's second-layer weight gradient from T = d_out^T . act(z1).
// d loss / d h_nb = d_out . Wout has rank <= out_dim, so dz2^T . a2 = Wout^T . (d_out^T . a2): a thin contraction over the
// batch plus this out_dim-deep product replace an H x H contraction over the batch, and dh_nb never goes to HBM.
__global__ __launch_bounds__(256) void lowrank_dw_kernel(const float* Wout, const float* T, int out_dim, int H, float* dW) {
  lowrank_dw_block(Wout, T, out_dim, H, dW, blockIdx.x);
}
void launch_lowrank_dw(const float* Wout, const float* T, int out_dim, int H, float* dW, hipStream_t s) {
  hipLaunchKernelGGL(lowrank_dw_kernel, dim3((unsigned)lowrank_dw_blocks(H)), dim3(256), 0, s, Wout, T, out_dim, H, dW);
}

// float64 statistics <-> (hi, lo) float32 pairs, so that they can ride in the fp32 gradient bucket of the data-parallel
// all-reduce without losing precision: one launch each way instead of a dozen elementwise torch kernels
__global__ void stats_split_kernel(const double* st, float* hi_lo, int n) {
  const int i = threadIdx.x;
  if (i >= n) return;
  const float hi = (float)st[i];
  hi_lo[i] = hi;
  hi_lo[n + i] = (float)(st[i] - (double)hi);
}
__global__ void stats_merge_kernel(const float* hi_lo, double* st, int n, int first_avg, int n_avg, double inv_world) {
  const int i = threadIdx.x;
  if (i >= n) return;
  double v = (double)hi_lo[i] + (double)hi_lo[n + i];
  if (i >= first_avg && i < first_avg + n_avg) v *= inv_world;  // global values every rank wrote, not partial sums
  st[i] = v;
}
void launch_stats_split(const double* st, float* hi_lo, int n, hipStream_t s) {
  hipLaunchKernelGGL(stats_split_kernel, dim3(1), dim3(64), 0, s, st, hi_lo, n);
}
void launch_stats_merge(const float* hi_lo, double* st, int n, int first_avg, int n_avg, double inv_world, hipStream_t s) {
  hipLaunchKernelGGL(stats_merge_kernel, dim3(1), dim3(64), 0, s, hi_lo, st, n, first_avg, n_avg, inv_world);
}

constexpr int ADAMW_MULTI_EPT = 8;  // elements per thread
__global__ __launch_bounds__(256) void adamw_multi_kernel(const AdamwSlots a) {
  int si = 0;
#pragma unroll
  for (int i = 1; i < 4; ++i)
    if (i < a.n && (int)blockIdx.x >= a.s[i].block0) si = i;
  const AdamwSlot& sl = a.s[si];
  __shared__ float c[4];
  // the block's elements are requested BEFORE the step constants are built (two double-precision pow() behind a global load,
  // then a barrier): issued behind the barrier, their latency came on top of that chain, on the update's critical path
  const int lb0 = blockIdx.x - sl.block0;
  constexpr int NV = ADAMW_MULTI_EPT / 4;
  float4 p4r[NV], m4r[NV], v4r[NV], g4r[NV];
  if (sl.vec4) {
    const int64_t n4 = sl.n >> 2;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int64_t q = (int64_t)lb0 * (256 * NV) + u * 256 + threadIdx.x;
      if (q < n4) p4r[u] = ((const float4*)sl.p)[q], m4r[u] = ((const float4*)sl.m)[q], v4r[u] = ((const float4*)sl.v)[q], g4r[u] = ((const float4*)sl.g)[q];
    }
  }
  // the two double-precision pow() calls are the longest dependency of a block: one wave each
  if (threadIdx.x == 0) {
    const double t = (double)(sl.step_dev[0] + 1), lr = (double)sl.lr_dev[0];
    const double bc1 = 1.0 - pow(sl.beta1, t);
    c[0] = (float)(1.0 - lr * sl.weight_decay), c[1] = (float)(lr / bc1);
  }
  if (threadIdx.x == 64) {
    const double t = (double)(sl.step_dev[0] + 1);
    c[2] = (float)sqrt(1.0 - pow(sl.beta2, t));
  }
  __syncthreads();
  float clip = 1.f;
  if (sl.sq_norm != nullptr) clip = fminf(sl.max_norm / ((float)sqrt(sl.sq_norm[0]) + 1e-6f), 1.0f);
  const float one_m_b1 = (float)(1.0 - sl.beta1), b2 = (float)sl.beta2, one_m_b2 = (float)(1.0 - sl.beta2);
  const bool clipping = sl.sq_norm != nullptr;
  auto upd = [&](float& pi, float gi, float& mi, float& vi) {
    if (clipping) gi *= clip;
    pi = pi * c[0];
    mi = mi + one_m_b1 * (gi - mi);
    vi = vi * b2 + one_m_b2 * (gi * gi);
    pi = pi - c[1] * (mi / (sqrtf(vi) / c[2] + sl.eps));
  };
  const int lb = blockIdx.x - sl.block0;
  if (sl.vec4) {  // 16-byte accesses: few, fat blocks (the arrival counter below is one contended address)
    const int64_t n4 = sl.n >> 2;
#pragma unroll
    for (int u = 0; u < ADAMW_MULTI_EPT / 4; ++u) {
      const int64_t q = (int64_t)lb * (256 * ADAMW_MULTI_EPT / 4) + u * 256 + threadIdx.x;
      if (q >= n4) break;
      float4 p4 = p4r[u], m4 = m4r[u], v4 = v4r[u];
      const float4 g4 = g4r[u];
      upd(p4.x, g4.x, m4.x, v4.x), upd(p4.y, g4.y, m4.y, v4.y), upd(p4.z, g4.z, m4.z, v4.z), upd(p4.w, g4.w, m4.w, v4.w);
      ((float4*)sl.p)[q] = p4, ((float4*)sl.m)[q] = m4, ((float4*)sl.v)[q] = v4;
    }
    if (lb == 0 && threadIdx.x < (sl.n & 3)) {  // the last n % 4 elements
      const int64_t i = (n4 << 2) + threadIdx.x;
      float pi = sl.p[i], mi = sl.m[i], vi = sl.v[i];
      upd(pi, sl.g[i], mi, vi);
      sl.p[i] = pi, sl.m[i] = mi, sl.v[i] = vi;
    }
  } else {
    const int64_t i0 = (int64_t)lb * (256 * ADAMW_MULTI_EPT) + threadIdx.x;
#pragma unroll
    for (int u = 0; u < ADAMW_MULTI_EPT; ++u) {
      const int64_t i = i0 + u * 256;
      if (i >= sl.n) break;
      float pi = sl.p[i], mi = sl.m[i], vi = sl.v[i];
      upd(pi, sl.g[i], mi, vi);
      sl.p[i] = pi, sl.m[i] = mi, sl.v[i] = vi;
    }
  }
  // every block of the slot has read step_dev[0] before it arrives here: the last one to arrive advances the count
  if (threadIdx.x == 0 && atomicAdd(&sl.step_dev[1], 1) == sl.blocks - 1) {
    sl.step_dev[0] += 1;
    sl.step_dev[1] = 0;
  }
}
void launch_adamw_multi(AdamwSlots& a, hipStream_t s) {
  int blocks = 0;
  for (int i = 0; i < a.n; ++i) {
    a.s[i].block0 = blocks;
    a.s[i].blocks = (int)((a.s[i].n + 256 * ADAMW_MULTI_EPT - 1) / (256 * ADAMW_MULTI_EPT));
    a.s[i].vec4 = (((uintptr_t)a.s[i].p | (uintptr_t)a.s[i].g | (uintptr_t)a.s[i].m | (uintptr_t)a.s[i].v) & 15) == 0;
    blocks += a.s[i].blocks;
  }
  if (blocks > 0) hipLaunchKernelGGL(adamw_multi_kernel, dim3(blocks), dim3(256), 0, s, a);
}

void launch_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr_wd_mul, float one_m_b1, float b2,
                  float one_m_b2, float step_size, float bc2_sqrt, float eps, const double* sq_norm, float max_norm,
                  hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, g, m, v, n, lr_wd_mul,
                     one_m_b1, b2, one_m_b2, step_size, bc2_sqrt, eps, sq_norm, max_norm);
}

}  // namespace dppo
