// See gaussian.h.  gfx950 only.  Compiled with -ffp-contract=off: the log-prob / surrogate arithmetic keeps the reference's
// op sequence (torch.distributions.Normal.log_prob: -(x - mu)^2 / (2 var) - log sigma - log sqrt(2 pi)).
#include "gaussian.h"

namespace dppo {

#define GAUSS_LOG_SQRT_2PI 0.91893853320467274178f
constexpr int GAUSS_SPB = 16;  // samples per 256-thread block (16 lanes per sample)

int gauss_blocks(int64_t N) { return (int)((N + GAUSS_SPB - 1) / GAUSS_SPB); }

// sigma of action dimension jd (mlp_gaussian.py:352-361: learned per-Da log-variance clamped to [min, max], or a constant;
// gaussian.py:74-76: deterministic => 1e-4 everywhere)
__device__ __forceinline__ float gauss_sigma(const dppo_gaussian_cfg& c, const float* logvar, int jd, float* inside) {
  *inside = 0.f;
  if (c.deterministic) return 1e-4f;
  if (c.std_mode == 1) {
    const float lv = logvar[jd];
    *inside = (lv >= c.logvar_min && lv <= c.logvar_max) ? 1.f : 0.f;  // torch.clamp passes the gradient on the closed interval
    return expf(0.5f * fminf(fmaxf(lv, c.logvar_min), c.logvar_max));
  }
  return c.fixed_std;
}

// ---- advantage moments over a gathered minibatch (float64, fixed order) --------------------------------------------
__global__ __launch_bounds__(256) void gauss_moments_kernel(const float* adv, int64_t N, double* scratch, int blocks) {
  __shared__ double sh[2][4];
  double s = 0, q = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (int64_t)blocks * 256) {
    const double v = adv[i];
    s += v, q += v * v;
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o), q += __shfl_down(q, o);
  if ((threadIdx.x & 63) == 0) sh[0][threadIdx.x >> 6] = s, sh[1][threadIdx.x >> 6] = q;
  __syncthreads();
  if (threadIdx.x == 0) {
    scratch[2 * blockIdx.x] = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]);
    scratch[2 * blockIdx.x + 1] = (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]);
  }
}
__global__ void gauss_moments_final_kernel(const double* scratch, int blocks, int64_t N, double* moments) {
  double s = 0, q = 0;
  for (int b = 0; b < blocks; ++b) s += scratch[2 * b], q += scratch[2 * b + 1];
  moments[0] = s, moments[1] = q, moments[2] = (double)N;
}
void launch_gauss_moments(const float* adv, int64_t N, double* moments, double* scratch, hipStream_t s) {
  const int blocks = (int)((N + 255) / 256 < 64 ? (N + 255) / 256 : 64);
  hipLaunchKernelGGL(gauss_moments_kernel, dim3(blocks), dim3(256), 0, s, adv, N, scratch, blocks);
  hipLaunchKernelGGL(gauss_moments_final_kernel, dim3(1), dim3(1), 0, s, scratch, blocks, N, moments);
}

// ---- sampling (gaussian.py:94-121): a = mu + sigma * clamp(z, +-randn_clip) -----------------------------------------
__global__ __launch_bounds__(256) void gauss_sample_kernel(const GaussArgs a) {
  const dppo_gaussian_cfg& c = a.cfg;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.N * a.AF) return;
  const int64_t n = i / a.AF;
  const int j = (int)(i - n * a.AF);
  const float m = a.mean_pre[n * a.ldm + j];
  const float mu = c.tanh_mean ? tanhf(m) : m;
  float inside;
  const float sg = gauss_sigma(c, a.logvar, j % c.action_dim, &inside);
  float z = a.noise != nullptr ? a.noise[i] : philox_normal((uint64_t)i, c.seed_lo, c.seed_hi);
  z = fminf(fmaxf(z, -c.randn_clip), c.randn_clip);
  // the reference samples mu + sigma z and clamps to [mu - c sigma, mu + c sigma]: the same number
  a.out_actions[i] = mu + sg * z;
  if (a.out_mean != nullptr) a.out_mean[i] = mu;
}
void launch_gauss_sample(const GaussArgs& a, hipStream_t s) {
  const int64_t tot = a.N * a.AF;
  hipLaunchKernelGGL(gauss_sample_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, a);
}

// ---- log-prob of given actions (gaussian_vpg.py:46-62): mean over Ta*Da of the element log-probs ----------------------
__global__ __launch_bounds__(256) void gauss_logprob_kernel(const GaussArgs a) {
  const dppo_gaussian_cfg& c = a.cfg;
  const int sub = threadIdx.x & 15;
  const int64_t n = (int64_t)blockIdx.x * GAUSS_SPB + (threadIdx.x >> 4);
  const int64_t nn = n < a.N ? n : a.N - 1;
  float sum = 0.f;
  for (int j = sub; j < a.AF; j += 16) {
    const float m = a.mean_pre[nn * a.ldm + j];
    const float mu = c.tanh_mean ? tanhf(m) : m;
    float inside;
    const float sg = gauss_sigma(c, a.logvar, j % c.action_dim, &inside);
    const float d = a.actions[nn * a.AF + j] - mu;
    sum += -(d * d) / (2.f * (sg * sg)) - logf(sg) - GAUSS_LOG_SQRT_2PI;
  }
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) sum += __shfl_xor(sum, o);
  if (n < a.N && sub == 0) a.out_logp[n] = sum / (float)a.AF;
}
void launch_gauss_logprob(const GaussArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(gauss_logprob_kernel, dim3(gauss_blocks(a.N)), dim3(256), 0, s, a);
}

// ---- PPO_Gaussian.loss (gaussian_ppo.py:39-128) and its gradient w.r.t. the trunk output, the value and logvar ---------
template <class P>
__global__ __launch_bounds__(256) void gauss_loss_kernel(const GaussArgs a) {
  typedef typename P::elem_t E;
  extern __shared__ float lds[];  // [GAUSS_SPB][AF] per-element d logp / d logvar contributions (std_mode 1)
  __shared__ float mom[2];
  const dppo_gaussian_cfg& c = a.cfg;
  const int AF = a.AF, Da = c.action_dim;
  if (threadIdx.x == 0) {
    const double Nm = a.moments[2], mean = a.moments[0] / Nm;
    const double varu = (a.moments[1] - Nm * mean * mean) / (Nm - 1.0);  // unbiased (torch.std)
    mom[0] = (float)mean, mom[1] = (float)sqrt(varu > 0 ? varu : 0);
  }
  __syncthreads();
  const double Nn = a.moments[2];  // samples in the (global) minibatch: every mean divides by it
  const int sub = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int64_t n = (int64_t)blockIdx.x * GAUSS_SPB + grp;
  const bool live = n < a.N;
  const int64_t nn = live ? n : a.N - 1;
  const float* mp = a.mean_pre + nn * a.ldm;
  const float* ac = a.actions + nn * AF;
  float sum = 0.f;
  for (int j = sub; j < AF; j += 16) {
    const float mu = c.tanh_mean ? tanhf(mp[j]) : mp[j];
    float inside;
    const float sg = gauss_sigma(c, a.logvar, j % Da, &inside);
    const float d = ac[j] - mu;
    sum += -(d * d) / (2.f * (sg * sg)) - logf(sg) - GAUSS_LOG_SQRT_2PI;
  }
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) sum += __shfl_xor(sum, o);
  const float logp = sum / (float)AF;
  const float newlp = fminf(fmaxf(logp, -5.f), 2.f), oldlp = fminf(fmaxf(a.oldlogp[nn], -5.f), 2.f);
  const float logratio = newlp - oldlp;
  const float ratio = expf(logratio);
  float adv = a.adv[nn];
  if (c.norm_adv) adv = (adv - mom[0]) / (mom[1] + 1e-8f);
  const float clip = (float)c.clip_ploss_coef;
  const float lo = 1.f - clip, hi = 1.f + clip;
  const float rc = fminf(fmaxf(ratio, lo), hi);
  const float pg1 = -adv * ratio, pg2 = -adv * rc;
  const float w1 = pg1 > pg2 ? 1.f : (pg1 == pg2 ? 0.5f : 0.f);  // torch.max splits ties 1/2 - 1/2
  const float within = (ratio >= lo && ratio <= hi) ? 1.f : 0.f;
  const float dL_dratio = -adv * (w1 + (1.f - w1) * within);
  const float pass = (logp >= -5.f && logp <= 2.f) ? 1.f : 0.f;
  const float coef = dL_dratio * ratio * pass / ((float)Nn * (float)AF);  // d mean(L) / d (element log-prob)
  // value loss
  const float v = a.vnew[nn * a.ldv], ret = a.returns[nn];
  float dv, lv;
  if (c.has_vclip) {
    const float ov = a.oldvalues[nn], cv = (float)c.clip_vloss_coef, dlt = v - ov;
    const float vc = ov + fminf(fmaxf(dlt, -cv), cv);
    const float lu = (v - ret) * (v - ret), lc = (vc - ret) * (vc - ret);
    lv = 0.5f * fmaxf(lu, lc);
    const float inr = (dlt >= -cv && dlt <= cv) ? 1.f : 0.f;
    const float wu = lu > lc ? 1.f : (lu == lc ? 0.5f : 0.f);
    dv = wu * (v - ret) + (1.f - wu) * (vc - ret) * inr;
  } else {
    lv = 0.5f * ((v - ret) * (v - ret));
    dv = v - ret;
  }
  // gradients: d_mean (padded to the GEMM K width with zeros), d_v, and the per-element logvar terms
  E* dm = (E*)a.d_mean + (size_t)nn * a.lddm;
  for (int j = sub; j < a.lddm; j += 16) {
    float g = 0.f, glv = 0.f;
    if (j < AF) {
      const float mu = c.tanh_mean ? tanhf(mp[j]) : mp[j];
      float inside;
      const float sg = gauss_sigma(c, a.logvar, j % Da, &inside);
      const float d = ac[j] - mu, var = sg * sg;
      g = coef * (d / var) * (c.tanh_mean ? 1.f - mu * mu : 1.f);
      // d lp / d logvar = 0.5 (d^2 / var - 1), through the clamp of logvar
      glv = coef * 0.5f * ((d * d) / var - 1.f) * inside;
      if (c.std_mode == 1) lds[grp * AF + j] = live ? glv : 0.f;
    }
    if (live) dm[j] = P::from_f32(g);
  }
  if (live) {
    E* dvp = (E*)a.d_v + (size_t)nn * a.lddv;
    for (int j = sub; j < a.lddv; j += 16) dvp[j] = P::from_f32(j == 0 ? dv / (float)Nn : 0.f);
  }
  __syncthreads();
  // per-block partials, fixed order: [pg, v, kl, clipfrac, ratio, -, -, -, logvar grad (Da)]
  __shared__ double red[GAUSS_SPB][5];
  if (sub == 0) {
    red[grp][0] = live ? (double)fmaxf(pg1, pg2) : 0.0;
    red[grp][1] = live ? (double)lv : 0.0;
    red[grp][2] = live ? (double)((ratio - 1.f) - logratio) : 0.0;
    red[grp][3] = live && fabsf(ratio - 1.f) > clip ? 1.0 : 0.0;
    red[grp][4] = live ? (double)ratio : 0.0;
  }
  __syncthreads();
  double* o = a.partial + (size_t)blockIdx.x * (8 + Da);
  if (threadIdx.x < 5) {
    double t = 0;
    for (int g2 = 0; g2 < GAUSS_SPB; ++g2) t += red[g2][threadIdx.x];
    o[threadIdx.x] = t;
  }
  if (c.std_mode == 1) {
    for (int jd = threadIdx.x; jd < Da; jd += 256) {
      double t = 0;
      for (int g2 = 0; g2 < GAUSS_SPB; ++g2)
        for (int j = jd; j < AF; j += Da) t += (double)lds[g2 * AF + j];
      o[8 + jd] = t;
    }
  }
}

// one block: statistics (means over the global count), entropy / mean std of the policy, logvar gradient
__global__ __launch_bounds__(256) void gauss_finalize_kernel(const GaussArgs a, int blocks) {
  __shared__ double sh[256];
  const dppo_gaussian_cfg& c = a.cfg;
  const int Da = c.action_dim, stride = 8 + Da;
  const double Nn = a.moments[2];
  for (int k = 0; k < 5 + (c.std_mode == 1 ? Da : 0); ++k) {
    const int col = k < 5 ? k : 8 + (k - 5);
    double s = 0;
    for (int b = threadIdx.x; b < blocks; b += 256) s += a.partial[(size_t)b * stride + col];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      if (k < 5)
        a.stats[k] = sh[0] / Nn;
      else
        a.logvar_grad[k - 5] = (float)sh[0];  // already scaled by 1 / (N AF) per element
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double mean = a.moments[0] / Nn;
    const double varu = Nn > 1 ? (a.moments[1] - Nn * mean * mean) / (Nn - 1.0) : 0.0;
    a.stats[DPPO_STAT_ADV_MEAN] = mean;
    a.stats[DPPO_STAT_ADV_STD] = sqrt(varu > 0 ? varu : 0);
    // dist.entropy().mean() = 0.5 + 0.5 log(2 pi) + mean_j log sigma_j ; dist.scale.mean() (gaussian_vpg.py:60-61)
    double ls = 0, sg_sum = 0;
    for (int jd = 0; jd < Da; ++jd) {
      float inside;
      const float sg = gauss_sigma(c, a.logvar, jd, &inside);
      ls += (double)logf(sg), sg_sum += (double)sg;
    }
    a.stats[7] = (0.5 + 0.5 * 1.8378770664093453) + ls / Da;  // entropy; slot 7 of the stats block
    a.stats[DPPO_STAT_COUNT + 0] = sg_sum / Da;               // mean std, one past the common block
  }
}

template <class P>
void launch_gauss_loss(const GaussArgs& a, hipStream_t s) {
  const int blocks = gauss_blocks(a.N);
  const size_t lds = a.cfg.std_mode == 1 ? (size_t)GAUSS_SPB * a.AF * sizeof(float) : sizeof(float);
  hipLaunchKernelGGL((gauss_loss_kernel<P>), dim3(blocks), dim3(256), lds, s, a);
  hipLaunchKernelGGL(gauss_finalize_kernel, dim3(1), dim3(256), 0, s, a, blocks);
}
template void launch_gauss_loss<F32>(const GaussArgs&, hipStream_t);
template void launch_gauss_loss<BF16>(const GaussArgs&, hipStream_t);

}  // namespace dppo
