// See gmm.h.  gfx950 only.  Compiled with -ffp-contract=off (the log-prob arithmetic keeps torch.distributions' op sequence).
// One wave per sample: the lanes stride over the num_modes * Ta*Da component elements, per-mode sums by wave reduction,
// then every lane holds the mixture: log p(a) = logsumexp_m(log pi_m + sum_j log N(a_j; mu_mj, sigma_mj)) (MixtureSameFamily over
// Independent(Normal, 1): gmm.py:66-90), responsibilities r_m, and the PPO scalars.
#include "gmm.h"

namespace dppo {

#define GMM_LOG_SQRT_2PI 0.91893853320467274178f
constexpr int GMM_SPB = 4;  // samples (waves) per 256-thread block

int gmm_blocks(int64_t N) { return (int)((N + GMM_SPB - 1) / GMM_SPB); }

// sigma of (mode m, action dimension d): learned log-variance of size num_modes * Da (mlp_gmm.py:56-63,92-96) clamped to
// [min, max], or a constant; deterministic => 1e-4 (gmm.py:57-59)
__device__ __forceinline__ float gmm_sigma(const dppo_gmm_cfg& c, const float* logvar, int m, int d, float* inside) {
  *inside = 0.f;
  if (c.deterministic) return 1e-4f;
  if (c.std_mode == 1) {
    const float lv = logvar[m * c.action_dim + d];
    *inside = (lv >= c.logvar_min && lv <= c.logvar_max) ? 1.f : 0.f;
    return expf(0.5f * fminf(fmaxf(lv, c.logvar_min), c.logvar_max));
  }
  return c.fixed_std;
}
__device__ __forceinline__ float wsum(float v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

struct GmmEval {
  float logpi[GMM_MAX_MODES], lp[GMM_MAX_MODES], r[GMM_MAX_MODES];
  float logp;
};
// every lane of the wave returns the same values
__device__ __forceinline__ void gmm_eval(const GmmArgs& a, int64_t n, int lane, GmmEval& ev) {
  const dppo_gmm_cfg& c = a.cfg;
  const int M = c.num_modes, AF = a.AF, Da = c.action_dim;
  float acc[GMM_MAX_MODES];
#pragma unroll
  for (int m = 0; m < GMM_MAX_MODES; ++m) acc[m] = 0.f;
  const float* mp = a.mean_pre + n * a.ldm;
  const float* ac = a.actions + n * AF;
  for (int e = lane; e < M * AF; e += 64) {
    const int m = e / AF, j = e - m * AF;
    float inside;
    const float sg = gmm_sigma(c, a.logvar, m, j % Da, &inside);
    const float d = ac[j] - tanhf(mp[e]);
    const float term = -(d * d) / (2.f * (sg * sg)) - logf(sg) - GMM_LOG_SQRT_2PI;
#pragma unroll
    for (int mm = 0; mm < GMM_MAX_MODES; ++mm) acc[mm] += mm == m ? term : 0.f;
  }
  float lmax = -INFINITY;
#pragma unroll
  for (int m = 0; m < GMM_MAX_MODES; ++m) {
    acc[m] = wsum(acc[m]);
    if (m < M) lmax = fmaxf(lmax, a.logits[n * a.ldl + m]);
  }
  float lse = 0.f;
  for (int m = 0; m < M; ++m) lse += expf(a.logits[n * a.ldl + m] - lmax);
  lse = lmax + logf(lse);
  float mx = -INFINITY;
  for (int m = 0; m < M; ++m) {
    ev.logpi[m] = a.logits[n * a.ldl + m] - lse;
    ev.lp[m] = ev.logpi[m] + acc[m];
    mx = fmaxf(mx, ev.lp[m]);
  }
  float s = 0.f;
  for (int m = 0; m < M; ++m) s += expf(ev.lp[m] - mx);
  ev.logp = mx + logf(s);
  for (int m = 0; m < M; ++m) ev.r[m] = expf(ev.lp[m] - ev.logp);
}

// ---- sampling (gmm.py:88-97): component k ~ Categorical(logits), a = mu_k + sigma_k z ---------------------------------------
__global__ __launch_bounds__(256) void gmm_sample_kernel(const GmmArgs a) {
  const dppo_gmm_cfg& c = a.cfg;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.N * a.AF) return;
  const int64_t n = i / a.AF;
  const int j = (int)(i - n * a.AF);
  int k;
  if (a.modes_in != nullptr) {
    k = (int)a.modes_in[n];
  } else {  // inverse CDF of softmax(logits) at a uniform draw keyed by the sample index (same value in all AF threads of a sample)
    const float g = philox_normal((uint64_t)(a.N * a.AF) + (uint64_t)n, c.seed_lo, c.seed_hi);
    const float u = 0.5f * erfcf(-g * 0.70710678118654752f);  // Phi(g): uniform on (0, 1)
    float lmax = -INFINITY, tot = 0.f;
    for (int m = 0; m < c.num_modes; ++m) lmax = fmaxf(lmax, a.logits[n * a.ldl + m]);
    for (int m = 0; m < c.num_modes; ++m) tot += expf(a.logits[n * a.ldl + m] - lmax);
    float cum = 0.f;
    k = c.num_modes - 1;
    for (int m = 0; m < c.num_modes; ++m) {
      cum += expf(a.logits[n * a.ldl + m] - lmax) / tot;
      if (u < cum) {
        k = m;
        break;
      }
    }
  }
  float inside;
  const float sg = gmm_sigma(c, a.logvar, k, j % c.action_dim, &inside);
  const float z = a.noise != nullptr ? a.noise[i] : philox_normal((uint64_t)i, c.seed_lo, c.seed_hi);
  a.out_actions[i] = tanhf(a.mean_pre[n * a.ldm + k * a.AF + j]) + sg * z;
}
void launch_gmm_sample(const GmmArgs& a, hipStream_t s) {
  const int64_t tot = a.N * a.AF;
  hipLaunchKernelGGL(gmm_sample_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, a);
}

__global__ __launch_bounds__(256) void gmm_logprob_kernel(const GmmArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t n = (int64_t)blockIdx.x * GMM_SPB + (threadIdx.x >> 6);
  if (n >= a.N) return;
  GmmEval ev;
  gmm_eval(a, n, lane, ev);
  if (lane == 0) a.out_logp[n] = ev.logp;
}
void launch_gmm_logprob(const GmmArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(gmm_logprob_kernel, dim3(gmm_blocks(a.N)), dim3(256), 0, s, a);
}

// ---- PPO_GMM.loss forward + d loss / d (mean_pre, logits, v, logvar) ---------------------------------------------------------
// The returned actor-side gradients are those of pg_loss + ent_coef * entropy_loss (the entropy term reaches the mixture logits
// and the log-variances: gmm.py:70-74), so that one backward of the two trunks serves the agent's total loss.
template <class P>
__global__ __launch_bounds__(256) void gmm_loss_kernel(const GmmArgs a) {
  typedef typename P::elem_t E;
  extern __shared__ float lds[];  // [GMM_SPB][num_modes * AF] per-element d loss / d logvar contributions (std_mode 1)
  __shared__ float mom[2];
  __shared__ double red[GMM_SPB][7];
  __shared__ float lvent[GMM_SPB][GMM_MAX_MODES];  // pi_m of each sample (entropy's logvar gradient)
  const dppo_gmm_cfg& c = a.cfg;
  const int M = c.num_modes, AF = a.AF, Da = c.action_dim, Ta = AF / Da;
  if (threadIdx.x == 0) {
    const double Nm = a.moments[2], mean = a.moments[0] / Nm;
    const double varu = (a.moments[1] - Nm * mean * mean) / (Nm - 1.0);
    mom[0] = (float)mean, mom[1] = (float)sqrt(varu > 0 ? varu : 0);
  }
  __syncthreads();
  const double Nn = a.moments[2];
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int64_t n = (int64_t)blockIdx.x * GMM_SPB + grp;
  const bool live = n < a.N;
  const int64_t nn = live ? n : a.N - 1;
  GmmEval ev;
  gmm_eval(a, nn, lane, ev);
  const float logp = ev.logp;
  const float newlp = fminf(fmaxf(logp, -5.f), 2.f), oldlp = fminf(fmaxf(a.oldlogp[nn], -5.f), 2.f);
  const float logratio = newlp - oldlp;
  const float ratio = expf(logratio);
  float adv = a.adv[nn];
  if (c.norm_adv) adv = (adv - mom[0]) / (mom[1] + 1e-8f);
  const float clip = (float)c.clip_ploss_coef;
  const float lo = 1.f - clip, hi = 1.f + clip;
  const float rc = fminf(fmaxf(ratio, lo), hi);
  const float pg1 = -adv * ratio, pg2 = -adv * rc;
  const float w1 = pg1 > pg2 ? 1.f : (pg1 == pg2 ? 0.5f : 0.f);
  const float within = (ratio >= lo && ratio <= hi) ? 1.f : 0.f;
  const float dL_dratio = -adv * (w1 + (1.f - w1) * within);
  const float pass = (logp >= -5.f && logp <= 2.f) ? 1.f : 0.f;
  const float coef = dL_dratio * ratio * pass / (float)Nn;  // d mean(L) / d log p(a)
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
  // entropy of the mixture as the reference approximates it: sum_m pi_m H_m, H_m = sum_j (0.5 + 0.5 log 2 pi + log sigma_mj)
  float H[GMM_MAX_MODES], sbar[GMM_MAX_MODES], Hbar = 0.f, std_s = 0.f;
  for (int m = 0; m < M; ++m) {
    float h = 0.f, sm = 0.f;
    for (int d = 0; d < Da; ++d) {
      float inside;
      const float sg = gmm_sigma(c, a.logvar, m, d, &inside);
      h += logf(sg), sm += sg;
    }
    H[m] = (float)AF * (0.5f + GMM_LOG_SQRT_2PI) + (float)Ta * h;
    sbar[m] = sm / (float)Da;
    const float pi = expf(ev.logpi[m]);
    Hbar += pi * H[m], std_s += pi * sbar[m];
  }
  const float ec = c.ent_coef / (float)Nn;  // d (ent_coef * -mean entropy) / d (per-sample entropy) = -ec
  // ---- gradients
  const float* mp = a.mean_pre + nn * a.ldm;
  const float* ac = a.actions + nn * AF;
  E* dm = (E*)a.d_mean + (size_t)nn * a.lddm;
  for (int e = lane; e < a.lddm; e += 64) {
    float g = 0.f, glv = 0.f;
    if (e < M * AF) {
      const int m = e / AF, j = e - m * AF;
      float inside;
      const float sg = gmm_sigma(c, a.logvar, m, j % Da, &inside);
      const float mu = tanhf(mp[e]);
      const float d = ac[j] - mu, var = sg * sg;
      float rm = 0.f;
#pragma unroll
      for (int mm = 0; mm < GMM_MAX_MODES; ++mm) rm = mm == m ? ev.r[mm] : rm;
      g = coef * rm * (d / var) * (1.f - mu * mu);
      glv = coef * rm * 0.5f * ((d * d) / var - 1.f) * inside;
      if (c.std_mode == 1) lds[grp * M * AF + e] = live ? glv : 0.f;
    }
    if (live) dm[e] = P::from_f32(g);
  }
  if (live) {
    E* dl = (E*)a.d_logits + (size_t)nn * a.lddl;
    for (int m = lane; m < a.lddl; m += 64) {
      float g = 0.f;
      if (m < M) {
        float rm = 0.f, lpi = 0.f, hm = 0.f;
#pragma unroll
        for (int mm = 0; mm < GMM_MAX_MODES; ++mm)
          if (mm == m) rm = ev.r[mm], lpi = ev.logpi[mm], hm = H[mm];
        const float pi = expf(lpi);
        g = coef * (rm - pi) - ec * pi * (hm - Hbar);  // d log p / d logit_m = r_m - pi_m ; d (sum pi H) / d logit_m = pi_m (H_m - Hbar)
      }
      dl[m] = P::from_f32(g);
    }
    E* dvp = (E*)a.d_v + (size_t)nn * a.lddv;
    for (int j = lane; j < a.lddv; j += 64) dvp[j] = P::from_f32(j == 0 ? dv / (float)Nn : 0.f);
  }
  if (lane == 0) {
    red[grp][0] = live ? (double)fmaxf(pg1, pg2) : 0.0;
    red[grp][1] = live ? (double)lv : 0.0;
    red[grp][2] = live ? (double)((ratio - 1.f) - logratio) : 0.0;
    red[grp][3] = live && fabsf(ratio - 1.f) > clip ? 1.0 : 0.0;
    red[grp][4] = live ? (double)ratio : 0.0;
    red[grp][5] = live ? (double)Hbar : 0.0;
    red[grp][6] = live ? (double)std_s : 0.0;
    for (int m = 0; m < M; ++m) lvent[grp][m] = live ? expf(ev.logpi[m]) : 0.f;
  }
  __syncthreads();
  const int stride = 8 + M * Da;
  double* o = a.partial + (size_t)blockIdx.x * stride;
  if (threadIdx.x < 7) {
    double t = 0;
    for (int g2 = 0; g2 < GMM_SPB; ++g2) t += red[g2][threadIdx.x];
    o[threadIdx.x] = t;
  }
  if (c.std_mode == 1)  // d / d logvar[m][d]: the pg part over (sample, chunk step) in fixed order + the entropy part
    for (int k = threadIdx.x; k < M * Da; k += 256) {
      const int m = k / Da, d = k % Da;
      float inside;
      (void)gmm_sigma(c, a.logvar, m, d, &inside);
      double t = 0;
      for (int g2 = 0; g2 < GMM_SPB; ++g2) {
        for (int tt = 0; tt < Ta; ++tt) t += (double)lds[g2 * M * AF + m * AF + tt * Da + d];
        t -= (double)(ec * lvent[g2][m] * 0.5f * (float)Ta * inside);
      }
      o[8 + k] = t;
    }
}

__global__ __launch_bounds__(256) void gmm_finalize_kernel(const GmmArgs a, int blocks) {
  __shared__ double sh[256];
  const dppo_gmm_cfg& c = a.cfg;
  const int K = c.num_modes * c.action_dim, stride = 8 + K;
  const double Nn = a.moments[2];
  for (int k = 0; k < 7 + (c.std_mode == 1 ? K : 0); ++k) {
    const int col = k < 7 ? k : 8 + (k - 7);
    double s = 0;
    for (int b = threadIdx.x; b < blocks; b += 256) s += a.partial[(size_t)b * stride + col];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      if (k < 5) a.stats[k] = sh[0] / Nn;
      else if (k == 5) a.stats[7] = sh[0] / Nn;                    // entropy (slot 7, like the Gaussian head's block)
      else if (k == 6) a.stats[DPPO_STAT_COUNT + 0] = sh[0] / Nn;  // mean std
      else a.logvar_grad[k - 7] = (float)sh[0];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double mean = a.moments[0] / Nn;
    const double varu = Nn > 1 ? (a.moments[1] - Nn * mean * mean) / (Nn - 1.0) : 0.0;
    a.stats[DPPO_STAT_ADV_MEAN] = mean;
    a.stats[DPPO_STAT_ADV_STD] = sqrt(varu > 0 ? varu : 0);
  }
}

template <class P>
void launch_gmm_loss(const GmmArgs& a, hipStream_t s) {
  const int blocks = gmm_blocks(a.N);
  const size_t lds = a.cfg.std_mode == 1 ? (size_t)GMM_SPB * a.cfg.num_modes * a.AF * sizeof(float) : sizeof(float);
  hipLaunchKernelGGL((gmm_loss_kernel<P>), dim3(blocks), dim3(256), lds, s, a);
  hipLaunchKernelGGL(gmm_finalize_kernel, dim3(1), dim3(256), 0, s, a, blocks);
}
template void launch_gmm_loss<F32>(const GmmArgs&, hipStream_t);
template void launch_gmm_loss<BF16>(const GmmArgs&, hipStream_t);

}  // namespace dppo
