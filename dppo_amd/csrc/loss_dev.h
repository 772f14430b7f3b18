// The per-sample body of the fused PPO loss and the posterior it differentiates, shared by ppo_loss_kernel (ppo.hip) and
// by the fused forward's epilogue (fused.hip: the policy half evaluated on the tile while its eps is still on chip).
// Contraction is switched off inside these functions so that both translation units -- and the log-prob precompute --
// round identically whatever their compile flags: the ratio of a recomputed to a precomputed log-prob must be exactly 1.
#pragma once
#include "common.h"
#include "loss_args.h"

namespace dppo {

// =================================================================================================
// posterior mean (VPGDiffusion.p_mean_var, diffusion_vpg.py:165-223) and its derivative wrt eps
// =================================================================================================
__device__ __forceinline__ void posterior(const dppo_diffusion_cfg& c, const dppo_step& st, float x, float eps,
                                          float& mu, float& dmu_deps) {
#pragma clang fp contract(off)
  if (!c.use_ddim) {
    float x0 = st.c0 * x - st.c1 * eps;
    float pass = 1.f;
    if (c.has_denoised_clip) {
      pass = (x0 >= -c.denoised_clip && x0 <= c.denoised_clip) ? 1.f : 0.f;  // clamp backward: inclusive
      x0 = fminf(fmaxf(x0, -c.denoised_clip), c.denoised_clip);
    }
    mu = st.c2 * x0 + st.c3 * x;
    dmu_deps = -(st.c2 * st.c1) * pass;
  } else {
    float x0 = (x - st.c1 * eps) / st.c0;
    float dx0 = -st.c1 / st.c0;  // d x0 / d eps
    float e2 = eps, de2 = 1.f;   // eps after the re-derivation, d e2 / d eps
    if (c.has_denoised_clip) {
      const float pass = (x0 >= -c.denoised_clip && x0 <= c.denoised_clip) ? 1.f : 0.f;
      x0 = fminf(fmaxf(x0, -c.denoised_clip), c.denoised_clip);
      dx0 *= pass;
      e2 = (x - st.c0 * x0) / st.c1;
      de2 = -(st.c0 / st.c1) * dx0;
    }
    if (c.has_eps_clip) {
      const float pass = (e2 >= -c.eps_clip && e2 <= c.eps_clip) ? 1.f : 0.f;
      e2 = fminf(fmaxf(e2, -c.eps_clip), c.eps_clip);
      de2 *= pass;
    }
    mu = st.c2 * x0 + st.c3 * e2;
    dmu_deps = st.c2 * dx0 + st.c3 * de2;
  }
}

#define DPPO_LOG_SQRT_2PI 0.91893853320467274178f

// One sample (16 lanes: lane `sub` owns chunk elements sub, sub + 16, ...) of PPODiffusion.loss.  n: sample index, ep: its
// eps row (global or LDS), tab: [Kft] discount, [Kft] clip range, [2] advantage mean / std.
template <class P>
__device__ __forceinline__ void ppo_loss_sample(const LossArgs& a, const float* tab, const int64_t n, const float* ep,
                                                const int sub, const double Nn, double& s_pg, double& s_v, double& s_kl,
                                                double& s_cf, double& s_ratio, float (&cs)[4], float& cs_v) {
#pragma clang fp contract(off)
  typedef typename P::elem_t E;
  const dppo_ppo_cfg& pc = a.pcfg;
  const int Kft = pc.ft_denoising_steps, AF = a.AF, Da = pc.action_dim;
  const int rh = pc.reward_horizon < pc.horizon_steps ? pc.reward_horizon : pc.horizon_steps;
  const int cnt = rh * Da;
  {
    const bool live = n < a.N;
    const int64_t nn = live ? n : a.N - 1;  // out-of-range lanes shadow the last sample (shuffles need all lanes), write nothing
    const int b = a.brow[nn], k = (a.part & 1) ? a.krow[nn] : 0;  // the value half never looks at the denoising step
    const dppo_step st = a.ksteps[k];
    const float* ch = a.gathered ? a.chains + (size_t)b * 2 * AF : a.chains + ((size_t)b * (Kft + 1) + k) * AF;
    const float* olp = a.gathered ? a.logprobs_k + (size_t)b * AF : a.logprobs_k + ((size_t)b * Kft + k) * AF;
    const float var = st.std * st.std, lstd = logf(st.std);
    const bool pol = (a.part & 1) != 0, val = (a.part & 2) != 0;
    // ---- new / old log-probs, clamped to [-5, 2], averaged over the first `rh` chunk steps (:93-102)
    float sum_new = 0.f, sum_old = 0.f;
    for (int j = sub; pol && j < cnt; j += 16) {
      float mu, dmu;
      posterior(a.dcfg, st, ch[j], ep[j], mu, dmu);
      const float d = ch[AF + j] - mu;
      const float lp = -(d * d) / (2.f * var) - lstd - DPPO_LOG_SQRT_2PI;
      sum_new += fminf(fmaxf(lp, -5.f), 2.f);
      sum_old += fminf(fmaxf(olp[j], -5.f), 2.f);
    }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
      sum_new += __shfl_xor(sum_new, o);
      sum_old += __shfl_xor(sum_old, o);
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
    // ---- value loss (:177-189)
    const float v = val ? a.vnew[(size_t)nn * a.ldv] : 0.f;
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
    if (live) {
      if (sub == 0 && pol) {
        s_kl += (double)((ratio - 1.f) - logratio);
        s_cf += fabsf(ratio - 1.f) > eps_k ? 1.0 : 0.0;
        s_ratio += ratio;
        s_pg += fmaxf(pg1, pg2);
      }
      if (sub == 0 && val) s_v += lv;
      // ---- d loss / d eps and d loss / d v, zero padded to the GEMM K width, 16 lanes x 4 elements per pass
      E* de = (E*)a.d_eps + (size_t)n * a.ldde;
      for (int j0 = 4 * sub; pol && j0 < a.ldde; j0 += 64) {
        float gq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int j = j0 + q;
          float gj = 0.f;
          if (j < cnt) {
            float mu, dmu;
            posterior(a.dcfg, st, ch[j], ep[j], mu, dmu);
            const float d = ch[AF + j] - mu;
            const float lp = -(d * d) / (2.f * var) - lstd - DPPO_LOG_SQRT_2PI;
            if (lp >= -5.f && lp <= 2.f) gj = coef * (d / var) * dmu;
          }
          gq[q] = gj;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) de[j0 + q] = P::from_f32(gq[q]);
        if (j0 < 64) {
#pragma unroll
          for (int q = 0; q < 4; ++q) cs[q] += gq[q];
        }
      }
      if (sub == 0) cs_v += dv / (float)Nn;
      E* dvp = (E*)a.d_v + (size_t)n * a.lddv;
      for (int j0 = 4 * sub; val && j0 < a.lddv; j0 += 64)
#pragma unroll
        for (int q = 0; q < 4; ++q) dvp[j0 + q] = P::from_f32(j0 + q == 0 ? dv / (float)Nn : 0.f);
    }
    }
}

}  // namespace dppo
