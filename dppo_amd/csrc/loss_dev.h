// The policy half of the PPO loss for ONE sample whose inputs are already in registers (fused_forward_merged_kernel's LOSSF
// epilogue: the forward's own output tile, the gathered chain pair and old log-probs).  The arithmetic is ppo_loss_kernel's
// (ppo.hip, NREG = 16, 16-byte path), statement by statement: log-probs summed over j in ascending order, the same clamp /
// ratio / clip-schedule / surrogate expressions, the same d loss / d eps.  fused.hip is compiled with floating-point
// contraction ON (its Mish must round as the inference forward's does); everything here must round as ppo.hip's kernels do
// (-ffp-contract=off) or the recomputed log-probs stop matching the precomputed ones bit for bit -- hence the pragma in every
// function body, and a private copy of the posterior (posterior.h belongs to the contraction-off translation units).
#pragma once
#include "common.h"
#include "dppo_hip.h"
#include "ppo.h"

namespace dppo {

#ifndef DPPO_LOG_SQRT_2PI
#define DPPO_LOG_SQRT_2PI 0.91893853320467274178f
#endif

// VPGDiffusion.p_mean_var (diffusion_vpg.py:165-223) and its derivative wrt eps: posterior.h's function, contraction off
__device__ __forceinline__ void posterior_nc(const dppo_diffusion_cfg& c, const dppo_step& st, float x, float eps, float& mu,
                                             float& dmu_deps) {
#pragma clang fp contract(off)
  if (!c.use_ddim) {
    float x0 = st.c0 * x - st.c1 * eps;
    float pass = 1.f;
    if (c.has_denoised_clip) {
      pass = (x0 >= -c.denoised_clip && x0 <= c.denoised_clip) ? 1.f : 0.f;
      x0 = fminf(fmaxf(x0, -c.denoised_clip), c.denoised_clip);
    }
    mu = st.c2 * x0 + st.c3 * x;
    dmu_deps = -(st.c2 * st.c1) * pass;
  } else {
    float x0 = (x - st.c1 * eps) / st.c0;
    float dx0 = -st.c1 / st.c0;
    float e2 = eps, de2 = 1.f;
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

// one element of the log-prob sums (ppo_loss_kernel's `element`)
__device__ __forceinline__ void loss_element_nc(const dppo_diffusion_cfg& dc, const dppo_step& st, float var, float lstd, float x,
                                                float xn, float e, float o, float& sum_new, float& sum_old, float& gs) {
#pragma clang fp contract(off)
  float mu, dmu;
  posterior_nc(dc, st, x, e, mu, dmu);
  const float d = xn - mu;
  const float lp = -(d * d) / (2.f * var) - lstd - DPPO_LOG_SQRT_2PI;
  sum_new += fminf(fmaxf(lp, -5.f), 2.f);
  sum_old += fminf(fmaxf(o, -5.f), 2.f);
  gs = (lp >= -5.f && lp <= 2.f) ? (d / var) * dmu : 0.f;
}

// tab: [Kft] denoising discount, [Kft] clip range, adv mean, adv std, [Kft] log std_k (the loss kernel's table + one row, in LDS).
// xs / xns / os: the sample's x_k, x_k+1 and old log-probs in LDS (4-byte aligned), eps: the forward's eps row (16-byte aligned);
// the first cnt elements count (cnt <= 16, a multiple of 4), read four at a time (all 64 in registers at once spilled the
// kernel).  de: the sample's row of d loss / d eps ([ldde] elem, whole 16-byte chunks).  s4: pg loss, approx kl, clip fraction,
// ratio of this sample.
template <class P>
__device__ __forceinline__ void policy_loss_row_nc(const LossArgs& a, const float* tab, int k, float adv, const float* xs,
                                                   const float* xns, const float* os, const float* eps, int cnt, double Nn,
                                                   typename P::elem_t* de, double (&s4)[4]) {
#pragma clang fp contract(off)
  constexpr int EPC = 16 / P::ESIZE;
  const dppo_ppo_cfg& pc = a.pcfg;
  const int Kft = pc.ft_denoising_steps;
  const dppo_step st = a.ksteps[k];
  // (log std_k from the table: logf expands differently under this translation unit's contraction setting, pragma or not)
  const float var = st.std * st.std, lstd = tab[2 * Kft + 2 + k];
  float sum_new = 0.f, sum_old = 0.f;
  float gsrc[16];
#pragma unroll
  for (int j0 = 0; j0 < 16; j0 += 4) {
    if (j0 < cnt) {
      f32x4 x, xn, o;
#pragma unroll
      for (int q = 0; q < 4; ++q) x[q] = lds_load(xs + j0 + q), xn[q] = lds_load(xns + j0 + q), o[q] = lds_load(os + j0 + q);
      const f32x4 e = lds_load((const f32x4*)(eps + j0));
      loss_element_nc(a.dcfg, st, var, lstd, x[0], xn[0], e[0], o[0], sum_new, sum_old, gsrc[j0]);
      loss_element_nc(a.dcfg, st, var, lstd, x[1], xn[1], e[1], o[1], sum_new, sum_old, gsrc[j0 + 1]);
      loss_element_nc(a.dcfg, st, var, lstd, x[2], xn[2], e[2], o[2], sum_new, sum_old, gsrc[j0 + 2]);
      loss_element_nc(a.dcfg, st, var, lstd, x[3], xn[3], e[3], o[3], sum_new, sum_old, gsrc[j0 + 3]);
      __builtin_amdgcn_sched_barrier(0);  // (one group of four at a time)
    } else {
      gsrc[j0] = gsrc[j0 + 1] = gsrc[j0 + 2] = gsrc[j0 + 3] = 0.f;
    }
  }
  const float newlp = sum_new / (float)cnt, oldlp = sum_old / (float)cnt;
  if (pc.norm_adv) adv = (adv - tab[2 * Kft]) / (tab[2 * Kft + 1] + 1e-8f);
  if (pc.has_adv_clip) adv = fminf(fmaxf(adv, pc.adv_clip_lo), pc.adv_clip_hi);
  adv *= tab[k];
  const float logratio = newlp - oldlp;
  const float ratio = expf(logratio);
  const float eps_k = tab[Kft + k];
  const float lo = 1.f - eps_k, hi = 1.f + eps_k;
  const float rc = fminf(fmaxf(ratio, lo), hi);
  const float pg1 = -adv * ratio, pg2 = -adv * rc;
  const float w1 = pg1 > pg2 ? 1.f : (pg1 == pg2 ? 0.5f : 0.f);
  const float within = (ratio >= lo && ratio <= hi) ? 1.f : 0.f;
  const float dL_dratio = -adv * (w1 + (1.f - w1) * within);
  const float coef = dL_dratio * ratio / ((float)Nn * (float)cnt);
  s4[0] = fmaxf(pg1, pg2);
  s4[1] = (double)((ratio - 1.f) - logratio);
  s4[2] = fabsf(ratio - 1.f) > eps_k ? 1.0 : 0.0;
  s4[3] = ratio;
  const int nch = a.ldde / EPC;
#pragma unroll
  for (int cc = 0; cc < 16 / EPC; ++cc) {
    if (cc < nch) {
      float v[EPC];
#pragma unroll
      for (int q = 0; q < EPC; ++q) v[q] = cc * EPC + q < cnt ? coef * gsrc[cc * EPC + q] : 0.f;
      u32x4 w;
      if constexpr (P::ESIZE == 4) {
        w = (u32x4){__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) w[q] = (uint32_t)f2bf(v[2 * q]) | ((uint32_t)f2bf(v[2 * q + 1]) << 16);
      }
      *(u32x4*)(de + cc * EPC) = w;
    }
  }
  const u32x4 z = (u32x4){0, 0, 0, 0};
  for (int c = 16 / EPC < nch ? 16 / EPC : nch; c < nch; ++c) *(u32x4*)(de + c * EPC) = z;
}

}  // namespace dppo
