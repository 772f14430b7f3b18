// Posterior mean of one denoising step and its derivative (shared by the loss / log-prob kernels and the conv
// denoiser's step kernel).  Include only from translation units compiled with -ffp-contract=off.
#pragma once
#include "common.h"
#include "dppo_hip.h"

namespace dppo {

// =================================================================================================
// posterior mean (VPGDiffusion.p_mean_var, diffusion_vpg.py:165-223) and its derivative wrt eps
// =================================================================================================
__device__ __forceinline__ void posterior(const dppo_diffusion_cfg& c, const dppo_step& st, float x, float eps,
                                          float& mu, float& dmu_deps) {
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

}  // namespace dppo
