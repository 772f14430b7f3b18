// Gaussian-policy PPO: the reference's PPO_Gaussian.loss / VPG_Gaussian.get_logprobs /
// GaussianModel.forward (model/rl/gaussian_ppo.py:39-128, model/rl/gaussian_vpg.py:46-68, model/common/gaussian.py:63-121)
// as epilogue kernels around the same fused MLP forward / backward the diffusion path uses.
#pragma once
#include "common.h"
#include "dppo_hip.h"

namespace dppo {

struct GaussArgs {
  dppo_gaussian_cfg cfg;
  const float* mean_pre;  // [N][ldm] trunk output before the tanh
  int ldm;
  const float* logvar;    // [Da] (std_mode 1) or null
  const float* actions;   // [N][AF]
  int64_t N;
  int AF;
  // sampling
  const float* noise;     // [N][AF] or null: Philox keyed by cfg.seed_lo / seed_hi, counter = element index
  float* out_actions;     // [N][AF]
  float* out_logp;        // [N] mean over AF of log N(a; mu, sigma), or null
  float* out_mean;        // [N][AF] or null
  // loss
  const float* vnew;      // [N][ldv] critic output (column 0)
  int ldv;
  const float *returns, *oldvalues, *adv, *oldlogp;  // [N]
  const double* moments;  // {sum adv, sum adv^2, count} of the (global) minibatch
  void* d_mean;           // [N][lddm] elem: d loss / d mean_pre
  int lddm;
  void* d_v;              // [N][lddv] elem
  int lddv;
  double* partial;        // [blocks][8 + Da]
  double* stats;          // [DPPO_STAT_COUNT]
  float* logvar_grad;     // [Da]: d pg_loss / d logvar (std_mode 1)
  double* adv_moments_out;  // local moments workspace [3 + 2 * blocks']
};

int gauss_blocks(int64_t N);
template <class P>
void launch_gauss_loss(const GaussArgs& a, hipStream_t s);  // loss + d_mean + d_v + per-block partials, then the finalize
void launch_gauss_sample(const GaussArgs& a, hipStream_t s);
void launch_gauss_logprob(const GaussArgs& a, hipStream_t s);
void launch_gauss_moments(const float* adv, int64_t N, double* moments, double* scratch, hipStream_t s);

}  // namespace dppo
