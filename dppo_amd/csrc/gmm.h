// Mixture-of-Gaussians policy PPO: the reference's GMM_MLP / GMMModel / VPG_GMM / PPO_GMM
// (model/common/mlp_gmm.py:11-110, model/common/gmm.py:14-97, model/rl/gmm_vpg.py:6-46, model/rl/gmm_ppo.py:19-112) as epilogue
// kernels around two trunks (component means, mixture logits) and the critic.
#pragma once
#include "common.h"
#include "dppo_hip.h"

namespace dppo {

constexpr int GMM_MAX_MODES = 8;

struct GmmArgs {
  dppo_gmm_cfg cfg;
  const float* mean_pre;  // [N][ldm]: num_modes * AF columns, before the tanh (mode-major: column m * AF + j)
  int ldm;
  const float* logits;    // [N][ldl]: num_modes columns
  int ldl;
  const float* logvar;    // [num_modes * Da] (std_mode 1) or null
  const float* actions;   // [N][AF]
  int64_t N;
  int AF;
  // sampling
  const int64_t* modes_in;  // [N] chosen components, or null: drawn in the kernel
  const float* noise;       // [N][AF] or null
  float* out_actions;       // [N][AF]
  float* out_logp;          // [N]
  // loss
  const float* vnew;
  int ldv;
  const float *returns, *oldvalues, *adv, *oldlogp;
  const double* moments;
  void *d_mean, *d_logits, *d_v;  // elem, zero padded to lddm / lddl / lddv columns
  int lddm, lddl, lddv;
  double* partial;     // [blocks][8 + num_modes * Da]
  double* stats;
  float* logvar_grad;  // [num_modes * Da]
};

int gmm_blocks(int64_t N);
template <class P>
void launch_gmm_loss(const GmmArgs& a, hipStream_t s);
void launch_gmm_sample(const GmmArgs& a, hipStream_t s);
void launch_gmm_logprob(const GmmArgs& a, hipStream_t s);

}  // namespace dppo
