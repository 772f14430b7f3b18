// Arguments of the fused PPO loss: shared by ppo.hip (the loss kernel) and fused.hip (its policy half in the fused
// forward's epilogue).
#pragma once
#include <stdint.h>

#include "../../include/dppo_hip.h"

namespace dppo {

// ---- fused PPO loss (diffusion_ppo.py:85-199) -----------------------------------------------------
struct LossArgs {
  const float* eps;  // [N][lde] actor output
  int lde;
  const float* vnew;  // [N][ldv] critic output (column 0)
  int ldv;
  const int32_t* brow;
  const int32_t* krow;
  int gathered;             // 1: chains [N][2][AF], logprobs_k [N][AF] (already gathered per sample)
  const float* chains;      // [R][Kft+1][AF]
  const float* logprobs_k;  // [R][Kft][AF]
  const float* returns_k;
  const float* values_k;
  const float* adv_k;
  const dppo_step* ksteps;
  dppo_diffusion_cfg dcfg;
  dppo_ppo_cfg pcfg;
  int AF;
  int64_t N;
  const double* moments;  // [3] sum(adv), sum(adv^2), count over the (global) minibatch
  const float* tab;       // [2 Kft] per-k discount and clip range built by the row builder (null: built per block)
  double n_count;  // > 0: the (global) minibatch sample count, instead of moments[2] (the value half must not wait for
                   // the advantage-moment kernel on the other stream)
  int part;  // bit 0: the policy half (log-probs, surrogate, d_eps; needs eps), bit 1: the value half (v loss, d_v; needs
             // vnew) -- the two halves of one update can then run on the actor's and the critic's stream, no join
  void* d_eps;            // [N][ldde] elem, zero padded
  int ldde;
  void* d_v;  // [N][lddv] elem, column 0, zero padded
  int lddv;
  double* stats;    // [DPPO_STAT_COUNT], zeroed by the caller
  double* partial;  // [loss_blocks(N)][8] scratch
  // optional fused out-layer bias gradients (needs ldde == 64): column sums of d_eps -> gb_actor[out_dim], of d_v -> gb_critic[0]
  float* partial_cs;  // [loss_blocks(N)][65] scratch, or null
  float* gb_actor;
  float* gb_critic;
  int out_dim;
};

}  // namespace dppo
