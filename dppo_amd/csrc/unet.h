// Training side of the conv denoiser (unet.hip): forward with a tape, backward to every parameter gradient.  Used by the
// PPO-update and supervised-loss entry points in api.hip, which own the loss kernels and the critic pipeline.
#pragma once
#include "common.h"
#include "dppo_hip.h"

namespace dppo {

struct UnetTrainIO {  // where sample n's network input comes from
  const float* chains;  // rollout mode: [R][Kft+1][AF]; gathered mode: [N][2][AF] (x first)
  const float* obs;     // [R or N][cond]
  const int32_t* brow;  // [N] buffer row of sample n
  const int32_t* krow;  // [N] denoising-step index of sample n
  const dppo_step* ksteps;  // device: ksteps[k].t = row of the time-embedding table
  int Kft, gathered;
};
template <class P>
struct UnetTrainer;
template <class P>
size_t unet_trainer_bytes(const dppo_unet_desc& d, int64_t N);
template <class P>
UnetTrainer<P>* unet_trainer_new(const dppo_unet_desc& d, const float* prm, const char* pk, int64_t N, void* ws, size_t wsb,
                                 hipStream_t s);
template <class P>
float* unet_trainer_forward(UnetTrainer<P>* t, const UnetTrainIO& io);  // eps [N][Ta*Da] f32 (lives in the workspace)
// d_eps: elem [N][ldde] = d loss / d eps (column t*Da + c), zero beyond Ta*Da; grad: flat fp32, state-dict order, OVERWRITTEN
// d_obs (optional): f32 [N][cond_dim] <- d loss / d observation (what a visual encoder in front of the network continues from)
template <class P>
void unet_trainer_backward(UnetTrainer<P>* t, const void* d_eps, int ldde, float* grad, float* d_obs = nullptr);
template <class P>
void unet_trainer_free(UnetTrainer<P>* t);
int unet_check_desc(const dppo_unet_desc* d);
void launch_chain_init(const float* noise, uint32_t k0, uint32_t k1, int64_t n, int AF, float* x, float* chains, int chain_len,
                       int init_slot, hipStream_t s);
void launch_chain_step(const dppo_diffusion_cfg& cfg, const dppo_step& st, float* x, const float* eps, int lde, const float* noise,
                       size_t nz0, int64_t n, int AF, int chain_len, int last, float* chains, float* traj, hipStream_t s);
// brow[n], krow[n] of sample n (rollout mode: ind / Kft, ind % Kft; gathered mode: n, kinds[n])
void launch_unet_index(const int64_t* inds, const int64_t* kinds, int Kft, int64_t N, int32_t* brow, int32_t* krow,
                       hipStream_t s);

}  // namespace dppo
