/* dppo_hip.h -- C ABI of libdppo_hip.so: the MI355X (gfx950) DPPO hot path.
 *
 * The reference (enyen/dppo) has no FFI layer; its seam is Hydra `_target_` + nn.Module duck typing
 * (SURVEY.md 8b).  This header is the boundary a native replacement exports so that the Python
 * module `dppo_amd` (a mirror of the reference's PPODiffusion / DiffusionMLP / CriticObs classes)
 * is a thin ctypes shim.  Each entry point names the reference code it replaces
 * (paths relative to /root/reference/dppo).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless the name ends in _host; plain C types only;
 *  - no entry point allocates, frees or synchronises; callers pass workspaces (sizes from the
 *    *_workspace_bytes queries) and the HIP stream (hipStream_t passed as void*);
 *  - return 0 = ok, <0 = bad argument / unsupported shape (dppo_last_error() has the text),
 *    >0 = hipError_t from a launch;
 *  - state: the error string is thread-local; the tuning knobs (dppo_tune_set) and the measurement probe are
 *    process-wide (plain globals read at launch time: set them before issuing work, never concurrently with it; they
 *    select between implementations that all pass the same parity tests, the product path never touches them).  dppo_ppo_loss_fwd_bwd / dppo_bc_loss_fwd_bwd / dppo_denoise_mse_fwd_bwd fork onto library-owned side
 *    streams (one set per device, joined back into the caller's stream before they return control of it, capture-safe):
 *    issue at most one such call per device at a time.  Every other entry point is re-entrant across streams.
 *
 * Networks are the reference's residual MLP family: Linear(in,H) -> n_blocks x
 * [h + l2(act(l1(act(h))))] -> Linear(H,out)  (model/common/mlp.py:84-154), for the actor preceded
 * by the sinusoidal time embedding MLP and concat [x, t_emb, state]
 * (model/diffusion/mlp_diffusion.py:191-196,246).  Parameters live in ONE flat fp32 buffer in
 * state-dict order:
 *   actor : time_embedding.1.{weight(2td x td),bias}, time_embedding.3.{weight(td x 2td),bias},
 *           [cond_mlp.moduleList.0.linear_1.{weight(ch x cond),bias}, cond_mlp.moduleList.1.linear_1.{weight(co x ch),bias}],
 *           layers.0.{weight(H x in),bias}, [layers.b.l1.{w,b}, layers.b.l2.{w,b} (, norm1.{w,b}, norm2.{w,b})] x n_blocks,
 *           layers.last.{weight(out x H),bias}
 *   critic: the same without the time embedding (Q1.layers.*).
 * nn.Linear layout (out,in) row-major.  Gradients use the same flat layout.
 */
#ifndef DPPO_HIP_H
#define DPPO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DPPO_PREC_F32 0  /* v_mfma_f32_16x16x4_f32 : exact fp32 products (parity mode)      */
#define DPPO_PREC_BF16 1 /* v_mfma_f32_16x16x32_bf16: bf16 operands, fp32 accumulate         */

#define DPPO_ACT_RELU 0
#define DPPO_ACT_MISH 1

typedef void* dppo_stream_t; /* hipStream_t */

typedef struct dppo_net_desc {
  int32_t kind;     /* 0 = actor (DiffusionMLP), 1 = trunk on the observation (CriticObs; Gaussian_MLP mean) */
  int32_t in_dim;   /* actor: Ta*Da + time_dim + cond_dim ; critic: cond_dim                 */
  int32_t hidden;   /* H, multiple of 128                                                    */
  int32_t n_blocks; /* residual blocks = (len(mlp_dims)-1)/2                                  */
  int32_t out_dim;  /* actor: Ta*Da ; critic: 1 ; Gaussian actor (kind 1): Ta*Da                */
  int32_t act;      /* DPPO_ACT_*                                                            */
  int32_t time_dim; /* actor: td (even, >= 4) ; critic: 0                                    */
  int32_t act_flat; /* actor: Ta*Da ; critic: 0                                              */
  int32_t cond_dim; /* To*Do                                                                 */
  /* optional observation encoder of the actor (DiffusionMLP cond_mlp_dims = [cond_hidden, cond_out],
   * model/diffusion/mlp_diffusion.py:201-207,240-241): state' = Linear(act(Linear(state))).  0/0 = none.
   * With it, in_dim = Ta*Da + time_dim + cond_out. */
  int32_t cond_hidden;
  int32_t cond_out;
  /* 1: the residual blocks are LayerNorm blocks, x + l2(act(norm2(l1(act(norm1(x)))))) with eps 1e-6
   * (model/common/mlp.py:139-154); parameters norm1.{weight,bias}, norm2.{weight,bias} follow l2 in each block.
   * Needs a fused-kernel shape: hidden in {256, 512, 1024}. */
  int32_t use_layernorm;
  /* 1: plain (non-residual) MLP trunk, model/common/mlp.py:27-81 (residual_style: False): Linear(in, H) -> act ->
   * n_blocks x [Linear(H, H) -> act] -> Linear(H, out), parameters moduleList.{i}.linear_1.{weight,bias}.  One hidden width
   * (a multiple of 64), n_blocks >= 1, no LayerNorm, no cond_mlp; runs on the layered gemm_nt path (no fused kernels, and
   * sampling through dppo_plain_sample_chain).  No shipped DPPO / PPO cfg uses it. */
  int32_t plain;
} dppo_net_desc;

/* One denoising step, host-prepared in fp32 exactly as the reference computes its tables
 * (model/diffusion/diffusion.py:98-196, diffusion_vpg.py:168-223,279-293).
 *   DDPM: c0 = sqrt(1/abar_t), c1 = sqrt(1/abar_t - 1), c2 = mu_coef1, c3 = mu_coef2
 *   DDIM: c0 = sqrt(alpha), c1 = sqrt(1 - alpha), c2 = sqrt(alpha_prev), c3 = sqrt(clamp(1-alpha_prev-sigma^2, 0))
 *   std : the noise scale used at this step (sampling: after the min-std / deterministic rule;
 *         log-prob tables: max(exp(.5 logvar), min_logprob_denoising_std))                      */
typedef struct dppo_step {
  int32_t net;        /* 0 = frozen base actor, 1 = fine-tuned actor                         */
  int32_t t;          /* row of the time-embedding table (the diffusion time)                 */
  int32_t chain_slot; /* sampler: chain position that receives x AFTER this step, or -1       */
  int32_t final_clip; /* sampler: 1 = clamp x to +-final_action_clip_value after this step    */
  float c0, c1, c2, c3;
  float std;
  float pad;
} dppo_step;

typedef struct dppo_diffusion_cfg {
  int32_t use_ddim;
  int32_t has_denoised_clip; /* denoised_clip_value is not None */
  int32_t has_eps_clip;      /* eps_clip_value is not None (DDIM only) */
  int32_t has_final_clip;
  float denoised_clip, eps_clip, randn_clip, final_clip;
  uint32_t seed_lo, seed_hi; /* dppo_sample_chain with noise == NULL: key of the in-kernel Philox4x32-10 generator */
} dppo_diffusion_cfg;

typedef struct dppo_ppo_cfg {
  int32_t ft_denoising_steps; /* Kft                                                         */
  int32_t horizon_steps;      /* Ta                                                          */
  int32_t action_dim;         /* Da                                                          */
  int32_t reward_horizon;     /* chunk steps whose log-probs enter the loss                  */
  int32_t norm_adv;
  int32_t has_adv_clip;       /* quantile thresholds below are active                        */
  int32_t has_vclip;
  int32_t pad;
  /* Python floats of the reference's constructor, kept as doubles: the denoising discount is
   * float32(pow(double gamma, Kft-k-1)) and the clip schedule mixes double scalars into fp32 math
   * (diffusion_ppo.py:138-159) */
  double gamma_denoising, clip_ploss_coef, clip_ploss_coef_base, clip_ploss_coef_rate, clip_vloss_coef;
  float adv_clip_lo, adv_clip_hi; /* torch.quantile thresholds of the normalised advantages */
} dppo_ppo_cfg;

/* index of each statistic in the `stats` output of dppo_ppo_loss_fwd_bwd (device doubles) */
enum {
  DPPO_STAT_PG_LOSS = 0,
  DPPO_STAT_V_LOSS = 1,
  DPPO_STAT_APPROX_KL = 2,
  DPPO_STAT_CLIPFRAC = 3,
  DPPO_STAT_RATIO = 4,
  DPPO_STAT_ADV_MEAN = 5,
  DPPO_STAT_ADV_STD = 6,
  DPPO_STAT_COUNT = 8
};

int dppo_version(void);
const char* dppo_last_error(void);

/* ---- parameters ------------------------------------------------------------------------ */
/* number of fp32 values in the flat parameter buffer of `net` */
int64_t dppo_net_param_count(const dppo_net_desc* net);
/* bytes of the packed (kernel-ready) image of `net`: time-embedding table for n_time diffusion
 * times, GEMM operand copies (W and W^T) and the sampler's per-wave fragment streams */
int64_t dppo_packed_bytes(const dppo_net_desc* net, int prec, int n_time);
/* build the packed image from the fp32 master parameters; call again after every optimiser step */
int dppo_pack_net(const dppo_net_desc* net, int prec, int n_time, const float* params, void* packed,
                  dppo_stream_t stream);
/* The same for two networks (actor_ft and critic after an optimiser step) in two launches instead of four: both
 * composites, then both networks' images. */
int dppo_pack_nets(const dppo_net_desc* net0, int n_time0, const float* params0, void* packed0,
                   const dppo_net_desc* net1, int n_time1, const float* params1, void* packed1, int prec,
                   dppo_stream_t stream);

/* ---- A2-A5: network forwards ------------------------------------------------------------- */
/* DiffusionMLP.forward (model/diffusion/mlp_diffusion.py:218-250): x (B,Ta*Da), t (B,) int64,
 * state (B,cond) -> eps (B,Ta*Da) */
int64_t dppo_mlp_forward_workspace_bytes(const dppo_net_desc* net, int prec, int64_t rows);
int dppo_actor_forward(const dppo_net_desc* net, int prec, const float* params, const void* packed,
                       const float* x, const int64_t* t, const float* state, int64_t rows, float* eps,
                       void* workspace, int64_t workspace_bytes, dppo_stream_t stream);
/* CriticObs.forward (model/common/critic.py:40-54): state (B,cond) -> value (B,) */
int dppo_critic_forward(const dppo_net_desc* net, int prec, const float* params, const void* packed,
                        const float* state, int64_t rows, float* values, void* workspace,
                        int64_t workspace_bytes, dppo_stream_t stream);

/* ---- A6-A7: VPGDiffusion.forward (model/diffusion/diffusion_vpg.py:227-315) --------------- */
/* obs (B,cond); noise (n_steps+1,B,Ta*Da): noise[0] is x_K, noise[i+1] the draw of step i (clamped
 * to +-randn_clip inside); sched: n_steps device entries; traj (B,Ta*Da); chains (B,chain_len,Ta*Da)
 * (may be NULL when chain_len == 0); init_slot: chain position of x_K or -1.
 * noise: (n_steps+1, B, Ta*Da) pre-drawn N(0,1) (noise[0] = x_K, noise[i+1] = the draw of step i; parity runs), or NULL:
 * the kernel then draws them itself (Philox4x32-10 keyed by cfg->seed_*, counter = the element's index in that tensor,
 * Box-Muller) -- same distribution as the reference's torch.randn / randn_like, one launch instead of two. */
/* Workspace: the cond_mlp encodings (if any) and, where the call runs as the eight-workgroups-per-tile kernel (knob 27: bf16,
 * hidden 512, one residual block, no LayerNorm, out_dim <= 64, in_dim <= 96, time_dim % 4 == 0, ceil(B / 16) * 8 <= the
 * device's CU count, i.e. B <= 512 on MI355X), its exchange block, which comes first; 0 when neither applies.  The
 * caller zeroes the FIRST 256 BYTES of a new workspace once (the sticky time-out word lives there; no call ever clears it);
 * the rest needs no initialisation (the call zeroes the exchange slots on `stream` before its launch).  Two calls that may
 * run concurrently (different streams) must not share a workspace.  The first 32-bit word of an exchange block is 0, or
 * 1 + the largest denoising step at which a workgroup of ANY call since the host last cleared it gave up waiting for its
 * tile's other seven (bounded spin; the rows of `traj` and every chain slot that workgroup's tile owned are NaN then) --
 * never observed outside the test that forces it (knob 29). */
int64_t dppo_sample_chain_workspace_bytes(const dppo_net_desc* actor, int prec, int64_t B);
/* Bytes of the exchange block at the head of that workspace, 0 when the call will run one workgroup per tile: tells a caller
 * whether the workspace's first word is the time-out word described above (a host that already synchronises once per rollout
 * reads it there and raises; dppo_amd.model.diffusion.DiffusionModel.check_sampler_health does). */
int64_t dppo_sample_chain_exchange_bytes(const dppo_net_desc* actor, int prec, int64_t B);
int dppo_sample_chain(const dppo_net_desc* actor, int prec, const float* params_base, const void* packed_base,
                      const float* params_ft, const void* packed_ft, const dppo_diffusion_cfg* cfg,
                      const dppo_step* sched, int n_steps, const float* obs, const float* noise, int64_t B,
                      float* traj, float* chains, int chain_len, int init_slot, void* workspace,
                      int64_t workspace_bytes, dppo_stream_t stream);

/* dppo_sample_chain for a plain (non-residual, net.plain = 1) denoiser: same arguments except that the step table is in HOST
 * memory (the loop over steps runs on the host: one layered forward + one posterior / noise kernel per step). */
int64_t dppo_plain_sample_workspace_bytes(const dppo_net_desc* actor, int prec, int64_t B);
int dppo_plain_sample_chain(const dppo_net_desc* actor, int prec, const float* params_base, const void* packed_base,
                            const float* params_ft, const void* packed_ft, const dppo_diffusion_cfg* cfg,
                            const dppo_step* sched_host, int n_steps, const float* obs, const float* noise, int64_t B,
                            float* traj, float* chains, int chain_len, int init_slot, void* workspace,
                            int64_t workspace_bytes, dppo_stream_t stream);

/* ---- A8: VPGDiffusion.get_logprobs (diffusion_vpg.py:319-396) ----------------------------- */
/* obs (B,cond), chains (B,Kft+1,Ta*Da) -> logprobs (B,Kft,Ta*Da).  ksteps: Kft device entries,
 * entry k describes chain position k (t = Kft-1-k for DDPM); all rows use the network passed in. */
int64_t dppo_chain_logprob_workspace_bytes(const dppo_net_desc* actor, int prec, int64_t B, int Kft);
int dppo_chain_logprob(const dppo_net_desc* actor, int prec, const float* params, const void* packed,
                       const dppo_diffusion_cfg* cfg, const dppo_step* ksteps, int Kft, const float* obs,
                       const float* chains, int64_t B, float* logprobs, void* workspace,
                       int64_t workspace_bytes, dppo_stream_t stream);

/* ---- A9(3): behaviour-cloning term of PPODiffusion.loss (diffusion_ppo.py:104-126) ----------- */
/* chains (B,Kft+1,Ta*Da): sampled by the caller with the BASE policy (dppo_sample_chain, base weights on both
 * network slots).  Evaluates the fine-tuned network (params / packed) on them, writes
 *   loss[0] = -mean over (b, k, Ta, Da) of clamp(log N(x_{k+1}; mu(x_k, t_k, s_b), sigma_k), -5, 2)   (double, device)
 * and grad (dppo_net_param_count floats, overwritten) = d loss / d params.  dppo_axpy adds it to the PPO gradient
 * with the caller's bc_loss_coeff (train_ppo_diffusion_agent.py:351-357). */
int64_t dppo_bc_loss_workspace_bytes(const dppo_net_desc* actor, int prec, int64_t B, int Kft);
int dppo_bc_loss_fwd_bwd(const dppo_net_desc* actor, int prec, const float* params, const void* packed,
                         const dppo_diffusion_cfg* cfg, const dppo_step* ksteps, int Kft, const float* obs,
                         const float* chains, int64_t B, float* grad, double* loss, void* workspace,
                         int64_t workspace_bytes, dppo_stream_t stream);
int dppo_axpy(float* y, const float* x, double alpha, int64_t n, dppo_stream_t stream);

/* ---- next row: supervised denoising loss of pre-training ---------------------------------
 * DiffusionModel.p_losses + loss.backward() (model/diffusion/diffusion.py:325-349, agent/pretrain/train_diffusion_agent.py):
 * row n = (x_noisy[n], time embedding of tsteps[kinds[n]].t, obs[n]) through the network;
 *   loss[0] = mean over (n, Ta*Da) of (eps_theta - target)^2   (double, device; summed in a fixed order)
 *   grad (dppo_net_param_count floats, overwritten) = d loss / d params.
 * pairs (N, 2, Ta*Da): [n][0] = x_noisy = sqrt(abar_t) x0 + sqrt(1 - abar_t) noise (q_sample, :351-363), [n][1] = the
 * regression target (the noise when predicting epsilon).  tsteps: n_time entries whose .t is the diffusion time (the
 * other fields are ignored); kinds (N,) int64 indexes it.  Same kernels as the PPO update: row builder, fused forward /
 * backward, grouped weight-gradient GEMM. */
int64_t dppo_denoise_mse_workspace_bytes(const dppo_net_desc* actor, int prec, int64_t N);
int dppo_denoise_mse_fwd_bwd(const dppo_net_desc* actor, int prec, const float* params, const void* packed,
                             const dppo_step* tsteps, int n_time, const float* obs, const float* pairs,
                             const int64_t* kinds, int64_t N, float* grad, double* loss, void* workspace,
                             int64_t workspace_bytes, dppo_stream_t stream); /* y += alpha * x */

/* ---- A10: GAE (agent/finetune/train_ppo_diffusion_agent.py:255-279) ----------------------- */
/* reward (S,E) float64 (already scaled by the host-side running scaler), values (S,E) fp32,
 * terminated (S,E) fp32 0/1, last_values (E,) fp32.  Float64 arithmetic like the reference's numpy
 * holders.  Outputs (any may be NULL): adv64/ret64 (S,E) float64, adv32/ret32 (S,E) fp32. */
int dppo_gae(const double* reward, const float* values, const float* terminated, const float* last_values,
             int n_steps, int n_envs, double gamma, double gae_lambda, double reward_scale_const, double* adv64,
             double* ret64, float* adv32, float* ret32, dppo_stream_t stream);

/* ---- A9 + A11: PPODiffusion.loss fused with the minibatch gather, forward AND backward ----- */
/* (model/diffusion/diffusion_ppo.py:57-199; agent/finetune/train_ppo_diffusion_agent.py:316-327)
 * Rollout buffer, R = n_steps*n_envs rows: obs_k (R,cond), chains_k (R,Kft+1,Ta*Da), returns_k,
 * values_k, adv_k (R,), logprobs_k (R,Kft,Ta*Da).  inds (N,) int64 in [0, R*Kft): sample n is
 * (row = ind / Kft, k = ind % Kft); kinds must be NULL.
 * Pre-gathered mode (the reference's loss() signature): inds == NULL, kinds (N,) int64 = denoising_inds,
 * obs_k (N,cond), chains_k (N,2,Ta*Da) = (chains_prev, chains_next), returns_k/values_k/adv_k (N,),
 * logprobs_k (N,Ta*Da).
 * Data parallel: global_moments (3 doubles: sum adv, sum adv^2, sample count over the GLOBAL minibatch,
 * all-reduced by the caller) or NULL for a single-process minibatch.  With it, advantages are normalised
 * by the global mean/std and every mean uses the global count, so SUM-all-reducing the gradients and
 * statistics of all ranks reproduces the single-process result.
 * Writes d(pg_loss)/d(actor params) to actor_grad and
 * d(v_loss)/d(critic params) to critic_grad (flat layouts, overwritten), and the statistics
 * (means over the minibatch) to stats[DPPO_STAT_*]. */
int64_t dppo_ppo_workspace_bytes(const dppo_net_desc* actor, const dppo_net_desc* critic, int prec, int64_t N);
int dppo_ppo_loss_fwd_bwd(const dppo_net_desc* actor, const dppo_net_desc* critic, int prec,
                          const float* actor_params, const void* actor_packed, const float* critic_params,
                          const void* critic_packed, const dppo_diffusion_cfg* dcfg, const dppo_ppo_cfg* pcfg,
                          const dppo_step* ksteps, const float* obs_k, const float* chains_k,
                          const float* returns_k, const float* values_k, const float* adv_k,
                          const float* logprobs_k, const int64_t* inds, const int64_t* kinds, int64_t N,
                          const double* global_moments, float* actor_grad, float* critic_grad, double* stats,
                          void* workspace, int64_t workspace_bytes, dppo_stream_t stream);

/* Data parallel (the reference is single-process, so this entry replaces nothing): the same call with a hook.
 * `critic_grads_enqueued(user, side)` is called ONCE, on the calling thread, from inside the call, at the point where
 * everything that writes `critic_grad` has been enqueued on `side` -- the library's critic stream, or `stream` itself when
 * the critic pipeline runs on the caller's stream (knob 2 = 0) -- and before the actor's forward / backward are enqueued.
 * Work the hook enqueues on `side` (the critic slice of the gradient all-reduce: torch.distributed under
 * torch.cuda.ExternalStream(side), or ncclAllReduce) therefore runs while the actor half still occupies `stream`, and is
 * joined into `stream` with the rest of the critic pipeline before the call's work on `stream` ends.  The hook must not
 * synchronise with `stream` (deadlock: the join comes later) and cannot be used under stream capture.  hook == NULL or a NULL
 * function pointer: exactly dppo_ppo_loss_fwd_bwd. */
typedef struct dppo_dp_hook {
  void (*critic_grads_enqueued)(void* user, dppo_stream_t side);
  void* user;
} dppo_dp_hook;
int dppo_ppo_loss_fwd_bwd_dp(const dppo_net_desc* actor, const dppo_net_desc* critic, int prec, const float* actor_params,
                             const void* actor_packed, const float* critic_params, const void* critic_packed,
                             const dppo_diffusion_cfg* dcfg, const dppo_ppo_cfg* pcfg, const dppo_step* ksteps,
                             const float* obs_k, const float* chains_k, const float* returns_k, const float* values_k,
                             const float* adv_k, const float* logprobs_k, const int64_t* inds, const int64_t* kinds,
                             int64_t N, const double* global_moments, float* actor_grad, float* critic_grad,
                             double* stats, void* workspace, int64_t workspace_bytes, dppo_stream_t stream,
                             const dppo_dp_hook* hook);

/* ---- Gaussian-policy PPO on the same trunk kernels --------------------------------------------
 * Replaces model/common/mlp_gaussian.py:283-362 (Gaussian_MLP.forward, fixed / learned-per-dimension std),
 * model/common/gaussian.py:63-121 (GaussianModel.forward_train / forward), model/rl/gaussian_vpg.py:46-62
 * (get_logprobs) and model/rl/gaussian_ppo.py:39-128 (PPO_Gaussian.loss).  The actor is a trunk on the flattened
 * observation: a kind-1 descriptor with out_dim = Ta*Da (a critic is the out_dim = 1 case). */
typedef struct dppo_gaussian_cfg {
  int32_t horizon_steps, action_dim;
  int32_t tanh_mean;     /* Gaussian_MLP.tanh_output: mean = tanh(trunk output)                                 */
  int32_t std_mode;      /* 0: fixed_std ; 1: sigma_j = exp(0.5 clamp(logvar[j], logvar_min, logvar_max)), j < Da */
  int32_t norm_adv, has_vclip;
  int32_t deterministic; /* forward_train(deterministic=True): sigma = 1e-4                                      */
  int32_t pad;
  float fixed_std, logvar_min, logvar_max, randn_clip;
  double clip_ploss_coef, clip_vloss_coef;
  uint32_t seed_lo, seed_hi; /* dppo_gaussian_sample with noise == NULL: key of the in-kernel Philox generator   */
} dppo_gaussian_cfg;
#define DPPO_GAUSS_STAT_ENTROPY 7 /* dist.entropy().mean()                                       */
#define DPPO_GAUSS_STAT_STD 8     /* dist.scale.mean()                                           */
#define DPPO_GAUSS_STAT_COUNT 9   /* stats[0..6] as DPPO_STAT_*                                  */
int64_t dppo_gaussian_workspace_bytes(const dppo_net_desc* actor, const dppo_net_desc* critic, int prec, int64_t N);
/* actions (B,Ta*Da) = mean + sigma * clamp(z, +-randn_clip); z = noise (B,Ta*Da) or, if NULL, drawn in the kernel.
 * mean_out (B,Ta*Da) optional. */
int dppo_gaussian_sample(const dppo_net_desc* actor, int prec, const float* params, const void* packed,
                         const dppo_gaussian_cfg* cfg, const float* logvar, const float* obs, const float* noise,
                         int64_t B, float* actions, float* mean_out, void* workspace, int64_t workspace_bytes,
                         dppo_stream_t stream);
/* logp (N,) = mean over Ta*Da of log N(actions; mean, sigma) */
int dppo_gaussian_logprob(const dppo_net_desc* actor, int prec, const float* params, const void* packed,
                          const dppo_gaussian_cfg* cfg, const float* logvar, const float* obs, const float* actions,
                          int64_t N, float* logp, void* workspace, int64_t workspace_bytes, dppo_stream_t stream);
/* obs (N,cond), actions (N,Ta*Da), returns / oldvalues / adv / oldlogp (N,).  global_moments as in
 * dppo_ppo_loss_fwd_bwd.  Writes d pg_loss / d actor params, d v_loss / d critic params, d pg_loss / d logvar (Da; only
 * std_mode 1, may be NULL otherwise) and stats[DPPO_GAUSS_STAT_COUNT]. */
int dppo_gaussian_ppo_loss_fwd_bwd(const dppo_net_desc* actor, const dppo_net_desc* critic, int prec,
                                   const float* actor_params, const void* actor_packed, const float* critic_params,
                                   const void* critic_packed, const dppo_gaussian_cfg* cfg, const float* logvar,
                                   const float* obs, const float* actions, const float* returns,
                                   const float* oldvalues, const float* adv, const float* oldlogp, int64_t N,
                                   const double* global_moments, float* actor_grad, float* critic_grad,
                                   float* logvar_grad, double* stats, void* workspace, int64_t workspace_bytes,
                                   dppo_stream_t stream);

/* ---- conv denoiser (Unet1D), inference side ------------------------------------------------------
 * Replaces model/diffusion/unet.py:27-327 (ResidualBlock1D, Unet1D.forward), model/diffusion/modules.py:28-95
 * (Downsample1d, Upsample1d, Conv1dBlock) as the actor of the same K-step sampler and log-prob evaluation
 * (diffusion_vpg.py:227-396).  Every convolution is an MFMA GEMM over a channel-last, time-padded activation image
 * (the im2col row of (b, t) is a contiguous window of it); GroupNorm + activation + FiLM / residual are one epilogue
 * kernel per block half.  Parameters: one flat fp32 buffer in the reference's state-dict order.
 * The training side (dppo_unet_ppo_loss_fwd_bwd, dppo_unet_denoise_mse_fwd_bwd) re-runs the forward keeping every block's
 * conv outputs and images, and back-propagates with the same two tricks: data gradients are convolutions of the padded
 * gradient image with the flipped kernel, weight gradients are gemm_tn over the forward's (overlapping-row) operands.
 * Not built: cond_mlp_dims (no shipped cfg sets it). */
typedef struct dppo_unet_desc {
  int32_t action_dim, cond_dim, horizon_steps;
  int32_t time_dim;           /* diffusion_step_embed_dim                                        */
  int32_t dim, n_levels;      /* channels of level i = dim * mults[i]                              */
  int32_t mults[4];
  int32_t kernel_size, n_groups;
  int32_t larger_encoder;     /* cond_mlp_dims is None and not smaller_encoder (unet.py:151)       */
  int32_t cond_predict_scale;
  int32_t act;                /* DPPO_ACT_*                                                        */
  float groupnorm_eps;
} dppo_unet_desc;
int64_t dppo_unet_param_count(const dppo_unet_desc* net);
int64_t dppo_unet_packed_bytes(const dppo_unet_desc* net, int prec, int n_time);
int dppo_unet_pack(const dppo_unet_desc* net, int prec, int n_time, const float* params, void* packed,
                   dppo_stream_t stream);
int64_t dppo_unet_workspace_bytes(const dppo_unet_desc* net, int prec, int64_t rows);
/* Workspace of one dppo_unet_sample_chain call with room for the FiLM tables of all its steps: the cond encoders of every
 * block see (t_k, obs) only, so with this much workspace they are evaluated once up front (all steps in one pass per network)
 * instead of inside every denoising step.  With the smaller dppo_unet_workspace_bytes(B) the call still works, encoders in place. */
int64_t dppo_unet_sample_workspace_bytes(const dppo_unet_desc* net, int prec, int64_t B, int n_steps);
/* eps (rows,Ta,Da) = Unet1D(x (rows,Ta,Da), t (rows,) int64, state (rows,cond)) */
int dppo_unet_forward(const dppo_unet_desc* net, int prec, const float* params, const void* packed, const float* x,
                      const int64_t* t, const float* state, int64_t rows, float* eps, void* workspace,
                      int64_t workspace_bytes, dppo_stream_t stream);
/* dppo_sample_chain with the conv denoiser.  sched_host: the n_steps dppo_step entries in HOST memory (the loop over
 * steps is on the host: one forward of a few dozen launches + one posterior / noise launch per step; capturable). */
int dppo_unet_sample_chain(const dppo_unet_desc* net, int prec, const float* params_base, const void* packed_base,
                           const float* params_ft, const void* packed_ft, const dppo_diffusion_cfg* cfg,
                           const dppo_step* sched_host, int n_steps, const float* obs, const float* noise, int64_t B,
                           float* traj, float* chains, int chain_len, int init_slot, void* workspace,
                           int64_t workspace_bytes, dppo_stream_t stream);
/* dppo_ppo_loss_fwd_bwd with a conv actor (same arguments, semantics and statistics; the critic is the MLP critic) */
int64_t dppo_unet_ppo_workspace_bytes(const dppo_unet_desc* actor, const dppo_net_desc* critic, int prec, int64_t N);
int dppo_unet_ppo_loss_fwd_bwd(const dppo_unet_desc* actor, const dppo_net_desc* critic, int prec, const float* actor_params,
                               const void* actor_packed, const float* critic_params, const void* critic_packed,
                               const dppo_diffusion_cfg* dcfg, const dppo_ppo_cfg* pcfg, const dppo_step* ksteps,
                               const float* obs_k, const float* chains_k, const float* returns_k, const float* values_k,
                               const float* adv_k, const float* logprobs_k, const int64_t* inds, const int64_t* kinds,
                               int64_t N, const double* global_moments, float* actor_grad, float* critic_grad,
                               double* stats, void* workspace, int64_t workspace_bytes, dppo_stream_t stream);
/* dppo_denoise_mse_fwd_bwd with a conv denoiser */
int64_t dppo_unet_denoise_mse_workspace_bytes(const dppo_unet_desc* net, int prec, int64_t N);
int dppo_unet_denoise_mse_fwd_bwd(const dppo_unet_desc* net, int prec, const float* params, const void* packed,
                                  const dppo_step* tsteps, int n_time, const float* obs, const float* pairs,
                                  const int64_t* kinds, int64_t N, float* grad, double* loss, void* workspace,
                                  int64_t workspace_bytes, dppo_stream_t stream);
/* dppo_chain_logprob with the conv denoiser: ksteps (device) / ksteps_host (host) hold the same Kft entries. */
int dppo_unet_chain_logprob(const dppo_unet_desc* net, int prec, const float* params, const void* packed,
                            const dppo_diffusion_cfg* cfg, const dppo_step* ksteps, const dppo_step* ksteps_host,
                            int Kft, const float* obs, const float* chains, int64_t B, float* logp, void* workspace,
                            int64_t workspace_bytes, dppo_stream_t stream);

/* ---- A12: optimiser (torch.optim.AdamW + clip_grad_norm_ semantics) ----------------------- */
/* out[0] = sum g^2 (float64), deterministic two-stage reduction; scratch >= 1024 doubles */
int dppo_grad_sq_norm(const float* grad, int64_t n, double* scratch, double* out, dppo_stream_t stream);
/* One AdamW step on a flat buffer.  If sq_norm != NULL the gradient is first scaled by
 * min(1, max_norm / (sqrt(*sq_norm) + 1e-6)).  `step` counts from 1. */
int dppo_adamw_step(float* params, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, int step,
                    double lr, double beta1, double beta2, double eps, double weight_decay, const double* sq_norm,
                    double max_norm, dppo_stream_t stream);
/* The same with the step count and the learning rate in device memory (step_dev[0] = steps taken so far, incremented by
 * the call; lr_dev[0]): nothing in the argument list changes from step to step, so a captured hipGraph of the update can
 * be replayed. */
int dppo_adamw_step_dev(float* params, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, int32_t* step_dev,
                        const float* lr_dev, double beta1, double beta2, double eps, double weight_decay,
                        const double* sq_norm, double max_norm, dppo_stream_t stream);

/* The same for up to four parameter vectors in ONE launch (actor_ft and critic: the optimiser tail of an update is a
 * chain of latency-bound launches).  step_dev points at int32[2] = {steps taken, 0}: the slot's last workgroup to finish
 * advances [0], so there is no separate tick launch; sq_norm NULL = no clipping. */
typedef struct dppo_adamw_slot {
  float* params;
  const float* grad;
  float* exp_avg;
  float* exp_avg_sq;
  int64_t n;
  int32_t* step_dev;
  const float* lr_dev;
  double beta1, beta2, eps, weight_decay;
  const double* sq_norm;
  double max_norm;
} dppo_adamw_slot;
int dppo_adamw_step_multi(const dppo_adamw_slot* slots, int n_slots, dppo_stream_t stream);

/* Data parallel: the float64 statistics ride in the fp32 gradient bucket of the one all-reduce per optimiser step
 *  as (hi, lo) float32 pairs.  split: hi_lo[0..8) = (float)stats, hi_lo[8..16) = the remainder; merge: back
 * to float64 after the SUM-reduce, with the two advantage statistics (global values every rank wrote) divided by
 * `world`.  One launch each way. */
int dppo_stats_split(const double* stats, float* hi_lo, dppo_stream_t stream);
int dppo_stats_merge(const float* hi_lo, double* stats, int world, dppo_stream_t stream);
/* The same with the number of rank-global slots from DPPO_STAT_ADV_MEAN on (values every rank wrote in full, divided by
 * `world` after the SUM) chosen by the caller: 2 for the diffusion / mixture losses, 3 for the Gaussian head, whose
 * DPPO_GAUSS_STAT_ENTROPY (slot 7) depends on sigma only and is the same on every rank (mean std, slot 8, never travels). */
int dppo_stats_merge_n(const float* hi_lo, double* stats, int world, int n_avg, dppo_stream_t stream);

/* ---- measurement hook (bench.py only; process-wide, not thread-safe, off by default) ----------- */
/* While armed for a kernel, each of its launches is bracketed by HIP events on the launch stream.
 * kernel_id: 1 = gemm_nt on an H x H layer (layered path), 2 = gemm_tn weight gradient (N1,N2 >= 128),
 * 3 = fused row-tile forward, 4 = fused row-tile backward, 5 = K-step sampler.  dppo_probe_collect waits for the
 * events, returns the summed kernel time, the launch count and the summed ALGORITHMIC FLOPs (2 x MACs of the true,
 * unpadded shapes), and disarms. */
int dppo_probe_arm(int kernel_id, int max_launches);
int dppo_probe_collect(double* total_ms_host, int* launches_host, double* flops_host);
/* same, plus the algorithmic bytes of the timed launches where the library knows them (kernel id 2 with knob 12: both
 * operands of every GEMM of the group once + its fp32 result), else 0 */
int dppo_probe_collect_bytes(double* total_ms_host, int* launches_host, double* flops_host, double* bytes_host);

/* ---- tuning / micro-benchmark hooks (tools/ and tests only; never used by the product path) ----- */
/* knob 0: gemm_nt operand staging, 0 = through registers, 1 = global_load_lds (LDS-DMA, default)
 * knob 1: big-batch MLP path, 1 = fused row-tile kernels (default), 0 = layer-by-layer gemm_nt chain
 * knob 2: PPO update, 1 = critic half on a side stream, overlapping the actor half (default), 0 = one stream
 * knob 3 / 4: weight-gradient GEMMs, workgroups aimed for (default 256) / cap on the row splits (default 128)
 * knob 7: fused kernels, bit 0 / 1 = 32-row tiles at two workgroups per CU in the backward / forward at H = 512,
 *         bit 2 / 3 = 64- / 32-row tiles at two workgroups per CU in the forward / backward at H = 256 (default 0: none
 *         of them pays)
 * knob 8: timing experiments (results are wrong while set): bit 0 / 1 the fused backward without its gradient stores / derivative
 *         fetch, bit 2 / 3 / 4 the reduction launch behind the weight-gradient GEMMs without its slab reductions / bias sums / loss
 *         statistics (tools/tail_reduce_parts.sh); knob 9: side streams at low priority
 * knob 5: weight-gradient GEMM kernel, 0 = register-staged (default), 1..8 = an LDS-DMA ring configuration, -1 = by shape
 * knob 6: thin (512 x 64) weight-gradient tiles on / off; knob 10: critic side stream gated on the actor's forward (0 off)
 * knob 11: time-embedding gradient from a one-hot of the denoising step in the K padding of the actor's input rows, so
 *          the first layer's weight-gradient GEMM also yields the per-step sums of dh0 (default 1); 0 = separate
 *          gemm_nt + segmented sum on the tail stream
 * knob 12: all weight-gradient GEMMs of one backward pass in one grouped launch of 128 x 128 tiles (default 1); 0 = one
 *          launch per GEMM (thin outputs then use the 512 x 64 tile of knob 6)
 * knob 13: dppo_pack_net writes all images of a network in one launch (default 1); 0 = one launch per image
 * knob 14: side streams of the update joined right behind the weight-gradient launch (default 1); 0 = at the call's end
 * knob 15: sampler keeps the layer-0 weight fragments of the network in use in LDS when they fit (default 1)
 * knob 16: the top block's second-layer weight gradient from the rank-out_dim factorisation
 *          dW2 = Wout^T . (d_out^T . act(z1)) instead of an H x H contraction over the batch (default 1)
 * knob 17: the sampler never runs the top block's second layer: out = Wout . h_in + (Wout . W2) . act(z1) + const
 *          (default 1)
 * knob 18: what follows the slab reduction of the actor's backward (low-rank dW2, time-embedding gradient) in one launch
 *          (default 1)
 * knob 22: fused forward of one-block networks (no LayerNorm; out_dim <= 16 at hidden <= 512, or 17-64 outputs at hidden 512
 *          with in_dim <= 96 for bf16 / 48 for fp32, where the Wout W2 fragments ride the weight ring) never runs the
 *          block's second layer: out = (Wout W0) x + (Wout W2) act(z1) + const, and the out-layer weight gradient is
 *          rebuilt from d_out^T x and d_out^T act(z1).  1 (default): on; 0: off; 3: heads of up to 16 outputs only
 * knob 23: fused backward of one-block networks (no LayerNorm, hidden <= 512, low-rank dW2 on): dh_1 = d_out . Wout is added
 *          last, into the W1^T layer's accumulators, instead of carried in registers; forward-sized tiles (default 1)
 * knob 25: the one-block kernels walk their short layers (K = in_dim on the input tile, K = out_dim on the d_out tile)
 *          as the 1-2 k-steps that hold data instead of the 4 the weight stream pads them to (default 1)
 * knob 26: LDS stages of the grouped weight-gradient GEMM: 1 (36.9 KB per workgroup, three workgroups per CU), 2 (double-buffered,
 *          two per CU), or 0 (default): two for a group of more than 768 workgroups over at least 32,768 samples, else one
 * knob 27: sampler, small env batches of one-block bf16 networks at hidden 512: one 16-row tile over eight workgroups with
 *          the weights resident in registers (default 1; see dppo_sample_chain_workspace_bytes) or over one (0)
 * knob 28: knob 27's kernel: 64-cycle sleep periods between a workgroup's exchange store and its first sweep (default 4)
 * knob 31: weight-gradient GEMMs of one-block bf16 networks (merged forward + one-block backward + one-hot time columns, no
 *          cond_mlp, PPO update without d loss / d obs): act(h_0), act(z1), dz1, dh_0 are written by the fused kernels as
 *          K-major MFMA operand fragments and contracted without any transpose (1), or row-major through the transposing
 *          LDS kernel (0, default: the fragment form is parity-green but not faster yet)
 * knob 32: 0 (default): the fragments of a workgroup's tile are fetched once per k-step by LDS-DMA into a four-stage ring and
 *          read back by its four waves; 2..4: every wave loads its own fragments into registers, that many k-steps ahead
 * knob 36: the minibatch's advantage moments (PPO update, single rank): partial sums by the last 64 blocks of the row builder's
 *          launch, added by every block of the loss kernel (1: always; 0, default: for minibatches of at most 16,384 samples, where
 *          the step is a chain of launches), or a launch of their own between the rows and the actor's forward (2: always; at
 *          50,000 samples it already overlaps the critic's forward)
 * knob 37: one-block bf16 networks whose informative input columns fit 32 (a denoiser: action chunk + observation + Kft - 1 one-hot
 *          step columns; a critic: its observation): the first layer's weight gradient is accumulated inside the fused backward
 *          kernel, per persistent workgroup in LDS, and d loss / d h_0 is neither stored nor read back (1, default); 0: dh_0 is
 *          stored and contracted by the weight-gradient GEMM launch
 * knob 38: with knob 37, a denoiser's backward: the reductions that the backward kernel alone feeds (its first-layer slabs, the bias
 *          sums, the loss statistics) and the time-embedding gradient behind them run on the library's second side stream under
 *          the weight-gradient GEMM launch (1, default) or behind it with everything else (0)
 * knob 40: knob 38's work as extra workgroups in front of the tiles of the actor's weight-gradient GEMM launch, the time-embedding
 *          blocks waiting for the reductions on an arrival counter (1: no side stream, no event, no launch -- measured slower, the
 *          riders' dependent chain runs under the GEMM's memory load and the launch cannot end before it) or on the side stream
 *          (0, default)
 * knob 41: with knob 38: the slab reductions of the actor's weight-gradient GEMMs and what depends on the two thin products among
 *          them (low-rank dW2, dWout, db2) are one launch -- the dependent workgroups poll the thin reductions' arrival -- (1,
 *          default) or two launches (0)
 * knob 39: PPO update of a bf16 one-block actor at hidden 512 with a head of at most 16 outputs, actor and critic on two streams:
 *          the policy half of the loss (log-probs, ratio, clipped surrogate, d loss / d eps, statistics) runs in the epilogue
 *          of the actor's fused forward kernel (1) or as a launch of its own between forward and backward (0, default: the fused
 *          variant's register footprint costs the overlapped step more than the launch it saves)
 * knob 35: what follows the weight-gradient GEMMs of a backward pass -- slab sums, bias column sums, loss statistics -- inside
 *          the GEMM launch (1: the last workgroup at an output tile sums its slabs, the small reductions ride as extra
 *          workgroups; measured slower, 200 vs 107 + 32 us) or as a launch of its own (0, default)
 * knob 33: k-steps (of 32 batch rows) the fragment GEMM's L2 prefetch runs ahead of its ring loads (default 12)
 * knob 30: minibatch rows per output column from which the top block's weight gradient is taken low-rank (knob 16) and the
 *          one-block backward (knob 23) runs: M >= value x out_dim (default 100)
 * knob 29: knob 27's kernel: sweeps a workgroup waits for its tile before it gives up (default 2^20; tests force a time-out
 *          with 1; <= 0 restores the default) */
int dppo_tune_set(int knob, int value);
/* one bare layer GEMM: out[M][ldo] = act(X[M][Kp] . W[N][Kp]^T + bias) with elem = prec operands;
 * out_f32 and/or out_elem may be NULL; ldo >= round_up(N,16) */
int dppo_gemm_nt_raw(int prec, const void* X, const void* W, const float* bias, int64_t M, int N, int Kp,
                     float* out_f32, void* out_elem, int ldo, int act, dppo_stream_t stream);
/* one bare weight-gradient GEMM: C[N1][N2] = A[M][N1]^T . B[M][N2] (elem operands, leading dims lda/ldb);
 * slab: scratch of at least splits*N1*N2 floats, splits = ceil(M / rows_per_split) */
int dppo_gemm_tn_raw(int prec, const void* A, int lda, int N1, const void* B, int ldb, int N2, int64_t M,
                     int rows_per_split, float* slab, float* C, dppo_stream_t stream);

/* ---- 8f row 4 (second half): mixture-of-Gaussians policy PPO ---------------------------------------------------------
 * Replaces model/common/mlp_gmm.py:11-110 (GMM_MLP.forward: component means tanh(mlp_mean(s)) (B, modes, Ta*Da), fixed or
 * learned per-(mode, action dim) std, mixture logits mlp_weights(s)), model/common/gmm.py:48-97 (GMMModel.forward_train /
 * forward: MixtureSameFamily(Categorical(logits), Independent(Normal, 1))), model/rl/gmm_vpg.py:33-43 (get_logprobs) and
 * model/rl/gmm_ppo.py:39-112 (PPO_GMM.loss).  The actor is TWO trunks on the observation (kind-1 descriptors: out_dim =
 * num_modes*Ta*Da and num_modes), each with its own flat parameter buffer; out_dim above 128 runs on the layered GEMM path.
 * log p(a) = logsumexp_m(log pi_m + sum_j log N(a_j; mu_mj, sigma_mj)) -- summed over Ta*Da, as the reference's Independent. */
typedef struct dppo_gmm_cfg {
  int32_t horizon_steps, action_dim, num_modes;
  int32_t std_mode;      /* 0: fixed_std ; 1: sigma_md = exp(0.5 clamp(logvar[m*Da + d], logvar_min, logvar_max))       */
  int32_t norm_adv, has_vclip;
  int32_t deterministic; /* forward_train(deterministic=True): sigma = 1e-4 (the component is still drawn)               */
  int32_t pad;
  float fixed_std, logvar_min, logvar_max;
  float ent_coef;        /* the returned actor-side gradients are those of pg_loss + ent_coef * entropy_loss             */
  double clip_ploss_coef, clip_vloss_coef;
  uint32_t seed_lo, seed_hi;
} dppo_gmm_cfg;
int64_t dppo_gmm_workspace_bytes(const dppo_net_desc* mean, const dppo_net_desc* weights, const dppo_net_desc* critic, int prec,
                                 int64_t N);
/* actions (B,Ta*Da) = mu_k + sigma_k z with k = modes[b] (or, if NULL, drawn from softmax(logits) in the kernel) and z = noise
 * (B,Ta*Da) (or drawn in the kernel). */
int dppo_gmm_sample(const dppo_net_desc* mean, const dppo_net_desc* weights, int prec, const float* mean_params,
                    const void* mean_packed, const float* weights_params, const void* weights_packed, const dppo_gmm_cfg* cfg,
                    const float* logvar, const float* obs, const int64_t* modes, const float* noise, int64_t B, float* actions,
                    void* workspace, int64_t workspace_bytes, dppo_stream_t stream);
int dppo_gmm_logprob(const dppo_net_desc* mean, const dppo_net_desc* weights, int prec, const float* mean_params,
                     const void* mean_packed, const float* weights_params, const void* weights_packed, const dppo_gmm_cfg* cfg,
                     const float* logvar, const float* obs, const float* actions, int64_t N, float* logp, void* workspace,
                     int64_t workspace_bytes, dppo_stream_t stream);
/* stats as dppo_gaussian_ppo_loss_fwd_bwd (entropy = mean_b sum_m pi_m H_m, std = mean_b sum_m pi_m mean_j sigma_mj).
 * mean_grad / weights_grad / logvar_grad (num_modes*Da) <- d (pg_loss + ent_coef * entropy_loss), critic_grad <- d v_loss. */
int dppo_gmm_ppo_loss_fwd_bwd(const dppo_net_desc* mean, const dppo_net_desc* weights, const dppo_net_desc* critic, int prec,
                              const float* mean_params, const void* mean_packed, const float* weights_params,
                              const void* weights_packed, const float* critic_params, const void* critic_packed,
                              const dppo_gmm_cfg* cfg, const float* logvar, const float* obs, const float* actions,
                              const float* returns, const float* oldvalues, const float* adv, const float* oldlogp, int64_t N,
                              const double* global_moments, float* mean_grad, float* weights_grad, float* critic_grad,
                              float* logvar_grad, double* stats, void* workspace, int64_t workspace_bytes, dppo_stream_t stream);

/* ---- loss entries that also return d loss / d observation (a visual encoder sits in front of the trunk) ------------
 * Same arguments as the entry without the suffix, in pre-gathered mode (one observation row per sample, `kinds` given), plus:
 *   obs_critic   (N, critic cond_dim) or NULL: the critic's own observation rows -- ViTCritic encodes the images with its own
 *                backbone (model/common/critic.py:177-205), so actor and critic see different vectors;
 *   d_obs_actor  (N, actor cond_dim)  or NULL: <- d pg_loss / d obs_k ;
 *   d_obs_critic (N, critic cond_dim) or NULL: <- d v_loss / d obs_critic (or obs_k).
 * d_obs of the supervised loss: (N, cond_dim) <- d loss / d obs.  Not built for actors with a cond_mlp. */
typedef struct dppo_obs_io {
  const float* obs_critic;
  float* d_obs_actor;
  float* d_obs_critic;
} dppo_obs_io;
int dppo_ppo_loss_fwd_bwd_obs(const dppo_net_desc* actor, const dppo_net_desc* critic, int prec, const float* actor_params,
                              const void* actor_packed, const float* critic_params, const void* critic_packed,
                              const dppo_diffusion_cfg* dcfg, const dppo_ppo_cfg* pcfg, const dppo_step* ksteps,
                              const float* obs_k, const float* chains_k, const float* returns_k, const float* values_k,
                              const float* adv_k, const float* logprobs_k, const int64_t* kinds, int64_t N,
                              const double* global_moments, float* actor_grad, float* critic_grad, double* stats,
                              void* workspace, int64_t workspace_bytes, dppo_stream_t stream, const dppo_obs_io* io);
int dppo_unet_ppo_loss_fwd_bwd_obs(const dppo_unet_desc* actor, const dppo_net_desc* critic, int prec, const float* actor_params,
                                   const void* actor_packed, const float* critic_params, const void* critic_packed,
                                   const dppo_diffusion_cfg* dcfg, const dppo_ppo_cfg* pcfg, const dppo_step* ksteps,
                                   const float* obs_k, const float* chains_k, const float* returns_k, const float* values_k,
                                   const float* adv_k, const float* logprobs_k, const int64_t* kinds, int64_t N,
                                   const double* global_moments, float* actor_grad, float* critic_grad, double* stats,
                                   void* workspace, int64_t workspace_bytes, dppo_stream_t stream, const dppo_obs_io* io);
int dppo_gaussian_ppo_loss_fwd_bwd_obs(const dppo_net_desc* actor, const dppo_net_desc* critic, int prec,
                                       const float* actor_params, const void* actor_packed, const float* critic_params,
                                       const void* critic_packed, const dppo_gaussian_cfg* cfg, const float* logvar,
                                       const float* obs, const float* actions, const float* returns,
                                       const float* oldvalues, const float* adv, const float* oldlogp, int64_t N,
                                       const double* global_moments, float* actor_grad, float* critic_grad,
                                       float* logvar_grad, double* stats, void* workspace, int64_t workspace_bytes,
                                       dppo_stream_t stream, const dppo_obs_io* io);
int dppo_denoise_mse_fwd_bwd_obs(const dppo_net_desc* actor, int prec, const float* params, const void* packed,
                                 const dppo_step* tsteps, int n_time, const float* obs, const float* pairs,
                                 const int64_t* kinds, int64_t N, float* grad, double* loss, void* workspace,
                                 int64_t workspace_bytes, dppo_stream_t stream, float* d_obs);
int dppo_unet_denoise_mse_fwd_bwd_obs(const dppo_unet_desc* net, int prec, const float* params, const void* packed,
                                      const dppo_step* tsteps, int n_time, const float* obs, const float* pairs,
                                      const int64_t* kinds, int64_t N, float* grad, double* loss, void* workspace,
                                      int64_t workspace_bytes, dppo_stream_t stream, float* d_obs);

/* ---- 8f row 2 (pixel observations): ViT patch encoder + SpatialEmb ------------------------------------------------
 * Replaces model/common/vit.py:28-62 (VitEncoder: PatchEmbed2 = Conv2d(k8,s4) -> ReLU -> Conv2d(k3,s2), + pos_embed,
 * `depth` pre-norm TransformerLayers, final LayerNorm) and model/common/modules.py:10-41 (SpatialEmb), assembled as
 * VisionDiffusionMLP / VisionUnet1D / ViTCritic do (mlp_diffusion.py:120-160, unet.py:548-581, critic.py:177-205): the
 * encoder turns cond = {rgb, state} into the observation vector cat[feat, state] of width spatial_emb * num_img + prop_dim,
 * and the trunk behind it is the state-observation network on that vector (dppo_sample_chain, dppo_chain_logprob,
 * dppo_unet_*, dppo_critic_forward take it as `obs`).  The feature does not depend on the denoising step: one encode per
 * observation serves the whole K-step chain.  Training: dppo_vis_encode(train = 1) keeps its tape in the workspace, the
 * *_obs loss entries below return d loss / d obs, dppo_vis_backward (same workspace) turns it into the encoder's gradients.
 * Flat parameter order = the reference's state dict: backbone.vit.{pos_embed, patch_embed.embed.0, .3, net.l.{layer_norm1,
 * mha.qkv_proj, mha.out_proj, layer_norm2, linear1, linear2}, norm}, then compress (or compress1, compress2).{weight,
 * input_proj.0, input_proj.1}.  Not built: embed_style "embed1", embed_norm, the Linear `compress` of spatial_emb = 0,
 * RandomShiftsAug inside the network (the host-side augmentation of dppo_amd shifts the images before the call). */
typedef struct dppo_vis_desc {
  int32_t in_ch;           /* 3 * img_cond_steps, per camera                                  */
  int32_t img_h, img_w;
  int32_t embed_dim, num_heads, depth;
  int32_t embed_norm;      /* must be 0                                                        */
  int32_t prop_dim;        /* To*Do of the low-dimensional state                               */
  int32_t spatial_emb;     /* SpatialEmb proj_dim                                              */
  int32_t num_img;         /* cameras: 1, or 2 sharing the backbone with one SpatialEmb each   */
} dppo_vis_desc;
int64_t dppo_vis_param_count(const dppo_vis_desc* net);
int64_t dppo_vis_packed_bytes(const dppo_vis_desc* net, int prec);
int dppo_vis_pack(const dppo_vis_desc* net, int prec, const float* params, void* packed, dppo_stream_t stream);
int64_t dppo_vis_workspace_bytes(const dppo_vis_desc* net, int prec, int64_t B, int train);
/* rgb: the reference's cond["rgb"], (B, img_cond_steps, 3 * num_img, H, W), fp32 or (rgb_u8 != 0) uint8, values 0..255;
 * state (B, prop_dim); obs (B, ld_obs) <- cat[feat, state]. */
int dppo_vis_encode(const dppo_vis_desc* net, int prec, const float* params, const void* packed, const void* rgb, int rgb_u8,
                    const float* state, int64_t B, float* obs, int ld_obs, int train, void* workspace,
                    int64_t workspace_bytes, dppo_stream_t stream);
/* d_obs (B, ld_dobs): d loss / d obs (its first spatial_emb * num_img columns are read).  grad: flat, OVERWRITTEN.
 * `workspace` must be the one the matching dppo_vis_encode(train = 1) call on the same B wrote. */
int dppo_vis_backward(const dppo_vis_desc* net, int prec, const float* params, const void* packed, const float* d_obs,
                      int ld_dobs, int64_t B, float* grad, void* workspace, int64_t workspace_bytes, dppo_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DPPO_HIP_H */
