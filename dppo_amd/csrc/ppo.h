// Element-wise / reduction kernels of the DPPO path: minibatch gather, fused PPO loss (forward
// statistics + d loss / d eps), Gaussian log-prob epilogue, time-embedding table forward/backward,
// GAE scan, AdamW, parameter packing for the GEMM path.
#pragma once
#include "common.h"
#include "fused.h"
#include "../../include/dppo_hip.h"

namespace dppo {

// ---- packing ------------------------------------------------------------------------------------
// dst[r][c] (ldd, elem) = c < cols ? src[r][c] (lds, f32) : 0     for r < rows, c < ldd
template <class P>
void launch_cast_pad(const float* src, int rows, int cols, int lds, void* dst, int ldd, hipStream_t s);
// dst[c][r] (ldd, elem) = src[r][coff + c] for r < rows, c < cols ; zero for rows <= r < ldd
template <class P>
void launch_transpose_cast(const float* src, int rows, int cols, int lds, int coff, void* dst, int ldd,
                           hipStream_t s);
// out[r][c] (dense, ncols wide) = src[r][col0 + c]
void launch_copy_cols(const float* src, int ld, int col0, int ncols, int64_t rows, float* out, hipStream_t s);
// time-embedding table: temb[t][td] for t < n_time (model/diffusion/mlp_diffusion.py:191-196,
// modules.py:20-27); hid[t][2td] keeps the pre-Mish activations for the backward
void launch_time_table(const float* w1, const float* b1, const float* w2, const float* b2, int td, int n_time,
                       float* temb, hipStream_t s);

// ---- row building -------------------------------------------------------------------------------
struct BuildRows {
  // rollout mode (kinds == null): sample n: ind = inds ? inds[n] : n ; b = ind / Kft ; k = ind % Kft ;
  //   chains [R][Kft+1][AF].   gathered mode (kinds != null): b = n, k = kinds[n], chains [M][2][AF] = (x_k, x_k+1)
  const int64_t* inds;
  const int64_t* kinds;
  const float* chains;  // [R][Kft+1][AF]
  const float* obs;     // [R][cond]
  const float* temb;    // [n_time][td]
  const dppo_step* ksteps;
  int Kft, AF, td, cond;
  int obs_in_a;  // 1: the observation goes into inA's state columns; 0: zeros there (a cond_mlp encoder fills them)
  int onehot0;   // >= 0: column onehot0 + k of inA is set to 1 (one-hot of the row's denoising step in the K padding: the
                 // first layer's weight-gradient GEMM then returns sum_{rows of step k} dh0 as extra columns, which is
                 // all the time-embedding gradient needs); -1: off.  The forward never sees these columns (zero weights).
  int64_t M;
  void* inA;  // [M][KpA] elem : [x_k | temb(t_k) | obs | 0]
  int KpA;
  void* inC;  // [M][KpC] elem : [obs | 0]   (may be null)
  int KpC;
  int32_t* brow;  // [M]
  int32_t* krow;  // [M]
  // side jobs of block 0 (all optional): the PPO loss's per-k table, [Kft] denoising discount then [Kft] clip range,
  // in the reference's precision recipe (double pow / exp, then fp32), and two small double arrays to zero
  float* loss_tab;
  dppo_ppo_cfg pcfg;
  double* zero_a;  // n <= 128 each; null / 0 = none
  double* zero_b;
  double* zero_c;
  int n_zero_a, n_zero_b, n_zero_c;
  // advantage moments as a rider (mom_adv != null): the LAST ADV_RIDER_BLOCKS blocks of the launch also sum adv_k[row] and
  // its square over the minibatch's samples (row from inds / kinds, nothing this launch writes) and leave their partial
  // sums in mom_out[8 + 2 i], [9 + 2 i]; the loss kernel adds the ADV_RIDER_BLOCKS partials itself (LossArgs::mom_blocks).
  // No separate moments launch in front of the actor's forward, no arrival counter.
  const float* mom_adv;
  double* mom_out;
};
template <class P>
void launch_build_rows(const BuildRows& a, hipStream_t s);

// direct rows (DiffusionMLP.forward / CriticObs.forward called on explicit tensors)
template <class P>
void launch_build_direct(const float* x, const int64_t* t, const float* state, const float* temb, int AF, int td,
                         int cond, int64_t M, void* in, int Kp, hipStream_t s);

// zero columns [c0, c1) of an [M][ld] elem matrix (K padding the GEMM epilogues do not write)
template <class P>
void launch_zero_cols(void* X, int M, int c0, int c1, int ld, hipStream_t s);

// ---- log-prob epilogue (diffusion_vpg.py:381-396) -------------------------------------------------
struct LogprobArgs {
  const float* eps;  // [M][lde]
  int lde;
  const float* chains;  // [B][Kft+1][AF] ; row n = (b = n / Kft, k = n % Kft)
  const dppo_step* ksteps;
  dppo_diffusion_cfg cfg;
  int Kft, AF;
  int64_t M;
  float* logp;  // [M][AF]
};
void launch_logprob(const LogprobArgs& a, hipStream_t s);

// ---- behaviour-cloning term (diffusion_ppo.py:104-126): loss = -mean clamp(logp, -5, 2) over all (row, element);
// d_eps [M][ldde] elem (zero padded) = d loss / d eps, loss accumulated into *loss (double, zeroed by the caller's launch)
struct BcArgs {
  const float* eps;  // [M][lde] network output on the base policy's chains
  int lde;
  const float* chains;  // [B][Kft+1][AF]
  const dppo_step* ksteps;
  dppo_diffusion_cfg cfg;
  int Kft, AF;
  int64_t M;
  void* d_eps;
  int ldde;
  double* loss;
  double* partial;  // [bc_loss_blocks(M, ldde)] per-block sums + one 8-byte arrival counter behind them: the block that
                    // finishes last adds them in block order (no floating-point atomics: the same bits every run)
};
int64_t bc_loss_blocks(int64_t M, int ldde);
template <class P>
void launch_bc_loss(const BcArgs& a, hipStream_t s);

// supervised denoising loss (DiffusionModel.p_losses, model/diffusion/diffusion.py:325-349):
// loss = mean((eps - target)^2) over [M][AF], d_eps = 2 (eps - target) / (M AF); pairs [M][2][AF] = (x_noisy, target)
struct MseArgs {
  const float* eps;
  int lde;
  const float* pairs;
  int AF;
  int64_t M;
  void* d_eps;
  int ldde;
  double* loss;
  double* partial;  // as BcArgs::partial
};
template <class P>
void launch_mse_loss(const MseArgs& a, hipStream_t s);
void launch_axpy(float* y, const float* x, float alpha, int64_t n, hipStream_t s);  // y += alpha * x

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
  int mom_blocks;         // > 0: instead, moments[8 + 2 i], [9 + 2 i], i < mom_blocks (<= 64), are partial sums to add up (the
                          // row builder's riders) and the count is N; block 0 of the policy half stores the totals in
                          // moments_out[0..2] for the statistics' finalisation
  double* moments_out;
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
};
int loss_blocks(int64_t N);
template <class P>
void launch_ppo_loss(const LossArgs& a, hipStream_t s);  // the loss kernel: d_out + per-block partial sums
void launch_loss_finalize(const LossArgs& a, hipStream_t s);  // partial sums -> stats
// moments[0] += sum adv_k[brow[n]], [1] += sum of squares, [2] += N (float64; zeroed by the caller)
// moments: 8 + 2 * ADV_MOMENT_BLOCKS doubles; [3] must be zero on entry (see the kernel)
constexpr int ADV_MOMENT_BLOCKS = 256;
constexpr int ADV_RIDER_BLOCKS = 64;  // one partial per lane of the loss kernel's one-wave blocks
void launch_adv_moments(const float* adv_k, const int32_t* brow, int64_t N, double* moments, hipStream_t s);

// ---- time-embedding backward ---------------------------------------------------------------------
// partial[blk][k][j] = sum over the block's rows with krow == k of dtemb[row][j]
void launch_temb_segsum(const float* dtemb, int ld, const int32_t* krow, int64_t M, int Kft, int td, float* partial,
                        int blocks, hipStream_t s);
// G[k][td] (already reduced) -> grads of time_embedding.{1,3}.{weight,bias}; ksteps[k].t gives the time
// G[k][j] = sum_h W0[h*ldw0 + AF + j] * S[h*Kft + k] (S = per-step column sums of dh0; G: Kft*td floats of scratch), then
// the time MLP's backward
void launch_time_backward_from_sums(const float* w1, const float* b1, const float* w2, const float* S, const float* W0,
                                    int ldw0, int AF, int H, float* G, const dppo_step* ksteps, int Kft, int td, float* gw1,
                                    float* gb1, float* gw2, float* gb2, hipStream_t s);
void launch_time_backward(const float* w1, const float* b1, const float* w2, const float* G, const dppo_step* ksteps,
                          int Kft, int td, float* gw1, float* gb1, float* gw2, float* gb2, hipStream_t s);

// out[r][c] (ldo) = scale * sum_s slab[s][r][c] (lds) for r < rows, c < cols ; transpose: out[c][r] instead
// several split-M slabs summed in ONE launch (blockIdx.y = job): the weight-gradient GEMMs of a network each write their
// own slab, and their reductions -- latency-bound alone -- run together after the last GEMM
struct SlabJob {
  const float* slab;  // [splits][rows][lds]; columns c0 .. c0 + cols of every row are reduced
  float* out;         // [rows][ldo], or its transpose when `transpose`
  int splits, rows, cols, lds, ldo, transpose, c0;
  int wide;  // 1: many slabs (one per workgroup of the fused backward): 64 elements per block, the block's waves share the slabs
};
constexpr int MAX_SLAB_JOBS = 24;
struct SlabJobs {
  SlabJob j[MAX_SLAB_JOBS];
  int n;
};
void launch_slab_reduce_batch(const SlabJobs& jobs, hipStream_t s);
struct LossArgs;
struct TailReduce {  // see tail_reduce_kernel
  SlabJobs jobs;
  const float* colsum;  // [slots][tiles][width] per-tile column sums of the fused backward
  int tiles, width;
  SlotOuts slots;       // n_slots = 0: none
  const double *fin_partial, *fin_moments;  // set by the launcher from the loss's arguments
  double* fin_stats;
  int fin_blocks, fin_part;
  double fin_n_count;
};
void launch_tail_reduce(TailReduce& t, const LossArgs* fin, hipStream_t s);  // fin: loss statistics to finalise, or null
void launch_slab_reduce_2d(const float* slab, int splits, int rows, int cols, int lds, float* out, int ldo,
                           float scale, hipStream_t s, int transpose = 0);

// ---- GAE / optimiser -----------------------------------------------------------------------------
void launch_gae(const double* reward, const float* values, const float* terminated, const float* last_values, int S,
                int E, double gamma, double lam, double rconst, double* adv64, double* ret64, float* adv32,
                float* ret32, hipStream_t s);
void launch_sq_norm(const float* g, int64_t n, double* scratch, double* out, hipStream_t s);
void launch_adamw_dev(float* p, const float* g, float* m, float* v, int64_t n, int32_t* step_dev, const float* lr_dev,
                      double beta1, double beta2, float eps, double weight_decay, const double* sq_norm, float max_norm,
                      hipStream_t s);
void launch_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr_wd_mul, float one_m_b1, float b2,
                  float one_m_b2, float step_size, float bc2_sqrt, float eps, const double* sq_norm, float max_norm,
                  hipStream_t s);

// Every kernel-ready image of one network in ONE launch of 64-lane blocks: the streamed fragments of all layers (both
// streams), the out-layer stream, the (n_time x td) time-embedding table, and one transposed operand copy.
struct PackNet {
  PackStream ps;
  int ps_x;  // blocks per layer of the stream part (the widest layer's count)
  const float* Wout;
  int out_dim, H, OT, CNT;
  u32x4* ostream;
  const float* Wc;  // non-null: fp32 [out_dim][H] composite Wout . W2 of the top block -> ostream2 (same layout as ostream)
  u32x4* ostream2;
  const float* W0c;  // non-null: fp32 [out_dim][Kp0s] composite Wout . W0 -> ostream0, [ks][to][lane] for ks < 8 * CNT0
  u32x4* ostream0;
  int Kp0s, CNT0;
  const float *te_w1, *te_b1, *te_w2, *te_b2;  // n_time = 0: no table
  int td, n_time;
  float* temb;
  const float* tsrc;  // t_cols = 0: no transpose (dst[c][r] = src[r][coff + c], zero for r >= rows)
  int t_rows, t_cols, t_lds, t_coff, t_ldd;
  void* tdst;
};
template <class P>
void launch_pack_net(const PackNet& n, hipStream_t s);
struct PackNets {  // two networks in one launch (dppo_pack_nets)
  PackNet n[2];
  int base1;  // first block of the second network
};
static_assert(sizeof(PackNets) <= 4096, "kernel arguments are limited to 4 KB");
template <class P>
void launch_pack_nets(PackNets& q, hipStream_t s);
bool pack_net_supports(int time_dim);

// AdamW on up to four flat parameter vectors in one launch; step counts / learning rates in device memory.
// step_dev: int32[2] = {steps taken, 0}: the slot's last block to finish advances [0] (no tick launch).
struct AdamwSlot {
  float* p;
  const float* g;
  float *m, *v;
  int64_t n;
  int32_t* step_dev;
  const float* lr_dev;
  double beta1, beta2, weight_decay;
  float eps, max_norm;
  const double* sq_norm;  // null: no clipping
  int block0, blocks;
  int vec4;  // set by the launcher: all four vectors are 16-byte aligned
};
struct AdamwSlots {
  AdamwSlot s[4];
  int n;
};
void launch_adamw_multi(AdamwSlots& a, hipStream_t s);
// Wc = Wout . W2 ([out_dim][H] fp32) and cbias = bout + Wout . b2 of the top block, for up to two networks in one launch
struct ComposeJob {
  const float *Wout, *W2, *b2, *bout;
  float *Wc, *cbias;
  int H, out_dim;
  // one-block networks, the fused forward's merged out layer (see FusedFwdArgs::merge_top): h_0 = W0 x + b0 is linear in
  // the input, so Wout . h_1 = (Wout . W0) x + (Wout . W2) act(z1) + Wout (b0 + b2).  W0c: fp32 [out_dim][Kp0s], columns
  // >= in_dim zero; cbias2 = bout + Wout (b0 + b2).  W0 == null: not wanted.
  const float *W0, *b0;
  float *W0c, *cbias2;
  int in_dim, Kp0s;
};
struct ComposeJobs {
  ComposeJob j[2];
  int n;
};
void launch_compose(const ComposeJobs& q, hipStream_t s);
struct PostReduce {  // see post_reduce_kernel; dW == null: no low-rank part; G == null: no time-embedding part
  const float *Wout, *T;
  float* dW;
  int out_dim, H;
  const float *S, *W0;
  int ldw0, AF, Kft, td;
  float* G;
  const float *w1, *b1, *w2;
  const dppo_step* ksteps;
  float *gw1, *gb1, *gw2, *gb2;
  unsigned* counter;  // zero on entry; left zero
  // merged top (U != null): dWout[o][h] = sum_c U[o][c] W0[h][c] + sum_j T[o][j] W2[h][j] + cs[o] (b0[h] + b2[h]), the
  // out-layer weight gradient d_out^T . h_1 without h_1 (U = d_out^T . x, [out_dim][ldu]; cs = column sums of d_out)
  const float *U, *W2, *b0, *b2, *cs;
  float* dWout;
  int ldu, in_dim;
  // one-block backward (db2 != null): the second layer's bias gradient colsum(dh_1) = cs . Wout_b, which the kernel no
  // longer forms (fused_backward_one_kernel)
  const float* Wout_b;
  float* db2;
  // in-kernel dW0 (api.hip, mlp_backward): S_rest != null: S holds the one-hot sums of the first Kft - 1 steps only, the last
  // one is S_rest[h] - (their sum) (S_rest = column sums of the rounded dh_0: every row carries exactly one step; G's last row is
  // then formed from S_rest and corrected by the block that finishes the time MLP's backward).  dW0t != null:
  // the time-embedding columns of the first layer's weight gradient are formed here from the same sums,
  // dW0t[h * ldw0 + AF + j] = sum_k S[h][k] elem(temb[t_k][j]) (temb: the table the input rows were built from, rounded as they were)
  const float* S_rest;
  float* dW0t;
  const float* temb;
  int temb_bf16;  // 1: the rows hold bf16 roundings of the table
  int n_lowrank, n_temb, n_wout, n_dw0t;  // set by the launcher
  // riding in the weight-gradient GEMM launch (post_blocks.h): S, S_rest and the bias sums come from other workgroups of the SAME
  // launch -- wait until *wait_cnt >= wait_need before reading them, and read them past the non-coherent caches.  null: a launch of its own
  unsigned* wait_cnt;
  int wait_need;
};
// slab reductions + the post-reduce parts that read only the first n_first_jobs jobs' results, in one launch (tail_post_kernel)
struct TailPost {
  SlabJobs jobs;
  int n_first_jobs;             // jobs [0, n_first_jobs) feed q (the thin products T and U); the others run beside it
  int job_blocks[MAX_SLAB_JOBS], n_first;  // set by the launcher
  PostReduce q;                 // low-rank dW2 / dWout / db2 parts only; q.wait_cnt = a zeroed counter
};
void launch_tail_post(TailPost& t, hipStream_t s);
void launch_wout_grad(const PostReduce& q, hipStream_t s);  // the merged-top / one-block parts alone (no arrival counter needed)
void launch_post_reduce(PostReduce& q, hipStream_t s);
size_t time_backward_lds_bytes(int Kft, int td);  // LDS of the time MLP's backward block: must stay <= 156 KB
void launch_lowrank_dw(const float* Wout, const float* T, int out_dim, int H, float* dW, hipStream_t s);
void launch_stats_split(const double* st, float* hi_lo, int n, hipStream_t s);
void launch_stats_merge(const float* hi_lo, double* st, int n, int first_avg, int n_avg, double inv_world, hipStream_t s);

}  // namespace dppo
