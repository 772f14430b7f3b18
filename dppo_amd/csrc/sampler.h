// Persistent K-step DDPM/DDIM sampler (reference: model/diffusion/diffusion_vpg.py:139-315).
//
// One 512-thread workgroup (8 waves) owns 16 env rows for ALL denoising steps: x, the observation
// and the hidden activations never leave LDS/registers, the chain is written once, coalesced.
// The MLP runs transposed on MFMA 16x16: A = weight fragments, B = the 16 rows' activations.  At 16
// rows per workgroup each weight element is used by exactly one wave, so weights are NOT staged in
// LDS: the host packs every network once into a per-wave "fragment stream" (consumption order, one
// perfectly coalesced 1 KiB read per wave-instruction) that each wave walks linearly through a
// 4-deep register ring that prefetches across layer and step boundaries.
#pragma once
#include "common.h"
#include "../../include/dppo_hip.h"

namespace dppo {

constexpr int SAMPLER_WAVES = 8;
// ring depth in k-step positions: 4 (16-32 KiB of weights in flight per wave); 2 at H = 1024 to stay under 256 VGPRs
__host__ __device__ constexpr int sampler_pd(int hidden) { return hidden >= 1024 ? 2 : 4; }
constexpr int MAX_BLOCKS = 8;

// geometry of a packed actor for the sampler
struct SamplerGeom {
  int H, nb, in_dim, out_dim, Kp0, KS0, KSH, TPW, OT, CNT, total_pos;
  size_t hidden_frags_per_wave, out_frags_per_wave;
};
template <class P>
SamplerGeom sampler_geom(const dppo_net_desc& d);

struct SampleArgs {
  const u32x4* wstream[2];
  const u32x4* ostream[2];
  const u32x4* ostream2[2];  // merged out layer: fragments of Wout . W2 (top block), applied to act(z1)
  const float* cbias[2];     // merged out layer: bout + Wout . b2
  int merge_top;  // request: out = Wout . h_in + (Wout . W2) . act(z1) + cbias, the top block's second layer is never run
                  // (h_nb = h_in + W2 act(z1) + b2 is only ever consumed by the out layer).  The launcher clears it when
                  // the extra LDS image does not fit.
  const float* params[2];
  const float* temb[2];  // [n_time][td]
  int bias_off[2 + 2 * MAX_BLOCKS];  // L0, (l1,l2) x nb, out  -- float offsets into params
  int ln_off[4 * MAX_BLOCKS];        // per block: norm1.weight, norm1.bias, norm2.weight, norm2.bias (use_ln only)
  int use_ln;
  const float* obs[2];  // per network: [B][ld_obs] state columns (raw observation, or its cond_mlp encoding)
  int ld_obs;
  const float* noise;   // [n_steps+1][B][AF], or null: drawn in the kernel (Philox4x32-10 keyed by seed_lo / seed_hi)
  uint32_t seed_lo, seed_hi;
  float* traj;          // [B][AF]
  float* chains;        // [B][chain_len][AF]
  const dppo_step* sched;
  int B, AF, td, cond, Kp0, nb, n_steps, chain_len, init_slot, act, use_ddim;
  int has_dclip, has_eclip, has_fclip;
  int consts_lds;  // set by the launcher: 1 = both networks' biases staged in LDS
  int ks0v;        // set by the launcher: k-steps of layer 0 that hold input columns (the rest of its KS0 is zero padding)
  int l0_lds;      // set by the launcher: 1 = layer-0 fragments of the current network live in LDS, not in the stream
  int pre_sweep;   // split sampler, set by its launcher: sleep periods between the exchange store and the first sweep (knob 28)
  unsigned spin_limit;  // split sampler, set by its launcher: sweeps a member waits for its tile before it gives up (knob 29)
  float dclip, eclip, rclip, fclip;
};

template <class P>
int launch_sample_chain(const SamplerGeom& g, const SampleArgs& a, hipStream_t s);  // 0 ok, <0 unsupported
void set_sampler_l0_lds(int v);  // tuning knob 15

// One tile over eight workgroups (sampler_split.hip): small env batches of one-block bf16 networks at hidden 512.
// `xch`: the exchange block at the START of the caller's workspace (zeroed by the launcher before every launch; its first
// word is the time-out word: 0 = every hand-over completed, else 1 + the step a member gave up at).
bool sampler_split_ok(const dppo_net_desc& d, bool bf16, int64_t B, bool merge_top);
size_t sampler_split_xch_bytes(const dppo_net_desc& d, int64_t B);
int launch_sample_chain_split(const SamplerGeom& g, const SampleArgs& a, void* xch, size_t xch_bytes, hipStream_t s);  // -1: not covered
void set_sampler_split(int v);  // tuning knob 27
void set_sampler_split_pre_sweep(int v);  // tuning knob 28
void set_sampler_split_spin_limit(int v);  // tuning knob 29 (tests only: force a time-out)

// W: [H][ld] fp32 (nn.Linear layout).  Writes the fragments of one hidden layer (KS k-steps) into
// every wave's stream at position pos0.
template <class P>
void launch_pack_hidden(const float* W, int H, int in_valid, int ld, int KS, int TPW, int pos0, int total_pos,
                        u32x4* stream, hipStream_t s);
template <class P>
void launch_pack_out(const float* W, int out_dim, int H, int OT, int CNT, u32x4* stream, hipStream_t s);

}  // namespace dppo
