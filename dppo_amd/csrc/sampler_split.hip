// K-step sampler for small env batches, one 16-row tile spread over EIGHT workgroups (reference:
// model/diffusion/diffusion_vpg.py:139-315, the same path as sampler.hip).  gfx950 only; compiled with -ffp-contract=off.
//
// sample_chain_kernel (sampler.hip) gives 16 env rows to one CU, so 512 envs keep 32 of the 256 CUs busy, and every
// denoising step streams the 512 x 512 layer (0.5 MB bf16) into each of them at the 64 B/clk L1 fill rate: 3.4 of its
// 6.5 us per step.  Here the tile's eight "waves" of the packed fragment streams become eight WORKGROUPS ("members") on
// eight CUs: member m owns the 64 hidden features of stream slice m, and a 4-wave workgroup holds everything it needs of
// both layers IN REGISTERS for the whole call (layer 0 in full -- its K is in_dim, every member computes all of h_0 --
// and its 64 x 512 slice of the block's first layer: 36 fragments = 144 VGPRs per lane at one wave per SIMD); no weight
// is read again until the step table switches network.  What members exchange per step is the thing the waves of the
// one-CU kernel exchange through LDS: their partial sums of the (merged) out layer, 16 rows x 16 columns of fp32 per out
// tile, through memory with write-through (sc1) 16-byte stores and L1-bypassing (sc1) loads, and the data is the flag (guide
// section 6, guideline 16, form R2, with a narrower granule): every 32-bit word carries a one-bit tag in the LEAST SIGNIFICANT
// BIT of the fp32 partial sum (slot = step parity, tag = parity of step / 2, inverted so that the zeroed block of the
// call's start never matches), so a word validates itself whatever the tearing of the 16-byte store, a lane's hand-over is
// ONE store and eight loads, and a sweep of the eight members' partials moves 8 KB instead of the 16 KB of {tag, value}
// pairs (a sweep is latency- and size-bound: 1.5 us at 16 KB).  Every member then adds the eight partials (tag bit and all:
// the sums differ from sample_chain_kernel's by at most one ulp of an fp32 partial, 6e-8 relative) in slice order and
// runs the posterior for all 16 rows redundantly, so no second exchange is needed and x_{t-1} is bit-identical on all
// eight members.  The step's noise (Philox is counter based) is drawn by the waves that own no out tile while the owners
// wait for the exchange.  Member 0 writes the chain and the trajectory.
//
// Requirements (the launcher checks them, the caller falls back to sample_chain_kernel): bf16 operands, hidden 512, one
// residual block with the merged out layer (knob 17), no LayerNorm, out_dim <= 64, in_dim <= 96, and tiles * 8 <= the
// device's CU count (B <= 512 on MI355X) so that all members of a tile are resident while they wait for each other;
// the wait is bounded all the same: a member that gives up sets the time-out word, poisons its rows of the trajectory
// with NaN and leaves.
#include "sampler.h"
#include "gemm.h"

namespace dppo {

typedef const __attribute__((address_space(1))) u32x4* sgfrag_p;
typedef const __attribute__((address_space(1))) float* sgfloat_p;
typedef __attribute__((address_space(1))) float* sgfloat_w;
typedef __attribute__((address_space(1))) unsigned* sgu32_p;

constexpr int SPLIT = SAMPLER_WAVES;  // members per tile = slices of the packed streams

// Phase stamps (debug build only: DPPO_STAMPS=1 build.sh -> libdppo_hip_stamps.so, tools/sampler_stamps.py): s_memtime of
// the four waves of workgroup 0 at the phase boundaries of denoising step 5, and the number of polling passes.
#ifdef DPPO_STAMPS
__device__ unsigned long long g_split_stamps[4][16];
#define SSTAMP(k)                                                                     \
  do {                                                                                \
    if (blockIdx.x == 0 && lane == 0 && i == 5) g_split_stamps[wid][k] = clock64();   \
  } while (0)
#define SSTAMP_VAL(k, v)                                                              \
  do {                                                                                \
    if (blockIdx.x == 0 && lane == 0 && i == 5) g_split_stamps[wid][k] = (v);         \
  } while (0)
#define SSTAMP_ONCE(k)                                                                \
  do {                                                                                \
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) g_split_stamps[threadIdx.x >> 6][k] = clock64(); \
  } while (0)
#else
#define SSTAMP(k)
#define SSTAMP_VAL(k, v)
#define SSTAMP_ONCE(k)
#endif
constexpr unsigned SPLIT_SPINS = 1u << 20;

template <class P>
__device__ __forceinline__ void split_lds_put(char* buf, int rb, int kmask, int row, int col, float v) {
  const int byte = col * P::ESIZE;
  char* p = buf + row * rb + ((((byte >> 4) ^ (row & kmask)) << 4) | (byte & 15));
  *(typename P::elem_t*)p = P::from_f32(v);
}
__device__ __forceinline__ int split_kmask_of(int rb) {
  const int n = rb >> 4;
  const int p = n & (-n);
  return (p > 16 ? 16 : p) - 1;
}

// Two fp32 values -> one packed bf16 pair (ONE v_cvt_pk_bf16_f32, round-to-nearest-even), activation applied.  ReLU is taken
// on the packed pair as a signed 16-bit max with 0 (v_pk_max_i16): rounding to bf16 keeps the sign, so max(round(x), 0) ==
// round(max(x, 0)) for every non-NaN x (-0 becomes +0).
typedef __attribute__((ext_vector_type(2))) float split_f2;
typedef __attribute__((ext_vector_type(2))) __bf16 split_bf2;
typedef __attribute__((ext_vector_type(2))) short split_s2;
template <int ACT>
__device__ __forceinline__ uint32_t pack2_act(float lo, float hi) {
  if constexpr (ACT == ACT_MISH) lo = mish_f(lo), hi = mish_f(hi);
  const split_bf2 b = __builtin_convertvector((split_f2){lo, hi}, split_bf2);
  if constexpr (ACT == ACT_RELU) {
    const split_s2 z = {0, 0};
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(split_s2, b), z));
  }
  return __builtin_bit_cast(uint32_t, b);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a workgroup-scope release / acquire of ALL memory: in
// front of s_barrier the compiler waits for vmcnt(0), i.e. for member 0's chain stores and everybody's exchange store to be
// acknowledged by memory -- ~2k cycles on the critical path of every step, for stores no wave of this workgroup ever reads.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <class P, int KS0V, int OT, int ACT>
__global__ __launch_bounds__(256) void sample_chain_split_kernel(const SampleArgs a, char* xch_, int xch_bytes, unsigned* tmo_) {
  constexpr int ES = P::ESIZE, KB = P::KB, TPW = 4, H = 512;
  constexpr int KSH = H / KB, CNT = KSH / SPLIT, HRB = H * ES;
  static_assert(KSH % SPLIT == 0 && ES == 2, "bf16 at H = 512");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  SSTAMP_ONCE(10);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);  // 0..3: tile tp of the member's slice; out tile
  const int r = lane & 15, g = lane >> 4;
  // Block id -> (tile, member): the eight members of a tile get ids that are congruent modulo 8 and consecutive within their
  // residue class.  Blocks are observed to be dealt round-robin over the 8 XCDs, each XCD placing its share in order, so a
  // tile's members sit on ONE XCD and whatever part of the grid is resident -- another queue or process may hold CUs -- is,
  // per XCD, a run of complete tiles plus at most one partial one: complete tiles always finish and free their CUs.  (With
  // tile = id / 8 a tile has one member on each XCD, and two kernels sharing the chip can each hold XCDs the other one's tiles
  // need: measured 8-9 ms per call with three processes sampling at once, against 0.21-0.26 ms with this mapping, and 70 us
  // instead of 77 us for one process alone.)  Only speed and
  // robustness depend on the placement, never results.  The grid is padded to a multiple of 64 blocks; empty tiles leave here.
  const int xcd_class = blockIdx.x & 7, jx = blockIdx.x >> 3;
  const int m = jx & 7, tile = xcd_class + 8 * (jx >> 3);
  if (tile * 16 >= a.B) return;
  const int grow0 = tile * 16;
  const int AF = a.AF, td = a.td, cond = a.cond, Kp0 = a.Kp0, B = a.B;
  const int in_rb = Kp0 * ES, in_km = split_kmask_of(in_rb);
  const int KS0 = Kp0 / KB;
  const int total = KS0 + 2 * KSH;  // positions of one slice's stream (one block)

  char* xin = smem;
  char* bufA = xin + 16 * in_rb;   // act(h_0), all 512 features
  char* bufB = bufA + 16 * HRB;    // act(z1), this member's 64 features
  char* bufC = bufB + 16 * HRB;    // raw h_0, this member's 64 features
  float* biasL = (float*)(bufC + 16 * HRB);  // [net][b0: H | b1: H | merged out bias: OT*16]
  constexpr int BSTR = 2 * H + OT * 16;
  int* failL = (int*)(biasL + 2 * BSTR);
  // the step table and every step's time-embedding row, staged once: a step is ~3 us long, a scalar load of its table entry
  // or a global load of its embedding row at the step's top would sit on the critical path of every step
  dppo_step* schedL = (dppo_step*)(failL + 4);
  float* teL = (float*)(schedL + a.n_steps);  // [n_steps][td]
  float* zL = teL + a.n_steps * a.td;         // [16][AF]: this step's draws (written by the waves that own no out tile)

  const sgfloat_p g_noise = (sgfloat_p)a.noise;
  const sgfloat_w g_chains = (sgfloat_w)a.chains, g_traj = (sgfloat_w)a.traj;
  // a wave owns out tile `wid` (columns wid*16 + 4g + e of batch row r) if that tile holds action columns; the other waves
  // ("helpers": 3 at Ta*Da <= 16, none above 48) draw the noise while the owners wait for the exchange
  const int n_own = (AF + 15) >> 4;
  const bool owner = wid < n_own;
  const int jcol = wid * 16 + 4 * g;  // first of the lane's four action columns
  const int grow = grow0 + r, growc = min(grow, B - 1);
  const bool writer = m == 0 && grow < B;

  // ---- the member's weights, resident in registers.  Layer 0: wave `wid` computes the tiles of stream slices 2 wid and
  // 2 wid + 1 (8 tiles x KS0V k-steps); first block layer: tile `wid` of slice m (KSH k-steps); out layer: out tile `wid`
  // of slice m's CNT k-steps, both fragment sets of the merged form.
  const size_t wave_stride = (size_t)total * TPW * 64;
  u32x4 w0f[8][KS0V], w1f[KSH], of[CNT], of2[CNT];
  auto load_weights = [&](int net) {
    const sgfrag_p wsn = (sgfrag_p)a.wstream[net] + lane;
#pragma unroll
    for (int j8 = 0; j8 < 8; ++j8) {
      const sgfrag_p s = wsn + (size_t)(2 * wid + (j8 >> 2)) * wave_stride;
#pragma unroll
      for (int ks = 0; ks < KS0V; ++ks) w0f[j8][ks] = s[((size_t)ks * TPW + (j8 & 3)) * 64];
    }
    const sgfrag_p s1 = wsn + (size_t)m * wave_stride;
#pragma unroll
    for (int ks = 0; ks < KSH; ++ks) w1f[ks] = s1[((size_t)(KS0 + ks) * TPW + wid) * 64];
    if (owner) {
      const sgfrag_p o1 = (sgfrag_p)a.ostream[net] + (size_t)m * CNT * OT * 64 + lane;
      const sgfrag_p o2 = (sgfrag_p)a.ostream2[net] + (size_t)m * CNT * OT * 64 + lane;
#pragma unroll
      for (int c = 0; c < CNT; ++c) of[c] = o1[(c * OT + wid) * 64], of2[c] = o2[(c * OT + wid) * 64];
    }
  };
  // ---- prologue.  Everything the call needs from memory is requested before anything is waited for (weights, biases,
  // step table, state columns, x_K): issued one after the other with a wait in between, as five dependent phases, this took
  // 17k cycles = 7 us of an 80 us call.
  const int net0 = a.sched[0].net;
  load_weights(net0);
  float bv[2][4], cbv[2];
#pragma unroll
  for (int net = 0; net < 2; ++net) {
    const float* p0 = a.params[net] + a.bias_off[0];
    const float* p1 = a.params[net] + a.bias_off[1];
    bv[net][0] = p0[tid], bv[net][1] = p0[tid + 256], bv[net][2] = p1[tid], bv[net][3] = p1[tid + 256];
    cbv[net] = tid < AF ? a.cbias[net][tid] : 0.f;  // (AF <= 64 < 256)
  }
  // x_K: given, or drawn here -- one element per thread (16 x AF <= 1024 elements), not four per owner lane
  float zk[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int idx = tid + 256 * q;
    zk[q] = 0.f;
    if (idx < 16 * AF) {
      const int row = idx / AF;
      const size_t ni = (size_t)min(grow0 + row, B - 1) * AF + (idx - row * AF);
      zk[q] = a.noise != nullptr ? g_noise[ni] : philox_normal(ni, a.seed_lo, a.seed_hi);
    }
  }
  // the first network's observation columns (up to 4 elements per thread here, the rest in put_state's loop) and the
  // step table's words, requested with the rest
  float ob0[4];
  int sw[2];
  {
    const float* ob = a.obs[net0];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int idx = tid + 256 * q;
      ob0[q] = 0.f;
      if (idx < 16 * cond) {
        const int row = idx / cond;
        ob0[q] = ob[(size_t)min(grow0 + row, B - 1) * a.ld_obs + (idx - row * cond)];
      }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int idx = tid + 256 * q;
      sw[q] = idx < a.n_steps * (int)(sizeof(dppo_step) / 4) ? ((const int*)a.sched)[idx] : 0;
    }
  }
  // zero the input image (padding columns stay zero for the whole call)
  for (int idx = tid; idx < in_rb; idx += 256) *(u32x4*)(xin + idx * 16) = (u32x4){0u, 0u, 0u, 0u};
  if (tid == 0) *failL = 0;
  __syncthreads();
#pragma unroll
  for (int net = 0; net < 2; ++net) {
    float* bl = biasL + net * BSTR;
    bl[tid] = bv[net][0], bl[tid + 256] = bv[net][1], bl[H + tid] = bv[net][2], bl[H + tid + 256] = bv[net][3];
    if (tid < OT * 16) bl[2 * H + tid] = cbv[net];
  }
#pragma unroll
  for (int q = 0; q < 2; ++q)
    if (tid + 256 * q < a.n_steps * (int)(sizeof(dppo_step) / 4)) ((int*)schedL)[tid + 256 * q] = sw[q];
  for (int idx = tid + 512; idx < a.n_steps * (int)(sizeof(dppo_step) / 4); idx += 256) ((int*)schedL)[idx] = ((const int*)a.sched)[idx];
  for (int idx = tid; idx < a.n_steps * (td >> 2); idx += 256) {  // one (step, four columns) piece per thread; td % 4 == 0
    const int st_i = idx / (td >> 2), c4 = (idx - st_i * (td >> 2)) * 4;
    const dppo_step sx = a.sched[st_i];
    const float* src = a.temb[sx.net] + sx.t * td + c4;
    const float t0 = src[0], t1 = src[1], t2 = src[2], t3 = src[3];
    float* dst = teL + st_i * td + c4;
    dst[0] = t0, dst[1] = t1, dst[2] = t2, dst[3] = t3;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q)
    if (tid + 256 * q < 16 * AF) zL[tid + 256 * q] = zk[q];

  auto put_state = [&](int net, int first) {  // the observation columns of the input image (the rest of it is never touched)
    const float* ob = a.obs[net];
    for (int idx = tid + first; idx < 16 * cond; idx += 256) {
      const int row = idx / cond, j = idx - row * cond;
      split_lds_put<P>(xin, in_rb, in_km, row, AF + td + j, ob[(size_t)min(grow0 + row, B - 1) * a.ld_obs + j]);
    }
  };
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int idx = tid + 256 * q;
    if (idx < 16 * cond) {
      const int row = idx / cond;
      split_lds_put<P>(xin, in_rb, in_km, row, AF + td + (idx - row * cond), ob0[q]);
    }
  }
  put_state(net0, 1024);
  __syncthreads();

  // x_K into the owners' registers, the image and the chain; the time embedding of step 0
  float xc[4] = {0.f, 0.f, 0.f, 0.f};
  if (owner) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int j = jcol + e;
      if (j < AF) {
        const float v = zL[r * AF + j];
        xc[e] = v;
        split_lds_put<P>(xin, in_rb, in_km, r, j, v);
        if (a.init_slot >= 0 && a.chains != nullptr && writer) g_chains[((size_t)grow * a.chain_len + a.init_slot) * AF + j] = v;
      }
    }
  }
  for (int idx = tid; idx < 16 * td; idx += 256) {
    const int row = idx / td, j = idx - row * td;
    split_lds_put<P>(xin, in_rb, in_km, row, AF + j, teL[j]);
  }
  __syncthreads();

  // exchange slots: [tile][step parity][member][out tile][lane] 16 bytes = the lane's four partial sums, tag in bit 0 of each
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(xch_, 0, xch_bytes, 0x00020000);
  constexpr int SLOT = OT * 64 * 16;  // bytes per member
  const bool active = owner && jcol < AF;  // lanes whose four columns hold no action dimension exchange nothing
  bool failed = false;
  // Loop-invariant arguments the posterior needs, pinned in VGPRs (the asm makes them opaque): left as kernel arguments the
  // compiler, short of SGPRs, re-loads them from the argument segment (s_load + s_waitcnt lgkmcnt(0)) for every element of
  // every step -- ~1k cycles of the posterior.  The flags are read back uniformly (readfirstlane) once per step.
  const int n_steps = a.n_steps, chain_len = a.chain_len;
  float dclip = a.dclip, eclip = a.eclip, rclip = a.rclip, fclip = a.fclip;
  int flags_v = (a.use_ddim ? 1 : 0) | (a.has_dclip ? 2 : 0) | (a.has_eclip ? 4 : 0) | (a.chains != nullptr ? 8 : 0);
  asm volatile("" : "+v"(dclip), "+v"(eclip), "+v"(rclip), "+v"(fclip), "+v"(flags_v));
  const bool have_noise = a.noise != nullptr;
  const uint32_t seed_lo = a.seed_lo, seed_hi = a.seed_hi;
  const int te_row = tid / td, te_j = tid - te_row * td;

  SSTAMP_ONCE(11);
  for (int i = 0; i < n_steps; ++i) {
    SSTAMP(0);
    const dppo_step st = schedL[i];
    const int net = __builtin_amdgcn_readfirstlane(st.net);
    const int nnet = __builtin_amdgcn_readfirstlane(schedL[min(i + 1, n_steps - 1)].net);
    const float* bL = biasL + net * BSTR;
    // ---- layer 0, all of it: h_0 = W0 [x, t, obs] + b0
    f32x4 acc[8];
#pragma unroll
    for (int j8 = 0; j8 < 8; ++j8)
      acc[j8] = *(const f32x4*)(bL + (2 * wid + (j8 >> 2)) * 64 + feat_off<P>(g, j8 & 3));
#pragma unroll
    for (int ks = 0; ks < KS0V; ++ks) {
      const u32x4 xb = *(const u32x4*)(xin + r * in_rb + (((ks * 4 + g) ^ (r & in_km)) << 4));
#pragma unroll
      for (int j8 = 0; j8 < 8; ++j8) acc[j8] = P::mma(w0f[j8][ks], xb, acc[j8]);
    }
#pragma unroll
    for (int j8 = 0; j8 < 8; j8 += 2) {  // tiles tp, tp + 1 of a slice: eight consecutive features of the lane = one 16-byte chunk
      const int slice = 2 * wid + (j8 >> 2);
      const int c = ((slice * 64 + feat_off<P>(g, j8 & 3)) * ES) >> 4;
      u32x4 o;
      o.x = pack2_act<ACT>(acc[j8][0], acc[j8][1]), o.y = pack2_act<ACT>(acc[j8][2], acc[j8][3]);
      o.z = pack2_act<ACT>(acc[j8 + 1][0], acc[j8 + 1][1]), o.w = pack2_act<ACT>(acc[j8 + 1][2], acc[j8 + 1][3]);
      *(u32x4*)(bufA + r * HRB + ((c ^ (r & 15)) << 4)) = o;
      if (slice == m) {  // raw h_0 of the member's own features, for the merged out layer
        u32x4 q;
        q.x = pack2_act<ACT_NONE>(acc[j8][0], acc[j8][1]), q.y = pack2_act<ACT_NONE>(acc[j8][2], acc[j8][3]);
        q.z = pack2_act<ACT_NONE>(acc[j8 + 1][0], acc[j8 + 1][1]), q.w = pack2_act<ACT_NONE>(acc[j8 + 1][2], acc[j8 + 1][3]);
        *(u32x4*)(bufC + r * HRB + ((c ^ (r & 15)) << 4)) = q;
      }
    }
    SSTAMP(1);
    lds_barrier();
    SSTAMP(2);
    // the input image is free until the next step's layer 0: the next step's time embedding goes in now
    if (i + 1 < n_steps) {
      if (tid < 16 * td) split_lds_put<P>(xin, in_rb, in_km, te_row, AF + te_j, teL[(i + 1) * td + te_j]);
      for (int idx = tid + 256; idx < 16 * td; idx += 256)  // time_dim > 16
        split_lds_put<P>(xin, in_rb, in_km, idx / td, AF + idx % td, teL[(i + 1) * td + idx % td]);
    }

    // ---- first block layer, the member's 64 features: z1 = W1 act(h_0) + b1
    f32x4 a1 = *(const f32x4*)(bL + H + m * 64 + feat_off<P>(g, wid));
    {
      // all B fragments first, then the chain of MFMAs: left alone, the compiler reuses ONE fragment register and runs
      // read -> wait -> MFMA sixteen times in series (1.8k cycles for 16 MFMAs)
      u32x4 xb1[KSH];
#pragma unroll
      for (int ks = 0; ks < KSH; ++ks) xb1[ks] = *(const u32x4*)(bufA + r * HRB + (((ks * 4 + g) ^ (r & 15)) << 4));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < KSH; ++ks) a1 = P::mma(w1f[ks], xb1[ks], a1);
    }
    {
      const int byte = (m * 64 + feat_off<P>(g, wid)) * ES;  // four features = 8 bytes, half a chunk
      u32x2 o;
      o.x = pack2_act<ACT>(a1[0], a1[1]), o.y = pack2_act<ACT>(a1[2], a1[3]);
      *(u32x2*)(bufB + r * HRB + ((((byte >> 4) ^ (r & 15)) << 4) | (byte & 15))) = o;
    }
    SSTAMP(3);
    lds_barrier();
    SSTAMP(4);

    // ---- merged out layer over the member's features, exchange, posterior
    u32x4 raw[SPLIT];
    float z[4] = {0.f, 0.f, 0.f, 0.f};
    bool have_chains = false;
    if (owner && !failed) {
      f32x4 oacc = (f32x4){0.f, 0.f, 0.f, 0.f};
      {
        u32x4 xo[CNT], xo2[CNT];
#pragma unroll
        for (int c = 0; c < CNT; ++c) {
          const int ks = m * CNT + c;
          xo[c] = *(const u32x4*)(bufC + r * HRB + (((ks * 4 + g) ^ (r & 15)) << 4));
          xo2[c] = *(const u32x4*)(bufB + r * HRB + (((ks * 4 + g) ^ (r & 15)) << 4));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < CNT; ++c) {
          oacc = P::mma(of[c], xo[c], oacc);
          oacc = P::mma(of2[c], xo2[c], oacc);
        }
      }
      const unsigned tbit = ((((unsigned)i >> 1) & 1u) ^ 1u);
      const int off0 = ((tile * 2 + (i & 1)) * SPLIT) * SLOT + (wid * 64 + lane) * 16;
      if (active) {
        u32x4 pv;
#pragma unroll
        for (int e = 0; e < 4; ++e) pv[e] = (__float_as_uint(oacc[e]) & ~1u) | tbit;
        __builtin_amdgcn_raw_buffer_store_b128(pv, rsrc, off0 + m * SLOT, 0, 16);  // aux 16 = sc1: write-through
      }
      SSTAMP(5);
      // One sweep = eight 16-byte sc1 loads per lane, ~2k cycles for the 6-8 KB a wave pulls across the fabric (two full
      // sweeps in flight take 3k: the cost is bytes as much as latency).  The members reach this point within a few hundred
      // cycles of each other and a store takes ~1.5k cycles to become visible, so the first sweep usually finds a slot or
      // two still empty: a retry re-reads ONLY the members a lane is still missing (`need`, one bit per member).
      unsigned need = active ? (1u << SPLIT) - 1u : 0u;
      auto sweep = [&]() {
#pragma unroll
        for (int w = 0; w < SPLIT; ++w)
          if ((need >> w) & 1u) raw[w] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off0 + w * SLOT, 0, 16);  // sc1
      };
      for (int q = 0; q < a.pre_sweep; ++q) __builtin_amdgcn_s_sleep(1);
      sweep();
      if (n_own == 4) {  // every wave owns an out tile: each draws its own columns' noise while its loads are in flight
        const size_t ni = (size_t)(i + 1) * B * AF + (size_t)growc * AF + jcol;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const size_t nie = jcol + e < AF ? ni + e : ni - jcol;  // (padding columns: any valid element, result unused)
          z[e] = have_noise ? g_noise[nie] : philox_normal(nie, seed_lo, seed_hi);
        }
      }
      unsigned spins = 0;
      for (;;) {
#pragma unroll
        for (int w = 0; w < SPLIT; ++w) {
          const bool got = ((raw[w][0] & raw[w][1] & raw[w][2] & raw[w][3] & 1u) == tbit) &&
                           (((raw[w][0] | raw[w][1] | raw[w][2] | raw[w][3]) & 1u) == tbit);
          if (got) need &= ~(1u << w);
        }
        if (__all(need == 0u)) break;
        if (++spins >= a.spin_limit) {
          failed = true;
          break;
        }
        sweep();
      }
      SSTAMP(6);
      SSTAMP_VAL(9, (unsigned long long)spins);
      if (failed && lane == 0) {
        // sticky: the word only ever grows and no launch clears it (split_zero_kernel leaves the head of the block alone), so a
        // time-out in ANY call since the host last read the word is still there when it looks (once per rollout)
        __hip_atomic_fetch_max((sgu32_p)tmo_, (unsigned)(i + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *failL = 1;
      }
    }
    if (n_own < 4) {
      // the waves without an out tile draw this step's noise for the whole tile while the owners wait for the exchange
      if (!owner) {
        const int NH = (4 - n_own) * 64;
        for (int idx = (wid - n_own) * 64 + lane; idx < 16 * AF; idx += NH) {
          const int row = idx / AF;
          const size_t ni = (size_t)(i + 1) * B * AF + (size_t)min(grow0 + row, B - 1) * AF + (idx - row * AF);
          zL[idx] = have_noise ? g_noise[ni] : philox_normal(ni, seed_lo, seed_hi);
        }
      }
      lds_barrier();
    }
    if (owner && !failed) {
      {
        // diffusion_vpg.py:165-223 (p_mean_var) and :279-311 (sampling loop); same operation sequence as sampler.hip.
        // Straight-line over the lane's four columns (padding columns compute on zeros and are never stored).
        const int fl = __builtin_amdgcn_readfirstlane(flags_v);
        const bool use_ddim = fl & 1, has_dclip = fl & 2, has_eclip = fl & 4;
        have_chains = fl & 8;
        float xn4[4], eps4[4], ze4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float eps = bL[2 * H + jcol + e];
#pragma unroll
          for (int w = 0; w < SPLIT; ++w) eps += __uint_as_float(raw[w][e]);  // (lanes without action columns: unused)
          eps4[e] = eps;
          ze4[e] = z[e];
          if (n_own < 4) ze4[e] = zL[r * AF + min(jcol + e, AF - 1)];
        }
        if (!use_ddim) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float x = xc[e];
            float x0 = st.c0 * x - st.c1 * eps4[e];
            if (has_dclip) x0 = fminf(fmaxf(x0, -dclip), dclip);
            xn4[e] = st.c2 * x0 + st.c3 * x;
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float x = xc[e];
            float eps = eps4[e];
            float x0 = (x - st.c1 * eps) / st.c0;
            if (has_dclip) {
              x0 = fminf(fmaxf(x0, -dclip), dclip);
              eps = (x - st.c0 * x0) / st.c1;
            }
            if (has_eclip) eps = fminf(fmaxf(eps, -eclip), eclip);
            xn4[e] = st.c2 * x0 + st.c3 * eps;
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float zc = fminf(fmaxf(ze4[e], -rclip), rclip);
          float xn = xn4[e] + st.std * zc;
          if (st.final_clip) xn = fminf(fmaxf(xn, -fclip), fclip);
          xn4[e] = xn;
          xc[e] = xn;
        }
        const bool chain_out = writer && st.chain_slot >= 0 && have_chains;
        const bool traj_out = writer && i + 1 == n_steps;
        if (jcol + 4 <= AF && (AF & 3) == 0) {  // the lane's four columns in one piece: 8 bytes of the bf16 image, 16 of the chain
          const int byte = jcol * ES;
          u32x2 o;
          o.x = pack2_act<ACT_NONE>(xn4[0], xn4[1]), o.y = pack2_act<ACT_NONE>(xn4[2], xn4[3]);
          *(u32x2*)(xin + r * in_rb + ((((byte >> 4) ^ (r & in_km)) << 4) | (byte & 15))) = o;
          const f32x4 v = (f32x4){xn4[0], xn4[1], xn4[2], xn4[3]};
          typedef __attribute__((address_space(1))) f32x4* sgf4_w;
          if (chain_out) *(sgf4_w)(g_chains + ((size_t)grow * chain_len + st.chain_slot) * AF + jcol) = v;
          if (traj_out) *(sgf4_w)(g_traj + (size_t)grow * AF + jcol) = v;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int j = jcol + e;
            if (j < AF) {
              split_lds_put<P>(xin, in_rb, in_km, r, j, xn4[e]);
              if (chain_out) g_chains[((size_t)grow * chain_len + st.chain_slot) * AF + j] = xn4[e];
              if (traj_out) g_traj[(size_t)grow * AF + j] = xn4[e];
            }
          }
        }
      }
    }
    SSTAMP(7);
    if (nnet != net) {  // the step table switches network: state columns of a cond_mlp actor, and every register weight
      if (a.obs[0] != a.obs[1]) put_state(nnet, 0);
      load_weights(nnet);
    }
    lds_barrier();
    SSTAMP(8);
    if (*(volatile int*)failL) break;
  }
  SSTAMP_ONCE(12);
  if (*(volatile int*)failL && m == 0 && owner && grow < B) {  // poison everything the call owed for these rows: the
    // trajectory and EVERY chain slot (the steps that did not run would otherwise keep a previous call's chain entries)
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (jcol + e < AF) {
        g_traj[(size_t)grow * AF + jcol + e] = __uint_as_float(0x7fc00000u);
        if (a.chains != nullptr)
          for (int sl = 0; sl < chain_len; ++sl) g_chains[((size_t)grow * chain_len + sl) * AF + jcol + e] = __uint_as_float(0x7fc00000u);
      }
  }
}

// Zeroes the exchange SLOTS in front of every launch -- not the 256-byte head of the block, whose first word is the sticky
// time-out word: only the host clears that, after reading it.  A kernel, not hipMemsetAsync: captured into a hipGraph, the memset node
// left 16 bytes of something else (a size and an address) at the head of the block from the second replay on (ROCm 7.2;
// tests/test_sampler_split.py::test_split_sampler_replays_from_a_hip_graph reads the time-out word there).
__global__ __launch_bounds__(256) void split_zero_kernel(u32x4* p, size_t n16) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n16) p[i] = (u32x4){0u, 0u, 0u, 0u};
}

// ------------------------------------------------------------------------------------------------
static int g_sampler_split = 1;  // tuning knob 27
void set_sampler_split(int v) { g_sampler_split = v; }
static int g_split_pre_sweep = 4;  // tuning knob 28: s_sleep(1) periods (64 cycles each) between a member's exchange store and its first sweep
void set_sampler_split_pre_sweep(int v) { g_split_pre_sweep = v < 0 ? 0 : (v > 64 ? 64 : v); }
static unsigned g_split_spin_limit = SPLIT_SPINS;  // tuning knob 29 (tests: 1 forces a time-out; <= 0 restores the default)
void set_sampler_split_spin_limit(int v) { g_split_spin_limit = v <= 0 ? SPLIT_SPINS : (unsigned)v; }

static int device_cus() {
  static std::atomic<int> cus[64];
  int d = 0;
  (void)hipGetDevice(&d);
  d &= 63;
  int v = cus[d].load(std::memory_order_relaxed);
  if (v == 0) {
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || v < 1) v = 1;
    cus[d].store(v, std::memory_order_relaxed);
  }
  return v;
}

bool sampler_split_ok(const dppo_net_desc& d, bool bf16, int64_t B, bool merge_top) {
  if (!g_sampler_split || !bf16 || !merge_top) return false;
  if (d.hidden != 512 || d.n_blocks != 1 || d.use_layernorm || d.plain) return false;
  if (d.out_dim > 64 || d.in_dim > 3 * BF16::KB) return false;
  if (d.act != ACT_RELU && d.act != ACT_MISH) return false;
  const int64_t tiles = (B + 15) / 16;
  return tiles * SPLIT <= device_cus();
}

size_t sampler_split_xch_bytes(const dppo_net_desc& d, int64_t B) {
  const size_t tiles = (size_t)((B + 15) / 16);
  const int ot = (d.out_dim + 15) / 16 <= 1 ? 1 : 4;
  // 256 bytes in front for the (sticky, host-cleared) time-out word, then the slots, which are zeroed before every launch
  return 256 + tiles * 2 * SPLIT * ot * 64 * 16;
}

template <int KS0V, int OT, int ACT>
static int launch_split_cfg(const SamplerGeom& g, const SampleArgs& a, void* xch, size_t xch_bytes, hipStream_t s) {
  typedef BF16 P;
  const size_t lds = (size_t)16 * a.Kp0 * P::ESIZE + 3 * (size_t)16 * 512 * P::ESIZE + 2 * (size_t)(2 * 512 + OT * 16) * 4 + 16 +
                     (size_t)a.n_steps * (sizeof(dppo_step) + (size_t)a.td * 4) + (size_t)16 * a.AF * 4;
  if (lds > 128 * 1024) return -1;  // a very long step table: the one-workgroup kernel reads it from memory
  auto kern = sample_chain_split_kernel<P, KS0V, OT, ACT>;
  static DevLatch attr_set;
  if (attr_set.need()) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    attr_set.done();
  }
  if ((xch_bytes & 15) || ((uintptr_t)xch & 15)) return -3;
  const size_t n16 = (xch_bytes - 256) / 16;  // the slots behind the head
  hipLaunchKernelGGL(split_zero_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, s, (u32x4*)xch + 16, n16);
  SampleArgs b = a;
  b.pre_sweep = g_split_pre_sweep;
  b.spin_limit = g_split_spin_limit;
  const int tiles = (a.B + 15) / 16;
  const bool probe = probe_begin(PROBE_SAMPLER, s);
  hipLaunchKernelGGL(kern, dim3((tiles + 7) / 8 * 64), dim3(256), lds, s, b, (char*)xch + 256, (int)(xch_bytes - 256), (unsigned*)xch);
  if (probe) probe_end(s, 2.0 * a.B * a.n_steps * ((double)g.in_dim * g.H + 2.0 * g.nb * g.H * g.H + (double)g.H * g.out_dim));
  return 0;
}

// 0 launched; -1 shape not covered (caller uses sample_chain_kernel); < -1 error
int launch_sample_chain_split(const SamplerGeom& g, const SampleArgs& a, void* xch, size_t xch_bytes, hipStream_t s) {
  const int ks0v = (g.in_dim + BF16::KB - 1) / BF16::KB;
  if (g.H != 512 || g.nb != 1 || !a.merge_top || a.use_ln || ks0v > 3 || g.OT > 4 || g.KS0 < ks0v || (a.td & 3)) return -1;
  const bool relu = a.act == ACT_RELU;
#define DPPO_SPLIT_CASE(K, O)                                                                      \
  if ((ks0v <= 2 ? 2 : 3) == K && g.OT == O)                                                       \
    return relu ? launch_split_cfg<K, O, ACT_RELU>(g, a, xch, xch_bytes, s)                        \
                : launch_split_cfg<K, O, ACT_MISH>(g, a, xch, xch_bytes, s);
  DPPO_SPLIT_CASE(2, 1)
  DPPO_SPLIT_CASE(2, 4)
  DPPO_SPLIT_CASE(3, 1)
  DPPO_SPLIT_CASE(3, 4)
#undef DPPO_SPLIT_CASE
  return -1;
}

}  // namespace dppo

#ifdef DPPO_STAMPS
extern "C" int dppo_debug_split_stamps(unsigned long long* out) {  // out: [4 waves][16]
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(dppo::g_split_stamps), sizeof(unsigned long long) * 4 * 16);
}
#endif
