// See sampler.h.  gfx950 only.  Compiled with -ffp-contract=off so the posterior arithmetic is
// the reference's op sequence (separate multiplies/adds, no fused contraction).
#include "sampler.h"
#include "gemm.h"
#include "tile_ln.h"
#include "pack_dev.h"

namespace dppo {

// LDS activation images: row-major [16 rows][rb bytes]; 16-byte chunk c of row q is stored at chunk
// c ^ (q & kmask).  kmask = min(16, largest power of two dividing rb/16) - 1, so the XOR never leaves
// the row.  With rb a multiple of 256 (every hidden buffer) kmask = 15 and the B-fragment
// ds_read_b128 (lane (r,g) reads chunk 4*ks+g of row r) is conflict-free.
__device__ __forceinline__ int kmask_of(int rb) {
  const int n = rb >> 4;
  const int p = n & (-n);
  return (p > 16 ? 16 : p) - 1;
}
template <class P>
__device__ __forceinline__ void lds_put(char* buf, int rb, int kmask, int row, int col, float v) {
  const int byte = col * P::ESIZE;
  char* p = buf + row * rb + ((((byte >> 4) ^ (row & kmask)) << 4) | (byte & 15));
  *(typename P::elem_t*)p = P::from_f32(v);
}

// Pointers that arrive inside the by-value argument struct and are then picked by a run-time index (ws[net], params[net]
// ...) reach the loads as GENERIC pointers: the compiler emitted flat_load for every weight fragment of the ring, and a
// FLAT access ticks both memory counters and may complete out of order with LDS traffic, so each use was preceded by
// s_waitcnt vmcnt(0) lgkmcnt(0) -- the whole ring drained at every k-step group and the prefetch distance was zero.
// Everything the step loop touches in global memory goes through explicit global-address-space pointers instead.
typedef const __attribute__((address_space(1))) u32x4* gfrag_p;
typedef const __attribute__((address_space(1))) float* gfloat_p;
typedef __attribute__((address_space(1))) float* gfloat_w;

template <class P, int TPW, int OT, bool LN, int ACT, int PDX = sampler_pd(128 * TPW)>
__global__ __launch_bounds__(512) void sample_chain_kernel(const SampleArgs a) {
  // PDX: ring depth.  (A depth of 8 at H = 512 -- legal once layer 0 is LDS-resident and never passes through the ring --
  // spills 98 VGPRs: the ring alone would be 128 registers.)
  constexpr int PD = PDX, ES = P::ESIZE, KB = P::KB;
  constexpr int H = 128 * TPW, KSH = H / KB, CNT = (KSH + SAMPLER_WAVES - 1) / SAMPLER_WAVES;
  constexpr int HRB = H * ES;  // hidden row bytes
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int grow0 = blockIdx.x * 16;
  const int AF = a.AF, td = a.td, cond = a.cond, Kp0 = a.Kp0, nb = a.nb, B = a.B;
  const int in_rb = Kp0 * ES, in_km = kmask_of(in_rb);
  const int KS0 = Kp0 / KB;
  const int total = KS0 + 2 * nb * KSH;

  // merge: the top block's second layer is folded into the out layer (see SampleArgs::merge_top); bufC then holds the raw
  // input h_in of that block and the out layer reads bufC and bufB
  constexpr bool MERGE_OK = OT < 8;  // compiled out for wide outputs (see the launcher): no register cost there
  const bool merge = MERGE_OK && a.merge_top != 0;
  char* xin = smem;
  char* bufA = xin + 16 * in_rb;
  char* bufB = bufA + 16 * HRB;
  char* bufC = bufB + 16 * HRB;
  float* xcur = (float*)(bufC + (merge ? 16 * HRB : 0));  // [16][AF]
  // out-layer partials [8 waves][OT*16 features][16 rows]: bufB is idle while the out layer reads bufA, reuse it when it fits
  constexpr bool PART_FITS_B = SAMPLER_WAVES * OT * 16 * 16 * 4 <= 16 * HRB;
  const bool part_in_b = PART_FITS_B && !merge;
  float* part = part_in_b ? (float*)bufB : xcur + ((16 * AF + 3) & ~3);
  float* lnred = (part_in_b ? xcur + ((16 * AF + 3) & ~3) : part + SAMPLER_WAVES * OT * 16 * 16);  // [8][16]
  // both networks' biases, staged once: [net][(1 + 2 nb) * H hidden | OT*16 out].  A global load at a layer's start
  // would wait behind every weight fragment the ring has in flight (vmcnt is in issue order), 3 x 20 times per call.
  float* biasL = lnred + LN_WAVES * 16;
  const int bias_stride = (1 + 2 * nb) * H + OT * 16;
  const bool bias_lds = a.consts_lds != 0;
  if (bias_lds) {
    for (int net = 0; net < 2; ++net) {
      for (int idx = tid; idx < (1 + 2 * nb) * H; idx += 512)
        biasL[net * bias_stride + idx] = a.params[net][a.bias_off[idx / H] + idx % H];
      for (int idx = tid; idx < OT * 16; idx += 512)
        biasL[net * bias_stride + (1 + 2 * nb) * H + idx] =
            idx < AF ? (merge ? a.cbias[net][idx] : a.params[net][a.bias_off[1 + 2 * nb] + idx]) : 0.f;
    }
  }

  // ---- one-time: state columns (of the first step's network) + zero padding of the input image, x_K, time
  // embedding of step 0
  auto put_state = [&](int net) {
    const float* ob = a.obs[net];
    for (int idx = tid; idx < 16 * Kp0; idx += 512) {
      const int row = idx / Kp0, c = idx - row * Kp0;
      if (c >= AF + td) {
        const int j = c - AF - td;
        const int grow = min(grow0 + row, B - 1);
        lds_put<P>(xin, in_rb, in_km, row, c, j < cond ? ob[(size_t)grow * a.ld_obs + j] : 0.f);
      }
    }
  };
  put_state(a.sched[0].net);
  {
    const dppo_step s0 = a.sched[0];
    for (int idx = tid; idx < 16 * AF; idx += 512) {
      const int row = idx / AF, j = idx - row * AF;
      const int grow = grow0 + row;
      const size_t ni = (size_t)min(grow, B - 1) * AF + j;
      const float v = a.noise != nullptr ? a.noise[ni] : philox_normal(ni, a.seed_lo, a.seed_hi);
      xcur[row * AF + j] = v;
      lds_put<P>(xin, in_rb, in_km, row, j, v);
      if (a.init_slot >= 0 && a.chains != nullptr && grow < B) a.chains[((size_t)grow * a.chain_len + a.init_slot) * AF + j] = v;
    }
    for (int idx = tid; idx < 16 * td; idx += 512) {
      const int row = idx / td, j = idx - row * td;
      lds_put<P>(xin, in_rb, in_km, row, AF + j, a.temb[s0.net][s0.t * td + j]);
    }
  }

  const size_t wave_stride = (size_t)total * TPW * 64;
  const gfrag_p ws[2] = {(gfrag_p)(a.wstream[0] + wid * wave_stride + lane), (gfrag_p)(a.wstream[1] + wid * wave_stride + lane)};
  const gfrag_p os[2] = {(gfrag_p)(a.ostream[0] + (size_t)wid * CNT * OT * 64 + lane),
                         (gfrag_p)(a.ostream[1] + (size_t)wid * CNT * OT * 64 + lane)};
  const gfrag_p os2[2] = {(gfrag_p)(a.ostream2[0] + (size_t)wid * CNT * OT * 64 + lane),
                          (gfrag_p)(a.ostream2[1] + (size_t)wid * CNT * OT * 64 + lane)};
  const gfloat_p g_params[2] = {(gfloat_p)a.params[0], (gfloat_p)a.params[1]};
  const gfloat_p g_temb[2] = {(gfloat_p)a.temb[0], (gfloat_p)a.temb[1]};
  const gfloat_p g_cbias[2] = {(gfloat_p)a.cbias[0], (gfloat_p)a.cbias[1]};
  const gfloat_p g_noise = (gfloat_p)a.noise;
  const gfloat_w g_chains = (gfloat_w)a.chains, g_traj = (gfloat_w)a.traj;
  const int total_eff = total - (merge ? KSH : 0);  // the ring never sees the top block's second layer when merged
  const int wbase = wid * 16 * TPW;  // this wave's feature slice; the lane's features: wbase + feat_off<P>(g, tp) + e

  // Layer 0 is 4 of the 36 k-step positions a step streams at H = 512 (2 of them zero padding): its fragments of the
  // network in use stay in LDS (each wave keeps its own slice: no barrier around a reload, which happens only when the
  // step table switches network) and the ring skips those positions.
  const bool l0_lds = a.l0_lds != 0;
  const int ks0v = a.ks0v, skip = l0_lds ? KS0 : 0;
  char* w0L = (char*)(biasL + (bias_lds ? 2 * bias_stride : 0)) + (size_t)wid * ks0v * TPW * 1024;
  auto load_w0 = [&](int net) {
    for (int q = 0; q < ks0v * TPW; ++q) *(u32x4*)(w0L + ((size_t)q * 64 + lane) * 16) = ws[net][(size_t)q * 64];
  };
  if (l0_lds) load_w0(a.sched[0].net);

  u32x4 ring[PD][TPW];
  {
    const int n0 = a.sched[0].net;
#pragma unroll
    for (int p = 0; p < PD; ++p)
#pragma unroll
      for (int tp = 0; tp < TPW; ++tp) ring[p][tp] = ws[n0][((size_t)(skip + p) * TPW + tp) * 64];
  }
  __syncthreads();

  for (int i = 0; i < a.n_steps; ++i) {
    const dppo_step st = a.sched[i];
    const int net = st.net;
    const int nnet = (i + 1 < a.n_steps) ? a.sched[i + 1].net : net;
    const gfrag_p cur = ws[net];
    const gfrag_p nxt = ws[nnet];
    const gfloat_p prm = g_params[net];
    const float* prm_ln = a.params[net];  // (LayerNorm parameters go through tile_ln.h's generic-pointer interface)
    // this step's noise draw and the next step's time embedding are needed at the very end of the step, right behind the
    // ring's prefetches: issued here, they have a whole step to land; issued there, they would wait for every weight
    // fragment in flight (vmcnt is in issue order) -- 20 times per call
    const dppo_step sn = a.sched[min(i + 1, a.n_steps - 1)];
    float z_pre = 0.f, te_pre = 0.f;
    if (tid < 16 * AF) {
      const int row = tid / AF, j = tid - row * AF;
      const size_t ni = (size_t)(i + 1) * B * AF + (size_t)min(grow0 + row, B - 1) * AF + j;
      z_pre = a.noise != nullptr ? g_noise[ni] : philox_normal(ni, a.seed_lo, a.seed_hi);
    }
    if (tid < 16 * td) te_pre = g_temb[sn.net][sn.t * td + tid % td];

    // out-layer fragments, in groups of OG out tiles: one group, prefetched at the top of the step, when it is small;
    // loaded right before use at H = 1024 (the registers are needed) and for wide outputs (two groups)
    constexpr int OG = OT < 4 ? OT : 4;
    constexpr bool OF_EARLY = TPW < 8 && OT <= 4;
    u32x4 of[CNT][OG], of2[CNT][OG];
    if constexpr (OF_EARLY) {
#pragma unroll
      for (int c = 0; c < CNT; ++c)
#pragma unroll
        for (int to = 0; to < OG; ++to) {
          of[c][to] = os[net][(c * OT + to) * 64];
          if (merge) of2[c][to] = os2[net][(c * OT + to) * 64];
        }
    }

    f32x4 h[TPW][1], acc[TPW][1];
    int pos = 0;

    // one hidden layer: nks k-step positions starting at stream position `pos`, B operand from `src`
    auto run_layer = [&](const char* src, int rb, int km, int nks, int layer) {
      if (bias_lds) {
#pragma unroll
        for (int tp = 0; tp < TPW; ++tp)
          acc[tp][0] = *(const f32x4*)(biasL + net * bias_stride + layer * H + wbase + feat_off<P>(g, tp));
      } else {
        const int boff = a.bias_off[layer];
#pragma unroll
        for (int tp = 0; tp < TPW; ++tp)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[tp][0][e] = prm[boff + wbase + feat_off<P>(g, tp) + e];
      }
      for (int k0 = 0; k0 < nks; k0 += PD) {
#pragma unroll
        for (int p = 0; p < PD; ++p) {
          const int ks = k0 + p;
          const u32x4 xb = *(const u32x4*)(src + r * rb + (((ks * 4 + g) ^ (r & km)) << 4));
#pragma unroll
          for (int tp = 0; tp < TPW; ++tp) acc[tp][0] = P::mma(ring[p][tp], xb, acc[tp][0]);
          // refill the slot just consumed with the fragments PD positions ahead (next layer / next step included)
          const int np = pos + ks + PD;
          const gfrag_p src_w =
              np < total_eff ? cur + (size_t)np * TPW * 64 : nxt + (size_t)(np - total_eff + skip) * TPW * 64;
#pragma unroll
          for (int tp = 0; tp < TPW; ++tp) ring[p][tp] = src_w[tp * 64];
        }
      }
      pos += nks;
    };
    // write this lane's 4*TPW features of batch row r (optionally activated) into an LDS image
    auto put_hidden = [&](char* dst, const f32x4 (&v)[TPW][1], int actk) {
      with_act(actk, [&](auto tag) {  // the activation kind resolved once per call, not per element
        constexpr int AK = decltype(tag)::value;
        if constexpr (ES == 4) {
#pragma unroll
          for (int tp = 0; tp < TPW; ++tp) {
            const int c = ((wbase + feat_off<P>(g, tp)) * 4) >> 4;
            float4 o = make_float4(act_c<AK>(v[tp][0][0]), act_c<AK>(v[tp][0][1]), act_c<AK>(v[tp][0][2]),
                                   act_c<AK>(v[tp][0][3]));
            *(float4*)(dst + r * HRB + ((c ^ (r & 15)) << 4)) = o;
          }
        } else {
#pragma unroll
          for (int tp = 0; tp < TPW; tp += 2) {
            const int c = ((wbase + feat_off<P>(g, tp)) * 2) >> 4;
            u32x4 o;
            o.x = (uint32_t)f2bf(act_c<AK>(v[tp][0][0])) | ((uint32_t)f2bf(act_c<AK>(v[tp][0][1])) << 16);
            o.y = (uint32_t)f2bf(act_c<AK>(v[tp][0][2])) | ((uint32_t)f2bf(act_c<AK>(v[tp][0][3])) << 16);
            if constexpr (TPW >= 2) {
              o.z = (uint32_t)f2bf(act_c<AK>(v[tp + 1][0][0])) | ((uint32_t)f2bf(act_c<AK>(v[tp + 1][0][1])) << 16);
              o.w = (uint32_t)f2bf(act_c<AK>(v[tp + 1][0][2])) | ((uint32_t)f2bf(act_c<AK>(v[tp + 1][0][3])) << 16);
              *(u32x4*)(dst + r * HRB + ((c ^ (r & 15)) << 4)) = o;
            } else {
              *(u32x2*)(dst + r * HRB + ((c ^ (r & 15)) << 4)) = (u32x2){o.x, o.y};
            }
          }
        }
      });
    };

    // LayerNorm variant of a block input: bufA <- act(LN1_b(h)) (acc is scratch here)
    float ln_m[1], ln_r[1];
    auto put_block_input = [&](int b) {
      if constexpr (LN) {
        ln_forward<P, TPW, 1>(h, acc, prm_ln + a.ln_off[4 * b], prm_ln + a.ln_off[4 * b + 1], H, wbase, g, r, wid, lnred, ln_m,
                              ln_r);
        put_hidden(bufA, acc, ACT);
      } else {
        put_hidden(bufA, h, ACT);
      }
    };
    // ---- layer 0
    if (l0_lds) {
      if (bias_lds) {
#pragma unroll
        for (int tp = 0; tp < TPW; ++tp) acc[tp][0] = *(const f32x4*)(biasL + net * bias_stride + wbase + feat_off<P>(g, tp));
      } else {
        const int boff = a.bias_off[0];
#pragma unroll
        for (int tp = 0; tp < TPW; ++tp)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[tp][0][e] = prm[boff + wbase + feat_off<P>(g, tp) + e];
      }
      for (int ks = 0; ks < ks0v; ++ks) {
        const u32x4 xb = *(const u32x4*)(xin + r * in_rb + (((ks * 4 + g) ^ (r & in_km)) << 4));
#pragma unroll
        for (int tp = 0; tp < TPW; ++tp)
          acc[tp][0] = P::mma(*(const u32x4*)(w0L + ((size_t)(ks * TPW + tp) * 64 + lane) * 16), xb, acc[tp][0]);
      }
      pos = KS0;
    } else {
      run_layer(xin, in_rb, in_km, KS0, 0);
    }
#pragma unroll
    for (int tp = 0; tp < TPW; ++tp) h[tp][0] = acc[tp][0];
    if (nb > 0) {
      put_block_input(0);
      if (merge && nb == 1) put_hidden(bufC, h, ACT_NONE);  // raw input of the top block, for the merged out layer
    } else {
      put_hidden(bufA, h, ACT_NONE);
    }
    __syncthreads();
    // ---- residual blocks: h += l2(act([LN2] l1(act([LN1] h))))
    for (int b = 0; b < nb; ++b) {
      run_layer(bufA, HRB, 15, KSH, 1 + 2 * b);
      if constexpr (LN)
        ln_forward<P, TPW, 1>(acc, acc, prm_ln + a.ln_off[4 * b + 2], prm_ln + a.ln_off[4 * b + 3], H, wbase, g, r, wid, lnred,
                              ln_m, ln_r);
      put_hidden(bufB, acc, ACT);
      __syncthreads();
      if (merge && b == nb - 1) break;  // W2 of the top block only ever feeds the out layer: folded into it
      run_layer(bufB, HRB, 15, KSH, 2 + 2 * b);
#pragma unroll
      for (int tp = 0; tp < TPW; ++tp) h[tp][0] += acc[tp][0];
      if (b + 1 < nb) {
        put_block_input(b + 1);
        if (merge && b + 1 == nb - 1) put_hidden(bufC, h, ACT_NONE);
      } else {
        put_hidden(bufA, h, ACT_NONE);
      }
      __syncthreads();
    }
    // ---- output layer: K split over the 8 waves, partial tiles reduced through LDS
    {
#pragma unroll
      for (int og = 0; og < OT; og += OG) {
        if constexpr (!OF_EARLY) {
#pragma unroll
          for (int c = 0; c < CNT; ++c)
#pragma unroll
            for (int to = 0; to < OG; ++to) {
              of[c][to] = os[net][(c * OT + og + to) * 64];
              if (merge) of2[c][to] = os2[net][(c * OT + og + to) * 64];
            }
        }
        f32x4 oacc[OG];
#pragma unroll
        for (int to = 0; to < OG; ++to) oacc[to] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < CNT; ++c) {
          const int ks = wid * CNT + c;
          if (ks < KSH) {
            const u32x4 xb = *(const u32x4*)((merge ? bufC : bufA) + r * HRB + (((ks * 4 + g) ^ (r & 15)) << 4));
#pragma unroll
            for (int to = 0; to < OG; ++to) oacc[to] = P::mma(of[c][to], xb, oacc[to]);
            if (merge) {
              const u32x4 xb2 = *(const u32x4*)(bufB + r * HRB + (((ks * 4 + g) ^ (r & 15)) << 4));
#pragma unroll
              for (int to = 0; to < OG; ++to) oacc[to] = P::mma(of2[c][to], xb2, oacc[to]);
            }
          }
        }
#pragma unroll
        for (int to = 0; to < OG; ++to)
#pragma unroll
          for (int e = 0; e < 4; ++e) part[(wid * OT * 16 + (og + to) * 16 + 4 * g + e) * 16 + r] = oacc[to][e];
      }
    }
    __syncthreads();
    // ---- posterior + noise: diffusion_vpg.py:165-223 (p_mean_var) and :279-311 (sampling loop)
    {
      const size_t nz0 = (size_t)(i + 1) * B * AF;
      for (int idx = tid; idx < 16 * AF; idx += 512) {
        const int row = idx / AF, j = idx - row * AF;
        const int grow = grow0 + row;
        float eps = bias_lds ? biasL[net * bias_stride + (1 + 2 * nb) * H + j]
                             : (merge ? g_cbias[net][j] : prm[a.bias_off[1 + 2 * nb] + j]);
#pragma unroll
        for (int w = 0; w < SAMPLER_WAVES; ++w) eps += part[(w * OT * 16 + j) * 16 + row];
        const float x = xcur[row * AF + j];
        float x0, mu;
        if (!a.use_ddim) {
          x0 = st.c0 * x - st.c1 * eps;
          if (a.has_dclip) x0 = fminf(fmaxf(x0, -a.dclip), a.dclip);
          mu = st.c2 * x0 + st.c3 * x;
        } else {
          x0 = (x - st.c1 * eps) / st.c0;
          if (a.has_dclip) {
            x0 = fminf(fmaxf(x0, -a.dclip), a.dclip);
            eps = (x - st.c0 * x0) / st.c1;
          }
          if (a.has_eclip) eps = fminf(fmaxf(eps, -a.eclip), a.eclip);
          mu = st.c2 * x0 + st.c3 * eps;
        }
        float z = z_pre;  // first pass: prefetched / drawn at the step's top
        if (idx != tid) {
          const size_t ni = nz0 + (size_t)min(grow, B - 1) * AF + j;
          z = a.noise != nullptr ? g_noise[ni] : philox_normal(ni, a.seed_lo, a.seed_hi);
        }
        z = fminf(fmaxf(z, -a.rclip), a.rclip);
        float xn = mu + st.std * z;
        if (st.final_clip) xn = fminf(fmaxf(xn, -a.fclip), a.fclip);
        xcur[row * AF + j] = xn;
        lds_put<P>(xin, in_rb, in_km, row, j, xn);
        if (grow < B) {
          if (st.chain_slot >= 0 && a.chains != nullptr) g_chains[((size_t)grow * a.chain_len + st.chain_slot) * AF + j] = xn;
          if (i + 1 == a.n_steps) g_traj[(size_t)grow * AF + j] = xn;
        }
      }
      for (int idx = tid; idx < 16 * td; idx += 512) {
        const int row = idx / td, j = idx - row * td;
        lds_put<P>(xin, in_rb, in_km, row, AF + j, idx == tid ? te_pre : g_temb[sn.net][sn.t * td + j]);
      }
      // a cond_mlp encodes the observation per network: swap the state columns when the next step switches network
      if (sn.net != st.net && a.obs[0] != a.obs[1]) put_state(sn.net);
      if (l0_lds && sn.net != st.net) load_w0(sn.net);  // wave-private slice: the wave that wrote it is the one that reads it
    }
    __syncthreads();
  }
}

template <class P>
SamplerGeom sampler_geom(const dppo_net_desc& d) {
  SamplerGeom g;
  g.H = d.hidden;
  g.nb = d.n_blocks;
  g.in_dim = d.in_dim;
  g.out_dim = d.out_dim;
  g.Kp0 = round_up(d.in_dim, sampler_pd(d.hidden) * P::KB);
  g.KS0 = g.Kp0 / P::KB;
  g.KSH = g.H / P::KB;
  g.TPW = g.H / 128;
  const int ot = (d.out_dim + 15) / 16;
  g.OT = ot <= 1 ? 1 : (ot <= 4 ? 4 : 8);
  g.CNT = (g.KSH + SAMPLER_WAVES - 1) / SAMPLER_WAVES;
  g.total_pos = g.KS0 + 2 * g.nb * g.KSH;
  g.hidden_frags_per_wave = (size_t)g.total_pos * g.TPW;
  g.out_frags_per_wave = (size_t)g.CNT * g.OT;
  return g;
}
template SamplerGeom sampler_geom<F32>(const dppo_net_desc&);
template SamplerGeom sampler_geom<BF16>(const dppo_net_desc&);

static int g_sampler_l0_lds = 1;  // tuning knob 15: layer-0 weight fragments of the sampler resident in LDS
void set_sampler_l0_lds(int v) { g_sampler_l0_lds = v; }

template <class P, int TPW, int OT, bool LN, int ACT>
static int launch_cfg(const SamplerGeom& g, const SampleArgs& a, hipStream_t s) {
  const int ES = P::ESIZE;
  const size_t part_bytes = (size_t)SAMPLER_WAVES * OT * 16 * 16 * 4;
  auto lds_of = [&](bool merge) {
    return (size_t)16 * a.Kp0 * ES + (merge ? 3 : 2) * (size_t)16 * g.H * ES + (size_t)((16 * a.AF + 3) & ~3) * 4 +
           (part_bytes <= (size_t)16 * g.H * ES && !merge ? 0 : part_bytes) + (size_t)LN_WAVES * 16 * 4;
  };
  SampleArgs b = a;
  // (wide outputs, OT = 8: the second set of out-layer fragments does not fit the registers next to the ring -- measured
  // slower at Ta*Da = 112 -- so the merge stays off there)
  if (b.merge_top && (OT >= 8 || lds_of(true) > 160 * 1024)) b.merge_top = 0;
  size_t lds = lds_of(b.merge_top != 0);
  if (lds > 160 * 1024) return -2;
  const size_t bias_bytes = 2 * ((size_t)(1 + 2 * a.nb) * g.H + OT * 16) * 4;
  b.consts_lds = lds + bias_bytes <= 160 * 1024 ? 1 : 0;
  if (b.consts_lds) lds += bias_bytes;
  b.ks0v = (g.in_dim + P::KB - 1) / P::KB;
  const size_t w0_bytes = (size_t)SAMPLER_WAVES * b.ks0v * TPW * 1024;
  b.l0_lds = g_sampler_l0_lds && lds + w0_bytes <= 160 * 1024 ? 1 : 0;
  if (b.l0_lds) lds += w0_bytes;
  static DevLatch attr_set;  // per device; raising the dynamic-LDS cap is idempotent, racing setters are harmless
  auto kern = sample_chain_kernel<P, TPW, OT, LN, ACT>;
  if (attr_set.need()) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set.done();
  }
  const bool probe = probe_begin(PROBE_SAMPLER, s);
  hipLaunchKernelGGL(kern, dim3((a.B + 15) / 16), dim3(512), lds, s, b);
  if (probe) probe_end(s, 2.0 * a.B * a.n_steps * ((double)g.in_dim * g.H + 2.0 * g.nb * g.H * g.H + (double)g.H * g.out_dim));
  return 0;
}

template <class P>
int launch_sample_chain(const SamplerGeom& g, const SampleArgs& a, hipStream_t s) {
  const bool relu = a.act == ACT_RELU;  // check_net admits ReLU and Mish only
#define DPPO_CASE(T, O) \
  if (g.TPW == T && g.OT == O)                                                                                      \
    return a.use_ln ? (relu ? launch_cfg<P, T, O, true, ACT_RELU>(g, a, s) : launch_cfg<P, T, O, true, ACT_MISH>(g, a, s)) \
                    : (relu ? launch_cfg<P, T, O, false, ACT_RELU>(g, a, s) : launch_cfg<P, T, O, false, ACT_MISH>(g, a, s));
  DPPO_CASE(2, 1)
  DPPO_CASE(2, 4)
  DPPO_CASE(4, 1)
  DPPO_CASE(4, 4)
  DPPO_CASE(8, 1)
  DPPO_CASE(8, 4)
  DPPO_CASE(2, 8)
  DPPO_CASE(4, 8)
  DPPO_CASE(8, 8)
  DPPO_CASE(6, 1)  // hidden 768: the robomimic square / transport image actors (mlp_dims [768, 768, 768])
  DPPO_CASE(6, 4)
  DPPO_CASE(6, 8)
#undef DPPO_CASE
  return -1;
}
template int launch_sample_chain<F32>(const SamplerGeom&, const SampleArgs&, hipStream_t);
template int launch_sample_chain<BF16>(const SamplerGeom&, const SampleArgs&, hipStream_t);

// ------------------------------------------------------------------------------------------------
// fragment-stream packing
// ------------------------------------------------------------------------------------------------
// hidden stream: [wave w][position][tp][lane] u32x4.  Lane (r,g) of tile tp holds
// W[feature(w,tp,r)][ks*KB + (16/ES)*g + 0..], feature(w,tp,i) = w*16*TPW + feat_off(i>>2, tp) + (i&3)  (common.h).
template <class P>
__global__ void pack_hidden_kernel(const float* W, int in_valid, int ld, int KS, int TPW, int pos0, int total_pos,
                                   u32x4* stream) {
  const int lane = threadIdx.x & 63;
  const int tp = blockIdx.x % TPW;
  const int ks = (blockIdx.x / TPW) % KS;
  const int w = blockIdx.x / (TPW * KS);
  const int r = lane & 15, g = lane >> 4;
  const int feat = w * 16 * TPW + feat_off<P>(r >> 2, tp) + (r & 3);
  constexpr int EPL = 16 / P::ESIZE;  // elements per lane
  const int k0 = ks * P::KB + EPL * g;
  uint32_t out[4];
  if constexpr (P::ESIZE == 4) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k0 + j;
      out[j] = __float_as_uint(k < in_valid ? W[(size_t)feat * ld + k] : 0.f);
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k0 + 2 * j;
      const float lo = k < in_valid ? W[(size_t)feat * ld + k] : 0.f;
      const float hi = k + 1 < in_valid ? W[(size_t)feat * ld + k + 1] : 0.f;
      out[j] = (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
    }
  }
  stream[(((size_t)w * total_pos + pos0 + ks) * TPW + tp) * 64 + lane] = (u32x4){out[0], out[1], out[2], out[3]};
}
template <class P>
void launch_pack_hidden(const float* W, int H, int in_valid, int ld, int KS, int TPW, int pos0, int total_pos,
                        u32x4* stream, hipStream_t s) {
  hipLaunchKernelGGL((pack_hidden_kernel<P>), dim3(SAMPLER_WAVES * KS * TPW), dim3(64), 0, s, W, in_valid, ld, KS, TPW,
                     pos0, total_pos, stream);
}
template void launch_pack_hidden<F32>(const float*, int, int, int, int, int, int, int, u32x4*, hipStream_t);
template void launch_pack_hidden<BF16>(const float*, int, int, int, int, int, int, int, u32x4*, hipStream_t);

// out stream: [wave w][c][to][lane]; wave w owns k-steps w*CNT + c; rows >= out_dim are zero.
template <class P>
__global__ void pack_out_kernel(const float* W, int out_dim, int H, int OT, int CNT, u32x4* stream) {
  pack_out_block<P>(W, out_dim, H, OT, CNT, stream, blockIdx.x);
}
template <class P>
void launch_pack_out(const float* W, int out_dim, int H, int OT, int CNT, u32x4* stream, hipStream_t s) {
  hipLaunchKernelGGL((pack_out_kernel<P>), dim3(SAMPLER_WAVES * CNT * OT), dim3(64), 0, s, W, out_dim, H, OT, CNT,
                     stream);
}
template void launch_pack_out<F32>(const float*, int, int, int, int, u32x4*, hipStream_t);
template void launch_pack_out<BF16>(const float*, int, int, int, int, u32x4*, hipStream_t);

}  // namespace dppo
