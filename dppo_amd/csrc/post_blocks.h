// Device code shared by the reduction / post-reduction launches (ppo.hip: tail_reduce_kernel, post_reduce_kernel) and the
// riders of the weight-gradient GEMM launch (gemm.hip, GemmTNExtra): the many-slab reduction of the in-kernel dW0, the
// time-embedding gradient (G = W0_temb^T . S, the time MLP's backward) and the time columns of dW0.
//
// Riding (PostReduce::wait_cnt != null, SlabJob results stored with `wt`): producers and consumers are workgroups of ONE launch,
// dispatched in index order with the producers first.  A producer stores its results write-through (agent-scope relaxed atomic
// stores = sc1), drains them, and thread 0 adds 1 to the counter; a consumer polls the counter (bounded) and reads the producers'
// results with agent-scope relaxed loads (sc1) -- the fence-free hand-over of common.h (DPPO_HANDOVER_*), release / acquire on
// targets where that shortcut is not an ISA property.
#pragma once
#include "common.h"
#include "ppo.h"

namespace dppo {

__device__ __forceinline__ float sinus_feat(int t, int j, int td) {
  const int half = td / 2;
  const float step = (float)(-(log(10000.0) / (double)(half - 1)));  // scalar cast to f32 like torch does
  const int jj = j < half ? j : j - half;
  const float ang = (float)t * expf((float)jj * step);
  return j < half ? sinf(ang) : cosf(ang);
}

// a value another workgroup of this launch may have produced (riding) or an earlier launch did (plain load)
__device__ __forceinline__ float post_in(const PostReduce& q, const float* p) {
  return q.wait_cnt != nullptr ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
}
// riding: wait (bounded: a logic error must fail a test, not hang the GPU) until `need` producer workgroups have arrived
__device__ __forceinline__ void post_wait(const PostReduce& q) {
  if (q.wait_cnt == nullptr) return;
  if (threadIdx.x == 0) {
    for (int spin = 0; spin < (1 << 22); ++spin) {
      if (__hip_atomic_load(q.wait_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)q.wait_need) break;
      __builtin_amdgcn_s_sleep(4);
    }
    DPPO_HANDOVER_ACQUIRE();
  }
  __syncthreads();
}
// a producer workgroup has stored its results (write-through): drain, then one arrival
__device__ __forceinline__ void post_arrive(unsigned* cnt) {
  DPPO_HANDOVER_DRAIN();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt, 1u, DPPO_HANDOVER_ARRIVE_ORDER, __HIP_MEMORY_SCOPE_AGENT);
}

// single block: recompute the tiny time MLP per fine-tuned step and back-propagate G[k][td] through it
// part A: what does not depend on the gradient G -- weights staged in LDS, sinusoidal features, z1 and a1 of every step
__device__ __forceinline__ void time_backward_prepare(const float* w1, const float* b1, const float* w2,
                                                      const dppo_step* ksteps, int Kft, int td, float* sh) {
  // sh: per k: e0[td], z1[2td], a1[2td], dz1[2td]; then w1[2td][td], w2[td][2td], b1[2td], G[Kft][td] staged once (each
  // phase below otherwise pays an L2 latency per inner-loop iteration: this block is the tail of the update's critical path)
  const int per = 7 * td;
  const int tid = threadIdx.x;
  float* w1s = sh + Kft * per;
  float* w2s = w1s + 2 * td * td;
  float* b1s = w2s + 2 * td * td;
  for (int i = tid; i < 2 * td * td; i += 256) w1s[i] = w1[i], w2s[i] = w2[i];
  for (int i = tid; i < 2 * td; i += 256) b1s[i] = b1[i];
  for (int i = tid; i < Kft * td; i += 256) {
    const int k = i / td, j = i % td;
    sh[k * per + j] = sinus_feat(ksteps[k].t, j, td);
  }
  __syncthreads();
  for (int i = tid; i < Kft * 2 * td; i += 256) {
    const int k = i / (2 * td), o = i % (2 * td);
    float s = b1s[o];
    for (int j = 0; j < td; ++j) s += w1s[o * td + j] * sh[k * per + j];
    sh[k * per + td + o] = mish_grad_f(s);  // (only the derivative of z1 is needed below)
    sh[k * per + 3 * td + o] = mish_f(s);
  }
}
// part B: G[Kft][td] -> gradients of the four parameter tensors
// last_is_total: G_in's last row holds the sum over ALL steps (PostReduce::S_rest): the other rows are subtracted here first
__device__ __forceinline__ void time_backward_finish(const float* G_in, int Kft, int td, float* gw1, float* gb1, float* gw2,
                                                     float* gb2, float* sh, bool last_is_total = false) {
  const int per = 7 * td;
  const int tid = threadIdx.x;
  const float* w2s = sh + Kft * per + 2 * td * td;
  float* Gs = sh + Kft * per + 4 * td * td + 2 * td;
  for (int i = tid; i < Kft * td; i += 256)  // sc1 loads: G may have been written by other workgroups of this launch
    Gs[i] = __hip_atomic_load(&G_in[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const float* G = Gs;
  __syncthreads();
  if (last_is_total) {
    for (int j = tid; j < td; j += 256) {
      float s = Gs[(Kft - 1) * td + j];
      for (int k = 0; k < Kft - 1; ++k) s -= Gs[k * td + j];
      Gs[(Kft - 1) * td + j] = s;
    }
    __syncthreads();
  }
  for (int i = tid; i < Kft * 2 * td; i += 256) {
    const int k = i / (2 * td), o = i % (2 * td);
    float s = 0.f;
    for (int j = 0; j < td; ++j) s += w2s[j * 2 * td + o] * G[k * td + j];
    sh[k * per + 5 * td + o] = s * sh[k * per + td + o];
  }
  __syncthreads();
  for (int i = tid; i < td * 2 * td; i += 256) {  // gw2[o][j] = sum_k G[k][o] a1[k][j]
    const int o = i / (2 * td), j = i % (2 * td);
    float s = 0.f;
    for (int k = 0; k < Kft; ++k) s += G[k * td + o] * sh[k * per + 3 * td + j];
    gw2[i] = s;
  }
  for (int o = tid; o < td; o += 256) {
    float s = 0.f;
    for (int k = 0; k < Kft; ++k) s += G[k * td + o];
    gb2[o] = s;
  }
  for (int i = tid; i < 2 * td * td; i += 256) {  // gw1[o][j] = sum_k dz1[k][o] e0[k][j]
    const int o = i / td, j = i % td;
    float s = 0.f;
    for (int k = 0; k < Kft; ++k) s += sh[k * per + 5 * td + o] * sh[k * per + j];
    gw1[i] = s;
  }
  for (int o = tid; o < 2 * td; o += 256) {
    float s = 0.f;
    for (int k = 0; k < Kft; ++k) s += sh[k * per + 5 * td + o];
    gb1[o] = s;
  }
}

// dW0[h][AF + j] = sum_k S[h][k] elem(temb[t_k][j]) (PostReduce::dW0t): one thread per output
__device__ __forceinline__ void dw0_temb_block(const PostReduce& q, int b) {
  const int out = b * 256 + threadIdx.x;
  if (out >= q.H * q.td) return;
  const int h = out / q.td, j = out - h * q.td;
  float acc = 0.f, rest = q.S_rest != nullptr ? post_in(q, q.S_rest + h) : 0.f;
  for (int k = 0; k < q.Kft; ++k) {
    float sk;
    if (q.S_rest != nullptr && k == q.Kft - 1) {
      sk = rest;
    } else {
      sk = post_in(q, q.S + (size_t)h * q.Kft + k);
      rest -= sk;
    }
    float t = q.temb[(size_t)q.ksteps[k].t * q.td + j];
    if (q.temb_bf16) t = bf2f(f2bf(t));
    acc += sk * t;
  }
  q.dW0t[(size_t)h * q.ldw0 + q.AF + j] = acc;
}

// A job of very many slabs (SlabJob::wide: the in-kernel dW0's one slab per workgroup of the fused backward, up to 256 of them):
// a thread per element would add 256 values in eight dependent batches on three CUs (measured 44 us for a 256 x 11 output).  Here
// 64 consecutive elements belong to a block, wave w of it adds slabs w, w + NW, w + 2 NW, ... (NW = waves per block, at most 16
// loads in flight per lane, one or two memory latencies), and wave 0 adds the NW partial sums in a fixed order.
template <bool WT = false>  // WT: results stored write-through (riding: read by other workgroups of this launch)
__device__ __forceinline__ void slab_job_block_wide(const SlabJob& J, int bx = -1, int nbx = 0) {  // bx >= 0: block bx of nbx instead of blockIdx.x / gridDim.x
  __shared__ float wred[16][64];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, NW = blockDim.x >> 6;
  const size_t n = (size_t)J.rows * J.cols;
  const size_t b0 = bx >= 0 ? (size_t)bx : blockIdx.x, bn = bx >= 0 ? (size_t)nbx : gridDim.x;
  for (size_t e0 = b0 * 64; e0 < n; e0 += bn * 64) {
    const size_t i = e0 + lane;
    const bool live = i < n;
    const int r = live ? (int)(i / J.cols) : 0, c = live ? (int)(i % J.cols) : 0;
    const float* src = J.slab + (size_t)r * J.lds + J.c0 + c;
    const size_t stride = (size_t)J.rows * J.lds;
    float p[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int k0 = w; k0 < J.splits; k0 += 8 * NW) {
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = live && k0 + u * NW < J.splits ? src[(size_t)(k0 + u * NW) * stride] : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) p[u] += t[u];
    }
    wred[w][lane] = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
    __syncthreads();
    if (w == 0 && live) {
      float v = 0.f;
      for (int u = 0; u < NW; ++u) v += wred[u][lane];
      float* dst = J.transpose ? J.out + (size_t)c * J.ldo + r : J.out + (size_t)r * J.ldo + c;
      if constexpr (WT)
        __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else
        *dst = v;
    }
    __syncthreads();
  }
}

// The same reduction for a 256-thread rider block of a memory-saturated launch: 16 consecutive elements x 16 slab lanes, every lane's
// loads (splits / 16 of them, at most 16) in flight at once -- under the GEMM's load a dependent batch costs several microseconds,
// and eight of them in a row (slab_job_block_wide at four waves) made the riders the launch's critical path.  Fixed order: a lane
// adds its slabs k = q, q + 16, ... in order, then the 16 lanes' partial sums are added in lane order.
template <bool WT>
__device__ __forceinline__ void slab_job_rider(const SlabJob& J, int bx, int nbx) {
  __shared__ float rred[16][17];
  const int el = threadIdx.x & 15, q = threadIdx.x >> 4;
  const size_t n = (size_t)J.rows * J.cols, stride = (size_t)J.rows * J.lds;
  for (size_t e0 = (size_t)bx * 16; e0 < n; e0 += (size_t)nbx * 16) {
    const size_t i = e0 + el;
    const bool live = i < n;
    const int r = live ? (int)(i / J.cols) : 0, c = live ? (int)(i % J.cols) : 0;
    const float* src = J.slab + (size_t)r * J.lds + J.c0 + c;
    float t[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) t[u] = live && q + 16 * u < J.splits ? src[(size_t)(q + 16 * u) * stride] : 0.f;
    float v = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) v += t[u];
    for (int k = q + 256; k < J.splits; k += 16) v += live ? src[(size_t)k * stride] : 0.f;  // (more than 256 slabs: never today)
    rred[q][el] = v;
    __syncthreads();
    if (q == 0 && live) {
      float sum = 0.f;
#pragma unroll
      for (int u = 0; u < 16; ++u) sum += rred[u][el];
      float* dst = J.transpose ? J.out + (size_t)c * J.ldo + r : J.out + (size_t)r * J.ldo + c;
      if constexpr (WT)
        __hip_atomic_store(dst, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else
        *dst = sum;
    }
    __syncthreads();
  }
}

// Block tb of the time-embedding part (PostReduce::n_temb blocks of 256 threads): G[k][j] = sum_h W0[h][AF + j] S[h][k], one wave
// per output, and -- in the last block of the range, once the others have arrived -- the time MLP's backward.  sh: dynamic LDS,
// time_backward_lds(Kft, td) bytes.
__device__ __forceinline__ void temb_g_block(const PostReduce& q, int tb, float* sh) {
  const int tid = threadIdx.x;
  post_wait(q);
  const int lane = tid & 63, out = tb * 4 + (tid >> 6);
  if (out < q.Kft * q.td) {
    const int k = out / q.td, j = out % q.td;
    // (S_rest: the last step's row is formed from the sums over ALL rows; time_backward_finish subtracts the other steps' rows)
    const bool rest = q.S_rest != nullptr && k == q.Kft - 1;
    float acc = 0.f;
    int h = lane;
    for (; h + 7 * 64 < q.H; h += 8 * 64) {  // (eight of the other workgroups' values in flight: riding, each is a trip past L2)
      float sv[8], wv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int hh = h + 64 * u;
        sv[u] = post_in(q, rest ? q.S_rest + hh : q.S + (size_t)hh * q.Kft + k);
        wv[u] = q.W0[(size_t)hh * q.ldw0 + q.AF + j];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += wv[u] * sv[u];
    }
    for (; h < q.H; h += 64) acc += q.W0[(size_t)h * q.ldw0 + q.AF + j] * post_in(q, rest ? q.S_rest + h : q.S + (size_t)h * q.Kft + k);
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    // write-through (sc1) store, read back by the last block with sc1 loads (time_backward_finish): the hand-over then needs
    // no fence on either side -- every storing wave drains its store, the block's barrier, ONE relaxed agent-scope add
    // (guide section 6, guideline 16).  With __threadfence() around the counter (buffer_wbl2 + buffer_inv, ~3.5 us each on
    // gfx950) this chain -- G, fence, add | poll, fence, finish -- was 11 of the launch's 16 us, on the update's critical path.
    if (lane == 0) __hip_atomic_store(&q.G[out], acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  DPPO_HANDOVER_DRAIN();  // (common.h: fence-free on gfx942 / gfx950, release / acquire elsewhere)
  __syncthreads();
  // The time MLP's backward belongs to the LAST block of the range: it prepares everything that does not depend on G while
  // the others finish, then waits for their arrivals.  (Workgroups are dispatched in index order, so every block it waits
  // for is already running or done: the wait cannot starve them.)
  if (tb != q.n_temb - 1) {
    if (tid == 0) __hip_atomic_fetch_add(q.counter, 1u, DPPO_HANDOVER_ARRIVE_ORDER, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  time_backward_prepare(q.w1, q.b1, q.w2, q.ksteps, q.Kft, q.td, sh);
  if (tid == 0) {
    while (__hip_atomic_load(q.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)q.n_temb - 1)
      __builtin_amdgcn_s_sleep(2);
    __hip_atomic_store(q.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next call
    DPPO_HANDOVER_ACQUIRE();
  }
  __syncthreads();
  time_backward_finish(q.G, q.Kft, q.td, q.gw1, q.gb1, q.gw2, q.gb2, sh, q.S_rest != nullptr);
}

}  // namespace dppo
