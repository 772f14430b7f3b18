// 256-thread forms of the two small reductions that ride in the weight-gradient GEMM launch (gemm.h, GemmTNExtra); the
// 1024-thread forms of the separate reduction launch are tail_reduce_kernel's (ppo.hip).  Same arithmetic per output, a
// different (fixed) grouping of the partial sums.
#pragma once
#include "common.h"
#include "dppo_hip.h"

namespace dppo {

// out[c] = sum over tiles of colsum[tile][c] for 16 columns starting at 16 x: 16 column-lanes x 16 tile-lanes
__device__ __forceinline__ void slot_reduce_block256(const float* src, int tiles, int width, int n_out, float* out, int x,
                                                     bool wt = false) {  // wt: result stored write-through (read by riders of the same launch)
  __shared__ float red[16][17];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = x * 16 + cl;
  float p[4] = {0.f, 0.f, 0.f, 0.f};
  if (c < width && n_out > 0) {
    int r = rl;
    for (; r + 48 < tiles; r += 64) {
#pragma unroll
      for (int u = 0; u < 4; ++u) p[u] += src[(size_t)(r + 16 * u) * width + c];
    }
    for (; r < tiles; r += 16) p[0] += src[(size_t)r * width + c];
  }
  red[rl][cl] = (p[0] + p[1]) + (p[2] + p[3]);
  __syncthreads();
  if (rl == 0 && c < n_out) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += red[i][cl];
    if (wt)
      __hip_atomic_store(out + c, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
      out[c] = s;
  }
}

// the five loss statistics summed over the loss kernel's blocks in a fixed order (256 threads; see loss_finalize_block)
__device__ __forceinline__ void loss_finalize_block256(const double* partial, int blocks, const double* moments, double* stats,
                                                       int part, double n_count) {
  __shared__ double shd[4][5];
  const double Nn = n_count > 0 ? n_count : moments[2];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  double v[5] = {0, 0, 0, 0, 0};
  for (int b = tid; b < blocks; b += 256) {
#pragma unroll
    for (int k = 0; k < 5; ++k) v[k] += partial[(size_t)b * 8 + k];
  }
#pragma unroll
  for (int k = 0; k < 5; ++k)
    for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_down(v[k], o);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 5; ++k) shd[w][k] = v[k];
  }
  __syncthreads();
  if (tid < 5 && ((tid == DPPO_STAT_V_LOSS ? 2 : 1) & part)) {  // the other half's launch owns the other entries
    const double t = (shd[0][tid] + shd[1][tid]) + (shd[2][tid] + shd[3][tid]);
    stats[tid] = t / Nn;
  }
  if (tid == 64 && (part & 1)) {
    const double mean = moments[0] / Nn;
    const double varu = Nn > 1 ? (moments[1] - Nn * mean * mean) / (Nn - 1.0) : 0.0;
    stats[DPPO_STAT_ADV_MEAN] = mean;
    stats[DPPO_STAT_ADV_STD] = sqrt(varu > 0 ? varu : 0);
  }
}

}  // namespace dppo
