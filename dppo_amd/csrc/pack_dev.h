// Device bodies of the weight-image packers, shared by their stand-alone kernels (fused.hip, sampler.hip) and by the
// one-launch-per-network packer (pack_net_kernel, ppo.hip).  One 64-lane block writes one 1-KiB fragment.
#pragma once
#include "fused.h"

namespace dppo {

// streamed-weight fragment bx = (wave, k-step, tile) of layer L (see PackLayer)
template <class P>
__device__ __forceinline__ void pack_stream_block(const PackLayer& L, const int TPW, const int bx) {
  if (bx >= SAMPLER_WAVES * L.KS * TPW) return;
  const int lane = threadIdx.x & 63;
  const int tp = bx % TPW;
  const int ks = (bx / TPW) % L.KS;
  const int w = bx / (TPW * L.KS);
  const int r = lane & 15, g = lane >> 4;
  const long feat = w * 16 * TPW + feat_off<P>(r >> 2, tp) + (r & 3);
  constexpr int EPL = 16 / P::ESIZE;
  const int k0 = ks * P::KB + EPL * g;
  uint32_t out[4];
  if constexpr (P::ESIZE == 4) {
#pragma unroll
    for (int j = 0; j < 4; ++j) out[j] = __float_as_uint(k0 + j < L.in_valid ? L.W[feat * L.rs + (long)(k0 + j) * L.cs] : 0.f);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k0 + 2 * j;
      const float lo = k < L.in_valid ? L.W[feat * L.rs + (long)k * L.cs] : 0.f;
      const float hi = k + 1 < L.in_valid ? L.W[feat * L.rs + (long)(k + 1) * L.cs] : 0.f;
      out[j] = (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
    }
  }
  L.stream[(((size_t)w * L.total_pos + L.pos0 + ks) * TPW + tp) * 64 + lane] = (u32x4){out[0], out[1], out[2], out[3]};
}

// out-layer stream: [wave w][c][to][lane]; wave w owns k-steps w*CNT + c; rows >= out_dim are zero
template <class P>
__device__ __forceinline__ void pack_out_block(const float* W, int out_dim, int H, int OT, int CNT, u32x4* stream,
                                               const int bx) {
  const int lane = threadIdx.x & 63;
  const int to = bx % OT;
  const int c = (bx / OT) % CNT;
  const int w = bx / (OT * CNT);
  const int r = lane & 15, g = lane >> 4;
  const int o = to * 16 + r;
  const int ks = w * CNT + c;
  constexpr int EPL = 16 / P::ESIZE;
  const int k0 = ks * P::KB + EPL * g;
  const bool ok = o < out_dim && k0 < H;
  uint32_t out[4];
  if constexpr (P::ESIZE == 4) {
#pragma unroll
    for (int j = 0; j < 4; ++j) out[j] = __float_as_uint(ok ? W[(size_t)o * H + k0 + j] : 0.f);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float lo = ok ? W[(size_t)o * H + k0 + 2 * j] : 0.f;
      const float hi = ok ? W[(size_t)o * H + k0 + 2 * j + 1] : 0.f;
      out[j] = (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
    }
  }
  stream[(((size_t)w * CNT + c) * OT + to) * 64 + lane] = (u32x4){out[0], out[1], out[2], out[3]};
}

}  // namespace dppo
