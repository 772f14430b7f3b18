// LayerNorm on the streamed-weight engine's register layout (sampler.hip, fused.hip).
//
// A 16*MR-row tile's H features are spread over the workgroup: wave w owns features [16*TPW*w, 16*TPW*(w+1)), lane
// (r, g) of it holds feat_off(g, tp) + e for batch rows 16m + r.  A per-row reduction over H therefore goes: in-lane
// over (tp, e) -> the four g lanes of a row (shuffle xor 16, 32) -> the 8 waves through a tiny LDS table.
// Reference: nn.LayerNorm(H, eps=1e-6) inside TwoLayerPreActivationResNetLinear (model/common/mlp.py:139-154):
// y = (x - mean) / sqrt(var_biased + eps) * gamma + beta, two-pass variance like torch.
#pragma once
#include "common.h"

namespace dppo {

constexpr float LN_EPS = 1e-6f;
constexpr int LN_WAVES = 8;

// v[k][m] -> sum over all H features of row (16m + r); result replicated in every lane of that row.
// lds: LN_WAVES * NV * 16 * MR floats.  Two barriers; all 512 threads must call.
template <int NV, int MR>
__device__ __forceinline__ void row_reduce(float (&v)[NV][MR], float* lds, int wid, int r, int g) {
#pragma unroll
  for (int k = 0; k < NV; ++k)
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      float x = v[k][m];
      x += __shfl_xor(x, 16);
      x += __shfl_xor(x, 32);
      if (g == 0) lds[((wid * NV + k) * MR + m) * 16 + r] = x;
    }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NV; ++k)
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < LN_WAVES; ++w) s += lds[((w * NV + k) * MR + m) * 16 + r];
      v[k][m] = s;
    }
  __syncthreads();  // the table is free again
}

// out = act-free LayerNorm of x over the H features of each row.  gamma/beta: this network's parameters (global),
// indexed by feature.  mean/rstd returned per row sub-tile (for saving to the backward).
template <class P, int TPW, int MR>
__device__ __forceinline__ void ln_forward(const f32x4 (&x)[TPW][MR], f32x4 (&out)[TPW][MR], const float* gamma,
                                           const float* beta, int H, int wbase, int g, int r, int wid, float* lds,
                                           float (&mean)[MR], float (&rstd)[MR]) {
  float s[1][MR];
#pragma unroll
  for (int m = 0; m < MR; ++m) {
    float t = 0.f;
#pragma unroll
    for (int tp = 0; tp < TPW; ++tp) t += (x[tp][m][0] + x[tp][m][1]) + (x[tp][m][2] + x[tp][m][3]);
    s[0][m] = t;
  }
  row_reduce<1, MR>(s, lds, wid, r, g);
  const float inv = 1.f / (float)H;
#pragma unroll
  for (int m = 0; m < MR; ++m) mean[m] = s[0][m] * inv;
#pragma unroll
  for (int m = 0; m < MR; ++m) {
    float t = 0.f;
#pragma unroll
    for (int tp = 0; tp < TPW; ++tp)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = x[tp][m][e] - mean[m];
        t += d * d;
      }
    s[0][m] = t;
  }
  row_reduce<1, MR>(s, lds, wid, r, g);
#pragma unroll
  for (int m = 0; m < MR; ++m) rstd[m] = 1.f / sqrtf(s[0][m] * inv + LN_EPS);
#pragma unroll
  for (int tp = 0; tp < TPW; ++tp) {
    const int f = wbase + feat_off<P>(g, tp);
    f32x4 ga, be;  // scalar loads: flat-parameter offsets are not always 16-byte aligned
#pragma unroll
    for (int e = 0; e < 4; ++e) ga[e] = gamma[f + e], be[e] = beta[f + e];
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
      for (int e = 0; e < 4; ++e) out[tp][m][e] = (x[tp][m][e] - mean[m]) * rstd[m] * ga[e] + be[e];
  }
}

// element (tile tp, row sub-tile m, e) of a packed fetch (fused.hip: CH 16-byte chunks per row sub-tile)
template <class P, int MR, int CH>
__device__ __forceinline__ float elem_at(const u32x4 (&d)[MR][CH], int tp, int m, int e) {
  if constexpr (P::ESIZE == 4) {
    return __uint_as_float(d[m][tp][e]);
  } else {
    const uint32_t w = d[m][tp >> 1][(tp & 1) * 2 + (e >> 1)];
    return bf2f((e & 1) ? (w >> 16) : (w & 0xffff));
  }
}

// Backward of y = act(LN(x)) on the register layout.  In: gacc = d loss / d y.  Out: gacc = d loss / d x,
// dgamma[tp][e] / dbeta[tp][e] = this lane's sums over its MR rows of (du * xhat) / du, with du = gacc * act'(u),
// u = xhat * gamma + beta (the caller reduces them over the 16 row lanes).  x arrives packed (pre-LN tensor saved by the
// forward), mean / rstd are the saved row statistics.  dx = rstd * (dxhat - mean_f(dxhat) - xhat * mean_f(dxhat * xhat)).
template <class P, int TPW, int MR, int CH>
__device__ __forceinline__ void ln_backward(f32x4 (&gacc)[TPW][MR], const u32x4 (&xraw)[MR][CH], const float* gamma,
                                            const float* beta, const float (&mean)[MR], const float (&rstd)[MR], int actk,
                                            int H, int wbase, int g, int r, int wid, float* lds, f32x4 (&dgamma)[TPW],
                                            f32x4 (&dbeta)[TPW]) {
  float s[2][MR];
#pragma unroll
  for (int m = 0; m < MR; ++m) s[0][m] = s[1][m] = 0.f;
  with_act(actk, [&](auto tag) {
    constexpr int ACT = decltype(tag)::value;
#pragma unroll
    for (int tp = 0; tp < TPW; ++tp) {
      const int f = wbase + feat_off<P>(g, tp);
      f32x4 ga, be, dg = (f32x4){0.f, 0.f, 0.f, 0.f}, db = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 4; ++e) ga[e] = gamma[f + e], be[e] = beta[f + e];
#pragma unroll
      for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float xh = (elem_at<P>(xraw, tp, m, e) - mean[m]) * rstd[m];
          const float du = gacc[tp][m][e] * act_grad_c<ACT>(xh * ga[e] + be[e]);
          dg[e] += du * xh;
          db[e] += du;
          const float dxh = du * ga[e];
          gacc[tp][m][e] = dxh;
          s[0][m] += dxh;
          s[1][m] += dxh * xh;
        }
      dgamma[tp] = dg;
      dbeta[tp] = db;
    }
  });
  row_reduce<2, MR>(s, lds, wid, r, g);
  const float inv = 1.f / (float)H;
#pragma unroll
  for (int tp = 0; tp < TPW; ++tp)
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xh = (elem_at<P>(xraw, tp, m, e) - mean[m]) * rstd[m];
        gacc[tp][m][e] = rstd[m] * (gacc[tp][m][e] - s[0][m] * inv - xh * s[1][m] * inv);
      }
}

}  // namespace dppo
