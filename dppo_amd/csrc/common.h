// Shared device/host definitions for the DPPO gfx950 kernels.
//
// Precision policy: every dense contraction runs on MFMA with fp32 accumulation.  Two operand
// precisions are built from one source via the Prec traits below:
//   F32  : v_mfma_f32_16x16x4_f32   (exact fp32 products; parity mode, tolerance 1e-5 class)
//   BF16 : v_mfma_f32_16x16x32_bf16 (operands rounded to bf16; throughput mode)
// Both consume operands as "64-byte k-steps": each of the 16 operand rows of a tile contributes 64
// contiguous bytes of K (16 fp32 or 32 bf16); lane l = (r = l & 15, g = l >> 4) holds bytes
// [16g, 16g+16) of row r.  For bf16 that is exactly the hardware map A[r][8g..8g+7]; for fp32 the
// four MFMAs of a k-step use element s of both operands, i.e. k = 4g + s -- a permutation of k that
// is the same for A and B and therefore leaves the product unchanged.
// D layout (both): lane (r, g) holds D[4g + e][r], e = 0..3.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

namespace dppo {

// hipFuncSetAttribute (the > 64 KB dynamic-LDS cap) is PER DEVICE: one latch bit per device ordinal instead of a
// process-wide bool, so a second device of the same process gets its cap raised too.  Atomic; two threads racing on the
// same device both set the attribute, which is idempotent.
struct DevLatch {
  std::atomic<uint64_t> bits{0};
  static int dev() {
    int d = 0;
    (void)hipGetDevice(&d);
    return d & 63;
  }
  bool need() const { return !((bits.load(std::memory_order_acquire) >> dev()) & 1); }
  void done() { bits.fetch_or(1ull << dev(), std::memory_order_release); }
};

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short i16x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;  // one 16-byte MFMA fragment per lane (native vector: stays in VGPRs)
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

enum { ACT_RELU = 0, ACT_MISH = 1, ACT_NONE = 2 };

// Fence-free last-block hand-over (guide guideline 16).  The producer's partial results are written with relaxed AGENT-scope
// atomic stores, the arrival is ONE relaxed atomic add behind `s_waitcnt vmcnt(0)`, and the consumer reads them with relaxed
// agent-scope atomic loads.  Under the HIP / LLVM memory model that has no release / acquire edge; it is correct BY CODE
// GENERATION on gfx942 / gfx950 only: agent-scope atomic stores are write-through (sc1) and complete -- vmcnt counts stores
// -- when L2 has them, agent-scope atomic loads bypass the non-coherent L1 (sc1), and the add is performed at L2 behind the
// drained stores.  Any other target gets the formal release / acquire pair instead (a __threadfence() each side measured
// ~3.5 us on gfx950, which is why the shortcut exists).  DPPO_HANDOVER_ARRIVE() / DPPO_HANDOVER_ACQUIRE() wrap the two forms.
#if defined(__gfx942__) || defined(__gfx950__)
#define DPPO_HANDOVER_FENCE_FREE 1
#define DPPO_HANDOVER_ARRIVE_ORDER __ATOMIC_RELAXED
#define DPPO_HANDOVER_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define DPPO_HANDOVER_ACQUIRE() \
  do {                          \
  } while (0)
#else
#define DPPO_HANDOVER_FENCE_FREE 0
#define DPPO_HANDOVER_ARRIVE_ORDER __ATOMIC_RELEASE
#define DPPO_HANDOVER_DRAIN() \
  do {                        \
  } while (0)
#define DPPO_HANDOVER_ACQUIRE() __atomic_thread_fence(__ATOMIC_ACQUIRE)
#endif

// Zero `bytes` (a multiple of 4) bytes at the 4-byte aligned `p` on stream `s` with a KERNEL (ppo.hip).  The library never uses
// hipMemsetAsync: captured into a hipGraph on ROCm 7.2 a memset node left 16 foreign bytes (a size and an address) at the head of
// its destination from the second replay on (found through the split sampler's time-out word, gpurun_out/s4d/t.log), and every
// entry point of this library may be captured (dppo_amd.util.graphed.GraphedUpdate takes any model).
void launch_zero_bytes(void* p, size_t bytes, hipStream_t s);

__device__ __forceinline__ uint16_t f2bf(float x) {
  __bf16 b = (__bf16)x;  // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN stays NaN
  return __builtin_bit_cast(uint16_t, b);
}
__device__ __forceinline__ float bf2f(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }

struct F32 {
  typedef float elem_t;
  static constexpr int KB = 16;  // elements per 64-byte k-step
  static constexpr int ESIZE = 4;
  static __device__ __forceinline__ f32x4 mma(u32x4 a, u32x4 b, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
    return c;
  }
  static __device__ __forceinline__ elem_t from_f32(float x) { return x; }
  static __device__ __forceinline__ float to_f32(elem_t x) { return x; }
};

struct BF16 {
  typedef uint16_t elem_t;
  static constexpr int KB = 32;
  static constexpr int ESIZE = 2;
  static __device__ __forceinline__ f32x4 mma(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c,
                                                   0, 0, 0);
  }
  static __device__ __forceinline__ elem_t from_f32(float x) { return f2bf(x); }
  static __device__ __forceinline__ float to_f32(elem_t x) { return bf2f(x); }
};

// nn.Mish = x * tanh(softplus(x)).  With e = exp(x): tanh(log(1 + e)) = ((1+e)^2 - 1) / ((1+e)^2 + 1)
// = n / (n + 2), n = e (e + 2) -- exact algebra, no cancellation, one exp + one division instead of
// expf + log1pf + tanhf (which made the Mish critic cost more than the 4x larger ReLU actor).  x is clamped at 20
// for the exponential only (n / (n + 2) == 1 in fp32 from x ~ 9, matching torch's softplus threshold behaviour).
// With w = n + 2: tanh(sp) = n / w, and since n + 1 = (1 + e)^2 the derivative
// d/dx [x tanh(sp(x))] = tanh(sp) + x (1 - tanh(sp)^2) sigmoid(x) collapses to n/w + 4 x e (1 + e) / w^2: value and
// derivative share ONE exp and ONE reciprocal.  The reciprocal is v_rcp_f32 (1 ulp): an IEEE divide is ~10
// instructions, and the activation of a 64-row tile costs as much as one of its 512 x 512 layers on the matrix cores
// unless it is kept to exp + rcp + a few multiplies.
__device__ __forceinline__ void mish_both(float x, float& val, float& grad) {
  const float e = __expf(fminf(x, 20.f));
  const float n = e * (e + 2.f);
  const float R = __builtin_amdgcn_rcpf(n + 2.f);
  const float th = n * R;
  val = x * th;
  grad = th + 4.f * x * (e * (1.f + e)) * (R * R);
}
// Two values at a time on the packed-fp32 VALU (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32: two lanes' worth of work per
// issue slot): the fused forward's Mish emits are VALU-bound (a 128-row critic tile spends 20k of its 34k cycles in them),
// and 12 of the ~14 instructions per value are plain multiplies and adds.  Same operation order per value as mish_both().
typedef float float2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void mish_both2(float2v x, float2v& val, float2v& grad) {
  // no contraction: the training forward (value + derivative) and the inference forward (value only: the derivative's half
  // is dead code) must round `val` identically -- the update's recomputed log-probs equal the precomputed ones bit for bit
  // (ratio == 1 with unchanged weights) only if the compiler cannot fuse n + 2 into an fma in one instantiation and not in
  // the other
#pragma clang fp contract(off)
  const float2v e = {__expf(fminf(x.x, 20.f)), __expf(fminf(x.y, 20.f))};
  const float2v n = e * (e + 2.f);
  const float2v w = n + 2.f;
  const float2v R = {__builtin_amdgcn_rcpf(w.x), __builtin_amdgcn_rcpf(w.y)};
  const float2v th = n * R;
  val = x * th;
  grad = th + 4.f * x * (e * (1.f + e)) * (R * R);
}
__device__ __forceinline__ float mish_f(float x) {
  const float e = __expf(fminf(x, 20.f));
  const float n = e * (e + 2.f);
  return x * (n * __builtin_amdgcn_rcpf(n + 2.f));
}
__device__ __forceinline__ float mish_grad_f(float x) {
  float v, g;
  mish_both(x, v, g);
  return g;
}
__device__ __forceinline__ float act_f(int act, float x) {
  return act == ACT_RELU ? fmaxf(x, 0.f) : (act == ACT_MISH ? mish_f(x) : x);
}
__device__ __forceinline__ float act_grad_f(int act, float x) {
  return act == ACT_RELU ? (x > 0.f ? 1.f : 0.f) : (act == ACT_MISH ? mish_grad_f(x) : 1.f);
}
// The same with the kind a compile-time constant, and with_act() to get there from the runtime kind ONCE per tile
// instead of once per element (the per-element form made a ReLU emit cost as much as a Mish one: measured 8k of a
// 52k-cycle tile, twice).
template <int ACT>
__device__ __forceinline__ float act_c(float x) {
  if constexpr (ACT == ACT_RELU) return fmaxf(x, 0.f);
  if constexpr (ACT == ACT_MISH) return mish_f(x);
  return x;
}
template <int ACT>
__device__ __forceinline__ float act_grad_c(float x) {
  if constexpr (ACT == ACT_RELU) return x > 0.f ? 1.f : 0.f;
  if constexpr (ACT == ACT_MISH) return mish_grad_f(x);
  return 1.f;
}
template <int ACT>
struct ActTag {
  static constexpr int value = ACT;
};
template <class F>
__device__ __forceinline__ void with_act(int actk, F&& body) {
  if (actk == ACT_RELU)
    body(ActTag<ACT_RELU>{});
  else if (actk == ACT_MISH)
    body(ActTag<ACT_MISH>{});
  else
    body(ActTag<ACT_NONE>{});
}

// N(0,1) draw number `idx` of the stream keyed by (k0, k1): Philox4x32-10 on counter (idx, 0), Box-Muller on two of its
// four words.  The counter is the element's index in the (n_steps+1, B, AF) noise tensor the host would have drawn.
__device__ __forceinline__ float philox_normal(uint64_t idx, uint32_t k0, uint32_t k1) {
  uint32_t c0 = (uint32_t)idx, c1 = (uint32_t)(idx >> 32), c2 = 0, c3 = 0;
#pragma unroll
  for (int rnd = 0; rnd < 10; ++rnd) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    c0 = hi1 ^ c1 ^ k0, c1 = lo1, c2 = hi0 ^ c3 ^ k1, c3 = lo0;
    k0 += 0x9E3779B9u, k1 += 0xBB67AE85u;
  }
  const float u1 = ((float)c0 + 1.f) * 2.3283064365386963e-10f;  // (0, 1]
  const float u2 = (float)c1 * 2.3283064365386963e-10f;          // [0, 1)
  return sqrtf(-2.f * __logf(u1)) * __cosf(6.283185307179586f * u2);
}

// Loads with the address space spelled out.  Where a value comes from LDS on one path and from global memory on the other,
// the compiler merges the two loads into ONE flat_load through a generic pointer -- and a FLAT access ticks both memory
// counters, so it is followed by s_waitcnt vmcnt(0) lgkmcnt(0): every weight fragment the ring has in flight is waited for.
template <class T>
__device__ __forceinline__ T lds_load(const T* p) {
  return *(const __attribute__((address_space(3))) T*)p;
}
template <class T>
__device__ __forceinline__ T glb_load(const T* p) {
  return *(const __attribute__((address_space(1))) T*)p;
}

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// Sum over the 16 lanes of a DPP row (lanes with the same lane >> 4), every lane gets the total: four VALU adds with a DPP
// operand (quad_perm xor 1, xor 2, row_half_mirror, row_mirror) instead of four ds_bpermute round trips through the LDS
// crossbar (__shfl_xor by 1, 2, 4, 8).  Same pairing of the partial sums as the xor butterfly: bit-identical results.
__device__ __forceinline__ float row16_sum(float x) {
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true));
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xF, 0xF, true));
  return x;
}

// Which features of its wave's 16*TPW-feature slice a lane owns in the streamed-weight kernels (sampler, fused):
// MFMA output row i = 4g + e of tile tp is feature feat_off(g, tp) + e.  Tiles are grouped so that one 16-byte chunk
// (EPC = 16 / ESIZE elements) holds consecutive features of ONE lane and the four lanes g = 0..3 own four consecutive
// chunks: a store / load instruction of chunk group c then covers 64 contiguous bytes per batch row.
template <class P>
__host__ __device__ __forceinline__ constexpr int feat_off(int g, int tp) {
  constexpr int EPC = 16 / P::ESIZE, TPC = EPC / 4;
  return (tp / TPC) * (4 * EPC) + EPC * g + 4 * (tp % TPC);
}

}  // namespace dppo
