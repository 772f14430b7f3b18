// See gemm.h.  gfx950 only.
#include "gemm.h"
#include "tail_blocks.h"
#include "post_blocks.h"

namespace dppo {

// ------------------------------------------------------------------------------------------------
// gemm_nt
// ------------------------------------------------------------------------------------------------
// Block = WN x WM waves; a wave owns TN x TM MFMA tiles (16 features x 16 batch rows each).
// One LDS stage holds 128 bytes of K (two 64-byte k-steps) for BN weight rows and BM batch rows;
// 16-byte chunk c of tile row q lives at chunk position c ^ (q & 7): conflict-free for both the
// 8-lanes-per-row ds_write_b128 of the loader and the fragment ds_read_b128 (lane (r,g) reads
// chunk 4s+g of row r: the 16 lanes of each b128 lane group land on 16 distinct 16-byte slots).
// LDS weight row q is NOT feature feat0+q: rows are permuted so that MFMA output row i = 4g+e of
// tile tn is feature 4*TN*g + 4*tn + e of the wave's slice, which gives each lane 4*TN consecutive
// features of one batch row in its accumulators.
// TAG only names the instantiation: TAG = 1 is used for the square hidden layers (N == Kp: forward l1 / l2 and both
// backward-data GEMMs), so that profiles list the dominant launches under their own symbol.
// DMA = 1 stages both operands with global_load_lds (LDS-DMA, 16 B per lane, no VGPR round trip).  The LDS write
// of a wave instruction is wave-uniform base + lane * 16, i.e. linear, so the XOR swizzle moves to the per-lane
// SOURCE address: LDS chunk position p of tile row q receives source chunk p ^ (q & 7) (the involution the
// fragment reads apply).  Needs every weight row of the tile to exist (N % BN == 0: no zero fill in a DMA).
template <class P, int WN, int WM, int TN, int TM, int TAG, int DMA>
__global__ __launch_bounds__(WN* WM * 64) void gemm_nt_kernel(const GemmNT a) {
  typedef typename P::elem_t E;
  constexpr int BN = WN * TN * 16, BM = WM * TM * 16, T = WN * WM * 64, ES = P::ESIZE;
  constexpr int NW = (BN * 8 + T - 1) / T, NX = (BM * 8 + T - 1) / T;
  constexpr int STAGE = (BN + BM) * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int wn = wid % WN, wm = wid / WN;
  const int row0 = blockIdx.x * BM, feat0 = blockIdx.y * BN;
  const char* Xb = (const char*)a.X;
  const char* Wb = (const char*)a.W;
  const int nk = a.Kp * ES / 128;

  u32x4 wreg[NW], xreg[NX];
  auto gload = [&](int kt) {
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int q = (tid + i * T) % (BN * 8);  // BN*8 < T (skinny tiles): surplus threads reload a valid chunk
      const int rho = q >> 3, c = q & 7;
      const int w_ = rho / (16 * TN), tn = (rho >> 4) % TN, ii = rho & 15;
      const int feat = feat0 + w_ * 16 * TN + 4 * TN * (ii >> 2) + 4 * tn + (ii & 3);
      const bool ok = feat < a.N;
      const u32x4 v = *(const u32x4*)(Wb + (size_t)(ok ? feat : 0) * a.ldw * ES + (size_t)kt * 128 + c * 16);
      wreg[i] = ok ? v : (u32x4){0, 0, 0, 0};
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int q = tid + i * T;
      const int rr = q >> 3, c = q & 7;
      int row = row0 + rr;
      row = row < a.M ? row : a.M - 1;
      xreg[i] = *(const u32x4*)(Xb + (size_t)row * a.ldx * ES + (size_t)kt * 128 + c * 16);
    }
  };
  auto sstore = [&](int st) {
    char* Ws = smem + st * STAGE;
    char* Xs = Ws + BN * 128;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int q = (tid + i * T) % (BN * 8);  // duplicates write identical bytes to the same address
      const int rho = q >> 3, c = q & 7;
      *(u32x4*)(Ws + rho * 128 + ((c ^ (rho & 7)) << 4)) = wreg[i];
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int q = tid + i * T;
      const int rr = q >> 3, c = q & 7;
      *(u32x4*)(Xs + rr * 128 + ((c ^ (rr & 7)) << 4)) = xreg[i];
    }
  };

  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* glb_ptr;
  auto dma = [&](int kt, int st) {
    char* Ws = smem + st * STAGE;
    char* Xs = Ws + BN * 128;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int q = tid + i * T;  // LDS chunk slot: row q>>3, position q&7 (requires BN*8 % T == 0)
      const int rho = q >> 3, c = (q & 7) ^ (rho & 7);
      const int w_ = rho / (16 * TN), tn = (rho >> 4) % TN, ii = rho & 15;
      const int feat = feat0 + w_ * 16 * TN + 4 * TN * (ii >> 2) + 4 * tn + (ii & 3);
      __builtin_amdgcn_global_load_lds((glb_ptr)(Wb + (size_t)feat * a.ldw * ES + (size_t)kt * 128 + c * 16),
                                       (lds_ptr)(Ws + (size_t)(i * T + (tid & ~63)) * 16), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int q = tid + i * T;
      const int rr = q >> 3, c = (q & 7) ^ (rr & 7);
      int row = row0 + rr;
      row = row < a.M ? row : a.M - 1;
      __builtin_amdgcn_global_load_lds((glb_ptr)(Xb + (size_t)row * a.ldx * ES + (size_t)kt * 128 + c * 16),
                                       (lds_ptr)(Xs + (size_t)(i * T + (tid & ~63)) * 16), 16, 0, 0);
    }
  };

  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if constexpr (DMA) {
    dma(0, 0);
  } else {
    gload(0);
    sstore(0);
  }
  __syncthreads();  // with an LDS-DMA outstanding the barrier's fence waits vmcnt(0): the stage has landed
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if constexpr (DMA) {
      if (kt + 1 < nk) dma(kt + 1, cur ^ 1);  // the other stage was last read before the previous barrier
    } else {
      if (kt + 1 < nk) gload(kt + 1);
    }
    const char* Ws = smem + cur * STAGE;
    const char* Xs = Ws + BN * 128;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      u32x4 af[TN], bf[TM];
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const int rho = wn * TN * 16 + tn * 16 + r;
        af[tn] = *(const u32x4*)(Ws + rho * 128 + (((s * 4 + g) ^ (rho & 7)) << 4));
      }
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
        const int rr = wm * TM * 16 + tm * 16 + r;
        bf[tm] = *(const u32x4*)(Xs + rr * 128 + (((s * 4 + g) ^ (rr & 7)) << 4));
      }
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) acc[tn][tm] = P::mma(af[tn], bf[tm], acc[tn][tm]);
    }
    if constexpr (!DMA) {
      if (kt + 1 < nk) sstore(cur ^ 1);
    }
    __syncthreads();
  }

  // ---- epilogue: lane (r,g) owns batch row (.. + r) and features f0 .. f0 + 4*TN - 1
  const int f0 = feat0 + wn * 16 * TN + 4 * TN * g;
  const int nst = (a.N + 15) & ~15;  // stored width: buffers are padded to >= round_up(N,16) columns
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int row = row0 + wm * TM * 16 + tm * 16 + r;
    if (row >= a.M) continue;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int f = f0 + 4 * tn;
      if (f >= nst) continue;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = acc[tn][tm][e];
        if (a.bias != nullptr && f + e < a.N) v[e] += a.bias[f + e];
      }
      if (a.dsrc_kind == 1) {
        const float4 d = *(const float4*)((const float*)a.dsrc + (size_t)row * a.dsrc_ld + f);
        v[0] *= act_grad_f(a.dact, d.x);
        v[1] *= act_grad_f(a.dact, d.y);
        v[2] *= act_grad_f(a.dact, d.z);
        v[3] *= act_grad_f(a.dact, d.w);
      } else if (a.dsrc_kind == 2) {
        const E* d = (const E*)a.dsrc + (size_t)row * a.dsrc_ld + f;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= act_grad_f(a.dact, P::to_f32(d[e]));
      }
      if (a.res != nullptr) {
        const float4 d = *(const float4*)(a.res + (size_t)row * a.ldres + f);
        v[0] += d.x;
        v[1] += d.y;
        v[2] += d.z;
        v[3] += d.w;
      }
      if (a.add != nullptr) {
        const E* d = (const E*)a.add + (size_t)row * a.ldadd + f;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += P::to_f32(d[e]);
      }
      if (a.out_f32 != nullptr)
        *(float4*)(a.out_f32 + (size_t)row * a.ldo32 + f) = make_float4(v[0], v[1], v[2], v[3]);
      if (a.out_pre != nullptr) {
        E* o = (E*)a.out_pre + (size_t)row * a.ldo + f;
        if constexpr (ES == 4) {
          *(float4*)o = make_float4(v[0], v[1], v[2], v[3]);
        } else {
          u32x2 pk;
          pk.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
          pk.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
          *(u32x2*)o = pk;
        }
      }
      if (a.out_act != nullptr) {
        E* o = (E*)a.out_act + (size_t)row * a.ldo + f;
        float w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = act_f(a.act, v[e]);
        if constexpr (ES == 4) {
          *(float4*)o = make_float4(w[0], w[1], w[2], w[3]);
        } else {
          u32x2 pk;
          pk.x = (uint32_t)f2bf(w[0]) | ((uint32_t)f2bf(w[1]) << 16);
          pk.y = (uint32_t)f2bf(w[2]) | ((uint32_t)f2bf(w[3]) << 16);
          *(u32x2*)o = pk;
        }
      }
    }
  }
}

struct Probe {
  bool armed = false;
  int id = 0, cap = 0, used = 0;
  hipEvent_t* ev = nullptr;  // 2 per launch
  double flops = 0, bytes = 0;
};
static Probe g_probe;
int probe_arm(int kernel_id, int max_launches) {
  if (g_probe.armed || max_launches < 1) return -1;
  g_probe.ev = new hipEvent_t[2 * max_launches];
  for (int i = 0; i < 2 * max_launches; ++i)
    if (hipEventCreate(&g_probe.ev[i]) != hipSuccess) return -1;
  g_probe.id = kernel_id, g_probe.cap = max_launches, g_probe.used = 0, g_probe.flops = 0, g_probe.bytes = 0, g_probe.armed = true;
  return 0;
}
int probe_collect(double* total_ms, int* launches, double* flops, double* bytes) {
  if (!g_probe.armed) return -1;
  double tot = 0;
  for (int i = 0; i < g_probe.used; ++i) {
    float ms = 0;
    if (hipEventSynchronize(g_probe.ev[2 * i + 1]) != hipSuccess) return -1;
    if (hipEventElapsedTime(&ms, g_probe.ev[2 * i], g_probe.ev[2 * i + 1]) != hipSuccess) return -1;
    tot += ms;
  }
  *total_ms = tot, *launches = g_probe.used, *flops = g_probe.flops;
  if (bytes) *bytes = g_probe.bytes;
  for (int i = 0; i < 2 * g_probe.cap; ++i) (void)hipEventDestroy(g_probe.ev[i]);
  delete[] g_probe.ev;
  g_probe = Probe();
  return 0;
}
bool probe_begin(int kernel_id, hipStream_t s) {
  if (!g_probe.armed || g_probe.id != kernel_id || g_probe.used >= g_probe.cap) return false;
  (void)hipEventRecord(g_probe.ev[2 * g_probe.used], s);
  return true;
}
void probe_end(hipStream_t s, double flops, double bytes) {
  (void)hipEventRecord(g_probe.ev[2 * g_probe.used + 1], s);
  g_probe.flops += flops, g_probe.bytes += bytes;
  ++g_probe.used;
}

static int g_nt_variant = 1;  // 0 = register staging everywhere, 1 = LDS-DMA staging where legal (bench A/B knob)
void set_gemm_nt_variant(int v) { g_nt_variant = v; }
static int g_nt_small = 1;  // tuning knob 21: smaller tiles for small problems
void set_gemm_nt_small(int v) { g_nt_small = v; }

template <class P, int WN, int WM, int TN, int TM, int TAG, int DMA>
static void launch_nt_cfg(const GemmNT& a, hipStream_t s) {
  constexpr int BN = WN * TN * 16, BM = WM * TM * 16;
  dim3 grid((a.M + BM - 1) / BM, (a.N + BN - 1) / BN);
  const size_t lds = 2 * (BN + BM) * 128;
  if constexpr (2 * (BN + BM) * 128 > 65536) {  // the 16 x 256 tile needs 68 KiB: raise the dynamic-LDS cap once
    static DevLatch attr_set;
    if (attr_set.need()) {
      (void)hipFuncSetAttribute((const void*)gemm_nt_kernel<P, WN, WM, TN, TM, TAG, DMA>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr_set.done();
    }
  }
  const bool probe = TAG == 1 && probe_begin(PROBE_GEMM_NT_HIDDEN, s);
  hipLaunchKernelGGL((gemm_nt_kernel<P, WN, WM, TN, TM, TAG, DMA>), grid, dim3(WN * WM * 64), lds, s, a);
  if (probe) probe_end(s, 2.0 * a.M * a.N * a.Kp);
}

template <class P>
void launch_gemm_nt(const GemmNT& a, hipStream_t s) {
  if (a.M <= 0) return;
  const bool dma = g_nt_variant == 1 && a.N % 128 == 0;
  // Small problems (the conv denoiser's per-block GEMMs: a few thousand rows, N = 64..256): 128 x 128 tiles leave most of
  // the 256 CUs idle and the launch is one workgroup's serial k-loop long.  Shrink the tile until ~200 workgroups exist
  // (each output's accumulation order over k is unchanged: bit-identical results).
  const int64_t wg128 = (int64_t)((a.M + 127) / 128) * ((a.N + 127) / 128);
  if (g_nt_small && a.N > 16 && wg128 < 192) {
    const bool dma64 = g_nt_variant == 1 && a.N % 64 == 0;
    const int64_t wg64 = (int64_t)((a.M + 63) / 64) * ((a.N + 63) / 64);
    if (wg64 >= 192) {
      if (dma64) launch_nt_cfg<P, 2, 2, 2, 2, 0, 1>(a, s);  // 64 x 64
      else launch_nt_cfg<P, 2, 2, 2, 2, 0, 0>(a, s);
    } else {
      if (dma64) launch_nt_cfg<P, 2, 2, 2, 1, 0, 1>(a, s);  // 64 features x 32 rows
      else launch_nt_cfg<P, 2, 2, 2, 1, 0, 0>(a, s);
    }
    return;
  }
  if (a.N <= 16)
    launch_nt_cfg<P, 1, 4, 1, 4, 0, 0>(a, s);  // 16 features x 256 rows
  else if (a.N <= 64)
    launch_nt_cfg<P, 1, 4, 4, 2, 0, 0>(a, s);  // 64 x 128
  else if (a.N == a.Kp && dma)
    launch_nt_cfg<P, 2, 2, 4, 4, 1, 1>(a, s);  // 128 x 128, square hidden layer (the dominant launches)
  else if (a.N == a.Kp)
    launch_nt_cfg<P, 2, 2, 4, 4, 1, 0>(a, s);
  else if (dma)
    launch_nt_cfg<P, 2, 2, 4, 4, 0, 1>(a, s);  // 128 x 128
  else
    launch_nt_cfg<P, 2, 2, 4, 4, 0, 0>(a, s);
}
template void launch_gemm_nt<F32>(const GemmNT&, hipStream_t);
template void launch_gemm_nt<BF16>(const GemmNT&, hipStream_t);

// ------------------------------------------------------------------------------------------------
// gemm_tn  (weight gradients): C[n1][n2] = sum_m A[m][n1] * B[m][n2]
// ------------------------------------------------------------------------------------------------
// 128 x 128 output tile per block, 4 waves (2 x 2), each 64 x 64 = 4 x 4 MFMA tiles.  The
// contraction runs over batch rows, which are the slow dimension of both operands, so fragments
// need a transpose on the way out of LDS:
//   bf16: ds_read_b64_tr_b16 -- a 16-lane group reads a 4-row x 16-column block and gets it
//         column-major; lane group g takes rows 8g..8g+7 of a 32-row k-step in two reads.  Odd g
//         swap the two halves (a k-permutation common to A and B) so that with a 288-byte row
//         stride the 8 rows touched by a 32-lane half are distinct mod 8: conflict-free.
//   fp32: one dword per lane per MFMA, read straight down the column (stride 576 B, conflict-free).
// Block tile = (WA*TA*16) A-features x (WB*TB*16) B-features, WA x WB = 4 waves.  Two shapes are built:
//   <2,2,4,4> 128 x 128 : the H x H gradients
//   <4,1,8,4> 512 x  64 : thin outputs (dW0: H x in_dim; dWout computed transposed as H x out_dim) -- one block
//                         covers the whole output, all parallelism comes from the split over batch rows
// NBUF: LDS stages.  2: the next stage's registers are stored while slower waves may still multiply the current one.  1: one
// buffer, two barriers per stage -- half the LDS, so that THREE workgroups fit a CU (154 VGPRs allow three waves per SIMD):
// the kernel is paced by the per-stage chain inside a workgroup (transposed LDS reads → MFMAs → LDS stores → barrier), which
// more resident waves hide.
template <class P, int WA, int WB, int TA, int TB, int NBUF = 2>
__device__ __forceinline__ void tn_tile(const GemmTN& a, const int split, const int fa0, const int fb0) {
  constexpr int ES = P::ESIZE;
  constexpr int BA = WA * TA * 16, BB = WB * TB * 16;
  constexpr int ROWS = (ES == 2) ? 64 : 32;  // batch rows per LDS stage (two k-steps)
  constexpr int RSA = BA * ES + 32, RSB = BB * ES + 32;  // LDS row strides: +8 dwords => 8 rows hit 8 bank groups
  constexpr int CPA = BA * ES / 16, CPB = BB * ES / 16;  // 16-byte chunks per tile row
  constexpr int NCA = (ROWS * CPA + 255) / 256, NCB = (ROWS * CPB + 255) / 256;
  constexpr int OPA = ROWS * RSA, OPB = ROWS * RSB;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int wa = wid % WA, wb = wid / WA;
  const int m_begin = split * a.rows_per_split;
  const int m_end = min(a.M, m_begin + a.rows_per_split);
  const char* Ab = (const char*)a.A;
  const char* Bb = (const char*)a.B;

  u32x4 areg[NCA], breg[NCB];
  const int lim_a = a.ncol_a > 0 ? a.ncol_a : a.lda, lim_b = a.ncol_b > 0 ? a.ncol_b : a.ldb;
  auto gload = [&](int m0) {
#pragma unroll
    for (int i = 0; i < NCA; ++i) {
      const int q = (tid + i * 256) % (ROWS * CPA);
      const int rr = q / CPA, c = q % CPA;
      const int row = m0 + rr, ca = fa0 + c * (16 / ES);
      areg[i] = (u32x4){0, 0, 0, 0};
      if (row < m_end && ca < lim_a) areg[i] = *(const u32x4*)(Ab + ((size_t)row * a.lda + ca) * ES);
    }
#pragma unroll
    for (int i = 0; i < NCB; ++i) {
      const int q = (tid + i * 256) % (ROWS * CPB);
      const int rr = q / CPB, c = q % CPB;
      const int row = m0 + rr, cb = fb0 + c * (16 / ES);
      breg[i] = (u32x4){0, 0, 0, 0};
      if (row < m_end && cb < lim_b) breg[i] = *(const u32x4*)(Bb + ((size_t)row * a.ldb + cb) * ES);
    }
  };
  auto sstore = [&](int st) {
    char* As = smem + st * (OPA + OPB);
    char* Bs = As + OPA;
#pragma unroll
    for (int i = 0; i < NCA; ++i) {
      const int q = (tid + i * 256) % (ROWS * CPA);
      *(u32x4*)(As + (q / CPA) * RSA + (q % CPA) * 16) = areg[i];
    }
#pragma unroll
    for (int i = 0; i < NCB; ++i) {
      const int q = (tid + i * 256) % (ROWS * CPB);
      *(u32x4*)(Bs + (q / CPB) * RSB + (q % CPB) * 16) = breg[i];
    }
  };

  f32x4 acc[TA][TB];
#pragma unroll
  for (int i = 0; i < TA; ++i)
#pragma unroll
    for (int j = 0; j < TB; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nst = (m_end - m_begin + ROWS - 1) / ROWS;
  if (nst > 0) {
    gload(m_begin);
    sstore(0);
  }
  __syncthreads();
  for (int st = 0; st < nst; ++st) {
    const int cur = NBUF == 2 ? (st & 1) : 0;
    if (st + 1 < nst) gload(m_begin + (st + 1) * ROWS);
    const char* As = smem + cur * (OPA + OPB);
    const char* Bs = As + OPA;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      u32x4 af[TA], bf[TB];
      if constexpr (ES == 2) {
        // rows of this k-step: 32s .. 32s+31 ; lane group g: 32s + 8g + {0..7}; odd g swap the two 4-row halves
        // (a k-permutation common to A and B) so a 32-lane half touches 8 rows that are distinct mod 8
        typedef __attribute__((address_space(3))) i16x4 lds_v;
        const int q = r >> 2, p = r & 3;
        const int rbase = 32 * s + 8 * g;
        const int r0 = rbase + ((g & 1) ? 4 : 0) + q, r1 = rbase + ((g & 1) ? 0 : 4) + q;
#pragma unroll
        for (int t = 0; t < TA; ++t) {
          const int ca = (wa * TA * 16 + t * 16 + 4 * p) * 2;
          const u32x2 u0 = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v*)(As + r0 * RSA + ca)));
          const u32x2 u1 = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v*)(As + r1 * RSA + ca)));
          af[t] = (u32x4){u0.x, u0.y, u1.x, u1.y};
        }
#pragma unroll
        for (int t = 0; t < TB; ++t) {
          const int cb = (wb * TB * 16 + t * 16 + 4 * p) * 2;
          const u32x2 u0 = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v*)(Bs + r0 * RSB + cb)));
          const u32x2 u1 = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v*)(Bs + r1 * RSB + cb)));
          bf[t] = (u32x4){u0.x, u0.y, u1.x, u1.y};
        }
      } else {
        // rows of this k-step: 16s .. 16s+15 ; MFMA j of the step uses row 16s + 4j + g
#pragma unroll
        for (int t = 0; t < TA; ++t) {
          const int ca = (wa * TA * 16 + t * 16 + r) * 4;
          uint32_t v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = *(const uint32_t*)(As + (16 * s + 4 * j + g) * RSA + ca);
          af[t] = (u32x4){v[0], v[1], v[2], v[3]};
        }
#pragma unroll
        for (int t = 0; t < TB; ++t) {
          const int cb = (wb * TB * 16 + t * 16 + r) * 4;
          uint32_t v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = *(const uint32_t*)(Bs + (16 * s + 4 * j + g) * RSB + cb);
          bf[t] = (u32x4){v[0], v[1], v[2], v[3]};
        }
      }
#pragma unroll
      for (int i = 0; i < TA; ++i)
#pragma unroll
        for (int j = 0; j < TB; ++j) acc[i][j] = P::mma(af[i], bf[j], acc[i][j]);
    }
    if constexpr (NBUF == 1) __syncthreads();  // every wave is done with the buffer before it is overwritten
    if (st + 1 < nst) sstore(NBUF == 2 ? cur ^ 1 : 0);
    __syncthreads();
  }

  if (a.red_cnt != nullptr) {
    // Folded reduction: the slab is stored TRANSPOSED, [split][n2][n1] with N1 padded to N1p = round_up(N1, 4): a lane's four
    // values (rows 4g .. 4g+3 of C, one column) are then 16 contiguous bytes, the four lanes g of a column 64 -- and they go
    // out as write-through (sc1) 16-byte stores, which the tile's last workgroup reads back with sc1 loads: no L2
    // write-back / invalidate on either side (a release fence per wave here made the launch 5x longer: buffer_wbl2 flushes
    // the whole XCD's L2 every time).
    const int N1p = (a.N1 + 3) & ~3;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.slab, 0, (int)((size_t)a.splits * a.N2 * N1p * 4), 0x00020000);
#pragma unroll
    for (int i = 0; i < TA; ++i) {
      const int n1 = fa0 + wa * TA * 16 + i * 16 + 4 * g;
      if (n1 >= N1p) continue;
#pragma unroll
      for (int j = 0; j < TB; ++j) {
        const int n2 = fb0 + wb * TB * 16 + j * 16 + r;
        if (n2 < a.N2)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), rsrc,
                                                 (int)((((size_t)split * a.N2 + n2) * N1p + n1) * 4), 0, 16);  // aux 16 = sc1
      }
    }
    return;
  }
  float* out = a.slab + (size_t)split * a.N1 * a.ldc;
#pragma unroll
  for (int i = 0; i < TA; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int n1 = fa0 + wa * TA * 16 + i * 16 + 4 * g + e;
      if (n1 >= a.N1) continue;
#pragma unroll
      for (int j = 0; j < TB; ++j) {
        const int n2 = fb0 + wb * TB * 16 + j * 16 + r;
        if (n2 < a.N2) out[(size_t)n1 * a.ldc + n2] = acc[i][j][e];
      }
    }
}

// blockIdx.x = row split: workgroups are dealt round-robin to the 8 XCDs, so with splits % 8 == 0 every output tile
// of one split runs on the same XCD and its A/B row tiles are fetched into that XCD's L2 once, not once per XCD
template <class P, int WA, int WB, int TA, int TB>
__global__ __launch_bounds__(256) void gemm_tn_kernel(const GemmTN a) {
  tn_tile<P, WA, WB, TA, TB>(a, blockIdx.x, blockIdx.y * (WA * TA * 16), blockIdx.z * (WB * TB * 16));
}

// Every weight gradient of one backward pass in one launch: 128 x 128 tiles of all jobs, job after job (long ones
// first), within a job the split index fastest (bases and split counts are multiples of 8: one split's tiles share an
// XCD, as above).  No inter-kernel gaps, and the last round of one GEMM is filled by the first of the next.
template <class P, int NBUF>
__global__ __launch_bounds__(256, NBUF == 1 ? 3 : 2) void gemm_tn_group_kernel(const GemmTNGroup gr) {
  if ((int)blockIdx.x < gr.ex.n_blocks) {  // riders: nothing here depends on this launch's GEMM tiles (gemm.h, GemmTNExtra)
    extern __shared__ __attribute__((aligned(16))) char rider_smem[];
    const GemmTNExtra& e = gr.ex;
    int b = blockIdx.x;
    const bool wt = e.arrive_cnt != nullptr;
    const int n_prod = e.n_slot_blocks + 1 + (e.n_rjobs > 0 ? e.rjob_blocks[0] : 0) + (e.n_rjobs > 1 ? e.rjob_blocks[1] : 0) +
                       (e.n_rjobs > 2 ? e.rjob_blocks[2] : 0);
    if (b < n_prod) {  // producers
      if (b < e.n_slot_blocks) {
        const int slot = b / e.slot_bx, x = b - slot * e.slot_bx;
        if (slot < e.n_slots)
          slot_reduce_block256(e.colsum + (size_t)slot * e.tiles * e.width, e.tiles, e.width, e.slot_n[slot], e.slot_out[slot], x, wt);
      } else if (b == e.n_slot_blocks) {
        if (e.fin_stats != nullptr) loss_finalize_block256(e.fin_partial, e.fin_blocks, e.fin_moments, e.fin_stats, e.fin_part, e.fin_n_count);
      } else {
        b -= e.n_slot_blocks + 1;
        int j = 0;
        while (j + 1 < e.n_rjobs && b >= e.rjob_blocks[j]) b -= e.rjob_blocks[j], ++j;
        // (slab_job_block_wide strides by gridDim.x: give it this job's own block index and count)
        slab_job_rider<true>(e.rjob[j], b, e.rjob_blocks[j]);
      }
      if (wt) post_arrive(e.arrive_cnt);
      return;
    }
    b -= n_prod;
    if (e.post.G != nullptr) {  // consumers
      if (b < e.post.n_temb) {
        temb_g_block(e.post, b, (float*)rider_smem);
      } else if (b < e.post.n_temb + e.post.n_dw0t) {
        post_wait(e.post);
        dw0_temb_block(e.post, b - e.post.n_temb);
      }
    }
    return;
  }
  const int bid = blockIdx.x - gr.ex.n_blocks;
  int j = 0;
#pragma unroll
  for (int i = 1; i < MAX_TN_JOBS; ++i)
    if (i < gr.n && bid >= gr.base[i]) j = i;
  const GemmTN& a = gr.j[j];
  const int local = bid - gr.base[j];
  const int split = local % a.splits, tile = local / a.splits;
  const int tb = (a.N2 + 127) / 128;
  const int fa0 = (tile / tb) * 128, fb0 = (tile % tb) * 128;
  tn_tile<P, 2, 2, 4, 4, NBUF>(a, split, fa0, fb0);
  if (a.red_cnt == nullptr) return;
  // ---- folded slab reduction (guide section 6, guideline 16: every handed-off byte an sc1 store drained before ONE relaxed
  // agent-scope add; the workgroup whose add came last reads with sc1 loads behind its barrier)
  __shared__ int last_s;
  DPPO_HANDOVER_DRAIN();
  __syncthreads();
  if (threadIdx.x == 0)
    last_s = __hip_atomic_fetch_add(a.red_cnt + tile, 1u, DPPO_HANDOVER_ARRIVE_ORDER, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)a.splits - 1;
  __syncthreads();
  if (!last_s) return;
  DPPO_HANDOVER_ACQUIRE();
  if (threadIdx.x == 0) __hip_atomic_store(a.red_cnt + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next call
  const int N1p = (a.N1 + 3) & ~3;
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.slab, 0, (int)((size_t)a.splits * a.N2 * N1p * 4), 0x00020000);
  const int nr4 = (min(128, N1p - fa0) + 3) / 4, nc = min(128, a.N2 - fb0);  // the tile in units of (4 rows of C, 1 column)
  const size_t sstride = (size_t)a.N2 * N1p * 4;                             // bytes between two splits' slabs
  for (int i = threadIdx.x; i < nr4 * nc; i += 256) {
    const int c = fb0 + i / nr4, r0 = fa0 + (i % nr4) * 4;
    const size_t off = ((size_t)c * N1p + r0) * 4;
    f32x4 p[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) p[u] = (f32x4){0.f, 0.f, 0.f, 0.f};  // same summation tree per element as slab_job_block()
    int k = 0;
    for (; k + 8 <= a.splits; k += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
        p[u] += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(off + (k + u) * sstride), 0, 16));  // sc1
    }
    for (; k < a.splits; ++k)
      p[k & 7] += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(off + k * sstride), 0, 16));
    const f32x4 v = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int r = r0 + e;
      if (r >= a.N1) continue;
      if (a.red_n2a >= 0 && c >= a.red_n2a)
        a.red_out2[(size_t)r * a.red_ldo2 + (c - a.red_n2a)] = v[e];
      else if (a.red_transpose)
        a.red_out[(size_t)c * a.red_ldo + r] = v[e];
      else
        a.red_out[(size_t)r * a.red_ldo + c] = v[e];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// gemm_tn, LDS-DMA pipeline.  Same contraction and slab output as gemm_tn_kernel, but both operands go global -> LDS
// with global_load_lds (16 B per lane, no VGPR round trip) into a ring of NST stages of ROWS batch rows, so that
// NST - 1 stages (tens of KB per workgroup) are in flight while one is multiplied: the register-staged kernel has one
// stage of lookahead and waits out an HBM latency per 64 rows.  One barrier per stage.  A DMA writes LDS linearly
// (wave base + 16 * lane), so the XOR swizzle that keeps the transposed fragment reads conflict-free sits on the
// SOURCE side: slot p of tile row q receives source chunk p ^ swz(q).  The transposed reads fetch 8 B per lane, 32
// contiguous bytes per (row, 4 lanes), 8 rows per 32-lane half: the swizzle therefore permutes 32-byte units by the
// row's low 3 bits, swz(q) = (q & 7) << 1 (rows of 16+ chunks; narrower rows use the bits they have).  A DMA cannot
// zero-fill: lanes whose row or column does not exist read a 16-byte page of zeros instead.
__device__ u32x4 g_zero_chunk[1];
template <int CP>  // chunk-index XOR of tile row q (CP chunks of 16 B per row)
__device__ __forceinline__ constexpr int swz(int q) {
  return CP >= 16 ? (q & 7) << 1 : (CP >= 8 ? (q & 3) << 1 : 0);
}

template <class P, int WA, int WB, int TA, int TB, int NST, int KS>
__global__ __launch_bounds__(WA* WB * 64) void gemm_tn_dma_kernel(const GemmTN a) {
  constexpr int ES = P::ESIZE, NT = WA * WB * 64;
  constexpr int BA = WA * TA * 16, BB = WB * TB * 16;
  constexpr int ROWS = KS * ((ES == 2) ? 32 : 16);  // batch rows per stage (KS k-steps)
  constexpr int RSA = BA * ES, RSB = BB * ES;  // dense rows
  constexpr int CPA = RSA / 16, CPB = RSB / 16;  // 16-byte chunks per tile row (>= 8: the swizzle needs 3 bits)
  static_assert(CPA >= 8 && CPB >= 8 && (ROWS * CPA) % NT == 0 && (ROWS * CPB) % NT == 0, "tile / thread-count mismatch");
  constexpr int NCA = ROWS * CPA / NT, NCB = ROWS * CPB / NT, NLD = NCA + NCB;
  constexpr int OPA = ROWS * RSA, STAGE = ROWS * (RSA + RSB);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* glb_ptr;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int wa = wid % WA, wb = wid / WA;
  const int fa0 = blockIdx.y * BA, fb0 = blockIdx.z * BB;  // blockIdx.x = row split: see gemm_tn_kernel
  const int m_begin = blockIdx.x * a.rows_per_split;
  const int m_end = min(a.M, m_begin + a.rows_per_split);
  const char* Ab = (const char*)a.A;
  const char* Bb = (const char*)a.B;
  const char* zero = (const char*)g_zero_chunk;

  auto dma = [&](int stage_idx, int buf) {
    char* As = smem + buf * STAGE;
    char* Bs = As + OPA;
    const int m0 = m_begin + stage_idx * ROWS;
#pragma unroll
    for (int i = 0; i < NCA; ++i) {
      const int q = tid + i * NT;  // LDS chunk slot
      const int rr = q / CPA, c = (q % CPA) ^ swz<CPA>(rr);
      const int row = m0 + rr, ca = fa0 + c * (16 / ES);
      const char* src = (row < m_end && ca < a.lda) ? Ab + ((size_t)row * a.lda + ca) * ES : zero;
      __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)(As + (size_t)(i * NT + (tid & ~63)) * 16), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NCB; ++i) {
      const int q = tid + i * NT;
      const int rr = q / CPB, c = (q % CPB) ^ swz<CPB>(rr);
      const int row = m0 + rr, cb = fb0 + c * (16 / ES);
      const char* src = (row < m_end && cb < a.ldb) ? Bb + ((size_t)row * a.ldb + cb) * ES : zero;
      __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)(Bs + (size_t)(i * NT + (tid & ~63)) * 16), 16, 0, 0);
    }
  };
  // byte offset of logical (row, byte column) in a swizzled tile
  auto atA = [](int row, int byte_col) { return row * RSA + ((((byte_col >> 4) ^ swz<CPA>(row)) << 4) | (byte_col & 15)); };
  auto atB = [](int row, int byte_col) { return row * RSB + ((((byte_col >> 4) ^ swz<CPB>(row)) << 4) | (byte_col & 15)); };

  f32x4 acc[TA][TB];
#pragma unroll
  for (int i = 0; i < TA; ++i)
#pragma unroll
    for (int j = 0; j < TB; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nst = (m_end - m_begin + ROWS - 1) / ROWS;
#pragma unroll
  for (int p = 0; p < NST - 1; ++p)
    if (p < nst) dma(p, p);
  for (int st = 0; st < nst; ++st) {
    // this wave's share of stage st has landed once at most the DMAs of the NST - 2 younger stages are outstanding
    // (vmcnt counts in issue order); near the end fewer are in flight: wait for all
    if (st + NST - 2 < nst)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * NLD) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // everyone's share landed, and everyone is done reading the buffer refilled next
    if (st + NST - 1 < nst) dma(st + NST - 1, (st + NST - 1) % NST);
    const char* As = smem + (st % NST) * STAGE;
    const char* Bs = As + OPA;
#pragma unroll
    for (int s2 = 0; s2 < KS; ++s2) {
      u32x4 af[TA], bf[TB];
      if constexpr (ES == 2) {
        // rows of this k-step: 32 s2 .. 32 s2 + 31; lane group g: 32 s2 + 8g + {0..7}; odd g swap the two 4-row halves
        // (a k-permutation common to A and B)
        typedef __attribute__((address_space(3))) i16x4 lds_v;
        const int q = r >> 2, p = r & 3;
        const int rbase = 32 * s2 + 8 * g;
        const int r0 = rbase + ((g & 1) ? 4 : 0) + q, r1 = rbase + ((g & 1) ? 0 : 4) + q;
#pragma unroll
        for (int t = 0; t < TA; ++t) {
          const int ca = (wa * TA * 16 + t * 16 + 4 * p) * 2;
          const u32x2 u0 = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v*)(As + atA(r0, ca))));
          const u32x2 u1 = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v*)(As + atA(r1, ca))));
          af[t] = (u32x4){u0.x, u0.y, u1.x, u1.y};
        }
#pragma unroll
        for (int t = 0; t < TB; ++t) {
          const int cb = (wb * TB * 16 + t * 16 + 4 * p) * 2;
          const u32x2 u0 = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v*)(Bs + atB(r0, cb))));
          const u32x2 u1 = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v*)(Bs + atB(r1, cb))));
          bf[t] = (u32x4){u0.x, u0.y, u1.x, u1.y};
        }
      } else {
        // rows of this k-step: 16 s2 .. 16 s2 + 15 ; MFMA j of the step uses row 16 s2 + 4j + g
#pragma unroll
        for (int t = 0; t < TA; ++t) {
          const int ca = (wa * TA * 16 + t * 16 + r) * 4;
          uint32_t v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = *(const uint32_t*)(As + atA(16 * s2 + 4 * j + g, ca));
          af[t] = (u32x4){v[0], v[1], v[2], v[3]};
        }
#pragma unroll
        for (int t = 0; t < TB; ++t) {
          const int cb = (wb * TB * 16 + t * 16 + r) * 4;
          uint32_t v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = *(const uint32_t*)(Bs + atB(16 * s2 + 4 * j + g, cb));
          bf[t] = (u32x4){v[0], v[1], v[2], v[3]};
        }
      }
#pragma unroll
      for (int i = 0; i < TA; ++i)
#pragma unroll
        for (int j = 0; j < TB; ++j) acc[i][j] = P::mma(af[i], bf[j], acc[i][j]);
    }
  }

  float* out = a.slab + (size_t)blockIdx.x * a.N1 * a.ldc;
#pragma unroll
  for (int i = 0; i < TA; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int n1 = fa0 + wa * TA * 16 + i * 16 + 4 * g + e;
      if (n1 >= a.N1) continue;
#pragma unroll
      for (int j = 0; j < TB; ++j) {
        const int n2 = fb0 + wb * TB * 16 + j * 16 + r;
        if (n2 < a.N2) out[(size_t)n1 * a.ldc + n2] = acc[i][j][e];
      }
    }
}

template <class P, int WA, int WB, int TA, int TB, int NST, int KS = 2>
static void launch_tn_dma_cfg(const GemmTN& a, hipStream_t s) {
  constexpr int ES = P::ESIZE, BA = WA * TA * 16, BB = WB * TB * 16, ROWS = KS * ((ES == 2) ? 32 : 16);
  constexpr int LDS = NST * ROWS * (BA + BB) * ES;
  static_assert(LDS <= 160 * 1024, "ring does not fit LDS");
  dim3 grid(a.splits, (a.N1 + BA - 1) / BA, (a.N2 + BB - 1) / BB);
  static DevLatch attr_set;
  if (attr_set.need()) {
    (void)hipFuncSetAttribute((const void*)gemm_tn_dma_kernel<P, WA, WB, TA, TB, NST, KS>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_set.done();
  }
  const bool probe = a.N1 >= 128 && a.N2 >= 128 && probe_begin(PROBE_GEMM_TN, s);  // H x H gradients
  hipLaunchKernelGGL((gemm_tn_dma_kernel<P, WA, WB, TA, TB, NST, KS>), grid, dim3(WA * WB * 64), LDS, s, a);
  if (probe) probe_end(s, 2.0 * a.M * a.N1 * a.N2);
}

template <class P, int WA, int WB, int TA, int TB>
static void launch_tn_cfg(const GemmTN& a, hipStream_t s) {
  constexpr int ES = P::ESIZE, BA = WA * TA * 16, BB = WB * TB * 16, ROWS = (ES == 2) ? 64 : 32;
  constexpr int LDS = 2 * ROWS * ((BA * ES + 32) + (BB * ES + 32));
  dim3 grid(a.splits, (a.N1 + BA - 1) / BA, (a.N2 + BB - 1) / BB);
  static DevLatch attr_set;  // > 64 KiB of dynamic LDS needs the cap raised once per kernel
  if (attr_set.need()) {
    (void)hipFuncSetAttribute((const void*)gemm_tn_kernel<P, WA, WB, TA, TB>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              LDS);
    attr_set.done();
  }
  const bool probe = BA == 128 && a.N1 >= 128 && a.N2 >= 128 && probe_begin(PROBE_GEMM_TN, s);  // H x H gradients
  hipLaunchKernelGGL((gemm_tn_kernel<P, WA, WB, TA, TB>), grid, dim3(256), LDS, s, a);
  if (probe) probe_end(s, 2.0 * a.M * a.N1 * a.N2);
}

// tuning knob 26: LDS stages of the grouped weight-gradient kernel: 1 (three workgroups per CU), 2 (two, double-buffered), or
// 0 (default): by the group's size -- one stage while all its workgroups are co-resident at three per CU (hopper: 576 of 768),
// two when they are not and the minibatch is large, 32,768 samples or more (a group that still carries the first layer's product: halfcheetah 0.393 ->
// 0.387 ms; hopper 0.347 -> 0.356 and can, 7,500 samples, 0.181 -> 0.185 the other way round)
static int g_tn_nbuf = 0;
void set_gemm_tn_nbuf(int v) { g_tn_nbuf = v == 2 ? 2 : (v == 1 ? 1 : 0); }
template <class P>
void launch_gemm_tn_group(const GemmTNGroup& gr, hipStream_t s) {
  if (gr.n <= 0) return;
  constexpr int ES = P::ESIZE, ROWS = (ES == 2) ? 64 : 32;
  constexpr int LDS = 2 * ROWS * 2 * (128 * ES + 32);
  static DevLatch attr_set;
  if (attr_set.need()) {
    (void)hipFuncSetAttribute((const void*)gemm_tn_group_kernel<P, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    attr_set.done();
  }
  double flops = 0, bytes = 0;
  for (int i = 0; i < gr.n; ++i) {
    const GemmTN& a = gr.j[i];
    flops += 2.0 * a.M * a.N1 * a.N2;
    bytes += (double)a.M * (a.N1 + a.N2) * ES + 4.0 * a.N1 * a.N2;  // both operands once + the fp32 result
  }
  const bool probe = probe_begin(PROBE_GEMM_TN, s);
  const dim3 grid(gr.base[gr.n] + gr.ex.n_blocks);
  const int nbuf = g_tn_nbuf != 0 ? g_tn_nbuf : ((int)grid.x > 3 * 256 && gr.n > 0 && gr.j[0].M >= 32768 ? 2 : 1);
  if (nbuf == 1)
    hipLaunchKernelGGL((gemm_tn_group_kernel<P, 1>), grid, dim3(256), LDS / 2, s, gr);
  else
    hipLaunchKernelGGL((gemm_tn_group_kernel<P, 2>), grid, dim3(256), LDS, s, gr);
  if (probe) probe_end(s, flops, bytes);
}
template void launch_gemm_tn_group<F32>(const GemmTNGroup&, hipStream_t);
template void launch_gemm_tn_group<BF16>(const GemmTNGroup&, hipStream_t);

// ------------------------------------------------------------------------------------------------
// gemm_tn from K-major fragment operands (gemm.h, GemmTNFrag): one WAVE owns a 64 x (16 TB) tile of C for one row split and
// walks its k-steps with D of them in flight in registers (4 + TB fragments of 1 KB each per k-step); a workgroup is four such
// waves on neighbouring tiles (2 x 2, or 4 x 1 for thin outputs) that share operand fragments through L1 -- and nothing else:
// no LDS, no barrier, no transposed read.  Loads return in issue order and the compiler counts vmcnt from the register
// dependences, so the loop body is "multiply slot d, refill slot d".
// Work order inside a job: split index fastest (multiples of 8 => a split's tiles share an XCD and its L2, as in
// gemm_tn_group_kernel).
typedef __attribute__((address_space(3))) int lds_int;
template <int TB, int D>
__device__ __forceinline__ void tn_frag_wave(const GemmTNFrag& a, const int split, const int ablk, const int bblk, const int lane,
                                             lds_int* progress) {  // progress: non-null on the wave that paces the prefetcher
  constexpr int TA = 4;
  const int r = lane & 15, g = lane >> 4;
  const int ks0 = split * a.ks_per_split;
  int ks1 = ks0 + a.ks_per_split;
  ks1 = ks1 < a.ks_total ? ks1 : a.ks_total;
  const int fa0 = ablk * TA, fb0 = bblk * TB;  // first feature tiles
  // tiles past the tensors' widths (N1 not a multiple of 64, ...) are clamped to the last one and never stored
  int ta[TA], tbv[TB];
#pragma unroll
  for (int i = 0; i < TA; ++i) ta[i] = fa0 + i < a.nta ? fa0 + i : a.nta - 1;
#pragma unroll
  for (int j = 0; j < TB; ++j) tbv[j] = fb0 + j < a.ntb ? fb0 + j : a.ntb - 1;
  f32x4 acc[TA][TB];
#pragma unroll
  for (int i = 0; i < TA; ++i)
#pragma unroll
    for (int j = 0; j < TB; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  u32x4 fa[D][TA], fb[D][TB];
  const u32x4* Ap = a.A + lane;
  const u32x4* Bp = a.B + lane;
  auto fetch = [&](int d, int ks) {
    ks = ks < ks1 ? ks : ks1 - 1;  // past the end: a harmless reload of the last k-step (never multiplied)
    const u32x4* pa = Ap + (size_t)ks * a.nta * 64;
    const u32x4* pb = Bp + (size_t)ks * a.ntb * 64;
#pragma unroll
    for (int i = 0; i < TA; ++i) fa[d][i] = pa[ta[i] * 64];
#pragma unroll
    for (int j = 0; j < TB; ++j) fb[d][j] = pb[tbv[j] * 64];
  };
  if (ks1 > ks0) {
#pragma unroll
    for (int d = 0; d < D; ++d) fetch(d, ks0 + d);
    for (int ks = ks0; ks < ks1; ks += D) {
      if (progress != nullptr && lane == 0) __hip_atomic_store(progress, ks, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
      for (int d = 0; d < D; ++d) {
        if (ks + d < ks1) {
#pragma unroll
          for (int i = 0; i < TA; ++i)
#pragma unroll
            for (int j = 0; j < TB; ++j) acc[i][j] = BF16::mma(fa[d][i], fb[d][j], acc[i][j]);
        }
        fetch(d, ks + d + D);
      }
    }
  }
  float* out = a.slab + (size_t)split * a.N1 * a.ldc;
#pragma unroll
  for (int i = 0; i < TA; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int n1 = (fa0 + i) * 16 + 4 * g + e;
      if (n1 >= a.N1) continue;
#pragma unroll
      for (int j = 0; j < TB; ++j) {
        const int n2 = (fb0 + j) * 16 + r;
        if (n2 < a.N2) out[(size_t)n1 * a.ldc + n2] = acc[i][j][e];
      }
    }
}

// The same contraction with the workgroup's fragments shared through an LDS ring, behind a deep L2 PREFETCH by a fifth wave.
// Measured (rocprofv3 --pmc, gpurun_out/r3g): in every form of this product tried here -- the transposing LDS kernel
// (gemm_tn_group_kernel, 75 us per actor group), the register-only form below (80 us), a plain LDS ring (80 us) -- the waves
// spend 60 % of their cycles in s_waitcnt / barriers with the matrix cores at ~15 %: 39 % of the L2 requests miss (every line is
// wanted by several output tiles at about the same time and the first asker pays the HBM latency), vmcnt is in ISSUE ORDER,
// and so each k-step of a ring waits for its slowest line: the look-ahead that should hide HBM is spent waiting on it.
// A prefetch issued by the consuming waves themselves does not help (it sits in the same in-order queue in front of the
// ring loads: knob 33 had no effect at any distance).  So the roles are split by WAVE:
//  * waves 0-3 (2 x 2 tiles of 64 x 64, or 4 x 1 for thin outputs) fetch each fragment of the workgroup's tile ONCE per
//    k-step, by LDS-DMA (1 KB per wave instruction, landed lane-linear), into NST one-k-step stages, and read their 4 + TB
//    fragments back with conflict-free ds_read_b128;
//  * wave 4 computes nothing: per k-step it issues this workgroup's share of the split's fragments (the workgroups of one row
//    split -- same XCD, same L2: split index fastest, splits a multiple of 8 -- divide each k-step's fragments among them)
//    PFD k-steps ahead, by LDS-DMA into a dump slot nobody reads, never waits for them, and joins the k-step's barrier to
//    keep pace.  Its vmcnt is its own: HBM latency lands on loads nobody waits for, the ring sees L2 hits.
template <int WGA, int TB, int NST, int NPW>
__device__ __forceinline__ void tn_fragl_wg(const GemmTNFrag& a, const int split, const int tile, char* smem) {
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* glb_ptr;
  constexpr int TA = 4, WGB = 4 / WGA, FA = WGA * TA, FB = WGB * TB, NF = FA + FB, NLD = (NF + 3) / 4;
  constexpr int STAGE = NLD * 4 * 1024;  // bytes; slots NF .. 4 NLD - 1 receive dummy loads (same vmcnt count on every wave)
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int wa = wid % WGA, wb = (wid & 3) / WGA;
  const int nab = ((a.N1 + 63) / 64 + WGA - 1) / WGA;
  const int nbb = ((a.N2 + 16 * TB - 1) / (16 * TB) + WGB - 1) / WGB;  // workgroup tiles along A, B
  const int fa0 = (tile / nbb) * FA, fb0 = (tile % nbb) * FB;           // the workgroup's first feature tiles
  const int ks0 = split * a.ks_per_split;
  int ks1 = ks0 + a.ks_per_split;
  ks1 = ks1 < a.ks_total ? ks1 : a.ks_total;
  const int nks = ks1 - ks0;
  if (wid == 4) {  // ---- the prefetcher
    if (nks <= 0) return;
    const int nAu = min(a.nta, (a.N1 + 15) / 16), nBu = min(a.ntb, (a.N2 + 15) / 16), G = nab * nbb;
    const u32x4* psrc[NPW];
    int pstride[NPW];
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      int f = tile + G * i;
      f = f < nAu + nBu ? f : nAu + nBu - 1;  // (nothing left for this workgroup: a duplicate -- one issue pattern)
      const bool isA = f < nAu;
      psrc[i] = (isA ? a.A + (size_t)f * 64 : a.B + (size_t)(f - nAu) * 64) + lane;
      pstride[i] = (isA ? a.nta : a.ntb) * 64;
    }
    char* dump = smem + NST * STAGE;
    const int lead = NST - 1 + a.pfd;
    for (int st = 0; st < nks; ++st) {
      int kp = ks0 + st + lead;
      if (kp < ks1 && !(a.dbg & 4)) {
#pragma unroll
        for (int i = 0; i < NPW; ++i)
          __builtin_amdgcn_global_load_lds((glb_ptr)(psrc[i] + (size_t)kp * pstride[i]), (lds_ptr)dump, 16, 0, 0);
      }
      __syncthreads();
    }
    return;
  }
  if (nks <= 0) {  // a surplus split: zeros (whole workgroup: no barrier is left behind)
    float* out = a.slab + (size_t)split * a.N1 * a.ldc;
#pragma unroll
    for (int i = 0; i < TA; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int n1 = (fa0 + wa * TA + i) * 16 + 4 * g + e;
        if (n1 >= a.N1) continue;
#pragma unroll
        for (int j = 0; j < TB; ++j) {
          const int n2 = (fb0 + wb * TB + j) * 16 + r;
          if (n2 < a.N2) out[(size_t)n1 * a.ldc + n2] = 0.f;
        }
      }
    return;
  }
  // this wave's NLD loads of a stage: slot f = wid + 4 i -> (operand, feature tile), clamped to the tensor's last tile
  const u32x4* src[NLD];
  int kstride[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int f = wid + 4 * i;
    const bool isA = f < FA || f >= NF;  // (dummy slots reload an A fragment)
    int t = f < FA ? fa0 + f : (f < NF ? fb0 + (f - FA) : fa0);
    const int nt = isA ? a.nta : a.ntb;
    t = t < nt ? t : nt - 1;
    src[i] = (isA ? a.A : a.B) + (size_t)t * 64 + lane;
    kstride[i] = nt * 64;
  }
  auto dma = [&](int ks, int buf) {
    char* st = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < NLD; ++i)
      __builtin_amdgcn_global_load_lds((glb_ptr)(src[i] + (size_t)ks * kstride[i]), (lds_ptr)(st + (wid + 4 * i) * 1024), 16, 0, 0);
  };
  f32x4 acc[TA][TB];
#pragma unroll
  for (int i = 0; i < TA; ++i)
#pragma unroll
    for (int j = 0; j < TB; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int p = 0; p < NST - 1; ++p)
    if (p < nks) dma(ks0 + p, p);
  for (int st = 0; st < nks; ++st) {
    // this wave's share of stage st has landed once at most the DMAs of the NST - 2 younger stages are outstanding (vmcnt
    // counts in issue order); near the end fewer are in flight: wait for all
    if (st + NST - 2 < nks)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * NLD) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // everyone's share landed, and everyone is done reading the stage refilled next
    if (st + NST - 1 < nks && !(a.dbg & 2)) dma(ks0 + st + NST - 1, (st + NST - 1) % NST);
    const char* sb = smem + (st % NST) * STAGE + lane * 16;
    u32x4 af[TA], bf[TB];
#pragma unroll
    for (int i = 0; i < TA; ++i) af[i] = *(const u32x4*)(sb + (wa * TA + i) * 1024);
#pragma unroll
    for (int j = 0; j < TB; ++j) bf[j] = *(const u32x4*)(sb + (FA + wb * TB + j) * 1024);
    if (!(a.dbg & 1)) {
#pragma unroll
      for (int i = 0; i < TA; ++i)
#pragma unroll
        for (int j = 0; j < TB; ++j) acc[i][j] = BF16::mma(af[i], bf[j], acc[i][j]);
    } else {  // (timing experiment: keep the LDS reads alive)
      acc[0][0][0] += __uint_as_float(af[0].x ^ af[1].x ^ af[2].x ^ af[3].x ^ bf[0].x ^ bf[TB - 1].y);
    }
  }
  float* out = a.slab + (size_t)split * a.N1 * a.ldc;
#pragma unroll
  for (int i = 0; i < TA; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int n1 = (fa0 + wa * TA + i) * 16 + 4 * g + e;
      if (n1 >= a.N1) continue;
#pragma unroll
      for (int j = 0; j < TB; ++j) {
        const int n2 = (fb0 + wb * TB + j) * 16 + r;
        if (n2 < a.N2) out[(size_t)n1 * a.ldc + n2] = acc[i][j][e];
      }
    }
}
// ring stages: 4 x 16 KB (2 x 2 waves) or 3 x 20 KB (4 x 1), + the prefetcher's 1 KB dump slot: 65 KB, two workgroups per CU
constexpr int TN_FRAGL_LDS = 4 * 16 * 1024 + 1024;
static_assert(3 * 20 * 1024 + 1024 <= TN_FRAGL_LDS, "thin configuration's ring");
__global__ __launch_bounds__(320, 2) void gemm_tn_fragl_kernel(const GemmTNFragGroup gr) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int jn = 0;
#pragma unroll
  for (int i = 1; i < MAX_TN_JOBS; ++i)
    if (i < gr.n && (int)blockIdx.x >= gr.base[i]) jn = i;
  const GemmTNFrag& a = gr.j[jn];
  const int local = blockIdx.x - gr.base[jn];
  const int split = local % a.splits, tile = local / a.splits;
  // (npf: fragments per k-step / workgroups per split, capped at what is instantiated: gemm_tn_frag_prepare)
  if (a.wga == 2) {
    if (a.npf <= 4)
      tn_fragl_wg<2, 4, 4, 4>(a, split, tile, smem);
    else
      tn_fragl_wg<2, 4, 4, 8>(a, split, tile, smem);
  } else if (a.tb == 4) {
    tn_fragl_wg<4, 4, 3, 8>(a, split, tile, smem);
  } else if (a.tb == 2) {
    tn_fragl_wg<4, 2, 3, 8>(a, split, tile, smem);
  } else {
    tn_fragl_wg<4, 1, 3, 8>(a, split, tile, smem);
  }
}

// ... with a fifth, prefetching wave per workgroup (see gemm_tn_fragl_kernel): it fetches this workgroup's share of the row
// split's fragments a.pfd k-steps ahead of wave 0's progress (an LDS word), by LDS-DMA into a dump slot, and waits for nothing.
template <int D>
__global__ __launch_bounds__(320, 2) void gemm_tn_frag_kernel(const GemmTNFragGroup gr) {
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* glb_ptr;
  __shared__ __attribute__((aligned(16))) char dump[1024];
  __shared__ int prog;
  int jn = 0;
#pragma unroll
  for (int i = 1; i < MAX_TN_JOBS; ++i)
    if (i < gr.n && (int)blockIdx.x >= gr.base[i]) jn = i;
  const GemmTNFrag& a = gr.j[jn];
  const int local = blockIdx.x - gr.base[jn];
  const int split = local % a.splits, tile = local / a.splits;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nab = ((a.N1 + 63) / 64 + a.wga - 1) / a.wga;
  const int nbb = ((a.N2 + 16 * a.tb - 1) / (16 * a.tb) + a.wgb - 1) / a.wgb;  // workgroup tiles along A, B
  const int ks0 = split * a.ks_per_split;
  int ks1 = ks0 + a.ks_per_split;
  ks1 = ks1 < a.ks_total ? ks1 : a.ks_total;
  if (threadIdx.x == 0) prog = ks0;
  __syncthreads();
  if (wid == 4) {  // ---- the prefetcher
    if (a.pfd <= 0 || (a.dbg & 4)) return;
    const int nAu = min(a.nta, (a.N1 + 15) / 16), nBu = min(a.ntb, (a.N2 + 15) / 16), G = nab * nbb;
    constexpr int NPW = 8;
    const int npw = a.npf < NPW ? a.npf : NPW;
    for (int kp = ks0 + D; kp < ks1; ++kp) {
      // stay at most pfd k-steps ahead of wave 0 (which may itself have left already: then run to the end)
      while (kp - __hip_atomic_load((lds_int*)&prog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) > a.pfd + D) __builtin_amdgcn_s_sleep(8);
      for (int i = 0; i < npw; ++i) {
        int f = tile + G * i;
        if (f >= nAu + nBu) break;
        const bool isA = f < nAu;
        const u32x4* src = (isA ? a.A + ((size_t)kp * a.nta + f) * 64 : a.B + ((size_t)kp * a.ntb + (f - nAu)) * 64) + lane;
        __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)dump, 16, 0, 0);
      }
    }
    return;
  }
  const int ablk = (tile / nbb) * a.wga + wid % a.wga, bblk = (tile % nbb) * a.wgb + wid / a.wga;
  if (ablk * 64 >= a.N1 || bblk * 16 * a.tb >= a.N2) {
    if (wid == 0 && lane == 0) prog = 0x3fffffff;  // (nothing to pace: let the prefetcher run out)
    return;
  }
  lds_int* pp = wid == 0 ? (lds_int*)&prog : nullptr;
  if (a.tb == 4)
    tn_frag_wave<4, D>(a, split, ablk, bblk, lane, pp);
  else if (a.tb == 2)
    tn_frag_wave<2, D>(a, split, ablk, bblk, lane, pp);
  else
    tn_frag_wave<1, D>(a, split, ablk, bblk, lane, pp);
  if (wid == 0 && lane == 0) __hip_atomic_store((lds_int*)&prog, 0x3fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
int gemm_tn_frag_blocks(const GemmTNFrag& j) {
  const int nab = ((j.N1 + 63) / 64 + j.wga - 1) / j.wga;
  const int nbb = ((j.N2 + 16 * j.tb - 1) / (16 * j.tb) + j.wgb - 1) / j.wgb;
  return nab * nbb * j.splits;
}
static int g_tn_frag_dbg = 0;  // tuning knob 34 (timing experiments, results wrong): 1 no MFMAs, 2 no ring loads, 4 no prefetch
void set_gemm_tn_frag_dbg(int v) { g_tn_frag_dbg = v; }
static int g_tn_frag_pfd = 12;  // tuning knob 33: k-steps the L2 prefetch runs ahead of the ring's own loads
void set_gemm_tn_frag_pfd(int v) { g_tn_frag_pfd = v < 0 ? 0 : (v > 64 ? 64 : v); }
// wave shape of a job's workgroups and its prefetch share; false if the LDS-ring kernel has no configuration for it
bool gemm_tn_frag_prepare(GemmTNFrag& j) {
  j.tb = j.N2 <= 16 ? 1 : (j.N2 <= 32 ? 2 : 4);
  const int nba = (j.N1 + 63) / 64, nbbk = (j.N2 + 16 * j.tb - 1) / (16 * j.tb);
  j.wgb = nbbk >= 2 ? 2 : 1, j.wga = 4 / j.wgb;
  const int nab = (nba + j.wga - 1) / j.wga, nbb = (nbbk + j.wgb - 1) / j.wgb;
  const int nAu = j.nta < (j.N1 + 15) / 16 ? j.nta : (j.N1 + 15) / 16, nBu = j.ntb < (j.N2 + 15) / 16 ? j.ntb : (j.N2 + 15) / 16;
  j.npf = (nAu + nBu + nab * nbb - 1) / (nab * nbb);  // fragments per k-step and workgroup of a split
  j.npf = j.npf < 8 ? j.npf : 8;  // what the kernel instantiates (4 or 8 per k-step: the prefetcher's vmcnt holds 63); a job
                                  // with more prefetches the first ones only
  j.pfd = g_tn_frag_pfd;
  j.dbg = g_tn_frag_dbg;
  return true;
}
static int g_tn_frag_depth = 0;  // tuning knob 32: 0 (default) the LDS-ring kernel; 2..4: the register-only kernel at that depth
void set_gemm_tn_frag_depth(int v) { g_tn_frag_depth = v <= 0 ? 0 : (v < 2 ? 2 : (v > 4 ? 4 : v)); }
void launch_gemm_tn_frag_group(const GemmTNFragGroup& gr, int64_t M, hipStream_t s) {
  if (gr.n <= 0) return;
  double flops = 0, bytes = 0;
  for (int i = 0; i < gr.n; ++i) {
    const GemmTNFrag& a = gr.j[i];
    flops += 2.0 * M * a.N1 * a.N2;
    bytes += (double)M * (a.N1 + a.N2) * 2 + 4.0 * a.N1 * a.N2;  // both operands once + the fp32 result
  }
  const bool probe = probe_begin(PROBE_GEMM_TN, s);
  const dim3 grid(gr.base[gr.n]);
  if (g_tn_frag_depth == 0) {
    static DevLatch attr_set;
    if (attr_set.need()) {
      (void)hipFuncSetAttribute((const void*)gemm_tn_fragl_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, TN_FRAGL_LDS);
      attr_set.done();
    }
    hipLaunchKernelGGL(gemm_tn_fragl_kernel, grid, dim3(320), TN_FRAGL_LDS, s, gr);
  } else if (g_tn_frag_depth == 2)
    hipLaunchKernelGGL((gemm_tn_frag_kernel<2>), grid, dim3(320), 0, s, gr);
  else if (g_tn_frag_depth == 4)
    hipLaunchKernelGGL((gemm_tn_frag_kernel<4>), grid, dim3(320), 0, s, gr);
  else
    hipLaunchKernelGGL((gemm_tn_frag_kernel<3>), grid, dim3(320), 0, s, gr);
  if (probe) probe_end(s, flops, bytes);
}

static int g_tn_thin = 1;  // tuning knob 6: 0 = no one-tile 512 x 64 configuration for thin outputs
void set_gemm_tn_thin(int v) { g_tn_thin = v; }
bool gemm_tn_thin(int N1, int N2) { return g_tn_thin && N2 <= 64 && N1 > 64; }

// 0 (default): register-staged kernel; 1..8: an LDS-DMA ring configuration (tools/gemm_bench.py sweeps them); -1: ring
// variant 2 for outputs of 512 x 512 and more.  Alone on the chip the ring wins at 512 x 512 (43.8 vs 51.6 us incl. the
// slab reduce, M = 50,000, bf16) and loses at 256 x 256 (34.7 vs 30.1); inside the update step, next to the other
// stream's kernels, it does not pay (65.9 vs 66.9 M samples/s), so it stays an option.  Either way the contraction is
// HBM-bound here (195 FLOP per byte of operand + slab traffic against a machine balance of ~310).
static int g_tn_variant = 0;
void set_gemm_tn_variant(int v) { g_tn_variant = v; }

template <class P>
void launch_gemm_tn(const GemmTN& a, hipStream_t s) {
  if (a.ncol_a > 0 || a.ncol_b > 0) {  // overlapping rows: the register-staged 128 x 128 kernel knows about them
    launch_tn_cfg<P, 2, 2, 4, 4>(a, s);
    return;
  }
  if (gemm_tn_thin(a.N1, a.N2)) {
    launch_tn_cfg<P, 4, 1, 8, 4>(a, s);
    return;
  }
  const int variant = g_tn_variant >= 0 ? g_tn_variant : ((size_t)a.N1 * a.N2 >= 512 * 512 ? 2 : 0);
  switch (variant) {
    case 1: launch_tn_dma_cfg<P, 2, 2, 4, 4, 3>(a, s); break;  // 128 x 128, 4 waves, 3 stages (96 KB)
    case 2: launch_tn_dma_cfg<P, 2, 2, 4, 4, 2>(a, s); break;  // 128 x 128, 4 waves, 2 stages (2 workgroups / CU)
    case 3: launch_tn_dma_cfg<P, 4, 2, 4, 4, 3>(a, s); break;  // 256 x 128, 8 waves, 3 stages (144 KB)
    case 4: launch_tn_dma_cfg<P, 2, 2, 4, 4, 4>(a, s); break;  // 128 x 128, 4 waves, 4 stages (128 KB)
    case 5: launch_tn_dma_cfg<P, 2, 2, 8, 4, 2, 1>(a, s); break;  // 256 x 128, 4 waves of 128 x 64, 2 stages of 1 k-step (48 KB)
    case 6: launch_tn_dma_cfg<P, 2, 2, 8, 4, 3, 1>(a, s); break;  // same, 3 stages (72 KB)
    case 7: launch_tn_dma_cfg<P, 2, 4, 8, 4, 2, 2>(a, s); break;  // 256 x 256, 8 waves of 128 x 64, 2 stages of 2 k-steps (128 KB)
    case 8: launch_tn_dma_cfg<P, 2, 2, 8, 4, 2, 2>(a, s); break;  // 256 x 128, 4 waves, 2 stages of 2 k-steps (96 KB)
    default: launch_tn_cfg<P, 2, 2, 4, 4>(a, s);
  }
}
template void launch_gemm_tn<F32>(const GemmTN&, hipStream_t);
template void launch_gemm_tn<BF16>(const GemmTN&, hipStream_t);

// ------------------------------------------------------------------------------------------------
// small reductions
// ------------------------------------------------------------------------------------------------
__global__ void slab_reduce_kernel(const float* slab, int splits, size_t n, float* out, float scale) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // fixed summation tree (reproducible): 8 independent chains keep 8 loads in flight instead of one
  float p[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int k = 0;
  for (; k + 8 <= splits; k += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) p[u] += slab[(size_t)(k + u) * n + i];
  }
  for (; k < splits; ++k) p[k & 7] += slab[(size_t)k * n + i];
  out[i] = (((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]))) * scale;
}
// few columns, many partial rows (bias gradients over hundreds of tiles): 64 columns x 16 row-lanes per block,
// 4 independent chains per lane, fixed combine order => reproducible
__global__ __launch_bounds__(1024) void reduce_rows_kernel(const float* in, int rows, size_t n, float* out, float scale) {
  __shared__ float red[16][65];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const size_t c = (size_t)blockIdx.x * 64 + cl;
  float p[4] = {0.f, 0.f, 0.f, 0.f};
  if (c < n) {
    int r = rl;
    for (; r + 48 < rows; r += 64) {
#pragma unroll
      for (int u = 0; u < 4; ++u) p[u] += in[(size_t)(r + 16 * u) * n + c];
    }
    for (; r < rows; r += 16) p[0] += in[(size_t)r * n + c];
  }
  red[rl][cl] = (p[0] + p[1]) + (p[2] + p[3]);
  __syncthreads();
  if (rl == 0 && c < n) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += red[i][cl];
    out[c] = s * scale;
  }
}

void launch_slab_reduce(const float* slab, int splits, size_t n, float* out, float scale, hipStream_t s) {
  if (n == 0) return;
  if (n <= 16384 && splits >= 32) {
    hipLaunchKernelGGL(reduce_rows_kernel, dim3((unsigned)((n + 63) / 64)), dim3(1024), 0, s, slab, splits, n, out, scale);
    return;
  }
  hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, slab, splits, n, out,
                     scale);
}

// partial[b][j] = sum over this block's row range of A[m][j]; 256 threads = 4 row-lanes x 64 column-lanes
template <class P>
__global__ __launch_bounds__(256) void colsum_kernel(const void* A_, int M, int N, int lda, float* partial) {
  typedef typename P::elem_t E;
  const E* A = (const E*)A_;
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int rows_per_block = (M + gridDim.x - 1) / gridDim.x;
  const int m0 = blockIdx.x * rows_per_block, m1 = min(M, m0 + rows_per_block);
  for (int c0 = 0; c0 < N; c0 += 64) {
    const int c = c0 + cl;
    float s = 0.f;
    if (c < N)
      for (int m = m0 + rl; m < m1; m += 4) s += P::to_f32(A[(size_t)m * lda + c]);
    red[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && c < N) partial[(size_t)blockIdx.x * N + c] = red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl];
    __syncthreads();
  }
}
template <class P>
void launch_colsum(const void* A, int M, int N, int lda, float* partial, int blocks, float* out, float scale,
                   hipStream_t s) {
  hipLaunchKernelGGL((colsum_kernel<P>), dim3(blocks), dim3(256), 0, s, A, M, N, lda, partial);
  launch_slab_reduce(partial, blocks, (size_t)N, out, scale, s);
}
template void launch_colsum<F32>(const void*, int, int, int, float*, int, float*, float, hipStream_t);
template void launch_colsum<BF16>(const void*, int, int, int, float*, int, float*, float, hipStream_t);

}  // namespace dppo
