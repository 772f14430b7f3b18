// See fused.h.  gfx950 only.
#include <type_traits>

#include "fused.h"
#include "gemm.h"
#include "tile_ln.h"
#include "pack_dev.h"
#include "loss_dev.h"

#ifndef DPPO_BWD_LATE
#define DPPO_BWD_LATE 1
#endif
#ifndef DPPO_BWD_PREFETCH
#define DPPO_BWD_PREFETCH 1  // the in-kernel-dW0 backward requests its next tile's small operands a phase early (0: at the tile's start, for A/B runs)
#endif
#ifndef DPPO_FLAGS
#define DPPO_FLAGS 1  // layer hand-over by per-wave LDS flags instead of workgroup barriers (0: barriers, for A/B runs)
#endif

namespace dppo {

// Phase stamps (debug build only: DPPO_STAMPS=1 build.sh -> libdppo_hip_stamps.so, tools/fused_bench.py --stamps):
// s_memtime of every wave of workgroup 0 at the phase boundaries of its second tile.
#ifdef DPPO_STAMPS
__device__ unsigned long long g_stamps[8][32];
#define STAMP(i)                                                                   \
  do {                                                                             \
    if (blockIdx.x == 0 && lane == 0 && tile == (int)gridDim.x) g_stamps[wid][i] = clock64(); \
  } while (0)
#else
#define STAMP(i) \
  do {           \
  } while (0)
#endif

namespace {

// the layer hand-over flags live in LDS and are accessed as workgroup-scope relaxed atomics through an LDS-address-space
// pointer: `volatile` on a generic pointer compiles to FLAT accesses with system coherence bits and an s_waitcnt vmcnt(0)
// behind each one -- every flag access then drained the weight ring and the emit's stores
typedef __attribute__((address_space(3))) uint32_t lds_u32;
__device__ __forceinline__ uint32_t flag_load(lds_u32* f) {
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}

__device__ __forceinline__ int kmask16(int rb) {
  const int n = rb >> 4;
  const int p = n & (-n);
  return (p > 16 ? 16 : p) - 1;
}

// ring depth (k-step positions in flight per wave): 4 unless the accumulators already fill the register file
template <int TPW, int MR>
constexpr int ring_depth() {
  return (TPW >= 8 || TPW * MR > 16) ? 2 : 4;
}

// rows of a [rows][ld] elem matrix -> swizzled LDS image [MT][rb bytes]; 16-byte chunks, zero fill outside
template <int MT>
__device__ __forceinline__ void load_tile(char* dst, int rb, int km, const char* src, int ld_bytes, int row0, int M) {
  const int nch = rb >> 4, vch = ld_bytes >> 4;
  int t0 = threadIdx.x;
  asm volatile("" : "+v"(t0));  // addresses recomputed per tile, never hoisted out of the tile loop and spilled (see emit())
  for (int q = t0; q < MT * nch; q += 512) {
    const int row = q / nch, c = q - row * nch;
    const int grow = row0 + row;
    u32x4 v = (u32x4){0, 0, 0, 0};
    if (c < vch && grow < M) v = *(const u32x4*)(src + (size_t)grow * ld_bytes + c * 16);
    *(u32x4*)(dst + row * rb + ((c ^ (row & km)) << 4)) = v;
  }
}

// The engine shared by both kernels: per-wave accumulators acc[TPW][MR] (16 features x 16 rows each), one
// k-step position = one B fragment per row sub-tile from LDS + TPW weight fragments from the ring.
template <class P, int TPW, int MR, int PD>
struct Engine {
  u32x4 ring[PD][TPW];
  const u32x4* stream;  // this wave's stream + lane
  int total, pos;

  __device__ __forceinline__ void prime(const u32x4* s, int total_pos, int pos0 = 0) {  // pos0 + PD <= total_pos
    stream = s;
    total = total_pos;
    pos = pos0;
#pragma unroll
    for (int p = 0; p < PD; ++p)
#pragma unroll
      for (int tp = 0; tp < TPW; ++tp) ring[p][tp] = s[((size_t)(pos0 + p) * TPW + tp) * 64];
  }
  // acc += W(layer at stream position pos .. pos+nks) . src^T ; src: swizzled LDS image with row bytes rb.
  // flags != nullptr: the image is being produced by the other waves' emits of the previous layer (wave p writes the
  // features of k-steps [p nks/8, (p+1) nks/8) and then sets flags[p] = need, see hand_over()): each group of PD k-steps
  // first makes sure its producers have arrived -- no workgroup barrier between two layers.
  __device__ __forceinline__ void run(f32x4 (&acc)[TPW][MR], const char* src, int rb, int km, int nks, int r, int g,
                                      lds_u32* flags = nullptr, uint32_t need = 0) {
    uint32_t ready = 0xffu;
    const int kpp_shift = 28 - __builtin_clz((unsigned)(nks | 8));  // log2(k-steps per producer wave): nks = 8, 16, 32, 64 with flags
    if (flags != nullptr) {
      ready = 0;
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if ((int32_t)(flag_load(flags + i) - need) >= 0) ready |= 1u << i;
      asm volatile("" ::: "memory");  // the image reads below stay behind the flag reads
    }
    for (int k0 = 0; k0 < nks; k0 += PD) {
      if (ready != 0xffu) {
        const int p_lo = k0 >> kpp_shift, p_hi = (k0 + PD - 1) >> kpp_shift;
        const uint32_t grp = ((2u << p_hi) - 1u) & ~((1u << p_lo) - 1u);
        if (grp & ~ready) {
          for (int p = p_lo; p <= p_hi; ++p)
            while ((int32_t)(flag_load(flags + p) - need) < 0) __builtin_amdgcn_s_sleep(1);
          ready |= grp;
          asm volatile("" ::: "memory");
        }
      }
#pragma unroll
      for (int p = 0; p < PD; ++p) {
        const int ks = k0 + p;
        u32x4 xb[MR];
#pragma unroll
        for (int m = 0; m < MR; ++m) xb[m] = *(const u32x4*)(src + (16 * m + r) * rb + (((ks * 4 + g) ^ (r & km)) << 4));
#pragma unroll
        for (int tp = 0; tp < TPW; ++tp)
#pragma unroll
          for (int m = 0; m < MR; ++m) acc[tp][m] = P::mma(ring[p][tp], xb[m], acc[tp][m]);
        int np = pos + ks + PD;
        np = np < total ? np : np - total;  // wrap: the next tile walks the same stream
        const u32x4* w = stream + (size_t)np * TPW * 64;
#pragma unroll
        for (int tp = 0; tp < TPW; ++tp) ring[p][tp] = w[tp * 64];
      }
    }
    pos += nks;
    if (pos >= total) pos -= total;
  }
};

// Layer hand-over by flags instead of a workgroup barrier (kernels without LayerNorm).  A wave that has written its
// features of a layer's output into the LDS image publishes the layer's sequence number in flags[wave]; the next layer's
// k-loop (Engine::run) waits per producer.  LDS executes one wave's instructions in order, so the data are in place when
// the flag is, and a reader's image loads are issued behind its flag load.  Why it is safe without a barrier on the two
// alternating images: a wave can finish layer L+1's k-loop -- and only then overwrite the image layer L read -- after it
// has seen EVERY wave's layer-L flag, which each wave sets after its own layer-L k-loop, the last reader of that image.
// What it buys: the two waves of a SIMD drift apart (the older one wins the MFMA arbitration), and one's emit phase
// (activation, stores, column sums) then runs under the other's MFMAs instead of both waiting at the barrier for the
// slower one: profiles/r01_h_fused_phase_stamps.txt shows 3.5-6.5k cycles of barrier wait per layer.
__device__ __forceinline__ void hand_over(lds_u32* flags, int wid, int lane, uint32_t seq) {
  asm volatile("" ::: "memory");  // the emit's LDS stores stay ahead of the flag store
  if (lane == 0) __hip_atomic_store(flags + wid, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  asm volatile("" ::: "memory");
}

// The one-block kernels' walk of a weight stream, without the padding of its short layers.  A stream position is one
// k-step of one layer, and the ring wants every layer to be a multiple of its depth long (slot = k-step mod depth is then a
// compile-time register index): the layers on the input tile (K = in_dim) and on the d_out tile (K = out_dim) are padded
// to 4 k-steps of which 1-2 carry data -- 2 x 3 of the backward's 24 positions per tile, 2 of the merged forward's 20,
// streamed and multiplied for nothing in kernels whose k-loops are paced by exactly that stream.  Here a tile is a FIXED
// sequence of S1 + KSH + S2 real positions (short layer, H-wide layer, short layer) padded with holes -- positions that
// are never loaded -- to a multiple of the depth, so that slots stay compile-time and every tile starts at slot 0.
// AUX: the S2 positions of the third segment do not come from the wave's stream but from a second fragment array: position j
// is the TPW fragments aux[(j TPW + tp) aux_stride], tp = 0 .. TPW-1 (the merged forward's wide out layer: a wave's K slice of
// one out tile of Wout W2, whose fragments lie OT x 64 apart in the out stream [k-step][out tile][lane]).
// Which segment a refill fetches from is known at COMPILE time everywhere but in the middle of the H-wide layer, where it is
// always that layer itself: the walk is written with compile-time compact positions (refill_ct) around a loop that only ever
// refills from the wide segment (refill_mid) -- no per-step segment selects, holes are not loaded at all, and the AUX
// segment's runtime-stride addresses exist only at the S2 places that use them (selecting per step cost 14-20 VGPRs in the
// k-loop: 18-26 spilled in the wide-head kernels).
template <class P, int TPW, int MR, int S1, int KSH, int S2, bool AUX = false>
struct CEngine {
  static constexpr int PD = 4, T = S1 + KSH + S2, HOLES = (PD - T % PD) % PD, TT = T + HOLES;
  static_assert(KSH % PD == 0 && KSH >= 2 * PD && S1 >= 1 && S1 <= PD && S2 >= 0 && S2 <= PD && T >= PD, "CEngine layout");
  u32x4 ring[PD][TPW];
  // Addresses are (wave-uniform base) + (lane index): the bases stay in SGPRs and ONE 32-bit VGPR serves every load.  As
  // per-lane 64-bit pointers the next tile's first positions were hoisted out of the persistent tile loop and spilled, and
  // each scratch reload in the last steps of the H-wide k-loop came with an s_waitcnt vmcnt(0) -- the ring drained four
  // times per tile.
  const u32x4* stream;  // this wave's stream (NO lane offset)
  const u32x4* aux;     // AUX: this wave's first third-segment fragment (NO lane offset)
  int aux_stride;       // AUX: distance (in u32x4 units) between two consecutive ones
  unsigned ln;          // the lane index
  int p1, pl, p2;       // stream positions where the three segments start

  // fetch compact position N (a compile-time constant, already wrapped into [0, TT)) into slot SLOT
  template <int SLOT, int N>
  __device__ __forceinline__ void fetch_ct() {
    unsigned ln = this->ln;
    asm volatile("" : "+v"(ln));  // the address is formed HERE, once per tile and use: nothing to hoist and spill
    if constexpr (N < S1 + KSH || (N < T && !(AUX && S2 > 0))) {
      const int q = N < S1 ? p1 + N : (N < S1 + KSH ? pl + (N - S1) : p2 + (N - S1 - KSH));
      const u32x4* w = stream + (size_t)q * TPW * 64;
#pragma unroll
      for (int tp = 0; tp < TPW; ++tp) ring[SLOT][tp] = (w + tp * 64)[ln];
    } else if constexpr (N < T) {  // AUX third segment
      const u32x4* w = aux + (size_t)((N - S1 - KSH) * TPW) * aux_stride;
#pragma unroll
      for (int tp = 0; tp < TPW; ++tp) ring[SLOT][tp] = (w + (size_t)tp * aux_stride)[ln];
    }  // else: a hole -- nothing to fetch, the slot is not multiplied before its next refill
  }
  template <int SLOT, int C>
  __device__ __forceinline__ void refill_ct() {  // the slot gave up compact position C: fetch C + PD (of the next tile, maybe)
    fetch_ct<SLOT, (C + PD >= TT ? C + PD - TT : C + PD)>();
  }
  template <int SLOT>
  __device__ __forceinline__ void refill_mid(int c) {  // c + PD lies inside the wide segment
    unsigned ln = this->ln;
    asm volatile("" : "+v"(ln));  // (as in fetch_ct: the unrolled k-loop's addresses were hoisted out of the tile loop as twelve 64-bit pairs)
    const u32x4* w = stream + (size_t)(pl + (c + PD - S1)) * TPW * 64;
#pragma unroll
    for (int tp = 0; tp < TPW; ++tp) ring[SLOT][tp] = (w + tp * 64)[ln];
  }
  __device__ __forceinline__ void set_aux(const u32x4* a, int stride) { aux = a, aux_stride = stride; }  // before prime()
  // s: the wave's stream WITHOUT the lane offset (wave-uniform)
  __device__ __forceinline__ void prime(const u32x4* s, int lane, int p1_, int pl_, int p2_) {
    stream = s, ln = (unsigned)lane, p1 = p1_, pl = pl_, p2 = p2_;
    fetch_ct<0, 0>();
    fetch_ct<1, 1>();
    fetch_ct<2, 2>();
    fetch_ct<3, 3>();
  }
  __device__ __forceinline__ void mult(int SLOTv, f32x4 (&acc)[TPW][MR], const u32x4 (&fr)[TPW], const char* src, int rb, int km, int ks,
                                       int r, int g) {
    (void)SLOTv;
    u32x4 xb[MR];
#pragma unroll
    for (int m = 0; m < MR; ++m) xb[m] = *(const u32x4*)(src + (16 * m + r) * rb + (((ks * 4 + g) ^ (r & km)) << 4));
#pragma unroll
    for (int tp = 0; tp < TPW; ++tp)
#pragma unroll
      for (int m = 0; m < MR; ++m) acc[tp][m] = P::mma(fr[tp], xb[m], acc[tp][m]);
  }
  template <int SLOT, int C>
  __device__ __forceinline__ void step_ct(f32x4 (&acc)[TPW][MR], const char* src, int rb, int km, int ks, int r, int g) {
    mult(SLOT, acc, ring[SLOT], src, rb, km, ks, r, g);
    refill_ct<SLOT, C>();
  }
  template <int SLOT>
  __device__ __forceinline__ void step_mid(f32x4 (&acc)[TPW][MR], const char* src, int rb, int km, int ks, int c, int r, int g) {
    mult(SLOT, acc, ring[SLOT], src, rb, km, ks, r, g);
    refill_mid<SLOT>(c);
  }
  // acc += (first short layer) . src^T : k-steps 0 .. S1-1 of src
  __device__ __forceinline__ void short1(f32x4 (&acc)[TPW][MR], const char* src, int rb, int km, int r, int g) {
    // (a compiler barrier for memory operations between the steps: hoisting every step's B-fragment reads to the top keeps
    // S1 x MR x 4 registers live beside the accumulators and the ring -- S1 = 2 / 3 spilled 20-26 VGPRs in the wide-head kernels)
    if constexpr (S1 >= 1) step_ct<0, 0>(acc, src, rb, km, 0, r, g);
    if constexpr (S1 >= 2) {
      asm volatile("" ::: "memory");
      step_ct<1, 1>(acc, src, rb, km, 1, r, g);
    }
    if constexpr (S1 >= 3) {
      asm volatile("" ::: "memory");
      step_ct<2, 2>(acc, src, rb, km, 2, r, g);
    }
    if constexpr (S1 >= 4) {
      asm volatile("" ::: "memory");
      step_ct<3, 3>(acc, src, rb, km, 3, r, g);
    }
  }
  // acc += (H-wide layer) . src^T, with the flag waits of Engine::run (KSH / 8 k-steps per producer wave)
  // TAIL_REFILL = false: the last four steps multiply without fetching what follows the wide segment -- the ring's registers are
  // dead from here to reprime() at the tile's end (the fused-loss forward: its epilogue needs them)
  template <bool TAIL_REFILL = true>
  __device__ __forceinline__ void wide(f32x4 (&acc)[TPW][MR], const char* src, int rb, int km, int r, int g,
                                       lds_u32* flags = nullptr, uint32_t need = 0) {
    uint32_t ready = 0xffu;
    constexpr int kpp_shift = KSH >= 64 ? 3 : (KSH >= 32 ? 2 : (KSH >= 16 ? 1 : 0));
    if (flags != nullptr) {
      ready = 0;
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if ((int32_t)(flag_load(flags + i) - need) >= 0) ready |= 1u << i;
      asm volatile("" ::: "memory");
    }
    auto wait_group = [&](int k0) {
      if (ready != 0xffu) {
        const int p_lo = k0 >> kpp_shift, p_hi = (k0 + PD - 1) >> kpp_shift;
        const uint32_t grp = ((2u << p_hi) - 1u) & ~((1u << p_lo) - 1u);
        if (grp & ~ready) {
          for (int p = p_lo; p <= p_hi; ++p)
            while ((int32_t)(flag_load(flags + p) - need) < 0) __builtin_amdgcn_s_sleep(1);
          ready |= grp;
          asm volatile("" ::: "memory");
        }
      }
    };
    for (int k0 = 0; k0 < KSH - PD; k0 += PD) {  // every refill of these steps stays inside the wide segment
      wait_group(k0);
      step_mid<(S1 + 0) % PD>(acc, src, rb, km, k0 + 0, S1 + k0 + 0, r, g);
      step_mid<(S1 + 1) % PD>(acc, src, rb, km, k0 + 1, S1 + k0 + 1, r, g);
      step_mid<(S1 + 2) % PD>(acc, src, rb, km, k0 + 2, S1 + k0 + 2, r, g);
      step_mid<(S1 + 3) % PD>(acc, src, rb, km, k0 + 3, S1 + k0 + 3, r, g);
    }
    constexpr int KL = KSH - PD;  // the last four: their refills reach the third segment, the holes or the next tile
    wait_group(KL);
    if constexpr (TAIL_REFILL) {
      step_ct<(S1 + 0) % PD, S1 + KL + 0>(acc, src, rb, km, KL + 0, r, g);
      step_ct<(S1 + 1) % PD, S1 + KL + 1>(acc, src, rb, km, KL + 1, r, g);
      step_ct<(S1 + 2) % PD, S1 + KL + 2>(acc, src, rb, km, KL + 2, r, g);
      step_ct<(S1 + 3) % PD, S1 + KL + 3>(acc, src, rb, km, KL + 3, r, g);
    } else {
      static_assert(TAIL_REFILL || S2 == 0, "without a third segment only");
      mult((S1 + 0) % PD, acc, ring[(S1 + 0) % PD], src, rb, km, KL + 0, r, g);
      mult((S1 + 1) % PD, acc, ring[(S1 + 1) % PD], src, rb, km, KL + 1, r, g);
      mult((S1 + 2) % PD, acc, ring[(S1 + 2) % PD], src, rb, km, KL + 2, r, g);
      mult((S1 + 3) % PD, acc, ring[(S1 + 3) % PD], src, rb, km, KL + 3, r, g);
    }
  }
  __device__ __forceinline__ void reprime() {  // the next tile's first four positions (a tile starts at slot 0)
    fetch_ct<0, 0>();
    fetch_ct<1, 1>();
    fetch_ct<2, 2>();
    fetch_ct<3, 3>();
  }
  // acc += (last short layer) . src^T : k-steps 0 .. S2-1 of src
  __device__ __forceinline__ void short2(f32x4 (&acc)[TPW][MR], const char* src, int rb, int km, int r, int g) {
    if constexpr (S2 >= 1) step_ct<(S1 + 0) % PD, S1 + KSH + 0>(acc, src, rb, km, 0, r, g);
    if constexpr (S2 >= 2) step_ct<(S1 + 1) % PD, S1 + KSH + 1>(acc, src, rb, km, 1, r, g);
    if constexpr (S2 >= 3) step_ct<(S1 + 2) % PD, S1 + KSH + 2>(acc, src, rb, km, 2, r, g);
    if constexpr (S2 >= 4) step_ct<(S1 + 3) % PD, S1 + KSH + 3>(acc, src, rb, km, 3, r, g);
  }
  // AUX third segment, position j: o[m] += (fragment tp of the position) . (k-step ks0 + j TPW + tp of src)^T -- one out tile,
  // TPW consecutive k-steps of this wave's K slice per position
  template <int J>
  __device__ __forceinline__ void aux_step(f32x4 (&o)[MR], const char* src, int rb, int km, int ks0, int r, int g) {
    constexpr int SLOT = (S1 + J) % PD;
#pragma unroll
    for (int tp = 0; tp < TPW; ++tp) {
      const int ks = ks0 + J * TPW + tp;
#pragma unroll
      for (int m = 0; m < MR; ++m) {
        const u32x4 xb = *(const u32x4*)(src + (16 * m + r) * rb + (((ks * 4 + g) ^ (r & km)) << 4));
        o[m] = P::mma(ring[SLOT][tp], xb, o[m]);
      }
    }
    refill_ct<SLOT, S1 + KSH + J>();
  }
  __device__ __forceinline__ void aux_all(f32x4 (&o)[MR], const char* src, int rb, int km, int ks0, int r, int g) {
    if constexpr (S2 >= 1) aux_step<0>(o, src, rb, km, ks0, r, g);
    if constexpr (S2 >= 2) aux_step<1>(o, src, rb, km, ks0, r, g);
    if constexpr (S2 >= 3) aux_step<2>(o, src, rb, km, ks0, r, g);
    if constexpr (S2 >= 4) aux_step<3>(o, src, rb, km, ks0, r, g);
  }
  __device__ __forceinline__ void end_tile() {  // the holes' slots take the next tile's positions
    if constexpr (HOLES >= 1) refill_ct<(T + 0) % PD, T + 0>();
    if constexpr (HOLES >= 2) refill_ct<(T + 1) % PD, T + 1>();
    if constexpr (HOLES >= 3) refill_ct<(T + 2) % PD, T + 2>();
  }
};

// ---- K-major fragment stores (gemm.h, GemmTNFrag): bf16 only -------------------------------------------------------------
// A 16-lane group reading a 4-row x 16-column block of bf16 with ds_read_b64_tr_b16 gets it transposed: lane i ends with
// column i of the 4 rows.  Two reads (rows 8 kg + 0..3 and + 4..7 of a 32-row k-step) are one lane's 16 bytes of the
// fragment: feature 16 ft + i, rows 32 ks + 8 kg + s.
typedef __attribute__((address_space(3))) i16x4 lds_i16x4;
__device__ __forceinline__ u32x2 lds_tr16(const char* p) {
  return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_i16x4*)p));
}
// this wave's 16 TPW features of a swizzled [16 MR][H] bf16 LDS image (as written by emit(): 16-byte chunk c of row R sits
// at chunk c ^ (R & 15)) -> out[ks'][ft][lane], ks' = 0 .. MR/2 - 1 the tile's k-steps, NT feature tiles per k-step.
// A wave reads back only what it wrote itself (LDS executes a wave's instructions in order): no barrier.
template <int TPW, int MR>
__device__ __forceinline__ void frag_store_image(const char* img, int HRB, int wid, int lane, u32x4* out, int NT) {
  static_assert(MR % 2 == 0, "a k-step is two row sub-tiles");
  const int i = lane & 15, kg = lane >> 4, q = i >> 2, p = i & 3;
#pragma unroll
  for (int ks = 0; ks < MR / 2; ++ks) {
    const int row0 = 32 * ks + 8 * kg + q, row1 = row0 + 4;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const int ft = wid * TPW + t, c = 2 * ft + (p >> 1);
      const u32x2 u0 = lds_tr16(img + row0 * HRB + ((c ^ (row0 & 15)) << 4) + (p & 1) * 8);
      const u32x2 u1 = lds_tr16(img + row1 * HRB + ((c ^ (row1 & 15)) << 4) + (p & 1) * 8);
      out[((size_t)ks * NT + ft) * 64 + lane] = (u32x4){u0.x, u0.y, u1.x, u1.y};
    }
  }
}
// the same for a narrow tile every wave can read (the d_out tile, the input rows: [16 MR][rb bytes], chunk swizzle by
// row & km, load_tile()): the (k-step, feature tile) pairs are dealt to the eight waves; behind the barrier that made the
// tile visible.  nt_src: feature tiles to copy (<= rb / 32), NT: tiles per k-step of the destination
template <int MR>
__device__ __forceinline__ void frag_store_tile(const char* img, int rb, int km, int nt_src, int wid, int lane, u32x4* out, int NT) {
  const int i = lane & 15, kg = lane >> 4, q = i >> 2, p = i & 3;
  for (int item = wid; item < (MR / 2) * nt_src; item += SAMPLER_WAVES) {
    const int ks = item / nt_src, ft = item - ks * nt_src;
    const int row0 = 32 * ks + 8 * kg + q, row1 = row0 + 4, c = 2 * ft + (p >> 1);
    const u32x2 u0 = lds_tr16(img + row0 * rb + ((c ^ (row0 & km)) << 4) + (p & 1) * 8);
    const u32x2 u1 = lds_tr16(img + row1 * rb + ((c ^ (row1 & km)) << 4) + (p & 1) * 8);
    out[((size_t)ks * NT + ft) * 64 + lane] = (u32x4){u0.x, u0.y, u1.x, u1.y};
  }
}
// accumulators v[tp][m] (features wbase + feat_off(g, tp) + e of rows 16 m + r) that go to NO LDS image (dh_0 of the one-block
// backward): staged k-step by k-step through a private [32 rows][16 TPW features] bf16 region of this wave, chunk c of row
// R at c ^ (R & (2 TPW - 1)), and read back transposed.  stage: this wave's 32 * 32 * TPW bytes.
template <int TPW, int MR>
__device__ __forceinline__ void frag_store_acc(const f32x4 (&v)[TPW][MR], char* stage, int wid, int lane, u32x4* out, int NT) {
  static_assert(MR % 2 == 0 && TPW % 2 == 0, "bf16 chunk = two MFMA tiles; a k-step = two row sub-tiles");
  constexpr int RS = 32 * TPW, CM = 2 * TPW - 1;  // row bytes, chunk mask
  const int r = lane & 15, g = lane >> 4;
  const int i = r, kg = g, q = i >> 2, p = i & 3;
#pragma unroll
  for (int ks = 0; ks < MR / 2; ++ks) {
#pragma unroll
    for (int mm = 0; mm < 2; ++mm) {
      const int m = 2 * ks + mm, row = 16 * mm + r;
#pragma unroll
      for (int tp = 0; tp < TPW; tp += 2) {
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float lo = v[tp + (2 * e) / 4][m][(2 * e) % 4], hi = v[tp + (2 * e + 1) / 4][m][(2 * e + 1) % 4];
          o[e] = (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
        }
        const int c = (feat_off<BF16>(g, tp) * 2) >> 4;  // chunk of the wave's slice: 4 (tp / 2) + g
        *(u32x4*)(stage + row * RS + ((c ^ (row & CM)) << 4)) = o;
      }
    }
    const int row0 = 8 * kg + q, row1 = row0 + 4;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const int c = 2 * t + (p >> 1);
      const u32x2 u0 = lds_tr16(stage + row0 * RS + ((c ^ (row0 & CM)) << 4) + (p & 1) * 8);
      const u32x2 u1 = lds_tr16(stage + row1 * RS + ((c ^ (row1 & CM)) << 4) + (p & 1) * 8);
      out[((size_t)ks * NT + wid * TPW + t) * 64 + lane] = (u32x4){u0.x, u0.y, u1.x, u1.y};
    }
  }
}

// In-kernel first-layer weight gradient (FusedBwdArgs::dw0_slab): C[feature][col] += sum over the tile's rows of
// bf16(v[row][feature]) * x[row][col'] for this wave's 16 TPW features and 32 columns col of the input rows.  The accumulators v
// (feature 4g + e of MFMA tile tp at row 16 m + r: the contraction index sits in the LANE) are staged pair of tiles by pair,
// k-step by k-step, through a private [32 rows][32 features] bf16 region (chunk c of row R at c ^ (R & 3)) and read back
// transposed as MFMA A fragments (frag_store_acc's path); the B fragments come from the input-row tile `xw` ([16 MR][128 bytes],
// chunk swizzle by row & 7, load_tile()) the same way, four columns per lane address: column block b of the 32 is the row's
// columns 4 b .. 4 b + 3, moved up by `skip` columns from column `af` on (a denoiser's rows are [x_k | temb | obs | one-hot]: the
// time-embedding columns are skipped; af and skip are multiples of 4).  C lives in LDS between tiles -- cacc[(tile t of the wave's
// 16 TPW features, column tile)][lane] in MFMA D layout, this wave's own region -- because the kernel has no register left to
// carry it; a tile of it is read and written once per 64 rows.  round_back: v is left holding the rounded values (the caller's
// column sums then add up what the product saw; a template parameter: as a run-time flag it cost 30 registers).
// stage: this wave's 2 KB; cacc: this wave's [TPW][2][64] f32x4.
template <int TPW, int MR, bool round_back>
__device__ __forceinline__ void dw0_accumulate(f32x4 (&v)[TPW][MR], char* stage, const char* xw, f32x4* cacc, int lane, int af,
                                               int skip) {
  static_assert(MR % 4 == 0 && TPW % 2 == 0, "bf16 chunk = two MFMA tiles; two k-steps of two row sub-tiles at a time");
  // Two k-steps (64 rows) at a time, one pair of feature tiles at a time, with compiler barriers for memory operations between
  // them: what is live beside the accumulators and the weight ring stays at 2 x 2 A fragments, 2 x 2 B fragments and one C tile.
  // (Everything hoisted to the top -- the compiler's choice -- is 64 more registers at MR = 8: the kernel then fills the register
  // file and no other stream's wave fits a SIMD beside its two, which costs the overlapped update step more than the stores saved.)
#pragma unroll
  for (int kc = 0; kc < MR / 2; kc += 2) {
    asm volatile("" : "+v"(lane)::"memory");  // (addresses formed here, once per use: see emit())
    const int r = lane & 15, g = lane >> 4, q = r >> 2, p = r & 3;
    u32x4 xb[2][2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int col = 16 * nt + 4 * p, sb = 2 * (col + (col >= af ? skip : 0));  // byte of the lane's four columns in a row
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int row0 = 32 * (kc + ks) + 8 * g + q, row1 = row0 + 4;
        const u32x2 u0 = lds_tr16(xw + row0 * 128 + (((sb >> 4) ^ (row0 & 7)) << 4) + (sb & 15));
        const u32x2 u1 = lds_tr16(xw + row1 * 128 + (((sb >> 4) ^ (row1 & 7)) << 4) + (sb & 15));
        xb[ks][nt] = (u32x4){u0.x, u0.y, u1.x, u1.y};
      }
    }
#pragma unroll
    for (int tp = 0; tp < TPW; tp += 2) {
      u32x4 fa[2][2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int mm = 0; mm < 2; ++mm) {
          const int m = 2 * (kc + ks) + mm, row = 16 * mm + r;
          u32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float lo = v[tp + (2 * e) / 4][m][(2 * e) % 4], hi = v[tp + (2 * e + 1) / 4][m][(2 * e + 1) % 4];
            o[e] = (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
            if constexpr (round_back) {
              v[tp + (2 * e) / 4][m][(2 * e) % 4] = __uint_as_float(o[e] << 16);
              v[tp + (2 * e + 1) / 4][m][(2 * e + 1) % 4] = __uint_as_float(o[e] & 0xffff0000u);
            }
          }
          *(u32x4*)(stage + row * 64 + ((g ^ (row & 3)) << 4)) = o;  // features 8 g .. 8 g + 7 of the pair's 32
        }
        const int sr0 = 8 * g + q, sr1 = sr0 + 4;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int c = 2 * t + (p >> 1);
          const u32x2 u0 = lds_tr16(stage + sr0 * 64 + ((c ^ (sr0 & 3)) << 4) + (p & 1) * 8);
          const u32x2 u1 = lds_tr16(stage + sr1 * 64 + ((c ^ (sr1 & 3)) << 4) + (p & 1) * 8);
          fa[ks][t] = (u32x4){u0.x, u0.y, u1.x, u1.y};
        }
        asm volatile("" ::: "memory");  // (the next k-step's staging stores stay behind this one's reads)
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          f32x4* cp = cacc + ((tp + t) * 2 + nt) * 64 + lane;
          f32x4 c = *cp;
          c = BF16::mma(fa[0][t], xb[0][nt], c);
          c = BF16::mma(fa[1][t], xb[1][nt], c);
          *cp = c;
        }
        asm volatile("" ::: "memory");
      }
    }
  }
}

// ReLU'(x) of a lane's 4*TPW features of one row is a bit mask: the forward stores it as one 32-bit word per (row, wave,
// lane group g) -- [M][SIGN_WORDS] -- and the backward reads 128 bytes per row instead of the whole activated tensor.
constexpr int SIGN_WORDS = 32;  // 8 waves x 4 lane groups

// v[tp][m] (features fb + 4tp + e of row 16m + r) -> activated elem chunks, written to the LDS image `lds`
// (may be null) and to the global [M][H] tensor `glb` (may be null).  `dglb` (may be null): the derivative source for the
// backward -- Mish: a second [M][H] tensor that receives act'(v), so the backward multiplies by it instead of recomputing the
// derivative from the pre-activation (value and derivative share one exp and one reciprocal here; a separate Mish' pass
// made the 128-row critic tile VALU-bound); ReLU: the [M][SIGN_WORDS] sign-bit words.
template <class P, int TPW, int MR>
__device__ __forceinline__ void emit(const f32x4 (&v)[TPW][MR], int actk, char* lds, void* glb, int H, int wbase, int g,
                                     int r, int row0, int M, void* dglb = nullptr) {
  constexpr int ES = P::ESIZE;
  const int HRB = H * ES;
  // keep this phase's addresses from being hoisted out of the persistent tile loop and spilled: a scratch reload waits
  // (vmcnt is in issue order) for every weight fragment the ring has in flight
  asm volatile("" : "+v"(r), "+v"(g));
  with_act(actk, [&](auto tag) {
    constexpr int ACT = decltype(tag)::value;
    auto body = [&](auto grad_tag) {  // resolved once per call, never per element
    constexpr bool with_grad = decltype(grad_tag)::value;
    constexpr bool with_mask = with_grad && ACT == ACT_RELU;  // ReLU: the derivative is one bit, see sign_mask_at()
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      const int grow = row0 + 16 * m + r;
      const bool live = grow < M;
      char* gp = glb != nullptr && live ? (char*)glb + (size_t)grow * HRB : nullptr;
      char* dp = with_grad && !with_mask && live ? (char*)dglb + (size_t)grow * HRB : nullptr;
      uint32_t bits = 0;
      char* lp = lds != nullptr ? lds + (16 * m + r) * HRB : nullptr;
      constexpr int TPC = ES == 4 ? 1 : 2;  // MFMA tiles per 16-byte chunk
#pragma unroll
      for (int tp = 0; tp < TPW; tp += TPC) {
        float av[4 * TPC], dv[4 * TPC];
        if constexpr (ACT == ACT_MISH) {  // pairs on the packed-fp32 VALU; without the derivative its half is dead code
#pragma unroll
          for (int q = 0; q < 4 * TPC; q += 2) {
            const float2v x = {v[tp + q / 4][m][q % 4], v[tp + (q + 1) / 4][m][(q + 1) % 4]};
            float2v a2, d2;
            mish_both2(x, a2, d2);
            av[q] = a2.x, av[q + 1] = a2.y;
            dv[q] = with_grad ? d2.x : 0.f, dv[q + 1] = with_grad ? d2.y : 0.f;
          }
        } else {
#pragma unroll
          for (int q = 0; q < 4 * TPC; ++q) {
            const float x = v[tp + q / 4][m][q % 4];
            if constexpr (with_mask) bits |= (x > 0.f ? 1u : 0u) << (4 * tp + q);
            av[q] = act_c<ACT>(x);
            dv[q] = 0.f;
          }
        }
        const int c = ((wbase + feat_off<P>(g, tp)) * ES) >> 4;
        u32x4 o, od;
        if constexpr (ES == 4) {
#pragma unroll
          for (int q = 0; q < 4; ++q) o[q] = __float_as_uint(av[q]), od[q] = __float_as_uint(dv[q]);
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            o[q] = (uint32_t)f2bf(av[2 * q]) | ((uint32_t)f2bf(av[2 * q + 1]) << 16);
            od[q] = (uint32_t)f2bf(dv[2 * q]) | ((uint32_t)f2bf(dv[2 * q + 1]) << 16);
          }
        }
        if (lp) *(u32x4*)(lp + ((c ^ (r & 15)) << 4)) = o;
        if (gp) *(u32x4*)(gp + c * 16) = o;
        if (dp) *(u32x4*)(dp + c * 16) = od;
        // one chunk at a time: interleaving the exp / rcp chains of all 16 chunks keeps their temporaries live at once,
        // on top of the accumulators, the residual stream and the weight ring
        if constexpr (ACT == ACT_MISH) __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (with_mask) {
        if (live) ((uint32_t*)dglb)[(size_t)grow * SIGN_WORDS + (wbase / (16 * TPW)) * 4 + g] = bits;
      }
    }
    };
    if ((ACT == ACT_MISH || ACT == ACT_RELU) && dglb != nullptr)
      body(std::true_type{});
    else
      body(std::false_type{});
  });
}

// this lane's 4*TPW values of row (16m + r) of a global [M][H] elem tensor, kept packed as loaded
// (CH 16-byte chunks per row sub-tile); grad_at() decodes one element and returns act'(.)
template <class P, int TPW>
struct Chunks {
  static constexpr int CH = P::ESIZE == 4 ? TPW : TPW / 2;
};
template <class P, bool MASK, int TPW, int MR, int CH>
__device__ __forceinline__ void fetch(u32x4 (&d)[MR][CH], const void* glb, int H, int wbase, int g, int r, int row0,
                                      int M) {
  constexpr int ES = P::ESIZE, TPC = ES == 4 ? 1 : 2;  // MFMA tiles per 16-byte chunk
  asm volatile("" : "+v"(r), "+v"(g));  // see emit()
  if constexpr (MASK) {  // ReLU sign words written by emit()
    if (glb != nullptr) {
#pragma unroll
      for (int m = 0; m < MR; ++m) {
        const int grow = row0 + 16 * m + r;
        d[m][0][0] = ((const uint32_t*)glb)[(size_t)(grow < M ? grow : M - 1) * SIGN_WORDS + (wbase / (16 * TPW)) * 4 + g];
      }
      return;
    }
  }
  if (glb == nullptr) {  // (timing experiments only)
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
      for (int c = 0; c < CH; ++c) d[m][c] = (u32x4){0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    return;
  }
#pragma unroll
  for (int m = 0; m < MR; ++m) {
    const int grow = row0 + 16 * m + r;
    const char* gp = (const char*)glb + (size_t)(grow < M ? grow : M - 1) * H * ES;
#pragma unroll
    for (int c = 0; c < CH; ++c) d[m][c] = *(const u32x4*)(gp + (size_t)(wbase + feat_off<P>(g, c * TPC)) * ES);
  }
}
// ReLU: d[m][0][0] holds the lane's sign bits; Mish: d holds act'(x) itself; both stored by the forward's emit()
template <class P, int ACT, int MR, int CH>
__device__ __forceinline__ float grad_at(const u32x4 (&d)[MR][CH], int tp, int m, int e) {
  if constexpr (ACT == ACT_RELU) return (d[m][0][0] >> (4 * tp + e)) & 1u ? 1.f : 0.f;
  float x;
  if constexpr (P::ESIZE == 4) {
    x = __uint_as_float(d[m][tp][e]);
  } else {
    const uint32_t w = d[m][tp >> 1][(tp & 1) * 2 + (e >> 1)];
    x = bf2f((e & 1) ? (w >> 16) : (w & 0xffff));
  }
  if constexpr (ACT == ACT_MISH) return x;
  return act_grad_c<ACT>(x);
}

}  // namespace

// =================================================================================================
// forward
// =================================================================================================
template <class P, int TPW, int MR, int OT, bool LN, int ACT, int OCC>
__global__ __launch_bounds__(512, 2 * OCC) void fused_forward_kernel(const FusedFwdArgs a) {
  // (HIP: the second launch bound is waves per SIMD: 2 per workgroup of 8 waves)
  // OCC = 2: two workgroups per CU (128 VGPRs each): a shallower ring, the other workgroup's MFMAs fill this one's
  // emit / barrier phases
  constexpr int PD = OCC > 1 ? 2 : ring_depth<TPW, MR>(), ES = P::ESIZE, KB = P::KB;
  constexpr int H = 128 * TPW, KSH = H / KB, HRB = H * ES, MT = 16 * MR;
  constexpr int KSPLIT = MR * OT >= 8 ? 1 : 8 / (MR * OT);
  constexpr int KPER = KSH / KSPLIT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  // layer hand-over flags, one per wave (see hand_over()); LayerNorm needs the whole workgroup at every layer anyway
  constexpr bool FLAGS = !LN && DPPO_FLAGS;
#ifndef DPPO_NO_SETPRIO
  // Barrier variants only: static priority for the younger half of the workgroup (MI355X_MICROARCH.md, two waves per SIMD,
  // item 4): waves 4-7 lose every VALU arbitration against their SIMD partners and reach each layer's barrier 3.5-6.5k
  // cycles late (profiles/r01_h_fused_phase_stamps.txt); measured +0.8 % on the update step (profiles/r02_i_setprio_ab.txt).
  // With the flag hand-over the lag is useful (the older wave's emit runs under the younger's MFMAs) and equal priorities
  // measure best (profiles/r02_flags_ab.txt).  The same line in the sampler costs 7 % there (one 16-row tile per workgroup:
  // the older half is the critical path).
  if (!FLAGS && wid >= 4) __builtin_amdgcn_s_setprio(1);
#endif
  const int r = lane & 15, g = lane >> 4;
  const int Kp0 = a.Kp0, nb = a.nb, M = a.M;
  const int in_rb = Kp0 * ES, in_km = kmask16(in_rb), KS0 = Kp0 / KB;
  const int total = KS0 + 2 * nb * KSH;
  char* bufA = smem;
  char* bufB = bufA + MT * HRB;
  char* xin = bufB;  // the input tile is dead once layer 0 has run, before the first block writes bufB
  // out-layer partials [KSPLIT][MR][OT*16 features][16 rows]: the out layer reads bufA only, so they live in bufB
  // (KSPLIT*OT <= 8 sub-tiles of 1 KB per 16 rows <= 16*H*ES always)
  float* part = (float*)bufB;
  static_assert(KSPLIT * OT * 16 * 16 * 4 <= 16 * HRB, "out-layer partials must fit in buffer B");
  lds_u32* flags = (lds_u32*)(bufB + MT * HRB);              // [16 words]
  float* lnred = (float*)(bufB + MT * HRB + 64);  // [8 waves][MR][16] LayerNorm row-reduction table (LN only)
  // Constants of the whole launch, staged once per workgroup where LDS allows (the launcher decides, a.consts_lds):
  // a global load issued at a layer's start waits behind every weight fragment the ring has in flight (vmcnt is in
  // issue order) -- measured ~2k cycles per layer for the bias and ~6k for the out-layer fragments of a 48k-cycle tile.
  float* biasL = lnred + (LN ? LN_WAVES * MR * 16 : 0);  // [1 + 2 nb][H] hidden-layer biases, then [OT*16] out bias
  u32x4* woutL = (u32x4*)(biasL + (((1 + 2 * a.nb) * H + OT * 16 + 3) & ~3));  // [KSH][OT][64 lanes] out-layer fragments
  const int wbase = wid * 16 * TPW;  // lane features: wbase + feat_off<P>(g, tp) + e
  const bool bias_lds = (a.consts_lds & 1) != 0, wout_lds = (a.consts_lds & 2) != 0;
  if (bias_lds) {
    for (int idx = tid; idx < (1 + 2 * a.nb) * H; idx += 512) biasL[idx] = a.params[a.bias_off[idx / H] + idx % H];
    for (int idx = tid; idx < OT * 16; idx += 512)
      biasL[(1 + 2 * a.nb) * H + idx] = idx < a.out_dim ? a.params[a.bias_off[1 + 2 * a.nb] + idx] : 0.f;
  }
  if (wout_lds)
    for (int idx = tid; idx < KSH * OT * 64; idx += 512) woutL[idx] = a.ostream[idx];
  if (tid < 16) flags[tid] = 0;
  uint32_t seq = 0;  // the same in every wave: hand-overs so far
  // (visible after the first tile's barrier)

  Engine<P, TPW, MR, PD> eng;
  eng.prime(a.wstream + (size_t)wid * total * TPW * 64 + lane, total);

  const int ntiles = (M + MT - 1) / MT;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int row0 = tile * MT;
    STAMP(0);
    load_tile<MT>(xin, in_rb, in_km, (const char*)a.in, a.ld_in * ES, row0, M);
    __syncthreads();
    STAMP(1);

    f32x4 h[TPW][MR], acc[TPW][MR];
    auto bias_init = [&](int layer) {  // layer: 0, then 1 + 2b / 2 + 2b
      // (two pointers, never one selected at run time: a pointer that may be LDS or global is a GENERIC pointer, its loads
      // are FLAT, and a FLAT load is followed by s_waitcnt vmcnt(0) lgkmcnt(0) -- the weight ring drained at every layer)
      const float* srcL = biasL + layer * H;
      const float* srcG = a.params + a.bias_off[layer];
#pragma unroll
      for (int tp = 0; tp < TPW; ++tp) {
        f32x4 b;
        if (bias_lds) {
          b = lds_load((const f32x4*)(srcL + wbase + feat_off<P>(g, tp)));
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) b[e] = glb_load(srcG + wbase + feat_off<P>(g, tp) + e);
        }
#pragma unroll
        for (int m = 0; m < MR; ++m) acc[tp][m] = b;
      }
    };
    // block input: bufA <- act([LN1_b] h), with the tensors the backward needs (training only)
    auto put_block_input = [&](int b) {
      if constexpr (LN) {
        if (a.hpre[b] != nullptr) emit<P, TPW, MR>(h, ACT_NONE, nullptr, a.hpre[b], H, wbase, g, r, row0, M);
        float mean[MR], rstd[MR];
        ln_forward<P, TPW, MR>(h, acc, a.params + a.ln_off[4 * b], a.params + a.ln_off[4 * b + 1], H, wbase, g, r, wid,
                               lnred, mean, rstd);
        if (a.ln_stats != nullptr && wid == 0 && g == 0) {
#pragma unroll
          for (int m = 0; m < MR; ++m)
            if (row0 + 16 * m + r < M)
              *(float2*)(a.ln_stats + (((size_t)(2 * b) * M) + row0 + 16 * m + r) * 2) = make_float2(mean[m], rstd[m]);
        }
        emit<P, TPW, MR>(acc, ACT, bufA, a.a1[b], H, wbase, g, r, row0, M);
      } else {
        emit<P, TPW, MR>(h, ACT, bufA, a.a1[b], H, wbase, g, r, row0, M, a.hpre[b]);  // hpre[b] <- act'(h_b) (Mish)
      }
    };
    // ---- layer 0
    bias_init(0);
    eng.run(acc, xin, in_rb, in_km, KS0, r, g);
    STAMP(2);
#pragma unroll
    for (int tp = 0; tp < TPW; ++tp)
#pragma unroll
      for (int m = 0; m < MR; ++m) h[tp][m] = acc[tp][m];
    if (nb > 0)
      put_block_input(0);
    else
      emit<P, TPW, MR>(h, ACT_NONE, bufA, a.hpre[0], H, wbase, g, r, row0, M);
    STAMP(3);
    if (FLAGS && nb > 0)
      hand_over(flags, wid, lane, ++seq);
    else
      __syncthreads();
    STAMP(4);
    // ---- residual blocks
    for (int b = 0; b < nb; ++b) {
      bias_init(1 + 2 * b);
      eng.run(acc, bufA, HRB, 15, KSH, r, g, FLAGS ? flags : nullptr, seq);
      STAMP(5);
      if constexpr (LN) {
        if (a.z1[b] != nullptr) emit<P, TPW, MR>(acc, ACT_NONE, nullptr, a.z1[b], H, wbase, g, r, row0, M);
        float mean[MR], rstd[MR];
        ln_forward<P, TPW, MR>(acc, acc, a.params + a.ln_off[4 * b + 2], a.params + a.ln_off[4 * b + 3], H, wbase, g, r,
                               wid, lnred, mean, rstd);
        if (a.ln_stats != nullptr && wid == 0 && g == 0) {
#pragma unroll
          for (int m = 0; m < MR; ++m)
            if (row0 + 16 * m + r < M)
              *(float2*)(a.ln_stats + (((size_t)(2 * b + 1) * M) + row0 + 16 * m + r) * 2) = make_float2(mean[m], rstd[m]);
        }
      }
      emit<P, TPW, MR>(acc, ACT, bufB, a.a2[b], H, wbase, g, r, row0, M, LN ? nullptr : a.z1[b]);  // z1[b] <- act'(z1_b) (Mish)
      STAMP(6);
      if constexpr (FLAGS)
        hand_over(flags, wid, lane, ++seq);
      else
        __syncthreads();
      STAMP(7);
      bias_init(2 + 2 * b);
      eng.run(acc, bufB, HRB, 15, KSH, r, g, FLAGS ? flags : nullptr, seq);
      STAMP(8);
#pragma unroll
      for (int tp = 0; tp < TPW; ++tp)
#pragma unroll
        for (int m = 0; m < MR; ++m) h[tp][m] += acc[tp][m];
      if (b + 1 < nb)
        put_block_input(b + 1);
      else
        emit<P, TPW, MR>(h, ACT_NONE, bufA, a.hpre[nb], H, wbase, g, r, row0, M);
      STAMP(9);
      if (FLAGS && b + 1 < nb)
        hand_over(flags, wid, lane, ++seq);
      else
        __syncthreads();  // the out layer's work items read every wave's features
      STAMP(10);
    }
    // ---- output layer: work items (row sub-tile m, out tile to, K slice kh) dealt to the 8 waves
    // (lane-derived indices pass through an empty asm here: otherwise the compiler hoists this phase's LDS addresses out
    // of the persistent tile loop, spills them, and every scratch reload then waits -- vmcnt is in issue order -- for the
    // whole ring of weight fragments in flight: measured 5k cycles of a 45k-cycle tile)
    int r_ = r, g_ = g, lane_ = lane;
    asm volatile("" : "+v"(r_), "+v"(g_), "+v"(lane_));
    auto out_items = [&](auto from_lds) {
      const int r = r_, g = g_, lane = lane_;
      const u32x4* os = a.ostream + lane;
      for (int it = wid; it < MR * OT * KSPLIT; it += SAMPLER_WAVES) {
        const int m = it % MR, to = (it / MR) % OT, kh = it / (MR * OT);
        f32x4 oacc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
        for (int c = 0; c < KPER; ++c) {
          const int ks = kh * KPER + c;
          u32x4 wf;
          if constexpr (decltype(from_lds)::value)
            wf = woutL[(ks * OT + to) * 64 + lane];
          else
            wf = os[(size_t)(ks * OT + to) * 64];
          const u32x4 xb = *(const u32x4*)(bufA + (16 * m + r) * HRB + (((ks * 4 + g) ^ (r & 15)) << 4));
          oacc = P::mma(wf, xb, oacc);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) part[(((kh * MR + m) * OT + to) * 16 + 4 * g + e) * 16 + r] = oacc[e];
      }
    };
    if (wout_lds)
      out_items(std::true_type{});
    else
      out_items(std::false_type{});
    STAMP(11);
    __syncthreads();
    STAMP(12);
    int tid_ = tid;
    asm volatile("" : "+v"(tid_));  // see emit()
    for (int idx = tid_; idx < MT * a.out_dim; idx += 512) {
      const int row = idx / a.out_dim, j = idx - row * a.out_dim;
      const int m = row >> 4, rr = row & 15, to = j >> 4, jj = j & 15;
      float s;  // (an if / else, not a ?: the compiler turns into ONE load through a generic pointer: see bias_init)
      if (bias_lds)
        s = lds_load(biasL + (1 + 2 * nb) * H + j);
      else
        s = glb_load(a.params + a.bias_off[1 + 2 * nb] + j);
#pragma unroll
      for (int kh = 0; kh < KSPLIT; ++kh) s += part[(((kh * MR + m) * OT + to) * 16 + jj) * 16 + rr];
      if (row0 + row < M) a.out[(size_t)(row0 + row) * a.ldout + j] = s;
    }
    STAMP(13);
    __syncthreads();  // the next tile's input lands in buffer B, where the partials were just read
    STAMP(14);
  }
}

// =================================================================================================
// forward, one-block networks: the block's second layer merged into the out layer
// =================================================================================================
// h_1 = h_0 + W2 act(z1) + b2 feeds the out layer only, and h_0 = W0 x + b0 is linear in the input rows, so
//     out = Wout h_1 + bout = (Wout W0) x + (Wout W2) act(z1) + [bout + Wout (b0 + b2)]
// (exact in real arithmetic; in bf16 one rounding of two composite weights instead of a rounding of the 512-wide h_1).
// The H x H layer W2 -- half of the tile's weight stream, which is what paces the k-loops (64 B/clk of L1 fill per CU) --
// is never run, h_1 is never formed or stored, and the out layer costs what it did: its K = H pass now reads act(z1)
// with the fragments of Wout W2 (the sampler's ostream2), and the K = in_dim pass on the input tile is 2-3 k-steps.
// The weight-gradient side (api.hip, mlp_backward): dWout = d_out^T h_1 is rebuilt from U = d_out^T x and T = d_out^T act(z1).
// S1 > 0 (ring depth 4): the input tile's layer is walked as its S1 = ks0v <= 2 k-steps that hold data instead of the
// padded four (CEngine); S1 = 0: the padded walk.
// FRAG (bf16, OT = 1, S1 = 1 or 2 only): act(h_0) / act(z1) leave as K-major fragments (a.a1f / a.a2f) instead of row-major.  A
// template parameter, not a run-time test: compiled into every variant, the fragment code cost the variants that never use it
// 8-24 spilled registers.
// LOSSF (bf16, OT = 1, not FRAG; training): the policy half of the PPO loss runs in the tile's epilogue (loss_dev.h) -- the tile's
// eps never goes to HBM, there is no loss launch between the actor's forward and backward.  A row's chain pair, old log-probs and
// advantage are requested behind the H-wide k-loop (their row / step indices were staged with the input tile), land in the dead
// half of buffer A behind the next barrier, and wave 0 then walks one sample per lane exactly as ppo_loss_kernel does: d loss / d eps
// rows to HBM, the tile's five statistics -- a tile is 64 samples, the loss kernel's block -- to la.partial[tile].
template <class P, int TPW, int MR, int OT, int ACT, int S1, bool FRAG, bool LOSSF>
__device__ __forceinline__ void fused_forward_merged_body(const FusedFwdArgs& a, const LossArgs& la) {
  static_assert(!LOSSF || (OT == 1 && !FRAG && MR == 4 && P::ESIZE == 2), "fused policy loss: bf16, one out tile, 64-row tiles");
  constexpr int PD = ring_depth<TPW, MR>(), ES = P::ESIZE, KB = P::KB;
  static_assert(S1 == 0 || PD == 4, "the compact walk is written for a ring of four positions");
  constexpr int H = 128 * TPW, KSH = H / KB, HRB = H * ES, MT = 16 * MR;
  // WIDE (more than one out tile, OT = 4: heads of 17-64 outputs): the Wout W2 fragments (KSH x OT KB) do not fit LDS beside
  // the two images, so they ride the weight ring instead: wave w owns out tile w % OT and K slice w / OT of the second pass
  // for ALL row sub-tiles, and its KPER fragments are the ring's third segment (CEngine AUX: KPER / TPW positions per tile,
  // +12 % of the tile's stream at H = 512 where the general forward streams a whole second H x H layer)
  constexpr bool WIDE = OT > 1;
  constexpr int KSPLIT = WIDE ? SAMPLER_WAVES / OT : (MR * OT >= 8 ? 1 : 8 / (MR * OT));
  constexpr int KPER = KSH / KSPLIT, NITEMS = MR * OT * KSPLIT, NI = (NITEMS + SAMPLER_WAVES - 1) / SAMPLER_WAVES;
  constexpr int S2 = WIDE ? KPER / TPW : 0;   // ring positions of a wave's K slice
  static_assert(!WIDE || (S1 > 0 && SAMPLER_WAVES % OT == 0 && KPER % TPW == 0 && S2 >= 1 && S2 <= 4), "wide merged head: layout");
  constexpr bool FLAGS = DPPO_FLAGS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int Kp0 = a.Kp0, M = a.M, ks0v = a.ks0v;
  // (WIDE: the input tile keeps only its ks0v k-steps that hold data, 64 bytes each, and lives beside the images -- see xin)
  const int in_rb = WIDE ? ks0v * 64 : Kp0 * ES, in_km = kmask16(in_rb), KS0 = Kp0 / KB;
  const int total = KS0 + 2 * KSH;  // the stream as packed (W2 last); the ring walks total - KSH positions
  char* bufA = smem;
  char* bufB = bufA + MT * HRB;
  float* part = (float*)bufA;   // out-layer partials: the K = H pass reads buffer B
  static_assert(KSPLIT * MR * OT * 16 * 16 * 4 <= MT * HRB, "out-layer partials must fit in buffer A");
  lds_u32* flags = (lds_u32*)(bufB + MT * HRB);              // [16 words]
  float* biasL = (float*)(bufB + MT * HRB + 64);                       // [2][H] b0, b1, then [OT*16] the out constant
  u32x4* woutL = (u32x4*)(biasL + ((2 * H + OT * 16 + 3) & ~3));       // [KSH][OT][64] fragments of Wout W2 (not WIDE)
  u32x4* w0cL = woutL + (WIDE ? 0 : KSH * OT * 64);                    // [ks0v][OT][64] fragments of Wout W0
  // the input tile: aliased with buffer B (dead once layer 0 and the out layer's first pass have run) -- except WIDE, where
  // the first pass runs late, in the second pass's phase (a partial result carried through the H-wide k-loop is 8-16 more
  // live registers in a kernel that has none to spare: 24-56 spilled), so the tile gets its own [MT][ks0v * 64 B] region
  char* xin = WIDE ? (char*)(w0cL + ks0v * OT * 64) : bufB;
  // LOSSF: [MT] (row, step) of the tile's samples, then the loss's per-step table [2 Kft + 2]; in buffer A behind the out-layer
  // partials (dead from the barrier behind the second emit on): eps tile [MT][16] f32 at +16 KB, gathered inputs [MT][64] f32 at +24 KB
  long long* metaL = (long long*)(w0cL + ks0v * OT * 64);  // per row: element offsets of its chain pair and of its old log-probs, then (sample's rollout row, step)
  float* tabL = (float*)(metaL + 3 * MT);
  float* epsL = (float*)(bufA + 16 * 1024);
  float* recL = (float*)(bufA + 24 * 1024);  // [MT][3 AF + 1] f32, rounded up to whole 512-dword DMA rounds (<= 16 KB)
  const int wbase = wid * 16 * TPW;
  const int to_w = wid % OT, kh_w = wid / OT;  // WIDE: this wave's out tile and K slice
  for (int idx = tid; idx < 2 * H; idx += 512) biasL[idx] = a.params[a.bias_off[idx / H] + idx % H];
  for (int idx = tid; idx < OT * 16; idx += 512) biasL[2 * H + idx] = idx < a.out_dim ? a.cbias2[idx] : 0.f;
  if constexpr (!WIDE)
    for (int idx = tid; idx < KSH * OT * 64; idx += 512) woutL[idx] = a.ostream2[idx];
  for (int idx = tid; idx < ks0v * OT * 64; idx += 512) w0cL[idx] = a.ostream0[idx];
  if constexpr (LOSSF) {  // ppo_loss_kernel's prologue: the per-step table, the advantage moments
    const int Kft = la.pcfg.ft_denoising_steps;
    for (int k = tid; k < 2 * Kft; k += 512) tabL[k] = la.tab[k];
    for (int k = tid; k < Kft; k += 512) tabL[2 * Kft + 2 + k] = la.tab[2 * Kft + k];  // log std_k (build_rows_kernel)
    if (tid == 0) {
      const double Nm = la.moments[2], mean = la.moments[0] / Nm;
      const double varu = (la.moments[1] - Nm * mean * mean) / (Nm - 1.0);  // unbiased (torch.std)
      tabL[2 * Kft] = (float)mean;
      tabL[2 * Kft + 1] = (float)sqrt(varu > 0 ? varu : 0);
    }
  }
  if (tid < 16) flags[tid] = 0;
  uint32_t seq = 0;
  // (visible after the first tile's barrier)

  typename std::conditional<(S1 > 0), CEngine<P, TPW, MR, (S1 > 0 ? S1 : 1), KSH, S2, WIDE>, Engine<P, TPW, MR, PD>>::type eng;
  if constexpr (S1 > 0) {
    if constexpr (WIDE) eng.set_aux(a.ostream2 + ((size_t)(kh_w * KPER) * OT + to_w) * 64, OT * 64);
    eng.prime(a.wstream + (size_t)wid * total * TPW * 64, lane, 0, KS0, 0);
  } else {
    eng.prime(a.wstream + (size_t)wid * total * TPW * 64 + lane, total - KSH);
  }

  const int ntiles = (M + MT - 1) / MT;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int row0 = tile * MT;
    STAMP(0);
    load_tile<MT>(xin, in_rb, in_km, (const char*)a.in, a.ld_in * ES, row0, M);
    if constexpr (LOSSF) {
      int t_ = tid;
      asm volatile("" : "+v"(t_));
      if (t_ < MT) {
        const int n = row0 + t_, Kft = la.pcfg.ft_denoising_steps, AF = la.AF;
        const long long b = n < M ? la.brow[n] : 0, k = n < M ? la.krow[n] : 0;
        metaL[3 * t_] = la.gathered ? b * 2 * AF : (b * (Kft + 1) + k) * AF;
        metaL[3 * t_ + 1] = la.gathered ? b * AF : (b * Kft + k) * AF;
        metaL[3 * t_ + 2] = (b << 32) | k;
      }
    }
    __syncthreads();
    STAMP(1);
    // LOSSF: touch the cache lines of the tile's loss inputs now (five dwords per row: both ends of the chain pair and of the old
    // log-probs, the advantage), a whole tile ahead of their LDS-DMA: gathered cold behind the second emit they cost every wave
    // 3-5 us of waiting per tile (the fused forward ran 114 us against 86 + a 30 us loss launch)
    uint32_t warm = 0;
    if constexpr (LOSSF) {
      int t_ = tid;
      asm volatile("" : "+v"(t_));
      const int row = t_ / 5, part = t_ - row * 5;
      if (row < MT && row0 + row < M) {
        const int AF = la.AF;
        const float* src = part < 2   ? la.chains + lds_load(metaL + 3 * row) + (part ? 2 * AF - 1 : 0)
                           : part < 4 ? la.logprobs_k + lds_load(metaL + 3 * row + 1) + (part == 3 ? AF - 1 : 0)
                                      : la.adv_k + (int)(lds_load(metaL + 3 * row + 2) >> 32);
        warm = *(const uint32_t*)src;
      }
    }
    f32x4 acc[TPW][MR];
    auto bias_init = [&](int layer) {
#pragma unroll
      for (int tp = 0; tp < TPW; ++tp) {
        const f32x4 b = *(const f32x4*)(biasL + layer * H + wbase + feat_off<P>(g, tp));
#pragma unroll
        for (int m = 0; m < MR; ++m) acc[tp][m] = b;
      }
    };
    // ---- layer 0, and the out layer's first pass on the same input tile: work item it = (row sub-tile m, out tile to,
    // K slice kh) belongs to wave it % 8 in both passes, so the partial result waits in registers
    bias_init(0);
    if constexpr (S1 > 0)
      eng.short1(acc, xin, in_rb, in_km, r, g);
    else
      eng.run(acc, xin, in_rb, in_km, KS0, r, g);
    STAMP(2);
    f32x4 o1[WIDE ? 1 : NI];
    if constexpr (!WIDE) {
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int it = wid + SAMPLER_WAVES * i;
        o1[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (it < MR * OT) {  // kh == 0
          const int m = it % MR, to = it / MR;
          for (int ks = 0; ks < ks0v; ++ks) {
            const u32x4 xb = *(const u32x4*)(xin + (16 * m + r) * in_rb + (((ks * 4 + g) ^ (r & in_km)) << 4));
            o1[i] = P::mma(w0cL[(ks * OT + to) * 64 + lane], xb, o1[i]);
          }
        }
      }
    }
    emit<P, TPW, MR>(acc, ACT, bufA, FRAG ? nullptr : a.a1[0], H, wbase, g, r, row0, M, a.hpre[0]);  // hpre[0] <- act'(h_0) (Mish) / sign words
    STAMP(3);
    if constexpr (FLAGS)
      hand_over(flags, wid, lane, ++seq);
    else
      __syncthreads();
    if constexpr (FRAG)  // act(h_0) as K-major fragments: read back from the image this wave has just written
      frag_store_image<TPW, MR>(bufA, HRB, wid, lane, a.a1f + (size_t)tile * (MR / 2) * (H / 16) * 64, H / 16);
    STAMP(4);
    // ---- the block's first layer
    bias_init(1);
    if constexpr (S1 > 0) {
      eng.template wide<!LOSSF>(acc, bufA, HRB, 15, r, g, FLAGS ? flags : nullptr, seq);
      if constexpr (!WIDE && !LOSSF) eng.end_tile();  // (WIDE: the third segment comes first, below; LOSSF: reprime() at the tile's end)
    } else {
      eng.run(acc, bufA, HRB, 15, KSH, r, g, FLAGS ? flags : nullptr, seq);
    }
    STAMP(5);
    emit<P, TPW, MR>(acc, ACT, bufB, FRAG ? nullptr : a.a2[0], H, wbase, g, r, row0, M, a.z1[0]);  // z1[0] <- act'(z1) (Mish) / sign words
    if constexpr (FRAG) frag_store_image<TPW, MR>(bufB, HRB, wid, lane, a.a2f + (size_t)tile * (MR / 2) * (H / 16) * 64, H / 16);
    STAMP(6);
    __syncthreads();  // the out layer's work items read every wave's features
    STAMP(7);
    // LOSSF: request the tile's loss inputs now, behind the second emit: per row its AF-element x_k, x_k+1 (adjacent in memory),
    // old log-probs and its advantage, one dword per slot (slot = row * DPR + element, DPR = 3 AF + 1), by LDS-DMA straight into the
    // dead part of buffer A -- no registers (held in registers across the emit they pushed the kernel from 223 to 237 VGPRs, and the
    // critic's value-loss launch could no longer slip a wave in beside these workgroups: 13 -> 80 us, step +16 us).  They travel
    // under the out layer's second pass.
    if constexpr (LOSSF) {
      asm volatile("" ::"v"(warm));  // (the touches have landed long ago: their register is free from here)
      typedef __attribute__((address_space(3))) void* lds_ptr;
      typedef const __attribute__((address_space(1))) void* glb_ptr;
      const int AF = la.AF, DPR = 3 * AF + 1, total = MT * DPR;
      int t_ = tid;
      asm volatile("" : "+v"(t_));
      for (int s0 = 0; s0 < total; s0 += 512) {
        const int sl = s0 + t_;
        int row = sl / DPR;
        const int d = sl - row * DPR;
        row = row < MT && row0 + row < M ? row : 0;  // (a DMA has no mask: slots past the tile fetch row 0's, nobody reads them)
        const float* src;
        if (d < 2 * AF)
          src = la.chains + lds_load(metaL + 3 * row) + d;
        else if (d < 3 * AF)
          src = la.logprobs_k + lds_load(metaL + 3 * row + 1) + (d - 2 * AF);
        else
          src = la.adv_k + (int)(lds_load(metaL + 3 * row + 2) >> 32);
        __builtin_amdgcn_global_load_lds((glb_ptr)src, (lds_ptr)(recL + s0 + (t_ & ~63)), 4, 0, 0);
      }
    }
    // ---- out layer, second pass: (Wout W2) on act(z1)
    int r_ = r, g_ = g, lane_ = lane;
    asm volatile("" : "+v"(r_), "+v"(g_), "+v"(lane_));  // (see fused_forward_kernel: addresses recomputed, not spilled)
    if constexpr (WIDE) {
      f32x4 ow[MR];
#pragma unroll
      for (int m = 0; m < MR; ++m) ow[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
      // first pass, (Wout W0) x: the waves of K slice 0 take it for every row sub-tile of their out tile (K = in_dim is not
      // split).  NOT shared out by row sub-tile: a row's partial sums must be added in the same order wherever the row sits
      // in a tile -- the update's recomputed log-probs equal the precomputed ones bit for bit only then (the inference and the
      // training call put a sample at different tile rows)
      if (kh_w == 0) {
        for (int ks = 0; ks < ks0v; ++ks) {
          const u32x4 wf = w0cL[(ks * OT + to_w) * 64 + lane_];
#pragma unroll
          for (int m = 0; m < MR; ++m) {
            const u32x4 xb = *(const u32x4*)(xin + (16 * m + r_) * in_rb + (((ks * 4 + g_) ^ (r_ & in_km)) << 4));
            ow[m] = P::mma(wf, xb, ow[m]);
          }
        }
      }
      if constexpr (S1 > 0) {
        eng.aux_all(ow, bufB, HRB, 15, kh_w * KPER, r_, g_);
        eng.end_tile();
      }
#pragma unroll
      for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) part[(((kh_w * MR + m) * OT + to_w) * 16 + 4 * g_ + e) * 16 + r_] = ow[m][e];
    } else
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int it = wid + SAMPLER_WAVES * i;
      if (it < NITEMS) {
        const int m = it % MR, to = (it / MR) % OT, kh = it / (MR * OT);
        f32x4 oacc = o1[i];
#pragma unroll 8
        for (int c = 0; c < KPER; ++c) {
          const int ks = kh * KPER + c;
          const u32x4 xb = *(const u32x4*)(bufB + (16 * m + r_) * HRB + (((ks * 4 + g_) ^ (r_ & 15)) << 4));
          oacc = P::mma(woutL[(ks * OT + to) * 64 + lane_], xb, oacc);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) part[(((kh * MR + m) * OT + to) * 16 + 4 * g_ + e) * 16 + r_] = oacc[e];
      }
    }
    STAMP(11);
    __syncthreads();
    STAMP(12);
    int tid_ = tid;
    asm volatile("" : "+v"(tid_));
    for (int idx = tid_; idx < MT * a.out_dim; idx += 512) {
      const int row = idx / a.out_dim, j = idx - row * a.out_dim;
      const int m = row >> 4, rr = row & 15, to = j >> 4, jj = j & 15;
      float s = biasL[2 * H + j];
#pragma unroll
      for (int kh = 0; kh < KSPLIT; ++kh) s += part[(((kh * MR + m) * OT + to) * 16 + jj) * 16 + rr];
      if constexpr (LOSSF)
        epsL[row * 16 + j] = s;
      else if (row0 + row < M)
        a.out[(size_t)(row0 + row) * a.ldout + j] = s;
    }
    STAMP(13);
    if constexpr (LOSSF) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of the gathered inputs has landed
      __syncthreads();  // the eps tile and the gathered inputs are in place
      if (wid == 0) {   // one sample per lane, as in ppo_loss_kernel (64 samples per block there: this tile)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int n = row0 + ln;
        const int cnt = (la.pcfg.reward_horizon < la.pcfg.horizon_steps ? la.pcfg.reward_horizon : la.pcfg.horizon_steps) * la.pcfg.action_dim;
        const double Nn = la.n_count > 0 ? la.n_count : la.moments[2];
        double s4[4] = {0, 0, 0, 0};
        if (n < M) {
          const int AF = la.AF;
          const float* rr = recL + ln * (3 * AF + 1);
          const float adv = lds_load(rr + 3 * AF);
          const int k = (int)(lds_load(metaL + 3 * ln + 2) & 0xffffffffll);
          policy_loss_row_nc<P>(la, tabL, k, adv, rr, rr + AF, rr + 2 * AF, epsL + ln * 16, cnt, Nn,
                                (typename P::elem_t*)la.d_eps + (size_t)n * la.ldde, s4);
        }
        // the tile's partial sums by the loss kernel's shuffle tree (v_loss belongs to the value half's launch)
        double v5[5] = {s4[0], 0.0, s4[1], s4[2], s4[3]};
#pragma unroll
        for (int q = 0; q < 5; ++q)
          for (int off = 32; off > 0; off >>= 1) v5[q] += __shfl_down(v5[q], off);
        if (ln == 0) {
          double* po = la.partial + (size_t)tile * 8;
          po[DPPO_STAT_PG_LOSS] = v5[0], po[DPPO_STAT_V_LOSS] = v5[1], po[DPPO_STAT_APPROX_KL] = v5[2];
          po[DPPO_STAT_CLIPFRAC] = v5[3], po[DPPO_STAT_RATIO] = v5[4];
        }
      }
    }
    if constexpr (LOSSF && S1 > 0) eng.reprime();  // (their latency hides under the next tile's input load and barrier)
    __syncthreads();  // the next tile's input lands in buffer B; its layer-0 emit in buffer A, where the partials were read
    STAMP(14);
  }
}

template <class P, int TPW, int MR, int OT, int ACT, int S1, bool FRAG = false, bool LOSSF = false>
__global__ __launch_bounds__(512, 2) void fused_forward_merged_kernel(const FusedFwdArgs a, const LossArgs la) {
  fused_forward_merged_body<P, TPW, MR, OT, ACT, S1, FRAG, LOSSF>(a, la);
}
// The LOSSF variant under a register cap: left alone it allocates 235 VGPRs, two of its waves leave a SIMD 32 free registers and
// the critic's launches can no longer slip a wave in beside them.  amdgpu_num_vgpr(112): on gfx90a+ the compiler
// DOUBLES the request (unified VGPR + AGPR file), so this is the plain kernel's 224; the allocator then spills eight cold values
// (hoisted tile-loop invariants, reloaded once per tile) to scratch, none inside the H-wide k-loop.
template <class P, int TPW, int MR, int ACT, int S1>
__global__ __launch_bounds__(512, 2) __attribute__((amdgpu_num_vgpr(112))) void fused_forward_merged_loss_kernel(const FusedFwdArgs a,
                                                                                                              const LossArgs la) {
  fused_forward_merged_body<P, TPW, MR, 1, ACT, S1, false, true>(a, la);
}

// =================================================================================================
// backward (data gradients + column sums)
// =================================================================================================
template <class P, int TPW, int MR, bool LN, int ACT, int OCC>
__global__ __launch_bounds__(512, 2 * OCC) void fused_backward_kernel(const FusedBwdArgs a) {
  constexpr int ES = P::ESIZE, KB = P::KB;
  // derivative sources are fetched AFTER the layer's k-loop where the registers are needed for a 4-deep ring instead
  // (16 accumulator tiles per wave), or cannot be carried at all (two workgroups per CU)
  constexpr bool FETCH_LATE = OCC > 1 || (DPPO_BWD_LATE && TPW * MR >= 16 && ES == 2 && !LN);
  constexpr int PD = (FETCH_LATE && OCC == 1) ? 4 : 2;
  constexpr int H = 128 * TPW, KSH = H / KB, HRB = H * ES, MT = 16 * MR;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr bool FLAGS = !LN && DPPO_FLAGS;  // see hand_over()
#ifndef DPPO_NO_SETPRIO
  if (!FLAGS && wid >= 4) __builtin_amdgcn_s_setprio(1);  // see fused_forward_kernel
#endif
  const int r = lane & 15, g = lane >> 4;
  const int nb = a.nb, M = a.M;
  const int in_rb = a.KpB0 * ES, in_km = kmask16(in_rb), KSB0 = a.KpB0 / KB;
  const int total = nb >= 1 ? 2 * KSB0 + (2 * nb - 1) * KSH : KSB0;
  char* bufA = smem;
  char* bufB = bufA + MT * HRB;
  char* xin = bufB;  // d_out tile: consumed by the first layer and by the top block's composite layer
  lds_u32* flags = (lds_u32*)(bufB + MT * HRB);              // [16 words]
  float* lnred = (float*)(bufB + MT * HRB + 64);  // [8 waves][2][MR][16] LayerNorm row-reduction table (LN only)
  constexpr int DRED_COLS = 128;             // d_out is at most 128 columns wide
  float* dred = lnred + (LN ? LN_WAVES * 2 * MR * 16 : 0);  // [8 waves][128] column sums of the d_out tile
  const int wbase = wid * 16 * TPW;  // lane features: wbase + feat_off<P>(g, tp) + e
  const int ntiles = (M + MT - 1) / MT;

  Engine<P, TPW, MR, PD> eng;
  eng.prime(a.bstream + (size_t)wid * total * TPW * 64 + lane, total);
  if (tid < 16) flags[tid] = 0;  // (visible after the first tile's barrier)
  uint32_t seq = 0;

  // column sums over this tile's rows of v -> colsum[slot][tile][H]; rows past M hold exact zeros
  auto colsum = [&](const f32x4 (&v)[TPW][MR], int slot, int tile) {
#pragma unroll
    for (int tp = 0; tp < TPW; ++tp) {
      f32x4 s = v[tp][0];
#pragma unroll
      for (int m = 1; m < MR; ++m) s += v[tp][m];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s[e] = row16_sum(s[e]);
      }
      if (r == 0) *(f32x4*)(a.colsum + ((size_t)slot * ntiles + tile) * H + wbase + feat_off<P>(g, tp)) = s;
    }
  };
  // same for per-lane sums that are already reduced over the row sub-tiles (LayerNorm d gamma / d beta)
  auto colsum1 = [&](const f32x4 (&v)[TPW], int slot, int tile) {
#pragma unroll
    for (int tp = 0; tp < TPW; ++tp) {
      f32x4 s = v[tp];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s[e] = row16_sum(s[e]);
      }
      if (r == 0) *(f32x4*)(a.colsum + ((size_t)slot * ntiles + tile) * H + wbase + feat_off<P>(g, tp)) = s;
    }
  };
  auto load_stats = [&](int which, int row0, float (&mean)[MR], float (&rstd)[MR]) {
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      const int grow = row0 + 16 * m + r;
      const float2 st = *(const float2*)(a.ln_stats + ((size_t)which * M + (grow < M ? grow : M - 1)) * 2);
      mean[m] = st.x, rstd[m] = st.y;
    }
  };

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int row0 = tile * MT;
    STAMP(16);
    load_tile<MT>(xin, in_rb, in_km, (const char*)a.d_out, a.ld_dout * ES, row0, M);
    __syncthreads();
    STAMP(17);
    if (a.dout_slot >= 0) {  // out-layer bias gradient: column sums of the d_out tile (rows past M are zero), part 1:
      typedef typename P::elem_t E;  // wave w adds rows w, w + 8, ...; lane = column
      for (int c = lane; c < a.out_valid; c += 64) {  // (columns past out_dim are zero padding)
        const int cb = c * ES;
        float t = 0.f;
        for (int row = wid; row < MT; row += SAMPLER_WAVES)
          t += P::to_f32(*(const E*)(xin + row * in_rb + ((((cb >> 4) ^ (row & in_km)) << 4) | (cb & 15))));
        dred[wid * DRED_COLS + c] = t;
      }
    }
    f32x4 dh[TPW][MR], acc[TPW][MR];
    auto zero_acc = [&]() {
#pragma unroll
      for (int tp = 0; tp < TPW; ++tp)
#pragma unroll
        for (int m = 0; m < MR; ++m) acc[tp][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    };
    // ---- dh[nb] = d_out . Wout
    zero_acc();
    eng.run(acc, xin, in_rb, in_km, KSB0, r, g);
    STAMP(18);
#pragma unroll
    for (int tp = 0; tp < TPW; ++tp)
#pragma unroll
      for (int m = 0; m < MR; ++m) dh[tp][m] = acc[tp][m];
    // (with blocks, nobody reads the LDS image of dh[nb]: the top block's first layer is the composite on d_out)
    emit<P, TPW, MR>(dh, ACT_NONE, nb >= 1 ? nullptr : bufA, a.dh[nb], H, wbase, g, r, row0, M);
    STAMP(19);
    colsum(dh, 0, tile);
    STAMP(20);
    auto dout_sums = [&]() {  // part 2: the eight waves' partial sums
      if (a.dout_slot >= 0 && tid < a.out_valid) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < SAMPLER_WAVES; ++w) t += dred[w * DRED_COLS + tid];
        a.colsum[((size_t)a.dout_slot * ntiles + tile) * H + tid] = t;
      }
    };
    // With blocks and flags nothing was written to an LDS image yet (the top block reads the d_out tile again), so
    // there is nothing to wait for; part 2 then runs behind the top block's second k-loop, which has seen every wave's flag.
    const bool defer_sums = FLAGS && nb >= 1;
    if (!defer_sums) {
      __syncthreads();
      dout_sums();
    }
    STAMP(21);
    // pa: image the block's first layer reads (dh[b+1]) and its second layer's emit target; pb: the other one.  The top
    // block reads the d_out tile (which lives in bufB) instead, so its dz1 goes to bufA and the roles swap from there on.
    char* pa = bufA;
    char* pb = bufB;
    for (int b = nb - 1; b >= 0; --b) {
      const bool top = b == nb - 1;
      // ---- dz1 = (dh . W2) * act'(z1)       [LayerNorm: back through act(LN2(z1))]
      // top block: dh[nb] . W2 = d_out . (Wout . W2) -- a K = out_dim layer on the d_out tile instead of a K = H one
      u32x4 d[MR][Chunks<P, TPW>::CH];
      // derivative sources: issued ahead of the layer so the latency hides under it -- except at two workgroups per CU,
      // where 128 VGPRs cannot carry them through the k-loop and the other workgroup hides the latency instead
      if constexpr (!FETCH_LATE) fetch<P, ACT == ACT_RELU && !LN, TPW>(d, a.m1[b], H, wbase, g, r, row0, M);
      zero_acc();
      // (one call site with selected operands: a second inlined copy of the k-loop costs 150 spilled VGPRs)
      eng.run(acc, top ? xin : pa, top ? in_rb : HRB, top ? in_km : 15, top ? KSB0 : KSH, r, g,
              FLAGS && !top ? flags : nullptr, seq);
      if constexpr (FETCH_LATE) fetch<P, ACT == ACT_RELU && !LN, TPW>(d, a.m1[b], H, wbase, g, r, row0, M);
      STAMP(22);
      if constexpr (LN) {
        float mean[MR], rstd[MR];
        f32x4 dga[TPW], dbe[TPW];
        load_stats(2 * b + 1, row0, mean, rstd);
        ln_backward<P, TPW, MR>(acc, d, a.params + a.ln_off[4 * b + 2], a.params + a.ln_off[4 * b + 3], mean, rstd, ACT, H,
                                wbase, g, r, wid, lnred, dga, dbe);
        const int ls = (2 * nb + 1) + 4 * (nb - 1 - b);
        colsum1(dga, ls + 2, tile);
        colsum1(dbe, ls + 3, tile);
      } else {
#pragma unroll
        for (int tp = 0; tp < TPW; ++tp)
#pragma unroll
          for (int m = 0; m < MR; ++m)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[tp][m][e] *= grad_at<P, ACT>(d, tp, m, e);
      }
      if (top) {  // xin == bufB is still being read by slower waves: dz1 goes to the other image
        char* t = pa;
        pa = pb, pb = t;
      }
      emit<P, TPW, MR>(acc, ACT_NONE, pb, a.dz1[b], H, wbase, g, r, row0, M);
      STAMP(23);
      colsum(acc, (nb + 1) + (nb - 1 - b), tile);
      STAMP(24);
      if constexpr (FLAGS)
        hand_over(flags, wid, lane, ++seq);
      else
        __syncthreads();
      STAMP(25);
      // ---- dh[b] = dh[b+1] + (dz1 . W1) * act'(h_b)       [LayerNorm: back through act(LN1(h_b))]
      if constexpr (!FETCH_LATE) fetch<P, ACT == ACT_RELU && !LN, TPW>(d, a.m0[b], H, wbase, g, r, row0, M);
      zero_acc();
      eng.run(acc, pb, HRB, 15, KSH, r, g, FLAGS ? flags : nullptr, seq);
      if (top && defer_sums) dout_sums();
      if constexpr (FETCH_LATE) fetch<P, ACT == ACT_RELU && !LN, TPW>(d, a.m0[b], H, wbase, g, r, row0, M);
      STAMP(26);
      if constexpr (LN) {
        float mean[MR], rstd[MR];
        f32x4 dga[TPW], dbe[TPW];
        load_stats(2 * b, row0, mean, rstd);
        ln_backward<P, TPW, MR>(acc, d, a.params + a.ln_off[4 * b], a.params + a.ln_off[4 * b + 1], mean, rstd, ACT, H,
                                wbase, g, r, wid, lnred, dga, dbe);
        const int ls = (2 * nb + 1) + 4 * (nb - 1 - b);
        colsum1(dga, ls, tile);
        colsum1(dbe, ls + 1, tile);
#pragma unroll
        for (int tp = 0; tp < TPW; ++tp)
#pragma unroll
          for (int m = 0; m < MR; ++m) dh[tp][m] += acc[tp][m];
      } else {
#pragma unroll
        for (int tp = 0; tp < TPW; ++tp)
#pragma unroll
          for (int m = 0; m < MR; ++m)
#pragma unroll
            for (int e = 0; e < 4; ++e) dh[tp][m][e] += acc[tp][m][e] * grad_at<P, ACT>(d, tp, m, e);
      }
      emit<P, TPW, MR>(dh, ACT_NONE, b > 0 ? pa : nullptr, a.dh[b], H, wbase, g, r, row0, M);  // (nobody reads dh[0]'s image)
      STAMP(27);
      colsum(dh, nb - b, tile);
      STAMP(28);
      if (FLAGS && b > 0)
        hand_over(flags, wid, lane, ++seq);
      else
        __syncthreads();  // tile end: the next tile's d_out lands in buffer B
      STAMP(29);
    }
  }
}

// =================================================================================================
// backward, one-block networks
// =================================================================================================
// With one block, dh_0 = dh_1 + (dz1 . W1) o act'(h_0) and dh_1 = d_out . Wout is a K = out_dim product on the d_out tile
// that stays in LDS for the whole tile: it is added LAST, into the accumulators of the W1^T layer, instead of first into a
// running dh that then occupies 16 * TPW * MR / 64 registers per lane through both H-wide phases.  What that buys: no
// dh_1 phase (emit + column sums: 6k of a 51k-cycle tile; the second layer's bias gradient colsum(dh_1) = colsum(d_out) . Wout
// is formed behind the slab reduce, PostReduce::db2), room to fetch the derivative sources BEFORE each k-loop again (their
// latency hides under it), and at H = 256 room for the forward's 128-row tile (half the weight stream per row).  The stream
// is the ordinary backward stream [Wout^T | (Wout W2)^T | W1^T] walked from its second layer on: the ring wraps at the end
// of a tile anyway.  Needs the low-rank dW2 (no dh_1 tensor is written).
// COMPACT (out_dim <= one k-step, ring depth 4): the two layers on the d_out tile are walked as ONE k-step each instead of
// their padded four (CEngine).
// FRAG (bf16, COMPACT only): dz1 / dh_0 leave as K-major fragments, with fragment copies of the d_out tile and the input rows.
// DW0 (bf16, COMPACT only; 1, or 2 = FusedBwdArgs::dw0_round): dh_0 is not stored at all -- its product with the input rows is
// accumulated in LDS (dw0_accumulate, FusedBwdArgs::dw0_slab) and buffer B shrinks to the d_out tile.
template <class P, int TPW, int MR, int ACT, bool COMPACT, bool FRAG = false, int DW0 = 0>
__global__ __launch_bounds__(512, 2) void fused_backward_one_kernel(const FusedBwdArgs a) {
  static_assert(!DW0 || (COMPACT && !FRAG && P::ESIZE == 2), "in-kernel dW0: bf16, compact walk, row-major dz1");
  constexpr int ES = P::ESIZE, KB = P::KB, PD = ring_depth<TPW, MR>();
  static_assert(!COMPACT || PD == 4, "the compact walk is written for a ring of four positions");
  constexpr int H = 128 * TPW, KSH = H / KB, HRB = H * ES, MT = 16 * MR;
  constexpr bool FLAGS = DPPO_FLAGS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int M = a.M;
  // (DW0: the compact walk reads the d_out tile's first k-step only -- its LDS tile keeps just those 64 bytes per row)
  const int in_rb = DW0 ? 64 : a.KpB0 * ES, in_km = kmask16(in_rb), KSB0 = a.KpB0 / KB;
  const int total = 2 * KSB0 + KSH;
  char* bufA = smem;
  char* bufB = bufA + MT * HRB;
  char* xin = bufB;  // the d_out tile: read by the composite layer first and by the Wout^T layer last
  const int bsz = DW0 ? MT * in_rb : MT * HRB;                // (DW0: buffer B is the d_out tile alone)
  lds_u32* flags = (lds_u32*)(bufB + bsz);                    // [16 words]
  constexpr int DRED_COLS = DW0 ? 32 : 128;  // (the compact walk: at most one k-step of outputs)
  float* dred = (float*)(bufB + bsz + 64);  // [8 waves][DRED_COLS] column sums of the d_out tile
  // DW0: the input-row tile [MT][128 B], a 2 KB staging region per wave, the workgroup's dW0 partial [8 waves][TPW][2][64] f32x4
  char* xw = (char*)(dred + SAMPLER_WAVES * DRED_COLS);
  char* stage_w = xw + MT * 128 + wid * 2048;
  f32x4* cacc = (f32x4*)(xw + MT * 128 + SAMPLER_WAVES * 2048) + (size_t)wid * TPW * 2 * 64;
  // fragment mode (a.dz1f): buffer B beyond the d_out tile (at most MT x 256 bytes of its MT x HRB) also holds the input-row
  // tile [MT][ld_x] and each wave's private staging region for dh_0 (frag_store_acc): 16 + 16 + 32 KB of 64 at H = 512
  // (the launcher checked MT (in_rb + ld_x ES) + 8 x 32 x 32 TPW <= MT HRB: fused_frag_fits())
  char* xt = bufB + MT * in_rb;
  char* stage = xt + MT * a.ld_x * ES + wid * (32 * 32 * TPW);
  const int wbase = wid * 16 * TPW;
  const int ntiles = (M + MT - 1) / MT;

  typename std::conditional<COMPACT, CEngine<P, TPW, MR, 1, KSH, 1>, Engine<P, TPW, MR, PD>>::type eng;
  if constexpr (COMPACT)
    eng.prime(a.bstream + (size_t)wid * total * TPW * 64, lane, KSB0, 2 * KSB0, 0);
  else
    eng.prime(a.bstream + (size_t)wid * total * TPW * 64 + lane, total, KSB0);
  if (tid < 16) flags[tid] = 0;  // (visible after the first tile's barrier)
  uint32_t seq = 0;
  if constexpr (DW0) {  // (a wave's region is its own: no barrier)
#pragma unroll
    for (int i = 0; i < TPW * 2; ++i) cacc[i * 64 + lane] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  auto colsum = [&](const f32x4 (&v)[TPW][MR], int slot, int tile) {
    int g = lane >> 4;
    asm volatile("" : "+v"(g));  // (the store addresses are formed here, not hoisted out of the tile loop as 64-bit pairs: see emit())
#pragma unroll
    for (int tp = 0; tp < TPW; ++tp) {
      f32x4 s = v[tp][0];
#pragma unroll
      for (int m = 1; m < MR; ++m) s += v[tp][m];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s[e] = row16_sum(s[e]);
      }
      if (r == 0) *(f32x4*)(a.colsum + ((size_t)slot * ntiles + tile) * H + wbase + feat_off<P>(g, tp)) = s;
    }
  };

  // DW0: the NEXT tile's d_out and input-row chunks are requested before this tile's dW0 phase and written to LDS behind the
  // tile-end barrier (pd / px: one or two 16-byte chunks per thread), and one dword per 128 bytes of its first derivative-source
  // rows is touched at the same point, so that the fetch at the tile's start finds them in L2: the tile start -- request, wait for
  // first-touch HBM rows, barrier -- was 5k cycles of a 32k-cycle tile with every wave idle (profiles/r02_final_fused_phase_stamps.txt)
  constexpr bool PF = DW0 != 0 && DPPO_BWD_PREFETCH;
  constexpr int ND = (MT * 4 + 511) / 512, NX = (MT * 8 + 511) / 512;
  u32x4 pd[PF ? ND : 1], px[PF ? NX : 1];
  uint32_t warm = 0;
  auto tile_prefetch = [&](int t) {
    const int prow0 = t * MT;
    int t0 = tid;
    asm volatile("" : "+v"(t0));  // (see load_tile)
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      const int q = t0 + 512 * i, row = q >> 2, c = q & 3, grow = prow0 + row;
      pd[i] = (u32x4){0, 0, 0, 0};
      if (q < MT * 4 && c * 16 < a.ld_dout * ES && grow < M) pd[i] = *(const u32x4*)((const char*)a.d_out + (size_t)grow * a.ld_dout * ES + c * 16);
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int q = t0 + 512 * i, row = q >> 3, c = q & 7, grow = prow0 + row;
      px[i] = (u32x4){0, 0, 0, 0};
      if (q < MT * 8 && grow < M) px[i] = *(const u32x4*)((const char*)a.xc + (size_t)grow * a.ld_xc * ES + c * 16);
    }
  };
  auto tile_commit = [&]() {
    int t0 = tid;
    asm volatile("" : "+v"(t0));
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      const int q = t0 + 512 * i, row = q >> 2, c = q & 3;
      if (q < MT * 4) *(u32x4*)(xin + row * 64 + ((c ^ (row & 3)) << 4)) = pd[i];
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int q = t0 + 512 * i, row = q >> 3, c = q & 7;
      if (q < MT * 8) *(u32x4*)(xw + row * 128 + ((c ^ (row & 7)) << 4)) = px[i];
    }
  };
  auto warm_rows = [&](const void* src, int t) {  // the derivative-source rows of tile t: [MT][H] elem, or the ReLU sign words
    constexpr int RB = ACT == ACT_RELU ? SIGN_WORDS * 4 : HRB;
    const size_t off = (size_t)t * MT * RB + (size_t)tid * 128;
    if (src != nullptr && tid * 128 < MT * RB && off + 4 <= (size_t)M * RB) warm = *(const uint32_t*)((const char*)src + off);
  };
  if constexpr (PF) tile_prefetch(blockIdx.x);

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int row0 = tile * MT;
    STAMP(16);
    // the first phase's derivative sources: issued with the tile's own loads, so that one memory latency covers both
    u32x4 d[MR][Chunks<P, TPW>::CH];
    fetch<P, ACT == ACT_RELU, TPW>(d, a.m1[0], H, wbase, g, r, row0, M);
    if constexpr (PF) {
      asm volatile("" ::"v"(warm));  // (the touch has landed: its register is free again)
      tile_commit();
    } else {
      load_tile<MT>(xin, in_rb, in_km, (const char*)a.d_out, a.ld_dout * ES, row0, M);
      if constexpr (DW0) load_tile<MT>(xw, 128, 7, (const char*)a.xc, a.ld_xc * ES, row0, M);  // (the rows' first 64 columns)
    }
    if constexpr (FRAG) {
      const int x_rb = a.ld_x * ES, x_km = kmask16(x_rb);
      load_tile<MT>(xt, x_rb, x_km, (const char*)a.x, x_rb, row0, M);
      __syncthreads();
      // fragment copies of the two small GEMM operands (rows past M are zero: load_tile)
      const size_t kbase = (size_t)tile * (MR / 2);
      frag_store_tile<MR>(xin, in_rb, in_km, a.dof_nt, wid, lane, a.doutf + kbase * a.dof_nt * 64, a.dof_nt);
      frag_store_tile<MR>(xt, x_rb, x_km, a.ld_x / 16, wid, lane, a.xf + kbase * (a.ld_x / 16) * 64, a.ld_x / 16);
    } else {
      __syncthreads();
    }
    STAMP(17);
    if (a.dout_slot >= 0) {  // out-layer bias gradient: column sums of the d_out tile (rows past M are zero), part 1
      typedef typename P::elem_t E;
      for (int c = lane; c < a.out_valid; c += 64) {  // (columns past out_dim are zero padding)
        const int cb = c * ES;
        float t = 0.f;
        for (int row = wid; row < MT; row += SAMPLER_WAVES)
          t += P::to_f32(*(const E*)(xin + row * in_rb + ((((cb >> 4) ^ (row & in_km)) << 4) | (cb & 15))));
        dred[wid * DRED_COLS + c] = t;
      }
    }
    STAMP(18);
    f32x4 acc[TPW][MR];
    auto zero_acc = [&]() {
#pragma unroll
      for (int tp = 0; tp < TPW; ++tp)
#pragma unroll
        for (int m = 0; m < MR; ++m) acc[tp][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    };
    // ---- dz1 = (dh_1 . W2) o act'(z1), dh_1 . W2 = d_out . (Wout . W2): a K = out_dim layer on the d_out tile
    zero_acc();
    STAMP(19);
    if constexpr (COMPACT)
      eng.short1(acc, xin, in_rb, in_km, r, g);
    else
      eng.run(acc, xin, in_rb, in_km, KSB0, r, g);
    STAMP(22);
#pragma unroll
    for (int tp = 0; tp < TPW; ++tp)
#pragma unroll
      for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[tp][m][e] *= grad_at<P, ACT>(d, tp, m, e);
    STAMP(20);
    emit<P, TPW, MR>(acc, ACT_NONE, bufA, FRAG ? nullptr : a.dz1[0], H, wbase, g, r, row0, M);
    STAMP(23);
    colsum(acc, 2, tile);
    STAMP(24);
    if constexpr (FLAGS)
      hand_over(flags, wid, lane, ++seq);
    else
      __syncthreads();
    if constexpr (FRAG) frag_store_image<TPW, MR>(bufA, HRB, wid, lane, a.dz1f + (size_t)tile * (MR / 2) * (H / 16) * 64, H / 16);
    STAMP(25);
    // ---- dh_0 = (dz1 . W1) o act'(h_0) + d_out . Wout
    fetch<P, ACT == ACT_RELU, TPW>(d, a.m0[0], H, wbase, g, r, row0, M);
    zero_acc();
    STAMP(21);
    if constexpr (COMPACT)
      eng.wide(acc, bufA, HRB, 15, r, g, FLAGS ? flags : nullptr, seq);
    else
      eng.run(acc, bufA, HRB, 15, KSH, r, g, FLAGS ? flags : nullptr, seq);
    STAMP(26);
    if (a.dout_slot >= 0 && tid < a.out_valid) {  // part 2 (every wave has passed part 1: its flag / the barrier came later)
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < SAMPLER_WAVES; ++w) t += dred[w * DRED_COLS + tid];
      a.colsum[((size_t)a.dout_slot * ntiles + tile) * H + tid] = t;
    }
#pragma unroll
    for (int tp = 0; tp < TPW; ++tp)
#pragma unroll
      for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[tp][m][e] *= grad_at<P, ACT>(d, tp, m, e);
    if constexpr (COMPACT) {
      eng.short2(acc, xin, in_rb, in_km, r, g);
      eng.end_tile();
    } else {
      eng.run(acc, xin, in_rb, in_km, KSB0, r, g);
    }
    if constexpr (PF) {
      if (tile + (int)gridDim.x < ntiles) {
        tile_prefetch(tile + gridDim.x);
        warm_rows(a.m1[0], tile + gridDim.x);
      }
    }
    if constexpr (DW0) {  // (rows past M: d_out is zero there, hence dh_0 too; the xc tile is zero there as well)
      dw0_accumulate<TPW, MR, DW0 == 2>(acc, stage_w, xw, cacc, lane, a.xc_af, a.xc_skip);
    } else if constexpr (FRAG)
      frag_store_acc<TPW, MR>(acc, stage, wid, lane, a.dh0f + (size_t)tile * (MR / 2) * (H / 16) * 64, H / 16);
    else
      emit<P, TPW, MR>(acc, ACT_NONE, nullptr, a.dh[0], H, wbase, g, r, row0, M);
    STAMP(27);
    colsum(acc, 1, tile);
    STAMP(28);
    __syncthreads();  // tile end: the next tile's d_out lands in buffer B, its dz1 image in buffer A
    STAMP(29);
  }
  if constexpr (DW0) {  // this workgroup's partial dW0 -> dw0_slab[blockIdx.x][H][32]
    float* sl = a.dw0_slab + (size_t)blockIdx.x * H * 32;
#pragma unroll
    for (int i = 0; i < TPW * 2; ++i) {
      const f32x4 c = cacc[i * 64 + lane];
#pragma unroll
      for (int e = 0; e < 4; ++e) sl[(size_t)(wbase + 16 * (i >> 1) + 4 * g + e) * 32 + 16 * (i & 1) + r] = c[e];
    }
  }
}

#ifdef DPPO_STAMPS
extern "C" int dppo_debug_stamps(unsigned long long* out) {  // out: [8 waves][32]
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 8 * 32);
}
#endif

// =================================================================================================
// host side
// =================================================================================================
// tuning knob 7: short tiles at two workgroups per CU (bf16, H = 512, no LayerNorm): bit 0 backward, bit 1 forward
static int g_short_tiles = 0;
void set_fused_short_tiles(int v) { g_short_tiles = v; }
template <class P>
static bool short_tiles(int hidden, int ln, int bit) {
  return P::ESIZE == 2 && hidden == 512 && !ln && ((g_short_tiles >> bit) & 1);
}
template <class P>
static int pick_mr(int hidden) {
  const int tpw = hidden / 128;
  if (hidden % 128) return 0;
  // rows per tile = 16*MR: as many as the two [rows][H] LDS images (<= 64 KiB each) and 256 VGPRs allow --
  // the per-tile weight stream is a fixed cost, so fewer, taller tiles win
  if (P::ESIZE == 2) return tpw == 2 ? 8 : (tpw == 4 ? 4 : (tpw == 8 ? 2 : 0));
  return tpw == 2 ? 4 : (tpw == 4 ? 2 : (tpw == 8 ? 1 : 0));
}
// the backward kernel also holds the running dh and the packed derivative sources: one size smaller at H = 1024
// (and once more with LayerNorm, whose backward keeps two more row-statistics / gradient sets live)
template <class P>
static int pick_mr_bwd(int hidden, int ln) {
  if (short_tiles<P>(hidden, ln, 0)) return 2;
  if (P::ESIZE == 2 && hidden == 256 && !ln && ((g_short_tiles >> 3) & 1)) return 2;
  int mr = pick_mr<P>(hidden);
  if (hidden >= 1024 && mr > 1) mr /= 2;
  if (P::ESIZE == 2 && hidden == 256) mr /= 2;  // 8 row sub-tiles of dh, acc, derivative chunks and B fragments do not fit 256 VGPRs
  if (ln && mr > 1) mr /= 2;
  return mr;
}
// one-block backward (fused_backward_one_kernel): the forward's tile height
static int g_bwd_one = 1;  // tuning knob 23
void set_fused_bwd_one(int v) { g_bwd_one = v; }
template <class P>
bool fused_bwd_one_block(const dppo_net_desc& d) {
  return g_bwd_one && !d.plain && !d.use_layernorm && d.n_blocks == 1 && d.hidden % 128 == 0 && d.hidden <= 1024 &&
         pick_mr<P>(d.hidden) > 0;
}
template bool fused_bwd_one_block<F32>(const dppo_net_desc&);
template bool fused_bwd_one_block<BF16>(const dppo_net_desc&);
template <class P>
int fused_rows_per_tile(const dppo_net_desc& d, bool one_block) {
  return 16 * (one_block ? pick_mr<P>(d.hidden) : pick_mr_bwd<P>(d.hidden, d.use_layernorm));
}
template int fused_rows_per_tile<F32>(const dppo_net_desc&, bool);
template int fused_rows_per_tile<BF16>(const dppo_net_desc&, bool);

template <class P>
FusedGeom fused_geom(const dppo_net_desc& d) {
  FusedGeom g;
  const int pd = sampler_pd(d.hidden);
  g.KpB0 = round_up(round_up(d.out_dim, 64), pd * P::KB);
  g.KSB0 = g.KpB0 / P::KB;
  // top block: its W2^T layer is replaced by the composite (Wout . W2)^T, as short as the Wout^T layer (see FusedBwdArgs)
  g.total_pos = d.n_blocks >= 1 ? 2 * g.KSB0 + (2 * d.n_blocks - 1) * (d.hidden / P::KB) : g.KSB0;
  g.frags_per_wave = (size_t)g.total_pos * (d.hidden / 128);
  return g;
}
template FusedGeom fused_geom<F32>(const dppo_net_desc&);
template FusedGeom fused_geom<BF16>(const dppo_net_desc&);

constexpr int NUM_CUS = 256;

template <class K>
static void raise_lds(K kern, DevLatch& done) {
  if (done.need()) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    done.done();
  }
}

template <class P, int TPW, int MR, int OT, bool LN, int ACT, int OCC = 1>
static int launch_fwd_cfg(const FusedFwdArgs& a, hipStream_t s) {
  constexpr int ES = P::ESIZE, MT = 16 * MR, H = 128 * TPW;
  size_t lds = 2 * (size_t)MT * H * ES + 64 + (LN ? (size_t)LN_WAVES * MR * 16 * 4 : 0);
  constexpr size_t cap = 160 * 1024 / OCC;  // OCC workgroups share a CU's LDS
  if (lds > cap || a.Kp0 > H) return -2;
  FusedFwdArgs b = a;  // constants staged in LDS as far as it reaches: biases first, then the out-layer fragments
  const size_t bias_bytes = ((size_t)((1 + 2 * a.nb) * H + OT * 16 + 3) & ~(size_t)3) * 4;
  const size_t wout_bytes = (size_t)(H / P::KB) * OT * 64 * 16;
  b.consts_lds = 0;
  if (lds + bias_bytes <= cap) {
    b.consts_lds |= 1, lds += bias_bytes;
    if (lds + wout_bytes <= cap) b.consts_lds |= 2, lds += wout_bytes;
  }
  static DevLatch attr;
  raise_lds(fused_forward_kernel<P, TPW, MR, OT, LN, ACT, OCC>, attr);
  const int ntiles = (a.M + MT - 1) / MT;
  const bool probe = probe_begin(PROBE_FUSED_FWD, s);
  hipLaunchKernelGGL((fused_forward_kernel<P, TPW, MR, OT, LN, ACT, OCC>),
                     dim3(ntiles < NUM_CUS * OCC ? ntiles : NUM_CUS * OCC), dim3(512), lds, s, b);
  if (probe) probe_end(s, 2.0 * a.M * ((double)a.in_valid * H + 2.0 * a.nb * H * H + (double)H * a.out_dim));
  return 0;
}

// merged-top forward (fused_forward_merged_kernel): its LDS need; everything it stages must fit, or there is no merge
template <class P>
static size_t merged_lds(int hidden, int out_tiles, int ks0v) {
  const int mr = pick_mr<P>(hidden);
  // (a wide head's Wout W2 fragments ride the weight ring instead of LDS)
  return 2 * (size_t)16 * mr * hidden * P::ESIZE + 64 + (((size_t)2 * hidden + out_tiles * 16 + 3) & ~(size_t)3) * 4 +
         ((size_t)(out_tiles > 1 ? 0 : hidden / P::KB) + ks0v) * out_tiles * 64 * 16 +
         (out_tiles > 1 ? (size_t)16 * mr * ks0v * 64 : 0);  // ... and its input tile has a region of its own
}
static int g_merge_fwd = 1;  // tuning knob 22: the fused forward of one-block networks merges the block's second layer into the out layer
static int g_merge_wide = 1;  // ... bit 1 of the knob's value switches the 17-64-output form off alone (22 = 1: both on, 3: narrow heads only)
void set_fused_merge_fwd(int v) { g_merge_fwd = v & 1, g_merge_wide = (v & 2) ? 0 : 1; }
int fused_compact_on();
static int round_up_i(int x, int m) { return (x + m - 1) / m * m; }
template <class P>
bool fused_can_merge(const dppo_net_desc& d) {
  if (!g_merge_fwd || d.plain || d.use_layernorm || d.n_blocks != 1 || d.hidden % 128 || pick_mr<P>(d.hidden) == 0) return false;
  const int ks0v = (d.in_dim + P::KB - 1) / P::KB;
  if (d.out_dim > 16) {  // 17-64 outputs (out-stream geometry OT = 4): hidden 512, compact walk of the input layer
    constexpr int PD = 4;
    return g_merge_wide && fused_compact_on() && d.out_dim <= 64 && d.hidden == 512 && ks0v >= 1 && ks0v <= 3 &&
           round_up_i(d.in_dim, PD * P::KB) / P::KB >= ks0v && merged_lds<P>(d.hidden, 4, ks0v) <= 160 * 1024;
  }
  return d.hidden <= 512 && merged_lds<P>(d.hidden, 1, ks0v) <= 160 * 1024;
}
// shapes the fragment-output variants are built for: bf16, head of one out tile, at most two input k-steps, compact walks on
bool fused_frag_shape(const dppo_net_desc& d) {
  return fused_compact_on() && d.out_dim <= 16 && d.in_dim <= 2 * BF16::KB && (d.hidden == 512 || d.hidden == 256) && fused_can_merge<BF16>(d);
}
template bool fused_can_merge<F32>(const dppo_net_desc&);
template bool fused_can_merge<BF16>(const dppo_net_desc&);

// shapes the fused policy loss covers (the caller adds its own conditions on the loss's arguments: fused_loss_shape())
constexpr int FUSED_LOSS_MAX_KFT = 64;
static size_t fused_loss_lds(int Kft) { return (size_t)3 * 64 * 8 + ((size_t)(3 * Kft + 2) * 4 + 15) / 16 * 16; }
template <class P, int TPW, int MR, int ACT, int S1, int OT = 1, bool FRAG = false, bool LOSSF = false>
static int launch_fwd_merged_cfg2(const FusedFwdArgs& a, hipStream_t s, const LossArgs* loss = nullptr) {
  constexpr int MT = 16 * MR, H = 128 * TPW;
  const size_t lds = merged_lds<P>(H, OT, a.ks0v) + (LOSSF ? fused_loss_lds(loss->pcfg.ft_denoising_steps) : 0);
  if (lds > 160 * 1024 || a.Kp0 > H || a.nb != 1) return -2;
  if (FRAG != (a.a1f != nullptr) || FRAG != (a.a2f != nullptr)) return -4;  // (fragment outputs: both or none, and a variant built for them)
  if (LOSSF != (loss != nullptr)) return -4;
  static DevLatch attr;
  const int ntiles = (a.M + MT - 1) / MT;
  static const LossArgs no_loss = {};
  if constexpr (LOSSF) {  // (its own kernel symbol: the same body under a register cap)
    raise_lds(fused_forward_merged_loss_kernel<P, TPW, MR, ACT, S1>, attr);
    const bool probe = probe_begin(PROBE_FUSED_FWD, s);
    hipLaunchKernelGGL((fused_forward_merged_loss_kernel<P, TPW, MR, ACT, S1>), dim3(ntiles < NUM_CUS ? ntiles : NUM_CUS), dim3(512), lds, s,
                       a, *loss);
    if (probe) probe_end(s, 2.0 * a.M * ((double)a.in_valid * H + 2.0 * a.nb * H * H + (double)H * a.out_dim));
    return 0;
  }
  raise_lds(fused_forward_merged_kernel<P, TPW, MR, OT, ACT, S1, FRAG, LOSSF>, attr);
  const bool probe = probe_begin(PROBE_FUSED_FWD, s);
  hipLaunchKernelGGL((fused_forward_merged_kernel<P, TPW, MR, OT, ACT, S1, FRAG, LOSSF>), dim3(ntiles < NUM_CUS ? ntiles : NUM_CUS), dim3(512),
                     lds, s, a, loss != nullptr ? *loss : no_loss);
  if (probe) probe_end(s, 2.0 * a.M * ((double)a.in_valid * H + 2.0 * a.nb * H * H + (double)H * a.out_dim));
  return 0;
}
template <class P, int TPW, int MR, int ACT>
static int launch_fwd_merged_cfg(const FusedFwdArgs& a, hipStream_t s, const LossArgs* loss) {
  if (loss != nullptr) {  // the policy loss in the epilogue: bf16 64-row tiles, one out tile, compact walk (fused_loss_shape())
    if constexpr (P::ESIZE == 2 && MR == 4 && ring_depth<TPW, MR>() == 4) {
      if (a.out_dim <= 16 && a.a1f == nullptr && fused_compact_on() && a.ks0v == 1) return launch_fwd_merged_cfg2<P, TPW, MR, ACT, 1, 1, false, true>(a, s, loss);
      if (a.out_dim <= 16 && a.a1f == nullptr && fused_compact_on() && a.ks0v == 2) return launch_fwd_merged_cfg2<P, TPW, MR, ACT, 2, 1, false, true>(a, s, loss);
    }
    return -4;
  }
  if (a.out_dim > 16) {  // the wide head (fused_can_merge admitted it: hidden 512, compact walk, ks0v <= 3)
    if constexpr (TPW == 4) {
      if (a.ks0v == 1) return launch_fwd_merged_cfg2<P, TPW, MR, ACT, 1, 4>(a, s);
      if (a.ks0v == 2) return launch_fwd_merged_cfg2<P, TPW, MR, ACT, 2, 4>(a, s);
      if (a.ks0v == 3) return launch_fwd_merged_cfg2<P, TPW, MR, ACT, 3, 4>(a, s);
    }
    return -1;
  }
  if constexpr (ring_depth<TPW, MR>() == 4) {
    if constexpr (P::ESIZE == 2) {  // fragment outputs: built for the compact narrow-head variants only (fused_frag_shape())
      if (a.a1f != nullptr) {
        if (fused_compact_on() && a.ks0v == 1) return launch_fwd_merged_cfg2<P, TPW, MR, ACT, 1, 1, true>(a, s);
        if (fused_compact_on() && a.ks0v == 2) return launch_fwd_merged_cfg2<P, TPW, MR, ACT, 2, 1, true>(a, s);
        return -4;
      }
    }
    if (fused_compact_on() && a.ks0v == 1) return launch_fwd_merged_cfg2<P, TPW, MR, ACT, 1>(a, s);
    if (fused_compact_on() && a.ks0v == 2) return launch_fwd_merged_cfg2<P, TPW, MR, ACT, 2>(a, s);
  }
  return launch_fwd_merged_cfg2<P, TPW, MR, ACT, 0>(a, s);
}

// the fused policy loss exists for: bf16, hidden 512 (64-row tiles), a head of at most 16 outputs, at most two input k-steps
bool fused_loss_shape(const dppo_net_desc& d) {
  return fused_compact_on() && d.hidden == 512 && d.out_dim <= 16 && d.in_dim <= 2 * BF16::KB && fused_can_merge<BF16>(d);
}
template <class P>
int launch_fused_forward(const dppo_net_desc& d, const FusedFwdArgs& a, hipStream_t s, const LossArgs* loss) {
  const int tpw = d.hidden / 128, mr = pick_mr<P>(d.hidden);
  const int nt = (d.out_dim + 15) / 16, ot = nt <= 1 ? 1 : (nt <= 4 ? 4 : (nt <= 8 ? 8 : 0));
  if (mr == 0 || ot == 0 || a.M <= 0) return -1;
  const bool relu = a.act == ACT_RELU;  // check_net admits ReLU and Mish only
  if (a.merge_top) {  // (the caller asked fused_can_merge() first)
#define DPPO_FWDM(T, R) \
  if (tpw == T && mr == R) return relu ? launch_fwd_merged_cfg<P, T, R, ACT_RELU>(a, s, loss) : launch_fwd_merged_cfg<P, T, R, ACT_MISH>(a, s, loss);
    if constexpr (P::ESIZE == 2) {
      DPPO_FWDM(2, 8) DPPO_FWDM(4, 4)
    } else {
      DPPO_FWDM(2, 4) DPPO_FWDM(4, 2)
    }
#undef DPPO_FWDM
    return -1;
  }
  if (loss != nullptr) return -4;
  if constexpr (P::ESIZE == 2) {
    if (short_tiles<P>(d.hidden, a.use_ln, 1) && ot == 1)
      return relu ? launch_fwd_cfg<P, 4, 2, 1, false, ACT_RELU, 2>(a, s) : launch_fwd_cfg<P, 4, 2, 1, false, ACT_MISH, 2>(a, s);
    // knob 7 bit 2: H = 256 (the critic) in 64-row tiles at two workgroups per CU: its Mish emits are VALU-bound, one
    // workgroup's emit can run under the other's MFMAs
    if (d.hidden == 256 && !a.use_ln && ot == 1 && ((g_short_tiles >> 2) & 1))
      return relu ? launch_fwd_cfg<P, 2, 4, 1, false, ACT_RELU, 2>(a, s) : launch_fwd_cfg<P, 2, 4, 1, false, ACT_MISH, 2>(a, s);
  }
#define DPPO_FWD(T, R, O) \
  if (tpw == T && mr == R && ot == O)                                                                                  \
    return a.use_ln ? (relu ? launch_fwd_cfg<P, T, R, O, true, ACT_RELU>(a, s) : launch_fwd_cfg<P, T, R, O, true, ACT_MISH>(a, s)) \
                    : (relu ? launch_fwd_cfg<P, T, R, O, false, ACT_RELU>(a, s) : launch_fwd_cfg<P, T, R, O, false, ACT_MISH>(a, s));
  if constexpr (P::ESIZE == 2) {
    DPPO_FWD(2, 8, 1) DPPO_FWD(2, 8, 4) DPPO_FWD(4, 4, 1) DPPO_FWD(4, 4, 4) DPPO_FWD(8, 2, 1) DPPO_FWD(8, 2, 4)
    DPPO_FWD(2, 8, 8) DPPO_FWD(4, 4, 8) DPPO_FWD(8, 2, 8)
  } else {
    DPPO_FWD(2, 4, 1) DPPO_FWD(2, 4, 4) DPPO_FWD(4, 2, 1) DPPO_FWD(4, 2, 4) DPPO_FWD(8, 1, 1) DPPO_FWD(8, 1, 4)
    DPPO_FWD(2, 4, 8) DPPO_FWD(4, 2, 8) DPPO_FWD(8, 1, 8)
  }
#undef DPPO_FWD
  return -1;
}
template int launch_fused_forward<F32>(const dppo_net_desc&, const FusedFwdArgs&, hipStream_t, const LossArgs*);
template int launch_fused_forward<BF16>(const dppo_net_desc&, const FusedFwdArgs&, hipStream_t, const LossArgs*);

template <class P, int TPW, int MR, bool LN, int ACT, int OCC = 1>
static int launch_bwd_cfg(const FusedBwdArgs& a, hipStream_t s) {
  constexpr int ES = P::ESIZE, MT = 16 * MR, H = 128 * TPW;
  const size_t lds = 2 * (size_t)MT * H * ES + 64 + (LN ? (size_t)LN_WAVES * 2 * MR * 16 * 4 : 0) + SAMPLER_WAVES * 128 * 4;
  if (lds > 160 * 1024 / OCC || a.KpB0 > H || a.KpB0 > 128) return -2;
  static DevLatch attr;
  raise_lds(fused_backward_kernel<P, TPW, MR, LN, ACT, OCC>, attr);
  const int ntiles = (a.M + MT - 1) / MT;
  const bool probe = probe_begin(PROBE_FUSED_BWD, s);
  hipLaunchKernelGGL((fused_backward_kernel<P, TPW, MR, LN, ACT, OCC>), dim3(ntiles < NUM_CUS * OCC ? ntiles : NUM_CUS * OCC),
                     dim3(512), lds, s, a);
  if (probe) probe_end(s, 2.0 * a.M * ((double)a.out_valid * H + 2.0 * a.nb * H * H));
  return 0;
}

static int g_compact = 1;  // tuning knob 25: the one-block kernels skip the padding k-steps of their short layers (CEngine)
void set_fused_compact(int v) { g_compact = v; }
int fused_compact_on() { return g_compact; }
// LDS of the in-kernel-dW0 variant: dz1 image, d_out tile, flags, d_out column sums, xc tile, staging, the dW0 partial
static size_t bwd_one_dw0_lds(int MT, int H) {
  return (size_t)MT * H * 2 + (size_t)MT * 64 + 64 + SAMPLER_WAVES * 32 * 4 + (size_t)MT * 128 + SAMPLER_WAVES * 2048 +
         (size_t)SAMPLER_WAVES * (H / 128) * 2 * 64 * 16;
}
template <class P, int TPW, int MR, int ACT, bool COMPACT, bool FRAG = false, int DW0 = 0>
static int launch_bwd_one_cfg2(const FusedBwdArgs& a, hipStream_t s) {
  constexpr int ES = P::ESIZE, MT = 16 * MR, H = 128 * TPW;
  const size_t lds = DW0 ? bwd_one_dw0_lds(MT, H) : 2 * (size_t)MT * H * ES + 64 + SAMPLER_WAVES * 128 * 4;
  if (lds > 160 * 1024 || a.KpB0 > H || a.KpB0 > 128 || a.nb != 1) return -2;
  if (FRAG != (a.dz1f != nullptr)) return -4;
  if ((DW0 != 0) != (a.dw0_slab != nullptr) || (DW0 == 2) != (a.dw0_round != 0)) return -4;
  if (DW0 && (a.xc == nullptr || a.ld_xc < 64 || a.ld_xc % 8 || a.xc_af % 4 || a.xc_skip % 4 || a.xc_skip < 0 || a.dh[0] != nullptr ||
              32 + a.xc_skip > 64 || a.out_valid > 32))
    return -4;
  if (a.dz1f != nullptr || a.dh0f != nullptr || a.xf != nullptr || a.doutf != nullptr) {  // fragment mode: all four or none
    if (ES != 2 || !a.dz1f || !a.dh0f || !a.xf || !a.doutf || !a.x || a.ld_x % 16 || a.dof_nt < 1 || a.dof_nt * 16 > a.KpB0 ||
        (size_t)MT * ((size_t)a.KpB0 * ES + (size_t)a.ld_x * ES) + (size_t)SAMPLER_WAVES * 32 * 32 * TPW > (size_t)MT * H * ES)
      return -4;
  }
  static DevLatch attr;
  raise_lds(fused_backward_one_kernel<P, TPW, MR, ACT, COMPACT, FRAG, DW0>, attr);
  const int ntiles = (a.M + MT - 1) / MT;
  const bool probe = probe_begin(PROBE_FUSED_BWD, s);
  hipLaunchKernelGGL((fused_backward_one_kernel<P, TPW, MR, ACT, COMPACT, FRAG, DW0>), dim3(ntiles < NUM_CUS ? ntiles : NUM_CUS),
                     dim3(512), lds, s, a);
  if (probe) probe_end(s, 2.0 * a.M * ((double)a.out_valid * H + 2.0 * a.nb * H * H));
  return 0;
}
template <class P, int TPW, int MR, int ACT>
static int launch_bwd_one_cfg(const FusedBwdArgs& a, hipStream_t s) {
  if constexpr (ring_depth<TPW, MR>() == 4) {
    if constexpr (P::ESIZE == 2) {
      if (a.dz1f != nullptr) return g_compact && a.out_valid <= P::KB ? launch_bwd_one_cfg2<P, TPW, MR, ACT, true, true>(a, s) : -4;
      if (a.dw0_slab != nullptr)
        return !(g_compact && a.out_valid <= P::KB) ? -4
               : a.dw0_round                        ? launch_bwd_one_cfg2<P, TPW, MR, ACT, true, false, 2>(a, s)
                                                    : launch_bwd_one_cfg2<P, TPW, MR, ACT, true, false, 1>(a, s);
    }
    if (a.dw0_slab != nullptr) return -4;
    if (g_compact && a.out_valid <= P::KB) return launch_bwd_one_cfg2<P, TPW, MR, ACT, true>(a, s);
  }
  return launch_bwd_one_cfg2<P, TPW, MR, ACT, false>(a, s);
}

template <class P>
int fused_bwd_one_grid(const dppo_net_desc& d, int64_t M) {
  const int mt = 16 * pick_mr<P>(d.hidden);
  if (mt <= 0 || M <= 0) return 0;
  const int64_t ntiles = (M + mt - 1) / mt;
  return (int)(ntiles < NUM_CUS ? ntiles : NUM_CUS);
}
template int fused_bwd_one_grid<F32>(const dppo_net_desc&, int64_t);
template int fused_bwd_one_grid<BF16>(const dppo_net_desc&, int64_t);
// shapes the in-kernel dW0 is built for: bf16 one-block backward on the compact walk with a ring of four (H = 256, 512), its
// LDS within the CU's 160 KB (the caller checks the 32-column limit of its input rows)
bool fused_dw0_shape(const dppo_net_desc& d) {
  if (!g_compact || !fused_bwd_one_block<BF16>(d) || d.out_dim > BF16::KB || (d.hidden != 512 && d.hidden != 256)) return false;
  return bwd_one_dw0_lds(16 * pick_mr<BF16>(d.hidden), d.hidden) <= 160 * 1024;
}
template <class P>
int launch_fused_backward(const dppo_net_desc& d, const FusedBwdArgs& a, hipStream_t s) {
  const bool relu = a.act == ACT_RELU;
  if (a.one_block) {  // (the caller asked fused_bwd_one_block() and sized the column sums for the forward's tile height)
    const int t = d.hidden / 128, q = pick_mr<P>(d.hidden);
    if (a.M <= 0) return -1;
#define DPPO_BWD1(T, R) \
  if (t == T && q == R) return relu ? launch_bwd_one_cfg<P, T, R, ACT_RELU>(a, s) : launch_bwd_one_cfg<P, T, R, ACT_MISH>(a, s);
    if constexpr (P::ESIZE == 2) {
      DPPO_BWD1(2, 8) DPPO_BWD1(4, 4) DPPO_BWD1(8, 2)
    } else {
      DPPO_BWD1(2, 4) DPPO_BWD1(4, 2) DPPO_BWD1(8, 1)
    }
#undef DPPO_BWD1
    return -1;
  }
  const int tpw = d.hidden / 128, mr = pick_mr_bwd<P>(d.hidden, a.use_ln);
  if (mr == 0 || a.M <= 0) return -1;
  if constexpr (P::ESIZE == 2) {
    if (short_tiles<P>(d.hidden, a.use_ln, 0))
      return relu ? launch_bwd_cfg<P, 4, 2, false, ACT_RELU, 2>(a, s) : launch_bwd_cfg<P, 4, 2, false, ACT_MISH, 2>(a, s);
    if (d.hidden == 256 && !a.use_ln && ((g_short_tiles >> 3) & 1))  // knob 7 bit 3: the same for the backward (32-row tiles)
      return relu ? launch_bwd_cfg<P, 2, 2, false, ACT_RELU, 2>(a, s) : launch_bwd_cfg<P, 2, 2, false, ACT_MISH, 2>(a, s);
  }
#define DPPO_BWD(T, R, L) \
  if (tpw == T && mr == R && (a.use_ln != 0) == L) \
    return relu ? launch_bwd_cfg<P, T, R, L, ACT_RELU>(a, s) : launch_bwd_cfg<P, T, R, L, ACT_MISH>(a, s);
  if constexpr (P::ESIZE == 2) {
    DPPO_BWD(2, 4, false) DPPO_BWD(2, 2, true) DPPO_BWD(4, 4, false) DPPO_BWD(4, 2, true) DPPO_BWD(8, 1, false)
    DPPO_BWD(8, 1, true)
  } else {
    DPPO_BWD(2, 4, false) DPPO_BWD(2, 2, true) DPPO_BWD(4, 2, false) DPPO_BWD(4, 1, true) DPPO_BWD(8, 1, false)
    DPPO_BWD(8, 1, true)
  }
#undef DPPO_BWD
  return -1;
}
template int launch_fused_backward<F32>(const dppo_net_desc&, const FusedBwdArgs&, hipStream_t);
template int launch_fused_backward<BF16>(const dppo_net_desc&, const FusedBwdArgs&, hipStream_t);

// fragment packing of a whole stream: blockIdx.y = layer, blockIdx.x = (wave, k-step, tile)
template <class P>
__global__ void pack_stream_kernel(const PackStream d) {
  pack_stream_block<P>(d.layer[blockIdx.y], d.TPW, blockIdx.x);
}
template <class P>
void launch_pack_stream(const PackStream& d, hipStream_t s) {
  int maxks = 0;
  for (int l = 0; l < d.n_layers; ++l) maxks = d.layer[l].KS > maxks ? d.layer[l].KS : maxks;
  hipLaunchKernelGGL((pack_stream_kernel<P>), dim3(SAMPLER_WAVES * maxks * d.TPW, d.n_layers), dim3(64), 0, s, d);
}
template void launch_pack_stream<F32>(const PackStream&, hipStream_t);
template void launch_pack_stream<BF16>(const PackStream&, hipStream_t);

// bias gradients: out[slot][c] = sum over tiles; 64 columns x 16 tile-lanes per block, blockIdx.y = slot
__global__ __launch_bounds__(1024) void reduce_slots_kernel(const float* in, int tiles, int n, const SlotOuts o) {
  __shared__ float red[16][65];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const float* src = in + (size_t)blockIdx.y * tiles * n;
  float p[4] = {0.f, 0.f, 0.f, 0.f};
  if (c < n) {
    int r = rl;
    for (; r + 48 < tiles; r += 64) {
#pragma unroll
      for (int u = 0; u < 4; ++u) p[u] += src[(size_t)(r + 16 * u) * n + c];
    }
    for (; r < tiles; r += 16) p[0] += src[(size_t)r * n + c];
  }
  red[rl][cl] = (p[0] + p[1]) + (p[2] + p[3]);
  __syncthreads();
  if (rl == 0 && c < n) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += red[i][cl];
    if (c < o.n[blockIdx.y]) o.out[blockIdx.y][c] = s;
  }
}
void launch_reduce_slots(const float* in, int tiles, int n, const SlotOuts& o, hipStream_t s) {
  hipLaunchKernelGGL(reduce_slots_kernel, dim3((n + 63) / 64, o.n_slots), dim3(1024), 0, s, in, tiles, n, o);
}

}  // namespace dppo
