// MFMA GEMM kernels for the big-batch side of the DPPO path (log-prob precompute, PPO update).
//
//  gemm_nt : Y[M,N] = epi( X[M,Kp] . W[N,Kp]^T )          forward layers and backward-data
//  gemm_tn : C[N1,N2] (+)= A[M,N1]^T . B[M,N2]             weight gradients, split over M into slabs
//
// Both are written "transposed": the MFMA's 16-row operand (A) is the weight / gradient-feature
// side and its 16-column operand (B) is the batch-row side, so every lane ends up owning ONE batch
// row and a run of 4*TN consecutive features -- epilogue loads/stores are 16-byte vectors and one
// wave instruction covers 16 rows x 128 contiguous bytes.
#pragma once
#include "common.h"
#include "ppo.h"

namespace dppo {

struct GemmNT {
  const void* X;  // [M][ldx] elem (row-major), columns >= Kp never read
  const void* W;  // [N][ldw] elem (nn.Linear layout), rows >= N treated as zero
  const float* bias;  // [N] or null
  int M, N, Kp;       // Kp: multiple of 128 bytes / esize
  int ldx, ldw;
  // epilogue, in this order:  v = acc + bias;  v *= act'(dsrc);  v += res + add;  stores
  const void* dsrc;  // pre-activation the derivative is taken at: f32 (dsrc_kind 1) or elem (2)
  int dsrc_kind, dsrc_ld, dact;
  const float* res;  // f32 addend [M][ldres]
  int ldres;
  const void* add;  // elem addend [M][ldadd]
  int ldadd;
  float* out_f32;  // [M][ldo32]
  int ldo32;
  void* out_pre;  // elem(v)        [M][ldo]
  void* out_act;  // elem(act(v))   [M][ldo]
  int ldo, act;
};

struct GemmTN {
  const void* A;  // [M][lda] elem ; contributes rows of C (N1)
  const void* B;  // [M][ldb] elem ; contributes columns of C (N2)
  int M, N1, N2, lda, ldb;
  float* slab;  // [splits][N1][ldc] partial sums
  int ldc, splits, rows_per_split;  // rows_per_split multiple of 64
  // readable columns of a row when they differ from its stride (0 = lda / ldb): rows may OVERLAP -- an im2col operand
  // that is a window of stride C and width k*C over a channel-last image (unet.hip).  Register-staged kernel only.
  int ncol_a, ncol_b;
  // Folded slab reduction (grouped kernel, red_cnt != null): the workgroup that arrives LAST at an output tile's counter
  // (one per 128 x 128 tile, zero on entry, left zero) sums the tile's `splits` slabs in the fixed order of
  // slab_job_block() and writes the result: columns [0, red_n2a) of the product to red_out (ld red_ldo; transposed if
  // red_transpose), columns [red_n2a, N2) to red_out2 (red_n2a < 0: everything to red_out).  No reduction launch follows.
  unsigned* red_cnt;
  float* red_out;
  float* red_out2;
  int red_ldo, red_ldo2, red_transpose, red_n2a;
};
// Work that only waits for what ran BEFORE the weight-gradient GEMMs rides in their launch as extra workgroups (the first
// ones of the grid): the per-tile column sums of the fused backward reduced over tiles (bias / LayerNorm-parameter
// gradients; 16 columns of one slot per workgroup) and the loss statistics (one workgroup).  With the folded slab
// reduction above nothing is left for a reduction launch behind the GEMMs.
constexpr int TN_MAX_SLOTS = 50;
struct GemmTNExtra {
  int n_blocks;  // extra workgroups in front of the GEMM tiles, a multiple of 8 (0: none)
  int n_slot_blocks, slot_bx;  // slot s, 16-column group x  <->  block s * slot_bx + x
  const float* colsum;         // [slots][tiles][width]
  int tiles, width, n_slots;
  float* slot_out[TN_MAX_SLOTS];
  int slot_n[TN_MAX_SLOTS];
  const double *fin_partial, *fin_moments;  // fin_stats != null: one more block finalises the loss statistics
  double* fin_stats;
  int fin_blocks, fin_part;
  double fin_n_count;
  // More riders (api.hip, knob 40: the in-kernel dW0's backward feeds all of this, not the GEMMs): behind the slot blocks and the
  // statistics block, rjob_blocks[i] blocks of slab job i (post_blocks.h, slab_job_block_wide) -- together the PRODUCERS: with
  // arrive_cnt != null each stores write-through and arrives on it -- then the CONSUMERS: post.n_temb blocks of the time-embedding
  // gradient and post.n_dw0t blocks of dW0's time columns (post.wait_cnt = arrive_cnt, post.wait_need = the producers' count).
  SlabJob rjob[3];
  int n_rjobs, rjob_blocks[3];
  PostReduce post;
  unsigned* arrive_cnt;
};

constexpr int MAX_TN_JOBS = 8;
struct GemmTNGroup {  // one launch over the 128 x 128 tiles of n weight-gradient GEMMs (gemm_tn_group_kernel)
  GemmTN j[MAX_TN_JOBS];
  int base[MAX_TN_JOBS + 1];  // first workgroup of job i (counted behind the extra ones); base[n] + ex.n_blocks = grid size
  int n;
  GemmTNExtra ex;
};

// ---- weight gradients from K-MAJOR FRAGMENT operands (bf16; gemm_tn_frag_kernel) ---------------------------------------------
// The tensors a weight-gradient GEMM contracts over batch rows -- act(h_0), act(z1), dz1, dh_0, the input rows x and d_out --
// are written by the one-block fused kernels (fused.hip, frag_store) not row-major but as the MFMA operand fragments the
// contraction wants: for a tensor of NT feature tiles (16 features each) and k-steps of 32 batch rows,
//     F[ks][ft][lane][8 bf16]      (16 bytes per lane, 1 KB per (k-step, feature tile), k-steps outermost)
// where lane (i = lane & 15, kg = lane >> 4) holds feature 16 ft + i of rows 32 ks + 8 kg + s, s = 0..7 -- exactly the register
// image of a 16x16x32 MFMA operand whose K index is the batch row.  Both operands use the same map, so
// C[16 ft1 + 4 g + e][16 ft2 + r] += mma(F1[ks][ft1], F2[ks][ft2]) with no transpose, no LDS and no barrier: every load is one
// perfectly coalesced 1 KB wave instruction.  Rows past the batch are zero in at least one operand of every product
// (the backward-side tensors are exact zeros there), and the fused kernels write whole tiles, so every k-step below
// ceil(M / 64) * 2 exists.
struct GemmTNFrag {
  const u32x4* A;   // fragments of the tensor that contributes rows of C
  const u32x4* B;   // ... columns of C
  int nta, ntb;     // feature tiles per k-step in A / B (their k-step strides)
  int N1, N2;       // extent of C (rows from A's features, columns from B's)
  int tb;           // B tiles per wave: 1, 2 or 4 (wave tile = 64 x 16 tb)
  int wga, wgb;     // waves of a workgroup along A / B (wga * wgb = 4)
  int dbg;          // timing experiments (knob 34)
  int npf, pfd;     // L2 prefetch: fragments per wave and k-step, k-steps ahead (gemm_tn_frag_prepare)
  int ks_total, ks_per_split, splits;
  float* slab;      // [splits][N1][ldc]
  int ldc;
};
struct GemmTNFragGroup {
  GemmTNFrag j[MAX_TN_JOBS];
  int base[MAX_TN_JOBS + 1];
  int n;
};
int gemm_tn_frag_blocks(const GemmTNFrag& j);  // workgroups of one job (tiles x splits)
bool gemm_tn_frag_prepare(GemmTNFrag& j);      // fills tb, wga, wgb, npf, pfd from N1, N2, nta, ntb; false: shape not covered
void set_gemm_tn_frag_pfd(int v);              // tuning knob 33
void set_gemm_tn_frag_dbg(int v);              // tuning knob 34 (timing experiments only)
void launch_gemm_tn_frag_group(const GemmTNFragGroup& gr, int64_t M, hipStream_t s);
void set_gemm_tn_frag_depth(int v);  // tuning knob 32: 0 (default) LDS-ring kernel; 2..4 register-only kernel, k-steps of lookahead

template <class P>
void launch_gemm_nt(const GemmNT& a, hipStream_t s);
template <class P>
void launch_gemm_tn_group(const GemmTNGroup& gr, hipStream_t s);
void set_gemm_tn_variant(int v);  // tuning knob 5
void set_gemm_tn_nbuf(int v);     // tuning knob 26
void set_gemm_tn_thin(int v);     // tuning knob 6

// Measurement hook (bench.py's roofline): while armed for a kernel id, every launch of that kernel is bracketed by
// HIP events recorded on its launch stream.  Process-wide, not thread-safe, off by default; never armed by the
// product path.
enum { PROBE_GEMM_NT_HIDDEN = 1, PROBE_GEMM_TN = 2, PROBE_FUSED_FWD = 3, PROBE_FUSED_BWD = 4, PROBE_SAMPLER = 5 };
void set_gemm_nt_variant(int v);  // 0 register staging, 1 LDS-DMA staging where legal (default)
void set_gemm_nt_small(int v);    // 1 (default): 64 x 64 / 64 x 32 tiles when 128 x 128 tiles would give < 192 workgroups
int probe_arm(int kernel_id, int max_launches);
int probe_collect(double* total_ms, int* launches, double* flops, double* bytes = nullptr);
bool probe_begin(int kernel_id, hipStream_t s);  // true if this launch is being timed
void probe_end(hipStream_t s, double flops, double bytes = 0);  // right after the launch when probe_begin returned true
template <class P>
void launch_gemm_tn(const GemmTN& a, hipStream_t s);
bool gemm_tn_thin(int N1, int N2);  // true: the 512 x 64 block shape is used (one output tile covers <= 64 columns)

// out[n] (+)= sum_s slab[s][n]  (fixed order => reproducible)
void launch_slab_reduce(const float* slab, int splits, size_t n, float* out, float scale, hipStream_t s);
// colsum[j] = sum_m A[m][j] for j < N ; deterministic two-stage
template <class P>
void launch_colsum(const void* A, int M, int N, int lda, float* partial /*[blocks][N]*/, int blocks, float* out,
                   float scale, hipStream_t s);

}  // namespace dppo
