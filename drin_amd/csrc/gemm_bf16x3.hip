// fp32-equivalent GEMM on the bf16 matrix cores ("bf16x3"): y[m, n] = sum_k x[m, k] w[n, k] + bias[n].
//
// Every fp32 operand is split into two bf16 numbers, v = hi + lo (+ <= 2^-17 |v|), and the product is
// evaluated as hi*hi + hi*lo + lo*hi with fp32 accumulation on v_mfma_f32_16x16x32_bf16: three MFMAs at
// the bf16 rate (16x the fp32 MFMA rate) instead of eight fp32 MFMAs.  bf16 x bf16 products are exact in
// fp32, so the only error is the dropped lo*lo term and the split residue: <= 7e-6 of sum |x||w| per
// output, ~1e-6 on the final scores (75x inside the 1e-4 parity bar; tests/test_gpu_parity.py pins it).
//
// This file: the activation operand x arrives as fp32 and is split ON THE FLY while it is staged
// (global -> VGPR -> split -> LDS); the weight operand is either split the same way or, when drin_prepare
// has already written its bf16 planes, streamed by LDS-DMA (W_PLANES).  gemm_x3_planes.hip is the
// variant for activations that their producer already wrote as planes.
//
// Two tile shapes of one template:
//   256 x 256 x 32, 8 waves (2 x 4, wave tile 128 x 64 = 8 x 4 MFMA tiles of 16 x 16): pair-sized problems.  At the bf16 rate a 128 x 128
//     tile would need > 30 TB/s from L2; 256 x 256 needs ~13 TB/s.  128 KiB LDS, one workgroup per CU.
//   64 x 128 x 32, 4 waves (2 x 2, wave tile 32 x 64): mention-sized problems (a few thousand rows), where
//     the number of workgroups, not the tile efficiency, decides the time.
// LDS rows are 64 B (32 bf16); the 16-byte chunk index is XOR-swizzled per row quad so that every
// ds_read_b128 lane group lands on 16 distinct bank quads (for LDS-DMA the swizzle goes on the source).
#include <stdlib.h>

#include "device_utils.h"
#include "internal.h"

namespace drin {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

namespace x3 {

constexpr int BK = 32;
constexpr unsigned kCUs = 256;  // MI355X: the tail split below only changes how the last round of tiles is dealt

// MFMA shape v_mfma_f32_16x16x32_bf16 (one k-step per K-block; +5 % over 32x32x16 at the clock the chip
// holds, see gemm_x3_planes.hip).  A lane's fragment is row (lane & 15), chunk (lane >> 4); the bank-conflict-
// free chunk swizzle for that access pattern is the row-quad permutation f = (0, 2, 3, 1).
__device__ __forceinline__ int swz_f(int row) { return (0x78 >> (((row >> 2) & 3) << 1)) & 3; }
__device__ __forceinline__ int swz(int row, int c) { return row * 64 + ((c ^ swz_f(row)) << 4); }

__device__ __forceinline__ void split4(float4 v, bf16x4& hi, bf16x4& lo) { split_bf16x4(v, hi, lo); }

// ROWS x 32 floats staged by THREADS threads: thread t loads float4 #(t & 7) of rows (t >> 3) + (THREADS/8) i.
// Loads are unconditional (no exec-masked branches in the K loop): rows past the end are clamped to the
// last row - their products land in output rows / columns that are never stored - and K % 32 == 0.
template <int ROWS, int THREADS>
struct Stager {
  static constexpr int RPP = THREADS / 8;  // rows per pass
  static constexpr int PASSES = (ROWS + RPP - 1) / RPP;   // (a last, partly used pass: its surplus rows are loaded clamped and not stored)
  static constexpr bool RAGGED = (ROWS % RPP) != 0;
  const float* p[PASSES];
  float4 v[PASSES];

  // (index: row m of the operand is row index[m] of a table - table-form training gathers the vertex-encoder inputs here)
  __device__ __forceinline__ void init(const float* __restrict__ src, int64_t ld, int64_t row0, int64_t rows,
                                       const int64_t* __restrict__ index = nullptr) {
    const int t = threadIdx.x, c4 = t & 7, r = t >> 3;
#pragma unroll
    for (int i = 0; i < PASSES; ++i) {
      int64_t row = row0 + (RAGGED && r + RPP * i >= ROWS ? ROWS - 1 : r + RPP * i);
      row = row < rows ? row : rows - 1;
      if (index != nullptr) row = index[row];
      p[i] = src + row * ld + c4 * 4;
    }
  }
  __device__ __forceinline__ void load(int k0) {
#pragma unroll
    for (int i = 0; i < PASSES; ++i) v[i] = ld4(p[i] + k0);
  }
  // registers -> (hi plane, lo plane): float4 #c4 of a row is the 8-byte half (c4 & 1) of chunk c4 >> 1
  template <bool WITH_LO = true>
  __device__ __forceinline__ void store(char* __restrict__ hi_plane, char* __restrict__ lo_plane) const {
    const int t = threadIdx.x, c4 = t & 7, r = t >> 3;
#pragma unroll
    for (int i = 0; i < PASSES; ++i) {
      if (RAGGED && r + RPP * i >= ROWS) continue;
      bf16x4 hi, lo;
      split4(v[i], hi, lo);
      const int off = swz(r + RPP * i, c4 >> 1) + ((c4 & 1) << 3);
      *reinterpret_cast<bf16x4*>(hi_plane + off) = hi;
      if (WITH_LO) *reinterpret_cast<bf16x4*>(lo_plane + off) = lo;
    }
  }
};

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// LDS-DMA of the weight planes: 2 planes x ROWS rows x 64 B = ROWS / 8 pieces of 1 KiB, dealt over the waves.
template <int ROWS, int WAVES>
struct PlaneDma {
  static constexpr int PIECES = ROWS / 8 / WAVES;  // per wave
  const char* src[PIECES];
  int lds_off[PIECES];

  __device__ __forceinline__ void init(const __bf16* hi, const __bf16* lo, int64_t ld, int64_t row0, int64_t rows) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
      const int piece = wave * PIECES + i;  // 0 .. ROWS/8 - 1; first half hi plane, second half lo plane
      const int plane = piece / (ROWS / 16), pr = piece % (ROWS / 16);
      const int row = pr * 16 + (lane >> 2);
      const int chunk = (lane & 3) ^ swz_f(row);
      int64_t g = row0 + row;
      g = g < rows ? g : rows - 1;
      src[i] = reinterpret_cast<const char*>((plane ? lo : hi) + g * ld) + chunk * 16;
      lds_off[i] = plane * ROWS * 64 + pr * 1024;
    }
  }
  template <bool WITH_LO = true>
  __device__ __forceinline__ void issue(char* planes_base, int kb) const {
    const int wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
      if (!WITH_LO && (wave * PIECES + i) >= ROWS / 16) continue;  // second half of the pieces is the lo plane
      __builtin_amdgcn_global_load_lds((gptr_t)(src[i] + (int64_t)kb * (BK * 2)), (lptr_t)(planes_base + lds_off[i]), 16,
                                       0, 0);
    }
  }
};

template <int BM, int BN, int WM, int WN>
struct Cfg {
  static constexpr int THREADS = 64 * WM * WN;
  static constexpr int A_PLANE = BM * 64, B_PLANE = BN * 64;
  static constexpr int BUF_BYTES = 2 * A_PLANE + 2 * B_PLANE;  // A hi, A lo, B hi, B lo
  static constexpr int LDS_BYTES = 2 * BUF_BYTES;
  static constexpr int MI = BM / WM / 16, NI = BN / WN / 16;  // 16 x 16 MFMA tiles per wave
};

// Tail split: the last, partly filled round of workgroups (tiles [full, tiles), fewer than half the CUs) is dealt as
// `ksplit` work items per tile, each over 1 / ksplit of K.  Part 0 stores to C as usual, the others store raw
// accumulators to `tail` ([tile - full][ksplit - 1][BM][BN]) and k_tail_add folds them in afterwards, in order.
// (B = 512 training: 606 tiles on 256 CUs = 2.37 rounds ran as 3; with the 94 tail tiles halved, as 2.5:
//  same-box A/B of the whole step 8.55 -> 8.44 ms.)
template <int BM, int BN, int WM, int WN, bool W_PLANES>
__global__ void __launch_bounds__(64 * WM * WN, (64 * WM * WN) / 256)
    k_gemm_bf16x3(const float* __restrict__ A, int64_t lda, const float* __restrict__ W, const __bf16* __restrict__ w_hi,
                  const __bf16* __restrict__ w_lo, int64_t ldw, const float* __restrict__ bias, float* __restrict__ C,
                  int64_t ldc, int64_t M, int N, int K, int accumulate, unsigned col_tiles, unsigned full, int ksplit,
                  float* __restrict__ tail, const int64_t* __restrict__ a_index) {
  using G = Cfg<BM, BN, WM, WN>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // XCD-aware tile order (as in gemm_x3_planes.hip): workgroups are dealt round-robin over the 8 XCDs, each with
  // its own L2; the column tiles of one row tile stream the same A rows, so XCD x takes a contiguous range of the
  // tile sequence (column index fastest) and A comes from HBM once instead of once per column tile.
  int64_t m0;
  int n0;
  int kpart = 0;
  unsigned tail_slot = 0;
  {
    const unsigned id = blockIdx.x;
    unsigned t;
    if (id < full) {
      const unsigned xcd = id & 7, k = id >> 3, q = full >> 3, rem = full & 7;
      t = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + k;  // measured: -1.4 % on x_i C_i^T
    } else {
      const unsigned u = id - full;
      t = full + u / (unsigned)ksplit;
      kpart = (int)(u % (unsigned)ksplit);
      tail_slot = (u / (unsigned)ksplit) * (unsigned)(ksplit - 1) + (unsigned)(kpart - 1);
    }
    n0 = (int)(t % col_tiles) * BN;
    m0 = (int64_t)(t / col_tiles) * BM;
  }
  if (ksplit > 1 && blockIdx.x >= full) {  // this work item's slice of K
    K /= ksplit;
    const int64_t k0 = (int64_t)kpart * K;
    A += k0;
    if (W_PLANES) {
      w_hi += k0;
      w_lo += k0;
    } else {
      W += k0;
    }
  }
  constexpr int KSTEP = BK;   // floats of K per iteration
  const int nkb = K / KSTEP;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 15, c = lane >> 4;

  f32x4 acc[G::MI][G::NI];
#pragma unroll
  for (int i = 0; i < G::MI; ++i)
#pragma unroll
    for (int j = 0; j < G::NI; ++j)
#pragma unroll
      for (int v = 0; v < 4; ++v) acc[i][j][v] = 0.f;

  Stager<BM, G::THREADS> sa;
  Stager<W_PLANES ? G::THREADS / 8 : BN, G::THREADS> sb;  // one dummy pass when the weights come by DMA
  PlaneDma<BN, WM * WN> dma;
  sa.init(A, lda, m0, M, a_index);
  if (W_PLANES)
    dma.init(w_hi, w_lo, ldw, n0, N);
  else
    sb.init(W, ldw, n0, N);
  auto stage_a = [&](char* buf) { sa.template store<true>(buf, buf + G::A_PLANE); };

  sa.load(0);
  if (W_PLANES)
    dma.template issue<true>(smem + 2 * G::A_PLANE, 0);
  else
    sb.load(0);
  stage_a(smem);
  if (!W_PLANES) sb.store(smem + 2 * G::A_PLANE, smem + 2 * G::A_PLANE + G::B_PLANE);
  if (nkb > 1) {  // tile 1 is in flight while tile 0 is computed
    sa.load(KSTEP);
    if (!W_PLANES) sb.load(KSTEP);
  }
  __syncthreads();

  // Row tiles [i0, i1) of the current buffer: 3 MFMAs per 16 x 16 tile, the whole K-block in one k-step.
  bf16x8 bh[G::NI], bl[G::NI];
  auto load_b = [&](const char* buf) {
#pragma unroll
    for (int j = 0; j < G::NI; ++j) {
      const int off = swz(wn * (BN / WN) + j * 16 + r, c);
      bh[j] = *reinterpret_cast<const bf16x8*>(buf + 2 * G::A_PLANE + off);
      bl[j] = *reinterpret_cast<const bf16x8*>(buf + 2 * G::A_PLANE + G::B_PLANE + off);
    }
  };
  auto row_tiles = [&](const char* buf, int i0, int i1) {
#pragma unroll
    for (int i = 0; i < G::MI; ++i) {
      if (i < i0 || i >= i1) continue;
      const int off = swz(wm * (BM / WM) + i * 16 + r, c);
      const bf16x8 ah = *reinterpret_cast<const bf16x8*>(buf + off);
      const bf16x8 al = *reinterpret_cast<const bf16x8*>(buf + G::A_PLANE + off);
#pragma unroll
      for (int j = 0; j < G::NI; ++j) {
        // the WEIGHT fragment is the first operand: a lane's four accumulator registers of a 16 x 16 tile are then four
        // consecutive output COLUMNS of one row (row lane & 15, columns 4 (lane >> 4) + v) - the epilogue stores 16 bytes
        // per lane instead of four scattered dwords (as in gemm_x3_planes.hip, where it took the epilogue from 37 k to 24 k cycles)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], al, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[j], ah, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], ah, acc[i][j], 0, 0, 0);
      }
    }
  };

  // Steady state of iteration kb: tile kb is in LDS buffer kb & 1, tile kb+1 is in the staging registers
  // (its loads were issued one full iteration ago).  Between the two halves of the row tiles the staged
  // tile is split and written to the other buffer and the loads of tile kb+2 are issued into the same
  // registers, so every global load has a whole iteration of MFMA work to land under.  Weight planes
  // (W_PLANES) go straight to the other buffer by LDS-DMA at the top of the iteration.
  for (int kb = 0; kb < nkb; ++kb) {
    const int cur = kb & 1;
    const char* buf = smem + cur * G::BUF_BYTES;
    char* nb = smem + (cur ^ 1) * G::BUF_BYTES;
    const bool more = kb + 1 < nkb;
#ifndef DRIN_ABLATE_NO_LOADS   // timing ablations only (wrong results): the K-loop without its global traffic / without its MFMAs
    if (W_PLANES && more) dma.template issue<true>(nb + 2 * G::A_PLANE, kb + 1);
#endif
    load_b(buf);
#ifndef DRIN_ABLATE_NO_MFMA
    row_tiles(buf, 0, G::MI / 2);
#endif
    if (more) {
      stage_a(nb);
      if (!W_PLANES) sb.store(nb + 2 * G::A_PLANE, nb + 2 * G::A_PLANE + G::B_PLANE);
#ifndef DRIN_ABLATE_NO_LOADS
      if (kb + 2 < nkb) {
        sa.load((kb + 2) * KSTEP);
        if (!W_PLANES) sb.load((kb + 2) * KSTEP);
      }
#endif
    }
#ifndef DRIN_ABLATE_NO_MFMA
    row_tiles(buf, G::MI / 2, G::MI);
#else
    row_tiles(buf, 0, 1);
#endif
    // (a counted wait that keeps the register loads of block kb + 2 outstanding across a raw barrier measured inside 1 % of this
    //  either way for the three-pass product - profiles/r4_one_pass_ab.txt)
    __syncthreads();
  }

  if (kpart > 0) {  // raw partial tile into the tail scratch
    float* part = tail + (size_t)tail_slot * (BM * BN);
#pragma unroll
    for (int i = 0; i < G::MI; ++i)
#pragma unroll
      for (int j = 0; j < G::NI; ++j)
        st4(part + (wm * (BM / WM) + i * 16 + r) * BN + wn * (BN / WN) + j * 16 + c * 4,
            make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]));
    return;
  }
  const bool vec_ok = (ldc % 4) == 0 && (reinterpret_cast<uintptr_t>(C) & 15u) == 0;
#pragma unroll
  for (int i = 0; i < G::MI; ++i) {
    const int64_t row = m0 + wm * (BM / WM) + i * 16 + r;  // C/D of a 16 x 16 tile: row lane & 15, columns 4 (lane >> 4) + v
    if (row >= M) continue;
#pragma unroll
    for (int j = 0; j < G::NI; ++j) {
      const int col = n0 + wn * (BN / WN) + j * 16 + c * 4;
      float* dst = C + row * ldc + col;
      if (vec_ok && col + 3 < N) {
        float4 o = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        if (bias != nullptr) o = o + ld4(bias + col);
        if (accumulate) o = o + ld4(dst);
        st4(dst, o);
      } else {
#pragma unroll
        for (int v = 0; v < 4; ++v)
          if (col + v < N) dst[v] = acc[i][j][v] + (bias != nullptr ? bias[col + v] : 0.f) + (accumulate ? dst[v] : 0.f);
      }
    }
  }
}

// C[tile] += sum of its ksplit - 1 partial tiles, in order.  Grid (BM rows, tail tiles), BN threads.
template <int BM, int BN>
__global__ void __launch_bounds__(BN) k_tail_add(const float* __restrict__ tail, float* __restrict__ C, int64_t ldc,
                                                 int64_t M, int N, unsigned col_tiles, unsigned full, int ksplit) {
  const unsigned t = full + blockIdx.y;
  const int col = (int)(t % col_tiles) * BN + threadIdx.x;
  const int64_t row = (int64_t)(t / col_tiles) * BM + blockIdx.x;
  if (row >= M || col >= N) return;
  const float* part = tail + (size_t)blockIdx.y * (ksplit - 1) * (BM * BN) + (size_t)blockIdx.x * BN + threadIdx.x;
  float s = C[row * ldc + col];
  for (int q = 0; q < ksplit - 1; ++q) s += part[(size_t)q * (BM * BN)];
  C[row * ldc + col] = s;
}

template <int BM, int BN, int WM, int WN, bool W_PLANES>
static int launch(const float* x, int64_t ldx, const float* w, const void* w_hi, const void* w_lo, int64_t ldw,
                  const float* bias, float* y, int64_t ldy, int64_t M, int N, int K, hipStream_t st, bool accumulate,
                  float* tail = nullptr, size_t tail_floats = 0, const int64_t* a_index = nullptr) {
  using G = Cfg<BM, BN, WM, WN>;
  const int64_t mt = cdiv(M, BM), nt = cdiv(N, BN);
  if (mt * nt > (int64_t)1 << 30) {
    set_error("gemm_bf16x3: %lld tiles exceed the grid limit; split the batch", (long long)(mt * nt));
    return DRIN_E_SHAPE;
  }
  auto kern = k_gemm_bf16x3<BM, BN, WM, WN, W_PLANES>;
  static DynLdsOptIn opt_in;  // one per template instantiation
  DRIN_TRY(ensure_dynamic_lds(opt_in, reinterpret_cast<const void*>(kern), G::LDS_BYTES, "hipFuncSetAttribute(gemm_bf16x3)"));
  // tail split (one workgroup per CU tiles only): the last round holds `frac` tiles; split K so that it fills the chip
  const unsigned tiles = (unsigned)(mt * nt);
  unsigned full = tiles;
  int ksplit = 1;
  if (BM == 256 && tail != nullptr && tiles <= 16 * kCUs) {
    const unsigned frac = tiles % kCUs;
    const int nkb = K / BK;
    for (int s = 4; s >= 2 && frac > 0; --s)
      if ((unsigned)s * frac <= kCUs && nkb % s == 0 && nkb / s >= 4 &&
          (size_t)frac * (s - 1) * (BM * BN) <= tail_floats) {
        ksplit = s;
        full = tiles - frac;
        break;
      }
  }
  if (BM == 64 && tail != nullptr && tiles <= 128) {
    // a handful of mentions (101 rows x 768 x 2048: 12 tiles walking 64 K-blocks each, 75 us): every tile splits K
    const int nkb = K / BK;
    for (int s = 16; s >= 2; s >>= 1)
      if ((unsigned)s * tiles <= 2 * kCUs && nkb % s == 0 && nkb / s >= 4 &&
          (size_t)tiles * (s - 1) * (BM * BN) <= tail_floats) {
        ksplit = s;
        full = 0;
        break;
      }
  }
  const unsigned items = full + (tiles - full) * (unsigned)ksplit;
  KernelTimer timer(DRIN_KC_GEMM_X3, st);
  hipLaunchKernelGGL(kern, dim3(items), dim3(G::THREADS), G::LDS_BYTES, st, x, ldx, w, (const __bf16*)w_hi,
                     (const __bf16*)w_lo, ldw, bias, y, ldy, M, N, K, accumulate ? 1 : 0, (unsigned)nt, full, ksplit, tail, a_index);
  DRIN_CHECK_LAUNCH("k_gemm_bf16x3");
  if (ksplit > 1) {
    hipLaunchKernelGGL((k_tail_add<BM, BN>), dim3(BM, tiles - full), dim3(BN), 0, st, tail, y, ldy, M, N, (unsigned)nt,
                       full, ksplit);
    DRIN_CHECK_LAUNCH("k_tail_add");
  }
  return DRIN_OK;
}


}  // namespace x3

int launch_tail_add_256(const float* tail, float* y, int64_t ldy, int64_t M, int N, unsigned col_tiles, unsigned full,
                        unsigned tail_tiles, int ksplit, hipStream_t st) {
  KernelTimer timer(DRIN_KC_GEMM_PLANES, st);
  hipLaunchKernelGGL((x3::k_tail_add<256, 256>), dim3(256, tail_tiles), dim3(256), 0, st, tail, y, ldy, M, N, col_tiles, full,
                     ksplit);
  DRIN_CHECK_LAUNCH("k_tail_add");
  return DRIN_OK;
}

// ---- one fp16 plane of a weight matrix under ONE power-of-two scale (DRIN_PREC_BF16X3_IF16: the folded W_h1 W_ei) ----------------
// out = fp16(x / s) with s = the power of two that brings max |x| into [0.5, 1] (clamped to 2^+-126: its reciprocal is a normal
// number), written to scale[0]; the contraction's epilogue multiplies it back.  Exact to apply; fp16's range (6e-5 .. 65 504) then
// never matters whatever the magnitude of the weights (an unscaled plane of weights ~1e-5 is mostly subnormal).
// The maximum is order-independent, so the integer atomic below gives the same bits every run.
__global__ void __launch_bounds__(256) k_abs_max_bits(const float* __restrict__ x, int64_t n4, unsigned* __restrict__ out_bits) {
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 v = ld4(x + i * 4);
    const float a = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
    m = fmaxf(m, a);   // (a NaN weight is not seen here: it reaches the scores through its own fp16 NaN)
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) atomicMax(out_bits, __builtin_bit_cast(unsigned, m));   // |x| >= 0: the bit patterns order like the values
}

__global__ void __launch_bounds__(256) k_to_f16_scaled(const float* __restrict__ x, _Float16* __restrict__ out, int64_t n4,
                                                       float* __restrict__ scale) {
  const float s = cache_field_scale(scale[1]);   // (an infinite maximum: scale 1, the infinity propagates as it is)
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i == 0) scale[0] = s;
  if (i >= n4) return;
  const float inv = 1.0f / s;
  const float4 v = ld4(x + i * 4);
  f16x4 h;
  h[0] = (_Float16)(v.x * inv), h[1] = (_Float16)(v.y * inv), h[2] = (_Float16)(v.z * inv), h[3] = (_Float16)(v.w * inv);
  *reinterpret_cast<f16x4*>(out + i * 4) = h;
}

// scale: two floats - [0] receives the scale, [1] is scratch (the running maximum's bits)
int launch_to_f16_scaled(const float* x, void* out, int64_t n, float* scale, hipStream_t st) {
  if (n <= 0) return DRIN_OK;
  if ((n % 4) || !aligned16(x) || (reinterpret_cast<uintptr_t>(out) & 7u) || scale == nullptr) {
    set_error("to_f16_scaled: %lld elements (multiple of 4) from a 16-byte aligned source, and a two-float scale buffer", (long long)n);
    return DRIN_E_SHAPE;
  }
  hipError_t e = hipMemsetAsync(scale, 0, 2 * sizeof(float), st);
  if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(to_f16_scaled)");
  KernelTimer timer(DRIN_KC_GCN, st);
  const int64_t n4 = n / 4;
  hipLaunchKernelGGL(k_abs_max_bits, dim3((unsigned)(cdiv(n4, 256) < 1024 ? cdiv(n4, 256) : 1024)), dim3(256), 0, st, x, n4,
                     reinterpret_cast<unsigned*>(scale + 1));
  DRIN_CHECK_LAUNCH("k_abs_max_bits");
  hipLaunchKernelGGL(k_to_f16_scaled, dim3((unsigned)cdiv(n4, 256)), dim3(256), 0, st, x, (_Float16*)out, n4, scale);
  DRIN_CHECK_LAUNCH("k_to_f16_scaled");
  return DRIN_OK;
}

// w_hi / w_lo: optional pre-split bf16 planes of w (same row stride); when given, w itself is not read.
int launch_gemm_nt_bf16x3(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, float* y,
                          int64_t ldy, int64_t M, int N, int K, hipStream_t st, const void* w_hi, const void* w_lo,
                          bool accumulate, float* tail, size_t tail_floats, const int64_t* a_index) {
  if (M <= 0 || N <= 0) return DRIN_OK;
  if (a_index != nullptr && ((K % x3::BK) || K <= 0)) {
    set_error("gemm_bf16x3: indexed rows need K %% 32 == 0 (got %d) on the split-bf16 kernel", K);
    return DRIN_E_UNSUPPORTED;
  }
  if ((K % x3::BK) || K <= 0) {  // odd reduction lengths take the exact fp32 kernel (guarded loads)
    if (!w) {
      set_error("gemm_bf16x3: K=%d is not a multiple of 32 and no fp32 weights were given", K);
      return DRIN_E_SHAPE;
    }
    return launch_gemm_nt(x, ldx, w, ldw, bias, y, ldy, M, N, K, accumulate, DRIN_PREC_F32, st);
  }
  const bool planes = w_hi != nullptr && w_lo != nullptr;
  if ((ldx % 4) || !aligned16(x) ||
      (planes ? ((ldw % 8) || !aligned16(w_hi) || !aligned16(w_lo)) : ((ldw % 4) || !aligned16(w)))) {
    set_error("gemm_bf16x3: operands must be 16-byte aligned with leading dimensions multiples of 4 (fp32) / 8 (bf16)");
    return DRIN_E_ALIGN;
  }
  // pair-sized problems: 256 x 256 tiles; anything that would not fill the chip with them: 64 x 128 tiles
  const bool big = cdiv(M, 256) * cdiv(N, 256) >= 192;
  if (tail != nullptr && !aligned16(tail)) tail = nullptr;
  // Mid-sized products (a few thousand to a few ten thousand rows: the training step at the reference's batch): 64 x 128 tiles
  // fill the chip but run at the L2 -> LDS bandwidth so small a tile needs (SQ counters: an MFMA executing in 0.30 of the
  // cycles, waves parked 0.51 of their time - profiles/r3_mfma_pmc.json).  Where a BM x 256 tile of eight waves comes to at
  // least three quarters of a whole number of rounds of the chip's 256 CUs, it is taken instead - BM in {96, 128, 160, 192},
  // the one with the least rounds x rows: 6 464 rows -> 96 (204 tiles), 12 928 -> 160 (243), 25 856 -> 160 (486).  Same
  // box, the B = 64 step's split-bf16 GEMM time 0.730 -> 0.679 ms, B = 128 1.360 -> 1.249 (profiles/r3_tile_shapes_ab.txt).
  // Measured there, NOT adopted and since removed: the same tiles at a fixed BM for every product (0.726 - 0.855 ms), and three
  // stream-K forms (every CU the same number of K-blocks, but below one round every tile is cut and takes a scratch + fix-up
  // path: 64 x 128 0.859, 128 x 256 0.781, 256 x 256 0.955 ms).
  if (!big && M >= 2048 && planes) {
    int best = 0;
    int64_t cost = 0;
    for (int cand : {192, 160, 128, 96}) {
      const int64_t tiles = cdiv(M, cand) * cdiv(N, 256), rounds = cdiv(tiles, x3::kCUs);
      const bool filled = 4 * tiles >= 3 * rounds * (int64_t)x3::kCUs;
      if (filled && (best == 0 || rounds * cand <= cost)) cost = rounds * cand, best = cand;
    }
    switch (best) {
      case 96: return x3::launch<96, 256, 2, 4, true>(x, ldx, w, w_hi, w_lo, ldw, bias, y, ldy, M, N, K, st, accumulate, nullptr, 0, a_index);
      case 128: return x3::launch<128, 256, 2, 4, true>(x, ldx, w, w_hi, w_lo, ldw, bias, y, ldy, M, N, K, st, accumulate, nullptr, 0, a_index);
      case 160: return x3::launch<160, 256, 2, 4, true>(x, ldx, w, w_hi, w_lo, ldw, bias, y, ldy, M, N, K, st, accumulate, nullptr, 0, a_index);
      case 192: return x3::launch<192, 256, 2, 4, true>(x, ldx, w, w_hi, w_lo, ldw, bias, y, ldy, M, N, K, st, accumulate, nullptr, 0, a_index);
      default: break;   // no tall tile fills its rounds: the 64 x 128 tiles below
    }
  }
  // big products on weight planes: the four-phase pipeline of gemm_x3_planes.hip (indexed rows and odd layouts: the two-phase kernel below)
  if (big && planes && a_index == nullptr && (N % 4) == 0 && (ldy % 4) == 0 && aligned16(y) && (bias == nullptr || aligned16(bias)))
    return launch_gemm_nt_bf16x3_p4(x, ldx, w_hi, w_lo, ldw, bias, y, ldy, M, N, K, st, accumulate, tail, tail_floats);
  if (big)
    return planes ? x3::launch<256, 256, 2, 4, true>(x, ldx, w, w_hi, w_lo, ldw, bias, y, ldy, M, N, K, st, accumulate, tail, tail_floats, a_index)
                  : x3::launch<256, 256, 2, 4, false>(x, ldx, w, w_hi, w_lo, ldw, bias, y, ldy, M, N, K, st, accumulate, tail, tail_floats, a_index);
  // (measured again in round 2, same box: 128 x 128 tiles with four waves of 64 x 64, two workgroups per CU, for 2 048 <= M <
  //  16 384 - the B = 64 training step's split-bf16 GEMMs 0.745 -> 0.827 ms, a 64-mention scoring call 0.100 -> 0.116: the
  //  three co-resident workgroups of the small tile hide each other's stage barriers better than the larger wave tile saves)
  return planes ? x3::launch<64, 128, 2, 2, true>(x, ldx, w, w_hi, w_lo, ldw, bias, y, ldy, M, N, K, st, accumulate, tail, tail_floats, a_index)
                : x3::launch<64, 128, 2, 2, false>(x, ldx, w, w_hi, w_lo, ldw, bias, y, ldy, M, N, K, st, accumulate, tail, tail_floats, a_index);
}

}  // namespace drin
