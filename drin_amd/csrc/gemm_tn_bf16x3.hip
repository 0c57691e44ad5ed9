// Weight-gradient contraction on the bf16 matrix cores, fp32-equivalent ("bf16x3", see gemm_bf16x3.hip):
//   y[n, k] += sum_m a[m, n] * b[m, k]          (dW = dY^T X: both operands are m-major, the reduction index
//                                                is the ROW of both matrices; M = pairs, y = one weight matrix)
//
// The bf16 MFMAs want 8 consecutive values of the reduction index per lane, but in memory consecutive m
// are a whole row apart.  Each thread therefore loads a 4 (m) x 4 (columns) fp32 block - one float4 from
// each of four consecutive rows - transposes it in registers, splits it into bf16 hi / lo and writes four
// 8-byte pieces (4 consecutive m of one column) to LDS.  LDS holds the tile as [column][32 m] rows of 64 B,
// exactly what the MFMA fragments read with ds_read_b128.
//
// LDS layout: the four columns of a thread's block (4 cg + j) would land 256 B apart - the bank period -
// so logical row p lives at physical row (p & 3) * ROWS/4 + (p >> 2): the pieces a wave writes with one
// instruction (j fixed, 8 neighbouring column groups x 8 m-groups) then form 512 contiguous bytes.  The
// 16-byte chunk index is XORed with (p & 3) so that the 16 rows of a fragment read (4 row residues x 4
// physical neighbours) fall on 16 distinct bank quads.
//
// The reduction is split over workgroups (M is 10^4 .. 10^5, the output has 9 .. 24 tiles of 256 x 256).  Each
// slice stores its partial tile plainly into caller scratch and a SliceSum launch (gemm_f32.hip) adds the slices to
// y in order: reproducible bit for bit.  (An atomic epilogue was the first version: 65 536 fp32 atomics per workgroup
// cost ~50 us whatever the slice length - MI355X_MICROARCH.md: one 256-byte atomic wave-instruction per ~50 ns per
// CU - more than the MFMA work at M ~ 10^4, and left the last bits of dW run-order dependent.  It is gone.)
#include <stdlib.h>

#include <atomic>

#include "device_utils.h"
#include "internal.h"

namespace drin {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

namespace x3tn {

constexpr int BK = 32;      // reduction rows per LDS stage
constexpr int TILE = 256;   // output tile: 256 (n) x 256 (k)
constexpr int THREADS = 512;
constexpr int PLANE = TILE * 64;             // one bf16 plane of one operand: 256 rows x 64 B
constexpr int BUF_BYTES = 4 * PLANE;         // A hi, A lo, B hi, B lo
constexpr int LDS_BYTES = 2 * BUF_BYTES;     // double buffered: 128 KiB
// (Measured and not adopted, same box: 128 x 128 tiles with four waves of 64 x 64 and TWO workgroups per CU - a quarter of
//  the slices, half the partial-tile traffic: B = 64 step 0.845 against 0.835 ms of split-bf16 GEMM time, B = 512 4.90
//  against 4.47.)

__device__ __forceinline__ int lds_off(int row, int chunk) {
  return ((((row & 3) * (TILE / 4)) + (row >> 2)) << 6) + ((chunk ^ (row & 3)) << 4);
}

// 32 (m) x 256 (columns) fp32 staged by 512 threads: lane = (m-group mg = lane >> 3, column group cgl = lane & 7),
// wave w covers column groups 8 w .. 8 w + 7; a column group is 4 adjacent columns.
// index != NULL: reduction row m of the operand is row index[m] of a TABLE (the entity tables of table-form training: the
// vertex-encoder inputs are gathered here instead of being materialised per step).  The indices of the NEXT stage are
// fetched one call ahead, so the dependent address -> data chain never sits inside a stage.
struct TransposeStager {
  const float* p;   // first row of the reduction range (or of the table), this thread's column
  int64_t ld;
  int mg, cg;
  float4 v[4];
  const int64_t* index;
  int64_t next_row[4];

  __device__ __forceinline__ void init(const float* __restrict__ src, int64_t ld_, int col0, int ncols,
                                       const int64_t* __restrict__ index_ = nullptr, int64_t m_first = 0, int64_t m_end = 0) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    mg = lane >> 3;
    cg = wave * 8 + (lane & 7);
    int col = col0 + 4 * cg;
    col = col + 4 <= ncols ? col : ncols - 4;  // clamped columns feed output rows / columns that are never stored
    p = src + col;
    ld = ld_;
    index = index_;
    if (index != nullptr) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t m = m_first + 4 * mg + i;
        next_row[i] = index[m < m_end ? m : m_end - 1];
      }
    }
  }
  // rows m0 + 4 mg + i; rows at or past m_end contribute zero (the loads stay unconditional: clamped row, then masked).
  // Calls walk m0 in steps of BK (the pipeline's stage order), which is what the index prefetch relies on.
  __device__ __forceinline__ void load(int64_t m0, int64_t m_end) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t m = m0 + 4 * mg + i;
      const bool live = m < m_end;
      const int64_t row = index != nullptr ? next_row[i] : (live ? m : m_end - 1);
      const float4 x = ld4(p + row * ld);
      v[i] = live ? x : make_float4(0.f, 0.f, 0.f, 0.f);
      if (index != nullptr) {
        const int64_t mn = m + BK;
        next_row[i] = index[mn < m_end ? mn : m_end - 1];
      }
    }
  }
  // the sum over m of this thread's four columns, of everything staged so far (rows past m_end were loaded as zeros)
  __device__ __forceinline__ void add_to(float4& s) const { s = s + ((v[0] + v[1]) + (v[2] + v[3])); }
  template <bool WITH_LO = true>
  __device__ __forceinline__ void store(char* __restrict__ hi_plane, char* __restrict__ lo_plane) const {
    const float col[4][4] = {{v[0].x, v[1].x, v[2].x, v[3].x},
                             {v[0].y, v[1].y, v[2].y, v[3].y},
                             {v[0].z, v[1].z, v[2].z, v[3].z},
                             {v[0].w, v[1].w, v[2].w, v[3].w}};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bf16x4 hi, lo;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        hi[i] = (__bf16)col[j][i];
        if (WITH_LO) lo[i] = (__bf16)(col[j][i] - (float)hi[i]);
      }
      const int off = lds_off(4 * cg + j, mg >> 1) + ((mg & 1) << 3);
      *reinterpret_cast<bf16x4*>(hi_plane + off) = hi;
      if (WITH_LO) *reinterpret_cast<bf16x4*>(lo_plane + off) = lo;
    }
  }
};

// One launch runs a GROUP of such products (TnGroup, internal.h: the pair-sized weight gradients of a backward pass).
// grid: 1-D; a problem's work items are (slice of the reduction, output tile n-tile major), tile fastest; problems in order.
struct Problem {
  const float* a;
  const float* b;
  float* partial;            // [slices][N][K] partial tiles
  const int64_t* b_index;
  int64_t lda, ldb, M, rows_per_slice;
  int N, K, k_tiles, tiles;
  unsigned first;            // first work item
  int slices;
  float* colsum_partial;     // optional: column sums of a (the bias gradient that goes with dW = dY^T X) as [slices][N] rows
};
struct GroupArgs {
  Problem p[TnGroup::MAX];
  int n;
};

// (Measured in round 4, not adopted and since removed - profiles/r4_dw_one_pass.txt: both operands rounded to bf16, ONE MFMA pass;
//  -4.5 % of the B = 64 step for weight gradients 2.5e-3 instead of 1.2e-5 from the fp64 oracle's.)
__global__ void __launch_bounds__(THREADS, 1) k_gemm_tn_bf16x3(const GroupArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // XCD-aware order (as in gemm_x3_planes.hip): workgroups are dealt round-robin over the 8 XCDs, each with its own L2.  All
  // tiles of one slice read the same rows of both operands (fp32: a slice is megabytes), so XCD x takes a contiguous range
  // of the work-item sequence - a slice's rows then come through ONE L2 instead of up to eight.
  unsigned t = blockIdx.x;
  {
    const unsigned total = gridDim.x, xcd = t & 7, k = t >> 3, q = total >> 3, rem = total & 7;
    t = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + k;
  }
  int pi = 0;
  for (int i = 1; i < g.n; ++i) pi = t >= g.p[i].first ? i : pi;
  const Problem& P = g.p[pi];
  const float* __restrict__ A = P.a;
  const float* __restrict__ Bm = P.b;
  float* __restrict__ partial = P.partial;
  const int64_t lda = P.lda, ldb = P.ldb, M = P.M, rows_per_slice = P.rows_per_slice;
  const int N = P.N, K = P.K, k_tiles = P.k_tiles, tiles = P.tiles;
  t -= P.first;
  const unsigned slice = t / (unsigned)tiles, tile = t - slice * (unsigned)tiles;
  const int n0 = (int)(tile / (unsigned)k_tiles) * TILE, k0 = (int)(tile % (unsigned)k_tiles) * TILE;
  const int64_t m_begin = (int64_t)slice * rows_per_slice;
  const int64_t m_end = m_begin + rows_per_slice < M ? m_begin + rows_per_slice : M;
  const int nkb = (int)((m_end - m_begin + BK - 1) / BK);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = wave >> 2, wn = wave & 3;  // 2 x 4 waves, wave tile 128 (n) x 64 (k)
  const int r = lane & 15, c = lane >> 4;
  constexpr int MI = 8, NI = 4;

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int v = 0; v < 4; ++v) acc[i][j][v] = 0.f;

  TransposeStager sa, sb;
  sa.init(A, lda, n0, N);
  sb.init(Bm, ldb, k0, K, P.b_index, m_begin, m_end);
  sa.load(m_begin, m_end);
  sb.load(m_begin, m_end);
  // the workgroups of the first k-tile also sum the columns of a (they stage every row of their n-tile exactly once)
  const bool sums = P.colsum_partial != nullptr && k0 == 0;
  float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
  if (sums) sa.add_to(csum);
  sa.template store<true>(smem, smem + PLANE);
  sb.template store<true>(smem + 2 * PLANE, smem + 3 * PLANE);
  if (nkb > 1) {
    sa.load(m_begin + BK, m_end);
    sb.load(m_begin + BK, m_end);
  }
  __syncthreads();

  bf16x8 bh[NI], bl[NI];
  auto load_b = [&](const char* buf) {
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int off = lds_off(wn * 64 + j * 16 + r, c);
      bh[j] = *reinterpret_cast<const bf16x8*>(buf + 2 * PLANE + off);
      bl[j] = *reinterpret_cast<const bf16x8*>(buf + 3 * PLANE + off);
    }
  };
  auto row_tiles = [&](const char* buf, int i0, int i1) {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      if (i < i0 || i >= i1) continue;
      const int off = lds_off(wm * 128 + i * 16 + r, c);
      const bf16x8 ah = *reinterpret_cast<const bf16x8*>(buf + off);
      const bf16x8 al = *reinterpret_cast<const bf16x8*>(buf + PLANE + off);
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        // the k-side fragment is the FIRST operand: a lane then holds four consecutive k of one n - one 16-byte store
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], al, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[j], ah, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], ah, acc[i][j], 0, 0, 0);
      }
    }
  };

  // Same software pipeline as k_gemm_bf16x3: stage kb in LDS buffer kb & 1, stage kb + 1 in the staging
  // registers; between the two halves of the MFMA work the staged tile goes to the other buffer and the loads
  // of stage kb + 2 are issued.
  for (int kb = 0; kb < nkb; ++kb) {
    const int cur = kb & 1;
    const char* buf = smem + cur * BUF_BYTES;
    char* nb = smem + (cur ^ 1) * BUF_BYTES;
    const bool more = kb + 1 < nkb;
    load_b(buf);
    row_tiles(buf, 0, MI / 2);
    if (more) {
      if (sums) sa.add_to(csum);
      sa.template store<true>(nb, nb + PLANE);
      sb.template store<true>(nb + 2 * PLANE, nb + 3 * PLANE);
      if (kb + 2 < nkb) {
        sa.load(m_begin + (int64_t)(kb + 2) * BK, m_end);
        sb.load(m_begin + (int64_t)(kb + 2) * BK, m_end);
      }
    }
    row_tiles(buf, MI / 2, MI);
    // (Measured in round 4 and left as it is - profiles/r4_train_fusions_ab.txt: a counted wait that keeps the stage kb + 2 loads
    //  in flight across a raw barrier instead of this drain.  Same box, alternating: split-bf16 GEMM time per step 0.673 against
    //  0.668 ms at B = 64, 4.07-4.10 against 4.05-4.08 at B = 512 - the drain is not what this kernel waits for.)
    __syncthreads();
  }

  if (sums) {  // the eight m-groups of a column group sit 8 lanes apart
#pragma unroll
    for (int d = 8; d < 64; d <<= 1) {
      csum.x += __shfl_xor(csum.x, d);
      csum.y += __shfl_xor(csum.y, d);
      csum.z += __shfl_xor(csum.z, d);
      csum.w += __shfl_xor(csum.w, d);
    }
    const int col = n0 + 4 * sa.cg;
    if (sa.mg == 0 && col + 4 <= N) st4(P.colsum_partial + (int64_t)slice * N + col, csum);
  }
  // C / D of a 16 x 16 tile (operands swapped above): n = lane & 15, k = 4 (lane >> 4) + v
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int nrow = n0 + wm * 128 + i * 16 + r;
    if (nrow >= N) continue;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int kcol = k0 + wn * 64 + j * 16 + c * 4;
      if (kcol >= K) continue;  // K % 4 == 0: a group of four is inside or outside as a whole
      st4(partial + ((int64_t)slice * N + nrow) * K + kcol, make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]));
    }
  }
}

}  // namespace x3tn

bool gemm_tn_bf16x3_fits(int64_t lda, int64_t ldb, int64_t M, int N, int K, const void* a, const void* b) {
  return M >= 1024 && N >= 128 && K >= 128 && (N % 4) == 0 && (K % 4) == 0 && (lda % 4) == 0 && (ldb % 4) == 0 &&
         aligned16(a) && aligned16(b);
}

int TnGroup::add(const float* a, int64_t lda, const float* b, int64_t ldb, float* y, int64_t ldy, int64_t M, int N, int K,
                 const int64_t* b_index, float* colsum) {
  if (y == nullptr || M <= 0 || N <= 0 || K <= 0) return DRIN_OK;
  if (!gemm_tn_bf16x3_fits(lda, ldb, M, N, K, a, b)) {
    set_error("gemm_tn_bf16x3: shape M=%lld N=%d K=%d / alignment outside the kernel's contract", (long long)M, N, K);
    return DRIN_E_SHAPE;
  }
  if (n == MAX) {
    set_error("internal: more than %d weight-gradient products in one group", MAX);
    return DRIN_E_SHAPE;
  }
  item[n++] = {a, lda, b, ldb, y, ldy, M, N, K, b_index, colsum};
  return DRIN_OK;
}

bool gemm_tn_bf16x3_scratch_ok(const float* y, int64_t ldy, int N, int K, const float* scratch, size_t scratch_floats) {
  return scratch != nullptr && aligned16(scratch) && aligned16(y) && (ldy % 4) == 0 && scratch_floats >= (size_t)N * K;
}

static int64_t tn_slices_of(const TnGroup::Item& it, int64_t target, int64_t* rows_out) {   // equal slices of one product
  int64_t slices = cdiv(it.M, target);
  const int64_t rows = cdiv(cdiv(it.M, slices), x3tn::BK) * x3tn::BK;
  if (rows_out) *rows_out = rows;
  return cdiv(it.M, rows);
}

// One workgroup per CU in all (the 128 KiB tile buffers allow one per CU at a time): every extra slice adds a pipeline
// fill and a 256 KiB partial tile, which at M ~ 10^4 cost more than the MFMA work itself (measured at M = 12 928:
// 86 slices 162 us, 28 slices 92 us).  So the chip's 256 workgroups are dealt over ALL the
// products of the group in proportion to their work: one target slice length, a whole number of 32-row stages,
// grown until the group fits - the chip and the scratch.  (Five products of one B = 64 backward pass: slices of ~62 stages
// instead of 15, same box 0.835 -> 0.74 ms of split-bf16 GEMM time per step; at B = 512, 58 .. 160 stages per slice
// already, neither better nor worse.)
int64_t tn_group_target(const TnGroup& grp, size_t scratch_floats) {
  int64_t work = 0, longest = 0;
  for (int i = 0; i < grp.n; ++i) {
    const auto& it = grp.item[i];
    work += cdiv(it.N, x3tn::TILE) * cdiv(it.K, x3tn::TILE) * it.M;
    longest = it.M > longest ? it.M : longest;
  }
  int64_t target = cdiv(cdiv(work, 256), x3tn::BK) * x3tn::BK;
  if (target < 4 * x3tn::BK) target = 4 * x3tn::BK;
  for (;; target += x3tn::BK) {
    int64_t wgs = 0;
    size_t part = 0;
    for (int i = 0; i < grp.n; ++i) {
      const auto& it = grp.item[i];
      const int64_t sl = tn_slices_of(it, target, nullptr);
      wgs += cdiv(it.N, x3tn::TILE) * cdiv(it.K, x3tn::TILE) * sl;
      part += (size_t)sl * it.N * ((size_t)it.K + (it.colsum ? 1 : 0));
    }
    if ((wgs <= 256 && part <= scratch_floats) || target >= longest) break;
  }
  return target;
}

int launch_gemm_tn_group(const TnGroup& grp, hipStream_t st, float* scratch, size_t scratch_floats, SliceSum* defer,
                         int64_t target_rows) {
  if (grp.n == 0) return DRIN_OK;
  if (scratch == nullptr || !aligned16(scratch)) {
    set_error("gemm_tn_bf16x3: the slice scratch is NULL or not 16-byte aligned");
    return DRIN_E_WORKSPACE;
  }
  const int64_t target = target_rows > 0 ? target_rows : tn_group_target(grp, scratch_floats);
  int tiles[TnGroup::MAX];
  for (int i = 0; i < grp.n; ++i) tiles[i] = (int)(cdiv(grp.item[i].N, x3tn::TILE) * cdiv(grp.item[i].K, x3tn::TILE));
  auto slices_of = [&](int i, int64_t tgt, int64_t* rows_out) { return tn_slices_of(grp.item[i], tgt, rows_out); };
  // products of one destination next to one another (their slices: segments of one slice-sum entry, added in this order)
  SliceSum local;
  SliceSum& sums = defer != nullptr ? *defer : local;
  x3tn::GroupArgs ga;
  ga.n = grp.n;
  int64_t items = 0;
  size_t part = 0;
  for (int i = 0; i < grp.n; ++i) {
    const auto& it = grp.item[i];
    int64_t rows = 0;
    const int64_t slices = slices_of(i, target, &rows);
    auto& P = ga.p[i];
    P.a = it.a, P.b = it.b, P.b_index = it.b_index;
    P.lda = it.lda, P.ldb = it.ldb, P.M = it.M, P.rows_per_slice = rows;
    P.N = it.N, P.K = it.K, P.k_tiles = (int)cdiv(it.K, x3tn::TILE), P.tiles = tiles[i];
    P.first = (unsigned)items, P.slices = (int)slices;
    P.partial = scratch + part;
    items += slices * tiles[i];
    part += (size_t)slices * it.N * it.K;
    P.colsum_partial = nullptr;
    if (it.colsum != nullptr) {
      P.colsum_partial = scratch + part;
      part += (size_t)slices * it.N;
    }
    if (part > scratch_floats) {
      set_error("gemm_tn_bf16x3: %zu floats of slice scratch, more than %zu needed", scratch_floats, part);
      return DRIN_E_WORKSPACE;
    }
    DRIN_TRY(sums.add(it.y, it.ldy, it.N, it.K, P.partial, (int)slices));
    if (it.colsum != nullptr) DRIN_TRY(sums.add(it.colsum, it.N, 1, it.N, P.colsum_partial, (int)slices));
  }
  if (items > (int64_t)1 << 30) {
    set_error("gemm_tn_bf16x3: %lld work items exceed the grid limit", (long long)items);
    return DRIN_E_SHAPE;
  }
  {
    static DynLdsOptIn opt_in;
    DRIN_TRY(ensure_dynamic_lds(opt_in, reinterpret_cast<const void*>(x3tn::k_gemm_tn_bf16x3), x3tn::LDS_BYTES,
                                "hipFuncSetAttribute(gemm_tn_bf16x3)"));
  }
  {
    KernelTimer timer(DRIN_KC_GEMM_X3, st);
    hipLaunchKernelGGL(x3tn::k_gemm_tn_bf16x3, dim3((unsigned)items), dim3(x3tn::THREADS), x3tn::LDS_BYTES, st, ga);
    DRIN_CHECK_LAUNCH("k_gemm_tn_bf16x3");
  }
  return defer != nullptr ? DRIN_OK : launch_slice_sum(local, st);
}

int launch_gemm_tn_bf16x3(const float* a, int64_t lda, const float* b, int64_t ldb, float* y, int64_t ldy, int64_t M,
                          int N, int K, hipStream_t st, float* scratch, size_t scratch_floats, const int64_t* b_index) {
  if (M <= 0 || N <= 0 || K <= 0) return DRIN_OK;
  TnGroup one;
  DRIN_TRY(one.add(a, lda, b, ldb, y, ldy, M, N, K, b_index));
  return launch_gemm_tn_group(one, st, scratch, scratch_floats);
}

}  // namespace drin
