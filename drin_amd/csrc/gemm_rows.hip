// Full-row split-bf16 contractions: a workgroup owns 128 rows x ALL 768 output columns of y = x W^T, so that its epilogue sees
// whole rows and the LayerNorm / GELU / cosine that follow the contraction in the reference (model.py:128, 207-209) run on the
// accumulators - the [M, 768] fp32 product never goes to HBM and the row kernel that re-read it is gone.
//
// Why (round 6, profiles/r6_fusion_bound_ablation.txt): with 256 x 256 tiles a 768-wide row spans three workgroups, so the product was
// stored (1.27 GB per contraction; 0.21 ms of the 1.18 ms of a K = 768 product is its store burst) and re-read by a row kernel
// that is vector-issue bound (k_pair_final: the vector pipe busy 100 % of its cycles - profiles/r6_stream_sq.json).
//
// Tile 96 x 768 x 32, 512 threads = 8 waves as 2 (rows) x 4 (columns): a wave accumulates 48 rows x 192 columns = 3 x 12 MFMA tiles of
// 16 x 16 = 144 accumulator registers (two waves per SIMD, 256 registers each; 128 rows per tile - 192 accumulators - was built first
// and spilled 41 registers in the K-loop alone).  The weight block of a K-step is 768 rows x 64 B x 2 planes = 96 KiB - it cannot be
// double-buffered in 160 KiB of LDS - so it moves as FOUR units of 24 KiB (192 weight rows, both planes) through a ring of four
// slots, two units ahead of their use; the activation block (one 16 KiB unit: 128 rows, of which the tile uses 96) has two buffers
// of its own.  A K-step is four phases; phase j reads the fragments of weight unit j (and, j = 0, the activation fragments, which
// stay in registers for the whole K-step), issues the unit that is due two phases later by LDS-DMA, waits - a counted s_waitcnt -
// for what the NEXT phase reads, and runs 27 MFMAs between two raw barriers; the two wave groups run one barrier apart, so that on
// every SIMD one wave issues MFMAs while the other reads LDS (the scheme of k_gemm_x3_planes_p4, gemm_x3_planes.hip).
//   unit h = 4 kb + j is read in phase (kb, j) from slot j and was issued in phase h - 2 into the slot last read in phase h - 4:
//   two phases before the issue (the distance the delayed wave group needs).  Per wave a weight unit is three DMA instructions, the
//   activation unit two (issued at j = 0, before that phase's weight unit); before the first barrier of phase (kb, j) everything
//   but the weight unit just issued (and, j = 0, the activation unit) must have landed: vmcnt(3), vmcnt(5) at j = 0.
// Accumulation order per output element: K-steps in order, per K-step hi x lo, lo x hi, hi x hi - the order of the 256 x 256 kernels:
// the contraction's values are theirs bit for bit; what differs from the two-kernel form is the association of the row sums.
#include <algorithm>

#include "device_utils.h"
#include "fused.h"
#include "internal.h"
#include "row_ops.h"

namespace drin {
namespace rows {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int BM = 96, BN = 768, BK = 32, THREADS = 512;
constexpr int A_ROWS = 128;               // rows of an activation unit in LDS (the tile computes the first BM of them)
constexpr int A_PLANE = A_ROWS * 64, A_UNIT = 2 * A_PLANE;   // 16 KiB: two planes x 128 rows x 64 B
constexpr int B_ROWS = 192;               // weight rows per unit (four units per K-step)
constexpr int B_PLANE = B_ROWS * 64, B_UNIT = 2 * B_PLANE;   // 24 KiB
constexpr int A_OFF = 0;                  // two activation buffers
constexpr int B_OFF = 2 * A_UNIT;         // four weight slots
constexpr int LDS_LOOP = 2 * A_UNIT + 4 * B_UNIT;   // 128 KiB
constexpr int MAX_SEG = 3;                // mentions a 96-row tile may touch (candidate lists of >= 48)

__device__ __forceinline__ int swz16(int row, int c) {
  const int f = (0x78 >> (((row >> 2) & 3) << 1)) & 3;
  return row * 64 + ((c ^ f) << 4);
}

// epilogue scratch (re-uses the K-loop's LDS once the loop has drained), in floats
constexpr int E_VEC = 0;                         // [3 MAX_SEG + 3][768]: per segment W_h2 mt', W_h2 mi', mt''; then b_h2, gamma, beta
constexpr int E_ROW = (3 * MAX_SEG + 3) * BN;    // [3][BM]: e1_tt, e1_it, segment (as int)
constexpr int E_RED = E_ROW + 3 * BM;            // [4][BM][4]: row sums per column group of waves
constexpr int E_XX = E_RED + 4 * BM * 4;         // [MAX_SEG]: |mt''|^2
constexpr int E_FLOATS = E_XX + 4;
static_assert(E_FLOATS * 4 <= LDS_LOOP, "epilogue scratch must fit the loop's LDS");

struct FinalRowsArgs {
  const __bf16 *a_hi, *a_lo;   // et' planes [M, K]
  const __bf16 *b_hi, *b_lo;   // W_h2 planes [768, K]
  int64_t lda, ldb;
  const float* hm2;            // [2][B][768]: W_h2 mt', W_h2 mi' (no bias)
  const float* b_h2;
  const float* gamma;
  const float* beta;
  const float* e1m;            // [4][M] layer-2 edges
  const float* mt2;            // [B][768]
  float* scores;               // [M]
  int64_t M;
  int B, N, K;
  float ln_eps, cos_eps;
};

// y = gelu(LN(x W_h2^T + e1_tt (W_h2 mt') + e1_it (W_h2 mi') + b_h2)), score = cos(mt'', y)     (model.py:128 for et'', :207-209)
__global__ void __launch_bounds__(THREADS) k_rows_final(const FinalRowsArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int64_t m0 = (int64_t)blockIdx.x * BM;
  const int nkb = a.K / BK;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = wave >> 2, wn = wave & 3;
  const int r = lane & 15, c = lane >> 4;

  // this lane's DMA sources.  Activation unit: 16 pieces of 16 rows x 64 B (plane wave >> 2, rows 32 (wave & 3) + 16 i + (lane >> 2),
  // i < 2); weight unit: 24 pieces (plane wave >> 2, rows 48 (wave & 3) + 16 i + (lane >> 2), i < 3).  A lane fills physical chunk
  // lane & 3 of its row with the logical chunk the fragment reads' swizzle expects there.
  const char* a_src[2];
  const char* b_src[3];
  {
    const int f = (0x78 >> (((lane >> 4) & 3) << 1)) & 3;
    const int chunk = (lane & 3) ^ f;
    const bool lo = wave >= 4;
    const char* ap = reinterpret_cast<const char*>(lo ? a.a_lo : a.a_hi) + chunk * 16;
    const char* bp = reinterpret_cast<const char*>(lo ? a.b_lo : a.b_hi) + chunk * 16;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int64_t g = m0 + (wave & 3) * 32 + 16 * i + (lane >> 2);
      g = g < a.M ? g : a.M - 1;   // rows past the end (and rows 96 .. 127 of the unit: the next tile's) are loaded, never used
      a_src[i] = ap + g * a.lda * 2;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) b_src[i] = bp + (int64_t)((wave & 3) * 48 + 16 * i + (lane >> 2)) * a.ldb * 2;
  }
  const int64_t b_unit_stride = (int64_t)B_ROWS * a.ldb * 2;   // bytes between weight units
  auto issue_a = [&](int buf, int kb) {
    char* dst = smem + A_OFF + buf * A_UNIT + (wave >> 2) * A_PLANE + (wave & 3) * 2048;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(a_src[i] + (int64_t)kb * (BK * 2)), (lptr_t)(dst + i * 1024), 16, 0, 0);
  };
  auto issue_b = [&](int slot, int u, int kb) {
    char* dst = smem + B_OFF + slot * B_UNIT + (wave >> 2) * B_PLANE + (wave & 3) * 3072;
#pragma unroll
    for (int i = 0; i < 3; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(b_src[i] + u * b_unit_stride + (int64_t)kb * (BK * 2)), (lptr_t)(dst + i * 1024), 16, 0, 0);
  };

  f32x4 acc[4][3][3];   // [weight unit][row tile][column tile]
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[u][i][j][v] = 0.f;

  bf16x8 ah[3], al[3], bh[3], bl[3];
  auto read_a = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const char* p = smem + A_OFF + buf * A_UNIT + swz16(wm * 48 + i * 16 + r, c);
      ah[i] = *reinterpret_cast<const bf16x8*>(p);
      al[i] = *reinterpret_cast<const bf16x8*>(p + A_PLANE);
    }
  };
  auto read_b = [&](int slot) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const char* p = smem + B_OFF + slot * B_UNIT + swz16(wn * 48 + j * 16 + r, c);
      bh[j] = *reinterpret_cast<const bf16x8*>(p);
      bl[j] = *reinterpret_cast<const bf16x8*>(p + B_PLANE);
    }
  };
  auto mma = [&](f32x4 (&cc)[3][3]) {
    __builtin_amdgcn_s_setprio(1);
    // term-major over the unit's nine tiles (an accumulator's next MFMA is nine issues away); per accumulator hi lo, lo hi, hi hi
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) cc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], al[i], cc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) cc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[j], ah[i], cc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) cc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], ah[i], cc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
  auto barrier = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  // prologue: the activation unit and weight units 0, 1 of K-step 0, landed and published
  issue_a(0, 0);
  issue_b(0, 0, 0);
  issue_b(1, 1, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  barrier();
  if (wm == 1) barrier();   // the second wave group runs one barrier behind the first from here on

  // phase J of K-step kb: unit 4 kb + J + 2 leaves (J < 2: weight unit J + 2 of this K-step, else unit J - 2 of the next one; the last
  // K-step fetches its own units again - same straight-line code, same counted waits, the data rewritten is identical)
#define DRIN_ROWS_PHASE(J, VM)                                                                     \
  {                                                                                                \
    if (J == 0) read_a(kb & 1);                                                                    \
    read_b(J);                                                                                     \
    if (J == 0) issue_a((kb + 1) & 1, kn);                                                         \
    issue_b((J + 2) % 4, (J + 2) % 4, J < 2 ? kb : kn);                                            \
    asm volatile("s_waitcnt vmcnt(" #VM ")" ::: "memory");                                         \
    barrier();                                                                                     \
    mma(acc[J]);                                                                                   \
    barrier();                                                                                     \
  }
  for (int kb = 0; kb < nkb; ++kb) {
    const int kn = kb + 1 < nkb ? kb + 1 : kb;
    DRIN_ROWS_PHASE(0, 5)
    DRIN_ROWS_PHASE(1, 3)
    DRIN_ROWS_PHASE(2, 3)
    DRIN_ROWS_PHASE(3, 3)
  }
#undef DRIN_ROWS_PHASE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the redundant units of the last K-step
  if (wm == 0) barrier();                             // pairs with the delayed group's last barrier
  __syncthreads();                                    // every fragment read is done: the LDS is the epilogue's from here

#ifdef DRIN_ROWS_STUB_EPILOGUE   // timing probe only (wrong scores): the K-loop and one store per row, nothing else
  {
    float t = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) t += (acc[u][i][j][0] + acc[u][i][j][1]) + (acc[u][i][j][2] + acc[u][i][j][3]);
    if (m0 + (threadIdx.x % BM) < a.M) a.scores[m0 + (threadIdx.x % BM)] = t;
    return;
  }
#endif
  // ---- epilogue: whole rows ------------------------------------------------------------------------------------------------
  float* const ef = reinterpret_cast<float*>(smem);
  float* const l_vec = ef + E_VEC;
  float* const l_ea = ef + E_ROW;
  float* const l_eb = l_ea + BM;
  int* const l_seg = reinterpret_cast<int*>(l_eb + BM);
  float* const l_red = ef + E_RED;
  float* const l_xx = ef + E_XX;
  const int D = BN;
  const int64_t last = (m0 + BM - 1 < a.M ? m0 + BM - 1 : a.M - 1);
  const int64_t b0 = m0 / a.N;
  const int segs = (int)(last / a.N - b0) + 1;   // <= MAX_SEG (checked by the launcher: N >= 64)
  for (int idx = threadIdx.x; idx < (3 * segs + 3) * (BN / 4); idx += THREADS) {
    const int v = idx / (BN / 4), c4 = idx - v * (BN / 4);
    const float* src;
    int slot;
    if (v < 3 * segs) {
      const int s = v / 3, which = v - 3 * s;
      const int64_t b = b0 + s;
      src = which == 0 ? a.hm2 + b * D : which == 1 ? a.hm2 + ((int64_t)a.B + b) * D : a.mt2 + b * D;
      slot = 3 * s + which;
    } else {
      const int which = v - 3 * segs;
      src = which == 0 ? a.b_h2 : which == 1 ? a.gamma : a.beta;
      slot = 3 * MAX_SEG + which;
    }
    st4(l_vec + slot * BN + c4 * 4, ld4(src + c4 * 4));
  }
  if (threadIdx.x < BM) {
    int64_t p = m0 + threadIdx.x;
    p = p < a.M ? p : a.M - 1;
    l_ea[threadIdx.x] = a.e1m[p];
    l_eb[threadIdx.x] = a.e1m[2 * a.M + p];
    l_seg[threadIdx.x] = (int)(p / a.N - b0);
  }
  __syncthreads();
  if (wave < segs) {   // |mt''|^2 of the tile's mentions
    const float* v = l_vec + (3 * wave + 2) * BN;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float4 x = ld4(v + (lane + 64 * j) * 4);
      s += dot4(x, x);
    }
    s = wave_sum(s);
    if (lane == 0) l_xx[wave] = s;
  }
  float ea[3], eb[3];
  int sg[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int row = wm * 48 + i * 16 + r;
    ea[i] = l_ea[row];
    eb[i] = l_eb[row];
    sg[i] = l_seg[row];
  }
  // the sum of this lane's values of a row, over the lanes that share the row (c = 0 .. 3), then over the four column groups of waves
  auto row_total = [&](float (&part)[3], int which) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      float v = part[i];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      if (c == 0) l_red[(which * BM + wm * 48 + i * 16 + r) * 4 + wn] = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float4 t = ld4(l_red + (which * BM + wm * 48 + i * 16 + r) * 4);
      part[i] = (t.x + t.y) + (t.z + t.w);
    }
  };
  // x = h + e1_tt (W_h2 mt') + e1_it (W_h2 mi') + b_h2 (in place), row sums
  float part[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int col = 192 * u + 48 * wn + 16 * j + 4 * c;
      const float4 bias = ld4(l_vec + (3 * MAX_SEG) * BN + col);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const float4 ut = ld4(l_vec + (3 * sg[i]) * BN + col), ui = ld4(l_vec + (3 * sg[i] + 1) * BN + col);
        const f32x2 w1 = splat2(ea[i]), w2 = splat2(eb[i]);
        const f32x2 lo = pk_fma(w1, lo2(ut), pk_fma(w2, lo2(ui), pk(acc[u][i][j][0], acc[u][i][j][1]) + lo2(bias)));
        const f32x2 hi = pk_fma(w1, hi2(ut), pk_fma(w2, hi2(ui), pk(acc[u][i][j][2], acc[u][i][j][3]) + hi2(bias)));
        acc[u][i][j][0] = lo.x;
        acc[u][i][j][1] = lo.y;
        acc[u][i][j][2] = hi.x;
        acc[u][i][j][3] = hi.y;
        const f32x2 s2 = lo + hi;
        part[i] += s2.x + s2.y;
      }
      __builtin_amdgcn_sched_barrier(0);   // one column group at a time: hoisting every group's LDS reads spills
    }
  row_total(part, 0);
  const float inv_d = 1.0f / (float)D;
  float mu[3], rstd[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    mu[i] = part[i] * inv_d;
    part[i] = 0.f;
  }
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const f32x2 m2 = splat2(mu[i]);
        const f32x2 dl = pk(acc[u][i][j][0], acc[u][i][j][1]) - m2, dh = pk(acc[u][i][j][2], acc[u][i][j][3]) - m2;
        const f32x2 q = pk_fma(dh, dh, dl * dl);
        part[i] += q.x + q.y;
      }
  row_total(part, 1);
#pragma unroll
  for (int i = 0; i < 3; ++i) rstd[i] = 1.0f / sqrtf(part[i] * inv_d + a.ln_eps);
  // y = gelu(LN(x)); dots with mt'' and with itself
  float xy[3] = {0.f, 0.f, 0.f}, yy[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int col = 192 * u + 48 * wn + 16 * j + 4 * c;
      const float4 g = ld4(l_vec + (3 * MAX_SEG + 1) * BN + col), bt = ld4(l_vec + (3 * MAX_SEG + 2) * BN + col);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const f32x2 m2 = splat2(mu[i]), r2 = splat2(rstd[i]);
        const f32x2 yl = gelu_fast2(pk_fma((pk(acc[u][i][j][0], acc[u][i][j][1]) - m2) * r2, lo2(g), lo2(bt)));
        const f32x2 yh = gelu_fast2(pk_fma((pk(acc[u][i][j][2], acc[u][i][j][3]) - m2) * r2, hi2(g), hi2(bt)));
        const float4 mt = ld4(l_vec + (3 * sg[i] + 2) * BN + col);
        const f32x2 d1 = pk_fma(yh, hi2(mt), yl * lo2(mt));
        const f32x2 d2 = pk_fma(yh, yh, yl * yl);
        xy[i] += d1.x + d1.y;
        yy[i] += d2.x + d2.y;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  row_total(xy, 2);
  row_total(yy, 3);
  if (wn == 0 && c == 0) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int64_t p = m0 + wm * 48 + i * 16 + r;
      if (p < a.M) a.scores[p] = cosine_from_sums(xy[i], l_xx[sg[i]], yy[i], a.cos_eps);
    }
  }
}

}  // namespace rows

bool rows_final_fits(int64_t M, int N_candidates, int D, int K, int64_t lda, int64_t ldb, const void* a_hi, const void* a_lo, const void* b_hi,
                     const void* b_lo) {
  static const bool off = [] {
    const char* e = getenv("DRIN_ROWS_FINAL");
    return e != nullptr && e[0] == '0';
  }();
  // at least one round of workgroups; a 96-row tile may touch at most three mentions
  return !off && D == rows::BN && K > 0 && (K % rows::BK) == 0 && K / rows::BK >= 2 && N_candidates >= 48 && M >= 256 * rows::BM &&
         cdiv(M, rows::BM) <= 0x7fffffff && (lda % 8) == 0 && (ldb % 8) == 0 && a_lo != nullptr && b_lo != nullptr && aligned16(a_hi) &&
         aligned16(a_lo) && aligned16(b_hi) && aligned16(b_lo);
}

int launch_rows_final(const void* a_hi, const void* a_lo, int64_t lda, const void* b_hi, const void* b_lo, int64_t ldb, const FinalArgs& f,
                      int K, hipStream_t st) {
  const int64_t M = (int64_t)f.B * f.N;
  if (!rows_final_fits(M, f.N, f.D4 * 4, K, lda, ldb, a_hi, a_lo, b_hi, b_lo) || f.act_v != DRIN_ACT_GELU) {
    set_error("rows_final: outside the kernel's contract (D = 768, candidate lists of >= 48, >= 24 576 pairs, default activation)");
    return DRIN_E_UNSUPPORTED;
  }
  rows::FinalRowsArgs a;
  a.a_hi = (const __bf16*)a_hi;
  a.a_lo = (const __bf16*)a_lo;
  a.b_hi = (const __bf16*)b_hi;
  a.b_lo = (const __bf16*)b_lo;
  a.lda = lda;
  a.ldb = ldb;
  a.hm2 = f.hm2;
  a.b_h2 = f.b_h2;
  a.gamma = f.gamma;
  a.beta = f.beta;
  a.e1m = f.e1m;
  a.mt2 = f.mt2;
  a.scores = f.scores;
  a.M = M;
  a.B = f.B;
  a.N = f.N;
  a.K = K;
  a.ln_eps = f.ln_eps;
  a.cos_eps = f.cos_eps;
  static DynLdsOptIn opt;
  DRIN_TRY(ensure_dynamic_lds(opt, reinterpret_cast<const void*>(rows::k_rows_final), rows::LDS_LOOP, "hipFuncSetAttribute(rows_final)"));
  KernelTimer timer(DRIN_KC_GEMM_PLANES, st);
  hipLaunchKernelGGL(rows::k_rows_final, dim3((unsigned)cdiv(M, rows::BM)), dim3(rows::THREADS), rows::LDS_LOOP, st, a);
  DRIN_CHECK_LAUNCH("k_rows_final");
  return DRIN_OK;
}

}  // namespace drin
