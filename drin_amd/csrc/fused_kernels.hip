// Kernels of the fused two-layer inference pipeline (fused_forward.hip explains the algebra).
//
// k_entity_stream is the HBM-bound heart of the path: ONE pass over the entity-side bytes of a
// mention (token features, image row, object rows) produces everything the first GCN layer needs from
// them - pooled text, the static edges, the dynamic edges of layer 1 (as dot products with
// per-mention vectors instead of a W_v contraction), and the edge-weighted sums that feed the
// mention vertices.  One 256-thread workgroup owns a chunk of one mention's candidates, so all the
// cross-candidate reductions stay on chip and no atomics are needed (results are bit-reproducible).
#include "device_utils.h"
#include "fused.h"
#include "internal.h"
#include "row_ops.h"

namespace drin {

// ------------------------------------------------------------------------------------------------
// grid (chunks, B), 256 threads.  Wave w of the workgroup takes candidates c0 + w, c0 + w + 4, ...
// EXACT: D = 256 DV and R = 256 RV exactly (768 / 2048): the column guards of the row helpers fold away
// FT: storage type of the feature tensors (float, or __bf16 with drin_config.feature_dtype = DRIN_FEAT_BF16 - half
// the bytes of this HBM-bound pass; all arithmetic stays fp32)
// XSCALE: also hand over the image row as ONE fp16 plane under a power-of-two scale per row (DRIN_PREC_BF16X3_IF16) - a separate
// instantiation, so that the code (and the register allocation: 230 VGPRs) of every other call is what it was
#ifdef DRIN_STREAM_STAMPS   // probe build (tools/stream_stamps_probe.py): cycle stamps of 64 workgroups, wave 0
__device__ unsigned long long g_stream_stamps[64 * 8];
#define STREAM_STAMP(i)                                                                                        \
  if (threadIdx.x == 0 && blockIdx.x == 0 && (blockIdx.y % (gridDim.y / 64 ? gridDim.y / 64 : 1)) == 0 &&       \
      blockIdx.y / (gridDim.y / 64 ? gridDim.y / 64 : 1) < 64)                                                   \
  g_stream_stamps[(blockIdx.y / (gridDim.y / 64 ? gridDim.y / 64 : 1)) * 8 + (i)] = __builtin_readcyclecounter()
#else
#define STREAM_STAMP(i)
#endif
template <int DV, int RV, bool TOKENS, bool EXACT, typename FT, bool XSCALE = false>
__global__ void __launch_bounds__(256, 2) k_entity_stream(const StreamArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  STREAM_STAMP(0);
  const FT* const f_text = static_cast<const FT*>(a.entity_text);
  const FT* const f_image = static_cast<const FT*>(a.entity_image);
  const FT* const f_object = static_cast<const FT*>(a.entity_object);
  const FT* const f_mobj = static_cast<const FT*>(a.mobj);
  const int D4 = EXACT ? DV * 64 : a.D4, R4 = EXACT ? RV * 64 : a.R4, D = D4 * 4, R = R4 * 4;
  // image / object rows stored as bf16: eight consecutive columns per lane and load (row_ops.h: the PAIR layout) - the sums
  // over candidates below stay per register slot, only the helpers that meet memory know the column map
  constexpr bool RP = sizeof(FT) == 2 && (RV % 2) == 0;
  auto load_r = [&](const FT* row) -> Row<RV> {
    if constexpr (RP)
      return load_row_stream_pair<RV>(row, (int)(threadIdx.x & 63), R4);
    else
      return load_row_stream<RV>(row, (int)(threadIdx.x & 63), R4);
  };
  float* l_mobj = lds;                          // [Km][R]
  float* l_q = l_mobj + a.Km * R;               // [2][R]   q_ti, q_ii
  float* l_red = l_q + 2 * R;                   // [2 D + 2 R] cross-wave reduction of the weighted sums
  float* l_mt = l_red + 2 * D + 2 * R;          // [3][D]   span mean, q_tt, q_it (kept out of the register file)
  float* l_small = l_mt + 3 * D;                // [Km] clamped object norms, then [4] edge sums
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t b = blockIdx.y;
  const int per = (a.N + a.chunks - 1) / a.chunks;
  const int n_begin = blockIdx.x * per, n_end = min(a.N, n_begin + per);
  const bool dyn = a.dynamic != 0;

  // ---- per-workgroup prologue: mention-side vectors into LDS / registers -----------------------------
  for (int i = threadIdx.x; i < a.Km * R4; i += 256) st4(l_mobj + i * 4, ld4(f_mobj + (b * a.Km) * R + i * 4));
  for (int i = threadIdx.x; i < D4; i += 256) st4(l_mt + i * 4, ld4(a.span_mean + b * D + i * 4));
  if (dyn) {
    for (int i = threadIdx.x; i < 2 * R4; i += 256) {
      const int which = i / R4, c4 = i - which * R4;  // rows fu_t (b) and fu_i (B + b) of q, columns D .. D + R
      st4(l_q + i * 4, ld4(a.q + ((int64_t)which * a.B + b) * a.ldq + D + c4 * 4));
    }
    for (int i = threadIdx.x; i < 2 * D4; i += 256) {
      const int which = i / D4, c4 = i - which * D4;  // columns 0 .. D of the same rows
      st4(l_mt + D + i * 4, ld4(a.q + ((int64_t)which * a.B + b) * a.ldq + c4 * 4));
    }
  }
  // (Measured in round 4 and dropped - profiles/r4_stream_stamps.txt: these four load -> store loops as ONE index space with eight
  //  16-byte loads per thread in flight before the first LDS store.  Cycle stamps of a WikiDiverse workgroup: this stage 16.5 k ->
  //  22.7 k cycles of 82 k, the kernel 1.16 -> 1.23 ms.  The stage does not wait for dependent round trips: its loads queue behind
  //  the ~100 KB the CU's other workgroup keeps in flight, and a burst of 2 048 of them queues longer.)
  __syncthreads();
  STREAM_STAMP(1);
  for (int i = wave; i < a.Km; i += 4) {  // |mobj_i| (model.py:88: every pair re-normalises the same rows)
    float s = 0.f;
    for (int c4 = lane; c4 < R4; c4 += 64) {
      const float4 v = ld4(l_mobj + i * R + c4 * 4);
      s += dot4(v, v);
    }
    s = wave_sum(s);
    if (lane == 0) l_small[i] = fmaxf(sqrtf(s), a.cos_eps);
  }
  float m_t_norm;
  {
    const Row<DV> m_t = load_row<DV>(l_mt, lane, D4);
    m_t_norm = fmaxf(sqrtf(wave_sum(dot_rows<DV>(m_t, m_t))), a.cos_eps);
  }
  float kap_tt = 0.f, kap_ti = 0.f, kap_it = 0.f, kap_ii = 0.f;
  if (dyn) {
    const Row<DV> fu_t = load_row<DV>(a.fu + b * a.ldfu, lane, D4);
    const Row<DV> fu_i = load_row<DV>(a.fu + ((int64_t)a.B + b) * a.ldfu, lane, D4);
    const Row<DV> k_t = load_row<DV>(a.k_t, lane, D4), k_i = load_row<DV>(a.k_i, lane, D4);
    kap_tt = wave_sum(dot_rows<DV>(fu_t, k_t));
    kap_ti = wave_sum(dot_rows<DV>(fu_t, k_i));
    kap_it = wave_sum(dot_rows<DV>(fu_i, k_t));
    kap_ii = wave_sum(dot_rows<DV>(fu_i, k_i));
  }
  STREAM_STAMP(2);
  __syncthreads();
  STREAM_STAMP(3);

  Row<DV> S_tt = zero_row<DV>(), S_it = zero_row<DV>();
  Row<RV> S_ti = zero_row<RV>(), S_ii = zero_row<RV>();
  float sg_tt = 0.f, sg_ti = 0.f, sg_it = 0.f, sg_ii = 0.f;
  const float inv_d = 1.0f / (float)D;

  // (Measured in round 5 and dropped - profiles/r5_stream_reorder_ab.txt: the mask words, the object row and the image row requested at
  //  the top of a pair and worked off BEFORE the token walk, so that the walk starts with its count already there.  In-process A/B on
  //  one resident batch (tools/probes/inprocess_lib_ab.py): 7.96 against 7.97 ms, bit-identical scores - the other seven waves of the CU
  //  already cover a wave's round trips - for 256 instead of 230 VGPRs and spills in the bf16 and XSCALE instantiations.)
  // (Measured in round 6 and dropped - profiles/r6_stream_prefetch_ab.txt: the NEXT pair's candidate row and mask word requested at the
  //  top of the current pair, so that the chain row index -> mask words -> token count -> first token loads no longer starts cold.
  //  In-process A/B on one resident batch: fp32 rows 8.31 against 8.31 ms, bf16-stored rows 4.75 against 4.78, WikiDiverse 1.16 / 1.17 -
  //  bit-identical scores, four registers: the CU's other seven waves already cover a wave's cold start.)
  for (int n = n_begin + wave; n < n_end; n += 4) {
    const int64_t p = b * a.N + n;
    // entity row: the pair itself, or a row of the entity tables (on-device form of data.py:87-93)
    int64_t e = p;
    if (a.entity_index) {
      const int64_t raw = a.entity_index[p];
      e = raw < 0 ? 0 : (raw >= a.num_entities ? a.num_entities - 1 : raw);
      if (e != raw && lane == 0) report_bad_index(a.index_status, p, raw);   // data.py:87-93 would raise: clamped, and reported
    }
    // ---- text: CLS / pooler cosine (model.py:71-76) and token mean (ghmfc.py:245-249) ---------------
    Row<DV> xt;
    float tt;
    if (TOKENS) {
      const int T = a.T;
      int cnt = 0;
      for (int t = lane; t < T; t += 64) cnt += (int)a.entity_mask[e * T + t];
      cnt = (int)wave_sum((float)cnt);
      int stop = cnt - 1;
      if (stop < 0) stop += T;
      stop = stop < 0 ? 0 : (stop > T ? T : stop);
      const FT* base = f_text + e * (int64_t)T * D;
      constexpr bool FLAT = sizeof(FT) == 2 && EXACT && DV == 3;   // the flat walk below
      const Row<DV> cls = load_row_stream<DV>(base, lane, D4);
      Row<DV> acc = zero_row<DV>();
      int t = 1;
      if constexpr (FLAT) {
        // bf16 token rows at D = 768: a row is 1 536 B - three 8-byte loads per lane, and 8-byte accesses move at 0.54-0.70 of
        // the 16-byte rate (MI355X_MICROARCH.md; round 3: 4.66 TB/s here against 6.1 for the fp32 rows).  The token rows of an
        // entity are contiguous, so TWO rows are exactly three 16-byte loads per lane with every lane busy: the pair is walked
        // FLAT - chunk k, lane l holds elements 8 (64 k + l) .. + 7 of the 1 536 - with one accumulator set per (chunk, lane),
        // four pairs (12 KB) in flight per wave.  First rows of the pairs land in flat positions 0 .. 95, second rows in
        // 96 .. 191 = the same columns 32 lanes further on: one half-wave swap per candidate folds them, and a trip through this
        // wave's own 3 KB of the (idle until the end) reduction area turns the 8-columns-per-lane order into the row layout
        // the rest of the kernel works in.  The sum is the rows' sum in another association: first rows, second rows, then
        // the unpaired last row (scores move by <= 2e-6 against the row-by-row order; tests/test_gpu_parity.py).
        // (Measured and left out: the CLS row taken the same way, as the first row of a flat pair with token 1 - same-box
        //  A/B inside the +-6 % placement spread of this kernel, eight registers more; profiles/r4_bf16_rows_ab.txt.)
        float f0[8], f1[8], f2[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) f0[i] = f1[i] = f2[i] = 0.f;
        const char* flat = reinterpret_cast<const char*>(base) + (size_t)lane * 16;
        auto add8 = [](float (&f)[8], const u32x4_t v) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            f[2 * i] += __builtin_bit_cast(float, v[i] << 16);
            f[2 * i + 1] += __builtin_bit_cast(float, v[i] & 0xffff0000u);
          }
        };
        for (; t + 8 <= stop; t += 8) {
          u32x4_t r[4][3];
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < 3; ++k) r[u][k] = ld16_stream(flat + (size_t)(t + 2 * u) * (2 * 768) + k * 1024);
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            add8(f0, r[u][0]);
            add8(f1, r[u][1]);
            add8(f2, r[u][2]);
          }
        }
        for (; t + 2 <= stop; t += 2) {
          const u32x4_t r0 = ld16_stream(flat + (size_t)t * (2 * 768)), r1 = ld16_stream(flat + (size_t)t * (2 * 768) + 1024),
                        r2 = ld16_stream(flat + (size_t)t * (2 * 768) + 2048);
          add8(f0, r0);
          add8(f1, r1);
          add8(f2, r2);
        }
        // fold the second rows onto the first: flat position p + 96 is chunk (p + 96) / 64, lane (p + 32) % 64
        float* scr = l_red + wave * 768;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float y1 = __shfl_xor(f1[i], 32), y2 = __shfl_xor(f2[i], 32);
          f0[i] += lane < 32 ? y1 : y2;   // columns 8 lane .. + 7
          f1[i] += y2;                    // lanes 0 .. 31: columns 512 + 8 lane .. + 7
        }
        st4(scr + 8 * lane, make_float4(f0[0], f0[1], f0[2], f0[3]));
        st4(scr + 8 * lane + 4, make_float4(f0[4], f0[5], f0[6], f0[7]));
        if (lane < 32) {
          st4(scr + 512 + 8 * lane, make_float4(f1[0], f1[1], f1[2], f1[3]));
          st4(scr + 512 + 8 * lane + 4, make_float4(f1[4], f1[5], f1[6], f1[7]));
        }
        __builtin_amdgcn_wave_barrier();   // this wave's own region: its LDS operations execute in order
#pragma unroll
        for (int j = 0; j < DV; ++j) acc.v[j] = ld4(scr + 4 * (lane + 64 * j));
        __builtin_amdgcn_wave_barrier();
      } else if constexpr (sizeof(FT) == 2 && (EXACT || DV == 1)) {
        // bf16 rows are half the bytes: twice as many of them in flight keeps the same bytes in flight per wave
        // (the pass is bound by per-CU memory latency, not by instruction issue); same left-to-right sum.
        // (Not in the guarded-column instantiation of the wide rows: with the column guards live, eight rows in flight
        //  spilled 26 VGPRs to scratch - tests/test_host.py::test_no_kernel_uses_scratch; four rows fit, same sum order.)
        for (; t + 8 <= stop; t += 8) {
          Row<DV> r[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) r[u] = load_row_stream<DV>(base + (int64_t)(t + u) * D, lane, D4);
#pragma unroll
          for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int j = 0; j < DV; ++j) acc.v[j] = acc.v[j] + r[u].v[j];
        }
      }
      // (round 6, measured and dropped: SIX fp32 token rows in flight per wave - 256 VGPRs, same bits - 8.61 = 8.62 ms and 8.41 = 8.41 ms in
      //  two processes: profiles/r6_stream_prefetch_ab.txt; eight rows spill)
      for (; t + 4 <= stop; t += 4) {  // 4 token rows (12 KB at D = 768) in flight per wave
        const Row<DV> r0 = load_row_stream<DV>(base + (int64_t)t * D, lane, D4);
        const Row<DV> r1 = load_row_stream<DV>(base + (int64_t)(t + 1) * D, lane, D4);
        const Row<DV> r2 = load_row_stream<DV>(base + (int64_t)(t + 2) * D, lane, D4);
        const Row<DV> r3 = load_row_stream<DV>(base + (int64_t)(t + 3) * D, lane, D4);
#pragma unroll
        for (int j = 0; j < DV; ++j) acc.v[j] = (((acc.v[j] + r0.v[j]) + r1.v[j]) + r2.v[j]) + r3.v[j];
      }
      for (; t < stop; ++t) {
        const Row<DV> r0 = load_row_stream<DV>(base + (int64_t)t * D, lane, D4);
#pragma unroll
        for (int j = 0; j < DV; ++j) acc.v[j] = acc.v[j] + r0.v[j];
      }
      const float den = stop > 1 ? (float)(stop - 1) : 0.0f;  // empty slice: 0 / 0 = NaN like the reference
#pragma unroll
      for (int j = 0; j < DV; ++j)
        xt.v[j] = make_float4(acc.v[j].x / den, acc.v[j].y / den, acc.v[j].z / den, acc.v[j].w / den);
      if (a.xt_out) store_row<DV>(a.xt_out + p * D, xt, lane, D4);
      const float xy = wave_sum(dot_row_lds<DV>(cls, l_mt, lane, D4)), yy = wave_sum(dot_rows<DV>(cls, cls));
      tt = xy / (m_t_norm * fmaxf(sqrtf(yy), a.cos_eps));
    } else {
      xt = load_row_stream<DV>(f_text + e * D, lane, D4);
      const float xy = wave_sum(dot_row_lds<DV>(xt, l_mt, lane, D4)), yy = wave_sum(dot_rows<DV>(xt, xt));
      tt = xy / (m_t_norm * fmaxf(sqrtf(yy), a.cos_eps));
    }
    if (a.xt_hi) store_row_planes<DV>(a.xt_hi, a.xt_lo, p * D, xt, lane, D4);
    // ---- objects: weighted pair similarity (model.py:84-92) ------------------------------------------
    float sim = 0.f, wsum = 0.f;
    for (int j = 0; j < a.Ke; ++j) {
      const Row<RV> eo = load_r(f_object + (e * a.Ke + j) * R);
      const float ny = fmaxf(sqrtf(wave_sum(dot_rows<RV>(eo, eo))), a.cos_eps);
      const float es = a.entity_object_score[e * a.Ke + j];
      for (int i = 0; i < a.Km; ++i) {
        const float xy = wave_sum(dot_row_lds<RV, RP>(eo, l_mobj + i * R, lane, R4));
        const float w = a.mscore[b * a.Km + i] * es;
        sim += xy / (l_small[i] * ny) * w;
        wsum += w;
      }
    }
    // the reference sums i-major, j-minor; with Ke = 1 (both datasets) the orders coincide
    const float ii = sim / (wsum + a.miei_eps);
    // ---- image row + edges ----------------------------------------------------------------------------
    const Row<RV> xi = load_r(f_image + e * R);
    if (a.xi_hi) store_row_planes<RV, RP>(a.xi_hi, a.xi_lo, p * R, xi, lane, R4);
    if constexpr (XSCALE) {
      // DRIN_PREC_BF16X3_IF16: the power of two that brings this row's largest |x| into [0.5, 1] (its reciprocal a normal number
      // whatever the row: cache_field_scale) - the row leaves as fp16(x / scale), 4 KB instead of the 8 KB the contraction would
      // read again as fp32, and the contraction's epilogue multiplies its output row back (exact both ways)
      float m = 0.f;
#pragma unroll
      for (int j = 0; j < RV; ++j) m = fmaxf(m, fmaxf(fmaxf(fabsf(xi.v[j].x), fabsf(xi.v[j].y)), fmaxf(fabsf(xi.v[j].z), fabsf(xi.v[j].w))));
      const float sc = cache_field_scale(wave_max(m));
      if (lane == 0) a.xi_scale[p] = sc;
      store_row_f16_scaled<RV, RP>(a.xi_f16, p * R, xi, 1.0f / sc, lane);
    }
    const float e_tt = tt * a.mask[0];
    const float e_ti = (a.mtei[p] / a.clip) * a.mask[1];
    const float e_it = (a.miet[p] / a.clip) * a.mask[2];
    const float e_ii = ii * a.mask[3];
    float n_tt = e_tt, n_ti = e_ti, n_it = e_it, n_ii = e_ii;  // layer-2 edges (static: pass-through, model.py:136)
    if (dyn) {  // e' = sigmoid(mean_d(W_u(u) * W_v(v)) + e) with W_v folded into q, kappa (model.py:148-153)
      const float d_tt = wave_sum(dot_row_lds<DV>(xt, l_mt + D, lane, D4));
      const float d_it = wave_sum(dot_row_lds<DV>(xt, l_mt + 2 * D, lane, D4));
      const float d_ti = wave_sum(dot_row_lds<RV, RP>(xi, l_q, lane, R4));
      const float d_ii = wave_sum(dot_row_lds<RV, RP>(xi, l_q + R, lane, R4));
      n_tt = edge_act_apply(a.act_e, (d_tt + kap_tt) * inv_d + e_tt);
      n_ti = edge_act_apply(a.act_e, (d_ti + kap_ti) * inv_d + e_ti);
      n_it = edge_act_apply(a.act_e, (d_it + kap_it) * inv_d + e_it);
      n_ii = edge_act_apply(a.act_e, (d_ii + kap_ii) * inv_d + e_ii);
    }
    if (lane == 0) {
      const int64_t M = (int64_t)a.B * a.N;
      a.e0m[p] = e_tt;
      a.e0m[M + p] = e_ti;
      a.e0m[2 * M + p] = e_it;
      a.e0m[3 * M + p] = e_ii;
      a.e1m[p] = n_tt * a.mask[0];
      a.e1m[M + p] = n_ti * a.mask[1];
      a.e1m[2 * M + p] = n_it * a.mask[2];
      a.e1m[3 * M + p] = n_ii * a.mask[3];
    }
    // ---- edge-weighted sums for the mention vertices (model.py:143-144, before the Linear) -----------
    axpy_row<DV>(S_tt, e_tt, xt);
    axpy_row<DV>(S_it, e_it, xt);
    axpy_row<RV>(S_ti, e_ti, xi);
    axpy_row<RV>(S_ii, e_ii, xi);
    sg_tt += e_tt;
    sg_ti += e_ti;
    sg_it += e_it;
    sg_ii += e_ii;
  }

  STREAM_STAMP(4);
  // ---- fixed-order cross-wave reduction through LDS, then one partial per (mention, chunk) -----------
  for (int w = 0; w < 4; ++w) {
    __syncthreads();
    if (wave == w) {
#pragma unroll
      for (int j = 0; j < DV; ++j) {
        const int c4 = lane + 64 * j;
        if (c4 < D4) {
          float* p0 = l_red + c4 * 4;
          float* p1 = l_red + D + c4 * 4;
          st4(p0, w == 0 ? S_tt.v[j] : ld4(p0) + S_tt.v[j]);
          st4(p1, w == 0 ? S_it.v[j] : ld4(p1) + S_it.v[j]);
        }
      }
#pragma unroll
      for (int j = 0; j < RV; ++j) {
        const int c4 = row_col4<RP>(lane, j);
        if (c4 < R4) {
          float* p0 = l_red + 2 * D + c4 * 4;
          float* p1 = l_red + 2 * D + R + c4 * 4;
          st4(p0, w == 0 ? S_ti.v[j] : ld4(p0) + S_ti.v[j]);
          st4(p1, w == 0 ? S_ii.v[j] : ld4(p1) + S_ii.v[j]);
        }
      }
      if (lane == 0) {
        float* s4 = l_small + a.Km;
        s4[0] = (w == 0 ? 0.f : s4[0]) + sg_tt;
        s4[1] = (w == 0 ? 0.f : s4[1]) + sg_ti;
        s4[2] = (w == 0 ? 0.f : s4[2]) + sg_it;
        s4[3] = (w == 0 ? 0.f : s4[3]) + sg_ii;
      }
    }
  }
  __syncthreads();
  STREAM_STAMP(5);
  if (a.chunks == 1) {
    // short candidate lists (N <= 16, WikiDiverse): the workgroup holds the whole mention - write the layout
    // the mention-side GEMMs read and skip the partial buffer and its reduction pass
    for (int i = threadIdx.x; i < 2 * D4; i += 256) {
      const int which = i / D4, c4 = i - which * D4;
      st4(a.s_text + ((int64_t)which * a.B + b) * D + c4 * 4, ld4(l_red + i * 4));
    }
    for (int i = threadIdx.x; i < 2 * R4; i += 256) {
      const int which = i / R4, c4 = i - which * R4;
      st4(a.s_img + ((int64_t)which * a.B + b) * R + c4 * 4, ld4(l_red + 2 * D + i * 4));
    }
    if (threadIdx.x < 4) a.sig[(int64_t)threadIdx.x * a.B + b] = l_small[a.Km + threadIdx.x];
    STREAM_STAMP(6);
    return;
  }
  float* out = a.s_part + (b * a.chunks + blockIdx.x) * (int64_t)(2 * D + 2 * R + 4);
  for (int i = threadIdx.x; i < (2 * D + 2 * R) / 4; i += 256) st4(out + i * 4, ld4(l_red + i * 4));
  if (threadIdx.x < 4) out[2 * D + 2 * R + threadIdx.x] = l_small[a.Km + threadIdx.x];
}

#ifdef DRIN_STREAM_STAMPS
}  // namespace drin
extern "C" __attribute__((visibility("default"))) int drin_debug_stream_stamps(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(drin::g_stream_stamps), sizeof(unsigned long long) * 64 * 8);
}
namespace drin {
#endif

size_t entity_stream_lds_bytes(const StreamArgs& a) {
  const size_t D = (size_t)a.D4 * 4, R = (size_t)a.R4 * 4;
  return sizeof(float) * (a.Km * R + 2 * R + 2 * D + 2 * R + 3 * D + a.Km + 4);
}

template <int DV, int RV, bool TOKENS, bool EXACT, typename FT, bool XSCALE = false>
static int launch_stream_ft(const StreamArgs& a, hipStream_t st) {
  const size_t lds = entity_stream_lds_bytes(a);
  auto kern = k_entity_stream<DV, RV, TOKENS, EXACT, FT, XSCALE>;
  static DynLdsOptIn opt_in;  // one per template instantiation
  if (lds > 48 * 1024)
    DRIN_TRY(ensure_dynamic_lds(opt_in, reinterpret_cast<const void*>(kern), (int)lds, "hipFuncSetAttribute(entity_stream)"));
  KernelTimer timer(DRIN_KC_STREAM, st);
  hipLaunchKernelGGL(kern, dim3((unsigned)a.chunks, (unsigned)a.B), dim3(256), lds, st, a);
  DRIN_CHECK_LAUNCH("k_entity_stream");
  return DRIN_OK;
}

template <int DV, int RV, bool TOKENS, bool EXACT>
static int launch_stream_t(const StreamArgs& a, hipStream_t st) {
  if (a.xi_scale != nullptr) {   // the exact widths only (fused_forward.hip decides)
    if constexpr (EXACT && DV == 3) {
      if (a.xi_f16 == nullptr) {
        set_error("entity_stream: xi_scale without xi_f16");
        return DRIN_E_NULL;
      }
      if (!a.bf16_features) return launch_stream_ft<DV, RV, TOKENS, EXACT, float, true>(a, st);
    }
    set_error("entity_stream: the scaled fp16 image plane is built for fp32-stored rows at D = 768, R = 2048");
    return DRIN_E_UNSUPPORTED;
  }
  return a.bf16_features ? launch_stream_ft<DV, RV, TOKENS, EXACT, __bf16>(a, st)
                         : launch_stream_ft<DV, RV, TOKENS, EXACT, float>(a, st);
}

int launch_entity_stream(const StreamArgs& a, hipStream_t st) {
  if (a.B <= 0) return DRIN_OK;
  if (a.B > 65535) {
    set_error("entity_stream: batch %d exceeds the grid limit; split the batch", a.B);
    return DRIN_E_SHAPE;
  }
  const bool tok = a.T > 0;
  if (a.D4 <= 64 && a.R4 <= 64)
    return tok ? launch_stream_t<1, 1, true, false>(a, st) : launch_stream_t<1, 1, false, false>(a, st);
  if (a.D4 == 192 && a.R4 == 512)
    return tok ? launch_stream_t<3, 8, true, true>(a, st) : launch_stream_t<3, 8, false, true>(a, st);
  if (a.D4 <= 192 && a.R4 <= 512)
    return tok ? launch_stream_t<3, 8, true, false>(a, st) : launch_stream_t<3, 8, false, false>(a, st);
  set_error("entity_stream: D=%d R=%d outside the built instantiations", a.D4 * 4, a.R4 * 4);
  return DRIN_E_UNSUPPORTED;
}

// ------------------------------------------------------------------------------------------------
// Sum the per-chunk partials in chunk order and lay them out for the mention-side GEMMs:
//   s_text [2][B][D] = (S_tt, S_it), s_img [2][B][R] = (S_ti, S_ii), sig [4][B] = (tt, ti, it, ii)
// (four columns per thread, the chunk loads of a trip independent of each other: 0.21 -> ~0.13 ms per 4096 mentions;
//  D and R are multiples of 4, so a float4 never straddles two of the output regions)
__global__ void __launch_bounds__(256) k_reduce_stream_partials(const float* __restrict__ part, float* __restrict__ s_text,
                                                                float* __restrict__ s_img, float* __restrict__ sig,
                                                                int B, int D, int R, int chunks) {
  const int64_t b = blockIdx.y;
  const int width = 2 * D + 2 * R + 4;
  const int i = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= width) return;
  const float* src = part + (b * chunks) * (int64_t)width + i;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  int c = 0;
  for (; c + 4 <= chunks; c += 4) {
    const float4 v0 = ld4(src + (int64_t)c * width), v1 = ld4(src + (int64_t)(c + 1) * width);
    const float4 v2 = ld4(src + (int64_t)(c + 2) * width), v3 = ld4(src + (int64_t)(c + 3) * width);
    s = (((s + v0) + v1) + v2) + v3;  // chunk order, as a serial loop would add them
  }
  for (; c < chunks; ++c) s = s + ld4(src + (int64_t)c * width);
  if (i < D) {
    st4(s_text + b * D + i, s);
  } else if (i < 2 * D) {
    st4(s_text + ((int64_t)B + b) * D + (i - D), s);
  } else if (i < 2 * D + R) {
    st4(s_img + b * R + (i - 2 * D), s);
  } else if (i < 2 * D + 2 * R) {
    st4(s_img + ((int64_t)B + b) * R + (i - 2 * D - R), s);
  } else {
    sig[b] = s.x;
    sig[(int64_t)B + b] = s.y;
    sig[2 * (int64_t)B + b] = s.z;
    sig[3 * (int64_t)B + b] = s.w;
  }
}

int launch_reduce_stream_partials(const float* part, float* s_text, float* s_img, float* sig, int B, int D, int R,
                                  int chunks, hipStream_t st) {
  if (B <= 0) return DRIN_OK;
  KernelTimer timer(DRIN_KC_GCN, st);
  hipLaunchKernelGGL(k_reduce_stream_partials, dim3((unsigned)cdiv((2 * D + 2 * R + 4) / 4, 256), (unsigned)B), dim3(256), 0,
                     st, part, s_text, s_img, sig, B, D, R, chunks);
  DRIN_CHECK_LAUNCH("k_reduce_stream_partials");
  return DRIN_OK;
}

// ------------------------------------------------------------------------------------------------
// W_h input of the two mention vertices of layer 1 (model.py:143-144 + :128 self term), from the
// contracted sums (T = S_text W_et^T, T2 = S_img W_ei^T):
//   out[t][b] = (T[t][b] + T2[t][b] + sig_a[t][b] b_et + sig_b[t][b] b_ei) / N + v0[t][b]
// with (sig_a, sig_b) = (tt, ti) for t = 0 (mention text) and (it, ii) for t = 1 (mention image).
__global__ void __launch_bounds__(256) k_mention_input1(const float* __restrict__ T, const float* __restrict__ T2,
                                                        const float* __restrict__ sig,
                                                        const float* __restrict__ b_et, const float* __restrict__ b_ei,
                                                        const float* __restrict__ v0, float* __restrict__ out, int B,
                                                        int D, float inv_n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= 2 * (int64_t)B * D) return;
  const int d = (int)(i % D);
  const int64_t row = i / D;  // t * B + b
  const int t = (int)(row / B);
  const int64_t b = row - (int64_t)t * B;
  const float sa = sig[(int64_t)(t == 0 ? 0 : 2) * B + b], sb = sig[(int64_t)(t == 0 ? 1 : 3) * B + b];
  out[i] = ((T[i] + T2[i]) + sa * b_et[d] + sb * b_ei[d]) * inv_n + v0[i];
}

int launch_mention_input1(const float* T, const float* T2, const float* sig, const float* b_et, const float* b_ei,
                          const float* v0, float* out, int B, int D, int N, hipStream_t st) {
  if (B <= 0) return DRIN_OK;
  KernelTimer timer(DRIN_KC_GCN, st);
  hipLaunchKernelGGL(k_mention_input1, dim3((unsigned)cdiv(2 * (int64_t)B * D, 256)), dim3(256), 0, st, T, T2, sig,
                     b_et, b_ei, v0, out, B, D, 1.0f / (float)N);
  DRIN_CHECK_LAUNCH("k_mention_input1");
  return DRIN_OK;
}

// ------------------------------------------------------------------------------------------------
// out[c][r] = in[r][c]   (one-off, drin_prepare: turns W_v1 [W_et | W_ei] into the nn.Linear layout)
__global__ void __launch_bounds__(256) k_transpose(const float* __restrict__ in, float* __restrict__ out, int rows,
                                                   int cols) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  for (int i = ty; i < 32; i += 8)
    if (r0 + i < rows && c0 + tx < cols) tile[i][tx] = in[(int64_t)(r0 + i) * cols + c0 + tx];
  __syncthreads();
  for (int i = ty; i < 32; i += 8)
    if (c0 + i < cols && r0 + tx < rows) out[(int64_t)(c0 + i) * rows + r0 + tx] = tile[tx][i];
}

int launch_transpose(const float* in, float* out, int rows, int cols, hipStream_t st) {
  if (rows <= 0 || cols <= 0) return DRIN_OK;
  KernelTimer timer(DRIN_KC_GCN, st);
  hipLaunchKernelGGL(k_transpose, dim3((unsigned)cdiv(cols, 32), (unsigned)cdiv(rows, 32)), dim3(256), 0, st, in, out,
                     rows, cols);
  DRIN_CHECK_LAUNCH("k_transpose");
  return DRIN_OK;
}

// ------------------------------------------------------------------------------------------------
// Layer-1 entity vertices from the folded contractions (grid (chunks, B), 256 threads):
//   et1[p] = gelu(LN(Hraw_t[p] + e_tt hm_t[b] + e_it hm_i[b] + c_t))          -> written (layer-2 GEMM operand)
//   ei1[p] = gelu(LN(Hraw_i[p] + e_ti hm_t[b] + e_ii hm_i[b] + c_i))          -> registers only
// and the layer-2 mention aggregates  S2_t[b] = sum_n e1_tt et1,  S2_i[b] = sum_n e1_ti ei1  per chunk.
template <int DV, bool EXACT, bool GENERIC_ACT = false>
__global__ void __launch_bounds__(256) k_pair_layer1(const PairArgs a) {
  __shared__ __attribute__((aligned(16))) float l_red[2 * DV * 256];
  __shared__ __attribute__((aligned(16))) float l_const[6 * DV * 256];  // hm_t, hm_i, c_t, c_i, gamma, beta
  const int D4 = EXACT ? DV * 64 : a.D4, D = D4 * 4;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t b = blockIdx.y;
  const int64_t M = (int64_t)a.B * a.N;
  const int per = (a.N + a.chunks - 1) / a.chunks;
  const int n_begin = blockIdx.x * per, n_end = min(a.N, n_begin + per);
  constexpr int LD = DV * 256;
  {
    const float* src[6] = {a.hm + b * a.ldhm, a.hm + ((int64_t)a.B + b) * a.ldhm, a.c_t, a.c_i, a.gamma, a.beta};
#pragma unroll
    for (int v = 0; v < 6; ++v)
      for (int i = threadIdx.x; i < D4; i += 256) st4(l_const + v * LD + i * 4, ld4(src[v] + i * 4));
  }
  __syncthreads();
  const float *l_hm_t = l_const, *l_hm_i = l_const + LD, *l_ct = l_const + 2 * LD, *l_ci = l_const + 3 * LD;
  const float *l_gamma = l_const + 4 * LD, *l_beta = l_const + 5 * LD;
  Row<DV> S_t = zero_row<DV>(), S_i = zero_row<DV>();
  for (int n = n_begin + wave; n < n_end; n += 4) {
    const int64_t p = b * a.N + n;
#ifdef DRIN_ABLATE_ROW_TRAFFIC   // timing ablation only (wrong results): the same arithmetic on four cache-resident rows, nothing stored
    const int64_t p_ld = b * a.N + (n & 3);
    const Row<DV> ht = load_row<DV>(a.h_text + p_ld * D, lane, D4);
    const Row<DV> hi = load_row<DV>(a.h_image + p_ld * D, lane, D4);
#else
    const Row<DV> ht = load_row_stream<DV>(a.h_text + p * D, lane, D4);  // read once: streaming cache policy
    const Row<DV> hi = load_row_stream<DV>(a.h_image + p * D, lane, D4);
#endif
    const float e_tt = a.e0m[p], e_ti = a.e0m[M + p], e_it = a.e0m[2 * M + p], e_ii = a.e0m[3 * M + p];
    const Row<DV> et1 = ln_gelu_row_lds<DV, GENERIC_ACT>(combine_rows_lds<DV>(ht, e_tt, l_hm_t, e_it, l_hm_i, l_ct, lane, D4),
                                                         l_gamma, l_beta, lane, D4, a.ln_eps, a.act_v);
#ifdef DRIN_ABLATE_ROW_TRAFFIC
    if (a.ln_eps < 0.f) store_row_planes<DV>(a.et1_hi, a.et1_lo, p * D, et1, lane, D4);   // never: keeps the values live
#else
    if (a.et1) store_row<DV>(a.et1 + p * D, et1, lane, D4);
    if (a.et1_hi) store_row_planes<DV>(a.et1_hi, a.et1_lo, p * D, et1, lane, D4);
#endif
    axpy_row_pk<DV>(S_t, a.e1m[p], et1);
    const Row<DV> ei1 = ln_gelu_row_lds<DV, GENERIC_ACT>(combine_rows_lds<DV>(hi, e_ti, l_hm_t, e_ii, l_hm_i, l_ci, lane, D4),
                                                         l_gamma, l_beta, lane, D4, a.ln_eps, a.act_v);
    axpy_row_pk<DV>(S_i, a.e1m[M + p], ei1);
  }
  for (int w = 0; w < 4; ++w) {
    __syncthreads();
    if (wave == w) {
#pragma unroll
      for (int j = 0; j < DV; ++j) {
        float* p0 = l_red + (j * 64 + lane) * 4;
        float* p1 = l_red + DV * 256 + (j * 64 + lane) * 4;
        st4(p0, w == 0 ? S_t.v[j] : ld4(p0) + S_t.v[j]);
        st4(p1, w == 0 ? S_i.v[j] : ld4(p1) + S_i.v[j]);
      }
    }
  }
  __syncthreads();
  float* out = a.s2_part + (b * a.chunks + blockIdx.x) * (int64_t)(2 * D);
  for (int i = threadIdx.x; i < 2 * D4; i += 256) {
    const int which = i / D4, c4 = i - which * D4;
    st4(out + which * D + c4 * 4, ld4(l_red + which * DV * 256 + c4 * 4));
  }
}

int launch_pair_layer1(const PairArgs& a, hipStream_t st) {
  if (a.B <= 0) return DRIN_OK;
  KernelTimer timer(DRIN_KC_GCN, st);
  if (a.act_v != DRIN_ACT_GELU && a.D4 <= 64)          // non-default vertex activation (args.py:35): the generic-width instantiations
    hipLaunchKernelGGL((k_pair_layer1<1, false, true>), dim3((unsigned)a.chunks, (unsigned)a.B), dim3(256), 0, st, a);
  else if (a.act_v != DRIN_ACT_GELU && a.D4 <= 192)
    hipLaunchKernelGGL((k_pair_layer1<3, false, true>), dim3((unsigned)a.chunks, (unsigned)a.B), dim3(256), 0, st, a);
  else if (a.D4 <= 64)
    hipLaunchKernelGGL((k_pair_layer1<1, false>), dim3((unsigned)a.chunks, (unsigned)a.B), dim3(256), 0, st, a);
  else if (a.D4 == 192)
    hipLaunchKernelGGL((k_pair_layer1<3, true>), dim3((unsigned)a.chunks, (unsigned)a.B), dim3(256), 0, st, a);
  else if (a.D4 < 192)
    hipLaunchKernelGGL((k_pair_layer1<3, false>), dim3((unsigned)a.chunks, (unsigned)a.B), dim3(256), 0, st, a);
  else {
    set_error("pair_layer1: D=%d outside the built instantiations", a.D4 * 4);
    return DRIN_E_UNSUPPORTED;
  }
  DRIN_CHECK_LAUNCH("k_pair_layer1");
  return DRIN_OK;
}

// ------------------------------------------------------------------------------------------------
// W_h input of the layer-2 mention-text vertex: out[b] = (sum_chunks (S2_t + S2_i)) / N + mt1[b]
__global__ void __launch_bounds__(256) k_mention_input2(const float* __restrict__ part, const float* __restrict__ mt1,
                                                        float* __restrict__ out, int D, int chunks, float inv_n) {
  const int64_t b = blockIdx.y;
  const int d = blockIdx.x * 256 + threadIdx.x;
  if (d >= D) return;
  float st = 0.f, si = 0.f;
  for (int c = 0; c < chunks; ++c) {
    const float* p = part + (b * chunks + c) * (int64_t)(2 * D);
    st += p[d];
    si += p[D + d];
  }
  out[b * D + d] = (st * inv_n + si * inv_n) + mt1[b * D + d];
}

int launch_mention_input2(const float* part, const float* mt1, float* out, int B, int D, int N, int chunks,
                          hipStream_t st) {
  if (B <= 0) return DRIN_OK;
  KernelTimer timer(DRIN_KC_GCN, st);
  hipLaunchKernelGGL(k_mention_input2, dim3((unsigned)cdiv(D, 256), (unsigned)B), dim3(256), 0, st, part, mt1, out, D,
                     chunks, 1.0f / (float)N);
  DRIN_CHECK_LAUNCH("k_mention_input2");
  return DRIN_OK;
}

// ------------------------------------------------------------------------------------------------
// Layer-2 entity-text vertex and the score (model.py:128 for et'', :207-209), grid (chunks, B), 256 threads:
//   et2 = gelu(LN(H2raw[p] + e1_tt hm2_t[b] + e1_it hm2_i[b] + b_h2)),  score[p] = cos(mt2[b], et2)
// The five per-mention / constant vectors live in LDS; each wave walks candidates of its chunk.
// (Measured in round 5 and dropped - profiles/r5_pair_final_two_rows_ab.txt: two candidates per wave and trip, so that one row's four
//  dependent wave reductions fill the other's waits.  133 VGPRs instead of 100 = three waves per SIMD instead of five: config 5's
//  3.39 ms became 3.75, the headline's row kernels 1.13 -> 1.17 ms.  Round 4's tries on the same kernel: single-round LayerNorm
//  statistics, the next row prefetched - profiles/r4_ln_stats_ab.txt, r4_pair_final_prefetch_ab.txt.)
template <int DV, bool EXACT, bool GENERIC_ACT = false>
__global__ void __launch_bounds__(256) k_pair_final(const FinalArgs a) {
  __shared__ __attribute__((aligned(16))) float l_const[6 * DV * 256];  // hm2_t, hm2_i, b_h2, gamma, beta, mt2
  const int D4 = EXACT ? DV * 64 : a.D4, D = D4 * 4;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t b = blockIdx.y;
  const int64_t M = (int64_t)a.B * a.N;
  const int per = (a.N + a.chunks - 1) / a.chunks;
  const int n_begin = blockIdx.x * per, n_end = min(a.N, n_begin + per);
  constexpr int LD = DV * 256;
  {
    const float* src[6] = {a.hm2 + b * D, a.hm2 + ((int64_t)a.B + b) * D, a.b_h2, a.gamma, a.beta, a.mt2 + b * D};
#pragma unroll
    for (int v = 0; v < 6; ++v)
      for (int i = threadIdx.x; i < D4; i += 256) st4(l_const + v * LD + i * 4, ld4(src[v] + i * 4));
  }
  __syncthreads();
  const Row<DV> mt2 = load_row<DV>(l_const + 5 * LD, lane, D4);
  const float xx = wave_sum(dot_rows<DV>(mt2, mt2));
  for (int n = n_begin + wave; n < n_end; n += 4) {
    const int64_t p = b * a.N + n;
#ifdef DRIN_ABLATE_ROW_TRAFFIC
    const Row<DV> h = load_row<DV>(a.h2 + (b * a.N + (n & 3)) * D, lane, D4);
#else
    const Row<DV> h = load_row_stream<DV>(a.h2 + p * D, lane, D4);
#endif
    const Row<DV> et2 = ln_gelu_row_lds<DV, GENERIC_ACT>(
        combine_rows_lds<DV>(h, a.e1m[p], l_const, a.e1m[2 * M + p], l_const + LD, l_const + 2 * LD, lane, D4),
        l_const + 3 * LD, l_const + 4 * LD, lane, D4, a.ln_eps, a.act_v);
    const float xy = wave_sum(dot_rows_pk<DV>(mt2, et2));
    const float yy = wave_sum(dot_rows_pk<DV>(et2, et2));
    if (lane == 0) a.scores[p] = cosine_from_sums(xy, xx, yy, a.cos_eps);
  }
}

int launch_pair_final(const FinalArgs& a, hipStream_t st) {
  if (a.B <= 0) return DRIN_OK;
  KernelTimer timer(DRIN_KC_GCN, st);
  if (a.act_v != DRIN_ACT_GELU && a.D4 <= 64)          // non-default vertex activation (args.py:35): the generic-width instantiations
    hipLaunchKernelGGL((k_pair_final<1, false, true>), dim3((unsigned)a.chunks, (unsigned)a.B), dim3(256), 0, st, a);
  else if (a.act_v != DRIN_ACT_GELU && a.D4 <= 192)
    hipLaunchKernelGGL((k_pair_final<3, false, true>), dim3((unsigned)a.chunks, (unsigned)a.B), dim3(256), 0, st, a);
  else if (a.D4 <= 64)
    hipLaunchKernelGGL((k_pair_final<1, false>), dim3((unsigned)a.chunks, (unsigned)a.B), dim3(256), 0, st, a);
  else if (a.D4 == 192)
    hipLaunchKernelGGL((k_pair_final<3, true>), dim3((unsigned)a.chunks, (unsigned)a.B), dim3(256), 0, st, a);
  else if (a.D4 < 192)
    hipLaunchKernelGGL((k_pair_final<3, false>), dim3((unsigned)a.chunks, (unsigned)a.B), dim3(256), 0, st, a);
  else {
    set_error("pair_final: D=%d outside the built instantiations", a.D4 * 4);
    return DRIN_E_UNSUPPORTED;
  }
  DRIN_CHECK_LAUNCH("k_pair_final");
  return DRIN_OK;
}

}  // namespace drin
