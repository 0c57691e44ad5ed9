// Stand-alone probe (not part of the library): how fast can one pass over 3 KB rows be consumed on MI355X
//   A  register path : global_load_dwordx4 (non-temporal), rows summed in VGPRs   - what k_entity_stream does today
//   B  LDS-DMA path  : global_load_lds dwordx4 into a per-wave LDS ring, rows summed from ds_read_b128
// Same access pattern as k_entity_stream: a workgroup owns 16 "pairs" of ROWS_PER_PAIR consecutive rows, a wave walks 4 of
// them, IN_FLIGHT rows requested ahead.  Prints GB/s of each.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/stream_probe.hip -o stream_probe && ./stream_probe [GiB]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x)                                                                   \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                      \
      exit(2);                                                                     \
    }                                                                              \
  } while (0)

#ifndef PROBE_IN_FLIGHT
#define PROBE_IN_FLIGHT 4
#endif
constexpr int ROW_FLOATS = 768, ROWS_PER_PAIR = 34, PAIRS_PER_WG = 16, IN_FLIGHT = PROBE_IN_FLIGHT;
// s_waitcnt immediate of vmcnt(n) with the other counters left alone (gfx9: vmcnt = simm16[15:14] : simm16[3:0])
constexpr int vmcnt_imm(int n) { return 0x0f70 | (n & 0xf) | (((n >> 4) & 3) << 14); }
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ldnt(const float* p) {
  const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
  return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

__global__ void __launch_bounds__(256) k_regs(const float* __restrict__ src, float* __restrict__ out, int64_t pairs) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int q = wave; q < PAIRS_PER_WG; q += 4) {
    const int64_t p = (int64_t)blockIdx.x * PAIRS_PER_WG + q;
    if (p >= pairs) return;
    const float* base = src + p * (int64_t)ROWS_PER_PAIR * ROW_FLOATS + lane * 4;
    float4 acc[3] = {make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0)};
    int r = 0;
    for (; r + IN_FLIGHT <= ROWS_PER_PAIR; r += IN_FLIGHT) {
      float4 v[IN_FLIGHT][3];
#pragma unroll
      for (int i = 0; i < IN_FLIGHT; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) v[i][j] = ldnt(base + (int64_t)(r + i) * ROW_FLOATS + j * 256);
#pragma unroll
      for (int i = 0; i < IN_FLIGHT; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[j] = add4(acc[j], v[i][j]);
    }
    for (; r < ROWS_PER_PAIR; ++r)
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[j] = add4(acc[j], ldnt(base + (int64_t)r * ROW_FLOATS + j * 256));
#pragma unroll
    for (int j = 0; j < 3; ++j) *reinterpret_cast<float4*>(out + p * ROW_FLOATS + lane * 4 + j * 256) = acc[j];
  }
}

// the same bytes with the four waves of a workgroup sharing each run: wave w takes rows w, w + 4, ... (partial sums per wave)
__global__ void __launch_bounds__(256) k_regs_shared(const float* __restrict__ src, float* __restrict__ out, int64_t pairs) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int q = 0; q < PAIRS_PER_WG; ++q) {
    const int64_t p = (int64_t)blockIdx.x * PAIRS_PER_WG + q;
    if (p >= pairs) return;
    const float* base = src + p * (int64_t)ROWS_PER_PAIR * ROW_FLOATS + lane * 4;
    float4 acc[3] = {make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0)};
    for (int r = wave; r < ROWS_PER_PAIR; r += 4)
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[j] = add4(acc[j], ldnt(base + (int64_t)r * ROW_FLOATS + j * 256));
    if (wave == 0)
#pragma unroll
      for (int j = 0; j < 3; ++j) *reinterpret_cast<float4*>(out + p * ROW_FLOATS + lane * 4 + j * 256) = acc[j];
  }
}

// per wave: ring of 2 * IN_FLIGHT rows of 3 KB.  Half h is requested while half h ^ 1 is summed.
template <int AUX>
__global__ void __launch_bounds__(256) k_dma(const float* __restrict__ src, float* __restrict__ out, int64_t pairs) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int ROW_BYTES = ROW_FLOATS * 4, HALF = IN_FLIGHT * ROW_BYTES;
  char* ring = smem + wave * 2 * HALF;
  for (int q = wave; q < PAIRS_PER_WG; q += 4) {
    const int64_t p = (int64_t)blockIdx.x * PAIRS_PER_WG + q;
    if (p >= pairs) return;
    const char* base = reinterpret_cast<const char*>(src + p * (int64_t)ROWS_PER_PAIR * ROW_FLOATS) + lane * 16;
    float4 acc[3] = {make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0)};
    auto issue = [&](int r0, int half) {  // rows r0 .. r0 + IN_FLIGHT - 1 (clamped: re-reads of the last row are discarded)
#pragma unroll
      for (int i = 0; i < IN_FLIGHT; ++i) {
        const int r = r0 + i < ROWS_PER_PAIR ? r0 + i : ROWS_PER_PAIR - 1;
#pragma unroll
        for (int j = 0; j < 3; ++j)
          __builtin_amdgcn_global_load_lds((gptr_t)(base + (int64_t)r * ROW_BYTES + j * 1024),
                                           (lptr_t)(ring + half * HALF + i * ROW_BYTES + j * 1024), 16, 0, AUX);
      }
    };
    issue(0, 0);
    int half = 0;
    for (int r = 0; r < ROWS_PER_PAIR; r += IN_FLIGHT, half ^= 1) {
      if (r + IN_FLIGHT < ROWS_PER_PAIR) {
        issue(r + IN_FLIGHT, half ^ 1);
        __builtin_amdgcn_s_waitcnt(vmcnt_imm(IN_FLIGHT * 3));  // the half requested one trip ago has landed
      } else {
        __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < IN_FLIGHT; ++i) {
        if (r + i >= ROWS_PER_PAIR) break;
#pragma unroll
        for (int j = 0; j < 3; ++j)
          acc[j] = add4(acc[j], *reinterpret_cast<const float4*>(ring + half * HALF + i * ROW_BYTES + j * 1024 + lane * 16));
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the ring half is free before it is requested again
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) *reinterpret_cast<float4*>(out + p * ROW_FLOATS + lane * 4 + j * 256) = acc[j];
  }
}

template <typename F>
static double time_ms(F launch, int iters) {
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  launch();
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(a));
  for (int i = 0; i < iters; ++i) launch();
  CHECK(hipEventRecord(b));
  CHECK(hipEventSynchronize(b));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, a, b));
  return ms / iters;
}

int main(int argc, char** argv) {
  const double gib = argc > 1 ? atof(argv[1]) : 16.0;
  const int64_t pair_bytes = (int64_t)ROWS_PER_PAIR * ROW_FLOATS * 4;
  int64_t pairs = (int64_t)(gib * (1ll << 30)) / pair_bytes;
  pairs -= pairs % PAIRS_PER_WG;
  float *src = nullptr, *out = nullptr;
  CHECK(hipMalloc((void**)&src, pairs * pair_bytes));
  CHECK(hipMalloc((void**)&out, pairs * ROW_FLOATS * 4));
  CHECK(hipMemset(src, 0, pairs * pair_bytes));
  const unsigned grid = (unsigned)(pairs / PAIRS_PER_WG);
  const size_t lds = 4 * 2 * IN_FLIGHT * ROW_FLOATS * 4;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_dma<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_dma<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const double bytes = (double)pairs * pair_bytes;
  const double t_regs = time_ms([&] { hipLaunchKernelGGL(k_regs, dim3(grid), dim3(256), 0, 0, src, out, pairs); }, 5);
  const double t_dma0 = time_ms([&] { hipLaunchKernelGGL(k_dma<0>, dim3(grid), dim3(256), lds, 0, src, out, pairs); }, 5);
  const double t_dma2 = time_ms([&] { hipLaunchKernelGGL(k_dma<2>, dim3(grid), dim3(256), lds, 0, src, out, pairs); }, 5);
  printf("%.1f GiB, %lld pairs of %d rows x 3 KB, %d rows in flight per wave\n", bytes / (1ll << 30), (long long)pairs, ROWS_PER_PAIR,
         IN_FLIGHT);
  const double t_shared = time_ms([&] { hipLaunchKernelGGL(k_regs_shared, dim3(grid), dim3(256), 0, 0, src, out, pairs); }, 5);
  printf("registers (nt loads)     : %7.3f ms  %7.1f GB/s\n", t_regs, bytes / t_regs / 1e6);
  printf("registers, runs shared by the 4 waves: %7.3f ms  %7.1f GB/s\n", t_shared, bytes / t_shared / 1e6);
  printf("LDS-DMA (default policy) : %7.3f ms  %7.1f GB/s\n", t_dma0, bytes / t_dma0 / 1e6);
  printf("LDS-DMA (aux = 2, nt)    : %7.3f ms  %7.1f GB/s\n", t_dma2, bytes / t_dma2 / 1e6);
  CHECK(hipFree(src));
  CHECK(hipFree(out));
  return 0;
}
