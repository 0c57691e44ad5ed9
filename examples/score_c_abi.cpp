// A caller of libdrin_hip.so that knows nothing about PyTorch: plain HIP allocations, the C ABI of include/drin_hip.h.
//
//   hipcc --offload-arch=gfx950 -O2 -Iinclude examples/score_c_abi.cpp -Ldrin_amd -ldrin_hip -Wl,-rpath,$PWD/drin_amd -o score_c_abi
//   ./score_c_abi case.bin scores.bin [f32|bf16x3]
//
// case.bin (written by tests/test_gpu_parity.py::test_c_abi_caller_without_torch): 12 int32 of geometry
// (B N D R L P Km Ke T dynamic layers reserved), then the 13 batch tensors in drin_batch order (fp32 / int64), then
// the parameters in the order of drin_params (vertex encoder, then per layer w_h b_h w_u b_u w_v b_v ln_w ln_b).
// scores.bin: B*N fp32 from drin_forward (layer by layer) followed by B*N fp32 from drin_prepare + drin_forward_prepared.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "drin_hip.h"

#define HIP_OK(x)                                                                  \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                      \
      return 2;                                                                    \
    }                                                                              \
  } while (0)
#define DRIN_OK_(x)                                                                \
  do {                                                                             \
    int s_ = (x);                                                                  \
    if (s_ != DRIN_OK) {                                                           \
      fprintf(stderr, "%s -> %d: %s\n", #x, s_, drin_last_error());                \
      return 3;                                                                    \
    }                                                                              \
  } while (0)

static std::vector<void*> g_allocs;

// reads `bytes` from the file into a fresh device buffer
static void* upload(FILE* f, size_t bytes) {
  std::vector<char> host(bytes);
  if (fread(host.data(), 1, bytes, f) != bytes) {
    fprintf(stderr, "case file truncated\n");
    exit(4);
  }
  void* d = nullptr;
  if (hipMalloc(&d, bytes ? bytes : 16) != hipSuccess || hipMemcpy(d, host.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) {
    fprintf(stderr, "hipMalloc / hipMemcpy failed\n");
    exit(5);
  }
  g_allocs.push_back(d);
  return d;
}

int main(int argc, char** argv) {
  if (argc < 3) {
    fprintf(stderr, "usage: %s case.bin scores.bin [f32|bf16x3]\n", argv[0]);
    return 1;
  }
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 1;
  int32_t g[12];
  if (fread(g, sizeof(int32_t), 12, f) != 12) return 1;
  const int64_t B = g[0], N = g[1], D = g[2], R = g[3], L = g[4], P = g[5], Km = g[6], Ke = g[7], T = g[8];
  const int64_t M = B * N;

  drin_config cfg;
  DRIN_OK_(drin_default_config(&cfg));
  cfg.batch = (int32_t)B;
  cfg.num_candidates = (int32_t)N;
  cfg.embed_dim = (int32_t)D;
  cfg.image_dim = (int32_t)R;
  cfg.mention_tokens = (int32_t)L;
  cfg.image_regions = (int32_t)P;
  cfg.mention_objects = (int32_t)Km;
  cfg.entity_objects = (int32_t)Ke;
  cfg.entity_tokens = (int32_t)T;
  cfg.dynamic_edges = g[9];
  cfg.num_layers = g[10];
  cfg.precision = (argc > 3 && strcmp(argv[3], "bf16x3") == 0) ? DRIN_PREC_BF16X3_ALL : DRIN_PREC_F32;

  drin_batch b;
  memset(&b, 0, sizeof(b));
  b.mention_text = (const float*)upload(f, B * L * D * 4);
  b.mention_start = (const int64_t*)upload(f, B * 8);
  b.mention_end = (const int64_t*)upload(f, B * 8);
  b.mention_image = (const float*)upload(f, B * P * R * 4);
  b.mention_object = (const float*)upload(f, B * Km * R * 4);
  b.mention_object_score = (const float*)upload(f, B * Km * 4);
  b.entity_text = (const float*)upload(f, (T > 0 ? M * T * D : M * D) * 4);
  b.entity_text_mask = T > 0 ? (const int64_t*)upload(f, M * T * 8) : nullptr;
  b.entity_image = (const float*)upload(f, M * R * 4);
  b.entity_object = (const float*)upload(f, M * Ke * R * 4);
  b.entity_object_score = (const float*)upload(f, M * Ke * 4);
  b.miet_similarity = (const float*)upload(f, M * 4);
  b.mtei_similarity = (const float*)upload(f, M * 4);

  drin_params p;
  memset(&p, 0, sizeof(p));
  p.w_mention_text = (const float*)upload(f, D * D * 4);
  p.b_mention_text = (const float*)upload(f, D * 4);
  p.w_entity_text = (const float*)upload(f, D * D * 4);
  p.b_entity_text = (const float*)upload(f, D * 4);
  p.w_mention_image = (const float*)upload(f, D * R * 4);
  p.b_mention_image = (const float*)upload(f, D * 4);
  p.w_entity_image = (const float*)upload(f, D * R * 4);
  p.b_entity_image = (const float*)upload(f, D * 4);
  for (int l = 0; l < cfg.num_layers; ++l) {
    p.layer[l].w_h = (const float*)upload(f, D * D * 4);
    p.layer[l].b_h = (const float*)upload(f, D * 4);
    p.layer[l].w_u = (const float*)upload(f, D * D * 4);
    p.layer[l].b_u = (const float*)upload(f, D * 4);
    p.layer[l].w_v = (const float*)upload(f, D * D * 4);
    p.layer[l].b_v = (const float*)upload(f, D * 4);
    p.layer[l].ln_weight = (const float*)upload(f, D * 4);
    p.layer[l].ln_bias = (const float*)upload(f, D * 4);
  }
  fclose(f);

  hipStream_t st;
  HIP_OK(hipStreamCreate(&st));
  float *scores_a = nullptr, *scores_b = nullptr;
  HIP_OK(hipMalloc((void**)&scores_a, M * 4));
  HIP_OK(hipMalloc((void**)&scores_b, M * 4));

  // (1) the layer-by-layer path: caller-owned workspace, one call
  const size_t ws_bytes = drin_workspace_bytes(&cfg, 0);
  void* ws = nullptr;
  HIP_OK(hipMalloc(&ws, ws_bytes));
  DRIN_OK_(drin_forward(&cfg, &b, &p, ws, ws_bytes, scores_a, 0, nullptr, st));

  // (2) the folded inference path: weights folded once, then any number of batches
  int have_b = 0;
  if (drin_fused_supported(&cfg) == DRIN_OK) {
    const size_t pb = drin_prepared_bytes(&cfg), fw = drin_fused_workspace_bytes(&cfg);
    void *prepared = nullptr, *fws = nullptr;
    HIP_OK(hipMalloc(&prepared, pb));
    HIP_OK(hipMalloc(&fws, fw));
    DRIN_OK_(drin_prepare(&cfg, &p, prepared, pb, st));
    DRIN_OK_(drin_forward_prepared(&cfg, &b, &p, prepared, fws, fw, scores_b, st));
    have_b = 1;
    HIP_OK(hipStreamSynchronize(st));
    HIP_OK(hipFree(prepared));
    HIP_OK(hipFree(fws));
  }
  HIP_OK(hipStreamSynchronize(st));

  std::vector<float> out(2 * M, 0.f);
  HIP_OK(hipMemcpy(out.data(), scores_a, M * 4, hipMemcpyDeviceToHost));
  if (have_b) HIP_OK(hipMemcpy(out.data() + M, scores_b, M * 4, hipMemcpyDeviceToHost));
  FILE* o = fopen(argv[2], "wb");
  if (!o || fwrite(out.data(), 4, out.size(), o) != out.size()) return 6;
  fclose(o);
  printf("%s: %lld x %lld scores, layer-by-layer[0]=%.7f folded[0]=%.7f\n", drin_build_info(), (long long)B, (long long)N, out[0],
         have_b ? out[M] : 0.f);
  for (void* d : g_allocs) (void)hipFree(d);
  (void)hipFree(ws);
  (void)hipFree(scores_a);
  (void)hipFree(scores_b);
  return 0;
}
