// A caller of libdrin_hip.so that knows nothing about PyTorch: plain HIP allocations, the C ABI of include/drin_hip.h.
//
//   hipcc --offload-arch=gfx950 -O2 -Iinclude examples/score_c_abi.cpp -Ldrin_amd -ldrin_hip -Wl,-rpath,$PWD/drin_amd -o score_c_abi
//   ./score_c_abi case.bin scores.bin [f32|bf16x3]
//
// case.bin (written by tests/test_gpu_parity.py::test_c_abi_caller_without_torch): 12 int32 of geometry
// (B N D R L P Km Ke T dynamic layers reserved), then the 13 batch tensors in drin_batch order (fp32 / int64), then
// the parameters in the order of drin_params (vertex encoder, then per layer w_h b_h w_u b_u w_v b_v ln_w ln_b).
// When the 12th geometry word is 1, the answer tensor [B, N-1] uint8 follows the parameters.
// scores.bin: B*N fp32 from drin_forward (layer by layer) followed by B*N fp32 from drin_prepare + drin_forward_prepared;
// with an answer tensor then also one training step's device work (train.py:30-37): the TripletLoss value (1 fp32), the
// top-1 hit count (1 fp32), the gradient of every parameter tensor in drin_params order (fp32; dead ones all zero) and,
// after the Adam step of train.py:55-56 through drin_adam_step, every parameter tensor again (those without a gradient
// unchanged) - a whole training step with no torch and no Python.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
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
  const uint8_t* answer = g[11] == 1 ? (const uint8_t*)upload(f, (size_t)(B * (N - 1))) : nullptr;
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

  // (3) the device work of one training step (train.py:30-37): forward keeping what backward reads, TripletLoss +
  //     top-1 + d loss / d scores in one call, backward into caller-owned gradient buffers
  if (answer != nullptr) {
    const size_t tw = drin_workspace_bytes(&cfg, 1), lw = drin_loss_workspace_bytes((int32_t)B);
    void *tws = nullptr, *lws = nullptr;
    float *scores_t = nullptr, *loss = nullptr, *d_scores = nullptr;
    int64_t* correct = nullptr;
    HIP_OK(hipMalloc(&tws, tw));
    HIP_OK(hipMalloc(&lws, lw));
    HIP_OK(hipMalloc((void**)&scores_t, M * 4));
    HIP_OK(hipMalloc((void**)&loss, 4));
    HIP_OK(hipMalloc((void**)&d_scores, M * 4));
    HIP_OK(hipMalloc((void**)&correct, 8));
    HIP_OK(hipMemsetAsync(correct, 0, 8, st));
    // gradient buffers: one zeroed arena, carved in drin_params order
    std::vector<size_t> sizes = {(size_t)(D * D), (size_t)D, (size_t)(D * D), (size_t)D, (size_t)(D * R), (size_t)D, (size_t)(D * R), (size_t)D};
    for (int l = 0; l < cfg.num_layers; ++l)
      for (size_t n : {(size_t)(D * D), (size_t)D, (size_t)(D * D), (size_t)D, (size_t)(D * D), (size_t)D, (size_t)D, (size_t)D}) sizes.push_back(n);
    size_t total = 0;
    for (size_t n : sizes) total += (n + 63) & ~(size_t)63;
    float* arena = nullptr;
    HIP_OK(hipMalloc((void**)&arena, total * 4));
    HIP_OK(hipMemsetAsync(arena, 0, total * 4, st));
    std::vector<float*> gp;
    {
      size_t off = 0;
      for (size_t n : sizes) {
        gp.push_back(arena + off);
        off += (n + 63) & ~(size_t)63;
      }
    }
    drin_param_grads gr;
    memset(&gr, 0, sizeof(gr));
    gr.w_mention_text = gp[0];
    gr.b_mention_text = gp[1];
    gr.w_entity_text = gp[2];
    gr.b_entity_text = gp[3];
    gr.w_mention_image = gp[4];
    gr.b_mention_image = gp[5];
    gr.w_entity_image = gp[6];
    gr.b_entity_image = gp[7];
    for (int l = 0; l < cfg.num_layers; ++l) {
      float** q = &gp[8 + 8 * l];
      gr.layer[l].w_h = q[0];
      gr.layer[l].b_h = q[1];
      // the last layer's edge update never reaches the score (model.py:130-134): its w_u / w_v get no gradient
      const bool live = cfg.dynamic_edges && l + 1 < cfg.num_layers;
      gr.layer[l].w_u = live ? q[2] : nullptr;
      gr.layer[l].b_u = live ? q[3] : nullptr;
      gr.layer[l].w_v = live ? q[4] : nullptr;
      gr.layer[l].b_v = live ? q[5] : nullptr;
      gr.layer[l].ln_weight = q[6];
      gr.layer[l].ln_bias = q[7];
    }
    const int32_t topk[1] = {1};
    DRIN_OK_(drin_forward(&cfg, &b, &p, tws, tw, scores_t, 1, nullptr, st));
    DRIN_OK_(drin_triplet_topk(scores_t, answer, (int32_t)B, (int32_t)N, 0.25f, topk, 1, loss, d_scores, correct, lws, lw, st));
    DRIN_OK_(drin_backward(&cfg, &b, &p, tws, tw, d_scores, &gr, st));
    HIP_OK(hipStreamSynchronize(st));
    float h_loss = 0.f;
    int64_t h_correct = 0;
    HIP_OK(hipMemcpy(&h_loss, loss, 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(&h_correct, correct, 8, hipMemcpyDeviceToHost));
    out.push_back(h_loss);
    out.push_back((float)h_correct);
    for (size_t i = 0; i < sizes.size(); ++i) {
      const size_t at = out.size();
      out.resize(at + sizes[i]);
      HIP_OK(hipMemcpy(out.data() + at, gp[i], sizes[i] * 4, hipMemcpyDeviceToHost));
    }
    printf("training step: loss %.7f, top-1 hits %lld of %lld\n", h_loss, (long long)h_correct, (long long)B);
    // (4) the optimiser step of train.py:55-56 (torch.optim.Adam, lr 1e-3, torch defaults), first step: drin_adam_step per
    //     tensor that received a gradient, zero moments in one arena laid out like the gradients.  The scalars are formed
    //     in double on the host exactly as torch/optim/adam.py forms them (include/drin_hip.h).
    {
      float* moments = nullptr;
      HIP_OK(hipMalloc((void**)&moments, 2 * total * 4));
      HIP_OK(hipMemsetAsync(moments, 0, 2 * total * 4, st));
      const float* pw[8 + 8 * DRIN_MAX_LAYERS] = {p.w_mention_text,  p.b_mention_text,  p.w_entity_text,  p.b_entity_text,
                                                  p.w_mention_image, p.b_mention_image, p.w_entity_image, p.b_entity_image};
      for (int l = 0; l < cfg.num_layers; ++l) {
        const drin_layer_params& q = p.layer[l];
        const float* row[8] = {q.w_h, q.b_h, q.w_u, q.b_u, q.w_v, q.b_v, q.ln_weight, q.ln_bias};
        for (int j = 0; j < 8; ++j) pw[8 + 8 * l + j] = row[j];
      }
      const double lr = 1e-3, beta1 = 0.9, beta2 = 0.999, eps = 1e-8, step = 1.0;
      const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);
      size_t off = 0;
      for (size_t i = 0; i < sizes.size(); ++i) {
        const int l = i < 8 ? -1 : (int)(i - 8) / 8, j = i < 8 ? -1 : (int)(i - 8) % 8;
        const bool dead = l >= 0 && j >= 2 && j <= 5 && !(cfg.dynamic_edges && l + 1 < cfg.num_layers);   // no gradient: torch's Adam skips it
        if (!dead)
          DRIN_OK_(drin_adam_step(const_cast<float*>(pw[i]), gp[i], moments + off, moments + total + off, (int64_t)sizes[i],
                                  (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)sqrt(bc2), (float)eps,
                                  (float)((lr / bc1) * -1.0), st));
        off += (sizes[i] + 63) & ~(size_t)63;
      }
      HIP_OK(hipStreamSynchronize(st));
      for (size_t i = 0; i < sizes.size(); ++i) {
        const size_t at = out.size();
        out.resize(at + sizes[i]);
        HIP_OK(hipMemcpy(out.data() + at, pw[i], sizes[i] * 4, hipMemcpyDeviceToHost));
      }
      (void)hipFree(moments);
    }
    for (void* d : {tws, lws, (void*)scores_t, (void*)loss, (void*)d_scores, (void*)correct, (void*)arena}) (void)hipFree(d);
  }
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
