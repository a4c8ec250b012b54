// chain_bench — torch-free micro-benchmark of the decode GEMV launches of libns_hip.so on the Llama-2-7B shapes
// (int4 sym g32 bf16 scales): per-shape launch times, the whole chain in one hipGraph, and an A/B + equality check
// of the decode kernel generations (ns_hip_set_tuning("gemv2", ...)).  Build: scripts/ubench/build_chain_bench.sh
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "ns_bestla.h"

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(2);                                                                 \
    }                                                                          \
  } while (0)
#define NSCK(x)                                                                \
  do {                                                                         \
    if ((x) != 0) {                                                            \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, ns_hip_last_error()); \
      exit(3);                                                                 \
    }                                                                          \
  } while (0)

static const uint32_t S4 = 4 | (1u << 8), BF16 = 16 | (1u << 16);

__global__ void fill_kernel(float* p, size_t n, uint32_t seed, float scale) {
  size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t x = uint32_t(i) * 2654435761u ^ seed;
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  uint32_t y = x * 747796405u + 2891336453u;
  y ^= y >> 16; y *= 0x7feb352dU; y ^= y >> 15;
  // sum of two uniforms, roughly bell shaped, zero mean
  const float u = (float(x >> 8) + float(y >> 8)) * (1.0f / 16777216.0f) - 1.0f;
  p[i] = u * scale;
}
__global__ void to_half_kernel(const float* a, _Float16* h, size_t n) {
  size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) h[i] = (_Float16)a[i];
}

static hipStream_t g_st;
static float* g_tmp = nullptr;  // fp32 staging for one weight
static size_t g_tmp_elems = 0;

static ns_weight* make_weight(int n, int k, uint32_t seed, float scale, int group) {
  const size_t elems = size_t(n) * k;
  if (elems > g_tmp_elems) {
    if (g_tmp) CK(hipFree(g_tmp));
    CK(hipMalloc((void**)&g_tmp, elems * 4));
    g_tmp_elems = elems;
  }
  fill_kernel<<<dim3((elems + 255) / 256), dim3(256), 0, g_st>>>(g_tmp, elems, seed, scale);
  const size_t size = ns_BTLAGemmPackBSize(n, k, group, S4, BF16, false, NS_COMP_INT8, nullptr);
  void* blob = nullptr;
  CK(hipMalloc(&blob, size));
  CK(hipMemsetAsync(blob, 0, size, g_st));
  NSCK(ns_hip_quant_pack_device(blob, g_tmp, n, k, k, group, S4, BF16, false, NS_COMP_INT8, true, g_st));
  ns_weight* w = ns_hip_weight_from_device_blob(blob, size, g_st);
  if (!w) {
    fprintf(stderr, "weight load failed: %s\n", ns_hip_last_error());
    exit(4);
  }
  CK(hipStreamSynchronize(g_st));
  CK(hipFree(blob));
  return w;
}

struct Layer {
  ns_weight *q, *k, *v, *o, *w1, *w3, *w2;
};

template <typename F>
static hipGraphExec_t capture(F&& body) {
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(g_st, hipStreamCaptureModeThreadLocal));
  body();
  CK(hipStreamEndCapture(g_st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipGraphDestroy(g));
  return ge;
}
static double time_graph(hipGraphExec_t ge, int reps, int warm = 3) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < warm; i++) CK(hipGraphLaunch(ge, g_st));
  CK(hipStreamSynchronize(g_st));
  CK(hipEventRecord(e0, g_st));
  for (int i = 0; i < reps; i++) CK(hipGraphLaunch(ge, g_st));
  CK(hipEventRecord(e1, g_st));
  CK(hipStreamSynchronize(g_st));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipEventDestroy(e0));
  CK(hipEventDestroy(e1));
  return double(ms) * 1e3 / reps;  // us per replay
}

int main(int argc, char** argv) {
  int L = 32, d = 4096, ff = 11008, V = 32000, group = 32, reps = 20, tp = 1;
  std::vector<int> modes = {0, 1};
  bool shadow = true, run_chain = false;
  for (int i = 1; i < argc; i++) {
    if (!strcmp(argv[i], "--layers")) L = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--reps")) reps = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--tp")) tp = atoi(argv[++i]);  // per-rank shard shapes of tensor-parallel `tp`
    else if (!strcmp(argv[i], "--no-shadow")) shadow = false;
    else if (!strcmp(argv[i], "--chain")) run_chain = true;
    else if (!strcmp(argv[i], "--modes")) {
      modes.clear();
      for (char* t = strtok(argv[++i], ","); t; t = strtok(nullptr, ",")) modes.push_back(atoi(t));
    }
  }
  CK(hipSetDevice(0));
  CK(hipStreamCreate(&g_st));
  const int dl = d / tp, ffl = ff / tp;
  std::vector<Layer> layers(L);
  for (int il = 0; il < L; il++) {
    const uint32_t s = 1000 + il * 8;
    const float sd = 1.0f / sqrtf(float(d)) * 1.7f, sf = 1.0f / sqrtf(float(ff)) * 1.7f / 0.6f;
    layers[il] = {make_weight(dl, d, s + 0, sd, group), make_weight(dl, d, s + 1, sd, group),
                  make_weight(dl, d, s + 2, sd, group), make_weight(d, dl, s + 3, sd, group),
                  make_weight(ffl, d, s + 4, sd, group), make_weight(ffl, d, s + 5, sd, group),
                  make_weight(d, ffl, s + 6, sf, group)};
  }
  ns_weight* head = make_weight(V, d, 999, 1.0f / sqrtf(float(d)) * 1.7f, group);
  uint64_t wbytes = ns_hip_weight_stream_bytes(head);
  for (auto& l : layers)
    for (ns_weight* w : {l.q, l.k, l.v, l.o, l.w1, l.w3, l.w2}) wbytes += ns_hip_weight_stream_bytes(w);

  float *x0, *x, *qkv, *attn, *t2, *logits;
  _Float16 *x0h, *xh, *qkvh, *attnh, *t2h;
  CK(hipMalloc((void**)&x0, d * 4)); CK(hipMalloc((void**)&x, d * 4)); CK(hipMalloc((void**)&qkv, 3 * dl * 4));
  CK(hipMalloc((void**)&attn, d * 4)); CK(hipMalloc((void**)&t2, ffl * 4)); CK(hipMalloc((void**)&logits, V * 4));
  CK(hipMalloc((void**)&x0h, d * 2)); CK(hipMalloc((void**)&xh, d * 2)); CK(hipMalloc((void**)&qkvh, 3 * dl * 2));
  CK(hipMalloc((void**)&attnh, d * 2)); CK(hipMalloc((void**)&t2h, ffl * 2));
  fill_kernel<<<dim3((d + 255) / 256), dim3(256), 0, g_st>>>(x0, d, 7, 1.7f);
  to_half_kernel<<<dim3((d + 255) / 256), dim3(256), 0, g_st>>>(x0, x0h, d);
  CK(hipStreamSynchronize(g_st));
  auto H = [&](_Float16* p) -> void* { return shadow ? (void*)p : nullptr; };

  auto op_qkv = [&](const Layer& l, const float* in, _Float16* inh) {
    NSCK(ns_hip_fusion_qkv_forward_h(in, H(inh), l.q, l.k, l.v, qkv, H(qkvh), 1, d, dl, g_st));
  };
  auto op_wo = [&](const Layer& l) {
    NSCK(ns_hip_f32f32_forward_h(qkv, H(qkvh), l.o, attn, H(attnh), 1, dl, d, NS_EPI_NONE, nullptr, 0, g_st));
  };
  auto op_gu = [&](const Layer& l) {
    NSCK(ns_hip_fusion_ffn3_gateup_h(attn, H(attnh), l.w1, l.w3, nullptr, t2, H(t2h), 1, NS_EPI_SILU, g_st));
  };
  auto op_dn = [&](const Layer& l) {
    NSCK(ns_hip_f32f32_forward_h(t2, H(t2h), l.w2, x, H(xh), 1, ffl, d, NS_EPI_NONE, nullptr, 0, g_st));
  };
  auto op_head = [&](const float* in, _Float16* inh) {
    NSCK(ns_hip_f32f32_forward_h(in, H(inh), head, logits, nullptr, 1, d, V, NS_EPI_NONE, nullptr, 0, g_st));
  };
  auto chain = [&]() {
    const float* in = x0;
    _Float16* inh = x0h;
    for (auto& l : layers) {
      op_qkv(l, in, inh);
      op_wo(l);
      op_gu(l);
      op_dn(l);
      in = x;
      inh = xh;
    }
    op_head(in, inh);
  };

  // reference outputs from mode 0 (first-generation kernel): logits of the whole chain + every op of layer 0
  std::vector<std::vector<float>> ref;
  auto snapshot = [&]() {
    std::vector<std::vector<float>> out;
    auto grab = [&](const float* p, size_t n) {
      std::vector<float> h(n);
      CK(hipMemcpy(h.data(), p, n * 4, hipMemcpyDeviceToHost));
      out.push_back(h);
    };
    const Layer& l = layers[0];
    op_qkv(l, x0, x0h); CK(hipStreamSynchronize(g_st)); grab(qkv, 3 * dl);
    op_wo(l); CK(hipStreamSynchronize(g_st)); grab(attn, d);
    op_gu(l); CK(hipStreamSynchronize(g_st)); grab(t2, ffl);
    op_dn(l); CK(hipStreamSynchronize(g_st)); grab(x, d);
    chain(); CK(hipStreamSynchronize(g_st)); grab(logits, V);
    return out;
  };
  auto rel_l2 = [](const std::vector<float>& a, const std::vector<float>& b) {
    double num = 0, den = 0;
    for (size_t i = 0; i < a.size(); i++) {
      num += double(a[i] - b[i]) * double(a[i] - b[i]);
      den += double(b[i]) * double(b[i]);
    }
    return den > 0 ? sqrt(num / den) : sqrt(num);
  };

  printf("{\"layers\": %d, \"tp\": %d, \"weights_bytes\": %llu, \"shadow\": %s, \"runs\": [\n", L, tp,
         (unsigned long long)wbytes, shadow ? "true" : "false");
  bool first_run = true;
  for (int mode : modes) {
    ns_hip_set_tuning("gemv2", mode);
    auto snap = snapshot();
    if (ref.empty()) ref = snap;
    const char* names[5] = {"qkv", "wo", "gateup", "down", "logits(chain)"};
    std::string diffs;
    for (size_t i = 0; i < snap.size(); i++) {
      char buf[96];
      bool same = memcmp(snap[i].data(), ref[i].data(), snap[i].size() * 4) == 0;
      snprintf(buf, sizeof buf, "%s\"%s\": %.3g%s", i ? ", " : "", names[i], rel_l2(snap[i], ref[i]), same ? "" : "");
      diffs += buf;
    }
    // per-shape: one launch per layer (each layer's own weights: nothing cache resident)
    hipGraphExec_t gq = capture([&] { for (auto& l : layers) op_qkv(l, x0, x0h); });
    hipGraphExec_t go = capture([&] { for (auto& l : layers) op_wo(l); });
    hipGraphExec_t gg = capture([&] { for (auto& l : layers) op_gu(l); });
    hipGraphExec_t gd = capture([&] { for (auto& l : layers) op_dn(l); });
    hipGraphExec_t gh = capture([&] { op_head(x0, x0h); });
    hipGraphExec_t gc = capture(chain);
    const double tq = time_graph(gq, reps) / L, to = time_graph(go, reps) / L, tg = time_graph(gg, reps) / L,
                 td = time_graph(gd, reps) / L, th = time_graph(gh, reps), tc = time_graph(gc, reps);
    printf("%s  {\"gemv2\": %d, \"us\": {\"qkv\": %.2f, \"wo\": %.2f, \"gateup\": %.2f, \"down\": %.2f, \"head\": %.2f}, "
           "\"layer_us\": %.2f, \"chain_us\": %.1f, \"tok_s\": %.1f, \"chain_GBps\": %.0f, \"rel_l2_vs_first_mode\": {%s}}",
           first_run ? "" : ",\n", mode, tq, to, tg, td, th, tq + to + tg + td, tc, 1e6 / tc, double(wbytes) / tc / 1e3,
           diffs.c_str());
    first_run = false;
    fflush(stdout);
    for (hipGraphExec_t g : {gq, go, gg, gd, gh, gc}) CK(hipGraphExecDestroy(g));
  }
  // ---- the whole chain as ONE persistent launch (experiments/ns_chain.hip; build with -DNS_HAVE_CHAIN) ----
#ifdef NS_HAVE_CHAIN
  if (run_chain) {
    std::vector<ns_chain_op> ops;
    const void* in = x0h;
    for (auto& l : layers) {
      ns_chain_op q{};
      q.mode = NS_CHAIN_MSEG, q.nmat = 3, q.epilogue = NS_EPI_NONE;
      q.w[0] = l.q, q.w[1] = l.k, q.w[2] = l.v;
      q.in16 = in;
      q.out16[0] = qkvh, q.out16[1] = qkvh + dl, q.out16[2] = qkvh + 2 * dl;
      ops.push_back(q);
      ns_chain_op o{};
      o.mode = NS_CHAIN_PLAIN, o.epilogue = NS_EPI_NONE, o.w[0] = l.o, o.in16 = qkvh, o.out16[0] = attnh;
      ops.push_back(o);
      ns_chain_op gu{};
      gu.mode = NS_CHAIN_DUAL, gu.epilogue = NS_EPI_SILU, gu.w[0] = l.w1, gu.w[1] = l.w3, gu.in16 = attnh, gu.out16[0] = t2h;
      ops.push_back(gu);
      ns_chain_op dn{};
      dn.mode = NS_CHAIN_PLAIN, dn.epilogue = NS_EPI_NONE, dn.w[0] = l.w2, dn.in16 = t2h, dn.out16[0] = xh;
      ops.push_back(dn);
      in = xh;
    }
    ns_chain_op hd{};
    hd.mode = NS_CHAIN_PLAIN, hd.epilogue = NS_EPI_NONE, hd.w[0] = head, hd.in16 = in, hd.out32[0] = logits;
    ops.push_back(hd);
    ns_chain* ch = ns_hip_chain_create(ops.data(), int(ops.size()));
    if (!ch) {
      fprintf(stderr, "chain create failed: %s\n", ns_hip_last_error());
      return 5;
    }
    CK(hipMemsetAsync(logits, 0xff, size_t(V) * 4, g_st));  // poison
    NSCK(ns_hip_chain_run(ch, g_st));
    CK(hipStreamSynchronize(g_st));
    const int err = ns_hip_chain_error(ch);
    std::vector<float> lg(V);
    CK(hipMemcpy(lg.data(), logits, size_t(V) * 4, hipMemcpyDeviceToHost));
    const double diff = ref.empty() ? -1.0 : rel_l2(lg, ref.back());
    hipGraphExec_t gch = capture([&] { NSCK(ns_hip_chain_run(ch, g_st)); });
    const double tch = time_graph(gch, reps);
    printf(",\n  {\"persistent_chain\": true, \"hand_off_error\": %d, \"chain_us\": %.1f, \"tok_s\": %.1f, \"chain_GBps\": %.0f, "
           "\"logits_rel_l2_vs_first_mode\": %.3g}", err, tch, 1e6 / tch, double(wbytes) / tch / 1e3, diff);
    CK(hipGraphExecDestroy(gch));
    if (getenv("CHAIN_TRACE")) {
      ns_hip_chain_trace(ch, 1, nullptr, 0);
      for (int r = 0; r < 3; r++) NSCK(ns_hip_chain_run(ch, g_st));
      CK(hipStreamSynchronize(g_st));
      const int nshow = 10;
      std::vector<double> tr(size_t(nshow) * 8);
      const int got = ns_hip_chain_trace(ch, 0, tr.data(), nshow);
      for (int i = 0; i < got; i++) {
        fprintf(stderr, "op %2d:", i);
        for (int k = 0; k < 7; k++) fprintf(stderr, " %8.2f", tr[size_t(i) * 8 + k]);
        fprintf(stderr, "\n");
      }
    }
    ns_hip_chain_free(ch);
  }
#else
  (void)run_chain;
#endif
  printf("\n]}\n");
  return 0;
}
