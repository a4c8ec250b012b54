// engine_bench — the decode engine (csrc/ns_engine.hip: the Llama-2-7B batch-1 GEMV chain as ONE persistent launch)
// against the launch-per-operator chain (gemv_kernel, hipGraph): bit-for-bit equality of every operator of layer 0 and
// of the whole chain's logits (the launches run with 8 waves per tile, the engine's summation order), then timing.
// Build: scripts/ubench/build_chain_bench.sh (builds both)
#include <hip/hip_runtime.h>

#include <algorithm>
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
  int L = 32, d = 4096, ff = 11008, V = 32000, group = 32, reps = 20;
  bool timing = true;
  for (int i = 1; i < argc; i++) {
    if (!strcmp(argv[i], "--layers")) L = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--reps")) reps = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--no-timing")) timing = false;
  }
  CK(hipSetDevice(0));
  CK(hipStreamCreate(&g_st));
  std::vector<Layer> layers(L);
  for (int il = 0; il < L; il++) {
    const uint32_t s = 1000 + il * 8;
    const float sd = 1.0f / sqrtf(float(d)) * 1.7f, sf = 1.0f / sqrtf(float(ff)) * 1.7f / 0.6f;
    layers[il] = {make_weight(d, d, s + 0, sd, group), make_weight(d, d, s + 1, sd, group), make_weight(d, d, s + 2, sd, group),
                  make_weight(d, d, s + 3, sd, group), make_weight(ff, d, s + 4, sd, group), make_weight(ff, d, s + 5, sd, group),
                  make_weight(d, ff, s + 6, sf, group)};
  }
  ns_weight* head = make_weight(V, d, 999, 1.0f / sqrtf(float(d)) * 1.7f, group);
  uint64_t wbytes = ns_hip_weight_stream_bytes(head);
  for (auto& l : layers)
    for (ns_weight* w : {l.q, l.k, l.v, l.o, l.w1, l.w3, l.w2}) wbytes += ns_hip_weight_stream_bytes(w);

  float *x0, *x, *qkv, *attn, *t2, *logits;
  _Float16 *x0h, *xh, *qkvh, *attnh, *t2h;
  CK(hipMalloc((void**)&x0, d * 4)); CK(hipMalloc((void**)&x, d * 4)); CK(hipMalloc((void**)&qkv, 3 * d * 4));
  CK(hipMalloc((void**)&attn, d * 4)); CK(hipMalloc((void**)&t2, ff * 4)); CK(hipMalloc((void**)&logits, V * 4));
  CK(hipMalloc((void**)&x0h, d * 2)); CK(hipMalloc((void**)&xh, d * 2)); CK(hipMalloc((void**)&qkvh, 3 * d * 2));
  CK(hipMalloc((void**)&attnh, d * 2)); CK(hipMalloc((void**)&t2h, ff * 2));
  fill_kernel<<<dim3((d + 255) / 256), dim3(256), 0, g_st>>>(x0, d, 7, 1.7f);
  to_half_kernel<<<dim3((d + 255) / 256), dim3(256), 0, g_st>>>(x0, x0h, d);
  CK(hipStreamSynchronize(g_st));

  auto op_qkv = [&](const Layer& l, const float* in, _Float16* inh) {
    NSCK(ns_hip_fusion_qkv_forward_h(in, inh, l.q, l.k, l.v, qkv, qkvh, 1, d, d, g_st));
  };
  auto op_wo = [&](const Layer& l) { NSCK(ns_hip_f32f32_forward_h(qkv, qkvh, l.o, attn, attnh, 1, d, d, NS_EPI_NONE, nullptr, 0, g_st)); };
  int ref_nw = 0;  // 0: the launches' own choice; 16: the engine's summation order (fused gate/up: 8)
  auto op_gu = [&](const Layer& l) {
    if (ref_nw) ns_hip_set_tuning("gv_nw", 8);
    NSCK(ns_hip_fusion_ffn3_gateup_h(attn, attnh, l.w1, l.w3, nullptr, t2, t2h, 1, NS_EPI_SILU, g_st));
    if (ref_nw) ns_hip_set_tuning("gv_nw", ref_nw);
  };
  auto op_dn = [&](const Layer& l) { NSCK(ns_hip_f32f32_forward_h(t2, t2h, l.w2, x, xh, 1, ff, d, NS_EPI_NONE, nullptr, 0, g_st)); };
  auto op_head = [&](const float* in, _Float16* inh) {
    NSCK(ns_hip_f32f32_forward_h(in, inh, head, logits, nullptr, 1, d, V, NS_EPI_NONE, nullptr, 0, g_st));
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
  auto grab = [&](const float* p, size_t n) {
    std::vector<float> h(n);
    CK(hipMemcpy(h.data(), p, n * 4, hipMemcpyDeviceToHost));
    return h;
  };

  // ---- reference: the launches with the engine's summation order (8 waves per tile) ----
  ref_nw = 8;
  ns_hip_set_tuning("gv_nw", 8);
  std::vector<std::vector<float>> ref;
  {
    const Layer& l = layers[0];
    op_qkv(l, x0, x0h); CK(hipStreamSynchronize(g_st)); ref.push_back(grab(qkv, 3 * d));
    op_wo(l); CK(hipStreamSynchronize(g_st)); ref.push_back(grab(attn, d));
    op_gu(l); CK(hipStreamSynchronize(g_st)); ref.push_back(grab(t2, ff));
    op_dn(l); CK(hipStreamSynchronize(g_st)); ref.push_back(grab(x, d));
    chain(); CK(hipStreamSynchronize(g_st)); ref.push_back(grab(logits, V));
  }

  // ---- the engine: the same operators, one launch ----
  float *eq, *eattn, *et2, *ex, *elog;  // layer 0's outputs + the logits
  CK(hipMalloc((void**)&eq, 3 * d * 4)); CK(hipMalloc((void**)&eattn, d * 4)); CK(hipMalloc((void**)&et2, ff * 4));
  CK(hipMalloc((void**)&ex, d * 4)); CK(hipMalloc((void**)&elog, V * 4));
  std::vector<ns_engine_op> ops;
  int prev = -1;
  for (int il = 0; il < L; il++) {
    const Layer& l = layers[il];
    const bool first = il == 0;
    const int iq = int(ops.size());
    ops.push_back(ns_engine_op{l.q, nullptr, prev, first ? eq : nullptr, NS_EPI_NONE});
    ops.push_back(ns_engine_op{l.k, nullptr, -2, first ? eq + d : nullptr, NS_EPI_NONE});
    ops.push_back(ns_engine_op{l.v, nullptr, -2, first ? eq + 2 * d : nullptr, NS_EPI_NONE});
    const int io = int(ops.size());
    ops.push_back(ns_engine_op{l.o, nullptr, iq, first ? eattn : nullptr, NS_EPI_NONE});
    const int ig = int(ops.size());
    ops.push_back(ns_engine_op{l.w1, l.w3, io, first ? et2 : nullptr, NS_EPI_SILU});
    const int id = int(ops.size());
    ops.push_back(ns_engine_op{l.w2, nullptr, ig, first ? ex : nullptr, NS_EPI_NONE});
    prev = id;
  }
  ops.push_back(ns_engine_op{head, nullptr, prev, elog, NS_EPI_NONE});
  ns_engine* eng = ns_hip_engine_create(ops.data(), int(ops.size()), x0h);
  if (!eng) {
    fprintf(stderr, "engine create failed: %s\n", ns_hip_last_error());
    return 5;
  }
  for (float* p : {eq, eattn, et2, ex, elog}) CK(hipMemsetAsync(p, 0xff, 16, g_st));  // poison
  CK(hipMemsetAsync(elog, 0xff, size_t(V) * 4, g_st));
  NSCK(ns_hip_engine_launch(eng, g_st));
  CK(hipStreamSynchronize(g_st));
  unsigned status = ns_hip_engine_status(eng);
  std::vector<std::vector<float>> got = {grab(eq, 3 * d), grab(eattn, d), grab(et2, ff), grab(ex, d), grab(elog, V)};
  auto rel_l2 = [](const std::vector<float>& a, const std::vector<float>& b) {
    double num = 0, den = 0;
    for (size_t i = 0; i < a.size(); i++) {
      num += double(a[i] - b[i]) * double(a[i] - b[i]);
      den += double(b[i]) * double(b[i]);
    }
    return den > 0 ? sqrt(num / den) : sqrt(num);
  };
  const char* names[5] = {"qkv", "wo", "gateup", "down", "logits"};
  printf("{\"layers\": %d, \"weights_bytes\": %llu, \"engine_status\": %u, \"equal\": {", L, (unsigned long long)wbytes, status);
  bool all_equal = status == 0;
  for (int i = 0; i < 5; i++) {
    const bool same = memcmp(got[i].data(), ref[i].data(), got[i].size() * 4) == 0;
    all_equal = all_equal && same;
    printf("%s\"%s\": [%s, %.3g]", i ? ", " : "", names[i], same ? "true" : "false", rel_l2(got[i], ref[i]));
    if (!same) {
      size_t nan_g = 0, nan_r = 0, ndiff = 0, first = got[i].size();
      for (size_t j = 0; j < got[i].size(); j++) {
        nan_g += std::isnan(got[i][j]);
        nan_r += std::isnan(ref[i][j]);
        if (memcmp(&got[i][j], &ref[i][j], 4)) {
          ndiff++;
          if (first == got[i].size()) first = j;
        }
      }
      fprintf(stderr, "%s: %zu of %zu differ (first at %zu), NaN got %zu ref %zu; got[0..3] %g %g %g %g ref %g %g %g %g; at first: got %g ref %g\n",
              names[i], ndiff, got[i].size(), first, nan_g, nan_r, got[i][0], got[i][1], got[i][2], got[i][3], ref[i][0], ref[i][1],
              ref[i][2], ref[i][3], got[i][first < got[i].size() ? first : 0], ref[i][first < ref[i].size() ? first : 0]);
    }
  }
  printf("}");
  // a second and third launch (epoch advance, stale granules of the previous token) must give the same logits
  for (int r = 0; r < 2; r++) {
    CK(hipMemsetAsync(elog, 0xff, size_t(V) * 4, g_st));
    NSCK(ns_hip_engine_launch(eng, g_st));
    CK(hipStreamSynchronize(g_st));
    const auto again = grab(elog, V);
    const bool same = memcmp(again.data(), ref[4].data(), size_t(V) * 4) == 0;
    all_equal = all_equal && same;
    printf(", \"relaunch%d_equal\": %s", r, same ? "true" : "false");
  }
  status = ns_hip_engine_status(eng);
  printf(", \"status_after\": %u, \"all_equal\": %s", status, all_equal ? "true" : "false");
  fflush(stdout);
  if (timing && status == 0) {
    hipGraphExec_t gc = capture(chain);
    const double tc8 = time_graph(gc, reps);
    CK(hipGraphExecDestroy(gc));
    ref_nw = 0;
    ns_hip_set_tuning("gv_nw", 0);
    gc = capture(chain);
    const double tc = time_graph(gc, reps);
    CK(hipGraphExecDestroy(gc));
    hipGraphExec_t ge = capture([&] { NSCK(ns_hip_engine_launch(eng, g_st)); });
    const double te = time_graph(ge, reps);
    CK(hipGraphExecDestroy(ge));
    // direct launches, back to back
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, g_st));
    for (int i = 0; i < reps; i++) NSCK(ns_hip_engine_launch(eng, g_st));
    CK(hipEventRecord(e1, g_st));
    CK(hipStreamSynchronize(g_st));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    status = ns_hip_engine_status(eng);
    printf(", \"launches_us\": %.1f, \"launches_nw8_us\": %.1f, \"engine_graph_us\": %.1f, \"engine_direct_us\": %.1f, "
           "\"engine_tok_s\": %.1f, \"launches_tok_s\": %.1f, \"engine_GBps\": %.0f, \"ratio\": %.3f, \"status_timed\": %u",
           tc, tc8, te, double(ms) * 1e3 / reps, 1e6 / te, 1e6 / tc, double(wbytes) / te / 1e3, te / tc, status);
  }
  if (getenv("ENG_TRACE")) {  // one more launch, then the stamps of a few workgroups relative to the launch's first stamp
    NSCK(ns_hip_engine_launch(eng, g_st));
    CK(hipStreamSynchronize(g_st));
    const int nwg = 256;
    std::vector<unsigned long long> tr(size_t(nwg) * 64 * 8);
    if (ns_hip_engine_trace(eng, tr.data(), nwg) == 0) {
      unsigned long long t0 = ~0ull;
      for (int g = 0; g < nwg; g++)
        for (int k = 0; k < 8; k++) {
          const unsigned long long v = tr[(size_t(g) * 64 + 0) * 8 + k];
          if (v && v < t0 && (k < 4 || k >= 6)) t0 = v;
        }
      const char* nm[8] = {"ld0", "ld1", "in0", "in1", "clk_wait", "clk_math", "c_done", "pub"};
      fprintf(stderr, "stamps in us since the first stamp of the launch; columns:");
      for (int k = 0; k < 8; k++) fprintf(stderr, " %s", nm[k]);
      fprintf(stderr, "\n");
      for (int g : {0, 1, 77, 128, 255})
        for (int op = 0; op < 20; op++) {
          fprintf(stderr, "wg %3d op %2d:", g, op);
          for (int k = 0; k < 8; k++) {
            const unsigned long long v = tr[(size_t(g) * 64 + op) * 8 + k];
            if (k >= 4 && k <= 5) fprintf(stderr, " %8llu", v);  // durations in shader clocks (wait / LDS + arithmetic / requests)
            else if (v) fprintf(stderr, " %8.2f", double(v - t0) / 100.0);
            else fprintf(stderr, " %8s", "-");
          }
          fprintf(stderr, "\n");
        }
      // per op: latest publish over all workgroups, latest input-ready over all workgroups
      for (int op = 0; op < 20; op++) {
        unsigned long long pmax = 0, rmax = 0, pmin = ~0ull, dmax = 0;
        for (int g = 0; g < nwg; g++) {
          const unsigned long long* r = &tr[(size_t(g) * 64 + op) * 8];
          if (r[7]) pmax = std::max(pmax, r[7]), pmin = std::min(pmin, r[7]);
          rmax = std::max(rmax, r[3]);
          dmax = std::max(dmax, r[6]);
        }
        fprintf(stderr, "op %2d: first publish %8.2f last publish %8.2f | last input staged %8.2f | last wave-0 done %8.2f\n", op,
                pmin == ~0ull ? 0.0 : double(pmin - t0) / 100.0, pmax ? double(pmax - t0) / 100.0 : 0.0, rmax ? double(rmax - t0) / 100.0 : 0.0,
                dmax ? double(dmax - t0) / 100.0 : 0.0);
      }
    }
  }
  printf("}\n");
  ns_hip_engine_destroy(eng);
  return all_equal ? 0 : 1;
}
