/*
 * llama_ref_harness.cpp — TEST INFRASTRUCTURE.  Flat C entry points over the reference's UNCHANGED model code:
 *   /root/reference/neural_speed/models/llama/llama.cpp        (the graph builder, model_eval_internal :83-790)
 *   /root/reference/neural_speed/models/llama/llama_utils.cpp  (Llama::init / ::load, the llama quant-layer rules)
 *   /root/reference/neural_speed/models/model_utils/model_utils.cpp, model_files.h (context, kv-cache allocation, the NE
 *       file reader incl. BTLA tensors :1177-1235, :1564-1571), quant_utils.cpp (model_quantize -> bestla_quantize
 *       :269-354 -> BTLAGemmQuantPackB), application/common.cpp (quant_params helpers)
 *   /root/reference/neural_speed/core/ne_layers.c              (the graph executor)
 * (and, built a second time with -DNS_FAMILY_NAME / _ARCH, models/gptj/gptj.cpp + gptj_utils.cpp -> libne_gptj_ref.so)
 * all compiled from where they lie into oracle/_ref/libne_llama_ref.so (oracle/Makefile target nellama) with
 * glue/shim in front of the include path (two shim headers replace the xbyak-dependent bestla_common.hpp /
 * bestla_parallel.h) and the product's glue files (glue/ne_bestla_hip_glue.c, glue/bestla_gemm_hip.cpp) in place of
 * core/layers/*.cpp.  Every bestla_* / ns_BTLAGemm* symbol binds at run time to whichever provider was loaded
 * RTLD_GLOBAL first: libns_hip.so (the product, GPU) or tests/tools/oracle_bestla_provider.c (the CPU oracle).
 *
 * This is SURVEY section 8 (b)'s claim made executable: "existing model graphs ... keep working unchanged".
 */
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "application/common.h"
#include "models/model_utils/model_config.h"
#include "models/model_utils/model_types.h"
#include "models/model_utils/model_utils.h"
#include "models/model_utils/quant_utils.h"

#ifndef NS_FAMILY_NAME  /* oracle/Makefile builds this file once per model family: llama (default), gptj */
#define NS_FAMILY_NAME "llama"
#define NS_FAMILY_ARCH MODEL_LLAMA
#endif

extern "C" {

/* what application/quant_model.cpp:37-72 does: f32 NE file -> BTLA-quantized NE file through the reference's quantizer
 * driver.  Returns 0 on success. */
int nellama_quantize(const char* in_path, const char* out_path, const char* weight_dtype, const char* alg, int group_size,
                     const char* scale_dtype, const char* compute_dtype) {
  model_init_backend();
  quant_params q;
  q.model_file = in_path;
  q.out_file = out_path;
  q.weight_dtype = weight_dtype;
  q.alg = alg;
  q.group_size = group_size;
  q.scale_dtype = scale_dtype;
  q.compute_dtype = compute_dtype;
  q.model_name = NS_FAMILY_NAME;
  q.model_arch = model_name_to_arch::init().find(q.model_name);
  q.nthread = 1;
  auto ql = ql_registry::create_ql(q.model_name);
  return model_quantize(q, ql);
}

/* Greedy generation the way application/main_run.cpp / main_pybind.cpp drive the model: one model_eval over the prompt,
 * then one per new token with n_past advanced.  kv_type: 0 auto (the library-managed cache when
 * bestla_reordered_attn_fp32_support says yes), 1 fp16, 2 fp32 (KV_MEM_TYPE, model_types.h:96-100).  out_tokens[n_new]; out_logits
 * [(n_new) x n_vocab] = the logits each token was picked from (may be NULL).  Returns the number of tokens generated, < 0
 * on failure. */
static double g_last_us_per_token = 0.0;
/* wall time per single-token eval of the last nellama_generate call (evals after the second one) */
double nellama_last_us_per_token(void) { return g_last_us_per_token; }

int nellama_generate(const char* model_path, const int* prompt, int n_prompt, int n_new, int n_ctx, int kv_type,
                     int* out_tokens, float* out_logits) {
  model_init_backend();
  double us_sum = 0;
  int us_n = 0;
  model_context_params p = model_context_default_params();
  p.arch = NS_FAMILY_ARCH;
  p.n_ctx = n_ctx;
  p.seed = 1;
  p.kv_type = static_cast<KV_MEM_TYPE>(kv_type);
  p.use_mmap = false;
  p.batch_size = 1;
  p.max_request_num = 1;
  p.beam_size = 1;
  p.beam_search = false;
  p.cont_batching = false;
  p.scratch_size_ratio = 0.125f; /* llama_mem_req sizes its scratch for 7B+ models (llama.h:30-80) */
  model_context* ctx = model_init_from_file(model_path, p);
  if (!ctx) return -1;
  const int n_vocab = model_n_vocab(ctx);
  std::vector<model_token> toks(prompt, prompt + n_prompt);
  int n_past = 0, n_total = 0, made = 0;
  std::vector<model_token> cur = toks;
  for (int step = 0; step < n_new; step++) {
    model_input in;
    in.tokens = cur.data();
    in.n_tokens = static_cast<uint32_t>(cur.size());
    in.n_prompt_tokens = static_cast<uint32_t>(n_prompt);
    in.n_past = static_cast<uint32_t>(n_past);
    in.n_total = static_cast<uint32_t>(n_total);
    in.request_idx = 0;
    in.beam_idx = 0;
    fprintf(stderr, "nellama_generate: step %d, %zu token(s) at n_past %d\n", step, cur.size(), n_past);
    const int64_t t0 = ne_time_us();
    if (model_eval(ctx, &in, 1, 1) != 0) {
      model_free(ctx);
      return -2;
    }
    if (cur.size() == 1 && step > 1) us_sum += double(ne_time_us() - t0), us_n++;
    n_past += static_cast<int>(cur.size());
    n_total += static_cast<int>(cur.size());
    const float* logits = model_get_logits(ctx);
    int best = 0;
    for (int i = 1; i < n_vocab; i++)
      if (logits[i] > logits[best]) best = i;
    if (out_logits) memcpy(out_logits + static_cast<size_t>(step) * n_vocab, logits, sizeof(float) * n_vocab);
    out_tokens[made++] = best;
    cur.assign(1, best);
  }
  g_last_us_per_token = us_n ? us_sum / us_n : 0.0;
  model_free(ctx);
  return made;
}

#ifdef NS_SYCL
/* The same loop with the model OFFLOADED: the reference's own device switch (-DNS_SYCL) — model_init_sycl ->
 * bestla_create_device, every layer's tensors on the device (n_gpu_layers = all: Llama::load, llama_utils.cpp:95-200;
 * BTLA weights through bestla_device_load_storage, model_files.h:1515-1527), the fp32 device kv cache
 * (model_utils.cpp:140-160) and the device branch of the graph builder (llama.cpp:190-330: ne_device_sync, device RoPE,
 * ne_cpy into the cache, ne_flash_attn -> bestla_device_mha_f32).  libns_hip.so + glue/ne_bestla_hip_device.c answer the
 * bestla_device_* calls.  out_us_per_token: wall time of the single-token evals (may be NULL). */
/* wall time of every single-token eval of the last nellama_generate_dev call, in order (the mean handed back by the call includes the
 * eval at whose end the device route captures its replay plan: scripts read the median / the steady tail from here) */
static std::vector<double> g_eval_us;
static double g_prompt_us = 0.0; /* wall time of the prompt's eval (all its tokens in one model_eval) of the last nellama_generate_dev call */
double nellama_prompt_us(void) { return g_prompt_us; }
/* NS_HARNESS_PROMPT_REPEAT=1: the prompt is evaluated a SECOND time in the same context (n_past 0 again: the same positions are rewritten) before the
 * generation goes on from it — its wall time, without the first-use costs the first evaluation of a process carries (untouched host pages behind the
 * logits tensor, device scratch allocations, cold clocks); 0 when not asked for */
static double g_prompt_warm_us = 0.0;
double nellama_prompt_warm_us(void) { return g_prompt_warm_us; }
int nellama_eval_times(double* out, int cap) {
  const int n = static_cast<int>(g_eval_us.size()) < cap ? static_cast<int>(g_eval_us.size()) : cap;
  for (int i = 0; i < n; i++) out[i] = g_eval_us[i];
  return static_cast<int>(g_eval_us.size());
}
int nellama_generate_dev(const char* model_path, const int* prompt, int n_prompt, int n_new, int n_ctx, int n_gpu_layers,
                         int* out_tokens, float* out_logits, double* out_us_per_token) {
  g_eval_us.clear();
  g_prompt_warm_us = 0.0;
  model_init_backend();
  ne_sycl_context* dev = model_init_sycl(false);
  if (!dev) return -3;
  model_context_params p = model_context_default_params();
  p.arch = NS_FAMILY_ARCH;
  p.n_ctx = n_ctx;
  p.seed = 1;
  p.kv_type = KV_MEM_TYPE_F32; /* model_init_from_gpt_params does the same with a device context (model_utils.cpp:1420-1422) */
  p.use_mmap = false;
  p.batch_size = 1;
  p.max_request_num = 1;
  p.beam_size = 1;
  p.beam_search = false;
  p.cont_batching = false;
  p.scratch_size_ratio = 0.125f;
  p.n_gpu_layers = n_gpu_layers; /* = the model's layer count: every layer (the loader sizes the device pool by it) */
  p.dev_ctx = dev;
  model_context* ctx = model_init_from_file(model_path, p);
  if (!ctx) return -1;
  const int n_vocab = model_n_vocab(ctx);
  int n_past = 0, made = 0;
  std::vector<model_token> cur(prompt, prompt + n_prompt);
  double us = 0;
  int timed = 0;
  for (int step = 0; step < n_new; step++) {
    model_input in;
    in.tokens = cur.data();
    in.n_tokens = static_cast<uint32_t>(cur.size());
    in.n_prompt_tokens = static_cast<uint32_t>(n_prompt);
    in.n_past = static_cast<uint32_t>(n_past);
    in.n_total = static_cast<uint32_t>(n_past);
    in.request_idx = 0;
    in.beam_idx = 0;
    const int64_t t0 = ne_time_us();
    if (model_eval(ctx, &in, 1, 1) != 0) {
      model_free(ctx);
      return -2;
    }
    if (cur.size() == 1 && step > 1) us += double(ne_time_us() - t0), timed++;
    if (cur.size() == 1) g_eval_us.push_back(double(ne_time_us() - t0));
    else if (step == 0) g_prompt_us = double(ne_time_us() - t0);
    if (step == 0 && cur.size() > 1 && getenv("NS_HARNESS_PROMPT_REPEAT") && atoi(getenv("NS_HARNESS_PROMPT_REPEAT")) != 0) {
      const int64_t t1 = ne_time_us();
      if (model_eval(ctx, &in, 1, 1) != 0) {
        model_free(ctx);
        return -2;
      }
      g_prompt_warm_us = double(ne_time_us() - t1);
    }
    n_past += static_cast<int>(cur.size());
    const float* logits = model_get_logits(ctx);
    int best = 0;
    for (int i = 1; i < n_vocab; i++)
      if (logits[i] > logits[best]) best = i;
    if (out_logits) memcpy(out_logits + static_cast<size_t>(step) * n_vocab, logits, sizeof(float) * n_vocab);
    out_tokens[made++] = best;
    cur.assign(1, best);
  }
  if (out_us_per_token) *out_us_per_token = timed ? us / timed : 0.0;
  model_free(ctx);
  model_release_sycl(dev);
  return made;
}
/* A conversation on the device route: turn t evaluates its prompt chunk (n_tok[t] tokens: one model_eval over all of them) at the position the
 * previous turn ended on — or, when rewind[t] != 0, at position 0 of the SAME context (the cache is written over: a new sequence) — and then
 * generates n_new[t] tokens greedily.  What the route's layers see: a plan held over single-token evals meets a multi-token eval (it falls back through the window),
 * the fp16 kv mirror converts a chunk in the middle of a cache / starts over, two agreeing tokens make the next plan.  out_tokens / out_logits: every
 * generated token of every turn in order ([sum n_new] and [sum n_new][n_vocab]).  Returns the number of generated tokens, < 0 on failure. */
int nellama_generate_dev_turns(const char* model_path, int n_turns, const int* chunks, const int* n_tok, const int* n_new, const int* rewind, int n_ctx,
                               int n_gpu_layers, int* out_tokens, float* out_logits) {
  model_init_backend();
  ne_sycl_context* dev = model_init_sycl(false);
  if (!dev) return -3;
  model_context_params p = model_context_default_params();
  p.arch = NS_FAMILY_ARCH;
  p.n_ctx = n_ctx;
  p.seed = 1;
  p.kv_type = KV_MEM_TYPE_F32;
  p.use_mmap = false;
  p.batch_size = 1;
  p.max_request_num = 1;
  p.beam_size = 1;
  p.beam_search = false;
  p.cont_batching = false;
  p.scratch_size_ratio = 0.125f;
  p.n_gpu_layers = n_gpu_layers;
  p.dev_ctx = dev;
  model_context* ctx = model_init_from_file(model_path, p);
  if (!ctx) return -1;
  const int n_vocab = model_n_vocab(ctx);
  int n_past = 0, made = 0, off = 0;
  for (int t = 0; t < n_turns; t++) {
    if (rewind[t]) n_past = 0;
    std::vector<model_token> cur(chunks + off, chunks + off + n_tok[t]);
    off += n_tok[t];
    const int n_prompt = n_tok[t];
    for (int step = 0; step < n_new[t]; step++) {
      model_input in;
      in.tokens = cur.data();
      in.n_tokens = static_cast<uint32_t>(cur.size());
      in.n_prompt_tokens = static_cast<uint32_t>(n_prompt);
      in.n_past = static_cast<uint32_t>(n_past);
      in.n_total = static_cast<uint32_t>(n_past);
      in.request_idx = 0;
      in.beam_idx = 0;
      if (model_eval(ctx, &in, 1, 1) != 0) {
        model_free(ctx);
        return -2;
      }
      n_past += static_cast<int>(cur.size());
      const float* logits = model_get_logits(ctx);
      int best = 0;
      for (int i = 1; i < n_vocab; i++)
        if (logits[i] > logits[best]) best = i;
      if (out_logits) memcpy(out_logits + static_cast<size_t>(made) * n_vocab, logits, sizeof(float) * n_vocab);
      out_tokens[made++] = best;
      cur.assign(1, best);
    }
    /* (the last generated token of a turn is not evaluated: the next chunk follows the cache as it stands, as a chat front end that cuts a reply does) */
  }
  model_free(ctx);
  model_release_sycl(dev);
  return made;
}
#endif

/* Continuous batching the way the reference's serving loop evaluates it (models/llama/llama.cpp:66-70, :330-350, :496-571:
 * batch_size == n_input requests concatenated into ONE graph without padding, per-request RoPE offsets, kv-cache blocks per
 * request, attention per group of requests that share (n_tokens, n_past)): two requests, greedy, every eval carries both —
 * the two prompts first, then one token each at its own n_past.  out_tokens [2][n_new], out_logits [2][n_new][n_vocab] (or
 * NULL).  Returns n_new, < 0 on failure. */
int nellama_generate2(const char* model_path, const int* prompt0, int n0, const int* prompt1, int n1, int n_new, int n_ctx,
                      int kv_type, int* out_tokens, float* out_logits) {
  model_init_backend();
  model_context_params p = model_context_default_params();
  p.arch = NS_FAMILY_ARCH;
  p.n_ctx = n_ctx;
  p.seed = 1;
  p.kv_type = static_cast<KV_MEM_TYPE>(kv_type);
  p.use_mmap = false;
  p.batch_size = 2;
  p.max_request_num = 2;
  p.beam_size = 1;
  p.beam_search = false;
  p.cont_batching = true;
  p.scratch_size_ratio = 0.125f;
  model_context* ctx = model_init_from_file(model_path, p);
  if (!ctx) return -1;
  const int n_vocab = model_n_vocab(ctx);
  std::vector<model_token> cur[2] = {std::vector<model_token>(prompt0, prompt0 + n0), std::vector<model_token>(prompt1, prompt1 + n1)};
  const int n_prompt[2] = {n0, n1};
  int n_past[2] = {0, 0};
  for (int step = 0; step < n_new; step++) {
    model_input in[2];
    for (int r = 0; r < 2; r++) {
      in[r].tokens = cur[r].data();
      in[r].n_tokens = static_cast<uint32_t>(cur[r].size());
      in[r].n_prompt_tokens = static_cast<uint32_t>(n_prompt[r]);
      in[r].n_past = static_cast<uint32_t>(n_past[r]);
      in[r].n_total = static_cast<uint32_t>(n_past[r]);
      in[r].request_idx = r;
      in[r].beam_idx = 0;
    }
    if (model_eval(ctx, in, 2, 1) != 0) {
      model_free(ctx);
      return -2;
    }
    const float* logits = model_get_logits(ctx);  // [2][n_vocab]: the last position of every request
    for (int r = 0; r < 2; r++) {
      n_past[r] += static_cast<int>(cur[r].size());
      const float* lr = logits + static_cast<size_t>(r) * n_vocab;
      int best = 0;
      for (int i = 1; i < n_vocab; i++)
        if (lr[i] > lr[best]) best = i;
      if (out_logits) memcpy(out_logits + (static_cast<size_t>(r) * n_new + step) * n_vocab, lr, sizeof(float) * n_vocab);
      out_tokens[r * n_new + step] = best;
      cur[r].assign(1, best);
    }
  }
  model_free(ctx);
  return n_new;
}

}  // extern "C"
