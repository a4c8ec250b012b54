/*
 * ne_ref_harness.c — TEST INFRASTRUCTURE.  Drives the reference's own graph executor
 * (/root/reference/neural_speed/core/ne_layers.c, compiled from where it lies into oracle/_ref/libne_ref.so by
 * oracle/Makefile) through a few flat C entry points, so that
 *   (1) the oracle's restatements of graph operators (RoPE in all its modes) are pinned against the real
 *       ne_compute_forward_* code, and
 *   (2) the reference's `ne_mul_mat` / fused QKV / fused FFN nodes over BTLA weight tensors can be executed with the
 *       product library (libns_hip.so) answering the `bestla_*` calls — the drop-in boundary exercised from the
 *       reference's side, unchanged graph code included (ne_compute_forward_mul_mat_q_f32_bestla, ne_layers.c:7219-7316).
 *
 * The three graph-struct entry points (bestla_parallel_for / bestla_support / bestla_backend_support) are the glue a
 * maintainer adds on the ggml side: glue/ne_bestla_hip_glue.c, written against the
 * reference's headers and linked into this library.
 * Every other `bestla_*` symbol has an aborting fallback in ne_ref_stubs.c, which libns_hip.so interposes when it is
 * loaded with RTLD_GLOBAL before this library.
 */
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>

#include "ne.h"
#include "ne_bestla.h"
#include "ne_layers.h"
#include "layers/mha_dense.h"

/* the three graph-struct entry points live in glue/ne_bestla_hip_glue.c (product glue, compiled into this library by
 * oracle/Makefile) */

/* non-static builder of every RoPE flavour (ne_layers.c:3377-3434) */
struct ne_tensor* ne_rope_impl(struct ne_context* ctx, struct ne_tensor* a, int n_past, int n_dims, int mode, int prompt_size,
                               bool inplace, int n_keep, struct ne_tensor* cossin, int* n_padding, bool padding_left,
                               float freq_base, float freq_scale, int yarn_orig_ctx, float ext_factor, float attn_factor,
                               float beta_fast, float beta_slow, struct ne_tensor* factor, float scale_factor);

static struct ne_cgraph g_graph; /* ~1 MB: not on the stack */

static void run_graph(struct ne_context* ctx, struct ne_tensor* out) {
  g_graph = ne_build_forward(out);
  g_graph.n_threads = 1;
  ne_graph_compute(ctx, &g_graph);
}

/* x, y: fp32 [batch][seq][heads][head_size].  op_freq_scale is the value the graph builder receives (the forward uses
 * its reciprocal, ne_layers.c:9262).  factors: long-rope divisor table [n_dims / 2] or NULL.  n_padding: [batch] or NULL. */
int neref_rope(const float* x, float* y, int batch, int seq, int heads, int head_size, int n_past, int n_dims, int mode,
               int prompt_size, float freq_base, float op_freq_scale, int yarn_orig_ctx, float ext_factor, float attn_factor,
               float beta_fast, float beta_slow, const int* n_padding, const float* factors, float scale_factor) {
  const size_t n = (size_t)batch * seq * heads * head_size;
  struct ne_init_params ip = {n * 4 + (16u << 20), NULL, false};
  struct ne_context* ctx = ne_init(ip);
  if (!ctx) return -1;
  struct ne_tensor* a = ne_new_tensor_4d(ctx, NE_TYPE_F32, head_size, heads, seq, batch, NE_SIZE_CALC, NE_BACKEND_CPU);
  memcpy(a->data, x, n * 4);
  struct ne_tensor* f = NULL;
  if (factors) {
    f = ne_new_tensor_1d(ctx, NE_TYPE_F32, n_dims / 2, NE_SIZE_CALC, NE_BACKEND_CPU);
    memcpy(f->data, factors, (size_t)(n_dims / 2) * 4);
  }
  int pads[64];
  for (int i = 0; i < batch && i < 64; i++) pads[i] = n_padding ? n_padding[i] : 0;
  struct ne_tensor* r = ne_rope_impl(ctx, a, n_past, n_dims, mode, prompt_size, true, -1, NULL, n_padding ? pads : NULL, true,
                                     freq_base, op_freq_scale, yarn_orig_ctx, ext_factor, attn_factor, beta_fast, beta_slow, f,
                                     scale_factor);
  run_graph(ctx, r);
  memcpy(y, a->data, n * 4);
  ne_free(ctx);
  return 0;
}

static struct ne_tensor* btla_tensor(struct ne_context* ctx, void* blob, size_t blob_bytes, int k, int n) {
  /* what the model loader does for a BTLA tensor (model_files.h:1564-1571): ne = {K, N}, size = blob bytes, data -> blob */
  struct ne_tensor* w = ne_new_tensor_2d(ctx, NE_TYPE_BTLA, k, n, blob_bytes, NE_BACKEND_CPU);
  w->data = blob;
  return w;
}

/* C[m][n] = A[m][k] . W through the reference graph: ne_mul_mat over a BTLA weight tensor -> NE_OP_MUL_MAT ->
 * ne_compute_forward_mul_mat_q_f32_bestla -> bestla_f32f32_forward (whoever provides it) */
int neref_mul_mat(const float* a, void* blob, size_t blob_bytes, float* c, int m, int n, int k) {
  struct ne_init_params ip = {(size_t)m * (k + n) * 4 + (64u << 20), NULL, false};
  struct ne_context* ctx = ne_init(ip);
  if (!ctx) return -1;
  /* the tensor object of a BTLA weight is created without payload: data points into the caller's blob */
  struct ne_init_params ipw = {1u << 20, NULL, true};
  struct ne_context* wctx = ne_init(ipw);
  if (!wctx) return -1;
  struct ne_tensor* w = btla_tensor(wctx, blob, blob_bytes, k, n);
  struct ne_tensor* x = ne_new_tensor_2d(ctx, NE_TYPE_F32, k, m, NE_SIZE_CALC, NE_BACKEND_CPU);
  memcpy(x->data, a, (size_t)m * k * 4);
  struct ne_tensor* y = ne_mul_mat(ctx, w, x);
  run_graph(ctx, y);
  memcpy(c, y->data, (size_t)m * n * 4);
  ne_free(ctx);
  ne_free(wctx);
  return 0;
}

/* out[m][d] = FFN_SiLU(A; W1 [ff x d], W2 [d x ff], W3 [ff x d]) through the reference's fused node
 * (ne_ffn_silu -> NE_OP_MUL_FFN_SILU -> bestla_fusion_FFN_SiLu_f32f32_forward, ne_layers.c:8037-8051) */
int neref_ffn_silu(const float* a, void* b1, size_t s1, void* b2, size_t s2, void* b3, size_t s3, float* out, int m, int d,
                   int ff) {
  struct ne_init_params ip = {(size_t)m * (d * 2 + ff * 2) * 4 + (64u << 20), NULL, false};
  struct ne_context* ctx = ne_init(ip);
  struct ne_init_params ipw = {1u << 20, NULL, true};
  struct ne_context* wctx = ne_init(ipw);
  if (!ctx || !wctx) return -1;
  struct ne_tensor* w1 = btla_tensor(wctx, b1, s1, d, ff);
  struct ne_tensor* w2 = btla_tensor(wctx, b2, s2, ff, d);
  struct ne_tensor* w3 = btla_tensor(wctx, b3, s3, d, ff);
  struct ne_tensor* x = ne_new_tensor_2d(ctx, NE_TYPE_F32, d, m, NE_SIZE_CALC, NE_BACKEND_CPU);
  memcpy(x->data, a, (size_t)m * d * 4);
  struct ne_tensor* y = ne_ffn_silu(ctx, w1, w2, w3, x);
  run_graph(ctx, y);
  memcpy(out, y->data, (size_t)m * d * 4);
  ne_free(ctx);
  ne_free(wctx);
  return 0;
}

/* out[m][n] = A[m] . W[ids[m][id]] through the reference's expert-indexed node: ne_mul_mat_id -> NE_OP_MUL_MAT_ID ->
 * ne_compute_forward_mul_mat_id_q_f32_bestla (ne_layers.c:7783-7916: token rows are grouped per expert in the INIT task,
 * then ONE bestla_f32f32_forward per (expert, token row)).  ids [m][n_ids] int32. */
int neref_mul_mat_id(const float* a, void* const* blobs, const size_t* blob_bytes, int n_as, const int32_t* ids, int n_ids,
                     int id, float* out, int m, int n, int k) {
  struct ne_init_params ip = {(size_t)m * (k + n + n_ids) * 4 + (64u << 20), NULL, false};
  struct ne_context* ctx = ne_init(ip);
  struct ne_init_params ipw = {1u << 20, NULL, true};
  struct ne_context* wctx = ne_init(ipw);
  if (!ctx || !wctx || n_as > 8) return -1;
  struct ne_tensor* as[8];
  for (int i = 0; i < n_as; i++) as[i] = btla_tensor(wctx, blobs[i], blob_bytes[i], k, n);
  struct ne_tensor* x = ne_new_tensor_2d(ctx, NE_TYPE_F32, k, m, NE_SIZE_CALC, NE_BACKEND_CPU);
  memcpy(x->data, a, (size_t)m * k * 4);
  struct ne_tensor* idt = ne_new_tensor_2d(ctx, NE_TYPE_I32, n_ids, m, NE_SIZE_CALC, NE_BACKEND_CPU);
  memcpy(idt->data, ids, (size_t)m * n_ids * 4);
  struct ne_tensor* y = ne_mul_mat_id(ctx, as, n_as, idt, id, x);
  run_graph(ctx, y);
  memcpy(out, y->data, (size_t)m * n * 4);
  ne_free(ctx);
  ne_free(wctx);
  return 0;
}

/* out[m][d] = FFN(A; gate / down / up of expert ids[0][id]) through ne_mul_id_ffn_silu / _gelu -> NE_OP_MUL_ID_FFN_* ->
 * ne_compute_forward_ffn_id_* (ne_layers.c:8053-8170): the expert of the FIRST token row serves all rows, the three
 * weights go to bestla_fusion_FFN_{SiLu,Gelu_Mul}_f32f32_forward.  blobs = gate[0..n_as) , down[..], up[..]. */
int neref_ffn_id(int gelu, const float* a, void* const* blobs, const size_t* blob_bytes, int n_as, const int32_t* ids,
                 int n_ids, int id, float* out, int m, int d, int ff) {
  struct ne_init_params ip = {(size_t)m * (2 * d + 2 * ff + n_ids) * 4 + (64u << 20), NULL, false};
  struct ne_context* ctx = ne_init(ip);
  struct ne_init_params ipw = {1u << 20, NULL, true};
  struct ne_context* wctx = ne_init(ipw);
  if (!ctx || !wctx || n_as > 8) return -1;
  struct ne_tensor *gate[8], *down[8], *up[8];
  for (int i = 0; i < n_as; i++) {
    gate[i] = btla_tensor(wctx, blobs[i], blob_bytes[i], d, ff);
    down[i] = btla_tensor(wctx, blobs[n_as + i], blob_bytes[n_as + i], ff, d);
    up[i] = btla_tensor(wctx, blobs[2 * n_as + i], blob_bytes[2 * n_as + i], d, ff);
  }
  struct ne_tensor* x = ne_new_tensor_2d(ctx, NE_TYPE_F32, d, m, NE_SIZE_CALC, NE_BACKEND_CPU);
  memcpy(x->data, a, (size_t)m * d * 4);
  struct ne_tensor* idt = ne_new_tensor_2d(ctx, NE_TYPE_I32, n_ids, m, NE_SIZE_CALC, NE_BACKEND_CPU);
  memcpy(idt->data, ids, (size_t)m * n_ids * 4);
  struct ne_tensor* y = gelu ? ne_mul_id_ffn_gelu(ctx, down, gate, up, n_as, idt, id, x)
                             : ne_mul_id_ffn_silu(ctx, down, gate, up, n_as, idt, id, x);
  run_graph(ctx, y);
  memcpy(out, y->data, (size_t)m * d * 4);
  ne_free(ctx);
  ne_free(wctx);
  return 0;
}

/* Attention as the reference's model graphs spell it when the fused kernel is not used (models/llama/llama.cpp non-fused
 * branch): KQ = mul_mat(K, Q) -> scale -> diag_mask_inf(n_past) -> soft_max -> mul_mat(V^T, P).  fp32 tensors throughout
 * (the caller passes fp16-representable K / V values), GQA through mul_mat's broadcast (head / (heads / heads_kv)).
 * q [heads][sl_q][hs], k / v [heads_kv][sl_kv][hs], out [heads][sl_q][hs]; causal != 0 masks keys j > i + (sl_kv - sl_q). */
int neref_attn_unfused(const float* q, const float* k, const float* v, float* out, int heads, int heads_kv, int hs, int sl_q,
                       int sl_kv, float scale, int causal) {
  const size_t nq = (size_t)heads * sl_q * hs, nk = (size_t)heads_kv * sl_kv * hs, np = (size_t)heads * sl_q * sl_kv;
  struct ne_init_params ip = {(nq * 2 + nk * 2 + np * 4) * 4 + (64u << 20), NULL, false};
  struct ne_context* ctx = ne_init(ip);
  if (!ctx) return -1;
  struct ne_tensor* Q = ne_new_tensor_3d(ctx, NE_TYPE_F32, hs, sl_q, heads, NE_SIZE_CALC, NE_BACKEND_CPU);
  struct ne_tensor* K = ne_new_tensor_3d(ctx, NE_TYPE_F32, hs, sl_kv, heads_kv, NE_SIZE_CALC, NE_BACKEND_CPU);
  struct ne_tensor* Vt = ne_new_tensor_3d(ctx, NE_TYPE_F32, sl_kv, hs, heads_kv, NE_SIZE_CALC, NE_BACKEND_CPU);
  memcpy(Q->data, q, nq * 4);
  memcpy(K->data, k, nk * 4);
  for (int h = 0; h < heads_kv; h++)
    for (int j = 0; j < sl_kv; j++)
      for (int e = 0; e < hs; e++)
        ((float*)Vt->data)[((size_t)h * hs + e) * sl_kv + j] = v[((size_t)h * sl_kv + j) * hs + e];
  struct ne_tensor* KQ = ne_mul_mat(ctx, K, Q);
  struct ne_tensor* KQs = ne_scale(ctx, KQ, ne_new_f32(ctx, scale));
  struct ne_tensor* KQm = causal ? ne_diag_mask_inf(ctx, KQs, sl_kv - sl_q) : KQs;
  struct ne_tensor* P = ne_soft_max(ctx, KQm);
  struct ne_tensor* O = ne_mul_mat(ctx, Vt, P);
  run_graph(ctx, O);
  memcpy(out, O->data, nq * 4);
  ne_free(ctx);
  return 0;
}

/* out[3][m][n] = {A Wq | A Wk | A Wv} through the reference's fused node (ne_mul_qkv -> NE_OP_MUL_QKV ->
 * bestla_fusion_QKV_f32f32_forward, ne_layers.c:8037-8051) */
int neref_mul_qkv(const float* a, void* bq, size_t sq, void* bk, size_t sk, void* bv, size_t sv, float* out, int m, int n,
                  int k) {
  struct ne_init_params ip = {(size_t)m * (k + 3 * n) * 4 + (64u << 20), NULL, false};
  struct ne_context* ctx = ne_init(ip);
  struct ne_init_params ipw = {1u << 20, NULL, true};
  struct ne_context* wctx = ne_init(ipw);
  if (!ctx || !wctx) return -1;
  struct ne_tensor* wq = btla_tensor(wctx, bq, sq, k, n);
  struct ne_tensor* wk = btla_tensor(wctx, bk, sk, k, n);
  struct ne_tensor* wv = btla_tensor(wctx, bv, sv, k, n);
  struct ne_tensor* x = ne_new_tensor_2d(ctx, NE_TYPE_F32, k, m, NE_SIZE_CALC, NE_BACKEND_CPU);
  memcpy(x->data, a, (size_t)m * k * 4);
  struct ne_tensor* y = ne_mul_qkv(ctx, wq, wk, wv, x);
  run_graph(ctx, y);
  memcpy(out, y->data, (size_t)3 * m * n * 4);
  ne_free(ctx);
  ne_free(wctx);
  return 0;
}

/* ne_norm / ne_rms_norm (ne_compute_forward_norm_f32 / _rms_norm_f32) on [rows][cols] fp32 */
int neref_norm(const float* x, float* y, int rows, int cols, float eps, int is_rms) {
  const size_t n = (size_t)rows * cols;
  struct ne_init_params ip = {n * 8 + (16u << 20), NULL, false};
  struct ne_context* ctx = ne_init(ip);
  if (!ctx) return -1;
  struct ne_tensor* a = ne_new_tensor_2d(ctx, NE_TYPE_F32, cols, rows, NE_SIZE_CALC, NE_BACKEND_CPU);
  memcpy(a->data, x, n * 4);
  struct ne_tensor* r = is_rms ? ne_rms_norm(ctx, a, eps) : ne_norm(ctx, a, eps);
  run_graph(ctx, r);
  memcpy(y, r->data, n * 4);
  ne_free(ctx);
  return 0;
}

/* Library-managed kv-cache through the reference's graph nodes, built the way models/llama/llama.cpp:496-571 builds them:
 * bestla_reordered_attn_fp32_batch_kv_info sizes two NE_TYPE_BTLA cache tensors, ne_flash_attn_update_k / _v append fp32
 * K / V rows (NE_OP_FLASH_ATTN_KV_UPDATE -> bestla_reordered_attn_fp32_update_k / _v, ne_layers.c:10529-10558), then ne_flash_attn over views whose strides come from the
 * info struct (-> ne_compute_forward_flash_attn_reordered -> bestla_reordered_attn_fp32_forward, :10216-10280).
 * q [bs][sl_q][heads][hs]; kall / vall [bs][sl_kv][heads_kv][hs]: with split != 0 the first sl_kv - sl_q rows are appended
 * by one update pair and the last sl_q rows by a second one (seq_off > 0).
 * The ring-full branch (ne_rope_shift_inplace over the K view, llama.cpp:551-558) is NOT driven: ne_rope_impl builds its
 * parameter tensor with 5 + batch elements (ne_layers.c:3402) while ne_compute_forward_rope_bestla and _rope_f16 assert
 * exactly 5 (:9437, :9619) and _rope_f32 refuses a shift (:9311) — no shift-RoPE node of the reference can execute as
 * shipped.  bestla_reordered_attn_fp32_shift_rope_k is tested by direct calls against the restated arithmetic instead. */
int neref_reordered_attn(const float* q, const float* kall, const float* vall, float* out, int bs, int heads, int heads_kv,
                         int hs, int sl_q, int sl_kv, int n_ctx, float scale, unsigned flags, int split) {
  kv_shape_t ks = {(uint32_t)heads_kv, (uint32_t)hs, (uint32_t)n_ctx};
  kv_cache_info_t info;
  memset(&info, 0, sizeof(info));
  bestla_reordered_attn_fp32_batch_kv_info(&ks, &info);
  if (!info.k_bytes || !info.v_bytes) return -2;
  const size_t fl = (size_t)bs * ((size_t)sl_q * heads * 2 + (size_t)sl_kv * heads_kv * 4) * hs * 4;
  struct ne_init_params ip = {(size_t)bs * (info.k_bytes + info.v_bytes) + fl + (128u << 20), NULL, false};
  struct ne_context* ctx = ne_init(ip);
  if (!ctx) return -1;
  struct ne_tensor* kc = ne_new_tensor_1d(ctx, NE_TYPE_BTLA, (int64_t)bs * info.k_bytes, (size_t)bs * info.k_bytes, NE_BACKEND_CPU);
  struct ne_tensor* vc = ne_new_tensor_1d(ctx, NE_TYPE_BTLA, (int64_t)bs * info.v_bytes, (size_t)bs * info.v_bytes, NE_BACKEND_CPU);
  memset(kc->data, 0x7f, (size_t)bs * info.k_bytes); /* garbage: rows past what was appended must never be read */
  memset(vc->data, 0x7f, (size_t)bs * info.v_bytes);
  memset(&g_graph, 0, sizeof(g_graph));
  g_graph.n_threads = 1;
  const int past = sl_kv - sl_q;
  const int nupd = (split && past > 0) ? 2 : 1;
  for (int u = 0; u < nupd; u++) {
    const int off = (nupd == 2 && u == 1) ? past : 0;
    const int len = nupd == 2 ? (u == 0 ? past : sl_q) : sl_kv;
    /* cur tensors ne = (head_size, heads_kv, len, bs): contiguous copies of the rows [off, off + len) */
    struct ne_tensor* kcur = ne_new_tensor_4d(ctx, NE_TYPE_F32, hs, heads_kv, len, bs, NE_SIZE_CALC, NE_BACKEND_CPU);
    struct ne_tensor* vcur = ne_new_tensor_4d(ctx, NE_TYPE_F32, hs, heads_kv, len, bs, NE_SIZE_CALC, NE_BACKEND_CPU);
    const size_t rows = (size_t)len * heads_kv * hs;
    for (int b = 0; b < bs; b++) {
      memcpy((float*)kcur->data + b * rows, kall + ((size_t)b * sl_kv + off) * heads_kv * hs, rows * 4);
      memcpy((float*)vcur->data + b * rows, vall + ((size_t)b * sl_kv + off) * heads_kv * hs, rows * 4);
    }
    struct ne_tensor* kcg = ne_view_4d(ctx, kc, hs, n_ctx, heads_kv, bs, 0, 0, info.k_bytes, 0);
    struct ne_tensor* vcg = ne_view_4d(ctx, vc, hs, n_ctx, heads_kv, bs, 0, 0, info.v_bytes, 0);
    ne_build_forward_expand(&g_graph, ne_flash_attn_update_k(ctx, kcg, kcur, off, false));
    ne_build_forward_expand(&g_graph, ne_flash_attn_update_v(ctx, vcg, vcur, off, false));
  }
  struct ne_tensor* qcur = ne_new_tensor_4d(ctx, NE_TYPE_F32, hs, heads, sl_q, bs, NE_SIZE_CALC, NE_BACKEND_CPU);
  memcpy(qcur->data, q, (size_t)bs * sl_q * heads * hs * 4);
  struct ne_tensor* Q = ne_permute(ctx, qcur, 0, 2, 1, 3);
  struct ne_tensor* K = ne_view_4d(ctx, kc, hs, sl_kv, heads_kv, bs, info.stride_k_sl, info.stride_k_head_num, info.k_bytes, 0);
  *(ATTN_FWD_LAYOUT*)(&K->nb[0]) = info.k_layout; /* nb0 carries the layout (llama.cpp:550) */
  struct ne_tensor* V = ne_view_4d(ctx, vc, sl_kv, hs, heads_kv, bs, info.stride_v_head_size, info.stride_v_head_num, info.v_bytes, 0);
  *(ATTN_FWD_LAYOUT*)(&V->nb[0]) = info.v_layout;
  struct ne_tensor* o = ne_flash_attn(ctx, Q, K, V, scale, (ne_attn_flags_t)flags);
  ne_build_forward_expand(&g_graph, o);
  ne_graph_compute(ctx, &g_graph);
  memcpy(out, o->data, (size_t)bs * sl_q * heads * hs * 4);
  ne_free(ctx);
  return 0;
}

/* The fused-attention node as the model graphs build it (models/llama/llama.cpp fused branch): Q / K / V are PERMUTED
 * VIEWS of [bs][sl][heads][hs] buffers, ne_flash_attn creates NE_OP_FLASH_ATTN (+ its tmp tensor sized by
 * bestla_fusion_attn_workspace_size), ne_compute_forward_flash_attn_f32_f16_f16 (ne_layers.c:10110-10214) turns the
 * tensors' nb[] strides into attn_fp32_fp16_fp16_fp32_fwd_args_t and calls bestla_fusion_attn_fp32_fp16_fp16_fp32_forward.
 * q fp32 [bs][sl_q][heads][hs]; k, v fp16 bits [bs][sl_kv][heads_kv][hs]; out fp32 [bs][sl_q][heads][hs]. */
int neref_flash_attn(const float* q, const uint16_t* k, const uint16_t* v, float* out, int bs, int heads, int heads_kv, int hs,
                     int sl_q, int sl_kv, float scale, unsigned flags) {
  const size_t nq = (size_t)bs * sl_q * heads * hs, nk = (size_t)bs * sl_kv * heads_kv * hs;
  struct ne_init_params ip = {nq * 8 + nk * 4 + (128u << 20), NULL, false};
  struct ne_context* ctx = ne_init(ip);
  if (!ctx) return -1;
  struct ne_tensor* Q4 = ne_new_tensor_4d(ctx, NE_TYPE_F32, hs, heads, sl_q, bs, NE_SIZE_CALC, NE_BACKEND_CPU);
  struct ne_tensor* K4 = ne_new_tensor_4d(ctx, NE_TYPE_F16, hs, heads_kv, sl_kv, bs, NE_SIZE_CALC, NE_BACKEND_CPU);
  struct ne_tensor* V4 = ne_new_tensor_4d(ctx, NE_TYPE_F16, hs, heads_kv, sl_kv, bs, NE_SIZE_CALC, NE_BACKEND_CPU);
  memcpy(Q4->data, q, nq * 4);
  memcpy(K4->data, k, nk * 2);
  memcpy(V4->data, v, nk * 2);
  struct ne_tensor* Q = ne_permute(ctx, Q4, 0, 2, 1, 3); /* ne = {hs, sl_q, heads, bs} */
  struct ne_tensor* K = ne_permute(ctx, K4, 0, 2, 1, 3); /* ne = {hs, sl_kv, heads_kv, bs} */
  struct ne_tensor* V = ne_permute(ctx, V4, 1, 2, 0, 3); /* ne = {sl_kv, hs, heads_kv, bs} */
  struct ne_tensor* O = ne_flash_attn(ctx, Q, K, V, scale, (ne_attn_flags_t)flags);
  run_graph(ctx, O);
  memcpy(out, O->data, nq * 4);
  ne_free(ctx);
  return 0;
}

/* The remaining fused nodes of the path (ne_layers.c:7945-8022, :8076-8170):
 *   kind 0  ne_mul_mat_with_bias(w1, bias[1][n], x)            -> bestla_fusion_add_f32f32_forward (broadcast bias)
 *   kind 1  ne_ffn_gelu(w1, w2, x)                              -> bestla_fusion_FFN_GeLu_f32f32_forward
 *   kind 2  ne_ffn_add_gelu(w1, w2, b1[1][ff], b2[1][d], x)     -> bestla_fusion_FFN_Add_GeLu_f32f32_forward
 *   kind 3  ne_ffn_gelu_mul(w1, w2, w3, x)                      -> bestla_fusion_FFN_Gelu_Mul_f32f32_forward
 * w1, w3: [ff x d]; w2: [d x ff]; x: [m][d]; out: [m][ff] for kind 0, [m][d] otherwise. */
int neref_fused(int kind, const float* a, void* b1w, size_t s1, void* b2w, size_t s2, void* b3w, size_t s3, const float* bias1,
                const float* bias2, float* out, int m, int d, int ff) {
  struct ne_init_params ip = {(size_t)m * (d * 2 + ff * 3) * 4 + (size_t)(d + ff) * 4 + (64u << 20), NULL, false};
  struct ne_context* ctx = ne_init(ip);
  struct ne_init_params ipw = {1u << 20, NULL, true};
  struct ne_context* wctx = ne_init(ipw);
  if (!ctx || !wctx) return -1;
  struct ne_tensor* w1 = btla_tensor(wctx, b1w, s1, d, ff);
  struct ne_tensor* w2 = b2w ? btla_tensor(wctx, b2w, s2, ff, d) : NULL;
  struct ne_tensor* w3 = b3w ? btla_tensor(wctx, b3w, s3, d, ff) : NULL;
  struct ne_tensor* x = ne_new_tensor_2d(ctx, NE_TYPE_F32, d, m, NE_SIZE_CALC, NE_BACKEND_CPU);
  memcpy(x->data, a, (size_t)m * d * 4);
  struct ne_tensor *t1 = NULL, *t2 = NULL;
  if (bias1) {
    t1 = ne_new_tensor_2d(ctx, NE_TYPE_F32, ff, 1, NE_SIZE_CALC, NE_BACKEND_CPU);
    memcpy(t1->data, bias1, (size_t)ff * 4);
  }
  if (bias2) {
    t2 = ne_new_tensor_2d(ctx, NE_TYPE_F32, d, 1, NE_SIZE_CALC, NE_BACKEND_CPU);
    memcpy(t2->data, bias2, (size_t)d * 4);
  }
  struct ne_tensor* y = NULL;
  switch (kind) {
    case 0: y = ne_mul_mat_with_bias(ctx, w1, t1, x); break;
    case 1: y = ne_ffn_gelu(ctx, w1, w2, x); break;
    case 2: y = ne_ffn_add_gelu(ctx, w1, w2, t1, t2, x); break;
    case 3: y = ne_ffn_gelu_mul(ctx, w1, w2, w3, x); break;
    default: return -1;
  }
  run_graph(ctx, y);
  memcpy(out, y->data, (size_t)m * (kind == 0 ? ff : d) * 4);
  ne_free(ctx);
  ne_free(wctx);
  return 0;
}

/* One Llama-style decoder layer over T tokens (prefill, no cache) written the way the reference's model code builds it
 * (models/llama/llama.cpp:200-330, fused branch): rms_norm * g -> fused QKV -> views -> RoPE(q), RoPE(k) -> K, V copied
 * to fp16 -> permuted views -> flash_attn (causal) -> attention output projection -> residual -> rms_norm * g ->
 * fused FFN (SiLU) -> residual.  Every compute node reaches the bestla_* provider; views, permutes, reshapes and the
 * fp32 -> fp16 copies are the reference's own code.  x, out: fp32 [T][d]; g1, g2: [d]; MHA (heads_kv == heads). */
int neref_decoder_layer(const float* x, float* out, int T, int d, int heads, int ff, float eps, float freq_base, const float* g1,
                        const float* g2, void* bq, size_t sq, void* bk, size_t sk, void* bv, size_t sv, void* bo, size_t so,
                        void* b1, size_t s1, void* b2, size_t s2, void* b3, size_t s3) {
  const int hs = d / heads;
  struct ne_init_params ip = {(size_t)T * (d * 16 + ff * 4) * 4 + (size_t)heads * T * T * 4 + (256u << 20), NULL, false};
  struct ne_context* ctx = ne_init(ip);
  struct ne_init_params ipw = {1u << 20, NULL, true};
  struct ne_context* wctx = ne_init(ipw);
  if (!ctx || !wctx) return -1;
  struct ne_tensor* wq = btla_tensor(wctx, bq, sq, d, d);
  struct ne_tensor* wk = btla_tensor(wctx, bk, sk, d, d);
  struct ne_tensor* wv = btla_tensor(wctx, bv, sv, d, d);
  struct ne_tensor* wo = btla_tensor(wctx, bo, so, d, d);
  struct ne_tensor* w1 = btla_tensor(wctx, b1, s1, d, ff);
  struct ne_tensor* w2 = btla_tensor(wctx, b2, s2, ff, d);
  struct ne_tensor* w3 = btla_tensor(wctx, b3, s3, d, ff);
  struct ne_tensor* inp = ne_new_tensor_2d(ctx, NE_TYPE_F32, d, T, NE_SIZE_CALC, NE_BACKEND_CPU);
  memcpy(inp->data, x, (size_t)T * d * 4);
  struct ne_tensor* G1 = ne_new_tensor_1d(ctx, NE_TYPE_F32, d, NE_SIZE_CALC, NE_BACKEND_CPU);
  struct ne_tensor* G2 = ne_new_tensor_1d(ctx, NE_TYPE_F32, d, NE_SIZE_CALC, NE_BACKEND_CPU);
  memcpy(G1->data, g1, (size_t)d * 4);
  memcpy(G2->data, g2, (size_t)d * 4);

  struct ne_tensor* cur = ne_mul(ctx, ne_rms_norm(ctx, inp, eps), G1);
  struct ne_tensor* qkv = ne_mul_qkv(ctx, wq, wk, wv, cur); /* [d, T, 3] */
  const size_t fsz = sizeof(float);
  struct ne_tensor* Qc = ne_view_3d(ctx, qkv, hs, heads, T, hs * fsz, (size_t)d * fsz, 0);
  struct ne_tensor* Kc = ne_view_3d(ctx, qkv, hs, heads, T, hs * fsz, (size_t)d * fsz, (size_t)1 * T * d * fsz);
  struct ne_tensor* Vc = ne_view_3d(ctx, qkv, hs, heads, T, hs * fsz, (size_t)d * fsz, (size_t)2 * T * d * fsz);
  Qc = ne_rope_inplace(ctx, Qc, 0, hs, 0, 0, freq_base, 1.0f);
  Kc = ne_rope_inplace(ctx, Kc, 0, hs, 0, 0, freq_base, 1.0f);
  struct ne_tensor* K16 = ne_cpy(ctx, Kc, ne_new_tensor_3d(ctx, NE_TYPE_F16, hs, heads, T, NE_SIZE_CALC, NE_BACKEND_CPU));
  struct ne_tensor* V16 = ne_cpy(ctx, Vc, ne_new_tensor_3d(ctx, NE_TYPE_F16, hs, heads, T, NE_SIZE_CALC, NE_BACKEND_CPU));
  struct ne_tensor* Q = ne_permute(ctx, Qc, 0, 2, 1, 3);  /* {hs, T, heads} */
  struct ne_tensor* K = ne_permute(ctx, K16, 0, 2, 1, 3); /* {hs, T, heads} */
  struct ne_tensor* V = ne_permute(ctx, V16, 1, 2, 0, 3); /* {T, hs, heads} */
  struct ne_tensor* att = ne_flash_attn(ctx, Q, K, V, 1.0f / sqrtf((float)hs), NE_ATTN_FLAG_IS_CAUSAL); /* {hs, heads, T} */
  struct ne_tensor* att2 = ne_reshape_2d(ctx, att, d, T);
  struct ne_tensor* r1 = ne_add(ctx, inp, ne_mul_mat(ctx, wo, att2));
  struct ne_tensor* h2 = ne_mul(ctx, ne_rms_norm(ctx, r1, eps), G2);
  struct ne_tensor* y = ne_add(ctx, r1, ne_ffn_silu(ctx, w1, w2, w3, h2));
  run_graph(ctx, y);
  memcpy(out, y->data, (size_t)T * d * 4);
  ne_free(ctx);
  ne_free(wctx);
  return 0;
}

