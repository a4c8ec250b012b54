/* TEST INFRASTRUCTURE: the part-1 surface of include/ns_bestla.h answered by the CPU ORACLE (oracle/libns_oracle.so).
 * It exists for one purpose: to debug and check, on a box without a GPU, multi-node graphs that the reference's own
 * executor (oracle/_ref/libne_ref.so) runs over BTLA tensors — the same graphs then run on libns_hip.so in the GPU test.
 * Never shipped, never loaded by the product, never measured. */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/ns_bestla.h"
#include "../../oracle/ns_oracle.h"

void bestla_init(void) {}
void* bestla_get_thread_handle(void) {
  static int handle;
  return &handle;
}
/* the model graph builders ask before choosing the fused nodes (llama.cpp:212-215, :600-603) */
static bool parses(void* w) {
  nso_blob_info info;
  return w && nso_blob_parse(w, &info) == 0;
}
bool bestla_fusion_QKV_f32f32_support(void* wq, void* wk, void* wv, int m, int n, int k) {
  (void)m, (void)n, (void)k;
  return parses(wq) && parses(wk) && parses(wv);
}
bool bestla_fusion_FFN_SiLu_f32f32_support(void* w1, void* w2, void* w3, int seq, int fin, int fmid, int fout) {
  (void)seq, (void)fin, (void)fmid, (void)fout;
  return parses(w1) && parses(w2) && parses(w3);
}
/* the quantizer entries the reference's driver reaches through glue/bestla_gemm_hip.cpp, answered by the oracle's packer.  The
 * core a new blob is laid out for follows the product's rule (csrc/ns_blob.cpp core_for_comp: the reference's walk on a
 * Sapphire-Rapids class host) so that both providers write the same bytes. */
static int core_for(int comp, uint32_t qt, bool asym, size_t bs) {
  const bool is_int = ((qt >> 8) & 0xff) == 1;
  if (comp == 4 && is_int && !(qt == NSO_S8 && asym)) {
    if (bs % 64 == 0) return NSO_CORE_AMX_INT8_KB;
    if (bs % 4 == 0) return NSO_CORE_AVX512_VNNI_KB;
  }
  if ((comp == 4 || comp == 2) && bs % 32 == 0) return NSO_CORE_AMX_BF16;
  return NSO_CORE_AVX512F;
}
size_t ns_BTLAGemmPackBSize(size_t n, size_t k, size_t blk, uint32_t qt, uint32_t st, bool asym, int comp, int* shuf) {
  const size_t bs = (long long)blk <= 0 ? k : blk;
  if (shuf) return nso_pack_size_gidx((int)n, (int)k, (int)blk, qt, st, asym, core_for(comp, qt, asym, bs));
  return nso_pack_size((int)n, (int)k, (int)blk, qt, st, asym, core_for(comp, qt, asym, bs));
}
bool ns_BTLAGemmQuantPackB(void* buf, const float* w, size_t n, size_t k, size_t ldb, size_t blk, uint32_t qt, uint32_t st, bool asym,
                           int comp, bool is_trans, void* tp) {
  (void)tp;
  const size_t bs = (long long)blk <= 0 ? k : blk;
  return nso_quant_pack(buf, w, (int)n, (int)k, (int)ldb, (int)blk, qt, st, asym, core_for(comp, qt, asym, bs), is_trans) == 0;
}

bool bestla_fusion_FFN_Add_GeLu_f32f32_support(void* w1, void* w2, int seq, int fin, int fmid, int fout) {
  (void)seq, (void)fin, (void)fmid, (void)fout;
  return parses(w1) && parses(w2);
}
bool bestla_fusion_add_f32f32_support(void* w, int m, int n, int k) {
  (void)m, (void)n, (void)k;
  return parses(w);
}
bool bestla_fusion_attn_fp32_fp16_fp16_fp32_support(const attn_shape_t* p) {
  (void)p;
  return true;
}
/* no library-managed kv cache on the CPU side: the model falls back to its own fp16 / fp32 cache and unfused attention */
bool bestla_reordered_attn_fp32_support(const attn_shape_t* p) {
  (void)p;
  return false;
}
void bestla_timer(bool m) { (void)m; }
int bestla_set_threads(int n) { return n > 0 ? n : 1; }
unsigned long long bestla_f32f32_get_workspace_size(int m, int n, int k, void* w) {
  (void)n, (void)w;
  return (unsigned long long)m * k * 4;
}
unsigned long long bestla_fusion_QKV_f32f32_get_workspace_size(int m, int n, int k, void* w) {
  (void)n, (void)w;
  return (unsigned long long)m * k * 4;
}
unsigned long long bestla_fusion_FFN_f32f32_get_workspace_size(int seq, int fin, int fmid, int fout, void* w1, void* w2) {
  (void)fin, (void)fout, (void)w1, (void)w2;
  return (unsigned long long)seq * fmid * 4;
}
size_t bestla_fusion_attn_workspace_size(const attn_shape_t* p) { return (size_t)p->head_num * p->sl_q * 16 + 64; }

static void gemm(const float* a, int lda, void* w, float* c, int ldc, int m, int n) {
  double* t = (double*)malloc((size_t)m * n * sizeof(double));
  if (nso_gemm_f64(a, lda, w, t, n, m)) abort();
  for (int i = 0; i < m; i++)
    for (int j = 0; j < n; j++) c[(size_t)i * ldc + j] = (float)t[(size_t)i * n + j];
  free(t);
}
void bestla_f32f32_forward(float* a, void* w, float* c, int m, int n, int k, int lda, int ldo, void* ws) {
  (void)k, (void)ws;
  gemm(a, lda, w, c, ldo, m, n);
}
void bestla_fusion_QKV_f32f32_forward(float* a, void* wq, void* wk, void* wv, float* out, int m, int n, int k, int lda, int ldo,
                                      void* ws) {
  (void)k, (void)ws;
  gemm(a, lda, wq, out, ldo, m, n);
  gemm(a, lda, wk, out + (size_t)m * ldo, ldo, m, n);
  gemm(a, lda, wv, out + (size_t)2 * m * ldo, ldo, m, n);
}
void bestla_fusion_FFN_SiLu_f32f32_forward(float* a, void* w1, void* w2, void* w3, float* t1, float* t2, float* out, int seq,
                                           int fin, int fmid, int fout, void* ws) {
  (void)ws;
  gemm(a, fin, w1, t1, fmid, seq, fmid);
  gemm(a, fin, w3, t2, fmid, seq, fmid);
  for (size_t i = 0; i < (size_t)seq * fmid; i++) t2[i] = t2[i] * nso_silu(t1[i]);
  gemm(t2, fmid, w2, out, fout, seq, fout);
}
/* tmp1 = gelu(A*W1), tmp2 = (A*W3) * tmp1, out = tmp2 * W2 (ip_fusion_ffn.cpp, Gelu_Mul form) */
static float gelu_tanh(float x) { return 0.5f * x * (1.f + tanhf(0.7978845834732056f * (x + 0.044714998453855515f * x * x * x))); }
/* ne_bestla.h:36-38 / inner_product.cpp: output = A W + bias (bias row 0 for every row when boardcast_bias) */
void bestla_fusion_add_f32f32_forward(float* a, void* w, float* bias, float* out, int m, int n, int k, int lda, int ldo,
                                      bool boardcast_bias, void* ws) {
  (void)k, (void)ws;
  gemm(a, lda, w, out, ldo, m, n);
  for (int i = 0; i < m; i++)
    for (int j = 0; j < n; j++) out[(size_t)i * ldo + j] += bias[(boardcast_bias ? 0 : (size_t)i * n) + j];
}
/* ip_fusion_ffn.cpp: out = gelu(A W1 + b1) W2 + b2 */
void bestla_fusion_FFN_Add_GeLu_f32f32_forward(float* a, void* w1, void* w2, float* b1, float* b2, float* t1, float* out, int seq,
                                               int fin, int fmid, int fout, bool boardcast_bias, void* ws) {
  (void)ws;
  gemm(a, fin, w1, t1, fmid, seq, fmid);
  for (int i = 0; i < seq; i++)
    for (int j = 0; j < fmid; j++)
      t1[(size_t)i * fmid + j] = gelu_tanh(t1[(size_t)i * fmid + j] + b1[(boardcast_bias ? 0 : (size_t)i * fmid) + j]);
  gemm(t1, fmid, w2, out, fout, seq, fout);
  for (int i = 0; i < seq; i++)
    for (int j = 0; j < fout; j++) out[(size_t)i * fout + j] += b2[(boardcast_bias ? 0 : (size_t)i * fout) + j];
}
void bestla_fusion_FFN_Gelu_Mul_f32f32_forward(float* a, void* w1, void* w2, void* w3, float* t1, float* t2, float* out, int seq,
                                               int fin, int fmid, int fout, void* ws) {
  (void)ws;
  gemm(a, fin, w1, t1, fmid, seq, fmid);
  gemm(a, fin, w3, t2, fmid, seq, fmid);
  for (size_t i = 0; i < (size_t)seq * fmid; i++) {
    const float x = t1[i];
    t2[i] = t2[i] * (0.5f * x * (1.f + tanhf(0.7978845834732056f * (x + 0.044714998453855515f * x * x * x))));
  }
  gemm(t2, fmid, w2, out, fout, seq, fout);
}
void bestla_fusion_attn_fp32_fp16_fp16_fp32_forward(const attn_fp32_fp16_fp16_fp32_fwd_args_t* p) {
  nso_attn_args a;
  memset(&a, 0, sizeof(a));
  a.q = p->Q, a.k = p->K, a.v = p->V, a.dst = p->dst;
  a.q_sc = p->Q_sc, a.k_sc = p->K_sc, a.v_sc = p->V_sc, a.dst_sc = p->dst_sc, a.qk_scale = p->QK_scale;
  a.flags = p->attn_flags & 3u;
  a.batch_size = p->batch_size, a.head_num = p->head_num, a.heads_kv = p->heads_kv, a.head_size = p->head_size;
  a.sl_q = p->sl_q, a.sl_kv = p->sl_kv;
  a.step_q_bs = p->step_q_bs, a.step_q_head_num = p->step_q_head_num, a.step_q_sl = p->step_q_sl;
  a.step_k_bs = p->step_k_bs, a.step_k_head_num = p->step_k_head_num, a.step_k_sl = p->step_k_sl;
  a.step_k_head_size = p->step_k_head_size;
  a.step_v_bs = p->step_v_bs, a.step_v_head_num = p->step_v_head_num, a.step_v_sl = p->step_v_sl;
  a.step_dst_bs = p->step_dst_bs, a.step_dst_head_num = p->step_dst_head_num, a.step_dst_sl = p->step_dst_sl;
  if (nso_attn_ref(&a, 0)) abort();
}
void bestla_layernormalization(int norm_count, int norm_size, bool isrms, float eps, const float* in, float* out) {
  for (int r = 0; r < norm_count; r++) {
    const float* x = in + (size_t)r * norm_size;
    float* y = out + (size_t)r * norm_size;
    double mean = 0, sq = 0;
    for (int i = 0; i < norm_size; i++) mean += x[i], sq += (double)x[i] * x[i];
    mean /= norm_size;
    if (isrms) {
      const double s = 1.0 / sqrt(sq / norm_size + eps);
      for (int i = 0; i < norm_size; i++) y[i] = (float)(x[i] * s);
    } else {
      const double var = sq / norm_size - mean * mean, s = 1.0 / sqrt(var + eps);
      for (int i = 0; i < norm_size; i++) y[i] = (float)((x[i] - mean) * s);
    }
  }
}
void bestla_mul(int batch, int vsize, const float* t, const float* v, int vstep, float* out) {
  for (int b = 0; b < batch; b++)
    for (int i = 0; i < vsize; i++) out[(size_t)b * vsize + i] = t[(size_t)b * vsize + i] * v[(size_t)b * vstep + i];
}
void bestla_add(int batch, int vsize, const float* t, const float* v, int vstep, float* out) {
  for (int b = 0; b < batch; b++)
    for (int i = 0; i < vsize; i++) out[(size_t)b * vsize + i] = t[(size_t)b * vsize + i] + v[(size_t)b * vstep + i];
}
