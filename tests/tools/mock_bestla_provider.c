/* TEST INFRASTRUCTURE: a recording stand-in for libns_hip.so's part-1 surface, loaded RTLD_GLOBAL before
 * oracle/_ref/libne_ref.so so that the reference graph's bestla_* calls land here.  It proves, on a box without a GPU,
 * that the harness, the symbol interposition and the reference's argument marshalling work; the GPU test then swaps in
 * the real library (tests/test_gpu_reference_graph.py). */
#include <stddef.h>
#include <string.h>

#include "../../include/ns_bestla.h" /* the C ABI under test: struct layouts of the attention surface */

struct mock_call {
  int which; /* 1 = f32f32_forward, 2 = FFN SiLU, 3 = QKV */
  int m, n, k, lda, ldo;
  void *a, *w, *c, *ws;
  void *w1, *w2, *w3;
  int seq, fin, fmid, fout;
};
static struct mock_call g_last;
const struct mock_call* mock_last_call(void) { return &g_last; }

static attn_fp32_fp16_fp16_fp32_fwd_args_t g_attn;
const attn_fp32_fp16_fp16_fp32_fwd_args_t* mock_last_attn(void) { return &g_attn; }
size_t bestla_fusion_attn_workspace_size(const attn_shape_t* p) { return (size_t)p->head_num * p->sl_q * 16; }
void bestla_fusion_attn_fp32_fp16_fp16_fp32_forward(const attn_fp32_fp16_fp16_fp32_fwd_args_t* p) {
  g_attn = *p;
  for (int i = 0; i < p->batch_size * p->sl_q * p->head_num * p->head_size; i++) p->dst[i] = 3.f;
}

void bestla_init(void) {}
int bestla_set_threads(int n) { return n > 0 ? n : 1; }
unsigned long long bestla_f32f32_get_workspace_size(int m, int n, int k, void* w) {
  (void)n, (void)w;
  return (unsigned long long)m * k * 4;
}
unsigned long long bestla_fusion_QKV_f32f32_get_workspace_size(int m, int n, int k, void* w) {
  (void)n, (void)w;
  return (unsigned long long)m * k * 4;
}
void bestla_fusion_QKV_f32f32_forward(float* a, void* wq, void* wk, void* wv, float* out, int m, int n, int k, int lda, int ldo,
                                      void* ws) {
  memset(&g_last, 0, sizeof(g_last));
  g_last.which = 3, g_last.a = a, g_last.w1 = wq, g_last.w2 = wk, g_last.w3 = wv, g_last.c = out, g_last.ws = ws;
  g_last.m = m, g_last.n = n, g_last.k = k, g_last.lda = lda, g_last.ldo = ldo;
  for (int i = 0; i < 3 * m * n; i++) out[i] = 7.f;
}
unsigned long long bestla_fusion_FFN_f32f32_get_workspace_size(int seq, int fin, int fmid, int fout, void* w1, void* w2) {
  (void)fin, (void)fout, (void)w1, (void)w2;
  return (unsigned long long)seq * fmid * 4;
}
void bestla_f32f32_forward(float* a, void* w, float* c, int m, int n, int k, int lda, int ldo, void* ws) {
  memset(&g_last, 0, sizeof(g_last));
  g_last.which = 1, g_last.m = m, g_last.n = n, g_last.k = k, g_last.lda = lda, g_last.ldo = ldo;
  g_last.a = a, g_last.w = w, g_last.c = c, g_last.ws = ws;
  for (int i = 0; i < m; i++)
    for (int j = 0; j < n; j++) c[(size_t)i * ldo + j] = a[(size_t)i * lda] + (float)j; /* recognisable pattern */
}
void bestla_fusion_FFN_SiLu_f32f32_forward(float* a, void* w1, void* w2, void* w3, float* t1, float* t2, float* out, int seq,
                                           int fin, int fmid, int fout, void* ws) {
  (void)t1, (void)t2;
  memset(&g_last, 0, sizeof(g_last));
  g_last.which = 2, g_last.a = a, g_last.w1 = w1, g_last.w2 = w2, g_last.w3 = w3, g_last.c = out, g_last.ws = ws;
  g_last.seq = seq, g_last.fin = fin, g_last.fmid = fmid, g_last.fout = fout;
  for (int i = 0; i < seq * fout; i++) out[i] = 42.f;
}
