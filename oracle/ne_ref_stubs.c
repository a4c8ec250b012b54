/*
 * ne_ref_stubs.c — TEST INFRASTRUCTURE, linked into oracle/_ref/libne_ref.so next to the reference's ne_layers.c.
 * ne_layers.c references operators that live in other reference sources (conv / argsort / padding mask) and the whole
 * `bestla_*` C surface.  Neither belongs to what the harness pins, so they get aborting fallbacks here (deliberately
 * WITHOUT the reference headers: only the symbol names matter).  When libns_hip.so is loaded with RTLD_GLOBAL before
 * this library, the dynamic linker binds the `bestla_*` references to the product's definitions instead — that is the
 * drop-in test; without it (CPU box, RoPE pinning) the fallbacks are never reached.
 */
#include <stdio.h>
#include <stdlib.h>

#define NE_REF_STUB(name)                                                                                   \
  void name(void) {                                                                                         \
    fprintf(stderr, "ne_ref: %s reached but no provider is loaded (load libns_hip.so RTLD_GLOBAL first)\n", \
            #name);                                                                                         \
    abort();                                                                                                \
  }

/* the part-1 surface of include/ns_bestla.h (ne_bestla.h:21-83) */
NE_REF_STUB(bestla_f32f32_forward)
NE_REF_STUB(bestla_fusion_add_f32f32_forward)
NE_REF_STUB(bestla_fusion_QKV_f32f32_forward)
NE_REF_STUB(bestla_fusion_FFN_SiLu_f32f32_forward)
NE_REF_STUB(bestla_fusion_FFN_GeLu_f32f32_forward)
NE_REF_STUB(bestla_fusion_FFN_Gelu_Mul_f32f32_forward)
NE_REF_STUB(bestla_fusion_FFN_Add_GeLu_f32f32_forward)
NE_REF_STUB(bestla_fusion_attn_fp32_fp16_fp16_fp32_forward)
NE_REF_STUB(bestla_layernormalization)
NE_REF_STUB(bestla_mul)
NE_REF_STUB(bestla_add)
/* the library-managed kv-cache entries (mha_dense.h:124-172) */
NE_REF_STUB(bestla_reordered_attn_fp32_batch_kv_info)
NE_REF_STUB(bestla_reordered_attn_fp32_forward)
NE_REF_STUB(bestla_reordered_attn_fp32_shift_rope_k)
NE_REF_STUB(bestla_reordered_attn_fp32_update_k)
NE_REF_STUB(bestla_reordered_attn_fp32_update_v)
NE_REF_STUB(bestla_reordered_attn_fp32_support)
NE_REF_STUB(bestla_fusion_attn_fp32_batch_cpy_k)
NE_REF_STUB(bestla_fusion_attn_fp32_batch_cpy_v)
/* asked by the model graph builders (llama.cpp:212, :600) and the quantizer driver (through glue/bestla_gemm_hip.cpp) */
NE_REF_STUB(bestla_fusion_QKV_f32f32_support)
NE_REF_STUB(bestla_fusion_FFN_SiLu_f32f32_support)
NE_REF_STUB(bestla_fusion_FFN_GeLu_f32f32_support)
NE_REF_STUB(bestla_fusion_FFN_Add_GeLu_f32f32_support)
NE_REF_STUB(bestla_fusion_FFN_Gelu_Mul_f32f32_support)
NE_REF_STUB(bestla_fusion_add_f32f32_support)
NE_REF_STUB(bestla_fusion_attn_fp32_fp16_fp16_fp32_support)
NE_REF_STUB(bestla_fusion_attn_fp16_support)
NE_REF_STUB(ns_BTLAGemmPackBSize)
NE_REF_STUB(ns_BTLAGemmQuantPackB)
NE_REF_STUB(ns_BTLAGemmPackB)
NE_REF_STUB(ns_BTLAGemmUnPackB)
/* operators of other reference source files, outside the path */
NE_REF_STUB(ne_attention_padding_mask_f32_forward)
#ifndef NS_REF_HAVE_ARGSORT /* the model libraries compile the reference's core/layers/argsort.cpp (ne_top_k of the MoE router) */
NE_REF_STUB(ne_compute_forward_argsort)
#endif
NE_REF_STUB(ne_compute_forward_conv_1d)
NE_REF_STUB(ne_compute_forward_conv_1d_1s)
NE_REF_STUB(ne_compute_forward_conv_1d_2s)

/* size / set-up functions must be callable without a provider (graphs without BTLA nodes never need a workspace) */
void bestla_init(void) {}
void* bestla_get_thread_handle(void) { return 0; }
int bestla_set_threads(int n) {
  (void)n;
  return 1;
}
unsigned long long bestla_f32f32_get_workspace_size(int m, int n, int k, void* w) {
  (void)m, (void)n, (void)k, (void)w;
  return 0;
}
unsigned long long bestla_fusion_QKV_f32f32_get_workspace_size(int m, int n, int k, void* w) {
  (void)m, (void)n, (void)k, (void)w;
  return 0;
}
unsigned long long bestla_fusion_FFN_f32f32_get_workspace_size(int seq, int fin, int fmid, int fout, void* w1, void* w2) {
  (void)seq, (void)fin, (void)fmid, (void)fout, (void)w1, (void)w2;
  return 0;
}
unsigned long long bestla_fusion_attn_workspace_size(const void* p) {
  (void)p;
  return 0;
}
