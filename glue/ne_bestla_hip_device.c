/*
 * glue/ne_bestla_hip_device.c — the tensor-level half of the device-backend set of neural-speed's BesTLA surface
 * (/root/reference/neural_speed/core/ne_bestla.h:101-111: bestla_device_mul_f32 / _add_f32 / _elewise_f32 / _rms_norm_f32 /
 * _rope_f32 / _dup_f32 / _mha_f32; reference implementation core/layers/ne_bestla_sycl.cpp:174-880).  They take ne_tensor
 * and ne_compute_params, so they are compiled against the REFERENCE's headers (with -DNS_SYCL, the reference's own switch
 * for its device hooks: ne_layers.c:4252, :4568, :5633, :6405, :6592, :9247, :9912) and forward to libns_hip.so, which
 * exports the pointer-only half under the reference's names (bestla_create_device ... bestla_device_f32f32_forward,
 * csrc/ns_device.hip).  A maintainer adds this file + ne_bestla_hip_glue.c to the ne_layers target in place of
 * core/layers/ne_bestla.cpp / ne_bestla_sycl.cpp.  params->dev_queue is a hipStream_t.
 */
#include <assert.h>
#include <math.h>
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "ne.h"
#include "ne_bestla.h"
#include "ne_layers.h"

/* libns_hip.so, include/ns_bestla.h part 3 (device pointers + stream) */
int ns_hip_binary_nd_f32(int is_mul, const float* dA, const float* dB, float* dDst, const long long ne0[4], const long long nb0[4],
                         const long long ne1[4], const long long nb1[4], const long long nbd[4], void* stream);
int ns_hip_silu_f32(const float* dSrc, float* dDst, size_t n, void* stream);
int ns_hip_layernormalization(int norm_count, int norm_size, bool isrms, float epsilon, const float* dIn, float* dOut, void* stream);
int ns_hip_rope_f32(const float* dSrc, float* dDst, int batch, int seq, int heads, int head_size, int n_past, int n_dims, int mode,
                    float freq_base, float freq_scale, float ext_factor, float attn_factor, void* stream);
int ns_hip_rope_f32_yarn(const float* dSrc, float* dDst, int batch, int seq, int heads, int head_size, int n_past, int n_dims, int mode,
                         float freq_base, float freq_scale, int n_orig_ctx, float ext_factor, float attn_factor, float beta_fast,
                         float beta_slow, void* stream);
int ns_hip_dup_f32(const float* dSrc, void* dDst, const long long ne[4], const long long src_nb[4], const long long dst_nb[4],
                   bool dst_is_f16, void* stream);
int ns_hip_mha_f32_device_layout(const float* dQ, const float* dK, const float* dV, float* dO, int batch, int seq, int seq_all, int heads,
                                 int heads_kv, int head_size, int n_ctx, float scale, int masked, void* stream);
int ns_hip_lazy_flush(void);
int ns_hip_lazy_rms_norm(int rows, int cols, float eps, const float* dIn, float* dOut, void* stream);
int ns_hip_lazy_silu(const float* dSrc, float* dDst, size_t n, void* stream);
int ns_hip_lazy_mul(const float* dA, const float* dB, float* dDst, const long long ne0[4], const long long nb0[4], const long long ne1[4],
                    const long long nb1[4], const long long nbd[4], void* stream);
const char* ns_hip_last_error(void);

static void device_fail(const char* who) { /* the reference's device functions have no error channel either */
  fprintf(stderr, "%s failed: %s\n", who, ns_hip_last_error());
  assert(0);
}

static void binary(const struct ne_compute_params* params, const struct ne_tensor* a, const struct ne_tensor* b, struct ne_tensor* dst,
                   int is_mul) {
  if (params->type == NE_TASK_INIT || params->type == NE_TASK_FINALIZE) return;
  long long ne0[4], nb0[4], ne1[4], nb1[4], nbd[4];
  for (int i = 0; i < 4; i++) {
    ne0[i] = a->ne[i], nb0[i] = (long long)a->nb[i], ne1[i] = b->ne[i], nbd[i] = (long long)dst->nb[i];
    nb1[i] = (i > 0 && b->ne[i] == 1) ? 0 : (long long)b->nb[i]; /* ne_bestla_sycl.cpp:199-202 */
  }
  /* a multiply may consume the norm / silu node recorded just before it (one launch for both: ns_hip_lazy_mul) */
  const int rc = is_mul ? ns_hip_lazy_mul((const float*)a->data, (const float*)b->data, (float*)dst->data, ne0, nb0, ne1, nb1, nbd, params->dev_queue)
                        : ns_hip_binary_nd_f32(0, (const float*)a->data, (const float*)b->data, (float*)dst->data, ne0, nb0, ne1, nb1, nbd,
                                               params->dev_queue);
  if (rc != 0) device_fail(is_mul ? "bestla_device_mul_f32" : "bestla_device_add_f32");
}
void bestla_device_mul_f32(const struct ne_compute_params* params, const struct ne_tensor* src0, const struct ne_tensor* src1,
                           struct ne_tensor* dst) {
  binary(params, src0, src1, dst, 1);
}
void bestla_device_add_f32(const struct ne_compute_params* params, const struct ne_tensor* src0, const struct ne_tensor* src1,
                           struct ne_tensor* dst) {
  binary(params, src0, src1, dst, 0);
}

/* NE_OP_SILU on contiguous data (ne_bestla_sycl.cpp:297-326 walks the flat index as well) */
void bestla_device_elewise_f32(const struct ne_compute_params* params, const struct ne_tensor* src0, struct ne_tensor* dst) {
  if (params->type == NE_TASK_INIT || params->type == NE_TASK_FINALIZE) return;
  if (dst->op != NE_OP_SILU) {
    fprintf(stderr, "bestla_device_elewise_f32: operator %d is not offloaded\n", (int)dst->op);
    assert(0);
    return;
  }
  const size_t n = (size_t)(src0->ne[0] * src0->ne[1] * src0->ne[2] * src0->ne[3]);
  if (ns_hip_lazy_silu((const float*)src0->data, (float*)dst->data, n, params->dev_queue) != 0) device_fail("bestla_device_elewise_f32");
}

/* rows of ne00 values; eps in dst->op_params (ne_bestla_sycl.cpp:328-407).  Rows must be dense and packed — what every
 * model graph hands over (the norm of a layer's input) */
void bestla_device_rms_norm_f32(const struct ne_compute_params* params, const struct ne_tensor* src0, struct ne_tensor* dst) {
  if (params->type == NE_TASK_INIT || params->type == NE_TASK_FINALIZE) return;
  float eps;
  memcpy(&eps, dst->op_params, sizeof(float));
  const long long rows = src0->ne[1] * src0->ne[2] * src0->ne[3];
  const bool packed = src0->nb[0] == sizeof(float) && src0->nb[1] == src0->nb[0] * (size_t)src0->ne[0] &&
                      src0->nb[2] == src0->nb[1] * (size_t)src0->ne[1] && src0->nb[3] == src0->nb[2] * (size_t)src0->ne[2] &&
                      dst->nb[1] == src0->nb[1] && dst->nb[2] == src0->nb[2] && dst->nb[3] == src0->nb[3];
  if (!packed) {
    fprintf(stderr, "bestla_device_rms_norm_f32: strided rows are not offloaded\n");
    assert(0);
    return;
  }
  if (ns_hip_lazy_rms_norm((int)rows, (int)src0->ne[0], eps, (const float*)src0->data, (float*)dst->data, params->dev_queue) != 0)
    device_fail("bestla_device_rms_norm_f32");
}

/* src1 (host, ne_layers.c:3403): n_past, n_dims, mode, prompt_size, n_keep; dst->op_params: freq_base, 1 / freq_scale,
 * n_orig_ctx, ext_factor, attn_factor, beta_fast, beta_slow, scale_factor (ne_bestla_sycl.cpp:430-520).  src0 / dst are
 * [head_size][heads][seq][batch] views of packed data. */
void bestla_device_rope_f32(const struct ne_compute_params* params, const struct ne_tensor* src0, const struct ne_tensor* src1,
                            struct ne_tensor* dst) {
  if (params->type == NE_TASK_INIT || params->type == NE_TASK_FINALIZE) return;
  const float* op = (const float*)dst->op_params;
  const float freq_base = op[0], freq_scale = 1.0f / op[1], ext_factor = op[3], attn_factor = op[4], beta_fast = op[5], beta_slow = op[6];
  const int n_orig_ctx = (int)op[2];
  const int32_t* sp = (const int32_t*)src1->data;
  const int n_past = sp[0], n_dims = sp[1], mode = sp[2];
  const int hs = (int)src0->ne[0], heads = (int)src0->ne[1], seq = (int)src0->ne[2], batch = (int)src0->ne[3];
  const bool packed = src0->nb[0] == sizeof(float) && src0->nb[1] == sizeof(float) * (size_t)hs && src0->nb[2] == src0->nb[1] * (size_t)heads &&
                      src0->nb[3] == src0->nb[2] * (size_t)seq && dst->nb[1] == src0->nb[1] && dst->nb[2] == src0->nb[2] &&
                      dst->nb[3] == src0->nb[3];
  if (!packed) {
    fprintf(stderr, "bestla_device_rope_f32: strided rows are not offloaded\n");
    assert(0);
    return;
  }
  if (ns_hip_lazy_flush() != 0) device_fail("bestla_device_rope_f32");
  const int rc = ext_factor != 0.0f
                     ? ns_hip_rope_f32_yarn((const float*)src0->data, (float*)dst->data, batch, seq, heads, hs, n_past, n_dims, mode, freq_base,
                                            freq_scale, n_orig_ctx, ext_factor, attn_factor, beta_fast, beta_slow, params->dev_queue)
                     : ns_hip_rope_f32((const float*)src0->data, (float*)dst->data, batch, seq, heads, hs, n_past, n_dims, mode, freq_base,
                                       freq_scale, ext_factor, attn_factor, params->dev_queue);
  if (rc != 0) device_fail("bestla_device_rope_f32");
}

/* 4-D strided copy of fp32 into fp32 / fp16 (ne_bestla_sycl.cpp:537-591): the kv-cache writes of the device graph */
void bestla_device_dup_f32(const struct ne_compute_params* params, const struct ne_tensor* src0, struct ne_tensor* dst) {
  if (params->type == NE_TASK_INIT || params->type == NE_TASK_FINALIZE) return;
  long long ne[4], snb[4], dnb[4];
  for (int i = 0; i < 4; i++) ne[i] = dst->ne[i], snb[i] = (long long)src0->nb[i], dnb[i] = (long long)dst->nb[i];
  if (dst->type != NE_TYPE_F32 && dst->type != NE_TYPE_F16) {
    fprintf(stderr, "bestla_device_dup_f32: destination type %d is not offloaded\n", (int)dst->type);
    assert(0);
    return;
  }
  /* the reference indexes the source with the DESTINATION's coordinates (same shape after the graph's permutes) */
  if (ns_hip_lazy_flush() != 0) device_fail("bestla_device_dup_f32");
  if (ns_hip_dup_f32((const float*)src0->data, dst->data, ne, snb, dnb, dst->type == NE_TYPE_F16, params->dev_queue) != 0)
    device_fail("bestla_device_dup_f32");
}

/* q [head_size][heads][seq][batch] packed; k view [head_size][seq_all][heads_kv] of a cache laid out [heads][n_ctx][head_size];
 * v view [seq_all][head_size][heads_kv] of a cache laid out [heads][head_size][n_ctx]; scale and n_ctx in dst->padding
 * (ne_layers.c:3682-3718, ne_bestla_sycl.cpp:836-878).  Causal for prompts (seq > 1), as the reference launches it. */
void bestla_device_mha_f32(const struct ne_compute_params* params, const struct ne_tensor* q, const struct ne_tensor* k,
                           const struct ne_tensor* v, struct ne_tensor* dst) {
  if (params->type == NE_TASK_INIT || params->type == NE_TASK_FINALIZE) return;
  const int hs = (int)q->ne[0], heads = (int)q->ne[1], seq = (int)q->ne[2], batch = (int)q->ne[3];
  const int seq_all = (int)k->ne[1], heads_kv = (int)k->ne[2];
  float scale;
  uint32_t n_ctx;
  memcpy(&scale, dst->padding, sizeof(float));
  memcpy(&n_ctx, dst->padding + 4, sizeof(uint32_t));
  if (ns_hip_mha_f32_device_layout((const float*)q->data, (const float*)k->data, (const float*)v->data, (float*)dst->data, batch, seq, seq_all,
                                   heads, heads_kv, hs, (int)n_ctx, scale, seq > 1, params->dev_queue) != 0)
    device_fail("bestla_device_mha_f32");
}
