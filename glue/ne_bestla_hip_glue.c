/*
 * glue/ne_bestla_hip_glue.c — the three `ne_tensor`-level entry points of neural-speed's BesTLA surface that cannot live
 * behind a plain C ABI because they take the reference's graph structs (/root/reference/neural_speed/core/ne_bestla.h:
 * bestla_parallel_for :27, bestla_support :81-83, bestla_backend_support :79; reference implementation
 * core/layers/ne_bestla.cpp:42-72, :176-276).  Compiled against the REFERENCE's headers; a maintainer adds this file to
 * the ne_layers target in place of core/layers/ne_bestla.cpp and links libns_hip.so, which exports every other
 * `bestla_*` symbol (include/ns_bestla.h part 1).  In this repository it is compiled into oracle/_ref/libne_ref.so
 * together with the reference's own ne_layers.c (oracle/Makefile neref), and the reference's graph executor then runs
 * on libns_hip.so through it (tests/test_reference_graph.py, tests/test_gpu_reference_graph.py).
 */
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#include "ne.h"
#include "ne_bestla.h"
#include "ne_layers.h"

void bestla_parallel_for(forward_compute_fptr f, struct ne_compute_params* mp, struct ne_tensor* node) {
  struct ne_compute_params p = *mp; /* graphs run with one host thread here: every node has n_tasks == 1 */
  p.ith = 0;
  p.nth = 1;
  p.type = NE_TASK_INIT;
  f(&p, node);
  p.type = NE_TASK_COMPUTE;
  f(&p, node);
  p.type = NE_TASK_FINALIZE;
  f(&p, node);
}

/* densely packed in ne order (the check ne_layers.c keeps to itself as a static inline, :667-673) */
static bool tensor_is_dense(const struct ne_tensor* t) {
  const size_t esz = ne_type_size(t->type);
  return t->nb[0] == esz && t->nb[1] == (t->nb[0] * (size_t)t->ne[0]) / (size_t)ne_blck_size(t->type) &&
         t->nb[2] == t->nb[1] * (size_t)t->ne[1] && t->nb[3] == t->nb[2] * (size_t)t->ne[2];
}

static int64_t tensor_rows(const struct ne_tensor* t) { return t->ne[1] * t->ne[2] * t->ne[3]; } /* ne_nrows, ne_layers.c:576 */

/* Which graph nodes this backend claims, and the host workspace each needs (the decision table of
 * core/layers/ne_bestla.cpp:205-276, minus its SYCL branches).  Claimed nodes run with n_tasks = 1: the executor calls
 * their forward on ONE host thread (bestla_parallel_for above) and the library fans out on the GPU. */
bool bestla_support(struct ne_tensor* node, int n_threads, size_t* workspace, size_t* dev_workspace) {
  (void)n_threads;
  size_t ws = 0;
  /* every node that lives on the device is this backend's (ne_bestla.cpp:209-211: the generic executor asserts a CPU node,
   * ne_layers.c:11900); its operator is one of bestla_device_* (glue/ne_bestla_hip_device.c, libns_hip.so) */
  bool claimed = node->backend == NE_BACKEND_SYCL;
  const struct ne_tensor *a = node->src0, *b = node->src1;
  switch (node->op) {
    case NE_OP_MUL_MAT:
    case NE_OP_MUL_MAT_BIAS:
    case NE_OP_MUL_MAT_ID: {
      /* MUL_MAT_ID: src0 is the first expert, the n_as experts are opt[0..] (ne_layers.c:7783-7916) */
      const struct ne_tensor* wei = node->op == NE_OP_MUL_MAT_ID ? node->opt[0] : a;
      if (a->type == NE_TYPE_BTLA) {
        if (a->backend == NE_BACKEND_CPU) /* a device-resident weight's data is a storage record, not a blob (ne_bestla.cpp:222-225) */
          ws = bestla_f32f32_get_workspace_size((int)b->ne[1], (int)wei->ne[1], (int)b->ne[0], wei->data);
        claimed = true;
      }
    } break;
    case NE_OP_MUL_QKV:
      ws = bestla_fusion_QKV_f32f32_get_workspace_size((int)a->ne[1], (int)b->ne[1], (int)b->ne[0], b->data);
      claimed = true;
      break;
    case NE_OP_MUL_FFN_SILU:
    case NE_OP_MUL_FFN_GELU:
    case NE_OP_MUL_FFN_GELU_MUL:
    case NE_OP_MUL_FFN_ADD_GELU:
      ws = bestla_fusion_FFN_f32f32_get_workspace_size((int)a->ne[1], (int)a->ne[0], (int)b->ne[1], (int)node->opt[0]->ne[1],
                                                       b->data, node->opt[0]->data);
      claimed = true;
      break;
    case NE_OP_MUL_ID_FFN_GELU:
    case NE_OP_MUL_ID_FFN_SILU: /* experts' w1 in opt[0..], w2 from opt[9] on (ne_layers.c:8053-8170) */
      ws = bestla_fusion_FFN_f32f32_get_workspace_size((int)a->ne[1], (int)a->ne[0], (int)node->opt[0]->ne[1],
                                                       (int)node->opt[9]->ne[1], node->opt[0]->data, node->opt[9]->data);
      claimed = true;
      break;
    case NE_OP_ADD:
    case NE_OP_MUL: /* bestla_add / bestla_mul: contiguous fp32, src1 one row or as many rows as src0 */
      claimed = claimed || (tensor_is_dense(b) && tensor_is_dense(a) && (tensor_rows(b) == 1 || tensor_rows(b) == tensor_rows(a)) &&
                            a->ne[0] == b->ne[0] && node->nb[0] == sizeof(float));
      break;
    case NE_OP_NORM:
    case NE_OP_RMS_NORM: /* bestla_layernormalization */
      claimed = claimed || tensor_is_dense(a);
      break;
    case NE_OP_ROPE: /* only the (CPU tile-packed) BTLA kv-cache form, which this backend never creates */
      claimed = claimed || node->type == NE_TYPE_BTLA;
      break;
    default:
      break;
  }
  if (claimed) node->n_tasks = 1;
  *workspace = ws;
  *dev_workspace = 0;
  return claimed;
}

/* Where a new node lives (ne_bestla.cpp:176-203).  Without NS_SYCL every node stays in host memory (the library's
 * host-pointer entries stage per call).  With NS_SYCL — the reference's own device switch; the device set it then calls
 * is bestla_device_* of libns_hip.so + glue/ne_bestla_hip_device.c — a BTLA matmul and the fp32 RMS_NORM / SILU / ADD /
 * MUL follow their inputs onto the device, exactly the reference's table. */
enum ne_backend bestla_backend_support(struct ne_tensor* a, struct ne_tensor* b, enum ne_op op) {
#ifdef NS_SYCL
  const bool on_device = a->backend == NE_BACKEND_SYCL || (b && b->backend == NE_BACKEND_SYCL);
  switch (op) {
    case NE_OP_MUL_MAT:
      if (a->type == NE_TYPE_BTLA) return on_device ? NE_BACKEND_SYCL : NE_BACKEND_CPU;
      break;
    case NE_OP_RMS_NORM:
    case NE_OP_SILU:
    case NE_OP_ADD:
    case NE_OP_MUL:
      if (a->type == NE_TYPE_F32) return on_device ? NE_BACKEND_SYCL : NE_BACKEND_CPU;
      break;
    default:
      break;
  }
#else
  (void)a;
  (void)b;
  (void)op;
#endif
  return NE_BACKEND_CPU;
}
