/*
 * attn_shim.cpp — TEST INFRASTRUCTURE.  C entry point over the reference's OWN attention reference,
 *   bestla_fusion_attn_forward_ref<float, fp16, fp16, float>   (neural_speed/core/layers/mha_dense_wrapper.h:1370-1514,
 *   the function its test suite compares the JIT kernels with, mha_dense_tests.cpp:147-284),
 * compiled from where it lies.  mha_dense_wrapper.h as a whole needs the xbyak JIT headers (its first 1360 lines are the
 * JIT MHA classes); the recipe (oracle/Makefile attnref) therefore cuts the four self-contained pieces the function
 * consists of out of the reference file AT BUILD TIME into oracle/_ref/attn_inc/mha_ref_extract.h — MHA_2ND_EXP /
 * MHA_PREFER_AVX512FP16 (:41-42), struct attn_fwd_args_t (:57-73), mha_exp_ref (:79-85) and the function itself — the way
 * packref cuts bestla_gemm.h down to its first 123 lines.  Everything the pieces reference comes from reference headers
 * that compile standalone (bestla_utils.h: fp16 / bf16; kernel_ref.h: exp_ps_0_1; mha_dense.h / data_types.h: the C
 * argument types) except ne_threading (bestla_common.hpp -> JIT): a serial parallel_for_collapse stands in below.
 * Nothing from the reference is stored in this repository.
 */
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <memory>
#include <random>
#include <type_traits>
#include <vector>

#include "bestla_utils.h"
#include "kernel_ref.h"
extern "C" {
#include "core/data_types.h"
#include "layers/mha_dense.h"
}

namespace ne_bestla {
struct serial_threading {
  void parallel_for_collapse(int b1, int e1, int s1, int b2, int e2, int s2, const std::function<void(int, int)>& f) {
    for (int i = b1; i < e1; i += s1)
      for (int j = b2; j < e2; j += s2) f(i, j);
  }
};
struct ne_threading {
  static serial_threading* get() {
    static serial_threading t;
    return &t;
  }
};
namespace custom {
namespace mha {
using namespace bestla;  // NOLINT
using bestla::utils::bf16;
using bestla::utils::fp16;
#include "mha_ref_extract.h"
}  // namespace mha
}  // namespace custom
}  // namespace ne_bestla

extern "C" {

/* forward_ref fills p.tmp with this many bytes of 'f' before it starts (it never reads them) */
size_t bestla_fusion_attn_workspace_size(const attn_shape_t* s) { return size_t(s->batch_size) * size_t(s->head_num) * 64; }

/* the argument struct of oracle/ns_oracle.h (nso_attn_args), same field order */
struct attnref_args {
  const float* q;
  const uint16_t* k;
  const uint16_t* v;
  float* dst;
  float q_sc, k_sc, v_sc, dst_sc, qk_scale;
  uint32_t flags; /* NE_ATTN_FLAG_* as the reference defines them */
  int batch_size, head_num, heads_kv, head_size, sl_q, sl_kv;
  long long step_q_bs, step_q_head_num, step_q_sl;
  long long step_k_bs, step_k_head_num, step_k_sl, step_k_head_size;
  long long step_v_bs, step_v_head_num, step_v_sl;
  long long step_dst_bs, step_dst_head_num, step_dst_sl;
};

int attnref_forward_f32_f16_f16_f32(const attnref_args* a) {
  using namespace ne_bestla::custom::mha;  // NOLINT
  attn_fwd_args_t<float, fp16, fp16, float> p;
  memset(&p, 0, sizeof(p));
  p.Q = const_cast<float*>(a->q);
  p.K = reinterpret_cast<fp16*>(const_cast<uint16_t*>(a->k));
  p.V = reinterpret_cast<fp16*>(const_cast<uint16_t*>(a->v));
  p.dst = a->dst;
  p.Q_sc = a->q_sc, p.K_sc = a->k_sc, p.V_sc = a->v_sc, p.dst_sc = a->dst_sc;
  p.QK_scale = a->qk_scale;
  p.attn_flags = static_cast<ne_attn_flags_t>(a->flags);
  p.batch_size = a->batch_size, p.head_num = a->head_num, p.heads_kv = a->heads_kv, p.head_size = a->head_size;
  p.sl_q = a->sl_q, p.sl_kv = a->sl_kv;
  p.Q_layout = p.K_layout = p.V_layout = p.dst_layout = ATTN_FWD_LAYOUT_PLAIN;
  p.step_q_bs = int(a->step_q_bs), p.step_q_head_num = int(a->step_q_head_num), p.step_q_sl = int(a->step_q_sl);
  p.step_k_bs = int(a->step_k_bs), p.step_k_head_num = int(a->step_k_head_num), p.step_k_sl = int(a->step_k_sl);
  p.step_k_head_size = int(a->step_k_head_size);
  p.step_v_bs = int(a->step_v_bs), p.step_v_head_num = int(a->step_v_head_num), p.step_v_sl = int(a->step_v_sl);
  p.step_v_head_size = 1;
  p.step_dst_bs = int(a->step_dst_bs), p.step_dst_head_num = int(a->step_dst_head_num), p.step_dst_sl = int(a->step_dst_sl);
  attn_shape_t shape{p.batch_size, p.head_num, p.heads_kv, p.head_size, p.sl_q, p.sl_kv};
  std::vector<char> tmp(bestla_fusion_attn_workspace_size(&shape));
  p.tmp = tmp.data();
  bestla_fusion_attn_forward_ref(p);
  return 0;
}

/* the flag values of the build, so the caller need not restate the enum */
unsigned attnref_flag_causal(void) { return NE_ATTN_FLAG_IS_CAUSAL; }
unsigned attnref_flag_alibi8(void) { return NE_ATTN_FLAG_IS_ALIBI8; }
unsigned attnref_flag_prefer_fp32(void) { return NE_ATTN_FLAG_PREFER_FP32; }

}  // extern "C"
