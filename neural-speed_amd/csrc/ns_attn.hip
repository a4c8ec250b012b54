// ns_attn.hip — fused attention behind the reference's mha_dense C surface.
//
// attn_kernel: one 256-thread workgroup per (query row, head, batch) — the unit the reference parallelises over
// (mha_dense_wrapper.h:1429-1432).  Keys are walked in chunks of 1024 with the online-softmax recurrence, so any
// context length runs in 4 KB of LDS:
//   scores   thread t takes keys t, t+256, ... of the chunk: fp32 dot of the (pre-scaled) query in LDS with the fp16 key
//            row (16-byte loads when the head dimension is contiguous, element strides otherwise, e.g. transposed K)
//   softmax  workgroup max / sum through LDS; running (m, l) rescale the accumulator
//   P.V      thread (d, part) accumulates output dim d over every `parts`-th key of the chunk: consecutive threads read
//            consecutive halves of a V row (coalesced); partitions are summed through LDS at the end
// fp32 throughout (the reference's NE_ATTN_FLAG_PREFER_FP32 form; its default path rounds Q, K and P to bf16 and is
// checked against this form at 1e-2 by its own tests, mha_dense_tests.cpp:147).
// HBM-bound on K and V (2 * sl_kv * head_size * 2 B per query row and kv head); this first version is written for
// correctness and the decode shape, not yet tuned (no MFMA for long query blocks, no split-K over the context).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>

#include "../../include/ns_bestla.h"
#include "ns_common.h"

namespace ns {

constexpr int kAttnThreads = 256;
constexpr int kAttnChunk = 1024;

struct AttnParams {
  const float* q;
  const _Float16* k;
  const _Float16* v;
  float* dst;
  float qk_scale;   // QK_scale * Q_sc * K_sc
  float out_scale;  // V_sc / dst_sc
  uint32_t flags;
  int head_num, heads_kv, head_size, sl_q, sl_kv;
  long long step_q_bs, step_q_head_num, step_q_sl;
  long long step_k_bs, step_k_head_num, step_k_sl, step_k_head_size;
  long long step_v_bs, step_v_head_num, step_v_sl, step_v_head_size;
  long long step_dst_bs, step_dst_head_num, step_dst_sl;
  int hs_pad;  // power of two >= head_size (<= 256)
  float alibi_m0, alibi_m1;
  int alibi_log2_floor;
};

__device__ __forceinline__ float block_reduce(float v, bool is_max, float* red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float o = __shfl_xor(v, off, 64);
    v = is_max ? fmaxf(v, o) : v + o;
  }
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float r = red[0];
#pragma unroll
  for (int i = 1; i < kAttnThreads / 64; i++) r = is_max ? fmaxf(r, red[i]) : r + red[i];
  return r;
}

__global__ __launch_bounds__(kAttnThreads) void attn_kernel(const AttnParams p) {
  __shared__ float q_s[256];
  __shared__ float s_s[kAttnChunk];
  __shared__ float red[kAttnThreads / 64];
  __shared__ float part_s[kAttnThreads];

  const int i = blockIdx.x, ihn = blockIdx.y, ibs = blockIdx.z;
  const int t = threadIdx.x;
  const int hs = p.head_size;
  const int ihkv = ihn / (p.head_num / p.heads_kv);
  const bool causal = (p.flags & NS_ATTN_FLAG_IS_CAUSAL) != 0;
  const bool alibi = (p.flags & NS_ATTN_FLAG_IS_ALIBI8) != 0;
  const bool tanh30 = (p.flags & NS_ATTN_FLAG_IS_TANH30) != 0;
  const int unmasked = causal ? (p.sl_kv - p.sl_q) + i + 1 : p.sl_kv;  // mha_dense_wrapper.h:1440-1441

  const float* q = p.q + ibs * p.step_q_bs + ihn * p.step_q_head_num + i * p.step_q_sl;
  const _Float16* kb = p.k + ibs * p.step_k_bs + ihkv * p.step_k_head_num;
  const _Float16* vb = p.v + ibs * p.step_v_bs + ihkv * p.step_v_head_num;
  float* dst = p.dst + ibs * p.step_dst_bs + ihn * p.step_dst_head_num + i * p.step_dst_sl;

  if (t < hs) q_s[t] = q[t] * p.qk_scale;
  float slope = 0.f;
  if (alibi)  // mha_dense_wrapper.h:1424-1447
    slope = ihn < p.alibi_log2_floor ? powf(p.alibi_m0, float(ihn + 1))
                                     : powf(p.alibi_m1, float(2 * (ihn - p.alibi_log2_floor) + 1));
  __syncthreads();

  const int parts = kAttnThreads / p.hs_pad;  // key partitions of the P.V phase
  const int d = t % p.hs_pad, part = t / p.hs_pad;
  const bool k_vec = p.step_k_head_size == 1 && (hs & 7) == 0 && (p.step_k_sl & 7) == 0 &&
                     ((reinterpret_cast<uintptr_t>(kb) & 15) == 0);
  float m_run = -INFINITY, l_run = 0.f, acc = 0.f;

  for (int c0 = 0; c0 < unmasked; c0 += kAttnChunk) {
    const int cn = min(kAttnChunk, unmasked - c0);
    // ---- scores of this chunk ----
    float cmax = -INFINITY;
    for (int jj = t; jj < cn; jj += kAttnThreads) {
      const int j = c0 + jj;
      const _Float16* kr = kb + (long long)j * p.step_k_sl;
      float s = 0.f;
      if (k_vec) {
        typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
        for (int e = 0; e < hs; e += 8) {
          const half8_t kv = *reinterpret_cast<const half8_t*>(kr + e);
#pragma unroll
          for (int x = 0; x < 8; x++) s += q_s[e + x] * float(kv[x]);
        }
      } else {
        for (int e = 0; e < hs; e++) s += q_s[e] * float(kr[(long long)e * p.step_k_head_size]);
      }
      if (tanh30) s = 30.f * tanhf(s * (1.f / 30.f));
      s += float(j) * slope;
      s_s[jj] = s;
      cmax = fmaxf(cmax, s);
    }
    cmax = block_reduce(cmax, true, red);
    const float m_new = fmaxf(m_run, cmax);
    // ---- probabilities (unnormalised) ----
    float csum = 0.f;
    for (int jj = t; jj < cn; jj += kAttnThreads) {
      const float e = expf(s_s[jj] - m_new);
      s_s[jj] = e;
      csum += e;
    }
    csum = block_reduce(csum, false, red);  // also orders the s_s writes before the reads below
    const float resc = expf(m_run - m_new);  // exp(-inf) = 0 on the first chunk
    l_run = l_run * resc + csum;
    acc *= resc;
    m_run = m_new;
    // ---- P . V ----
    if (d < hs) {
      const _Float16* vd = vb + (long long)d * p.step_v_head_size;
      for (int jj = part; jj < cn; jj += parts) acc += s_s[jj] * float(vd[(long long)(c0 + jj) * p.step_v_sl]);
    }
    __syncthreads();  // s_s is rewritten by the next chunk
  }
  part_s[t] = acc;
  __syncthreads();
  if (t < hs) {
    float o = 0.f;
    for (int pp = 0; pp < parts; pp++) o += part_s[pp * p.hs_pad + t];
    dst[t] = o / l_run * p.out_scale;
  }
}

static bool attn_shape_ok(int head_num, int heads_kv, int head_size, int sl_q, int sl_kv, bool causal, std::string* why) {
  if (head_size < 1 || head_size > 256) {
    *why = "attention: head_size must be 1..256";
    return false;
  }
  if (heads_kv < 1 || head_num % heads_kv != 0) {
    *why = "attention: head_num must be a multiple of heads_kv";
    return false;
  }
  if (causal && sl_q > sl_kv) {
    *why = "attention: causal needs sl_q <= sl_kv";
    return false;
  }
  if (sl_q < 1 || sl_kv < 1) {
    *why = "attention: empty sequence";
    return false;
  }
  return true;
}

static hipError_t launch_attn(const attn_fp32_fp16_fp16_fp32_fwd_args_t& a, hipStream_t st, std::string* why) {
  if (a.Q_layout != ATTN_FWD_LAYOUT_PLAIN || a.K_layout != ATTN_FWD_LAYOUT_PLAIN || a.V_layout != ATTN_FWD_LAYOUT_PLAIN ||
      a.dst_layout != ATTN_FWD_LAYOUT_PLAIN) {
    *why = "attention: only ATTN_FWD_LAYOUT_PLAIN tensors are supported";
    return hipErrorInvalidValue;
  }
  if (!attn_shape_ok(a.head_num, a.heads_kv, a.head_size, a.sl_q, a.sl_kv, (a.attn_flags & NS_ATTN_FLAG_IS_CAUSAL) != 0, why))
    return hipErrorInvalidValue;
  AttnParams p;
  memset(&p, 0, sizeof(p));
  p.q = a.Q;
  p.k = reinterpret_cast<const _Float16*>(a.K);
  p.v = reinterpret_cast<const _Float16*>(a.V);
  p.dst = a.dst;
  p.qk_scale = a.QK_scale * a.Q_sc * a.K_sc;
  p.out_scale = a.V_sc / a.dst_sc;
  p.flags = a.attn_flags;
  p.head_num = a.head_num, p.heads_kv = a.heads_kv, p.head_size = a.head_size, p.sl_q = a.sl_q, p.sl_kv = a.sl_kv;
  p.step_q_bs = a.step_q_bs, p.step_q_head_num = a.step_q_head_num, p.step_q_sl = a.step_q_sl;
  p.step_k_bs = a.step_k_bs, p.step_k_head_num = a.step_k_head_num, p.step_k_sl = a.step_k_sl;
  p.step_k_head_size = a.step_k_head_size;
  p.step_v_bs = a.step_v_bs, p.step_v_head_num = a.step_v_head_num, p.step_v_sl = a.step_v_sl;
  p.step_v_head_size = a.step_v_head_size;
  p.step_dst_bs = a.step_dst_bs, p.step_dst_head_num = a.step_dst_head_num, p.step_dst_sl = a.step_dst_sl;
  p.hs_pad = a.head_size <= 64 ? 64 : (a.head_size <= 128 ? 128 : 256);
  const int lf = 1 << int(floor(log2(double(a.head_num))));  // mha_dense_wrapper.h:1424-1426
  p.alibi_log2_floor = lf;
  p.alibi_m0 = powf(2.0f, -8.f / float(lf));
  p.alibi_m1 = powf(2.0f, -4.f / float(lf));
  const dim3 grid(unsigned(a.sl_q), unsigned(a.head_num), unsigned(a.batch_size));
  if (a.head_num > 65535 || a.batch_size > 65535) {
    *why = "attention: head_num / batch_size above the grid limit";
    return hipErrorInvalidValue;
  }
  hipLaunchKernelGGL(attn_kernel, grid, dim3(kAttnThreads), 0, st, p);
  return hipGetLastError();
}

// elements spanned by a strided 4-d tensor
static size_t span4(int n0, long long s0, int n1, long long s1, int n2, long long s2, int n3, long long s3) {
  return size_t((n0 - 1) * s0 + (n1 - 1) * s1 + (n2 - 1) * s2 + (n3 - 1) * s3 + 1);
}

}  // namespace ns

using namespace ns;  // NOLINT

extern "C" {

size_t bestla_fusion_attn_workspace_size(const attn_shape_t* params) {
  (void)params;
  return 64;  // the kernel needs no caller scratch; non-zero so that callers that allocate it get a valid pointer
}

bool bestla_fusion_attn_fp32_fp16_fp16_fp32_support(const attn_shape_t* s) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return false;
  std::string why;
  return attn_shape_ok(s->head_num, s->heads_kv, s->head_size, s->sl_q, s->sl_kv, false, &why);
}

bool bestla_reordered_attn_fp32_support(const attn_shape_t* params) {
  (void)params;
  return false;
}

int ns_hip_attn_fp32_fp16_fp16_fp32_forward(const attn_fp32_fp16_fp16_fp32_fwd_args_t* a, void* stream) {
  std::string why;
  const hipError_t e = launch_attn(*a, static_cast<hipStream_t>(stream), &why);
  if (e != hipSuccess) {
    set_error(why.empty() ? std::string("attention launch: ") + hipGetErrorString(e) : why);
    return -1;
  }
  return 0;
}

void bestla_fusion_attn_fp32_fp16_fp16_fp32_forward(const attn_fp32_fp16_fp16_fp32_fwd_args_t* hp) {
  // host tensors (reference semantics): upload the spans the strides describe, run, download dst, synchronous
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
    set_error("no HIP device visible: libns_hip.so has no CPU fallback");
    fprintf(stderr, "Err: invalid parameters (bestla_fusion_attn_fp32_fp16_fp16_fp32_forward: no HIP device visible: "
                    "libns_hip.so has no CPU fallback)\n");
    return;
  }
  attn_fp32_fp16_fp16_fp32_fwd_args_t a = *hp;
  const size_t nq = span4(a.batch_size, a.step_q_bs, a.head_num, a.step_q_head_num, a.sl_q, a.step_q_sl, a.head_size, 1);
  const size_t nk = span4(a.batch_size, a.step_k_bs, a.heads_kv, a.step_k_head_num, a.sl_kv, a.step_k_sl, a.head_size,
                          a.step_k_head_size);
  const size_t nv = span4(a.batch_size, a.step_v_bs, a.heads_kv, a.step_v_head_num, a.sl_kv, a.step_v_sl, a.head_size,
                          a.step_v_head_size);
  const size_t nd = span4(a.batch_size, a.step_dst_bs, a.head_num, a.step_dst_head_num, a.sl_q, a.step_dst_sl,
                          a.head_size, 1);
  void *dq = nullptr, *dk = nullptr, *dv = nullptr, *dd = nullptr;
  bool ok = hipMalloc(&dq, nq * 4) == hipSuccess && hipMalloc(&dk, nk * 2) == hipSuccess &&
            hipMalloc(&dv, nv * 2) == hipSuccess && hipMalloc(&dd, nd * 4) == hipSuccess;
  ok = ok && hipMemcpy(dq, hp->Q, nq * 4, hipMemcpyHostToDevice) == hipSuccess &&
       hipMemcpy(dk, hp->K, nk * 2, hipMemcpyHostToDevice) == hipSuccess &&
       hipMemcpy(dv, hp->V, nv * 2, hipMemcpyHostToDevice) == hipSuccess &&
       hipMemcpy(dd, hp->dst, nd * 4, hipMemcpyHostToDevice) == hipSuccess;  // keeps bytes between strided rows
  if (ok) {
    a.Q = static_cast<float*>(dq);
    a.K = static_cast<uint16_t*>(dk);
    a.V = static_cast<uint16_t*>(dv);
    a.dst = static_cast<float*>(dd);
    ok = ns_hip_attn_fp32_fp16_fp16_fp32_forward(&a, nullptr) == 0 && hipDeviceSynchronize() == hipSuccess &&
         hipMemcpy(hp->dst, dd, nd * 4, hipMemcpyDeviceToHost) == hipSuccess;
  } else {
    set_error("attention: device allocation / upload failed");
  }
  if (dq) hipFree(dq);
  if (dk) hipFree(dk);
  if (dv) hipFree(dv);
  if (dd) hipFree(dd);
  if (!ok) fprintf(stderr, "Err: invalid parameters (bestla_fusion_attn_fp32_fp16_fp16_fp32_forward: %s)\n", ns_hip_last_error());
}

}  // extern "C"
