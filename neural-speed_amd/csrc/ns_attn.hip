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
//
// attn_split_kernel (the fast path: contiguous head dimension, head group of 1/2/4/8): flash-decoding layout.
//   grid = (context splits, kv heads x query rows, batch).  A workgroup reads each K / V row of its context range ONCE
//   and serves all `G` query heads that share the kv head (GQA); 16 lanes x 16 B cover one row, so a wave works on 4 keys
//   and the workgroup on 16 keys at a time, each 16-lane group carrying its own online-softmax state (m, l, acc) — no
//   communication inside the loop.  States are merged by shuffles inside a wave, through LDS across waves, and across
//   splits by attn_merge_kernel (plain stores + a kernel boundary: no device-scope fences, which cost an L2 write-back
//   per XCD on MI355X).  HBM-bound on K and V: 2 * sl_kv * heads_kv * head_size * 2 B per query row.
#include <hip/hip_runtime.h>

#include <map>
#include <mutex>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdlib>
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
  _Float16* dst16;  // optional fp16 shadow of dst (same element strides): the next GEMM's A operand
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
  int alibi_head_off;  // tensor parallel: index of this rank's first head in the full model (0 otherwise)
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
  {
    const int gh = ihn + p.alibi_head_off;
    slope = gh < p.alibi_log2_floor ? powf(p.alibi_m0, float(gh + 1)) : powf(p.alibi_m1, float(2 * (gh - p.alibi_log2_floor) + 1));
  }
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
    const float y = o / l_run * p.out_scale;
    dst[t] = y;
    if (p.dst16) p.dst16[dst - p.dst + t] = (_Float16)y;
  }
}


// ---------------------------------------------------------------------------------------------------------------
// split-KV kernel
// ---------------------------------------------------------------------------------------------------------------
struct AttnSplitParams {
  AttnParams a;
  float* ws;       // [batch][sl_q][head][nsplit][2 + head_size] partial (m, l, acc) when nsplit > 1
  int nsplit;
  int keys_per_split;
  uint32_t* tickets;  // one self-resetting counter per (batch, query row, kv head, chunk): the LAST split to finish merges inside the
                      // launch (nullptr: attn_merge_kernel does it in a second launch)
  int heads_first;     // 1: blockIdx.x walks the (kv head, row) pairs and blockIdx.y the context ranges — neighbouring workgroups then read
                       // neighbouring 256-byte pieces of a position-major cache ([position][head][dim]); 0: the ranges of one head first
  int g_full, chunks;  // query heads per kv head, and in how many workgroups of G heads each they are served (round 4: any head
                       // group — Falcon's 71 and StarCoder's 48 query heads on one kv head ran on the one-workgroup-per-head kernel)
  // replayed device route (round 6, ns_common.h Affine): the context length moves with the captured graph's token counter —
  // sl_kv = a.sl_kv + kdelta * *kmove.  Grid, keys_per_split and the partials' layout are those of the LONGEST context (nsplit ranges);
  // ranges past the live length leave at once, a single live range writes the output row itself, the merge reads the live ones only.
  const int* kmove;
  int kdelta;
  // ... and the ranges follow the LIVE length (round 6, second half): a plan laid out for n_ctx = 2048 (8 ranges of 256 keys) used to run 1530 cached positions
  // on 6 of its 8 workgroups per head, 600 positions on 3 — with n_ctx = 4096 half as many again.  Every workgroup applies the host's range rule
  // (attn_nsplit) to the live length itself: dyn_min_keys / dyn_batch_keys / dyn_single_below are that rule's constants for this launch, nsplit its
  // upper bound (grid and partials' layout stay those of the longest context).  dyn_min_keys == 0: ranges as laid out (keys_per_split).
  int dyn_min_keys, dyn_batch_keys, dyn_single_below;
};
// live context length / live ranges of a launch (kmove == nullptr: what the host said)
__device__ __forceinline__ int attn_live_kv(const AttnSplitParams& sp) { return sp.kmove ? sp.a.sl_kv + sp.kdelta * *sp.kmove : sp.a.sl_kv; }
// keys per range at the live length (the host's rule: as many ranges as the launch has workgroups for, at least dyn_min_keys keys each, whole softmax batches)
__device__ __forceinline__ int attn_live_kps(const AttnSplitParams& sp, int sl_kv) {
  if (!sp.kmove || sp.dyn_min_keys <= 0) return sp.keys_per_split;
  if (sl_kv <= sp.dyn_single_below) return max(sl_kv, 1);
  const int want = max(1, min(sp.nsplit, (sl_kv + sp.dyn_min_keys - 1) / sp.dyn_min_keys));
  int kps = (sl_kv + want - 1) / want;
  if (sp.dyn_batch_keys > 1 && want > 1) kps = (kps + sp.dyn_batch_keys - 1) / sp.dyn_batch_keys * sp.dyn_batch_keys;
  return kps;
}
__device__ __forceinline__ int attn_live_splits(const AttnSplitParams& sp, int sl_kv) {
  if (!sp.kmove) return sp.nsplit;
  const int kps = attn_live_kps(sp, sl_kv);
  return max(1, (sl_kv + kps - 1) / kps);
}

// stores / loads past every cache (sc0 sc1): partial results cross XCDs inside one launch
__device__ __forceinline__ void st_through(float* p, float v) {
  __hip_atomic_store(reinterpret_cast<uint32_t*>(p), __builtin_bit_cast(uint32_t, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ float ld_through(const float* p) {
  return __builtin_bit_cast(float, __hip_atomic_load(reinterpret_cast<const uint32_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
}

// What follows the key loop of a split kernel (attn_split_kernel and attn_stream_kernel share it): the four key groups of a wave are
// combined by shuffles, the waves through LDS (acc_s: [4 waves][G][16 * DPL] floats), then either the output row (one split) or the
// split's partial (max, sum, unnormalised output) is stored — and, with tickets, the last split of a kv head merges inside the launch.
template <int G, int DPL, int LPK = 16>  // LPK: lanes per key (64 / LPK key groups in a wave)
__device__ __forceinline__ void attn_split_finish(const AttnSplitParams& sp, float (&acc)[G][DPL], float (&m)[G], float (&lsum)[G], float* acc_s,
                                                  float (&ml_s)[4][G][2], int split, int chunk, int ihkv, int i, int ibs, int live) {
  const AttnParams& p = sp.a;
  const int t = threadIdx.x, w = t >> 6, l = t & 63;
  const int hs = p.head_size;
  const int d0 = (l & (LPK - 1)) * DPL;
  auto head_of = [&](int g) { return ihkv * sp.g_full + min(chunk * G + g, sp.g_full - 1); };
  auto head_live = [&](int g) { return chunk * G + g < sp.g_full; };
  // ---- merge the 4 key groups of a wave (lanes with equal dl) by shuffles ----
#pragma unroll
  for (int g = 0; g < G; g++) {
#pragma unroll
    for (int off = LPK; off <= 32; off <<= 1) {
      const float mo = __shfl_xor(m[g], off, 64), lo = __shfl_xor(lsum[g], off, 64);
      const float mn = fmaxf(m[g], mo);
      const float ca = mn == -INFINITY ? 0.f : expf(m[g] - mn), cb = mn == -INFINITY ? 0.f : expf(mo - mn);
      lsum[g] = lsum[g] * ca + lo * cb;
#pragma unroll
      for (int e = 0; e < DPL; e++) acc[g][e] = acc[g][e] * ca + __shfl_xor(acc[g][e], off, 64) * cb;
      m[g] = mn;
    }
  }
  // ---- across the 4 waves through LDS ----
  if (l < LPK) {
#pragma unroll
    for (int g = 0; g < G; g++) {
      if (l == 0) {
        ml_s[w][g][0] = m[g];
        ml_s[w][g][1] = lsum[g];
      }
#pragma unroll
      for (int e = 0; e < DPL; e++) acc_s[(w * G + g) * LPK * DPL + d0 + e] = acc[g][e];
    }
  }
  __syncthreads();
  // thread (g, d) finishes output dim d of head g
  for (int idx = t; idx < G * LPK * DPL; idx += kAttnThreads) {
    const int g = idx / (LPK * DPL), dd = idx % (LPK * DPL);
    if (dd >= hs || !head_live(g)) continue;
    float mb = -INFINITY;
#pragma unroll
    for (int ww = 0; ww < 4; ww++) mb = fmaxf(mb, ml_s[ww][g][0]);
    float lb = 0.f, ab = 0.f;
#pragma unroll
    for (int ww = 0; ww < 4; ww++) {
      const float c = ml_s[ww][g][0] == -INFINITY ? 0.f : expf(ml_s[ww][g][0] - mb);
      lb += ml_s[ww][g][1] * c;
      ab += acc_s[(ww * G + g) * LPK * DPL + dd] * c;
    }
    const int ihn = head_of(g);
    if (live == 1) {
      float* dst = p.dst + ibs * p.step_dst_bs + ihn * p.step_dst_head_num + i * p.step_dst_sl;
      const float y = ab / lb * p.out_scale;
      dst[dd] = y;
      if (p.dst16) p.dst16[dst - p.dst + dd] = (_Float16)y;
    } else {
      float* wp = sp.ws + ((((size_t)ibs * p.sl_q + i) * p.head_num + ihn) * sp.nsplit + split) * (2 + hs);
      if (sp.tickets) {
        if (dd == 0) {
          st_through(wp, mb);
          st_through(wp + 1, lb);
        }
        st_through(wp + 2 + dd, ab);
      } else {
        if (dd == 0) {
          wp[0] = mb;
          wp[1] = lb;
        }
        wp[2 + dd] = ab;
      }
    }
  }
  if (live == 1 || !sp.tickets) return;
  // ---- merge inside the launch: the write-through stores above are drained, the workgroup draws a ticket, and the one that
  // draws the last of (batch, query row, kv head) combines all splits — the sums attn_merge_kernel forms, in the same order ----
  __shared__ uint32_t drawn_s;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  uint32_t* tk = sp.tickets + (((size_t)ibs * p.sl_q + i) * p.heads_kv + ihkv) * sp.chunks + chunk;
  if (t == 0) drawn_s = __hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (drawn_s != uint32_t(live - 1)) return;
  if (t == 0) __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // zero again for the next launch
  const int ns = live;
  for (int idx = t; idx < G * LPK * DPL; idx += kAttnThreads) {
    const int g = idx / (LPK * DPL), dd = idx % (LPK * DPL);
    if (dd >= hs || !head_live(g)) continue;
    const int ihn = head_of(g);
    const float* wq = sp.ws + (((size_t)ibs * p.sl_q + i) * p.head_num + ihn) * sp.nsplit * (2 + hs);
    float mb = -INFINITY, lb = 0.f, ab = 0.f;
    if (ns <= 16) {  // every load of the merge requested at once
      float mv[16], lv[16], ov[16];
#pragma unroll
      for (int s2 = 0; s2 < 16; s2++) {
        const bool on = s2 < ns;
        mv[s2] = on ? ld_through(wq + s2 * (2 + hs)) : -INFINITY;
        lv[s2] = on ? ld_through(wq + s2 * (2 + hs) + 1) : 0.f;
        ov[s2] = on ? ld_through(wq + s2 * (2 + hs) + 2 + dd) : 0.f;
      }
#pragma unroll
      for (int s2 = 0; s2 < 16; s2++) mb = fmaxf(mb, mv[s2]);
#pragma unroll
      for (int s2 = 0; s2 < 16; s2++) {
        if (s2 < ns) {
          const float c = mv[s2] != -INFINITY ? expf(mv[s2] - mb) : 0.f;
          lb += lv[s2] * c;
          ab += ov[s2] * c;
        }
      }
    } else {
      for (int s2 = 0; s2 < ns; s2++) mb = fmaxf(mb, ld_through(wq + s2 * (2 + hs)));
      for (int s2 = 0; s2 < ns; s2++) {
        const float ms = ld_through(wq + s2 * (2 + hs));
        const float c = ms != -INFINITY ? expf(ms - mb) : 0.f;
        lb += ld_through(wq + s2 * (2 + hs) + 1) * c;
        ab += ld_through(wq + s2 * (2 + hs) + 2 + dd) * c;
      }
    }
    float* dst = p.dst + ibs * p.step_dst_bs + ihn * p.step_dst_head_num + i * p.step_dst_sl;
    const float y = ab / lb * p.out_scale;
    dst[dd] = y;
    if (p.dst16) p.dst16[dst - p.dst + dd] = (_Float16)y;
  }
}

template <int G, int DPL>  // G query heads per kv head; DPL head dims per lane (8: head_size <= 128, 16: <= 256)
__global__ __launch_bounds__(kAttnThreads) void attn_split_kernel(const AttnSplitParams sp) {
  const AttnParams& p = sp.a;
  __shared__ float ml_s[4][G][2];
  extern __shared__ float acc_s[];  // [4 waves][G][16 * DPL]
  const int split = sp.heads_first ? blockIdx.y : blockIdx.x;
  const int by = sp.heads_first ? blockIdx.x : blockIdx.y;
  const int chunk = by % sp.chunks, ykv = by / sp.chunks;
  const int ihkv = ykv % p.heads_kv, i = ykv / p.heads_kv, ibs = blockIdx.z;
  // query head of slot g: heads past the group's end repeat its last head (computed, never stored)
  auto head_of = [&](int g) { return ihkv * sp.g_full + min(chunk * G + g, sp.g_full - 1); };
  auto head_live = [&](int g) { return chunk * G + g < sp.g_full; };
  const int t = threadIdx.x, w = t >> 6, l = t & 63;
  const int kg = t >> 4;  // key group 0..15
  const int dl = l & 15;  // dim lane
  const int hs = p.head_size;
  const bool causal = (p.flags & NS_ATTN_FLAG_IS_CAUSAL) != 0;
  const bool alibi = (p.flags & NS_ATTN_FLAG_IS_ALIBI8) != 0;
  const bool tanh30 = (p.flags & NS_ATTN_FLAG_IS_TANH30) != 0;
  const int sl_kv = attn_live_kv(sp), live = attn_live_splits(sp, sl_kv);
  if (split >= live) return;  // (moving context length: a range past the live keys)
  const int unmasked = causal ? (sl_kv - p.sl_q) + i + 1 : sl_kv;
  const int kps = attn_live_kps(sp, sl_kv);
  const int j0 = split * kps, j1 = min(unmasked, j0 + kps);
  const int d0 = dl * DPL;
  const bool dact = d0 < hs;  // head sizes that are not a multiple of DPL are rejected by the host

  const _Float16* kb = p.k + ibs * p.step_k_bs + ihkv * p.step_k_head_num + d0;
  const _Float16* vb = p.v + ibs * p.step_v_bs + ihkv * p.step_v_head_num + d0;

  float q[G][DPL], acc[G][DPL], m[G], lsum[G], slope[G];
#pragma unroll
  for (int g = 0; g < G; g++) {
    const int ihn = head_of(g);
    const float* qp = p.q + ibs * p.step_q_bs + ihn * p.step_q_head_num + i * p.step_q_sl + d0;
#pragma unroll
    for (int e = 0; e < DPL; e++) {
      q[g][e] = dact ? qp[e] * p.qk_scale : 0.f;
      acc[g][e] = 0.f;
    }
    m[g] = -INFINITY;
    lsum[g] = 0.f;
    slope[g] = 0.f;
    if (alibi) {
      const int gh = ihn + p.alibi_head_off;
      slope[g] = gh < p.alibi_log2_floor ? powf(p.alibi_m0, float(gh + 1)) : powf(p.alibi_m1, float(2 * (gh - p.alibi_log2_floor) + 1));
    }
  }

  typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
  // U keys per thread and step: their K and V rows are requested together (a decode step of this kernel is bound by
  // the bytes in flight — 2 x 16 B per lane and step kept a CU at ~16 KiB, 2.4 TB/s at 2048 positions,
  // profiles/r02o), and one softmax update serves all U scores.
#ifndef NS_ATTN_U
#define NS_ATTN_U 4
#endif
  // NS_ATTN_DB=1 (round 4, measured and NOT adopted): two register sets, the rows of step s + 1 requested BEFORE step s is computed, half as
  // many keys per step.  The stream alone is 8.4 us of this kernel's ~10.8 (-DNS_ATTN_ABL=1), the steps' arithmetic sits between
  // their request batches — yet requesting ahead is SLOWER: split + merge 15.5 us against 14.4 on one box (profiles/r04bb_*), with 4
  // keys per step as well.  Like the decode GEMV, this kernel loses when more requests are in flight earlier.
#ifndef NS_ATTN_DB
#define NS_ATTN_DB 0
#endif
  constexpr int U0 = G * DPL <= 16 ? NS_ATTN_U : (G * DPL <= 64 ? 2 : 1);  // register budget: acc and q are G x DPL each (8: no gain)
  constexpr int U = NS_ATTN_DB ? (U0 > 1 ? U0 / 2 : 1) : U0;
  typedef half8_t KvRows[U][DPL / 8];
  auto request = [&](int jb, KvRows& kv, KvRows& vv) {
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int j = jb + 16 * u;
      const bool live = dact && j < j1;
#pragma unroll
      for (int c = 0; c < DPL / 8; c++) {
        kv[u][c] = half8_t{0, 0, 0, 0, 0, 0, 0, 0};
        vv[u][c] = kv[u][c];
        if (live) {
#ifndef NS_ATTN_NO_NT  // streaming (non-temporal) loads: every K / V byte is read once per token (15.5 -> 14.6 us at 2048 positions)
          kv[u][c] = __builtin_nontemporal_load(reinterpret_cast<const half8_t*>(kb + (long long)j * p.step_k_sl + 8 * c));
          vv[u][c] = __builtin_nontemporal_load(reinterpret_cast<const half8_t*>(vb + (long long)j * p.step_v_sl + 8 * c));
#else
          kv[u][c] = *reinterpret_cast<const half8_t*>(kb + (long long)j * p.step_k_sl + 8 * c);
          vv[u][c] = *reinterpret_cast<const half8_t*>(vb + (long long)j * p.step_v_sl + 8 * c);
#endif
        }
      }
    }
  };
  auto consume = [&](int jb, const KvRows& kv, const KvRows& vv) {
#if defined(NS_ATTN_ABL) && NS_ATTN_ABL == 1  // timing ablation (diagnostic builds): the K / V stream alone
#pragma unroll
    for (int u = 0; u < U; u++) acc[0][0] += float(kv[u][0][0]) + float(vv[u][0][0]);
    m[0] = 0.f, lsum[0] = 1.f;
    return;
#endif
#pragma unroll
    for (int g = 0; g < G; g++) {
      float s[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        float t2 = 0.f;
#pragma unroll
        for (int e = 0; e < DPL; e++) t2 += q[g][e] * float(kv[u][e / 8][e % 8]);
        s[u] = t2;
      }
#pragma unroll
      for (int off = 8; off > 0; off >>= 1) {
#pragma unroll
        for (int u = 0; u < U; u++) s[u] += __shfl_xor(s[u], off, 64);
      }
      float m_new = m[g];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int j = jb + 16 * u;
        if (tanh30) s[u] = 30.f * tanhf(s[u] * (1.f / 30.f));
        s[u] += float(j) * slope[g];
        if (j >= j1) s[u] = -INFINITY;
        m_new = fmaxf(m_new, s[u]);
      }
      // jb < j1: the first of the U keys is live, so m_new is finite here
      const float corr = expf(m[g] - m_new);
      float pj[U], psum = 0.f;
#pragma unroll
      for (int u = 0; u < U; u++) {
        pj[u] = expf(s[u] - m_new);  // exp(-inf) = 0 for the keys past the end
        psum += pj[u];
      }
      lsum[g] = lsum[g] * corr + psum;
#pragma unroll
      for (int e = 0; e < DPL; e++) {
        float a2 = acc[g][e] * corr;
#pragma unroll
        for (int u = 0; u < U; u++) a2 += pj[u] * float(vv[u][e / 8][e % 8]);
        acc[g][e] = a2;
      }
      m[g] = m_new;
    }
  };
  {
    KvRows kvA, vvA, kvB, vvB;
    int jb = j0 + kg;
    if (jb < j1) request(jb, kvA, vvA);
    while (jb < j1) {
      int jn = jb + 16 * U;
      if (NS_ATTN_DB && jn < j1) request(jn, kvB, vvB);
      consume(jb, kvA, vvA);
      jb = jn;
      if (jb >= j1) break;
      jn = jb + 16 * U;
      if (NS_ATTN_DB) {
        if (jn < j1) request(jn, kvA, vvA);
        consume(jb, kvB, vvB);
      } else {
        request(jb, kvA, vvA);
        continue;
      }
      jb = jn;
    }
  }

  attn_split_finish<G, DPL>(sp, acc, m, lsum, acc_s, ml_s, split, chunk, ihkv, i, ibs, live);
}

// ---------------------------------------------------------------------------------------------------------------
// attn_stream_kernel (round 5) — attn_split_kernel's arithmetic on the decode GEMV's streaming skeleton (ns_gemv.hip): the K and V rows
// of the workgroup's context range go HBM -> LDS by DMA (buffer_load ... lds, 16 B per lane, non-temporal) into a PRIVATE ring per
// wave, every request of the range in flight before the first key is multiplied; the lanes then read back exactly the 16 bytes they
// requested (lane-linear image: no bank conflicts, no staging registers held across the memory round trip) behind hand-counted
// s_waitcnt vmcnt.  What this changes against attn_split_kernel: that kernel holds U = 4 keys per lane in registers, so a range of
// 128 keys is two dependent round trips of 32 KiB per workgroup (the stream alone is 8.4 of its 10.8 us at 2048 positions,
// profiles/r04bb_*); here a workgroup has its whole range — 64 KiB at 128 keys — requested within the first few hundred cycles.
// Same key -> lane mapping (wave w, step u: keys j0 + 16 u + 4 w .. + 3, lane group l >> 4 one key each, lane l & 15 eight head
// dims), same U-key softmax update, same finish (attn_split_finish); the two kernels split a context differently (own range rule below) and hipcc contracts
// their sums differently, so their outputs agree to fp32 rounding (2e-6 relative, tests/test_gpu_attention.py), not bit for bit.
// A step = 4 keys = one 1 KiB request of K + one of V per wave; ring of kAsSteps steps per wave (refilled behind the reads for longer
// ranges).  Rows past the range's end lie outside the buffer descriptor (the DMA writes zeros for them); their scores are masked and
// their V is selected to zero as in attn_split_kernel, so nothing depends on what such a slot holds.
#ifndef NS_AS_ABL
#define NS_AS_ABL 0  // timing ablations (diagnostic builds, wrong results): 1 no arithmetic in the key loop, 2 nothing behind the key loop, 3 no key loop at all (nothing streamed), 4 streamed but never read
#endif
constexpr int kAsSteps = 8;
constexpr size_t kAsLdsBytes = size_t(4) * kAsSteps * 2048;
template <int G, int LPK = 16>  // LPK lanes per key: 16 (head sizes 72 .. 128) or 8 (40 .. 64: a 1 KiB request then holds 8 key rows, a wave step 8 keys)
__global__ __launch_bounds__(kAttnThreads) void attn_stream_kernel(const AttnSplitParams sp) {
  constexpr int DPL = 8;
  constexpr int KPW = 64 / LPK;  // keys per wave and step
  constexpr int KPS = 4 * KPW;   // keys per workgroup and step
  const AttnParams& p = sp.a;
  __shared__ float ml_s[4][G][2];
  extern __shared__ __attribute__((aligned(16))) unsigned char ring_s[];  // [4 waves][kAsSteps][K image 1 KiB | V image 1 KiB]; afterwards acc_s
  typedef __attribute__((address_space(3))) unsigned char* LdsPtr;
  const int split = sp.heads_first ? blockIdx.y : blockIdx.x;
  const int by = sp.heads_first ? blockIdx.x : blockIdx.y;
  const int chunk = by % sp.chunks, ykv = by / sp.chunks;
  const int ihkv = ykv % p.heads_kv, i = ykv / p.heads_kv, ibs = blockIdx.z;
  auto head_of = [&](int g) { return ihkv * sp.g_full + min(chunk * G + g, sp.g_full - 1); };
  const int t = threadIdx.x, l = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int dl = l & (LPK - 1);
  const int hs = p.head_size;
  const bool causal = (p.flags & NS_ATTN_FLAG_IS_CAUSAL) != 0;
  const bool alibi = (p.flags & NS_ATTN_FLAG_IS_ALIBI8) != 0;
  const bool tanh30 = (p.flags & NS_ATTN_FLAG_IS_TANH30) != 0;
  const int sl_kv = attn_live_kv(sp), live = attn_live_splits(sp, sl_kv);
  if (split >= live) return;  // (moving context length: a range past the live keys)
  const int unmasked = causal ? (sl_kv - p.sl_q) + i + 1 : sl_kv;
  const int kps = attn_live_kps(sp, sl_kv);
  const int j0 = split * kps, j1 = min(unmasked, j0 + kps);
  const int d0 = dl * DPL;
  const bool dact = d0 < hs;
  constexpr int U = G * DPL <= 16 ? 4 : 2;  // keys per lane and softmax update: attn_split_kernel's rule (the update order decides the bits)
  static_assert(kAsSteps % U == 0, "a batch of U steps never wraps inside the ring");
  const int nsteps = NS_AS_ABL == 3 ? 0 : j1 > j0 ? (((j1 - j0 + KPS - 1) / KPS) + U - 1) / U * U : 0;  // whole batches: steps past the end fetch nothing (outside the descriptor)

  // ---- 0. the query rows are requested first (ordinary loads, OLDER than the stream: loads return in order, a row requested behind the
  //      ring would be usable only when the whole ring has landed) ----
  //      Written by hand: for a load it knows hipcc waits with vmcnt(0) at the first use, i.e. for the whole ring (it does not count
  //      the LDS-DMA requests behind it); the counted wait is in step 2.
  typedef float f4_t __attribute__((ext_vector_type(4)));
  f4_t qa[G], qb[G];
#pragma unroll
  for (int g = 0; g < G; g++) {
    // (no branch around the asm and no other definition of its outputs: hipcc takes an asm's outputs for complete and would copy
    // them — e.g. to merge them with a zero — while the data is still on its way; lanes past the head size load dims 0..7, unused)
    const float* qp = p.q + ibs * p.step_q_bs + head_of(g) * p.step_q_head_num + i * p.step_q_sl + (dact ? d0 : 0);
    asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16" : "=&v"(qa[g]), "=&v"(qb[g]) : "v"(qp) : "memory");
  }
  __builtin_amdgcn_sched_barrier(0);
  // ---- 1. the whole range (up to the ring's depth) is requested before anything else is touched ----
  const _Float16* kh = p.k + ibs * p.step_k_bs + ihkv * p.step_k_head_num;
  const _Float16* vh = p.v + ibs * p.step_v_bs + ihkv * p.step_v_head_num;
  const uint32_t k_bytes = j1 > j0 ? uint32_t((size_t(j1 - 1) * p.step_k_sl + hs) * 2) : 0u;
  const uint32_t v_bytes = j1 > j0 ? uint32_t((size_t(j1 - 1) * p.step_v_sl + hs) * 2) : 0u;
  // (the descriptors are wave-uniform; said explicitly, or hipcc wraps every request into a readfirstlane loop)
  auto uniform_ptr = [](const _Float16* ptr) {
    const uint64_t v = reinterpret_cast<uint64_t>(ptr);
    const uint32_t lo = __builtin_amdgcn_readfirstlane(uint32_t(v)), hi = __builtin_amdgcn_readfirstlane(uint32_t(v >> 32));
    return reinterpret_cast<_Float16*>((uint64_t(hi) << 32) | lo);
  };
  const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(kh), 0, __builtin_amdgcn_readfirstlane(k_bytes), 0x00020000);
  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(vh), 0, __builtin_amdgcn_readfirstlane(v_bytes), 0x00020000);
  const uint32_t key_l = uint32_t(j0 + KPW * w + l / LPK);  // this lane's key of step 0
  const uint32_t k_voff = (key_l * uint32_t(p.step_k_sl) + uint32_t(d0)) * 2u, k_step = uint32_t(KPS) * uint32_t(p.step_k_sl) * 2u;
  const uint32_t v_voff = (key_l * uint32_t(p.step_v_sl) + uint32_t(d0)) * 2u, v_step = uint32_t(KPS) * uint32_t(p.step_v_sl) * 2u;
  const LdsPtr ring = (LdsPtr)(ring_s) + w * (kAsSteps * 2048);
  auto issue = [&](int s) {  // step s -> slot s % kAsSteps; the key offset is part of the per-lane offset (the part the descriptor checks)
#if defined(__HIP_DEVICE_COMPILE__)
    const LdsPtr dst = ring + (s & (kAsSteps - 1)) * 2048;
    if (dact) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, reinterpret_cast<__attribute__((address_space(3))) void*>(dst), 16, k_voff + uint32_t(s) * k_step, 0, 0, 2);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, reinterpret_cast<__attribute__((address_space(3))) void*>(dst + 1024), 16, v_voff + uint32_t(s) * v_step, 0, 0, 2);
    }
#endif
  };
#pragma unroll
  for (int s = 0; s < kAsSteps; s++)
    if (s < nsteps) issue(s);
  __builtin_amdgcn_sched_barrier(0);

  // ---- 2. the query rows' first use (their wait is counted: the stream's requests are younger and stay in flight) ----
  switch (min(nsteps, kAsSteps)) {  // requests issued behind the query loads: two per step
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
  }
  static_assert(kAsSteps == 8, "the cases above");
  float q[G][DPL], acc[G][DPL], m[G], lsum[G], slope[G];
#pragma unroll
  for (int g = 0; g < G; g++) {
    const int ihn = head_of(g);
    asm volatile("" : "+v"(qa[g]), "+v"(qb[g]));  // (uses of the rows stay behind the wait)
#pragma unroll
    for (int e = 0; e < DPL; e++) {
      q[g][e] = dact ? (e < 4 ? qa[g][e & 3] : qb[g][e & 3]) * p.qk_scale : 0.f;
      acc[g][e] = 0.f;
    }
    m[g] = -INFINITY;
    lsum[g] = 0.f;
    slope[g] = 0.f;
    if (alibi) {
      const int gh = ihn + p.alibi_head_off;
      slope[g] = gh < p.alibi_log2_floor ? powf(p.alibi_m0, float(gh + 1)) : powf(p.alibi_m1, float(2 * (gh - p.alibi_log2_floor) + 1));
    }
  }

  typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
  typedef uint32_t u4_t __attribute__((ext_vector_type(4)));
  const uint32_t lane_lds = uint32_t(reinterpret_cast<uintptr_t>(ring)) + uint32_t(l) * 16u;
  // at most `younger` STEPS requested after the batch about to be read are still in flight (two requests per step)
  auto wait_steps = [&](int younger) {
    switch (younger) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
      case 5: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    }
  };
  for (int s0 = 0; s0 < nsteps; s0 += U) {
    wait_steps(min(nsteps - s0 - U, kAsSteps - U));
    if (NS_AS_ABL == 4) {
#pragma unroll
      for (int u = 0; u < U; u++)
        if (s0 + kAsSteps + u < nsteps) issue(s0 + kAsSteps + u);
      continue;
    }
    u4_t kr[U], vr[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const uint32_t a = lane_lds + uint32_t((s0 + u) & (kAsSteps - 1)) * 2048u;
      asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024" : "=&v"(kr[u]), "=&v"(vr[u]) : "v"(a) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < U; u++) asm volatile("" : "+v"(kr[u]), "+v"(vr[u]));  // (uses stay behind the wait)
    __builtin_amdgcn_sched_barrier(0);
    // the slots are free: the steps one ring ahead are requested before this batch is multiplied
#pragma unroll
    for (int u = 0; u < U; u++)
      if (s0 + kAsSteps + u < nsteps) issue(s0 + kAsSteps + u);
    __builtin_amdgcn_sched_barrier(0);
    if (NS_AS_ABL == 1) {
      acc[0][0] += __builtin_bit_cast(float, kr[0].x ^ vr[U - 1].y);
      continue;
    }
    const int jb = j0 + KPS * s0 + KPW * w + l / LPK;
    // rows past the range's end and lanes past the head size count as zeros, as attn_split_kernel's unrequested registers do
    half8_t kv[U], vv[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const bool live = dact && jb + KPS * u < j1;
      const u4_t zero = {0u, 0u, 0u, 0u};
      kv[u] = __builtin_bit_cast(half8_t, live ? kr[u] : zero);
      vv[u] = __builtin_bit_cast(half8_t, live ? vr[u] : zero);
    }
#pragma unroll
    for (int g = 0; g < G; g++) {
      float s[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        float t2 = 0.f;
#pragma unroll
        for (int e = 0; e < DPL; e++) t2 += q[g][e] * float(kv[u][e]);
        s[u] = t2;
      }
#pragma unroll
      for (int off = LPK / 2; off > 0; off >>= 1) {
#pragma unroll
        for (int u = 0; u < U; u++) s[u] += __shfl_xor(s[u], off, 64);
      }
      float m_new = m[g];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int j = jb + KPS * u;
        if (tanh30) s[u] = 30.f * tanhf(s[u] * (1.f / 30.f));
        s[u] += float(j) * slope[g];
        if (j >= j1) s[u] = -INFINITY;
        m_new = fmaxf(m_new, s[u]);
      }
      // a lane group whose keys of this batch are all past the end keeps m = -inf: then m_new = -inf and exp(-inf - -inf) would be NaN
      const float corr = m_new == -INFINITY ? 1.f : expf(m[g] - m_new);
      float pj[U], psum = 0.f;
#pragma unroll
      for (int u = 0; u < U; u++) {
        pj[u] = m_new == -INFINITY ? 0.f : expf(s[u] - m_new);
        psum += pj[u];
      }
      lsum[g] = lsum[g] * corr + psum;
#pragma unroll
      for (int e = 0; e < DPL; e++) {
        float a2 = acc[g][e] * corr;
#pragma unroll
        for (int u = 0; u < U; u++) a2 += pj[u] * float(vv[u][e]);
        acc[g][e] = a2;
      }
      m[g] = m_new;
    }
  }
  if (NS_AS_ABL == 2) {
    if (acc[0][0] == 123.456f) p.dst[0] = acc[0][1] + m[0] + lsum[0];
    return;
  }
  __syncthreads();  // every wave has left its ring: its first bytes become attn_split_finish's acc_s
  attn_split_finish<G, DPL, LPK>(sp, acc, m, lsum, reinterpret_cast<float*>(ring_s), ml_s, split, chunk, ihkv, i, ibs, live);
}

// one workgroup per (batch, query row, head): combine the splits' (m, l, acc).  The per-split (m, l) are fetched by
// one thread each (one round trip instead of nsplit dependent ones), the weights exp(m_s - m) go through LDS, and the
// accumulator columns are summed with independent loads.
__global__ __launch_bounds__(128) void attn_merge_kernel(const AttnSplitParams sp) {
  const AttnParams& p = sp.a;
  __shared__ float c_s[64];
  __shared__ float l_s[64];
  const int ihn = blockIdx.x, i = blockIdx.y, ibs = blockIdx.z;
  const int hs = p.head_size, t = threadIdx.x;
  const int ns = attn_live_splits(sp, attn_live_kv(sp));  // ns <= 64
  if (sp.kmove && ns == 1) return;  // (a single live range wrote the output row itself)
  const float* wp = sp.ws + (((size_t)ibs * p.sl_q + i) * p.head_num + ihn) * sp.nsplit * (2 + hs);
  // the common decode shapes (<= 32 splits, one output element per thread): every thread fetches all (max, sum) pairs — wave-
  // uniform addresses, scalar loads — and its own column of the partial outputs in ONE batch of loads, then merges in registers:
  // one memory round trip behind the kernel boundary instead of two with a workgroup barrier between them (round 4: the launch
  // is 4.9 us for a few KB of work; the same sums in the same order as the general form below).  Round 5: also for 17 .. 32 splits
  // (the LDS-ring kernel splits a kv head shared by 4 query heads into 32 ranges at 2048 positions).
  auto fast = [&](auto n_c) {
    constexpr int N = decltype(n_c)::value;
    float mv[N], lv[N], ov[N];
#pragma unroll
    for (int s2 = 0; s2 < N; s2++) {
      const bool on = s2 < ns;
      mv[s2] = on ? wp[s2 * (2 + hs)] : -INFINITY;
      lv[s2] = on ? wp[s2 * (2 + hs) + 1] : 0.f;
      ov[s2] = (on && t < hs) ? wp[s2 * (2 + hs) + 2 + t] : 0.f;
    }
    float mb2 = -INFINITY;
#pragma unroll
    for (int s2 = 0; s2 < N; s2++) mb2 = fmaxf(mb2, mv[s2]);
    float lb2 = 0.f, ab2 = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < N; s2++) {
      if (s2 < ns) {
        const float c = mv[s2] != -INFINITY ? expf(mv[s2] - mb2) : 0.f;
        lb2 += lv[s2] * c;
        ab2 += ov[s2] * c;
      }
    }
    if (t < hs) {
      float* dst2 = p.dst + ibs * p.step_dst_bs + ihn * p.step_dst_head_num + i * p.step_dst_sl;
      const float y = ab2 / lb2 * p.out_scale;
      dst2[t] = y;
      if (p.dst16) p.dst16[dst2 - p.dst + t] = (_Float16)y;
    }
  };
  if (ns <= 16 && hs <= int(blockDim.x)) return fast(std::integral_constant<int, 16>{});
  if (ns <= 32 && hs <= int(blockDim.x)) return fast(std::integral_constant<int, 32>{});
  float ms = -INFINITY, ls = 0.f;
  if (t < ns) {
    ms = wp[t * (2 + hs)];
    ls = wp[t * (2 + hs) + 1];
  }
  float mb = ms;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mb = fmaxf(mb, __shfl_xor(mb, off, 64));  // splits live in wave 0 (ns <= 64)
  if (t < 64) {
    c_s[t] = (t < ns && ms != -INFINITY) ? expf(ms - mb) : 0.f;
    l_s[t] = ls;
  }
  __syncthreads();
  float lb = 0.f;
  for (int s = 0; s < ns; s++) lb += l_s[s] * c_s[s];
  float* dst = p.dst + ibs * p.step_dst_bs + ihn * p.step_dst_head_num + i * p.step_dst_sl;
  for (int d = t; d < hs; d += blockDim.x) {
    float ab = 0.f;
#pragma unroll 8
    for (int s = 0; s < ns; s++) ab += wp[s * (2 + hs) + 2 + d] * c_s[s];
    const float y = ab / lb * p.out_scale;
    dst[d] = y;
    if (p.dst16) p.dst16[dst - p.dst + d] = (_Float16)y;
  }
}

// ns_hip_set_tuning("attn_stream", 0 / 1): 1 (default) = decode attention of head sizes <= 128 streams K / V through LDS rings (attn_stream_kernel),
// 0 = attn_split_kernel (registers).  The two agree to fp32 rounding.
static std::atomic<int> g_attn_stream{getenv("NS_ATTN_STREAM") ? atoi(getenv("NS_ATTN_STREAM")) != 0 : 1};
void set_attn_stream(int on) { g_attn_stream.store(on != 0); }
// whether a fast-path call streams through the LDS rings: head sizes 40 .. 128 (72 .. 128: sixteen lanes per key; 40 .. 64: eight — with sixteen, half of a
// request's lanes would carry nothing: measured slower from 4096 positions on), row offsets inside a 32-bit buffer descriptor, 16-byte query loads
static bool attn_streams(const AttnParams& a) {
  const bool in32 = (size_t(a.sl_kv) * size_t(a.step_k_sl) + 256) * 2 < (size_t(1) << 32) && (size_t(a.sl_kv) * size_t(a.step_v_sl) + 256) * 2 < (size_t(1) << 32);
  const bool q16 = (reinterpret_cast<uintptr_t>(a.q) & 15) == 0 && a.step_q_bs % 4 == 0 && a.step_q_head_num % 4 == 0 && a.step_q_sl % 4 == 0;
  return a.head_size > 32 && a.head_size <= 128 && in32 && q16 && g_attn_stream.load(std::memory_order_relaxed) != 0;
}
template <int G>
static hipError_t launch_split_g(const AttnSplitParams& sp, dim3 grid, hipStream_t st, bool stream) {
  const int hs = sp.a.head_size;
  if (stream && hs > 64) {
    static const hipError_t attr =
        hipFuncSetAttribute(reinterpret_cast<const void*>(attn_stream_kernel<G, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, int(kAsLdsBytes));
    if (attr != hipSuccess) return attr;
    hipLaunchKernelGGL((attn_stream_kernel<G, 16>), grid, dim3(kAttnThreads), kAsLdsBytes, st, sp);
    return hipGetLastError();
  }
  if (stream) {  // head sizes 40 .. 64: eight lanes per key
    static const hipError_t attr =
        hipFuncSetAttribute(reinterpret_cast<const void*>(attn_stream_kernel<G, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, int(kAsLdsBytes));
    if (attr != hipSuccess) return attr;
    hipLaunchKernelGGL((attn_stream_kernel<G, 8>), grid, dim3(kAttnThreads), kAsLdsBytes, st, sp);
    return hipGetLastError();
  }
  if (hs <= 128) {
    hipLaunchKernelGGL((attn_split_kernel<G, 8>), grid, dim3(kAttnThreads), size_t(4) * G * 128 * 4, st, sp);
  } else {
    hipLaunchKernelGGL((attn_split_kernel<G, 16>), grid, dim3(kAttnThreads), size_t(4) * G * 256 * 4, st, sp);
  }
  return hipGetLastError();
}

// context splits of the fast path for a shape (1 = unsplit) — shared by the launcher and the workspace-size query
static std::atomic<int> g_attn_mfma2_rows{getenv("NS_ATTN_MFMA2_ROWS") ? atoi(getenv("NS_ATTN_MFMA2_ROWS")) : 128};  // query rows from which attn_mfma2_kernel serves a prefill
void set_attn_mfma2_rows(int rows) { g_attn_mfma2_rows.store(rows > 0 ? rows : 128); }
// ns_hip_set_tuning("attn_inlaunch"): 1 = the last split merges inside the launch, 0 (default) = attn_merge_kernel.  Measured equal
// (profiles/r04ab_attn_merge_in_launch.txt: split 11.8 us + merge 4.7 us against 16.5 us in one launch, whole token 600 vs 590 tok/s):
// the cost is the write-through / ticket / read-back chain across XCDs, not the kernel boundary.
static std::atomic<int> g_attn_inlaunch{getenv("NS_ATTN_INLAUNCH") ? atoi(getenv("NS_ATTN_INLAUNCH")) != 0 : 0};
static constexpr size_t kAttnTicketCap = 65536;
static std::atomic<int> g_attn_heads_first{getenv("NS_ATTN_HEADS_FIRST") ? atoi(getenv("NS_ATTN_HEADS_FIRST")) : -1};  // ns_hip_set_tuning("attn_heads_first"): -1 by layout
void set_attn_heads_first(int on) { g_attn_heads_first.store(on < 0 ? -1 : (on != 0)); }
static std::atomic<int> g_attn_wg_target{1024}, g_attn_min_keys{128};  // ns_hip_set_tuning("attn_wg_target" / "attn_min_keys")
void set_attn_inlaunch(int on) { g_attn_inlaunch.store(on != 0); }
void set_attn_tuning(int wg_target, int min_keys) {
  if (wg_target > 0) g_attn_wg_target.store(wg_target);
  if (min_keys > 0) g_attn_min_keys.store(min_keys);
}
// query heads per workgroup of the split kernel (its template argument) and workgroups per kv head for a head group of g_full
static void attn_groups(int g_full, int* G, int* chunks) {
  *G = g_full <= 1 ? 1 : (g_full == 2 ? 2 : (g_full <= 4 ? 4 : 8));
  *chunks = (g_full + *G - 1) / *G;
}
// The LDS-ring kernel has its own rule (ns_hip_set_tuning("attn_stream_wg_target" / "attn_stream_min_keys"; profiles/r05p_attn_stream.txt): a
// workgroup keeps 64 KiB in flight whatever its range, so ONE workgroup per CU streams best — 32 kv heads at 2048 positions: 8 ranges of 256
// keys 12.9 us (split + merge) against 13.7 with 16 x 128 and 15.2 with 32 x 64; 8 kv heads: 32 ranges of 64 keys 13.5 us against 16.6 with
// 16 x 128 (128 workgroups: half the chip idle).
static std::atomic<int> g_attn_wg_target_s{256}, g_attn_min_keys_s{32};
void set_attn_stream_tuning(int wg_target, int min_keys) {
  if (wg_target > 0) g_attn_wg_target_s.store(wg_target);
  if (min_keys > 0) g_attn_min_keys_s.store(min_keys);
}
// batch_keys (ring kernel): keys one softmax update of a workgroup covers (U steps); a range that is not a multiple of it computes masked key slots,
// which costs where the update is VALU-heavy (8 query heads per workgroup: Falcon-7B's 71 heads on one kv head at 2048 keys, 29 ranges of 71 keys
// 32.5 us against 16 x 128 keys 28.5) — ranges are rounded up to it while that leaves at least 128 workgroups
// heavy (workgroups that serve 4 or 8 query heads per key: twice the arithmetic per byte): two workgroups per CU overlap better than one once a batch
// brings 64+ workgroups per range — batch 8 x 32 / 8 heads at 2048 keys 36.7 -> 32.0 us, batch 16 64.0 -> 50.5 (scripts/r05/attn_batch_rule.py)
static int attn_nsplit(int batch, int heads_kv, int sl_q, int sl_kv, bool stream, int batch_keys = 0, bool heavy = false) {  // heads_kv: kv heads x workgroups per kv head
  const size_t base_blocks = size_t(heads_kv) * sl_q * batch;
  int target = stream ? g_attn_wg_target_s.load() : g_attn_wg_target.load();
  const int mk = stream ? g_attn_min_keys_s.load() : g_attn_min_keys.load();
  if (stream && heavy && base_blocks >= 64) target *= 2;
  if (stream && sl_kv <= 128 && base_blocks >= 32) return 1;  // (a range's worth of keys on 32+ workgroups: the merge launch costs more than it saves, 7.0 vs 7.8 us)
  int nsplit = int((target + base_blocks - 1) / base_blocks);     // aim at ~4 workgroups per CU
  nsplit = std::min(nsplit, std::max(1, (sl_kv + mk - 1) / mk));  // at least 128 keys per split
  nsplit = std::min(nsplit, 64);
  if (stream && batch_keys > 0 && nsplit > 1) {
    const int kps = ((sl_kv + nsplit - 1) / nsplit + batch_keys - 1) / batch_keys * batch_keys;
    const int rounded = (sl_kv + kps - 1) / kps;
    if (size_t(rounded) * base_blocks >= 128) nsplit = rounded;
  }
  return nsplit;
}
static size_t attn_ws_bytes(int batch, int head_num, int heads_kv, int head_size, int sl_q, int sl_kv) {
  int G, chunks;
  attn_groups(head_num / std::max(1, heads_kv), &G, &chunks);
  const int ns = std::max(attn_nsplit(batch, heads_kv * chunks, sl_q, sl_kv, false),
                          std::max(attn_nsplit(batch, heads_kv * chunks, sl_q, sl_kv, true), attn_nsplit(batch, heads_kv * chunks, sl_q, sl_kv, true, 0, true)));  // (any rule)
  return ns > 1 ? size_t(batch) * sl_q * head_num * ns * (2 + head_size) * 4 : 0;
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

// ============================================================================================================
// attn_mfma_kernel — prefill / multi-row attention on the matrix cores (head size 64 or 128, contiguous head dim)
//
// One 256-thread workgroup = 64 query rows of one (batch, head); wave w owns rows 16w .. 16w+15.  KV positions are
// walked in blocks of 64 (four score tiles, two 32-deep P.V slices) with the online-softmax recurrence:
//   S^T = K . Q^T   v_mfma_f32_16x16x32_f16 with A = K rows (lane (nn, g): 16 B of K[pos0 + 16t + nn][32j + 8g ..]) and
//                   B = Q rows (lane (nn, g): Q[q0 + nn][32j + 8g ..], converted to fp16 once): lane (nn, g) ends up
//                   with the scores of query q0 + nn at positions pos0 + 16t + 4g + r (t = 0, 1; r = 0..3)
//   softmax         per query = per nn: 8 values per lane, reduced over g with two xor-shuffles; running (m, l)
//   O += P . V      the 8 probabilities a lane holds ARE its A operand if MFMA k-slot 8g + i is read as position
//                   16 (i >> 2) + 4g + (i & 3) — the product does not care how k is enumerated as long as B agrees, so
//                   no data moves between lanes.  B = V in that enumeration: V^T[d][pos] staged once per block and
//                   workgroup in LDS (coalesced 16-byte global reads, transposed 2-byte LDS writes), read back as two
//                   8-byte runs (positions 4g..4g+3 and 16+4g..) per 16-column output tile
// The accumulator rows of a lane (queries q0 + 4g + r) differ from the query its softmax state belongs to (q0 + nn):
// the rescale factors travel by __shfl from lane 4g + r.  fp16 operands, fp32 scores / statistics / accumulators.
// ============================================================================================================
typedef _Float16 ahalf8_t __attribute__((ext_vector_type(8)));
typedef _Float16 ahalf4_t __attribute__((ext_vector_type(4)));
typedef float afloatx4 __attribute__((ext_vector_type(4)));
constexpr int kAttnKB = 64;           // kv positions per block (32: two barriers and one softmax update per 32 keys)
constexpr int kAttnVStr = kAttnKB + 4;  // halves per V^T row in LDS: keeps the 8-byte reads aligned, skews banks

template <int HS>
__global__ __launch_bounds__(256) void attn_mfma_kernel(const AttnParams p) {
  constexpr int NJ = HS / 32, NDT = HS / 16, CH = HS / 8;  // k-slices of QK^T, output column tiles, V halves per thread
  __shared__ __attribute__((aligned(16))) _Float16 vt[HS * kAttnVStr];
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, nn = l & 15, g = l >> 4;
  const int ihn = blockIdx.y, ibs = blockIdx.z;
  const int ihkv = ihn / (p.head_num / p.heads_kv);
  const bool causal = (p.flags & NS_ATTN_FLAG_IS_CAUSAL) != 0;
  // causal: the LAST query block sees the most keys — dispatch it first, so that the light blocks fill the tail of the
  // launch (in natural order the 32-block tiles of a 2048-token prompt started last and the kernel took twice its
  // average workgroup time, profiles/r02t_attn_prefill.txt)
  const int qblk = causal ? int(gridDim.x) - 1 - int(blockIdx.x) : int(blockIdx.x);
  const int off = p.sl_kv - p.sl_q;
  const int q0 = qblk * 64 + w * 16;
  const float* qb = p.q + ibs * p.step_q_bs + ihn * p.step_q_head_num;
  const _Float16* kb = p.k + ibs * p.step_k_bs + ihkv * p.step_k_head_num;
  const _Float16* vb = p.v + ibs * p.step_v_bs + ihkv * p.step_v_head_num;
  float* db = p.dst + ibs * p.step_dst_bs + ihn * p.step_dst_head_num;

  ahalf8_t qf[NJ];
  {
    const float* qr = qb + (long long)min(q0 + nn, p.sl_q - 1) * p.step_q_sl + 8 * g;
#pragma unroll
    for (int j = 0; j < NJ; j++)
#pragma unroll
      for (int i = 0; i < 8; i++) qf[j][i] = (_Float16)qr[32 * j + i];
  }
  afloatx4 o[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; dt++) o[dt] = afloatx4{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;
  const float sc = p.qk_scale * 1.4426950408889634f;  // scores in the exp2 domain
  const int q_last = min(qblk * 64 + 63, p.sl_q - 1);
  const int kv_end = causal ? min(p.sl_kv, q_last + off + 1) : p.sl_kv;            // workgroup-uniform
  const int visible = min(p.sl_kv, causal ? q0 + nn + off + 1 : p.sl_kv);          // mha_dense_wrapper.h:1440-1441
  const int vis_wave = min(p.sl_kv, causal ? q0 + off + 1 : p.sl_kv);              // what the wave's FIRST row sees

  constexpr int NT = kAttnKB / 16, NH = kAttnKB / 32;  // score tiles, 32-deep P.V slices per block
  // The K fragments and the V rows of block b + 1 are requested BEFORE block b is multiplied: without that every block
  // paid two exposed global round trips (K in front of the score MFMAs, V behind the first barrier) — 10 us per 64-key
  // block and workgroup, 110 TFLOPS at 2048 tokens, whatever the block size (profiles/r02t_attn_prefill.txt).
  typedef _Float16 ahalf2_t __attribute__((ext_vector_type(2)));
  const int vp = (tid >> 3) * 2, vc = tid & 7;  // V^T staging: this thread's two neighbouring positions, its CH head dims
  ahalf8_t kn[NT][NJ], vn[2][CH / 8];
  // (one register set each: the next K fragments are requested right behind the score MFMAs that consumed the current
  // ones, the next V rows right behind the LDS stores of the current ones)
  // row pointers advance by a uniform stride per block; only a block that reaches past the last key clamps its rows
  // (64-bit multiplies per row and block were a fifth of this loop's VALU work)
  const long long kstep_blk = (long long)kAttnKB * p.step_k_sl, vstep_blk = (long long)kAttnKB * p.step_v_sl;
  const _Float16* kp[NT];
  const _Float16* vpp[2];
#pragma unroll
  for (int t = 0; t < NT; t++) kp[t] = kb + (long long)(16 * t + nn) * p.step_k_sl + 8 * g;
#pragma unroll
  for (int q2 = 0; q2 < 2; q2++) vpp[q2] = vb + (long long)(vp + q2) * p.step_v_sl + vc * CH;
  auto fetch_k = [&](int pos0) {
    if (pos0 + kAttnKB <= p.sl_kv) {  // workgroup-uniform
#pragma unroll
      for (int t = 0; t < NT; t++)
#pragma unroll
        for (int j = 0; j < NJ; j++) kn[t][j] = *reinterpret_cast<const ahalf8_t*>(kp[t] + 32 * j);
    } else {
#pragma unroll
      for (int t = 0; t < NT; t++) {
        const _Float16* kr = kb + (long long)min(pos0 + 16 * t + nn, p.sl_kv - 1) * p.step_k_sl + 8 * g;
#pragma unroll
        for (int j = 0; j < NJ; j++) kn[t][j] = *reinterpret_cast<const ahalf8_t*>(kr + 32 * j);
      }
    }
#pragma unroll
    for (int t = 0; t < NT; t++) kp[t] += kstep_blk;
  };
  auto fetch_v = [&](int pos0) {
    if (pos0 + kAttnKB <= p.sl_kv) {
#pragma unroll
      for (int q2 = 0; q2 < 2; q2++)
#pragma unroll
        for (int u = 0; u < CH / 8; u++) vn[q2][u] = *reinterpret_cast<const ahalf8_t*>(vpp[q2] + 8 * u);
    } else {
#pragma unroll
      for (int q2 = 0; q2 < 2; q2++) {
        const _Float16* vr = vb + (long long)min(pos0 + vp + q2, p.sl_kv - 1) * p.step_v_sl + vc * CH;
#pragma unroll
        for (int u = 0; u < CH / 8; u++) vn[q2][u] = *reinterpret_cast<const ahalf8_t*>(vr + 8 * u);
      }
    }
#pragma unroll
    for (int q2 = 0; q2 < 2; q2++) vpp[q2] += vstep_blk;
  };
  if (kv_end > 0) {
    fetch_k(0);
    fetch_v(0);
  }
  for (int pos0 = 0; pos0 < kv_end; pos0 += kAttnKB) {
    const bool more = pos0 + kAttnKB < kv_end;
    afloatx4 sv[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) {
      sv[t] = afloatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < NJ; j++) sv[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kn[t][j], qf[j], sv[t], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (more) fetch_k(pos0 + kAttnKB);
    __builtin_amdgcn_sched_barrier(0);
    float x[4 * NT], mx = -INFINITY;
    if (pos0 + kAttnKB <= vis_wave) {  // wave-uniform: every row of this wave sees the whole block (all but the diagonal)
#pragma unroll
      for (int e = 0; e < 4 * NT; e++) {
        x[e] = sv[e >> 2][e & 3] * sc;
        mx = fmaxf(mx, x[e]);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4 * NT; e++) {
        const int pos = pos0 + 16 * (e >> 2) + 4 * g + (e & 3);
        x[e] = pos < visible ? sv[e >> 2][e & 3] * sc : -INFINITY;
        mx = fmaxf(mx, x[e]);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float m_use = m_new == -INFINITY ? 0.f : m_new;  // nothing visible yet: every exp2 below is exp2(-inf) = 0
    const float alpha = exp2f(m_run - m_use);
    float ps = 0.f;
    ahalf8_t pf[NH];  // slice h covers positions pos0 + 32 h ..: k-slot 8g + i <-> position 32 h + 16 (i >> 2) + 4g + (i & 3)
#pragma unroll
    for (int e = 0; e < 4 * NT; e++) {
      const float pe = exp2f(x[e] - m_use);
      ps += pe;
      pf[e >> 3][e & 7] = (_Float16)pe;
    }
    ps += __shfl_xor(ps, 16, 64);
    ps += __shfl_xor(ps, 32, 64);
    l_run = l_run * alpha + ps;
    m_run = m_new;
    float ar[4];
#pragma unroll
    for (int r = 0; r < 4; r++) ar[r] = __shfl(alpha, 4 * g + r, 64);

    __syncthreads();  // the previous block's V^T reads are done
    {
      // V^T[d][pos] for the block: (pos, pos + 1) pairs as 4-byte LDS stores; positions past the end hold zeros (P is zero
      // there; keeps 0 * x away from inf / nan bits)
      if (pos0 + kAttnKB <= p.sl_kv) {
#pragma unroll
        for (int i = 0; i < CH; i++)
          *reinterpret_cast<ahalf2_t*>(vt + (vc * CH + i) * kAttnVStr + vp) = ahalf2_t{vn[0][i >> 3][i & 7], vn[1][i >> 3][i & 7]};
      } else {
        const bool ok0 = pos0 + vp < p.sl_kv, ok1 = pos0 + vp + 1 < p.sl_kv;
#pragma unroll
        for (int i = 0; i < CH; i++) {
          const _Float16 a0 = ok0 ? vn[0][i >> 3][i & 7] : (_Float16)0.f, a1 = ok1 ? vn[1][i >> 3][i & 7] : (_Float16)0.f;
          *reinterpret_cast<ahalf2_t*>(vt + (vc * CH + i) * kAttnVStr + vp) = ahalf2_t{a0, a1};
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (more) fetch_v(pos0 + kAttnKB);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
#pragma unroll
    for (int dt = 0; dt < NDT; dt++) {
#pragma unroll
      for (int r = 0; r < 4; r++) o[dt][r] *= ar[r];
#pragma unroll
      for (int h = 0; h < NH; h++) {
        const _Float16* vr = vt + (16 * dt + nn) * kAttnVStr + 32 * h + 4 * g;
        const ahalf4_t lo = *reinterpret_cast<const ahalf4_t*>(vr), hi = *reinterpret_cast<const ahalf4_t*>(vr + 16);
        const ahalf8_t vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pf[h], vf, o[dt], 0, 0, 0);
      }
    }
  }
  const float inv = l_run > 0.f ? p.out_scale / l_run : 0.f;
  float ir[4];
#pragma unroll
  for (int r = 0; r < 4; r++) ir[r] = __shfl(inv, 4 * g + r, 64);
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int row = q0 + 4 * g + r;
    if (row >= p.sl_q) continue;
    float* dr = db + (long long)row * p.step_dst_sl + nn;
#pragma unroll
    for (int dt = 0; dt < NDT; dt++) {
      const float y = o[dt][r] * ir[r];
      dr[16 * dt] = y;
      if (p.dst16) p.dst16[dr - p.dst + 16 * dt] = (_Float16)y;
    }
  }
}

// ============================================================================================================
// attn_mfma2_kernel (round 4) — prefill on the matrix cores, second generation: 128 query rows per workgroup, 32 per wave,
// v_mfma_f32_32x32x16_f16, K and V tiles of 64 keys staged ONCE per workgroup in LDS (two stages: the next tile travels
// global -> registers while the current one is multiplied, registers -> LDS behind the P.V products, one barrier per tile).
//   S^T = K . Q^T   A = K rows from LDS (lane (key l & 31, half h = l >> 5): 16 bytes K[key][16 j + 8 h ..]; the 16-byte slots of a
//                   row are XOR-swizzled by the row so that the 16 lanes of a ds_read_b128 group hit 16 distinct slots),
//                   B = Q rows in registers (fp16, converted once).  Lane (query n = l & 31, h) receives the scores of ITS query at
//                   keys (i & 3) + 8 (i >> 2) + 4 h of the 32-key tile, i = 0..15: softmax needs one exchange (lane ^ 32) per tile
//                   for the maximum; row sums stay per lane until the end.
//   O^T += V^T . P^T  A = V^T, B = P^T: the 8 probabilities i = 8 s .. 8 s + 7 a lane holds ARE its B operand of 16-deep step s if
//                   k-slot 8 h + i' is read as key 16 s + (i' & 3) + 8 (i' >> 2) + 4 h — no data moves between lanes — and the
//                   accumulators of a lane all belong to its own query: the rescale is lane-local.  A in that enumeration = two runs
//                   of 4 consecutive keys at one head dim: ds_read_b64_tr_b16 (the CDNA4 transposing LDS read) from a ROW-major
//                   image, [16-dim subtile][key][16 dims] with the subtiles 128 B off a 256 B multiple (the two 16-lane halves of
//                   a read then cover disjoint banks).
// Causal balance: all workgroups of a prompt are resident at once (2 per CU at 2048 x 32 heads), so the launch order cannot
// balance; the first half of the grid takes its query blocks heaviest-first, the second half lightest-first: the two
// workgroups that meet on a CU sum to the same work.  fp16 operands, fp32 scores / statistics / accumulators.
// ============================================================================================================
typedef float afloatx16 __attribute__((ext_vector_type(16)));
typedef short ashort4_t __attribute__((ext_vector_type(4)));
#ifndef NS_A2_ABL
#define NS_A2_ABL 0  // timing ablations (diagnostic builds, wrong results): 1 no softmax, 2 no P.V, 3 no K.Q^T, 4 no tile traffic, 5 = 4 + no barrier
#endif
constexpr int kA2KB = 64;                  // keys per tile
constexpr int kA2VSub = kA2KB * 32 + 128;  // bytes per V subtile: [64 keys][16 dims] fp16 + the bank offset
template <int HS>
constexpr int a2_stage_bytes() { return kA2KB * HS * 2 + (HS / 16) * kA2VSub; }

// SB: the scores are biased before the softmax — ALiBi (+ key position x the head's slope, mha_dense_wrapper.h:1418-1447) and / or the
// 30 tanh(s / 30) soft cap — as attn_split_kernel applies them; a separate instantiation, the plain kernel's loop is unchanged.
// RG: 32-row groups per wave.  1 is what ships.  2 (256 query rows per workgroup: every K / V operand read from LDS feeds two MFMAs, the
// tile traffic is spread over twice the rows; all 512 registers, one wave per SIMD) was built, is correct (the whole attention suite and the
// shape fuzz pass on it) and is SLOWER: 4096 tokens 0.484 vs 0.268 ms, 8192 1.44 vs 0.85, 16384 4.89 vs 2.95 (profiles/r04bg_*) — two
// co-resident waves per SIMD hide more than halving the LDS traffic saves, and the 128-wide form spills 24 registers.
template <int HS, bool SB, bool PAD, int RG = 1>
__device__ __forceinline__ void attn_mfma2_body(const AttnParams& p, const int nqb, const int aligned_dst, const int xcd_map) {
  constexpr int NJ = HS / 16, NDT = HS / 32, NCH = HS / 8, KROW = HS * 2, NSUB = HS / 16;
  constexpr int KTILE = kA2KB * KROW, STAGE = a2_stage_bytes<HS>();
  constexpr int KU = NCH / 4;               // 16-byte K chunks per thread and tile
  // a wave-wide V load: lane = (half of a 32-byte subtile row, row of a group of VRG, subtile, group); 8 consecutive lanes write
  // 128 contiguous LDS bytes of one subtile (head size 256: two runs of 64 in neighbouring subtiles)
  constexpr int VRG = NSUB <= 8 ? 4 : 2, VRGL = NSUB <= 8 ? 2 : 1;
  constexpr int VRW = VRG * (32 / VRG / NSUB);  // V rows per wave-wide load
  constexpr int VU = kA2KB / (4 * VRW);         // V chunks per thread and tile
  extern __shared__ __attribute__((aligned(16))) unsigned char smem2[];
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, n = l & 31, h = l >> 5;
  const bool causal = (p.flags & NS_ATTN_FLAG_IS_CAUSAL) != 0;
  // workgroup -> (query block, head, batch).  Workgroups go to the XCDs round-robin (id % 8) and each XCD has its own L2: with the
  // plain order every XCD walks every head's K / V (32 MB at 2048 x 32 heads against 4 MB of L2 — the kernel then streams K / V from
  // memory at ~4 TB/s and is bound by that).  xcd_map: all query blocks of a kv head (and of the query heads sharing it) on ONE XCD,
  // head after head, so a head's K / V is fetched from memory once and re-read from that L2.
  // Causal balance: every (head, batch) unit walks its query blocks in ONE direction; the first half of the units (of the XCD's list)
  // heaviest-first, the second half lightest-first.
  const unsigned G = unsigned(p.head_num / p.heads_kv), units = gridDim.x / unsigned(nqb);
  unsigned unit = blockIdx.x / unsigned(nqb), ordn = blockIdx.x % unsigned(nqb);
  bool heavy_first = unit < (units + 1) / 2;
  if (xcd_map) {
    const unsigned x = blockIdx.x & 7u, sl = blockIdx.x >> 3;  // XCD, index inside the XCD's list
    const unsigned ul = sl / unsigned(nqb);                     // the XCD's ul-th (kv head, query head of its group)
    ordn = sl % unsigned(nqb);
    unit = ((ul / G) * 8u + x) * G + ul % G;
    heavy_first = ul < (units / 8u + 1) / 2;
  }
  const int hb = int(unit);
  const int qblk = causal && heavy_first ? nqb - 1 - int(ordn) : int(ordn);
  const int ihn = hb % p.head_num, ibs = hb / p.head_num;
  const int ihkv = ihn / (p.head_num / p.heads_kv);
  const int off = p.sl_kv - p.sl_q;
  constexpr int WGR = 128 * RG;  // query rows per workgroup
  const int q0 = qblk * WGR + w * (32 * RG);
  const float* qb = p.q + ibs * p.step_q_bs + ihn * p.step_q_head_num;
  const _Float16* kb = p.k + ibs * p.step_k_bs + ihkv * p.step_k_head_num;
  const _Float16* vb = p.v + ibs * p.step_v_bs + ihkv * p.step_v_head_num;
  float* db = p.dst + ibs * p.step_dst_bs + ihn * p.step_dst_head_num;

  // PAD: HS is the PADDED head size of this instantiation (64 / 128 / 256) and the call's own head size hs < HS a multiple of 8:
  // query dims, K / V chunks and output dims from hs on are zeros / never stored (head sizes 80, 96, 112, 160, 192 ... ride on the
  // next size up; a separate instantiation: the predicates cost the exact sizes a third of their speed)
  const int hs = PAD ? p.head_size : HS;
  ahalf8_t qf[RG][NJ];
  afloatx16 o[RG][NDT];
  float m_run[RG], l_run[RG];  // l_run: this lane's half of the row sum
#pragma unroll
  for (int rg = 0; rg < RG; rg++) {
    const float* qr = qb + (long long)min(q0 + 32 * rg + n, p.sl_q - 1) * p.step_q_sl + 8 * h;
#pragma unroll
    for (int j = 0; j < NJ; j++)
#pragma unroll
      for (int i = 0; i < 8; i++) qf[rg][j][i] = 16 * j + 8 * h < hs ? (_Float16)qr[16 * j + i] : (_Float16)0.f;
#pragma unroll
    for (int dt = 0; dt < NDT; dt++)
#pragma unroll
      for (int i = 0; i < 16; i++) o[rg][dt][i] = 0.f;
    m_run[rg] = -INFINITY, l_run[rg] = 0.f;
  }
  const float sc = SB ? 1.f : p.qk_scale * 1.4426950408889634f;  // scores in the exp2 domain (SB: the biased scores are scaled up front)
  float slope = 0.f;
  const bool tanh30 = SB && (p.flags & NS_ATTN_FLAG_IS_TANH30) != 0;
  if (SB && (p.flags & NS_ATTN_FLAG_IS_ALIBI8) != 0) {
    const int gh = ihn + p.alibi_head_off;
    slope = gh < p.alibi_log2_floor ? powf(p.alibi_m0, float(gh + 1)) : powf(p.alibi_m1, float(2 * (gh - p.alibi_log2_floor) + 1));
  }
  const int q_last_wg = min(qblk * WGR + WGR - 1, p.sl_q - 1);
  const int kv_end = causal ? min(p.sl_kv, q_last_wg + off + 1) : p.sl_kv;                     // workgroup-uniform
  int visible[RG];                                                                             // mha_dense_wrapper.h:1440-1441
#pragma unroll
  for (int rg = 0; rg < RG; rg++) visible[rg] = min(p.sl_kv, causal ? min(q0 + 32 * rg + n, p.sl_q - 1) + off + 1 : p.sl_kv);
  const int vis_first = min(p.sl_kv, causal ? q0 + off + 1 : p.sl_kv);                          // the wave's first row
  const int vis_last = min(p.sl_kv, causal ? min(q0 + 32 * RG - 1, p.sl_q - 1) + off + 1 : p.sl_kv);  // the wave's last row
  const bool wave_live = q0 < p.sl_q;

  // ---- tile staging: global -> registers -> LDS ----
  ahalf8_t kst[KU], vst[VU];
  int krow[KU], kslot[KU], vrow[VU];
#pragma unroll
  for (int u = 0; u < KU; u++) {
    const int id = tid + 256 * u;
    krow[u] = id / NCH;
    const int c = id % NCH;
    const int swz = HS >= 128 ? (krow[u] & 15) : ((krow[u] >> 1) & 7);
    kslot[u] = krow[u] * KROW + ((c ^ swz) << 4);
  }
  const int vsub = (l >> (1 + VRGL)) % NSUB, vrq = (l >> (1 + VRGL)) / NSUB, vr = (l >> 1) & (VRG - 1), vhf = l & 1;
#pragma unroll
  for (int u = 0; u < VU; u++) vrow[u] = u * (4 * VRW) + w * VRW + vrq * VRG + vr;
  const int kcol = (tid % NCH) * 8, vcol = 16 * vsub + 8 * vhf;
  // row pointers advance by a uniform stride per tile; only the tile that reaches past the last key clamps its rows
  const _Float16* kp0 = kb + (long long)krow[0] * p.step_k_sl + kcol;
  const _Float16* vp0 = vb + (long long)vrow[0] * p.step_v_sl + vcol;
  const long long ku_step = (long long)(256 / NCH) * p.step_k_sl, vu_step = (long long)(4 * VRW) * p.step_v_sl;
  const long long ktile_step = (long long)kA2KB * p.step_k_sl, vtile_step = (long long)kA2KB * p.step_v_sl;
  const bool k_in = !PAD || kcol < hs, v_in = !PAD || vcol < hs;  // this thread's 16-byte column of the rows exists (else: zeros, once)
  if (!k_in) {
#pragma unroll
    for (int u = 0; u < KU; u++) kst[u] = ahalf8_t{0, 0, 0, 0, 0, 0, 0, 0};
  }
  if (!v_in) {
#pragma unroll
    for (int u = 0; u < VU; u++) vst[u] = ahalf8_t{0, 0, 0, 0, 0, 0, 0, 0};
  }
  auto fetch = [&](int pos0) {
    if (pos0 + kA2KB <= p.sl_kv) {  // workgroup-uniform
      if (k_in) {
#pragma unroll
        for (int u = 0; u < KU; u++) kst[u] = *reinterpret_cast<const ahalf8_t*>(kp0 + u * ku_step);
      }
      if (v_in) {
#pragma unroll
        for (int u = 0; u < VU; u++) vst[u] = *reinterpret_cast<const ahalf8_t*>(vp0 + u * vu_step);
      }
    } else {
      if (k_in) {
#pragma unroll
        for (int u = 0; u < KU; u++)
          kst[u] = *reinterpret_cast<const ahalf8_t*>(kb + (long long)min(pos0 + krow[u], p.sl_kv - 1) * p.step_k_sl + kcol);
      }
      if (v_in) {
#pragma unroll
        for (int u = 0; u < VU; u++)
          vst[u] = *reinterpret_cast<const ahalf8_t*>(vb + (long long)min(pos0 + vrow[u], p.sl_kv - 1) * p.step_v_sl + vcol);
      }
    }
    kp0 += ktile_step;
    vp0 += vtile_step;
  };
  auto park = [&](int stage) {
    unsigned char* sb = smem2 + stage * STAGE;
#pragma unroll
    for (int u = 0; u < KU; u++) *reinterpret_cast<ahalf8_t*>(sb + kslot[u]) = kst[u];
#pragma unroll
    for (int u = 0; u < VU; u++) *reinterpret_cast<ahalf8_t*>(sb + KTILE + vsub * kA2VSub + vrow[u] * 32 + vhf * 16) = vst[u];
  };
  // operand addresses inside a stage
  const int swz_n = HS >= 128 ? (n & 15) : ((n >> 1) & 7);
  const int k_rd = n * KROW;                                                        // + 32 T rows, slot (2 j + h) ^ swizzle
  const int v_rd = KTILE + ((l >> 4) & 1) * kA2VSub + (4 * h + ((l & 15) >> 2)) * 32 + (l & 3) * 8;  // + 2 dt subtiles, + (32 T + 16 s) rows, + 8 rows

  const int nb = (kv_end + kA2KB - 1) / kA2KB;
  if (nb > 0) {
    fetch(0);
    park(0);
  }
  __syncthreads();
  for (int b = 0; b < nb; b++) {
    const int pos0 = b * kA2KB;
    const unsigned char* sb = smem2 + (b & 1) * STAGE;
    if (b + 1 < nb && NS_A2_ABL < 4) fetch(pos0 + kA2KB);
    if (wave_live && pos0 < vis_last) {
      // operands travel LDS -> registers one group of four MFMAs ahead of their use (two register sets, fenced: left alone hipcc
      // re-uses ONE operand register and serialises read -> wait -> MFMA, ~100 exposed cycles per MFMA)
      afloatx16 sv[RG][2];
#pragma unroll
      for (int rg = 0; rg < RG; rg++)
#pragma unroll
        for (int T = 0; T < 2; T++)
#pragma unroll
          for (int i = 0; i < 16; i++) sv[rg][T][i] = 0.f;
      {
        constexpr int CPT = NJ / 4, NCK = 2 * CPT;  // groups per 32-key tile, groups per block
        ahalf8_t kf[2][4];
        auto ldk = [&](int c, ahalf8_t(&dst)[4]) {
          const int T = c / CPT, j0 = (c % CPT) * 4;
#pragma unroll
          for (int u = 0; u < 4; u++)
            dst[u] = *reinterpret_cast<const ahalf8_t*>(sb + k_rd + 32 * T * KROW + (((2 * (j0 + u) + h) ^ swz_n) << 4));
        };
        if (NS_A2_ABL != 3) ldk(0, kf[0]);
#pragma unroll
        for (int c = 0; c < (NS_A2_ABL == 3 ? 0 : NCK); c++) {
          if (c + 1 < NCK) ldk(c + 1, kf[(c + 1) & 1]);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < 4; u++)
#pragma unroll
            for (int rg = 0; rg < RG; rg++)
              sv[rg][c / CPT] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[c & 1][u], qf[rg][(c % CPT) * 4 + u], sv[rg][c / CPT], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // softmax in the exp2 domain on the raw scores (sc > 0: the launcher sends other scales to the 64-row kernel): p = exp2(s sc - m sc)
      // is one fused multiply-add and one v_exp_f32 per score; the running maximum is kept unscaled
      ahalf8_t pf[RG][4];  // (T, s) -> 2 T + s
#pragma unroll
      for (int rg = 0; rg < RG; rg++) {
#if NS_A2_ABL == 1
#pragma unroll
      for (int e = 0; e < 32; e++) pf[rg][e >> 3][e & 7] = (_Float16)sv[rg][e >> 4][e & 15];
      l_run[rg] += 1.f;
#else
      if constexpr (SB) {
#pragma unroll
        for (int e = 0; e < 32; e++) {
          const int i = e & 15;
          const int pos = pos0 + 32 * (e >> 4) + (i & 3) + 8 * (i >> 2) + 4 * h;
          float v = sv[rg][e >> 4][i] * p.qk_scale;
          if (tanh30) v = 30.f * tanhf(v * (1.f / 30.f));
          v += float(pos) * slope;
          sv[rg][e >> 4][i] = v * 1.4426950408889634f;
        }
      }
      if (pos0 + kA2KB > vis_first) {  // wave-uniform: only tiles on the diagonal / past the last key are masked
#pragma unroll
        for (int e = 0; e < 32; e++) {
          const int i = e & 15;
          const int pos = pos0 + 32 * (e >> 4) + (i & 3) + 8 * (i >> 2) + 4 * h;
          if (pos >= visible[rg]) sv[rg][e >> 4][i] = -INFINITY;
        }
      }
      float mx = sv[rg][0][0];
#pragma unroll
      for (int e = 1; e < 32; e++) mx = fmaxf(mx, sv[rg][e >> 4][e & 15]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run[rg], mx);
      const float neg_m = m_new == -INFINITY ? 0.f : -m_new * sc;  // nothing visible yet: every exp2 below is exp2(-inf) = 0
      const float alpha = __builtin_amdgcn_exp2f(fmaf(m_run[rg], sc, neg_m));
      float ps = 0.f;
#pragma unroll
      for (int e = 0; e < 32; e++) {
        const float pe = __builtin_amdgcn_exp2f(fmaf(sv[rg][e >> 4][e & 15], sc, neg_m));
        ps += pe;
        pf[rg][e >> 3][e & 7] = (_Float16)pe;
      }
      l_run[rg] = l_run[rg] * alpha + ps;
      m_run[rg] = m_new;
      if (__any(alpha != 1.f)) {
#pragma unroll
        for (int dt = 0; dt < NDT; dt++)
#pragma unroll
          for (int i = 0; i < 16; i++) o[rg][dt][i] *= alpha;
      }
#endif
      }
      {
        ahalf8_t vf[2][4];
        auto ldv = [&](int dt, ahalf8_t(&dst)[4]) {
#pragma unroll
          for (int ts = 0; ts < 4; ts++) {
            const unsigned char* va = sb + v_rd + 2 * dt * kA2VSub + 16 * ts * 32;
            const ashort4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ashort4_t __attribute__((address_space(3)))*)(va));
            const ashort4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ashort4_t __attribute__((address_space(3)))*)(va + 8 * 32));
            dst[ts] = __builtin_bit_cast(ahalf8_t, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
          }
        };
        if (NS_A2_ABL != 2) ldv(0, vf[0]);
#pragma unroll
        for (int dt = 0; dt < (NS_A2_ABL == 2 ? 0 : NDT); dt++) {
          if (dt + 1 < NDT) ldv(dt + 1, vf[(dt + 1) & 1]);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int ts = 0; ts < 4; ts++)
#pragma unroll
            for (int rg = 0; rg < RG; rg++) o[rg][dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[dt & 1][ts], pf[rg][ts], o[rg][dt], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if (b + 1 < nb && NS_A2_ABL < 4) park((b + 1) & 1);
    if (NS_A2_ABL < 5) __syncthreads();
  }
#pragma unroll
  for (int rg = 0; rg < RG; rg++) {
  l_run[rg] += __shfl_xor(l_run[rg], 32, 64);
  const float inv = l_run[rg] > 0.f ? p.out_scale / l_run[rg] : 0.f;
  const int row = q0 + 32 * rg + n;
  if (row < p.sl_q) {
    float* dr = db + (long long)row * p.step_dst_sl;
#pragma unroll
    for (int dt = 0; dt < NDT; dt++) {
#pragma unroll
      for (int bq = 0; bq < 4; bq++) {
        const int d = 32 * dt + 8 * bq + 4 * h;
        if (PAD && d >= hs) continue;
        const afloatx4 y = afloatx4{o[rg][dt][4 * bq] * inv, o[rg][dt][4 * bq + 1] * inv, o[rg][dt][4 * bq + 2] * inv, o[rg][dt][4 * bq + 3] * inv};
        if (aligned_dst) {
          *reinterpret_cast<afloatx4*>(dr + d) = y;
          if (p.dst16) *reinterpret_cast<ahalf4_t*>(p.dst16 + (dr - p.dst) + d) = ahalf4_t{(_Float16)y[0], (_Float16)y[1], (_Float16)y[2], (_Float16)y[3]};
        } else {
#pragma unroll
          for (int r = 0; r < 4; r++) {
            dr[d + r] = y[r];
            if (p.dst16) p.dst16[(dr - p.dst) + d + r] = (_Float16)y[r];
          }
        }
      }
    }
  }
  }
}

// ============================================================================================================
// attn_mfma3_kernel (round 6) — the operands, layouts and arithmetic of attn_mfma2_kernel (exact head sizes 64 / 128, no score bias) with the K / V tiles
// travelling HBM -> LDS by DMA (buffer_load ... lds, 1 KiB per wave-wide request): the XOR swizzle of a K row's 16-byte slots and the V subtile image are
// made by WHICH global 16 bytes a lane asks for (the DMA writes lane-linear), so there are no staging registers and no register -> LDS pass ("park") —
// 8 requests per wave and tile instead of 8 loads + 8 ds_write_b128 + their address arithmetic, and 32 registers free.  Rows past the last key lie outside
// the buffer descriptor and arrive as zeros (their scores are masked).  2048 tokens 500 -> 530 TFLOPS causal, 4096 545 -> 605, 8192 667 -> 750 (standalone,
// profiles/r06_attn_prefill_dma.txt); a layer's launch inside the 2048-token prompt -3 us.
// Built on top of it, correct (the whole attention suite passes) and NOT adopted: P.V lagging one tile so that its MFMAs issue between the exponentials of the
// next tile (sched_group_barrier pattern: one MFMA, two transposing LDS reads, two v_exp, ten VALU): the second set of probabilities pushes the 128-wide
// instantiation to 256 registers + 21 spilled, each MFMA waits for the LDS reads placed right in front of it — 410-439 TFLOPS at 2048 tokens against 530.
// Requesting the first V operands in front of the exponentials: also slower (504 vs 540).  What bounds 2048 tokens now is the pairing: a CU's two workgroups
// (heavy + light q-block: equal SUM of tiles) run together for the light one's tiles only, the heavy one then has its SIMDs to itself — 1.9 us per
// tile-unit against 1.41 at 8192 tokens, where eight workgroups per CU even that out.
template <int HS>
__device__ __forceinline__ void attn_mfma3_body(const AttnParams& p, const int nqb, const int aligned_dst, const int xcd_map) {
  constexpr int NJ = HS / 16, NDT = HS / 32, NCH = HS / 8, KROW = HS * 2, NSUB = HS / 16;
  constexpr int KTILE = kA2KB * KROW, VTILE = NSUB * kA2VSub;
  constexpr int KREQ = KTILE / 1024 / 4, VREQ = 2 * NSUB / 4;  // DMA requests per wave and tile
  extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];  // K[2][KTILE] | V[2][VTILE]
  typedef __attribute__((address_space(3))) unsigned char* LdsPtr;
  const int tid = threadIdx.x, l = tid & 63, n = l & 31, h = l >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool causal = (p.flags & NS_ATTN_FLAG_IS_CAUSAL) != 0;
  const unsigned G = unsigned(p.head_num / p.heads_kv), units = gridDim.x / unsigned(nqb);
  unsigned unit = blockIdx.x / unsigned(nqb), ordn = blockIdx.x % unsigned(nqb);
  bool heavy_first = unit < (units + 1) / 2;
  if (xcd_map & 1) {
    const unsigned x = blockIdx.x & 7u, sl = blockIdx.x >> 3;
    const unsigned ul = sl / unsigned(nqb);
    ordn = sl % unsigned(nqb);
    unit = ((ul / G) * 8u + x) * G + ul % G;
    heavy_first = ul < (units / 8u + 1) / 2;
  }
  const int hb = int(unit);
  const int qblk = causal && heavy_first ? nqb - 1 - int(ordn) : int(ordn);
  const int ihn = hb % p.head_num, ibs = hb / p.head_num;
  const int ihkv = ihn / (p.head_num / p.heads_kv);
  const int off = p.sl_kv - p.sl_q;
  const int q0 = qblk * 128 + w * 32;
  const float* qb = p.q + ibs * p.step_q_bs + ihn * p.step_q_head_num;
  const _Float16* kb = p.k + ibs * p.step_k_bs + ihkv * p.step_k_head_num;
  const _Float16* vb = p.v + ibs * p.step_v_bs + ihkv * p.step_v_head_num;
  float* db = p.dst + ibs * p.step_dst_bs + ihn * p.step_dst_head_num;

  ahalf8_t qf[NJ];
  afloatx16 o[NDT];
  float m_run = -INFINITY, l_run = 0.f;
  {
    const float* qr = qb + (long long)min(q0 + n, p.sl_q - 1) * p.step_q_sl + 8 * h;
#pragma unroll
    for (int j = 0; j < NJ; j++)
#pragma unroll
      for (int i = 0; i < 8; i++) qf[j][i] = (_Float16)qr[16 * j + i];
#pragma unroll
    for (int dt = 0; dt < NDT; dt++)
#pragma unroll
      for (int i = 0; i < 16; i++) o[dt][i] = 0.f;
  }
  const float sc = p.qk_scale * 1.4426950408889634f;
  const int q_last_wg = min(qblk * 128 + 127, p.sl_q - 1);
  const int kv_end = causal ? min(p.sl_kv, q_last_wg + off + 1) : p.sl_kv;
  const int visible = min(p.sl_kv, causal ? min(q0 + n, p.sl_q - 1) + off + 1 : p.sl_kv);
  const int vis_first = min(p.sl_kv, causal ? q0 + off + 1 : p.sl_kv);
  const int vis_last = min(p.sl_kv, causal ? min(q0 + 31, p.sl_q - 1) + off + 1 : p.sl_kv);
  const bool wave_live = q0 < p.sl_q;

  // ---- tile requests (DMA): descriptors over the head's rows; a row past the last key lies outside them and arrives as zeros (its scores are masked) ----
  auto uniform_ptr = [](const _Float16* ptr) {
    const uint64_t v = reinterpret_cast<uint64_t>(ptr);
    const uint32_t lo = __builtin_amdgcn_readfirstlane(uint32_t(v)), hi = __builtin_amdgcn_readfirstlane(uint32_t(v >> 32));
    return reinterpret_cast<_Float16*>((uint64_t(hi) << 32) | lo);
  };
  const uint32_t k_bytes = uint32_t((size_t(p.sl_kv - 1) * p.step_k_sl + HS) * 2), v_bytes = uint32_t((size_t(p.sl_kv - 1) * p.step_v_sl + HS) * 2);
  const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(kb), 0, __builtin_amdgcn_readfirstlane(k_bytes), 0x00020000);
  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(vb), 0, __builtin_amdgcn_readfirstlane(v_bytes), 0x00020000);
  uint32_t kvoff[KREQ], vvoff[VREQ];  // byte offsets inside tile 0 of what this lane asks for in its wave's u-th request
  constexpr int RPC = 1024 / KROW;    // K rows per 1 KiB request
#pragma unroll
  for (int u = 0; u < KREQ; u++) {
    const int r = (w * KREQ + u) * RPC + l / NCH, slot = l % NCH;
    const int swz = HS >= 128 ? (r & 15) : ((r >> 1) & 7);
    kvoff[u] = (uint32_t(r) * uint32_t(p.step_k_sl) + uint32_t((slot ^ swz) * 8)) * 2u;
  }
#pragma unroll
  for (int u = 0; u < VREQ; u++) {
    const int c = w * VREQ + u, sub = c >> 1, half = c & 1;  // request c = (subtile, rows 0..31 / 32..63)
    vvoff[u] = (uint32_t(half * 32 + (l >> 1)) * uint32_t(p.step_v_sl) + uint32_t(16 * sub + 8 * (l & 1))) * 2u;
  }
  const uint32_t ktile_step = uint32_t(kA2KB) * uint32_t(p.step_k_sl) * 2u, vtile_step = uint32_t(kA2KB) * uint32_t(p.step_v_sl) * 2u;
  auto issue_k = [&](int t, int buf) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int u = 0; u < KREQ; u++)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, reinterpret_cast<__attribute__((address_space(3))) void*>((LdsPtr)(smem3) + buf * KTILE + (w * KREQ + u) * 1024), 16,
                                               kvoff[u] + uint32_t(t) * ktile_step, 0, 0, 0);
#endif
  };
  auto issue_v = [&](int t, int buf) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int u = 0; u < VREQ; u++) {
      const int c = w * VREQ + u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, reinterpret_cast<__attribute__((address_space(3))) void*>((LdsPtr)(smem3) + 2 * KTILE + buf * VTILE + (c >> 1) * kA2VSub + (c & 1) * 1024),
                                               16, vvoff[u] + uint32_t(t) * vtile_step, 0, 0, 0);
    }
#endif
  };
  const int swz_n = HS >= 128 ? (n & 15) : ((n >> 1) & 7);
  const int k_rd = n * KROW;
  const int v_rd = ((l >> 4) & 1) * kA2VSub + (4 * h + ((l & 15) >> 2)) * 32 + (l & 3) * 8;

  // K(b) . Q^T -> sv (two 32-key halves), operands one group of four MFMAs ahead of their use
  auto qk = [&](const unsigned char* kt, afloatx16 (&sv)[2]) {
    constexpr int CPT = NJ / 4, NCK = 2 * CPT;
#pragma unroll
    for (int T = 0; T < 2; T++)
#pragma unroll
      for (int i = 0; i < 16; i++) sv[T][i] = 0.f;
    ahalf8_t kf[2][4];
    auto ldk = [&](int c, ahalf8_t(&dst)[4]) {
      const int T = c / CPT, j0 = (c % CPT) * 4;
#pragma unroll
      for (int u = 0; u < 4; u++) dst[u] = *reinterpret_cast<const ahalf8_t*>(kt + k_rd + 32 * T * KROW + (((2 * (j0 + u) + h) ^ swz_n) << 4));
    };
    ldk(0, kf[0]);
#pragma unroll
    for (int c = 0; c < NCK; c++) {
      if (c + 1 < NCK) ldk(c + 1, kf[(c + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 4; u++) sv[c / CPT] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[c & 1][u], qf[(c % CPT) * 4 + u], sv[c / CPT], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // the exponentials of a tile: scores -> probabilities (fp16, the B operand of P.V), running maximum / sum; returns the factor the accumulators take
  auto softmax = [&](afloatx16 (&sv)[2], ahalf8_t (&pf)[4], int pos0, bool masked) -> float {
    if (masked) {
#pragma unroll
      for (int e = 0; e < 32; e++) {
        const int i = e & 15;
        const int pos = pos0 + 32 * (e >> 4) + (i & 3) + 8 * (i >> 2) + 4 * h;
        if (pos >= visible) sv[e >> 4][i] = -INFINITY;
      }
    }
    float mx = sv[0][0];
#pragma unroll
    for (int e = 1; e < 32; e++) mx = fmaxf(mx, sv[e >> 4][e & 15]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float neg_m = m_new == -INFINITY ? 0.f : -m_new * sc;
    const float alpha = __builtin_amdgcn_exp2f(fmaf(m_run, sc, neg_m));
    float ps = 0.f;
#pragma unroll
    for (int e = 0; e < 32; e++) {
      const float pe = __builtin_amdgcn_exp2f(fmaf(sv[e >> 4][e & 15], sc, neg_m));
      ps += pe;
      pf[e >> 3][e & 7] = (_Float16)pe;
    }
    l_run = l_run * alpha + ps;
    m_run = m_new;
    return alpha;
  };
  // O^T += V(t)^T . P^T, operands one group of four MFMAs ahead of their use
  auto pv = [&](const unsigned char* vt, const ahalf8_t (&pf)[4]) {
    auto ldv = [&](int dt, ahalf8_t(&dst)[4]) {
#pragma unroll
      for (int ts = 0; ts < 4; ts++) {
        const unsigned char* va = vt + v_rd + 2 * dt * kA2VSub + 16 * ts * 32;
        const ashort4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ashort4_t __attribute__((address_space(3)))*)(va));
        const ashort4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ashort4_t __attribute__((address_space(3)))*)(va + 8 * 32));
        dst[ts] = __builtin_bit_cast(ahalf8_t, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
      }
    };
    ahalf8_t vf[2][4];
    ldv(0, vf[0]);
#pragma unroll
    for (int dt = 0; dt < NDT; dt++) {
      if (dt + 1 < NDT) ldv(dt + 1, vf[(dt + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ts = 0; ts < 4; ts++) o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[dt & 1][ts], pf[ts], o[dt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto rescale = [&](float alpha) {
    if (__any(alpha != 1.f)) {
#pragma unroll
      for (int dt = 0; dt < NDT; dt++)
#pragma unroll
        for (int i = 0; i < 16; i++) o[dt][i] *= alpha;
    }
  };

  const int nb = (kv_end + kA2KB - 1) / kA2KB;
  const int variant = xcd_map >> 8;  // bit 0: the next tile is requested behind K.Q^T instead of in front of it; bit 1: raised priority while K.Q^T issues
  if (nb > 0) issue_k(0, 0), issue_v(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int b = 0; b < nb; b++) {
    const int pos0 = b * kA2KB;
    const bool more = b + 1 < nb;
    if (!(variant & 1) && more) issue_k(b + 1, (b + 1) & 1), issue_v(b + 1, (b + 1) & 1);
    if (wave_live && pos0 < vis_last) {
      afloatx16 sv[2];
      ahalf8_t pf[4];
      if (variant & 2) __builtin_amdgcn_s_setprio(2);
      qk(smem3 + (b & 1) * KTILE, sv);
      if (variant & 2) __builtin_amdgcn_s_setprio(0);
      if ((variant & 1) && more) issue_k(b + 1, (b + 1) & 1), issue_v(b + 1, (b + 1) & 1);
      const float alpha = softmax(sv, pf, pos0, pos0 + kA2KB > vis_first);  // (masked: wave-uniform — only tiles on the diagonal / past the last key)
      rescale(alpha);
      pv(smem3 + 2 * KTILE + (b & 1) * VTILE, pf);
    } else if ((variant & 1) && more) {
      issue_k(b + 1, (b + 1) & 1), issue_v(b + 1, (b + 1) & 1);
    }
    // this wave's requests have landed; every wave is done with tile b's buffers
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  l_run += __shfl_xor(l_run, 32, 64);
  const float inv = l_run > 0.f ? p.out_scale / l_run : 0.f;
  const int row = q0 + n;
  if (row < p.sl_q) {
    float* dr = db + (long long)row * p.step_dst_sl;
#pragma unroll
    for (int dt = 0; dt < NDT; dt++) {
#pragma unroll
      for (int bq = 0; bq < 4; bq++) {
        const int d = 32 * dt + 8 * bq + 4 * h;
        const afloatx4 y = afloatx4{o[dt][4 * bq] * inv, o[dt][4 * bq + 1] * inv, o[dt][4 * bq + 2] * inv, o[dt][4 * bq + 3] * inv};
        if (aligned_dst) {
          *reinterpret_cast<afloatx4*>(dr + d) = y;
          if (p.dst16) *reinterpret_cast<ahalf4_t*>(p.dst16 + (dr - p.dst) + d) = ahalf4_t{(_Float16)y[0], (_Float16)y[1], (_Float16)y[2], (_Float16)y[3]};
        } else {
#pragma unroll
          for (int r = 0; r < 4; r++) {
            dr[d + r] = y[r];
            if (p.dst16) p.dst16[(dr - p.dst) + d + r] = (_Float16)y[r];
          }
        }
      }
    }
  }
}
template <int HS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_mfma3_kernel(const AttnParams p, const int nqb, const int aligned_dst, const int xcd_map) {
  attn_mfma3_body<HS>(p, nqb, aligned_dst, xcd_map);
}

template <int HS, bool SB = false, bool PAD = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_mfma2_kernel(const AttnParams p, const int nqb, const int aligned_dst, const int xcd_map) {
  attn_mfma2_body<HS, SB, PAD>(p, nqb, aligned_dst, xcd_map);
}
// head size 256 (GPT-J, Gemma): 128 accumulator + 64 query registers per lane -> one wave per SIMD with the whole register file,
// one workgroup per CU (two stages of 66 KB)
template <bool SB = false, bool PAD = false>
__global__ __launch_bounds__(256) void attn_mfma2_hs256_kernel(const AttnParams p, const int nqb, const int aligned_dst, const int xcd_map) {
  attn_mfma2_body<256, SB, PAD>(p, nqb, aligned_dst, xcd_map);
}
static std::atomic<int> g_alibi_heads{0}, g_alibi_off{0};  // ns_hip_attn_set_head_partition

static hipError_t launch_attn(const attn_fp32_fp16_fp16_fp32_fwd_args_t& a, hipStream_t st, std::string* why,
                              bool device_tmp, void* dst16 = nullptr) {
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
  p.dst16 = static_cast<_Float16*>(dst16);
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
  // mha_dense_wrapper.h:1418-1447: under tensor parallelism the slopes follow the FULL model's head count and this
  // rank's first head (the reference takes world/rank from parallel_context; here ns_hip_attn_set_head_partition)
  const int all_heads = g_alibi_heads.load() > 0 ? g_alibi_heads.load() : a.head_num;
  p.alibi_head_off = g_alibi_heads.load() > 0 ? g_alibi_off.load() : 0;
  if (p.alibi_head_off < 0 || p.alibi_head_off + a.head_num > all_heads) {
    *why = "attention: head partition (ns_hip_attn_set_head_partition) does not contain this call's heads";
    return hipErrorInvalidValue;
  }
  const int lf = 1 << int(floor(log2(double(all_heads))));
  p.alibi_log2_floor = lf;
  p.alibi_m0 = powf(2.0f, -8.f / float(lf));
  p.alibi_m1 = powf(2.0f, -4.f / float(lf));
  if (a.head_num > 65535 || a.batch_size > 65535) {
    *why = "attention: head_num / batch_size above the grid limit";
    return hipErrorInvalidValue;
  }
  // replayed device route (ns_route.cpp): the context length of a captured decode step moves with the graph's token counter
  const Affine aff = g_affine;
  if (aff.k && (a.sl_q != 1 || aff.cap < a.sl_kv)) {
    *why = "attention: a moving context length is served for decode steps (one query row) within the cache's capacity";
    return hipErrorInvalidValue;
  }
  // ---- several query rows: matrix cores (head size 64 / 128, contiguous 16-byte aligned rows, no alibi / tanh) ----
  static const bool no_mfma = getenv("NS_ATTN_NO_MFMA") != nullptr;  // diagnostics
  const bool rows_ok = a.step_k_head_size == 1 && a.step_v_head_size == 1 && a.step_k_sl % 8 == 0 && a.step_v_sl % 8 == 0 &&
                       a.step_k_head_num % 8 == 0 && a.step_v_head_num % 8 == 0 && a.step_k_bs % 8 == 0 &&
                       a.step_v_bs % 8 == 0 && (reinterpret_cast<uintptr_t>(a.K) & 15) == 0 &&
                       (reinterpret_cast<uintptr_t>(a.V) & 15) == 0;
  const bool biased = (a.attn_flags & (NS_ATTN_FLAG_IS_ALIBI8 | NS_ATTN_FLAG_IS_TANH30)) != 0;
  const size_t nqb = (size_t(a.sl_q) + 127) / 128, wgs2 = nqb * a.head_num * a.batch_size;
  // shapes the 64-row kernel cannot take (biased scores, head sizes other than 64 / 128) use the 128-row kernel from 16 rows on: a
  // workgroup with idle waves is still far from the one-row-per-workgroup kernel they would fall to
  const bool only128 = biased || (a.head_size != 64 && a.head_size != 128);
  const bool rows128 = a.sl_q >= (only128 ? 16 : g_attn_mfma2_rows.load(std::memory_order_relaxed)) && wgs2 < (size_t(1) << 31) &&
                       (biased || p.qk_scale > 0.f);
  // the 128-row kernel pads any head size that is a multiple of 8 to 64 / 128 / 256; the 64-row kernel takes 64 and 128 as they are
  const bool exact = a.head_size == 64 || a.head_size == 128, hs256 = a.head_size > 128;
  if (!no_mfma && a.sl_q >= 16 && a.head_size % 8 == 0 && a.head_size <= 256 && rows_ok && ((!biased && exact) || rows128)) {  // (the 64-row kernel has no biased form)
    if (rows128) {
      // 128-row workgroups, 32x32x16 MFMA, K / V tiles shared through LDS (attn_mfma2_kernel)
      const int aligned = (reinterpret_cast<uintptr_t>(a.dst) & 15) == 0 && a.step_dst_sl % 4 == 0 && a.step_dst_head_num % 4 == 0 &&
                          a.step_dst_bs % 4 == 0 && (!dst16 || (reinterpret_cast<uintptr_t>(dst16) & 7) == 0);
      static const bool no_xcd = getenv("NS_ATTN_NO_XCD_MAP") != nullptr;  // diagnostics (A/B)
      const int xcd_map = !no_xcd && (size_t(a.heads_kv) * a.batch_size) % 8 == 0;
      auto go = [&](auto kern, int stage, int extra = 0) {
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * stage);
        if (attr != hipSuccess) return attr;
        hipLaunchKernelGGL(kern, dim3(unsigned(wgs2)), dim3(256), size_t(2) * stage, st, p, int(nqb), aligned, xcd_map | extra);
        return hipGetLastError();
      };
      const bool pad = a.head_size != 64 && a.head_size != 128 && a.head_size != 256;
      // round 6: the pipelined schedule (attn_mfma3_kernel) for the exact head sizes without score bias; rows inside a 32-bit buffer descriptor
      // round 6: K / V tiles by DMA (attn_mfma3_kernel) for the exact head sizes without score bias, rows inside a 32-bit buffer descriptor; NS_ATTN_PIPE=0:
      // attn_mfma2_kernel.  Variant (NS_ATTN_PVAR): bit 0 = the next tile is requested behind K.Q^T instead of in front of it, bit 1 = raised priority while
      // K.Q^T issues — both gain from 4096 keys on (601-611 vs 558-585 TFLOPS causal, 742-751 vs 722 at 8192) and lose below (519-525 vs 529-530 at 2048)
      static const int pipe = getenv("NS_ATTN_PIPE") ? atoi(getenv("NS_ATTN_PIPE")) : 1;
      static const int pvar_env = getenv("NS_ATTN_PVAR") ? atoi(getenv("NS_ATTN_PVAR")) : -1;
      const int pvar = pvar_env >= 0 ? pvar_env : (a.sl_kv >= 3072 ? 3 : 0);
      const bool in32 = (size_t(a.sl_kv) * size_t(a.step_k_sl) + 256) * 2 < (size_t(1) << 32) && (size_t(a.sl_kv) * size_t(a.step_v_sl) + 256) * 2 < (size_t(1) << 32);
      if (pipe && !biased && !pad && !hs256 && in32)
        return a.head_size == 64 ? go(attn_mfma3_kernel<64>, a2_stage_bytes<64>(), pvar << 8) : go(attn_mfma3_kernel<128>, a2_stage_bytes<128>(), pvar << 8);
      if (hs256)
        return biased ? (pad ? go(attn_mfma2_hs256_kernel<true, true>, a2_stage_bytes<256>()) : go(attn_mfma2_hs256_kernel<true, false>, a2_stage_bytes<256>()))
                      : (pad ? go(attn_mfma2_hs256_kernel<false, true>, a2_stage_bytes<256>()) : go(attn_mfma2_hs256_kernel<false, false>, a2_stage_bytes<256>()));
      auto pick = [&](auto hs_c) {
        constexpr int H = decltype(hs_c)::value;
        if (biased) return pad ? go(attn_mfma2_kernel<H, true, true>, a2_stage_bytes<H>()) : go(attn_mfma2_kernel<H, true, false>, a2_stage_bytes<H>());
        return pad ? go(attn_mfma2_kernel<H, false, true>, a2_stage_bytes<H>()) : go(attn_mfma2_kernel<H, false, false>, a2_stage_bytes<H>());
      };
      return a.head_size <= 64 ? pick(std::integral_constant<int, 64>{}) : pick(std::integral_constant<int, 128>{});
    }
    const dim3 grid(unsigned((a.sl_q + 63) / 64), unsigned(a.head_num), unsigned(a.batch_size));
    if (a.head_size == 64)
      hipLaunchKernelGGL(attn_mfma_kernel<64>, grid, dim3(256), 0, st, p);
    else
      hipLaunchKernelGGL(attn_mfma_kernel<128>, grid, dim3(256), 0, st, p);
    return hipGetLastError();
  }
  // ---- fast path: contiguous head dimension, 16-byte aligned rows, head group 1/2/4/8 ----
  static const bool no_split = getenv("NS_ATTN_V1") != nullptr;  // diagnostics
  int G, chunks;
  attn_groups(a.head_num / a.heads_kv, &G, &chunks);
  const int dpl = a.head_size <= 128 ? 8 : 16;
  const bool fast = !no_split && a.step_k_head_size == 1 &&
                    a.step_v_head_size == 1 && a.head_size % dpl == 0 && a.step_k_sl % 8 == 0 && a.step_v_sl % 8 == 0 &&
                    a.step_k_head_num % 8 == 0 && a.step_v_head_num % 8 == 0 && a.step_k_bs % 8 == 0 &&
                    a.step_v_bs % 8 == 0 && (reinterpret_cast<uintptr_t>(a.K) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(a.V) & 15) == 0;
  if (fast && size_t(a.heads_kv) * a.sl_q * chunks <= 65535) {
    AttnSplitParams sp;
    sp.a = p;
    sp.g_full = a.head_num / a.heads_kv, sp.chunks = chunks;
    const int rule_kv = aff.k ? int(aff.cap) : a.sl_kv;  // (moving length: ranges, partials and descriptors laid out for the longest context)
    AttnParams pc = p;
    pc.sl_kv = rule_kv;
    const bool stream = attn_streams(pc);
    const int batch_keys = (G * 8 <= 16 ? 4 : 2) * 4 * (a.head_size > 64 ? 4 : 8);  // attn_stream_kernel's U x keys per workgroup step
    // head sizes above 128 (register kernel, 16 dims per lane) take the ring kernel's range rule too: 16 x 256 heads 14.6 -> 10.4 us at 512 keys,
    // 51.8 -> 36.5 at 8192; 8 heads on one kv head 51 -> 40 at 2048 (scripts/r05/attn_regs_rule.py; head sizes <= 32 lose with it at 2048+ keys)
    int nsplit = attn_nsplit(a.batch_size, a.heads_kv * chunks, a.sl_q, rule_kv, stream || a.head_size > 128, stream ? batch_keys : 0, stream && G >= 4);
    float* ws = nullptr;
    if (nsplit > 1) {
      // partials go to the caller's workspace (`tmp`, sized by bestla_fusion_attn_workspace_size: the reference's own
      // contract, and the only choice that works on a stream being captured for the first time); without one, to a
      // grow-only per-stream scratch
      ws = device_tmp ? reinterpret_cast<float*>(a.tmp) : nullptr;
      if (!ws) ws = static_cast<float*>(stream_scratch(st, attn_ws_bytes(a.batch_size, a.head_num, a.heads_kv, a.head_size, a.sl_q, rule_kv), 1));
      if (!ws) nsplit = 1;  // scratch allocation failed: unsplit, still correct
    }
    sp.ws = ws;
    sp.tickets = nullptr;
    const size_t nticket = size_t(a.batch_size) * a.sl_q * a.heads_kv * chunks;
    // (the replayed route merges inside the launch: one launch less per layer of a token that is all launches — profiles/r05r_*)
    if (nsplit > 1 && (g_attn_inlaunch.load(std::memory_order_relaxed) != 0 || (aff.k && aff.inlaunch)) && nticket <= kAttnTicketCap)
      sp.tickets = static_cast<uint32_t*>(stream_scratch_zeroed(st, kAttnTicketCap * 4, 22));  // nullptr (e.g. first use on a capturing stream): the merge launch
    sp.nsplit = nsplit;
    sp.keys_per_split = (rule_kv + nsplit - 1) / nsplit;
    sp.kmove = aff.k, sp.kdelta = int(aff.delta);
    sp.dyn_min_keys = sp.dyn_batch_keys = sp.dyn_single_below = 0;
    static const bool dyn_off = getenv("NS_ATTN_DYN_RANGES") && atoi(getenv("NS_ATTN_DYN_RANGES")) == 0;  // A-B runs
    if (aff.k && nsplit > 1 && !dyn_off && a.sl_q == 1) {
      // the constants attn_nsplit used above, for the workgroups to apply to the live length
      const bool ring = stream || a.head_size > 128;
      sp.dyn_min_keys = ring ? g_attn_min_keys_s.load() : g_attn_min_keys.load();
      sp.dyn_batch_keys = stream ? batch_keys : 0;
      sp.dyn_single_below = (ring && size_t(a.heads_kv) * chunks * a.sl_q * a.batch_size >= 32) ? 128 : 0;
    }
    // dispatch order (round 5, profiles/r05m_attn_layout_order_ab.txt): on a position-major cache ([position][head][dim]: a head's rows are
    // pieces one position stride apart) neighbouring workgroups should be neighbouring HEADS of one context range — split + merge 14.3 -> 13.7 us
    // at 2048 positions, 22.0 -> 19.9 at 4096 (Llama-2-7B shape); on a head-major cache (one slab per head) the ranges of one head first
    // (20.7 vs 23.4 us at 4096).  ns_hip_set_tuning("attn_heads_first", 0 / 1) forces one, -1 = by layout.
    const int hf = g_attn_heads_first.load();
    const bool position_major = a.step_k_head_num < a.step_k_sl;
    sp.heads_first = (hf < 0 ? position_major : hf != 0) && size_t(nsplit) <= 65535;
    const dim3 grid = sp.heads_first ? dim3(unsigned(a.heads_kv * a.sl_q * chunks), unsigned(nsplit), unsigned(a.batch_size))
                                     : dim3(unsigned(nsplit), unsigned(a.heads_kv * a.sl_q * chunks), unsigned(a.batch_size));
    hipError_t e = G == 1 ? launch_split_g<1>(sp, grid, st, stream)
                 : G == 2 ? launch_split_g<2>(sp, grid, st, stream)
                 : G == 4 ? launch_split_g<4>(sp, grid, st, stream)
                          : launch_split_g<8>(sp, grid, st, stream);
    if (e != hipSuccess) return e;
    if (nsplit > 1 && !sp.tickets) {
      hipLaunchKernelGGL(attn_merge_kernel, dim3(unsigned(a.head_num), unsigned(a.sl_q), unsigned(a.batch_size)), dim3(128), 0,
                         st, sp);
      e = hipGetLastError();
    }
    return e;
  }
  const dim3 grid(unsigned(a.sl_q), unsigned(a.head_num), unsigned(a.batch_size));
  hipLaunchKernelGGL(attn_kernel, grid, dim3(kAttnThreads), 0, st, p);
  return hipGetLastError();
}

// scratch a moving-length decode launch needs (partials for the longest context, the merge tickets): allocated BEFORE the route's capture
bool attn_prepare_moving(hipStream_t st, int batch, int heads, int heads_kv, int head_size, int cap) {
  const size_t b = attn_ws_bytes(batch, heads, heads_kv, head_size, 1, cap);
  if (b && !stream_scratch(st, b, 1)) return false;
  return stream_scratch_zeroed(st, kAttnTicketCap * 4, 22) != nullptr;
}

// elements spanned by a strided 4-d tensor
static size_t span4(int n0, long long s0, int n1, long long s1, int n2, long long s2, int n3, long long s3) {
  return size_t((n0 - 1) * s0 + (n1 - 1) * s1 + (n2 - 1) * s2 + (n3 - 1) * s3 + 1);
}


// ---- library-managed ("reordered") kv-cache, MI355X form: plain fp16 [batch][head][seq_max][head_size] -----------------
// The reference hands the cache to BesTLA as an opaque buffer (sizes / strides from bestla_reordered_attn_fp32_batch_kv_info,
// contents only through the update / shift / copy / forward entries, mha_dense.h:124-172) and packs it for AMX / AVX tiles.
// Nothing but this module looks inside, so the layout here is the one the attention kernels stream best.
// slab (optional): the device mirror of the cache, rows seq_off .. of every (batch, head) slab of seq_max rows
__global__ void kv_update_kernel(const float* __restrict__ src, _Float16* __restrict__ out, int batch, int heads, int hs,
                                 int seq, long long step_bs, long long step_head, long long step_seq, long long step_hs,
                                 _Float16* __restrict__ slab = nullptr, int seq_max = 0, int seq_off = 0) {
  const size_t gid = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const size_t total = size_t(batch) * heads * seq * hs;
  if (gid >= total) return;
  const int j = int(gid % hs);
  const int i = int((gid / hs) % seq);
  const int h = int((gid / (size_t(hs) * seq)) % heads);
  const int b = int(gid / (size_t(hs) * seq * heads));
  const _Float16 v = (_Float16)src[b * step_bs + h * step_head + i * step_seq + j * step_hs];
  out[gid] = v;  // out: [batch][head][seq][hs]
  if (slab) slab[((size_t(b) * heads + h) * seq_max + seq_off + i) * hs + j] = v;
}
// rows [seq_keep, seq_max) of every (batch, head): adjacent pairs rotated by the one angle set in cossin = {cos_0, sin_0,
// cos_1, sin_1, ...} (fp16), as ne_compute_forward_rope_bestla prepares it (ne_layers.c:9636-9651)
__global__ void kv_shift_rope_kernel(_Float16* __restrict__ rows, const _Float16* __restrict__ cossin, size_t nrows, int hs) {
  const size_t gid = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const int half = hs / 2;
  if (gid >= nrows * half) return;
  const size_t r = gid / half;
  const int pr = int(gid % half);
  _Float16* x = rows + r * hs + 2 * pr;
  const float c = float(cossin[2 * pr]), sn = float(cossin[2 * pr + 1]);
  const float x0 = float(x[0]), x1 = float(x[1]);
  x[0] = (_Float16)(x0 * c - x1 * sn);
  x[1] = (_Float16)(x0 * sn + x1 * c);
}

}  // namespace ns

using namespace ns;  // NOLINT

extern "C" {

int ns_hip_attn_set_head_partition(int global_head_num, int head_offset) {
  if (global_head_num < 0 || head_offset < 0 || (global_head_num > 0 && head_offset >= global_head_num)) {
    set_error("ns_hip_attn_set_head_partition: need 0 <= head_offset < global_head_num (or 0, 0 to clear)");
    return -1;
  }
  g_alibi_heads.store(global_head_num), g_alibi_off.store(global_head_num > 0 ? head_offset : 0);
  return 0;
}

size_t bestla_fusion_attn_workspace_size(const attn_shape_t* s) {
  // (m, l, acc) partials of the context splits; 64 bytes minimum so that callers always get a valid pointer
  return std::max<size_t>(64, attn_ws_bytes(s->batch_size, s->head_num, s->heads_kv, s->head_size, s->sl_q, s->sl_kv));
}

bool bestla_fusion_attn_fp32_fp16_fp16_fp32_support(const attn_shape_t* s) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return false;
  std::string why;
  return attn_shape_ok(s->head_num, s->heads_kv, s->head_size, s->sl_q, s->sl_kv, false, &why);
}

bool bestla_fusion_attn_fp16_support(const attn_shape_t*) { return false; }  // mha_dense.h:106: fp16 Q / dst form, not offered

bool bestla_reordered_attn_fp32_support(const attn_shape_t* params) {
  // mha_dense.cpp:70-80 answers by CPU features; here: whatever the attention kernels take (fp16 cache, see below)
  std::string why;
  int count = 0;
  if (!params || hipGetDeviceCount(&count) != hipSuccess || count <= 0) return false;
  return attn_shape_ok(params->head_num, params->heads_kv, params->head_size, params->sl_q, params->sl_kv, false, &why);
}

void bestla_reordered_attn_fp32_batch_kv_info(const kv_shape_t* params, kv_cache_info_t* out) {
  // mha_dense.cpp:82-112.  Byte strides, as the graph code uses them for its views (llama.cpp:544-560)
  if (!params || !out) return;
  const size_t row = size_t(params->head_size) * 2;
  out->k_layout = ATTN_FWD_LAYOUT_PLAIN;
  out->v_layout = ATTN_FWD_LAYOUT_PLAIN;
  out->stride_k_head_size = 2;
  out->stride_k_sl = int(row);
  out->stride_k_head_num = int(row * params->sl_kv_max);
  out->k_bytes = size_t(out->stride_k_head_num) * params->heads_kv;
  out->stride_v_head_size = 2;
  out->stride_v_sl = int(row);
  out->stride_v_head_num = int(row * params->sl_kv_max);
  out->v_bytes = size_t(out->stride_v_head_num) * params->heads_kv;
}

// ---- device mirrors of the library-managed kv caches -----------------------------------------------------------------------
// The graph keeps a library-managed cache (NE_TYPE_BTLA tensors, llama.cpp:544-560) in HOST memory and hands its address to every
// entry below; nothing but these entries ever reads or writes its bytes (the layout is this library's own).  Uploading the
// visible K / V rows for every attention call (32 MB per layer and token at 2048 positions) made the default, host-pointer
// route PCIe-bound.  So every cache the update entries see gets a DEVICE MIRROR, keyed by its host address range: updates,
// shifts and beam copies are applied to both copies (the host copy stays authoritative and complete), the attention entry
// reads the mirror.  A range that is not (wholly) mirrored falls back to the upload path; a new range that overlaps an old
// one replaces it (the tensor was re-created); ns_hip_cache_clear() and NS_KV_MIRROR=0 drop / disable the mirrors — needed
// only by a caller that writes cache bytes behind the library's back (restoring a saved session with memcpy).
static std::mutex& g_attn_host_mu_ref();
struct KvMirror {
  char* dev = nullptr;
  size_t bytes = 0;
};
static std::map<const char*, KvMirror> g_kv_mirrors;  // host base -> mirror; guarded by g_attn_host_mu
static bool kv_mirror_enabled() {
  static const bool on = !(getenv("NS_KV_MIRROR") && atoi(getenv("NS_KV_MIRROR")) == 0);
  return on;
}
// device address of host range [p, p + bytes) when a mirror covers it, else nullptr
static char* kv_mirror_find(const char* p, size_t bytes) {
  auto it = g_kv_mirrors.upper_bound(p);
  if (it == g_kv_mirrors.begin()) return nullptr;
  --it;
  if (p >= it->first && p + bytes <= it->first + it->second.bytes) return it->second.dev + (p - it->first);
  return nullptr;
}
// mirror of exactly [p, p + bytes): the covering one, or a new one filled from the host copy (overlapping older ones go)
static char* kv_mirror_get(const char* p, size_t bytes) {
  if (!kv_mirror_enabled() || bytes == 0) return nullptr;
  if (char* d = kv_mirror_find(p, bytes)) return d;
  for (auto it = g_kv_mirrors.begin(); it != g_kv_mirrors.end();) {
    if (it->first < p + bytes && p < it->first + it->second.bytes) {
      (void)hipFree(it->second.dev);
      it = g_kv_mirrors.erase(it);
    } else {
      ++it;
    }
  }
  KvMirror m;
  if (hipMalloc(reinterpret_cast<void**>(&m.dev), bytes) != hipSuccess || hipMemcpy(m.dev, p, bytes, hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipGetLastError();
    if (m.dev) (void)hipFree(m.dev);
    return nullptr;  // no mirror: the upload path serves this cache
  }
  m.bytes = bytes;
  g_kv_mirrors[p] = m;
  return m.dev;
}
static void kv_mirrors_clear_impl() {
  std::lock_guard<std::mutex> lock(g_attn_host_mu_ref());
  for (auto& kv : g_kv_mirrors) (void)hipFree(kv.second.dev);
  g_kv_mirrors.clear();
}

// the host-tensor entries stage through slots of the null stream's scratch; one call at a time
static std::mutex g_attn_host_mu;
static std::mutex& g_attn_host_mu_ref() { return g_attn_host_mu; }
constexpr int kScratchHostA = 10, kScratchHostB = 11, kScratchHostC = 12, kScratchHostD = 13;
// pinned, device-mapped staging for the decode-sized host calls (guarded by g_attn_host_mu): the kernels read / write host
// memory over PCIe, a call is memcpy + launches + ONE synchronisation instead of blocking hipMemcpy round trips
struct AttnPinned {
  void* host = nullptr;
  void* dev = nullptr;
  size_t cap = 0;
};
static AttnPinned g_attn_pin[4];
constexpr size_t kAttnZeroCopyMax = size_t(2) << 20;
static bool attn_zero_copy() {
  static const bool off = getenv("NS_NO_ZERO_COPY") != nullptr;  // diagnostics
  return !off;
}
static void* attn_pinned(int slot, size_t bytes, void** dev) {
  AttnPinned& e = g_attn_pin[slot];
  if (bytes > e.cap) {
    if (e.host) (void)hipHostFree(e.host);
    e = AttnPinned{};
    const size_t want = bytes < (size_t(1) << 16) ? (size_t(1) << 16) : bytes;
    if (hipHostMalloc(&e.host, want, hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer(&e.dev, e.host, 0) != hipSuccess) {
      (void)hipGetLastError();
      if (e.host) (void)hipHostFree(e.host);
      e = AttnPinned{};
      return nullptr;
    }
    e.cap = want;
  }
  *dev = e.dev;
  return e.host;
}
static bool kv_device() {
  int count = 0;
  if (hipGetDeviceCount(&count) == hipSuccess && count > 0) return true;
  (void)hipGetLastError();
  set_error("no HIP device visible: libns_hip.so has no CPU fallback");
  fprintf(stderr, "Err: invalid parameters (bestla_reordered_attn: no HIP device visible: libns_hip.so has no CPU fallback)\n");
  return false;
}

// update_k / update_v share one layout here: fp32 rows (any element steps) -> fp16 at [batch][head][seq_off + i][:]
static void kv_update(const bestla_fusion_attn_fp32_update_kv_args_t* pp, const char* who) {
  ns::HostScope host_scope("bestla_reordered_attn_fp32_update_k/v");
  if (!kv_device()) return;
  const auto& a = *pp;
  if (!a.src || !a.cache || a.batch_size < 0 || a.heads_kv <= 0 || a.head_size <= 0 || a.seq_size < 0 || a.seq_off < 0 ||
      a.seq_off + a.seq_size > a.seq_max) {
    set_error(std::string(who) + ": invalid argument");
    fprintf(stderr, "Err: invalid parameters (%s)\n", who);
    return;
  }
  const size_t total = size_t(a.batch_size) * a.heads_kv * a.seq_size * a.head_size;
  if (total == 0) return;
  const size_t nsrc = span4(a.batch_size, a.step_bs, a.heads_kv, a.step_head_num, a.seq_size, a.step_seq, a.head_size, a.step_head_size);
  std::lock_guard<std::mutex> host_lock(g_attn_host_mu);
  const size_t row_b = size_t(a.head_size) * 2;
  if (attn_zero_copy() && nsrc * 4 + total * 2 <= kAttnZeroCopyMax) {  // a token's rows: pinned staging, one launch, one synchronisation
    void *zsrc = nullptr, *zout = nullptr;
    float* hsrc = static_cast<float*>(attn_pinned(0, nsrc * 4, &zsrc));
    _Float16* hout = static_cast<_Float16*>(attn_pinned(1, total * 2, &zout));
    if (hsrc && hout) {
      const size_t slab = size_t(a.batch_size) * a.heads_kv * a.seq_max * row_b;
      const bool had = kv_mirror_find(a.cache, slab) != nullptr;  // (a mirror created now is filled from the host copy below)
      _Float16* mir = had ? reinterpret_cast<_Float16*>(kv_mirror_find(a.cache, slab)) : nullptr;
      memcpy(hsrc, a.src, nsrc * 4);
      hipLaunchKernelGGL(kv_update_kernel, dim3(unsigned((total + 255) / 256)), dim3(256), 0, nullptr, static_cast<const float*>(zsrc),
                         static_cast<_Float16*>(zout), a.batch_size, a.heads_kv, a.head_size, a.seq_size, (long long)a.step_bs,
                         (long long)a.step_head_num, (long long)a.step_seq, (long long)a.step_head_size, mir, a.seq_max, a.seq_off);
      bool ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(nullptr) == hipSuccess;
      if (ok) {
        const size_t chunk = size_t(a.seq_size) * row_b;
        for (size_t bh = 0; bh < size_t(a.batch_size) * a.heads_kv; bh++)
          memcpy(a.cache + (bh * a.seq_max + a.seq_off) * row_b, reinterpret_cast<const char*>(hout) + bh * chunk, chunk);
        if (!had) (void)kv_mirror_get(a.cache, slab);  // first sight of this cache: mirror = the (now complete) host copy
      } else {
        (void)hipGetLastError();
        set_error(std::string(who) + ": device launch failed");
        fprintf(stderr, "Err: invalid parameters (%s: device launch failed)\n", who);
      }
      return;
    }
  }
  // staging in the grow-only per-stream scratch (a hipMalloc / hipFree pair per call synchronises the whole device twice)
  float* dsrc = static_cast<float*>(stream_scratch(nullptr, nsrc * 4, kScratchHostA));
  _Float16* dout = static_cast<_Float16*>(stream_scratch(nullptr, total * 2, kScratchHostB));
  bool ok = dsrc && dout && hipMemcpy(dsrc, a.src, nsrc * 4, hipMemcpyHostToDevice) == hipSuccess;
  if (ok) {
    hipLaunchKernelGGL(kv_update_kernel, dim3(unsigned((total + 255) / 256)), dim3(256), 0, nullptr, dsrc, dout, a.batch_size,
                       a.heads_kv, a.head_size, a.seq_size, (long long)a.step_bs, (long long)a.step_head_num, (long long)a.step_seq,
                       (long long)a.step_head_size);
    // [batch x head] chunks of seq_size rows land at row seq_off of their seq_max-row slab
    const size_t row = size_t(a.head_size) * 2;
    ok = hipGetLastError() == hipSuccess &&
         hipMemcpy2D(a.cache + size_t(a.seq_off) * row, size_t(a.seq_max) * row, dout, size_t(a.seq_size) * row,
                     size_t(a.seq_size) * row, size_t(a.batch_size) * a.heads_kv, hipMemcpyDeviceToHost) == hipSuccess;
    // the device mirror of this cache (created from the host copy — which the line above just completed — on first sight)
    if (ok) {
      const size_t slab = size_t(a.batch_size) * a.heads_kv * a.seq_max * row;
      const bool fresh = kv_mirror_find(a.cache, slab) == nullptr;
      char* mir = kv_mirror_get(a.cache, slab);
      if (mir && !fresh)
        ok = hipMemcpy2D(mir + size_t(a.seq_off) * row, size_t(a.seq_max) * row, dout, size_t(a.seq_size) * row, size_t(a.seq_size) * row,
                         size_t(a.batch_size) * a.heads_kv, hipMemcpyDeviceToDevice) == hipSuccess;
    }
  }
  if (!ok) {
    (void)hipGetLastError();
    set_error(std::string(who) + ": device copy / launch failed");
    fprintf(stderr, "Err: invalid parameters (%s: device copy / launch failed)\n", who);
  }
}
void bestla_reordered_attn_fp32_update_k(const bestla_fusion_attn_fp32_update_kv_args_t* params) {
  kv_update(params, "bestla_reordered_attn_fp32_update_k");
}
void bestla_reordered_attn_fp32_update_v(const bestla_fusion_attn_fp32_update_kv_args_t* params) {
  kv_update(params, "bestla_reordered_attn_fp32_update_v");
}

void bestla_reordered_attn_fp32_shift_rope_k(char* cache, const uint16_t* cossin, int batch_size, int heads_kv, int head_size,
                                             int seq_max, int seq_keep) {
  if (!kv_device()) return;
  if (!cache || !cossin || batch_size < 0 || heads_kv <= 0 || head_size <= 0 || (head_size & 1) || seq_keep < 0 || seq_keep > seq_max) {
    set_error("bestla_reordered_attn_fp32_shift_rope_k: invalid argument");
    fprintf(stderr, "Err: invalid parameters (bestla_reordered_attn_fp32_shift_rope_k)\n");
    return;
  }
  const size_t row = size_t(head_size) * 2, slabs = size_t(batch_size) * heads_kv, n = size_t(seq_max - seq_keep);
  if (slabs == 0 || n == 0) return;
  std::lock_guard<std::mutex> host_lock(g_attn_host_mu);
  _Float16* drows = static_cast<_Float16*>(stream_scratch(nullptr, slabs * n * row, kScratchHostA));
  _Float16* dcs = static_cast<_Float16*>(stream_scratch(nullptr, row, kScratchHostB));
  bool ok = drows && dcs && hipMemcpy(dcs, cossin, row, hipMemcpyHostToDevice) == hipSuccess &&
            hipMemcpy2D(drows, n * row, cache + size_t(seq_keep) * row, size_t(seq_max) * row, n * row, slabs, hipMemcpyHostToDevice) ==
                hipSuccess;
  if (ok) {
    const size_t work = slabs * n * (head_size / 2);
    hipLaunchKernelGGL(kv_shift_rope_kernel, dim3(unsigned((work + 255) / 256)), dim3(256), 0, nullptr, drows, dcs, slabs * n, head_size);
    ok = hipGetLastError() == hipSuccess &&
         hipMemcpy2D(cache + size_t(seq_keep) * row, size_t(seq_max) * row, drows, n * row, n * row, slabs, hipMemcpyDeviceToHost) ==
             hipSuccess;
    if (ok)
      if (char* mir = kv_mirror_find(cache, slabs * size_t(seq_max) * row))
        ok = hipMemcpy2D(mir + size_t(seq_keep) * row, size_t(seq_max) * row, drows, n * row, n * row, slabs, hipMemcpyDeviceToDevice) ==
             hipSuccess;
  }
  if (!ok) {
    (void)hipGetLastError();
    set_error("bestla_reordered_attn_fp32_shift_rope_k: device copy / launch failed");
    fprintf(stderr, "Err: invalid parameters (bestla_reordered_attn_fp32_shift_rope_k: device copy / launch failed)\n");
  }
}

// beam search: rows [seq_off, seq_off + seq_size) of every head from one sequence's cache to another's (both in host
// memory, as the graph allocates them): a plain copy in this layout — data movement, no arithmetic
static void kv_batch_cpy(const bestla_fusion_attn_fp32_batch_cpy_kv_args_t* pp) {
  ns::HostScope host_scope("bestla_fusion_attn_fp32_batch_cpy_k/v");
  const auto& a = *pp;
  if (!a.src || !a.dst || a.heads_kv <= 0 || a.head_size <= 0 || a.seq_size <= 0 || a.seq_off < 0 || a.seq_off + a.seq_size > a.seq_max) return;
  const size_t row = size_t(a.head_size) * 2;
  for (int h = 0; h < a.heads_kv; h++)
    memcpy(a.dst + (size_t(h) * a.seq_max + a.seq_off) * row, a.src + (size_t(h) * a.seq_max + a.seq_off) * row, size_t(a.seq_size) * row);
  // the destination's device mirror takes the same rows (from the host copy just written: a beam copy is a few rows)
  std::lock_guard<std::mutex> host_lock(g_attn_host_mu);
  const size_t slab = size_t(a.heads_kv) * a.seq_max * row;
  if (char* mir = kv_mirror_find(a.dst, slab)) {
    if (hipMemcpy2D(mir + size_t(a.seq_off) * row, size_t(a.seq_max) * row, a.dst + size_t(a.seq_off) * row, size_t(a.seq_max) * row,
                    size_t(a.seq_size) * row, size_t(a.heads_kv), hipMemcpyHostToDevice) != hipSuccess) {
      (void)hipGetLastError();
      for (auto it = g_kv_mirrors.begin(); it != g_kv_mirrors.end(); ++it)  // cannot keep it current: drop it, the upload path takes over
        if (a.dst >= it->first && a.dst < it->first + it->second.bytes) {
          (void)hipFree(it->second.dev);
          g_kv_mirrors.erase(it);
          break;
        }
    }
  }
}
void bestla_fusion_attn_fp32_batch_cpy_k(const bestla_fusion_attn_fp32_batch_cpy_kv_args_t* params) { kv_batch_cpy(params); }
void bestla_fusion_attn_fp32_batch_cpy_v(const bestla_fusion_attn_fp32_batch_cpy_kv_args_t* params) { kv_batch_cpy(params); }

void bestla_reordered_attn_fp32_forward(const bestla_reordered_attn_fp32_fp32_fwd_args_t* rp) {
  ns::HostScope host_scope("bestla_reordered_attn_fp32_forward");
  // the graph passes BYTE strides of its K / V views (ne_layers.c:10238-10279; the sequence stride of V is not passed at
  // all: every layout knows its own) — here they are plain fp16 rows, so this is the fp16 attention entry
  attn_fp32_fp16_fp16_fp32_fwd_args_t a;
  memset(&a, 0, sizeof(a));
  a.Q = rp->Q, a.K = reinterpret_cast<uint16_t*>(rp->K), a.V = reinterpret_cast<uint16_t*>(rp->V), a.dst = rp->dst;
  a.Q_sc = rp->Q_sc, a.K_sc = rp->K_sc, a.V_sc = rp->V_sc, a.dst_sc = rp->dst_sc;
  a.tmp = rp->tmp, a.QK_scale = rp->QK_scale, a.attn_flags = rp->attn_flags;
  a.batch_size = rp->batch_size, a.head_num = rp->head_num, a.heads_kv = rp->heads_kv, a.head_size = rp->head_size;
  a.sl_q = rp->sl_q, a.sl_kv = rp->sl_kv;
  a.Q_layout = a.K_layout = a.V_layout = a.dst_layout = ATTN_FWD_LAYOUT_PLAIN;
  a.step_q_bs = rp->step_q_bs, a.step_q_head_num = rp->step_q_head_num, a.step_q_sl = rp->step_q_sl;
  a.step_k_bs = rp->stride_k_bs / 2, a.step_k_head_num = rp->stride_k_head_num / 2;
  a.step_k_sl = rp->stride_k_sl ? rp->stride_k_sl / 2 : rp->head_size, a.step_k_head_size = 1;
  a.step_v_bs = rp->stride_v_bs / 2, a.step_v_head_num = rp->stride_v_head_num / 2, a.step_v_sl = rp->head_size, a.step_v_head_size = 1;
  a.step_dst_bs = rp->step_dst_bs, a.step_dst_head_num = rp->step_dst_head_num, a.step_dst_sl = rp->step_dst_sl;
  if (rp->K_layout != ATTN_FWD_LAYOUT_PLAIN || rp->V_layout != ATTN_FWD_LAYOUT_PLAIN) {
    set_error("bestla_reordered_attn_fp32_forward: the cache was not laid out by this library (bestla_reordered_attn_fp32_batch_kv_info)");
    fprintf(stderr, "Err: invalid parameters (bestla_reordered_attn_fp32_forward: foreign cache layout)\n");
    return;
  }
  bestla_fusion_attn_fp32_fp16_fp16_fp32_forward(&a);
}

int ns_hip_attn_fp32_fp16_fp16_fp32_forward(const attn_fp32_fp16_fp16_fp32_fwd_args_t* a, void* stream) {
  return ns_hip_attn_fp32_fp16_fp16_fp32_forward_h(a, nullptr, stream);
}

int ns_hip_attn_fp32_fp16_fp16_fp32_forward_h(const attn_fp32_fp16_fp16_fp32_fwd_args_t* a, void* dst16, void* stream) {
  std::string why;
  // device API: `tmp`, when given, is DEVICE memory of bestla_fusion_attn_workspace_size(shape) bytes
  const hipError_t e = launch_attn(*a, static_cast<hipStream_t>(stream), &why, a->tmp != nullptr, dst16);
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
  // staging in the grow-only per-stream scratch (four hipMalloc / hipFree pairs per call synchronised the device eight times)
  std::lock_guard<std::mutex> host_lock(g_attn_host_mu);
  // K / V of a library-managed cache are already on the device (kv mirrors above): nothing to upload
  void* mk = kv_mirror_find(reinterpret_cast<const char*>(hp->K), nk * 2);
  void* mv = kv_mirror_find(reinterpret_cast<const char*>(hp->V), nv * 2);
  if (mk && mv && attn_zero_copy() && (nq + nd) * 4 <= kAttnZeroCopyMax) {  // decode over a mirrored cache: Q and dst through pinned memory
    void *zq = nullptr, *zd = nullptr;
    float* hq = static_cast<float*>(attn_pinned(2, nq * 4, &zq));
    float* hd = static_cast<float*>(attn_pinned(3, nd * 4, &zd));
    if (hq && hd) {
      memcpy(hq, hp->Q, nq * 4);
      memcpy(hd, hp->dst, nd * 4);  // keeps the bytes between strided rows
      a.Q = static_cast<float*>(zq);
      a.K = static_cast<uint16_t*>(mk);
      a.V = static_cast<uint16_t*>(mv);
      a.dst = static_cast<float*>(zd);
      a.tmp = nullptr;
      const bool okz = ns_hip_attn_fp32_fp16_fp16_fp32_forward(&a, nullptr) == 0 && hipStreamSynchronize(nullptr) == hipSuccess;
      if (okz) memcpy(hp->dst, hd, nd * 4);
      else fprintf(stderr, "Err: invalid parameters (bestla_fusion_attn_fp32_fp16_fp16_fp32_forward: %s)\n", ns_hip_last_error());
      return;
    }
  }
  void* dq = stream_scratch(nullptr, nq * 4, kScratchHostA);
  void* dk = mk ? mk : stream_scratch(nullptr, nk * 2, kScratchHostB);
  void* dv = mv ? mv : stream_scratch(nullptr, nv * 2, kScratchHostC);
  void* dd = stream_scratch(nullptr, nd * 4, kScratchHostD);
  bool ok = dq && dk && dv && dd;
  ok = ok && hipMemcpy(dq, hp->Q, nq * 4, hipMemcpyHostToDevice) == hipSuccess &&
       (mk || hipMemcpy(dk, hp->K, nk * 2, hipMemcpyHostToDevice) == hipSuccess) &&
       (mv || hipMemcpy(dv, hp->V, nv * 2, hipMemcpyHostToDevice) == hipSuccess) &&
       hipMemcpy(dd, hp->dst, nd * 4, hipMemcpyHostToDevice) == hipSuccess;  // keeps bytes between strided rows
  if (ok) {
    a.Q = static_cast<float*>(dq);
    a.K = static_cast<uint16_t*>(dk);
    a.V = static_cast<uint16_t*>(dv);
    a.dst = static_cast<float*>(dd);
    a.tmp = nullptr;  // the caller's tmp is host memory: partials go to the internal device scratch
    ok = ns_hip_attn_fp32_fp16_fp16_fp32_forward(&a, nullptr) == 0 && hipDeviceSynchronize() == hipSuccess &&
         hipMemcpy(hp->dst, dd, nd * 4, hipMemcpyDeviceToHost) == hipSuccess;
  } else {
    set_error("attention: device allocation / upload failed");
  }
  if (!ok) fprintf(stderr, "Err: invalid parameters (bestla_fusion_attn_fp32_fp16_fp16_fp32_forward: %s)\n", ns_hip_last_error());
}

}  // extern "C"

namespace ns {
void kv_mirrors_clear() { kv_mirrors_clear_impl(); }
void touch_attn_module() {
  hipFuncAttributes fa;
  (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(attn_merge_kernel));
  (void)hipGetLastError();
}
}  // namespace ns
