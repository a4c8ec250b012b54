// ns_device.hip — the device-backend half of neural-speed's BesTLA surface
// (/root/reference/neural_speed/core/ne_bestla.h:85-112, the set the reference guards with NS_SYCL; reference
// implementation core/layers/ne_bestla_sycl.cpp) for MI355X.  With these symbols a reference tree built with -DNS_SYCL
// runs its UNCHANGED graph device-resident: tensors of the offloaded layers live in HBM (ne_new_device_tensor_impl,
// ne_layers.c:904-1050), BTLA weights are re-laid-out once at load (model_files.h:1515-1527), every operator of those
// layers is a HIP launch on one stream and only the token ids go in and the logits come out over PCIe.
//
// This file holds the functions whose signatures are plain pointers — exported under the reference's own names, so
// libns_hip.so is the drop-in for them — and two kernels the tensor-level functions need (the tensor-level functions
// themselves take ne_tensor and live in glue/ne_bestla_hip_device.c, compiled against the reference's headers):
//   * nd_binary_kernel: ne's broadcasting add / mul over four strided dimensions (ne_bestla_sycl.cpp:174-295)
//   * mha_f32_kernel:   the device prototype's attention over its fp32 kv cache, K [batch][heads][n_ctx][head_size],
//                       V [batch][heads][head_size][n_ctx] (ne_bestla_sycl.cpp:592-880; llama.cpp:241-285 lays the cache
//                       out like that).  One workgroup per (head, query row, batch): scores, soft-max in LDS, P*V.
//                       HBM-bound on the kv stream; written for correctness first — the tuned kernels of this library
//                       (ns_attn.hip) read an fp16 cache, which the reference's device path does not create.
// "queue" is a hipStream_t throughout.
#include <hip/hip_runtime.h>
#include <vector>
#include <mutex>
#include <chrono>
#include <atomic>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <algorithm>
#include <sys/mman.h>
#include <unistd.h>
#include <thread>

#include "../../include/ns_bestla.h"
#include "ns_common.h"
#include "ns_route.h"

namespace ns {

struct NdArgs {
  long long ne0[4];      // extents of src0 / dst
  long long nb0[4];      // byte strides of src0
  long long ne1[4];      // extents of src1 (each 1 or ne0[i]: broadcast by modulo, as the reference does)
  long long nb1[4];
  long long nbd[4];
};
__global__ __launch_bounds__(256) void nd_binary_kernel(const char* a, const char* b, char* d, NdArgs g, long long total, int op) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const long long i0 = i % g.ne0[0];
  i /= g.ne0[0];
  const long long i1 = i % g.ne0[1];
  i /= g.ne0[1];
  const long long i2 = i % g.ne0[2];
  const long long i3 = i / g.ne0[2];
  const float x = *reinterpret_cast<const float*>(a + i3 * g.nb0[3] + i2 * g.nb0[2] + i1 * g.nb0[1] + i0 * g.nb0[0]);
  const float y = *reinterpret_cast<const float*>(b + (i3 % g.ne1[3]) * g.nb1[3] + (i2 % g.ne1[2]) * g.nb1[2] + (i1 % g.ne1[1]) * g.nb1[1] +
                                                  (i0 % g.ne1[0]) * g.nb1[0]);
  *reinterpret_cast<float*>(d + i3 * g.nbd[3] + i2 * g.nbd[2] + i1 * g.nbd[1] + i0 * g.nbd[0]) = op ? x * y : x + y;
}

// the common shapes of those nodes — dense tensors of one shape (residual adds, silu(x) * up) or a dense tensor with a row vector (mul by the norm weight) —
// without the per-element index arithmetic: four elements per thread (round 5: a prompt's 192 such launches took 29.7 us each on 1500 x 4096 floats)
__global__ __launch_bounds__(256) void dense_binary_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ d, long long total4,
                                                           int bcols4, int op) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total4) return;
  typedef float nfloat4 __attribute__((ext_vector_type(4)));
  const nfloat4 x = reinterpret_cast<const nfloat4*>(a)[i];
  const nfloat4 y = reinterpret_cast<const nfloat4*>(b)[bcols4 ? i % bcols4 : i];
  reinterpret_cast<nfloat4*>(d)[i] = op ? x * y : x + y;
}

// q [batch][seq][heads][hs] (contiguous), k [batch][heads_kv][n_ctx][hs], v [batch][heads_kv][hs][n_ctx], o like q.
// masked: key j is visible to query row iq when j <= iq + (seq_all - seq)  (ne_bestla_sycl.cpp:633-644)
__global__ __launch_bounds__(256) void mha_f32_kernel(const float* q, const float* k, const float* v, float* o, int seq, int seq_all,
                                                      int heads, int heads_kv, int hs, int n_ctx, float scale, int masked) {
  extern __shared__ float sm[];  // [seq_all] scores, then [256] reduction scratch
  float* sc = sm;
  float* red = sm + ((seq_all + 3) & ~3);
  const int ih = blockIdx.x, iq = blockIdx.y, ib = blockIdx.z;
  const int hkv = ih / (heads / heads_kv);
  const int t = threadIdx.x;
  const float* qp = q + ((size_t(ib) * seq + iq) * heads + ih) * hs;
  const float* kp = k + (size_t(ib) * heads_kv + hkv) * size_t(n_ctx) * hs;
  const float* vp = v + (size_t(ib) * heads_kv + hkv) * size_t(hs) * n_ctx;
  const int visible = masked ? min(seq_all, iq + (seq_all - seq) + 1) : seq_all;
  // ---- scores: a quarter-wave (16 lanes) per key, lanes split the head dimension (coalesced 64-byte reads of K) ----
  const int sub = t & 15, grp = t >> 4;  // 16 keys in flight per workgroup pass
  float mx = -INFINITY;
  for (int j0 = 0; j0 < visible; j0 += 16) {
    const int j = j0 + grp;
    float s = 0.f;
    if (j < visible) {
      const float* kr = kp + size_t(j) * hs;
      for (int e = sub; e < hs; e += 16) s += qp[e] * kr[e];
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    s += __shfl_xor(s, 8, 64);
    s *= scale;
    if (j < visible) {
      if (sub == 0) sc[j] = s;
      mx = fmaxf(mx, s);
    }
  }
  red[t] = mx;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (t < w) red[t] = fmaxf(red[t], red[t + w]);
    __syncthreads();
  }
  mx = red[0];
  __syncthreads();
  float sum = 0.f;
  for (int j = t; j < visible; j += 256) {
    const float e = expf(sc[j] - mx);
    sc[j] = e;
    sum += e;
  }
  red[t] = sum;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (t < w) red[t] += red[t + w];
    __syncthreads();
  }
  const float inv = 1.f / red[0];
  __syncthreads();
  // ---- P * V: V is [hs][n_ctx]: a quarter-wave per output element walks the keys (coalesced reads along n_ctx) ----
  for (int e0 = 0; e0 < hs; e0 += 16) {
    const int e = e0 + grp;
    float acc = 0.f;
    if (e < hs) {
      const float* vr = vp + size_t(e) * n_ctx;
      for (int j = sub; j < visible; j += 16) acc += sc[j] * vr[j];
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 4, 64);
    acc += __shfl_xor(acc, 8, 64);
    if (e < hs && sub == 0) o[((size_t(ib) * seq + iq) * heads + ih) * hs + e] = acc * inv;
  }
}

// The same attention with the context split over workgroups (round 4): one workgroup per (128-key range, head, query row) instead
// of one per (head, query row) — at 2048 cached positions the single-workgroup form has 32 workgroups stream 2 MB of fp32 K / V
// each.  Scores: a quarter-wave per key, the lane's DPL = head_size / 16 dims as 16-byte loads, all 8 keys of the quarter-wave
// requested before the first is used; softmax over the range in LDS; P.V: a quarter-wave per head dim walks its 128 keys of the
// TRANSPOSED V (8 consecutive per lane).  Partials (max, sum, unnormalised output) go to a per-stream workspace,
// mha_f32_merge_kernel combines them (one launch more; a single range writes the output itself).
constexpr int kMhaKS = 128;  // keys per workgroup
// stores / loads past every cache (sc0 sc1): partial results cross XCDs inside one launch (as ns_attn.hip st_through / ld_through)
__device__ __forceinline__ void dev_st_through(float* p, float v) {
  __hip_atomic_store(reinterpret_cast<uint32_t*>(p), __builtin_bit_cast(uint32_t, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ float dev_ld_through(const float* p) {
  return __builtin_bit_cast(float, __hip_atomic_load(reinterpret_cast<const uint32_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
}
typedef float dfloat4 __attribute__((ext_vector_type(4)));
template <int DPL>
__global__ __launch_bounds__(256) void mha_f32_split_kernel(const float* q, const float* k, const float* v, float* o, float* ws, int nsplit,
                                                            int seq, int seq_all, int heads, int heads_kv, int n_ctx, float scale,
                                                            int masked, const int* __restrict__ kmove, int kdelta, uint32_t* tickets = nullptr,
                                                            _Float16* o16 = nullptr) {
  constexpr int HS = 16 * DPL;
  // replayed device route (ns_common.h Affine): the context length moves with the graph's token counter; the grid and the partials'
  // layout are those of the longest context (nsplit = ranges of n_ctx), ranges past the live length leave at once and the merge
  // (always launched then) reads the live ones only
  if (kmove) {
    seq_all += kdelta * *kmove;
    if (int(blockIdx.x) * kMhaKS >= seq_all) return;
  }
  __shared__ float sc[kMhaKS];
  __shared__ float red[4];
  const int split = blockIdx.x, ih = blockIdx.y, bq = blockIdx.z, ib = bq / seq, iq = bq % seq;
  const int hkv = ih / (heads / heads_kv);
  const int t = threadIdx.x, sub = t & 15, grp = t >> 4, w = t >> 6;
  const float* qp = q + (size_t(bq) * heads + ih) * HS + sub * DPL;
  const float* kp = k + (size_t(ib) * heads_kv + hkv) * size_t(n_ctx) * HS;
  const float* vp = v + (size_t(ib) * heads_kv + hkv) * size_t(HS) * n_ctx;
  const int visible = masked ? min(seq_all, iq + (seq_all - seq) + 1) : seq_all;
  const int j0 = split * kMhaKS, j1 = min(visible, j0 + kMhaKS);
  float* wp = ws + ((size_t(bq) * heads + ih) * nsplit + split) * (2 + HS);
  if (j0 >= j1) {  // a range past this row's causal extent
    if (t == 0) wp[0] = -INFINITY, wp[1] = 0.f;
    return;
  }
  dfloat4 qv[DPL / 4];
#pragma unroll
  for (int c = 0; c < DPL / 4; c++) qv[c] = *reinterpret_cast<const dfloat4*>(qp + 4 * c);
  // ---- scores ----
  dfloat4 kv[kMhaKS / 16][DPL / 4];
#pragma unroll
  for (int i = 0; i < kMhaKS / 16; i++) {
    const int j = min(j0 + grp + 16 * i, j1 - 1);
#pragma unroll
    for (int c = 0; c < DPL / 4; c++) kv[i][c] = *reinterpret_cast<const dfloat4*>(kp + size_t(j) * HS + sub * DPL + 4 * c);
  }
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < kMhaKS / 16; i++) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < DPL / 4; c++)
#pragma unroll
      for (int e = 0; e < 4; e++) s += qv[c][e] * kv[i][c][e];
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    s += __shfl_xor(s, 8, 64);
    s = j0 + grp + 16 * i < j1 ? s * scale : -INFINITY;
    if (sub == 0) sc[grp + 16 * i] = s;
    mx = fmaxf(mx, s);
  }
#pragma unroll
  for (int off = 16; off < 64; off <<= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  if ((t & 63) == 0) red[w] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));  // finite: j0 < j1
  float pe = 0.f;
  if (t < kMhaKS) {
    pe = expf(sc[t] - mx);  // exp(-inf) = 0 past the range
    sc[t] = pe;
  }
  float sum = pe;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) sum += __shfl_xor(sum, off, 64);
  __syncthreads();  // red[] read by everybody, sc[] complete
  if ((t & 63) == 0) red[w] = sum;
  // ---- P . V over the transposed V: lane sub takes keys j0 + 8 sub .. + 7 of head dim grp + 16 i ----
  const bool v16 = (n_ctx & 3) == 0 && (reinterpret_cast<uintptr_t>(v) & 15) == 0;
  float pv[8];
#pragma unroll
  for (int e = 0; e < 8; e++) pv[e] = sc[8 * sub + e];
  const int jl = j0 + 8 * sub;
  float part[DPL];
#pragma unroll
  for (int i = 0; i < DPL; i++) {
    const float* vr = vp + size_t(grp + 16 * i) * n_ctx + jl;
    float acc = 0.f;
    if (v16 && jl + 8 <= j1) {
      const dfloat4 a = *reinterpret_cast<const dfloat4*>(vr), b2 = *reinterpret_cast<const dfloat4*>(vr + 4);
#pragma unroll
      for (int e = 0; e < 4; e++) acc += pv[e] * a[e] + pv[4 + e] * b2[e];
    } else {
#pragma unroll
      for (int e = 0; e < 8; e++)
        if (jl + e < j1) acc += pv[e] * vr[e];  // (cache cells past the range are not read: they may hold anything)
    }
    part[i] = acc;
  }
#pragma unroll
  for (int i = 0; i < DPL; i++) {
    float acc = part[i];
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 4, 64);
    acc += __shfl_xor(acc, 8, 64);
    part[i] = acc;
  }
  __syncthreads();
  const float l = red[0] + red[1] + red[2] + red[3];
  // replayed route with tickets (round 5): no merge launch — the live ranges of a (row, head) draw a self-resetting ticket once their
  // partials are out (stores and loads past the per-XCD L2s), the last one combines them in range order: the sums mha_f32_merge_kernel forms.
  // A single live range (contexts up to 128 keys) writes the output row itself.
  const int live = kmove ? (seq_all + kMhaKS - 1) / kMhaKS : nsplit;  // (seq_all already moved above)
  if (tickets && live == 1) {
    if (sub == 0) {
      float* op = o + (size_t(bq) * heads + ih) * HS;
#pragma unroll
      for (int i = 0; i < DPL; i++) {
        const float y = part[i] / l;
        op[grp + 16 * i] = y;
        if (o16) o16[(size_t(bq) * heads + ih) * HS + grp + 16 * i] = (_Float16)y;
      }
    }
    return;
  }
  if (sub == 0) {
    if (nsplit == 1 && !kmove) {
      float* op = o + (size_t(bq) * heads + ih) * HS;
#pragma unroll
      for (int i = 0; i < DPL; i++) op[grp + 16 * i] = part[i] / l;
    } else if (tickets) {
#pragma unroll
      for (int i = 0; i < DPL; i++) dev_st_through(wp + 2 + grp + 16 * i, part[i]);
      if (t == 0) dev_st_through(wp, mx), dev_st_through(wp + 1, l);
    } else {
#pragma unroll
      for (int i = 0; i < DPL; i++) wp[2 + grp + 16 * i] = part[i];
      if (t == 0) wp[0] = mx, wp[1] = l;
    }
  }
  if (!tickets) return;
  __shared__ uint32_t drawn_s;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  uint32_t* tk = tickets + size_t(bq) * heads + ih;
  if (t == 0) drawn_s = __hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (drawn_s != uint32_t(live - 1)) return;
  if (t == 0) __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // zero again for the next token
  if (t < HS) {
    const float* wq = ws + (size_t(bq) * heads + ih) * nsplit * (2 + HS);
    float mb = -INFINITY;
    for (int s2 = 0; s2 < live; s2++) mb = fmaxf(mb, dev_ld_through(wq + size_t(s2) * (2 + HS)));
    float lb = 0.f, ab = 0.f;
    for (int s2 = 0; s2 < live; s2++) {
      const float ms = dev_ld_through(wq + size_t(s2) * (2 + HS));
      const float c = ms != -INFINITY ? expf(ms - mb) : 0.f;
      lb += dev_ld_through(wq + size_t(s2) * (2 + HS) + 1) * c;
      if (ms != -INFINITY) ab += dev_ld_through(wq + size_t(s2) * (2 + HS) + 2 + t) * c;
    }
    const float y = ab / lb;
    o[(size_t(bq) * heads + ih) * HS + t] = y;
    if (o16) o16[(size_t(bq) * heads + ih) * HS + t] = (_Float16)y;
  }
}
// one workgroup of head_size threads per (query row, head): combine the ranges' (max, sum, output) in range order
__global__ void mha_f32_merge_kernel(const float* ws, float* o, int nsplit, int hs, int seq_all, const int* __restrict__ kmove, int kdelta, _Float16* o16) {
  const size_t row = blockIdx.x;  // (batch * seq + iq) * heads + head
  const int t = threadIdx.x;
  const float* wp = ws + row * nsplit * (2 + hs);
  if (kmove) nsplit = (seq_all + kdelta * *kmove + kMhaKS - 1) / kMhaKS;  // the live ranges of a layout made for the longest context
  float mb = -INFINITY;
  for (int s2 = 0; s2 < nsplit; s2++) mb = fmaxf(mb, wp[size_t(s2) * (2 + hs)]);
  float lb = 0.f, ab = 0.f;
  for (int s2 = 0; s2 < nsplit; s2++) {
    const float ms = wp[size_t(s2) * (2 + hs)];
    const float c = ms != -INFINITY ? expf(ms - mb) : 0.f;
    lb += wp[size_t(s2) * (2 + hs) + 1] * c;
    if (ms != -INFINITY) ab += wp[size_t(s2) * (2 + hs) + 2 + t] * c;  // (an empty range wrote no output columns)
  }
  const float y = ab / lb;
  o[row * hs + t] = y;
  if (o16) o16[row * hs + t] = (_Float16)y;  // (replayed route: the projection behind it streams fp16 activations)
}

// ---- the fp16 mirror's converters (round 5 made them for prompt-sized calls — the kernels above serve one query row per workgroup, a 1500-token prompt
//      re-read its whole K / V once per row: 184 ms through the reference's unchanged model_eval — round 6 feeds every call shape from them, ns_route.h) ----
// K [heads][n_ctx][hs] fp32 -> fp16 [heads][out_ctx][hs], positions lo .. hi of every head, 4 elements per thread (out_ctx = n_ctx: the mirror)
__global__ void mha_k_to_f16_kernel(const float* __restrict__ k, _Float16* __restrict__ out, int heads, int n_ctx, int out_ctx, int lo, int hi, int hs,
                                    uint32_t* overflow) {
  const size_t per_head = size_t(hi - lo) * hs / 4;
  const size_t gid = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (gid >= per_head * heads) return;
  const size_t h = gid / per_head, e = size_t(lo) * hs + (gid - h * per_head) * 4;
  const dfloat4 v = *reinterpret_cast<const dfloat4*>(k + h * size_t(n_ctx) * hs + e);
  typedef _Float16 dhalf4 __attribute__((ext_vector_type(4)));
  if (overflow && fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))) > 65504.f)
    __hip_atomic_store(overflow, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  *reinterpret_cast<dhalf4*>(out + h * size_t(out_ctx) * hs + e) = dhalf4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
}
// V [heads][hs][n_ctx] fp32 (transposed) -> fp16 [heads][out_ctx][hs], positions lo .. hi: 32 x 32 tiles through LDS (reads run along the positions, writes along the head dims)
__global__ __launch_bounds__(256) void mha_vt_to_f16_kernel(const float* __restrict__ v, _Float16* __restrict__ out, int n_ctx, int out_ctx, int lo, int hi, int hs,
                                                            uint32_t* overflow) {
  __shared__ float tile[32][33];
  const int h = blockIdx.z, j0 = lo + blockIdx.x * 32, d0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const float* src = v + size_t(h) * hs * n_ctx;
  bool big = false;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int d = d0 + ty + 8 * r, j = j0 + tx;
    const float x = (d < hs && j < hi) ? src[size_t(d) * n_ctx + j] : 0.f;
    big = big || fabsf(x) > 65504.f;
    tile[ty + 8 * r][tx] = x;
  }
  if (big && overflow) __hip_atomic_store(overflow, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __syncthreads();
  _Float16* dst = out + size_t(h) * out_ctx * hs;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int j = j0 + ty + 8 * r, d = d0 + tx;
    if (j < hi && d < hs) dst[size_t(j) * hs + d] = (_Float16)tile[tx][ty + 8 * r];
  }
}

// what bestla_device_load_storage leaves in the tensor object behind a device-resident BTLA weight
// (ne_layers.c:946-949 reserves bestla_device_storage_size() bytes there).  It starts with a word no BTLA blob can start
// with (a blob's first field is its size, bestla_storage.h:250-317): the host-pointer entry points (_support, forward)
// recognise a blob by that field and refuse this struct instead of parsing it.
struct DeviceStorage {
  uint64_t not_a_blob;  // 0xffffffffffffffff
  uint64_t magic;       // "NSHIPDEV"
  ns_weight* w;
  int n, k;
  uint64_t reserved[4];
};
constexpr uint64_t kDevMagic = 0x564544504948534eull;

struct Device {
  hipStream_t stream;
  int id;
};

// a strided binary node of the device route as plain data (ns_route.cpp)
inline RouteOp binary_op(uint32_t kind, const float* a, const float* b, float* d, const long long ne0[4], const long long nb0[4], const long long ne1[4],
                         const long long nb1[4], const long long nbd[4]) {
  RouteOp op;
  memset(&op, 0, sizeof(op));
  op.kind = kind, op.p[0] = a, op.p[1] = b, op.p[2] = d;
  for (int i = 0; i < 4; i++) op.i[i] = ne0[i], op.i[4 + i] = nb0[i], op.i[8 + i] = ne1[i], op.i[12 + i] = nb1[i], op.i[16 + i] = nbd[i];
  return op;
}

// ---- the fp16 mirror of the reference's fp32 device kv cache (round 6; ns_route.h says why and how it stays coherent) ----
namespace {
struct KvMirror {
  const char* k32 = nullptr;   // the fp32 K cache the graph handed to the attention node ([slots][heads_kv][n_ctx][hs])
  const char* v32 = nullptr;   // ... and its V cache ([slots][heads_kv][hs][n_ctx])
  size_t bytes32 = 0;          // of each
  int slots = 0, heads_kv = 0, hs = 0, n_ctx = 0;
  _Float16 *k16 = nullptr, *v16 = nullptr;  // [slots][heads_kv][n_ctx][hs] both
  std::vector<int> valid;      // per slot: positions [0, valid) of the mirror hold the cache
  int fresh_lo = 0, fresh_hi = 0;  // slot 0: positions a producer stored into the mirror itself since the last attention (kvm_for_producer)
};
std::vector<KvMirror*> g_kvms;
const char *g_kvm_lo = nullptr, *g_kvm_hi = nullptr;  // hull of every mirrored fp32 range (one compare rules a pointer out)
uint32_t* g_kvm_overflow = nullptr;
std::atomic<int> g_kv16{-1};

void kvm_hull() {
  g_kvm_lo = g_kvm_hi = nullptr;
  for (const KvMirror* m : g_kvms)
    for (const char* b : {m->k32, m->v32}) {
      if (!g_kvm_lo || b < g_kvm_lo) g_kvm_lo = b;
      if (!g_kvm_hi || b + m->bytes32 > g_kvm_hi) g_kvm_hi = b + m->bytes32;
    }
}
void kvm_drop(size_t i) {
  KvMirror* m = g_kvms[i];
  (void)hipFree(m->k16);
  (void)hipFree(m->v16);
  delete m;
  g_kvms.erase(g_kvms.begin() + long(i));
}
inline bool overlaps(const char* a, size_t na, const char* b, size_t nb) { return a < b + nb && b < a + na; }
// the mirror that holds the `slots` cache slots starting at (dK, dV), made (and older, overlapping ones dropped) when there is none
KvMirror* kvm_get(const float* dK, const float* dV, int slots, int heads_kv, int hs, int n_ctx, int* slot0) {
  const char *k = reinterpret_cast<const char*>(dK), *v = reinterpret_cast<const char*>(dV);
  const size_t per_slot = size_t(heads_kv) * n_ctx * hs * 4, bytes = per_slot * slots;
  for (KvMirror* m : g_kvms) {
    if (m->heads_kv != heads_kv || m->hs != hs || m->n_ctx != n_ctx || k < m->k32 || k + bytes > m->k32 + m->bytes32) continue;
    const size_t off = size_t(k - m->k32);
    if (off % per_slot != 0 || v != m->v32 + off) continue;
    *slot0 = int(off / per_slot);
    return m;
  }
  for (size_t i = g_kvms.size(); i-- > 0;)
    if (overlaps(k, bytes, g_kvms[i]->k32, g_kvms[i]->bytes32) || overlaps(v, bytes, g_kvms[i]->v32, g_kvms[i]->bytes32) ||
        overlaps(k, bytes, g_kvms[i]->v32, g_kvms[i]->bytes32) || overlaps(v, bytes, g_kvms[i]->k32, g_kvms[i]->bytes32))
      kvm_drop(i);
  KvMirror* m = new KvMirror();
  m->k32 = k, m->v32 = v, m->bytes32 = bytes, m->slots = slots, m->heads_kv = heads_kv, m->hs = hs, m->n_ctx = n_ctx;
  m->valid.assign(size_t(slots), 0);
  if (hipMalloc(reinterpret_cast<void**>(&m->k16), bytes / 2) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&m->v16), bytes / 2) != hipSuccess) {
    (void)hipGetLastError();
    (void)hipFree(m->k16);
    delete m;
    kvm_hull();
    return nullptr;
  }
  g_kvms.push_back(m);
  kvm_hull();
  *slot0 = 0;
  return m;
}
}  // namespace

bool kv16_enabled() {
  int v = g_kv16.load();
  if (v < 0) {
    const char* e = getenv("NS_DEVICE_KV");
    v = (e && (!strcmp(e, "f32") || !strcmp(e, "fp32") || !strcmp(e, "0"))) ? 0 : 1;
    g_kv16.store(v);
  }
  return v != 0;
}
void kv16_set(int on) { g_kv16.store(on < 0 ? -1 : (on != 0)); }
uint32_t* kvm_overflow_word() {
  if (!g_kvm_overflow) {
    if (hipHostMalloc(reinterpret_cast<void**>(&g_kvm_overflow), 64, hipHostMallocDefault) != hipSuccess) {
      (void)hipGetLastError();
      g_kvm_overflow = nullptr;
      return nullptr;
    }
    *g_kvm_overflow = 0;
  }
  return g_kvm_overflow;
}
bool kvm_overflowed() { return g_kvm_overflow && *reinterpret_cast<volatile uint32_t*>(g_kvm_overflow) != 0; }
void kvm_overflow_reset() {
  if (g_kvm_overflow) *reinterpret_cast<volatile uint32_t*>(g_kvm_overflow) = 0;
}
bool kvm_args_for_cell(const void* cell, KvMirrorArgs* out) {
  const char* c = static_cast<const char*>(cell);
  if (!g_kvm_lo || c < g_kvm_lo || c >= g_kvm_hi) return false;
  for (const KvMirror* m : g_kvms) {
    const bool in_k = c >= m->k32 && c < m->k32 + m->bytes32, in_v = c >= m->v32 && c < m->v32 + m->bytes32;
    if (!in_k && !in_v) continue;
    out->base32 = in_k ? m->k32 : m->v32, out->m16 = in_k ? m->k16 : m->v16, out->elems = (long long)(m->bytes32 / 4);
    out->n_ctx = m->n_ctx, out->hs = m->hs, out->transposed = in_k ? 0 : 1, out->overflow = kvm_overflow_word();
    return true;
  }
  return false;
}
bool kvm_note_foreign_write(const void* dst, size_t bytes) {
  const char* c = static_cast<const char*>(dst);
  if (!g_kvm_lo || !c || c >= g_kvm_hi || c + bytes <= g_kvm_lo) return false;
  bool hit = false;
  for (KvMirror* m : g_kvms)
    if (overlaps(c, bytes ? bytes : 1, m->k32, m->bytes32) || overlaps(c, bytes ? bytes : 1, m->v32, m->bytes32)) {
      std::fill(m->valid.begin(), m->valid.end(), 0);
      m->fresh_lo = m->fresh_hi = 0;
      hit = true;
    }
  return hit;
}
void kvm_set_valid(const void* k32, int valid) {
  const char* c = static_cast<const char*>(k32);
  if (!g_kvm_lo || c < g_kvm_lo || c >= g_kvm_hi) return;
  for (KvMirror* m : g_kvms)
    if (c >= m->k32 && c < m->k32 + m->bytes32) {
      const size_t per_slot = m->bytes32 / size_t(m->slots);
      m->valid[size_t(c - m->k32) / per_slot] = std::min(valid, m->n_ctx);
    }
}
bool kvm_for_producer(const void* k32, const void* v32, int heads_kv, int hs, int n_ctx, int n_past, int m, _Float16** k16, _Float16** v16) {
  if (!kv16_enabled() || n_past < 0 || m < 1 || n_past + m > n_ctx) return false;
  int slot0 = 0;
  KvMirror* mir = kvm_get(static_cast<const float*>(k32), static_cast<const float*>(v32), 1, heads_kv, hs, n_ctx, &slot0);
  if (!mir) return false;
  const size_t per_slot = size_t(heads_kv) * n_ctx * hs;
  *k16 = mir->k16 + size_t(slot0) * per_slot, *v16 = mir->v16 + size_t(slot0) * per_slot;
  if (slot0 == 0) mir->fresh_lo = n_past, mir->fresh_hi = n_past + m;
  return true;
}
void kvm_clear() {
  while (!g_kvms.empty()) kvm_drop(g_kvms.size() - 1);
  kvm_hull();
}

}  // namespace ns

extern "C" {
int ns_hip_lazy_flush(void);  // (defined with the lazy peephole below: a recorded norm / silu node is launched before anything else runs)

// The runtime builds its pinned staging for copies from / to pageable host memory under the first such copy (a prompt's embeddings: 24.6 MB in 2.1 ms the
// first time, 0.5 ms afterwards).  One 32 MB round trip at device creation — model-load time — moves that out of the first prompt.
static void warm_pageable_copies(hipStream_t s) {
  static const bool off = getenv("NS_WARM_UP") && atoi(getenv("NS_WARM_UP")) == 0;
  static std::once_flag once;
  if (off) return;
  std::call_once(once, [s] {
    const size_t n = size_t(32) << 20;
    void* d = nullptr;
    char* h = static_cast<char*>(calloc(n, 1));
    if (h && hipMalloc(&d, n) == hipSuccess) {
      (void)hipMemcpyAsync(d, h, n, hipMemcpyDefault, s);
      (void)hipMemcpyAsync(h, d, n, hipMemcpyDefault, s);
      (void)hipStreamSynchronize(s);
    }
    if (d) (void)hipFree(d);
    free(h);
    (void)hipGetLastError();
  });
}
/* ---- ne_bestla.h:86-96, ne_bestla_sycl.cpp:26-92 ---- */
void* bestla_create_device(bool profile) {
  (void)profile;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    ns::set_error("bestla_create_device: no HIP device visible");
    return nullptr;
  }
  ns::Device* d = new ns::Device();
  d->id = 0;
  if (hipGetDevice(&d->id) != hipSuccess || hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking) != hipSuccess) {
    delete d;
    ns::set_error("bestla_create_device: stream creation failed");
    return nullptr;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, d->id) == hipSuccess)
    fprintf(stderr, "bestla device: %s, %d CUs, %.1f GB\n", prop.name, prop.multiProcessorCount, double(prop.totalGlobalMem) / 1e9);
  ns::route_attach(d->stream);  // ns_route.cpp: the per-token graph this queue carries is verified and replayed
  (void)ns_hip_warm_up();       // the code objects of the GEMM / GEMV / attention / operator kernels are loaded here, not under the first prompt and token
  warm_pageable_copies(d->stream);
  return d;
}
void* bestla_get_device_queue(void* device) { return device ? static_cast<ns::Device*>(device)->stream : nullptr; }
static void finish_pending_loads_if_any();
void bestla_release_device(void* device) {
  if (!device) return;
  (void)ns_hip_lazy_flush();
  finish_pending_loads_if_any();
  ns::Device* d = static_cast<ns::Device*>(device);
  (void)ns::route_sync_point(d->stream);
  ns::route_invalidate();  // (the mirrors go below: no plan of ANY queue may keep a captured cache write into one)
  ns::route_detach(d->stream);
  ns::kvm_clear();
  (void)hipStreamSynchronize(d->stream);
  (void)hipStreamDestroy(d->stream);
  delete d;
}
size_t bestla_device_gmem_size(void* device) {
  (void)device;
  size_t fr = 0, total = 0;
  return hipMemGetInfo(&fr, &total) == hipSuccess ? total : 0;
}
// the reference's device pools come from here: what lies inside one of these allocations is device memory (no runtime query per copy)
static std::mutex g_pool_mu;
static std::vector<std::pair<const char*, size_t>> g_pools;
static bool in_device_pool(const void* p) {
  const char* c = static_cast<const char*>(p);
  std::lock_guard<std::mutex> lk(g_pool_mu);
  for (const auto& a : g_pools)
    if (c >= a.first && c < a.first + a.second) return true;
  return false;
}
void* bestla_device_malloc(size_t size, void* queue) {
  (void)queue;
  void* p = nullptr;
  if (hipMalloc(&p, size ? size : 1) != hipSuccess) {
    ns::set_error("bestla_device_malloc: out of device memory");
    return nullptr;
  }
  std::lock_guard<std::mutex> lk(g_pool_mu);
  g_pools.emplace_back(static_cast<const char*>(p), size ? size : 1);
  return p;
}
void bestla_device_free(void* ptr, void* queue) {
  (void)queue;
  // nothing recorded or in flight may still refer to the memory: the lazy node's operands, the route's window and plans, the kv mirrors,
  // the loads into the graph's slices
  (void)ns_hip_lazy_flush();
  ns::route_invalidate();
  ns::kvm_clear();
  finish_pending_loads_if_any();
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (size_t i = 0; i < g_pools.size(); i++)
      if (g_pools[i].first == static_cast<const char*>(ptr)) {
        g_pools.erase(g_pools.begin() + long(i));
        break;
      }
  }
  if (ptr) (void)hipFree(ptr);
}
// (Round 6 measured and did NOT adopt a library-side path for large copies to / from pageable host memory — two pinned 16 MB stages, the DMA of chunk i + 1 under
// the host copy of chunk i, that copy cut over eight threads — for the 192 MB of logits a 1500-token prompt's evaluation ends with (ne_layers.c:8345-8346): the
// runtime's own staged copy is as fast once the destination's pages exist (3.45 vs 3.8 ms) and what a FIRST evaluation pays is the first touch of those pages
// (10-18 ms either way, run to run); towards the device the runtime was faster outright (24.6 MB: 1.3 vs 3.5 ms).  profiles/r06_route_timings.txt.)
// Also measured and not adopted: faulting the destination's pages in by eight threads IN FRONT of the copy (the queue already empty): populate + copy
// 23.6-24.4 ms against 13.9-17.2 ms for the copy alone.  docs/kernels/experiments.md.)
// What IS done: two threads, while the queue still works on what the copy waits for — bestla_device_sync in front of the copy is deferred (ns_route.cpp
// route_defer_sync), so the copy call arrives with the prompt's launches in flight and the host otherwise idle.
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
// The destination of a large copy to pageable host memory, first-touched while the queue still works on what the copy waits for (contents untouched;
// 8 MB at a time, stopping as soon as the queue is empty; anything the call refuses — an older kernel, a special mapping — is left to the copy)
static void touch_destination_while_queue_runs(void* dst, size_t size, hipStream_t s) {
  static const bool off = getenv("NS_DEVICE_PRETOUCH") && atoi(getenv("NS_DEVICE_PRETOUCH")) == 0;
  if (off || size < (size_t(8) << 20)) return;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone || hipStreamQuery(s) != hipErrorNotReady) {  // (a query would end a capture)
    (void)hipGetLastError();
    return;
  }
  hipPointerAttribute_t at;
  const bool known = hipPointerGetAttributes(&at, dst) == hipSuccess && at.type != hipMemoryTypeUnregistered;
  (void)hipGetLastError();
  if (known) return;  // pinned / device / managed: nothing to fault in
  // two threads: 192 MB of fresh pages measured 17-19 ms with one, 9-11 with two, 5-13 with four, 15-16 with eight or sixteen (the GPU boxes' hosts)
  const uintptr_t pg = uintptr_t(sysconf(_SC_PAGESIZE)), step = uintptr_t(8) << 20;
  const uintptr_t lo = (reinterpret_cast<uintptr_t>(dst) + pg - 1) & ~(pg - 1), hi = (reinterpret_cast<uintptr_t>(dst) + size) & ~(pg - 1);
  if (hi <= lo) return;
  const uintptr_t mid = (lo + (hi - lo) / 2) & ~(pg - 1);
  std::atomic<bool> stop{false};
  auto run = [&stop, step](uintptr_t a, uintptr_t e) {
    for (; a < e && !stop.load(std::memory_order_relaxed); a += step)
      if (madvise(reinterpret_cast<void*>(a), size_t(std::min(step, e - a)), MADV_POPULATE_WRITE) != 0) break;
  };
  std::thread helper(run, mid, hi);
  for (uintptr_t a = lo; a < mid; a += step) {
    if (madvise(reinterpret_cast<void*>(a), size_t(std::min(step, mid - a)), MADV_POPULATE_WRITE) != 0) break;
    if (hipStreamQuery(s) != hipErrorNotReady) break;  // the queue is empty: the rest is left to the copy
  }
  stop.store(true);
  helper.join();
  (void)hipGetLastError();
}
void bestla_device_memcpy(void* dstptr, const void* srcptr, size_t size, void* queue) {
  (void)ns_hip_lazy_flush();
  // what the route holds back (its window) goes out first; a copy FROM device memory reads a tensor behind the window's last op
  const bool from_dev = srcptr && in_device_pool(srcptr);
  (void)ns::route_sync_point(queue, from_dev ? srcptr : nullptr, from_dev ? size : 0);
  if (!dstptr || !srcptr || !size) return;
  if (from_dev && !in_device_pool(dstptr)) touch_destination_while_queue_runs(dstptr, size, static_cast<hipStream_t>(queue));
  // (replayed tokens run on the plan's activations: the embeddings go there as well, the logits come from there — ns_route.cpp)
  void* twin = ns::route_twin_dst(dstptr, queue);
  srcptr = ns::route_translate_src(srcptr, queue);
  if (twin && hipMemcpyAsync(twin, srcptr, size, hipMemcpyDefault, static_cast<hipStream_t>(queue)) != hipSuccess)
    ns::set_error("bestla_device_memcpy failed");
  if (hipMemcpyAsync(dstptr, srcptr, size, hipMemcpyDefault, static_cast<hipStream_t>(queue)) != hipSuccess)
    ns::set_error("bestla_device_memcpy failed");
  ns::route_note_copy(queue);
  if (in_device_pool(dstptr)) {
    // a copy into a mirrored kv cache (a restored session, a beam's rows): its mirror starts over, and no plan may go on replaying on the old one.
    // The order matters: plans are dropped first (that leaves every mirror marked up to the last replayed token), THEN the mirror is marked empty
    if (ns::kvm_note_foreign_write(dstptr, size)) {
      ns::route_invalidate();
      (void)ns::kvm_note_foreign_write(dstptr, size);
    }
    ns::route_note_input(dstptr, size, queue);     // an evaluation's input: kept so that the evaluation can be issued again
  }
}
static bool device_sync_impl(void* queue) {
  (void)ns_hip_lazy_flush();
  ns::route_time_mark(queue, 0);
  (void)ns::route_sync_point(queue);
  if (ns::route_defer_sync(queue)) return false;  // (only launches in flight: the next copy's wait covers them — ns_route.cpp)
  (void)hipStreamSynchronize(static_cast<hipStream_t>(queue));
  return ns::route_after_sync(queue);
}
void bestla_device_sync(void* queue) { (void)device_sync_impl(queue); }
void bestla_device_memcpy_sync(void* dstptr, const void* srcptr, size_t size, void* queue) {
  static const bool timing = getenv("NS_ROUTE_TIMING") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  bestla_device_memcpy(dstptr, srcptr, size, queue);
  const bool again = device_sync_impl(queue);
  if (timing && size >= (size_t(8) << 20))
    fprintf(stderr, "route timing: bestla_device_memcpy_sync of %.1f MB took %.2f ms\n", size / 1e6,
            std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() / 1e3);
  if (again) {  // the evaluation this copy reads from was run again (fp16 overflow, ns_route.h): its results are fetched again
    bestla_device_memcpy(dstptr, srcptr, size, queue);
    (void)hipStreamSynchronize(static_cast<hipStream_t>(queue));
    (void)ns::route_after_sync(queue);
  }
  ns::route_time_mark(queue, 1);
}

/* ---- ne_bestla.h:97-98, ne_bestla_sycl.cpp:92-141 ---- */
size_t bestla_device_storage_size(void) { return sizeof(ns::DeviceStorage); }

/* hoststor: the BTLA blob as read from the model file (host memory, freed by the caller right after: model_files.h:1526);
 * devstor: the tensor's storage area; deviceptr: the slice of the device pool the graph reserved for this tensor (blob-sized,
 * 256-byte aligned: ne_layers.c:918-945).  The weight is re-laid-out into this library's streaming layout INSIDE that slice
 * whenever the layout is no larger than the blob (every 4- / 8-bit integer, f4 and fp8-with-fp32-scale format: other tile
 * padding, no reduce section); formats the layout widens (1-3 / 5-7 bit planes, E8M0 scales expanded to fp32) get an
 * allocation of their own and leave the slice unused.  Nothing is synchronised per tensor: the load is completed by the
 * first forward that meets a pending weight (one stream synchronisation per model). */
namespace {
struct PendingLoad {
  ns_weight* w;    // the weight itself, NOT the tensor's storage area: a model whose tensors are freed before any forward ran
  uint32_t* info;  // must not leave pointers into freed tensor memory behind
};
std::mutex g_load_mu;
std::vector<PendingLoad> g_pending;
std::vector<uint32_t*> g_info_chunks;  // pinned host memory, 1024 pairs per chunk
size_t g_info_used = 0;
uint64_t g_stats[6] = {0, 0, 0, 0, 0, 0};
hipStream_t g_load_stream = nullptr;

uint32_t* next_info_pair() {
  constexpr size_t kPairs = 1024;
  if (g_info_chunks.empty() || g_info_used == kPairs) {
    uint32_t* p = nullptr;
    if (hipHostMalloc((void**)&p, kPairs * 8, hipHostMallocDefault) != hipSuccess) return nullptr;
    g_info_chunks.push_back(p);
    g_info_used = 0;
  }
  return g_info_chunks.back() + 2 * g_info_used++;
}
long long now_us() {
  return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
// completes every pending load: ONE synchronisation of the load stream, then the per-weight words
void finish_pending_loads() {
  std::lock_guard<std::mutex> lk(g_load_mu);
  if (g_pending.empty()) return;
  const long long t0 = now_us();
  (void)hipStreamSynchronize(g_load_stream);
  for (const PendingLoad& pl : g_pending) {
    // a rejected blob marks the weight (load_failed): the forward refuses it loudly; the storage area is not touched here
    if (ns_hip_weight_finish_load(pl.w, pl.info) != 0) fprintf(stderr, "bestla_device_load_storage: %s\n", ns_hip_last_error());
  }
  g_pending.clear();
  g_stats[4] += uint64_t(now_us() - t0);
  g_stats[5] = 0;
  if (getenv("NS_LOAD_STATS"))
    fprintf(stderr, "bestla device load: %llu tensors, %.1f MB of blobs -> %.1f MB in the graph's slices + %.1f MB in own allocations, %.3f s\n",
            (unsigned long long)g_stats[0], g_stats[1] / 1e6, g_stats[2] / 1e6, g_stats[3] / 1e6, g_stats[4] / 1e6);
}
std::atomic<int> g_have_pending{0};
}  // namespace
static void finish_pending_loads_if_any() {
  if (g_have_pending.load()) {
    g_have_pending.store(0);
    finish_pending_loads();
  }
}

void bestla_device_load_storage(void* hoststor, void* devstor, void* deviceptr, void* queue) {
  if (!hoststor || !devstor) return;
  const long long t0 = now_us();
  ns::DeviceStorage* s = static_cast<ns::DeviceStorage*>(devstor);
  memset(s, 0, sizeof(*s));
  s->not_a_blob = ~0ull;
  uint64_t blob_bytes = 0;
  memcpy(&blob_bytes, hoststor, 8);  // a blob starts with its serialized size (bestla_storage.h:250-317) = the slice's size
  std::lock_guard<std::mutex> lk(g_load_mu);
  uint32_t* info = next_info_pair();
  static const bool own_alloc = getenv("NS_LOAD_OWN_ALLOC") != nullptr;  // diagnostics: the round-3 behaviour (slice left unused)
  ns_weight* w = info ? ns_hip_weight_load_async(hoststor, own_alloc ? nullptr : deviceptr, blob_bytes, queue, info) : nullptr;
  if (!w) {
    fprintf(stderr, "bestla_device_load_storage: %s\n", info ? ns_hip_last_error() : "no pinned memory for the load record");
    return;  // magic stays 0: the forward refuses the tensor loudly
  }
  s->magic = ns::kDevMagic;
  s->w = w;
  int bits = 0, bs = 0;
  uint64_t db = 0;
  ns_hip_weight_info(w, &s->n, &s->k, &bits, &bs, &db);
  g_pending.push_back(PendingLoad{w, info});
  g_load_stream = static_cast<hipStream_t>(queue);
  g_stats[0]++, g_stats[1] += blob_bytes;
  g_stats[ns_hip_weight_is_external(w) ? 2 : 3] += db;
  g_stats[4] += uint64_t(now_us() - t0);
  g_stats[5] = g_pending.size();
  g_have_pending.store(1);
}

void ns_hip_device_load_stats(uint64_t out[6]) {
  std::lock_guard<std::mutex> lk(g_load_mu);
  for (int i = 0; i < 6; i++) out[i] = g_stats[i];
}

/* ---- ne_bestla.h:99-100, ne_bestla_sycl.cpp:149-171; called by ne_compute_forward_mul_mat_q_f32_bestla
 *      (ne_layers.c:7305-7309) with device pointers ---- */
void bestla_device_f32f32_forward(float* activation, void* weiptr, float* output, int _m, int _n, int _k, int lda, int ldo,
                                  void* workspace, void* queue) {
  (void)workspace;
  (void)ns_hip_lazy_flush();
  if (g_have_pending.load()) {  // first forward after a model load: one synchronisation completes every weight
    g_have_pending.store(0);
    finish_pending_loads();
  }
  const ns::DeviceStorage* s = static_cast<const ns::DeviceStorage*>(weiptr);
  if (!s || s->magic != ns::kDevMagic || !s->w || s->n != _n || s->k != _k) {
    ns::set_error("bestla_device_f32f32_forward: not a weight loaded by bestla_device_load_storage (or a shape mismatch)");
    printf("Err: invalid parameters (bestla_device_f32f32_forward: %s)\n", ns_hip_last_error());
    return;
  }
  if (s->w->load_failed) {  // (refused HERE: a window or a plan would otherwise carry the weight into a fused launch — ADVICE r05)
    ns::set_error("bestla_device_f32f32_forward: the blob behind this weight was rejected when it was loaded");
    printf("Err: invalid parameters (bestla_device_f32f32_forward: %s)\n", ns_hip_last_error());
    return;
  }
  int rc;
  if (ns::route_hook(queue)) {
    ns::RouteOp op;
    memset(&op, 0, sizeof(op));
    op.kind = ns::RK_GEMM;
    op.p[0] = activation, op.p[1] = s->w, op.p[2] = output;
    op.i[0] = _m, op.i[1] = _n, op.i[2] = _k, op.i[3] = lda, op.i[4] = ldo;
    rc = ns::route_submit(op);
  } else {
    rc = ns_hip_f32f32_forward(activation, s->w, output, _m, lda, ldo, NS_EPI_NONE, nullptr, 0, queue);
  }
  if (rc != 0) printf("Err: invalid parameters (bestla_device_f32f32_forward: %s)\n", ns_hip_last_error());
}

/* releases the device copy behind a storage area (the reference never frees its device weights either; offered for
 * callers that reload models in one process) */
void ns_hip_device_storage_release(void* devstor) {
  ns::DeviceStorage* s = static_cast<ns::DeviceStorage*>(devstor);
  if (g_have_pending.load()) {
    g_have_pending.store(0);
    finish_pending_loads();
  }
  if (s && s->magic == ns::kDevMagic && s->w) {
    ns::route_invalidate();  // (a plan holds the weight's device arrays and recognises it by its address — which the next load may be given again)
    ns_hip_weight_free(s->w);
    s->w = nullptr;
    s->magic = 0;
  }
}

/* ---- lazy peephole of the device route (round 4).  The reference's graph runs node by node: rms_norm, then the multiply by the norm
 * weight; silu(gate), then the multiply by up — four launches per layer that are two.  The glue hands rms_norm and silu to
 * ns_hip_lazy_*; they are RECORDED, not launched.  The next call decides: a multiply that consumes the recorded result (the
 * llama graph always does) runs ONE kernel that writes BOTH tensors — the recorded node's own output and the product, bit for bit
 * what the two kernels write, so every tensor of the graph keeps its contents whatever else reads it; anything else (another
 * operator from the glue, a GEMM, a copy, a synchronisation) first launches the recorded node as it stands (ns_hip_lazy_flush).
 * One recorded node at most; the reference's executor issues device nodes from one thread. ---- */
extern "C++" {
namespace ns {
hipError_t launch_rmsnorm(int norm_count, int norm_size, bool isrms, float eps, const float* in, float* out, hipStream_t st, const float* gamma,
                          void* out16);
hipError_t launch_rmsnorm_mul2(int norm_count, int norm_size, bool isrms, float eps, const float* in, float* plain, const float* gamma, float* out,
                               hipStream_t st);
hipError_t launch_silu(const float* x, float* y, size_t n, hipStream_t st);
hipError_t launch_silu_mul2(const float* x, const float* y, float* s, float* out, size_t n, int silu_first, hipStream_t st);
namespace {
struct LazyNode {
  int kind = 0;  // 0 none, 1 rms norm, 2 silu
  const float* src = nullptr;
  float* dst = nullptr;
  int rows = 0, cols = 0;
  float eps = 0.f;
  size_t n = 0;
  hipStream_t st = nullptr;
};
LazyNode g_lazy;
std::atomic<int> g_lazy_on{-1};  // NS_DEV_LAZY=0: every node launches as it comes (A/B)
bool lazy_on() {
  int v = g_lazy_on.load();
  if (v < 0) {
    const char* e = getenv("NS_DEV_LAZY");
    v = e ? atoi(e) != 0 : 1;
    g_lazy_on.store(v);
  }
  return v != 0;
}
int lazy_flush_impl() {
  const LazyNode n = g_lazy;
  g_lazy.kind = 0;
  hipError_t e = hipSuccess;
  if (n.kind == 1) e = launch_rmsnorm(n.rows, n.cols, true, n.eps, n.src, n.dst, n.st, nullptr, nullptr);
  if (n.kind == 2) e = launch_silu(n.src, n.dst, n.n, n.st);
  if (e != hipSuccess) {
    set_error("device route: launching a deferred node failed");
    return -1;
  }
  return 0;
}
}  // namespace
}  // namespace ns
}  // extern "C++"
int ns_hip_lazy_flush(void) { return ns::g_lazy.kind ? ns::lazy_flush_impl() : 0; }
int ns_hip_lazy_rms_norm(int rows, int cols, float eps, const float* dIn, float* dOut, void* stream) {
  if (ns::route_hook(stream)) {
    ns::RouteOp op;
    memset(&op, 0, sizeof(op));
    op.kind = ns::RK_RMSNORM, op.p[0] = dIn, op.p[1] = dOut, op.i[0] = rows, op.i[1] = cols, op.f[0] = eps;
    return ns::route_submit(op);
  }
  if (ns_hip_lazy_flush() != 0) return -1;
  if (!dIn || !dOut || rows < 1 || cols < 1) {
    ns::set_error("lazy rms_norm: invalid argument");
    return -1;
  }
  ns::g_lazy = ns::LazyNode{1, dIn, dOut, rows, cols, eps, size_t(rows) * cols, static_cast<hipStream_t>(stream)};
  return ns::lazy_on() ? 0 : ns_hip_lazy_flush();
}
int ns_hip_lazy_silu(const float* dSrc, float* dDst, size_t n, void* stream) {
  if (ns::route_hook(stream)) {
    ns::RouteOp op;
    memset(&op, 0, sizeof(op));
    op.kind = ns::RK_SILU, op.p[0] = dSrc, op.p[1] = dDst, op.i[0] = (long long)n;
    return ns::route_submit(op);
  }
  if (ns_hip_lazy_flush() != 0) return -1;
  if (n && (!dSrc || !dDst)) {
    ns::set_error("lazy silu: null argument");
    return -1;
  }
  if (!n) return 0;
  ns::g_lazy = ns::LazyNode{2, dSrc, dDst, 0, 0, 0.f, n, static_cast<hipStream_t>(stream)};
  return ns::lazy_on() ? 0 : ns_hip_lazy_flush();
}
int ns_hip_binary_nd_f32(int is_mul, const float* dA, const float* dB, float* dDst, const long long ne0[4], const long long nb0[4],
                         const long long ne1[4], const long long nb1[4], const long long nbd[4], void* stream);
/* the multiply node: fused with the recorded node when it consumes it, else the recorded node first and the plain kernel */
int ns_hip_lazy_mul(const float* dA, const float* dB, float* dDst, const long long ne0[4], const long long nb0[4], const long long ne1[4],
                    const long long nb1[4], const long long nbd[4], void* stream) {
  if (ns::route_hook(stream)) return ns::route_submit(ns::binary_op(ns::RK_MUL, dA, dB, dDst, ne0, nb0, ne1, nb1, nbd));
  const ns::LazyNode n = ns::g_lazy;
  auto packed = [](const long long ne[4], const long long nb[4]) {
    return nb[0] == 4 && nb[1] == 4 * ne[0] && nb[2] == nb[1] * ne[1] && nb[3] == nb[2] * ne[2];
  };
  if (n.kind && dA && dB && dDst && n.st == static_cast<hipStream_t>(stream) && packed(ne0, nb0) && packed(ne0, nbd)) {
    const long long total = ne0[0] * ne0[1] * ne0[2] * ne0[3];
    if (n.kind == 1 && dA == n.dst && total == (long long)n.n && ne0[0] == n.cols && ne1[0] == n.cols && ne1[1] <= 1 && ne1[2] <= 1 && ne1[3] <= 1 &&
        nb1[0] == 4 && dDst != n.src && dDst != n.dst) {  // (in place — ne_mul_inplace: dst == the norm's output — the two stores
                                                             // of the fused kernel would alias: flush and run the plain kernel)
      ns::g_lazy.kind = 0;
      if (ns::launch_rmsnorm_mul2(n.rows, n.cols, true, n.eps, n.src, n.dst, dB, dDst, n.st) != hipSuccess) {
        ns::set_error("device route: norm . weight launch failed");
        return -1;
      }
      return 0;
    }
    const bool same_shape = ne1[0] == ne0[0] && ne1[1] == ne0[1] && ne1[2] == ne0[2] && ne1[3] == ne0[3] && packed(ne1, nb1);
    if (n.kind == 2 && same_shape && total == (long long)n.n && (dA == n.dst) != (dB == n.dst) && dDst != n.src && dDst != n.dst) {
      ns::g_lazy.kind = 0;
      const bool first = dA == n.dst;
      if (ns::launch_silu_mul2(n.src, first ? dB : dA, n.dst, dDst, n.n, first ? 1 : 0, n.st) != hipSuccess) {
        ns::set_error("device route: silu . up launch failed");
        return -1;
      }
      return 0;
    }
  }
  if (ns_hip_lazy_flush() != 0) return -1;
  return ns_hip_binary_nd_f32(1, dA, dB, dDst, ne0, nb0, ne1, nb1, nbd, stream);
}

/* ---- kernels behind the tensor-level functions of glue/ne_bestla_hip_device.c ---- */
int ns_hip_binary_nd_f32(int is_mul, const float* dA, const float* dB, float* dDst, const long long ne0[4], const long long nb0[4],
                         const long long ne1[4], const long long nb1[4], const long long nbd[4], void* stream) {
  if (ns::route_hook(stream)) return ns::route_submit(ns::binary_op(is_mul ? ns::RK_MUL : ns::RK_ADD, dA, dB, dDst, ne0, nb0, ne1, nb1, nbd));
  if (ns_hip_lazy_flush() != 0) return -1;
  if (!dA || !dB || !dDst) {
    ns::set_error("binary_nd: null argument");
    return -1;
  }
  ns::NdArgs g;
  long long total = 1;
  for (int i = 0; i < 4; i++) {
    g.ne0[i] = ne0[i], g.nb0[i] = nb0[i], g.ne1[i] = ne1[i] > 0 ? ne1[i] : 1, g.nb1[i] = nb1[i], g.nbd[i] = nbd[i];
    total *= ne0[i];
  }
  if (total <= 0) return 0;
  {
    auto dense = [&](const long long* ne, const long long* nb) {
      long long st = 4;
      for (int i = 0; i < 4; i++) {
        if (ne[i] != 1 && nb[i] != st) return false;
        st *= ne[i];
      }
      return true;
    };
    const bool same = g.ne1[0] == g.ne0[0] && g.ne1[1] == g.ne0[1] && g.ne1[2] == g.ne0[2] && g.ne1[3] == g.ne0[3];
    const bool rowvec = g.ne1[0] == g.ne0[0] && g.ne1[1] == 1 && g.ne1[2] == 1 && g.ne1[3] == 1 && g.nb1[0] == 4;
    static const bool off = getenv("NS_DENSE_BINARY") && atoi(getenv("NS_DENSE_BINARY")) == 0;  // diagnostics
    if (!off && total >= 4096 && (g.ne0[0] & 3) == 0 && dense(g.ne0, g.nb0) && dense(g.ne0, g.nbd) && ((same && dense(g.ne1, g.nb1)) || rowvec) &&
        ((reinterpret_cast<uintptr_t>(dA) | reinterpret_cast<uintptr_t>(dB) | reinterpret_cast<uintptr_t>(dDst)) & 15) == 0) {
      const long long total4 = total / 4;
      hipLaunchKernelGGL(ns::dense_binary_kernel, dim3(unsigned((total4 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), dA, dB, dDst, total4,
                         same ? 0 : int(g.ne0[0] / 4), is_mul ? 1 : 0);
      if (hipGetLastError() != hipSuccess) {
        ns::set_error("binary_nd: launch failed");
        return -1;
      }
      return 0;
    }
  }
  hipLaunchKernelGGL(ns::nd_binary_kernel, dim3(unsigned((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     reinterpret_cast<const char*>(dA), reinterpret_cast<const char*>(dB), reinterpret_cast<char*>(dDst), g, total,
                     is_mul ? 1 : 0);
  if (hipGetLastError() != hipSuccess) {
    ns::set_error("binary_nd: launch failed");
    return -1;
  }
  return 0;
}

int ns_hip_mha_f32_device_layout(const float* dQ, const float* dK, const float* dV, float* dO, int batch, int seq, int seq_all, int heads,
                                 int heads_kv, int head_size, int n_ctx, float scale, int masked, void* stream) {
  if (!dQ || !dK || !dV || !dO || batch < 1 || seq < 1 || seq_all < seq || heads < 1 || heads_kv < 1 || heads % heads_kv || head_size < 1 ||
      n_ctx < seq_all) {
    ns::set_error("mha_f32: invalid argument");
    return -1;
  }
  if (ns::route_hook(stream)) {
    ns::RouteOp op;
    memset(&op, 0, sizeof(op));
    op.kind = ns::RK_MHA, op.p[0] = dQ, op.p[1] = dK, op.p[2] = dV, op.p[3] = dO;
    op.i[0] = batch, op.i[1] = seq, op.i[2] = seq_all, op.i[3] = heads, op.i[4] = heads_kv, op.i[5] = head_size, op.i[6] = n_ctx, op.i[7] = masked;
    op.f[0] = scale;
    return ns::route_submit(op);
  }
  if (ns_hip_lazy_flush() != 0) return -1;
  ns::g_mha_out16_written = false;
  const ns::Affine aff = ns::g_affine;
  // ---- fp16 mirror of the cache + this library's attention kernels (round 6, ns_route.h): every call shape — a prompt's rows on the matrix-core
  //      prefill kernels (round 5 converted the call's K / V into scratch for those: 184 -> 81 ms for a 1500-token prompt), a decode step on the
  //      LDS-ring kernel, a replayed decode step (aff.k) on the same kernel with its context length moving with the graph's token counter ----
  if (ns::kv16_enabled() && head_size % 8 == 0 && head_size <= 256 && (reinterpret_cast<uintptr_t>(dK) & 15) == 0 && (reinterpret_cast<uintptr_t>(dV) & 15) == 0 &&
      (reinterpret_cast<uintptr_t>(dQ) & 15) == 0 && size_t(batch) * heads_kv <= 65535 && size_t(seq) * heads * head_size < (size_t(1) << 31) &&
      size_t(n_ctx) * heads_kv * head_size < (size_t(1) << 31) && (!aff.k || seq == 1)) {
    attn_shape_t shp;
    memset(&shp, 0, sizeof(shp));
    shp.batch_size = batch, shp.head_num = heads, shp.heads_kv = heads_kv, shp.head_size = head_size, shp.sl_q = seq, shp.sl_kv = seq_all;
    hipStream_t st = static_cast<hipStream_t>(stream);
    int slot0 = 0;
    ns::KvMirror* mir = bestla_fusion_attn_fp32_fp16_fp16_fp32_support(&shp) ? ns::kvm_get(dK, dV, batch, heads_kv, head_size, n_ctx, &slot0) : nullptr;
    if (mir) {
      const size_t per_slot = size_t(heads_kv) * n_ctx * head_size;
      _Float16 *k16 = mir->k16 + size_t(slot0) * per_slot, *v16 = mir->v16 + size_t(slot0) * per_slot;
      if (!aff.k) {
        // positions this evaluation wrote (the last `seq`) are converted again, older ones once; several slots: each from its own mark
        // (a caller outside the route — a test, a bench — may have filled the cache by any means: everything live is converted for it)
        int lo = ns::route_executing() ? seq_all - seq : 0;
        for (int b = 0; b < batch; b++) lo = std::min(lo, mir->valid[size_t(slot0 + b)]);
        lo = std::max(0, lo);
        // rows the fused QKV launch of this evaluation stored into the mirror itself (ns_route.cpp XK_QKV_ROPE_M) are in place
        if (ns::route_executing() && batch == 1 && slot0 == 0 && mir->fresh_hi > mir->fresh_lo && lo >= mir->fresh_lo && seq_all <= mir->fresh_hi) lo = seq_all;
        mir->fresh_lo = mir->fresh_hi = 0;
        if (lo < seq_all) {
          const int hb = batch * heads_kv, npos = seq_all - lo;
          const size_t units = size_t(hb) * npos * head_size / 4;
          uint32_t* ovf = ns::kvm_overflow_word();
          hipLaunchKernelGGL(ns::mha_k_to_f16_kernel, dim3(unsigned((units + 255) / 256)), dim3(256), 0, st, dK, k16, hb, n_ctx, n_ctx, lo, seq_all, head_size, ovf);
          hipLaunchKernelGGL(ns::mha_vt_to_f16_kernel, dim3(unsigned((npos + 31) / 32), unsigned((head_size + 31) / 32), unsigned(hb)), dim3(256), 0, st, dV, v16,
                             n_ctx, n_ctx, lo, seq_all, head_size, ovf);
          if (hipGetLastError() != hipSuccess) {
            ns::set_error("mha_f32: fp16 mirror conversion launch failed");
            return -1;
          }
        }
        for (int b = 0; b < batch; b++) mir->valid[size_t(slot0 + b)] = seq_all;
      }
      attn_fp32_fp16_fp16_fp32_fwd_args_t a;
      memset(&a, 0, sizeof(a));
      a.Q = const_cast<float*>(dQ), a.K = reinterpret_cast<uint16_t*>(k16), a.V = reinterpret_cast<uint16_t*>(v16), a.dst = dO;
      a.Q_sc = a.K_sc = a.V_sc = a.dst_sc = 1.f;
      a.QK_scale = scale;
      a.attn_flags = masked ? NS_ATTN_FLAG_IS_CAUSAL : NS_ATTN_FLAG_NONE;
      a.batch_size = batch, a.head_num = heads, a.heads_kv = heads_kv, a.head_size = head_size, a.sl_q = seq, a.sl_kv = seq_all;
      a.step_q_bs = seq * heads * head_size, a.step_q_head_num = head_size, a.step_q_sl = heads * head_size;
      a.step_k_bs = heads_kv * n_ctx * head_size, a.step_k_head_num = n_ctx * head_size, a.step_k_sl = head_size, a.step_k_head_size = 1;
      a.step_v_bs = a.step_k_bs, a.step_v_head_num = a.step_k_head_num, a.step_v_sl = head_size, a.step_v_head_size = 1;
      a.step_dst_bs = a.step_q_bs, a.step_dst_head_num = head_size, a.step_dst_sl = heads * head_size;
      if (aff.k) {  // (launch_attn reads the moving length from g_affine; the cache's capacity bounds it)
        ns::g_affine.cap = n_ctx;
        static const bool inl_off = getenv("NS_MHA_INLAUNCH") && atoi(getenv("NS_MHA_INLAUNCH")) == 0;
        ns::g_affine.inlaunch = inl_off ? 0 : 1;
      }
      // (an fp16 copy of the output rows for the projection behind the attention: asked for by a replayed plan and by a window's prompt evaluation)
      const int rc = ns_hip_attn_fp32_fp16_fp16_fp32_forward_h(&a, ns::route_executing() ? ns::g_mha_out16 : nullptr, stream);
      ns::g_mha_out16_written = rc == 0 && ns::route_executing() && ns::g_mha_out16 != nullptr;
      return rc;
    }
  }
  // ---- the context split over workgroups: head sizes 64 / 128 / 256, from two 128-key ranges on ----
  static const bool no_split = getenv("NS_MHA_NO_SPLIT") != nullptr;  // diagnostics (A/B)
  if (aff.k && !(seq == 1 && (head_size == 64 || head_size == 128 || head_size == 256) && n_ctx >= seq_all)) {
    ns::set_error("mha_f32: a moving context length needs the context-split kernel (decode step, head size 64 / 128 / 256)");
    return -1;
  }
  const int nsplit = aff.k ? (n_ctx + ns::kMhaKS - 1) / ns::kMhaKS : (seq_all + ns::kMhaKS - 1) / ns::kMhaKS;
  const size_t rows = size_t(batch) * seq * heads;
  if ((!no_split || aff.k) && (nsplit >= 2 || aff.k) && (head_size == 64 || head_size == 128 || head_size == 256) && size_t(batch) * seq <= 65535 && heads <= 65535 &&
      (reinterpret_cast<uintptr_t>(dQ) & 15) == 0 && (reinterpret_cast<uintptr_t>(dK) & 15) == 0) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    // a prompt's rows (batch 1, causal) go through in chunks so that the partials stay within 64 MB of scratch whatever the prompt
    // length (2048 rows x 32 heads x 16 ranges would be 545 MB, kept for the life of the stream): rows r0 .. r0 + nr of a causal call
    // are the same call on `nr` rows over the first seq_all - seq + r0 + nr keys
    const size_t per_row = size_t(heads) * nsplit * (2 + head_size) * sizeof(float);
    int chunk = seq;
    if (batch == 1 && masked && size_t(seq) * per_row > (size_t(64) << 20)) chunk = int(std::max<size_t>(1, (size_t(64) << 20) / per_row));
    float* ws = static_cast<float*>(ns::stream_scratch(st, size_t(batch) * chunk * per_row, 24));
    if (ws) {
      for (int r0 = 0; r0 < seq; r0 += chunk) {
        const int nr = std::min(chunk, seq - r0);
        const int sa = chunk == seq ? seq_all : seq_all - seq + r0 + nr;  // keys this chunk's last row sees
        const int ns_c = aff.k ? nsplit : (sa + ns::kMhaKS - 1) / ns::kMhaKS;
        const float* q_c = dQ + size_t(r0) * heads * head_size;
        float* o_c = dO + size_t(r0) * heads * head_size;
        const dim3 grid(unsigned(ns_c), unsigned(heads), unsigned(batch * nr));
        // replayed decode step: the last live range merges inside the launch (tickets zeroed when allocated, self-resetting; sized by
        // ns_route.cpp before the capture) — NS_MHA_INLAUNCH=0: the merge launch
        static const bool inl_off = getenv("NS_MHA_INLAUNCH") && atoi(getenv("NS_MHA_INLAUNCH")) == 0;
        uint32_t* tickets = nullptr;
        if (aff.k && seq == 1 && !inl_off && size_t(batch) * heads <= 65536)
          tickets = static_cast<uint32_t*>(ns::stream_scratch_zeroed(st, 65536 * 4, 25));
        _Float16* o16 = (aff.k && seq == 1) ? static_cast<_Float16*>(ns::g_mha_out16) : nullptr;
        if (tickets) {
          if (head_size == 64)
            hipLaunchKernelGGL(ns::mha_f32_split_kernel<4>, grid, dim3(256), 0, st, q_c, dK, dV, o_c, ws, ns_c, nr, sa, heads, heads_kv, n_ctx, scale, masked, aff.k, int(aff.delta), tickets, o16);
          else if (head_size == 128)
            hipLaunchKernelGGL(ns::mha_f32_split_kernel<8>, grid, dim3(256), 0, st, q_c, dK, dV, o_c, ws, ns_c, nr, sa, heads, heads_kv, n_ctx, scale, masked, aff.k, int(aff.delta), tickets, o16);
          else
            hipLaunchKernelGGL(ns::mha_f32_split_kernel<16>, grid, dim3(256), 0, st, q_c, dK, dV, o_c, ws, ns_c, nr, sa, heads, heads_kv, n_ctx, scale, masked, aff.k, int(aff.delta), tickets, o16);
          continue;
        }
        if (head_size == 64)
          hipLaunchKernelGGL(ns::mha_f32_split_kernel<4>, grid, dim3(256), 0, st, q_c, dK, dV, o_c, ws, ns_c, nr, sa, heads, heads_kv, n_ctx, scale, masked, aff.k, int(aff.delta), static_cast<uint32_t*>(nullptr), static_cast<_Float16*>(nullptr));
        else if (head_size == 128)
          hipLaunchKernelGGL(ns::mha_f32_split_kernel<8>, grid, dim3(256), 0, st, q_c, dK, dV, o_c, ws, ns_c, nr, sa, heads, heads_kv, n_ctx, scale, masked, aff.k, int(aff.delta), static_cast<uint32_t*>(nullptr), static_cast<_Float16*>(nullptr));
        else
          hipLaunchKernelGGL(ns::mha_f32_split_kernel<16>, grid, dim3(256), 0, st, q_c, dK, dV, o_c, ws, ns_c, nr, sa, heads, heads_kv, n_ctx, scale, masked, aff.k, int(aff.delta), static_cast<uint32_t*>(nullptr), static_cast<_Float16*>(nullptr));
        if (ns_c > 1 || aff.k)
          hipLaunchKernelGGL(ns::mha_f32_merge_kernel, dim3(unsigned(size_t(batch) * nr * heads)), dim3(unsigned(head_size)), 0, st, ws, o_c, ns_c, head_size,
                             sa, aff.k, int(aff.delta), (aff.k && seq == 1) ? static_cast<_Float16*>(ns::g_mha_out16) : nullptr);
      }
      if (hipGetLastError() != hipSuccess) {
        ns::set_error("mha_f32: launch failed");
        return -1;
      }
      return 0;
    }
  }
  if (aff.k) {
    ns::set_error("mha_f32: the moving-length form needs 16-byte aligned q / k and scratch for the partials");
    return -1;
  }
  const size_t lds = (size_t((seq_all + 3) & ~3) + 256) * sizeof(float);
  if (lds > 160 * 1024) {
    ns::set_error("mha_f32: context too long for the device-layout attention kernel");
    return -1;
  }
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(ns::mha_f32_kernel),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (attr != hipSuccess && lds > 64 * 1024) {
    ns::set_error("mha_f32: cannot raise the LDS limit");
    return -1;
  }
  hipLaunchKernelGGL(ns::mha_f32_kernel, dim3(heads, seq, batch), dim3(256), lds, static_cast<hipStream_t>(stream), dQ, dK, dV, dO, seq,
                     seq_all, heads, heads_kv, head_size, n_ctx, scale, masked);
  if (hipGetLastError() != hipSuccess) {
    ns::set_error("mha_f32: launch failed");
    return -1;
  }
  return 0;
}

}  // extern "C"
