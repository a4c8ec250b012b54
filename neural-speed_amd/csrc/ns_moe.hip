// ns_moe.hip — expert-indexed matmul with the routing ON THE DEVICE
//
// Reference: ne_compute_forward_mul_mat_id_q_f32_bestla (/root/reference/neural_speed/core/ne_layers.c:7783-7916): for
// every token row t of src1, `row_id = ids[t][id]` selects one of `n_as` expert blobs and
//       dst[t] = src1[t] . W[row_id]
// is computed by one `bestla_f32f32_forward(row, expert blob, dst row, 1, ...)` call per (token, expert) after the HOST
// has read the ids and grouped the rows (`matrix_rows`, :7855-7866).  That host round trip is what a device-resident,
// graph-captured decode step cannot afford (the ids are the output of the router's top-k on the device), so here the
// kernel itself reads the id: grid = (column tiles, token rows), each workgroup looks up its row's expert in a device
// table of weight descriptors and streams that expert's tile.  The host-pointer surface keeps working unchanged through
// bestla_f32f32_forward; this entry is its device twin.
//
// Numerics: the default semantics of this library — fp16-rounded activations, w = (code - zp) * scale (or LUT[code] *
// scale for the 4-bit float types), fp32 accumulation — evaluated with VALU FMAs: a token row is an M = 1 product, the
// matrix cores have nothing to add and the kernel is bound by the expert's weight stream.  First version: plain
// streaming loop with four records in flight per wave, not tuned like smallm_kernel.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ns_bestla.h"
#include "ns_common.h"
#include "ns_dev.h"

namespace ns {
namespace {

constexpr int kMoeWaves = 8, kMoeThreads = kMoeWaves * 64;

struct MoeExpert {  // one row of the device table
  const uint8_t* codes;
  const uint8_t* scales;
  const int8_t* zps;
};

struct MoeParams {
  const MoeExpert* table;
  int n_as;
  const int32_t* ids;
  int ids_stride, id;
  uint32_t qstride, sstride, zstride;
  int ksteps, kstep_len, kind;
  int sps, srows, srow_mul, srow_shift;
  uint32_t scale_dt;
  int asym;
  int n, k, m;
  const float* a;
  int lda;
  float* c;
  int ldc;
  int epilogue;
  const float* d;
  int ldd;
  float lut[16];  // 4-bit float types: code -> value
};

// ---- prefill size (round 5): the token rows grouped by expert, ONE tiled GEMM per expert over its rows ----------------------------
// The per-row kernels below let every token stream its whole expert: 2048 tokens x 2 experts of a Mixtral layer read 2048 x 29 MB per
// matrix.  The reference groups the rows on the host too (`matrix_rows`, ne_layers.c:7855-7866) and then still calls one
// bestla_f32f32_forward per row; here the id column comes back to the host once (m ints), the rows are gathered in expert order as the
// fp16 operand the tiled GEMM multiplies anyway, every expert's rows are ONE gemm3_kernel launch, and a scatter kernel applies the
// epilogue per token row.  Not capturable (the host reads the ids): a capturing stream keeps the per-row form.
__global__ void moe_ids_column_kernel(const int32_t* __restrict__ ids, int ids_stride, int id, int m, int32_t* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < m) out[t] = ids[size_t(t) * ids_stride + id];
}
__global__ void moe_gather_rows_kernel(const float* __restrict__ a, int lda, const int32_t* __restrict__ perm, int rows, int k, int kpad,
                                       _Float16* __restrict__ out) {
  const size_t gid = size_t(blockIdx.x) * blockDim.x + threadIdx.x;  // one thread per 8 columns
  const int per_row = kpad / 8;
  if (gid >= size_t(rows) * per_row) return;
  const int j = int(gid / per_row), c = int(gid % per_row) * 8;
  const float* src = a + size_t(perm[j]) * lda + c;
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  h8 v;
#pragma unroll
  for (int e = 0; e < 8; e++) v[e] = c + e < k ? (_Float16)src[e] : (_Float16)0.f;
  *reinterpret_cast<h8*>(out + size_t(j) * kpad + c) = v;
}
__global__ void moe_scatter_epilogue_kernel(const float* __restrict__ cg, const int32_t* __restrict__ perm, const int32_t* __restrict__ valid, int rows,
                                            int n, float* __restrict__ c, int ldc, int epilogue, const float* __restrict__ d, int ldd) {
  const size_t gid = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (gid >= size_t(rows) * n) return;
  const int j = int(gid / n), col = int(gid % n);
  const int t = perm[j];
  float v = valid[j] ? cg[size_t(j) * n + col] : 0.f;  // an id outside the group: a zero product (the per-row kernels do the same)
  const float dv = d ? d[size_t(t) * ldd + col] : 0.f;
  switch (epilogue) {
    case 1: v = v + dv; break;
    case 2: v = v * dv; break;
    case 3: v = epi_gelu(v + dv); break;
    case 4: v = epi_gelu(v); break;
    case 5: v = epi_silu(v); break;
    default: break;
  }
  c[size_t(t) * ldc + col] = v;
}

template <int KIND>  // WK_INT4, WK_INT8 or WK_F4
__global__ __launch_bounds__(kMoeThreads) void moe_gemv_kernel(const MoeParams p) {
  constexpr int NJ = KIND == WK_INT8 ? 2 : 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char moe_smem[];
  __shared__ float red[kMoeWaves][16];
  __shared__ float lut_s[16];
  _Float16* a_lds = reinterpret_cast<_Float16*>(moe_smem);  // the token row, rounded to fp16 like every default kernel
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, nn = l & 15, cslot = l >> 4;
  const int tile = blockIdx.x, row = blockIdx.y;
  const int32_t e_id = p.ids[size_t(row) * p.ids_stride + p.id];
  const bool valid = e_id >= 0 && e_id < p.n_as;  // the reference asserts this (:7860); here the row is zeroed
  const MoeExpert ex = p.table[valid ? e_id : 0];
  const int kpad = p.ksteps * p.kstep_len;
  for (int i = tid; i < kpad; i += kMoeThreads) a_lds[i] = i < p.k ? (_Float16)p.a[size_t(row) * p.lda + i] : (_Float16)0.f;
  if (KIND == WK_F4 && tid < 16) lut_s[tid] = p.lut[tid];
  __syncthreads();
  const int sbytes = p.scale_dt == DT_F32 ? 4 : 2;
  const int rec_sbytes = p.sps * sbytes;
  struct Corr {
    uint32_t s[4];
    uint32_t z;
  };
  auto fetch_corr = [&](int s) {
    Corr c{{0, 0, 0, 0}, 0};
    const uint32_t srow = uint32_t(s * p.srow_mul) >> p.srow_shift;
    const size_t crow = size_t(tile) * p.srows + srow;
    const uint8_t* sp = ex.scales + crow * p.sstride + size_t(nn) * rec_sbytes;
    if (rec_sbytes == 16) {
      const uint4v v = *reinterpret_cast<const uint4v*>(sp);
      c.s[0] = v.x, c.s[1] = v.y, c.s[2] = v.z, c.s[3] = v.w;
    } else if (rec_sbytes == 8) {
      const uint2 v = *reinterpret_cast<const uint2*>(sp);
      c.s[0] = v.x, c.s[1] = v.y;
    } else if (rec_sbytes == 4) {
      c.s[0] = *reinterpret_cast<const uint32_t*>(sp);
    } else {
      c.s[0] = *reinterpret_cast<const uint16_t*>(sp);
    }
    if (p.asym) {
      const int8_t* zp = ex.zps + crow * p.zstride + nn * p.sps;
      for (int e = 0; e < p.sps; e++) c.z |= uint32_t(uint8_t(zp[e])) << (8 * e);
    }
    return c;
  };
  float acc = 0.f;
  auto consume = [&](const uint4v& rec, const Corr& cr, int s) {
    const uint32_t xw[4] = {rec.x, rec.y, rec.z, rec.w};
#pragma unroll
    for (int j = 0; j < NJ; j++) {
      const int e = (j * p.sps) / NJ;
      auto pick = [&](int i) { return i == 0 ? cr.s[0] : (i == 1 ? cr.s[1] : (i == 2 ? cr.s[2] : cr.s[3])); };
      float sb;
      if (p.scale_dt == DT_F32) {
        sb = __builtin_bit_cast(float, pick(e));
      } else {
        const uint32_t h = (pick(e >> 1) >> (16 * (e & 1))) & 0xffffu;
        sb = p.scale_dt == DT_BF16 ? __builtin_bit_cast(float, h << 16) : f16_bits_to_f32(h);
      }
      const float zb = p.asym ? float(int(int8_t((cr.z >> (8 * e)) & 0xffu))) : 0.f;
      const half8_t av = *reinterpret_cast<const half8_t*>(a_lds + s * p.kstep_len + 32 * j + 8 * cslot);
      float dot = 0.f;
#pragma unroll
      for (int i = 0; i < 8; i++) {
        float wv;
        if constexpr (KIND == WK_INT8) {
          const uint32_t word = i < 4 ? xw[2 * j] : xw[2 * j + 1];
          wv = float(int(int8_t((word >> (8 * (i & 3))) & 255u))) - zb;
        } else {
          const uint32_t code = (xw[j] >> (((i & 1) << 4) + ((i >> 1) << 2))) & 15u;  // nibble i at bit {0,16,4,20,...}
          if constexpr (KIND == WK_F4)
            wv = lut_s[code];
          else
            wv = float(int(code) - 8) - zb;
        }
        dot = fmaf(float(av[i]), wv, dot);
      }
      acc = fmaf(dot, sb, acc);
    }
  };
  auto load_rec = [&](int s) {
    return *reinterpret_cast<const uint4v*>(ex.codes + (size_t(tile) * p.ksteps + s) * p.qstride + l * 16);
  };
  if (valid) {
    for (int s = w; s < p.ksteps; s += 4 * kMoeWaves) {
      uint4v rec[4];
      Corr cr[4];
#pragma unroll
      for (int t = 0; t < 4; t++) {
        const int st = s + t * kMoeWaves;
        const bool on = st < p.ksteps;
        rec[t] = on ? load_rec(st) : uint4v{0, 0, 0, 0};
        cr[t] = on ? fetch_corr(st) : Corr{{0, 0, 0, 0}, 0};
      }
#pragma unroll
      for (int t = 0; t < 4; t++)
        if (s + t * kMoeWaves < p.ksteps) consume(rec[t], cr[t], s + t * kMoeWaves);
    }
  }
  acc += __shfl_xor(acc, 16);
  acc += __shfl_xor(acc, 32);
  if (l < 16) red[w][l] = acc;
  __syncthreads();
  if (tid < 16) {
    const int col = tile * 16 + tid;
    if (col < p.n) {
      float v = 0.f;
#pragma unroll
      for (int ww = 0; ww < kMoeWaves; ww++) v += red[ww][tid];
      const float dv = p.d ? p.d[size_t(row) * p.ldd + col] : 0.f;
      switch (p.epilogue) {
        case 1: v = v + dv; break;
        case 2: v = v * dv; break;
        case 3: v = epi_gelu(v + dv); break;
        case 4: v = epi_gelu(v); break;
        case 5: v = epi_silu(v); break;
        default: break;
      }
      p.c[size_t(row) * p.ldc + col] = v;
    }
  }
}

}  // namespace
}  // namespace ns

struct ns_expert_group {
  std::vector<const ns_weight*> experts;
  ns::MoeExpert* table = nullptr;  // device
};

using namespace ns;

namespace ns {
// 0 = done, -1 = error (set_error), 1 = not taken (the caller's per-row path serves the call)
static int mul_mat_id_grouped(const float* dA, const int32_t* dIds, int ids_stride, int id, const ns_expert_group* g, float* dC, int m, int lda,
                              int ldc, int epilogue, const float* dD, int ldd, hipStream_t st) {
  const ns_weight* w0 = g->experts[0];
  const int n_as = int(g->experts.size()), n = w0->n, k = w0->k;
  if (w0->kind == WK_F8 || w0->shuf || (k % 64) != 0) return 1;
  const int kpad = k;
  // scratch: ids column + permutation + validity (3 m ints), gathered fp16 rows [m + 1][k], raw products [m + 1][n] (one padding row: a
  // single-row group is launched with two rows — the tiled kernel's minimum — and its second row belongs to the NEXT group, launched later)
  int32_t* ints = static_cast<int32_t*>(stream_scratch(st, size_t(3) * m * 4, 30));
  _Float16* ag = static_cast<_Float16*>(stream_scratch(st, size_t(m + 1) * kpad * 2, 31));
  float* cg = static_cast<float*>(stream_scratch(st, size_t(m + 1) * n * 4, 32));
  if (!ints || !ag || !cg) return 1;
  int32_t *d_col = ints, *d_perm = ints + m, *d_valid = ints + 2 * size_t(m);
  hipLaunchKernelGGL(moe_ids_column_kernel, dim3((m + 255) / 256), dim3(256), 0, st, dIds, ids_stride, id, m, d_col);
  std::vector<int32_t> col(m), perm(m), valid(m);
  if (hipMemcpyAsync(col.data(), d_col, size_t(m) * 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
    set_error("mul_mat_id: reading the expert ids failed");
    return -1;
  }
  // counting sort by expert; ids outside the group go last (their rows are written as zero products)
  std::vector<int> count(n_as + 1, 0), off(n_as + 2, 0);
  for (int t = 0; t < m; t++) count[col[t] >= 0 && col[t] < n_as ? col[t] : n_as]++;
  for (int e = 0; e <= n_as; e++) off[e + 1] = off[e] + count[e];
  std::vector<int> fill(off.begin(), off.end() - 1);
  for (int t = 0; t < m; t++) {
    const int e = col[t] >= 0 && col[t] < n_as ? col[t] : n_as;
    perm[fill[e]] = t, valid[fill[e]] = e < n_as ? 1 : 0;
    fill[e]++;
  }
  if (hipMemcpyAsync(d_perm, perm.data(), size_t(m) * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
      hipMemcpyAsync(d_valid, valid.data(), size_t(m) * 4, hipMemcpyHostToDevice, st) != hipSuccess) {
    set_error("mul_mat_id: uploading the row order failed");
    return -1;
  }
  {
    const size_t units = size_t(m) * (kpad / 8);
    hipLaunchKernelGGL(moe_gather_rows_kernel, dim3(unsigned((units + 255) / 256)), dim3(256), 0, st, dA, lda, d_perm, m, k, kpad, ag);
    // the padding row: zeros (read by the last group when it has a single row)
    if (hipMemsetAsync(ag + size_t(m) * kpad, 0, size_t(kpad) * 2, st) != hipSuccess) return -1;
  }
  for (int e = 0; e < n_as; e++) {
    const int r = count[e];
    if (!r) continue;
    SmallMArgs a{};
    a.a = nullptr, a.a16 = ag + size_t(off[e]) * kpad, a.lda = kpad, a.m = r < 2 ? 2 : r, a.ldc = n, a.nseg = 1;
    a.seg[0] = {g->experts[e], cg + size_t(off[e]) * n, nullptr};
    a.epilogue = NS_EPI_NONE;
    const hipError_t er = launch_gemm2(a, st);
    if (er == hipErrorNotSupported) {
      // (nothing has been written to dC yet: the per-row path can still serve the whole call)
      (void)hipStreamSynchronize(st);  // the host vectors above are being read by the two uploads
      return 1;
    }
    if (er != hipSuccess) {
      set_error(std::string("mul_mat_id (grouped): ") + hipGetErrorString(er));
      (void)hipStreamSynchronize(st);
      return -1;
    }
  }
  {
    const size_t total = size_t(m) * n;
    hipLaunchKernelGGL(moe_scatter_epilogue_kernel, dim3(unsigned((total + 255) / 256)), dim3(256), 0, st, cg, d_perm, d_valid, m, n, dC, ldc, epilogue, dD, ldd);
  }
  // the uploads read host vectors that die with this frame
  if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) {
    set_error("mul_mat_id (grouped): launch failed");
    return -1;
  }
  return 0;
}
}  // namespace ns
using ns::mul_mat_id_grouped;

extern "C" {

ns_expert_group* ns_hip_expert_group_create(const ns_weight* const* experts, int n_as) {
  if (!experts || n_as <= 0 || n_as > 4096) {
    set_error("expert group: need 1..4096 experts");
    return nullptr;
  }
  const ns_weight* w0 = experts[0];
  std::vector<MoeExpert> host(n_as);
  for (int i = 0; i < n_as; i++) {
    const ns_weight* w = experts[i];
    if (!w || !w0) {
      set_error("expert group: null weight");
      return nullptr;
    }
    // one kernel launch serves every expert: shapes and formats must agree (they do in every MoE checkpoint)
    if (w->n != w0->n || w->k != w0->k || w->kind != w0->kind || w->qtype != w0->qtype || w->blocksize != w0->blocksize ||
        w->scale_dt != w0->scale_dt || w->asym != w0->asym || w->qstride != w0->qstride || w->sstride != w0->sstride ||
        w->zstride != w0->zstride || w->shuf || w->device != w0->device) {
      set_error("expert group: experts differ in shape / format (or carry an activation shuffle)");
      return nullptr;
    }
    host[i] = {reinterpret_cast<const uint8_t*>(w->codes), static_cast<const uint8_t*>(w->scales), w->zps};
  }
  if (w0->kind != WK_INT4 && w0->kind != WK_INT8 && w0->kind != WK_F4) {
    set_error("expert group: S1..S8 and the 4-bit float types are supported (fp8 experts are not)");
    return nullptr;
  }
  ns_expert_group* g = new ns_expert_group;
  g->experts.assign(experts, experts + n_as);
  if (hipMalloc((void**)&g->table, sizeof(MoeExpert) * n_as) != hipSuccess ||
      hipMemcpy(g->table, host.data(), sizeof(MoeExpert) * n_as, hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipGetLastError();
    set_error("expert group: device table allocation failed");
    if (g->table) (void)hipFree(g->table);
    delete g;
    return nullptr;
  }
  return g;
}

void ns_hip_expert_group_free(ns_expert_group* g) {
  if (!g) return;
  if (g->table) (void)hipFree(g->table);
  delete g;
}

int ns_hip_mul_mat_id(const float* dA, const int32_t* dIds, int ids_stride, int id, const ns_expert_group* g, float* dC,
                      int m, int lda, int ldc, int epilogue, const float* dD, int ldd, void* stream) {
  if (!g || !dA || !dIds || !dC || m <= 0 || id < 0 || id >= ids_stride) {
    set_error("mul_mat_id: bad argument");
    return -1;
  }
  if (m > 65535) {
    set_error("mul_mat_id: at most 65535 token rows per call");
    return -1;
  }
  const ns_weight* w = g->experts[0];
  MoeParams p{};
  p.table = g->table;
  p.n_as = int(g->experts.size());
  p.ids = dIds, p.ids_stride = ids_stride, p.id = id;
  p.qstride = w->qstride, p.sstride = w->sstride, p.zstride = w->zstride;
  p.ksteps = w->ksteps, p.kstep_len = w->kstep_len, p.kind = w->kind;
  p.sps = w->sps, p.srows = w->srows;
  if (!srow_params(w, &p.srow_mul, &p.srow_shift)) {
    set_error("mul_mat_id: group size not expressible for this weight");
    return -1;
  }
  p.scale_dt = w->scale_dt;
  p.asym = w->asym ? 1 : 0;
  p.n = w->n, p.k = w->k, p.m = m;
  p.a = dA, p.lda = lda, p.c = dC, p.ldc = ldc;
  p.epilogue = epilogue, p.d = dD, p.ldd = ldd;
  for (int i = 0; i < 16; i++) p.lut[i] = w->lutf[i];
  const size_t lds = size_t(w->ksteps) * w->kstep_len * 2;
  if (lds > 60 * 1024) {
    set_error("mul_mat_id: K beyond 30720 is not supported by this first version");
    return -1;
  }
  hipStream_t st = (hipStream_t)stream;
  // prefill-sized calls: rows grouped by expert, one tiled GEMM per expert (see moe_gather_rows_kernel)
  static const int grouped_from = getenv("NS_MOE_GROUPED_ROWS") ? atoi(getenv("NS_MOE_GROUPED_ROWS")) : 32;
  if (m >= grouped_from && grouped_from > 0) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone;
    (void)hipGetLastError();
    if (!capturing) {
      const int rc = mul_mat_id_grouped(dA, dIds, ids_stride, id, g, dC, m, lda, ldc, epilogue, dD, ldd, st);
      if (rc <= 0) return rc;  // 0 done, -1 failed; 1: outside the tiled kernel's envelope — the per-row kernels below
    }
  }
  // decode-sized calls: one launch of the decode kernel (ns_gemv.hip, XV = 4) per token row — LDS-DMA rings, MFMA on the raw
  // codes, the expert's base pointer picked from the table on the device — instead of this file's VALU loop (Mixtral shapes,
  // one token, 8 x {14336 x 4096, 4096 x 14336} int4: 169 us per MoE FFN layer with the loop, profiles/r04s_moe_mixtral_m1.json)
  static const int moe_gemv_rows = getenv("NS_MOE_GEMV_ROWS") ? atoi(getenv("NS_MOE_GEMV_ROWS")) : 8;
  if (m <= moe_gemv_rows && w->single_span) {
    bool all = true;
    for (int t = 0; t < m && all; t++) {
      MoeRoute route{g->table, dIds + size_t(t) * ids_stride + id, int(g->experts.size())};
      SmallMArgs a{};
      a.a = dA + size_t(t) * lda, a.lda = lda, a.m = 1, a.ldc = ldc, a.nseg = 1;
      a.seg[0] = {w, dC + size_t(t) * ldc, nullptr};
      a.epilogue = epilogue, a.d = dD ? dD + size_t(t) * ldd : nullptr, a.ldd = ldd;
      a.moe = &route;
      const hipError_t e = launch_gemv(a, st);
      if (e == hipErrorNotSupported && t == 0) {
        all = false;  // outside that kernel's envelope: the loop below serves the whole call
      } else if (e != hipSuccess) {
        set_error(std::string("mul_mat_id launch: ") + hipGetErrorString(e));
        return -1;
      }
    }
    if (all) return 0;
  }
  const dim3 grid(w->ntiles, m), block(kMoeThreads);
  if (w->kind == WK_INT8)
    hipLaunchKernelGGL(moe_gemv_kernel<WK_INT8>, grid, block, lds, st, p);
  else if (w->kind == WK_F4)
    hipLaunchKernelGGL(moe_gemv_kernel<WK_F4>, grid, block, lds, st, p);
  else
    hipLaunchKernelGGL(moe_gemv_kernel<WK_INT4>, grid, block, lds, st, p);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error(std::string("mul_mat_id launch: ") + hipGetErrorString(e));
    return -1;
  }
  return 0;
}

}  // extern "C"
