// ns_i8ref.hip — the reference's int8-COMPUTE semantics on the device (opt-in numerics mode, NS_COMPUTE_REF_INT8)
//
// BesTLA blobs packed for an integer compute core (AVX512_VNNI / AMX_INT8 ..., the default `compute_dtype=int8`) are
// run by the reference with DYNAMICALLY QUANTIZED activations: `ActivationKBlockQuantize` (bestla_prologue_a.h:133-154,
// kernel quantize_fp_u8_colblock, kernel_ref.h:1824-1883) turns every (row, k-block) of A into u8 codes + scale + zero
// point, and the GEMV / GEMM accumulates
//       acc[m][n] += (a_q - zp_a[m][kb]) * (code - zp_b[kb][n]) * (scale_a[m][kb] * scale_b[kb][n])
// (gemv_4bit_u8s8_fp32, kernel_ref.h:2371-2429; GEMM form ref_kblock_int8, bestla/ut/bestla_gemm.cpp:159-190).  The
// default kernels of this library compute the same weights against fp16 activations instead (closer to the fp32 result
// than the reference's own int8 path).  This file is the other choice: bit-exact activation
// quantization (aquant_u8_kernel, ns_quant.hip) followed by an exact integer dot product per 32-deep slice and the fp32
// scale product per k-block, so that a caller who needs the CPU int8 path's numbers (regression baselines, token-exact
// comparisons) gets them from the GPU.  Differences from the scalar reference are fp32 summation order only.
//
// Kernel: one 256-thread workgroup per 16-column tile of the streaming layout (ns_common.h: ns_weight), wave w takes
// k-steps w, w+4, ...; lane (nn, c) owns column nn and the eight k of slot c in every 32-deep slice, exactly the bytes
// the MFMA kernels read.  Up to four rows of A per pass (weights are re-read for more rows: this is a numerics mode, the
// streaming loop is VALU-light enough to stay HBM-bound at decode sizes but it is not tuned).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "ns_common.h"
#include "ns_dev.h"

namespace ns {
namespace {

constexpr int kI8Rows = 4;

struct I8RefParams {
  const uint8_t* codes;
  const uint8_t* scales;
  const int8_t* zps;
  uint32_t qstride, sstride, zstride;
  int ksteps, kstep_len, nj;  // 128 / 4 (4-bit containers) or 64 / 2 (8-bit)
  int sps, srows, srow_mul, srow_shift;
  uint32_t scale_dt;
  int asym;
  int blocksize, nblk;  // k-block of BOTH quantizations (weights and activations), blocks per row
  int n, k, m;
  const uint8_t* aq;    // [m][k] u8 activation codes
  const float* ascale;  // [m][nblk]
  const uint8_t* azp;   // [m][nblk]
  float* c;
  _Float16* c16;
  int ldc;
  int epilogue;
  const float* d;
  int ldd;
};

__device__ __forceinline__ float load_scale(const uint8_t* p, uint32_t dt) {
  if (dt == DT_F32) return *reinterpret_cast<const float*>(p);
  const uint32_t h = *reinterpret_cast<const uint16_t*>(p);
  if (dt == DT_BF16) return __builtin_bit_cast(float, h << 16);
  return f16_bits_to_f32(h);
}

__global__ __launch_bounds__(256) void i8ref_kernel(const I8RefParams p) {
  __shared__ float red[4][kI8Rows][16];
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, nn = l & 15, cslot = l >> 4;
  const int tile = blockIdx.x;
  const bool four = p.nj == 4;
  for (int r0 = 0; r0 < p.m; r0 += kI8Rows) {
    const int rows = min(kI8Rows, p.m - r0);
    float acc[kI8Rows] = {0.f, 0.f, 0.f, 0.f};
    for (int s = w; s < p.ksteps; s += 4) {
      const uint4v rec = *reinterpret_cast<const uint4v*>(p.codes + (size_t(tile) * p.ksteps + s) * p.qstride + l * 16);
      const uint32_t xw[4] = {rec.x, rec.y, rec.z, rec.w};
      const uint32_t srow = uint32_t(s * p.srow_mul) >> p.srow_shift;
      const size_t crow = size_t(tile) * p.srows + srow;
      for (int j = 0; j < p.nj; j++) {
        const int k0 = s * p.kstep_len + 32 * j + 8 * cslot;
        if (k0 >= p.k) continue;
        const int e = (j * p.sps) / p.nj;
        const int sbytes = p.scale_dt == DT_F32 ? 4 : 2;
        const float sb = load_scale(p.scales + crow * p.sstride + (size_t(nn) * p.sps + e) * sbytes, p.scale_dt);
        const int zb = p.asym ? int(p.zps[crow * p.zstride + nn * p.sps + e]) : 0;
        // the eight weight codes of this lane's slot, as signed integers q = stored - bias
        int q[8];
        if (four) {
          const uint32_t x = xw[j];  // nibble i sits at bit {0,16,4,20,8,24,12,28}[i]
#pragma unroll
          for (int i = 0; i < 8; i++) q[i] = int((x >> (((i & 1) << 4) + ((i >> 1) << 2))) & 15u) - 8;
        } else {
          const uint32_t lo = xw[2 * j], hi = xw[2 * j + 1];
#pragma unroll
          for (int i = 0; i < 4; i++) q[i] = int(int8_t((lo >> (8 * i)) & 255u)), q[4 + i] = int(int8_t((hi >> (8 * i)) & 255u));  // raw int8
        }
        const int kb = k0 / p.blocksize;  // a 32-deep slice never straddles a k-block (blocksize % 32 == 0)
        const int kn = min(8, p.k - k0);
        for (int r = 0; r < rows; r++) {
          const size_t row = size_t(r0 + r);
          const uint8_t* ap = p.aq + row * p.k + k0;
          const int za = int(p.azp[row * p.nblk + kb]);
          int isum = 0;
#pragma unroll
          for (int i = 0; i < 8; i++)
            if (i < kn) isum += (int(ap[i]) - za) * (q[i] - zb);
          acc[r] += float(isum) * (p.ascale[row * p.nblk + kb] * sb);
        }
      }
    }
    // the four k-slots of a column live in lanes nn, nn+16, nn+32, nn+48; then the four waves through LDS
#pragma unroll
    for (int r = 0; r < kI8Rows; r++) {
      acc[r] += __shfl_xor(acc[r], 16);
      acc[r] += __shfl_xor(acc[r], 32);
    }
    if (l < 16)
#pragma unroll
      for (int r = 0; r < kI8Rows; r++) red[w][r][l] = acc[r];
    __syncthreads();
    if (tid < 16 * rows) {
      const int r = tid >> 4, col = tile * 16 + (tid & 15);
      if (col < p.n) {
        float v = red[0][r][tid & 15] + red[1][r][tid & 15] + red[2][r][tid & 15] + red[3][r][tid & 15];
        const size_t row = size_t(r0 + r);
        const float dv = p.d ? p.d[row * p.ldd + col] : 0.f;
        switch (p.epilogue) {
          case 1: v = v + dv; break;
          case 2: v = v * dv; break;
          case 3: v = epi_gelu(v + dv); break;
          case 4: v = epi_gelu(v); break;
          case 5: v = epi_silu(v); break;
          default: break;
        }
        p.c[row * p.ldc + col] = v;
        if (p.c16) p.c16[row * p.ldc + col] = (_Float16)v;
      }
    }
    __syncthreads();
  }
}

}  // namespace

bool i8ref_supported(const ns_weight* w) {
  return (w->kind == WK_INT4 || w->kind == WK_INT8) && w->blocksize > 0 && (w->blocksize % 32 == 0 || w->blocksize >= w->k);
}

// C[m][n] = epi(dequant-free int8-compute product of quantize_u8(A) and the integer weight `w`)
hipError_t launch_i8ref(const float* a, int lda, const ns_weight* w, float* c, void* c16, int m, int ldc, int epilogue,
                        const float* d, int ldd, hipStream_t st) {
  if (!i8ref_supported(w)) return hipErrorNotSupported;
  const int bs = w->blocksize >= w->k ? w->k : w->blocksize;
  const int nblk = (w->k + bs - 1) / bs;
  // scratch: u8 codes [m][k] | fp32 scales [m][nblk] | u8 zero points [m][nblk]
  const size_t aq_bytes = (size_t(m) * w->k + 15) & ~size_t(15);
  const size_t sc_bytes = size_t(m) * nblk * 4;
  uint8_t* base = static_cast<uint8_t*>(stream_scratch(st, aq_bytes + sc_bytes + size_t(m) * nblk + 16, 4));
  if (!base) return hipErrorOutOfMemory;
  uint8_t* aq = base;
  float* as = reinterpret_cast<float*>(base + aq_bytes);
  uint8_t* az = base + aq_bytes + sc_bytes;
  hipError_t e = launch_aquant_u8(m, w->k, a, lda, aq, w->k, as, nblk, az, bs, nullptr, st);
  if (e != hipSuccess) return e;
  I8RefParams p{};
  p.codes = reinterpret_cast<const uint8_t*>(w->codes);
  p.scales = static_cast<const uint8_t*>(w->scales);
  p.zps = w->zps;
  p.qstride = w->qstride;
  p.sstride = w->sstride;
  p.zstride = w->zstride;
  p.ksteps = w->ksteps;
  p.kstep_len = w->kstep_len;
  p.nj = w->kind == WK_INT8 ? 2 : 4;
  p.sps = w->sps;
  p.srows = w->srows;
  if (!srow_params(w, &p.srow_mul, &p.srow_shift)) return hipErrorNotSupported;
  p.scale_dt = w->scale_dt;
  p.asym = w->asym ? 1 : 0;
  p.blocksize = bs;
  p.nblk = nblk;
  p.n = w->n, p.k = w->k, p.m = m;
  p.aq = aq, p.ascale = as, p.azp = az;
  p.c = c;
  p.c16 = static_cast<_Float16*>(c16);
  p.ldc = ldc;
  p.epilogue = epilogue;
  p.d = d;
  p.ldd = ldd;
  hipLaunchKernelGGL(i8ref_kernel, dim3(w->ntiles), dim3(256), 0, st, p);
  return hipGetLastError();
}

}  // namespace ns
