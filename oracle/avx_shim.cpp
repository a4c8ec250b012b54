/*
 * avx_shim.cpp — TEST INFRASTRUCTURE.  extern "C" wrappers around the reference's AVX512 / AVX2 quantizer kernels
 * (bestla/bestla/kernel_avx512f.h, kernel_avx2.h, compiled from /root/reference where they lie) — what
 * kernel_wrapper.h:546-603 dispatches to on an AVX512 / AVX2 host for the F4 weight quantizer and the u8 activation
 * quantizer, while the integer weight quantizer always takes the scalar kernel (:540-543).  The product and the oracle
 * follow the SCALAR kernels (kernel_ref.h); this library lets tests/test_oracle_vs_avx.py show where the vector kernels
 * agree bit for bit and where they round differently.  Their one non-buildable include, kernel_jit.h (xbyak), is not used
 * by these functions: oracle/Makefile (target avxref) satisfies it with a generated stand-in.
 * Built into oracle/_ref/libkernel_avx_ref.so with -mavx512f ...: only loadable work on a CPU that has the ISA (the
 * test checks /proc/cpuinfo first).
 */
#include "kernel_avx512f.h"
#include "kernel_avx512_vnni.h"
#include "kernel_avx2.h"

#include <cstdint>
#include <cstring>
#ifdef _OPENMP
#include <omp.h>
#endif

using namespace bestla;  // NOLINT

extern "C" {

int avx512_quantize_f4(const float* src, int8_t* dst, int row, int col, int ld_src, int ld_dst, float* scales, int blocksize,
                       uint32_t f4type) {
  namespace k = bestla::kernel::avx512f;
  switch ((BTLA_DTYPE)f4type) {
    case BTLA_DTYPE::F4_NF4:
      return (int)k::quantize_f32_f4_rowblock<BTLA_DTYPE::F4_NF4>(src, dst, row, col, ld_src, ld_dst, scales, blocksize);
    case BTLA_DTYPE::F4_BNB:
      return (int)k::quantize_f32_f4_rowblock<BTLA_DTYPE::F4_BNB>(src, dst, row, col, ld_src, ld_dst, scales, blocksize);
    case BTLA_DTYPE::F4_E2M1:
      return (int)k::quantize_f32_f4_rowblock<BTLA_DTYPE::F4_E2M1>(src, dst, row, col, ld_src, ld_dst, scales, blocksize);
    default:
      return -1;
  }
}

int avx512_quantize_fp_u8_colblock(int row, int col, const float* src, int ld_src, uint8_t* dst, int ld_dst, float* scales,
                                   int ld_scale, uint8_t* zps, int blocksize, float* blkreduce) {
  return (int)bestla::kernel::avx512f::quantize_fp_u8_colblock<float>(row, col, src, ld_src, dst, ld_dst, scales, ld_scale, zps,
                                                                      blocksize, blkreduce);
}

/* fp8 weight tile -> fp32 with the scale applied (what the AVX512 product path dequantizes with, kernel_avx512f.h:653-720):
 * ONE call over a whole tile, the kernel itself steps the scale row every `kblock` packed rows */
int avx512_decompress_kblock_f8_fp(uint32_t f8type, int packrow, int8_t* src, float* dst, int row, int col, void* scales,
                                   int scale_is_e8m0, int k_offset, int kblock, int npad) {
  namespace k = bestla::kernel::avx512f;
#define NS_F8(P)                                                                                                          \
  do {                                                                                                                    \
    if (scale_is_e8m0)                                                                                                    \
      return (int)k::decompress_kblock_f8_fp<true, float, P, utils::f8>((utils::f8*)src, dst, row, col, col, col,          \
                                                                        (utils::f8*)scales, k_offset, kblock, npad,        \
                                                                        (BTLA_DTYPE)f8type);                               \
    return (int)k::decompress_kblock_f8_fp<true, float, P, float>((utils::f8*)src, dst, row, col, col, col, (float*)scales, \
                                                                  k_offset, kblock, npad, (BTLA_DTYPE)f8type);             \
  } while (0)
  if (packrow == 1) NS_F8(1);
  if (packrow == 2) NS_F8(2);
#undef NS_F8
  return -1;
}

int avx2_quantize_fp_u8_colblock(int row, int col, const float* src, int ld_src, uint8_t* dst, int ld_dst, float* scales,
                                 int ld_scale, uint8_t* zps, int blocksize, float* blkreduce) {
  return (int)bestla::kernel::avx2::quantize_fp_u8_colblock<float>(row, col, src, ld_src, dst, ld_dst, scales, ld_scale, zps,
                                                                   blocksize, blkreduce);
}

/* The reference's decode hot loop on an AVX512-VNNI host (kernel_wrapper.h:1456-1466 dispatches here for M <= 4): the
 * activation row is quantized per k-block by avx512f::quantize_fp_u8_colblock, then avx512f::vnni::gemv_4bit_u8s8_fp32
 * <ScaleT, 48, 1> produces one 48-column tile per call from the packed blob sections (int4, NTILE 48, PACK_ROW 4 — the
 * tAVX512_VNNI_KBlock layout).  The reference spreads the tiles over its thread pool (bestla_wrapper.h, behind the JIT
 * headers); here a plain OpenMP loop does.  q: packed codes, scales: [nblk][cstep] fp32 or bf16, zps: [nblk][cstep] or
 * NULL.  scratch: k + 5 * nblk bytes-ish, see below.  Returns 0 on success. */
}  // extern "C"

template <typename ScaleT>
static int gemv_tiles(const uint8_t* aq, const float* as, const uint8_t* azp, int nblk, const uint8_t* q, const void* scales,
                      const int8_t* zps, int cstep, int kpad, int n, int k, int blocksize, float* c, int nthreads) {
  const int ntiles = (n + 47) / 48;
  int rc = 0;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(static)
  for (int t = 0; t < ntiles; t++) {
    alignas(64) int8_t tmp[16384];
    alignas(64) float ct[48];
    utils::GemvParamA A{const_cast<uint8_t*>(aq), const_cast<float*>(as), const_cast<uint8_t*>(azp), k, nblk};
    utils::GemvParamB<ScaleT> B{const_cast<uint8_t*>(q) + size_t(t) * 48 * kpad / 2, nullptr, nullptr,
                                const_cast<ScaleT*>(static_cast<const ScaleT*>(scales)) + size_t(t) * 48,
                                zps ? const_cast<int8_t*>(zps) + size_t(t) * 48 : nullptr, 4, cstep, kpad};
    const int r = (int)bestla::kernel::avx512f::vnni::gemv_4bit_u8s8_fp32<ScaleT, 48, 1>(A, B, ct, 48, k, blocksize, tmp, sizeof(tmp));
    if (r) rc = r;
    const int cols = n - t * 48 < 48 ? n - t * 48 : 48;
    memcpy(c + size_t(t) * 48, ct, sizeof(float) * cols);
  }
  return rc;
}

extern "C" {

int avx512vnni_gemv_4bit_u8s8(const float* a, const uint8_t* q, const void* scales, int scale_is_bf16, const int8_t* zps, int cstep,
                              int kpad, int n, int k, int blocksize, float* c, int nthreads, uint8_t* scratch) {
  const int nblk = (k + blocksize - 1) / blocksize;
  uint8_t* aq = scratch;                                              /* [k] */
  float* as = reinterpret_cast<float*>(scratch + ((k + 63) & ~63));  /* [nblk] */
  uint8_t* azp = reinterpret_cast<uint8_t*>(as + nblk);               /* [nblk] */
  int rc = (int)bestla::kernel::avx512f::quantize_fp_u8_colblock<float>(1, k, a, k, aq, k, as, nblk, azp, blocksize, nullptr);
  if (rc) return rc;
  return scale_is_bf16 ? gemv_tiles<utils::bf16>(aq, as, azp, nblk, q, scales, zps, cstep, kpad, n, k, blocksize, c, nthreads)
                       : gemv_tiles<float>(aq, as, azp, nblk, q, scales, zps, cstep, kpad, n, k, blocksize, c, nthreads);
}

}  // extern "C"
