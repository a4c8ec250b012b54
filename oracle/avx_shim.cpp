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
#include "kernel_avx2.h"

#include <cstdint>

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

int avx2_quantize_fp_u8_colblock(int row, int col, const float* src, int ld_src, uint8_t* dst, int ld_dst, float* scales,
                                 int ld_scale, uint8_t* zps, int blocksize, float* blkreduce) {
  return (int)bestla::kernel::avx2::quantize_fp_u8_colblock<float>(row, col, src, ld_src, dst, ld_dst, scales, ld_scale, zps,
                                                                   blocksize, blkreduce);
}

}  // extern "C"
