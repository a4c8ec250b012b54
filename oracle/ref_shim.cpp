/*
 * ref_shim.cpp — thin extern "C" wrapper around the REAL reference scalar kernels
 * (bestla/bestla/kernel_ref.h, compiled from /root/reference where it lies; no reference source is copied).
 * Built by oracle/Makefile into oracle/_ref/libkernel_ref.so (git-ignored).  Test infrastructure only: it is
 * used to validate oracle/ns_oracle.cpp and to mint tests/golden/ fixtures.  kernel_ref.h is the only BesTLA
 * header besides bestla.h / bestla_utils.h that compiles without xbyak.
 */
#include "kernel_ref.h"

#include <cstdint>
#include <cstring>

using namespace bestla;               // NOLINT
namespace kr = bestla::kernel::ref;   // NOLINT

extern "C" {

int ref_quantize_int(const float* src, int8_t* dst, int row, int col, int ld_src, int ld_dst, float* scales,
                     int8_t* zps, int blocksize, uint32_t qtype) {
  return (int)kr::quantize_f32_sign_int_rowblock(src, dst, row, col, ld_src, ld_dst, scales, zps, blocksize,
                                                 (BTLA_DTYPE)qtype);
}

int ref_quantize_f4(const float* src, int8_t* dst, int row, int col, int ld_src, int ld_dst, float* scales,
                    int blocksize, uint32_t f4type) {
  switch ((BTLA_DTYPE)f4type) {
    case BTLA_DTYPE::F4_NF4:
      return (int)kr::quantize_f32_f4_rowblock<BTLA_DTYPE::F4_NF4>(src, dst, row, col, ld_src, ld_dst, scales, blocksize);
    case BTLA_DTYPE::F4_BNB:
      return (int)kr::quantize_f32_f4_rowblock<BTLA_DTYPE::F4_BNB>(src, dst, row, col, ld_src, ld_dst, scales, blocksize);
    case BTLA_DTYPE::F4_E2M1:
      return (int)kr::quantize_f32_f4_rowblock<BTLA_DTYPE::F4_E2M1>(src, dst, row, col, ld_src, ld_dst, scales, blocksize);
    default:
      return -1;
  }
}

float ref_f4_unpack(uint32_t f4type, int code) {
  switch ((BTLA_DTYPE)f4type) {
    case BTLA_DTYPE::F4_NF4:
      return kr::f4_unpack<BTLA_DTYPE::F4_NF4>((int8_t)code);
    case BTLA_DTYPE::F4_BNB:
      return kr::f4_unpack<BTLA_DTYPE::F4_BNB>((int8_t)code);
    default:
      return kr::f4_unpack<BTLA_DTYPE::F4_E2M1>((int8_t)code);
  }
}

int ref_f4_quantize(uint32_t f4type, float x) {
  switch ((BTLA_DTYPE)f4type) {
    case BTLA_DTYPE::F4_NF4:
      return kr::f4_quantize<BTLA_DTYPE::F4_NF4>(x);
    case BTLA_DTYPE::F4_BNB:
      return kr::f4_quantize<BTLA_DTYPE::F4_BNB>(x);
    default:
      return kr::f4_quantize<BTLA_DTYPE::F4_E2M1>(x);
  }
}

float ref_lut(uint32_t f4type, int idx) {
  switch ((BTLA_DTYPE)f4type) {
    case BTLA_DTYPE::F4_NF4:
      return nf4_dequant_fp32_LUT[idx];
    case BTLA_DTYPE::F4_BNB:
      return fp4_bnb_dequant_fp32_LUT[idx];
    default:
      return fp4_e2m1_dequant_fp32_LUT[idx];
  }
}

void ref_padding_interleave(const int8_t* src, int8_t* dst, int row, int col, int rowpad, int colpad, int src_step,
                            int dst_step, int ntile, int rowpack) {
  kr::padding_interleave(src, dst, row, col, rowpad, colpad, src_step, dst_step, ntile, rowpack);
}

/* plane pointers are computed by the caller exactly as bestla_prologue_b.h:512-547 does */
void ref_compress_s8_s4(const int8_t* s, uint8_t* d, size_t n) { kr::compress_s8_s4(s, (utils::int4x2*)d, n); }
void ref_compress_f4(const int8_t* s, uint8_t* d, size_t n) { kr::compress_f4(s, (utils::f4x2*)d, n); }
void ref_compress_7bit(const int8_t* s, uint8_t* b4, uint8_t* b2, uint8_t* b1, size_t n) {
  kr::compress_7bit(s, (utils::bit4x2*)b4, (utils::bit2x4*)b2, (utils::bit1x8*)b1, n);
}
void ref_compress_6bit(const int8_t* s, uint8_t* b4, uint8_t* b2, size_t n) {
  kr::compress_6bit(s, (utils::bit4x2*)b4, (utils::bit2x4*)b2, n);
}
void ref_compress_5bit(const int8_t* s, uint8_t* b4, uint8_t* b1, size_t n) {
  kr::compress_5bit(s, (utils::bit4x2*)b4, (utils::bit1x8*)b1, n);
}
void ref_compress_3bit(const int8_t* s, uint8_t* b2, uint8_t* b1, size_t n) {
  kr::compress_3bit(s, (utils::bit2x4*)b2, (utils::bit1x8*)b1, n);
}
void ref_compress_2bit(const int8_t* s, uint8_t* b2, size_t n) { kr::compress_2bit(s, (utils::bit2x4*)b2, n); }
void ref_compress_1bit(const int8_t* s, uint8_t* b1, size_t n) { kr::compress_1bit(s, (utils::bit1x8*)b1, n); }

void ref_decompress_s4_s8(uint8_t* s, int8_t* d, size_t n) { kr::decompress_s4_s8((utils::int4x2*)s, d, n, nullptr, 0); }
void ref_decompress_s2_s8(uint8_t* s, int8_t* d, size_t n) { kr::decompress_s2_s8((utils::bit2x4*)s, d, n, nullptr, 0); }
void ref_decompress_s1_s8(uint8_t* b1, int8_t* d, size_t n) { kr::decompress_s1_s8((utils::bit1x8*)b1, d, n, nullptr, 0); }
void ref_decompress_s3_s8(uint8_t* b2, uint8_t* b1, int8_t* d, size_t n) {
  kr::decompress_s3_s8((utils::bit2x4*)b2, (utils::bit1x8*)b1, d, n, nullptr, 0);
}
void ref_decompress_s5_s8(uint8_t* b4, uint8_t* b1, int8_t* d, size_t n) {
  kr::decompress_s5_s8((utils::bit4x2*)b4, (utils::bit1x8*)b1, d, n, nullptr, 0);
}
void ref_decompress_s6_s8(uint8_t* b4, uint8_t* b2, int8_t* d, size_t n) {
  kr::decompress_s6_s8((utils::bit4x2*)b4, (utils::bit2x4*)b2, d, n, nullptr, 0);
}
void ref_decompress_s7_s8(uint8_t* b4, uint8_t* b2, uint8_t* b1, int8_t* d, size_t n) {
  kr::decompress_s7_s8((utils::bit4x2*)b4, (utils::bit2x4*)b2, (utils::bit1x8*)b1, d, n, nullptr, 0);
}

/* tile dequant: decompress_kblock_s4_fp<PackRow,48,float> (kernel_ref.h:1112-1127).  scales dtype F32 or BF16. */
int ref_decompress_kblock_s4_fp(int packrow, uint8_t* src, float* dst, int row, void* scales, uint32_t sdtype,
                                int8_t* zps, int k_offset, int n_offset, int blocksize, int ldzp) {
  int8_t tmp[48 * 4];
  switch (packrow) {
    case 1:
      return (int)kr::decompress_kblock_s4_fp<1, 48, float>((utils::int4x2*)src, dst, row, 48, scales,
                                                            (BTLA_DTYPE)sdtype, zps, k_offset, n_offset, blocksize,
                                                            ldzp, tmp, sizeof(tmp));
    case 2:
      return (int)kr::decompress_kblock_s4_fp<2, 48, float>((utils::int4x2*)src, dst, row, 48, scales,
                                                            (BTLA_DTYPE)sdtype, zps, k_offset, n_offset, blocksize,
                                                            ldzp, tmp, sizeof(tmp));
    case 4:
      return (int)kr::decompress_kblock_s4_fp<4, 48, float>((utils::int4x2*)src, dst, row, 48, scales,
                                                            (BTLA_DTYPE)sdtype, zps, k_offset, n_offset, blocksize,
                                                            ldzp, tmp, sizeof(tmp));
    default:
      return -1;
  }
}

int ref_decompress_kblock_s8_fp(int packrow, int8_t* src, float* dst, int row, void* scales, uint32_t sdtype,
                                int8_t* zps, int k_offset, int n_offset, int blocksize, int ldzp) {
  switch (packrow) {
    case 1:
      return (int)kr::decompress_kblock_s8_fp<1, 48, float>(src, dst, row, 48, scales, (BTLA_DTYPE)sdtype, zps,
                                                            k_offset, n_offset, blocksize, ldzp, nullptr, 0);
    case 4:
      return (int)kr::decompress_kblock_s8_fp<4, 48, float>(src, dst, row, 48, scales, (BTLA_DTYPE)sdtype, zps,
                                                            k_offset, n_offset, blocksize, ldzp, nullptr, 0);
    default:
      return -1;
  }
}

/* NF4/FP4 tile dequant: decompress_kblock_f4_fp<F4_T,float,PackRow,float> (kernel_ref.h:1456-1478), fp32 scales */
int ref_decompress_kblock_f4_fp(uint32_t f4type, int packrow, uint8_t* src, float* dst, int row, int col, float* scales,
                                int k_offset, int kblock, int npad) {
#define NS_F4(T, P)                                                                                                 \
  return (int)kr::decompress_kblock_f4_fp<T, float, P, float>((utils::f4x2*)src, dst, row, col, col, col, scales,   \
                                                              k_offset, kblock, npad, nullptr, 0)
  if ((BTLA_DTYPE)f4type == BTLA_DTYPE::F4_NF4) {
    if (packrow == 1) NS_F4(BTLA_DTYPE::F4_NF4, 1);
    if (packrow == 2) NS_F4(BTLA_DTYPE::F4_NF4, 2);
  } else if ((BTLA_DTYPE)f4type == BTLA_DTYPE::F4_BNB) {
    if (packrow == 1) NS_F4(BTLA_DTYPE::F4_BNB, 1);
  } else {
    if (packrow == 1) NS_F4(BTLA_DTYPE::F4_E2M1, 1);
  }
#undef NS_F4
  return -1;
}

/* fp8 weights: quantize_f32_f8_rowblock_mxscale<F8_T> (kernel_ref.h:1763-1799), f8_to_fp32 (:984-1002),
 * decompress_kblock_f8_fp<float, PackRow, scale type> (:1004-1026) */
int ref_quantize_f8(const float* src, int8_t* dst, int row, int col, int ld_src, int ld_dst, float* scales, int blocksize,
                    uint32_t f8type, uint32_t stype) {
  if ((BTLA_DTYPE)f8type == BTLA_DTYPE::F8_E4M3)
    return (int)kr::quantize_f32_f8_rowblock_mxscale<BTLA_DTYPE::F8_E4M3>(src, dst, row, col, ld_src, ld_dst, scales,
                                                                          blocksize, (BTLA_DTYPE)stype);
  if ((BTLA_DTYPE)f8type == BTLA_DTYPE::F8_E5M2)
    return (int)kr::quantize_f32_f8_rowblock_mxscale<BTLA_DTYPE::F8_E5M2>(src, dst, row, col, ld_src, ld_dst, scales,
                                                                          blocksize, (BTLA_DTYPE)stype);
  return -1;
}
float ref_f8_to_f32(uint32_t f8type, int code) { return kr::f8_to_fp32(utils::f8((int8_t)code), (BTLA_DTYPE)f8type); }
int ref_decompress_kblock_f8_fp(uint32_t f8type, int packrow, int8_t* src, float* dst, int row, int col, void* scales,
                                int scale_is_e8m0, int k_offset, int kblock, int npad) {
#define NS_F8(P)                                                                                                      \
  do {                                                                                                                \
    if (scale_is_e8m0)                                                                                                \
      return (int)kr::decompress_kblock_f8_fp<float, P, utils::f8>((utils::f8*)src, dst, row, col, col, col,          \
                                                                   (utils::f8*)scales, k_offset, kblock, npad,        \
                                                                   (BTLA_DTYPE)f8type);                               \
    return (int)kr::decompress_kblock_f8_fp<float, P, float>((utils::f8*)src, dst, row, col, col, col, (float*)scales, \
                                                             k_offset, kblock, npad, (BTLA_DTYPE)f8type);             \
  } while (0)
  if (packrow == 1) NS_F8(1);
  if (packrow == 2) NS_F8(2);
#undef NS_F8
  return -1;
}

/* activation shuffle of g_idx blobs: kernel_ref.h:28-37 */
int ref_shuffle_activation(float* src, float* dst, int m, int k, int m_offset, int k_offset, int* indices, int src_stride,
                           int dst_stride) {
  return (int)kr::shuffle_activation<float>(src, dst, m, k, m_offset, k_offset, indices, src_stride, dst_stride);
}

void ref_row_reduce_sum_bf16(const float* src, int ldsrc, int row, int col, uint16_t* reduce) {
  kr::row_reduce_sum<utils::bf16>(src, ldsrc, row, col, (utils::bf16*)reduce);
}

int ref_quantize_fp_u8_colblock(int row, int col, const float* src, int ld_src, uint8_t* dst, int ld_dst, float* scales,
                                int ld_scale, uint8_t* zps, int blocksize, float* blkreduce) {
  return (int)kr::quantize_fp_u8_colblock<float>(row, col, src, ld_src, dst, ld_dst, scales, ld_scale, zps, blocksize,
                                                 blkreduce);
}

/* gemv_4bit_fp32_fp32<ScaleT,48,MTILE> over ONE 48-column tile in the PACK_ROW=1 layout (kernel_ref.h:2489-2531) */
int ref_gemv_4bit_fp32_fp32(const float* A, int lda, uint8_t* b4, void* scales, int scale_is_bf16, int8_t* zps,
                            int ldzp, float* C, int ldc, int k, int blocksize, int mtile) {
#define NS_GEMV(ST, M)                                                      \
  {                                                                         \
    utils::GemvParamB<ST> B{b4, nullptr, nullptr, (ST*)scales, zps, 4, ldzp, k}; \
    return (int)kr::gemv_4bit_fp32_fp32<ST, 48, M>(A, lda, B, C, ldc, k, blocksize, nullptr, 0); \
  }
  if (scale_is_bf16) {
    if (mtile == 1) NS_GEMV(utils::bf16, 1);
    if (mtile == 2) NS_GEMV(utils::bf16, 2);
    if (mtile == 4) NS_GEMV(utils::bf16, 4);
  } else {
    if (mtile == 1) NS_GEMV(float, 1);
    if (mtile == 2) NS_GEMV(float, 2);
    if (mtile == 4) NS_GEMV(float, 4);
  }
#undef NS_GEMV
  return -1;
}

/* gemv_4bit_u8s8_fp32<float,48,1> over ONE 48-column tile in the PACK_ROW=4 layout (kernel_ref.h:2371-2429) */
int ref_gemv_4bit_u8s8_fp32(uint8_t* aq, float* ascale, uint8_t* azp, int lda, int ldazp, uint8_t* b4, float* scales,
                            int8_t* zps, int ldzp, float* C, int ldc, int k, int blocksize) {
  utils::GemvParamA A{aq, ascale, azp, lda, ldazp};
  utils::GemvParamB<float> B{b4, nullptr, nullptr, scales, zps, 4, ldzp, k};
  return (int)kr::gemv_4bit_u8s8_fp32<float, 48, 1>(A, B, C, ldc, k, blocksize, nullptr, 0);
}

uint16_t ref_f32_to_bf16(float v) { return utils::bf16(v).x; }
float ref_bf16_to_f32(uint16_t v) { return utils::bf16::from_bin(v).tofloat(); }
uint16_t ref_f32_to_f16(float v) { return utils::fp16(v).x; }
float ref_f16_to_f32(uint16_t v) {
  utils::fp16 h;
  h.x = v;
  return static_cast<float>(h);
}
int ref_cast_f32_s8(float v) { return utils::cast<float, int8_t>(v); }
int ref_cast_f32_u8(float v) { return utils::cast<float, uint8_t>(v); }

float ref_postop(float x, int op) {
  return kr::postop(x, op == 0 ? BTLA_ELTWISEOP::GELU : BTLA_ELTWISEOP::SWISH, nullptr);
}

/* the reference's own DQ8_BNB code map (bestla_utils.h:794-...), for the entry-by-entry pin of the oracle's construction */
const float* ref_dq8_lut(void) { return bestla::dq8_bnb_LUT; }

}  // extern "C"
