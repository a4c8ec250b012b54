/*
 * ns_oracle.cpp — CPU ORACLE (test infrastructure, NOT product code).  See ns_oracle.h for the rules.
 *
 * Scalar restatement of the BesTLA weight-only-quant path.  Written from the reference's behaviour
 * (file:line cited per function, paths relative to /root/reference); pinned against the reference's own
 * kernel_ref.h through oracle/_ref/libkernel_ref.so (tests/test_oracle_vs_ref.py) and frozen golden vectors.
 */
#include "ns_oracle.h"

#include <algorithm>
#include <cassert>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

// ---------------------------------------------------------------- dtype helpers
inline int dt_bits(uint32_t t) { return int(t & 0xff); }        // bestla_utils.h:408-410
inline int dt_is_int(uint32_t t) { return ((t >> 8) & 0xff) == 1; }  // bestla_utils.h:414-416
inline int dt_bytes(uint32_t t) { return dt_bits(t) >> 3; }     // bestla_utils.h:418-421
inline bool dt_is_f8(uint32_t t) { return t == NSO_F8_E4M3 || t == NSO_F8_E5M2; }
inline size_t updiv(size_t a, size_t b) { return (a + b - 1) / b; }
inline size_t padto(size_t a, size_t b) { return updiv(a, b) * b; }

// CompType / CoreAttr bit fields — bestla_gemm.h:22-123
constexpr int COMP_FP32 = 0, COMP_BF16_FP32 = 0x011, COMP_FP16_FP32 = 0x022, COMP_INT8_US_FP32 = 0x034;
constexpr int ISA_AVX2 = 2, ISA_AVX_VNNI = 3, ISA_AVX512F = 4, ISA_AVX512BW = 5, ISA_AVX512_VNNI = 6, ISA_AMX_BF16 = 9,
              ISA_AMX_INT8 = 10, ISA_AMX_FP16 = 11;
struct CoreRow {
  int ntile, packrow, ktile, comp, isa;
};
// neural_speed/core/layers/bestla_defs.h:36-54 with the tile constants of bestla_gemm.h (code::*::NTILE/KTILE/PackRow)
const CoreRow kCores[NSO_CORE_COUNT] = {
    {24, 1, 1, COMP_FP32, ISA_AVX2},                 // tAVX2
    {48, 1, 1, COMP_FP32, ISA_AVX512F},              // tAVX512F
    {48, 2, 32, COMP_BF16_FP32, ISA_AMX_BF16},       // tAMX_BF16
    {48, 2, 32, COMP_FP16_FP32, ISA_AMX_FP16},       // tAMX_FP16
    {48, 4, 4, COMP_INT8_US_FP32, ISA_AVX512_VNNI},  // tAVX512_VNNI_KBlock
    {48, 4, 4, COMP_INT8_US_FP32, ISA_AVX512BW},     // tAVX512BW_KBlock
    {24, 4, 4, COMP_INT8_US_FP32, ISA_AVX_VNNI},     // tAVX_VNNI_KBlock
    {24, 4, 4, COMP_INT8_US_FP32, ISA_AVX2},         // tAVX2_VNNI_KBlock
    {48, 4, 64, COMP_INT8_US_FP32, ISA_AMX_INT8},    // tAMX_INT8_US_KBlock
};
inline uint64_t make_core_id(const CoreRow& c) {  // bestla_gemm.h:91-94
  return uint64_t(c.ntile) | (uint64_t(c.packrow) << 8) | (uint64_t(c.comp) << 16) | (uint64_t(c.isa) << 32);
}
inline bool comp_is_integer(int comp) {  // bestla_gemm.h:73-79: B operand type is tS8(3) or tU8(4)
  int b = (comp >> 4) & 0xf;
  return b == 3 || b == 4;
}

// ---------------------------------------------------------------- casts — bestla_utils.h:502-538
inline int8_t cast_f32_s8(float v) {
  if (std::isnan(v)) return 0;  // reference is UB here; x86 g++ -O2 was observed to produce 0
  v = roundf(v);
  v = std::min(v, 127.f);
  v = std::max(v, -128.f);
  return static_cast<int8_t>(v);
}
inline uint8_t cast_f32_u8(float v) {
  if (std::isnan(v)) return 0;
  v += 0.5f;
  v = std::min(v, 255.f);
  v = std::max(v, 0.f);
  return static_cast<uint8_t>(v);
}
inline int cast_f32_int(float v) {
  if (std::isnan(v)) return INT32_MIN;  // cvttss2si "integer indefinite"
  float r = roundf(v);
  if (r >= 2147483648.f || r < -2147483648.f) return INT32_MIN;
  return int(r);
}

// f4 tables: value LUT (bestla_utils.h:749-789 == the decision trees kernel_ref.h:1209-1230,1300-1366) and the
// quantisation thresholds of kernel_ref.h:1234-1298,1373-1413 flattened into "count thresholds below x".
const float kLutNF4[16] = {0.f,
                           -0.6961928009986877f,
                           -0.5250730514526367f,
                           -0.39491748809814453f,
                           -0.28444138169288635f,
                           -0.18477343022823334f,
                           -0.09105003625154495f,
                           -1.f,
                           0.07958029955625534f,
                           0.16093020141124725f,
                           0.24611230194568634f,
                           0.33791524171829224f,
                           0.44070982933044434f,
                           0.5626170039176941f,
                           0.7229568362236023f,
                           1.0f};
const float kLutBNB[8] = {0.f, 5.208333333e-03f, 0.66666667f, 1.f, 0.33333333f, 0.5f, 0.16666667f, 0.25f};
const float kLutE2M1[8] = {0.f, 0.010416666666666666f, 0.16666666666666666f, 0.25f, 0.3333333333333333f,
                           0.5f, 0.6666666666666666f, 1.f};
const float kThrNF4[15] = {-0.8480964004993439f, -0.6106329262256622f,  -0.4599952697753906f,  -0.33967943489551544f,
                           -0.23460740596055984f, -0.13791173323988914f, -0.045525018125772476f, 0.03979014977812767f,
                           0.1202552504837513f,  0.2035212516784668f,   0.2920137718319893f,   0.3893125355243683f,
                           0.5016634166240692f,  0.6427869200706482f,   0.8614784181118011f};
const int8_t kCodeNF4[16] = {7, 1, 2, 3, 4, 5, 6, 0, 8, 9, 10, 11, 12, 13, 14, 15};
const float kThrBNB[7] = {0.00260417f, 0.0859375f, 0.20833333f, 0.29166667f, 0.4166667f, 0.583333f, 0.8333333f};
const int8_t kCodeBNB[8] = {0, 1, 6, 7, 4, 5, 2, 3};
const float kThrE2M1[7] = {0.03125f / 6, 0.53125f / 6, 1.25f / 6, 1.75f / 6, 2.5f / 6, 3.5f / 6, 5.f / 6};

inline int f4_quantize(uint32_t t, float x) {
  if (t == NSO_F4_NF4) {
    int c = 0;
    for (int i = 0; i < 15; i++) c += x > kThrNF4[i];
    return kCodeNF4[c];
  }
  int sign = x < 0 ? 8 : 0;
  float ax = fabsf(x);
  int c = 0;
  if (t == NSO_F4_BNB) {
    for (int i = 0; i < 7; i++) c += ax > kThrBNB[i];
    return kCodeBNB[c] + sign;
  }
  for (int i = 0; i < 7; i++) c += ax > kThrE2M1[i];
  return c + sign;
}
inline float f4_unpack(uint32_t t, int code) {
  code &= 15;
  if (t == NSO_F4_NF4) return kLutNF4[code];
  float mag = (t == NSO_F4_BNB) ? kLutBNB[code & 7] : kLutE2M1[code & 7];
  return (code & 8) ? -1.f * mag : mag;
}

// IEEE-754 binary16 round-to-nearest-even round trip (NOT the reference's fp16 recipe): used only to model the
// activation rounding the HIP kernels apply before their fp16 dot/MFMA units.
// ---- fp8 weights (MX-style) ------------------------------------------------------------------------------------
// f8_to_fp32 — kernel_ref.h:984-1002: sign | exponent (ebits) | mantissa (7 - ebits); value = 2^(e - (2^(ebits-1) - 1))
// * (1 + m / 2^mbits) for EVERY code: no zero, no subnormals, no inf/nan — code 0x00 decodes to 2^-7 (E4M3) / 2^-15 (E5M2).
inline int f8_ebits(uint32_t t) { return t == NSO_F8_E4M3 ? 4 : 5; }        // bestla_utils.h:414-427
inline int f8_quant_mbits(uint32_t t) { return t == NSO_F8_E4M3 ? 5 : 4; }  // bestla_utils.h:429-442
inline float f8_to_f32(uint8_t code, uint32_t t) {
  const int ebits = f8_ebits(t), mbits = 7 - ebits;
  uint32_t sign = (uint32_t(code) << 24) & 0x80000000u;
  uint32_t e = (uint32_t(code) & 0x7f) >> mbits;
  e = e - (1u << (ebits - 1)) + 1 + 127;
  uint32_t m = (uint32_t(code) << (23 - mbits)) & 0x007fffffu;
  uint32_t bits = sign | (e << 23) | m;
  float f;
  memcpy(&f, &bits, 4);
  return f;
}
// get_mxfp_maxnorm — bestla_utils.h:444-454 (E4M3: 2^8 * 1.75 = 448, E5M2: 2^15 * 1.75 = 57344)
inline float f8_maxnorm(uint32_t t) {
  const int ebits = f8_ebits(t), mant = f8_quant_mbits(t);
  double emax = std::pow(2.0, ebits - 1);
  if (t == NSO_F8_E5M2) emax -= 1;
  double mx = std::pow(2.0, emax);
  if (t != NSO_F8_E4M3)
    mx *= (std::pow(2.0, mant - 1) - 1) / std::pow(2.0, mant - 2);
  else
    mx *= 1.75;
  return float(mx);
}
// f8_mx_quantize — kernel_ref.h:1721-1761.  `scale` is the shared exponent (E8M0 scales) or the fp32 scale.
// All intermediate types follow the reference: std::log2(float) and std::floor(float) are the float overloads,
// std::pow(int, ...) promotes to double, each statement's result is rounded to float where the reference casts.
inline int8_t f8_mx_quantize(float v, float scale, uint32_t t, uint32_t stype) {
  if (stype == NSO_F8_E8M0)
    v /= float(std::pow(2, scale));
  else
    v /= scale;
  const int ebits = f8_ebits(t), quant_mantissa = f8_quant_mbits(t), store_mantissa = 7 - ebits;
  float private_exp = std::floor(std::log2(std::abs(v == 0 ? v + 1 : v)));
  const float min_exp = float(-1 * (std::pow(2, ebits - 1)) + 2);
  private_exp = private_exp < min_exp ? min_exp : private_exp;
  v = float(v / std::pow(2, private_exp) * std::pow(2, quant_mantissa - 2));
  const int sign = v > 0 ? 1 : -1;
  v = sign * float(std::floor(std::abs(v) + 0.5));
  v = float(v / std::pow(2, quant_mantissa - 2) * std::pow(2, private_exp));
  const float max_norm = f8_maxnorm(t);
  v = std::clamp(v, -1 * max_norm, max_norm);
  uint32_t bits;
  memcpy(&bits, &v, 4);
  const uint8_t store_signbit = uint8_t((bits >> 24) & 0x80);
  bits <<= 1;
  uint8_t store_ebit = uint8_t(bits >> 24);
  store_ebit = uint8_t(store_ebit - 127 + uint8_t(std::pow(2, ebits - 1)) - 1);
  if (store_ebit > 15 && t == NSO_F8_E4M3) store_ebit = 0;
  if (store_ebit > 31 && t == NSO_F8_E5M2) store_ebit = 0;
  store_ebit = uint8_t(store_ebit << store_mantissa);
  bits <<= 8;
  const int8_t ox80_shift = int8_t(-128 >> (store_mantissa - 1));
  uint8_t store_mantissabit = uint8_t(uint8_t(bits >> 24) & uint8_t(ox80_shift));
  store_mantissabit = uint8_t(store_mantissabit >> (1 + ebits));
  return int8_t(store_signbit | store_ebit | store_mantissabit);
}

inline float round_through_ieee_f16(float f) {
  uint32_t x;
  memcpy(&x, &f, 4);
  const uint32_t sign = x & 0x80000000u;
  uint32_t ax = x & 0x7fffffffu;
  if (ax >= 0x7f800000u) return f;
  float r;
  if (ax >= 0x477ff000u) {  // >= 65520 rounds to inf
    ax = 0x7f800000u;
    memcpy(&r, &ax, 4);
  } else if (ax < 0x38800000u) {  // below 2^-14: fp16 subnormal grid of 2^-24
    float a;
    memcpy(&a, &ax, 4);
    r = (a + 0.5f) - 0.5f;
  } else {
    ax += 0xfffu + ((ax >> 13) & 1u);
    ax &= ~0x1fffu;
    memcpy(&r, &ax, 4);
  }
  uint32_t rb;
  memcpy(&rb, &r, 4);
  rb |= sign;
  memcpy(&r, &rb, 4);
  return r;
}

// ---- DQ8_BNB: double-quantised scales --------------------------------------------------------------------------------------
// The code map (bestla_utils.h:794-...) is the bitsandbytes dynamic map create_dynamic_map(signed, 7 exponent bits, 8 bits) written
// with FIVE decimals; restated as that construction + that rounding and pinned entry by entry to the reference's table
// (tests/test_oracle_vs_ref.py).
const float* dq8_lut() {
  static float lut[256];
  static bool init = false;
  if (!init) {
    std::vector<double> data;
    const int max_exponent_bits = 7, non_sign_bits = 7;
    auto add_level = [&](int items, double mag) {  // means of `items` + 1 boundaries linspace(0.1, 1)
      for (int i = 0; i < items; i++) {
        const double b0 = 0.1 + (1.0 - 0.1) * double(i) / double(items), b1 = 0.1 + (1.0 - 0.1) * double(i + 1) / double(items);
        const double mean = (b0 + b1) / 2.0;
        data.push_back(mag * mean);
        data.push_back(-mag * mean);
      }
    };
    for (int i = 0; i < max_exponent_bits; i++) add_level((1 << (i + non_sign_bits - max_exponent_bits)), std::pow(10.0, -(max_exponent_bits - 1) + i));
    data.push_back(0.0);
    data.push_back(1.0);
    std::sort(data.begin(), data.end());
    for (int i = 0; i < 256; i++) {
      char buf[32];
      snprintf(buf, sizeof(buf), "%.5f", data[size_t(i)]);
      lut[i] = strtof(buf, nullptr);
    }
    init = true;
  }
  return lut;
}
// get_dq8_bnb — kernel_ref.h:1930-1950: nearest entry by binary search (ties to the lower index via `<`)
inline uint8_t dq8_code(float v) {
  const float* lut = dq8_lut();
  int left = 0, right = 255;
  while (left <= right) {
    const int mid = left + (right - left) / 2;
    if (lut[mid] == v) return uint8_t(mid);
    if (lut[mid] < v)
      left = mid + 1;
    else
      right = mid - 1;
  }
  if (right < 0) return 0;
  if (left >= 256) return 255;
  return uint8_t((v - lut[right] < lut[left] - v) ? right : left);
}
// dq8_bnb_double_quant<false> — kernel_ref.h:1951-1978, AS WRITTEN: the scales become their codes (as floats) in place; dq_buf holds a
// maximum per dq block and the offset (mean of all scales) at index updiv(size, dq_blocksize) — and a trailing partial block writes ITS
// maximum to that same index + 0 ... i.e. i / dq_blocksize + 1, on top of the offset (the reference's own indexing; sizes that are
// multiples of the block never get there)
void dq8_double_quant(float* scale, size_t scale_size, int dq_blocksize, std::vector<float>& dq_buf) {
  dq_buf.assign(updiv(scale_size, size_t(dq_blocksize)) + 1, 0.f);
  float offset = 0.f;
  for (size_t i = 0; i < scale_size; i++) offset += scale[i];
  offset /= scale_size;
  dq_buf[updiv(scale_size, size_t(dq_blocksize))] = offset;
  const size_t align_blk_size = scale_size / dq_blocksize * dq_blocksize;
  size_t i = 0;
  auto calc_scale = [&](size_t blksize) {
    float absmax = std::numeric_limits<float>::min();
    for (size_t j = 0; j < blksize; j++) {
      scale[i + j] -= offset;
      absmax = std::max(absmax, std::abs(scale[i + j]));
    }
    for (size_t j = 0; j < blksize; j++) {
      scale[i + j] /= absmax;
      scale[i + j] = dq8_code(scale[i + j]);
    }
    return absmax;
  };
  for (; i < align_blk_size; i += dq_blocksize) dq_buf[i / dq_blocksize] = calc_scale(size_t(dq_blocksize));
  if (i < scale_size) dq_buf[i / dq_blocksize + 1] = calc_scale(scale_size - i);
}

inline float scale_to_f32(const uint8_t* sbase, uint32_t stype, size_t idx) {
  if (stype == NSO_F8_E8M0)  // decompress_kblock_f8_fp, kernel_ref.h:1013-1016: scale = pow(2, int8 shared exponent)
    return float(std::pow(2, int(int8_t(sbase[idx]))));
  if (stype == NSO_F32) {
    float f;
    memcpy(&f, sbase + idx * 4, 4);
    return f;
  }
  uint16_t h;
  memcpy(&h, sbase + idx * 2, 2);
  return stype == NSO_BF16 ? nso_bf16_to_f32(h) : nso_f16_to_f32(h);
}
// scale of (k-block kb, column c) of a parsed blob.  DQ8_BNB — dq8_get_fp_scale, kernel_ref.h:1980-1992, called from
// bestla_prologue_b.h:699-707: LUT[code] * dq[(kb * N + c) / dq_blocksize] + dq[last float of the buffer]
inline float scale_at(const nso_blob_info& bi, const uint8_t* base, int kb, int c) {
  if (bi.scale_dtype != NSO_DQ8_BNB) return scale_to_f32(base + bi.scale_off, bi.scale_dtype, size_t(kb) * bi.cstep + c);
  const float* dq = reinterpret_cast<const float*>(base + bi.dq_off);
  const size_t last = size_t(bi.dq_bytes / 4 - 1);
  float block, offset;
  memcpy(&block, dq + std::min((size_t(kb) * bi.n + c) / size_t(bi.dq_blocksize), last), 4);  // (padded columns: any block, never used)
  memcpy(&offset, dq + last, 4);
  return dq8_lut()[base[bi.scale_off + size_t(kb) * bi.cstep + c]] * block + offset;
}

// ---------------------------------------------------------------- blob container — bestla_storage.h
// Serialised order (bestla_storage.h:274-283, :335-339, :821-826, :191-201, :85-95, :131-136):
//   u64 mSize | u32 mPrologueID | u64 mCoreId | i32 mNPad mKPad mN mK | u32 mDType | i32 mBlockSize mDqBlockSize
//   QBuf{u64 size, u64 offset, <offset pad>, data}
//   Correction{u32 mScaT mZpT mRedT, i32 mCStep, u64 mCSize, ScaleBuf, bool+ZpBuf, bool+RedBuf, bool+DQBuf}
//   bool+ShuffleIndices
// "offset" is the padding that makes the data start 64-byte aligned in ABSOLUTE address terms (pointer_align,
// bestla_utils.h:556-560), so the same weights serialise differently at different base addresses mod 64.
struct Layout {
  nso_blob_info bi;
  uint8_t* base;
};

struct Cursor {
  uint8_t* base;
  uint8_t* p;
  bool write;
  template <typename T>
  void field(T& v) {
    if (write)
      memcpy(p, &v, sizeof(T));
    else
      memcpy(&v, p, sizeof(T));
    p += sizeof(T);
  }
  // ObjectAlignedBuffer<64>::{serializeToBuffer,deserializeBuffer} — bestla_storage.h:85-109
  void aligned_buf(uint64_t& bytes, uint64_t& off_from_base) {
    field(bytes);
    uint64_t offset = 0;
    if (write) {
      uintptr_t after = reinterpret_cast<uintptr_t>(p + 8);
      offset = (after + 63) / 64 * 64 - after;
    }
    field(offset);
    p += offset;
    off_from_base = uint64_t(p - base);
    p += bytes;
  }
  // ObjectOptionalBuffer<64> — bestla_storage.h:131-146 (bool is one byte)
  void optional_buf(uint64_t& bytes, uint64_t& off_from_base) {
    uint8_t notempty = bytes > 0;
    field(notempty);
    if (notempty) {
      aligned_buf(bytes, off_from_base);
    } else {
      bytes = 0;
      off_from_base = 0;
    }
  }
};

size_t qbuf_bytes(size_t elts, uint32_t qtype) {  // bestla_storage.h:729-745, :844-847
  switch (qtype) {
    case NSO_S3_CLIP:
      return updiv(elts * 2, 8) + updiv(elts * 1, 8);
    case NSO_S5_CLIP:
      return updiv(elts * 4, 8) + updiv(elts * 1, 8);
    case NSO_S6_CLIP:
      return updiv(elts * 4, 8) + updiv(elts * 2, 8);
    case NSO_S7_CLIP:
      return updiv(elts * 4, 8) + updiv(elts * 2, 8) + updiv(elts * 1, 8);
    default:
      return updiv(elts * dt_bits(qtype), 8);
  }
}

// Fill every size field for a fresh storage object: createStorage + resize
// (bestla_prologue_b.h:120-127 / :1011-1017; bestla_storage.h:725-753 / :842-858; :165-181)
bool describe(nso_blob_info& bi, int n, int k, int blocksize, uint32_t qtype, uint32_t stype, int asym, int core,
              bool shuffle = false) {
  if (core < 0 || core >= NSO_CORE_COUNT) return false;
  const CoreRow& c = kCores[core];
  memset(&bi, 0, sizeof(bi));
  bool is_int = dt_is_int(qtype);
  if (!is_int && qtype != NSO_F4_NF4 && qtype != NSO_F4_BNB && qtype != NSO_F4_E2M1 && !dt_is_f8(qtype)) return false;
  if (dt_is_f8(qtype)) {  // quantize_f32_f8_rowblock_mxscale asserts E8M0 or F32 scales (kernel_ref.h:1775-1789)
    if (stype != NSO_F8_E8M0 && stype != NSO_F32) return false;
  } else if (stype == NSO_DQ8_BNB) {  // initDoubleQuantBlkSize asserts symmetric weights and a block that is a multiple of 8 (bestla_storage.h:755-759)
    if (asym || blocksize <= 0 || blocksize % 8 != 0) return false;
    if (qtype != NSO_S4_CLIP && qtype != NSO_F4_NF4) return false;  // what the reference can read back (bestla_prologue_b.h:742-751, :1298-1306)
  } else if (stype != NSO_F32 && stype != NSO_BF16 && stype != NSO_F16) {
    return false;
  }
  bi.prologue_id = is_int ? 1 : 2;
  bi.core_id = make_core_id(c);
  bi.ntile = c.ntile;
  bi.packrow = c.packrow;
  bi.comp = c.comp;
  bi.isa = c.isa;
  bi.kpad = int(padto(k, c.ktile));
  bi.npad = int(padto(n, c.ntile));
  bi.n = n;
  bi.k = k;
  bi.dtype = qtype;
  bi.blocksize = blocksize <= 0 ? bi.kpad : blocksize;
  bi.dq_blocksize = 0;
  bi.q_bytes = qbuf_bytes(size_t(bi.npad) * bi.kpad, qtype);
  int nk_scale = int(updiv(bi.kpad, bi.blocksize));
  bi.scale_dtype = stype;
  bi.cstep = bi.npad;
  bi.csize = uint64_t(nk_scale) * bi.npad;
  bi.scale_bytes = bi.csize * dt_bytes(stype);
  if (stype == NSO_DQ8_BNB) {  // the dq block is the weight block (bestla_storage.h:750, :851); maxima of nk_scale * N codes + the offset
    bi.dq_blocksize = bi.blocksize;
    bi.dq_bytes = (updiv(size_t(nk_scale) * n, size_t(bi.dq_blocksize)) + 1) * sizeof(float);
  }
  if (is_int) {
    bi.zp_dtype = NSO_S8;      // bestla_storage.h:727
    bi.red_dtype = NSO_BF16;   // bestla_gemm.cpp:229,308 (reduce dtype fixed to BF16 by every caller)
    bi.is_asym = asym != 0;
    bi.has_reduce = comp_is_integer(c.comp);  // bestla_storage.h:747-749
    bi.zp_bytes = bi.is_asym ? bi.csize * 1 : 0;
    bi.red_bytes = bi.has_reduce ? bi.csize * 2 : 0;
    if (shuffle) {  // enable_shuffle, bestla_storage.h:761-765: int[K] after the correction buffers
      bi.shuf_bytes = uint64_t(k) * sizeof(int);
      bi.has_shuffle = 1;
    }
  } else {
    if (shuffle) return false;  // BTLAGemmPackBImpl ignores shuffle indices for float weights (bestla_gemm.cpp:416-419)
    bi.zp_dtype = 0;  // EleBitsUndef, bestla_storage.h:849-850
    bi.red_dtype = 0;
  }
  // getSerializedSize chain: bestla_storage.h:44-48,:272,:306-316,:330-333,:352-356,:78-84,:123-130,:183-190,:808-812
  auto abuf = [](uint64_t bytes) { return size_t(8 + 8 + bytes + 64); };
  auto obuf = [&](uint64_t bytes) { return size_t(1 + (bytes ? abuf(bytes) : 0)); };
  size_t info = 8 + 4 + 8 + 4 * 4 + 4 + 4 + 4;
  size_t corr = 4 * 3 + 4 + 8 + abuf(bi.scale_bytes) + obuf(bi.zp_bytes) + obuf(bi.red_bytes) + obuf(bi.dq_bytes);
  size_t total = info + abuf(bi.q_bytes) + corr;
  if (is_int) total += obuf(bi.shuf_bytes);  // NFloat's final mSize leaves the (empty) shuffle flag out, :853-856
  bi.size = padto(total, 64);
  return true;
}

// walk the header either writing it (assign(), map_buf=true: bestla_storage.h:814-819) or reading it (:828-833)
bool walk(Cursor& cur, nso_blob_info& bi) {
  cur.field(bi.size);
  cur.field(bi.prologue_id);
  cur.field(bi.core_id);
  cur.field(bi.npad);
  cur.field(bi.kpad);
  cur.field(bi.n);
  cur.field(bi.k);
  cur.field(bi.dtype);
  cur.field(bi.blocksize);
  cur.field(bi.dq_blocksize);
  if (!cur.write && bi.prologue_id != 1 && bi.prologue_id != 2) return false;
  cur.aligned_buf(bi.q_bytes, bi.q_off);
  cur.field(bi.scale_dtype);
  cur.field(bi.zp_dtype);
  cur.field(bi.red_dtype);
  cur.field(bi.cstep);
  cur.field(bi.csize);
  cur.aligned_buf(bi.scale_bytes, bi.scale_off);
  cur.optional_buf(bi.zp_bytes, bi.zp_off);
  cur.optional_buf(bi.red_bytes, bi.red_off);
  cur.optional_buf(bi.dq_bytes, bi.dq_off);
  if (!cur.write && (bi.dq_bytes != 0) != (bi.scale_dtype == NSO_DQ8_BNB)) return false;
  cur.optional_buf(bi.shuf_bytes, bi.shuf_off);
  if (!cur.write) {
    bi.ntile = int(bi.core_id & 0xff);
    bi.packrow = int((bi.core_id >> 8) & 0xff);
    bi.comp = int((bi.core_id >> 16) & 0xffff);
    bi.isa = int((bi.core_id >> 32) & 0xff);
    bi.is_asym = bi.zp_bytes > 0;
    bi.has_reduce = bi.red_bytes > 0;
    bi.has_shuffle = bi.shuf_bytes > 0;
  }
  return true;
}

// location of element (k, n) in the interleaved image — follows from padding_interleave + reorderWeight
// (kernel_ref.h:39-57, bestla_prologue_b.h:490-510): [N/NTILE][KPad/PACK][NTILE][PACK]
inline size_t tiled_index(const nso_blob_info& bi, int k, int n) {
  return size_t(n / bi.ntile) * bi.ntile * bi.kpad + size_t(k / bi.packrow) * bi.ntile * bi.packrow +
         size_t(n % bi.ntile) * bi.packrow + (k % bi.packrow);
}

}  // namespace

extern "C" {

uint64_t nso_core_id(int core) { return (core < 0 || core >= NSO_CORE_COUNT) ? 0 : make_core_id(kCores[core]); }

int nso_core_attr(int core, int* ntile, int* packrow, int* ktile, int* comp, int* isa) {
  if (core < 0 || core >= NSO_CORE_COUNT) return -1;
  *ntile = kCores[core].ntile;
  *packrow = kCores[core].packrow;
  *ktile = kCores[core].ktile;
  *comp = kCores[core].comp;
  *isa = kCores[core].isa;
  return 0;
}

// bf16::fromfloat — bestla_utils.h:146-153 (round-to-nearest-even by adding 0x7fff + lsb, no NaN special case)
uint16_t nso_f32_to_bf16(float v) {
  uint32_t u;
  memcpy(&u, &v, 4);
  uint32_t lsb = (u >> 16) & 1;
  u += 0x7fff + lsb;
  return uint16_t(u >> 16);
}
float nso_bf16_to_f32(uint16_t v) {  // bestla_utils.h:127-131
  uint32_t u = uint32_t(v) << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
// fp16::operator=(float) — bestla_utils.h:184-196 (branch-free bit recipe; saturates to 0x7FFF, no inf)
uint16_t nso_f32_to_f16(float val) {
  uint32_t bits;
  memcpy(&bits, &val, 4);
  const uint32_t b = bits + 0x00001000;
  const uint32_t e = (b & 0x7F800000) >> 23;
  const uint32_t m = b & 0x007FFFFF;
  uint32_t r = (b & 0x80000000) >> 16;
  if (e > 112) r |= (((e - 112) << 10) & 0x7C00) | (m >> 13);
  if (e < 113 && e > 101) r |= (((0x007FF000 + m) >> (125 - e)) + 1) >> 1;
  if (e > 143) r |= 0x7FFF;
  return uint16_t(r);
}
// fp16::operator float — bestla_utils.h:197-207
float nso_f16_to_f32(uint16_t x) {
  const uint32_t e = (x & 0x7C00) >> 10;
  const uint32_t m = (x & 0x03FF) << 13;
  float mf = float(m);
  uint32_t mb;
  memcpy(&mb, &mf, 4);
  const uint32_t v = mb >> 23;
  uint32_t r = uint32_t(x & 0x8000) << 16;
  if (e != 0) r |= ((e + 112) << 23) | m;
  if (e == 0 && m != 0) r |= ((v - 37) << 23) | ((m << (150 - v)) & 0x007FE000);
  float f;
  memcpy(&f, &r, 4);
  return f;
}

// quantize_f32_sign_int_rowblock — kernel_ref.h:1608-1719.  One scale (and zero point) per (k-block, column).
// `abs(sum)` at :1662 resolves to the float overload, restated as fabsf (SURVEY.md §7 "Hard parts").
int nso_quantize_int_rowblock(const float* src, int8_t* dst, int row, int col, int ld_src, int ld_dst, float* scales,
                              int8_t* zero_points, int blocksize, uint32_t qtype) {
  if (!dt_is_int(qtype)) return -1;
  const int nbits = dt_bits(qtype);
  const int full = 1 << (nbits - 1);
  const int symv = full - 1;
  auto clip = [&](int s) { return std::min(std::max(s, -full), symv); };
  for (int i = 0; i < col; i++) {
    for (int j = 0; j < row; j += blocksize) {
      const int bs = std::min(blocksize, row - j);  // tail block: kernel_ref.h:1715-1716
      const size_t sidx = size_t(j / blocksize) * ld_dst + i;
      if (zero_points == nullptr) {  // :1651-1671
        float maxval = FLT_MIN, minval = FLT_MAX, absmax = 0.f;
        for (int ij = 0; ij < bs; ij++) {
          float v = src[size_t(j + ij) * ld_src + i];
          maxval = std::max(maxval, v);
          minval = std::min(minval, v);
          absmax = std::max(absmax, std::abs(v));
        }
        float nval = symv + 0.5f;
        float sum = maxval + minval;
        if (fabsf(sum) >= absmax / full) nval = sum > 0.f ? float(-full) : float(full);
        float scale = absmax / nval;
        float rscale = 1.f / scale;
        scales[sidx] = scale;
        for (int ij = 0; ij < bs; ij++)
          dst[size_t(j + ij) * ld_dst + i] = int8_t(clip(cast_f32_s8(src[size_t(j + ij) * ld_src + i] * rscale)));
      } else {  // :1673-1692
        float maxval = 0.f, minval = 0.f;
        for (int ij = 0; ij < bs; ij++) {
          float v = src[size_t(j + ij) * ld_src + i];
          maxval = std::max(maxval, v);
          minval = std::min(minval, v);
        }
        float scale = (maxval - minval) / float((1 << nbits) - 1);
        float rscale = 1.f / scale;
        scales[sidx] = scale;
        int bzp = clip(cast_f32_int((0 - minval) * rscale) - full);
        zero_points[sidx] = int8_t(bzp);
        for (int ij = 0; ij < bs; ij++) {
          int t = cast_f32_int(src[size_t(j + ij) * ld_src + i] * rscale) + bzp;
          dst[size_t(j + ij) * ld_dst + i] = int8_t(clip(t));
        }
      }
    }
  }
  return 0;
}

// quantize_f32_f4_rowblock — kernel_ref.h:1801-1822: scale = absmax (seeded with FLT_MIN), code = tree(x * (1/absmax))
int nso_quantize_f4_rowblock(const float* src, int8_t* dst, int row, int col, int ld_src, int ld_dst, float* scales,
                             int blocksize, uint32_t f4type) {
  for (int i = 0; i < col; i++) {
    for (int j = 0; j < row; j += blocksize) {
      const int bs = std::min(blocksize, row - j);
      float absmax = FLT_MIN;
      for (int ij = 0; ij < bs; ij++) absmax = std::max(absmax, std::abs(src[size_t(j + ij) * ld_src + i]));
      scales[size_t(j / blocksize) * ld_dst + i] = absmax;
      for (int ij = 0; ij < bs; ij++)
        dst[size_t(j + ij) * ld_dst + i] = int8_t(f4_quantize(f4type, src[size_t(j + ij) * ld_src + i] * (1.f / absmax)));
    }
  }
  return 0;
}
float nso_f4_unpack(uint32_t f4type, int code) { return f4_unpack(f4type, code); }
int nso_f4_quantize(uint32_t f4type, float x) { return f4_quantize(f4type, x); }

// quantize_f32_f8_rowblock_mxscale — kernel_ref.h:1763-1799.  E8M0: scale = floor(log2(absmax)) - emax (shared
// exponent, clamped at -127), stored as a float holding an integer; F32: scale = absmax / max_norm.
int nso_quantize_f8_rowblock(const float* src, int8_t* dst, int row, int col, int ld_src, int ld_dst, float* scales,
                             int blocksize, uint32_t f8type, uint32_t stype) {
  if (!dt_is_f8(f8type) || (stype != NSO_F8_E8M0 && stype != NSO_F32)) return -1;
  for (int i = 0; i < col; i++) {
    for (int j = 0; j < row; j += blocksize) {
      const int blk = std::min(blocksize, row - j);
      float scale = std::numeric_limits<float>::min();
      for (int ij = 0; ij < blk; ij++) scale = std::max(scale, std::abs(src[size_t(j + ij) * ld_src + i]));
      if (stype == NSO_F8_E8M0) {
        if (scale == 0) scale += std::abs(std::numeric_limits<float>::min());
        scale = std::floor(std::log2(scale));
        float emax = float(std::pow(2, f8_ebits(f8type) - 1));
        if (f8type == NSO_F8_E5M2) emax -= 1;
        scale -= emax;
        const float scale_max = float(std::pow(2, 7)) - 1;
        scale = scale < (-1 * scale_max) ? (-1 * scale_max) : scale;
      } else {
        scale /= f8_maxnorm(f8type);
      }
      scales[size_t(j / blocksize) * ld_dst + i] = scale;
      for (int ij = 0; ij < blk; ij++)
        dst[size_t(j + ij) * ld_dst + i] = f8_mx_quantize(src[size_t(j + ij) * ld_src + i], scale, f8type, stype);
    }
  }
  return 0;
}
float nso_f8_to_f32(uint32_t f8type, int code) { return f8_to_f32(uint8_t(code), f8type); }
int nso_f8_quantize(uint32_t f8type, uint32_t stype, float v, float scale) {
  return int(uint8_t(f8_mx_quantize(v, scale, f8type, stype)));
}

// padding_interleave — kernel_ref.h:39-57
void nso_padding_interleave(const int8_t* src, int8_t* dst, int row, int col, int rowpad, int colpad, int src_step,
                            int dst_step, int ntile, int rowpack) {
  for (int i = 0; i < rowpad; i += rowpack)
    for (int j = 0; j < colpad; j += ntile)
      for (int jj = 0; jj < ntile; jj++)
        for (int ii = 0; ii < rowpack; ii++)
          dst[size_t(i) * ntile + size_t(j) * dst_step + jj * rowpack + ii] =
              ((i + ii) < row && (j + jj) < col) ? src[size_t(i + ii) * src_step + (j + jj)] : int8_t(0);
}

size_t nso_qbytes(size_t elts, uint32_t qtype) { return qbuf_bytes(elts, qtype); }

// compress_* — kernel_ref.h:155-365 with the plane offsets of bestla_prologue_b.h:512-547.
// Bitfield order (bestla_utils.h:231-258): first member = least-significant bits.
// The 5th element of every 8 in compress_3bit / compress_1bit is read from src[j + FullRange] (kernel_ref.h:313,:355):
// j+4 for 3-bit, j+1 for 1-bit.  Restated as written.
void nso_compress(const int8_t* src, uint8_t* dst, size_t size, uint32_t qtype) {
  const int nbits = dt_bits(qtype);
  if (dt_is_f8(qtype)) {  // fp8 codes are stored as they are (packQWeight, bestla_prologue_b.h:1118-1119)
    memcpy(dst, src, size);
    return;
  }
  if (!dt_is_int(qtype)) {  // compress_f4, :167-176
    for (size_t i = 0; i < size; i += 2) dst[i / 2] = uint8_t((src[i] & 0xf) | ((src[i + 1] & 0xf) << 4));
    return;
  }
  const int full = 1 << (nbits - 1);
  if (nbits == 8) {
    memcpy(dst, src, size);
    return;
  }
  uint8_t* b4 = dst;
  uint8_t* b2 = dst;
  uint8_t* b1 = dst;
  switch (nbits) {
    case 7:  // :539-547  [4-bit plane | 2-bit plane | 1-bit plane]
      b2 = dst + size / 2;
      b1 = b2 + size / 4;
      break;
    case 6:  // :530-537
      b2 = dst + size / 2;
      break;
    case 5:  // :521-528
      b1 = dst + size / 2;
      break;
    case 3:  // :512-519  [2-bit plane | 1-bit plane]
      b1 = dst + size / 4;
      break;
    default:
      break;
  }
  const bool has4 = nbits >= 4, has2 = (nbits == 7 || nbits == 6 || nbits == 3 || nbits == 2),
             has1 = (nbits == 7 || nbits == 5 || nbits == 3 || nbits == 1);
  const int sh2 = has4 ? 4 : 0;                  // where the 2-bit field sits in the biased code
  const int sh1 = (has4 ? 4 : 0) + (has2 ? 2 : 0);  // where the 1-bit field sits
  const size_t plane_bytes[3] = {has4 ? size / 2 : 0, has2 ? size / 4 : 0, has1 ? size / 8 : 0};
  if (has4) memset(b4, 0, plane_bytes[0]);
  if (has2) memset(b2, 0, plane_bytes[1]);
  if (has1) memset(b1, 0, plane_bytes[2]);
  for (size_t i = 0; i < size; i++) {
    size_t si = i;
    if ((nbits == 3 || nbits == 1) && (i % 8) == 4) si = i - 4 + full;  // the src[j + FullRange] quirk
    int tmp = src[si] + full;
    if (has4) b4[i / 2] |= uint8_t((tmp & 0xf) << (4 * (i % 2)));
    if (has2) b2[i / 4] |= uint8_t(((tmp >> sh2) & 0x3) << (2 * (i % 4)));
    if (has1) b1[i / 8] |= uint8_t(((tmp >> sh1) & 0x1) << (i % 8));
  }
}

// decompress_s{1..7}_s8 / decompress_s4_s8 — kernel_ref.h:367-526: stored unsigned code minus 2^(b-1).
void nso_decompress(const uint8_t* src, int8_t* dst, size_t size, uint32_t qtype) {
  const int nbits = dt_bits(qtype);
  if (dt_is_f8(qtype)) {
    memcpy(dst, src, size);
    return;
  }
  if (!dt_is_int(qtype)) {
    for (size_t i = 0; i < size; i++) dst[i] = int8_t((src[i / 2] >> (4 * (i % 2))) & 0xf);
    return;
  }
  if (nbits == 8) {
    memcpy(dst, src, size);
    return;
  }
  const int full = 1 << (nbits - 1);
  const uint8_t* b4 = src;
  const uint8_t* b2 = src;
  const uint8_t* b1 = src;
  if (nbits == 7) {
    b2 = src + size / 2;
    b1 = b2 + size / 4;
  } else if (nbits == 6) {
    b2 = src + size / 2;
  } else if (nbits == 5) {
    b1 = src + size / 2;
  } else if (nbits == 3) {
    b1 = src + size / 4;
  }
  const bool has4 = nbits >= 4, has2 = (nbits == 7 || nbits == 6 || nbits == 3 || nbits == 2),
             has1 = (nbits == 7 || nbits == 5 || nbits == 3 || nbits == 1);
  const int sh2 = has4 ? 4 : 0;
  const int sh1 = (has4 ? 4 : 0) + (has2 ? 2 : 0);
  for (size_t i = 0; i < size; i++) {
    int v = 0;
    if (has4) v |= (b4[i / 2] >> (4 * (i % 2))) & 0xf;
    if (has2) v |= ((b2[i / 4] >> (2 * (i % 4))) & 0x3) << sh2;
    if (has1) v |= ((b1[i / 8] >> (i % 8)) & 0x1) << sh1;
    dst[i] = int8_t(v - full);
  }
}

size_t nso_pack_size(int n, int k, int blocksize, uint32_t qtype, uint32_t stype, int asym, int core) {
  nso_blob_info bi;
  if (!describe(bi, n, k, blocksize, qtype, stype, asym, core)) return 0;
  return bi.size;
}

const float* nso_dq8_lut(void) { return dq8_lut(); }

int nso_blob_parse(const void* blob, nso_blob_info* info) {
  Cursor cur{(uint8_t*)blob, (uint8_t*)blob, false};
  memset(info, 0, sizeof(*info));
  return walk(cur, *info) ? 0 : -1;
}

// packQWeight — bestla_prologue_b.h:378-398 (integer) / :1109-1127 (float):
//   setQuantCorrection (:244-335), reorderWeight (:490-510), compressWeight (:606-617), reduceWeight (:455-470)
static int pack_q_impl(void* blob, const int8_t* q, int ldq, const float* scales, const int8_t* zps, int n, int k,
                       int blocksize, uint32_t qtype, uint32_t stype, int asym, int core, const int* g_idx) {
  nso_blob_info bi;
  if (!describe(bi, n, k, blocksize, qtype, stype, asym, core, g_idx != nullptr)) return -1;
  uint8_t* base = (uint8_t*)blob;
  Cursor cur{base, base, true};
  if (!walk(cur, bi)) return -1;  // == stor.assign(PackedBuf), bestla_gemm.cpp:312,:414
  const int rawnk = int(updiv(k, bi.blocksize));
  const int nk = int(updiv(bi.kpad, bi.blocksize));
  // scales: converted to the storage dtype, zero-filled in padded columns / rows (:254-280)
  uint8_t* sp = base + bi.scale_off;
  const int sb = dt_bytes(stype);
  memset(sp, 0, bi.scale_bytes);
  std::vector<float> dq_codes;
  if (stype == NSO_DQ8_BNB) {  // packQWeight :378-387 / :1109-1116: the raw scale array [rawnk][N] is double-quantised first
    dq_codes.assign(scales, scales + size_t(rawnk) * n);
    std::vector<float> dq_buf;
    dq8_double_quant(dq_codes.data(), dq_codes.size(), bi.dq_blocksize, dq_buf);
    memset(base + bi.dq_off, 0, bi.dq_bytes);
    memcpy(base + bi.dq_off, dq_buf.data(), std::min<size_t>(dq_buf.size() * sizeof(float), bi.dq_bytes));
    scales = dq_codes.data();
  }
  for (int r = 0; r < std::min(rawnk, nk); r++)
    for (int c = 0; c < n; c++) {
      float s = scales[size_t(r) * n + c];
      size_t idx = size_t(r) * bi.npad + c;
      if (stype == NSO_DQ8_BNB) {  // static_cast<uint8_t>(code), bestla_prologue_b.h:313-325
        sp[idx] = uint8_t(s);
      } else if (stype == NSO_F8_E8M0) {  // static_cast<int8_t>(shared exponent), bestla_prologue_b.h:1187
        sp[idx] = uint8_t(int8_t(s));
      } else if (stype == NSO_F32) {
        memcpy(sp + idx * 4, &s, 4);
      } else {
        uint16_t h = stype == NSO_BF16 ? nso_f32_to_bf16(s) : nso_f32_to_f16(s);
        memcpy(sp + idx * sb, &h, 2);
      }
    }
  if (bi.is_asym) {  // :282-291
    int8_t* zp = (int8_t*)(base + bi.zp_off);
    memset(zp, 0, bi.zp_bytes);
    for (int r = 0; r < std::min(rawnk, nk); r++)
      for (int c = 0; c < n; c++) zp[size_t(r) * bi.npad + c] = zps[size_t(r) * n + c];
  }
  // reorder to [N/NTILE][KPad/PACK][NTILE][PACK] with zero padding, then bit-compress the whole padded image
  std::vector<int8_t> tiled(size_t(bi.npad) * bi.kpad);
  nso_padding_interleave(q, tiled.data(), k, n, bi.kpad, bi.npad, ldq, bi.kpad, bi.ntile, bi.packrow);
  nso_compress(tiled.data(), base + bi.q_off, tiled.size(), qtype);
  // reduce[kb][n] = sum over the block's rows of the DEQUANTISED weight (stored-precision scale), fp32 sequential
  // sum (row_reduce_sum, kernel_ref.h:2132-2142) -> bf16.  Padded columns are left untouched, as in the reference.
  // reduceWeight (:455-470) dequantises the blob it has just WRITTEN (unpackWeight), i.e. the codes as the bit planes hold
  // them: identical to q for every type except S1, where compress_1bit stores element 1's bit in place of element 4's in
  // every group of eight PACKED elements (kernel_ref.h:355) — so the codes are read back from the compressed image.
  if (bi.has_reduce) {
    std::vector<int8_t> stored(tiled.size());
    nso_decompress(base + bi.q_off, stored.data(), stored.size(), qtype);
    uint16_t* rp = (uint16_t*)(base + bi.red_off);
    for (int c = 0; c < n; c++)
      for (int kb = 0; kb < rawnk; kb++) {
        float tmp = 0.f;
        float s = scale_at(bi, base, kb, c);
        int z = bi.is_asym ? zps[size_t(kb) * n + c] : 0;
        for (int kk = kb * bi.blocksize; kk < std::min(k, (kb + 1) * bi.blocksize); kk++)
          tmp += float(int(stored[tiled_index(bi, kk, c)]) - z) * s;
        rp[size_t(kb) * bi.cstep + c] = nso_f32_to_bf16(tmp);
      }
  }
  // setShuffleIndices — bestla_prologue_b.h:337-356: position g * blocksize + (running count of group g) holds the
  // original k index; the caller has already sorted the rows of q the same way (convert/common.py:667-681)
  if (g_idx) {
    int* sp32 = (int*)(base + bi.shuf_off);
    const int groups = int(updiv(k, bi.blocksize));
    std::vector<int> count(groups, 0);
    for (int i = 0; i < k; i++) {
      const int g = g_idx[i];
      if (g < 0 || g >= groups) return -2;
      const size_t pos = size_t(g) * bi.blocksize + count[g]++;
      if (pos >= size_t(k)) return -2;
      sp32[pos] = i;
    }
  }
  return 0;
}
int nso_pack_q(void* blob, const int8_t* q, int ldq, const float* scales, const int8_t* zps, int n, int k,
               int blocksize, uint32_t qtype, uint32_t stype, int asym, int core) {
  return pack_q_impl(blob, q, ldq, scales, zps, n, k, blocksize, qtype, stype, asym, core, nullptr);
}
int nso_pack_q_gidx(void* blob, const int8_t* q, int ldq, const float* scales, const int8_t* zps, int n, int k,
                    int blocksize, uint32_t qtype, uint32_t stype, int asym, int core, const int* g_idx) {
  return pack_q_impl(blob, q, ldq, scales, zps, n, k, blocksize, qtype, stype, asym, core, g_idx);
}
size_t nso_pack_size_gidx(int n, int k, int blocksize, uint32_t qtype, uint32_t stype, int asym, int core) {
  nso_blob_info bi;
  if (!describe(bi, n, k, blocksize, qtype, stype, asym, core, true)) return 0;
  return bi.size;
}

// BTLAGemmQuantPackB — bestla_gemm.cpp:302-319 -> packTransposeWeight/packWeight (prologue_b.h:180-210, :1019-1040)
int nso_quant_pack(void* blob, const float* w, int n, int k, int ldw, int blocksize, uint32_t qtype, uint32_t stype,
                   int asym, int core, int is_trans) {
  nso_blob_info bi;
  if (!describe(bi, n, k, blocksize, qtype, stype, asym, core)) return -1;
  std::vector<float> wt;
  const float* kn = w;
  int ld = ldw;
  if (is_trans) {  // [N][K] -> [K][N]
    wt.resize(size_t(n) * k);
    for (int i = 0; i < n; i++)
      for (int j = 0; j < k; j++) wt[size_t(j) * n + i] = w[size_t(i) * ldw + j];
    kn = wt.data();
    ld = n;
  }
  const int nblk = int(updiv(k, bi.blocksize));
  std::vector<int8_t> q(size_t(n) * k), zp(bi.is_asym ? size_t(nblk) * n : 0);
  std::vector<float> sc(size_t(nblk) * n);
  if (dt_is_int(qtype))
    nso_quantize_int_rowblock(kn, q.data(), k, n, ld, n, sc.data(), bi.is_asym ? zp.data() : nullptr, bi.blocksize,
                              qtype);
  else if (dt_is_f8(qtype))
    nso_quantize_f8_rowblock(kn, q.data(), k, n, ld, n, sc.data(), bi.blocksize, qtype, stype);
  else
    nso_quantize_f4_rowblock(kn, q.data(), k, n, ld, n, sc.data(), bi.blocksize, qtype);
  return nso_pack_q(blob, q.data(), n, sc.data(), bi.is_asym ? zp.data() : nullptr, n, k, blocksize, qtype, stype,
                    asym, core);
}

int nso_unpack_canonical(const void* blob, int8_t* q, float* scales, int8_t* zps) {
  nso_blob_info bi;
  if (nso_blob_parse(blob, &bi)) return -1;
  const uint8_t* base = (const uint8_t*)blob;
  std::vector<int8_t> tiled(size_t(bi.npad) * bi.kpad);
  nso_decompress(base + bi.q_off, tiled.data(), tiled.size(), bi.dtype);
  for (int kk = 0; kk < bi.k; kk++)
    for (int c = 0; c < bi.n; c++) q[size_t(kk) * bi.n + c] = tiled[tiled_index(bi, kk, c)];
  const int nblk = int(updiv(bi.k, bi.blocksize));
  for (int r = 0; r < nblk; r++)
    for (int c = 0; c < bi.n; c++) {
      scales[size_t(r) * bi.n + c] = scale_at(bi, base, r, c);
      if (zps) zps[size_t(r) * bi.n + c] = bi.is_asym ? ((const int8_t*)(base + bi.zp_off))[size_t(r) * bi.cstep + c] : 0;
    }
  return 0;
}

// unpackWeight — bestla_prologue_b.h:211-242 -> getFpWeight -> decompress_kblock_s*_fp (kernel_ref.h:1027-1180):
//   w = float(code - zp) * scale ;  NFloat (kernel_ref.h:1456-1478): w = LUT[code] * scale
int nso_unpack_fp32(const void* blob, float* out, int ldb) {
  nso_blob_info bi;
  if (nso_blob_parse(blob, &bi)) return -1;
  std::vector<int8_t> q(size_t(bi.n) * bi.k), zp;
  const int nblk = int(updiv(bi.k, bi.blocksize));
  std::vector<float> sc(size_t(nblk) * bi.n);
  zp.resize(size_t(nblk) * bi.n);
  nso_unpack_canonical(blob, q.data(), sc.data(), zp.data());
  const bool is_int = bi.prologue_id == 1;
#pragma omp parallel for schedule(static)
  for (int kk = 0; kk < bi.k; kk++)
    for (int c = 0; c < bi.n; c++) {
      const size_t si = size_t(kk / bi.blocksize) * bi.n + c;
      const int8_t code = q[size_t(kk) * bi.n + c];
      float v = is_int ? float(int(code) - int(zp[si]))
                       : (dt_is_f8(bi.dtype) ? f8_to_f32(uint8_t(code), bi.dtype) : f4_unpack(bi.dtype, code));
      out[size_t(kk) * ldb + c] = v * sc[si];
    }
  return 0;
}

// fp64 GEMM over the unpacked weights; per output the products are added in ascending k — column blocks of 128 and row
// blocks of 8 only change which outputs share a pass over W, not any output's summation order.  which: bit 0 = fp32
// activations -> c, bit 1 = activations rounded through fp16 first -> c16 (one unpack serves both).
static int gemm_f64_impl(const float* a, int lda, const void* blob, double* c, double* c16, int ldc, int m, int which) {
  nso_blob_info bi;
  if (nso_blob_parse(blob, &bi)) return -1;
  std::vector<float> w(size_t(bi.k) * bi.n);
  nso_unpack_fp32(blob, w.data(), bi.n);
  const int* shuf = bi.has_shuffle ? (const int*)((const uint8_t*)blob + bi.shuf_off) : nullptr;
  const int K = bi.k, N = bi.n;
  for (int pass = 0; pass < 2; pass++) {
    if (!(which & (1 << pass))) continue;
    const bool a16 = pass == 1;
    double* out = a16 ? c16 : c;
    std::vector<double> arows(size_t(m) * K);
    for (int i = 0; i < m; i++)
      for (int kk = 0; kk < K; kk++) {
        // activation shuffle (g_idx blobs): A'[j] = A[indices[j]], kernel_ref.h:28-37 via prologue_a.h:322-330
        float av = a[size_t(i) * lda + (shuf ? shuf[kk] : kk)];
        if (a16) av = round_through_ieee_f16(av);  // what v_cvt_f16_f32 (RNE) does on the device
        arows[size_t(i) * K + kk] = av;
      }
    constexpr int JB = 128, IB = 8;
#pragma omp parallel for schedule(dynamic, 1)
    for (int jb = 0; jb < N; jb += JB) {
      const int jn = std::min(JB, N - jb);
      for (int ib = 0; ib < m; ib += IB) {
        const int in = std::min(IB, m - ib);
        double acc[IB][JB];
        for (int i = 0; i < in; i++)
          for (int j = 0; j < jn; j++) acc[i][j] = 0;
        for (int kk = 0; kk < K; kk++) {
          const float* wr = &w[size_t(kk) * N + jb];
          for (int i = 0; i < in; i++) {
            const double av = arows[size_t(ib + i) * K + kk];
            for (int j = 0; j < jn; j++) acc[i][j] += av * double(wr[j]);
          }
        }
        for (int i = 0; i < in; i++)
          for (int j = 0; j < jn; j++) out[size_t(ib + i) * ldc + jb + j] = acc[i][j];
      }
    }
  }
  return 0;
}
int nso_gemm_f64(const float* a, int lda, const void* blob, double* c, int ldc, int m) {
  return gemm_f64_impl(a, lda, blob, c, nullptr, ldc, m, 1);
}
int nso_gemm_f64_a16(const float* a, int lda, const void* blob, double* c, int ldc, int m) {
  return gemm_f64_impl(a, lda, blob, nullptr, c, ldc, m, 2);
}
// both forms from one unpack of the blob (full-size parity tests)
int nso_gemm_f64_pair(const float* a, int lda, const void* blob, double* c, double* c16, int ldc, int m) {
  return gemm_f64_impl(a, lda, blob, c, c16, ldc, m, 3);
}

// gemv_{N}bit_fp32_fp32 — kernel_ref.h:2489-2531: acc[n] += a[k] * (code - zp) * scale, fp32, k ascending.
// Reads the packed blob directly (streams the packed image once, like the reference GEMV) — 4-bit and 8-bit
// integer weights and f4 weights; this is the timed "port" CPU baseline.
int nso_gemv_f32(const float* a, int lda, const void* blob, float* c, int ldc, int m, int nthreads) {
  nso_blob_info bi;
  if (nso_blob_parse(blob, &bi)) return -1;
  const int nbits = dt_bits(bi.dtype);
  if (nbits != 4 && nbits != 8) return -2;
  if (bi.has_shuffle) return -4;
  const bool is_int = bi.prologue_id == 1;
  const uint8_t* base = (const uint8_t*)blob;
  const uint8_t* qb = base + bi.q_off;
  const uint8_t* sp = base + bi.scale_off;
  const int8_t* zp = bi.is_asym ? (const int8_t*)(base + bi.zp_off) : nullptr;
  const int ntiles = bi.npad / bi.ntile;
  const int NT = bi.ntile, PR = bi.packrow;
  (void)nthreads;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
  if (m > 8 || NT > 96) return -3;
#pragma omp parallel for schedule(dynamic, 1)
  for (int t = 0; t < ntiles; t++) {
    float acc[8][96], sc[96], wv[96];
    int zz[96];
    for (int i = 0; i < m; i++)
      for (int j = 0; j < NT; j++) acc[i][j] = 0.f;
    for (int j = 0; j < NT; j++) zz[j] = 0;
    const size_t tile_base = size_t(t) * NT * bi.kpad;
    for (int kb = 0; kb * bi.blocksize < bi.k; kb++) {
      for (int j = 0; j < NT; j++) {
        sc[j] = scale_at(bi, base, kb, t * NT + j);
        if (zp) zz[j] = zp[size_t(kb) * bi.cstep + t * NT + j];
      }
      const int kend = std::min(bi.k, (kb + 1) * bi.blocksize);
      for (int kk = kb * bi.blocksize; kk < kend; kk++) {
        const size_t rowoff = tile_base + size_t(kk / PR) * NT * PR + (kk % PR);
        if (nbits == 8 && !is_int) {
          for (int j = 0; j < NT; j++) wv[j] = f8_to_f32(qb[rowoff + size_t(j) * PR], bi.dtype);
        } else if (nbits == 8) {
          for (int j = 0; j < NT; j++) wv[j] = float(int(int8_t(qb[rowoff + size_t(j) * PR])) - zz[j]);
        } else if (is_int) {
          for (int j = 0; j < NT; j++) {
            const size_t e = rowoff + size_t(j) * PR;
            wv[j] = float(int((qb[e >> 1] >> (4 * (e & 1))) & 0xf) - 8 - zz[j]);
          }
        } else {
          for (int j = 0; j < NT; j++) {
            const size_t e = rowoff + size_t(j) * PR;
            wv[j] = f4_unpack(bi.dtype, (qb[e >> 1] >> (4 * (e & 1))) & 0xf);
          }
        }
        for (int i = 0; i < m; i++) {
          const float av = a[size_t(i) * lda + kk];
          for (int j = 0; j < NT; j++) acc[i][j] += av * wv[j] * sc[j];  // (a * w) * scale, kernel_ref.h:2517
        }
      }
    }
    for (int i = 0; i < m; i++)
      for (int j = 0; j < NT; j++)
        if (t * NT + j < bi.n) c[size_t(i) * ldc + t * NT + j] = acc[i][j];
  }
  return 0;
}

// quantize_fp_u8_colblock — kernel_ref.h:1824-1883 (dynamic per-(row, k-block) asymmetric u8 activation quant)
int nso_quantize_fp_u8_colblock(int row, int col, const float* src, int ld_src, uint8_t* dst, int ld_dst,
                                float* scales, int ld_scale, uint8_t* zps, int blocksize, float* blkreduce) {
  for (int i = 0; i < row; i++) {
    for (int j = 0; j < col; j += blocksize) {
      const bool tail = j + blocksize > col;
      const int bs = tail ? col - j : blocksize;
      float maxval = tail ? 0.f : FLT_MIN;  // :1832 vs :1857
      float minval = 0.f;
      for (int ij = 0; ij < bs; ij++) {
        float f = src[size_t(i) * ld_src + j + ij];
        maxval = std::max(f, maxval);
        minval = std::min(f, minval);
      }
      float scale = (maxval - minval) / 255;
      uint8_t zp = cast_f32_u8((0 - minval) / scale);
      float rscale = 1.f / scale;
      scales[size_t(i) * ld_scale + j / blocksize] = scale;
      zps[size_t(i) * ld_scale + j / blocksize] = zp;
      int sum = 0;
      float zpf = float(zp);
      for (int ij = 0; ij < bs; ij++) {
        int qtmp = cast_f32_int(src[size_t(i) * ld_src + j + ij] * rscale);
        sum += qtmp;
        dst[size_t(i) * ld_dst + j + ij] = cast_f32_u8(zpf + qtmp);
      }
      if (blkreduce) blkreduce[size_t(i) * ld_scale + j / blocksize] = sum * scale;
    }
  }
  return 0;
}

// int8-compute semantics — gemv_4bit_u8s8_fp32 (kernel_ref.h:2371-2429): per 4-k step and per element,
//   acc += int(a_q - zp_a) * (code - zp_b) * (scale_a * scale_b)   in fp32, k ascending.
int nso_gemm_u8s8_f32(const float* a, int lda, const void* blob, float* c, int ldc, int m) {
  nso_blob_info bi;
  if (nso_blob_parse(blob, &bi) || bi.prologue_id != 1) return -1;
  const int nblk = int(updiv(bi.k, bi.blocksize));
  std::vector<int8_t> q(size_t(bi.n) * bi.k), zp(size_t(nblk) * bi.n);
  std::vector<float> sc(size_t(nblk) * bi.n);
  nso_unpack_canonical(blob, q.data(), sc.data(), zp.data());
  std::vector<uint8_t> aq(size_t(m) * bi.k), azp(size_t(m) * nblk);
  std::vector<float> as(size_t(m) * nblk);
  nso_quantize_fp_u8_colblock(m, bi.k, a, lda, aq.data(), bi.k, as.data(), nblk, azp.data(), bi.blocksize, nullptr);
  for (int i = 0; i < m; i++)
#pragma omp parallel for schedule(static)
    for (int j = 0; j < bi.n; j++) {
      float acc = 0.f;
      for (int kk = 0; kk < bi.k; kk++) {
        int kb = kk / bi.blocksize;
        float vscale = as[size_t(i) * nblk + kb] * sc[size_t(kb) * bi.n + j];
        acc += float(int(aq[size_t(i) * bi.k + kk]) - int(azp[size_t(i) * nblk + kb])) *
               float(int(q[size_t(kk) * bi.n + j]) - int(zp[size_t(kb) * bi.n + j])) * vscale;
      }
      c[size_t(i) * ldc + j] = acc;
    }
  return 0;
}

// gemv_4bit_u8s8_fp32 read straight from the packed blob (kernel_ref.h:2371-2429 driven the way
// LauncherIntKBlock::GEMVWrapper::gemv_kblock does it, bestla_wrapper.h:643-688): activations quantized once per call
// to u8 (quantize_fp_u8_colblock), then per column tile and 4-k step acc += (a_q - zp_a) * (code - zp_b) * (scale_a *
// scale_b) in fp32, k ascending — the same sums as nso_gemm_u8s8_f32 (tested equal), streamed and threaded over the
// tiles.  4-bit and 8-bit integer weights.  This is the timed "port" of the reference's DEFAULT decode path.
int nso_gemv_u8s8_f32(const float* a, int lda, const void* blob, float* c, int ldc, int m, int nthreads) {
  nso_blob_info bi;
  if (nso_blob_parse(blob, &bi) || bi.prologue_id != 1) return -1;
  const int nbits = dt_bits(bi.dtype);
  if (nbits != 4 && nbits != 8) return -2;
  if (bi.has_shuffle) return -4;
  const uint8_t* base = (const uint8_t*)blob;
  const uint8_t* qb = base + bi.q_off;
  const uint8_t* sp = base + bi.scale_off;
  const int8_t* zp = bi.is_asym ? (const int8_t*)(base + bi.zp_off) : nullptr;
  const int ntiles = bi.npad / bi.ntile;
  const int NT = bi.ntile, PR = bi.packrow;
  if (m > 8 || NT > 96) return -3;
  const int nblk = int(updiv(bi.k, bi.blocksize));
  std::vector<uint8_t> aq(size_t(m) * bi.k), azp(size_t(m) * nblk);
  std::vector<float> as(size_t(m) * nblk);
  nso_quantize_fp_u8_colblock(m, bi.k, a, lda, aq.data(), bi.k, as.data(), nblk, azp.data(), bi.blocksize, nullptr);
  (void)nthreads;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(dynamic, 1)
  for (int t = 0; t < ntiles; t++) {
    float acc[8][96], sc[96];
    int zz[96], wq[96];
    for (int i = 0; i < m; i++)
      for (int j = 0; j < NT; j++) acc[i][j] = 0.f;
    for (int j = 0; j < NT; j++) zz[j] = 0;
    const size_t tile_base = size_t(t) * NT * bi.kpad;
    for (int kb = 0; kb < nblk; kb++) {
      for (int j = 0; j < NT; j++) {
        sc[j] = scale_at(bi, base, kb, t * NT + j);
        if (zp) zz[j] = zp[size_t(kb) * bi.cstep + t * NT + j];
      }
      const int kend = std::min(bi.k, (kb + 1) * bi.blocksize);
      for (int kk = kb * bi.blocksize; kk < kend; kk++) {
        const size_t rowoff = tile_base + size_t(kk / PR) * NT * PR + (kk % PR);
        if (nbits == 8) {
          for (int j = 0; j < NT; j++) wq[j] = int(int8_t(qb[rowoff + size_t(j) * PR])) - zz[j];
        } else {
          for (int j = 0; j < NT; j++) {
            const size_t e = rowoff + size_t(j) * PR;
            wq[j] = int((qb[e >> 1] >> (4 * (e & 1))) & 0xf) - 8 - zz[j];
          }
        }
        for (int i = 0; i < m; i++) {
          const float av = float(int(aq[size_t(i) * bi.k + kk]) - int(azp[size_t(i) * nblk + kb]));
          const float sa = as[size_t(i) * nblk + kb];
          for (int j = 0; j < NT; j++) acc[i][j] += av * float(wq[j]) * (sa * sc[j]);
        }
      }
    }
    for (int i = 0; i < m; i++)
      for (int j = 0; j < NT; j++)
        if (t * NT + j < bi.n) c[size_t(i) * ldc + t * NT + j] = acc[i][j];
  }
  return 0;
}

// postop — kernel_ref.h:1569-1578
float nso_gelu(float x) { return 0.5f * x * (1.f + tanhf(0.7978845834732056f * (x + 0.044714998453855515f * x * x * x))); }
float nso_silu(float x) { return float(x / (1 + exp(-x))); }

// ne_compute_forward_rope_f32 — ne_layers.c:9243-9428 (see ns_oracle.h for the covered modes)
// rope_yarn_ramp / rope_yarn — ne_layers.c:9196-9217
static float rope_yarn_ramp(const float low, const float high, const int i0) {
  const float y = (i0 / 2 - low) / std::max(0.001f, high - low);
  return float(1.0 - std::min(1.0, std::max(0.0, double(y))));
}
static void rope_yarn(float theta_extrap, float freq_scale, const float corr_dims[2], int64_t i0, float ext_factor,
                      float mscale, float* cos_theta, float* sin_theta) {
  float theta_interp = freq_scale * theta_extrap;
  float theta = theta_interp;
  if (ext_factor != 0.0f) {
    float ramp_mix = rope_yarn_ramp(corr_dims[0], corr_dims[1], int(i0)) * ext_factor;
    theta = theta_interp * (1 - ramp_mix) + theta_extrap * ramp_mix;
    mscale *= 1.0f + 0.1f * logf(1.0f / freq_scale);
  }
  *cos_theta = cosf(theta) * mscale;
  *sin_theta = sinf(theta) * mscale;
}
// ggml_rope_yarn_corr_dim(s) — ne_layers.c:9219-9231
static void rope_yarn_corr_dims(int n_dims, int n_orig_ctx, float freq_base, float beta_fast, float beta_slow, float dims[2]) {
  auto corr_dim = [&](float n_rot) {
    return n_dims * logf(n_orig_ctx / (n_rot * 2 * (float)3.14159265358979323846)) / (2 * logf(freq_base));
  };
  dims[0] = std::max(0.f, floorf(corr_dim(beta_fast)));
  dims[1] = std::min(float(n_dims - 1), ceilf(corr_dim(beta_slow)));
}

static int rope_impl(const float* src, float* dst, int batch, int seq, int heads, int head_size, int n_past, int n_dims,
                     int mode, float freq_base, float freq_scale, int n_orig_ctx, float ext_factor, float attn_factor,
                     float beta_fast, float beta_slow, const float* longrope_factor, float scale_factor) {
  if ((mode & ~(2 | 8 | 0x10)) != 0 || n_dims > head_size || (n_dims & 1) || n_dims <= 0) return -1;
  const bool is_neox = (mode & 2) != 0;
  const bool is_longrope = (mode & 0x10) != 0;
  if (is_longrope && !longrope_factor) return -1;
  const float theta_scale = powf(freq_base, -2.0f / n_dims);  // :9300
  const float inv_ndims = -1.f / n_dims;                      // :9301
  float corr_dims[2] = {0.f, 0.f};
  if (ext_factor != 0.f) rope_yarn_corr_dims(n_dims, n_orig_ctx, freq_base, beta_fast, beta_slow, corr_dims);  // :9302-9303
  for (int i3 = 0; i3 < batch; i3++)
    for (int i2 = 0; i2 < seq; i2++) {
      const int p = n_past + i2;  // :9316 (mode & 1 == 0)
      for (int i1 = 0; i1 < heads; i1++) {
        const size_t row = ((size_t(i3) * seq + i2) * heads + i1) * head_size;
        const float* x = src + row;
        float* y = dst + row;
        memcpy(y, x, size_t(head_size) * 4);  // dims not touched by the NeoX loop keep their value (dst == src in-place)
        float theta_base = float(p);
        if (is_longrope) {  // :9349-9377 (tested before the NeoX flag)
          theta_base = theta_base * freq_scale;
          for (int ib = 0; ib < head_size / n_dims; ib++)
            for (int ic = 0; ic < n_dims; ic += 2) {
              const float cur_rot = inv_ndims * ic - ib;
              float c, s_;
              const float tmp_factor = longrope_factor[ic / 2];
              const float tmp_theta_base = theta_base / tmp_factor;
              rope_yarn(tmp_theta_base, freq_scale, corr_dims, (int)cur_rot, ext_factor, attn_factor, &c, &s_);
              c *= scale_factor;
              s_ *= scale_factor;
              theta_base *= theta_scale;
              const int i0 = ib * n_dims + ic / 2;
              const float x0 = x[i0], x1 = x[i0 + n_dims / 2];
              y[i0] = x0 * c - x1 * s_;
              y[i0 + n_dims / 2] = x0 * s_ + x1 * c;
            }
        } else if (!is_neox) {  // :9379-9395
          for (int i0 = 0; i0 < head_size; i0 += 2) {
            float c, s_;
            rope_yarn(theta_base, freq_scale, corr_dims, i0, ext_factor, attn_factor, &c, &s_);
            theta_base *= theta_scale;
            const float x0 = x[i0], x1 = x[i0 + 1];
            y[i0] = x0 * c - x1 * s_;
            y[i0 + 1] = x0 * s_ + x1 * c;
          }
        } else {  // :9396-9423
          theta_base = theta_base * freq_scale;
          for (int ib = 0; ib < head_size / n_dims; ib++)
            for (int ic = 0; ic < n_dims; ic += 2) {
              const float cur_rot = inv_ndims * ic - ib;  // :9403
              float c, s_;
              rope_yarn(theta_base, freq_scale, corr_dims, (int)cur_rot, ext_factor, attn_factor, &c, &s_);
              theta_base *= theta_scale;
              const int i0 = ib * n_dims + ic / 2;
              const float x0 = x[i0], x1 = x[i0 + n_dims / 2];
              y[i0] = x0 * c - x1 * s_;
              y[i0 + n_dims / 2] = x0 * s_ + x1 * c;
            }
        }
      }
    }
  return 0;
}
int nso_rope_f32_yarn(const float* src, float* dst, int batch, int seq, int heads, int head_size, int n_past, int n_dims,
                      int mode, float freq_base, float freq_scale, int n_orig_ctx, float ext_factor, float attn_factor,
                      float beta_fast, float beta_slow) {
  if (mode & 0x10) return -1;
  return rope_impl(src, dst, batch, seq, heads, head_size, n_past, n_dims, mode, freq_base, freq_scale, n_orig_ctx, ext_factor,
                   attn_factor, beta_fast, beta_slow, nullptr, 1.f);
}
int nso_rope_f32_longrope(const float* src, float* dst, int batch, int seq, int heads, int head_size, int n_past, int n_dims,
                          float freq_base, float freq_scale, int n_orig_ctx, float ext_factor, float attn_factor,
                          float beta_fast, float beta_slow, const float* factors, float scale_factor) {
  return rope_impl(src, dst, batch, seq, heads, head_size, n_past, n_dims, 0x10, freq_base, freq_scale, n_orig_ctx, ext_factor,
                   attn_factor, beta_fast, beta_slow, factors, scale_factor);
}
int nso_rope_f32(const float* src, float* dst, int batch, int seq, int heads, int head_size, int n_past, int n_dims,
                 int mode, float freq_base, float freq_scale, float attn_factor) {
  if ((mode & ~2) != 0) return -1;
  return nso_rope_f32_yarn(src, dst, batch, seq, heads, head_size, n_past, n_dims, mode, freq_base, freq_scale, 0, 0.f,
                           attn_factor, 0.f, 0.f);
}

// GLM branch of ne_compute_forward_rope_f32 (mode & 4) — ne_layers.c:9317-9347: two-dimensional position encoding of
// ChatGLM.  Per row the first half of the head is rotated by the clamped token position, the second half by the block
// position; both angles are multiplied by theta_scale after every element, cos / sin in fp32.  n_padding comes from
// src1[ROPE_PARAMS_NUM + batch index] (:9319).  mode & 1 ("skip"): rows of positions < n_past are left untouched and
// p = i2 (:9313-9314).
int nso_rope_f32_glm(const float* src, float* dst, int batch, int seq, int heads, int head_size, int n_past, int n_dims,
                     int mode, float freq_base, int prompt_size, const int* n_padding) {
  if (!(mode & 4) || (mode & ~5) || n_dims % 2 || n_dims / 2 * 3 + head_size / 4 > head_size) return -1;
  const bool skip = mode & 1;
  const float theta_scale = powf(freq_base, -2.0f / n_dims);
  for (int64_t i3 = 0; i3 < batch; i3++)
    for (int64_t i2 = (skip ? n_past : 0); i2 < seq; i2++) {
      const int64_t p = skip ? i2 : n_past + i2;
      for (int64_t i1 = 0; i1 < heads; i1++) {
        const int64_t npad = n_padding[i3];
        float theta_base = float(std::min(std::max(p - npad, int64_t(0)), int64_t(prompt_size) - 2 - npad));
        float block_theta = float(std::max(p - (int64_t(prompt_size) - 2), int64_t(0)));
        const size_t row = ((size_t(i3) * seq + i2) * heads + i1) * head_size;
        for (int64_t i0 = 0; i0 < head_size / 4; i0++) {
          const float cos_theta = cosf(theta_base), sin_theta = sinf(theta_base);
          const float cos_block_theta = cosf(block_theta), sin_block_theta = sinf(block_theta);
          theta_base *= theta_scale;
          block_theta *= theta_scale;
          const float* s = src + row + i0;
          float* d = dst + row + i0;
          const float x0 = s[0], x1 = s[n_dims / 2], x2 = s[n_dims], x3 = s[n_dims / 2 * 3];
          d[0] = x0 * cos_theta - x1 * sin_theta;
          d[n_dims / 2] = x0 * sin_theta + x1 * cos_theta;
          d[n_dims] = x2 * cos_block_theta - x3 * sin_block_theta;
          d[n_dims / 2 * 3] = x2 * sin_block_theta + x3 * cos_block_theta;
        }
      }
    }
  return 0;
}

// bestla_fusion_attn_forward_ref — mha_dense_wrapper.h:1371-1517 (PLAIN layouts; fp32 accumulation in the loop order
// of the reference: scores j ascending / k ascending, then exp, then P.V k ascending)
/* mha_exp_ref with MHA_2ND_EXP = 1 (mha_dense_wrapper.h:41, :79-85) = kernel::ref::exp_ps_0_1 (kernel_ref.h:2253-2262): 2^z times a
 * second-order polynomial in the fraction — the exp the reference's attention reference (and its kernels) use */
static float nso_exp_ps_0_1(float x) {
  static const float log2e = std::log2(std::exp(1.f));
  const float x1 = x * log2e + .5f;
  const float z = std::floor(x1);
  const float f = x1 - z;
  return ldexpf(0.240226507f * f * f + 0.452920674f * f + 0.713483036f, static_cast<int>(z));
}

int nso_attn_ref(const nso_attn_args* a, int mode) {
  const int bf16_gemm = mode & 1;
  const bool exp2nd = (mode & 2) != 0;
  const bool is_causal = (a->flags & 1u) != 0, is_alibi = (a->flags & 2u) != 0;
  if (is_causal && a->sl_q > a->sl_kv) return -1;
  if (a->heads_kv <= 0 || a->head_num % a->heads_kv) return -1;
  const int group_heads = a->head_num / a->heads_kv;
  const int lf = 1 << int(floor(log2(double(a->head_num))));  // :1424-1426
  const float m0 = powf(2.0f, -(8.f) / lf), m1 = powf(2.0f, -(8.f / 2.0f) / lf);
  auto bf = [&](float x) { return bf16_gemm ? nso_bf16_to_f32(nso_f32_to_bf16(x)) : x; };
  // K and V are fp16: static_cast<bf16>(fp16) is fp16::operator bf16() (bestla_utils.h:208-229) — the mantissa is TRUNCATED to
  // seven bits, fp16 subnormals become zero, exponent 31 becomes sign | 0x7fff
  auto bfh = [&](uint16_t h) -> float {
    if (!bf16_gemm) return nso_f16_to_f32(h);
    const int e = (h >> 10) & 0x1f, m = h & 0x3ff;
    uint16_t b;
    if (e == 0) b = 0;
    else if (e == 31) b = uint16_t(h | 0x7fff);
    else b = uint16_t((h & 0x8000) | ((e + 128 - 16) << 7) | (m >> 3));
    return nso_bf16_to_f32(b);
  };
  std::vector<float> row(size_t(a->sl_kv));
  for (int ibs = 0; ibs < a->batch_size; ibs++)
    for (int ihn = 0; ihn < a->head_num; ihn++)
      for (int i = 0; i < a->sl_q; i++) {
        const int ihkv = ihn / group_heads;
        const float* q = a->q + ibs * a->step_q_bs + ihn * a->step_q_head_num + i * a->step_q_sl;
        float* dst = a->dst + ibs * a->step_dst_bs + ihn * a->step_dst_head_num + i * a->step_dst_sl;
        const uint16_t* kc = a->k + ibs * a->step_k_bs + ihkv * a->step_k_head_num;
        const uint16_t* vc = a->v + ibs * a->step_v_bs + ihkv * a->step_v_head_num;
        const int unmasked = is_causal ? (a->sl_kv - a->sl_q) + i + 1 : a->sl_kv;  // :1440-1441
        const float slope = !is_alibi ? 0.f : (ihn < lf ? powf(m0, float(ihn + 1)) : powf(m1, float(2 * (ihn - lf) + 1)));
        float row_max = -INFINITY;
        for (int j = 0; j < unmasked; j++) {  // :1451-1474
          float s = 0.f;
          for (int k = 0; k < a->head_size; k++)
            s += bf(q[k]) * bfh(kc[j * a->step_k_sl + k * a->step_k_head_size]);
          s = s * a->qk_scale * a->q_sc * a->k_sc + j * slope;
          row[size_t(j)] = s;
          row_max = std::max(row_max, s);
        }
        float exp_sum = 0.f;
        for (int j = 0; j < unmasked; j++) {  // :1477-1481
          row[size_t(j)] = exp2nd ? nso_exp_ps_0_1(row[size_t(j)] - row_max) : expf(row[size_t(j)] - row_max);
          exp_sum += row[size_t(j)];
        }
        for (int j = 0; j < unmasked; j++) row[size_t(j)] = bf(row[size_t(j)] / exp_sum);  // :1487-1490
        for (int j = 0; j < a->head_size; j++) {  // :1494-1512
          float acc = 0.f;
          for (int k = 0; k < unmasked; k++) acc += row[size_t(k)] * bfh(vc[k * a->step_v_sl + j]);
          dst[j] = acc * a->v_sc / a->dst_sc;
        }
      }
  return 0;
}

}  // extern "C"
