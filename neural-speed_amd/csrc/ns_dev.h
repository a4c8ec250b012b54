// ns_dev.h — device-side helpers shared by the gfx950 kernels of libns_hip.so (ns_kernels.hip, ns_gemv.hip, ns_gemm.hip):
// code -> fp16 converters, raw scale / zero-point records, buffer-descriptor loads, epilogue activations.
#pragma once
#include <hip/hip_runtime.h>

#include "ns_common.h"

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef uint32_t uint4v __attribute__((ext_vector_type(4)));

namespace ns {

// x86 float -> integer conversions as the reference binary performs them (the activation quantizer of the int8-compute path,
// kernel_ref.h:1824-1883; ns_quant.hip holds the same functions for the offline quantizer).  Every operation is a correctly rounded
// intrinsic: no FMA contraction whatever the translation unit's flags are.
__device__ __forceinline__ float x86_round_half_away(float x) {  // roundf, exact: x - trunc(x) is always representable
  const float t = truncf(x);
  return (fabsf(__fsub_rn(x, t)) >= 0.5f) ? __fadd_rn(t, copysignf(1.f, x)) : t;
}
__device__ __forceinline__ int x86_cvt_round_int(float v) {  // cast<float,int>: cvttss2si yields INT_MIN for NaN / out of range
  const float r = x86_round_half_away(v);
  if (!(r >= -2147483648.f && r < 2147483648.f)) return (-2147483647 - 1);
  return int(r);
}
__device__ __forceinline__ int x86_cast_f32_u8(float v) {  // bestla_utils.h:515-521: +0.5, clamp, truncate; NaN -> 0
  if (v != v) return 0;
  v = __fadd_rn(v, 0.5f);
  v = v > 255.f ? 255.f : v;
  v = v < 0.f ? 0.f : v;
  return int(v);
}

// ============================================================================================================
// small device helpers
// ============================================================================================================
__device__ __forceinline__ half2_t as_half2(uint32_t u) { return __builtin_bit_cast(half2_t, u); }
__device__ __forceinline__ uint32_t as_u32(half2_t h) { return __builtin_bit_cast(uint32_t, h); }
__device__ __forceinline__ float bf16_bits_to_f32(uint32_t b) { return __builtin_bit_cast(float, b << 16); }
__device__ __forceinline__ float f16_bits_to_f32(uint32_t b) {
  return float(__builtin_bit_cast(_Float16, (unsigned short)b));
}
__device__ __forceinline__ float load_scale(const void* base, size_t idx, uint32_t dt) {
  if (dt == DT_F32) return static_cast<const float*>(base)[idx];
  uint32_t h = static_cast<const unsigned short*>(base)[idx];
  return dt == DT_BF16 ? bf16_bits_to_f32(h) : f16_bits_to_f32(h);
}
// streaming (read-once) 16-byte load: non-temporal so the weight stream does not evict A / scales from L2/MALL
__device__ __forceinline__ uint4 ld_stream(const uint4* p) {
  const uint4v v = __builtin_nontemporal_load(reinterpret_cast<const uint4v*>(p));
  return uint4{v.x, v.y, v.z, v.w};
}
// nibble i (0..7, ascending k) of a device dword sits at this bit: pairs (0,1) (2,3) (4,5) (6,7) are MFMA k-pairs
__device__ __host__ __forceinline__ int nib_shift(int i) { return ((i & 1) << 4) + ((i >> 1) << 2); }

// ============================================================================================================
// code -> fp16 converters (exact for integer codes)
// ============================================================================================================
__device__ __forceinline__ uint32_t and_or(uint32_t x, uint32_t mask, uint32_t bits) {
  uint32_t r;  // hipcc splits (x & m) | c into v_and + v_or when both are literals; one VOP3 does it
  asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(mask), "v"(bits));
  return r;
}
// 8 nibbles of one dword -> 8 fp16 = (code - 8 - zp).  Magic 0x6400 = 1024.0h whose mantissa LSBs take the nibble:
// (x & 0x000f000f)|0x64006400 = {1024+u, 1024+u'}, (x & 0x00f000f0)|0x64006400 = {1024+16u, 1024+16u'}.
struct I4Consts {
  uint32_t mlo, mhi, magic;
};
__device__ __forceinline__ half8_t cvt_i4x8(uint32_t x, const I4Consts& c, half2_t off_lo, half2_t off_hi) {
  const half2_t k16 = {(_Float16)0.0625f, (_Float16)0.0625f};
  const uint32_t y = x >> 8;
  half2_t h0 = as_half2(and_or(x, c.mlo, c.magic)) + off_lo;
  half2_t h1 = as_half2(and_or(x, c.mhi, c.magic)) * k16 + off_hi;
  half2_t h2 = as_half2(and_or(y, c.mlo, c.magic)) + off_lo;
  half2_t h3 = as_half2(and_or(y, c.mhi, c.magic)) * k16 + off_hi;
  uint4v r = {as_u32(h0), as_u32(h1), as_u32(h2), as_u32(h3)};
  return __builtin_bit_cast(half8_t, r);
}
// 8 signed bytes (two dwords) -> 8 fp16 = (q - zp): bias to unsigned, splice under 0x64, subtract 1152 + zp
__device__ __forceinline__ half8_t cvt_i8x8(uint32_t x0, uint32_t x1, half2_t off) {
  const uint32_t a = x0 ^ 0x80808080u, b = x1 ^ 0x80808080u;
  half2_t h0 = as_half2(__builtin_amdgcn_perm(0x64646464u, a, 0x04010400u)) + off;
  half2_t h1 = as_half2(__builtin_amdgcn_perm(0x64646464u, a, 0x04030402u)) + off;
  half2_t h2 = as_half2(__builtin_amdgcn_perm(0x64646464u, b, 0x04010400u)) + off;
  half2_t h3 = as_half2(__builtin_amdgcn_perm(0x64646464u, b, 0x04030402u)) + off;
  uint4v r = {as_u32(h0), as_u32(h1), as_u32(h2), as_u32(h3)};
  return __builtin_bit_cast(half8_t, r);
}
// the same for UNSIGNED bytes (native bit-plane records hold the stored codes): splice under 0x64, subtract 1024 + bias + zp
__device__ __forceinline__ half8_t cvt_u8x8(uint32_t a, uint32_t b, half2_t off) {
  half2_t h0 = as_half2(__builtin_amdgcn_perm(0x64646464u, a, 0x04010400u)) + off;
  half2_t h1 = as_half2(__builtin_amdgcn_perm(0x64646464u, a, 0x04030402u)) + off;
  half2_t h2 = as_half2(__builtin_amdgcn_perm(0x64646464u, b, 0x04010400u)) + off;
  half2_t h3 = as_half2(__builtin_amdgcn_perm(0x64646464u, b, 0x04030402u)) + off;
  uint4v r = {as_u32(h0), as_u32(h1), as_u32(h2), as_u32(h3)};
  return __builtin_bit_cast(half8_t, r);
}
// 8 fp8 codes (two dwords) -> 8 fp16, exact.  The reference's fp8 has no zero, subnormals, inf or nan (f8_to_fp32,
// kernel_ref.h:984-1002): value = +-2^(e - bias) * (1 + m / 2^mbits) for every code.  With the byte in the high half
// of a 16-bit lane (u = code << 8, sign already in place):
//   E4M3 (bias 7):  fp16 = (u & 0x7f00) >> 1 + 0x2000        exponent field e + 8, mantissa m << 7
//   E5M2 (bias 15): fp16 = u & 0x7f00 for e >= 1 (same fields); e == 0 is 2^-15 * (1 + m/4), an fp16 SUBNORMAL with
//                   bits 0x200 + (m << 7) — both cases are max(u, (u >> 1) + 0x200)
// so one formula with two constants covers both: abs = max(u & M, (u >> 1) + C);  E4M3: C = 0x2000, M = 0;
// E5M2: C = 0x0200, M = 0xffff.  v_mfma_f32_16x16x32_f16 keeps fp16 subnormal inputs (default kernel mode).
constexpr bool kind_is_8bit(int kind) { return kind == WK_INT8 || kind == WK_F8; }
struct F8Consts {
  uint32_t c2, m2;  // C and M replicated into both 16-bit halves
};
__device__ __forceinline__ uint32_t cvt_f8x2(uint32_t t, const F8Consts& k) {
  typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
  const uint32_t u = t & 0x7f007f00u;
  const uint32_t v = (u >> 1) + k.c2;
  const u16x2 mx = __builtin_elementwise_max(__builtin_bit_cast(u16x2, u & k.m2), __builtin_bit_cast(u16x2, v));
  return (t & 0x80008000u) | __builtin_bit_cast(uint32_t, mx);
}
__device__ __forceinline__ half8_t cvt_f8x8(uint32_t x0, uint32_t x1, const F8Consts& k) {
  uint4v r = {cvt_f8x2(__builtin_amdgcn_perm(0u, x0, 0x010c000cu), k), cvt_f8x2(__builtin_amdgcn_perm(0u, x0, 0x030c020cu), k),
              cvt_f8x2(__builtin_amdgcn_perm(0u, x1, 0x010c000cu), k), cvt_f8x2(__builtin_amdgcn_perm(0u, x1, 0x030c020cu), k)};
  return __builtin_bit_cast(half8_t, r);
}
inline F8Consts f8_consts(uint32_t qtype) {
  return qtype == DT_F8_E5M2 ? F8Consts{0x02000200u, 0xffffffffu} : F8Consts{0x20002000u, 0u};
}
// 16-entry fp16 LUT held as byte planes: lo[e] / hi[e] for e = 0..15, four entries per dword
struct F4Lut {
  uint32_t lo[4], hi[4];
};
// byte-plane lookup of four codes (one per byte, 0..15): entries 0..7 and 8..15 through one v_perm_b32 each on the code's
// low three bits, then a third v_perm_b32 picks per byte by bit 3 (sel2 = 0x03020100 + 4 * bit3: bytes 0..3 of the first
// result, 4..7 of the second).  sel / sel2 are shared by the low- and the high-byte plane.
__device__ __forceinline__ uint32_t lut_bytes2(uint32_t sel, uint32_t sel2, const uint32_t* t) {
  const uint32_t a = __builtin_amdgcn_perm(t[1], t[0], sel);
  const uint32_t b = __builtin_amdgcn_perm(t[3], t[2], sel);
  return __builtin_amdgcn_perm(b, a, sel2);
}
__device__ __forceinline__ half8_t cvt_f4x8(uint32_t x, const F4Lut& lut) {
  // nibbles of x by byte: low nibbles = codes i0 i4 i1 i5, high nibbles = codes i2 i6 i3 i7
  const uint32_t xs = x >> 4;
  const uint32_t s0 = x & 0x07070707u, s1 = xs & 0x07070707u;
  const uint32_t p0 = and_or(x >> 1, 0x04040404u, 0x03020100u), p1 = and_or(xs >> 1, 0x04040404u, 0x03020100u);
  const uint32_t lo0 = lut_bytes2(s0, p0, lut.lo), hi0 = lut_bytes2(s0, p0, lut.hi);
  const uint32_t lo1 = lut_bytes2(s1, p1, lut.lo), hi1 = lut_bytes2(s1, p1, lut.hi);
  uint4v r = {__builtin_amdgcn_perm(hi0, lo0, 0x06020400u), __builtin_amdgcn_perm(hi1, lo1, 0x06020400u),
              __builtin_amdgcn_perm(hi0, lo0, 0x07030501u), __builtin_amdgcn_perm(hi1, lo1, 0x07030501u)};
  return __builtin_bit_cast(half8_t, r);
}

// raw (unconverted) per-k-step correction words of one lane: loaded early, converted at use
enum ScaleKind { SK_BF16 = 0, SK_F16 = 1, SK_F32 = 2 };
template <int SPS, int SK, bool ASYM>
struct CorrRaw {
  static constexpr bool S32 = SK == SK_F32;
  static constexpr int NW32 = S32 ? SPS : (SPS + 1) / 2;
  uint32_t s[NW32];
  uint32_t z[ASYM ? 1 : 0];
};

// Buffer (SRD) loads: the per-lane part of every address is a loop-invariant 32-bit voffset and everything that
// changes per k-step is a scalar soffset, so the streaming loop carries no 64-bit VGPR address arithmetic.
using Rsrc = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ Rsrc make_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
template <int SPS, int SK, bool ASYM>
__device__ __forceinline__ void corr_issue(Rsrc rs, Rsrc rz, uint32_t voff_s, uint32_t voff_z, uint32_t soff_s,
                                           uint32_t soff_z, CorrRaw<SPS, SK, ASYM>& r) {
  constexpr int kBytes = SPS * (SK == SK_F32 ? 4 : 2);
  if constexpr (kBytes == 16) {
    const uint4v v = __builtin_bit_cast(uint4v, __builtin_amdgcn_raw_buffer_load_b128(rs, voff_s, soff_s, 0));
    r.s[0] = v.x, r.s[1] = v.y, r.s[2] = v.z, r.s[3] = v.w;
  } else if constexpr (kBytes == 8) {
    typedef uint32_t uint2v __attribute__((ext_vector_type(2)));
    const uint2v v = __builtin_bit_cast(uint2v, __builtin_amdgcn_raw_buffer_load_b64(rs, voff_s, soff_s, 0));
    r.s[0] = v.x, r.s[1] = v.y;
  } else if constexpr (kBytes == 4) {
    r.s[0] = __builtin_amdgcn_raw_buffer_load_b32(rs, voff_s, soff_s, 0);
  } else {
    r.s[0] = __builtin_amdgcn_raw_buffer_load_b16(rs, voff_s, soff_s, 0);
  }
  if constexpr (ASYM) {
    if constexpr (SPS == 4)
      r.z[0] = __builtin_amdgcn_raw_buffer_load_b32(rz, voff_z, soff_z, 0);
    else if constexpr (SPS == 2)
      r.z[0] = __builtin_amdgcn_raw_buffer_load_b16(rz, voff_z, soff_z, 0);
    else
      r.z[0] = __builtin_amdgcn_raw_buffer_load_b8(rz, voff_z, soff_z, 0);
  }
}

// same record through plain global loads (ps / pz = the lane's scale / zero-point address)
template <int SPS, int SK, bool ASYM>
__device__ __forceinline__ void corr_issue_g(const uint8_t __attribute__((address_space(1))) * ps,
                                             const uint8_t __attribute__((address_space(1))) * pz,
                                             CorrRaw<SPS, SK, ASYM>& r) {
  constexpr int kBytes = SPS * (SK == SK_F32 ? 4 : 2);
  if constexpr (kBytes == 16) {
    const uint4v v = *reinterpret_cast<const uint4v __attribute__((address_space(1)))*>(ps);
    r.s[0] = v.x, r.s[1] = v.y, r.s[2] = v.z, r.s[3] = v.w;
  } else if constexpr (kBytes == 8) {
    typedef uint32_t uint2v __attribute__((ext_vector_type(2)));
    const uint2v v = *reinterpret_cast<const uint2v __attribute__((address_space(1)))*>(ps);
    r.s[0] = v.x, r.s[1] = v.y;
  } else if constexpr (kBytes == 4) {
    r.s[0] = *reinterpret_cast<const uint32_t __attribute__((address_space(1)))*>(ps);
  } else {
    r.s[0] = *reinterpret_cast<const uint16_t __attribute__((address_space(1)))*>(ps);
  }
  if constexpr (ASYM) {
    if constexpr (SPS == 4)
      r.z[0] = *reinterpret_cast<const uint32_t __attribute__((address_space(1)))*>(pz);
    else if constexpr (SPS == 2)
      r.z[0] = *reinterpret_cast<const uint16_t __attribute__((address_space(1)))*>(pz);
    else
      r.z[0] = *pz;
  }
}

template <int SPS, int SK, bool ASYM, int NJ>
__device__ __forceinline__ void corr_decode(const CorrRaw<SPS, SK, ASYM>& r, float (&sc)[4], float (&zp)[4]) {
  float s[SPS];
#pragma unroll
  for (int i = 0; i < SPS; i++) {
    if constexpr (SK == SK_F32) {
      s[i] = __builtin_bit_cast(float, r.s[i]);
    } else {
      const uint32_t word = r.s[i >> 1];
      if constexpr (SK == SK_BF16)
        s[i] = __builtin_bit_cast(float, (i & 1) ? (word & 0xffff0000u) : (word << 16));
      else
        s[i] = f16_bits_to_f32((i & 1) ? (word >> 16) : (word & 0xffffu));
    }
  }
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int e = ((j % NJ) * SPS) / NJ;
    sc[j] = s[e];
    if constexpr (ASYM)
      zp[j] = float(int(int8_t((r.z[0] >> (8 * e)) & 0xff)));
    else
      zp[j] = 0.f;
  }
}

__device__ __forceinline__ float epi_gelu(float x) {  // kernel_ref.h:1570-1572
  return 0.5f * x * (1.f + tanhf(0.7978845834732056f * (x + 0.044714998453855515f * x * x * x)));
}
__device__ __forceinline__ float epi_silu(float x) { return x / (1.f + expf(-x)); }  // kernel_ref.h:1573-1575

// host side: fp16 LUT -> byte planes of F4Lut
inline void f4_lut_planes(const _Float16* lut, F4Lut* out) {
  for (int i = 0; i < 4; i++) out->lo[i] = out->hi[i] = 0;
  for (int e = 0; e < 16; e++) {
    unsigned short bits = __builtin_bit_cast(unsigned short, lut[e]);
    out->lo[e >> 2] |= uint32_t(bits & 0xff) << (8 * (e & 3));
    out->hi[e >> 2] |= uint32_t(bits >> 8) << (8 * (e & 3));
  }
}
}  // namespace ns
