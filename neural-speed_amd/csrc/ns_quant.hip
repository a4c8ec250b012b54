// ns_quant.hip — the offline side on the GPU: quantize, interleave, bit-pack, reduce (bit-exact with the
// reference's CPU pipeline) plus the small element-wise entry points of the operator surface.
//
// Reference pipeline (paths under /root/reference/bestla/bestla):
//   packTransposeWeight/packWeight     bestla_prologue_b.h:180-210
//   quantizeWeight -> quantize_f32_sign_int_rowblock   kernel_ref.h:1608-1719 (always the scalar code: kernel_wrapper.h:511-530)
//                  -> quantize_f32_f4_rowblock         kernel_ref.h:1801-1822
//   packQWeight                        bestla_prologue_b.h:378-398
//     setQuantCorrection               :244-335   (fp32 -> bf16/f16/f32, zero padding)
//     reorderWeight/padding_interleave :490-510, kernel_ref.h:39-57
//     compressWeight                   :606-617, kernel_ref.h:155-365
//     reduceWeight/row_reduce_sum      :455-470, kernel_ref.h:2132-2142
// All fp32 arithmetic below is written with explicit single-rounding intrinsics so that hipcc cannot contract
// a*b+c into an FMA: the CPU reference rounds after every operation.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdlib>

#include "ns_common.h"
#include "ns_route.h"

namespace ns {

namespace {

// ---- x86 float->int conversions as the reference binary performs them ------------------------------------------
// cast<float,int>(x) = int(roundf(x)) (bestla_utils.h:523-526): cvttss2si yields INT_MIN for NaN / out of range.
// roundf (half away from zero), exact: x - trunc(x) is always representable
__device__ __forceinline__ float round_half_away(float x) {
  const float t = truncf(x);
  return (fabsf(__fsub_rn(x, t)) >= 0.5f) ? __fadd_rn(t, copysignf(1.f, x)) : t;
}
__device__ __forceinline__ int cvt_round_int_x86(float v) {
  const float r = round_half_away(v);
  if (!(r >= -2147483648.f && r < 2147483648.f)) return INT_MIN;
  return int(r);
}
// cast<float,int8_t>(x) (bestla_utils.h:507-513): roundf, clamp to [-128,127]; NaN observed to come out as 0
__device__ __forceinline__ int cvt_round_s8_x86(float v) {
  if (v != v) return 0;
  float r = round_half_away(v);
  r = r > 127.f ? 127.f : r;
  r = r < -128.f ? -128.f : r;
  return int(r);
}
__device__ __forceinline__ int wrap_add(int a, int b) { return int(uint32_t(a) + uint32_t(b)); }
__device__ __forceinline__ int clipi(int s, int lo, int hi) { return min(max(s, lo), hi); }

// bf16::fromfloat (bestla_utils.h:146-153)
__device__ __forceinline__ uint16_t f32_to_bf16_ref(float f) {
  uint32_t u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return uint16_t(u >> 16);
}
__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float(uint32_t(h) << 16); }
// fp16::operator=(float) (bestla_utils.h:184-196) — the reference's own bit recipe, not IEEE cvt
__device__ __forceinline__ uint16_t f32_to_f16_ref(float f) {
  const uint32_t b = __float_as_uint(f) + 0x00001000u;
  const uint32_t e = (b & 0x7F800000u) >> 23;
  const uint32_t m = b & 0x007FFFFFu;
  uint32_t r = (b & 0x80000000u) >> 16;
  if (e > 112) r |= (((e - 112) << 10) & 0x7C00u) | (m >> 13);
  if (e < 113 && e > 101) r |= (((0x007FF000u + m) >> (125 - e)) + 1) >> 1;
  if (e > 143) r |= 0x7FFFu;
  return uint16_t(r);
}
// fp16::operator float (bestla_utils.h:197-207)
__device__ __forceinline__ float f16_to_f32_ref(uint16_t x) {
  const uint32_t e = (x & 0x7C00u) >> 10;
  const uint32_t m = (x & 0x03FFu) << 13;
  const uint32_t v = __float_as_uint(float(m)) >> 23;
  uint32_t r = uint32_t(x & 0x8000u) << 16;
  if (e != 0) r |= ((e + 112) << 23) | m;
  if (e == 0 && m != 0) r |= ((v - 37) << 23) | ((m << (150 - v)) & 0x007FE000u);
  return __uint_as_float(r);
}

// ---- f4 code books: thresholds of the reference decision trees flattened to "count thresholds below x" ---------
// (kernel_ref.h:1234-1298, :1373-1413); values bestla_utils.h:749-789
__constant__ float kThrNF4[15] = {-0.8480964004993439f, -0.6106329262256622f,  -0.4599952697753906f,
                                  -0.33967943489551544f, -0.23460740596055984f, -0.13791173323988914f,
                                  -0.045525018125772476f, 0.03979014977812767f, 0.1202552504837513f,
                                  0.2035212516784668f,   0.2920137718319893f,   0.3893125355243683f,
                                  0.5016634166240692f,   0.6427869200706482f,   0.8614784181118011f};
__constant__ int kCodeNF4[16] = {7, 1, 2, 3, 4, 5, 6, 0, 8, 9, 10, 11, 12, 13, 14, 15};
__constant__ float kThrBNB[7] = {0.00260417f, 0.0859375f, 0.20833333f, 0.29166667f, 0.4166667f, 0.583333f, 0.8333333f};
__constant__ int kCodeBNB[8] = {0, 1, 6, 7, 4, 5, 2, 3};
__constant__ float kThrE2M1[7] = {0.03125f / 6, 0.53125f / 6, 1.25f / 6, 1.75f / 6, 2.5f / 6, 3.5f / 6, 5.f / 6};

__device__ __forceinline__ int f4_code(uint32_t t, float x) {
  int c = 0;
  if (t == DT_F4_NF4) {
#pragma unroll
    for (int i = 0; i < 15; i++) c += x > kThrNF4[i];
    return kCodeNF4[c];
  }
  const int sign = x < 0.f ? 8 : 0;
  const float ax = fabsf(x);
  if (t == DT_F4_BNB) {
#pragma unroll
    for (int i = 0; i < 7; i++) c += ax > kThrBNB[i];
    return kCodeBNB[c] + sign;
  }
#pragma unroll
  for (int i = 0; i < 7; i++) c += ax > kThrE2M1[i];
  return c + sign;
}

// ---- fp8 weights (kernel_ref.h:1721-1799) --------------------------------------------------------------------------
// floor(std::log2(float x)) as the reference's host libm evaluates it: log2f rounds to the integer k for the few floats
// just below 2^k, so the floor is k there, not k-1.  The correctly rounded log2f reproduces that (checked against glibc
// for every binade); it is obtained from the double-precision log2.
__device__ __forceinline__ float floor_log2f_libm(float x) { return floorf(float(log2(double(x)))); }
__device__ __forceinline__ float f8_maxnorm(uint32_t t) { return t == DT_F8_E4M3 ? 448.f : 57344.f; }  // get_mxfp_maxnorm
__device__ __forceinline__ int f8_mx_code(float v, float scale, uint32_t t, bool e8m0) {
  const int ebits = t == DT_F8_E4M3 ? 4 : 5, qm = t == DT_F8_E4M3 ? 5 : 4, sm = 7 - ebits;
  v = __fdiv_rn(v, e8m0 ? ldexpf(1.f, int(scale)) : scale);
  float pe = floor_log2f_libm(fabsf(v == 0.f ? __fadd_rn(v, 1.f) : v));
  const float min_exp = float(2 - (1 << (ebits - 1)));
  pe = pe < min_exp ? min_exp : pe;
  // scale so that the kept mantissa bits are the integer part, round half away, scale back: all exact in fp32
  v = ldexpf(v, (qm - 2) - int(pe));
  const float av = fabsf(v), fl = floorf(av);
  const float r = (__fsub_rn(av, fl) >= 0.5f) ? __fadd_rn(fl, 1.f) : fl;
  v = ldexpf(v > 0.f ? r : -r, int(pe) - (qm - 2));
  const float mx = f8_maxnorm(t);
  v = (v < -mx) ? -mx : ((mx < v) ? mx : v);
  uint32_t bits = __float_as_uint(v);
  const uint32_t sign = (bits >> 24) & 0x80u;
  bits <<= 1;
  uint32_t e = ((bits >> 24) - 127u + (1u << (ebits - 1)) - 1u) & 0xffu;
  if (e > (t == DT_F8_E4M3 ? 15u : 31u)) e = 0;
  e = (e << sm) & 0xffu;
  bits <<= 8;
  const uint32_t mmask = (0xffu << (8 - sm)) & 0xffu;  // int8_t(-128 >> (sm - 1)): the top `sm` bits of the byte
  const uint32_t m = ((bits >> 24) & mmask) >> (1 + ebits);
  return int(sign | e | m);
}

struct SrcView {  // fp32 weight, [N][K] when trans (torch layout) else [K][N]
  const float* w;
  size_t ld;
  bool trans;
  __device__ __forceinline__ float at(size_t k, size_t n) const { return trans ? w[n * ld + k] : w[k * ld + n]; }
};

// one thread per (k-block, column).  Codes go to q[k*qsk + n*qsn] (int8), scales/zps to [kb][N].
__global__ void quantize_kernel(SrcView src, size_t n, size_t k, int bs, uint32_t qtype, uint32_t stype, bool asym,
                                int8_t* q, size_t qsk, size_t qsn, float* scales, int8_t* zps) {
  const size_t nblk = (k + bs - 1) / bs;
  const size_t gid = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (gid >= nblk * n) return;
  // consecutive threads walk the contiguous axis of the source
  const size_t kb = src.trans ? gid % nblk : gid / n;
  const size_t col = src.trans ? gid / nblk : gid % n;
  const size_t k0 = kb * bs;
  const int len = int(min(size_t(bs), k - k0));
  const size_t sidx = kb * n + col;
  if (dt_is_f8(qtype)) {  // quantize_f32_f8_rowblock_mxscale, kernel_ref.h:1763-1799
    const bool e8m0 = stype == DT_F8_E8M0;
    float scale = FLT_MIN;
    for (int i = 0; i < len; i++) {
      const float av = fabsf(src.at(k0 + i, col));
      scale = (scale < av) ? av : scale;
    }
    if (e8m0) {  // shared exponent: floor(log2(absmax)) - emax, not below -127
      const float emax = qtype == DT_F8_E4M3 ? 8.f : 15.f;
      scale = __fsub_rn(floor_log2f_libm(scale), emax);
      scale = scale < -127.f ? -127.f : scale;
    } else {
      scale = __fdiv_rn(scale, f8_maxnorm(qtype));
    }
    scales[sidx] = scale;
    for (int i = 0; i < len; i++)
      q[(k0 + i) * qsk + col * qsn] = int8_t(f8_mx_code(src.at(k0 + i, col), scale, qtype, e8m0));
    return;
  }
  if (!dt_is_int(qtype)) {  // kernel_ref.h:1801-1822
    float absmax = FLT_MIN;
    for (int i = 0; i < len; i++) absmax = fmaxf(absmax, fabsf(src.at(k0 + i, col)));
    scales[sidx] = absmax;
    const float r = __fdiv_rn(1.f, absmax);
    for (int i = 0; i < len; i++)
      q[(k0 + i) * qsk + col * qsn] = int8_t(f4_code(qtype, __fmul_rn(src.at(k0 + i, col), r)));
    return;
  }
  const int nbits = dt_bits(qtype);
  const int full = 1 << (nbits - 1);
  const int symv = full - 1;
  if (!asym) {  // kernel_ref.h:1651-1671
    float maxval = FLT_MIN, minval = FLT_MAX, absmax = 0.f;
    for (int i = 0; i < len; i++) {
      const float v = src.at(k0 + i, col);
      // std::max(a,b) = (a<b)?b:a keeps `a` when b is NaN; fmaxf would drop the NaN the same way for finite data
      maxval = (maxval < v) ? v : maxval;
      minval = (v < minval) ? v : minval;
      const float av = fabsf(v);
      absmax = (absmax < av) ? av : absmax;
    }
    float nval = float(symv) + 0.5f;
    const float sum = __fadd_rn(maxval, minval);
    if (fabsf(sum) >= __fdiv_rn(absmax, float(full))) nval = sum > 0.f ? float(-full) : float(full);
    const float scale = __fdiv_rn(absmax, nval);
    const float rscale = __fdiv_rn(1.f, scale);
    scales[sidx] = scale;
    for (int i = 0; i < len; i++) {
      const int c = cvt_round_s8_x86(__fmul_rn(src.at(k0 + i, col), rscale));
      q[(k0 + i) * qsk + col * qsn] = int8_t(clipi(c, -full, symv));
    }
  } else {  // kernel_ref.h:1673-1692
    float maxval = 0.f, minval = 0.f;
    for (int i = 0; i < len; i++) {
      const float v = src.at(k0 + i, col);
      maxval = (maxval < v) ? v : maxval;
      minval = (v < minval) ? v : minval;
    }
    const float scale = __fdiv_rn(__fsub_rn(maxval, minval), float((1 << nbits) - 1));
    const float rscale = __fdiv_rn(1.f, scale);
    scales[sidx] = scale;
    int bzp = wrap_add(cvt_round_int_x86(__fmul_rn(__fsub_rn(0.f, minval), rscale)), -full);
    bzp = clipi(bzp, -full, symv);
    zps[sidx] = int8_t(bzp);
    for (int i = 0; i < len; i++) {
      const int t = wrap_add(cvt_round_int_x86(__fmul_rn(src.at(k0 + i, col), rscale)), bzp);
      q[(k0 + i) * qsk + col * qsn] = int8_t(clipi(t, -full, symv));
    }
  }
}

// scales (fp32 [nblk][N]) -> blob scale section [rows][cstep] in `stype`, zero padded (prologue_b.h:244-281)
__global__ void pack_scales_kernel(const float* __restrict__ s, uint8_t* __restrict__ out, size_t n, size_t rawnk,
                                   size_t rows, size_t cstep, uint32_t stype) {
  const size_t gid = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (gid >= rows * cstep) return;
  const size_t r = gid / cstep, c = gid % cstep;
  const bool live = r < rawnk && c < n;
  const float v = live ? s[r * n + c] : 0.f;
  if (stype == DT_F8_E8M0) {  // static_cast<int8_t>(shared exponent), bestla_prologue_b.h:1179-1195
    out[gid] = live ? uint8_t(int8_t(v)) : uint8_t(0);
  } else if (stype == DT_F32) {
    reinterpret_cast<float*>(out)[gid] = live ? v : 0.f;
  } else {
    uint16_t h = 0;
    if (live) h = (stype == DT_BF16) ? f32_to_bf16_ref(v) : f32_to_f16_ref(v);
    reinterpret_cast<uint16_t*>(out)[gid] = h;
  }
}
__global__ void pack_zps_kernel(const int8_t* __restrict__ z, int8_t* __restrict__ out, size_t n, size_t rawnk,
                                size_t rows, size_t cstep) {
  const size_t gid = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (gid >= rows * cstep) return;
  const size_t r = gid / cstep, c = gid % cstep;
  out[gid] = (r < rawnk && c < n) ? z[r * n + c] : int8_t(0);
}

// codes -> interleaved, bit-packed image.  One thread per 8 consecutive elements of the interleaved image
// [N/NTILE][KPad/PACK][NTILE][PACK]; NTILE*PACK is always a multiple of 8.
struct TiledSrc {
  const int8_t* q;
  size_t sk, sn;  // strides of the canonical code array
  size_t n, k;
  int ntile, packrow, kpad;
  __device__ __forceinline__ int at(size_t e) const {
    const size_t per_tile = size_t(ntile) * kpad;
    const size_t tile = e / per_tile, rem = e % per_tile;
    const size_t kg = rem / (size_t(ntile) * packrow), r2 = rem % (size_t(ntile) * packrow);
    const size_t kk = kg * packrow + r2 % packrow;
    const size_t col = tile * ntile + r2 / packrow;
    return (kk < k && col < n) ? int(q[kk * sk + col * sn]) : 0;
  }
};
__global__ void pack_codes_kernel(TiledSrc src, uint8_t* __restrict__ out, size_t elts, uint32_t qtype) {
  const size_t g8 = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const size_t e0 = g8 * 8;
  if (e0 >= elts) return;
  const int nbits = dt_bits(qtype);
  const bool is_int = dt_is_int(qtype);
  int v[8];
#pragma unroll
  for (int i = 0; i < 8; i++) v[i] = src.at(e0 + i);
  if (nbits == 8) {  // S8 and the fp8 types are stored as they are (bestla_prologue_b.h:389-390, :1118-1119)
#pragma unroll
    for (int i = 0; i < 8; i++) out[e0 + i] = uint8_t(v[i]);
    return;
  }
  const int full = is_int ? (1 << (nbits - 1)) : 0;
  // compress_3bit / compress_1bit read the 5th element of every 8 from src[j + FullRange] (kernel_ref.h:313,:355)
  if (is_int && nbits == 1) v[4] = v[1];
  int u[8];
#pragma unroll
  for (int i = 0; i < 8; i++) u[i] = v[i] + full;
  uint8_t* p = out;
  int sh = 0;
  const bool has4 = !is_int || nbits >= 4;
  const bool has2 = is_int && (nbits == 7 || nbits == 6 || nbits == 3 || nbits == 2);
  const bool has1 = is_int && (nbits == 7 || nbits == 5 || nbits == 3 || nbits == 1);
  if (has4) {
#pragma unroll
    for (int i = 0; i < 4; i++) p[e0 / 2 + i] = uint8_t((u[2 * i] & 0xf) | ((u[2 * i + 1] & 0xf) << 4));
    p += elts / 2;
    sh = 4;
  }
  if (has2) {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      int b = 0;
#pragma unroll
      for (int t = 0; t < 4; t++) b |= ((u[4 * i + t] >> sh) & 0x3) << (2 * t);
      p[e0 / 4 + i] = uint8_t(b);
    }
    p += elts / 4;
    sh += 2;
  }
  if (has1) {
    int b = 0;
#pragma unroll
    for (int t = 0; t < 8; t++) b |= ((u[t] >> sh) & 0x1) << t;
    p[e0 / 8] = uint8_t(b);
  }
}

// reduce[kb][n] = bf16( sum_k float(code - zp) * stored_scale ), sequential fp32 sum (row_reduce_sum).
// The reference's reduceWeight dequantises the blob it has just written (bestla_prologue_b.h:455-470), i.e. the codes as
// the bit planes hold them: the canonical codes for every type except S1, where compress_1bit stores element 1's bit in
// place of element 4's in every group of eight PACKED elements (kernel_ref.h:355).  `s1`: read position p - 3 of the tiled
// image for a position p with p % 8 == 4 (`ts` maps tiled positions back to the canonical array).
__global__ void reduce_kernel(const int8_t* __restrict__ q, size_t sk, size_t sn, const uint8_t* __restrict__ sblob,
                              const int8_t* __restrict__ zblob, uint16_t* __restrict__ rblob, size_t n, size_t k,
                              int bs, size_t cstep, uint32_t stype, TiledSrc ts, bool s1) {
  const size_t nblk = (k + bs - 1) / bs;
  const size_t gid = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (gid >= nblk * n) return;
  const size_t kb = gid / n, col = gid % n;
  float s;
  if (stype == DT_F32)
    s = reinterpret_cast<const float*>(sblob)[kb * cstep + col];
  else if (stype == DT_BF16)
    s = bf16_to_f32(reinterpret_cast<const uint16_t*>(sblob)[kb * cstep + col]);
  else
    s = f16_to_f32_ref(reinterpret_cast<const uint16_t*>(sblob)[kb * cstep + col]);
  const int z = zblob ? zblob[kb * cstep + col] : 0;
  float tmp = 0.f;
  const size_t kend = min(k, (kb + 1) * size_t(bs));
  for (size_t kk = kb * bs; kk < kend; kk++) {
    int code = int(q[kk * sk + col * sn]);
    if (s1) {
      const size_t p = (col / ts.ntile) * size_t(ts.ntile) * ts.kpad + (kk / ts.packrow) * size_t(ts.ntile) * ts.packrow +
                       (col % ts.ntile) * ts.packrow + kk % ts.packrow;
      if ((p & 7) == 4) code = ts.at(p - 3);
    }
    tmp = __fadd_rn(tmp, __fmul_rn(float(code - z), s));
  }
  rblob[kb * cstep + col] = f32_to_bf16_ref(tmp);
}

template <typename F>
inline dim3 grid1d(size_t n, F bs) {
  return dim3((unsigned)((n + bs - 1) / bs));
}

hipError_t pack_sections(const int8_t* q, size_t sk, size_t sn, const float* scales, const int8_t* zps, size_t n,
                         size_t k, int bs, uint32_t qtype, uint32_t stype, int ref_ntile, int ref_packrow, int ref_kpad,
                         int ref_npad, int cstep, bool has_reduce, uint8_t* q_out, uint8_t* s_out, int8_t* z_out,
                         uint16_t* r_out, hipStream_t st) {
  const size_t rawnk = (k + bs - 1) / bs;
  const size_t rows = (size_t(ref_kpad) + bs - 1) / bs;
  const size_t nsc = rows * cstep;
  hipLaunchKernelGGL(pack_scales_kernel, grid1d(nsc, 256), dim3(256), 0, st, scales, s_out, n, rawnk, rows,
                     size_t(cstep), stype);
  if (z_out)
    hipLaunchKernelGGL(pack_zps_kernel, grid1d(nsc, 256), dim3(256), 0, st, zps, z_out, n, rawnk, rows, size_t(cstep));
  TiledSrc ts{q, sk, sn, n, k, ref_ntile, ref_packrow, ref_kpad};
  const size_t elts = size_t(ref_npad) * ref_kpad;
  hipLaunchKernelGGL(pack_codes_kernel, grid1d(elts / 8, 256), dim3(256), 0, st, ts, q_out, elts, qtype);
  if (has_reduce)
    hipLaunchKernelGGL(reduce_kernel, grid1d(rawnk * n, 256), dim3(256), 0, st, q, sk, sn, (const uint8_t*)s_out,
                       (const int8_t*)z_out, r_out, n, k, bs, size_t(cstep), stype, ts, dt_is_int(qtype) && dt_bits(qtype) == 1);
  return hipGetLastError();
}

}  // namespace

hipError_t launch_quant_pack(const QuantArgs& a, hipStream_t st) {
  const size_t nblk = (a.k + a.blocksize - 1) / a.blocksize;
  int8_t* q = nullptr;
  float* sc = nullptr;
  int8_t* zp = nullptr;
  auto release = [&]() {
    if (q) hipFreeAsync(q, st);
    if (sc) hipFreeAsync(sc, st);
    if (zp) hipFreeAsync(zp, st);
  };
  hipError_t e;
  if ((e = hipMallocAsync((void**)&q, a.n * a.k, st)) != hipSuccess ||
      (e = hipMallocAsync((void**)&sc, nblk * a.n * sizeof(float), st)) != hipSuccess ||
      (a.asym && (e = hipMallocAsync((void**)&zp, nblk * a.n, st)) != hipSuccess)) {
    release();
    return e;
  }
  // canonical code array follows the source's contiguous axis so both reads and writes coalesce
  const size_t qsk = a.is_trans ? 1 : a.n, qsn = a.is_trans ? a.k : 1;
  SrcView sv{a.w, a.ld, a.is_trans};
  hipLaunchKernelGGL(quantize_kernel, grid1d(nblk * a.n, 256), dim3(256), 0, st, sv, a.n, a.k, a.blocksize, a.qtype,
                     a.stype, a.asym, q, qsk, qsn, sc, zp);
  e = pack_sections(q, qsk, qsn, sc, zp, a.n, a.k, a.blocksize, a.qtype, a.stype, a.ref_ntile, a.ref_packrow,
                    a.ref_kpad, a.ref_npad, a.cstep, a.has_reduce, a.q_out, a.s_out, a.asym ? a.z_out : nullptr,
                    a.r_out, st);
  release();
  return e;
}

hipError_t launch_pack_q(const PackQArgs& a, hipStream_t st) {
  return pack_sections(a.q, a.ldq, 1, a.scales, a.zps, a.n, a.k, a.blocksize, a.qtype, a.stype, a.ref_ntile,
                       a.ref_packrow, a.ref_kpad, a.ref_npad, a.cstep, a.has_reduce, a.q_out, a.s_out,
                       a.zps ? a.z_out : nullptr, a.r_out, st);
}

// ============================================================================================================
// element-wise members of the operator surface (ne_bestla.h:71-75)
// ============================================================================================================
// layernorm / rmsnorm — kernel_ref.h:2199-2245 semantics (no scale/bias at this entry: BTLALayerNorm passes nullptr)
// optional fusions for a device-resident layer: gamma (the ne_mul by the norm weight that always follows, llama.cpp) and
// an fp16 shadow of the result for the next GEMV's activations.
// One workgroup per row.  Rows of up to 256 * 16 floats (every decoder width up to 4096) are read ONCE with all loads
// in flight together and kept in registers; a single-workgroup kernel is latency-bound, a load-per-iteration loop took
// 16 us per 4096-wide row.
template <bool CACHED>
__global__ __launch_bounds__(256) void norm_kernel(const float* __restrict__ in, float* __restrict__ out, int size,
                                                   bool rms, float eps, const float* __restrict__ gamma,
                                                   _Float16* __restrict__ out16, float* __restrict__ out_plain = nullptr) {
  // out_plain (with gamma): the normalised values BEFORE the gamma product as well — the two tensors of the reference's norm and
  // mul nodes from one launch (ns_device.hip: lazy peephole of the device route)
  __shared__ float red[2][4];
  const float* src = in + size_t(blockIdx.x) * size;
  float* dst = out + size_t(blockIdx.x) * size;
  float4 v[4];
  float s = 0.f, s2 = 0.f;
  if constexpr (CACHED) {  // size % 4 == 0, size <= 4096, 16-byte aligned rows
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const int i = (c * 256 + threadIdx.x) * 4;
      v[c] = i < size ? *reinterpret_cast<const float4*>(src + i) : float4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int c = 0; c < 4; c++) {
      s += (v[c].x + v[c].y) + (v[c].z + v[c].w);
      s2 += (v[c].x * v[c].x + v[c].y * v[c].y) + (v[c].z * v[c].z + v[c].w * v[c].w);
    }
  } else {
    for (int i = threadIdx.x; i < size; i += 256) {
      const float x = src[i];
      s += x;
      s2 += x * x;
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o);
    s2 += __shfl_xor(s2, o);
  }
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = s;
    red[1][threadIdx.x >> 6] = s2;
  }
  __syncthreads();
  s = red[0][0] + red[0][1] + red[0][2] + red[0][3];
  s2 = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  const float mean = s / size;
  const float var = rms ? sqrtf(s2 / size + eps) : sqrtf(s2 / size - mean * mean + eps);
  const float inv = 1.f / var;
  auto fin = [&](float x, int i) {
    float y = rms ? x * inv : (x - mean) * inv;
    if (gamma) y = y * gamma[i];  // separate rounding, like the reference's two operators
    return y;
  };
  if constexpr (CACHED) {
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const int i = (c * 256 + threadIdx.x) * 4;
      if (i >= size) continue;
      const float4 y = {fin(v[c].x, i), fin(v[c].y, i + 1), fin(v[c].z, i + 2), fin(v[c].w, i + 3)};
      if (out) *reinterpret_cast<float4*>(dst + i) = y;  // (round 5: the fp16 shadow alone where the consumer is this library's next GEMM)
      if (out_plain) {
        const float4 yp = rms ? float4{v[c].x * inv, v[c].y * inv, v[c].z * inv, v[c].w * inv}
                              : float4{(v[c].x - mean) * inv, (v[c].y - mean) * inv, (v[c].z - mean) * inv, (v[c].w - mean) * inv};
        *reinterpret_cast<float4*>(out_plain + size_t(blockIdx.x) * size + i) = yp;
      }
      if (out16) {
        typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
        *reinterpret_cast<half4_t*>(out16 + size_t(blockIdx.x) * size + i) =
            half4_t{(_Float16)y.x, (_Float16)y.y, (_Float16)y.z, (_Float16)y.w};
      }
    }
  } else {
    for (int i = threadIdx.x; i < size; i += 256) {
      const float y = fin(src[i], i);
      if (out) dst[i] = y;
      if (out16) out16[size_t(blockIdx.x) * size + i] = (_Float16)y;
      if (out_plain) out_plain[size_t(blockIdx.x) * size + i] = rms ? src[i] * inv : (src[i] - mean) * inv;
    }
  }
}
hipError_t launch_rmsnorm(int norm_count, int norm_size, bool isrms, float eps, const float* in, float* out,
                          hipStream_t st, const float* gamma, void* out16) {
  const bool cached = norm_size <= 4096 && (norm_size & 3) == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0 &&
                      (reinterpret_cast<uintptr_t>(out) & 15) == 0 && (reinterpret_cast<uintptr_t>(out16) & 7) == 0 &&
                      (!gamma || (reinterpret_cast<uintptr_t>(gamma) & 3) == 0);
  if (cached)
    hipLaunchKernelGGL(norm_kernel<true>, dim3(norm_count), dim3(256), 0, st, in, out, norm_size, isrms, eps, gamma,
                       static_cast<_Float16*>(out16));
  else
    hipLaunchKernelGGL(norm_kernel<false>, dim3(norm_count), dim3(256), 0, st, in, out, norm_size, isrms, eps, gamma,
                       static_cast<_Float16*>(out16));
  return hipGetLastError();
}

// DQ8_BNB scale codes -> fp32 scales (ns_api.cpp dq8_expand)
__global__ void dq8_expand_kernel(const uint8_t* __restrict__ codes, const float* __restrict__ dq, const float* __restrict__ lut,
                                  float* __restrict__ out, int rows, int cstep, int n, int dq_blocksize, uint32_t dq_last) {
  const size_t gid = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (gid >= size_t(rows) * cstep) return;
  const int r = int(gid / cstep), c = int(gid % cstep);
  const size_t b = (size_t(r) * n + c) / size_t(dq_blocksize);
  // separately rounded product and sum, like the reference's scalar expression (this file is compiled with -ffp-contract=off: a fused
  // multiply-add differs in the last bit for one scale in five); padded columns (c >= n) take any block: never used
  out[gid] = __fadd_rn(__fmul_rn(lut[codes[gid]], dq[b < dq_last ? b : dq_last]), dq[dq_last]);
}
hipError_t launch_dq8_expand(const uint8_t* codes, const float* dq, const float* lut, float* out, int rows, int cstep, int n, int dq_blocksize,
                             uint32_t dq_last, hipStream_t st) {
  const size_t total = size_t(rows) * cstep;
  if (!total) return hipSuccess;
  hipLaunchKernelGGL(dq8_expand_kernel, dim3(unsigned((total + 255) / 256)), dim3(256), 0, st, codes, dq, lut, out, rows, cstep, n, dq_blocksize,
                     dq_last);
  return hipGetLastError();
}

// norm and the product with the norm weight in one launch, BOTH tensors written (plain = norm(in), out = plain * gamma): same values
// as launch_rmsnorm(in -> plain) followed by the row-broadcast multiply
hipError_t launch_rmsnorm_mul2(int norm_count, int norm_size, bool isrms, float eps, const float* in, float* plain, const float* gamma,
                               float* out, hipStream_t st) {
  const bool cached = norm_size <= 4096 && (norm_size & 3) == 0 && ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out) |
                                                                   reinterpret_cast<uintptr_t>(plain)) & 15) == 0 &&
                      (reinterpret_cast<uintptr_t>(gamma) & 3) == 0;
  if (cached)
    hipLaunchKernelGGL(norm_kernel<true>, dim3(norm_count), dim3(256), 0, st, in, out, norm_size, isrms, eps, gamma, nullptr, plain);
  else
    hipLaunchKernelGGL(norm_kernel<false>, dim3(norm_count), dim3(256), 0, st, in, out, norm_size, isrms, eps, gamma, nullptr, plain);
  return hipGetLastError();
}
// silu and the product that follows it in a gated FFN, both tensors written: s = silu(x) (silu_kernel's arithmetic), out = s * y
__global__ void silu_mul2_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ s, float* __restrict__ out,
                                 size_t n, int silu_first) {
  const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = __fdiv_rn(x[i], __fadd_rn(1.0f, expf(-x[i])));
  s[i] = v;
  out[i] = silu_first ? v * y[i] : y[i] * v;
}
hipError_t launch_silu_mul2(const float* x, const float* y, float* s, float* out, size_t n, int silu_first, hipStream_t st) {
  if (!n) return hipSuccess;
  hipLaunchKernelGGL(silu_mul2_kernel, grid1d(n, 256), dim3(256), 0, st, x, y, s, out, n, silu_first);
  return hipGetLastError();
}

// Producer side of a carried RMS norm (ns_norm_link) for a tensor no GEMM produced: x16 = fp16(x * gamma), and per row and
// 16-column tile the sum of x^2, in the layout and summation order of gemv_kernel's epilogue.
__global__ __launch_bounds__(256) void norm_prep_kernel(int m, int n, const float* __restrict__ x, int ldx,
                                                        const float* __restrict__ gamma, _Float16* __restrict__ x16,
                                                        float* __restrict__ ssq, int ssq_stride) {
  const int row = blockIdx.y;
  const int col = blockIdx.x * 256 + threadIdx.x;
  const bool ok = col < n;
  const float v = ok ? x[size_t(row) * ldx + col] : 0.f;
  if (ok && x16) x16[size_t(row) * ldx + col] = (_Float16)(gamma ? v * gamma[col] : v);
  float t = v * v;
  t += __shfl_xor(t, 1, 64);
  t += __shfl_xor(t, 2, 64);
  t += __shfl_xor(t, 4, 64);
  t += __shfl_xor(t, 8, 64);
  if ((threadIdx.x & 15) == 0 && ok && ssq) ssq[size_t(row) * ssq_stride + (col >> 4)] = t;
}
hipError_t launch_norm_prep(int m, int n, const float* x, int ldx, const float* gamma, void* x16, float* ssq,
                            int ssq_stride, hipStream_t st) {
  if (m <= 0 || n <= 0) return hipSuccess;
  hipLaunchKernelGGL(norm_prep_kernel, dim3(unsigned((n + 255) / 256), unsigned(m)), dim3(256), 0, st, m, n, x, ldx, gamma,
                     static_cast<_Float16*>(x16), ssq, ssq_stride);
  return hipGetLastError();
}

__global__ void bcast_kernel(const float* __restrict__ t, const float* __restrict__ v, float* __restrict__ out,
                             size_t total, int vsize, int vstep, bool mul) {
  const size_t gid = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (gid >= total) return;
  const size_t b = gid / vsize, i = gid % vsize;
  const float x = t[gid], y = v[b * vstep + i];
  out[gid] = mul ? x * y : x + y;
}
// ---- activation side of the int8-compute path: quantize_fp_u8_colblock (kernel_ref.h:1824-1883) ------------------
// per (row, k-block): min / max including 0 (a full block starts max at FLT_MIN, a tail block at 0 — :1832 vs :1857),
// scale = (max - min) / 255, zp = u8(-min / scale + 0.5), q = u8(zp + round(a / scale) + 0.5 truncated), optional block
// sum * scale.  One thread per (row, k-block); every operation rounds like the scalar reference (no FMA contraction).
__device__ __forceinline__ int cast_f32_u8_x86(float v) {  // bestla_utils.h:515-521: +0.5, clamp, truncate; NaN -> 0
  if (v != v) return 0;
  v = __fadd_rn(v, 0.5f);
  v = v > 255.f ? 255.f : v;
  v = v < 0.f ? 0.f : v;
  return int(v);
}
__global__ void aquant_u8_kernel(int row, int col, const float* __restrict__ src, int ld_src, uint8_t* __restrict__ dst,
                                 int ld_dst, float* __restrict__ scales, int ld_scale, uint8_t* __restrict__ zps,
                                 int blocksize, float* __restrict__ blkreduce) {
  const int nblk = (col + blocksize - 1) / blocksize;
  const size_t gid = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (gid >= size_t(row) * nblk) return;
  const int i = int(gid / nblk), kb = int(gid % nblk);
  const int j = kb * blocksize;
  const bool tail = j + blocksize > col;
  const int bs = tail ? col - j : blocksize;
  const float* s = src + size_t(i) * ld_src + j;
  float maxval = tail ? 0.f : FLT_MIN, minval = 0.f;
  for (int ij = 0; ij < bs; ij++) {
    const float f = s[ij];
    maxval = f > maxval ? f : maxval;  // std::max(f, maxval): NaN in f keeps maxval
    minval = f < minval ? f : minval;
  }
  const float scale = __fdiv_rn(__fsub_rn(maxval, minval), 255.f);
  const int zp = cast_f32_u8_x86(__fdiv_rn(__fsub_rn(0.f, minval), scale));
  const float rscale = __fdiv_rn(1.f, scale);
  scales[size_t(i) * ld_scale + kb] = scale;
  zps[size_t(i) * ld_scale + kb] = uint8_t(zp);
  int sum = 0;
  const float zpf = float(zp);
  uint8_t* d = dst + size_t(i) * ld_dst + j;
  for (int ij = 0; ij < bs; ij++) {
    const int qtmp = cvt_round_int_x86(__fmul_rn(s[ij], rscale));
    sum = wrap_add(sum, qtmp);
    d[ij] = uint8_t(cast_f32_u8_x86(__fadd_rn(zpf, float(qtmp))));
  }
  if (blkreduce) blkreduce[size_t(i) * ld_scale + kb] = __fmul_rn(float(sum), scale);
}
// The same arithmetic with LPB lanes per (row, k-block): lane e takes elements e, e + LPB, ... (coalesced), the block's
// max / min and the integer code sum are combined by xor-shuffles.  All three are order-independent (max / min never
// select a NaN, the sum wraps in two's complement), every element is quantized by exactly the operations of the
// one-thread form above, so the results are bit-identical; only the latency of a decode-sized call changes (one
// thread walking 32 dependent loads -> 4).
template <int LPB>
__global__ void aquant_u8_coop_kernel(int row, int col, const float* __restrict__ src, int ld_src, uint8_t* __restrict__ dst,
                                      int ld_dst, float* __restrict__ scales, int ld_scale, uint8_t* __restrict__ zps,
                                      int blocksize, float* __restrict__ blkreduce) {
  const int nblk = (col + blocksize - 1) / blocksize;
  const size_t gid = (size_t(blockIdx.x) * blockDim.x + threadIdx.x) / LPB;
  const int e = threadIdx.x % LPB;
  const bool live = gid < size_t(row) * nblk;  // whole lane groups are live or dead together (blockDim % LPB == 0)
  const int i = live ? int(gid / nblk) : 0, kb = live ? int(gid % nblk) : 0;
  const int j = kb * blocksize;
  const bool tail = j + blocksize > col;
  const int bs = live ? (tail ? col - j : blocksize) : 0;
  const float* s = src + size_t(i) * ld_src + j;
  float maxval = tail ? 0.f : FLT_MIN, minval = 0.f;
  for (int ij = e; ij < bs; ij += LPB) {
    const float f = s[ij];
    maxval = f > maxval ? f : maxval;
    minval = f < minval ? f : minval;
  }
#pragma unroll
  for (int d = 1; d < LPB; d <<= 1) {
    const float om = __shfl_xor(maxval, d), on = __shfl_xor(minval, d);
    maxval = om > maxval ? om : maxval;
    minval = on < minval ? on : minval;
  }
  const float scale = __fdiv_rn(__fsub_rn(maxval, minval), 255.f);
  const int zp = cast_f32_u8_x86(__fdiv_rn(__fsub_rn(0.f, minval), scale));
  const float rscale = __fdiv_rn(1.f, scale);
  int sum = 0;
  const float zpf = float(zp);
  uint8_t* d = dst + size_t(i) * ld_dst + j;
  for (int ij = e; ij < bs; ij += LPB) {
    const int qtmp = cvt_round_int_x86(__fmul_rn(s[ij], rscale));
    sum = wrap_add(sum, qtmp);
    d[ij] = uint8_t(cast_f32_u8_x86(__fadd_rn(zpf, float(qtmp))));
  }
#pragma unroll
  for (int dd = 1; dd < LPB; dd <<= 1) sum = wrap_add(sum, __shfl_xor(sum, dd));
  if (live && e == 0) {
    scales[size_t(i) * ld_scale + kb] = scale;
    zps[size_t(i) * ld_scale + kb] = uint8_t(zp);
    if (blkreduce) blkreduce[size_t(i) * ld_scale + kb] = __fmul_rn(float(sum), scale);
  }
}
// GEMM-sized calls: eight lanes per (row, k-block) again, but lane e takes the block's elements [e bs/8, (e + 1) bs/8) as
// 16-byte loads and writes its codes as dwords (the form above stores single bytes).  The same per-element operations and
// order-independent reductions: bit-identical.  Optionally the operand the int8-reference GEMM multiplies
// (ns_i8ref.hip i8mfma2_kernel) next to the codes: ap[r][k] = fp16(code - zp), with ap_scale16 (nibble containers) / 16 for
// k mod 8 in {2, 3, 6, 7} (i8prep_kernel's output), so that kernel is not launched.
// Needs blocksize % 32 == 0, col % blocksize == 0 and 16-byte / 4-byte aligned rows.
typedef _Float16 aq_half2 __attribute__((ext_vector_type(2)));
template <bool AP>
__global__ __launch_bounds__(256) void aquant_u8_vec_kernel(int row, int col, const float* __restrict__ src, int ld_src,
                                                            uint8_t* __restrict__ dst, int ld_dst, float* __restrict__ scales, int ld_scale,
                                                            uint8_t* __restrict__ zps, int blocksize, uint16_t* __restrict__ ap, int ld_ap,
                                                            int ap_scale16) {
  const int nblk = col / blocksize;
  const size_t gid = (size_t(blockIdx.x) * 256 + threadIdx.x) >> 3;
  const int e = threadIdx.x & 7;
  const bool live = gid < size_t(row) * nblk;  // whole lane groups are live or dead together
  const int i = live ? int(gid / nblk) : 0, kb = live ? int(gid % nblk) : 0;
  const int per = blocksize >> 3;  // elements of this lane: a multiple of four
  const int j = kb * blocksize + e * per;
  const float* s = src + size_t(i) * ld_src + j;
  float maxval = FLT_MIN, minval = 0.f;
  if (live)
    for (int ij = 0; ij < per; ij += 4) {
      const float4 f = *reinterpret_cast<const float4*>(s + ij);
      const float fv[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
      for (int t = 0; t < 4; t++) {
        maxval = fv[t] > maxval ? fv[t] : maxval;  // std::max(f, maxval): NaN in f keeps maxval
        minval = fv[t] < minval ? fv[t] : minval;
      }
    }
#pragma unroll
  for (int d = 1; d < 8; d <<= 1) {
    const float om = __shfl_xor(maxval, d), on = __shfl_xor(minval, d);
    maxval = om > maxval ? om : maxval;
    minval = on < minval ? on : minval;
  }
  if (!live) return;
  const float scale = __fdiv_rn(__fsub_rn(maxval, minval), 255.f);
  const int zp = cast_f32_u8_x86(__fdiv_rn(__fsub_rn(0.f, minval), scale));
  const float rscale = __fdiv_rn(1.f, scale);
  const float zpf = float(zp);
  uint8_t* d = dst + size_t(i) * ld_dst + j;
  for (int ij = 0; ij < per; ij += 4) {
    const float4 f = *reinterpret_cast<const float4*>(s + ij);  // (L1 / L2: the block was just read)
    const float fv[4] = {f.x, f.y, f.z, f.w};
    int c[4];
#pragma unroll
    for (int t = 0; t < 4; t++) c[t] = cast_f32_u8_x86(__fadd_rn(zpf, float(cvt_round_int_x86(__fmul_rn(fv[t], rscale)))));
    *reinterpret_cast<uint32_t*>(d + ij) = uint32_t(c[0]) | (uint32_t(c[1]) << 8) | (uint32_t(c[2]) << 16) | (uint32_t(c[3]) << 24);
    if constexpr (AP) {  // small integers: exact in fp16, and so is the division by 16
      const aq_half2 lo = {(_Float16)float(c[0] - zp), (_Float16)float(c[1] - zp)};
      const float hs = ap_scale16 ? 0.0625f : 1.f;
      const aq_half2 hi = {(_Float16)(float(c[2] - zp) * hs), (_Float16)(float(c[3] - zp) * hs)};
      *reinterpret_cast<uint2*>(ap + size_t(i) * ld_ap + j + ij) = uint2{__builtin_bit_cast(uint32_t, lo), __builtin_bit_cast(uint32_t, hi)};
    }
  }
  if (e == 0) {
    scales[size_t(i) * ld_scale + kb] = scale;
    zps[size_t(i) * ld_scale + kb] = uint8_t(zp);
  }
}
// codes + scales + zero points, and (ap != nullptr) the fp16 operand of i8mfma2_kernel with row stride ld_ap halves;
// hipErrorNotSupported when the shape is outside the vector kernel's envelope (the caller then takes launch_aquant_u8)
hipError_t launch_aquant_u8_vec(int row, int col, const float* src, int ld_src, uint8_t* dst, int ld_dst, float* scales,
                                int ld_scale, uint8_t* zps, int blocksize, void* ap, int ld_ap, bool ap_scale16, hipStream_t st) {
  if (blocksize <= 0 || blocksize % 32 != 0 || col % blocksize != 0 || (ld_src & 3) || (reinterpret_cast<uintptr_t>(src) & 15) ||
      (ld_dst & 3) || (reinterpret_cast<uintptr_t>(dst) & 3) || (ap && ((ld_ap & 3) || (reinterpret_cast<uintptr_t>(ap) & 7))))
    return hipErrorNotSupported;
  const size_t total = size_t(row) * (col / blocksize);
  if (total == 0) return hipSuccess;
  if (ap)
    hipLaunchKernelGGL(aquant_u8_vec_kernel<true>, grid1d(total * 8, 256), dim3(256), 0, st, row, col, src, ld_src, dst, ld_dst, scales,
                       ld_scale, zps, blocksize, static_cast<uint16_t*>(ap), ld_ap, ap_scale16 ? 1 : 0);
  else
    hipLaunchKernelGGL(aquant_u8_vec_kernel<false>, grid1d(total * 8, 256), dim3(256), 0, st, row, col, src, ld_src, dst, ld_dst, scales,
                       ld_scale, zps, blocksize, static_cast<uint16_t*>(nullptr), 0, 0);
  return hipGetLastError();
}

hipError_t launch_aquant_u8(int row, int col, const float* src, int ld_src, uint8_t* dst, int ld_dst, float* scales,
                            int ld_scale, uint8_t* zps, int blocksize, float* blkreduce, hipStream_t st) {
  const int nblk = (col + blocksize - 1) / blocksize;
  const size_t total = size_t(row) * nblk;
  if (total == 0) return hipSuccess;
  static const bool one_thread = getenv("NS_AQUANT_SERIAL") != nullptr;  // diagnostics: the one-thread-per-block form
  if (one_thread) {
    hipLaunchKernelGGL(aquant_u8_kernel, grid1d(total, 128), dim3(128), 0, st, row, col, src, ld_src, dst, ld_dst, scales,
                       ld_scale, zps, blocksize, blkreduce);
  } else if (blocksize >= 512) {  // per-channel / very wide groups: a whole wave per block
    hipLaunchKernelGGL(aquant_u8_coop_kernel<64>, grid1d(total * 64, 256), dim3(256), 0, st, row, col, src, ld_src, dst,
                       ld_dst, scales, ld_scale, zps, blocksize, blkreduce);
  } else {
    hipLaunchKernelGGL(aquant_u8_coop_kernel<8>, grid1d(total * 8, 256), dim3(256), 0, st, row, col, src, ld_src, dst,
                       ld_dst, scales, ld_scale, zps, blocksize, blkreduce);
  }
  return hipGetLastError();
}

// ---- RoPE: ne_compute_forward_rope_f32 (ne_layers.c:9243-9428), modes 0 and 2 (NeoX), ext_factor == 0 -------------
// one thread per rotated pair; theta = p * theta_scale^idx built by the same sequential fp32 products as the reference
// (this file is compiled with -ffp-contract=off), so only cosf / sinf differ from the CPU libm in the last ulp
// YaRN (rope_yarn / rope_yarn_ramp, ne_layers.c:9196-9217): theta = interp * (1 - mix) + extrap * mix with
// mix = (1 - clamp((i0 / 2 - corr0) / max(0.001, corr1 - corr0), 0, 1)) * ext_factor; i0 is the element index in mode 0
// and (int)(-ic / n_dims - ib) in the NeoX loop (:9399-9405), restated as written.  mscale already carries the
// 1 + 0.1 * log(1 / freq_scale) factor (host libm, like the reference).
struct RopeYarn {
  float ext_factor, corr0, corr1;
  const float* lr_factor;  // long-rope (mode 0x10, ne_layers.c:9349-9377): per-pair divisor of theta, else null
  float lr_scale;          // and the factor applied to cos / sin
};
__device__ __forceinline__ float rope_theta(float theta_extrap, float freq_scale, int i0, const RopeYarn& y) {
  const float interp = __fmul_rn(freq_scale, theta_extrap);
  if (y.ext_factor == 0.f) return interp;
  const float t = __fdiv_rn(__fsub_rn(float(i0 / 2), y.corr0), fmaxf(0.001f, __fsub_rn(y.corr1, y.corr0)));
  const float ramp = __fsub_rn(1.0f, fminf(1.0f, fmaxf(0.0f, t)));
  const float mix = __fmul_rn(ramp, y.ext_factor);
  return __fadd_rn(__fmul_rn(interp, __fsub_rn(1.f, mix)), __fmul_rn(theta_extrap, mix));
}
__global__ void rope_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t rows, int heads, int seq,
                            int head_size, int n_past, int n_dims, int neox, float theta_scale, float freq_scale,
                            float attn_factor, RopeYarn yarn, const int* __restrict__ kmove, int kdelta) {
  if (kmove) n_past += kdelta * *kmove;  // replayed device route: the position moves with the graph's token counter (ns_common.h Affine)
  const int half = head_size / 2;  // pairs per row in both modes ((head_size / n_dims) * (n_dims / 2) for NeoX)
  const size_t gid = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  const size_t npairs = neox ? size_t(head_size / n_dims) * (n_dims / 2) : size_t(half);
  if (gid >= rows * npairs) return;
  const size_t row = gid / npairs;
  const int pr = int(gid % npairs);
  const int i2 = int((row / heads) % seq);
  float theta_base = float(n_past + i2);
  if (neox) theta_base = __fmul_rn(theta_base, freq_scale);
  for (int t = 0; t < pr; t++) theta_base = __fmul_rn(theta_base, theta_scale);
  int ia, ib, i0;
  if (neox) {
    const int blk = pr / (n_dims / 2), ic = pr % (n_dims / 2);
    ia = blk * n_dims + ic;
    ib = ia + n_dims / 2;
    i0 = int(__fsub_rn(__fmul_rn(__fdiv_rn(-1.f, float(n_dims)), float(2 * ic)), float(blk)));  // (int)cur_rot
  } else {
    ia = 2 * pr;
    ib = ia + 1;
    i0 = ia;
  }
  float c, s;
  if (yarn.lr_factor) {  // NeoX indexing; theta_base / factor[ic / 2] goes through rope_yarn, then scale_factor
    const float theta = rope_theta(__fdiv_rn(theta_base, yarn.lr_factor[pr % (n_dims / 2)]), freq_scale, i0, yarn);
    c = __fmul_rn(__fmul_rn(cosf(theta), attn_factor), yarn.lr_scale);
    s = __fmul_rn(__fmul_rn(sinf(theta), attn_factor), yarn.lr_scale);
  } else {
    const float theta = rope_theta(theta_base, freq_scale, i0, yarn);
    c = __fmul_rn(cosf(theta), attn_factor);
    s = __fmul_rn(sinf(theta), attn_factor);
  }
  const float* x = src + row * head_size;
  float* y = dst + row * head_size;
  const float x0 = x[ia], x1 = x[ib];
  y[ia] = __fsub_rn(__fmul_rn(x0, c), __fmul_rn(x1, s));
  y[ib] = __fadd_rn(__fmul_rn(x0, s), __fmul_rn(x1, c));
}
// RoPE of Q (in place) and K fused with the kv-cache append: K is rotated and stored as fp16 at cache position
// n_past + i, V is converted and stored — the three operators ne_rope(q), ne_rope(k), kv-cache cpy of the llama graph
// (models/llama/llama.cpp:232-262) in one launch.  batch 1; cache element strides are given per position and per head.
__global__ void rope_qkv_append_kernel(float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                       _Float16* __restrict__ kc, _Float16* __restrict__ vc, int seq, int heads, int heads_kv,
                                       int head_size, int n_past, int n_dims, int neox, float theta_scale,
                                       float freq_scale, float attn_factor, long long c_sl, long long c_head) {
  const size_t npairs = neox ? size_t(head_size / n_dims) * (n_dims / 2) : size_t(head_size / 2);
  const size_t nq = size_t(seq) * heads * npairs, nk = size_t(seq) * heads_kv * npairs;
  const size_t nv = size_t(seq) * heads_kv * head_size;
  const size_t gid = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (gid < nq + nk) {
    const bool is_k = gid >= nq;
    const size_t g = is_k ? gid - nq : gid;
    const int hn = is_k ? heads_kv : heads;
    const size_t row = g / npairs;  // (i2, head)
    const int pr = int(g % npairs), i2 = int(row / hn), ih = int(row % hn);
    float theta_base = float(n_past + i2);
    if (neox) theta_base = __fmul_rn(theta_base, freq_scale);
    for (int t = 0; t < pr; t++) theta_base = __fmul_rn(theta_base, theta_scale);
    const float theta = __fmul_rn(freq_scale, theta_base);
    const float c = __fmul_rn(cosf(theta), attn_factor), s = __fmul_rn(sinf(theta), attn_factor);
    int ia, ib;
    if (neox) {
      const int blk = pr / (n_dims / 2), ic = pr % (n_dims / 2);
      ia = blk * n_dims + ic;
      ib = ia + n_dims / 2;
    } else {
      ia = 2 * pr;
      ib = ia + 1;
    }
    const float* x = (is_k ? k : q) + row * head_size;
    const float x0 = x[ia], x1 = x[ib];
    const float y0 = __fsub_rn(__fmul_rn(x0, c), __fmul_rn(x1, s)), y1 = __fadd_rn(__fmul_rn(x0, s), __fmul_rn(x1, c));
    if (is_k) {
      _Float16* d = kc + (long long)(n_past + i2) * c_sl + (long long)ih * c_head;
      d[ia] = (_Float16)y0;
      d[ib] = (_Float16)y1;
    } else {
      q[row * head_size + ia] = y0;
      q[row * head_size + ib] = y1;
    }
  } else if (gid < nq + nk + nv) {
    const size_t g = gid - nq - nk;
    const int e = int(g % head_size);
    const size_t row = g / head_size;
    const int i2 = int(row / heads_kv), ih = int(row % heads_kv);
    vc[(long long)(n_past + i2) * c_sl + (long long)ih * c_head + e] = (_Float16)v[g];
  }
}
// (cos, sin) * attn_factor of the mode-0 pairs of positions n_past .. n_past + m - 1: the angle arithmetic of
// rope_qkv_append_kernel, once per token for all layers (ns_qkv_rope)
__global__ void rope_cos_sin_kernel(int m, int n_past, int npairs, float theta_scale, float freq_scale, float attn_factor,
                                    float2* __restrict__ out, const int* __restrict__ kmove, int kdelta) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= m * npairs) return;
  if (kmove) n_past += kdelta * *kmove;  // replayed device route: the position moves with the graph's token counter
  const int i2 = gid / npairs, pr = gid % npairs;
  float theta_base = float(n_past + i2);
  for (int t = 0; t < pr; t++) theta_base = __fmul_rn(theta_base, theta_scale);
  const float theta = __fmul_rn(freq_scale, theta_base);
  out[gid] = float2{__fmul_rn(cosf(theta), attn_factor), __fmul_rn(sinf(theta), attn_factor)};
}
hipError_t launch_rope_cos_sin(int m, int n_past, int n_dims, float freq_base, float freq_scale, float attn_factor,
                               float* out, hipStream_t st) {
  const int total = m * (n_dims / 2);
  if (total <= 0) return hipSuccess;
  hipLaunchKernelGGL(rope_cos_sin_kernel, grid1d(size_t(total), 64), dim3(64), 0, st, m, n_past, n_dims / 2,
                     powf(freq_base, -2.0f / n_dims), freq_scale, attn_factor, reinterpret_cast<float2*>(out), g_affine.k, int(g_affine.delta));
  return hipGetLastError();
}

// ---- prompt-sized form (round 4): the angles of a position are the same for every head and layer, and the kernel above spends its
// time re-deriving them per pair (up to npairs sequential multiplies + cosf + sinf per thread: 86 us for 2048 tokens x 64 heads).
// Here a table kernel writes (cos, sin) * attn_factor per (position, pair) with the SAME arithmetic, and the rotation kernel is a
// pure stream: 4 pairs per thread (two 16-byte loads, two 16-byte stores or one 16-byte fp16 store), 8 V elements per thread. ----
__global__ void rope_table_kernel(int m, int n_past, int npairs, int neox, float theta_scale, float freq_scale, float attn_factor,
                                  float2* __restrict__ out) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= m * npairs) return;
  const int i2 = gid / npairs, pr = gid % npairs;
  float theta_base = float(n_past + i2);
  if (neox) theta_base = __fmul_rn(theta_base, freq_scale);
  for (int t = 0; t < pr; t++) theta_base = __fmul_rn(theta_base, theta_scale);
  const float theta = __fmul_rn(freq_scale, theta_base);
  out[gid] = float2{__fmul_rn(cosf(theta), attn_factor), __fmul_rn(sinf(theta), attn_factor)};
}
typedef float qfloat4 __attribute__((ext_vector_type(4)));
typedef _Float16 qhalf4 __attribute__((ext_vector_type(4)));
typedef _Float16 qhalf8 __attribute__((ext_vector_type(8)));
__global__ void rope_qkv_append_tab_kernel(float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                           _Float16* __restrict__ kc, _Float16* __restrict__ vc, const float2* __restrict__ tab,
                                           int seq, int heads, int heads_kv, int head_size, int n_past, int n_dims, int neox,
                                           long long c_sl, long long c_head) {
  const int npairs = neox ? (head_size / n_dims) * (n_dims / 2) : head_size / 2;
  const int qpr = npairs / 4;  // groups of 4 pairs per row
  const size_t nq = size_t(seq) * heads * qpr, nk = size_t(seq) * heads_kv * qpr;
  const size_t nv = size_t(seq) * heads_kv * (head_size / 8);
  const size_t gid = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (gid < nq + nk) {
    const bool is_k = gid >= nq;
    const size_t g = is_k ? gid - nq : gid;
    const int hn = is_k ? heads_kv : heads;
    const size_t row = g / qpr;  // (i2, head)
    const int pr = int(g % qpr) * 4, i2 = int(row / hn), ih = int(row % hn);
    const float2* cs = tab + size_t(i2) * npairs + pr;
    const qfloat4 cs01 = *reinterpret_cast<const qfloat4*>(cs), cs23 = *reinterpret_cast<const qfloat4*>(cs + 2);
    const float c[4] = {cs01[0], cs01[2], cs23[0], cs23[2]}, sn[4] = {cs01[1], cs01[3], cs23[1], cs23[3]};
    const float* x = (is_k ? k : q) + row * head_size;
    float a[4], b[4];
    int ia, ib;
    if (neox) {  // pairs (ia + e, ia + e + n_dims / 2)
      const int blk = pr / (n_dims / 2), ic = pr % (n_dims / 2);
      ia = blk * n_dims + ic;
      ib = ia + n_dims / 2;
      const qfloat4 xa = *reinterpret_cast<const qfloat4*>(x + ia), xb = *reinterpret_cast<const qfloat4*>(x + ib);
#pragma unroll
      for (int e = 0; e < 4; e++) a[e] = xa[e], b[e] = xb[e];
    } else {  // adjacent pairs: 8 consecutive elements
      ia = 2 * pr;
      ib = ia + 4;
      const qfloat4 x0 = *reinterpret_cast<const qfloat4*>(x + ia), x1 = *reinterpret_cast<const qfloat4*>(x + ib);
      a[0] = x0[0], b[0] = x0[1], a[1] = x0[2], b[1] = x0[3], a[2] = x1[0], b[2] = x1[1], a[3] = x1[2], b[3] = x1[3];
    }
    float y0[4], y1[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
      y0[e] = __fsub_rn(__fmul_rn(a[e], c[e]), __fmul_rn(b[e], sn[e]));
      y1[e] = __fadd_rn(__fmul_rn(a[e], sn[e]), __fmul_rn(b[e], c[e]));
    }
    if (is_k) {
      _Float16* d = kc + (long long)(n_past + i2) * c_sl + (long long)ih * c_head;
      if (neox) {
        *reinterpret_cast<qhalf4*>(d + ia) = qhalf4{(_Float16)y0[0], (_Float16)y0[1], (_Float16)y0[2], (_Float16)y0[3]};
        *reinterpret_cast<qhalf4*>(d + ib) = qhalf4{(_Float16)y1[0], (_Float16)y1[1], (_Float16)y1[2], (_Float16)y1[3]};
      } else {
        *reinterpret_cast<qhalf8*>(d + ia) = qhalf8{(_Float16)y0[0], (_Float16)y1[0], (_Float16)y0[1], (_Float16)y1[1],
                                                     (_Float16)y0[2], (_Float16)y1[2], (_Float16)y0[3], (_Float16)y1[3]};
      }
    } else {
      float* d = q + row * head_size;
      if (neox) {
        *reinterpret_cast<qfloat4*>(d + ia) = qfloat4{y0[0], y0[1], y0[2], y0[3]};
        *reinterpret_cast<qfloat4*>(d + ib) = qfloat4{y1[0], y1[1], y1[2], y1[3]};
      } else {
        *reinterpret_cast<qfloat4*>(d + ia) = qfloat4{y0[0], y1[0], y0[1], y1[1]};
        *reinterpret_cast<qfloat4*>(d + ib) = qfloat4{y0[2], y1[2], y0[3], y1[3]};
      }
    }
  } else if (gid < nq + nk + nv) {
    const size_t g = gid - nq - nk;
    const int e = int(g % (head_size / 8)) * 8;
    const size_t row = g / (head_size / 8);
    const int i2 = int(row / heads_kv), ih = int(row % heads_kv);
    const float* x = v + row * head_size + e;
    const qfloat4 x0 = *reinterpret_cast<const qfloat4*>(x), x1 = *reinterpret_cast<const qfloat4*>(x + 4);
    *reinterpret_cast<qhalf8*>(vc + (long long)(n_past + i2) * c_sl + (long long)ih * c_head + e) =
        qhalf8{(_Float16)x0[0], (_Float16)x0[1], (_Float16)x0[2], (_Float16)x0[3], (_Float16)x1[0], (_Float16)x1[1], (_Float16)x1[2], (_Float16)x1[3]};
  }
}

hipError_t launch_rope_qkv_append(float* q, const float* k, const float* v, void* kc, void* vc, int seq, int heads,
                                  int heads_kv, int head_size, int n_past, int n_dims, int mode, float freq_base,
                                  float freq_scale, float attn_factor, long long c_sl, long long c_head, hipStream_t st) {
  const bool neox = (mode & 2) != 0;
  const size_t npairs = neox ? size_t(head_size / n_dims) * (n_dims / 2) : size_t(head_size / 2);
  const size_t total = size_t(seq) * (heads + heads_kv) * npairs + size_t(seq) * heads_kv * head_size;
  if (total == 0) return hipSuccess;
  const float theta_scale = powf(freq_base, -2.0f / n_dims);
  // prompt-sized calls: table + streaming kernel (16-byte accesses: every row, cache row and pair group aligned)
  static const bool no_tab = getenv("NS_ROPE_NO_TABLE") != nullptr;  // diagnostics (A/B)
  const auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  if (!no_tab && seq >= 16 && head_size % 8 == 0 && npairs % 4 == 0 && (!neox || (n_dims / 2) % 4 == 0) && c_sl % 8 == 0 && c_head % 8 == 0 &&
      al16(q) && al16(k) && al16(v) && al16(kc) && al16(vc)) {
    float2* tab = static_cast<float2*>(stream_scratch(st, size_t(seq) * npairs * sizeof(float2), 23));
    if (tab) {
      const size_t nt = size_t(seq) * npairs;
      hipLaunchKernelGGL(rope_table_kernel, grid1d(nt, 256), dim3(256), 0, st, seq, n_past, int(npairs), neox ? 1 : 0, theta_scale,
                         freq_scale, attn_factor, tab);
      const size_t items = size_t(seq) * (heads + heads_kv) * (npairs / 4) + size_t(seq) * heads_kv * (head_size / 8);
      hipLaunchKernelGGL(rope_qkv_append_tab_kernel, grid1d(items, 256), dim3(256), 0, st, q, k, v, static_cast<_Float16*>(kc),
                         static_cast<_Float16*>(vc), tab, seq, heads, heads_kv, head_size, n_past, n_dims, neox ? 1 : 0, c_sl, c_head);
      return hipGetLastError();
    }
  }
  hipLaunchKernelGGL(rope_qkv_append_kernel, grid1d(total, 256), dim3(256), 0, st, q, k, v, static_cast<_Float16*>(kc),
                     static_cast<_Float16*>(vc), seq, heads, heads_kv, head_size, n_past, n_dims, neox ? 1 : 0, theta_scale,
                     freq_scale, attn_factor, c_sl, c_head);
  return hipGetLastError();
}

thread_local Affine g_affine;
thread_local void* g_mha_out16 = nullptr;
thread_local bool g_mha_out16_written = false;
thread_local KvMirrorPair g_kvm;
// a cache cell at `addr` received `v`: its fp16 mirror cell, found from the cell's address (ns_route.h)
__device__ __forceinline__ void kvm_store(const KvMirrorArgs& m, const char* addr, float v) {
  if (!m.m16) return;
  const long long idx = (addr - m.base32) >> 2;
  if (idx < 0 || idx >= m.elems) return;
  if (fabsf(v) > 65504.f && m.overflow) __hip_atomic_store(m.overflow, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (!m.transposed) {
    m.m16[idx] = (_Float16)v;
  } else {  // V cache [head][dim][position] -> mirror [head][position][dim]
    const long long per = (long long)m.hs * m.n_ctx, bh = idx / per, r = idx - bh * per;
    const long long dim = r / m.n_ctx, pos = r - dim * m.n_ctx;
    m.m16[(bh * m.n_ctx + pos) * m.hs + dim] = (_Float16)v;
  }
}

hipError_t launch_rope(const float* src, float* dst, int batch, int seq, int heads, int head_size, int n_past, int n_dims,
                       int mode, float freq_base, float freq_scale, float attn_factor, hipStream_t st, float ext_factor,
                       float corr0, float corr1, const float* lr_factor, float lr_scale) {
  const size_t rows = size_t(batch) * seq * heads;
  if (rows == 0) return hipSuccess;
  const bool neox = (mode & 2) != 0;
  // dims the NeoX loop does not visit (head_size not a multiple of n_dims) keep their value
  if (dst != src && neox && head_size % n_dims != 0) {
    const hipError_t e = hipMemcpyAsync(dst, src, rows * head_size * 4, hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return e;
  }
  const float theta_scale = powf(freq_base, -2.0f / n_dims);
  const size_t npairs = neox ? size_t(head_size / n_dims) * (n_dims / 2) : size_t(head_size / 2);
  hipLaunchKernelGGL(rope_kernel, grid1d(rows * npairs, 256), dim3(256), 0, st, src, dst, rows, heads, seq, head_size, n_past,
                     n_dims, neox ? 1 : 0, theta_scale, freq_scale, attn_factor, RopeYarn{ext_factor, corr0, corr1, lr_factor, lr_scale},
                     g_affine.k, int(g_affine.delta));
  return hipGetLastError();
}

// GLM branch of ne_compute_forward_rope_f32 (mode & 4, ne_layers.c:9317-9347): ChatGLM's two-dimensional positions.
// One thread per (row, i0 < head_size / 4): the clamped token position rotates (x[i0], x[i0 + n_dims/2]), the block
// position rotates (x[i0 + n_dims], x[i0 + 3 n_dims / 2]); both angles are the reference's sequential fp32 products
// angle * theta_scale^i0.  Rows the "skip" form (mode & 1) does not visit are left alone.
struct GlmPads {
  int v[32];  // n_padding per batch entry (src1[ROPE_PARAMS_NUM + i3], :9319)
};
__global__ void rope_glm_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t rows, int heads, int seq,
                                int head_size, int n_past, int n_dims, int skip, float theta_scale, int prompt_size,
                                GlmPads pads) {
  const int quarter = head_size / 4;
  const size_t gid = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (gid >= rows * quarter) return;
  const size_t row = gid / quarter;
  const int i0 = int(gid % quarter);
  const int i2 = int((row / heads) % seq), i3 = int(row / (size_t(heads) * seq));
  if (skip && i2 < n_past) return;
  const long long p = skip ? i2 : (long long)n_past + i2;
  const long long npad = pads.v[i3];
  const long long tb = min(max(p - npad, 0ll), (long long)prompt_size - 2 - npad);
  float theta_base = float(tb);
  float block_theta = float(max(p - ((long long)prompt_size - 2), 0ll));
  for (int t = 0; t < i0; t++) {
    theta_base = __fmul_rn(theta_base, theta_scale);
    block_theta = __fmul_rn(block_theta, theta_scale);
  }
  const float c = cosf(theta_base), s = sinf(theta_base), cb = cosf(block_theta), sb = sinf(block_theta);
  const float* x = src + row * head_size + i0;
  float* y = dst + row * head_size + i0;
  const float x0 = x[0], x1 = x[n_dims / 2], x2 = x[n_dims], x3 = x[n_dims / 2 * 3];
  y[0] = __fsub_rn(__fmul_rn(x0, c), __fmul_rn(x1, s));
  y[n_dims / 2] = __fadd_rn(__fmul_rn(x0, s), __fmul_rn(x1, c));
  y[n_dims] = __fsub_rn(__fmul_rn(x2, cb), __fmul_rn(x3, sb));
  y[n_dims / 2 * 3] = __fadd_rn(__fmul_rn(x2, sb), __fmul_rn(x3, cb));
}
hipError_t launch_rope_glm(const float* src, float* dst, int batch, int seq, int heads, int head_size, int n_past, int n_dims,
                           bool skip, float freq_base, int prompt_size, const int* n_padding, hipStream_t st) {
  const size_t rows = size_t(batch) * seq * heads;
  if (rows == 0) return hipSuccess;
  if (batch > 32) return hipErrorInvalidValue;
  GlmPads pads{};
  for (int i = 0; i < batch; i++) pads.v[i] = n_padding[i];
  // elements the loop does not visit (and, in the skip form, whole rows) keep their value
  if (dst != src) {
    const hipError_t e = hipMemcpyAsync(dst, src, rows * head_size * 4, hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return e;
  }
  const float theta_scale = powf(freq_base, -2.0f / n_dims);
  hipLaunchKernelGGL(rope_glm_kernel, grid1d(rows * (head_size / 4), 256), dim3(256), 0, st, src, dst, rows, heads, seq,
                     head_size, n_past, n_dims, skip ? 1 : 0, theta_scale, prompt_size, pads);
  return hipGetLastError();
}

// ---- the remaining members of the reference's device-backend operator set (ne_bestla.h:99-111, ne_bestla_sycl.cpp) ----
// bestla_device_elewise_f32 (:297-326): NE_OP_SILU is the one operator it implements — ne_silu_f32(x) = x / (1 + expf(-x))
__global__ void silu_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n) {
  const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) y[i] = __fdiv_rn(x[i], __fadd_rn(1.0f, expf(-x[i])));
}
hipError_t launch_silu(const float* x, float* y, size_t n, hipStream_t st) {
  if (!n) return hipSuccess;
  hipLaunchKernelGGL(silu_kernel, grid1d(n, 256), dim3(256), 0, st, x, y, n);
  return hipGetLastError();
}
// bestla_device_dup_f32 (:537-591): 4-D strided copy of fp32 into fp32 or fp16 (the graph's kv-cache writes and permutes)
struct DupDims {
  long long ne[4], snb[4], dnb[4];  // extents of dst, byte strides of src and dst
};
__global__ void dup_kernel(const char* __restrict__ src, char* __restrict__ dst, DupDims d, int dst_f16, const int* __restrict__ kmove,
                           long long kdelta) {
  if (kmove) dst += kdelta * (long long)*kmove;  // replayed device route: the kv-cache cell moves with the graph's token counter
  const long long total = d.ne[0] * d.ne[1] * d.ne[2] * d.ne[3];
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const long long i0 = i % d.ne[0];
  i /= d.ne[0];
  const long long i1 = i % d.ne[1];
  i /= d.ne[1];
  const long long i2 = i % d.ne[2];
  const long long i3 = i / d.ne[2];
  const float v = *reinterpret_cast<const float*>(src + i0 * d.snb[0] + i1 * d.snb[1] + i2 * d.snb[2] + i3 * d.snb[3]);
  char* o = dst + i0 * d.dnb[0] + i1 * d.dnb[1] + i2 * d.dnb[2] + i3 * d.dnb[3];
  if (dst_f16)
    *reinterpret_cast<_Float16*>(o) = (_Float16)v;
  else
    *reinterpret_cast<float*>(o) = v;
}
// RoPE of q and k (adjacent rows, in place, one position) + the two kv-cache writes of a decode step in ONE launch (ns_route.cpp fuses the
// reference's rope(q), rope(k), cpy(k), cpy(v) nodes of a replayed token: llama.cpp:232-262).  Thread t < pairs rotates pair t of the q / k rows
// with rope_kernel's arithmetic (same angle products) and, for a k element, also stores it at its cache cell; the next nv threads copy v.
__global__ void rope_append_kernel(float* __restrict__ qk, int rows_front, int rows_k_first, int rows_k, int head_size, int n_past, int n_dims,
                                   int neox, float theta_scale, float freq_scale, float attn_factor, RopeYarn yarn, const float* __restrict__ ksrc,
                                   char* __restrict__ kdst, DupDims dk, const char* __restrict__ vsrc, char* __restrict__ vdst, DupDims dv,
                                   const int* __restrict__ kmove, int kd_pos, long long kd_k, long long kd_v, KvMirrorPair kvm) {
  if (kmove) {
    const int kk = *kmove;
    n_past += kd_pos * kk;
    kdst += kd_k * (long long)kk;
    vdst += kd_v * (long long)kk;
  }
  const int half = head_size / 2;
  const size_t npairs = neox ? size_t(head_size / n_dims) * (n_dims / 2) : size_t(half);
  const size_t rows = size_t(rows_front);
  const long long nk = dk.ne[0] * dk.ne[1] * dk.ne[2] * dk.ne[3], nv = dv.ne[0] * dv.ne[1] * dv.ne[2] * dv.ne[3];
  const size_t gid = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (gid < rows * npairs) {
    const size_t row = gid / npairs;
    const int pr = int(gid % npairs);
    float theta_base = float(n_past);
    if (neox) theta_base = __fmul_rn(theta_base, freq_scale);
    for (int t = 0; t < pr; t++) theta_base = __fmul_rn(theta_base, theta_scale);
    int ia, ib, i0;
    if (neox) {
      const int blk = pr / (n_dims / 2), ic = pr % (n_dims / 2);
      ia = blk * n_dims + ic;
      ib = ia + n_dims / 2;
      i0 = int(__fsub_rn(__fmul_rn(__fdiv_rn(-1.f, float(n_dims)), float(2 * ic)), float(blk)));
    } else {
      ia = 2 * pr;
      ib = ia + 1;
      i0 = ia;
    }
    const float theta = rope_theta(theta_base, freq_scale, i0, yarn);
    const float c = __fmul_rn(cosf(theta), attn_factor), sn = __fmul_rn(sinf(theta), attn_factor);
    float* x = qk + row * head_size;
    const float x0 = x[ia], x1 = x[ib];
    const float y0 = __fsub_rn(__fmul_rn(x0, c), __fmul_rn(x1, sn)), y1 = __fadd_rn(__fmul_rn(x0, sn), __fmul_rn(x1, c));
    x[ia] = y0;
    x[ib] = y1;
    // a k row: its two elements also go to the cache (k is [head_size][1][heads_kv] here: element index = head * head_size + dim)
    const long long krow = (long long)row - rows_k_first;
    if (krow >= 0 && krow < rows_k) {
      for (int e = 0; e < 2; e++) {
        long long i = krow * head_size + (e ? ib : ia);
        const float val = e ? y1 : y0;
        const long long i0d = i % dk.ne[0];
        i /= dk.ne[0];
        const long long i1d = i % dk.ne[1];
        i /= dk.ne[1];
        const long long i2d = i % dk.ne[2];
        const long long i3d = i / dk.ne[2];
        // (the cpy node indexes the source with the destination's coordinates: the element order of the packed source IS that walk)
        char* cell = kdst + i0d * dk.dnb[0] + i1d * dk.dnb[1] + i2d * dk.dnb[2] + i3d * dk.dnb[3];
        *reinterpret_cast<float*>(cell) = val;
        kvm_store(kvm.k, cell, val);
      }
    }
    return;
  }
  long long i = (long long)(gid - rows * npairs);
  if (i >= nv) return;
  const long long i0 = i % dv.ne[0];
  i /= dv.ne[0];
  const long long i1 = i % dv.ne[1];
  i /= dv.ne[1];
  const long long i2 = i % dv.ne[2];
  const long long i3 = i / dv.ne[2];
  const float vval = *reinterpret_cast<const float*>(vsrc + i0 * dv.snb[0] + i1 * dv.snb[1] + i2 * dv.snb[2] + i3 * dv.snb[3]);
  char* vcell = vdst + i0 * dv.dnb[0] + i1 * dv.dnb[1] + i2 * dv.dnb[2] + i3 * dv.dnb[3];
  *reinterpret_cast<float*>(vcell) = vval;
  kvm_store(kvm.v, vcell, vval);
  (void)ksrc;
  (void)nk;
}
hipError_t launch_rope_append(float* qk, int rows_front, int rows_k_first, int rows_k, int head_size, int n_past, int n_dims, int mode, float freq_base,
                              float freq_scale, float attn_factor, float ext_factor, float corr0, float corr1, const void* ksrc, void* kdst,
                              const long long* kne, const long long* ksnb, const long long* kdnb, const void* vsrc, void* vdst, const long long* vne,
                              const long long* vsnb, const long long* vdnb, long long kd_k, long long kd_v, hipStream_t st) {
  DupDims dk, dv;
  for (int i = 0; i < 4; i++) dk.ne[i] = kne[i], dk.snb[i] = ksnb[i], dk.dnb[i] = kdnb[i], dv.ne[i] = vne[i], dv.snb[i] = vsnb[i], dv.dnb[i] = vdnb[i];
  const bool neox = (mode & 2) != 0;
  const float theta_scale = powf(freq_base, -2.0f / n_dims);
  const size_t npairs = neox ? size_t(head_size / n_dims) * (n_dims / 2) : size_t(head_size / 2);
  const size_t total = size_t(rows_front) * npairs + size_t(vne[0] * vne[1] * vne[2] * vne[3]);
  if (!total) return hipSuccess;
  hipLaunchKernelGGL(rope_append_kernel, grid1d(total, 256), dim3(256), 0, st, qk, rows_front, rows_k_first, rows_k, head_size, n_past, n_dims,
                     neox ? 1 : 0, theta_scale, freq_scale, attn_factor, RopeYarn{ext_factor, corr0, corr1, nullptr, 1.f}, static_cast<const float*>(ksrc),
                     static_cast<char*>(kdst), dk, static_cast<const char*>(vsrc), static_cast<char*>(vdst), dv, g_affine.k, int(g_affine.delta), kd_k, kd_v, g_kvm);
  return hipGetLastError();
}
__global__ void dup2_kernel(const char* __restrict__ src0, char* __restrict__ dst0, DupDims d0, int f16_0, const char* __restrict__ src1,
                            char* __restrict__ dst1, DupDims d1, int f16_1, const int* __restrict__ kmove, long long kdelta0, long long kdelta1,
                            KvMirrorPair kvm) {
  const long long total0 = d0.ne[0] * d0.ne[1] * d0.ne[2] * d0.ne[3];
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool second = i >= total0;
  if (second) i -= total0;
  const DupDims& d = second ? d1 : d0;
  if (i >= d.ne[0] * d.ne[1] * d.ne[2] * d.ne[3]) return;
  const char* src = second ? src1 : src0;
  char* dst = second ? dst1 : dst0;
  if (kmove) dst += (second ? kdelta1 : kdelta0) * (long long)*kmove;
  const long long i0 = i % d.ne[0];
  i /= d.ne[0];
  const long long i1 = i % d.ne[1];
  i /= d.ne[1];
  const long long i2 = i % d.ne[2];
  const long long i3 = i / d.ne[2];
  const float v = *reinterpret_cast<const float*>(src + i0 * d.snb[0] + i1 * d.snb[1] + i2 * d.snb[2] + i3 * d.snb[3]);
  char* o = dst + i0 * d.dnb[0] + i1 * d.dnb[1] + i2 * d.dnb[2] + i3 * d.dnb[3];
  if (second ? f16_1 : f16_0) {
    *reinterpret_cast<_Float16*>(o) = (_Float16)v;
  } else {
    *reinterpret_cast<float*>(o) = v;
    kvm_store(second ? kvm.v : kvm.k, o, v);  // (first copy: the K cells, second: the V cells — ns_route.cpp XK_DUP2)
  }
}
// A copy whose destination runs along one axis and whose source runs along ANOTHER (the V cache write of a prompt: the cache is transposed,
// [head][dim][position], the source rows are [position][head][dim]): dup_kernel reads 4 bytes out of every 16 KB row per thread — 68 us for the two cache
// writes of a 1500-token prompt and layer.  32 x 32 tiles through LDS: reads run along the source's contiguous axis `ax`, writes along axis 0.
__global__ __launch_bounds__(256) void dup_transpose_kernel(const char* __restrict__ src, char* __restrict__ dst, DupDims d, int ax, int dst_f16) {
  __shared__ float tile[32][33];
  const int o1 = ax == 1 ? 2 : 1, o2 = ax == 3 ? 2 : 3;  // the two axes that are neither 0 nor ax
  const long long z = blockIdx.z, z1 = z % d.ne[o1], z2 = z / d.ne[o1];
  const long long a0 = (long long)blockIdx.x * 32, b0 = (long long)blockIdx.y * 32;  // tile origin along axis 0 / axis ax
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const char* sb = src + z1 * d.snb[o1] + z2 * d.snb[o2];
  char* db = dst + z1 * d.dnb[o1] + z2 * d.dnb[o2];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const long long i0 = a0 + ty + 8 * r, ia = b0 + tx;
    if (i0 < d.ne[0] && ia < d.ne[ax]) tile[ty + 8 * r][tx] = *reinterpret_cast<const float*>(sb + i0 * d.snb[0] + ia * d.snb[ax]);
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const long long ia = b0 + ty + 8 * r, i0 = a0 + tx;
    if (i0 < d.ne[0] && ia < d.ne[ax]) {
      char* o = db + i0 * d.dnb[0] + ia * d.dnb[ax];
      const float v = tile[tx][ty + 8 * r];
      if (dst_f16) *reinterpret_cast<_Float16*>(o) = (_Float16)v;
      else *reinterpret_cast<float*>(o) = v;
    }
  }
}
// fp32 -> fp32 with both sides contiguous along axis 0 (the K cache write of a prompt: rows of head_size floats): four elements per thread, 16-byte accesses
__global__ void dup_vec4_kernel(const char* __restrict__ src, char* __restrict__ dst, DupDims d) {
  const long long n0 = d.ne[0] >> 2, total = n0 * d.ne[1] * d.ne[2] * d.ne[3];
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const long long i0 = (i % n0) << 2;
  i /= n0;
  const long long i1 = i % d.ne[1];
  i /= d.ne[1];
  const long long i2 = i % d.ne[2];
  const long long i3 = i / d.ne[2];
  typedef float f4 __attribute__((ext_vector_type(4)));
  *reinterpret_cast<f4*>(dst + i0 * 4 + i1 * d.dnb[1] + i2 * d.dnb[2] + i3 * d.dnb[3]) =
      *reinterpret_cast<const f4*>(src + i0 * 4 + i1 * d.snb[1] + i2 * d.snb[2] + i3 * d.snb[3]);
}
static bool dup_vec4_ok(const void* src, const void* dst, const long long* ne, const long long* snb, const long long* dnb, bool f16) {
  if (f16 || g_affine.k || g_kvm.k.m16 || g_kvm.v.m16) return false;
  if (ne[0] * ne[1] * ne[2] * ne[3] < (1 << 16) || (ne[0] & 3) || snb[0] != 4 || dnb[0] != 4) return false;
  if ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) return false;
  for (int i = 1; i < 4; i++)
    if (ne[i] > 1 && ((snb[i] | dnb[i]) & 15)) return false;
  return true;
}
// the axis the transposing form walks on the source side, or -1: a plain launch serves the copy as well (or it is small)
static int dup_transpose_axis(const long long* ne, const long long* snb, const long long* dnb, bool f16) {
  if (g_affine.k || g_kvm.k.m16 || g_kvm.v.m16) return -1;
  const long long total = ne[0] * ne[1] * ne[2] * ne[3];
  if (total < (1 << 16) || ne[0] < 32 || dnb[0] != (f16 ? 2 : 4) || snb[0] == 4) return -1;
  for (int ax = 1; ax < 4; ax++)
    if (ne[ax] >= 32 && snb[ax] == 4) {
      const int o1 = ax == 1 ? 2 : 1, o2 = ax == 3 ? 2 : 3;
      if (ne[o1] * ne[o2] > 65535) return -1;
      return ax;
    }
  return -1;
}
static hipError_t launch_dup_transpose(const void* src, void* dst, const DupDims& d, int ax, bool f16, hipStream_t st) {
  const int o1 = ax == 1 ? 2 : 1, o2 = ax == 3 ? 2 : 3;
  const dim3 grid(unsigned((d.ne[0] + 31) / 32), unsigned((d.ne[ax] + 31) / 32), unsigned(d.ne[o1] * d.ne[o2]));
  hipLaunchKernelGGL(dup_transpose_kernel, grid, dim3(256), 0, st, static_cast<const char*>(src), static_cast<char*>(dst), d, ax, f16 ? 1 : 0);
  return hipGetLastError();
}
hipError_t launch_dup2(const void* src0, void* dst0, const long long* ne0, const long long* snb0, const long long* dnb0, bool f16_0,
                       const void* src1, void* dst1, const long long* ne1, const long long* snb1, const long long* dnb1, bool f16_1, hipStream_t st) {
  {  // prompt-sized cache writes: a copy that transposes takes the tiled form, the other one a launch of its own
    const int ax0 = dup_transpose_axis(ne0, snb0, dnb0, f16_0), ax1 = dup_transpose_axis(ne1, snb1, dnb1, f16_1);
    if (ax0 >= 0 || ax1 >= 0) {
      hipError_t e = launch_dup(src0, dst0, ne0, snb0, dnb0, f16_0, st);
      return e != hipSuccess ? e : launch_dup(src1, dst1, ne1, snb1, dnb1, f16_1, st);
    }
  }
  DupDims d0, d1;
  for (int i = 0; i < 4; i++) d0.ne[i] = ne0[i], d0.snb[i] = snb0[i], d0.dnb[i] = dnb0[i], d1.ne[i] = ne1[i], d1.snb[i] = snb1[i], d1.dnb[i] = dnb1[i];
  const long long total = ne0[0] * ne0[1] * ne0[2] * ne0[3] + ne1[0] * ne1[1] * ne1[2] * ne1[3];
  if (total <= 0) return hipSuccess;
  hipLaunchKernelGGL(dup2_kernel, grid1d(size_t(total), 256), dim3(256), 0, st, static_cast<const char*>(src0), static_cast<char*>(dst0), d0,
                     f16_0 ? 1 : 0, static_cast<const char*>(src1), static_cast<char*>(dst1), d1, f16_1 ? 1 : 0, g_affine.k, g_affine.delta,
                     g_affine.delta2, g_kvm);
  return hipGetLastError();
}
hipError_t launch_dup(const void* src, void* dst, const long long* ne, const long long* snb, const long long* dnb, bool dst_f16,
                      hipStream_t st) {
  DupDims d;
  for (int i = 0; i < 4; i++) d.ne[i] = ne[i], d.snb[i] = snb[i], d.dnb[i] = dnb[i];
  const long long total = ne[0] * ne[1] * ne[2] * ne[3];
  if (total <= 0) return hipSuccess;
  if (const int ax = dup_transpose_axis(ne, snb, dnb, dst_f16); ax >= 0) return launch_dup_transpose(src, dst, d, ax, dst_f16, st);
  if (dup_vec4_ok(src, dst, ne, snb, dnb, dst_f16)) {
    hipLaunchKernelGGL(dup_vec4_kernel, grid1d(size_t(total / 4), 256), dim3(256), 0, st, static_cast<const char*>(src), static_cast<char*>(dst), d);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(dup_kernel, grid1d(size_t(total), 256), dim3(256), 0, st, static_cast<const char*>(src),
                     static_cast<char*>(dst), d, dst_f16 ? 1 : 0, g_affine.k, g_affine.delta);
  return hipGetLastError();
}

__global__ void gather_cols_kernel(const float* __restrict__ a, int lda, const int* __restrict__ idx,
                                   float* __restrict__ out, int m, int k) {
  const size_t gid = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (gid >= size_t(m) * k) return;
  const int r = int(gid / k), j = int(gid % k);
  out[gid] = a[size_t(r) * lda + idx[j]];
}
hipError_t launch_gather_cols(const float* a, int lda, const int* idx, float* out, int m, int k, hipStream_t st) {
  const size_t total = size_t(m) * k;
  hipLaunchKernelGGL(gather_cols_kernel, dim3(unsigned((total + 255) / 256)), dim3(256), 0, st, a, lda, idx, out, m, k);
  return hipGetLastError();
}

hipError_t launch_bcast_binary(int batch, int vsize, const float* t, const float* v, int vstep, float* out, bool mul,
                               hipStream_t st) {
  const size_t total = size_t(batch) * vsize;
  hipLaunchKernelGGL(bcast_kernel, grid1d(total, 256), dim3(256), 0, st, t, v, out, total, vsize, vstep, mul);
  return hipGetLastError();
}

void touch_quant_module() {
  hipFuncAttributes fa;
  (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(silu_kernel));
  (void)hipGetLastError();
}
}  // namespace ns
