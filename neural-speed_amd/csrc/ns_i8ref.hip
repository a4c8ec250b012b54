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
// Kernel: one 512-thread workgroup per 16-column tile of the streaming layout (ns_common.h: ns_weight), wave w takes
// k-steps w, w+8, ...; lane (nn, c) owns column nn and the eight k of slot c in every 32-deep slice, exactly the bytes
// the MFMA kernels read; integer dots with v_dot4_u32_u8 on the stored codes, activation codes staged in LDS.  Up to
// four rows of A per pass (weights are re-read for more rows: a numerics mode for decode-sized calls, not a prefill
// kernel).
//
// Who runs what (launch_i8ref below):   1 - 4 rows: the streaming decode kernel's int8 variant (ns_gemv.hip, XV = 3; ns_api.cpp)
//   5 - 15 rows, k-steps the matrix-core kernels do not take: i8ref_kernel (this file)
//   16 rows and up: i8mfma2_kernel (ns_i8g2.hip: one exact fp16 MFMA per slice on zero-point-folded operands; its fp16 A'
//   comes from the quantizer's GEMM-sized form, ns_quant.hip aquant_u8_vec_kernel, or from i8prep_kernel here), or the
//   first matrix-core kernel i8mfma_kernel (this file: integer MFMA + corrections) when "i8_mfma" = 1.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdlib>
#include <map>
#include <mutex>

#include "ns_common.h"
#include "ns_dev.h"
#include "ns_i8g2.h"

namespace ns {
namespace {

// which matrix-core kernel GEMM-sized calls take: 2 = i8mfma2_kernel (default), 1 = i8mfma_kernel
std::atomic<int> g_i8_mfma_gen{[] {
  const char* e = getenv("NS_I8_MFMA");
  return e && atoi(e) == 1 ? 1 : 2;
}()};

std::atomic<int> g_i8_tile{0};  // workgroup tile of i8mfma2_kernel: 0 = by size, 1 = 64 x 64, 4 = 64 x 256 (tests / A-B runs)

constexpr int kI8Rows = 4;
constexpr int kI8Waves = 8, kI8Threads = kI8Waves * 64;  // wave w takes k-steps w, w + 8, ...; four records in flight each


__device__ __forceinline__ float load_scale(const uint8_t* p, uint32_t dt) {
  if (dt == DT_F32) return *reinterpret_cast<const float*>(p);
  const uint32_t h = *reinterpret_cast<const uint16_t*>(p);
  if (dt == DT_BF16) return __builtin_bit_cast(float, h << 16);
  return f16_bits_to_f32(h);
}

// u8 x u8 dot products (v_dot4_u32_u8) on the STORED codes: with u = q + bias (bias 8 for nibbles, 128 for bytes),
//   sum (a - za)(q - zb) = sum a*u - (zb + bias) * sum a - za * sum u + n * za * (zb + bias),   n = 8 per lane and slice,
// every term an exact integer.  Activation codes of the row group are staged in LDS once per workgroup and chunk, the
// eight codes of a lane's slot already in the byte order the nibble unpack produces ((x & 0x0f0f0f0f) = codes 0,4,1,5
// and ((x >> 4) & 0x0f0f0f0f) = codes 2,6,3,7 of the dword), so a slice costs six dot instructions per row and no
// per-element work.  Columns of A beyond K are staged as the row's last zero point: their term vanishes identically.
template <bool FOUR>  // nibble container (4 slices of 32 per 128-deep k-step) or byte container (2 per 64-deep)
__global__ __launch_bounds__(kI8Threads) void i8ref_kernel(const I8RefParams p) {
  constexpr int NJ = FOUR ? 4 : 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char i8_smem[];
  __shared__ float red[kI8Waves][kI8Rows][16];
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, nn = l & 15, cslot = l >> 4;
  const int tile = blockIdx.x;
  constexpr bool four = FOUR;
  const uint32_t bias = four ? 8u : 128u;
  const int chunk_k = p.chunk_steps * p.kstep_len;  // bytes per staged row
  const int sbytes = p.scale_dt == DT_F32 ? 4 : 2;
  const int rec_sbytes = p.sps * sbytes;            // scale bytes of one lane's k-step record: 2, 4, 8 or 16
  // LDS: [kI8Rows][chunk_k] activation codes | [kI8Rows][nblk] activation scales | [kI8Rows][nblk] zero points (int)
  float* as_lds = reinterpret_cast<float*>(i8_smem + size_t(kI8Rows) * chunk_k);
  int* az_lds = reinterpret_cast<int*>(as_lds + size_t(kI8Rows) * p.nblk);
  for (int r0 = 0; r0 < p.m; r0 += kI8Rows) {
    const int rows = min(kI8Rows, p.m - r0);
    float acc[kI8Rows] = {0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    for (int idx = tid; idx < rows * p.nblk; idx += kI8Threads) {  // the row group's activation scales / zero points
      const int r = idx / p.nblk, kb = idx - r * p.nblk;
      as_lds[r * p.nblk + kb] = p.ascale[size_t(r0 + r) * p.nblk + kb];
      az_lds[r * p.nblk + kb] = int(p.azp[size_t(r0 + r) * p.nblk + kb]);
    }
    for (int c0 = 0; c0 < p.ksteps; c0 += p.chunk_steps) {
      const int cend = min(c0 + p.chunk_steps, p.ksteps);
      __syncthreads();  // previous chunk / row group fully consumed
      // ---- stage: 8-byte groups, permuted for the nibble container ----
      const int groups = chunk_k >> 3;
      const bool vec_ok = (p.k & 7) == 0 && (reinterpret_cast<uintptr_t>(p.aq) & 7) == 0;
      for (int idx = tid; idx < rows * groups; idx += kI8Threads) {
        const int r = idx / groups, gq = idx - r * groups;
        const int k0 = c0 * p.kstep_len + gq * 8;
        const size_t row = size_t(r0 + r);
        const uint8_t* src = p.aq + row * p.k + k0;
        uint32_t lo, hi;  // codes 0..3 and 4..7 of the group
        if (vec_ok && k0 + 8 <= p.k) {
          const uint2 v = *reinterpret_cast<const uint2*>(src);
          lo = v.x, hi = v.y;
        } else {
          const uint32_t zpad = p.azp[row * p.nblk + (p.nblk - 1)];
          uint32_t b[8];
#pragma unroll
          for (int i = 0; i < 8; i++) b[i] = (k0 + i < p.k) ? uint32_t(src[i]) : zpad;
          lo = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
          hi = b[4] | (b[5] << 8) | (b[6] << 16) | (b[7] << 24);
        }
        uint32_t d0 = lo, d1 = hi;
        if (four) {  // (a0,a4,a1,a5) and (a2,a6,a3,a7): the byte order of the nibble unpack
          d0 = __builtin_amdgcn_perm(hi, lo, 0x05010400u);
          d1 = __builtin_amdgcn_perm(hi, lo, 0x07030602u);
        }
        *reinterpret_cast<uint2*>(i8_smem + size_t(r) * chunk_k + gq * 8) = uint2{d0, d1};
      }
      __syncthreads();
      // the scale / zero-point words of a lane's k-step record, fetched together with the codes
      struct Corr {
        uint32_t s[4];
        uint32_t z;
      };
      auto fetch_corr = [&](int s) {
        Corr c{{0, 0, 0, 0}, 0};
        const uint32_t srow = uint32_t(s * p.srow_mul) >> p.srow_shift;
        const size_t crow = size_t(tile) * p.srows + srow;
        const uint8_t* sp = p.scales + crow * p.sstride + size_t(nn) * rec_sbytes;
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
          const int8_t* zp = p.zps + crow * p.zstride + nn * p.sps;
          for (int e = 0; e < p.sps; e++) c.z |= uint32_t(uint8_t(zp[e])) << (8 * e);
        }
        return c;
      };
      auto consume = [&](const uint4v& rec, const Corr& cr, int s) {
        const uint32_t xw[4] = {rec.x, rec.y, rec.z, rec.w};
#pragma unroll
        for (int j = 0; j < NJ; j++) {
          const int k0 = s * p.kstep_len + 32 * j + 8 * cslot;
          if (k0 >= p.k) continue;
          const int e = (j * p.sps) / NJ;
          // register-resident selects (a dynamically indexed local array would live in scratch memory)
          auto pick = [&](int i) { return i == 0 ? cr.s[0] : (i == 1 ? cr.s[1] : (i == 2 ? cr.s[2] : cr.s[3])); };
          float sb;
          if (p.scale_dt == DT_F32) {
            sb = __builtin_bit_cast(float, pick(e));
          } else {
            const uint32_t h = (pick(e >> 1) >> (16 * (e & 1))) & 0xffffu;
            sb = p.scale_dt == DT_BF16 ? __builtin_bit_cast(float, h << 16) : f16_bits_to_f32(h);
          }
          const int zb = (p.asym ? int(int8_t((cr.z >> (8 * e)) & 0xffu)) : 0) + int(bias);
          uint32_t u0, u1;
          if (four) {
            u0 = xw[j] & 0x0f0f0f0fu, u1 = (xw[j] >> 4) & 0x0f0f0f0fu;
          } else {
            u0 = xw[2 * j] ^ 0x80808080u, u1 = xw[2 * j + 1] ^ 0x80808080u;  // raw int8 -> q + 128
          }
          const int su = int(__builtin_amdgcn_udot4(u0, 0x01010101u, __builtin_amdgcn_udot4(u1, 0x01010101u, 0u, false), false));
          const int kb = min(k0 / p.blocksize, p.nblk - 1);  // a 32-deep slice never straddles a k-block
          const int loff = (s - c0) * p.kstep_len + 32 * j + 8 * cslot;
#pragma unroll
          for (int r = 0; r < kI8Rows; r++) {
            if (r >= rows) break;
            const uint2 av = *reinterpret_cast<const uint2*>(i8_smem + size_t(r) * chunk_k + loff);
            const int za = az_lds[r * p.nblk + kb];
            const int dot = int(__builtin_amdgcn_udot4(av.x, u0, __builtin_amdgcn_udot4(av.y, u1, 0u, false), false));
            const int sa = int(__builtin_amdgcn_udot4(av.x, 0x01010101u, __builtin_amdgcn_udot4(av.y, 0x01010101u, 0u, false), false));
            const int isum = dot - zb * sa - za * su + 8 * za * zb;
            acc[r] += float(isum) * (as_lds[r * p.nblk + kb] * sb);
          }
        }
      };
      // four records (codes + scale words) in flight per wave
      auto load_rec = [&](int s) {
        return *reinterpret_cast<const uint4v*>(p.codes + (size_t(tile) * p.ksteps + s) * p.qstride + l * 16);
      };
      for (int s = c0 + w; s < cend; s += 4 * kI8Waves) {
        uint4v rec[4];
        Corr cr[4];
#pragma unroll
        for (int t = 0; t < 4; t++) {
          const int st = s + t * kI8Waves;
          const bool on = st < cend;
          rec[t] = on ? load_rec(st) : uint4v{0, 0, 0, 0};
          cr[t] = on ? fetch_corr(st) : Corr{{0, 0, 0, 0}, 0};
        }
#pragma unroll
        for (int t = 0; t < 4; t++)
          if (s + t * kI8Waves < cend) consume(rec[t], cr[t], s + t * kI8Waves);
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
        float v = 0.f;
#pragma unroll
        for (int ww = 0; ww < kI8Waves; ww++) v += red[ww][r][tid & 15];
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
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The same semantics on the matrix cores, for GEMM-sized calls (prefill): v_mfma_i32_16x16x32_i8 per 32-deep slice.
//   sum_k (a - za)(q - zb) with  a' = a - 128 (the u8 code as an s8: a ^ 0x80),  u = stored code (nibble 0..15, i.e.
//   q + 8; bytes: the raw s8 q), zbb = zb + 8 (nibbles) or zb (bytes):
//       = [sum a' u]  +  (128 - za) * [sum u]  +  zbb * (32 za - [sum a])          every term an exact integer
//   [sum a' u]  : one MFMA per (16 rows, 16 columns, slice); the 8 bytes a lane holds are exactly its record's codes
//   [sum u]     : one more MFMA per slice with an all-ones A — every accumulator of a lane then holds its column's sum
//   [sum a], za, scale_a : per (row, slice), computed once while the activation codes are staged into LDS
// The fp32 side is the reference's: float(integer sum) * (scale_a * scale_b) accumulated per slice (ref_kblock_int8,
// bestla/ut/bestla_gemm.cpp:159-190, accumulates per k-block; a k-block of 64 / 128 is two / four slices here — fp32
// summation order is the only difference, as in the kernel above).
// Workgroup = 4 waves = 64 rows x 64 columns: wave w owns column tile 4 bx + w and all four 16-row tiles; A is staged per
// 512-deep chunk: [64 rows][512 + 16 pad] s8 (row stride shifted by four banks: ds_read_b64 of 16 rows x 2 k-groups is
// conflict-free) + three [16 slices][64 rows] fp32 planes: 128 - za, 32 za - sum a, scale_a.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kGM = 64, kGChunk = 512, kGRowStride = kGChunk + 16, kGSlices = kGChunk / 32;
typedef int int4v __attribute__((ext_vector_type(4)));

// SDT: scale dtype of the weight (0 bf16, 1 f16, 2 f32), SPS: scales per k-step record and column, ASYM: zero points — compile
// time, so that the per-slice scale selection costs no scalar branches (the kernel is issue-bound, profiles/r02z_*)
template <bool FOUR, int SDT, int SPS, bool ASYM>
__global__ __launch_bounds__(256) void i8mfma_kernel(const I8RefParams p) {
  constexpr int NJ = FOUR ? 4 : 2;            // slices per k-step record
  constexpr int CS = kGChunk / (32 * NJ);     // k-step records per chunk: 4 (128-deep) or 8 (64-deep)
  extern __shared__ __attribute__((aligned(16))) unsigned char g_smem[];
  unsigned char* a_lds = g_smem;                                                         // [kGM][kGRowStride]
  int4v* meta = reinterpret_cast<int4v*>(g_smem + size_t(kGM) * kGRowStride);            // 3 planes [kGSlices][kGM] fp32
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, nn = l & 15, g = l >> 4;
  const int tile = blockIdx.x * 4 + w;
  const bool tile_on = tile < (p.n + 15) / 16;
  const int r0 = blockIdx.y * kGM;
  const int zbias = FOUR ? 8 : 0;
  constexpr int sbytes = SDT == 2 ? 4 : 2;
  constexpr int rec_sbytes = SPS * sbytes;
  const bool vec_ok = (p.k & 7) == 0 && (reinterpret_cast<uintptr_t>(p.aq) & 7) == 0;
  float acc[4][4];
#pragma unroll
  for (int rt = 0; rt < 4; rt++)
#pragma unroll
    for (int i = 0; i < 4; i++) acc[rt][i] = 0.f;
  const long ones = 0x0101010101010101L;

  for (int c0 = 0; c0 < p.ksteps; c0 += CS) {
    // ---- this wave's weight records of the chunk: requested before the staging so that their latency hides behind it
    uint4v rec[CS];
    uint32_t sw[CS][4], zw[CS];
#pragma unroll
    for (int t = 0; t < CS; t++) {
      const int s = c0 + t;
      const bool on = tile_on && s < p.ksteps;
      rec[t] = uint4v{0, 0, 0, 0};
      sw[t][0] = sw[t][1] = sw[t][2] = sw[t][3] = 0, zw[t] = 0;
      if (on) {
        rec[t] = *reinterpret_cast<const uint4v*>(p.codes + (size_t(tile) * p.ksteps + s) * p.qstride + l * 16);
        const uint32_t srow = uint32_t(s * p.srow_mul) >> p.srow_shift;
        const size_t crow = size_t(tile) * p.srows + srow;
        const uint8_t* sp = p.scales + crow * p.sstride + size_t(nn) * rec_sbytes;
        if constexpr (rec_sbytes == 16) {
          const uint4v v = *reinterpret_cast<const uint4v*>(sp);
          sw[t][0] = v.x, sw[t][1] = v.y, sw[t][2] = v.z, sw[t][3] = v.w;
        } else if constexpr (rec_sbytes == 8) {
          const uint2 v = *reinterpret_cast<const uint2*>(sp);
          sw[t][0] = v.x, sw[t][1] = v.y;
        } else if constexpr (rec_sbytes == 4) {
          sw[t][0] = *reinterpret_cast<const uint32_t*>(sp);
        } else {
          sw[t][0] = *reinterpret_cast<const uint16_t*>(sp);
        }
        if constexpr (ASYM) {
          const int8_t* zp = p.zps + crow * p.zstride + nn * SPS;
#pragma unroll
          for (int e = 0; e < SPS; e++) zw[t] |= uint32_t(uint8_t(zp[e])) << (8 * e);
        }
      }
    }
    __syncthreads();  // the previous chunk is consumed
    // ---- stage: one (row, slice) per thread and pass: 32 codes -> s8, permuted for the nibble container; the slice's
    //      integer corrections and activation scale next to them
    for (int idx = tid; idx < kGM * kGSlices; idx += 256) {
      const int r = idx & (kGM - 1), q = idx / kGM;
      const int k0 = c0 * (32 * NJ) + 32 * q;
      const int row = r0 + r;
      int4v mt = {0, 0, 0, 0};
      uint2 out[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
      if (row < p.m && k0 < p.k) {
        const int kb = min(k0 / p.blocksize, p.nblk - 1);
        const int za = int(p.azp[size_t(row) * p.nblk + kb]);
        const uint32_t zpad = p.azp[size_t(row) * p.nblk + (p.nblk - 1)];
        const uint8_t* src = p.aq + size_t(row) * p.k + k0;
        uint32_t sum = 0;
#pragma unroll
        for (int gq = 0; gq < 4; gq++) {
          uint32_t lo, hi;
          if (vec_ok && k0 + 8 * gq + 8 <= p.k) {
            const uint2 v = *reinterpret_cast<const uint2*>(src + 8 * gq);
            lo = v.x, hi = v.y;
          } else {
            uint32_t b[8];
#pragma unroll
            for (int i = 0; i < 8; i++) b[i] = (k0 + 8 * gq + i < p.k) ? uint32_t(src[8 * gq + i]) : zpad;
            lo = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
            hi = b[4] | (b[5] << 8) | (b[6] << 16) | (b[7] << 24);
          }
          sum = __builtin_amdgcn_udot4(lo, 0x01010101u, __builtin_amdgcn_udot4(hi, 0x01010101u, sum, false), false);
          uint32_t d0 = lo, d1 = hi;
          if (FOUR) {  // (a0,a4,a1,a5) and (a2,a6,a3,a7): the byte order of the nibble unpack
            d0 = __builtin_amdgcn_perm(hi, lo, 0x05010400u);
            d1 = __builtin_amdgcn_perm(hi, lo, 0x07030602u);
          }
          out[gq] = uint2{d0 ^ 0x80808080u, d1 ^ 0x80808080u};
        }
        // a tail slice is padded with the row's last zero point: (a - za) vanishes there because kb is the last block
        // (small integers: exact in fp32, and so is every sum of them the compute loop forms — all below 2^24)
        mt.x = __builtin_bit_cast(int, float(128 - za));
        mt.y = __builtin_bit_cast(int, float(32 * za - int(sum)));
        mt.z = __builtin_bit_cast(int, p.ascale[size_t(row) * p.nblk + kb]);
      }
      unsigned char* dst = a_lds + size_t(r) * kGRowStride + 32 * q;
#pragma unroll
      for (int gq = 0; gq < 4; gq++) *reinterpret_cast<uint2*>(dst + 8 * gq) = out[gq];
      // three planes [c1 | e | scale_a][slice][row]: a lane's four consecutive rows of one plane are one 16-byte read
      float* mf = reinterpret_cast<float*>(meta);
      mf[(0 * kGSlices + q) * kGM + r] = __builtin_bit_cast(float, int(mt.x));
      mf[(1 * kGSlices + q) * kGM + r] = __builtin_bit_cast(float, int(mt.y));
      mf[(2 * kGSlices + q) * kGM + r] = __builtin_bit_cast(float, int(mt.z));
    }
    __syncthreads();
    if (!tile_on) continue;
    // ---- compute
#pragma unroll
    for (int t = 0; t < CS; t++) {
      const int s = c0 + t;
      if (s >= p.ksteps) break;
      const uint32_t xw[4] = {rec[t].x, rec[t].y, rec[t].z, rec[t].w};
#pragma unroll
      for (int j = 0; j < NJ; j++) {
        const int k0 = s * (32 * NJ) + 32 * j;
        if (k0 >= p.k) continue;
        const int q = t * NJ + j;
        const int e = (j * SPS) / NJ;  // a constant once the loops are unrolled
        float sb;
        if constexpr (SDT == 2) {
          sb = __builtin_bit_cast(float, sw[t][e]);
        } else {
          const uint32_t h = (sw[t][e >> 1] >> (16 * (e & 1))) & 0xffffu;
          sb = SDT == 0 ? __builtin_bit_cast(float, h << 16) : f16_bits_to_f32(h);
        }
        const int zbb = (ASYM ? int(int8_t((zw[t] >> (8 * e)) & 0xffu)) : 0) + zbias;
        uint32_t u0, u1;
        if (FOUR) {
          u0 = xw[j] & 0x0f0f0f0fu, u1 = (xw[j] >> 4) & 0x0f0f0f0fu;
        } else {
          u0 = xw[2 * j], u1 = xw[2 * j + 1];  // the raw s8 codes
        }
        const long b = long(uint64_t(u0) | (uint64_t(u1) << 32));
        const int4v zero = {0, 0, 0, 0};
        const int4v sv = __builtin_amdgcn_mfma_i32_16x16x32_i8(ones, b, zero, 0, 0, 0);  // column sum of the stored codes
        int4v d[4];
#pragma unroll
        for (int rt = 0; rt < 4; rt++) {  // the four row tiles back to back: their latency overlaps
          const long a = *reinterpret_cast<const long*>(a_lds + size_t(rt * 16 + nn) * kGRowStride + 32 * q + 8 * g);
          d[rt] = __builtin_amdgcn_mfma_i32_16x16x32_i8(a, b, zero, 0, 0, 0);
        }
        const float suf = float(int(sv.x)), zbf = float(zbb);
#pragma unroll
        for (int rt = 0; rt < 4; rt++) {
          const float* mf = reinterpret_cast<const float*>(meta) + q * kGM + rt * 16 + 4 * g;
          const float4 c1 = *reinterpret_cast<const float4*>(mf);                         // 128 - za      of rows 4g .. 4g+3
          const float4 ee = *reinterpret_cast<const float4*>(mf + kGSlices * kGM);        // 32 za - sum a
          const float4 sa = *reinterpret_cast<const float4*>(mf + 2 * kGSlices * kGM);    // scale_a
          const int dd[4] = {d[rt].x, d[rt].y, d[rt].z, d[rt].w};
          const float c1v[4] = {c1.x, c1.y, c1.z, c1.w}, eev[4] = {ee.x, ee.y, ee.z, ee.w}, sav[4] = {sa.x, sa.y, sa.z, sa.w};
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const float isum = __builtin_fmaf(suf, c1v[i], __builtin_fmaf(zbf, eev[i], float(dd[i])));  // exact integer
            acc[rt][i] = __builtin_fmaf(isum, sav[i] * sb, acc[rt][i]);
          }
        }
      }
    }
  }
  if (!tile_on) return;
  const int col = tile * 16 + nn;
  if (col >= p.n) return;
#pragma unroll
  for (int rt = 0; rt < 4; rt++)
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int row = r0 + rt * 16 + 4 * g + i;
      if (row >= p.m) continue;
      float v = acc[rt][i];
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
      if (p.c16) p.c16[size_t(row) * p.ldc + col] = (_Float16)v;
    }
}

template <bool F, int S, int D>
void launch_i8mfma_a(bool asym, dim3 grid, size_t lds, hipStream_t st, const I8RefParams& p) {
  if (asym)
    hipLaunchKernelGGL((i8mfma_kernel<F, D, S, true>), grid, dim3(256), lds, st, p);
  else
    hipLaunchKernelGGL((i8mfma_kernel<F, D, S, false>), grid, dim3(256), lds, st, p);
}
template <bool F, int S>
void launch_i8mfma(int sdt, bool asym, dim3 grid, size_t lds, hipStream_t st, const I8RefParams& p) {
  if (sdt == 0)
    launch_i8mfma_a<F, S, 0>(asym, grid, lds, st, p);
  else if (sdt == 1)
    launch_i8mfma_a<F, S, 1>(asym, grid, lds, st, p);
  else
    launch_i8mfma_a<F, S, 2>(asym, grid, lds, st, p);
}

// (The second matrix-core kernel, i8mfma2_kernel, lives in ns_i8g2.hip; its operand preparation follows.)
typedef float float2v __attribute__((ext_vector_type(2)));

struct I8PrepParams {
  const uint8_t* aq;   // [m][k]
  const uint8_t* azp;  // [m][nblk]
  uint8_t* out;        // [m][kp] fp16, kp = 32 nsl
  int m, k, nsl, blocksize, nblk;
  int scale16;         // nibble containers: codes 2, 3, 6, 7 of every eight stored / 16
};

// A'[row][k] = fp16(a - za(row, k-block)) (nibble containers: / 16 for k mod 8 in {2, 3, 6, 7}), zero beyond K: eight codes per thread
__global__ __launch_bounds__(256) void i8prep_kernel(const I8PrepParams p) {
  const size_t idx = size_t(blockIdx.x) * 256 + threadIdx.x;
  const size_t per_row = size_t(p.nsl) * 4;
  if (idx >= size_t(p.m) * per_row) return;
  const int row = int(idx / per_row);
  const int piece = int(idx - size_t(row) * per_row);
  const int k0 = piece * 8;  // = 32 * slice + 8 * g
  uint32_t lo = 0, hi = 0, za = 0;
  if (k0 < p.k) {
    za = p.azp[size_t(row) * p.nblk + min(k0 / p.blocksize, p.nblk - 1)];
    const uint8_t* src = p.aq + size_t(row) * p.k + k0;
    if (k0 + 8 <= p.k && (reinterpret_cast<uintptr_t>(src) & 7) == 0) {
      const uint2 v = *reinterpret_cast<const uint2*>(src);
      lo = v.x, hi = v.y;
    } else {  // beyond K: the zero point itself, a - za = 0
      uint32_t b[8];
#pragma unroll
      for (int i = 0; i < 8; i++) b[i] = (k0 + i < p.k) ? uint32_t(src[i]) : za;
      lo = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
      hi = b[4] | (b[5] << 8) | (b[6] << 16) | (b[7] << 24);
    }
  }
  // 0x6400 | byte = 1024 + a as fp16 (ulp 1 there); minus 1024 + za: exact
  const half2_t zc = __builtin_bit_cast(half2_t, (0x6400u + za) * 0x00010001u);
  const uint32_t src2[4] = {__builtin_amdgcn_perm(0x64646464u, lo, 0x04010400u), __builtin_amdgcn_perm(0x64646464u, lo, 0x04030402u),
                            __builtin_amdgcn_perm(0x64646464u, hi, 0x04010400u), __builtin_amdgcn_perm(0x64646464u, hi, 0x04030402u)};
  // nibble containers: codes 2, 3, 6, 7 of the eight are stored as (a - za) / 16 (exact: eight significant bits): the GEMM builds
  // their B' as 16 (u - zbb) straight from the nibbles' position in the dword, one shift less per pair
  const half2_t sixteenth = {(_Float16)0.0625f, (_Float16)0.0625f};
  uint32_t o[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    half2_t v = __builtin_bit_cast(half2_t, src2[i]) - zc;
    if ((i & 1) && p.scale16) v = v * sixteenth;
    o[i] = __builtin_bit_cast(uint32_t, v);
  }
  *reinterpret_cast<uint4v*>(p.out + idx * 16) = uint4v{o[0], o[1], o[2], o[3]};
}


// which form of A' the stream's scratch (slot 7) holds for its current activation codes (slot 4): 0 none, 1 the nibble
// containers' (odd pairs / 16), 2 the byte containers'
int prep_state(hipStream_t st, bool set, int value = 0) {
  static std::mutex mu;
  static std::map<hipStream_t, int> state;
  std::lock_guard<std::mutex> lock(mu);
  if (set) state[st] = value;
  auto it = state.find(st);
  return it == state.end() ? 0 : it->second;
}

}  // namespace

// quantize_fp_u8_colblock(A) into the stream's scratch (slot 6) in the layout gemv_kernel's int8-reference variant stages:
// codes [m][ldq] (ldq = K rounded up to 16) | [m][nblk] fp32 scales | [m][nblk] u8 zero points.  reuse: the scratch already
// holds it for this (A, m, K, blocksize) — the previous call of a fused QKV / gate-up group
struct LastAq {
  const float* a;
  int m, k, bs;
  hipStream_t st;
  bool valid;
};
static LastAq g_last_aq = {nullptr, 0, 0, 0, nullptr, false};  // what the decode scratch (slot 6) holds codes of
hipError_t i8_quantize_for_decode(const float* a, int lda, const ns_weight* w, int m, hipStream_t st, bool reuse, I8Act* out) {
  const int bs = w->blocksize >= w->k ? w->k : w->blocksize;
  const int nblk = (w->k + bs - 1) / bs;
  const int ldq = (w->k + 15) & ~15;
  const size_t aq_bytes = size_t(m) * ldq;
  uint8_t* base = static_cast<uint8_t*>(stream_scratch(st, aq_bytes + size_t(m) * nblk * 5 + 16, 6));
  if (!base) return hipErrorOutOfMemory;
  float* as = reinterpret_cast<float*>(base + aq_bytes);
  uint8_t* az = base + aq_bytes + size_t(m) * nblk * 4;
  out->aq = base, out->ldq = ldq;
  out->corr = base + aq_bytes;
  out->nblk = nblk, out->blocksize = bs;
  // the quantizer launch itself is deferred: the decode kernel quantizes inside its own launch where it can (gemv_kernel XV = 5)
  out->a32 = a, out->lda32 = lda, out->rows = m, out->cols = w->k;
  // `reuse`: the caller's previous call quantized the same rows — true only if that call really ran the quantizer launch (it may
  // have quantized inside its kernel instead, leaving the scratch untouched)
  out->quantized = reuse && g_last_aq.valid && g_last_aq.a == a && g_last_aq.m == m && g_last_aq.k == w->k && g_last_aq.bs == bs && g_last_aq.st == st;
  if (!out->quantized) g_last_aq.valid = false;
  return hipSuccess;
}
hipError_t i8_quantize_finish(I8Act* q, hipStream_t st) {
  if (q->quantized) return hipSuccess;
  uint8_t* base = const_cast<uint8_t*>(q->aq);
  float* as = reinterpret_cast<float*>(const_cast<uint8_t*>(q->corr));
  uint8_t* az = const_cast<uint8_t*>(q->corr) + size_t(q->rows) * q->nblk * 4;
  const hipError_t e = launch_aquant_u8(q->rows, q->cols, q->a32, q->lda32, base, q->ldq, as, q->nblk, az, q->blocksize, nullptr, st);
  if (e == hipSuccess) {
    q->quantized = true;
    g_last_aq = LastAq{q->a32, q->rows, q->cols, q->blocksize, st, true};
  }
  return e;
}

void set_i8_mfma_gen(int gen) { g_i8_mfma_gen.store(gen == 1 ? 1 : 2); }
int i8_tile_forced() { return g_i8_tile.load(std::memory_order_relaxed); }
void set_i8_tile(int tile) { g_i8_tile.store(tile >= 1 && tile <= 4 ? tile : 0); }

bool i8ref_supported(const ns_weight* w) {
  return (w->kind == WK_INT4 || w->kind == WK_INT8) && w->blocksize > 0 && (w->blocksize % 32 == 0 || w->blocksize >= w->k);
}

// C[m][n] = epi(dequant-free int8-compute product of quantize_u8(A) and the integer weight `w`)
// reuse_aq: the stream's scratch already holds quantize_u8(A) for this (A, m, k, blocksize) from the previous call — the
// fused QKV / gate-up entries quantize A once for all their weights, as the reference does (ip_fusion_qkv.cpp:84-86)
hipError_t launch_i8ref(const float* a, int lda, const ns_weight* w, float* c, void* c16, int m, int ldc, int epilogue,
                        const float* d, int ldd, hipStream_t st, bool reuse_aq) {
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
  // GEMM-sized calls take the matrix-core kernels (NS_I8_MFMA_MIN_M rows and up, default 16; 0 disables them); nibble
  // containers the second one: one exact fp16 MFMA per slice on operands with the zero points folded in ("i8_mfma" 1 /
  // NS_I8_MFMA=1: the first kernel everywhere)
  static const int mfma_min_m = [] {
    const char* e = getenv("NS_I8_MFMA_MIN_M");
    return e ? atoi(e) : 16;
  }();
  const bool four = w->kind != WK_INT8;
  const bool mfma = mfma_min_m > 0 && m >= mfma_min_m && w->kstep_len == (four ? 128 : 64);
  const bool v2 = mfma && g_i8_mfma_gen.load(std::memory_order_relaxed) != 1 &&
                  (four ? (w->sps == 4 || w->sps == 2 || w->sps == 1) : (w->sps == 2 || w->sps == 1));
  const int nsl = w->ksteps * (four ? 4 : 2);  // 32-deep slices per row of A'
  const int want = four ? 1 : 2;               // the form of A' this weight multiplies
  uint8_t* pa = nullptr;
  if (v2) {
    pa = static_cast<uint8_t*>(stream_scratch(st, size_t(m) * nsl * 64, 7));
    if (!pa) return hipErrorOutOfMemory;
  }
  if (!reuse_aq) {
    hipError_t e = hipErrorNotSupported;
    bool with_ap = false;
    if (mfma) {  // the vector form of the quantizer; it writes A' as well when the rows need no padding (K a multiple of the k-step)
      with_ap = v2 && w->k == nsl * 32;
      e = launch_aquant_u8_vec(m, w->k, a, lda, aq, w->k, as, nblk, az, bs, with_ap ? pa : nullptr, nsl * 32, four, st);
    }
    if (e == hipErrorNotSupported) {
      with_ap = false;
      e = launch_aquant_u8(m, w->k, a, lda, aq, w->k, as, nblk, az, bs, nullptr, st);
    }
    if (e != hipSuccess) return e;
    prep_state(st, true, with_ap ? want : 0);  // does slot 7 match slot 4?
  }
  I8RefParams p{};
  p.codes = reinterpret_cast<const uint8_t*>(w->codes);
  p.scales = static_cast<const uint8_t*>(w->scales);
  p.zps = w->zps;
  p.qstride = w->qstride;
  p.sstride = w->sstride;
  p.zstride = w->zstride;
  p.ksteps = w->ksteps;
  p.kstep_len = w->kstep_len;
  p.nj = four ? 4 : 2;
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
  if (mfma) {
    const size_t lds = size_t(kGM) * kGRowStride + size_t(3) * kGSlices * kGM * 4;
    const dim3 grid(unsigned((w->ntiles + 3) / 4), unsigned((m + kGM - 1) / kGM));
    const int sdt = w->scale_dt == DT_BF16 ? 0 : (w->scale_dt == DT_F32 ? 2 : 1);
    if (v2) {
      I8Gemm2Params g2{};
      g2.b = p;
      g2.nsl = nsl;
      g2.pa = pa;
      if (prep_state(st, false) != want) {  // (a reused quantization consumed by other kernels so far, or rows that need padding)
        const I8PrepParams pr{aq, az, pa, m, w->k, nsl, bs, nblk, four ? 1 : 0};
        hipLaunchKernelGGL(i8prep_kernel, dim3(unsigned((size_t(m) * nsl * 4 + 255) / 256)), dim3(256), 0, st, pr);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        prep_state(st, true, want);
      }
      if (four) {
        if (w->sps == 4) return launch_i8g2_n4(sdt, w->asym, m, w->ntiles, st, g2);
        if (w->sps == 2) return launch_i8g2_n2(sdt, w->asym, m, w->ntiles, st, g2);
        return launch_i8g2_n1(sdt, w->asym, m, w->ntiles, st, g2);
      }
      if (w->sps == 2) return launch_i8g2_b2(sdt, w->asym, m, w->ntiles, st, g2);
      return launch_i8g2_b1(sdt, w->asym, m, w->ntiles, st, g2);
    }
    if (four && w->sps == 4) launch_i8mfma<true, 4>(sdt, w->asym, grid, lds, st, p);
    else if (four && w->sps == 2) launch_i8mfma<true, 2>(sdt, w->asym, grid, lds, st, p);
    else if (four && w->sps == 1) launch_i8mfma<true, 1>(sdt, w->asym, grid, lds, st, p);
    else if (!four && w->sps == 2) launch_i8mfma<false, 2>(sdt, w->asym, grid, lds, st, p);
    else if (!four && w->sps == 1) launch_i8mfma<false, 1>(sdt, w->asym, grid, lds, st, p);
    else return hipErrorNotSupported;
    return hipGetLastError();
  }
  // LDS (<= 60 KiB): the row group's activation scales / zero points, then up to four rows of u8 codes per chunk
  const size_t corr_bytes = size_t(kI8Rows) * nblk * 8;
  if (corr_bytes > 40 * 1024) return hipErrorNotSupported;
  const size_t code_budget = 60 * 1024 - corr_bytes;
  int chunk_steps = w->ksteps;
  while ((size_t((chunk_steps + kI8Waves - 1) / kI8Waves * kI8Waves) * w->kstep_len * kI8Rows > code_budget) && chunk_steps > kI8Waves)
    chunk_steps = (chunk_steps + 1) / 2;
  chunk_steps = (chunk_steps + kI8Waves - 1) / kI8Waves * kI8Waves;  // whole rounds of the waves
  p.chunk_steps = chunk_steps;
  const size_t lds = size_t(chunk_steps) * w->kstep_len * kI8Rows + corr_bytes;
  if (p.nj == 4)
    hipLaunchKernelGGL(i8ref_kernel<true>, dim3(w->ntiles), dim3(kI8Threads), lds, st, p);
  else
    hipLaunchKernelGGL(i8ref_kernel<false>, dim3(w->ntiles), dim3(kI8Threads), lds, st, p);
  return hipGetLastError();
}

}  // namespace ns
