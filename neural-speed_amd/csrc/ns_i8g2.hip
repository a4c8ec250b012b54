// ns_i8g2.hip — i8mfma2_kernel, the second matrix-core kernel of the int8-reference mode (ns_i8ref.hip holds the mode's host
// logic, the first kernel and the operand preparation).  Compiled once per (container kind, scales per k-step record):
//   -DNS_I8G2_NAME=launch_i8g2_n4 -DNS_I8G2_FOUR=true -DNS_I8G2_SPS=4   etc. (Makefile)
//
// ---------------------------------------------------------------------------------------------------------------
// Second matrix-core kernel (nibble containers — the Q4_0 case — and byte containers): the whole integer sum of a 32-deep slice out of ONE
// MFMA, already an fp32 number, nothing to correct or convert per accumulator.  i8mfma_kernel above spends five VALU
// instructions and three 16-byte LDS reads per accumulator and slice on the zero-point terms and the int -> float
// conversion; its VALU pipe is the bound (profiles/r03_pmc_i8mfma_sq_counters.txt).  Here both zero points are folded into
// the operands and the operands are small integers held in fp16:
//       A' = a - za  in [-255, 255],      B' = u - zbb  in [-15, 15]     (u the stored nibble = q + 8, zbb = zb + 8)
//                                         B' = q - zb   in [-255, 255]   (byte containers)
// every product (<= 3825; bytes 65025) and every partial sum of a slice (<= 122400; bytes 2.1 M) is an integer below 2^24, so
// v_mfma_f32_16x16x32_f16 returns float(sum_k (a - za)(q - zb)) EXACTLY — the integer dot of the reference, computed on the
// fp16 matrix pipe.  A' depends on the activation alone: i8prep_kernel writes it once per call as [m][K'] fp16 and every
// workgroup stages it with plain 16-byte copies; B' costs the lane that holds the record eleven VALU instructions per
// slice (0x6400 | nibble = 1024 + u as fp16, minus 1024 + zbb: exact).  What is left per accumulator and slice is the
// reference's fp32 side, float(isum) * (scale_a * scale_b) added in slice order — the same expression on the same numbers
// as i8mfma_kernel, bit for bit (tests/test_gpu_int8_mode.py compares the two) — as packed fp32 instructions, two
// accumulators each.
// Workgroup = 4 waves = 64 rows x (64 or 256) columns; A' per 256-deep chunk in four planes [g][64 rows][8 slices x 16 B]
// (row stride 144 B: the sixteen lanes of every ds_read_b128 group hold sixteen different rows, conflict-free) + the
// activation scales [8 slices][64 + 4 rows]; the next chunk's A' pieces and scales and the next k-step's weight records are
// in flight in registers while the current ones are consumed.
// ---------------------------------------------------------------------------------------------------------------
#include <hip/hip_runtime.h>

#include "ns_common.h"
#include "ns_dev.h"
#include "ns_i8g2.h"

#ifndef NS_I8G2_NAME  // a bare `hipcc -c ns_i8g2.hip` builds the Q4_0 variant
#define NS_I8G2_NAME launch_i8g2_n4
#define NS_I8G2_FOUR true
#define NS_I8G2_SPS 4
#endif

namespace ns {
namespace {

constexpr int kGM = 64;  // rows of a 64-row half
typedef float float2v __attribute__((ext_vector_type(2)));
// Tile geometry of i8mfma2_kernel: RH 64-row halves per workgroup (and per wave), kSL slices per staged chunk
template <int RH>
struct I8G2Geom {
  static constexpr int kRows = 64 * RH;
  static constexpr int kSL = 8 / RH;             // 256-deep chunks of 64 rows, 128-deep chunks of 128 rows
  static constexpr int kRow = kSL * 16 + 16;     // 144 / 80 B = 9 / 5 sixteen-byte slots: rows 0 .. 15 land in sixteen different slots
  static constexpr int kPlane = kRows * kRow;    // 9216 / 10240: multiples of 256 B, so the plane does not move the bank
  static constexpr int kSaStride = kRows + 4;    // floats per slice row of the scale plane
  static constexpr size_t kBuf = size_t(4) * kPlane + size_t(kSL) * kSaStride * 4;   // one staged chunk: 39040 / 43072 B
  // 64-row workgroups double-buffer the chunk (two workgroups per CU: 2 x 2 x 39040 B <= 160 KiB): one barrier per chunk and
  // the staging writes of the next chunk sit between the two k-steps of the current one
  static constexpr int kBufs = RH == 1 ? 2 : 1;
  static constexpr size_t kLds = kBuf * kBufs;
};

// Tile: a workgroup of WV waves covers 64 RH rows x 16 CT WV columns; each wave all the rows and CT 16-column tiles.
//   (1, 1, 16)  64 x 256 as sixteen waves: the large-problem shape.  112 VGPRs: four waves per SIMD, which is what this
//               kernel needs — its MFMA, LDS and VALU work per slice are each a third of the time and do not overlap inside
//               one wave.  Every byte of A' staged feeds sixteen waves; L2 / fabric traffic per flop is 40 % of the
//               64 x 64 workgroup's (the first version of this kernel sat at 6 TB/s of cache traffic).  Chunks
//               double-buffered in LDS: one barrier per 256-deep chunk, staging writes between its two k-steps
//   (2, 2, 4)   128 x 128, four waves of 128 x 32: every B' fragment (VALU work) feeds eight MFMAs, but 248 VGPRs (two waves
//               per SIMD): 10 - 12 % behind at 2048 x 4096 x 4096.  (1, 4, 4): 64 x 256 as four waves: 20 % behind
//               (profiles/r03_i8_prefill_kernels.json).  Measured, then left out of the build (compile time)
//   (1, 1, 4)   64 x 64, double-buffered: small problems (four times the workgroups)
template <bool FOUR, int SDT, int SPS, bool ASYM, int RH, int CT, int WV>
__global__ __launch_bounds__(64 * WV, WV == 4 ? 2 : 4) void i8mfma2_kernel(const I8Gemm2Params pp) {
  using G = I8G2Geom<RH>;
  const I8RefParams& p = pp.b;
  constexpr int NJ = FOUR ? 4 : 2;   // 32-deep slices per k-step record (128-deep nibble / 64-deep byte containers)
  constexpr int KS = 32 * NJ;
  constexpr int CS = G::kSL / NJ;    // k-steps per chunk: 2 / 1 (nibbles), 4 / 2 (bytes)
  constexpr int kThreads = 64 * WV;
  constexpr int kPieces = G::kRows * G::kSL * 4 / kThreads;  // 16-byte A' pieces per thread and chunk: 8 (2 with sixteen waves)
  constexpr int kRowPieces = G::kSL * 4;                // pieces per row and chunk: 32 / 16
  constexpr int kInstrRows = 64 / kRowPieces;           // rows one wave-instruction of the staging covers: 2 / 4
  constexpr int kSaAll = G::kRows * G::kSL;             // activation scales per chunk: 512
  constexpr int kSaPer = (kSaAll + kThreads - 1) / kThreads;  // per thread: 2 (1 for the first half of sixteen waves)
  extern __shared__ __attribute__((aligned(16))) unsigned char g2_smem[];
  constexpr size_t kSaOff = size_t(4) * G::kPlane;  // the scale plane behind the four A' planes of a buffer
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, nn = l & 15, g = l >> 4;
  const int tile0 = (blockIdx.x * WV + w) * CT;
  const int ntiles = (p.n + 15) / 16;
  const bool active = tile0 < ntiles;
  const int r0 = blockIdx.y * G::kRows;
  constexpr int sbytes = SDT == 2 ? 4 : 2;
  constexpr int rec_sbytes = SPS * sbytes;
  float2v acc[RH][CT][4][2];
#pragma unroll
  for (int h = 0; h < RH; h++)
#pragma unroll
    for (int ct = 0; ct < CT; ct++)
#pragma unroll
      for (int rt = 0; rt < 4; rt++) acc[h][ct][rt][0] = acc[h][ct][rt][1] = float2v{0.f, 0.f};
  uint32_t magic = 0x64006400u;  // held in a register: (x & mask) | magic is then one v_and_or_b32
  asm volatile("" : "+v"(magic));

  struct KRec {  // the records of one k-step: codes, scale words, zero points of the wave's CT tiles
    uint4v rec[CT];
    uint32_t sw[CT][4], zw[CT];
  };
  auto fetch_w = [&](int s, KRec& r) {
#pragma unroll
    for (int ct = 0; ct < CT; ct++) {
      const int tile = tile0 + ct;
      r.rec[ct] = uint4v{0, 0, 0, 0};
      r.sw[ct][0] = r.sw[ct][1] = r.sw[ct][2] = r.sw[ct][3] = 0, r.zw[ct] = 0;
      if (tile < ntiles && s < p.ksteps) {
        r.rec[ct] = *reinterpret_cast<const uint4v*>(p.codes + (size_t(tile) * p.ksteps + s) * p.qstride + l * 16);
        const uint32_t srow = uint32_t(s * p.srow_mul) >> p.srow_shift;
        const size_t crow = size_t(tile) * p.srows + srow;
        const uint8_t* sp = p.scales + crow * p.sstride + size_t(nn) * rec_sbytes;
        if constexpr (rec_sbytes == 16) {
          const uint4v v = *reinterpret_cast<const uint4v*>(sp);
          r.sw[ct][0] = v.x, r.sw[ct][1] = v.y, r.sw[ct][2] = v.z, r.sw[ct][3] = v.w;
        } else if constexpr (rec_sbytes == 8) {
          const uint2 v = *reinterpret_cast<const uint2*>(sp);
          r.sw[ct][0] = v.x, r.sw[ct][1] = v.y;
        } else if constexpr (rec_sbytes == 4) {
          r.sw[ct][0] = *reinterpret_cast<const uint32_t*>(sp);
        } else {
          r.sw[ct][0] = *reinterpret_cast<const uint16_t*>(sp);
        }
        if constexpr (ASYM) {
          const int8_t* zp = p.zps + crow * p.zstride + nn * SPS;
#pragma unroll
          for (int e = 0; e < SPS; e++) r.zw[ct] |= uint32_t(uint8_t(zp[e])) << (8 * e);
        }
      }
    }
  };
  uint4v pa[kPieces];
  float psa[kSaPer];
  // staging lane map: wave-instruction i of wave w covers kInstrRows whole rows of the chunk (1 KiB of A', contiguous per row);
  // lane -> (row l / kRowPieces, slice l % kSL, k-group (l / kSL) % 4): eight consecutive lanes write eight consecutive 16-byte slots
  const int st_row = (l / kRowPieces), st_sl = l % G::kSL, st_g = (l / G::kSL) & 3;
  auto fetch_a = [&](int c0) {  // this thread's share of the chunk that starts at k-step c0
#pragma unroll
    for (int i = 0; i < kPieces; i++) {
      const int row = r0 + (w + WV * i) * kInstrRows + st_row, sl = c0 * NJ + st_sl;
      pa[i] = uint4v{0, 0, 0, 0};
      if (row < p.m && sl < pp.nsl) pa[i] = *reinterpret_cast<const uint4v*>(pp.pa + ((size_t(row) * pp.nsl + sl) * 4 + st_g) * 16);
    }
#pragma unroll
    for (int i = 0; i < kSaPer; i++) {  // activation scales: idx -> (slice idx % kSL, row idx / kSL)
      const int idx = tid + kThreads * i;
      const int row = r0 + idx / G::kSL, k0 = (c0 * NJ + idx % G::kSL) * 32;
      psa[i] = 0.f;
      if (idx < kSaAll && row < p.m && k0 < p.k) psa[i] = p.ascale[size_t(row) * p.nblk + min(k0 / p.blocksize, p.nblk - 1)];
    }
  };
  auto write_chunk = [&](unsigned char* buf) {  // registers -> LDS
    float* sa_lds = reinterpret_cast<float*>(buf + kSaOff);
#pragma unroll
    for (int i = 0; i < kPieces; i++)
      *reinterpret_cast<uint4v*>(buf + size_t(st_g) * G::kPlane + size_t((w + WV * i) * kInstrRows + st_row) * G::kRow + st_sl * 16) = pa[i];
#pragma unroll
    for (int i = 0; i < kSaPer; i++) {
      const int idx = tid + kThreads * i;
      if (idx < kSaAll) sa_lds[(idx % G::kSL) * G::kSaStride + idx / G::kSL] = psa[i];
    }
  };
  auto stage = [&](int c0) {  // single buffer: registers -> LDS for chunk c0 between two barriers, then the next chunk's loads
    __syncthreads();          // the previous chunk is consumed
    write_chunk(g2_smem);
    __syncthreads();
    if (c0 + CS < p.ksteps) fetch_a(c0 + CS);
  };
  // one k-step (four slices) of the staged chunk against the records in wc; t = its position in the chunk
  auto kstep = [&](const KRec& wc, int s, int t, const unsigned char* a_lds) {
    const float* sa_lds = reinterpret_cast<const float*>(a_lds + kSaOff);
    // Slices that share a scale pair (group sizes of 64 and more: NJ / SPS slices per scale of the k-step record) are CHAINED through
    // the MFMA accumulator — the integer sum of the whole k-block, still exact (<= 4 x 122400 / 4 x 2.1 M < 2^24), scaled ONCE: the
    // reference's own order (one int32 sum per k-block, bestla_wrapper.h:768-831) and a quarter of the fp32 work at g128.
    constexpr int JG = SPS >= NJ ? 1 : NJ / SPS;  // slices per scale group inside a k-step record
    floatx4 dch[RH][CT][4];
    floatx4 sa_g[RH][4];
    float sb_g[CT];
#pragma unroll
    for (int j = 0; j < NJ; j++) {
      const int k0 = s * KS + 32 * j;
      const bool first = j % JG == 0, last = j % JG == JG - 1;
      if (k0 >= p.k) {
        if constexpr (JG > 1) {  // (wave-uniform) a group cut short by K: what it holds so far is scaled now
          if (!first && (s * KS + 32 * (j - 1)) < p.k) {
#pragma unroll
            for (int h = 0; h < RH; h++)
#pragma unroll
              for (int ct = 0; ct < CT; ct++) {
                const float2v sb2 = {sb_g[ct], sb_g[ct]};
#pragma unroll
                for (int rt = 0; rt < 4; rt++) {
                  acc[h][ct][rt][0] = __builtin_elementwise_fma(float2v{dch[h][ct][rt].x, dch[h][ct][rt].y}, float2v{sa_g[h][rt].x, sa_g[h][rt].y} * sb2, acc[h][ct][rt][0]);
                  acc[h][ct][rt][1] = __builtin_elementwise_fma(float2v{dch[h][ct][rt].z, dch[h][ct][rt].w}, float2v{sa_g[h][rt].z, sa_g[h][rt].w} * sb2, acc[h][ct][rt][1]);
                }
              }
          }
        }
        continue;
      }
      const int q = t * NJ + j;
      const int e = (j * SPS) / NJ;  // a constant once the loops are unrolled
      half8_t b[CT];
      float sb[CT];
      auto make_b = [&](int ct) {
        if constexpr (SDT == 2) {
          sb[ct] = __builtin_bit_cast(float, wc.sw[ct][e]);
        } else {
          const uint32_t h = (wc.sw[ct][e >> 1] >> (16 * (e & 1))) & 0xffffu;
          sb[ct] = SDT == 0 ? __builtin_bit_cast(float, h << 16) : f16_bits_to_f32(h);
        }
        if constexpr (FOUR) {
          // B' as fp16.  The dword's nibbles 4s and 4s + 4 (s = 0 .. 3) are codes 2s and 2s + 1 of the lane's eight.
          //   s = 0, 2:  (x >> 4s & 0x000f000f) | 0x6400 twice = 1024 + u;        minus 1024 + zbb     = u - zbb
          //   s = 1, 3:  (x >> 4(s-1) & 0x00f000f0) | 0x6400 twice = 1024 + 16 u; minus 1024 + 16 zbb  = 16 (u - zbb), against A' / 16
          uint32_t zbb = 8;
          if constexpr (ASYM) zbb = uint32_t(8 + int(int8_t((wc.zw[ct] >> (8 * e)) & 0xffu)));
          const half2_t z1 = __builtin_bit_cast(half2_t, (0x6400u + zbb) * 0x00010001u);
          const half2_t z16 = __builtin_bit_cast(half2_t, (0x6400u + (zbb << 4)) * 0x00010001u);
          const uint32_t x = j == 0 ? wc.rec[ct].x : (j == 1 ? wc.rec[ct].y : (j == 2 ? wc.rec[ct].z : wc.rec[ct].w));
          const uint32_t y = x >> 8;
          const uint32_t bw0 = __builtin_bit_cast(uint32_t, __builtin_bit_cast(half2_t, (x & 0x000f000fu) | magic) - z1);
          const uint32_t bw1 = __builtin_bit_cast(uint32_t, __builtin_bit_cast(half2_t, (x & 0x00f000f0u) | magic) - z16);
          const uint32_t bw2 = __builtin_bit_cast(uint32_t, __builtin_bit_cast(half2_t, (y & 0x000f000fu) | magic) - z1);
          const uint32_t bw3 = __builtin_bit_cast(uint32_t, __builtin_bit_cast(half2_t, (y & 0x00f000f0u) | magic) - z16);
          b[ct] = __builtin_bit_cast(half8_t, uint4v{bw0, bw1, bw2, bw3});
        } else {
          // byte containers: the slice's eight s8 codes are two dwords in k order.  q ^ 0x80 = q + 128 as a u8 under the high
          // byte 0x64 = 1024 + 128 + q as fp16; minus 1024 + 128 + zb: q - zb in [-255, 255], exact (products < 2^16, a slice's
          // sum < 2^21)
          int zb = 0;
          if constexpr (ASYM) zb = int(int8_t((wc.zw[ct] >> (8 * e)) & 0xffu));
          const half2_t zc = __builtin_bit_cast(half2_t, uint32_t(0x6480 + zb) * 0x00010001u);
          const uint32_t x0 = (j == 0 ? wc.rec[ct].x : wc.rec[ct].z) ^ 0x80808080u, x1 = (j == 0 ? wc.rec[ct].y : wc.rec[ct].w) ^ 0x80808080u;
          const uint32_t bw0 = __builtin_bit_cast(uint32_t, __builtin_bit_cast(half2_t, __builtin_amdgcn_perm(0x64646464u, x0, 0x04010400u)) - zc);
          const uint32_t bw1 = __builtin_bit_cast(uint32_t, __builtin_bit_cast(half2_t, __builtin_amdgcn_perm(0x64646464u, x0, 0x04030402u)) - zc);
          const uint32_t bw2 = __builtin_bit_cast(uint32_t, __builtin_bit_cast(half2_t, __builtin_amdgcn_perm(0x64646464u, x1, 0x04010400u)) - zc);
          const uint32_t bw3 = __builtin_bit_cast(uint32_t, __builtin_bit_cast(half2_t, __builtin_amdgcn_perm(0x64646464u, x1, 0x04030402u)) - zc);
          b[ct] = __builtin_bit_cast(half8_t, uint4v{bw0, bw1, bw2, bw3});
        }
      };
      if constexpr (RH > 1) {  // shared by the row halves: all of them first
#pragma unroll
        for (int ct = 0; ct < CT; ct++) make_b(ct);
      }
#pragma unroll
      for (int h = 0; h < RH; h++) {
        half8_t a[4];
        floatx4 sa[4];
#pragma unroll
        for (int rt = 0; rt < 4; rt++) {
          a[rt] = *reinterpret_cast<const half8_t*>(a_lds + size_t(g) * G::kPlane + size_t(h * 64 + rt * 16 + nn) * G::kRow + q * 16);
          sa[rt] = *reinterpret_cast<const floatx4*>(sa_lds + q * G::kSaStride + h * 64 + rt * 16 + 4 * g);  // rows 4g .. 4g+3
        }
#pragma unroll
        for (int ct = 0; ct < CT; ct++) {
          if constexpr (RH == 1) make_b(ct);  // one use: built right before it (registers)
          const floatx4 zero = {0.f, 0.f, 0.f, 0.f};
          floatx4 d[4];
#pragma unroll
          for (int rt = 0; rt < 4; rt++)
            d[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[rt], b[ct], (JG > 1 && !first) ? dch[h][ct][rt] : zero, 0, 0, 0);
          if (JG > 1 && !last) {  // compile-time once unrolled: the group goes on with the next slice
#pragma unroll
            for (int rt = 0; rt < 4; rt++) dch[h][ct][rt] = d[rt], sa_g[h][rt] = sa[rt];
            sb_g[ct] = sb[ct];
            continue;
          }
          const float2v sb2 = {sb[ct], sb[ct]};
#pragma unroll
          for (int rt = 0; rt < 4; rt++) {
            // fma(float(isum), scale_a * scale_b, acc), two accumulators per instruction (v_pk_mul_f32, v_pk_fma_f32)
            acc[h][ct][rt][0] = __builtin_elementwise_fma(float2v{d[rt].x, d[rt].y}, float2v{sa[rt].x, sa[rt].y} * sb2, acc[h][ct][rt][0]);
            acc[h][ct][rt][1] = __builtin_elementwise_fma(float2v{d[rt].z, d[rt].w}, float2v{sa[rt].z, sa[rt].w} * sb2, acc[h][ct][rt][1]);
          }
        }
      }
    }
  };
  KRec w0, w1;  // even and odd k-steps: each is refilled one k-step ahead of its use, across chunk boundaries as well
  if (active) fetch_w(0, w0);
  fetch_a(0);
  static_assert(CS == 1 || CS % 2 == 0, "the two record buffers alternate by k-step parity inside a chunk");
  if constexpr (G::kBufs == 2) {
    static_assert(CS % 2 == 0, "the next chunk is written in the middle of the current one");
    write_chunk(g2_smem);
    if (CS < p.ksteps) fetch_a(CS);
    for (int s0 = 0; s0 < p.ksteps; s0 += CS) {
      unsigned char* cur = g2_smem + ((s0 / CS) & 1) * G::kBuf;
      unsigned char* nxt = g2_smem + (((s0 / CS) & 1) ^ 1) * G::kBuf;
      __syncthreads();  // this chunk is written by everyone, the previous one (the buffer written next) is consumed by everyone
#pragma unroll
      for (int t = 0; t < CS; t++) {
        const int s = s0 + t;
        if (s >= p.ksteps) break;
        if (active) {
          fetch_w(s + 1, (t & 1) ? w0 : w1);  // (all zero beyond the last k-step)
          kstep((t & 1) ? w1 : w0, s, t, cur);
        }
        if (t == CS / 2 - 1 && s0 + CS < p.ksteps) {
          write_chunk(nxt);
          if (s0 + 2 * CS < p.ksteps) fetch_a(s0 + 2 * CS);
        }
      }
    }
  } else {
    for (int s = 0; s < p.ksteps; s += 2) {
      if (s % CS == 0) stage(s);
      if (active) {
        fetch_w(s + 1, w1);  // (all zero beyond the last k-step)
        kstep(w0, s, s % CS, g2_smem);
      }
      if (s + 1 >= p.ksteps) break;
      if ((s + 1) % CS == 0) stage(s + 1);
      if (active) {
        fetch_w(s + 2, w0);
        kstep(w1, s + 1, (s + 1) % CS, g2_smem);
      }
    }
  }
  if (!active) return;
#pragma unroll
  for (int ct = 0; ct < CT; ct++) {
    const int col = (tile0 + ct) * 16 + nn;
    if (tile0 + ct >= ntiles || col >= p.n) continue;
#pragma unroll
    for (int h = 0; h < RH; h++)
#pragma unroll
      for (int rt = 0; rt < 4; rt++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int row = r0 + h * 64 + rt * 16 + 4 * g + i;
          if (row >= p.m) continue;
          float v = (i & 1) ? acc[h][ct][rt][i >> 1].y : acc[h][ct][rt][i >> 1].x;
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
}

template <bool F, int S, int D, int RH, int CT, int WV>
hipError_t launch_i8mfma2_a(bool asym, dim3 grid, hipStream_t st, const I8Gemm2Params& p) {
  auto go = [&](auto kern) {
    static const hipError_t attr =
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(I8G2Geom<RH>::kLds));
    if (attr != hipSuccess) return attr;
    hipLaunchKernelGGL(kern, grid, dim3(64 * WV), I8G2Geom<RH>::kLds, st, p);
    return hipGetLastError();
  };
  return asym ? go(i8mfma2_kernel<F, D, S, true, RH, CT, WV>) : go(i8mfma2_kernel<F, D, S, false, RH, CT, WV>);
}
template <bool F, int S, int RH, int CT, int WV>
hipError_t launch_i8mfma2_t(int sdt, bool asym, int m, int ntiles, hipStream_t st, const I8Gemm2Params& p) {
  const dim3 grid(unsigned((ntiles + WV * CT - 1) / (WV * CT)), unsigned((m + 64 * RH - 1) / (64 * RH)));
  if (sdt == 0) return launch_i8mfma2_a<F, S, 0, RH, CT, WV>(asym, grid, st, p);
  if (sdt == 1) return launch_i8mfma2_a<F, S, 1, RH, CT, WV>(asym, grid, st, p);
  return launch_i8mfma2_a<F, S, 2, RH, CT, WV>(asym, grid, st, p);
}
template <bool F, int S>
hipError_t launch_i8mfma2(int sdt, bool asym, int m, int ntiles, hipStream_t st, const I8Gemm2Params& p) {
  // tile: "i8_tile" 0 = by size (4 once 64 x 256 workgroups give every CU one, else 1), 1 = 64 x 64 as four waves, 4 = 64 x 256 as
  // sixteen waves of one column tile each.  (The two four-wave large tiles measured in profiles/r03_i8_prefill_kernels.json — 2 = 64 x 256
  // with four column tiles per wave, 3 = 128 x 128 — are not instantiated any more: 36 kernels of 128 unrolled MFMAs each took
  // minutes to compile and lost to 4 by 10 - 20 %; the template still covers them: launch_i8mfma2_t<F, S, 1, 4, 4> / <F, S, 2, 2, 4>.)
  const int force = i8_tile_forced();
  const bool wide = size_t((ntiles + 15) / 16) * size_t((m + 63) / 64) >= 256;
  const int tile = (force == 1 || force == 4) ? force : (wide ? 4 : 1);
  if (tile == 4) return launch_i8mfma2_t<F, S, 1, 1, 16>(sdt, asym, m, ntiles, st, p);
  return launch_i8mfma2_t<F, S, 1, 1, 4>(sdt, asym, m, ntiles, st, p);
}
}  // namespace

hipError_t NS_I8G2_NAME(int sdt, bool asym, int m, int ntiles, hipStream_t st, const I8Gemm2Params& p) {
  return launch_i8mfma2<NS_I8G2_FOUR, NS_I8G2_SPS>(sdt, asym, m, ntiles, st, p);
}

}  // namespace ns
