// ns_gemm.hip — prefill / large-M GEMM, second generation:  C[M][N] = A[M][K] * dequant(W)      (MFMA-bound)
//
// Design for MI355X (one workgroup = 256 threads = 4 waves, 256 x 128 output tile, 64-deep K chunks):
//   * each wave owns a 128 x 64 sub-tile = 8 x 4 v_mfma_f32_16x16x32_f16 accumulators (128 AGPR/VGPRs): per 32-deep
//     k-slice it reads 8 A and 4 B fragments from LDS for 32 MFMAs — at 64 x 64 per wave the four SIMDs of a CU ask
//     LDS for exactly its 128 B/clk, i.e. LDS would cap the kernel at the MFMA peak with no slack
//   * weights are dequantised ONCE per workgroup and chunk into LDS, already in MFMA B-fragment order and already
//     multiplied by their group scale (fp16: (code - zp) is an exact small integer, the product is rounded to 11 bits,
//     2^-12 relative, far inside the 1e-3 budget), so the inner loop is a pure MFMA accumulate chain — the first
//     generation applied the group scale to every MFMA result with 4 VALU FMAs, as much VALU time as MFMA time
//   * activations are fp16 in memory (caller's shadow, or one conversion pass into a per-stream scratch buffer that
//     is zero padded to the chunk size): 16-byte global loads, no conversion in the loop
//   * two LDS stages; the global loads of chunk c+1 are in flight while chunk c is multiplied; one barrier per chunk
//   * workgroup id -> (row block, column block) keeps the column blocks that share an XCD's L2 together
// Reference semantics: w = (code - zp) * scale, fp32 accumulate (bestla/bestla/kernel_ref.h:1027-1127 dequant,
// bestla_wrapper.h LauncherBase GEMM loop); A rounded to fp16 (north_star: fp16 activations).
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <map>
#include <type_traits>
#include <mutex>
#include <vector>

#include "../../include/ns_bestla.h"
#include "ns_common.h"
#include "ns_dev.h"

namespace ns {

#ifndef NS_G2_MI
#define NS_G2_MI 4
#endif
#ifndef NS_G2_STAGES
#define NS_G2_STAGES 2
#endif
#ifndef NS_G2_OCC
#define NS_G2_OCC 2
#endif
constexpr int kG2Stages = NS_G2_STAGES;  // LDS stages (2: one barrier per chunk; 1: two barriers, more workgroups per CU)
constexpr int kG2MI = NS_G2_MI;  // 16-row MFMA tiles per wave along M: wave tile = 16*MI x 64, workgroup = 32*MI x 128
constexpr int kG2BM = 32 * kG2MI, kG2Tiles = 8, kG2KC = 64;  // rows x 8 column tiles (128 columns) x 64-deep chunks
constexpr int kG2AStr = kG2KC + 8;                    // halves per A row in LDS (+16 B skews banks)
constexpr int kG2StageBytes = kG2BM * kG2AStr * 2 + kG2Tiles * 2 * 64 * 16;

struct Gemm2Params {
  const _Float16* a16;  // [m][lda16] fp16, zero padded to a multiple of 64 columns
  int lda16, m, k, n;
  int nchunks;          // ceil(k / 64)
  int ksteps, ntiles;
  const uint4* codes;
  const void* scales;
  const int8_t* zps;
  uint32_t codes_bytes, scales_bytes, zps_bytes;
  uint32_t qstride, sstride, zstride;
  float* c;
  _Float16* c16;  // optional fp16 shadow of C (same leading dimension) for the next GEMM's activations
  int ldc;
  int srows, srow_mul, srow_shift;
  int epilogue;
  const float* d;
  int ldd;
  int nbn, cpx;  // column blocks; column blocks per XCD
  // gemm3_kernel, fused QKV with RoPE(q, k) + kv-cache append as its epilogue (round 5; ns_qkv_rope at prefill size): segment 0 = q (rotated, fp32),
  // 1 = k (rotated -> fp16 cache [+ fp32]), 2 = v (-> fp16 cache [+ fp32]); cos / sin pairs per (row, pair) from the caller's table
  float seg_spre[3], seg_spost[3];  // fused QKV: each matrix's own factors (round 5: real models' q / k / v scales differ in range)
  int rope_on, rope_hs, rope_npast;
  const float2* rope_tab;
  _Float16 *rope_kc, *rope_vc;
  long long rope_csl, rope_chead;
  int ksplit;    // > 1: split-K (few output tiles): workgroup z = blockIdx.y takes chunks [z*cps, (z+1)*cps) and writes a
  int cps;       //      raw fp32 partial tile to `part` [ksplit][m][n]; gemm2_reduce_kernel sums them and applies the epilogue
  float* part;
  float spre, spost;  // power-of-two pair (spre * spost == 1): group scales are multiplied by spre before they are rounded
                      // to fp16 and the accumulators by spost on the way out, so that weights with very small or very
                      // large magnitudes stay inside fp16's normal range (1, 1 for ordinary LLM weights)
  F4Lut lut;
  // gemm3_kernel, fused QKV (ip_fusion_qkv.cpp:84-86 at GEMM size): up to three matrices of one K and one format side by
  // side along the column blocks; nseg <= 1: the single matrix above
  int nseg;
  int seg_bn0[3];   // first column block of matrix s
  int seg_n[3];
  const void* seg_codes[3];
  const void* seg_scales[3];
  const int8_t* seg_zps[3];
  uint32_t seg_codes_bytes[3], seg_scales_bytes[3], seg_zps_bytes[3];
  float* seg_c[3];
  _Float16* seg_c16[3];
  int bm3;   // gemm3_kernel: rows of the workgroup tile (256 / 128)
  int tall3; // gemm3_kernel, bm3 = 256: waves 1 x 4 of 256 x 32 instead of 2 x 2 of 128 x 64
  int diag;  // NS_G3_DIAG (diagnostics): 1 = skip the output stores, 2 = skip the main loop, 3 = DMA and barriers only, 4 = no DMA
  // gemm3_kernel, fused gate/up (round 5; ip_fusion_ffn.cpp:364-406 at GEMM size): codes / scales / zps = W1 (gate), *2 = W3 (up), same
  // shape and format; a column block is 4 tile PAIRS = 64 output columns, wave w multiplies gate tile and up tile 4 bn + w, the MFMA
  // layout puts gate and up of one (row, column) into the same lane: out = act(gate) * up in registers.  c / c16 / c2 are each optional.
  int dual;
  const void* codes2;
  const void* scales2;
  const int8_t* zps2;
  float* c2;  // optional act(gate) (the reference's tmp1)
  int wide;   // 1: the cross-wave epilogue (waves 1 x 4 only): whole tile rows leave as 256 / 512-byte runs, the fp16 shadow as 16-byte stores
};

template <int KIND, int SPS, int SK, bool ASYM>
__global__ __launch_bounds__(256, NS_G2_OCC) void gemm2_kernel(const Gemm2Params p) {
  constexpr int NJ = (KIND == WK_INT8) ? 2 : 4;   // 32-deep slices per k-step record
  constexpr int CPS = NJ / 2;                      // 64-deep chunks per k-step
  constexpr int SBYTES = SPS * (SK == SK_F32 ? 4 : 2);
  using Corr = CorrRaw<SPS, SK, ASYM>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: keeps buffer soffsets in SGPRs
  const int l = tid & 63, nn = l & 15, g = l >> 4;
  const int wm = w >> 1, wn = w & 1;
  // workgroup -> (bm, bn): consecutive ids go to consecutive XCDs, so give each XCD its own run of column blocks and
  // let it sweep the row blocks: an XCD's L2 then holds its few column blocks' weights while A streams through once
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  const int bn = xcd * p.cpx + local % p.cpx, bm = local / p.cpx;
  if (bn >= p.nbn) return;
  const int tile0 = bn * kG2Tiles, row0 = bm * kG2BM;

  const Rsrc rq = make_rsrc(p.codes, p.codes_bytes);
  const Rsrc rs = make_rsrc(p.scales, p.scales_bytes);
  const Rsrc rz = make_rsrc(p.zps, p.zps_bytes);
  const Rsrc ra = make_rsrc(p.a16, uint32_t(p.m) * uint32_t(p.lda16) * 2u);  // rows >= m read as zeros
  const I4Consts i4c = {0x000f000fu, 0x00f000f0u, 0x64006400u};

  floatx4 acc[kG2MI][4];
#pragma unroll
  for (int mi = 0; mi < kG2MI; mi++)
#pragma unroll
    for (int ni = 0; ni < 4; ni++) acc[mi][ni] = floatx4{0.f, 0.f, 0.f, 0.f};

  // ---- staging registers of the NEXT chunk ----
  uint4v areg[kG2MI];
  uint32_t breg[2][4];  // 16 weight codes per staging lane (INT8: 16 B; 4-bit: 8 B)
  Corr creg[2];

  auto load_chunk = [&](int c) {
    const int s = c / CPS, h = c % CPS;  // k-step record and its 64-deep half
#pragma unroll
    for (int it = 0; it < kG2MI; it++) {
      const int u = tid + 256 * it, r = u >> 3, ch = u & 7;
      const uint32_t off = (uint32_t(row0 + r) * uint32_t(p.lda16) + uint32_t(c * kG2KC + ch * 8)) * 2u;
      areg[it] = __builtin_bit_cast(uint4v, __builtin_amdgcn_raw_buffer_load_b128(ra, off, 0, 0));
    }
    const uint32_t srow = uint32_t(s * p.srow_mul) >> p.srow_shift;
#pragma unroll
    for (int r = 0; r < 2; r++) {
      const int tile = tile0 + w + 4 * r;
      const bool live = tile < p.ntiles;
      const uint32_t soff = (uint32_t(tile) * p.ksteps + s) * p.qstride;
      if constexpr (KIND == WK_INT8) {
        const uint4v v = live ? __builtin_bit_cast(uint4v, __builtin_amdgcn_raw_buffer_load_b128(rq, l * 16, soff, 0))
                              : uint4v{0, 0, 0, 0};
        breg[r][0] = v.x, breg[r][1] = v.y, breg[r][2] = v.z, breg[r][3] = v.w;
      } else {
        typedef uint32_t uint2v __attribute__((ext_vector_type(2)));
        const uint2v v = live ? __builtin_bit_cast(uint2v, __builtin_amdgcn_raw_buffer_load_b64(rq, l * 16 + 8 * h, soff, 0))
                              : uint2v{0x88888888u, 0x88888888u};
        breg[r][0] = v.x, breg[r][1] = v.y;
      }
      const uint32_t crow = uint32_t(live ? tile : 0) * p.srows + srow;
      corr_issue<SPS, SK, ASYM>(rs, rz, nn * SBYTES, nn * SPS, crow * p.sstride, crow * p.zstride, creg[r]);
    }
  };

  // registers -> LDS stage: A as is; B dequantised, scaled, in MFMA B-fragment order [tile][slice][g][nn] x 16 B
  auto store_chunk = [&](int c, unsigned char* stage) {
    const int h = c % CPS;
    _Float16* a_lds = reinterpret_cast<_Float16*>(stage);
    uint4v* b_lds = reinterpret_cast<uint4v*>(stage + kG2BM * kG2AStr * 2);
#pragma unroll
    for (int it = 0; it < kG2MI; it++) {
      const int u = tid + 256 * it, r = u >> 3, ch = u & 7;
      *reinterpret_cast<uint4v*>(a_lds + r * kG2AStr + ch * 8) = areg[it];
    }
#pragma unroll
    for (int r = 0; r < 2; r++) {
      const int tl = w + 4 * r;
      float sc[4], zp[4];
      corr_decode<SPS, SK, ASYM, NJ>(creg[r], sc, zp);
#pragma unroll
      for (int jj = 0; jj < 2; jj++) {
        const int j = (KIND == WK_INT8) ? jj : 2 * h + jj;  // slice of the k-step record this chunk slice is
        // corr_decode fills sc[j] / zp[j] for j < 4 with compile-time indices only: select without dynamic indexing
        float s_j, z_j;
        if constexpr (KIND == WK_INT8) {
          s_j = sc[jj], z_j = zp[jj];
        } else {
          // bit select on registers: a ternary on the runtime `h` becomes a dynamically indexed scratch array
          const uint32_t hm = 0u - uint32_t(h);
          s_j = __builtin_bit_cast(float, (__builtin_bit_cast(uint32_t, sc[jj]) & ~hm) |
                                              (__builtin_bit_cast(uint32_t, sc[2 + jj]) & hm));
          z_j = __builtin_bit_cast(float, (__builtin_bit_cast(uint32_t, zp[jj]) & ~hm) |
                                              (__builtin_bit_cast(uint32_t, zp[2 + jj]) & hm));
        }
        (void)j;
        half8_t b;
        if constexpr (KIND == WK_INT4) {
          const _Float16 zl = (_Float16)(-1032.f - z_j), zh = (_Float16)(-72.f - z_j);
          b = cvt_i4x8(breg[r][jj], i4c, half2_t{zl, zl}, half2_t{zh, zh});
        } else if constexpr (KIND == WK_INT8) {
          const _Float16 zo = (_Float16)(-1152.f - z_j);
          b = cvt_i8x8(breg[r][2 * jj], breg[r][2 * jj + 1], half2_t{zo, zo});
        } else {
          b = cvt_f4x8(breg[r][jj], p.lut);
        }
        const _Float16 sh = (_Float16)(s_j * p.spre);
        b = b * half8_t{sh, sh, sh, sh, sh, sh, sh, sh};
        b_lds[((tl * 2 + jj) * 4 + g) * 16 + nn] = __builtin_bit_cast(uint4v, b);
      }
    }
  };

  auto compute = [&](const unsigned char* stage) {
    const _Float16* a_lds = reinterpret_cast<const _Float16*>(stage);
    const uint4v* b_lds = reinterpret_cast<const uint4v*>(stage + kG2BM * kG2AStr * 2);
    // all fragments of the chunk are requested up front (one wave per SIMD: nobody else hides the LDS latency);
    // the MFMAs of slice 0 then run while slice 1's fragments arrive
    half8_t bf[2][4], af[2][kG2MI];
#pragma unroll
    for (int jj = 0; jj < 2; jj++) {
#pragma unroll
      for (int ni = 0; ni < 4; ni++)
        bf[jj][ni] = __builtin_bit_cast(half8_t, b_lds[(((wn * 4 + ni) * 2 + jj) * 4 + g) * 16 + nn]);
#pragma unroll
      for (int mi = 0; mi < kG2MI; mi++)
        af[jj][mi] =
            *reinterpret_cast<const half8_t*>(a_lds + (wm * 16 * kG2MI + mi * 16 + nn) * kG2AStr + 32 * jj + 8 * g);
    }
    __builtin_amdgcn_sched_barrier(0);  // hipcc otherwise sinks each fragment load to just before its first MFMA
#pragma unroll
    for (int jj = 0; jj < 2; jj++)
#pragma unroll
      for (int mi = 0; mi < kG2MI; mi++)
#pragma unroll
        for (int ni = 0; ni < 4; ni++)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[jj][mi], bf[jj][ni], acc[mi][ni], 0, 0, 0);
  };

  unsigned char* stage0 = smem;
  const int cbeg = p.ksplit > 1 ? int(blockIdx.y) * p.cps : 0;
  const int cend = p.ksplit > 1 ? min(p.nchunks, cbeg + p.cps) : p.nchunks;
  if constexpr (kG2Stages == 2) {
    unsigned char* stage1 = smem + kG2StageBytes;
    load_chunk(cbeg);
    store_chunk(cbeg, stage0);
    __syncthreads();
    for (int c = cbeg; c < cend; c++) {
      unsigned char* cur = ((c - cbeg) & 1) ? stage1 : stage0;
      unsigned char* nxt = ((c - cbeg) & 1) ? stage0 : stage1;
      const bool more = c + 1 < cend;
      if (more) load_chunk(c + 1);
      compute(cur);
      if (more) store_chunk(c + 1, nxt);
      __syncthreads();
    }
  } else {
    load_chunk(cbeg);
    for (int c = cbeg; c < cend; c++) {
      store_chunk(c, stage0);
      __syncthreads();
      if (c + 1 < cend) load_chunk(c + 1);
      compute(stage0);
      __syncthreads();
    }
  }

  // ---- epilogue ----
#pragma unroll
  for (int mi = 0; mi < kG2MI; mi++)
#pragma unroll
    for (int ni = 0; ni < 4; ni++) {
      const int col = (tile0 + wn * 4 + ni) * 16 + nn;
      if (col >= p.n) continue;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = row0 + wm * 16 * kG2MI + mi * 16 + 4 * g + r;
        if (row >= p.m) continue;
        float v = acc[mi][ni][r] * p.spost;
        if (p.ksplit > 1) {  // raw partial; epilogue happens in gemm2_reduce_kernel
          p.part[(size_t(blockIdx.y) * p.m + row) * p.n + col] = v;
          continue;
        }
        const float dv = p.d ? p.d[size_t(row) * p.ldd + col] : 0.f;
        switch (p.epilogue) {
          case 1: v = v + dv; break;            // custom::epilogue::Add
          case 2: v = v * dv; break;            // custom::epilogue::Mul
          case 3: v = epi_gelu(v + dv); break;  // custom::epilogue::Add_Gelu
          case 4: v = epi_gelu(v); break;
          case 5: v = epi_silu(v); break;
          default: break;
        }
        p.c[size_t(row) * p.ldc + col] = v;
        if (p.c16) p.c16[size_t(row) * p.ldc + col] = (_Float16)v;
      }
    }
}

// ================================================================================================================
// gemm3_kernel — third generation.  What bounded gemm2_kernel was the LDS pipe, not the matrix cores
// (profiles/r01i: SQ_VALU_MFMA_BUSY 0.39): every chunk the workgroup WROTE 16 KiB of A and 16 KiB of dequantised B
// with ds_write_b128 (~79 B/clk, MI355X_MICROARCH.md LDS table) and read both back.  Here
//   * A (fp16) goes HBM/L2 -> LDS by DMA (buffer_load ... lds, 16 B per lane): no staging registers, no ds_write.  The
//     image is row-major [256 rows][64 halves] with the 16-byte pieces of a row XOR-ed by (row >> 1) & 7 — applied on
//     the SOURCE address of the DMA (the LDS side of a DMA is lane-linear) and on the fragment reads — so every
//     ds_read_b128 lane group covers all 64 banks once;
//   * B never touches LDS: each wave loads the raw records of ITS four column tiles straight into registers (16 B per
//     lane = one record = 128 deep for 4-bit codes) and dequantises them into MFMA B fragments itself.  A record is
//     shared by the two waves that split the rows (one extra L1/L2 hit), in exchange the workgroup moves 1/4 of the
//     LDS bytes and no LDS writes at all;
//   * workgroup tile 256 x 128, four waves of 128 x 64 (8 x 4 accumulator tiles: 12 fragment reads feed 32 MFMAs per
//     32-deep slice), two 32 KiB A stages -> two workgroups per CU, whose waves fill each other's dequantisation
//     and barrier bubbles.
// Synchronisation per 64-deep chunk: s_waitcnt vmcnt(0) (the wave's own DMA of this chunk, issued one chunk ago, has
// landed) -> s_barrier (everybody's has, and everybody is done reading the other stage) -> issue the DMA of the next
// chunk into the other stage -> multiply this chunk.
constexpr int kG3BM = 256, kG3KC = 64, kG3Tiles = 8;
// MFMA shape of the main loop (round 4): v_mfma_f32_32x32x16_f16 — twice the multiply-adds per instruction and per operand
// register of v_mfma_f32_16x16x32_f16 (8 vs 5 cycles per CU for 32 K vs 16 K MACs, MI355X_MICROARCH.md: 2382 vs 2075 TFLOPS
// micro-benchmark ceilings).  A lane is then (row or column l & 31, k half l >> 5) of a 32 x 16 operand: the A image and its
// swizzle serve that read as they are (the sixteen lanes of a ds_read_b128 group still cover sixteen distinct 16-byte slots),
// a B operand is word t of record lane (2 e + k half, column & 15) of the column's tile, e = which 16 of the 32-deep slice.
// Measured (profiles/r04t_gemm3_mfma_shape_ab.txt, M = 2048, same box, alternating): 4096^2 888 vs 882 TFLOPS, 11008 x 4096 945 vs 983,
// 4096 x 11008 988 vs 974, 32000 x 4096 984 vs 1015 (32x32x16 vs 16x16x32) — no gain: the loop is not bound by the matrix pipe's issue rate.
// The 16x16x32 loop stays the default; NS_G3_M32=1 builds this one (parity-tested on both: tests/test_gpu_gemm3.py, test_gpu_fuzz.py).
#ifndef NS_G3_M32
#define NS_G3_M32 0
#endif
constexpr bool kG3M32 = NS_G3_M32 != 0;
typedef float floatx16 __attribute__((ext_vector_type(16)));
// rows from which the third-generation kernel is used (tuning: g3_min_m).  Round 3 (scripts/m_sweep.py, profiles/r03_m_sweep.txt):
// with its 128-row tile it beats gemm2_kernel everywhere below the old threshold of 192 — 4096 x 4096: 17.5 vs 24.2 us at 65
// rows, 21.7 vs 34.2 at 128, 25.0 vs 36.4 at 191; 11008 x 4096: 26.0 vs 44.6, 29.0 vs 49.6, 40.5 vs 71.0 — and the streaming
// kernel from 17 rows on wide outputs (11008 x 4096: 22.5 vs 33.0 us at 17 rows, 25.9 vs 44.8 at 64; ns_api.cpp decides)
constexpr int kG3MinM = 2;  // (the caller decides from where on: ns_api.cpp forward_impl)
constexpr int kG3StageBytes = kG3BM * kG3KC * 2;  // 32 KiB
constexpr int kG3Stages = 2;
constexpr int kG3BStageMax = 8 * 2 * 1024 + 8 * 2 * 16 * 16 + 8 * 2 * 16 * 4;  // B stage upper bound: codes + scale rows + zero points

// BM: rows of the workgroup tile.  256: waves 2 (rows) x 2 (columns), wave tile 128 x 64.  128: waves 1 x 4, wave tile 128 x 32
// — the same MFMA : dequantisation ratio, half the A stage (three workgroups per CU), twice the tiles: for outputs with
// few tiles (4096 wide at 2048 rows: 256 of the tall tiles, one per CU) instead of a K split and its reduction pass.
// TALL (BM = 256 only): waves 1 x 4 like BM = 128, but 256 rows each — wave tile 256 x 32: the dequantisation of a B fragment
// (the VALU work of the loop) is shared by 16 row fragments instead of 8, the same 128 accumulator registers per lane
template <int KIND, int SPS, int SK, bool ASYM, int BM, bool TALL = false, bool DUAL = false>
__global__ __launch_bounds__(256, (BM == 256 ? 2 : 3)) void gemm3_kernel(const Gemm2Params p) {  // (128- / 64-row tiles: three workgroups per CU = three waves per SIMD, at most 168 registers)
  static_assert(!TALL || BM == 256, "the tall wave tile is a 256-row workgroup tile");
  static_assert(!DUAL || (BM <= 128 && !kG3M32), "gate/up pairs: waves 1 x 4 with two column tiles each");
  constexpr bool SQ = BM == 256 && !TALL;       // waves 2 x 2
  constexpr int MI = (SQ ? 128 : BM) / 16;      // row fragments (16 rows) per wave
  constexpr int NIW = SQ ? 4 : 2;               // column tiles (16 wide) per wave
  constexpr int APW = BM / 32;                   // A DMA pieces (8 rows each) per wave and chunk
  constexpr int kStage = BM * kG3KC * 2;         // bytes of one A stage
  constexpr bool B8 = KIND == WK_INT8;          // 8-bit codes: a record is 64 deep, two per 128-deep superstep
  constexpr int NJ = B8 ? 2 : 4;                // 32-deep slices per record
  constexpr int RPS = B8 ? 2 : 1;               // records per superstep (128 deep = chunks 2u, 2u + 1)
  constexpr int SBYTES = SPS * (SK == SK_F32 ? 4 : 2);
  constexpr bool M32 = kG3M32;
  constexpr int MB = MI / 2, NP = NIW / 2;       // 32-row / 32-column fragments of the wave tile (M32)
  using Corr = CorrRaw<SPS, SK, ASYM>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __attribute__((address_space(3))) unsigned char* LdsPtr;

  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = tid & 63, nn = l & 15, g = l >> 4;
  const int wm = SQ ? (w >> 1) : 0, wn = SQ ? (w & 1) : w;
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  const int bn = xcd * p.cpx + local % p.cpx, bm = local / p.cpx;
  if (bn >= p.nbn) return;
  // fused QKV: the column block belongs to one of up to three matrices (indexed scalar loads of its pointers)
  int sg = 0;
  if (p.nseg > 1) sg = int(bn >= p.seg_bn0[1]) + int(p.nseg > 2 && bn >= p.seg_bn0[2]);
  const int bnl = p.nseg > 1 ? bn - p.seg_bn0[sg] : bn;  // column block inside its matrix
  const int n_cols = p.nseg > 1 ? p.seg_n[sg] : p.n;
  float* const c_out = p.nseg > 1 ? p.seg_c[sg] : p.c;
  _Float16* const c16_out = p.nseg > 1 ? p.seg_c16[sg] : p.c16;
  // the weight's fp16-range factors (scales are multiplied by spre before the fp16 product, results by spost): per matrix in a fused launch
  const float spre_ = p.nseg > 1 ? p.seg_spre[sg] : p.spre, spost_ = p.nseg > 1 ? p.seg_spost[sg] : p.spost;
  const int tile0 = DUAL ? bnl * 4 + w : bnl * kG3Tiles + wn * NIW, row0 = bm * BM;  // DUAL: the tile of BOTH matrices (output columns 16 tile0 ..)

  const Rsrc rq = p.nseg > 1 ? make_rsrc(p.seg_codes[sg], p.seg_codes_bytes[sg]) : make_rsrc(p.codes, p.codes_bytes);
  const Rsrc rs = p.nseg > 1 ? make_rsrc(p.seg_scales[sg], p.seg_scales_bytes[sg]) : make_rsrc(p.scales, p.scales_bytes);
  const Rsrc rz = p.nseg > 1 ? make_rsrc(p.seg_zps[sg], p.seg_zps_bytes[sg]) : make_rsrc(p.zps, p.zps_bytes);
  // DUAL: the up matrix (LDS tile slot 2 w + 1 of the wave; slot 2 w is the gate tile)
  const Rsrc rq2 = DUAL ? make_rsrc(p.codes2, p.codes_bytes) : rq;
  const Rsrc rs2 = DUAL ? make_rsrc(p.scales2, p.scales_bytes) : rs;
  const Rsrc rz2 = DUAL ? make_rsrc(p.zps2, p.zps_bytes) : rz;
  const Rsrc ra = make_rsrc(p.a16, uint32_t(p.m) * uint32_t(p.lda16) * 2u);  // rows >= m read as zeros
  const I4Consts i4c = {0x000f000fu, 0x00f000f0u, 0x64006400u};

  floatx4 acc[M32 ? 1 : MI][M32 ? 1 : NIW];
  floatx16 acc32[M32 ? MB : 1][M32 ? NP : 1];
  if constexpr (M32) {
#pragma unroll
    for (int mb = 0; mb < MB; mb++)
#pragma unroll
      for (int pp = 0; pp < NP; pp++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc32[mb][pp][r] = 0.f;
  } else {
#pragma unroll
    for (int mi = 0; mi < MI; mi++)
#pragma unroll
      for (int ni = 0; ni < NIW; ni++) acc[mi][ni] = floatx4{0.f, 0.f, 0.f, 0.f};
  }

  // ---- A: DMA of chunk c into a stage.  Wave w, request i covers rows (8 w + i) * 8 .. + 7; lane l writes LDS piece
  //      l & 7 of row l >> 3 of them and therefore FETCHES piece (l & 7) ^ ((row >> 1) & 7) ----
  const uint32_t lrow = uint32_t(l >> 3);
  uint32_t a_voff[2];  // (row >> 1) & 7 = 4 * (i & 1) + (l >> 4): two per-lane source offsets, for even and odd i
#pragma unroll
  for (int par = 0; par < 2; par++)
    a_voff[par] = (uint32_t(row0) + uint32_t(w) * uint32_t(APW * 8) + lrow) * uint32_t(p.lda16) * 2u +
                  ((uint32_t(l & 7) ^ (uint32_t(4 * par) + uint32_t(l >> 4))) << 4);
  const uint32_t a_istride = 8u * uint32_t(p.lda16) * 2u;  // source bytes between consecutive requests of a wave
  auto issue_a_piece = [&](int c, int stage, int i) {
#if defined(__HIP_DEVICE_COMPILE__)
    const LdsPtr dst = (LdsPtr)(smem) + stage * kStage + (w * APW + i) * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, reinterpret_cast<__attribute__((address_space(3))) void*>(dst), 16,
                                             a_voff[i & 1], uint32_t(c) * (kG3KC * 2) + uint32_t(i) * a_istride, 0, 0);
#endif
  };
  auto issue_a = [&](int c, int stage) {
#pragma unroll
    for (int i = 0; i < APW; i++) issue_a_piece(c, stage, i);
  };
  // fragment read offsets of this lane: row wm * 128 + mi * 16 + nn, piece (4 jj + g) ^ ((nn >> 1) & 7)
  uint32_t a_roff[2];
#pragma unroll
  for (int jj = 0; jj < 2; jj++)
    a_roff[jj] = uint32_t(wm * 128 + nn) * 128u + ((uint32_t(4 * jj + g) ^ uint32_t((nn >> 1) & 7)) << 4);
  // M32: lane (row l & 31, k half l >> 5) of the 32 x 16 operand kk (0..3 inside the 64-deep chunk) reads piece 2 kk + half of
  // its row, swizzled like every piece by (row >> 1) & 7
  uint32_t a_roff32[4];
#pragma unroll
  for (int kk = 0; kk < 4; kk++)
    a_roff32[kk] = uint32_t(wm * 128 + (l & 31)) * 128u + ((uint32_t(2 * kk + (l >> 5)) ^ uint32_t(((l & 31) >> 1) & 7)) << 4);

  // ---- B: raw records + their scale / zero-point rows of the workgroup's 8 column tiles for one 128-deep superstep,
  //      HBM -> LDS by DMA as well (an ordinary load in this loop would make hipcc drain every DMA in flight at its
  //      first use — measured: s_waitcnt vmcnt(0) right behind the issue).  ONE B stage: the waves copy their records
  //      into registers at the start of the superstep, the DMA of the next superstep refills the stage one chunk later.
  //      Wave w fetches tiles 2w, 2w + 1. ----
  constexpr uint32_t kBCodes = kG3Tiles * RPS * 1024u;         // [tile][record][64 lanes x 16 B]
  constexpr uint32_t kBScal = kG3Tiles * RPS * 16u * SBYTES;   // [record][tile][16 columns x SBYTES]
  constexpr uint32_t kBZp = ASYM ? kG3Tiles * RPS * 16u * SPS : 0u;
  unsigned char* const b_lds = smem + kG3Stages * kStage;
  const uint32_t btile0 = DUAL ? uint32_t(bnl * 4 + w) : uint32_t(bnl * kG3Tiles + 2 * w);
  // scale rows: 16 * SBYTES bytes per (tile, row) = SBYTES lanes of 16 B; lanes [0, 2 * SBYTES) cover the wave's two tiles
  // (DUAL: the two slots are the same tile of two matrices — one request per matrix, lanes [0, SBYTES) each)
  const uint32_t s_lane_tile = DUAL ? 0u : uint32_t(l) / uint32_t(SBYTES), s_lane_piece = uint32_t(l) % uint32_t(SBYTES);
  const uint32_t s_voff = (btile0 + s_lane_tile) * uint32_t(p.srows) * p.sstride + s_lane_piece * 16u;
  const uint32_t z_lane_tile = DUAL ? 0u : uint32_t(l) / uint32_t(SPS), z_lane_piece = uint32_t(l) % uint32_t(SPS);
  const uint32_t z_voff = (btile0 + z_lane_tile) * uint32_t(p.srows) * p.zstride + z_lane_piece * 16u;
  auto issue_b = [&](int u) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int r = 0; r < RPS; r++) {
      const uint32_t s = uint32_t(min(u * RPS + r, p.ksteps - 1));
#pragma unroll
      for (int t = 0; t < 2; t++) {
        const LdsPtr dst = (LdsPtr)(b_lds) + ((2 * w + t) * RPS + r) * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(DUAL && t ? rq2 : rq, reinterpret_cast<__attribute__((address_space(3))) void*>(dst), 16, uint32_t(l) * 16u,
                                                 ((btile0 + (DUAL ? 0u : uint32_t(t))) * uint32_t(p.ksteps) + s) * p.qstride, 0, 0);
      }
      const uint32_t srow = (s * uint32_t(p.srow_mul)) >> p.srow_shift;
      if constexpr (DUAL) {
        if (l < SBYTES) {
#pragma unroll
          for (int t = 0; t < 2; t++) {
            const LdsPtr dst = (LdsPtr)(b_lds) + kBCodes + (r * kG3Tiles + 2 * w + t) * (16 * SBYTES);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(t ? rs2 : rs, reinterpret_cast<__attribute__((address_space(3))) void*>(dst), 16, s_voff,
                                                     srow * p.sstride, 0, 0);
          }
        }
        if constexpr (ASYM) {
          if (l < SPS) {
#pragma unroll
            for (int t = 0; t < 2; t++) {
              const LdsPtr dst = (LdsPtr)(b_lds) + kBCodes + kBScal + (r * kG3Tiles + 2 * w + t) * (16 * SPS);
              __builtin_amdgcn_raw_ptr_buffer_load_lds(t ? rz2 : rz, reinterpret_cast<__attribute__((address_space(3))) void*>(dst), 16, z_voff,
                                                       srow * p.zstride, 0, 0);
            }
          }
        }
      } else {
        if (l < 2 * SBYTES) {
          const LdsPtr dst = (LdsPtr)(b_lds) + kBCodes + (r * kG3Tiles + 2 * w) * (16 * SBYTES);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, reinterpret_cast<__attribute__((address_space(3))) void*>(dst), 16, s_voff,
                                                   srow * p.sstride, 0, 0);
        }
        if constexpr (ASYM) {
          if (l < 2 * SPS) {
            const LdsPtr dst = (LdsPtr)(b_lds) + kBCodes + kBScal + (r * kG3Tiles + 2 * w) * (16 * SPS);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rz, reinterpret_cast<__attribute__((address_space(3))) void*>(dst), 16, z_voff,
                                                     srow * p.zstride, 0, 0);
          }
        }
      }
    }
#endif
  };
  struct BRec {
    uint32_t q[NIW][4 * RPS];
    Corr c[NIW][RPS];
  };
  // M32: per PAIR of column tiles pp a lane holds, of the tile its column l & 31 lies in, the two record lanes it is an operand
  // lane of — (k slot 2 e + (l >> 5), column l & 15), e = 0, 1 — and that column's scales / zero points
  struct BRec32 {
    uint32_t q[NP > 0 ? NP : 1][2][4 * RPS];
    Corr c[NP > 0 ? NP : 1][RPS];
  };
  auto read_b32 = [&](BRec32& b) {
#pragma unroll
    for (int pp = 0; pp < NP; pp++) {
      const int t = wn * NIW + 2 * pp + ((l >> 4) & 1);
#pragma unroll
      for (int r = 0; r < RPS; r++) {
#pragma unroll
        for (int e = 0; e < 2; e++) {
          const uint4v v = *reinterpret_cast<const uint4v*>(b_lds + (t * RPS + r) * 1024 + ((2 * e + (l >> 5)) * 16 + nn) * 16);
          b.q[pp][e][4 * r + 0] = v.x, b.q[pp][e][4 * r + 1] = v.y, b.q[pp][e][4 * r + 2] = v.z, b.q[pp][e][4 * r + 3] = v.w;
        }
        const unsigned char* sp = b_lds + kBCodes + ((r * kG3Tiles + t) * 16 + nn) * SBYTES;
        if constexpr (SBYTES == 16) {
          const uint4v sv = *reinterpret_cast<const uint4v*>(sp);
          b.c[pp][r].s[0] = sv.x, b.c[pp][r].s[1] = sv.y, b.c[pp][r].s[2] = sv.z, b.c[pp][r].s[3] = sv.w;
        } else if constexpr (SBYTES == 8) {
          b.c[pp][r].s[0] = reinterpret_cast<const uint32_t*>(sp)[0];
          b.c[pp][r].s[1] = reinterpret_cast<const uint32_t*>(sp)[1];
        } else if constexpr (SBYTES == 4) {
          b.c[pp][r].s[0] = *reinterpret_cast<const uint32_t*>(sp);
        } else {
          b.c[pp][r].s[0] = *reinterpret_cast<const uint16_t*>(sp);
        }
        if constexpr (ASYM) {
          const unsigned char* zp = b_lds + kBCodes + kBScal + ((r * kG3Tiles + t) * 16 + nn) * SPS;
          if constexpr (SPS == 4)
            b.c[pp][r].z[0] = *reinterpret_cast<const uint32_t*>(zp);
          else if constexpr (SPS == 2)
            b.c[pp][r].z[0] = *reinterpret_cast<const uint16_t*>(zp);
          else
            b.c[pp][r].z[0] = *zp;
        }
      }
    }
  };
  // LDS -> registers: this wave's four column tiles (wn * 4 + ni)
  auto read_b = [&](BRec& b) {
#pragma unroll
    for (int ni = 0; ni < NIW; ni++) {
      const int t = wn * NIW + ni;
#pragma unroll
      for (int r = 0; r < RPS; r++) {
        const uint4v v = *reinterpret_cast<const uint4v*>(b_lds + (t * RPS + r) * 1024 + l * 16);
        b.q[ni][4 * r + 0] = v.x, b.q[ni][4 * r + 1] = v.y, b.q[ni][4 * r + 2] = v.z, b.q[ni][4 * r + 3] = v.w;
        const unsigned char* sp = b_lds + kBCodes + ((r * kG3Tiles + t) * 16 + nn) * SBYTES;
        if constexpr (SBYTES == 16) {
          const uint4v sv = *reinterpret_cast<const uint4v*>(sp);
          b.c[ni][r].s[0] = sv.x, b.c[ni][r].s[1] = sv.y, b.c[ni][r].s[2] = sv.z, b.c[ni][r].s[3] = sv.w;
        } else if constexpr (SBYTES == 8) {
          b.c[ni][r].s[0] = reinterpret_cast<const uint32_t*>(sp)[0];
          b.c[ni][r].s[1] = reinterpret_cast<const uint32_t*>(sp)[1];
        } else if constexpr (SBYTES == 4) {
          b.c[ni][r].s[0] = *reinterpret_cast<const uint32_t*>(sp);
        } else {
          b.c[ni][r].s[0] = *reinterpret_cast<const uint16_t*>(sp);
        }
        if constexpr (ASYM) {
          const unsigned char* zp = b_lds + kBCodes + kBScal + ((r * kG3Tiles + t) * 16 + nn) * SPS;
          if constexpr (SPS == 4)
            b.c[ni][r].z[0] = *reinterpret_cast<const uint32_t*>(zp);
          else if constexpr (SPS == 2)
            b.c[ni][r].z[0] = *reinterpret_cast<const uint16_t*>(zp);
          else
            b.c[ni][r].z[0] = *zp;
        }
      }
    }
  };

  // B fragments of 32-deep slice t (0..3) of the superstep held in `b`: codes -> fp16 (code - zp) * scale
  auto dequant = [&](const BRec& b, auto tc, half8_t (&bf)[NIW]) {
    constexpr int t = decltype(tc)::value;
    constexpr int h = t >> 1, jj = t & 1;
#pragma unroll
    for (int ni = 0; ni < NIW; ni++) {
      // 4-bit: word t of the one record; 8-bit: record h, words 2 jj, 2 jj + 1
      float sc[4], zp[4];
      corr_decode<SPS, SK, ASYM, NJ>(b.c[ni][B8 ? h : 0], sc, zp);
      constexpr int js = B8 ? jj : t;  // slice inside its record
      half8_t v;
      if constexpr (KIND == WK_INT4) {
        const _Float16 zl = (_Float16)(-1032.f - zp[js]), zh = (_Float16)(-72.f - zp[js]);
        v = cvt_i4x8(b.q[ni][js], i4c, half2_t{zl, zl}, half2_t{zh, zh});
      } else if constexpr (KIND == WK_INT8) {
        const _Float16 zo = (_Float16)(-1152.f - zp[js]);
        v = cvt_i8x8(b.q[ni][4 * h + 2 * jj], b.q[ni][4 * h + 2 * jj + 1], half2_t{zo, zo});
      } else {
        v = cvt_f4x8(b.q[ni][js], p.lut);
      }
      const _Float16 sh = (_Float16)(sc[js] * spre_);
      bf[ni] = v * half8_t{sh, sh, sh, sh, sh, sh, sh, sh};
    }
  };
  auto load_af = [&](int stage, int jj, half8_t (&af)[MI]) {
    const unsigned char* a_lds = smem + stage * kStage + a_roff[jj];
#pragma unroll
    for (int mi = 0; mi < MI; mi++) af[mi] = *reinterpret_cast<const half8_t*>(a_lds + mi * (16 * 128));
  };
  // one slice: 32 MFMAs; each A fragment is reloaded for the NEXT slice (stage / jj given) as soon as its four MFMAs are
  // issued — in place, so the next slice's fragments cost no registers beyond this slice's
  // `after(mi)`: hook behind the four MFMAs of fragment row mi — the DMA requests of the next chunk are issued there,
  // one per row, instead of in a burst behind the barrier (a request costs ~60 cycles of issue among MFMAs, 100-185 in a
  // phase that already carries fragment reads, MI355X_MICROARCH.md; 8 of them in front of the MFMAs stall every wave)
  auto mma = [&](half8_t (&af)[MI], const half8_t (&bf)[NIW], auto reload, int stage, int jj, auto&& after) {
    const unsigned char* nxt = smem + stage * kStage + a_roff[jj];
#pragma unroll
    for (int mi = 0; mi < MI; mi++) {
#pragma unroll
      for (int ni = 0; ni < NIW; ni++)
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mi], bf[ni], acc[mi][ni], 0, 0, 0);
      if constexpr (decltype(reload)::value) af[mi] = *reinterpret_cast<const half8_t*>(nxt + mi * (16 * 128));
      after(mi);
    }
  };
  auto nothing = [](int) {};
  // ask the scheduler to spread the preparation of the NEXT slice (8 fragment reads, ~60 VALU of dequantisation) between
  // this slice's 32 MFMAs instead of in front of them: one wave then keeps its matrix pipe busy on its own
  auto interleave = [&](auto vpm, auto ds, auto dma) {
    constexpr int valu_per_mfma = decltype(vpm)::value;
#pragma unroll
    for (int i = 0; i < MI * NIW; i++) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                            // 1 MFMA
      if (decltype(ds)::value && (i & (NIW - 1)) == NIW - 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 LDS read
      if (decltype(dma)::value && (i & (NIW - 1)) == 1 && i / NIW < APW) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // 1 DMA request
      if (valu_per_mfma) __builtin_amdgcn_sched_group_barrier(0x002, valu_per_mfma, 0);             // VALU
    }
  };
  // ---- M32 forms of the same four steps ----
  auto dequant32 = [&](const BRec32& b, auto tc, half8_t (&bf)[NP > 0 ? NP : 1][2]) {
    constexpr int t = decltype(tc)::value;
    constexpr int h = t >> 1, jj = t & 1;
#pragma unroll
    for (int pp = 0; pp < NP; pp++) {
      float sc[4], zp[4];
      corr_decode<SPS, SK, ASYM, NJ>(b.c[pp][B8 ? h : 0], sc, zp);
      constexpr int js = B8 ? jj : t;
      const _Float16 sh = (_Float16)(sc[js] * spre_);
#pragma unroll
      for (int e = 0; e < 2; e++) {
        half8_t v;
        if constexpr (KIND == WK_INT4) {
          const _Float16 zl = (_Float16)(-1032.f - zp[js]), zh = (_Float16)(-72.f - zp[js]);
          v = cvt_i4x8(b.q[pp][e][js], i4c, half2_t{zl, zl}, half2_t{zh, zh});
        } else if constexpr (KIND == WK_INT8) {
          const _Float16 zo = (_Float16)(-1152.f - zp[js]);
          v = cvt_i8x8(b.q[pp][e][4 * h + 2 * jj], b.q[pp][e][4 * h + 2 * jj + 1], half2_t{zo, zo});
        } else {
          v = cvt_f4x8(b.q[pp][e][js], p.lut);
        }
        bf[pp][e] = v * half8_t{sh, sh, sh, sh, sh, sh, sh, sh};
      }
    }
  };
  auto load_af32 = [&](int stage, int jj, half8_t (&af)[MB > 0 ? MB : 1][2]) {
#pragma unroll
    for (int e = 0; e < 2; e++) {
      const unsigned char* a_lds = smem + stage * kStage + a_roff32[2 * jj + e];
#pragma unroll
      for (int mb = 0; mb < MB; mb++) af[mb][e] = *reinterpret_cast<const half8_t*>(a_lds + mb * (32 * 128));
    }
  };
  // one 32-deep slice = two 16-deep operand sets e: (MB x NP) v_mfma_f32_32x32x16_f16 each; operand (mb, e) is reloaded in place
  // for the next slice behind its NP instructions, `after` is called MI times like in the 16x16x32 loop
  auto mma32 = [&](half8_t (&af)[MB > 0 ? MB : 1][2], const half8_t (&bf)[NP > 0 ? NP : 1][2], auto reload, int stage, int jj, auto&& after) {
#pragma unroll
    for (int e = 0; e < 2; e++) {
      const unsigned char* nxt = smem + stage * kStage + a_roff32[2 * jj + e];
#pragma unroll
      for (int mb = 0; mb < MB; mb++) {
#pragma unroll
        for (int pp = 0; pp < NP; pp++)
          acc32[mb][pp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[mb][e], bf[pp][e], acc32[mb][pp], 0, 0, 0);
        if constexpr (decltype(reload)::value) af[mb][e] = *reinterpret_cast<const half8_t*>(nxt + mb * (32 * 128));
        after(e * MB + mb);
      }
    }
  };
  auto interleave32 = [&](auto vpm, auto ds, auto dma) {
    constexpr int valu_per_mfma = decltype(vpm)::value;
#pragma unroll
    for (int i = 0; i < 2 * MB * NP; i++) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                   // 1 MFMA (twice the work of a 16x16x32)
      if (decltype(ds)::value && (i % NP) == NP - 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);    // 1 LDS read
      if (decltype(dma)::value && (i % NP) == 0 && i / NP < APW) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // 1 DMA request
      if (valu_per_mfma) __builtin_amdgcn_sched_group_barrier(0x002, 2 * valu_per_mfma, 0);                // VALU
    }
  };
  using T_ = std::true_type;
  using F_ = std::false_type;

  // chunk range of this workgroup (split-K: an even number of chunks per split, so supersteps never straddle)
  const int cbeg = p.ksplit > 1 ? int(blockIdx.y) * p.cps : 0;
  const int cend = p.ksplit > 1 ? min(p.nchunks, cbeg + p.cps) : p.nchunks;
  const int ubeg = cbeg >> 1, uend = (cend + 1) >> 1;

  if constexpr (M32) {
    BRec32 breg;
    half8_t af[MB > 0 ? MB : 1][2], bf0[NP > 0 ? NP : 1][2], bf1[NP > 0 ? NP : 1][2];
    issue_a(cbeg, 0);
    issue_b(ubeg);
    for (int u = ubeg; u < (p.diag == 2 ? ubeg + 1 : uend); u++) {
      const int c0 = 2 * u;
      // ---- chunk 2u (A stage 0): this superstep's records move to registers ----
      // __syncthreads() = s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier here (hipcc drains the DMAs in flight before a
      // barrier): the wave's own DMAs — A(2u), B(u) — have landed, then everybody's have, and everybody is done with A
      // stage 1.
      __syncthreads();
      read_b32(breg);
      load_af32(0, 0, af);
      dequant32(breg, std::integral_constant<int, 0>{}, bf0);
      __builtin_amdgcn_sched_barrier(0);
      // slice 0 multiplies while slice 1 is prepared and the next chunk's A image is requested piece by piece
      // (requested unconditionally: past the last chunk the pieces land in a stage nobody reads — the source offsets stay
      // inside the buffer descriptor or read as zeros — and the loop body stays one straight-line block to schedule)
      const bool more1 = c0 + 1 < cend;
      dequant32(breg, std::integral_constant<int, 1>{}, bf1);
      mma32(af, bf0, T_{}, 0, 1, [&](int mi) {
        if (mi < APW) issue_a_piece(c0 + 1, 1, mi);
      });
      interleave32(std::integral_constant<int, 2>{}, T_{}, T_{});
      __builtin_amdgcn_sched_barrier(0);
      // slice 1 multiplies while slice 2's B fragments are prepared (its A image is behind the next barrier)
      dequant32(breg, std::integral_constant<int, 2>{}, bf0);
      mma32(af, bf1, F_{}, 0, 0, nothing);
      interleave32(std::integral_constant<int, 2>{}, F_{}, F_{});
      __builtin_amdgcn_sched_barrier(0);
      // ---- chunk 2u + 1 (A stage 1) ----
      if (more1) {
        __syncthreads();  // A(2u + 1) has landed; everybody has copied B(u) and left A stage 0
        if (u + 1 < uend) issue_b(u + 1);
        load_af32(1, 0, af);
        __builtin_amdgcn_sched_barrier(0);
        dequant32(breg, std::integral_constant<int, 3>{}, bf1);
        mma32(af, bf0, T_{}, 1, 1, [&](int mi) {
          if (mi < APW) issue_a_piece(c0 + 2, 0, mi);
        });
        interleave32(std::integral_constant<int, 2>{}, T_{}, T_{});
        __builtin_amdgcn_sched_barrier(0);
        mma32(af, bf1, F_{}, 0, 0, nothing);
      }
    }
  } else {
    BRec breg;
    half8_t af[MI], bf0[NIW], bf1[NIW];
    issue_a(cbeg, 0);
    issue_b(ubeg);
    for (int u = ubeg; u < (p.diag == 2 ? ubeg + 1 : uend); u++) {
      const int c0 = 2 * u;
      // ---- chunk 2u (A stage 0): this superstep's records move to registers ----
      // __syncthreads() = s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier here (hipcc drains the DMAs in flight before a
      // barrier): the wave's own DMAs — A(2u), B(u) — have landed, then everybody's have, and everybody is done with A
      // stage 1.
      __syncthreads();
      read_b(breg);
      load_af(0, 0, af);
      dequant(breg, std::integral_constant<int, 0>{}, bf0);
      __builtin_amdgcn_sched_barrier(0);
      // slice 0 multiplies while slice 1 is prepared and the next chunk's A image is requested piece by piece
      // (requested unconditionally: past the last chunk the pieces land in a stage nobody reads — the source offsets stay
      // inside the buffer descriptor or read as zeros — and the loop body stays one straight-line block to schedule)
      const bool more1 = c0 + 1 < cend;
      dequant(breg, std::integral_constant<int, 1>{}, bf1);
      mma(af, bf0, T_{}, 0, 1, [&](int mi) {
        if (mi < APW) issue_a_piece(c0 + 1, 1, mi);
      });
      interleave(std::integral_constant<int, 2>{}, T_{}, T_{});
      __builtin_amdgcn_sched_barrier(0);
      // slice 1 multiplies while slice 2's B fragments are prepared (its A image is behind the next barrier)
      dequant(breg, std::integral_constant<int, 2>{}, bf0);
      mma(af, bf1, F_{}, 0, 0, nothing);
      interleave(std::integral_constant<int, 2>{}, F_{}, F_{});
      __builtin_amdgcn_sched_barrier(0);
      // ---- chunk 2u + 1 (A stage 1) ----
      if (more1) {
        __syncthreads();  // A(2u + 1) has landed; everybody has copied B(u) and left A stage 0
        if (u + 1 < uend) issue_b(u + 1);
        load_af(1, 0, af);
        __builtin_amdgcn_sched_barrier(0);
        dequant(breg, std::integral_constant<int, 3>{}, bf1);
        mma(af, bf0, T_{}, 1, 1, [&](int mi) {
          if (mi < APW) issue_a_piece(c0 + 2, 0, mi);
        });
        interleave(std::integral_constant<int, 2>{}, T_{}, T_{});
        __builtin_amdgcn_sched_barrier(0);
        mma(af, bf1, F_{}, 0, 0, nothing);
      }
    }
  }

  // ---- epilogue: through LDS, so that C leaves in whole rows.  A lane of the MFMA layout holds column nn of rows
  //      4g .. 4g + 3: storing from there writes 64-byte (fp32) and 32-byte (fp16 shadow) pieces — 50 MB per 2048 x 4096
  //      output at a fraction of the HBM write rate, a third of the whole GEMM's time at K = 4096.  Each wave parks half
  //      of its 128 x 64 tile (64 rows, padded to 68 floats: conflict-free ds_write_b32) in the stage memory, reads it
  //      back as float4 along the rows and stores 256-byte runs; the operator is applied on the way out, in a plain loop
  //      (nothing below indexes the accumulators dynamically — with the switch inside the unrolled accumulator loops hipcc
  //      kept all 128 accumulator registers in scratch memory, a store behind every MFMA of the main loop). ----
  const int epi = p.epilogue;
  auto finish = [&](float v, float dv) {
    switch (epi) {
      case 1: return v + dv;            // custom::epilogue::Add
      case 2: return v * dv;            // custom::epilogue::Mul
      case 3: return epi_gelu(v + dv);  // custom::epilogue::Add_Gelu
      case 4: return epi_gelu(v);
      case 5: return epi_silu(v);
      default: return v;
    }
  };
  // ---- the cross-wave form of that epilogue (round 5; waves 1 x 4).  Measured on the 7B shapes at 2048 rows (profiles/r05a_prefill_probe.txt):
  //      with the stores skipped a 11008-wide GEMM takes 184 us, with the fp32 stores 187, with fp32 + the fp16 shadow 205 — the shadow left
  //      each wave as 8-byte stores in 64-byte runs (half a memory line).  Here the four waves park their 64-row halves side by side
  //      (row = the workgroup tile's whole width), each wave then owns 16 of the 64 rows: fp32 leaves as float4 per lane in 512-byte
  //      runs (256 for gate/up pairs), the fp16 shadow in a second read-back as 16-byte stores (8 columns per lane) in 256-byte runs
  //      (128).  DUAL: gate and up of one (row, column) sit in the same lane, the product is formed before parking. ----
  if constexpr (!SQ && !M32) {
    if (p.wide && p.ksplit <= 1) {
      constexpr int PC = NIW * 16;               // parked columns per wave (DUAL: 16 gate + 16 up)
      constexpr int WROW = 4 * PC + 4;           // floats per parked row (ds_write_b32 of the MFMA layout stays conflict-free)
      constexpr int OC = DUAL ? 16 : PC;         // output columns per wave
      constexpr int W = 4 * OC;                  // output columns of the workgroup tile
      constexpr int LA = W / 4, RA = 64 / LA, IA = 16 / RA;  // fp32 pass: lanes per row, rows per wave instruction, iterations
      constexpr int LB = W / 8, RB = 64 / LB, IB = 16 / RB;  // fp16 pass
      float* parkw = reinterpret_cast<float*>(smem);
      const int colb = bnl * W;  // first output column of the tile inside its matrix
      const bool full = colb + W <= n_cols;
      const bool al = (p.ldc & 7) == 0 && (!c_out || (reinterpret_cast<uintptr_t>(c_out) & 15) == 0) &&
                      (!c16_out || (reinterpret_cast<uintptr_t>(c16_out) & 15) == 0) &&
                      (!p.d || ((p.ldd & 3) == 0 && (reinterpret_cast<uintptr_t>(p.d) & 15) == 0)) &&
                      (!DUAL || !p.c2 || (reinterpret_cast<uintptr_t>(p.c2) & 15) == 0);
      const bool use_d = !DUAL && p.d && epi >= 1 && epi <= 3;
      typedef _Float16 half8o_t __attribute__((ext_vector_type(8)));
      // parked column of output column oc: the wave that multiplied it, then its place among that wave's columns
      auto pcol = [&](int oc) { return DUAL ? (oc >> 4) * PC + (oc & 15) : oc; };
      // one float4 of finished outputs at (parked row pr, output columns oc .. oc + 3); DUAL: act(gate) * up, gate4 = act(gate)
      auto out4 = [&](int pr, int oc, int row, float4* gate4) {
        const float* q = parkw + pr * WROW + pcol(oc);
        float4 v = *reinterpret_cast<const float4*>(q);
        if constexpr (DUAL) {
          const float4 u = *reinterpret_cast<const float4*>(q + 16);
          v = float4{finish(v.x, 0.f), finish(v.y, 0.f), finish(v.z, 0.f), finish(v.w, 0.f)};
          if (gate4) *gate4 = v;
          v = float4{v.x * u.x, v.y * u.y, v.z * u.z, v.w * u.w};
        } else {
          float4 dv = {0.f, 0.f, 0.f, 0.f};
          if (use_d) dv = *reinterpret_cast<const float4*>(p.d + size_t(row) * p.ldd + colb + oc);
          v = float4{finish(v.x, dv.x), finish(v.y, dv.y), finish(v.z, dv.z), finish(v.w, dv.w)};
        }
        return v;
      };
#pragma unroll
      for (int hh = 0; hh < MI / 4; hh++) {
        __syncthreads();  // every wave is done with the A stages / with the previous read-back
#pragma unroll
        for (int mi = 0; mi < 4; mi++)
#pragma unroll
          for (int ni = 0; ni < NIW; ni++)
#pragma unroll
            for (int r = 0; r < 4; r++)
              parkw[(mi * 16 + 4 * g + r) * WROW + w * PC + ni * 16 + nn] = acc[4 * hh + mi][ni][r] * spost_;
        __syncthreads();
        if (p.diag == 1) continue;
        const int rb = row0 + hh * 64 + w * 16;  // this wave's 16 rows of the half
        if (full && al) {
          if (c_out || (DUAL && p.c2)) {
            for (int it = 0; it < IA; it++) {
              const int rl = it * RA + l / LA, row = rb + rl, oc = (l % LA) * 4;
              if (row >= p.m) continue;
              float4 gate4;
              const float4 v = out4(w * 16 + rl, oc, row, &gate4);
              if (c_out) *reinterpret_cast<float4*>(c_out + size_t(row) * p.ldc + colb + oc) = v;
              if constexpr (DUAL) {
                if (p.c2) *reinterpret_cast<float4*>(p.c2 + size_t(row) * p.ldc + colb + oc) = gate4;
              }
            }
          }
          if (c16_out) {
            for (int it = 0; it < IB; it++) {
              const int rl = it * RB + l / LB, row = rb + rl, oc = (l % LB) * 8;
              if (row >= p.m) continue;
              const float4 v0 = out4(w * 16 + rl, oc, row, nullptr), v1 = out4(w * 16 + rl, oc + 4, row, nullptr);
              *reinterpret_cast<half8o_t*>(c16_out + size_t(row) * p.ldc + colb + oc) =
                  half8o_t{(_Float16)v0.x, (_Float16)v0.y, (_Float16)v0.z, (_Float16)v0.w, (_Float16)v1.x, (_Float16)v1.y, (_Float16)v1.z, (_Float16)v1.w};
            }
          }
        } else {  // ragged right edge or unaligned outputs: element by element, row-major over the wave's 16 x W part
          for (int e = l; e < 16 * W; e += 64) {
            const int rl = e / W, oc = e % W, row = rb + rl, col = colb + oc;
            if (row >= p.m || col >= n_cols) continue;
            const float* q = parkw + (w * 16 + rl) * WROW + pcol(oc);
            float v = q[0];
            if constexpr (DUAL) {
              v = finish(v, 0.f);
              if (p.c2) p.c2[size_t(row) * p.ldc + col] = v;
              v *= q[16];
            } else {
              v = finish(v, use_d ? p.d[size_t(row) * p.ldd + col] : 0.f);
            }
            if (c_out) c_out[size_t(row) * p.ldc + col] = v;
            if (c16_out) c16_out[size_t(row) * p.ldc + col] = (_Float16)v;
          }
        }
      }
      return;
    }
  }
  if constexpr (DUAL) return;  // (gate/up launches always take the cross-wave epilogue)
  constexpr int kCols = NIW * 16;     // columns of the wave tile
  constexpr int kRowF = kCols + 4;    // floats per parked row
  constexpr int kLpr = kCols / 4;     // lanes per row in the float4 read-back
  float* park = reinterpret_cast<float*>(smem) + w * (64 * kRowF);
  const int colw = tile0 * 16;  // first column of this wave's 64
  const bool vec4 = (p.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(c_out) & 15) == 0 && colw + kCols <= n_cols &&
                    (p.ksplit > 1 ? (p.n & 3) == 0 && (reinterpret_cast<uintptr_t>(p.part) & 15) == 0 : true) &&
                    (!p.d || ((p.ldd & 3) == 0 && (reinterpret_cast<uintptr_t>(p.d) & 15) == 0)) &&
                    (!c16_out || (reinterpret_cast<uintptr_t>(c16_out) & 7) == 0);
  __syncthreads();  // every wave is done with the A stages
#pragma unroll
  for (int hh = 0; hh < MI / 4; hh++) {
    if constexpr (M32) {  // C of v_mfma_f32_32x32x16: lane (column l & 31, half l >> 5), register r -> row (r & 3) + 8 (r >> 2) + 4 half
#pragma unroll
      for (int mbl = 0; mbl < 2; mbl++)
#pragma unroll
        for (int pp = 0; pp < NP; pp++)
#pragma unroll
          for (int r = 0; r < 16; r++)
            park[(mbl * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * kRowF + pp * 32 + (l & 31)] = acc32[2 * hh + mbl][pp][r] * spost_;
    } else {
#pragma unroll
      for (int mi = 0; mi < 4; mi++)
#pragma unroll
        for (int ni = 0; ni < NIW; ni++)
#pragma unroll
          for (int r = 0; r < 4; r++) park[(mi * 16 + 4 * g + r) * kRowF + ni * 16 + nn] = acc[4 * hh + mi][ni][r] * spost_;
    }
    // (LDS operations of one wave complete in order: no barrier between its own writes and reads)
    const int rbase = row0 + wm * 128 + hh * 64;
    if (p.diag == 1) continue;
    if (vec4) {
      for (int it = 0; it < kLpr; it++) {
        const int rl = it * (64 / kLpr) + l / kLpr, row = rbase + rl, col = colw + (l % kLpr) * 4;
        if (row >= p.m) continue;
        float4 v = *reinterpret_cast<const float4*>(park + rl * kRowF + (l % kLpr) * 4);
        if (p.ksplit > 1) {  // raw partial; the operator is applied by gemm2_reduce_kernel
          *reinterpret_cast<float4*>(p.part + (size_t(blockIdx.y) * p.m + row) * p.n + col) = v;
          continue;
        }
        if (p.rope_on) {  // fused QKV + RoPE + cache append: rope_qkv_append_tab_kernel's arithmetic on the four columns (two adjacent pairs) of this lane
          typedef _Float16 half4r_t __attribute__((ext_vector_type(4)));
          const int hd = col / p.rope_hs, dcol = col - hd * p.rope_hs;
          if (sg < 2) {
#pragma clang fp contract(off)  // (rope_kernel's products and sums are rounded one by one: ns_quant.hip is built with -ffp-contract=off, this file is not)
            const float4 cs = *reinterpret_cast<const float4*>(p.rope_tab + (size_t(row) * size_t(p.rope_hs >> 1) + size_t(dcol >> 1)));
            const float x0 = v.x, x1 = v.y, x2 = v.z, x3 = v.w;
            // (plain operators: hipcc's __fmul_rn / __fadd_rn are inline x * y / x + y carrying their header's contraction mode, which fuses them here)
            const float p0 = x0 * cs.x, p1 = x1 * cs.y, p2 = x0 * cs.y, p3 = x1 * cs.x, p4 = x2 * cs.z, p5 = x3 * cs.w, p6 = x2 * cs.w, p7 = x3 * cs.z;
            v.x = p0 - p1, v.y = p2 + p3, v.z = p4 - p5, v.w = p6 + p7;
          }
          if (sg > 0) {
            _Float16* cell = (sg == 1 ? p.rope_kc : p.rope_vc) + (long long)(p.rope_npast + row) * p.rope_csl + (long long)hd * p.rope_chead + dcol;
            *reinterpret_cast<half4r_t*>(cell) = half4r_t{(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
          }
          if (c_out) *reinterpret_cast<float4*>(c_out + size_t(row) * p.ldc + col) = v;
          continue;
        }
        float4 dv = {0.f, 0.f, 0.f, 0.f};
        if (p.d && epi >= 1 && epi <= 3) dv = *reinterpret_cast<const float4*>(p.d + size_t(row) * p.ldd + col);
        v = float4{finish(v.x, dv.x), finish(v.y, dv.y), finish(v.z, dv.z), finish(v.w, dv.w)};
        if (c_out) *reinterpret_cast<float4*>(c_out + size_t(row) * p.ldc + col) = v;
        if (c16_out) {
          typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
          *reinterpret_cast<half4_t*>(c16_out + size_t(row) * p.ldc + col) =
              half4_t{(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
        }
      }
    } else {
      for (int it = 0; it < 64 * kCols / 64; it++) {  // 64 elements per step, row-major over the 64 x kCols half tile
        const int e = it * 64 + l, rl = e / kCols, cl = e % kCols;
        const int row = rbase + rl, col = colw + cl;
        if (row >= p.m || col >= n_cols) continue;
        float v = park[rl * kRowF + cl];
        if (p.ksplit > 1) {
          p.part[(size_t(blockIdx.y) * p.m + row) * p.n + col] = v;
          continue;
        }
        const float dv = (p.d && epi >= 1 && epi <= 3) ? p.d[size_t(row) * p.ldd + col] : 0.f;
        v = finish(v, dv);
        if (c_out) c_out[size_t(row) * p.ldc + col] = v;
        if (c16_out) c16_out[size_t(row) * p.ldc + col] = (_Float16)v;
      }
    }
  }
}

#ifdef NS_WITH_GEMM3D  // measured slower than gemm3_kernel on every int4 shape (profiles/r02q_gemm3_vs_gemm3d.txt): not built by default
// ================================================================================================================
// gemm3d_kernel — gemm3_kernel with a DEEP pipeline, for one workgroup per CU.  gemm3_kernel relies on a second
// workgroup on the CU to cover its memory latency (each chunk's DMA is issued one chunk ahead): with 256 output tiles
// on 256 CUs (every 4096-wide GEMM at 2048 rows) a chunk took 1.4 us, 30 % matrix-core utilisation, and splitting K to
// get the second workgroup costs a reduction pass as long as a third of the GEMM (profiles/r02q).  Here
//   * eight waves: the four 128 x 64 wave tiles twice, wave group wk multiplying 32-deep slice wk of every 64-deep
//     chunk (two waves per SIMD from ONE workgroup; the halves are added through LDS at the end);
//   * four A stages (three for 8-bit codes, whose B stages are twice as large): the DMA of chunk c + 3 is issued while
//     chunk c is multiplied; two B stages, a superstep's records arrive a chunk before they are needed;
//   * counted waits: s_waitcnt vmcnt(n) leaves the younger chunks' DMAs in flight across the barrier (raw s_barrier —
//     __syncthreads() would drain them).  Per wave a chunk's batch is 4 A requests, an odd chunk's batch also the NB
//     requests of the next superstep's B records; requests retire in order, so before chunk c only the batches younger
//     than the one that carried A(c) — and, before an odd chunk, B(u + 1) — may still be in flight.
constexpr int kG3dThreads = 512;
template <int KIND, int SPS, int SK, bool ASYM>
__global__ __launch_bounds__(kG3dThreads, 1) void gemm3d_kernel(const Gemm2Params p) {
  constexpr bool B8 = KIND == WK_INT8;
  constexpr int NJ = B8 ? 2 : 4;
  constexpr int RPS = B8 ? 2 : 1;
  constexpr int SBYTES = SPS * (SK == SK_F32 ? 4 : 2);
  constexpr int NS = B8 ? 3 : 4;   // A stages
  constexpr int D = NS - 1;        // chunks of A in flight ahead of the one being multiplied
  constexpr int NB = RPS * (2 + (ASYM ? 1 : 0));  // B requests per wave and superstep
  using Corr = CorrRaw<SPS, SK, ASYM>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __attribute__((address_space(3))) unsigned char* LdsPtr;

  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = tid & 63, nn = l & 15, g = l >> 4;
  const int wk = w >> 2, wm = (w >> 1) & 1, wn = w & 1;
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  const int bn = xcd * p.cpx + local % p.cpx, bm = local / p.cpx;
  if (bn >= p.nbn) return;
  const int tile0 = bn * kG3Tiles + wn * 4, row0 = bm * kG3BM;

  const Rsrc rq = make_rsrc(p.codes, p.codes_bytes);
  const Rsrc rs = make_rsrc(p.scales, p.scales_bytes);
  const Rsrc rz = make_rsrc(p.zps, p.zps_bytes);
  const Rsrc ra = make_rsrc(p.a16, uint32_t(p.m) * uint32_t(p.lda16) * 2u);  // rows >= m read as zeros
  const I4Consts i4c = {0x000f000fu, 0x00f000f0u, 0x64006400u};

  floatx4 acc[8][4];
#pragma unroll
  for (int mi = 0; mi < 8; mi++)
#pragma unroll
    for (int ni = 0; ni < 4; ni++) acc[mi][ni] = floatx4{0.f, 0.f, 0.f, 0.f};

  // ---- A: wave w, request i (0..3) covers rows (4 w + i) * 8 .. + 7 of the 256; same XOR image as gemm3_kernel ----
  const uint32_t lrow = uint32_t(l >> 3);
  uint32_t a_voff[2];
#pragma unroll
  for (int par = 0; par < 2; par++)
    a_voff[par] = (uint32_t(row0) + uint32_t(w) * 32u + lrow) * uint32_t(p.lda16) * 2u +
                  ((uint32_t(l & 7) ^ (uint32_t(4 * par) + uint32_t(l >> 4))) << 4);
  const uint32_t a_istride = 8u * uint32_t(p.lda16) * 2u;
  auto issue_a = [&](int c, int stage) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const LdsPtr dst = (LdsPtr)(smem) + stage * kG3StageBytes + (w * 4 + i) * 1024;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, reinterpret_cast<__attribute__((address_space(3))) void*>(dst), 16,
                                               a_voff[i & 1], uint32_t(c) * (kG3KC * 2) + uint32_t(i) * a_istride, 0, 0);
    }
#endif
  };
  // this wave multiplies slice wk of a chunk: fragment of row wm * 128 + mi * 16 + nn, piece (4 wk + g) ^ ((nn >> 1) & 7)
  const uint32_t a_roff = uint32_t(wm * 128 + nn) * 128u + ((uint32_t(4 * wk + g) ^ uint32_t((nn >> 1) & 7)) << 4);

  // ---- B: two stages of {records, scale rows, zero-point rows} of the 8 column tiles; wave w fetches tile w ----
  constexpr uint32_t kBCodes = kG3Tiles * RPS * 1024u;
  constexpr uint32_t kBScal = kG3Tiles * RPS * 16u * SBYTES;
  constexpr uint32_t kBZp = ASYM ? kG3Tiles * RPS * 16u * SPS : 0u;
  constexpr uint32_t kBStage = kBCodes + kBScal + kBZp;
  unsigned char* const b_lds0 = smem + NS * kG3StageBytes;
  const uint32_t btile = uint32_t(bn * kG3Tiles + w);
  const uint32_t s_voff = btile * uint32_t(p.srows) * p.sstride + uint32_t(l) * 16u;  // lanes < SBYTES
  const uint32_t z_voff = btile * uint32_t(p.srows) * p.zstride + uint32_t(l) * 16u;  // lanes < SPS
  auto issue_b = [&](int u) {
#if defined(__HIP_DEVICE_COMPILE__)
    const LdsPtr bs = (LdsPtr)(b_lds0) + (u & 1) * kBStage;
#pragma unroll
    for (int r = 0; r < RPS; r++) {
      const uint32_t s = uint32_t(min(u * RPS + r, p.ksteps - 1));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, reinterpret_cast<__attribute__((address_space(3))) void*>(bs + (w * RPS + r) * 1024), 16,
                                               uint32_t(l) * 16u, (btile * uint32_t(p.ksteps) + s) * p.qstride, 0, 0);
      const uint32_t srow = (s * uint32_t(p.srow_mul)) >> p.srow_shift;
      if (l < SBYTES)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, reinterpret_cast<__attribute__((address_space(3))) void*>(bs + kBCodes + (r * kG3Tiles + w) * (16 * SBYTES)),
                                                 16, s_voff, srow * p.sstride, 0, 0);
      if constexpr (ASYM) {
        if (l < SPS)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rz, reinterpret_cast<__attribute__((address_space(3))) void*>(bs + kBCodes + kBScal + (r * kG3Tiles + w) * (16 * SPS)),
                                                   16, z_voff, srow * p.zstride, 0, 0);
      }
    }
#endif
  };
  // a wave keeps, per column tile, only the words of ITS two slices of the superstep (t = wk and 2 + wk)
  struct BRec {
    uint32_t q[4][B8 ? 4 : 2];  // 4-bit: words wk, 2 + wk of the record; 8-bit: words 2 wk, 2 wk + 1 of records 0 and 1
    Corr c[4][RPS];
  };
  auto read_b = [&](int u, BRec& b) {
    const unsigned char* bs = b_lds0 + (u & 1) * kBStage;
#pragma unroll
    for (int ni = 0; ni < 4; ni++) {
      const int t = wn * 4 + ni;
#pragma unroll
      for (int r = 0; r < RPS; r++) {
        const unsigned char* rec = bs + (t * RPS + r) * 1024 + l * 16;
        if constexpr (B8) {
          b.q[ni][2 * r + 0] = reinterpret_cast<const uint32_t*>(rec)[2 * wk];
          b.q[ni][2 * r + 1] = reinterpret_cast<const uint32_t*>(rec)[2 * wk + 1];
        } else {
          b.q[ni][0] = reinterpret_cast<const uint32_t*>(rec)[wk];
          b.q[ni][1] = reinterpret_cast<const uint32_t*>(rec)[2 + wk];
        }
        const unsigned char* sp = bs + kBCodes + ((r * kG3Tiles + t) * 16 + nn) * SBYTES;
        if constexpr (SBYTES == 16) {
          const uint4v sv = *reinterpret_cast<const uint4v*>(sp);
          b.c[ni][r].s[0] = sv.x, b.c[ni][r].s[1] = sv.y, b.c[ni][r].s[2] = sv.z, b.c[ni][r].s[3] = sv.w;
        } else if constexpr (SBYTES == 8) {
          b.c[ni][r].s[0] = reinterpret_cast<const uint32_t*>(sp)[0];
          b.c[ni][r].s[1] = reinterpret_cast<const uint32_t*>(sp)[1];
        } else if constexpr (SBYTES == 4) {
          b.c[ni][r].s[0] = *reinterpret_cast<const uint32_t*>(sp);
        } else {
          b.c[ni][r].s[0] = *reinterpret_cast<const uint16_t*>(sp);
        }
        if constexpr (ASYM) {
          const unsigned char* zp = bs + kBCodes + kBScal + ((r * kG3Tiles + t) * 16 + nn) * SPS;
          if constexpr (SPS == 4)
            b.c[ni][r].z[0] = *reinterpret_cast<const uint32_t*>(zp);
          else if constexpr (SPS == 2)
            b.c[ni][r].z[0] = *reinterpret_cast<const uint16_t*>(zp);
          else
            b.c[ni][r].z[0] = *zp;
        }
      }
    }
  };
  // B fragments of this wave's slice of chunk h (0 / 1) of the superstep in `b`
  auto dequant = [&](const BRec& b, auto hc, half8_t (&bf)[4]) {
    constexpr int h = decltype(hc)::value;
#pragma unroll
    for (int ni = 0; ni < 4; ni++) {
      float sc[4], zp[4];
      corr_decode<SPS, SK, ASYM, NJ>(b.c[ni][B8 ? h : 0], sc, zp);
      // slice inside its record: 4-bit t = 2 h + wk; 8-bit wk.  wk is wave-uniform but not a compile-time constant:
      // select with it instead of indexing
      const float s_sel = B8 ? (wk ? sc[1] : sc[0]) : (wk ? sc[2 * h + 1] : sc[2 * h]);
      const float z_sel = B8 ? (wk ? zp[1] : zp[0]) : (wk ? zp[2 * h + 1] : zp[2 * h]);
      half8_t v;
      if constexpr (KIND == WK_INT4) {
        const _Float16 zl = (_Float16)(-1032.f - z_sel), zh = (_Float16)(-72.f - z_sel);
        v = cvt_i4x8(b.q[ni][h], i4c, half2_t{zl, zl}, half2_t{zh, zh});
      } else if constexpr (KIND == WK_INT8) {
        const _Float16 zo = (_Float16)(-1152.f - z_sel);
        v = cvt_i8x8(b.q[ni][2 * h], b.q[ni][2 * h + 1], half2_t{zo, zo});
      } else {
        v = cvt_f4x8(b.q[ni][h], p.lut);
      }
      const _Float16 sh = (_Float16)(s_sel * p.spre);
      bf[ni] = v * half8_t{sh, sh, sh, sh, sh, sh, sh, sh};
    }
  };
  auto load_af = [&](int stage, half8_t (&af)[8]) {
    const unsigned char* a_lds = smem + stage * kG3StageBytes + a_roff;
#pragma unroll
    for (int mi = 0; mi < 8; mi++) af[mi] = *reinterpret_cast<const half8_t*>(a_lds + mi * (16 * 128));
  };
  auto mma = [&](const half8_t (&af)[8], const half8_t (&bf)[4]) {
#pragma unroll
    for (int mi = 0; mi < 8; mi++)
#pragma unroll
      for (int ni = 0; ni < 4; ni++)
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mi], bf[ni], acc[mi][ni], 0, 0, 0);
  };
  auto wait_vm = [&](auto nc) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(decltype(nc)::value) : "memory"); };
  using IC0 = std::integral_constant<int, 0>;
  using IC1 = std::integral_constant<int, 1>;

  // chunk range (split-K: an even number of chunks per split)
  const int cbeg = p.ksplit > 1 ? int(blockIdx.y) * p.cps : 0;
  const int cend = p.ksplit > 1 ? min(p.nchunks, cbeg + p.cps) : p.nchunks;
  const int ubeg = cbeg >> 1, uend = (cend + 1) >> 1;

  // ---- prologue: B(first), A(first D chunks); then the first superstep's B is read after a full drain ----
  issue_b(ubeg);
#pragma unroll
  for (int d = 0; d < D; d++)
    if (cbeg + d < cend) issue_a(cbeg + d, d % NS);
  BRec bcur, bnxt;
  half8_t af[8], bfa[4], bfb[4];
  wait_vm(IC0{});
  asm volatile("s_barrier" ::: "memory");
  read_b(ubeg, bcur);
  if (ubeg + 1 < uend) issue_b(ubeg + 1);  // in the steady state B(u + 1) is issued at the top of chunk 2u - 1
  dequant(bcur, IC0{}, bfa);
  if (p.diag >= 5) {  // diagnostics: operands that no longer come from LDS / the dequantiser
    dequant(bcur, IC1{}, bfb);
    load_af(0, af);
  }
  // The B fragments of a chunk are always prepared during the MFMAs of the chunk before it (bfa: even chunks, bfb: odd
  // chunks), so that behind a barrier only the A fragment reads stand in front of the matrix cores; the scheduler is
  // asked to put the fragment reads first and then to alternate MFMAs with the dequantisation's VALU work.
  auto interleave = [&]() {
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);  // the 8 A fragment reads
#pragma unroll
    for (int i = 0; i < 32; i++) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);  // up to 3 VALU
    }
  };
  // from here on, at the top of chunk c the batches in flight are those issued at the tops of chunks c - 1, c - 2, ...
  for (int u = ubeg; u < uend; u++) {
    const int c0 = 2 * u;
    // ================= even chunk c0 =================
    {
      const int c = c0;
      if (c > cbeg) {
        // A(c) came with the batch of chunk c - D; younger batches: chunk c - 1 (odd: 4 + NB) [and c - 2 (even: 4) when D = 3]
        if (c + D + 2 >= cend) wait_vm(IC0{});  // the last chunks: the younger batches are no longer full ones
        else if constexpr (D == 3) wait_vm(std::integral_constant<int, 8 + NB>{});
        else wait_vm(std::integral_constant<int, 4 + NB>{});
        if (p.diag != 7) asm volatile("s_barrier" ::: "memory");
      }
      if (c + D < cend && p.diag < 4) issue_a(c + D, (c - cbeg + D) % NS);
      __builtin_amdgcn_sched_barrier(0);
      if (p.diag != 3) {
        if (p.diag < 6) load_af((c - cbeg) % NS, af);
        if (p.diag < 5) dequant(bcur, IC1{}, bfb);  // for the odd chunk of this superstep
        mma(af, bfa);
        interleave();
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // ================= odd chunk c0 + 1 =================
    if (c0 + 1 < cend) {
      const int c = c0 + 1;
      // A(c) and B(u + 1) (batch of chunk c - 2) must have landed; younger: the batch of chunk c - 1 (even: 4 requests)
      if (c + D + 2 >= cend) wait_vm(IC0{});
      else wait_vm(std::integral_constant<int, 4>{});
      if (p.diag != 7) asm volatile("s_barrier" ::: "memory");
      if (c + D < cend && p.diag < 4) issue_a(c + D, (c - cbeg + D) % NS);
      if (u + 2 < uend && p.diag < 4) issue_b(u + 2);
      if (u + 1 < uend) read_b(u + 1, bnxt);
      __builtin_amdgcn_sched_barrier(0);
      if (p.diag != 3) {
        if (p.diag < 6) load_af((c - cbeg) % NS, af);
        if (u + 1 < uend && p.diag < 5) dequant(bnxt, IC0{}, bfa);  // for the even chunk of the next superstep
        mma(af, bfb);
        interleave();
      }
      __builtin_amdgcn_sched_barrier(0);
      bcur = bnxt;
    }
  }

  // ---- the two slice halves are added through LDS: wave group 1 parks its accumulators, group 0 adds them ----
  __syncthreads();
  {
    floatx4* park4 = reinterpret_cast<floatx4*>(smem) + (w & 3) * (32 * 64) + l;
    if (wk == 1) {
#pragma unroll
      for (int mi = 0; mi < 8; mi++)
#pragma unroll
        for (int ni = 0; ni < 4; ni++) park4[(mi * 4 + ni) * 64] = acc[mi][ni];
    }
    __syncthreads();
    if (wk == 0) {
#pragma unroll
      for (int mi = 0; mi < 8; mi++)
#pragma unroll
        for (int ni = 0; ni < 4; ni++) acc[mi][ni] += park4[(mi * 4 + ni) * 64];
    }
    __syncthreads();
  }
  if (wk != 0) return;

  // ---- epilogue of wave group 0: through LDS in whole rows (see gemm3_kernel) ----
  constexpr int kRowF = 68;
  float* park = reinterpret_cast<float*>(smem) + w * (64 * kRowF);
  const int colw = tile0 * 16;
  const bool vec4 = (p.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(p.c) & 15) == 0 && colw + 64 <= p.n &&
                    (p.ksplit > 1 ? (p.n & 3) == 0 && (reinterpret_cast<uintptr_t>(p.part) & 15) == 0 : true) &&
                    (!p.d || ((p.ldd & 3) == 0 && (reinterpret_cast<uintptr_t>(p.d) & 15) == 0)) &&
                    (!p.c16 || (reinterpret_cast<uintptr_t>(p.c16) & 7) == 0);
  const int epi = p.epilogue;
  auto finish = [&](float v, float dv) {
    switch (epi) {
      case 1: return v + dv;
      case 2: return v * dv;
      case 3: return epi_gelu(v + dv);
      case 4: return epi_gelu(v);
      case 5: return epi_silu(v);
      default: return v;
    }
  };
#pragma unroll
  for (int hh = 0; hh < 2; hh++) {
#pragma unroll
    for (int mi = 0; mi < 4; mi++)
#pragma unroll
      for (int ni = 0; ni < 4; ni++)
#pragma unroll
        for (int r = 0; r < 4; r++) park[(mi * 16 + 4 * g + r) * kRowF + ni * 16 + nn] = acc[4 * hh + mi][ni][r] * p.spost;
    const int rbase = row0 + wm * 128 + hh * 64;
    if (p.diag == 1) continue;
    if (vec4) {
      for (int it = 0; it < 16; it++) {
        const int rl = it * 4 + (l >> 4), row = rbase + rl, col = colw + (l & 15) * 4;
        if (row >= p.m) continue;
        float4 v = *reinterpret_cast<const float4*>(park + rl * kRowF + (l & 15) * 4);
        if (p.ksplit > 1) {
          *reinterpret_cast<float4*>(p.part + (size_t(blockIdx.y) * p.m + row) * p.n + col) = v;
          continue;
        }
        float4 dv = {0.f, 0.f, 0.f, 0.f};
        if (p.d && epi >= 1 && epi <= 3) dv = *reinterpret_cast<const float4*>(p.d + size_t(row) * p.ldd + col);
        v = float4{finish(v.x, dv.x), finish(v.y, dv.y), finish(v.z, dv.z), finish(v.w, dv.w)};
        *reinterpret_cast<float4*>(p.c + size_t(row) * p.ldc + col) = v;
        if (p.c16) {
          typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
          *reinterpret_cast<half4_t*>(p.c16 + size_t(row) * p.ldc + col) =
              half4_t{(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
        }
      }
    } else {
      for (int it = 0; it < 64; it++) {
        const int row = rbase + it, col = colw + l;
        if (row >= p.m || col >= p.n) continue;
        float v = park[it * kRowF + l];
        if (p.ksplit > 1) {
          p.part[(size_t(blockIdx.y) * p.m + row) * p.n + col] = v;
          continue;
        }
        const float dv = (p.d && epi >= 1 && epi <= 3) ? p.d[size_t(row) * p.ldd + col] : 0.f;
        v = finish(v, dv);
        p.c[size_t(row) * p.ldc + col] = v;
        if (p.c16) p.c16[size_t(row) * p.ldc + col] = (_Float16)v;
      }
    }
  }
}

#endif  // NS_WITH_GEMM3D

// split-K tail: C = epilogue(sum over the K splits, in split order -> deterministic)
__global__ void gemm2_reduce_kernel(const Gemm2Params p) {
  const size_t idx = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (idx >= size_t(p.m) * p.n) return;
  const int row = int(idx / p.n), col = int(idx % p.n);
  float v = 0.f;
  for (int z = 0; z < p.ksplit; z++) v += p.part[(size_t(z) * p.m + row) * p.n + col];
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

// fp32 [m][lda] -> fp16 [m][ld16], columns k..ld16-1 zero.  One thread per 8 outputs: two 16-byte loads, one 16-byte store.
__global__ void cvt_a16_kernel(const float* __restrict__ a, _Float16* __restrict__ out, int m, int k, int lda, int ld16) {
  const size_t idx = (size_t(blockIdx.x) * blockDim.x + threadIdx.x) * 8;
  const size_t total = size_t(m) * ld16;
  if (idx >= total) return;
  const int r = int(idx / ld16), c0 = int(idx % ld16);
  const float* src = a + size_t(r) * lda + c0;
  float f[8];
  if (c0 + 8 <= k && (lda & 3) == 0 && (reinterpret_cast<uintptr_t>(a) & 15) == 0) {
    const float4 v0 = *reinterpret_cast<const float4*>(src), v1 = *reinterpret_cast<const float4*>(src + 4);
    f[0] = v0.x, f[1] = v0.y, f[2] = v0.z, f[3] = v0.w, f[4] = v1.x, f[5] = v1.y, f[6] = v1.z, f[7] = v1.w;
  } else {
#pragma unroll
    for (int e = 0; e < 8; e++) f[e] = c0 + e < k ? src[e] : 0.f;
  }
  half2_t h[4];
#pragma unroll
  for (int e = 0; e < 4; e++) h[e] = half2_t{(_Float16)f[2 * e], (_Float16)f[2 * e + 1]};
  *reinterpret_cast<uint4v*>(out + idx) = uint4v{as_u32(h[0]), as_u32(h[1]), as_u32(h[2]), as_u32(h[3])};
}

hipError_t launch_cvt_a16(const float* a, void* out16, int m, int k, int lda, int ld16, hipStream_t st) {
  const size_t units = size_t(m) * ld16 / 8;
  hipLaunchKernelGGL(cvt_a16_kernel, dim3(unsigned((units + 255) / 256)), dim3(256), 0, st, a, static_cast<_Float16*>(out16), m,
                     k, lda, ld16);
  return hipGetLastError();
}

// Per-(stream, slot) device scratch, grow-only.  Work on one stream is ordered, so one buffer per stream and purpose is
// enough; two streams never share one.  A buffer that was handed out while its stream was CAPTURING is baked into the
// graph being built: such buffers are never freed or moved again (a larger request gets a new buffer, the old one is
// retired until gemm_scratch_release()).  Allocating during a capture needs the thread's capture mode relaxed for the
// duration of the hipMalloc (what PyTorch's caching allocator does as well), otherwise the capture is invalidated.
static std::mutex g_scratch_mutex;
struct ScratchBuf {
  void* p = nullptr;
  size_t bytes = 0;
  bool in_graph = false;
};
static std::map<std::pair<hipStream_t, int>, ScratchBuf> g_scratch;
static std::vector<void*> g_scratch_retired;
static void* stream_scratch_impl(hipStream_t st, size_t bytes, int slot, bool* fresh) {
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  ScratchBuf& e = g_scratch[{st, slot}];
  if (fresh) *fresh = false;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess) return nullptr;
  const bool capturing = cs != hipStreamCaptureStatusNone;
  if (e.bytes >= bytes) {
    e.in_graph |= capturing;
    return e.p;
  }
  if (e.p) {
    if (e.in_graph || capturing) {
      g_scratch_retired.push_back(e.p);  // a captured graph may point at it / no synchronous free inside a capture
    } else {
      hipStreamSynchronize(st);
      hipStreamCaptureMode fmode = hipStreamCaptureModeRelaxed;
      hipThreadExchangeStreamCaptureMode(&fmode);
      hipFree(e.p);
      hipThreadExchangeStreamCaptureMode(&fmode);
    }
    e = ScratchBuf{};
  }
  void* p = nullptr;
  // relaxed for the duration of the allocation: in the default (global) mode a hipMalloc invalidates ANY capture in
  // progress in the process, this stream's or another thread's
  hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
  hipThreadExchangeStreamCaptureMode(&mode);
  const hipError_t err = hipMalloc(&p, bytes);
  hipThreadExchangeStreamCaptureMode(&mode);
  if (err != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  e.p = p;
  e.bytes = bytes;
  e.in_graph = capturing;
  if (fresh) *fresh = true;
  return p;
}
void* stream_scratch(hipStream_t st, size_t bytes, int slot) { return stream_scratch_impl(st, bytes, slot, nullptr); }
// the same, zero-filled when (re)allocated — counters that every user leaves at zero again (ns_gemvs.hip: split-K tickets).
void* stream_scratch_zeroed(hipStream_t st, size_t bytes, int slot) {
  bool fresh = false;
  void* p = stream_scratch_impl(st, bytes, slot, &fresh);
  if (p && fresh) {
    // outside a capture: filled now.  Allocated in the middle of a capture (a caller that never ran eagerly first): the fill
    // becomes a node of that graph in front of the first user — a small memset per replay of that one graph, harmless (every
    // user leaves the counters at zero)
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(st, &cs);
    const hipError_t err = hipMemsetAsync(p, 0, bytes, st);
    if (err != hipSuccess || (cs == hipStreamCaptureStatusNone && hipStreamSynchronize(st) != hipSuccess)) {
      (void)hipGetLastError();
      return nullptr;
    }
  }
  return p;
}
void gemm_scratch_release() {
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  for (auto& kv : g_scratch)
    if (kv.second.p) hipFree(kv.second.p);
  g_scratch.clear();
  for (void* p : g_scratch_retired) hipFree(p);
  g_scratch_retired.clear();
}

template <int KIND, int SPS, int SK>
static hipError_t launch_gemm2_k(const Gemm2Params& p, bool asym, dim3 grid, size_t lds, hipStream_t st) {
  auto go = [&](auto kern) {
    static const hipError_t attr =
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(kG2Stages * kG2StageBytes));
    if (attr != hipSuccess) return attr;
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, p);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && p.ksplit > 1) {
      const size_t total = size_t(p.m) * p.n;
      hipLaunchKernelGGL(gemm2_reduce_kernel, dim3(unsigned((total + 255) / 256)), dim3(256), 0, st, p);
      e = hipGetLastError();
    }
    return e;
  };
  if constexpr (KIND == WK_F4) {
    (void)asym;
    return go(gemm2_kernel<KIND, SPS, SK, false>);
  } else {
    if (asym) return go(gemm2_kernel<KIND, SPS, SK, true>);
    return go(gemm2_kernel<KIND, SPS, SK, false>);
  }
}
template <int KIND, int SPS, int SK, int BM, bool TALL = false, bool DUAL = false>
static hipError_t launch_gemm3_k(const Gemm2Params& p, bool asym, dim3 grid, hipStream_t st) {
  // A stages + the B stage of this format: records, scale rows, zero-point rows of 8 column tiles for one superstep
  constexpr int rps = KIND == WK_INT8 ? 2 : 1;
  constexpr int sbytes = SPS * (SK == SK_F32 ? 4 : 2);
  // (the epilogue parks 64 rows x (wave columns + 4) floats per wave over the stage memory: the 64-row tile's stages alone are
  // smaller than that)
  constexpr size_t park_old = size_t(4) * 64 * (((BM == 256 && !TALL) ? 64 : 32) + 4) * 4;
  // the cross-wave epilogue (waves 1 x 4) parks 64 rows of the whole tile width: 128 columns (64 for gate/up pairs) + 4
  constexpr size_t park_wide = (BM == 256 && !TALL) ? 0 : size_t(64) * (128 + 4) * 4;
  const size_t park_bytes = std::max(park_old, p.wide ? park_wide : size_t(0));
  const size_t lds3 = std::max(size_t(kG3Stages) * BM * kG3KC * 2 + size_t(kG3Tiles) * rps * (1024 + 16 * sbytes + (asym ? 16 * SPS : 0)), park_bytes);
  auto go = [&](auto kern) {
    static const hipError_t attr =
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(kG3Stages * kG3StageBytes + kG3BStageMax));
    if (attr != hipSuccess) return attr;
    hipLaunchKernelGGL(kern, grid, dim3(256), lds3, st, p);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && p.ksplit > 1) {
      const size_t total = size_t(p.m) * p.n;
      hipLaunchKernelGGL(gemm2_reduce_kernel, dim3(unsigned((total + 255) / 256)), dim3(256), 0, st, p);
      e = hipGetLastError();
    }
    return e;
  };
  if constexpr (KIND == WK_F4) {
    (void)asym;
    return go(gemm3_kernel<KIND, SPS, SK, false, BM, TALL, DUAL>);
  } else {
    if (asym) return go(gemm3_kernel<KIND, SPS, SK, true, BM, TALL, DUAL>);
    return go(gemm3_kernel<KIND, SPS, SK, false, BM, TALL, DUAL>);
  }
}
#ifdef NS_WITH_GEMM3D
template <int KIND, int SPS, int SK>
static hipError_t launch_gemm3d_k(const Gemm2Params& p, bool asym, dim3 grid, hipStream_t st) {
  constexpr int rps = KIND == WK_INT8 ? 2 : 1;
  constexpr int ns = KIND == WK_INT8 ? 3 : 4;
  constexpr int sbytes = SPS * (SK == SK_F32 ? 4 : 2);
  const size_t ldsd = size_t(ns) * kG3StageBytes + 2 * size_t(kG3Tiles) * rps * (1024 + 16 * sbytes + (asym ? 16 * SPS : 0));
  auto go = [&](auto kern) {
    static const hipError_t attr =
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (attr != hipSuccess) return attr;
    hipLaunchKernelGGL(kern, grid, dim3(kG3dThreads), ldsd, st, p);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && p.ksplit > 1) {
      const size_t total = size_t(p.m) * p.n;
      hipLaunchKernelGGL(gemm2_reduce_kernel, dim3(unsigned((total + 255) / 256)), dim3(256), 0, st, p);
      e = hipGetLastError();
    }
    return e;
  };
  if constexpr (KIND == WK_F4) {
    (void)asym;
    return go(gemm3d_kernel<KIND, SPS, SK, false>);
  } else {
    if (asym) return go(gemm3d_kernel<KIND, SPS, SK, true>);
    return go(gemm3d_kernel<KIND, SPS, SK, false>);
  }
}
#endif
template <int KIND, int SPS>
static hipError_t launch_gemm3_s(const Gemm2Params& p, uint32_t scale_dt, bool asym, dim3 grid, hipStream_t st, bool deep) {
#ifdef NS_WITH_GEMM3D
  if (deep) {
    if (scale_dt == DT_F32) return launch_gemm3d_k<KIND, SPS, SK_F32>(p, asym, grid, st);
    if (scale_dt == DT_F16) return launch_gemm3d_k<KIND, SPS, SK_F16>(p, asym, grid, st);
    return launch_gemm3d_k<KIND, SPS, SK_BF16>(p, asym, grid, st);
  }
#else
  (void)deep;
#endif
  if (p.dual) {  // gate/up pairs: the 1 x 4 wave tiles only
    if (p.bm3 == 64) {
      if (scale_dt == DT_F32) return launch_gemm3_k<KIND, SPS, SK_F32, 64, false, true>(p, asym, grid, st);
      if (scale_dt == DT_F16) return launch_gemm3_k<KIND, SPS, SK_F16, 64, false, true>(p, asym, grid, st);
      return launch_gemm3_k<KIND, SPS, SK_BF16, 64, false, true>(p, asym, grid, st);
    }
    if (scale_dt == DT_F32) return launch_gemm3_k<KIND, SPS, SK_F32, 128, false, true>(p, asym, grid, st);
    if (scale_dt == DT_F16) return launch_gemm3_k<KIND, SPS, SK_F16, 128, false, true>(p, asym, grid, st);
    return launch_gemm3_k<KIND, SPS, SK_BF16, 128, false, true>(p, asym, grid, st);
  }
  if (p.bm3 == 64) {  // calls of at most 64 rows: a quarter / half of the 128-row tile's MFMA work is padding otherwise
    if (scale_dt == DT_F32) return launch_gemm3_k<KIND, SPS, SK_F32, 64>(p, asym, grid, st);
    if (scale_dt == DT_F16) return launch_gemm3_k<KIND, SPS, SK_F16, 64>(p, asym, grid, st);
    return launch_gemm3_k<KIND, SPS, SK_BF16, 64>(p, asym, grid, st);
  }
  if (p.bm3 == 128) {
    if (scale_dt == DT_F32) return launch_gemm3_k<KIND, SPS, SK_F32, 128>(p, asym, grid, st);
    if (scale_dt == DT_F16) return launch_gemm3_k<KIND, SPS, SK_F16, 128>(p, asym, grid, st);
    return launch_gemm3_k<KIND, SPS, SK_BF16, 128>(p, asym, grid, st);
  }
  if constexpr (KIND == WK_INT4) {
    if (p.tall3) {
      if (scale_dt == DT_F32) return launch_gemm3_k<KIND, SPS, SK_F32, 256, true>(p, asym, grid, st);
      if (scale_dt == DT_F16) return launch_gemm3_k<KIND, SPS, SK_F16, 256, true>(p, asym, grid, st);
      return launch_gemm3_k<KIND, SPS, SK_BF16, 256, true>(p, asym, grid, st);
    }
  }
  if (scale_dt == DT_F32) return launch_gemm3_k<KIND, SPS, SK_F32, 256>(p, asym, grid, st);
  if (scale_dt == DT_F16) return launch_gemm3_k<KIND, SPS, SK_F16, 256>(p, asym, grid, st);
  return launch_gemm3_k<KIND, SPS, SK_BF16, 256>(p, asym, grid, st);
}

template <int KIND, int SPS>
static hipError_t launch_gemm2_s(const Gemm2Params& p, uint32_t scale_dt, bool asym, dim3 grid, size_t lds,
                                 hipStream_t st) {
  if (scale_dt == DT_F32) return launch_gemm2_k<KIND, SPS, SK_F32>(p, asym, grid, lds, st);
  if (scale_dt == DT_F16) return launch_gemm2_k<KIND, SPS, SK_F16>(p, asym, grid, lds, st);
  return launch_gemm2_k<KIND, SPS, SK_BF16>(p, asym, grid, lds, st);
}

static std::atomic<int> g_g3_bm{0};
static std::atomic<int> g_g3_wide{-1};  // ns_hip_set_tuning("g3_wide", 0 / 1): -1 = NS_G3_WIDE or the default (0)
void set_gemm3_wide(int on) { g_g3_wide.store(on < 0 ? -1 : (on != 0)); }
static std::atomic<int> g_g3_min_m{0};  // ns_hip_set_tuning("g3_min_m", rows): 0 = the default below
void set_gemm3_min_m(int m) { g_g3_min_m.store(m > 0 ? m : 0); }
static int gemm3_min_m() {
  const int v = g_g3_min_m.load();
  return v > 0 ? v : kG3MinM;
}
void set_gemm3_bm(int bm) { g_g3_bm.store(bm == 64 || bm == 128 || bm == 256 || bm == 257 || bm == 258 ? bm : 0); }  // 257: 256-row tile, tall wave tiles; 258: the automatic choice without them (A-B runs)

// hipErrorNotSupported = use the first-generation kernel (scratch allocation failed, sizes beyond 32-bit offsets ...)
hipError_t launch_gemm2(const SmallMArgs& a, hipStream_t st) {
  static const bool off = getenv("NS_GEMM_V1") != nullptr;  // diagnostics
  if (off) return hipErrorNotSupported;
  const ns_weight* w0 = a.seg[0].w;
  // fp8 weights span 2^-15 .. 2^15 per code before their scale: they stay on the first-generation kernel, which
  // applies the group scale to the fp32 MFMA result
  if (w0->kind == WK_F8) return hipErrorNotSupported;
  Gemm2Params p;
  memset(&p, 0, sizeof(p));
  p.m = a.m;
  p.k = w0->k;
  p.n = w0->n;
  p.nchunks = (w0->k + kG2KC - 1) / kG2KC;
  const int kpad = p.nchunks * kG2KC;
  // fp16 activations: the caller's shadow when it is usable as is, else one conversion pass into scratch
  const _Float16* a16 = static_cast<const _Float16*>(a.a16);
  if (a16 && (w0->k % kG2KC != 0 || (a.lda & 7) != 0 || (reinterpret_cast<uintptr_t>(a16) & 15) != 0)) a16 = nullptr;
  if (a16) {
    p.a16 = a16;
    p.lda16 = a.lda;
  } else if (!a.a) {
    return hipErrorNotSupported;  // fp16-only activations that cannot be multiplied as they are: the caller's general path
  } else {
    const size_t bytes = size_t(a.m) * kpad * 2;
    if (bytes >= (size_t(1) << 32)) return hipErrorNotSupported;
    _Float16* sc = static_cast<_Float16*>(stream_scratch(st, bytes, 0));
    if (!sc) return hipErrorNotSupported;
    const size_t units = size_t(a.m) * kpad / 8;
    hipLaunchKernelGGL(cvt_a16_kernel, dim3(unsigned((units + 255) / 256)), dim3(256), 0, st, a.a, sc, a.m, w0->k, a.lda, kpad);
    p.a16 = sc;
    p.lda16 = kpad;
  }
  if (size_t(a.m) * size_t(p.lda16) * 2 >= (size_t(1) << 32)) return hipErrorNotSupported;
  p.ksteps = w0->ksteps;
  p.ntiles = w0->ntiles;
  p.codes = w0->codes;
  p.scales = w0->scales;
  p.zps = w0->zps;
  p.codes_bytes = uint32_t(w0->codes_bytes);
  p.scales_bytes = uint32_t(w0->scales_bytes);
  p.zps_bytes = uint32_t(w0->zps_bytes);
  p.qstride = w0->qstride;
  p.sstride = w0->sstride;
  p.zstride = w0->zstride;
  p.c = a.seg[0].c;
  p.c16 = static_cast<_Float16*>(a.seg[0].c16);
  p.ldc = a.ldc;
  p.srows = w0->srows;
  if (!srow_params(w0, &p.srow_mul, &p.srow_shift)) return hipErrorNotSupported;
  p.epilogue = a.epilogue;
  p.d = a.d;
  p.ldd = a.ldd;
  if (w0->kind == WK_F4) f4_lut_planes(w0->lut, &p.lut);
  p.spre = w0->g2_pre;
  p.spost = w0->g2_post;
  static const int g3_diag = getenv("NS_G3_DIAG") ? atoi(getenv("NS_G3_DIAG")) : 0;
  p.diag = g3_diag;
  p.nbn = (w0->ntiles + kG2Tiles - 1) / kG2Tiles;
  if (a.dual) {  // fused gate/up at GEMM size (round 5): gemm3_kernel only, a column block = 4 tile pairs = 64 output columns
    static const bool g3_off_env = getenv("NS_GEMM3") && atoi(getenv("NS_GEMM3")) != 1;
    const ns_weight* w = a.seg[1].w;
    if (a.nseg != 2 || !w || a.m < gemm3_min_m() || (p.lda16 & 7) != 0 || g3_off_env || a.d || kG3M32) return hipErrorNotSupported;
    if (w->k != w0->k || w->n != w0->n || w->kind != w0->kind || w->sps != w0->sps || w->scale_dt != w0->scale_dt || w->asym != w0->asym ||
        w->ksteps != w0->ksteps || w->qstride != w0->qstride || w->sstride != w0->sstride || w->zstride != w0->zstride ||
        w->srows != w0->srows || w->g2_pre != w0->g2_pre || w->g2_post != w0->g2_post || w->qtype != w0->qtype ||
        w->codes_bytes != w0->codes_bytes || w->scales_bytes != w0->scales_bytes || w->zps_bytes != w0->zps_bytes)
      return hipErrorNotSupported;
    if (!p.c && !p.c16) return hipErrorInvalidValue;
    p.dual = 1;
    p.codes2 = w->codes, p.scales2 = w->scales, p.zps2 = w->zps;
    p.c2 = a.c2;
    p.nbn = (w0->ntiles + 3) / 4;
    p.cpx = (p.nbn + 7) / 8;
    p.wide = 1;
    p.bm3 = a.m <= 64 ? 64 : 128;
    p.ksplit = 1;
    p.cps = p.nchunks;
    const int nbm3 = (a.m + p.bm3 - 1) / p.bm3;
    const dim3 grid3(unsigned(8 * p.cpx * nbm3), 1u);
    switch (w0->kind) {
      case WK_INT4:
        switch (w0->sps) {
          case 4: return launch_gemm3_s<WK_INT4, 4>(p, w0->scale_dt, w0->asym, grid3, st, false);
          case 2: return launch_gemm3_s<WK_INT4, 2>(p, w0->scale_dt, w0->asym, grid3, st, false);
          default: return launch_gemm3_s<WK_INT4, 1>(p, w0->scale_dt, w0->asym, grid3, st, false);
        }
      case WK_INT8:
        if (w0->sps == 2) return launch_gemm3_s<WK_INT8, 2>(p, w0->scale_dt, w0->asym, grid3, st, false);
        return launch_gemm3_s<WK_INT8, 1>(p, w0->scale_dt, w0->asym, grid3, st, false);
      default:
        switch (w0->sps) {
          case 4: return launch_gemm3_s<WK_F4, 4>(p, w0->scale_dt, w0->asym, grid3, st, false);
          case 2: return launch_gemm3_s<WK_F4, 2>(p, w0->scale_dt, w0->asym, grid3, st, false);
          default: return launch_gemm3_s<WK_F4, 1>(p, w0->scale_dt, w0->asym, grid3, st, false);
        }
    }
  }
  if (a.nseg > 1) {  // fused QKV at GEMM size: gemm3_kernel only, whole column blocks per matrix
    static const bool g3_off_env = getenv("NS_GEMM3") && atoi(getenv("NS_GEMM3")) != 1;
    if (a.nseg > 3 || a.dual || a.m < gemm3_min_m() || (p.lda16 & 7) != 0 || g3_off_env) return hipErrorNotSupported;
    p.nseg = a.nseg;
    int bn0 = 0;
    for (int i = 0; i < a.nseg; i++) {
      const ns_weight* w = a.seg[i].w;
      if (w->k != w0->k || w->kind != w0->kind || w->sps != w0->sps || w->scale_dt != w0->scale_dt || w->asym != w0->asym ||
          w->ksteps != w0->ksteps || w->qstride != w0->qstride || w->sstride != w0->sstride || w->zstride != w0->zstride ||
          w->srows != w0->srows || w->qtype != w0->qtype)
        return hipErrorNotSupported;
      p.seg_spre[i] = w->g2_pre, p.seg_spost[i] = w->g2_post;
      p.seg_bn0[i] = bn0;
      p.seg_n[i] = w->n;
      p.seg_codes[i] = w->codes, p.seg_scales[i] = w->scales, p.seg_zps[i] = w->zps;
      p.seg_codes_bytes[i] = uint32_t(w->codes_bytes), p.seg_scales_bytes[i] = uint32_t(w->scales_bytes);
      p.seg_zps_bytes[i] = uint32_t(w->zps_bytes);
      p.seg_c[i] = a.seg[i].c;
      p.seg_c16[i] = static_cast<_Float16*>(a.seg[i].c16);
      bn0 += (w->ntiles + kG2Tiles - 1) / kG2Tiles;
    }
    p.nbn = bn0;
    if (a.rope) {  // RoPE + kv-cache append in the epilogue: plain whole-head RoPE, whole 128-column blocks per matrix, 16-byte fp32 / 8-byte cache stores
      const ns_qkv_rope& r = *a.rope;
      const bool ok = a.nseg == 3 && r.mode == 0 && r.n_dims == r.head_size && r.head_size >= 4 && (r.head_size & 3) == 0 && r.kcache16 && r.vcache16 && r.cos_sin &&
                      r.n_past >= 0 && a.seg[0].w->n == r.heads * r.head_size && a.seg[1].w->n == r.heads_kv * r.head_size &&
                      a.seg[2].w->n == r.heads_kv * r.head_size && a.seg[0].w->n % 128 == 0 && a.seg[1].w->n % 128 == 0 && (a.ldc & 3) == 0 &&
                      (r.cache_step_sl & 3) == 0 && (r.cache_step_head & 3) == 0 && (reinterpret_cast<uintptr_t>(r.kcache16) & 7) == 0 &&
                      (reinterpret_cast<uintptr_t>(r.vcache16) & 7) == 0 && (reinterpret_cast<uintptr_t>(r.cos_sin) & 15) == 0 && a.seg[0].c &&
                      (reinterpret_cast<uintptr_t>(a.seg[0].c) & 15) == 0 && (!a.seg[1].c || (reinterpret_cast<uintptr_t>(a.seg[1].c) & 15) == 0) &&
                      (!a.seg[2].c || (reinterpret_cast<uintptr_t>(a.seg[2].c) & 15) == 0) && !a.seg[0].c16 && !a.seg[1].c16 && !a.seg[2].c16 &&
                      a.epilogue == NS_EPI_NONE && !kG3M32;
      if (!ok) return hipErrorNotSupported;
      p.rope_on = 1, p.rope_hs = r.head_size, p.rope_npast = r.n_past;
      p.rope_tab = reinterpret_cast<const float2*>(r.cos_sin);
      p.rope_kc = static_cast<_Float16*>(r.kcache16), p.rope_vc = static_cast<_Float16*>(r.vcache16);
      p.rope_csl = r.cache_step_sl, p.rope_chead = r.cache_step_head;
    }
  } else if (a.rope) {
    return hipErrorNotSupported;
  }
  p.cpx = (p.nbn + 7) / 8;
  // third-generation kernel (256-row tiles, A by LDS DMA, B dequantised in registers): from 192 rows up; below that
  // its row tile would be mostly padding.  NS_GEMM3=0 keeps the second generation (diagnostics / A-B runs).
  static const int g3_mode = getenv("NS_GEMM3") ? atoi(getenv("NS_GEMM3")) : 1;  // 0: gemm2, 1: gemm3, 2: gemm3d (NS_WITH_GEMM3D builds)
  const bool g3_off = g3_mode == 0;
  const bool deep = g3_mode == 2;
  if (!g3_off && a.m >= gemm3_min_m() && (p.lda16 & 7) == 0) {
    // 128-row tiles (three workgroups per CU hide each other's barrier / dequantisation / output phases) unless the
    // output has so many tiles that the tall ones' halved A traffic wins (profiles/r02r_gemm3_bm.txt: at 2048 rows equal
    // or better up to 11008 columns, 7 % worse at 32000)
    static const int bm_env0 = getenv("NS_G3_BM") ? atoi(getenv("NS_G3_BM")) : 0;  // diagnostics
    const int bm_env = g_g3_bm.load() ? g_g3_bm.load() : bm_env0;                  // ns_hip_set_tuning("g3_bm", 128 / 256 / 0)
    const int tall_tiles = p.nbn * ((a.m + 255) / 256);
    // tall wave tiles (256 x 32, TALL; ns_hip_set_tuning("g3_bm", 257)): shape by shape at 2048 rows they gain 2-4 % where a
    // workgroup runs long (11008 x 4096, 4096 x 11008, 32000 x 4096) and lose 14 % on 4096 x 4096 (profiles/r03v_gemm3_tall.txt);
    // in a layer's sequence of GEMMs, steady state, choosing them by that rule gained nothing (986 vs 993 TFLOPS,
    // profiles/r03w_prefill_ab.txt) — not selected automatically
    p.tall3 = bm_env == 257 && w0->kind == WK_INT4;
    p.bm3 = p.tall3 ? 256 : bm_env == 64 || bm_env == 128 || bm_env == 256 ? bm_env : a.m <= 64 ? 64 : (tall_tiles >= 1024 && w0->kind != WK_INT8 ? 256 : 128);  // 8-bit codes: the tall tile's LDS
                                                                                            // footprint (81 KiB) leaves one workgroup per CU
    // outputs: the per-wave epilogue by default; the cross-wave one (ns_hip_set_tuning("g3_wide", 1) / NS_G3_WIDE=1) measured 2-8 % SLOWER
    // with both outputs on the 7B shapes at 2048 rows (profiles/r05b_gemm3_epilogue_ab.txt: 4096^2 100.4 vs 92.9 us, 11008 x 4096 208.2 vs
    // 204.8, 4096 x 11008 197.5 vs 191.7) and level with the fp32 output alone — its two extra workgroup barriers per 64-row half cost more
    // than the longer store runs return.  A launch without an fp32 output (fp16 only: the consumer is this library's next GEMM) and the
    // gate / up pairs need it (their per-wave runs would be 16 columns)
    static const int wide_env = getenv("NS_G3_WIDE") ? atoi(getenv("NS_G3_WIDE")) : 0;
    const int wide_on = g_g3_wide.load() >= 0 ? g_g3_wide.load() : wide_env;
    if (!p.c && p.nseg <= 1) {
      if (!p.c16) return hipErrorInvalidValue;
      if (p.bm3 == 256 && !p.tall3) p.bm3 = 128;
    }
    p.wide = (p.bm3 != 256 || p.tall3) && !kG3M32 && !p.rope_on && (wide_on != 0 || (!p.c && p.nseg <= 1));  // (the RoPE epilogue lives in the per-wave form)
    const int nbm3 = (a.m + p.bm3 - 1) / p.bm3;
    p.ksplit = 1;
    p.cps = p.nchunks;
    {
      static const bool no_splitk = getenv("NS_NO_SPLITK") != nullptr;  // diagnostics
      const int tiles = p.nbn * nbm3;
      int ks = 1;
      // gemm3_kernel wants two workgroups per CU, the deep kernel one
      while (ks < 8 && tiles * ks * 2 <= (deep ? 256 : 512) && p.nchunks / (ks * 2) >= 8) ks *= 2;
      if (ks > 1 && !no_splitk && p.nseg <= 1 && p.c) {  // (a launch with the fp16 output only never splits K: the reduction pass writes fp32)
        const size_t bytes = size_t(ks) * a.m * w0->n * 4;
        float* part = static_cast<float*>(stream_scratch(st, bytes, 2));
        if (part) {
          p.ksplit = ks;
          p.cps = ((p.nchunks + ks - 1) / ks + 1) & ~1;  // even: a 128-deep superstep never straddles two splits
          p.part = part;
        }
      }
    }
    const dim3 grid3(unsigned(8 * p.cpx * nbm3), unsigned(p.ksplit));
#define NS_G3DISPATCH(KIND)                                                           \
  switch (w0->sps) {                                                                  \
    case 4: return launch_gemm3_s<KIND, 4>(p, w0->scale_dt, w0->asym, grid3, st, deep);     \
    case 2: return launch_gemm3_s<KIND, 2>(p, w0->scale_dt, w0->asym, grid3, st, deep);     \
    default: return launch_gemm3_s<KIND, 1>(p, w0->scale_dt, w0->asym, grid3, st, deep);    \
  }
    if (w0->kind == WK_INT4) {
      NS_G3DISPATCH(WK_INT4)
    } else if (w0->kind == WK_INT8) {
      if (w0->sps == 2) return launch_gemm3_s<WK_INT8, 2>(p, w0->scale_dt, w0->asym, grid3, st, deep);
      return launch_gemm3_s<WK_INT8, 1>(p, w0->scale_dt, w0->asym, grid3, st, deep);
    } else {
      NS_G3DISPATCH(WK_F4)
    }
#undef NS_G3DISPATCH
  }
  if (p.rope_on) return hipErrorNotSupported;  // (the RoPE epilogue is gemm3_kernel's)
  const int nbm = (a.m + kG2BM - 1) / kG2BM;
  // few output tiles (M up to a few hundred rows): split K so that the launch still fills the chip
  p.ksplit = 1;
  p.cps = p.nchunks;
  {
    static const bool no_splitk = getenv("NS_NO_SPLITK") != nullptr;  // diagnostics
    const int tiles = p.nbn * nbm;
    int ks = 1;
    while (ks < 8 && tiles * ks * 2 <= 256 && p.nchunks / (ks * 2) >= 8) ks *= 2;
    if (ks > 1 && !no_splitk) {
      const size_t bytes = size_t(ks) * a.m * w0->n * 4;
      float* part = static_cast<float*>(stream_scratch(st, bytes, 2));
      if (part) {
        p.ksplit = ks;
        p.cps = (p.nchunks + ks - 1) / ks;
        p.part = part;
      }
    }
  }
  const dim3 grid(unsigned(8 * p.cpx * nbm), unsigned(p.ksplit));
  const size_t lds = size_t(kG2Stages) * kG2StageBytes;
#ifdef NS_GEMM_MIN  // development builds: the two bench formats only
  if (w0->scale_dt != DT_BF16 || w0->asym) return hipErrorNotSupported;
  if (w0->kind == WK_INT4 && w0->sps == 4) return launch_gemm2_k<WK_INT4, 4, SK_BF16>(p, false, grid, lds, st);
  if (w0->kind == WK_INT8 && w0->sps == 2) return launch_gemm2_k<WK_INT8, 2, SK_BF16>(p, false, grid, lds, st);
  return hipErrorNotSupported;
#else
#define NS_G2DISPATCH(KIND)                                                                \
  switch (w0->sps) {                                                                       \
    case 4: return launch_gemm2_s<KIND, 4>(p, w0->scale_dt, w0->asym, grid, lds, st);      \
    case 2: return launch_gemm2_s<KIND, 2>(p, w0->scale_dt, w0->asym, grid, lds, st);      \
    default: return launch_gemm2_s<KIND, 1>(p, w0->scale_dt, w0->asym, grid, lds, st);     \
  }
  if (w0->kind == WK_INT4) {
    NS_G2DISPATCH(WK_INT4)
  } else if (w0->kind == WK_INT8) {
    if (w0->sps == 2) return launch_gemm2_s<WK_INT8, 2>(p, w0->scale_dt, w0->asym, grid, lds, st);
    return launch_gemm2_s<WK_INT8, 1>(p, w0->scale_dt, w0->asym, grid, lds, st);
  } else {
    NS_G2DISPATCH(WK_F4)
  }
#undef NS_G2DISPATCH
#endif
}

void touch_gemm_module() {
  hipFuncAttributes fa;
  (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(gemm3_kernel<WK_INT4, 4, SK_BF16, false, 128, false, false>));
  (void)hipGetLastError();
}
}  // namespace ns
