// ns_gemvs.hip — gemvs_kernel: the small-batch (2 .. 16 rows) and narrow-output member of libns_hip.so's weight-streaming
// family.  Same arithmetic as gemv_kernel (ns_gemv.hip): per 1 KiB weight record NJ x v_mfma_f32_16x16x32_f16 on the raw
// codes, the group scale applied to the fp32 MFMA result, fp16 activations — w = (code - zp) * scale exactly, fp32
// accumulation (reference: bestla/bestla/kernel_ref.h:2489-2531 gemv_4bit_fp32_fp32 and the M <= 4 envelope of
// bestla_wrapper.h:643-688; the batched decode step is models/llama/llama.cpp:65-71).  What differs is the DECOMPOSITION.
//
// gemv_kernel gives every 16-column tile its own workgroup, and every workgroup stages all rows of A over the whole K.
// At one row that is 8 KB beside 32 KB of weights; at 8 rows of a 4096-deep matrix it is 64 KB of activations from L2
// for 32 KB of weights from HBM — a 2:1 activation-to-weight ratio per workgroup, one workgroup of 4 waves per CU
// (LDS), staging and streaming serialised by the barrier between them: 0.19 of the HBM roofline on BASELINE config 4
// (profiles/r03y_config_bench.json).  Here:
//
//   * ONE workgroup per CU, persistent over its share of the launch's tiles: the activations are staged ONCE per
//     workgroup and shared by every tile it streams (FFN gate/up at 8 rows: 64 KB of A per 230 KB of weights instead
//     of per 64 KB); the prologue, the A wait and the barrier are paid once per launch, not once per tile.
//   * split-K across workgroups where the rows x K slice does not fit LDS (FFN down: 8 x 14336 fp16 = 229 KB) or the
//     output is too narrow to give every CU a tile: workgroup (slice s, tile group t) stages only A[:, slice s] and
//     streams slice s of its tiles; the per-slice partial sums go to a slab in HBM and are added in slice order
//     (deterministic: no atomics on values) with the epilogue — by gemvs_finalize_kernel in a second launch (default), or
//     inside the launch by the tile's OWNER, the workgroup of slice (tile ordinal mod S): every workgroup streams the
//     tiles it owns LAST, so when an owner reaches its tile the other slices of it were published (write-through stores,
//     then a ticket) at least a tile's streaming time earlier and the hand-off is not waited for.  Both give the same bits;
//     they measured equal (profiles/r04u_split_k_finish.txt), and splitting beyond what LDS forces does not pay: fused
//     gate/up at 8 rows 23.0 us unsplit, 28.2 with 2 slices, 34.2 with 4 (per-tile flush + reduction + publish).
//   * NS streaming waves + ONE service wave per workgroup.  The streaming waves split a tile's k-steps as in
//     gemv_kernel (private LDS-DMA rings, hand-counted vmcnt) and run straight on into the next tile: at the end of a
//     tile each writes its partial sums to an LDS slot and bumps an LDS counter — no workgroup barrier.  The service
//     wave polls the counter, adds the partials in wave order, applies the epilogue and stores — so reduction,
//     epilogue-operand fetch and the output stores of tile j overlap the stream of tile j + 1.
//   * f4 weights (NF4 / FP4: a 16-entry value table, kernel_ref.h:1456-1478) are decoded through LDS, not the VALU: a
//     table of all 256 code PAIRS (byte -> two fp16), replicated once per LDS bank (32 KB) so that 64 lanes read it
//     conflict-free; a lane's 16 B of codes cost 16 ds_read_b32 + 32 address operations instead of ~100 v_perm / v_bfi
//     (profiles/r04b_ablation.txt: the VALU decode was 9-10 of the gate/up launch's 26 us).  A byte of the streaming
//     layout holds codes (i0, i2) of a lane's eight, not an MFMA k-pair, so the staged activations are shuffled once in
//     LDS to the matching order (a0 a2 a4 a6 a1 a3 a5 a7 per 16-byte fragment) instead of shuffling every weight.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <array>
#include <map>
#include <mutex>
#include <vector>

#include "../../include/ns_bestla.h"
#include "ns_common.h"
#include "ns_dev.h"

namespace ns {

#ifndef NS_GVS_PF
#define NS_GVS_PF 2
#endif
// diagnostic builds (scripts/build_variants.sh): 1 = no code -> fp16 conversion (raw bits as operands), 2 = no activation
// fragment reads, 3 = records are waited for and refilled but not read or multiplied (the structure's streaming rate)
#ifndef NS_GVS_ABL
#define NS_GVS_ABL 0
#endif
constexpr int kGsPF = NS_GVS_PF;        // records each streaming wave keeps in flight (with up to 15 waves per CU; deeper
                                        // rings measured no faster from 12 waves on and cost LDS, profiles/r04b_ablation.txt)
constexpr int kGsR = 2;                 // reduction slots (tiles whose partial sums may be pending in LDS)
constexpr int kGsMaxRows = 16;
constexpr int kGsMaxNS = 15;            // streaming waves (+ the service wave = 16 waves = 1024 threads)
constexpr size_t kGsMaxLds = 160 * 1024;
constexpr uint32_t kGsCtlBytes = 32;    // LDS control words: cnt[kGsR] at 0, done at 16
constexpr uint32_t kGsTblBytes = 32 * 1024;  // f4 pair table: 256 entries x 32 bank copies x 4 B, at LDS offset 0
constexpr uint32_t kGsSpinLimit = 1u << 22;  // polls of an LDS word before a wave gives up (about 0.2 s)

struct GvsMat {
  const uint8_t* wbase;  // ONE allocation: records at 0, scales at s_off, zero points at z_off
  uint32_t s_off, z_off;
  uint32_t tile_begin;
  int n;
  float* c;
  _Float16* c16;
};

struct GvsParams {
  // ---- hot: the prologue's words ----
  const uint8_t* wb0;
  const uint8_t* wb1;
  const uint8_t* wb2;
  const void* a;            // activations, fp16 [m][lda]
  uint32_t so0, so1, so2;   // scale offsets of the matrices
  uint32_t zo0, zo1, zo2;   // zero-point offsets (asymmetric only)
  uint32_t ks;              // k-steps per tile (whole K)
  uint32_t ksl;             // k-steps per slice
  uint32_t s_log2;          // log2(slices): workgroup b streams slice b & (S - 1) of tile group b >> s_log2
  uint32_t qstride, sstride, zstride;
  uint32_t srows, srow_mul, srow_shift;
  uint32_t tb1, tb2;        // first tile of matrices 1 and 2 of a fused QKV launch (2^32 - 1: absent)
  uint32_t ntiles;          // tile units of the launch (dual: tile pairs)
  uint32_t tg_count;        // tile groups = workgroups per slice; group t streams tiles t, t + tg_count, ...
  uint32_t tiles_base, tiles_rem;  // ntiles = tiles_base * tg_count + tiles_rem
  uint32_t ns;              // streaming waves
  int m, k, lda;
  uint32_t row_stride;      // halves per staged row
  uint32_t ctl_off, red_off, ring_off, ring_stride;
  uint32_t red_wave, red_slot;  // bytes of partial sums per wave / per slot
  uint32_t a_off;           // LDS offset of the staged activations (behind the f4 pair table when there is one)
  uint32_t a_bytes;         // bytes of the staged activations (rows x row_stride halves)
  uint32_t lut16[8];        // f4: the 16 table values as fp16, two per word
  const void* tbl_img;      // f4: the 32 KB pair table as a device image (round 5: fetched by LDS-DMA beside the activations instead of
                            // ~1.4 us of VALU per launch); nullptr: the waves build it (the image did not exist yet at capture time)
  // ---- cold ----
  GvsMat mat[3];
  float* c2;
  const float* d;
  int ldc, ldd, epilogue;
  float* slab;              // split-K: [slice][tile unit][q][64 lanes][4] fp32 partial sums
  uint32_t* tickets;        // split-K finished inside the launch: one counter per tile unit (zero between launches); nullptr:
                            // the partial sums are added by gemvs_finalize_kernel
  uint32_t slab_bytes;
  F4Lut lut;
  F8Consts f8;
#ifdef NS_TRACE
  unsigned long long* trace;
#endif
};
#ifdef NS_TRACE
unsigned long long* trace_buffer();
#define NS_SSTAMP(i)                                                                                     \
  do {                                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 4096)                                                     \
      p.trace[(size_t(blockIdx.x) * 16 + (threadIdx.x >> 6)) * 8 + (i)] = wall_clock64();                 \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
  } while (0)
#else
#define NS_SSTAMP(i)
#endif
using GvsKArgs = const __attribute__((address_space(4))) GvsParams*;
__device__ __forceinline__ GvsKArgs gvs_late_args() {
  uint64_t v = reinterpret_cast<uint64_t>(__builtin_amdgcn_kernarg_segment_ptr());
  asm volatile("" : "+s"(v));
  return reinterpret_cast<GvsKArgs>(v);
}

template <int N>
__device__ __forceinline__ void gvs_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

enum GvsMode { GS_PLAIN = 0, GS_DUAL = 1, GS_MSEG = 2 };

// ---- f4 pair-table images (host side): entry e = {value[e & 15], value[e >> 4]} as two fp16, 32 bank copies each — what build_table
//      writes.  One image per value table (NF4 / FP4-BNB / FP4-E2M1), created on the first launch that wants it OUTSIDE a stream
//      capture (an allocation + a synchronous copy); a launch that finds none gets nullptr and builds the table with its waves ----
static std::mutex g_tbl_mu;
static std::map<std::array<uint16_t, 16>, void*> g_tbl_img;
static const void* gvs_f4_table_image(const _Float16* lut, hipStream_t st) {
  static const bool off = getenv("NS_GVS_TABLE_DMA") && atoi(getenv("NS_GVS_TABLE_DMA")) == 0;  // A-B runs
  if (off) return nullptr;
  std::array<uint16_t, 16> key;
  for (int i = 0; i < 16; i++) key[i] = __builtin_bit_cast(uint16_t, lut[i]);
  std::lock_guard<std::mutex> lk(g_tbl_mu);
  auto it = g_tbl_img.find(key);
  if (it != g_tbl_img.end()) return it->second;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
    (void)hipGetLastError();
    return nullptr;
  }
  std::vector<uint32_t> img(kGsTblBytes / 4);
  for (uint32_t e = 0; e < 256; e++)
    for (uint32_t c = 0; c < 32; c++) img[e * 32 + c] = uint32_t(key[e & 15]) | (uint32_t(key[e >> 4]) << 16);
  void* d = nullptr;
  if (hipMalloc(&d, kGsTblBytes) != hipSuccess || hipMemcpy(d, img.data(), kGsTblBytes, hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipGetLastError();
    if (d) (void)hipFree(d);
    return nullptr;  // (not cached: tried again next time)
  }
  g_tbl_img.emplace(key, d);
  return d;
}

// epilogue of one tile: lane (nn, g) holds rows 4g .. 4g+3 of column `col` (custom::epilogue::*, bestla_f32f32_forward;
// dual: tmp1 = act(A*W1), out = (A*W3) * tmp1, neural_speed/core/layers/ip_fusion_ffn.cpp:364-406)
template <bool DUAL>
__device__ __forceinline__ void gvs_epilogue(const floatx4* sum, int m, int col, bool col_ok, int g, float* cbase, _Float16* c16,
                                             int ldc, const float* dvp, float* c2, int epi) {
#pragma unroll
  for (int rr = 0; rr < 4; rr++) {
    const int row = 4 * g + rr;
    if (!(col_ok && row < m)) continue;
    float v = sum[0][rr];
    if constexpr (DUAL) {
      const float t1 = (epi == 5) ? epi_silu(v) : epi_gelu(v);
      if (c2) c2[size_t(row) * ldc + col] = t1;
      v = sum[DUAL ? 1 : 0][rr] * t1;
    } else {
      const float dv = dvp[rr];
      switch (epi) {
        case 1: v = v + dv; break;
        case 2: v = v * dv; break;
        case 3: v = epi_gelu(v + dv); break;
        case 4: v = epi_gelu(v); break;
        case 5: v = epi_silu(v); break;
        default: break;
      }
    }
    cbase[size_t(row) * ldc + col] = v;
    if (c16) c16[size_t(row) * ldc + col] = (_Float16)v;
  }
}

template <int KIND, int SPS, int SK, bool ASYM, int MODE>
__global__ __launch_bounds__(1024) void gemvs_kernel(const GvsParams p) {
  constexpr bool DUAL = MODE == GS_DUAL, MSEG = MODE == GS_MSEG;
  constexpr int NJ = kind_is_8bit(KIND) ? 2 : 4;
  constexpr int KSTEP = NJ * 32;
  constexpr int NQ = DUAL ? 2 : 1;
  constexpr int PF = kGsPF;
  static_assert(PF % NQ == 0, "ring slots alternate between the two matrices");
  constexpr int SBYTES = SPS * (SK == SK_F32 ? 4 : 2);
  // f4 codes are decoded through the LDS pair table when the launch has enough records per wave to pay for building it
  // (host: p.a_off != 0 — the table sits in front of the activations), by v_perm lookups otherwise
  constexpr bool TBLK = KIND == WK_F4 && NS_GVS_ABL != 1;
  const bool TBL = TBLK && p.a_off != 0;
  using Corr = CorrRaw<SPS, SK, ASYM>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __attribute__((address_space(3))) unsigned char* LdsPtr;

  // one batch of scalar loads for everything the prologue needs (hipcc would otherwise sink each load to its first use)
  asm volatile("" ::"s"(p.wb0), "s"(p.a), "s"(p.so0), "s"(p.ks), "s"(p.ksl), "s"(p.s_log2), "s"(p.qstride), "s"(p.sstride),
               "s"(p.srows), "s"(p.srow_mul), "s"(p.srow_shift), "s"(p.ntiles), "s"(p.tg_count), "s"(p.tiles_base), "s"(p.tiles_rem),
               "s"(p.ns));
  asm volatile("" ::"s"(p.m), "s"(p.k), "s"(p.lda), "s"(p.row_stride), "s"(p.ctl_off), "s"(p.red_off), "s"(p.ring_off),
               "s"(p.ring_stride), "s"(p.red_wave), "s"(p.red_slot), "s"(p.a_off));
  if constexpr (DUAL || MSEG) asm volatile("" ::"s"(p.wb1), "s"(p.so1));
  if constexpr (MSEG) asm volatile("" ::"s"(p.wb2), "s"(p.so2), "s"(p.tb1), "s"(p.tb2));
  if constexpr (ASYM) asm volatile("" ::"s"(p.zo0), "s"(p.zo1), "s"(p.zo2), "s"(p.zstride));

  NS_SSTAMP(0);
  const int tid = threadIdx.x;
  const uint32_t w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = tid & 63;
  const int nn = l & 15, g = l >> 4;
  const uint32_t NS = p.ns;
  const uint32_t sl = blockIdx.x & ((1u << p.s_log2) - 1u);   // K slice of this workgroup
  const uint32_t tg = blockIdx.x >> p.s_log2;                 // tile group
  const uint32_t kb = sl * p.ksl;                             // first k-step of the slice
  const uint32_t nks = min(p.ks - kb, p.ksl);                 // k-steps of the slice (host: kb < ks)
  const uint32_t ntw = p.tiles_base + (tg < p.tiles_rem ? 1u : 0u);  // tiles of this workgroup
  // order in which this workgroup streams its tiles (ordinals j = 0 .. ntw - 1 -> tile tg + j * tg_count): in-launch split-K
  // puts the tiles this slice OWNS (j % S == slice) last, the others in ascending order in front
  const uint32_t S1 = (1u << p.s_log2) - 1u;
  const bool own_last = p.tickets != nullptr && S1 != 0;
  const uint32_t n_owned = own_last && sl < ntw ? (ntw - sl + S1) >> p.s_log2 : 0u;
  const uint32_t n_other = ntw - n_owned;
  auto tile_ord = [&](uint32_t i) -> uint32_t {
    if (!own_last) return i;
    if (i >= n_other) return sl + ((i - n_other) << p.s_log2);
    const uint32_t b = i / S1, r = i - b * S1;  // S - 1 foreign ordinals per block of S
    return (b << p.s_log2) + (r < sl ? r : r + 1u);
  };
  const int rows = min(p.m, kGsMaxRows);
  const int rows_q = (rows + 3) >> 2;                         // lane groups g that hold live output rows
  const uint32_t lds0 = uint32_t(reinterpret_cast<uintptr_t>((LdsPtr)(smem)));
  const uint32_t voff_q = uint32_t(l) * 16u;

  // ---- the slice of the activations: fp16 rows, HBM/L2 -> LDS by DMA in 1 KiB pieces, dealt to ALL waves.  Requested
  //      AFTER the streaming waves' first weight records (those come from HBM and take longest; profiles/r04e_trace.txt: with
  //      the activations and the table in front, the first weight request left 3.4 us after entry) ----
  const uint32_t a_row_bytes = nks * uint32_t(KSTEP) * 2u;
  const uint32_t a_pieces = (a_row_bytes + 1023u) >> 10;
  const uint32_t a_total = uint32_t(rows) * a_pieces;
  const LdsPtr al = (LdsPtr)(smem) + p.a_off;
  auto stage_a = [&]() {
    const Rsrc ra = make_rsrc(p.a, uint32_t(rows - 1) * uint32_t(p.lda) * 2u + uint32_t(p.k) * 2u);
    uint32_t r = 0, c = w;  // piece u = w + i * (NS + 1) is (row r, piece c): stepped, not divided
    for (uint32_t u = w; u < a_total; u += NS + 1u) {
      while (c >= a_pieces) c -= a_pieces, r++;
      const uint32_t left = a_row_bytes - (c << 10);
#if defined(__HIP_DEVICE_COMPILE__)
      if (uint32_t(l) * 16u < left)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, reinterpret_cast<__attribute__((address_space(3))) void*>(al + r * p.row_stride * 2u + (c << 10)),
                                                 16, voff_q, r * uint32_t(p.lda) * 2u + kb * uint32_t(KSTEP) * 2u + (c << 10), 0, 0);
#endif
      c += NS + 1u;
    }
  };
  // f4, round 5: the pair table is the same 32 KB for every launch on a value table — one device image per table (host:
  // gvs_f4_table_image), its 32 one-KiB pieces dealt to all waves and requested right behind the activation pieces (L2-resident:
  // every workgroup of every launch reads the same 32 KB); same queue position as the activations, so the same wait covers them
  auto stage_table = [&]() {
#if defined(__HIP_DEVICE_COMPILE__)
    if (TBL) {
      const Rsrc rt = make_rsrc(p.tbl_img, kGsTblBytes);
      for (uint32_t u = w; u < kGsTblBytes / 1024u; u += NS + 1u)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rt, reinterpret_cast<__attribute__((address_space(3))) void*>((LdsPtr)(smem) + (u << 10)), 16, voff_q, u << 10, 0, 0);
    }
#endif
  };
  auto build_table = [&]() {
    // f4: every wave writes its share of the pair table while its requests are in flight.  Lane l < 16 holds table
    // value l (picked out of the eight argument words); entry e = {value[e & 15], value[e >> 4]} is fetched from those lanes
    // by ds_bpermute.  Stores by hand: for a visible LDS store hipcc would first wait for every LDS-DMA request in flight.
    if (TBL) {
      uint32_t lv = 0;
#pragma unroll
      for (int i = 0; i < 8; i++) lv = ((l >> 1) == i) ? p.lut16[i] : lv;
      lv = (l & 1) ? (lv >> 16) : (lv & 0xffffu);
      const uint32_t nthreads = (NS + 1u) * 64u;
      for (uint32_t idx = uint32_t(tid); idx < kGsTblBytes / 4u; idx += nthreads) {
        const uint32_t e = idx >> 5;  // 32 consecutive words = the 32 bank copies of entry e
        const uint32_t v0 = uint32_t(__builtin_amdgcn_ds_bpermute(int((e & 15u) << 2), int(lv)));
        const uint32_t v1 = uint32_t(__builtin_amdgcn_ds_bpermute(int((e >> 4) << 2), int(lv)));
        asm volatile("ds_write_b32 %0, %1" ::"v"(lds0 + idx * 4u), "v"(v0 | (v1 << 16)) : "memory");
      }
    }
  };
  // f4: a lane's code bytes pair its eight weights as (i0,i2) (i4,i6) (i1,i3) (i5,i7); every 16-byte activation fragment
  // a0..a7 is therefore re-ordered to a0 a2 a4 a6 a1 a3 a5 a7 — by the wave that requested the piece (one fragment per lane
  // and piece), once its own requests have landed and before the workgroup barrier
  auto shuffle_own_a = [&]() {
    if (TBL) {
      constexpr int B = 4;  // fragments read before the first is written back
      uint32_t r = 0, c = w;
      for (uint32_t u = w; u < a_total; u += (NS + 1u) * B) {
        uint32_t addr[B];
        uint4v d[B];
#pragma unroll
        for (int b = 0; b < B; b++) {
          addr[b] = 0xffffffffu;
          if (u + uint32_t(b) * (NS + 1u) < a_total) {
            while (c >= a_pieces) c -= a_pieces, r++;
            const uint32_t left = a_row_bytes - (c << 10);
            if (uint32_t(l) * 16u < left) addr[b] = lds0 + p.a_off + r * p.row_stride * 2u + (c << 10) + uint32_t(l) * 16u;
            c += NS + 1u;
          }
          d[b] = uint4v{0u, 0u, 0u, 0u};
          if (addr[b] != 0xffffffffu) asm volatile("ds_read_b128 %0, %1" : "=v"(d[b]) : "v"(addr[b]) : "memory");
        }
#pragma unroll
        for (int b = 0; b < B; b++) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d[b])::"memory");
#pragma unroll
        for (int b = 0; b < B; b++) {
          const uint4v o = {__builtin_amdgcn_perm(d[b].y, d[b].x, 0x05040100u), __builtin_amdgcn_perm(d[b].w, d[b].z, 0x05040100u),
                            __builtin_amdgcn_perm(d[b].y, d[b].x, 0x07060302u), __builtin_amdgcn_perm(d[b].w, d[b].z, 0x07060302u)};
          if (addr[b] != 0xffffffffu) asm volatile("ds_write_b128 %0, %1" ::"v"(addr[b]), "v"(o) : "memory");
        }
      }
    }
  };

  if (w < NS) {
    // =========================================== streaming wave ===========================================
    // the workgroup's work is the flat sequence of units u = tile ordinal * nks + k-step; unit u belongs to wave u % NS: every
    // wave gets the same number of units (+-1) whatever NS and nks are, and runs from one tile into the next
    const uint32_t nunits = ntw * nks;
    const uint32_t nun = w < nunits ? (nunits - w + NS - 1u) / NS : 0u;  // this wave's units
    const uint32_t total = nun * uint32_t(NQ);                           // items (records) this wave consumes
    const uint32_t j0 = w / nks, s0 = w - j0 * nks;                      // its first unit

    const uint8_t* wbs[3] = {p.wb0, p.wb1, p.wb2};
    const uint32_t sos[3] = {p.so0, p.so1, p.so2};
    const uint32_t zos[3] = {p.zo0, p.zo1, p.zo2};
    // issue-side tile context
    Rsrc irw[NQ];
    uint32_t iso[NQ], izo[NQ];
    uint32_t itile_q = 0, itile_c = 0;
    auto tile_ctx = [&](uint32_t T) {
      uint32_t tl = T;
      if constexpr (MSEG) {
        const int sg = int(T >= p.tb1) + int(T >= p.tb2);
        const uint8_t* wbp = sg == 0 ? wbs[0] : (sg == 1 ? wbs[1] : wbs[2]);
        irw[0] = make_rsrc(wbp, 0x80000000u);
        iso[0] = sg == 0 ? sos[0] : (sg == 1 ? sos[1] : sos[2]);
        izo[0] = ASYM ? (sg == 0 ? zos[0] : (sg == 1 ? zos[1] : zos[2])) : 0u;
        tl = T - (sg == 0 ? 0u : (sg == 1 ? p.tb1 : p.tb2));
      }
      itile_q = tl * p.ks * p.qstride;
      itile_c = tl * p.srows;
    };
    if constexpr (!MSEG) {
#pragma unroll
      for (int q = 0; q < NQ; q++) {
        irw[q] = make_rsrc(wbs[q], 0x80000000u);
        iso[q] = sos[q];
        izo[q] = zos[q];
      }
    }
    if (nun) tile_ctx(tg + tile_ord(j0) * p.tg_count);

    constexpr uint32_t SLOT = 1024u + 16u * SBYTES + (ASYM ? 16u * SPS : 0u);
    constexpr int OPS = ASYM ? 3 : 2;
    static_assert(OPS * PF <= 48, "vmcnt is a 6-bit counter: the ring and a few activation pieces must fit");
    const LdsPtr ring = (LdsPtr)(smem) + p.ring_off + w * p.ring_stride;
    const uint32_t ring_lane = uint32_t(reinterpret_cast<uintptr_t>(ring)) + uint32_t(l) * 16u;
    const uint32_t ring_corr = uint32_t(reinterpret_cast<uintptr_t>(ring)) + 1024u + uint32_t(nn) * SBYTES;
    const I4Consts i4c = {0x000f000fu, 0x00f000f0u, 0x64006400u};

    uint32_t ji = j0, ki = s0;  // issue side: tile ordinal, k-step inside the slice
    auto issue = [&](auto slot_c) {
      constexpr int slot = decltype(slot_c)::value;
      constexpr int q = slot % NQ;
#if defined(__HIP_DEVICE_COMPILE__)
      const uint32_t s = kb + ki;
      const uint32_t crow = itile_c + ((s * p.srow_mul) >> p.srow_shift);
      const LdsPtr dst = ring + slot * SLOT;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(irw[q], reinterpret_cast<__attribute__((address_space(3))) void*>(dst), 16, voff_q,
                                               itile_q + s * p.qstride, 0, 2);
      if (l < SBYTES)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(irw[q], reinterpret_cast<__attribute__((address_space(3))) void*>(dst + 1024), 16,
                                                 voff_q, iso[q] + crow * p.sstride, 0, 2);
      if constexpr (ASYM) {
        if (l < SPS)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(irw[q], reinterpret_cast<__attribute__((address_space(3))) void*>(dst + 1024 + 16 * SBYTES),
                                                   16, voff_q, izo[q] + crow * p.zstride, 0, 2);
      }
#endif
      if constexpr (q == NQ - 1) {  // next unit of this wave: NS units on
        ki += NS;
        if (ki >= nks) {
          do {
            ki -= nks;
            ji++;
          } while (ki >= nks);
          if (ji < ntw) tile_ctx(tg + tile_ord(ji) * p.tg_count);
        }
      }
    };
    auto wait_records = [&](uint32_t younger) {
      if (younger == uint32_t(PF - 1)) {
        gvs_wait_vmcnt<OPS * (PF - 1)>();
        return;
      }
      [&]<int... K>(std::integer_sequence<int, K...>) {
        (void)((younger == uint32_t(K) ? (gvs_wait_vmcnt<OPS * K>(), true) : false) || ...);
      }(std::make_integer_sequence<int, PF + 1>{});
    };
#define NS_FOR_SLOTS(BODY)                                        \
  {                                                               \
    [&]<int... I>(std::integer_sequence<int, I...>) {             \
      (([&] { constexpr int i = I; std::integral_constant<int, I> ic; (void)i; (void)ic; BODY }()), ...); \
    }(std::make_integer_sequence<int, PF>{});                     \
  }

    // ---- 1. this wave's share of the activations (a handful of requests, L2), its first weight records (HBM), then its
    //      share of the f4 table while both are in flight (the table in FRONT of the weight requests delayed them to 3.4 us
    //      after entry, profiles/r04e_trace.txt) ----
    const bool tbl_dma = p.tbl_img != nullptr;  // (uniform)
    stage_a();
    if (tbl_dma) stage_table();
    NS_FOR_SLOTS({ if (uint32_t(i) < total) issue(ic); })
    __builtin_amdgcn_sched_barrier(0);
    NS_SSTAMP(1);
    if (!tbl_dma) build_table();
    __builtin_amdgcn_sched_barrier(0);
    // ---- 2. the activation pieces are the oldest requests in this wave's queue: landed once only ring requests are left ----
    wait_records(min(total, uint32_t(PF)));
    NS_SSTAMP(2);
    shuffle_own_a();
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    NS_SSTAMP(3);

    const uint32_t aoff = uint32_t(min(nn, rows - 1)) * p.row_stride + 8 * g;
    const _Float16* a_lds = reinterpret_cast<const _Float16*>(smem + p.a_off);
    const uint32_t tbl_lane = lds0 + uint32_t(l & 31) * 4u;  // this lane's bank copy of the f4 pair table
    floatx4 acc[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) acc[q] = floatx4{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](auto slot_c, auto tbl_c, uint32_t srel) {  // srel: k-step inside the slice; tbl_c: f4 codes through the LDS pair table (decided ONCE per launch,
      constexpr int slot = decltype(slot_c)::value;              // outside the stream: as a run-time test it sat in front of every dword of every record)
      constexpr bool TBLC = decltype(tbl_c)::value;
      constexpr int q = slot % NQ;
      if constexpr (NS_GVS_ABL == 3) return;
      const _Float16* abase = a_lds + srel * KSTEP + aoff;
      Corr cr;
      {
        typedef __attribute__((address_space(3))) const uint32_t* L32;
        const uint32_t ca = ring_corr + uint32_t(slot) * SLOT;
        if constexpr (SBYTES == 2) {
          cr.s[0] = *reinterpret_cast<__attribute__((address_space(3))) const uint16_t*>(ca);
        } else {
#pragma unroll
          for (int t = 0; t < Corr::NW32; t++) cr.s[t] = reinterpret_cast<L32>(ca)[t];
        }
        if constexpr (ASYM) {
          const uint32_t za = uint32_t(reinterpret_cast<uintptr_t>(ring)) + uint32_t(slot) * SLOT + 1024u + 16u * SBYTES + uint32_t(nn) * SPS;
          if constexpr (SPS == 4)
            cr.z[0] = *reinterpret_cast<L32>(za);
          else if constexpr (SPS == 2)
            cr.z[0] = *reinterpret_cast<__attribute__((address_space(3))) const uint16_t*>(za);
          else
            cr.z[0] = *reinterpret_cast<__attribute__((address_space(3))) const uint8_t*>(za);
        }
      }
      float sc[4], zp[4];
      corr_decode<SPS, SK, ASYM, NJ>(cr, sc, zp);
      const uint4v qvv = *reinterpret_cast<const __attribute__((address_space(3))) uint4v*>(ring_lane + uint32_t(slot) * SLOT);
      const uint32_t xw[4] = {qvv.x, qvv.y, qvv.z, qvv.w};
      half8_t bq[NJ];
#pragma unroll
      for (int j = 0; j < NJ; j++) {
        if constexpr (NS_GVS_ABL == 1) {
          const uint4v raw = {xw[j], xw[(j + 1) & 3], xw[(j + 2) & 3], xw[(j + 3) & 3]};
          bq[j] = __builtin_bit_cast(half8_t, raw);
        } else if constexpr (KIND == WK_INT4) {
          const _Float16 zl = (_Float16)(-1032.f - zp[j]), zh = (_Float16)(-72.f - zp[j]);
          bq[j] = cvt_i4x8(xw[j], i4c, half2_t{zl, zl}, half2_t{zh, zh});
        } else if constexpr (KIND == WK_INT8) {
          const _Float16 zo8 = (_Float16)(-1152.f - zp[j]);
          bq[j] = cvt_i8x8(xw[2 * j], xw[2 * j + 1], half2_t{zo8, zo8});
        } else if constexpr (KIND == WK_F8) {
          bq[j] = cvt_f8x8(xw[2 * j], xw[2 * j + 1], p.f8);
        } else if constexpr (TBLK && TBLC) {
          // byte b of the word = codes (i0,i2) (i4,i6) (i1,i3) (i5,i7): entry b of this lane's table copy = their two values
          typedef __attribute__((address_space(3))) const uint32_t* L32;
          uint4v r;
          r.x = *reinterpret_cast<L32>(__builtin_amdgcn_ubfe(xw[j], 0, 8) * 128u + tbl_lane);
          r.y = *reinterpret_cast<L32>(__builtin_amdgcn_ubfe(xw[j], 8, 8) * 128u + tbl_lane);
          r.z = *reinterpret_cast<L32>(__builtin_amdgcn_ubfe(xw[j], 16, 8) * 128u + tbl_lane);
          r.w = *reinterpret_cast<L32>((xw[j] >> 24) * 128u + tbl_lane);
          bq[j] = __builtin_bit_cast(half8_t, r);
        } else {
          bq[j] = cvt_f4x8(xw[j], p.lut);
        }
      }
      if constexpr (SPS == 1 && NS_GVS_ABL == 0) {
        // one scale for the whole record: the NJ products are chained through the accumulator, one scaling per record
        floatx4 dsum = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NJ; j++) {
          const half8_t afrag = *reinterpret_cast<const half8_t*>(abase + 32 * j);
          dsum = __builtin_amdgcn_mfma_f32_16x16x32_f16(afrag, bq[j], dsum, 0, 0, 0);
        }
        acc[q] += dsum * sc[0];
        return;
      }
      floatx4 dd[NJ];
#pragma unroll
      for (int j = 0; j < NJ; j++) {
        half8_t afrag;
        if constexpr (NS_GVS_ABL == 2) {
          const uint4v raw = {uint32_t(srel) | 0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
          afrag = __builtin_bit_cast(half8_t, raw);
        } else {
          afrag = *reinterpret_cast<const half8_t*>(abase + 32 * j);
        }
        dd[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(afrag, bq[j], floatx4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < NJ; j++) acc[q] += dd[j] * sc[j];
    };

    // end of a tile: this wave's partial sums -> LDS slot, counter + 1.  All by hand: for a visible LDS store hipcc would
    // first wait for every LDS-DMA request in flight (the whole ring)
    const uint32_t ctl = lds0 + p.ctl_off;
    const uint32_t red_lane = lds0 + p.red_off + w * p.red_wave + uint32_t(l) * 16u;
    uint32_t jc = j0, kc = s0;  // compute side: tile ordinal, k-step inside the slice
    auto flush = [&]() {
      const uint32_t slot_r = jc & uint32_t(kGsR - 1);
      if (jc >= uint32_t(kGsR)) {  // the slot's previous tile (jc - kGsR) must have been taken by the service wave
        uint32_t dv, spins = 0;
        do {  // (bounded: a broken hand-back must end as a wrong result the tests see, not as a hung GPU)
          asm volatile("ds_read_b32 %0, %1 offset:16\n\ts_waitcnt lgkmcnt(0)" : "=v"(dv) : "v"(ctl) : "memory");
          dv = __builtin_amdgcn_readfirstlane(dv);
          if (dv + uint32_t(kGsR) <= jc) __builtin_amdgcn_s_sleep(1);
        } while (dv + uint32_t(kGsR) <= jc && ++spins < kGsSpinLimit);
      }
      if (g < rows_q) {
#pragma unroll
        for (int q = 0; q < NQ; q++)
          asm volatile("ds_write_b128 %0, %1" ::"v"(red_lane + slot_r * p.red_slot + uint32_t(q) * uint32_t(rows_q) * 256u), "v"(acc[q])
                       : "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (l == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(ctl + slot_r * 4u), "v"(1u) : "memory");
#pragma unroll
      for (int q = 0; q < NQ; q++) acc[q] = floatx4{0.f, 0.f, 0.f, 0.f};
    };

    // ---- 4. stream: consume the oldest record, refill its slot with the record PF items ahead; tiles follow each
    //      other without a pause ----
    auto stream = [&](auto tbl_c) {
    for (uint32_t t0 = 0; t0 < total; t0 += PF) {
      NS_FOR_SLOTS({
        const uint32_t t = t0 + i;
        if (t < total) {
          wait_records(min(total - t - 1u, uint32_t(PF - 1)));
          compute(ic, tbl_c, kc);
          if constexpr (i % NQ == NQ - 1) {
            kc += NS;
            if (kc >= nks || t + 1u == total) {  // the wave's next unit lies in another tile (or there is none): hand over this tile's sums
              flush();
#ifdef NS_TRACE
              if (jc == j0) NS_SSTAMP(4);
#endif
              while (kc >= nks) {
                kc -= nks;
                jc++;
              }
            }
          }
          __builtin_amdgcn_sched_barrier(0);
          if (t + PF < total) issue(ic);
          __builtin_amdgcn_sched_barrier(0);
        }
      })
    }
    };
    if constexpr (TBLK) {
      if (TBL) stream(std::true_type{});
      else stream(std::false_type{});
    } else {
      stream(std::false_type{});
    }
#undef NS_FOR_SLOTS
    NS_SSTAMP(5);
  } else {
    // =========================================== service wave ===========================================
    // one serial instruction stream beside three or four streaming waves on its SIMD: at equal priority it gets a quarter of
    // the issue slots and needed 2.5 us per fused gate/up tile of a 3.5 us tile period (profiles/r04l_trace_warm.txt)
    __builtin_amdgcn_s_setprio(3);
    // the LDS control words start at zero: written here, in front of the barrier every streaming wave passes before its
    // first flush
    if (l < int(kGsCtlBytes / 4)) asm volatile("ds_write_b32 %0, %1" ::"v"(lds0 + p.ctl_off + uint32_t(l) * 4u), "v"(0u) : "memory");
    stage_a();
    if (p.tbl_img) stage_table(); else build_table();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    NS_SSTAMP(2);
    shuffle_own_a();
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    NS_SSTAMP(3);
    // tile ordinal j's units are j * nks .. j * nks + nks - 1: its contributors are the waves (j * nks + i) % NS, i < active
    const uint32_t active = min(NS, nks);
    const uint32_t cstep = nks % NS;
    uint32_t c0 = 0;  // (j * nks) % NS
    const GvsKArgs cold = gvs_late_args();
    const int ldc = cold->ldc, ldd = cold->ldd, epi = cold->epilogue;
    const float* dptr = cold->d;
    float* c2 = cold->c2;
    float* slab = cold->slab;
    uint32_t* tickets = cold->tickets;
    const Rsrc rslab = make_rsrc(slab, cold->slab_bytes);
    const uint32_t nslices = 1u << p.s_log2;
    const uint32_t ctl = lds0 + p.ctl_off;
    const uint32_t red0 = lds0 + p.red_off + uint32_t(l) * 16u;
    typedef const __attribute__((address_space(3))) floatx4* LdsF4;  // (re-read every tile: the poll below is a compiler barrier)
    for (uint32_t jt = 0; jt < ntw; jt++) {
      const uint32_t jord = tile_ord(jt);
      const uint32_t T = tg + jord * p.tg_count;
      const bool owner = own_last && (jord & S1) == sl;
      uint32_t tl = T;
      int sg = 0;
      if constexpr (MSEG) {
        sg = int(T >= p.tb1) + int(T >= p.tb2);
        tl = T - (sg == 0 ? 0u : (sg == 1 ? p.tb1 : p.tb2));
      }
      const auto* mp = &cold->mat[MSEG ? sg : 0];
      const int ncols = mp->n;
      float* cbase = mp->c;
      _Float16* c16 = mp->c16;
      const int col = int(tl) * 16 + nn;
      const bool col_ok = col < ncols;
      // what the epilogue has to fetch is requested before the wait for the tile's partial sums
      float dvp[4] = {0.f, 0.f, 0.f, 0.f};
      if constexpr (!DUAL) {
        if (dptr && col_ok && (!slab || owner)) {
#pragma unroll
          for (int rr = 0; rr < 4; rr++)
            if (4 * g + rr < p.m) dvp[rr] = dptr[size_t(4 * g + rr) * ldd + col];
        }
      }
      const uint32_t slot_r = jt & uint32_t(kGsR - 1);
      {
        uint32_t cv, spins = 0;
        do {
          asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(cv) : "v"(ctl + slot_r * 4u) : "memory");
          cv = __builtin_amdgcn_readfirstlane(cv);
          if (cv < active) __builtin_amdgcn_s_sleep(1);
        } while (cv < active && ++spins < kGsSpinLimit);
      }
      floatx4 sum[NQ];
#pragma unroll
      for (int q = 0; q < NQ; q++) sum[q] = floatx4{0.f, 0.f, 0.f, 0.f};
      if (g < rows_q) {
        // the contributors' partial sums in unit order, four contributors' loads in flight at a time
        constexpr int RB = 4;
        uint32_t ww = c0;
        for (uint32_t ci = 0; ci < active; ci += RB) {
          floatx4 v[RB][NQ];
#pragma unroll
          for (int b = 0; b < RB; b++) {
            const bool on = ci + uint32_t(b) < active;
#pragma unroll
            for (int q = 0; q < NQ; q++)
              v[b][q] = on ? *reinterpret_cast<LdsF4>(red0 + slot_r * p.red_slot + ww * p.red_wave + uint32_t(q) * uint32_t(rows_q) * 256u)
                           : floatx4{0.f, 0.f, 0.f, 0.f};
            ww = (ww + 1u == NS ? 0u : ww + 1u);
          }
#pragma unroll
          for (int b = 0; b < RB; b++)
#pragma unroll
            for (int q = 0; q < NQ; q++) sum[q] += v[b][q];
        }
      }
      c0 += cstep;
      if (c0 >= NS) c0 -= NS;
      // hand the slot back: counter to zero first, then the "done" ordinal the streaming waves wait for
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (l == 0) {
        asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\tds_write_b32 %2, %3 offset:16" ::"v"(ctl + slot_r * 4u), "v"(0u), "v"(ctl),
                     "v"(jt + 1u)
                     : "memory");
      }
      if (slab && !owner) {
        // split-K: the raw partial sums of (slice, tile).  With tickets: write-through stores (visible to the owner's CU
        // whatever XCD it is on), drained, then the ticket — the owner adds the slices in slice order; without: plain stores,
        // gemvs_finalize_kernel adds them after the launch
        if (g < rows_q) {
#pragma unroll
          for (int q = 0; q < NQ; q++) {
            const uint32_t off = uint32_t(((size_t(sl) * p.ntiles + T) * NQ + q) * 1024u + uint32_t(l) * 16u);
            if (tickets)
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4v, sum[q]), rslab, off, 0, 17 /* sc0 sc1: system scope, write-through */);
            else
              *reinterpret_cast<floatx4*>(reinterpret_cast<unsigned char*>(slab) + off) = sum[q];
          }
        }
        if (tickets) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (l == 0) __hip_atomic_fetch_add(tickets + T, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      } else if (slab) {
        // the owner: every other slice's ticket (drawn long ago: owned tiles are streamed last), then their sums past the
        // caches, added in slice order with this workgroup's own in its place
        uint32_t seen = 0, spins = 0;
        do {
          seen = __hip_atomic_load(tickets + T, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          seen = __builtin_amdgcn_readfirstlane(seen);
          if (seen < nslices - 1u) __builtin_amdgcn_s_sleep(2);
        } while (seen < nslices - 1u && ++spins < kGsSpinLimit);
        // a slice never arrived (a grid larger than what is resident at once), now or in an earlier launch on this ticket: no
        // sum of whatever the slab holds — the tile's outputs become NaN and the ticket stays poisoned (high bit; late
        // arrivals only add to it), so that every later launch on this scratch fails as loudly
        const bool expired = seen < nslices - 1u || (seen & 0x80000000u) != 0u;
        floatx4 tot[NQ];
#pragma unroll
        for (int q = 0; q < NQ; q++) tot[q] = floatx4{0.f, 0.f, 0.f, 0.f};
        if (expired) {
          const float qnan = __builtin_nanf("");
#pragma unroll
          for (int q = 0; q < NQ; q++) tot[q] = floatx4{qnan, qnan, qnan, qnan};
        } else if (g < rows_q) {
          for (uint32_t sx = 0; sx < nslices; sx++) {
#pragma unroll
            for (int q = 0; q < NQ; q++) {
              floatx4 v = sum[q];
              if (sx != sl) {
                const uint32_t off = uint32_t(((size_t(sx) * p.ntiles + T) * NQ + q) * 1024u + uint32_t(l) * 16u);
                v = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rslab, off, 0, 17 /* sc0 sc1: past every cache */));
              }
              tot[q] += v;
            }
          }
        }
        if (l == 0) __hip_atomic_store(tickets + T, expired ? 0x80000000u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // zero again for the next launch
        gvs_epilogue<DUAL>(tot, p.m, col, col_ok, g, cbase, c16, ldc, dvp, c2, epi);
      } else {
        gvs_epilogue<DUAL>(sum, p.m, col, col_ok, g, cbase, c16, ldc, dvp, c2, epi);
      }
#ifdef NS_TRACE
      if (jt == 0) NS_SSTAMP(4);
#endif
    }
    NS_SSTAMP(5);
  }
#ifdef NS_TRACE
  if ((threadIdx.x & 63) == 0 && blockIdx.x < 4096)
    p.trace[(size_t(blockIdx.x) * 16 + (threadIdx.x >> 6)) * 8 + 7] =
        (uint64_t(__builtin_amdgcn_s_getreg((31 << 11) | 20)) << 32) | uint32_t(__builtin_amdgcn_s_getreg((31 << 11) | 4));
#endif
}

// split-K, second pass: one wave per tile unit adds the slices' partial sums in slice order and applies the epilogue
struct GvsFinParams {
  const float* slab;
  uint32_t slices, ntiles, nq;
  uint32_t tb1, tb2;
  int m;
  GvsMat mat[3];
  float* c2;
  const float* d;
  int ldc, ldd, epilogue;
};
template <int MODE>
__global__ __launch_bounds__(256) void gemvs_finalize_kernel(const GvsFinParams p) {
  constexpr bool DUAL = MODE == GS_DUAL, MSEG = MODE == GS_MSEG;
  constexpr int NQ = DUAL ? 2 : 1;
  const int l = threadIdx.x & 63;
  const uint32_t T = blockIdx.x * 4u + (threadIdx.x >> 6);
  if (T >= p.ntiles) return;
  const int nn = l & 15, g = l >> 4;
  if (4 * g >= p.m) return;
  uint32_t tl = T;
  int sg = 0;
  if constexpr (MSEG) {
    sg = int(T >= p.tb1) + int(T >= p.tb2);
    tl = T - (sg == 0 ? 0u : (sg == 1 ? p.tb1 : p.tb2));
  }
  const GvsMat& mp = p.mat[MSEG ? sg : 0];
  const int col = int(tl) * 16 + nn;
  const bool col_ok = col < mp.n;
  float dvp[4] = {0.f, 0.f, 0.f, 0.f};
  if constexpr (!DUAL) {
    if (p.d && col_ok) {
#pragma unroll
      for (int rr = 0; rr < 4; rr++)
        if (4 * g + rr < p.m) dvp[rr] = p.d[size_t(4 * g + rr) * p.ldd + col];
    }
  }
  floatx4 sum[NQ];
#pragma unroll
  for (int q = 0; q < NQ; q++) {
    sum[q] = floatx4{0.f, 0.f, 0.f, 0.f};
    for (uint32_t s = 0; s < p.slices; s++)
      sum[q] += *reinterpret_cast<const floatx4*>(p.slab + ((size_t(s) * p.ntiles + T) * NQ + q) * 256 + size_t(l) * 4);
  }
  gvs_epilogue<DUAL>(sum, p.m, col, col_ok, g, mp.c, mp.c16, p.ldc, dvp, p.c2, p.epilogue);
}

// ============================================================================================================
// host side
// ============================================================================================================
template <int KIND, int SPS, int SK, bool ASYM>
static hipError_t launch_gvs_k(const GvsParams& p, int mode, int grid, int nwaves, size_t lds, hipStream_t st) {
  const dim3 g(grid), b(nwaves * 64);
#define NS_GS_LAUNCH(MODEV)                                                                                     \
  {                                                                                                             \
    auto k = gemvs_kernel<KIND, SPS, SK, ASYM, MODEV>;                                                          \
    static const hipError_t attr =                                                                              \
        hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, int(kGsMaxLds)); \
    if (attr != hipSuccess && lds > 64 * 1024) return attr;                                                     \
    hipLaunchKernelGGL(k, g, b, lds, st, p);                                                                    \
  }
  if (mode == GS_DUAL)
    NS_GS_LAUNCH(GS_DUAL)
  else if (mode == GS_MSEG)
    NS_GS_LAUNCH(GS_MSEG)
  else
    NS_GS_LAUNCH(GS_PLAIN)
#undef NS_GS_LAUNCH
  return hipGetLastError();
}
template <int KIND, int SPS, int SK>
static hipError_t launch_gvs_a(const GvsParams& p, bool asym, int mode, int grid, int nwaves, size_t lds, hipStream_t st) {
  if constexpr (KIND == WK_F4 || KIND == WK_F8) {
    (void)asym;
    return launch_gvs_k<KIND, SPS, SK, false>(p, mode, grid, nwaves, lds, st);
  } else {
    if (asym) return launch_gvs_k<KIND, SPS, SK, true>(p, mode, grid, nwaves, lds, st);
    return launch_gvs_k<KIND, SPS, SK, false>(p, mode, grid, nwaves, lds, st);
  }
}
template <int KIND, int SPS>
static hipError_t launch_gvs_s(const GvsParams& p, uint32_t scale_dt, bool asym, int mode, int grid, int nwaves, size_t lds,
                               hipStream_t st) {
  if (scale_dt == DT_F32) return launch_gvs_a<KIND, SPS, SK_F32>(p, asym, mode, grid, nwaves, lds, st);
  if (scale_dt == DT_F16) return launch_gvs_a<KIND, SPS, SK_F16>(p, asym, mode, grid, nwaves, lds, st);
  return launch_gvs_a<KIND, SPS, SK_BF16>(p, asym, mode, grid, nwaves, lds, st);
}

// tuning / diagnostics (ns_hip_set_tuning): "gvs" 0 = off, 1 = from 2 rows (default), 2 = from 1 row;
// "gvs_slices" / "gvs_waves" / "gvs_grid" force the decomposition (0 = by shape)
static std::atomic<int> g_gvs_mode{-1};
static std::atomic<int> g_gvs_slices{-1}, g_gvs_waves{-1}, g_gvs_grid{-1}, g_gvs_tbl{-2}, g_gvs_fin{-1};
static int env_or(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}
void set_gemvs_tuning(int what, int value) {
  (what == 0 ? g_gvs_mode : what == 1 ? g_gvs_slices : what == 2 ? g_gvs_waves : what == 3 ? g_gvs_grid : what == 4 ? g_gvs_tbl : g_gvs_fin).store(value);
}
static int gvs_knob(std::atomic<int>& k, const char* env, int dflt) {
  int v = k.load();
  if (v < (dflt < 0 ? -1 : 0)) {
    v = env_or(env, dflt);
    k.store(v);
  }
  return v;
}
static int gvs_cus() {
  static const int cus = [] {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    return prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }();
  return cus;
}

struct GvsPlan {
  int slices_log2 = 0, ns = 8, tg = 0;
  size_t a_bytes = 0, lds = 0;
  bool tbl = false;
  uint32_t ksl = 0, row_stride = 0, red_wave = 0, red_slot = 0, ring_stride = 0;
  double cost = 0;
};

// hipErrorNotSupported: outside the kernel's envelope or not the better decomposition — the caller goes on to gemv_kernel
hipError_t launch_gemvs(const SmallMArgs& a, hipStream_t st) {
  const int mode_knob = gvs_knob(g_gvs_mode, "NS_GVS", 1);  // 0 off, 1 by shape (below), 2 always from one row, 3 always from two rows
  if (mode_knob == 0 || a.m < (mode_knob == 2 ? 1 : 2) || a.m > kGsMaxRows) return hipErrorNotSupported;
  if (a.link || a.rope || a.i8 || a.moe) return hipErrorNotSupported;  // carried norm / fused RoPE / int8-reference numerics: gemv_kernel
  const ns_weight* w0 = a.seg[0].w;
  const int nmat = a.nseg;
  if (nmat < 1 || nmat > 3 || (a.dual && nmat != 2)) return hipErrorNotSupported;
  // fp16 activations with 16-byte aligned rows; several rows need K to fill whole k-steps (a row's padding columns
  // would otherwise read the next row through the descriptor).  A caller without the fp16 shadow (the reference's device
  // graph hands over fp32 tensors) gets one conversion pass into scratch first (round to nearest even, what the shadow's
  // producers do: bit-identical results either way, tests/test_gpu_fullsize.py)
  const int kstep = w0->kstep_len;
  const bool shadow = a.a16 != nullptr;
  if (shadow ? ((a.lda & 7) || (reinterpret_cast<uintptr_t>(a.a16) & 15)) : (a.a == nullptr)) return hipErrorNotSupported;
  if ((w0->k & 7) || (a.m > 1 && w0->k % kstep != 0)) return hipErrorNotSupported;
  const int lda16 = shadow ? a.lda : w0->k;
  if (uint64_t(a.m) * uint64_t(lda16) * 2 >= (uint64_t(1) << 30)) return hipErrorNotSupported;

  GvsParams p;
  memset(&p, 0, sizeof(p));
  uint32_t tiles = 0;
  uint32_t tbeg[3] = {0, 0xffffffffu, 0xffffffffu};
  const uint8_t* wb[3] = {nullptr, nullptr, nullptr};
  uint32_t soff[3] = {0, 0, 0}, zoff[3] = {0, 0, 0};
  for (int i = 0; i < nmat; i++) {
    const ns_weight* w = a.seg[i].w;
    if (!w->single_span || w->alloc_bytes >= (size_t(1) << 31)) return hipErrorNotSupported;
    if (w->kind != w0->kind || w->sps != w0->sps || w->scale_dt != w0->scale_dt || w->asym != w0->asym || w->k != w0->k ||
        w->ksteps != w0->ksteps || w->qstride != w0->qstride || w->sstride != w0->sstride || w->zstride != w0->zstride ||
        w->srows != w0->srows || w->blocksize != w0->blocksize)
      return hipErrorNotSupported;
    if (a.dual && w->ntiles != w0->ntiles) return hipErrorNotSupported;
    wb[i] = reinterpret_cast<const uint8_t*>(w->codes);
    soff[i] = uint32_t(reinterpret_cast<const uint8_t*>(w->scales) - wb[i]);
    zoff[i] = w->zps ? uint32_t(reinterpret_cast<const uint8_t*>(w->zps) - wb[i]) : 0u;
    tbeg[i] = a.dual ? 0u : tiles;
    if (!a.dual || i == 0) tiles += uint32_t(w->ntiles);
    p.mat[i] = GvsMat{wb[i], soff[i], zoff[i], tbeg[i], w->n, a.seg[i].c, static_cast<_Float16*>(a.seg[i].c16)};
  }
  const bool mseg = !a.dual && nmat > 1;
  const int mode = a.dual ? GS_DUAL : (mseg ? GS_MSEG : GS_PLAIN);
  const uint32_t ks = uint32_t(w0->ksteps);
  if (tiles == 0 || ks == 0) return hipErrorNotSupported;

  const int rows = a.m;
  const int rows_q = (rows + 3) / 4;
  const int nq = a.dual ? 2 : 1;
  if (mode_knob == 1) {
    // which kernel by shape (profiles/r04n_rows.txt, us, one-tile-per-workgroup gemv_kernel -> this kernel).  This kernel's time
    // hardly depends on the row count, gemv_kernel's grows with the activations every tile stages; gemv_kernel's prologue is
    // about 0.7 us leaner, so short launches of a few rows stay there:
    //   4096 x 4096 int4      2 rows 5.6 -> 6.2    6 rows 6.1 -> 6.8    12 rows 8.5 -> 7.4   16 rows 9.0 -> 7.4
    //   gate/up 11008 int4    2 rows 13.5 -> 15.2  4 rows 15.3 -> 16.2  6 rows 19.0 -> 16.7  12 rows 20.2 -> 17.6
    //   gate/up 14336 nf4     2 rows 22.7 -> 20.8  6 rows 27.4 -> 21.4  16 rows 39.2 -> 28.7
    //   4096 x 11008 int4     2 rows 8.7 -> 9.0    3 rows 11.5 -> 9.3 (66 KB of activations: beyond gemv_kernel)   16 rows 24.8 -> 14.3
    const bool beyond_gemv = size_t(rows) * (size_t(ks) * kstep + 8) * 2 > 64 * 1024;  // gemv_kernel's activation envelope
    const bool wide = a.dual || tiles >= 512;
    const bool f4_wide = w0->kind == WK_F4 && wide;
    if (!(beyond_gemv || f4_wide || (rows >= 5 && wide) || rows >= 9)) return hipErrorNotSupported;
  }
  const uint32_t sbytes = uint32_t(w0->sps) * (w0->scale_dt == DT_F32 ? 4u : 2u);
  const uint32_t slot = 1024u + 16u * sbytes + (w0->asym ? 16u * uint32_t(w0->sps) : 0u);
  const int cus = gvs_cus();
  const int force_s = gvs_knob(g_gvs_slices, "NS_GVS_SLICES", 0);
  const int force_ns = gvs_knob(g_gvs_waves, "NS_GVS_WAVES", 0);
  const int force_grid = gvs_knob(g_gvs_grid, "NS_GVS_GRID", 0);
  const int force_tbl = gvs_knob(g_gvs_tbl, "NS_GVS_TABLE", -1);  // f4 pair table: -1 by size, 0 never, 1 always
  const int max_wg = force_grid > 0 ? force_grid : cus;

  // ---- the decomposition: slices S (power of two), tile groups, streaming waves ----
  // cost of a candidate, in bytes a workgroup moves (the launch ends with its slowest workgroup): its weight records +
  // half its activation slice (L2, about twice the HBM stream's rate per CU) + what the finalize launch of a split costs
  auto plan_for = [&](int s_log2, GvsPlan* out) -> bool {
    const uint32_t S = 1u << s_log2;
    if (S > ks || int(S) > max_wg) return false;
    GvsPlan pl;
    pl.slices_log2 = s_log2;
    pl.ksl = (ks + S - 1) / S;
    if (uint64_t(pl.ksl) * (S - 1) >= ks) return false;  // an empty last slice
    pl.row_stride = pl.ksl * uint32_t(kstep) + 8;
    pl.a_bytes = (size_t(rows) * pl.row_stride * 2 + 15) & ~size_t(15);
    pl.tg = int(std::min<uint32_t>(tiles, uint32_t(max_wg) / S));
    if (pl.tg < 1) return false;
    // f4: the pair table (32 KB of LDS, ~1.5 us to build and to shuffle the activations for) pays from about a hundred
    // records per workgroup on (4096 x 4096 at 8 rows, 32 records: 8.6 us with it, 8.1 without; 4096 x 14336 in 4 slices, 112
    // records: 16.4 vs 17.8; fused gate/up 14336, 256 records: 22.9 vs 27.9 — profiles/r04h_ablation.txt, r04y_table_threshold.txt)
    const uint32_t tiles_max0 = (tiles + uint32_t(pl.tg) - 1) / uint32_t(pl.tg);
    const bool use_tbl = w0->kind == WK_F4 && force_tbl != 0 && (force_tbl > 0 || uint64_t(tiles_max0) * pl.ksl * nq >= 96u);
    pl.tbl = use_tbl;
    const size_t tbl = use_tbl ? size_t(kGsTblBytes) : 0;
    pl.red_wave = uint32_t(nq) * uint32_t(rows_q) * 256u;
    // streaming waves: as many as LDS holds and the workgroup has units for (units are dealt round-robin: any count
    // balances); the time of a workgroup ~ its units / waves, padded when there are fewer units than waves
    const uint32_t tiles_max = (tiles + uint32_t(pl.tg) - 1) / uint32_t(pl.tg);
    auto ring_of = [&](int ns) {
      const size_t items = (size_t(pl.ksl) * tiles_max + ns - 1) / ns * nq;
      return (std::min<size_t>(items, size_t(kGsPF)) * slot + 15) & ~size_t(15);
    };
    int best_ns = 0;
    for (int ns = kGsMaxNS; ns >= 1; ns--) {
      if (force_ns > 0 && ns != std::min(force_ns, kGsMaxNS)) continue;
      if (force_ns <= 0 && uint32_t(ns) > std::max<uint32_t>(1u, pl.ksl * tiles_max)) continue;
      const size_t need = tbl + pl.a_bytes + kGsCtlBytes + size_t(kGsR) * ns * pl.red_wave + size_t(ns) * ring_of(ns);
      if (need > kGsMaxLds) continue;
      best_ns = ns;
      break;
    }
    const double best_t = best_ns ? double(pl.ksl) * (1.0 + 0.5 * std::max(0, 12 - best_ns) / 12.0) : 0.0;
    if (best_ns == 0) return false;
    pl.ns = best_ns;
    pl.ring_stride = uint32_t(ring_of(pl.ns));
    pl.red_slot = uint32_t(pl.ns) * pl.red_wave;
    pl.lds = tbl + pl.a_bytes + kGsCtlBytes + size_t(kGsR) * pl.red_slot + size_t(pl.ns) * pl.ring_stride;
    // + a fixed cost per tile of a workgroup (flush, reduction, epilogue or slab store: about 8 KB of streaming each: the service wave needs ~0.8 us per tile)
    pl.cost = double(tiles_max) * (best_t * nq * slot + 8e3) + 0.5 * double(pl.a_bytes) + (S > 1 ? 75e3 : 0.0);
    *out = pl;
    return true;
  };
  GvsPlan best;
  bool have = false;
  for (int s_log2 = 0; s_log2 <= 5; s_log2++) {
    if (force_s > 0 && (1 << s_log2) != force_s) continue;
    GvsPlan pl;
    if (!plan_for(s_log2, &pl)) continue;
    if (!have || pl.cost < best.cost) best = pl, have = true;
  }
  if (!have) return hipErrorNotSupported;
  const uint32_t S = 1u << best.slices_log2;

  float* slab = nullptr;
  uint32_t* tickets = nullptr;
  const size_t slab_bytes = size_t(S) * tiles * nq * 256 * sizeof(float);
  if (S > 1) {
    if (slab_bytes >= (size_t(1) << 31)) return hipErrorNotSupported;
    slab = static_cast<float*>(stream_scratch(st, slab_bytes, 2));
    if (!slab) return hipErrorNotSupported;
    // Who adds the slices: a second launch (gemvs_finalize_kernel, the default) or — "gvs_finalize" 0 — the tile's owner inside
    // the launch, behind one self-resetting ticket per tile unit.  Measured equal (profiles/r04u_split_k_finish.txt: 4096 x 14336
    // at 8 rows 17.1 vs 17.3 us with 2 slices, 17.8 vs 17.8 with 4): what the in-launch finish saves in launch boundary the owner
    // spends reading the other slices past the caches; it also needs every workgroup of the launch resident at once (an owner
    // spins on tickets other workgroups draw), which a second launch does not — hence the default.
    static const int kTicketCap = 1 << 16;
    const int force_fin = gvs_knob(g_gvs_fin, "NS_GVS_FINALIZE", 1);
    if (!force_fin && tiles <= uint32_t(kTicketCap)) tickets = static_cast<uint32_t*>(stream_scratch_zeroed(st, size_t(kTicketCap) * 4, 21));
  }

  const void* a16 = a.a16;
  if (!shadow) {
    void* sc = stream_scratch(st, size_t(a.m) * w0->k * 2, 0);
    if (!sc) return hipErrorNotSupported;
    const hipError_t ce = launch_cvt_a16(a.a, sc, a.m, w0->k, a.lda, w0->k, st);
    if (ce != hipSuccess) return ce;
    a16 = sc;
  }
  p.wb0 = wb[0], p.wb1 = wb[1], p.wb2 = wb[2];
  p.a = a16;
  p.so0 = soff[0], p.so1 = soff[1], p.so2 = soff[2];
  p.zo0 = zoff[0], p.zo1 = zoff[1], p.zo2 = zoff[2];
  p.ks = ks;
  p.ksl = best.ksl;
  p.s_log2 = uint32_t(best.slices_log2);
  p.qstride = w0->qstride, p.sstride = w0->sstride, p.zstride = w0->zstride;
  p.srows = uint32_t(w0->srows);
  {
    int mul, shift;
    if (!srow_params(w0, &mul, &shift)) return hipErrorNotSupported;
    p.srow_mul = uint32_t(mul), p.srow_shift = uint32_t(shift);
  }
  p.tb1 = mseg ? tbeg[1] : 0xffffffffu;
  p.tb2 = (mseg && nmat > 2) ? tbeg[2] : 0xffffffffu;
  p.ntiles = tiles;
  p.tg_count = uint32_t(best.tg);
  p.tiles_base = tiles / uint32_t(best.tg);
  p.tiles_rem = tiles % uint32_t(best.tg);
  p.ns = uint32_t(best.ns);
  p.m = a.m, p.k = w0->k, p.lda = lda16;
  p.row_stride = best.row_stride;
  p.a_off = best.tbl ? kGsTblBytes : 0u;
  p.a_bytes = uint32_t(best.a_bytes);
  p.ctl_off = p.a_off + uint32_t(best.a_bytes);
  p.red_off = p.ctl_off + kGsCtlBytes;
  p.ring_off = p.red_off + uint32_t(kGsR) * best.red_slot;
  p.ring_stride = best.ring_stride;
  p.red_wave = best.red_wave, p.red_slot = best.red_slot;
  p.c2 = a.c2, p.d = a.d, p.ldc = a.ldc, p.ldd = a.ldd, p.epilogue = a.epilogue;
  p.slab = slab;
  p.tickets = tickets;
  p.slab_bytes = uint32_t(slab_bytes);
  if (w0->kind == WK_F4) {
    f4_lut_planes(w0->lut, &p.lut);
    for (int i = 0; i < 8; i++)
      p.lut16[i] = uint32_t(__builtin_bit_cast(unsigned short, w0->lut[2 * i])) | (uint32_t(__builtin_bit_cast(unsigned short, w0->lut[2 * i + 1])) << 16);
    p.tbl_img = best.tbl ? gvs_f4_table_image(w0->lut, st) : nullptr;
  }
  p.f8 = f8_consts(w0->qtype);
#ifdef NS_TRACE
  p.trace = trace_buffer();
#endif
  const size_t lds = best.lds;
  if (lds > kGsMaxLds) return hipErrorNotSupported;
  const int grid = best.tg * int(S);
  const int nwaves = best.ns + 1;
  static const bool dbg = getenv("NS_GVS_DEBUG") != nullptr;
  if (dbg)
    fprintf(stderr, "gemvs: m %d tiles %u ks %u mode %d -> slices %u (ksl %u) groups %d waves %d lds %zu (A %zu) cost %.0f\n", a.m, tiles, ks,
            mode, S, best.ksl, best.tg, best.ns, lds, best.a_bytes, best.cost);
  if (dbg && w0->kind == WK_F4) fprintf(stderr, "gemvs: f4 pair table %s\n", best.tbl ? "on" : "off");

  hipError_t e = hipSuccess;
#define NS_DISPATCH(KIND)                                                                          \
  switch (w0->sps) {                                                                               \
    case 4: e = launch_gvs_s<KIND, 4>(p, w0->scale_dt, w0->asym, mode, grid, nwaves, lds, st); break;  \
    case 2: e = launch_gvs_s<KIND, 2>(p, w0->scale_dt, w0->asym, mode, grid, nwaves, lds, st); break;  \
    default: e = launch_gvs_s<KIND, 1>(p, w0->scale_dt, w0->asym, mode, grid, nwaves, lds, st); break; \
  }
  if (w0->kind == WK_INT4) {
    NS_DISPATCH(WK_INT4)
  } else if (w0->kind == WK_INT8) {
    if (w0->sps == 2) e = launch_gvs_s<WK_INT8, 2>(p, w0->scale_dt, w0->asym, mode, grid, nwaves, lds, st);
    else e = launch_gvs_s<WK_INT8, 1>(p, w0->scale_dt, w0->asym, mode, grid, nwaves, lds, st);
  } else if (w0->kind == WK_F8) {
    if (w0->sps == 2) e = launch_gvs_a<WK_F8, 2, SK_F32>(p, false, mode, grid, nwaves, lds, st);
    else e = launch_gvs_a<WK_F8, 1, SK_F32>(p, false, mode, grid, nwaves, lds, st);
  } else {
    NS_DISPATCH(WK_F4)
  }
#undef NS_DISPATCH
  if (e != hipSuccess || S == 1 || tickets) return e;

  GvsFinParams f;
  memset(&f, 0, sizeof(f));
  f.slab = slab;
  f.slices = S, f.ntiles = tiles, f.nq = uint32_t(nq);
  f.tb1 = p.tb1, f.tb2 = p.tb2;
  f.m = a.m;
  for (int i = 0; i < 3; i++) f.mat[i] = p.mat[i];
  f.c2 = a.c2, f.d = a.d, f.ldc = a.ldc, f.ldd = a.ldd, f.epilogue = a.epilogue;
  const dim3 fg((tiles + 3) / 4), fb(256);
  if (mode == GS_DUAL)
    hipLaunchKernelGGL(gemvs_finalize_kernel<GS_DUAL>, fg, fb, 0, st, f);
  else if (mode == GS_MSEG)
    hipLaunchKernelGGL(gemvs_finalize_kernel<GS_MSEG>, fg, fb, 0, st, f);
  else
    hipLaunchKernelGGL(gemvs_finalize_kernel<GS_PLAIN>, fg, fb, 0, st, f);
  return hipGetLastError();
}

}  // namespace ns
