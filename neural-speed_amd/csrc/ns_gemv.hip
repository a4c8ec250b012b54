// ns_gemv.hip — gemv_kernel: the decode (M <= 16 rows) weight-streaming kernel of libns_hip.so, second generation.
//
// Same arithmetic as smallm_kernel (ns_kernels.hip): per 1 KiB weight record NJ x v_mfma_f32_16x16x32_f16 on the raw
// codes, the group scale applied to the fp32 MFMA result — exactly w = (code - zp) * scale with fp32 accumulation
// (reference: bestla/bestla/kernel_ref.h:2489-2531 gemv_4bit_fp32_fp32, :1027-1127 decompress_kblock_s4_fp,
// :1456-1478 decompress_kblock_f4_fp), fp16 activations; one workgroup = one 16-column tile over the whole K, waves
// split the k-steps, deterministic cross-wave reduction.  What changed is everything AROUND the arithmetic, because a
// decode launch on MI355X is latency-, not bandwidth-bound:
//
//   * lean prologue.  Every launch starts with cold instruction and scalar caches, so the time to the first weight
//     request is the number of code lines and of dependent kernel-argument fetches in front of it.  smallm_kernel had
//     1.8 KB of code and two argument round trips there (first request 1.3-1.8 us after entry); here ONE batch of scalar
//     loads fetches the 30 words the prologue needs (epilogue arguments are read late, through the argument pointer,
//     so they neither delay the batch nor sit in SGPRs through the loop) and the ring is filled ~0.6 KB into the code.
//   * weight records go HBM -> LDS directly (buffer_load ... lds into the wave's PRIVATE ring: codes, scales and zero
//     points; no VGPRs, no cross-wave synchronisation).  The ring depth is therefore set by LDS, not by the register
//     file: up to 8 records per wave, up to ~140 KiB per CU in flight from the first microsecond of a launch.
//   * waits by hand.  No request of the stream returns to a register, so hipcc inserts no vmcnt waits of its own (it
//     does not order an LDS read behind an LDS-DMA write at all); before a record is consumed the kernel waits for
//     exactly the requests older than the ones that may stay in flight (requests retire in order): s_waitcnt
//     vmcnt(OPS * min(PF - 1, records still to come)).  Nothing is ever requested that is not consumed (out-of-range
//     "dead" requests were measured to cost a full pass through the address pipeline each, profiles/r02d-r02e).
//
// Tried here and dropped (profiles/r02a-r02b): cutting the 688 gate/up tiles into equal K-ranges over the CUs
// ("stream-K") with partial sums handed between workgroups through write-through stores + flags.  The hand-off costs
// 3-5 us under load (the poll queues behind the consumer CU's own weight requests) — more than the 2 us of imbalance it
// removes in an 11 us launch; narrow tensor-parallel shards lost 3 us per launch to it.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <utility>

#include "../../include/ns_bestla.h"
#include "ns_common.h"
#include "ns_route.h"
#include "ns_dev.h"

namespace ns {

#ifndef NS_GV_PF
#define NS_GV_PF 4
#endif
// records each wave keeps in flight (fused gate/up: half of them per matrix).  A CU's memory pipeline holds about
// 50 KiB of requests; beyond that the ISSUE of further requests stalls (profiles/r02h_wave_trace: ring fills of a
// 24-wave CU complete 0.5 ... 6 us after entry), so a deeper ring only delays the other waves' first records:
// 8 deep measured 12 % slower on the whole chain than 4 deep (profiles/r02g_sweep.txt)
constexpr int kGvPF = NS_GV_PF;
constexpr int kGvMaxRows = 16;
constexpr size_t kGvMaxALds = 64 * 1024;    // staged activations (fp16) per workgroup
constexpr size_t kGvMaxLds = 160 * 1024;

// one matrix of a launch as the kernel sees it; a fused QKV launch looks its matrix up BY INDEX in the kernel-argument
// segment (one scalar load) instead of carrying three of everything in SGPRs
struct GemvMat {
  const uint8_t* wbase;  // ONE allocation: records at 0, scales at s_off, zero points at z_off
  uint32_t s_off, z_off;
  uint32_t tile_begin;   // first global tile of this matrix in the launch
  int n;
  float* c;
  _Float16* c16;
};
static_assert(sizeof(GemvMat) == 40, "GemvMat is addressed by index in the kernel-argument segment");

// RoPE of q and k + kv-cache append as the epilogue of a fused QKV launch (ns_qkv_rope): adjacent-pair mode only, so
// a pair always lies inside one 16-column tile
struct GemvRope {
  _Float16* kc;
  _Float16* vc;
  long long c_sl, c_head;  // cache element strides per position / per head
  const float2* cos_sin;   // [row][head_size / 2] (cos, sin) * attn_factor of position n_past + row (ns_hip_rope_cos_sin)
  int head_size, n_past;
  int on;
  // replayed device route (QkvRopeRoute, ns_common.h; one row): k (rotated) and v also go to the reference's fp32 cache cells, and the
  // position follows the captured graph's token counter
  int kd_pos;
  const int* kmove;
  float* k32;
  float* v32;
  long long k32_head, k32_dim, k32_tok, v32_head, v32_dim, v32_tok;
  uint32_t* ovf;
};

// one row of an expert group's device table (ns_moe.hip: MoeExpert — same layout)
struct MoeExpertRow {
  const uint8_t* codes;
  const uint8_t* scales;
  const int8_t* zps;
};

struct GemvParams {
  // ---- hot head: everything the prologue needs, fetched by one batch of scalar loads ----
  const uint8_t* wbase0;    // matrix 0 (and, for the fused gate/up launch, matrix 1)
  const uint8_t* wbase1;
  const void* a;            // activations, fp16 [m][lda]
  uint32_t ks;              // k-steps per tile
  uint32_t qstride;         // bytes per (tile, k-step) record
  uint32_t nw_log2;         // log2(waves per workgroup)
  uint32_t s_off0, s_off1;
  uint32_t sstride;
  uint32_t srows, srow_mul, srow_shift;
  uint32_t tb1, tb2;        // first global tile of matrices 1 and 2 of a fused QKV launch (2^32 - 1: absent)
  int m, k, lda;
  uint32_t row_stride;      // halves per staged row in LDS
  uint32_t ring_off;        // byte offset of the per-wave rings in LDS (the reduction scratch reuses them)
  uint32_t ring_stride;     // bytes of one wave's ring = slots x slot size
  uint32_t z_off0, z_off1, zstride;  // asymmetric formats only: last, so that the rest is one contiguous run of words
  // int8-reference numerics (XV = 3): a = u8 activation codes [m][lda]; i8_corr = [m][nblk] fp32 scales followed by
  // [m][nblk] u8 zero points (one span, staged at ssq_off); k-block of column kk = kk >> i8_bshift
  const uint8_t* i8_corr;
  uint32_t i8_span, i8_nblk, i8_bshift;
  // expert-indexed launch (XV = 4, ns_hip_mul_mat_id at decode size): the weight base is table[*moe_id].codes — every expert of a
  // group has the same shape and layout, so the offsets above hold for all of them; an id outside [0, moe_n) zeroes the row
  const MoeExpertRow* moe_table;
  const int32_t* moe_id;
  int moe_n;
  // native bit-plane records (PL = true: ns_weight::native): the format's bit width and the lanes of a record request
  // (record bytes / 16; the record's planes are contiguous, so a k-step is still ONE request)
  uint32_t pl_bits, pl_lanes;
  // ---- cold: read late, through the kernel-argument pointer (keeps them out of the streaming loop's SGPRs) ----
  GemvMat mat[3];
  float* c2;
  const float* d;
  int ldc, ldd, epilogue;
  // carried RMS norm, consumer side (ns_norm_link): per row, in_parts partial sums of squares of the un-normalised
  // activations, staged into LDS at ssq_off beside A (nullptr: A is already normalised);
  // row scale = 1 / sqrt(sum * in_inv_size + in_eps)
  const float* in_ssq;
  uint32_t in_parts, in_stride;
  uint32_t ssq_off;
  float in_eps, in_inv_size;
  const float* out_gamma;     // carried norm, producer: fp16 shadow = v * gamma[col] ...
  float* out_ssq;             // ... and out_ssq[row * out_stride + tile] = sum of v^2 over the tile's columns
  uint32_t out_stride;
  uint32_t* out_ovf;          // pinned host word, set when gamma * v does not fit the fp16 shadow (ns_route.h: the route then evaluates the token again without carried norms)
  GemvRope rope;
  F4Lut lut;
  F8Consts f8;
#ifdef NS_TRACE
  unsigned long long* trace;
#endif
};
// the kernel-argument segment as an opaque pointer: loads through it cannot be hoisted above the point it is made
using KArgs = const __attribute__((address_space(4))) GemvParams*;
__device__ __forceinline__ KArgs late_args() {
  uint64_t v = reinterpret_cast<uint64_t>(__builtin_amdgcn_kernarg_segment_ptr());
  asm volatile("" : "+s"(v));
  return reinterpret_cast<KArgs>(v);
}

#ifdef NS_TRACE
#define NS_GSTAMP(i)                                                                                     \
  do {                                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 4096)                                                     \
      p.trace[(size_t(blockIdx.x) * 16 + (threadIdx.x >> 6)) * 8 + (i)] = wall_clock64();                 \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
  } while (0)
#else
#define NS_GSTAMP(i)
#endif

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

enum GemvMode { GV_PLAIN = 0, GV_DUAL = 1, GV_MSEG = 2 };
// MODE: one matrix / two matrices of one shape streamed in lockstep (gate/up, SiLU-mul epilogue) / several matrices
// side by side along N (QKV).  Activations arrive as fp16 (the producer's shadow); fp32-only callers stay on
// smallm_kernel, which converts while staging.
// EXT: the launch carries an RMS norm (ns_norm_link) and / or the RoPE + kv-append epilogue (ns_qkv_rope).  A separate
// instantiation, because even untaken these paths cost every launch 0.1-0.3 us (later argument fetch, longer cold
// epilogue code: 1057 vs 1036 us on the 7B chain, profiles/r02o_ext_ab.txt).
// XV: 0 plain; 1 = EXT; 2 = A32: the activations arrive as fp32 only (a caller without a producer shadow: the reference's
// device graph hands bestla_device_f32f32_forward fp32 tensors).  The wave's share of A is loaded into registers in
// front of the ring requests, converted (round to nearest even, as the shadow's producers do) and written to LDS once
// only ring requests are left in flight — the weights are requested exactly as early as with a shadow.
constexpr int kGvA32Regs = 8;  // 16-byte loads of fp32 activations a wave holds in registers
// XV = 4 (MOE): A32 + the weight picked on the DEVICE: ne_compute_forward_mul_mat_id_q_f32_bestla (ne_layers.c:7783-7916) reads
// ids[token][id] on the host and calls bestla_f32f32_forward per (token, expert); here the id stays where the router's top-k left
// it — one scalar load of the id, one of the expert's base pointer, in front of the first weight request (two dependent round
// trips, ~1 us, instead of a host synchronisation) — and the token's row streams the expert at this kernel's rate.
// XV = 3 (I8S): the REFERENCE'S int8-compute numerics (its default for Q4_0: gemv_4bit_u8s8_fp32, kernel_ref.h:2371-2429) on the
// same streaming skeleton: A arrives as the u8 codes / scales / zero points of quantize_fp_u8_colblock (aquant_u8_kernel,
// bit-exact), a lane (column nn, k-slot g) takes exact integer dots of its eight codes per 32-deep slice with v_dot4 on the
// stored codes,  sum (a - za)(q - zb) = sum a u - (zb + bias) sum a - za sum u + 8 za (zb + bias)  (u = q + bias), and adds
// float(sum) * (scale_a * scale_b) per slice; up to four rows.  Reduction, epilogues, fused QKV / gate-up modes are shared.
// XV = 5 (I8Q, round 5): I8S with the activation quantizer INSIDE the launch — the rows arrive as fp32 (staged through registers like
// XV = 2), every workgroup quantizes them itself (quantize_fp_u8_colblock's arithmetic operation by operation: the lanes of a k-block
// combine max / min by xor-shuffles, order-independent like in aquant_u8_coop_kernel, so codes / scales / zero points are the same bits)
// and leaves codes, scales and zero points in LDS where XV = 3 finds the ones it fetched.  Saves the aquant launch in front of every
// GEMV of an int8-reference decode step (129 per Llama-2-7B token, ~3 us each); k-blocks of 32 .. 256 that divide K.
// PL (round 4): the codes arrive as the NATIVE bit planes of a 1-3 / 5-7 bit format (ns_weight::native, repack_planes_kernel) —
// 256 .. 896 bytes per k-step instead of the 1 KiB nibble / byte container — and each lane rebuilds its container words from its
// plane words with shifts and masks (stored codes, bias 2^(bits-1) folded into the conversion constants: the same fp16 values,
// hence the same bits out, as from the widened records).  KIND says which container: WK_INT4 for 1-3 bits, WK_INT8 for 5-7.
template <int KIND, int SPS, int SK, bool ASYM, int MODE, int XV, bool PL = false>
__global__ __launch_bounds__(1024) void gemv_kernel(const GemvParams p) {
  constexpr bool EXT = XV == 1, MOE = XV == 4, I8Q = XV == 5, A32 = XV == 2 || MOE || I8Q, I8S = XV == 3 || I8Q;
  static_assert(!PL || ((KIND == WK_INT4 || KIND == WK_INT8) && !I8S && !MOE), "native planes: integer formats, fp16 numerics");
  static_assert(!I8S || KIND == WK_INT4 || KIND == WK_INT8, "integer weights only");
  constexpr uint32_t AEL = I8S ? 1u : 2u;  // bytes per staged activation element
  constexpr bool DUAL = MODE == GV_DUAL, MSEG = MODE == GV_MSEG;
  constexpr int NJ = kind_is_8bit(KIND) ? 2 : 4;
  constexpr int KSTEP = NJ * 32;
  constexpr int NQ = DUAL ? 2 : 1;
  constexpr int PF = kGvPF;
  static_assert(PF % NQ == 0, "ring slots alternate between the two matrices");
  constexpr int SBYTES = SPS * (SK == SK_F32 ? 4 : 2);
  using Corr = CorrRaw<SPS, SK, ASYM>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  _Float16* a_lds = reinterpret_cast<_Float16*>(smem);
  NS_GSTAMP(0);

  // every kernel argument the prologue needs is fetched by ONE batch of scalar loads: without this hipcc sinks each
  // load to its first use and the first weight request waits for several dependent argument round trips
  {
    asm volatile("" ::"s"(p.wbase0), "s"(p.a), "s"(p.ks), "s"(p.qstride), "s"(p.nw_log2), "s"(p.s_off0), "s"(p.sstride),
                 "s"(p.srows), "s"(p.srow_mul), "s"(p.srow_shift), "s"(p.m), "s"(p.k), "s"(p.lda), "s"(p.row_stride),
                 "s"(p.ring_off), "s"(p.ring_stride));
    if constexpr (DUAL) asm volatile("" ::"s"(p.wbase1), "s"(p.s_off1));
    if constexpr (MSEG)
      asm volatile("" ::"s"(p.tb1), "s"(p.tb2), "s"(p.mat[0].wbase), "s"(p.mat[1].wbase), "s"(p.mat[2].wbase), "s"(p.mat[0].s_off),
                   "s"(p.mat[1].s_off), "s"(p.mat[2].s_off));
    if constexpr (ASYM) asm volatile("" ::"s"(p.z_off0), "s"(p.z_off1), "s"(p.zstride));
    if constexpr (PL) asm volatile("" ::"s"(p.pl_bits), "s"(p.pl_lanes));
  }
  const int tid = threadIdx.x;
  const uint32_t w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = tid & 63;
  const int nn = l & 15, g = l >> 4;
  const uint32_t NW = 1u << p.nw_log2;
  const uint32_t ks = p.ks;
  const uint32_t T = blockIdx.x;  // global tile (across the matrices of a fused QKV launch)

  typedef __attribute__((address_space(3))) unsigned char* LdsPtr;
  // ---- descriptors: one per matrix over its whole allocation (codes, scales and zero points share the base) ----
  Rsrc rw[NQ];
  uint32_t so[NQ], zo[NQ];
  int sg = 0;       // matrix of the fused launch the tile belongs to (QKV)
  uint32_t tl = T;  // tile inside that matrix
  if constexpr (DUAL) {
    rw[0] = make_rsrc(p.wbase0, 0x80000000u);
    rw[1] = make_rsrc(p.wbase1, 0x80000000u);
    so[0] = p.s_off0, so[1] = p.s_off1;
    zo[0] = p.z_off0, zo[1] = p.z_off1;
  } else if constexpr (MSEG) {
    // all three matrices' bases and offsets come with the one batch of argument loads and are SELECTED (a lookup by
    // index in the argument segment would be a second, dependent round trip in front of the first weight request)
    sg = int(T >= p.tb1) + int(T >= p.tb2);
    const uint8_t* wb = sg == 0 ? p.mat[0].wbase : (sg == 1 ? p.mat[1].wbase : p.mat[2].wbase);
    rw[0] = make_rsrc(wb, 0x80000000u);
    so[0] = sg == 0 ? p.mat[0].s_off : (sg == 1 ? p.mat[1].s_off : p.mat[2].s_off);
    if constexpr (ASYM) zo[0] = sg == 0 ? p.mat[0].z_off : (sg == 1 ? p.mat[1].z_off : p.mat[2].z_off);
    else zo[0] = 0;
    tl = T - (sg == 0 ? 0u : (sg == 1 ? p.tb1 : p.tb2));
  } else if constexpr (MOE) {
    const int e = *p.moe_id;  // wave-uniform: scalar loads
    if (e < 0 || e >= p.moe_n) {  // an out-of-range id zeroes its row (the reference would assert)
      const KArgs cz = late_args();
      const int colz = int(T) * 16 + nn;
      if (w == 0 && g == 0 && colz < cz->mat[0].n) {
        cz->mat[0].c[colz] = 0.f;
        if (cz->mat[0].c16) cz->mat[0].c16[colz] = (_Float16)0.f;
      }
      return;
    }
    rw[0] = make_rsrc(p.moe_table[e].codes, 0x80000000u);
    so[0] = p.s_off0;
    zo[0] = p.z_off0;
  } else {
    rw[0] = make_rsrc(p.wbase0, 0x80000000u);
    so[0] = p.s_off0;
    zo[0] = p.z_off0;
  }
  const uint32_t tile_q = tl * ks * p.qstride;
  const uint32_t tile_c = tl * p.srows;
  const uint32_t voff_q = l * 16, voff_s = nn * SBYTES, voff_z = nn * SPS;  // the only per-lane address parts
  const I4Consts i4c = {0x000f000fu, 0x00f000f0u, 0x64006400u};

  // this wave's private ring: up to PF slots, a slot = image of one record {1024 B codes | 16 x SBYTES scales | 16 x SPS
  // zero points}; slot i holds the record of item i (mod PF)
  constexpr uint32_t SLOT = 1024u + 16u * SBYTES + (ASYM ? 16u * SPS : 0u);
  constexpr int OPS = ASYM ? 3 : 2;  // requests per record
  static_assert(OPS * PF + kGvMaxRows * 2 <= 63, "vmcnt is a 6-bit counter (ring + the carried norm's pieces)");
  const LdsPtr ring = (LdsPtr)(smem) + p.ring_off + w * p.ring_stride;
  const uint32_t ring_lane = uint32_t(reinterpret_cast<uintptr_t>(ring)) + uint32_t(l) * 16u;  // LDS byte address
  const uint32_t ring_corr = uint32_t(reinterpret_cast<uintptr_t>(ring)) + 1024u + uint32_t(nn) * SBYTES;

  // item = (k-step s, matrix q = slot % NQ): codes (64 lanes x 16 B), then its scale row (SBYTES lanes x 16 B), then its
  // zero-point row (SPS lanes x 16 B), all non-temporal HBM -> LDS
  auto issue = [&](auto slot_c, uint32_t s) {
    constexpr int slot = decltype(slot_c)::value;
    constexpr int q = slot % NQ;
#if defined(__HIP_DEVICE_COMPILE__)  // (the host pass of hipcc cannot type-check this builtin and would drop the kernel stub)
    const uint32_t crow = tile_c + ((s * p.srow_mul) >> p.srow_shift);
    const LdsPtr dst = ring + slot * SLOT;
    if (!PL || uint32_t(l) < p.pl_lanes)  // (a native record is shorter than 64 x 16 bytes)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw[q], reinterpret_cast<__attribute__((address_space(3))) void*>(dst), 16, voff_q,
                                               tile_q + s * p.qstride, 0, 2);
    if (l < SBYTES)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw[q], reinterpret_cast<__attribute__((address_space(3))) void*>(dst + 1024), 16,
                                               voff_q, so[q] + crow * p.sstride, 0, 2);
    if constexpr (ASYM) {
      if (l < SPS)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw[q], reinterpret_cast<__attribute__((address_space(3))) void*>(dst + 1024 + 16 * SBYTES),
                                                 16, voff_q, zo[q] + crow * p.zstride, 0, 2);
    }
#endif
  };
  // wait until at most `younger` RECORDS requested after the one about to be consumed are still in flight
  auto wait_records = [&](uint32_t younger) {
    if (younger == uint32_t(PF - 1)) {  // steady state first
      wait_vmcnt<OPS * (PF - 1)>();
      return;
    }
    [&]<int... K>(std::integer_sequence<int, K...>) {
      (void)((younger == uint32_t(K) ? (wait_vmcnt<OPS * K>(), true) : false) || ...);
    }(std::make_integer_sequence<int, PF + 1>{});
  };

#define NS_FOR_SLOTS(BODY)                                        \
  {                                                               \
    [&]<int... I>(std::integer_sequence<int, I...>) {             \
      (([&] { constexpr int i = I; std::integral_constant<int, I> ic; (void)i; (void)ic; BODY }()), ...); \
    }(std::make_integer_sequence<int, PF>{});                     \
  }

  // ---- 1. activations requested first (small, and they must be in LDS before the first MFMA): fp16 rows go
  //      HBM/L2 -> LDS directly in 1 KiB pieces, piece c of row r by wave (r * pieces + c) % NW; LDS row r holds
  //      ks * KSTEP halves (columns >= K read as zero through the descriptor: k-step padding) ----
  const int rows = min(p.m, kGvMaxRows);
  floatx4 areg[A32 ? kGvA32Regs : 1];
  if constexpr (A32) {
    // fp32 rows: 1 KiB pieces (256 columns) by wave (r * pieces + c) % NW again, at most kGvA32Regs per wave (host)
    // the descriptor make_rsrc() builds, as four words for the asm operand: base, stride 0, bytes, flags
    const uint64_t abase = reinterpret_cast<uint64_t>(p.a);
    const uint4v ra_words = {uint32_t(abase), uint32_t(abase >> 32) & 0xffffu,
                             uint32_t(rows - 1) * uint32_t(p.lda) * 4u + uint32_t(p.k) * 4u, 0x00020000u};
    const uint32_t row_bytes = ks * uint32_t(KSTEP) * 4u;
    const uint32_t pieces = (row_bytes + 1023u) >> 10;
    const uint32_t total = uint32_t(rows) * pieces;
    uint32_t r = 0, c = w;  // piece u = w + i * NW is (row r, column piece c): stepped, not divided
#pragma unroll
    for (int i = 0; i < kGvA32Regs; i++) {
      const uint32_t u = w + (uint32_t(i) << p.nw_log2);
      areg[i] = floatx4{0.f, 0.f, 0.f, 0.f};
      if (u < total) {
        while (c >= pieces) c -= pieces, r++;
#if defined(__HIP_DEVICE_COMPILE__)
        // columns >= K of the last k-step come back as zero through the descriptor
        // by hand: hipcc does not count LDS-DMA requests, so for a load it knows of it waits for everything in flight
        // (vmcnt(0): the whole ring) before the first use; wait_records() below is this load's wait
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen"
                     : "=v"(areg[i])
                     : "v"(voff_q), "s"(ra_words), "s"(r * uint32_t(p.lda) * 4u + (c << 10))
                     : "memory");
#endif
        c += NW;
      }
    }
  } else {
    const Rsrc ra = make_rsrc(p.a, uint32_t(rows - 1) * uint32_t(p.lda) * AEL + uint32_t(p.k) * AEL);
    const uint32_t row_bytes = ks * uint32_t(KSTEP) * AEL;
    const uint32_t pieces = (row_bytes + 1023u) >> 10;
    const uint32_t total = uint32_t(rows) * pieces;
    const LdsPtr al = (LdsPtr)(smem);
    for (uint32_t u = w; u < total; u += NW) {
      uint32_t r = 0, c = u;
      if (rows > 1) {
        r = u / pieces;
        c = u - r * pieces;
      }
      const uint32_t left = row_bytes - (c << 10);  // bytes of the row from this piece on
#if defined(__HIP_DEVICE_COMPILE__)
      if (uint32_t(l) * 16u < left)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, reinterpret_cast<__attribute__((address_space(3))) void*>(al + r * p.row_stride * 2u + (c << 10)),
                                                 16, voff_q, r * uint32_t(p.lda) * AEL + (c << 10), 0, 0);
#endif
    }
    if constexpr (I8S) {  // the rows' activation scales + zero points: one span, 1 KiB pieces, read zero past its end
      const Rsrc rc = make_rsrc(p.i8_corr, p.i8_span);
      for (uint32_t u = w; u < ((p.i8_span + 1023u) >> 10); u += NW) {
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rc, reinterpret_cast<__attribute__((address_space(3))) void*>(al + p.ssq_off + (u << 10)), 16,
                                                 voff_q, u << 10, 0, 0);
#endif
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);

  // ---- 2. fill the ring: k-steps w, w + NW, ... of the tile belong to wave w ----
  const uint32_t first = w;
  const uint32_t nst = first < ks ? (ks - first + NW - 1) >> p.nw_log2 : 0u;
  NS_FOR_SLOTS({ if (uint32_t(i / NQ) < nst) issue(ic, first + ((i / NQ) << p.nw_log2)); })
  __builtin_amdgcn_sched_barrier(0);
  NS_GSTAMP(1);

  // ---- 2b. carried norm (consumer side): the rows' partial sums of squares, in_parts floats per row, go to LDS as
  //      1 KiB pieces too — requested AFTER the ring so that nothing is added in front of the first weight request
  //      (its arguments are fetched here, late); only the epilogue reads them.  The requesting wave waits a little
  //      more strictly for its first records (a piece counts as one more request in flight), that is all.  With up
  //      to 16 rows x pieces the 6-bit request counter cannot overflow: PF x OPS + 16 <= 63 is asserted below.
  const float* in_ssq = nullptr;
  uint32_t in_parts = 0, ssq_off = 0;
  if constexpr (EXT) {
    const KArgs mid = late_args();
    in_ssq = mid->in_ssq;
    in_parts = mid->in_parts;
    ssq_off = mid->ssq_off;
  }
  if (EXT && in_ssq && w == 0) {  // wave 0 finishes the tile: it fetches them itself and needs no barrier to read them
    const uint32_t in_stride = late_args()->in_stride;
    const uint32_t spieces = (in_parts * 4u + 1023u) >> 10;
    const Rsrc rs = make_rsrc(in_ssq, (uint32_t(rows - 1) * in_stride + in_parts) * 4u);
    const LdsPtr al = (LdsPtr)(smem);
    for (uint32_t u = 0; u < uint32_t(rows) * spieces; u++) {
      const uint32_t r = u / spieces, c = u - r * spieces;
      const uint32_t left = in_parts * 4u - (c << 10);
#if defined(__HIP_DEVICE_COMPILE__)
      if (uint32_t(l) * 16u < left)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, reinterpret_cast<__attribute__((address_space(3))) void*>(al + ssq_off + (u << 10)),
                                                 16, voff_q, r * in_stride * 4u + (c << 10), 0, 0);
#endif
    }
  }
  __builtin_amdgcn_sched_barrier(0);

  // ---- 3. this wave's activation pieces (the oldest requests in its queue) have landed once only its ring
  //      requests are left in flight ----
  wait_records(min(nst * uint32_t(NQ), uint32_t(PF)));
  if constexpr (A32) {
#pragma unroll
    for (int i = 0; i < kGvA32Regs; i++) asm volatile("" : "+v"(areg[i]));  // not to be read above the wait
    const uint32_t row_bytes = ks * uint32_t(KSTEP) * 4u;
    const uint32_t pieces = (row_bytes + 1023u) >> 10;
    const uint32_t total = uint32_t(rows) * pieces;
    uint32_t r = 0, c = w;
    if constexpr (I8Q) {
      const uint32_t lds0q = uint32_t(reinterpret_cast<uintptr_t>((LdsPtr)(smem)));
      const uint32_t lpb = 1u << (p.i8_bshift - 2u);  // lanes per k-block (4 columns per lane): 8 .. 64
      const uint32_t k_bytes = uint32_t(p.k) * 4u;
#pragma unroll
      for (int i = 0; i < kGvA32Regs; i++) {
        const uint32_t u = w + (uint32_t(i) << p.nw_log2);
        if (u < total) {  // (wave-uniform)
          while (c >= pieces) c -= pieces, r++;
          const floatx4 v = areg[i];
          float maxval = 1.17549435e-38f /* FLT_MIN: a full block, kernel_ref.h:1832 */, minval = 0.f;
#pragma unroll
          for (int e = 0; e < 4; e++) {
            maxval = v[e] > maxval ? v[e] : maxval;  // std::max(f, maxval): a NaN in f keeps maxval
            minval = v[e] < minval ? v[e] : minval;
          }
          // the k-block's lanes combine max / min (order-independent).  Up to 16 lanes by DPP — quad swaps, then the mirror of the
          // 8-lane half, then of the 16-lane row: no LDS operation, for which hipcc would first drain every LDS-DMA request of the ring —
          // wider blocks by ds_bpermute on top
          auto dpp = [](float x, auto ctrl) {
            return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, false));
          };
          auto combine = [&](float om, float on) {
            maxval = om > maxval ? om : maxval;
            minval = on < minval ? on : minval;
          };
          combine(dpp(maxval, std::integral_constant<int, 0xB1>{}), dpp(minval, std::integral_constant<int, 0xB1>{}));    // quad_perm [1,0,3,2]
          combine(dpp(maxval, std::integral_constant<int, 0x4E>{}), dpp(minval, std::integral_constant<int, 0x4E>{}));    // quad_perm [2,3,0,1]
          combine(dpp(maxval, std::integral_constant<int, 0x141>{}), dpp(minval, std::integral_constant<int, 0x141>{}));  // row_half_mirror: 8 lanes
          if (lpb > 8u) combine(dpp(maxval, std::integral_constant<int, 0x140>{}), dpp(minval, std::integral_constant<int, 0x140>{}));  // row_mirror: 16
#pragma unroll
          for (uint32_t d = 16; d < 64u; d <<= 1) {
            if (d < lpb) {  // (wave-uniform)
              const float om = __shfl_xor(maxval, int(d), 64), on = __shfl_xor(minval, int(d), 64);
              combine(om, on);
            }
          }
          const float scale = __fdiv_rn(__fsub_rn(maxval, minval), 255.f);
          const int zp = x86_cast_f32_u8(__fdiv_rn(__fsub_rn(0.f, minval), scale));
          const float rscale = __fdiv_rn(1.f, scale), zpf = float(zp);
          uint32_t codes = 0;
#pragma unroll
          for (int e = 0; e < 4; e++)
            codes |= uint32_t(x86_cast_f32_u8(__fadd_rn(zpf, float(x86_cvt_round_int(__fmul_rn(v[e], rscale)))))) << (8 * e);
          const uint32_t col_b = (c << 10) + uint32_t(l) * 16u;  // byte offset of this lane's four columns in the fp32 row
          if (col_b < k_bytes) {  // (K is a multiple of the k-block: a block's lanes are live or dead together)
            // by hand: for a visible LDS store hipcc would first wait for every LDS-DMA request in flight (the whole ring)
            asm volatile("ds_write_b32 %0, %1" ::"v"(lds0q + r * p.row_stride * 2u + (col_b >> 2)), "v"(codes) : "memory");
            if ((uint32_t(l) & (lpb - 1u)) == 0u) {
              const uint32_t kb = (col_b >> 2) >> p.i8_bshift;
              asm volatile("ds_write_b32 %0, %1" ::"v"(lds0q + p.ssq_off + (r * p.i8_nblk + kb) * 4u), "v"(scale) : "memory");
              asm volatile("ds_write_b8 %0, %1" ::"v"(lds0q + p.ssq_off + uint32_t(rows) * p.i8_nblk * 4u + r * p.i8_nblk + kb), "v"(uint32_t(zp)) : "memory");
            }
          }
          c += NW;
        }
      }
    } else {
#pragma unroll
    for (int i = 0; i < kGvA32Regs; i++) {
      const uint32_t u = w + (uint32_t(i) << p.nw_log2);
      if (u < total) {
        while (c >= pieces) c -= pieces, r++;
        const half2_t lo = half2_t{(_Float16)areg[i][0], (_Float16)areg[i][1]};
        const half2_t hi = half2_t{(_Float16)areg[i][2], (_Float16)areg[i][3]};
        const uint32_t dst = uint32_t(reinterpret_cast<uintptr_t>((LdsPtr)(smem))) + r * p.row_stride * 2u + (c << 9) + uint32_t(l) * 8u;
        // by hand: for a visible LDS store hipcc would first wait for every LDS-DMA request in flight (the whole ring)
        if ((c << 10) + uint32_t(l) * 16u < row_bytes)
          asm volatile("ds_write_b64 %0, %1" ::"v"(dst), "v"((uint64_t(as_u32(hi)) << 32) | as_u32(lo)) : "memory");
        c += NW;
      }
    }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  // workgroup barrier over the staged activations, written by hand: for __syncthreads() hipcc first waits for every
  // LDS-DMA request in flight (it cannot know that the rings are wave-private), i.e. for the whole ring to land
  asm volatile("s_barrier" ::: "memory");
  NS_GSTAMP(2);

  // A-fragment LDS offset of this lane (rows >= m are clamped: their output rows are discarded)
  const uint32_t aoff = uint32_t(min(nn, rows - 1)) * p.row_stride + 8 * g;
  floatx4 acc[NQ];
#pragma unroll
  for (int q = 0; q < NQ; q++) acc[q] = floatx4{0.f, 0.f, 0.f, 0.f};

  float accr[NQ][4];  // I8S: rows 0..3 of column nn, this lane's k-slots
#pragma unroll
  for (int q = 0; q < NQ; q++)
#pragma unroll
    for (int r = 0; r < 4; r++) accr[q][r] = 0.f;
  const uint32_t i8_rsb = p.row_stride * 2u;  // I8S: bytes per staged row of codes
  auto compute = [&](auto slot_c, uint32_t s) {
    constexpr int slot = decltype(slot_c)::value;
    constexpr int q = slot % NQ;
#if defined(NS_GV_ABL) && NS_GV_ABL == 1  // timing ablation (diagnostic builds): the stream alone, no record is decoded or multiplied
    (void)s;
    return;
#endif
    if constexpr (I8S) {
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
      constexpr int BIAS = KIND == WK_INT4 ? 8 : 128;
      const uint32_t lds0 = uint32_t(reinterpret_cast<uintptr_t>((LdsPtr)(smem)));
      const uint32_t kbase = s * uint32_t(KSTEP);
      const uint32_t i8_sbase = lds0 + p.ssq_off;                            // [rows][nblk] fp32 scales
      const uint32_t i8_zbase = i8_sbase + uint32_t(rows) * p.i8_nblk * 4u;  // [rows][nblk] u8 zero points
      // the record's weight side, once for all rows: codes as bytes (k ascending), their sums, zero points + bias, k-blocks
      uint32_t u0[NJ], u1[NJ], kb[NJ];
      int su[NJ], zbq[NJ];
#pragma unroll
      for (int j = 0; j < NJ; j++) {
        if constexpr (KIND == WK_INT4) {
          const uint32_t t0 = xw[j] & 0x0f0f0f0fu, t1 = (xw[j] >> 4) & 0x0f0f0f0fu;  // codes 0,4,1,5 and 2,6,3,7
          u0[j] = __builtin_amdgcn_perm(t1, t0, 0x06040200u);
          u1[j] = __builtin_amdgcn_perm(t1, t0, 0x07050301u);
        } else {
          u0[j] = xw[2 * j] ^ 0x80808080u, u1[j] = xw[2 * j + 1] ^ 0x80808080u;  // raw int8 -> q + 128
        }
        su[j] = int(__builtin_amdgcn_udot4(u0[j], 0x01010101u, __builtin_amdgcn_udot4(u1[j], 0x01010101u, 0u, false), false));
        zbq[j] = (ASYM ? int(zp[j]) : 0) + BIAS;
        kb[j] = min((kbase + 32u * uint32_t(j)) >> p.i8_bshift, p.i8_nblk - 1u);  // (slices past K: clamped, never used)
      }
#pragma unroll
      for (int r = 0; r < 4; r++) {
        if (r < rows) {
          // LDS reads by hand, one batch per row and record: for a visible LDS read hipcc waits for EVERY LDS-DMA request in
          // flight (vmcnt(0): the whole ring) — it cannot know that the rings and the staged activations do not overlap
          uint64_t av[4] = {0, 0, 0, 0};
          uint32_t zau[4] = {0, 0, 0, 0}, asu[4] = {0, 0, 0, 0};
          const uint32_t aaddr = lds0 + uint32_t(r) * i8_rsb + kbase + 8u * uint32_t(g);
          const uint32_t zrow = i8_zbase + uint32_t(r) * p.i8_nblk, srow = i8_sbase + uint32_t(r) * p.i8_nblk * 4u;
          if constexpr (NJ == 4) {
            asm volatile(
                "ds_read_b64 %0, %12\n\tds_read_b64 %1, %12 offset:32\n\tds_read_b64 %2, %12 offset:64\n\tds_read_b64 %3, %12 offset:96\n\t"
                "ds_read_u8 %4, %13\n\tds_read_u8 %5, %14\n\tds_read_u8 %6, %15\n\tds_read_u8 %7, %16\n\t"
                "ds_read_b32 %8, %17\n\tds_read_b32 %9, %18\n\tds_read_b32 %10, %19\n\tds_read_b32 %11, %20\n\ts_waitcnt lgkmcnt(0)"
                : "=&v"(av[0]), "=&v"(av[1]), "=&v"(av[2]), "=&v"(av[3]), "=&v"(zau[0]), "=&v"(zau[1]), "=&v"(zau[2]), "=&v"(zau[3]),
                  "=&v"(asu[0]), "=&v"(asu[1]), "=&v"(asu[2]), "=&v"(asu[3])
                : "v"(aaddr), "v"(zrow + kb[0]), "v"(zrow + kb[1]), "v"(zrow + kb[2]), "v"(zrow + kb[3]), "v"(srow + kb[0] * 4u),
                  "v"(srow + kb[1] * 4u), "v"(srow + kb[2] * 4u), "v"(srow + kb[3] * 4u)
                : "memory");
          } else {
            asm volatile(
                "ds_read_b64 %0, %6\n\tds_read_b64 %1, %6 offset:32\n\tds_read_u8 %2, %7\n\tds_read_u8 %3, %8\n\t"
                "ds_read_b32 %4, %9\n\tds_read_b32 %5, %10\n\ts_waitcnt lgkmcnt(0)"
                : "=&v"(av[0]), "=&v"(av[1]), "=&v"(zau[0]), "=&v"(zau[1]), "=&v"(asu[0]), "=&v"(asu[1])
                : "v"(aaddr), "v"(zrow + kb[0]), "v"(zrow + kb[1]), "v"(srow + kb[0] * 4u), "v"(srow + kb[1] * 4u)
                : "memory");
          }
#pragma unroll
          for (int j = 0; j < NJ; j++) {
            if (kbase + 32u * uint32_t(j) < uint32_t(p.k)) {  // wave-uniform: a 32-deep slice lies inside K or outside
              const uint32_t avx = uint32_t(av[j]), avy = uint32_t(av[j] >> 32);
              const int za = int(zau[j]);
              const int dot = int(__builtin_amdgcn_udot4(avx, u0[j], __builtin_amdgcn_udot4(avy, u1[j], 0u, false), false));
              const int sa = int(__builtin_amdgcn_udot4(avx, 0x01010101u, __builtin_amdgcn_udot4(avy, 0x01010101u, 0u, false), false));
              const int isum = dot - zbq[j] * sa - za * su[j] + 8 * za * zbq[j];
              accr[q][r] += float(isum) * (__builtin_bit_cast(float, asu[j]) * sc[j]);
            }
          }
        }
      }
      return;
    }
    const _Float16* abase = a_lds + s * KSTEP + aoff;
    // the record's scales / zero points of column nn and the lane's 16 B of codes, out of the ring slot
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
    uint32_t xw[4];
    if constexpr (!PL) {
      const uint4v qvv = *reinterpret_cast<const __attribute__((address_space(3))) uint4v*>(ring_lane + uint32_t(slot) * SLOT);
      xw[0] = qvv.x, xw[1] = qvv.y, xw[2] = qvv.z, xw[3] = qvv.w;
    } else {
      // the lane's plane words out of the slot (layouts: repack_planes_kernel), assembled into the container words
      typedef const __attribute__((address_space(3))) uint32_t* L32;
      typedef const __attribute__((address_space(3))) uint16_t* L16;
      typedef uint32_t uint2v __attribute__((ext_vector_type(2)));
      const uint32_t rec = uint32_t(reinterpret_cast<uintptr_t>(ring)) + uint32_t(slot) * SLOT;
      const uint32_t bits = p.pl_bits;  // wave-uniform
      if constexpr (KIND == WK_INT4) {
        uint32_t a0 = 0, a1 = 0, cw = 0;
        if (bits >= 2) {
          const uint2v av = *reinterpret_cast<const __attribute__((address_space(3))) uint2v*>(rec + 8u * uint32_t(l));
          a0 = av.x, a1 = av.y;
        }
        if (bits != 2) cw = *reinterpret_cast<L32>(rec + (bits == 3 ? 512u : 0u) + 4u * uint32_t(l));
        // the 1-bit plane's bit goes to nibble bit 0 (S1) or 2 (S3): one shift + one and-or per word
        const uint32_t cm = bits == 3 ? 0x44444444u : 0x11111111u;
        const uint32_t c0 = bits == 3 ? cw << 2 : cw, c1 = bits == 3 ? cw << 1 : cw >> 1, c2 = bits == 3 ? cw : cw >> 2,
                       c3 = bits == 3 ? cw >> 1 : cw >> 3;
        xw[0] = and_or(c0, cm, a0 & 0x33333333u);
        xw[1] = and_or(c1, cm, (a0 >> 2) & 0x33333333u);
        xw[2] = and_or(c2, cm, a1 & 0x33333333u);
        xw[3] = and_or(c3, cm, (a1 >> 2) & 0x33333333u);
      } else {
        const uint2v nv = *reinterpret_cast<const __attribute__((address_space(3))) uint2v*>(rec + 8u * uint32_t(l));
        const uint32_t pw = *reinterpret_cast<L32>(rec + 512u + 4u * uint32_t(l));
        // S5: bit 4 of byte b of word d sits at bit 8 b + d of the plane word; S6: bits 4..5 at 8 b + 2 d
        const uint32_t pm = bits == 5 ? 0x10101010u : 0x30303030u;
        const uint32_t st = bits == 5 ? 1u : 2u;
        xw[0] = and_or(pw << 4, pm, nv.x & 0x0f0f0f0fu);
        xw[1] = and_or(pw << (4u - st), pm, (nv.x >> 4) & 0x0f0f0f0fu);
        xw[2] = and_or(pw << (4u - 2u * st), pm, nv.y & 0x0f0f0f0fu);
        xw[3] = and_or(bits == 5 ? pw << 1 : pw >> 2, pm, (nv.y >> 4) & 0x0f0f0f0fu);
      }
    }
    half8_t bq[NJ];
    float full = 0.f;  // PL: the codes are the stored ones (q + 2^(bits-1)); widened records were re-biased at load
    if constexpr (PL) full = float(1u << (p.pl_bits - 1u));
#pragma unroll
    for (int j = 0; j < NJ; j++) {
      if constexpr (KIND == WK_INT4) {
        const _Float16 zl = PL ? (_Float16)(-1024.f - full - zp[j]) : (_Float16)(-1032.f - zp[j]);
        const _Float16 zh = PL ? (_Float16)(-64.f - full - zp[j]) : (_Float16)(-72.f - zp[j]);
        bq[j] = cvt_i4x8(xw[j], i4c, half2_t{zl, zl}, half2_t{zh, zh});
      } else if constexpr (KIND == WK_INT8) {
        if constexpr (PL) {
          const _Float16 zo8 = (_Float16)(-1024.f - full - zp[j]);
          bq[j] = cvt_u8x8(xw[2 * j], xw[2 * j + 1], half2_t{zo8, zo8});
        } else {
          const _Float16 zo8 = (_Float16)(-1152.f - zp[j]);
          bq[j] = cvt_i8x8(xw[2 * j], xw[2 * j + 1], half2_t{zo8, zo8});
        }
      } else if constexpr (KIND == WK_F8) {
        bq[j] = cvt_f8x8(xw[2 * j], xw[2 * j + 1], p.f8);
      } else {
        bq[j] = cvt_f4x8(xw[j], p.lut);
      }
    }
    floatx4 dd[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) {
      const half8_t afrag = *reinterpret_cast<const half8_t*>(abase + 32 * j);
      dd[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(afrag, bq[j], floatx4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < NJ; j++) acc[q] += dd[j] * sc[j];
  };

  // ---- 4. stream: consume the oldest record, refill its slot with the record PF items ahead ----
  constexpr int SPR = PF / NQ;  // k-steps per round
  const uint32_t rounds = (nst + SPR - 1) / SPR;
  const uint32_t nitems = nst * NQ;
  for (uint32_t r = 0; r < rounds; r++) {
    NS_FOR_SLOTS({
      const uint32_t ord = r * SPR + i / NQ;  // ordinal of the k-step among this wave's
      if (ord < nst) {
        const uint32_t t = r * PF + i;  // ordinal of the item
        wait_records(min(nitems - t - 1, uint32_t(PF - 1)));
        compute(ic, first + (ord << p.nw_log2));
        __builtin_amdgcn_sched_barrier(0);
        if (ord + SPR < nst) issue(ic, first + ((ord + SPR) << p.nw_log2));
        __builtin_amdgcn_sched_barrier(0);
      }
    })
  }
  NS_GSTAMP(4);

  if constexpr (I8S) {  // the four k-slots of a column live in lanes nn, nn + 16, nn + 32, nn + 48; rows 0..3 = lane group 0's
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      float t[4];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        float v = accr[q][r];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        t[r] = v;
      }
      acc[q] = g == 0 ? floatx4{t[0], t[1], t[2], t[3]} : floatx4{0.f, 0.f, 0.f, 0.f};
    }
  }
  // ---- 5. cross-wave reduction: every wave writes its partial sums over ITS OWN ring, wave 0 adds them in wave order ----
  floatx4* red = reinterpret_cast<floatx4*>(smem + p.ring_off);
  const uint32_t kRedWave = p.ring_stride / 16;  // floatx4 per wave region (>= NQ KiB, checked by the host)
#pragma unroll
  for (int q = 0; q < NQ; q++) red[w * kRedWave + q * 64 + l] = acc[q];
  const KArgs cold = late_args();  // epilogue-only arguments: fetched here, not held through the streaming loop
  const auto* mp = &cold->mat[MSEG ? sg : 0];  // indexed scalar loads
  const int ncols = mp->n;
  float* cbase = mp->c;
  _Float16* c16 = mp->c16;
  const int ldc = cold->ldc, ldd = cold->ldd, epi = cold->epilogue;
  const float* dptr = cold->d;
  float* c2 = cold->c2;
  float in_eps = 0.f, in_inv = 0.f;
  const float* ogamma = nullptr;
  float* ossq = nullptr;
  uint32_t ostride = 0;
  if constexpr (EXT) {
    in_eps = cold->in_eps, in_inv = cold->in_inv_size;
    ogamma = cold->out_gamma;
    ossq = cold->out_ssq;
    ostride = cold->out_stride;
  }
  // Wave 0 finishes the tile.  Everything its epilogue has to FETCH (the epilogue operand, the next norm's weight, the
  // RoPE table entry) is requested before the workgroup barrier, so the round trips overlap the wait for the slowest
  // wave instead of following it (each is 1-2 us at the tail of a 5-12 us launch).
  const int col = int(tl) * 16 + nn;
  const bool col_ok = col < ncols;
  float dvp[4] = {0.f, 0.f, 0.f, 0.f};
  float gam = 1.f;
  float2 cs[4];
  int rope_head = 0, rope_e = 0;
  bool rope_on = false;
  _Float16* rope_cache = nullptr;
  float* rope_cell = nullptr;
  long long rope_sl = 0;
  // carried norm, consumer side: A was gamma * x, not yet normalised; the row's 1 / rms scales the finished dot
  // products (ne_compute_forward_rms_norm_f32, ne_layers.c: scale = 1 / sqrtf(mean + eps)).  Per row the 64 lanes add
  // the staged partial sums in a fixed order (lane-strided, then a butterfly).
  float rscale[4] = {1.f, 1.f, 1.f, 1.f};
  if (w == 0) {
    if constexpr (!DUAL) {
      if (dptr && col_ok) {
#pragma unroll
        for (int rr = 0; rr < 4; rr++)
          if (4 * g + rr < p.m) dvp[rr] = dptr[size_t(4 * g + rr) * ldd + col];
      }
    }
    if constexpr (EXT) {
      if (ogamma && col_ok) gam = ogamma[col];
      if constexpr (MSEG) {
        rope_on = cold->rope.on != 0;
        if (rope_on) {
          const int hs = cold->rope.head_size;
          rope_head = col / hs;
          rope_e = col - rope_head * hs;
          if (sg < 2) {
#pragma unroll
            for (int rr = 0; rr < 4; rr++) cs[rr] = cold->rope.cos_sin[min(4 * g + rr, rows - 1) * (hs >> 1) + (rope_e >> 1)];
          }
          rope_sl = cold->rope.c_sl;
          const long long kk = cold->rope.kmove ? (long long)*cold->rope.kmove : 0ll;  // tokens since the plan was captured
          rope_cache = (sg == 1 ? cold->rope.kc : cold->rope.vc) + ((long long)cold->rope.n_past + kk * cold->rope.kd_pos) * rope_sl +
                       (long long)rope_head * cold->rope.c_head + rope_e;
          if (sg > 0 && cold->rope.k32)
            rope_cell = sg == 1 ? cold->rope.k32 + kk * cold->rope.k32_tok + rope_head * cold->rope.k32_head + rope_e * cold->rope.k32_dim
                                : cold->rope.v32 + kk * cold->rope.v32_tok + rope_head * cold->rope.v32_head + rope_e * cold->rope.v32_dim;
        }
      }
      if (in_ssq) {  // wave 0 requested the pieces itself and has waited for all its requests (last record: vmcnt 0)
        const uint32_t ssq_ld = ((in_parts * 4u + 1023u) >> 10) << 8;  // floats per staged row
        const float* sl = reinterpret_cast<const float*>(smem + ssq_off);
        for (int row = 0; row < rows; row++) {
          float t = 0.f;
          for (uint32_t j = uint32_t(l); j < in_parts; j += 64u) t += sl[uint32_t(row) * ssq_ld + j];
#pragma unroll
          for (int o = 1; o < 64; o <<= 1) t += __shfl_xor(t, o, 64);
          const float r = 1.0f / sqrtf(t * in_inv + in_eps);
          if ((row >> 2) == g) {
#pragma unroll
            for (int rr = 0; rr < 4; rr++)
              if ((row & 3) == rr) rscale[rr] = r;
          }
        }
      }
    }
  }
  __syncthreads();
  if (w == 0) {
    floatx4 sum[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      sum[q] = floatx4{0.f, 0.f, 0.f, 0.f};
      for (uint32_t ww = 0; ww < NW; ww++) sum[q] += red[ww * kRedWave + q * 64 + l];
    }
    // ---- 6. epilogue: lane (nn, g) holds rows 4g .. 4g+3 of column nn ----
#pragma unroll
    for (int rr = 0; rr < 4; rr++) {
      const int row = 4 * g + rr;
      const bool ok = col_ok && row < p.m;
      float v = sum[0][rr] * rscale[rr];
      if constexpr (MSEG && EXT) {
        if (rope_on) {
          // ne_rope (mode 0) on q and k, then the kv-cache append (models/llama/llama.cpp:232-262): the arithmetic of
          // rope_qkv_append_kernel (ns_quant.hip), separately rounded multiplies; (cos, sin) from the per-token table
          // (evaluating cosf / sinf here — argument reduction for angles up to the context length — cost 12 us per
          // launch, profiles/r02n)
          const float vp = __shfl_xor(v, 1, 64);  // the pair's other element (same tile: 16 | even head_size)
          if (sg < 2)
            v = (rope_e & 1) ? __fadd_rn(__fmul_rn(vp, cs[rr].y), __fmul_rn(v, cs[rr].x))
                             : __fsub_rn(__fmul_rn(v, cs[rr].x), __fmul_rn(vp, cs[rr].y));
          if (ok && sg > 0) {
            rope_cache[(long long)row * rope_sl] = (_Float16)v;
            if (rope_cell && row == 0) {
              *rope_cell = v;
              if (fabsf(v) > 65504.f && cold->rope.ovf) __hip_atomic_store(cold->rope.ovf, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
          }
        }
      }
      if (ok) {
        if constexpr (DUAL) {
          // tmp1 = act(A*W1) ; tmp2 = (A*W3) * tmp1   (neural_speed/core/layers/ip_fusion_ffn.cpp:364-406)
          const float t1 = (epi == 5) ? epi_silu(v) : epi_gelu(v);
          if (c2) c2[size_t(row) * ldc + col] = t1;
          v = sum[1][rr] * rscale[rr] * t1;
        } else {
          const float dv = dvp[rr];
          switch (epi) {
            case 1: v = v + dv; break;            // custom::epilogue::Add
            case 2: v = v * dv; break;            // custom::epilogue::Mul
            case 3: v = epi_gelu(v + dv); break;  // custom::epilogue::Add_Gelu
            case 4: v = epi_gelu(v); break;
            case 5: v = epi_silu(v); break;
            default: break;
          }
        }
        cbase[size_t(row) * ldc + col] = v;
        // carried norm, producer side: the shadow the NEXT operator streams is gamma * v (its norm's weight), the
        // normalisation itself follows from the partial sums below
        if (c16) c16[size_t(row) * ldc + col] = (_Float16)(v * gam);
        if constexpr (EXT) {
          if (ogamma && fabsf(v * gam) > 65504.f && cold->out_ovf) __hip_atomic_store(cold->out_ovf, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
      if (EXT && ossq) {  // sum of squares of this tile's 16 columns of row `row`, fixed order
        float t = ok ? v * v : 0.f;
        t += __shfl_xor(t, 1, 64);
        t += __shfl_xor(t, 2, 64);
        t += __shfl_xor(t, 4, 64);
        t += __shfl_xor(t, 8, 64);
        if (nn == 0 && row < p.m) ossq[size_t(row) * ostride + tl] = t;
      }
    }
  }
#undef NS_FOR_SLOTS
  NS_GSTAMP(6);
#ifdef NS_TRACE
  if ((threadIdx.x & 63) == 0 && blockIdx.x < 4096)  // where the wave ran: XCC_ID (reg 20) and HW_ID (reg 4)
    p.trace[(size_t(blockIdx.x) * 16 + (threadIdx.x >> 6)) * 8 + 7] =
        (uint64_t(__builtin_amdgcn_s_getreg((31 << 11) | 20)) << 32) | uint32_t(__builtin_amdgcn_s_getreg((31 << 11) | 4));
#endif
}

// ============================================================================================================
// host side
// ============================================================================================================
constexpr int kGvModeA32 = 0x100;  // or-ed into the launch mode: fp32 activations (XV = 2)
constexpr int kGvModeI8 = 0x200;   // int8-reference numerics (XV = 3)
constexpr int kGvModeMoe = 0x400;  // expert picked on the device (XV = 4; fp32 activations)
constexpr int kGvModePlanes = 0x800;  // native bit-plane records (PL = true)
template <int KIND, int SPS, int SK, bool ASYM>
static hipError_t launch_gemv_planes(const GemvParams& p, int mode, int grid, int nw, size_t lds, hipStream_t st) {
  if constexpr ((KIND == WK_INT4 || KIND == WK_INT8) && SK != SK_F16) {
    const dim3 g(grid), b(nw * 64);
    const bool ext = p.in_ssq || p.out_gamma || p.out_ssq || p.rope.on;
    const bool a32 = (mode & kGvModeA32) != 0;
    mode &= 3;
#define NS_GVP(MODEV, EXTV)                                                                                                  \
  {                                                                                                                         \
    auto k = gemv_kernel<KIND, SPS, SK, ASYM, MODEV, EXTV, true>;                                                            \
    static const hipError_t attr =                                                                                          \
        hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, int(kGvMaxLds));   \
    if (attr != hipSuccess && lds > 64 * 1024) return attr;                                                                 \
    hipLaunchKernelGGL(k, g, b, lds, st, p);                                                                                \
  }
#define NS_GVP_M(MODEV)                                                              \
  {                                                                                 \
    if (ext) NS_GVP(MODEV, 1) else if (a32) NS_GVP(MODEV, 2) else NS_GVP(MODEV, 0)   \
  }
    if (mode == GV_DUAL)
      NS_GVP_M(GV_DUAL)
    else if (mode == GV_MSEG)
      NS_GVP_M(GV_MSEG)
    else
      NS_GVP_M(GV_PLAIN)
#undef NS_GVP_M
#undef NS_GVP
    return hipGetLastError();
  } else {
    return hipErrorNotSupported;
  }
}
template <int KIND, int SPS, int SK, bool ASYM>
static hipError_t launch_gemv_k(const GemvParams& p, int mode, int grid, int nw, size_t lds, hipStream_t st) {
  if (mode & kGvModePlanes) return launch_gemv_planes<KIND, SPS, SK, ASYM>(p, mode & ~kGvModePlanes, grid, nw, lds, st);
  const dim3 g(grid), b(nw * 64);
  const bool ext = p.in_ssq || p.out_gamma || p.out_ssq || p.rope.on;
  const bool a32 = (mode & kGvModeA32) != 0;  // never together with ext (launch_gemv)
  const bool i8s = (mode & kGvModeI8) != 0;
  const bool moe = (mode & kGvModeMoe) != 0;
  mode &= ~(kGvModeA32 | kGvModeI8 | kGvModeMoe);
  if (moe) {
    if constexpr (KIND == WK_F8) return hipErrorNotSupported;
    if (mode != GV_PLAIN) return hipErrorNotSupported;
  }
  if constexpr (KIND != WK_INT4 && KIND != WK_INT8)
    if (i8s) return hipErrorNotSupported;
#define NS_GV_LAUNCH(MODEV)                                                                                     \
  {                                                                                                             \
    if (moe) NS_GV_LAUNCH_MOE(MODEV) else if (ext) NS_GV_LAUNCH_E(MODEV, 1) else if (i8s && a32) NS_GV_LAUNCH_I8Q(MODEV) else if (a32) NS_GV_LAUNCH_E(MODEV, 2) else if (i8s) NS_GV_LAUNCH_I8(MODEV)    \
    else NS_GV_LAUNCH_E(MODEV, 0)                                                                               \
  }
#define NS_GV_LAUNCH_MOE(MODEV)                                                                                  \
  {                                                                                                             \
    if constexpr (KIND != WK_F8 && MODEV == GV_PLAIN) NS_GV_LAUNCH_E(GV_PLAIN, 4)                               \
  }
#define NS_GV_LAUNCH_I8Q(MODEV)                                                                                  \
  {                                                                                                             \
    if constexpr (KIND == WK_INT4 || KIND == WK_INT8) NS_GV_LAUNCH_E(MODEV, 5)                                  \
  }
#define NS_GV_LAUNCH_I8(MODEV)                                                                                   \
  {                                                                                                             \
    if constexpr (KIND == WK_INT4 || KIND == WK_INT8) NS_GV_LAUNCH_E(MODEV, 3)                                  \
  }
#define NS_GV_LAUNCH_E(MODEV, EXTV)                                                                             \
  {                                                                                                             \
    auto k = gemv_kernel<KIND, SPS, SK, ASYM, MODEV, EXTV>;                                                      \
    static const hipError_t attr =                                                                              \
        hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, int(kGvMaxLds)); \
    if (attr != hipSuccess && lds > 64 * 1024) return attr;                                                     \
    hipLaunchKernelGGL(k, g, b, lds, st, p);                                                                    \
  }
  if (mode == GV_DUAL)
    NS_GV_LAUNCH(GV_DUAL)
  else if (mode == GV_MSEG)
    NS_GV_LAUNCH(GV_MSEG)
  else
    NS_GV_LAUNCH(GV_PLAIN)
#undef NS_GV_LAUNCH
#undef NS_GV_LAUNCH_E
#undef NS_GV_LAUNCH_I8
#undef NS_GV_LAUNCH_I8Q
#undef NS_GV_LAUNCH_MOE
  return hipGetLastError();
}
template <int KIND, int SPS, int SK>
static hipError_t launch_gemv_a(const GemvParams& p, bool asym, int mode, int grid, int nw, size_t lds,
                                hipStream_t st) {
  if constexpr (KIND == WK_F4 || KIND == WK_F8) {
    (void)asym;
    return launch_gemv_k<KIND, SPS, SK, false>(p, mode, grid, nw, lds, st);
  } else {
    if (asym) return launch_gemv_k<KIND, SPS, SK, true>(p, mode, grid, nw, lds, st);
    return launch_gemv_k<KIND, SPS, SK, false>(p, mode, grid, nw, lds, st);
  }
}
template <int KIND, int SPS>
static hipError_t launch_gemv_s(const GemvParams& p, uint32_t scale_dt, bool asym, int mode, int grid, int nw,
                                size_t lds, hipStream_t st) {
  if (scale_dt == DT_F32) return launch_gemv_a<KIND, SPS, SK_F32>(p, asym, mode, grid, nw, lds, st);
  if (scale_dt == DT_F16) return launch_gemv_a<KIND, SPS, SK_F16>(p, asym, mode, grid, nw, lds, st);
  return launch_gemv_a<KIND, SPS, SK_BF16>(p, asym, mode, grid, nw, lds, st);
}

#ifdef NS_TRACE
unsigned long long* trace_buffer();
#endif

static std::atomic<int> g_gemv_mode{-1};  // -1: read NS_GEMV2 once; 0 off (first-generation kernel); 1 on
void set_gemv_mode(int mode) { g_gemv_mode.store(mode); }
static int gemv_mode() {
  int m = g_gemv_mode.load();
  if (m < 0) {
    const char* e = getenv("NS_GEMV2");
    m = e ? atoi(e) : 1;
    g_gemv_mode.store(m);
  }
  return m;
}

// waves per workgroup of a decode launch (one 16-column tile per workgroup, k-steps dealt round-robin to the waves).
// Shared with smallm_kernel so that both kernels split K the same way — bit-identical sums between a caller that passes
// the fp16 shadow of A and an fp32-only caller — wherever this rule alone decides: launch_gemv additionally halves the
// wave count until activations + rings fit in LDS (several rows of a large K), where smallm_kernel keeps the rule's value;
// the two then differ in fp32 summation order only (fp32-only callers of decode shapes are served by gemv_kernel itself
// since round 3, bit-equal to the shadow path: tests/test_gpu_fullsize.py).
static std::atomic<int> g_decode_waves{0};  // ns_hip_set_tuning("gv_nw", n)
void set_decode_waves(int nw) { g_decode_waves.store(nw); }
int decode_waves(int grid, int ks, bool dual) {
  // measured on the 7B shapes (profiles/r02g_sweep.txt): 256 tiles x 32 k-steps (attention output) 16 waves, 256 tiles
  // x 86 k-steps (FFN down) 8 waves (7.5 vs 8.0 us at 16), 768 tiles 4 waves, 2000 tiles (lm_head) 2 waves
  int nw = (grid <= 320 && ks >= 32 && ks <= 48 && !dual) ? 16 : 8;
  if (dual) {
    nw = grid * 4 >= 1300 ? 4 : 8;
  } else {
    const int target_waves = 2560;
    while (nw > 2 && grid * (nw / 2) >= target_waves) nw /= 2;
  }
  static const int env_nw0 = getenv("NS_GV_NW") ? atoi(getenv("NS_GV_NW")) : 0;  // diagnostics
  const int forced = g_decode_waves.load();
  const int env_nw = forced ? forced : env_nw0;
  if (env_nw == 2 || env_nw == 4 || env_nw == 8 || (env_nw == 16 && !dual)) nw = env_nw;
  while (nw > 1 && nw > ks) nw /= 2;
  return nw;
}

// hipErrorNotSupported: outside the kernel's envelope — the caller falls back to smallm_kernel
static std::atomic<int> g_gemv_planes{-1};  // ns_hip_set_tuning("planes"): 1 (default) = bit-plane formats stream their native records
void set_gemv_planes(int on) { g_gemv_planes.store(on != 0); }
static bool gemv_planes() {
  int v = g_gemv_planes.load();
  if (v < 0) {
    const char* e = getenv("NS_GEMV_PLANES");
    v = e ? atoi(e) != 0 : 1;
    g_gemv_planes.store(v);
  }
  return v != 0;
}

hipError_t launch_gemv(const SmallMArgs& a, hipStream_t st) {
  if (gemv_mode() == 0 || a.m < 1 || a.m > kGvMaxRows) return hipErrorNotSupported;
  const int nmat = a.nseg;  // matrices the launch touches (dual: 2)
  // bit-plane formats: every matrix of the launch has its native clone (same shapes and scales, shorter code records) -> stream those
  bool planes = gemv_planes() && !a.i8 && !a.moe;
  for (int i = 0; i < nmat; i++)
    planes = planes && a.seg[i].w->native && a.seg[i].w->native->scale_dt != DT_F16 && a.seg[i].w->native->pl_bits == a.seg[0].w->native->pl_bits;
  auto pick = [&](const ns_weight* w) { return planes ? static_cast<const ns_weight*>(w->native) : w; };
  const ns_weight* w0 = pick(a.seg[0].w);
  GemvParams p;
  memset(&p, 0, sizeof(p));
  uint32_t tiles = 0;
  uint32_t tbeg[3] = {0, 0xffffffffu, 0xffffffffu};  // absent matrices begin beyond every tile
  const uint8_t* wb[3] = {nullptr, nullptr, nullptr};
  uint32_t soff[3] = {0, 0, 0}, zoff[3] = {0, 0, 0};
  for (int i = 0; i < nmat; i++) {
    const ns_weight* w = pick(a.seg[i].w);
    if (!w->single_span || w->alloc_bytes >= (size_t(1) << 31)) return hipErrorNotSupported;
    wb[i] = reinterpret_cast<const uint8_t*>(w->codes);
    soff[i] = uint32_t(reinterpret_cast<const uint8_t*>(w->scales) - wb[i]);
    zoff[i] = w->zps ? uint32_t(reinterpret_cast<const uint8_t*>(w->zps) - wb[i]) : 0u;
    tbeg[i] = a.dual ? 0u : tiles;
    if (!a.dual || i == 0) tiles += uint32_t(w->ntiles);
    p.mat[i] = GemvMat{wb[i], soff[i], zoff[i], tbeg[i], w->n, a.seg[i].c, static_cast<_Float16*>(a.seg[i].c16)};
  }
  const bool mseg = !a.dual && nmat > 1;
  const int mode = a.dual ? GV_DUAL : (mseg ? GV_MSEG : GV_PLAIN);
  p.wbase0 = wb[0];
  p.wbase1 = wb[1];
  p.s_off0 = soff[0], p.s_off1 = soff[1], p.z_off0 = zoff[0], p.z_off1 = zoff[1];
  p.tb1 = mseg ? tbeg[1] : 0xffffffffu;
  p.tb2 = (mseg && nmat > 2) ? tbeg[2] : 0xffffffffu;
  const uint32_t ks = uint32_t(w0->ksteps);
  const int kstep = w0->kstep_len;
  if (tiles == 0 || ks == 0) return hipErrorNotSupported;

  // staged activations: [rows][ks * KSTEP + 8] halves (int8-reference numerics: [rows][ks * KSTEP + 16] bytes)
  const int rows = a.m;
  const bool i8s = a.i8 != nullptr;
  bool i8q = false;
  const uint32_t row_stride = i8s ? (ks * uint32_t(kstep) + 16) / 2 : ks * uint32_t(kstep) + 8;
  const size_t a_bytes = size_t(rows) * row_stride * 2;
  if (a_bytes > kGvMaxALds) return hipErrorNotSupported;
  uint32_t i8_shift = 0;
  if (i8s) {
    const I8Act& q = *a.i8;
    const bool int_w = w0->kind == WK_INT4 || w0->kind == WK_INT8;
    // a 32-deep slice must lie inside one k-block and inside or outside K; block index by shift (one block: any shift >= 31)
    const bool one_block = q.nblk == 1;
    while (!one_block && (1u << i8_shift) < uint32_t(q.blocksize)) i8_shift++;
    if (!int_w || rows > 4 || a.link || a.rope || (w0->k & 31) || !q.aq || !q.corr || (reinterpret_cast<uintptr_t>(q.aq) & 15) ||
        (reinterpret_cast<uintptr_t>(q.corr) & 15) || (q.ldq & 15) || q.ldq < w0->k ||
        (!one_block && ((1u << i8_shift) != uint32_t(q.blocksize) || q.blocksize < 32)))
      return hipErrorNotSupported;
    if (one_block) i8_shift = 31;
    // XV = 5: quantize inside the launch when the rows are at hand as fp32 and the k-block is 32 .. 256 columns dividing K
    static const bool i8q_off = getenv("NS_I8_INKERNEL") && atoi(getenv("NS_I8_INKERNEL")) == 0;  // A-B runs
    i8q = !i8q_off && q.a32 && !one_block && i8_shift >= 5 && i8_shift <= 8 && w0->k % q.blocksize == 0 && (q.lda32 & 3) == 0 &&
          (w0->k & 3) == 0 && (reinterpret_cast<uintptr_t>(q.a32) & 15) == 0;
    if (!i8q && !q.quantized) return hipErrorNotReady;  // the caller runs the quantizer launch (i8_quantize_finish) and comes back
    p.i8_corr = q.corr;
    p.i8_nblk = uint32_t(q.nblk);
    p.i8_span = (uint32_t(rows) * uint32_t(q.nblk) * 5u + 3u) & ~3u;  // whole words: a buffer load drops a word that straddles the bound (the scratch has the slack)
    p.i8_bshift = i8_shift;
  }
  // fp16 activations with 16-byte aligned rows; several rows need K to fill whole k-steps (a row's padding columns
  // would otherwise read the next row through the descriptor)
  const bool a16 = (i8s && !i8q) || (!i8s && a.a16 != nullptr && (a.lda & 7) == 0 && (w0->k & 7) == 0 && (reinterpret_cast<uintptr_t>(a.a16) & 15) == 0);
  // fp32-only callers: converted while staging (XV = 2); not together with a carried norm / fused RoPE, whose producers
  // always leave a shadow
  const bool moe = a.moe != nullptr;
  if (moe && (a.m != 1 || nmat != 1 || a.dual || a.link || a.rope || i8s || a.a16 || !a.moe->table || !a.moe->id)) return hipErrorNotSupported;
  const bool a32 = i8q || (!i8s && !a16 && a.a != nullptr && !a.link && !a.rope && (a.lda & 3) == 0 && (w0->k & 3) == 0 &&
                   (reinterpret_cast<uintptr_t>(a.a) & 15) == 0);
  if (moe && !a32) return hipErrorNotSupported;
  if ((!a16 && !a32) || (rows > 1 && w0->k % kstep != 0)) return hipErrorNotSupported;
  p.a = i8q ? static_cast<const void*>(a.i8->a32) : i8s ? static_cast<const void*>(a.i8->aq) : (a16 ? a.a16 : static_cast<const void*>(a.a));
  // carried RMS norm (ns_norm_link): consumer side stages in_parts floats per row behind A
  size_t ssq_bytes = 0;
  if (i8s) ssq_bytes = (size_t(p.i8_span) + 1023) >> 10 << 10;  // the scales / zero points span sits where a carried norm's sums would
  if (a.link) {
    const ns_norm_link& k = *a.link;
    if (k.in_ssq) {
      if (k.in_parts < 1 || k.in_stride < k.in_parts || (k.in_stride & 3) || (reinterpret_cast<uintptr_t>(k.in_ssq) & 15) ||
          k.norm_size < 1)
        return hipErrorInvalidValue;
      ssq_bytes = size_t(rows) * ((size_t(k.in_parts) * 4 + 1023) >> 10 << 10);
      if (ssq_bytes > 32 * 1024) return hipErrorNotSupported;  // at most 32 one-KiB pieces (request counter budget)
      p.in_ssq = k.in_ssq;
      p.in_parts = uint32_t(k.in_parts), p.in_stride = uint32_t(k.in_stride);
      p.in_eps = k.eps, p.in_inv_size = 1.0f / float(k.norm_size);
    }
    if (k.out_ssq || k.out_gamma) {
      if (a.dual || nmat != 1 || (k.out_ssq && k.out_stride < w0->ntiles)) return hipErrorInvalidValue;
      p.out_gamma = k.out_gamma, p.out_ssq = k.out_ssq, p.out_stride = uint32_t(k.out_stride);
      p.out_ovf = k.out_gamma ? kvm_overflow_word() : nullptr;
    }
  }
  if (a.rope) {
    const ns_qkv_rope& r = *a.rope;
    if (mode != GV_MSEG || nmat != 3 || r.mode != 0 || r.head_size < 2 || (r.head_size & 1) || r.n_dims != r.head_size ||
        !r.kcache16 || !r.vcache16 || !r.cos_sin || r.n_past < 0 || p.mat[0].n != r.heads * r.head_size ||
        p.mat[1].n != r.heads_kv * r.head_size || p.mat[2].n != r.heads_kv * r.head_size)
      return hipErrorInvalidValue;
    p.rope.kc = static_cast<_Float16*>(r.kcache16), p.rope.vc = static_cast<_Float16*>(r.vcache16);
    p.rope.c_sl = r.cache_step_sl, p.rope.c_head = r.cache_step_head;
    p.rope.head_size = r.head_size, p.rope.n_past = r.n_past;
    p.rope.cos_sin = reinterpret_cast<const float2*>(r.cos_sin);
    p.rope.on = 1;
    if (a.rope_route) {
      const QkvRopeRoute& q = *a.rope_route;
      if (rows != 1) return hipErrorInvalidValue;
      p.rope.kmove = q.kmove, p.rope.kd_pos = q.kd_pos;
      p.rope.k32 = q.k32, p.rope.v32 = q.v32;
      p.rope.k32_head = q.k32_head, p.rope.k32_dim = q.k32_dim, p.rope.k32_tok = q.k32_tok;
      p.rope.v32_head = q.v32_head, p.rope.v32_dim = q.v32_dim, p.rope.v32_tok = q.v32_tok;
      p.rope.ovf = q.overflow;
    }
  }
  if (uint64_t(rows) * uint64_t(a.lda) * 4 >= (uint64_t(1) << 30)) return hipErrorNotSupported;  // staging offsets

  // waves per workgroup: enough waves on the chip to overlap dequantisation with the stream (as tuned for
  // smallm_kernel, profiles/r01*); the rings of a workgroup must fit in LDS beside the staged activations
  const int grid = int(tiles);
  const int nq = a.dual ? 2 : 1;
  const uint32_t sbytes = uint32_t(w0->sps) * (w0->scale_dt == DT_F32 ? 4u : 2u);
  const uint32_t slot = 1024u + 16u * sbytes + (w0->asym ? 16u * uint32_t(w0->sps) : 0u);
  auto ring_bytes = [&](int waves) {  // a wave's ring: one slot per item it can have in flight, at least the reduction scratch
    const uint32_t items = ((ks + uint32_t(waves) - 1) / uint32_t(waves)) * uint32_t(nq);
    const size_t b = size_t(std::min<uint32_t>(items, uint32_t(kGvPF))) * slot;
    return std::max<size_t>((b + 15) & ~size_t(15), size_t(nq) * 1024);
  };
  int nw = decode_waves(grid, int(ks), a.dual);
  {
    while (nw > 1 && ((a_bytes + 15) & ~size_t(15)) + ssq_bytes + size_t(nw) * ring_bytes(nw) > kGvMaxLds) nw /= 2;
  }
  uint32_t nw_log2 = 0;
  while ((1 << nw_log2) < nw) nw_log2++;
  if (a32 && uint64_t(rows) * ((uint64_t(ks) * uint32_t(kstep) * 4u + 1023u) >> 10) > uint64_t(nw) * kGvA32Regs) {
    if (i8q) return a.i8->quantized ? hipErrorNotSupported : hipErrorNotReady;  // (quantized beforehand it fits: the caller's retry takes XV = 3)
    return hipErrorNotSupported;  // more fp32 pieces than the waves hold in registers: smallm_kernel stages those
  }

  p.ks = ks;
  p.qstride = w0->qstride;
  p.nw_log2 = nw_log2;
  p.sstride = w0->sstride;
  p.zstride = w0->zstride;
  p.srows = uint32_t(w0->srows);
  {
    int mul, shift;
    if (!srow_params(w0, &mul, &shift)) return hipErrorNotSupported;
    p.srow_mul = uint32_t(mul), p.srow_shift = uint32_t(shift);
  }
  p.m = a.m;
  p.k = w0->k;
  p.lda = i8q ? a.i8->lda32 : i8s ? a.i8->ldq : a.lda;
  p.row_stride = row_stride;
  p.ssq_off = uint32_t((a_bytes + 15) & ~size_t(15));
  p.ring_off = p.ssq_off + uint32_t(ssq_bytes);
  p.c2 = a.c2;
  p.d = a.d;
  p.ldc = a.ldc;
  p.ldd = a.ldd;
  p.epilogue = a.epilogue;
  if (w0->kind == WK_F4) f4_lut_planes(w0->lut, &p.lut);
  p.f8 = f8_consts(w0->qtype);
#ifdef NS_TRACE
  p.trace = trace_buffer();
#endif
  p.ring_stride = uint32_t(ring_bytes(nw));
  const size_t lds = size_t(p.ring_off) + size_t(nw) * p.ring_stride;
  if (lds > kGvMaxLds) return hipErrorNotSupported;
  if (moe) {
    p.moe_table = static_cast<const MoeExpertRow*>(a.moe->table);
    p.moe_id = a.moe->id;
    p.moe_n = a.moe->n_as;
  }
  p.pl_bits = planes ? uint32_t(w0->pl_bits) : 0u;
  p.pl_lanes = planes ? w0->code_rec / 16u : 64u;
  const int mode_x = mode | (a32 && !moe ? kGvModeA32 : 0) | (i8s ? kGvModeI8 : 0) | (moe ? kGvModeMoe : 0) | (planes ? kGvModePlanes : 0);

#define NS_DISPATCH(KIND)                                                                       \
  switch (w0->sps) {                                                                            \
    case 4: return launch_gemv_s<KIND, 4>(p, w0->scale_dt, w0->asym, mode_x, grid, nw, lds, st);  \
    case 2: return launch_gemv_s<KIND, 2>(p, w0->scale_dt, w0->asym, mode_x, grid, nw, lds, st);  \
    default: return launch_gemv_s<KIND, 1>(p, w0->scale_dt, w0->asym, mode_x, grid, nw, lds, st); \
  }
  if (w0->kind == WK_INT4) {
    NS_DISPATCH(WK_INT4)
  } else if (w0->kind == WK_INT8) {
    if (w0->sps == 2) return launch_gemv_s<WK_INT8, 2>(p, w0->scale_dt, w0->asym, mode_x, grid, nw, lds, st);
    return launch_gemv_s<WK_INT8, 1>(p, w0->scale_dt, w0->asym, mode_x, grid, nw, lds, st);
  } else if (w0->kind == WK_F8) {  // device scales are always fp32 (E8M0 shared exponents are expanded at load)
    if (w0->sps == 2) return launch_gemv_a<WK_F8, 2, SK_F32>(p, false, mode_x, grid, nw, lds, st);
    return launch_gemv_a<WK_F8, 1, SK_F32>(p, false, mode_x, grid, nw, lds, st);
  } else {
    NS_DISPATCH(WK_F4)
  }
#undef NS_DISPATCH
}

void touch_gemv_module() {
  hipFuncAttributes fa;
  (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(gemv_kernel<WK_INT4, 4, SK_BF16, false, GV_PLAIN, 0>));
  (void)hipGetLastError();
}
}  // namespace ns
