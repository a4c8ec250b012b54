// ns_engine.hip — the batch-1 decode GEMV chain as ONE persistent launch ("decode engine") of libns_hip.so.
//
// Why: a decode GEMV launch on MI355X costs ~3.4 us that stream nothing (kernel boundary, cold caches, first byte,
// reduction tail) — 129 launches per Llama-2-7B token = 0.44 ms of the 1.04 ms chain with HBM
// idle.  Here one workgroup of 16 waves per CU lives for the whole token and every wave keeps gemv_kernel's private
// LDS-DMA ring (ns_gemv.hip) — but the ring runs AHEAD ACROSS OPERATOR BOUNDARIES: while an operator's input row is
// still being handed over between the workgroups, every wave already has its next kEngR records of the NEXT operator
// in flight or in LDS (weights do not depend on activations).  That run-ahead is what a launch boundary cannot have.
//
//   * per wave: records {1024 B codes | 128 B scales} HBM -> its ring by `buffer_load ... lds`, non-temporal, counted
//     vmcnt waits, consume = gemv_kernel's arithmetic unchanged (4 x v_mfma_f32_16x16x32_f16 on the raw codes, group
//     scale on the fp32 result); wave w owns k-steps w, w + 16, ... of a tile (fused gate/up: waves 0-7 the gate
//     matrix, 8-15 the up matrix, k-steps w & 7, + 8, ...)
//   * per tile the LAST wave to arrive adds the partial sums in wave order — bit for bit gemv_kernel's sum with 16 (fused:
//     8) waves per tile —, applies the epilogue, stores fp32 C and PUBLISHES the outputs as 8-byte {fp16 x 2, tag}
//     granules (one sc1 store each; cdna_hip_programming.md Guideline 16 form R2, MI355X_MICROARCH.md price list rows
//     allgather / prefetch-credit)
//   * hand-in: all 16 waves sweep a 1/16 slice of the producers' granules (relaxed agent-scope loads, re-read until
//     every tag matches), stage them as the fp16 activation row in LDS, one raw s_barrier (LDS only: the DMA queue
//     keeps running)
// Every spin is bounded and reports through the status word; granule tags carry an epoch kept in device memory, so a
// graph replay needs no per-launch memset.
//
// History (profiles/r03c, r03d): v0 / v1 used one loader wave + one gather wave + 8 consumer waves (the guide's
// engine shape); with int4 dequantisation the 8 consumers, not the stream, were the limit (0.45 us per record and
// wave) and a 1024-thread workgroup has no room for 16 consumers beside them.
//
// Arithmetic reference: bestla/bestla/kernel_ref.h:2489-2531 (gemv_4bit_fp32_fp32), :1027-1127 (decompress_kblock_s4_fp);
// fused gate/up: neural_speed/core/layers/ip_fusion_ffn.cpp:364-406.
//
// Envelope: one row, int4 symmetric weights with four bf16 group scales per 128-deep k-step (the Q4_0 headline
// format: interleaved records of 1152 B), K a multiple of 128.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <utility>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/ns_bestla.h"
#include "ns_common.h"
#include "ns_dev.h"

namespace ns {

// rounds of two records only for waves with an EVEN number of records per tile: a paired loop followed by a single
// record in the same tile ("pair, pair, single": the 7B down projection, 5 records for waves 6-15) gives wrong sums when
// built with hipcc 7.2 although it reads the same slots in the same order as five single rounds (bisected on the GPU:
// not the waits, not the scheduling barriers, not the cursor) — left as a compiler issue to revisit
#ifndef NS_ENG_PAIRCOND
#define NS_ENG_PAIRCOND ((nmine & 1u) == 0u)
#endif
#ifndef NS_ENG_R
#define NS_ENG_R 6
#endif
constexpr int kEngW = 16;                    // waves per workgroup = partial sums per tile (fused op: 2 x 8)
constexpr uint32_t kEngRec = 1152;           // bytes of one record in HBM and in a ring slot
constexpr int kEngR = NS_ENG_R;              // ring slots per wave = records a wave keeps requested ahead
constexpr int kEngPs = 8;                    // partial-sum slots (tiles a wave may run ahead of the slowest)
constexpr uint32_t kEngMaxK = 11008;         // longest input row staged (halves)
constexpr uint32_t kEngABytes = ((kEngMaxK * 2 + 255) / 256) * 256;
constexpr uint32_t kEngCtrlBytes = 256;
constexpr uint32_t kEngPsBytes = kEngPs * kEngW * 16 * 4;
constexpr uint32_t kEngLds = 160 * 1024;
// LDS map: control block | 16 rings | partial sums | two activation rows
constexpr uint32_t kEngRingOff = kEngCtrlBytes;
constexpr uint32_t kEngRingBytes = kEngR * kEngRec;
constexpr uint32_t kEngPsOff = (kEngRingOff + kEngW * kEngRingBytes + 255) / 256 * 256;
constexpr uint32_t kEngAOff = kEngPsOff + kEngPsBytes;
static_assert(kEngAOff + 2 * kEngABytes <= kEngLds, "LDS map");
static_assert(2 * kEngR <= 63, "vmcnt is a 6-bit counter");
constexpr size_t kEngWordsBytes = 64 + size_t(304) * 64 * 16 * 4;  // epoch, status + the trace area
constexpr uint32_t kEngSpinLimit = 1u << 19;  // sweeps before a wave gives up (~0.5 s)

enum EngIn : int32_t { ENG_IN_EXTERNAL = -1, ENG_IN_SAME = -2 };

struct EngOp {            // read with scalar loads; 64 bytes
  const uint8_t* w0;      // weight allocation(s): records at (tile * ks + s) * 1152
  const uint8_t* w1;
  float* c;               // fp32 output [n] (may be null)
  uint32_t ks, ntiles, nq, n;
  int32_t in;             // byte offset of the input's granule region in the arena / ENG_IN_*
  uint32_t in_tag;        // tag low bits of the producer (its op index + 1)
  int32_t out;            // byte offset of this op's granule region, -1: none
  uint32_t epi;           // enum ns_epilogue (fused gate/up: SILU / GELU)
  uint32_t tq, tr;        // ntiles / grid and ntiles % grid: workgroup b owns tiles [b * tq + min(b, tr), ... + tq + (b < tr))
};
static_assert(sizeof(EngOp) == 64, "EngOp is fetched as one 64-byte scalar load");

struct EngParams {
  const EngOp* ops;
  uint32_t nops;
  const void* x16;          // external input of op 0 (fp16 [k])
  uint8_t* arena;           // granule regions
  uint32_t* epoch;          // device word: token counter, tags = (epoch << 10) | in_tag
  uint32_t* status;         // device word: 0 ok, else a give-up code
  uint32_t* debug;          // NS_ENG_TRACE builds: stamp area
};

// LDS control block
struct EngCtrl {
  uint32_t arrive[kEngPs];  // per partial-sum slot: waves arrived for the tile that uses it now
  uint32_t gen[kEngPs];     // per slot: tiles finished in it (tile T may use slot T % kEngPs once gen == T / kEngPs)
};
static_assert(sizeof(EngCtrl) <= kEngCtrlBytes, "control block");

typedef __attribute__((address_space(3))) unsigned char* LdsB;
typedef __attribute__((address_space(3))) EngCtrl* LdsCtrl;
typedef __attribute__((address_space(3))) float* LdsF32;
typedef __attribute__((address_space(3))) uint32_t* LdsU32;
typedef __attribute__((address_space(1))) unsigned long long gu64;
// operator descriptors and the epoch word are read through the constant address space: scalar loads, never a vector
// load the compiler would wait for with vmcnt(0) (which would drain the DMA queue)
struct EngOpRaw {
  uint4v v[4];
};
__device__ __forceinline__ EngOp eng_op(const EngParams& p, uint32_t op) {
  typedef const __attribute__((address_space(4))) uint4v* CVec;
  const CVec q = reinterpret_cast<CVec>(reinterpret_cast<uintptr_t>(p.ops + op));
  EngOpRaw r;
#pragma unroll
  for (int i = 0; i < 4; i++) r.v[i] = q[i];
  return __builtin_bit_cast(EngOp, r);
}

#ifdef NS_ENG_TRACE
// [workgroup][op][8] 100 MHz stamps of wave 0 (7: of the wave that finished the workgroup's last tile):
// 0 hand-in starts, 1 input staged (after the barrier), 2 first record consumed, 3 last record consumed, 7 last tile published
#define ENG_STAMP(op, i)                                                                         \
  do {                                                                                           \
    if ((threadIdx.x & 63) == 0 && (op) < 64) p.debug[(size_t(blockIdx.x) * 64 + (op)) * 16 + (i) * 2] = uint32_t(wall_clock64()), \
        p.debug[(size_t(blockIdx.x) * 64 + (op)) * 16 + (i) * 2 + 1] = uint32_t(wall_clock64() >> 32);                               \
  } while (0)
#else
#define ENG_STAMP(op, i)
#endif
#define ENG_LDS_STORE(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)

__device__ __forceinline__ void eng_fail(const EngParams& p, uint32_t code, uint32_t op) {
  if ((threadIdx.x & 63) == 0) atomicOr(p.status, code | (op << 8) | (uint32_t(blockIdx.x) << 20));
}

// this workgroup's contiguous tile range of an operator: the first `tr` workgroups own one tile more (no division on
// the device: the cursor crosses operator boundaries inside the streaming loop)
__device__ __forceinline__ void eng_tiles(const EngOp& o, uint32_t& t0, uint32_t& t1) {
  const uint32_t b = blockIdx.x;
  t0 = b * o.tq + min(b, o.tr);
  t1 = t0 + o.tq + (b < o.tr ? 1u : 0u);
}

// ---------------------------------------------------------------------------------------------------------------
// the request cursor of a wave: walks ITS records of the whole token in consumption order — operator by operator,
// tile by tile, k-steps kfirst, kfirst + kstride, ... — kEngR records ahead of the arithmetic
// ---------------------------------------------------------------------------------------------------------------
struct EngCursor {
  Rsrc rsrc;
  uint32_t op, ks, kstride, kfirst;
  uint32_t tiles_left;  // tiles of this operator still to request after the current one
  uint32_t k;           // k-step of the next record to request
  uint32_t off;         // its byte offset in the weight allocation
  uint32_t tile_off;    // byte offset of the current tile's k-step kfirst
  bool done;
};

// position the cursor on the first record of operator `op` or the first later operator in which this wave has records
__device__ __forceinline__ void eng_cursor_seek(EngCursor& c, const EngParams& p, uint32_t op, uint32_t w) {
  for (;; op++) {
    if (op >= p.nops) {
      c.done = true;
      c.op = op;
      return;
    }
    const EngOp o = eng_op(p, op);
    uint32_t t0, t1;
    eng_tiles(o, t0, t1);
    const bool dual = o.nq > 1;
    const uint32_t kfirst = dual ? (w & 7u) : w, kstride = dual ? 8u : 16u;
    if (t1 == t0 || kfirst >= o.ks) continue;
    c.rsrc = make_rsrc((dual && w >= 8) ? o.w1 : o.w0, 0x80000000u);
    c.op = op, c.ks = o.ks, c.kstride = kstride, c.kfirst = kfirst;
    c.tiles_left = t1 - t0 - 1;
    c.k = kfirst;
    c.tile_off = (t0 * o.ks + kfirst) * kEngRec;
    c.off = c.tile_off;
    return;
  }
}

// request the cursor's record into ring slot `slot` of this wave and step the cursor
__device__ __forceinline__ void eng_issue(EngCursor& c, const EngParams& p, LdsB ring, uint32_t slot, uint32_t w, uint32_t voff, uint32_t l) {
#if defined(__HIP_DEVICE_COMPILE__)
  const LdsB dst = ring + slot * kEngRec;
  // the cursor is wave-uniform by construction; say so, or hipcc wraps every request in a readfirstlane loop
  const uint32_t off = __builtin_amdgcn_readfirstlane(c.off);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(c.rsrc, reinterpret_cast<__attribute__((address_space(3))) void*>(dst), 16, voff, off, 0, 2);
  if (l < 8)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(c.rsrc, reinterpret_cast<__attribute__((address_space(3))) void*>(dst + 1024), 16, voff,
                                             off + 1024, 0, 2);
#endif
  c.k += c.kstride;
  c.off += c.kstride * kEngRec;
  if (c.k >= c.ks) {  // next tile of the operator, or the next operator
    if (c.tiles_left) {
      c.tiles_left--;
      c.k = c.kfirst;
      c.tile_off += c.ks * kEngRec;
      c.off = c.tile_off;
    } else {
      eng_cursor_seek(c, p, c.op + 1, w);
    }
  }
  // keep the cursor in scalar registers on every path (hipcc otherwise gives `off` a vector home in one loop and a
  // scalar one in the next)
  c.k = __builtin_amdgcn_readfirstlane(c.k);
  c.off = __builtin_amdgcn_readfirstlane(c.off);
  c.tile_off = __builtin_amdgcn_readfirstlane(c.tile_off);
  c.tiles_left = __builtin_amdgcn_readfirstlane(c.tiles_left);
}

// ---------------------------------------------------------------------------------------------------------------
// hand-in: stage the operator's input row (fp16) in LDS; every wave sweeps a slice of the granules
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool eng_hand_in(const EngParams& p, const EngOp& o, uint32_t op, LdsB abuf, uint32_t w, uint32_t epoch) {
  const uint32_t l = threadIdx.x & 63;
  bool ok_all = true;
  if (o.in == ENG_IN_EXTERNAL) {
    const uint4v* src = static_cast<const uint4v*>(p.x16);
    for (uint32_t i = w * 64 + l; i * 8 < o.ks * 128u; i += kEngW * 64) *reinterpret_cast<__attribute__((address_space(3))) uint4v*>(abuf + i * 16) = src[i];
  } else {
    const gu64* G = (const gu64*)(p.arena + o.in);
    const uint32_t ng = o.ks * 64u;
    const uint32_t tag = (epoch << 10) | o.in_tag;
    const LdsU32 a32 = reinterpret_cast<LdsU32>(abuf);
    // granule gi belongs to wave (gi / 64) % 16: a wave's loads are 512-byte contiguous runs, 8 of them per pass
    for (uint32_t base = w * 64; base < ng; base += 8 * kEngW * 64) {
      for (uint32_t spins = 0;; spins++) {
        unsigned long long x[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
          const uint32_t gi = base + uint32_t(i) * (kEngW * 64) + l;
          x[i] = gi < ng ? __hip_atomic_load(G + gi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((unsigned long long)tag << 32);
        }
        bool ok = true;
#pragma unroll
        for (int i = 0; i < 8; i++) {
          const uint32_t gi = base + uint32_t(i) * (kEngW * 64) + l;
          const bool hit = uint32_t(x[i] >> 32) == tag;
          ok &= hit;
          if (hit && gi < ng) a32[gi] = uint32_t(x[i]);
        }
        if (__all(ok)) break;
        if (spins > kEngSpinLimit) {
          eng_fail(p, 2, op);
          ok_all = false;
          break;
        }
        __builtin_amdgcn_s_sleep(8);
      }
      if (!ok_all) break;
    }
  }
  // LDS writes done, then the workgroup barrier — written by hand: __syncthreads() would first wait for every DMA
  // request in flight (hipcc cannot know the rings are wave-private)
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  return ok_all;
}

// ---------------------------------------------------------------------------------------------------------------
// one operator: this wave's records of this workgroup's tiles
// ---------------------------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void wait_vmcnt_eng() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// what a wave holds of one record while it works on it
struct EngRec {
  uint4v q;
  uint32_t s0, s1;
  half8_t a[4];
};

template <bool DUAL>
__device__ __forceinline__ void eng_consume_op(const EngParams& p, LdsB smem, const EngOp& o, uint32_t op, uint32_t w, EngCursor& cur,
                                               uint32_t& slot, uint32_t& rslot, uint32_t& ahead, uint32_t& tile_seq, LdsB abuf,
                                               uint32_t epoch) {
  const LdsCtrl ctrl = reinterpret_cast<LdsCtrl>(smem);
  const uint32_t l = threadIdx.x & 63;
  const uint32_t nn = l & 15, g = l >> 4;
  const uint32_t voff = l * 16;
  const I4Consts i4c = {0x000f000fu, 0x00f000f0u, 0x64006400u};
  const LdsB ring = smem + kEngRingOff + w * kEngRingBytes;
  const uint32_t ring_q = uint32_t(reinterpret_cast<uintptr_t>(ring)) + l * 16u;
  const uint32_t ring_s = uint32_t(reinterpret_cast<uintptr_t>(ring)) + 1024u + nn * 8u;
  const uint32_t a_u32 = uint32_t(reinterpret_cast<uintptr_t>(abuf)) + 8u * g * 2u;
  const LdsF32 psum = reinterpret_cast<LdsF32>(smem + kEngPsOff);
  uint32_t t0, t1;
  eng_tiles(o, t0, t1);
  const uint32_t ks = o.ks;
  const uint32_t kfirst = DUAL ? (w & 7u) : w, kstride = DUAL ? 8u : 16u;
  const uint32_t nmine = kfirst < ks ? (ks - kfirst + kstride - 1) / kstride : 0u;  // this wave's records per tile
  using Corr = CorrRaw<4, SK_BF16, false>;

  auto fetch = [&](EngRec& r, uint32_t sl, uint32_t k) {  // LDS reads of the record in ring slot sl and of k-step k's activations
    const uint32_t ro = sl * kEngRec;
    typedef __attribute__((address_space(3))) const uint32_t* L32;
    r.q = *reinterpret_cast<const __attribute__((address_space(3))) uint4v*>(ring_q + ro);
    r.s0 = reinterpret_cast<L32>(ring_s + ro)[0];
    r.s1 = reinterpret_cast<L32>(ring_s + ro)[1];
#pragma unroll
    for (int jj = 0; jj < 4; jj++)
      r.a[jj] = *reinterpret_cast<const __attribute__((address_space(3))) half8_t*>(a_u32 + k * 256u + uint32_t(jj) * 64u);
  };
  auto compute = [&](const EngRec& r, floatx4& acc) {
    Corr cr;
    cr.s[0] = r.s0, cr.s[1] = r.s1;
    float sc[4], zp[4];
    corr_decode<4, SK_BF16, false, 4>(cr, sc, zp);
    const uint32_t xw[4] = {r.q.x, r.q.y, r.q.z, r.q.w};
    floatx4 dd[4];
#pragma unroll
    for (int jj = 0; jj < 4; jj++) {
      const _Float16 zl = (_Float16)(-1032.f), zh = (_Float16)(-72.f);
      const half8_t bq = cvt_i4x8(xw[jj], i4c, half2_t{zl, zl}, half2_t{zh, zh});
      dd[jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(r.a[jj], bq, floatx4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    }
    // all four result rows are carried although a one-row launch needs row 0 only: using ONE element of an MFMA result
    // lets hipcc recycle the other three registers while the MFMA that writes them is still in flight (ROCm 7.2: scale
    // words overwritten by late MFMA writes — garbage sums)
#pragma unroll
    for (int jj = 0; jj < 4; jj++) acc += dd[jj] * sc[jj];
  };

#ifdef NS_ENG_TRACE
  unsigned long long tw = 0, tc = 0, tis = 0, te = 0;
#define ENG_T(acc_) { const unsigned long long _t1 = __builtin_amdgcn_s_memtime(); acc_ += _t1 - _tl; _tl = _t1; }
#else
#define ENG_T(acc_)
#endif
  // ring bookkeeping of this wave: `ahead` records are requested and not yet consumed (they sit in slots slot,
  // slot + 1, ...); the next request goes into slot rslot.  Requests retire in order, so before consuming the oldest
  // `need` records at most ahead - need younger ones may still be in flight.
  auto wait_ahead = [&](uint32_t younger) {
    [&]<int... K>(std::integer_sequence<int, K...>) {
      (void)((younger == uint32_t(K) ? (wait_vmcnt_eng<2 * K>(), true) : false) || ...);
    }(std::make_integer_sequence<int, kEngR>{});
  };
  auto refill = [&]() {  // request until the ring is full again
    while (ahead < uint32_t(kEngR) && !cur.done) {
      eng_issue(cur, p, ring, rslot, w, voff, l);
      rslot = rslot + 1 == uint32_t(kEngR) ? 0u : rslot + 1;
      ahead++;
    }
  };
  const uint32_t ntl = t1 - t0;
  for (uint32_t ti = 0; ti < ntl; ti++) {
    floatx4 acc = floatx4{0.f, 0.f, 0.f, 0.f};
    uint32_t k = kfirst, left = nmine;
    // When to request: a request blocks the wave for 1000-2000 clocks while the CU's memory pipe is full
    // (profiles/r03h: Q / K / V / WO spent 60-85 % of their time in the two requests behind each pair of records).  So
    // the ring is refilled behind a record only while the cursor is still INSIDE this operator (a streaming operator:
    // more records than ring slots — the stream must not pause); once every record of the operator is requested the
    // wave computes straight through — also through the following operators that share the input — and refills at
    // the next hand-in, where it would wait anyway.
#define ENG_HOLD (cur.op != op)
#ifdef NS_ENG_TRACE
    unsigned long long _tl = __builtin_amdgcn_s_memtime();
#endif
    // two records per round: both are read from LDS before either is computed, the two dependency chains (LDS ->
    // dequantise -> MFMA -> scale) interleave
    for (; left >= 2 && NS_ENG_PAIRCOND; left -= 2, k += 2 * kstride) {
      if (ahead < 2) refill();
      wait_ahead(ahead - 2);
      ENG_T(tw)
      const uint32_t slot_b = slot + 1 == uint32_t(kEngR) ? 0u : slot + 1;
      EngRec ra, rb;
      fetch(ra, slot, k);
      fetch(rb, slot_b, k + kstride);
      compute(ra, acc);  // k ascending: the wave's sum keeps gemv_kernel's order
      compute(rb, acc);
      // both slots are free (their LDS reads completed: the values were used)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      ENG_T(tc)
      slot = slot_b + 1 == uint32_t(kEngR) ? 0u : slot_b + 1;
      ahead -= 2;
      if (!ENG_HOLD) refill();
      __builtin_amdgcn_sched_barrier(0);
      ENG_T(tis)
    }
    for (; left; left--, k += kstride) {
      if (ahead < 1) refill();
      wait_ahead(ahead - 1);
      EngRec ra;
      fetch(ra, slot, k);
      compute(ra, acc);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      slot = slot + 1 == uint32_t(kEngR) ? 0u : slot + 1;
      ahead -= 1;
      if (!ENG_HOLD) refill();
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- the tile is complete for this wave: park row 0 of its sums; the LAST wave to arrive finishes the tile ----
    const uint32_t T = tile_seq;  // tiles are numbered per workgroup, in the order every wave meets them
    const uint32_t psl = T & uint32_t(kEngPs - 1);
    const LdsF32 ps = psum + psl * (kEngW * 16);
    // nothing couples the waves' progress inside an operator: a wave that is kEngPs tiles ahead of the slowest waits here
    for (uint32_t spins = 0; __builtin_amdgcn_readfirstlane(__hip_atomic_load(&ctrl->gen[psl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) !=
                             T / uint32_t(kEngPs);
         spins++) {
      if (spins > kEngSpinLimit * 8) {
        eng_fail(p, 5, op);
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    if (g == 0) ps[w * 16 + nn] = acc[0];
    // Every lane's four result rows are the same number (all sixteen A rows of a one-row launch hold the same
    // activations).  Checking that keeps ALL FOUR registers of every MFMA result live: when only element 0 is used, hipcc
    // (ROCm 7.2) hands the other three registers to later instructions while the MFMA that writes them is still in
    // flight, and its late write corrupts them (seen twice: scale words in v0, the paired loop in v2) — and it is a
    // cheap self-test of the LDS staging on top.
    {
      const uint32_t b0 = __builtin_bit_cast(uint32_t, acc[0]);
      const uint32_t df = (b0 ^ __builtin_bit_cast(uint32_t, acc[1])) | (b0 ^ __builtin_bit_cast(uint32_t, acc[2])) |
                          (b0 ^ __builtin_bit_cast(uint32_t, acc[3]));
      if (df) atomicOr(p.status, 0x80u);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    uint32_t old = 0;
    if (l == 0) old = __hip_atomic_fetch_add(&ctrl->arrive[psl], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    old = __builtin_amdgcn_readfirstlane(old);
    tile_seq++;
    if (old != uint32_t(kEngW - 1)) continue;
    if (l == 0) ENG_LDS_STORE(&ctrl->arrive[psl], 0u);
    asm volatile("" ::: "memory");
    float v;
    if constexpr (DUAL) {
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int x = 0; x < 8; x++) s0 += ps[x * 16 + int(nn)];
#pragma unroll
      for (int x = 8; x < 16; x++) s1 += ps[x * 16 + int(nn)];
      // tmp1 = act(A*W1) ; out = (A*W3) * tmp1   (ip_fusion_ffn.cpp:364-406)
      const float t1v = (o.epi == NS_EPI_SILU) ? epi_silu(s0) : epi_gelu(s0);
      v = s1 * t1v;
    } else {
      float s0 = 0.f;
#pragma unroll
      for (int x = 0; x < 16; x++) s0 += ps[x * 16 + int(nn)];
      v = s0;
      if (o.epi == NS_EPI_GELU) v = epi_gelu(v);
      else if (o.epi == NS_EPI_SILU) v = epi_silu(v);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the partial sums are read: the slot may be reused
    if (l == 0) ENG_LDS_STORE(&ctrl->gen[psl], T / uint32_t(kEngPs) + 1u);
    const uint32_t col = (t0 + ti) * 16 + nn;
    const bool okc = col < o.n && g == 0;
    if (okc && o.c) reinterpret_cast<__attribute__((address_space(1))) float*>(reinterpret_cast<uintptr_t>(o.c))[col] = v;
    if (o.out >= 0) {
      const _Float16 h = okc ? (_Float16)v : (_Float16)0.f;
      const uint32_t hb = uint32_t(__builtin_bit_cast(unsigned short, h));
      const uint32_t hn = uint32_t(__shfl_xor(int(hb), 1, 64));
      if (g == 0 && !(nn & 1) && col < o.n) {
        const unsigned long long gran = ((unsigned long long)((epoch << 10) | (op + 1)) << 32) | (hb | (hn << 16));
        __hip_atomic_store((gu64*)(p.arena + o.out) + (col >> 1), gran, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
#ifdef NS_ENG_DRAIN
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
#ifdef NS_ENG_TRACE
    ENG_T(te)
#endif
    // (these stores share the request counter with the ring's loads but cannot weaken the counted waits: a wait for
    // "at most N outstanding" bounds the outstanding LOADS by N whatever the stores do, and loads retire in order)
    if (ti + 1 == t1 - t0) ENG_STAMP(op, 7);
  }
#undef ENG_HOLD
#ifdef NS_ENG_TRACE
  if (w == 0 && l == 0 && op < 64) {  // durations of wave 0 in this operator, shader clocks: wait, LDS + arithmetic, requests
    unsigned long long* d = reinterpret_cast<unsigned long long*>(p.debug) + (size_t(blockIdx.x) * 64 + op) * 8;
    d[4] = tw, d[5] = tc, d[6] = tis;
  }
#endif
}

__global__ __launch_bounds__(kEngW * 64) void engine_kernel(const EngParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_generic[];
  const LdsB smem = (LdsB)(smem_generic);
  const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t l = threadIdx.x & 63;
  if (threadIdx.x < sizeof(EngCtrl) / 4) reinterpret_cast<LdsU32>(smem)[threadIdx.x] = 0;
  const uint32_t epoch =
      *reinterpret_cast<const __attribute__((address_space(4))) uint32_t*>(reinterpret_cast<uintptr_t>(p.epoch)) & 0x3fffffu;
  __syncthreads();
  // ---- fill this wave's ring: its first kEngR records of the token ----
  EngCursor cur;
  cur.done = false;
  eng_cursor_seek(cur, p, 0, w);
  const LdsB ring = smem + kEngRingOff + w * kEngRingBytes;
  uint32_t slot = 0, rslot = 0, ahead = 0, tile_seq = 0, gseq = 0;
  while (ahead < uint32_t(kEngR) && !cur.done) {
    eng_issue(cur, p, ring, rslot, w, l * 16, l);
    rslot = rslot + 1 == uint32_t(kEngR) ? 0u : rslot + 1;
    ahead++;
  }
  bool alive = true;
  for (uint32_t op = 0; op < p.nops; op++) {
    const EngOp o = eng_op(p, op);
    if (o.in != ENG_IN_SAME) {
      gseq++;
      if (w == 0) ENG_STAMP(op, 0);
      const LdsB abuf_new = smem + kEngAOff + (gseq & 1) * kEngABytes;
      while (ahead < uint32_t(kEngR) && !cur.done) {  // refill the ring under the hand-in wait
        eng_issue(cur, p, ring, rslot, w, l * 16, l);
        rslot = rslot + 1 == uint32_t(kEngR) ? 0u : rslot + 1;
        ahead++;
      }
      if (alive) alive = eng_hand_in(p, o, op, abuf_new, w, epoch);
      else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // keep the barrier count of the workgroup
      if (w == 0) ENG_STAMP(op, 1);
    }
    const LdsB abuf = smem + kEngAOff + (gseq & 1) * kEngABytes;
    if (w == 0) ENG_STAMP(op, 2);
    if (o.nq == 2) eng_consume_op<true>(p, smem, o, op, w, cur, slot, rslot, ahead, tile_seq, abuf, epoch);
    else eng_consume_op<false>(p, smem, o, op, w, cur, slot, rslot, ahead, tile_seq, abuf, epoch);
    if (w == 0) ENG_STAMP(op, 3);
  }
  // the token is over when workgroup 0 is: every workgroup has long read the epoch by then (its outputs were needed on
  // the way), so the next launch's tags can be armed
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(p.epoch, 1u);
}

}  // namespace ns

// =================================================================================================================
// host side
// =================================================================================================================
struct ns_engine {
  std::vector<ns::EngOp> ops;
  ns::EngOp* d_ops = nullptr;
  uint8_t* arena = nullptr;
  size_t arena_bytes = 0;
  uint32_t* words = nullptr;  // [0] epoch, [1] status, [16..] debug dump
  int grid = 0;
  int device = 0;
  ns::EngParams params;
};

extern "C" {

ns_engine* ns_hip_engine_create(const ns_engine_op* ops, int nops, const void* x16) {
  using namespace ns;
  if (!ops || nops < 1 || !x16) {
    set_error("engine: null argument");
    return nullptr;
  }
  ns_engine* e = new ns_engine();
  size_t arena = 0;
  std::vector<int32_t> out_off(size_t(nops), -1);
  // which ops feed a later one
  std::vector<char> feeds(size_t(nops), 0);
  for (int i = 0; i < nops; i++)
    if (ops[i].input >= 0) {
      if (ops[i].input >= i) {
        set_error("engine: an operator's input must be an earlier operator");
        delete e;
        return nullptr;
      }
      feeds[size_t(ops[i].input)] = 1;
    }
  for (int i = 0; i < nops; i++) {
    const ns_engine_op& s = ops[i];
    const ns_weight* w = s.w0;
    auto bad = [&](const char* m) {
      set_error(std::string("engine: operator ") + std::to_string(i) + ": " + m);
      delete e;
      return static_cast<ns_engine*>(nullptr);
    };
    if (!w) return bad("null weight");
    for (const ns_weight* x : {s.w0, s.w1}) {
      if (!x) continue;
      if (x->kind != WK_INT4 || x->asym || x->sps != 4 || x->scale_dt != DT_BF16 || !x->interleaved || x->qstride != kEngRec ||
          x->s_off != 1024 || !x->single_span || x->alloc_bytes >= (size_t(1) << 31) || x->shuf)
        return bad("format outside the engine's envelope (int4 symmetric, group 32, bf16 scales)");
      if (x->k != w->k || x->n != w->n || x->alloc_bytes != w->alloc_bytes) return bad("the two matrices of a fused operator differ in shape");
      if (x->k % 128 != 0 || uint32_t(x->k) > kEngMaxK || x->ksteps < 1) return bad("K outside the engine's envelope");
    }
    EngOp o;
    memset(&o, 0, sizeof(o));
    o.w0 = reinterpret_cast<const uint8_t*>(s.w0->codes);
    o.w1 = s.w1 ? reinterpret_cast<const uint8_t*>(s.w1->codes) : o.w0;
    o.c = s.c;
    o.ks = uint32_t(w->ksteps), o.ntiles = uint32_t(w->ntiles), o.nq = s.w1 ? 2u : 1u, o.n = uint32_t(w->n);
    o.epi = uint32_t(s.epilogue);
    if (s.w1 && s.epilogue != NS_EPI_SILU && s.epilogue != NS_EPI_GELU) return bad("a fused gate/up operator needs SILU or GELU");
    if (!s.w1 && s.epilogue != NS_EPI_NONE && s.epilogue != NS_EPI_SILU && s.epilogue != NS_EPI_GELU) return bad("epilogue not supported");
    if (s.input == -1) {
      o.in = ENG_IN_EXTERNAL;
    } else if (s.input == -2) {
      if (i == 0 || ops[i - 1].w0->k != w->k) return bad("'same input' needs a previous operator of the same K");
      o.in = ENG_IN_SAME;
    } else {
      const ns_weight* pw = ops[s.input].w0;
      if (pw->n < w->k) return bad("the producing operator has fewer outputs than this one has inputs");
      o.in = out_off[size_t(s.input)];
      o.in_tag = uint32_t(s.input) + 1;
    }
    o.out = -1;
    if (feeds[size_t(i)]) {
      if (w->n & 1) return bad("an operator that feeds another needs an even N");
      out_off[size_t(i)] = int32_t(arena);
      o.out = int32_t(arena);
      arena += (size_t(w->n) * 4 + 255) & ~size_t(255);  // one 8-byte granule per two outputs
    }
    e->ops.push_back(o);
  }
  if (nops > 1000) {
    set_error("engine: at most 1000 operators (10-bit tag)");
    delete e;
    return nullptr;
  }
  hipDeviceProp_t prop;
  int dev = 0;
  bool ok = hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess;
  e->device = dev;
  e->grid = ok ? prop.multiProcessorCount : 0;
  if (const char* g = getenv("NS_ENG_GRID")) e->grid = atoi(g);
  for (EngOp& o : e->ops)
    if (e->grid > 0) o.tq = o.ntiles / uint32_t(e->grid), o.tr = o.ntiles % uint32_t(e->grid);
  e->arena_bytes = std::max<size_t>(arena, 256);
  ok = ok && e->grid > 0 && hipMalloc(reinterpret_cast<void**>(&e->d_ops), e->ops.size() * sizeof(EngOp)) == hipSuccess &&
       hipMalloc(reinterpret_cast<void**>(&e->arena), e->arena_bytes) == hipSuccess &&
       hipMalloc(reinterpret_cast<void**>(&e->words), kEngWordsBytes) == hipSuccess &&
       hipMemcpy(e->d_ops, e->ops.data(), e->ops.size() * sizeof(EngOp), hipMemcpyHostToDevice) == hipSuccess &&
       hipMemset(e->arena, 0, e->arena_bytes) == hipSuccess && hipMemset(e->words, 0, kEngWordsBytes) == hipSuccess;
  const uint32_t one = 1;
  ok = ok && hipMemcpy(e->words, &one, 4, hipMemcpyHostToDevice) == hipSuccess;  // epoch starts at 1: tag 0 is "never written"
  ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(engine_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 int(kEngLds)) == hipSuccess;
  if (!ok) {
    set_error("engine: device set-up failed");
    ns_hip_engine_destroy(e);
    return nullptr;
  }
  e->params.ops = e->d_ops;
  e->params.nops = uint32_t(e->ops.size());
  e->params.x16 = x16;
  e->params.arena = e->arena;
  e->params.epoch = e->words;
  e->params.status = e->words + 1;
  e->params.debug = e->words + 16;
  return e;
}

int ns_hip_engine_launch(ns_engine* e, void* stream) {
  if (!e) return -1;
  hipLaunchKernelGGL(ns::engine_kernel, dim3(e->grid), dim3(ns::kEngW * 64), ns::kEngLds, (hipStream_t)stream, e->params);
  if (hipGetLastError() != hipSuccess) {
    ns::set_error("engine: launch failed");
    return -1;
  }
  return 0;
}

/* NS_ENG_TRACE builds: copies the stamp area [workgroups][64 ops][8 stamps] (100 MHz ticks, uint64) to host */
int ns_hip_engine_trace(ns_engine* e, unsigned long long* out, int nwg) {
  if (!e || !out) return -1;
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  return hipMemcpy(out, e->words + 16, size_t(nwg) * 64 * 16 * 4, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}

/* 0 = every launch so far ran to the end; otherwise the first give-up code (low byte: 1 loader / ring space, 2 gather,
 * 3 consumer / record, 4 consumer / input; bits 8..19 operator, 20.. workgroup).  Synchronises the device. */
unsigned ns_hip_engine_status(ns_engine* e) {
  if (!e) return ~0u;
  uint32_t st = 0;
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(&st, e->words + 1, 4, hipMemcpyDeviceToHost) != hipSuccess) return ~0u;
  return st;
}

void ns_hip_engine_destroy(ns_engine* e) {
  if (!e) return;
  if (e->d_ops) hipFree(e->d_ops);
  if (e->arena) hipFree(e->arena);
  if (e->words) hipFree(e->words);
  delete e;
}

}  // extern "C"
