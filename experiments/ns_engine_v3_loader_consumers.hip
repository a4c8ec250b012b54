// ns_engine.hip — the batch-1 decode GEMV chain as ONE persistent launch ("decode engine") of libns_hip.so.
//
// Why: a decode GEMV launch on MI355X costs ~3.4 us that stream nothing (kernel boundary, cold caches, first byte,
// reduction tail) — 129 launches per Llama-2-7B token = 0.44 ms of the 1.04 ms chain with HBM
// idle.  Here one workgroup per CU lives for the whole token: a LOADER wave streams this CU's share of every operator's
// weight records HBM -> LDS ring by DMA, in consumption order, steadily, ACROSS operator boundaries — while an
// operator's input row is still being handed over between the workgroups its weights (which do not depend on the
// activations) keep arriving.  That run-ahead is what a launch boundary cannot have.  Recipe: MI355X_MICROARCH.md
// "Persistent kernels" price list (rows prefetch-credit, allgather, ldsdma-fill, engine-vs-launches),
// cdna_hip_programming.md Guideline 16 form R2.
//
//   * wave 0            loader.  The unit of the stream is a ROW: the 8 consecutive k-steps of one tile that the 8
//                       consumers take side by side = 8 x 1152 B = nine 1 KiB requests (`buffer_load ... lds`,
//                       non-temporal), contiguous in HBM and in the ring.  kEngD rows in flight (counted vmcnt), `filled`
//                       published in LDS, ring space from the consumers' progress words.  The loader is the only wave
//                       that ever waits at a request: a request blocks its wave for 1000-2000 clocks whenever the CU's
//                       memory pipe is full (which is the normal state of a stream at the HBM limit), and a wave that
//                       also computes would stand still with it (profiles/r03h).
//   * waves 1 .. 8      consumers: gemv_kernel's arithmetic (ns_gemv.hip) unchanged — per record 4 x
//                       v_mfma_f32_16x16x32_f16 on the raw codes, group scale on the fp32 result; consumer c owns k-steps
//                       c, c + 8, ... of a tile (fused gate/up: both matrices' records of its k-steps).  Per tile the LAST
//                       consumer to arrive adds the 8 partial sums in consumer order — bit for bit gemv_kernel's sum with
//                       8 waves per tile —, applies the epilogue, stores fp32 C and PUBLISHES the outputs as 8-byte
//                       {fp16 x 2, tag} granules (one sc1 store each).
//   * hand-in           the 8 consumers sweep a 1/8 slice each of the granules the producers of the operator's input
//                       published (relaxed agent-scope loads, re-read until every tag matches) and stage them as the fp16
//                       activation row in LDS; an LDS counter is their barrier (the loader never stops).
// Every spin is bounded and reports through the status word; granule tags carry an epoch kept in device memory, so a
// graph replay needs no per-launch memset.
//
// History (profiles/r03c ... r03j, experiments/ns_engine_v2_symmetric.hip): v1 had this shape with ONE gather wave and a
// slow consumer loop (0.45 us per record); v2 used 16 symmetric waves, each with gemv_kernel's private ring — every wave
// then blocks at its own requests and the arithmetic (316 clocks per record and wave) waits with it.
//
// Arithmetic reference: bestla/bestla/kernel_ref.h:2489-2531 (gemv_4bit_fp32_fp32), :1027-1127 (decompress_kblock_s4_fp);
// fused gate/up: neural_speed/core/layers/ip_fusion_ffn.cpp:364-406.
//
// Envelope: one row, int4 symmetric weights with four bf16 group scales per 128-deep k-step (the Q4_0 headline
// format: interleaved records of 1152 B), K a multiple of 128.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>

#include "../../include/ns_bestla.h"
#include "ns_common.h"
#include "ns_dev.h"

namespace ns {

#ifndef NS_ENG_D
#define NS_ENG_D 5
#endif
constexpr int kEngC = 8;                     // consumer waves per workgroup = partial sums per tile
constexpr int kEngWaves = kEngC + 1;         // + the loader
constexpr uint32_t kEngRec = 1152;           // bytes of one record in HBM and in the ring
constexpr uint32_t kEngRow = kEngC * kEngRec;        // 9216
constexpr int kEngRowPieces = int(kEngRow / 1024);   // 9
static_assert(kEngRow % 1024 == 0, "a row must be a whole number of 1 KiB pieces");
constexpr int kEngD = NS_ENG_D;              // ROWS the loader keeps in flight (9 requests each: vmcnt is 6 bits)
constexpr int kEngPs = 8;                    // partial-sum slots (tiles a consumer may run ahead of the slowest)
constexpr uint32_t kEngMaxK = 11008;         // longest input row staged (halves)
constexpr uint32_t kEngABytes = ((kEngMaxK * 2 + 255) / 256) * 256;
constexpr uint32_t kEngCtrlBytes = 256;
constexpr uint32_t kEngPsBytes = kEngPs * 2 * kEngC * 16 * 4;
constexpr uint32_t kEngLds = 160 * 1024;
// LDS map: control block | ring | partial sums | two activation rows
constexpr uint32_t kEngRingOff = kEngCtrlBytes;
constexpr uint32_t kEngNRow = (kEngLds - kEngCtrlBytes - kEngPsBytes - 2 * kEngABytes) / kEngRow;  // ring slots (rows)
constexpr uint32_t kEngPsOff = (kEngRingOff + kEngNRow * kEngRow + 255) / 256 * 256;
constexpr uint32_t kEngAOff = kEngPsOff + kEngPsBytes;
static_assert(kEngAOff + 2 * kEngABytes <= kEngLds, "LDS map");
static_assert(kEngRowPieces * kEngD <= 63, "vmcnt is a 6-bit counter");
static_assert(kEngNRow >= uint32_t(kEngD + 4), "ring too small for the in-flight window");
constexpr size_t kEngWordsBytes = 64 + size_t(304) * 64 * 16 * 4;  // epoch, status + the trace area
constexpr uint32_t kEngSpinLimit = 1u << 21;  // polls before a wave gives up

enum EngIn : int32_t { ENG_IN_EXTERNAL = -1, ENG_IN_SAME = -2 };

struct EngOp {            // read with scalar loads; 64 bytes
  const uint8_t* w0;      // weight allocation(s): records at (tile * ks + s) * 1152
  const uint8_t* w1;
  float* c;               // fp32 output [n] (may be null)
  uint32_t ks, wbytes, nq, n;  // wbytes: bytes of a weight allocation = the bound of its buffer descriptor
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
  uint32_t filled;          // rows landed (FIFO index), written by the loader
  uint32_t hin;             // hand-in barrier: consumers that staged their slice, all hand-ins so far
  uint32_t gathering;       // a hand-in is in progress: the loader keeps ONE row in flight (MI355X_MICROARCH.md row gather-pass)
  uint32_t pad0[1];
  uint32_t freed[kEngC];    // per consumer: every row below this FIFO index is behind it
  uint32_t arrive[kEngPs];  // per partial-sum slot: consumers arrived for the tile that uses it now
  uint32_t gen[kEngPs];     // per slot: tiles finished in it (tile T may use slot T % kEngPs once gen == T / kEngPs)
};
static_assert(sizeof(EngCtrl) <= kEngCtrlBytes, "control block");

typedef __attribute__((address_space(3))) unsigned char* LdsB;
typedef __attribute__((address_space(3))) EngCtrl* LdsCtrl;
typedef __attribute__((address_space(3))) float* LdsF32;
typedef __attribute__((address_space(3))) uint32_t* LdsU32;
typedef __attribute__((address_space(1))) unsigned long long gu64;
// operator descriptors and the epoch word are read through the constant address space: scalar loads, never a vector
// load the compiler would wait for with vmcnt(0) (which would drain the loader's DMA queue)
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

// The LOADER's own LDS words go through inline asm: hipcc orders every LDS access it can see behind the wave's pending
// LDS-DMA writes with s_waitcnt vmcnt(0) (it cannot tell the control words from the ring), which would drain the DMA
// queue at every row (cdna_hip_programming.md section 5.7 item 1: asm memory operations are invisible to that pass).
__device__ __forceinline__ void lds_store_asm(uint32_t addr, uint32_t v) {
  asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ uint32_t lds_load_asm(uint32_t addr) {
  uint32_t v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory");
  return v;
}
#ifdef NS_ENG_TRACE
// [workgroup][op][8] 100 MHz stamps: 0 loader first request, 1 loader last request, 2 consumer 0 starts the hand-in,
// 3 input staged (hand-in barrier passed), 6 consumer 0 done with its records, 7 last tile published; 4 / 5: consumer 0's
// shader clocks spent waiting for rows / in LDS reads + arithmetic
#define ENG_STAMP(op, i)                                                                         \
  do {                                                                                           \
    if ((threadIdx.x & 63) == 0 && (op) < 64) p.debug[(size_t(blockIdx.x) * 64 + (op)) * 16 + (i) * 2] = uint32_t(wall_clock64()), \
        p.debug[(size_t(blockIdx.x) * 64 + (op)) * 16 + (i) * 2 + 1] = uint32_t(wall_clock64() >> 32);                               \
  } while (0)
#else
#define ENG_STAMP(op, i)
#endif
#define ENG_LDS_LOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define ENG_LDS_STORE(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)

__device__ __forceinline__ void eng_fail(const EngParams& p, uint32_t code, uint32_t op) {
  if ((threadIdx.x & 63) == 0) atomicOr(p.status, code | (op << 8) | (uint32_t(blockIdx.x) << 20));
}

// this workgroup's contiguous tile range of an operator: the first `tr` workgroups own one tile more
__device__ __forceinline__ void eng_tiles(const EngOp& o, uint32_t& t0, uint32_t& t1) {
  const uint32_t b = blockIdx.x;
  t0 = b * o.tq + min(b, o.tr);
  t1 = t0 + o.tq + (b < o.tr ? 1u : 0u);
}

// ---------------------------------------------------------------------------------------------------------------
// loader
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void eng_loader(const EngParams& p, LdsB smem) {
  const LdsCtrl ctrl = reinterpret_cast<LdsCtrl>(smem);
  const uint32_t l = threadIdx.x & 63;
  const uint32_t voff = l * 16;
  const LdsB ring = smem + kEngRingOff;
  const uint32_t a_filled = uint32_t(reinterpret_cast<uintptr_t>(&ctrl->filled));
  const uint32_t a_freed = uint32_t(reinterpret_cast<uintptr_t>(&ctrl->freed[0])) + (l < uint32_t(kEngC) ? l : 0u) * 4u;
  const uint32_t a_gath = uint32_t(reinterpret_cast<uintptr_t>(&ctrl->gathering));
  uint32_t issued = 0, minfreed = 0, slot = 0;  // in rows
  bool dead = false;
  for (uint32_t op = 0; op < p.nops && !dead; op++) {
    const EngOp o = eng_op(p, op);
    uint32_t t0, t1;
    eng_tiles(o, t0, t1);
    const Rsrc r0 = make_rsrc(o.w0, o.wbytes);
    const Rsrc r1 = make_rsrc(o.nq > 1 ? o.w1 : o.w0, o.wbytes);
    const uint32_t nkr = (o.ks + uint32_t(kEngC) - 1) / uint32_t(kEngC);
    ENG_STAMP(op, 0);
    for (uint32_t t = t0; t < t1 && !dead; t++) {
      uint32_t off = t * o.ks * kEngRec;
      for (uint32_t kr = 0; kr < nkr && !dead; kr++, off += kEngRow) {
        for (uint32_t q = 0; q < o.nq; q++) {
          if (issued - minfreed >= kEngNRow) {  // ring full: wait for the slowest consumer
            for (uint32_t spins = 0;; spins++) {
              uint32_t v = lds_load_asm(a_freed);  // lanes >= C re-read consumer 0's word: harmless for a minimum
#pragma unroll
              for (int o2 = 1; o2 < 8; o2 <<= 1) v = min(v, uint32_t(__shfl_xor(int(v), o2, 64)));
              minfreed = __builtin_amdgcn_readfirstlane(v);
              if (issued - minfreed < kEngNRow) break;
              if (spins > kEngSpinLimit) {
                eng_fail(p, 1, op);
                dead = true;
                break;
              }
              if ((spins & 63) == 63) {  // nothing will be requested for a while: let everything land and say so
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (l == 0) lds_store_asm(a_filled, issued);
              }
              __builtin_amdgcn_s_sleep(2);
            }
          }
#ifndef NS_ENG_NOTHIN
          // while the consumers sweep the granules of a hand-in their loads queue behind this wave's requests in the
          // CU's memory pipe: keep one row in flight instead of kEngD until they are done
          if (__builtin_amdgcn_readfirstlane(lds_load_asm(a_gath))) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kEngRowPieces) : "memory");
#endif
#if defined(__HIP_DEVICE_COMPILE__)
          const LdsB dst = ring + slot * kEngRow;
          const Rsrc rq = q ? r1 : r0;
          // nine 1 KiB pieces; the last row of a tile whose k-steps are not a multiple of 8 reads on into the next tile's
          // records (or past the matrix: the descriptor's bound returns zeros) — never consumed
#pragma unroll
          for (int pc = 0; pc < kEngRowPieces; pc++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, reinterpret_cast<__attribute__((address_space(3))) void*>(dst + pc * 1024), 16, voff,
                                                     off + uint32_t(pc) * 1024u, 0, 2);
#endif
          issued++;
          slot = slot + 1 == kEngNRow ? 0 : slot + 1;
          // requests retire in order: everything older than the youngest kEngD rows has landed
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kEngRowPieces * kEngD) : "memory");
          if (l == 0 && issued > uint32_t(kEngD)) lds_store_asm(a_filled, issued - uint32_t(kEngD));
        }
      }
    }
    ENG_STAMP(op, 1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (l == 0) lds_store_asm(a_filled, issued);
}

// ---------------------------------------------------------------------------------------------------------------
// consumers
// ---------------------------------------------------------------------------------------------------------------
struct EngState {   // a consumer's wave-uniform state
  uint32_t fcache;  // rows known to have landed
  uint32_t tile_seq;
  uint32_t hins;    // hand-ins so far
  bool dead;        // a bounded wait gave up: stop waiting anywhere (the status word says so; results are void)
};

// hand-in: stage the operator's input row (fp16) in LDS; each consumer sweeps a slice of the granules
__device__ __forceinline__ void eng_hand_in(const EngParams& p, const EngOp& o, uint32_t op, LdsB smem, LdsB abuf, uint32_t cw,
                                            uint32_t epoch, EngState& st) {
  const LdsCtrl ctrl = reinterpret_cast<LdsCtrl>(smem);
  const uint32_t l = threadIdx.x & 63;
  const uint32_t kk = o.ks * 128u;
  if (o.in == ENG_IN_EXTERNAL) {
    const uint4v* src = static_cast<const uint4v*>(p.x16);
    for (uint32_t i = cw * 64 + l; i * 8 < kk; i += kEngC * 64) *reinterpret_cast<__attribute__((address_space(3))) uint4v*>(abuf + i * 16) = src[i];
  } else {
    const gu64* G = (const gu64*)(p.arena + o.in);
    const uint32_t ng = kk >> 1;
    const uint32_t tag = (epoch << 10) | o.in_tag;
    const LdsU32 a32 = reinterpret_cast<LdsU32>(abuf);
    // granule gi belongs to consumer (gi / 64) % 8: a wave's loads are 512-byte contiguous runs, up to kSw per pass
    // (11 cover the longest row, 11008 values)
    constexpr int kSw = 11;
    for (uint32_t base = cw * 64; base < ng && !st.dead; base += kSw * kEngC * 64) {
      for (uint32_t spins = 0;; spins++) {
        unsigned long long x[kSw];
#pragma unroll
        for (int i = 0; i < kSw; i++) {
          const uint32_t gi = base + uint32_t(i) * (kEngC * 64) + l;
          x[i] = gi < ng ? __hip_atomic_load(G + gi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((unsigned long long)tag << 32);
        }
        bool ok = true;
#pragma unroll
        for (int i = 0; i < kSw; i++) {
          const uint32_t gi = base + uint32_t(i) * (kEngC * 64) + l;
          const bool hit = uint32_t(x[i] >> 32) == tag;
          ok &= hit;
          if (hit && gi < ng) a32[gi] = uint32_t(x[i]);
        }
#ifdef NS_ENG_NOWAIT  // diagnostics: the mechanism's cost without the wait for the producers (results void)
        break;
#endif
        if (__all(ok)) break;
        if (spins > kEngSpinLimit / 16) {
          eng_fail(p, 2, op);
          st.dead = true;
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
    }
  }
  // barrier of the 8 consumers over the staged row: an LDS counter (the loader takes no part and never stops)
  st.hins++;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  if (l == 0) __hip_atomic_fetch_add(&ctrl->hin, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  for (uint32_t spins = 0; !st.dead; spins++) {
    if (__builtin_amdgcn_readfirstlane(ENG_LDS_LOAD(&ctrl->hin)) >= st.hins * uint32_t(kEngC)) break;
    if (spins > kEngSpinLimit) {
      eng_fail(p, 4, op);
      st.dead = true;
    }
    __builtin_amdgcn_s_sleep(1);
  }
  asm volatile("" ::: "memory");
}

// what a consumer holds of one record while it works on it
struct EngRec {
  uint4v q;
  uint32_t s0, s1;
};

template <int NQ>
__device__ __forceinline__ void eng_consume_op(const EngParams& p, LdsB smem, const EngOp& o, uint32_t op, uint32_t cw, uint32_t rbase,
                                               LdsB abuf, uint32_t epoch, EngState& st) {
  const LdsCtrl ctrl = reinterpret_cast<LdsCtrl>(smem);
  const uint32_t l = threadIdx.x & 63;
  const uint32_t nn = l & 15, g = l >> 4;
  const I4Consts i4c = {0x000f000fu, 0x00f000f0u, 0x64006400u};
  // this consumer's record inside a ring row, per lane: codes at + l * 16, the column's four scales at + 1024 + nn * 8
  const uint32_t ring_q = uint32_t(reinterpret_cast<uintptr_t>(smem + kEngRingOff)) + cw * kEngRec + l * 16u;
  const uint32_t ring_s = uint32_t(reinterpret_cast<uintptr_t>(smem + kEngRingOff)) + cw * kEngRec + 1024u + nn * 8u;
  const uint32_t a_u32 = uint32_t(reinterpret_cast<uintptr_t>(abuf)) + (cw * 128u + 8u * g) * 2u;  // k-step cw, the lane's k-slot
  const LdsF32 psum = reinterpret_cast<LdsF32>(smem + kEngPsOff);
  uint32_t t0, t1;
  eng_tiles(o, t0, t1);
  const uint32_t ntl = t1 - t0;
  const uint32_t ks = o.ks;
  const uint32_t nkr = (ks + uint32_t(kEngC) - 1) / uint32_t(kEngC);
  // k-rows of a tile in which this consumer has a record (the last row of a tile may be partial)
  const uint32_t mykr = (ks > cw) ? (ks - cw + uint32_t(kEngC) - 1) / uint32_t(kEngC) : 0u;
  using Corr = CorrRaw<4, SK_BF16, false>;
#ifdef NS_ENG_TRACE
  unsigned long long tw = 0, tc = 0;
  unsigned long long _tl = __builtin_amdgcn_s_memtime();
#define ENG_T(acc_) { const unsigned long long _t1 = __builtin_amdgcn_s_memtime(); acc_ += _t1 - _tl; _tl = _t1; }
#else
#define ENG_T(acc_)
#endif

  // wait until the loader has landed rows [0, upto)
  auto wait_rows = [&](uint32_t upto) {
    if (upto <= st.fcache) return;
    for (uint32_t spins = 0; !st.dead; spins++) {
      st.fcache = __builtin_amdgcn_readfirstlane(ENG_LDS_LOAD(&ctrl->filled));
      if (upto <= st.fcache) break;
      if (spins > kEngSpinLimit) {
        eng_fail(p, 3, op);
        st.dead = true;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
  };
  auto fetch = [&](EngRec& r, uint32_t row) {  // this consumer's record of ring row `row`
    const uint32_t rw = __builtin_amdgcn_readfirstlane(row);
    const uint32_t slot = rw - (__umulhi(rw, uint32_t((0x100000000ull + kEngNRow - 1) / kEngNRow)) * kEngNRow);  // rw % kEngNRow
    const uint32_t ro = slot * kEngRow;
    typedef __attribute__((address_space(3))) const uint32_t* L32;
    r.q = *reinterpret_cast<const __attribute__((address_space(3))) uint4v*>(ring_q + ro);
    r.s0 = reinterpret_cast<L32>(ring_s + ro)[0];
    r.s1 = reinterpret_cast<L32>(ring_s + ro)[1];
  };
  auto fetch_a = [&](half8_t (&a)[4], uint32_t kr) {  // activation fragments of k-step kr * 8 + cw
#pragma unroll
    for (int jj = 0; jj < 4; jj++)
      a[jj] = *reinterpret_cast<const __attribute__((address_space(3))) half8_t*>(a_u32 + kr * (kEngC * 256u) + uint32_t(jj) * 64u);
  };
  auto compute = [&](const EngRec& r, const half8_t (&a)[4], floatx4& acc) {
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
      dd[jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[jj], bq, floatx4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    }
#pragma unroll
    for (int jj = 0; jj < 4; jj++) acc += dd[jj] * sc[jj];
  };

  for (uint32_t ti = 0; ti < ntl; ti++) {
    floatx4 acc[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) acc[q] = floatx4{0.f, 0.f, 0.f, 0.f};
    const uint32_t trow = rbase + ti * nkr * NQ;  // first ring row of the tile
    // Two records per round, read from LDS before either is computed, so that two dependency chains (LDS -> dequantise
    // -> MFMA -> scale) interleave: a fused operator's two records of one k-step, or the records of two consecutive
    // k-rows.  ("Two k-rows, then a single one" is avoided: hipcc 7.2 builds that sequence wrongly — see
    // experiments/ns_engine_v2_symmetric.hip — so an odd count runs as single rounds.)
    uint32_t kr = 0;
    if constexpr (NQ == 2) {
      for (; kr < mykr; kr++) {
        wait_rows(trow + kr * 2 + 2);
        ENG_T(tw)
        EngRec r0, r1;
        half8_t a[4];
        fetch(r0, trow + kr * 2);
        fetch(r1, trow + kr * 2 + 1);
        fetch_a(a, kr);
        compute(r0, a, acc[0]);
        compute(r1, a, acc[1]);
        if (l == 0) ENG_LDS_STORE(&ctrl->freed[cw], trow + kr * 2 + 2);
        ENG_T(tc)
      }
    } else {
      if ((mykr & 1u) == 0u) {
        for (; kr < mykr; kr += 2) {
          wait_rows(trow + kr + 2);
          ENG_T(tw)
          EngRec r0, r1;
          half8_t a0[4], a1[4];
          fetch(r0, trow + kr);
          fetch(r1, trow + kr + 1);
          fetch_a(a0, kr);
          fetch_a(a1, kr + 1);
          compute(r0, a0, acc[0]);  // k ascending: the wave's sum keeps gemv_kernel's order
          compute(r1, a1, acc[0]);
          if (l == 0) ENG_LDS_STORE(&ctrl->freed[cw], trow + kr + 2);
          ENG_T(tc)
        }
      } else {
        for (; kr < mykr; kr++) {
          wait_rows(trow + kr + 1);
          ENG_T(tw)
          EngRec r0;
          half8_t a0[4];
          fetch(r0, trow + kr);
          fetch_a(a0, kr);
          compute(r0, a0, acc[0]);
          if (l == 0) ENG_LDS_STORE(&ctrl->freed[cw], trow + kr + 1);
          ENG_T(tc)
        }
      }
    }
    // every row of the tile is behind this consumer (also a last, partial row in which it has no record)
    if (l == 0) ENG_LDS_STORE(&ctrl->freed[cw], trow + nkr * NQ);

#ifdef NS_ENG_NOTILEEND  // diagnostics (results void)
    if (acc[0][0] == 12345.678f) atomicOr(p.status, 0x40u);
    continue;
#endif
    // ---- the tile is complete for this consumer: park row 0 of its sums; the LAST one to arrive finishes the tile ----
    const uint32_t T = st.tile_seq++;  // tiles are numbered per workgroup, in the order every consumer meets them
    const uint32_t psl = T & uint32_t(kEngPs - 1);
    const LdsF32 ps = psum + psl * (2 * kEngC * 16);
    // (the ring couples the consumers: none can be more than kEngNRow rows = at most kEngNRow / 2 < kEngPs tiles ahead of
    // the slowest — the host refuses operators with fewer than two rows per tile — so a slot is never reused early)
    uint32_t df = 0;
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      if (g == 0) ps[(q * kEngC + int(cw)) * 16 + int(nn)] = acc[q][0];
      // Every lane's four result rows are the same number (all sixteen A rows of a one-row launch hold the same
      // activations).  Checking that keeps ALL FOUR registers of every MFMA result live — when only element 0 is used,
      // hipcc 7.2 hands the other three to later instructions while the MFMA that writes them is still in flight (seen
      // in v0: scale words overwritten, garbage sums) — and it is a cheap self-test of the staging on top.
      const uint32_t b0 = __builtin_bit_cast(uint32_t, acc[q][0]);
      df |= (b0 ^ __builtin_bit_cast(uint32_t, acc[q][1])) | (b0 ^ __builtin_bit_cast(uint32_t, acc[q][2])) |
            (b0 ^ __builtin_bit_cast(uint32_t, acc[q][3]));
    }
    if (df) atomicOr(p.status, 0x80u);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    uint32_t old = 0;
    if (l == 0) old = __hip_atomic_fetch_add(&ctrl->arrive[psl], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    old = __builtin_amdgcn_readfirstlane(old);
    if (old != uint32_t(kEngC - 1)) continue;
    if (l == 0) ENG_LDS_STORE(&ctrl->arrive[psl], 0u);
    asm volatile("" ::: "memory");
    float sum[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      sum[q] = 0.f;
#pragma unroll
      for (int x = 0; x < kEngC; x++) sum[q] += ps[(q * kEngC + x) * 16 + int(nn)];
    }
    float v = sum[0];
    if constexpr (NQ == 2) {
      // tmp1 = act(A*W1) ; out = (A*W3) * tmp1   (ip_fusion_ffn.cpp:364-406)
      const float t1v = (o.epi == NS_EPI_SILU) ? epi_silu(v) : epi_gelu(v);
      v = sum[1] * t1v;
    } else {
      if (o.epi == NS_EPI_GELU) v = epi_gelu(v);
      else if (o.epi == NS_EPI_SILU) v = epi_silu(v);
    }
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
    if (ti + 1 == ntl) ENG_STAMP(op, 7);
  }
#ifdef NS_ENG_TRACE
  if (cw == 0 && l == 0 && op < 64) {
    unsigned long long* d = reinterpret_cast<unsigned long long*>(p.debug) + (size_t(blockIdx.x) * 64 + op) * 8;
    d[4] = tw, d[5] = tc;
  }
#endif
}

__device__ __forceinline__ void eng_consumer(const EngParams& p, LdsB smem, uint32_t cw, uint32_t epoch) {
  EngState st;
  st.fcache = 0, st.tile_seq = 0, st.hins = 0, st.dead = false;
  uint32_t rbase = 0, gseq = 0;
  EngOp onext = eng_op(p, 0);
  for (uint32_t op = 0; op < p.nops; op++) {
    const EngOp o = onext;
    onext = eng_op(p, op + 1 < p.nops ? op + 1 : op);  // the next descriptor's scalar loads fly under this operator
#ifdef NS_ENG_NOHANDIN  // diagnostics (results void)
    if (false) {
#else
    if (o.in != ENG_IN_SAME) {
#endif
      gseq++;
      if (cw == 0) ENG_STAMP(op, 2);
      if (cw == 0 && (threadIdx.x & 63) == 0) ENG_LDS_STORE(&reinterpret_cast<LdsCtrl>(smem)->gathering, 1u);
      eng_hand_in(p, o, op, smem, smem + kEngAOff + (gseq & 1) * kEngABytes, cw, epoch, st);
      if (cw == 0 && (threadIdx.x & 63) == 0) ENG_LDS_STORE(&reinterpret_cast<LdsCtrl>(smem)->gathering, 0u);
      if (cw == 0) ENG_STAMP(op, 3);
    }
    const LdsB abuf = smem + kEngAOff + (gseq & 1) * kEngABytes;
    uint32_t t0, t1;
    eng_tiles(o, t0, t1);
    if (o.nq == 2) eng_consume_op<2>(p, smem, o, op, cw, rbase, abuf, epoch, st);
    else eng_consume_op<1>(p, smem, o, op, cw, rbase, abuf, epoch, st);
    if (cw == 0) ENG_STAMP(op, 6);
    rbase += (t1 - t0) * ((o.ks + uint32_t(kEngC) - 1) / uint32_t(kEngC)) * o.nq;  // rows
  }
}

__global__ __launch_bounds__(kEngWaves * 64) void engine_kernel(const EngParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_generic[];
  const LdsB smem = (LdsB)(smem_generic);
  const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (threadIdx.x < sizeof(EngCtrl) / 4) reinterpret_cast<LdsU32>(smem)[threadIdx.x] = 0;
  const uint32_t epoch =
      *reinterpret_cast<const __attribute__((address_space(4))) uint32_t*>(reinterpret_cast<uintptr_t>(p.epoch)) & 0x3fffffu;
  __syncthreads();
  if (w == 0) eng_loader(p, smem);
  else eng_consumer(p, smem, w - 1, epoch);
  // the token is over when workgroup 0's consumers are: every workgroup has long read the epoch by then (its outputs
  // were needed on the way), so the next launch's tags can be armed
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
  std::vector<uint32_t> ntiles;
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
      if (x->k % 128 != 0 || uint32_t(x->k) > kEngMaxK || x->ksteps < 9) return bad("K outside the engine's envelope");
    }
    EngOp o;
    memset(&o, 0, sizeof(o));
    o.w0 = reinterpret_cast<const uint8_t*>(s.w0->codes);
    o.w1 = s.w1 ? reinterpret_cast<const uint8_t*>(s.w1->codes) : o.w0;
    o.c = s.c;
    o.ks = uint32_t(w->ksteps), o.wbytes = uint32_t(w->alloc_bytes), o.nq = s.w1 ? 2u : 1u, o.n = uint32_t(w->n);
    e->ntiles.push_back(uint32_t(w->ntiles));
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
  for (size_t i = 0; i < e->ops.size() && e->grid > 0; i++)
    e->ops[i].tq = e->ntiles[i] / uint32_t(e->grid), e->ops[i].tr = e->ntiles[i] % uint32_t(e->grid);
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
  hipLaunchKernelGGL(ns::engine_kernel, dim3(e->grid), dim3(ns::kEngWaves * 64), ns::kEngLds, (hipStream_t)stream, e->params);
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
