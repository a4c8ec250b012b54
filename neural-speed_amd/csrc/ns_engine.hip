// ns_engine.hip — the batch-1 decode GEMV chain as ONE persistent launch ("decode engine") of libns_hip.so.
//
// Why: a decode GEMV launch on MI355X costs ~3.4 us that stream nothing (kernel boundary, cold caches, first byte,
// reduction tail) — 129 launches per Llama-2-7B token = 0.44 ms of the 1.04 ms chain with HBM
// idle.  Here one workgroup per CU lives for the whole token; a LOADER wave streams this CU's share of every
// operator's weight records HBM -> LDS ring by DMA, in consumption order, and keeps streaming ACROSS operator
// boundaries while the operator's input vector is still being handed over — the run-ahead is what a launch boundary
// cannot have.  Recipe: /opt/skills/guides MI355X_MICROARCH.md "Persistent kernels" price list (rows prefetch-credit,
// allgather, gather-pass, engine-vs-launches) and cdna_hip_programming.md Guideline 16 form R2:
//   * wave 0            loader: records {1024 B codes | 128 B scales} by `buffer_load ... lds`, non-temporal, kEngD records
//                       in flight (counted vmcnt), `filled` count published in LDS; ring space from the consumers'
//                       progress words
//   * wave 1            gather: sweeps the 8-byte {fp16 x 2, tag} granules the producers of this operator's input
//                       published (relaxed agent-scope loads, re-read until every tag matches), stages them as the fp16
//                       activation row in LDS, sets `a_ready`
//   * waves 2 .. 2+C-1  consumers: the arithmetic of gemv_kernel (ns_gemv.hip) unchanged — per record 4 x
//                       v_mfma_f32_16x16x32_f16 on the raw codes, group scale on the fp32 result, consumer c owns
//                       k-steps c, c + C, ... of a tile; per tile the LAST consumer to arrive adds the C partial sums in
//                       consumer order (bit for bit gemv_kernel's sum with C waves per tile), applies the epilogue,
//                       stores fp32 C and PUBLISHES the outputs as granules (one sc1 store each)
// No s_barrier anywhere after the prologue: the roles synchronise through LDS words only.  Every spin is bounded and
// reports through the status word; granule tags carry an epoch kept in device memory, so a graph replay needs no
// per-launch memset.
//
// Arithmetic reference: bestla/bestla/kernel_ref.h:2489-2531 (gemv_4bit_fp32_fp32), :1027-1127 (decompress_kblock_s4_fp);
// fused gate/up: neural_speed/core/layers/ip_fusion_ffn.cpp:364-406.
//
// v0 envelope: one row, int4 symmetric weights with four bf16 group scales per 128-deep k-step (the Q4_0 headline
// format: interleaved records of 1152 B), K a multiple of 128.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/ns_bestla.h"
#include "ns_common.h"
#include "ns_dev.h"

namespace ns {

#ifndef NS_ENG_C
#define NS_ENG_C 8
#endif
#ifndef NS_ENG_D
#define NS_ENG_D 5
#endif
constexpr int kEngC = NS_ENG_C;              // consumer waves per workgroup (= partial sums per tile)
constexpr int kEngWaves = kEngC + 2;         // + loader + gather
constexpr uint32_t kEngRec = 1152;           // bytes of one record in HBM and in the ring
// The unit of the weight stream is a ROW: the C = 8 consecutive k-steps of one tile that the 8 consumers take side by
// side — 8 x 1152 B = 9216 B = exactly nine 1 KiB DMA pieces, contiguous in HBM (a tile's records are consecutive) and in
// the ring.  One row = 9 full-wave requests and ~20 scalar instructions: the loader stays far below one issue slot
// per 1 KiB (the first version issued two requests per RECORD behind ~35 scalar instructions and could not exceed 6 GB/s
// per CU, profiles/r03c_engine_trace_v0.txt).
constexpr uint32_t kEngRow = kEngC * kEngRec;        // 9216
constexpr int kEngRowPieces = int(kEngRow / 1024);   // 9
static_assert(kEngC == 8 && kEngRow % 1024 == 0, "a row must be a whole number of 1 KiB pieces");
constexpr int kEngD = NS_ENG_D;              // ROWS the loader keeps in flight (9 requests each: vmcnt is 6 bits)
constexpr int kEngPs = 8;                    // partial-sum slots (tiles a consumer may run ahead of the slowest)
constexpr uint32_t kEngMaxK = 11008;         // longest input row staged (halves)
constexpr uint32_t kEngABytes = ((kEngMaxK * 2 + 255) / 256) * 256;
constexpr uint32_t kEngCtrlBytes = 256;
constexpr uint32_t kEngPsBytes = kEngPs * 2 * kEngC * 16 * 4;
constexpr uint32_t kEngLds = 160 * 1024;
// LDS map: control block | ring | partial sums | two activation rows
constexpr uint32_t kEngRingOff = kEngCtrlBytes;
#ifdef NS_ENG_NREC
constexpr uint32_t kEngNRec = NS_ENG_NREC;
#else
constexpr uint32_t kEngNRec = (kEngLds - kEngCtrlBytes - kEngPsBytes - 2 * kEngABytes) / kEngRow;  // ring slots (rows)
#endif
constexpr uint32_t kEngPsOff = (kEngRingOff + kEngNRec * kEngRow + 255) / 256 * 256;
constexpr uint32_t kEngAOff = kEngPsOff + kEngPsBytes;
static_assert(kEngAOff + 2 * kEngABytes <= kEngLds, "LDS map");
static_assert(kEngRowPieces * kEngD <= 63, "vmcnt is a 6-bit counter");
static_assert(kEngNRec >= uint32_t(kEngD + 4), "ring too small for the in-flight window");
constexpr size_t kEngWordsBytes = 64 + size_t(304) * 64 * 16 * 4;  // epoch, status + the trace / dump area
constexpr uint32_t kEngSpinLimit = 4u << 20;  // polls before a wave gives up (~0.3 s)

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
  uint32_t k;             // input length (halves)
  uint32_t wbytes;        // bytes of a weight allocation: the buffer descriptors' bound (a tile's last row may read past its records)
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
  uint32_t filled;          // records landed (FIFO index), written by the loader
  uint32_t a_ready;         // inputs staged so far (gather sequence number)
  uint32_t gathering;       // the gather wave is sweeping (loader thinning hint)
  uint32_t pad0;
  uint32_t freed[16];       // per consumer: every record it owns below this FIFO index is consumed
  uint32_t arrive[kEngPs];  // per partial-sum slot: consumers arrived
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
// queue at every record (cdna_hip_programming.md section 5.7 item 1: asm memory operations are invisible to that pass).
__device__ __forceinline__ void lds_store_asm(uint32_t addr, uint32_t v) {
  asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ uint32_t lds_load_asm(uint32_t addr) {
  uint32_t v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory");
  return v;
}
#ifdef NS_ENG_TRACE
// [workgroup][op][8] 100 MHz stamps: 0 loader first issue, 1 loader last issue, 2 gather start, 3 gather done,
// 4 consumer 0 starts waiting for the input, 5 consumer 0 input ready, 6 consumer 0 done with its records, 7 last tile published
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

// this workgroup's contiguous tile range of an operator
__device__ __forceinline__ void eng_tiles(uint32_t ntiles, uint32_t& t0, uint32_t& t1) {
  const uint32_t g = gridDim.x, b = blockIdx.x;
  t0 = uint32_t((uint64_t(b) * ntiles) / g);
  t1 = uint32_t((uint64_t(b + 1) * ntiles) / g);
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
  uint32_t issued = 0, minfreed = 0, slot = 0;  // in rows
  for (uint32_t op = 0; op < p.nops; op++) {
    const EngOp o = eng_op(p, op);
    uint32_t t0, t1;
    eng_tiles(o.ntiles, t0, t1);
    const Rsrc r0 = make_rsrc(o.w0, o.wbytes);
    const Rsrc r1 = make_rsrc(o.nq > 1 ? o.w1 : o.w0, o.wbytes);
    const uint32_t nkr = (o.ks + uint32_t(kEngC) - 1) / uint32_t(kEngC);
    ENG_STAMP(op, 0);
    for (uint32_t t = t0; t < t1; t++) {
      uint32_t off = t * o.ks * kEngRec;
      for (uint32_t kr = 0; kr < nkr; kr++, off += kEngRow) {
        for (uint32_t q = 0; q < o.nq; q++) {
          if (issued - minfreed >= kEngNRec) {  // ring full: wait for the slowest consumer
            for (uint32_t spins = 0;; spins++) {
              uint32_t v = lds_load_asm(a_freed);  // lanes >= C re-read consumer 0's word: harmless for a minimum
#pragma unroll
              for (int o2 = 1; o2 < 16; o2 <<= 1) v = min(v, uint32_t(__shfl_xor(int(v), o2, 64)));
              minfreed = __builtin_amdgcn_readfirstlane(v);
              if (issued - minfreed < kEngNRec) break;
              if (spins > kEngSpinLimit) {
                eng_fail(p, 1, op);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                return;
              }
              if ((spins & 63) == 63) {  // nothing will be issued for a while: let everything land and say so
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (l == 0) lds_store_asm(a_filled, issued);
              }
              __builtin_amdgcn_s_sleep(2);
            }
          }
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
          slot = slot + 1 == kEngNRec ? 0 : slot + 1;
#ifdef NS_ENG_SYNC  // diagnostics: one row in flight at a time
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (l == 0) lds_store_asm(a_filled, issued);
#else
          // requests retire in order: everything older than the youngest kEngD rows has landed
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kEngRowPieces * kEngD) : "memory");
          if (l == 0 && issued > uint32_t(kEngD)) lds_store_asm(a_filled, issued - uint32_t(kEngD));
#endif
        }
      }
    }
    ENG_STAMP(op, 1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (l == 0) lds_store_asm(a_filled, issued);
}

// ---------------------------------------------------------------------------------------------------------------
// gather
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void eng_gather(const EngParams& p, LdsB smem, uint32_t epoch) {
  const LdsCtrl ctrl = reinterpret_cast<LdsCtrl>(smem);
  const uint32_t l = threadIdx.x & 63;
  uint32_t gseq = 0;
  for (uint32_t op = 0; op < p.nops; op++) {
    const EngOp o = eng_op(p, op);
    if (o.in == ENG_IN_SAME) continue;
    gseq++;
    ENG_STAMP(op, 2);
    const LdsB abuf = smem + kEngAOff + (gseq & 1) * kEngABytes;
    if (o.in == ENG_IN_EXTERNAL) {
      const uint4v* src = static_cast<const uint4v*>(p.x16);
      for (uint32_t i = l; i * 8 < o.k; i += 64) *reinterpret_cast<__attribute__((address_space(3))) uint4v*>(abuf + i * 16) = src[i];
    } else {
      if (l == 0) ENG_LDS_STORE(&ctrl->gathering, 1u);
      const gu64* G = (const gu64*)(p.arena + o.in);
      const uint32_t ng = o.k >> 1;
      const uint32_t tag = (epoch << 10) | o.in_tag;
      const LdsU32 a32 = reinterpret_cast<LdsU32>(abuf);
      for (uint32_t base = 0; base < ng; base += 1024) {
        for (uint32_t spins = 0;; spins++) {
          unsigned long long x[16];
#pragma unroll
          for (int i = 0; i < 16; i++) {
            const uint32_t gi = base + uint32_t(i) * 64 + l;
            x[i] = gi < ng ? __hip_atomic_load(G + gi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((unsigned long long)tag << 32);
          }
          bool ok = true;
#pragma unroll
          for (int i = 0; i < 16; i++) {
            const uint32_t gi = base + uint32_t(i) * 64 + l;
            const bool hit = uint32_t(x[i] >> 32) == tag;
            ok &= hit;
            if (hit && gi < ng) a32[gi] = uint32_t(x[i]);
          }
          if (__all(ok)) break;
          if (spins > kEngSpinLimit / 8) {
            eng_fail(p, 2, op);
            return;
          }
          __builtin_amdgcn_s_sleep(4);
        }
      }
      if (l == 0) ENG_LDS_STORE(&ctrl->gathering, 0u);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (l == 0) ENG_LDS_STORE(&ctrl->a_ready, gseq);
    ENG_STAMP(op, 3);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// consumers
// ---------------------------------------------------------------------------------------------------------------
// what a consumer holds of one k-step: its record of each matrix and the activation fragments (loaded one k-step ahead)
template <int NQ>
struct EngItem {
  uint4v q[NQ];
  uint32_t s0[NQ], s1[NQ];
  half8_t a[4];
};

template <int NQ>
__device__ __forceinline__ bool eng_consume_op(const EngParams& p, LdsB smem, const EngOp& o, uint32_t op, uint32_t cw,
                                               uint32_t rbase, uint32_t& fcache, uint32_t& tile_seq, LdsB abuf, uint32_t epoch) {
  const LdsCtrl ctrl = reinterpret_cast<LdsCtrl>(smem);
  const uint32_t l = threadIdx.x & 63;
  const uint32_t nn = l & 15, g = l >> 4;
  const I4Consts i4c = {0x000f000fu, 0x00f000f0u, 0x64006400u};
  // this consumer's record inside a ring row, per lane: codes at + l * 16, the column's four scales at + 1024 + nn * 8
  const uint32_t ring_q = uint32_t(reinterpret_cast<uintptr_t>(smem + kEngRingOff)) + cw * kEngRec + l * 16u;
  const uint32_t ring_s = uint32_t(reinterpret_cast<uintptr_t>(smem + kEngRingOff)) + cw * kEngRec + 1024u + nn * 8u;
  const uint32_t a_u32 = uint32_t(reinterpret_cast<uintptr_t>(abuf)) + (cw * 128u + 8u * g) * 2u;  // k-step cw, lane's k-slot
  const LdsF32 psum = reinterpret_cast<LdsF32>(smem + kEngPsOff);
  uint32_t t0, t1;
  eng_tiles(o.ntiles, t0, t1);
  const uint32_t ntl = t1 - t0;
  if (ntl == 0) return true;
  const uint32_t ks = o.ks;
  const uint32_t nkr = (ks + uint32_t(kEngC) - 1) / uint32_t(kEngC);
  // k-rows of a tile in which this consumer has a record (the last row of a tile may be partial)
  const uint32_t mykr = (ks > cw) ? (ks - cw + uint32_t(kEngC) - 1) / uint32_t(kEngC) : 0u;
  using Item = EngItem<NQ>;
  using Corr = CorrRaw<4, SK_BF16, false>;

  // wait until ring rows [.., row] have landed, then request the item's LDS reads
  auto fetch = [&](Item& it, uint32_t ti, uint32_t kr) -> bool {
    const uint32_t row0 = rbase + (ti * nkr + kr) * NQ;
    if (row0 + NQ > fcache) {
      for (uint32_t spins = 0;; spins++) {
        fcache = __builtin_amdgcn_readfirstlane(ENG_LDS_LOAD(&ctrl->filled));
        if (row0 + NQ <= fcache) break;
        if (spins > kEngSpinLimit) {
          eng_fail(p, 3, op);
          return false;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      const uint32_t row = row0 + q;
      const uint32_t slot = row - (__umulhi(row, uint32_t((0x100000000ull + kEngNRec - 1) / kEngNRec)) * kEngNRec);  // row % kEngNRec
      const uint32_t ro = slot * kEngRow;
      it.q[q] = *reinterpret_cast<const __attribute__((address_space(3))) uint4v*>(ring_q + ro);
      typedef __attribute__((address_space(3))) const uint32_t* L32;
      it.s0[q] = reinterpret_cast<L32>(ring_s + ro)[0];
      it.s1[q] = reinterpret_cast<L32>(ring_s + ro)[1];
    }
#pragma unroll
    for (int jj = 0; jj < 4; jj++)
      it.a[jj] = *reinterpret_cast<const __attribute__((address_space(3))) half8_t*>(a_u32 + kr * (kEngC * 256u) + uint32_t(jj) * 64u);
    return true;
  };
  auto compute = [&](const Item& it, floatx4 (&acc)[NQ]) {
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      Corr cr;
      cr.s[0] = it.s0[q], cr.s[1] = it.s1[q];
      float sc[4], zp[4];
      corr_decode<4, SK_BF16, false, 4>(cr, sc, zp);
      const uint32_t xw[4] = {it.q[q].x, it.q[q].y, it.q[q].z, it.q[q].w};
      floatx4 dd[4];
#pragma unroll
      for (int jj = 0; jj < 4; jj++) {
        const _Float16 zl = (_Float16)(-1032.f), zh = (_Float16)(-72.f);
        const half8_t bq = cvt_i4x8(xw[jj], i4c, half2_t{zl, zl}, half2_t{zh, zh});
        dd[jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(it.a[jj], bq, floatx4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      }
      // all four result rows are carried although a one-row launch needs row 0 only: using ONE element of an MFMA
      // result lets hipcc recycle the other three registers while the MFMA that writes them is still in flight (ROCm
      // 7.2: scale words overwritten by late MFMA writes — garbage sums)
#pragma unroll
      for (int jj = 0; jj < 4; jj++) acc[q] += dd[jj] * sc[jj];
    }
  };
  // the tile is complete for this consumer: park row 0 of its sums; the LAST consumer to arrive finishes the tile
  auto tile_end = [&](uint32_t ti, floatx4 (&acc)[NQ]) {
    const uint32_t psl = tile_seq & uint32_t(kEngPs - 1);
    const LdsF32 ps = psum + psl * (2 * kEngC * 16);
    if (g == 0) {
#pragma unroll
      for (int q = 0; q < NQ; q++) ps[(q * kEngC + int(cw)) * 16 + int(nn)] = acc[q][0];
    }
#pragma unroll
    for (int q = 0; q < NQ; q++) acc[q] = floatx4{0.f, 0.f, 0.f, 0.f};
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    uint32_t old = 0;
    if (l == 0) old = __hip_atomic_fetch_add(&ctrl->arrive[psl], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    old = __builtin_amdgcn_readfirstlane(old);
    tile_seq++;
    if (old != uint32_t(kEngC - 1)) return;
    if (l == 0) ENG_LDS_STORE(&ctrl->arrive[psl], 0u);
    asm volatile("" ::: "memory");
    float sum[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      sum[q] = 0.f;
#pragma unroll
      for (int w = 0; w < kEngC; w++) sum[q] += ps[(q * kEngC + w) * 16 + int(nn)];
    }
    const uint32_t col = (t0 + ti) * 16 + nn;
    const bool okc = col < o.n && g == 0;
    float v = sum[0];
    if constexpr (NQ == 2) {
      // tmp1 = act(A*W1) ; out = (A*W3) * tmp1   (ip_fusion_ffn.cpp:364-406)
      const float t1v = (o.epi == NS_EPI_SILU) ? epi_silu(v) : epi_gelu(v);
      v = sum[1] * t1v;
    } else {
      if (o.epi == NS_EPI_GELU) v = epi_gelu(v);
      else if (o.epi == NS_EPI_SILU) v = epi_silu(v);
    }
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
  };

  floatx4 acc[NQ];
#pragma unroll
  for (int q = 0; q < NQ; q++) acc[q] = floatx4{0.f, 0.f, 0.f, 0.f};
  if (mykr == 0) {  // (cannot happen with ks >= C; kept for the arrival count)
    for (uint32_t ti = 0; ti < ntl; ti++) tile_end(ti, acc);
    return true;
  }
  // this consumer's items in order: (tile 0, k-row 0), (0, 1), ... (0, mykr - 1), (1, 0), ...; item n + 1 is fetched
  // before item n is computed (two register sets, the loop is unrolled by two)
  const uint32_t nitems = ntl * mykr;
  Item ia, ib;
  uint32_t ti = 0, kr = 0;  // the item about to be COMPUTED
  if (!fetch(ia, 0, 0)) return false;
  for (uint32_t n = 0; n < nitems; n += 2) {
    {  // ---- compute item n from ia; fetch item n + 1 into ib ----
      uint32_t ti2 = ti, kr2 = kr + 1;
      if (kr2 == mykr) ti2++, kr2 = 0;
      if (n + 1 < nitems && !fetch(ib, ti2, kr2)) return false;
      compute(ia, acc);
      if (l == 0) ENG_LDS_STORE(&ctrl->freed[cw], rbase + (ti * nkr + kr + 1) * NQ);
      if (kr + 1 == mykr) tile_end(ti, acc);
      ti = ti2, kr = kr2;
    }
    if (n + 1 >= nitems) break;
    {  // ---- compute item n + 1 from ib; fetch item n + 2 into ia ----
      uint32_t ti2 = ti, kr2 = kr + 1;
      if (kr2 == mykr) ti2++, kr2 = 0;
      if (n + 2 < nitems && !fetch(ia, ti2, kr2)) return false;
      compute(ib, acc);
      if (l == 0) ENG_LDS_STORE(&ctrl->freed[cw], rbase + (ti * nkr + kr + 1) * NQ);
      if (kr + 1 == mykr) tile_end(ti, acc);
      ti = ti2, kr = kr2;
    }
  }
  // every row of this operator is behind this consumer now (a consumer without a record in a tile's last, partial row
  // would otherwise hold that row's slot until its next operator)
  if (l == 0) ENG_LDS_STORE(&ctrl->freed[cw], rbase + ntl * nkr * NQ);
  return true;
}

__device__ __forceinline__ void eng_consumer(const EngParams& p, LdsB smem, uint32_t cw, uint32_t epoch) {
  const LdsCtrl ctrl = reinterpret_cast<LdsCtrl>(smem);
  uint32_t jbase = 0, fcache = 0, tile_seq = 0, gseq = 0;
  for (uint32_t op = 0; op < p.nops; op++) {
    const EngOp o = eng_op(p, op);
    if (cw == 0) ENG_STAMP(op, 4);
    if (o.in != ENG_IN_SAME) {
      gseq++;
      for (uint32_t spins = 0;; spins++) {
        if (__builtin_amdgcn_readfirstlane(ENG_LDS_LOAD(&ctrl->a_ready)) >= gseq) break;
        if (spins > kEngSpinLimit) {
          eng_fail(p, 4, op);
          return;
        }
        __builtin_amdgcn_s_sleep(2);
      }
      asm volatile("" ::: "memory");
    }
    if (cw == 0) ENG_STAMP(op, 5);
    const LdsB abuf = smem + kEngAOff + (gseq & 1) * kEngABytes;
    uint32_t t0, t1;
    eng_tiles(o.ntiles, t0, t1);
    const bool ok = o.nq == 2 ? eng_consume_op<2>(p, smem, o, op, cw, jbase, fcache, tile_seq, abuf, epoch)
                              : eng_consume_op<1>(p, smem, o, op, cw, jbase, fcache, tile_seq, abuf, epoch);
    if (!ok) return;
    if (cw == 0) ENG_STAMP(op, 6);
    jbase += (t1 - t0) * ((o.ks + uint32_t(kEngC) - 1) / uint32_t(kEngC)) * o.nq;  // rows
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
  if (w == 0) {
    eng_loader(p, smem);
  } else if (w == 1) {
    eng_gather(p, smem, epoch);
  } else {
    eng_consumer(p, smem, w - 2, epoch);
  }
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
      if (x->k % 128 != 0 || uint32_t(x->k) > kEngMaxK || x->ksteps < kEngC) return bad("K outside the engine's envelope");
    }
    EngOp o;
    memset(&o, 0, sizeof(o));
    o.w0 = reinterpret_cast<const uint8_t*>(s.w0->codes);
    o.w1 = s.w1 ? reinterpret_cast<const uint8_t*>(s.w1->codes) : o.w0;
    o.c = s.c;
    o.ks = uint32_t(w->ksteps), o.ntiles = uint32_t(w->ntiles), o.nq = s.w1 ? 2u : 1u, o.n = uint32_t(w->n);
    o.k = uint32_t(w->k);
    o.wbytes = uint32_t(w->alloc_bytes);
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
