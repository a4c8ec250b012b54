// ns_chain.hip — chain_kernel: a whole batch-1 decode GEMV chain (every projection of every layer + lm_head) as ONE
// persistent launch.
//
// Why: a decode launch costs ~3.4 us before and after its stream (kernel boundary 1.2, cold caches + argument fetch
// 0.7, first-byte latency 1.0, reduction / drain 0.5; profiles/r02h_wave_trace*), and a 7B layer's four launches
// stream only 19 us worth of bytes.  Here the boundary and the cold start disappear and the first-byte latency of
// operator i+1 hides behind the end of operator i: every wave keeps its private LDS ring (ns_gemv.hip) and requests
// the first records of its share of the NEXT operator before it joins the hand-off of the current one — weights do
// not depend on activations.
//
// Structure: one workgroup per CU (16 waves), operators from a table in device memory.  Operator = one GEMV
// y = dequant(W) x (plain), the fused gate/up pair with the SiLU-mul epilogue (dual) or several matrices side by side
// along N (QKV).  Workgroup c owns the 16-column tiles c, c + G, c + 2G, ... of every operator; its waves deal the
// k-steps of a tile round-robin (as gemv_kernel does, same summation order: results are bit-identical to the
// launch-per-operator path).  Hand-off between operators (all-to-all: every workgroup needs the whole vector):
//   producer: fp16 outputs with write-through (sc1) stores, drained (s_waitcnt vmcnt(0)), workgroup barrier, ONE
//             device-scope arrive on the operator's counter;
//   consumer: one wave polls the counter (relaxed device-scope loads, s_sleep), then the vector is staged into LDS by
//             LDS-DMA with sc1 requests (L2-served, never this CU's L1).
// No fences: 8 non-coherent L2s make an agent-scope release a whole-L2 write-back (MI355X_MICROARCH.md).  Counters are
// zeroed by a memset node in front of the launch; every spin is bounded (a protocol bug must not hang the GPU: the
// error word is raised and the result is wrong, which the parity tests report).
//
// Arithmetic per record is gemv_kernel's: NJ x v_mfma_f32_16x16x32_f16 on the raw codes, group scale applied to the
// fp32 result, w = (code - zp) * scale, fp32 accumulation (reference: bestla/bestla/kernel_ref.h:2489-2531, :1027-1127).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>

#include "../../include/ns_bestla.h"
#include "ns_common.h"
#include "ns_dev.h"

namespace ns {

constexpr int kChPF = 4;         // records a wave keeps in flight (ns_gemv.hip: the CU's request queue holds ~50 KiB)
constexpr int kChWaves = 16;     // waves per workgroup = per CU
constexpr int kChMaxTiles = 8;   // tiles of one operator a workgroup may own (lm_head 32000 / 16 / 256 = 7.8)
constexpr size_t kChMaxLds = 160 * 1024;

// one operator as the kernel reads it (scalar loads from device memory; constant for the life of the chain)
struct ChainOp {
  const uint8_t* wbase[3];   // per matrix ONE allocation: records at 0, scales at s_off, zero points at z_off
  uint32_t s_off[3], z_off[3];
  uint32_t tile_begin[3];    // first global tile of each matrix (absent: 2^32 - 1)
  uint32_t tiles;            // global tiles of the operator (dual: of ONE matrix)
  uint32_t ks;               // k-steps per tile
  uint32_t qstride, sstride, zstride;
  uint32_t srows, srow_mul, srow_shift;
  uint32_t mode;             // 0 plain, 1 dual (gate/up), 2 several matrices along N
  uint32_t epilogue;         // dual: 5 SiLU, else GELU; plain: 0 only
  uint32_t k;                // input length (elements)
  uint32_t n[3];             // columns of each matrix
  const _Float16* in16;      // input vector, fp16 [k]
  _Float16* out16[3];        // fp16 output of each matrix (null: not written)
  float* out32[3];           // fp32 output of each matrix (null: not written)
  uint32_t pad_[3];
};
static_assert(sizeof(ChainOp) % 16 == 0, "ChainOp rows are fetched with wide scalar loads");

struct ChainParams {
  const ChainOp* ops;
  uint32_t nops;
  uint32_t a_bytes;     // bytes of ONE activation buffer in LDS (two are kept)
  uint32_t ring_off;    // byte offset of the rings
  uint32_t part_off;    // byte offset of the partial-sum scratch
  uint32_t* counters;   // [nops][16]: 8 group counters, top counter, release word; zero before the launch
  uint32_t* error;      // raised when a bounded spin gives up
  uint32_t spin_limit;
  uint32_t grid;
  uint32_t poll_first, poll_gap;  // sleeps of 512 cycles before the first poll / between polls
  unsigned long long* trace;  // diagnostics (may be null): [nops][8] wall-clock stamps (100 MHz) of workgroup 0, wave 0
};

struct ns_chain_impl {
  std::vector<ChainOp> host_ops;
  ChainOp* d_ops = nullptr;
  uint32_t* d_sync = nullptr;  // [nops] counters + 1 error word
  unsigned long long* d_trace = nullptr;
  ChainParams p{};
  int kind = 0, sps = 0, sk = 0;
  bool asym = false;
  size_t lds = 0;
  int grid = 0;
};

// workgroup barrier written by hand: for __syncthreads() hipcc first waits for every LDS-DMA request of the wave (it
// cannot know that the rings are wave-private), i.e. for the look-ahead records to land
__device__ __forceinline__ void ch_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <int N>
__device__ __forceinline__ void ch_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

typedef __attribute__((address_space(3))) unsigned char* ChLds;
typedef const __attribute__((address_space(4))) ChainOp* ChOpPtr;  // constant address space: scalar loads

template <int KIND, int SPS, int SK, bool ASYM>
__global__ __launch_bounds__(kChWaves * 64) void chain_kernel(const ChainParams p) {
  constexpr int NJ = kind_is_8bit(KIND) ? 2 : 4;
  constexpr int KSTEP = NJ * 32;
  constexpr int PF = kChPF;
  constexpr int NW = kChWaves;
  constexpr int SBYTES = SPS * (SK == SK_F32 ? 4 : 2);
  constexpr uint32_t SLOT = 1024u + 16u * SBYTES + (ASYM ? 16u * SPS : 0u);
  constexpr int OPS = ASYM ? 3 : 2;  // requests per record
  static_assert(OPS * PF <= 63, "vmcnt is a 6-bit counter");
  using Corr = CorrRaw<SPS, SK, ASYM>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const uint32_t w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = tid & 63;
  const int nn = l & 15, g = l >> 4;
  const uint32_t cu = blockIdx.x, G = p.grid;
  const ChLds lds0 = (ChLds)(smem);
  const ChLds ring = lds0 + p.ring_off + w * uint32_t(PF * SLOT);
  const uint32_t ring_addr = uint32_t(reinterpret_cast<uintptr_t>(ring));
  floatx4* part = reinterpret_cast<floatx4*>(smem + p.part_off);  // [unit][wave][16 lanes] floatx4 (rows 0..3 of a column)
  const uint32_t voff_q = l * 16;
  const I4Consts i4c = {0x000f000fu, 0x00f000f0u, 0x64006400u};
  const ChOpPtr optab = reinterpret_cast<ChOpPtr>(reinterpret_cast<uint64_t>(p.ops));

  auto wait_records = [&](uint32_t younger) {  // at most `younger` records requested after the wanted one stay in flight
    if (younger == uint32_t(PF - 1)) {
      ch_wait_vmcnt<OPS * (PF - 1)>();
      return;
    }
    [&]<int... K>(std::integer_sequence<int, K...>) {
      (void)((younger == uint32_t(K) ? (ch_wait_vmcnt<OPS * K>(), true) : false) || ...);
    }(std::make_integer_sequence<int, PF + 1>{});
  };

  // ---- per-operator state of a wave: which tiles / k-steps are its own, where their records are ----
  struct OpState {
    Rsrc rw[2];            // descriptors of the matrices streamed in lockstep (dual: two)
    uint32_t so[2], zo[2];
    uint32_t ks, qstride, sstride, zstride, srows, srow_mul, srow_shift;
    uint32_t ntl;          // tiles of this workgroup in the operator
    uint32_t nst;          // k-steps of this wave per tile
    uint32_t nq;           // 1 or 2 matrices per k-step
    uint32_t items;        // records of this wave in the operator = ntl * nst * nq
    uint32_t mode, tiles;
  };
  // tile tl (local) of operator state S: global tile, its matrix, the record / scale-row bases
  auto op_load = [&](uint32_t i, OpState& S) {
    const ChOpPtr o = optab + i;
    S.ks = o->ks, S.qstride = o->qstride, S.sstride = o->sstride, S.zstride = o->zstride;
    S.srows = o->srows, S.srow_mul = o->srow_mul, S.srow_shift = o->srow_shift;
    S.mode = o->mode, S.tiles = o->tiles;
    S.nq = S.mode == 1 ? 2u : 1u;
    S.ntl = cu < S.tiles ? (S.tiles - cu + G - 1) / G : 0u;
    S.nst = w < S.ks ? (S.ks - w + NW - 1) / NW : 0u;
    S.items = S.ntl * S.nst * S.nq;
  };
  // descriptors for global tile T of operator i (matrix lookup for QKV; both matrices for dual)
  auto tile_bind = [&](uint32_t i, const OpState& S, uint32_t T, Rsrc (&rw)[2], uint32_t (&so)[2], uint32_t (&zo)[2], uint32_t& tq,
                       uint32_t& tc) {
    const ChOpPtr o = optab + i;
    uint32_t sg = 0;
    if (S.mode == 2) sg = uint32_t(T >= o->tile_begin[1]) + uint32_t(T >= o->tile_begin[2]);
    const uint32_t tl = T - (S.mode == 2 ? o->tile_begin[sg] : 0u);
    rw[0] = make_rsrc(o->wbase[sg], 0x80000000u);
    so[0] = o->s_off[sg], zo[0] = o->z_off[sg];
    if (S.mode == 1) {
      rw[1] = make_rsrc(o->wbase[1], 0x80000000u);
      so[1] = o->s_off[1], zo[1] = o->z_off[1];
    } else {
      rw[1] = rw[0], so[1] = so[0], zo[1] = zo[0];
    }
    tq = tl * S.ks * S.qstride;
    tc = tl * S.srows;
  };

  auto issue_record = [&](uint32_t slot, const OpState& S, const Rsrc& rwq, uint32_t soq, uint32_t zoq, uint32_t tq, uint32_t tc,
                          uint32_t s) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t crow = tc + ((s * S.srow_mul) >> S.srow_shift);
    const ChLds dst = ring + slot * SLOT;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rwq, reinterpret_cast<__attribute__((address_space(3))) void*>(dst), 16, voff_q,
                                             tq + s * S.qstride, 0, 2);
    if (l < SBYTES)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rwq, reinterpret_cast<__attribute__((address_space(3))) void*>(dst + 1024), 16,
                                               voff_q, soq + crow * S.sstride, 0, 2);
    if constexpr (ASYM) {
      if (l < SPS)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rwq, reinterpret_cast<__attribute__((address_space(3))) void*>(dst + 1024 + 16 * SBYTES),
                                                 16, voff_q, zoq + crow * S.zstride, 0, 2);
    }
#endif
  };

  // ---- the two cursors of a wave: ISSUE runs PF items ahead of CONSUME, both inside one operator ----
  struct Walk {
    uint32_t tl, si, q, t;  // local tile, k-step ordinal, matrix, item ordinal
    Rsrc rw[2];
    uint32_t so[2], zo[2], tq, tc;
  };
  auto walk_start = [&](uint32_t i, const OpState& S, Walk& k) {
    k.tl = 0, k.si = 0, k.q = 0, k.t = 0;
    if (S.items) tile_bind(i, S, cu, k.rw, k.so, k.zo, k.tq, k.tc);
  };
  auto walk_next = [&](uint32_t i, const OpState& S, Walk& k) {  // advance one item
    k.t++;
    if (++k.q < S.nq) return;
    k.q = 0;
    if (++k.si < S.nst) return;
    k.si = 0;
    k.tl++;
    if (k.tl < S.ntl) tile_bind(i, S, cu + k.tl * G, k.rw, k.so, k.zo, k.tq, k.tc);
  };
  auto issue_at = [&](uint32_t slot, const OpState& S, const Walk& k) {
    const uint32_t s = w + k.si * NW;
    if (k.q == 0)
      issue_record(slot, S, k.rw[0], k.so[0], k.zo[0], k.tq, k.tc, s);
    else
      issue_record(slot, S, k.rw[1], k.so[1], k.zo[1], k.tq, k.tc, s);
  };

  // request the first min(PF, items) records of operator i into ring slots 0.. (the ring is empty)
  auto fill = [&](uint32_t i, const OpState& S, Walk& isu) {
    walk_start(i, S, isu);
#pragma unroll
    for (int sl = 0; sl < PF; sl++) {
      if (isu.t < S.items) {
        issue_at(uint32_t(sl), S, isu);
        walk_next(i, S, isu);
      }
    }
  };

  // stage the fp16 input vector of operator i into activation buffer `buf` (LDS-DMA, sc1 requests: written by other
  // CUs in this launch); 1 KiB pieces dealt to the waves; columns >= k read as zero through the descriptor
  auto stage_input = [&](uint32_t i, uint32_t buf) -> uint32_t {
    const ChOpPtr o = optab + i;
    const uint32_t ks_ = o->ks, k_ = o->k;
    const Rsrc ra = make_rsrc(o->in16, k_ * 2u);
    const uint32_t row_bytes = ks_ * uint32_t(KSTEP) * 2u;
    const uint32_t pieces = (row_bytes + 1023u) >> 10;
    const ChLds al = lds0 + buf * p.a_bytes;
    uint32_t mine = 0;
    for (uint32_t c = w; c < pieces; c += NW) {
      const uint32_t left = row_bytes - (c << 10);
#if defined(__HIP_DEVICE_COMPILE__)
      if (uint32_t(l) * 16u < left)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, reinterpret_cast<__attribute__((address_space(3))) void*>(al + (c << 10)), 16,
                                                 voff_q, c << 10, 0, 16 /* sc1 */);
#endif
      mine++;
    }
    return mine;
  };

  // =====================================================================================================
  OpState S;
  Walk isu, con;
  op_load(0, S);
  // operator 0: its input is ready before the launch
  stage_input(0, 0);
  fill(0, S, isu);
  wait_records(min(S.items, uint32_t(PF)));  // this wave's input pieces are older than its ring requests: landed
  ch_barrier();

  auto stamp = [&](uint32_t i, int k) {
    if (p.trace && cu == p.grid / 2 && tid == 0) p.trace[size_t(i) * 8 + k] = wall_clock64();
  };
  for (uint32_t i = 0; i < p.nops; i++) {
    const uint32_t buf = i & 1u;
    stamp(i, 0);
    const _Float16* a_lds = reinterpret_cast<const _Float16*>(smem + buf * p.a_bytes);
    const uint32_t aoff = 8 * g;  // row 0 only (batch 1): rows >= 1 of the MFMA are discarded
    // ---- stream operator i ----
    walk_start(i, S, con);
    floatx4 acc = floatx4{0.f, 0.f, 0.f, 0.f};
    auto flush = [&](uint32_t tl, uint32_t q) {  // partial sums of (local tile, matrix): lanes g == 0 hold rows 0..3
      if (g == 0) part[((tl * S.nq + q) * NW + w) * 16 + nn] = acc;
      acc = floatx4{0.f, 0.f, 0.f, 0.f};
    };
    // dual: the two matrices of a k-step alternate in the item order, so two accumulators are live
    floatx4 acc1 = floatx4{0.f, 0.f, 0.f, 0.f};
    const uint32_t rounds = (S.items + PF - 1) / PF;
    for (uint32_t r = 0; r < rounds; r++) {
#pragma unroll
      for (int sl = 0; sl < PF; sl++) {
        if (con.t < S.items) {
          wait_records(min(S.items - con.t - 1, uint32_t(PF - 1)));
          // ---- consume the record in slot sl ----
          {
            const uint32_t s = w + con.si * NW;
            const _Float16* abase = a_lds + s * KSTEP + aoff;
            Corr cr;
            {
              typedef __attribute__((address_space(3))) const uint32_t* L32;
              const uint32_t ca = ring_addr + uint32_t(sl) * SLOT + 1024u + uint32_t(nn) * SBYTES;
              if constexpr (SBYTES == 2) {
                cr.s[0] = *reinterpret_cast<__attribute__((address_space(3))) const uint16_t*>(ca);
              } else {
#pragma unroll
                for (int t = 0; t < Corr::NW32; t++) cr.s[t] = reinterpret_cast<L32>(ca)[t];
              }
              if constexpr (ASYM) {
                const uint32_t za = ring_addr + uint32_t(sl) * SLOT + 1024u + 16u * SBYTES + uint32_t(nn) * SPS;
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
            const uint4v qvv = *reinterpret_cast<const __attribute__((address_space(3))) uint4v*>(ring_addr + uint32_t(sl) * SLOT + uint32_t(l) * 16u);
            const uint32_t xw[4] = {qvv.x, qvv.y, qvv.z, qvv.w};
            half8_t bq[NJ];
#pragma unroll
            for (int j = 0; j < NJ; j++) {
              if constexpr (KIND == WK_INT4) {
                const _Float16 zl = (_Float16)(-1032.f - zp[j]), zh = (_Float16)(-72.f - zp[j]);
                bq[j] = cvt_i4x8(xw[j], i4c, half2_t{zl, zl}, half2_t{zh, zh});
              } else {
                const _Float16 zo8 = (_Float16)(-1152.f - zp[j]);
                bq[j] = cvt_i8x8(xw[2 * j], xw[2 * j + 1], half2_t{zo8, zo8});
              }
            }
            floatx4 dd[NJ];
#pragma unroll
            for (int j = 0; j < NJ; j++) {
              const half8_t afrag = *reinterpret_cast<const half8_t*>(abase + 32 * j);
              dd[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(afrag, bq[j], floatx4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            }
            if (con.q == 0) {
#pragma unroll
              for (int j = 0; j < NJ; j++) acc += dd[j] * sc[j];
            } else {
#pragma unroll
              for (int j = 0; j < NJ; j++) acc1 += dd[j] * sc[j];
            }
          }
          __builtin_amdgcn_sched_barrier(0);
          // last k-step of the tile for this wave: park the partial sums
          if (con.si + 1 == S.nst && con.q + 1 == S.nq) {
            flush(con.tl, 0);
            if (S.nq == 2) {
              acc = acc1;
              flush(con.tl, 1);
              acc1 = floatx4{0.f, 0.f, 0.f, 0.f};
            }
          }
          walk_next(i, S, con);
          // ---- refill the slot with the record PF items ahead ----
          if (isu.t < S.items) {
            issue_at(uint32_t(sl), S, isu);
            walk_next(i, S, isu);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    stamp(i, 1);
    // every wave has parked its partial sums of operator i (waves without k-steps park nothing: nst == 0)
    ch_barrier();
    stamp(i, 2);

    // ---- look ahead: the first records of operator i+1 are requested NOW by the waves that do not finish tiles ----
    OpState N = S;
    Walk nisu = isu;
    const bool has_next = i + 1 < p.nops;
    if (has_next) op_load(i + 1, N);
    const uint32_t nred = S.ntl;  // wave j < nred finishes local tile j

    if (w < nred) {
      const ChOpPtr o = optab + i;
      const uint32_t T = cu + w * G;
      floatx4 sum0 = floatx4{0.f, 0.f, 0.f, 0.f}, sum1 = floatx4{0.f, 0.f, 0.f, 0.f};
      const uint32_t nwave = min(uint32_t(NW), S.ks);  // waves that had k-steps
      if (g == 0) {
        for (uint32_t ww = 0; ww < nwave; ww++) sum0 += part[((w * S.nq + 0) * NW + ww) * 16 + nn];
        if (S.nq == 2)
          for (uint32_t ww = 0; ww < nwave; ww++) sum1 += part[((w * S.nq + 1) * NW + ww) * 16 + nn];
      }
      uint32_t sg = 0;
      if (S.mode == 2) sg = uint32_t(T >= o->tile_begin[1]) + uint32_t(T >= o->tile_begin[2]);
      const uint32_t tl = T - (S.mode == 2 ? o->tile_begin[sg] : 0u);
      const int col = int(tl) * 16 + nn;
      float v = sum0[0];
      if (S.mode == 1) {
        const float t1 = (o->epilogue == 5) ? epi_silu(v) : epi_gelu(v);
        v = sum1[0] * t1;
      }
      float* o32 = o->out32[sg];
      _Float16* o16 = o->out16[sg];
      const bool live = g == 0 && col < int(o->n[sg]);
      if (live && o32) __hip_atomic_store(o32 + col, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (o16) {
        // two columns per dword, even lanes store (write-through)
        const uint32_t hb = uint32_t(__builtin_bit_cast(unsigned short, (_Float16)v));
        const uint32_t nb = uint32_t(__shfl_down(int(hb), 1));
        if (live && (nn & 1) == 0) {
          const bool pair = col + 1 < int(o->n[sg]);
          if (pair)
            __hip_atomic_store(reinterpret_cast<uint32_t*>(o16 + col), hb | (nb << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else
            __hip_atomic_store(reinterpret_cast<unsigned short*>(o16 + col), (unsigned short)hb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the outputs have reached memory (the CU's request queue is empty:
                                                        // nobody has started the look-ahead yet, so this is one round trip)
    }
    stamp(i, 3);
    if (!has_next) break;
    // ---- hand-off: every tile of this workgroup is in memory -> ONE arrive; wave 0 waits for all workgroups ----
    ch_barrier();
    stamp(i, 4);
    // look-ahead: the first records of operator i+1, requested by every wave but the one that runs the hand-off
    if (w != 0) fill(i + 1, N, nisu);
    if (w == 0) {
      // two-level arrive: 8 group counters (workgroup c -> group c % 8, i.e. its XCD as dispatched today — a speed
      // assumption only), the last arriver of a group bumps the top counter, the last one there raises the release
      // word that everybody polls: same-address atomics serialise at ~12 ns each, 32 + 8 instead of 256 in a row
      // every word on its own 128-byte line: group counters 0..7, top counter 8, release words 9..16 (one per group:
      // 32 pollers per word — 255 pollers on one word cut the chip's bandwidth by half, MI355X_MICROARCH.md)
      uint32_t* cnt = p.counters + size_t(i) * (17 * 32);
      uint32_t seen = 0;
      if (l == 0) {
        const uint32_t grp = cu & 7u;
        const uint32_t members = (G - grp + 7u) >> 3;
        const uint32_t old = __hip_atomic_fetch_add(cnt + grp * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == members) {
          const uint32_t groups = min(G, 8u);
          const uint32_t oldt = __hip_atomic_fetch_add(cnt + 8 * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (oldt + 1 == groups)
            for (uint32_t gg = 0; gg < groups; gg++)
              __hip_atomic_store(cnt + (9 + gg) * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      for (uint32_t k = 0; k < p.poll_first; k++) __builtin_amdgcn_s_sleep(8);  // nobody is released before a store drain + two atomics
      for (uint32_t spin = 0; spin < p.spin_limit; spin++) {
        if (l == 0) seen = __hip_atomic_load(cnt + (9 + (cu & 7u)) * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        seen = __builtin_amdgcn_readfirstlane(seen);
        if (seen) break;
        for (uint32_t k = 0; k < p.poll_gap; k++) __builtin_amdgcn_s_sleep(8);
      }
      if (!seen && l == 0) __hip_atomic_store(p.error, 1u + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      fill(i + 1, N, nisu);  // wave 0's look-ahead comes after its polling: its requests would sit in front of the polls
    }
    stamp(i, 5);
    ch_barrier();
    // ---- stage the input of operator i+1 (written by every workgroup) into the other activation buffer ----
    stage_input(i + 1, (i + 1) & 1u);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (also drains the look-ahead ring: it has had the whole hand-off to land)
    stamp(i, 6);
    ch_barrier();
    S = N;
    isu = nisu;
  }
}

// ============================================================================================================
// host side
// ============================================================================================================
template <int KIND, int SPS, int SK>
static hipError_t chain_launch_a(const ns_chain_impl* c, hipStream_t st) {
  const dim3 g(c->grid), b(kChWaves * 64);
#define NS_CH_LAUNCH(ASYMV)                                                                                      \
  {                                                                                                              \
    auto k = chain_kernel<KIND, SPS, SK, ASYMV>;                                                                  \
    static const hipError_t attr =                                                                               \
        hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, int(kChMaxLds)); \
    if (attr != hipSuccess && c->lds > 64 * 1024) return attr;                                                   \
    hipLaunchKernelGGL(k, g, b, c->lds, st, c->p);                                                               \
  }
  if (c->asym)
    NS_CH_LAUNCH(true)
  else
    NS_CH_LAUNCH(false)
#undef NS_CH_LAUNCH
  return hipGetLastError();
}
template <int KIND, int SPS>
static hipError_t chain_launch_s(const ns_chain_impl* c, hipStream_t st) {
  if (c->sk == SK_F32) return chain_launch_a<KIND, SPS, SK_F32>(c, st);
  if (c->sk == SK_F16) return chain_launch_a<KIND, SPS, SK_F16>(c, st);
  return chain_launch_a<KIND, SPS, SK_BF16>(c, st);
}

}  // namespace ns

using namespace ns;

extern "C" {

struct ns_chain {
  ns_chain_impl impl;
};

ns_chain* ns_hip_chain_create(const ns_chain_op* ops, int nops) {
  if (!ops || nops <= 0) {
    set_error("chain: no operators");
    return nullptr;
  }
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
    set_error("chain: no device");
    return nullptr;
  }
  const int cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 1;
  ns_chain* ch = new ns_chain();
  ns_chain_impl& c = ch->impl;
  const ns_weight* w0 = ops[0].w[0];
  if (!w0) {
    set_error("chain: null weight");
    delete ch;
    return nullptr;
  }
  c.kind = w0->kind, c.sps = w0->sps, c.asym = w0->asym;
  c.sk = w0->scale_dt == DT_F32 ? SK_F32 : (w0->scale_dt == DT_F16 ? SK_F16 : SK_BF16);
  if (c.kind != WK_INT4 && c.kind != WK_INT8) {
    set_error("chain: integer weight formats only");
    delete ch;
    return nullptr;
  }
  const uint32_t sbytes = uint32_t(c.sps) * (w0->scale_dt == DT_F32 ? 4u : 2u);
  const uint32_t slot = 1024u + 16u * sbytes + (c.asym ? 16u * uint32_t(c.sps) : 0u);
  size_t a_max = 0;
  uint32_t max_units = 1;
  c.grid = cus;
  for (int i = 0; i < nops; i++) {
    const ns_chain_op& o = ops[i];
    ChainOp d;
    memset(&d, 0, sizeof(d));
    const int nmat = o.mode == NS_CHAIN_PLAIN ? 1 : (o.mode == NS_CHAIN_DUAL ? 2 : o.nmat);
    if (nmat < 1 || nmat > 3 || !o.in16) {
      set_error("chain: bad operator");
      delete ch;
      return nullptr;
    }
    uint32_t tiles = 0;
    for (int m = 0; m < 3; m++) d.tile_begin[m] = 0xffffffffu;
    const ns_weight* wf = o.w[0];
    for (int m = 0; m < nmat; m++) {
      const ns_weight* w = o.w[m];
      if (!w || w->kind != c.kind || w->sps != c.sps || w->asym != c.asym || w->scale_dt != w0->scale_dt || !w->single_span ||
          w->alloc_bytes >= (size_t(1) << 31) || w->k != wf->k || w->blocksize != wf->blocksize || w->shuf ||
          (o.mode == NS_CHAIN_DUAL && w->n != wf->n)) {
        set_error("chain: the weights of a chain share one format; matrices of one operator share K");
        delete ch;
        return nullptr;
      }
      d.wbase[m] = reinterpret_cast<const uint8_t*>(w->codes);
      d.s_off[m] = uint32_t(reinterpret_cast<const uint8_t*>(w->scales) - d.wbase[m]);
      d.z_off[m] = w->zps ? uint32_t(reinterpret_cast<const uint8_t*>(w->zps) - d.wbase[m]) : 0u;
      d.tile_begin[m] = o.mode == NS_CHAIN_DUAL ? 0u : tiles;
      if (o.mode != NS_CHAIN_DUAL || m == 0) tiles += uint32_t(w->ntiles);
      d.n[m] = uint32_t(w->n);
      d.out16[m] = static_cast<_Float16*>(o.out16[m]);
      d.out32[m] = o.out32[m];
    }
    d.tiles = tiles;
    d.ks = uint32_t(wf->ksteps);
    d.qstride = wf->qstride, d.sstride = wf->sstride, d.zstride = wf->zstride;
    d.srows = uint32_t(wf->srows);
    int mul, shift;
    if (!srow_params(wf, &mul, &shift) || (wf->k & 7) != 0) {
      set_error("chain: weight geometry not supported");
      delete ch;
      return nullptr;
    }
    d.srow_mul = uint32_t(mul), d.srow_shift = uint32_t(shift);
    d.mode = o.mode == NS_CHAIN_PLAIN ? 0u : (o.mode == NS_CHAIN_DUAL ? 1u : 2u);
    d.epilogue = uint32_t(o.epilogue);
    d.k = uint32_t(wf->k);
    d.in16 = static_cast<const _Float16*>(o.in16);
    const uint32_t tiles_per_wg = (tiles + uint32_t(cus) - 1) / uint32_t(cus);
    if (tiles_per_wg > uint32_t(kChMaxTiles) || tiles_per_wg > uint32_t(kChWaves)) {
      set_error("chain: an operator has more than 8 column tiles per CU");
      delete ch;
      return nullptr;
    }
    max_units = std::max(max_units, tiles_per_wg * (o.mode == NS_CHAIN_DUAL ? 2u : 1u));
    a_max = std::max(a_max, size_t(d.ks) * wf->kstep_len * 2 + 16);
    c.host_ops.push_back(d);
  }
  a_max = (a_max + 1023) & ~size_t(1023);
  c.p.a_bytes = uint32_t(a_max);
  c.p.ring_off = uint32_t(2 * a_max);
  c.p.part_off = uint32_t(c.p.ring_off + size_t(kChWaves) * kChPF * slot);
  c.p.part_off = (c.p.part_off + 15u) & ~15u;
  c.lds = size_t(c.p.part_off) + size_t(max_units) * kChWaves * 16 * 16;
  if (c.lds > kChMaxLds) {
    set_error("chain: operators too long for the LDS plan (two activation buffers + rings + partial sums)");
    delete ch;
    return nullptr;
  }
  if (hipMalloc((void**)&c.d_ops, sizeof(ChainOp) * nops) != hipSuccess ||
      hipMalloc((void**)&c.d_sync, sizeof(uint32_t) * (size_t(nops) * 17 * 32 + 32)) != hipSuccess ||
      hipMemcpy(c.d_ops, c.host_ops.data(), sizeof(ChainOp) * nops, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemset(c.d_sync, 0, sizeof(uint32_t) * (size_t(nops) * 17 * 32 + 32)) != hipSuccess) {
    set_error("chain: device allocation failed");
    ns_hip_chain_free(ch);
    return nullptr;
  }
  c.p.ops = c.d_ops;
  c.p.nops = uint32_t(nops);
  c.p.counters = c.d_sync;
  c.p.error = c.d_sync + size_t(nops) * 17 * 32;
  c.p.spin_limit = 1u << 16;
  c.p.poll_first = getenv("NS_CHAIN_POLL_FIRST") ? uint32_t(atoi(getenv("NS_CHAIN_POLL_FIRST"))) : 6u;
  c.p.poll_gap = getenv("NS_CHAIN_POLL_GAP") ? uint32_t(atoi(getenv("NS_CHAIN_POLL_GAP"))) : 2u;
  c.p.grid = uint32_t(c.grid);
  return ch;
}

void ns_hip_chain_free(ns_chain* ch) {
  if (!ch) return;
  if (ch->impl.d_ops) hipFree(ch->impl.d_ops);
  if (ch->impl.d_sync) hipFree(ch->impl.d_sync);
  if (ch->impl.d_trace) hipFree(ch->impl.d_trace);
  delete ch;
}

int ns_hip_chain_run(ns_chain* ch, void* stream) {
  if (!ch) return -1;
  ns_chain_impl& c = ch->impl;
  hipStream_t st = (hipStream_t)stream;
  // counters back to zero in front of the launch (a memset node under capture, replayed first)
  if (hipMemsetAsync(c.d_sync, 0, sizeof(uint32_t) * size_t(c.p.nops) * 17 * 32, st) != hipSuccess) {
    set_error("chain: memset failed");
    return -1;
  }
  hipError_t e;
  if (c.kind == WK_INT4) {
    e = c.sps == 4 ? chain_launch_s<WK_INT4, 4>(&c, st) : (c.sps == 2 ? chain_launch_s<WK_INT4, 2>(&c, st) : chain_launch_s<WK_INT4, 1>(&c, st));
  } else {
    e = c.sps == 2 ? chain_launch_s<WK_INT8, 2>(&c, st) : chain_launch_s<WK_INT8, 1>(&c, st);
  }
  if (e != hipSuccess) {
    set_error(std::string("chain launch: ") + hipGetErrorString(e));
    return -1;
  }
  return 0;
}

// diagnostics: per-operator time stamps of one workgroup (us relative to the first), 8 per operator: 0 stream begins,
// 1 wave 0's stream ends, 2 all waves' partial sums parked, 3 tiles stored, 4 workgroup ready to arrive, 5 all arrived,
// 6 next input staged
int ns_hip_chain_trace(ns_chain* ch, int enable, double* out_us, int max_ops) {
  if (!ch) return -1;
  ns_chain_impl& c = ch->impl;
  if (enable && !c.d_trace) {
    if (hipMalloc((void**)&c.d_trace, size_t(c.p.nops) * 64) != hipSuccess) return -1;
    hipMemset(c.d_trace, 0, size_t(c.p.nops) * 64);
  }
  c.p.trace = enable ? c.d_trace : nullptr;
  if (out_us && c.d_trace) {
    std::vector<unsigned long long> h(size_t(c.p.nops) * 8);
    if (hipMemcpy(h.data(), c.d_trace, h.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    const unsigned long long t0 = h[0];
    const int n = std::min<int>(max_ops, int(c.p.nops));
    for (int i = 0; i < n; i++)
      for (int k = 0; k < 8; k++) out_us[i * 8 + k] = h[size_t(i) * 8 + k] ? double(h[size_t(i) * 8 + k] - t0) * 0.01 : -1.0;
    return n;
  }
  return 0;
}

int ns_hip_chain_error(ns_chain* ch) {
  if (!ch) return -1;
  uint32_t v = 0;
  if (hipMemcpy(&v, ch->impl.p.error, 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return int(v);
}

}  // extern "C"
