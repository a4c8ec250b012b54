// ns_gemv.hip — gemv_kernel: the decode (M <= 16 rows) weight-streaming kernel of libns_hip.so, second generation.
//
// Same arithmetic as smallm_kernel (ns_kernels.hip): per 1 KiB weight record NJ x v_mfma_f32_16x16x32_f16 on the raw
// codes, the group scale applied to the fp32 MFMA result — exactly w = (code - zp) * scale with fp32 accumulation
// (reference: bestla/bestla/kernel_ref.h:2489-2531 gemv_4bit_fp32_fp32, :1027-1127 decompress_kblock_s4_fp,
// :1456-1478 decompress_kblock_f4_fp), fp16 activations.  What changed is everything AROUND the streaming loop,
// because a decode launch on MI355X is latency-, not bandwidth-bound:
//
//   * lean prologue.  Every launch starts with cold instruction and scalar caches, so the time to the first weight
//     request is the number of code lines and dependent kernarg fetches in front of it.  smallm_kernel had 1.8 KB of
//     code and two kernarg round trips there (1.3-1.6 us); here the first 64 bytes of the argument block hold all the
//     first loads need and the ring is filled within the first few cache lines of code.
//   * balanced work.  MI355X shares HBM bandwidth per CU: 688 column tiles on 256 CUs leave 176 CUs with three tiles
//     and 80 with two, and the launch ends 2 us after the light CUs went idle.  The grid is therefore a hybrid:
//     whole tiles for floor(tiles / CUs) * CUs workgroups, and the remaining tiles cut into equal K-ranges over one
//     workgroup per CU ("stream-K" part, dispatched first).  A tile shared by several workgroups is finished by the one
//     that holds its first k-step; the others publish their partial sums through the weight's workspace (write-through
//     stores + flag, no fences: 8 non-coherent L2s make an agent-scope release cost a whole-L2 write-back).  The
//     stream-K workgroups are shorter than the whole-tile ones, so the hand-off is off the critical path.
//     The same split lets a narrow tensor-parallel shard (fewer tiles than CUs) use every CU.
//   * the activation vector is staged ONCE per workgroup and reused for every tile segment it streams.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <utility>

#include "ns_common.h"
#include "ns_dev.h"

namespace ns {

#ifndef NS_GV_PF
#define NS_GV_PF 4
#endif
#ifndef NS_GV_PF_WIDE
#define NS_GV_PF_WIDE 2
#endif
constexpr int kGvPF = NS_GV_PF;           // records each wave keeps in flight (<= 8-wave workgroups)
constexpr int kGvPFWide = NS_GV_PF_WIDE;  // same for 16-wave workgroups (128-VGPR budget)
constexpr int kGvMaxRows = 16;
constexpr size_t kGvMaxALds = 64 * 1024;  // staged activations (fp16) per workgroup

struct GemvParams {
  // ---- first 64 bytes: everything the first weight loads need ----
  const uint8_t* wbase[3];  // per matrix ONE allocation: records at 0, scales at s_off, zero points at z_off
  const _Float16* a16;      // fp16 activations (null: convert `a` while staging)
  uint32_t ks;              // k-steps per tile
  uint32_t qstride;         // bytes per (tile, k-step) record
  uint32_t n_sk;            // stream-K workgroups = blocks [0, n_sk); block n_sk + t owns whole tile t
  uint32_t sk_u0;           // first stream-K unit (= whole tiles * ks); a unit is one k-step of one tile
  uint32_t sk_q, sk_r;      // stream-K block i covers units [sk_u0 + i * sk_q + min(i, sk_r), ... + sk_q + (i < sk_r))
  uint32_t ks_magic;        // ceil(2^32 / ks)
  uint32_t nw_log2;         // log2(waves per workgroup)
  // ---- second line ----
  uint32_t s_off[3], z_off[3];
  uint32_t sstride, zstride;
  uint32_t srows, srow_mul, srow_shift;
  uint32_t tile_begin[4];   // first global tile of each matrix laid side by side along N (QKV); [nseg] = total
  int m, k, lda;
  uint32_t upr, upr_magic;  // 16-byte fp16 units per staged row (= ks * KSTEP / 8) and ceil(2^32 / upr)
  uint32_t row_stride;      // halves per staged row in LDS
  uint32_t red_off;         // byte offset of the reduction scratch in LDS
  const float* a;
  // ---- epilogue ----
  float* c[3];
  _Float16* c16[3];
  float* c2;
  const float* d;
  float* parts;     // stream-K partial sums: [n_sk][NQ][16 rows][16 columns]
  uint32_t* flags;  // [n_sk], zero between launches
  int n[3];
  int ldc, ldd, nseg, epilogue;
  uint32_t spin_limit;
  F4Lut lut;
  F8Consts f8;
#ifdef NS_TRACE
  unsigned long long* trace;
#endif
};

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

template <int KIND, int SPS, int SK, bool ASYM, bool DUAL, bool WIDE>
__global__ __launch_bounds__(WIDE ? 1024 : 512) void gemv_kernel(const GemvParams p) {
  constexpr int NJ = kind_is_8bit(KIND) ? 2 : 4;
  constexpr int KSTEP = NJ * 32;
  constexpr int NQ = DUAL ? 2 : 1;
  constexpr int PF = WIDE ? kGvPFWide : kGvPF;
  static_assert(PF % NQ == 0, "ring slots alternate between the two matrices");
  constexpr int SBYTES = SPS * (SK == SK_F32 ? 4 : 2);
  using Corr = CorrRaw<SPS, SK, ASYM>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  _Float16* a_lds = reinterpret_cast<_Float16*>(smem);
  NS_GSTAMP(0);

  // every kernel argument the prologue needs is fetched by ONE batch of scalar loads: without this hipcc sinks each
  // load to its first use and the first weight request waits for three dependent kernarg round trips
  {
    asm volatile("" ::"s"(p.wbase[0]), "s"(p.wbase[1]), "s"(p.wbase[2]), "s"(p.a16), "s"(p.ks), "s"(p.qstride),
                 "s"(p.n_sk), "s"(p.sk_u0), "s"(p.sk_q), "s"(p.sk_r), "s"(p.ks_magic), "s"(p.nw_log2), "s"(p.s_off[0]),
                 "s"(p.s_off[1]), "s"(p.s_off[2]), "s"(p.sstride), "s"(p.srows), "s"(p.srow_mul), "s"(p.srow_shift),
                 "s"(p.tile_begin[1]), "s"(p.tile_begin[2]), "s"(p.m), "s"(p.k), "s"(p.lda), "s"(p.upr),
                 "s"(p.upr_magic), "s"(p.row_stride), "s"(p.red_off));
    if constexpr (ASYM) asm volatile("" ::"s"(p.z_off[0]), "s"(p.z_off[1]), "s"(p.z_off[2]), "s"(p.zstride));
  }
  const int tid = threadIdx.x;
  const uint32_t w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = tid & 63;
  const int nn = l & 15, g = l >> 4;
  const uint32_t NW = 1u << p.nw_log2;
  const uint32_t ks = p.ks;
  const uint32_t b = blockIdx.x;

  // ---- this workgroup's unit range [f0, f1): units are ordered [tile][k-step] like the records in memory ----
  uint32_t f0, f1;
  if (b < p.n_sk) {
    f0 = p.sk_u0 + b * p.sk_q + min(b, p.sk_r);
    f1 = f0 + p.sk_q + (b < p.sk_r ? 1u : 0u);
  } else {
    f0 = (b - p.n_sk) * ks;
    f1 = f0 + ks;
  }
  uint32_t T = __umulhi(f0, p.ks_magic);  // current tile (global numbering across the matrices of a fused launch)
  uint32_t sa = f0 - T * ks;              // segment = k-steps [sa, sb) of tile T
  uint32_t sb = min(ks, f1 - T * ks);

  // per-segment stream state: descriptors over the whole allocation of each matrix (codes, scales and zero points are
  // reached from one base), record / scale-row bases of the tile
  Rsrc rw[NQ];
  uint32_t so[NQ], zo[NQ];
  uint32_t tile_q, tile_c;
  int sg = 0;        // matrix of the fused launch the tile belongs to (QKV)
  uint32_t tl = 0;   // tile inside that matrix
  auto setup = [&]() {
    if constexpr (DUAL) {
      rw[0] = make_rsrc(p.wbase[0], 0x80000000u);
      rw[1] = make_rsrc(p.wbase[1], 0x80000000u);
      so[0] = p.s_off[0], so[1] = p.s_off[1];
      zo[0] = p.z_off[0], zo[1] = p.z_off[1];
      tl = T;
    } else {
      // masks instead of selects: hipcc turns a select chain over kernel arguments into a table in scratch memory
      const uint64_t b0 = reinterpret_cast<uint64_t>(p.wbase[0]), b1 = reinterpret_cast<uint64_t>(p.wbase[1]),
                     b2 = reinterpret_cast<uint64_t>(p.wbase[2]);
      const bool ge1 = T >= p.tile_begin[1], ge2 = T >= p.tile_begin[2];  // absent matrices begin at 2^32 - 1
      const uint64_t M1 = 0ull - uint64_t(ge1), M2 = 0ull - uint64_t(ge2);
      const uint32_t m1 = 0u - uint32_t(ge1), m2 = 0u - uint32_t(ge2);
      rw[0] = make_rsrc(reinterpret_cast<const void*>(b0 + ((b1 - b0) & M1) + ((b2 - b1) & M2)), 0x80000000u);
      so[0] = p.s_off[0] + ((p.s_off[1] - p.s_off[0]) & m1) + ((p.s_off[2] - p.s_off[1]) & m2);
      zo[0] = p.z_off[0] + ((p.z_off[1] - p.z_off[0]) & m1) + ((p.z_off[2] - p.z_off[1]) & m2);
      sg = int(ge1) + int(ge2);
      tl = T - ((p.tile_begin[1] & m1) + ((p.tile_begin[2] - p.tile_begin[1]) & m2));
    }
    tile_q = tl * ks * p.qstride;
    tile_c = tl * p.srows;
  };
  const uint32_t voff_q = l * 16, voff_s = nn * SBYTES, voff_z = nn * SPS;  // the only per-lane address parts
  const I4Consts i4c = {0x000f000fu, 0x00f000f0u, 0x64006400u};

  uint4v qv[PF];
  Corr cr[PF];
  auto issue = [&](auto slot_c, uint32_t s) {
    constexpr int slot = decltype(slot_c)::value;
    constexpr int q = slot % NQ;
    const uint32_t srow = (s * p.srow_mul) >> p.srow_shift;
    qv[slot] = __builtin_bit_cast(uint4v, __builtin_amdgcn_raw_buffer_load_b128(rw[q], voff_q, tile_q + s * p.qstride, 2));
    const uint32_t crow = tile_c + srow;
    corr_issue<SPS, SK, ASYM>(rw[q], rw[q], voff_s, voff_z, so[q] + crow * p.sstride, zo[q] + crow * p.zstride, cr[slot]);
  };

#define NS_FOR_SLOTS(BODY)                                        \
  {                                                               \
    [&]<int... I>(std::integer_sequence<int, I...>) {             \
      (([&] { constexpr int i = I; std::integral_constant<int, I> ic; (void)i; (void)ic; BODY }()), ...); \
    }(std::make_integer_sequence<int, PF>{});                     \
  }

  // items of this wave in segment [sa, sb): k-steps sa + w, sa + w + NW, ...; item t = (k-step ordinal t / NQ, matrix t % NQ)
  uint32_t first, nitems;
  auto ring_fill = [&]() {
    first = sa + w;
    const uint32_t nst = first < sb ? (sb - first + NW - 1) >> p.nw_log2 : 0u;
    nitems = nst * NQ;
    if (nitems >= uint32_t(PF)) {
      NS_FOR_SLOTS({ issue(ic, first + ((i / NQ) << p.nw_log2)); })
    } else {
      NS_FOR_SLOTS({ if (uint32_t(i) < nitems) issue(ic, first + ((i / NQ) << p.nw_log2)); })
    }
  };

  // ---- 1. activations requested first (they are small and must be in LDS before the first MFMA), then the ring ----
  const int rows = min(p.m, kGvMaxRows);
  const uint32_t total_units = uint32_t(rows) * p.upr;
  const uint32_t nthreads = blockDim.x;
  constexpr int UN = 2;  // 16-byte units a thread has in flight per staging batch
  const bool use_a16 = p.a16 != nullptr;
  const Rsrc ra = use_a16 ? make_rsrc(p.a16, uint32_t(rows) * uint32_t(p.lda) * 2u)
                          : make_rsrc(p.a, uint32_t(rows) * uint32_t(p.lda) * 4u);
  // unit u of the staging = (row r, chunk ko of 8 elements): element offset in A, or kOob (the load then returns 0)
  constexpr uint32_t kOob = 0xffffffffu;
  auto a_unit = [&](uint32_t u, uint32_t& lds_halves) -> uint32_t {
    uint32_t r = 0, ko = u;
    if (rows > 1) {
      r = __umulhi(u, p.upr_magic);
      ko = u - r * p.upr;
    }
    lds_halves = r * p.row_stride + ko * 8;
    const bool ok = u < total_units && int(ko * 8) < p.k;
    return ok ? r * uint32_t(p.lda) + ko * 8 : kOob;
  };
  auto a16_load = [&](uint32_t e) {  // byte offset 2^31 is beyond every descriptor: reads as zero, no memory access
    return __builtin_bit_cast(uint4v, __builtin_amdgcn_raw_buffer_load_b128(ra, e == kOob ? 0x80000000u : e * 2u, 0, 0));
  };
  uint4v a_first[UN];
  uint32_t a_first_dst[UN];
  if (use_a16) {
#pragma unroll
    for (int i = 0; i < UN; i++) {
      a_first[i] = a16_load(a_unit(uint32_t(tid) + uint32_t(i) * nthreads, a_first_dst[i]));
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  setup();
  ring_fill();
  __builtin_amdgcn_sched_barrier(0);
  NS_GSTAMP(1);

  // ---- 2. stage A as fp16: [rows][row_stride] halves, columns >= K are zero (k-step padding) ----
  if (use_a16) {
#pragma unroll
    for (int i = 0; i < UN; i++)
      if (uint32_t(tid) + uint32_t(i) * nthreads < total_units)
        *reinterpret_cast<uint4v*>(a_lds + a_first_dst[i]) = a_first[i];
    for (uint32_t u0 = UN * nthreads; u0 < total_units; u0 += UN * nthreads) {
      uint4v v[UN];
      uint32_t dst[UN];
#pragma unroll
      for (int i = 0; i < UN; i++) {
        v[i] = a16_load(a_unit(u0 + uint32_t(tid) + uint32_t(i) * nthreads, dst[i]));
      }
#pragma unroll
      for (int i = 0; i < UN; i++)
        if (u0 + uint32_t(tid) + uint32_t(i) * nthreads < total_units) *reinterpret_cast<uint4v*>(a_lds + dst[i]) = v[i];
    }
  } else {
    // fp32 activations: 8 floats per unit, converted on the way in; rows need not be 16-byte aligned
    const bool vec_ok = ((p.lda & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.a) & 15) == 0);
    for (uint32_t u = tid; u < total_units; u += nthreads) {
      uint32_t dst;
      const uint32_t e = a_unit(u, dst);
      float f[8];
      const int kk = e == kOob ? p.k : int(e % uint32_t(p.lda));  // column of the unit's first element
      if (e != kOob && vec_ok && kk + 8 <= p.k) {
        const uint4v v0 = __builtin_bit_cast(uint4v, __builtin_amdgcn_raw_buffer_load_b128(ra, e * 4u, 0, 0));
        const uint4v v1 = __builtin_bit_cast(uint4v, __builtin_amdgcn_raw_buffer_load_b128(ra, e * 4u, 16, 0));
        const uint32_t vw[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
        for (int t = 0; t < 8; t++) f[t] = __builtin_bit_cast(float, vw[t]);
      } else {
#pragma unroll
        for (int t = 0; t < 8; t++)
          f[t] = kk + t < p.k ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, (e + t) * 4u, 0, 0)) : 0.f;
      }
      half2_t h0 = {(_Float16)f[0], (_Float16)f[1]}, h1 = {(_Float16)f[2], (_Float16)f[3]};
      half2_t h2 = {(_Float16)f[4], (_Float16)f[5]}, h3 = {(_Float16)f[6], (_Float16)f[7]};
      *reinterpret_cast<uint4v*>(a_lds + dst) = uint4v{as_u32(h0), as_u32(h1), as_u32(h2), as_u32(h3)};
    }
  }
  __syncthreads();
  NS_GSTAMP(2);

  // A-fragment LDS offset of this lane (rows >= m are clamped: their output rows are discarded)
  const uint32_t aoff = uint32_t(min(nn, rows - 1)) * p.row_stride + 8 * g;
  floatx4 acc[NQ];

  auto compute = [&](auto slot_c, uint32_t s) {
    constexpr int slot = decltype(slot_c)::value;
    constexpr int q = slot % NQ;
    const _Float16* abase = a_lds + s * KSTEP + aoff;
    float sc[4], zp[4];
    corr_decode<SPS, SK, ASYM, NJ>(cr[slot], sc, zp);
    const uint32_t xw[4] = {qv[slot].x, qv[slot].y, qv[slot].z, qv[slot].w};
    half8_t bq[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) {
      if constexpr (KIND == WK_INT4) {
        const _Float16 zl = (_Float16)(-1032.f - zp[j]), zh = (_Float16)(-72.f - zp[j]);
        bq[j] = cvt_i4x8(xw[j], i4c, half2_t{zl, zl}, half2_t{zh, zh});
      } else if constexpr (KIND == WK_INT8) {
        const _Float16 zo8 = (_Float16)(-1152.f - zp[j]);
        bq[j] = cvt_i8x8(xw[2 * j], xw[2 * j + 1], half2_t{zo8, zo8});
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

  floatx4* red = reinterpret_cast<floatx4*>(smem + p.red_off);  // [2 parities][NW][NQ][64 lanes]
  uint32_t parity = 0;
  for (;;) {
#pragma unroll
    for (int q = 0; q < NQ; q++) acc[q] = floatx4{0.f, 0.f, 0.f, 0.f};
    // ---- 3. stream the segment: steady-state rounds consume slot i and refill it PF items ahead with no conditions
    //      inside, so the compiler emits counted vmcnt waits; the last rounds are peeled ----
    constexpr int SPR = PF / NQ;  // k-steps per round
    const uint32_t rounds = nitems / PF, rem = nitems - rounds * PF;
    if (nitems >= uint32_t(PF)) {
      uint32_t r = 0;
      for (; r + 1 < rounds; r++) {
        NS_FOR_SLOTS({
          compute(ic, first + ((r * SPR + i / NQ) << p.nw_log2));
          __builtin_amdgcn_sched_barrier(0);
          issue(ic, first + (((r + 1) * SPR + i / NQ) << p.nw_log2));
          __builtin_amdgcn_sched_barrier(0);
        })
      }
      NS_FOR_SLOTS({
        compute(ic, first + ((r * SPR + i / NQ) << p.nw_log2));
        __builtin_amdgcn_sched_barrier(0);
        if (uint32_t(i) < rem) issue(ic, first + (((r + 1) * SPR + i / NQ) << p.nw_log2));
        __builtin_amdgcn_sched_barrier(0);
      })
      r++;
      NS_FOR_SLOTS({
        if (uint32_t(i) < rem) compute(ic, first + ((r * SPR + i / NQ) << p.nw_log2));
        __builtin_amdgcn_sched_barrier(0);
      })
    } else {
      NS_FOR_SLOTS({
        if (uint32_t(i) < nitems) compute(ic, first + ((i / NQ) << p.nw_log2));
        __builtin_amdgcn_sched_barrier(0);
      })
    }
    NS_GSTAMP(4);

    // ---- 4. cross-wave reduction through LDS (two scratch parities: one barrier per segment) ----
    floatx4* rp = red + size_t(parity) * NW * NQ * 64;
#pragma unroll
    for (int q = 0; q < NQ; q++) rp[(w * NQ + q) * 64 + l] = acc[q];
    __syncthreads();
    const bool has_next = T * ks + sb < f1;
    if (w == 0) {
      floatx4 sum[NQ];
#pragma unroll
      for (int q = 0; q < NQ; q++) {
        sum[q] = floatx4{0.f, 0.f, 0.f, 0.f};
        for (uint32_t ww = 0; ww < NW; ww++) sum[q] += rp[(ww * NQ + q) * 64 + l];
      }
      if (sa != 0) {
        // the tile started in an earlier workgroup, which owns it: publish this workgroup's share.  Write-through
        // (sc1) stores, drained, then the flag — no release fence (it would write back a whole L2)
#pragma unroll
        for (int q = 0; q < NQ; q++) {
          float* dst = p.parts + ((size_t(b) * NQ + q) * 64 + l) * 4;
#pragma unroll
          for (int e = 0; e < 4; e++)
            if (4 * g + e < rows) __hip_atomic_store(dst + e, sum[q][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (l == 0) __hip_atomic_store(p.flags + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        if (sb < ks) {
          // continued by the following stream-K workgroups: add their published shares in workgroup order (bit-wise
          // reproducible), then hand the flags back at zero for the next launch
          for (uint32_t c = b + 1; c < p.n_sk; c++) {
            const uint32_t f0c = p.sk_u0 + c * p.sk_q + min(c, p.sk_r);
            if (f0c >= (T + 1) * ks) break;
            // bounded: a protocol bug must not hang the GPU (the result is then wrong and the parity tests say so)
            for (uint32_t spin = 0; spin < p.spin_limit; spin++) {
              if (__hip_atomic_load(p.flags + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
              __builtin_amdgcn_s_sleep(2);
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int q = 0; q < NQ; q++) {
              const float* src = p.parts + ((size_t(c) * NQ + q) * 64 + l) * 4;
#pragma unroll
              for (int e = 0; e < 4; e++)
                if (4 * g + e < rows) sum[q][e] += __hip_atomic_load(src + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (l == 0) __hip_atomic_store(p.flags + c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        // ---- 5. epilogue: lane (nn, g) holds rows 4g .. 4g+3 of column nn ----
        const int col = int(tl) * 16 + nn;
        const int ncols = DUAL ? p.n[0] : (sg == 0 ? p.n[0] : (sg == 1 ? p.n[1] : p.n[2]));
        if (col < ncols) {
          float* cbase = DUAL ? p.c[0] : (sg == 0 ? p.c[0] : (sg == 1 ? p.c[1] : p.c[2]));
          _Float16* c16 = DUAL ? p.c16[0] : (sg == 0 ? p.c16[0] : (sg == 1 ? p.c16[1] : p.c16[2]));
#pragma unroll
          for (int rr = 0; rr < 4; rr++) {
            const int row = 4 * g + rr;
            if (row >= p.m) continue;
            float v = sum[0][rr];
            if constexpr (DUAL) {
              // tmp1 = act(A*W1) ; tmp2 = (A*W3) * tmp1   (neural_speed/core/layers/ip_fusion_ffn.cpp:364-406)
              const float t1 = (p.epilogue == 5) ? epi_silu(v) : epi_gelu(v);
              if (p.c2) p.c2[size_t(row) * p.ldc + col] = t1;
              v = sum[1][rr] * t1;
            } else {
              const float dv = p.d ? p.d[size_t(row) * p.ldd + col] : 0.f;
              switch (p.epilogue) {
                case 1: v = v + dv; break;            // custom::epilogue::Add
                case 2: v = v * dv; break;            // custom::epilogue::Mul
                case 3: v = epi_gelu(v + dv); break;  // custom::epilogue::Add_Gelu
                case 4: v = epi_gelu(v); break;
                case 5: v = epi_silu(v); break;
                default: break;
              }
            }
            cbase[size_t(row) * p.ldc + col] = v;
            if (c16) c16[size_t(row) * p.ldc + col] = (_Float16)v;
          }
        }
      }
    }
    if (!has_next) break;
    // ---- next segment: the first k-steps of the following tile ----
    T += 1;
    sa = 0;
    sb = min(ks, f1 - T * ks);
    parity ^= 1u;
    setup();
    ring_fill();
    __builtin_amdgcn_sched_barrier(0);
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
template <int KIND, int SPS, int SK, bool ASYM>
static hipError_t launch_gemv_k(const GemvParams& p, bool dual, int grid, int nw, size_t lds, hipStream_t st) {
  const dim3 g(grid), b(nw * 64);
#define NS_GV_LAUNCH(DUALV, WIDEV)                                                                              \
  {                                                                                                             \
    auto k = gemv_kernel<KIND, SPS, SK, ASYM, DUALV, WIDEV>;                                                     \
    static const hipError_t attr =                                                                              \
        hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); \
    if (attr != hipSuccess && lds > 64 * 1024) return attr;                                                     \
    hipLaunchKernelGGL(k, g, b, lds, st, p);                                                                    \
  }
  if (dual) {
    if (nw > 8) return hipErrorInvalidValue;
    NS_GV_LAUNCH(true, false)
  } else {
    // plain launches run the 128-VGPR / 2-deep-ring instantiation at every wave count (measured +2 % over the 4-deep
    // one), the fused gate/up launch the 4-deep one
    NS_GV_LAUNCH(false, true)
  }
#undef NS_GV_LAUNCH
  return hipGetLastError();
}
template <int KIND, int SPS, int SK>
static hipError_t launch_gemv_a(const GemvParams& p, bool asym, bool dual, int grid, int nw, size_t lds, hipStream_t st) {
  if constexpr (KIND == WK_F4 || KIND == WK_F8) {
    (void)asym;
    return launch_gemv_k<KIND, SPS, SK, false>(p, dual, grid, nw, lds, st);
  } else {
    if (asym) return launch_gemv_k<KIND, SPS, SK, true>(p, dual, grid, nw, lds, st);
    return launch_gemv_k<KIND, SPS, SK, false>(p, dual, grid, nw, lds, st);
  }
}
template <int KIND, int SPS>
static hipError_t launch_gemv_s(const GemvParams& p, uint32_t scale_dt, bool asym, bool dual, int grid, int nw,
                                size_t lds, hipStream_t st) {
  if (scale_dt == DT_F32) return launch_gemv_a<KIND, SPS, SK_F32>(p, asym, dual, grid, nw, lds, st);
  if (scale_dt == DT_F16) return launch_gemv_a<KIND, SPS, SK_F16>(p, asym, dual, grid, nw, lds, st);
  return launch_gemv_a<KIND, SPS, SK_BF16>(p, asym, dual, grid, nw, lds, st);
}

static int gv_device_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 1;
  }
  return cus;
}

#ifdef NS_TRACE
unsigned long long* trace_buffer();
#endif

static std::atomic<int> g_gemv_mode{-1};  // -1: read NS_GEMV2 once; 0 off; 1 on; 2 on, whole tiles only (no stream-K part)
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

// hipErrorNotSupported: outside the kernel's envelope — the caller falls back to smallm_kernel
hipError_t launch_gemv(const SmallMArgs& a, hipStream_t st) {
  const int mode = gemv_mode();
  const ns_weight* w0 = a.seg[0].w;
  if (mode == 0 || a.m < 1 || a.m > kGvMaxRows) return hipErrorNotSupported;
  const int nq = a.dual ? 2 : 1;
  const int nmat = a.nseg;  // matrices the launch touches (dual: 2)
  GemvParams p;
  memset(&p, 0, sizeof(p));
  uint32_t tiles = 0;
  for (int i = 0; i < nmat; i++) {
    const ns_weight* w = a.seg[i].w;
    if (!w->single_span || w->alloc_bytes >= (size_t(1) << 31)) return hipErrorNotSupported;
    p.wbase[i] = reinterpret_cast<const uint8_t*>(w->codes);
    p.s_off[i] = uint32_t(reinterpret_cast<const uint8_t*>(w->scales) - reinterpret_cast<const uint8_t*>(w->codes));
    p.z_off[i] = w->zps ? uint32_t(reinterpret_cast<const uint8_t*>(w->zps) - reinterpret_cast<const uint8_t*>(w->codes)) : 0u;
    p.tile_begin[i] = tiles;
    if (!a.dual || i == 0) tiles += uint32_t(w->ntiles);
    p.c[i] = a.seg[i].c;
    p.c16[i] = static_cast<_Float16*>(a.seg[i].c16);
    p.n[i] = w->n;
  }
  for (int i = a.dual ? 1 : nmat; i < 4; i++) p.tile_begin[i] = 0xffffffffu;  // absent: no tile is >= it
  const uint32_t ks = uint32_t(w0->ksteps);
  const int kstep = w0->kstep_len;
  if (tiles == 0 || ks == 0 || uint64_t(tiles) * ks * ks >= (uint64_t(1) << 32)) return hipErrorNotSupported;
  if (!w0->ws_flags || !w0->ws_parts) return hipErrorNotSupported;

  // staged activations: [rows][ks * KSTEP + 8] halves
  const int rows = a.m;
  const uint32_t row_stride = ks * uint32_t(kstep) + 8;
  const size_t a_bytes = size_t(rows) * row_stride * 2;
  if (a_bytes > kGvMaxALds) return hipErrorNotSupported;
  p.a = a.a;
  p.a16 = static_cast<const _Float16*>(a.a16);
  if (p.a16 && ((a.lda & 7) != 0 || (w0->k & 7) != 0 || (reinterpret_cast<uintptr_t>(p.a16) & 15) != 0)) p.a16 = nullptr;
  if (uint64_t(rows) * uint64_t(a.lda) * 4 >= (uint64_t(1) << 30)) return hipErrorNotSupported;  // staging offsets

  // ---- work split ----
  const uint32_t cus = uint32_t(gv_device_cus());
  uint32_t t_dp = tiles, n_sk = 0, sk_q = 0, sk_r = 0;
  static const int env_minu = getenv("NS_GV_MIN_UNITS") ? atoi(getenv("NS_GV_MIN_UNITS")) : 4;  // diagnostics
  const uint32_t min_units = uint32_t(std::max(1, env_minu));
  if (mode != 2 && tiles % cus != 0) {
    const uint32_t per_cu_hi = (tiles + cus - 1) / cus;
    const double imbalance = double(per_cu_hi) * cus / double(tiles);
    if (imbalance > 1.04) {
      t_dp = (tiles / cus) * cus;
      const uint64_t U = uint64_t(tiles - t_dp) * ks;
      n_sk = uint32_t(std::min<uint64_t>(cus, std::max<uint64_t>(1, U / min_units)));
      if (n_sk > uint32_t(kMaxDecodeGrid / 4)) n_sk = uint32_t(kMaxDecodeGrid / 4);  // workspace: 2 KiB per workgroup
      sk_q = uint32_t(U / n_sk);
      sk_r = uint32_t(U % n_sk);
    }
  }
  const uint32_t grid = n_sk + t_dp;

  // waves per workgroup (as tuned for smallm_kernel, profiles/r01*): many tiles -> few waves each
  int nw = 8;
  {
    const int pf = a.dual ? kGvPF : kGvPFWide;
    const uint32_t per_wg = n_sk ? std::min<uint32_t>(ks, sk_q ? sk_q : ks) : ks;
    nw = (grid <= 320 && per_wg >= 32 && !a.dual) ? 16 : 8;
    if (a.dual) {
      nw = grid * 4 >= 1300 ? 4 : 8;
    } else {
      const int target_waves = 2560;
      while (nw > 2 && int(grid) * (nw / 2) >= target_waves) nw /= 2;
    }
    while (nw > 2 && int(per_wg) * nq < nw * pf) nw /= 2;  // keep the ring full
    static const int env_nw = getenv("NS_GV_NW") ? atoi(getenv("NS_GV_NW")) : 0;  // diagnostics
    if (env_nw == 2 || env_nw == 4 || env_nw == 8 || (env_nw == 16 && !a.dual)) nw = env_nw;
    if (uint32_t(nw) > ks) {
      nw = 1;
      while (uint32_t(nw) * 2 <= ks) nw *= 2;
    }
  }
  uint32_t nw_log2 = 0;
  while ((1 << nw_log2) < nw) nw_log2++;

  p.ks = ks;
  p.qstride = w0->qstride;
  p.n_sk = n_sk;
  p.sk_u0 = t_dp * ks;
  p.sk_q = sk_q;
  p.sk_r = sk_r;
  p.ks_magic = uint32_t(((uint64_t(1) << 32) + ks - 1) / ks);
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
  p.lda = a.lda;
  p.upr = ks * uint32_t(kstep) / 8;
  p.upr_magic = uint32_t(((uint64_t(1) << 32) + p.upr - 1) / p.upr);
  if (uint64_t(rows) * p.upr * p.upr >= (uint64_t(1) << 32)) return hipErrorNotSupported;
  p.row_stride = row_stride;
  p.red_off = uint32_t((a_bytes + 15) & ~size_t(15));
  p.c2 = a.c2;
  p.d = a.d;
  p.parts = w0->ws_parts;
  p.flags = w0->ws_flags;
  p.ldc = a.ldc;
  p.ldd = a.ldd;
  p.nseg = a.dual ? 1 : a.nseg;
  p.epilogue = a.epilogue;
  p.spin_limit = 1u << 18;
  if (w0->kind == WK_F4) f4_lut_planes(w0->lut, &p.lut);
  p.f8 = f8_consts(w0->qtype);
#ifdef NS_TRACE
  p.trace = trace_buffer();
#endif
  const size_t lds = size_t(p.red_off) + size_t(2) * nw * nq * 64 * 16;

#define NS_DISPATCH(KIND)                                                                         \
  switch (w0->sps) {                                                                              \
    case 4: return launch_gemv_s<KIND, 4>(p, w0->scale_dt, w0->asym, a.dual, grid, nw, lds, st);  \
    case 2: return launch_gemv_s<KIND, 2>(p, w0->scale_dt, w0->asym, a.dual, grid, nw, lds, st);  \
    default: return launch_gemv_s<KIND, 1>(p, w0->scale_dt, w0->asym, a.dual, grid, nw, lds, st); \
  }
  if (w0->kind == WK_INT4) {
    NS_DISPATCH(WK_INT4)
  } else if (w0->kind == WK_INT8) {
    if (w0->sps == 2) return launch_gemv_s<WK_INT8, 2>(p, w0->scale_dt, w0->asym, a.dual, grid, nw, lds, st);
    return launch_gemv_s<WK_INT8, 1>(p, w0->scale_dt, w0->asym, a.dual, grid, nw, lds, st);
  } else if (w0->kind == WK_F8) {  // device scales are always fp32 (E8M0 shared exponents are expanded at load)
    if (w0->sps == 2) return launch_gemv_a<WK_F8, 2, SK_F32>(p, false, a.dual, grid, nw, lds, st);
    return launch_gemv_a<WK_F8, 1, SK_F32>(p, false, a.dual, grid, nw, lds, st);
  } else {
    NS_DISPATCH(WK_F4)
  }
#undef NS_DISPATCH
}

}  // namespace ns
