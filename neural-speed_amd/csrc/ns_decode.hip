// ns_decode.hip — decode_kernel: persistent, bandwidth-balanced weight-streaming kernel for M <= 4 rows
// (token-by-token decode: the HBM-bound hot path of the framework).
//
// Why a second small-M kernel (measured with scripts/wave_trace.py, profiles/r01g_wave_trace.txt):
//   * MI355X shares HBM bandwidth per CU, not per wave: with one workgroup per 16-column tile, 688 tiles on 256 CUs
//     leave 176 CUs with 3 tiles and 80 with 2, and the kernel ends 2 us after the light CUs went idle.
//   * every workgroup staged all of A behind a barrier before its first MFMA: 3-4 us of a 6-12 us kernel.
// decode_kernel therefore runs ONE workgroup per CU and cuts the weight stream — which the device layout stores as
// consecutive 1 KiB(+scales) records ordered [tile][k-step] — into equal contiguous shares ("stream-K"):
//   workgroup b streams records [f0, f1), its waves interleave over them (wave w: f0+w, f0+w+NW, ...), each wave
//   keeps a kPF-deep load ring, its own fp32 accumulator and a PRIVATE fp16 copy of the A slices it needs (no
//   workgroup barrier before the first MFMA).  A wave flushes its accumulator to LDS whenever its k-step cursor
//   enters the next tile; after the stream the workgroup reduces every tile it touched in fixed wave order.
//   A tile that straddles two workgroups is finished by the one that holds its first k-step: the later workgroup
//   publishes its partial sums (fixed order too, so results are bit-reproducible) through a small per-weight
//   workspace as soon as its waves have left that tile, i.e. long before the owner needs them.
// Per k-step math is the same as smallm_kernel's (ns_kernels.hip): NJ x v_mfma_f32_16x16x32_f16 on the raw codes,
// group scale applied to the fp32 result: w = (code - zp) * scale with fp32 accumulation
// (reference: bestla/bestla/kernel_ref.h:2489-2531 gemv_4bit_fp32_fp32, :1027-1127 decompress_kblock_s4_fp).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <utility>

#include "ns_common.h"
#include "ns_dev.h"

namespace ns {

#ifndef NS_PF
#define NS_PF 4
#endif
constexpr int kDPF = NS_PF;         // k-step records each wave keeps in flight
constexpr int kDecMaxNW = 12;       // waves per workgroup: 12 x ~168 VGPRs = one workgroup per CU
constexpr int kDecMaxUnits = 512;   // 16-byte A units a wave may stage (8 loads per lane)
constexpr int kDecMaxRows = 4;
constexpr int kDecMaxLds = 150 * 1024;

// One weight as the kernel sees it: ONE allocation (ns_api.cpp alloc_weight) = code records, scales and zero points
// at 32-bit offsets from `base`.
struct DecSeg {
  const uint8_t* base;
  uint32_t s_off, z_off;
};
struct DecodeParams {
  // ---- needed before the first load can be issued: kept together at the front of the kernarg segment ----
  DecSeg seg[3];
  const _Float16* a16;
  const float* a;
  uint32_t fbase, frem;         // workgroup b streams k-step records [b*fbase + min(b, frem), ... + fbase + (b < frem))
  uint32_t ksteps, ks_magic;    // k-steps per tile; ceil(2^32 / ksteps)
  uint32_t nw, nw_magic;        // waves per workgroup; ceil(2^16 / nw)
  uint32_t qstride, sstride, zstride;
  uint32_t srows, srow_mul, srow_shift;
  uint32_t maxtl;               // tiles a workgroup can touch
  uint32_t rshift;              // A rows are stored 1 (m == 1) or 4 (m <= 4) per k-step slot
  uint32_t a_wave;              // halves of one wave's private A region
  uint32_t round_barrier;       // waves of a workgroup meet once per ring round
  uint32_t contig;              // contiguous k-step run per wave instead of every NW-th record
  int m, k, lda;
  int tile_begin[4];
  // ---- epilogue ----
  float* c[3];
  _Float16* c16[3];
  float* c2;
  const float* d;
  uint32_t* flags;  // stream-K fix-up: flags[b] = 1 when workgroup b has published parts[b]
  float* parts;     // [workgroup][NQ][16 columns][4 rows]
  int n[3];
  int ldc, ldd, nseg, epilogue;
  F4Lut lut;
#ifdef NS_TRACE
  unsigned long long* trace;
#endif
};

#ifdef NS_TRACE
#define NS_DSTAMP(i)                                                                                     \
  do {                                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 4096)                                                     \
      p.trace[(size_t(blockIdx.x) * 16 + (threadIdx.x >> 6)) * 8 + (i)] = wall_clock64();                 \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
  } while (0)
#else
#define NS_DSTAMP(i)
#endif

struct Cursor {
  uint32_t tile, s;
};

// keeps a kernarg value in an SGPR: without it hipcc turns `sg == 0 ? p.x[0] : p.x[1]` into a load from a computed
// kernarg address, i.e. one more dependent scalar-memory round trip in front of the first weight load
__device__ __forceinline__ uint32_t pin_s(uint32_t x) {
  asm volatile("" : "+s"(x));
  return x;
}
__device__ __forceinline__ uint64_t pin_s64(uint64_t x) {
  asm volatile("" : "+s"(x));
  return x;
}

// MSEG: several weights side by side along N (fused QKV); DUAL: two weights of one shape streamed in lockstep
template <int KIND, int SPS, bool DUAL, int SK, bool ASYM, bool MSEG>
__global__ __launch_bounds__(kDecMaxNW * 64) void decode_kernel(const DecodeParams p) {
  static_assert(!(DUAL && MSEG), "a dual launch has one segment");
  constexpr int NJ = (KIND == WK_INT8) ? 2 : 4;
  constexpr int KSTEP = NJ * 32;
  constexpr int NQ = DUAL ? 2 : 1;
  static_assert(kDPF % NQ == 0, "ring slots alternate between the two matrices");
  constexpr int SBYTES = SPS * (SK == SK_F32 ? 4 : 2);
  constexpr int RS = KSTEP + 8;  // halves per staged A row: +16 B keeps alignment and skews banks
  using Corr = CorrRaw<SPS, SK, ASYM>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  NS_DSTAMP(0);

  const int tid = threadIdx.x;
  const uint32_t w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = tid & 63;
  const int nn = l & 15, g = l >> 4;
  const uint32_t NW = p.nw, ks = p.ksteps;
  const uint32_t b = blockIdx.x;
  const uint32_t f0 = b * p.fbase + min(b, p.frem);
  const uint32_t f1 = f0 + p.fbase + (b < p.frem ? 1u : 0u);
  // k-steps of wave w: either every NW-th record from f0 + w (lattice) or a contiguous run of nst_max records
  // (p.contig): 12 x 256 independent sequential streams spread over the HBM channels more evenly than 256 wide ones
  const uint32_t nst_max = ((f1 - f0 + NW - 1) * p.nw_magic) >> 16;  // k-steps of the busiest wave
  const uint32_t wstride = p.contig ? 1u : NW;
  const uint32_t fw = f0 + (p.contig ? w * nst_max : w);
  const uint32_t nst = fw >= f1 ? 0u : (p.contig ? min(nst_max, f1 - fw) : ((f1 - fw + NW - 1) * p.nw_magic) >> 16);
  const uint32_t T0 = __umulhi(f0, p.ks_magic);                                    // first tile of the workgroup
  const bool incoming = T0 * ks < f0;  // tile T0 started in an earlier workgroup, which owns it
#ifdef NS_TRACE
  asm volatile("" ::"s"(T0));
  NS_DSTAMP(7);  // kernel arguments have arrived
#endif
  const int rows = p.m;

  // LDS: [partials: maxtl x NW x NQ x 256 B][arrival counter, 16 B][NW private A regions]
  floatx4* part = reinterpret_cast<floatx4*>(smem);
  uint32_t* arrivals = reinterpret_cast<uint32_t*>(smem + size_t(p.maxtl) * NW * NQ * 256);
  _Float16* a_w = reinterpret_cast<_Float16*>(smem + size_t(p.maxtl) * NW * NQ * 256 + 16) + size_t(w) * p.a_wave;
  const uint32_t a_slot = uint32_t(RS) << p.rshift;  // halves per k-step slot
  if (tid == 0) *arrivals = 0;  // made visible by the barrier that follows the ring fill

  // ---- descriptors: one per matrix, covering the weight's whole allocation (records, scales, zero points) ----
  auto seg_of = [&](uint32_t tile) {  // epilogue only
    int sg = 0;
    if constexpr (MSEG) {
      if (int(tile) >= p.tile_begin[1]) sg = 1;
      if (int(tile) >= p.tile_begin[2]) sg = 2;
    }
    return sg;
  };
  Cursor I;
  I.tile = __umulhi(fw, p.ks_magic);
  I.s = fw - I.tile * ks;
  Cursor C = I;
  // weight streams are read with plain global loads: uniform 64-bit record address (SGPR pair) + per-lane 32-bit offset
  const uint64_t wb0 = pin_s64(reinterpret_cast<uint64_t>(p.seg[0].base));
  const uint64_t wb1 = (MSEG || DUAL) ? pin_s64(reinterpret_cast<uint64_t>(p.seg[1].base)) : 0;
  const uint64_t wb2 = MSEG ? pin_s64(reinterpret_cast<uint64_t>(p.seg[2].base)) : 0;
  const uint32_t so0 = pin_s(p.seg[0].s_off), zo0 = pin_s(p.seg[0].z_off);
  const uint32_t so1 = (MSEG || DUAL) ? pin_s(p.seg[1].s_off) : 0, zo1 = (MSEG || DUAL) ? pin_s(p.seg[1].z_off) : 0;
  const uint32_t so2 = MSEG ? pin_s(p.seg[2].s_off) : 0, zo2 = MSEG ? pin_s(p.seg[2].z_off) : 0;
  const uint32_t tb1 = MSEG ? pin_s(uint32_t(p.tile_begin[1])) : 0, tb2 = MSEG ? pin_s(uint32_t(p.tile_begin[2])) : 0;
  const uint32_t tb3 = MSEG ? pin_s(uint32_t(p.tile_begin[3])) : 0;
  // current segment of the issue cursor (MSEG): plain scalars, switched arithmetically (masks, no selects: hipcc
  // turns a select chain over kernargs into a table in scratch memory)
  uint64_t cur_b = wb0;
  uint32_t cur_so = so0, cur_zo = zo0, seg_t0 = 0, seg_t1 = 0xffffffffu;
  auto load_seg = [&](uint32_t tile) {
    const uint32_t m1 = 0u - uint32_t(tile >= tb1), m2 = 0u - uint32_t(tile >= tb2);
    const uint64_t M1 = 0ull - uint64_t(tile >= tb1), M2 = 0ull - uint64_t(tile >= tb2);
    cur_b = wb0 + ((wb1 - wb0) & M1) + ((wb2 - wb1) & M2);
    cur_so = so0 + ((so1 - so0) & m1) + ((so2 - so1) & m2);
    cur_zo = zo0 + ((zo1 - zo0) & m1) + ((zo2 - zo1) & m2);
    seg_t0 = (tb1 & m1) + ((tb2 - tb1) & m2);
    seg_t1 = tb1 + ((tb2 - tb1) & m1) + ((tb3 - tb2) & m2);
  };
  if constexpr (MSEG) load_seg(I.tile);
  const uint32_t voff_q = l * 16, voff_s = nn * SBYTES, voff_z = nn * SPS;
  const I4Consts i4c = {0x000f000fu, 0x00f000f0u, 0x64006400u};

  floatx4 acc[NQ];
#pragma unroll
  for (int q = 0; q < NQ; q++) acc[q] = floatx4{0.f, 0.f, 0.f, 0.f};
  uint32_t acc_tile = C.tile;  // tile the accumulator belongs to

  uint4v qv[kDPF];
  Corr cr[kDPF];

  auto advance = [&](Cursor& c) {
    c.s += wstride;
    if (c.s >= ks) {
      c.s -= ks;
      c.tile += 1;
    }
  };
  // Every ring refill IS a load (a wave past its last k-step re-reads record 0, a cache hit): the number of loads in
  // flight is then a compile-time constant everywhere, which is what lets hipcc emit counted vmcnt waits, and all
  // waves of the workgroup run the same number of rounds (they meet at one s_barrier per round, see below)
  uint32_t ti = 0;  // k-steps issued so far
  auto issue = [&](auto slot_c) {
    constexpr int slot = decltype(slot_c)::value;
    constexpr int q = slot % NQ;
    if constexpr (MSEG) {
      if (I.tile >= seg_t1) load_seg(I.tile);
    }
    const bool ok = ti < nst;
    const uint64_t wbq = MSEG ? cur_b : (q == 0 ? wb0 : wb1);
    const uint32_t so = MSEG ? cur_so : (q == 0 ? so0 : so1), zo = MSEG ? cur_zo : (q == 0 ? zo0 : zo1);
    const uint32_t tl = I.tile - seg_t0;
    // descriptor over the weight's allocation (< 2 GiB); a dead refill adds 2 GiB to the lane offset: out of range,
    // so the load returns zeros WITHOUT touching memory but still retires in order
    const Rsrc rw = make_rsrc(reinterpret_cast<const void*>(wbq), 0x80000000u);
    const uint32_t dead = ok ? 0u : 0x80000000u;
    qv[slot] = __builtin_bit_cast(
        uint4v, __builtin_amdgcn_raw_buffer_load_b128(rw, voff_q + dead, (tl * ks + I.s) * p.qstride, 2));
    const uint32_t crow = tl * p.srows + ((I.s * p.srow_mul) >> p.srow_shift);
    corr_issue<SPS, SK, ASYM>(rw, rw, voff_s + dead, voff_z + dead, so + crow * p.sstride, zo + crow * p.zstride,
                              cr[slot]);
    if constexpr (q == NQ - 1) {
      advance(I);
      ti++;
    }
  };

  // A fragment of this lane inside a k-step slot: row min(nn, rows-1) (rows >= m only feed discarded outputs)
  const uint32_t arow_off = uint32_t(min(nn, rows - 1)) * RS + 8 * g;

  // ---- flush the accumulator of tile `acc_tile` into this wave's LDS partial; if that was the part of an
  //      incoming tile, the LAST wave to arrive publishes the workgroup's partial for the owner ----
  auto flush = [&]() {
    const uint32_t tl = acc_tile - T0;
    if (g == 0) {
#pragma unroll
      for (int q = 0; q < NQ; q++) part[((tl * NW + w) * NQ + q) * 16 + nn] = acc[q];
    }
#pragma unroll
    for (int q = 0; q < NQ; q++) acc[q] = floatx4{0.f, 0.f, 0.f, 0.f};
    if (tl == 0 && incoming) {
      const uint32_t e0 = min(f1, (T0 + 1) * ks);
      // waves 0..ncontrib-1 have k-steps in [f0, e0)
      uint32_t ncontrib = min(NW, e0 - f0);
      if (p.contig) {
        ncontrib = 0;
        for (uint32_t w2 = 0; w2 < NW; w2++) ncontrib += (f0 + w2 * nst_max < e0) ? 1u : 0u;
      }
      // LDS executes a wave's instructions in order, so "partial written, then counter bumped" needs no fence;
      // the asm statements only stop the compiler from moving LDS accesses across the atomic
      asm volatile("" ::: "memory");
      uint32_t old = 0;
      if (l == 0) old = __hip_atomic_fetch_add(arrivals, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      old = __builtin_amdgcn_readfirstlane(old);
      asm volatile("" ::: "memory");
      if (old + 1 == ncontrib) {
        // device-scope (sc1) stores go straight to memory: no L2 write-back / invalidate of a whole XCD's cache,
        // which is what a release fence costs on MI355X (8 non-coherent L2s)
        if (g == 0) {
#pragma unroll
          for (int q = 0; q < NQ; q++) {
            floatx4 sum = floatx4{0.f, 0.f, 0.f, 0.f};
            for (uint32_t w2 = 0; w2 < ncontrib; w2++) sum += part[((0 * NW + w2) * NQ + q) * 16 + nn];
            float* dst = p.parts + ((size_t(b) * NQ + q) * 16 + nn) * 4;
#pragma unroll
            for (int e = 0; e < 4; e++) __hip_atomic_store(dst + e, sum[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the partial has reached memory before the flag is raised
        if (l == 0) __hip_atomic_store(p.flags + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  };

  auto compute = [&](auto slot_c, const _Float16* abase) {
    constexpr int slot = decltype(slot_c)::value;
    constexpr int q = slot % NQ;
#if defined(NS_ABLATE) && (NS_ABLATE & 1)  // diagnostics: keep the loads alive, skip all math
    acc[q][0] += __builtin_bit_cast(float, (qv[slot].x ^ qv[slot].y ^ qv[slot].z ^ qv[slot].w ^ cr[slot].s[0]) & 0x007fffffu);
    (void)abase;
    return;
#endif
    float sc[4], zp[4];
    corr_decode<SPS, SK, ASYM, NJ>(cr[slot], sc, zp);
    const uint32_t xw[4] = {qv[slot].x, qv[slot].y, qv[slot].z, qv[slot].w};
    half8_t bfr[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) {
      if constexpr (KIND == WK_INT4) {
        const _Float16 zl = (_Float16)(-1032.f - zp[j]), zh = (_Float16)(-72.f - zp[j]);
        bfr[j] = cvt_i4x8(xw[j], i4c, half2_t{zl, zl}, half2_t{zh, zh});
      } else if constexpr (KIND == WK_INT8) {
        const _Float16 zo = (_Float16)(-1152.f - zp[j]);
        bfr[j] = cvt_i8x8(xw[2 * j], xw[2 * j + 1], half2_t{zo, zo});
      } else {
        bfr[j] = cvt_f4x8(xw[j], p.lut);
      }
    }
    floatx4 dd[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) {
      const half8_t afrag = *reinterpret_cast<const half8_t*>(abase + arow_off + 32 * j);
      dd[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(afrag, bfr[j], floatx4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < NJ; j++) acc[q] += dd[j] * sc[j];
  };

#define NS_FOR_SLOTS(BODY)                                        \
  {                                                               \
    [&]<int... II>(std::integer_sequence<int, II...>) {           \
      (([&] { constexpr int i = II; std::integral_constant<int, II> ic; (void)i; (void)ic; BODY }()), ...); \
    }(std::make_integer_sequence<int, kDPF>{});                   \
  }

  // ---- 1. fill the weight ring: the HBM requests go out before anything else is set up ----
  NS_FOR_SLOTS({ issue(ic); })
  __builtin_amdgcn_sched_barrier(0);
  NS_DSTAMP(1);

  // ---- 2. this wave's A slices (L2 / MALL hits: they return right behind the first weight records) ----
  constexpr int CHB = (KSTEP == 128) ? 4 : 3;      // log2 of the 16-byte chunks per k-step row (KSTEP / 8)
  const uint32_t units = nst << (CHB + p.rshift);  // 16-byte units: [k-step ordinal][row slot][chunk]
  const uint32_t rmask = (1u << p.rshift) - 1u;
  constexpr int kAIt = kDecMaxUnits / 64;
  uint4v areg[kAIt];
  uint32_t adst[kAIt];  // LDS destination (halves) or ~0u
  int atail[kAIt];      // valid halves of the unit (8 unless it straddles K)
  const bool use_a16 = p.a16 != nullptr;
  const Rsrc ra = use_a16 ? make_rsrc(p.a16, uint32_t(rows) * uint32_t(p.lda) * 2u)
                          : make_rsrc(p.a, uint32_t(rows) * uint32_t(p.lda) * 4u);
  auto a_unit = [&](int it, uint32_t& off_elems) {
    const uint32_t u = uint32_t(l) + 64u * uint32_t(it);
    const uint32_t i = u >> (CHB + p.rshift), r = (u >> CHB) & rmask, ch = u & ((1u << CHB) - 1u);
    const uint32_t x = C.s + i * wstride;  // k-step of ordinal i, modulo ks  (C = the wave's first k-step)
    const uint32_t sx = x - __umulhi(x, p.ks_magic) * ks;
    const uint32_t gk = sx * KSTEP + ch * 8;
    const bool slot_ok = u < units && int(r) < rows;  // slots past K (k-step padding) are stored as zeros
    adst[it] = slot_ok ? (i * a_slot + r * RS + ch * 8) : 0xffffffffu;
    atail[it] = max(0, min(8, p.k - int(gk)));
    off_elems = r * uint32_t(p.lda) + gk;
    return slot_ok && int(gk) < p.k;  // otherwise the load is sent out of range and returns 0
  };
  if (use_a16) {
#pragma unroll
    for (int it = 0; it < kAIt; it++) {
      if (it < 2 || units > 128u) {
        uint32_t off;
        const bool valid = a_unit(it, off);
        areg[it] = __builtin_bit_cast(
            uint4v, __builtin_amdgcn_raw_buffer_load_b128(ra, valid ? off * 2u : 0x80000000u, 0, 0));
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();  // orders the arrival-counter reset; free: every wave's loads are already in flight

  // ---- 3. A units -> private LDS region (fp16); units straddling K are zero-filled past K ----
  auto a_store = [&](int it, uint4v v) {
    if (adst[it] != 0xffffffffu) {
      if (atail[it] < 8) {
        uint32_t* vw = reinterpret_cast<uint32_t*>(&v);
#pragma unroll
        for (int e = 0; e < 8; e++)
          if (e >= atail[it]) vw[e >> 1] &= (e & 1) ? 0x0000ffffu : 0xffff0000u;
      }
      *reinterpret_cast<uint4v*>(a_w + adst[it]) = v;
    }
  };
  if (use_a16) {
#pragma unroll
    for (int it = 0; it < kAIt; it++)
      if (it < 2 || units > 128u) a_store(it, areg[it]);
  } else {
    // fp32 activations (no fp16 shadow): convert on the way in
    for (int it = 0; it * 64 < int(units); it++) {
      const uint32_t u = uint32_t(l) + 64u * uint32_t(it);
      const uint32_t i = u >> (CHB + p.rshift), r = (u >> CHB) & rmask, ch = u & ((1u << CHB) - 1u);
      const uint32_t x = C.s + i * wstride;
      const uint32_t sx = x - __umulhi(x, p.ks_magic) * ks;
      const uint32_t gk = sx * KSTEP + ch * 8;
      const bool slot_ok = u < units && int(r) < rows;
      const uint32_t off = (slot_ok && int(gk) < p.k) ? (r * uint32_t(p.lda) + gk) * 4u : 0x80000000u;
      float f[8];
      if ((p.lda & 3) == 0 && (reinterpret_cast<uintptr_t>(p.a) & 15) == 0) {
        const uint4v v0 = __builtin_bit_cast(uint4v, __builtin_amdgcn_raw_buffer_load_b128(ra, off, 0, 0));
        const uint4v v1 = __builtin_bit_cast(uint4v, __builtin_amdgcn_raw_buffer_load_b128(ra, off, 16, 0));
        const uint32_t vw[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
        for (int e = 0; e < 8; e++) f[e] = __builtin_bit_cast(float, vw[e]);
      } else {
#pragma unroll
        for (int e = 0; e < 8; e++)
          f[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, off, 4 * e, 0));
      }
#pragma unroll
      for (int e = 0; e < 8; e++)
        if (int(gk) + e >= p.k) f[e] = 0.f;
      if (slot_ok) {
        half2_t h0 = {(_Float16)f[0], (_Float16)f[1]}, h1 = {(_Float16)f[2], (_Float16)f[3]};
        half2_t h2 = {(_Float16)f[4], (_Float16)f[5]}, h3 = {(_Float16)f[6], (_Float16)f[7]};
        *reinterpret_cast<uint4v*>(a_w + (i * a_slot + r * RS + ch * 8)) =
            uint4v{as_u32(h0), as_u32(h1), as_u32(h2), as_u32(h3)};
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  NS_DSTAMP(2);

  // ---- 4. stream.  Same trip count for every wave of the workgroup; one s_barrier per round keeps the waves within a
  //      round of each other, so the CU's load queue stays full until the very end instead of draining wave by wave
  //      (measured: without it the waves of a workgroup finish up to 4 us apart) ----
  const _Float16* a_cur = a_w;
  uint32_t tc = 0;  // k-steps consumed so far
  auto consume = [&](auto slot_c) {
    constexpr int slot = decltype(slot_c)::value;
    constexpr int q = slot % NQ;
    if (tc < nst) {
      if constexpr (q == 0) {
        if (C.tile != acc_tile) {
          flush();
          acc_tile = C.tile;
        }
      }
      compute(slot_c, a_cur);
    }
    if constexpr (q == NQ - 1) {
      advance(C);
      a_cur += a_slot;
      tc++;
    }
  };
  const uint32_t nrounds = (nst_max * NQ + uint32_t(kDPF) - 1) / uint32_t(kDPF);  // >= 1
#ifdef NS_TRACE
  asm volatile("" ::"v"(qv[0].x));
  NS_DSTAMP(3);
#endif
  for (uint32_t r = 0; r + 1 < nrounds; r++) {
    NS_FOR_SLOTS({
      consume(ic);
      __builtin_amdgcn_sched_barrier(0);
      issue(ic);
      __builtin_amdgcn_sched_barrier(0);
    })
    if (p.round_barrier) __builtin_amdgcn_s_barrier();
  }
  NS_FOR_SLOTS({
    consume(ic);
    __builtin_amdgcn_sched_barrier(0);
  })
#undef NS_FOR_SLOTS
  if (nst > 0) flush();
  NS_DSTAMP(4);
  __syncthreads();
  NS_DSTAMP(5);

  // ---- 5. reduce every tile this workgroup owns (waves take tiles round-robin), fixed summation order ----
  const uint32_t Tlast = __umulhi(f1 - 1, p.ks_magic);
  const uint32_t ntl = Tlast - T0 + 1;
  for (uint32_t tl = (incoming ? 1u : 0u) + w; tl < ntl; tl += NW) {
    const uint32_t T = T0 + tl;
    const uint32_t ra0 = max(f0, T * ks), re0 = min(f1, (T + 1) * ks);
    floatx4 sum[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) sum[q] = floatx4{0.f, 0.f, 0.f, 0.f};
    for (uint32_t w2 = 0; w2 < NW; w2++) {
      bool touched;
      if (p.contig) {
        const uint32_t fw2 = f0 + w2 * nst_max;
        touched = max(fw2, ra0) < min(min(fw2 + nst_max, f1), re0);
      } else {
        const uint32_t fw2 = f0 + w2;
        uint32_t first = fw2;  // first k-step >= ra0 on wave w2's lattice
        if (fw2 < ra0) first = fw2 + (((ra0 - fw2 + NW - 1) * p.nw_magic) >> 16) * NW;
        touched = first < re0;
      }
      if (touched && g == 0) {
#pragma unroll
        for (int q = 0; q < NQ; q++) sum[q] += part[((tl * NW + w2) * NQ + q) * 16 + nn];
      }
    }
    if ((T + 1) * ks > f1) {  // continued by later workgroups: add their published partials, in order
      for (uint32_t c = b + 1; c < gridDim.x; c++) {
        const uint32_t f0c = c * p.fbase + min(c, p.frem);
        if (f0c >= (T + 1) * ks) break;
        // bounded: a protocol bug must not hang the GPU (the result is then wrong and the parity tests say so)
        for (uint32_t spin = 0; spin < (1u << 20); spin++) {
          if (__hip_atomic_load(p.flags + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
          __builtin_amdgcn_s_sleep(1);
        }
        asm volatile("" ::: "memory");
        if (g == 0) {
#pragma unroll
          for (int q = 0; q < NQ; q++) {
            const float* src = p.parts + ((size_t(c) * NQ + q) * 16 + nn) * 4;
#pragma unroll
            for (int e = 0; e < 4; e++) sum[q][e] += __hip_atomic_load(src + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (l == 0) __hip_atomic_store(p.flags + c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    // epilogue: lanes g == 0 hold rows 0..3 of column nn
    const int sg = seg_of(T);
    const int tile_in_seg = int(T) - (sg == 0 ? 0 : (sg == 1 ? p.tile_begin[1] : p.tile_begin[2]));
    const int col = tile_in_seg * 16 + nn;
    const int ncols = sg == 0 ? p.n[0] : (sg == 1 ? p.n[1] : p.n[2]);
    if (g == 0 && col < ncols) {
      float* cbase = sg == 0 ? p.c[0] : (sg == 1 ? p.c[1] : p.c[2]);
      _Float16* c16 = sg == 0 ? p.c16[0] : (sg == 1 ? p.c16[1] : p.c16[2]);
#pragma unroll
      for (int rr = 0; rr < kDecMaxRows; rr++) {
        if (rr >= p.m) continue;
        float v = sum[0][rr];
        if constexpr (DUAL) {
          // tmp1 = act(A*W1) ; tmp2 = (A*W3) * tmp1   (neural_speed/core/layers/ip_fusion_ffn.cpp:364-406)
          const float t1 = (p.epilogue == 5) ? epi_silu(v) : epi_gelu(v);
          if (p.c2) p.c2[size_t(rr) * p.ldc + col] = t1;
          v = sum[1][rr] * t1;
        } else {
          const float dv = p.d ? p.d[size_t(rr) * p.ldd + col] : 0.f;
          switch (p.epilogue) {
            case 1: v = v + dv; break;            // custom::epilogue::Add
            case 2: v = v * dv; break;            // custom::epilogue::Mul
            case 3: v = epi_gelu(v + dv); break;  // custom::epilogue::Add_Gelu
            case 4: v = epi_gelu(v); break;
            case 5: v = epi_silu(v); break;
            default: break;
          }
        }
        cbase[size_t(rr) * p.ldc + col] = v;
        if (c16) c16[size_t(rr) * p.ldc + col] = (_Float16)v;
      }
    }
  }
  NS_DSTAMP(6);
}

// ============================================================================================================
// host side
// ============================================================================================================
template <int KIND, int SPS, int SK, bool ASYM>
static hipError_t launch_decode_k(const DecodeParams& p, bool dual, int grid, int nw, size_t lds, hipStream_t st) {
  const dim3 g(grid), b(nw * 64);
  if (dual) {
    auto k = decode_kernel<KIND, SPS, true, SK, ASYM, false>;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, kDecMaxLds);
    if (attr != hipSuccess && lds > 64 * 1024) return attr;
    hipLaunchKernelGGL(k, g, b, lds, st, p);
  } else if (p.nseg > 1) {
    auto k = decode_kernel<KIND, SPS, false, SK, ASYM, true>;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, kDecMaxLds);
    if (attr != hipSuccess && lds > 64 * 1024) return attr;
    hipLaunchKernelGGL(k, g, b, lds, st, p);
  } else {
    auto k = decode_kernel<KIND, SPS, false, SK, ASYM, false>;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, kDecMaxLds);
    if (attr != hipSuccess && lds > 64 * 1024) return attr;
    hipLaunchKernelGGL(k, g, b, lds, st, p);
  }
  return hipGetLastError();
}
template <int KIND, int SPS, int SK>
static hipError_t launch_decode_a(const DecodeParams& p, bool asym, bool dual, int grid, int nw, size_t lds,
                                  hipStream_t st) {
  if constexpr (KIND == WK_F4) {
    (void)asym;
    return launch_decode_k<KIND, SPS, SK, false>(p, dual, grid, nw, lds, st);
  } else {
    if (asym) return launch_decode_k<KIND, SPS, SK, true>(p, dual, grid, nw, lds, st);
    return launch_decode_k<KIND, SPS, SK, false>(p, dual, grid, nw, lds, st);
  }
}
template <int KIND, int SPS>
static hipError_t launch_decode_s(const DecodeParams& p, uint32_t scale_dt, bool asym, bool dual, int grid, int nw,
                                  size_t lds, hipStream_t st) {
#ifdef NS_DECODE_MIN  // development builds: bf16 scales only
  return launch_decode_a<KIND, SPS, SK_BF16>(p, asym, dual, grid, nw, lds, st);
#else
  if (scale_dt == DT_F32) return launch_decode_a<KIND, SPS, SK_F32>(p, asym, dual, grid, nw, lds, st);
  if (scale_dt == DT_F16) return launch_decode_a<KIND, SPS, SK_F16>(p, asym, dual, grid, nw, lds, st);
  return launch_decode_a<KIND, SPS, SK_BF16>(p, asym, dual, grid, nw, lds, st);
#endif
}

static int device_cus() {
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

// Returns hipErrorNotSupported when the shape is outside decode_kernel's envelope (the caller then uses
// smallm_kernel); any other error is a launch failure.
hipError_t launch_decode(const SmallMArgs& a, hipStream_t st) {
  // Experimental: measured slower than smallm_kernel on MI355X for the Llama-7B shapes (profiles/r01g_*), so it is
  // opt-in (NS_DECODE_KERNEL=1) until it wins
  static const bool off = getenv("NS_DECODE_KERNEL") == nullptr || atoi(getenv("NS_DECODE_KERNEL")) == 0;
  const ns_weight* w0 = a.seg[0].w;
  if (off || a.m > kDecMaxRows || a.m < 1 || !w0->ws_flags || w0->kind == WK_F8) return hipErrorNotSupported;
#ifdef NS_DECODE_MIN
  if (w0->scale_dt != DT_BF16) return hipErrorNotSupported;
#endif
  const int nq = a.dual ? 2 : 1;
  DecodeParams p;
  memset(&p, 0, sizeof(p));
  int tiles = 0;
  for (int i = 0; i < a.nseg; i++) {
    const ns_weight* w = a.seg[i].w;
    if (!w->single_span || w->alloc_bytes >= (size_t(1) << 31)) return hipErrorNotSupported;
    p.tile_begin[i] = tiles;
    if (!a.dual || i == 0) tiles += w->ntiles;
    p.seg[i].base = reinterpret_cast<const uint8_t*>(w->codes);
    p.seg[i].s_off = w->s_off;
    p.seg[i].z_off = w->z_off;
    p.c[i] = a.seg[i].c;
    p.c16[i] = static_cast<_Float16*>(a.seg[i].c16);
    p.n[i] = w->n;
  }
  for (int i = a.nseg; i < 4; i++) p.tile_begin[i] = tiles;
  if (a.dual) p.tile_begin[1] = p.tile_begin[2] = p.tile_begin[3] = tiles;
  const uint64_t ks = uint64_t(w0->ksteps);
  const uint64_t F = uint64_t(tiles) * ks;  // k-step records to stream (per matrix)
  if (F == 0 || F * ks >= (uint64_t(1) << 32) || ks < 4) return hipErrorNotSupported;

  // workgroups: one per CU, fewer when there is too little to stream; waves: enough k-steps each to fill the ring
  int grid = device_cus();
  if (grid > kMaxDecodeGrid) grid = kMaxDecodeGrid;
  static const int env_nw = getenv("NS_DEC_NW") ? atoi(getenv("NS_DEC_NW")) : 0;  // diagnostics
  static const int env_grid = getenv("NS_DEC_GRID") ? atoi(getenv("NS_DEC_GRID")) : 0;
  if (env_grid > 0 && env_grid <= kMaxDecodeGrid) grid = env_grid;
  const uint64_t min_share = 8;  // k-steps
  if (F / uint64_t(grid) < min_share) grid = int(F / min_share) > 0 ? int(F / min_share) : 1;
  const uint32_t fbase = uint32_t(F / uint64_t(grid)), frem = uint32_t(F % uint64_t(grid));
  int nw = kDecMaxNW;
  while (nw > 2 && uint64_t(fbase) * nq < uint64_t(nw) * 4) nw -= 2;  // >= 4 ring items per wave
  if (env_nw >= 1 && env_nw <= kDecMaxNW) nw = env_nw;
  if (uint64_t(nw) > ks) nw = int(ks);
  if ((fbase + 1 + nw) * uint64_t(nw) >= 65536) return hipErrorNotSupported;  // nw_magic division range
  const uint32_t nst_max = (fbase + 1 + nw - 1) / nw;
  const uint32_t rshift = a.m == 1 ? 0 : 2;
  if ((nst_max << (4 + rshift)) > uint32_t(kDecMaxUnits)) return hipErrorNotSupported;
  const int kstep = w0->kstep_len;
  const uint32_t maxtl = uint32_t((fbase + 1 + ks - 1) / ks) + 1;
  const uint32_t a_wave = nst_max * ((uint32_t(kstep) + 8) << rshift);  // halves
  const size_t lds = size_t(maxtl) * nw * nq * 256 + 16 + size_t(nw) * a_wave * 2;
  if (lds > size_t(kDecMaxLds)) return hipErrorNotSupported;

  p.a = a.a;
  p.a16 = static_cast<const _Float16*>(a.a16);
  if (p.a16 && ((a.lda & 7) != 0 || (reinterpret_cast<uintptr_t>(p.a16) & 15) != 0)) p.a16 = nullptr;
  p.c2 = a.c2;
  p.d = a.d;
  p.flags = w0->ws_flags;
  p.parts = w0->ws_parts;
  p.fbase = fbase;
  p.frem = frem;
  p.ksteps = uint32_t(ks);
  p.ks_magic = uint32_t(((uint64_t(1) << 32) + ks - 1) / ks);
  p.nw = uint32_t(nw);
  p.nw_magic = uint32_t((65536 + nw - 1) / nw);
  p.qstride = w0->qstride;
  p.sstride = w0->sstride;
  p.zstride = w0->zstride;
  p.srows = uint32_t(w0->srows);
  {
    int mul, shift;
    if (!srow_params(w0, &mul, &shift)) return hipErrorNotSupported;
    p.srow_mul = uint32_t(mul), p.srow_shift = uint32_t(shift);
  }
  p.maxtl = maxtl;
  p.rshift = rshift;
  p.a_wave = a_wave;
  static const bool no_rb = getenv("NS_DEC_NO_ROUND_BARRIER") != nullptr;  // diagnostics
  p.round_barrier = no_rb ? 0u : 1u;
  static const int env_contig = getenv("NS_DEC_CONTIG") ? atoi(getenv("NS_DEC_CONTIG")) : 1;  // diagnostics
  p.contig = env_contig ? 1u : 0u;
  p.m = a.m;
  p.k = w0->k;
  p.lda = a.lda;
  p.ldc = a.ldc;
  p.ldd = a.ldd;
  p.nseg = a.dual ? 1 : a.nseg;
  p.epilogue = a.epilogue;
  if (w0->kind == WK_F4) f4_lut_planes(w0->lut, &p.lut);
#ifdef NS_TRACE
  p.trace = trace_buffer();
#endif

#ifdef NS_DECODE_MIN
  if (w0->kind != WK_INT4 || w0->sps != 4) return hipErrorNotSupported;
  return launch_decode_s<WK_INT4, 4>(p, w0->scale_dt, w0->asym, a.dual, grid, nw, lds, st);
#else
#define NS_DISPATCH(KIND)                                                                           \
  switch (w0->sps) {                                                                                \
    case 4: return launch_decode_s<KIND, 4>(p, w0->scale_dt, w0->asym, a.dual, grid, nw, lds, st);  \
    case 2: return launch_decode_s<KIND, 2>(p, w0->scale_dt, w0->asym, a.dual, grid, nw, lds, st);  \
    default: return launch_decode_s<KIND, 1>(p, w0->scale_dt, w0->asym, a.dual, grid, nw, lds, st); \
  }
  if (w0->kind == WK_INT4) {
    NS_DISPATCH(WK_INT4)
  } else if (w0->kind == WK_INT8) {
    if (w0->sps == 2) return launch_decode_s<WK_INT8, 2>(p, w0->scale_dt, w0->asym, a.dual, grid, nw, lds, st);
    return launch_decode_s<WK_INT8, 1>(p, w0->scale_dt, w0->asym, a.dual, grid, nw, lds, st);
  } else {
    NS_DISPATCH(WK_F4)
  }
#undef NS_DISPATCH
#endif
}

}  // namespace ns
