// ns_gemvb.hip — gemvb_kernel: the batch-decode weight-streaming kernel (2 <= M <= 16 rows whose activations do not fit
// gemv_kernel's whole-K staging: 8 rows x 4096, anything x 14336 ...).
//
// Same arithmetic and weight path as gemv_kernel (ns_gemv.hip): 16-column tiles, the waves of a tile split its k-steps,
// every wave streams its records HBM -> LDS by DMA into a private ring, NJ x v_mfma_f32_16x16x32_f16 on the raw
// codes per record, the group scale applied to the fp32 result, deterministic cross-wave reduction, fused epilogue
// (reference: bestla/bestla/kernel_ref.h:2489-2531, :1027-1127, :1456-1478).  What differs is the ACTIVATION side:
// gemv_kernel stages all M x K activations once per workgroup (64 KiB at most); with 8 rows of 4096 that no longer fits
// and first-generation smallm_kernel — which converts and stages all of A per 16-column workgroup at one or two
// workgroups per CU — ran BASELINE config 4 (Mistral-7B NF4 g128, batch 8) at 0.16 of the HBM peak.  Here A moves through
// LDS in PHASES of `pk` k-steps, double buffered and fetched by DMA one phase ahead, so the LDS footprint is
// 2 x M x pk x 256 B whatever K is, and the weight rings run across the phase boundaries untouched (weights do not depend
// on A).  A workgroup is 16 waves = FOUR tiles x four k-step lanes sharing the staged activations: one tile per workgroup
// (first version) was bound by the latency of its own activation phases — a tile streams only 32-64 KiB of weights per
// 64 KiB of activations — and ran no faster than smallm_kernel (27.7 vs 30.7 us on Mistral's gate/up at 8 rows).
//
// Waiting by sequence number: all DMA requests of a wave retire in order, so "request X has landed" is s_waitcnt
// vmcnt(number of requests issued after X).  Activation pieces and ring records interleave in a wave's queue (which wave
// fetches which piece depends on M and the phase length), so every request batch records the wave's running request
// count when it was issued and a wait is vmcnt(issued_now - issued_then), clamped to 15 (a stricter wait is always safe).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <utility>

#include "../../include/ns_bestla.h"
#include "ns_common.h"
#include "ns_dev.h"

namespace ns {

constexpr int kGbPF = 4;        // ring slots per wave
constexpr int kGbMaxRows = 16;
constexpr size_t kGbMaxLds = 160 * 1024;

struct GemvbMat {
  const uint8_t* wbase;  // ONE allocation: records at 0, scales at s_off, zero points at z_off
  uint32_t s_off, z_off;
  uint32_t tile_begin;
  int n;
  float* c;
  _Float16* c16;
};

struct GemvbParams {
  GemvbMat mat[3];
  int nmat;                 // 1; 2 = fused gate/up (dual); 3 = fused QKV (side by side along N)
  int dual;
  const void* a;            // fp16 [m][lda]
  int m, k, lda;
  uint32_t ks, qstride, sstride, zstride, srows, srow_mul, srow_shift;
  uint32_t nw_log2;         // log2(waves per TILE: the k-step lanes)
  uint32_t tpw_log2;        // log2(tiles per workgroup)
  uint32_t tiles;           // tiles of the launch
  uint32_t pk;              // k-steps per activation phase (a multiple of the wave count)
  uint32_t a_row_bytes;     // bytes of one staged row of a phase buffer (pk x KSTEP x 2 + 16 pad)
  uint32_t a_buf_bytes;     // bytes of one phase buffer
  uint32_t ring_off, ring_stride;
  float* c2;
  const float* d;
  int ldc, ldd, epilogue;
  F4Lut lut;
  F8Consts f8;
};

template <int N>
__device__ __forceinline__ void gb_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// wait until at most n of this wave's requests are still in flight (n clamped to 15: stricter is safe)
__device__ __forceinline__ void gb_wait(uint32_t n) {
  [&]<int... K>(std::integer_sequence<int, K...>) {
    (void)((n == uint32_t(K) ? (gb_wait_vmcnt<K>(), true) : false) || ...);
  }(std::make_integer_sequence<int, 15>{});
  if (n >= 15u) gb_wait_vmcnt<15>();
}

template <int KIND, int SPS, int SK, bool ASYM, bool DUAL>
__global__ __launch_bounds__(1024) void gemvb_kernel(const GemvbParams p) {
  constexpr int NJ = kind_is_8bit(KIND) ? 2 : 4;
  constexpr int KSTEP = NJ * 32;
  constexpr int NQ = DUAL ? 2 : 1;
  constexpr int PF = kGbPF;
  constexpr int SBYTES = SPS * (SK == SK_F32 ? 4 : 2);
  constexpr uint32_t SLOT = 1024u + 16u * SBYTES + (ASYM ? 16u * SPS : 0u);
  using Corr = CorrRaw<SPS, SK, ASYM>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __attribute__((address_space(3))) unsigned char* LdsPtr;

  const int tid = threadIdx.x;
  const uint32_t wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave of the workgroup
  const int l = tid & 63, nn = l & 15, g = l >> 4;
  const uint32_t NW = 1u << p.nw_log2;                            // k-step lanes of a tile
  const uint32_t NWG = NW << p.tpw_log2;                          // waves of the workgroup
  const uint32_t w = wv & (NW - 1u), ts = wv >> p.nw_log2;        // k-step lane, tile of the workgroup
  const uint32_t ks = p.ks;
  const uint32_t Traw = (blockIdx.x << p.tpw_log2) + ts;
  const bool live = Traw < p.tiles;                               // the last workgroup may hold fewer tiles
  const uint32_t T = live ? Traw : p.tiles - 1u;

  // the matrix this tile belongs to (fused QKV: three side by side; dual: W1 and W3 share the tile index)
  int sg = 0;
  if (!DUAL && p.nmat > 1) sg = int(T >= p.mat[1].tile_begin) + int(p.nmat > 2 && T >= p.mat[2].tile_begin);
  const uint32_t tl = T - (DUAL ? 0u : p.mat[sg].tile_begin);
  Rsrc rw[NQ];
  uint32_t so[NQ], zo[NQ];
#pragma unroll
  for (int q = 0; q < NQ; q++) {
    const GemvbMat& mt = p.mat[DUAL ? q : sg];
    rw[q] = make_rsrc(mt.wbase, 0x80000000u);
    so[q] = mt.s_off;
    zo[q] = mt.z_off;
  }
  const uint32_t tile_q = tl * ks * p.qstride;
  const uint32_t tile_c = tl * p.srows;
  const I4Consts i4c = {0x000f000fu, 0x00f000f0u, 0x64006400u};

  const int rows = min(p.m, kGbMaxRows);
  const LdsPtr ring = (LdsPtr)(smem) + p.ring_off + wv * p.ring_stride;
  const uint32_t ring_base = uint32_t(reinterpret_cast<uintptr_t>(ring));

  uint32_t seq = 0;         // requests this wave has issued so far
  uint32_t slot_end[PF];    // `seq` right after the last request of the record in each ring slot
  uint32_t a_end0 = 0, a_end1 = 0;  // ... after the wave's last piece of each activation buffer

  // ---- weight record of item t (k-step ordinal t / NQ of this wave, matrix t % NQ) into ring slot t % PF ----
  auto issue_item = [&](uint32_t t) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t i = t / NQ, q = t % NQ, slot = t % PF;
    const uint32_t s = w + (i << p.nw_log2);
    const uint32_t crow = tile_c + ((s * p.srow_mul) >> p.srow_shift);
    const LdsPtr dst = ring + slot * SLOT;
    const Rsrc r = rw[NQ == 1 ? 0 : q];
    const uint32_t sof = so[NQ == 1 ? 0 : q], zof = zo[NQ == 1 ? 0 : q];
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, reinterpret_cast<__attribute__((address_space(3))) void*>(dst), 16, uint32_t(l) * 16u,
                                             tile_q + s * p.qstride, 0, 2);
    if (l < SBYTES)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r, reinterpret_cast<__attribute__((address_space(3))) void*>(dst + 1024), 16,
                                               uint32_t(l) * 16u, sof + crow * p.sstride, 0, 2);
    if constexpr (ASYM) {
      if (l < SPS)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, reinterpret_cast<__attribute__((address_space(3))) void*>(dst + 1024 + 16 * SBYTES), 16,
                                                 uint32_t(l) * 16u, zof + crow * p.zstride, 0, 2);
    }
    seq += ASYM ? 3u : 2u;
    // (slot_end is a four-entry array indexed by a wave-uniform value: selects, no scratch)
#pragma unroll
    for (int z = 0; z < PF; z++)
      if (slot == uint32_t(z)) slot_end[z] = seq;
#endif
  };

  // ---- activation phase ph into buffer ph & 1: rows x (pk x KSTEP x 2) bytes in 1 KiB pieces, piece u by wave u % NW ----
  const uint32_t ph_bytes = p.pk * uint32_t(KSTEP) * 2u;           // bytes of one row's share of a phase
  const uint32_t ppr = (ph_bytes + 1023u) >> 10;                   // pieces per row
  const uint32_t row_total = ks * uint32_t(KSTEP) * 2u;            // bytes of a whole (k-step padded) row
  const Rsrc ra = make_rsrc(p.a, uint32_t(rows - 1) * uint32_t(p.lda) * 2u + uint32_t(p.k) * 2u);
  auto issue_a = [&](uint32_t ph) {
#if defined(__HIP_DEVICE_COMPILE__)
    const LdsPtr buf = (LdsPtr)(smem) + (ph & 1u) * p.a_buf_bytes;
    const uint32_t k0 = ph * ph_bytes;  // byte offset of the phase inside a row
    for (uint32_t u = wv; u < uint32_t(rows) * ppr; u += NWG) {
      const uint32_t r = u / ppr, c = u - r * ppr;
      const uint32_t off = k0 + (c << 10);
      const uint32_t left = off < row_total ? min(row_total - off, ph_bytes - (c << 10)) : 0u;
      // `left` is wave-uniform.  Only requests that really go out are counted: a count that ran AHEAD of the hardware's
      // would make later waits allow more requests in flight than were issued after their target (too weak); a count
      // that lags only makes them stricter.
      if (left > 0u) {
        if (uint32_t(l) * 16u < left)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, reinterpret_cast<__attribute__((address_space(3))) void*>(buf + r * p.a_row_bytes + (c << 10)),
                                                   16, uint32_t(l) * 16u, r * uint32_t(p.lda) * 2u + off, 0, 0);
        seq += 1u;
      }
    }
    if (ph & 1u) a_end1 = seq; else a_end0 = seq;
#endif
  };

  const uint32_t first = w;
  const uint32_t nst = (live && first < ks) ? (ks - first + NW - 1) >> p.nw_log2 : 0u;  // k-steps of this wave
  const uint32_t nitems = nst * NQ;
  const uint32_t R = p.pk >> p.nw_log2;                                       // k-steps of a wave per phase
  const uint32_t nph = (ks + p.pk - 1) / p.pk;

#pragma unroll
  for (int z = 0; z < PF; z++) slot_end[z] = 0;
  issue_a(0);
  for (uint32_t t = 0; t < uint32_t(PF) && t < nitems; t++) issue_item(t);

  floatx4 acc[NQ];
#pragma unroll
  for (int q = 0; q < NQ; q++) acc[q] = floatx4{0.f, 0.f, 0.f, 0.f};
  const uint32_t arow = uint32_t(min(nn, rows - 1)) * p.a_row_bytes + uint32_t(g) * 16u;  // A fragment offset of this lane

  for (uint32_t ph = 0; ph < nph; ph++) {
    // the wave's own pieces of this phase have landed; then everybody's have, and everybody has left the other buffer
    gb_wait(seq - ((ph & 1u) ? a_end1 : a_end0));
    asm volatile("s_barrier" ::: "memory");
    if (ph + 1 < nph) issue_a(ph + 1);
    const unsigned char* abuf = smem + (ph & 1u) * p.a_buf_bytes + arow;
    for (uint32_t r = 0; r < R; r++) {
      const uint32_t i = ph * R + r;  // ordinal of the k-step among this wave's
      if (i >= nst) break;
      const uint32_t s = first + (i << p.nw_log2);
      const uint32_t sl = s - ph * p.pk;  // k-step inside the phase
#pragma unroll
      for (int q = 0; q < NQ; q++) {
        const uint32_t t = i * NQ + q, slot = t % PF;
        uint32_t e = slot_end[0];
#pragma unroll
        for (int z = 1; z < PF; z++)
          if (slot == uint32_t(z)) e = slot_end[z];
        gb_wait(seq - e);
        // ---- the record: scales / zero points of column nn, the lane's 16 B of codes ----
        const uint32_t sb = ring_base + slot * SLOT;
        Corr cr;
        {
          typedef __attribute__((address_space(3))) const uint32_t* L32;
          const uint32_t ca = sb + 1024u + uint32_t(nn) * SBYTES;
          if constexpr (SBYTES == 2) {
            cr.s[0] = *reinterpret_cast<__attribute__((address_space(3))) const uint16_t*>(ca);
          } else {
#pragma unroll
            for (int z = 0; z < Corr::NW32; z++) cr.s[z] = reinterpret_cast<L32>(ca)[z];
          }
          if constexpr (ASYM) {
            const uint32_t za = sb + 1024u + 16u * SBYTES + uint32_t(nn) * SPS;
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
        const uint4v qvv = *reinterpret_cast<const __attribute__((address_space(3))) uint4v*>(sb + uint32_t(l) * 16u);
        const uint32_t xw[4] = {qvv.x, qvv.y, qvv.z, qvv.w};
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
          const half8_t afrag = *reinterpret_cast<const half8_t*>(abuf + (sl * KSTEP + 32 * j) * 2);
          dd[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(afrag, bq[j], floatx4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NJ; j++) acc[q] += dd[j] * sc[j];
        // the LDS reads of this slot are done (consumed above) before its refill is requested
        __builtin_amdgcn_sched_barrier(0);
        if (t + PF < nitems) issue_item(t + PF);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  // ---- reduction over the k-step lanes of each tile (their own ring regions), lane 0 of a tile finishes it ----
  gb_wait(0);
  floatx4* red = reinterpret_cast<floatx4*>(smem + p.ring_off);
  const uint32_t kRedWave = p.ring_stride / 16;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int q = 0; q < NQ; q++) red[wv * kRedWave + q * 64 + l] = acc[q];
  __syncthreads();
  if (w != 0 || !live) return;
  floatx4 sum[NQ];
#pragma unroll
  for (int q = 0; q < NQ; q++) {
    sum[q] = floatx4{0.f, 0.f, 0.f, 0.f};
    for (uint32_t ww = 0; ww < NW; ww++) sum[q] += red[((ts << p.nw_log2) + ww) * kRedWave + q * 64 + l];
  }
  const GemvbMat& mo = p.mat[DUAL ? 0 : sg];
  const int col = int(tl) * 16 + nn;
  if (col >= mo.n) return;
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
    mo.c[size_t(row) * p.ldc + col] = v;
    if (mo.c16) mo.c16[size_t(row) * p.ldc + col] = (_Float16)v;
  }
}

// ============================================================================================================
template <int KIND, int SPS, int SK, bool ASYM>
static hipError_t launch_gemvb_k(const GemvbParams& p, int grid, int nw, size_t lds, hipStream_t st) {
  auto go = [&](auto kern) {
    static const hipError_t attr =
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(kGbMaxLds));
    if (attr != hipSuccess && lds > 64 * 1024) return attr;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(nw * 64), lds, st, p);
    return hipGetLastError();
  };
  if (p.dual) return go(gemvb_kernel<KIND, SPS, SK, ASYM, true>);
  return go(gemvb_kernel<KIND, SPS, SK, ASYM, false>);
}
template <int KIND, int SPS, int SK>
static hipError_t launch_gemvb_a(const GemvbParams& p, bool asym, int grid, int nw, size_t lds, hipStream_t st) {
  if constexpr (KIND == WK_F4 || KIND == WK_F8) {
    (void)asym;
    return launch_gemvb_k<KIND, SPS, SK, false>(p, grid, nw, lds, st);
  } else {
    if (asym) return launch_gemvb_k<KIND, SPS, SK, true>(p, grid, nw, lds, st);
    return launch_gemvb_k<KIND, SPS, SK, false>(p, grid, nw, lds, st);
  }
}
template <int KIND, int SPS>
static hipError_t launch_gemvb_s(const GemvbParams& p, uint32_t scale_dt, bool asym, int grid, int nw, size_t lds, hipStream_t st) {
  if (scale_dt == DT_F32) return launch_gemvb_a<KIND, SPS, SK_F32>(p, asym, grid, nw, lds, st);
  if (scale_dt == DT_F16) return launch_gemvb_a<KIND, SPS, SK_F16>(p, asym, grid, nw, lds, st);
  return launch_gemvb_a<KIND, SPS, SK_BF16>(p, asym, grid, nw, lds, st);
}

// hipErrorNotSupported: outside the envelope — the caller falls back to smallm_kernel
hipError_t launch_gemvb(const SmallMArgs& a, hipStream_t st) {
  static const bool off = getenv("NS_GEMVB") != nullptr && atoi(getenv("NS_GEMVB")) == 0;  // diagnostics
  const ns_weight* w0 = a.seg[0].w;
  if (off || a.m < 2 || a.m > kGbMaxRows || a.link || a.rope) return hipErrorNotSupported;
  const int kstep = w0->kstep_len;
  if (w0->k % kstep != 0 || (w0->k & 7) != 0) return hipErrorNotSupported;
  const void* a16p = a.a16;
  int lda16 = a.lda;
  if (a16p && ((a.lda & 7) != 0 || (reinterpret_cast<uintptr_t>(a16p) & 15) != 0)) a16p = nullptr;
  if (!a16p) {
    // fp32-only caller: from 5 rows on one conversion pass (about 2 us) into the per-stream scratch buys this kernel
    // (smallm_kernel converts and stages all of A in every 16-column workgroup); fewer rows stay on smallm_kernel
    if (a.m < 5 || !a.a) return hipErrorNotSupported;
    void* sc = stream_scratch(st, size_t(a.m) * w0->k * 2, 0);
    if (!sc || launch_cvt_a16(a.a, sc, a.m, w0->k, a.lda, w0->k, st) != hipSuccess) return hipErrorNotSupported;
    a16p = sc;
    lda16 = w0->k;
  }
  if (uint64_t(a.m) * uint64_t(lda16) * 2 >= (uint64_t(1) << 31)) return hipErrorNotSupported;
  GemvbParams p;
  memset(&p, 0, sizeof(p));
  uint32_t tiles = 0;
  for (int i = 0; i < a.nseg; i++) {
    const ns_weight* w = a.seg[i].w;
    if (!w->single_span || w->alloc_bytes >= (size_t(1) << 31)) return hipErrorNotSupported;
    const uint8_t* wb = reinterpret_cast<const uint8_t*>(w->codes);
    p.mat[i] = GemvbMat{wb, uint32_t(reinterpret_cast<const uint8_t*>(w->scales) - wb),
                        w->zps ? uint32_t(reinterpret_cast<const uint8_t*>(w->zps) - wb) : 0u, a.dual ? 0u : tiles, w->n,
                        a.seg[i].c, static_cast<_Float16*>(a.seg[i].c16)};
    if (!a.dual || i == 0) tiles += uint32_t(w->ntiles);
  }
  p.nmat = a.nseg;
  p.dual = a.dual ? 1 : 0;
  p.a = a16p;
  p.m = a.m, p.k = w0->k, p.lda = lda16;
  p.ks = uint32_t(w0->ksteps);
  if (tiles == 0 || p.ks == 0) return hipErrorNotSupported;
  p.qstride = w0->qstride, p.sstride = w0->sstride, p.zstride = w0->zstride;
  p.srows = uint32_t(w0->srows);
  {
    int mul, shift;
    if (!srow_params(w0, &mul, &shift)) return hipErrorNotSupported;
    p.srow_mul = uint32_t(mul), p.srow_shift = uint32_t(shift);
  }
  // four tiles x four k-step lanes per workgroup (fewer lanes for very short K); phase length so that a phase buffer stays
  // within 32 KiB (at least one k-step per lane)
  int nw = 4;
  while (nw > 1 && nw > int(p.ks)) nw /= 2;
  uint32_t nw_log2 = 0;
  while ((1 << nw_log2) < nw) nw_log2++;
  const int tpw = 16 / nw >= 4 ? 4 : 16 / nw;
  uint32_t tpw_log2 = 0;
  while ((1 << tpw_log2) < tpw) tpw_log2++;
  const int nwg = nw * tpw;
  const size_t per_r = size_t(a.m) * nw * kstep * 2;  // bytes of a phase buffer per k-step of a lane
  int r = int(std::max<size_t>(1, std::min<size_t>(8, (32 * 1024) / per_r)));
  r = std::min<int>(r, int((p.ks + nw - 1) / nw));
  p.nw_log2 = nw_log2;
  p.tpw_log2 = tpw_log2;
  p.tiles = tiles;
  p.pk = uint32_t(nw * r);
  p.a_row_bytes = p.pk * uint32_t(kstep) * 2u + 16u;
  p.a_buf_bytes = (uint32_t(a.m) * p.a_row_bytes + 15u) & ~15u;
  const int nq = a.dual ? 2 : 1;
  const uint32_t sbytes = uint32_t(w0->sps) * (w0->scale_dt == DT_F32 ? 4u : 2u);
  const uint32_t slot = 1024u + 16u * sbytes + (w0->asym ? 16u * uint32_t(w0->sps) : 0u);
  p.ring_stride = uint32_t(std::max<size_t>((size_t(kGbPF) * slot + 15) & ~size_t(15), size_t(nq) * 1024));
  p.ring_off = 2u * p.a_buf_bytes;
  const size_t lds = size_t(p.ring_off) + size_t(nwg) * p.ring_stride;
  if (lds > kGbMaxLds) return hipErrorNotSupported;
  p.c2 = a.c2;
  p.d = a.d;
  p.ldc = a.ldc, p.ldd = a.ldd, p.epilogue = a.epilogue;
  if (w0->kind == WK_F4) f4_lut_planes(w0->lut, &p.lut);
  p.f8 = f8_consts(w0->qtype);
  const int grid = int((tiles + tpw - 1) / tpw);
  nw = nwg;  // threads of the launch
#define NS_GB_DISPATCH(KIND)                                                              \
  switch (w0->sps) {                                                                      \
    case 4: return launch_gemvb_s<KIND, 4>(p, w0->scale_dt, w0->asym, grid, nw, lds, st);  \
    case 2: return launch_gemvb_s<KIND, 2>(p, w0->scale_dt, w0->asym, grid, nw, lds, st);  \
    default: return launch_gemvb_s<KIND, 1>(p, w0->scale_dt, w0->asym, grid, nw, lds, st); \
  }
  if (w0->kind == WK_INT4) {
    NS_GB_DISPATCH(WK_INT4)
  } else if (w0->kind == WK_INT8) {
    if (w0->sps == 2) return launch_gemvb_s<WK_INT8, 2>(p, w0->scale_dt, w0->asym, grid, nw, lds, st);
    return launch_gemvb_s<WK_INT8, 1>(p, w0->scale_dt, w0->asym, grid, nw, lds, st);
  } else if (w0->kind == WK_F8) {
    if (w0->sps == 2) return launch_gemvb_a<WK_F8, 2, SK_F32>(p, false, grid, nw, lds, st);
    return launch_gemvb_a<WK_F8, 1, SK_F32>(p, false, grid, nw, lds, st);
  } else {
    NS_GB_DISPATCH(WK_F4)
  }
#undef NS_GB_DISPATCH
}

}  // namespace ns
