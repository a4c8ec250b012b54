/*
 * pack_shim.cpp — TEST INFRASTRUCTURE.  The reference's REAL weight packer: prologue_b::gemm::WeightKBlockNInteger /
 * WeightKBlockNFloat (bestla/bestla/bestla_prologue_b.h, compiled from /root/reference where it lies) — the body of
 * BTLAGemmQuantPackB / BTLAGemmPackB / BTLAGemmUnPackB (neural_speed/core/layers/bestla_gemm.cpp:302-319, :401-422,
 * :673-749): createStorage -> assign -> packTransposeWeight (quantize -> padding-interleave -> bit-plane compress ->
 * reduce) — instantiated over stand-in GEMM core types that carry only what the packer reads from a core (NTILE, KTILE,
 * PACK_ROW, ID, BType; values of bestla_gemm.h's nine cores).  What does NOT build is the GEMM itself (the JIT micro-kernels
 * need the un-vendored xbyak); oracle/Makefile (target packref) satisfies the packer's includes with generated stand-ins:
 *   bestla_gemm.h        = lines 1-123 of the reference's (CompType, CoreAttr), as for the storref target
 *   kernel_jit.h         = <immintrin.h> + trapping stand-ins for the three JIT entries the kernel headers name
 *   xbyak/xbyak_util.h   = a CPUID-based `Xbyak::util::Cpu` (xbyak is the reference's third-party dependency for CPU
 *                          detection; NS_PACKREF_ISA=nosimd makes it report no vector ISA, so that the reference's
 *                          runtime dispatch takes its scalar kernels — the ones the oracle restates)
 * Built into oracle/_ref/libpack_ref.so; tests/test_oracle_vs_packer.py compares the oracle's blobs with this packer's
 * byte for byte.
 */
#include "bestla_prologue_b.h"

#include <cstdint>
#include <cstring>

using namespace bestla;  // NOLINT

namespace {

template <int NT, int KT, int PR, gemm::CompType COMP_, BTLA_ISA ISA_, typename BT>
struct Core {
  static constexpr int NTILE = NT, KTILE = KT, PACK_ROW = PR;
  static constexpr auto COMP = COMP_;
  static constexpr auto ISA = ISA_;
  using BType = BT;
  static constexpr auto ID = gemm::CoreAttr::make_core_id(NT, PR, COMP_, ISA_);
};
// the nine cores a blob can be laid out for (neural_speed/core/layers/bestla_defs.h:36-54; bestla_gemm.h class definitions)
using C0 = Core<24, 1, 1, gemm::CompType::COMP_FP32, BTLA_ISA::AVX2, float>;
using C1 = Core<48, 1, 1, gemm::CompType::COMP_FP32, BTLA_ISA::AVX512F, float>;
using C2 = Core<48, 32, 2, gemm::CompType::COMP_BF16_FP32, BTLA_ISA::AMX_BF16, utils::bf16>;
using C3 = Core<48, 32, 2, gemm::CompType::COMP_FP16_FP32, BTLA_ISA::AMX_FP16, utils::fp16>;
using C4 = Core<48, 4, 4, gemm::CompType::COMP_INT8_US_FP32, BTLA_ISA::AVX512_VNNI, int8_t>;
using C5 = Core<48, 4, 4, gemm::CompType::COMP_INT8_US_FP32, BTLA_ISA::AVX512BW, int8_t>;
using C6 = Core<24, 4, 4, gemm::CompType::COMP_INT8_US_FP32, BTLA_ISA::AVX_VNNI, int8_t>;
using C7 = Core<24, 4, 4, gemm::CompType::COMP_INT8_US_FP32, BTLA_ISA::AVX2, int8_t>;
using C8 = Core<48, 64, 4, gemm::CompType::COMP_INT8_US_FP32, BTLA_ISA::AMX_INT8, int8_t>;

bool is_float_weight(BTLA_DTYPE q) {
  return q == BTLA_DTYPE::F4_BNB || q == BTLA_DTYPE::F4_NF4 || q == BTLA_DTYPE::F4_E2M1 || q == BTLA_DTYPE::F8_E4M3 ||
         q == BTLA_DTYPE::F8_E5M2;
}

template <class CoreT>
size_t do_size(int n, int k, int bs, BTLA_DTYPE q, BTLA_DTYPE s, bool asym) {
  if (is_float_weight(q)) return prologue_b::gemm::WeightKBlockNFloat<CoreT>::createStorage(n, k, bs, q, s).mSize;
  return prologue_b::gemm::WeightKBlockNInteger<CoreT>::createStorage(n, k, bs, q, s, BTLA_DTYPE::BF16, asym).mSize;
}

template <class CoreT>
int do_pack(void* blob, const float* w, int n, int k, int ldw, int bs, BTLA_DTYPE q, BTLA_DTYPE s, bool asym, bool trans) {
  parallel::SingleThread th;
  if (is_float_weight(q)) {
    using P = prologue_b::gemm::WeightKBlockNFloat<CoreT>;
    auto stor = P::createStorage(n, k, bs, q, s);
    stor.assign(reinterpret_cast<int8_t*>(blob));
    if (trans) P::packTransposeWeight(n, k, w, ldw, &stor, &th); else P::packWeight(n, k, w, ldw, &stor, &th);
  } else {
    using P = prologue_b::gemm::WeightKBlockNInteger<CoreT>;
    auto stor = P::createStorage(n, k, bs, q, s, BTLA_DTYPE::BF16, asym);
    stor.assign(reinterpret_cast<int8_t*>(blob));
    if (trans) P::packTransposeWeight(n, k, w, ldw, &stor, &th); else P::packWeight(n, k, w, ldw, &stor, &th);
  }
  return 0;
}

// BTLAGemmPackBImpl (bestla_gemm.cpp:400-422): pre-quantized codes [K][N], fp32 scales / zero points [nblk][N], optional
// GPTQ act-order group indices (enableShuffle + setShuffleIndices), integer weights
template <class CoreT>
int do_pack_q(void* blob, const int8_t* q, int ldq, const float* scales, const int8_t* zps, int n, int k, int bs, BTLA_DTYPE qt,
              BTLA_DTYPE st, bool asym, int* g_idx) {
  parallel::SingleThread th;
  using P = prologue_b::gemm::WeightKBlockNInteger<CoreT>;
  auto stor = P::createStorage(n, k, bs, qt, st, BTLA_DTYPE::BF16, asym);
  if (g_idx) P::enableShuffle(&stor);
  stor.assign(reinterpret_cast<int8_t*>(blob));
  if (g_idx) P::setShuffleIndices(g_idx, &stor, &th);
  P::packQWeight(n, k, q, ldq, scales, asym ? zps : nullptr, &stor, &th);
  return 0;
}

template <class CoreT>
size_t do_size_gidx(int n, int k, int bs, BTLA_DTYPE q, BTLA_DTYPE s, bool asym) {
  using P = prologue_b::gemm::WeightKBlockNInteger<CoreT>;
  auto stor = P::createStorage(n, k, bs, q, s, BTLA_DTYPE::BF16, asym);
  P::enableShuffle(&stor);
  return stor.mSize;
}

template <class CoreT>
int do_unpack(void* blob, float* out, int ld, bool is_float) {
  parallel::SingleThread th;
  if (is_float) {
    using P = prologue_b::gemm::WeightKBlockNFloat<CoreT>;
    typename P::StorageWeight stor(0);
    stor.deserialize(reinterpret_cast<int8_t*>(blob));
    P::unpackWeight(stor.mN, stor.mK, &stor, out, ld, &th);
  } else {
    using P = prologue_b::gemm::WeightKBlockNInteger<CoreT>;
    typename P::StorageWeight stor(0);
    stor.deserialize(reinterpret_cast<int8_t*>(blob));
    P::unpackWeight(stor.mN, stor.mK, &stor, out, ld, &th);
  }
  return 0;
}

#define NS_CORE_SWITCH(core, EXPR)  \
  switch (core) {                   \
    case 0: { using CT = C0; EXPR; } \
    case 1: { using CT = C1; EXPR; } \
    case 2: { using CT = C2; EXPR; } \
    case 3: { using CT = C3; EXPR; } \
    case 4: { using CT = C4; EXPR; } \
    case 5: { using CT = C5; EXPR; } \
    case 6: { using CT = C6; EXPR; } \
    case 7: { using CT = C7; EXPR; } \
    case 8: { using CT = C8; EXPR; } \
    default: break;                 \
  }

}  // namespace

extern "C" {

size_t packref_size(int n, int k, int blocksize, uint32_t qtype, uint32_t stype, int asym, int core) {
  NS_CORE_SWITCH(core, return do_size<CT>(n, k, blocksize, (BTLA_DTYPE)qtype, (BTLA_DTYPE)stype, asym != 0));
  return 0;
}

/* BTLAGemmQuantPackB: fp32 weight ([N][K] when is_trans, the torch layout) -> blob (size packref_size, 64-byte aligned) */
int packref_quant_pack(void* blob, const float* w, int n, int k, int ldw, int blocksize, uint32_t qtype, uint32_t stype, int asym,
                       int core, int is_trans) {
  NS_CORE_SWITCH(core, return do_pack<CT>(blob, w, n, k, ldw, blocksize, (BTLA_DTYPE)qtype, (BTLA_DTYPE)stype, asym != 0, is_trans != 0));
  return -1;
}

/* BTLAGemmPackB (integer weights): codes [K][N] (ld = ldq), scales / zps [nblk][N]; g_idx (may be NULL): group of input channel k */
int packref_pack_q(void* blob, const int8_t* q, int ldq, const float* scales, const int8_t* zps, int n, int k, int blocksize,
                   uint32_t qtype, uint32_t stype, int asym, int core, int* g_idx) {
  NS_CORE_SWITCH(core, return do_pack_q<CT>(blob, q, ldq, scales, zps, n, k, blocksize, (BTLA_DTYPE)qtype, (BTLA_DTYPE)stype, asym != 0, g_idx));
  return -1;
}

size_t packref_size_gidx(int n, int k, int blocksize, uint32_t qtype, uint32_t stype, int asym, int core) {
  NS_CORE_SWITCH(core, return do_size_gidx<CT>(n, k, blocksize, (BTLA_DTYPE)qtype, (BTLA_DTYPE)stype, asym != 0));
  return 0;
}

/* BTLAGemmUnPackB: blob -> fp32 [K][N] */
int packref_unpack(void* blob, float* out, int ld, int core, int is_float) {
  NS_CORE_SWITCH(core, return do_unpack<CT>(blob, out, ld, is_float != 0));
  return -1;
}

uint64_t packref_core_id(int core) {
  NS_CORE_SWITCH(core, return CT::ID);
  return 0;
}

}  // extern "C"
