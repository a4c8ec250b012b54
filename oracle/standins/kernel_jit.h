/*
 * TEST INFRASTRUCTURE — stand-in for bestla/bestla/kernel_jit.h when the reference's kernel / packer headers are compiled
 * for the oracle's pins (oracle/Makefile targets avxref, packref).  The real header defines BesTLA's JIT kernels on top of
 * xbyak, a third-party dependency that is not vendored in the reference tree.  The headers that include it name exactly three
 * of its classes; these are NOT implementations of them:
 *   JitMemcpy2DAvx512f / JitMemcpy2DAvx2 :: forward / forward1 — report NotSupport, upon which kernel_wrapper.h:136-190 falls
 *       back to the reference's own scalar memcpy2d / memcpy2d_withop;
 *   DecompressS3 :: forward_avx512f / forward_avx2 — the vector 3-bit plane decompressor (kernel_avx512f.h:783,
 *       kernel_avx2.h:3422); trapping: no pinned path reaches it (the packer compresses; 3-bit unpack runs the scalar kernel
 *       under NS_PACKREF_ISA=nosimd).
 */
#pragma once
#include <immintrin.h>

#include "bestla.h"

namespace bestla {
namespace kernel {
namespace jit {

struct DecompressS3 {
  template <typename... A>
  static void forward_avx512f(A...) {
    __builtin_trap();
  }
  template <typename... A>
  static void forward_avx2(A...) {
    __builtin_trap();
  }
};

struct JitMemcpy2DAvx512f {
  template <typename S, typename D>
  static BTLA_CODE forward(const S*, D*, int, int, int, int, void* = nullptr) {
    return BTLA_CODE::NotSupport;
  }
  template <typename S, typename D, BTLA_ELTWISEOP OP>
  static BTLA_CODE forward1(const S*, D*, int, int, int, int, void* = nullptr) {
    return BTLA_CODE::NotSupport;
  }
};
struct JitMemcpy2DAvx2 : JitMemcpy2DAvx512f {};

}  // namespace jit
}  // namespace kernel
}  // namespace bestla
