// stor_shim.cpp — TEST INFRASTRUCTURE.  Thin C wrappers over the reference's OWN blob container classes
// (bestla/bestla/bestla_storage.h: StorageWeightKBlockNInteger :697-834, StorageWeightKBlockNFloat :836-859,
// PackedWeightParser :861-894), compiled from where they lie under /root/reference by `make -C oracle storref` into
// oracle/_ref/libstor_ref.so.  bestla_storage.h includes bestla_gemm.h, whose JIT half needs xbyak (not vendored); the
// container only uses CompType / CompTypeHelper / CoreAttr from its first 123 lines, so the Makefile GENERATES
// oracle/_ref/stor_inc/bestla_gemm.h from those lines (minus the bestla_jit.h include, namespaces closed) next to
// symlinks of bestla.h / bestla_utils.h / bestla_storage.h.  Nothing of the reference is copied into the repository.
// Used by tests/test_storage_pin.py to pin the oracle's restatement of the container (header bytes, section offsets,
// sizes) at every base alignment mod 64.  The product never links or loads this.
#include <cstdint>
#include <cstring>

#include "bestla_storage.h"

using namespace bestla;
using bestla::storage::gemm::StorageWeightKBlockNFloat;
using bestla::storage::gemm::StorageWeightKBlockNInteger;

namespace {
void fill_out(StorageWeightKBlockNInteger& s, const int8_t* base, uint64_t* out) {
  auto off = [&](const int8_t* p) { return p ? uint64_t(p - base) : 0; };
  out[0] = off(s.mQBuf.mBufPtr);
  out[1] = s.mQBuf.mBufSize;
  out[2] = off(s.mCorrection.mScaleBuf.mBufPtr);
  out[3] = s.mCorrection.mScaleBuf.mBufSize;
  out[4] = s.mCorrection.mZpBuf.mNotEmpty ? off(s.mCorrection.mZpBuf.mBufPtr) : 0;
  out[5] = s.mCorrection.mZpBuf.mNotEmpty ? s.mCorrection.mZpBuf.mBufSize : 0;
  out[6] = s.mCorrection.mRedBuf.mNotEmpty ? off(s.mCorrection.mRedBuf.mBufPtr) : 0;
  out[7] = s.mCorrection.mRedBuf.mNotEmpty ? s.mCorrection.mRedBuf.mBufSize : 0;
  out[8] = s.mShuffleIndices.mNotEmpty ? off(s.mShuffleIndices.mBufPtr) : 0;
  out[9] = s.mShuffleIndices.mNotEmpty ? s.mShuffleIndices.mBufSize : 0;
  out[10] = uint64_t(s.mCorrection.mCStep);
  out[11] = s.mCorrection.mCSize;
  out[12] = s.mSize;
  out[13] = uint64_t(s.mPrologueID);
  out[14] = s.mCoreId;
  out[15] = (uint64_t(uint32_t(s.mNPad)) << 32) | uint32_t(s.mKPad);
  out[16] = (uint64_t(uint32_t(s.mN)) << 32) | uint32_t(s.mK);
  out[17] = uint64_t(s.mDType);
  out[18] = (uint64_t(uint32_t(s.mBlockSize)) << 32) | uint32_t(s.mDqBlockSize);
  out[19] = uint64_t(s.mCorrection.mScaT);
  out[20] = uint64_t(s.mCorrection.mZpT);
  out[21] = uint64_t(s.mCorrection.mRedT);
}
}  // namespace

extern "C" {
// resize() + [enable_shuffle()] + assign(buf): the reference writes every non-payload byte of the blob into `buf`
// (which the caller zero-fills, at least the returned size) and locates its sections; out[0..21] as fill_out.
// buf == nullptr: size only.
uint64_t stor_assign(int is_float, uint64_t core_id, int npad, int kpad, int block, int n, int k, uint32_t qtype,
                     uint32_t stype, uint32_t redt, int asym, int shuffle, int8_t* buf, uint64_t* out) {
  if (is_float) {
    StorageWeightKBlockNFloat s(core_id);
    s.resize(npad, kpad, block, n, k, BTLA_DTYPE(qtype), BTLA_DTYPE(stype));
    if (buf) {
      s.assign(buf);
      fill_out(s, buf, out);
    }
    return s.mSize;
  }
  StorageWeightKBlockNInteger s(core_id);
  s.resize(npad, kpad, block, n, k, BTLA_DTYPE(qtype), BTLA_DTYPE(stype), BTLA_DTYPE(redt), asym != 0);
  if (shuffle) s.enable_shuffle();
  if (buf) {
    s.assign(buf);
    fill_out(s, buf, out);
  }
  return s.mSize;
}

// PackedWeightParser::deserialBuffer on a finished blob (e.g. one the oracle or the product wrote): 0 on success
int stor_deserialize(const int8_t* blob, uint64_t* out) {
  auto* w = storage::gemm::PackedWeightParser::deserialBuffer(blob);
  if (!w) return -1;
  int rc = -2;
  if (w->mPrologueID == BTLA_PROLOGUEB_IDS::WeightKBlockNInteger ||
      w->mPrologueID == BTLA_PROLOGUEB_IDS::WeightKBlockNFloat) {
    fill_out(*static_cast<StorageWeightKBlockNInteger*>(w), blob, out);
    rc = 0;
  }
  delete w;
  return rc;
}
}
