// ns_blob.cpp — the reference's serialized packed-weight container ("blob"), host side of libns_hip.so.
//
// Format owner: /root/reference/bestla/bestla/bestla_storage.h
//   ISerializable::mSize                         :41-56
//   IWeightBase  (prologue id, core id, dims)    :250-317
//   IWeightKBlockBase (block sizes)              :319-357
//   ObjectAlignedBuffer<64> / OptionalBuffer     :59-147   (u64 size, u64 offset, pad to 64 B ABSOLUTE, data)
//   ObjectQuantCorrection (scales, zp, reduce)   :151-248
//   StorageWeightKBlockNInteger / ...NFloat      :697-859
// Only the header (a few dozen bytes + per-section size/offset words) is handled on the host; every payload
// byte is produced and consumed by HIP kernels (ns_kernels.hip).
#include <cstring>

#include "ns_common.h"

namespace ns {

namespace {
// CompType values: bestla_gemm.h:22-50.  ISA values: bestla.h:23-37.
constexpr int kCompFp32 = 0x000, kCompBf16 = 0x011, kCompFp16 = 0x022, kCompU8S8Fp32 = 0x034;
const CoreDesc kCoreTable[9] = {
    {24, 1, 1, kCompFp32, 2},       // NS_CORE_AVX2            SCoreRowNAvx2<24,4>
    {48, 1, 1, kCompFp32, 4},       // NS_CORE_AVX512F         SCoreRowNAvx512f<48,8>
    {48, 2, 32, kCompBf16, 9},      // NS_CORE_AMX_BF16        HCoreRowNAmxbf16<48,16>
    {48, 2, 32, kCompFp16, 11},     // NS_CORE_AMX_FP16        HCoreRowNAmxfp16<48,16>
    {48, 4, 4, kCompU8S8Fp32, 6},   // NS_CORE_AVX512_VNNI_KB  ICoreRowNAvx512vnniKBlock<48,4>
    {48, 4, 4, kCompU8S8Fp32, 5},   // NS_CORE_AVX512BW_KB     ICoreRowNAvx512bwKBlock<48,8>
    {24, 4, 4, kCompU8S8Fp32, 3},   // NS_CORE_AVX_VNNI_KB     ICoreRowNAvxvnniKBlock<24,2>
    {24, 4, 4, kCompU8S8Fp32, 2},   // NS_CORE_AVX2_VNNI_KB    ICoreRowNAvx2vnniKBlock<24,2>
    {48, 4, 64, kCompU8S8Fp32, 10}, // NS_CORE_AMX_INT8_KB     ICoreRowNAmxint8KBlock<48,16>
};

inline size_t ceil_div(size_t a, size_t b) { return (a + b - 1) / b; }
inline size_t round_up(size_t a, size_t b) { return ceil_div(a, b) * b; }

// A cursor that either reads the header words out of a blob or lays a new header down.  The words are scattered
// between the payload sections, so access goes through a callback: plain memcpy for host blobs, small
// hipMemcpy's for blobs that live in device memory (ns_api.cpp).
class HeaderIO {
 public:
  HeaderIO(const BlobIo& io, uintptr_t base_addr, bool writing) : io_(io), pos_(0), addr_(base_addr), w_(writing) {}
  // Reading: nothing is fetched at or beyond `limit` (the blob's own serialized size once its first word is known),
  // so a corrupt section size cannot send the walk outside the buffer; ok() turns false instead.
  void set_limit(uint64_t limit) { limit_ = limit; }
  bool ok() const { return ok_; }
  template <typename T>
  void word(T& v) {
    if (!w_ && (!ok_ || pos_ > limit_ || limit_ - pos_ < sizeof(T))) {
      ok_ = false;
      v = T();
      return;
    }
    if (io_) io_(pos_, &v, sizeof(T), w_);
    pos_ += sizeof(T);
  }
  // {u64 bytes, u64 pad, pad bytes, payload}: pad makes the payload start a multiple of 64 in absolute address
  void section(uint64_t& bytes, uint64_t& off) {
    word(bytes);
    uint64_t pad = 0;
    if (w_) {
      uintptr_t after = addr_ + pos_ + 8;
      pad = round_up(after, 64) - after;
    }
    word(pad);
    if (!w_ && (!ok_ || pad > limit_ - pos_ || bytes > limit_ - pos_ - pad)) {
      ok_ = false;
      bytes = 0, off = 0;
      return;
    }
    pos_ += pad;
    off = pos_;
    pos_ += bytes;
  }
  void optional_section(uint64_t& bytes, uint64_t& off) {
    uint8_t present = bytes != 0;
    word(present);
    if (present) {
      section(bytes, off);
    } else {
      bytes = 0;
      off = 0;
    }
  }
  size_t pos() const { return pos_; }

 private:
  const BlobIo& io_;
  size_t pos_;
  uintptr_t addr_;
  bool w_;
  uint64_t limit_ = ~uint64_t(0);
  bool ok_ = true;
};

bool traverse(HeaderIO& io, BlobView& v, std::string* err) {
  io.word(v.size);
  io.set_limit(v.size);
  io.word(v.prologue);
  io.word(v.core_id);
  io.word(v.npad);
  io.word(v.kpad);
  io.word(v.n);
  io.word(v.k);
  io.word(v.dtype);
  io.word(v.blocksize);
  io.word(v.dq_blocksize);
  if (v.prologue != 1 && v.prologue != 2) {
    if (err) *err = "blob: prologue id is not a k-block weight (WeightKBlockNInteger/NFloat)";
    return false;
  }
  io.section(v.q_bytes, v.q_off);
  io.word(v.scale_dt);
  io.word(v.zp_dt);
  io.word(v.red_dt);
  io.word(v.cstep);
  io.word(v.csize);
  io.section(v.s_bytes, v.s_off);
  io.optional_section(v.z_bytes, v.z_off);
  io.optional_section(v.r_bytes, v.r_off);
  io.optional_section(v.dq_bytes, v.dq_off);
  if ((v.dq_bytes != 0) != (v.scale_dt == DT_DQ8_BNB) || (v.dq_bytes && (v.dq_blocksize <= 0 || v.dq_bytes % 4 != 0 || v.dq_bytes < 8))) {
    if (err) *err = "blob: DQ8_BNB scales and the double-quantisation section do not go together";
    return false;
  }
  io.optional_section(v.shuf_bytes, v.shuf_off);
  if (!io.ok()) {
    if (err) *err = "blob: a header word or section lies beyond the serialized size";
    return false;
  }
  return true;
}
}  // namespace

const CoreDesc& core_desc(int c) { return kCoreTable[(c < 0 || c > 8) ? 1 : c]; }

// Which reference core a NEW blob is laid out for.  The reference walks CPUID from the requested compute type
// downwards (bestla_gemm.cpp:241-300).  NS_CORE_AUTO reproduces the walk for a Sapphire-Rapids class host
// (AMX_INT8 + AMX_BF16 + AVX512_VNNI, no AMX_FP16) — the reference's own deployment target.
int core_for_comp(int comp_type, uint32_t qtype, bool asym, size_t blocksize, int forced_core) {
  if (forced_core >= 0 && forced_core <= 8) return forced_core;
  const bool is_int = dt_is_int(qtype);
  switch (comp_type) {
    case 4:  // NE_COMP_INT8: integer weights only, and not asymmetric S8 (bestla_gemm.cpp:250)
      if (is_int && !(qtype == DT_S8 && asym)) {
        if (blocksize % 64 == 0) return 8;  // tAMX_INT8_US_KBlock, KTILE 64
        if (blocksize % 4 == 0) return 4;   // tAVX512_VNNI_KBlock, KTILE 4
      }
      [[fallthrough]];
    case 2:  // NE_COMP_BF16
      if (blocksize % 32 == 0) return 2;  // tAMX_BF16, KTILE 32
      [[fallthrough]];
    default:     // NE_COMP_F16 (no AMX_FP16 on SPR), NE_COMP_F32, NE_COMP_UNDEF
      return 1;  // tAVX512F, KTILE 1
  }
}

size_t code_bytes(size_t elts, uint32_t qtype);

// A header that parsed is not yet a blob that can be trusted: the loaders launch kernels over the sections, so every
// section must be at least as large as the geometry in the header implies (a truncated or corrupt model file would
// otherwise make the repack read out of bounds in HBM — silent garbage weights or a GPU fault), and all of it must lie
// inside the serialized size.
static bool check_view(const BlobView* out, std::string* err) {
  auto bad = [&](const char* why) {
    if (err) *err = std::string("blob: ") + why;
    return false;
  };
  if (out->n <= 0 || out->k <= 0 || out->npad < out->n || out->kpad < out->k || out->blocksize <= 0 ||
      out->ntile() <= 0 || out->packrow() <= 0 || out->npad % out->ntile() || out->kpad % out->packrow())
    return bad("inconsistent header");
  const int sbits = dt_bits(out->scale_dt);
  if (sbits != 8 && sbits != 16 && sbits != 32) return bad("unknown scale dtype");
  if (out->cstep < out->n) return bad("correction step smaller than N");
  const uint64_t rows = ceil_div(size_t(out->kpad), size_t(out->blocksize));
  if (out->csize < rows * uint64_t(out->cstep)) return bad("correction size smaller than (k-blocks x step)");
  if (out->q_bytes < code_bytes(size_t(out->npad) * out->kpad, out->dtype)) return bad("code section smaller than NPad x KPad codes");
  if (out->s_bytes < out->csize * uint64_t(sbits / 8)) return bad("scale section smaller than the correction size");
  if (out->z_bytes && out->z_bytes < out->csize) return bad("zero-point section smaller than the correction size");
  if (out->r_bytes && out->r_bytes < out->csize * 2) return bad("reduce section smaller than the correction size");
  if (out->shuf_bytes && out->shuf_bytes < uint64_t(out->k) * sizeof(int)) return bad("shuffle section smaller than K indices");
  const uint64_t ends[5] = {out->q_off + out->q_bytes, out->s_off + out->s_bytes, out->z_bytes ? out->z_off + out->z_bytes : 0,
                            out->r_bytes ? out->r_off + out->r_bytes : 0, out->shuf_bytes ? out->shuf_off + out->shuf_bytes : 0};
  for (uint64_t e : ends)
    if (e > out->size) return bad("a section ends beyond the serialized size");
  return true;
}

bool blob_parse_io(const BlobIo& io, BlobView* out, std::string* err) {
  HeaderIO h(io, 0, false);
  *out = BlobView();
  if (!traverse(h, *out, err)) return false;
  return check_view(out, err);
}

bool blob_parse(const void* blob, BlobView* out, std::string* err) {
  if (!blob) {
    if (err) *err = "blob: null pointer";
    return false;
  }
  const uint8_t* base = static_cast<const uint8_t*>(blob);
  BlobIo io = [base](size_t off, void* buf, size_t n, bool) { memcpy(buf, base + off, n); };
  return blob_parse_io(io, out, err);
}

size_t code_bytes(size_t elts, uint32_t qtype) {  // bestla_storage.h:729-745
  const int b = dt_bits(qtype);
  if (!dt_is_int(qtype) || b == 1 || b == 2 || b == 4 || b == 8) return ceil_div(elts * b, 8);
  size_t total = 0;
  for (int plane : {4, 2, 1})
    if (b & plane) total += ceil_div(elts * plane, 8);
  return total;
}

bool blob_describe(BlobView* out, size_t n, size_t k, size_t blocksize, uint32_t qtype, uint32_t stype, bool asym,
                   int core, uintptr_t base_addr, std::string* err, bool shuffle) {
  BlobView v;
  const bool is_int = dt_is_int(qtype);
  if (!is_int && !dt_is_f4(qtype) && !dt_is_f8(qtype)) {
    if (err) *err = "pack: unsupported weight dtype";
    return false;
  }
  if (dt_is_f8(qtype)) {  // quantize_f32_f8_rowblock_mxscale handles E8M0 and F32 scales only (kernel_ref.h:1775-1789)
    if (stype != DT_F8_E8M0 && stype != DT_F32) {
      if (err) *err = "pack: fp8 weights take F8_E8M0 or F32 scales";
      return false;
    }
  } else if (stype != DT_F32 && stype != DT_BF16 && stype != DT_F16) {
    if (err) *err = "pack: unsupported scale dtype";
    return false;
  }
  const CoreDesc& cd = core_desc(core);
  v.prologue = is_int ? 1 : 2;
  v.core_id = cd.id();
  v.kpad = int(round_up(k, cd.ktile));  // createStorage, bestla_prologue_b.h:120-127
  v.npad = int(round_up(n, cd.ntile));
  v.n = int(n);
  v.k = int(k);
  v.dtype = qtype;
  v.blocksize = (int64_t(blocksize) <= 0) ? v.kpad : int(blocksize);
  v.dq_blocksize = 0;
  v.q_bytes = code_bytes(size_t(v.npad) * v.kpad, qtype);
  const size_t rows = ceil_div(v.kpad, v.blocksize);
  v.scale_dt = stype;
  v.cstep = v.npad;
  v.csize = rows * v.npad;
  v.s_bytes = v.csize * (dt_bits(stype) / 8);
  if (is_int) {
    v.zp_dt = DT_S8;
    v.red_dt = DT_BF16;  // every caller passes BF16 (bestla_gemm.cpp:229,:308,:408)
    v.z_bytes = asym ? v.csize : 0;
    const int btype = (cd.comp >> 4) & 0xf;  // integer compute cores carry the per-block reduce (storage.h:747-749)
    v.r_bytes = (btype == 3 || btype == 4) ? v.csize * 2 : 0;
    if (shuffle) v.shuf_bytes = uint64_t(k) * sizeof(int);  // enable_shuffle, bestla_storage.h:761-765
  }
  // serialized size = sum of every object's getSerializedSize(): each present section reserves 16 + bytes + 64
  auto sec = [](uint64_t b) { return size_t(16 + b + 64); };
  auto opt = [&](uint64_t b) { return size_t(1 + (b ? sec(b) : 0)); };
  size_t total = (8 + 4 + 8 + 16 + 4) + 8 + sec(v.q_bytes) + (12 + 4 + 8) + sec(v.s_bytes) + opt(v.z_bytes) +
                 opt(v.r_bytes) + opt(0);
  if (is_int) total += opt(v.shuf_bytes);  // NFloat::resize recomputes mSize without the shuffle flag (:853-856)
  v.size = round_up(total, 64);
  BlobIo none;
  HeaderIO io(none, base_addr, true);
  BlobView tmp = v;
  if (!traverse(io, tmp, err)) return false;
  v.q_off = tmp.q_off;
  v.s_off = tmp.s_off;
  v.z_off = tmp.z_off;
  v.r_off = tmp.r_off;
  v.shuf_off = tmp.shuf_off;
  *out = v;
  return true;
}

void blob_write_header_io(const BlobView& v, const BlobIo& io, uintptr_t base_addr) {
  BlobView tmp = v;
  HeaderIO h(io, base_addr, true);
  traverse(h, tmp, nullptr);
}

void blob_write_header(const BlobView& v, void* base) {
  uint8_t* b = static_cast<uint8_t*>(base);
  BlobIo io = [b](size_t off, void* buf, size_t n, bool) { memcpy(b + off, buf, n); };
  blob_write_header_io(v, io, reinterpret_cast<uintptr_t>(base));
}

}  // namespace ns
