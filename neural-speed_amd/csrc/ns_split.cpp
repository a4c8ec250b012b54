// ns_split.cpp — exact tensor-parallel shards of reference-format blobs ON THE HOST, before anything is uploaded.
//
// The reference cuts a BTLA weight for a rank by dequantising the whole tensor to fp32, taking the rank's rows / columns and
// quantising them again (model_load_tensor / bestla_split_weight, models/model_utils/model_files.h:1538-1563: bestla_unpackweight_fp32
// -> bestla_packweight_copyattr; the split rules are :145-190).  Quantisation is per (k-block, column), so whenever the cut falls on
// block boundaries the re-quantised shard holds exactly the codes, scales, zero points and block sums the full tensor held there:
// this file copies them — no fp32 round trip, no upload of what the rank drops, bit-identical shards (the re-quantising route stays
// available through the two surface functions above for cuts inside a block).  Host memory in, host memory out; what a rank uploads
// afterwards (ns_hip_weight_from_blob / bestla_device_load_storage) is its shard only.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ns_bestla.h"
#include "ns_common.h"

namespace ns {
namespace {

// element (k, n) of the reference's interleaved image [N/NTILE][KPad/PACK][NTILE][PACK] (padding_interleave, kernel_ref.h:39-57)
inline size_t tiled_index(int k, int n, int ntile, int packrow, int kpad) {
  return size_t(n / ntile) * ntile * kpad + size_t(k / packrow) * ntile * packrow + size_t(n % ntile) * packrow + (k % packrow);
}
// stored code of element e of a bit-plane image (compress_*, kernel_ref.h:155-365; plane offsets bestla_prologue_b.h:512-547)
inline int read_code(const uint8_t* img, size_t e, size_t elts, int bits) {
  if (bits == 8) return img[e];
  const uint8_t* p = img;
  int v = 0, sh = 0;
  if (bits & 4) {
    v |= (p[e >> 1] >> ((e & 1) * 4)) & 0xf;
    p += elts / 2;
    sh = 4;
  }
  if (bits & 2) {
    v |= ((p[e >> 2] >> ((e & 3) * 2)) & 0x3) << sh;
    p += elts / 4;
    sh += 2;
  }
  if (bits & 1) v |= ((p[e >> 3] >> (e & 7)) & 0x1) << sh;
  return v;
}
inline void write_code(uint8_t* img, size_t e, size_t elts, int bits, int v) {
  if (bits == 8) {
    img[e] = uint8_t(v);
    return;
  }
  uint8_t* p = img;
  int sh = 0;
  if (bits & 4) {
    const int s = int(e & 1) * 4;
    p[e >> 1] = uint8_t((p[e >> 1] & ~(0xf << s)) | ((v & 0xf) << s));
    p += elts / 2;
    sh = 4;
  }
  if (bits & 2) {
    const int s = int(e & 3) * 2;
    p[e >> 2] = uint8_t((p[e >> 2] & ~(0x3 << s)) | (((v >> sh) & 0x3) << s));
    p += elts / 4;
    sh += 2;
  }
  if (bits & 1) {
    const int s = int(e & 7);
    p[e >> 3] = uint8_t((p[e >> 3] & ~(1 << s)) | (((v >> sh) & 1) << s));
  }
}
// the image as the packer leaves it where the matrix has no element (zero-padded int8 codes, then compressed): an integer type
// of b < 8 bits stores code + 2^(b-1) (compress_s8_s4 and the plane kernels), everything else stores the zero byte
void fill_padding(uint8_t* img, size_t elts, uint32_t qtype, size_t q_bytes) {
  const int bits = dt_bits(qtype);
  if (!dt_is_int(qtype) || bits == 8) {
    memset(img, 0, q_bytes);
    return;
  }
  const int full = 1 << (bits - 1);
  uint8_t* p = img;
  int sh = 0;
  if (bits & 4) {
    const int nib = full & 0xf;
    memset(p, nib | (nib << 4), elts / 2);
    p += elts / 2;
    sh = 4;
  }
  if (bits & 2) {
    memset(p, ((full >> sh) & 3) * 0x55, elts / 4);
    p += elts / 4;
    sh += 2;
  }
  if (bits & 1) memset(p, ((full >> sh) & 1) ? 0xff : 0, (elts + 7) / 8);
}

int core_of(uint64_t core_id) {
  for (int c = 0; c <= 8; c++)
    if (core_desc(c).id() == core_id) return c;
  return -1;
}

bool describe_shard(const BlobView& v, int n, int k, uintptr_t base_addr, bool keep_shuffle, BlobView* out, std::string* err) {
  const int core = core_of(v.core_id);
  if (core < 0) {
    *err = "split: the blob's GEMM core is not one of the reference's";
    return false;
  }
  // a per-channel blob stores blocksize = KPad (bestla_prologue_b.h:120-127): the shard is per-channel again
  const size_t bs = v.blocksize >= v.kpad ? size_t(-1) : size_t(v.blocksize);
  return blob_describe(out, size_t(n), size_t(k), bs, v.dtype, v.scale_dt, v.asym(), core, base_addr, err, keep_shuffle);
}

}  // namespace
}  // namespace ns

extern "C" {

int ns_blob_shape(const void* blob, int* n, int* k) {
  ns::BlobView v;
  std::string err;
  if (!blob || !ns::blob_parse(blob, &v, &err)) {
    ns::set_error(err.empty() ? "blob: null pointer" : err);
    return -1;
  }
  if (n) *n = v.n;
  if (k) *k = v.k;
  return 0;
}

unsigned long long ns_bestla_split_weight_size(const void* src_blob, int dst_n, int dst_k) {
  ns::BlobView v, d;
  std::string err;
  if (!src_blob || !ns::blob_parse(src_blob, &v, &err) || dst_n < 1 || dst_k < 1 || dst_n > v.n || dst_k > v.k ||
      !ns::describe_shard(v, dst_n, dst_k, 0, v.shuf_bytes != 0 && dst_k == v.k, &d, &err)) {
    ns::set_error(err.empty() ? "split: invalid argument" : err);
    return 0;
  }
  return d.size + 64;  // + the slack a base that is not 64-byte aligned may need (sections are aligned in absolute terms)
}

int ns_bestla_split_weight(const void* src_blob, void* dst_blob, unsigned long long dst_capacity, int n0, int n1, int k0, int k1) {
  using namespace ns;
  BlobView v, d;
  std::string err;
  if (!src_blob || !dst_blob || !blob_parse(src_blob, &v, &err)) {
    set_error(err.empty() ? "split: null argument" : err);
    return -1;
  }
  if (n0 < 0 || n1 > v.n || n0 >= n1 || k0 < 0 || k1 > v.k || k0 >= k1) {
    set_error("split: range outside the weight");
    return -1;
  }
  if (v.scale_dt == DT_DQ8_BNB) {  // dq blocks run across rows and columns of the scale array: no exact cut
    set_error("split: a blob with DQ8_BNB scales needs the re-quantising route");
    return -2;
  }
  const int n = n1 - n0, k = k1 - k0;
  const bool per_channel = v.blocksize >= v.kpad;
  if (!per_channel && (k0 % v.blocksize != 0 || (k1 != v.k && k1 % v.blocksize != 0))) {
    set_error("split: a K cut inside a quantisation block needs the re-quantising route (bestla_unpackweight_fp32 + bestla_packweight_copyattr)");
    return -2;
  }
  if (per_channel && v.has_reduce() && k != v.k) {
    set_error("split: a K cut of a per-channel blob with block sums needs the re-quantising route");
    return -2;
  }
  if (v.shuf_bytes && k != v.k) {
    set_error("split: an activation-order (g_idx) blob cannot be cut along K");
    return -2;
  }
  if (!describe_shard(v, n, k, reinterpret_cast<uintptr_t>(dst_blob), v.shuf_bytes != 0, &d, &err)) {
    set_error(err);
    return -1;
  }
  if (d.size > dst_capacity) {
    set_error("split: destination too small (" + std::to_string(d.size) + " bytes needed)");
    return -1;
  }
  const uint8_t* sb = static_cast<const uint8_t*>(src_blob);
  uint8_t* db = static_cast<uint8_t*>(dst_blob);
  memset(db, 0, d.size);
  blob_write_header(d, db);
  const int bits = dt_bits(v.dtype);
  const int ntile = v.ntile(), pack = v.packrow();
  const size_t selts = size_t(v.npad) * v.kpad, delts = size_t(d.npad) * d.kpad;
  fill_padding(db + d.q_off, delts, v.dtype, d.q_bytes);
  const uint8_t* simg = sb + v.q_off;
  uint8_t* dimg = db + d.q_off;
  // ---- codes: destination tile by tile (a tile's bytes are written by one thread only, planes included: element ranges of
  //      different tiles never share a byte because ntile * kpad is a multiple of 8) ----
  const int dtiles = d.npad / ntile;
  const bool whole_tiles = n0 % ntile == 0 && k0 % pack == 0 && (bits == 8 || bits == 4);
  auto do_tiles = [&](int t0, int t1) {
    for (int t = t0; t < t1; t++) {
      const int cols = std::min(ntile, n - t * ntile);
      if (whole_tiles && cols == ntile && k % pack == 0) {
        // [k/pack][ntile][pack] rows of the destination tile are contiguous runs of the source tile
        const size_t s0 = tiled_index(k0, n0 + t * ntile, ntile, pack, v.kpad), d0 = tiled_index(0, t * ntile, ntile, pack, d.kpad);
        const size_t run = size_t(k / pack) * ntile * pack;
        if (bits == 8) memcpy(dimg + d0, simg + s0, run);
        else memcpy(dimg + d0 / 2, simg + s0 / 2, run / 2);  // both even: ntile * pack is
        continue;
      }
      for (int kk = 0; kk < k; kk++)
        for (int c = 0; c < cols; c++)
          write_code(dimg, tiled_index(kk, t * ntile + c, ntile, pack, d.kpad), delts, bits,
                     read_code(simg, tiled_index(k0 + kk, n0 + t * ntile + c, ntile, pack, v.kpad), selts, bits));
    }
  };
  {
    const int hw = int(std::max(1u, std::min(16u, std::thread::hardware_concurrency())));
    const int nthr = std::max(1, std::min(hw, dtiles / 4));
    std::vector<std::thread> th;
    for (int i = 1; i < nthr; i++) th.emplace_back(do_tiles, int(int64_t(dtiles) * i / nthr), int(int64_t(dtiles) * (i + 1) / nthr));
    do_tiles(0, int(int64_t(dtiles) / nthr));
    for (auto& t : th) t.join();
  }
  // ---- scales, zero points, block sums: [nblk][cstep] rows of the rank's blocks, the rank's columns ----
  const int kb0 = per_channel ? 0 : k0 / v.blocksize;
  const int rows = per_channel ? 1 : (k + v.blocksize - 1) / v.blocksize;
  auto copy_rows = [&](uint64_t soff, uint64_t doff, size_t esz) {
    for (int r = 0; r < rows; r++)
      memcpy(db + doff + (size_t(r) * d.cstep) * esz, sb + soff + (size_t(kb0 + r) * v.cstep + n0) * esz, size_t(n) * esz);
  };
  copy_rows(v.s_off, d.s_off, dt_bits(v.scale_dt) / 8);
  if (v.asym() && d.z_bytes) copy_rows(v.z_off, d.z_off, 1);
  if (v.has_reduce() && d.r_bytes) copy_rows(v.r_off, d.r_off, 2);
  if (v.shuf_bytes && d.shuf_bytes) memcpy(db + d.shuf_off, sb + v.shuf_off, size_t(v.k) * sizeof(int));
  return 0;
}

}  // extern "C"
