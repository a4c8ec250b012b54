// ns_common.h — internal declarations of libns_hip.so (MI355X backend behind neural-speed's BesTLA surface).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <functional>
#include <string>

struct ns_norm_link;  // include/ns_bestla.h
struct ns_qkv_rope;

namespace ns {

// ---- BTLA_DTYPE bit encoding (reference: bestla/bestla/bestla.h:38-87) --------------------------------------
constexpr uint32_t DT_F32 = 32, DT_F16 = 16, DT_BF16 = 16 | (1u << 16), DT_S8 = 8 | (1u << 8);
constexpr uint32_t DT_S4 = 4 | (1u << 8);
constexpr uint32_t DT_F4_E2M1 = 4, DT_F4_BNB = 4 | (1u << 16), DT_F4_NF4 = 4 | (2u << 16);
__host__ __device__ inline int dt_bits(uint32_t t) { return int(t & 0xff); }
__host__ __device__ inline bool dt_is_int(uint32_t t) { return ((t >> 8) & 0xff) == 1; }
constexpr uint32_t DT_F8_E4M3 = 8, DT_F8_E5M2 = 8 | (1u << 16), DT_F8_E8M0 = 8 | (3u << 16);
constexpr uint32_t DT_DQ8_BNB = 8 | (4u << 16);  // scale dtype: u8 codes into the bitsandbytes dynamic map, double-quantised (bestla.h:72)
inline bool dt_is_f4(uint32_t t) { return t == DT_F4_E2M1 || t == DT_F4_BNB || t == DT_F4_NF4; }
__host__ __device__ inline bool dt_is_f8(uint32_t t) { return t == DT_F8_E4M3 || t == DT_F8_E5M2; }

// ---- reference GEMM cores a blob may be laid out for (bestla_defs.h:36-54, bestla_gemm.h CoreAttr :83-123) ----
struct CoreDesc {
  int ntile, packrow, ktile, comp, isa;
  uint64_t id() const {
    return uint64_t(ntile) | (uint64_t(packrow) << 8) | (uint64_t(comp) << 16) | (uint64_t(isa) << 32);
  }
};
const CoreDesc& core_desc(int ns_core);  // index = enum ns_core
int core_for_comp(int comp_type, uint32_t qtype, bool asym, size_t blocksize, int forced_core);

// ---- parsed view of a reference-format blob (bestla_storage.h:250-357, :697-859) ------------------------------
struct BlobView {
  uint64_t size = 0;
  uint32_t prologue = 0;  // 1 = WeightKBlockNInteger, 2 = WeightKBlockNFloat
  uint64_t core_id = 0;
  int npad = 0, kpad = 0, n = 0, k = 0;
  uint32_t dtype = 0;
  int blocksize = 0, dq_blocksize = 0;
  uint32_t scale_dt = 0, zp_dt = 0, red_dt = 0;
  int cstep = 0;
  uint64_t csize = 0;
  // section offsets (bytes from blob base) and sizes; 0 size = absent
  uint64_t q_off = 0, q_bytes = 0, s_off = 0, s_bytes = 0, z_off = 0, z_bytes = 0, r_off = 0, r_bytes = 0,
           shuf_off = 0, shuf_bytes = 0;
  uint64_t dq_off = 0, dq_bytes = 0;  // DQ8_BNB scales: fp32 maxima per dq block of the scale codes + the offset as the last float
  int ntile() const { return int(core_id & 0xff); }
  int packrow() const { return int((core_id >> 8) & 0xff); }
  int comp() const { return int((core_id >> 16) & 0xffff); }
  bool asym() const { return z_bytes > 0; }
  bool has_reduce() const { return r_bytes > 0; }
};
// Header access callback: (byte offset from the blob base, buffer, length, writing?)
using BlobIo = std::function<void(size_t, void*, size_t, bool)>;
// Parse the header found at `blob` (host memory) / behind an IO callback (device memory).
bool blob_parse(const void* blob, BlobView* out, std::string* err);
bool blob_parse_io(const BlobIo& io, BlobView* out, std::string* err);
// Fill sizes/offsets for a NEW blob that will be serialised at address `base_addr`: the reference aligns each
// section to 64 B in absolute-address terms (bestla_storage.h:85-95), so the layout depends on base_addr % 64.
bool blob_describe(BlobView* out, size_t n, size_t k, size_t blocksize, uint32_t qtype, uint32_t stype, bool asym,
                   int ns_core, uintptr_t base_addr, std::string* err, bool shuffle = false);
// Write every non-payload byte (sizes, pads, flags, ids) of a described blob.
void blob_write_header(const BlobView& v, void* host_base);
void blob_write_header_io(const BlobView& v, const BlobIo& io, uintptr_t base_addr);

// ---- device weight -------------------------------------------------------------------------------------------
enum WKind { WK_INT4 = 0, WK_INT8 = 1, WK_F4 = 2, WK_F8 = 3 };

}  // namespace ns

// One weight matrix resident in HBM in the MI355X streaming layout:
//   codes : [ntiles][ksteps][64 lanes][16 B]   lane = c*16 + nn  (nn = column in the 16-wide tile, c = 0..3)
//           4-bit: KSTEP = 128, dword j (0..3) of a lane holds k = s*128 + 32*j + 8*c + i, i = 0..7, nibble i at
//                  bit {0,16,4,20,8,24,12,28}[i]   (so (x & 0x000f000f) and (x & 0x00f000f0) are MFMA k-pairs)
//           8-bit: KSTEP = 64,  bytes 8*j .. 8*j+7 (j = 0..1) hold k = s*64 + 32*j + 8*c + i
//   scales: [ntiles][G][16][SPS] in the blob's scale dtype; SPS = max(1, KSTEP/blocksize), G = ksteps (SPS>1 or
//           blocksize == KSTEP) else ceil(K/blocksize)
//   zps   : int8, same shape as scales (asymmetric only)
struct ns_weight {
  int n = 0, k = 0;
  int ntiles = 0, ksteps = 0, kstep_len = 128;
  int kind = 0;        // ns::WKind
  uint32_t qtype = 0;  // BTLA_DTYPE of the codes
  int blocksize = 0;
  int sps = 1, srows = 0;  // scales per k-step per column; scale rows G
  int srow_shift = 0;      // scale row of k-step s = (SPS>1||bs==KSTEP) ? s : (s*KSTEP)/blocksize
  uint32_t scale_dt = 0;
  bool asym = false;
  uint4* codes = nullptr;
  void* scales = nullptr;
  int8_t* zps = nullptr;
  size_t codes_bytes = 0, scales_bytes = 0, zps_bytes = 0;
  // byte strides: codes per (tile, k-step) record, scales / zps per (tile, scale row).  When every k-step has its own
  // scale row the three arrays are INTERLEAVED into one stream of records {1024 B codes | scales | zps} (qstride ==
  // sstride == zstride, `scales`/`zps` point inside the `codes` allocation) so that a k-step's scales sit in the same
  // DRAM page as its codes.
  uint32_t qstride = 1024, sstride = 0, zstride = 0;
  bool interleaved = false;
  // Bit-plane formats (S1..S3, S5..S7).  The arrays above hold them WIDENED to nibbles / bytes — what the multi-row, prefill,
  // unpack and slicing code reads.  `native` (round 4) is the same weight a second time with code records of the format's own
  // width (code_rec = 256 .. 896 bytes per (tile, k-step) instead of 1024: the planes of the k-step back to back, laid out for
  // the decode kernel's in-register assembly, ns_kernels.hip repack_planes_kernel), scales / zero points repeated behind them:
  // what gemv_kernel streams, so a decode step reads the format's bits and not the container's.  Owned by this weight.
  uint32_t code_rec = 1024;  // bytes of codes per record of THIS layout
  uint8_t pl_bits = 0;       // != 0: this layout is a native clone of that bit width
  ns_weight* native = nullptr;
  // codes, scales and zps are ONE allocation: scales = codes + s_off, zps = codes + z_off
  uint32_t s_off = 0, z_off = 0;
  size_t alloc_bytes = 0;
  bool external = false;  // the arrays live in memory the caller owns (the slice a graph reserved, bestla_device_load_storage): never freed here
  bool load_pending = false;  // loaded by ns_hip_weight_load_async: ns_hip_weight_finish_load has not run yet
  bool load_failed = false;   // ns_hip_weight_finish_load rejected the blob: every forward refuses the weight
  bool single_span = false;  // the allocation is < 4 GiB, i.e. the offsets above are usable as 32-bit soffsets
  uint64_t stream_bytes = 0;  // algorithmic bytes (reference formula)
  int device = 0;
  // gemm2 (ns_gemm.hip) rounds `scale * g2_pre` to fp16 and multiplies its accumulators by g2_post = 1 / g2_pre: a
  // power of two chosen at load time from the largest |scale| so that dequantised weights stay in fp16's normal range
  float g2_pre = 1.f, g2_post = 1.f;
  // activation shuffle of GPTQ act-order blobs (ShuffleIndices, bestla_storage.h:704): int[k] on the device, own
  // allocation; the forward gathers A'[j] = A[shuf[j]] before the GEMM (prologue_a.h:322-330).  null = none.
  int* shuf = nullptr;
  _Float16 lut[16];  // f4 value table rounded to fp16: the MFMA operand (kind == WK_F4)
  float lutf[16];    // the same table in fp32: exact unpack
};

namespace ns {
// ---- launchers implemented in ns_kernels.hip ------------------------------------------------------------------
struct RepackArgs {
  const uint8_t* q;       // device: reference packed image
  const uint8_t* scales;  // device: reference scales [nblk][cstep] in scale_dt
  const int8_t* zps;      // device or null
  int ref_ntile, ref_packrow, ref_kpad, ref_npad, cstep, ref_nblk;
  uint32_t src_scale_dt = 0;  // dtype of `scales` when it differs from the device layout's (F8_E8M0 -> fp32)
  uint32_t* flags = nullptr;  // device word, OR-ed with 1 when an E5M2 code lies outside fp16's range
};
hipError_t launch_repack(const RepackArgs& a, ns_weight* w, hipStream_t st);
// DQ8_BNB scale codes [rows][cstep] -> fp32 scales of the same shape: lut[code] * dq[(row * n + col) / dq_blocksize] + dq[dq_last]
// (dq8_get_fp_scale, kernel_ref.h:1980-1992 as bestla_prologue_b.h:699-707 calls it); `lut` = 256 floats on the device
hipError_t launch_dq8_expand(const uint8_t* codes, const float* dq, const float* lut, float* out, int rows, int cstep, int n, int dq_blocksize,
                             uint32_t dq_last, hipStream_t st);
// max |scale| over a reference scale section (finite values only), as fp32 bits, atomically max-ed into *out_bits
hipError_t launch_scale_absmax(const void* scales, size_t count, uint32_t scale_dt, uint32_t* out_bits, hipStream_t st);

struct GemmSeg {
  const ns_weight* w;
  float* c;       // output base of this segment
  void* c16;      // optional fp16 shadow of the output (same shape / ldc), consumed as `a16` by the next GEMM
};
// int8-reference numerics for a decode launch (gemv_kernel XV = 3): the activations as quantize_fp_u8_colblock leaves them
struct I8Act {
  const uint8_t* aq;    // [m][ldq] u8 codes (ldq >= K)
  int ldq;
  const uint8_t* corr;  // [m][nblk] fp32 scales immediately followed by [m][nblk] u8 zero points, 16-byte aligned
  int nblk, blocksize;
  // round 5: the quantizer launch is DEFERRED — a32 / lda32 are the fp32 rows, `quantized` says whether aq / corr hold their codes yet.
  // launch_gemv quantizes inside its own launch where it can (gemv_kernel XV = 5) and answers hipErrorNotReady where it cannot: the
  // caller then runs i8_quantize_finish() and launches again (XV = 3 on the codes)
  const float* a32 = nullptr;
  int lda32 = 0, rows = 0, cols = 0;
  bool quantized = true;
};
hipError_t i8_quantize_finish(I8Act* q, hipStream_t st);  // ns_i8ref.hip
// expert-indexed decode launch (gemv_kernel XV = 4): the weight base comes from table[*id] on the device
struct MoeRoute {
  const void* table;   // device: rows {codes, scales, zps} of the group's experts (ns_moe.hip)
  const int32_t* id;   // device: &ids[token][id]
  int n_as;
};
// replayed device route (ns_route.cpp XK_QKV_ROPE; gemv_kernel only, one row): the fused QKV launch's RoPE epilogue (ns_qkv_rope) also stores k (rotated)
// and v into the reference's fp32 cache cells, and its position — RoPE angle table aside, which a captured launch of its own fills — follows the graph's counter
struct QkvRopeRoute {
  float* k32;      // the K cell of (head 0, dim 0) at the plan token's position; element strides per head / dim, elements per token
  float* v32;
  long long k32_head, k32_dim, k32_tok, v32_head, v32_dim, v32_tok;
  const int* kmove;
  int kd_pos;      // positions per token
  uint32_t* overflow;
};
struct SmallMArgs {
  const float* a;
  const void* a16;  // optional fp16 copy of A (same shape / lda): skips the fp32->fp16 staging conversion
  int lda, m;
  int ldc;
  int nseg;        // 1..3 segments laid side by side in the grid (QKV); all share K and format
  GemmSeg seg[3];
  int epilogue;    // enum ns_epilogue
  const float* d;  // epilogue operand
  int ldd;
  bool dual;       // gate/up fusion: seg[0] = W1, seg[1] = W3, out = act(A*W1) * (A*W3) -> seg[0].c ; tmp1 -> c2
  float* c2;       // optional tmp1 output in dual mode
  const ns_norm_link* link = nullptr;  // carried RMS norm (include/ns_bestla.h); gemv_kernel only
  const ns_qkv_rope* rope = nullptr;   // RoPE(q, k) + kv-cache append as the QKV epilogue; gemv_kernel only
  const QkvRopeRoute* rope_route = nullptr;  // ... of a replayed token of the device route (with `rope`)
  const I8Act* i8 = nullptr;           // int8-reference numerics (m <= 4); gemv_kernel only: launch_gemv() directly
  const MoeRoute* moe = nullptr;       // expert picked on the device (m = 1, fp32 activations); gemv_kernel only: launch_gemv() directly
};
hipError_t launch_smallm(const SmallMArgs& a, hipStream_t st);
// ns_gemm.hip: second-generation prefill GEMM; hipErrorNotSupported = use the first-generation gemm_kernel
hipError_t launch_gemm2(const SmallMArgs& a, hipStream_t st);
void gemm_scratch_release();  // frees the per-stream scratch buffers
// grow-only device scratch per (stream, slot): slot 0 = fp16 copy of A (prefill GEMM), 1 = attention partials, (4, 6, 7: ns_i8ref.hip; 10-13: ns_attn.hip host entries; 21: ns_gemvs.hip split-K tickets;)
// 2 = split-K partials, 3 = shuffled activations.  Safe under stream capture (buffers handed out while capturing are
// never freed or moved until gemm_scratch_release()); nullptr only when the allocation itself fails.
void* stream_scratch(hipStream_t st, size_t bytes, int slot);
void* stream_scratch_zeroed(hipStream_t st, size_t bytes, int slot);  // zero-filled when (re)allocated: self-resetting counters
// fp32 [m][lda] -> fp16 [m][ld16] (ld16 a multiple of 8, columns k..ld16-1 zero)
hipError_t launch_cvt_a16(const float* a, void* out16, int m, int k, int lda, int ld16, hipStream_t st);
// ns_gemv.hip: second-generation decode kernel (m <= 16): lean prologue, one 16-column tile per workgroup;
// hipErrorNotSupported = outside its envelope (use smallm_kernel)
hipError_t launch_gemv(const SmallMArgs& a, hipStream_t st);
void set_gemv_mode(int mode);  // 0 off (first-generation kernel), 1 on, -1 re-read NS_GEMV2
// ns_gemvs.hip: small-batch / narrow-output decode kernel (2 .. 16 rows; activations staged once per workgroup and shared
// by the tiles it streams, split-K across workgroups); hipErrorNotSupported = outside its envelope (use gemv_kernel)
hipError_t launch_gemvs(const SmallMArgs& a, hipStream_t st);
void set_gemvs_tuning(int what, int value);  // what: 0 mode (0 off, 1 from 2 rows, 2 from 1 row), 1 slices, 2 waves, 3 workgroups
void set_attn_tuning(int wg_target, int min_keys);
void set_attn_heads_first(int on);  // ns_attn.hip: dispatch order of the decode attention's workgroups (AttnSplitParams::heads_first)
void set_gemv_planes(int on);  // 1 (default): bit-plane formats stream their native records at decode (ns_weight::native); 0: the widened ones
void set_attn_mfma2_rows(int rows);  // query rows from which the 128-row prefill attention kernel is used (default 128; huge = never)
void set_attn_stream_tuning(int wg_target, int min_keys);  // context-range rule of the LDS-ring kernel (defaults 256 workgroups, >= 32 keys)
void set_attn_stream(int on);    // 1 (default): decode attention streams K / V through LDS rings (attn_stream_kernel); 0: attn_split_kernel
void set_attn_inlaunch(int on);  // 0 (default): attn_merge_kernel combines the context splits in a second launch; 1: the last split to finish merges inside attn_split_kernel's launch
// NS_HOST_PROFILE=1: wall time and call count of every host-tensor entry, printed at exit (diagnostics of the default,
// host-pointer route: where a token's milliseconds go between the graph executor and the GPU)
struct HostScope {
  const char* name;
  long long t0;
  explicit HostScope(const char* n);
  ~HostScope();
};
void kv_mirrors_clear();  // ns_attn.hip: drops the device mirrors of library-managed kv caches (ns_hip_cache_clear)
void set_i8_tile(int tile);     // ns_i8ref.hip: workgroup tile of i8mfma2_kernel (0 = by size)
void set_i8_mfma_gen(int gen);  // ns_i8ref.hip: 2 = i8mfma2_kernel for nibble containers (default), 1 = i8mfma_kernel everywhere
void set_gemm3_min_m(int m);  // ns_gemm.hip: rows from which gemm3_kernel is used (0 = default)
void set_gemm3_wide(int on);  // ns_gemm.hip: 1 the cross-wave output epilogue of gemm3_kernel's 1 x 4 wave tiles, 0 (default) the per-wave one, -1 = environment / default
void set_gemm3_bm(int bm);  // ns_gemm.hip: force gemm3_kernel's row-tile height (tests / A-B runs); 0 = automatic  // ns_attn.hip: context-split rule of the decode attention kernel
void set_decode_waves(int nw);  // 0 = by shape
int decode_waves(int grid, int ks, bool dual);  // waves per workgroup of a decode launch (both kernel generations)
void srow_rule(const ns_weight* w, int* num, int* den);          // scale row of k-step s = s * num / den
// the same rule as a branch-free (s * mul) >> shift, verified for every k-step; false = not expressible
bool srow_params(const ns_weight* w, int* mul, int* shift);
hipError_t launch_gemm(const SmallMArgs& a, hipStream_t st);  // large-M tiled MFMA GEMM (single segment)
bool smallm_supported(const ns_weight* w, int m);
bool smallm_dual_ok(int m);  // the fused gate/up launch handles up to 16 rows

hipError_t launch_unpack_fp32(const ns_weight* w, float* out, int ld, hipStream_t st);  // device [K][N]

struct QuantArgs {
  const float* w;  // device fp32
  size_t n, k, ld;
  bool is_trans;  // true: w is [N][K]
  int blocksize;
  uint32_t qtype, stype;
  bool asym;
  int ref_ntile, ref_packrow, ref_kpad, ref_npad, cstep;
  bool has_reduce;
  uint8_t* q_out;  // device blob sections
  uint8_t* s_out;
  int8_t* z_out;
  uint16_t* r_out;
};
hipError_t launch_quant_pack(const QuantArgs& a, hipStream_t st);
struct PackQArgs {
  const int8_t* q;      // device [K][ldq] codes
  const float* scales;  // device [nblk][N]
  const int8_t* zps;    // device [nblk][N] or null
  size_t n, k, ldq;
  int blocksize;
  uint32_t qtype, stype;
  int ref_ntile, ref_packrow, ref_kpad, ref_npad, cstep;
  bool has_reduce;
  uint8_t* q_out;
  uint8_t* s_out;
  int8_t* z_out;
  uint16_t* r_out;
};
hipError_t launch_pack_q(const PackQArgs& a, hipStream_t st);

hipError_t launch_rmsnorm(int norm_count, int norm_size, bool isrms, float eps, const float* in, float* out,
                          hipStream_t st, const float* gamma = nullptr, void* out16 = nullptr);
hipError_t launch_norm_prep(int m, int n, const float* x, int ldx, const float* gamma, void* x16, float* ssq,
                            int ssq_stride, hipStream_t st);
// Replay of the reference's device route (round 5, ns_device.hip "route"): a captured launch may take ONE moving value — a position, a
// context length, a cache address — as base + delta * (*k), k a device word the replayed graph increments once per token.  The
// launchers below that have such a value (launch_rope: n_past, launch_dup: dst, the device-layout attention: seq_all) read this
// thread-local; k == nullptr (always, outside a route capture) = the plain value.
struct Affine {
  const int* k = nullptr;
  long long delta = 0;
  long long delta2 = 0;  // launch_dup2: the second copy's destination
  long long cap = 0;     // decode attention (ns_attn.hip): the largest value the moving context length can take (the cache's n_ctx)
  int inlaunch = 1;      // decode attention: the context ranges merge inside the launch (tickets)
};
// ns_attn.hip: per-stream scratch of a moving-length decode attention, allocated before the capture that uses it
bool attn_prepare_moving(hipStream_t st, int batch, int heads, int heads_kv, int head_size, int cap);
// One function per translation unit with kernels of the hot paths: asks the runtime for one kernel's attributes, which makes it load that unit's code object
// for the device NOW (HIP loads a code object at the first launch from it: 36 ms for the tiled GEMM's, 21 ms for the decode GEMV's — otherwise paid by the first
// prompt and the first generated token).  ns_hip_warm_up() / bestla_create_device call them once.
void touch_gemm_module();
void touch_gemv_module();
void touch_attn_module();
void touch_quant_module();
extern thread_local Affine g_affine;
// ... and the device-layout attention may be asked for an fp16 copy of its output row beside the fp32 one (the carried norm of the route's
// attention-output projection needs the fp16 shadow of its activations): set around the captured call, nullptr otherwise
extern thread_local void* g_mha_out16;
// ns_api.cpp: may this weight's decode launch carry an RMS norm (ns_norm_link)?  (what link_ok() refuses, without setting an error)
bool route_link_weight_ok(const ns_weight* w);
// ns_route.cpp: one launch of the reference's device route as plain data (compared bytewise: zero-initialise before filling)
enum RouteKind : uint32_t { RK_GEMM = 1, RK_ADD, RK_MUL, RK_SILU, RK_RMSNORM, RK_ROPE, RK_ROPE_YARN, RK_DUP, RK_MHA };
struct RouteOp {
  uint32_t kind, flags;
  const void* p[4];
  long long i[24];
  float f[8];
};
bool route_hook(void* stream);             // true: the caller describes its launch in a RouteOp and hands it to route_submit
int route_submit(const RouteOp& op);
// bestla_device_sync / _memcpy: the window goes out, an evaluation ends here; src / bytes: the device-side source of the copy that follows
int route_sync_point(void* stream, const void* src = nullptr, size_t bytes = 0);
bool route_after_sync(void* stream);       // the queue has been waited for: the fp16 overflow flag is looked at (ns_route.h); true: the evaluation was run again
bool route_defer_sync(void* stream);       // a wait with only launches in flight may be left to the next copy's
void route_note_copy(void* stream);
void route_note_input(void* dst, size_t bytes, void* stream);  // a copy into device memory in front of an evaluation: its input (kept for a re-issue)
void route_time_mark(void* stream, int what);                   // NS_ROUTE_TIMING
void route_attach(void* stream);           // bestla_create_device
void route_detach(void* stream);
void route_invalidate();                   // bestla_device_free
void* route_twin_dst(void* dst, void* stream);  // bestla_device_memcpy while a plan is held (ns_route.cpp): the plan's twin of an activation, or nullptr
const void* route_translate_src(const void* src, void* stream);
hipError_t launch_rope(const float* src, float* dst, int batch, int seq, int heads, int head_size, int n_past, int n_dims,
                       int mode, float freq_base, float freq_scale, float attn_factor, hipStream_t st,
                       float ext_factor = 0.f, float corr0 = 0.f, float corr1 = 0.f, const float* lr_factor = nullptr,
                       float lr_scale = 1.f);
hipError_t launch_rope_glm(const float* src, float* dst, int batch, int seq, int heads, int head_size, int n_past, int n_dims,
                           bool skip, float freq_base, int prompt_size, const int* n_padding, hipStream_t st);
hipError_t launch_rope_cos_sin(int m, int n_past, int n_dims, float freq_base, float freq_scale, float attn_factor,
                               float* out, hipStream_t st);  // (n_past follows g_affine inside a route capture)
// ns_api.cpp: the fused QKV + RoPE + cache-write launch of a replayed token: weights by ROLE, each result at its own tensor
int qkv_rope_route_forward_m(const float* dA, const void* dA16, const ns_weight* wq, const ns_weight* wk, const ns_weight* wv, float* cq, float* ck, float* cv, int m,
                             int lda, int ldc, const ns_qkv_rope* rope, hipStream_t st);  // a window's prompt-sized form (-2: shape not taken)
extern thread_local bool g_mha_out16_written;  // the device-layout attention wrote the fp16 copy it was asked for (g_mha_out16)
int qkv_rope_route_forward(const float* dA, const void* dA16, const ns_weight* wq, const ns_weight* wk, const ns_weight* wv, float* cq, float* ck,
                           float* cv, int lda, const ns_norm_link* link, const ns_qkv_rope* rope, const QkvRopeRoute* rr, hipStream_t st);
hipError_t launch_rope_qkv_append(float* q, const float* k, const float* v, void* kc, void* vc, int seq, int heads,
                                  int heads_kv, int head_size, int n_past, int n_dims, int mode, float freq_base,
                                  float freq_scale, float attn_factor, long long c_sl, long long c_head, hipStream_t st);
hipError_t launch_aquant_u8(int row, int col, const float* src, int ld_src, uint8_t* dst, int ld_dst, float* scales,
                            int ld_scale, uint8_t* zps, int blocksize, float* blkreduce, hipStream_t st);
// GEMM-sized form (16-byte loads, dword stores; bit-identical codes), optionally with the fp16 operand of the int8-reference GEMM
hipError_t launch_aquant_u8_vec(int row, int col, const float* src, int ld_src, uint8_t* dst, int ld_dst, float* scales,
                                int ld_scale, uint8_t* zps, int blocksize, void* ap, int ld_ap, bool ap_scale16, hipStream_t st);
// A'[r][j] = A[r][idx[j]] (kernel_ref.h:28-37 shuffle_activation), fp32 [m][k] with leading dimension k
hipError_t launch_gather_cols(const float* a, int lda, const int* idx, float* out, int m, int k, hipStream_t st);
// default-policy read of [offset, offset + bytes) of the weight's stream (codes, scales, zero points) into the cache
// hierarchy; bytes is clamped to the stream, grid <= 0 picks 64 workgroups
hipError_t launch_prefetch(const ns_weight* w, size_t offset, size_t bytes, int grid, hipStream_t st);
// ns_i8ref.hip: the reference's int8-compute semantics (u8 activation quantization + integer dot per k-block)
bool i8ref_supported(const ns_weight* w);
hipError_t i8_quantize_for_decode(const float* a, int lda, const ns_weight* w, int m, hipStream_t st, bool reuse, I8Act* out);
hipError_t launch_i8ref(const float* a, int lda, const ns_weight* w, float* c, void* c16, int m, int ldc, int epilogue,
                        const float* d, int ldd, hipStream_t st, bool reuse_aq = false);
hipError_t launch_silu(const float* x, float* y, size_t n, hipStream_t st);
// rope(q), rope(k) in place on adjacent rows + the K and V cache writes of one decode position in one launch (ns_route.cpp)
hipError_t launch_rope_append(float* qk, int rows_front, int rows_k_first, int rows_k, int head_size, int n_past, int n_dims, int mode, float freq_base,
                              float freq_scale, float attn_factor, float ext_factor, float corr0, float corr1, const void* ksrc, void* kdst,
                              const long long* kne, const long long* ksnb, const long long* kdnb, const void* vsrc, void* vdst, const long long* vne,
                              const long long* vsnb, const long long* vdnb, long long kd_k, long long kd_v, hipStream_t st);
// two strided copies in one launch (the K and V cache writes of a decode step: ns_route.cpp fuses the reference's two cpy nodes)
hipError_t launch_dup2(const void* src0, void* dst0, const long long* ne0, const long long* snb0, const long long* dnb0, bool f16_0,
                       const void* src1, void* dst1, const long long* ne1, const long long* snb1, const long long* dnb1, bool f16_1, hipStream_t st);
hipError_t launch_dup(const void* src, void* dst, const long long* ne, const long long* snb, const long long* dnb, bool dst_f16,
                      hipStream_t st);
hipError_t launch_bcast_binary(int batch, int vsize, const float* t, const float* v, int vstep, float* out, bool mul,
                               hipStream_t st);

void set_error(const std::string& s);
}  // namespace ns
