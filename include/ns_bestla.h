/*
 * ns_bestla.h — C ABI of libns_hip.so, the MI355X (gfx950) backend that drops in behind neural-speed's
 * BesTLA operator surface.  Plain pointers and sizes only; no torch / HIP types in the signatures.
 *
 * Part 1 mirrors, name for name and argument for argument, the extern "C" surface of
 *   /root/reference/neural_speed/core/ne_bestla.h:21-83
 * so that a neural-speed build can link this library instead of its bestla layer.
 * Pointers in part 1 are HOST pointers exactly as in the reference; calls are synchronous.
 *
 * Part 2 is the quantizer/packer side, C-callable twins of the C++ functions in
 *   /root/reference/neural_speed/core/layers/bestla_gemm.h:38-55
 *
 * Part 3 is the device-resident API modelled on the reference's own device backend precedent
 *   /root/reference/neural_speed/core/ne_bestla.h:85-112 (bestla_device_*):
 * weights are loaded once into HBM in the MI355X layout, activations/outputs are DEVICE pointers and the
 * `stream` argument is a hipStream_t passed as void*.
 *
 * Error behaviour follows the reference: forwards print "Err: invalid parameters" and
 * return (release-build behaviour of inner_product.cpp:31-35); *_support() returning false is the only
 * graceful refusal; sizes return 0 on failure; part-3 functions additionally return an int status
 * (0 = ok) because they have no reference counterpart to stay silent for.
 * If no HIP device is present every compute entry fails loudly (prints + returns error); there is NO CPU
 * fallback in this library.
 */
#ifndef NS_BESTLA_H
#define NS_BESTLA_H
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------------
 * Part 1 — reference operator surface (ne_bestla.h).  Host pointers, synchronous.
 * ---------------------------------------------------------------------------------------------- */
void bestla_init(void);                    /* ne_bestla.h:33  */
void bestla_timer(bool _init);             /* ne_bestla.h:23  */
int bestla_set_threads(int _nth);          /* ne_bestla.h:25  (no host thread pool here: returns _nth) */
void* bestla_get_thread_handle(void);      /* ne_bestla.h:27  (returns the backend context) */

/* ne_bestla.h:35 / inner_product.cpp:20-25 */
unsigned long long bestla_f32f32_get_workspace_size(int _m, int _n, int _k, void* wptr);
/* ne_bestla.h:37-38 / inner_product.cpp:28-36:  output[m][n] = activation[m][k] * W[k][n] */
void bestla_f32f32_forward(float* activation, void* weiptr, float* output, int _m, int _n, int _k, int lda, int ldo,
                           void* workspace);

/* ne_bestla.h:40-42 / inner_product.cpp:113-244 */
bool bestla_fusion_add_f32f32_support(void* weiptr, int _m, int _n, int _k);
void bestla_fusion_add_f32f32_forward(float* activation, void* weiptr, float* bias, float* output, int _m, int _n,
                                      int _k, int lda, int ldo, bool boardcast_bias, void* workspace);

/* ne_bestla.h:44-51 / ip_fusion_qkv.cpp:155-307: output = {Q | K | V}, each [m][ldo], stacked along dim 0 */
unsigned long long bestla_fusion_QKV_f32f32_get_workspace_size(int _m, int _n, int _k, void* w1ptr);
bool bestla_fusion_QKV_f32f32_support(void* wqptr, void* wkptr, void* wvptr, int _m, int _n, int _k);
void bestla_fusion_QKV_f32f32_forward(float* activation, void* wqptr, void* wkptr, void* wvptr, float* output, int _m,
                                      int _n, int _k, int lda, int ldo, void* workspace);

/* ne_bestla.h:53-81 / ip_fusion_ffn.cpp:20-29,724-779 */
unsigned long long bestla_fusion_FFN_f32f32_get_workspace_size(int seq, int fin, int fmid, int fout, void* w1ptr,
                                                               void* w2ptr);
bool bestla_fusion_FFN_Gelu_Mul_f32f32_support(void* w1ptr, void* w2ptr, void* w3ptr, int seq, int fin, int fmid,
                                               int fout);
void bestla_fusion_FFN_Gelu_Mul_f32f32_forward(float* activation, void* w1ptr, void* w2ptr, void* w3ptr, float* tmp1,
                                               float* tmp2, float* output, int seq, int fin, int fmid, int fout,
                                               void* workspace);
bool bestla_fusion_FFN_SiLu_f32f32_support(void* w1ptr, void* w2ptr, void* w3ptr, int seq, int fin, int fmid, int fout);
void bestla_fusion_FFN_SiLu_f32f32_forward(float* activation, void* w1ptr, void* w2ptr, void* w3ptr, float* tmp1,
                                           float* tmp2, float* output, int seq, int fin, int fmid, int fout,
                                           void* workspace);
bool bestla_fusion_FFN_GeLu_f32f32_support(void* w1ptr, void* w2ptr, int seq, int fin, int fmid, int fout);
void bestla_fusion_FFN_GeLu_f32f32_forward(float* activation, void* w1ptr, void* w2ptr, float* tmp1, float* output,
                                           int seq, int fin, int fmid, int fout, void* workspace);
bool bestla_fusion_FFN_Add_GeLu_f32f32_support(void* w1ptr, void* w2ptr, int seq, int fin, int fmid, int fout);
void bestla_fusion_FFN_Add_GeLu_f32f32_forward(float* activation, void* w1ptr, void* w2ptr, float* b1ptr, float* b2ptr,
                                               float* tmp1, float* output, int seq, int fin, int fmid, int fout,
                                               bool boardcast_bias, void* workspace);

/* ne_bestla.h:67-69 / ne_bestla.cpp:74-111 */
void bestla_unpackweight_fp32(void* wptr, int n, int k, float* fp32data, int ld);
void bestla_packweight_copyattr(const float* f32ptr, void* dstpr, int n, int k, int ld, void* srcptr);

/* ne_bestla.h:71-75 / ne_bestla.cpp:113-164 */
void bestla_layernormalization(int norm_count, int norm_size, bool isrms, float epsilon, const float* FpIn,
                               float* FpOut);
void bestla_mul(int batch, int vsize, const float* tensor, const float* vector, int vstep, float* out);
void bestla_add(int batch, int vsize, const float* tensor, const float* vector, int vstep, float* out);

/* ------------------------------------------------------------------------------------------------
 * Part 2 — quantize / pack (bestla_gemm.h:38-55).  dtypes are BTLA_DTYPE values (bestla.h:38-87),
 * CompType is ne_comp_type (neural_speed/core/data_types.h:57-63), same numeric values as ns_comp_type below.
 * The blob layout is CPU-ISA specific in the reference (chosen from CPUID, bestla_gemm.cpp:241-300); this
 * library has no CPUID to consult, so the target core is a process-wide setting (default
 * NS_CORE_AUTO: comp int8 -> AVX512_VNNI k-block core, bf16 -> AMX_BF16, fp16 -> AMX_FP16, else AVX512F),
 * settable with ns_set_pack_core().  Loading accepts every reference core.
 * ---------------------------------------------------------------------------------------------- */
enum ns_comp_type { NS_COMP_UNDEF = 0, NS_COMP_F32 = 1, NS_COMP_BF16 = 2, NS_COMP_F16 = 3, NS_COMP_INT8 = 4 };
enum ns_core {
  NS_CORE_AVX2 = 0,
  NS_CORE_AVX512F = 1,
  NS_CORE_AMX_BF16 = 2,
  NS_CORE_AMX_FP16 = 3,
  NS_CORE_AVX512_VNNI_KB = 4,
  NS_CORE_AVX512BW_KB = 5,
  NS_CORE_AVX_VNNI_KB = 6,
  NS_CORE_AVX2_VNNI_KB = 7,
  NS_CORE_AMX_INT8_KB = 8,
  NS_CORE_AUTO = -1
};
void ns_set_pack_core(int core);
/* BTLAGemmPackBSize — bestla_gemm.cpp:626-639.  shuffle_indice != NULL (GPTQ act-order g_idx, host int[K]) adds the
 * int[K] ShuffleIndices section to integer-weight blobs (bestla_gemm.cpp:230-232) */
size_t ns_BTLAGemmPackBSize(size_t N, size_t K, size_t BlkSize, uint32_t QuantType, uint32_t ScaleDtype, bool isAsym,
                            int CompType, int* shuffle_indice);
/* BTLAGemmQuantPackB — bestla_gemm.cpp:641-655 (quantize + pack on the GPU, bit-exact blob) */
bool ns_BTLAGemmQuantPackB(void* PackedBuf, const float* FpData, size_t N, size_t K, size_t ldb, size_t BlkSize,
                           uint32_t QuantType, uint32_t ScaleDtype, bool isAsym, int CompType, bool isTrans,
                           void* ThreadPool);
/* BTLAGemmPackB — bestla_gemm.cpp:657-671 (pre-quantized int8 codes + fp32 scales + zero points).  With
 * shuffle_indice (g_idx[k] = group of input channel k; QData rows already sorted by group, as the reference converter
 * does) the blob records the permutation (setShuffleIndices, bestla_prologue_b.h:337-356) and every forward on it
 * gathers A'[j] = A[indices[j]] first. */
bool ns_BTLAGemmPackB(void* PackedBuf, const int8_t* QData, const float* Scales, const int8_t* Zp, size_t N, size_t K,
                      size_t ldb, size_t BlkSize, uint32_t QuantType, uint32_t ScaleDtype, bool isAsym, int CompType,
                      int* shuffle_indice, void* ThreadPool);
/* BTLAGemmUnPackB — bestla_gemm.cpp:673-749 */
bool ns_BTLAGemmUnPackB(float* FpData, const void* PackedBuf, size_t N, size_t K, size_t ldb, void* ThreadPool);

/* ------------------------------------------------------------------------------------------------
 * Part 3 — device-resident API (precedent: bestla_device_* , ne_bestla.h:85-112)
 * ---------------------------------------------------------------------------------------------- */
typedef struct ns_weight ns_weight; /* opaque: one weight matrix resident in HBM in the MI355X layout */

int ns_hip_device_count(void);
/* clears a sticky HIP runtime error (e.g. left by an invalidated stream capture) and ns_hip_last_error() */
void ns_hip_reset_error(void);
/* part-1 entry points cache {host blob pointer -> device weight} (validated by a content fingerprint); this
 * drops every cached device weight, e.g. after the model that owned the blobs was unloaded. */
void ns_hip_cache_clear(void);
/* per calling thread (like errno); valid until that thread's next failing call */
const char* ns_hip_last_error(void);
/* Header-only check of a reference-format blob, no device needed: the same validation every loader entry applies —
 * geometry (n, k, pads, tile divisibility) AND that each section (codes, scales, zero points, reductions, shuffle
 * indices) is at least as large as the geometry requires and ends inside the blob's serialized size.  avail_bytes
 * (0 = unknown) additionally bounds the serialized size, e.g. by what is left of the model file.  0 = ok; -1 with
 * ns_hip_last_error() otherwise.  A truncated / corrupt model file is refused here instead of faulting in HBM. */
int ns_hip_blob_validate(const void* host_blob, size_t avail_bytes);
/* host blob (reference format) -> device weight.  The blob is only read. */
ns_weight* ns_hip_weight_from_blob(const void* host_blob, void* stream);
/* Load path of the reference's device loader (model_files.h:1515-1527): like ns_hip_weight_from_blob, but (1) the streaming
 * layout is written INTO `dst` (dst_bytes: the slice the graph reserved for the tensor, ne_layers.c:918-945) whenever it fits
 * there — no second copy of the model in HBM; (2) nothing is synchronised: the two words the load needs back land in
 * `pinned_info[2]` (host memory the copy can target asynchronously) once the stream has been synchronised, and
 * ns_hip_weight_finish_load(w, pinned_info) completes the weight then.  The host blob may be freed on return. */
ns_weight* ns_hip_weight_load_async(const void* host_blob, void* dst, uint64_t dst_bytes, void* stream, uint32_t* pinned_info);
int ns_hip_weight_finish_load(ns_weight* w, const uint32_t* pinned_info);
int ns_hip_weight_is_external(const ns_weight* w);  /* 1: the weight lives in the caller's slice (not freed by ns_hip_weight_free) */
void ns_hip_load_staging_release(void);              /* frees the loader's device staging buffer */
/* what bestla_device_load_storage has done so far: out[0] tensors, [1] blob bytes, [2] bytes of streaming layout placed in the
 * graph's slices, [3] bytes in allocations of their own (layout larger than the blob), [4] microseconds inside the calls
 * (+ the one synchronisation), [5] tensors still waiting for that synchronisation */
void ns_hip_device_load_stats(uint64_t out[6]);
/* same, blob bytes already in device memory (dev_blob_base_mod64 = (host address the blob was packed at) & 63;
 * blobs packed by this library at 64-byte aligned bases use 0) */
ns_weight* ns_hip_weight_from_device_blob(const void* dev_blob, size_t blob_bytes, void* stream);
void ns_hip_weight_free(ns_weight* w);
/* Tensor-parallel shard producer (the reference's bestla_split_weight, models/model_utils/model_files.h:1538-1563,
 * split rules :145-190): returns a NEW device weight holding columns [n0, n1) x rows [k0, k1) of `w`.
 * Unlike the reference (dequantize -> slice -> re-quantize) the quantized codes/scales are sliced directly, so the
 * shards are bit-identical to the unsharded quantization.  n0 must be a multiple of 16 and k0 a multiple of
 * lcm(128 (64 for 8-bit), group size); the reference's re-quantizing behaviour is available on the host path via
 * bestla_unpackweight_fp32 + bestla_packweight_copyattr. */
ns_weight* ns_hip_weight_slice(const ns_weight* w, int n0, int n1, int k0, int k1, void* stream);
/* The same cut on the HOST, blob to blob, before anything is uploaded — what the reference's loader does per rank with
 * bestla_split_weight (model_files.h:1538-1563, :1593-1640), without the fp32 round trip: codes, scales, zero points and block
 * sums of columns [n0, n1) x rows [k0, k1) are copied into a new reference-format blob of that shape (same core, dtype, group
 * size), byte-identical to quantising the cut matrix afresh.  k0 (and k1 unless it is K) must be multiples of the group size;
 * returns -2 for cuts that need the re-quantising route, -1 for errors.  ns_bestla_split_weight_size: bytes to provide. */
int ns_blob_shape(const void* blob, int* n, int* k);  /* N and K of a host blob (header only); -1 if it is not one */
unsigned long long ns_bestla_split_weight_size(const void* src_blob, int dst_n, int dst_k);
int ns_bestla_split_weight(const void* src_blob, void* dst_blob, unsigned long long dst_capacity, int n0, int n1, int k0, int k1);
int ns_hip_weight_info(const ns_weight* w, int* n, int* k, int* bits, int* blocksize, uint64_t* device_bytes);
/* algorithmic bytes one forward over this weight streams: packed codes + scales (+ zero points), i.e.
 * N*K*bits/8 + N*(K/g)*sizeof(scale) [+ N*(K/g)] — the reference benchmark's formula (ut/bestla_benchmark.cpp:583-586) */
uint64_t ns_hip_weight_stream_bytes(const ns_weight* w);
/* Pull [offset, offset + bytes) of a weight's device stream (codes, scales, zero points; bytes is clamped) into the
 * Infinity Cache with a light read-only kernel of `workgroups` (<= 0: 64) workgroups on `stream`.  No reference
 * counterpart (the CPU path relies on hardware prefetchers); meant for a second stream / graph branch running beside
 * the previous GEMV of a decode chain, whose ramp-up and tail leave HBM idle.  Purely a cache hint: results of every
 * forward are unchanged whether or not it ran. */
int ns_hip_weight_prefetch(const ns_weight* w, uint64_t offset, uint64_t bytes, int workgroups, void* stream);

/* Numerics of the forwards over INTEGER weights (S1..S8).  The reference runs blobs packed for an integer compute core
 * (compute_dtype=int8, the default) with dynamically quantized u8 activations: ActivationKBlockQuantize
 * (bestla_prologue_a.h:133-154) + gemv_4bit_u8s8_fp32 (kernel_ref.h:2371-2429) / ref_kblock_int8
 * (bestla/ut/bestla_gemm.cpp:159-190).  NS_COMPUTE_FP16 (default) multiplies the same weights by fp16 activations with
 * fp32 accumulation — within 1e-3 of the reference's fp32-compute path and closer to it than the int8 path is;
 * NS_COMPUTE_REF_INT8 reproduces the int8 path itself (bit-exact activation quantization, exact integer dots per
 * k-block, fp32 scale products; only the fp32 summation order differs from the scalar reference) for callers that
 * compare against CPU int8 baselines.  Float weights (NF4 / FP4 / FP8) have no int8 path in the reference either and
 * are unaffected.  Process-wide; also NS_COMPUTE=ref_int8 in the environment.  set returns the previous mode, -1 on a
 * bad argument.  A numerics mode, not a tuned kernel. */
enum ns_compute_mode { NS_COMPUTE_FP16 = 0, NS_COMPUTE_REF_INT8 = 1 };
int ns_hip_set_compute_mode(int mode);
int ns_hip_get_compute_mode(void);

/* Diagnostics / A-B switches of the kernels (process-wide, take effect at the next launch or capture):
 *   "gemv2"           0 = first-generation decode kernel only, 1 = gemv_kernel (default); also NS_GEMV2 in the environment
 *   "g3_bm"           row-tile height of the prefill GEMM (128 / 256), 0 = automatic
 *   "g3_wide"         1 = the prefill GEMM's cross-wave output epilogue wherever the wave tiles are 1 x 4 (always used by launches
 *                     without an fp32 output and by the fused gate / up GEMM), 0 (default) = the per-wave one, -1 = NS_G3_WIDE / default
 *   "i8_mfma"         NS_COMPUTE_REF_INT8 at 16 rows and up: 2 = one exact fp16 MFMA per 32-deep slice on
 *                     operands with both zero points folded in (default), 1 = the first kernel (integer MFMA + corrections
 *                     per accumulator); bit-identical results; also NS_I8_MFMA in the environment
 *   "i8_tile"         workgroup tile of that kernel: 0 = by problem size (default), 1 = 64 x 64 (four waves), 4 = 64 x 256
 *                     (sixteen waves; what large problems take); other values = default
 *   "attn_wg_target", "attn_min_keys"   context-split rule of the decode attention kernel (defaults 1024, 128)
 *   "planes_load"     1 = weights of the 1-3 / 5 / 6 bit formats loaded FROM NOW ON also get a second device copy whose code records
 *                     have the format's own width (bit planes back to back, 256 .. 768 bytes per k-step instead of the 1 KiB nibble /
 *                     byte record every other kernel reads); 0 (default; NS_PLANES=1 in the environment flips it) = no such copy
 *   "planes"          1 (default) = the decode kernel streams that copy where a weight has one, 0 = always the widened records;
 *                     bit-identical results (measured: 0.89-1.07 x, DESIGN.md section 4.2c - hence off at load by default)
 *   "attn_mfma2_rows" query rows from which a prefill takes the 128-row matrix-core attention kernel (0 = default 128; a value
 *                     above every sl_q keeps the 64-row kernel of rounds 1-3)
 *   "attn_stream"     1 (default) = decode attention (head sizes 40 .. 128, contiguous head dimension) moves K / V HBM -> LDS by DMA into
 *                     per-wave rings, the whole context range of a workgroup in flight at once (attn_stream_kernel); 0 = through
 *                     registers (attn_split_kernel); the two agree to fp32 rounding.  "attn_stream_wg_target" / "attn_stream_min_keys": its
 *                     context-range rule (defaults 256 workgroups, >= 32 keys per range)
 *   "attn_inlaunch"   0 (default) = attn_merge_kernel combines the context splits in a second launch, 1 = the split that finishes
 *                     last does inside the launch (one self-resetting counter per kv head); same sums in the same order, same time
 *   "attn_heads_first" dispatch order of the decode attention's workgroups: 1 = the kv heads of one context range side by side, 0 = the
 *                     ranges of one head, -1 (default) = by the cache layout (position-major caches take 1: 5-10 % faster at 2048 / 4096 keys)
 *   "gv_nw"           waves per 16-column tile of the decode kernels (2 / 4 / 8 / 16), 0 = by shape (default); the
 *                     partial sums of a tile are added in wave order, so this selects the summation order
 *   "g3_min_m"        rows from which the tiled prefill GEMM is used inside its envelope (0 = default)
 * Returns 0, or -1 for an unknown key. */
int ns_hip_set_tuning(const char* key, int value);
/* Loads the code objects of the hot kernels (tiled GEMM, decode GEMV, attention, the operators between them) for the current device now instead of at the
 * first launch from each (36 + 21 + ... ms otherwise paid by the first prompt and the first generated token); idempotent, called by bestla_create_device.
 * NS_WARM_UP=0 turns it off. */
int ns_hip_warm_up(void);

/* The reference's per-token device graph: deferred, fused, verified, replayed (csrc/ns_route.cpp; the reference rebuilds its graph every token,
 * models/llama/llama.cpp:148, and issues it node by node, core/ne_layers.c:11915-12028).  The launches of the bestla_device_* route on a
 * queue bestla_create_device made (one route per queue) are recorded, not launched: they go out at the evaluation's next synchronisation point
 * (bestla_device_sync / _memcpy) as this library's fused launches — one-launch QKV, gate / up, mul_mat + residual add, rope + cache writes (the WINDOW,
 * round 6: a prompt and the first tokens of a generation too).  Two consecutive tokens whose launch sequences differ only in one moving value per launch
 * (RoPE position, kv-cache cell, context length) make a PLAN of HIP-graph segments; later tokens are compared launch by launch and each
 * segment is replayed once its last launch has matched - a token that deviates falls back to the window without side effects (its input is put back from a
 * device-side copy), and from the second fall-back in a row the next plan waits for 4, 8 .. 64 agreeing tokens.
 * ns_hip_route_set_enabled(on) (environment NS_DEVICE_REPLAY=0: all off; NS_ROUTE_WINDOW=0: window off) returns the previous plan setting:
 *   0 = the layer is off, every operator launches when it is handed over; 1 = window + plans; 8 = the window alone.
 * A plan CARRIES the RMS norms (ns_norm_link below: rms_norm + mul(gamma) in front of a mul_mat launch are not launched, the launch that made the normed
 * tensor writes fp16(gamma . x) and the sums of squares; 7B-shaped model 400 -> 451 tok/s) unless NS_ROUTE_LINKS=0 or ns_hip_route_set_enabled(5) say
 * otherwise ((3) forces it on).  The route's attention reads an fp16 MIRROR of the reference's fp32 device kv cache (csrc/ns_route.h: NS_DEVICE_KV=f32 /
 * ns_hip_set_tuning("device_kv_f16", 0) keep the fp32 kernels).  Both fp16 shortcuts are guarded: a value beyond the fp16 range raises a flag, the route
 * turns them off for the process (one line on stderr) and evaluates the token again on the fp32 forms before its results are read.
 * ns_hip_route_stats (sums over the process's routes): [0] tokens replayed,
 * [1] evaluations not replayed (window / plain), [2] plans built, [3] fall-backs, [4] the reference's launches per token in the last plan, [5] the launches its graphs hold for them (runs of
 * single operators become the library's fused launches at capture time),
 * [6] plans that could not be captured, [7] 1 while a plan is held. */
int ns_hip_route_set_enabled(int on);
void ns_hip_route_stats(uint64_t out[8]);

/* epilogue selector for the device forwards */
enum ns_epilogue {
  NS_EPI_NONE = 0,      /* AccumulatorWriteBackFp32 (bestla_epilogue.h:114-136) */
  NS_EPI_ADD = 1,       /* custom::epilogue::Add      (bestla_common.hpp:121-147): C = acc + D */
  NS_EPI_MUL = 2,       /* custom::epilogue::Mul      (bestla_common.hpp:149-181): C = acc * D */
  NS_EPI_ADD_GELU = 3,  /* custom::epilogue::Add_Gelu (bestla_common.hpp:183-213): C = gelu(acc + D) */
  NS_EPI_GELU = 4,      /* AccumulatorWriteBackWithGeluFp32 */
  NS_EPI_SILU = 5       /* AccumulatorWriteBackWithSwishFp32 (alpha = -1) */
};

/* dC[m][ldc] = epi(dA[m][lda] * W, dD[m][ldd]);  ldd = 0 broadcasts one row of D (bias) */
int ns_hip_f32f32_forward(const float* dA, const ns_weight* w, float* dC, int m, int lda, int ldc, int epilogue,
                          const float* dD, int ldd, void* stream);
/* dC = {A*Wq | A*Wk | A*Wv} stacked along dim 0 exactly as ip_fusion_qkv.cpp:84-86 (C, C+m*ldc, C+2*m*ldc) */
int ns_hip_fusion_qkv_forward(const float* dA, const ns_weight* wq, const ns_weight* wk, const ns_weight* wv, float* dC,
                              int m, int lda, int ldc, void* stream);
/* act: NS_EPI_SILU or NS_EPI_GELU.  tmp1 = act(A*W1); tmp2 = (A*W3) * tmp1; out = tmp2 * W2
 * (ip_fusion_ffn.cpp:364-406).  tmp1 may be NULL on the device path (the fused gate/up kernel writes tmp2 only). */
int ns_hip_fusion_ffn3_forward(const float* dA, const ns_weight* w1, const ns_weight* w2, const ns_weight* w3,
                               float* dTmp1, float* dTmp2, float* dOut, int seq, int act, void* stream);
/* the first stage of ffn3 alone: tmp2 = (A*W3) * act(A*W1) in ONE launch (tmp1 optional) */
int ns_hip_fusion_ffn3_gateup(const float* dA, const ns_weight* w1, const ns_weight* w3, float* dTmp1, float* dTmp2,
                              int seq, int act, void* stream);
/* tmp1 = gelu(A*W1 [+ b1]); out = tmp1*W2 [+ b2]   (ip_fusion_ffn.cpp ffn_2w) */
int ns_hip_fusion_ffn2_forward(const float* dA, const ns_weight* w1, const ns_weight* w2, const float* dB1,
                               const float* dB2, float* dTmp1, float* dOut, int seq, bool broadcast_bias, void* stream);

/* "_h" variants: the same operators with an optional fp16 SHADOW of the activations.  dA16 (may be NULL) is an fp16
 * copy of dA with the same shape/leading dimension — the kernels compute with fp16 activations anyway, so a producer
 * that already holds them saves the fp32->fp16 staging pass; dC16 (may be NULL) asks the epilogue to also write the
 * fp16 copy of the fp32 output for the next GEMM.  Results are bit-identical to the plain entry points. */
int ns_hip_f32f32_forward_h(const float* dA, const void* dA16, const ns_weight* w, float* dC, void* dC16, int m, int lda,
                            int ldc, int epilogue, const float* dD, int ldd, void* stream);
int ns_hip_fusion_qkv_forward_h(const float* dA, const void* dA16, const ns_weight* wq, const ns_weight* wk,
                                const ns_weight* wv, float* dC, void* dC16, int m, int lda, int ldc, void* stream);
int ns_hip_fusion_ffn3_gateup_h(const float* dA, const void* dA16, const ns_weight* w1, const ns_weight* w3,
                                float* dTmp1, float* dTmp2, void* dTmp2_16, int seq, int act, void* stream);
int ns_hip_fusion_ffn3_forward_h(const float* dA, const void* dA16, const ns_weight* w1, const ns_weight* w2,
                                 const ns_weight* w3, float* dTmp1, float* dTmp2, void* dTmp2_16, float* dOut,
                                 void* dOut16, int seq, int act, void* stream);

/* "_x" variants: the RMS norm that precedes a GEMM in every model graph (ne_rms_norm -> ne_mul(gamma) -> ne_mul_mat,
 * e.g. models/llama/llama.cpp:178-184, :385-391) CARRIED across operators instead of run as launches of its own.
 * y = W (gamma . x / rms(x)) = (W (gamma . x)) / rms(x): the operator that PRODUCES x (attention-output or FFN-down
 * projection with the residual add as its epilogue) also writes the fp16 shadow of gamma . x and, per 16-column output
 * tile, the sum of x^2; the operator that CONSUMES it streams that shadow and divides its finished dot products by
 * rms(x) = sqrt(sum / norm_size + eps) (ne_compute_forward_rms_norm_f32: scale = 1 / sqrtf(mean + eps)).  Sums are
 * added in a fixed order (run-to-run identical results).  Decode sizes only: m <= 16, fp16 shadow required; anything
 * else returns -1 (never a silently un-normalised result).
 *   consumer side (in_ssq != NULL): dA16 holds gamma . x, not normalised; in_ssq[row * in_stride + t], t < in_parts
 *                  (16-byte aligned, in_stride a multiple of 4); eps, norm_size of the norm
 *   producer side (out_gamma and/or out_ssq != NULL; single-matrix forward only): dC16 <- fp16(v * out_gamma[col]),
 *                  out_ssq[row * out_stride + tile] <- sum over the tile's columns of v^2 (tiles = ceil(N / 16))
 * ns_hip_norm_prep does the producer side for a tensor that no GEMM produced (the embedding row of layer 0).
 * RANGE: the shadow holds gamma . x BEFORE normalisation in fp16: |gamma[col] * x[col]| must stay below 65504 (an
 * un-carried norm's shadow holds the normalised value and has no such limit).  Residual streams of the model families the
 * reference ships stay orders of magnitude below that; a caller whose residual stream can exceed it keeps the norm as a
 * launch of its own (ns_hip_norm_mul_h) — an overflowed element reads as inf and the consumer's output row as inf / nan,
 * never as a silently wrong finite value. */
typedef struct ns_norm_link {
  const float* in_ssq;
  int in_parts, in_stride;
  float eps;
  int norm_size;
  const float* out_gamma;
  float* out_ssq;
  int out_stride;
} ns_norm_link;
int ns_hip_f32f32_forward_x(const float* dA, const void* dA16, const ns_weight* w, float* dC, void* dC16, int m, int lda,
                            int ldc, int epilogue, const float* dD, int ldd, const ns_norm_link* link, void* stream);
int ns_hip_fusion_qkv_forward_x(const float* dA, const void* dA16, const ns_weight* wq, const ns_weight* wk,
                                const ns_weight* wv, float* dC, void* dC16, int m, int lda, int ldc,
                                const ns_norm_link* link, void* stream);
int ns_hip_fusion_ffn3_gateup_x(const float* dA, const void* dA16, const ns_weight* w1, const ns_weight* w3,
                                float* dTmp1, float* dTmp2, void* dTmp2_16, int seq, int act, const ns_norm_link* link,
                                void* stream);
/* The fused QKV launch with ne_rope(q), ne_rope(k) and the kv-cache append (models/llama/llama.cpp:232-262; the three
 * operators of ns_hip_rope_qkv_append) as its EPILOGUE: q is written rotated to dC[0], k rotated and v to dC[1], dC[2]
 * and, as fp16, to cache position n_past + row.  RoPE mode 0 (adjacent pairs) over the whole head (n_dims ==
 * head_size), no YaRN; wq->n == heads * head_size, wk->n == wv->n == heads_kv * head_size; m <= 16 rows = consecutive
 * positions n_past, n_past + 1, ...; fp16 shadow of A required.  The angles do not depend on the layer: cos_sin is the
 * table ns_hip_rope_cos_sin fills ONCE per token ([m][head_size / 2] pairs (cos, sin) * attn_factor, theta built by the
 * reference's sequential fp32 products), shared by every layer's launch.  Same arithmetic as ns_hip_rope_qkv_append
 * (bitwise).  m > 16 (round 5): the tiled GEMM carries the epilogue (see NS_QKV_ROPE_KV_CACHE_ONLY below); no norm link there. */
typedef struct ns_qkv_rope {
  void* kcache16;
  void* vcache16;
  const float* cos_sin;
  int heads, heads_kv, head_size, n_past, n_dims, mode;
  long long cache_step_sl, cache_step_head; /* cache element strides per position / per head */
  int flags;                                /* NS_QKV_ROPE_* (0 = as before).  Added in round 5 as the LAST field: zero the struct before filling it —
                                             * unknown bits are refused (-1), never interpreted */
} ns_qkv_rope;
/* Round 5: the same epilogue at PREFILL size (m > 16: the tiled GEMM, fused QKV as column segments; head_size a multiple of 4, matrix widths
 * multiples of 128, cos_sin rows for all m positions).  k and v then need not exist as fp32 tensors at all - the attention reads the cache: */
#define NS_QKV_ROPE_KV_CACHE_ONLY 1 /* m > 16 only: k (rotated) and v go to the fp16 cache alone, dC[1] / dC[2] are not written */
int ns_hip_rope_cos_sin(int m, int n_past, int n_dims, float freq_base, float freq_scale, float attn_factor,
                        float* dCosSin, void* stream);
int ns_hip_fusion_qkv_rope_forward_x(const float* dA, const void* dA16, const ns_weight* wq, const ns_weight* wk,
                                     const ns_weight* wv, float* dC, int m, int lda, int ldc, const ns_norm_link* link,
                                     const ns_qkv_rope* rope, void* stream);
/* dX [m][ldx] fp32 -> dX16 [m][ldx] = fp16(x * dGamma[col]) and dSsq[row * ssq_stride + t] = sum of x^2 over columns
 * 16 t .. 16 t + 15 (ssq_stride >= ceil(n / 16)) */
int ns_hip_norm_prep(int m, int n, const float* dX, int ldx, const float* dGamma, void* dX16, float* dSsq, int ssq_stride,
                     void* stream);

/* device-pointer twins of bestla_layernormalization / bestla_mul / bestla_add (ne_bestla.h:79-83; device precedent
 * bestla_device_rms_norm_f32 / _mul_f32 / _add_f32, ne_bestla.h:99-105): asynchronous on `stream`, capturable */
int ns_hip_layernormalization(int norm_count, int norm_size, bool isrms, float epsilon, const float* dIn, float* dOut,
                              void* stream);
/* bestla_device_elewise_f32 (NE_OP_SILU; ne_bestla.h:103-104, ne_bestla_sycl.cpp:297-326): y = x / (1 + expf(-x)) */
int ns_hip_silu_f32(const float* dSrc, float* dDst, size_t n, void* stream);
/* bestla_device_dup_f32 (ne_bestla.h:109, ne_bestla_sycl.cpp:537-591): 4-D strided copy of fp32 into fp32 or fp16;
 * ne = extents of dst, strides in BYTES as in ne_tensor::nb */
int ns_hip_dup_f32(const float* dSrc, void* dDst, const long long ne[4], const long long src_nb[4],
                   const long long dst_nb[4], bool dst_is_f16, void* stream);
int ns_hip_mul(int batch, int vsize, const float* dTensor, const float* dVector, int vstep, float* dOut, void* stream);
/* layernormalization fused with the multiplication by the norm weight that follows it in every model graph
 * (dGamma [norm_size], may be NULL) and with the fp16 shadow of the result (dOut16, may be NULL) that the "_h" GEMM
 * entries read.  Same arithmetic as the two separate operators (the product is rounded separately). */
int ns_hip_norm_mul_h(int norm_count, int norm_size, bool isrms, float epsilon, const float* dIn, const float* dGamma,
                      float* dOut, void* dOut16, void* stream);
/* RoPE on a contiguous fp32 tensor [batch][seq][heads][head_size] (in place when dDst == dSrc):
 * ne_compute_forward_rope_f32, /root/reference/neural_speed/core/ne_layers.c:9243-9428 (device precedent
 * bestla_device_rope_f32, ne_bestla.h:106).  mode 0 = adjacent pairs over the whole row, mode 2 = NeoX halves;
 * freq_scale is the reciprocal already (1 / op_params[1]); ext_factor must be 0 (GLM, long-rope, shift and the YaRN
 * mix are refused). */
int ns_hip_rope_f32(const float* dSrc, float* dDst, int batch, int seq, int heads, int head_size, int n_past, int n_dims,
                    int mode, float freq_base, float freq_scale, float ext_factor, float attn_factor, void* stream);
/* the same with the YaRN extrapolation mix (ext_factor != 0; rope_yarn, ne_layers.c:9196-9231): op_params n_orig_ctx,
 * beta_fast, beta_slow give the correction dims, attn_factor is scaled by 1 + 0.1 * log(1 / freq_scale) */
int ns_hip_rope_f32_yarn(const float* dSrc, float* dDst, int batch, int seq, int heads, int head_size, int n_past,
                         int n_dims, int mode, float freq_base, float freq_scale, int n_orig_ctx, float ext_factor,
                         float attn_factor, float beta_fast, float beta_slow, void* stream);
/* long-rope (mode bit 0x10, ne_layers.c:9349-9377): dFactors = device array of n_dims / 2 per-pair divisors (the
 * graph's dst->opt[1]), scale_factor multiplies cos and sin; rows are walked like the NeoX mode */
int ns_hip_rope_f32_longrope(const float* dSrc, float* dDst, int batch, int seq, int heads, int head_size, int n_past,
                             int n_dims, float freq_base, float freq_scale, int n_orig_ctx, float ext_factor,
                             float attn_factor, float beta_fast, float beta_slow, const float* dFactors,
                             float scale_factor, void* stream);
/* GLM branch of the same operator (mode & 4, /root/reference/neural_speed/core/ne_layers.c:9317-9347; ChatGLM's
 * two-dimensional position encoding): the first half of every head is rotated by min(max(p - n_padding, 0),
 * prompt_size - 2 - n_padding), the second half by max(p - (prompt_size - 2), 0).  mode 4, or 5 (= 4 | skip: positions
 * < n_past are left untouched and p = the row's index).  n_padding: HOST array [batch] (src1[ROPE_PARAMS_NUM + i],
 * :9319), batch <= 32.  The "shift" form (n_keep >= 0) is asserted against by the reference itself (:9312). */
int ns_hip_rope_f32_glm(const float* dSrc, float* dDst, int batch, int seq, int heads, int head_size, int n_past, int n_dims,
                        int mode, float freq_base, int prompt_size, const int* n_padding, void* stream);
int ns_hip_add(int batch, int vsize, const float* dTensor, const float* dVector, int vstep, float* dOut, void* stream);

/* RoPE of Q (in place, [seq][heads][head_size]) and of K ([seq][heads_kv][head_size]) fused with the kv-cache append:
 * rotated K and V are stored as fp16 at cache positions n_past .. n_past+seq-1; cache element (position, head, e) lives
 * at position*cache_step_sl + head*cache_step_head + e.  One launch for ne_rope(q), ne_rope(k) and the two kv-cache
 * copies of the llama graph (/root/reference/neural_speed/models/llama/llama.cpp:232-262); same arithmetic as
 * ns_hip_rope_f32 followed by a float->half conversion.  Batch 1. */
int ns_hip_rope_qkv_append(float* dQ, const float* dK, const float* dV, void* dKcache16, void* dVcache16, int seq, int heads,
                           int heads_kv, int head_size, int n_past, int n_dims, int mode, float freq_base, float freq_scale,
                           float ext_factor, float attn_factor, long long cache_step_sl, long long cache_step_head,
                           void* stream);

/* activation prologue of the reference's int8-compute path: quantize_fp_u8_colblock
 * (/root/reference/bestla/bestla/kernel_ref.h:1824-1883, driven by ActivationKBlockQuantize::run, bestla_prologue_a.h:133-154).
 * dSrc fp32 [row][ld_src] -> dDst u8 [row][ld_dst], per (row, k-block) dScales / dZps [row][ld_scale] and, if not NULL,
 * dBlkReduce = sum(round(a / scale)) * scale.  Bit-exact with the scalar reference, tail blocks included. */
int ns_hip_quantize_fp_u8_colblock(int row, int col, const float* dSrc, int ld_src, uint8_t* dDst, int ld_dst,
                                   float* dScales, int ld_scale, uint8_t* dZps, int blocksize, float* dBlkReduce,
                                   void* stream);

/* quantize + pack entirely on the device: dW fp32 [N][K] (is_trans) or [K][N]; writes the reference-format blob
 * into dBlob (device memory, ns_BTLAGemmPackBSize bytes, 64-byte aligned) */
int ns_hip_quant_pack_device(void* dBlob, const float* dW, size_t N, size_t K, size_t ldb, size_t BlkSize,
                             uint32_t QuantType, uint32_t ScaleDtype, bool isAsym, int CompType, bool isTrans,
                             void* stream);

/* ---- Part 3a — the reference's device-backend set under its own names (ne_bestla.h:85-112, guarded there by NS_SYCL;
 * csrc/ns_device.hip).  The pointer-only functions are exported by libns_hip.so AS IS — a reference tree built with
 * -DNS_SYCL binds to them — and the tensor-level ones (bestla_device_mul_f32 / _add_f32 / _elewise_f32 / _rms_norm_f32 /
 * _rope_f32 / _dup_f32 / _mha_f32) are glue/ne_bestla_hip_device.c over the ns_hip_* entries.  "queue" = hipStream_t. ---- */
/* Threads: the device set is driven by ONE host thread at a time, as the reference's executor drives it (claimed nodes run with n_tasks = 1,
 * ne_layers.c:11915-12028).  Several device contexts may live in a process and be used in turn — each keeps its own recorded graph and plan — but the
 * registries behind them (queues, kv mirrors, pools) are not guarded against two threads inside bestla_device_* at the same moment. */
void* bestla_create_device(bool profile);
void* bestla_get_device_queue(void* device);
void bestla_release_device(void* device);
size_t bestla_device_gmem_size(void* device);
void* bestla_device_malloc(size_t size, void* queue);
void bestla_device_free(void* ptr, void* queue);
void bestla_device_memcpy(void* dstptr, const void* srcptr, size_t size, void* queue);
void bestla_device_memcpy_sync(void* dstptr, const void* srcptr, size_t size, void* queue);
/* Waits for the queue — except when nothing but kernel launches went onto it since it was last waited for (no copy in either direction): the wait is
 * then left to the next copy / synchronisation on the queue, which is ordered behind those launches (the reference ends an evaluation with sync, copy,
 * sync: ne_layers.c:8345-8346; a prompt's logits copy can then be prepared by the host while the launches run).  NS_ROUTE_LAZY_SYNC=0: always waits. */
void bestla_device_sync(void* queue);
size_t bestla_device_storage_size(void);
/* hoststor: BTLA blob in host memory; devstor: bestla_device_storage_size() bytes inside the tensor object
 * (ne_layers.c:946-949); deviceptr: the device-pool slice the graph reserved (unused: the weight gets its own allocation
 * in this library's streaming layout) */
void bestla_device_load_storage(void* hoststor, void* devstor, void* deviceptr, void* queue);
void bestla_device_f32f32_forward(float* activation, void* weiptr, float* output, int _m, int _n, int _k, int lda, int ldo,
                                  void* workspace, void* queue);
void ns_hip_device_storage_release(void* devstor);
/* ne's broadcasting add (is_mul = 0) / mul over four strided dimensions: dst[i] = a[i] op b[i mod ne1]; strides in bytes */
int ns_hip_binary_nd_f32(int is_mul, const float* dA, const float* dB, float* dDst, const long long ne0[4], const long long nb0[4],
                         const long long ne1[4], const long long nb1[4], const long long nbd[4], void* stream);
/* attention over the device prototype's fp32 kv cache (ne_bestla_sycl.cpp:592-880): q / o [batch][seq][heads][head_size],
 * k [batch][heads_kv][n_ctx][head_size], v [batch][heads_kv][head_size][n_ctx]; masked: causal with n_past = seq_all - seq */
/* Lazy peephole of the device route (round 4): ns_hip_lazy_rms_norm / ns_hip_lazy_silu RECORD their node instead of launching it;
 * ns_hip_lazy_mul fuses a multiply that consumes the recorded result with it — one launch writing BOTH tensors, bit for bit what
 * the two kernels write — and everything else launches the recorded node first (ns_hip_lazy_flush; the bestla_device_* pointer
 * entries and ns_hip_binary_nd_f32 / ns_hip_mha_f32_device_layout call it themselves, the glue calls it in front of the ns_hip_*
 * entries it forwards to).  One recorded node at most, one issuing thread.  NS_DEV_LAZY=0: nothing is deferred. */
int ns_hip_lazy_flush(void);
int ns_hip_lazy_rms_norm(int rows, int cols, float eps, const float* dIn, float* dOut, void* stream);
int ns_hip_lazy_silu(const float* dSrc, float* dDst, size_t n, void* stream);
int ns_hip_lazy_mul(const float* dA, const float* dB, float* dDst, const long long ne0[4], const long long nb0[4], const long long ne1[4],
                    const long long nb1[4], const long long nbd[4], void* stream);
int ns_hip_mha_f32_device_layout(const float* dQ, const float* dK, const float* dV, float* dO, int batch, int seq, int seq_all, int heads,
                                 int heads_kv, int head_size, int n_ctx, float scale, int masked, void* stream);


/* ----------------------------------------------------------------------------------------------
 * Part 4 — fused attention: the C surface `ne_compute_forward_flash_attn_f32_f16_f16`
 * (/root/reference/neural_speed/core/ne_layers.c:10110-10214) marshals into.  Struct layouts are those of
 * /root/reference/neural_speed/core/layers/mha_dense.h:24-95 so that the ggml-side dispatch code compiles unchanged.
 * Semantics = bestla_fusion_attn_forward_ref (mha_dense_wrapper.h:1371-1517) in its PREFER_FP32 form:
 *   S[i][j] = (q_i . k_j) * QK_scale * Q_sc * K_sc  [tanh30: 30*tanh(S/30)]  + j * alibi_slope(head)
 *   causal: j <= i + (sl_kv - sl_q);  P = softmax_j(S);  dst[i] = (P . V) * V_sc / dst_sc
 * GQA: kv head = head / (head_num / heads_kv).  Only ATTN_FWD_LAYOUT_PLAIN tensors with arbitrary element strides
 * (incl. transposed K) are taken by the fp16 entries; the library-managed kv-cache entries below use their own layout.
 * ---------------------------------------------------------------------------------------------- */
typedef struct attn_shape_t {
  int batch_size, head_num, heads_kv, head_size, sl_q, sl_kv;
} attn_shape_t; /* mha_dense.h:24-26 */

typedef enum ATTN_FWD_LAYOUT {
  ATTN_FWD_LAYOUT_PLAIN,
  ATTN_FWD_LAYOUT_NTILE48_ROWPACK4,
  ATTN_FWD_LAYOUT_NTILE48_ROWPACK2,
  ATTN_FWD_LAYOUT_NTILE24_ROWPACK1,
} ATTN_FWD_LAYOUT; /* mha_dense.h:35-47 */

typedef uint32_t ns_attn_flags_t; /* ne_attn_flags_t, ne_layers.h:65-72 */
enum {
  NS_ATTN_FLAG_NONE = 0,
  NS_ATTN_FLAG_IS_CAUSAL = 1 << 0,
  NS_ATTN_FLAG_IS_ALIBI8 = 1 << 1,
  NS_ATTN_FLAG_PREFER_FP32 = 1 << 2,
  NS_ATTN_FLAG_IS_TANH30 = 1 << 3,
};

typedef struct attn_fp32_fp16_fp16_fp32_fwd_args_t {
  float* Q;
  uint16_t* K; /* ne_fp16_t */
  uint16_t* V;
  float* dst;
  float Q_sc, K_sc, V_sc, dst_sc;
  char* tmp;
  float QK_scale;
  ns_attn_flags_t attn_flags;
  int batch_size, head_num, heads_kv, head_size, sl_q, sl_kv;
  ATTN_FWD_LAYOUT Q_layout, K_layout, V_layout, dst_layout;
  int step_q_bs, step_q_head_num, step_q_sl;
  int step_k_bs, step_k_head_num, step_k_sl, step_k_head_size;
  int step_v_bs, step_v_head_num, step_v_sl, step_v_head_size;
  int step_dst_bs, step_dst_head_num, step_dst_sl;
} attn_fp32_fp16_fp16_fp32_fwd_args_t; /* mha_dense.h:66-81 */

/* mha_dense.h:27: scratch the CALLER allocates (`tmp`).  Here: the partial (max, sum, accumulator) records of the
 * context splits; the device entry below uses `tmp` as DEVICE memory of this size when it is not NULL */
size_t bestla_fusion_attn_workspace_size(const attn_shape_t* params);
/* mha_dense.h:85-86.  Host pointers: Q/K/V are uploaded, dst downloaded, synchronous (reference semantics). */
bool bestla_fusion_attn_fp32_fp16_fp16_fp32_support(const attn_shape_t* params);
/* mha_dense.h:106 — the all-fp16 variant (fp16 Q and dst) has one caller, compiled out by its own switch
 * (models/gptj/gptj.cpp:42-43, :141); declined, so that such a build still links: Q and dst are fp32 in every graph */
bool bestla_fusion_attn_fp16_support(const attn_shape_t* params);
void bestla_fusion_attn_fp32_fp16_fp16_fp32_forward(const attn_fp32_fp16_fp16_fp32_fwd_args_t* params);
/* The library-managed ("reordered") kv-cache, mha_dense.h:124-172.  The reference hands the cache to BesTLA as an opaque
 * buffer — sizes and view strides from batch_kv_info, contents only through update_k / update_v / shift_rope_k /
 * batch_cpy / forward — and packs it in AMX / AVX tile order.  Nothing else looks inside, so the MI355X form is the layout
 * its attention kernels stream best: plain fp16 [batch][head][seq_max][head_size] (k_layout = v_layout = PLAIN, byte
 * strides as the graph code expects them for its views, llama.cpp:544-560).  Host pointers, synchronous, like the
 * reference; a device-resident backend uses ns_hip_rope_qkv_append / ns_qkv_rope and the device attention entry. */
typedef struct kv_shape_t {
  uint32_t heads_kv, head_size, sl_kv_max;
} kv_shape_t; /* mha_dense.h:29-33 */
typedef struct kv_cache_info_t {
  size_t k_bytes, v_bytes;
  ATTN_FWD_LAYOUT k_layout, v_layout;
  int stride_k_head_num, stride_k_sl, stride_k_head_size;
  int stride_v_head_num, stride_v_sl, stride_v_head_size;
} kv_cache_info_t; /* mha_dense.h:49-54 */
typedef struct bestla_fusion_attn_fp32_update_kv_args_t {
  float* src;
  char* cache;
  int batch_size, heads_kv, head_size, seq_off, seq_size, seq_max;
  int step_bs, step_head_num, step_seq, step_head_size;
  bool no_zeroing;
} bestla_fusion_attn_fp32_update_kv_args_t; /* mha_dense.h:130-136 */
typedef struct bestla_fusion_attn_fp32_batch_cpy_kv_args_t {
  char* src;
  char* dst;
  int heads_kv, head_size, seq_off, seq_size, seq_max;
  bool no_zeroing;
} bestla_fusion_attn_fp32_batch_cpy_kv_args_t; /* mha_dense.h:145-150 */
typedef struct bestla_reordered_attn_fp32_fp32_fwd_args_t {
  float* Q;
  char* K;
  char* V;
  float* dst;
  float Q_sc, K_sc, V_sc, dst_sc;
  char* tmp;
  float QK_scale;
  ns_attn_flags_t attn_flags;
  int batch_size, head_num, heads_kv, head_size, sl_q, sl_kv;
  ATTN_FWD_LAYOUT Q_layout, K_layout, V_layout, dst_layout;
  int step_q_bs, step_q_head_num, step_q_sl;
  int stride_k_bs, stride_k_head_num, stride_k_sl, stride_k_head_size;
  int stride_v_bs, stride_v_head_num, stride_v_sl, stride_v_head_size;
  int step_dst_bs, step_dst_head_num, step_dst_sl;
} bestla_reordered_attn_fp32_fp32_fwd_args_t; /* mha_dense.h:156-171 */
bool bestla_reordered_attn_fp32_support(const attn_shape_t* params);                          /* mha_dense.h:125 */
void bestla_reordered_attn_fp32_batch_kv_info(const kv_shape_t* params, kv_cache_info_t* out); /* :128 */
void bestla_reordered_attn_fp32_update_k(const bestla_fusion_attn_fp32_update_kv_args_t* params); /* :138 */
void bestla_reordered_attn_fp32_update_v(const bestla_fusion_attn_fp32_update_kv_args_t* params); /* :140 */
/* :142-143: rows [seq_keep, seq_max) of every (batch, head) rotated by the one angle set cossin = {cos_0, sin_0, ...} (fp16) */
void bestla_reordered_attn_fp32_shift_rope_k(char* cache, const uint16_t* cossin, int batch_size, int heads_kv, int head_size,
                                             int seq_max, int seq_keep);
void bestla_fusion_attn_fp32_batch_cpy_k(const bestla_fusion_attn_fp32_batch_cpy_kv_args_t* params); /* :152 */
void bestla_fusion_attn_fp32_batch_cpy_v(const bestla_fusion_attn_fp32_batch_cpy_kv_args_t* params); /* :154 */
void bestla_reordered_attn_fp32_forward(const bestla_reordered_attn_fp32_fp32_fwd_args_t* params);   /* :172 */
/* same operator on DEVICE pointers, asynchronous on `stream`; returns 0 on success */
int ns_hip_attn_fp32_fp16_fp16_fp32_forward(const attn_fp32_fp16_fp16_fp32_fwd_args_t* dparams, void* stream);
/* same, also writing the fp16 shadow of dst (dst16: same element strides as dst; may be NULL) that the "_h" / "_x" GEMM
 * entries take as dA16 — the attention-output projection then needs no conversion pass */
int ns_hip_attn_fp32_fp16_fp16_fp32_forward_h(const attn_fp32_fp16_fp16_fp32_fwd_args_t* dparams, void* dst16, void* stream);
/* Tensor-parallel head split for NS_ATTN_FLAG_IS_ALIBI8 (mha_dense_wrapper.h:1418-1447 under NS_TP_MODEL: the slopes
 * follow head_num * world and start at rank * head_num).  Process-wide (one process per GPU); (0, 0) clears it.
 * A forward call whose heads do not fit the partition is refused. */
int ns_hip_attn_set_head_partition(int global_head_num, int head_offset);

/* ----------------------------------------------------------------------------------------------
 * Part 4b — mixture-of-experts matmul with the routing on the device.  Device twin of
 * ne_compute_forward_mul_mat_id_q_f32_bestla (/root/reference/neural_speed/core/ne_layers.c:7783-7916), which reads the
 * expert ids on the HOST and calls bestla_f32f32_forward once per (token, expert) — that path keeps working through the
 * part-1 surface.  Here the ids stay on the device (they are the router's top-k output), so a decode step with experts
 * can be captured in one HIP graph:
 *     dC[t][:] = epi( dA[t][:] . W[ dIds[t * ids_stride + id] ],  dD[t][:] )        t = 0 .. m-1
 * An expert group is n_as weights of identical shape and format (S1..S8, NF4 / FP4); an id outside [0, n_as) — the
 * reference asserts — produces a zero product for that row.  fp16-activation numerics like every default forward.
 * The `ffn_id_*` nodes (ne_layers.c:8053-8170) are three such calls (gate with NS_EPI_SILU, up with NS_EPI_MUL and
 * dD = gate output, down).
 * ---------------------------------------------------------------------------------------------- */
typedef struct ns_expert_group ns_expert_group;
ns_expert_group* ns_hip_expert_group_create(const ns_weight* const* experts, int n_as);
void ns_hip_expert_group_free(ns_expert_group* g);
int ns_hip_mul_mat_id(const float* dA, const int32_t* dIds, int ids_stride, int id, const ns_expert_group* g, float* dC,
                      int m, int lda, int ldc, int epilogue, const float* dD, int ldd, void* stream);

/* ----------------------------------------------------------------------------------------------
 * Part 5 — tensor-parallel all-reduce over peer-mapped HBM.  The decode-sized fast path of
 * `reduce_add` (/root/reference/neural_speed/core/parallel_context.cpp:47-58): the reference hands small buffers to
 * `shm_all_reduce` (/root/reference/neural_speed/core/shared_memory_ccl.hpp:100-139 — copy into a shared segment,
 * flag, wait, sum) and the rest to oneCCL; here small buffers go through ONE kernel that exchanges flags and reads the
 * peers' copies over xGMI (HIP IPC mappings), the rest through RCCL (neural-speed_amd/parallel.py).  One process per
 * GPU.  In-place fp32 sum, summed in rank order on every rank (bit-identical results everywhere).  Capturable.
 *   create  : allocates this rank's segment (2 slots of max_bytes + flag page) and returns its IPC handle (64 bytes)
 *   connect : maps the segments of all ranks; `all_handles` = world x 64 bytes in rank order (own entry ignored)
 *   all_reduce_f32 : n * 4 <= max_bytes, dBuf 16-byte aligned; asynchronous on `stream`.  ONE stream per context at a
 *             time (the sequence counter in the segment is owned by the kernel in flight; a captured graph must be
 *             replayed on a stream ordered with every other user of the context).  Returns -2, without launching,
 *             once an earlier call's flag wait has timed out (the kernel raises a pinned host word; no sync needed)
 *   error   : synchronous read of the sticky device status: 0 ok, 1 = a flag wait exceeded NS_P2P_TIMEOUT_MS (default
 *             10000; a peer died or never launched the matching call — results after that are undefined), -1 = bad ctx
 *   disconnect (every rank) -> barrier -> destroy
 * ---------------------------------------------------------------------------------------------- */
#define NS_P2P_HANDLE_BYTES 64
typedef struct ns_p2p ns_p2p;
ns_p2p* ns_hip_p2p_create(int rank, int world, size_t max_bytes, void* handle_out);
int ns_hip_p2p_connect(ns_p2p* ctx, const void* all_handles);
int ns_hip_p2p_all_reduce_f32(ns_p2p* ctx, float* dBuf, size_t n, void* stream);
int ns_hip_p2p_error(ns_p2p* ctx);
void ns_hip_p2p_disconnect(ns_p2p* ctx);
void ns_hip_p2p_destroy(ns_p2p* ctx);

/* ----------------------------------------------------------------------------------------------
 * Part 6 — tensor-parallel communication layer: the eight functions of
 * /root/reference/neural_speed/core/parallel_context.h:40-47 (impl parallel_context.cpp:19-160: oneCCL over MPI, one
 * process per CPU socket) as a native C ABI over RCCL / xGMI, one process per GPU, DEVICE fp32 buffers, asynchronous on
 * a stream, capturable.  RCCL is loaded at ns_tp_init (dlopen), not at library load.
 *   bootstrap : rank 0 calls ns_tp_unique_id (128 bytes) and hands the bytes to the other ranks by any out-of-band means
 *               (the reference uses MPI_Bcast of the oneCCL kvs address, parallel_context.cpp:86-98); every rank then
 *               calls ns_tp_init(rank, world, id, device).  world == 1 accepts id == NULL.
 *   reduce_add: fp32 sum over ranks (ne_compute_forward_all_reduce, ne_layers.c:5466-5476, calls it in place); buffers
 *               that fit an attached peer-memory context (ns_tp_attach_p2p, part 5) take the one-shot xGMI kernel,
 *               like the reference's shm_all_reduce shortcut (parallel_context.cpp:47-58)
 *   broadcast : from rank 0 (parallel_context.cpp:59-62);  alltoall: count elements per peer (:63-65, no caller)
 *   barrier   : synchronises the stream as well (the reference's is a blocking host call)
 * The reference-named HOST-pointer functions (init_parallel_context, reduce_add, ...) for a ggml build are
 * glue/parallel_context_hip.cpp.  All functions return 0 / non-NULL on success; ns_hip_last_error() has the reason.
 * ---------------------------------------------------------------------------------------------- */
#define NS_TP_UNIQUE_ID_BYTES 128
typedef struct ns_tp ns_tp;
int ns_tp_unique_id(void* out128);
ns_tp* ns_tp_init(int rank, int world, const void* unique_id128, int device);
void ns_tp_destroy(ns_tp* tp);
int ns_tp_size(const ns_tp* tp);
int ns_tp_rank(const ns_tp* tp);
int ns_tp_is_master(const ns_tp* tp);
int ns_tp_attach_p2p(ns_tp* tp, ns_p2p* p2p, size_t max_bytes);
int ns_tp_reduce_add(ns_tp* tp, const float* dSend, float* dRecv, size_t count, void* stream);
int ns_tp_broadcast(ns_tp* tp, float* dBuf, size_t count, void* stream);
int ns_tp_alltoall(ns_tp* tp, const float* dSend, float* dRecv, size_t count, void* stream);
int ns_tp_barrier(ns_tp* tp, void* stream);
/* host-pointer forms (blocking, staged through device memory; with one rank plain copies that touch no GPU —
 * ns_tp_init(0, 1, NULL, -1) creates such a context without a device) */
int ns_tp_reduce_add_host(ns_tp* tp, const float* send, float* recv, size_t count);
int ns_tp_broadcast_host(ns_tp* tp, float* buf, size_t count);
int ns_tp_alltoall_host(ns_tp* tp, const float* send, float* recv, size_t count);
int ns_tp_barrier_host(ns_tp* tp);

#ifdef __cplusplus
}
#endif
#endif /* NS_BESTLA_H */
