/*
 * ns_oracle.h — CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain scalar restatement of the BesTLA weight-only-quant path of intel/neural-speed, used ONLY as the
 * checker by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing under
 * neural-speed_amd/ links, imports or calls this file.
 *
 * Parity pinning: the reference holds no golden vectors for this path, so every arithmetic
 * routine here is pinned against the reference's own scalar kernels (bestla/bestla/kernel_ref.h) compiled
 * from /root/reference into oracle/_ref/libkernel_ref.so (see oracle/Makefile, oracle/ref_shim.cpp) and the
 * outputs are frozen as fixtures under tests/golden/ (generator: tests/golden/make_golden.py).  The blob
 * container (bestla_storage.h) cannot be compiled here (it pulls in xbyak, which is not vendored), so the
 * serialisation order is restated from the source text and self-checked by pack→parse→unpack round trips.
 *
 * Every function cites the reference file:line it follows.
 */
#ifndef NS_ORACLE_H
#define NS_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* BTLA_DTYPE bit encoding — bestla/bestla/bestla.h:38-87 */
enum {
  NSO_F32 = 32,
  NSO_F16 = 16,
  NSO_BF16 = 16 | (1 << 16),
  NSO_S8 = 8 | (1 << 8),
  NSO_S1_CLIP = 1 | (1 << 8),
  NSO_S2_CLIP = 2 | (1 << 8),
  NSO_S3_CLIP = 3 | (1 << 8),
  NSO_S4_CLIP = 4 | (1 << 8),
  NSO_S5_CLIP = 5 | (1 << 8),
  NSO_S6_CLIP = 6 | (1 << 8),
  NSO_S7_CLIP = 7 | (1 << 8),
  NSO_F4_E2M1 = 4,
  NSO_F4_BNB = 4 | (1 << 16),
  NSO_F4_NF4 = 4 | (2 << 16),
  NSO_F8_E4M3 = 8,              /* weight types of WeightKBlockNFloat besides the f4 family */
  NSO_F8_E5M2 = 8 | (1 << 16),
  NSO_F8_E8M0 = 8 | (3 << 16),  /* scale type: int8 shared exponent, scale = 2^e */
  NSO_DQ8_BNB = 8 | (4 << 16),  /* scale type: double-quantised scales, u8 codes into the bitsandbytes dynamic map (bestla.h:72) */
};

/* GEMM cores a blob can be laid out for — neural_speed/core/layers/bestla_defs.h:36-54 */
enum {
  NSO_CORE_AVX2 = 0,            /* SCoreRowNAvx2<24,4>            NTILE 24 PACK 1 KTILE 1  fp32 */
  NSO_CORE_AVX512F = 1,         /* SCoreRowNAvx512f<48,8>         NTILE 48 PACK 1 KTILE 1  fp32 */
  NSO_CORE_AMX_BF16 = 2,        /* HCoreRowNAmxbf16<48,16>        NTILE 48 PACK 2 KTILE 32 bf16 */
  NSO_CORE_AMX_FP16 = 3,        /* HCoreRowNAmxfp16<48,16>        NTILE 48 PACK 2 KTILE 32 fp16 */
  NSO_CORE_AVX512_VNNI_KB = 4,  /* ICoreRowNAvx512vnniKBlock<48,4> NTILE 48 PACK 4 KTILE 4 u8s8 */
  NSO_CORE_AVX512BW_KB = 5,     /* ICoreRowNAvx512bwKBlock<48,8>   NTILE 48 PACK 4 KTILE 4 u8s8 */
  NSO_CORE_AVX_VNNI_KB = 6,     /* ICoreRowNAvxvnniKBlock<24,2>    NTILE 24 PACK 4 KTILE 4 u8s8 */
  NSO_CORE_AVX2_VNNI_KB = 7,    /* ICoreRowNAvx2vnniKBlock<24,2>   NTILE 24 PACK 4 KTILE 4 u8s8 */
  NSO_CORE_AMX_INT8_KB = 8,     /* ICoreRowNAmxint8KBlock<48,16>   NTILE 48 PACK 4 KTILE 64 u8s8 */
  NSO_CORE_COUNT = 9
};

typedef struct nso_blob_info {
  uint64_t size;
  uint32_t prologue_id; /* 1 = WeightKBlockNInteger, 2 = WeightKBlockNFloat (bestla.h:91-102) */
  uint64_t core_id;
  int32_t npad, kpad, n, k;
  uint32_t dtype;
  int32_t blocksize, dq_blocksize;
  uint32_t scale_dtype, zp_dtype, red_dtype;
  int32_t cstep;
  uint64_t csize;
  int32_t ntile, packrow, comp, isa;
  int32_t is_asym, has_reduce, has_shuffle;
  /* byte offsets from the blob base (0 = absent) */
  uint64_t q_off, q_bytes, scale_off, scale_bytes, zp_off, zp_bytes, red_off, red_bytes, shuf_off, shuf_bytes;
  uint64_t dq_off, dq_bytes; /* DQ8_BNB: fp32 block maxima of the scale codes + the offset as the last float (bestla_storage.h:222-229) */
} nso_blob_info;

/* the 256 values of the DQ8_BNB code map (bestla_utils.h:794-...: the bitsandbytes dynamic map printed with five decimals) */
const float* nso_dq8_lut(void);

uint64_t nso_core_id(int core);
int nso_core_attr(int core, int* ntile, int* packrow, int* ktile, int* comp, int* isa);

/* scalar dtype helpers (bestla_utils.h:116-229) */
uint16_t nso_f32_to_bf16(float v);
float nso_bf16_to_f32(uint16_t v);
uint16_t nso_f32_to_f16(float v);
float nso_f16_to_f32(uint16_t v);

/* kernel_ref.h:1608-1719  (src is the [K][N] matrix, i.e. already transposed to K-major) */
int nso_quantize_int_rowblock(const float* src, int8_t* dst, int row, int col, int ld_src, int ld_dst, float* scales,
                              int8_t* zero_points, int blocksize, uint32_t qtype);
/* kernel_ref.h:1801-1822 */
int nso_quantize_f4_rowblock(const float* src, int8_t* dst, int row, int col, int ld_src, int ld_dst, float* scales,
                             int blocksize, uint32_t f4type);
/* kernel_ref.h:1763-1799 (+ f8_mx_quantize :1721-1761, f8_to_fp32 :984-1002); stype = NSO_F8_E8M0 or NSO_F32.  The
 * shared exponent is floor(std::log2(float)): it follows the host libm exactly where the reference would. */
int nso_quantize_f8_rowblock(const float* src, int8_t* dst, int row, int col, int ld_src, int ld_dst, float* scales,
                             int blocksize, uint32_t f8type, uint32_t stype);
float nso_f8_to_f32(uint32_t f8type, int code);
int nso_f8_quantize(uint32_t f8type, uint32_t stype, float v, float scale);
float nso_f4_unpack(uint32_t f4type, int code);
int nso_f4_quantize(uint32_t f4type, float x);

/* kernel_ref.h:39-57 */
void nso_padding_interleave(const int8_t* src, int8_t* dst, int row, int col, int rowpad, int colpad, int src_step,
                            int dst_step, int ntile, int rowpack);
/* kernel_ref.h:155-365: compress `size` int8 codes into the bit planes of `qtype`; dst must hold nso_qbytes() */
void nso_compress(const int8_t* src, uint8_t* dst, size_t size, uint32_t qtype);
/* inverse (kernel_ref.h:420-526): planes -> signed codes (stored - 2^(b-1)); f4 -> raw 0..15 code */
void nso_decompress(const uint8_t* src, int8_t* dst, size_t size, uint32_t qtype);
size_t nso_qbytes(size_t elts, uint32_t qtype);

/* bestla_storage.h:697-859 + bestla_prologue_b.h:120-127,1011-1017: serialized size of the blob */
size_t nso_pack_size(int n, int k, int blocksize, uint32_t qtype, uint32_t stype, int asym, int core);
/* BTLAGemmPackB (bestla_gemm.cpp:401-422, prologue_b.h:378-398): q [K][N] (ld = ldq), scales/zp [ceil(K/blk)][N] */
int nso_pack_q(void* blob, const int8_t* q, int ldq, const float* scales, const int8_t* zps, int n, int k,
               int blocksize, uint32_t qtype, uint32_t stype, int asym, int core);
/* the same with GPTQ act-order group indices (g_idx[k] = group of input channel k; bestla_gemm.cpp:409-413 ->
 * setShuffleIndices, bestla_prologue_b.h:337-356).  q's rows are already in group-sorted order (the converter does
 * that, neural_speed/convert/common.py:667-681); the blob gains an int[K] section, and the GEMM gathers
 * A'[j] = A[indices[j]] first (nso_gemm_f64* honour it; nso_unpack_* return the stored, sorted-order weights, as
 * BTLAGemmUnPackB does). */
int nso_pack_q_gidx(void* blob, const int8_t* q, int ldq, const float* scales, const int8_t* zps, int n, int k,
                    int blocksize, uint32_t qtype, uint32_t stype, int asym, int core, const int* g_idx);
size_t nso_pack_size_gidx(int n, int k, int blocksize, uint32_t qtype, uint32_t stype, int asym, int core);
/* BTLAGemmQuantPackB (bestla_gemm.cpp:302-319): fp32 weight, [N][K] if is_trans (torch layout) else [K][N] */
int nso_quant_pack(void* blob, const float* w, int n, int k, int ldw, int blocksize, uint32_t qtype, uint32_t stype,
                   int asym, int core, int is_trans);
int nso_blob_parse(const void* blob, nso_blob_info* info);
/* BTLAGemmUnPackB (bestla_gemm.cpp:673-749): blob -> fp32 [K][N] (ld = ldb) */
int nso_unpack_fp32(const void* blob, float* out, int ldb);
/* blob -> canonical pieces: signed codes minus nothing (q [K][N]), scales fp32 [nblk][N], zp [nblk][N] (0 if sym) */
int nso_unpack_canonical(const void* blob, int8_t* q, float* scales, int8_t* zps);

/* comp-fp32 semantics (kernel_ref.h:2489-2531 generalised): C[m][n] = sum_k A[m][k] * W[k][n], W = unpacked fp32,
 * accumulated in fp64 — the parity target for every HIP GEMM/GEMV. */
int nso_gemm_f64(const float* a, int lda, const void* blob, double* c, int ldc, int m);
/* same but A first rounded to fp16 (what the HIP kernels feed the MFMA/dot units) — used to separate
 * activation-rounding error from kernel error in tests */
int nso_gemm_f64_a16(const float* a, int lda, const void* blob, double* c, int ldc, int m);
/* both of the above from ONE unpack of the blob: c = fp32 activations, c16 = activations rounded through fp16 */
int nso_gemm_f64_pair(const float* a, int lda, const void* blob, double* c, double* c16, int ldc, int m);
/* sequential-k fp32 accumulation exactly as gemv_4bit_fp32_fp32 does it (kernel_ref.h:2489-2531), threaded over N
 * with OpenMP; used as the timed CPU baseline ("port") */
int nso_gemv_f32(const float* a, int lda, const void* blob, float* c, int ldc, int m, int nthreads);

/* activation side of the int8 compute path: kernel_ref.h:1824-1883 */
int nso_quantize_fp_u8_colblock(int row, int col, const float* src, int ld_src, uint8_t* dst, int ld_dst,
                                float* scales, int ld_scale, uint8_t* zps, int blocksize, float* blkreduce);
/* int8-compute GEMM semantics (ut/bestla_gemm.cpp:159-190 ref_kblock_int8 / kernel_ref.h:2371-2429) */
int nso_gemm_u8s8_f32(const float* a, int lda, const void* blob, float* c, int ldc, int m);
/* the same sums read straight from the packed blob, OpenMP over the column tiles (the timed CPU-baseline port of the
 * reference's default int8-compute decode path, bestla_wrapper.h:643-688) */
int nso_gemv_u8s8_f32(const float* a, int lda, const void* blob, float* c, int ldc, int m, int nthreads);

/* epilogue helpers — bestla_common.hpp:121-215, ip_fusion_ffn.cpp */
float nso_gelu(float x);
float nso_silu(float x);

/* fused attention — bestla_fusion_attn_forward_ref, neural_speed/core/layers/mha_dense_wrapper.h:1371-1517, for
 * Q fp32 / K,V fp16 / dst fp32 in ATTN_FWD_LAYOUT_PLAIN with element strides.  `bf16_gemm` != 0 reproduces the
 * reference's default rounding of Q, K and P to bf16 (IS_BF16_GEMM, :1389-1394); 0 = its NE_ATTN_FLAG_PREFER_FP32
 * form; `mode` bit 1 (value 2) uses the reference's own exp (MHA_2ND_EXP: exp_ps_0_1, a second-order polynomial, relative
 * error up to 2e-3) instead of expf.  flags: 1 causal, 2 alibi8 (TANH30 is not part of forward_ref).  PARITY: PINNED
 * to the function itself — bestla_fusion_attn_forward_ref<float, fp16, fp16, float> compiled from the reference file
 * (oracle/Makefile attnref -> _ref/libattn_ref.so): with mode bit 1 set the restatement reproduces it to fp32 rounding on
 * every case of mha_dense_tests.cpp:43-62, :92-112 (plain / transposed K, causal, alibi, GQA, PREFER_FP32 or the bf16
 * default), tests/test_attention_oracle.py — and, in its semantics, to the reference's unfused attention graph executed
 * by ne_layers.c (oracle/_ref/libne_ref.so; tolerance = that graph's fp16-table soft_max).  The PRODUCT evaluates the
 * exact exp (mode bit 1 clear is its oracle); the distance between the two exps is measured in the same test. */
typedef struct nso_attn_args {
  const float* q;
  const uint16_t* k;
  const uint16_t* v;
  float* dst;
  float q_sc, k_sc, v_sc, dst_sc, qk_scale;
  uint32_t flags;
  int batch_size, head_num, heads_kv, head_size, sl_q, sl_kv;
  long long step_q_bs, step_q_head_num, step_q_sl;
  long long step_k_bs, step_k_head_num, step_k_sl, step_k_head_size;
  long long step_v_bs, step_v_head_num, step_v_sl;
  long long step_dst_bs, step_dst_head_num, step_dst_sl;
} nso_attn_args;
int nso_attn_ref(const nso_attn_args* a, int mode); /* mode: bit 0 bf16 GEMM rounding, bit 1 the reference's polynomial exp */

/* RoPE — ne_compute_forward_rope_f32, neural_speed/core/ne_layers.c:9243-9428, for contiguous fp32 tensors
 * [batch][seq][heads][head_size], modes 0 (adjacent pairs over the whole row) and 2 (NeoX halves inside n_dims-wide
 * blocks), ext_factor = 0 (no YaRN mix), no GLM / long-rope / shift.  theta is the reference's sequential fp32 product
 * (theta *= theta_scale per pair); the NeoX branch applies freq_scale twice, as the reference does (:9398 + :9207).
 * PINNED: bit-identical to the reference's own ne_compute_forward_rope_f32 run through its graph executor
 * (oracle/_ref/libne_ref.so = ne_layers.c compiled from where it lies, oracle/ne_ref_harness.c; tests/test_rope.py),
 * and checked against an fp64 closed form. */
int nso_rope_f32(const float* src, float* dst, int batch, int seq, int heads, int head_size, int n_past, int n_dims,
                 int mode, float freq_base, float freq_scale, float attn_factor);
/* the same with the YaRN extrapolation mix (rope_yarn / rope_yarn_ramp / ggml_rope_yarn_corr_dims,
 * ne_layers.c:9196-9231); ext_factor = 0 reduces to nso_rope_f32.  PINNED likewise (bit-identical). */
int nso_rope_f32_yarn(const float* src, float* dst, int batch, int seq, int heads, int head_size, int n_past, int n_dims,
                      int mode, float freq_base, float freq_scale, int n_orig_ctx, float ext_factor, float attn_factor,
                      float beta_fast, float beta_slow);
/* long-rope (mode 0x10, ne_layers.c:9349-9377): theta / factors[pair] through rope_yarn, cos / sin times scale_factor */
/* GLM branch (mode & 4; mode & 1 = skip): ne_layers.c:9317-9347.  n_padding[batch].  PINNED likewise (bit-identical). */
int nso_rope_f32_glm(const float* src, float* dst, int batch, int seq, int heads, int head_size, int n_past, int n_dims,
                     int mode, float freq_base, int prompt_size, const int* n_padding);
int nso_rope_f32_longrope(const float* src, float* dst, int batch, int seq, int heads, int head_size, int n_past, int n_dims,
                          float freq_base, float freq_scale, int n_orig_ctx, float ext_factor, float attn_factor,
                          float beta_fast, float beta_slow, const float* factors, float scale_factor);

#ifdef __cplusplus
}
#endif
#endif
