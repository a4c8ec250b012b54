// glue/bestla_gemm_hip.cpp — the C++-linkage quantizer / packer functions of neural-speed
// (/root/reference/neural_speed/core/layers/bestla_gemm.h:38-55, called from models/model_utils/quant_utils.cpp:245-247,
// :343-347, model_files.h:1546-1560 and the pybind statics) forwarded to libns_hip.so's C twins.  Compiled against the
// REFERENCE's own header (tests/test_glue.py does that here); a maintainer adds this file to the ne_layers target in
// place of core/layers/bestla_gemm.cpp and links libns_hip.so.
#include "layers/bestla_gemm.h"

#include "ns_bestla.h"

size_t BTLAGemmPackBSize(size_t N, size_t K, size_t BlkSize, BTLA_DTYPE QuantType, BTLA_DTYPE ScaleDtype, bool isAsym,
                         ne_comp_type CompType, int* shuffle_indice) {
  return ns_BTLAGemmPackBSize(N, K, BlkSize, static_cast<uint32_t>(QuantType), static_cast<uint32_t>(ScaleDtype), isAsym,
                              static_cast<int>(CompType), shuffle_indice);
}

bool BTLAGemmQuantPackB(void* PackedBuf, const float* FpData, size_t N, size_t K, size_t ldb, size_t BlkSize,
                        BTLA_DTYPE QuantType, BTLA_DTYPE ScaleDtype, bool isAsym, ne_comp_type CompType, bool isTrans,
                        void* ThreadPool) {
  return ns_BTLAGemmQuantPackB(PackedBuf, FpData, N, K, ldb, BlkSize, static_cast<uint32_t>(QuantType),
                               static_cast<uint32_t>(ScaleDtype), isAsym, static_cast<int>(CompType), isTrans, ThreadPool);
}

bool BTLAGemmPackB(void* PackedBuf, const int8_t* QData, const float* Scales, const int8_t* Zp, size_t N, size_t K,
                   size_t ldb, size_t BlkSize, BTLA_DTYPE QuantType, BTLA_DTYPE ScaleDtype, bool isAsym,
                   ne_comp_type CompType, int* shuffle_indice, void* ThreadPool) {
  return ns_BTLAGemmPackB(PackedBuf, QData, Scales, Zp, N, K, ldb, BlkSize, static_cast<uint32_t>(QuantType),
                          static_cast<uint32_t>(ScaleDtype), isAsym, static_cast<int>(CompType), shuffle_indice, ThreadPool);
}

bool BTLAGemmUnPackB(float* FpData, const void* PackedBuf, size_t N, size_t K, size_t ldb, void* ThreadPool) {
  return ns_BTLAGemmUnPackB(FpData, PackedBuf, N, K, ldb, ThreadPool);
}

// BTLAGemmBatchDriver (bestla_gemm.cpp:508-624): BatchN packed GEMMs C_i = A_i * B_i — what inner_product.cpp:28-36
// wraps for one problem; the blob decides the kernel
bool BTLAGemmBatchDriver(const size_t M, const size_t N, const size_t K, const size_t BatchN,
                         const BTLA_GEMM_DATA_PACKED_PARAMS* DataParams, int8_t* WorkSpace, void* ThreadPool) {
  (void)ThreadPool;
  for (size_t i = 0; i < BatchN; i++)
    bestla_f32f32_forward(const_cast<float*>(DataParams[i].A), const_cast<void*>(DataParams[i].B), DataParams[i].C,
                          static_cast<int>(M), static_cast<int>(N), static_cast<int>(K), DataParams[i].lda, DataParams[i].ldc,
                          WorkSpace);
  return true;
}

bool BTLALayerNorm(size_t norm_count, size_t norm_size, bool isrms, float epsilon, const float* FpIn, float* FpOut,
                   void* ThreadPool) {
  (void)ThreadPool;
  bestla_layernormalization(static_cast<int>(norm_count), static_cast<int>(norm_size), isrms, epsilon, FpIn, FpOut);
  return true;
}
