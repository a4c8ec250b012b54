// ns_i8g2.h — interface between ns_i8ref.hip (host logic of the int8-reference mode) and the translation units that hold the
// instantiations of its second matrix-core kernel (ns_i8g2.hip, compiled once per container kind and scales-per-record count:
// sixty kernels of this size in one translation unit took eight minutes to compile)
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace ns {

struct I8RefParams {
  const uint8_t* codes;
  const uint8_t* scales;
  const int8_t* zps;
  uint32_t qstride, sstride, zstride;
  int ksteps, kstep_len, nj;  // 128 / 4 (4-bit containers) or 64 / 2 (8-bit)
  int chunk_steps;            // k-steps of A staged in LDS at a time
  int sps, srows, srow_mul, srow_shift;
  uint32_t scale_dt;
  int asym;
  int blocksize, nblk;  // k-block of BOTH quantizations (weights and activations), blocks per row
  int n, k, m;
  const uint8_t* aq;    // [m][k] u8 activation codes
  const float* ascale;  // [m][nblk]
  const uint8_t* azp;   // [m][nblk]
  float* c;
  _Float16* c16;
  int ldc;
  int epilogue;
  const float* d;
  int ldd;
};

struct I8Gemm2Params {
  I8RefParams b;
  const uint8_t* pa;  // i8prep_kernel's output
  int nsl;            // slices per row of it
};

int i8_tile_forced();  // ns_i8ref.hip: the "i8_tile" tuning value (0 = by problem size)

// C = int8-reference product on i8mfma2_kernel for nibble (n) / byte (b) containers with 4 / 2 / 1 scales per k-step record
hipError_t launch_i8g2_n4(int sdt, bool asym, int m, int ntiles, hipStream_t st, const I8Gemm2Params& p);
hipError_t launch_i8g2_n2(int sdt, bool asym, int m, int ntiles, hipStream_t st, const I8Gemm2Params& p);
hipError_t launch_i8g2_n1(int sdt, bool asym, int m, int ntiles, hipStream_t st, const I8Gemm2Params& p);
hipError_t launch_i8g2_b2(int sdt, bool asym, int m, int ntiles, hipStream_t st, const I8Gemm2Params& p);
hipError_t launch_i8g2_b1(int sdt, bool asym, int m, int ntiles, hipStream_t st, const I8Gemm2Params& p);

}  // namespace ns
