#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(const int* addr, short* out) {
  __shared__ __attribute__((aligned(16))) short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
  __syncthreads();
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3)))*)((__attribute__((address_space(3))) char*)lds + addr[threadIdx.x]));
  for (int j = 0; j < 4; j++) out[threadIdx.x * 4 + j] = v[j];
}
int main() {
  int ha[64]; short ho[256];
  // lane 4r+c of each 16-lane group g: row r of a [4][16] block, cols 4c..; row stride 64 elements, group g at cols 16 g
  for (int l = 0; l < 64; l++) { int g = l >> 4, i = l & 15, r = i >> 2, c = i & 3; ha[l] = ((r * 64) + 16 * g + 4 * c) * 2; }
  int* da; short* dout; hipMalloc(&da, sizeof ha); hipMalloc(&dout, sizeof ho);
  hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice);
  k<<<1, 64>>>(da, dout); hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; l++) printf("lane %2d: %4d %4d %4d %4d\n", l, ho[4*l], ho[4*l+1], ho[4*l+2], ho[4*l+3]);
}
