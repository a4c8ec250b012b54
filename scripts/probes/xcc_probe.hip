// Which XCC does workgroup i of a dispatch run on?  (s_getreg_b32 HW_REG_XCC_ID.)  Prints, for several grid shapes, plain launches and a replayed
// graph, how many workgroups satisfy xcc == linear_id % 8.   hipcc --offload-arch=gfx950 -O2 scripts/probes/xcc_probe.hip -o /tmp/xcc_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(unsigned* out) {
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  if (threadIdx.x == 0) out[lin] = x;
}
__global__ void busy(float* p, int n) {
  float v = p[threadIdx.x];
  for (int i = 0; i < n; i++) v = v * 1.0001f + 0.5f;
  p[threadIdx.x] = v;
}
int main() {
  unsigned* d;
  float* f;
  hipMalloc(&d, 1 << 20);
  hipMalloc(&f, 4096);
  hipStream_t st;
  hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  auto report = [&](const char* what, dim3 g) {
    const unsigned n = g.x * g.y * g.z;
    std::vector<unsigned> h(n);
    hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
    unsigned ok = 0, raw = 0;
    for (unsigned i = 0; i < n; i++) ok += (h[i] & 15u) == i % 8, raw |= h[i];
    printf("%-34s grid %3u x %2u x %u: %u of %u workgroups on xcc == id %% 8 (or of all register values 0x%x; first 10:", what, g.x, g.y, g.z, ok, n, raw);
    for (unsigned i = 0; i < 10 && i < n; i++) printf(" %u", h[i] & 15u);
    printf(")\n");
  };
  for (dim3 g : {dim3(32, 8, 1), dim3(32, 12, 1), dim3(8, 16, 1), dim3(40, 8, 1), dim3(256, 1, 1), dim3(31, 8, 1)}) {
    hipMemsetAsync(d, 0xff, 1 << 20, st);
    hipLaunchKernelGGL(probe, g, dim3(256), 0, st, d);
    hipStreamSynchronize(st);
    report("plain launch", g);
    hipMemsetAsync(d, 0xff, 1 << 20, st);
    hipLaunchKernelGGL(busy, dim3(300), dim3(256), 0, st, f, 2000);  // something else in front: does the next dispatch start where this one ended?
    hipLaunchKernelGGL(probe, g, dim3(256), 0, st, d);
    hipStreamSynchronize(st);
    report("behind a 300-workgroup launch", g);
  }
  hipGraph_t gr;
  hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
  hipLaunchKernelGGL(busy, dim3(301), dim3(256), 0, st, f, 100);
  hipLaunchKernelGGL(probe, dim3(32, 8, 1), dim3(256), 0, st, d);
  hipStreamEndCapture(st, &gr);
  hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0);
  for (int r = 0; r < 3; r++) {
    hipMemsetAsync(d, 0xff, 1 << 20, st);
    hipGraphLaunch(ge, st);
    hipStreamSynchronize(st);
    report("graph replay", dim3(32, 8, 1));
  }
  return 0;
}
