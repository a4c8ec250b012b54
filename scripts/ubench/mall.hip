// mall.hip — does a prefetch stream that touches the NEXT weight buffer (filling Infinity Cache / L2) while the
// current one is consumed shorten the chain?  Compares: cold reads, same-buffer re-reads (cache resident), and a
// two-stream pipeline {consume(buf i) || prefetch(buf i+1)} with event dependencies, as a hipGraph.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <bool NT>
__global__ void consume(const u4* __restrict__ p, uint32_t* out, size_t n16) {
  size_t base = size_t(blockIdx.x) * blockDim.x * 4 + threadIdx.x;
  u4 v[4];
#pragma unroll
  for (int i = 0; i < 4; i++) { size_t idx = base + size_t(i) * blockDim.x; v[i] = idx < n16 ? (NT ? __builtin_nontemporal_load(p + idx) : p[idx]) : u4{0,0,0,0}; }
  uint32_t acc = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) acc ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ void prefetch(const u4* __restrict__ p, uint32_t* out, size_t n16) {  // few workgroups, grid-stride, plain loads
  const size_t stride = size_t(gridDim.x) * blockDim.x;
  uint32_t acc = 0;
  for (size_t idx = size_t(blockIdx.x) * blockDim.x + threadIdx.x; idx < n16; idx += 4 * stride) {
    u4 v[4];
#pragma unroll
    for (int i = 0; i < 4; i++) v[i] = (idx + i * stride < n16) ? p[idx + i * stride] : u4{0,0,0,0};
#pragma unroll
    for (int i = 0; i < 4; i++) acc ^= v[i].x ^ v[i].w;
  }
  if (acc == 0x12345678u) out[1] = acc;
}

int main(int argc, char** argv) {
  const size_t bytes = argc > 1 ? size_t(atof(argv[1]) * 1e6) : size_t(50.8e6);
  const int pf_grid = argc > 2 ? atoi(argv[2]) : 256;
  const int NBUF = 24;
  std::vector<u4*> bufs(NBUF);
  for (auto& b : bufs) { CK(hipMalloc(&b, bytes)); CK(hipMemset(b, 0x5a, bytes)); }
  uint32_t* out; CK(hipMalloc(&out, 8));
  hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t n16 = bytes / 16;
  const dim3 g((n16 + 1023) / 1024), b(256);
  auto run = [&](const char* name, hipGraphExec_t ge) {
    for (int i = 0; i < 3; i++) CK(hipGraphLaunch(ge, s1));
    CK(hipStreamSynchronize(s1));
    CK(hipEventRecord(e0, s1));
    const int reps = 10;
    for (int i = 0; i < reps; i++) CK(hipGraphLaunch(ge, s1));
    CK(hipEventRecord(e1, s1)); CK(hipStreamSynchronize(s1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double us = ms * 1e3 / (reps * NBUF);
    printf("%-52s %8.2f us/buffer  %7.1f GB/s\n", name, us, bytes / us / 1e3);
  };
  auto capture = [&](auto body) { hipGraph_t gr; hipGraphExec_t ge; CK(hipStreamBeginCapture(s1, hipStreamCaptureModeGlobal)); body(); CK(hipStreamEndCapture(s1, &gr)); CK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0)); return ge; };
  printf("buffer %.1f MB, prefetch grid %d\n", bytes / 1e6, pf_grid);
  run("cold: consume<nt>(buf i), 24 distinct buffers", capture([&] { for (int i = 0; i < NBUF; i++) hipLaunchKernelGGL(consume<true>, g, b, 0, s1, bufs[i], out, n16); }));
  run("cold: consume<plain>", capture([&] { for (int i = 0; i < NBUF; i++) hipLaunchKernelGGL(consume<false>, g, b, 0, s1, bufs[i], out, n16); }));
  run("warm: consume<plain>(same buffer)", capture([&] { for (int i = 0; i < NBUF; i++) hipLaunchKernelGGL(consume<false>, g, b, 0, s1, bufs[0], out, n16); }));
  run("warm: consume<nt>(same buffer)", capture([&] { for (int i = 0; i < NBUF; i++) hipLaunchKernelGGL(consume<true>, g, b, 0, s1, bufs[0], out, n16); }));
  run("serial: prefetch(i) then consume<plain>(i)", capture([&] { for (int i = 0; i < NBUF; i++) { hipLaunchKernelGGL(prefetch, dim3(pf_grid), b, 0, s1, bufs[i], out, n16); hipLaunchKernelGGL(consume<false>, g, b, 0, s1, bufs[i], out, n16); } }));
  for (int nt = 0; nt < 2; nt++) {
    char nm[96]; snprintf(nm, 96, "pipeline: consume<%s>(i) || prefetch(i+1)", nt ? "nt" : "plain");
    run(nm, capture([&] {
      std::vector<hipEvent_t> evp(NBUF + 1), evc(NBUF + 1);
      for (size_t q = 0; q < evp.size(); q++) CK(hipEventCreateWithFlags(&evp[q], hipEventDisableTiming));
      for (size_t q = 0; q < evc.size(); q++) CK(hipEventCreateWithFlags(&evc[q], hipEventDisableTiming));
      CK(hipEventRecord(evc[0], s1)); CK(hipStreamWaitEvent(s2, evc[0], 0));   // fork
      hipLaunchKernelGGL(prefetch, dim3(pf_grid), b, 0, s2, bufs[0], out, n16);
      CK(hipEventRecord(evp[0], s2));
      for (int i = 0; i < NBUF; i++) {
        CK(hipStreamWaitEvent(s1, evp[i], 0));            // consume(i) needs prefetch(i) done
        if (i + 1 < NBUF) { hipLaunchKernelGGL(prefetch, dim3(pf_grid), b, 0, s2, bufs[i + 1], out, n16); CK(hipEventRecord(evp[i + 1], s2)); }
        if (nt) hipLaunchKernelGGL(consume<true>, g, b, 0, s1, bufs[i], out, n16); else hipLaunchKernelGGL(consume<false>, g, b, 0, s1, bufs[i], out, n16);
        CK(hipEventRecord(evc[i + 1], s1));
        if (i + 2 < NBUF) CK(hipStreamWaitEvent(s2, evc[i + 1], 0));  // throttle: prefetch(i+2) after consume(i)
      }
      CK(hipStreamWaitEvent(s1, evp[NBUF - 1], 0));       // join
    }));
  }
  return 0;
}
