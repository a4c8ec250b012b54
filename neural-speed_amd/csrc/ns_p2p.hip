// ns_p2p.hip — one-shot all-reduce over peer-mapped HBM for the tensor-parallel decode step   (latency-bound)
//
// What it replaces: `shm_all_reduce` (/root/reference/neural_speed/core/shared_memory_ccl.hpp:100-139), the reference's
// own small-message fast path beside oneCCL (`parallel_context.cpp:47-58` picks it for the decode-sized buffers): every
// rank copies its vector into a shared segment, raises a flag, waits for the others' flags and sums all copies.  The
// MI355X form of the same idea: the "shared segment" is one uncached allocation per GPU, mapped
// into every peer process through HIP IPC; a peer's flag store and the 16-32 KB payload reads travel over xGMI
// point-to-point links, nothing goes through the host.  One kernel launch per all-reduce, capturable in a HIP graph:
// the sequence number lives in device memory, so a replayed graph keeps counting.
//
//   rank r, call number seq (parity par = seq & 1):
//     1. copy the input into own.data[par]                      (write-through: the allocation is uncached)
//     2. system-scope release store of seq into peer.flags[par][r] of every peer
//     3. wait until own.flags[par][p] == seq for every peer p   (bounded: a timeout raises a sticky error word)
//     4. out[i] = sum over ranks 0..world-1, in rank order, of rank.data[par][i]   (same order everywhere: every rank
//        ends with bit-identical sums, as the reference's in-place fp32 reduce does on one shared buffer)
//   data[par] is rewritten at call seq+2, which a rank can only reach after every peer has posted seq+1, i.e. after
//   every peer has finished reading seq: two slots and one flag exchange per call are enough, no second barrier.
//
// RCCL stays the path for anything larger than the slot (prefill-sized buffers) and the fallback when IPC is not
// available; neural-speed_amd/parallel.py decides.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/ns_bestla.h"
#include "ns_common.h"

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "ns_p2p.hip: the fence-free hand-off relies on gfx94x/gfx95x lowering system-scope accesses to sc0 sc1"
#endif

namespace ns {
namespace {

constexpr int kP2PMaxWorld = 16;
constexpr size_t kP2PDataOff = 4096;  // header page: counter, error word, flags

struct P2PHeader {
  uint32_t counter;  // calls completed by the OWNER of this allocation (only its own kernels touch it)
  uint32_t error;    // sticky: a flag wait of the owner timed out
  uint32_t pad[14];
  uint32_t flags[2][kP2PMaxWorld];  // flags[parity][source rank] = sequence number the source has published
};
static_assert(sizeof(P2PHeader) <= kP2PDataOff, "header page");

struct P2PParams {
  unsigned char* peers[kP2PMaxWorld];  // base of every rank's allocation as mapped in THIS process (own included)
  int rank, world;
  unsigned long long slot_bytes;
  float* buf;
  unsigned int n;
  unsigned long long timeout_ticks;  // 100 MHz wall clock
  uint32_t* host_error;              // pinned host word (device-mapped): raised with the sticky error below
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(1024) void p2p_allreduce_kernel(const P2PParams p) {
  __shared__ uint32_t s_seq;
  __shared__ int s_fail;
  const int tid = threadIdx.x;
  P2PHeader* self = reinterpret_cast<P2PHeader*>(p.peers[p.rank]);
  if (tid == 0) {
    s_seq = __hip_atomic_load(&self->counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) + 1u;
    // once a wait has timed out the context is dead: later calls do not wait again (no 10-second stalls per call)
    s_fail = __hip_atomic_load(&self->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u;
  }
  __syncthreads();
  const uint32_t seq = s_seq, par = seq & 1u;
  const size_t slot = kP2PDataOff + size_t(par) * p.slot_bytes;
  const unsigned int n4 = p.n >> 2;
  const bool dead = s_fail != 0;  // read before any lane can raise s_fail below

  // 1. publish: volatile stores are system-scope write-through (and the allocation is uncached anyway)
  {
    volatile f32x4* mine4 = reinterpret_cast<volatile f32x4*>(p.peers[p.rank] + slot);
    const f32x4* in4 = reinterpret_cast<const f32x4*>(p.buf);
    for (unsigned int i = tid; i < n4; i += blockDim.x) mine4[i] = in4[i];
    volatile float* mine = reinterpret_cast<volatile float*>(p.peers[p.rank] + slot);
    for (unsigned int i = (n4 << 2) + tid; i < p.n; i += blockDim.x) mine[i] = p.buf[i];
  }
  // No release / acquire fences anywhere: a system-scope fence writes back (and invalidates) the whole L2 of the XCD,
  // which costs microseconds per call.  Every access to a segment is itself system-scope (sc0 sc1: write-through
  // stores, cache-bypassing loads, and the segments are uncached memory), so "payload stores drained (vmcnt 0) ->
  // barrier -> flag store" on the producer and "flag seen -> barrier -> payload loads" on the consumer are ordered
  // by the waits alone (the drained-sc1 hand-off of MI355X_MICROARCH.md).
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // 2. + 3. one lane per peer: raise my flag over there, then wait for that peer's flag here
  if (tid < p.world && tid != p.rank) {
    P2PHeader* peer = reinterpret_cast<P2PHeader*>(p.peers[tid]);
    __hip_atomic_store(&peer->flags[par][p.rank], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long t0 = wall_clock64();
    while (!dead && __hip_atomic_load(&self->flags[par][tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
      if (wall_clock64() - t0 > p.timeout_ticks) {
        s_fail = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (s_fail && tid == 0) {
    __hip_atomic_store(&self->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(p.host_error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // 4. sum every rank's copy in rank order (own copy through the same path: identical arithmetic on every rank)
  {
    f32x4* out4 = reinterpret_cast<f32x4*>(p.buf);
    for (unsigned int i = tid; i < n4; i += blockDim.x) {
      f32x4 v[kP2PMaxWorld];
#pragma unroll
      for (int r = 0; r < kP2PMaxWorld; r++)
        if (r < p.world) v[r] = reinterpret_cast<const volatile f32x4*>(p.peers[r] + slot)[i];
      f32x4 acc = v[0];
#pragma unroll
      for (int r = 1; r < kP2PMaxWorld; r++)
        if (r < p.world) acc += v[r];
      out4[i] = acc;
    }
    for (unsigned int i = (n4 << 2) + tid; i < p.n; i += blockDim.x) {
      float acc = reinterpret_cast<const volatile float*>(p.peers[0] + slot)[i];
      for (int r = 1; r < p.world; r++) acc += reinterpret_cast<const volatile float*>(p.peers[r] + slot)[i];
      p.buf[i] = acc;
    }
  }
  if (tid == 0) __hip_atomic_store(&self->counter, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

bool p2p_ok(hipError_t e, const char* what) {
  if (e == hipSuccess) return true;
  set_error(std::string(what) + ": " + hipGetErrorString(e));
  (void)hipGetLastError();
  return false;
}

}  // namespace
}  // namespace ns

struct ns_p2p {
  int rank = 0, world = 0, device = 0;
  size_t max_bytes = 0, slot_bytes = 0, total = 0;
  unsigned char* peers[ns::kP2PMaxWorld] = {};
  bool connected = false;
  unsigned long long timeout_ticks = 0;
  uint32_t* host_error = nullptr;  // pinned + mapped: the kernel raises it, the launcher reads it without a sync
};

using namespace ns;

static_assert(sizeof(hipIpcMemHandle_t) == NS_P2P_HANDLE_BYTES, "handle size is part of the C ABI");

extern "C" {

ns_p2p* ns_hip_p2p_create(int rank, int world, size_t max_bytes, void* handle_out) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
    (void)hipGetLastError();
    set_error("p2p: no HIP device visible");
    return nullptr;
  }
  if (world < 2 || world > kP2PMaxWorld || rank < 0 || rank >= world || !max_bytes || !handle_out) {
    set_error("p2p: need 2 <= world <= 16, 0 <= rank < world, a slot size and a handle buffer");
    return nullptr;
  }
  ns_p2p* c = new ns_p2p;
  c->rank = rank, c->world = world, c->max_bytes = max_bytes;
  c->slot_bytes = (max_bytes + 4095) & ~size_t(4095);
  c->total = kP2PDataOff + 2 * c->slot_bytes;
  (void)hipGetDevice(&c->device);
  const char* t = getenv("NS_P2P_TIMEOUT_MS");
  const unsigned long long ms = t && atoll(t) > 0 ? (unsigned long long)atoll(t) : 10000ull;
  c->timeout_ticks = ms * 100000ull;  // 100 MHz
  void* base = nullptr;
  // uncached (fine-grained) device memory: peers' stores and loads are visible without cache maintenance
  if (!p2p_ok(hipExtMallocWithFlags(&base, c->total, hipDeviceMallocUncached), "p2p: hipExtMallocWithFlags")) {
    delete c;
    return nullptr;
  }
  c->peers[rank] = static_cast<unsigned char*>(base);
  hipIpcMemHandle_t h;
  void* herr = nullptr;
  if (!p2p_ok(hipHostMalloc(&herr, 64, hipHostMallocMapped), "p2p: hipHostMalloc") ||
      !p2p_ok(hipMemset(base, 0, c->total), "p2p: hipMemset") || !p2p_ok(hipDeviceSynchronize(), "p2p: synchronize") ||
      !p2p_ok(hipIpcGetMemHandle(&h, base), "p2p: hipIpcGetMemHandle")) {
    if (herr) (void)hipHostFree(herr);
    (void)hipFree(base);
    delete c;
    return nullptr;
  }
  c->host_error = static_cast<uint32_t*>(herr);
  *c->host_error = 0;
  memcpy(handle_out, &h, sizeof(h));
  return c;
}

int ns_hip_p2p_connect(ns_p2p* c, const void* all_handles) {
  if (!c || !all_handles || c->connected) {
    set_error("p2p connect: bad argument");
    return -1;
  }
  const unsigned char* hs = static_cast<const unsigned char*>(all_handles);
  for (int r = 0; r < c->world; r++) {
    if (r == c->rank) continue;
    hipIpcMemHandle_t h;
    memcpy(&h, hs + size_t(r) * sizeof(h), sizeof(h));
    void* p = nullptr;
    if (!p2p_ok(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess), "p2p: hipIpcOpenMemHandle")) {
      for (int q = 0; q < r; q++)
        if (q != c->rank && c->peers[q]) (void)hipIpcCloseMemHandle(c->peers[q]), c->peers[q] = nullptr;
      return -1;
    }
    c->peers[r] = static_cast<unsigned char*>(p);
  }
  c->connected = true;
  return 0;
}

int ns_hip_p2p_all_reduce_f32(ns_p2p* c, float* dBuf, size_t n, void* stream) {
  if (!c || !c->connected || (n && !dBuf)) {
    set_error("p2p all-reduce: not connected or null buffer");
    return -1;
  }
  if (n * sizeof(float) > c->max_bytes || (reinterpret_cast<uintptr_t>(dBuf) & 15)) {
    set_error("p2p all-reduce: buffer larger than the slot (use RCCL) or not 16-byte aligned");
    return -1;
  }
  // A flag wait of an EARLIER call timed out (a peer died or never launched): the sums since then included stale
  // peer slots.  Refuse from here on so the caller notices and falls back to RCCL; ns_hip_p2p_error() is the
  // synchronous form of the same check.
  if (__atomic_load_n(c->host_error, __ATOMIC_RELAXED)) {
    set_error("p2p all-reduce: an earlier call timed out waiting for a peer; the context is dead");
    return -2;
  }
  P2PParams p;
  p.host_error = c->host_error;
  for (int r = 0; r < kP2PMaxWorld; r++) p.peers[r] = r < c->world ? c->peers[r] : nullptr;
  p.rank = c->rank, p.world = c->world, p.slot_bytes = c->slot_bytes, p.buf = dBuf, p.n = (unsigned int)n;
  p.timeout_ticks = c->timeout_ticks;
  hipLaunchKernelGGL(p2p_allreduce_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, p);
  return p2p_ok(hipGetLastError(), "p2p all-reduce launch") ? 0 : -1;
}

int ns_hip_p2p_error(ns_p2p* c) {
  if (!c || !c->peers[c->rank]) return -1;
  uint32_t hdr[2] = {0, 0};
  if (!p2p_ok(hipMemcpy(hdr, c->peers[c->rank], sizeof(hdr), hipMemcpyDeviceToHost), "p2p: read status")) return -1;
  return int(hdr[1]);
}

void ns_hip_p2p_disconnect(ns_p2p* c) {
  if (!c) return;
  for (int r = 0; r < c->world; r++)
    if (r != c->rank && c->peers[r]) (void)hipIpcCloseMemHandle(c->peers[r]), c->peers[r] = nullptr;
  c->connected = false;
}

void ns_hip_p2p_destroy(ns_p2p* c) {
  if (!c) return;
  ns_hip_p2p_disconnect(c);
  if (c->peers[c->rank]) (void)hipFree(c->peers[c->rank]);
  if (c->host_error) (void)hipHostFree(c->host_error);
  delete c;
}

}  // extern "C"
