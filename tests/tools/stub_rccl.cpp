// tests/tools/stub_rccl.cpp — TEST INFRASTRUCTURE, never shipped and never loaded unless NS_TP_RCCL_LIB names it.
//
// RCCL refuses two ranks on one device, so a one-GPU box cannot run libns_hip.so's tensor-parallel layer (csrc/ns_tp.cpp)
// with more than one rank on the real library.  This file implements the slice of the RCCL ABI ns_tp.cpp binds
// (ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclGetErrorString, ncclAllReduce, ncclBroadcast, ncclAllToAll)
// over a POSIX shared-memory segment, so that world = 2 / 4 / 8 PROCESSES SHARING ONE GPU really exchange and sum data:
// unique-id hand-over, communicator set-up, routing between the peer-memory kernel and "RCCL", stream ordering, HIP-graph
// capture and replay, the host-pointer forms and glue/parallel_context_hip.cpp's bootstrap are exercised end to end.
//
// Every collective is three stream-ordered steps, all of them capturable:
//   hipMemcpyAsync  device -> pinned staging                  (memcpy node)
//   hipLaunchHostFunc: staging -> this rank's slot of the segment, barrier, combine the ranks' slots IN RANK ORDER into
//                      the pinned result, barrier             (host node; no HIP call inside)
//   hipMemcpyAsync  pinned result -> device                   (memcpy node)
// Sums are fp32 (or int32) additions in rank order 0, 1, ..., world - 1: deterministic and equal on every rank.
// A rank that waits longer than NS_STUB_RCCL_TIMEOUT_S (default 60) for the others poisons the segment: every later
// call on every rank returns ncclSystemError instead of hanging.
//
// Build (tests/test_gpu_tp_stub.py does it): g++ -O2 -fPIC -shared -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include
//        stub_rccl.cpp -L/opt/rocm/lib -lamdhip64 -lrt -o libns_stub_rccl.so
#include <fcntl.h>
#include <hip/hip_runtime_api.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <random>
#include <vector>

namespace {

constexpr int kSuccess = 0, kUnhandledHipError = 1, kSystemError = 2, kInvalidArgument = 4;
constexpr int kInt32 = 2, kFloat32 = 7, kSum = 0;
constexpr char kMagic[8] = {'N', 'S', 'S', 'T', 'U', 'B', '1', 0};
constexpr int kMaxWorld = 16;

struct Header {  // zero-filled by ftruncate: a valid initial state
  std::atomic<uint32_t> arrived;
  std::atomic<uint32_t> generation;
  std::atomic<uint32_t> poisoned;
  std::atomic<uint32_t> attached;
};

struct Comm;
struct Op {  // one enqueued collective: lives as long as the communicator (a captured graph replays it)
  Comm* comm;
  int kind;  // 0 all-reduce, 1 broadcast, 2 all-to-all
  int dtype, root;
  size_t count;
};

struct Comm {
  int rank = 0, world = 1;
  char name[64] = {0};
  uint8_t* base = nullptr;
  size_t total = 0, slot = 0;
  Header* hdr = nullptr;
  uint8_t* stage_in = nullptr;   // pinned
  uint8_t* stage_out = nullptr;  // pinned
  double timeout_s = 60.0;
  std::mutex mu;
  std::vector<Op*> ops;
  uint8_t* slot_of(int r) const { return base + 4096 + size_t(r) * slot; }
};

bool barrier(Comm* c) {
  Header* h = c->hdr;
  if (h->poisoned.load()) return false;
  const uint32_t g = h->generation.load();
  if (h->arrived.fetch_add(1) + 1 == uint32_t(c->world)) {
    h->arrived.store(0);
    h->generation.store(g + 1);
    return true;
  }
  const auto t0 = std::chrono::steady_clock::now();
  uint32_t spins = 0;
  while (h->generation.load() == g) {
    if (h->poisoned.load()) return false;
    if ((++spins & 1023u) == 0) {
      sched_yield();
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > c->timeout_s) {
        h->poisoned.store(1);
        fprintf(stderr, "stub_rccl: rank %d waited %.0f s for its peers: communicator poisoned\n", c->rank, c->timeout_s);
        return false;
      }
    }
  }
  return true;
}

void run_op(void* user) {
  Op* op = static_cast<Op*>(user);
  Comm* c = op->comm;
  const size_t esz = 4;
  if (op->kind == 0) {
    const size_t bytes = op->count * esz;
    memcpy(c->slot_of(c->rank), c->stage_in, bytes);
    if (!barrier(c)) return;
    if (op->dtype == kFloat32) {
      float* out = reinterpret_cast<float*>(c->stage_out);
      const float* s0 = reinterpret_cast<const float*>(c->slot_of(0));
      for (size_t i = 0; i < op->count; i++) out[i] = s0[i];
      for (int r = 1; r < c->world; r++) {
        const float* s = reinterpret_cast<const float*>(c->slot_of(r));
        for (size_t i = 0; i < op->count; i++) out[i] += s[i];
      }
    } else {
      int32_t* out = reinterpret_cast<int32_t*>(c->stage_out);
      memcpy(out, c->slot_of(0), bytes);
      for (int r = 1; r < c->world; r++) {
        const int32_t* s = reinterpret_cast<const int32_t*>(c->slot_of(r));
        for (size_t i = 0; i < op->count; i++) out[i] += s[i];
      }
    }
    barrier(c);
  } else if (op->kind == 1) {
    const size_t bytes = op->count * esz;
    if (c->rank == op->root) memcpy(c->slot_of(op->root), c->stage_in, bytes);
    if (!barrier(c)) return;
    memcpy(c->stage_out, c->slot_of(op->root), bytes);
    barrier(c);
  } else {
    const size_t chunk = op->count * esz;  // per peer
    memcpy(c->slot_of(c->rank), c->stage_in, chunk * size_t(c->world));
    if (!barrier(c)) return;
    for (int r = 0; r < c->world; r++) memcpy(c->stage_out + size_t(r) * chunk, c->slot_of(r) + size_t(c->rank) * chunk, chunk);
    barrier(c);
  }
}

int enqueue(Comm* c, int kind, const void* send, void* recv, size_t count, int dtype, int root, size_t in_elems, size_t out_elems,
            hipStream_t st) {
  if (!c || (dtype != kFloat32 && dtype != kInt32)) return kInvalidArgument;
  if (c->hdr->poisoned.load()) return kSystemError;
  if (in_elems * 4 > c->slot || out_elems * 4 > c->slot) {
    fprintf(stderr, "stub_rccl: %zu bytes exceed the slot (%zu; raise NS_STUB_RCCL_SLOT_MB)\n", in_elems * 4, c->slot);
    return kInvalidArgument;
  }
  if (count == 0) return kSuccess;
  Op* op = new Op{c, kind, dtype, root, count};
  {
    std::lock_guard<std::mutex> lk(c->mu);
    c->ops.push_back(op);
  }
  const bool sends = kind != 1 || c->rank == root;
  if (sends && hipMemcpyAsync(c->stage_in, send, in_elems * 4, hipMemcpyDeviceToHost, st) != hipSuccess) return kUnhandledHipError;
  if (hipLaunchHostFunc(st, run_op, op) != hipSuccess) return kUnhandledHipError;
  if (hipMemcpyAsync(recv, c->stage_out, out_elems * 4, hipMemcpyHostToDevice, st) != hipSuccess) return kUnhandledHipError;
  return kSuccess;
}

}  // namespace

extern "C" {

struct ncclUniqueId {
  char internal[128];
};
typedef Comm* ncclComm_t;

int ncclGetUniqueId(ncclUniqueId* id) {
  if (!id) return kInvalidArgument;
  memset(id, 0, sizeof(*id));
  memcpy(id->internal, kMagic, 8);
  std::random_device rd;
  snprintf(id->internal + 8, 56, "/ns_stub_rccl_%d_%08x%08x", int(getpid()), unsigned(rd()), unsigned(rd()));
  return kSuccess;
}

int ncclCommInitRank(ncclComm_t* out, int world, ncclUniqueId id, int rank) {
  if (!out || world < 1 || world > kMaxWorld || rank < 0 || rank >= world || memcmp(id.internal, kMagic, 8) != 0) return kInvalidArgument;
  Comm* c = new Comm;
  c->rank = rank, c->world = world;
  memcpy(c->name, id.internal + 8, 56);
  const char* mb = getenv("NS_STUB_RCCL_SLOT_MB");
  c->slot = size_t(mb ? atoi(mb) : 48) << 20;
  const char* to = getenv("NS_STUB_RCCL_TIMEOUT_S");
  if (to) c->timeout_s = atof(to);
  c->total = 4096 + size_t(world) * c->slot;
  const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, off_t(c->total)) != 0) {
    perror("stub_rccl: shm_open / ftruncate");
    if (fd >= 0) close(fd);
    delete c;
    return kSystemError;
  }
  void* p = mmap(nullptr, c->total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) {
    perror("stub_rccl: mmap");
    delete c;
    return kSystemError;
  }
  c->base = static_cast<uint8_t*>(p);
  c->hdr = reinterpret_cast<Header*>(p);
  if (hipHostMalloc(reinterpret_cast<void**>(&c->stage_in), c->slot, hipHostMallocDefault) != hipSuccess ||
      hipHostMalloc(reinterpret_cast<void**>(&c->stage_out), c->slot, hipHostMallocDefault) != hipSuccess) {
    fprintf(stderr, "stub_rccl: pinned staging allocation failed\n");
    munmap(p, c->total);
    delete c;
    return kUnhandledHipError;
  }
  c->hdr->attached.fetch_add(1);
  if (!barrier(c)) {  // like ncclCommInitRank: returns once every rank of the communicator has called it
    (void)hipHostFree(c->stage_in);
    (void)hipHostFree(c->stage_out);
    munmap(p, c->total);
    delete c;
    return kSystemError;
  }
  *out = c;
  return kSuccess;
}

int ncclCommDestroy(ncclComm_t c) {
  if (!c) return kSuccess;
  const bool last = c->hdr->attached.fetch_sub(1) == 1;
  if (last) shm_unlink(c->name);
  (void)hipHostFree(c->stage_in);
  (void)hipHostFree(c->stage_out);
  munmap(c->base, c->total);
  for (Op* op : c->ops) delete op;
  delete c;
  return kSuccess;
}

const char* ncclGetErrorString(int rc) {
  switch (rc) {
    case kSuccess: return "no error";
    case kUnhandledHipError: return "unhandled HIP error (stub)";
    case kSystemError: return "unhandled system error (stub: a peer did not arrive)";
    case kInvalidArgument: return "invalid argument (stub)";
    default: return "unknown result code (stub)";
  }
}

int ncclAllReduce(const void* send, void* recv, size_t count, int dtype, int op, ncclComm_t c, hipStream_t st) {
  if (op != kSum) return kInvalidArgument;
  return enqueue(c, 0, send, recv, count, dtype, 0, count, count, st);
}

int ncclBroadcast(const void* send, void* recv, size_t count, int dtype, int root, ncclComm_t c, hipStream_t st) {
  if (!c || root < 0 || root >= c->world) return kInvalidArgument;
  return enqueue(c, 1, send, recv, count, dtype, root, count, count, st);
}

int ncclAllToAll(const void* send, void* recv, size_t count, int dtype, ncclComm_t c, hipStream_t st) {
  if (!c) return kInvalidArgument;
  return enqueue(c, 2, send, recv, count, dtype, 0, count * size_t(c->world), count * size_t(c->world), st);
}

}  // extern "C"
