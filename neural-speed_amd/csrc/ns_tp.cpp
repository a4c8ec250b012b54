// ns_tp.cpp — tensor-parallel communication layer of libns_hip.so: the MI355X replacement of neural-speed's
// parallel_class (/root/reference/neural_speed/core/parallel_context.h:40-47, parallel_context.cpp:19-160), native
// C ABI, no Python, no torch.  The reference bootstraps oneCCL over MPI with one process per CPU socket; here it is one
// process per GPU and RCCL over the xGMI mesh.  RCCL is dlopen()ed at ns_tp_init, so the library loads (and every
// non-TP entry point works) on a box without it.
//
//   reference (host fp32 buffers)                  here (DEVICE fp32 buffers + stream, asynchronous)
//   init_parallel_context()                        ns_tp_unique_id() on rank 0 + ns_tp_init(rank, world, id)
//   get_tp_size / get_tp_rank / is_master          ns_tp_size / ns_tp_rank / ns_tp_is_master
//   barrier                                        ns_tp_barrier          (4-byte all-reduce, ordered on the stream)
//   broadcast(buf, count)        root 0            ns_tp_broadcast
//   alltoall(send, recv, count)                    ns_tp_alltoall         (count elements per peer, as ccl::alltoall)
//   reduce_add(send, recv, count)  fp32 sum        ns_tp_reduce_add       (in place when send == recv)
//
// Decode-sized buffers can additionally go through the one-shot peer-memory kernel (ns_p2p.hip), which the caller
// sets up with ns_hip_p2p_*; ns_tp_reduce_add routes to it when ns_tp_attach_p2p() was called and the buffer fits —
// the same split the reference makes between shm_all_reduce and oneCCL (parallel_context.cpp:47-58).
// The reference-named HOST-pointer functions a ggml build links against are in glue/parallel_context_hip.cpp.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include "../../include/ns_bestla.h"
#include "ns_common.h"

namespace {

// the slice of the RCCL ABI used here (rccl.h: ncclUniqueId 128 opaque bytes; ncclFloat32 = 7, ncclInt32 = 2, ncclSum = 0)
struct NcclUniqueId {
  char internal[128];
};
typedef void* NcclComm;
typedef int (*FnGetUniqueId)(NcclUniqueId*);
typedef int (*FnCommInitRank)(NcclComm*, int, NcclUniqueId, int);
typedef int (*FnCommDestroy)(NcclComm);
typedef const char* (*FnGetErrorString)(int);
typedef int (*FnAllReduce)(const void*, void*, size_t, int, int, NcclComm, hipStream_t);
typedef int (*FnBroadcast)(const void*, void*, size_t, int, int, NcclComm, hipStream_t);
typedef int (*FnAllToAll)(const void*, void*, size_t, int, NcclComm, hipStream_t);
constexpr int kNcclFloat32 = 7, kNcclInt32 = 2, kNcclSum = 0;

struct Rccl {
  void* so = nullptr;
  FnGetUniqueId get_unique_id = nullptr;
  FnCommInitRank comm_init_rank = nullptr;
  FnCommDestroy comm_destroy = nullptr;
  FnGetErrorString error_string = nullptr;
  FnAllReduce all_reduce = nullptr;
  FnBroadcast broadcast = nullptr;
  FnAllToAll all_to_all = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mu;

bool load_rccl() {
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (g_rccl.so) return true;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* so = nullptr;
  // NS_TP_RCCL_LIB: the collective library to bind instead (a differently named RCCL build; the shared-memory stand-in
  // of tests/tools/stub_rccl.cpp that lets several ranks share one GPU).  Bound privately: a process that already
  // holds the real RCCL (torch) keeps using it for everything else.
  const char* forced = getenv("NS_TP_RCCL_LIB");
  if (forced && *forced) {
    so = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
  } else {
    for (const char* n : names)
      if ((so = dlopen(n, RTLD_NOW | RTLD_GLOBAL)) != nullptr) break;
  }
  if (!so) {
    ns::set_error(std::string("ns_tp: cannot load RCCL: ") + (dlerror() ? dlerror() : "not found"));
    return false;
  }
  Rccl r;
  r.so = so;
  r.get_unique_id = reinterpret_cast<FnGetUniqueId>(dlsym(so, "ncclGetUniqueId"));
  r.comm_init_rank = reinterpret_cast<FnCommInitRank>(dlsym(so, "ncclCommInitRank"));
  r.comm_destroy = reinterpret_cast<FnCommDestroy>(dlsym(so, "ncclCommDestroy"));
  r.error_string = reinterpret_cast<FnGetErrorString>(dlsym(so, "ncclGetErrorString"));
  r.all_reduce = reinterpret_cast<FnAllReduce>(dlsym(so, "ncclAllReduce"));
  r.broadcast = reinterpret_cast<FnBroadcast>(dlsym(so, "ncclBroadcast"));
  r.all_to_all = reinterpret_cast<FnAllToAll>(dlsym(so, "ncclAllToAll"));
  if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.all_reduce || !r.broadcast || !r.all_to_all) {
    ns::set_error("ns_tp: RCCL lacks an expected symbol");
    dlclose(so);
    return false;
  }
  g_rccl = r;
  return true;
}

bool nccl_ok(int rc, const char* what) {
  if (rc == 0) return true;
  ns::set_error(std::string("ns_tp: ") + what + ": " + (g_rccl.error_string ? g_rccl.error_string(rc) : "RCCL error"));
  return false;
}

}  // namespace

struct ns_tp {
  int rank = 0, world = 1, device = 0;
  NcclComm comm = nullptr;
  ns_p2p* p2p = nullptr;     // optional decode-sized fast path (owned by the caller)
  size_t p2p_max_bytes = 0;
  int* d_token = nullptr;    // barrier payload
  // host-pointer entry points: grow-only staging buffers + a private stream
  float* d_stage[2] = {nullptr, nullptr};
  size_t stage_bytes[2] = {0, 0};
  hipStream_t stream = nullptr;
};

namespace {
float* tp_stage(ns_tp* t, int which, size_t bytes) {
  if (bytes <= t->stage_bytes[which]) return t->d_stage[which];
  if (t->d_stage[which]) hipFree(t->d_stage[which]);
  t->d_stage[which] = nullptr;
  t->stage_bytes[which] = 0;
  const size_t want = (bytes + (size_t(1) << 20) - 1) & ~((size_t(1) << 20) - 1);
  if (hipMalloc((void**)&t->d_stage[which], want) != hipSuccess) {
    ns::set_error("ns_tp: staging allocation failed");
    return nullptr;
  }
  t->stage_bytes[which] = want;
  return t->d_stage[which];
}
bool tp_stream(ns_tp* t) {
  if (t->stream) return true;
  if (hipSetDevice(t->device) != hipSuccess || hipStreamCreateWithFlags(&t->stream, hipStreamNonBlocking) != hipSuccess) {
    ns::set_error("ns_tp: stream creation failed");
    return false;
  }
  return true;
}
}  // namespace

extern "C" {

int ns_tp_unique_id(void* out128) {
  if (!out128 || !load_rccl()) return -1;
  NcclUniqueId id;
  if (!nccl_ok(g_rccl.get_unique_id(&id), "ncclGetUniqueId")) return -1;
  memcpy(out128, id.internal, sizeof(id.internal));
  return 0;
}

ns_tp* ns_tp_init(int rank, int world, const void* unique_id128, int device) {
  if (world < 1 || rank < 0 || rank >= world || (world > 1 && !unique_id128)) {
    ns::set_error("ns_tp_init: bad rank / world / id");
    return nullptr;
  }
  if (world == 1 && device < 0) {  // a single rank without a GPU: every collective is the identity
    ns_tp* solo = new ns_tp();
    solo->device = -1;
    return solo;
  }
  if (hipSetDevice(device) != hipSuccess) {
    ns::set_error("ns_tp_init: hipSetDevice failed (no such GPU)");
    return nullptr;
  }
  ns_tp* t = new ns_tp();
  t->rank = rank, t->world = world, t->device = device;
  if (hipMalloc((void**)&t->d_token, sizeof(int)) != hipSuccess || hipMemset(t->d_token, 0, sizeof(int)) != hipSuccess) {
    ns::set_error("ns_tp_init: device allocation failed");
    delete t;
    return nullptr;
  }
  if (!load_rccl()) {
    hipFree(t->d_token);
    delete t;
    return nullptr;
  }
  NcclUniqueId id;
  if (unique_id128)
    memcpy(id.internal, unique_id128, sizeof(id.internal));
  else if (!nccl_ok(g_rccl.get_unique_id(&id), "ncclGetUniqueId")) {  // a single rank needs no exchange
    hipFree(t->d_token);
    delete t;
    return nullptr;
  }
  if (!nccl_ok(g_rccl.comm_init_rank(&t->comm, world, id, rank), "ncclCommInitRank")) {
    hipFree(t->d_token);
    delete t;
    return nullptr;
  }
  return t;
}

void ns_tp_destroy(ns_tp* t) {
  if (!t) return;
  if (t->comm && g_rccl.comm_destroy) g_rccl.comm_destroy(t->comm);
  if (t->d_token) hipFree(t->d_token);
  for (int i = 0; i < 2; i++)
    if (t->d_stage[i]) hipFree(t->d_stage[i]);
  if (t->stream) hipStreamDestroy(t->stream);
  delete t;
}

int ns_tp_size(const ns_tp* t) { return t ? t->world : 1; }
int ns_tp_rank(const ns_tp* t) { return t ? t->rank : 0; }
int ns_tp_is_master(const ns_tp* t) { return !t || t->rank == 0; }

int ns_tp_attach_p2p(ns_tp* t, ns_p2p* p2p, size_t max_bytes) {
  if (!t) return -1;
  t->p2p = p2p;
  t->p2p_max_bytes = p2p ? max_bytes : 0;
  return 0;
}

int ns_tp_reduce_add(ns_tp* t, const float* dSend, float* dRecv, size_t count, void* stream) {
  if (!t || !dSend || !dRecv) {
    ns::set_error("ns_tp_reduce_add: null argument");
    return -1;
  }
  // NS_TP_FORCE_RCCL=1: a single rank still goes through its (one-rank) RCCL communicator — lets a one-GPU box
  // exercise library loading, communicator set-up and the collective calls themselves
  static const bool force_env = getenv("NS_TP_FORCE_RCCL") != nullptr;  // the environment only: contexts differ
  const bool force = force_env && t->comm != nullptr;
  if (t->world == 1 && !force) {
    if (dSend != dRecv &&
        hipMemcpyAsync(dRecv, dSend, count * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) {
      ns::set_error("ns_tp_reduce_add: copy failed");
      return -1;
    }
    return 0;
  }
  // decode-sized, in place, 16-byte aligned: the one-shot kernel over peer-mapped HBM (xGMI); a timed-out flag wait
  // there is sticky (ns_hip_p2p_error) — the caller detaches the fast path and falls back to RCCL for good
  if (t->p2p && dSend == dRecv && count * sizeof(float) <= t->p2p_max_bytes && (reinterpret_cast<uintptr_t>(dRecv) & 15) == 0)
    return ns_hip_p2p_all_reduce_f32(t->p2p, dRecv, count, stream);
  return nccl_ok(g_rccl.all_reduce(dSend, dRecv, count, kNcclFloat32, kNcclSum, t->comm, (hipStream_t)stream), "ncclAllReduce") ? 0 : -1;
}

int ns_tp_broadcast(ns_tp* t, float* dBuf, size_t count, void* stream) {
  if (!t || !dBuf) {
    ns::set_error("ns_tp_broadcast: null argument");
    return -1;
  }
  if (t->world == 1 && !(getenv("NS_TP_FORCE_RCCL") && t->comm)) return 0;
  return nccl_ok(g_rccl.broadcast(dBuf, dBuf, count, kNcclFloat32, 0, t->comm, (hipStream_t)stream), "ncclBroadcast") ? 0 : -1;
}

int ns_tp_alltoall(ns_tp* t, const float* dSend, float* dRecv, size_t count, void* stream) {
  if (!t || !dSend || !dRecv) {
    ns::set_error("ns_tp_alltoall: null argument");
    return -1;
  }
  if (t->world == 1)
    return hipMemcpyAsync(dRecv, dSend, count * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream) == hipSuccess ? 0 : -1;
  return nccl_ok(g_rccl.all_to_all(dSend, dRecv, count, kNcclFloat32, t->comm, (hipStream_t)stream), "ncclAllToAll") ? 0 : -1;
}

int ns_tp_barrier(ns_tp* t, void* stream) {
  if (!t) return -1;
  if (t->device < 0) return 0;  // the solo context without a GPU: nothing to order
  if ((t->world > 1 || (getenv("NS_TP_FORCE_RCCL") && t->comm)) &&
      !nccl_ok(g_rccl.all_reduce(t->d_token, t->d_token, 1, kNcclInt32, kNcclSum, t->comm, (hipStream_t)stream), "barrier"))
    return -1;
  return hipStreamSynchronize((hipStream_t)stream) == hipSuccess ? 0 : -1;
}

// ---- host-pointer forms (what ne_compute_forward_all_reduce hands over, ne_layers.c:5466-5476): blocking, staged
//      through device memory; one rank = a plain copy, no GPU touched ----
int ns_tp_reduce_add_host(ns_tp* t, const float* send, float* recv, size_t count) {
  if (!t || !send || !recv) {
    ns::set_error("ns_tp_reduce_add_host: null argument");
    return -1;
  }
  if (t->world == 1) {
    if (send != recv) memmove(recv, send, count * sizeof(float));
    return 0;
  }
  float* d = nullptr;
  if (!tp_stream(t) || !(d = tp_stage(t, 0, count * sizeof(float)))) return -1;
  if (hipMemcpyAsync(d, send, count * sizeof(float), hipMemcpyHostToDevice, t->stream) != hipSuccess ||
      ns_tp_reduce_add(t, d, d, count, t->stream) != 0 ||
      hipMemcpyAsync(recv, d, count * sizeof(float), hipMemcpyDeviceToHost, t->stream) != hipSuccess ||
      hipStreamSynchronize(t->stream) != hipSuccess) {
    if (!ns_hip_last_error()[0]) ns::set_error("ns_tp_reduce_add_host: copy failed");
    return -1;
  }
  return 0;
}

int ns_tp_broadcast_host(ns_tp* t, float* buf, size_t count) {
  if (!t || !buf) return -1;
  if (t->world == 1) return 0;
  float* d = nullptr;
  if (!tp_stream(t) || !(d = tp_stage(t, 0, count * sizeof(float)))) return -1;
  if ((t->rank == 0 && hipMemcpyAsync(d, buf, count * sizeof(float), hipMemcpyHostToDevice, t->stream) != hipSuccess) ||
      ns_tp_broadcast(t, d, count, t->stream) != 0 ||
      hipMemcpyAsync(buf, d, count * sizeof(float), hipMemcpyDeviceToHost, t->stream) != hipSuccess ||
      hipStreamSynchronize(t->stream) != hipSuccess)
    return -1;
  return 0;
}

int ns_tp_alltoall_host(ns_tp* t, const float* send, float* recv, size_t count) {
  if (!t || !send || !recv) return -1;
  if (t->world == 1) {
    memmove(recv, send, count * sizeof(float));
    return 0;
  }
  const size_t bytes = count * sizeof(float) * size_t(t->world);
  float *ds = nullptr, *dr = nullptr;
  if (!tp_stream(t) || !(ds = tp_stage(t, 0, bytes)) || !(dr = tp_stage(t, 1, bytes))) return -1;
  if (hipMemcpyAsync(ds, send, bytes, hipMemcpyHostToDevice, t->stream) != hipSuccess ||
      ns_tp_alltoall(t, ds, dr, count, t->stream) != 0 ||
      hipMemcpyAsync(recv, dr, bytes, hipMemcpyDeviceToHost, t->stream) != hipSuccess ||
      hipStreamSynchronize(t->stream) != hipSuccess)
    return -1;
  return 0;
}

int ns_tp_barrier_host(ns_tp* t) {
  if (!t) return -1;
  if (t->world == 1) return 0;
  return tp_stream(t) ? ns_tp_barrier(t, t->stream) : -1;
}

}  // extern "C"
