// glue/parallel_context_hip.cpp — neural-speed's tensor-parallel communication surface
// (/root/reference/neural_speed/core/parallel_context.h:40-47; reference implementation parallel_context.cpp: oneCCL
// over MPI, one process per CPU socket) on top of libns_hip.so's ns_tp_* layer: one process per GPU, RCCL over xGMI.
// Compiled against the REFERENCE's header; a maintainer builds it into NS_TP_MODEL targets in place of
// core/parallel_context.cpp.  ne_compute_forward_all_reduce (ne_layers.c:5466-5476), ne_split / ne_tp_concat
// (:1667-1760) and the model loaders (model_files.h:1425-1530) call it unchanged.
//
// Bootstrap without MPI (the launcher provides the usual variables, e.g. torchrun or `mpirun -x`):
//   WORLD_SIZE / RANK / LOCAL_RANK  (or NS_TP_WORLD_SIZE / NS_TP_RANK / NS_TP_LOCAL_RANK)
//   NS_TP_ID_FILE  path on a filesystem every rank sees (default /tmp/ns_tp_id.<uid>.<MASTER_PORT>): rank 0 writes the
//                  128-byte RCCL unique id there (write + rename), the others wait for it (60 s)
// With one rank nothing is initialised and every call is the identity.
//
// NOTE on `count`: parallel_context.cpp hands it to ccl::allreduce as an ELEMENT count, and so does this file; the
// reference's own caller passes ne_nbytes(dst) (ne_layers.c:5474), i.e. four times the tensor — reproduced as is when
// that call site is left unchanged; the fix on the ggml side is ne_nelements(dst).
#include <cstddef>  // the reference header uses size_t without including it

#include "parallel_context.h"

#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>

#include "ns_bestla.h"

struct parallel_context {
  ns_tp* tp;
};

namespace {
int env_int(const char* a, const char* b, int dflt) {
  const char* v = getenv(a);
  if (!v) v = getenv(b);
  return v ? atoi(v) : dflt;
}

ns_tp* make_tp() {
  const int world = env_int("NS_TP_WORLD_SIZE", "WORLD_SIZE", 1);
  const int rank = env_int("NS_TP_RANK", "RANK", 0);
  const int local = env_int("NS_TP_LOCAL_RANK", "LOCAL_RANK", rank);
  if (world <= 1) return ns_tp_init(0, 1, nullptr, -1);
  std::string path = getenv("NS_TP_ID_FILE") ? getenv("NS_TP_ID_FILE")
                                              : "/tmp/ns_tp_id." + std::to_string(getuid()) + "." +
                                                    (getenv("MASTER_PORT") ? getenv("MASTER_PORT") : "0");
  unsigned char id[NS_TP_UNIQUE_ID_BYTES];
  if (rank == 0) {
    if (ns_tp_unique_id(id) != 0) return nullptr;
    const std::string tmp = path + ".tmp";
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f || fwrite(id, 1, sizeof(id), f) != sizeof(id)) {
      fprintf(stderr, "parallel_context: cannot write %s\n", tmp.c_str());
      if (f) fclose(f);
      return nullptr;
    }
    fclose(f);
    rename(tmp.c_str(), path.c_str());
  } else {
    bool got = false;
    for (int i = 0; i < 600 && !got; i++) {
      FILE* f = fopen(path.c_str(), "rb");
      if (f) {
        got = fread(id, 1, sizeof(id), f) == sizeof(id);
        fclose(f);
      }
      if (!got) std::this_thread::sleep_for(std::chrono::milliseconds(100));
    }
    if (!got) {
      fprintf(stderr, "parallel_context: rank %d never saw %s\n", rank, path.c_str());
      return nullptr;
    }
  }
  ns_tp* tp = ns_tp_init(rank, world, id, local);
  if (tp) {
    ns_tp_barrier_host(tp);            // every rank has read the id
    if (rank == 0) unlink(path.c_str());  // a later run must not pick up a stale id
  }
  return tp;
}

ns_tp* instance() {
  static ns_tp* tp = make_tp();  // the reference keeps one parallel_class per process as well (get_instance)
  if (!tp) {
    fprintf(stderr, "parallel_context: initialisation failed: %s\n", ns_hip_last_error());
    abort();  // the reference asserts / lets MPI abort
  }
  return tp;
}
}  // namespace

extern "C" {

parallel_context* init_parallel_context() {
  parallel_context* p = new parallel_context();  // (the reference leaks one of these per call too)
  p->tp = instance();
  return p;
}
int get_tp_size(parallel_context* p) { return ns_tp_size(p->tp); }
int get_tp_rank(parallel_context* p) { return ns_tp_rank(p->tp); }
bool is_master(parallel_context* p) { return ns_tp_is_master(p->tp) != 0; }
void barrier(parallel_context* p) { ns_tp_barrier_host(p->tp); }
void broadcast(parallel_context* p, float* buffer, size_t count) { ns_tp_broadcast_host(p->tp, buffer, count); }
void alltoall(parallel_context* p, float* send_buffer, float* recv_buffer, size_t count) {
  ns_tp_alltoall_host(p->tp, send_buffer, recv_buffer, count);
}
void reduce_add(parallel_context* p, float* send_buffer, float* recv_buffer, size_t count) {
  if (ns_tp_reduce_add_host(p->tp, send_buffer, recv_buffer, count) != 0) {
    fprintf(stderr, "reduce_add failed: %s\n", ns_hip_last_error());
    abort();
  }
}

}  // extern "C"
