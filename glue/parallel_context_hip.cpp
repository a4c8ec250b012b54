// glue/parallel_context_hip.cpp — neural-speed's tensor-parallel communication surface
// (/root/reference/neural_speed/core/parallel_context.h:40-47; reference implementation parallel_context.cpp: oneCCL
// over MPI, one process per CPU socket) on top of libns_hip.so's ns_tp_* layer: one process per GPU, RCCL over xGMI.
// Compiled against the REFERENCE's header; a maintainer builds it into NS_TP_MODEL targets in place of
// core/parallel_context.cpp.  ne_compute_forward_all_reduce (ne_layers.c:5466-5476), ne_split / ne_tp_concat
// (:1667-1760) and the model loaders (model_files.h:1425-1530) call it unchanged.
//
// Bootstrap without MPI (the launcher provides the usual variables, e.g. torchrun or `mpirun -x`):
//   WORLD_SIZE / RANK / LOCAL_RANK  (or NS_TP_WORLD_SIZE / NS_TP_RANK / NS_TP_LOCAL_RANK)
//   NS_TP_ID_FILE  path on a filesystem every rank sees (default /tmp/ns_tp_id.<uid>.<launch nonce>): rank 0 removes
//                  whatever is there, writes {magic, launch nonce, 128-byte RCCL unique id} (O_EXCL, 0600, write +
//                  rename) and the others wait up to 60 s for a file carrying THEIR launch's nonce
//   NS_TP_RUN_ID / TORCHELASTIC_RUN_ID  the launcher's run id, part of the nonce (else MASTER_PORT + the parent pid)
// With one rank nothing is initialised and every call is the identity.
//
// NOTE on `count`: parallel_context.cpp hands it to ccl::allreduce as an ELEMENT count, and so does this file; the
// reference's own caller passes ne_nbytes(dst) (ne_layers.c:5474), i.e. four times the tensor — reproduced as is when
// that call site is left unchanged; the fix on the ggml side is ne_nelements(dst).
#include <cstddef>  // the reference header uses size_t without including it

#include "parallel_context.h"

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cerrno>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
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

// Per-launch nonce shared by the ranks of ONE launch and by nobody else: the launcher's run id when it exports one
// (torchrun: TORCHELASTIC_RUN_ID; anything: NS_TP_RUN_ID), mixed with the parent pid on a single node (the ranks of a
// torchrun / mpirun launch are siblings).  It names the id file and is written into it, so a file left behind by a
// crashed earlier launch is neither opened by name nor accepted by content (a stale 128-byte id made
// ncclCommInitRank hang with no time-out).
uint64_t launch_nonce() {
  uint64_t h = 1469598103934665603ull;
  auto mix = [&](const char* t) {
    for (; t && *t; t++) h = (h ^ uint64_t(uint8_t(*t))) * 1099511628211ull;
  };
  const char* run = getenv("NS_TP_RUN_ID");
  if (!run) run = getenv("TORCHELASTIC_RUN_ID");
  // only values every rank of one launch shares, whatever spawned it (torchrun, mpirun -n 2 sh -c ..., srun, several nodes
  // over one NS_TP_ID_FILE): a parent pid is not one of them — ranks behind per-rank wrapper shells would never agree
  mix(run);
  mix(getenv("MASTER_ADDR"));
  mix(getenv("MASTER_PORT"));
  mix(getenv("NS_TP_WORLD_SIZE") ? getenv("NS_TP_WORLD_SIZE") : getenv("WORLD_SIZE"));
  return h ? h : 1;
}

const time_t g_start_time = time(nullptr);

struct IdFile {  // what rank 0 publishes
  char magic[8];
  uint64_t nonce;
  unsigned char id[NS_TP_UNIQUE_ID_BYTES];
};

ns_tp* make_tp() {
  const int world = env_int("NS_TP_WORLD_SIZE", "WORLD_SIZE", 1);
  const int rank = env_int("NS_TP_RANK", "RANK", 0);
  const int local = env_int("NS_TP_LOCAL_RANK", "LOCAL_RANK", rank);
  if (world <= 1) return ns_tp_init(0, 1, nullptr, -1);
  const uint64_t nonce = launch_nonce();
  char hex[17];
  snprintf(hex, sizeof(hex), "%016llx", static_cast<unsigned long long>(nonce));
  std::string path = getenv("NS_TP_ID_FILE") ? getenv("NS_TP_ID_FILE")
                                              : "/tmp/ns_tp_id." + std::to_string(getuid()) + "." + hex;
  IdFile rec;
  memset(&rec, 0, sizeof(rec));
  if (rank == 0) {
    memcpy(rec.magic, "NSTPID1", 8);
    rec.nonce = nonce;
    const std::string tmp = path + ".tmp";
    unlink(path.c_str());  // whatever an earlier launch left under this name is gone before any reader can match it
    unlink(tmp.c_str());
    if (ns_tp_unique_id(rec.id) != 0) return nullptr;
    // O_EXCL | O_NOFOLLOW, 0600: never write through a link or into a file somebody else created first
    const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0600);
    bool ok = fd >= 0 && write(fd, &rec, sizeof(rec)) == ssize_t(sizeof(rec)) && fsync(fd) == 0;
    if (fd >= 0) ok = (close(fd) == 0) && ok;
    ok = ok && rename(tmp.c_str(), path.c_str()) == 0;
    if (!ok) {
      fprintf(stderr, "parallel_context: cannot publish %s (%s)\n", path.c_str(), strerror(errno));
      unlink(tmp.c_str());
      unlink(path.c_str());
      return nullptr;
    }
  } else {
    bool got = false;
    for (int i = 0; i < 600 && !got; i++) {
      const int fd = open(path.c_str(), O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
      if (fd >= 0) {
        struct stat sb;
        IdFile in;
        // a regular file of this user, not writable by anyone else, complete, carrying THIS launch's nonce — and not a
        // leftover of an earlier launch with the same address / port / size that died before rank 0 could remove it:
        // written no earlier than five minutes before this process started (rank 0 also unlinks the name first thing)
        got = fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_uid == getuid() && !(sb.st_mode & (S_IWGRP | S_IWOTH)) &&
              sb.st_mtime + 300 >= g_start_time &&
              read(fd, &in, sizeof(in)) == ssize_t(sizeof(in)) && !memcmp(in.magic, "NSTPID1", 8) && in.nonce == nonce;
        if (got) rec = in;
        close(fd);
      }
      if (!got) std::this_thread::sleep_for(std::chrono::milliseconds(100));
    }
    if (!got) {
      fprintf(stderr, "parallel_context: rank %d never saw a valid %s\n", rank, path.c_str());
      return nullptr;
    }
  }
  ns_tp* tp = ns_tp_init(rank, world, rec.id, local);
  if (tp) ns_tp_barrier_host(tp);            // every rank has read the id
  if (rank == 0) unlink(path.c_str());       // success or failure: a later launch must not find it
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
