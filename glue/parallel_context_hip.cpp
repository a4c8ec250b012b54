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
//   NS_TP_RUN_ID / TORCHELASTIC_RUN_ID  the launcher's run id, part of the nonce; without one the nonce mixes in the parent pid (NS_TP_NONCE_NO_PPID=1 drops it for launches behind per-rank wrapper shells)
//   NS_TP_ID_MAX_AGE_S  how much older than this process an id file may be (default 30 s)
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

// Per-launch nonce shared by the ranks of ONE launch and by nobody else.  It names the id file and is written into it, so a
// file left behind by a crashed earlier launch is neither opened by name nor accepted by content (a stale
// 128-byte id made ncclCommInitRank hang with no time-out).  Ingredients:
//   * the launcher's run id when it exports one (NS_TP_RUN_ID, else a TORCHELASTIC_RUN_ID other than torchrun's default
//     "none"): unique per launch by contract, nothing else is needed beside it;
//   * otherwise (MASTER_* + WORLD_SIZE alone are shared by every launch of that shape) the PARENT pid — the ranks of a torchrun / mpirun / srun launch on one node are siblings.  Ranks behind per-rank wrapper
//     shells (`mpirun -n 2 sh -c ...`) have different parents: such launches set NS_TP_RUN_ID, or NS_TP_NONCE_NO_PPID=1 to
//     drop the parent pid (the failure message says so);
//   * MASTER_ADDR / MASTER_PORT / world size, whatever the case.
bool nonce_uses_ppid() {
  const char* run = getenv("NS_TP_RUN_ID");
  if (run && *run) return false;
  const char* te = getenv("TORCHELASTIC_RUN_ID");
  if (te && *te && strcmp(te, "none") != 0) return false;
  // an id file the launcher named itself (shared between the nodes of a multi-node launch) or more than one node: the parent pids differ from node to
  // node, the nonce would too — the named file / the launcher's rendezvous is what ties the ranks together then
  const char* idf = getenv("NS_TP_ID_FILE");
  if (idf && *idf) return false;
  const char* nn = getenv("NS_TP_NNODES") ? getenv("NS_TP_NNODES") : getenv("GROUP_WORLD_SIZE");
  if (nn && atoi(nn) > 1) return false;
  const char *lw = getenv("LOCAL_WORLD_SIZE"), *ww = getenv("NS_TP_WORLD_SIZE") ? getenv("NS_TP_WORLD_SIZE") : getenv("WORLD_SIZE");
  if (lw && ww && atoi(lw) > 0 && atoi(ww) > atoi(lw)) return false;  // (more ranks than this node holds)
  const char* off = getenv("NS_TP_NONCE_NO_PPID");
  return !(off && atoi(off) != 0);
}
uint64_t launch_nonce() {
  uint64_t h = 1469598103934665603ull;
  auto mix = [&](const char* t) {
    for (; t && *t; t++) h = (h ^ uint64_t(uint8_t(*t))) * 1099511628211ull;
    h = (h ^ 0xffu) * 1099511628211ull;  // field separator
  };
  const char* run = getenv("NS_TP_RUN_ID");
  if (!run || !*run) run = getenv("TORCHELASTIC_RUN_ID");
  mix(run);
  mix(getenv("MASTER_ADDR"));
  mix(getenv("MASTER_PORT"));
  mix(getenv("NS_TP_WORLD_SIZE") ? getenv("NS_TP_WORLD_SIZE") : getenv("WORLD_SIZE"));
  if (nonce_uses_ppid()) {
    char buf[64];
    snprintf(buf, sizeof(buf), "ppid%ld", long(getppid()));  // (not the session id: torchrun starts every worker in a session of its own)
    mix(buf);
  }
  return h ? h : 1;
}

// how old an id file may be when a rank READS it (its age against the reader's clock at that moment, not against the reader's start — a rank that
// starts a minute behind rank 0 must still be able to join): rank 0 removes the file as soon as every rank has read it, so a file that is
// still there after this long is the leftover of a launch that died (NS_TP_ID_MAX_AGE_S overrides; the ranks wait 60 s for it in any case)
time_t id_max_age_s() {
  const char* v = getenv("NS_TP_ID_MAX_AGE_S");
  const long s = v ? atol(v) : 300;
  return time_t(s > 0 ? s : 300);
}


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
    // a regular file of this user, not writable by anyone else, complete, carrying THIS launch's nonce, and not a leftover
    // of an earlier launch with the same nonce ingredients that died before its rank 0 could remove it: no older
    // than id_max_age_s() at the moment it is read
    auto read_valid = [&](IdFile* out) {
      const int fd = open(path.c_str(), O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
      if (fd < 0) return false;
      struct stat sb;
      IdFile in;
      const bool ok = fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_uid == getuid() && !(sb.st_mode & (S_IWGRP | S_IWOTH)) &&
                      sb.st_mtime + id_max_age_s() >= time(nullptr) &&
                      read(fd, &in, sizeof(in)) == ssize_t(sizeof(in)) && !memcmp(in.magic, "NSTPID1", 8) && in.nonce == nonce;
      close(fd);
      if (ok) *out = in;
      return ok;
    };
    bool got = false;
    for (int i = 0; i < 600 && !got; i++) {
      IdFile first, again;
      if (read_valid(&first)) {
        // rank 0 unlinks the name first thing: a reader that opened a leftover just before that sees, a moment later, no
        // file or different bytes under the name — accept only an id that is still there unchanged
        std::this_thread::sleep_for(std::chrono::milliseconds(150));
        got = read_valid(&again) && !memcmp(&first, &again, sizeof(first));
        if (got) rec = first;
      }
      if (!got) std::this_thread::sleep_for(std::chrono::milliseconds(100));
    }
    if (!got) {
      fprintf(stderr, "parallel_context: rank %d never saw a valid %s%s\n", rank, path.c_str(),
              nonce_uses_ppid() ? " (no launcher run id: the nonce includes the parent pid — ranks started behind per-rank wrapper "
                                  "shells must export NS_TP_RUN_ID, or NS_TP_NONCE_NO_PPID=1)" : "");
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

// test hook (tests/test_glue.py): the nonce this process would name its id file with
unsigned long long ns_pc_launch_nonce(void) { return launch_nonce(); }

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
