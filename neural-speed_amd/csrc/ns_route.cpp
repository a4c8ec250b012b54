// ns_route.cpp — the reference's per-token device graph: deferred, fused, verified, replayed (round 5: replay; round 6: one route per device
// queue, the deferred-launch window for evaluations that are not replayed, the fp16 kv mirror inside the plan, re-evaluation of a token)
//
// What it serves: a reference tree built with its device switch (-DNS_SYCL) rebuilds its graph every token
// (/root/reference/neural_speed/models/llama/llama.cpp:148) and its executor issues the ~780 nodes one by one
// (core/ne_layers.c:11915-12028: bestla_parallel_for per node, strictly serial, :11973).  On this library's bestla_device_* set every
// node is one HIP launch, ~3.5 us of host time each: an unchanged Model.generate() was bound by the HOST (242-297 tok/s),
// not by the GPU.  The reference's executor cannot be changed; what it calls can:
//   * every launch of the route (bestla_device_f32f32_forward and the pointer-level functions behind glue/ne_bestla_hip_device.c) is
//     described by one plain RouteOp and handed to route_submit();
//   * NOTHING is launched when it is handed over.  The reference's executor only synchronises where data crosses to the host
//     (bestla_device_sync / _memcpy: the embeddings go in and the logits come out there, ne_layers.c:8345-8346), so an evaluation's launches
//     are collected until then (the WINDOW) and go out as the library's fused forms — the reference builds its graph from single operators
//     (llama.cpp:203-330), a window knows what follows each of them (optimize() below).  Round 5 had these fusions under replay only: a 1500-token
//     prompt was ~800 single launches (58 ms), the first tokens of every generation too;
//   * two consecutive tokens whose op sequences agree in everything except ONE moving value per op (RoPE's n_past, the kv-cache cell a
//     cpy writes, the attention's context length) make a PLAN: the sequence is cut into segments of a few dozen ops, each captured into
//     a HIP graph in which a moving value is base + delta * (*k) — k one device word that the first segment increments (Affine, ns_common.h);
//   * from then on a token's ops are only COMPARED with the plan (a memcmp per node: ~0.1 us instead of a launch); when the last op of a
//     segment has matched, that segment's graph is launched (the GPU starts on a token while the executor is still walking it) — nothing of a
//     plan is launched before it is verified, so a token that deviates (another prompt, a context shift, a different batch) falls back
//     without side effects: what it handed over so far goes through the window, the plan is dropped, and the next agreeing tokens make a new one
//     (after consecutive fall-backs with a growing pause: interleaved sequences must not pay a capture every other token).
// One Route per device queue (bestla_create_device): two models — or a draft and a target model — in one process keep separate traces and plans.
// NS_DEVICE_REPLAY=0 turns plans off, NS_ROUTE_WINDOW=0 the window (every op launches as it comes: the round-4 behaviour).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <unordered_set>
#include <vector>

#include "../../include/ns_bestla.h"
#include "ns_common.h"
#include "ns_route.h"

extern "C" {
int ns_hip_binary_nd_f32(int is_mul, const float* dA, const float* dB, float* dDst, const long long ne0[4], const long long nb0[4],
                         const long long ne1[4], const long long nb1[4], const long long nbd[4], void* stream);
int ns_hip_lazy_flush(void);
int ns_hip_lazy_rms_norm(int rows, int cols, float eps, const float* dIn, float* dOut, void* stream);
int ns_hip_lazy_silu(const float* dSrc, float* dDst, size_t n, void* stream);
int ns_hip_lazy_mul(const float* dA, const float* dB, float* dDst, const long long ne0[4], const long long nb0[4], const long long ne1[4],
                    const long long nb1[4], const long long nbd[4], void* stream);
int ns_hip_mha_f32_device_layout(const float* dQ, const float* dK, const float* dV, float* dO, int batch, int seq, int seq_all, int heads,
                                 int heads_kv, int head_size, int n_ctx, float scale, int masked, void* stream);
}

namespace ns {
namespace {

__global__ void route_count_kernel(int* k) { *k += 1; }

struct PlanOp {
  RouteOp op;       // the values of the token the plan was made from (k = 0)
  int moving;       // 0 nothing moves, 1 the integer field route_moving_int(kind), 2 pointer p[1]
  long long delta;  // per token
  unsigned pshift;  // bit x: pointer p[x] is an ACTIVATION of the reference's device pool — it arrives shifted by Route::act_delta per token
                    // (see below) and is replayed at the plan's address
};
struct Segment {
  int beg, end;  // the reference's launches [beg, end) this segment stands for (it is replayed when launch end - 1 has matched)
  int xbeg, xend;  // the captured launches: entries of Route::xops
  hipGraphExec_t exec;
};
// What a segment's graph actually launches.  The reference builds its decode graph from single operators (llama.cpp:203-330: rms_norm,
// mul, three mul_mat, two rope, two cpy, flash_attn, mul_mat, add, ...); a plan knows the whole token, so at capture time runs of them
// become the library's fused launches — same tensors written, same values (the reference's own fused nodes compute exactly these):
//   XK_QKV      three mul_mat of one input, outputs equally spaced   -> ns_hip_fusion_qkv_forward       (ip_fusion_qkv.cpp:84-86)
//   XK_ROPE2    rope(q) and rope(k) in place on adjacent rows         -> one rope launch over both
//   XK_DUP2     the K and V cache writes                               -> one copy launch
//   XK_ROPE_APPEND  both of the above (plain RoPE, fp32 cache cells)       -> one launch (rope_append_kernel)
//   XK_GEMM_ADD mul_mat whose only reader is the residual add           -> the GEMV's Add epilogue         (bestla_common.hpp:121-147)
//   XK_GATEUP   mul_mat(w1), silu, mul_mat(w3), mul                     -> ns_hip_fusion_ffn3_gateup      (ip_fusion_ffn.cpp:364-406)
// The intermediate tensors a fused launch does not write (the raw mul_mat results) are checked to have no other reader in the token.
//   carried RMS norms (round 5, ns_norm_link; llama.cpp:178-184, :385-391): rms_norm + mul(gamma) in front of a QKV / gate-up / mul_mat launch
//   whose input tensor came out of a XK_GEMM_ADD launch are not launched at all — that producer also writes fp16(gamma . x) and the tiles'
//   sums of squares, the consumer streams them and divides its dot products by rms(x).  The producer needs the fp16 shadow of ITS input:
//   the attention's merge kernel and the gate/up launch write one when asked (ExecOp::o16).
//   XK_QKV_ROPE (round 6)  XK_QKV with a carried norm + XK_ROPE_APPEND                                 -> ONE launch: the decode GEMV's RoPE epilogue (ns_qkv_rope, as the
//               library's own whole-token path uses it) rotates q and k from a per-token angle table (XK_ROPE_TABLE: one captured launch per token, shared by
//               every layer) and stores k / v into the fp16 mirror AND the reference's fp32 cache cells: 32 launches of a 7B token gone
// A window's PROMPT-sized evaluation (round 6, prefill_pass): the tiled GEMMs multiply fp16 activations, so every producer that can hands its consumer an fp16 copy
// beside the fp32 tensor the graph defines (ExecOp::o16p -> a16p: the attention's output rows, the gate / up product, the normed rows) instead of a conversion pass
// per GEMM, and
//   XK_NORM16      rms_norm + mul(gamma) whose plain result nobody else reads       -> ns_hip_norm_mul_h: gamma . norm(x) as fp32 and fp16, one launch
//   XK_QKV_ROPE_M  XK_QKV + rope(q) + rope(k)                                         -> the fused-QKV GEMM with the RoPE epilogue (as the library's own prefill path):
//                  q, k rotated and v where the graph has them, k / v also as fp16 straight into the kv mirror (no conversion in front of the attention)
enum ExecKind : uint32_t { XK_OP = 0, XK_QKV, XK_ROPE2, XK_DUP2, XK_GEMM_ADD, XK_GATEUP, XK_ROPE_APPEND, XK_QKV_ROPE, XK_ROPE_TABLE, XK_NORM16, XK_QKV_ROPE_M };
constexpr int kXIdx = 10;
struct ExecOp {
  uint32_t xk;
  int idx[kXIdx];    // the plan ops it stands for (-1: unused); XK_OP: idx[0]; a carried norm's rms_norm / mul ride in idx[4], idx[5] of its consumer;
                     // XK_QKV_ROPE: q, k, v mul_mat, rope(q), (norm, mul), rope(k), cpy(k), cpy(v)
  int in_link = -1;  // consumes Route::links[in_link] (its activations are that link's shadow)
  int out_link = -1; // XK_GEMM_ADD: produces Route::links[out_link]
  int a16 = -1;      // XK_GEMM_ADD: fp16 shadow of its input = Route::shadows[a16]
  int o16 = -1;      // attention / XK_GATEUP: also writes the fp16 shadow Route::shadows[o16] of its output
  int aux = -1;      // XK_ROPE_TABLE: the plan's rope op whose parameters the table is made from (the launch stands for none of the reference's)
  void* a16p = nullptr;  // window, prompt size: fp16 copy of the launch's activations (written by an earlier launch of the window) ...
  void* o16p = nullptr;  // ... and where this launch leaves the fp16 copy of its result
};
inline ExecOp xop(uint32_t xk, int a, int b = -1, int c = -1, int d = -1) { return ExecOp{xk, {a, b, c, d, -1, -1, -1, -1, -1, -1}}; }
struct NormLink {
  const float* gamma;
  float eps;
  int n, stride;           // norm size; floats per row of the sums (a multiple of 4 >= ceil(n / 16))
  size_t h_off, s_off;     // fp16(gamma . x) and the tile sums inside Route::link_mem
};
// a copy that brought an evaluation's input tensor (the embeddings): kept on the device so that the evaluation can be issued again
struct Stash {
  void* dst;
  size_t bytes, off;
};
enum { TM_PREV_END = 0, TM_FIRST_OP, TM_FIRST_LAUNCH, TM_LAST_LAUNCH, TM_SYNC_IN, TM_SYNC_OUT, TM_N };
struct Route {
  hipStream_t st = nullptr;
  bool off = false;  // this queue's graphs could not be captured (three times): plans stay off for it
  std::vector<RouteOp> cur, prev;
  size_t launched = 0;  // ops of `cur` that are on the stream already (through replayed segments or a flushed window); the rest is the window
  bool have_plan = false;
  std::vector<PlanOp> plan;
  std::vector<Segment> segs;
  std::vector<ExecOp> xops;
  int* kdev = nullptr;
  long long khost = 0;  // tokens handed to the plan (the one being verified included)
  long long kdone = 0;  // ... of which completed on it
  int pos = 0, seg = 0;
  // The reference's device pool is a bump allocator that its graph builder rewinds per layer but not per token
  // (ne_new_device_tensor_impl, ne_layers.c:904-945; llama.cpp ne_buffer_save / _load): every activation tensor of token t + 1 sits
  // a constant number of bytes above its twin of token t (4864 on the small test model).  Nothing but the token's own launches and
  // the two copies at its ends (embeddings in, logits out) ever touches those tensors, so a replayed token runs on the PLAN's
  // activations: an incoming pointer is expected at plan address + act_delta * k, the graphs use the plan addresses, and the two
  // copies are redirected there (route_translate_*).  Weights and the kv cache do not move and are compared as they are.
  long long act_delta = 0;
  std::unordered_set<const void*> act_ptrs;  // plan addresses of the shifted pointers
  bool last_replayed = false;  // the token that just ended ran from the plan: its logits are at the plan's address
  // NS_ROUTE_TIMING=1 (diagnostics): events around a replayed token's segments -> GPU span per token; host marks of a token's phases; printed when the route detaches
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool ev_pending = false;
  double gpu_ms_sum = 0.0, host_us_sum = 0.0;
  long long gpu_tokens = 0;
  long long t_first_us = 0;
  long long tm[TM_N] = {0, 0, 0, 0, 0, 0};
  double tm_sum[TM_N] = {0, 0, 0, 0, 0, 0};  // [i]: us from mark i to mark i + 1 (the last: sync out -> end of the token's last copy), replayed tokens
  long long tm_tokens = 0;
  bool tm_replayed = false;
  std::vector<NormLink> links;     // carried norms of the plan
  std::vector<size_t> shadows;     // fp16 shadows (offsets into link_mem)
  char* link_mem = nullptr;        // device memory behind both (and the RoPE angle table of XK_QKV_ROPE launches, at rope_tab_off)
  size_t rope_tab_off = 0;
  int failures = 0;  // plans that could not be captured: after a few the layer turns itself off for this queue
  // fall-backs in a row: a server alternating contexts or a beam switch makes two agreeing tokens, a plan (a capture + instantiation
  // of ~200 launches, milliseconds), and drops it on the third — from the second fall-back in a row on the next plan waits 4, 8 .. 64 agreeing tokens
  // (eight replayed tokens in a row clear the count)
  int bails_in_a_row = 0, plan_pause = 0;
  // the evaluation's inputs (see Stash): [0, stash_used) of stash_mem
  std::vector<Stash> stash;
  char* stash_mem = nullptr;
  size_t stash_cap = 0, stash_used = 0;
  bool stash_complete = true;  // false: an input of the evaluation could not be kept (it cannot be issued again from its start)
  bool stash_stale = false;    // the stash belongs to an evaluation that has ended: the next input copy starts it over
  bool tm_prev_pending = false;
  int copies_pending = 0;      // copies issued on the queue since it was last waited for (a wait with none pending can be left to the next copy: route_defer_sync)
  long long win_t0 = 0, win_ops = 0, win_launches = 0, win_issue_us = 0;  // NS_ROUTE_TIMING: the last window that went out
  uint64_t stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // tokens replayed, tokens eager, plans built, bail-outs, ops per token, captured launches per token, capture failures, -
};
std::vector<Route*> g_routes;        // one per device queue (a handful at most)
thread_local Route* t_R = nullptr;   // the queue the calling thread is serving: set by every entry that takes a stream
#define R (*t_R)
uint64_t g_stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // sums over every route this process had (ns_hip_route_stats)
std::atomic<int> g_enabled{-1};        // NS_DEVICE_REPLAY / ns_hip_route_set_enabled
std::atomic<int> g_links_enabled{-1};  // NS_ROUTE_LINKS / ns_hip_route_set_enabled(3 / 5)
thread_local bool t_in_exec = false;
inline void stat(int i, uint64_t v = 1) { R.stats[i] += v, g_stats[i] += v; }
inline void stat_set(int i, uint64_t v) { R.stats[i] = v, g_stats[i] = v; }

Route* find_route(void* stream) {
  for (Route* r : g_routes)
    if (r->st == static_cast<hipStream_t>(stream)) return r;
  return nullptr;
}
bool enabled() {
  int v = g_enabled.load();
  if (v < 0) {
    const char* e = getenv("NS_DEVICE_REPLAY");
    v = e ? atoi(e) != 0 : 1;
    g_enabled.store(v);
  }
  return v != 0 && !R.off;
}
std::atomic<int> g_window{-1};  // NS_ROUTE_WINDOW / ns_hip_route_set_enabled
// the window: an op that is not replayed waits for the evaluation's next synchronisation point and goes out fused.  NS_ROUTE_WINDOW=0 (or the whole layer
// off: NS_DEVICE_REPLAY=0, ns_hip_route_set_enabled(0)): it launches when it is handed over, one launch per operator (round 5 / round 4)
bool window_on() {
  int v = g_window.load();
  if (v < 0) {
    const char *w = getenv("NS_ROUTE_WINDOW"), *e = getenv("NS_DEVICE_REPLAY");
    v = (w && atoi(w) == 0) || (e && atoi(e) == 0) ? 0 : 1;
    g_window.store(v);
  }
  return v != 0;
}
int seg_ops() {
  // (launches per segment; measured on the 7B-shaped model, 355 launches per token: 12 -> 378 tok/s, 24 -> 392, 64 -> 403 —
  // fewer graph launches on the host; the first segment still starts after a few dozen comparisons)
  static const int v = getenv("NS_ROUTE_SEG") ? atoi(getenv("NS_ROUTE_SEG")) : 64;
  return v > 0 ? v : 64;
}
// the FIRST segment is shorter: the GPU starts on a token when the reference's executor has handed over its first 16 launches, not its
// first third (with the norms carried 64 launches are ~210 of the reference's nodes) — NS_ROUTE_SEG0, launches.  Worth little
// (profiles/r05r_route_carried_norms.txt: 64 / 16 / 6 -> 429 / 436 / 434 tok/s at n_ctx 2048, 451 / 454 / 436 at 512): what precedes a token's
// first launch is the reference building its graph, not this comparison loop.
int seg0_ops() {
  static const int v = getenv("NS_ROUTE_SEG0") ? atoi(getenv("NS_ROUTE_SEG0")) : 16;
  return v > 0 ? v : 16;
}
// the one integer field of a kind that may move from token to token (index into RouteOp::i), -1: none
int moving_int(uint32_t kind) {
  switch (kind) {
    case RK_ROPE:
    case RK_ROPE_YARN: return 4;  // n_past
    case RK_MHA: return 2;        // seq_all
    default: return -1;
  }
}
bool moving_ptr(uint32_t kind) { return kind == RK_DUP; }  // p[1]: the destination (a kv-cache cell)

// the kv mirrors a plan's cache writes kept current (ns_route.h): positions below the context length of the last COMPLETED token hold the cache
void mark_mirrors() {
  if (!R.have_plan) return;
  for (const PlanOp& po : R.plan)
    if (po.op.kind == RK_MHA) kvm_set_valid(po.op.p[1], int(po.op.i[2] + (po.moving == 1 ? po.delta : 0) * R.kdone));
}
void drop_plan() {
  mark_mirrors();
  // graphs that may still be running are not destroyed under them (only hipFree(link_mem) synchronised, and only with links)
  if (!R.segs.empty() && R.st) (void)hipStreamSynchronize(R.st);
  for (Segment& s : R.segs)
    if (s.exec) (void)hipGraphExecDestroy(s.exec);
  R.segs.clear();
  R.xops.clear();
  R.links.clear();
  R.shadows.clear();
  if (R.link_mem) (void)hipFree(R.link_mem), R.link_mem = nullptr;
  R.plan.clear();
  R.have_plan = false;
  R.pos = R.seg = 0;
  R.khost = R.kdone = 0;
  R.act_delta = 0;
  R.act_ptrs.clear();
  R.last_replayed = false;
}

int execute(const RouteOp& op, hipStream_t st) {
  struct Guard {
    Guard() { t_in_exec = true; }
    ~Guard() { t_in_exec = false; }
  } guard;
  const long long* i = op.i;
  // the glue launches a recorded norm / silu node before anything that is not its fusable consumer (ne_bestla_hip_device.c): the same here
  if (op.kind != RK_MUL && op.kind != RK_RMSNORM && op.kind != RK_SILU && ns_hip_lazy_flush() != 0) return -1;
  // an operator other than the cache-write copy that stores into a mirrored kv cache (an in-place shift): the mirror starts over
  if (op.kind != RK_DUP && op.kind != RK_MHA) kvm_note_foreign_write(op.kind == RK_GEMM || op.kind == RK_ADD || op.kind == RK_MUL ? op.p[2] : op.p[1], 1);
  switch (op.kind) {
    case RK_GEMM:
      return ns_hip_f32f32_forward(static_cast<const float*>(op.p[0]), static_cast<const ns_weight*>(op.p[1]), static_cast<float*>(const_cast<void*>(op.p[2])),
                                   int(i[0]), int(i[3]), int(i[4]), NS_EPI_NONE, nullptr, 0, st);
    case RK_ADD:
      return ns_hip_binary_nd_f32(0, static_cast<const float*>(op.p[0]), static_cast<const float*>(op.p[1]), static_cast<float*>(const_cast<void*>(op.p[2])),
                                  i, i + 4, i + 8, i + 12, i + 16, st);
    case RK_MUL:
      return ns_hip_lazy_mul(static_cast<const float*>(op.p[0]), static_cast<const float*>(op.p[1]), static_cast<float*>(const_cast<void*>(op.p[2])), i, i + 4,
                             i + 8, i + 12, i + 16, st);
    case RK_SILU:
      return ns_hip_lazy_silu(static_cast<const float*>(op.p[0]), static_cast<float*>(const_cast<void*>(op.p[1])), size_t(i[0]), st);
    case RK_RMSNORM:
      return ns_hip_lazy_rms_norm(int(i[0]), int(i[1]), op.f[0], static_cast<const float*>(op.p[0]), static_cast<float*>(const_cast<void*>(op.p[1])), st);
    case RK_ROPE:
      return ns_hip_rope_f32(static_cast<const float*>(op.p[0]), static_cast<float*>(const_cast<void*>(op.p[1])), int(i[0]), int(i[1]), int(i[2]), int(i[3]),
                             int(i[4]), int(i[5]), int(i[6]), op.f[0], op.f[1], op.f[2], op.f[3], st);
    case RK_ROPE_YARN:
      return ns_hip_rope_f32_yarn(static_cast<const float*>(op.p[0]), static_cast<float*>(const_cast<void*>(op.p[1])), int(i[0]), int(i[1]), int(i[2]),
                                  int(i[3]), int(i[4]), int(i[5]), int(i[6]), op.f[0], op.f[1], int(i[7]), op.f[2], op.f[3], op.f[4], op.f[5], st);
    case RK_DUP:
      return ns_hip_dup_f32(static_cast<const float*>(op.p[0]), const_cast<void*>(op.p[1]), i, i + 4, i + 8, i[12] != 0, st);
    case RK_MHA:
      return ns_hip_mha_f32_device_layout(static_cast<const float*>(op.p[0]), static_cast<const float*>(op.p[1]), static_cast<const float*>(op.p[2]),
                                          static_cast<float*>(const_cast<void*>(op.p[3])), int(i[0]), int(i[1]), int(i[2]), int(i[3]), int(i[4]), int(i[5]),
                                          int(i[6]), op.f[0], int(i[7]), st);
    default:
      set_error("device route: unknown operator in the plan");
      return -1;
  }
}

// plan op j as token k of the plan would hand it over
RouteOp expected(const PlanOp& po, long long k) {
  RouteOp e = po.op;
  if (po.moving == 1) e.i[moving_int(e.kind)] += po.delta * k;
  if (po.moving == 2) e.p[1] = static_cast<const char*>(e.p[1]) + po.delta * k;
  for (int x = 0; x < 4; x++)
    if (po.pshift & (1u << x)) e.p[x] = static_cast<const char*>(e.p[x]) + R.act_delta * k;
  return e;
}

// The evaluation's inputs are put back where the graph asked for them (Stash): segments of a plan run on the PLAN's activations, and a
// token's own tensors sit only act_delta * k above those — inside memory the segments have written.
int restore_inputs() {
  if (R.stash_stale) return 0;  // (this evaluation brought no input through the queue)
  for (const Stash& si : R.stash)
    if (hipMemcpyAsync(si.dst, R.stash_mem + si.off, si.bytes, hipMemcpyDeviceToDevice, R.st) != hipSuccess) {
      set_error("device route: restoring an evaluation's input failed");
      return -1;
    }
  return 0;
}
// The token deviates from the plan.  Segments launched so far ran on the PLAN's activations, the reference's remaining launches will read
// the addresses it asked for: the token is issued again from its first launch at those addresses — its verified ops are in R.cur, every
// launch of the route is a pure function of its inputs, kv-cache cells are rewritten with the same values, and the token's input is put back
// from its stash.  "Issued" = left to the window (R.launched says from where); the plan is forgotten.
int bail_out() {
  stat(3);
  int rc = 0;
  if (R.act_delta != 0) {
    if (R.seg > 0) rc = restore_inputs();
    R.launched = 0;
  } else {
    R.launched = size_t(R.seg < int(R.segs.size()) ? R.segs[R.seg].beg : R.pos);
  }
  R.bails_in_a_row = std::min(R.bails_in_a_row + 1, 6);
  R.plan_pause = R.bails_in_a_row >= 2 ? 1 << R.bails_in_a_row : 0;  // (one fall-back — a new prompt — costs nothing: the next two agreeing tokens make a plan)
  drop_plan();
  return rc;
}

bool fuse_on() {
  static const bool on = !(getenv("NS_ROUTE_FUSE") && atoi(getenv("NS_ROUTE_FUSE")) == 0);
  return on;
}
bool is_input_of(const RouteOp& o, const void* ptr) {
  switch (o.kind) {
    case RK_GEMM: return o.p[0] == ptr;
    case RK_ADD:
    case RK_MUL: return o.p[0] == ptr || o.p[1] == ptr;
    case RK_MHA: return o.p[0] == ptr || o.p[1] == ptr || o.p[2] == ptr;
    default: return o.p[0] == ptr;  // silu, norm, rope (in place: also its output), dup
  }
}
const void* output_of(const RouteOp& o) {
  switch (o.kind) {
    case RK_GEMM:
    case RK_ADD:
    case RK_MUL: return o.p[2];
    case RK_MHA: return o.p[3];
    default: return o.p[1];
  }
}
// a window flushed by a copy: the copy's source is read behind the window's last op (a fusion must not leave out a tensor it reads)
struct ExtraRead {
  const char* p = nullptr;
  size_t bytes = 0;
};
thread_local ExtraRead t_extra;
// tensor `ptr` (nbytes long) is produced by plan op `producer`: is it read by any op after `from` other than `allowed` before somebody rewrites it?
// (exact addresses: the graph's tensors are distinct allocations; a view into the middle of one would not be seen — the reference's
// llama graph has none on these tensors, and a fusion is only made from the exact node shapes that graph builds)
bool read_later(const std::vector<PlanOp>& plan, const void* ptr, size_t nbytes, size_t from, int allowed) {
  for (size_t j = from; j < plan.size(); j++) {
    if (int(j) == allowed) continue;
    if (is_input_of(plan[j].op, ptr)) return true;
    if (output_of(plan[j].op) == ptr) return false;
  }
  const char* c = static_cast<const char*>(ptr);
  return t_extra.p && c < t_extra.p + t_extra.bytes && t_extra.p < c + std::max<size_t>(nbytes, 1);
}
// a dense [m][n] fp32 tensor as ne hands it over (ne = {n, m, 1, 1})
bool packed_mat(const long long* ne, const long long* nb, long long n, long long m) {
  return ne[0] == n && ne[1] == m && ne[2] == 1 && ne[3] == 1 && nb[0] == 4 && (m == 1 || nb[1] == 4 * n);
}
bool packed_vec(const long long* ne, const long long* nb, long long n) { return packed_mat(ne, nb, n, 1); }
bool same_rope(const RouteOp& a, const RouteOp& b) {
  return a.kind == b.kind && a.i[0] == b.i[0] && a.i[1] == b.i[1] && a.i[3] == b.i[3] && a.i[4] == b.i[4] && a.i[5] == b.i[5] && a.i[6] == b.i[6] &&
         a.i[7] == b.i[7] && !memcmp(a.f, b.f, sizeof(a.f));
}
// the launches of a plan / a window, fused where the shapes allow (round 6: any row count — a prompt's window fuses like a decode step's plan,
// except the two forms that need ONE position: rope(q) + rope(k) on adjacent rows and the rope + cache-write launch)
std::vector<ExecOp> optimize(const std::vector<PlanOp>& plan) {
  const int n = int(plan.size());
  std::vector<ExecOp> x;
  std::vector<char> used(n, 0);
  auto K = [&](int j) { return j < n ? plan[j].op.kind : 0u; };
  auto O = [&](int j) -> const RouteOp& { return plan[j].op; };
  for (int j = 0; j < n; j++) {
    if (used[j]) continue;
    const RouteOp& o = O(j);
    const long long M = o.i[0];
    if (fuse_on() && o.kind == RK_GEMM && M >= 1) {
      // ---- K, V, Q: GEMM rope dup GEMM dup GEMM rope on one input (llama.cpp:232-262 as its graph expands) ----
      if (j + 6 < n && K(j + 1) == RK_ROPE && K(j + 2) == RK_DUP && K(j + 3) == RK_GEMM && K(j + 4) == RK_DUP && K(j + 5) == RK_GEMM &&
          K(j + 6) == RK_ROPE && O(j + 3).p[0] == o.p[0] && O(j + 5).p[0] == o.p[0] && O(j + 3).i[0] == M && O(j + 5).i[0] == M &&
          O(j + 1).p[0] == o.p[2] && O(j + 1).p[1] == o.p[2] && O(j + 2).p[0] == o.p[2] && O(j + 4).p[0] == O(j + 3).p[2] &&
          O(j + 6).p[0] == O(j + 5).p[2] && O(j + 6).p[1] == O(j + 5).p[2] && o.i[2] == O(j + 3).i[2] && o.i[2] == O(j + 5).i[2] &&
          o.i[3] == O(j + 3).i[3] && o.i[3] == O(j + 5).i[3]) {
        // the three outputs in address order, equally spaced: C, C + m * ldc, C + 2 m * ldc (ip_fusion_qkv.cpp:84-86), each at least its own width
        int g[3] = {j, j + 3, j + 5};
        for (int a = 0; a < 3; a++)
          for (int b = a + 1; b < 3; b++)
            if (O(g[b]).p[2] < O(g[a]).p[2]) std::swap(g[a], g[b]);
        const long long s0 = static_cast<const char*>(O(g[1]).p[2]) - static_cast<const char*>(O(g[0]).p[2]);
        const long long s1 = static_cast<const char*>(O(g[2]).p[2]) - static_cast<const char*>(O(g[1]).p[2]);
        const long long ldc = s0 / (4 * M);
        bool qkv = s0 == s1 && s0 % (4 * M) == 0;
        for (int a = 0; a < 3 && qkv; a++) qkv = ldc >= O(g[a]).i[1] && (M == 1 || O(g[a]).i[4] == ldc);
        // rope(k) + rope(q): the q rows directly in front of the k rows (or the other way round), same parameters, one position
        const RouteOp &rk = O(j + 1), &rq = O(j + 6);
        const long long qbytes = rq.i[2] * rq.i[3] * 4, kbytes = rk.i[2] * rk.i[3] * 4;
        const bool q_first = static_cast<const char*>(rq.p[0]) + qbytes == static_cast<const char*>(rk.p[0]);
        const bool k_first = static_cast<const char*>(rk.p[0]) + kbytes == static_cast<const char*>(rq.p[0]);
        const bool rope2 = same_rope(rk, rq) && rk.i[0] == 1 && rk.i[1] == 1 && (q_first || k_first);
        // ... and the two cache writes with them: plain RoPE, K handed over as packed [heads_kv][head_size] rows, fp32 cells
        const RouteOp &dk = O(j + 2), &dvv = O(j + 4);
        static const bool no_append = getenv("NS_ROUTE_ROPE_APPEND") && atoi(getenv("NS_ROUTE_ROPE_APPEND")) == 0;
        const bool append = rope2 && !no_append && rk.kind == RK_ROPE && rk.f[2] == 0.f && dk.i[0] == rk.i[3] && dk.i[1] == 1 && dk.i[2] == rk.i[2] &&
                            dk.i[3] == 1 && dk.i[4] == 4 && dk.i[6] == rk.i[3] * 4 && dk.i[12] == 0 && dvv.i[12] == 0;
        if (qkv) {
          x.push_back(xop(XK_QKV, g[0], g[1], g[2], -1));
          if (append) {
            x.push_back(xop(XK_ROPE_APPEND, q_first ? j + 6 : j + 1, q_first ? j + 1 : j + 6, j + 2, j + 4));
            for (int t = j; t <= j + 6; t++) used[t] = 1;
            continue;
          }
          if (rope2) x.push_back(xop(XK_ROPE2, q_first ? j + 6 : j + 1, q_first ? j + 1 : j + 6, -1, -1));
          else x.push_back(xop(XK_OP, j + 1, -1, -1, -1)), x.push_back(xop(XK_OP, j + 6, -1, -1, -1));
          x.push_back(xop(XK_DUP2, j + 2, j + 4, -1, -1));
          for (int t = j; t <= j + 6; t++) used[t] = 1;
          continue;
        }
      }
      // ---- gate / up: GEMM(w1) silu GEMM(w3) mul (dense rows: A [m][K], both results [m][N]) ----
      if (j + 3 < n && K(j + 1) == RK_SILU && K(j + 2) == RK_GEMM && K(j + 3) == RK_MUL && O(j + 2).p[0] == o.p[0] && O(j + 2).i[0] == M &&
          O(j + 1).p[0] == o.p[2] && O(j + 1).i[0] == M * o.i[1] && o.i[1] == O(j + 2).i[1] && o.i[2] == O(j + 2).i[2] && o.i[3] == O(j + 2).i[3] &&
          o.i[4] == o.i[1] && O(j + 2).i[4] == o.i[1] && (M == 1 || o.i[3] == o.i[2])) {
        const RouteOp& mu = O(j + 3);
        const void *s = O(j + 1).p[1], *t3 = O(j + 2).p[2];
        const size_t tb = size_t(M) * size_t(o.i[1]) * 4;
        const bool operands = (mu.p[0] == s && mu.p[1] == t3) || (mu.p[0] == t3 && mu.p[1] == s);
        if (operands && packed_mat(mu.i, mu.i + 4, o.i[1], M) && packed_mat(mu.i + 8, mu.i + 12, o.i[1], M) && mu.i[16] == 4 && (M == 1 || mu.i[17] == 4 * o.i[1]) &&
            mu.p[2] != s && mu.p[2] != t3 && !read_later(plan, o.p[2], tb, j + 2, -1) && !read_later(plan, t3, tb, j + 4, -1)) {
          x.push_back(xop(XK_GATEUP, j, j + 1, j + 2, j + 3));
          for (int t = j; t <= j + 3; t++) used[t] = 1;
          continue;
        }
      }
      // ---- GEMM whose result only feeds the residual add ----
      if (j + 1 < n && K(j + 1) == RK_ADD && o.i[4] == o.i[1]) {
        const RouteOp& ad = O(j + 1);
        const bool first = ad.p[0] == o.p[2], second = ad.p[1] == o.p[2];
        if ((first != second) && packed_mat(ad.i, ad.i + 4, o.i[1], M) && packed_mat(ad.i + 8, ad.i + 12, o.i[1], M) && ad.i[16] == 4 &&
            (M == 1 || ad.i[17] == 4 * o.i[1]) && ad.p[2] != o.p[2] && !read_later(plan, o.p[2], size_t(M) * size_t(o.i[1]) * 4, j + 2, -1)) {
          x.push_back(xop(XK_GEMM_ADD, j, j + 1, -1, -1));
          used[j] = used[j + 1] = 1;
          continue;
        }
      }
    }
    // a cache write outside those groups (fp32 cells): as a one-copy XK_DUP2, which keeps the fp16 mirror current under replay
    if (o.kind == RK_DUP && o.i[12] == 0) {
      x.push_back(xop(XK_DUP2, j, -1, -1, -1));
      used[j] = 1;
      continue;
    }
    x.push_back(xop(XK_OP, j, -1, -1, -1));
    used[j] = 1;
  }
  return x;
}

// tensor `ptr` from plan op `from` on, until somebody rewrites it: read by exactly the ops in `allowed` (all of them) and nobody else?
bool only_read_by(const std::vector<PlanOp>& plan, const void* ptr, size_t from, const int* allowed, int nallowed) {
  int seen = 0;
  for (size_t j = from; j < plan.size(); j++) {
    if (is_input_of(plan[j].op, ptr)) {
      bool ok = false;
      for (int a = 0; a < nallowed; a++) ok = ok || allowed[a] == int(j);
      if (!ok) return false;
      seen++;
    }
    if (output_of(plan[j].op) == ptr) break;
  }
  return seen == nallowed;
}
bool route_debug();
// On by default (NS_ROUTE_LINKS=0 or ns_hip_route_set_enabled(5) turn it off, (3) forces it on).  7B-shaped model, n_ctx 512, same box, alternating
// (profiles/r05r_route_carried_norms.txt): 323 -> 195 captured launches, GPU span per token 2009 / 1989 -> 1731 / 1737 us, 399 / 400 -> 451 / 452 tok/s.
// The carried form holds gamma . x un-normalised in fp16 (range note at ns_norm_link): an overflow shows as inf / nan, never as a wrong finite value.
bool links_on() {
  int v = g_links_enabled.load();
  if (v < 0) {
    v = getenv("NS_ROUTE_LINKS") && atoi(getenv("NS_ROUTE_LINKS")) == 0 ? 0 : 1;
    g_links_enabled.store(v);
  }
  return v != 0;
}
// carried RMS norms (see ExecKind): rms_norm, mul(gamma), consumer launch -> the consumer alone, if the normed tensor came out of a residual
// add that a XK_GEMM_ADD launch makes and that launch can be given the fp16 shadow of its own input.  `bytes`: device memory asked for.
void link_norms(std::vector<ExecOp>& x, const std::vector<PlanOp>& plan, std::vector<NormLink>& links, std::vector<size_t>& shadows, size_t* bytes) {
  auto O = [&](int j) -> const RouteOp& { return plan[j].op; };
  auto Wt = [](const void* p) { return static_cast<const ns_weight*>(p); };
  auto take = [&](size_t nbytes) {
    const size_t off = *bytes;
    *bytes += (nbytes + 255) / 256 * 256;
    return off;
  };
  std::vector<char> dead(x.size(), 0);
  auto why = [&](int jn, const char* reason) {
    if (route_debug()) fprintf(stderr, "route: the norm at launch %d stays a launch: %s\n", jn, reason);
  };
  for (size_t e = 0; e + 2 < x.size(); e++) {
    if (x[e].xk != XK_OP || x[e + 1].xk != XK_OP) continue;
    const int jn = x[e].idx[0], jm = x[e + 1].idx[0];
    const RouteOp &no = O(jn), &mu = O(jm);
    if (no.kind != RK_RMSNORM || mu.kind != RK_MUL || no.i[0] != 1 || jm != jn + 1) continue;
    const long long n = no.i[1];
    const void *X = no.p[0], *N = no.p[1], *N2 = mu.p[2];
    const bool first = mu.p[0] == N, second = mu.p[1] == N;
    if (first == second || N2 == X) { why(jn, "the mul does not multiply its result by a vector"); continue; }
    const float* gamma = static_cast<const float*>(first ? mu.p[1] : mu.p[0]);
    if (!packed_vec(mu.i, mu.i + 4, n) || !packed_vec(mu.i + 8, mu.i + 12, n) || mu.i[16] != 4) { why(jn, "gamma is not a dense vector of the norm's size"); continue; }
    // ---- the consumer: every mul_mat of it reads the normed tensor, one row, K = the norm's size ----
    ExecOp& c = x[e + 2];
    int gi[3] = {-1, -1, -1}, ng = 0;
    if (c.xk == XK_QKV) gi[0] = c.idx[0], gi[1] = c.idx[1], gi[2] = c.idx[2], ng = 3;
    else if (c.xk == XK_GATEUP) gi[0] = c.idx[0], gi[1] = c.idx[2], ng = 2;
    else if (c.xk == XK_GEMM_ADD) gi[0] = c.idx[0], ng = 1;
    else if (c.xk == XK_OP && O(c.idx[0]).kind == RK_GEMM) gi[0] = c.idx[0], ng = 1;
    else { why(jn, "what follows is not a mul_mat launch"); continue; }
    bool ok = c.in_link < 0;
    for (int g = 0; g < ng && ok; g++) {
      const RouteOp& go = O(gi[g]);
      ok = go.p[0] == N2 && go.i[0] == 1 && go.i[2] == n && route_link_weight_ok(Wt(go.p[1]));
    }
    if (!ok) { why(jn, "the consumer's mul_mat do not all read the normed row (one row, K = norm size, a weight the decode kernel carries norms for)"); continue; }
    if (!only_read_by(plan, N, size_t(jn) + 1, &jm, 1) || !only_read_by(plan, N2, size_t(jm) + 1, gi, ng)) { why(jn, "the normed tensors have other readers"); continue; }
    // ---- the producer: the launch whose residual add wrote the tensor that is normed ----
    int P = -1;
    for (int q = int(e) - 1; q >= 0 && P < 0; q--)
      if (x[q].xk == XK_GEMM_ADD && O(x[q].idx[1]).p[2] == X) P = q;
    if (P < 0 || x[P].out_link >= 0) { why(jn, "its input is not the result of a mul_mat + add launch"); continue; }
    const RouteOp& pg = O(x[P].idx[0]);
    if (pg.i[1] != n || pg.i[0] != 1 || !route_link_weight_ok(Wt(pg.p[1]))) { why(jn, "the producing mul_mat cannot carry a norm"); continue; }
    // ... needs its own activations as fp16: from the attention's merge or from the gate/up launch (itself on fp16 activations)
    int sh = x[P].a16;
    if (sh < 0) {
      int Q = -1;
      for (int q = P - 1; q >= 0 && Q < 0; q--) {
        if (x[q].xk == XK_OP && O(x[q].idx[0]).kind == RK_MHA && O(x[q].idx[0]).p[3] == pg.p[0]) Q = q;
        else if (x[q].xk == XK_GATEUP && O(x[q].idx[3]).p[2] == pg.p[0]) Q = q;
      }
      if (Q < 0) { why(jn, "no launch can hand the producer its activations as fp16"); continue; }
      if (x[Q].xk == XK_OP) {
        const RouteOp& mh = O(x[Q].idx[0]);
        if (mh.i[0] != 1 || mh.i[1] != 1 || mh.i[3] * mh.i[5] != pg.i[2] || !(mh.i[5] == 64 || mh.i[5] == 128 || mh.i[5] == 256)) { why(jn, "the attention in front of the producer is not a one-row decode step"); continue; }
      } else if (x[Q].in_link < 0 || O(x[Q].idx[0]).i[1] != pg.i[2]) {
        why(jn, "the gate/up launch in front of the producer is not on fp16 activations itself");
        continue;
      }
      if (x[Q].o16 < 0) {
        shadows.push_back(take(size_t(pg.i[2]) * 2));
        x[Q].o16 = int(shadows.size()) - 1;
      }
      sh = x[Q].o16;
    }
    NormLink L;
    L.gamma = gamma, L.eps = no.f[0], L.n = int(n), L.stride = (int((n + 15) / 16) + 3) & ~3;
    L.h_off = take(size_t(n) * 2), L.s_off = take(size_t(L.stride) * 4);
    links.push_back(L);
    const int li = int(links.size()) - 1;
    x[P].out_link = li, x[P].a16 = sh;
    c.in_link = li, c.idx[4] = jn, c.idx[5] = jm;
    dead[e] = dead[e + 1] = 1;
  }
  size_t o = 0;
  for (size_t e = 0; e < x.size(); e++)
    if (!dead[e]) x[o++] = x[e];
  x.resize(o);
}

// where the cpy node `d` (destination extents, source / destination byte strides) puts element (head, dim) of a packed [heads_kv][head_size] row:
// cell = p[1] + head * *sh + dim * *sd floats.  false: not that shape.
bool cell_strides(const RouteOp& d, long long hs, long long heads_kv, long long* sd, long long* sh) {
  int dim_ax = -1, head_ax = -1;
  for (int ax = 0; ax < 4; ax++) {
    if (d.i[ax] == 1) continue;
    if (d.i[ax] == hs && d.i[4 + ax] == 4 && dim_ax < 0) dim_ax = ax;
    else if (d.i[ax] == heads_kv && d.i[4 + ax] == 4 * hs && head_ax < 0) head_ax = ax;
    else return false;
  }
  if (dim_ax < 0 || (heads_kv > 1 && head_ax < 0) || d.i[12] != 0) return false;
  if (d.i[8 + dim_ax] % 4 || (head_ax >= 0 && d.i[8 + head_ax] % 4)) return false;
  *sd = d.i[8 + dim_ax] / 4, *sh = head_ax >= 0 ? d.i[8 + head_ax] / 4 : 0;
  return true;
}
// XK_QKV (with a carried norm: its activations are an fp16 shadow) + XK_ROPE_APPEND -> XK_QKV_ROPE, one angle table per token in front (see ExecKind)
void fuse_qkv_rope(std::vector<ExecOp>& x, const std::vector<PlanOp>& plan, size_t* bytes, size_t* tab_off) {
  static const bool off = getenv("NS_ROUTE_QKV_ROPE") && atoi(getenv("NS_ROUTE_QKV_ROPE")) == 0;
  if (off || !kv16_enabled()) return;
  auto O = [&](int j) -> const RouteOp& { return plan[j].op; };
  auto Wt = [](const void* p) { return static_cast<const ns_weight*>(p); };
  int first_rope = -1;
  std::vector<char> dead(x.size(), 0);
  for (size_t e = 0; e + 1 < x.size(); e++) {
    if (x[e].xk != XK_QKV || x[e].in_link < 0 || x[e + 1].xk != XK_ROPE_APPEND) continue;
    const ExecOp& ap = x[e + 1];
    const RouteOp &ra = O(ap.idx[0]), &rb = O(ap.idx[1]), &dk = O(ap.idx[2]), &dv = O(ap.idx[3]);
    const bool k_front = dk.p[0] == ra.p[0];
    const int jrk = k_front ? ap.idx[0] : ap.idx[1], jrq = k_front ? ap.idx[1] : ap.idx[0];
    const RouteOp &rk = O(jrk), &rq = O(jrq);
    // roles of the three mul_mat: k feeds the K cache write, v the V cache write, q is the third
    int jq = -1, jk = -1, jv = -1;
    for (int t = 0; t < 3; t++) {
      const int j = x[e].idx[t];
      if (O(j).p[2] == dk.p[0]) jk = j;
      else if (O(j).p[2] == dv.p[0]) jv = j;
      else jq = j;
    }
    if (jq < 0 || jk < 0 || jv < 0 || O(jq).p[2] != rq.p[0]) continue;
    const long long hs = rk.i[3], hkv = rk.i[2], hq = rq.i[2];
    // plain RoPE over the whole head in adjacent pairs, one position; the launch's weights as wide as the rows
    if (rk.kind != RK_ROPE || rk.i[6] != 0 || rk.i[5] != hs || (hs & 1) || Wt(O(jq).p[1])->n != hq * hs || Wt(O(jk).p[1])->n != hkv * hs ||
        Wt(O(jv).p[1])->n != hkv * hs || plan[jrk].delta != plan[jrq].delta)
      continue;
    if (first_rope >= 0 && !(same_rope(O(first_rope), rk) && plan[first_rope].delta == plan[jrk].delta)) continue;  // (one table serves every layer)
    KvMirrorArgs mk, mv;
    long long sd, sh;
    if (!kvm_args_for_cell(dk.p[1], &mk) || !kvm_args_for_cell(dv.p[1], &mv) || mk.transposed || !mv.transposed || mk.hs != hs || mv.hs != hs ||
        !cell_strides(dk, hs, hkv, &sd, &sh) || !cell_strides(dv, hs, hkv, &sd, &sh))
      continue;
    // the K cell's position in its cache is the RoPE position (the epilogue has one position for both)
    const long long kidx = (static_cast<const char*>(dk.p[1]) - mk.base32) / 4, vidx = (static_cast<const char*>(dv.p[1]) - mv.base32) / 4;
    if ((kidx / hs) % mk.n_ctx != rk.i[4] || vidx % mv.n_ctx != rk.i[4] || (kidx % hs) != 0 || plan[ap.idx[2]].delta != plan[jrk].delta * hs * 4 ||
        plan[ap.idx[3]].delta != plan[jrk].delta * 4)
      continue;
    if (first_rope < 0) first_rope = jrk;
    ExecOp f = x[e];
    f.xk = XK_QKV_ROPE;
    f.idx[0] = jq, f.idx[1] = jk, f.idx[2] = jv, f.idx[3] = jrq, f.idx[6] = jrk, f.idx[7] = ap.idx[2], f.idx[8] = ap.idx[3];
    x[e] = f;
    dead[e + 1] = 1;
  }
  if (first_rope < 0) return;
  *tab_off = *bytes;
  *bytes += (size_t(O(first_rope).i[3]) / 2 * 8 + 255) / 256 * 256;
  std::vector<ExecOp> y;
  ExecOp tab = xop(XK_ROPE_TABLE, -1);
  tab.aux = first_rope;
  y.push_back(tab);
  for (size_t e = 0; e < x.size(); e++)
    if (!dead[e]) y.push_back(x[e]);
  x.swap(y);
}

// per-stream scratch of a window's prompt-sized evaluation (ns_common.h stream_scratch slots)
constexpr int kSlotNorm16 = 30, kSlotAttn16 = 31, kSlotFfn16 = 32, kSlotRopeTab = 33;
// see ExecKind: fp16 hand-overs and the two fused forms of a prompt-sized window
void prefill_pass(std::vector<ExecOp>& x, const std::vector<PlanOp>& plan, hipStream_t st) {
  static const bool off = getenv("NS_ROUTE_PREFILL_FUSE") && atoi(getenv("NS_ROUTE_PREFILL_FUSE")) == 0;
  if (off || !fuse_on()) return;
  auto O = [&](int j) -> const RouteOp& { return plan[j].op; };
  auto Wt = [](const void* p) { return static_cast<const ns_weight*>(p); };
  std::vector<char> dead(x.size(), 0);
  auto gemm_of = [&](const ExecOp& c, int* gi) {  // the mul_mat ops of a launch that multiplies activations
    if (c.xk == XK_QKV || c.xk == XK_QKV_ROPE_M) return gi[0] = c.idx[0], gi[1] = c.idx[1], gi[2] = c.idx[2], 3;
    if (c.xk == XK_GATEUP) return gi[0] = c.idx[0], gi[1] = c.idx[2], 2;
    if (c.xk == XK_GEMM_ADD) return gi[0] = c.idx[0], 1;
    if (c.xk == XK_OP && O(c.idx[0]).kind == RK_GEMM) return gi[0] = c.idx[0], 1;
    return 0;
  };
  for (size_t e = 0; e < x.size(); e++) {
    // ---- QKV + the two ropes -> the GEMM's RoPE epilogue (k / v into the kv mirror) ----
    if (x[e].xk == XK_QKV && e + 3 < x.size() && x[e + 1].xk == XK_OP && x[e + 2].xk == XK_OP && x[e + 3].xk == XK_DUP2 && x[e + 3].idx[1] >= 0 && kv16_enabled()) {
      const RouteOp &r1 = O(x[e + 1].idx[0]), &r2 = O(x[e + 2].idx[0]), &dk = O(x[e + 3].idx[0]), &dv = O(x[e + 3].idx[1]);
      const long long M = O(x[e].idx[0]).i[0];
      int jq = -1, jk = -1, jv = -1;
      for (int t = 0; t < 3; t++) {
        const int j = x[e].idx[t];
        if (O(j).p[2] == dk.p[0]) jk = j;
        else if (O(j).p[2] == dv.p[0]) jv = j;
        else jq = j;
      }
      const bool ropes = r1.kind == RK_ROPE && r2.kind == RK_ROPE && same_rope(r1, r2) && r1.p[0] == r1.p[1] && r2.p[0] == r2.p[1];
      if (M > 16 && ropes && jq >= 0 && jk >= 0 && jv >= 0) {
        const RouteOp& rk = r1.p[0] == O(jk).p[2] ? r1 : r2;
        const RouteOp& rq = &rk == &r1 ? r2 : r1;
        const long long hs = rk.i[3], hkv = rk.i[2], hq = rq.i[2], np = rk.i[4];
        // plain whole-head RoPE in adjacent pairs over M consecutive positions of one sequence; K cells [head][n_ctx][hs], V cells [head][hs][n_ctx] from np on
        bool ok = rk.p[0] == O(jk).p[2] && rq.p[0] == O(jq).p[2] && rk.i[0] == 1 && rk.i[1] == M && rk.i[6] == 0 && rk.i[5] == hs && rk.f[2] == 0.f && (hs & 3) == 0 &&
                  Wt(O(jq).p[1])->n == hq * hs && Wt(O(jk).p[1])->n == hkv * hs && Wt(O(jv).p[1])->n == hkv * hs && dk.i[12] == 0 && dv.i[12] == 0 &&
                  dk.i[0] == hs && dk.i[1] == M && dk.i[2] == hkv && dk.i[3] == 1 && dk.i[8] == 4 && dk.i[9] == 4 * hs && dk.i[10] % (4 * hs) == 0 &&
                  dv.i[0] == M && dv.i[1] == hs && dv.i[2] == hkv && dv.i[3] == 1 && dv.i[8] == 4 && dv.i[9] % 4 == 0 && dv.i[10] == dv.i[9] * hs;
        const long long n_ctx = ok ? dk.i[10] / (4 * hs) : 0;
        ok = ok && n_ctx >= np + M && dv.i[9] == 4 * n_ctx;
        _Float16 *k16 = nullptr, *v16 = nullptr;
        if (ok) {
          const char *kbase = static_cast<const char*>(dk.p[1]) - np * hs * 4, *vbase = static_cast<const char*>(dv.p[1]) - np * 4;
          ok = kvm_for_producer(kbase, vbase, int(hkv), int(hs), int(n_ctx), int(np), int(M), &k16, &v16) &&
               stream_scratch(st, size_t(M) * size_t(hs / 2) * 8, kSlotRopeTab) != nullptr;
        }
        if (ok) {
          ExecOp f = x[e];
          f.xk = XK_QKV_ROPE_M;
          f.idx[0] = jq, f.idx[1] = jk, f.idx[2] = jv;
          f.idx[3] = &rq == &r1 ? x[e + 1].idx[0] : x[e + 2].idx[0], f.idx[6] = &rk == &r1 ? x[e + 1].idx[0] : x[e + 2].idx[0];
          f.idx[7] = x[e + 3].idx[0], f.idx[8] = x[e + 3].idx[1];  // (the cache writes stay a launch of their own: x[e + 3]; here for the mirror's geometry)
          x[e] = f;
          dead[e + 1] = dead[e + 2] = 1;
        }
      }
    }
  }
  {
    size_t o = 0;
    for (size_t e = 0; e < x.size(); e++)
      if (!dead[e]) x[o++] = x[e];
    x.resize(o);
    dead.assign(x.size(), 0);
  }
  for (size_t e = 0; e < x.size(); e++) {
    int gi[3];
    // ---- rms_norm + mul(gamma) in front of a prompt-sized mul_mat launch -> one launch, fp32 + fp16 ----
    if (e + 2 < x.size() && x[e].xk == XK_OP && x[e + 1].xk == XK_OP && O(x[e].idx[0]).kind == RK_RMSNORM && O(x[e + 1].idx[0]).kind == RK_MUL) {
      const int jn = x[e].idx[0], jm = x[e + 1].idx[0];
      const RouteOp &no = O(jn), &mu = O(jm);
      const long long rows = no.i[0], n = no.i[1];
      const void *N = no.p[1], *N2 = mu.p[2];
      const bool first = mu.p[0] == N, second = mu.p[1] == N;
      const int ng = gemm_of(x[e + 2], gi);
      // (ne_mul(normed rows, gamma): the rows are the first operand, gamma the row vector — llama.cpp:178-184)
      bool ok = rows > 16 && jm == jn + 1 && first && !second && N2 != no.p[0] && N2 != N && ng > 0 && !x[e + 2].a16p && packed_mat(mu.i, mu.i + 4, n, rows) &&
                packed_vec(mu.i + 8, mu.i + 12, n) && mu.i[16] == 4 && mu.i[17] == 4 * n && !read_later(plan, N, size_t(rows) * n * 4, size_t(jm) + 1, -1);
      for (int g = 0; g < ng && ok; g++) ok = O(gi[g]).p[0] == N2 && O(gi[g]).i[0] == rows && O(gi[g]).i[2] == n && O(gi[g]).i[3] == n;
      void* sh = ok ? stream_scratch(st, size_t(rows) * n * 2, kSlotNorm16) : nullptr;
      if (sh) {
        ExecOp f = xop(XK_NORM16, jn, jm);
        f.o16p = sh;
        x[e] = f;
        dead[e + 1] = 1;
        x[e + 2].a16p = sh;
        continue;
      }
    }
    // ---- the attention's rows / the gate-up product as fp16 for the projection that multiplies them ----
    const bool is_mha = x[e].xk == XK_OP && O(x[e].idx[0]).kind == RK_MHA && O(x[e].idx[0]).i[1] > 16 && O(x[e].idx[0]).i[0] == 1 && kv16_enabled();
    const bool is_gu = x[e].xk == XK_GATEUP && O(x[e].idx[0]).i[0] > 16;
    if (is_mha || is_gu) {
      const void* out = is_mha ? O(x[e].idx[0]).p[3] : O(x[e].idx[3]).p[2];
      const long long rows = is_mha ? O(x[e].idx[0]).i[1] : O(x[e].idx[0]).i[0];
      const long long cols = is_mha ? O(x[e].idx[0]).i[3] * O(x[e].idx[0]).i[5] : O(x[e].idx[0]).i[1];
      for (size_t c = e + 1; c < x.size() && c < e + 4; c++) {
        const int ng = gemm_of(x[c], gi);
        if (ng == 1 && O(gi[0]).p[0] == out && O(gi[0]).i[0] == rows && O(gi[0]).i[2] == cols && O(gi[0]).i[3] == cols && !x[c].a16p) {
          void* sh = stream_scratch(st, size_t(rows) * cols * 2, is_mha ? kSlotAttn16 : kSlotFfn16);
          if (sh) x[e].o16p = sh, x[c].a16p = sh;
          break;
        }
      }
    }
  }
  size_t o = 0;
  for (size_t e = 0; e < x.size(); e++)
    if (!dead[e]) x[o++] = x[e];
  x.resize(o);
}

// one launch of a plan (inside a stream capture; the device counter moves what moves) or of a window (kdev == nullptr: plain values, launched now)
int capture_xop(const ExecOp& xo, const std::vector<PlanOp>& plan, const int* kdev, hipStream_t st) {
  struct Guard {
    Guard() { t_in_exec = true; }
    ~Guard() {
      t_in_exec = false;
      g_affine = Affine{};
      g_kvm = KvMirrorPair{};
    }
  } guard;
  // a captured cache write also stores into the fp16 mirror of the cache it writes (ns_route.h); a window's attention converts what it needs itself
  auto mirror_of = [&](const void* cell, KvMirrorArgs* out) {
    if (kdev && kv16_enabled()) (void)kvm_args_for_cell(cell, out);
  };
  auto P = [&](int k) -> const PlanOp& { return plan[xo.idx[k]]; };
  auto F = [](const void* p) { return static_cast<const float*>(p); };
  auto M = [](const void* p) { return static_cast<float*>(const_cast<void*>(p)); };
  auto W = [](const void* p) { return static_cast<const ns_weight*>(p); };
  auto in_link = [&](int li) {  // the consumer side of a carried norm
    const NormLink& L = R.links[li];
    ns_norm_link lk{};
    lk.in_ssq = reinterpret_cast<const float*>(R.link_mem + L.s_off), lk.in_parts = (L.n + 15) / 16, lk.in_stride = L.stride, lk.eps = L.eps, lk.norm_size = L.n;
    return lk;
  };
  if (xo.xk != XK_OP && ns_hip_lazy_flush() != 0) return -1;
  switch (xo.xk) {
    case XK_NORM16: {
      const RouteOp &no = P(0).op, &mu = P(1).op;
      return ns_hip_norm_mul_h(int(no.i[0]), int(no.i[1]), true, no.f[0], F(no.p[0]), F(mu.p[1]), M(mu.p[2]), xo.o16p, st);
    }
    case XK_QKV_ROPE_M: {
      const RouteOp &gq = P(0).op, &gk = P(1).op, &gv = P(2).op, &rk = P(6).op, &dk = P(7).op, &dv = P(8).op;
      const long long hs = rk.i[3], hkv = rk.i[2], np = rk.i[4], m = gq.i[0], n_ctx = dk.i[10] / (4 * hs);
      _Float16 *k16 = nullptr, *v16 = nullptr;
      float* tab = static_cast<float*>(stream_scratch(st, size_t(m) * size_t(hs / 2) * 8, kSlotRopeTab));
      int rc = -2;
      if (tab && kvm_for_producer(static_cast<const char*>(dk.p[1]) - np * hs * 4, static_cast<const char*>(dv.p[1]) - np * 4, int(hkv), int(hs), int(n_ctx), int(np), int(m),
                                  &k16, &v16) &&
          launch_rope_cos_sin(int(m), int(np), int(rk.i[5]), rk.f[0], rk.f[1], rk.f[3], tab, st) == hipSuccess) {
        ns_qkv_rope r{};
        r.kcache16 = k16, r.vcache16 = v16, r.cos_sin = tab;
        r.heads = int(P(3).op.i[2]), r.heads_kv = int(hkv), r.head_size = int(hs), r.n_past = int(np), r.n_dims = int(rk.i[5]), r.mode = 0;
        r.cache_step_sl = hs, r.cache_step_head = n_ctx * hs, r.flags = 0;
        rc = qkv_rope_route_forward_m(F(gq.p[0]), xo.a16p, W(gq.p[1]), W(gk.p[1]), W(gv.p[1]), M(gq.p[2]), M(gk.p[2]), M(gv.p[2]), int(m), int(gq.i[3]), int(gq.i[4]), &r, st);
      }
      if (rc != -2) return rc;
      kvm_note_foreign_write(dk.p[1], 1);  // (nothing was stored into the mirror after all: it starts over)
      // the GEMM does not take this shape: the three mul_mat in one launch and the two ropes, as the window had them (the mirror's rows are converted by the attention)
      ns_hip_reset_error();
      {
        const long long ldc = gq.i[4];
        const RouteOp* g3[3] = {&gq, &gk, &gv};
        for (const RouteOp* g : g3)
          if (ns_hip_f32f32_forward_h(F(g->p[0]), xo.a16p, W(g->p[1]), M(g->p[2]), nullptr, int(m), int(g->i[3]), int(ldc), NS_EPI_NONE, nullptr, 0, st) != 0) return -1;
      }
      if (execute(P(6).op, st) != 0) return -1;
      t_in_exec = true;  // (execute()'s guard cleared it)
      const int rq = execute(P(3).op, st);
      t_in_exec = true;
      return rq;
    }
    case XK_ROPE_TABLE: {
      const PlanOp& po = plan[xo.aux];
      const RouteOp& r = po.op;
      g_affine = Affine{kdev, po.delta, 0};
      if (launch_rope_cos_sin(1, int(r.i[4]), int(r.i[5]), r.f[0], r.f[1], r.f[3], reinterpret_cast<float*>(R.link_mem + R.rope_tab_off), st) != hipSuccess) {
        set_error("device route: rope table launch failed");
        return -1;
      }
      return 0;
    }
    case XK_QKV_ROPE: {
      const RouteOp &gq = P(0).op, &gk = P(1).op, &gv = P(2).op, &rk = P(6).op, &dk = P(7).op, &dv = P(8).op;
      KvMirrorArgs mk, mv;
      if (!kvm_args_for_cell(dk.p[1], &mk) || !kvm_args_for_cell(dv.p[1], &mv)) {
        set_error("device route: the kv mirror of a fused QKV launch is gone");
        return -1;
      }
      const long long hs = rk.i[3], hkv = rk.i[2];
      const long long kidx = (static_cast<const char*>(dk.p[1]) - mk.base32) / 4, per_slot = hkv * mk.n_ctx * hs;
      const long long slot = kidx / per_slot;
      ns_qkv_rope r{};
      r.kcache16 = mk.m16 + slot * per_slot, r.vcache16 = mv.m16 + slot * per_slot;
      r.cos_sin = reinterpret_cast<const float*>(R.link_mem + R.rope_tab_off);
      r.heads = int(P(3).op.i[2]), r.heads_kv = int(hkv), r.head_size = int(hs), r.n_past = int(rk.i[4]), r.n_dims = int(rk.i[5]), r.mode = 0;
      r.cache_step_sl = hs, r.cache_step_head = mk.n_ctx * hs, r.flags = 0;
      QkvRopeRoute rr{};
      rr.k32 = static_cast<float*>(const_cast<void*>(dk.p[1])), rr.v32 = static_cast<float*>(const_cast<void*>(dv.p[1]));
      if (!cell_strides(dk, hs, hkv, &rr.k32_dim, &rr.k32_head) || !cell_strides(dv, hs, hkv, &rr.v32_dim, &rr.v32_head)) return -1;
      rr.k32_tok = P(7).delta / 4, rr.v32_tok = P(8).delta / 4;
      rr.kmove = kdev, rr.kd_pos = int(P(6).delta);
      rr.overflow = kvm_overflow_word();
      const ns_norm_link lk = in_link(xo.in_link);
      return qkv_rope_route_forward(F(gq.p[0]), R.link_mem + R.links[xo.in_link].h_off, W(gq.p[1]), W(gk.p[1]), W(gv.p[1]), M(gq.p[2]), M(gk.p[2]), M(gv.p[2]),
                                    int(gq.i[3]), &lk, &r, &rr, st);
    }
    case XK_QKV: {
      const RouteOp &a = P(0).op, &b = P(1).op, &c = P(2).op;
      const int m = int(a.i[0]);
      const long long ldc = (static_cast<const char*>(b.p[2]) - static_cast<const char*>(a.p[2])) / (4 * m);
      if (xo.in_link >= 0) {
        const ns_norm_link lk = in_link(xo.in_link);
        return ns_hip_fusion_qkv_forward_x(F(a.p[0]), R.link_mem + R.links[xo.in_link].h_off, W(a.p[1]), W(b.p[1]), W(c.p[1]), M(a.p[2]), nullptr, 1, int(a.i[3]),
                                           int(ldc), &lk, st);
      }
      if (xo.a16p) return ns_hip_fusion_qkv_forward_h(F(a.p[0]), xo.a16p, W(a.p[1]), W(b.p[1]), W(c.p[1]), M(a.p[2]), nullptr, m, int(a.i[3]), int(ldc), st);
      return ns_hip_fusion_qkv_forward(F(a.p[0]), W(a.p[1]), W(b.p[1]), W(c.p[1]), M(a.p[2]), m, int(a.i[3]), int(ldc), st);
    }
    case XK_ROPE2: {
      const RouteOp& a = P(0).op;  // the rows in front; the other tensor's rows follow directly
      const RouteOp& b = P(1).op;
      g_affine = Affine{kdev, P(0).delta, 0};
      const int heads = int(a.i[2] + b.i[2]);
      if (a.kind == RK_ROPE_YARN)
        return ns_hip_rope_f32_yarn(F(a.p[0]), M(a.p[1]), 1, 1, heads, int(a.i[3]), int(a.i[4]), int(a.i[5]), int(a.i[6]), a.f[0], a.f[1], int(a.i[7]), a.f[2],
                                    a.f[3], a.f[4], a.f[5], st);
      return ns_hip_rope_f32(F(a.p[0]), M(a.p[1]), 1, 1, heads, int(a.i[3]), int(a.i[4]), int(a.i[5]), int(a.i[6]), a.f[0], a.f[1], a.f[2], a.f[3], st);
    }
    case XK_ROPE_APPEND: {
      const RouteOp &a = P(0).op, &b = P(1).op, &dk = P(2).op, &dv = P(3).op;  // a: the rope whose rows come first; dk / dv: the cache writes
      const bool k_is_front = dk.p[0] == a.p[0];
      const RouteOp& rk = k_is_front ? a : b;
      g_affine = Affine{kdev, P(0).delta, 0};
      mirror_of(dk.p[1], &g_kvm.k), mirror_of(dv.p[1], &g_kvm.v);
      if (launch_rope_append(M(a.p[0]), int(a.i[2] + b.i[2]), k_is_front ? 0 : int(a.i[2]), int(rk.i[2]), int(a.i[3]), int(a.i[4]), int(a.i[5]), int(a.i[6]),
                             a.f[0], a.f[1], a.f[3], 0.f, 0.f, 0.f, dk.p[0], const_cast<void*>(dk.p[1]), dk.i, dk.i + 4, dk.i + 8, dv.p[0],
                             const_cast<void*>(dv.p[1]), dv.i, dv.i + 4, dv.i + 8, P(2).delta, P(3).delta, st) != hipSuccess) {
        set_error("device route: rope + kv-cache write launch failed");
        return -1;
      }
      return 0;
    }
    case XK_DUP2: {
      const RouteOp& a = P(0).op;
      static const long long none[4] = {0, 0, 0, 0};
      if (xo.idx[1] < 0) {  // one cache write on its own
        g_affine = Affine{kdev, P(0).delta, 0};
        mirror_of(a.p[1], &g_kvm.k);
        if (launch_dup2(a.p[0], const_cast<void*>(a.p[1]), a.i, a.i + 4, a.i + 8, a.i[12] != 0, a.p[0], const_cast<void*>(a.p[1]), none, none, none, false, st) != hipSuccess) {
          set_error("device route: kv-cache write launch failed");
          return -1;
        }
        return 0;
      }
      const RouteOp& b = P(1).op;
      g_affine = Affine{kdev, P(0).delta, P(1).delta};
      if (a.i[12] == 0) mirror_of(a.p[1], &g_kvm.k);
      if (b.i[12] == 0) mirror_of(b.p[1], &g_kvm.v);
      if (launch_dup2(a.p[0], const_cast<void*>(a.p[1]), a.i, a.i + 4, a.i + 8, a.i[12] != 0, b.p[0], const_cast<void*>(b.p[1]), b.i, b.i + 4, b.i + 8,
                      b.i[12] != 0, st) != hipSuccess) {
        set_error("device route: kv-cache write launch failed");
        return -1;
      }
      return 0;
    }
    case XK_GEMM_ADD: {
      const RouteOp &g = P(0).op, &ad = P(1).op;
      const void* other = ad.p[0] == g.p[2] ? ad.p[1] : ad.p[0];
      if (xo.in_link >= 0 || xo.out_link >= 0) {
        ns_norm_link lk = xo.in_link >= 0 ? in_link(xo.in_link) : ns_norm_link{};
        void* c16 = nullptr;
        if (xo.out_link >= 0) {
          const NormLink& L = R.links[xo.out_link];
          lk.out_gamma = L.gamma, lk.out_ssq = reinterpret_cast<float*>(R.link_mem + L.s_off), lk.out_stride = L.stride;
          c16 = R.link_mem + L.h_off;
        }
        const void* a16 = xo.in_link >= 0 ? R.link_mem + R.links[xo.in_link].h_off : (xo.a16 >= 0 ? R.link_mem + R.shadows[xo.a16] : nullptr);
        return ns_hip_f32f32_forward_x(F(g.p[0]), a16, W(g.p[1]), M(ad.p[2]), c16, 1, int(g.i[3]), int(g.i[1]), NS_EPI_ADD, F(other), int(g.i[1]), &lk, st);
      }
      if (xo.a16p)
        return ns_hip_f32f32_forward_h(F(g.p[0]), xo.a16p, W(g.p[1]), M(ad.p[2]), nullptr, int(g.i[0]), int(g.i[3]), int(g.i[1]), NS_EPI_ADD, F(other), int(g.i[1]), st);
      return ns_hip_f32f32_forward(F(g.p[0]), W(g.p[1]), M(ad.p[2]), int(g.i[0]), int(g.i[3]), int(g.i[1]), NS_EPI_ADD, F(other), int(g.i[1]), st);
    }
    case XK_GATEUP: {
      const RouteOp &g1 = P(0).op, &si = P(1).op, &g3 = P(2).op, &mu = P(3).op;
      if (xo.in_link >= 0) {
        const ns_norm_link lk = in_link(xo.in_link);
        return ns_hip_fusion_ffn3_gateup_x(F(g1.p[0]), R.link_mem + R.links[xo.in_link].h_off, W(g1.p[1]), W(g3.p[1]), M(si.p[1]), M(mu.p[2]),
                                           xo.o16 >= 0 ? R.link_mem + R.shadows[xo.o16] : nullptr, 1, NS_EPI_SILU, &lk, st);
      }
      if (xo.a16p || xo.o16p)
        return ns_hip_fusion_ffn3_gateup_x(F(g1.p[0]), xo.a16p, W(g1.p[1]), W(g3.p[1]), M(si.p[1]), M(mu.p[2]), xo.o16p, int(g1.i[0]), NS_EPI_SILU, nullptr, st);
      return ns_hip_fusion_ffn3_gateup(F(g1.p[0]), W(g1.p[1]), W(g3.p[1]), M(si.p[1]), M(mu.p[2]), int(g1.i[0]), NS_EPI_SILU, st);
    }
    default: {
      const PlanOp& po = P(0);
      if (xo.in_link >= 0 && po.op.kind == RK_GEMM) {  // (the model's last norm in front of the output projection)
        const ns_norm_link lk = in_link(xo.in_link);
        return ns_hip_f32f32_forward_x(F(po.op.p[0]), R.link_mem + R.links[xo.in_link].h_off, W(po.op.p[1]), M(po.op.p[2]), nullptr, int(po.op.i[0]), int(po.op.i[3]),
                                       int(po.op.i[4]), NS_EPI_NONE, nullptr, 0, &lk, st);
      }
      if (xo.a16p && po.op.kind == RK_GEMM)  // (a window's prompt-sized projection on the fp16 rows its producer left)
        return ns_hip_f32f32_forward_h(F(po.op.p[0]), xo.a16p, W(po.op.p[1]), M(po.op.p[2]), nullptr, int(po.op.i[0]), int(po.op.i[3]), int(po.op.i[4]), NS_EPI_NONE, nullptr, 0, st);
      t_in_exec = false;  // (execute() guards itself)
      if (po.moving) g_affine = Affine{kdev, po.delta, 0};
      if (xo.o16 >= 0 && po.op.kind == RK_MHA) g_mha_out16 = R.link_mem + R.shadows[xo.o16];
      if (xo.o16p && po.op.kind == RK_MHA) g_mha_out16 = xo.o16p;
      const int rc = execute(po.op, st);
      g_mha_out16 = nullptr;
      return rc;
    }
  }
}

bool route_timing() {
  static const bool on = getenv("NS_ROUTE_TIMING") != nullptr;
  return on;
}
long long now_us() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (long long)ts.tv_sec * 1000000ll + ts.tv_nsec / 1000;
}
bool route_debug() {
  static const bool on = getenv("NS_ROUTE_DEBUG") != nullptr;
  return on;
}
void explain_mismatch(size_t j, const RouteOp& a, const RouteOp& b) {
  fprintf(stderr, "route: op %zu (kind %u vs %u) differs between consecutive tokens in more than its moving value:", j, a.kind, b.kind);
  for (int x = 0; x < 4; x++)
    if (a.p[x] != b.p[x]) fprintf(stderr, " p[%d] %p -> %p (%+lld)", x, a.p[x], b.p[x], (long long)(static_cast<const char*>(b.p[x]) - static_cast<const char*>(a.p[x])));
  for (int x = 0; x < 24; x++)
    if (a.i[x] != b.i[x]) fprintf(stderr, " i[%d] %lld -> %lld", x, a.i[x], b.i[x]);
  for (int x = 0; x < 8; x++)
    if (memcmp(&a.f[x], &b.f[x], 4)) fprintf(stderr, " f[%d] %g -> %g", x, a.f[x], b.f[x]);
  fprintf(stderr, "\n");
}
bool make_plan() {
  const size_t n = R.cur.size();
  if (route_debug()) fprintf(stderr, "route: token ended with %zu ops (previous token: %zu)\n", n, R.prev.size());
  if (n == 0 || n != R.prev.size()) return false;
  std::vector<PlanOp> plan(n);
  long long act_delta = 0;
  std::unordered_set<const void*> act_ptrs;
  for (size_t j = 0; j < n; j++) {
    RouteOp a = R.prev[j];
    const RouteOp& b = R.cur[j];
    if (a.kind != b.kind) return false;
    PlanOp po{b, 0, 0, 0u};
    // activation pointers: all shifted by ONE constant per token (the reference's device pool, see Route::act_delta)
    for (int x = 0; x < 4; x++) {
      if (x == 1 && moving_ptr(b.kind)) continue;
      const long long d = static_cast<const char*>(b.p[x]) - static_cast<const char*>(a.p[x]);
      if (d == 0) continue;
      if (act_delta == 0) act_delta = d;
      if (d != act_delta) {
        if (route_debug()) explain_mismatch(j, R.prev[j], b);
        return false;
      }
      po.pshift |= 1u << x;
      act_ptrs.insert(b.p[x]);
      a.p[x] = b.p[x];
    }
    const int mi = moving_int(b.kind);
    if (mi >= 0 && a.i[mi] != b.i[mi]) {
      po.moving = 1, po.delta = b.i[mi] - a.i[mi];
      a.i[mi] = b.i[mi];
    } else if (mi >= 0) {
      po.moving = 1;  // (stands still between these two tokens: delta 0, still verified every token)
    }
    if (moving_ptr(b.kind)) {
      po.moving = 2, po.delta = static_cast<const char*>(b.p[1]) - static_cast<const char*>(a.p[1]);
      a.p[1] = b.p[1];
    }
    if (memcmp(&a, &b, sizeof(RouteOp)) != 0) {  // something else moves: not a token loop this layer can replay
      if (route_debug()) explain_mismatch(j, R.prev[j], b);
      return false;
    }
    // a moving context length is only served by the context-split attention at a decode step
    if (b.kind == RK_MHA && !(b.i[1] == 1 && b.i[0] == 1)) return false;
    // a plan keeps the kv mirrors current through its cache-write copies only
    KvMirrorArgs into;
    if (b.kind != RK_DUP && b.kind != RK_MHA && kvm_args_for_cell(b.kind == RK_GEMM || b.kind == RK_ADD || b.kind == RK_MUL ? b.p[2] : b.p[1], &into)) return false;
    plan[j] = po;
  }
  if (!R.kdev && hipMalloc(reinterpret_cast<void**>(&R.kdev), 64) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  if (hipMemsetAsync(R.kdev, 0, 64, R.st) != hipSuccess || hipStreamSynchronize(R.st) != hipSuccess) return false;
  // scratch the moving-length attention needs (partials laid out for the longest context): sized BEFORE any capture
  for (const PlanOp& po : plan)
    if (po.op.kind == RK_MHA) {
      const long long* i = po.op.i;
      const size_t nsplit_max = size_t((i[6] + 127) / 128);
      if (!stream_scratch(R.st, size_t(i[0]) * i[1] * i[3] * nsplit_max * (2 + i[5]) * sizeof(float), 24)) return false;
      if (!stream_scratch_zeroed(R.st, 65536 * 4, 25)) return false;  // the tickets of the merge inside the launch (ns_device.hip)
      // ... and what the fp16-mirror form needs (ns_attn.hip, moving context length)
      if (kv16_enabled() && !attn_prepare_moving(R.st, int(i[0]), int(i[3]), int(i[4]), int(i[5]), int(i[6]))) return false;
    }
  // the launches (fused where possible), then segments of about seg_ops() of them.  A segment may end only where the launches so far
  // stand for a PREFIX of the reference's launches (fusion reorders inside a layer), and never on a node the lazy peephole only records
  std::vector<ExecOp> xops;
  std::vector<Segment> segs;
  // one attempt: the launches, their segments, the captures.  With carried norms first; should their captures be refused (a weight or shape the
  // decode kernel does not carry a norm for), once more without them
  auto attempt = [&](bool with_links) {
    xops = optimize(plan);
    segs.clear();
    R.links.clear(), R.shadows.clear();
    if (R.link_mem) (void)hipFree(R.link_mem), R.link_mem = nullptr;
    if (with_links) {
      size_t bytes = 0;
      const size_t before = xops.size();
      link_norms(xops, plan, R.links, R.shadows, &bytes);
      fuse_qkv_rope(xops, plan, &bytes, &R.rope_tab_off);
      if (route_debug()) fprintf(stderr, "route: %zu carried norms, %zu fp16 shadows, %zu -> %zu launches, %zu bytes\n", R.links.size(), R.shadows.size(), before, xops.size(), bytes);
      if (bytes && hipMalloc(reinterpret_cast<void**>(&R.link_mem), bytes) != hipSuccess) {
        (void)hipGetLastError();
        return false;
      }
    }
    {
      std::vector<char> covered(n, 0);
      int ncov = 0, maxcov = -1, xbeg = 0, obeg = 0;
      for (int e = 0; e < int(xops.size()); e++) {
        for (int q = 0; q < kXIdx; q++)
          if (xops[e].idx[q] >= 0 && !covered[xops[e].idx[q]]) covered[xops[e].idx[q]] = 1, ncov++, maxcov = std::max(maxcov, xops[e].idx[q]);
        const bool prefix = ncov == maxcov + 1;
        const uint32_t lastk = xops[e].xk == XK_OP ? plan[xops[e].idx[0]].op.kind : 0u;
        const bool enough = e + 1 - xbeg >= (segs.empty() ? seg0_ops() : seg_ops()) && int(xops.size()) - (e + 1) >= seg_ops() / 3;
        if (e + 1 == int(xops.size()) || (prefix && enough && lastk != RK_RMSNORM && lastk != RK_SILU)) {
          segs.push_back(Segment{obeg, maxcov + 1, xbeg, e + 1, nullptr});
          xbeg = e + 1, obeg = maxcov + 1;
        }
      }
      if (ncov != int(n)) return false;  // (cannot happen: every op is in exactly one launch)
    }
    bool ok = true;
    for (size_t sg = 0; sg < segs.size() && ok; sg++) {
      hipGraph_t graph = nullptr;
      if (hipStreamBeginCapture(R.st, hipStreamCaptureModeRelaxed) != hipSuccess) {
        ok = false;
        break;
      }
      if (sg == 0) hipLaunchKernelGGL(route_count_kernel, dim3(1), dim3(1), 0, R.st, R.kdev);
      for (int e = segs[sg].xbeg; e < segs[sg].xend && ok; e++) ok = capture_xop(xops[e], plan, R.kdev, R.st) == 0;
      {
        t_in_exec = true;
        ok = ns_hip_lazy_flush() == 0 && ok;
        t_in_exec = false;
      }
      const hipError_t ec = hipStreamEndCapture(R.st, &graph);
      ok = ok && ec == hipSuccess && graph != nullptr;
      if (ok) ok = hipGraphInstantiate(&segs[sg].exec, graph, nullptr, nullptr, 0) == hipSuccess;
      if (graph) (void)hipGraphDestroy(graph);
    }
    if (!ok) {
      if (route_debug()) fprintf(stderr, "route: capture %s failed: %s\n", with_links ? "with carried norms" : "", ns_hip_last_error());
      (void)hipGetLastError();
      ns_hip_reset_error();
      for (Segment& sg : segs)
        if (sg.exec) (void)hipGraphExecDestroy(sg.exec), sg.exec = nullptr;
    }
    return ok;
  };
  const bool try_links = fuse_on() && links_on();
  if (!(try_links && attempt(true)) && !attempt(false)) {
    R.links.clear(), R.shadows.clear();
    if (R.link_mem) (void)hipFree(R.link_mem), R.link_mem = nullptr;
    stat(6);
    if (++R.failures >= 3) {  // this queue's graphs cannot be captured: its evaluations stay on the window
      R.off = true;
      fprintf(stderr, "ns route: the per-token graph of device queue %p could not be captured %d times (%s): replay is off for this queue, "
                      "its evaluations keep running as fused launches\n", (void*)R.st, R.failures, ns_hip_last_error());
    }
    return false;
  }
  if (getenv("NS_ROUTE_DUMP")) {
    static const char* names[] = {"?", "GEMM", "ADD", "MUL", "SILU", "RMSNORM", "ROPE", "ROPE_YARN", "DUP", "MHA"};
    const int lim = atoi(getenv("NS_ROUTE_DUMP"));
    for (size_t j = 0; j < n && int(j) < lim; j++) {
      const RouteOp& o = plan[j].op;
      fprintf(stderr, "route plan %3zu %-8s p %p %p %p %p  i %lld %lld %lld %lld %lld  shift %x moving %d delta %lld\n", j, names[o.kind < 10 ? o.kind : 0],
              o.p[0], o.p[1], o.p[2], o.p[3], o.i[0], o.i[1], o.i[2], o.i[3], o.i[4], plan[j].pshift, plan[j].moving, plan[j].delta);
    }
  }
  R.plan.swap(plan);
  R.segs.swap(segs);
  stat_set(5, xops.size());
  R.xops.swap(xops);
  R.act_delta = act_delta;
  R.act_ptrs.swap(act_ptrs);
  R.last_replayed = false;
  R.have_plan = true;
  R.khost = R.kdone = 0;
  R.pos = R.seg = 0;
  stat(2);
  stat_set(4, n);
  return true;
}

// ---- the window: ops handed over and not launched yet, cur[launched ..) ----
// They go out at the next synchronisation point of the queue, as the fused launches optimize() finds among them.  `src` / `bytes`: the copy that
// ends the window reads this (a fusion then keeps the tensor it would have left out).
int flush_window(const void* src, size_t bytes) {
  if (R.launched >= R.cur.size()) return 0;
  std::vector<PlanOp> w(R.cur.size() - R.launched);
  for (size_t j = 0; j < w.size(); j++) w[j] = PlanOp{R.cur[R.launched + j], 0, 0, 0u};
  t_extra = ExtraRead{static_cast<const char*>(src), bytes};
  const long long t0 = route_timing() ? now_us() : 0;
  std::vector<ExecOp> x = optimize(w);
  bool prompt_sized = false;
  for (const PlanOp& po : w) prompt_sized = prompt_sized || (po.op.kind == RK_GEMM && po.op.i[0] > 16);
  if (prompt_sized) prefill_pass(x, w, R.st);
  t_extra = ExtraRead{};
  int rc = 0;
  for (size_t e = 0; e < x.size() && rc == 0; e++) {
    rc = capture_xop(x[e], w, nullptr, R.st);
    // an attention that took the fp32 kernels left no fp16 rows behind: the projection that was promised them converts its own
    if (rc == 0 && x[e].o16p && x[e].xk == XK_OP && w[x[e].idx[0]].op.kind == RK_MHA && !g_mha_out16_written)
      for (size_t c = e + 1; c < x.size(); c++)
        if (x[c].a16p == x[e].o16p) x[c].a16p = nullptr;
  }
  if (route_timing()) R.win_t0 = t0, R.win_ops = (long long)w.size(), R.win_launches = (long long)x.size(), R.win_issue_us = now_us() - t0;
  {
    t_in_exec = true;
    rc = ns_hip_lazy_flush() != 0 ? -1 : rc;
    t_in_exec = false;
  }
  R.launched = R.cur.size();
  if (rc != 0) fprintf(stderr, "Err: invalid parameters (device route: %s)\n", ns_hip_last_error());
  return rc;
}
constexpr size_t kWindowMax = 16384;  // ops; a window this long goes out without waiting for a synchronisation (no model graph comes near)

// an evaluation that ended is evaluated AGAIN from its first launch, plainly (no plan, the fp32 kernels): R.cur holds its ops, the stash its inputs
int reevaluate() {
  if (!R.stash_complete || R.cur.empty()) return -1;
  if (restore_inputs() != 0) return -1;
  R.launched = 0;
  return flush_window(nullptr, 0);
}

}  // namespace

bool route_executing() { return t_in_exec; }
void route_attach(void* stream) {
  if (find_route(stream)) return;
  Route* r = new Route();
  r->st = static_cast<hipStream_t>(stream);
  g_routes.push_back(r);
}
void route_detach(void* stream) {
  Route* r = find_route(stream);
  if (!r) return;
  t_R = r;
  if (route_timing() && R.gpu_tokens)
    fprintf(stderr, "route timing: %lld replayed tokens, GPU span first segment -> last %.1f us per token, host first launch -> last launch %.1f us\n",
            R.gpu_tokens, 1e3 * R.gpu_ms_sum / R.gpu_tokens, R.host_us_sum / R.gpu_tokens);
  if (route_timing() && R.tm_tokens)
    fprintf(stderr, "route timing: per replayed token (%lld): previous token's end -> first op handed over %.1f us (the reference samples, looks the embedding up, "
                    "builds its graph) | -> first segment launched %.1f | -> last segment launched %.1f | -> the token's synchronisation entered %.1f | "
                    "-> returned (GPU finishing) %.1f | -> logits copied, token ended %.1f\n",
            R.tm_tokens, R.tm_sum[0] / R.tm_tokens, R.tm_sum[1] / R.tm_tokens, R.tm_sum[2] / R.tm_tokens, R.tm_sum[3] / R.tm_tokens, R.tm_sum[4] / R.tm_tokens,
            R.tm_sum[5] / R.tm_tokens);
  drop_plan();
  if (R.stash_mem) (void)hipFree(R.stash_mem);
  if (R.kdev) (void)hipFree(R.kdev);
  if (R.ev0) (void)hipEventDestroy(R.ev0), (void)hipEventDestroy(R.ev1);
  g_routes.erase(std::find(g_routes.begin(), g_routes.end(), r));
  delete r;
  t_R = nullptr;
}
bool route_hook(void* stream) {
  if (t_in_exec) return false;
  Route* r = find_route(stream);
  if (!r) return false;
  t_R = r;
  return window_on() || enabled();
}

int route_submit(const RouteOp& op) {
  if (route_timing() && R.cur.empty()) R.tm[TM_FIRST_OP] = now_us(), R.tm_replayed = false;
  if (R.have_plan) {
    if (R.pos == 0) R.khost++, R.last_replayed = false;
    if (R.pos < int(R.plan.size())) {
      const RouteOp e = expected(R.plan[R.pos], R.khost);
      if (memcmp(&e, &op, sizeof(RouteOp)) == 0) {
        R.cur.push_back(op);
        R.pos++;
        if (R.pos == R.segs[R.seg].end) {
          if (route_timing() && R.seg == 0) {
            if (!R.ev0) (void)hipEventCreate(&R.ev0), (void)hipEventCreate(&R.ev1);
            (void)hipEventRecord(R.ev0, R.st);
            R.t_first_us = R.tm[TM_FIRST_LAUNCH] = now_us();
          }
          if (hipGraphLaunch(R.segs[R.seg].exec, R.st) != hipSuccess) {
            set_error("device route: launching a replayed segment failed");
            return -1;
          }
          R.seg++;
          R.launched = size_t(R.pos);
          if (route_timing() && R.seg == int(R.segs.size())) {
            (void)hipEventRecord(R.ev1, R.st);
            R.ev_pending = true;
            R.tm[TM_LAST_LAUNCH] = now_us();
            R.host_us_sum += double(R.tm[TM_LAST_LAUNCH] - R.t_first_us);
          }
        }
        return 0;
      }
    }
    if (route_debug()) {
      fprintf(stderr, "route: token %lld of the plan deviates at launch %d of %zu\n", R.khost, R.pos, R.plan.size());
      if (R.pos < int(R.plan.size())) explain_mismatch(size_t(R.pos), expected(R.plan[R.pos], R.khost), op);
    }
    if (bail_out() != 0) return -1;  // (this op is not in R.cur yet)
  }
  R.cur.push_back(op);
  if (!window_on()) {  // round-5 behaviour: launched as it comes
    R.launched = R.cur.size();
    return execute(op, R.st);
  }
  return R.cur.size() - R.launched >= kWindowMax ? flush_window(nullptr, 0) : 0;
}

// a synchronisation point of the route's stream: everything handed over so far must be on the stream; a non-empty trace ends a token.
// `src` / `bytes`: the device-side source of the copy that follows (nullptr: a plain synchronisation)
int route_sync_point(void* stream, const void* src, size_t bytes) {
  if (t_in_exec) return 0;
  Route* r = find_route(stream);
  if (!r) return 0;
  t_R = r;
  if (!window_on() && !enabled()) return 0;
  int rc = 0;
  if (R.have_plan) {
    if (R.pos == int(R.plan.size())) {  // the whole token matched: every segment is on the stream
      stat(0);
      R.kdone = R.khost;
      R.pos = R.seg = 0;
      R.launched = 0;
      R.prev.swap(R.cur);
      R.cur.clear();
      R.last_replayed = true;
      R.tm_replayed = true;
      R.stash_stale = R.tm_prev_pending = true;
      if (R.bails_in_a_row && R.kdone >= 8) R.bails_in_a_row = 0;
      if (R.ev_pending) {
        float ms = 0.f;
        if (hipEventSynchronize(R.ev1) == hipSuccess && hipEventElapsedTime(&ms, R.ev0, R.ev1) == hipSuccess) R.gpu_ms_sum += ms, R.gpu_tokens++;
        R.ev_pending = false;
      }
      return 0;
    }
    if (R.pos > 0) rc = bail_out();  // the token ended (or synchronised) inside the plan
  }
  if (!R.cur.empty()) {
    if (flush_window(src, bytes) != 0) rc = -1;
    // the evaluation ends here (its trace is complete): a plan is made from two consecutive agreeing ones
    stat(1);
    if (!R.have_plan && rc == 0 && enabled()) {
      if (R.plan_pause > 0 && R.cur.size() == R.prev.size()) R.plan_pause--;
      else if (R.plan_pause == 0) (void)make_plan();
    }
    R.prev.swap(R.cur);
    R.cur.clear();
    R.launched = 0;
    R.stash_stale = R.tm_prev_pending = true;
  }
  return rc;
}

// bestla_device_sync has waited for the queue: the end of an evaluation.  A value that did not fit fp16 in one of the route's fp16 shortcuts (the kv
// mirror, a carried norm's shadow) raised the flag: the shortcuts are turned off for the process and the evaluation that just ended — its ops are in
// `prev`, its inputs in the stash — is evaluated again on the fp32 forms before anybody reads its results.
bool route_after_sync(void* stream) {
  if (t_in_exec) return false;
  Route* r = find_route(stream);
  if (!r) return false;
  t_R = r;
  R.copies_pending = 0;
  if (route_timing() && R.tm[TM_SYNC_OUT] < R.tm[TM_SYNC_IN]) R.tm[TM_SYNC_OUT] = now_us();
  if (route_timing() && R.win_t0 && R.win_ops >= 64) {  // a window went out and the queue has been waited for: where an evaluation that is not replayed spends its time
    fprintf(stderr, "route timing: window of %lld ops -> %lld launches: first op handed over %.1f us before the flush, issued in %.1f us, queue empty %.1f us after the flush began\n",
            R.win_ops, R.win_launches, double(R.win_t0 - R.tm[TM_FIRST_OP]), double(R.win_issue_us), double(now_us() - R.win_t0));
    R.win_t0 = 0;
  }
  if (!kvm_overflowed()) return false;
  kvm_overflow_reset();
  const bool had_f16 = kv16_enabled() || links_on();
  kv16_set(0);
  g_links_enabled.store(0);
  if (!had_f16) return false;
  bool again = false;
  fprintf(stderr, "ns route: a value beyond the fp16 range (|x| > 65504) met an fp16 shortcut of the device route (the kv mirror or a carried norm's "
                  "shadow): both are off for this process from here on (the fp32 kernels of the reference's device path take over), the evaluation is run again\n");
  for (Route* q : g_routes) {
    t_R = q;
    const bool mine = q == r && !q->prev.empty() && q->cur.empty();
    drop_plan();
    if (mine) {
      R.cur.swap(R.prev);  // the evaluation that just ended
      const int rc = reevaluate();
      R.prev.swap(R.cur);
      R.cur.clear();
      R.launched = 0;
      if (rc != 0) fprintf(stderr, "ns route: the evaluation could not be run again (%s): its results hold the overflow (inf / nan)\n", ns_hip_last_error());
      (void)hipStreamSynchronize(R.st);
      kvm_overflow_reset();
      again = rc == 0;
    }
  }
  t_R = r;
  return again;
}
// bestla_device_sync with nothing but launches on the queue since it was last waited for: nothing the host can see depends on them before a copy
// is asked for — and that copy is ordered behind them on the queue.  The reference's evaluation ends with sync, copy, sync (ne_layers.c:8345-8346): the
// first of the two waits is left to the second.  For a token this measured level (2074 vs 2065 / 2088 us at 1500 cached positions).  For a prompt it is
// what lets the copy's host-side preparation run UNDER the launches instead of behind them: the copy of a 1500-token prompt's logits (192 MB into pages
// the reference has not touched yet) arrives while the queue still has 28 ms of work, the runtime maps the destination and ns_device.hip first-touches
// it meanwhile — first evaluation of a process 59.0 -> 52.1 (deferral alone) -> 46.2 / 48.4 ms on one box (profiles/r06_route_timings.txt), later
// evaluations unchanged.  NS_ROUTE_LAZY_SYNC=0 waits where the caller asked.
bool route_defer_sync(void* stream) {
  static const bool on = !getenv("NS_ROUTE_LAZY_SYNC") || atoi(getenv("NS_ROUTE_LAZY_SYNC")) != 0;
  if (!on || t_in_exec) return false;
  Route* r = find_route(stream);
  return r && r->copies_pending == 0 && (window_on() || g_enabled.load() != 0);
}
void route_note_copy(void* stream) {
  Route* r = find_route(stream);
  if (r) r->copies_pending++;
}
void route_time_mark(void* stream, int what) {  // NS_ROUTE_TIMING: ns_device.hip marks a token's synchronisation and its end
  if (!route_timing() || t_in_exec) return;
  Route* r = find_route(stream);
  if (!r) return;
  const long long now = now_us();
  if (what == 0) {
    if (!r->tm[TM_SYNC_IN] || r->tm[TM_SYNC_IN] < r->tm[TM_LAST_LAUNCH]) r->tm[TM_SYNC_IN] = now;
  } else if (what == 1) {  // a copy has completed: if it ended a replayed token, the token's marks are summed
    if (r->tm_replayed && r->cur.empty() && r->tm[TM_PREV_END] && r->tm[TM_FIRST_OP] > r->tm[TM_PREV_END] && r->tm[TM_SYNC_OUT] >= r->tm[TM_SYNC_IN]) {
      const long long m[TM_N + 1] = {r->tm[TM_PREV_END], r->tm[TM_FIRST_OP], r->tm[TM_FIRST_LAUNCH], r->tm[TM_LAST_LAUNCH], r->tm[TM_SYNC_IN], r->tm[TM_SYNC_OUT], now};
      bool ordered = true;
      for (int i = 0; i < TM_N; i++) ordered = ordered && m[i + 1] >= m[i];
      if (ordered) {
        for (int i = 0; i < TM_N; i++) r->tm_sum[i] += double(m[i + 1] - m[i]);
        r->tm_tokens++;
      }
      r->tm_replayed = false;
    }
    if (r->cur.empty() && r->tm_prev_pending) r->tm[TM_PREV_END] = now, r->tm_prev_pending = false;
  }
}

// bestla_device_memcpy on the route's queue while a plan is held.  The copy that brings a token's embeddings has an activation of the NEXT
// token as its destination (expected at plan address + act_delta * (k + 1)): the bytes go where they were asked for AND to the plan's twin of
// that tensor (returned here; nullptr: no twin), so a token that is replayed finds them and a token that falls back does too.  The copy that
// fetches a replayed token's logits reads the plan's address.  Pointers that are not the plan's activations pass through.
void* route_twin_dst(void* dst, void* stream) {
  Route* r = find_route(stream);
  if (t_in_exec || !r || !r->have_plan || r->act_delta == 0 || r->pos != 0) return nullptr;
  void* cand = static_cast<char*>(dst) - r->act_delta * (r->khost + 1);
  return r->act_ptrs.count(cand) ? cand : nullptr;
}
const void* route_translate_src(const void* src, void* stream) {
  Route* r = find_route(stream);
  if (t_in_exec || !r || !r->have_plan || r->act_delta == 0 || !r->last_replayed) return src;
  const void* cand = static_cast<const char*>(src) - r->act_delta * r->khost;
  return r->act_ptrs.count(cand) ? cand : src;
}
// a copy INTO device memory on the route's queue has been issued.  In front of an evaluation's first op it brings an input of that evaluation:
// a device-side copy is kept (D2D behind it on the queue) so that the evaluation can be issued again from its start (bail_out, route_after_sync).
void route_note_input(void* dst, size_t bytes, void* stream) {
  Route* r = find_route(stream);
  if (t_in_exec || !r || !dst || !bytes) return;
  t_R = r;
  if (!R.cur.empty() || R.pos != 0) return;  // (cannot happen: a copy is a synchronisation point, which ends the trace)
  if (R.stash_stale) R.stash.clear(), R.stash_used = 0, R.stash_complete = true, R.stash_stale = false;
  if (bytes > (size_t(64) << 20)) return;  // (a model part being uploaded, not an evaluation's input: nothing an evaluation writes over)
  const size_t need = ((R.stash_used + 255) & ~size_t(255)) + bytes;
  if (need > R.stash_cap) {  // grows; what this evaluation has kept so far moves along
    char* bigger = nullptr;
    const size_t want = std::max<size_t>(need * 2, size_t(1) << 20);
    if (want > (size_t(512) << 20) || hipMalloc(reinterpret_cast<void**>(&bigger), want) != hipSuccess) {
      (void)hipGetLastError();
      R.stash_complete = false;
      return;
    }
    if (R.stash_used) (void)hipMemcpyAsync(bigger, R.stash_mem, R.stash_used, hipMemcpyDeviceToDevice, R.st);
    (void)hipStreamSynchronize(R.st);
    if (R.stash_mem) (void)hipFree(R.stash_mem);
    R.stash_mem = bigger, R.stash_cap = want;
  }
  const size_t off = (R.stash_used + 255) & ~size_t(255);
  if (off + bytes > R.stash_cap || hipMemcpyAsync(R.stash_mem + off, dst, bytes, hipMemcpyDeviceToDevice, R.st) != hipSuccess) {
    (void)hipGetLastError();
    R.stash_complete = false;
    return;
  }
  R.stash.push_back(Stash{dst, bytes, off});
  R.stash_used = off + bytes;
}
void route_invalidate() {  // device memory is being freed: no captured launch, no recorded op may outlive it
  if (t_in_exec) return;
  Route* keep = t_R;
  for (Route* r : g_routes) {
    t_R = r;
    drop_plan();
    R.cur.clear(), R.prev.clear();
    R.launched = 0;
  }
  t_R = keep;
}

}  // namespace ns

extern "C" void ns_hip_route_stats(uint64_t out[8]) {
  for (int i = 0; i < 8; i++) out[i] = ns::g_stats[i];
  out[7] = 0;
  for (const ns::Route* r : ns::g_routes) out[7] |= r->have_plan ? 1 : 0;
}
extern "C" int ns_hip_route_set_enabled(int on) {
  int prev = ns::g_enabled.load();
  if (prev < 0) {
    const char* e = getenv("NS_DEVICE_REPLAY");
    prev = e ? atoi(e) != 0 : 1;
  }
  // 0: the layer is off (every op launches when it is handed over); 1: window + plans (3: with carried norms, 5: without, 1: NS_ROUTE_LINKS, default on);
  // 8: the window alone (evaluations go out fused, no plan is made)
  if (!(on & 1)) ns::route_invalidate();
  ns::g_enabled.store((on & 1) ? 1 : 0);
  ns::g_window.store(on ? 1 : 0);
  if (on & 1) ns::g_links_enabled.store((on & 2) ? 1 : (on & 4) ? 0 : -1);
  for (ns::Route* r : ns::g_routes) r->failures = 0, r->off = false, r->bails_in_a_row = r->plan_pause = 0;
  return prev;
}
