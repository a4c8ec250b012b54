// ns_route.h — what the device route's pieces share (ns_route.cpp, ns_device.hip, the operator kernels of ns_quant.hip): the fp16
// mirror of the reference's fp32 device kv cache (round 6) and the per-token bookkeeping around it.
//
// Why a mirror: a reference tree built with its device switch keeps its kv cache in fp32 on the device — K [batch][heads_kv][n_ctx][head_size],
// V transposed [batch][heads_kv][head_size][n_ctx] (/root/reference/neural_speed/models/llama/llama.cpp:241-285, core/layers/ne_bestla_sycl.cpp:592-880).
// At 1500 cached positions a decode step streams 49 MB of K / V per layer from it.  The reference's own DEFAULT caches are fp16
// (models/model_utils/model_types.h:96-100 KV_MEM_TYPE_AUTO: the BesTLA-managed cache, else fp16; model_utils.cpp:252, :1072-1076), and this library's
// tuned attention kernels (ns_attn.hip: LDS-ring decode kernel, matrix-core prefill kernels) read fp16 rows [head][position][head_size].  The mirror is that:
// an fp16 copy of the cache in ONE layout for K and V, kept beside the fp32 tensors, which stay the reference's (every cell is still written as the
// graph asks: whoever copies the cache out sees what the reference's device build would have put there).  Attention reads the mirror on EVERY path —
// prompt, eager decode step, replayed decode step — so a chunked prompt and a token-by-token evaluation see the same K / V (round 5 rounded
// a prompt's K / V to fp16 and left the decode steps on fp32).  NS_DEVICE_KV=f32 (ns_hip_set_tuning("device_kv_f16", 0)) keeps the fp32 kernels everywhere:
// bit-for-bit the numerics of the reference's device kernel, at its speed.
//
// How it stays coherent:
//   * eager calls (ns_hip_mha_f32_device_layout): positions [lo, seq_all) are converted in front of the attention launch, lo = min(valid, seq_all - seq)
//     — the rows this evaluation wrote are always converted again, older rows once;
//   * replayed decode steps: the captured cache-write launch (rope_append_kernel / dup2_kernel) stores every cell also into the mirror, addressed
//     from the CELL's address (KvMirrorArgs below), and `valid` follows the token counter on the host;
//   * any other write into a mirrored cache (a memcpy, an in-place operator) resets `valid` to 0: the next attention converts the whole live range.
// Values beyond fp16's range (|x| > 65504) cannot be mirrored: the converting kernels raise a flag in pinned host memory, the route reads it at the
// token's end (behind the synchronisation the reference issues there anyway), turns the mirror off for the process with one line on stderr and
// evaluates the token again on the fp32 kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace ns {

// one destination of a cache-writing launch: cells inside [base32, base32 + 4 * elems) are mirrored to m16
struct KvMirrorArgs {
  const char* base32 = nullptr;  // the fp32 cache (K: [slots][heads_kv][n_ctx][hs]; V: [slots][heads_kv][hs][n_ctx])
  _Float16* m16 = nullptr;       // the mirror, [slots][heads_kv][n_ctx][hs] for both
  long long elems = 0;
  int n_ctx = 0, hs = 0;
  int transposed = 0;            // 1: the fp32 cache is the V layout
  uint32_t* overflow = nullptr;  // pinned host word: set when a value does not fit fp16
};
struct KvMirrorPair {
  KvMirrorArgs k, v;
};
extern thread_local KvMirrorPair g_kvm;  // set around a captured cache-write launch (like g_affine), zero otherwise

bool route_executing();  // ns_route.cpp: the calling thread is inside a launch the route issues (its cache protocol is known)

// ---- ns_device.hip ----
bool kv16_enabled();          // NS_DEVICE_KV / ns_hip_set_tuning("device_kv_f16")
void kv16_set(int on);        // -1: environment / default (on)
// the mirror behind the fp32 cache cell at `cell` (a cpy node's destination): fills `out`, false when no mirror holds it
bool kvm_args_for_cell(const void* cell, KvMirrorArgs* out);
// an operator other than a recognised cache write / a copy stores to `dst`: mirrors that hold it start over (true: there was one — a plan that
// keeps a mirror current by its own cache writes must not go on: the caller drops it)
bool kvm_note_foreign_write(const void* dst, size_t bytes);
// a replayed decode step advanced the mirrors its plan writes: positions [0, valid) of slot 0.. hold the cache
void kvm_set_valid(const void* k32, int valid);
// a producer is about to store positions [n_past, n_past + m) of slot 0.. of the cache at (k32, v32) into its mirror itself (the fused QKV launch's epilogue at
// prompt size): the mirror (made when there is none) and where its rows start; the attention that follows finds those rows in place.  false: no mirror
bool kvm_for_producer(const void* k32, const void* v32, int heads_kv, int hs, int n_ctx, int n_past, int m, _Float16** k16, _Float16** v16);
void kvm_clear();             // device memory is being freed
uint32_t* kvm_overflow_word();  // pinned; nullptr when it could not be allocated
bool kvm_overflowed();        // reads (and leaves) the flag
void kvm_overflow_reset();

}  // namespace ns
