// glue/shim/bestla/bestla_parallel.h — stands in for /root/reference/bestla/bestla/bestla_parallel.h (which pulls
// bestla_device.h and, through it, the xbyak JIT headers) when neural-speed's MODEL code is compiled against libns_hip.so.
// The model code uses exactly one thing from it: `bestla::parallel::IThreading::parallel_for_collapse` on the pointer
// that `bestla_get_thread_handle()` returns (models/llama/llama.cpp:750-757 and the same lines in chatglm / gptneox /
// gptj; models/model_utils/model_utils.cpp:1978-1990, :2169, :2333, :2970 through ne_bestla::ne_threading::get()) — host
// loops over a handful of memcpy calls (logits of each batch entry, kv rows of a beam).
//
// With this backend the handle is the library's opaque context (include/ns_bestla.h: bestla_get_thread_handle), not a
// thread pool: the GEMMs fan out on the GPU, the graph runs on one host thread (glue/ne_bestla_hip_glue.c).  So the two
// collapse loops are plain loops that never touch `this`, which makes them valid on ANY handle value.
#pragma once
#include <functional>

namespace bestla {
namespace parallel {

using thread_func = std::function<void(int)>;

class IThreading {
 public:
  // for (i = begin1; i < end1; i += step1) func(i)
  void parallel_for_collapse(const int& begin1, const int& end1, const int& step1, const std::function<void(int)>& func) {
    for (int i = begin1; i < end1; i += step1) func(i);
  }
  // for (i ...) for (j ...) func(i, j)
  void parallel_for_collapse(const int& begin1, const int& end1, const int& step1, const int& begin2, const int& end2,
                             const int& step2, const std::function<void(int, int)>& func) {
    for (int i = begin1; i < end1; i += step1)
      for (int j = begin2; j < end2; j += step2) func(i, j);
  }
  void parallel_for(const thread_func& func) { func(0); }
  void sync(int /*tidx*/, int /*idx*/ = 0) {}
  int num_threads() const { return 1; }
  void set_threads(int /*nthreads*/) {}
};

}  // namespace parallel
}  // namespace bestla
