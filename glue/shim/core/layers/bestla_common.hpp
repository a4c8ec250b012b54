// glue/shim/core/layers/bestla_common.hpp — stands in for /root/reference/neural_speed/core/layers/bestla_common.hpp when
// neural-speed's MODEL code (models/*/*.cpp, models/model_utils/*) is compiled against libns_hip.so.  The reference's
// header exists for the CPU kernels (it includes bestla_prologue_b.h / bestla_device.h, i.e. the JIT GEMM and xbyak, and
// defines the epilogue / ISA helpers of core/layers/*.cpp, which this backend replaces wholesale); the model code needs
// three things from it:
//   * bestla::utils::amalloc / afree  (models/model_utils/model_files.h:1516, :1526; util.h:410-415) — the reference's own
//     bestla_utils.h, which compiles standalone, provides them;
//   * bestla::parallel::IThreading    (llama.cpp:750-757) — glue/shim/bestla/bestla_parallel.h;
//   * ne_bestla::ne_threading::get()  (model_utils.cpp:1978 ...) — below.
// Put `-I <this repo>/glue/shim` in front of the reference's include directories; nothing else changes
// (oracle/Makefile target nellama does exactly that with the reference's sources compiled from where they lie).
#pragma once
#include "ne_bestla.h"
#include "bestla/bestla_utils.h"
#include "bestla/bestla_parallel.h"

namespace ne_bestla {

class ne_threading {
 public:
  static bestla::parallel::IThreading* get() { return reinterpret_cast<bestla::parallel::IThreading*>(bestla_get_thread_handle()); }
  static void set_threads(int n_thread) { bestla_set_threads(n_thread); }
};

}  // namespace ne_bestla
