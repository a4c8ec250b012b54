/*
 * TEST INFRASTRUCTURE — stand-in for xbyak's xbyak_util.h (herumi/xbyak, the reference's un-vendored third-party
 * dependency; bestla_device.h:19 uses its `Xbyak::util::Cpu` for CPU detection only).  Written against the members
 * bestla_device.h calls: has(t<ISA>), getNumCores(level), getCpuid / getCpuidEx, getDataCacheSize(level).  The feature
 * bits come from CPUID leaf 1 / 7 as the SDM defines them.  NS_PACKREF_ISA=nosimd makes has() answer false for every
 * vector ISA: the reference's runtime dispatch (kernel_wrapper.h forward_auto) then takes its scalar kernels — the ones
 * the oracle restates — on any host.
 */
#pragma once
#include <cpuid.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace Xbyak {
namespace util {

enum IntelCpuTopologyLevel { SmtLevel = 1, CoreLevel = 2 };

class Cpu {
 public:
  typedef uint64_t Type;
  static constexpr Type tAVX = 1, tAVX2 = 2, tAVX512F = 3, tAVX512BW = 4, tAVX512_VNNI = 5, tAVX_VNNI = 6, tAMX_BF16 = 7,
                        tAMX_INT8 = 8, tAVX512_BF16 = 9, tAVX512_FP16 = 10, tAMX_FP16 = 11;

  static void cpuid(uint32_t leaf, uint32_t sub, uint32_t r[4]) { __cpuid_count(leaf, sub, r[0], r[1], r[2], r[3]); }

  bool has(Type t) const {
    const char* force = getenv("NS_PACKREF_ISA");
    if (force && !strcmp(force, "nosimd")) return false;
    uint32_t r1[4], r7[4], r71[4];
    cpuid(1, 0, r1);
    cpuid(7, 0, r7);
    cpuid(7, 1, r71);
    switch (t) {
      case tAVX: return (r1[2] >> 28) & 1;
      case tAVX2: return (r7[1] >> 5) & 1;
      case tAVX512F: return (r7[1] >> 16) & 1;
      case tAVX512BW: return (r7[1] >> 30) & 1;
      case tAVX512_VNNI: return (r7[2] >> 11) & 1;
      case tAVX_VNNI: return (r71[0] >> 4) & 1;
      case tAVX512_BF16: return (r71[0] >> 5) & 1;
      case tAVX512_FP16: return (r7[3] >> 23) & 1;
      // the AMX tile state needs an arch_prctl permission request the pins have no use for: reported absent
      case tAMX_BF16: case tAMX_INT8: case tAMX_FP16: return false;
      default: return false;
    }
  }
  int getNumCores(IntelCpuTopologyLevel level) const {
    return level == SmtLevel ? 1 : static_cast<int>(std::thread::hardware_concurrency());
  }
  void getCpuid(uint32_t leaf, uint32_t r[4]) const { cpuid(leaf, 0, r); }
  void getCpuidEx(uint32_t leaf, uint32_t sub, uint32_t r[4]) const { cpuid(leaf, sub, r); }
  uint32_t getDataCacheSize(int level) const {  // only sizes work partitioning; nominal values
    static const uint32_t sz[3] = {48u << 10, 2u << 20, 32u << 20};
    return level >= 0 && level < 3 ? sz[level] : 0;
  }
};

}  // namespace util
}  // namespace Xbyak
