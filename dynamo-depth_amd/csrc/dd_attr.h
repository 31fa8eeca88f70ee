// dd_attr.h -- hipFuncAttributeMaxDynamicSharedMemorySize belongs to (kernel, DEVICE): a process that drives a second GPU
// (opt.cuda_ids, tests that switch devices) has to set it there too, and launches come from several host threads (the autograd
// workers of the multi-stream backward).  One object per kernel instantiation: a bit per device ordinal, set with release order
// behind the successful call; a thread that loses the race sets the attribute a second time, which is harmless.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>

namespace dd {

struct LdsAttrOnce {
  std::atomic<unsigned long long> done{0};
  int ensure(const void* kern, int bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return 0;
    e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return (int)e;
    done.fetch_or(bit, std::memory_order_release);
    return 0;
  }
};

}  // namespace dd
