// Shared helpers for libvqvae_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/vqvae_hip.h"

namespace vq {

void set_error(const char* fmt, ...);

#define VQ_CHECK_HIP(expr)                                                        \
  do {                                                                            \
    hipError_t _e = (expr);                                                       \
    if (_e != hipSuccess) {                                                       \
      vq::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr,                 \
                    hipGetErrorString(_e));                                       \
      return (int)_e;                                                             \
    }                                                                             \
  } while (0)

#define VQ_REQUIRE(cond, ...)                                                     \
  do {                                                                            \
    if (!(cond)) {                                                                \
      vq::set_error(__VA_ARGS__);                                                 \
      return VQVAE_E_INVALID;                                                     \
    }                                                                             \
  } while (0)

#define VQ_LAUNCH_CHECK()                                                         \
  do {                                                                            \
    hipError_t _e = hipGetLastError();                                            \
    if (_e != hipSuccess) {                                                       \
      vq::set_error("%s:%d: kernel launch -> %s", __FILE__, __LINE__,             \
                    hipGetErrorString(_e));                                       \
      return (int)_e;                                                             \
    }                                                                             \
  } while (0)

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// profiling hooks (runtime.hip): bracket a launch with events when enabled
void prof_begin(int tag, hipStream_t s);
void prof_end(int tag, hipStream_t s);

struct ProfScope {
  int tag; hipStream_t s;
  ProfScope(int t, hipStream_t st) : tag(t), s(st) { if (tag) prof_begin(tag, s); }
  ~ProfScope() { if (tag) prof_end(tag, s); }
};

// A launch timed by the events of its OWN dispatch packet (hipExtLaunchKernelGGL(start, stop)): no event packets
// before and after the kernel on the stream -- those cost ~5.6 us of idle queue each, 0.22 ms per step when the 20 gate
// launches of the bench are timed (tools/phase_trace.py).  prof_attach returns false when the tag is not enabled.
bool prof_attach(int tag, hipEvent_t* start, hipEvent_t* stop);
bool prof_enabled(int tag);

}  // namespace vq
