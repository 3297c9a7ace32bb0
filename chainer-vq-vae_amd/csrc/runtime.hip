// Device / memory / stream / event plumbing and the per-kernel event profiler.
#include "common.h"
#include <stdarg.h>
#include <mutex>
#include <vector>

namespace vq {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---- profiler: pairs of events per tagged launch, read back on demand -------
struct ProfPair { hipEvent_t a, b; int tag; };
static unsigned g_prof_mask = 0;   // bit t enables tag t
static std::vector<ProfPair> g_prof_pairs;
static std::vector<hipEvent_t> g_prof_free;
static hipEvent_t g_prof_open[VQVAE_PROF_NTAGS];
static std::mutex g_prof_mu;

static hipEvent_t prof_get_event() {
  if (!g_prof_free.empty()) {
    hipEvent_t e = g_prof_free.back();
    g_prof_free.pop_back();
    return e;
  }
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}

void prof_begin(int tag, hipStream_t s) {
  if (tag <= 0 || tag >= VQVAE_PROF_NTAGS || !(g_prof_mask & (1u << tag))) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  hipEvent_t e = prof_get_event();
  if (!e) return;
  (void)hipEventRecord(e, s);
  g_prof_open[tag] = e;
}

bool prof_enabled(int tag) { return tag > 0 && tag < VQVAE_PROF_NTAGS && (g_prof_mask & (1u << tag)) != 0; }

// (call it immediately in front of the launch the events go to: a pair registered for a launch that then never happens
// would hold two never-recorded -- or, recycled from an earlier profile, stale -- events)
bool prof_attach(int tag, hipEvent_t* start, hipEvent_t* stop) {
  if (tag <= 0 || tag >= VQVAE_PROF_NTAGS || !(g_prof_mask & (1u << tag))) return false;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  hipEvent_t a = prof_get_event(), b = prof_get_event();
  if (!a || !b) return false;
  *start = a; *stop = b;
  g_prof_pairs.push_back({a, b, tag});
  return true;
}
void prof_end(int tag, hipStream_t s) {
  if (tag <= 0 || tag >= VQVAE_PROF_NTAGS || !(g_prof_mask & (1u << tag))) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (!g_prof_open[tag]) return;
  hipEvent_t e = prof_get_event();
  if (!e) return;
  (void)hipEventRecord(e, s);
  g_prof_pairs.push_back({g_prof_open[tag], e, tag});
  g_prof_open[tag] = nullptr;
}

}  // namespace vq

using namespace vq;

extern "C" {

const char* vqvae_last_error_string(void) { return g_err; }
int vqvae_abi_version(void) { return 5; }

int vqvae_device_count(int* n) {
  VQ_REQUIRE(n, "vqvae_device_count: null");
  hipError_t e = hipGetDeviceCount(n);
  if (e != hipSuccess) { *n = 0; set_error("hipGetDeviceCount: %s", hipGetErrorString(e)); return (int)e; }
  return 0;
}
int vqvae_set_device(int dev) { VQ_CHECK_HIP(hipSetDevice(dev)); return 0; }

int vqvae_device_info(char* name, int cap, int* n_cu, size_t* total_mem) {
  int dev = 0;
  VQ_CHECK_HIP(hipGetDevice(&dev));
  hipDeviceProp_t p;
  VQ_CHECK_HIP(hipGetDeviceProperties(&p, dev));
  if (name && cap > 0) { snprintf(name, cap, "%s (%s)", p.name, p.gcnArchName); }
  if (n_cu) *n_cu = p.multiProcessorCount;
  if (total_mem) *total_mem = p.totalGlobalMem;
  return 0;
}

int vqvae_device_pci_bus_id(char* out, int cap) {       // "0000:c1:00.0": which sysfs device (NUMA node) the current GPU is
  VQ_REQUIRE(out && cap >= 16, "device_pci_bus_id: buffer of >= 16 bytes");
  int dev = 0;
  VQ_CHECK_HIP(hipGetDevice(&dev));
  VQ_CHECK_HIP(hipDeviceGetPCIBusId(out, cap, dev));
  return 0;
}

int vqvae_malloc(void** p, size_t bytes) {
  VQ_REQUIRE(p, "vqvae_malloc: null");
  VQ_CHECK_HIP(hipMalloc(p, bytes ? bytes : 4));
  return 0;
}
int vqvae_free(void* p) { if (p) VQ_CHECK_HIP(hipFree(p)); return 0; }

int vqvae_memcpy_h2d(void* dst, const void* src, size_t bytes, vqvae_stream_t s) {
  if (!bytes) return 0;
  VQ_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)s));
  VQ_CHECK_HIP(hipStreamSynchronize((hipStream_t)s));   // pageable host memory: keep it simple & safe
  return 0;
}
// Page-locked host memory + a copy that only ENQUEUES: the input leg of a training step (updaters.py:8: the
// converter's host -> device copy) on a copy stream, overlapped with the previous step's kernels.
int vqvae_host_alloc(void** p, size_t bytes) {
  VQ_REQUIRE(p, "vqvae_host_alloc: null");
  VQ_CHECK_HIP(hipHostMalloc(p, bytes ? bytes : 4, hipHostMallocDefault));
  return 0;
}
int vqvae_host_free(void* p) { if (p) VQ_CHECK_HIP(hipHostFree(p)); return 0; }
int vqvae_memcpy_h2d_async(void* dst, const void* pinned_src, size_t bytes, vqvae_stream_t s) {
  if (!bytes) return 0;
  VQ_CHECK_HIP(hipMemcpyAsync(dst, pinned_src, bytes, hipMemcpyHostToDevice, (hipStream_t)s));
  return 0;
}
int vqvae_memcpy_d2h(void* dst, const void* src, size_t bytes, vqvae_stream_t s) {
  if (!bytes) return 0;
  VQ_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)s));
  VQ_CHECK_HIP(hipStreamSynchronize((hipStream_t)s));
  return 0;
}
int vqvae_memcpy_d2d(void* dst, const void* src, size_t bytes, vqvae_stream_t s) {
  if (!bytes) return 0;
  VQ_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)s));
  return 0;
}
int vqvae_memcpy2d_d2d(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, vqvae_stream_t s) {
  if (!width || !height) return 0;
  VQ_CHECK_HIP(hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, hipMemcpyDeviceToDevice, (hipStream_t)s));
  return 0;
}
int vqvae_memset(void* p, int v, size_t bytes, vqvae_stream_t s) {
  if (!bytes) return 0;
  VQ_CHECK_HIP(hipMemsetAsync(p, v, bytes, (hipStream_t)s));
  return 0;
}
int vqvae_stream_create(vqvae_stream_t* s) {
  VQ_REQUIRE(s, "vqvae_stream_create: null");
  hipStream_t st;
  VQ_CHECK_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  *s = (vqvae_stream_t)st;
  return 0;
}
int vqvae_stream_destroy(vqvae_stream_t s) { VQ_CHECK_HIP(hipStreamDestroy((hipStream_t)s)); return 0; }
int vqvae_stream_synchronize(vqvae_stream_t s) { VQ_CHECK_HIP(hipStreamSynchronize((hipStream_t)s)); return 0; }
int vqvae_device_synchronize(void) { VQ_CHECK_HIP(hipDeviceSynchronize()); return 0; }

int vqvae_event_create(void** ev) {
  VQ_REQUIRE(ev, "vqvae_event_create: null");
  hipEvent_t e;
  VQ_CHECK_HIP(hipEventCreate(&e));
  *ev = (void*)e;
  return 0;
}
int vqvae_event_destroy(void* ev) { VQ_CHECK_HIP(hipEventDestroy((hipEvent_t)ev)); return 0; }
int vqvae_event_record(void* ev, vqvae_stream_t s) { VQ_CHECK_HIP(hipEventRecord((hipEvent_t)ev, (hipStream_t)s)); return 0; }
int vqvae_stream_wait_event(vqvae_stream_t s, void* ev) {
  VQ_CHECK_HIP(hipStreamWaitEvent((hipStream_t)s, (hipEvent_t)ev, 0));
  return 0;
}
int vqvae_event_synchronize(void* ev) { VQ_CHECK_HIP(hipEventSynchronize((hipEvent_t)ev)); return 0; }
int vqvae_event_elapsed_ms(float* ms, void* a, void* b) {
  VQ_CHECK_HIP(hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b));
  return 0;
}

int vqvae_prof_enable(int tag_mask) { g_prof_mask = (unsigned)tag_mask; return 0; }

int vqvae_prof_reset(void) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto& p : g_prof_pairs) { g_prof_free.push_back(p.a); g_prof_free.push_back(p.b); }
  g_prof_pairs.clear();
  for (int i = 0; i < VQVAE_PROF_NTAGS; ++i) g_prof_open[i] = nullptr;
  return 0;
}

int vqvae_prof_read(int tag, double* total_ms, int* launches) {
  VQ_REQUIRE(total_ms && launches, "vqvae_prof_read: null");
  VQ_CHECK_HIP(hipDeviceSynchronize());
  std::lock_guard<std::mutex> lk(g_prof_mu);
  double tot = 0;
  int n = 0;
  for (auto& p : g_prof_pairs) {
    if (p.tag != tag) continue;
    float ms = 0;
    if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { tot += ms; ++n; }
    else (void)hipGetLastError();            // (a pair that was never recorded: not this profile's, and not the next launch's error either)
  }
  *total_ms = tot;
  *launches = n;
  return 0;
}

}  // extern "C"

// ---- hipGraph capture / replay ------------------------------------------------------------------
extern "C" int vqvae_graph_capture_begin(vqvae_stream_t s) {
  VQ_CHECK_HIP(hipStreamBeginCapture((hipStream_t)s, hipStreamCaptureModeThreadLocal));
  return 0;
}
// relaxed capture: the capturing thread may still call hipMalloc / hipEventCreate (the Python-side pool allocator and
// event pool may have to grow while a whole training step is being recorded)
extern "C" int vqvae_graph_capture_begin_relaxed(vqvae_stream_t s) {
  VQ_CHECK_HIP(hipStreamBeginCapture((hipStream_t)s, hipStreamCaptureModeRelaxed));
  return 0;
}
extern "C" int vqvae_graph_capture_end(vqvae_stream_t s, void** graph_exec) {
  VQ_REQUIRE(graph_exec, "graph_capture_end: null output");
  hipGraph_t g = nullptr;
  VQ_CHECK_HIP(hipStreamEndCapture((hipStream_t)s, &g));
  hipGraphExec_t ex = nullptr;
  hipError_t e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  VQ_CHECK_HIP(e);
  *graph_exec = (void*)ex;
  return 0;
}
extern "C" int vqvae_graph_launch(void* graph_exec, vqvae_stream_t s) {
  VQ_REQUIRE(graph_exec, "graph_launch: null graph");
  VQ_CHECK_HIP(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)s));
  return 0;
}
extern "C" int vqvae_graph_destroy(void* graph_exec) {
  if (graph_exec) VQ_CHECK_HIP(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
  return 0;
}
