// Error reporting and version queries of the C ABI (include/dvsg_amd.h).
#include "common.h"

namespace dvsg {

char *error_buffer() {
  static thread_local char buf[512] = "";
  return buf;
}

int fail(int status, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(error_buffer(), 512, fmt, ap);
  va_end(ap);
  return status;
}

}  // namespace dvsg

extern "C" {
int dvsg_abi_version(void) { return DVSG_ABI_VERSION; }
const char *dvsg_last_error_string(void) { return dvsg::error_buffer(); }
const char *dvsg_target_arch(void) { return "gfx950"; }
}

// ---- per-kernel-class timing -------------------------------------------------------------
#include <atomic>
#include <mutex>
#include <vector>
namespace dvsg {
namespace {
// The library may be called from several host threads (the error buffer is thread_local): the armed
// class is an atomic, so the disarmed path costs one relaxed load, and the event lists are only
// touched under the mutex.
struct Prof {
  std::atomic<int> cls{-1};
  std::mutex mu;
  std::vector<hipEvent_t> start, stop;
  std::vector<hipEvent_t> pool;   // events made by dvsg_prof_begin (and handed back by dvsg_prof_end): a profiled launch
                                  // inside a timed region costs two hipEventRecord, no hipEventCreate / Destroy
  double flops = 0, bytes = 0;
} g_prof;
constexpr size_t kProfPoolEvents = 2048;   // 1024 launches per armed region without a creation
}  // namespace

ProfScope::ProfScope(int cls, hipStream_t s, double flops, double bytes) : idx_(-1), s_(s) {
  if (cls != g_prof.cls.load(std::memory_order_relaxed)) return;
  std::lock_guard<std::mutex> lock(g_prof.mu);
  if (cls != g_prof.cls.load(std::memory_order_relaxed)) return;  // disarmed meanwhile
  hipEvent_t a, b;
  if (g_prof.pool.size() >= 2) {
    a = g_prof.pool.back(); g_prof.pool.pop_back();
    b = g_prof.pool.back(); g_prof.pool.pop_back();
  } else {   // a region of more launches than the pool holds: create (and keep) more
    if (hipEventCreate(&a) != hipSuccess) return;
    if (hipEventCreate(&b) != hipSuccess) {
      (void)hipEventDestroy(a);
      return;
    }
  }
  stop_ = b;
  g_prof.start.push_back(a);
  g_prof.stop.push_back(b);
  g_prof.flops += flops;
  g_prof.bytes += bytes;
  idx_ = (int)g_prof.start.size() - 1;
  (void)hipEventRecord(a, s);
}
ProfScope::~ProfScope() {
  if (idx_ >= 0) (void)hipEventRecord(stop_, s_);
}
}  // namespace dvsg

extern "C" {
int dvsg_prof_begin(int kernel_class) {
  DVSG_REQUIRE(kernel_class >= 0 && kernel_class < dvsg::kNumCls, "dvsg_prof_begin: class %d outside [0,%d)",
               kernel_class, dvsg::kNumCls);
  std::lock_guard<std::mutex> lock(dvsg::g_prof.mu);
  DVSG_REQUIRE(dvsg::g_prof.cls.load() < 0, "dvsg_prof_begin: profiling already armed");
  while (dvsg::g_prof.pool.size() < dvsg::kProfPoolEvents) {   // once per process in practice: dvsg_prof_end hands them back
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) break;
    dvsg::g_prof.pool.push_back(e);
  }
  dvsg::g_prof.flops = dvsg::g_prof.bytes = 0;
  dvsg::g_prof.cls.store(kernel_class);
  return DVSG_OK;
}

int dvsg_prof_end(double *total_ms, int *launches, double *flops, double *bytes) {
  using dvsg::g_prof;
  std::lock_guard<std::mutex> lock(g_prof.mu);
  DVSG_REQUIRE(g_prof.cls.load() >= 0, "dvsg_prof_end: profiling not armed");
  g_prof.cls.store(-1);   // launches racing with this call stop recording
  double ms = 0;
  int rc = DVSG_OK;
  for (size_t i = 0; i < g_prof.start.size(); ++i) {
    float t = 0;
    if (hipEventSynchronize(g_prof.stop[i]) != hipSuccess ||
        hipEventElapsedTime(&t, g_prof.start[i], g_prof.stop[i]) != hipSuccess)
      rc = dvsg::fail(DVSG_ERR_HIP, "dvsg_prof_end: event query failed");
    ms += t;
    g_prof.pool.push_back(g_prof.start[i]);
    g_prof.pool.push_back(g_prof.stop[i]);
  }
  if (total_ms) *total_ms = ms;
  if (launches) *launches = (int)g_prof.start.size();
  if (flops) *flops = g_prof.flops;
  if (bytes) *bytes = g_prof.bytes;
  g_prof.start.clear();
  g_prof.stop.clear();
  return rc;
}
}

// ---- roctx ranges (DVSG_ROCTX=1) ------------------------------------------------------------------------
#include <dlfcn.h>

#include <cstdlib>
namespace dvsg {
namespace {
struct Roctx {
  int (*push)(const char *) = nullptr;
  int (*pop)() = nullptr;
  Roctx() {
    const char *e = std::getenv("DVSG_ROCTX");
    if (!e || e[0] == '0' || e[0] == 0) return;
    // rocprofv3 intercepts the rocprofiler-sdk flavour; the classic libroctx64 serves older tools
    for (const char *lib : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
      void *h = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
      if (!h) continue;
      push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
      pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
      if (push && pop) return;
      push = nullptr;
      pop = nullptr;
    }
    std::fprintf(stderr, "dvsg: DVSG_ROCTX is set but no roctx library could be loaded; ranges disabled\n");
  }
};
const Roctx &roctx() {
  static const Roctx r;   // thread-safe one-time initialisation
  return r;
}
}  // namespace

MarkerRange::MarkerRange(const char *name) : open_(false) {
  const Roctx &r = roctx();
  if (r.push) {
    r.push(name);
    open_ = true;
  }
}
MarkerRange::~MarkerRange() {
  if (open_) roctx().pop();
}
}  // namespace dvsg

extern "C" int dvsg_markers_enabled(void) { return dvsg::roctx().push != nullptr; }
