// nk_util.hip -- error state + version
#include "nk_util.h"

#include <cstdio>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

static thread_local char g_err[512] = "";

int nk_set_error(int code, const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg ? msg : "");
  return code;
}

int nk_set_hip_error(hipError_t e, const char* what) {
  snprintf(g_err, sizeof(g_err), "%s: %s", what ? what : "hip", hipGetErrorString(e));
  return e == hipErrorOutOfMemory ? NK_ERR_NOMEM : NK_ERR_RUNTIME;
}

extern "C" const char* nk_last_error(void) { return g_err; }
extern "C" int nk_version(void) { return 100; }

namespace {
std::mutex g_red_mutex;
std::map<std::pair<int, hipStream_t>, NkRedScratch> g_red_map;
thread_local int t_red_dev = -1;
thread_local hipStream_t t_red_stream = nullptr;
thread_local NkRedScratch t_red_last = {nullptr, nullptr};
}  // namespace

int nk_red_scratch(hipStream_t st, NkRedScratch* out) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return nk_set_hip_error(e, "hipGetDevice");
  if (t_red_last.partial && t_red_dev == dev && t_red_stream == st) {  // the common case: same stream as last time
    *out = t_red_last;
    return NK_OK;
  }
  std::lock_guard<std::mutex> lock(g_red_mutex);
  auto key = std::make_pair(dev, st);
  auto it = g_red_map.find(key);
  if (it == g_red_map.end()) {
    const size_t bytes = sizeof(double) * NK_RED_MEMBER_STRIDE * NK_MAX_BATCH + 256;
    char* p = nullptr;
    e = hipMalloc((void**)&p, bytes);
    if (e != hipSuccess) return nk_set_hip_error(e, "hipMalloc(reduction scratch)");
    // zeroed ON THE LAUNCH STREAM: a plain hipMemset of device memory may still be running when the first kernel of a
    // non-blocking stream starts, and would then wipe the ticket under it (seen once the scratch grew to 512 KiB)
    e = hipMemsetAsync(p, 0, bytes, st);
    if (e != hipSuccess) return nk_set_hip_error(e, "hipMemsetAsync(reduction scratch)");
    NkRedScratch s;
    s.partial = (double*)p;
    s.ticket = (unsigned int*)(p + sizeof(double) * NK_RED_MEMBER_STRIDE * NK_MAX_BATCH);
    it = g_red_map.emplace(key, s).first;
  }
  t_red_dev = dev, t_red_stream = st, t_red_last = it->second;
  *out = it->second;
  return NK_OK;
}

// ---- live profiling ------------------------------------------------------------------------------------------------
namespace {
struct ProfRec {
  hipEvent_t e0, e1;
  int key, weight;
};
bool g_prof_on = false;
std::vector<ProfRec> g_prof_recs;
std::vector<hipEvent_t> g_prof_pool;
std::mutex g_prof_mu;
constexpr size_t NK_PROF_MAX = 1 << 17;

hipEvent_t prof_event() {
  if (!g_prof_pool.empty()) {
    hipEvent_t e = g_prof_pool.back();
    g_prof_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}
}  // namespace

NkProfScope::NkProfScope(hipStream_t s, int kernel, int pro, int epi, int w)
    : st(s), key(kernel * 25 + pro * 5 + epi), weight(w), on(g_prof_on) {
  if (!on) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (g_prof_recs.size() >= NK_PROF_MAX) {
    on = false;
    return;
  }
  e0 = prof_event();
  e1 = prof_event();
  if (!e0 || !e1) {
    on = false;
    return;
  }
  (void)hipEventRecord(e0, st);
}

NkProfScope::~NkProfScope() {
  if (!on) return;
  (void)hipEventRecord(e1, st);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_recs.push_back(ProfRec{e0, e1, key, weight});
}

extern "C" int nk_profile_enable(int on) {
  g_prof_on = on != 0;
  return NK_OK;
}

// ms[NK_PROF_KEYS], count[NK_PROF_KEYS] indexed by kernel*25 + pro*5 + epi (see nk_util.h)
extern "C" int nk_profile_collect(double* ms, int64_t* count) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (int i = 0; i < NK_PROF_KEYS; ++i) ms[i] = 0.0, count[i] = 0;
  for (ProfRec& r : g_prof_recs) {
    float t = 0.f;
    hipError_t e = hipEventSynchronize(r.e1);
    if (e == hipSuccess) e = hipEventElapsedTime(&t, r.e0, r.e1);
    if (e == hipSuccess && r.key >= 0 && r.key < NK_PROF_KEYS) {
      ms[r.key] += t;
      count[r.key] += r.weight;
    }
    g_prof_pool.push_back(r.e0);
    g_prof_pool.push_back(r.e1);
  }
  g_prof_recs.clear();
  return NK_OK;
}
