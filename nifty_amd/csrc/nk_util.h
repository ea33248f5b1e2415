// nk_util.h -- error plumbing of libniftyk (never throw across the C ABI)
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/niftyk.h"

int nk_set_error(int code, const char* msg);
int nk_set_hip_error(hipError_t e, const char* what);

static inline int nk_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return nk_set_hip_error(e, what);
  return NK_OK;
}

// Function attributes (dynamic LDS size) are per DEVICE: a launcher keeps one bit per device id in a static mask.
// Returns true the first time it is called for the current device (ids >= 64 are set every time).
static inline bool nk_first_on_device(unsigned long long& mask) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
  const unsigned long long bit = 1ULL << dev;
  if (mask & bit) return false;
  mask |= bit;
  return true;
}

// Scratch of the deterministic (ticket-ordered) reductions of nk_vec.hip: block partials + ticket, one set per
// (device, stream), allocated on first use and kept for the life of the process.  Launches on ONE stream are serialised
// and leave the ticket at zero; launches on different streams / from different host threads use different scratch, so
// the C ABI may be driven from several streams at once.
constexpr int NK_RED_MAX = 4;          // reductions per launch
constexpr int NK_RED_MAX_BLOCKS = 16384;  // workgroups per launch (NK_RED_UNITS units x up to 256 workgroups each)
constexpr int NK_RED_UNITS = 64;          // a long array is reduced as 64 equal contiguous UNITS (see NkRedLayout)
struct NkRedScratch {
  double* partial;       // [NK_MAX_BATCH][NK_RED_MAX][NK_RED_MAX_BLOCKS]: one set per member of a batched launch
  unsigned int* ticket;  // [NK_MAX_BATCH], zero between launches
};
constexpr size_t NK_RED_MEMBER_STRIDE = (size_t)NK_RED_MAX * NK_RED_MAX_BLOCKS;  // partials of one batch member
int nk_red_scratch(hipStream_t st, NkRedScratch* out);

// Publishing a workgroup's partial sum to the LAST workgroup of the launch (ticket-ordered deterministic reductions) without
// a device-scope fence.  The eight XCDs' L2s are not coherent with each other, so `__threadfence()` makes the workgroup
// write back its XCD's whole L2 -- with one fence per workgroup a launch then costs time in proportion to its number of
// workgroups (round 5: k_vjp_red2 15 us alone, 100 us for a batch of eight members; 23 us with this).  Instead the partial
// is stored by an atomic EXCHANGE at agent scope: a read-modify-write is performed at the device's coherence point (past
// the L2s), and once its old value has come back it HAS been performed.  Only then is the ticket taken (another agent-scope
// RMW at the same point), so whoever draws the last ticket finds every partial there; it reads them with agent-scope
// atomic loads, which bypass its own L2 as well.  No cache maintenance anywhere.
// This path leans on where agent-scope read-modify-writes and loads are performed on THIS chip, not on a happens-before edge
// of the HIP memory model.  -DNK_RED_FENCE=1 builds the textbook protocol instead -- plain stores, a device-scope
// fence before the ticket and after it -- as `make fence` -> build/libniftyk_fence.so; tests/test_kernels_gpu.py holds the two
// libraries against each other over many launches and grid sizes (same order of additions: same bits).
#ifndef NK_RED_FENCE
#define NK_RED_FENCE 0
#endif
#if defined(__HIPCC__)
__device__ __forceinline__ void nk_publish_partial(double* slot, double v) {
#if NK_RED_FENCE
  *slot = v;
#else
  const unsigned long long old = __hip_atomic_exchange(reinterpret_cast<unsigned long long*>(slot), (unsigned long long)__double_as_longlong(v),
                                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // the returned value is consumed: the exchange has completed before anything that follows in program order is issued
  asm volatile("" ::"v"(old) : "memory");
#endif
}
__device__ __forceinline__ bool nk_take_last_ticket(unsigned int* ticket, unsigned int count) {
#if NK_RED_FENCE
  __threadfence();
  const bool last = atomicAdd(ticket, 1u) == count - 1;
  __threadfence();
  return last;
#else
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // (waits for the outstanding memory operations of this wavefront)
  return __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == count - 1;
#endif
}
__device__ __forceinline__ double nk_read_partial(const double* slot) {
#if NK_RED_FENCE
  return *reinterpret_cast<const volatile double*>(slot);
#else
  return __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
// every thread of the LAST workgroup, after the barrier that tells it so and before its first nk_read_partial
__device__ __forceinline__ void nk_acquire_partials() {
#if NK_RED_FENCE
  __threadfence();
#endif
}
// the last workgroup leaves the ticket at zero for the next launch on the stream: an agent-scope store, like its increments
__device__ __forceinline__ void nk_reset_ticket(unsigned int* ticket) {
  __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#endif

// How a reduction launch groups its partial sums.  An array of n elements whose length qualifies (nk_red_unit_elems) is
// cut into NK_RED_UNITS contiguous units; every unit is reduced by its own sub-grid -- a function of the unit's length
// only -- and the unit sums are added in unit order.  A SHARD of such an array that consists of whole units (the CG
// vectors of a multi-rank KL minimisation) therefore yields, unit by unit, the very bits a single process computes for
// the full array: the ranks exchange the unit sums (every unit lives on exactly one rank, the others hold a zero, so any
// all-reduce is exact) and add them in the same order (nk_red_finish).  Rank-count-independent dot products
// (reference utilities.py:349-414 for the sums over samples; this is the counterpart for the sharded vectors).
struct NkRedLayout {
  int k_local;         // units in this launch (1: the whole array is one unit)
  int64_t unit_elems;  // elements per unit (k_local > 1)
  int k_global;        // units of the full array
  int seg_units;       // the shard is made of segments of seg_units units ...
  int seg_stride;      // ... that start every seg_stride units of the full array ...
  int seg_off;         // ... at unit seg_off of each stride
  double* units_out;   // [NRED][k_global] unit sums (zeros for units of other ranks) INSTEAD of the result; or nullptr
};

// Live profiling for bench.py (nk_profile_enable / nk_profile_collect): a scope brackets ONE kernel launch with HIP events
// on its launch stream.  key = kernel * 25 + pro * 5 + epi:
//   kernel 0 pass1d, 1 passA, 2 passB, 3 passC, 4 passD; sandwich: 5 contiguous first pass, 6 in-place middle-axis pass,
//   7 fused first-axis pass (pro / epi = prologue / epilogue class of the launch; pro 4 = the sandwich's first pass with
//   the AMP_JVP prologue AND the pending CG direction update, nk_fuse.cg_r: two more streams);
//   kernel 8 = nk_csr_rowsum (pro = lanes class 0..3 for 1 / 4 / 16 / 64 lanes per row, epi = 0 weighted, 1 unweighted).
constexpr int NK_PROF_KEYS = 250;
struct NkProfScope {
  hipStream_t st;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int key;
  int weight;  // members of a batched launch: it counts as `weight` launches of the key
  bool on;
  NkProfScope(hipStream_t s, int kernel, int pro, int epi, int weight = 1);
  ~NkProfScope();
};
