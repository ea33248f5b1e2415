// TEST-ONLY host emulation of the device random-number path (nifty_amd/csrc/nk_rng.h): the same per-chunk bodies the
// kernels of nk_rng.hip run, executed chunk after chunk on the host, plus the plain serial restatement of numpy's
// algorithm.  tests/test_rng.py compares both with numpy itself, bit for bit.  Never linked into the product library.
#define NK_HOST_EMU 1
#include <vector>

#include "../../nifty_amd/csrc/nk_rng.h"

static NkZig host_tables() {
  return NkZig{NK_ZIG_KI, reinterpret_cast<const double*>(NK_ZIG_WI_BITS), reinterpret_cast<const double*>(NK_ZIG_FI_BITS)};
}

extern "C" int emu_pcg64_normal_serial(const uint64_t* state, const uint64_t* inc, int64_t n, double mean, double std,
                                       double* out, uint64_t* consumed) {
  NkRaw g;
  g.s = NkU128{state[0], state[1]};
  g.inc = NkU128{inc[0], inc[1]};
  g.pos = 0;
  const NkZig z = host_tables();
  uint64_t used = 0;
  for (int64_t i = 0; i < n; ++i) {
    g.pos = 0;
    out[i] = mean + std * nk_zig_normal(g, z);
    used += (uint64_t)g.pos;
  }
  *consumed = used;
  return 0;
}

extern "C" int emu_pcg64_advance(const uint64_t* state, const uint64_t* inc, uint64_t delta, uint64_t* out_state) {
  NkPcgJump jt;
  nk_pcg_jump_table(NkU128{inc[0], inc[1]}, jt);
  const NkU128 s = nk_pcg_advance(NkU128{state[0], state[1]}, delta, jt);
  out_state[0] = s.hi;
  out_state[1] = s.lo;
  return 0;
}

// dtype 0: float32 output, 1: float64.  Returns the status bits.
extern "C" unsigned emu_pcg64_normal_chunked(const uint64_t* state, const uint64_t* inc, int64_t n, double mean, double std,
                                             int dtype, void* out, uint64_t* consumed, int64_t nchunks) {
  NkRngArgs a;
  a.state = NkU128{state[0], state[1]};
  a.inc = NkU128{inc[0], inc[1]};
  a.n = n;
  a.nchunks = nchunks;
  a.mean = mean;
  a.std = std;
  NkPcgJump jt;
  nk_pcg_jump_table(a.inc, jt);
  const NkZig z = host_tables();
  std::vector<uint8_t> over(nchunks), cnt(nchunks);
  unsigned err = 0;
  int prev_over = 0;
  for (int64_t k = 0; k < nchunks; ++k) {
    int c0, ov;
    uint64_t m;
    nk_rng_pass_a(a, jt, z, k, c0, ov, m);
    cnt[k] = (uint8_t)nk_rng_pass_a2(a, jt, z, k, prev_over, c0, m, &err);
    over[k] = (uint8_t)ov;
    prev_over = ov;
  }
  int64_t off = 0;
  *consumed = 0;
  for (int64_t k = 0; k < nchunks; ++k) {
    const int entry = k > 0 ? over[k - 1] : 0;
    if (dtype == 0)
      nk_rng_pass_b<float>(a, jt, z, k, entry, off, (float*)out, consumed);
    else
      nk_rng_pass_b<double>(a, jt, z, k, entry, off, (double*)out, consumed);
    off += cnt[k];
  }
  if (off < n) err |= NK_RNG_ERR_SHORT;
  return err;
}

extern "C" int64_t emu_rng_chunks_for(int64_t n, int attempt) { return nk_rng_chunks_for(n, attempt); }

// fixed-rate draws (uniform / pm1): the per-thread body of k_rng_fixed, thread after thread.  mode as nk_rng_fixed_body.
extern "C" int emu_pcg64_fixed(const uint64_t* state, const uint64_t* inc, int64_t n, double low, double high, int mode,
                               int dtype, void* out) {
  const NkU128 s{state[0], state[1]}, c{inc[0], inc[1]};
  NkPcgJump jt;
  nk_pcg_jump_table(c, jt);
  const int64_t per = mode == 0 ? NK_RNG_FIX : 2 * NK_RNG_FIX;
  const int64_t threads = (n + per - 1) / per;
  for (int64_t k = 0; k < threads; ++k) {
    if (dtype == 0) {
      if (mode == 0) nk_rng_fixed_body<float, 0>(s, c, jt, n, low, high - low, k, (float*)out);
      else if (mode == 1) nk_rng_fixed_body<float, 1>(s, c, jt, n, 0, 0, k, (float*)out);
      else nk_rng_fixed_body<float, 2>(s, c, jt, n, 0, 0, k, (float*)out);
    } else {
      if (mode == 0) nk_rng_fixed_body<double, 0>(s, c, jt, n, low, high - low, k, (double*)out);
      else if (mode == 1) nk_rng_fixed_body<double, 1>(s, c, jt, n, 0, 0, k, (double*)out);
      else nk_rng_fixed_body<double, 2>(s, c, jt, n, 0, 0, k, (double*)out);
    }
  }
  return 0;
}

// bounded integers (nk_pcg64_integers): pass 1 and pass 2 thread after thread.  Returns the status bits; *words = words consumed
extern "C" unsigned emu_pcg64_integers(const uint64_t* state, const uint64_t* inc, int64_t n, int64_t low, uint64_t rng, int64_t nthreads,
                                       int64_t* out, uint64_t* words) {
  NkIntArgs a;
  a.state = NkU128{state[0], state[1]};
  a.inc = NkU128{inc[0], inc[1]};
  a.n = n;
  a.rng = rng;
  a.low = low;
  a.wide = rng > 0xFFFFFFFFull ? 1 : 0;
  if (a.wide) a.threshold = rng == ~0ull ? 0 : (~0ull - rng) % (rng + 1ull);
  else a.threshold = rng == 0xFFFFFFFFull ? 0 : (0xFFFFFFFFull - rng) % (rng + 1ull);
  NkPcgJump jt;
  nk_pcg_jump_table(a.inc, jt);
  std::vector<int> cnt(nthreads);
  for (int64_t k = 0; k < nthreads; ++k) cnt[k] = nk_int_count(a, jt, k);
  int64_t off = 0;
  *words = 0;
  for (int64_t k = 0; k < nthreads; ++k) {
    if (off < n) nk_int_write(a, jt, k, off, out, words);
    off += cnt[k];
  }
  return off < n ? NK_RNG_ERR_SHORT : 0u;
}
