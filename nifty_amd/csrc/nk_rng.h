// nk_rng.h -- numpy's `Generator(PCG64).normal` stream, reproduced draw for draw on the device.
//
// The reference draws every random field on the host with `np.random.default_rng(SeedSequence).normal(mean, std, shape)`
// (nifty/cl/random.py:219-237, called from field.py:128-156 / multi_field.py:109-153 / kl_energies.py:91-159); at 1024^3
// that is 10-15 s per field, more than a whole MGVI iteration takes on the GPU.  numpy (third party; reference pin
// numpy>=1.23, this image 2.2.6) implements it as
//   * bit generator PCG64 = PCG XSL-RR 128/64 (O'Neill 2014): state <- state * M + inc (mod 2^128), output
//     rotr64(hi ^ lo, hi >> 58) of the NEW state; M = 0x2360ED051FC65DA44385DF649FCCF645
//     (numpy/random/src/pcg64/pcg64.h: pcg_setseq_128_step_r, pcg_output_xsl_rr_128_64)
//   * next_double = (next_uint64 >> 11) * 2^-53
//   * standard normal: 256-strip ziggurat on ONE raw 64-bit draw (bits 0-7 strip, bit 8 sign, bits 9-60 magnitude), wedge
//     test with one more double, tail (strip 0) by Marsaglia's exponential rejection with two doubles per try
//     (numpy/random/src/distributions/distributions.c: random_standard_normal); normal = mean + std * standard normal.
// A normal consumes 1 raw draw 97.9 % of the time and 2+ otherwise, so the position of the i-th normal in the raw stream
// depends on all earlier rejections.  The device algorithm (nk_rng.hip) makes that parallel without changing the stream:
//   A   the raw stream is cut into CHUNKS of NK_RNG_CHUNK draws (PCG64 jumps ahead in O(log) steps); every chunk walks
//       the ziggurat from its first draw: number of normals started in the chunk, overrun into the next chunk, bit mask of
//       the positions where a normal starts
//   A2  the true entry of a chunk is the overrun of its predecessor; whenever it is not 0 the chunk walks from there
//       until it lands on a position of the mask (the two chains have merged: every later draw is shared), which corrects
//       the count -- and shows that the overrun of a chunk does NOT depend on its entry, so no serial pass over chunks is
//       needed.  (No merge inside one chunk has probability < 1e-60; it raises NK_RNG_ERR_NOMERGE, never a wrong stream.)
//   S   exclusive prefix sum of the counts = output offset of every chunk
//   B   every chunk walks again from its true entry and writes its normals; the chunk that writes normal n-1 reports
//       the number of raw draws consumed, so that the host generator can be advanced (`bit_generator.advance`) and
//       stays in lockstep with the reference for the draws that follow.
// The functions below are the per-chunk bodies; the test-only host emulation (tests/emu) runs the same ones.
#pragma once
#include <math.h>
#include <stdint.h>

#include "nk_core.h"
#include "nk_ziggurat_tables.h"

#define NK_RNG_CHUNK 64            // raw draws per chunk (= bits of the start mask)
#define NK_RNG_ERR_NOMERGE 1u      // status bits (device word)
#define NK_RNG_ERR_SHORT 2u        // the chunks did not hold n normals (the caller retries with more)
#define NK_ZIG_R 3.6541528853610087963519472518
#define NK_ZIG_INV_R 0.27366123732975827203338247596

struct NkU128 {
  uint64_t hi, lo;
};
NK_HD uint64_t nk_mulhi64(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul64hi(a, b);
#else
  return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}
NK_HD NkU128 nk_mul128(NkU128 a, NkU128 b) {
  NkU128 r;
  r.lo = a.lo * b.lo;
  r.hi = nk_mulhi64(a.lo, b.lo) + a.hi * b.lo + a.lo * b.hi;
  return r;
}
NK_HD NkU128 nk_add128(NkU128 a, NkU128 b) {
  NkU128 r;
  r.lo = a.lo + b.lo;
  r.hi = a.hi + b.hi + (r.lo < a.lo ? 1u : 0u);
  return r;
}

// jump table: (m[i], p[i]) advance the generator by 2^i steps, state <- m[i] * state + p[i]
struct NkPcgJump {
  NkU128 m[64], p[64];
};
NK_HD void nk_pcg_jump_table(NkU128 inc, NkPcgJump& t) {
  NkU128 m{0x2360ED051FC65DA4ull, 0x4385DF649FCCF645ull}, p = inc;
  for (int i = 0; i < 64; ++i) {
    t.m[i] = m;
    t.p[i] = p;
    p = nk_mul128(nk_add128(m, NkU128{0, 1}), p);
    m = nk_mul128(m, m);
  }
}
NK_HD NkU128 nk_pcg_advance(NkU128 s, uint64_t delta, const NkPcgJump& t) {
  NkU128 am{0, 1}, ap{0, 0};
  for (int i = 0; delta; ++i, delta >>= 1)
    if (delta & 1) {
      am = nk_mul128(am, t.m[i]);
      ap = nk_add128(nk_mul128(ap, t.m[i]), t.p[i]);
    }
  return nk_add128(nk_mul128(am, s), ap);
}

// the raw stream from a given position on; pos counts the draws taken
struct NkRaw {
  NkU128 s, inc;
  int pos;
  NK_HD uint64_t next() {
    s = nk_add128(nk_mul128(s, NkU128{0x2360ED051FC65DA4ull, 0x4385DF649FCCF645ull}), inc);
    ++pos;
    const uint64_t x = s.hi ^ s.lo;
    const unsigned rot = (unsigned)(s.hi >> 58);
    return (x >> rot) | (x << ((64u - rot) & 63u));
  }
  NK_HD double next_double() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
};

struct NkZig {
  const uint64_t* ki;
  const double *wi, *fi;
};

// products and sums below must round separately like numpy's (built without FMA contraction): the accept / reject
// decisions and mean + std * x have to come out the same.  hipcc contracts by default (and its __dmul_rn / __dadd_rn are
// plain operators that fuse after inlining), so contraction is switched off per function
#if defined(__clang__)
#define NK_NO_CONTRACT _Pragma("clang fp contract(off)")
#else
#define NK_NO_CONTRACT
#endif
#define NK_DMUL(a, b) ((a) * (b))
#define NK_DADD(a, b) ((a) + (b))

// one standard normal starting at the current position of the raw stream (random_standard_normal)
NK_HD double nk_zig_normal(NkRaw& g, const NkZig& z) {
  NK_NO_CONTRACT
  for (;;) {
    uint64_t r = g.next();
    const int idx = (int)(r & 0xff);
    r >>= 8;
    const int sign = (int)(r & 1);
    const uint64_t rabs = (r >> 1) & 0x000fffffffffffffull;
    double x = NK_DMUL((double)rabs, z.wi[idx]);
    if (sign) x = -x;
    if (rabs < z.ki[idx]) return x;
    if (idx == 0) {
      for (;;) {
        const double xx = NK_DMUL(-NK_ZIG_INV_R, log1p(-g.next_double()));
        const double yy = -log1p(-g.next_double());
        if (NK_DADD(yy, yy) > NK_DMUL(xx, xx)) return ((rabs >> 8) & 1) ? -NK_DADD(NK_ZIG_R, xx) : NK_DADD(NK_ZIG_R, xx);
      }
    } else {
      const double lhs = NK_DADD(NK_DMUL(z.fi[idx - 1] - z.fi[idx], g.next_double()), z.fi[idx]);
      if (lhs < exp(NK_DMUL(NK_DMUL(-0.5, x), x))) return x;
    }
  }
}

struct NkRngArgs {
  NkU128 state, inc;   // generator state BEFORE the first draw
  int64_t n;           // normals wanted
  int64_t nchunks;
  double mean, std;
};

NK_HD NkRaw nk_rng_chunk_start(const NkRngArgs& a, const NkPcgJump& jt, int64_t k, int entry) {
  NkRaw g;
  g.inc = a.inc;
  g.s = nk_pcg_advance(a.state, (uint64_t)k * NK_RNG_CHUNK + (uint64_t)entry, jt);
  g.pos = entry;
  return g;
}

// pass A: chain from the first draw of chunk k -> c0 normals started, overrun ov, start mask m
NK_HD void nk_rng_pass_a(const NkRngArgs& a, const NkPcgJump& jt, const NkZig& z, int64_t k, int& c0, int& ov, uint64_t& m) {
  NkRaw g = nk_rng_chunk_start(a, jt, k, 0);
  m = 0;
  c0 = 0;
  while (g.pos < NK_RNG_CHUNK) {
    m |= 1ull << g.pos;
    (void)nk_zig_normal(g, z);
    ++c0;
  }
  ov = g.pos - NK_RNG_CHUNK;
  if (ov > 255) ov = 255;
}

// pass A2: the count of chunk k for its true entry e (= overrun of chunk k - 1); sets status bits on failure
NK_HD int nk_rng_pass_a2(const NkRngArgs& a, const NkPcgJump& jt, const NkZig& z, int64_t k, int e, int c0, uint64_t m,
                         unsigned* err) {
  if (e == 0) return c0;
  if (e >= NK_RNG_CHUNK) {
    *err |= NK_RNG_ERR_NOMERGE;
    return 0;
  }
  NkRaw g = nk_rng_chunk_start(a, jt, k, e);
  int c = 0;
  while (g.pos < NK_RNG_CHUNK && !((m >> g.pos) & 1)) {
    (void)nk_zig_normal(g, z);
    ++c;
  }
  if (g.pos >= NK_RNG_CHUNK) {  // the chains never met inside this chunk: its overrun is not the recorded one
    *err |= NK_RNG_ERR_NOMERGE;
    return 0;
  }
  // merged at g.pos: from there on the chunk's own chain is followed
  int before = 0;
  for (int b = 0; b < g.pos; ++b) before += (int)((m >> b) & 1);
  return c + c0 - before;
}

// pass B: write the normals of chunk k (true entry, output offset off); the chunk holding normal n-1 reports the raw
// draws consumed up to and including it
template <typename T>
NK_HD void nk_rng_pass_b(const NkRngArgs& a, const NkPcgJump& jt, const NkZig& z, int64_t k, int entry, int64_t off, T* out,
                         uint64_t* consumed) {
  NK_NO_CONTRACT
  if (off >= a.n || entry >= NK_RNG_CHUNK) return;
  NkRaw g = nk_rng_chunk_start(a, jt, k, entry);
  while (g.pos < NK_RNG_CHUNK && off < a.n) {
    const double x = nk_zig_normal(g, z);
    out[off] = (T)NK_DADD(a.mean, NK_DMUL(a.std, x));
    if (off == a.n - 1) *consumed = (uint64_t)k * NK_RNG_CHUNK + (uint64_t)g.pos;
    ++off;
  }
}

static inline int64_t nk_rng_chunks_for(int64_t n, int attempt) {
  // a normal takes 1.0220 raw draws on average (2.1 % need a second one); margin 0.5 % (+ 32 chunks for small n),
  // doubled on every retry
  const double need = (double)n * (1.0225 + 0.005 * (double)(1 << attempt)) / NK_RNG_CHUNK;
  return (int64_t)need + 32 * (int64_t)(1 + attempt);
}

// ---- draws that consume a FIXED number of raw values per output: plain jump-ahead parallelism ------------------------
// numpy's `Generator.uniform(low, high, n)` (random_uniform: low + (high - low) * next_double, one raw draw each) and the
// power-of-two bounded integers behind the reference's `Random.pm1` (random.py:239-247: `integers(0, 2, n)` for real,
// `integers(0, 4, n)` for complex fields).  For a range 2^k numpy's buffered 32-bit Lemire path never rejects: every raw
// 64-bit draw yields TWO outputs, the top k bits of its low 32-bit half first, then those of its high half
// (numpy/random/src/distributions/distributions.c: buffered_bounded_lemire_uint32; checked against numpy in
// tests/test_rng.py).  A thread takes NK_RNG_FIX raw draws: one O(log) jump, then the plain recurrence.
#define NK_RNG_FIX 32
// MODE 0: uniform -> T;  1: +-1 -> T;  2: one of 1, i, -1, -i -> C2<T> (interleaved re, im)
template <typename T, int MODE>
NK_HD void nk_rng_fixed_body(NkU128 state, NkU128 inc, const NkPcgJump& jt, int64_t n, double low, double scale, int64_t k,
                             T* out) {
  NK_NO_CONTRACT
  const int64_t per = MODE == 0 ? NK_RNG_FIX : 2 * NK_RNG_FIX;  // outputs of this thread
  const int64_t e0 = k * per;
  if (e0 >= n) return;
  NkRaw g;
  g.s = nk_pcg_advance(state, (uint64_t)k * NK_RNG_FIX, jt);
  g.inc = inc;
  g.pos = 0;
  for (int i = 0; i < NK_RNG_FIX; ++i) {
    const uint64_t raw = g.next();
    if (MODE == 0) {
      const int64_t e = e0 + i;
      if (e >= n) return;
      const double u = (double)(raw >> 11) * (1.0 / 9007199254740992.0);
      out[e] = (T)NK_DADD(low, NK_DMUL(scale, u));
    } else {
      for (int h = 0; h < 2; ++h) {
        const int64_t e = e0 + 2 * i + h;
        if (e >= n) return;
        const uint32_t half = h ? (uint32_t)(raw >> 32) : (uint32_t)raw;
        if (MODE == 1) {
          out[e] = (half >> 31) ? (T)1 : (T)-1;
        } else {
          const uint32_t code = half >> 30;  // 0: 1, 1: i, 2: -1, 3: -i
          out[2 * e] = code == 0 ? (T)1 : code == 2 ? (T)-1 : (T)0;
          out[2 * e + 1] = code == 1 ? (T)1 : code == 3 ? (T)-1 : (T)0;
        }
      }
    }
  }
}

// ---- bounded integers: numpy's Generator.integers(low, high + 1, size) (reference random.py:252-256, Random.uniform of
// integer fields; numpy/random/src/distributions/distributions.c: random_bounded_uint64_fill, use_masked = false) ------------
// rng = high - low:  0 < rng < 2^32 - 1  -> Lemire's method on 32-bit WORDS, rng >= 2^32 (and < 2^64 - 1) on 64-bit draws:
//     m = word * (rng + 1);  leftover = low half of m;  accept iff leftover >= threshold = (MAX - rng) % (rng + 1);  value = high half
// (numpy tests leftover < rng + 1 first -- threshold < rng + 1, so "leftover >= threshold" is the same decision).  A rejected
// word is skipped, so output i is the i-th ACCEPTED word: a thread walks NK_RNG_FIX raw draws (64 words in 32-bit mode: low
// half first, then the high half, PCG64's next_uint32 buffering), pass 1 counts its accepted words, pass 2 writes them behind
// the exclusive sum of the counts; the thread that writes output n - 1 reports how many words the draw consumed.
// rng + 1 a power of two never rejects.  (rng = 2^32 - 1 and 2^64 - 1 take the raw words themselves: threshold 0.)
struct NkIntArgs {
  NkU128 state, inc;
  int64_t n;        // outputs wanted
  uint64_t rng;     // high - low
  int64_t low;
  int wide;         // 0: 32-bit words, 1: 64-bit draws
  uint64_t threshold;
};
NK_HD bool nk_int_accept(const NkIntArgs& a, uint64_t word, int64_t& value) {
  if (!a.wide) {
    const uint32_t excl = (uint32_t)a.rng + 1u;
    if (excl == 0u) {  // rng = 2^32 - 1
      value = a.low + (int64_t)word;
      return true;
    }
    const uint64_t m = word * (uint64_t)excl;
    value = a.low + (int64_t)(m >> 32);
    return (uint32_t)m >= (uint32_t)a.threshold;
  }
  const uint64_t excl = a.rng + 1ull;
  if (excl == 0ull) {
    value = (int64_t)((uint64_t)a.low + word);
    return true;
  }
  const uint64_t lo = word * excl, hi = nk_mulhi64(word, excl);
  value = (int64_t)((uint64_t)a.low + hi);
  return lo >= a.threshold;
}
// words of thread k: accepted count (out == nullptr) or the accepted values written from output index `off` on
template <typename F>
NK_HD void nk_int_thread_words(const NkIntArgs& a, const NkPcgJump& jt, int64_t k, F f) {
  NkRaw g;
  g.s = nk_pcg_advance(a.state, (uint64_t)k * NK_RNG_FIX, jt);
  g.inc = a.inc;
  g.pos = 0;
  const int per = a.wide ? NK_RNG_FIX : 2 * NK_RNG_FIX;
  uint64_t raw = 0;
  for (int i = 0; i < per; ++i) {
    uint64_t word;
    if (a.wide) {
      word = g.next();
    } else {
      if ((i & 1) == 0) raw = g.next();
      word = (i & 1) ? (raw >> 32) : (raw & 0xffffffffull);
    }
    int64_t v;
    const bool ok = nk_int_accept(a, word, v);  // (before the call: the order of argument evaluation is unspecified)
    if (!f(k * per + i, ok, v)) return;
  }
}
NK_HD int nk_int_count(const NkIntArgs& a, const NkPcgJump& jt, int64_t k) {
  int c = 0;
  nk_int_thread_words(a, jt, k, [&](int64_t, bool ok, int64_t) {
    c += ok ? 1 : 0;
    return true;
  });
  return c;
}
NK_HD void nk_int_write(const NkIntArgs& a, const NkPcgJump& jt, int64_t k, int64_t off, int64_t* out, uint64_t* words_used) {
  nk_int_thread_words(a, jt, k, [&](int64_t word_index, bool ok, int64_t v) {
    if (!ok) return true;
    if (off >= a.n) return false;
    out[off] = v;
    if (off == a.n - 1) *words_used = (uint64_t)word_index + 1;
    ++off;
    return true;
  });
}
