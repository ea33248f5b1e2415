// col_tile_bench.hip -- what a pass over the FIRST axis of a 4096 x 4096 fp64 grid can reach on MI355X, arithmetic left out:
// every workgroup loads a tile of R rows x SEG bytes (rows r_step apart) into registers and stores it again, in place or into
// a second array.  Variants: the tile of today's single-kernel pass (4096 rows x 64 B, one workgroup of 1024 threads per CU),
// the same with two workgroups per CU, 128-byte segments, and the two tiles of a two-level (64 x 64) schedule.
// build: hipcc --offload-arch=gfx950 -O3 -o col_tile_bench col_tile_bench.hip ; run: ./col_tile_bench [row pad bytes]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef double __attribute__((ext_vector_type(2))) d2;  // one fp64 complex = 16 B

// R rows, S16 16-byte elements per row segment, THREADS threads: E = R * S16 / THREADS elements per thread
template <int R, int S16, int THREADS, int XMAP>
__global__ void __launch_bounds__(THREADS) k_tile(const d2* __restrict__ a, d2* __restrict__ b, size_t row_elems, int col_tiles,
                                                  int rg_count, size_t rg_step, size_t r_step, unsigned nblocks) {
  extern __shared__ char lds[];
  constexpr int E = R * S16 / THREADS, PR = THREADS / S16;
  unsigned blk = blockIdx.x;
  if (XMAP) blk = (blk % 8) * (nblocks / 8) + blk / 8;
  const int ct = blk % col_tiles, rg = blk / col_tiles;
  const int t = threadIdx.x % S16, pp = threadIdx.x / S16;
  const size_t base = (size_t)rg * rg_step * row_elems + (size_t)ct * S16 + t;
  d2 v[E];
#pragma unroll
  for (int e = 0; e < E; ++e) v[e] = a[base + (size_t)(pp + e * PR) * r_step * row_elems];
  if (lds[0] == 77 && threadIdx.x == 5000) v[0].x += 1.0;  // keep the dynamic LDS allocation alive
#pragma unroll
  for (int e = 0; e < E; ++e) __builtin_nontemporal_store(v[e], b + base + (size_t)(pp + e * PR) * r_step * row_elems);
}

template <int R, int S16, int THREADS, int XMAP>
static void run(const char* name, const d2* a, d2* b, size_t row_elems, int rows, int cols16, int rg_count, size_t rg_step, size_t r_step,
                size_t lds_bytes) {
  const int col_tiles = cols16 / S16;
  const unsigned nblocks = (unsigned)col_tiles * rg_count;
  CK(hipFuncSetAttribute((const void*)k_tile<R, S16, THREADS, XMAP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k_tile<R, S16, THREADS, XMAP>), dim3(nblocks), dim3(THREADS), lds_bytes, 0, a, b, row_elems, col_tiles, rg_count, rg_step, r_step, nblocks);
  CK(hipEventRecord(e0));
  const int reps = 20;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k_tile<R, S16, THREADS, XMAP>), dim3(nblocks), dim3(THREADS), lds_bytes, 0, a, b, row_elems, col_tiles, rg_count, rg_step, r_step, nblocks);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = 2.0 * rows * (double)cols16 * 16;
  printf("%-64s %7.1f us  %6.2f TB/s  (%u workgroups x %d threads, %zu KiB LDS)\n", name, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e12, nblocks, THREADS, lds_bytes / 1024);
}

int main(int argc, char** argv) {
  const size_t pad_bytes = argc > 1 ? atol(argv[1]) : 0;
  const int rows = 4096, cols16 = 2048;  // 4096 x 4096 real fp64 = 4096 rows of 2048 complex
  const size_t row_elems = cols16 + pad_bytes / 16;
  d2 *a, *b;
  CK(hipMalloc(&a, rows * row_elems * 16)); CK(hipMalloc(&b, rows * row_elems * 16));
  CK(hipMemset(a, 0, rows * row_elems * 16)); CK(hipMemset(b, 0, rows * row_elems * 16));
  printf("4096 x 4096 fp64 (128 MiB), row stride %zu B; 1R + 1W\n", row_elems * 16);
  for (int inplace = 0; inplace < 2; ++inplace) {
    d2* out = inplace ? a : b;
    printf("-- %s\n", inplace ? "in place" : "a -> b");
    run<4096, 4, 1024, 1>("4096 rows x 64 B, 1024 thr, 1 wg/CU (today)", a, out, row_elems, rows, cols16, 1, 0, 1, 128 * 1024);
    run<4096, 4, 1024, 0>("4096 rows x 64 B, 1024 thr, 1 wg/CU, natural block order", a, out, row_elems, rows, cols16, 1, 0, 1, 128 * 1024);
    run<4096, 4, 1024, 1>("4096 rows x 64 B, 1024 thr, 2 wg/CU", a, out, row_elems, rows, cols16, 1, 0, 1, 64 * 1024);
    run<4096, 8, 1024, 1>("4096 rows x 128 B, 1024 thr x 32 el, 1 wg/CU", a, out, row_elems, rows, cols16, 1, 0, 1, 128 * 1024);
    run<4096, 2, 512, 1>("4096 rows x 32 B, 512 thr, 2 wg/CU", a, out, row_elems, rows, cols16, 1, 0, 1, 64 * 1024);
    run<64, 8, 64, 1>("two-level A: 64 rows (64 apart) x 128 B, 64 thr x 8 el", a, out, row_elems, rows, cols16, 64, 1, 64, 0);
    run<64, 8, 64, 1>("two-level B: 64 consecutive rows x 128 B, 64 thr x 8 el", a, out, row_elems, rows, cols16, 64, 64, 1, 0);
    run<64, 16, 128, 1>("two-level A: 64 rows (64 apart) x 256 B, 128 thr x 8 el", a, out, row_elems, rows, cols16, 64, 1, 64, 0);
    run<64, 16, 128, 1>("two-level B: 64 consecutive rows x 256 B, 128 thr x 8 el", a, out, row_elems, rows, cols16, 64, 64, 1, 0);
    run<64, 8, 256, 1>("two-level A: 64 rows (64 apart) x 128 B, 256 thr x 2 el", a, out, row_elems, rows, cols16, 64, 1, 64, 0);
    run<64, 8, 256, 1>("two-level B: 64 consecutive rows x 128 B, 256 thr x 2 el", a, out, row_elems, rows, cols16, 64, 64, 1, 0);
    run<64, 8, 64, 1>("two-level A, 64 thr x 8 el, 32 KiB LDS each", a, out, row_elems, rows, cols16, 64, 1, 64, 32 * 1024);
    run<256, 8, 256, 1>("16 x 256 split, A: 256 rows (16 apart) x 128 B, 256 thr x 8 el", a, out, row_elems, rows, cols16, 16, 1, 16, 0);
    run<256, 8, 256, 1>("16 x 256 split, B': 256 consecutive rows x 128 B", a, out, row_elems, rows, cols16, 16, 256, 1, 0);
  }
  return 0;
}
