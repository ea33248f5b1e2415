// copy_bench.hip -- what limits a 1R+1W stream on MI355X?  Variants of a device copy over 4 GiB.
// build: hipcc --offload-arch=gfx950 -O3 -o copy_bench copy_bench.hip ; run: ./copy_bench [log2 elements]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef float __attribute__((ext_vector_type(4))) f4;
typedef float __attribute__((ext_vector_type(2))) f2;

// V0: grid-stride, one float4 per thread per trip
__global__ void __launch_bounds__(256) k_gs(const f4* __restrict__ a, f4* __restrict__ b, size_t n4) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) b[i] = a[i];
}
// V1: grid-stride, U float4 per thread per trip issued before the stores; NT = non-temporal stores / loads
template <int U, int NTL, int NTS>
__global__ void __launch_bounds__(256) k_gsu(const f4* __restrict__ a, f4* __restrict__ b, size_t n4) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride * U) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (i + u * stride < n4) v[u] = NTL ? __builtin_nontemporal_load(a + i + u * stride) : a[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (i + u * stride < n4) {
        if (NTS) __builtin_nontemporal_store(v[u], b + i + u * stride); else b[i + u * stride] = v[u];
      }
  }
}
// V2: chunked -- every workgroup owns a contiguous chunk of `chunk4` float4 (many workgroups, one chunk each)
template <int U, int NTS>
__global__ void __launch_bounds__(256) k_chunk(const f4* __restrict__ a, f4* __restrict__ b, size_t n4, size_t chunk4) {
  const size_t lo = (size_t)blockIdx.x * chunk4;
  const size_t hi = lo + chunk4 < n4 ? lo + chunk4 : n4;
  for (size_t i = lo + threadIdx.x; i < hi; i += 256 * U) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + u * 256 < hi) v[u] = a[i + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + u * 256 < hi) { if (NTS) __builtin_nontemporal_store(v[u], b + i + u * 256); else b[i + u * 256] = v[u]; }
  }
}
// V3: read-only (sum) and write-only for reference
__global__ void __launch_bounds__(256) k_read(const f4* __restrict__ a, float* out, size_t n4) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  f4 s = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) s += a[i];
  if (s.x + s.y + s.z + s.w == 12345.678f) out[0] = 1.f;
}
__global__ void __launch_bounds__(256) k_write(f4* __restrict__ b, size_t n4) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const f4 v = {1, 2, 3, 4};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) __builtin_nontemporal_store(v, b + i);
}
// V4: FFT-pass-like: each workgroup loads a 128 KiB tile as rows of 128 B at a large stride, then stores it the same way
template <int NTS>
__global__ void __launch_bounds__(512) k_tile(const f2* __restrict__ a, f2* __restrict__ b, size_t rstride, int tiles_per_slab, int rows) {
  const size_t o = blockIdx.x / tiles_per_slab, c0 = (size_t)(blockIdx.x % tiles_per_slab) * 16;
  const int t = threadIdx.x % 16, pp = threadIdx.x / 16;  // 32 row threads x 16 columns
  const f2* base = a + o * rows * rstride + c0 + t;
  f2* ob = b + o * rows * rstride + c0 + t;
  f2 v[32];
#pragma unroll
  for (int r = 0; r < 32; ++r) v[r] = base[(size_t)(pp + 32 * r) * rstride];
#pragma unroll
  for (int r = 0; r < 32; ++r) {
    if (NTS) __builtin_nontemporal_store(v[r], ob + (size_t)(pp + 32 * r) * rstride); else ob[(size_t)(pp + 32 * r) * rstride] = v[r];
  }
}

// V5: general tile: ROWS rows of TW complex (8 B) columns, row stride rstride, 512 threads; XMAP: blocks of one XCD
// (blockIdx % 8) take consecutive tiles
template <int TW, int ROWS, int XMAP>
__global__ void __launch_bounds__(512) k_tile2(const f2* __restrict__ a, f2* __restrict__ b, size_t rstride, size_t slab_stride,
                                               int tiles_per_row, int rowblocks, unsigned nblocks) {
  unsigned blk = blockIdx.x;
  if (XMAP) blk = (blk % 8) * (nblocks / 8) + blk / 8;
  const int ct = blk % tiles_per_row;
  const int rb = (blk / tiles_per_row) % rowblocks;
  const size_t o = blk / tiles_per_row / rowblocks;
  constexpr int TPR = 512 / TW;           // row threads
  constexpr int PER = ROWS / TPR;         // rows per thread
  const int t = threadIdx.x % TW, pp = threadIdx.x / TW;
  const size_t off = o * slab_stride + (size_t)rb * ROWS * rstride + (size_t)ct * TW + t;
  f2 v[PER];
#pragma unroll
  for (int r = 0; r < PER; ++r) v[r] = a[off + (size_t)(pp + TPR * r) * rstride];
#pragma unroll
  for (int r = 0; r < PER; ++r) b[off + (size_t)(pp + TPR * r) * rstride] = v[r];
}

// V6: final-pass-like: a workgroup of 128 threads reads the line pair (a,b), (A-a, M-b) (4 KiB each, contiguous)
// and writes the two lines of the output; XMAP as above
template <int XMAP>
__global__ void __launch_bounds__(128) k_pair(const f4* __restrict__ a, f4* __restrict__ b, unsigned nblocks, size_t ss4) {
  unsigned blk = blockIdx.x;
  if (XMAP) blk = (blk % 8) * (nblocks / 8) + blk / 8;
  const unsigned bb = blk % 512, aa = blk / 512;           // a in [0,512], b in [0,1024): simplified pairing
  const unsigned am = (1024 - aa) % 1024, bm = (1024 - bb) % 1024;
  const size_t l0 = ((size_t)aa * ss4 + (size_t)bb * 256), l1 = ((size_t)am * ss4 + (size_t)bm * 256);  // float4 units
  f4 v[4];
  v[0] = a[l0 + threadIdx.x]; v[1] = a[l0 + 128 + threadIdx.x]; v[2] = a[l1 + threadIdx.x]; v[3] = a[l1 + 128 + threadIdx.x];
  b[l0 + threadIdx.x] = v[0]; b[l0 + 128 + threadIdx.x] = v[1]; b[l1 + threadIdx.x] = v[2]; b[l1 + 128 + threadIdx.x] = v[3];
}

// V7: tile copy with a fake compute phase (SPIN dependent FMAs per element + 4 barriers) between load and store --
// the shape of an FFT pass.  PERSIST: one workgroup per CU walks its tiles; PREF: the next tile is fetched into
// registers while the current one "computes".
template <int SPIN, int PERSIST, int PREF, int XMAP, int TW = 0, int LDSX = 0>
__global__ void __launch_bounds__(512) k_fft_like(const f2* __restrict__ a, f2* __restrict__ b, size_t rstride, size_t slab_stride,
                                                   int tiles_per_row, unsigned nblocks, const f2* __restrict__ table = nullptr) {
  extern __shared__ float lds[];
  const int t = threadIdx.x % 16, pp = threadIdx.x / 16;
  auto off_of = [&](unsigned blk) {
    const int ct = blk % tiles_per_row;
    const size_t o = blk / tiles_per_row;
    return o * slab_stride + (size_t)ct * 16 + t + (size_t)pp * rstride;
  };
  auto tile_of = [&](unsigned j, unsigned& blk) {  // j-th tile of this workgroup
    if (!PERSIST) { blk = XMAP ? (blockIdx.x % 8) * (nblocks / 8) + blockIdx.x / 8 : blockIdx.x; return j == 0; }
    if (XMAP) { const unsigned q = nblocks / 8, x = blockIdx.x % 8, i = blockIdx.x / 8 + j * (gridDim.x / 8); blk = x * q + i; return i < q; }
    blk = blockIdx.x + j * gridDim.x; return blk < nblocks;
  };
  f2 v[32], nx[32];
  unsigned blk, nblk_;
  if (!tile_of(0, blk)) return;
  size_t off = off_of(blk);
#pragma unroll
  for (int r = 0; r < 32; ++r) (PREF ? nx[r] : v[r]) = a[off + (size_t)(32 * r) * rstride];
  for (unsigned j = 0;; ++j) {
    if (PREF) {
#pragma unroll
      for (int r = 0; r < 32; ++r) v[r] = nx[r];
    }
    // "stage 0"
#pragma unroll
    for (int r = 0; r < 32; ++r)
      for (int k = 0; k < SPIN / 2; ++k) v[r].x = v[r].x * 1.0001f + v[(r + 1) & 31].y * 1e-9f;
    const bool more = tile_of(j + 1, nblk_);
    if (PREF && more) {
      const size_t no = off_of(nblk_);
#pragma unroll
      for (int r = 0; r < 32; ++r) nx[r] = a[no + (size_t)(32 * r) * rstride];
    }
    lds[threadIdx.x] = v[0].x; __syncthreads(); v[1].y += lds[(threadIdx.x + 17) & 511] * 1e-9f; __syncthreads();
    lds[threadIdx.x] = v[2].x; __syncthreads(); v[3].y += lds[(threadIdx.x + 33) & 511] * 1e-9f; __syncthreads();
    if (TW) {  // twiddle-like loads: per thread TW table entries, 4 distinct addresses per wave
#pragma unroll
      for (int r = 1; r <= TW; ++r) { const f2 w = table[((pp & 31) * r) & 1023]; v[r & 31].x = v[r & 31].x * w.x - v[r & 31].y * w.y; }
    }
    if (LDSX) {  // full LDS exchange like the FFT: 32 b32 writes + reads, twice
      float* pl = lds + 512;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int r = 0; r < 32; ++r) pl[(32 * pp + r) * 16 + t] = h ? v[r].y : v[r].x;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 32; ++r) { const float x = pl[(pp + 32 * r) * 16 + t]; if (h) v[r].y = x; else v[r].x = x; }
        __syncthreads();
      }
    }
#pragma unroll
    for (int r = 0; r < 32; ++r)
      for (int k = 0; k < SPIN / 2; ++k) v[r].y = v[r].y * 1.0001f + v[(r + 1) & 31].x * 1e-9f;
#pragma unroll
    for (int r = 0; r < 32; ++r) b[off + (size_t)(32 * r) * rstride] = v[r];
    if (!more) break;
    off = off_of(nblk_);
    if (!PREF) {
#pragma unroll
      for (int r = 0; r < 32; ++r) v[r] = a[off + (size_t)(32 * r) * rstride];
    }
  }
}

template <typename F>
static void timed(const char* tag, double bytes, F fn, int reps = 10) {
  fn();
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) fn();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  printf("%-44s %7.3f ms  %7.1f GB/s\n", tag, ms, bytes / ms / 1e6);
}

int main(int argc, char** argv) {
  const int lg = argc > 1 ? atoi(argv[1]) : 30;
  const size_t n = (size_t)1 << lg, n4 = n / 4;
  float *a, *b;
  CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4));
  CK(hipMemset(a, 1, n * 4)); CK(hipMemset(b, 0, n * 4));
  const double B2 = 2.0 * n * 4;
  for (int g : {2048, 4096, 8192, 16384}) {
    char tag[96]; snprintf(tag, 96, "grid-stride f4, %d blocks", g);
    timed(tag, B2, [&] { hipLaunchKernelGGL(k_gs, dim3(g), dim3(256), 0, 0, (const f4*)a, (f4*)b, n4); });
  }
  timed("gs U=4", B2, [&] { hipLaunchKernelGGL((k_gsu<4, 0, 0>), dim3(2048), dim3(256), 0, 0, (const f4*)a, (f4*)b, n4); });
  timed("gs U=4 NT store", B2, [&] { hipLaunchKernelGGL((k_gsu<4, 0, 1>), dim3(2048), dim3(256), 0, 0, (const f4*)a, (f4*)b, n4); });
  timed("gs U=4 NT load+store", B2, [&] { hipLaunchKernelGGL((k_gsu<4, 1, 1>), dim3(2048), dim3(256), 0, 0, (const f4*)a, (f4*)b, n4); });
  timed("gs U=8 NT store", B2, [&] { hipLaunchKernelGGL((k_gsu<8, 0, 1>), dim3(2048), dim3(256), 0, 0, (const f4*)a, (f4*)b, n4); });
  timed("gs U=1 NT store", B2, [&] { hipLaunchKernelGGL((k_gsu<1, 0, 1>), dim3(4096), dim3(256), 0, 0, (const f4*)a, (f4*)b, n4); });
  for (size_t kib : {64, 256, 1024, 4096}) {
    const size_t chunk4 = kib * 1024 / 16;
    const unsigned blocks = (unsigned)((n4 + chunk4 - 1) / chunk4);
    char tag[96]; snprintf(tag, 96, "chunked %zu KiB/WG U=4 (%u blocks)", kib, blocks);
    timed(tag, B2, [&] { hipLaunchKernelGGL((k_chunk<4, 0>), dim3(blocks), dim3(256), 0, 0, (const f4*)a, (f4*)b, n4, chunk4); });
    snprintf(tag, 96, "chunked %zu KiB/WG U=4 NT store", kib);
    timed(tag, B2, [&] { hipLaunchKernelGGL((k_chunk<4, 1>), dim3(blocks), dim3(256), 0, 0, (const f4*)a, (f4*)b, n4, chunk4); });
  }
  timed("read only", B2 / 2, [&] { hipLaunchKernelGGL(k_read, dim3(2048), dim3(256), 0, 0, (const f4*)a, b, n4); });
  timed("write only NT", B2 / 2, [&] { hipLaunchKernelGGL(k_write, dim3(2048), dim3(256), 0, 0, (f4*)b, n4); });
  timed("hipMemcpyAsync D2D", B2, [&] { CK(hipMemcpyAsync(b, a, n * 4, hipMemcpyDeviceToDevice, 0)); });
  if (lg == 30) {
    // 1024^3 fp32 viewed as [1024 slabs][1024 rows][512 complex]: middle-axis tiles (row stride 512) and first-axis tiles
    // (row stride 512*1024 + pad)
    const int tiles = 512 / 16;
    timed("tile 1024x128B rows, stride 4 KiB (mid axis)", B2, [&] { hipLaunchKernelGGL((k_tile<0>), dim3(1024 * tiles), dim3(512), 0, 0, (const f2*)a, (f2*)b, (size_t)512, tiles, 1024); });
    timed("tile mid axis, NT store", B2, [&] { hipLaunchKernelGGL((k_tile<1>), dim3(1024 * tiles), dim3(512), 0, 0, (const f2*)a, (f2*)b, (size_t)512, tiles, 1024); });
  }
  if (lg == 30) {
    const size_t rs = 512, ss = 512 * 1024;
#define TILE2(TW, ROWS, XM)                                                                                     \
  {                                                                                                             \
    const int tpr = 512 / TW, rbs = 1024 / ROWS;                                                                \
    const unsigned nb = 1024u * tpr * rbs;                                                                      \
    char tag[96];                                                                                               \
    snprintf(tag, 96, "mid-axis tile %d x %d B rows, xmap %d", ROWS, TW * 8, XM);                              \
    timed(tag, B2, [&] { hipLaunchKernelGGL((k_tile2<TW, ROWS, XM>), dim3(nb), dim3(512), 0, 0, (const f2*)a, (f2*)b, rs, ss, tpr, rbs, nb); }); \
  }
#define TILE4(TW, ROWS, XM)                                                                                     \
  {                                                                                                             \
    const int tpr = 512 * 1024 / TW, rbs = 1024 / ROWS;                                                         \
    const unsigned nb = (unsigned)tpr * rbs;                                                                    \
    char tag[96];                                                                                               \
    snprintf(tag, 96, "first-axis PADDED tile %d x %d B rows, xmap %d", ROWS, TW * 8, XM);                     \
    timed(tag, B2, [&] { hipLaunchKernelGGL((k_tile2<TW, ROWS, XM>), dim3(nb), dim3(512), 0, 0, (const f2*)a2, (f2*)b2, (size_t)512 * 1024 + 2080, (size_t)0, tpr, rbs, nb); }); \
  }
    float *a2, *b2;
    CK(hipMalloc(&a2, n * 4 + 1024 * 2080 * 8)); CK(hipMalloc(&b2, n * 4 + 1024 * 2080 * 8));
    CK(hipMemset(a2, 1, n * 4));
    {
      const int tpr = 32; const unsigned nb = 1024u * tpr;
#define FFTL(SPIN, PERS, PREF, XM, LDSK, PADDED)                                                                  \
  {                                                                                                               \
    auto k = k_fft_like<SPIN, PERS, PREF, XM>;                                                                    \
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDSK * 1024));            \
    char tag[128];                                                                                                \
    snprintf(tag, 128, "fft-like %s spin %d persist %d prefetch %d xmap %d LDS %dK", PADDED ? "first-axis" : "mid-axis", SPIN, PERS, PREF, XM, LDSK); \
    const unsigned grid = PERS ? 256 : nb;                                                                        \
    if (PADDED) timed(tag, B2, [&] { hipLaunchKernelGGL(k, dim3(grid), dim3(512), LDSK * 1024, 0, (const f2*)a2, (f2*)b2, (size_t)512 * 1024 + 2080, (size_t)0, 512 * 1024 / 16, nb, (const f2*)nullptr); }); \
    else timed(tag, B2, [&] { hipLaunchKernelGGL(k, dim3(grid), dim3(512), LDSK * 1024, 0, (const f2*)a, (f2*)b, rs, ss, tpr, nb, (const f2*)nullptr); }); \
  }
      {
        f2* tab; CK(hipMalloc(&tab, 1024 * 8)); CK(hipMemset(tab, 0, 1024 * 8));
#define FFTT(SPIN, XM, TWN, LX)                                                                                   \
  {                                                                                                               \
    auto k = k_fft_like<SPIN, 0, 0, XM, TWN, LX>;                                                                 \
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));             \
    char tag[128];                                                                                                \
    snprintf(tag, 128, "fft-like mid-axis spin %d xmap %d twiddle loads %d lds-exchange %d", SPIN, XM, TWN, LX);   \
    timed(tag, B2, [&] { hipLaunchKernelGGL(k, dim3(nb), dim3(512), 128 * 1024, 0, (const f2*)a, (f2*)b, rs, ss, tpr, nb, (const f2*)tab); }); \
  }
        FFTT(20, 1, 0, 0) FFTT(20, 1, 31, 0) FFTT(20, 1, 0, 1) FFTT(20, 1, 31, 1) FFTT(20, 0, 31, 1) FFTT(40, 1, 31, 1) FFTT(0, 1, 31, 1) FFTT(0, 1, 0, 1) FFTT(0, 1, 31, 0)
      }
      FFTL(0, 0, 0, 0, 128, 0) FFTL(20, 0, 0, 0, 128, 0) FFTL(20, 0, 0, 1, 128, 0) FFTL(20, 0, 0, 0, 64, 0) FFTL(20, 0, 0, 1, 64, 0)
      FFTL(20, 1, 0, 0, 128, 0) FFTL(20, 1, 1, 0, 128, 0) FFTL(20, 1, 1, 1, 128, 0) FFTL(0, 1, 1, 1, 128, 0)
      FFTL(40, 0, 0, 0, 128, 0) FFTL(40, 1, 1, 1, 128, 0) FFTL(40, 0, 0, 1, 64, 0)
      FFTL(20, 0, 0, 0, 128, 1) FFTL(20, 0, 0, 0, 64, 1) FFTL(20, 1, 1, 0, 128, 1) FFTL(20, 1, 1, 1, 128, 1) FFTL(40, 0, 0, 0, 128, 1) FFTL(40, 1, 1, 0, 128, 1)
    }
    // occupancy: the same copy with a dynamic LDS allocation that admits 4 / 2 / 1 workgroups per CU
    for (int kib : {0, 40, 64, 128}) for (int xm : {0, 1}) {
      const int tpr = 512 / 16; const unsigned nb = 1024u * tpr;
      if (kib > 64) {
        CK(hipFuncSetAttribute((const void*)k_tile2<16, 1024, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, kib * 1024));
        CK(hipFuncSetAttribute((const void*)k_tile2<16, 1024, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, kib * 1024));
      }
      char tag[96]; snprintf(tag, 96, "mid-axis 1024x128B, LDS %d KiB/WG, xmap %d", kib, xm);
      if (xm) timed(tag, B2, [&] { hipLaunchKernelGGL((k_tile2<16, 1024, 1>), dim3(nb), dim3(512), kib * 1024, 0, (const f2*)a, (f2*)b, rs, ss, tpr, 1, nb); });
      else timed(tag, B2, [&] { hipLaunchKernelGGL((k_tile2<16, 1024, 0>), dim3(nb), dim3(512), kib * 1024, 0, (const f2*)a, (f2*)b, rs, ss, tpr, 1, nb); });
    }
    TILE2(8, 1024, 0) TILE2(8, 1024, 1) TILE4(8, 1024, 0) TILE4(8, 1024, 1)
    TILE2(16, 1024, 0) TILE2(16, 1024, 1) TILE2(32, 512, 0) TILE2(32, 512, 1) TILE2(64, 256, 0) TILE2(64, 256, 1)
    TILE2(128, 128, 0) TILE2(512, 32, 0) TILE2(16, 256, 0) TILE2(16, 256, 1)
    // first-axis pattern: rows are whole slabs apart (4 MiB), "slab" index = (b, c-tile)
#define TILE3(TW, ROWS, XM)                                                                                     \
  {                                                                                                             \
    const int tpr = 512 * 1024 / TW, rbs = 1024 / ROWS;                                                         \
    const unsigned nb = (unsigned)tpr * rbs;                                                                    \
    char tag[96];                                                                                               \
    snprintf(tag, 96, "first-axis tile %d x %d B rows, xmap %d", ROWS, TW * 8, XM);                            \
    timed(tag, B2, [&] { hipLaunchKernelGGL((k_tile2<TW, ROWS, XM>), dim3(nb), dim3(512), 0, 0, (const f2*)a, (f2*)b, (size_t)512 * 1024, (size_t)0, tpr, rbs, nb); }); \
  }
    TILE3(16, 1024, 0) TILE3(16, 1024, 1) TILE3(32, 512, 0) TILE3(64, 256, 0) TILE3(16, 256, 0)
    // with the padded slab stride of the real work array (2080 complex elements)
    TILE4(16, 1024, 0) TILE4(16, 1024, 1) TILE4(32, 512, 0) TILE4(32, 512, 1) TILE4(64, 256, 0)
    {
      const unsigned nb = 512u * 512u;  // half of the (a, b) pairs: every line exactly once except self-paired ones
      timed("final-pass pairs, natural", B2, [&] { hipLaunchKernelGGL((k_pair<0>), dim3(nb), dim3(128), 0, 0, (const f4*)a, (f4*)b, nb, (size_t)1024 * 256); });
      timed("final-pass pairs, xmap", B2, [&] { hipLaunchKernelGGL((k_pair<1>), dim3(nb), dim3(128), 0, 0, (const f4*)a, (f4*)b, nb, (size_t)1024 * 256); });
      timed("final-pass pairs PADDED, natural", B2, [&] { hipLaunchKernelGGL((k_pair<0>), dim3(nb), dim3(128), 0, 0, (const f4*)a2, (f4*)b2, nb, (size_t)1024 * 256 + 1040); });
      timed("final-pass pairs PADDED, xmap", B2, [&] { hipLaunchKernelGGL((k_pair<1>), dim3(nb), dim3(128), 0, 0, (const f4*)a2, (f4*)b2, nb, (size_t)1024 * 256 + 1040); });
    }
  }
  return 0;
}
