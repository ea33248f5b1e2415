// pk_fma_bench.hip -- issue rate of packed fp32 arithmetic (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) against the scalar
// v_fma_f32 on gfx950, per FLOP, at the register pressure and occupancy of the transform pass kernels (VERDICT r3 item 3:
// k3_mid<float,1024,false> issues 1615 scalar and 445 packed fp32 instructions; is a packed butterfly worth writing?).
// Each kernel runs a long dependent-chain-free stream of complex multiply-accumulates on NREG complex registers:
//   scalar: 4 v_fma_f32 per complex multiply-add      packed: 2 v_pk_fma_f32 (on (re,im) pairs, with one swizzled operand)
// build: hipcc --offload-arch=gfx950 -O3 -o pk_fma_bench pk_fma_bench.hip ; run: ./pk_fma_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err_)); exit(1);} } while (0)
typedef float __attribute__((ext_vector_type(2))) f2;

// The compiler's SLP vectoriser turns plain C++ of either form into a mix of packed and scalar instructions, so the streams
// are written in inline assembly: NREG independent accumulators, no dependence between neighbouring instructions.
// scalar complex multiply-add z <- z * w + c: 4 v_fma_f32;  packed: v_pk_mul_f32 + v_pk_fma_f32 with op_sel / neg_lo
// (lane 0: -z.y * w.y + t.x, lane 1: z.x * w.y + t.y) -- here issued as 2 v_pk_fma_f32, the same issue slots.
template <int NREG, int WAVES_PER_EU>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES_PER_EU, WAVES_PER_EU)))
k_scalar(float* out, int iters, float wr, float wi) {
  float re[NREG], im[NREG];
#pragma unroll
  for (int i = 0; i < NREG; ++i) re[i] = threadIdx.x * 1e-3f + i, im[i] = blockIdx.x * 1e-3f - i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NREG; ++i) {
      asm volatile("v_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3\n\tv_fma_f32 %0, %0, %3, %2\n\tv_fma_f32 %1, %1, %3, %2"
                   : "+v"(re[i]), "+v"(im[i]) : "v"(wr), "v"(wi));
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NREG; ++i) s += re[i] + im[i];
  if (s == 12345.678f) out[0] = s;
}
template <int NREG, int WAVES_PER_EU>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES_PER_EU, WAVES_PER_EU)))
k_packed(float* out, int iters, float wr, float wi) {
  f2 z[NREG];
#pragma unroll
  for (int i = 0; i < NREG; ++i) z[i] = f2{threadIdx.x * 1e-3f + i, blockIdx.x * 1e-3f - i};
  const f2 w = {wr, wi};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NREG; ++i) {
      asm volatile("v_pk_fma_f32 %0, %0, %1, %1 op_sel:[0,0,1] op_sel_hi:[1,0,1]\n\t"
                   "v_pk_fma_f32 %0, %0, %1, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]"
                   : "+v"(z[i]) : "v"(w));
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NREG; ++i) s += z[i].x + z[i].y;
  if (s == 12345.678f) out[0] = s;
}
// butterfly streams (a, b) <- (a + b, a - b): scalar 4 v_add / v_sub, packed 2 v_pk_add_f32 (neg_lo / neg_hi for the difference)
template <int NREG, int WAVES_PER_EU>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES_PER_EU, WAVES_PER_EU)))
k_bfly_scalar(float* out, int iters) {
  float re[NREG], im[NREG];
#pragma unroll
  for (int i = 0; i < NREG; ++i) re[i] = threadIdx.x * 1e-3f + i, im[i] = blockIdx.x * 1e-3f - i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NREG; i += 2) {
      asm volatile("v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %3\n\tv_sub_f32 %2, %0, %2\n\tv_sub_f32 %3, %1, %3"
                   : "+v"(re[i]), "+v"(im[i]), "+v"(re[i + 1]), "+v"(im[i + 1]));
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NREG; ++i) s += re[i] + im[i];
  if (s == 12345.678f) out[0] = s;
}
template <int NREG, int WAVES_PER_EU>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES_PER_EU, WAVES_PER_EU)))
k_bfly_packed(float* out, int iters) {
  f2 z[NREG];
#pragma unroll
  for (int i = 0; i < NREG; ++i) z[i] = f2{threadIdx.x * 1e-3f + i, blockIdx.x * 1e-3f - i};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NREG; i += 2) {
      asm volatile("v_pk_add_f32 %0, %0, %1\n\tv_pk_add_f32 %1, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(z[i]), "+v"(z[i + 1]));
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NREG; ++i) s += z[i].x + z[i].y;
  if (s == 12345.678f) out[0] = s;
}

template <typename K>
static float run(K kern, float* out, int iters, int blocks, float wr, float wi, bool two) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  if (two) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters, wr, wi);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  if (two) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters, wr, wi);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms;
}
template <typename K>
static float run1(K kern, float* out, int iters, int blocks) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  float* out;
  CK(hipMalloc(&out, 64));
  const int iters = 4096;
  printf("device %s, %d CUs; complex multiply-add = 8 flop, butterfly pair = 4 flop\n", prop.name, cus);
#define ROW(NREG, W)                                                                                                     \
  {                                                                                                                      \
    const int blocks = cus * W;  /* W waves per SIMD: 4 SIMDs x W waves = W workgroups of 256 threads per CU */         \
    const double flop = (double)blocks * 256 * iters * NREG * 8.0;                                                       \
    const float ms_s = run(k_scalar<NREG, W>, out, iters, blocks, 0.999f, 0.01f, true);                                  \
    const float ms_p = run(k_packed<NREG, W>, out, iters, blocks, 0.999f, 0.01f, true);                                  \
    const double flopb = (double)blocks * 256 * iters * (NREG / 2) * 4.0;                                                \
    const float mb_s = run1(k_bfly_scalar<NREG, W>, out, iters, blocks);                                                 \
    const float mb_p = run1(k_bfly_packed<NREG, W>, out, iters, blocks);                                                 \
    printf("%2d complex regs, %d waves/SIMD: cmad scalar %7.2f TF  packed %7.2f TF (x%.2f) | butterfly scalar %7.2f TF  packed %7.2f TF (x%.2f)\n", \
           NREG, W, flop / ms_s * 1e-9, flop / ms_p * 1e-9, ms_s / ms_p, flopb / mb_s * 1e-9, flopb / mb_p * 1e-9, mb_s / mb_p); \
  }
  ROW(16, 1) ROW(16, 2) ROW(16, 4) ROW(32, 1) ROW(32, 2) ROW(32, 4) ROW(64, 1) ROW(64, 2)
  return 0;
}
