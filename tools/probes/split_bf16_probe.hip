// Probe: an fp32 matrix product formed from bf16 pieces on the bf16 MFMA, against the f32 MFMA.
//   C[32x32] = A[32xK] * B[Kx32], K = 128, one wave.
// Variants: f32 (v_mfma_f32_32x32x2_f32), bf16x1, bf16x3 (a1b1 + a1b2 + a2b1), bf16x6 (+ a1b3 + a2b2 + a3b1).
// Prints the error of each against an fp64 host product (uniform operands, and a cancelling case whose
// exact result is ~1e-3 of the operand scale) and the shader clocks per product.
//   build: hipcc -O3 --offload-arch=gfx950 -o tools/probes/split_bf16_probe tools/probes/split_bf16_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int K = 128;

__device__ __forceinline__ __bf16 to_bf16(float x) { return (__bf16)x; }

// lane l: row/col l & 31; f32 MFMA takes k = l >> 5 of each pair; bf16 MFMA takes eight consecutive k: 8 * (l >> 5) ..
template <int variant>
__global__ void __launch_bounds__(64) probe(const float* A, const float* B, float* C, long long* clocks, int reps) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  f32x16 acc;
  long long t0 = 0, t1 = 0;
  if (variant == 0) {
    float a[K / 2], b[K / 2];
#pragma unroll
    for (int s = 0; s < K / 2; ++s) { a[s] = A[r * K + 2 * s + h]; b[s] = B[(2 * s + h) * 32 + r]; }
    t0 = clock64();
    for (int rep = 0; rep < reps; ++rep) {
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
      for (int s = 0; s < K / 2; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
    }
    t1 = clock64();
  } else {
    constexpr int NB = K / 16;
    bf16x8 a1[NB], a2[NB], a3[NB], b1[NB], b2[NB], b3[NB];
#pragma unroll
    for (int blk = 0; blk < NB; ++blk)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = blk * 16 + 8 * h + e;
        float x = A[r * K + k];
        __bf16 p1 = to_bf16(x); x -= (float)p1;
        __bf16 p2 = to_bf16(x); x -= (float)p2;
        a1[blk][e] = p1; a2[blk][e] = p2; a3[blk][e] = to_bf16(x);
        float y = B[k * 32 + r];
        __bf16 q1 = to_bf16(y); y -= (float)q1;
        __bf16 q2 = to_bf16(y); y -= (float)q2;
        b1[blk][e] = q1; b2[blk][e] = q2; b3[blk][e] = to_bf16(y);
      }
    t0 = clock64();
    for (int rep = 0; rep < reps; ++rep) {
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
      for (int blk = 0; blk < NB; ++blk) {
        // smallest terms first
        if (variant >= 6) {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[blk], b3[blk], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[blk], b1[blk], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[blk], b2[blk], acc, 0, 0, 0);
        }
        if (variant >= 3) {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[blk], b2[blk], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[blk], b1[blk], acc, 0, 0, 0);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[blk], b1[blk], acc, 0, 0, 0);
      }
    }
    t1 = clock64();
  }
  for (int i = 0; i < 16; ++i) C[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = acc[i];
  if (l == 0) clocks[0] = t1 - t0;
}

int main() {
  std::vector<float> A(32 * K), B(K * 32);
  float *dA, *dB, *dC;
  long long* dclk;
  if (hipMalloc(&dA, A.size() * 4) != hipSuccess || hipMalloc(&dB, B.size() * 4) != hipSuccess ||
      hipMalloc(&dC, 32 * 32 * 4) != hipSuccess || hipMalloc(&dclk, 8) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
  const int reps = 200;
  for (int data = 0; data < 2; ++data) {
    srand(7 + data);
    for (auto& x : A) x = 2.f * rand() / RAND_MAX - 1.f;
    for (auto& x : B) x = 2.f * rand() / RAND_MAX - 1.f;
    if (data == 1)      // cancelling: the second half of every row of A nearly undoes the first half against B
      for (int i = 0; i < 32; ++i)
        for (int k = 0; k < K / 2; ++k) {
          A[i * K + K / 2 + k] = -A[i * K + k] * (1.f + 1e-3f * (rand() / (float)RAND_MAX));
          for (int j = 0; j < 32; ++j) B[(K / 2 + k) * 32 + j] = B[k * 32 + j];
        }
    std::vector<double> ref(32 * 32, 0.0);
    double scale = 0.0;
    for (int i = 0; i < 32; ++i)
      for (int j = 0; j < 32; ++j) {
        double s = 0.0, sa = 0.0;
        for (int k = 0; k < K; ++k) { s += (double)A[i * K + k] * B[k * 32 + j]; sa += fabs((double)A[i * K + k] * B[k * 32 + j]); }
        ref[i * 32 + j] = s;
        scale = fmax(scale, sa);
      }
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    printf("%s operands (sum of |terms| up to %.1f):\n", data == 0 ? "uniform" : "cancelling", scale);
    const int variants[4] = {0, 1, 3, 6};
    for (int v = 0; v < 4; ++v) {
      switch (variants[v]) {
        case 0: hipLaunchKernelGGL(probe<0>, dim3(1), dim3(64), 0, 0, dA, dB, dC, dclk, reps); break;
        case 1: hipLaunchKernelGGL(probe<1>, dim3(1), dim3(64), 0, 0, dA, dB, dC, dclk, reps); break;
        case 3: hipLaunchKernelGGL(probe<3>, dim3(1), dim3(64), 0, 0, dA, dB, dC, dclk, reps); break;
        default: hipLaunchKernelGGL(probe<6>, dim3(1), dim3(64), 0, 0, dA, dB, dC, dclk, reps); break;
      }
      if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
      std::vector<float> C(32 * 32);
      long long clk = 0;
      hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
      hipMemcpy(&clk, dclk, 8, hipMemcpyDeviceToHost);
      double emax = 0.0, rmax = 0.0;
      for (int i = 0; i < 32 * 32; ++i) {
        const double e = fabs(C[i] - ref[i]);
        emax = fmax(emax, e);
        rmax = fmax(rmax, e / fmax(fabs(ref[i]), 1e-30));
      }
      printf("  %-7s max |err| %.3e  (%.2e of the sum of |terms|)  max rel err %.3e  clocks per K=128 product %lld\n",
             v == 0 ? "f32" : (v == 1 ? "bf16x1" : (v == 2 ? "bf16x3" : "bf16x6")), emax, emax / scale, rmax, clk / reps);
    }
  }
  return 0;
}
