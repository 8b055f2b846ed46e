// Operand layout of v_mfma_f32_4x4x1_16b_f32: 16 independent 4x4 outer products per instruction.
// Prints, for a few lanes, which (block, row, column) each output register holds.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(float* out) {
  const int l = threadIdx.x;
  // a = 100 * lane + 1, b = lane + 1: the product identifies both source lanes
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(l + 1), 1000.f * (l + 1), c, 0, 0, 0);
  for (int e = 0; e < 4; ++e) out[l * 4 + e] = c[e];
}
int main() {
  float* d;
  hipMalloc(&d, 256 * sizeof(float));
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  float h[256];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int e = 0; e < 4; ++e) {
      // which (a lane, b lane) produced it: value = (la + 1) * 1000 * (lb + 1)
      const double v = h[l * 4 + e] / 1000.0;
      int la = -1, lb = -1;
      for (int x = 0; x < 64 && la < 0; ++x)
        for (int y = 0; y < 64; ++y)
          if ((double)(x + 1) * (y + 1) == v && (x >> 2) == (l >> 2) && (y >> 2) == (l >> 2)) { la = x; lb = y; break; }
      if (l < 8 || l == 63) printf("lane %2d reg %d: a from lane %2d, b from lane %2d\n", l, e, la, lb);
      // expected: a from lane 4*(l/4) + e (row e of the block), b from lane l (column = lane)
      if (la != 4 * (l >> 2) + e || lb != l) ++bad;
    }
  printf("layout D[reg e][lane l] = A[lane 4*(l/4)+e] * B[lane l]: %s\n", bad ? "NO" : "yes");
  return 0;
}
