// Rate at which 256 workgroups of 512 threads can hand 112 KB each to memory, by store form and address pattern
// (the slab traffic of the one-launch VAE step at 4096 lanes, nothing else in the kernel):
//   form:    0 ordinary stores (they stay dirty in the L2; the launch's end writes them back)
//            1 agent-scope stores of 16 bytes (global_store_dwordx4 ... sc1)     2 the same, 4 bytes per lane
//   pattern: 0 a wave's store covers 16 rows x 64 bytes (rows 512 bytes apart: the 16 x 16 gradient blocks)
//            1 a wave's store covers 1 KB of consecutive bytes
// Build: hipcc -O3 --offload-arch=gfx950 -o write_through_probe write_through_probe.hip ; run: ./write_through_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));

template <int FORM, int PATTERN>
__global__ __launch_bounds__(512) void probe(float* slab, int floats_per_wg, int spin) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* base = slab + (long)blockIdx.x * floats_per_wg;
  const int per_wave = floats_per_wg / 8;                  // 3584 floats = 14 KB: 14 stores of 1 KB per wave
  float acc = (float)tid;
  for (int it = 0; it < per_wave / 256; ++it) {
    for (int s = 0; s < spin; ++s) acc = acc * 1.0001f + 0.5f;     // stand-in for the work between two stores
    float* p;
    if (PATTERN == 0) p = base + wave * per_wave + it * 256 + 0;    // block of 16 rows x 16 floats inside a 128-float-wide matrix
    else p = base + wave * per_wave + it * 256;
    long off;
    if (PATTERN == 0) off = (long)(lane & 15) * 128 + 4 * (lane >> 4) + (long)(it & 7) * 16 + (long)(it >> 3) * 2048 - it * 256;
    else off = 4 * lane;
    float* q = p + off;
    if (FORM == 0) {
      *reinterpret_cast<float4*>(q) = make_float4(acc, acc, acc, acc);
    } else if (FORM == 1) {
      const v4f t = {acc, acc, acc, acc};
      asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(q), "v"(t) : "memory");
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) __hip_atomic_store(q + e, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

template <int FORM, int PATTERN>
static void run(float* d, int fpw, int spin, const char* name) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((probe<FORM, PATTERN>), dim3(256), dim3(512), 0, 0, d, fpw, spin);
  hipDeviceSynchronize();
  hipEventRecord(a, 0);
  const int n = 200;
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL((probe<FORM, PATTERN>), dim3(256), dim3(512), 0, 0, d, fpw, spin);
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms = 0.f;
  hipEventElapsedTime(&ms, a, b);
  printf("%-34s spin %5d : %7.2f us per launch\n", name, spin, 1e3f * ms / n);
}

int main() {
  const int fpw = 8 * 3584 * 1;                     // 28672 floats = 112 KB per workgroup
  float* d;
  hipMalloc(&d, (size_t)256 * fpw * 4 + (1 << 20));
  for (int spin : {0, 200, 1000}) {
    run<0, 0>(d, fpw, spin, "ordinary, 16 rows x 64 B");
    run<0, 1>(d, fpw, spin, "ordinary, 1 KB consecutive");
    run<1, 0>(d, fpw, spin, "sc1 x4, 16 rows x 64 B");
    run<1, 1>(d, fpw, spin, "sc1 x4, 1 KB consecutive");
    run<2, 0>(d, fpw, spin, "sc1 dword x4, 16 rows x 64 B");
    run<2, 1>(d, fpw, spin, "sc1 dword x4, 1 KB consecutive");
  }
  hipFree(d);
  return 0;
}
