// What does a hand-off between the workgroups of ONE launch cost on this chip, against the boundary between two
// dependent launches?  n workgroups (one per CU at most, all resident) run `rounds` rounds of: write a value derived
// from the previous round, meet at a barrier built from a device-scope counter (release fence, atomic add, bounded
// spin, acquire fence), read another workgroup's value.  The spin is bounded: a workgroup that waits longer than
// `limit` polls gives up and raises a flag, so the launch always ends.  Printed: microseconds per round (HIP events
// around the launch / rounds), for n = 8 .. 256, and the time of the same chain as `rounds` dependent launches.
//   hipcc -O3 --offload-arch=gfx950 -o grid_barrier_probe grid_barrier_probe.hip && ./grid_barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void chain_kernel(float* buf, unsigned* counter, int* gave_up, int rounds, int limit) {
  const int n = gridDim.x, w = blockIdx.x, tid = threadIdx.x;
  float v = 1.0f + w;
  for (int r = 0; r < rounds; ++r) {
    buf[(r & 1) * n * 64 + w * 64 + (tid & 63)] = v;            // this round's result of the workgroup
    __syncthreads();
    if (tid == 0) {
      __threadfence();                                          // release: the stores above become visible device-wide
      atomicAdd(counter, 1u);
      const unsigned want = (unsigned)(r + 1) * n;
      int polls = 0;
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        __builtin_amdgcn_s_sleep(1);
        if (++polls > limit) { *gave_up = 1; break; }
      }
      __threadfence();                                          // acquire
    }
    __syncthreads();
    // the neighbour's value of this round (written by another workgroup, possibly on another XCD)
    v = 0.5f * v + 0.5f * __builtin_nontemporal_load(&buf[(r & 1) * n * 64 + ((w + 1) % n) * 64 + (tid & 63)]);
  }
  if (tid == 0) buf[2 * n * 64 + w] = v;
}

__global__ void step_kernel(float* buf, int n, int r) {
  const int w = blockIdx.x, tid = threadIdx.x;
  const float prev = r == 0 ? 1.0f + w : 0.5f * buf[((r - 1) & 1) * n * 64 + w * 64 + (tid & 63)] +
                                          0.5f * buf[((r - 1) & 1) * n * 64 + ((w + 1) % n) * 64 + (tid & 63)];
  buf[(r & 1) * n * 64 + w * 64 + (tid & 63)] = prev;
}

int main() {
  const int rounds = 200, limit = 200000;
  float* buf;
  unsigned* counter;
  int* gave_up;
  hipMalloc(&buf, (2 * 256 * 64 + 256) * sizeof(float));
  hipMalloc(&counter, sizeof(unsigned));
  hipMalloc(&gave_up, sizeof(int));
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int n : {8, 16, 32, 64, 128, 256}) {
    for (int threads : {64, 512}) {
      float best = 1e9f;
      int flag = 0;
      for (int rep = 0; rep < 5; ++rep) {
        hipMemset(counter, 0, sizeof(unsigned));
        hipMemset(gave_up, 0, sizeof(int));
        hipDeviceSynchronize();
        hipEventRecord(a, 0);
        hipLaunchKernelGGL(chain_kernel, dim3(n), dim3(threads), 0, 0, buf, counter, gave_up, rounds, limit);
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
        int f;
        hipMemcpy(&f, gave_up, sizeof(int), hipMemcpyDeviceToHost);
        flag |= f;
      }
      printf("in-kernel hand-off: %3d workgroups x %3d threads: %6.2f us per round%s\n", n, threads, 1e3f * best / rounds,
             flag ? "  (a workgroup gave up waiting)" : "");
    }
  }
  for (int n : {8, 64, 256}) {
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      hipDeviceSynchronize();
      hipEventRecord(a, 0);
      for (int r = 0; r < rounds; ++r) hipLaunchKernelGGL(step_kernel, dim3(n), dim3(64), 0, 0, buf, n, r);
      hipEventRecord(b, 0);
      hipEventSynchronize(b);
      float ms;
      hipEventElapsedTime(&ms, a, b);
      if (ms < best) best = ms;
    }
    printf("dependent launches:  %3d workgroups x  64 threads: %6.2f us per launch\n", n, 1e3f * best / rounds);
  }
  return 0;
}
