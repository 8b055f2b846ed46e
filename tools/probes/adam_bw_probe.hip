// Stream-shape probe for the Adam update at sizes far beyond the caches: which loop shape gets closest to
// the achievable HBM rate (a float4 copy reaches about 6.3 of the 8 TB/s peak).  Same arithmetic in every
// variant (28 B per parameter).  Build: hipcc -O3 --offload-arch=gfx950 adam_bw_probe.hip -o adam_bw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ void one(float& p, float g, float& m, float& v) {
  m = 0.9f * m + 0.1f * g;
  v = 0.999f * v + 0.001f * g * g;
  p -= 1e-3f * m / (sqrtf(v) * 1.0f + 1e-8f);
}
__device__ __forceinline__ void four(float4& P, const float4& G, float4& M, float4& V) {
  one(P.x, G.x, M.x, V.x); one(P.y, G.y, M.y, V.y); one(P.z, G.z, M.z, V.z); one(P.w, G.w, M.w, V.w);
}

typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load(const float4* a) {
  const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(a));
  return make_float4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ void nt_store(const float4& x, float4* a) {
  v4f t; t.x = x.x; t.y = x.y; t.z = x.z; t.w = x.w;
  __builtin_nontemporal_store(t, reinterpret_cast<v4f*>(a));
}

// 0: grid-stride, one float4 per array per iteration
__global__ __launch_bounds__(256) void k_stride(float4* p, const float4* __restrict__ g, float4* m, float4* v, long n4) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 P = p[i]; const float4 G = g[i]; float4 M = m[i]; float4 V = v[i];
    four(P, G, M, V);
    p[i] = P; m[i] = M; v[i] = V;
  }
}
// 1: grid-stride, U float4 per array in flight per thread
template <int U>
__global__ __launch_bounds__(256) void k_unroll(float4* p, const float4* __restrict__ g, float4* m, float4* v, long n4) {
  const long stride = (long)gridDim.x * blockDim.x;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (U - 1) * stride < n4; i += U * stride) {
    float4 P[U], G[U], M[U], V[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { P[u] = p[i + u * stride]; G[u] = g[i + u * stride]; M[u] = m[i + u * stride]; V[u] = v[i + u * stride]; }
#pragma unroll
    for (int u = 0; u < U; ++u) four(P[u], G[u], M[u], V[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) { p[i + u * stride] = P[u]; m[i + u * stride] = M[u]; v[i + u * stride] = V[u]; }
  }
  for (; i < n4; i += stride) {
    float4 P = p[i]; const float4 G = g[i]; float4 M = m[i]; float4 V = v[i];
    four(P, G, M, V);
    p[i] = P; m[i] = M; v[i] = V;
  }
}
// 2: as 1 with non-temporal loads and stores
template <int U>
__global__ __launch_bounds__(256) void k_nt(float4* p, const float4* __restrict__ g, float4* m, float4* v, long n4) {
  const long stride = (long)gridDim.x * blockDim.x;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (U - 1) * stride < n4; i += U * stride) {
    float4 P[U], G[U], M[U], V[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      P[u] = nt_load(p + i + u * stride); G[u] = nt_load(g + i + u * stride);
      M[u] = nt_load(m + i + u * stride); V[u] = nt_load(v + i + u * stride);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) four(P[u], G[u], M[u], V[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      nt_store(P[u], p + i + u * stride); nt_store(M[u], m + i + u * stride); nt_store(V[u], v + i + u * stride);
    }
  }
  for (; i < n4; i += stride) {
    float4 P = p[i]; const float4 G = g[i]; float4 M = m[i]; float4 V = v[i];
    four(P, G, M, V);
    p[i] = P; m[i] = M; v[i] = V;
  }
}
// 3: scalar grid-stride
__global__ __launch_bounds__(256) void k_scalar(float* p, const float* __restrict__ g, float* m, float* v, long n) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) one(p[i], g[i], m[i], v[i]);
}
// 4: plain float4 copy of the same byte count, as the yardstick (reads 16 B, writes 12 B per parameter -> 4 in, 3 out)
__global__ __launch_bounds__(256) void k_copy(float4* p, const float4* __restrict__ g, float4* m, float4* v, long n4) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 a = p[i], b = g[i], c = m[i], d = v[i];
    p[i] = b; m[i] = make_float4(a.x + d.x, a.y, a.z, a.w); v[i] = c;
  }
}

int main(int argc, char** argv) {
  const long n = argc > 1 ? atol(argv[1]) : 100000000L;
  const long n4 = n / 4;
  float *p, *g, *m, *v;
  CK(hipMalloc(&p, n * 4)); CK(hipMalloc(&g, n * 4)); CK(hipMalloc(&m, n * 4)); CK(hipMalloc(&v, n * 4));
  CK(hipMemset(p, 0, n * 4)); CK(hipMemset(g, 0, n * 4)); CK(hipMemset(m, 0, n * 4)); CK(hipMemset(v, 0, n * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float4 *P = (float4*)p, *G = (float4*)g, *M = (float4*)m, *V = (float4*)v;
  const int grids[] = {1024, 2048, 4096, 8192, 16384};
  for (int gi = 0; gi < 5; ++gi) {
    const int grid = grids[gi];
    for (int var = 0; var < 8; ++var) {
      const char* name = "";
      float best = 1e30f;
      for (int rep = 0; rep < 6; ++rep) {
        CK(hipEventRecord(e0));
        switch (var) {
          case 0: name = "float4 stride"; hipLaunchKernelGGL(k_stride, dim3(grid), dim3(256), 0, 0, P, G, M, V, n4); break;
          case 1: name = "float4 x2"; hipLaunchKernelGGL(k_unroll<2>, dim3(grid), dim3(256), 0, 0, P, G, M, V, n4); break;
          case 2: name = "float4 x4"; hipLaunchKernelGGL(k_unroll<4>, dim3(grid), dim3(256), 0, 0, P, G, M, V, n4); break;
          case 3: name = "float4 nt x1"; hipLaunchKernelGGL(k_nt<1>, dim3(grid), dim3(256), 0, 0, P, G, M, V, n4); break;
          case 4: name = "float4 nt x2"; hipLaunchKernelGGL(k_nt<2>, dim3(grid), dim3(256), 0, 0, P, G, M, V, n4); break;
          case 5: name = "float4 nt x4"; hipLaunchKernelGGL(k_nt<4>, dim3(grid), dim3(256), 0, 0, P, G, M, V, n4); break;
          case 6: name = "scalar"; hipLaunchKernelGGL(k_scalar, dim3(grid), dim3(256), 0, 0, p, g, m, v, n); break;
          case 7: name = "copy 4in 3out"; hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, P, G, M, V, n4); break;
        }
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
      }
      printf("grid %5d  %-14s %8.1f us  %7.1f GB/s\n", grid, name, best * 1e3, 28.0 * n / (best * 1e-3) / 1e9);
    }
  }
  CK(hipDeviceSynchronize());
  return 0;
}
