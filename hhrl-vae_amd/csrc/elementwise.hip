// Elementwise / reduction kernels of the VAE loss and the TD3 update rule,
// and the flat-buffer optimiser kernels.  All are HBM-bound streams.
#include "common.h"

#include <cmath>

namespace hhrl {

// ---------------------------------------------------------------- VAE
__global__ void vae_reparam_kernel(const float* __restrict__ ml, const float* __restrict__ eps, float* z, int B, int g) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * g) return;
  const int b = idx / g, j = idx % g;
  const float mean = ml[b * 2 * g + j], lv = ml[b * 2 * g + g + j];
  z[idx] = mean + eps[idx] * expf(0.5f * lv);
}

// One pass over the B*N reconstruction terms and the B*g KL terms.  Each
// thread accumulates privately, a wave reduces with shuffles, a workgroup
// with LDS, and one atomic per workgroup lands in *loss.
__global__ __launch_bounds__(256) void vae_elbo_kernel(const float* __restrict__ decoded, const float* __restrict__ data,
                                                       const float* __restrict__ ml, float recon_scale, float* loss,
                                                       float* d_decoded, int BN, int Bg, int g) {
  __shared__ float red[4];
  const int stride = gridDim.x * blockDim.x;
  const int t0 = blockIdx.x * blockDim.x + threadIdx.x;
  float s = 0.f;
  for (int i = t0; i < BN; i += stride) {
    const float d = decoded[i] - data[2 * i + 1];
    s += recon_scale * d * d;
    d_decoded[i] = 2.f * recon_scale * d;
  }
  for (int i = t0; i < Bg; i += stride) {
    const int b = i / g, j = i % g;
    const float mean = ml[b * 2 * g + j], lv = ml[b * 2 * g + g + j];
    s += -0.5f * (1.f + lv - mean * mean - expf(lv));
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) unsafeAtomicAdd(loss, red[0] + red[1] + red[2] + red[3]);
}

__global__ void vae_latent_grad_kernel(const float* __restrict__ dz, const float* __restrict__ ml,
                                       const float* __restrict__ eps, float* d_ml, int B, int g) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * g) return;
  const int b = idx / g, j = idx % g;
  const float mean = ml[b * 2 * g + j], lv = ml[b * 2 * g + g + j];
  const float d = dz[idx];
  d_ml[b * 2 * g + j] = d + mean;
  d_ml[b * 2 * g + g + j] = d * eps[idx] * 0.5f * expf(0.5f * lv) + 0.5f * (expf(lv) - 1.f);
}

// ---------------------------------------------------------------- TD3
__global__ void td3_smooth_action_kernel(const float* __restrict__ pi, int ldp, const float* __restrict__ eps,
                                         float* out, int ldo, int B, int a) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * a) return;
  const int b = idx / a, j = idx % a;
  float n = 0.2f * eps[idx];
  n = fminf(fmaxf(n, -0.5f), 0.5f);
  const float v = pi[(long)b * ldp + j] + n;
  out[(long)b * ldo + j] = fminf(fmaxf(v, -1.f), 1.f);
}

__global__ void td3_target_q_kernel(const float* __restrict__ q1, const float* __restrict__ q2,
                                    const float* __restrict__ r, int ldr, float gamma, float* y, int B) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < B) y[idx] = r[(long)idx * ldr] + gamma * fminf(q1[idx], q2[idx]);
}

__global__ __launch_bounds__(256) void twin_mse_kernel(const float* __restrict__ q1, const float* __restrict__ q2,
                                                       const float* __restrict__ y, float inv_n, float* loss,
                                                       float* dq1, float* dq2, int B) {
  __shared__ float red[2][4];
  const int stride = gridDim.x * blockDim.x;
  float s1 = 0.f, s2 = 0.f;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B; i += stride) {
    const float d1 = q1[i] - y[i], d2 = q2[i] - y[i];
    s1 += d1 * d1;
    s2 += d2 * d2;
    dq1[i] = 2.f * inv_n * d1;
    dq2[i] = 2.f * inv_n * d2;
  }
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s1; red[1][threadIdx.x >> 6] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsafeAtomicAdd(&loss[0], inv_n * (red[0][0] + red[0][1] + red[0][2] + red[0][3]));
    unsafeAtomicAdd(&loss[1], inv_n * (red[1][0] + red[1][1] + red[1][2] + red[1][3]));
  }
}

// Target value and twin critic loss in one pass: y = r + gamma min(q1t, q2t), then the two mean squared errors
// and their gradients as in twin_mse_kernel.  ONE: a single workgroup walks all rows and stores the two losses;
// otherwise the workgroups add their shares to a loss that was zeroed before the launch.
template <bool ONE>
__global__ __launch_bounds__(1024) void td3_target_mse_kernel(const float* __restrict__ q1t, const float* __restrict__ q2t,
                                                              const float* __restrict__ r, int ldr, float gamma,
                                                              const float* __restrict__ q1, const float* __restrict__ q2,
                                                              float inv_n, float* y, float* loss, float* dq1, float* dq2, int B) {
  __shared__ float red[2][16];
  const int nt = blockDim.x, stride = gridDim.x * nt;
  float s1 = 0.f, s2 = 0.f;
  // four rows per thread and pass, every load issued before the first use (one memory round trip per pass)
  for (int base = blockIdx.x * nt + threadIdx.x; base < B; base += 4 * stride) {
    float a[4], b[4], c[4], d[4], e[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = base + u * stride;
      const bool in = i < B;
      a[u] = in ? r[(long)i * ldr] : 0.f;
      b[u] = in ? q1t[i] : 0.f;
      c[u] = in ? q2t[i] : 0.f;
      d[u] = in ? q1[i] : 0.f;
      e[u] = in ? q2[i] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = base + u * stride;
      if (i >= B) continue;
      const float yi = a[u] + gamma * fminf(b[u], c[u]);
      y[i] = yi;
      const float d1 = d[u] - yi, d2 = e[u] - yi;
      s1 += d1 * d1;
      s2 += d2 * d2;
      dq1[i] = 2.f * inv_n * d1;
      dq2[i] = 2.f * inv_n * d2;
    }
  }
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s1; red[1][threadIdx.x >> 6] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float l1 = 0.f, l2 = 0.f;
    for (int w = 0; w < nt / 64; ++w) {
      l1 += red[0][w];
      l2 += red[1][w];
    }
    l1 *= inv_n;
    l2 *= inv_n;
    if (ONE) {
      loss[0] = l1;
      loss[1] = l2;
    } else {
      unsafeAtomicAdd(&loss[0], l1);
      unsafeAtomicAdd(&loss[1], l2);
    }
  }
}

// ---------------------------------------------------------- optimiser
// One coalesced pass: reads p,g,m,v (16 B each), writes p,m,v: 28 B / parameter.
// copy_dst (optional): one float copied from copy_src by the first thread, for a value that travels behind
// the gradient (the all-reduced loss of a data-parallel step)
typedef float v4f __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ float4 ld4(const float4* a) {
  if (NT) {
    const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(a));
    return make_float4(t.x, t.y, t.z, t.w);
  }
  return *a;
}
template <bool NT> __device__ __forceinline__ void st4(float4* a, const float4& x) {
  if (NT) {
    v4f t;
    t.x = x.x; t.y = x.y; t.z = x.z; t.w = x.w;
    __builtin_nontemporal_store(t, reinterpret_cast<v4f*>(a));
  } else {
    *a = x;
  }
}

// NT: non-temporal loads and stores, for buffers far larger than the caches (nothing of them is read again
// before it is evicted): 5.3 against 4.6 TB/s at 10^8 parameters, where a plain copy with the same four-in,
// three-out stream mix reaches 4.7-5.1 TB/s.  The networks of this package fit the caches and take NT = false.
template <bool NT>
__global__ __launch_bounds__(256) void adam_kernel(float* p, const float* __restrict__ g, float* m, float* v, long n, AdamK k,
                                                   const float* copy_src, float* copy_dst) {
  if (copy_dst != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *copy_dst = *copy_src;
  const long n4 = n >> 2;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 P = ld4<NT>(reinterpret_cast<const float4*>(p) + i);
    const float4 G = ld4<NT>(reinterpret_cast<const float4*>(g) + i);
    float4 Mo = ld4<NT>(reinterpret_cast<const float4*>(m) + i);
    float4 V = ld4<NT>(reinterpret_cast<const float4*>(v) + i);
    adam_one(P.x, G.x, Mo.x, V.x, k);
    adam_one(P.y, G.y, Mo.y, V.y, k);
    adam_one(P.z, G.z, Mo.z, V.z, k);
    adam_one(P.w, G.w, Mo.w, V.w, k);
    st4<NT>(reinterpret_cast<float4*>(p) + i, P);
    st4<NT>(reinterpret_cast<float4*>(m) + i, Mo);
    st4<NT>(reinterpret_cast<float4*>(v) + i, V);
  }
  const long tail = n4 << 2;
  const long i = tail + (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) adam_one(p[i], g[i], m[i], v[i], k);
}

__global__ __launch_bounds__(256) void adam_scalar_kernel(float* p, const float* __restrict__ g, float* m, float* v, long n, AdamK k,
                                                          const float* copy_src, float* copy_dst) {
  if (copy_dst != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *copy_dst = *copy_src;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) adam_one(p[i], g[i], m[i], v[i], k);
}

// Same update with the scalars read from device memory, so that a captured HIP graph can be
// replayed with a new step count / learning rate (the host refreshes the 8 floats per step).
// Writes the scalars of one Adam step into device memory (they travel as kernel arguments, so
// the host may stage the next step at once).
__global__ void adam_stage_kernel(float* hyper, AdamK k) {
  if (threadIdx.x == 0) {
    hyper[0] = k.step_size; hyper[1] = k.beta1; hyper[2] = k.beta2; hyper[3] = k.one_minus_b1;
    hyper[4] = k.one_minus_b2; hyper[5] = k.bc2_sqrt; hyper[6] = k.eps; hyper[7] = k.gscale;
  }
}


// VEC: float4 accesses (all base pointers 16-byte aligned) with a scalar tail.  POLYAK: the target network's
// copy of each parameter moves towards the new value in the same pass (target = tau p + (1 - tau) target,
// 36 instead of 28 + 12 bytes per parameter and one launch instead of two).
template <bool VEC, bool POLYAK>
__global__ __launch_bounds__(256) void adam_dev_kernel(float* p, const float* __restrict__ g, float* m, float* v, long n,
                                                       const float* __restrict__ hyper, float* t, float tau, float om) {
  AdamK k;
  k.step_size = hyper[0]; k.beta1 = hyper[1]; k.beta2 = hyper[2]; k.one_minus_b1 = hyper[3];
  k.one_minus_b2 = hyper[4]; k.bc2_sqrt = hyper[5]; k.eps = hyper[6]; k.gscale = hyper[7];
  const long stride = (long)gridDim.x * blockDim.x;
  const long first = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long done = 0;
  if (VEC) {
    const long n4 = n >> 2;
    for (long i = first; i < n4; i += stride) {
      float4 P = reinterpret_cast<float4*>(p)[i];
      const float4 G = reinterpret_cast<const float4*>(g)[i];
      float4 Mo = reinterpret_cast<float4*>(m)[i];
      float4 V = reinterpret_cast<float4*>(v)[i];
      adam_one(P.x, G.x, Mo.x, V.x, k);
      adam_one(P.y, G.y, Mo.y, V.y, k);
      adam_one(P.z, G.z, Mo.z, V.z, k);
      adam_one(P.w, G.w, Mo.w, V.w, k);
      reinterpret_cast<float4*>(p)[i] = P;
      reinterpret_cast<float4*>(m)[i] = Mo;
      reinterpret_cast<float4*>(v)[i] = V;
      if (POLYAK) {
        float4 T = reinterpret_cast<float4*>(t)[i];
        T.x = polyak_one(T.x, P.x, tau, om);
        T.y = polyak_one(T.y, P.y, tau, om);
        T.z = polyak_one(T.z, P.z, tau, om);
        T.w = polyak_one(T.w, P.w, tau, om);
        reinterpret_cast<float4*>(t)[i] = T;
      }
    }
    done = n4 << 2;
  }
  for (long i = done + first; i < n; i += stride) {
    adam_one(p[i], g[i], m[i], v[i], k);
    if (POLYAK) t[i] = polyak_one(t[i], p[i], tau, om);
  }
}

__global__ __launch_bounds__(256) void polyak_kernel(float* t, const float* __restrict__ p, long n, float tau, float om, int vec) {
  const long stride = (long)gridDim.x * blockDim.x;
  if (vec) {
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
      const float4 P = reinterpret_cast<const float4*>(p)[i];
      float4 T = reinterpret_cast<float4*>(t)[i];
      T.x = polyak_one(T.x, P.x, tau, om);
      T.y = polyak_one(T.y, P.y, tau, om);
      T.z = polyak_one(T.z, P.z, tau, om);
      T.w = polyak_one(T.w, P.w, tau, om);
      reinterpret_cast<float4*>(t)[i] = T;
    }
    const long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) t[i] = polyak_one(t[i], p[i], tau, om);
  } else {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) t[i] = polyak_one(t[i], p[i], tau, om);
  }
}

static inline bool aligned16(const void* a) { return (reinterpret_cast<uintptr_t>(a) & 15) == 0; }
static inline int stream_grid(long n_items) {
  long b = (n_items + 255) / 256;
  if (b < 1) b = 1;
  if (b > 2048) b = 2048;      // 256 CUs x 8 resident workgroups; grid-stride beyond
  return (int)b;
}

AdamK make_adam_k(double lr, double beta1, double beta2, double eps, int step, double grad_scale) {
  // Bias corrections in double on the host, as torch computes them.
  const double bc1 = 1.0 - std::pow(beta1, step);
  const double bc2 = 1.0 - std::pow(beta2, step);
  AdamK k;
  k.step_size = (float)(lr / bc1);
  k.beta1 = (float)beta1;
  k.beta2 = (float)beta2;
  k.one_minus_b1 = (float)(1.0 - beta1);
  k.one_minus_b2 = (float)(1.0 - beta2);
  k.bc2_sqrt = (float)std::sqrt(bc2);
  k.eps = (float)eps;
  k.gscale = (float)grad_scale;
  return k;
}

}  // namespace hhrl

using namespace hhrl;

extern "C" {

int hhrl_vae_reparam(const float* ml, const float* eps, float* z, int B, int g, void* stream) {
  HHRL_REQUIRE(ml && eps && z, "vae_reparam: null pointer");
  HHRL_REQUIRE(B > 0 && g > 0, "vae_reparam: bad shape B=%d g=%d", B, g);
  ProfScope ps("vae_reparam_kernel", as_stream(stream), 0.0, 12.0 * B * g * 4 / 3);
  hipLaunchKernelGGL(vae_reparam_kernel, dim3(ceil_div((long)B * g, 256)), dim3(256), 0, as_stream(stream), ml, eps, z, B, g);
  return launch_status("vae_reparam_kernel");
}

int hhrl_vae_elbo(const float* decoded, const float* data, const float* ml, float recon_scale, float* loss,
                  float* d_decoded, int B, int N, int g, void* stream) {
  HHRL_REQUIRE(decoded && data && ml && loss && d_decoded, "vae_elbo: null pointer");
  HHRL_REQUIRE(B > 0 && N > 0 && g > 0, "vae_elbo: bad shape B=%d N=%d g=%d", B, N, g);
  const long bn = (long)B * N;
  HHRL_REQUIRE(bn < (1L << 30), "vae_elbo: B*N too large");
  const int grid = stream_grid(bn > (long)B * g ? bn : (long)B * g);
  ProfScope ps("vae_elbo_kernel", as_stream(stream), 0.0, 4.0 * (3.0 * bn + 2.0 * B * g));
  hipLaunchKernelGGL(vae_elbo_kernel, dim3(grid), dim3(256), 0, as_stream(stream), decoded, data, ml, recon_scale, loss,
                     d_decoded, (int)bn, B * g, g);
  return launch_status("vae_elbo_kernel");
}

int hhrl_vae_latent_grad(const float* dz, const float* ml, const float* eps, float* d_ml, int B, int g, void* stream) {
  HHRL_REQUIRE(dz && ml && eps && d_ml, "vae_latent_grad: null pointer");
  HHRL_REQUIRE(B > 0 && g > 0, "vae_latent_grad: bad shape B=%d g=%d", B, g);
  ProfScope ps("vae_latent_grad_kernel", as_stream(stream), 0.0, 24.0 * B * g);
  hipLaunchKernelGGL(vae_latent_grad_kernel, dim3(ceil_div((long)B * g, 256)), dim3(256), 0, as_stream(stream), dz, ml, eps, d_ml, B, g);
  return launch_status("vae_latent_grad_kernel");
}

int hhrl_td3_smooth_action(const float* pi, int ldp, const float* eps, float* out, int ldo, int B, int a, void* stream) {
  HHRL_REQUIRE(pi && eps && out, "td3_smooth_action: null pointer");
  HHRL_REQUIRE(B > 0 && a > 0 && ldp >= a && ldo >= a, "td3_smooth_action: bad shape B=%d a=%d ldp=%d ldo=%d", B, a, ldp, ldo);
  ProfScope ps("td3_smooth_action_kernel", as_stream(stream), 0.0, 12.0 * B * a);
  hipLaunchKernelGGL(td3_smooth_action_kernel, dim3(ceil_div((long)B * a, 256)), dim3(256), 0, as_stream(stream), pi, ldp, eps, out, ldo, B, a);
  return launch_status("td3_smooth_action_kernel");
}

int hhrl_td3_target_q(const float* q1, const float* q2, const float* r, int ldr, float gamma, float* y, int B,
                      void* stream) {
  HHRL_REQUIRE(q1 && q2 && r && y, "td3_target_q: null pointer");
  HHRL_REQUIRE(B > 0 && ldr >= 1, "td3_target_q: B=%d ldr=%d", B, ldr);
  ProfScope ps("td3_target_q_kernel", as_stream(stream), 0.0, 16.0 * B);
  hipLaunchKernelGGL(td3_target_q_kernel, dim3(ceil_div(B, 256)), dim3(256), 0, as_stream(stream), q1, q2, r, ldr, gamma, y, B);
  return launch_status("td3_target_q_kernel");
}

int hhrl_twin_mse(const float* q1, const float* q2, const float* y, float inv_n, float* loss, float* dq1, float* dq2,
                  int B, void* stream) {
  HHRL_REQUIRE(q1 && q2 && y && loss && dq1 && dq2, "twin_mse: null pointer");
  HHRL_REQUIRE(B > 0, "twin_mse: B=%d", B);
  ProfScope ps("twin_mse_kernel", as_stream(stream), 0.0, 20.0 * B);
  hipLaunchKernelGGL(twin_mse_kernel, dim3(stream_grid(B)), dim3(256), 0, as_stream(stream), q1, q2, y, inv_n, loss, dq1, dq2, B);
  return launch_status("twin_mse_kernel");
}

static int td3_target_mse_impl(const float* q1t, const float* q2t, const float* r, int ldr, float gamma, const float* q1,
                               const float* q2, float inv_n, float* y, float* loss, float* dq1, float* dq2, int B, bool zeroed,
                               void* stream) {
  HHRL_REQUIRE(q1t && q2t && r && q1 && q2 && y && loss && dq1 && dq2, "td3_target_mse: null pointer");
  HHRL_REQUIRE(B > 0 && ldr >= 1, "td3_target_mse: B=%d ldr=%d", B, ldr);
  hipStream_t st = as_stream(stream);
  ProfScope ps("td3_target_mse_kernel", st, 0.0, 32.0 * B);
  if (B < 2048) {
    // one workgroup (deterministic loss, no memset); beyond, one CU's load rate on the strided reward column makes it
    // longer than a memset node plus a grid of workgroups that add their shares (10 -> 4 us at 4096 rows)
    hipLaunchKernelGGL(td3_target_mse_kernel<true>, dim3(1), dim3(B >= 1024 ? 1024 : 256), 0, st, q1t, q2t, r, ldr, gamma, q1, q2,
                       inv_n, y, loss, dq1, dq2, B);
  } else {
    if (!zeroed && hipMemsetAsync(loss, 0, 2 * sizeof(float), st) != hipSuccess)
      return fail(HHRL_ERR_HIP, "td3_target_mse: memset failed");
    hipLaunchKernelGGL(td3_target_mse_kernel<false>, dim3(stream_grid(B)), dim3(256), 0, st, q1t, q2t, r, ldr, gamma, q1, q2,
                       inv_n, y, loss, dq1, dq2, B);
  }
  return launch_status("td3_target_mse_kernel");
}

int hhrl_td3_target_mse(const float* q1t, const float* q2t, const float* r, int ldr, float gamma, const float* q1,
                        const float* q2, float inv_n, float* y, float* loss, float* dq1, float* dq2, int B, void* stream) {
  return td3_target_mse_impl(q1t, q2t, r, ldr, gamma, q1, q2, inv_n, y, loss, dq1, dq2, B, false, stream);
}

int hhrl_td3_target_mse_zeroed(const float* q1t, const float* q2t, const float* r, int ldr, float gamma, const float* q1,
                               const float* q2, float inv_n, float* y, float* loss, float* dq1, float* dq2, int B, void* stream) {
  return td3_target_mse_impl(q1t, q2t, r, ldr, gamma, q1, q2, inv_n, y, loss, dq1, dq2, B, true, stream);
}

static int adam_launch(float* p, const float* g, float* m, float* v, long n, double lr, double beta1, double beta2, double eps,
                       int step, double grad_scale, const float* copy_src, float* copy_dst, void* stream) {
  HHRL_REQUIRE(p && g && m && v, "adam_step: null pointer");
  HHRL_REQUIRE(n > 0 && step >= 1, "adam_step: n=%ld step=%d", n, step);
  const AdamK k = make_adam_k(lr, beta1, beta2, eps, step, grad_scale);
  hipStream_t st = as_stream(stream);
  ProfScope ps("adam_kernel", st, 0.0, 28.0 * n);
  if (aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v)) {
    if (n >= (64L << 20))      // 4 x 256 MB and more: beyond the 256 MB Infinity Cache
      hipLaunchKernelGGL(adam_kernel<true>, dim3(stream_grid((n + 3) / 4)), dim3(256), 0, st, p, g, m, v, n, k, copy_src, copy_dst);
    else
      hipLaunchKernelGGL(adam_kernel<false>, dim3(stream_grid((n + 3) / 4)), dim3(256), 0, st, p, g, m, v, n, k, copy_src, copy_dst);
  } else {
    hipLaunchKernelGGL(adam_scalar_kernel, dim3(stream_grid(n)), dim3(256), 0, st, p, g, m, v, n, k, copy_src, copy_dst);
  }
  return launch_status("adam_kernel");
}

int hhrl_adam_step(float* p, const float* g, float* m, float* v, long n, double lr, double beta1, double beta2,
                   double eps, int step, double grad_scale, void* stream) {
  return adam_launch(p, g, m, v, n, lr, beta1, beta2, eps, step, grad_scale, nullptr, nullptr, stream);
}

int hhrl_adam_step_copy(float* p, const float* g, float* m, float* v, long n, double lr, double beta1, double beta2,
                        double eps, int step, double grad_scale, const float* copy_src, float* copy_dst, void* stream) {
  HHRL_REQUIRE(copy_src && copy_dst, "adam_step_copy: null pointer");
  return adam_launch(p, g, m, v, n, lr, beta1, beta2, eps, step, grad_scale, copy_src, copy_dst, stream);
}

int hhrl_adam_stage(float* hyper_dev, double lr, double beta1, double beta2, double eps, int step, double grad_scale,
                    void* stream) {
  HHRL_REQUIRE(hyper_dev && step >= 1, "adam_stage: null pointer or step < 1");
  const AdamK k = make_adam_k(lr, beta1, beta2, eps, step, grad_scale);
  hipLaunchKernelGGL(adam_stage_kernel, dim3(1), dim3(64), 0, as_stream(stream), hyper_dev, k);
  return launch_status("adam_stage_kernel");
}

static int adam_dev_launch(float* p, const float* g, float* m, float* v, long n, const float* hyper_dev, float* target,
                           double tau, void* stream) {
  hipStream_t st = as_stream(stream);
  const bool vec = aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v) && (target == nullptr || aligned16(target));
  ProfScope ps(target ? "adam_polyak_dev_kernel" : "adam_dev_kernel", st, 0.0, (target ? 36.0 : 28.0) * n);
  const dim3 grid(stream_grid(vec ? (n + 3) / 4 : n));
  const float ft = (float)tau, om = (float)(1.0 - tau);
  if (vec && target) hipLaunchKernelGGL((adam_dev_kernel<true, true>), grid, dim3(256), 0, st, p, g, m, v, n, hyper_dev, target, ft, om);
  else if (vec) hipLaunchKernelGGL((adam_dev_kernel<true, false>), grid, dim3(256), 0, st, p, g, m, v, n, hyper_dev, target, ft, om);
  else if (target) hipLaunchKernelGGL((adam_dev_kernel<false, true>), grid, dim3(256), 0, st, p, g, m, v, n, hyper_dev, target, ft, om);
  else hipLaunchKernelGGL((adam_dev_kernel<false, false>), grid, dim3(256), 0, st, p, g, m, v, n, hyper_dev, target, ft, om);
  return launch_status("adam_dev_kernel");
}

int hhrl_adam_step_dev(float* p, const float* g, float* m, float* v, long n, const float* hyper_dev, void* stream) {
  HHRL_REQUIRE(p && g && m && v && hyper_dev, "adam_step_dev: null pointer");
  HHRL_REQUIRE(n > 0, "adam_step_dev: n=%ld", n);
  return adam_dev_launch(p, g, m, v, n, hyper_dev, nullptr, 0.0, stream);
}

int hhrl_adam_polyak_step_dev(float* p, const float* g, float* m, float* v, long n, const float* hyper_dev, float* target,
                              double tau, void* stream) {
  HHRL_REQUIRE(p && g && m && v && hyper_dev && target, "adam_polyak_step_dev: null pointer");
  HHRL_REQUIRE(n > 0 && tau >= 0.0 && tau <= 1.0, "adam_polyak_step_dev: n=%ld tau=%g", n, tau);
  return adam_dev_launch(p, g, m, v, n, hyper_dev, target, tau, stream);
}

int hhrl_polyak_update(float* target, const float* p, long n, double tau, void* stream) {
  HHRL_REQUIRE(target && p, "polyak_update: null pointer");
  HHRL_REQUIRE(n > 0, "polyak_update: n=%ld", n);
  const int vec = aligned16(target) && aligned16(p);
  ProfScope ps("polyak_kernel", as_stream(stream), 0.0, 12.0 * n);
  hipLaunchKernelGGL(polyak_kernel, dim3(stream_grid(vec ? (n + 3) / 4 : n)), dim3(256), 0, as_stream(stream), target, p, n,
                     (float)tau, (float)(1.0 - tau), vec);
  return launch_status("polyak_kernel");
}

}  // extern "C"
