// Shared helpers for the gfx950 kernels of libhhrl_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdarg>
#include <cstdio>

#include "../../include/hhrl_hip.h"

namespace hhrl {

// Thread-local text of the last error returned through the C ABI.
char* err_buf();
int fail(int code, const char* fmt, ...);

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Checks the launch that was just enqueued.
inline int launch_status(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(HHRL_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
  return HHRL_OK;
}

#define HHRL_REQUIRE(cond, ...) \
  do {                          \
    if (!(cond)) return ::hhrl::fail(HHRL_ERR_ARG, __VA_ARGS__); \
  } while (0)

#define HHRL_TRY(expr)              \
  do {                              \
    int _rc = (expr);               \
    if (_rc != HHRL_OK) return _rc; \
  } while (0)

// Kernels that use more than 64 KiB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize raised.
// The attribute belongs to the CURRENT DEVICE's instance of the function, so it is set once per
// (device, kernel) here, under a lock (first calls may come from several threads or for several devices).
int ensure_dyn_lds(const void* fn, int bytes, const char* who);

constexpr int kWave = 64;  // CDNA wavefront width

// Workgroup barrier for hand-offs through LDS only: waits for this wave's LDS traffic, not for its global
// loads and stores, so that requests issued before the barrier stay in flight across it (__syncthreads() would
// wait for them: its fence covers global memory too).  Only for phases that exchange nothing through global memory.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Stores for results that a LATER kernel reads (gradient slabs, activations handed on): written through the L2
// (agent scope) as they are produced.  Left as ordinary stores they stay
// dirty in the L2 until the kernel ends, and the end of the kernel waits for their write-back: 25 MB of slabs add
// 2.6 us to a 36 us kernel.  -DHHRL_PLAIN_STORES builds the ordinary stores for comparison.
__device__ __forceinline__ void store_through(float* p, float v) {
#ifdef HHRL_PLAIN_STORES
  *p = v;
#else
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
__device__ __forceinline__ void store_through(float4* p, const float4& v) {
#ifdef HHRL_PLAIN_STORES
  *p = v;
#else
  // the 16-byte form of the agent-scope store above (the compiler offers it for 4 bytes only); a streaming store
  // (nt) leaves the kernel as early, but the kernel that reads the data next then takes longer to get it
  typedef float v4f __attribute__((ext_vector_type(4)));
  const v4f t = {v.x, v.y, v.z, v.w};
  // (s_nop: a store of more than 8 bytes reads its data registers up to two cycles after it issues, and inside an
  //  asm statement the compiler does not keep the next instruction from overwriting them)
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(t) : "memory");
#endif
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

__device__ __forceinline__ float act_apply(float v, int act) {
  switch (act) {
    case HHRL_ACT_RELU: return v > 0.f ? v : 0.f;
    case HHRL_ACT_TANH: return tanhf(v);
    case HHRL_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
    default: return v;
  }
}

// d act / d pre-activation, written in terms of the activation OUTPUT y.
__device__ __forceinline__ float act_grad_from_output(float y, int act) {
  switch (act) {
    case HHRL_ACT_RELU: return y > 0.f ? 1.f : 0.f;
    case HHRL_ACT_TANH: return 1.f - y * y;
    case HHRL_ACT_SIGMOID: return y * (1.f - y);
    default: return 1.f;
  }
}

// Scalars of one Adam step (torch.optim.Adam, single-tensor rule), prepared in double on the host.
struct AdamK {
  float step_size, beta1, beta2, one_minus_b1, one_minus_b2, bc2_sqrt, eps, gscale;
};
AdamK make_adam_k(double lr, double beta1, double beta2, double eps, int step, double grad_scale);

// Every operation is spelled out with a rounding intrinsic, so that the compiler cannot contract the
// update differently at different call sites: the stand-alone optimizer kernels and the reduction that
// applies Adam in place give the same bits.
__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamK& k) {
  g = __fmul_rn(g, k.gscale);
  m = __fmaf_rn(k.one_minus_b1, __fsub_rn(g, m), m);                               // exp_avg.lerp_(grad, 1-beta1)
  v = __fmaf_rn(k.one_minus_b2, __fmul_rn(g, g), __fmul_rn(v, k.beta2));           // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1-beta2)
  const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(v), k.bc2_sqrt), k.eps);
  p = __fsub_rn(p, __fmul_rn(k.step_size, __fdiv_rn(m, denom)));
}

// soft update of a target network's copy of a parameter (om = 1 - tau), spelled out like adam_one
__device__ __forceinline__ float polyak_one(float t, float p, float tau, float om) { return __fmaf_rn(tau, p, __fmul_rn(om, t)); }

inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// Optional per-launch timing (off by default).  When enabled through
// hhrl_prof_enable(1), every kernel launch site brackets its launch with a
// pair of HIP events recorded on the launch stream and notes the algorithmic
// work (flops, bytes) of that launch; hhrl_prof_report() aggregates by name.
bool prof_on();
bool prof_shapes();                          // profiling on and HHRL_PROF_SHAPES=1: product launches are reported per shape
const char* prof_intern(const char* name);   // a copy of `name` that lives as long as the process
struct ProfScope {
  // own_launch: the events are not recorded around the launch but handed to it (launch_timed below), so that
  // they carry the dispatch's own begin and end timestamps, the interval a kernel trace reports
  ProfScope(const char* name, hipStream_t st, double flops, double bytes, bool own_launch = false);
  ~ProfScope();
  hipEvent_t beg() const;
  hipEvent_t end() const;
  int slot;
  hipStream_t st;
  bool own;
};

// Launch of `kernel` that, while profiling is on, is stamped with its own start and stop events.
template <typename... KArgs, typename... Args>
inline void launch_timed(const ProfScope& ps, void (*kernel)(KArgs...), dim3 grid, dim3 block, unsigned lds, hipStream_t st,
                         Args... args) {
  if (ps.slot >= 0 && ps.own)
    hipExtLaunchKernelGGL(kernel, grid, block, lds, st, ps.beg(), ps.end(), 0, static_cast<KArgs>(args)...);
  else
    hipLaunchKernelGGL(kernel, grid, block, lds, st, static_cast<KArgs>(args)...);
}

// Internal GEMM front door (gemm.hip):  C[i][j] (op)= sum_k A(i,k) * B(k,j)
//   A(i,k) = A[i*a_si + k*a_sk],  B(k,j) = B[k*b_sk + j*b_sj]
// mode 0: C = act(acc + bias[j]); mode 1: C = act(C + acc + bias[j]);
// mode 2: atomicAdd(C, acc) with the reduction split over workgroups.
int gemm(const float* A, long a_si, long a_sk, const float* B, long b_sk, long b_sj, float* C, long ldc,
         int M, int N, int K, const float* bias, int act, int mode, hipStream_t st);

// The same product as a value, and several independent ones enqueued as ONE launch when all of them
// are small (batch-64 agent updates are bound by the number of launches, not by their work).
struct GemmArgs {
  const float* A;
  long a_si, a_sk;
  const float* B;
  long b_sk, b_sj;
  float* C;
  long ldc;
  const float* bias;
  int M, N, K;
  int k_chunk;
  int act, mode;
  int a_kcontig, b_kcontig;
  // Optional (small-product kernels only): operand A is used as A(i,k) * act'(Y(i,k)), Y addressed like A with its
  // own strides; rowsum[i] (and rowsum2[i]) += sum over this product's reduction range of that effective A(i,k).  This is how a
  // layer's backward products apply the activation gradient and form the bias gradient without a pass of their own.
  const float* Y;
  long y_si, y_sk;
  int y_act;
  float *rowsum, *rowsum2;       // rowsum2: a second destination of the same sums (two biases that share a gradient)
  // Optional (small-product kernels, modes 0/1): target-policy smoothing of the stored value into a second output,
  // y2[i*ldy2 + j] = clamp(C(i,j) + clamp(0.2*smooth_eps[i*N + j], -0.5, 0.5), -1, 1)
  const float* smooth_eps;
  float* y2;
  long ldy2;
  // The accumulated product is multiplied by alpha before bias, accumulation and activation (1 unless a caller sets it).
  float alpha;
};
constexpr int kGemmGroupMax = 16;
struct GemmGroup {
  GemmArgs g[kGemmGroupMax];
  int splits;
};
GemmArgs gemm_args(const float* A, long a_si, long a_sk, const float* B, long b_sk, long b_sj, float* C, long ldc,
                   int M, int N, int K, const float* bias, int act, int mode);
int gemm_group(const GemmArgs* gs, int n, hipStream_t st);
bool gemm_is_small(const GemmArgs& g);        // served by the small-product kernels (which honour Y / rowsum)

}  // namespace hhrl
