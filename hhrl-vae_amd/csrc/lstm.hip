// Bidirectional single-layer LSTM over a short window, last time index only,
// followed by Linear(2H -> H): forward and back-propagation through time.
//
// Layout choices: the [B, T, o] observation window is transposed once to
// time-major [T, B, o], so that every time step is a contiguous [B, .] block:
// the input projection of all T steps is ONE GEMM, each recurrent step is one
// [B,H]x[H,4H] GEMM accumulated onto its block, and the weight gradients of
// the whole window are ONE split-reduction GEMM each (dW_ih, dW_hh).
// Hidden and cell states are kept with a leading zero slot (slot t+1 = state
// after step t), so "state before step t" for all t is the same buffer
// without the last slot.
#include "common.h"

namespace hhrl {

static inline long align64(long n) { return (n + 63) & ~63L; }

struct LstmWs {
  float *x_tm, *gates, *hbuf, *cbuf, *gates_r, *c_r, *cat, *dgates, *dgates_r, *dcat, *dh, *dc, *dh2, *dc2;
  long total;
};

static LstmWs carve(float* ws, long B, long T, long o, long H) {
  LstmWs w;
  long off = 0;
  auto take = [&](long n) { float* p = ws ? ws + off : nullptr; off += align64(n); return p; };
  w.x_tm = take(T * B * o);
  w.gates = take(T * B * 4 * H);
  w.hbuf = take((T + 1) * B * H);
  w.cbuf = take((T + 1) * B * H);
  w.gates_r = take(B * 4 * H);
  w.c_r = take(B * H);
  w.cat = take(B * 2 * H);
  w.dgates = take(T * B * 4 * H);
  w.dgates_r = take(B * 4 * H);
  w.dcat = take(B * 2 * H);
  w.dh = take(B * H);
  w.dc = take(B * H);
  w.dh2 = take(B * H);
  w.dc2 = take(B * H);
  w.total = off;
  return w;
}

// [B, T*o] rows with leading dimension ld -> [T, B, o]
__global__ void to_time_major_kernel(const float* __restrict__ in, int ld, float* out, int B, int T, int o) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * T * o) return;
  const int c = idx % o;
  const long bt = idx / o;
  const int t = bt % T, b = bt / T;
  out[((long)t * B + b) * o + c] = in[(long)b * ld + t * o + c];
}

// gates: [B,4H] pre-activations (input projection + recurrent product), in
// place -> activated i,f,g,o.  bias2 (bias_hh) is added first.
__global__ void lstm_cell_forward_kernel(float* gates, const float* __restrict__ bias2,
                                         const float* __restrict__ c_prev, float* c_out, float* h_out, int ldh,
                                         float* h_out2, int ldh2, int B, int H) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * H) return;
  const int b = idx / H, j = idx % H;
  float* gr = gates + (long)b * 4 * H;
  const float i = 1.f / (1.f + expf(-(gr[j] + bias2[j])));
  const float f = 1.f / (1.f + expf(-(gr[H + j] + bias2[H + j])));
  const float g = tanhf(gr[2 * H + j] + bias2[2 * H + j]);
  const float o = 1.f / (1.f + expf(-(gr[3 * H + j] + bias2[3 * H + j])));
  gr[j] = i; gr[H + j] = f; gr[2 * H + j] = g; gr[3 * H + j] = o;
  const float cp = c_prev ? c_prev[idx] : 0.f;
  const float c = f * cp + i * g;
  const float h = o * tanhf(c);
  c_out[idx] = c;
  h_out[(long)b * ldh + j] = h;
  if (h_out2) h_out2[(long)b * ldh2 + j] = h;
}

// dh: gradient at this step's hidden output; dc: running cell gradient
// (read as the contribution from step t+1 unless first, overwritten with the
// gradient for step t-1).  dgates receives pre-activation gate gradients.
__global__ void lstm_cell_backward_kernel(const float* __restrict__ gates, const float* __restrict__ c_prev,
                                          const float* __restrict__ c, const float* __restrict__ dh, int lddh,
                                          float* dc, int dc_first, float* dgates, int B, int H) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * H) return;
  const int b = idx / H, j = idx % H;
  const float* gr = gates + (long)b * 4 * H;
  const float i = gr[j], f = gr[H + j], g = gr[2 * H + j], o = gr[3 * H + j];
  const float tc = tanhf(c[idx]);
  const float dhv = dh[(long)b * lddh + j];
  const float dcv = (dc_first ? 0.f : dc[idx]) + dhv * o * (1.f - tc * tc);
  const float cp = c_prev ? c_prev[idx] : 0.f;
  float* dg = dgates + (long)b * 4 * H;
  dg[j] = dcv * g * i * (1.f - i);
  dg[H + j] = dcv * cp * f * (1.f - f);
  dg[2 * H + j] = dcv * i * (1.f - g * g);
  dg[3 * H + j] = dhv * tc * o * (1.f - o);
  if (dc) dc[idx] = dcv * f;
}

// db1[col] += colsum(x), db2[col] += colsum(x)   (bias_ih and bias_hh share a gradient)
__global__ __launch_bounds__(256) void colsum2_kernel(const float* __restrict__ x, int ld, int M, int N, float* db1,
                                                      float* db2, int rows_per_block) {
  __shared__ float red[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + tx;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
  float s = 0.f;
  if (col < N)
    for (int r = r0 + ty; r < r1; r += 4) s += x[(long)r * ld + col];
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && col < N) {
    s = red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx];
    unsafeAtomicAdd(&db1[col], s);
    unsafeAtomicAdd(&db2[col], s);
  }
}


typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int drow16(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// One forward time step in ONE launch: gates = xproj_t + h_prev * W_hh^T + b_hh, then the cell.
// A workgroup owns 32 batch rows x 8 hidden units: a 32x32 tile of gate columns ordered
// (gate q, unit j) -> column 8*q + j.  Its four waves split the H-long reduction (short MFMA
// chains, many workgroups: the step is latency-bound), the partial tiles are summed through
// LDS in wave order, and each thread finishes one (row, unit) cell.  gates_t is overwritten
// with the ACTIVATED gates.
__global__ __launch_bounds__(256) void lstm_step_forward_kernel(float* gates_t, const float* __restrict__ w_hh,
                                                                const float* __restrict__ b_hh,
                                                                const float* __restrict__ h_prev,
                                                                const float* __restrict__ c_prev, float* c_out,
                                                                float* h_out, int ldh, float* h_out2, int ldh2, int B,
                                                                int H) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int u0 = blockIdx.x * 8, row0 = blockIdx.y * 32;
  const int ld = H + 1;
  float* Ah = lds;                       // [32][H+1]  h_prev rows
  float* Wq = lds + 32 * ld;             // [32][H+1]  W_hh rows (q*H + u0 + j) at image row 8*q + j
  float* part = Wq + 32 * ld;            // [4][32][33] partial gate tiles
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  if (h_prev != nullptr) {
    // 16-byte loads, all issued before the first LDS write (H % 32 == 0 on this path)
    const int hq = H >> 2;
    float4 v[4], wv[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {            // 32*hq float4 each, <= 4 per thread for H <= 128
      const int i4 = it * 256 + tid;
      const int rr = i4 / hq, c4 = i4 - rr * hq;
      const bool in = i4 < 32 * hq;
      v[it] = (in && row0 + rr < B) ? reinterpret_cast<const float4*>(h_prev + (long)(row0 + rr) * H)[c4]
                                    : make_float4(0.f, 0.f, 0.f, 0.f);
      const int q = rr >> 3, j = rr & 7;
      wv[it] = (in && u0 + j < H) ? reinterpret_cast<const float4*>(w_hh + ((long)q * H + u0 + j) * H)[c4]
                                  : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int i4 = it * 256 + tid;
      const int rr = i4 / hq, c4 = i4 - rr * hq;
      if (i4 < 32 * hq) {
        float* p = Ah + rr * ld + 4 * c4;
        p[0] = v[it].x; p[1] = v[it].y; p[2] = v[it].z; p[3] = v[it].w;
        float* q = Wq + rr * ld + 4 * c4;
        q[0] = wv[it].x; q[1] = wv[it].y; q[2] = wv[it].z; q[3] = wv[it].w;
      }
    }
    __syncthreads();
    const int kq = H >> 2;                       // this wave's quarter of the reduction
    const float* ap = Ah + r * ld + wave * kq + h;
    const float* bp = Wq + r * ld + wave * kq + h;
    for (int k = 0; k < kq; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[k], bp[k], acc, 0, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) part[(wave * 32 + drow16(i, h)) * 33 + r] = acc[i];
  __syncthreads();
  {
    const int rr = tid >> 3, j = tid & 7;
    const int b = row0 + rr, u = u0 + j;
    if (b < B && u < H) {
      float pre[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = 8 * q + j;
        pre[q] = (part[(0 * 32 + rr) * 33 + col] + part[(1 * 32 + rr) * 33 + col]) +
                 (part[(2 * 32 + rr) * 33 + col] + part[(3 * 32 + rr) * 33 + col]);
      }
      float* gr = gates_t + (long)b * 4 * H;
      const float gi = sigmoidf_(gr[u] + pre[0] + b_hh[u]);
      const float gf = sigmoidf_(gr[H + u] + pre[1] + b_hh[H + u]);
      const float gg = tanhf(gr[2 * H + u] + pre[2] + b_hh[2 * H + u]);
      const float go = sigmoidf_(gr[3 * H + u] + pre[3] + b_hh[3 * H + u]);
      gr[u] = gi; gr[H + u] = gf; gr[2 * H + u] = gg; gr[3 * H + u] = go;
      const float cp = c_prev ? c_prev[(long)b * H + u] : 0.f;
      const float c = gf * cp + gi * gg;
      const float hv = go * tanhf(c);
      c_out[(long)b * H + u] = c;
      h_out[(long)b * ldh + u] = hv;
      if (h_out2) h_out2[(long)b * ldh2 + u] = hv;
    }
  }
}

// One backward time step in ONE launch: dh_{t-1} = dgates_t * W_hh on the matrix cores, then,
// in the epilogue, the cell's backward for step t-1 on the same (row, unit) elements:
//   dc_{t-1} = dc + dh_{t-1} * o * (1 - tanh(c_{t-1})^2), the four gate gradients of step t-1,
//   and dc <- dc_{t-1} * f for the step before.
// A workgroup owns 32 batch rows x 32 hidden units; its four waves each reduce one gate's H
// columns of dgates_t (staged in LDS with the matching rows of W_hh), partial tiles are summed
// in LDS in wave order.  Every (row, unit) element is owned by exactly one thread of the grid,
// so dc is updated in place and dh never goes to memory.
__global__ __launch_bounds__(256) void lstm_step_backward_kernel(const float* __restrict__ dgates_t,
                                                                 const float* __restrict__ w_hh,
                                                                 const float* __restrict__ gates_p,   // activated gates of step t-1
                                                                 const float* __restrict__ c_pp,      // cell state before step t-1 (NULL = 0)
                                                                 const float* __restrict__ c_p,       // cell state after step t-1
                                                                 float* dc, float* dgates_p, int B, int H) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int u0 = blockIdx.x * 32, row0 = blockIdx.y * 32;
  const int G4 = 4 * H, lda = G4 + 1;
  float* Dg = lds;                        // [32][4H+1]  dgates_t rows
  float* Wt = Dg + 32 * lda;              // [4H][33]    W_hh[k][u0 + j]
  float* part = Wt + G4 * 33;             // [4][32][33]
  {
    // 16-byte loads in batches of 8 per thread, all issued before their LDS writes
    const int a4 = 32 * H;                // float4 count of the dgates tile (32 rows x 4H)
    const int hq = H;                     // float4 per row = 4H/4
    for (int base = 0; base < a4; base += 8 * 256) {
      float4 v[8];
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int i4 = base + it * 256 + tid;
        const int rr = i4 / hq, c4 = i4 - rr * hq;
        v[it] = (i4 < a4 && row0 + rr < B) ? reinterpret_cast<const float4*>(dgates_t + (long)(row0 + rr) * G4)[c4]
                                           : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int i4 = base + it * 256 + tid;
        const int rr = i4 / hq, c4 = i4 - rr * hq;
        if (i4 < a4) {
          float* p = Dg + rr * lda + 4 * c4;
          p[0] = v[it].x; p[1] = v[it].y; p[2] = v[it].z; p[3] = v[it].w;
        }
      }
    }
    const int n4 = G4 * 8;                // W_hh[k][u0 .. u0+31]: 8 float4 per row k
    for (int base = 0; base < n4; base += 8 * 256) {
      float4 wv[8];
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int i4 = base + it * 256 + tid;
        const int k = i4 >> 3, c4 = i4 & 7;
        wv[it] = (i4 < n4 && u0 + 4 * c4 + 3 < H) ? reinterpret_cast<const float4*>(w_hh + (long)k * H + u0)[c4]
                                                    : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int i4 = base + it * 256 + tid;
        const int k = i4 >> 3, c4 = i4 & 7;
        if (i4 < n4) {
          float* p = Wt + k * 33 + 4 * c4;
          p[0] = wv[it].x; p[1] = wv[it].y; p[2] = wv[it].z; p[3] = wv[it].w;
        }
      }
    }
  }
  __syncthreads();
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const float* ap = Dg + r * lda + wave * H + h;
  const float* bp = Wt + (wave * H + h) * 33 + r;
  for (int k = 0; k < H; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[k], bp[k * 33], acc, 0, 0, 0);
#pragma unroll
  for (int i = 0; i < 16; ++i) part[(wave * 32 + drow16(i, h)) * 33 + r] = acc[i];
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int idx = t * 256 + tid;
    const int rr = idx >> 5, j = idx & 31;
    const int b = row0 + rr, u = u0 + j;
    if (b >= B || u >= H) continue;
    const float dhv = (part[(0 * 32 + rr) * 33 + j] + part[(1 * 32 + rr) * 33 + j]) +
                      (part[(2 * 32 + rr) * 33 + j] + part[(3 * 32 + rr) * 33 + j]);
    const float* gr = gates_p + (long)b * G4;
    const float gi = gr[u], gf = gr[H + u], gg = gr[2 * H + u], go = gr[3 * H + u];
    const float tc = tanhf(c_p[(long)b * H + u]);
    const float dcv = dc[(long)b * H + u] + dhv * go * (1.f - tc * tc);
    const float cp = c_pp ? c_pp[(long)b * H + u] : 0.f;
    float* dg = dgates_p + (long)b * G4;
    dg[u] = dcv * gg * gi * (1.f - gi);
    dg[H + u] = dcv * cp * gf * (1.f - gf);
    dg[2 * H + u] = dcv * gi * (1.f - gg * gg);
    dg[3 * H + u] = dhv * tc * go * (1.f - go);
    dc[(long)b * H + u] = dcv * gf;
  }
}

static int set_step_attrs(int H);
static long step_fwd_lds_bytes(int H) { return 4L * (2L * 32 * (H + 1) + 4 * 32 * 33); }
static long step_bwd_lds_bytes(int H) { return 4L * (32L * (4 * H + 1) + 4L * H * 33 + 4 * 32 * 33); }
// The fused step kernels hold their tiles in LDS; beyond this hidden width the per-step
// GEMM + cell pair is used instead.
static int set_step_attrs(int H) {
  static int done_for = -1;
  if (done_for >= H) return HHRL_OK;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_step_forward_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)step_fwd_lds_bytes(H)) != hipSuccess ||
      hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_step_backward_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)step_bwd_lds_bytes(H)) != hipSuccess)
    return fail(HHRL_ERR_HIP, "lstm: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
  done_for = H;
  return HHRL_OK;
}
static bool fused_steps_fit(int H) {
  return H % 32 == 0 && H <= 128 && step_bwd_lds_bytes(H) <= 160 * 1024 && step_fwd_lds_bytes(H) <= 160 * 1024;
}

}  // namespace hhrl

using namespace hhrl;

extern "C" {

long hhrl_lstm_workspace_floats(int B, int T, int o, int H) {
  if (B <= 0 || T <= 0 || o <= 0 || H <= 0) return 0;
  return carve(nullptr, B, T, o, H).total;
}

int hhrl_lstm_forward(const float* obs, int ldobs, const float* const* params, float* out, int ldo, float* ws, int B,
                      int T, int o, int H, void* stream) {
  HHRL_REQUIRE(obs && params && out && ws, "lstm_forward: null pointer");
  HHRL_REQUIRE(B > 0 && T > 0 && o > 0 && H > 0 && ldo >= H && ldobs >= T * o,
               "lstm_forward: bad shape B=%d T=%d o=%d H=%d ldo=%d ldobs=%d", B, T, o, H, ldo, ldobs);
  for (int i = 0; i < 10; ++i) HHRL_REQUIRE(params[i], "lstm_forward: params[%d] is null", i);
  const float *w_ih = params[0], *w_hh = params[1], *b_ih = params[2], *b_hh = params[3];
  const float *w_ih_r = params[4], *b_ih_r = params[6], *b_hh_r = params[7];
  const float *fc_w = params[8], *fc_b = params[9];
  hipStream_t st = as_stream(stream);
  LstmWs w = carve(ws, B, T, o, H);
  const int G = 4 * H, BH = B * H;

  hipLaunchKernelGGL(to_time_major_kernel, dim3(ceil_div((long)B * T * o, 256)), dim3(256), 0, st, obs, ldobs, w.x_tm, B, T, o);
  HHRL_TRY(launch_status("to_time_major_kernel"));
  if (hipMemsetAsync(w.hbuf, 0, sizeof(float) * BH, st) != hipSuccess || hipMemsetAsync(w.cbuf, 0, sizeof(float) * BH, st) != hipSuccess)
    return fail(HHRL_ERR_HIP, "lstm_forward: hipMemsetAsync failed");
  // input projections: all T steps of the forward direction, last step of the reverse direction
  HHRL_TRY(gemm(w.x_tm, o, 1, w_ih, 1, o, w.gates, G, T * B, G, o, b_ih, HHRL_ACT_NONE, 0, st));
  HHRL_TRY(gemm(w.x_tm + (long)(T - 1) * B * o, o, 1, w_ih_r, 1, o, w.gates_r, G, B, G, o, b_ih_r, HHRL_ACT_NONE, 0, st));
  const bool fused = fused_steps_fit(H);
  if (fused) HHRL_TRY(set_step_attrs(H));
  for (int t = 0; t < T && fused; ++t) {
    float* gt = w.gates + (long)t * B * G;
    const bool last = t == T - 1;
    ProfScope ps("lstm_step_forward_kernel", st, t > 0 ? 2.0 * B * G * H : 0.0, 4.0 * (B * (2.0 * G + 4.0 * H) + (t > 0 ? (double)G * H : 0.0)));
    hipLaunchKernelGGL(lstm_step_forward_kernel, dim3(ceil_div(H, 8), ceil_div(B, 32)), dim3(256), step_fwd_lds_bytes(H), st,
                       gt, w_hh, b_hh, t > 0 ? (const float*)(w.hbuf + (long)t * BH) : (const float*)nullptr,
                       t > 0 ? (const float*)(w.cbuf + (long)t * BH) : (const float*)nullptr, w.cbuf + (long)(t + 1) * BH,
                       w.hbuf + (long)(t + 1) * BH, H, last ? w.cat : (float*)nullptr, 2 * H, B, H);
    HHRL_TRY(launch_status("lstm_step_forward_kernel"));
  }
  for (int t = 0; t < T && !fused; ++t) {
    float* gt = w.gates + (long)t * B * G;
    if (t > 0) HHRL_TRY(gemm(w.hbuf + (long)t * BH, H, 1, w_hh, 1, H, gt, G, B, G, H, nullptr, HHRL_ACT_NONE, 1, st));
    const bool last = t == T - 1;
    ProfScope ps("lstm_cell_forward_kernel", st, 0.0, 28.0 * BH);
    hipLaunchKernelGGL(lstm_cell_forward_kernel, dim3(ceil_div(BH, 256)), dim3(256), 0, st, gt, b_hh,
                       w.cbuf + (long)t * BH, w.cbuf + (long)(t + 1) * BH, w.hbuf + (long)(t + 1) * BH, H,
                       last ? w.cat : nullptr, 2 * H, B, H);
    HHRL_TRY(launch_status("lstm_cell_forward_kernel"));
  }
  hipLaunchKernelGGL(lstm_cell_forward_kernel, dim3(ceil_div(BH, 256)), dim3(256), 0, st, w.gates_r, b_hh_r,
                     (const float*)nullptr, w.c_r, w.cat + H, 2 * H, (float*)nullptr, 0, B, H);
  HHRL_TRY(launch_status("lstm_cell_forward_kernel(reverse)"));
  return gemm(w.cat, 2 * H, 1, fc_w, 1, 2 * H, out, ldo, B, H, 2 * H, fc_b, HHRL_ACT_NONE, 0, st);
}

int hhrl_lstm_backward(const float* const* params, const float* d_out, int lddo, float* const* grads, float* ws,
                       int B, int T, int o, int H, void* stream) {
  HHRL_REQUIRE(params && d_out && grads && ws, "lstm_backward: null pointer");
  HHRL_REQUIRE(B > 0 && T > 0 && o > 0 && H > 0 && lddo >= H, "lstm_backward: bad shape B=%d T=%d o=%d H=%d lddo=%d", B, T, o, H, lddo);
  for (int i = 0; i < 10; ++i) HHRL_REQUIRE(params[i] && grads[i], "lstm_backward: params/grads[%d] is null", i);
  const float *w_hh = params[1], *fc_w = params[8];
  float *g_w_ih = grads[0], *g_w_hh = grads[1], *g_b_ih = grads[2], *g_b_hh = grads[3];
  float *g_w_ih_r = grads[4], *g_b_ih_r = grads[6], *g_b_hh_r = grads[7];
  float *g_fc_w = grads[8], *g_fc_b = grads[9];
  hipStream_t st = as_stream(stream);
  LstmWs w = carve(ws, B, T, o, H);
  const int G = 4 * H, BH = B * H;
  const dim3 cell_grid(ceil_div(BH, 256)), blk(256);

  // fc: out = cat * fc_w^T + fc_b
  HHRL_TRY(hhrl_linear_backward(w.cat, 2 * H, fc_w, nullptr, 0, const_cast<float*>(d_out), lddo, w.dcat, 2 * H, 0,
                                g_fc_w, g_fc_b, B, 2 * H, H, HHRL_ACT_NONE, stream));
  // reverse direction: a single cell step from a zero state on x[T-1]
  hipLaunchKernelGGL(lstm_cell_backward_kernel, cell_grid, blk, 0, st, w.gates_r, (const float*)nullptr, w.c_r,
                     w.dcat + H, 2 * H, (float*)nullptr, 1, w.dgates_r, B, H);
  HHRL_TRY(launch_status("lstm_cell_backward_kernel(reverse)"));
  HHRL_TRY(gemm(w.dgates_r, 1, G, w.x_tm + (long)(T - 1) * B * o, o, 1, g_w_ih_r, o, G, o, B, nullptr, HHRL_ACT_NONE, 2, st));
  hipLaunchKernelGGL(colsum2_kernel, dim3(ceil_div(G, 64), ceil_div(B, 64)), blk, 0, st, w.dgates_r, G, B, G, g_b_ih_r, g_b_hh_r, 64);
  HHRL_TRY(launch_status("colsum2_kernel(reverse)"));
  // forward direction: T steps back through time
  const float* dh = w.dcat;
  int lddh = 2 * H;
  const bool fused_bwd = fused_steps_fit(H);
  if (fused_bwd) {
    HHRL_TRY(set_step_attrs(H));
    // last step: its hidden gradient comes from the fc layer; cell backward on its own
    {
      ProfScope ps("lstm_cell_backward_kernel", st, 0.0, 48.0 * BH);
      hipLaunchKernelGGL(lstm_cell_backward_kernel, cell_grid, blk, 0, st, w.gates + (long)(T - 1) * B * G,
                         w.cbuf + (long)(T - 1) * BH, w.cbuf + (long)T * BH, dh, lddh, w.dc, 1,
                         w.dgates + (long)(T - 1) * B * G, B, H);
      HHRL_TRY(launch_status("lstm_cell_backward_kernel"));
    }
    // steps T-1 .. 1: dh_{t-1} = dgates_t W_hh fused with the cell backward of step t-1
    for (int t = T - 1; t >= 1; --t) {
      ProfScope ps("lstm_step_backward_kernel", st, 2.0 * B * G * H, 4.0 * (B * (3.0 * G + 4.0 * H) + (double)G * H));
      hipLaunchKernelGGL(lstm_step_backward_kernel, dim3(ceil_div(H, 32), ceil_div(B, 32)), dim3(256), step_bwd_lds_bytes(H), st,
                         (const float*)(w.dgates + (long)t * B * G), w_hh, (const float*)(w.gates + (long)(t - 1) * B * G),
                         (const float*)(w.cbuf + (long)(t - 1) * BH), (const float*)(w.cbuf + (long)t * BH), w.dc,
                         w.dgates + (long)(t - 1) * B * G, B, H);
      HHRL_TRY(launch_status("lstm_step_backward_kernel"));
    }
  }
  for (int t = T - 1; t >= 0 && !fused_bwd; --t) {
    float* dgt = w.dgates + (long)t * B * G;
    ProfScope ps("lstm_cell_backward_kernel", st, 0.0, 48.0 * BH);
    hipLaunchKernelGGL(lstm_cell_backward_kernel, cell_grid, blk, 0, st, w.gates + (long)t * B * G,
                       w.cbuf + (long)t * BH, w.cbuf + (long)(t + 1) * BH, dh, lddh, w.dc, t == T - 1 ? 1 : 0, dgt, B, H);
    HHRL_TRY(launch_status("lstm_cell_backward_kernel"));
    if (t > 0) {
      HHRL_TRY(gemm(dgt, G, 1, w_hh, H, 1, w.dh, H, B, H, G, nullptr, HHRL_ACT_NONE, 0, st));
      dh = w.dh;
      lddh = H;
    }
  }
  // weight gradients of the whole window: rows (t,b) of dgates against x_tm / previous hidden state
  HHRL_TRY(gemm(w.dgates, 1, G, w.x_tm, o, 1, g_w_ih, o, G, o, T * B, nullptr, HHRL_ACT_NONE, 2, st));
  HHRL_TRY(gemm(w.dgates, 1, G, w.hbuf, H, 1, g_w_hh, H, G, H, T * B, nullptr, HHRL_ACT_NONE, 2, st));
  hipLaunchKernelGGL(colsum2_kernel, dim3(ceil_div(G, 64), ceil_div(T * B, 64)), blk, 0, st, w.dgates, G, T * B, G, g_b_ih, g_b_hh, 64);
  return launch_status("colsum2_kernel");
}

}  // extern "C"
