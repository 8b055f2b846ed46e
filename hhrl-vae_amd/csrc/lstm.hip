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
  float *x_tm, *gates, *hbuf, *cbuf, *gates_r, *c_r, *cat, *dgates, *dgates_r, *dcat, *dh, *dc;
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
  for (int t = 0; t < T; ++t) {
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
  for (int t = T - 1; t >= 0; --t) {
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
