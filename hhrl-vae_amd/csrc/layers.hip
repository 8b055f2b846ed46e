// Dense layer forward/backward and the strided data-movement kernels.
#include "common.h"

#include <map>
#include <mutex>
#include <utility>

namespace hhrl {

static thread_local char g_err[512];
char* err_buf() { return g_err; }
int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int ensure_dyn_lds(const void* fn, int bytes, const char* who) {
  static std::mutex mu;
  static std::map<std::pair<int, const void*>, int> done;     // (device, kernel) -> bytes granted
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return fail(HHRL_ERR_HIP, "%s: hipGetDevice failed", who);
  std::lock_guard<std::mutex> lock(mu);
  auto it = done.find({dev, fn});
  if (it != done.end() && it->second >= bytes) return HHRL_OK;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess)
    return fail(HHRL_ERR_HIP, "%s: hipFuncSetAttribute(MaxDynamicSharedMemorySize, %d) failed on device %d", who, bytes, dev);
  done[{dev, fn}] = bytes;
  return HHRL_OK;
}

// dy <- dy * act'(y) in place, and db[col] += sum over rows.  A workgroup
// covers CW columns x (256/CW) rows per pass over a slab of rows; CW is the
// power of two >= min(N,64) so narrow layers still use every lane.
__global__ __launch_bounds__(256) void act_grad_colsum_kernel(const float* __restrict__ y, int ldy, float* dy, int lddy,
                                                              float* db, int M, int N, int act, int cw,
                                                              int rows_per_block) {
  __shared__ float red[256];
  const int t = threadIdx.x;
  const int tx = t % cw, ty = t / cw, rp = 256 / cw;
  const int col = blockIdx.x * cw + tx;
  const int r0 = blockIdx.y * rows_per_block;
  const int r1 = min(M, r0 + rows_per_block);
  float s = 0.f;
  if (col < N) {
    // rows_per_block = 16 * rp: every thread owns 16 rows, all of them requested before the first use
    float gv[16], yv[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int r = r0 + ty + u * rp;
      gv[u] = r < r1 ? dy[(long)r * lddy + col] : 0.f;
      yv[u] = (act != HHRL_ACT_NONE && r < r1) ? y[(long)r * ldy + col] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int r = r0 + ty + u * rp;
      if (act != HHRL_ACT_NONE && r < r1) {
        gv[u] *= act_grad_from_output(yv[u], act);
        dy[(long)r * lddy + col] = gv[u];
      }
      s += gv[u];
    }
  }
  if (db == nullptr) return;
  red[t] = s;
  __syncthreads();
  for (int h = rp >> 1; h > 0; h >>= 1) {
    if (ty < h) red[t] += red[t + h * cw];
    __syncthreads();
  }
  if (ty == 0 && col < N) unsafeAtomicAdd(&db[col], red[t]);
}

// The same for several layers in one launch (blockIdx.z = layer); rows M shared.
struct ActGradInst { const float* y; float *dy, *db; int ldy, lddy, N, act, cw; };
struct ActGradGroup { ActGradInst in[kGemmGroupMax]; };
__global__ __launch_bounds__(256) void act_grad_colsum_group_kernel(ActGradGroup grp, int M, int rows_per_unit) {
  __shared__ float red[256];
  const ActGradInst& g = grp.in[blockIdx.z];
  const int t = threadIdx.x;
  const int cw = g.cw, tx = t % cw, ty = t / cw, rp = 256 / cw;
  const int col = blockIdx.x * cw + tx;
  const int rows_per_block = rp * rows_per_unit;
  const int r0 = blockIdx.y * rows_per_block;
  const int r1 = min(M, r0 + rows_per_block);
  if (blockIdx.x * cw >= g.N || r0 >= M) return;
  float s = 0.f;
  if (col < g.N) {
    // rows_per_unit rows per thread (16), all of them requested before the first use
    float gv[16], yv[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int r = r0 + ty + u * rp;
      gv[u] = (u < rows_per_unit && r < r1) ? g.dy[(long)r * g.lddy + col] : 0.f;
      yv[u] = (g.act != HHRL_ACT_NONE && u < rows_per_unit && r < r1) ? g.y[(long)r * g.ldy + col] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int r = r0 + ty + u * rp;
      if (g.act != HHRL_ACT_NONE && u < rows_per_unit && r < r1) {
        gv[u] *= act_grad_from_output(yv[u], g.act);
        g.dy[(long)r * g.lddy + col] = gv[u];
      }
      s += gv[u];
    }
  }
  if (g.db == nullptr) return;
  red[t] = s;
  __syncthreads();
  for (int h = rp >> 1; h > 0; h >>= 1) {
    if (ty < h) red[t] += red[t + h * cw];
    __syncthreads();
  }
  if (ty == 0 && col < g.N) unsafeAtomicAdd(&g.db[col], red[t]);
}

__global__ void sum_pool_kernel(const float* __restrict__ in, int ldi, float* out, int ldo, int B, int N, int C) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * C) return;
  const int b = idx / C, c = idx % C;
  const float* p = in + (long)b * N * ldi + c;
  float s = 0.f;
  for (int n = 0; n < N; ++n) s += p[(long)n * ldi];
  out[(long)b * ldo + c] = s;
}

__global__ void broadcast_rows_kernel(const float* __restrict__ in, int ldi, float* out, int ldo, int B, int N, int C) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * N * C) return;
  const int c = idx % C;
  const long row = idx / C;
  out[row * ldo + c] = in[(row / N) * ldi + c];
}

__global__ void copy2d_kernel(const float* __restrict__ in, int ldi, float* out, int ldo, int M, int C) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)M * C) return;
  const int c = idx % C;
  const long r = idx / C;
  out[r * ldo + c] = in[r * ldi + c];
}

// out[r][0..C) = in[idx[r]][0..C): the rows of a sampled minibatch out of a device-resident replay buffer
__global__ void gather_rows_kernel(const float* __restrict__ in, long ldi, const int* __restrict__ idx, int n_in, float* out,
                                   int ldo, int M, int C) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)M * C) return;
  const int c = e % C;
  const long r = e / C;
  const int src = min(max(idx[r], 0), n_in - 1);          // an index outside the buffer cannot leave it
  out[r * ldo + c] = in[(long)src * ldi + c];
}

struct CopyInst { const float* in; float* out; int ldi, ldo, C; };
struct CopyGroup { CopyInst in[HHRL_COPY_GROUP_MAX]; };
__global__ void copy2d_group_kernel(CopyGroup grp, int M) {
  const CopyInst& g = grp.in[blockIdx.y];
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g.in == nullptr) {                    // a zero fill of C floats: four per thread
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (4 * idx + e < g.C) g.out[4 * idx + e] = 0.f;
    return;
  }
  if (idx >= (long)M * g.C) return;
  const int c = idx % g.C;
  const long r = idx / g.C;
  g.out[r * g.ldo + c] = g.in[r * g.ldi + c];
}

// Standard normals from a counter-based generator: Philox4x32-10 (Salmon et al., SC'11) keyed by the seed, counter =
// index of the output quadruple, four uniforms per counter turned into four normals by two Box-Muller pairs.  Element
// i of the stream depends on (seed, i) only: a fill of n values at offset o equals the same range of one long fill,
// whatever the grid, so a pool can be refilled piecewise and a run can be repeated from its seed.
__device__ __forceinline__ void philox_round(unsigned (&c)[4], unsigned k0, unsigned k1) {
  const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
  const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1;
  c[1] = (unsigned)p1;
  c[3] = (unsigned)p0;
  c[0] = n0;
  c[2] = n2;
}
__device__ __forceinline__ void philox4x32_10(unsigned long long ctr, unsigned long long seed, unsigned (&out)[4]) {
  unsigned c[4] = {(unsigned)ctr, (unsigned)(ctr >> 32), 0u, 0u};
  unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) out[i] = c[i];
}
// (0, 1]: never zero, so the logarithm below is finite
__device__ __forceinline__ float u01(unsigned x) { return ((float)(x >> 8) + 1.0f) * (1.0f / 16777216.0f); }
// the four normals of stream quadruple q: two Box-Muller pairs from one Philox block
__device__ __forceinline__ void normals4(unsigned long long q, unsigned long long seed, float (&z)[4]) {
  unsigned r[4];
  philox4x32_10(q, seed, r);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float rad = sqrtf(-2.0f * logf(u01(r[2 * h])));
    float sn, cs;
    sincospif(2.0f * u01(r[2 * h + 1]), &sn, &cs);
    z[2 * h] = rad * cs;
    z[2 * h + 1] = rad * sn;
  }
}

// copy2d_group_kernel whose copies may read row idx[r] of their source instead of row r: the minibatch gather out of
// a device-resident replay buffer and every copy of replay columns into a network input as ONE launch (the copies then
// read the replay buffer directly instead of waiting for the gathered rows).  A workgroup takes a tile of consecutive
// rows of one block (blockIdx.y) and reads the tile's row numbers ONCE, into LDS: they may lie in pinned host memory
// (the update's staging block, read in place), where every read is a trip over the host link.  gather == 2: a flat
// copy of C floats (the staging block itself on its way into device memory), dealt over the workgroups like a zero fill.
struct GCopyInst { const float* in; float* out; long ldi; int ldo, C, gather; };
struct GCopyGroup { GCopyInst in[HHRL_COPY_GROUP_MAX]; };
constexpr int kGcElems = 1024;      // elements per workgroup and row-copy tile, about
constexpr int kGcMaxRows = 64;
__host__ __device__ inline int gc_tile_rows(int C) {
  const int r = kGcElems / C;
  return r < 1 ? 1 : (r > kGcMaxRows ? kGcMaxRows : r);
}
__global__ __launch_bounds__(256) void gather_copy_group_kernel(GCopyGroup grp, int M, const int* __restrict__ idx, int n_in) {
  __shared__ int sidx[kGcMaxRows];
  const GCopyInst& g = grp.in[blockIdx.y];
  const int tid = threadIdx.x, C = g.C;
  if (g.in == nullptr || g.gather == 2 || g.gather == 3) {   // flat: C floats, an equal run of whole 16-byte groups per workgroup
    const long quads = ((long)C + 3) / 4, per = (quads + gridDim.x - 1) / gridDim.x;
    const long lo = 4 * per * blockIdx.x, hi = lo + 4 * per < C ? lo + 4 * per : C;
    if (g.in == nullptr) {
      for (long i = lo + tid; i < hi; i += 256) g.out[i] = 0.f;
    } else if (g.gather == 2) {
      for (long i = lo + tid; i < hi; i += 256) g.out[i] = g.in[i];
    } else {
      // standard normals: out[i] = element (offset + i) of the stream of `seed`; the two 64-bit numbers are read from
      // memory (a replayed launch draws new values when the caller has advanced them); offset is a multiple of four
      const unsigned long long* so = reinterpret_cast<const unsigned long long*>(g.in);
      const unsigned long long seed = so[0], q0 = so[1] >> 2;
      for (long qd = per * blockIdx.x + tid; qd < per * (blockIdx.x + 1) && qd < quads; qd += 256) {
        float z[4];
        normals4(q0 + (unsigned long long)qd, seed, z);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (4 * qd + e < C) g.out[4 * qd + e] = z[e];
      }
    }
    return;
  }
  const int R = gc_tile_rows(C), row0 = blockIdx.x * R;
  if (g.gather == 4) {
    // rows stored since the last update on their way into the device-resident replay buffer: in = {int count;
    // int position[63]} followed, 64 floats on, by the rows [count][C]; row k goes to out[position[k]][0..C).  Count and
    // positions are read from memory, so a replayed launch scatters what the caller has put there (possibly nothing).
    const int* hdr = reinterpret_cast<const int*>(g.in);
    const int n_new = min(hdr[0], 63);
    if (row0 >= n_new) return;
    const float* rows = g.in + 64;
    const int n = min(R, n_new - row0) * C;
    for (int e = tid; e < n; e += 256) {
      const int rr = e / C, c = e - rr * C;
      g.out[(long)min(max(hdr[1 + row0 + rr], 0), n_in - 1) * g.ldo + c] = rows[(long)(row0 + rr) * C + c];
    }
    return;
  }
  if (row0 >= M) return;
  if (g.gather) {
    if (tid < R && row0 + tid < M) sidx[tid] = min(max(idx[row0 + tid], 0), n_in - 1);   // an index outside the buffer cannot leave it
    __syncthreads();
  }
  const int n = min(R, M - row0) * C;
  for (int base = 0; base < n; base += 4 * 256) {      // four loads in flight per thread
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = min(base + u * 256 + tid, n - 1), rr = e / C, c = e - rr * C;
      const long src = g.gather ? (long)sidx[rr] : (long)(row0 + rr);
      v[u] = g.in[src * g.ldi + c];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = base + u * 256 + tid, rr = e / C, c = e - rr * C;
      if (e < n) g.out[(long)(row0 + rr) * g.ldo + c] = v[u];
    }
  }
}

__global__ void fill_kernel(float* p, long n, float v) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < n) p[idx] = v;
}

__global__ __launch_bounds__(256) void randn_fill_kernel(float* __restrict__ out, long n, unsigned long long seed, unsigned long long offset) {
  // thread t produces elements 4 t' .. 4 t' + 3 of the STREAM (t' counted from the stream's origin) that fall into
  // [offset, offset + n)
  const unsigned long long q0 = offset >> 2;
  const unsigned long long q = q0 + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  float z[4];
  normals4(q, seed, z);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const unsigned long long i = 4 * q + e;
    if (i >= offset && i - offset < (unsigned long long)n) out[i - offset] = z[e];
  }
}

static int valid_act(int act) { return act >= HHRL_ACT_NONE && act <= HHRL_ACT_SIGMOID; }

}  // namespace hhrl

using namespace hhrl;

extern "C" {

int hhrl_version(void) { return HHRL_ABI_VERSION; }
const char* hhrl_last_error(void) { return err_buf(); }

int hhrl_linear_forward(const float* x, int ldx, const float* w, const float* b, float* y, int ldy, int M, int K,
                        int N, int act, int accumulate, void* stream) {
  HHRL_REQUIRE(x && w && y, "linear_forward: null pointer");
  HHRL_REQUIRE(M > 0 && K > 0 && N > 0, "linear_forward: non-positive size M=%d K=%d N=%d", M, K, N);
  HHRL_REQUIRE(ldx >= K && ldy >= N, "linear_forward: leading dimension too small (ldx=%d K=%d ldy=%d N=%d)", ldx, K, ldy, N);
  HHRL_REQUIRE(valid_act(act), "linear_forward: bad activation %d", act);
  return gemm(x, ldx, 1, w, 1, K, y, ldy, M, N, K, b, act, accumulate ? 1 : 0, as_stream(stream));
}

int hhrl_linear_backward(const float* x, int ldx, const float* w, const float* y, int ldy, float* dy, int lddy,
                         float* dx, int lddx, int dx_accumulate, float* dw, float* db, int M, int K, int N, int act,
                         void* stream) {
  HHRL_REQUIRE(dy && w, "linear_backward: null pointer");
  HHRL_REQUIRE(M > 0 && K > 0 && N > 0, "linear_backward: non-positive size M=%d K=%d N=%d", M, K, N);
  HHRL_REQUIRE(valid_act(act), "linear_backward: bad activation %d", act);
  HHRL_REQUIRE(act == HHRL_ACT_NONE || (y && ldy >= N), "linear_backward: activation gradient needs y");
  HHRL_REQUIRE(lddy >= N, "linear_backward: lddy=%d < N=%d", lddy, N);
  HHRL_REQUIRE(!dx || lddx >= K, "linear_backward: lddx=%d < K=%d", lddx, K);
  HHRL_REQUIRE(!dw || (x && ldx >= K), "linear_backward: dW needs x with ldx >= K");
  hipStream_t st = as_stream(stream);
  // input and weight gradient read the same dy and write different buffers: one launch for both
  GemmArgs g[2];
  int m = 0;
  if (dx) g[m++] = gemm_args(dy, lddy, 1, w, K, 1, dx, lddx, M, K, N, nullptr, HHRL_ACT_NONE, dx_accumulate ? 1 : 0);
  if (dw) g[m++] = gemm_args(dy, 1, lddy, x, ldx, 1, dw, K, N, K, M, nullptr, HHRL_ACT_NONE, 2);
  // When every product runs in the small-product kernels, the activation gradient is applied while they load dy
  // and the bias gradient is the row sum of the weight-gradient product's operand: no pass of its own, dy untouched.
  bool fuse = (act != HHRL_ACT_NONE || db) && (dw || !db) && m > 0;
  for (int i = 0; i < m; ++i) fuse = fuse && gemm_is_small(g[i]);
  if (fuse) {
    for (int i = 0; i < m; ++i) {
      if (act != HHRL_ACT_NONE) {
        g[i].Y = y;
        g[i].y_act = act;
        if (g[i].mode == 2) { g[i].y_si = 1; g[i].y_sk = ldy; } else { g[i].y_si = ldy; g[i].y_sk = 1; }
      }
      if (g[i].mode == 2) g[i].rowsum = db;
    }
    return gemm_group(g, m, st);
  }
  if (act != HHRL_ACT_NONE || db) {
    int cw = 64;
    while (cw > 1 && cw / 2 >= N) cw >>= 1;
    const int rp = 256 / cw;
    int rows_per_block = rp * 16;
    dim3 grid(ceil_div(N, cw), ceil_div(M, rows_per_block));
    ProfScope ps("act_grad_colsum_kernel", st, 2.0 * M * N, 4.0 * M * N * (act != HHRL_ACT_NONE ? 3 : 1));
    hipLaunchKernelGGL(act_grad_colsum_kernel, grid, dim3(256), 0, st, y, ldy, dy, lddy, db, M, N, act, cw,
                       rows_per_block);
    HHRL_TRY(launch_status("act_grad_colsum_kernel"));
  }
  return gemm_group(g, m, st);
}

int hhrl_linear_backward_row(const float* w, const float* y, int ldy, const float* dy_row, float alpha, float* dx, int lddx,
                             int dx_accumulate, int M, int K, int N, int act, void* stream) {
  HHRL_REQUIRE(w && dy_row && dx, "linear_backward_row: null pointer");
  HHRL_REQUIRE(M > 0 && K > 0 && N > 0, "linear_backward_row: non-positive size M=%d K=%d N=%d", M, K, N);
  HHRL_REQUIRE(valid_act(act), "linear_backward_row: bad activation %d", act);
  HHRL_REQUIRE(act == HHRL_ACT_NONE || (y && ldy >= N), "linear_backward_row: activation gradient needs y");
  HHRL_REQUIRE(lddx >= K, "linear_backward_row: lddx=%d < K=%d", lddx, K);
  // operand A(i, n) = dy_row[n] for every row i (row stride 0), masked by the activation gradient while it is loaded
  GemmArgs g = gemm_args(dy_row, 0, 1, w, K, 1, dx, lddx, M, K, N, nullptr, HHRL_ACT_NONE, dx_accumulate ? 1 : 0);
  HHRL_REQUIRE(gemm_is_small(g), "linear_backward_row: M=%d K=%d is beyond the small-product kernels", M, K);
  if (act != HHRL_ACT_NONE) {
    g.Y = y; g.y_si = ldy; g.y_sk = 1; g.y_act = act;
  }
  g.alpha = alpha;
  return gemm_group(&g, 1, as_stream(stream));
}

int hhrl_linear_backward_cols(const float* w, int K, const float* y, int ldy, const float* dy, int lddy, float* dx, int lddx,
                              int dx_accumulate, int c0, int nc, int M, int N, int act, void* stream) {
  HHRL_REQUIRE(w && dy && dx, "linear_backward_cols: null pointer");
  HHRL_REQUIRE(M > 0 && K > 0 && N > 0, "linear_backward_cols: non-positive size M=%d K=%d N=%d", M, K, N);
  HHRL_REQUIRE(c0 >= 0 && nc > 0 && c0 + nc <= K, "linear_backward_cols: columns [%d, %d) of %d", c0, c0 + nc, K);
  HHRL_REQUIRE(valid_act(act), "linear_backward_cols: bad activation %d", act);
  HHRL_REQUIRE(act == HHRL_ACT_NONE || (y && ldy >= N), "linear_backward_cols: activation gradient needs y");
  HHRL_REQUIRE(lddy >= N && lddx >= K, "linear_backward_cols: lddy=%d < N=%d or lddx=%d < K=%d", lddy, N, lddx, K);
  GemmArgs g = gemm_args(dy, lddy, 1, w + c0, K, 1, dx + c0, lddx, M, nc, N, nullptr, HHRL_ACT_NONE, dx_accumulate ? 1 : 0);
  HHRL_REQUIRE(gemm_is_small(g), "linear_backward_cols: M=%d is beyond the small-product kernels", M);
  if (act != HHRL_ACT_NONE) {
    g.Y = y; g.y_si = ldy; g.y_sk = 1; g.y_act = act;
  }
  return gemm_group(&g, 1, as_stream(stream));
}

int hhrl_linear_forward_group(const hhrl_linear_fwd_desc* d, int n, int M, void* stream) {
  HHRL_REQUIRE(d && n >= 1 && n <= kGemmGroupMax, "linear_forward_group: %d layers (1..%d supported)", n, kGemmGroupMax);
  HHRL_REQUIRE(M > 0, "linear_forward_group: M=%d", M);
  GemmArgs g[kGemmGroupMax];
  for (int i = 0; i < n; ++i) {
    const hhrl_linear_fwd_desc& l = d[i];
    HHRL_REQUIRE(l.x && l.w && l.y, "linear_forward_group: layer %d: null pointer", i);
    HHRL_REQUIRE(l.K > 0 && l.N > 0 && l.ldx >= l.K && l.ldy >= l.N, "linear_forward_group: layer %d: bad shape K=%d N=%d ldx=%d ldy=%d",
                 i, l.K, l.N, l.ldx, l.ldy);
    HHRL_REQUIRE(valid_act(l.act), "linear_forward_group: layer %d: bad activation %d", i, l.act);
    g[i] = gemm_args(l.x, l.ldx, 1, l.w, 1, l.K, l.y, l.ldy, M, l.N, l.K, l.b, l.act, l.accumulate ? 1 : 0);
    HHRL_REQUIRE(!l.smooth_eps || (l.y2 && l.ldy2 >= l.N), "linear_forward_group: layer %d: smoothing needs y2 with ldy2 >= N", i);
    if (l.smooth_eps && gemm_is_small(g[i])) {
      g[i].smooth_eps = l.smooth_eps;
      g[i].y2 = l.y2;
      g[i].ldy2 = l.ldy2;
    }
  }
  HHRL_TRY(gemm_group(g, n, as_stream(stream)));
  for (int i = 0; i < n; ++i)        // layers too large for the small-product kernels: smoothing as a launch of its own
    if (d[i].smooth_eps && !g[i].smooth_eps)
      HHRL_TRY(hhrl_td3_smooth_action(d[i].y, d[i].ldy, d[i].smooth_eps, d[i].y2, d[i].ldy2, M, d[i].N, stream));
  return HHRL_OK;
}

int hhrl_linear_backward_group(const hhrl_linear_bwd_desc* d, int n, int M, void* stream) {
  HHRL_REQUIRE(d && n >= 1 && n <= kGemmGroupMax, "linear_backward_group: %d layers (1..%d supported)", n, kGemmGroupMax);
  HHRL_REQUIRE(M > 0, "linear_backward_group: M=%d", M);
  hipStream_t st = as_stream(stream);
  ActGradGroup ag;
  int n_ag = 0, max_n = 0, min_cw = 64;
  // As in hhrl_linear_backward: with small products only, activation gradient and bias gradient ride in the products.
  bool fuse = n <= kGemmGroupMax;
  for (int i = 0; i < n && fuse; ++i) {
    const hhrl_linear_bwd_desc& l = d[i];
    if (l.db && !l.dw) fuse = false;
    if (l.dx && !gemm_is_small(gemm_args(l.dy, l.lddy, 1, l.w, l.K, 1, l.dx, l.lddx, M, l.K, l.N, nullptr, HHRL_ACT_NONE, 0))) fuse = false;
    if (l.dw && !gemm_is_small(gemm_args(l.dy, 1, l.lddy, l.x, l.ldx, 1, l.dw, l.K, l.N, l.K, M, nullptr, HHRL_ACT_NONE, 2))) fuse = false;
  }
  for (int i = 0; i < n; ++i) {
    const hhrl_linear_bwd_desc& l = d[i];
    HHRL_REQUIRE(l.dy && l.w, "linear_backward_group: layer %d: null pointer", i);
    HHRL_REQUIRE(l.K > 0 && l.N > 0 && l.lddy >= l.N, "linear_backward_group: layer %d: bad shape K=%d N=%d lddy=%d", i, l.K, l.N, l.lddy);
    HHRL_REQUIRE(valid_act(l.act), "linear_backward_group: layer %d: bad activation %d", i, l.act);
    HHRL_REQUIRE(l.act == HHRL_ACT_NONE || (l.y && l.ldy >= l.N), "linear_backward_group: layer %d: activation gradient needs y", i);
    HHRL_REQUIRE(!l.dx || l.lddx >= l.K, "linear_backward_group: layer %d: lddx=%d < K=%d", i, l.lddx, l.K);
    HHRL_REQUIRE(!l.dw || (l.x && l.ldx >= l.K), "linear_backward_group: layer %d: dW needs x with ldx >= K", i);
    if (!fuse && (l.act != HHRL_ACT_NONE || l.db)) {
      int cw = 64;
      while (cw > 1 && cw / 2 >= l.N) cw >>= 1;
      ag.in[n_ag++] = ActGradInst{l.y, l.dy, l.db, l.ldy, l.lddy, l.N, l.act, cw};
      max_n = l.N > max_n ? l.N : max_n;
      min_cw = cw < min_cw ? cw : min_cw;
    }
  }
  if (n_ag > 0) {
    // every workgroup covers (256 / cw) * 16 rows of its layer; the grid is sized for the narrowest column block
    const int rows_min = (256 / 64) * 16;
    dim3 grid(ceil_div(max_n, min_cw), ceil_div(M, rows_min), n_ag);
    ProfScope ps("act_grad_colsum_group_kernel", st, 0.0, 0.0);
    hipLaunchKernelGGL(act_grad_colsum_group_kernel, grid, dim3(256), 0, st, ag, M, 16);
    HHRL_TRY(launch_status("act_grad_colsum_group_kernel"));
  }
  // Input gradients: layers that overwrite their dx first, then the ones that accumulate (they may share
  // a dx).  The weight gradients read the same dy and write their own buffers, so they ride along with
  // the first of those launches.
  GemmArgs g[2 * kGemmGroupMax];
  auto flush = [&](int m) -> int {
    for (int at = 0; at < m; at += kGemmGroupMax)
      HHRL_TRY(gemm_group(g + at, m - at < kGemmGroupMax ? m - at : kGemmGroupMax, st));
    return HHRL_OK;
  };
  bool dw_done = false;
  for (int pass = 0; pass < 2; ++pass) {
    int m = 0;
    for (int i = 0; i < n; ++i) {
      const hhrl_linear_bwd_desc& l = d[i];
      if (l.dx && (l.dx_accumulate ? 1 : 0) == pass) {
        g[m] = gemm_args(l.dy, l.lddy, 1, l.w, l.K, 1, l.dx, l.lddx, M, l.K, l.N, nullptr, HHRL_ACT_NONE, pass);
        if (fuse && l.act != HHRL_ACT_NONE) {
          g[m].Y = l.y; g[m].y_si = l.ldy; g[m].y_sk = 1; g[m].y_act = l.act;
        }
        ++m;
      }
    }
    if (pass == 1) {
      // accumulating products into one buffer must not run concurrently with each other
      for (int a = 0; a < m; ++a)
        for (int b = a + 1; b < m; ++b)
          HHRL_REQUIRE(g[a].C != g[b].C, "linear_backward_group: two layers accumulate into the same dx");
    }
    if (!dw_done && (m > 0 || pass == 1)) {
      for (int i = 0; i < n; ++i) {
        const hhrl_linear_bwd_desc& l = d[i];
        if (!l.dw) continue;
        g[m] = gemm_args(l.dy, 1, l.lddy, l.x, l.ldx, 1, l.dw, l.K, l.N, l.K, M, nullptr, HHRL_ACT_NONE, 2);
        if (fuse) {
          if (l.act != HHRL_ACT_NONE) { g[m].Y = l.y; g[m].y_si = 1; g[m].y_sk = l.ldy; g[m].y_act = l.act; }
          g[m].rowsum = l.db;
        }
        ++m;
      }
      dw_done = true;
    }
    HHRL_TRY(flush(m));
  }
  return HHRL_OK;
}

int hhrl_sum_pool(const float* in, int ldi, float* out, int ldo, int B, int N, int C, void* stream) {
  HHRL_REQUIRE(in && out, "sum_pool: null pointer");
  HHRL_REQUIRE(B > 0 && N > 0 && C > 0 && ldi >= C && ldo >= C, "sum_pool: bad shape B=%d N=%d C=%d ldi=%d ldo=%d", B, N, C, ldi, ldo);
  ProfScope ps("sum_pool_kernel", as_stream(stream), 0.0, 4.0 * B * C * (N + 1));
  hipLaunchKernelGGL(sum_pool_kernel, dim3(ceil_div((long)B * C, 256)), dim3(256), 0, as_stream(stream), in, ldi, out, ldo, B, N, C);
  return launch_status("sum_pool_kernel");
}

int hhrl_broadcast_rows(const float* in, int ldi, float* out, int ldo, int B, int N, int C, void* stream) {
  HHRL_REQUIRE(in && out, "broadcast_rows: null pointer");
  HHRL_REQUIRE(B > 0 && N > 0 && C > 0 && ldi >= C && ldo >= C, "broadcast_rows: bad shape B=%d N=%d C=%d ldi=%d ldo=%d", B, N, C, ldi, ldo);
  ProfScope ps("broadcast_rows_kernel", as_stream(stream), 0.0, 4.0 * B * C * (N + 1));
  hipLaunchKernelGGL(broadcast_rows_kernel, dim3(ceil_div((long)B * N * C, 256)), dim3(256), 0, as_stream(stream), in, ldi, out, ldo, B, N, C);
  return launch_status("broadcast_rows_kernel");
}

int hhrl_copy2d(const float* in, int ldi, float* out, int ldo, int M, int C, void* stream) {
  HHRL_REQUIRE(in && out, "copy2d: null pointer");
  HHRL_REQUIRE(M > 0 && C > 0 && ldi >= C && ldo >= C, "copy2d: bad shape M=%d C=%d ldi=%d ldo=%d", M, C, ldi, ldo);
  ProfScope ps("copy2d_kernel", as_stream(stream), 0.0, 8.0 * M * C);
  hipLaunchKernelGGL(copy2d_kernel, dim3(ceil_div((long)M * C, 256)), dim3(256), 0, as_stream(stream), in, ldi, out, ldo, M, C);
  return launch_status("copy2d_kernel");
}

int hhrl_gather_rows(const float* in, long ldi, int n_in, const int* idx, float* out, int ldo, int M, int C, void* stream) {
  HHRL_REQUIRE(in && idx && out, "gather_rows: null pointer");
  HHRL_REQUIRE(M > 0 && C > 0 && n_in > 0 && ldi >= C && ldo >= C, "gather_rows: bad shape M=%d C=%d n_in=%d ldi=%ld ldo=%d", M, C, n_in, ldi, ldo);
  ProfScope ps("gather_rows_kernel", as_stream(stream), 0.0, 8.0 * M * C);
  hipLaunchKernelGGL(gather_rows_kernel, dim3(ceil_div((long)M * C, 256)), dim3(256), 0, as_stream(stream), in, ldi, idx, n_in, out, ldo, M, C);
  return launch_status("gather_rows_kernel");
}

int hhrl_copy2d_group(const hhrl_copy_desc* d, int n, int M, void* stream) {
  HHRL_REQUIRE(d && n >= 1 && n <= HHRL_COPY_GROUP_MAX, "copy2d_group: %d blocks (1..%d supported)", n, HHRL_COPY_GROUP_MAX);
  HHRL_REQUIRE(M > 0, "copy2d_group: M=%d", M);
  CopyGroup g;
  long threads = 0;                       // of the block that needs most
  double bytes = 0.0;
  for (int i = 0; i < n; ++i) {
    HHRL_REQUIRE(d[i].out, "copy2d_group: block %d: null pointer", i);
    HHRL_REQUIRE(d[i].C > 0, "copy2d_group: block %d: C=%d", i, d[i].C);
    long need;
    if (d[i].in) {
      HHRL_REQUIRE(d[i].ldi >= d[i].C && d[i].ldo >= d[i].C, "copy2d_group: block %d: bad shape C=%d ldi=%d ldo=%d", i, d[i].C, d[i].ldi,
                   d[i].ldo);
      need = (long)M * d[i].C;
      bytes += 8.0 * M * d[i].C;
    } else {
      need = ((long)d[i].C + 3) / 4;
      bytes += 4.0 * d[i].C;
    }
    g.in[i] = CopyInst{d[i].in, d[i].out, d[i].ldi, d[i].ldo, d[i].C};
    threads = need > threads ? need : threads;
  }
  ProfScope ps("copy2d_group_kernel", as_stream(stream), 0.0, bytes);
  hipLaunchKernelGGL(copy2d_group_kernel, dim3(ceil_div(threads, 256), n), dim3(256), 0, as_stream(stream), g, M);
  return launch_status("copy2d_group_kernel");
}

int hhrl_gather_copy_group(const hhrl_gcopy_desc* d, int n, int M, const int* idx, int n_in, void* stream) {
  HHRL_REQUIRE(d && n >= 1 && n <= HHRL_COPY_GROUP_MAX, "gather_copy_group: %d blocks (1..%d supported)", n, HHRL_COPY_GROUP_MAX);
  HHRL_REQUIRE(M > 0, "gather_copy_group: M=%d", M);
  GCopyGroup g;
  long wgs = 1;
  double bytes = 0.0;
  for (int i = 0; i < n; ++i) {
    HHRL_REQUIRE(d[i].out, "gather_copy_group: block %d: null pointer", i);
    HHRL_REQUIRE(d[i].C > 0, "gather_copy_group: block %d: C=%d", i, d[i].C);
    HHRL_REQUIRE(d[i].gather >= 0 && d[i].gather <= 4, "gather_copy_group: block %d: gather=%d", i, d[i].gather);
    HHRL_REQUIRE(d[i].gather != 4 || (d[i].in && n_in > 0 && d[i].ldo >= d[i].C),
                 "gather_copy_group: block %d scatters rows, but there is no source, no table size or ldo < C", i);
    HHRL_REQUIRE(d[i].gather != 3 || (d[i].in && reinterpret_cast<uintptr_t>(d[i].in) % 8 == 0),
                 "gather_copy_group: block %d: the seed and offset of a block of normals must be 8-byte aligned", i);
    long need;
    if (d[i].in && d[i].gather == 4) {
      need = ceil_div(63, gc_tile_rows(d[i].C));
      bytes += 8.0 * 63 * d[i].C;
    } else if (d[i].in && d[i].gather < 2) {
      HHRL_REQUIRE(d[i].ldi >= d[i].C && d[i].ldo >= d[i].C, "gather_copy_group: block %d: bad shape C=%d ldi=%ld ldo=%d", i, d[i].C,
                   d[i].ldi, d[i].ldo);
      HHRL_REQUIRE(!d[i].gather || (idx && n_in > 0), "gather_copy_group: block %d gathers, but there are no row numbers", i);
      need = ceil_div(M, gc_tile_rows(d[i].C));
      bytes += 8.0 * M * d[i].C;
    } else {
      need = ceil_div(((long)d[i].C + 3) / 4, 256);           // four floats per thread
      bytes += (d[i].in ? 8.0 : 4.0) * d[i].C;
    }
    g.in[i] = GCopyInst{d[i].in, d[i].out, d[i].ldi, d[i].ldo, d[i].C, d[i].gather};
    wgs = need > wgs ? need : wgs;
  }
  ProfScope ps("gather_copy_group_kernel", as_stream(stream), 0.0, bytes);
  hipLaunchKernelGGL(gather_copy_group_kernel, dim3((unsigned)wgs, n), dim3(256), 0, as_stream(stream), g, M, idx, n_in);
  return launch_status("gather_copy_group_kernel");
}

int hhrl_fill(float* p, long n, float value, void* stream) {
  HHRL_REQUIRE(p && n > 0, "fill: null pointer or n <= 0");
  ProfScope ps("fill_kernel", as_stream(stream), 0.0, 4.0 * n);
  hipLaunchKernelGGL(fill_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, as_stream(stream), p, n, value);
  return launch_status("fill_kernel");
}

int hhrl_randn_fill(float* out, long n, unsigned long long seed, unsigned long long offset, void* stream) {
  HHRL_REQUIRE(out && n > 0, "randn_fill: null pointer or n <= 0");
  const unsigned long long quads = ((offset + (unsigned long long)n + 3) >> 2) - (offset >> 2);
  HHRL_REQUIRE(quads <= 0x7fffffffull * 256ull, "randn_fill: n=%ld too large for one launch", n);
  ProfScope ps("randn_fill_kernel", as_stream(stream), 0.0, 4.0 * n);
  hipLaunchKernelGGL(randn_fill_kernel, dim3(ceil_div((long)quads, 256)), dim3(256), 0, as_stream(stream), out, n, seed, offset);
  return launch_status("randn_fill_kernel");
}

}  // extern "C"
