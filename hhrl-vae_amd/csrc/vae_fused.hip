// Fused DeepSets-VAE training step for gfx950: four launches per step.
//
//   vae_lane_forward   per traffic lane: phi (per car, pooled), rho, heads,
//                      reparameterisation, latent branch of the decoder, and the
//                      per-lane half of func.0 (see the per-car kernel's header)
//   vae_car_fwd_bwd    per car: position term of func.0, func.2/4, squared error,
//                      the backward of those layers, df1 summed per traffic lane
//   vae_lane_backward  per traffic lane: the per-lane half of func.0's backward, then
//                      the backward of everything in the first kernel (from its saved
//                      activations), KL gradient included
//   vae_reduce_grads   fixed-order sum of the per-workgroup gradient slabs
//
// Every matrix product runs on the f32 matrix instructions (32x32x2, and 16x16x4
// for thin outputs) with operands read from LDS "images": activations as
// [row][feature] with an odd row stride, weights as [out][in] with an odd row
// stride.  With those strides the three products of a layer (forward, input
// gradient, weight gradient) all read their operands conflict-free from the same
// images, and an accumulator tile (rows on registers, columns on lanes) is
// written back conflict-free.  Weight-gradient tiles stay in accumulator
// registers across a workgroup's tiles and are written once, to that workgroup's
// slab; what crosses tile boundaries (the sums of df1 per traffic lane) travels as
// one row per (car tile, traffic lane), added up in tile order: no atomics.
#include "common.h"
#include "mfma_tiles.h"

#include <mutex>
#include <type_traits>

namespace hhrl {
namespace vf {

using namespace mt;


// Diagnostic builds (-DHHRL_STAMPS) record the shader clock of workgroup 0, thread 0
// at marked points of the fused kernels; hhrl_debug_stamps() copies them out.
#ifdef HHRL_STAMPS
__device__ long long g_stamps[4][64];
#define HHRL_STAMP(k, i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_stamps[k][i] = clock64(); } while (0)
// start and end of every workgroup of the one-launch step: constant-rate counter and shader clock
__device__ long long g_wg_span[256][4];
#define HHRL_WG_SPAN(i) do { if (threadIdx.x == 0 && blockIdx.x < 256) { g_wg_span[blockIdx.x][i] = wall_clock64(); g_wg_span[blockIdx.x][2 + i] = clock64(); } } while (0)
#else
#define HHRL_STAMP(k, i) do { } while (0)
#define HHRL_WG_SPAN(i) do { } while (0)
#endif

// parameter order == Encoder.parameters() order (weight, bias interleaved)
enum { PHI0 = 0, PHI2, RHO0, RHO2, MEAN, LOGVAR, DEC0, DEC2, LOC, FUNC0, FUNC2, FUNC4, NLAYER };

struct Offsets {
  long w[NLAYER], b[NLAYER];
  long total;
};

// ===================================================================== per-car kernel
// func.0 acts on [d2 of the car's traffic lane | loc embedding of the car].  Both parts are affine in
// something much smaller than a car row: d2 is shared by the cars of a lane, the embedding is
// loc * wloc + bloc.  With W0 = [A | E] (columns 0..31 and 32..63) the pre-activation of func.0 is
//     U[lane] + loc * p        U = d2 A^T + b0 + E bloc  (per lane: the forward lane kernel)     p = E wloc
// and with S[lane] = sum of df1 over the lane's cars, s = sum of loc * df1 and db0 = sum of df1 over all cars
//     dd2 = S A      dA = S^T d2                          (per lane: the backward lane kernel)
//     dE = s wloc^T + db0 bloc^T      dwloc = E^T s      dbloc = E^T db0
// so no product with func.0's matrix is left per car.  The per-car kernel computes f1 element-wise,
// runs func.2 / func.4 and their backward on the matrix core, and leaves S (one row per (car tile,
// traffic lane in it), summed on the matrix core with a 0/1 indicator operand), s and db0.
struct CarArgs {
  const float* data;  // [rows][2]  (loc, v)
  const float* u;     // [B][128]
  const float* pvec;  // [128]
  const float *w0, *w2, *b2, *w4, *b4;
  float* spart;       // [tiles * R][128] sum of df1 per traffic lane within a tile of 16 cars:
                      // row tile*R + s belongs to traffic lane (tile*16)/N + s
  float* slab;        // [gridDim.x][total]
  float* loss_part;   // [gridDim.x]
  Offsets off;
  long rows;          // B*N
  int N, tiles, R;    // tiles of CAR_TILE cars; R = most traffic lanes a tile can touch
  float recon_scale;
  int vec4;           // slab rows and the func.2 weight offset are 16-byte aligned
};

// A wave owns a tile of 16 cars; eight waves per workgroup, two per SIMD, so that one wave's element-wise
// phases and LDS round trips run under the other's matrix instructions.
// A workgroup's slab entries and the activation images that one kernel of the four-launch form leaves for the
// next go through the L2 as they are produced (store_through), so that the end of the kernel does not wait for
// their write-back.
__device__ __forceinline__ void slab_put(float* p, float v) { store_through(p, v); }
__device__ __forceinline__ void stream_put(float4* p, const float4& v) { store_through(p, v); }

constexpr int CAR_TILE = 16, CAR_WAVES = 8, CAR_THREADS = 64 * CAR_WAVES;
constexpr int CAR_W2 = 0;                          // [32][129]
constexpr int CAR_SMALL = CAR_W2 + 32 * 129;       // b2[32] w4[32] b4[1]
constexpr int CAR_WAVE0 = CAR_SMALL + 96;
constexpr int CAR_F1 = 0;                          // per wave: [16][129] f1, later df1
constexpr int CAR_F2 = CAR_F1 + 16 * 129;          // [16][33]
constexpr int CAR_ROW = CAR_F2 + 16 * 33;          // dpre[16] (+48 spare)
constexpr int CAR_WAVE_FLOATS = CAR_ROW + 64;
// the end of the kernel reuses the LDS: a buffer of CAR_Q1 floats per wave, then s | db0 | partial products
constexpr int CAR_Q1 = 4420, CAR_FIN = CAR_WAVES * CAR_Q1, CAR_FIN_FLOATS = 256 + 16 * 64;
constexpr int CAR_IMG_FLOATS = CAR_WAVE0 + CAR_WAVES * CAR_WAVE_FLOATS;
constexpr int CAR_LDS_FLOATS = CAR_IMG_FLOATS > CAR_FIN + CAR_FIN_FLOATS ? CAR_IMG_FLOATS : CAR_FIN + CAR_FIN_FLOATS;
static_assert(CAR_LDS_FLOATS * 4 <= 163840, "per-car kernel must fit the 160 KiB LDS");

// most traffic lanes that CAR_TILE consecutive cars can belong to
static inline int lanes_per_car_tile(int N) { const int v = (CAR_TILE - 1) / N + 2; return v < CAR_TILE ? v : CAR_TILE; }

__global__ __launch_bounds__(CAR_THREADS, 1) void vae_car_fwd_bwd_kernel(CarArgs g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;            // operand lanes of the 32x32 products (dW2)
  const int j = lane & 15, q = lane >> 4;            // 16x16 blocks: column j, rows 4 q + reg

  HHRL_STAMP(0, 0);
  float* W2s = lds + CAR_W2;
  float* b2s = lds + CAR_SMALL;
  float* w4s = b2s + 32;
  float* b4s = w4s + 32;
  float* wl = lds + CAR_WAVE0 + wave * CAR_WAVE_FLOATS;
  float* f1i = wl + CAR_F1;
  float* f2i = wl + CAR_F2;
  float* dpre = wl + CAR_ROW;

  // Inputs of a tile.  Lane j (all four groups q) holds car j; for the four rows 4 q + reg of the
  // accumulator layout a lane also holds their loc, and column 16 blk + j of their traffic lane's U row.
  float in_v, loc_i[4], ind[4];
  int in_b;
  f32x4 f1[8];                                       // U rows on arrival, f1 after the activation
  auto fetch_tile = [&](int tile) {
    const int row = tile * CAR_TILE + j;                             // the host keeps the car count below 2^31
    const bool valid = tile < g.tiles && row < g.rows;
    const float2 lv = valid ? *reinterpret_cast<const float2*>(g.data + 2 * (long)row) : make_float2(0.f, 0.f);
    in_b = valid ? (int)((unsigned)row / (unsigned)g.N) : -1;
    // all exchanges first, then all loads from clamped addresses (no branch between them), so that the
    // round trips overlap.  Rows without a car read lane 0's row; their gradients are zero (dpre = 0).
    int b_i[4], b_k[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      b_i[e] = __shfl(in_b, 4 * q + e, 64);
      b_k[e] = __shfl(in_b, 4 * e + q, 64);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float* up = g.u + (long)max(b_i[e], 0) * 128 + j;
#pragma unroll
      for (int blk = 0; blk < 8; ++blk) f1[blk][e] = up[16 * blk];
    }
    // 0/1 operand of the per-lane sums: step st of the product takes car 4 st + q, output row s = lane j
    const int b_first = __builtin_amdgcn_readfirstlane(in_b);
#pragma unroll
    for (int e = 0; e < 4; ++e) ind[e] = (b_k[e] >= 0 && b_k[e] - b_first == j) ? 1.f : 0.f;
    in_v = lv.y;
#pragma unroll
    for (int e = 0; e < 4; ++e) loc_i[e] = __shfl(lv.x, 4 * q + e, 64);
  };

  float p_t[8];
  {
    // the matrix is requested first (the longest transfer), then the wave's tile; the narrow layers
    // travel in the same round trip
    float4 v2[2];
    request_w4<32, 128, 2, CAR_THREADS>(v2, g.w2, tid);
    const float p_b2 = g.b2[r], p_w4 = g.w4[r], p_b4 = g.b4[0];
#pragma unroll
    for (int blk = 0; blk < 8; ++blk) p_t[blk] = g.pvec[16 * blk + j];
    fetch_tile(blockIdx.x * CAR_WAVES + wave);
    commit_w4<32, 128, 2, CAR_THREADS>(W2s, 129, v2, tid);
    if (tid < 32) {
      b2s[tid] = p_b2;
      w4s[tid] = p_w4;
    }
    if (tid == 0) b4s[0] = p_b4;
  }
  __syncthreads();
  HHRL_STAMP(0, 1);

  // weight-gradient accumulators of func.2 and the column sums, kept across the wave's tiles
  f32x16 dW2[4];
  float db0[8], sacc[8];
#pragma unroll
  for (int blk = 0; blk < 8; ++blk) db0[blk] = sacc[blk] = 0.f;
  float w4acc = 0.f, db2acc = 0.f, b4acc = 0.f, lossacc = 0.f;

  auto process_tile = [&](auto first_tag, int tile) {
    constexpr bool FIRST = decltype(first_tag)::value;
    if (!FIRST) fetch_tile(tile);                                  // the first tile was requested in the prologue
    const float v = in_v;
    const bool valid = in_b >= 0;
    HHRL_STAMP(0, 2);

    // ---- forward: func.0 = ReLU(U[lane] + loc * p)
#pragma unroll
    for (int blk = 0; blk < 8; ++blk) {
#pragma unroll
      for (int e = 0; e < 4; ++e) f1[blk][e] = fmaxf(fmaf(loc_i[e], p_t[blk], f1[blk][e]), 0.f);
      store_block16<129>(f1i, 0, 16 * blk, f1[blk], lane);
    }
    wave_sync();
    HHRL_STAMP(0, 3);

    // ---- forward: func.2 (128 -> 32, ReLU)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      f32x4 c = wave_mma16<128, 129, 1, 1, 129>(f1i, W2s + 16 * t * 129, lane, b2s[16 * t + j]);
#pragma unroll
      for (int e = 0; e < 4; ++e) c[e] = fmaxf(c[e], 0.f);
      store_block16<33>(f2i, 0, 16 * t, c, lane);
    }
    wave_sync();
    HHRL_STAMP(0, 4);

    // ---- forward: func.4 (32 -> 1, sigmoid), squared error and its gradient per car (lane j, eight features per group q)
    {
      float s = 0.f;
#pragma unroll
      for (int n = 0; n < 8; ++n) s += w4s[8 * q + n] * f2i[j * 33 + 8 * q + n];
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      s += b4s[0];
      const float out = 1.f / (1.f + expf(-s));
      const float d = out - v;
      float dp = 0.f;
      if (valid) {
        if (q == 0) lossacc += g.recon_scale * d * d;
        dp = 2.f * g.recon_scale * d * out * (1.f - out);
      }
      if (q == 0) {
        dpre[j] = dp;
        b4acc += dp;
      }
    }
    wave_sync();
    HHRL_STAMP(0, 5);

    // ---- backward func.4 / ReLU of func.2: dW4, db2 and df2 (in place over the f2 image); lane: feature r, cars 8 h ..
    {
      const float w4n = w4s[r];
#pragma unroll
      for (int rr = 8 * h; rr < 8 * h + 8; ++rr) {
        const float fv = f2i[rr * 33 + r];
        const float dp = dpre[rr];
        w4acc += dp * fv;
        const float df = fv > 0.f ? dp * w4n : 0.f;
        db2acc += df;
        f2i[rr * 33 + r] = df;
      }
    }
    wave_sync();
    HHRL_STAMP(0, 6);

    // ---- backward func.2: dW2 += df2^T f1 (32x32 tiles over the 16 cars) ; df1 = (df2 W2) o relu'(f1)
    if (FIRST) {
#pragma unroll
      for (int t = 0; t < 4; ++t) dW2[t] = splat(0.f);
    }
    wave_mma<4, CAR_TILE, 1, 33, 129, 1>(dW2, f2i, f1i, lane);
    f32x4 df1[8];
#pragma unroll
    for (int blk = 0; blk < 8; ++blk) df1[blk] = wave_mma16<32, 33, 1, 129, 1>(f2i, W2s + 16 * blk, lane, 0.f);
    wave_sync();                           // all reads of the f1 image are done
    HHRL_STAMP(0, 7);
    // ---- ReLU mask of func.0, its column sums (db0, and s weighted by loc), df1 over the f1 image
#pragma unroll
    for (int blk = 0; blk < 8; ++blk) {
      float cs = 0.f, ss = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float dv = f1[blk][e] > 0.f ? df1[blk][e] : 0.f;
        df1[blk][e] = dv;
        cs += dv;
        ss = fmaf(dv, loc_i[e], ss);
      }
      db0[blk] += cs;
      sacc[blk] += ss;
      store_block16<129>(f1i, 0, 16 * blk, df1[blk], lane);
    }
    wave_sync();
    HHRL_STAMP(0, 8);

    // ---- df1 summed per traffic lane within the tile on the matrix core:
    //      P[s][col] = sum over the tile's cars of [lane of car == first lane of the tile + s] * df1[car][col]
    //      (exact: the indicator is 0 or 1, the cars of a lane add up in car order)
    {
      f32x4 P[8];
#pragma unroll
      for (int blk = 0; blk < 8; ++blk) P[blk] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int st = 0; st < 4; ++st) {
#pragma unroll
        for (int blk = 0; blk < 8; ++blk)
          P[blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(ind[st], f1i[(4 * st + q) * 129 + 16 * blk + j], P[blk], 0, 0, 0);
      }
      // every tile owns R rows; a tile's rows past its last traffic lane are zero and nobody reads them
      float* dst = g.spart + (long)tile * g.R * 128 + j;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (4 * q + e < g.R) {
#pragma unroll
          for (int blk = 0; blk < 8; ++blk) slab_put(&dst[(4 * q + e) * 128 + 16 * blk], P[blk][e]);
        }
      }
    }
    wave_sync();
    HHRL_STAMP(0, 9);
  };
  {
    const int tile0 = blockIdx.x * CAR_WAVES + wave, stride = gridDim.x * CAR_WAVES;
    if (tile0 < g.tiles) {
      process_tile(std::true_type(), tile0);
      for (int tile = tile0 + stride; tile < g.tiles; tile += stride) process_tile(std::false_type(), tile);
    } else {
#pragma unroll
      for (int a = 0; a < 4; ++a) dW2[a] = splat(0.f);
    }
  }

  // ---- combine the lane groups that hold different rows of the same columns
#pragma unroll
  for (int blk = 0; blk < 8; ++blk) {
    db0[blk] += __shfl_xor(db0[blk], 16, 64);
    db0[blk] += __shfl_xor(db0[blk], 32, 64);
    sacc[blk] += __shfl_xor(sacc[blk], 16, 64);
    sacc[blk] += __shfl_xor(sacc[blk], 32, 64);
  }
  w4acc += __shfl_xor(w4acc, 32, 64);
  db2acc += __shfl_xor(db2acc, 32, 64);
  b4acc = wave_sum(b4acc);
  lossacc = wave_sum(lossacc);

  // ---- combine the eight waves in LDS in a fixed order.  Every image is dead by now: each wave stores
  //      its sums into a buffer of its own, all threads then add the eight buffers element-wise
  //      (((w0 + w1) + (w2 + w3)) + ((w4 + w5) + (w6 + w7))) while writing the slab.
  __syncthreads();
  HHRL_STAMP(0, 10);
  // W0[8 og + k][32 + jj] for the loc gradients at the very end, requested now
  float w0e[8];
  {
    const int jj = tid & 31, og = tid >> 5;
#pragma unroll
    for (int k = 0; k < 8; ++k) w0e[k] = g.w0[(8 * og + k) * 64 + 32 + jj];
  }
  float* slab = g.slab + (long)blockIdx.x * g.off.total;
  constexpr int Q1 = CAR_Q1;
  constexpr int B_W2 = 0, B_B0 = 4096, B_S = B_B0 + 128, B_B2 = B_S + 128, B_W4 = B_B2 + 32, B_B4 = B_W4 + 32;
  constexpr int B_LOSS = B_B4 + 1;
  static_assert(B_LOSS + 1 <= Q1 && Q1 % 4 == 0, "per-wave buffer layout");
  {
    float* mine = lds + wave * Q1;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int i = 0; i < 16; ++i) mine[B_W2 + drow(i, h) * 128 + jt * 32 + r] = dW2[jt][i];
    if (q == 0) {
#pragma unroll
      for (int blk = 0; blk < 8; ++blk) {
        mine[B_B0 + 16 * blk + j] = db0[blk];
        mine[B_S + 16 * blk + j] = sacc[blk];
      }
    }
    if (h == 0) {
      mine[B_B2 + r] = db2acc;
      mine[B_W4 + r] = w4acc;
      if (r == 0) {
        mine[B_B4] = b4acc;
        mine[B_LOSS] = lossacc;
      }
    }
  }
  __syncthreads();
  HHRL_STAMP(0, 11);
  auto add4 = [](const float4& a, const float4& b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); };
  auto all8 = [&](int i) {
    return ((lds[i] + lds[Q1 + i]) + (lds[2 * Q1 + i] + lds[3 * Q1 + i])) +
           ((lds[4 * Q1 + i] + lds[5 * Q1 + i]) + (lds[6 * Q1 + i] + lds[7 * Q1 + i]));
  };
  if (g.vec4) {                                      // 16-byte aligned slab and offsets (wave-uniform): four elements per access
    const float4* qv = reinterpret_cast<const float4*>(lds + B_W2);
    float4* dst = reinterpret_cast<float4*>(slab + g.off.w[FUNC2]);
    constexpr int S4 = Q1 / 4;
#pragma unroll
    for (int u = 0; u < 1024 / CAR_THREADS; ++u) {
      const int i = u * CAR_THREADS + tid;
      dst[i] = add4(add4(add4(qv[i], qv[S4 + i]), add4(qv[2 * S4 + i], qv[3 * S4 + i])),
                    add4(add4(qv[4 * S4 + i], qv[5 * S4 + i]), add4(qv[6 * S4 + i], qv[7 * S4 + i])));
    }
  } else {
    for (int i = tid; i < 4096; i += CAR_THREADS) slab_put(&slab[g.off.w[FUNC2] + i], all8(B_W2 + i));
  }
  float* fin = lds + CAR_FIN;
  if (tid < 128) {
    const float sv = all8(B_S + tid);
    fin[tid] = sv;
    slab_put(&slab[g.off.w[FUNC0] + tid * 64 + 32], sv);       // s[o] travels in the first position of row o that no sum uses
  } else if (tid < 256) {
    const float dv = all8(B_B0 + tid - 128);
    fin[tid] = dv;
    slab_put(&slab[g.off.b[FUNC0] + tid - 128], dv);
  } else if (tid < 288) {
    slab_put(&slab[g.off.b[FUNC2] + tid - 256], all8(B_B2 + tid - 256));
  } else if (tid < 320) {
    slab_put(&slab[g.off.w[FUNC4] + tid - 288], all8(B_W4 + tid - 288));
  } else if (tid == 320) {
    slab_put(&slab[g.off.b[FUNC4]], all8(B_B4));
  } else if (tid == 321) {
    g.loss_part[blockIdx.x] = all8(B_LOSS);
  }
  __syncthreads();
  // ---- loc embedding: dwloc[jj] = sum_o s[o] W0[o][32 + jj], dbloc[jj] = sum_o db0[o] W0[o][32 + jj]
  //      (this workgroup's share; sixteen groups of eight o each, combined in a fixed order)
  {
    const int jj = tid & 31, og = tid >> 5;
    float pw = 0.f, pb = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      pw = fmaf(fin[8 * og + k], w0e[k], pw);
      pb = fmaf(fin[128 + 8 * og + k], w0e[k], pb);
    }
    fin[256 + og * 64 + jj] = pw;
    fin[256 + og * 64 + 32 + jj] = pb;
  }
  __syncthreads();
  if (tid < 64) {
    const float* pp = fin + 256 + tid;
    float sum = 0.f;
#pragma unroll
    for (int og = 0; og < 16; og += 4) sum += (pp[og * 64] + pp[(og + 1) * 64]) + (pp[(og + 2) * 64] + pp[(og + 3) * 64]);
    slab_put(&slab[(tid < 32 ? g.off.w[LOC] + tid : g.off.b[LOC] + tid - 32)], sum);
  }
  HHRL_STAMP(0, 12);
}

// ===================================================================== per-lane kernels
// One workgroup owns tiles of 32 traffic lanes.  ALL per-lane weights are
// copied into LDS once per workgroup and every activation of the chain lives
// in an LDS image.  Every global load a tile needs (weights, car states, saved
// activations, the gradient rows of its cars) is requested up front into
// registers, so the round trips overlap, and is written to LDS once.
// Products with at most 64 output columns run on v_mfma_f32_16x16x4_f32 so that
// the four waves own disjoint 16x16 output blocks and no partial sums have to
// be combined; wider layers give each wave a 32-column tile of
// v_mfma_f32_32x32x2_f32.  The forward kernel saves its activation images,
// the backward kernel reloads them and overwrites activations with their
// gradients in place.
struct LaneArgs {
  const float* data;   // [B*N][2]
  const float* eps;    // [B][g]
  const float* w[NLAYER];
  const float* b[NLAYER];
  float* u;            // [B][128] forward output: func.0 pre-activation of the lane's cars without the position term
  float* aux;          // forward output: p[128] = W0[:, 32:] wloc, then copies of wloc[32] and bloc[32]
  float* act;          // [tiles][LN_ACT_FLOATS] activation images: written by the forward, read by the backward kernel
  float* wimg;         // [LN_WIMG_FLOATS] the weight and bias images exactly as they lie in LDS: written by the forward
                       // kernel's first workgroup, copied verbatim by the backward kernel
  const float* spart;  // [car tiles * R][128] backward input: df1 summed per (car tile, traffic lane in it)
  float* slab;
  float* kl_part;      // [gridDim.x]
  Offsets off;
  int B, N, g, tiles, R;
};


// LDS map (floats)
constexpr int LW_PHI2 = 0;                        // [32][17]
constexpr int LW_RHO0 = LW_PHI2 + 32 * 17;        // [128][33]
constexpr int LW_RHO2 = LW_RHO0 + 128 * 33;       // [64][129]
constexpr int LW_HEAD = LW_RHO2 + 64 * 129;       // [32][65]   rows 0..g-1 to_mean, g..2g-1 to_logvar, rest zero
constexpr int LW_DEC0 = LW_HEAD + 32 * 65;        // [128][17]  columns >= g zero
constexpr int LW_DEC2 = LW_DEC0 + 128 * 17;       // [32][129]
constexpr int LW_BIAS = LW_DEC2 + 32 * 129;       // phi2[32] rho0[128] rho2[64] head[32] dec0[128] dec2[32]
constexpr int LB_PHI2 = 0, LB_RHO0 = 32, LB_RHO2 = 160, LB_HEAD = 224, LB_DEC0 = 256, LB_DEC2 = 384, LB_END = 416;
constexpr int LB_Q = LB_END, LB_ALL = LB_Q + 128; // q[128] = b0 + W0[:, 32:] bloc (forward kernel only)
constexpr int LN_WIMG_FLOATS = LW_BIAS + LB_END;  // every weight image and the staged biases, contiguous from 0
static_assert(LN_WIMG_FLOATS % 4 == 0, "the weight images are copied 16 bytes at a time");
constexpr int LI_H1 = LW_BIAS + LB_ALL;           // [32][33]  summed hidden of phi (16 columns used)
constexpr int LI_POOL = LI_H1 + 32 * 33;          // [32][33]
constexpr int LI_R1 = LI_POOL + 32 * 33;          // [32][129]
constexpr int LI_R2 = LI_R1 + 32 * 129;           // [32][65]
constexpr int LI_ML = LI_R2 + 32 * 65;            // [32][33]  mean | logvar
constexpr int LI_Z = LI_ML + 32 * 33;             // [32][17]  (in a 33-wide slot)
constexpr int LI_D1 = LI_Z + 32 * 33;             // [32][129]
constexpr int LI_D2 = LI_D1 + 32 * 129;           // [32][33]
constexpr int LI_DML = LI_D2 + 32 * 33;           // [32][33]
constexpr int LI_TMP = LI_DML + 32 * 33;          // [32][33]
constexpr int LI_CARS = LI_TMP + 32 * 33;         // the tile's car states, 32 lanes x N x 2 (N <= STAGE_MAX_CARS)
constexpr int STAGE_MAX_CARS = 16;
constexpr int LN_LDS_FLOATS = LI_CARS + 64 * STAGE_MAX_CARS;
static_assert(LN_LDS_FLOATS * 4 <= 163840 - 64, "per-lane kernels must fit the 160 KiB LDS");
constexpr int LN_ACT_FLOATS = LI_DML - LI_H1;     // h1 | pooled | r1 | r2 | mean,logvar | z | d1 | d2, verbatim
static_assert(LI_H1 % 4 == 0 && LN_ACT_FLOATS % 4 == 0, "activation images are copied 16 bytes at a time");
constexpr int ACT_V4 = LN_ACT_FLOATS / 4;

__device__ __forceinline__ float colsum32(const float* img, int ld, int col) {
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int rr = 0; rr < 32; rr += 2) {
    s0 += img[rr * ld + col];
    s1 += img[(rr + 1) * ld + col];
  }
  return s0 + s1;
}

// ---- registers that carry global loads from request to LDS, grouped by when the chain needs them:
//      group P (phi.2, rho.0, car states), group R (rho.2, heads), group D (decoder.0, decoder.2), biases.
//      Every request of a tile is issued before the first LDS write, so the round trips overlap.
//      NT = threads of the workgroup that share the copies (256, or 512 when four more waves help with them).
template <int NT> struct RegsP { float4 rho0[W4Map<128, 32, NT>::STEPS], phi2; float cars[1024 / NT]; };
template <int NT> struct RegsR { float4 rho2[W4Map<64, 128, NT>::STEPS]; float head[2048 / NT]; };
template <int NT> struct RegsD { float4 dec2[W4Map<32, 128, NT>::STEPS]; float dec0[2048 / NT]; };
template <int NT> struct RegsB { float bias[512 / NT]; };

__device__ __forceinline__ const float* lane_bias_src(const LaneArgs& g, int idx) {
  const int gd = g.g;
  if (idx < LB_RHO0) return g.b[PHI2] + idx;
  if (idx < LB_RHO2) return g.b[RHO0] + (idx - LB_RHO0);
  if (idx < LB_HEAD) return g.b[RHO2] + (idx - LB_RHO2);
  if (idx < LB_DEC0) {
    const int c = idx - LB_HEAD;
    return c < gd ? g.b[MEAN] + c : (c < 2 * gd ? g.b[LOGVAR] + (c - gd) : nullptr);
  }
  if (idx < LB_DEC2) return g.b[DEC0] + (idx - LB_DEC0);
  if (idx < LB_END) return g.b[DEC2] + (idx - LB_DEC2);
  return nullptr;
}

template <int COLS>
__device__ __forceinline__ void commit4(float* img, int ld, int i4, const float4& v) {
  const int idx = i4 * 4, r = idx / COLS, c = idx % COLS;
  float* p = img + r * ld + c;
  p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w;
}

template <int NT>
__device__ __forceinline__ void request_bias(RegsB<NT>& R, const LaneArgs& g, int tid) {
#pragma unroll
  for (int u = 0; u < 512 / NT; ++u) {
    const float* src = lane_bias_src(g, u * NT + tid);
    R.bias[u] = src ? *src : 0.f;
  }
}
template <int NT>
__device__ __forceinline__ void commit_bias(float* lds, const RegsB<NT>& R, int tid) {
#pragma unroll
  for (int u = 0; u < 512 / NT; ++u) {
    const int idx = u * NT + tid;
    if (idx < LB_END) lds[LW_BIAS + idx] = R.bias[u];
  }
}

template <int NT>
__device__ __forceinline__ void request_cars(float (&cars)[1024 / NT], const LaneArgs& g, int row0, int tid) {
  const long base = (long)row0 * g.N * 2;
  const long lim = (long)g.B * g.N * 2;
  const int n = g.N <= STAGE_MAX_CARS ? 64 * g.N : 0;
#pragma unroll
  for (int u = 0; u < 1024 / NT; ++u) {
    const int i = u * NT + tid;
    cars[u] = (i < n && base + i < lim) ? g.data[base + i] : 0.f;
  }
}
template <int NT>
__device__ __forceinline__ void commit_cars(float* lds, const float (&cars)[1024 / NT], const LaneArgs& g, int tid) {
  const int n = g.N <= STAGE_MAX_CARS ? 64 * g.N : 0;
#pragma unroll
  for (int u = 0; u < 1024 / NT; ++u) {
    const int i = u * NT + tid;
    if (i < n) lds[LI_CARS + i] = cars[u];
  }
}

template <int NT>
__device__ __forceinline__ void request_P(RegsP<NT>& R, const LaneArgs& g, int row0, int tid) {
  request_w4<128, 32, W4Map<128, 32, NT>::STEPS, NT>(R.rho0, g.w[RHO0], tid);
  R.phi2 = tid < 128 ? reinterpret_cast<const float4*>(g.w[PHI2])[tid] : make_float4(0.f, 0.f, 0.f, 0.f);
  request_cars<NT>(R.cars, g, row0, tid);
}
template <int NT>
__device__ __forceinline__ void commit_P(float* lds, const RegsP<NT>& R, const LaneArgs& g, int tid) {
  commit_w4<128, 32, W4Map<128, 32, NT>::STEPS, NT>(lds + LW_RHO0, 33, R.rho0, tid);
  if (tid < 128) commit4<16>(lds + LW_PHI2, 17, tid, R.phi2);
  commit_cars<NT>(lds, R.cars, g, tid);
}

template <int NT>
__device__ __forceinline__ void request_R(RegsR<NT>& R, const LaneArgs& g, int tid) {
  request_w4<64, 128, W4Map<64, 128, NT>::STEPS, NT>(R.rho2, g.w[RHO2], tid);
  const int gd = g.g, nh = 128 * gd;                   // heads: 2g rows x 64
#pragma unroll
  for (int u = 0; u < 2048 / NT; ++u) {
    const int e = u * NT + tid;
    R.head[u] = e < nh ? (e < 64 * gd ? g.w[MEAN][e] : g.w[LOGVAR][e - 64 * gd]) : 0.f;
  }
}
template <int NT>
__device__ __forceinline__ void commit_R(float* lds, const RegsR<NT>& R, const LaneArgs& g, int tid) {
  commit_w4<64, 128, W4Map<64, 128, NT>::STEPS, NT>(lds + LW_RHO2, 129, R.rho2, tid);
  const int gd = g.g, nh = 128 * gd;
#pragma unroll
  for (int u = 0; u < 2048 / NT; ++u) {
    const int e = u * NT + tid;
    if (e < nh) lds[LW_HEAD + (e >> 6) * 65 + (e & 63)] = R.head[u];
  }
  for (int i = tid; i < 32 * 65; i += NT)              // rows >= 2g of the heads image are zero
    if (i / 65 >= 2 * gd) lds[LW_HEAD + i] = 0.f;
}

template <int NT>
__device__ __forceinline__ void request_D(RegsD<NT>& R, const LaneArgs& g, int tid) {
  request_w4<32, 128, W4Map<32, 128, NT>::STEPS, NT>(R.dec2, g.w[DEC2], tid);
  const int nd = 128 * g.g;                            // decoder.0: 128 rows x g
#pragma unroll
  for (int u = 0; u < 2048 / NT; ++u) {
    const int e = u * NT + tid;
    R.dec0[u] = e < nd ? g.w[DEC0][e] : 0.f;
  }
}
template <int NT>
__device__ __forceinline__ void commit_D(float* lds, const RegsD<NT>& R, const LaneArgs& g, int tid) {
  commit_w4<32, 128, W4Map<32, 128, NT>::STEPS, NT>(lds + LW_DEC2, 129, R.dec2, tid);
  const int gd = g.g, nd = 128 * gd, lg = __ffs(gd) - 1;
#pragma unroll
  for (int u = 0; u < 2048 / NT; ++u) {
    const int e = u * NT + tid;
    if (e < nd) lds[LW_DEC0 + (e >> lg) * 17 + (e & (gd - 1))] = R.dec0[u];
  }
  for (int i = tid; i < 128 * 17; i += NT)             // columns >= g of the decoder.0 image are zero
    if (i % 17 >= gd) lds[LW_DEC0 + i] = 0.f;
}

// A contiguous range of the saved activation images, 16 bytes per thread and step.
template <int LO, int HI, int NT>
struct ActRange {
  static_assert(LO % 4 == 0 && HI % 4 == 0, "image boundaries are multiples of 4 floats");
  static constexpr int V4 = (HI - LO) / 4, STEPS = (V4 + NT - 1) / NT;
  float4 v[STEPS];
  __device__ __forceinline__ void request(const float* act_tile, int tid) { request_from(act_tile + (LO - LI_H1), tid); }
  __device__ __forceinline__ void request_from(const float* at_lo, int tid) {   // at_lo: global address of element LO
    const float4* src = reinterpret_cast<const float4*>(at_lo);
#pragma unroll
    for (int u = 0; u < STEPS; ++u) {
      const int i4 = u * NT + tid;
      v[u] = i4 < V4 ? src[i4] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  __device__ __forceinline__ void commit_at(float* slot, int tid) const {   // slot: 16-byte aligned
    float4* dst = reinterpret_cast<float4*>(slot);
#pragma unroll
    for (int u = 0; u < STEPS; ++u) {
      const int i4 = u * NT + tid;
      if (i4 < V4) dst[i4] = v[u];
    }
  }
  __device__ __forceinline__ void commit(float* lds, int tid) const { commit_at(lds + LO, tid); }
};

// ------------------------------------------------------------------ forward
// NT = 256, or 512: four more waves share the copies into LDS and the element-wise phases (at one wave
// per SIMD those are bound by instruction issue); the matrix phases stay on waves 0..3.
template <int NT>
__global__ __launch_bounds__(NT, 1) void vae_lane_forward_kernel(LaneArgs g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31;
  const int rb = wave & 1, cb = wave >> 1;      // 16-row / 16-column block owned by this wave in 32x32 outputs
  const int j16 = lane & 15;
  const int gd = g.g;
  float* bs = lds + LW_BIAS;
  float* Ih1 = lds + LI_H1;
  float* Ipool = lds + LI_POOL;
  float* Ir1 = lds + LI_R1;
  float* Ir2 = lds + LI_R2;
  float* Iml = lds + LI_ML;
  float* Iz = lds + LI_Z;
  float* Id1 = lds + LI_D1;
  float* Id2 = lds + LI_D2;
  const bool staged = g.N <= STAGE_MAX_CARS;
  const bool mat = wave < 4;                      // this wave takes part in the matrix phases
  HHRL_STAMP(1, 0);
  RegsB<NT> rB;
  RegsP<NT> rP;
  RegsR<NT> rR;
  RegsD<NT> rD;
  // phi.0 of hidden unit tid & 15 (every element this thread computes in the phi.0 phase uses that unit)
  const float wa = g.w[PHI0][2 * (tid & 15)], wb = g.w[PHI0][2 * (tid & 15) + 1], bb = g.b[PHI0][tid & 15];
  request_bias(rB, g, tid);
  request_P(rP, g, blockIdx.x * 32, tid);
  request_R(rR, g, tid);
  request_D(rD, g, tid);
  // func.0's columns for d2, as the B operand of the last product: row 32 wave + r, elements 16 h .. 16 h + 15
  // (step s of the product takes k = 16 h + s from both operands)
  float4 w0a[4];
  if (mat) {
    const float4* src = reinterpret_cast<const float4*>(g.w[FUNC0] + (32 * wave + r) * 64 + 16 * (lane >> 5));
#pragma unroll
    for (int e = 0; e < 4; ++e) w0a[e] = src[e];
  }
  // the position part of func.0, by the last 128 threads (output o each): p = E wloc, q = b0 + E bloc
  // with E = W0[:, 32:]
  float4 w0e[8];
  float b0o = 0.f;
  const int o_pq = tid - (NT - 128);
  if (o_pq >= 0) {
    const float4* src = reinterpret_cast<const float4*>(g.w[FUNC0] + o_pq * 64 + 32);
#pragma unroll
    for (int e = 0; e < 8; ++e) w0e[e] = src[e];
    b0o = g.b[FUNC0][o_pq];
  }
  commit_bias(lds, rB, tid);
  commit_P(lds, rP, g, tid);
  commit_R(lds, rR, g, tid);
  commit_D(lds, rD, g, tid);
  if (blockIdx.x == 0 && tid < 64) g.aux[128 + tid] = tid < 32 ? g.w[LOC][tid] : g.b[LOC][tid - 32];
  __syncthreads();
  HHRL_STAMP(1, 1);
  // The weight and bias images as they lie in LDS (the same in every workgroup), for the backward kernel,
  // which then copies 16 bytes per instruction instead of placing rows, padding and zero fill again.
  // The workgroups share the copy (one store per thread of the first few at the usual sizes), issued here so
  // that it completes under the first phase.
  {
    const float4* src = reinterpret_cast<const float4*>(lds);
    float4* dst = reinterpret_cast<float4*>(g.wimg);
    for (int i4 = blockIdx.x * NT + tid; i4 < LN_WIMG_FLOATS / 4; i4 += gridDim.x * NT) dst[i4] = src[i4];
  }

  for (int tile = blockIdx.x; tile < g.tiles; tile += gridDim.x) {
    const int row0 = tile * 32;
    const bool first = tile == (int)blockIdx.x;
    if (!first) {                                // later tiles: the images of the previous tile are dead (barrier at loop end)
      request_cars<NT>(rP.cars, g, row0, tid);
      commit_cars<NT>(lds, rP.cars, g, tid);
      __syncthreads();
    }
    // the tile's reparameterisation noise, requested long before its phase: elements tid (+ NT) of [32][16]
    float eps_r[512 / NT];
#pragma unroll
    for (int u = 0; u < 512 / NT; ++u) {
      const int idx = u * NT + tid, rr = idx >> 4, j = idx & 15;
      eps_r[u] = (j < gd && row0 + rr < g.B) ? g.eps[(long)(row0 + rr) * gd + j] : 0.f;
    }
    HHRL_STAMP(1, 2);
    // ---- phi.0 + ReLU summed over each lane's cars
    for (int idx = tid; idx < 32 * 16; idx += NT) {
      const int rr = idx >> 4, k = idx & 15;
      const int b = row0 + rr;
      float s = 0.f;
      if (b < g.B) {
        // two loops, so that each reads through a pointer of a known address space
        if (staged) {
          const float* x = lds + LI_CARS + rr * g.N * 2;
          for (int n = 0; n < g.N; ++n) s += fmaxf(wa * x[2 * n] + wb * x[2 * n + 1] + bb, 0.f);
        } else {
          const float* x = g.data + (long)b * g.N * 2;
          for (int n = 0; n < g.N; ++n) s += fmaxf(wa * x[2 * n] + wb * x[2 * n + 1] + bb, 0.f);
        }
      }
      Ih1[rr * 33 + k] = s;
    }
    __syncthreads();
    HHRL_STAMP(1, 3);
    // ---- phi.2 on the summed hidden: pooled = W2 h1s + N b2
    if (mat) {
      const f32x4 c = wave_mma16<16, 33, 1, 1, 17>(Ih1 + 16 * rb * 33, lds + LW_PHI2 + 16 * cb * 17, lane,
                                                   (float)g.N * bs[LB_PHI2 + 16 * cb + j16]);
      store_block16<33>(Ipool, 16 * rb, 16 * cb, c, lane);
    }
    __syncthreads();
    HHRL_STAMP(1, 4);
    // ---- rho.0 (32 -> 128, ReLU): one 32-column tile per wave
    if (mat) {
      f32x16 acc[1];
      acc[0] = splat(bs[LB_RHO0 + wave * 32 + r]);
      wave_mma<1, 32, 33, 1, 1, 33>(acc, Ipool, lds + LW_RHO0 + wave * 32 * 33, lane);
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[0][i] = fmaxf(acc[0][i], 0.f);
      store_tile<129>(Ir1, wave * 32, acc[0], lane);
    }
    __syncthreads();
    HHRL_STAMP(1, 5);
    // p and q (first tile only), by threads that have nothing to do in the matrix phases when helper
    // waves exist; q is first read after four more barriers
    if (first && o_pq >= 0) {
      float pv = 0.f, qv = b0o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float* wl4 = g.w[LOC] + 4 * e;
        const float* bl4 = g.b[LOC] + 4 * e;
        pv += w0e[e].x * wl4[0] + w0e[e].y * wl4[1] + w0e[e].z * wl4[2] + w0e[e].w * wl4[3];
        qv += w0e[e].x * bl4[0] + w0e[e].y * bl4[1] + w0e[e].z * bl4[2] + w0e[e].w * bl4[3];
      }
      bs[LB_Q + o_pq] = qv;
      if (blockIdx.x == 0) g.aux[o_pq] = pv;
    }
    // ---- rho.2 (128 -> 64): two 16x16 blocks per wave
    if (mat) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int col0 = 32 * cb + 16 * t;
        const f32x4 c = wave_mma16<128, 129, 1, 1, 129>(Ir1 + 16 * rb * 129, lds + LW_RHO2 + col0 * 129, lane,
                                                        bs[LB_RHO2 + col0 + j16]);
        store_block16<65>(Ir2, 16 * rb, col0, c, lane);
      }
    }
    __syncthreads();
    HHRL_STAMP(1, 6);
    // ---- heads (64 -> 2g; columns >= 2g come out as zero)
    if (mat) {
      const f32x4 c = wave_mma16<64, 65, 1, 1, 65>(Ir2 + 16 * rb * 65, lds + LW_HEAD + 16 * cb * 65, lane,
                                                   bs[LB_HEAD + 16 * cb + j16]);
      store_block16<33>(Iml, 16 * rb, 16 * cb, c, lane);
    }
    __syncthreads();
    HHRL_STAMP(1, 7);
    // ---- reparameterisation z = mean + eps * exp(logvar / 2), zero-padded to 16 columns
#pragma unroll
    for (int u = 0; u < 512 / NT; ++u) {
      const int idx = u * NT + tid, rr = idx >> 4, j = idx & 15;
      float zv = 0.f;
      if (j < gd && row0 + rr < g.B) zv = Iml[rr * 33 + j] + eps_r[u] * expf(0.5f * Iml[rr * 33 + gd + j]);
      Iz[rr * 17 + j] = zv;
    }
    __syncthreads();
    HHRL_STAMP(1, 8);
    // ---- decoder.0 (g -> 128, ReLU)
    if (mat) {
      f32x16 acc[1];
      acc[0] = splat(bs[LB_DEC0 + wave * 32 + r]);
      wave_mma<1, 16, 17, 1, 1, 17>(acc, Iz, lds + LW_DEC0 + wave * 32 * 17, lane);
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[0][i] = fmaxf(acc[0][i], 0.f);
      store_tile<129>(Id1, wave * 32, acc[0], lane);
    }
    __syncthreads();
    HHRL_STAMP(1, 9);
    // ---- decoder.2 (128 -> 32)
    if (mat) {
      const f32x4 c = wave_mma16<128, 129, 1, 1, 129>(Id1 + 16 * rb * 129, lds + LW_DEC2 + 16 * cb * 129, lane,
                                                      bs[LB_DEC2 + 16 * cb + j16]);
      store_block16<33>(Id2, 16 * rb, 16 * cb, c, lane);
    }
    __syncthreads();
    HHRL_STAMP(1, 10);
    // ---- outputs: U = d2 A^T + q for the per-car kernel (A = W0[:, :32]; column tile `wave`, straight from
    //      the accumulators), the activation images for the backward kernel
    if (mat) {
      f32x16 acc = splat(bs[LB_Q + wave * 32 + r]);
      const float* ap = Id2 + r * 33 + 16 * (lane >> 5);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[4 * e], w0a[e].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[4 * e + 1], w0a[e].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[4 * e + 2], w0a[e].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[4 * e + 3], w0a[e].w, acc, 0, 0, 0);
      }
      float* dst = g.u + (long)row0 * 128 + 32 * wave + r;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int rr = drow(i, lane >> 5);
        if (row0 + rr < g.B) slab_put(&dst[rr * 128], acc[i]);
      }
    }
    HHRL_STAMP(1, 11);
    {
      float4* dst = reinterpret_cast<float4*>(g.act + (long)tile * LN_ACT_FLOATS);
      const float4* src = reinterpret_cast<const float4*>(lds + LI_H1);
#pragma unroll
      for (int u = 0; u < (ACT_V4 + NT - 1) / NT; ++u) {
        const int i4 = u * NT + tid;
        if (i4 < ACT_V4) stream_put(&dst[i4], src[i4]);
      }
    }
    __syncthreads();
    HHRL_STAMP(1, 12);
  }
}

// ------------------------------------------------------------------ inference: encode
// DeepSetEncoder.forward alone (algorithm.py:40-47, Encoder.encode :109-116) for B lanes in one launch:
// the forward chain above up to the heads.  Lanes may hold different numbers of cars (lane_start), a lane
// without cars yields zeros like the reference's early return.
struct EncArgs {
  LaneArgs g;              // data = cars [total][2]; w, b, B, N, g, tiles
  const int* lane_start;   // [B+1] first car of every lane, or null: N cars per lane
  float* out;              // [B][ldo] mean | logvar
  int ldo;
};

__global__ __launch_bounds__(256, 1) void vae_encode_kernel(EncArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const LaneArgs& g = a.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31;
  const int rb = wave & 1, cb = wave >> 1, j16 = lane & 15, q16 = lane >> 4;
  const int gd = g.g;
  float* bs = lds + LW_BIAS;
  float* Ih1 = lds + LI_H1;
  float* Ipool = lds + LI_POOL;
  float* Ir1 = lds + LI_R1;
  float* Ir2 = lds + LI_R2;
  float* Iml = lds + LI_ML;
  float* cnt = lds + LI_TMP;           // cars per lane of the tile
  {
    RegsB<256> rB;
    RegsR<256> rR;
    float4 rho0[4];
    request_bias(rB, g, tid);
    request_w4<128, 32>(rho0, g.w[RHO0], tid);
    const float4 phi2 = tid < 128 ? reinterpret_cast<const float4*>(g.w[PHI2])[tid] : make_float4(0.f, 0.f, 0.f, 0.f);
    request_R(rR, g, tid);
    commit_bias(lds, rB, tid);
    commit_w4<128, 32>(lds + LW_RHO0, 33, rho0, tid);
    if (tid < 128) commit4<16>(lds + LW_PHI2, 17, tid, phi2);
    commit_R(lds, rR, g, tid);
  }
  __syncthreads();
  for (int tile = blockIdx.x; tile < g.tiles; tile += gridDim.x) {
    const int row0 = tile * 32;
    // ---- phi.0 + ReLU summed over each lane's cars
    for (int idx = tid; idx < 32 * 16; idx += 256) {
      const int rr = idx >> 4, k = idx & 15;
      const int b = row0 + rr;
      float s = 0.f;
      long c0 = 0, c1 = 0;
      if (b < g.B) {
        c0 = a.lane_start ? a.lane_start[b] : (long)b * g.N;
        c1 = a.lane_start ? a.lane_start[b + 1] : c0 + g.N;
        const float wa = g.w[PHI0][2 * k], wb = g.w[PHI0][2 * k + 1], bb = g.b[PHI0][k];
        for (long c = c0; c < c1; ++c) s += fmaxf(wa * g.data[2 * c] + wb * g.data[2 * c + 1] + bb, 0.f);
      }
      Ih1[rr * 33 + k] = s;
      if (k == 0) cnt[rr] = (float)(c1 - c0);
    }
    __syncthreads();
    // ---- phi.2 on the summed hidden: pooled = W2 h1s + count * b2
    {
      const float b2 = bs[LB_PHI2 + 16 * cb + j16];
      f32x4 init;
#pragma unroll
      for (int e = 0; e < 4; ++e) init[e] = cnt[16 * rb + 4 * q16 + e] * b2;
      const f32x4 c = wave_mma16<16, 33, 1, 1, 17>(Ih1 + 16 * rb * 33, lds + LW_PHI2 + 16 * cb * 17, lane, init);
      store_block16<33>(Ipool, 16 * rb, 16 * cb, c, lane);
    }
    __syncthreads();
    // ---- rho.0 (32 -> 128, ReLU)
    {
      f32x16 acc[1];
      acc[0] = splat(bs[LB_RHO0 + wave * 32 + r]);
      wave_mma<1, 32, 33, 1, 1, 33>(acc, Ipool, lds + LW_RHO0 + wave * 32 * 33, lane);
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[0][i] = fmaxf(acc[0][i], 0.f);
      store_tile<129>(Ir1, wave * 32, acc[0], lane);
    }
    __syncthreads();
    // ---- rho.2 (128 -> 64)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int col0 = 32 * cb + 16 * t;
      const f32x4 c = wave_mma16<128, 129, 1, 1, 129>(Ir1 + 16 * rb * 129, lds + LW_RHO2 + col0 * 129, lane,
                                                      bs[LB_RHO2 + col0 + j16]);
      store_block16<65>(Ir2, 16 * rb, col0, c, lane);
    }
    __syncthreads();
    // ---- heads (64 -> 2g)
    {
      const f32x4 c = wave_mma16<64, 65, 1, 1, 65>(Ir2 + 16 * rb * 65, lds + LW_HEAD + 16 * cb * 65, lane,
                                                   bs[LB_HEAD + 16 * cb + j16]);
      store_block16<33>(Iml, 16 * rb, 16 * cb, c, lane);
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = u * 256 + tid, rr = idx >> 5, c = idx & 31;
      if (row0 + rr < g.B && c < 2 * gd) a.out[(long)(row0 + rr) * a.ldo + c] = cnt[rr] > 0.f ? Iml[rr * 33 + c] : 0.f;
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------ inference: decode
// DeepSetDecoder.forward (algorithm.py:68-73) on rows (lane b, location n): z_b = mean + eps * exp(logvar / 2)
// (Encoder.decode :121-127; eps == null: z = mean), decoder(z_b), loc embedding, func -> speed in (0, 1).
struct DecArgs {
  const float* w[NLAYER];
  const float* b[NLAYER];
  const float *goal, *eps, *loc;
  float* out;
  long rows;               // B * N
  int ldg, ldl, ldo, N, g;
};
constexpr int DW_DEC0 = 0;                         // [128][17]
constexpr int DW_DEC2 = DW_DEC0 + 128 * 17;        // [32][129]
constexpr int DW_F0 = DW_DEC2 + 32 * 129;          // [128][65]
constexpr int DW_F2 = DW_F0 + 128 * 65;            // [32][129]
constexpr int DW_SMALL = DW_F2 + 32 * 129;         // bdec0[128] bdec2[32] bf0[128] bf2[32] w4[32] wloc[32] bloc[32] b4[1]
constexpr int DI_Z = DW_SMALL + 420;               // [32][17]
constexpr int DI_D1 = DI_Z + 32 * 17;              // [32][129]
constexpr int DI_CAT = DI_D1 + 32 * 129;           // [32][65]
constexpr int DI_F1 = DI_CAT + 32 * 65;            // [32][129]
constexpr int DI_F2 = DI_F1 + 32 * 129;            // [32][33]
constexpr int DEC_LDS_FLOATS = DI_F2 + 32 * 33;

__global__ __launch_bounds__(256, 1) void vae_decode_kernel(DecArgs g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31;
  const int rb = wave & 1, cb = wave >> 1, j16 = lane & 15;
  const int gd = g.g;
  float* sm = lds + DW_SMALL;
  float *bdec0 = sm, *bdec2 = sm + 128, *bf0 = sm + 160, *bf2 = sm + 288, *w4s = sm + 320, *wlocs = sm + 352, *blocs = sm + 384,
        *b4s = sm + 416;
  float* Iz = lds + DI_Z;
  float* Id1 = lds + DI_D1;
  float* Icat = lds + DI_CAT;
  float* If1 = lds + DI_F1;
  float* If2 = lds + DI_F2;
  {
    float4 f0[8], d2[4], f2[4];
    float d0[8];
    request_w4<128, 64>(f0, g.w[FUNC0], tid);
    request_w4<32, 128>(d2, g.w[DEC2], tid);
    request_w4<32, 128>(f2, g.w[FUNC2], tid);
    const int nd = 128 * gd, lg = __ffs(gd) - 1;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = u * 256 + tid;
      d0[u] = e < nd ? g.w[DEC0][e] : 0.f;
    }
    const float s0 = tid < 128 ? g.b[DEC0][tid] : g.b[FUNC0][tid - 128];
    float s1 = 0.f;
    if (tid < 32) s1 = g.b[DEC2][tid];
    else if (tid < 64) s1 = g.b[FUNC2][tid - 32];
    else if (tid < 96) s1 = g.w[FUNC4][tid - 64];
    else if (tid < 128) s1 = g.w[LOC][tid - 96];
    else if (tid < 160) s1 = g.b[LOC][tid - 128];
    else if (tid == 160) s1 = g.b[FUNC4][0];
    for (int i = tid; i < 128 * 17; i += 256)
      if (i % 17 >= gd) lds[DW_DEC0 + i] = 0.f;
    commit_w4<128, 64>(lds + DW_F0, 65, f0, tid);
    commit_w4<32, 128>(lds + DW_DEC2, 129, d2, tid);
    commit_w4<32, 128>(lds + DW_F2, 129, f2, tid);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = u * 256 + tid;
      if (e < nd) lds[DW_DEC0 + (e >> lg) * 17 + (e & (gd - 1))] = d0[u];
    }
    if (tid < 128) bdec0[tid] = s0; else bf0[tid - 128] = s0;
    if (tid < 32) bdec2[tid] = s1;
    else if (tid < 64) bf2[tid - 32] = s1;
    else if (tid < 96) w4s[tid - 64] = s1;
    else if (tid < 128) wlocs[tid - 96] = s1;
    else if (tid < 160) blocs[tid - 128] = s1;
    else if (tid == 160) b4s[0] = s1;
  }
  __syncthreads();
  for (long row0 = (long)blockIdx.x * 32; row0 < g.rows; row0 += (long)gridDim.x * 32) {
    // ---- z of every row's lane, zero-padded to 16 columns
    for (int idx = tid; idx < 32 * 16; idx += 256) {
      const int rr = idx >> 4, j = idx & 15;
      const long row = row0 + rr;
      float zv = 0.f;
      if (j < gd && row < g.rows) {
        const long b = row / g.N;
        const float mean = g.goal[b * g.ldg + j];
        zv = g.eps ? mean + g.eps[b * gd + j] * expf(0.5f * g.goal[b * g.ldg + gd + j]) : mean;
      }
      Iz[rr * 17 + j] = zv;
    }
    __syncthreads();
    // ---- decoder.0 (g -> 128, ReLU)
    {
      f32x16 acc[1];
      acc[0] = splat(bdec0[wave * 32 + r]);
      wave_mma<1, 16, 17, 1, 1, 17>(acc, Iz, lds + DW_DEC0 + wave * 32 * 17, lane);
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[0][i] = fmaxf(acc[0][i], 0.f);
      store_tile<129>(Id1, wave * 32, acc[0], lane);
    }
    __syncthreads();
    // ---- decoder.2 (128 -> 32) into cat[:, 0:32]; loc embedding into cat[:, 32:64]
    {
      const f32x4 c = wave_mma16<128, 129, 1, 1, 129>(Id1 + 16 * rb * 129, lds + DW_DEC2 + 16 * cb * 129, lane, bdec2[16 * cb + j16]);
      store_block16<65>(Icat, 16 * rb, 16 * cb, c, lane);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = u * 256 + tid, rr = idx >> 5, k = idx & 31;
        const long row = row0 + rr;
        const float loc = row < g.rows ? g.loc[(row / g.N) * g.ldl + (row % g.N)] : 0.f;
        Icat[rr * 65 + 32 + k] = loc * wlocs[k] + blocs[k];
      }
    }
    __syncthreads();
    // ---- func.0 (64 -> 128, ReLU)
    {
      f32x16 acc[1];
      acc[0] = splat(bf0[wave * 32 + r]);
      wave_mma<1, 64, 65, 1, 1, 65>(acc, Icat, lds + DW_F0 + wave * 32 * 65, lane);
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[0][i] = fmaxf(acc[0][i], 0.f);
      store_tile<129>(If1, wave * 32, acc[0], lane);
    }
    __syncthreads();
    // ---- func.2 (128 -> 32, ReLU)
    {
      f32x4 c = wave_mma16<128, 129, 1, 1, 129>(If1 + 16 * rb * 129, lds + DW_F2 + 16 * cb * 129, lane, bf2[16 * cb + j16]);
#pragma unroll
      for (int e = 0; e < 4; ++e) c[e] = fmaxf(c[e], 0.f);
      store_block16<33>(If2, 16 * rb, 16 * cb, c, lane);
    }
    __syncthreads();
    // ---- func.4 (32 -> 1) + sigmoid
    if (tid < 32) {
      const long row = row0 + tid;
      if (row < g.rows) {
        float s = b4s[0];
#pragma unroll 8
        for (int k = 0; k < 32; ++k) s += w4s[k] * If2[tid * 33 + k];
        g.out[(row / g.N) * g.ldo + (row % g.N)] = 1.f / (1.f + expf(-s));
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------ slab sums
// Column `col` of the per-workgroup gradient slabs, rows q, q+4, q+8, ... of n, in a fixed order: eight
// independent running sums keep eight slab rows in flight per thread.
__device__ __forceinline__ float slab_quarter_sum(const float* __restrict__ col, long total, int n, int q) {
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f, s5 = 0.f, s6 = 0.f, s7 = 0.f;
  int k = q;
  for (; k + 28 < n; k += 32) {
    const float v0 = col[(long)k * total], v1 = col[(long)(k + 4) * total], v2 = col[(long)(k + 8) * total],
                v3 = col[(long)(k + 12) * total], v4 = col[(long)(k + 16) * total], v5 = col[(long)(k + 20) * total],
                v6 = col[(long)(k + 24) * total], v7 = col[(long)(k + 28) * total];
    s0 += v0; s1 += v1; s2 += v2; s3 += v3; s4 += v4; s5 += v5; s6 += v6; s7 += v7;
  }
  for (; k < n; k += 4) s0 += col[(long)k * total];
  return ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7));
}

// The gradient of the per-car layers (flat range [lo, hi)) is complete when the per-car kernel ends.  When
// the per-lane backward kernel leaves CUs idle (one tile per workgroup, fewer than 256 tiles), extra
// workgroups of THAT launch sum the per-car slabs (and apply Adam to those parameters) beside it, and the
// reduction kernel at the end of the step only has the per-lane layers left.
struct CarReduce {
  const float* slab;
  long total, lo, hi;                 // columns [lo, hi) without [hole_lo, hole_hi): func.0's matrix waits for the per-lane slabs
  long hole_lo, hole_hi;
  float* svec;                        // [128] out: s[o] (slab position hole_lo + o*64 + 32) summed over the slabs
  float *grads, *params, *m, *v;      // params == nullptr: gradients only
  AdamK k;
  int n_car, first_block, n_blocks;   // workgroups first_block .. first_block + n_blocks - 1 do this work; n_blocks 0: off
};

// 512 threads: two groups of 64 columns, four waves (slab quarters) each; `red` = 512 floats of LDS
__device__ __forceinline__ void car_reduce_block(const CarReduce& R, int blk, int tid, float* red) {
  const int group = tid >> 8, q = (tid >> 6) & 3, p = tid & 63;
  const long before = (R.hole_lo - R.lo + 63) / 64;             // groups in front of the hole
  const long ncols = before + (R.hi - R.hole_hi + 63) / 64;     // groups of parameter columns
  const long ngroups = ncols + 2;                               // and two groups for the 128 entries of s
  for (long base = 2L * blk; base < ngroups; base += 2L * R.n_blocks) {
    const long gi64 = base + group;
    const bool sgrp = gi64 >= ncols;
    const long i = sgrp ? R.hole_lo + ((gi64 - ncols) * 64 + p) * 64 + 32
                        : (gi64 < before ? R.lo + gi64 * 64 + p : R.hole_hi + (gi64 - before) * 64 + p);
    const bool live = sgrp ? gi64 < ngroups : (gi64 < before ? i < R.hole_lo : i < R.hi);
    red[(group * 4 + q) * 64 + p] = live ? slab_quarter_sum(R.slab + i, R.total, R.n_car, q) : 0.f;
    __syncthreads();
    if (q == 0 && live) {
      const float* rg = red + group * 256;
      const float gsum = (rg[p] + rg[64 + p]) + (rg[128 + p] + rg[192 + p]);
      if (sgrp) R.svec[(gi64 - ncols) * 64 + p] = gsum;
      else R.grads[i] = gsum;
      if (!sgrp && R.params != nullptr) {
        float pi = R.params[i], mi = R.m[i], vi = R.v[i];
        adam_one(pi, gsum, mi, vi, R.k);
        R.params[i] = pi;
        R.m[i] = mi;
        R.v[i] = vi;
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------ backward
// One 16x16 output block as mt::wave_mma16 computes it (same two accumulation chains, same order), with the B
// operand already in registers: b[s] = B(4 s + lane / 16, lane % 16).
template <int K, int ASI, int ASK>
__device__ __forceinline__ f32x4 wave_mma16_breg(const float* a, const float (&b)[K / 4], int lane, float init) {
  static_assert(K % 8 == 0, "K must be a multiple of 8");
  const float* ap = a + (lane & 15) * ASI + (lane >> 4) * ASK;
  f32x4 c0 = {init, init, init, init}, c1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < K; k += 8) {
    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[k * ASK], b[k / 4], c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[(k + 4) * ASK], b[k / 4 + 1], c1, 0, 0, 0);
  }
  return c0 + c1;
}

// SINGLE: the grid covers every tile (one tile per workgroup, the case up to 8192 traffic lanes); a layer's
// slab entries are then written right after the phase that completes them, under the later phases.
// NT = 256, or 512 (with SINGLE only: the multi-tile form needs the full register file per wave): four more
// waves share the copies into LDS and the element-wise phases; the matrix phases stay on waves 0..3.
template <bool SINGLE, int NT>
__global__ __launch_bounds__(NT, 1) void vae_lane_backward_kernel(LaneArgs g, CarReduce cr) {
  static_assert(NT == 256 || (NT == 512 && SINGLE), "helper waves only in the one-tile-per-workgroup form");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (NT == 512 && cr.n_blocks > 0 && (int)blockIdx.x >= cr.first_block) {      // whole workgroups: no barrier is split
    car_reduce_block(cr, blockIdx.x - cr.first_block, tid, lds);
    return;
  }
  const int r = lane & 31, h = lane >> 5;
  const int rb = wave & 1, cb = wave >> 1;
  const int gd = g.g;
  float* Ih1 = lds + LI_H1;
  float* Ipool = lds + LI_POOL;
  float* Ir1 = lds + LI_R1;     // S (df1 summed per lane) until dd2 and dA are formed, then r1, then dr1
  float* Ir2 = lds + LI_R2;     // r2, then dr2, then scratch of the phi.0 backward
  float* Iml = lds + LI_ML;     // mean|logvar, then dpooled
  float* Iz = lds + LI_Z;
  float* Id1 = lds + LI_D1;     // d1, then dd1
  float* Id2 = lds + LI_D2;     // dd2
  float* Idml = lds + LI_DML;
  float* Itmp = lds + LI_TMP;   // d2 (forward value) until dA is formed, then dz, then t = dpooled * W_phi2
  const bool staged = g.N <= STAGE_MAX_CARS;
  const bool mat = wave < 4;                      // this wave takes part in the matrix phases
  HHRL_STAMP(2, 0);
  // phi.0 of hidden unit tid & 15 (every element this thread handles in the phi.0 backward uses that unit)
  const float wa = g.w[PHI0][2 * (tid & 15)], wb = g.w[PHI0][2 * (tid & 15) + 1], bb = g.b[PHI0][tid & 15];

  // weight-gradient tiles owned by this wave, kept across tiles
  f32x16 aDec2 = splat(0.f), aDec0 = splat(0.f), aHead = splat(0.f), aRho2a = splat(0.f), aRho2b = splat(0.f);
  f32x16 aRho0 = splat(0.f), aPhi2 = splat(0.f), aF0 = splat(0.f);
  // bias gradients: one thread owns a column of a layer
  float bDec2 = 0.f, bDec0 = 0.f, bHead = 0.f, bRho2 = 0.f, bRho0 = 0.f, bPhi2 = 0.f;
  // With helper waves the column sums belong to their threads (index bt), so that no matrix wave carries
  // them on top of its products; without, to the threads that had them before.
  const int bt = NT == 512 ? tid - 256 : tid;
  const int bt_dec0 = NT == 512 ? bt : tid - 128;
  const int bt_phi2 = NT == 512 ? bt : (wave == 1 ? lane : -1);
  auto owns = [](int i, int n) { return i >= 0 && i < n; };
  float gPhi0a = 0.f, gPhi0b = 0.f, gPhi0c = 0.f;     // threads 0..47: dW_phi0[k][0], dW_phi0[k][1], db_phi0[k]
  float klacc = 0.f;

  float* slab = g.slab + (long)blockIdx.x * g.off.total;
  auto put_tile = [&](long base, int ld, int row_lim, int col0, int col_lim, const f32x16& t) {
    const int col = col0 + r;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = drow(i, h);
      if (row < row_lim && col < col_lim) slab_put(&slab[base + (long)row * ld + col], t[i]);
    }
  };

  auto write_f0 = [&]() {     // columns 0..31 of func.0's rows 32 wave .. 32 wave + 31
    if (mat) put_tile(g.off.w[FUNC0] + (long)32 * wave * 64, 64, 32, 0, 32, aF0);
  };
  auto write_dec2 = [&]() {
    if (mat) put_tile(g.off.w[DEC2], 128, 32, 32 * wave, 128, aDec2);
    if (owns(bt, 32)) slab_put(&slab[g.off.b[DEC2] + bt], bDec2);
  };
  auto write_dec0 = [&]() {
    if (mat) put_tile(g.off.w[DEC0] + (long)32 * wave * gd, gd, 32, 0, gd, aDec0);
    if (owns(bt_dec0, 128)) slab_put(&slab[g.off.b[DEC0] + bt_dec0], bDec0);
  };
  auto write_heads = [&]() {
    if (wave < 2) {
      // rows < g belong to to_mean, rows g..2g-1 to to_logvar (g is a power of two: a register quad of four
      // consecutive rows never straddles the two); register quads past row 2g hold nothing
      float* wm = slab + g.off.w[MEAN] + 32 * wave + r;
      float* wl = slab + g.off.w[LOGVAR] + 32 * wave + r;
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        if (8 * qd < 2 * gd) {                           // wave-uniform
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int row = e + 8 * qd + 4 * h;
            float* dst = row < gd ? wm + row * 64 : wl + (row - gd) * 64;
            if (row < 2 * gd) *dst = aHead[4 * qd + e];
          }
        }
      }
    }
    if (owns(bt, gd)) slab_put(&slab[g.off.b[MEAN] + bt], bHead);
    else if (owns(bt, 2 * gd)) slab_put(&slab[g.off.b[LOGVAR] + bt - gd], bHead);
  };
  auto write_rho2 = [&]() {
    if (mat) {
      const int it = wave >> 1, jt0 = (wave & 1) * 2;
      put_tile(g.off.w[RHO2] + (long)32 * it * 128, 128, 32, 32 * jt0, 128, aRho2a);
      put_tile(g.off.w[RHO2] + (long)32 * it * 128, 128, 32, 32 * (jt0 + 1), 128, aRho2b);
    }
    if (owns(bt, 64)) slab_put(&slab[g.off.b[RHO2] + bt], bRho2);
  };
  auto write_rho0 = [&]() {
    if (mat) put_tile(g.off.w[RHO0] + (long)32 * wave * 32, 32, 32, 0, 32, aRho0);
    if (owns(bt, 128)) slab_put(&slab[g.off.b[RHO0] + bt], bRho0);
  };
  auto write_phi2 = [&]() {
    if (wave == 0) put_tile(g.off.w[PHI2], 16, 32, 0, 16, aPhi2);
    if (owns(bt_phi2, 32)) slab_put(&slab[g.off.b[PHI2] + bt_phi2], bPhi2);
  };

  int tile = blockIdx.x;
  do {
    const int row0 = tile * 32;
    const bool first = SINGLE || tile == (int)blockIdx.x;
    // ---- request everything the tile needs in the order of use
    ActRange<0, LN_WIMG_FLOATS, NT> wimg;   // all weight images and biases, as the forward kernel left them
    float cars_r[1024 / NT];
    ActRange<LI_ML, LI_D2, NT> actD;  // mean|logvar, z, d1
    ActRange<LI_R1, LI_ML, NT> actR;  // r1, r2
    ActRange<LI_H1, LI_R1, NT> actP;  // h1, pooled
    const float* act_tile = g.act + (long)tile * LN_ACT_FLOATS;
    actD.request(act_tile, tid);
    if (first) wimg.request_from(g.wimg, tid);
    // the noise of the tile's logvar columns, for the latent phase: elements tid + 256 u of [32][32]
    float eps_r[1024 / NT];
#pragma unroll
    for (int u = 0; u < 1024 / NT; ++u) {
      const int idx = u * NT + tid, rr = idx >> 5, c = idx & 31;
      eps_r[u] = (c >= gd && c < 2 * gd && row0 + rr < g.B) ? g.eps[(long)(row0 + rr) * gd + c - gd] : 0.f;
    }
    // S of a traffic lane = sum of the per-(car tile, lane) rows of the car tiles its cars span, in tile
    // order.  A thread owns one lane and CPT columns of it.  The first two tiles (all there are when
    // N <= 17) are read from clamped addresses without a branch, so the round trips of a thread overlap;
    // longer lanes loop over the rest.
    constexpr int TPR = NT / 32, CPT = 128 / TPR;        // threads per lane row, columns per thread
    // The rows are only requested here; they are added up after every other request of the tile has been
    // issued, so that the tile costs one memory round trip and not two.
    const int s_row = tid / TPR, s_col = 4 * (tid % TPR);
    const unsigned s_n = (unsigned)g.N, s_R = (unsigned)g.R;
    const int s_b = row0 + s_row;
    const unsigned s_lo = (unsigned)s_b * s_n;
    const unsigned s_t0 = s_lo / CAR_TILE, s_t1 = (s_lo + s_n - 1) / CAR_TILE;
    const bool ok0 = s_b < g.B, ok1 = ok0 && s_t1 > s_t0;
    float4 p0[CPT / 4], p1[CPT / 4];
    {
      const long a0 = ((long)s_t0 * s_R + ((unsigned)s_b - (s_t0 * CAR_TILE) / s_n)) * 128;
      const long a1 = ((long)(s_t0 + 1) * s_R + ((unsigned)s_b - ((s_t0 + 1) * CAR_TILE) / s_n)) * 128;
      const float4* q0 = reinterpret_cast<const float4*>(g.spart + (ok0 ? a0 : 0) + s_col);
      const float4* q1 = reinterpret_cast<const float4*>(g.spart + (ok1 ? a1 : 0) + s_col);
#pragma unroll
      for (int u = 0; u < CPT / 4; ++u) {
        p0[u] = q0[u * TPR];
        p1[u] = q1[u * TPR];
      }
    }
    // func.0's columns for d2 as the B operand of dd2 = S A (16x16 blocks, step s takes k = 4 s + lane / 16)
    // (requested here when a workgroup has one tile; the multi-tile form has no registers to spare for
    // it during the copies and asks after them)
    float w0r[32];
    auto request_w0r = [&]() {
      const float* src = g.w[FUNC0] + (lane >> 4) * 64 + 16 * cb + (lane & 15);
#pragma unroll
      for (int s2 = 0; s2 < 32; ++s2) w0r[s2] = src[s2 * 256];
    };
    if (SINGLE && mat) request_w0r();
    ActRange<LI_D2, LI_DML, NT> actO;   // d2
    actO.request(act_tile, tid);
    actR.request(act_tile, tid);
    actP.request(act_tile, tid);
    request_cars<NT>(cars_r, g, row0, tid);
    // ---- and write it to LDS (S takes the slot of r1 until it has been used; r1 and r2 wait in registers)
#pragma unroll
    for (int u = 0; u < CPT / 4; ++u) {
      float4 sum = ok0 ? p0[u] : make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok1) { sum.x += p1[u].x; sum.y += p1[u].y; sum.z += p1[u].z; sum.w += p1[u].w; }
      if (ok1 && s_t1 > s_t0 + 1) {
        for (unsigned t = s_t0 + 2; t <= s_t1; ++t) {
          const float4 e = *reinterpret_cast<const float4*>(g.spart + ((long)t * s_R + ((unsigned)s_b - (t * CAR_TILE) / s_n)) * 128 + s_col + 4 * u * TPR);
          sum.x += e.x; sum.y += e.y; sum.z += e.z; sum.w += e.w;
        }
      }
      float* q = Ir1 + s_row * 129 + s_col + 4 * u * TPR;
      q[0] = sum.x; q[1] = sum.y; q[2] = sum.z; q[3] = sum.w;
    }
    actO.commit_at(Itmp, tid);
    actD.commit(lds, tid);
    if (first) wimg.commit(lds, tid);
    actP.commit(lds, tid);
    commit_cars<NT>(lds, cars_r, g, tid);
    __syncthreads();
    HHRL_STAMP(2, 1);
    // ---- func.0, the part shared by a lane's cars: dd2 = S A (16x16 blocks), dA[128][32] += S^T d2 (row tile `wave`)
    if (mat) {
      if (!SINGLE) request_w0r();
      const int q4 = lane >> 4;
      const float* ap = Ir1 + (16 * rb + (lane & 15)) * 129 + q4;
      f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s2 = 0; s2 < 32; s2 += 2) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[4 * s2], w0r[s2], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[4 * s2 + 4], w0r[s2 + 1], c1, 0, 0, 0);
      }
      store_block16<33>(Id2, 16 * rb, 16 * cb, c0 + c1, lane);
      f32x16 acc[1] = {aF0};
      wave_mma<1, 32, 1, 129, 33, 1>(acc, Ir1 + 32 * wave, Itmp, lane);
      aF0 = acc[0];
    }
    if (SINGLE) write_f0();
    __syncthreads();
    HHRL_STAMP(2, 2);
    actR.commit(lds, tid);               // r1 over S, r2: first read after the next barriers
    // ---- decoder.2 backward: dW[32][128] += dd2^T d1, db, dd1 = (dd2 W) o relu'(d1) in place (column block `wave`)
    if (mat) {
      f32x16 acc[1] = {aDec2};
      wave_mma<1, 32, 1, 33, 129, 1>(acc, Id2, Id1 + 32 * wave, lane);
      aDec2 = acc[0];
      f32x16 d[1];
      d[0] = splat(0.f);
      wave_mma<1, 32, 33, 1, 129, 1>(d, Id2, lds + LW_DEC2 + 32 * wave, lane);
      wave_sync();
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float* p = Id1 + drow(i, h) * 129 + 32 * wave + r;
        *p = *p > 0.f ? d[0][i] : 0.f;
      }
    }
    if (owns(bt, 32)) bDec2 += colsum32(Id2, 33, bt);
    if (SINGLE) write_dec2();
    __syncthreads();
    HHRL_STAMP(2, 3);
    // ---- decoder.0 backward: dW[128][g] += dd1^T z (row tile `wave`), db, dz = dd1 W (waves 0 and 1, 16 rows each)
    if (mat) {
      f32x16 acc[1] = {aDec0};
      wave_mma<1, 32, 1, 129, 17, 1>(acc, Id1 + 32 * wave, Iz, lane);
      aDec0 = acc[0];
      if (wave < 2) {
        const f32x4 c = wave_mma16<128, 129, 1, 17, 1>(Id1 + 16 * wave * 129, lds + LW_DEC0, lane, 0.f);
        store_block16<33>(Itmp, 16 * wave, 0, c, lane);
      }
    }
    if (owns(bt_dec0, 128)) bDec0 += colsum32(Id1, 129, bt_dec0);
    if (SINGLE) write_dec0();
    __syncthreads();
    HHRL_STAMP(2, 4);
    // ---- latent gradient (KL + reparameterisation): dml = [dz + mean | dz*eps*0.5*std + 0.5*(exp(lv)-1) | 0...]
#pragma unroll
    for (int u = 0; u < 1024 / NT; ++u) {
      const int idx = u * NT + tid, rr = idx >> 5, c = idx & 31;
      float outv = 0.f;
      if (c < 2 * gd && row0 + rr < g.B) {
        const int j = c < gd ? c : c - gd;
        const float mean = Iml[rr * 33 + j], lv = Iml[rr * 33 + gd + j];
        const float dz = Itmp[rr * 33 + j];
        if (c < gd) {
          outv = dz + mean;
          klacc += -0.5f * (1.f + lv - mean * mean - expf(lv));
        } else {
          outv = dz * eps_r[u] * 0.5f * expf(0.5f * lv) + 0.5f * (expf(lv) - 1.f);
        }
      }
      Idml[rr * 33 + c] = outv;
    }
    __syncthreads();
    HHRL_STAMP(2, 5);
    // ---- heads backward: dW[2g][64] += dml^T r2 (waves 0,1), dr2 = dml W (waves 2,3; replaces r2 afterwards)
    {
      f32x16 d[1];
      d[0] = splat(0.f);
      if (wave < 2) {
        f32x16 acc[1] = {aHead};
        wave_mma<1, 32, 1, 33, 65, 1>(acc, Idml, Ir2 + 32 * wave, lane);
        aHead = acc[0];
      } else if (mat) {
        wave_mma<1, 32, 33, 1, 65, 1>(d, Idml, lds + LW_HEAD + 32 * (wave - 2), lane);
      }
      if (owns(bt, 32)) bHead += colsum32(Idml, 33, bt);
      __syncthreads();
      if (wave >= 2 && mat) store_tile<65>(Ir2, 32 * (wave - 2), d[0], lane);
    }
    if (SINGLE) write_heads();
    __syncthreads();
    HHRL_STAMP(2, 6);
    // ---- rho.2 backward: dW[64][128] += dr2^T r1 (two tiles per wave), db, dr1 = (dr2 W) o relu'(r1) in place
    {
      f32x16 d[1];
      d[0] = splat(0.f);
      if (mat) {
        const int it = wave >> 1, jt0 = (wave & 1) * 2;
        f32x16 acc[2] = {aRho2a, aRho2b};
        wave_mma<2, 32, 1, 65, 129, 1>(acc, Ir2 + 32 * it, Ir1 + 32 * jt0, lane);
        aRho2a = acc[0];
        aRho2b = acc[1];
        wave_mma<1, 64, 65, 1, 129, 1>(d, Ir2, lds + LW_RHO2 + 32 * wave, lane);
      }
      if (owns(bt, 64)) bRho2 += colsum32(Ir2, 65, bt);
      __syncthreads();                                  // every wave is done reading r1
      if (mat) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          float* p = Ir1 + drow(i, h) * 129 + 32 * wave + r;
          *p = *p > 0.f ? d[0][i] : 0.f;
        }
      }
    }
    if (SINGLE) write_rho2();
    __syncthreads();
    HHRL_STAMP(2, 7);
    // ---- rho.0 backward: dW[128][32] += dr1^T pooled (row tile `wave`), db, dpooled = dr1 W (16x16 blocks) -> ml slot
    if (mat) {
      f32x16 acc[1] = {aRho0};
      wave_mma<1, 32, 1, 129, 33, 1>(acc, Ir1 + 32 * wave, Ipool, lane);
      aRho0 = acc[0];
      const f32x4 c = wave_mma16<128, 129, 1, 33, 1>(Ir1 + 16 * rb * 129, lds + LW_RHO0 + 16 * cb, lane, 0.f);
      store_block16<33>(Iml, 16 * rb, 16 * cb, c, lane);
    }
    if (owns(bt, 128)) bRho0 += colsum32(Ir1, 129, bt);
    if (SINGLE) write_rho0();
    __syncthreads();
    HHRL_STAMP(2, 8);
    // ---- phi.2 backward on the summed hidden: dW[32][16] += dpooled^T h1s (wave 0), db = N colsum(dpooled) (wave 1),
    //      t = dpooled W_phi2 (waves 2 and 3): gradient at the hidden layer of every car of the lane, before its ReLU mask
    if (wave == 0) {
      f32x16 acc[1] = {aPhi2};
      wave_mma<1, 32, 1, 33, 33, 1>(acc, Iml, Ih1, lane);       // columns >= 16 are junk, dropped below
      aPhi2 = acc[0];
    } else if (wave == 2 || wave == 3) {
      const f32x4 c = wave_mma16<32, 33, 1, 17, 1>(Iml + 16 * (wave - 2) * 33, lds + LW_PHI2, lane, 0.f);
      store_block16<33>(Itmp, 16 * (wave - 2), 0, c, lane);
    }
    if (owns(bt_phi2, 32)) bPhi2 += (float)g.N * colsum32(Iml, 33, bt_phi2);
    if (SINGLE) write_phi2();
    __syncthreads();
    HHRL_STAMP(2, 9);
    // ---- phi.0 backward: per (lane, hidden unit) walk the cars again and apply the ReLU mask
    {
      float* red = Ir2;       // [32][49] partial sums
      for (int idx = tid; idx < 32 * 16; idx += NT) {
        const int rr = idx >> 4, k = idx & 15;
        const int b = row0 + rr;
        float a0 = 0.f, a1 = 0.f, c = 0.f;
        if (b < g.B) {
          auto walk = [&](const float* x) {
            for (int n = 0; n < g.N; ++n) {
              const float x0 = x[2 * n], x1 = x[2 * n + 1];
              if (wa * x0 + wb * x1 + bb > 0.f) { a0 += x0; a1 += x1; c += 1.f; }
            }
          };
          if (staged) walk(lds + LI_CARS + rr * g.N * 2);
          else walk(g.data + (long)b * g.N * 2);
        }
        const float t = Itmp[rr * 33 + k];
        red[rr * 49 + k] = t * a0;
        red[rr * 49 + 16 + k] = t * a1;
        red[rr * 49 + 32 + k] = t * c;
      }
      __syncthreads();
      if (tid < 48) {
        float s = 0.f;
        for (int rr = 0; rr < 32; ++rr) s += red[rr * 49 + tid];
        if (tid < 16) gPhi0a += s; else if (tid < 32) gPhi0b += s; else gPhi0c += s;
      }
    }
    __syncthreads();
    HHRL_STAMP(2, 10);
  } while (!SINGLE && (tile += gridDim.x) < g.tiles);

  if (!SINGLE) {
    write_f0();
    write_dec2();
    write_dec0();
    write_heads();
    write_rho2();
    write_rho0();
    write_phi2();
  }
  if (tid < 16) {
    slab_put(&slab[g.off.w[PHI0] + 2 * tid], gPhi0a);
  } else if (tid < 32) {
    slab_put(&slab[g.off.w[PHI0] + 2 * (tid - 16) + 1], gPhi0b);
  } else if (tid < 48) {
    slab_put(&slab[g.off.b[PHI0] + tid - 32], gPhi0c);
  }
  // KL partial of this workgroup
  klacc = wave_sum(klacc);
  __shared__ float klred[NT / 64];
  if (lane == 0) klred[wave] = klacc;
  __syncthreads();
  if (tid == 0) {
    float kl = klred[0];
#pragma unroll
    for (int w = 1; w < NT / 64; ++w) kl += klred[w];
    g.kl_part[blockIdx.x] = kl;
  }
  HHRL_STAMP(2, 11);
}

// grads[i] = sum over slabs, in a fixed order: the same inputs give the same bits.
// A workgroup owns 64 consecutive parameters; its four waves each sum a quarter of the
// slabs (eight independent running sums keep loads in flight), LDS combines the quarters.
// func.0's matrix [128][64] at f0_lo: columns 0..31 of a row are sums over the per-lane slabs, columns
// 32 + j are s[o] wloc[j] + db0[o] bloc[j] with s[o] (slab position o*64 + 32) and db0[o] (position
// b0_off + o) summed over the per-car slabs (or taken from svec and grads when those sums exist already) and
// wloc | bloc as they were at the start of the step (locsave).
__global__ __launch_bounds__(256) void vae_reduce_grads_kernel(const float* __restrict__ slab, long total, int n_car, int n_lane,
                                                               long car_lo, long car_hi, long f0_lo, long b0_off,
                                                               const float* __restrict__ locsave, const float* __restrict__ svec,
                                                               long skip_lo, long skip_mid,
                                                               long skip_back, long skip_hi, float* grads,
                                                               const float* __restrict__ loss_part, const float* __restrict__ kl_part,
                                                               float* loss, float* params, float* m, float* v, AdamK k) {
  __shared__ float red[2][4][64];
  const int p = threadIdx.x & 63, q = threadIdx.x >> 6;
  const long i0 = (long)blockIdx.x * 64;
  // columns already summed beside the per-lane backward kernel: [skip_lo, skip_mid) and [skip_back, skip_hi)
  if ((i0 >= skip_lo && i0 < skip_mid) || (i0 >= skip_back && i0 < skip_hi)) return;
  const long i = i0 + p;
  const long e = i - f0_lo;
  const bool derived = e >= 0 && e < 8192 && (e & 63) >= 32;
  float s = 0.f, s2 = 0.f;
  if (derived) {
    const long o = e >> 6;
    if (svec != nullptr) {                               // both sums were formed beside the per-lane backward kernel
      s = q == 0 ? svec[o] : 0.f;
      s2 = q == 0 ? grads[b0_off + o] : 0.f;
    } else {
      s = slab_quarter_sum(slab + f0_lo + o * 64 + 32, total, n_car, q);
      s2 = slab_quarter_sum(slab + b0_off + o, total, n_car, q);
    }
  } else if (i < total) {
    // per-car layers (loc, func.*) occupy [car_lo, car_hi) and come from the per-car kernel's slabs,
    // except the first 32 columns of func.0's rows
    const bool car = i >= car_lo && i < car_hi && !(e >= 0 && e < 8192);
    s = slab_quarter_sum(slab + i, total, car ? n_car : n_lane, q);
  }
  red[0][q][p] = s;
  red[1][q][p] = s2;
  __syncthreads();
  if (q == 0 && i < total) {
    float gi = (red[0][0][p] + red[0][1][p]) + (red[0][2][p] + red[0][3][p]);
    if (derived) {
      const float di = (red[1][0][p] + red[1][1][p]) + (red[1][2][p] + red[1][3][p]);
      const int j = (int)(e & 31);
      gi = gi * locsave[j] + di * locsave[32 + j];
    }
    grads[i] = gi;
    if (params != nullptr) {                           // fused optimizer step (single device)
      float pi = params[i], mi = m[i], vi = v[i];
      adam_one(pi, gi, mi, vi, k);
      params[i] = pi;
      m[i] = mi;
      v[i] = vi;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < 64) {
    float t = 0.f;
    for (int kk = threadIdx.x; kk < n_car; kk += 64) t += loss_part[kk];
    for (int kk = threadIdx.x; kk < n_lane; kk += 64) t += kl_part[kk];
    t = wave_sum(t);
    if (threadIdx.x == 0) *loss = t;
  }
}


// ===================================================================== the whole step in one launch
// One workgroup (eight waves) owns tiles of 16 traffic lanes TOGETHER WITH their cars and runs the whole
// chain for a tile without leaving the CU: per-lane forward (16-row products on 16x16x4 matrix
// instructions, every wave owning output blocks), the lane's cars (one tile of 16 cars per wave and round,
// func.0 in factored form exactly as in the per-car kernel above), per-lane backward.  What the four-launch
// form passes through global memory stays in LDS or registers here: the saved activations of the lanes,
// U (func.0's pre-activation per lane), S (df1 summed per lane).  The lane weights (86 KB) and the cars' images
// (83 KB) do not fit LDS together, so they share a region: the weights are staged for the forward chain,
// give way to the car phase, and are requested again from L2 under the end of the car phase.
// func.2's weight gradient is formed across the waves (every wave owns two 16x16 blocks of it and walks all
// waves' df2 / f1 images), so there is no per-wave copy of it to combine.  With at most 256 tiles the grid
// covers the batch; beyond that a workgroup loops over tiles and keeps every weight-gradient block in
// registers.  Gradients leave as one slab per workgroup; vae_reduce_grads_kernel sums the slabs in a fixed
// order (and applies Adam): the same inputs give the same bits.
struct StepArgs {
  const float* data;   // [B*N][2]
  const float* eps;    // [B][g]
  const float* params; // flat parameter buffer; layer l's weight at params + ow[l], bias at params + ob[l]
  int ow[NLAYER], ob[NLAYER];
  int total;           // floats per slab (= flat buffer length)
  float* aux;          // out (workgroup 0): wloc[32] bloc[32] as the step found them, at aux + 128
  float* slab;         // [gridDim.x][total]
  float* loss_part;    // [gridDim.x]
  float* kl_part;      // [gridDim.x]
  int B, N, g, tiles, R;   // tiles of ST_ROWS lanes; R = most traffic lanes a car tile can touch
  float recon_scale;
};

__device__ __forceinline__ const float* lane_bias_src_step(const StepArgs& g, int idx) {
  const int gd = g.g;
  const float* p = g.params;
  if (idx < LB_RHO0) return p + g.ob[PHI2] + idx;
  if (idx < LB_RHO2) return p + g.ob[RHO0] + (idx - LB_RHO0);
  if (idx < LB_HEAD) return p + g.ob[RHO2] + (idx - LB_RHO2);
  if (idx < LB_DEC0) {
    const int c = idx - LB_HEAD;
    return c < gd ? p + g.ob[MEAN] + c : (c < 2 * gd ? p + g.ob[LOGVAR] + (c - gd) : nullptr);
  }
  if (idx < LB_DEC2) return p + g.ob[DEC0] + (idx - LB_DEC0);
  if (idx < LB_END) return p + g.ob[DEC2] + (idx - LB_DEC2);
  return nullptr;
}

constexpr int ST_ROWS = 16, ST_NT = 512, ST_WAVES = 8;
// LDS map (floats).  Permanent part:
constexpr int FS_W2 = 0;                            // [32][129] func.2
constexpr int FS_SMALL = FS_W2 + 32 * 129;          // b2[32] w4[32] b4[1]
constexpr int FS_BIAS = FS_SMALL + 96;              // lane biases (LB_* map), then q[128] and p[128]
constexpr int FB_Q = LB_END, FB_P = LB_END + 128, FB_PHI0 = LB_END + 256, FB_ALL = LB_END + 256 + 48;   // phi.0: w[16][2] then b[16]
constexpr int FS_H1 = FS_BIAS + FB_ALL;             // [16][33]  summed hidden of phi (16 columns used)
constexpr int FS_POOL = FS_H1 + 16 * 33;            // [16][33]
constexpr int FS_R1 = FS_POOL + 16 * 33;            // [16][129]
constexpr int FS_R2 = FS_R1 + 16 * 129;             // [16][65]
constexpr int FS_ML = FS_R2 + 16 * 65;              // [16][33]  mean | logvar, later dpooled
constexpr int FS_Z = FS_ML + 16 * 33;               // [16][17]
constexpr int FS_D1 = FS_Z + 16 * 17;               // [16][129]
constexpr int FS_D2 = FS_D1 + 16 * 129;             // [16][33]
constexpr int FS_DML = FS_D2 + 16 * 33;             // [16][33]
constexpr int FS_TMP = FS_DML + 16 * 33;            // [16][33]
constexpr int FS_CARS = FS_TMP + 16 * 33;           // 16 lanes x N x 2 (N <= STAGE_MAX_CARS)
constexpr int FS_OVL = ((FS_CARS + 2 * ST_ROWS * STAGE_MAX_CARS + 3) / 4) * 4;
// Shared part, by phase.  Lane phases: the weight images (LW_* map from FS_OVL, LW_BIAS floats), then a spare
// area.  Car phase: U [16][129] (afterwards S in its place), then one image block per wave.
constexpr int FO_U = 0, FO_WAVE0 = 16 * 129;
constexpr int FO_SPARE = LW_BIAS;                   // behind the weight images: the four partial products of dz
constexpr int FO_END = FO_WAVE0 + ST_WAVES * CAR_WAVE_FLOATS;
static_assert(FS_OVL % 4 == 0 && FO_WAVE0 % 4 == 0 && CAR_WAVE_FLOATS % 4 == 0, "16-byte aligned regions");
static_assert(FO_SPARE + 4 * 16 * 17 <= FO_END, "partial products of dz fit behind the weight images");
static_assert(16 * 128 <= CAR_WAVE_FLOATS, "a wave's image block holds its per-lane sums");
constexpr int ST_LDS_FLOATS = FS_OVL + FO_END;
static_assert(ST_LDS_FLOATS * 4 <= 163840 - 64, "the one-launch step must fit the 160 KiB LDS");

__device__ __forceinline__ float colsum16(const float* img, int ld, int col) {
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int rr = 0; rr < 16; rr += 2) {
    s0 += img[rr * ld + col];
    s1 += img[(rr + 1) * ld + col];
  }
  return s0 + s1;
}

// SINGLE: the grid covers every tile (up to 256 tiles = 4096 traffic lanes): the tile loop runs once and no
// gradient block has to live in a register across the car phase.
// The same product with the B operand as 16-byte pieces of a weight row: lane (j, q) holds
// b[t] = W[row j][16 t + 4 q .. 16 t + 4 q + 3], and step (t, e) of the product takes k = 16 t + 4 q + e from lane
// group q (any one-to-one assignment of k to steps and groups gives the same sum).  A(i, k) = a[i * ASI + k].
template <int K, int ASI>
__device__ __forceinline__ f32x4 wave_mma16_bvec(const float* a, const float4 (&b)[K / 16], int lane, float init) {
  static_assert(K % 16 == 0, "K must be a multiple of 16");
  const float* ap = a + (lane & 15) * ASI + 4 * (lane >> 4);
  f32x4 c0 = {init, init, init, init}, c1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < K / 16; ++t) {
    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[16 * t], b[t].x, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[16 * t + 1], b[t].y, c1, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[16 * t + 2], b[t].z, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[16 * t + 3], b[t].w, c1, 0, 0, 0);
  }
  return c0 + c1;
}

template <bool SINGLE>
__global__ __launch_bounds__(ST_NT, 1) void vae_step_kernel(StepArgs g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int NT = ST_NT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, q = lane >> 4;            // 16x16 blocks: column j, rows 4 q + reg
  const int r32 = lane & 31, h32 = lane >> 5;
  const int gd = g.g;
  float* W2s = lds + FS_W2;
  float* b2s = lds + FS_SMALL;
  float* w4s = b2s + 32;
  float* b4s = w4s + 32;
  float* bs = lds + FS_BIAS;
  float* Ih1 = lds + FS_H1;
  float* Ipool = lds + FS_POOL;
  float* Ir1 = lds + FS_R1;
  float* Ir2 = lds + FS_R2;
  float* Iml = lds + FS_ML;
  float* Iz = lds + FS_Z;
  float* Id1 = lds + FS_D1;
  float* Id2 = lds + FS_D2;
  float* Idml = lds + FS_DML;
  float* Itmp = lds + FS_TMP;
  float* Icars = lds + FS_CARS;
  float* ovl = lds + FS_OVL;                         // weight images (LW_* offsets) or U | wave images
  float* Iu = ovl + FO_U;
  float* wl = ovl + FO_WAVE0 + wave * CAR_WAVE_FLOATS;
  float* f1i = wl + CAR_F1;
  float* f2i = wl + CAR_F2;
  float* dpre = wl + CAR_ROW;
  const bool staged = g.N <= STAGE_MAX_CARS;
  HHRL_STAMP(3, 0);
  HHRL_WG_SPAN(0);

  // ---- requests of the prologue: everything is asked for before the first LDS write
  RegsP<NT> rP;
  RegsR<NT> rR;
  RegsD<NT> rD;
  float rBias[1];
  auto request_weights = [&](int tid) {
    request_w4<128, 32, W4Map<128, 32, NT>::STEPS, NT>(rP.rho0, (g.params + g.ow[RHO0]), tid);
    rP.phi2 = tid < 128 ? reinterpret_cast<const float4*>((g.params + g.ow[PHI2]))[tid] : make_float4(0.f, 0.f, 0.f, 0.f);
    request_w4<64, 128, W4Map<64, 128, NT>::STEPS, NT>(rR.rho2, (g.params + g.ow[RHO2]), tid);
    const int nh = 128 * gd;
#pragma unroll
    for (int u = 0; u < 2048 / NT; ++u) {
      const int e = u * NT + tid;
      rR.head[u] = e < nh ? (e < 64 * gd ? (g.params + g.ow[MEAN])[e] : (g.params + g.ow[LOGVAR])[e - 64 * gd]) : 0.f;
    }
    request_w4<32, 128, W4Map<32, 128, NT>::STEPS, NT>(rD.dec2, (g.params + g.ow[DEC2]), tid);
#pragma unroll
    for (int u = 0; u < 2048 / NT; ++u) {
      const int e = u * NT + tid;
      rD.dec0[u] = e < nh ? (g.params + g.ow[DEC0])[e] : 0.f;
    }
  };
  auto commit_weights = [&](int tid) {
    commit_w4<128, 32, W4Map<128, 32, NT>::STEPS, NT>(ovl + LW_RHO0, 33, rP.rho0, tid);
    if (tid < 128) commit4<16>(ovl + LW_PHI2, 17, tid, rP.phi2);
    commit_w4<64, 128, W4Map<64, 128, NT>::STEPS, NT>(ovl + LW_RHO2, 129, rR.rho2, tid);
    const int nh = 128 * gd, lg = __ffs(gd) - 1;
#pragma unroll
    for (int u = 0; u < 2048 / NT; ++u) {
      const int e = u * NT + tid;
      if (e < nh) ovl[LW_HEAD + (e >> 6) * 65 + (e & 63)] = rR.head[u];
    }
    for (int i = tid; i < 32 * 65; i += NT)
      if (i / 65 >= 2 * gd) ovl[LW_HEAD + i] = 0.f;
    commit_w4<32, 128, W4Map<32, 128, NT>::STEPS, NT>(ovl + LW_DEC2, 129, rD.dec2, tid);
#pragma unroll
    for (int u = 0; u < 2048 / NT; ++u) {
      const int e = u * NT + tid;
      if (e < nh) ovl[LW_DEC0 + (e >> lg) * 17 + (e & (gd - 1))] = rD.dec0[u];
    }
    for (int i = tid; i < 128 * 17; i += NT)
      if (i % 17 >= gd) ovl[LW_DEC0 + i] = 0.f;
  };
  // the tile's car states: 16 lanes x N x 2 floats (N <= STAGE_MAX_CARS), one float per thread
  // (a float outside the tile or the batch is asked for at the batch's first float and replaced by zero when it is
  //  stored: a conditional request would be joined with its alternative right here, and that join waits for the
  //  value, i.e. for every request issued so far)
  auto request_cars16 = [&](int row0) -> float {
    const long base = (long)row0 * g.N * 2, lim = (long)g.B * g.N * 2;
    const int n = staged ? 2 * ST_ROWS * g.N : 0;
    return g.data[(tid < n && base + tid < lim) ? base + tid : 0];
  };
  auto commit_cars16 = [&](int row0, float v) {
    const long base = (long)row0 * g.N * 2, lim = (long)g.B * g.N * 2;
    if (staged && tid < 2 * ST_ROWS * g.N) Icars[tid] = base + tid < lim ? v : 0.f;
  };

  // ---- requests of the prologue.  Before the first barrier only what the first phases read from LDS: phi.0's
  // parameters, the biases, the small vectors of the per-car layers, the tile's cars, wloc and bloc.  Everything
  // that stays in registers or is stored later follows: with one tile per workgroup right after the barrier, where
  // it travels under phi.0 and the phases behind it (issued before the barrier, these ~45 requests per wave made
  // the barrier wait for a memory pipeline that all 256 workgroups had just filled with the same requests).
  const int o_pq = tid - (NT - 128);
  const float loc_v = (g.params + ((tid & 32) ? g.ob[LOC] : g.ow[LOC]))[tid & 31];   // wloc | bloc, stored by threads 0..63
  HHRL_STAMP(3, 44);
  const float phi0_v = tid < 32 ? (g.params + g.ow[PHI0])[tid] : (tid < 48 ? (g.params + g.ob[PHI0])[tid - 32] : 0.f);
  {
    const float* src = lane_bias_src_step(g, tid);
    rBias[0] = src ? *src : 0.f;
  }
  // (func.4's bias through a per-lane address: as a scalar load its arrival would be waited for by the next wait
  //  on the scalar counter)
  HHRL_STAMP(3, 45);
  int lane_zero = 0;
  asm volatile("" : "+v"(lane_zero));
  const float p_b2 = (g.params + g.ob[FUNC2])[r32], p_w4 = (g.params + g.ow[FUNC4])[r32], p_b4 = (g.params + g.ob[FUNC4])[lane_zero];
  float car0 = request_cars16(blockIdx.x * ST_ROWS);
  HHRL_STAMP(3, 46);
  // The position part of func.0, by the last 128 threads (output o each): p = E wloc, q = b0 + E bloc with
  // E = W0[:, 32:].  The rows of E are kept in registers, wloc and bloc go through LDS (the slot of the backward's
  // dd2 image, free until then), and the sums are formed under decoder.2, where these waves have nothing else to do.
  float4 pq_e[8];
  float pq_b0 = 0.f;
  float4 v2[2];                                        // func.2's matrix, stored as an LDS image under decoder.2
  // With one tile per workgroup every weight matrix of the lane layers is the B operand of one forward product
  // (output block = 16 rows of the matrix, a lane holds 16-byte pieces of row j) and of one backward product, so
  // the waves ask for their own operands and no image of these matrices is staged in LDS.  Forward: phi.2 on
  // waves 0, 1; rho.0 and decoder.0 one block per wave; rho.2 on waves 0..3, decoder.2 on waves 4, 5 and the heads
  // on waves 6, 7, which share one register array.
  float4 fPhi2[1], fRho0[2], fBig[8];
  float fDec0[4];
  float w0u[8];
  // These requests are spread over the first phases (req_a before the first barrier, req_b right after it, req_c
  // behind phi.2's product, req_d and req_e behind rho.2's): a wave's 16-byte request occupies the CU's one path to
  // the L1 for 16 clocks, the ~250 of them of a workgroup for 3-4 k clocks, and a barrier behind a burst of them
  // waits for the burst.  They are issued by all waves alike where that is possible: a request that only some waves
  // issue makes every later wait count as if none had.
  auto req_a = [&]() {                                   // phi.2 and rho.0 operands
    if (SINGLE) {
      if (wave < 2) fPhi2[0] = *reinterpret_cast<const float4*>((g.params + g.ow[PHI2]) + (16 * wave + j) * 16 + 4 * q);
      const float* src = (g.params + g.ow[RHO0]) + (16 * wave + j) * 32 + 4 * q;
#pragma unroll
      for (int t = 0; t < 2; ++t) fRho0[t] = *reinterpret_cast<const float4*>(src + 16 * t);
    }
  };
  auto req_b = [&]() {
    if (SINGLE) {
      // rows of rho.2 (waves 0..3), decoder.2 (waves 4, 5) or the heads (waves 6, 7: to_mean rows, then to_logvar
      // rows; a block row beyond 2 g reads the first row, and its output column is set to zero)
      const int hr = 16 * (wave - 6) + j;
      const int at = wave < 4 ? g.ow[RHO2] + (16 * wave + j) * 128
                   : wave < 6 ? g.ow[DEC2] + (16 * (wave - 4) + j) * 128
                   : hr < gd ? g.ow[MEAN] + hr * 64 : g.ow[LOGVAR] + (hr < 2 * gd ? (hr - gd) * 64 : 0);
      const float* src = g.params + at + 4 * q;
#pragma unroll
      for (int t = 0; t < 4; ++t) fBig[t] = *reinterpret_cast<const float4*>(src + 16 * t);
      if (wave < 6) {
#pragma unroll
        for (int t = 4; t < 8; ++t) fBig[t] = *reinterpret_cast<const float4*>(src + 16 * t);
      }
    }
  };
  auto req_c = [&]() {
    if (SINGLE) {
      // decoder.0: W[row][k], k = 4 s + q < g; the other entries meet the zero columns of z
      const float* src = (g.params + g.ow[DEC0]) + (16 * wave + j) * gd;
#pragma unroll
      for (int s = 0; s < 4; ++s) fDec0[s] = src[4 * s + q < gd ? 4 * s + q : 0];
    }
    request_w4<32, 128, 2, NT>(v2, (g.params + g.ow[FUNC2]), tid);
  };
  auto req_d = [&]() {
    // func.0's columns for d2 (A = W0[:, :32]) as the B operand of U = d2 A^T: output column 16 wave + j,
    // step s takes k = 4 s + q
    const float* src = (g.params + g.ow[FUNC0]) + (16 * wave + j) * 64 + q;
#pragma unroll
    for (int s = 0; s < 8; ++s) w0u[s] = src[4 * s];
  };
  auto req_e = [&]() {
    if (o_pq >= 0) {
      const float4* src = reinterpret_cast<const float4*>((g.params + g.ow[FUNC0]) + o_pq * 64 + 32);
#pragma unroll
      for (int e = 0; e < 8; ++e) pq_e[e] = src[e];
      pq_b0 = (g.params + g.ob[FUNC0])[o_pq];
    }
  };
  req_a();
  if (!SINGLE) {
    request_weights(tid);
    req_b(); req_c(); req_d(); req_e();
  }
  HHRL_STAMP(3, 40);
  if (tid < 64) Itmp[tid] = loc_v;
  if (tid < 48) bs[FB_PHI0 + tid] = phi0_v;
  if (tid < LB_END) bs[tid] = rBias[0];
  HHRL_STAMP(3, 41);
  if (!SINGLE) commit_weights(tid);
  if (tid < 32) {
    b2s[tid] = p_b2;
    w4s[tid] = p_w4;
  }
  if (tid == 0) b4s[0] = p_b4;
  if (tid < ST_WAVES) b4s[8 + tid] = 0.f;               // first_lane[] of the car rounds
  commit_cars16(blockIdx.x * ST_ROWS, car0);
  HHRL_STAMP(3, 42);
  lds_barrier();
  if (SINGLE) req_b();
  HHRL_STAMP(3, 1);

  // ---- what a workgroup keeps in registers across its tiles
  // weight-gradient blocks (16x16) of the per-lane layers owned by this wave
  f32x4 aF0[2], aDec2[2], aDec0, aHead, aRho2[4], aRho0[2], aPhi2;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  aF0[0] = aF0[1] = aDec2[0] = aDec2[1] = aDec0 = aHead = aRho0[0] = aRho0[1] = aPhi2 = zero4;
#pragma unroll
  for (int t = 0; t < 4; ++t) aRho2[t] = zero4;
  float bDec2 = 0.f, bDec0 = 0.f, bHead = 0.f, bRho2 = 0.f, bRho0 = 0.f, bPhi2 = 0.f;
  float gPhi0 = 0.f;                                  // threads 0..47: dW_phi0[k][0], dW_phi0[k][1], db_phi0[k]
  float klacc = 0.f;
  // per-car layers: func.2's weight gradient (two blocks per wave), column sums of df1, func.4, the loss
  f32x4 dW2a[2];
  dW2a[0] = dW2a[1] = zero4;
  float db0[8], sacc[8];
#pragma unroll
  for (int blk = 0; blk < 8; ++blk) db0[blk] = sacc[blk] = 0.f;
  float w4acc = 0.f, db2acc = 0.f, b4acc = 0.f, lossacc = 0.f;

  // With one tile per workgroup (SINGLE) a layer's slab entries are written right after the phase that completes
  // them, under the later phases, and no gradient block lives in a register longer than its phase.
  float* slab = g.slab + (long)blockIdx.x * g.total;
  auto put16 = [&](long base, int ld, int row0, int row_lim, int col0, int col_lim, const f32x4& t) {
    const int col = col0 + j;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = row0 + 4 * q + e;
      if (row < row_lim && col < col_lim) slab_put(&slab[base + (long)row * ld + col], t[e]);
    }
  };
  // The weight-gradient products below are formed with their operands in each other's place: a lane then holds four
  // consecutive columns of one row of dW (row row0 + j, columns col0 + 4 q ..) and hands them over with one 16-byte
  // store, a quarter of the store instructions of the row-per-register layout.
  auto put16t = [&](long base, int ld, int row0, int col0, const f32x4& t) {
    store_through(reinterpret_cast<float4*>(&slab[base + (long)(row0 + j) * ld + col0 + 4 * q]), make_float4(t[0], t[1], t[2], t[3]));
  };
  const int bt = tid - 256;                            // owner index of the bias-gradient column sums
  auto write_f0 = [&]() {
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) put16t(g.ow[FUNC0], 64, 16 * wave, 16 * cb, aF0[cb]);
  };
  auto write_dec2 = [&]() {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) put16t(g.ow[DEC2], 128, 16 * rb, 16 * wave, aDec2[rb]);
    if (bt >= 0 && bt < 32) slab_put(&slab[g.ob[DEC2] + bt], bDec2);
  };
  auto write_dec0 = [&]() {
    put16(g.ow[DEC0], gd, 16 * wave, 128, 0, gd, aDec0);
    if (bt >= 0 && bt < 128) slab_put(&slab[g.ob[DEC0] + bt], bDec0);
  };
  auto write_heads = [&]() {
    // rows < g belong to to_mean, rows g..2g-1 to to_logvar
    const int row = 16 * (wave >> 2) + j, col = 16 * (wave & 3) + 4 * q;
    const float4 t = make_float4(aHead[0], aHead[1], aHead[2], aHead[3]);
    if (row < gd) store_through(reinterpret_cast<float4*>(&slab[g.ow[MEAN] + (long)row * 64 + col]), t);
    else if (row < 2 * gd) store_through(reinterpret_cast<float4*>(&slab[g.ow[LOGVAR] + (long)(row - gd) * 64 + col]), t);
    if (bt >= 0 && bt < gd) slab_put(&slab[g.ob[MEAN] + bt], bHead);
    else if (bt >= 0 && bt < 2 * gd) slab_put(&slab[g.ob[LOGVAR] + bt - gd], bHead);
  };
  auto write_rho2_lo = [&]() {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) put16t(g.ow[RHO2], 128, 16 * rb, 16 * wave, aRho2[rb]);
    if (bt >= 0 && bt < 64) slab_put(&slab[g.ob[RHO2] + bt], bRho2);
  };
  auto write_rho2_hi = [&]() {
#pragma unroll
    for (int rb = 2; rb < 4; ++rb) put16t(g.ow[RHO2], 128, 16 * rb, 16 * wave, aRho2[rb]);
  };
  auto write_rho2 = [&]() {
    write_rho2_lo();
    write_rho2_hi();
  };
  auto write_rho0 = [&]() {
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) put16t(g.ow[RHO0], 32, 16 * wave, 16 * cb, aRho0[cb]);
    if (bt >= 0 && bt < 128) slab_put(&slab[g.ob[RHO0] + bt], bRho0);
  };
  auto write_phi2 = [&]() {
    if (wave < 2) put16t(g.ow[PHI2], 16, 16 * wave, 0, aPhi2);
    if (bt >= 0 && bt < 32) slab_put(&slab[g.ob[PHI2] + bt], bPhi2);
  };
  auto write_phi0 = [&]() {
    if (tid < 16) slab_put(&slab[g.ow[PHI0] + 2 * tid], gPhi0);
    else if (tid < 32) slab_put(&slab[g.ow[PHI0] + 2 * (tid - 16) + 1], gPhi0);
    else if (tid < 48) slab_put(&slab[g.ob[PHI0] + tid - 32], gPhi0);
  };
  auto write_func2 = [&]() {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) put16t(g.ow[FUNC2], 128, 16 * rb, 16 * wave, dW2a[rb]);
  };

  // The waves' small sums meet in LDS, Q1 floats per wave: db0[128] s[128] db2[32] w4[32] b4 loss kl.  With one tile
  // per workgroup they are final when the car rounds end and go there at once (into the part of each wave's image
  // block that held its func.2 activations, which nothing writes afterwards), so that they hold no registers during
  // the lane backward; otherwise they are parked at the end, over the dead images.
  constexpr int Q1 = 328;
  constexpr int B_B0 = 0, B_S = 128, B_B2 = 256, B_W4 = 288, B_B4 = 320, B_LOSS = 321, B_KL = 322;
  static_assert(Q1 <= 16 * 33, "a wave's sums fit its func.2 image");
  static_assert(FO_WAVE0 + (ST_WAVES - 1) * CAR_WAVE_FLOATS + CAR_F2 >= FO_SPARE + 4 * 16 * 17,
                "the last wave's parked sums lie behind the partial products of dz");
  static_assert(FO_WAVE0 + (ST_WAVES - 2) * CAR_WAVE_FLOATS + CAR_F2 + Q1 <= FO_SPARE,
                "the other waves' parked sums lie before the partial products of dz");
  constexpr int QS = SINGLE ? CAR_WAVE_FLOATS : Q1;
  float* cmb = SINGLE ? ovl + FO_WAVE0 + CAR_F2 : ovl;
  auto park_car_sums = [&]() {
#pragma unroll
    for (int blk = 0; blk < 8; ++blk) {
      db0[blk] += __shfl_xor(db0[blk], 16, 64);
      db0[blk] += __shfl_xor(db0[blk], 32, 64);
      sacc[blk] += __shfl_xor(sacc[blk], 16, 64);
      sacc[blk] += __shfl_xor(sacc[blk], 32, 64);
    }
    w4acc += __shfl_xor(w4acc, 32, 64);
    db2acc += __shfl_xor(db2acc, 32, 64);
    b4acc = wave_sum(b4acc);
    lossacc = wave_sum(lossacc);
    float* mine = cmb + wave * QS;
    if (q == 0) {
#pragma unroll
      for (int blk = 0; blk < 8; ++blk) {
        mine[B_B0 + 16 * blk + j] = db0[blk];
        mine[B_S + 16 * blk + j] = sacc[blk];
      }
    }
    if (h32 == 0) {
      mine[B_B2 + r32] = db2acc;
      mine[B_W4 + r32] = w4acc;
    }
    if (lane == 0) {
      mine[B_B4] = b4acc;
      mine[B_LOSS] = lossacc;
    }
  };
  auto park_kl = [&]() {
    klacc = wave_sum(klacc);
    if (lane == 0) cmb[wave * QS + B_KL] = klacc;
  };

  int tile = blockIdx.x;
  do {
    const int row0 = tile * ST_ROWS;
    const bool first = SINGLE || tile == (int)blockIdx.x;
    const int nlanes = min(ST_ROWS, g.B - row0);
    const int ncars = nlanes * g.N;                    // cars of this tile: rows car_base .. car_base + ncars - 1
    const long car_base = (long)row0 * g.N;
    const int ntiles = (ncars + CAR_TILE - 1) / CAR_TILE;
    if (!first) {                                      // the previous tile's images are dead (barrier at loop end)
      car0 = request_cars16(row0);
      commit_cars16(row0, car0);
      lds_barrier();
    }
    // the tile's reparameterisation noise: one element of [16][32] per thread (column c: mean part c < g, logvar part)
    // (threads without an element ask for the batch's first one and never use it: a conditional request would be
    //  joined with its alternative here, and the join waits for every request in flight)
    float eps_t;
    {
      const int rr = tid >> 5, c = tid & 31;
      const int jj = c < gd ? c : c - gd;
      eps_t = g.eps[(c < 2 * gd && row0 + rr < g.B) ? (long)(row0 + rr) * gd + jj : 0];
    }
    HHRL_STAMP(3, 2);
    // ================================================================= per-lane forward
    // ---- phi.0 + ReLU summed over each lane's cars
    if (tid < 256) {
      const int rr = tid >> 4, k = tid & 15;
      const float wa = bs[FB_PHI0 + 2 * k], wb = bs[FB_PHI0 + 2 * k + 1], bb = bs[FB_PHI0 + 32 + k];
      float s = 0.f;
      if (rr < nlanes) {
        // two loops, so that each reads through a pointer of a known address space
        if (staged) {
          const float* x = Icars + rr * g.N * 2;
          for (int n = 0; n < g.N; ++n) s += fmaxf(wa * x[2 * n] + wb * x[2 * n + 1] + bb, 0.f);
        } else {
          const float* x = g.data + (car_base + (long)rr * g.N) * 2;
          for (int n = 0; n < g.N; ++n) s += fmaxf(wa * x[2 * n] + wb * x[2 * n + 1] + bb, 0.f);
        }
      }
      Ih1[rr * 33 + k] = s;
    }
    lds_barrier();
    HHRL_STAMP(3, 22);
    // ---- phi.2 on the summed hidden: pooled = W2 h1s + N b2 (two blocks)
    if (wave < 2) {
      const float b0v = (float)g.N * bs[LB_PHI2 + 16 * wave + j];
      const f32x4 c = SINGLE ? wave_mma16_bvec<16, 33>(Ih1, fPhi2, lane, b0v)
                             : wave_mma16<16, 33, 1, 1, 17>(Ih1, ovl + LW_PHI2 + 16 * wave * 17, lane, b0v);
      store_block16<33>(Ipool, 0, 16 * wave, c, lane);
    }
    if (SINGLE) req_c();
    lds_barrier();
    HHRL_STAMP(3, 23);
    // ---- rho.0 (32 -> 128, ReLU): block `wave`
    {
      f32x4 c = SINGLE ? wave_mma16_bvec<32, 33>(Ipool, fRho0, lane, bs[LB_RHO0 + 16 * wave + j])
                       : wave_mma16<32, 33, 1, 1, 33>(Ipool, ovl + LW_RHO0 + 16 * wave * 33, lane, bs[LB_RHO0 + 16 * wave + j]);
#pragma unroll
      for (int e = 0; e < 4; ++e) c[e] = fmaxf(c[e], 0.f);
      store_block16<129>(Ir1, 0, 16 * wave, c, lane);
    }
    lds_barrier();
    HHRL_STAMP(3, 24);
    // ---- rho.2 (128 -> 64): four blocks on waves 0..3
    if (wave < 4) {
      const f32x4 c = SINGLE ? wave_mma16_bvec<128, 129>(Ir1, fBig, lane, bs[LB_RHO2 + 16 * wave + j])
                             : wave_mma16<128, 129, 1, 1, 129>(Ir1, ovl + LW_RHO2 + 16 * wave * 129, lane, bs[LB_RHO2 + 16 * wave + j]);
      store_block16<65>(Ir2, 0, 16 * wave, c, lane);
    }
    if (SINGLE) {
      req_d();
      req_e();
    }
    lds_barrier();
    HHRL_STAMP(3, 25);
    // ---- heads (64 -> 2g; columns >= 2g come out as zero): two blocks
    if (SINGLE) {
      if (wave >= 6) {
        const int hb = wave - 6;
        const float4 (&fHead)[4] = reinterpret_cast<const float4 (&)[4]>(fBig);
        f32x4 c = wave_mma16_bvec<64, 65>(Ir2, fHead, lane, bs[LB_HEAD + 16 * hb + j]);
        if (16 * hb + j >= 2 * gd) c = zero4;
        store_block16<33>(Iml, 0, 16 * hb, c, lane);
      }
    } else if (wave < 2) {
      const f32x4 c = wave_mma16<64, 65, 1, 1, 65>(Ir2, ovl + LW_HEAD + 16 * wave * 65, lane, bs[LB_HEAD + 16 * wave + j]);
      store_block16<33>(Iml, 0, 16 * wave, c, lane);
    }
    lds_barrier();
    HHRL_STAMP(3, 26);
    // ---- reparameterisation z = mean + eps * exp(logvar / 2), zero-padded to 16 columns
    {
      const int rr = tid >> 5, c = tid & 31;
      if (c < 16) {
        float zv = 0.f;
        if (c < gd && row0 + rr < g.B) zv = Iml[rr * 33 + c] + eps_t * expf(0.5f * Iml[rr * 33 + gd + c]);
        Iz[rr * 17 + c] = zv;
      }
    }
    lds_barrier();
    HHRL_STAMP(3, 27);
    // ---- decoder.0 (g -> 128, ReLU): block `wave`
    {
      f32x4 c = SINGLE ? wave_mma16_breg<16, 17, 1>(Iz, fDec0, lane, bs[LB_DEC0 + 16 * wave + j])
                       : wave_mma16<16, 17, 1, 1, 17>(Iz, ovl + LW_DEC0 + 16 * wave * 17, lane, bs[LB_DEC0 + 16 * wave + j]);
#pragma unroll
      for (int e = 0; e < 4; ++e) c[e] = fmaxf(c[e], 0.f);
      store_block16<129>(Id1, 0, 16 * wave, c, lane);
    }
    lds_barrier();
    HHRL_STAMP(3, 28);
    // ---- decoder.2 (128 -> 32): two blocks
    if (SINGLE) {
      if (wave == 4 || wave == 5) {
        const int db = wave - 4;
        const f32x4 c = wave_mma16_bvec<128, 129>(Id1, fBig, lane, bs[LB_DEC2 + 16 * db + j]);
        store_block16<33>(Id2, 0, 16 * db, c, lane);
      }
    } else if (wave < 2) {
      const f32x4 c = wave_mma16<128, 129, 1, 1, 129>(Id1, ovl + LW_DEC2 + 16 * wave * 129, lane, bs[LB_DEC2 + 16 * wave + j]);
      store_block16<33>(Id2, 0, 16 * wave, c, lane);
    }
    if (first) {
      // what the prologue's later requests brought and the phases from here on read from LDS: func.2's image, p and q
      commit_w4<32, 128, 2, NT>(W2s, 129, v2, tid);
      if (o_pq >= 0) {
        float pv = 0.f, qv = pq_b0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float* wl = Itmp + 4 * e;
          const float* bl = Itmp + 32 + 4 * e;
          pv += pq_e[e].x * wl[0] + pq_e[e].y * wl[1] + pq_e[e].z * wl[2] + pq_e[e].w * wl[3];
          qv += pq_e[e].x * bl[0] + pq_e[e].y * bl[1] + pq_e[e].z * bl[2] + pq_e[e].w * bl[3];
        }
        bs[FB_Q + o_pq] = qv;
        bs[FB_P + o_pq] = pv;
      }
    }
    lds_barrier();
    HHRL_STAMP(3, 3);
    // ---- U = d2 A^T + q for the tile's cars: block `wave`, into the region the weight images leave
    //      (no phase below reads them before they are staged again)
    {
      f32x4 c0, c1 = zero4;
      const float qv = bs[FB_Q + 16 * wave + j];
      c0 = f32x4{qv, qv, qv, qv};
      const float* ap = Id2 + j * 33 + q;
#pragma unroll
      for (int s = 0; s < 8; s += 2) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[4 * s], w0u[s], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[4 * s + 4], w0u[s + 1], c1, 0, 0, 0);
      }
      store_block16<129>(Iu, 0, 16 * wave, c0 + c1, lane);
    }
    lds_barrier();
    HHRL_STAMP(3, 4);

    // ================================================================= the tile's cars
    float4 S4 = make_float4(0.f, 0.f, 0.f, 0.f);       // S[lane tid / 32][columns 4 (tid % 32) ..]: df1 summed over the lane's cars
    float p_t[8];
#pragma unroll
    for (int blk = 0; blk < 8; ++blk) p_t[blk] = bs[FB_P + 16 * blk + j];
    // S += a round's tiles, in tile order: a thread owns one traffic lane and four columns.  Each wave has left
    // the first traffic lane of its car tile in first_lane[]; a tile that holds no row of the thread's lane is read
    // at row 0 and not added, so that the eight reads are independent of one another.
    float* first_lane = b4s + 8;
    auto add_round_to_S = [&](int nw) {
      const int rr = tid >> 5, c4 = 4 * (tid & 31);
#pragma unroll
      for (int u = 0; u < ST_WAVES; ++u) {
        const int s = rr - (int)first_lane[u];
        const bool ok = u < nw && s >= 0 && s < g.R;
        const float4 e = *reinterpret_cast<const float4*>(ovl + FO_WAVE0 + u * CAR_WAVE_FLOATS + (ok ? s : 0) * 128 + c4);
        S4.x += ok ? e.x : 0.f; S4.y += ok ? e.y : 0.f; S4.z += ok ? e.z : 0.f; S4.w += ok ? e.w : 0.f;
      }
    };
    for (int c_lo = 0; c_lo < ntiles; c_lo += ST_WAVES) {
      const int nw = min(ST_WAVES, ntiles - c_lo);      // waves with a car tile in this round
      const int ct = c_lo + wave;
      const bool active = wave < nw;                    // wave-uniform
      float loc_i[4], ind[4];
      float in_v = 0.f;
      int in_b = -1;
      unsigned relu0 = 0u;                               // bit 4 blk + e: func.0's output (block blk, row 4 q + e) is positive
      if (active) {
        f32x4 f1[8];
        // ---- inputs.  Lane j (all four groups q) holds car j of the tile; for the four rows 4 q + e of the
        //      accumulator layout a lane also holds their loc and column 16 blk + j of their traffic lane's U row.
        const int crow = ct * CAR_TILE + j;             // car within the lane tile
        const bool valid = crow < ncars;
        float2 lv = make_float2(0.f, 0.f);
        if (valid) {
          if (staged) lv = *reinterpret_cast<const float2*>(Icars + 2 * crow);
          else lv = *reinterpret_cast<const float2*>(g.data + 2 * (car_base + crow));
        }
        in_b = valid ? (int)((unsigned)crow / (unsigned)g.N) : -1;     // traffic lane within the tile, 0..15
        int b_i[4], b_k[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          b_i[e] = __shfl(in_b, 4 * q + e, 64);
          b_k[e] = __shfl(in_b, 4 * e + q, 64);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float* up = Iu + max(b_i[e], 0) * 129 + j;
#pragma unroll
          for (int blk = 0; blk < 8; ++blk) f1[blk][e] = up[16 * blk];
        }
        const int b_first = __builtin_amdgcn_readfirstlane(in_b);
        if (lane == 0) first_lane[wave] = (float)b_first;
#pragma unroll
        for (int e = 0; e < 4; ++e) ind[e] = (b_k[e] >= 0 && b_k[e] - b_first == j) ? 1.f : 0.f;
        in_v = lv.y;
#pragma unroll
        for (int e = 0; e < 4; ++e) loc_i[e] = __shfl(lv.x, 4 * q + e, 64);
        // ---- forward: func.0 = ReLU(U[lane] + loc * p)
#pragma unroll
        for (int blk = 0; blk < 8; ++blk) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            f1[blk][e] = fmaxf(fmaf(loc_i[e], p_t[blk], f1[blk][e]), 0.f);
            relu0 |= (f1[blk][e] > 0.f ? 1u : 0u) << (4 * blk + e);
          }
          store_block16<129>(f1i, 0, 16 * blk, f1[blk], lane);
        }
        wave_sync();
        HHRL_STAMP(3, 10);
    // ---- forward: func.2 (128 -> 32, ReLU)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          f32x4 c = wave_mma16<128, 129, 1, 1, 129>(f1i, W2s + 16 * t * 129, lane, b2s[16 * t + j]);
#pragma unroll
          for (int e = 0; e < 4; ++e) c[e] = fmaxf(c[e], 0.f);
          store_block16<33>(f2i, 0, 16 * t, c, lane);
        }
        wave_sync();
        HHRL_STAMP(3, 11);
    // ---- forward: func.4 (32 -> 1, sigmoid), squared error and its gradient per car
        {
          float s = 0.f;
#pragma unroll
          for (int n = 0; n < 8; ++n) s += w4s[8 * q + n] * f2i[j * 33 + 8 * q + n];
          s += __shfl_xor(s, 16, 64);
          s += __shfl_xor(s, 32, 64);
          s += b4s[0];
          const float out = 1.f / (1.f + expf(-s));
          const float d = out - in_v;
          float dp = 0.f;
          if (in_b >= 0) {
            if (q == 0) lossacc += g.recon_scale * d * d;
            dp = 2.f * g.recon_scale * d * out * (1.f - out);
          }
          if (q == 0) {
            dpre[j] = dp;
            b4acc += dp;
          }
        }
        wave_sync();
        HHRL_STAMP(3, 12);
    // ---- backward func.4 / ReLU of func.2: dW4, db2 and df2 (in place over the f2 image); lane: feature r32, cars 8 h32 ..
        {
          const float w4n = w4s[r32];
#pragma unroll
          for (int rr = 8 * h32; rr < 8 * h32 + 8; ++rr) {
            const float fv = f2i[rr * 33 + r32];
            const float dp = dpre[rr];
            w4acc += dp * fv;
            const float df = fv > 0.f ? dp * w4n : 0.f;
            db2acc += df;
            f2i[rr * 33 + r32] = df;
          }
        }
        wave_sync();
      }
      HHRL_STAMP(3, 13);
    // ---- df1 before its ReLU mask: (df2 W2), from the wave's own images (under the barrier's wait)
      f32x4 df1[8];
      if (active) {
#pragma unroll
        for (int blk = 0; blk < 8; ++blk) df1[blk] = wave_mma16<32, 33, 1, 129, 1>(f2i, W2s + 16 * blk, lane, 0.f);
      }
      HHRL_STAMP(3, 14);
      lds_barrier();
      HHRL_STAMP(3, 15);
      // ---- func.2's weight gradient across the waves: dW2[32][128] += sum over the round's tiles of df2^T f1;
      //      this wave owns the blocks (rows 0..15 and 16..31) x (columns 16 wave ..)
#pragma unroll
      for (int u = 0; u < ST_WAVES; ++u) {
        if (u >= nw) break;
        const float* fu = ovl + FO_WAVE0 + u * CAR_WAVE_FLOATS;
        const float* a0 = fu + CAR_F2 + q * 33 + j;         // A(i = feature, k = car): df2[car][feature]
        const float* b0 = fu + CAR_F1 + q * 129 + 16 * wave + j;   // B(k = car, j): f1[car][column]
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const float bv = b0[4 * s * 129];
          dW2a[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv, a0[4 * s * 33], dW2a[0], 0, 0, 0);
          dW2a[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv, a0[4 * s * 33 + 16], dW2a[1], 0, 0, 0);
        }
      }
      HHRL_STAMP(3, 16);
      lds_barrier();                                   // every wave is done with the f1 images
      HHRL_STAMP(3, 17);
      if (active) {
        // ---- ReLU mask of func.0, its column sums (db0, and s weighted by loc), df1 over the f1 image
#pragma unroll
        for (int blk = 0; blk < 8; ++blk) {
          float cs = 0.f, ss = 0.f;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float dv = ((relu0 >> (4 * blk + e)) & 1u) ? df1[blk][e] : 0.f;
            df1[blk][e] = dv;
            cs += dv;
            ss = fmaf(dv, loc_i[e], ss);
          }
          db0[blk] += cs;
          sacc[blk] += ss;
          store_block16<129>(f1i, 0, 16 * blk, df1[blk], lane);
        }
        wave_sync();
        HHRL_STAMP(3, 18);
    // ---- df1 summed per traffic lane within the car tile on the matrix core (0/1 indicator operand):
        //      P[s][col], s = traffic lane - first lane of the car tile; left in the wave's image block, [R][128]
        f32x4 P[8];
#pragma unroll
        for (int blk = 0; blk < 8; ++blk) P[blk] = zero4;
#pragma unroll
        for (int st = 0; st < 4; ++st) {
#pragma unroll
          for (int blk = 0; blk < 8; ++blk)
            P[blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(ind[st], f1i[(4 * st + q) * 129 + 16 * blk + j], P[blk], 0, 0, 0);
        }
        wave_sync();                                     // the image has been read
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (4 * q + e < g.R) {
#pragma unroll
            for (int blk = 0; blk < 8; ++blk) wl[(4 * q + e) * 128 + 16 * blk + j] = P[blk][e];
          }
        }
      }
      HHRL_STAMP(3, 19);
      lds_barrier();
      HHRL_STAMP(3, 20);
      // With one tile per workgroup the last round's sums are added after the loop, under the requests for the
      // operands of the lane backward.
      if (SINGLE && c_lo + ST_WAVES >= ntiles) break;
      add_round_to_S(nw);
      lds_barrier();                                   // the image blocks are free for the next round
    }
    HHRL_STAMP(3, 5);
    // ---- the weight images are requested again (they are in L2); S takes U's place meanwhile.  The thread index
    //      goes through an opaque copy, so that the addresses are formed again here instead of being kept in
    //      registers (or spilled) since the prologue.
    int tid_b = tid;
    asm volatile("" : "+v"(tid_b));
    float w0r[32];                                       // func.0's columns for d2 as the B operand of dd2 = S A (waves 0, 1)
    if (wave < 2) {
      const float* src = (g.params + g.ow[FUNC0]) + q * 64 + 16 * wave + j;
#pragma unroll
      for (int s2 = 0; s2 < 32; ++s2) w0r[s2] = src[s2 * 256];
    }
    // With one tile per workgroup no forward pass follows, and every weight matrix is the B operand of exactly one
    // product of the lane backward, in its own row-major orientation: B(k = output, j = input).  Each wave then
    // asks for its operand columns directly (they are in L2), a phase or two before the product that uses them,
    // and the images in LDS are not written again.  Where the zero-padded image had no weight the request goes to
    // the matrix's first element instead (no branches, and no select that would wait for the value here): those
    // operand entries meet zero columns of dml (heads) or produce columns of dz that nothing reads (decoder.0).
    float bDec2r[8], bDec0r[8], bHeadr[8], bRho2r[16], bRho0r[32], bPhi2r[8];
    if (SINGLE) {
      const int lane_b = tid_b & 63, wave_b = tid_b >> 6, jb = lane_b & 15, qb = lane_b >> 4;
      const float* s2 = (g.params + g.ow[DEC2]) + qb * 128 + 16 * wave_b + jb;
#pragma unroll
      for (int s = 0; s < 8; ++s) bDec2r[s] = s2[s * 4 * 128];
      if (wave_b < 4) {
        const bool in0 = jb < gd;
        const float* s0 = (g.params + g.ow[DEC0]) + (in0 ? (32 * wave_b + qb) * gd + jb : 0);
        const int col = 16 * wave_b + jb;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          bDec0r[s] = s0[in0 ? s * 4 * gd : 0];
          const int k = 4 * s + qb;
          const int at = k < gd ? g.ow[MEAN] + k * 64 + col : g.ow[LOGVAR] + (k < 2 * gd ? (k - gd) * 64 + col : 0);
          bHeadr[s] = g.params[at];
        }
      }
      write_func2();
      park_car_sums();
      if (ntiles > 0) add_round_to_S(ntiles - (ntiles - 1) / ST_WAVES * ST_WAVES);
    } else {
      request_weights(tid_b);
    }
    {
      float* p = Iu + (tid >> 5) * 129 + 4 * (tid & 31);
      p[0] = S4.x; p[1] = S4.y; p[2] = S4.z; p[3] = S4.w;
    }
    lds_barrier();

    HHRL_STAMP(3, 30);
    // ================================================================= per-lane backward
    // ---- func.0, the part shared by a lane's cars: dd2 = S A (waves 0, 1), dA[128][32] += S^T d2 (row block `wave`)
    {
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) aF0[cb] = wave_mma16<16, 1, 33, 129, 1>(Id2 + 16 * cb, Iu + 16 * wave, lane, aF0[cb]);
      if (SINGLE) write_f0();
      if (wave < 2) {
        const float* ap = Iu + j * 129 + q;
        f32x4 c0 = zero4, c1 = zero4;
#pragma unroll
        for (int s2 = 0; s2 < 32; s2 += 2) {
          c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[4 * s2], w0r[s2], c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[4 * s2 + 4], w0r[s2 + 1], c1, 0, 0, 0);
        }
        store_block16<33>(Itmp, 0, 16 * wave, c0 + c1, lane);
      }
    }
    HHRL_STAMP(3, 31);
    lds_barrier();
    if (!SINGLE) {
      commit_weights(tid_b);                             // over S (consumed) and the car images
      lds_barrier();
    }
    HHRL_STAMP(3, 6);
    // ---- decoder.2 backward: dW[32][128] += dd2^T d1 (column block `wave`), db, dd1 = (dd2 W) o relu'(d1) in place
    {
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) aDec2[rb] = wave_mma16<16, 1, 129, 33, 1>(Id1 + 16 * wave, Itmp + 16 * rb, lane, aDec2[rb]);
      if (SINGLE) {
        const float* s = (g.params + g.ow[RHO2]) + q * 128 + 16 * wave + j;
#pragma unroll
        for (int u = 0; u < 16; ++u) bRho2r[u] = s[u * 4 * 128];
      }
      const f32x4 d = SINGLE ? wave_mma16_breg<32, 33, 1>(Itmp, bDec2r, lane, 0.f)
                             : wave_mma16<32, 33, 1, 129, 1>(Itmp, ovl + LW_DEC2 + 16 * wave, lane, 0.f);
      wave_sync();
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float* p = Id1 + (4 * q + e) * 129 + 16 * wave + j;
        *p = *p > 0.f ? d[e] : 0.f;
      }
      if (tid >= 256 && tid < 288) bDec2 += colsum16(Itmp, 33, tid - 256);
    }
    if (SINGLE) write_dec2();
    lds_barrier();
    HHRL_STAMP(3, 32);
    // ---- decoder.0 backward: dW[128][g] += dd1^T z (row block `wave`), db, dz = dd1 W as four partial products (waves 0..3)
    {
      aDec0 = wave_mma16<16, 1, 129, 17, 1>(Id1 + 16 * wave, Iz, lane, aDec0);
      if (wave < 4) {
        const f32x4 c = SINGLE ? wave_mma16_breg<32, 129, 1>(Id1 + 32 * wave, bDec0r, lane, 0.f)
                               : wave_mma16<32, 129, 1, 17, 1>(Id1 + 32 * wave, ovl + LW_DEC0 + 32 * wave * 17, lane, 0.f);
        store_block16<17>(ovl + FO_SPARE + wave * 16 * 17, 0, 0, c, lane);
      }
      if (tid >= 256 && tid < 384) bDec0 += colsum16(Id1, 129, tid - 256);
    }
    if (SINGLE) write_dec0();
    lds_barrier();
    HHRL_STAMP(3, 33);
    // ---- latent gradient (KL + reparameterisation): dml = [dz + mean | dz*eps*0.5*std + 0.5*(exp(lv)-1) | 0...]
    {
      const int rr = tid >> 5, c = tid & 31;
      float outv = 0.f;
      if (c < 2 * gd && row0 + rr < g.B) {
        const int jj = c < gd ? c : c - gd;
        const float mean = Iml[rr * 33 + jj], lv = Iml[rr * 33 + gd + jj];
        const float* dzp = ovl + FO_SPARE + rr * 17 + jj;
        const float dz = (dzp[0] + dzp[16 * 17]) + (dzp[2 * 16 * 17] + dzp[3 * 16 * 17]);
        if (c < gd) {
          outv = dz + mean;
          klacc += -0.5f * (1.f + lv - mean * mean - expf(lv));
        } else {
          outv = dz * eps_t * 0.5f * expf(0.5f * lv) + 0.5f * (expf(lv) - 1.f);
        }
      }
      Idml[rr * 33 + c] = outv;
    }
    if (SINGLE) park_kl();
    lds_barrier();
    HHRL_STAMP(3, 34);
    // ---- heads backward: dW[2g][64] += dml^T r2 (block (wave / 4, wave % 4)), db, dr2 = dml W (waves 0..3; replaces r2)
    {
      aHead = wave_mma16<16, 1, 65, 33, 1>(Ir2 + 16 * (wave & 3), Idml + 16 * (wave >> 2), lane, aHead);
      f32x4 d = zero4;
      if (wave < 4)
        d = SINGLE ? wave_mma16_breg<32, 33, 1>(Idml, bHeadr, lane, 0.f)
                   : wave_mma16<32, 33, 1, 65, 1>(Idml, ovl + LW_HEAD + 16 * wave, lane, 0.f);
      if (tid >= 256 && tid < 288) bHead += colsum16(Idml, 33, tid - 256);
      lds_barrier();
      if (wave < 4) store_block16<65>(Ir2, 0, 16 * wave, d, lane);
    }
    if (SINGLE) write_heads();
    lds_barrier();
    HHRL_STAMP(3, 35);
    // ---- rho.2 backward: dW[64][128] += dr2^T r1 (column block `wave`, four row blocks), db, dr1 = (dr2 W) o relu'(r1) in place
    {
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) aRho2[rb] = wave_mma16<16, 1, 129, 65, 1>(Ir1 + 16 * wave, Ir2 + 16 * rb, lane, aRho2[rb]);
      if (SINGLE && wave < 2) {
        const float* s = (g.params + g.ow[RHO0]) + q * 32 + 16 * wave + j;
#pragma unroll
        for (int u = 0; u < 32; ++u) bRho0r[u] = s[u * 4 * 32];
      }
      if (SINGLE && wave == 2) {
        const float* s = (g.params + g.ow[PHI2]) + q * 16 + j;
#pragma unroll
        for (int u = 0; u < 8; ++u) bPhi2r[u] = s[u * 4 * 16];
      }
      const f32x4 d = SINGLE ? wave_mma16_breg<64, 65, 1>(Ir2, bRho2r, lane, 0.f)
                             : wave_mma16<64, 65, 1, 129, 1>(Ir2, ovl + LW_RHO2 + 16 * wave, lane, 0.f);
      wave_sync();
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float* p = Ir1 + (4 * q + e) * 129 + 16 * wave + j;
        *p = *p > 0.f ? d[e] : 0.f;
      }
      if (tid >= 256 && tid < 320) bRho2 += colsum16(Ir2, 65, tid - 256);
    }
    // (the slab entries of the last layers leave in portions of about 16 KB per phase, one phase behind their
    //  products where needed: a CU hands about 21 bytes per clock to memory through agent-scope stores
    //  (tools/probes/write_through_probe.hip), its queue takes about that much without holding the storing waves
    //  up, and rho.2's 32 KB in one go held them 2.6 k clocks)
    if (SINGLE) write_rho2_lo();
    lds_barrier();
    HHRL_STAMP(3, 36);
    // ---- rho.0 backward: dW[128][32] += dr1^T pooled (row block `wave`), db, dpooled = dr1 W (waves 0, 1) -> ml slot
    {
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) aRho0[cb] = wave_mma16<16, 1, 33, 129, 1>(Ipool + 16 * cb, Ir1 + 16 * wave, lane, aRho0[cb]);
      if (wave < 2) {
        const f32x4 c = SINGLE ? wave_mma16_breg<128, 129, 1>(Ir1, bRho0r, lane, 0.f)
                               : wave_mma16<128, 129, 1, 33, 1>(Ir1, ovl + LW_RHO0 + 16 * wave, lane, 0.f);
        store_block16<33>(Iml, 0, 16 * wave, c, lane);
      }
      if (tid >= 256 && tid < 384) bRho0 += colsum16(Ir1, 129, tid - 256);
    }
    if (SINGLE) write_rho2_hi();
    lds_barrier();
    HHRL_STAMP(3, 37);
    // ---- phi.2 backward on the summed hidden: dW[32][16] += dpooled^T h1s (waves 0, 1), db = N colsum(dpooled),
    //      t = dpooled W_phi2 (wave 2): gradient at the hidden layer of every car of the lane, before its ReLU mask
    {
      if (wave < 2) {
        aPhi2 = wave_mma16<16, 1, 33, 33, 1>(Ih1, Iml + 16 * wave, lane, aPhi2);
      } else if (wave == 2) {
        const f32x4 c = SINGLE ? wave_mma16_breg<32, 33, 1>(Iml, bPhi2r, lane, 0.f)
                               : wave_mma16<32, 33, 1, 17, 1>(Iml, ovl + LW_PHI2, lane, 0.f);
        store_block16<33>(Itmp, 0, 0, c, lane);
      }
      if (tid >= 256 && tid < 288) bPhi2 += (float)g.N * colsum16(Iml, 33, tid - 256);
    }
    if (SINGLE) write_rho0();
    lds_barrier();
    HHRL_STAMP(3, 38);
    // ---- phi.0 backward: per (lane, hidden unit) walk the cars again and apply the ReLU mask
    {
      float* red = Ir2;       // [16][49] partial sums
      if (tid < 256) {
        const int rr = tid >> 4, k = tid & 15;
        const float wa = bs[FB_PHI0 + 2 * k], wb = bs[FB_PHI0 + 2 * k + 1], bb = bs[FB_PHI0 + 32 + k];
        float a0 = 0.f, a1 = 0.f, c = 0.f;
        if (rr < nlanes) {
          auto walk = [&](const float* x) {
            for (int n = 0; n < g.N; ++n) {
              const float x0 = x[2 * n], x1 = x[2 * n + 1];
              if (wa * x0 + wb * x1 + bb > 0.f) { a0 += x0; a1 += x1; c += 1.f; }
            }
          };
          if (staged) walk(Icars + rr * g.N * 2);
          else walk(g.data + (car_base + (long)rr * g.N) * 2);
        }
        const float t = Itmp[rr * 33 + k];
        red[rr * 49 + k] = t * a0;
        red[rr * 49 + 16 + k] = t * a1;
        red[rr * 49 + 32 + k] = t * c;
      }
      lds_barrier();
      if (tid < 48) {
        float s = 0.f;
        for (int rr = 0; rr < 16; ++rr) s += red[rr * 49 + tid];
        gPhi0 += s;
      }
    }
    if (SINGLE) {
      write_phi2();
      write_phi0();
    }
    lds_barrier();
    HHRL_STAMP(3, 7);
  } while (!SINGLE && (tile += gridDim.x) < g.tiles);

  // ================================================================= the workgroup's slab
  // W0[8 og + k][32 + jj] for the loc gradients at the very end, requested now
  float w0x[8];
  {
    const int jj = tid & 31, og = tid >> 5;
#pragma unroll
    for (int k = 0; k < 8; ++k) w0x[k] = (g.params + g.ow[FUNC0])[(8 * og + k) * 64 + 32 + jj];
  }
  // the loc embedding as it is before this step's update, for the slab sum: every thread asks (no branch around
  // the request), workgroup 0 stores at the very end
  const float loc_copy = (g.params + ((tid & 32) ? g.ob[LOC] : g.ow[LOC]))[tid & 31];
  if (!SINGLE) {
    write_f0();
    write_dec2();
    write_dec0();
    write_heads();
    write_rho2();
    write_rho0();
    write_phi2();
    write_phi0();
    write_func2();
  }
  // ---- the per-car layers' small sums: lane groups first, then the eight waves through LDS in a fixed order
  if (!SINGLE) {
    park_car_sums();
    park_kl();
    lds_barrier();
  }
  // (the few hundred entries from here on are ordinary stores: written through, their acknowledgements would be
  //  waited for at the very end of the workgroup)
  auto all8 = [&](int i) {
    return ((cmb[i] + cmb[QS + i]) + (cmb[2 * QS + i] + cmb[3 * QS + i])) +
           ((cmb[4 * QS + i] + cmb[5 * QS + i]) + (cmb[6 * QS + i] + cmb[7 * QS + i]));
  };
  // s[128] | db0[128] | partial products [16][64]
  float* fin = SINGLE ? ovl : cmb + ST_WAVES * Q1;
  if (tid < 128) {
    const float sv = all8(B_S + tid);
    fin[tid] = sv;
    slab[g.ow[FUNC0] + tid * 64 + 32] = sv;           // s[o] travels in the first position of row o that no sum uses
  } else if (tid < 256) {
    const float dv = all8(B_B0 + tid - 128);
    fin[tid] = dv;
    slab[g.ob[FUNC0] + tid - 128] = dv;
  } else if (tid < 288) {
    slab[g.ob[FUNC2] + tid - 256] = all8(B_B2 + tid - 256);
  } else if (tid < 320) {
    slab[g.ow[FUNC4] + tid - 288] = all8(B_W4 + tid - 288);
  } else if (tid == 320) {
    slab[g.ob[FUNC4]] = all8(B_B4);
  } else if (tid == 321) {
    g.loss_part[blockIdx.x] = all8(B_LOSS);
  } else if (tid == 322) {
    g.kl_part[blockIdx.x] = all8(B_KL);
  }
  lds_barrier();
  // ---- loc embedding: dwloc[jj] = sum_o s[o] W0[o][32 + jj], dbloc[jj] = sum_o db0[o] W0[o][32 + jj]
  //      (this workgroup's share; sixteen groups of eight o each, combined in a fixed order)
  {
    const int jj = tid & 31, og = tid >> 5;
    float pw = 0.f, pb = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      pw = fmaf(fin[8 * og + k], w0x[k], pw);
      pb = fmaf(fin[128 + 8 * og + k], w0x[k], pb);
    }
    fin[256 + og * 64 + jj] = pw;
    fin[256 + og * 64 + 32 + jj] = pb;
  }
  lds_barrier();
  if (tid < 64) {
    const float* pp = fin + 256 + tid;
    float sum = 0.f;
#pragma unroll
    for (int og = 0; og < 16; og += 4) sum += (pp[og * 64] + pp[(og + 1) * 64]) + (pp[(og + 2) * 64] + pp[(og + 3) * 64]);
    slab[(tid < 32 ? g.ow[LOC] + tid : g.ob[LOC] + tid - 32)] = sum;
    if (blockIdx.x == 0) g.aux[128 + tid] = loc_copy;
  }
  HHRL_STAMP(3, 8);
  HHRL_WG_SPAN(1);
}

// The same reduction for the one-launch step, where every column is summed over the same n slabs: 64
// consecutive parameters per workgroup, read 16 bytes per thread.  Thread (rq = tid / 16, c4 = tid % 16) adds
// rows rq, rq + 16, ... of columns 4 c4 .. 4 c4 + 3 in row order (four running sums of every fourth of its rows
// keep several loads in flight), LDS combines the sixteen row phases in a fixed order: the same inputs give
// the same bits.  Requires total % 4 == 0, a 16-byte aligned slab and f0_lo % 64 == 0 (a row of func.0's matrix
// is then exactly one workgroup's block, whose column 32 carries s[o]).
__global__ __launch_bounds__(256) void vae_reduce_slabs_kernel(const float* __restrict__ slab, long total, int n, long f0_lo, long b0_off,
                                                               const float* __restrict__ locsave, float* grads,
                                                               const float* __restrict__ loss_part, const float* __restrict__ kl_part,
                                                               float* loss, float* params, float* m, float* v, AdamK k) {
  __shared__ float part[16][68];
  __shared__ float dsum[4];
  const int tid = threadIdx.x, c4 = tid & 15, rq = tid >> 4;
  const long i0 = (long)blockIdx.x * 64;
  const long e0 = i0 - f0_lo;
  const bool f0_block = e0 >= 0 && e0 < 8192;             // a row of func.0's matrix (block-uniform)
  // Everything the workgroup reads is requested before the first sum: the optimizer state of its 64 parameters, the
  // db0 column of a func.0 block, and the slab rows in batches of sixteen loads per thread (256 rows), so that a
  // 256-slab reduction is one memory round trip instead of four dependent ones.  The order of the additions is
  // that of the plain loop: four running sums over every fourth of a thread's rows.
  const bool own = tid < 64 && i0 + tid < total;
  float pi = 0.f, mi = 0.f, vi = 0.f;
  if (own && params != nullptr) {
    pi = params[i0 + tid];
    mi = m[i0 + tid];
    vi = v[i0 + tid];
  }
  float d = 0.f;
  if (f0_block) {
    const long o = e0 >> 6;
    for (int r = tid; r < n; r += 256) d += slab[(long)r * total + b0_off + o];
  }
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
  if (i0 + 4 * c4 < total) {
    const float* col = slab + i0 + 4 * c4;
    auto add = [](float4& a, const float4& b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; };
    int r = rq;
    for (; r + 240 < n; r += 256) {
      float4 t[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) t[u] = *reinterpret_cast<const float4*>(col + (long)(r + 16 * u) * total);
#pragma unroll
      for (int u = 0; u < 16; u += 4) {
        add(a0, t[u]); add(a1, t[u + 1]); add(a2, t[u + 2]); add(a3, t[u + 3]);
      }
    }
    for (; r + 48 < n; r += 64) {
      const float4 v0 = *reinterpret_cast<const float4*>(col + (long)r * total);
      const float4 v1 = *reinterpret_cast<const float4*>(col + (long)(r + 16) * total);
      const float4 v2 = *reinterpret_cast<const float4*>(col + (long)(r + 32) * total);
      const float4 v3 = *reinterpret_cast<const float4*>(col + (long)(r + 48) * total);
      add(a0, v0); add(a1, v1); add(a2, v2); add(a3, v3);
    }
    for (; r < n; r += 16) add(a0, *reinterpret_cast<const float4*>(col + (long)r * total));
  }
  {
    float* p = &part[rq][4 * c4];
    p[0] = (a0.x + a1.x) + (a2.x + a3.x);
    p[1] = (a0.y + a1.y) + (a2.y + a3.y);
    p[2] = (a0.z + a1.z) + (a2.z + a3.z);
    p[3] = (a0.w + a1.w) + (a2.w + a3.w);
  }
  // db0[o] of this row of func.0's matrix, summed over the slabs: rows tid, tid + 256, ...; waves then in a fixed order
  if (f0_block) {
    d = wave_sum(d);
    if ((tid & 63) == 0) dsum[tid >> 6] = d;
  }
  __syncthreads();
  if (tid < 64 && i0 + tid < total) {
    const long i = i0 + tid;
    float s = 0.f;
#pragma unroll
    for (int rr = 0; rr < 16; rr += 4) s += (part[rr][tid] + part[rr + 1][tid]) + (part[rr + 2][tid] + part[rr + 3][tid]);
    float gi = s;
    if (f0_block && tid >= 32) {                           // column 32 + j: s[o] wloc[j] + db0[o] bloc[j]
      float s32 = 0.f;
#pragma unroll
      for (int rr = 0; rr < 16; rr += 4) s32 += (part[rr][32] + part[rr + 1][32]) + (part[rr + 2][32] + part[rr + 3][32]);
      const float di = (dsum[0] + dsum[1]) + (dsum[2] + dsum[3]);
      const int jj = tid - 32;
      gi = s32 * locsave[jj] + di * locsave[32 + jj];
    }
    grads[i] = gi;
    if (params != nullptr) {                               // fused optimizer step (single device)
      adam_one(pi, gi, mi, vi, k);
      params[i] = pi;
      m[i] = mi;
      v[i] = vi;
    }
  }
  if (blockIdx.x == 0 && tid >= 64 && tid < 128) {
    float t = 0.f;
    for (int kk = tid - 64; kk < n; kk += 64) t += loss_part[kk] + kl_part[kk];
    t = wave_sum(t);
    if (tid == 64) *loss = t;
  }
}

struct Ws {
  float *u, *aux, *spart, *act, *wimg, *loss_part, *kl_part, *slab;
  long total;
};

static long a64(long n) { return (n + 63) & ~63L; }

static int grid_car(long rows) {
  const long tiles = (rows + CAR_TILE - 1) / CAR_TILE;
  long gsz = (tiles + CAR_WAVES - 1) / CAR_WAVES;
  if (gsz > 256) gsz = 256;
  return (int)(gsz < 1 ? 1 : gsz);
}
static int grid_lane(int B) {
  long tiles = (B + 31) / 32;
  if (tiles > 256) tiles = 256;
  return (int)(tiles < 1 ? 1 : tiles);
}

static int grid_step(int B) {
  long tiles = (B + ST_ROWS - 1) / ST_ROWS;
  if (tiles > 256) tiles = 256;
  return (int)(tiles < 1 ? 1 : tiles);
}

// The one-launch form needs aux, the loss / KL partials and one slab per workgroup; the four-launch form
// the rest as well.  One layout serves both, so that the same workspace works whichever form runs.
static Ws carve(float* ws, int B, int N, int g, long n_flat) {
  Ws w;
  long off = 0;
  auto take = [&](long n) { float* p = ws ? ws + off : nullptr; off += a64(n); return p; };
  w.u = take((long)B * 128);
  w.aux = take(320);   // p[128] | wloc[32] bloc[32] as the step found them | s[128]
  w.spart = take(((long)B * N + CAR_TILE - 1) / CAR_TILE * lanes_per_car_tile(N) * 128);
  w.act = take((long)((B + 31) / 32) * LN_ACT_FLOATS);
  w.wimg = take(LN_WIMG_FLOATS);
  w.loss_part = take(256);
  w.kl_part = take(256);
  int ns = grid_car((long)B * N) > grid_lane(B) ? grid_car((long)B * N) : grid_lane(B);
  if (grid_step(B) > ns) ns = grid_step(B);
  w.slab = take((long)ns * n_flat);
  w.total = off;
  return w;
}

}  // namespace vf
}  // namespace hhrl

using namespace hhrl;
using namespace hhrl::vf;

extern "C" {

long hhrl_vae_workspace_floats(int B, int N, int g, long n_flat) {
  if (B <= 0 || N <= 0 || g <= 0 || n_flat <= 0) return 0;
  return carve(nullptr, B, N, g, n_flat).total;
}

}  // extern "C"

// Forward, ELBO, backward and slab reduction; with adam != nullptr the reduction also applies the
// optimizer step to params / m / v.
static int vae_step_impl(float* params, float* grads, float* m, float* v, const AdamK* adam, const long* offsets,
                         long n_flat, const float* data, const float* eps, int B, int N, int g, float recon_scale,
                         float* loss, float* ws, void* stream) {
  HHRL_REQUIRE(params && grads && offsets && data && eps && loss && ws, "vae_train_step: null pointer");
  HHRL_REQUIRE(B > 0 && N > 0, "vae_train_step: bad shape B=%d N=%d", B, N);
  HHRL_REQUIRE((long)B * N < (1L << 31) - 64, "vae_train_step: B*N = %ld cars exceed the 32-bit row index", (long)B * N);
  HHRL_REQUIRE(g == 2 || g == 4 || g == 8 || g == 16, "vae_train_step: latent size %d not in {2,4,8,16}", g);
  HHRL_REQUIRE((long)B * N < (1L << 31) - 64, "vae_train_step: B*N too large");
  static const long numel_w[NLAYER] = {32, 512, 4096, 8192, -64, -64, -128, 4096, 32, 8192, 4096, 32};
  static const long numel_b[NLAYER] = {16, 32, 128, 64, -1, -1, 128, 32, 32, 128, 32, 1};
  Offsets off;
  for (int l = 0; l < NLAYER; ++l) {
    off.w[l] = offsets[2 * l];
    off.b[l] = offsets[2 * l + 1];
    const long nw = numel_w[l] < 0 ? -numel_w[l] * g : numel_w[l];
    const long nb = numel_b[l] < 0 ? g : numel_b[l];
    HHRL_REQUIRE(off.w[l] >= 0 && off.w[l] + nw <= n_flat && off.b[l] >= 0 && off.b[l] + nb <= n_flat,
                 "vae_train_step: offsets[%d] outside the flat buffer", l);
    HHRL_REQUIRE((off.w[l] & 3) == 0, "vae_train_step: weight %d is not 16-byte aligned inside the flat buffer", l);
  }
  HHRL_REQUIRE(reinterpret_cast<uintptr_t>(params) % 16 == 0 && reinterpret_cast<uintptr_t>(ws) % 16 == 0,
               "vae_train_step: params and ws must be 16-byte aligned");
  off.total = n_flat;
  // per-car layers must be contiguous at the end of the layout (loc, func.0, func.2, func.4)
  const long car_lo = off.w[LOC], car_hi = off.b[FUNC4] + 1;
  const long f0_lo = off.w[FUNC0], f0_hi = f0_lo + 8192, b0_off = off.b[FUNC0];
  HHRL_REQUIRE(car_lo > off.b[DEC2] && off.b[LOC] > car_lo && f0_lo > off.b[LOC] && b0_off >= f0_hi && off.w[FUNC2] > b0_off &&
                   off.b[FUNC4] > off.w[FUNC2],
               "vae_train_step: unexpected parameter order");
  hipStream_t st = as_stream(stream);
  Ws w = carve(ws, B, N, g, n_flat);
  const long rows = (long)B * N;
  const int ncar = grid_car(rows), nlane = grid_lane(B);

  // Form of the step, both ending in the slab reduction: one launch for forward, ELBO and backward while the
  // grid covers the batch with one tile of 16 lanes per workgroup (up to 4096 lanes), the four-launch form
  // beyond (its kernels amortise their prologues over many tiles per workgroup).  A workgroup of the one-launch
  // form walks the 16 N cars of its lanes alone (measured 31 + 1.5 N us per step whatever the batch), while the
  // four-launch form spreads the cars of a small batch over the idle CUs (64 lanes: 44 + 0.3 N us; 1024: 40 + 0.8 N;
  // 2048: 43 + 1.0 N; 4096: 49 + 1.5 N): long lanes in small batches go to the four-launch form.
  // HHRL_VAE_STEP=1 / =4 force one form for comparisons.
  static int step_form = -1;
  if (step_form < 0) {
    const char* e = getenv("HHRL_VAE_STEP");
    step_form = e ? atoi(e) : 0;
  }
  const int n_tiles16 = (B + ST_ROWS - 1) / ST_ROWS;
  const bool one_launch_pays = n_tiles16 <= 256 && (N <= 12 || n_tiles16 >= 192 || (n_tiles16 >= 128 && N <= 24));
  if (step_form == 1 || (step_form != 4 && one_launch_pays)) {
    HHRL_REQUIRE(n_flat < (1L << 31), "vae_train_step: flat buffer of %ld floats exceeds the 32-bit offsets", n_flat);
    HHRL_TRY(ensure_dyn_lds(reinterpret_cast<const void*>(vae_step_kernel<true>), ST_LDS_FLOATS * 4, "vae_train_step"));
    HHRL_TRY(ensure_dyn_lds(reinterpret_cast<const void*>(vae_step_kernel<false>), ST_LDS_FLOATS * 4, "vae_train_step"));
    StepArgs sa;
    sa.data = data;
    sa.eps = eps;
    sa.params = params;
    for (int l = 0; l < NLAYER; ++l) {
      sa.ow[l] = (int)off.w[l];
      sa.ob[l] = (int)off.b[l];
    }
    sa.total = (int)n_flat;
    sa.aux = w.aux;
    sa.slab = w.slab;
    sa.loss_part = w.loss_part;
    sa.kl_part = w.kl_part;
    sa.B = B; sa.N = N; sa.g = g;
    sa.tiles = (B + ST_ROWS - 1) / ST_ROWS;
    sa.R = lanes_per_car_tile(N);
    sa.recon_scale = recon_scale;
    const int ngrid = grid_step(B);
    {
      // the algorithm's products of the whole step: forward, input gradients, weight gradients
      const double macs = 12896.0 * N + 16384.0 + 256.0 * g;
      ProfScope ps("vae_step_kernel", st, 6.0 * macs * B, 4.0 * B * (2.0 * N + g), true);
      if (ngrid >= sa.tiles)
        launch_timed(ps, vae_step_kernel<true>, dim3(ngrid), dim3(ST_NT), ST_LDS_FLOATS * 4, st, sa);
      else
        launch_timed(ps, vae_step_kernel<false>, dim3(ngrid), dim3(ST_NT), ST_LDS_FLOATS * 4, st, sa);
      HHRL_TRY(launch_status("vae_step_kernel"));
    }
    {
      const bool vec = n_flat % 4 == 0 && reinterpret_cast<uintptr_t>(w.slab) % 16 == 0 && f0_lo % 64 == 0;
      ProfScope ps(vec ? "vae_reduce_slabs_kernel" : "vae_reduce_grads_kernel", st, 0.0,
                   4.0 * ((double)n_flat * ngrid + n_flat * (adam ? 7.0 : 1.0)), true);
      if (vec)
        launch_timed(ps, vae_reduce_slabs_kernel, dim3(ceil_div(n_flat, 64)), dim3(256), 0, st, w.slab, n_flat, ngrid, f0_lo, b0_off,
                     w.aux + 128, grads, w.loss_part, w.kl_part, loss, adam ? params : nullptr, m, v, adam ? *adam : AdamK());
      else
        launch_timed(ps, vae_reduce_grads_kernel, dim3(ceil_div(n_flat, 64)), dim3(256), 0, st, w.slab, n_flat, ngrid, ngrid,
                     car_lo, car_hi, f0_lo, b0_off, w.aux + 128, (const float*)nullptr, 0L, 0L, 0L, 0L, grads, w.loss_part, w.kl_part,
                     loss, adam ? params : nullptr, m, v, adam ? *adam : AdamK());
      HHRL_TRY(launch_status("vae_reduce_slabs_kernel"));
    }
    return HHRL_OK;
  }

  // Threads per workgroup of the per-lane kernels: 512 (four waves that only help with the copies into LDS
  // and the element-wise phases) unless HHRL_LANE_THREADS=256 asks for the plain form (for comparisons).
  static int lane_threads = 0;
  if (lane_threads == 0) {
    const char* e = getenv("HHRL_LANE_THREADS");
    lane_threads = (e && atoi(e) == 256) ? 256 : 512;
  }
  {
    const void* fns[] = {reinterpret_cast<const void*>(vae_car_fwd_bwd_kernel),
                         reinterpret_cast<const void*>(vae_lane_forward_kernel<256>),
                         reinterpret_cast<const void*>(vae_lane_forward_kernel<512>),
                         reinterpret_cast<const void*>(vae_lane_backward_kernel<true, 256>),
                         reinterpret_cast<const void*>(vae_lane_backward_kernel<true, 512>),
                         reinterpret_cast<const void*>(vae_lane_backward_kernel<false, 256>)};
    for (int i = 0; i < 6; ++i)
      HHRL_TRY(ensure_dyn_lds(fns[i], (i == 0 ? CAR_LDS_FLOATS : LN_LDS_FLOATS) * 4, "vae_train_step"));
  }

  LaneArgs la;
  la.data = data;
  la.eps = eps;
  for (int l = 0; l < NLAYER; ++l) {
    la.w[l] = params + off.w[l];
    la.b[l] = params + off.b[l];
  }
  la.u = w.u;
  la.aux = w.aux;
  la.spart = w.spart;
  la.act = w.act;
  la.wimg = w.wimg;
  la.slab = w.slab;
  la.kl_part = w.kl_part;
  la.off = off;
  la.B = B; la.N = N; la.g = g; la.tiles = (B + 31) / 32;
  la.R = lanes_per_car_tile(N);

  const double lane_macs = 16.0 * 2 * N + 512 + 4096 + 8192 + 128.0 * g + 128.0 * g + 4096;
  {
    ProfScope ps("vae_lane_forward_kernel", st, 2.0 * lane_macs * B, 4.0 * B * (2.0 * N + g + 32), true);
    if (lane_threads == 512)
      launch_timed(ps, vae_lane_forward_kernel<512>, dim3(nlane), dim3(512), LN_LDS_FLOATS * 4, st, la);
    else
      launch_timed(ps, vae_lane_forward_kernel<256>, dim3(nlane), dim3(256), LN_LDS_FLOATS * 4, st, la);
    HHRL_TRY(launch_status("vae_lane_forward_kernel"));
  }

  CarArgs ca;
  ca.data = data;
  ca.u = w.u;
  ca.pvec = w.aux;
  ca.w0 = la.w[FUNC0];
  ca.w2 = la.w[FUNC2]; ca.b2 = la.b[FUNC2];
  ca.w4 = la.w[FUNC4]; ca.b4 = la.b[FUNC4];
  ca.spart = w.spart;
  ca.slab = w.slab;
  ca.loss_part = w.loss_part;
  ca.off = off;
  ca.rows = rows;
  ca.N = N;
  ca.tiles = (int)((rows + CAR_TILE - 1) / CAR_TILE);
  ca.R = la.R;
  ca.recon_scale = recon_scale;
  ca.vec4 = (reinterpret_cast<uintptr_t>(w.slab) % 16 == 0 && n_flat % 4 == 0 && off.w[FUNC2] % 4 == 0) ? 1 : 0;
  {
    // the algorithm's forward + input-gradient + weight-gradient products of loc/func.0/func.2/func.4:
    // 3 * 2 * 12352 flop per car (func.0's share is not executed per car here, see the kernel's header)
    ProfScope ps("vae_car_fwd_bwd_kernel", st, 6.0 * (8192 + 4096 + 32 + 32) * rows, 4.0 * (2.0 * rows + 64.0 * B), true);
    launch_timed(ps, vae_car_fwd_bwd_kernel, dim3(ncar), dim3(CAR_THREADS), CAR_LDS_FLOATS * 4, st, ca);
    HHRL_TRY(launch_status("vae_car_fwd_bwd_kernel"));
  }
  // per-car gradient slabs summed by extra workgroups of the per-lane backward launch, on CUs it leaves idle
  CarReduce cr;
  cr.slab = w.slab; cr.total = n_flat; cr.lo = car_lo; cr.hi = car_hi;
  cr.hole_lo = f0_lo; cr.hole_hi = f0_hi;
  cr.svec = w.aux + 192;
  cr.grads = grads; cr.params = adam ? params : nullptr; cr.m = m; cr.v = v;
  cr.k = adam ? *adam : AdamK();
  cr.n_car = ncar; cr.first_block = nlane; cr.n_blocks = 0;
  if (nlane >= la.tiles && lane_threads == 512 && nlane < 256 && car_lo % 64 == 0 && f0_lo % 64 == 0 && b0_off == f0_hi) {
    const long pairs = ((f0_lo - car_lo + 63) / 64 + (car_hi - f0_hi + 63) / 64 + 2 + 1) / 2;
    cr.n_blocks = (int)(pairs < 256 - nlane ? pairs : 256 - nlane);
  }
  // 64-column blocks the reduction kernel leaves out: the ranges in front of and behind func.0's matrix
  const bool co = cr.n_blocks > 0;
  const long skip_lo = co ? car_lo : 0, skip_mid = co ? f0_lo : 0, skip_back = co ? f0_hi : 0, skip_hi = co ? ((car_hi + 63) / 64) * 64 : 0;
  {
    ProfScope ps("vae_lane_backward_kernel", st, 4.0 * lane_macs * B, 4.0 * B * (2.0 * N + g + 32), true);
    if (nlane >= la.tiles && lane_threads == 512)
      launch_timed(ps, (vae_lane_backward_kernel<true, 512>), dim3(nlane + cr.n_blocks), dim3(512), LN_LDS_FLOATS * 4, st, la, cr);
    else if (nlane >= la.tiles)
      launch_timed(ps, (vae_lane_backward_kernel<true, 256>), dim3(nlane), dim3(256), LN_LDS_FLOATS * 4, st, la, cr);
    else
      launch_timed(ps, (vae_lane_backward_kernel<false, 256>), dim3(nlane), dim3(256), LN_LDS_FLOATS * 4, st, la, cr);
    HHRL_TRY(launch_status("vae_lane_backward_kernel"));
  }
  {
    // reads every slab column once (per-car columns from ncar slabs, the others from nlane), writes the gradient
    const double slab_floats = (cr.n_blocks > 0 ? 256.0 * ncar : (double)(car_hi - car_lo - 8192 + 128) * ncar) +
                               (double)(n_flat - (car_hi - car_lo) + 4096) * nlane;
    ProfScope ps("vae_reduce_grads_kernel", st, 0.0, 4.0 * (slab_floats + n_flat * (adam ? 7.0 : 1.0)), true);
    launch_timed(ps, vae_reduce_grads_kernel, dim3(ceil_div(n_flat, 64)), dim3(256), 0, st, w.slab, n_flat, ncar, nlane,
                       car_lo, car_hi, f0_lo, b0_off, w.aux + 128, co ? w.aux + 192 : nullptr, skip_lo, skip_mid, skip_back, skip_hi, grads, w.loss_part, w.kl_part, loss, adam ? params : nullptr, m, v,
                       adam ? *adam : AdamK());
    HHRL_TRY(launch_status("vae_reduce_grads_kernel"));
  }
  return HHRL_OK;
}

extern "C" {

int hhrl_vae_train_step(const float* params, float* grads, const long* offsets, long n_flat, const float* data,
                        const float* eps, int B, int N, int g, float recon_scale, float* loss, float* ws,
                        void* stream) {
  return vae_step_impl(const_cast<float*>(params), grads, nullptr, nullptr, nullptr, offsets, n_flat, data, eps, B, N, g,
                       recon_scale, loss, ws, stream);
}

int hhrl_vae_learn_step(float* params, float* grads, float* m, float* v, const long* offsets, long n_flat,
                        const float* data, const float* eps, int B, int N, int g, float recon_scale, double lr,
                        double beta1, double beta2, double adam_eps, int step, float* loss, float* ws, void* stream) {
  HHRL_REQUIRE(m && v, "vae_learn_step: null pointer");
  HHRL_REQUIRE(step >= 1, "vae_learn_step: step=%d", step);
  const AdamK k = make_adam_k(lr, beta1, beta2, adam_eps, step, 1.0);
  return vae_step_impl(params, grads, m, v, &k, offsets, n_flat, data, eps, B, N, g, recon_scale, loss, ws, stream);
}

int hhrl_vae_learn_steps(float* params, float* grads, float* m, float* v, const long* offsets, long n_flat, int n_steps,
                         const float* const* data, const float* const* eps, const int* B, const int* N, int g, double lr,
                         double beta1, double beta2, double adam_eps, int first_step, float* const* loss, float* const* ws,
                         void* stream) {
  HHRL_REQUIRE(m && v && data && eps && B && N && loss && ws, "vae_learn_steps: null pointer");
  HHRL_REQUIRE(n_steps >= 1 && first_step >= 1, "vae_learn_steps: n_steps=%d first_step=%d", n_steps, first_step);
  for (int k = 0; k < n_steps; ++k) {
    HHRL_REQUIRE(B[k] > 0 && N[k] > 0, "vae_learn_steps: step %d: B=%d N=%d", k, B[k], N[k]);
    const AdamK ak = make_adam_k(lr, beta1, beta2, adam_eps, first_step + k, 1.0);
    const float scale = (float)(1.0 / ((double)B[k] * (double)N[k]));
    HHRL_TRY(vae_step_impl(params, grads, m, v, &ak, offsets, n_flat, data[k], eps[k], B[k], N[k], g, scale, loss[k], ws[k], stream));
  }
  return HHRL_OK;
}

// ---------------------------------------------------------------- a training step that allocates nothing
// For callers that cannot allocate device memory (code under HIP graph capture, tight serving loops,
// harnesses that forbid it): hhrl_vae_train_step with its workspace in a file-scope device array instead of
// a caller-provided buffer.  Every GPU the library runs on holds its own copy of the array until the process
// ends; the array's name is only a host-side handle, so its device address is looked up once per device
// (hipGetSymbolAddress with that device current) and that address is what the kernels and the memset get.
static __device__ __align__(256) float scratch[HHRL_VAE_SCRATCH_FLOATS];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mu;
float* g_scratch_ptr[kMaxDevices];          // device address of `scratch` on device d (nullptr: not looked up yet)
}  // namespace

int hhrl_vae_train_step_noalloc(const float* params, float* grads, const long* offsets, long n_flat, const float* data,
                                const float* eps, int B, int N, int g, float recon_scale, float* loss, void* stream) {
  HHRL_REQUIRE(B > 0 && N > 0 && g > 0 && n_flat > 0, "vae_train_step_noalloc: bad shape B=%d N=%d g=%d n_flat=%ld", B, N, g, n_flat);
  const long need = carve(nullptr, B, N, g, n_flat).total;
  HHRL_REQUIRE(need <= (long)HHRL_VAE_SCRATCH_FLOATS,
               "vae_train_step_noalloc: B=%d lanes x N=%d cars needs %ld workspace floats, the built-in scratch holds %ld "
               "(use hhrl_vae_train_step with a workspace of your own for larger batches)",
               B, N, need, (long)HHRL_VAE_SCRATCH_FLOATS);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return fail(HHRL_ERR_HIP, "vae_train_step_noalloc: hipGetDevice failed");
  HHRL_REQUIRE(dev >= 0 && dev < kMaxDevices, "vae_train_step_noalloc: device %d out of range", dev);
  hipStream_t st = as_stream(stream);
  float* ws = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (g_scratch_ptr[dev] == nullptr) {
      // The address lookup is a set-up call of the runtime, not stream work: it must not happen while `st` is
      // recording a graph.  A caller that captures makes one call on the device outside the capture first.
      hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
      if (hipStreamIsCapturing(st, &cap) != hipSuccess) return fail(HHRL_ERR_HIP, "vae_train_step_noalloc: hipStreamIsCapturing failed");
      HHRL_REQUIRE(cap == hipStreamCaptureStatusNone,
                   "vae_train_step_noalloc: the first call on device %d resolves the scratch array's address and cannot run "
                   "under stream capture; call once on this device outside the capture first", dev);
      float* p = nullptr;
      const hipError_t e = hipGetSymbolAddress(reinterpret_cast<void**>(&p), HIP_SYMBOL(scratch));
      if (e != hipSuccess || p == nullptr)
        return fail(HHRL_ERR_HIP, "vae_train_step_noalloc: hipGetSymbolAddress(scratch) on device %d: %s", dev, hipGetErrorString(e));
      g_scratch_ptr[dev] = p;
    }
    ws = g_scratch_ptr[dev];
  }
  // The array keeps the last call's contents, another (B, N, g, n_flat) carves it differently, and the step reads
  // the layout's padding (slab columns nobody writes) as zeros.  The part this call uses is therefore cleared by
  // EVERY call, as stream work in front of the kernels: what the array holds then never depends on which calls
  // the host made before -- a captured call replays correctly after eager calls of other shapes in between.
  if (hipMemsetAsync(ws, 0, (size_t)need * sizeof(float), st) != hipSuccess)
    return fail(HHRL_ERR_HIP, "vae_train_step_noalloc: clearing the scratch array failed");
  return vae_step_impl(const_cast<float*>(params), grads, nullptr, nullptr, nullptr, offsets, n_flat, data, eps, B, N, g,
                       recon_scale, loss, ws, stream);
}

static int lane_layers(LaneArgs& la, const float* params, const long* offsets, long n_flat, int g, const char* who) {
  static const long numel_w[NLAYER] = {32, 512, 4096, 8192, -64, -64, -128, 4096, 32, 8192, 4096, 32};
  static const long numel_b[NLAYER] = {16, 32, 128, 64, -1, -1, 128, 32, 32, 128, 32, 1};
  HHRL_REQUIRE(g == 2 || g == 4 || g == 8 || g == 16, "%s: latent size %d not in {2,4,8,16}", who, g);
  for (int l = 0; l < NLAYER; ++l) {
    const long ow = offsets[2 * l], ob = offsets[2 * l + 1];
    const long nw = numel_w[l] < 0 ? -numel_w[l] * g : numel_w[l];
    const long nb = numel_b[l] < 0 ? g : numel_b[l];
    HHRL_REQUIRE(ow >= 0 && ow + nw <= n_flat && ob >= 0 && ob + nb <= n_flat, "%s: offsets[%d] outside the flat buffer", who, l);
    HHRL_REQUIRE((ow & 3) == 0, "%s: weight %d is not 16-byte aligned inside the flat buffer", who, l);
    la.w[l] = params + ow;
    la.b[l] = params + ob;
  }
  return HHRL_OK;
}

int hhrl_vae_encode(const float* params, const long* offsets, long n_flat, const float* cars, const int* lane_start,
                    int B, int N, int g, float* out, int ldo, void* stream) {
  HHRL_REQUIRE(params && offsets && cars && out, "vae_encode: null pointer");
  HHRL_REQUIRE(B > 0 && (lane_start != nullptr || N > 0) && ldo >= 2 * g, "vae_encode: bad shape B=%d N=%d ldo=%d", B, N, ldo);
  EncArgs a;
  HHRL_TRY(lane_layers(a.g, params, offsets, n_flat, g, "vae_encode"));
  a.g.data = cars;
  a.g.eps = nullptr; a.g.u = nullptr; a.g.aux = nullptr; a.g.act = nullptr; a.g.wimg = nullptr; a.g.spart = nullptr; a.g.slab = nullptr; a.g.kl_part = nullptr;
  a.g.B = B; a.g.N = N; a.g.g = g; a.g.tiles = (B + 31) / 32;
  a.lane_start = lane_start;
  a.out = out;
  a.ldo = ldo;
  HHRL_TRY(ensure_dyn_lds(reinterpret_cast<const void*>(vae_encode_kernel), LN_LDS_FLOATS * 4, "vae_encode"));
  hipStream_t st = as_stream(stream);
  ProfScope ps("vae_encode_kernel", st, 0.0, 0.0);
  hipLaunchKernelGGL(vae_encode_kernel, dim3(grid_lane(B)), dim3(256), LN_LDS_FLOATS * 4, st, a);
  return launch_status("vae_encode_kernel");
}

int hhrl_vae_decode(const float* params, const long* offsets, long n_flat, const float* goal, int ldg, const float* eps,
                    const float* loc, int ldl, int B, int N, int g, float* out, int ldo, void* stream) {
  HHRL_REQUIRE(params && offsets && goal && loc && out, "vae_decode: null pointer");
  HHRL_REQUIRE(B > 0 && N > 0 && ldg >= 2 * g && ldl >= N && ldo >= N, "vae_decode: bad shape B=%d N=%d ldg=%d ldl=%d ldo=%d", B, N, ldg,
               ldl, ldo);
  LaneArgs la;
  HHRL_TRY(lane_layers(la, params, offsets, n_flat, g, "vae_decode"));
  DecArgs a;
  for (int l = 0; l < NLAYER; ++l) {
    a.w[l] = la.w[l];
    a.b[l] = la.b[l];
  }
  a.goal = goal; a.eps = eps; a.loc = loc; a.out = out;
  a.rows = (long)B * N;
  a.ldg = ldg; a.ldl = ldl; a.ldo = ldo; a.N = N; a.g = g;
  HHRL_TRY(ensure_dyn_lds(reinterpret_cast<const void*>(vae_decode_kernel), DEC_LDS_FLOATS * 4, "vae_decode"));
  const long tiles = (a.rows + 31) / 32;
  hipStream_t st = as_stream(stream);
  ProfScope ps("vae_decode_kernel", st, 0.0, 0.0);
  hipLaunchKernelGGL(vae_decode_kernel, dim3((unsigned)(tiles < 1024 ? tiles : 1024)), dim3(256), DEC_LDS_FLOATS * 4, st, a);
  return launch_status("vae_decode_kernel");
}

#ifdef HHRL_STAMPS
// diagnostic builds only: copies the recorded clocks (4 kernels x 64 stamps) to the host
int hhrl_debug_stamps(long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(hhrl::vf::g_stamps), sizeof(long long) * 4 * 64);
}
int hhrl_debug_wg_spans(long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(hhrl::vf::g_wg_span), sizeof(long long) * 256 * 4);
}
#endif

}  // extern "C"
