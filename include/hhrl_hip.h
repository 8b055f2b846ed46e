/* hhrl_hip.h — C ABI of libhhrl_hip.so, the MI355X (gfx950) kernels behind the
 * HHRL-VAE learning path.
 *
 * The reference has no FFI: its boundary is the Python class surface of
 * src/algorithm.py.  Each entry point below names the reference statements it
 * replaces (file:line under the reference's src/).  A Python host binds these
 * with ctypes (see hhrl-vae_amd/_lib.py); no torch type crosses the boundary.
 *
 * Conventions
 *   - every pointer is DEVICE memory, fp32, row-major, borrowed for the call;
 *   - "ld*" is a leading dimension in ELEMENTS (row stride), so callers can
 *     read from / write into column slices of wider buffers (concatenations
 *     are never materialised by a copy);
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream);
 *     all work is enqueued asynchronously on it, nothing synchronises;
 *   - return value: HHRL_OK, or an error code; hhrl_last_error() returns a
 *     thread-local message for the last non-zero return;
 *   - weights use nn.Linear layout W[out][in]; LSTM gate order is i,f,g,o.
 */
#ifndef HHRL_HIP_H
#define HHRL_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define HHRL_OK 0
#define HHRL_ERR_ARG 1 /* null pointer, non-positive size, bad enum, ld too small */
#define HHRL_ERR_HIP 2 /* a HIP runtime call or kernel launch failed */

#define HHRL_ACT_NONE 0
#define HHRL_ACT_RELU 1
#define HHRL_ACT_TANH 2
#define HHRL_ACT_SIGMOID 3

/* Library identification / diagnostics. */
/* The ABI version this header describes; hhrl_version() returns the one the library was built with.  A binding checks
 * that they agree before it passes descriptor structs (their layout is part of the ABI).
 * 101: hhrl_lstm_desc gained `no_backward`; hhrl_gather_copy_group, hhrl_randn_fill, hhrl_vae_learn_steps added.
 * 102: hhrl_mlp3_backward_group_adam (hhrl_adam_ride), hhrl_td3_target_mse_zeroed added.
 * 103: hhrl_mlp3_desc gained the mse_* fields; hhrl_mlp3_slab_floats is positive for every M (one more entry per row).
 * 104: hhrl_gcopy_desc.gather gained 2 (flat copy), 3 (standard normals) and 4 (scattered rows); row numbers may lie in
 *      pinned memory. */
#define HHRL_ABI_VERSION 104
int hhrl_version(void);
const char* hhrl_last_error(void);

/* ------------------------------------------------------------------ layers
 * y[M,N] = act( (accumulate ? y : 0) + x[M,K] * w[N,K]^T + b[N] )
 * Replaces nn.Linear + ReLU/Tanh/Sigmoid: algorithm.py:21-25, 27-31, 37-38,
 * 53-66, 169-171, 177-179, 199-206, 214-220, 1197-1199, 1225-1232.
 * b may be NULL.  fp32 MFMA (v_mfma_f32_32x32x2_f32), LDS-staged tiles. */
int hhrl_linear_forward(const float* x, int ldx, const float* w, const float* b,
                        float* y, int ldy, int M, int K, int N, int act,
                        int accumulate, void* stream);

/* Backward of the layer above (autograd of the same lines; loss.backward()
 * at algorithm.py:105, 311, 319, 513, 530, 806, 823, 1151, 1159, 1348, 1357).
 *   dy  : in  dL/dy (post-activation).  With act != NONE its contents after the
 *         call are unspecified: products of up to 512 output tiles of 32x32 apply
 *         the activation gradient while they load dy and leave it as it was,
 *         larger ones first rewrite it in place to dL/d(pre-activation);
 *   y   : the forward output (needed when act != NONE, else may be NULL);
 *   dx  : dL/dx, written (dx_accumulate=0) or added to; NULL to skip;
 *   dw,db: ADDED to (torch .grad semantics; the caller zeroes them); either
 *          may be NULL to skip.  dW is reduced over M with a split reduction
 *          and float atomics. */
int hhrl_linear_backward(const float* x, int ldx, const float* w,
                         const float* y, int ldy, float* dy, int lddy,
                         float* dx, int lddx, int dx_accumulate,
                         float* dw, float* db, int M, int K, int N, int act,
                         void* stream);
/* The input gradient of such a layer when dL/dy is the SAME row for every sample, scaled by alpha:
 *   dx[i][k] (op)= alpha * sum_n dy_row[n] * act'(y[i][n]) * w[n][k].
 * That is the case of the layer below a one-output layer whose own output gradient is a constant: for the actor loss
 * -mean(Q1) (algorithm.py:317, 528, 821, 1157, 1355) dQ1/d(hidden 2) is the row of l3's weights for every sample and
 * alpha = -1/batch, so the last layer's input gradient takes no launch of its own (nor, in that step, its forward value).
 * No parameter gradients (the policy step does not update the critic). */
int hhrl_linear_backward_row(const float* w, const float* y, int ldy, const float* dy_row, float alpha,
                             float* dx, int lddx, int dx_accumulate, int M, int K, int N, int act, void* stream);
/* Columns [c0, c0 + nc) of a layer's input gradient only:
 *   dx[i][c0 + c] (op)= sum_n dy[i][n] * act'(y[i][n]) * w[n][c0 + c]      (w: [N, K] as in hhrl_linear_forward).
 * The actor loss is differentiated with respect to ONE actor's action columns of the critic's input
 * (algorithm.py:820-823: only optimizer i steps), not with respect to the states and the other actions beside them. */
int hhrl_linear_backward_cols(const float* w, int K, const float* y, int ldy, const float* dy, int lddy,
                              float* dx, int lddx, int dx_accumulate, int c0, int nc, int M, int N, int act, void* stream);

/* Several independent dense layers on the same M rows in ONE launch per stage
 * (the twin Q heads l1/l4, l2/l5, l3/l6 of the critics, algorithm.py:206-221,
 * 381-400, 650-672, 1236-1254): a batch-64 update is bound by the number of
 * launches.  Same meaning as the arguments of hhrl_linear_forward/backward.
 * Backward runs the stages activation-gradient + bias gradient, input
 * gradients (overwriting layers first, accumulating layers in a second launch:
 * an accumulating layer may add to a dx that an overwriting layer of the same
 * call wrote), weight gradients. */
#define HHRL_LINEAR_GROUP_MAX 16
typedef struct {
  const float* x; int ldx;
  const float* w; const float* b;
  float* y; int ldy;
  int K, N, act, accumulate;
  /* optional (NULL = off): the target-policy smoothing of hhrl_td3_smooth_action applied to the layer's output in the
   * same launch: y2[row*ldy2 + col] = clamp(y + clamp(0.2*smooth_eps[row*N + col], -0.5, 0.5), -1, 1) */
  const float* smooth_eps; float* y2; int ldy2;
} hhrl_linear_fwd_desc;
typedef struct {
  const float* x; int ldx;
  const float* w;
  const float* y; int ldy;
  float* dy; int lddy;
  float* dx; int lddx; int dx_accumulate;
  float* dw; float* db;
  int K, N, act;
} hhrl_linear_bwd_desc;
int hhrl_linear_forward_group(const hhrl_linear_fwd_desc* layers, int n, int M, void* stream);
int hhrl_linear_backward_group(const hhrl_linear_bwd_desc* layers, int n, int M, void* stream);

/* ------------------------------------------------------------ data movement
 * out[b, c] = sum_n in[(b*N+n), c]          (torch.sum(x, dim=-2), algorithm.py:42;
 *                                            also the gradient of the latent repeat, :101) */
int hhrl_sum_pool(const float* in, int ldi, float* out, int ldo, int B, int N, int C, void* stream);
/* out[(b*N+n), c] = in[b, c]                (z.unsqueeze(-2).repeat(1,N,1), algorithm.py:101;
 *                                            also the gradient of the sum-pool, :42) */
int hhrl_broadcast_rows(const float* in, int ldi, float* out, int ldo, int B, int N, int C, void* stream);
/* out[m, c] = in[m, c] for a [M,C] block    (torch.cat pieces, algorithm.py:71, 212, 383, 652, 1207, 1238) */
int hhrl_copy2d(const float* in, int ldi, float* out, int ldo, int M, int C, void* stream);
/* Several such copies of M rows each in one launch (every torch.cat piece of an update that
 * depends only on the sampled rows). */
#define HHRL_COPY_GROUP_MAX 32
/* A block with in == NULL is a zero fill instead: out[0..C) = 0, one contiguous run (ldi, ldo and M do not apply):
 * optimizer.zero_grad() (algorithm.py:310, 318, 512, 529, 805, 822, 1150, 1158, 1347, 1356) riding in the same launch. */
typedef struct { const float* in; int ldi; float* out; int ldo; int C; } hhrl_copy_desc;
int hhrl_copy2d_group(const hhrl_copy_desc* blocks, int n, int M, void* stream);
/* out[r][0..C) = in[idx[r]][0..C) for r < M: the minibatch gather `self.memory[indices, :]`
 * (algorithm.py:287, 483, 772, 1320) when the replay buffer is mirrored in device memory
 * (in: [n_in][ldi] fp32, idx: M int32 row numbers; numbers outside [0, n_in) are clamped). */
int hhrl_gather_rows(const float* in, long ldi, int n_in, const int* idx, float* out, int ldo, int M, int C,
                     void* stream);
/* hhrl_copy2d_group whose blocks may gather: a block with gather != 0 reads row idx[r] of its source (in: a column of
 * the replay buffer, ldi: the buffer's row width) instead of row r; row numbers outside [0, n_in) are clamped.  The
 * minibatch gather of hhrl_gather_rows, every torch.cat piece that is a column range of the sampled rows and the
 * zero_grad fills of an update as ONE launch (algorithm.py:287, 483, 772, 1320 and the lines cited above).
 * gather == 2: a flat copy of C floats from in to out (ldi, ldo and M do not apply), e.g. a caller's pinned staging
 * block on its way into device memory in the same launch.  gather == 3: out[0..C) = elements offset .. offset + C - 1
 * of the standard-normal stream of hhrl_randn_fill, where `in` points at two 64-bit integers {seed, offset} (offset a
 * multiple of four) that the launch reads from memory: the draws of `torch.randn_like` for the target-policy smoothing
 * (algorithm.py:790, 1333) inside a replayed launch.  gather == 4: rows stored since the last update
 * (`store_transition`, algorithm.py:273, 465, 750, 1299) on their way into the device-resident replay buffer `out`
 * ([n_in][ldo]): `in` points at {int count (<= 63); int position[63]} and, 64 floats further, the rows [count][C];
 * row k goes to out[position[k]].  The caller sees to it that no block of the launch gathers one of these positions.
 * `idx` and the sources may lie in pinned host memory that the device can address: the row numbers are read once per
 * tile of rows. */
typedef struct { const float* in; long ldi; float* out; int ldo; int C; int gather; } hhrl_gcopy_desc;
int hhrl_gather_copy_group(const hhrl_gcopy_desc* blocks, int n, int M, const int* idx, int n_in, void* stream);
/* p[i] = value */
int hhrl_fill(float* p, long n, float value, void* stream);
/* out[i] = element (offset + i) of the standard-normal stream of `seed`, i < n: the draws of `torch.randn_like`
 * (Encoder.reparameterize, algorithm.py:89) made on the device by a counter-based generator (Philox4x32-10 and
 * Box-Muller).  An element depends on (seed, offset + i) only: consecutive fills continue one stream, the same
 * (seed, offset) gives the same values on every run.  The reference's own draws come from torch's global
 * generator and cannot be reproduced; callers that need given values pass their own eps to the step instead. */
int hhrl_randn_fill(float* out, long n, unsigned long long seed, unsigned long long offset, void* stream);

/* --------------------------------------------------------------------- VAE
 * z = mean + eps * exp(0.5*logvar)          (Encoder.reparameterize, algorithm.py:87-90;
 *                                            eps is supplied by the caller instead of randn_like)
 * ml is [B, 2g] = mean || logvar. */
int hhrl_vae_reparam(const float* ml, const float* eps, float* z, int B, int g, void* stream);

/* ELBO and its gradient at the decoder output (algorithm.py:102-104):
 *   loss += recon_scale * sum((decoded - v)^2) - 0.5*sum(1 + logvar - mean^2 - exp(logvar))
 *   d_decoded = 2*recon_scale*(decoded - v)
 * data is [B*N, 2] (col 1 = v); recon_scale = 1/(B*N) on one device, 1/(global B*N)
 * on a data-parallel shard.  *loss must be zeroed by the caller (it is added to).
 * Wave-level shuffle reductions, one atomic per workgroup. */
int hhrl_vae_elbo(const float* decoded, const float* data, const float* ml, float recon_scale,
                  float* loss, float* d_decoded, int B, int N, int g, void* stream);

/* Gradient at the encoder heads (autograd of algorithm.py:87-90, 103):
 *   d_mean   = dz + mean
 *   d_logvar = dz*eps*0.5*exp(0.5*logvar) + 0.5*(exp(logvar) - 1)       -> d_ml [B,2g] */
int hhrl_vae_latent_grad(const float* dz, const float* ml, const float* eps, float* d_ml,
                         int B, int g, void* stream);

/* One whole training step's forward + ELBO + backward, fused
 * (Encoder.learn minus the optimiser, algorithm.py:94-105):
 *   params/grads : flat fp32 buffers of n_flat floats holding the 24 tensors of
 *                  Encoder.parameters() at offsets[0..23] (HOST array, element
 *                  offsets, order weight,bias per layer: phi.0 phi.2 rho.0 rho.2
 *                  to_mean to_logvar decoder.0 decoder.2 loc func.0 func.2 func.4);
 *   data [B*N,2] (loc, v), eps [B,g]; g in {2,4,8,16};
 *   grads is OVERWRITTEN with dL/dparams (fixed-order reduction: the same
 *   inputs give the same bits), *loss with recon_scale*sum(err^2) + KL;
 *   ws: hhrl_vae_workspace_floats(B,N,g,n_flat) floats, zero-filled once by the
 *   caller and then left to the library;
 *   params and ws 16-byte aligned, every weight offset a multiple of 4 floats.
 * Four launches: per-lane forward, per-car forward+backward, per-lane
 * backward, slab reduction; all matrix products on fp32 MFMA with LDS-resident
 * weights.  func.0 (algorithm.py:61-70) over [d2 of the lane | loc embedding] is
 * evaluated in factored form: its d2 half once per lane, its loc half as a
 * rank-two term; same sums, another order. */
long hhrl_vae_workspace_floats(int B, int N, int g, long n_flat);
int hhrl_vae_train_step(const float* params, float* grads, const long* offsets, long n_flat,
                        const float* data, const float* eps, int B, int N, int g,
                        float recon_scale, float* loss, float* ws, void* stream);
/* The whole of Encoder.learn (algorithm.py:94-106) on one device: the step
 * above with the Adam update (algorithm.py:83, 106; torch defaults) applied by
 * the slab reduction, which holds each gradient element in a register anyway.
 *   m, v: Adam moments, flat like params; step: 1-based count of this update.
 * grads still receives the gradient. Not for data-parallel runs: those
 * all-reduce grads between hhrl_vae_train_step and hhrl_adam_step. */
int hhrl_vae_learn_step(float* params, float* grads, float* m, float* v, const long* offsets, long n_flat,
                        const float* data, const float* eps, int B, int N, int g, float recon_scale,
                        double lr, double beta1, double beta2, double adam_eps, int step,
                        float* loss, float* ws, void* stream);

/* n_steps consecutive hhrl_vae_learn_step calls in ONE library call: the chunk loop of the reference's
 * training script (VAE_train.py:65-68: one optimizer step per chunk of <= 64 lanes, chunks of different car
 * counts) without a return to the host language between steps -- at that chunk size the host's per-call cost
 * exceeds the step's GPU time.  Step k trains on data[k] = [B[k], N[k], 2] with noise eps[k] = [B[k], g] and
 * workspace ws[k] (hhrl_vae_workspace_floats(B[k], N[k], g, n_flat) floats; steps of one shape may share one),
 * divides its squared error by B[k]*N[k] (nn.MSELoss, algorithm.py:84, 102), takes Adam step number
 * first_step + k and writes its loss to loss[k].  The pointer arrays are host arrays of device pointers. */
int hhrl_vae_learn_steps(float* params, float* grads, float* m, float* v, const long* offsets, long n_flat, int n_steps,
                         const float* const* data, const float* const* eps, const int* B, const int* N, int g,
                         double lr, double beta1, double beta2, double adam_eps, int first_step,
                         float* const* loss, float* const* ws, void* stream);

/* hhrl_vae_train_step for callers that CANNOT ALLOCATE device memory (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same arguments and
 * results as hhrl_vae_train_step (bit for bit) without `ws`: the workspace is a device array built into the
 * library, HHRL_VAE_SCRATCH_FLOATS floats (8 MiB) per GPU, held by every GPU the library runs on until the
 * process ends.  The call allocates and frees nothing.  Limit: hhrl_vae_workspace_floats(B, N, g, n_flat)
 * must not exceed HHRL_VAE_SCRATCH_FLOATS (every chunk of the reference's training script fits: 64 lanes of
 * up to 59 cars need 1.2 M floats); larger batches are rejected with HHRL_ERR_ARG and a message.  All calls
 * on one GPU share that GPU's array: calls that overlap on one GPU (two streams, two threads) race.  Every call
 * clears the part of the array it uses as stream work ahead of its kernels (the array keeps the previous call's
 * contents), so a captured call replays correctly whatever shapes were run in between.  The first call on a
 * device looks the array's address up and is refused under stream capture: call once outside the capture first. */
#define HHRL_VAE_SCRATCH_FLOATS (2 * 1024 * 1024)
int hhrl_vae_train_step_noalloc(const float* params, float* grads, const long* offsets, long n_flat,
                                const float* data, const float* eps, int B, int N, int g,
                                float recon_scale, float* loss, void* stream);

/* Inference, one launch each.
 * Encoder.encode / DeepSetEncoder.forward (algorithm.py:40-47, 109-116) for B lanes:
 *   cars [total][2] (location, speed); lane_start [B+1] device ints = first car of every lane
 *   (lanes of different lengths), or NULL for N cars per lane; out [B, 2g] (ld ldo) = mean | logvar.
 *   A lane without cars gives zeros, as the reference's early return does.
 * Encoder.decode / DeepSetDecoder.forward (algorithm.py:68-73, 121-127) for B lanes x N locations:
 *   goal [B, 2g] (ld ldg) = mean | logvar; z = mean + eps*exp(logvar/2) with eps [B,g] from the
 *   caller (NULL: z = mean); loc [B,N] (ld ldl); out [B,N] (ld ldo) = decoded speed in (0,1). */
int hhrl_vae_encode(const float* params, const long* offsets, long n_flat, const float* cars,
                    const int* lane_start, int B, int N, int g, float* out, int ldo, void* stream);
int hhrl_vae_decode(const float* params, const long* offsets, long n_flat, const float* goal, int ldg,
                    const float* eps, const float* loc, int ldl, int B, int N, int g,
                    float* out, int ldo, void* stream);

/* --------------------------------------------------------------------- TD3
 * Target-policy smoothing (algorithm.py:295-296, 494-497, 786-791, 1135-1136, 1330-1331):
 *   out = clamp(pi + clamp(0.2*eps, -0.5, 0.5), -1, 1);  eps ~ N(0,1) from the caller. */
int hhrl_td3_smooth_action(const float* pi, int ldp, const float* eps, float* out, int ldo,
                           int B, int a, void* stream);
/* y = r + gamma*min(q1,q2)                  (algorithm.py:300-301, 502-503, 795-796, 1140-1141, 1336-1337)
 * r is a column of the sampled replay rows: element b at r[b*ldr]. */
int hhrl_td3_target_q(const float* q1, const float* q2, const float* r, int ldr, float gamma, float* y,
                      int B, void* stream);
/* Twin critic loss and its gradient (F.mse_loss x2, algorithm.py:307, 509, 802, 1147, 1344):
 *   loss[0] += inv_n*sum((q1-y)^2); loss[1] += inv_n*sum((q2-y)^2);
 *   dq1 = 2*inv_n*(q1-y); dq2 = 2*inv_n*(q2-y).   loss[2] zeroed by the caller. */
int hhrl_twin_mse(const float* q1, const float* q2, const float* y, float inv_n, float* loss,
                  float* dq1, float* dq2, int B, void* stream);
/* hhrl_td3_target_q followed by hhrl_twin_mse in one launch (same lines of algorithm.py): y is computed from the
 * target critic's q1t, q2t and the reward column and stored, then the two losses and their gradients follow.
 * loss[0..1] are OVERWRITTEN (no zeroing by the caller). */
int hhrl_td3_target_mse(const float* q1t, const float* q2t, const float* r, int ldr, float gamma,
                        const float* q1, const float* q2, float inv_n, float* y, float* loss,
                        float* dq1, float* dq2, int B, void* stream);
/* The same for a caller that has already set loss[0..1] to zero on the stream (e.g. inside the launch that stages
 * the minibatch, hhrl_gather_copy_group): large batches then take no memset node of their own. */
int hhrl_td3_target_mse_zeroed(const float* q1t, const float* q2t, const float* r, int ldr, float gamma,
                               const float* q1, const float* q2, float inv_n, float* y, float* loss,
                               float* dq1, float* dq2, int B, void* stream);

/* ------------------------------------------------- three dense layers, fused
 * y = act3(W3 relu(W2 relu(W1 x + b1) + b2) + b3) as one launch per direction for a GROUP of
 * networks on the same M rows, for hidden widths 128/128 (WorkerActor / WorkerCritic,
 * algorithm.py:1187-1258; both Q heads of a critic, or a network and its target, are separate
 * entries of one group).  hhrl_mlp3_supported tells whether a shape is covered (K0 <= 64,
 * H1 = H2 = 128, N3 <= 32); other shapes go through hhrl_linear_*.
 * Forward writes y and, when h1/h2 are given, the two hidden activations [M,128] that the backward
 * call reads.  Backward: dy is dL/dy (null: every element equals dy_const, the gradient of a mean);
 * the six gradient pointers are all given (gradients are ADDED) or all null (input gradient only);
 * dx, when given, receives dL/dx.
 * Weights stay in registers (a wave owns 16 output columns of each hidden layer), rows go through LDS in tiles
 * of 16; a launch has at most 256 workgroups, each looping over its tiles. */
#define HHRL_MLP3_GROUP_MAX 8
typedef struct {
  const float* x; int ldx; int K0;
  const float *w1, *b1, *w2, *b2, *w3, *b3;
  int N3, act3;
  float *h1, *h2;
  float* y; int ldy;
  const float* dy; int lddy; float dy_const;
  float *gw1, *gb1, *gw2, *gb2, *gw3, *gb3;
  float* dx; int lddx;
  /* forward only, optional: the target-policy smoothing of hhrl_td3_smooth_action applied to the output in the
   * same launch: y2[row*ldy2 + col] = clamp(y + clamp(0.2*smooth_eps[row*N3 + col], -0.5, 0.5), -1, 1) */
  const float* smooth_eps; float* y2; int ldy2;
  /* forward only, optional: nonzero = the launch also writes dx, the input gradient for dL/dy = dy_const on
   * every row (d(-mean Q1)/d(s, a) of a policy step, algorithm.py:1353-1357), and then does not save h1 / h2 */
  int forward_dx;
  /* backward only, optional: 16-byte aligned workspace of gslab_floats >= hhrl_mlp3_slab_floats(K0, N3, M) floats.
   * With it the workgroups hand their parameter gradients over as slab rows that a second launch sums in a fixed
   * order (same bits on every run) instead of adding them with float atomics. */
  float* gslab; long gslab_floats;
  /* backward only, optional (mse_q1t non-null; N3 = 1, y given, dy null): the network is a critic head whose loss is the
   * TD3 critic loss (algorithm.py:1329-1345 / 785-803).  The launch forms dL/dy itself, as hhrl_td3_target_mse does:
   * target = mse_r[row*mse_ldr] + mse_gamma * min(mse_q1t[row], mse_q2t[row]);  dL/dy = 2*mse_inv_n*(y - target);
   * *mse_loss += mse_inv_n * sum (y - target)^2 (zeroed by the caller; summed in a fixed order with a slab, by one
   * atomic addition per workgroup without); mse_target, when given, receives the target [M]. */
  const float *mse_q1t, *mse_q2t, *mse_r; int mse_ldr; float mse_gamma, mse_inv_n;
  float* mse_loss; float* mse_target;
} hhrl_mlp3_desc;
int hhrl_mlp3_supported(int K0, int H1, int H2, int N3);
long hhrl_mlp3_slab_floats(int K0, int N3, int M);   /* size of that workspace for M rows */
int hhrl_mlp3_forward_group(const hhrl_mlp3_desc* nets, int n, int M, void* stream);
int hhrl_mlp3_backward_group(const hhrl_mlp3_desc* nets, int n, int M, void* stream);
/* hhrl_mlp3_backward_group followed by the optimizer step of the group's networks: `*_optimizer.step()` right
 * after `backward()` (algorithm.py:806-808, 823-824, 1348-1350, 1356-1357) and, with `target`, the soft update of
 * the target network (:827-834, 1361-1364), i.e. hhrl_adam_step_dev / hhrl_adam_polyak_step_dev over the n floats
 * of the flat buffers.  The group's gradient tensors must be ALL of the flat gradient buffer `grads` apart from
 * zero padding (one optimizer whose parameters are exactly these networks).  When every network of the group hands
 * its gradients over as slab rows, the step is applied by the launch that sums the slab (no launch and no pass over
 * the buffers of its own); otherwise it follows as a separate launch.  Same bits either way. */
typedef struct {
  const float* grads;            /* flat gradient buffer: holds gw1..gb3 of every network of the group */
  float *params, *m, *v;         /* flat buffers of the same layout */
  long n;                        /* floats in each */
  const float* hyper_dev;        /* the eight scalars of the step in device memory (hhrl_adam_stage) */
  float* target; double tau;     /* optional (null: none): target = tau*params + (1-tau)*target */
} hhrl_adam_ride;
int hhrl_mlp3_backward_group_adam(const hhrl_mlp3_desc* nets, int n, int M, const hhrl_adam_ride* opt, void* stream);

/* -------------------------------------------------------------------- LSTM
 * Bidirectional one-layer LSTM over a window of T steps, last time index only,
 * followed by Linear(2H -> H)             (LSTM.forward, algorithm.py:152-156).
 * The forward direction runs T steps; the reverse direction contributes its
 * first step only (input x[:, T-1], zero state), as r_out[:, -1, :] does.
 * params: 10 device pointers in state_dict order:
 *   weight_ih_l0[4H,o] weight_hh_l0[4H,H] bias_ih_l0[4H] bias_hh_l0[4H]
 *   weight_ih_l0_reverse weight_hh_l0_reverse bias_ih_l0_reverse bias_hh_l0_reverse
 *   fc.weight[H,2H] fc.bias[H]
 * obs is [B, T*o] with leading dimension ldobs (a column slice of a replay
 * row block); out [B,H] with leading dimension ldo; ws is a workspace of
 * hhrl_lstm_workspace_floats(B,T,o,H) floats that must stay untouched until
 * the matching backward has run. */
long hhrl_lstm_workspace_floats(int B, int T, int o, int H);
int hhrl_lstm_forward(const float* obs, int ldobs, const float* const* params, float* out, int ldo,
                      float* ws, int B, int T, int o, int H, void* stream);
/* BPTT of the above.  d_out [B,H] (ld lddo) is dL/d(out).  grads: 10 device
 * pointers laid out like params; gradients are ADDED (weight_hh_l0_reverse
 * receives nothing: its input state is zero).  No gradient w.r.t. obs. */
int hhrl_lstm_backward(const float* const* params, const float* d_out, int lddo,
                       float* const* grads, float* ws, int B, int T, int o, int H, void* stream);

/* The same two operations for a GROUP of independent LSTM modules that share
 * B, T and H (the per-stream LSTMs of a critic, `self.lstm[k]` at
 * algorithm.py:366-371, 630-637; the actors' and targets' LSTMs evaluated on
 * the same minibatch): every stage is one launch for the whole group, which is
 * what bounds a batch-64 update.  Fields not used by a direction may be null. */
#define HHRL_LSTM_GROUP_MAX 16
typedef struct {
  const float* obs;            /* forward: [B, T*o] rows, leading dimension ldobs */
  int ldobs;
  int o;                       /* observation width per time step */
  const float* const* params;  /* 10 device pointers, as above */
  float* const* grads;         /* backward: 10 device pointers, gradients are ADDED */
  float* out;                  /* forward: [B,H], leading dimension ldo */
  int ldo;
  const float* d_out;          /* backward: [B,H], leading dimension lddo (equal for all instances) */
  int lddo;
  float* ws;                   /* hhrl_lstm_workspace_floats(B,T,o,H) floats per instance */
  int no_backward;             /* forward: nonzero = this evaluation will not be back-propagated (a target network under
                                * torch.no_grad, algorithm.py:785-796; the critic's streams on a policy step, :820-821, whose
                                * gradient is taken w.r.t. the actions only): the per-step gates, cell and hidden states are
                                * then not written to ws, and hhrl_lstm_backward_group must not follow on it */
} hhrl_lstm_desc;
int hhrl_lstm_forward_group(const hhrl_lstm_desc* descs, int n, int B, int T, int H, void* stream);
int hhrl_lstm_backward_group(const hhrl_lstm_desc* descs, int n, int B, int T, int H, void* stream);

/* --------------------------------------------------------------- optimiser
 * torch.optim.Adam.step with default betas/eps on one flat buffer
 * (algorithm.py:83, 106, 259, 266, 312, 320 ...): bias-corrected, no weight
 * decay, no amsgrad.  `step` is the 1-based step count AFTER this update;
 * lr comes from the host-side StepLR rule.  g is multiplied by grad_scale
 * first (1.0 normally; 1/world for an averaged all-reduce).  Hyper-parameters
 * are doubles: 1-beta and the bias corrections are formed in double on the
 * host, as torch forms them, and only then rounded to fp32 kernel scalars. */
int hhrl_adam_step(float* p, const float* g, float* m, float* v, long n, double lr,
                   double beta1, double beta2, double eps, int step, double grad_scale, void* stream);
/* The same launch also copies one float, *copy_dst = *copy_src: the data-parallel form of
 * Encoder.learn (algorithm.py:92-107) all-reduces the loss in a tail behind the flat gradient, and this
 * call hands it to the caller's loss slot together with optimizer.step() (:106) instead of a launch of
 * its own. */
int hhrl_adam_step_copy(float* p, const float* g, float* m, float* v, long n, double lr,
                        double beta1, double beta2, double eps, int step, double grad_scale,
                        const float* copy_src, float* copy_dst, void* stream);
/* The same update for use inside a captured HIP graph: the eight kernel scalars
 * (step size with bias correction, betas, 1-betas, sqrt(bias correction 2), eps,
 * gradient scale) live in DEVICE memory, 8 floats at hyper_dev.
 * hhrl_adam_stage() enqueues a one-workgroup kernel that writes them for the
 * given step (outside the graph, before each replay); hhrl_adam_step_dev() is
 * the capturable launch that reads them. */
int hhrl_adam_stage(float* hyper_dev, double lr, double beta1, double beta2, double eps, int step,
                    double grad_scale, void* stream);
int hhrl_adam_step_dev(float* p, const float* g, float* m, float* v, long n, const float* hyper_dev,
                       void* stream);
/* target = tau*p + (1-tau)*target          (algorithm.py:323-326, 534-539, 827-834, 1163-1166, 1361-1364) */
int hhrl_polyak_update(float* target, const float* p, long n, double tau, void* stream);
/* hhrl_adam_step_dev followed by hhrl_polyak_update(target, p, ...) in one pass over the parameters: the
 * optimizer step of a network (algorithm.py:320, 530, 824, 1159, 1357: `*_optimizer.step()`) and the soft
 * update of its target (same lines as above) when no launch in between reads the target. */
int hhrl_adam_polyak_step_dev(float* p, const float* g, float* m, float* v, long n, const float* hyper_dev,
                              float* target, double tau, void* stream);

/* ------------------------------------------------------------- diagnostics
 * Per-launch timing with HIP events recorded on the launch stream (no
 * counterpart in the reference; used by bench.py for roofline figures).
 * While enabled, every kernel launch inside the library is bracketed by two
 * events and tagged with its algorithmic flops and bytes.
 * hhrl_prof_report() SYNCHRONISES on the recorded events and writes one line
 * per kernel name: "name launches total_ms total_flops total_bytes".  It
 * returns the buffer size needed (including the terminating NUL). */
int hhrl_prof_enable(int on);
int hhrl_prof_reset(void);
long hhrl_prof_report(char* buf, long cap);

#ifdef __cplusplus
}
#endif
#endif /* HHRL_HIP_H */
