"""VAE training path on the GPU (through the C ABI) against the golden vectors
recorded from the reference and, at the benchmark sizes (4096 lanes), against the CPU oracle on
the same inputs.  Bar: fp32, rtol 1e-4 (plus an absolute floor of 2e-6 of each
tensor's largest magnitude, since fp32 sums in a different order cannot meet a
pure rtol at values near zero)."""
import os

import numpy as np
import pytest
import torch

import gen
import hhrl_oracle as O
from golden_util import assert_close, load_case, sub
from test_gpu_kernels import scaled_close

pytestmark = pytest.mark.gpu


def _enc(g, init, fused=True):
    from hhrl_vae_amd.algorithm import Encoder
    e = Encoder(2, g, fused=fused)
    e.load_state_dict(init)
    return e


def _np(sd):
    return {k: v.numpy() for k, v in sd.items()}


def dict_close(name, got, want, **kw):
    assert set(want) <= set(got)
    for k in want:
        scaled_close(name + "/" + k, got[k], want[k], **kw)


LR = 1e-3


def params_close(name, got, want, steps):
    """Parameters after `steps` chained Adam updates: rtol 1e-4, plus 0.5 % of
    one Adam step (lr) per update as absolute allowance.  Adam divides the
    gradient by its own running magnitude, so a last-bit difference in a small
    gradient element moves that weight by a visible fraction of lr although
    the gradient itself is well within rtol."""
    assert set(want) <= set(got)
    for k in want:
        w = np.asarray(want[k], dtype=np.float64)
        assert_close(name + "/" + k, got[k], w, rtol=1e-4, atol=2e-6 * float(np.abs(w).max()) + 5e-3 * LR * steps)


@pytest.mark.parametrize("g", [2, 4, 8, 16])
def test_vae_golden_steps(g):
    fx = load_case("vae_g%d" % g)
    e = _enc(g, sub(fx, "init"))
    for step in range(3):
        loss = e.learn(fx["step%d/x" % step].tolist(), eps=fx["step%d/eps" % step])
        assert_close("loss%d" % step, loss, fx["step%d/loss" % step], rtol=1e-4, atol=0)
        if step == 0 and g == 4:
            dict_close("grad0", _np(e.grads()), sub(fx, "step0/grad"))
            params_close("param0", _np(e.state_dict()), sub(fx, "step0/param"), 1)
    from golden_util import adam_path_close
    vae = O.VAE({k: torch.from_numpy(np.asarray(v)) for k, v in sub(fx, "init").items()}, g)
    grads_by_step = []
    for step in range(3):
        vae.learn(fx["step%d/x" % step], torch.from_numpy(fx["step%d/eps" % step]))
        grads_by_step.append({k: v.numpy().copy() for k, v in vae.grads.items()})
    adam_path_close("final", _np(e.state_dict()), sub(fx, "final/param"), grads_by_step, [LR] * 3, max_saturated=0.05, zero_is_exact=True)
    if g == 4:
        dict_close("exp_avg", _np(e.opt.m.state_dict()), sub(fx, "final/exp_avg"))
        dict_close("exp_avg_sq", _np(e.opt.v.state_dict()), sub(fx, "final/exp_avg_sq"))


@pytest.mark.parametrize("tag", ["edge_n1", "edge_b1", "edge_n59"])
def test_vae_golden_ragged(tag):
    """One car per lane, one lane per batch, and the longest lane the state
    log can hold (59 cars)."""
    fx = load_case("vae_g4")
    e = _enc(4, sub(fx, "final/param"))
    loss = e.loss_and_grads(fx[tag + "/x"], fx[tag + "/eps"])
    assert_close("loss", float(loss), fx[tag + "/loss"], rtol=1e-4, atol=0)
    dict_close("grad", _np(e.grads()), sub(fx, tag + "/grad"))


def test_vae_golden_encode_decode():
    fx = load_case("vae_g4")
    e = _enc(4, sub(fx, "final/param"))
    scaled_close("encode", e.encode(fx["encode/x"].tolist()), fx["encode/out"])
    empty = e.encode([])
    assert empty.shape == (8,) and not empty.any()
    scaled_close("decode", e.decode(fx["decode/goal"], fx["decode/loc"], eps=fx["decode/eps"]), fx["decode/out"])


def test_vae_golden_batch_loop():
    """The chunked, shuffled one-pass training loop of VAE_train.py:55-69."""
    from hhrl_vae_amd.vae_train import chunk_dataset, train_one_pass
    fx = load_case("vae_loop_g2")
    dataset = {k: v.tolist() for k, v in sub(fx, "bucket").items()}
    chunks = chunk_dataset(dataset, int(fx["batch_size"]))
    order = fx["order"].tolist()
    assert [len(chunks[i]) for i in order] == fx["chunk_sizes"].tolist()
    e = _enc(2, sub(fx, "init"))
    losses = train_one_pass(e, [chunks[i] for i in order], eps_list=[fx["eps%d" % s] for s in range(len(order))])
    assert_close("losses", losses, fx["losses"], rtol=1e-4, atol=0)
    # parameters after the seven steps: the bar per element is built from the gradient that element had at every step
    # (the CPU oracle's, which is pinned to the same fixture), instead of a blanket share of lr per step
    from golden_util import adam_path_close
    vae = O.VAE({k: torch.from_numpy(np.asarray(v)) for k, v in sub(fx, "init").items()}, 2)
    grads_by_step = []
    for s_, ci in enumerate(order):
        vae.learn(chunks[ci], torch.from_numpy(fx["eps%d" % s_]))
        grads_by_step.append({k: v.numpy().copy() for k, v in vae.grads.items()})
    adam_path_close("final", _np(e.state_dict()), sub(fx, "final/param"), grads_by_step, [LR] * len(order), max_saturated=0.05, zero_is_exact=True)


def test_vae_train_all_converts_chunks_once(tmp_path):
    """train_all (VAE_train.py:55-69): the shuffled chunks are turned into device tensors once and replayed
    for every latent size; each model's loss sequence is what a fresh model gets from the same chunks as
    nested lists, bit for bit, and the reference's three files per model are written."""
    import os
    import random
    from hhrl_vae_amd.algorithm import Encoder
    from hhrl_vae_amd.vae_train import chunk_dataset, shuffle_chunks, train_all, train_one_pass
    rs = np.random.RandomState(9)
    dataset = {str(n): rs.uniform(0, 1, size=(cnt, n, 2)).tolist() for n, cnt in ((1, 70), (5, 64), (17, 30))}
    out = train_all(dataset, goal_dims=(2, 8), batch_size=64, save_dir=str(tmp_path), shuffle_rng=random.Random(4), sync_every=8)
    chunks = shuffle_chunks(chunk_dataset(dataset, 64), random.Random(4))
    assert [len(c) for c in chunks].count(6) == 1 and len(chunks) == 4              # 64 + 6, 64, 30
    for g in (2, 8):
        enc, losses = out[g]
        assert len(losses) == len(chunks) and enc.opt.step_count == len(chunks)
        assert losses == train_one_pass(Encoder(2, g), chunks)
        for suf in ("_enc", "_dec", "_encoder_optimizer"):
            assert os.path.exists(os.path.join(str(tmp_path), "encoder_dim_%d%s" % (g, suf)))


def test_inference_calls_staged_and_in_place_agree(monkeypatch):
    """encode / encode_lanes / decode with the calls' pinned blocks read and written in place by the kernels (the
    default for small blocks) and staged through device buffers (HHRL_ZERO_COPY=0): the same bits."""
    from hhrl_vae_amd.algorithm import Encoder
    rs = np.random.RandomState(4)
    lanes = [rs.uniform(0, 1, size=(n, 2)).tolist() for n in (7, 1, 0, 30, 12)]
    goal, eps = rs.standard_normal(8).astype(np.float32) * 0.3, rs.standard_normal((1, 4)).astype(np.float32)

    def run():
        e = Encoder(2, 4, seed=9)
        assert e._io.n_in == 0
        out = [e.encode(lanes[0]), e.encode_lanes(lanes), e.decode(goal, [0.4], eps=eps), e.encode(lanes[3])]
        return e, out

    a, ra = run()
    assert a._io.zc_in and a._io.zc_out, "pinned blocks of this size are expected to be used in place"
    monkeypatch.setenv("HHRL_ZERO_COPY", "0")
    b, rb = run()
    assert not b._io.zc_in and not b._io.zc_out
    for x, y in zip(ra, rb):
        assert np.array_equal(np.asarray(x), np.asarray(y))


def test_vae_learn_many_equals_one_call_per_chunk():
    """Encoder.learn_many (hhrl_vae_learn_steps: a group of chunks per library call) takes exactly the steps of
    learn() chunk by chunk: ragged chunk shapes, more chunks than the loss ring holds, injected noise and the
    encoder's own; losses, parameters and Adam state bit for bit."""
    from hhrl_vae_amd.algorithm import Encoder
    rs = np.random.RandomState(11)
    shapes = [(64, 7), (64, 1), (6, 17), (64, 30), (1, 3), (33, 12)] * 50               # 300 chunks > LOSS_RING
    chunks = [torch.from_numpy(rs.uniform(0, 1, size=(b, n, 2)).astype(np.float32)).cuda() for b, n in shapes]
    for inject in (False, True):
        a, b = Encoder(2, 4, seed=3), Encoder(2, 4, seed=3)
        eps = [rs.standard_normal((c.shape[0], 4)).astype(np.float32) for c in chunks] if inject else None
        la = [a.learn(c, eps=None if eps is None else eps[i], sync=False).item() for i, c in enumerate(chunks)]
        lb = []
        for lo in range(0, len(chunks), 100):                     # fetched in groups, as the training loop does
            lb += [t.item() for t in b.learn_many(chunks[lo:lo + 100], None if eps is None else eps[lo:lo + 100])]
        assert la == lb
        assert a.opt.step_count == b.opt.step_count == len(chunks)
        sa, sb = a.state_dict(), b.state_dict()
        for k in sa:
            assert torch.equal(sa[k], sb[k]), k
        assert torch.equal(a.opt.m.data, b.opt.m.data) and torch.equal(a.opt.v.data, b.opt.v.data)


def test_reference_training_script_body_runs_unchanged(tmp_path):
    """The train-mode body of the reference's script (VAE_train.py:52-69) with only its import swapped to
    `hhrl_vae_amd.algorithm.Encoder`: state log read from disk in the script's own format, chunks of 64,
    shuffle, `Encoder(2, goal_dim).to(device)`, one `learn` per chunk with the loss handed to a tensorboard-like
    writer, `save`.  The statements below are the script's call sequence; the writer and the paths are local."""
    import json
    import os
    import random
    from hhrl_vae_amd.algorithm import Encoder
    from hhrl_vae_amd.vae_train import read_state_log, train_one_pass
    rs = np.random.RandomState(3407)
    log = {str(n): rs.uniform(0, 1, size=(cnt, n, 2)).tolist() for n, cnt in ((0, 0), (1, 70), (4, 130), (9, 64), (23, 17))}
    log_path = os.path.join(str(tmp_path), "0814_sample_2.txt")
    with open(log_path, "w") as f:
        json.dump(log, f)

    class Writer:
        def __init__(self):
            self.scalars = []

        def add_scalar(self, tag, value, step):
            self.scalars.append((tag, float(value), step))
    writer = Writer()
    device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    my_dataset = read_state_log(log_path)
    batch_size = 64
    dataset = {k: [v[i: i + batch_size] for i in range(0, len(v), batch_size)] for k, v in my_dataset.items() if len(v) > 0}
    shuffled_dataset = [item for sublist in dataset.values() for item in sublist]
    random.Random(3407).shuffle(shuffled_dataset)
    for goal_dim in [2, 4, 8, 16]:
        cnt, loss = 0, 0
        encoder = Encoder(2, goal_dim).to(device)
        for data in shuffled_dataset:
            loss = encoder.learn(data)
            writer.add_scalar(str(goal_dim) + '_loss', loss, cnt)
            cnt += 1
        encoder.save(os.path.join(str(tmp_path), 'encoder_dim_' + str(goal_dim)))
    n_chunks = 2 + 3 + 1 + 1
    assert len(shuffled_dataset) == n_chunks and len(writer.scalars) == 4 * n_chunks
    for goal_dim in [2, 4, 8, 16]:
        got = [v for tag, v, _ in writer.scalars if tag == str(goal_dim) + '_loss']
        assert all(np.isfinite(got)) and isinstance(got[0], float)
        assert got == train_one_pass(Encoder(2, goal_dim), shuffled_dataset)      # same seeds, same steps: same bits
        for suf in ("_enc", "_dec", "_encoder_optimizer"):
            assert os.path.exists(os.path.join(str(tmp_path), "encoder_dim_%d%s" % (goal_dim, suf)))
    e = Encoder(2, 4)
    assert e.to("cuda") is e and e.eval() is e and e.train() is e and len(e.parameters()) == 24
    with pytest.raises(RuntimeError):
        e.to("cpu")


def adam_conditioned_close(name, got_p, want_p, want_g, lr, g_floor, rtol=1e-4):
    """Parameters after an Adam step.  Adam normalises the gradient, so an
    early step moves a weight by about lr * g/|g|: a gradient element known
    only to within dg (its fp32 summation noise, g_floor * max|g| of the
    tensor) makes the step uncertain by about lr * dg/|g|, capped at one full
    step (2.2 * lr) where the sign itself is undetermined.  That term is added
    to the rtol bar element by element."""
    for k in want_p:
        g = np.abs(np.asarray(want_g[k], dtype=np.float64))
        dg = g_floor * max(float(g.max()), 1e-30)
        gp, wp = np.asarray(got_p[k], dtype=np.float64), np.asarray(want_p[k], dtype=np.float64)
        tol = rtol * np.abs(wp) + 2e-6 * float(np.abs(wp).max()) + lr * np.minimum(2.2, 6.0 * dg / (g + 1e-30))
        err = np.abs(gp - wp)
        assert np.all(err <= tol), "%s/%s: %d elements out, worst excess %.3e" % (name, k, int((err > tol).sum()), float((err - tol).max()))


@pytest.mark.parametrize("B,N,g", [(4096, 7, 4), (4096, 30, 4), (1000, 3, 16), (4096, 30, 2), (4096, 30, 8), (4096, 30, 16)])
def test_vae_full_size_vs_oracle(B, N, g):
    """Benchmark batch sizes (4096 lanes x 7 and x 30 cars): loss, every gradient, Adam moments and updated
    parameters against the CPU oracle on identical inputs, for two consecutive
    steps (each step starts from the oracle's exact state)."""
    shapes = O.vae_param_shapes(g)
    init = gen.gen_state(shapes, 1234 + g)
    oracle = O.VAE({k: torch.from_numpy(v) for k, v in init.items()}, g)
    e = _enc(g, init)
    for step in range(2):
        e.load_state_dict({k: v.detach() for k, v in oracle.p.items()})
        e.opt.m.load(dict(zip(oracle.names, oracle.opt.exp_avg)))
        e.opt.v.load(dict(zip(oracle.names, oracle.opt.exp_avg_sq)))
        e.opt.step_count = oracle.opt.step_count
        x = gen.gen_lanes(B, N, 77 + step)
        eps = torch.randn(B, g, generator=torch.Generator().manual_seed(3407 + step))
        # fp64 yardstick at the same parameters: how far the fp32 oracle itself sits from the exact sums
        o64 = O.VAE({k: v.detach() for k, v in oracle.p.items()}, g, dtype=torch.float64)
        _, g64 = o64.loss_and_grads(torch.from_numpy(x), eps)
        want = oracle.learn(x, eps)
        got = e.learn(torch.from_numpy(x).cuda(), eps=eps)
        assert_close("loss", got, want, rtol=1e-4, atol=0)
        og = {k: v.numpy() for k, v in oracle.grads.items()}
        gg = _np(e.grads())
        # Weight gradients here are sums over B*N rows of terms that largely cancel, and a ReLU whose
        # input rounds to the other side of zero moves a whole term.  Both fp32 evaluations (this one and
        # the CPU oracle) land up to ~2e-4 of a tensor's largest entry away from the fp64 value on single
        # elements (tests/diag_vae_accuracy.py prints the per-tensor distances: e.g. 2.1e-7 for the oracle on
        # decoder.func.0.weight at B=4096, N=7, largest entry 1.1e-3).  The absolute floor is therefore
        # 10 * 2^-24 * sqrt(rows) of the tensor's largest entry (2.1e-4 at 122,880 rows), or three times
        # the fp32 oracle's own worst distance from the fp64 value on that tensor, whichever is larger;
        # elements above the floor meet rtol 1e-4.
        floor = max(1e-5, 10 * 2.0 ** -24 * (B * N) ** 0.5)
        gm = _np(e.opt.m.state_dict())
        om = dict(zip(oracle.names, [t.numpy() for t in oracle.opt.exp_avg]))
        for k in og:
            noise = float(np.max(np.abs(og[k].astype(np.float64) - g64[k].numpy())))
            scale = float(np.max(np.abs(og[k])))
            # accuracy: within 3x of the fp32 oracle's own distance from the fp64 value; agreement with the
            # fp32 oracle then follows from the triangle inequality (that allowance + the oracle's distance)
            assert_close("grad%d/%s vs fp64" % (step, k), gg[k], g64[k].numpy(), rtol=1e-4, atol=max(floor * scale, 3 * noise))
            assert_close("grad%d/%s" % (step, k), gg[k], og[k], rtol=1e-4, atol=max(floor * scale, 3 * noise) + noise)
            # first moment m = 0.9 m_prev + 0.1 g from identical m_prev: a tenth of the gradient's allowance
            assert_close("exp_avg%d/%s" % (step, k), gm[k], om[k], rtol=1e-4, atol=0.1 * (max(floor * scale, 3 * noise) + noise) + 1e-12)
        adam_conditioned_close("param%d" % step, _np(e.state_dict()), {k: v.detach().numpy() for k, v in oracle.p.items()}, og, 1e-3, floor)


@pytest.mark.parametrize("B,N,g", [(1, 1, 2), (3, 1, 4), (31, 7, 4), (33, 7, 8), (64, 7, 16), (5, 33, 4), (2, 59, 2),
                                   (257, 5, 4), (1000, 3, 16), (4096, 7, 4), (4096, 30, 4), (2048, 59, 8), (9000, 2, 4), (20000, 7, 4),
                                   (37, 15, 4), (37, 16, 4), (40, 17, 2), (21, 18, 8)])
def test_vae_fused_matches_layerwise(B, N, g):
    """The fused four-launch step against the one-launch-per-layer path on the
    same device and inputs: ragged batch sizes, one car per lane, lanes longer
    than a 32-car tile, every latent size."""
    init = gen.gen_state(O.vae_param_shapes(g), 500 + B + N)
    ef, el = _enc(g, init, fused=True), _enc(g, init, fused=False)
    assert ef.fused and not el.fused
    x = torch.from_numpy(gen.gen_lanes(B, N, 9)).cuda()
    eps = torch.randn(B, g, generator=torch.Generator().manual_seed(B * N)).cuda()
    lf, ll = float(ef.loss_and_grads(x, eps)), float(el.loss_and_grads(x, eps))
    assert_close("loss", lf, ll, rtol=1e-4, atol=0)
    # Two fp32 evaluations in different summation orders (accuracy itself is judged against the oracle and its fp64
    # yardstick in test_vae_full_size_vs_oracle).  Small batches: 1e-5 of a tensor's largest entry.  From 4096 cars
    # on: 3e-5 for the longer sums, and an allowance for ReLU flips: among millions of hidden pre-activations one or
    # two lie within rounding of zero, the two paths may put such a value on different sides of the ReLU, and that
    # switches ONE car's whole contribution through that unit -- a change of the order of 1/(B*N) of a sum over all
    # cars, spread over the gradient tensors upstream of the unit (tests/diag_vae_fused_vs_layerwise.py: 2.4e-4 of
    # decoder.0's largest weight gradient at 4096 x 7 = 28672 cars, nothing beyond 1e-5 at 20000 x 7 or 2048 x 59).
    # The allowance is therefore 8/(B*N) per element, and at most 5 % of all elements may use it.
    gf, gl = _np(ef.grads()), _np(el.grads())
    if B * N < 4096:
        dict_close("grad", gf, gl, atol_rel=1e-5)
    else:
        base, flip = 3e-5, 8.0 / (B * N)
        dict_close("grad", gf, gl, atol_rel=base + flip)
        beyond = total = 0
        for k in gl:
            a, b = gf[k].astype(np.float64), gl[k].astype(np.float64)
            beyond += int((np.abs(a - b) > 1e-4 * np.abs(b) + base * np.abs(b).max()).sum())
            total += b.size
        assert beyond <= 0.05 * total, "%d of %d gradient elements beyond the summation-order bar" % (beyond, total)
    # same inputs, same bits: every sum of the fused step runs in a fixed order
    g1 = _np(ef.grads())
    ef.loss_and_grads(x, eps)
    g2 = _np(ef.grads())
    assert all(np.array_equal(g1[k], g2[k]) for k in g1)


def test_vae_properties_full_size():
    """Size-independent properties at B=4096: (a) permuting the lanes of a
    batch (with their noise rows) leaves loss and gradients unchanged up to
    summation order; (b) permuting the cars inside a lane does too (DeepSets
    invariance of the encoder + per-car decoder); (c) gradients of two half
    batches scaled by the global count add up to the full-batch gradient."""
    g, B, N = 4, 4096, 7
    init = gen.gen_state(O.vae_param_shapes(g), 99)
    e = _enc(g, init)
    x = torch.from_numpy(gen.gen_lanes(B, N, 5)).cuda()
    eps = torch.randn(B, g, generator=torch.Generator().manual_seed(1)).cuda()
    l0 = float(e.loss_and_grads(x, eps))
    g0 = _np(e.grads())
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(2)).cuda()
    l1 = float(e.loss_and_grads(x[perm], eps[perm]))
    assert_close("lane-perm loss", l1, l0, rtol=1e-5, atol=0)
    dict_close("lane-perm grad", _np(e.grads()), g0, atol_rel=1e-5)
    cperm = torch.randperm(N, generator=torch.Generator().manual_seed(3)).cuda()
    l2 = float(e.loss_and_grads(x[:, cperm], eps))
    assert_close("car-perm loss", l2, l0, rtol=1e-5, atol=0)
    dict_close("car-perm grad", _np(e.grads()), g0, atol_rel=1e-5)
    h = B // 2
    la = float(e.loss_and_grads(x[:h], eps[:h], recon_count=B * N))
    ga = _np(e.grads())
    lb = float(e.loss_and_grads(x[h:], eps[h:], recon_count=B * N))
    gb = _np(e.grads())
    assert_close("shard loss", la + lb, l0, rtol=1e-5, atol=0)
    dict_close("shard grad", {k: ga[k] + gb[k] for k in ga}, g0, atol_rel=1e-5)


def test_vae_checkpoint_roundtrip(tmp_path):
    """save/load writes the reference's three files with its key names
    (algorithm.py:129-138) and restores parameters and Adam state."""
    fx = load_case("vae_g4")
    e = _enc(4, sub(fx, "init"))
    e.learn(fx["step0/x"], eps=fx["step0/eps"])
    prefix = os.path.join(str(tmp_path), "encoder_dim_4")
    e.save(prefix)
    enc_sd = torch.load(prefix + "_enc", weights_only=True)
    assert list(enc_sd)[:2] == ["phi.0.weight", "phi.0.bias"] and "to_logvar.bias" in enc_sd
    opt_sd = torch.load(prefix + "_encoder_optimizer", weights_only=True)
    assert len(opt_sd["param_groups"][0]["params"]) == 24 and float(opt_sd["state"][0]["step"]) == 1.0
    e2 = _enc(4, sub(fx, "init"))
    e2.load(prefix)
    for step in (1, 2):
        a = e.learn(fx["step%d/x" % step], eps=fx["step%d/eps" % step])
        b = e2.learn(fx["step%d/x" % step], eps=fx["step%d/eps" % step])
        assert_close("loss", b, a, rtol=1e-6, atol=0)
    params_close("final", _np(e2.state_dict()), sub(fx, "final/param"), 3)


@pytest.mark.parametrize("g", [2, 4, 8, 16])
def test_vae_encode_decode_fused_vs_oracle(g):
    """hhrl_vae_encode / hhrl_vae_decode (one launch each) against the CPU oracle: lanes of different
    lengths incl. empty and 59-car lanes, uniform batches, batched decode over several locations."""
    init = gen.gen_state(O.vae_param_shapes(g), 321 + g)
    st = {k: torch.from_numpy(v) for k, v in init.items()}
    e = _enc(g, init)
    rs = np.random.RandomState(g)
    counts = [7, 0, 1, 59, 16, 17, 3, 0, 33] + [int(c) for c in rs.randint(1, 40, size=60)]
    lanes = [gen.gen_lanes(1, c, 100 + i)[0] if c else np.zeros((0, 2), dtype=np.float32) for i, c in enumerate(counts)]
    got = e.encode_lanes(lanes)
    assert got.shape == (len(lanes), 2 * g)
    for i, lane in enumerate(lanes):
        if len(lane) == 0:
            assert not got[i].any()
            continue
        mean, logvar = O.deepset_encoder_forward(st, torch.from_numpy(lane))
        scaled_close("encode_lanes[%d]" % i, got[i], torch.cat([mean, logvar], -1).numpy(), atol_rel=5e-6)
        if i < 6:
            scaled_close("encode[%d]" % i, e.encode(lane.tolist()), got[i], atol_rel=5e-6)
    xb = gen.gen_lanes(300, 9, 5)
    mean, logvar = O.deepset_encoder_forward(st, torch.from_numpy(xb))
    scaled_close("encode_batch", e.encode_batch(torch.from_numpy(xb)).cpu().numpy(), torch.cat([mean, logvar], -1).numpy(), atol_rel=5e-6)
    # decode: B lanes x N locations
    B, N = 45, 5
    goal = rs.standard_normal((B, 2 * g)).astype(np.float32) * 0.5
    loc = rs.uniform(0, 1, size=(B, N)).astype(np.float32)
    eps = rs.standard_normal((B, g)).astype(np.float32)
    z = torch.from_numpy(goal[:, :g]) + torch.from_numpy(eps) * torch.exp(0.5 * torch.from_numpy(goal[:, g:]))
    want = O.deepset_decoder_forward(st, z.unsqueeze(1).repeat(1, N, 1), torch.from_numpy(loc).unsqueeze(-1)).squeeze(-1)
    scaled_close("decode_batch", e.decode_batch(goal, loc, eps).cpu().numpy(), want.numpy(), atol_rel=5e-6)
    one = e.decode(goal[3], [loc[3, 2]], eps=eps[3])
    scaled_close("decode", one, want[3, 2:3].numpy(), atol_rel=5e-6)
    # the per-layer path gives the same numbers
    el = _enc(g, init, fused=False)
    scaled_close("decode_batch per-layer", el.decode_batch(goal, loc, eps).cpu().numpy(), want.numpy(), atol_rel=5e-6)


def test_vae_train_step_rejects_misaligned_layout():
    """The fused step reads weight rows 16 bytes at a time: a layout whose weight offsets are not
    multiples of four floats is refused before any launch, with a message."""
    import ctypes
    from hhrl_vae_amd import _lib as L
    lib = L.load()
    from hhrl_vae_amd.algorithm import Encoder
    enc = Encoder(2, 4, device="cuda:0", seed=3)
    B, N, g = 8, 3, 4
    x = torch.rand(B, N, 2, device="cuda:0")
    eps = torch.randn(B, g, device="cuda:0")
    n_flat = enc.layout.size
    ws = torch.zeros(lib.hhrl_vae_workspace_floats(B, N, g, n_flat), device="cuda:0")
    loss = torch.zeros(1, device="cuda:0")
    off = list(enc._offsets)
    bad = (ctypes.c_long * len(off))(*off)
    bad[18] += 2                                   # func.0 weight
    rc = lib.hhrl_vae_train_step(enc.params.base_ptr, enc.opt.g.base_ptr, bad, n_flat, x.data_ptr(), eps.data_ptr(),
                                 B, N, g, 1.0 / (B * N), loss.data_ptr(), ws.data_ptr(), None)
    assert rc != 0 and b"aligned" in lib.hhrl_last_error()
    good = (ctypes.c_long * len(off))(*off)
    rc = lib.hhrl_vae_train_step(enc.params.base_ptr, enc.opt.g.base_ptr, good, n_flat, x.data_ptr(), eps.data_ptr(),
                                 B, N, g, 1.0 / (B * N), loss.data_ptr(), ws.data_ptr(), None)
    torch.cuda.synchronize()
    assert rc == 0 and np.isfinite(float(loss[0]))
