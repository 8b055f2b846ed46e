"""Host-side logic that needs no GPU: flat buffer layouts, the StepLR rule,
the VAE batch loop's bucketing, family shapes, replay layouts."""
import random

import numpy as np
import pytest
import torch

import gen
import hhrl_oracle as O
from golden_util import load_case, sub


def test_flat_layout_and_views():
    from hhrl_vae_amd.flat import ALIGN, FlatBuffer, Layout
    from hhrl_vae_amd.vae import vae_shapes
    for g, n_params in ((2, 30389), (4, 30905), (8, 31937), (16, 34001)):      # parameter counts of Encoder(2, g)
        lay = Layout(vae_shapes(g))
        assert lay.numel() == n_params and len(lay.names) == 24
        assert all(off % ALIGN == 0 for off in lay.offsets.values()) and lay.size % ALIGN == 0
    lay = Layout(vae_shapes(4))
    assert [n for n, _ in O.vae_param_shapes(4)] == lay.names
    fb = FlatBuffer(lay, "cpu")
    init = gen.gen_state(vae_shapes(4), 1)
    fb.load(init)
    for k, v in fb.state_dict().items():
        assert np.array_equal(v.numpy(), init[k])
    fb["encoder.phi.0.bias"].fill_(2.0)                        # views alias the flat storage
    off = lay.offsets["encoder.phi.0.bias"]
    assert torch.all(fb.data[off: off + 16] == 2.0) and float(fb.data[off + 16]) == 0.0
    with pytest.raises(KeyError):
        fb.load({"nope": np.zeros(3)})
    with pytest.raises(ValueError):
        fb.load(dict(init, **{"encoder.phi.0.bias": np.zeros(3, dtype=np.float32)}))


def test_family_shapes_match_reference_counts():
    from hhrl_vae_amd.flat import Layout
    from hhrl_vae_amd.spec import actor_shapes, critic_shapes, family_from_cfg
    light, cav = gen.light_cfg(), gen.cav_cfg()
    fam = family_from_cfg("triple", light)
    assert [Layout(actor_shapes(fam, k)).numel() for k in range(3)] == [351529, 352432, 364720]
    assert Layout(critic_shapes(fam)).numel() == 1216898 and fam.row_width == 1030 and fam.critic_in == 393
    w = family_from_cfg("worker", cav)
    assert Layout(actor_shapes(w, 0)).numel() == 18945 and Layout(critic_shapes(w)).numel() == 38146 and w.row_width == 36
    assert Layout(critic_shapes(family_from_cfg("single", light, "time"))).numel() == 524530
    assert Layout(critic_shapes(family_from_cfg("manager", light))).numel() == 539218
    assert Layout(critic_shapes(family_from_cfg("double", light, "time"))).numel() == 902402
    assert Layout(critic_shapes(family_from_cfg("tp", light))).numel() == 890114
    for kind, task, cfg in (("triple", None, light), ("worker", None, cav), ("double", "phase", light), ("tp", None, light),
                            ("manager", None, light), ("single", "time", light)):
        f, of = family_from_cfg(kind, cfg, task), O.family_from_cfg(kind, cfg, task)
        for k in range(len(f.a_dims)):
            assert actor_shapes(f, k) == O.actor_param_shapes(of, k)
        assert critic_shapes(f) == O.critic_param_shapes(of)
        assert f.row_width == of.row_width


def test_steplr_rule_matches_torch():
    from hhrl_vae_amd.optim import StepLR

    class FakeOpt:
        lr = base_lr = 1e-3
    p = torch.nn.Parameter(torch.zeros(1))
    topt = torch.optim.Adam([p], lr=1e-3)
    tsch = torch.optim.lr_scheduler.StepLR(topt, step_size=3, gamma=0.5)
    fo = FakeOpt()
    sch = StepLR(fo, 3)
    for _ in range(11):
        topt.step()
        tsch.step()
        sch.step()
        assert fo.lr == pytest.approx(topt.param_groups[0]["lr"], rel=1e-15)


def test_steplr_keeps_a_restored_lr_like_torch():
    """After an optimizer state with an already-decayed lr is loaded, torch's StepLR goes on halving THAT lr
    (algorithm.py:1375-1382 restores param_groups; the scheduler object only counts calls).  Checked against
    torch itself and against the lr sequence the reference produced after its own load()
    (tests/golden/ckpt/ref_worker_decayed_expect.npz, written by make_ckpt.py)."""
    import os
    from hhrl_vae_amd.optim import StepLR

    class FakeOpt:
        lr = base_lr = 1e-3
    p = torch.nn.Parameter(torch.zeros(1))
    src = torch.optim.Adam([p], lr=1e-3)
    ssch = torch.optim.lr_scheduler.StepLR(src, step_size=2, gamma=0.5)
    for _ in range(5):
        src.step()
        ssch.step()
    saved = src.state_dict()
    assert saved["param_groups"][0]["lr"] == 2.5e-4
    topt = torch.optim.Adam([p], lr=1e-3)
    tsch = torch.optim.lr_scheduler.StepLR(topt, step_size=2, gamma=0.5)
    topt.load_state_dict(saved)
    fo = FakeOpt()
    sch = StepLR(fo, 2)
    fo.lr, fo.base_lr = saved["param_groups"][0]["lr"], saved["param_groups"][0]["initial_lr"]   # FlatAdam.load_state_dict
    for _ in range(7):
        topt.step()
        tsch.step()
        sch.step()
        assert fo.lr == topt.param_groups[0]["lr"]
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ckpt", "ref_worker_decayed_expect.npz"))
    crit, act = FakeOpt(), FakeOpt()
    crit.lr, act.lr = float(fx["saved/critic_lr"]), float(fx["saved/actor_lr"])
    assert crit.lr == 2.5e-4 and act.lr == 2.5e-5                    # the checkpoint carries decayed rates
    cs, as_ = StepLR(crit, 2), StepLR(act, 1)
    for call, (want_c, want_a) in enumerate(fx["next/lrs"]):
        cs.step()
        if (call + 1) % 2 == 0:                                      # actor scheduler: policy steps only (algorithm.py:1353-1366)
            as_.step()
        assert (crit.lr, act.lr) == (want_c, want_a), call


def test_vae_chunking_matches_golden_loop():
    from hhrl_vae_amd.vae_train import chunk_dataset, shuffle_chunks
    fx = load_case("vae_loop_g2")
    dataset = {k: v.tolist() for k, v in sub(fx, "bucket").items()}
    chunks = chunk_dataset(dataset, int(fx["batch_size"]))
    assert [len(chunks[i]) for i in fx["order"]] == fx["chunk_sizes"].tolist()
    assert chunks == O.vae_chunk_dataset(dataset, int(fx["batch_size"]))
    assert all(len({len(lane) for lane in c}) == 1 for c in chunks)            # one car count per chunk
    a = shuffle_chunks(chunks, random.Random(3407))
    order = list(range(len(chunks)))
    random.Random(3407).shuffle(order)
    assert a == [chunks[i] for i in order] and order == fx["order"].tolist()
    assert chunk_dataset({"3": [], "1": [[[0.1, 0.2]]]}, 64) == [[[[0.1, 0.2]]]]


def test_shard_bounds_cover_batch_once():
    from hhrl_vae_amd.dist import shard_bounds
    for n in (1, 7, 64, 4096, 4097):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _closed_form_policy(w, c):
    return lambda o, g: np.tanh((np.asarray(o, dtype=np.float64) * w).sum(axis=1, keepdims=True) + np.asarray(g, dtype=np.float64) * c)


@pytest.mark.parametrize("impl", ["oracle", "product"])
def test_off_policy_correction_matches_reference(impl):
    """ManagerTD3.off_policy_corrections against the reference's output (tests/golden/make_opc.py)
    with a closed-form worker policy: same global numpy seed, bit-equal corrected goals.  Covers lanes
    with no controlled vehicle and the lane whose only marked row is row 0."""
    fx = load_case("td3_manager_opc")
    if impl == "oracle":
        fn = O.off_policy_corrections
    else:
        from hhrl_vae_amd.td3 import ManagerTD3
        fn = ManagerTD3.off_policy_corrections
    calls = []

    def policy(o, g):
        calls.append(len(o))
        return _closed_form_policy(fx["fn/w"], float(fx["fn/c"]))(o, g)
    np.random.seed(int(fx["fn/seed"]))
    got = fn(policy, fx["fn/goals"], fx["fn/worker_obs"], fx["fn/worker_act"])
    assert got.shape == fx["fn/corrected"].shape == (6, 8)
    assert np.array_equal(got, fx["fn/corrected"])
    assert np.array_equal(got[:, 2], fx["fn/goals"][:, 2].astype(np.float64))       # all candidates tie -> stored goal kept
    if impl == "product":
        assert calls == [8 * 10 * 6 * 25]                        # every lane and candidate in one policy call


def test_copy_queue_reads_a_fetched_block_in_place():
    """ops.CopyQueue with a staging block that the same launch fetches: whatever a copy (or the row numbers of a
    gather) would read out of the block's device copy is redirected to the pinned block, copies of gathered rows go to
    the replay table, everything else is left alone; the fetch pieces come first, then the gather of the rows."""
    from hhrl_vae_amd import ops
    PIN, DEV, ROWS, TABLE, OUT = 0x10000, 0x90000, 0x50000, 0x200000, 0x700000
    W = 36
    fetch = ((PIN + 4 * 512, DEV + 4 * 512, 64), (PIN + 4 * 100, DEV + 4 * 100, 28))
    q = ops.CopyQueue(8, gather=(ROWS, W, TABLE, 1000, DEV + 4 * 520), fetch=fetch)
    assert q.items[:2] == [(PIN + 4 * 512, 0, DEV + 4 * 512, 0, 64, 2), (PIN + 4 * 100, 0, DEV + 4 * 100, 0, 28, 2)]
    assert q.items[2] == (TABLE, W, ROWS, W, W, 1)
    q.add(ROWS + 4 * 5, W, OUT, 12, 7)                   # a column range of the gathered rows: straight from the table
    q.add(DEV + 4 * 110, 3, OUT + 64, 3, 3)              # inside the second fetched piece: from the pinned block
    q.add(DEV + 4 * 300, 3, OUT + 128, 3, 3)             # in the device block but not fetched by this launch: as given
    q.add(0x400000, 9, OUT + 256, 9, 9)                  # elsewhere: as given
    q.add_zero(OUT + 512, 40)
    q.add_normals(OUT + 1024, 16, PIN + 4 * 530)
    assert q.items[3:] == [(TABLE + 4 * 5, W, OUT, 12, 7, 1), (PIN + 4 * 110, 3, OUT + 64, 3, 3, 0),
                           (DEV + 4 * 300, 3, OUT + 128, 3, 3, 0), (0x400000, 9, OUT + 256, 9, 9, 0),
                           (None, 0, OUT + 512, 0, 40, 0), (PIN + 4 * 530, 0, OUT + 1024, 0, 16, 3)]
    assert q._pinned(DEV + 4 * 520) == PIN + 4 * 520 and q._pinned(DEV + 4 * 576) == DEV + 4 * 576      # row numbers; one past the piece
    plain = ops.CopyQueue(8)
    plain.add(DEV + 4 * 110, 3, OUT, 3, 3)
    assert plain.items == [(DEV + 4 * 110, 3, OUT, 3, 3, 0)]
