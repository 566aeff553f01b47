"""CPU tests of the caller-side rows:
  * oracle/caller_oracle.py pinned against fixtures produced by running the reference (g7-g10);
  * the product's host bookkeeping (ChunkLineage tables, clip counts, flip permutations) checked against the same
    fixtures by feeding its tables through the oracle's gather.
No HIP compute here (that is tests/test_hip_caller.py, -m gpu)."""
import os

import numpy as np
import pytest
import torch

from d3dp_amd.clips import clip_count, flip_perm
from d3dp_amd.data import ChunkLineage, build_pairs
from d3dp_amd.weights import H36M_JOINTS_LEFT as KL, H36M_JOINTS_RIGHT as KR, make_state_dict
from oracle import caller_oracle as co
from oracle import d3dp_oracle as orc


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def dataset(seed, lengths):
    """Same construction as tools/make_goldens.py::_gen_dataset (data, not reference code)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    p2 = [rng.uniform(-1, 1, (n, 17, 2)).astype(np.float32) for n in lengths]
    p3 = [(rng.standard_normal((n, 17, 3)) * 0.3).astype(np.float32) for n in lengths]
    cams = [rng.uniform(-1, 1, (9,)).astype(np.float32) for _ in lengths]
    return cams, p3, p2


# ---- N2 ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [60, 54, 27, 20, 100])
def test_clip_gather_oracle_matches_reference(golden_dir, n):
    g = load(golden_dir, "g7_clips")
    F = int(g["frames"])
    for d in ("2d", "3d"):
        got = co.clip_gather(g[f"seq{d}_{n}"], F)
        assert got.shape == g[f"clips{d}_{n}"].shape
        assert np.array_equal(got, g[f"clips{d}_{n}"])
    assert clip_count(n, F) == g[f"clips2d_{n}"].shape[0]


@pytest.mark.parametrize("n", [60, 54, 27, 100, 243, 244, 485])
def test_clip_scatter_oracle_round_trip(n):
    """gather -> scatter returns the sequence for every n >= F (the last partial clip overlaps its predecessor and
    must win only on its own tail frames)."""
    F = 27 if n < 200 else 243
    rng = np.random.default_rng(n)
    seq = rng.standard_normal((n, 17, 3)).astype(np.float32)
    clips = co.clip_gather(seq, F)                                   # (nc,F,J,3)
    pred = np.broadcast_to(clips[:, None, None], (clips.shape[0], 2, 3) + clips.shape[1:]).copy()
    out = co.clip_scatter(pred, n)
    assert out.shape == (2, 3, n, 17, 3)
    assert np.array_equal(out[1, 2], seq)


def test_flip_perm_is_the_reference_swap():
    x = np.arange(17)
    y = x.copy()
    y[KL + KR] = x[KR + KL]
    assert list(y) == flip_perm(KL, KR, 17)


# ---- N3: batches --------------------------------------------------------------------------------------------------------
def test_batches_oracle_matches_reference(golden_dir):
    g = load(golden_dir, "g8_batches")
    lengths = [int(v) for v in g["lengths"]]
    cams, p3, p2 = dataset(int(g["seed"]), lengths)
    k = 0
    for cam, b3, b2 in co.chunk_batches(4, cams, p3, p2, int(g["frames"]), True, True, KL, KR, KL, KR, n_epochs=2):
        assert np.array_equal(b2.astype(np.float32), g[f"b2_{k}"]) and np.array_equal(b3.astype(np.float32), g[f"b3_{k}"])
        assert np.array_equal(cam, g[f"cam_{k}"])
        k += 1
    assert k == int(g["n_batches_total"])
    j = 0
    for _, b3, b2 in co.chunk_batches(3, None, p3, p2, int(g["frames"]), False, False, None, None, None, None):
        assert np.array_equal(b2.astype(np.float32), g[f"plain_b2_{j}"]) and np.array_equal(b3.astype(np.float32), g[f"plain_b3_{j}"])
        j += 1
    assert j == int(g["plain_batches"])


def test_product_lineage_reproduces_reference_batches(golden_dir):
    """d3dp_amd.data.ChunkLineage (pairs, per-epoch permutation, batch slicing, int32 tables) + the gather semantics
    of d3dp_batch_gather restated in numpy == the reference generator's batches, two epochs, flips included."""
    g = load(golden_dir, "g8_batches")
    lengths = [int(v) for v in g["lengths"]]
    F = int(g["frames"])
    cams, p3, p2 = dataset(int(g["seed"]), lengths)
    lin = ChunkLineage(lengths, 4, F, shuffle=True, augment=True)
    assert lin.batch_num() == int(g["num_batches"]) and lin.num_frames() == int(g["num_frames"])
    pool2, pool3 = np.concatenate(p2), np.concatenate(p3)
    perm = np.array(flip_perm(KL, KR, 17))
    k = 0
    for _ in range(2):
        _, pairs = lin.next_pairs()
        table = lin.table(pairs)
        for b in range(lin.batch_num()):
            rows = table[b * 4:(b + 1) * 4]
            for i, (off, ln, start, flip) in enumerate(rows):
                fr = np.clip(start + np.arange(F), 0, ln - 1) + off
                a2, a3 = pool2[fr], pool3[fr]
                if flip:
                    a2, a3 = a2[:, perm].copy(), a3[:, perm].copy()
                    a2[..., 0] *= -1
                    a3[..., 0] *= -1
                assert np.array_equal(a2, g[f"b2_{k}"][i]) and np.array_equal(a3, g[f"b3_{k}"][i]), (k, i)
            k += 1
    assert k == int(g["n_batches_total"])


def test_build_pairs_quirks():
    # centred chunks: 70 frames at F=27 -> 3 chunks starting at -5
    p = build_pairs([70], 27, augment=False)
    assert p.tolist() == [[0, -5, 22, 0], [0, 22, 49, 0], [0, 49, 76, 0]]
    p = build_pairs([27, 20], 27, augment=True)
    assert p.tolist() == [[0, 0, 27, 0], [0, 0, 27, 1], [1, -3, 24, 0], [1, -3, 24, 1]]


# ---- N4 ---------------------------------------------------------------------------------------------------------------
def _g5_tensors(golden_dir):
    g5 = load(golden_dir, "g5_caller")
    F = int(g5["frames"])
    gt3 = np.stack([g5["seq3d"][s:s + F] for s in g5["starts"]]).copy()
    gt2 = np.stack([g5["seq2d"][s:s + F] for s in g5["starts"]])
    gt3[:, :, 0] = 0
    pred = g5["pred"].copy()
    pred[:, :, :, :, 0] = 0
    return pred, gt3, g5["reproj"], gt2


def test_pmpjpe_oracle_matches_reference(golden_dir):
    g = load(golden_dir, "g9_pmpjpe")
    pred, gt3, reproj, gt2 = _g5_tensors(golden_dir)
    m = co.p_mpjpe_metrics(pred, gt3, reproj, gt2)
    for name, key in (("J_Best", "e_jbest"), ("P_Best", "e_pbest"), ("P_Agg", "e_pagg"), ("J_Agg", "e_jagg")):
        assert np.allclose(m[name], g[key], rtol=0, atol=1e-6), (name, m[name], g[key])
    single = np.mean(np.linalg.norm(co.procrustes_align(pred[:, 0, 0].reshape(-1, 17, 3), gt3.reshape(-1, 17, 3))
                                    - gt3.reshape(-1, 17, 3), axis=-1))
    assert abs(single - float(g["single"])) < 1e-6
    refl = gt3.copy()
    refl[..., 0] *= -1
    e = co.procrustes_errors(pred, refl)
    p_best = e.transpose(1, 2, 0, 3, 4).reshape(pred.shape[1], pred.shape[2], -1).mean(-1).min(axis=1)
    assert np.allclose(p_best, g["e_pbest_reflected"], rtol=0, atol=1e-6)


# ---- N3: AdamW + loop -------------------------------------------------------------------------------------------------
def test_adamw_restatement_matches_torch():
    torch.manual_seed(0)
    p0 = torch.randn(1000)
    p_ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([p_ref], lr=3e-3, weight_decay=0.1)
    p, m, v = p0.clone(), torch.zeros(1000), torch.zeros(1000)
    for step in range(1, 6):
        g = torch.randn(1000) * (10.0 ** (step - 3))
        p_ref.grad = g.clone()
        opt.step()
        co.adamw_step(p, g, m, v, step, 3e-3)
        assert torch.equal(p, p_ref.detach()), step


def test_train_loop_oracle_matches_reference(golden_dir):
    g = load(golden_dir, "g10_train_loop")
    cs, dep, Fr = int(g["cs"]), int(g["dep"]), int(g["frames"])
    lengths = [int(v) for v in g["lengths"]]
    cams, p3, p2 = dataset(int(g["data_seed"]), lengths)
    batches = [(b3, b2) for _, b3, b2 in co.chunk_batches(4, cams, p3, p2, Fr, True, True, KL, KR, KL, KR, n_epochs=2)]
    n_it = len(g["losses"])
    assert len(batches) == n_it
    draws = [(g[f"t_{i}"], g[f"noise_{i}"]) for i in range(n_it)]
    params = orc.strip_prefix(make_state_dict(int(g["seed"]), cs, dep, Fr))
    losses, ps, lr = co.train_loop(params, dep, 1.0, batches, draws, float(g["lr"]), float(g["lr_decay"]), n_it // 2)
    assert np.allclose(losses, g["losses"], rtol=1e-5), (losses, g["losses"])
    assert abs(lr - float(g["final_lr"])) < 1e-12
    for k in g.files:
        if k.startswith("param::"):
            name = k[len("param::pose_estimator."):]
            assert np.allclose(ps[name].numpy(), g[k], rtol=0, atol=2e-6), name


# ---- N4: 3DHP caller side -------------------------------------------------------------------------------------------------
def test_3dhp_oracle_and_host_helpers_match_reference(golden_dir):
    from d3dp_amd import eval3dhp as e3
    g = load(golden_dir, "g11_3dhp")
    pred, gt, valid = g["pred"], g["gt"], g["valid"]
    # oracle
    assert np.allclose(co.mpjpe_diffusion_3dhp(pred, gt, valid), g["e_pbest"], rtol=1e-6)
    assert np.allclose(co.mpjpe_diffusion_3dhp(pred, gt, valid, mean_pos=True), g["e_pagg"], rtol=1e-6)
    absol = pred + g["traj"][:, None, None]
    assert np.allclose(co.project_linear(absol, g["cam1"]), g["reproj_linear"], rtol=1e-6, atol=1e-3)
    assert np.allclose(co.image_coordinates(g["x2d"], 2048, 2048), g["target_pix"], rtol=1e-6)
    assert np.array_equal(co.stitch_last_wins(pred[:, :, 0], int(g["n_frames"]), int(g["frames"])).transpose(3, 2, 1, 0), g["stitched"])
    assert np.array_equal(co.cam_mm_to_pix([7.32506, 7.32506, -0.0322884, 0.0929296, 0, 0, 0, 0, 0], [2048, 2048, 10, 10]), g["cam1"])
    # product host helpers (device-agnostic torch code paths; the kernels are covered by tests/test_hip_caller.py)
    assert np.array_equal(e3.camera_for("TS1")[0].numpy(), g["cam1"]) and e3.camera_for("TS3")[2] is True
    assert np.array_equal(e3.camera_for("TS5")[0].numpy(), g["cam2"]) and e3.camera_for("TS6")[2] is False
    assert np.allclose(e3.image_coordinates(torch.from_numpy(g["x2d"]), 2048, 2048).numpy(), g["target_pix"], rtol=1e-6)
    P, G, V = torch.from_numpy(pred), torch.from_numpy(gt), torch.from_numpy(valid) > 0.5
    assert np.allclose(e3.mpjpe_diffusion_3dhp(P, G, V).numpy(), g["e_pbest"], rtol=1e-6)
    assert np.allclose(e3.mpjpe_diffusion_3dhp(P, G, V, mean_pos=True).numpy(), g["e_pagg"], rtol=1e-6)
