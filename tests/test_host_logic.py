"""CPU tests of host-side logic that needs no GPU: the reference checkpoint layout, the entrypoints' flags, screen
normalisation, clip bookkeeping.  (The kernels behind them are covered by tests/test_hip_caller.py, -m gpu.)"""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from d3dp_amd import D3DP, D3DP3DHP, _lib
from d3dp_amd import cli, serve, trainer
from d3dp_amd.clips import clip_count
from d3dp_amd.data import ChunkLineage
from d3dp_amd.weights import H36M_JOINTS_LEFT as KL, H36M_JOINTS_RIGHT as KR, make_state_dict


def tiny_args(frames=9, cs=64, dep=2):
    return SimpleNamespace(number_of_frames=frames, test_time_augmentation=True, timestep=1000, scale=1.0, cs=cs, dep=dep)


def test_checkpoint_layout_round_trip(tmp_path):
    """main.py:543-552 / 252-258, 335-343: {'epoch','lr','random_state','optimizer','model_pos'} with DataParallel-style
    `module.` keys; written by trainer.checkpoint_dict, read back by trainer.load_checkpoint (and by a plain
    load_state_dict after stripping the prefix, which is what the reference's DataParallel wrapper does)."""
    m = D3DP(tiny_args(), KL, KR, is_train=True)
    m.load_state_dict(make_state_dict(3, 64, 2, 9), strict=False)
    opt = torch.optim.AdamW(m.parameters(), lr=6e-5, weight_decay=0.1)
    for p in m.parameters():
        p.grad = torch.full_like(p, 1e-3)
    opt.step()
    lin = ChunkLineage([40, 9], 4, 9, shuffle=True, augment=True)
    lin.next_pairs()                                       # advance the RandomState like one epoch does
    ck = trainer.checkpoint_dict(7, 5.5e-5, lin, opt, m)
    assert set(ck) == {"epoch", "lr", "random_state", "optimizer", "model_pos"}
    assert all(k.startswith("module.") for k in ck["model_pos"]) and len(ck["model_pos"]) == len(m.state_dict())
    assert ck["model_pos"]["module.betas"].dtype == torch.float64
    path = os.path.join(tmp_path, "epoch_7.bin")
    torch.save(ck, path)
    m2 = D3DP(tiny_args(), KL, KR, is_train=True)
    opt2 = torch.optim.AdamW(m2.parameters(), lr=1.0, weight_decay=0.1)
    lin2 = ChunkLineage([40, 9], 4, 9, shuffle=True, augment=True)
    got = trainer.load_checkpoint(path, m2, opt2, lin2)
    assert got["epoch"] == 7 and got["lr"] == 5.5e-5
    for (k, a), b in zip(m.state_dict().items(), m2.state_dict().values()):
        assert torch.equal(a, b), k
    assert opt2.param_groups[0]["lr"] == 6e-5 and float(opt2.state[next(iter(m2.parameters()))]["step"]) == 1.0
    # the restored RandomState continues the permutation stream exactly
    assert np.array_equal(lin.next_pairs()[1], lin2.next_pairs()[1])
    # an evaluation model (fewer buffers needed, different H/K) loads the same file non-strictly, as main.py:257 does
    ev = D3DP(tiny_args(), KL, KR, is_train=False, num_proposals=3, sampling_timesteps=2)
    ev.load_state_dict({k[len("module."):]: v for k, v in got["model_pos"].items()}, strict=False)


def test_entrypoint_flags_follow_the_reference():
    a = cli.parse_args(["--synthetic", "-c", "ck", "--evaluate", "best_epoch.bin", "-num_proposals", "20",
                        "-sampling_timesteps", "10", "-b", "4", "-f", "243", "-cs", "512", "-dep", "8", "--p2"])
    assert (a.evaluate, a.num_proposals, a.sampling_timesteps, a.batch_size, a.number_of_frames, a.p2) == \
           ("best_epoch.bin", 20, 10, 4, 243, True) and a.test_time_augmentation is True
    t = cli.parse_args(["--synthetic", "-e", "3", "-lr", "0.0001", "-lrd", "0.99", "-s", "243", "-b", "972", "-cf", "5",
                        "-r", "epoch_2.bin", "--no-eval", "-no-da"])
    assert (t.epochs, t.learning_rate, t.lr_decay, t.stride, t.batch_size, t.checkpoint_frequency, t.resume, t.no_eval,
            t.data_augmentation, t.evaluate) == (3, 1e-4, 0.99, 243, 972, 5, "epoch_2.bin", True, False, "")
    d = cli.parse_args([])                                  # reference defaults (arguments.py:27-47)
    assert (d.epochs, d.learning_rate, d.lr_decay, d.checkpoint_frequency, d.stride) == (400, 6e-5, 0.993, 20, 243)


def test_entrypoints_refuse_to_run_without_a_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(SystemExit):
        cli.main(["--synthetic", "--evaluate", "x.bin"])
    with pytest.raises(SystemExit):
        cli.main_3dhp(["--synthetic", "--evaluate", "x.bin"])
    m = D3DP(tiny_args(), KL, KR, is_train=False)
    with pytest.raises(_lib.D3DPHipError):
        serve.predict_video(m, np.zeros((5, 17, 2), np.float32), 100, 100)


def test_3dhp_model_is_the_same_network_in_millimetres():
    a, b = D3DP(tiny_args(), KL, KR, is_train=False), D3DP3DHP(tiny_args(), KL, KR, is_train=False)
    assert list(a.state_dict()) == list(b.state_dict()) and D3DP3DHP.MM == 1000.0


def test_normalisation_and_clip_counts():
    x = torch.tensor([[0.0, 0.0], [1920.0, 1080.0], [960.0, 540.0]])
    y = serve.normalize_screen_coordinates(x, 1920, 1080)                  # camera.py:7-11
    assert torch.allclose(y, torch.tensor([[-1.0, -0.5625], [1.0, 0.5625], [0.0, 0.0]]))
    assert [clip_count(n, 243) for n in (1, 242, 243, 244, 486, 487, 100000)] == [1, 1, 1, 2, 2, 3, 412]


def test_train_model_is_always_differentiable_numerics():
    """One argparse namespace builds the train and the evaluation models (main.py:228-230): --numerics selects the
    inference arithmetic only, the train model stays in 'train' numerics."""
    from types import SimpleNamespace
    from d3dp_amd import D3DP
    args = SimpleNamespace(number_of_frames=9, test_time_augmentation=True, timestep=1000, scale=1.0, cs=64, dep=1,
                           numerics="fast")
    assert D3DP(args, [4, 5, 6], [1, 2, 3], is_train=True).pose_estimator.numerics == "train"
    assert D3DP(args, [4, 5, 6], [1, 2, 3], is_train=False).pose_estimator.numerics == "fast"


def test_bench_relaunches_itself_under_torchrun(monkeypatch):
    """`python bench.py --gpus N` outside a torchrun job becomes `python -m torch.distributed.run --nproc-per-node N ...
    bench.py <same args>` on 127.0.0.1 with a free port; inside a job whose WORLD_SIZE differs it refuses to run."""
    import importlib.util
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(repo, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    def fake_execv(exe, argv):
        seen["exe"], seen["argv"] = exe, argv
        raise SystemExit(0)

    monkeypatch.setattr(os, "execv", fake_execv)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "2", "--warmup", "1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit):
        bench.main()
    a = seen["argv"]
    assert a[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=8" in a and "--nnodes=1" in a
    assert a[a.index("--master-addr") + 1] == "127.0.0.1" and int(a[a.index("--master-port") + 1]) > 0
    assert a[-6:] == ["--gpus", "8", "--steps", "2", "--warmup", "1"] and a[-7].endswith("bench.py")


def test_bench_reports_counter_traffic_only_for_the_build_it_was_measured_on(monkeypatch):
    """roofline.traffic comes from separate rocprofv3 --pmc passes stored in profiles/ next to the sha256 of the library
    they were measured on ('refuse to print it when it differs')."""
    import json
    import bench
    tj = json.load(open(os.path.join(os.path.dirname(bench.__file__), "profiles", "r02_gemm_traffic.json")))
    sha = tj["exact"]["gemm_qkv"]["lib_sha256"]
    rows = tj["exact"]["gemm_qkv"]["mean_rows_per_launch"]
    roof = {"kernel": "gemm_qkv", "traffic": None, "mean_rows_per_launch": rows * 1.03}
    monkeypatch.setattr(bench, "lib_sha256", lambda: sha)
    bench.attach_traffic(roof, "exact", 0)
    assert roof["traffic"] == tj["exact"]["gemm_qkv"]["hbm_bytes_per_launch"] > tj["exact"]["gemm_qkv"]["algorithmic_bytes_per_launch"]
    # ... and only at the pass size the counters were taken at (the r03 numbers came from --batch 4)
    roof = {"kernel": "gemm_qkv", "traffic": None, "mean_rows_per_launch": rows * 1.14}
    bench.attach_traffic(roof, "exact", 0)
    assert roof["traffic"] is None and "rows per launch" in roof["traffic_note"]
    roof = {"kernel": "gemm_qkv", "traffic": None, "mean_rows_per_launch": rows}
    monkeypatch.setattr(bench, "lib_sha256", lambda: "0" * 64)
    bench.attach_traffic(roof, "exact", 0)
    assert roof["traffic"] is None and "different build" in roof["traffic_note"]


def test_bench_roofline_is_algorithmic_flop_over_the_guides_dense_peak():
    """`roofline.frac` = ALGORITHMIC FLOP / time / the guide's dense fp16 peak (2.5 PFLOP/s); the three
    MFMA products per algorithmic product of exact mode are reported beside it as matrix-pipe work, never as `frac`.  And
    the newest traffic file whose library hash matches wins; the CPU baseline carries the un-extrapolated configs[1] run."""
    import json
    import bench
    B, H, K = 16, 20, 10
    prof = {"gemm_qkv": (3360, 1610.618), "gemm_proj": (3360, 663.7), "gemm_fc1": (3360, 1290.1), "gemm_fc2": (3360, 1123.6)}
    r = bench.roofline_from_profile(prof, "exact", B, H, K)
    flops = 2 * 3 * 512 * 512 * (2 * B * H * 243 * 17 * K) * 16
    assert r["kernel"] == "gemm_qkv" and r["peak"] == 2500.0 and r["bound"] == "mfma"
    assert abs(r["achieved"] - flops / 1.610618 / 1e12) < 1e-6 * r["achieved"]
    assert abs(r["frac"] - r["achieved"] / 2500.0) < 1e-12 and r["frac"] == r["frac_algorithmic_of_dense_peak"] < 0.2
    assert r["mfma_passes_per_product"] == 3 and abs(r["matrix_pipe_work_tflops"] - 3 * r["achieved"]) < 1e-9
    rf = bench.roofline_from_profile(prof, "fast", B, H, K)
    assert rf["mfma_passes_per_product"] == 1 and rf["matrix_pipe_work_tflops"] == rf["achieved"]
    full = bench.full_config_cpu_run()
    assert full and 300 < full["gflops"] < 400 and abs(full["k10_units_per_s"] * 2 - full["hypothesis_clips_per_s_K5_units"]) < 1e-12
    prof_dir = os.path.join(os.path.dirname(bench.__file__), "profiles")
    t3 = json.load(open(os.path.join(prof_dir, "r03_gemm_traffic.json")))["exact"]["gemm_qkv"]
    roof = {"kernel": "gemm_qkv", "traffic": None, "mean_rows_per_launch": t3["mean_rows_per_launch"]}
    assert abs(r["mean_rows_per_launch"] - 2 * B * H * 243 * 17 * K * 16 / 3360) < 1e-6
    import pytest as _pt
    mp = _pt.MonkeyPatch()
    try:
        mp.setattr(bench, "lib_sha256", lambda: t3["lib_sha256"])
        bench.attach_traffic(roof, "exact", 0)
    finally:
        mp.undo()
    assert roof["traffic"] == t3["hbm_bytes_per_launch"] and 1.3 < roof["traffic"] / t3["algorithmic_bytes_per_launch"] < 1.7


def test_bench_telemetry_reads_the_hwmon_of_the_hip_device(tmp_path, monkeypatch):
    """bench.py samples sclk and package power of the timed steps from the amdgpu hwmon directory of the card at the
    HIP device's PCI address (no subprocess, nothing on the GPU's queues).  Here on a fake sysfs tree: two cards with sensors, the second
    is the device; units (Hz, microwatts) and the mean / min / max arithmetic."""
    import glob as _glob
    import bench
    cards = {}
    for i, pci in enumerate(("0000:05:00.0", "0000:75:00.0")):
        dev = tmp_path / "devices" / pci
        hw = dev / "hwmon" / f"hwmon{i}"
        hw.mkdir(parents=True)
        (hw / "freq1_input").write_text(str((1500 + 300 * i) * 1000000))
        (hw / "power1_input").write_text(str((900 + 499 * i) * 1000000))
        (hw / "power1_cap").write_text("1400000000")
        link = tmp_path / "drm" / f"card{i}"
        link.mkdir(parents=True)
        (link / "device").symlink_to(dev)
        cards[i] = str(link / "device" / "hwmon" / f"hwmon{i}")
    monkeypatch.setattr(_glob, "glob", lambda pat: sorted(cards.values()))
    props = SimpleNamespace(pci_domain_id=0, pci_bus_id=0x75, pci_device_id=0)
    monkeypatch.setattr(torch.cuda, "get_device_properties", lambda i: props)
    t = bench.GpuTelemetry(0, period=0.005)
    assert t.dir == cards[1] and "0000:75:00.0" in t.source and t.cap_w == 1400.0
    t.start()
    import time
    time.sleep(0.05)
    t.stop()
    r = t.report()
    assert r["samples"] >= 3 and r["clock_mhz_mean"] == 1800.0 and r["clock_mhz_min"] == r["clock_mhz_max"] == 1800.0
    assert r["power_w_mean"] == 1399.0 and r["power_cap_w"] == 1400.0
    # no sensors readable: the keys are None, nothing raises
    monkeypatch.setattr(_glob, "glob", lambda pat: [])
    r0 = bench.GpuTelemetry(0).report()
    assert r0["clock_mhz_mean"] is None and r0["power_w_mean"] is None and r0["samples"] == 0


def test_bench_training_roofline_is_algorithmic_work_over_corrected_event_time():
    """configs.c5_train_step.roofline_by_kernel: per class ALGORITHMIC FLOP (bytes) over the library's event time minus the empty
    event pair's time per launch; both attention passes of short sequences run as ONE kernel and its class carries both passes' work;
    no fraction above 1 is ever printed (a row kernel faster than HBM is reported as cache-resident)."""
    import bench
    B, T = 4, 4 * 243 * 17
    lin = 8 * 512 * 512 * 2 * T * 16
    prof = {"gemm_qkv": (10, 5.0),                                   # (the denoiser's classes are not the training step's)
            "train_linear": (128 * 3, 3 * (6.4 + 128 * 0.004)), "train_wgrad": (16 * 3, 3 * (2.7 + 16 * 0.004)),
            "train_attn_bwd_q_spatial": (8 * 3, 3 * (0.64 + 8 * 0.004)), "train_attn_bwd_kv_spatial": (8 * 3, 3 * 8 * 0.004),
            "train_attn_bwd_q_temporal": (8 * 3, 3 * (0.7 + 8 * 0.004)), "train_operand_pass": (145 * 3, 3 * (0.001 + 145 * 0.004)),
            "event_pair_overhead": (24, 24 * 0.004)}
    r = bench.train_roofline_by_kernel(prof, B, 3)
    assert "gemm_qkv" not in r and "event_pair_overhead" not in r and "train_attn_bwd_kv_spatial" not in r
    assert r["train_linear"]["launches"] == 128 and r["train_linear"]["ms_per_step"] == pytest.approx(6.4, abs=1e-3)
    assert r["train_linear"]["achieved"] == pytest.approx(2 * lin / 6.4e-3 / 1e12, rel=1e-3)
    assert r["train_wgrad"]["frac"] == pytest.approx(lin / 2.7e-3 / 1e12 / 2500.0, abs=2e-4)
    assert r["train_attn_bwd_q_spatial"]["algorithmic_gflop_per_step"] == pytest.approx(10 * 17 * 512 * T * 8 / 1e9, rel=1e-3)
    assert r["train_attn_bwd_q_temporal"]["algorithmic_gflop_per_step"] == pytest.approx(6 * 243 * 512 * T * 8 / 1e9, rel=1e-3)
    op = r["train_operand_pass"]                                     # 14 GB in 1 us: not an HBM rate
    assert op["bound"] == "cache" and op["frac"] is None and op["peak"] is None
    assert all(v.get("frac") is None or v["frac"] <= 1.0 for v in r.values())


def test_rational_erf_of_the_exact_fc1_epilogue_is_fp32_class():
    """common.h gelu_erf_rational (one rational x P(x^2)/Q(x^2) instead of libm's erff), restated in numpy with the same
    coefficients and fma order: against fp64 its GELU is as accurate as torch's own fp32 GELU on N(0,1) inputs."""
    import re
    import numpy as np
    from scipy.special import erf
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "d3dp_amd", "csrc", "common.h")).read()
    body = src[src.index("float gelu_erf_rational(float x)"):src.index("typedef float f32x2")]
    coef = [np.float32(c) for c in re.findall(r"(-?\d\.\d+e-\d+)f", body)]
    assert len(coef) == 12
    A, B = coef[:7], coef[7:]

    def fma(x, y, z):
        return (x.astype(np.float64) * y.astype(np.float64) + z.astype(np.float64)).astype(np.float32)

    def gelu(x):
        z = np.clip(x * np.float32(0.70710678118654752440), np.float32(-4), np.float32(4))
        z2 = z * z
        p = np.full_like(x, A[0])
        for c in A[1:]:
            p = fma(p, z2, np.full_like(x, c))
        p = p * z
        q = np.full_like(x, B[0])
        for c in B[1:]:
            q = fma(q, z2, np.full_like(x, c))
        e = (p / q).astype(np.float32)
        hx = np.float32(0.5) * x
        return fma(hx, e, hx)

    x = np.random.default_rng(0).standard_normal(400000).astype(np.float32)
    truth = 0.5 * x.astype(np.float64) * (1 + erf(x.astype(np.float64) / np.sqrt(2)))
    err = np.abs(gelu(x).astype(np.float64) - truth)
    terr = np.abs(torch.nn.functional.gelu(torch.from_numpy(x)).numpy().astype(np.float64) - truth)
    assert err.mean() < 1.5 * terr.mean() and err.max() < 1.5e-6, (err.mean(), terr.mean(), err.max())
    big = np.array([-30.0, -6.0, -4.0, 4.0, 6.0, 30.0], dtype=np.float32)      # beyond the clamp: erf = -1 / +1
    assert np.allclose(gelu(big), [0, 0, -1.3e-4, 4.0, 6.0, 30.0], atol=2e-4)


def test_h2i_lds_image_of_the_exact_linear_is_conflict_free_and_complete():
    """The LDS image of the EXACT Linear's operand slabs (gemm_x2.hip): rows of 128 bytes = 8 slots of 16 B (q = 4 plane +
    k-group), slot q of row r at physical slot q ^ ((r >> 1) & 7).  (1) Every LDS-DMA piece (64 lanes x 16 B, lane ->
    row lane >> 3, physical slot lane & 7) covers each (row, logical slot) of its 8 rows exactly once, from ONE 128-byte line
    per row.  (2) A fragment read (lane: row fi = lane & 15, k-group fg = lane >> 4, one plane) is bank-conflict-free: the four
    16-lane groups ds_read_b128 is serviced in (MI355X_MICROARCH.md, LDS table) each hit 16 different 16-byte slots of the
    256-byte bank row.  (On the GPU: SQ_LDS_BANK_CONFLICT = 0, profiles/r03_gemm_pmc.md.)"""
    swz = lambda row, q: q ^ ((row >> 1) & 7)
    for piece in range(32):
        seen = set()
        for lane in range(64):
            row, phys = piece * 8 + (lane >> 3), lane & 7
            q = swz(row, phys)                               # the logical slot this lane fetches (involution)
            assert swz(row, q) == phys and 0 <= q < 8
            seen.add((row, q))
        assert len(seen) == 64
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    groups += [[l + 32 for l in g] for g in groups]
    for base_row in (0, 16, 64, 192):                        # row blocks of a wave advance in multiples of 16
        for plane in (0, 1):
            for g in groups:
                slots = set()
                for lane in g:
                    row, q = base_row + (lane & 15), plane * 4 + (lane >> 4)
                    addr = row * 128 + swz(row, q) * 16
                    slots.add((addr // 16) % 16)
                assert len(slots) == 16, (base_row, plane, g)
    # the lo plane's slot is the hi plane's ^ 4 = byte offset ^ 64, as the kernel computes it
    for row in range(32):
        for kg in range(4):
            assert (swz(row, kg) * 16) ^ 64 == swz(row, 4 + kg) * 16


def test_h2i_layout_and_w_row_permutation_are_bijections():
    """h2i (common.h): column c of a K-column row -> hi at ((c >> 5) << 6) | (c & 31), lo 32 further: a bijection onto the 2 K
    fp16 of the row, 128-byte blocks = [hi of 32 columns | lo of the same].  W-row permutation of the Linear (gemm_x2.hip
    colperm): LDS row q of a 64-column strip carries output column (q & 15) * 4 + (q >> 4), so the four MFMA tiles of a lane
    (same operand row i, tile ni = 0..3) are four CONSECUTIVE columns -> one 16-byte store per row."""
    for K in (64, 512, 1024):
        offs = [((c >> 5) << 6) | (c & 31) for c in range(K)]
        both = offs + [o + 32 for o in offs]
        assert sorted(both) == list(range(2 * K))
        for c in range(K):                                   # one k-step (32 columns), both values: one 128-byte block
            assert offs[c] * 2 // 128 == c // 32 == (offs[c] + 32) * 2 // 128
    colperm = lambda q: (q & 15) * 4 + (q >> 4)
    assert sorted(colperm(q) for q in range(64)) == list(range(64))
    for i in range(16):
        assert [colperm(ni * 16 + i) for ni in range(4)] == [4 * i, 4 * i + 1, 4 * i + 2, 4 * i + 3]


class _FakeLib:
    """Stands in for libd3dp_hip.so in lifecycle tests: hands out context handles, records every destroy."""

    def __init__(self):
        self.created, self.destroyed = [], []

    def d3dp_create(self, cfg, out):
        import ctypes
        h = 0x1000 + 0x100 * len(self.created)
        ctypes.cast(out, ctypes.POINTER(ctypes.c_void_p))[0] = h
        self.created.append(h)
        return 0

    def d3dp_destroy(self, h):
        self.destroyed.append(h.value if hasattr(h, "value") else h)
        return 0


@pytest.fixture
def fake_lib(monkeypatch):
    """MixSTE2._context on a box without a GPU: fake library, no device switch, no weight push."""
    import contextlib
    from d3dp_amd import model as M
    lib = _FakeLib()
    monkeypatch.setattr(_lib, "load", lambda: lib)
    monkeypatch.setattr(torch.cuda, "device", lambda d: contextlib.nullcontext())
    monkeypatch.setattr(M.MixSTE2, "_push_weights", lambda self, st, device, borrowed=False: None)
    return lib


def _replicate_like_torch(network):
    """torch/nn/parallel/replicate.py for ONE replica, without devices: every module is shallow-copied by
    `_replicate_for_data_parallel` (which empties `_parameters`), children are re-wired, and each parameter is set on the
    replica as a plain NON-parameter attribute holding a non-leaf copy (Broadcast's output on a real box)."""
    modules = list(network.modules())
    idx = {m: i for i, m in enumerate(modules)}
    copies = [m._replicate_for_data_parallel() for m in modules]
    for m, r in zip(modules, copies):
        for key, child in m._modules.items():
            setattr(r, key, None if child is None else copies[idx[child]])
        for key, param in m._parameters.items():
            setattr(r, key, None if param is None else param * 1.0)       # non-leaf copy with an autograd edge to the source
        for key, buf in m._buffers.items():
            setattr(r, key, buf)
    return copies[0]


def test_data_parallel_replicas_never_free_the_parents_context(fake_lib):
    """VERDICT r3 weak 7 (reference caller: main.py:242-248 wraps every model in nn.DataParallel, :698 calls it).  A replica
    is a shallow copy made on every forward; it must share the per-device contexts, create the one of a new device once,
    and never pass any handle to d3dp_destroy -- only the module that owns the states does, once per device."""
    import gc
    net = D3DP(tiny_args(), KL, KR, is_train=False).pose_estimator
    d0, d1 = torch.device("cuda", 0), torch.device("cuda", 1)
    h0 = net._context(d0).value
    assert fake_lib.created == [h0] and net._ctx.value == h0
    for _ in range(3):                                     # three forwards of a 2-device DataParallel
        r0, r1 = _replicate_like_torch(net), _replicate_like_torch(net)
        assert r0._states is net._states and not r0._owns_states and net._owns_states
        assert r0._context(d0).value == h0                 # device 0: the parent's context, reused
        h1 = r1._context(d1).value                         # device 1: its own, created once
        assert h1 != h0 and net._states[d1].ctx.value == h1
        r1._drop_ctx()                                     # even an explicit drop on a replica frees nothing
        del r0, r1
        gc.collect()
    assert fake_lib.created == [h0, h1] and fake_lib.destroyed == []
    assert net._context(d0).value == h0                    # the parent's handle is still the live one
    del net
    gc.collect()
    assert sorted(fake_lib.destroyed) == sorted([h0, h1])  # the owner frees each device's context exactly once


def test_data_parallel_replicas_see_new_weights(fake_lib, monkeypatch):
    """a replica's `parameters()` is EMPTY, so a signature built from it never changes and the packed
    weights of a device were pushed once and never again -- main.py's flow (load the training weights into model_pos every
    epoch, evaluate through nn.DataParallel, :242-258, :450) silently evaluated with the first epoch's weights.  The
    signature now follows the SOURCE module's parameters; a replica re-packs exactly when they changed, from its own
    (broadcast) tensors, and never borrows them."""
    from d3dp_amd import model as M
    pushes = []

    def push(self, st, device, borrowed=False):
        pushes.append((borrowed, float(self.head[1].weight.detach().reshape(-1)[0]), len(self._weight_tensors())))

    monkeypatch.setattr(M.MixSTE2, "_push_weights", push)
    for is_train in (False, True):
        pushes.clear()
        m = D3DP(tiny_args(), KL, KR, is_train=is_train)
        net = m.pose_estimator
        d0 = torch.device("cuda", 0)
        n_params = len(list(net.parameters()))
        r = _replicate_like_torch(m).pose_estimator
        assert list(r.parameters()) == [] and len(r._weight_tensors()) == n_params     # what the old signature was built from
        assert not r._owns_states and r._states is net._states
        r._context(d0)
        assert len(pushes) == 1 and pushes[0][0] is False and pushes[0][2] == n_params  # never borrowed on a replica
        _replicate_like_torch(m).pose_estimator._context(d0)                           # next forward, same weights: no re-pack
        assert len(pushes) == 1
        with torch.no_grad():
            net.head[1].weight.add_(1.0)                                               # optimizer step / load_state_dict
        _replicate_like_torch(m).pose_estimator._context(d0)
        assert len(pushes) == 2 and abs(pushes[1][1] - pushes[0][1] - 1.0) < 1e-6      # ...re-packed, from the NEW values
        m.load_state_dict(m.state_dict())                                              # copy_ in place bumps the versions too
        _replicate_like_torch(m).pose_estimator._context(d0)
        assert len(pushes) == 3
        # the weight tensors of a replica keep their autograd edge to the source parameters (multi-GPU training)
        w = _replicate_like_torch(m).pose_estimator._weight_tensors()
        assert all(t.grad_fn is not None for t in w) and len(w) == n_params
        # and on the module itself they ARE its parameters, in order
        assert all(a is b for a, b in zip(net._weight_tensors(), net.parameters()))


def test_deepcopy_and_pickle_of_a_model_do_not_share_library_handles(fake_lib):
    import copy, gc, pickle
    m = D3DP(tiny_args(), KL, KR, is_train=False)
    h = m.pose_estimator._context(torch.device("cuda", 0)).value
    m2 = copy.deepcopy(m)
    m3 = pickle.loads(pickle.dumps(m))
    for c in (m2, m3):
        assert c.pose_estimator._states == {} and c.pose_estimator._owns_states and c.pose_estimator._ctx is None
        assert torch.equal(c.pose_estimator.head[1].weight, m.pose_estimator.head[1].weight)
    h2 = m2.pose_estimator._context(torch.device("cuda", 0)).value
    assert h2 != h
    del m2, m3
    gc.collect()
    assert fake_lib.destroyed == [h2]
    m.pose_estimator.set_numerics("fast")                  # switching arithmetic drops the owner's contexts
    assert fake_lib.destroyed == [h2, h] and m.pose_estimator._ctx is None


def test_droppath_masks_follow_the_rate_and_never_divide_by_zero():
    """the cached keep-rates are keyed on (device, drop_path_rate, depth) -- a changed rate takes effect on the next
    step --, a rate of 1.0 drops every sample without 0/0 (timm guards `keep_prob > 0`), and a module pickled before
    `_param_names` existed still resolves its weights."""
    import pickle
    m = D3DP(tiny_args(), KL, KR, is_train=True)
    net = m.pose_estimator.train()
    dev = torch.device("cpu")
    net.drop_path_rate = 0.0
    assert net._droppath_masks(2, dev) is None
    net.drop_path_rate = 0.5
    a = net._droppath_masks(64, dev)
    last = a[-2:]                                          # the last block: rate 0.5 -> scales 0 or 2
    assert set(last.unique().tolist()) <= {0.0, 2.0} and (last == 0).any() and torch.all(a[:2] == 1.0)
    net.drop_path_rate = 1.0                               # picked up without touching the cache by hand
    b = net._droppath_masks(64, dev)
    assert torch.isfinite(b).all() and torch.all(b[-2:] == 0.0)
    d = net.__getstate__()
    d.pop("_param_names")
    old = object.__new__(type(net))
    old.__setstate__(d)
    assert old._param_names == net._param_names and len(old._weight_tensors()) == len(list(net.parameters()))
    assert pickle.loads(pickle.dumps(net))._param_names == net._param_names


def test_reference_attributes_exist():
    """common/diffusionpose.py:74, 87-90: attributes ported code may read."""
    m = D3DP(tiny_args(), KL, KR, is_train=False)
    assert (m.objective, m.self_condition, m.box_renewal, m.use_ensemble, m.ddim_sampling_eta) == \
           ('pred_x0', False, True, True, 1.)


def _simulate_skewed_linear(M, tm, tiles_n, G, NK, D):
    """Index-level model of gemm_f16x2_skew_kernel (d3dp_amd/csrc/gemm_x2.hip), written from the kernel's formulas: per
    workgroup the loader's row pointers (rotating LDS image), the compute waves' park / shift / store sequence.  Returns,
    per (row tile, strip, row class), the list of k-steps that were accumulated into what got stored, and the store count."""
    from collections import defaultdict
    Q = G // tiles_n
    stored = defaultdict(list)                     # (tile_row, strip, cls) -> [tuple of k-steps accumulated] per store
    for L in range(G):
        strip, rg = L % tiles_n, L // tiles_n
        lo = rg * tm // Q if rg < Q else 0
        hi = (rg + 1) * tm // Q if rg < Q else 0
        n_tiles = hi - lo
        if n_tiles <= 0:
            continue
        gtot = n_tiles * NK + 3 * D
        # ---- loader: slab g holds, in LDS row slot p, rows of (class, tile j) at k-offset ks
        slabs, pa, ti, ks = [], [None] * 4, 0, 0
        for g in range(gtot):
            if ks % D == 0 and ks < 4 * D:
                rot = (ks // D + 1) & 3
                for p in range(4):
                    cls = (p + rot) & 3
                    j = ti if ks >= cls * D else ti - 1
                    pa[p] = (cls, min(max(j, 0), n_tiles - 1), j)      # (class, clamped tile, wanted tile)
            slabs.append((list(pa), ks))
            ks += 1
            if ks == NK:
                ks, ti = 0, ti + 1
        # ---- compute: acc[p] = list of (class, tile, ks) contributions
        acc = [[] for _ in range(4)]
        g = 0

        def park_and_shift(cls, tile):
            parked = acc[0]
            acc[0], acc[1], acc[2], acc[3] = acc[1], acc[2], acc[3], []
            if tile >= 0:
                stored[(lo + tile, strip, cls)].append(tuple(parked))

        def kstep():
            nonlocal g
            rows, ksg = slabs[g]
            for p in range(4):
                acc[p].append((rows[p][0], rows[p][1], rows[p][2], ksg))
            g += 1

        for ti in range(n_tiles + 1):
            for grp in range(4 if ti < n_tiles else 3):
                park_and_shift(grp, ti - 1)
                for _ in range(D):
                    kstep()
            if ti < n_tiles:
                for _ in range(NK - 4 * D):
                    kstep()
        park_and_shift(3, n_tiles - 1)
        assert g == gtot                                   # compute and loader waves count the same barriers
    return stored


@pytest.mark.parametrize("tm,tiles_n,G,NK,D", [(504, 12, 256, 16, 4), (504, 8, 256, 16, 2), (37, 12, 256, 16, 4),
                                               (274, 4, 256, 16, 1), (100, 8, 64, 8, 2), (23, 12, 30, 4, 1)])
def test_skewed_linear_schedule_covers_every_tile_exactly_once(tm, tiles_n, G, NK, D):
    """The skewed schedule of the EXACT qkv / fc1 Linears: every (row tile, strip, 16-row class) is stored exactly once, and
    what is stored has accumulated every k-step of ITS OWN tile and class exactly once, in the rotated order class D, ...,
    NK - 1, 0, ..., class D - 1 -- nothing from a neighbouring tile, nothing from the ramp or the flush."""
    stored = _simulate_skewed_linear(tm * 256, tm, tiles_n, G, NK, D)
    assert len(stored) == tm * tiles_n * 4
    for (tile, strip, cls), stores in stored.items():
        assert len(stores) == 1, (tile, strip, cls)
        contrib = stores[0]
        assert len(contrib) == NK
        want_order = [(cls * D + i) % NK for i in range(NK)]
        assert [k for (_, _, _, k) in contrib] == want_order
        # rows fetched: the class's own, from the tile it was on (clamped == wanted: no ramp / flush filler inside)
        assert all(c == cls for (c, _, _, _) in contrib)
        Q = G // tiles_n
        rg = next(r for r in range(Q) if r * tm // Q <= tile < (r + 1) * tm // Q)
        assert all(jc == jw == tile - rg * tm // Q for (_, jc, jw, _) in contrib)


def test_step_counter_summary_counts_a_kernel_once(tmp_path):
    """tools/pmc_step_summary.py merges one rocprofv3 counter CSV per kernel class; a kernel matched by the filters of two
    passes (embed_ln_kernel by "ln_kernel<") must not be added twice (round 4: it was, +0.5 % on the step's HBM bytes)."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("pmc_step_summary", os.path.join(root, "tools", "pmc_step_summary.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    head = "Kernel_Name,Counter_Name,Counter_Value\n"
    a = tmp_path / "a.csv"
    b = tmp_path / "b.csv"
    a.write_text(head + "ln_kernel<512>(float*),FETCH_SIZE,10\nln_kernel<512>(float*),FETCH_SIZE,12\nembed_ln_kernel<512>(float*),FETCH_SIZE,5\n")
    b.write_text(head + "embed_ln_kernel<512>(float*),FETCH_SIZE,5\nembed_ln_kernel<512>(float*),WRITE_SIZE,7\n")
    out = tmp_path / "out.md"
    mod.main(str(out), str(a), str(b))
    rows = {l.split("|")[1].strip().strip("`"): [c.strip() for c in l.split("|")[2:-1]] for l in out.read_text().splitlines()[2:]}
    assert rows["ln_kernel<512>"] == ["2", "22", "0"]
    assert rows["embed_ln_kernel<512>"] == ["1", "5", "7"]


def test_whole_sequence_generator_appends_the_mirrored_copy():
    """UnchunkedSequences (generators.py:174-250): one sequence per batch; with augment on, element 1 is the mirrored
    sequence (x negated, left/right swapped) and its camera has cx and the tangential term negated."""
    from d3dp_amd.data import UnchunkedSequences
    rng = np.random.default_rng(5)
    p2 = [rng.standard_normal((n, 17, 2)).astype(np.float32) for n in (5, 8)]
    p3 = [rng.standard_normal((n, 17, 3)).astype(np.float32) for n in (5, 8)]
    cams = [rng.standard_normal(9) for _ in range(2)]
    gen = UnchunkedSequences(cams, p3, p2, augment=True, kps_left=KL, kps_right=KR, joints_left=KL, joints_right=KR,
                             device="cpu")
    assert gen.augment_enabled() is False and gen.num_frames() == 13      # the constructor's augment is ignored (:197)
    for i, (cam, b3, b2) in enumerate(gen.next_epoch()):
        assert cam.shape == (1, 9) and b3.shape == (1,) + p3[i].shape and np.array_equal(b2[0].numpy(), p2[i])
    gen.set_augment(True)
    for i, (cam, b3, b2) in enumerate(gen.next_epoch()):
        for got, src in ((b2, p2[i]), (b3, p3[i])):
            want = src.copy()
            want[..., 0] *= -1
            want[:, KL + KR] = want[:, KR + KL]
            assert np.array_equal(got[0].numpy(), src) and np.array_equal(got[1].numpy(), want)
        want_cam = np.array(cams[i])
        want_cam[[2, 7]] *= -1
        assert np.array_equal(cam[0], cams[i]) and np.array_equal(cam[1], want_cam)
    for cam, b3, b2 in UnchunkedSequences(None, None, p2, device="cpu").next_epoch():
        assert cam is None and b3 is None and b2.shape[0] == 1
