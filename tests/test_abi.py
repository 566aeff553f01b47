"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports every symbol that
include/d3dp_hip.h declares; the Python mirror keeps the reference's state_dict contract; and the product path
fails loudly without a GPU (no CPU fallback)."""
import os
import re

import numpy as np
import pytest
import torch
from types import SimpleNamespace

import __graft_entry__ as entry
from d3dp_amd import D3DP, _lib
from d3dp_amd.weights import H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT, make_state_dict, param_shapes
from oracle import d3dp_oracle as orc

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        entry.build()
    return _lib.load()


def header_functions():
    src = open(os.path.join(REPO, "include", "d3dp_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(d3dp_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree(lib):
    declared = header_functions()
    assert declared, "no functions parsed from include/d3dp_hip.h"
    assert sorted(_lib.PROTOTYPES) == declared


def test_every_declared_symbol_is_exported(lib):
    for name in header_functions():
        assert hasattr(lib, name), name
    assert lib.d3dp_abi_version() == _lib.ABI_VERSION
    names = [lib.d3dp_profile_class_name(i).decode() for i in range(_lib.PROFILE_CLASSES)]
    assert names[0] == "gemm_qkv" and len(set(names)) == _lib.PROFILE_CLASSES


def test_exports_are_exactly_the_header(lib):
    """the library's dynamic symbols are the functions include/d3dp_hip.h declares (the C ABI plus its
    documented "test hooks" section) and nothing else -- no C++-mangled internals (`-fvisibility=hidden`, csrc/exports.map).
    hipcc's own per-translation-unit markers (`__hip_cuid_*`) are the only other names."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(l.split()[-1] for l in out.splitlines() if l.strip() and not l.split()[-1].startswith("__hip_cuid_"))
    assert exported == header_functions()
    assert not any(n.startswith("_Z") for n in exported)
    src = open(os.path.join(REPO, "include", "d3dp_hip.h")).read()
    hooks = src[src.index("---- test hooks"):]
    assert sorted(set(re.findall(r"\b(d3dp_[a-z0-9_]+)\s*\(", re.sub(r"/\*.*?\*/", "", hooks, flags=re.S)))) == \
        ["d3dp_debug_train_linear", "d3dp_debug_x2_variants"]


def make_model(frames=243, cs=512, dep=8, H=20, K=10, is_train=False):
    args = SimpleNamespace(number_of_frames=frames, test_time_augmentation=True, timestep=1000, scale=1.0, cs=cs, dep=dep)
    return D3DP(args, H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT, is_train=is_train, num_proposals=H, sampling_timesteps=K)


def test_state_dict_contract_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "g0_state_dict_contract.npz"))
    sd = make_model().state_dict()
    assert list(sd.keys()) == [str(n) for n in g["names"]]
    assert [",".join(map(str, v.shape)) for v in sd.values()] == [str(s) for s in g["shapes"]]
    assert [str(v.dtype) for v in sd.values()] == [str(s) for s in g["dtypes"]]
    m = make_model()
    assert sum(p.numel() for p in m.parameters()) == int(g["n_params"])
    # the seed-generated weights cover exactly the pose_estimator parameters
    assert ["pose_estimator." + k for k in param_shapes(512, 8, 243)] == [k for k in sd if k.startswith("pose_estimator.")]


def test_schedule_buffers_and_time_pairs(golden_dir):
    g = np.load(os.path.join(golden_dir, "g1_schedule.npz"))
    for K in (1, 5, 10):
        m = make_model(frames=9, cs=64, dep=1, H=1, K=K)
        for k in ("betas", "alphas_cumprod", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
                  "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod"):
            assert np.array_equal(getattr(m, k).numpy(), g[k]), k
        assert np.array_equal(np.array(m.time_pairs(), dtype=np.int64), g[f"pairs_K{K}"])


def test_flip_permutation_matches_reference_indexing():
    m = make_model(frames=9, cs=64, dep=1, H=1, K=1)
    perm = m._perm(torch.device("cpu")).tolist()
    x = torch.arange(17.0).reshape(1, 17, 1).repeat(1, 1, 3)
    assert torch.equal(orc.flip_pose(x, H36M_JOINTS_LEFT, H36M_JOINTS_RIGHT)[0, :, 1], x[0, perm, 1])


def test_loads_module_prefixed_checkpoint_via_dataparallel():
    m = make_model(frames=9, cs=64, dep=2, H=1, K=1)
    sd = {"module." + k: v for k, v in make_state_dict(3, 64, 2, 9).items()}
    missing, unexpected = torch.nn.DataParallel(m).load_state_dict(sd, strict=False)
    assert not unexpected and all("pose_estimator" not in k for k in missing)
    assert torch.equal(m.pose_estimator.head[1].weight.detach(), sd["module.pose_estimator.head.1.weight"])


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_fails_loudly_without_gpu(lib):
    import ctypes as C
    cfg = _lib.Cfg(27, 17, 512, 8, 8, 1024, 1e-6, 1e-5, _lib.MODE_EXACT, 0)
    h = C.c_void_p()
    rc = lib.d3dp_create(C.byref(cfg), C.byref(h))
    assert rc == -3 and b"no CPU fallback" in lib.d3dp_last_error()
    m = make_model(frames=9, cs=64, dep=1, H=1, K=1).eval()
    with pytest.raises(_lib.D3DPHipError):
        m(torch.zeros(1, 9, 17, 2), None, input_2d_flip=torch.zeros(1, 9, 17, 2))
    with pytest.raises(_lib.D3DPHipError):
        m.pose_estimator(torch.zeros(1, 9, 17, 2), torch.zeros(1, 1, 9, 17, 3), torch.zeros(1, dtype=torch.long))


def test_create_validates_widths_and_joints_before_it_looks_for_a_device(lib):
    """d3dp_create checks the shape first (no GPU needed to see the verdict): the instantiated widths in every mode, any other
    width the reference's 8 heads divide (head dim % 4 == 0, <= 1024 channels) in EXACT and TRAIN mode, up to 256 joints -- and a
    refusal names its reason (D3DP_ENOTSUP = -2); a shape the library takes gets as far as the device check (-3 here)."""
    import ctypes as C

    def create(frames, joints, cs, heads, hidden, mode):
        cfg = _lib.Cfg(frames, joints, cs, 8, heads, hidden, 1e-6, 1e-5, mode, 0)   # (frames, joints, channels, depth, heads, hidden, ...)
        h = C.c_void_p()
        return lib.d3dp_create(C.byref(cfg), C.byref(h)), lib.d3dp_last_error().decode()

    for mode in (_lib.MODE_EXACT, _lib.MODE_FAST, _lib.MODE_TRAIN):
        for cs in (64, 128, 256, 512):
            assert create(27, 17, cs, 8, 2 * cs, mode)[0] == -3
        assert create(27, 40, 512, 8, 1024, mode)[0] == -3              # more than 32 joints
        assert create(27, 256, 512, 8, 1024, mode)[0] == -3
    for cs in (32, 96, 224, 384, 768, 1024):                            # other widths: EXACT and TRAIN (fp32 implementation), not FAST
        assert create(27, 17, cs, 8, 2 * cs, _lib.MODE_EXACT)[0] == -3
        assert create(27, 17, cs, 8, 2 * cs, _lib.MODE_TRAIN)[0] == -3
        rc, msg = create(27, 17, cs, 8, 2 * cs, _lib.MODE_FAST)
        assert rc == -2 and "FAST contexts exist" in msg, msg
    rc, msg = create(243, 17, 1024, 8, 2048, _lib.MODE_TRAIN)           # heads of 128: the fp32 attention backward holds <= 153 tokens
    assert rc == -2 and "holds a whole sequence in LDS" in msg, msg
    assert create(243, 17, 1024, 8, 2048, _lib.MODE_EXACT)[0] == -3
    for args, needle in (((27, 17, 200, 8, 400, _lib.MODE_EXACT), "head dim a multiple of 4"),
                         ((27, 17, 2048, 8, 4096, _lib.MODE_EXACT), "channels <= 1024"),
                         ((27, 257, 512, 8, 1024, _lib.MODE_EXACT), "joints=257 not in [1,256]"),
                         ((1025, 17, 512, 8, 1024, _lib.MODE_EXACT), "frames=1025")):
        rc, msg = create(*args)
        assert rc == -2 and needle in msg, msg
    assert create(27, 17, 512, 7, 1024, _lib.MODE_EXACT)[0] == -1       # heads must divide channels: D3DP_EINVAL


def test_product_code_never_imports_the_oracle():
    for root, _, files in os.walk(os.path.join(REPO, "d3dp_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f


def test_header_is_plain_c_and_links():
    """include/d3dp_hip.h is the drop-in boundary: it must compile as C99 (no C++/torch types) and a C translation unit
    that references every declared function must link against libd3dp_hip.so (no compute call is made)."""
    import re
    import shutil
    import subprocess
    import tempfile
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = os.path.join(root, "include", "d3dp_hip.h")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", hdr], check=True)
    names = sorted(set(re.findall(r"^D3DP_API (?:int|const char\*)\s+(d3dp_[a-z0-9_]+)\s*\(", open(hdr).read(), flags=re.M)))
    assert len(names) >= 30 and set(names) == set(_lib.PROTOTYPES), set(names) ^ set(_lib.PROTOTYPES)
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "link.c")
        with open(src, "w") as f:
            f.write('#include "d3dp_hip.h"\n#include <stdio.h>\nint main(void) {\n  void* p[] = {\n')
            f.write("".join(f"    (void*){n},\n" for n in names))
            f.write('  };\n  printf("%d %d\\n", (int)(sizeof p / sizeof p[0]), d3dp_abi_version());\n  return 0;\n}\n')
        exe = os.path.join(td, "link")
        libdir = os.path.dirname(_lib.LIB_PATH)
        subprocess.run(["gcc", "-std=c99", "-I", os.path.join(root, "include"), src, "-o", exe, "-L", libdir, "-ld3dp_hip",
                        "-Wl,-rpath," + libdir, "-Wl,--allow-shlib-undefined"], check=True)
        out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()
        assert int(out[0]) == len(names) and int(out[1]) == _lib.ABI_VERSION


@pytest.mark.parametrize("flags", [[], ["-DD3DP_ATTN_OVERLAP=1"], ["-DD3DP_ATTN_KPIPE=1"], ["-DD3DP_ATTN_MIXLO=1", "-DD3DP_ATTN_W16=1"]],
                         ids=["default", "overlapped-softmax", "k-fragment-pipeline", "mixlo+w16-kernel-compiles"])
def test_kernels_with_untracked_loads_do_not_spill(tmp_path, flags):
    """attn_temporal_x2_kernel prefetches its queries with loads the compiler does not track (inline asm; see
    attention.hip gload16_untracked): a register spill placed right after such a load would save the register before the
    data has arrived.  The kernel is sized to fit its 256 registers exactly -- hold the build to that, for the default
    form and for the two measured alternatives kept behind build switches."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src = os.path.join(os.path.dirname(_lib.LIB_PATH), "..", "csrc", "attention.hip")
    out = str(tmp_path / "attention.s")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", *flags, "-o", out, src],
                   check=True, capture_output=True, timeout=600)
    text = open(out).read()
    sizes = re.findall(r"\.set (\S*attn_temporal_x2_kernel\S*)\.private_seg_size, (\d+)", text)
    assert len(sizes) >= 8, sizes
    assert all(int(v) == 0 for _, v in sizes), [s for s in sizes if int(s[1])]
    # the asm returns its destination registers before the data lands, so NOTHING -- no compiler copy, VALU
    # operation or reuse -- may touch them between the load and the counted s_waitcnt that covers it.  Scan the ISA of every
    # instantiation: from each untracked load forward to the first wait on vmcnt, following the problem loop's back edge.
    n_loads = sum(_scan_untracked_loads(body, name) for name, body in _kernel_bodies(text, "attn_temporal_x2_kernel"))
    assert n_loads >= 8 * 4, n_loads


def _kernel_bodies(text, needle):
    """(symbol, instruction lines) of every kernel whose mangled name contains `needle`."""
    for m in re.finditer(r"^(_Z\S*%s\S*):\s*(?:;.*)?$" % needle, text, flags=re.M):
        end = text.index("s_endpgm", m.end())
        yield m.group(1), text[m.end():end].splitlines()


def _vregs(line):
    code = line.split(";")[0]
    regs = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", code):
        regs.update(range(int(a), int(b) + 1))
    regs.update(int(a) for a in re.findall(r"\bv(\d+)\b", code))
    return regs


def _scan_untracked_loads(lines, name):
    labels = {m.group(1): i for i, l in enumerate(lines) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
    in_asm, count = False, 0
    for i, l in enumerate(lines):
        if "#ASMSTART" in l:
            in_asm = True
        elif "#ASMEND" in l:
            in_asm = False
        m = re.match(r"\s+global_load_dwordx4 v\[(\d+):(\d+)\], v\[\d+:\d+\], off\s*$", l.split(";")[0]) if in_asm else None
        if not m:
            continue
        count += 1
        dst = set(range(int(m.group(1)), int(m.group(2)) + 1))

        def first_touch(start, stop):
            """index of the first vmcnt wait in [start, stop), asserting nothing touches dst before it; None if no wait"""
            for j in range(start, stop):
                code = lines[j].split(";")[0].strip()
                if not code or code.endswith(":") or code.startswith("."):
                    continue
                if code.startswith("s_waitcnt") and "vmcnt" in code:
                    return j
                assert not (_vregs(lines[j]) & dst), f"{name}: `{code}` touches v{sorted(dst)} of an untracked load before its wait"
            return None

        # the wait may sit behind the problem loop's back edge: scan to the first backward branch whose target precedes the
        # load, then from that target (the loop header) up to the load
        br = re.compile(r"\s+s_c?branch\S*\s+(\.LBB\d+_\d+)")
        back = next((j for j in range(i + 1, len(lines))
                     if (b := br.match(lines[j])) and labels.get(b.group(1), 1 << 30) <= i), None)
        if first_touch(i + 1, back if back is not None else len(lines)) is None:
            assert back is not None, f"{name}: no vmcnt wait after the untracked load at line {i}"
            assert first_touch(labels[br.match(lines[back]).group(1)], i) is not None, \
                f"{name}: no vmcnt wait covers the untracked load at line {i}"
    return count
