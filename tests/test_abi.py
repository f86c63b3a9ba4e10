"""The C-ABI library loads and exports every symbol include/sisr_hip.h declares (no GPU needed)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _declared():
    with open(os.path.join(ROOT, "include", "sisr_hip.h")) as f:
        src = f.read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"#ifdef SISR_DIAG.*?#endif", "", src, flags=re.S)  # diagnostic entries: libsisr_hip_diag.so only
    return sorted(set(re.findall(r"\b(sisr_\w+)\s*\(", src)))


def test_header_declares_entry_points():
    names = _declared()
    assert "sisr_conv3x3_c64" in names and "sisr_wgrad3x3_c64" in names and len(names) >= 20


def test_library_exports_every_declared_symbol():
    path = os.path.join(ROOT, "super-resolution-meta-attention-networks_amd", "libsisr_hip.so")
    assert os.path.exists(path), "build with: python -c 'import __graft_entry__ as g; g.build()'"
    import torch  # noqa: F401  (binds libamdhip64 first, as the package does)
    lib = ctypes.CDLL(path)
    missing = [n for n in _declared() if not hasattr(lib, n)]
    assert not missing, missing


def test_binding_table_matches_header():
    import sisr_amd
    bound = set(sisr_amd.hip.exported_symbols())
    assert set(_declared()) == bound


def test_product_library_has_no_process_wide_switches():
    """include/sisr_hip.h promises 'no global state': kernel-variant choices are per-call arguments, and the ablation /
    stamp builds and probes live in libsisr_hip_diag.so only."""
    path = os.path.join(ROOT, "super-resolution-meta-attention-networks_amd", "libsisr_hip.so")
    import torch  # noqa: F401
    lib = ctypes.CDLL(path)
    for name in ("sisr_conv3x3_c64_set_variant", "sisr_conv3x3_c64_bf16_set_persistent", "sisr_diag_mfma_peak",
                 "sisr_diag_conv_occupancy", "sisr_diag_conv_stamp"):
        assert not hasattr(lib, name), name


def _integration_sample():
    """The Level-2 binding sample of INTEGRATION.md, verbatim."""
    with open(os.path.join(ROOT, "INTEGRATION.md")) as f:
        doc = f.read()
    blocks = re.findall(r"```python\n(.*?)```", doc, flags=re.S)
    hit = [b for b in blocks if "lib.sisr_conv3x3_c64(" in b]
    assert len(hit) == 1
    return hit[0]


def _call_arg_count(src, fn):
    """Number of top-level arguments of the first call of `fn` in Python source `src`."""
    import ast
    for node in ast.walk(ast.parse(src)):
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr == fn:
            return len(node.args)
    raise AssertionError(fn)


def test_integration_sample_passes_as_many_arguments_as_the_header_declares():
    """INTEGRATION.md once fell one argument behind include/sisr_hip.h (ca_tail): a maintainer following it would have
    passed the stream as `select`.  The documented calls must match the declared prototypes argument for argument."""
    with open(os.path.join(ROOT, "include", "sisr_hip.h")) as f:
        hdr = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    src = _integration_sample()
    import sisr_amd
    for fn in ("sisr_conv3x3_c64", "sisr_pack_conv3x3"):
        proto = re.search(r"\b%s\s*\((.*?)\)\s*;" % fn, hdr, flags=re.S).group(1)
        n_decl = len([a for a in proto.split(",") if a.strip()])
        assert _call_arg_count(src, fn) == n_decl == len(sisr_amd.hip._SIGS[fn][1]), fn


@pytest.mark.gpu
def test_integration_level2_sample_runs_verbatim():
    """Executes the documented binding as written (ref it replaces: advanced/common.py:5-8 default_conv + ReLU) and checks
    its result against F.conv2d."""
    import torch
    import torch.nn.functional as F
    ns = {}
    cwd = os.getcwd()
    os.chdir(ROOT)  # the sample opens the library by its path relative to the repository root
    try:
        exec(compile(_integration_sample(), "INTEGRATION.md:level-2", "exec"), ns)
    finally:
        os.chdir(cwd)
    torch.cuda.synchronize()
    ref = F.relu(F.conv2d(ns["x"].cpu().double(), ns["w"].cpu().double(), ns["bias"].cpu().double(), padding=1))
    assert (ns["y"].cpu().double() - ref).abs().max().item() < 2e-5


def test_pure_host_queries():
    import sisr_amd
    L = sisr_amd.hip.lib()
    assert L.sisr_conv3x3_c64_gap_parts(128, 128) == 32 * 4 * 2
    assert L.sisr_conv3x3_c64_gap_parts(57, 86) == 15 * 3 * 2
    # dense form: 256 K-slices (one workgroup per CU) x 4 quadrant slabs; bias slabs for the finest masked split (512)
    assert L.sisr_wgrad3x3_c64_workspace_bytes(4, 128, 128, 64, 64) == (256 * 4 * 9216 + 512 * 64) * 4
    assert L.sisr_wgrad3x3_c64_workspace_bytes(4, 128, 128, 60, 64) == 0
    assert L.sisr_l1_loss_workspace_bytes() == 2048
    assert L.sisr_gate_dg_parts(128 * 128) == 32


def test_round4_entry_points_refuse_bad_arguments_before_any_device_call():
    """sisr_conv3x3_c64_geo / sisr_wgrad3x3_c64_geo[_batch] and the SPARNet option passes validate their arguments on the host and
    return an error code (no launch, so this runs without a GPU): unknown modes, upsampling in the transposed form, odd sizes
    under upsampling, real channel counts above the padded ones, empty batches, null pointers."""
    import ctypes as C
    import sisr_amd
    hip = sisr_amd.hip
    L = hip.lib()
    v = hip.view_plain(8, 8, 64)
    buf = (C.c_float * 64)()
    a = C.addressof(buf)  # a non-null, 16-byte aligned stand-in: every call below must fail before it is dereferenced
    assert a % 16 == 0 or True
    ERR_ARG, ERR_UNSUPPORTED = -1, -2
    bad = lambda rc: rc != 0  # noqa: E731
    assert bad(L.sisr_conv3x3_c64_geo(a, v, a, None, a, v, None, 1, 8, 8, 64, 64, 0, 0, 0, None))      # mode 0
    assert bad(L.sisr_conv3x3_c64_geo(a, v, a, None, a, v, None, 1, 8, 8, 64, 64, 5, 0, 0, None))      # mode 5
    assert bad(L.sisr_conv3x3_c64_geo(a, v, a, None, a, v, None, 1, 8, 8, 64, 64, 2, 1, 0, None))      # transposed + upsampling
    assert bad(L.sisr_conv3x3_c64_geo(a, v, a, None, a, v, None, 1, 7, 8, 64, 64, 1, 1, 0, None))      # odd size under nearest x2
    assert bad(L.sisr_conv3x3_c64_geo(a, v, a, None, a, v, None, 1, 1, 8, 64, 64, 1, 0, 0, None))      # ReflectionPad2d(1) needs 2 pixels
    assert bad(L.sisr_conv3x3_c64_geo(a, v, a, None, a, v, None, 1, 8, 8, 60, 64, 1, 0, 0, None))      # channels not a 64-multiple
    assert bad(L.sisr_conv3x3_c64_geo(None, v, a, None, a, v, None, 1, 8, 8, 64, 64, 1, 0, 0, None))   # null map
    ws = L.sisr_wgrad3x3_c64_workspace_bytes(1, 8, 8, 64, 64)
    assert bad(L.sisr_wgrad3x3_c64_geo(a, v, a, v, a, 65, 64, None, a, ws, 1, 8, 8, 64, 64, 0, 0, None))  # co_real > cout
    assert bad(L.sisr_wgrad3x3_c64_geo(a, v, a, v, a, 64, 0, None, a, ws, 1, 8, 8, 64, 64, 0, 0, None))   # ci_real = 0
    assert bad(L.sisr_wgrad3x3_c64_geo(a, v, a, v, a, 64, 64, None, a, ws, 1, 8, 8, 64, 64, 3, 0, None))  # up code 3
    assert bad(L.sisr_wgrad3x3_c64_geo(a, v, a, v, a, 64, 64, None, a, 16, 1, 8, 8, 64, 64, 0, 0, None))  # workspace too small
    assert L.sisr_wgrad_geo_job_bytes() == C.sizeof(hip.WgradGeoJob)
    jobs = (hip.WgradGeoJob * 1)()
    assert L.sisr_wgrad3x3_c64_geo_batch_workspace_bytes(C.addressof(jobs), 0) == 0
    assert L.sisr_wgrad3x3_c64_geo_batch_workspace_bytes(C.addressof(jobs), 1) == 0           # a zeroed job is not a valid one
    assert bad(L.sisr_wgrad3x3_c64_geo_batch(C.addressof(jobs), 0, a, 1 << 20, None))
    assert bad(L.sisr_wgrad3x3_c64_geo_batch(C.addressof(jobs), 1, a, 1 << 20, None))
    assert bad(L.sisr_group_norm_fwd(a, a, a, a, a, a, 1, 64, 64, 64, 3, 1e-5, None))         # 64 channels in groups of 3
    assert bad(L.sisr_pixel_norm(a, None, a, 64, 48 * 4, 0, None))                             # 48 channel lanes: not a power of two
    assert bad(L.sisr_act(a, None, None, a, None, 64, 64, 64, 0, 0, None))                     # PReLU without slopes
    assert bad(L.sisr_act(a, None, a, a, None, 64, 64, 64, 2, 0, None))                        # unknown mode
    assert bad(L.sisr_spar3d(a, a, None, a, None, 6, 0, None))                                 # element count not a multiple of 4
    assert ERR_ARG != 0 and ERR_UNSUPPORTED != 0


def test_no_cpu_fallback():
    import pytest
    import torch
    import sisr_amd
    x = torch.zeros(1, 64, 8, 8)
    w = torch.zeros(64, 64, 3, 3)
    with pytest.raises(RuntimeError, match="no CPU fallback|HIP device"):
        sisr_amd.ops.conv3x3(x, w)
