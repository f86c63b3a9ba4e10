"""The C-ABI library loads and exports every symbol include/sisr_hip.h declares (no GPU needed)."""
import ctypes
import os
import re

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
                 "sisr_diag_conv_occupancy"):
        assert not hasattr(lib, name), name


def test_pure_host_queries():
    import sisr_amd
    L = sisr_amd.hip.lib()
    assert L.sisr_conv3x3_c64_gap_parts(128, 128) == 32 * 4 * 2
    assert L.sisr_conv3x3_c64_gap_parts(57, 86) == 15 * 3 * 2
    assert L.sisr_wgrad3x3_c64_workspace_bytes(4, 128, 128, 64, 64) == (128 * 4 * 9216 + 512 * 64) * 4  # 512 K-slices x units; bias slabs for the finest masked split
    assert L.sisr_wgrad3x3_c64_workspace_bytes(4, 128, 128, 60, 64) == 0
    assert L.sisr_l1_loss_workspace_bytes() == 2048
    assert L.sisr_gate_dg_parts(128 * 128) == 32


def test_no_cpu_fallback():
    import pytest
    import torch
    import sisr_amd
    x = torch.zeros(1, 64, 8, 8)
    w = torch.zeros(64, 64, 3, 3)
    with pytest.raises(RuntimeError, match="no CPU fallback|HIP device"):
        sisr_amd.ops.conv3x3(x, w)
