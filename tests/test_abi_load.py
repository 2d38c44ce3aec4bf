"""CPU-only checks of the C-ABI shared library: it loads, exports every symbol include/sicn.h
declares, and its pure-host entry points behave (no compute calls — those need the GPU)."""
import ctypes
import re
from dataclasses import replace
from pathlib import Path

import pytest

from simple_image_compression_network_amd import _lib
from simple_image_compression_network_amd.config import REFERENCE_DESCS, eight_layer_descs

ROOT = Path(__file__).resolve().parent.parent


def _declared_symbols():
    text = (ROOT / "include" / "sicn.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sicn_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = _lib.lib()        # (loads the HIP runtime PyTorch ships first, then libsicn.so)
    syms = _declared_symbols()
    assert len(syms) >= 16
    for s in syms:
        assert hasattr(L, s), f"libsicn.so does not export {s}"
    assert set(syms) == set(_lib.ABI), "python binding table and sicn.h disagree"


def test_version_and_strerror():
    L = _lib.lib()
    assert L.sicn_version() >= 1
    assert L.sicn_strerror(0) == b"ok"
    assert b"invalid" in L.sicn_strerror(-22)


def test_validate_desc_matches_python_validate():
    L = _lib.lib()
    for d in REFERENCE_DESCS + eight_layer_descs(1920, 1080) + eight_layer_descs(37, 21):
        assert L.sicn_validate_desc(ctypes.byref(d.to_c())) == 0
    d = REFERENCE_DESCS[1]
    bad = [replace(d, K=3), replace(d, S=1), replace(d, SIMD=7), replace(d, PE=5), replace(d, W_TILES=1),
           replace(d, OFM_ROW=d.OFM_ROW + 1), replace(d, IN_BIT=4), replace(d, W_BIT=8), replace(d, IFM_ROW=0),
           replace(d, transposed=1), replace(d, transposed=2)]
    for b in bad:
        assert L.sicn_validate_desc(ctypes.byref(b.to_c())) == -22
        with pytest.raises(ValueError):
            b.validate()
    assert L.sicn_validate_desc(None) == -22


def test_kernel_selection_for_reference_net():
    L = _lib.lib()
    kinds = [L.sicn_kernel_for(ctypes.byref(d.to_c())).decode() for d in REFERENCE_DESCS]
    assert kinds == ["l0_rgb", "mfma_conv", "mfma_conv", "mfma_conv", "mfma_deconv", "mfma_deconv",
                     "mfma_deconv", "l7_rgb"]
    odd = replace(REFERENCE_DESCS[1], IFM_CH=6, OFM_CH=4, SIMD=3, PE=2, W_TILES=(4 // 2) * (150 // 3))
    assert L.sicn_kernel_for(ctypes.byref(odd.to_c())) == b"generic"


def test_null_arguments_are_rejected_without_touching_the_gpu():
    L = _lib.lib()
    d = REFERENCE_DESCS[0].to_c()
    out = ctypes.c_void_p()
    assert L.sicn_weights_from_finn_tiles(ctypes.byref(d), None, 8, None, ctypes.byref(out)) == -22
    assert L.sicn_conv2d(ctypes.byref(d), None, None, None, 1, None) == -22
    assert L.sicn_net_create(None, None, 0, ctypes.byref(out)) == -22
    assert L.sicn_net_workspace_bytes(None, 1) == 0


def test_options_defaults_and_validation_without_gpu():
    """sicn_options: plain data per call / per net; defaults come from sicn_options_init; out-of-range fields are
    rejected before anything touches the GPU (the library has no process-wide switches any more)."""
    L = _lib.lib()
    o = _lib.COptions()
    L.sicn_options_init(ctypes.byref(o))
    assert o.struct_bytes == ctypes.sizeof(_lib.COptions) == 64
    assert (o.force_generic, o.strip_chunks, o.no_phase_layout) == (0, 0, 0)
    d = REFERENCE_DESCS[1].to_c()
    out = ctypes.c_void_p()
    assert L.sicn_has_alt_kernels() == 0       # the product build: the alternate kernel families are in libsicn_alt.so only
    for field, bad in (("mfma_shape", 8), ("mfma_shape", 32), ("tile_x", 24), ("strip_chunks", -1), ("no_phase_layout", 3), ("split_n", 7),
                       ("struct_bytes", 4), ("struct_bytes", 4096)):
        o = _lib.make_options()
        setattr(o, field, bad)
        assert L.sicn_conv2d_opt(ctypes.byref(d), None, None, None, 1, ctypes.byref(o), None) == -22
        assert L.sicn_net_create_opt(ctypes.byref(d), ctypes.byref(out), 1, ctypes.byref(o), ctypes.byref(out)) == -22
    assert not hasattr(L, "sicn_set_force_generic")


def test_isa_hazard_checker_flags_the_known_bugs():
    """tools/isa_hazards.py (run over the asm kernels' ISA by build()): its self-test holds the two round-2 bugs and the round-3
    one as ISA snippets — a VALU-unpacked bias read as an asm MFMA's C operand, an asm atomic whose address SGPRs come from a
    v_readlane, an accumulator read out right behind the asm MFMA that wrote it — plus an LDS fragment used without a wait;
    every one must be flagged and a clean accumulate chain must pass."""
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, str(root / "tools" / "isa_hazards.py"), "--selftest"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok ") == 5 and "FAIL" not in r.stdout
