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
                       ("split_k", 9), ("struct_bytes", 4), ("struct_bytes", 4096)):
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


def _plan(d, n_images, n_cu, **opts):
    L = _lib.lib()
    out = (ctypes.c_int32 * 12)()
    o = _lib.make_options(**opts)
    assert L.sicn_debug_plan(ctypes.byref(d.to_c()), n_images, ctypes.byref(o), n_cu, out) == 0
    keys = ("n_cu", "n_xcd", "kind", "family", "tile_x", "split_n", "split_k", "gx", "gy", "gz", "chunks", "ty_per")
    return dict(zip(keys, list(out)))


def test_launch_planning_follows_the_chip_size_without_gpu():
    """Nothing in csrc/ hard-wires 256 CUs / 8 XCDs: grids, strip cuts, the wide / narrow / split choices
    come from sicn_plan.h as functions of the device's CU count.  Walked here for a whole MI355X (256), a DPX partition (128)
    and a CPX partition (32), no GPU needed."""
    from simple_image_compression_network_amd.config import eight_layer_descs
    L = _lib.lib()
    # XCD rule and the work-list mapping: a bijection onto the items for every XCD count, padding workgroups get -1
    for n_cu, n_xcd in ((256, 8), (240, 8), (304, 8), (128, 4), (64, 2), (32, 1), (36, 1), (1, 1)):
        assert _plan(eight_layer_descs(256, 256)[0], 1, n_cu)["n_xcd"] == n_xcd
    for n_xcd in (1, 2, 4, 8):
        for n_items in (1, 7, 8, 9, 255, 1000):
            grid = (n_items + n_xcd - 1) // n_xcd * n_xcd
            items = [L.sicn_debug_xcd_item(b, n_items, n_xcd) for b in range(grid)]
            assert sorted(i for i in items if i >= 0) == list(range(n_items))
            assert items.count(-1) == grid - n_items
    d4k = eight_layer_descs(3840, 2160)
    # 8 x 4K: layers 1, 2, 5, 6 on the wide persistent kernels, ONE workgroup per CU of whatever chip this is
    for n_cu in (256, 128, 32):
        for l in (1, 2, 5, 6):
            p = _plan(d4k[l], 8, n_cu)
            assert p["family"] == 2 and p["gx"] == n_cu, (n_cu, l, p)
        # ... and part of the tiles dealt dynamically across the XCDs only where a workgroup walks >= 16 of them (32 in layers 1 / 6, 8 in 2 / 5
        # on the whole chip, 64 and 15.9 on half of it; every layer of a 32-CU partition; the slot is "chunks" for the other kinds)
        assert [_plan(d4k[l], 8, n_cu)["chunks"] for l in (1, 2, 5, 6)] == {256: [1, 0, 0, 1], 128: [1, 0, 0, 1], 32: [1, 1, 1, 1]}[n_cu]
        p7 = _plan(d4k[7], 8, n_cu)                     # 60 strips x 8 images = 480 workgroups: never cut, grid padded to the XCD count
        assert p7["chunks"] == 1 and p7["gx"] % p7["n_xcd"] == 0 and p7["gx"] == 480
    # one 1080p image: the wide conv needs >= 3.5 full rounds of the CUs -> not on 256, but on a 32-CU partition
    d1080 = eight_layer_descs(1920, 1080)
    assert _plan(d1080[1], 1, 256)["family"] == 1 and _plan(d1080[1], 1, 256)["tile_x"] == 32
    assert _plan(d1080[1], 1, 32)["family"] == 2 and _plan(d1080[1], 1, 32)["gx"] == 32
    # layer 7 strips of one 1080p image: about two workgroups per CU in all
    assert _plan(d1080[7], 1, 256)["chunks"] == 512 // 30
    assert _plan(d1080[7], 1, 128)["chunks"] == 256 // 30
    assert _plan(d1080[7], 1, 32)["chunks"] == 64 // 30
    # the channel split is a small-grid measure: on for a 256^2 image on the whole chip, off on a 32-CU partition; the K split is
    # never automatic (measured a loss on this chip, sicn_plan.h) and takes IFM_CH / 64 slices when forced
    d256 = eight_layer_descs(256, 256)
    p = _plan(d256[2], 1, 256)
    assert (p["family"], p["tile_x"], p["split_n"], p["split_k"], p["gy"], p["gz"]) == (1, 16, 2, 1, 2, 1)
    assert (_plan(d256[3], 1, 256)["split_n"], _plan(d256[4], 1, 256)["split_n"]) == (3, 2)
    p = _plan(d256[1], 1, 32)                           # 32 tiles on 32 CUs: no split
    assert (p["split_n"], p["split_k"]) == (1, 1)
    # layer 0 runs: at most 9 tiles per workgroup, about four workgroups per CU on small images
    p0 = _plan(d4k[0], 8, 256)
    assert p0["ty_per"] <= 9 and p0["gy"] * p0["ty_per"] >= (1080 + 7) // 8
    d768 = eight_layer_descs(768, 512)
    assert _plan(d768[0], 1, 256)["gy"] == 32 and _plan(d768[0], 1, 32)["gy"] == 11


def test_product_library_rejects_the_alt_only_forms_and_the_alt_session_runs():
    """the kernel forms that measured a loss (k_l0p, k_l7s, k_l7g, the K split) are built into libsicn_alt.so
    only.  The product library answers their options with SICN_EINVAL; the planning asserts that need them (below, marked `alt`)
    run in a child pytest on the ALT library — no GPU needed for sicn_debug_plan."""
    from conftest import run_alt_session
    L = _lib.lib()
    d = REFERENCE_DESCS[1].to_c()
    out = ctypes.c_void_p()
    plan = (ctypes.c_int32 * 12)()
    for opt in ({"l0_form": 2}, {"l7_loader": 2}, {"gdn_fuse": 2}, {"split_k": 2}):
        o = _lib.make_options(**opt)
        assert L.sicn_conv2d_opt(ctypes.byref(d), None, None, None, 1, ctypes.byref(o), None) == -22, opt
        assert L.sicn_net_create_opt(ctypes.byref(d), ctypes.byref(out), 1, ctypes.byref(o), ctypes.byref(out)) == -22, opt
        assert L.sicn_debug_plan(ctypes.byref(d), 1, ctypes.byref(o), 256, plan) == -22, opt
    for opt in ({"l0_form": 1}, {"l7_loader": 1}, {"gdn_fuse": 1}, {"split_k": 1}):      # "never" stays a valid request
        o = _lib.make_options(**opt)
        assert L.sicn_debug_plan(ctypes.byref(d), 1, ctypes.byref(o), 256, plan) == 0, opt
    r = run_alt_session("alt and not gpu", timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


@pytest.mark.alt
def test_alt_launch_planning_of_the_k_split_and_the_fused_rgb_layer():
    """The launch plans of the ALT-only forms (run by the driver test above on libsicn_alt.so): the K split is never automatic and takes
    IFM_CH / 64 slices when forced; k_l7g's strip cut."""
    from simple_image_compression_network_amd.config import eight_layer_descs
    assert _lib.lib().sicn_has_alt_kernels() == 1
    d256, d1080, d4k, d768 = (eight_layer_descs(*wh) for wh in ((256, 256), (1920, 1080), (3840, 2160), (768, 512)))
    p = _plan(d256[3], 1, 256, split_k=2)
    assert (p["split_n"], p["split_k"], p["gz"]) == (3, 2, 2)
    p = _plan(d256[4], 1, 256, split_k=2)               # 192 input channels: three channel-group pairs
    assert (p["split_n"], p["split_k"], p["gz"]) == (2, 3, 3)
    assert _plan(d1080[2], 1, 256, split_n=2, split_k=2, tile_x=16)["split_k"] == 2      # forced on a full grid
    # the RGB layer behind a layer with an activation (k_l7g, gdn_fuse = 2): strips of 62 columns, steps of 4 rows, one workgroup per
    # CU at a time; the cut = fewest step-times on the busiest CU (a cut costs about two steps), the smallest such
    def l7g_cut(iw, ih, n, n_cu):
        strips, steps = (iw + 61) // 62 * n, (ih + 3) // 4
        cost = lambda y: -(-strips * y // n_cu) * (-(-steps // y) + 2)
        return min(range(1, min(steps, 64) + 1), key=lambda y: (cost(y), y))
    for d, n in ((d4k[7], 8), (d1080[7], 1), (d768[7], 2)):
        for n_cu in (256, 128, 32):
            p = _plan(d, n, n_cu, gdn_fuse=2)
            yc = l7g_cut(d.IFM_ROW, d.IFM_COL, n, n_cu)
            assert p["chunks"] == yc and p["gx"] == (d.IFM_ROW + 61) // 62 * yc * n and p["ty_per"] * yc >= (d.IFM_COL + 3) // 4, (n, n_cu, p)
    assert _plan(d4k[7], 8, 256, gdn_fuse=2)["chunks"] == 1 and _plan(d1080[7], 1, 256, gdn_fuse=2)["chunks"] == 16   # 960 x 544: 16 strips x 16 runs of 9 steps = one workgroup per CU
