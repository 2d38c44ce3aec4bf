"""Layer descriptors for the conv/deconv transform path.

A `LayerDesc` carries exactly the `CONV_n_*` macro set of the reference's generated config
(config_nonsquare.h:2-16) as run-time fields, plus `transposed` (0 = `conv2d<>`,
conv_nonsquare_top.cpp:198-280; 1 = `deconv522<>`, conv_nonsquare_top.cpp:71-195).

Naming follows the reference, including its trap: `IFM_ROW` is the WIDTH (x, the fast stream
dimension) and `IFM_COL` is the HEIGHT (y) — conv_nonsquare_top.cpp:283-285 maps
`CONV_0_IFM_ROW -> IFMDim1_x`, and the testbench streams `ox` innermost
(conv3_nonsquare_tb.cpp:789-790).  A stream of H*W words of C*8 bits is byte-identical to a
row-major `[H][W][C] uint8` array, which is the only tensor layout used here.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass, fields
from typing import List, Sequence, Tuple

__all__ = ["LayerDesc", "CLayerDesc", "eight_layer_descs", "REFERENCE_DESCS", "NET_CHANNELS"]


class CLayerDesc(ctypes.Structure):
    """ctypes image of `sicn_layer_desc` (include/sicn.h)."""

    _fields_ = [(n, ctypes.c_int32) for n in (
        "K", "S", "P", "IFM_CH", "IFM_ROW", "IFM_COL", "OFM_CH", "OFM_ROW", "OFM_COL",
        "SIMD", "PE", "IN_BIT", "OUT_BIT", "W_BIT", "W_TILES", "transposed")]


@dataclass(frozen=True)
class LayerDesc:
    K: int = 5
    S: int = 2
    P: int = 2
    IFM_CH: int = 3
    IFM_ROW: int = 768      # width  (x)
    IFM_COL: int = 512      # height (y)
    OFM_CH: int = 128
    OFM_ROW: int = 384      # width  (x)
    OFM_COL: int = 256      # height (y)
    SIMD: int = 3
    PE: int = 8
    IN_BIT: int = 8
    OUT_BIT: int = 8
    W_BIT: int = 4
    W_TILES: int = 400
    transposed: int = 0

    # ---- derived ---------------------------------------------------------------------------
    @property
    def SF(self) -> int:
        """Synapse fold = MatrixW / SIMD (mvau.hpp:105)."""
        return self.K * self.K * self.IFM_CH // self.SIMD

    @property
    def NF(self) -> int:
        """Neuron fold = MatrixH / PE (mvau.hpp:101)."""
        return self.OFM_CH // self.PE

    @property
    def in_shape(self) -> Tuple[int, int, int]:
        return (self.IFM_COL, self.IFM_ROW, self.IFM_CH)     # [H][W][C]

    @property
    def out_shape(self) -> Tuple[int, int, int]:
        return (self.OFM_COL, self.OFM_ROW, self.OFM_CH)

    @property
    def algorithmic_macs(self) -> int:
        """Zero-skipped MAC count: conv OH*OW*Cout*25*Cin, deconv IH*IW*Cin*Cout*25."""
        kk = self.K * self.K * self.IFM_CH * self.OFM_CH
        if self.transposed:
            return self.IFM_COL * self.IFM_ROW * kk
        return self.OFM_COL * self.OFM_ROW * kk

    def validate(self) -> None:
        """Shape preconditions of the reference (slidingwindow.h:1259, mvau.hpp:101-105,
        conv_nonsquare_top.cpp:94-95,246-259) — same checks as csrc/sicn_abi.hip."""
        if (self.K, self.S, self.P) != (5, 2, 2):
            raise ValueError("only the reference's K=5,S=2,P=2 layer geometry is defined")
        if (self.IN_BIT, self.OUT_BIT, self.W_BIT) != (8, 8, 4):
            raise ValueError("only IN_BIT=8, OUT_BIT=8, W_BIT=4 is defined")
        if min(self.IFM_CH, self.OFM_CH, self.IFM_ROW, self.IFM_COL, self.SIMD, self.PE) <= 0:
            raise ValueError("non-positive dimension")
        if self.IFM_CH % self.SIMD or self.OFM_CH % self.PE:
            raise ValueError("IFM_CH % SIMD or OFM_CH % PE != 0")
        if self.W_TILES != self.NF * self.SF:
            raise ValueError("W_TILES != (OFM_CH/PE)*(25*IFM_CH/SIMD)")
        if self.transposed:
            exp = (2 * self.IFM_ROW, 2 * self.IFM_COL)
        else:
            exp = ((self.IFM_ROW + 1) // 2, (self.IFM_COL + 1) // 2)
        if (self.OFM_ROW, self.OFM_COL) != exp:
            raise ValueError(f"OFM dims {(self.OFM_ROW, self.OFM_COL)} != {exp}")

    def to_c(self) -> CLayerDesc:
        return CLayerDesc(**{f.name: getattr(self, f.name) for f in fields(self)})


# (IFM_CH, OFM_CH, SIMD, PE, transposed) per layer — config_nonsquare.h:1-135
NET_CHANNELS: Sequence[Tuple[int, int, int, int, int]] = (
    (3, 128, 3, 8, 0),
    (128, 128, 8, 16, 0),
    (128, 128, 8, 16, 0),
    (128, 192, 8, 24, 0),
    (192, 128, 12, 16, 1),
    (128, 128, 8, 16, 1),
    (128, 128, 8, 16, 1),
    (128, 3, 8, 3, 1),
)


def eight_layer_descs(width: int, height: int) -> List[LayerDesc]:
    """The reference's 8-layer net (conv_nonsquare_top.cpp:295-357) re-dimensioned for a
    `height x width` RGB image: conv out = ceil(in/2), deconv out = 2*in."""
    descs = []
    w, h = width, height
    for cin, cout, simd, pe, tr in NET_CHANNELS:
        ow, oh = (2 * w, 2 * h) if tr else ((w + 1) // 2, (h + 1) // 2)
        d = LayerDesc(IFM_CH=cin, IFM_ROW=w, IFM_COL=h, OFM_CH=cout, OFM_ROW=ow, OFM_COL=oh,
                      SIMD=simd, PE=pe, W_TILES=(cout // pe) * (25 * cin // simd), transposed=tr)
        d.validate()
        descs.append(d)
        w, h = ow, oh
    return descs


#: config_nonsquare.h verbatim (768 wide x 512 high).
REFERENCE_DESCS: List[LayerDesc] = eight_layer_descs(768, 512)
