"""Host-buffer front end of `eight_layers_net`: the reference's caller owns HOST streams (`hls::stream<ap_uint<24>>` filled
and drained by the testbench, conv3_nonsquare_tb.cpp:789-821), so a drop-in that is handed host memory has to cross PCIe.

`HostPipeline` keeps `depth` device-side slots (input batch, reconstruction, latent) and three HIP streams — upload, compute,
download — tied together by events, so that batch i+1 goes up and batch i-1 comes down while batch i is in the kernels:

    up     :  H2D(i+1)                      waits for: compute(i+1-depth) has read the slot's input
    compute:  eight_layers_net(i)           waits for: H2D(i), D2H(i-depth) has drained the slot's outputs
    down   :  D2H(i-1)                      waits for: compute(i-1)

Nothing here computes: the kernels are the library's (api.EightLayersNet.forward on the compute stream); this is plumbing
around them, and the PCIe-inclusive rate it reaches is what DESIGN.md §3.5 quotes beside the HBM-resident headline."""
from __future__ import annotations

from typing import Optional, Sequence

__all__ = ["HostPipeline"]


class HostPipeline:
    def __init__(self, net, n_images: int, depth: int = 2, want_latent: bool = True):
        import torch
        if depth < 1:
            raise ValueError("depth >= 1")
        self.net, self.n, self.depth, self.want_latent = net, int(n_images), int(depth), bool(want_latent)
        dev = net.device
        d = net.descs
        u8 = dict(dtype=torch.uint8, device=dev)
        self.d_in = [torch.empty((self.n,) + d[0].in_shape, **u8) for _ in range(depth)]
        self.d_out = [torch.empty((self.n,) + d[-1].out_shape, **u8) for _ in range(depth)]
        self.d_lat = [torch.empty((self.n,) + d[3].out_shape, **u8) for _ in range(depth)] if want_latent and len(d) > 3 else None
        self.up, self.compute, self.down = (torch.cuda.Stream(device=dev) for _ in range(3))
        mk = lambda: [torch.cuda.Event() for _ in range(depth)]
        self.ev_up, self.ev_compute, self.ev_down = mk(), mk(), mk()
        self._used = [False] * depth
        net.workspace(self.n)

    @staticmethod
    def pinned(shape):
        """A page-locked host tensor (asynchronous copies need one)."""
        import torch
        return torch.empty(shape, dtype=torch.uint8).pin_memory()

    def run(self, host_in: Sequence, host_out: Sequence, host_latent: Optional[Sequence] = None) -> None:
        """host_in[i] -> host_out[i] (and host_latent[i]) for every batch i; pinned uint8 tensors of the net's batch shapes.
        Returns after everything has been ENQUEUED; `synchronize()` waits for the last download."""
        import torch
        if len(host_in) != len(host_out) or (host_latent is not None and len(host_latent) != len(host_in)):
            raise ValueError("one output (and latent) buffer per input batch")
        for t in list(host_in) + list(host_out) + list(host_latent or []):
            if not (isinstance(t, torch.Tensor) and not t.is_cuda and t.is_pinned() and t.dtype == torch.uint8 and t.is_contiguous()):
                raise TypeError("host buffers must be contiguous pinned uint8 CPU tensors (HostPipeline.pinned)")
        for i in range(len(host_in)):
            s = i % self.depth
            with torch.cuda.stream(self.up):
                if self._used[s]:
                    self.up.wait_event(self.ev_compute[s])       # the slot's previous batch has been read by layer 0
                self.d_in[s].copy_(host_in[i], non_blocking=True)
                self.ev_up[s].record(self.up)
            with torch.cuda.stream(self.compute):
                self.compute.wait_event(self.ev_up[s])
                if self._used[s]:
                    self.compute.wait_event(self.ev_down[s])     # the slot's previous outputs have left
                self.net.forward(self.d_in[s], self.d_out[s], self.d_lat[s] if self.d_lat else None,
                                 want_latent=self.d_lat is not None, stream=self.compute)
                self.ev_compute[s].record(self.compute)
            with torch.cuda.stream(self.down):
                self.down.wait_event(self.ev_compute[s])
                host_out[i].copy_(self.d_out[s], non_blocking=True)
                if host_latent is not None and self.d_lat:
                    host_latent[i].copy_(self.d_lat[s], non_blocking=True)
                self.ev_down[s].record(self.down)
            self._used[s] = True

    def synchronize(self) -> None:
        self.down.synchronize()
        self.compute.synchronize()
        self.up.synchronize()
