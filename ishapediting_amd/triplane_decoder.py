"""Host-side mirror of triplane_decoder/axisnetworks.py:MultiTriplane and the dense-grid decode of
triplane_decoder/visualize.py:create_obj_o3d, backed by the fp32-MFMA decode kernel.

`decoder.net.load_state_dict(...)`, `decoder.embeddings[i] = plane[1,32,S,S]` and
`decoder(obj_idx, coords[1,N,3]) -> logits[1,N,1]` work as in the reference
(drag_utils.py:188,246,295-298,455).  The grid decode keeps the whole 256^3 volume on the device
instead of 336 host round trips (visualize.py:89-95).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional

import torch

from . import _lib
from .unet_spec import DECODER_SHAPES


class _Net:
    """Stands in for the nn.Sequential `net` (axisnetworks.py:526-535): holds the seven tensors on the device."""

    def __init__(self, device):
        self.device = device
        self.sd: Dict[str, torch.Tensor] = {k: torch.zeros(s, dtype=torch.float32, device=device)
                                            for k, s in DECODER_SHAPES.items()}
        self.loaded = False

    def load_state_dict(self, sd, strict: bool = True):
        missing = [k for k in DECODER_SHAPES if k not in sd]
        unexpected = [k for k in sd if k not in DECODER_SHAPES]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict: missing {missing}, unexpected {unexpected}")
        for k, shape in DECODER_SHAPES.items():
            if k in sd:
                v = sd[k]
                if tuple(v.shape) != tuple(shape):
                    raise RuntimeError(f"size mismatch for {k}")
                self.sd[k] = v.detach().to(device=self.device, dtype=torch.float32).contiguous()
        self.loaded = True

    def state_dict(self):
        return dict(self.sd)

    def parameters(self):
        return iter(self.sd.values())

    def weights_c(self) -> _lib.DecoderWeightsC:
        s = self.sd
        return _lib.DecoderWeightsC(s["0._B"].data_ptr(), s["1.weight"].data_ptr(), s["1.bias"].data_ptr(),
                                    s["3.weight"].data_ptr(), s["3.bias"].data_ptr(), s["5.weight"].data_ptr(),
                                    s["5.bias"].data_ptr())


class MultiTriplane:
    def __init__(self, num_objs: int = 1, input_dim: int = 3, output_dim: int = 1, noise_val=None, device="cuda"):
        assert input_dim == 3 and output_dim == 1
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("MultiTriplane needs a GPU device: there is no CPU fallback")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.num_objs = num_objs
        self.embeddings: List[torch.Tensor] = [torch.randn(1, 32, 128, 128, device=self.device) * 0.001
                                               for _ in range(3 * num_objs)]
        self.net = _Net(self.device)
        self.training = False

    def to(self, device):
        return self

    def eval(self):
        self.training = False
        return self

    def _planes(self, obj_idx: int) -> torch.Tensor:
        """embeddings[3*obj..] ([1,32,S,S] each) -> channels-last [3][S][S][32] for the kernel."""
        e = [self.embeddings[3 * obj_idx + i] for i in range(3)]
        S = e[0].shape[-1]
        latent = torch.cat([t.reshape(1, 32, S, S) for t in e], dim=1).to(device=self.device, dtype=torch.float32).contiguous()
        planes = torch.empty((3, S, S, 32), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().ishap_planes_prepare(latent.data_ptr(), None, None, S, planes.data_ptr(),
                                                       _lib.stream_ptr(self.device)))
        return planes

    def forward(self, obj_idx: int, coordinates: torch.Tensor, debug: bool = False) -> torch.Tensor:
        """axisnetworks.py:546-562: coordinates [B, N, 3] -> logits [B, N, 1]."""
        b, n, d = coordinates.shape
        assert d == 3
        planes = self._planes(obj_idx)
        coords = coordinates.detach().to(device=self.device, dtype=torch.float32).reshape(-1, 3).contiguous()
        out = torch.empty(coords.shape[0], dtype=torch.float32, device=self.device)
        w = self.net.weights_c()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().ishap_triplane_decode_points(planes.data_ptr(), planes.shape[1], C.byref(w),
                                                               coords.data_ptr(), coords.shape[0], out.data_ptr(),
                                                               _lib.stream_ptr(self.device)))
        return out.reshape(b, n, 1)

    __call__ = forward

    def points_loss_grad(self, planes: torch.Tensor, coords: torch.Tensor, gt: torch.Tensor):
        """drag_utils.py:455-458 on explicit planes: loss = -BCEWithLogitsLoss()(decoder(coords), gt) and
        d loss / d planes ([3,S,S,32]); also returns the logits."""
        coords = coords.detach().to(device=self.device, dtype=torch.float32).reshape(-1, 3).contiguous()
        gt = gt.detach().to(device=self.device, dtype=torch.float32).reshape(-1).contiguous()
        assert coords.shape[0] == gt.shape[0]
        sd = self.net.sd
        w1t, w2t = sd["1.weight"].t().contiguous(), sd["3.weight"].t().contiguous()
        dplanes = torch.empty_like(planes)
        loss = torch.empty(1, dtype=torch.float32, device=self.device)
        logits = torch.empty(coords.shape[0], dtype=torch.float32, device=self.device)
        w = self.net.weights_c()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().ishap_triplane_points_loss_grad(
                planes.data_ptr(), planes.shape[1], C.byref(w), w1t.data_ptr(), w2t.data_ptr(), coords.data_ptr(),
                gt.data_ptr(), coords.shape[0], dplanes.data_ptr(), loss.data_ptr(), logits.data_ptr(),
                _lib.stream_ptr(self.device)))
        return loss, dplanes, logits


def prepare_planes(latent: torch.Tensor, rng: Optional[torch.Tensor], mid: Optional[torch.Tensor]) -> torch.Tensor:
    """(tri_feat * range + middle).reshape(3,32,S,S) (drag_utils.py:295) as channels-last planes, one kernel."""
    assert latent.shape[0] == 1 and latent.shape[1] == 96
    dev = latent.device
    S = latent.shape[-1]
    latent = latent.detach().to(dtype=torch.float32).contiguous()

    def vec(v, default):
        if v is None:
            return None
        if torch.is_tensor(v):
            v = v.detach().to(device=dev, dtype=torch.float32).reshape(-1)
            return (v.expand(96) if v.numel() == 1 else v).contiguous()
        return None if float(v) == default else torch.full((96,), float(v), dtype=torch.float32, device=dev)
    r, m = vec(rng, 1.0), vec(mid, 0.0)
    planes = torch.empty((3, S, S, 32), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().ishap_planes_prepare(latent.data_ptr(), _lib.ptr(r), _lib.ptr(m), S, planes.data_ptr(),
                                                   _lib.stream_ptr(dev)))
    return planes


def decode_planes_grid(decoder: MultiTriplane, planes: torch.Tensor, res: int) -> torch.Tensor:
    """visualize.py:79-97: occupancy logits on linspace(-1,1,res)^3 ('ij', x slowest) -> [res,res,res] on device."""
    dev = planes.device
    axis = torch.linspace(-1, 1, res).to(dev)          # same values the reference builds on the host (:79-81)
    vol = torch.empty((res, res, res), dtype=torch.float32, device=dev)
    w = decoder.net.weights_c()
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().ishap_triplane_decode_grid(planes.data_ptr(), planes.shape[1], C.byref(w), axis.data_ptr(),
                                                         res, vol.data_ptr(), _lib.stream_ptr(dev)))
    return vol


def decode_volume(decoder: MultiTriplane, latent: torch.Tensor, rng, mid, res: int) -> torch.Tensor:
    """get_mesh's decode half (drag_utils.py:295-298 + visualize.py:79-97) without leaving the device."""
    return decode_planes_grid(decoder, prepare_planes(latent, rng, mid), res)
