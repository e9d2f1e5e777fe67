"""Host-side mirror of guided_diffusion/unet.py:UNetModel backed by libishap_hip.so.

Same call surface as the reference module for the path the editor uses
(`model(x, timesteps, feat_layer=k)` -> `(out, inter_feat)` or `out`, `load_state_dict(strict=True)`,
`convert_to_fp16()`, `eval()`, `parameters()`), but the network itself is one C call that
enqueues the hand-written HIP kernels on torch's current stream.  torch only owns the tensors.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Iterator, Optional

import numpy as np
import torch

from . import _lib
from .unet_spec import UNetConfig, build_spec, param_shapes


class UNetModel:
    def __init__(self, cfg: UNetConfig, device: Optional[torch.device] = None, max_batch: int = 1):
        if not cfg.use_fp16:
            raise NotImplementedError("the path's precision contract is use_fp16=True (drag_utils.py:51, "
                                      "generate.py:67): fp16 torso, fp32 norms/embeddings/head")
        self.cfg = cfg
        self.spec = build_spec(cfg)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("UNetModel needs a GPU device: there is no CPU fallback")
        self.image_size = cfg.image_size
        self.in_channels = cfg.in_channels
        self.out_channels = cfg.out_channels
        self.model_channels = cfg.model_channels
        self.dtype = torch.float16
        self.max_batch = max_batch
        L = _lib.lib()
        c = _lib.UNetConfigC()
        c.image_size, c.in_channels, c.model_channels = cfg.image_size, cfg.in_channels, cfg.model_channels
        c.out_channels, c.num_res_blocks = cfg.out_channels, cfg.num_res_blocks
        mult = cfg.resolved_channel_mult()
        if any(int(m) != m for m in mult):
            raise NotImplementedError("fractional channel_mult (image_size 512) is not on the path")
        c.n_mult = len(mult)
        for i, m in enumerate(mult):
            c.channel_mult[i] = int(m)
        att = cfg.attention_ds()
        c.n_att = len(att)
        for i, a in enumerate(att):
            c.attention_ds[i] = int(a)
        c.num_head_channels = cfg.num_head_channels
        c.max_batch = max_batch
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(L.ishap_unet_create(C.byref(c), self.device.index or 0, C.byref(h)))
        self._h = h
        self._L = L
        self._param_probe = torch.zeros(1, dtype=torch.float32, device=self.device)
        self._fp16 = False
        self._last_shape = None

    # ------------------------------------------------------------------ module-like surface
    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._L.ishap_unet_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def to(self, device):
        if torch.device(device).type != "cuda":
            raise RuntimeError("UNetModel lives on the GPU it was created on")
        return self

    def eval(self):
        return self

    def parameters(self) -> Iterator[torch.Tensor]:
        # drag_utils.py:257-264 only asks the first parameter for its dtype (time_embed.0.weight: fp32)
        yield self._param_probe

    def convert_to_fp16(self):
        """unet.py:618-624 -- the packed weights already follow that contract (done at load)."""
        self._fp16 = True

    def param_table(self) -> Dict[str, tuple]:
        L, out = self._L, {}
        name = C.create_string_buffer(256)
        nd = C.c_int()
        shp = (C.c_longlong * 4)()
        for i in range(L.ishap_unet_num_params(self._h)):
            _lib.check(L.ishap_unet_param_info(self._h, i, name, 256, C.byref(nd), shp))
            out[name.value.decode()] = tuple(int(shp[k]) for k in range(nd.value))
        return out

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        table = param_shapes(self.cfg)
        missing = [k for k in table if k not in sd]
        unexpected = [k for k in sd if k not in table]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict: missing keys {missing[:5]}..., "
                               f"unexpected keys {unexpected[:5]}...")
        s = _lib.stream_ptr(self.device)
        with torch.cuda.device(self.device):
            for k, shape in table.items():
                if k not in sd:
                    continue
                v = sd[k]
                if tuple(v.shape) != tuple(shape):
                    raise RuntimeError(f"size mismatch for {k}: {tuple(v.shape)} vs {tuple(shape)}")
                v = v.detach().to(device=self.device, dtype=torch.float32).contiguous()
                _lib.check(self._L.ishap_unet_load_param(self._h, k.encode(), v.data_ptr(), v.numel(), s))
                del v
            torch.cuda.current_stream(self.device).synchronize()
        return self

    # ------------------------------------------------------------------ forward / backward
    def forward(self, x: torch.Tensor, timesteps, y=None, feat_layer: int = -1, keep_for_backward: bool = False,
                want_inter_feat: bool = True, overlap_tail: bool = False):
        """`overlap_tail` (with a tap): the blocks after the tap and the head are only PLANNED; `run_tail()` enqueues them on the
        context's own stream beside whatever the caller has enqueued meanwhile (loss, backward) -- a join, the next forward or a
        full-depth backward does so by itself.  The returned model output is valid only after `join_tail()`; the model keeps a
        reference to it until then (the library writes it through the raw pointer)."""
        assert y is None, "class conditioning is not on the path"
        assert x.dim() == 4 and x.shape[1] == self.in_channels and x.shape[2] == x.shape[3] == self.image_size
        N = x.shape[0]
        x = x.detach().to(device=self.device, dtype=torch.float32).contiguous()
        ts = (C.c_float * N)(*[float(t) for t in (timesteps.tolist() if torch.is_tensor(timesteps) else timesteps)])
        out = torch.empty((N, self.out_channels, self.image_size, self.image_size), dtype=torch.float32,
                          device=self.device)
        inter = None
        if feat_layer >= 0 and want_inter_feat:
            ch, sz = C.c_int(), C.c_int()
            _lib.check(self._L.ishap_unet_tap_shape(self._h, feat_layer, C.byref(ch), C.byref(sz)))
            inter = torch.empty((N, ch.value, sz.value, sz.value), dtype=torch.float16, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._L.ishap_unet_forward(self._h, x.data_ptr(), ts, N, int(feat_layer), out.data_ptr(),
                                                  _lib.ptr(inter), int(bool(keep_for_backward)) | (2 if overlap_tail else 0),
                                                  _lib.stream_ptr(self.device)))
        self._last_shape = tuple(x.shape)
        self._tail_out = out if overlap_tail else None       # alive until the tail that writes it has been joined
        if feat_layer < 0:
            return out
        return out, inter

    __call__ = forward

    def run_tail(self):
        """Enqueue the forward tail that `forward(overlap_tail=True)` only planned; no-op when there is none."""
        with torch.cuda.device(self.device):
            _lib.check(self._L.ishap_unet_run_tail(self._h))

    def join_tail(self):
        """Order the current stream behind an overlapped forward tail (no-op when there is none)."""
        with torch.cuda.device(self.device):
            _lib.check(self._L.ishap_unet_join_tail(self._h, _lib.stream_ptr(self.device)))
        self._tail_out = None

    def prepare_timesteps(self, timesteps):
        """Compute the timestep-dependent FiLM rows of a sampling loop once (they do not depend on x): forwards at these
        timesteps skip the embedding launches.  An empty list drops them.  Values are bit-identical either way."""
        ts = [float(t) for t in timesteps]
        arr = (C.c_float * max(len(ts), 1))(*ts)
        with torch.cuda.device(self.device):
            _lib.check(self._L.ishap_unet_prepare_timesteps(self._h, arr, len(ts), _lib.stream_ptr(self.device)))

    def tap_shape(self, feat_layer: int):
        ch, sz = C.c_int(), C.c_int()
        _lib.check(self._L.ishap_unet_tap_shape(self._h, feat_layer, C.byref(ch), C.byref(sz)))
        return ch.value, sz.value

    def tap_ptr(self) -> int:
        return self._L.ishap_unet_tap_ptr(self._h)

    def copy_tap(self, feat_layer: int, N: Optional[int] = None) -> torch.Tensor:
        """The resident tap of the last forward as an NHWC fp16 tensor [N, S*S, C] (device copy); N is the last forward's
        batch size (the library copies that many images)."""
        ch, sz = self.tap_shape(feat_layer)
        last = self._last_shape[0] if self._last_shape is not None else 1
        if N is not None and N != last:
            raise ValueError(f"copy_tap: the resident tap holds {last} image(s), not {N}")
        N = last
        t = torch.empty((N, sz * sz, ch), dtype=torch.float16, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._L.ishap_unet_copy_tap(self._h, t.data_ptr(), _lib.stream_ptr(self.device)))
        return t

    def block_output(self, group: int, index: int = 0) -> torch.Tensor:
        """Output of input_blocks[index] (group 0), middle_block (1) or output_blocks[index] (2) of the last
        forward(keep_for_backward=True), as fp16 NCHW like the reference's activations.  The batch size is the kept
        forward's (the library writes that many images)."""
        ch, sz = C.c_int(), C.c_int()
        _lib.check(self._L.ishap_unet_block_output(self._h, group, index, C.byref(ch), C.byref(sz), None, None))
        if self._last_shape is None:
            raise RuntimeError("block outputs stay resident only after a forward with keep_for_backward=1")
        t = torch.empty((self._last_shape[0], ch.value, sz.value, sz.value), dtype=torch.float16, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._L.ishap_unet_block_output(self._h, group, index, C.byref(ch), C.byref(sz), t.data_ptr(),
                                                       _lib.stream_ptr(self.device)))
        return t

    def workspace_bytes(self) -> int:
        return int(self._L.ishap_unet_workspace_bytes(self._h))

    def backward_input(self, cot_nhwc_f16: torch.Tensor, scale2: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Gradient w.r.t. x of sum(tap * cot) for the last forward(keep_for_backward=True)."""
        dx = torch.empty(self._last_shape, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._L.ishap_unet_backward_input(self._h, cot_nhwc_f16.data_ptr(), _lib.ptr(scale2),
                                                         dx.data_ptr(), _lib.stream_ptr(self.device)))
        return dx

    def backward_from_output(self, cot_out: torch.Tensor, scale2: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Full-depth input gradient of sum(out * cot).  `cot_out` fp32, or fp16 pre-multiplied by scale2[0]."""
        dx = torch.empty(self._last_shape, dtype=torch.float32, device=self.device)
        is_f16 = cot_out.dtype == torch.float16
        cot_out = cot_out.detach().to(device=self.device).contiguous()
        if not is_f16:
            cot_out = cot_out.float()
        with torch.cuda.device(self.device):
            _lib.check(self._L.ishap_unet_backward_from_output(self._h, cot_out.data_ptr(), int(is_f16), _lib.ptr(scale2),
                                                               dx.data_ptr(), _lib.stream_ptr(self.device)))
        return dx
