"""Structure of the triplane-diffusion UNet as a plain data description.

This is the host-side mirror of the reference's module tree
(neural_field_diffusion/guided_diffusion/unet.py:427-616, built by
script_util.py:132-187).  It yields
  * the block graph (which ResBlock / AttentionBlock sits where, channel
    counts, up/down flags), and
  * the exact state_dict key names and shapes a reference checkpoint holds
    (drag_utils.py:229-230 loads with strict=True).

It carries no arithmetic: the HIP library rebuilds the same graph from the
same config (csrc/unet.hip) and tests compare the two parameter tables.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple


@dataclass
class UNetConfig:
    image_size: int = 128
    in_channels: int = 96
    model_channels: int = 256
    out_channels: int = 192            # in_out_channels * 2 when learn_sigma
    num_res_blocks: int = 2
    attention_resolutions: str = "32,16,8"
    channel_mult: Tuple[int, ...] = ()
    num_head_channels: int = 64
    num_heads: int = 4                 # only used when num_head_channels == -1
    use_scale_shift_norm: bool = True
    resblock_updown: bool = True
    use_fp16: bool = True

    def resolved_channel_mult(self) -> Tuple[int, ...]:
        # script_util.py:151-161
        if self.channel_mult:
            return tuple(self.channel_mult)
        table = {512: (0.5, 1, 1, 2, 2, 4, 4), 256: (1, 1, 2, 2, 4, 4),
                 128: (1, 1, 2, 3, 4), 64: (1, 2, 3, 4)}
        if self.image_size not in table:
            raise ValueError(f"unsupported image size: {self.image_size}")
        return table[self.image_size]

    def attention_ds(self) -> Tuple[int, ...]:
        # script_util.py:163-165
        return tuple(self.image_size // int(r) for r in self.attention_resolutions.split(","))


def full_config() -> UNetConfig:
    """drag_utils.py:44-57 / generate.py:64-71."""
    return UNetConfig()


def tiny_config(num_res_blocks: int = 1) -> UNetConfig:
    """Small net with every block kind (plain/up/down ResBlock, attention,
    skip 1x1, concat); used by the golden fixtures."""
    return UNetConfig(image_size=16, in_channels=6, model_channels=32, out_channels=12,
                      num_res_blocks=num_res_blocks, attention_resolutions="8",
                      channel_mult=(1, 2), num_head_channels=32)


@dataclass
class ResSpec:
    path: str
    cin: int
    cout: int
    up: bool = False
    down: bool = False
    kind: str = "res"


@dataclass
class AttnSpec:
    path: str
    channels: int
    heads: int
    kind: str = "attn"


@dataclass
class ConvSpec:
    path: str
    cin: int
    cout: int
    kind: str = "conv"


@dataclass
class BlockSpec:
    """One TimestepEmbedSequential (unet.py:66-78)."""
    name: str
    layers: list
    cin: int = 0       # channels entering the block (after concat for output blocks)
    cout: int = 0
    res_in: int = 0    # spatial size entering
    res_out: int = 0
    skip_ch: int = 0   # channels of the popped skip tensor (output blocks)


@dataclass
class UNetSpec:
    cfg: UNetConfig
    input_blocks: List[BlockSpec] = field(default_factory=list)
    middle_block: Optional[BlockSpec] = None
    output_blocks: List[BlockSpec] = field(default_factory=list)
    time_embed_dim: int = 0
    final_ch: int = 0

    def all_blocks(self) -> List[BlockSpec]:
        return self.input_blocks + [self.middle_block] + self.output_blocks


def _heads(cfg: UNetConfig, ch: int) -> int:
    if cfg.num_head_channels == -1:
        return cfg.num_heads
    assert ch % cfg.num_head_channels == 0
    return ch // cfg.num_head_channels


def build_spec(cfg: UNetConfig) -> UNetSpec:
    """Walk the constructor logic of unet.py:470-616 without building modules."""
    if not (cfg.use_scale_shift_norm and cfg.resblock_updown):
        raise NotImplementedError("only the scale-shift / resblock_updown variant is on the path "
                                  "(drag_utils.py:51)")
    mc = cfg.model_channels
    mult = cfg.resolved_channel_mult()
    att = cfg.attention_ds()
    spec = UNetSpec(cfg=cfg, time_embed_dim=mc * 4)
    ch = int(mult[0] * mc)
    res = cfg.image_size
    spec.input_blocks.append(BlockSpec("input_blocks.0", [ConvSpec("input_blocks.0.0", cfg.in_channels, ch)],
                                       cfg.in_channels, ch, res, res))
    chans = [ch]
    ds = 1
    for level, m in enumerate(mult):
        for _ in range(cfg.num_res_blocks):
            idx = len(spec.input_blocks)
            name = f"input_blocks.{idx}"
            cout = int(m * mc)
            layers = [ResSpec(f"{name}.0", ch, cout)]
            cin = ch
            ch = cout
            if ds in att:
                layers.append(AttnSpec(f"{name}.1", ch, _heads(cfg, ch)))
            spec.input_blocks.append(BlockSpec(name, layers, cin, ch, res, res))
            chans.append(ch)
        if level != len(mult) - 1:
            idx = len(spec.input_blocks)
            name = f"input_blocks.{idx}"
            spec.input_blocks.append(BlockSpec(name, [ResSpec(f"{name}.0", ch, ch, down=True)],
                                               ch, ch, res, res // 2))
            chans.append(ch)
            ds *= 2
            res //= 2
    spec.middle_block = BlockSpec("middle_block", [
        ResSpec("middle_block.0", ch, ch),
        AttnSpec("middle_block.1", ch, _heads(cfg, ch)),
        ResSpec("middle_block.2", ch, ch)], ch, ch, res, res)
    for level, m in list(enumerate(mult))[::-1]:
        for i in range(cfg.num_res_blocks + 1):
            ich = chans.pop()
            idx = len(spec.output_blocks)
            name = f"output_blocks.{idx}"
            cout = int(mc * m)
            layers = [ResSpec(f"{name}.0", ch + ich, cout)]
            cin = ch + ich
            ch = cout
            res_in = res
            if ds in att:
                layers.append(AttnSpec(f"{name}.{len(layers)}", ch, _heads(cfg, ch)))
            if level and i == cfg.num_res_blocks:
                layers.append(ResSpec(f"{name}.{len(layers)}", ch, ch, up=True))
                ds //= 2
                res *= 2
            spec.output_blocks.append(BlockSpec(name, layers, cin, ch, res_in, res, skip_ch=ich))
    spec.final_ch = ch
    return spec


def param_shapes(cfg: UNetConfig) -> Dict[str, Tuple[int, ...]]:
    """state_dict keys -> shapes, as torch names them for the reference model."""
    spec = build_spec(cfg)
    mc = cfg.model_channels
    ted = spec.time_embed_dim
    out: Dict[str, Tuple[int, ...]] = {
        "time_embed.0.weight": (ted, mc), "time_embed.0.bias": (ted,),
        "time_embed.2.weight": (ted, ted), "time_embed.2.bias": (ted,),
    }

    def add_res(r: ResSpec):
        p = r.path
        out[f"{p}.in_layers.0.weight"] = (r.cin,)
        out[f"{p}.in_layers.0.bias"] = (r.cin,)
        out[f"{p}.in_layers.2.weight"] = (r.cout, r.cin, 3, 3)
        out[f"{p}.in_layers.2.bias"] = (r.cout,)
        out[f"{p}.emb_layers.1.weight"] = (2 * r.cout, ted)
        out[f"{p}.emb_layers.1.bias"] = (2 * r.cout,)
        out[f"{p}.out_layers.0.weight"] = (r.cout,)
        out[f"{p}.out_layers.0.bias"] = (r.cout,)
        out[f"{p}.out_layers.3.weight"] = (r.cout, r.cout, 3, 3)
        out[f"{p}.out_layers.3.bias"] = (r.cout,)
        if r.cin != r.cout:
            out[f"{p}.skip_connection.weight"] = (r.cout, r.cin, 1, 1)
            out[f"{p}.skip_connection.bias"] = (r.cout,)

    def add_attn(a: AttnSpec):
        p = a.path
        out[f"{p}.norm.weight"] = (a.channels,)
        out[f"{p}.norm.bias"] = (a.channels,)
        out[f"{p}.qkv.weight"] = (3 * a.channels, a.channels, 1)
        out[f"{p}.qkv.bias"] = (3 * a.channels,)
        out[f"{p}.proj_out.weight"] = (a.channels, a.channels, 1)
        out[f"{p}.proj_out.bias"] = (a.channels,)

    for b in spec.all_blocks():
        for l in b.layers:
            if l.kind == "conv":
                out[f"{l.path}.weight"] = (l.cout, l.cin, 3, 3)
                out[f"{l.path}.bias"] = (l.cout,)
            elif l.kind == "res":
                add_res(l)
            else:
                add_attn(l)
    out["out.0.weight"] = (spec.final_ch,)
    out["out.0.bias"] = (spec.final_ch,)
    out["out.2.weight"] = (cfg.out_channels, spec.final_ch, 3, 3)
    out["out.2.bias"] = (cfg.out_channels,)
    return out


def is_torso_conv(name: str) -> bool:
    """True for the tensors `convert_to_fp16` halves (unet.py:618-624,
    fp16_util.py:14-21): conv weight/bias inside input/middle/output blocks."""
    if name.startswith(("time_embed", "out.")):
        return False
    # conv leaves: stem "<block>.0.{weight,bias}", ResBlock convs, skip, qkv, proj_out
    if ".emb_layers." in name:
        return False
    if ".in_layers.0." in name or ".out_layers.0." in name or ".norm." in name:
        return False
    return True


DECODER_SHAPES: Dict[str, Tuple[int, ...]] = {
    # triplane_decoder/axisnetworks.py:526-535 (state_dict of `net` only, drag_utils.py:246)
    "0._B": (32, 64),
    "1.weight": (128, 128), "1.bias": (128,),
    "3.weight": (128, 128), "3.bias": (128,),
    "5.weight": (1, 128), "5.bias": (1,),
}
