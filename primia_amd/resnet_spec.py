"""ResNet-18 structure as PriMIA builds it (torchlib/models.py:345-423, 499-516 with the arguments
of train.py:257-266): layer table, state-dict key order and the reference's initialisation.

This module holds no arithmetic of the network itself — the forward/backward passes are HIP
kernels driven by primia_amd.engine.  What lives here is host logic: which tensors exist, in which
order the reference's `state_dict()` lists them, and how the reference's constructor consumes the
torch RNG, so that `torch.manual_seed(s)` followed by `init_state_dict(...)` reproduces the same
initial weights as `torch.manual_seed(s); resnet18(...)` in the reference.
"""
import math
from collections import OrderedDict
from dataclasses import dataclass
from typing import List, Optional

import torch


@dataclass
class ConvSpec:
    name: str          # state-dict prefix, e.g. "layer1.0.conv1"
    cin: int
    cout: int
    k: int
    stride: int
    pad: int


@dataclass
class BlockSpec:
    prefix: str        # "layer1.0"
    conv1: ConvSpec
    conv2: ConvSpec
    down: Optional[ConvSpec]   # 1x1 stride-s projection ("downsample.0"), BN is "downsample.1"


@dataclass
class NetSpec:
    in_channels: int
    num_classes: int
    input_size: int
    pooling: str
    stem: ConvSpec
    blocks: List[BlockSpec]

    @property
    def convs(self) -> List[ConvSpec]:
        out = [self.stem]
        for b in self.blocks:
            out += [b.conv1, b.conv2]
            if b.down is not None:
                out.append(b.down)
        return out


def resnet18_spec(num_classes=3, in_channels=3, input_size=224, pooling="max") -> NetSpec:
    if pooling not in ("max", "avg"):
        raise NotImplementedError("pooling type unknown: {:s}".format(str(pooling)))
    stem = ConvSpec("conv1", in_channels, 64, 7, 2, 3)
    blocks = []
    inplanes = 64
    for li, (planes, stride) in enumerate([(64, 1), (128, 2), (256, 2), (512, 2)], start=1):
        for bi in range(2):
            s = stride if bi == 0 else 1
            prefix = f"layer{li}.{bi}"
            down = None
            if bi == 0 and (s != 1 or inplanes != planes):
                down = ConvSpec(prefix + ".downsample.0", inplanes, planes, 1, s, 0)
            blocks.append(
                BlockSpec(
                    prefix,
                    ConvSpec(prefix + ".conv1", inplanes, planes, 3, s, 1),
                    ConvSpec(prefix + ".conv2", planes, planes, 3, 1, 1),
                    down,
                )
            )
            inplanes = planes
    return NetSpec(in_channels, num_classes, input_size, pooling, stem, blocks)


def bn_name(conv_name: str) -> str:
    """BatchNorm that follows a conv: conv1->bn1, conv2->bn2, downsample.0->downsample.1."""
    if conv_name.endswith("downsample.0"):
        return conv_name[:-1] + "1"
    return conv_name.replace("conv", "bn")


def param_entries(spec: NetSpec):
    """(key, shape) of every parameter in reference `named_parameters()` order."""
    out = []

    def conv_bn(c: ConvSpec):
        out.append((c.name + ".weight", (c.cout, c.cin, c.k, c.k)))
        b = bn_name(c.name)
        out.append((b + ".weight", (c.cout,)))
        out.append((b + ".bias", (c.cout,)))

    conv_bn(spec.stem)
    for blk in spec.blocks:
        conv_bn(blk.conv1)
        conv_bn(blk.conv2)
        if blk.down is not None:
            conv_bn(blk.down)
    out.append(("fc.weight", (spec.num_classes, 512)))
    out.append(("fc.bias", (spec.num_classes,)))
    return out


def buffer_entries(spec: NetSpec, norm: str = "batch"):
    """(key, shape) of the float buffers (running_mean, running_var) in reference order.
    GroupNorm (norm="group", the DP configuration) has none."""
    out = []
    if norm == "group":
        return out
    for c in spec.convs:
        b = bn_name(c.name)
        out.append((b + ".running_mean", (c.cout,)))
        out.append((b + ".running_var", (c.cout,)))
    return out


def state_dict_keys(spec: NetSpec, norm: str = "batch"):
    """Key order of the reference model's state_dict() (122 keys for ResNet-18 with BatchNorm; with
    `norm_layer` = GroupNorm the running statistics and counters do not exist)."""
    keys = []

    def conv_bn(c: ConvSpec):
        b = bn_name(c.name)
        keys.extend([c.name + ".weight", b + ".weight", b + ".bias"])
        if norm == "batch":
            keys.extend([b + ".running_mean", b + ".running_var", b + ".num_batches_tracked"])

    conv_bn(spec.stem)
    for blk in spec.blocks:
        conv_bn(blk.conv1)
        conv_bn(blk.conv2)
        if blk.down is not None:
            conv_bn(blk.down)
    keys += ["fc.weight", "fc.bias"]
    return keys


def _default_conv_init_(w):
    # nn.Conv2d.reset_parameters (torch): kaiming_uniform_(a=sqrt(5)); consumed, then overwritten.
    torch.nn.init.kaiming_uniform_(w, a=math.sqrt(5))


def _linear_init(out_f, in_f):
    w = torch.empty(out_f, in_f)
    torch.nn.init.kaiming_uniform_(w, a=math.sqrt(5))
    bound = 1 / math.sqrt(in_f)
    b = torch.empty(out_f)
    torch.nn.init.uniform_(b, -bound, bound)
    return w, b


def init_state_dict(spec: NetSpec, norm: str = "batch") -> "OrderedDict[str, torch.Tensor]":
    """Fresh CPU state dict drawn from the global torch RNG exactly as the reference constructor
    does (torchlib/models.py:379-413, 495): default Conv2d/Linear initialisers run first in
    construction order, then every conv is re-drawn with kaiming_normal_(fan_out, relu), BN is
    (1, 0), and the 1000-way fc is replaced by a freshly drawn num_classes-way one."""
    weights = {}
    # construction order: stem; per layer: downsample conv BEFORE the block's own convs
    _default_conv_init_(weights.setdefault(spec.stem.name, torch.empty(spec.stem.cout, spec.stem.cin, 7, 7)))
    for blk in spec.blocks:
        order = ([blk.down] if blk.down is not None else []) + [blk.conv1, blk.conv2]
        for c in order:
            w = torch.empty(c.cout, c.cin, c.k, c.k)
            _default_conv_init_(w)
            weights[c.name] = w
    _linear_init(1000, 512)  # self.fc = nn.Linear(512, 1000): RNG consumed, result discarded
    # modules() order: stem, then per block conv1, conv2, downsample.0
    for c in spec.convs:
        torch.nn.init.kaiming_normal_(weights[c.name], mode="fan_out", nonlinearity="relu")
    fc_w, fc_b = _linear_init(spec.num_classes, 512)

    sd = OrderedDict()
    for key in state_dict_keys(spec, norm):
        mod, leaf = key.rsplit(".", 1)
        if key == "fc.weight":
            sd[key] = fc_w
        elif key == "fc.bias":
            sd[key] = fc_b
        elif mod in weights:
            sd[key] = weights[mod]
        elif leaf in ("weight", "running_var"):
            c = _bn_channels(spec, mod)
            sd[key] = torch.ones(c)
        elif leaf in ("bias", "running_mean"):
            c = _bn_channels(spec, mod)
            sd[key] = torch.zeros(c)
        elif leaf == "num_batches_tracked":
            sd[key] = torch.tensor(0, dtype=torch.long)
        else:  # pragma: no cover
            raise KeyError(key)
    return sd


def _bn_channels(spec: NetSpec, bn: str) -> int:
    for c in spec.convs:
        if bn_name(c.name) == bn:
            return c.cout
    raise KeyError(bn)
