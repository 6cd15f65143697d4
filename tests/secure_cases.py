"""Test infrastructure: the per-operation cases of tests/golden/secure_ref_ops.npz and the full-forward case of
secure_ref_forward.npz — seeded plaintext inputs and the call sequence, written once for any context with the
OracleContext / SecureContext interface (share, fpt_mul, fpt_matmul, relu, reciprocal_newton, batch_norm_eval,
conv2d, max_pool2d_3x3s2, avg_pool2d, linear).  The minting script runs the REFERENCE on the same inputs."""
from collections import OrderedDict

import numpy as np

# name -> (input shapes)
CONV_SHAPES = {
    "conv_stem": (((1, 3, 16, 16), (8, 3, 7, 7)), 2, 3),
    "conv_3x3": (((1, 4, 9, 9), (6, 4, 3, 3)), 1, 1),
    "conv_s2": (((1, 4, 8, 8), (5, 4, 3, 3)), 2, 1),
    "conv_ds": (((1, 4, 8, 8), (8, 4, 1, 1)), 2, 0),
}
SHAPES = {
    "mul": ((5, 7), (5, 7)),
    "mul_bcast": ((4,), (6, 4)),
    "matmul": ((1, 9, 12), (12, 5)),
    "relu": ((1, 3, 5, 6),),
    "newton": ((4,),),
    "bn_eval": ((1, 4, 5, 5), (4,), (4,), (4,), (4,)),
    "maxpool": ((1, 2, 8, 8),),
    "avgpool": ((1, 4, 7, 7),),
    "linear": ((1, 16), (3, 16), (3,)),
}
SHAPES.update({k: v[0] for k, v in CONV_SHAPES.items()})
CASES = list(SHAPES)


def make_inputs(name):
    rng = np.random.RandomState(sum(map(ord, name)))
    xs = [rng.standard_normal(s).astype(np.float32) for s in SHAPES[name]]
    if name == "newton":
        xs[0] = (rng.rand(4) * 1.5 + 0.5).astype(np.float32)
    if name == "bn_eval":
        xs[2] = (rng.rand(4) * 1.5 + 0.5).astype(np.float32)   # running_var
    return xs


def run_case(ctx, name, inputs, encode):
    """Share the inputs in order (each consumes one mask), then the operation; returns the output shares."""
    xs = [ctx.share(encode(x, ctx.base, ctx.pf)) for x in inputs]
    if name in ("mul", "mul_bcast"):
        return ctx.fpt_mul(xs[0], xs[1])
    if name == "matmul":
        return ctx.fpt_matmul(xs[0], xs[1])
    if name == "relu":
        return ctx.relu(xs[0])
    if name == "newton":
        return ctx.reciprocal_newton(xs[0])
    if name == "bn_eval":
        return ctx.batch_norm_eval(xs[0], xs[1], xs[2], xs[3], xs[4])
    if name.startswith("conv_"):
        _, stride, pad = CONV_SHAPES[name]
        return ctx.conv2d(xs[0], xs[1], stride, pad)
    if name == "maxpool":
        return ctx.max_pool2d_3x3s2(xs[0])
    if name == "avgpool":
        return ctx.avg_pool2d(xs[0], 7)
    if name == "linear":
        return ctx.linear(xs[0], xs[1], xs[2])
    raise KeyError(name)


FWD_SIZE = 32


def forward_model_and_image():
    """Full-width ResNet-18 state dict (the reference constructor's own initialisation, reproduced bit for bit by
    resnet_spec.init_state_dict — pinned in tests/golden/train_*.npz) with non-trivial BatchNorm statistics, and
    one 32x32 image."""
    import torch

    from primia_amd import resnet_spec as rs

    torch.manual_seed(42)
    sd = rs.init_state_dict(rs.resnet18_spec(3, 3, FWD_SIZE, "max"))
    g = torch.Generator().manual_seed(1)
    out = OrderedDict()
    for k, v in sd.items():
        if k.endswith("running_mean"):
            v = torch.randn(v.shape, generator=g) * 0.1
        elif k.endswith("running_var"):
            v = torch.rand(v.shape, generator=g) * 1.5 + 0.5
        elif ("bn" in k or "downsample.1" in k) and k.endswith("weight"):
            v = torch.rand(v.shape, generator=g) + 0.5
        elif ("bn" in k or "downsample.1" in k) and k.endswith("bias"):
            v = torch.randn(v.shape, generator=g) * 0.1
        out[k] = v.numpy().copy()
    image = torch.randn(1, 3, FWD_SIZE, FWD_SIZE, generator=g).numpy()
    return out, image
