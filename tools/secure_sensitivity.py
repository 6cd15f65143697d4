"""Does the encrypted forward see its input?  (GPU box)  The reference shares at precision_fractional = 16 in a 2^64 ring
(inference.py:280): the product of two encoded values is scaled 10^32 and wraps — SURVEY.md §8c quirk (a).  This prints, for
precision_fractional 3 / 6 / 10 / 16, the decoded logits of three different images through the mini network of
tests/test_gpu_secure.py and the magnitude of the stem's intermediate values against the plaintext convolution.
Measured (round 6): at 3 and 6 the logits track the plaintext model and differ per image; at 10 conv1's outputs are 10x too
small; at 16 they are <= 1.8e-13 and the logits of EVERY image equal the fc bias exactly.  The product is bit-identical to the
reference's own code at both 3 and 16 (tests/test_gpu_secure_ref.py) — the degenerate result is the reference's arithmetic,
reproduced, not fixed.   python tools/secure_sensitivity.py"""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from tests.test_gpu_secure import mini_state_dict
from primia_amd.secure import Dealer, SecureContext, SecureResNet18
cuda = torch.device("cuda:0")
gen = torch.Generator().manual_seed(21)
sd = mini_state_dict(gen)
blocks = [("layer1.0", 1), ("layer2.0", 2)]
imgs = [torch.randn(1, 3, 16, 16, generator=gen).to(cuda) for _ in range(3)]
for pf in (3, 6, 10, 16):
    ctx = SecureContext(Dealer(cuda, seed=5), 10, pf)
    m = SecureResNet18(ctx, sd, 16, blocks)
    print("pf", pf, [[round(v, 5) for v in m(im).flatten().tolist()] for im in imgs], "fc bias", [round(v, 5) for v in sd["fc.bias"].tolist()])
    # intermediate: the stem's output
    c = ctx
    xs = c.share(c.encode(imgs[0]), owner=1)
    m._inv = m.precompute_inv()
    x = c.conv2d(xs, m.p["conv1.weight"], 2, 3)
    v = c.decode(c.reconstruct(x)).flatten()
    print("   conv1 out: mean |v|", float(v.abs().mean()), "max", float(v.abs().max()))
    x = m._bn(x, "bn1"); v = c.decode(c.reconstruct(x)).flatten(); print("   bn1 out: mean |v|", float(v.abs().mean()), "max", float(v.abs().max()))
    x = c.relu(c.max_pool2d_3x3s2(x)); v = c.decode(c.reconstruct(x)).flatten(); print("   stem out: mean |v|", float(v.abs().mean()), "nonzero frac", float((v != 0).float().mean()))
# plaintext
from oracle import train_oracle as O
full = dict(sd)
print("plaintext conv1 mean|v|", float(torch.nn.functional.conv2d(imgs[0].cpu(), sd["conv1.weight"], None, 2, 3).abs().mean()))
