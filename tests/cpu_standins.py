"""Test-side CPU stand-ins for the HIP operations the multi-rank control flow calls (arena arithmetic, ChaCha20
keystream, a toy engine), so that `primia_amd.fed`'s collectives, schedules and masks run under gloo without a GPU.
Nothing here is product code; the product refuses to run without the HIP library."""
import numpy as np
import torch


class CpuArenaOps:
    """primia_amd.fed.HipArenaOps with torch-CPU arithmetic (same semantics as the kernels)."""

    def scale(self, x, a):
        x.mul_(torch.tensor(a, dtype=torch.float32))

    def divide(self, x, d):
        x.div_(torch.tensor(d, dtype=torch.float32))

    def encode(self, x, q, scale):
        q.copy_((x * torch.tensor(scale, dtype=torch.float32)).long())

    def decode(self, q, x, scale):
        x.copy_(q.float() / torch.tensor(scale, dtype=torch.float32))


def chacha20_words(key, block0, n):
    """n little-endian 64-bit keystream words from block `block0` on: original ChaCha20 layout (64-bit counter in
    words 12-13, 64-bit nonce in 14-15), key = (k0, k1, k2, k3, nonce) as 64-bit integers — csrc/chacha.hip."""
    nb = (n + 7) // 8
    kw = []
    for k in key[:4]:
        kw += [k & 0xFFFFFFFF, (k >> 32) & 0xFFFFFFFF]
    ctr = np.uint64(block0) + np.arange(nb, dtype=np.uint64)
    s = np.empty((16, nb), dtype=np.uint32)
    for i, c in enumerate((0x61707865, 0x3320646e, 0x79622d32, 0x6b206574)):
        s[i] = c
    for i in range(8):
        s[4 + i] = kw[i]
    s[12] = (ctr & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    s[13] = (ctr >> np.uint64(32)).astype(np.uint32)
    s[14] = key[4] & 0xFFFFFFFF
    s[15] = (key[4] >> 32) & 0xFFFFFFFF
    x = s.copy()

    def rotl(v, c):
        return (v << np.uint32(c)) | (v >> np.uint32(32 - c))

    def qr(a, b, c, d):
        x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 16)
        x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 12)
        x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 8)
        x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 7)

    with np.errstate(over="ignore"):
        for _ in range(10):
            qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
            qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
        x += s
    words = x.T.astype(np.uint64)                       # [nb, 16] 32-bit words of each block
    out = words[:, 0::2] | (words[:, 1::2] << np.uint64(32))
    return out.reshape(-1)[:n]


class CpuMaskOps:
    """primia_amd.fed.HipMaskOps on the CPU."""

    def keystream(self, key, block0, out):
        out.copy_(torch.from_numpy(chacha20_words(key, block0, out.numel()).view(np.int64)))

    def ring_accumulate(self, q, m, subtract):
        if subtract:
            q.sub_(m)
        else:
            q.add_(m)


class ToyEngine:
    """What fed.federated_epoch touches of an engine, on the CPU: a flat fp32 arena, forward / loss_backward / an SGD
    step (a least-squares model over the arena's first `dim` entries), BatchNorm counters, refresh_weights."""

    def __init__(self, n_words=257, dim=16, seed=0):
        g = torch.Generator().manual_seed(seed)
        self.flat = torch.randn(n_words, generator=g) * 0.1
        self.grads = torch.zeros(n_words)
        self.dim = dim
        self.p_entries = [("w", (n_words,))]
        self.num_batches_tracked = {"bn1": 0}
        self.opt_state, self.opt_steps = None, 0
        self.refreshes = 0
        self._x = None

    def reset_optimizer(self):
        self.opt_state, self.opt_steps = None, 0

    def forward(self, data):
        self._x = data
        self.num_batches_tracked["bn1"] += 1
        return data @ self.flat[:self.dim]

    def loss_backward(self, target, soft=False):
        pred = self._x @ self.flat[:self.dim]
        err = pred - target
        self.grads.zero_()
        self.grads[:self.dim] = 2.0 * (self._x.t() @ err) / err.numel()
        return (err * err).mean()

    def sgd_step(self, lr, weight_decay):
        self.flat -= torch.tensor(lr, dtype=torch.float32) * (self.grads + torch.tensor(weight_decay, dtype=torch.float32) * self.flat)
        self.opt_steps += 1

    def refresh_weights(self):
        self.refreshes += 1
