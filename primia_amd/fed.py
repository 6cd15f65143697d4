"""Federated averaging across clients, one client per GPU / process.

Restates the exchange step of the reference's federated epoch —
`aggregation()` torchlib/utils.py:1000-1092, `send_new_models()` :1095-1105 and the sync schedule of
`secure_aggregation_epoch()` :1108-1233 — as collectives on the flat parameter arena:

  reference (hub and spoke, per state-dict key)         here (per client rank k, whole arena)
  ------------------------------------------------       ---------------------------------------------
  plaintext: sum_k (theta_k * w_k)  [/K if unweighted]   scale by w_k -> all_reduce(SUM) [-> /K]
  secure   : fix_prec(theta_k * w_k).share().get(),      scale by w_k -> fx_encode -> all_reduce(SUM) on
             ring sum of shares, reconstruct, float_prec int64 (wraps mod 2^64) -> fx_decode [-> /K]
             [/K if unweighted]                          (additive sharing commutes with ring addition, so
                                                         the reconstructed value is bit-identical)
  send_new_models to workers that still have batches     a rank adopts the average only if it still has
                                                         batches (mid-epoch) — always at the epoch end

`num_batches_tracked` never takes part (utils.py:1028,1040): it is not in the arena.
The arithmetic on the arena (scale / encode / decode) runs as HIP kernels (`HipArenaOps`); there is
no CPU implementation in this package.
"""
import torch
import torch.distributed as dist

from . import _lib


class HipArenaOps:
    """Element-wise arena arithmetic on the GPU through the C ABI."""

    def scale(self, x, a):
        _lib.call("primia_scale", x, x.numel(), float(a))

    def divide(self, x, d):
        _lib.call("primia_divide", x, x.numel(), float(d))

    def encode(self, x, q, scale):
        _lib.call("primia_fx_encode", x, q, x.numel(), float(scale))

    def decode(self, q, x, scale):
        _lib.call("primia_fx_decode", q, x, x.numel(), float(scale))


class HipMaskOps:
    """The two device operations of PairwiseMasks: ChaCha20 keystream and ring add / subtract (HIP kernels)."""

    def keystream(self, key, block0, out):
        _lib.call("primia_chacha20_fill", key[0], key[1], key[2], key[3], key[4], block0, out, out.numel())

    def ring_accumulate(self, q, m, subtract):
        _lib.call("primia_ring_sub" if subtract else "primia_ring_add", q, m, q, q.numel(), q.numel())


class PairwiseMasks:
    """Masks m_k with sum_k m_k = 0 (mod 2^64) for the secure all-reduce: every pair of clients (i < j) shares a
    256-bit ChaCha20 key; client i ADDS the pair's keystream to its encoded update, client j SUBTRACTS it.  What a
    client puts on the wire is then uniformly random to everybody who lacks one of its pair keys, the masks cancel in
    the ring sum, and the decoded average is bit-identical to the unmasked one — the role additive sharing between
    the workers plays in the reference (`.fix_prec().share(*workers)`, torchlib/utils.py:1046-1060), without a hub.

    Pair keys come from an X25519 key agreement (RFC 7748) run once per training (`setup`): every client draws a
    private scalar from `os.urandom`, only the PUBLIC points travel (one all_gather), and each pair derives its ChaCha20
    key and nonce as SHA-256 of the shared point — nobody who merely observes the transport (the adversary the masks are
    there for) learns a key, and key material never leaves host memory.  The exchange is not authenticated: an ACTIVE
    man in the middle of a multi-node deployment still needs the authenticated channel the reference expects of its
    websocket links.  Every sync consumes a fresh keystream segment."""

    def __init__(self, keys, rank, n_words, ops=None):
        self.keys, self.rank = keys, rank          # {peer: (k0, k1, k2, k3, nonce)}
        self.blocks_per_sync = (n_words + 7) // 8
        self.syncs = 0
        self._tmp = None
        self.ops = ops or HipMaskOps()

    @staticmethod
    def x25519(k: bytes, u: bytes) -> bytes:
        """RFC 7748 section 5: scalar multiplication on Curve25519 (Montgomery ladder, Python integers)."""
        P, A24 = 2 ** 255 - 19, 121665
        kn = int.from_bytes(k, "little")
        kn = (kn & ~7 & ~(128 << 8 * 31)) | (64 << 8 * 31)
        x1 = int.from_bytes(u, "little") & ((1 << 255) - 1)
        x2, z2, x3, z3, swap = 1, 0, x1, 1, 0
        for t in reversed(range(255)):
            kt = (kn >> t) & 1
            if swap ^ kt:
                x2, x3, z2, z3 = x3, x2, z3, z2
            swap = kt
            a, b = (x2 + z2) % P, (x2 - z2) % P
            aa, bb = a * a % P, b * b % P
            e = (aa - bb) % P
            c, d = (x3 + z3) % P, (x3 - z3) % P
            da, cb = d * a % P, c * b % P
            x3 = (da + cb) ** 2 % P
            z3 = x1 * (da - cb) ** 2 % P
            x2 = aa * bb % P
            z2 = e * (aa + A24 * e) % P
        if swap:
            x2, z2 = x3, z3
        return (x2 * pow(z2, P - 2, P) % P).to_bytes(32, "little")

    @classmethod
    def pair_key(cls, shared: bytes, i: int, j: int):
        """(k0, k1, k2, k3, nonce) of the pair i < j from the X25519 shared point."""
        import hashlib

        tag = b"primia-pairwise-mask:%d:%d:" % (i, j)
        key = hashlib.sha256(tag + b"key:" + shared).digest()
        nonce = hashlib.sha256(tag + b"nonce:" + shared).digest()[:8]
        return tuple(int.from_bytes(key[8 * t:8 * t + 8], "little") for t in range(4)) + (int.from_bytes(nonce, "little"),)

    @classmethod
    def setup(cls, n_words, device, group=None, ops=None):
        import os

        K = dist.get_world_size(group)
        rank = dist.get_rank(group)
        secret = os.urandom(32)
        public = cls.x25519(secret, (9).to_bytes(32, "little"))
        # (public points only; RCCL moves device buffers, gloo host buffers)
        mine = torch.tensor(list(public), dtype=torch.uint8, device=device if dist.get_backend(group) == "nccl" else "cpu")
        everyone = [torch.empty_like(mine) for _ in range(K)]
        dist.all_gather(everyone, mine, group=group)
        keys = {}
        for peer in range(K):
            if peer == rank:
                continue
            shared = cls.x25519(secret, bytes(everyone[peer].cpu().tolist()))
            if shared == bytes(32):
                raise RuntimeError("X25519: peer {:d} sent a low-order point".format(peer))
            keys[peer] = cls.pair_key(shared, min(rank, peer), max(rank, peer))
        return cls(keys, rank, n_words, ops)

    def apply(self, q):
        """q += sum_{j > rank} PRG(key_rank,j) - sum_{i < rank} PRG(key_i,rank)   (mod 2^64), in place."""
        n = q.numel()
        if self._tmp is None or self._tmp.numel() != n:
            self._tmp = torch.empty(n, dtype=torch.int64, device=q.device)
        block0 = self.syncs * self.blocks_per_sync
        for peer, key in sorted(self.keys.items()):
            self.ops.keystream(key, block0, self._tmp)
            self.ops.ring_accumulate(q, self._tmp, subtract=peer < self.rank)
        self.syncs += 1


def fedavg_allreduce(flat, out, weight=None, secure=False, precision_fractional=16, base=10, group=None,
                     ops=None, scratch=None, masks=None):
    """Average the clients' arenas.

    flat   : this client's fp32 arena (not modified)
    out    : fp32 tensor of the same size receiving the average (may alias nothing else)
    weight : this client's w_k, or None for the unweighted mean (sum, then / K)
    secure : fixed-precision encode -> ring sum -> decode (the numerics of the reference's secure aggregation);
             with `masks` (a PairwiseMasks) every client's encoded update is hidden under pairwise one-time masks
             that cancel in the sum.  Without `masks` the all-reduce carries the encoded updates in the clear: the
             numbers are identical, the confidentiality is not — train.py always passes masks when K > 1.
    """
    ops = ops or HipArenaOps()
    K = dist.get_world_size(group) if dist.is_initialized() else 1
    out.copy_(flat)
    if weight is not None:
        ops.scale(out, weight)
    if secure:
        scale = float(base ** precision_fractional)
        q = scratch if scratch is not None else torch.empty(out.numel(), dtype=torch.int64, device=out.device)
        ops.encode(out, q, scale)
        if K > 1:
            if masks is not None:
                masks.apply(q)
            dist.all_reduce(q, op=dist.ReduceOp.SUM, group=group)
        ops.decode(q, out, scale)
    elif K > 1:
        dist.all_reduce(out, op=dist.ReduceOp.SUM, group=group)
    if weight is None:
        ops.divide(out, float(K))
    return out


def exchange_mean_std(mean, std, group=None, ops=None, precision_fractional=3, base=10, masks=None):
    """setup_pysyft's secure average of the clients' data statistics (torchlib/utils.py:764-794): each rank
    holds its local (mean, std); they are fixed-point encoded (`fix_precision()` defaults: 10^3), summed in
    the 2^64 ring across the ranks (one int64 all-reduce — additive sharing commutes with the ring sum),
    decoded and divided by the number of clients.  Returns (mean, std) as the reference's `val_mean_std`.
    With `masks` (a PairwiseMasks) a client's statistics leave it under one-time masks, like the model updates."""
    ops = ops or HipArenaOps()
    K = dist.get_world_size(group) if dist.is_initialized() else 1
    both = torch.cat([mean.reshape(-1), std.reshape(-1)]).to(torch.float32).contiguous()
    q = torch.empty(both.numel(), dtype=torch.int64, device=both.device)
    scale = float(base ** precision_fractional)
    ops.encode(both, q, scale)
    if K > 1:
        if masks is not None:
            masks.apply(q)
        dist.all_reduce(q, op=dist.ReduceOp.SUM, group=group)
    out = torch.empty_like(both)
    ops.decode(q, out, scale)
    ops.divide(out, float(K))
    n = mean.numel()
    return out[:n].reshape(mean.shape), out[n:].reshape(std.shape)


def masked_sum(counts, masks, group=None):
    """Ring sum of small non-negative integer vectors (per-class sample counts) across the ranks under the pairwise
    one-time masks: what leaves a client is uniformly random, the masks cancel in the sum.  Returns fp32 counts."""
    q = counts.to(torch.int64).contiguous()
    masks.apply(q)
    dist.all_reduce(q, op=dist.ReduceOp.SUM, group=group)
    return q.to(torch.float32)


def secure_mean_of(stats, ops=None, precision_fractional=3, base=10):
    """exchange_mean_std for clients that live in ONE process: `stats` = [(mean, std), ...]; same arithmetic
    (encode each, ring sum, decode, / K)."""
    ops = ops or HipArenaOps()
    scale = float(base ** precision_fractional)
    acc = None
    for mean, std in stats:
        both = torch.cat([mean.reshape(-1), std.reshape(-1)]).to(torch.float32).contiguous()
        q = torch.empty(both.numel(), dtype=torch.int64, device=both.device)
        ops.encode(both, q, scale)
        if acc is None:
            acc = q
        else:
            _lib.call("primia_ring_add", acc, q, acc, acc.numel(), acc.numel())
    out = torch.empty(acc.numel(), dtype=torch.float32, device=acc.device)
    ops.decode(acc, out, scale)
    ops.divide(out, float(len(stats)))
    n = stats[0][0].numel()
    return out[:n].reshape(stats[0][0].shape), out[n:].reshape(stats[0][1].shape)


class SyncSchedule:
    """Which batches end with a FedAvg sync and who adopts the result — the control flow of
    secure_aggregation_epoch (utils.py:1159-1230), shared by every rank."""

    def __init__(self, num_batches_per_client, sync_every_n_batch):
        self.num_batches = list(num_batches_per_client)
        self.max_batches = max(self.num_batches)
        self.sync_every = int(sync_every_n_batch)

    def trains(self, client, batch_idx):
        return batch_idx < self.num_batches[client]

    def sync_after(self, batch_idx):
        return batch_idx > 0 and batch_idx % self.sync_every == 0

    def adopts(self, client, batch_idx):
        """send_new_models is restricted to workers with num_batches > batch_idx (utils.py:1189-1194)."""
        return self.num_batches[client] > batch_idx


def all_gather_int(value, group=None, device="cpu"):
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return [int(value)]
    t = torch.tensor([int(value)], dtype=torch.int64, device=device)
    outs = [torch.zeros_like(t) for _ in range(dist.get_world_size(group))]
    dist.all_gather(outs, t, group=group)
    return [int(o.item()) for o in outs]


def federated_epoch(engine, loader, args, optimizer=None, group=None, ops=None, local_flat=None, soft_targets=False,
                    masks=None):
    """One federated epoch for THIS rank's client — the body of secure_aggregation_epoch (torchlib/utils.py:1108-1233)
    with one client per rank: the per-worker loop becomes "this rank trains if it still has batches", aggregation()
    + send_new_models() become `fedavg_allreduce` + "adopt if batches are left" (always at the epoch end).

    `args` carries the reference's settings (optimizer, lr, weight_decay, beta1/2, sync_every_n_batch,
    keep_optim_dict, weighted_averaging, unencrypted_aggregation, precision_fractional).  Unless keep_optim_dict the
    optimizer is re-created with lr = args.lr at the start of the epoch and after every mid-epoch sync
    (utils.py:1131-1145,1208-1218).

    Returns (mean loss over ALL clients' steps, as the reference's avg_loss; this client's step count; local_flat =
    the last global average, the reference's "local_model"; the optimizer object now in use)."""
    from .optim import EngineOptimizer

    rank = dist.get_rank(group) if dist.is_initialized() else 0
    dev = engine.flat.device
    counts = all_gather_int(len(loader), group, dev)
    sched = SyncSchedule(counts, args.sync_every_n_batch)
    weight = counts[rank] / float(sum(counts)) if args.weighted_averaging else None
    secure = not args.unencrypted_aggregation
    pf = int(getattr(args, "precision_fractional", 16))
    if local_flat is None:
        local_flat = torch.empty_like(engine.flat)
    scratch = torch.empty(engine.flat.numel(), dtype=torch.int64, device=dev) if secure else None
    if not args.keep_optim_dict or optimizer is None:
        optimizer = EngineOptimizer.from_args(engine, args)
    losses = []
    it = iter(loader)

    def sync(final, batch_idx):
        fedavg_allreduce(engine.flat, local_flat, weight, secure, pf, 10, group, ops, scratch, masks)
        if final or sched.adopts(rank, batch_idx):
            engine.flat.copy_(local_flat)
            # the adopted state dict carries num_batches_tracked = 0 (see torchlib_compat.aggregation)
            for b in engine.num_batches_tracked:
                engine.num_batches_tracked[b] = 0
            engine.refresh_weights()

    for batch_idx in range(sched.max_batches):
        if sched.trains(rank, batch_idx):
            data, target = next(it)
            optimizer.zero_grad()
            sib = getattr(engine, "sibling", None)       # (the ragged final batch of a client's loader)
            eng = engine if sib is None else sib(data.shape[0])
            eng.forward(data)
            losses.append(eng.loss_backward(target, soft=soft_targets).clone())
            optimizer.step() if eng is engine else optimizer.step(eng)
        if sched.sync_after(batch_idx):
            sync(False, batch_idx)
            if not args.keep_optim_dict:
                optimizer = EngineOptimizer.from_args(engine, args)
    sync(True, sched.max_batches)
    tot = torch.zeros(2, dtype=torch.float64, device=dev)
    if losses:
        tot[0] = torch.stack(losses).double().sum()
        tot[1] = len(losses)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(tot, op=dist.ReduceOp.SUM, group=group)
    mean_loss = float((tot[0] / tot[1]).item()) if tot[1] > 0 else float("nan")
    return mean_loss, len(losses), local_flat, optimizer
