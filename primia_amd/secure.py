"""Encrypted inference on GPU: fixed-precision additive secret sharing over Z_2^64 between two
parties (model_owner = party 0, data_owner = party 1) with a dealer (crypto provider) handing out
Beaver triples and FSS keys — the arithmetic of PySyft's FPT > AST tensor chain as PriMIA's
inference.py drives it (inference.py:279-321, protocol "fss").

Each method below is the host-side orchestration of one reference operation; the arithmetic runs
in the party-local HIP kernels of libprimia_hip.so (the functions PySyft registers with
@allow_command).  A shared tensor is a pair [share_0, share_1] of int64 device tensors.  An
"open" sums the two parties' buffers: with both parties on one GPU that is one ring-add kernel,
with one party per GPU it is a 2-rank RCCL all-reduce (`DistOpener`).

Randomness is never drawn implicitly: every triple, FSS key and re-sharing mask comes from a
`Dealer`, so a run is a bit-exact function of the dealer's stream (and can be replayed by the CPU
oracle in the tests).

Reference map (paths under syft/frameworks/torch/):
  encode / share / open      tensors/interpreters/precision.py:117-144, additive_shared.py:287-365
  beaver mul / matmul        mpc/spdz.py:21-197, mpc/beaver.py:7-63
  truncation                 precision.py:146-160, additive_shared.py:672-678
  fss comparison             mpc/fss.py:97-281, mpc/primitives.py:237-253
  relu                       additive_shared.py:922-925
  conv2d / pools / bn / lin  nn/functional.py:10-14, 44-75, 204-308, 460-525
  public add / sub / rsub    additive_shared.py:440-527 (constants are RE-SHARED with fresh randomness)
"""
import torch

from . import _lib
from ._lib import call

I64 = torch.int64


def _empty_like(t):
    return torch.empty_like(t)


class Dealer:
    """The crypto provider: generates correlated randomness on the GPU from a seeded generator.

    build_triple (mpc/beaver.py:7-63): a, b uniform int64, c = a∘b, each split into two shares.
    build_fss_keys (mpc/primitives.py:237-253): DIF keys + alpha additively split mod 2^32.
    """

    def __init__(self, device, seed=0):
        self.device = torch.device(device)
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(seed)
        self.log = None  # set to a list to record every primitive handed out (tests replay it)
        self.tape = None  # set to a list to keep every primitive ON THE DEVICE (offline phase, see PreloadedDealer)
        self.requests = None  # set to a list to record (method, args) of every request (GraphedSecureInference)

    def rand64(self, *shape):
        return torch.randint(-2 ** 63, 2 ** 63 - 1, shape, dtype=I64, device=self.device, generator=self.gen)

    def _split(self, v):
        r = self.rand64(*v.shape)
        s1 = _empty_like(v)
        call("primia_ring_sub", v, r, s1, v.numel(), v.numel())
        return [r, s1]

    def triple(self, op, xshape, yshape):
        if self.requests is not None:
            self.requests.append(("triple", (op, tuple(xshape), tuple(yshape))))
        a, b = self.rand64(*xshape), self.rand64(*yshape)
        if op == "mul":
            # element-wise with the smaller operand broadcast over the leading dims
            big, small = (a, b) if a.numel() >= b.numel() else (b, a)
            c = _empty_like(big)
            call("primia_ring_mul", big, small, c, big.numel(), small.numel())
        else:
            M, K, N = xshape[-2], xshape[-1], yshape[-1]
            c = torch.empty(*xshape[:-1], N, dtype=I64, device=self.device)
            call("primia_ring_matmul", a, b, c, M, K, N, 0)
        sa, sb, sc = self._split(a), self._split(b), self._split(c)
        t = [(sa[j], sb[j], sc[j]) for j in range(2)]
        if self.log is not None:
            self.log.append(("triple", op, [tuple(x.cpu().numpy() for x in t[j]) for j in range(2)]))
        if self.tape is not None:
            self.tape.append(t)
        return t

    def dif_keys(self, n):
        if self.requests is not None:
            self.requests.append(("dif_keys", (n,)))
        dev = self.device
        alpha = torch.randint(0, 2 ** 32, (n,), dtype=I64, device=dev, generator=self.gen)
        s0 = self.rand64(2, 2, n)
        s0[:, 0] &= 0x7FFFFFFFFFFFFFFF  # randbit: word 0 carries 63 bits (fss.py:495-501)
        bits = torch.empty(32, n, dtype=torch.uint8, device=dev)
        cw_sigma = torch.empty(32, 2, n, dtype=I64, device=dev)
        cw_s = torch.empty(32, 2, n, dtype=I64, device=dev)
        leaf = torch.empty(33, n, dtype=torch.int32, device=dev)
        call("primia_dif_keygen", alpha, s0, bits, cw_sigma, cw_s, leaf, n)
        r = torch.randint(0, 2 ** 32, (n,), dtype=I64, device=dev, generator=self.gen)
        a1 = (alpha - r) & 0xFFFFFFFF
        keys = [dict(alpha=[r, a1][b], s0=s0[b].contiguous(), bits=bits, cw_sigma=cw_sigma, cw_s=cw_s, cw_leaf=leaf)
                for b in range(2)]
        if self.log is not None:
            self.log.append(("dif", n, alpha.cpu().numpy(), s0.cpu().numpy(), r.cpu().numpy()))
        if self.tape is not None:
            self.tape.append(keys)
        return keys

    def const_mask(self, *shape):
        if self.requests is not None:
            self.requests.append(("const_mask", tuple(shape)))
        r = self.rand64(*shape)
        if self.log is not None:
            self.log.append(("mask", r.cpu().numpy()))
        if self.tape is not None:
            self.tape.append(r)
        return r


class PreloadedDealer:
    """Online phase only: hands out primitives that a Dealer generated earlier (its `tape`), in
    the same order — the reference's pre-provisioned crypto store (mpc/primitives.py:161-235)."""

    def __init__(self, tape, device):
        self.tape, self.pos, self.device = tape, 0, torch.device(device)
        self.log = None

    def _next(self):
        v = self.tape[self.pos]
        self.pos += 1
        return v

    def triple(self, op, xshape, yshape):
        return self._next()

    def dif_keys(self, n):
        return self._next()

    def const_mask(self, *shape):
        return self._next()


class LocalOpener:
    """Both parties' shares live on this GPU: open = one ring add."""

    def open(self, s0, s1):
        out = _empty_like(s0)
        call("primia_ring_add", s0, s1, out, s0.numel(), s0.numel())
        return out


class DistOpener:
    """One party per rank (party j = rank j of a 2-rank group, e.g. two GPUs over xGMI): every rank
    passes ITS share in slot `rank` and a dummy in the other; open = all_reduce(SUM) of the local
    share — the 2-party exchange the reference routes through the orchestrator
    (mpc/spdz.py:162-176, mpc/fss.py:158-170)."""

    def __init__(self, group=None):
        import torch.distributed as dist

        self.dist, self.group = dist, group
        self.rank = dist.get_rank(group)

    def open(self, s0, s1):
        mine = (s0, s1)[self.rank].clone()
        self.dist.all_reduce(mine, op=self.dist.ReduceOp.SUM, group=self.group)  # int64 sum wraps mod 2^64
        return mine


class SecureContext:
    def __init__(self, dealer, base=10, precision_fractional=16, opener=None):
        self.dealer = dealer
        self.base = base
        self.pf = precision_fractional
        self.scale = base ** precision_fractional
        self.opener = opener or LocalOpener()
        self.stats = {"beaver_mul": 0, "beaver_matmul": 0, "dif_evals": 0}

    # ---- encode / share / reconstruct -----------------------------------------------------------
    def encode(self, x):
        q = torch.empty(x.shape, dtype=I64, device=x.device)
        call("primia_fx_encode", x.contiguous().to(torch.float32), q, x.numel(), float(self.scale))
        return q

    def decode(self, q):
        x = torch.empty(q.shape, dtype=torch.float32, device=q.device)
        call("primia_fx_decode", q, x, q.numel(), float(self.scale))
        return x

    def share(self, q):
        """share_secret (additive_shared.py:317-365): (r, q - r)."""
        r = self.dealer.const_mask(*q.shape)
        s1 = _empty_like(q)
        call("primia_ring_sub", q, r, s1, q.numel(), q.numel())
        return [r, s1]

    def reconstruct(self, x):
        return self.opener.open(x[0], x[1])

    # ---- local (per-share) ops -------------------------------------------------------------------
    def _ew(self, fn, a, b):
        out = []
        for j in range(2):
            big, small = (a[j], b[j])
            if small.numel() > big.numel():
                raise ValueError("second operand must not be larger")
            o = _empty_like(big)
            call(fn, big, small, o, big.numel(), small.numel())
            out.append(o)
        return out

    def add(self, a, b):
        return self._ew("primia_ring_add", a, b)

    def sub(self, a, b):
        return self._ew("primia_ring_sub", a, b)

    def neg(self, a):
        out = []
        for j in range(2):
            o = _empty_like(a[j])
            call("primia_ring_scale", a[j], -1, o, a[j].numel())
            out.append(o)
        return out

    def trunc(self, a, d):
        out = []
        for j in range(2):
            o = _empty_like(a[j])
            call("primia_trunc_div", a[j], int(d), o, a[j].numel())
            out.append(o)
        return out

    def sub_public_scalar(self, a, value):
        """AST - int (additive_shared.py:453-484, 506-524): the constant becomes a FRESH random
        sharing of shape [1] that is subtracted share-wise (broadcast)."""
        c = torch.full((1,), int(value), dtype=I64, device=a[0].device)  # device-side fill: graph-capturable
        return self.sub(a, self.share(c))

    # ---- Beaver -----------------------------------------------------------------------------------
    def beaver_mul(self, x, y):
        """Element-wise private product (no truncation).  One operand may be a vector broadcast over
        the other's leading dims; the ring product is symmetric so the big one is taken first."""
        swap = x[0].numel() < y[0].numel()
        t = self.dealer.triple("mul", tuple(x[0].shape), tuple(y[0].shape))
        if swap:
            x, y = y, x
            t = [(tj[1], tj[0], tj[2]) for tj in t]
        n, nb = x[0].numel(), y[0].numel()
        d = [_empty_like(x[0]) for _ in range(2)]
        e = [_empty_like(y[0]) for _ in range(2)]
        for j in range(2):  # spdz_mask
            call("primia_ring_sub", x[j], t[j][0], d[j], n, n)
            call("primia_ring_sub", y[j], t[j][1], e[j], nb, nb)
        delta, eps = self.opener.open(d[0], d[1]), self.opener.open(e[0], e[1])
        z = []
        for j in range(2):  # spdz_compute
            o = _empty_like(x[0])
            call("primia_beaver_combine_mul", j, delta, eps, t[j][0], t[j][1], t[j][2], o, n, nb)
            z.append(o)
        self.stats["beaver_mul"] += 1
        return z

    def beaver_matmul(self, x, y):
        M, K = x[0].shape[-2], x[0].shape[-1]
        N = y[0].shape[-1]
        t = self.dealer.triple("matmul", tuple(x[0].shape), tuple(y[0].shape))
        d = [_empty_like(x[0]) for _ in range(2)]
        e = [_empty_like(y[0]) for _ in range(2)]
        for j in range(2):
            call("primia_ring_sub", x[j], t[j][0], d[j], x[j].numel(), x[j].numel())
            call("primia_ring_sub", y[j], t[j][1], e[j], y[j].numel(), y[j].numel())
        delta, eps = self.opener.open(d[0], d[1]), self.opener.open(e[0], e[1])
        z = []
        scratch = torch.empty(K * N, dtype=I64, device=x[0].device)
        for j in range(2):
            o = torch.empty(*x[0].shape[:-1], N, dtype=I64, device=x[0].device)
            call("primia_beaver_combine_matmul", j, delta, eps, t[j][0], t[j][1], t[j][2], o, scratch, M, K, N)
            z.append(o)
        self.stats["beaver_matmul"] += 1
        return z

    def fpt_mul(self, x, y):
        """FPT * FPT (precision.py:309-316, 356-358): Beaver mul, then per-share truncation."""
        return self.trunc(self.beaver_mul(x, y), self.scale)

    def fpt_matmul(self, x, y):
        """FPT @ FPT (precision.py:419-463)."""
        return self.trunc(self.beaver_matmul(x, y), self.scale)

    # ---- FSS comparison ------------------------------------------------------------------------------
    def le(self, x1, x2):
        """fss.le(x1, x2) (mpc/fss.py:97-185, 279): int64 shares of the bit [x1 <= x2]."""
        n = x1[0].numel()
        keys = self.dealer.dif_keys(n)
        r = []
        for j in range(2):  # mask_builder
            o = _empty_like(x1[j])
            call("primia_fss_mask", x1[j], x2[j], keys[j]["alpha"], o, n)
            r.append(o)
        masked = torch.empty(n, dtype=torch.int32, device=x1[0].device)
        call("primia_fss_open", r[0], r[1], masked, n)
        out = []
        for j in range(2):  # evaluate
            o = _empty_like(x1[j])
            k = keys[j]
            call("primia_dif_eval", j, masked, k["s0"], k["bits"], k["cw_sigma"], k["cw_s"], k["cw_leaf"], o, n)
            out.append(o)
        self.stats["dif_evals"] += n
        return out

    def relu(self, x):
        """AST.relu under fss (additive_shared.py:922-925): x * (x >= 0), (x >= 0) = le(x - x, x)."""
        zero = self.sub(x, x)
        return self.beaver_mul(x, self.le(zero, x))

    def _max_pair(self, left, right):
        """left + (right >= left) * (right - left)  (nn/functional.py:494)."""
        bit = self.le(left, right)
        return self.add(left, self.beaver_mul(bit, self.sub(right, left)))

    def _cols(self, x, rows, w, start, length):
        out = []
        for j in range(2):
            o = torch.empty(rows, length, dtype=I64, device=x[j].device)
            call("primia_ring_slice_cols", x[j], o, rows, w, start, length)
            out.append(o)
        return out

    # ---- layers (nn/functional.py) ------------------------------------------------------------------
    def conv2d(self, x, w, stride, padding):
        """conv2d (nn/functional.py:204-308): per-share im2col, Beaver matmul + truncation,
        per-share reshape.  x shares [1,C,H,W]; w shares [O,C,R,S]; no bias in ResNet convs."""
        B, C, H, W = x[0].shape
        O, _, R, S = w[0].shape
        Ho, Wo = (H + 2 * padding - R) // stride + 1, (W + 2 * padding - S) // stride + 1
        K = C * R * S
        im, wt = [], []
        for j in range(2):
            a = torch.empty(B, Ho * Wo, K, dtype=I64, device=x[j].device)
            call("primia_im2col_syft", x[j], a, B, C, H, W, R, S, stride, padding)
            im.append(a)
            # weight.reshape(O, -1).t(): [K, O]
            t = torch.empty(K, O, dtype=I64, device=x[j].device)
            call("primia_col2out_syft", w[j], None, t, 1, O, K)
            wt.append(t)
        res = self.fpt_matmul(im, wt)
        out = []
        for j in range(2):
            o = torch.empty(B, O, Ho, Wo, dtype=I64, device=x[j].device)
            call("primia_col2out_syft", res[j], None, o, B, Ho * Wo, O)
            out.append(o)
        return out

    def reciprocal_newton(self, v):
        """FPT.reciprocal(method="newton") (precision.py:507-518), C = 20, 80 iterations."""
        C = 20
        # x0 = (C + 1 - v) / C
        y = self.neg(self.sub_public_scalar(v, (C + 1) * self.scale))
        x = self.trunc(y, C)
        for _ in range(79):
            xx = self.fpt_mul(x, x)
            vxx = self.fpt_mul(v, xx)
            y = self.neg(self.sub_public_scalar(vxx, (C + 1) * self.scale))
            x = self.trunc(self.fpt_mul(y, x), C)
        return x

    def batch_norm_eval(self, x, mean, var, weight, bias, inv=None):
        """batch_norm in eval mode (nn/functional.py:44-75): ((x - mean) * newton(var)) * w + b on
        [H*W, C] rows — no explicit sqrt and no eps, exactly as the reference computes it (its
        "newton" iteration converges to var^-1/2).  `inv` may carry shares of newton(var) computed
        earlier (see SecureResNet18.precompute_inv)."""
        B, C, H, W = x[0].shape
        if B != 1:
            raise ValueError("encrypted inference runs one image at a time (inference.py:292)")
        rows = []
        for j in range(2):  # permute(1,0,2,3).reshape(C,-1).t()  -> [B*H*W, C]  (B == 1)
            o = torch.empty(B * H * W, C, dtype=I64, device=x[j].device)
            call("primia_col2out_syft", x[j], None, o, 1, C, B * H * W)
            rows.append(o)
        if inv is None:
            inv = self.reciprocal_newton(var)
        normalized = self.fpt_mul(inv, self.sub(rows, mean))
        result = self.add(self.fpt_mul(normalized, weight), bias)
        out = []
        for j in range(2):
            o = torch.empty(B, C, H, W, dtype=I64, device=x[j].device)
            call("primia_col2out_syft", result[j], None, o, 1, B * H * W, C)
            out.append(o)
        return out

    def max_pool2d_3x3s2(self, x):
        """_pool2d(mode="max") for a 3x3 window (nn/functional.py:460-508): unroll to 9 columns,
        binary tree on the first 8, then against the 9th."""
        B, C, H, W = x[0].shape
        Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
        rows = B * C * Ho * Wo
        im = []
        for j in range(2):
            o = torch.empty(rows, 9, dtype=I64, device=x[j].device)
            call("primia_pool_unroll_syft", x[j], o, B, C, H, W, 3, 2, 1)
            im.append(o)
        res = self._max_pair(self._cols(im, rows, 9, 0, 4), self._cols(im, rows, 9, 4, 4))
        res = self._max_pair(self._cols(res, rows, 4, 0, 2), self._cols(res, rows, 4, 2, 2))
        left = self._max_pair(self._cols(res, rows, 2, 0, 1), self._cols(res, rows, 2, 1, 1))
        res = self._max_pair(left, self._cols(im, rows, 9, 8, 1))
        return [r.view(B, C, Ho, Wo) for r in res]

    def avg_pool2d(self, x, k):
        """_pool2d(mode="avg"), stride = kernel (nn.AvgPool2d(k)): per-share window sum, then the
        per-share truncating division of AST.mean (additive_shared.py:719-729)."""
        B, C, H, W = x[0].shape
        Ho, Wo = (H - k) // k + 1, (W - k) // k + 1
        rows = B * C * Ho * Wo
        out = []
        for j in range(2):
            im = torch.empty(rows, k * k, dtype=I64, device=x[j].device)
            call("primia_pool_unroll_syft", x[j], im, B, C, H, W, k, k, 0)
            s = torch.empty(rows, dtype=I64, device=x[j].device)
            call("primia_ring_rowsum", im, s, rows, k * k)
            o = _empty_like(s)
            call("primia_trunc_div", s, k * k, o, rows)
            out.append(o.view(B, C, Ho, Wo))
        return out

    def linear(self, x, w, b):
        """F.linear -> torch.addmm(bias, input, weight.t()) -> FPT.addmm (nn/functional.py:10-14,
        precision.py:822-827): matmul + truncation, then + bias."""
        O, I = w[0].shape
        wt = []
        for j in range(2):
            t = torch.empty(I, O, dtype=I64, device=w[j].device)
            call("primia_col2out_syft", w[j], None, t, 1, O, I)
            wt.append(t)
        return self.add(self.fpt_matmul(x, wt), b)


class SecureResNet18:
    """ResNet-18 forward on secret shares — `model.fix_precision().share()` followed by
    `model(data)` in inference.py:279-321, including the stem swap `model.pool, model.relu =
    model.relu, model.pool` (:289): conv1 -> bn1 -> MAXPOOL -> RELU."""

    def __init__(self, ctx: SecureContext, state_dict, input_size=224, blocks=None):
        self.ctx = ctx
        self.input_size = input_size
        dev = ctx.dealer.device
        self.p = {}
        # hook.py:626-632,738-765: every parameter AND buffer is encoded and shared
        for k, v in state_dict.items():
            if k.endswith("num_batches_tracked"):
                continue
            self.p[k] = ctx.share(ctx.encode(v.to(dev)))
        self.blocks = blocks if blocks is not None else [
            (f"layer{li}.{bi}", (2 if (li > 1 and bi == 0) else 1)) for li in range(1, 5) for bi in range(2)]

    def bn_prefixes(self):
        out = ["bn1"]
        for prefix, _ in self.blocks:
            out += [prefix + ".bn1", prefix + ".bn2"]
            if (prefix + ".downsample.0.weight") in self.p:
                out.append(prefix + ".downsample.1")
        return out

    def precompute_inv(self):
        """newton(running_var) of every BatchNorm layer in ONE batched 80-step iteration.

        The reference recomputes it layer by layer inside each forward (nn/functional.py:62-69);
        it does not depend on the image, and the iteration is element-wise, so running all layers'
        channels as one vector performs exactly the same arithmetic per channel — it only turns
        20 x 80 x 3 tiny Beaver rounds into 80 x 3 (the per-element randomness is whatever slice of
        the batched triple that channel receives)."""
        c = self.ctx
        names = self.bn_prefixes()
        var = [torch.cat([self.p[n + ".running_var"][j] for n in names]) for j in range(2)]
        inv = c.reciprocal_newton(var)
        out, off = {}, 0
        for n in names:
            k = self.p[n + ".running_var"][0].numel()
            out[n] = [inv[j][off:off + k].contiguous() for j in range(2)]
            off += k
        return out

    def _bn(self, x, prefix):
        p = self.p
        return self.ctx.batch_norm_eval(x, p[prefix + ".running_mean"], p[prefix + ".running_var"],
                                        p[prefix + ".weight"], p[prefix + ".bias"], inv=self._inv[prefix])

    def forward_shares(self, x):
        c, p = self.ctx, self.p
        self._inv = self.precompute_inv()
        x = c.conv2d(x, p["conv1.weight"], 2, 3)
        x = self._bn(x, "bn1")
        x = c.max_pool2d_3x3s2(x)      # swapped stem (inference.py:289)
        x = c.relu(x)
        for prefix, stride in self.blocks:
            identity = x
            out = c.conv2d(x, p[prefix + ".conv1.weight"], stride, 1)
            out = c.relu(self._bn(out, prefix + ".bn1"))
            out = c.conv2d(out, p[prefix + ".conv2.weight"], 1, 1)
            out = self._bn(out, prefix + ".bn2")
            if (prefix + ".downsample.0.weight") in p:
                identity = c.conv2d(x, p[prefix + ".downsample.0.weight"], stride, 0)
                identity = self._bn(identity, prefix + ".downsample.1")
            x = c.relu(c.add(out, identity))
        k = x[0].shape[-1]
        x = c.avg_pool2d(x, k)
        x = [t.reshape(1, -1) for t in x]
        return c.linear(x, p["fc.weight"], p["fc.bias"])

    def __call__(self, image):
        """image: fp32 [1, C, S, S] on the GPU -> decoded fp32 logits [1, classes]."""
        c = self.ctx
        xs = c.share(c.encode(image))
        out = self.forward_shares(xs)
        return c.decode(c.reconstruct(out))


def _copy_primitive(dst, src):
    """In-place refill of one tape entry (tensor, tuple/list of tensors, or key dicts) from a fresh one."""
    if torch.is_tensor(dst):
        dst.copy_(src)
    elif isinstance(dst, dict):
        for k in dst:
            _copy_primitive(dst[k], src[k])
    else:
        for d, s_ in zip(dst, src):
            _copy_primitive(d, s_)


class GraphedSecureInference:
    """Serving form of the encrypted forward: the online phase (about 6,500 small launches per image) is
    captured ONCE as a hipGraph over static buffers — the input image and every correlated-randomness
    primitive — and replayed per image; `refill()` has the dealer regenerate all per-image primitives into
    the same buffers (the reference's pre-provisioned crypto store, mpc/primitives.py:161-235, refilled
    between requests).  Results are bit-identical to the eager SecureResNet18 fed the same primitives."""

    def __init__(self, state_dict, device, input_size=224, precision_fractional=16, base=10, seed=0, blocks=None):
        self.device = torch.device(device)
        self.image = torch.zeros(1, 3, input_size, input_size, dtype=torch.float32, device=self.device)
        self.dealer = Dealer(self.device, seed)
        self.dealer.tape, self.dealer.requests = [], []
        ctx = SecureContext(self.dealer, base, precision_fractional)
        model = SecureResNet18(ctx, state_dict, input_size, blocks)
        self._n_model = len(self.dealer.tape)          # primitives consumed by sharing the model (kept)
        model(self.image)                              # offline pass: fills the tape, warms every kernel
        self.tape, self.requests = self.dealer.tape, self.dealer.requests
        self.dealer.tape = self.dealer.requests = None
        self.stats = dict(ctx.stats)
        pre = PreloadedDealer(self.tape, self.device)
        self._ctx = SecureContext(pre, base, precision_fractional)
        self._model = SecureResNet18(self._ctx, state_dict, input_size, blocks)   # re-shares with the same masks
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=side):
                self.out = self._model(self.image)
        torch.cuda.current_stream().wait_stream(side)

    def refill(self):
        """Fresh per-image primitives from the dealer, written into the captured buffers."""
        for i in range(self._n_model, len(self.tape)):
            kind, args = self.requests[i]
            _copy_primitive(self.tape[i], getattr(self.dealer, kind)(*args))

    def __call__(self, image, refill=True):
        if refill:
            self.refill()
        self.image.copy_(image)
        self.graph.replay()
        return self.out
