"""Minimal runtime that EXECUTES the reference's own MPC source files where they lie under
/root/reference (build container only; test infrastructure for the golden-minting scripts).

The vendored PySyft package does not import under torch 2 / Python 3.10, but the files
that hold the arithmetic of the encrypted-inference path do load one by one once the control plane they
import is replaced by stand-ins.  Loaded from the reference, unmodified unless listed:

    syft/exceptions.py                               EmptyCryptoPrimitiveStoreError only (AST-extracted)
    syft/generic/frameworks/overload.py              @overloaded.method / .module
    syft/frameworks/torch/mpc/__init__.py            crypto_protocol
    syft/frameworks/torch/mpc/fss.py                 with the three documented NumPy-2 shims
    syft/frameworks/torch/mpc/beaver.py              build_triple
    syft/frameworks/torch/mpc/primitives.py          PrimitiveStorage (get_keys / provide / add)
    syft/frameworks/torch/mpc/spdz.py                spdz_mask / spdz_compute / spdz_mul
    syft/frameworks/torch/tensors/interpreters/additive_shared.py   AdditiveSharingTensor
    syft/frameworks/torch/tensors/interpreters/precision.py         FixedPrecisionTensor
    syft/frameworks/torch/nn/functional.py           conv2d / batch_norm / pools / linear

Stand-ins written here (control plane only — no protocol arithmetic):
  * workers: an id, a PrimitiveStorage, and message passing reduced to "call it on that worker";
  * pointers: a share held by a worker is a local int64 tensor of class `Share` that remembers its
    owner (`PointerTensor` := Share; `send`/`get` are identity);
  * `remote(f, location)`: calls f with tensor arguments re-owned by `location`;
  * hook_args: unwrap = `.child`, wrap = `cls(**attrs).on(x, wrap=False)` (what the real rules do for
    FPT > AST chains); TorchHook's auto-forwarding of tensor methods (`permute`, `reshape`, `t`, ...) to
    `.child` (per share for an AdditiveSharingTensor) is `AbstractTensor.__getattr__` below;
  * torch 1.4 semantics the reference relies on and torch 2 changed: integer `tensor / int` truncates
    toward zero (`Share.__truediv__`; precision.py:149-151);
  * `multiprocessing.Pool` inside spdz_compute -> serial map (same partition / concat code runs);
  * `shaloop` -> independent SHA-256/512 of each 16-byte row (oracle/sha_loop.c or hashlib).

Every primitive the crypto provider hands out is recorded in `Runtime.log` in the order the
reference requests it, in the layout oracle.secure_oracle.ReplayDealer consumes:
    ("triple", op, [(a0, b0, c0), (a1, b1, c1)])
    ("dif", n, alpha, s0_pair, r)          party 1 holds r, party 0 holds alpha - r (primitives.py:250-251)
    ("mask", r)                            share 0 of a fresh sharing (additive_shared.py:336-365)
"""
import ast
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
SYFT = f"{REF}/syft"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# ---------------------------------------------------------------------------------------------------
# shares / pointers
# ---------------------------------------------------------------------------------------------------
def _first_owner(args):
    for a in args:
        if isinstance(a, Share) and getattr(a, "owner", None) is not None:
            return a.owner
        if isinstance(a, (list, tuple)):
            o = _first_owner(a)
            if o is not None:
                return o
    return None


def _tag(out, owner):
    if isinstance(out, Share):
        out.owner = owner
    elif isinstance(out, (list, tuple)):
        for o in out:
            _tag(o, owner)


class Share(torch.Tensor):
    """An int64 share living on a worker.  Arithmetic is torch's own; only integer `/` is put back to
    its torch-1.4 meaning (C truncation)."""

    is_wrapper = False

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        out = super().__torch_function__(func, types, args, kwargs or {})
        _tag(out, _first_owner(args))
        return out

    @property
    def location(self):
        return self.owner

    def __truediv__(self, d):
        if not self.dtype.is_floating_point and (isinstance(d, int) or (torch.is_tensor(d) and not d.dtype.is_floating_point)):
            return torch.div(self, d, rounding_mode="trunc")
        return torch.Tensor.__truediv__(self, d)

    def send(self, owner, **kw):
        self.owner = _RT.worker(owner)
        return self

    def get(self):
        return self

    def wrap(self):
        return self


def as_share(t, owner):
    s = t.detach().as_subclass(Share) if not isinstance(t, Share) else t.detach().view_as(t).as_subclass(Share)
    s.owner = owner
    return s


# ---------------------------------------------------------------------------------------------------
# workers
# ---------------------------------------------------------------------------------------------------
class Worker:
    verbose = False

    def __init__(self, rt, wid):
        self.rt, self.id = rt, wid
        self.crypto_store = None
        self.clients = []

    def get_worker(self, w):
        return self.rt.worker(w)

    def de_register_obj(self, obj):
        pass

    # PrimitiveStorage.provide_primitives: message = ("feed_crypto_primitive_store", payload)
    def create_worker_command_message(self, name, _ret, payload):
        assert name == "feed_crypto_primitive_store"
        return payload

    def send_msg(self, payload, worker):
        self.rt.record(worker, payload)
        worker.crypto_store.add_primitives(payload)

    def __repr__(self):
        return f"<Worker {self.id}>"


# ---------------------------------------------------------------------------------------------------
# syft.generic.* stand-ins
# ---------------------------------------------------------------------------------------------------
class AbstractTensor:
    is_wrapper = False

    def __init__(self, id=None, owner=None, tags=None, description=None, child=None):
        self.id, self.owner, self.tags, self.description = id, owner, tags, description
        self.child = child

    def on(self, tensor, wrap=True):
        assert not wrap, "the runtime only builds unwrapped chains"
        self.child = tensor
        return self

    def wrap(self, **kw):
        return self

    def has_child(self):
        return self.child is not None

    def get_class_attributes(self):
        return {}

    @property
    def shape(self):
        return self.child.shape

    def __len__(self):
        return self.shape[0]

    def __neg__(self):
        return self * -1

    def get(self):
        # syft/generic/abstract/tensor.py: fetch the child and keep this tensor type on top
        return type(self)(**self.get_class_attributes()).on(self.child.get(), wrap=False)

    def float_prec(self):
        return self.float_precision()

    def __getattr__(self, name):
        # TorchHook gives every syft tensor type the torch.Tensor methods it does not define itself: the
        # call is forwarded to .child (to every share of an AdditiveSharingTensor) and the result re-wrapped
        # (syft/generic/frameworks/hook/hook.py _get_hooked_syft_method / _get_hooked_additive_shared_method).
        if name.startswith("_") or name in ("child",) or not hasattr(torch.Tensor, name):
            raise AttributeError(name)

        def forwarded(*args, **kwargs):
            child = self.child
            if isinstance(child, dict):
                def pick(a, w):
                    a = _unwrap(a)
                    return a[w] if isinstance(a, dict) else a

                resp = {w: getattr(s, name)(*[pick(a, w) for a in args], **{k: pick(v, w) for k, v in kwargs.items()})
                        for w, s in child.items()}
            else:
                resp = getattr(child, name)(*_unwrap(args), **_unwrap(kwargs))
            return _wrap(resp, type(self), self.get_class_attributes())

        return forwarded


def _unwrap(x):
    if isinstance(x, AbstractTensor):
        return x.child
    if isinstance(x, tuple):
        return tuple(_unwrap(a) for a in x)
    if isinstance(x, list):
        return [_unwrap(a) for a in x]
    if isinstance(x, dict):
        return {k: _unwrap(v) for k, v in x.items()}
    return x


def _wrap(resp, wrap_type, wrap_args):
    if isinstance(resp, tuple):
        return tuple(_wrap(r, wrap_type, wrap_args) for r in resp)
    if isinstance(resp, (torch.Tensor, dict, AbstractTensor)):
        return wrap_type(**wrap_args).on(resp, wrap=False)
    return resp


class _HookArgs(types.ModuleType):
    def __init__(self):
        super().__init__("syft.generic.frameworks.hook.hook_args")

    @staticmethod
    def unwrap_args_from_method(attr, method_self, args_, kwargs_):
        return method_self.child, _unwrap(args_), _unwrap(kwargs_)

    @staticmethod
    def unwrap_args_from_function(attr, args_, kwargs_):
        return _unwrap(args_), _unwrap(kwargs_), None

    @staticmethod
    def hook_response(attr, response, wrap_type, wrap_args={}, new_self=None):
        if attr[0:3] == "__i" and attr != "__iter__":
            return new_self
        return _wrap(response, wrap_type, wrap_args)

    @staticmethod
    def default_register_tensor(*cls):
        pass


def _remote(func, location):
    worker = _RT.worker(location)

    def call(*args, return_value=False, return_arity=1, **kwargs):
        args = [as_share(a, worker) if torch.is_tensor(a) else a for a in args]
        out = func(*args, **kwargs)
        if isinstance(out, tuple):
            return tuple(as_share(o, worker) if torch.is_tensor(o) else o for o in out)
        return as_share(out, worker) if torch.is_tensor(out) else out

    return call


class _SerialPool:
    def starmap(self, f, argss):
        return [f(*a) for a in argss]

    def close(self):
        pass


class _SerialMP:
    Pool = _SerialPool

    @staticmethod
    def cpu_count():
        return 4


# ---------------------------------------------------------------------------------------------------
# loading
# ---------------------------------------------------------------------------------------------------
def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _load(name, path, patch=None):
    src = open(path, encoding="utf-8").read()
    if patch:
        src = patch(src)
    m = types.ModuleType(name)
    m.__file__ = path
    sys.modules[name] = m
    exec(compile(src, path, "exec"), m.__dict__)
    return m


def _extract(path, names, glob):
    tree = ast.parse(open(path, encoding="utf-8").read())
    body = [n for n in tree.body if isinstance(n, (ast.ClassDef, ast.FunctionDef)) and n.name in names]
    assert {n.name for n in body} == set(names), (path, names)
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), glob)
    return glob


def _fss_shims(src):
    shims = [
        ("CW_n = (-1) ** t[n, 1] * (", "CW_n = (1 - 2 * t[n, 1].astype(np.int64)) * ("),
        ("CW_leaf[i] = (-1) ** τ[i + 1, 1] * (", "CW_leaf[i] = (1 - 2 * τ[i + 1, 1].astype(np.int64)) * ("),
        ("CW_leaf[n] = (-1) ** t[n, 1] * (", "CW_leaf[n] = (1 - 2 * t[n, 1].astype(np.int64)) * ("),
    ]
    for old, new in shims:
        assert src.count(old) == 1, old
        src = src.replace(old, new)
    return src


_RT = None


class Runtime:
    """One orchestrator (local worker = crypto provider of the autogenerate path, spdz.py:156-160 /
    fss.py:142-146) and two parties, as inference.py sets them up (model_owner, data_owner)."""

    def __init__(self, parties=("model_owner", "data_owner")):
        global _RT
        _RT = self
        self.log = []
        self._pending = {}
        self.workers = {}
        self.local_worker = Worker(self, "me")
        self.workers["me"] = self.local_worker
        self.parties = [Worker(self, p) for p in parties]
        for w in self.parties:
            self.workers[w.id] = w
        self.local_worker.clients = self.parties
        self._install()

    def worker(self, w):
        return w if isinstance(w, Worker) else self.workers[w]

    def _install(self):
        from oracle import secure_oracle as S

        rt = self
        if not hasattr(np, "bool"):
            np.bool = np.bool_
        # torch.Tensor attributes the hook adds and the loaded code reads on plain tensors
        torch.Tensor.owner = self.local_worker
        torch.Tensor.is_wrapper = False

        def t_send(t, owner, **kw):
            return as_share(t, rt.worker(owner))

        torch.Tensor.send = t_send

        def t_share(t, *owners, protocol="snn", field=None, dtype=None, crypto_provider=None, no_wrap=False, **kw):
            # native.py:887-949, the branch without a child
            assert not t.dtype.is_floating_point
            sy_ = sys.modules["syft"]
            return sy_.AdditiveSharingTensor(protocol=protocol, field=field, dtype=dtype, crypto_provider=crypto_provider,
                                             owner=t.owner).on(t.clone(), wrap=False).share_secret(*owners)

        torch.Tensor.share = t_share
        torch.Tensor.wrap = lambda t, **kw: t
        torch.Tensor.native_lt = torch.Tensor.lt   # the hook keeps torch's own methods as native_*

        syft = _module("syft", Plan=object, local_worker=self.local_worker)
        syft.hook = types.SimpleNamespace(local_worker=self.local_worker)
        syft.PointerTensor = Share
        syft.MultiPointerTensor = type("MultiPointerTensor", (), {})
        from typing import Tuple

        exc = _module("syft.exceptions", Tuple=Tuple, sy=syft)
        _extract(f"{SYFT}/exceptions.py", ["EmptyCryptoPrimitiveStoreError"], exc.__dict__)
        _module("syft.generic")
        _module("syft.generic.utils", allow_command=lambda f: f, remote=_remote, memorize=lambda f: f)
        _module("syft.generic.abstract")
        _module("syft.generic.abstract.tensor", AbstractTensor=AbstractTensor)
        _module("syft.generic.frameworks")
        _module("syft.generic.frameworks.types", FrameworkTensor=torch.Tensor)
        hook = _module("syft.generic.frameworks.hook")
        hook.hook_args = _HookArgs()
        sys.modules["syft.generic.frameworks.hook.hook_args"] = hook.hook_args
        _load("syft.generic.frameworks.overload", f"{SYFT}/generic/frameworks/overload.py")
        _module("syft.generic.pointers")
        _module("syft.generic.pointers.multi_pointer", MultiPointerTensor=syft.MultiPointerTensor)
        _module("syft.workers")
        _module("syft.workers.abstract", AbstractWorker=Worker)
        _module("syft.workers.websocket_client", WebsocketClientWorker=type("WebsocketClientWorker", (), {}))
        for pb, cls in (("additive_shared_pb2", "AdditiveSharingTensor"), ("precision_pb2", "FixedPrecisionTensor")):
            for i in range(1, 7):
                _module(".".join(["syft_proto", "frameworks", "torch", "tensors", "interpreters", "v1"][:i]))
            _module(f"syft_proto.frameworks.torch.tensors.interpreters.v1.{pb}", **{cls: object})
        shaloop = _module("shaloop")
        shaloop.sha256_loop_func = lambda x, out: out.__setitem__(Ellipsis, S.sha_loop(np.ascontiguousarray(x), 256))
        shaloop.sha512_loop_func = lambda x, out: out.__setitem__(Ellipsis, S.sha_loop(np.ascontiguousarray(x), 512))

        fw = _module("syft.frameworks")
        fwt = _module("syft.frameworks.torch")
        syft.frameworks, fw.torch = fw, fwt
        mpc = _load("syft.frameworks.torch.mpc", f"{SYFT}/frameworks/torch/mpc/__init__.py")
        mpc.__path__ = []
        fwt.mpc = mpc
        mpc.fss = _load("syft.frameworks.torch.mpc.fss", f"{SYFT}/frameworks/torch/mpc/fss.py", _fss_shims)
        mpc.fss.multiprocessing = _SerialMP
        mpc.beaver = _load("syft.frameworks.torch.mpc.beaver", f"{SYFT}/frameworks/torch/mpc/beaver.py")
        mpc.primitives = _load("syft.frameworks.torch.mpc.primitives", f"{SYFT}/frameworks/torch/mpc/primitives.py")
        mpc.spdz = _load("syft.frameworks.torch.mpc.spdz", f"{SYFT}/frameworks/torch/mpc/spdz.py")
        mpc.spdz.multiprocessing = _SerialMP
        mpc.securenn = _module("syft.frameworks.torch.mpc.securenn")
        _module("syft.frameworks.torch.tensors")
        _module("syft.frameworks.torch.tensors.interpreters")
        a = _load("syft.frameworks.torch.tensors.interpreters.additive_shared",
                  f"{SYFT}/frameworks/torch/tensors/interpreters/additive_shared.py")
        syft.AdditiveSharingTensor = a.AdditiveSharingTensor
        nnpkg = _module("syft.frameworks.torch.nn", nn=types.SimpleNamespace())
        p = _load("syft.frameworks.torch.tensors.interpreters.precision",
                  f"{SYFT}/frameworks/torch/tensors/interpreters/precision.py")
        syft.FixedPrecisionTensor = p.FixedPrecisionTensor
        self.F = _load("syft.frameworks.torch.nn.functional", f"{SYFT}/frameworks/torch/nn/functional.py")
        nnpkg.functional = self.F
        self.syft, self.mpc = syft, mpc
        self.AST, self.FPT = a.AdditiveSharingTensor, p.FixedPrecisionTensor
        for w in self.workers.values():
            w.crypto_store = mpc.primitives.PrimitiveStorage(owner=w)

        # fresh sharings: record share 0 of every generate_shares call that is not part of build_triple
        orig_gen = a.AdditiveSharingTensor.generate_shares
        in_triple = {"on": False}

        def generate_shares(self_, secret, n_workers, random_type):
            shares = orig_gen(self_, secret, n_workers, random_type)
            if not in_triple["on"]:
                rt.log.append(("mask", shares[0].numpy().copy()))
            return shares

        a.AdditiveSharingTensor.generate_shares = generate_shares
        orig_bt = mpc.beaver.build_triple

        def build_triple(*args, **kw):
            in_triple["on"] = True
            try:
                return orig_bt(*args, **kw)
            finally:
                in_triple["on"] = False

        mpc.primitives.build_triple = build_triple
        mpc.beaver.build_triple = build_triple

    # ---- recording of the crypto provider's output, in request order ---------------------------------
    def record(self, worker, payload):
        j = self.parties.index(worker)
        for op, prims in payload.items():
            slot = self._pending.setdefault(op, {})
            slot[j] = prims
            if len(slot) < 2:
                continue
            p0, p1 = slot[0], slot[1]
            del self._pending[op]
            if op in ("mul", "matmul"):
                for (cfg0, sh0), (cfg1, sh1) in zip(p0, p1):
                    assert cfg0 == cfg1 and sh0[0].shape[0] == 1, "one instance per request (spdz.py:33,86)"
                    self.log.append(("triple", op, [tuple(t[0].numpy().copy() for t in sh0),
                                                    tuple(t[0].numpy().copy() for t in sh1)]))
            elif op == "fss_comp":
                a0, s00, *cw0 = p0
                a1, s01, *cw1 = p1
                alpha = (a0.astype(np.uint64) + a1.astype(np.uint64)) % np.uint64(2 ** 32)
                self.log.append(("dif", int(alpha.shape[-1]), alpha.astype(np.int64), np.stack([s00, s01]).view(np.int64),
                                 a1.astype(np.int64)))
                self.last_fss = (p0, p1)
            else:
                raise AssertionError(op)

    # ---- building FPT > AST chains the way inference.py does (fix_precision().share(...)) -------------
    def fix_share(self, x, precision_fractional=16, base=10, protocol="fss"):
        """tensor.fix_precision(precision_fractional=...).share(*parties, crypto_provider=..., protocol="fss")
        (inference.py:279-300; native.py fix_prec / precision.py:117-132 / :910-940)."""
        fpt = self.FPT(owner=self.local_worker, base=base, precision_fractional=precision_fractional, dtype="long")
        fpt = fpt.on(x.clone(), wrap=False).fix_precision()
        return fpt.share(*self.parties, protocol=protocol, crypto_provider=self.local_worker)

    @staticmethod
    def shares_of(fpt):
        ast_ = fpt.child if not isinstance(fpt.child, dict) else fpt
        return [s.detach().as_subclass(torch.Tensor).numpy().copy() for s in ast_.child.values()]

    def decode(self, fpt):
        """.get().float_precision() (additive_shared.py:287-301, precision.py:134-144)."""
        plain = self.FPT(**fpt.get_class_attributes()).on(fpt.child.get().as_subclass(torch.Tensor), wrap=False)
        return plain.float_precision()

    # ---- driving torchlib/models.py on FPT > AST chains ---------------------------------------------------
    def share_model(self, model, precision_fractional=16):
        """model.fix_precision(...).share(*workers, ...) (inference.py:279-287): TorchHook's module_fix_precision_ /
        module_share_ walk `parameters()` and THEN `buffers()` (syft/frameworks/torch/hook/hook.py:624-632,738-765)."""
        slots = []
        for it in ("named_parameters", "named_buffers"):
            for name, _ in getattr(model, it)():
                mod, leaf = model, name
                while "." in leaf:
                    head, leaf = leaf.split(".", 1)
                    mod = getattr(mod, head)
                slots.append((name, mod._parameters if it == "named_parameters" else mod._buffers, leaf))
        for name, store, leaf in slots:
            t = store[leaf].detach()
            if t.dim() == 0:
                # num_batches_tracked: generate_shares builds LongTensor(torch.Size([])) = an EMPTY tensor
                # (additive_shared.py:352): nothing is drawn and the buffer is never read in eval mode
                continue
            store[leaf] = self.fix_share(t.float(), precision_fractional)
        return model

    def hooked(self):
        """What TorchHook does for the six functionals the ResNet forward calls: a call whose first argument is a
        FixedPrecisionTensor goes to syft/frameworks/torch/nn/functional.py (the registry
        syft/frameworks/torch/nn/__init__.py installs; relu -> AdditiveSharingTensor.relu,
        additive_shared.py:896-899; linear -> torch-1.4's Python F.linear = torch.addmm(bias, input, weight.t())
        -> precision.py:822-825)."""
        import contextlib

        import torch.nn.functional as TF

        rt, F, FPT = self, self.F, self.FPT

        def relu(x, inplace=False):
            return FPT(**x.get_class_attributes()).on(x.child.relu(), wrap=False)

        def linear(x, weight, bias=None):
            assert x.dim() == 2 and bias is not None
            return FPT.torch.addmm(bias, x, weight.t())

        table = {"conv2d": F.conv2d, "batch_norm": F.batch_norm, "relu": relu, "max_pool2d": F.max_pool2d,
                 "avg_pool2d": F.avg_pool2d, "linear": linear}

        @contextlib.contextmanager
        def ctx():
            saved = {n: getattr(TF, n) for n in table}
            saved_flatten = torch.flatten

            def route(name):
                def f(*args, **kw):
                    return (table[name] if isinstance(args[0], FPT) else saved[name])(*args, **kw)

                return f

            for n in table:
                setattr(TF, n, route(n))
            torch.flatten = lambda x, *a, **k: (x.flatten(*a, **k) if isinstance(x, FPT) else saved_flatten(x, *a, **k))
            try:
                yield rt
            finally:
                for n in table:
                    setattr(TF, n, saved[n])
                torch.flatten = saved_flatten

        return ctx()


class RemoteTensor:
    """A float tensor that lives on a training worker, as the orchestrator sees it (a PointerTensor wrapper): every
    method runs "there"; `.get()` brings the value home."""

    def __init__(self, rt, value):
        self.rt, self.value = rt, value

    @property
    def shape(self):
        return self.value.shape

    @property
    def data(self):
        return self

    def copy(self):
        return RemoteTensor(self.rt, self.value.clone())

    def __mul__(self, w):
        return RemoteTensor(self.rt, self.value * w)

    def get(self):
        return self.value

    def fix_prec(self, precision_fractional=3, **kw):
        fpt = self.rt.FPT(owner=self.rt.local_worker, base=10, precision_fractional=precision_fractional, dtype="long")
        return RemoteTensor(self.rt, fpt.on(self.value.clone(), wrap=False).fix_precision())

    def share(self, *owners, crypto_provider=None, protocol="snn", **kw):
        return RemoteTensor(self.rt, self.value.share(*owners, protocol=protocol, crypto_provider=crypto_provider))


class RemoteModel:
    def __init__(self, rt, state_dict):
        from collections import OrderedDict

        self._sd = OrderedDict((k, RemoteTensor(rt, v)) for k, v in state_dict.items())

    def state_dict(self):
        return self._sd


def torch_namespace(rt):
    """The `torch` the extracted aggregation() sees: stack / sum routed like TorchHook routes them for
    FixedPrecisionTensor arguments (AdditiveSharingTensor.torch.stack, additive_shared.py:813-818; sum is the
    per-share method)."""
    def stack(tensors, **kw):
        if isinstance(tensors[0], rt.FPT):
            ast_ = rt.AST.torch.stack([t.child for t in tensors], **kw)
            return rt.FPT(**tensors[0].get_class_attributes()).on(ast_, wrap=False)
        return torch.stack(tensors, **kw)

    def sum_(x, **kw):
        return x.sum(**kw) if isinstance(x, rt.FPT) else torch.sum(x, **kw)

    return types.SimpleNamespace(stack=stack, sum=sum_)


def load_reference_models():
    spec = importlib.util.spec_from_file_location("ref_models", f"{REF}/torchlib/models.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m
