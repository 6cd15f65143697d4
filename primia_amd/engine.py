"""Device-side ResNet-18 training engine: one federated client = one GPU.

Replaces what a PySyft worker executes natively for one plaintext training step
(SURVEY.md §8b "commands one worker receives per step"; reference call site
torchlib/utils.py:1168-1174): forward, loss, backward and the optimizer step, all as HIP kernels
called through the C ABI of libprimia_hip.so.  torch is used for device memory, streams and (in
primia_amd.fed) torch.distributed — never for arithmetic.

Memory layout (all on one GPU):
  flat   fp32 [P + B]   parameters in reference named_parameters() order (OIHW conv weights), then
                        the BatchNorm running_mean/var buffers — exactly the 11,187,651 elements
                        FedAvg exchanges (torchlib/utils.py:1000-1092)
  grads  fp32 [P]       same order as the parameter part of `flat`
  per conv: compute-dtype weight copies in the implicit-GEMM layouts (+ fp32 wgrad accumulator)
  activations / gradients: NHWC tensors in the compute dtype (bf16 for throughput, fp32 for parity)
"""
import os
from collections import OrderedDict

import torch

from . import _lib
from ._lib import ConvDesc, call, query
from .resnet_spec import NetSpec, bn_name, buffer_entries, init_state_dict, param_entries, resnet18_spec, state_dict_keys

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


class _Conv:
    def __init__(self, spec, desc, c_real):
        self.spec, self.desc, self.c_real = spec, desc, c_real


class ResNet18Engine:
    def __init__(self, batch_size, num_classes=3, in_channels=3, input_size=224, pooling="max",
                 dtype=torch.bfloat16, device="cuda:0", norm="batch", groups=32, options=None, share=None):
        """norm="batch": the reference model.  norm="group": GroupNorm(groups, C) in place of every
        BatchNorm (ResNet's `norm_layer` hook, torchlib/models.py:355) — the BN-free network the
        DP-SGD configuration needs (train.py:308).
        `options`: {name: value} overriding the schedule switches below (class attributes such as wgrad_group,
        wgrad_overlap, masked_acc, stem_bwd_fused, dp_keep ...) for THIS engine before its buffers are sized.  Nothing here
        reads the environment; the kernel library's own switches are `_lib.set_option` (primia_set_option).
        `share`: another engine of the same network whose PARAMETERS this one uses (see `sibling`)."""
        self._root = self if share is None else share._root
        self._options = dict(options or {})
        for k, v in self._options.items():
            if not hasattr(type(self), k):
                raise ValueError(f"unknown engine option {k!r}")
            setattr(self, k, v)
        if not torch.cuda.is_available():
            raise _lib.PrimiaError("ResNet18Engine needs a GPU (HIP kernels only, no CPU fallback)")
        _lib.lib()
        # every buffer below is sized from the library's *_bytes / *_slots queries, which read the process-wide option
        # table: remember its epoch — forward() refuses to run once it has moved (a later
        # primia_set_option("c64_blocks" | "lh2" | ...) would make kernels write more partials than were allocated)
        self._options_epoch = query("primia_options_epoch")
        self.spec: NetSpec = resnet18_spec(num_classes, in_channels, input_size, pooling)
        if norm not in ("batch", "group"):
            raise ValueError("norm must be 'batch' or 'group'")
        self.norm, self.groups = norm, int(groups)
        self.N = int(batch_size)
        self.dtype = dtype
        self.dt = _lib.dtype_code(dtype)
        self.device = torch.device(device)
        self.training = True
        dev = self.device

        # ---- flat arenas ---------------------------------------------------------------------
        self.p_entries = param_entries(self.spec)
        self.b_entries = buffer_entries(self.spec, norm)
        self.P = sum(int(torch.Size(s).numel()) for _, s in self.p_entries)
        self.B = sum(int(torch.Size(s).numel()) for _, s in self.b_entries)
        if share is None:
            self.flat = torch.zeros(self.P + self.B, dtype=torch.float32, device=dev)
            self._grads = torch.zeros(self.P, dtype=torch.float32, device=dev)   # (read through the `grads` property)
        else:       # the same arenas: a step of either engine moves the one set of weights
            if (share.spec.num_classes, share.spec.in_channels, share.spec.input_size, share.spec.pooling, share.norm,
                    share.dtype) != (num_classes, in_channels, input_size, pooling, norm, dtype):
                raise ValueError("a sibling engine must be the same network, norm and dtype as the engine it shares with")
            self.flat, self._grads = share.flat, share._grads
        self.views, self._gviews = {}, {}
        off = 0
        for k, s in self.p_entries:
            n = int(torch.Size(s).numel())
            self.views[k] = self.flat[off:off + n].view(s)
            self._gviews[k] = self._grads[off:off + n].view(s)
            off += n
        for k, s in self.b_entries:
            n = int(torch.Size(s).numel())
            self.views[k] = self.flat[off:off + n].view(s)
            off += n
        # BatchNorm's num_batches_tracked (only ever READ by torch when momentum is None, never by this network): a host
        # counter advanced by forward(); a captured hipGraph replays kernels, not Python, so a caller that replays steps
        # (bench.py) and then exports state_dict() must add its replay count itself (note_replayed_steps).
        self.num_batches_tracked = ({bn_name(c.name): 0 for c in self.spec.convs} if share is None
                                    else share.num_batches_tracked)
        if share is None:
            self._opt_state = None  # Adam moments, created lazily
            self._opt_steps = 0

        # ---- conv descriptors, weight copies --------------------------------------------------
        N = self.N
        self.convs = {}
        H = input_size
        stem = self.spec.stem
        if in_channels > 4:
            raise _lib.PrimiaError("stem supports in_channels <= 4")
        self.convs[stem.name] = _Conv(stem, ConvDesc.make(N, H, H, 4, 64, 7, 7, 2, 3), in_channels)
        H = self.convs[stem.name].desc.Ho  # 112
        self.stem_hw = H
        H = (H + 2 - 3) // 2 + 1           # 56 after the 3x3/2 pool
        self.pool_hw = H
        for blk in self.spec.blocks:
            c1 = blk.conv1
            d1 = ConvDesc.make(N, H, H, c1.cin, c1.cout, 3, 3, c1.stride, 1)
            self.convs[c1.name] = _Conv(c1, d1, c1.cin)
            if blk.down is not None:
                dd = ConvDesc.make(N, H, H, blk.down.cin, blk.down.cout, 1, 1, blk.down.stride, 0)
                self.convs[blk.down.name] = _Conv(blk.down, dd, blk.down.cin)
            H = d1.Ho
            c2 = blk.conv2
            self.convs[c2.name] = _Conv(c2, ConvDesc.make(N, H, H, c2.cin, c2.cout, 3, 3, 1, 1), c2.cin)
        self.final_hw = H
        if self.final_hw != input_size // 32:
            raise _lib.PrimiaError("input_size must be a multiple of 32")

        # fp32 weight-gradient accumulators (forward layout), one arena.  The layers served by the atomic-free
        # path (primia_conv2d_wgrad_ws OVERWRITES its accumulator) sit behind the ones that accumulate with
        # atomics, so that only the front part needs zeroing every step.
        acc_total = 0
        ws_need = {c.spec.name: query("primia_conv_wgrad_ws_bytes", c.desc, self.dt) for c in self.convs.values()}
        self._stem_ws_bytes = query("primia_stem_conv_wgrad_ws_bytes", N, input_size, input_size)
        for overwriting in (False, True):
            for c in self.convs.values():
                # (conv1's 64 KiB slab stays in the zeroed part: odd input sizes take its accumulate path)
                if (ws_need[c.spec.name] > 0 and c.spec.name != "conv1") != overwriting:
                    continue
                c.wfwd_n = query("primia_conv_wfwd_elems", c.desc)
                c.acc_off = acc_total
                acc_total += (c.wfwd_n + 3) // 4 * 4
            if not overwriting:
                self._acc_zero_n = acc_total
                # ... and nothing but conv1's slab may be in it: the padded-input stem kernels overwrite that one too
                self._acc_zero_only_stem = all(ws_need[c.spec.name] > 0 for c in self.convs.values() if c.spec.name != "conv1")
        self.dw_acc = torch.zeros(acc_total, dtype=torch.float32, device=dev)
        for c in self.convs.values():
            if share is not None:       # kernel-layout weight copies do not depend on the batch size: one set
                c.w_fwd, c.w_dgrad = share.convs[c.spec.name].w_fwd, share.convs[c.spec.name].w_dgrad
                c.acc = self.dw_acc[c.acc_off:c.acc_off + c.wfwd_n]
                continue
            c.w_fwd = torch.empty(c.wfwd_n, dtype=dtype, device=dev)
            c.w_dgrad = None
            if c.spec.name != stem.name:
                c.w_dgrad = torch.empty(query("primia_conv_wdgrad_elems", c.desc), dtype=dtype, device=dev)
            c.acc = self.dw_acc[c.acc_off:c.acc_off + c.wfwd_n]
        # a transition block's conv1 + downsample weight gradients as one launch (primia_conv2d_wgrad_pair_ws)
        self._pair_ws = {}
        for blk in self.spec.blocks:
            if blk.down is not None:
                self._pair_ws[blk.conv1.name] = query("primia_conv_wgrad_pair_ws_bytes", self.convs[blk.conv1.name].desc,
                                                      self.convs[blk.down.name].desc, self.dt)
        # workspace of the atomic-free weight-gradient path (primia_conv2d_wgrad_ws): the layers run one after
        # the other on one stream, so they share one buffer sized for the largest (38 MB at batch 256)
        # several layers of one shape in ONE weight-gradient launch (primia_conv2d_wgrad_group_ws): the 3x3 / stride-1
        # layers of a stage share a shape; a layer's (x, dy) is held back until its siblings' are ready
        self._wg_group = {}        # conv name -> (shape key, preferred group size)
        self._wg_held = {}         # shape key -> [(name, x, dy), ...]
        self._wg_count = {}        # shape key -> layers of that shape in the network
        self._wg_seen = {}         # shape key -> layers of that shape this backward pass has reached
        group_ws = [0]
        if dtype == torch.bfloat16 and self.wgrad_group:
            shapes = {}
            for c in self.spec.convs:
                d = self.convs[c.name].desc
                if d.R == 3 and d.stride == 1 and c.name != stem.name:
                    shapes.setdefault((d.H, d.W, d.C, d.K), []).append(c.name)
            for key, names in shapes.items():
                d = self.convs[names[0]].desc
                n = query("primia_conv_wgrad_group_size", d, len(names), self.dt)
                if n >= 2:
                    for nm in names:
                        self._wg_group[nm] = (key, n)
                    self._wg_count[key] = len(names)
                    group_ws.append(query("primia_conv_wgrad_group_ws_bytes", d, n, self.dt))
        ws_bytes = max(max(ws_need.values()), self._stem_ws_bytes, max(list(self._pair_ws.values()) + [0]), max(group_ws))
        self.wgrad_ws = torch.empty(max(ws_bytes, 16) // 4, dtype=torch.float32, device=dev) if ws_bytes > 0 else None
        self.wgrad_ws_bytes = ws_bytes

        # ---- activations ------------------------------------------------------------------------
        def act(hw, ch):
            return torch.empty(N * hw * hw, ch, dtype=dtype, device=dev)

        self.x0 = torch.empty(N * input_size * input_size, 4, dtype=dtype, device=dev)
        # bf16: the stem also keeps a spatially padded copy of the input for primia_stem_conv_fwd (3 zero rows
        # above / below, 3 zero columns left, 5 right; zeroed once, only the interior is rewritten)
        self.x0p = None
        if dtype == torch.bfloat16 and input_size % 32 == 0:
            self.x0p_dims = (input_size + 6, input_size + 8)
            self.x0p = torch.zeros(N * self.x0p_dims[0] * self.x0p_dims[1], 4, dtype=dtype, device=dev)
        self.t = {}  # named tensors
        t = self.t
        t["stem.y"] = act(self.stem_hw, 64)
        t["stem.z"] = act(self.stem_hw, 64)
        t["stem.dz"] = act(self.stem_hw, 64)
        t["stem.dy"] = act(self.stem_hw, 64)
        t["pool.out"] = act(self.pool_hw, 64)
        self.pool_argmax = torch.empty(N * self.pool_hw * self.pool_hw * 64, dtype=torch.uint8, device=dev)
        for blk in self.spec.blocks:
            p = blk.prefix
            d1 = self.convs[blk.conv1.name].desc
            hw, ch = d1.Ho, blk.conv1.cout
            for nm in ("y1", "a1", "y2", "out", "dy1", "da1", "dy2"):
                t[f"{p}.{nm}"] = act(hw, ch)
            if blk.down is not None:
                for nm in ("yd", "idn", "dyd"):
                    t[f"{p}.{nm}"] = act(hw, ch)
        # Gradient w.r.t. each block's output.  An identity block accumulates its input gradient in
        # place (masked residual gradient + conv1 dgrad), so its input-gradient buffer IS its own
        # `dout`; a projection block gets a fresh buffer of the input's shape.
        blocks = self.spec.blocks
        for i in range(len(blocks) - 1, -1, -1):
            blk = blocks[i]
            d1 = self.convs[blk.conv1.name].desc
            key = blk.prefix + ".dout"
            if key not in t:
                t[key] = act(d1.Ho, blk.conv1.cout)
            din = t[key] if blk.down is None else act(d1.H, blk.conv1.cin)
            t[(blocks[i - 1].prefix + ".dout") if i > 0 else "pool.dout"] = din
        # Fused BN statistics: the conv epilogue can accumulate the batch sums (primia_conv2d_fwd_stats).
        # Measured on MI355X at batch 256 it is a wash — the epilogue reduction costs the forward kernels
        # +0.38 ms/step, the separate statistics pass it removes costs 0.39 ms/step — so it stays off.
        self.fuse_stats = False
        # stem tail bn1 -> relu -> maxpool as ONE fused op in training (z = relu(bn(y)) is never written)
        self.fuse_stem = True
        self._stem_fused = False
        self.use_relu_masks = True   # residual layers: 1-bit ReLU masks for the backward passes (see _bn)
        self.relu_masks = {}
        self._stem_padded = False
        self.stat_slots = query("primia_conv_stat_slots")
        for c in self.spec.convs:   # slots the kernel serving this conv writes (per-block partials for layer1's)
            self.convs[c.name].stat_slots = (query("primia_conv_stat_slots_for", self.convs[c.name].desc, self.dt)
                                             if c.name != "conv1" else self.stat_slots)
        per = lambda c: self.convs[c.name].stat_slots * 2 * c.cout
        self.stat_sums = torch.zeros(sum(per(c) for c in self.spec.convs), dtype=torch.float32, device=dev)
        off = 0
        for c in self.spec.convs:
            self.convs[c.name].sums = self.stat_sums[off:off + per(c)]
            off += per(c)
        # Layers whose conv kernel emits the BatchNorm partial sums for free (deterministic per-block partials out
        # of the write-back phase): the statistics pass over their output is dropped.  Decided per layer,
        # independent of the global `fuse_stats` experiment above.
        self.free_stats = {c.name for c in self.spec.convs
                           if dtype == torch.bfloat16 and norm == "batch" and c.name != "conv1"
                           and query("primia_conv_stats_per_tile", self.convs[c.name].desc, self.dt) == 1}
        # (Two experiments lived here and were measured to lose, profiles/r01_negative_results.txt: atomically accumulated
        # batch sums for the late stages, and BatchNorm-backward sums emitted by the data-gradient kernels' write-backs —
        # 6.375 -> 6.449 ms per step.)
        # identity blocks: can conv1's accumulating data gradient apply bn2's ReLU mask to the old values itself?
        self.masked_acc_ok = {}
        if self.masked_acc:
            for blk in self.spec.blocks:
                if blk.down is None:
                    self.masked_acc_ok[blk.conv1.name] = query("primia_conv_dgrad_masked_acc_ok",
                                                               self.convs[blk.conv1.name].desc, self.dt) == 1
        self.save = {}
        for c in self.spec.convs:
            b = bn_name(c.name)
            self.save[b] = (torch.empty(c.cout, dtype=torch.float32, device=dev),
                            torch.empty(c.cout, dtype=torch.float32, device=dev))
        self.bn_ws_bytes = max(query("primia_bn_workspace_bytes", 1, 512), 1024 * 3 * 512 * 4)   # (pair backward: 3 sums)
        if norm == "group":
            self.bn_ws_bytes = max(self.bn_ws_bytes, query("primia_gn_workspace_bytes", N, 512, self.groups))
            # statistics are per (sample, group); per-sample affine gradients [N][C] per layer
            self.save = {bn_name(c.name): (torch.empty(N * self.groups, dtype=torch.float32, device=dev),
                                           torch.empty(N * self.groups, dtype=torch.float32, device=dev))
                         for c in self.spec.convs}
            self.ps_affine = {bn_name(c.name): (torch.empty(N, c.cout, dtype=torch.float32, device=dev),
                                                torch.empty(N, c.cout, dtype=torch.float32, device=dev))
                              for c in self.spec.convs}
            self.ones_n = torch.ones(N, dtype=torch.float32, device=dev)
        self.bn_ws = torch.zeros(self.bn_ws_bytes, dtype=torch.uint8, device=dev)  # holds a completion counter
        # bn_inline: one flag word per 4 channels and BatchNorm layer (primia_bn_fwd_train_apply_inline), cleared once per step
        self.bn_flags = torch.zeros(len(self.spec.convs) * 128, dtype=torch.int32, device=dev)
        self._bn_flag_of = {bn_name(c.name): i * 128 for i, c in enumerate(self.spec.convs)}
        self.dp = None  # set by dp_backward: {"wgrads": [...]} defers the weight gradients
        self.feat = torch.empty(N, 512, dtype=torch.float32, device=dev)
        self.dfeat = torch.empty(N, 512, dtype=torch.float32, device=dev)
        self.logits = torch.empty(N, num_classes, dtype=torch.float32, device=dev)
        self.dlogits = torch.empty(N, num_classes, dtype=torch.float32, device=dev)
        self.loss = torch.zeros(1, dtype=torch.float32, device=dev)
        self.class_weight = None  # optional fp32 [classes] device tensor (CrossEntropyLoss weight)

    # ------------------------------------------------------------------------------------------
    # state
    # ------------------------------------------------------------------------------------------
    def init_weights(self):
        """Draw fresh weights from the global torch RNG the way the reference constructor does."""
        self.load_state_dict(init_state_dict(self.spec, self.norm))

    def state_dict(self):
        """Reference-compatible state dict (CPU tensors, OIHW, 122 keys incl. num_batches_tracked)."""
        host = self.flat.detach().cpu()
        sd = OrderedDict()
        off = {}
        o = 0
        for k, s in self.p_entries + self.b_entries:
            n = int(torch.Size(s).numel())
            off[k] = (o, n, s)
            o += n
        for k in state_dict_keys(self.spec, self.norm):
            if k.endswith("num_batches_tracked"):
                sd[k] = torch.tensor(self.num_batches_tracked[k.rsplit(".", 1)[0]], dtype=torch.long)
            else:
                o, n, s = off[k]
                sd[k] = host[o:o + n].view(s).clone()
        return sd

    def load_state_dict(self, sd):
        keys = state_dict_keys(self.spec, self.norm)
        missing = [k for k in keys if k not in sd]
        extra = [k for k in sd if k not in keys]
        if missing or extra:
            raise KeyError(f"state_dict mismatch: missing {missing[:3]} unexpected {extra[:3]}")
        for k in keys:
            if k.endswith("num_batches_tracked"):
                self.num_batches_tracked[k.rsplit(".", 1)[0]] = int(sd[k])
                continue
            v = sd[k]
            if tuple(v.shape) != tuple(self.views[k].shape):
                raise ValueError(f"shape mismatch for {k}: {tuple(v.shape)} vs {tuple(self.views[k].shape)}")
            self.views[k].copy_(v.to(torch.float32))
        self.refresh_weights()

    def _many_args(self):
        """Host-side argument arrays of the batched weight-refresh / wgrad-finalize calls (built once)."""
        if getattr(self, "_many", None) is None:
            import ctypes

            cs = list(self.convs.values())
            n = len(cs)
            vp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
            self._many = dict(
                n=n, descs=(ConvDesc * n)(*[c.desc for c in cs]), creal=(ctypes.c_int * n)(*[c.c_real for c in cs]),
                w=(ctypes.c_void_p * n)(*[vp(self.views[c.spec.name + ".weight"]) for c in cs]),
                wf=(ctypes.c_void_p * n)(*[vp(c.w_fwd) for c in cs]),
                wd=(ctypes.c_void_p * n)(*[vp(c.w_dgrad) for c in cs]),
                acc=(ctypes.c_void_p * n)(*[vp(c.acc) for c in cs]),
                gw=(ctypes.c_void_p * n)(*[vp(self._gviews[c.spec.name + ".weight"]) for c in cs]))
        return self._many

    # Adam's moments and step count belong to the PARAMETERS: siblings read and write the root engine's
    @property
    def opt_state(self):
        return self._root._opt_state

    @opt_state.setter
    def opt_state(self, v):
        self._root._opt_state = v

    @property
    def opt_steps(self):
        return self._root._opt_steps

    @opt_steps.setter
    def opt_steps(self, v):
        self._root._opt_steps = v

    @property
    def gviews(self):
        """{name: view of the gradient arena}.  Reading it finishes a deferred conv-gradient pass first (fuse_sgd_tail leaves
        the conv gradients in their accumulators between loss_backward() and step()), like `grads`."""
        self.materialize_grads()
        return self._gviews

    max_siblings = 6      # batch sizes kept besides the root's (each owns its activations and workspaces)

    def sibling(self, batch_size):
        """An engine for ANOTHER batch size on the SAME parameters, gradients, running statistics, kernel-layout weight
        copies and optimizer state (activations and workspaces are its own): forward / loss_backward / sgd_step /
        adam_step on either one train the one model.  The reference's local training loop needs it — MixUp halves a batch
        with probability mixup_prob (torchlib/utils.py:1262-1267, :336-347), so consecutive steps see B or B / 2 samples."""
        n = int(batch_size)
        if n == self.N:
            return self
        if n == self._root.N:         # (asked of a sibling: the root itself, not a second full-size engine)
            self._root.train(self.training)
            return self._root
        sib = self._root.__dict__.setdefault("_siblings", {})
        if n in sib:
            sib[n] = sib.pop(n)       # (a hit moves the entry to the back of the dict: the front is the least recently USED)
        elif len(sib) >= self.max_siblings:
            # activations and workspaces of an engine are ~12 MB per image: keep the most recently used few batch sizes only
            # (MixUp halves, the ragged final batch of train / validation loaders); the freed blocks stay in torch's
            # caching allocator for the engine that replaces them
            del sib[next(iter(sib))]
        if n not in sib:
            sib[n] = ResNet18Engine(n, self.spec.num_classes, self.spec.in_channels, self.spec.input_size,
                                    self.spec.pooling, dtype=self.dtype, device=self.device, norm=self.norm,
                                    groups=self.groups, options=self._root._options, share=self._root)
        # what the host sets on the root after construction travels with every call: the optimizer's fused tail and —
        # above all — the DP-SGD parameters (a halved or ragged batch must be clipped and noised like every other)
        for attr in ("fuse_sgd_tail", "dp_params", "class_weight"):
            setattr(sib[n], attr, getattr(self._root, attr))
        sib[n].train(self.training)
        return sib[n]

    def refresh_weights(self):
        """fp32 master (OIHW) -> compute-dtype implicit-GEMM copies; call after any change of `flat`."""
        m = self._many_args()
        call("primia_conv_weight_prepare_many", m["descs"], m["creal"], m["w"], m["wf"], m["wd"], m["n"], self.dt)

    def train(self, mode=True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    # ------------------------------------------------------------------------------------------
    # forward
    # ------------------------------------------------------------------------------------------
    def _bn(self, conv_name, y, z, residual, relu):
        b = bn_name(conv_name)
        C = y.shape[1]
        M = y.shape[0]
        g, be = self.views[b + ".weight"], self.views[b + ".bias"]
        if self.norm == "group":  # identical in train and eval mode: no running statistics
            sm, si = self.save[b]
            if self.training and residual is not None and relu and self.use_relu_masks and self.gn_relu_masks:
                # residual layer: also write the 1-bit ReLU mask the backward passes read instead of z
                if b not in self.relu_masks:
                    self.relu_masks[b] = torch.empty(y.numel() * y.element_size() // 16, dtype=torch.uint8, device=y.device)
                call("primia_gn_fwd_mask", y, residual, z, self.relu_masks[b], g, be, sm, si, self.N, M // self.N, C,
                     self.groups, BN_EPS, self.bn_ws, self.bn_ws_bytes, self.dt)
                return
            call("primia_gn_fwd", y, residual, z, g, be, sm, si, self.N, M // self.N, C, self.groups, BN_EPS, int(relu),
                 self.bn_ws, self.bn_ws_bytes, self.dt)
            return
        rm, rv = self.views[b + ".running_mean"], self.views[b + ".running_var"]
        if self.training and residual is not None and relu and self.use_relu_masks:
            # residual layer: also write the 1-bit ReLU mask the backward passes read instead of z
            sm, si = self.save[b]
            if b not in self.relu_masks:
                self.relu_masks[b] = torch.empty(y.numel() * y.element_size() // 16, dtype=torch.uint8, device=y.device)
            have_sums = self.fuse_stats or conv_name in self.free_stats
            if have_sums and self.bn_inline:
                fl = self.bn_flags[self._bn_flag_of[b]:self._bn_flag_of[b] + 128]
                call("primia_bn_fwd_train_apply_inline", y, residual, z, self.relu_masks[b], g, be, rm, rv, sm, si,
                     self.convs[conv_name].sums, self.convs[conv_name].stat_slots, M, C, BN_EPS, BN_MOMENTUM, 1, fl, self.dt)
                self.num_batches_tracked[b] += 1
                return
            call("primia_bn_fwd_train_mask", y, residual, z, self.relu_masks[b], g, be, rm, rv, sm, si,
                 self.convs[conv_name].sums if have_sums else None, self.convs[conv_name].stat_slots if have_sums else 0,
                 M, C, BN_EPS, BN_MOMENTUM, self.bn_ws, self.bn_ws_bytes, self.dt)
            self.num_batches_tracked[b] += 1
        elif self.training:
            sm, si = self.save[b]
            if (self.fuse_stats or conv_name in self.free_stats) and self.bn_inline:
                fl = self.bn_flags[self._bn_flag_of[b]:self._bn_flag_of[b] + 128]
                call("primia_bn_fwd_train_apply_inline", y, residual, z, None, g, be, rm, rv, sm, si,
                     self.convs[conv_name].sums, self.convs[conv_name].stat_slots, M, C, BN_EPS, BN_MOMENTUM, int(relu), fl,
                     self.dt)
            elif self.fuse_stats or conv_name in self.free_stats:
                call("primia_bn_fwd_train_from_sums", y, residual, z, g, be, rm, rv, sm, si,
                     self.convs[conv_name].sums, self.convs[conv_name].stat_slots, M, C, BN_EPS, BN_MOMENTUM,
                     int(relu), self.dt)
            else:
                call("primia_bn_fwd_train", y, residual, z, g, be, rm, rv, sm, si, M, C, BN_EPS, BN_MOMENTUM,
                     int(relu), self.bn_ws, self.bn_ws_bytes, self.dt)
            self.num_batches_tracked[b] += 1
        else:
            call("primia_bn_fwd_eval", y, residual, z, g, be, rm, rv, M, C, BN_EPS, int(relu), self.dt)

    # Optional per-launch timing of the convolution kernels (bench.py roofline leg): when
    # `self.prof` is a list, every conv launch is bracketed by events on the launch stream.
    prof = None
    # Optional copies of buffers the backward pass consumes IN PLACE (tests): when `self.taps` is a dict, backward() stores
    # "<block>.dout_in" — the gradient w.r.t. the block's output as its bn2 backward pass reads it, before conv1's
    # accumulating data gradient turns the same buffer into the gradient of the block in front.
    taps = None

    # the BatchNorm finalize launch folded into the apply kernel (primia_bn_fwd_train_apply_inline): bit-identical, and
    # SLOWER — 4.745 -> 4.79 ms per step, same box: 2,048 apply blocks each fetching 2C constants through agent-scope loads
    # contend at the coherence point the way the producers' atomics did (profiles/r05_bn_finalize_atomics.txt).  Off.
    bn_inline = False
    # conv2's data gradient also forms the backward sums of bn1 (primia_conv2d_dgrad_bnsums + primia_bn_relu_bwd_from_sums):
    # the reduction pass over (y1, da1) is dropped where the linear-halo kernels serve conv2
    dgrad_bnsums = True
    # ... and the transition block's paired data gradient the sums of the residual BatchNorm in front of the block
    # (primia_conv2d_dgrad_pair_bnsums: conv_s2lh_kernel for 64-channel dx, conv_igemm_kernel's parity-class walk for layer3.0 /
    # layer4.0).  The 64-channel one alone measured neutral (4.770 vs 4.770 ms); with the two igemm launches, whose LDS
    # write-back loop already walks 16-byte chunks of whole pixel rows, 4.882 -> 4.862 ms (8 alternating rounds, one box).
    pair_bnsums = True
    # ... and the ACCUMULATING data gradient of a 64 -> 64 identity block's conv1 the sums of the BatchNorm whose output gradient
    # it completes: layer1.0's bn2 (layer1.1.conv1) and the stem's bn1 through the max-pool (layer1.0.conv1) —
    # primia_conv2d_dgrad_masked_acc_bnsums; the two most expensive reduction passes of the step (43 + 48 us)
    acc_bnsums = True
    # ... and the head's backward pass the sums of the last block's bn2 (primia_head_bwd_bnsums)
    head_bnsums = True
    # conv1 + downsample data gradients of a transition block in one pass (primia_conv2d_dgrad_pair)
    pair_dgrad = True
    # identity blocks: conv1's accumulating data gradient applies bn2's ReLU mask to the old values itself
    masked_acc = True
    # transition blocks: bn2's and the downsample BatchNorm's backward passes as one (primia_bn_bwd_pair)
    bn_pair = True
    # order of a layer's two gradient kernels: weight gradient first, so that the BatchNorm backward pass that follows
    # the data gradient reads it while it is still in the Infinity Cache (6.42 -> 6.39 ms per step; wgrad_first = False
    # restores the other order)
    wgrad_first = True

    @staticmethod
    def _macs(c):
        d = c.desc
        return d.N * d.Ho * d.Wo * d.K * c.c_real * d.R * d.S

    def _timed(self, kind, c, fn, extra_macs=0):
        if self.prof is None:
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        self.prof.append((kind, c.spec.name, 2.0 * (self._macs(c) + extra_macs), e0, e1))

    def _conv_fwd(self, name, x, y):
        c = self.convs[name]
        if self.training and (self.fuse_stats or name in self.free_stats):
            self._timed("fwd", c, lambda: call("primia_conv2d_fwd_stats", c.desc, x, c.w_fwd, y, c.sums, self.dt))
        else:
            self._timed("fwd", c, lambda: call("primia_conv2d_fwd", c.desc, x, c.w_fwd, y, self.dt))

    fwd_pair = True
    # eval mode, BatchNorm + max pooling: the stem head as one kernel (primia_stem_conv_pool_eval).  One block per image —
    # from this batch size on it beats the three-kernel chain (178 us for any batch up to 256; the chain: 1.4 us per image)
    stem_eval_one_pass_min_batch = 160

    def _stem_eval_one_pass(self):
        if self.training or self.norm != "batch" or self.spec.pooling != "max" or self.prof is not None:
            return False
        if self.N < self.stem_eval_one_pass_min_batch:
            return False
        S = self.spec.input_size
        return query("primia_stem_conv_pool_ok", self.N, S, S, self.dt) == 1

    def _conv_fwd_pair(self, blk, x, y1, yd):
        """conv1 + downsample of a transition block as one launch; False where the library does not serve the pair."""
        c1, cd = self.convs[blk.conv1.name], self.convs[blk.down.name]
        if getattr(c1, "pair_ok", None) is None:
            c1.pair_ok = self.fwd_pair and query("primia_conv_fwd_pair_ok", c1.desc, cd.desc, self.dt) == 1
        if not c1.pair_ok:
            return False
        stats = self.training and self.norm == "batch"
        s1 = c1.sums if stats and (self.fuse_stats or blk.conv1.name in self.free_stats) else None
        sd = cd.sums if stats and (self.fuse_stats or blk.down.name in self.free_stats) else None
        if stats and (s1 is None or sd is None):
            return False        # (a layer whose statistics come from a separate pass: keep the single launches)
        self._timed("fwd", c1, lambda: call("primia_conv2d_fwd_stats_pair", c1.desc, x, c1.w_fwd, y1, s1, cd.desc, cd.w_fwd,
                                            yd, sd, self.dt), extra_macs=self._macs(cd))
        return True

    def forward(self, x_nchw):
        """x_nchw: fp32 [N, in_channels, S, S] on this GPU.  Returns fp32 logits [N, classes]."""
        N, S, t = self.N, self.spec.input_size, self.t
        if tuple(x_nchw.shape) != (N, self.spec.in_channels, S, S) or x_nchw.dtype != torch.float32:
            raise ValueError(f"expected fp32 input {(N, self.spec.in_channels, S, S)}, got {tuple(x_nchw.shape)}")
        x_nchw = x_nchw.contiguous()
        if query("primia_options_epoch") != self._options_epoch:
            raise _lib.PrimiaError("library options changed (primia_set_option) after this engine sized its buffers: "
                                   "set options first, then construct the engine")
        if self.training and self.fuse_stats:
            self.stat_sums.zero_()
        if self.training and self.bn_inline and self.norm == "batch":
            self.bn_flags.zero_()
        self._stem_padded = self.x0p is not None and not (self.training and self.fuse_stats)
        # (the unpadded copy is read only where the halo kernels on the padded one do not serve the shape)
        self._x0_valid = not self._stem_padded or (self.norm == "group" and self._stem_ws_bytes <= 0)
        if self._x0_valid:
            call("primia_nchw_to_nhwc", x_nchw, self.x0, N, self.spec.in_channels, S, S, 4, self.dt)
        stem_done = False
        if self._stem_padded:
            call("primia_nchw_to_nhwc_padded", x_nchw, self.x0p, N, self.spec.in_channels, S, S, 4, 3, 3,
                 self.x0p_dims[0], self.x0p_dims[1], self.dt)
            c = self.convs["conv1"]
            if self.training and self.norm == "batch":   # + bn1's per-block partial sums, for free
                if getattr(self, "_stem_sums", None) is None:
                    self._stem_slots = query("primia_stem_conv_stat_slots", N, S, S)
                    self._stem_sums = torch.zeros(self._stem_slots, 2, 64, dtype=torch.float32, device=self.device)
                self._timed("fwd", c, lambda: call("primia_stem_conv_fwd_stats", self.x0p, c.w_fwd, t["stem.y"],
                                                   self._stem_sums, N, S, S, self.dt))
                self._stem_has_sums = True
            elif self._stem_eval_one_pass():
                # eval mode: conv1 -> bn1 -> relu -> maxpool as one pass over the input (neither stem tensor is written)
                call("primia_stem_conv_pool_eval", self.x0p, c.w_fwd, t["pool.out"], self.pool_argmax,
                     self.views["bn1.weight"], self.views["bn1.bias"], self.views["bn1.running_mean"],
                     self.views["bn1.running_var"], BN_EPS, N, S, S, self.dt)
                self._stem_has_sums = False
                stem_done = True
            else:
                self._timed("fwd", c, lambda: call("primia_stem_conv_fwd", self.x0p, c.w_fwd, t["stem.y"], N, S, S,
                                                   self.dt))
                self._stem_has_sums = False
        else:
            self._conv_fwd("conv1", self.x0, t["stem.y"])
        hw = self.stem_hw
        self._stem_fused = (self.fuse_stem and self.training and self.norm == "batch" and self.spec.pooling == "max"
                            and not self.fuse_stats)
        # GroupNorm: gn1 -> relu -> maxpool as one op each way (primia_gn_relu_maxpool_fwd / _bwd; _bwd wants even sizes)
        self._stem_fused_gn = (self.fuse_stem and self.gn_stem_fused and self.norm == "group"
                               and self.spec.pooling == "max" and hw % 2 == 0)
        if stem_done:
            pass
        elif self._stem_fused and getattr(self, "_stem_has_sums", False) and self._stem_padded:
            sm, si = self.save["bn1"]
            call("primia_bn_relu_maxpool_fwd_from_sums", t["stem.y"], t["pool.out"], self.pool_argmax,
                 self.views["bn1.weight"], self.views["bn1.bias"], self.views["bn1.running_mean"],
                 self.views["bn1.running_var"], sm, si, self._stem_sums, self._stem_slots, N, hw, hw, 64, BN_EPS,
                 BN_MOMENTUM, self.dt)
            self.num_batches_tracked["bn1"] += 1
        elif self._stem_fused:
            sm, si = self.save["bn1"]
            call("primia_bn_relu_maxpool_fwd", t["stem.y"], t["pool.out"], self.pool_argmax, self.views["bn1.weight"],
                 self.views["bn1.bias"], self.views["bn1.running_mean"], self.views["bn1.running_var"], sm, si, N, hw,
                 hw, 64, BN_EPS, BN_MOMENTUM, self.bn_ws, self.bn_ws_bytes, self.dt)
            self.num_batches_tracked["bn1"] += 1
        elif self._stem_fused_gn:
            sm, si = self.save["bn1"]
            call("primia_gn_relu_maxpool_fwd", t["stem.y"], t["pool.out"], self.pool_argmax, self.views["bn1.weight"],
                 self.views["bn1.bias"], sm, si, N, hw, hw, 64, self.groups, BN_EPS, self.bn_ws, self.bn_ws_bytes, self.dt)
        else:
            self._bn("conv1", t["stem.y"], t["stem.z"], None, True)
        if self._stem_fused or self._stem_fused_gn or stem_done:
            pass
        elif self.spec.pooling == "max":
            call("primia_maxpool3x3s2_fwd", t["stem.z"], t["pool.out"], self.pool_argmax, N, hw, hw, 64, self.dt)
        else:
            call("primia_avgpool3x3s2_fwd", t["stem.z"], t["pool.out"], N, hw, hw, 64, self.dt)
        x = t["pool.out"]
        for blk in self.spec.blocks:
            p = blk.prefix
            # transition block: conv1 and the downsample read the same x — one launch (primia_conv2d_fwd_stats_pair)
            down_done = blk.down is not None and self._conv_fwd_pair(blk, x, t[p + ".y1"], t[p + ".yd"])
            if not down_done:
                self._conv_fwd(blk.conv1.name, x, t[p + ".y1"])
            self._bn(blk.conv1.name, t[p + ".y1"], t[p + ".a1"], None, True)
            self._conv_fwd(blk.conv2.name, t[p + ".a1"], t[p + ".y2"])
            idn = x
            if blk.down is not None and self.training and self.norm == "batch" and self.use_relu_masks and self.bn_pair:
                # transition block: both BatchNorms in one apply pass (the downsample branch's output is never stored)
                if not down_done:
                    self._conv_fwd(blk.down.name, x, t[p + ".yd"])
                b2, bd = bn_name(blk.conv2.name), bn_name(blk.down.name)
                y2 = t[p + ".y2"]
                if b2 not in self.relu_masks:
                    self.relu_masks[b2] = torch.empty(y2.numel() * y2.element_size() // 16, dtype=torch.uint8,
                                                      device=y2.device)
                c2 = self.convs[blk.conv2.name]
                have = self.fuse_stats or blk.conv2.name in self.free_stats
                cd = self.convs[blk.down.name]
                have_d = self.fuse_stats or blk.down.name in self.free_stats
                (sm2, si2), (smd, sid) = self.save[b2], self.save[bd]
                call("primia_bn_fwd_train_pair", y2, t[p + ".yd"], t[p + ".out"], self.relu_masks[b2],
                     self.views[b2 + ".weight"], self.views[b2 + ".bias"], self.views[b2 + ".running_mean"],
                     self.views[b2 + ".running_var"], sm2, si2, c2.sums if have else None, c2.stat_slots if have else 0,
                     self.views[bd + ".weight"], self.views[bd + ".bias"], self.views[bd + ".running_mean"],
                     self.views[bd + ".running_var"], smd, sid, cd.sums if have_d else None, cd.stat_slots if have_d else 0,
                     y2.shape[0], y2.shape[1], BN_EPS, BN_MOMENTUM, self.bn_ws, self.bn_ws_bytes, self.dt)
                self.num_batches_tracked[b2] += 1
                self.num_batches_tracked[bd] += 1
                x = t[p + ".out"]
                continue
            if blk.down is not None:
                if not down_done:
                    self._conv_fwd(blk.down.name, x, t[p + ".yd"])
                self._bn(blk.down.name, t[p + ".yd"], t[p + ".idn"], None, False)
                idn = t[p + ".idn"]
            self._bn(blk.conv2.name, t[p + ".y2"], t[p + ".out"], idn, True)
            x = t[p + ".out"]
        hw = self.final_hw
        call("primia_head_fwd", x, self.views["fc.weight"], self.views["fc.bias"], self.feat, self.logits, N, hw * hw,
             512, self.spec.num_classes, self.dt)
        return self.logits

    __call__ = forward

    # ------------------------------------------------------------------------------------------
    # loss + backward
    # ------------------------------------------------------------------------------------------
    dp_params = None  # e.g. {"max_grad_norm": 1.0, "noise_multiplier": 1.3}: loss_backward() becomes DP-SGD

    def loss_backward(self, target, soft=False):
        """Cross entropy (hard int64 labels, or soft [N, classes] fp32 targets as in
        Cross_entropy_one_hot) + full backward into `grads`.  Returns the device loss scalar."""
        if self.dp_params is not None:
            if soft:
                raise _lib.PrimiaError("DP-SGD path takes hard labels")
            return self.dp_loss_backward(target, **self.dp_params)
        if soft:
            call("primia_xent_soft", self.logits, target, self.class_weight, self.loss, self.dlogits, self.N,
                 self.spec.num_classes)
        else:
            call("primia_xent_hard", self.logits, target, self.class_weight, self.loss, self.dlogits, self.N,
                 self.spec.num_classes)
        self.backward()
        return self.loss

    def _bn_bwd(self, conv_name, y, z, dz, dy, g_out, relu, keep_g=True, from_sums=None):
        """keep_g=False (residual layers with a 1-bit mask only): the masked gradient g is NOT written back over dz —
        the accumulating data gradient that consumes it applies the mask itself (primia_conv2d_dgrad_masked_acc)."""
        b = bn_name(conv_name)
        sm, si = self.save[b]
        if self.norm == "group":
            psg, psb = self.ps_affine[b]
            C = y.shape[1]
            if relu and g_out is None and self.gn_relu_recompute:
                # z = relu(gn(y)), no residual: the mask is recomputed from y, z is not read
                call("primia_gn_relu_bwd", y, dz, dy, self.views[b + ".weight"], self.views[b + ".bias"], sm, si, psg, psb,
                     self.N, y.shape[0] // self.N, C, self.groups, self.bn_ws, self.bn_ws_bytes, self.dt)
            elif relu and g_out is not None and b in self.relu_masks:
                # residual layer: mask bytes instead of z; keep_g = False: the masked gradient is not written either
                call("primia_gn_bwd_mask", y, self.relu_masks[b], dz, dy, g_out if keep_g else None,
                     self.views[b + ".weight"], sm, si, psg, psb, self.N, y.shape[0] // self.N, C, self.groups, self.bn_ws,
                     self.bn_ws_bytes, self.dt)
            else:
                call("primia_gn_bwd", y, z, dz, dy, g_out, self.views[b + ".weight"], sm, si, psg, psb, self.N,
                     y.shape[0] // self.N, C, self.groups, int(relu), self.bn_ws, self.bn_ws_bytes, self.dt)
            if self.dp is None:  # plain training: dgamma / dbeta = sum over samples
                call("primia_weighted_colsum", psg, self.ones_n, self._gviews[b + ".weight"], self.N, C)
                call("primia_weighted_colsum", psb, self.ones_n, self._gviews[b + ".bias"], self.N, C)
            return
        if relu and g_out is not None and b in self.relu_masks and from_sums is not None:
            call("primia_bn_bwd_mask_from_sums", y, self.relu_masks[b], dz, dy, g_out if keep_g else None,
                 self.views[b + ".weight"], sm, si, self._gviews[b + ".weight"], self._gviews[b + ".bias"], from_sums[0],
                 from_sums[1], y.shape[0], y.shape[1], self.dt)
            return
        if relu and g_out is not None and b in self.relu_masks:
            call("primia_bn_bwd_mask", y, self.relu_masks[b], dz, dy, g_out if keep_g else None,
                 self.views[b + ".weight"], sm, si,
                 self._gviews[b + ".weight"], self._gviews[b + ".bias"], y.shape[0], y.shape[1], self.bn_ws,
                 self.bn_ws_bytes, self.dt)
            return
        if relu and g_out is None:
            # z = relu(bn(y)), no residual: the mask is recomputed from y, z is not read
            call("primia_bn_relu_bwd", y, dz, dy, self.views[b + ".weight"], self.views[b + ".bias"], sm, si,
                 self._gviews[b + ".weight"], self._gviews[b + ".bias"], y.shape[0], y.shape[1], self.bn_ws,
                 self.bn_ws_bytes, self.dt)
            return
        call("primia_bn_bwd", y, z, dz, dy, g_out, self.views[b + ".weight"], sm, si, self._gviews[b + ".weight"],
             self._gviews[b + ".bias"], y.shape[0], y.shape[1], int(relu), self.bn_ws, self.bn_ws_bytes, self.dt)

    wgrad_pair = True
    # a stage's LAST same-shape layer does not wait for a group that can no longer fill (layer4 at batch 256: groups of 2 for 3
    # layers): its weight gradient runs as soon as its dy exists instead of at the end of the backward pass (round 6)
    wgrad_flush_last = True
    gn_relu_recompute = True
    # GroupNorm residual layers with the BatchNorm path's 1-bit ReLU masks (and its mask-applying accumulate dgrad)
    gn_relu_masks = True
    gn_ds_mask = True
    gn_stem_fused = True
    wgrad_group = True
    # (all 13 layers in ONE launch at the end of the backward pass was built and measured no better than the per-stage
    # groups — 5.16 vs 5.14 ms: by then every operand comes from HBM, while a stage's group still finds its newest dy in
    # the Infinity Cache — and is not kept: profiles/r03_negative_results.txt)

    def _wgrad_transition(self, blk, x, dy1, dyd):
        """conv1 and the downsample of a transition block: one launch where the library serves the pair."""
        c1, cd = self.convs[blk.conv1.name], self.convs[blk.down.name]
        if (self.wgrad_pair and self.dp is None and self.wgrad_ws is not None
                and self._pair_ws.get(blk.conv1.name, 0) > 0):
            self._on_wgrad_stream(lambda: self._timed(
                "wgrad", c1, lambda: call("primia_conv2d_wgrad_pair_ws", c1.desc, x, dy1, c1.acc, cd.desc, dyd, cd.acc,
                                          self.wgrad_ws, self.wgrad_ws_bytes, self.dt), extra_macs=self._macs(cd)), transition=True)
            return
        self._wgrad(blk.conv1.name, x, dy1)
        self._wgrad(blk.down.name, x, dyd)

    def _wgrad(self, name, x, dy):
        c = self.convs[name]
        if self.dp is not None:  # DP-SGD: weight gradients wait for the per-sample clip factors
            self.dp["wgrads"].append((name, x, dy))
            return
        if self.wgrad_ws is not None and name in self._wg_group:
            key, n = self._wg_group[name]
            held = self._wg_held.setdefault(key, [])
            held.append((name, x, dy))
            self._wg_seen[key] = self._wg_seen.get(key, 0) + 1
            # a full group — or the stage's last layer (layer4 at batch 256: groups of 2 for 3 layers): it runs NOW, while its
            # dy is the newest tensor in the Infinity Cache, not at the end of the backward pass
            if len(held) == n or (self.wgrad_flush_last and self._wg_seen[key] == self._wg_count.get(key, 0)):
                self._flush_wgrad_group(key)
            return
        if self.wgrad_ws is not None:   # (one workspace: the weight gradients stay in order on whichever stream)
            self._on_wgrad_stream(lambda: self._timed("wgrad", c, lambda: call(
                "primia_conv2d_wgrad_ws", c.desc, x, dy, c.acc, self.wgrad_ws, self.wgrad_ws_bytes, self.dt)))
            return
        self._on_wgrad_stream(lambda: self._timed("wgrad", c, lambda: call("primia_conv2d_wgrad", c.desc, x, dy, c.acc,
                                                                             self.dt)))

    # Weight gradients are leaves of the backward graph (nothing reads them before the finalize) and every layer
    # has its own dy buffer, so they can run on a second stream.  Overlapping them with EVERYTHING was slower
    # (7.70 -> 7.91 ms: wgrad and dgrad tiles evict each other from the CUs).  With `wgrad_overlap` the schedule is
    # narrower: a layer's wgrad starts after its sibling dgrad and runs beside the BatchNorm backward chain that
    # follows (HBM-bound kernels and 7-us finalize launches); the next dgrad waits for it.  Also slower on
    # MI355X (7.10 ms serial -> 7.30 ms): kept as an option for other shapes, off by default.
    # wgrad_overlap = 3: see _on_wgrad_stream.
    # Re-measured in round 3 with the workspace kernels (wgrad_overlap = 1: the schedule above; = 2: the weight
    # gradients free-running on the second stream until the finalize): see profiles/r03_negative_results.txt.
    wgrad_overlap = 0
    stem_bwd_fused = True

    def _on_wgrad_stream(self, fn, transition=False):
        if self.wgrad_overlap == 3 and not transition and self.prof is None:
            # mode 3 (round 6): ONLY a transition block's paired weight gradient leaves the main stream — it runs beside the
            # block's paired data gradient, two launches that each leave the matrix pipe idle 80 % of the time (short reduction
            # axes: their tiles are write-back and staging latency) — and every other weight gradient stays in line, behind it
            # (one workspace)
            self._join_wgrad_stream()
            return fn()
        if not self.wgrad_overlap or self.prof is not None:
            return fn()
        if getattr(self, "_wg_stream", None) is None:
            self._wg_stream = torch.cuda.Stream(device=self.device)
        main = torch.cuda.current_stream()
        self._wg_stream.wait_stream(main)        # dy (and the zeroed accumulators) are ready
        with torch.cuda.stream(self._wg_stream):
            fn()
        self._wg_pending = True

    def _join_wgrad_stream(self):
        if getattr(self, "_wg_pending", False):
            torch.cuda.current_stream().wait_stream(self._wg_stream)
            self._wg_pending = False

    def _head_bnsums_ok(self):
        # primia_head_bwd_bnsums: 256 threads must be a multiple of the 16-byte chunks of a 512-channel row
        return 256 % (512 // (4 if self.dtype == torch.float32 else 8)) == 0

    def _stem_bwd_fused_wanted(self):
        """The stem's backward tail runs as primia_bn_relu_maxpool_bwd (sums only) + primia_stem_bwd_fused."""
        if getattr(self, "_stem_bwd_fused_ok", None) is None:     # asked once: the library's own gate for this shape
            S0 = self.spec.input_size
            self._stem_bwd_fused_ok = query("primia_stem_bwd_fused_ok", self.N, S0, S0, self.dt) == 1
        return bool(self._stem_fused and self._stem_padded and self.dp is None and self.stem_bwd_fused
                    and self.wgrad_ws is not None and self._stem_bwd_fused_ok)

    def _dgrad_bnsums_slots(self, name):
        """Rows of the partial table conv `name`'s data gradient writes for the BatchNorm in front of it (0: not served)."""
        c = self.convs[name]
        if getattr(c, "bnsums_slots", None) is None:      # (a property of the layer: asked once, whatever self.dp is then)
            c.bnsums_slots = query("primia_conv_dgrad_bnsums_slots", c.desc, self.dt) if self.dtype == torch.bfloat16 else 0
        return c.bnsums_slots if self.dp is None else 0

    def _bwd_sums(self, name, slots, channels):
        tab = self.__dict__.setdefault("_bwd_sum_bufs", {})
        if name not in tab or tab[name].numel() != slots * 2 * channels:
            tab[name] = torch.empty(slots * 2 * channels, dtype=torch.float32, device=self.device)
        return tab[name]

    def _dgrad(self, name, dy, dx, accumulate, consumer=None, consumer_y=None):
        """Data gradient of conv `name` into dx (`consumer`, `consumer_y`: the conv whose BatchNorm backward reads dx
        next — kept in the signature for the callers; the kernels no longer form that BatchNorm's sums, see __init__)."""
        c = self.convs[name]
        if self.wgrad_overlap == 1:
            self._join_wgrad_stream()   # (narrow overlap) two MFMA-bound kernels never run side by side
        self._timed("dgrad", c,
                    lambda: call("primia_conv2d_dgrad", c.desc, dy, c.w_dgrad, dx, int(accumulate), self.dt))

    def backward(self):
        N, t = self.N, self.t
        nc = self.spec.num_classes
        if query("primia_options_epoch") != self._options_epoch:
            raise _lib.PrimiaError("library options changed (primia_set_option) after this engine sized its buffers")
        self._grads_pending = False      # (accumulators of an earlier pass that nobody consumed are overwritten now)
        self._wg_seen = {}
        self._dout_sums = {}             # block prefix -> (partials, slots): backward sums of its bn2 formed by the producer of dout
        if self.wgrad_ws is not None and self.dp is None:
            # the other layers' accumulators are overwritten — conv1's too where its weight gradient runs on the padded input
            # (primia_stem_bwd_fused ends in an ordered reduce that WRITES the slab): then the step has
            # no fill launch at all (tests: test_weight_gradient_accumulators_are_overwritten poisons the arena between steps)
            stem_overwrites = self._acc_zero_only_stem and self._stem_bwd_fused_wanted()
            if not stem_overwrites:
                self.dw_acc[:self._acc_zero_n].zero_()
        else:
            self.dw_acc.zero_()
        # (under DP-SGD fc.weight / fc.bias gradients are overwritten later from the clipped dlogits)
        call("primia_linear_bwd", self.feat, self.views["fc.weight"], self.dlogits, None,
             self._gviews["fc.weight"], self._gviews["fc.bias"], N, 512, nc)
        last = self.spec.blocks[-1].prefix
        hw = self.final_hw
        blocks = self.spec.blocks
        lb2 = bn_name(blocks[-1].conv2.name)
        if (self.head_bnsums and self.norm == "batch" and self.dp is None and blocks[-1].down is None and lb2 in self.relu_masks
                and self._head_bnsums_ok()):
            # ... forming the backward sums of the last block's bn2 on the way (one value per image and channel)
            sums = self._bwd_sums("head", N, 512)
            sml, sil = self.save[lb2]
            call("primia_head_bwd_bnsums", self.views["fc.weight"], self.dlogits, t[last + ".dout"], t[last + ".y2"],
                 self.relu_masks[lb2], sml, sil, sums, N, hw * hw, 512, nc, self.dt)
            self._dout_sums[last] = (sums, N)
        else:
            call("primia_head_bwd", self.views["fc.weight"], self.dlogits, t[last + ".dout"], N, hw * hw, 512, nc, self.dt)
        for i in range(len(blocks) - 1, -1, -1):
            blk = blocks[i]
            p = blk.prefix
            x_in = t[blocks[i - 1].prefix + ".out"] if i > 0 else t["pool.out"]
            dx_in = t[blocks[i - 1].prefix + ".dout"] if i > 0 else t["pool.dout"]
            dout = t[p + ".dout"]
            if self.taps is not None:
                self.taps[p + ".dout_in"] = dout.clone()
            # bn2 (+residual, relu): dy2, and the masked gradient g written back over dout — unless this is an identity
            # block whose conv1 data gradient can mask the old values itself (one tensor write less)
            b2 = bn_name(blk.conv2.name)
            masked_acc = (blk.down is None and self.masked_acc_ok.get(blk.conv1.name, False)
                          and b2 in self.relu_masks)
            # transition block: bn2 and the downsample BatchNorm share the incoming gradient -> ONE fused backward
            bn_pair = (blk.down is not None and self.pair_dgrad and self.bn_pair and self.norm == "batch"
                       and b2 in self.relu_masks)
            if bn_pair:
                bd = bn_name(blk.down.name)
                (sm2, si2), (smd, sid) = self.save[b2], self.save[bd]
                call("primia_bn_bwd_pair", t[p + ".y2"], t[p + ".yd"], dout, self.relu_masks[b2], t[p + ".dy2"],
                     t[p + ".dyd"], self.views[b2 + ".weight"], sm2, si2, self.views[bd + ".weight"], smd, sid,
                     self._gviews[b2 + ".weight"], self._gviews[b2 + ".bias"], self._gviews[bd + ".weight"],
                     self._gviews[bd + ".bias"], t[p + ".y2"].shape[0], t[p + ".y2"].shape[1], self.bn_ws,
                     self.bn_ws_bytes, self.dt)
            else:
                # GroupNorm transition block: the downsample's backward pass applies bn2's ReLU mask to dout itself
                # (primia_gn_bwd_mask on yd), so the masked gradient is not written here either
                gn_ds_mask = (blk.down is not None and self.norm == "group" and self.pair_dgrad and self.gn_ds_mask
                              and b2 in self.relu_masks)
                self._bn_bwd(blk.conv2.name, t[p + ".y2"], t[p + ".out"], dout, t[p + ".dy2"], dout, True,
                             keep_g=not (masked_acc or gn_ds_mask), from_sums=self._dout_sums.pop(p, None))
            # data gradient first: the weight gradient (a leaf) then runs beside the BatchNorm chain that follows
            fused_sums = self._dgrad_bnsums_slots(blk.conv2.name) if (self.dgrad_bnsums and self.norm == "batch") else 0
            if self.wgrad_first:   # weight gradient, then data gradient, so that the BatchNorm backward pass which
                # follows finds the data gradient it reads still in the Infinity Cache
                self._wgrad(blk.conv2.name, t[p + ".a1"], t[p + ".dy2"])
            if fused_sums:
                c2, b1 = self.convs[blk.conv2.name], bn_name(blk.conv1.name)
                sm1, si1 = self.save[b1]
                if self.wgrad_overlap == 1:
                    self._join_wgrad_stream()   # (as _dgrad does: two MFMA-bound kernels never run side by side)
                sums = self._bwd_sums(blk.conv2.name, fused_sums, blk.conv1.cout)
                self._timed("dgrad", c2, lambda: call(
                    "primia_conv2d_dgrad_bnsums", c2.desc, t[p + ".dy2"], c2.w_dgrad, t[p + ".da1"], t[p + ".y1"], sm1, si1,
                    self.views[b1 + ".weight"], self.views[b1 + ".bias"], sums, self.dt))
            else:
                self._dgrad(blk.conv2.name, t[p + ".dy2"], t[p + ".da1"], False, blk.conv1.name, t[p + ".y1"])
            if not self.wgrad_first:
                self._wgrad(blk.conv2.name, t[p + ".a1"], t[p + ".dy2"])
            if fused_sums:
                y1 = t[p + ".y1"]
                call("primia_bn_relu_bwd_from_sums", y1, t[p + ".da1"], t[p + ".dy1"], self.views[b1 + ".weight"],
                     self.views[b1 + ".bias"], sm1, si1, self._gviews[b1 + ".weight"], self._gviews[b1 + ".bias"], sums,
                     fused_sums, y1.shape[0], y1.shape[1], self.dt)
            else:
                self._bn_bwd(blk.conv1.name, t[p + ".y1"], t[p + ".a1"], t[p + ".da1"], t[p + ".dy1"], None, True)
            if blk.down is not None and self.pair_dgrad:
                # both BatchNorm backward passes first, then ONE data-gradient pass for conv1 + downsample
                if not bn_pair and gn_ds_mask:
                    bd = bn_name(blk.down.name)
                    (smd, sid), (psg, psb) = self.save[bd], self.ps_affine[bd]
                    yd = t[p + ".yd"]
                    call("primia_gn_bwd_mask", yd, self.relu_masks[b2], dout, t[p + ".dyd"], None, self.views[bd + ".weight"],
                         smd, sid, psg, psb, self.N, yd.shape[0] // self.N, yd.shape[1], self.groups, self.bn_ws,
                         self.bn_ws_bytes, self.dt)
                    if self.dp is None:
                        call("primia_weighted_colsum", psg, self.ones_n, self._gviews[bd + ".weight"], self.N, yd.shape[1])
                        call("primia_weighted_colsum", psb, self.ones_n, self._gviews[bd + ".bias"], self.N, yd.shape[1])
                elif not bn_pair:
                    self._bn_bwd(blk.down.name, t[p + ".yd"], None, dout, t[p + ".dyd"], None, False)
                c1, cd = self.convs[blk.conv1.name], self.convs[blk.down.name]
                if self.wgrad_overlap == 1:
                    self._join_wgrad_stream()
                if self.wgrad_first:
                    self._wgrad_transition(blk, x_in, t[p + ".dy1"], t[p + ".dyd"])
                prev = blocks[i - 1] if i > 0 else None
                pb2 = bn_name(prev.conv2.name) if prev is not None else None
                pslots = 0
                if (self.pair_bnsums and self.norm == "batch" and self.dp is None and prev is not None and prev.down is None
                        and pb2 in self.relu_masks and self.dtype == torch.bfloat16):
                    if getattr(c1, "pair_bnsums_slots", None) is None:
                        c1.pair_bnsums_slots = query("primia_conv_dgrad_pair_bnsums_slots", c1.desc, self.dt)
                    pslots = c1.pair_bnsums_slots
                if pslots > 0:
                    # ... and the backward sums of the previous block's bn2 out of the same write-back
                    sums = self._bwd_sums(blk.conv1.name + ".pair", pslots, prev.conv2.cout)
                    smp, sip = self.save[pb2]
                    self._timed("dgrad", c1, lambda: call(
                        "primia_conv2d_dgrad_pair_bnsums", c1.desc, t[p + ".dy1"], c1.w_dgrad, cd.desc, t[p + ".dyd"], cd.w_dgrad,
                        dx_in, t[prev.prefix + ".y2"], self.relu_masks[pb2], smp, sip, sums, self.dt), extra_macs=self._macs(cd))
                    self._dout_sums[prev.prefix] = (sums, pslots)
                else:
                    self._timed("dgrad", c1, lambda: call("primia_conv2d_dgrad_pair", c1.desc, t[p + ".dy1"], c1.w_dgrad,
                                                          cd.desc, t[p + ".dyd"], cd.w_dgrad, dx_in, self.dt),
                                extra_macs=self._macs(cd))
                if not self.wgrad_first:
                    self._wgrad_transition(blk, x_in, t[p + ".dy1"], t[p + ".dyd"])
            elif blk.down is not None:
                self._dgrad(blk.conv1.name, t[p + ".dy1"], dx_in, False)
                self._wgrad(blk.conv1.name, x_in, t[p + ".dy1"])
                self._bn_bwd(blk.down.name, t[p + ".yd"], None, dout, t[p + ".dyd"], None, False)
                self._dgrad(blk.down.name, t[p + ".dyd"], dx_in, True)
                self._wgrad(blk.down.name, x_in, t[p + ".dyd"])
            else:
                # identity skip: dx_in aliases dout, which now holds the masked gradient g
                if self.wgrad_first:
                    self._wgrad(blk.conv1.name, x_in, t[p + ".dy1"])
                if masked_acc:
                    c1 = self.convs[blk.conv1.name]
                    if self.wgrad_overlap == 1:
                        self._join_wgrad_stream()
                    # ... whose write-back also forms the backward sums of the BatchNorm whose output gradient it completes:
                    # the previous block's bn2 (mode 2), or the stem's bn1 through the max-pool (mode 3)
                    aslots, amode = 0, 0
                    if self.acc_bnsums and self.norm == "batch" and self.dp is None and self.dtype == torch.bfloat16:
                        prev = blocks[i - 1] if i > 0 else None
                        pb2 = bn_name(prev.conv2.name) if prev is not None else None
                        if prev is not None and prev.down is None and pb2 in self.relu_masks:
                            amode = 2
                        elif prev is None and self._stem_bwd_fused_wanted():
                            amode = 3
                        if amode:
                            if getattr(c1, "acc_bnsums_slots", None) is None:
                                c1.acc_bnsums_slots = query("primia_conv_dgrad_masked_acc_bnsums_slots", c1.desc, self.dt)
                            aslots = c1.acc_bnsums_slots
                    if aslots > 0 and amode == 2:
                        sums = self._bwd_sums(blk.conv1.name + ".acc", aslots, prev.conv2.cout)
                        smp, sip = self.save[pb2]
                        self._timed("dgrad", c1, lambda: call(
                            "primia_conv2d_dgrad_masked_acc_bnsums", c1.desc, t[p + ".dy1"], c1.w_dgrad, dx_in,
                            self.relu_masks[b2], 2, t[prev.prefix + ".y2"], self.relu_masks[pb2], smp, sip, sums, self.dt))
                        self._dout_sums[prev.prefix] = (sums, aslots)
                    elif aslots > 0 and amode == 3:
                        sums = self._bwd_sums(blk.conv1.name + ".acc", aslots, 64)
                        self._timed("dgrad", c1, lambda: call(
                            "primia_conv2d_dgrad_masked_acc_bnsums", c1.desc, t[p + ".dy1"], c1.w_dgrad, dx_in,
                            self.relu_masks[b2], 3, t["pool.out"], None, self.views["bn1.bias"], self.views["bn1.weight"],
                            sums, self.dt))
                        self._pool_sums = (sums, aslots)
                    else:
                        self._timed("dgrad", c1, lambda: call("primia_conv2d_dgrad_masked_acc", c1.desc, t[p + ".dy1"],
                                                              c1.w_dgrad, dx_in, self.relu_masks[b2], self.dt))
                elif i > 0:
                    self._dgrad(blk.conv1.name, t[p + ".dy1"], dx_in, True, blocks[i - 1].conv2.name,
                                t[blocks[i - 1].prefix + ".y2"])
                else:
                    self._dgrad(blk.conv1.name, t[p + ".dy1"], dx_in, True)
                if not self.wgrad_first:
                    self._wgrad(blk.conv1.name, x_in, t[p + ".dy1"])
        hw = self.stem_hw
        # the stem's tail in two launches less and without the dy tensor (411 MB at batch 256): bn1's backward sums at
        # pooled resolution, then conv1's weight gradient forming its dy tiles on the fly (primia_stem_bwd_fused)
        stem_bwd_fused = self._stem_bwd_fused_wanted()
        self.stem_bwd_fused_active = stem_bwd_fused
        pool_sums, self._pool_sums = getattr(self, "_pool_sums", None), None
        if stem_bwd_fused:
            sm, si = self.save["bn1"]
            c = self.convs["conv1"]
            S = self.spec.input_size
            if pool_sums is not None:     # the reduction over (pooled, dpooled) happened in layer1.0.conv1's data gradient
                call("primia_bn_relu_maxpool_bwd_from_sums", t["stem.y"], t["pool.out"], t["pool.dout"], self.pool_argmax,
                     self.views["bn1.weight"], self.views["bn1.bias"], sm, si, self._gviews["bn1.weight"],
                     self._gviews["bn1.bias"], pool_sums[0], pool_sums[1], N, hw, hw, 64, self.dt)
            else:
                call("primia_bn_relu_maxpool_bwd", t["stem.y"], t["pool.out"], t["pool.dout"], self.pool_argmax, None,
                     self.views["bn1.weight"], self.views["bn1.bias"], sm, si, self._gviews["bn1.weight"],
                     self._gviews["bn1.bias"], N, hw, hw, 64, self.bn_ws, self.bn_ws_bytes, self.dt)
            self._on_wgrad_stream(lambda: self._timed("wgrad", c, lambda: call(
                "primia_stem_bwd_fused", self.x0p, t["stem.y"], t["pool.dout"], self.pool_argmax,
                self.views["bn1.weight"], self.views["bn1.bias"], sm, si, self._gviews["bn1.weight"],
                self._gviews["bn1.bias"], c.acc, self.wgrad_ws, self.wgrad_ws_bytes, N, S, S, self.dt)))
            if self.dp is None:
                self._finalize_wgrads()
            return
        if self._stem_fused:
            sm, si = self.save["bn1"]
            call("primia_bn_relu_maxpool_bwd", t["stem.y"], t["pool.out"], t["pool.dout"], self.pool_argmax, t["stem.dy"],
                 self.views["bn1.weight"], self.views["bn1.bias"], sm, si, self._gviews["bn1.weight"],
                 self._gviews["bn1.bias"], N, hw, hw, 64, self.bn_ws, self.bn_ws_bytes, self.dt)
        elif getattr(self, "_stem_fused_gn", False):
            sm, si = self.save["bn1"]
            psg, psb = self.ps_affine["bn1"]
            call("primia_gn_relu_maxpool_bwd", t["stem.y"], t["pool.out"], t["pool.dout"], self.pool_argmax, t["stem.dy"],
                 self.views["bn1.weight"], self.views["bn1.bias"], sm, si, psg, psb, N, hw, hw, 64, self.groups,
                 self.bn_ws, self.bn_ws_bytes, self.dt)
            if self.dp is None:
                call("primia_weighted_colsum", psg, self.ones_n, self._gviews["bn1.weight"], N, 64)
                call("primia_weighted_colsum", psb, self.ones_n, self._gviews["bn1.bias"], N, 64)
        else:
            if self.spec.pooling == "max":
                call("primia_maxpool3x3s2_bwd", t["pool.dout"], self.pool_argmax, t["stem.dz"], N, hw, hw, 64, self.dt)
            else:
                call("primia_avgpool3x3s2_bwd", t["pool.dout"], t["stem.dz"], N, hw, hw, 64, self.dt)
            self._bn_bwd("conv1", t["stem.y"], t["stem.z"], t["stem.dz"], t["stem.dy"], None, True)
        if self._stem_padded and self.dp is None:
            c = self.convs["conv1"]
            S = self.spec.input_size
            if self.wgrad_ws is not None:
                self._on_wgrad_stream(lambda: self._timed("wgrad", c, lambda: call(
                    "primia_stem_conv_wgrad_ws", self.x0p, t["stem.dy"], c.acc, self.wgrad_ws, self.wgrad_ws_bytes, N, S,
                    S, self.dt)))
            else:
                self._on_wgrad_stream(lambda: self._timed("wgrad", c, lambda: call("primia_stem_conv_wgrad", self.x0p,
                                                                                 t["stem.dy"], c.acc, N, S, S, self.dt)))
        else:
            self._wgrad("conv1", self.x0, t["stem.dy"])
        if self.dp is None:
            self._finalize_wgrads()

    # measurement hook (tools/power_idle_ab.py): called after every grouped 3x3 weight-gradient launch; None in every product run
    after_wgrad_hook = None

    def _flush_wgrad_group(self, key):
        held = self._wg_held.pop(key, [])
        if self.after_wgrad_hook is not None and held:
            try:
                return self._flush_wgrad_group_inner(held)
            finally:
                self.after_wgrad_hook()
        return self._flush_wgrad_group_inner(held)

    def _flush_wgrad_group_inner(self, held):
        if len(held) >= 2:
            c = self.convs[held[0][0]]
            args = []
            for i in range(4):
                if i < len(held):
                    nm, x, dy = held[i]
                    args += [x, dy, self.convs[nm].acc]
                else:
                    args += [None, None, None]
            self._on_wgrad_stream(lambda: self._timed(
                "wgrad", c, lambda: call("primia_conv2d_wgrad_group_ws", c.desc, len(held), *args, self.wgrad_ws,
                                         self.wgrad_ws_bytes, self.dt), extra_macs=(len(held) - 1) * self._macs(c)))
        else:
            for nm, x, dy in held:
                c = self.convs[nm]
                self._on_wgrad_stream(lambda c=c, x=x, dy=dy: self._timed("wgrad", c, lambda: call(
                    "primia_conv2d_wgrad_ws", c.desc, x, dy, c.acc, self.wgrad_ws, self.wgrad_ws_bytes, self.dt)))

    def _finalize_wgrads(self):
        for key in list(self._wg_held):      # (groups that never filled: odd layer counts)
            self._flush_wgrad_group(key)
        self._join_wgrad_stream()
        if self.fuse_sgd_tail and self.dp is None and self._sgd_tail_plan() is not None:
            # the accumulators stay as they are: sgd_step() turns them into gradients, new weights and new kernel-layout
            # copies in one pass per tile; anything else that wants the gradients calls materialize_grads()
            self._grads_pending = True
            return
        m = self._many_args()
        call("primia_conv_wgrad_finalize_many", m["descs"], m["creal"], m["acc"], m["gw"], m["n"])

    # The step's tail — accumulator -> OIHW gradient, SGD on the fp32 master, master -> compute-dtype copies — as ONE
    # pass per weight tile (primia_conv_sgd_step_many: 95 -> ~55 us at batch 256, bit-identical weights and gradients).
    # Opt-in: between loss_backward() and sgd_step() the conv gradients then live in the accumulators only;
    # materialize_grads() (called by every other reader in this class) writes them out the unfused way.
    fuse_sgd_tail = False
    _grads_pending = False

    def _sgd_tail_plan(self):
        """(fused conv args, remaining conv args or None, SGD ranges outside the fused weights) — built once."""
        if getattr(self, "_sgd_plan", None) is None:
            import ctypes

            cs = list(self.convs.values())
            fus = [c for c in cs if query("primia_conv_sgd_fusable", c.desc, c.c_real) == 1 and c.w_dgrad is not None
                   and all(tt.data_ptr() % 16 == 0 for tt in (c.acc, c.w_fwd, c.w_dgrad, self.views[c.spec.name + ".weight"],
                                                              self._gviews[c.spec.name + ".weight"]))]
            rest = [c for c in cs if c not in fus]
            vp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)

            def pack(lst):
                n = len(lst)
                if n == 0:
                    return None
                return dict(
                    n=n, descs=(ConvDesc * n)(*[c.desc for c in lst]), creal=(ctypes.c_int * n)(*[c.c_real for c in lst]),
                    w=(ctypes.c_void_p * n)(*[vp(self.views[c.spec.name + ".weight"]) for c in lst]),
                    wf=(ctypes.c_void_p * n)(*[vp(c.w_fwd) for c in lst]),
                    wd=(ctypes.c_void_p * n)(*[vp(c.w_dgrad) for c in lst]),
                    acc=(ctypes.c_void_p * n)(*[vp(c.acc) for c in lst]),
                    gw=(ctypes.c_void_p * n)(*[vp(self._gviews[c.spec.name + ".weight"]) for c in lst]))

            fused_keys = {c.spec.name + ".weight" for c in fus}
            ranges, off = [], 0
            for k, shp in self.p_entries:
                n = int(torch.Size(shp).numel())
                if k not in fused_keys:
                    if ranges and ranges[-1][0] + ranges[-1][1] == off:
                        ranges[-1][1] += n
                    else:
                        ranges.append([off, n])
                off += n
            if not fus or len(ranges) > 32:
                self._sgd_plan = False
            else:
                nr = len(ranges)
                self._sgd_plan = (pack(fus), pack(rest), (ctypes.c_int64 * nr)(*[r[0] for r in ranges]),
                                  (ctypes.c_int64 * nr)(*[r[1] for r in ranges]), nr)
        return self._sgd_plan or None

    @property
    def grads(self):
        """The flat gradient arena.  Reading it finishes a deferred conv-gradient pass first (fuse_sgd_tail), so a
        caller never sees gradients that still live in the accumulators; `gviews[...]` are views of the same storage —
        call materialize_grads() before reading THOSE between loss_backward() and sgd_step() under fuse_sgd_tail."""
        self.materialize_grads()
        return self._grads

    def materialize_grads(self):
        """Write the conv gradients out of the accumulators if a fused tail has left them there."""
        if self._grads_pending:
            m = self._many_args()
            call("primia_conv_wgrad_finalize_many", m["descs"], m["creal"], m["acc"], m["gw"], m["n"])
            self._grads_pending = False

    # ------------------------------------------------------------------------------------------
    # DP-SGD (BASELINE.json configs[3]; parameter values of train.py:325-334)
    # ------------------------------------------------------------------------------------------
    def dp_loss_backward(self, target, max_grad_norm=1.0, noise_multiplier=1.3, noise=None, generator=None):
        """Per-sample gradient clipping + Gaussian noise, pytorch-dp semantics:
            g = (1/B) * ( sum_n min(1, C / (||g_n|| + 1e-6)) * g_n  +  N(0, (noise_multiplier*C)^2 I) )
        with g_n the gradient of sample n's OWN loss over all 62 parameter tensors (flat L2 norm).
        Needs norm="group".  `noise` (fp32 [P], standard normal) may be given for reproducibility.

        How: one ordinary backward pass yields every layer's activation gradient dy (samples are
        independent under GroupNorm).  Pass 1 runs the weight-gradient kernels with one pixel split per image
        and has every block add the squares of ITS (complete) per-sample tile to ||g_n||^2 — the per-sample
        gradients are never written; then each sample's rows of
        dy are scaled by its clip factor and the ordinary batched wgrad — linear in dy — produces
        sum_n clip_n * g_n directly."""
        if self.norm != "group":
            raise _lib.PrimiaError("DP-SGD needs the BatchNorm-free network: ResNet18Engine(norm='group')")
        N, nc, dev = self.N, self.spec.num_classes, self.device
        # per-sample loss gradients: softmax - onehot (xent gives them divided by the batch size)
        call("primia_xent_hard", self.logits, target, None, self.loss, self.dlogits, N, nc)
        call("primia_scale", self.dlogits, self.dlogits.numel(), float(N))
        self.dp = {"wgrads": []}
        try:
            self.backward()
            sq = torch.zeros(N, dtype=torch.float64, device=dev)
            # fc: g_W[n] = dlogits[n]^T feat[n], g_b[n] = dlogits[n]
            ps_fc = torch.empty(N, nc * 512 + nc, dtype=torch.float32, device=dev)
            call("primia_fc_persample_grads", self.feat, self.dlogits, ps_fc, N, 512, nc)
            trace = getattr(self, "dp_trace", None)  # optional {name: cumulative sq norms} for debugging

            def mark(name):
                if trace is not None:
                    trace[name] = sq.clone()

            call("primia_persample_sqnorm", ps_fc, N, nc * 512 + nc, sq)
            mark("fc")
            if trace is None:   # all 40 per-sample affine gradients in one launch
                if getattr(self, "_sqnorm_many", None) is None:
                    xs = [tt.data_ptr() for pair in self.ps_affine.values() for tt in pair]
                    ws_ = [tt.shape[1] for pair in self.ps_affine.values() for tt in pair]
                    self._sqnorm_many = (torch.tensor(xs, dtype=torch.int64, device=dev),
                                         torch.tensor(ws_, dtype=torch.int32, device=dev), len(xs))
                call("primia_persample_sqnorm_many", self._sqnorm_many[0], self._sqnorm_many[1], self._sqnorm_many[2], N, sq)
            else:
                for b, (psg, psb) in self.ps_affine.items():
                    call("primia_persample_sqnorm", psg, N, psg.shape[1], sq)
                    call("primia_persample_sqnorm", psb, N, psb.shape[1], sq)
                    mark(b)
            # layers whose per-sample gradient tiles are small enough to KEEP (the stem, layer1's 64 -> 64 convs): the
            # norm pass stores them and the clipped sum is a weighted reduce — no row scaling of dy, no second pass
            kept = self._dp_keep_buffers()
            for name, x, dy in self.dp["wgrads"]:
                c = self.convs[name]
                done = False
                if name in kept:
                    if name == "conv1":
                        S = self.spec.input_size
                        call("primia_stem_conv_wgrad_persample_sqnorm_keep", self.x0p, dy, sq, kept[name],
                             kept[name].numel() * 4, N, S, S, self.dt)
                    else:
                        call("primia_conv2d_wgrad_persample_sqnorm_keep", c.desc, x, dy, sq, kept[name],
                             kept[name].numel() * 4, self.dt)
                    mark(name)
                    continue
                if name == "conv1" and self._stem_padded:   # halo kernel on the padded input, one block per image
                    S = self.spec.input_size
                    try:
                        call("primia_stem_conv_wgrad_persample_sqnorm", self.x0p, dy, sq, N, S, S, self.dt)
                        done = True
                    except _lib.PrimiaError:
                        done = False
                if not done:
                    if name == "conv1" and not self._x0_valid:
                        raise _lib.PrimiaError("per-sample stem gradient: the padded-input kernel refused a shape it "
                                               "advertised (primia_stem_conv_wgrad_ws_bytes > 0)")
                    call("primia_conv2d_wgrad_persample_sqnorm", c.desc, x, dy, sq, self.dt)
                mark(name)
            clip = torch.empty(N, dtype=torch.float32, device=dev)
            call("primia_dp_clip_factors", sq, clip, N, float(max_grad_norm))
            # clipped sums
            # (a transition block's conv1 + downsample: ONE launch as in plain training, once both dy are scaled)
            pair_of, pair_wait = {}, {}
            if self.wgrad_pair and self.wgrad_ws is not None:
                for blk in self.spec.blocks:
                    if (blk.down is not None and self._pair_ws.get(blk.conv1.name, 0) > 0
                            and blk.conv1.name not in kept and blk.down.name not in kept):
                        pair_of[blk.conv1.name] = pair_of[blk.down.name] = blk
            # every dy of a layer whose tiles are not kept, scaled by the clip factors in ONE launch (a dozen tensors)
            to_scale = [dy for name, _, dy in self.dp["wgrads"] if name not in kept]
            many = self.dp_scale_many and 0 < len(to_scale) <= 16
            if many:
                key = tuple(tt.data_ptr() for tt in to_scale)
                if getattr(self, "_scale_many", (None,))[0] != key:
                    import ctypes

                    self._scale_many = (key, (ctypes.c_void_p * len(key))(*key),
                                        (ctypes.c_int64 * len(key))(*[tt.numel() // N for tt in to_scale]))
                call("primia_scale_rows_many", self._scale_many[1], self._scale_many[2], len(key), clip, N, self.dt)
            for name, x, dy in self.dp["wgrads"]:
                c = self.convs[name]
                if name in pair_of:
                    blk = pair_of[name]
                    if not many:
                        call("primia_scale_rows", dy, clip, N, dy.numel() // N, self.dt)
                    got = pair_wait.setdefault(blk.prefix, {})
                    got[name] = (x, dy)
                    if len(got) == 2:
                        c1, cd = self.convs[blk.conv1.name], self.convs[blk.down.name]
                        call("primia_conv2d_wgrad_pair_ws", c1.desc, got[blk.conv1.name][0], got[blk.conv1.name][1], c1.acc,
                             cd.desc, got[blk.down.name][1], cd.acc, self.wgrad_ws, self.wgrad_ws_bytes, self.dt)
                    continue
                if name in kept:
                    if name == "conv1":
                        call("primia_stem_conv_wgrad_clipped_sum", kept[name], clip, c.acc, N)
                    else:
                        call("primia_conv_wgrad_clipped_sum", c.desc, kept[name], clip, c.acc, self.dt)
                    continue
                if not many:
                    call("primia_scale_rows", dy, clip, N, dy.numel() // N, self.dt)
                # the clipped SUM is an ordinary batched weight gradient: atomic-free kernels, and for the stem the
                # halo kernel on the padded input (118 us instead of 444 us for the per-tap one)
                if self.wgrad_ws is not None and name in self._wg_group:     # same-shape layers: one launch per stage
                    key, n = self._wg_group[name]
                    held = self._wg_held.setdefault(key, [])
                    held.append((name, x, dy))
                    if len(held) == n:
                        self._flush_wgrad_group(key)
                    continue
                if self.wgrad_ws is None:
                    call("primia_conv2d_wgrad", c.desc, x, dy, c.acc, self.dt)
                elif name == "conv1" and self._stem_padded:
                    S = self.spec.input_size
                    call("primia_stem_conv_wgrad_ws", self.x0p, dy, c.acc, self.wgrad_ws, self.wgrad_ws_bytes, N, S, S,
                         self.dt)
                else:
                    call("primia_conv2d_wgrad_ws", c.desc, x, dy, c.acc, self.wgrad_ws, self.wgrad_ws_bytes, self.dt)
            self._finalize_wgrads()
            # clipped sums of the per-sample affine gradients: all 40 (dgamma, dbeta) pairs in ONE launch
            if getattr(self, "_colsum_many", None) is None:
                xs, outs, ws_ = [], [], []
                for b, (psg, psb) in self.ps_affine.items():
                    xs += [psg.data_ptr(), psb.data_ptr()]
                    outs += [self._gviews[b + ".weight"].data_ptr(), self._gviews[b + ".bias"].data_ptr()]
                    ws_ += [psg.shape[1], psb.shape[1]]
                self._colsum_many = (torch.tensor(xs, dtype=torch.int64, device=dev),
                                     torch.tensor(outs, dtype=torch.int64, device=dev),
                                     torch.tensor(ws_, dtype=torch.int32, device=dev), len(xs), max(ws_))
            xs_d, outs_d, ws_d, cnt, mw = self._colsum_many
            call("primia_weighted_colsum_many", xs_d, clip, outs_d, ws_d, cnt, mw, N)
            call("primia_weighted_colsum", ps_fc[:, :nc * 512].contiguous(), clip, self._gviews["fc.weight"].view(-1), N,
                 nc * 512)
            call("primia_weighted_colsum", ps_fc[:, nc * 512:].contiguous(), clip, self._gviews["fc.bias"], N, nc)
        finally:
            if getattr(self, "dp_keep_operands", False):      # tests: the (layer, x, dy) triples the norm pass walked
                self.dp_operands = list(self.dp["wgrads"])
            self.dp = None
        if noise is None:
            noise = torch.randn(self.P, dtype=torch.float32, device=dev, generator=generator)
        call("primia_dp_add_noise", self.grads, noise, self.P, float(noise_multiplier * max_grad_norm), 1.0 / N)
        self.dp_stats = {"sq_norms": sq, "clip": clip}
        return self.loss

    dp_keep = True
    dp_scale_many = True

    def _dp_keep_buffers(self):
        """{conv name: fp32 buffer} for the layers whose per-sample tiles the DP-SGD norm pass keeps (built once)."""
        if getattr(self, "_dp_keep", None) is None:
            self._dp_keep = {}
            if self.dp_keep and self.dtype == torch.bfloat16 and self.wgrad_ws is not None:
                for c in self.spec.convs:
                    if c.name == "conv1":
                        S = self.spec.input_size
                        n = query("primia_stem_conv_wgrad_persample_slab_bytes", self.N, S, S) if self.x0p is not None else 0
                    else:
                        n = query("primia_conv_wgrad_persample_slab_bytes", self.convs[c.name].desc, self.dt)
                    if n > 0:
                        self._dp_keep[c.name] = torch.empty(n // 4, dtype=torch.float32, device=self.device)
        if "conv1" in self._dp_keep and not self._stem_padded:
            return {k: v for k, v in self._dp_keep.items() if k != "conv1"}
        return self._dp_keep

    # ------------------------------------------------------------------------------------------
    # optimizer
    # ------------------------------------------------------------------------------------------
    def zero_grad(self):
        self._grads_pending = False
        self.grads.zero_()

    def note_replayed_steps(self, n):
        """Training steps that ran as hipGraph replays: advance the host-side counters forward() would have advanced."""
        for b in self.num_batches_tracked:
            self.num_batches_tracked[b] += int(n)
        if self.opt_state is not None:
            self.opt_steps += int(n)

    def reset_optimizer(self):
        """The reference re-creates its optimizers at every FedAvg sync (utils.py:1131-1145,1208-1218)."""
        self.opt_state = None
        self.opt_steps = 0

    def sgd_step(self, lr, weight_decay=0.0):
        if self._grads_pending:
            fus, rest, rb, rl, nr = self._sgd_tail_plan()
            self._grads_pending = False
            call("primia_conv_sgd_step_many", fus["descs"], fus["creal"], fus["acc"], fus["gw"], fus["w"], fus["wf"],
                 fus["wd"], fus["n"], float(lr), float(weight_decay), self.dt)
            if rest is not None:
                call("primia_conv_wgrad_finalize_many", rest["descs"], rest["creal"], rest["acc"], rest["gw"], rest["n"])
            call("primia_sgd_step_ranges", self.flat, self._grads, rb, rl, nr, float(lr), float(weight_decay))
            if rest is not None:
                call("primia_conv_weight_prepare_many", rest["descs"], rest["creal"], rest["w"], rest["wf"], rest["wd"],
                     rest["n"], self.dt)
            return
        call("primia_sgd_step", self.flat, self.grads, self.P, float(lr), float(weight_decay))
        self.refresh_weights()

    def adam_step(self, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.materialize_grads()
        if self.opt_state is None:
            self.opt_state = (torch.zeros_like(self.grads), torch.zeros_like(self.grads))
            self.opt_steps = 0
        self.opt_steps += 1
        m, v = self.opt_state
        call("primia_adam_step", self.flat, self.grads, m, v, self.P, float(lr), float(betas[0]), float(betas[1]),
             float(eps), float(weight_decay), self.opt_steps)
        self.refresh_weights()

    def train_step(self, x_nchw, target, lr, weight_decay=0.0, soft=False):
        """forward + loss + backward + SGD: the body of the reference's per-batch loop."""
        self.forward(x_nchw)
        self.loss_backward(target, soft=soft)
        self.sgd_step(lr, weight_decay)
        return self.loss
