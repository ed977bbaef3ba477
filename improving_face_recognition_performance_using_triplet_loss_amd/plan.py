"""Static execution plan: lowers a `graph.Sym` DAG onto the HIP kernels and runs it forward and backward.

This plays the role MXNet's executor (`Module.bind` / Gluon `hybridize`, ref: mutli_gpu_v3.py:153-154,
train_efm.py:209) plays for the reference, designed for one MI355X per process:

  * all parameters live in ONE flat fp32 buffer in the kernels' packed layout (OHWI, padded; see
    include/efm_hip.h) — the optimiser is a single fused launch over it and the data-parallel gradient
    exchange is a handful of large RCCL all-reduces over slices of the matching flat gradient buffer;
  * activations are NHWC with channels padded to 4; a conv's bias and a residual add ride in its epilogue;
  * backward is a hand-ordered reverse sweep that calls `ready_cb(lo, hi)` as soon as a slice of the flat
    gradient is final, so the all-reduce of late layers overlaps the backward of early ones.
"""
import collections
import os

import numpy as np
import torch

from . import graph, ops
from ._lib import pad4


_STREAMS = {}


def _shared_stream(device, kind):
    """ONE weight-gradient stream and ONE reduction stream per device and process, shared by every plan.  High priority: such a
    stream keeps a hardware queue of its own even after RCCL has created its streams (with a default-priority stream the
    wgrad / dgrad overlap disappeared once a process group existed: measured).  Shared: every new torch stream comes from a pool
    that maps onto a few hardware queues — a third plan's private stream landed on the main stream's queue and its backward ran
    45 % slower (the deeper-CNN leg of bench.py's `secondary`, 6.9 -> 10.0 ms, after the headline network had taken two streams)."""
    key = (device.type, device.index, kind)
    st = _STREAMS.get(key)
    if st is None:
        st = _STREAMS[key] = torch.cuda.Stream(device=device, priority=int(os.environ.get("EFM_SIDE_PRIO", "-1")))
    return st


class ParamSpec:
    __slots__ = ("name", "kind", "offset", "numel", "desc", "mx_shape", "step", "rowpack")

    def __init__(self, name, kind, offset, numel, desc, mx_shape, step, rowpack=False):
        self.name, self.kind, self.offset, self.numel = name, kind, offset, numel
        self.desc, self.mx_shape, self.step = desc, mx_shape, step
        # the first convolution on a row-packed input: `desc` is the kh x 1 convolution on kw*cin channels, the packed weight is
        # w'[n][(j, ci)][kh] = w[n][ci][kh][j]; mx_shape stays MXNet's (cout, cin, kh, kw) and load / export / init re-index
        self.rowpack = rowpack

    def to_packed_order(self, a):
        """MXNet (cout, cin, kh, kw) -> the (cout, cin', kh', kw') array `desc` packs."""
        if not self.rowpack:
            return a
        cout, cin, kh, kw = self.mx_shape
        return a.permute(0, 3, 1, 2).reshape(cout, kw * cin, kh, 1).contiguous()

    def from_packed_order(self, w):
        if not self.rowpack:
            return w
        cout, cin, kh, kw = self.mx_shape
        return w.reshape(cout, kw, cin, kh).permute(0, 2, 3, 1).contiguous()


class Step:
    """One lowered operation."""

    def __init__(self, op, node, inputs, shape):
        self.op = op
        self.node = node
        self.inputs = inputs        # producer steps
        self.shape = shape          # (C, H, W) of the output
        self.residual = None        # conv only: step whose output is added in the epilogue
        self.epi = None             # conv only: fused MFM (+ pool) epilogue {ways, order, pool, conv_shape}
        self.desc = None
        self.pname = None
        self.no_bias = False
        self.f32 = False            # bf16 plans: this convolution (the embedding head) runs on the fp32 kernels
        self.needs_grad = False     # does any gradient have to flow into this step's output?
        self.index = -1


class Plan:
    def __init__(self, outputs, input_shape, device="cuda", fuse=None, dtype="f32"):
        """fuse: fold `conv -> MFM [-> pool]` chains into the convolution's epilogue (default on; EFM_FUSE=0 or
        fuse=False keeps one kernel per graph node — the form the oracle-routing parity test uses).
        dtype: "f32" (exact fp32 MFMA) or "bf16" (BASELINE configs[2]: bf16 activations / operands, fp32 accumulate, fp32
        master weights and gradients; supported for networks whose every MFM / pooling is fused into a convolution)."""
        if dtype not in ("f32", "bf16"):
            raise ValueError("dtype must be 'f32' or 'bf16'")
        self.dtype = dtype
        self.device = torch.device(device)
        self.batch = int(input_shape[0])
        self.input_shape = tuple(int(v) for v in input_shape)
        self.steps = []
        self.params = collections.OrderedDict()
        self.rowpack = os.environ.get("EFM_ROWPACK", "1") != "0"
        self.fused_wgrad = os.environ.get("EFM_FUSED_WGRAD", "1") != "0"   # bf16: weight gradient straight from dz + route bytes where it applies
        self._lower(outputs)
        self.fuse = (os.environ.get("EFM_FUSE", "1") != "0") if fuse is None else bool(fuse)
        self.fused = 0
        if self.fuse:
            self._fuse()
        if self.dtype == "bf16":
            self._check_bf16()
        self._acts = None
        self._views = {}
        self._wd_scratch = None
        self._side = None
        self._red = None
        self._ws2 = None
        self._ev_pool = []
        self._u_dgrad_ready = False
        self.two_streams = os.environ.get("EFM_TWO_STREAMS", "1") != "0"
        # the slab reduction of layer L (pure streaming work) on a third stream, under the matrix-core kernel of layer L-1
        self.reduce_stream = os.environ.get("EFM_REDUCE_STREAM", "1") != "0"

    # ------------------------------------------------------------------------------ lowering
    def _lower(self, outputs):
        nodes = graph.topo_sort(outputs)
        consumers = collections.Counter()
        for n in nodes:
            for i in n.inputs:
                consumers[i.id] += 1
        for o in outputs:
            consumers[o.id] += 1
        step_of = {}
        b = self.batch
        offset = 0
        use_count = collections.Counter()
        for n in nodes:
            if n.op == "var":
                c, h, w = self.input_shape[1:]
                st = Step("input", n, [], (c, h, w))
            elif n.op in ("conv", "fc"):
                src = step_of[n.inputs[0].id]
                c, h, w = src.shape
                if n.op == "conv":
                    (kh, kw), (ph, pw), cout = n.attrs["kernel"], n.attrs["pad"], n.attrs["num_filter"]
                else:
                    (kh, kw), (ph, pw), cout = (h, w), (0, 0), n.attrs["num_hidden"]
                rowpack = False
                if (self.rowpack and n.op == "conv" and src.op == "input" and kw > 1 and 2 * pw == kw - 1 and kw * c <= 16
                        and n.name + "_weight" not in self.params):
                    # FIRST convolution (5x5 on 3 / 1 channels, ref: efm_symbol.py:84, lightcnn.py:82): the kw taps of a kernel row become
                    # channels of a row-packed input (efm_rowpack_nchw), the layer runs as a kh x 1 convolution on kw*c channels and
                    # K = kh*kw*c packs densely (80 instead of 112 in fp32, 96 instead of 224 in bf16)
                    rp = Step("rowpack", n, [src], (kw * c, h, w))
                    rp.kw, rp.pad_w = kw, pw
                    rp.index = len(self.steps)
                    self.steps.append(rp)
                    src, rowpack = rp, True
                    d = ops.conv_desc(b, h, w, kw * c, cout, kh, 1, ph, 0)
                else:
                    d = ops.conv_desc(b, h, w, c, cout, kh, kw, ph, pw)
                st = Step("conv", n, [src], (cout, d.hout, d.wout))
                st.desc, st.pname, st.no_bias = d, n.name, n.attrs["no_bias"]
                wname = n.name + "_weight"
                if wname in self.params:  # weight sharing (ref: lightcnn.py:47-48 reused in the loop 52-69)
                    prev = self.params[wname]
                    if prev.mx_shape != (cout, c, kh, kw):
                        raise ValueError("shared parameter %s used with different shapes" % wname)
                else:
                    nw = d.n_pad16 * d.k_pad
                    self.params[wname] = ParamSpec(wname, "weight", offset, nw, d, (cout, c, kh, kw), st, rowpack)
                    offset += nw
                    if not st.no_bias:
                        self.params[n.name + "_bias"] = ParamSpec(n.name + "_bias", "bias", offset, d.n_pad16, d, (cout,), st)
                        offset += d.n_pad16
                use_count[wname] += 1
            elif n.op == "mfm":
                src = step_of[n.inputs[0].id]
                c, h, w = src.shape
                ways = n.attrs["ways"]
                if c % ways:
                    raise ValueError("MFM%d on %d channels" % (ways, c))
                st = Step("mfm", n, [src], (ops.mfm_out_channels(c, ways), h, w))
            elif n.op == "pool":
                src = step_of[n.inputs[0].id]
                c, h, w = src.shape
                st = Step("pool", n, [src], (c, h // 2, w // 2))
            elif n.op == "l2norm":
                src = step_of[n.inputs[0].id]
                if src.shape[1:] != (1, 1):
                    raise ValueError("l2norm expects a (B, C) feature")
                st = Step("l2norm", n, [src], src.shape)
            elif n.op == "add":
                a, bb = step_of[n.inputs[0].id], step_of[n.inputs[1].id]
                # fuse into the epilogue of whichever side is a single-consumer conv
                conv, other = None, None
                for cand, oth, node_in in ((bb, a, n.inputs[1]), (a, bb, n.inputs[0])):
                    if cand.op == "conv" and cand.residual is None and consumers[node_in.id] == 1:
                        conv, other = cand, oth
                        break
                if conv is None:
                    raise NotImplementedError("add is only supported as the residual of a single-consumer convolution")
                if conv.shape != other.shape:
                    raise ValueError("residual add of mismatching shapes %s vs %s" % (conv.shape, other.shape))
                conv.residual = other
                step_of[n.id] = conv
                continue
            else:
                raise NotImplementedError("op %s" % n.op)
            st.index = len(self.steps)
            self.steps.append(st)
            step_of[n.id] = st
        self.outputs = [step_of[o.id] for o in outputs]
        inputs = [s for s in self.steps if s.op == "input"]
        self._input_raw = bool(inputs) and all(u.op == "rowpack" for u in self.steps for i in u.inputs if i.op == "input") \
            and not any(o.op == "input" for o in self.outputs)
        self.num_flat = offset
        self._use_count = dict(use_count)
        # gradient need: everything downstream of a parameterised step; the raw input needs none
        for st in self.steps:
            if st.op == "input":
                st.needs_grad = False
            elif st.op == "conv":
                st.needs_grad = True
            else:
                st.needs_grad = any(i.needs_grad for i in st.inputs)
        self.max_dgrad_elems = max([s.desc.dn_pad16 * s.desc.dk_pad for s in self.steps
                                    if s.op == "conv" and s.inputs[0].needs_grad] or [0])
        # a residual source whose only other path needs a gradient
        self.flops_fwd = sum(2 * s.desc.batch * s.desc.hout * s.desc.wout * s.desc.cout * s.desc.cin * s.desc.kh * s.desc.kw
                             for s in self.steps if s.op == "conv")

    def _fuse(self):
        """conv -> MFM [-> pool 2x2] with single consumers becomes ONE step (the fused epilogue of efm_conv_mfm_fwd)."""
        users = collections.defaultdict(list)
        for st in self.steps:
            for i in st.inputs:
                users[i.index].append(st)
            if st.residual is not None:
                users[st.residual.index].append(st)
        out_idx = {st.index for st in self.outputs}
        dead, remap = set(), {}
        for m in self.steps:
            if m.op != "mfm":
                continue
            s = m.inputs[0]
            if s.op != "conv" or s.residual is not None or s.index in out_idx or users[s.index] != [m]:
                continue
            if getattr(s, "epi", None) is not None or not ops.conv_mfm_supported(s.desc):
                continue
            pool = None
            if m.index not in out_idx and len(users[m.index]) == 1 and users[m.index][0].op == "pool" \
                    and users[m.index][0].inputs[0] is m:
                pool = users[m.index][0]
            s.epi = {"ways": m.node.attrs["ways"], "order": m.node.attrs["order"], "pool": pool is not None,
                     "conv_shape": s.shape, "names": [m.node.name] + ([pool.node.name] if pool else [])}
            last = pool if pool is not None else m
            s.shape = last.shape
            dead.add(m.index)
            remap[m.index] = s
            if pool is not None:
                dead.add(pool.index)
                remap[pool.index] = s
            self.fused += 1
        if not dead:
            return
        for st in self.steps:
            st.inputs = [remap.get(i.index, i) for i in st.inputs]
            if st.residual is not None:
                st.residual = remap.get(st.residual.index, st.residual)
        self.outputs = [remap.get(o.index, o) for o in self.outputs]
        self.steps = [st for st in self.steps if st.index not in dead]
        for k, st in enumerate(self.steps):
            st.index = k

    def _check_bf16(self):
        users = collections.defaultdict(list)
        for st in self.steps:
            for i in st.inputs:
                users[i.index].append(st)
        outs = {st.index for st in self.outputs}
        for st in self.steps:
            if st.op == "pool":
                raise NotImplementedError("bf16 plan: stand-alone pooling '%s' (only pooling fused into a convolution)" % st.node.name)
            if st.op == "mfm" and (st.index in outs or any(u.op == "l2norm" for u in users[st.index])):
                raise NotImplementedError("bf16 plan: a stand-alone MFM may not feed the fp32 head")
            if st.op == "conv":
                # everything downstream of the L2 normalisation (the embedding head) stays fp32: fp32 kernels, fp32 weights
                st.f32 = st.inputs[0].op == "l2norm" or getattr(st.inputs[0], "f32", False)
                if st.f32:
                    continue
                feeds_f32 = st.index in outs or any(u.op == "l2norm" for u in users[st.index])
                if feeds_f32 and st.epi is None:
                    raise NotImplementedError("bf16 plan: a plain convolution may not feed the fp32 head")
                st.out_f32 = feeds_f32
                st.wb = st.wdb = None

    def _cast_weights(self, v):
        """bf16 copies of every weight (forward + data-gradient layouts) from the fp32 master buffer: once per step."""
        for st in self.steps:
            if st.op == "conv" and not st.f32:
                need_d = st.inputs[0].needs_grad
                st.wb, st.wdb = ops.convb_cast_weights(st.desc, v[st.pname + "_weight"], st.wb, st.wdb, need_dgrad=need_d)

    def _make_wino_u(self, v, train):
        """Transformed weights U = G g G^T of every Winograd layer — forward and (when training) data gradient — in ONE launch at the
        start of the step: the weights are fixed inside a step, and ~54 per-layer transforms were 8-microsecond kernels strung
        between the convolutions (0.45 ms of the step).  Buffers are per step-object and re-used."""
        jobs = []
        for st in self.steps:
            if st.op != "conv" or getattr(st, "f32", False):
                continue
            w = v[st.pname + "_weight"]
            if getattr(st, "wino_fwd", False):
                ways = st.epi["ways"] if st.epi is not None else 0
                n = ops.wino_u_numel(st.desc, False, ways)
                if getattr(st, "u_fwd", None) is None or st.u_fwd.numel() != n:
                    st.u_fwd = torch.empty((n,), dtype=torch.float32, device=self.device)
                jobs.append((st.desc, w, st.u_fwd, False, ways))
            if train and getattr(st, "wino_dgrad", False) and st.inputs[0].needs_grad:
                n = ops.wino_u_numel(st.desc, True, 0)
                if getattr(st, "u_dgrad", None) is None or st.u_dgrad.numel() != n:
                    st.u_dgrad = torch.empty((n,), dtype=torch.float32, device=self.device)
                jobs.append((st.desc, w, st.u_dgrad, True, 0))
        # an inference forward between a training forward and its backward leaves the data-gradient U in place (same weights)
        self._u_dgrad_ready = bool(train) or self._u_dgrad_ready
        ops.wino_make_u_batch(jobs)

    # --------------------------------------------------------------------------- parameters
    def new_flat(self):
        return torch.zeros(self.num_flat, dtype=torch.float32, device=self.device)

    def views(self, flat):
        """name -> view of `flat` in packed shape (cached per buffer)."""
        key = (flat.data_ptr(), flat.numel())
        v = self._views.get(key)
        if v is None:
            v = {}
            for name, ps in self.params.items():
                t = flat.narrow(0, ps.offset, ps.numel)
                v[name] = t.view(ps.desc.n_pad16, ps.desc.k_pad) if ps.kind == "weight" else t
            if len(self._views) > 8:
                self._views.clear()
            self._views[key] = v
        return v

    def load_params(self, flat, params):
        """params: name -> array in MXNet layout ((Cout,Cin,KH,KW) / (N, C*H*W) for fc / (Cout,))."""
        v = self.views(flat)
        for name, ps in self.params.items():
            if name not in params:
                raise KeyError("missing parameter %s" % name)
            a = torch.as_tensor(np.asarray(params[name]), dtype=torch.float32).to(self.device)
            if ps.kind == "weight":
                a = ps.to_packed_order(a.reshape(ps.mx_shape).contiguous())
                ops.conv_pack_weights_into(ps.desc, a, v[name])
            else:
                v[name].zero_()
                v[name][: ps.mx_shape[0]].copy_(a.reshape(-1))

    def export_params(self, flat):
        """-> name -> torch tensor in MXNet layout (fc weights as (N, C*H*W))."""
        v = self.views(flat)
        out = collections.OrderedDict()
        for name, ps in self.params.items():
            if ps.kind == "weight":
                w = ps.from_packed_order(ops.conv_unpack_weights(ps.desc, v[name]))
                if ps.step.node.op == "fc":
                    w = w.reshape(ps.mx_shape[0], -1)
                out[name] = w
            else:
                out[name] = v[name][: ps.mx_shape[0]].clone()
        return out

    def init_xavier(self, flat, seed=42, magnitude=3.0, factor_type="avg"):
        """Gluon init.Xavier(): uniform(+-sqrt(magnitude / factor)), factor = (fan_in+fan_out)/2 | fan_in | fan_out;
        biases zero (ref: train_efm.py:208; mutli_gpu_v3.py:156 uses factor_type='in', magnitude=2.34)."""
        gen = torch.Generator(device=self.device)
        gen.manual_seed(seed)
        v = self.views(flat)
        flat.zero_()
        for name, ps in self.params.items():
            if ps.kind != "weight":
                continue
            cout, cin, kh, kw = ps.mx_shape
            fan_in, fan_out = cin * kh * kw, cout * kh * kw
            factor = {"avg": (fan_in + fan_out) / 2.0, "in": fan_in, "out": fan_out}[factor_type]
            scale = float(np.sqrt(magnitude / factor))
            w = (torch.rand(ps.mx_shape, generator=gen, device=self.device, dtype=torch.float32) * 2 - 1) * scale
            ops.conv_pack_weights_into(ps.desc, ps.to_packed_order(w), v[name])

    # ------------------------------------------------------------------------------ forward
    def forward(self, x, flat, train=True):
        """x: (B, C, H, W) fp32 device tensor (NCHW, as ImageRecordIter emits).  Returns the output buffers:
        contiguous (B, pad4(C)) tensors for vector outputs (pads zero), NHWC tensors otherwise."""
        if tuple(x.shape) != self.input_shape:
            raise ValueError("plan was compiled for input %s, got %s" % (self.input_shape, tuple(x.shape)))
        v = self.views(flat)
        acts = {}
        aux = {}
        bf = self.dtype == "bf16"
        if bf:
            self._cast_weights(v)
        else:
            self._make_wino_u(v, train)
        for st in self.steps:
            if st.op == "input":
                if self._input_raw:   # only row-packing consumers: they read the NCHW tensor themselves
                    acts[st.index] = x.contiguous()
                else:
                    acts[st.index] = ops.nchw_to_nhwc_bf16(x.contiguous()) if bf else ops.nchw_to_nhwc(x.contiguous())
            elif st.op == "rowpack":
                acts[st.index] = ops.rowpack_nchw(x.contiguous(), st.kw, st.pad_w, bf16=bf)
            elif st.op == "conv" and bf and not st.f32:
                bias = None if st.no_bias else v[st.pname + "_bias"]
                src = acts[st.inputs[0].index]
                if st.epi is not None:
                    acts[st.index], aux[st.index] = ops.convb_mfm_fwd(st.desc, src, st.wb, bias, st.epi["ways"], st.epi["order"],
                                                                      st.epi["pool"], out_f32=st.out_f32)
                else:
                    res = acts[st.residual.index] if st.residual is not None else None
                    acts[st.index] = ops.convb_fwd(st.desc, src, st.wb, bias, res)
            elif st.op == "conv":
                w = v[st.pname + "_weight"]
                bias = None if st.no_bias else v[st.pname + "_bias"]
                if st.epi is not None:
                    e = st.epi
                    if getattr(st, "wino_fwd", False):
                        acts[st.index], aux[st.index] = ops.wino_mfm_fwd(st.desc, acts[st.inputs[0].index], st.u_fwd, bias, e["ways"],
                                                                         e["order"], e["pool"])
                    else:
                        acts[st.index], aux[st.index] = ops.conv_mfm_fwd(st.desc, acts[st.inputs[0].index], w, bias, e["ways"],
                                                                         e["order"], e["pool"])
                    continue
                res = acts[st.residual.index] if st.residual is not None else None
                if getattr(st, "wino_fwd", False):  # Winograd F(2x2,3x3): chosen by autotune() where it is faster
                    acts[st.index] = ops.wino_fwd(st.desc, acts[st.inputs[0].index], st.u_fwd, bias, res)
                else:
                    acts[st.index] = ops.conv_fwd(st.desc, acts[st.inputs[0].index], w, bias, res)
            elif st.op == "mfm":
                acts[st.index] = (ops.mfmb_fwd if bf else ops.mfm_fwd)(acts[st.inputs[0].index], st.inputs[0].shape[0], st.node.attrs["ways"])
            elif st.op == "pool":
                acts[st.index] = ops.maxpool2_fwd(acts[st.inputs[0].index], st.shape[0])
            elif st.op == "l2norm":
                src = acts[st.inputs[0].index]
                c = st.shape[0]
                cp = pad4(c)
                y = torch.zeros((self.batch, 1, 1, cp), dtype=torch.float32, device=self.device) if cp != c else \
                    torch.empty((self.batch, 1, 1, cp), dtype=torch.float32, device=self.device)
                norm = torch.empty((self.batch,), dtype=torch.float32, device=self.device)
                ops.check(ops._lib.load().efm_l2norm_fwd(ops._p(src), ops._p(y), ops._p(norm), self.batch, c, cp, cp, 0,
                                                         ops._stream()), "efm_l2norm_fwd")
                acts[st.index] = y
                aux[st.index] = norm
        if train:  # an inference forward (same-batch evaluation inside a step) leaves the saved state of the training forward alone
            self._acts, self._aux = acts, aux
        outs = []
        for st in self.outputs:
            t = acts[st.index]
            outs.append(t.view(self.batch, -1) if st.shape[1:] == (1, 1) else t)
        return outs

    # ----------------------------------------------------------------------------- backward
    def backward(self, out_grads, flat, grad_flat, ready_cb=None, need_input_grad=False):
        """out_grads: one tensor per plan output, same padded shape as forward() returned (pads must be zero),
        or None.  Writes d(loss)/d(params) into `grad_flat` (overwrite; shared parameters accumulate within the
        sweep).  Returns the NHWC input gradient when `need_input_grad`."""
        if self._acts is None:
            raise RuntimeError("backward() needs a forward(train=True) first")
        acts, aux = self._acts, self._aux
        v, gv = self.views(flat), self.views(grad_flat)
        gr = {}
        for st, g in zip(self.outputs, out_grads):
            if g is None:
                continue
            g = g.contiguous()
            if st.index in gr:
                raise NotImplementedError("the same step listed twice as an output")
            gr[st.index] = g
        written = set()
        remaining = dict(self._use_count)
        # Weight gradient and data gradient of a layer both consume dy and are independent: the weight gradients run on
        # a second HIP stream so that each kernel's last, partially filled round of blocks is covered by the other's.
        main = torch.cuda.current_stream(self.device) if self.device.type == "cuda" else None
        side = self._side_stream() if (main is not None and self.two_streams) else None
        red = None
        if side is not None and self.reduce_stream and self.dtype == "f32":
            red = self._reduce_stream()
            need = max(ops.conv_wgrad_workspace_bytes(s.desc) for s in self.steps if s.op == "conv")
            if self._ws2 is None or self._ws2[0].numel() * 4 < need:
                torch.cuda.synchronize(self.device)
                self._ws2 = [torch.empty(need // 4 + 1024, dtype=torch.float32, device=self.device) for _ in range(2)]
            self._ws_free = [None, None]
            self._ws_turn = 0
            self._ev_next = 0
        if self.max_dgrad_elems and (self._wd_scratch is None or self._wd_scratch.numel() < self.max_dgrad_elems):
            self._wd_scratch = torch.empty(self.max_dgrad_elems, dtype=torch.float32, device=self.device)
        dx_input = None
        for st in reversed(self.steps):
            dy = gr.pop(st.index, None)
            if dy is None:
                continue
            if st.op == "conv":
                d = st.desc
                src = st.inputs[0]
                bf = self.dtype == "bf16" and not st.f32
                wgrad = ops.convb_bwd_weight if bf else ops.conv_bwd_weight
                wname = st.pname + "_weight"
                acc = wname in written
                dwv, dbv = gv[wname], (None if st.no_bias else gv[st.pname + "_bias"])
                wants_dx = src.needs_grad or (src.op in ("input", "rowpack") and need_input_grad)
                if (st.epi is not None and bf and not wants_dx and st.residual is None and dy.dtype == torch.bfloat16
                        and self.fused_wgrad and ops.convb_mfm_bwd_weight_supported(d, st.epi["ways"], st.epi["pool"])):
                    # the only consumer of this layer's conv-output gradient is its own weight gradient (first convolution): the
                    # kernel forms it in LDS from dz + the route bytes; nothing of it touches HBM
                    route, dz, e = aux[st.index], dy, st.epi
                    wgrad = lambda d_, x_, _dy, **kw: ops.convb_mfm_bwd_weight(d_, x_, route, dz, e["ways"], e["pool"], **kw)  # noqa: E731
                elif st.epi is not None:  # gradient of the fused MFM (+ pool) epilogue -> full conv-output gradient
                    dy = (ops.convb_mfm_pool_bwd if bf else ops.mfm_pool_bwd)(d, aux[st.index], dy, st.epi["ways"], st.epi["pool"])
                if side is not None and red is not None and not bf:
                    # three streams: slabs of this layer on `side`; their reduction on `red`, i.e. under the NEXT layer's slabs
                    # kernel.  Two workspaces alternate; a workspace is written again only after the reduction that read it.
                    k = self._ws_turn
                    self._ws_turn ^= 1
                    side.wait_stream(main)
                    dy.record_stream(side)
                    with torch.cuda.stream(side):
                        if self._ws_free[k] is not None:
                            side.wait_event(self._ws_free[k])
                        ops.conv_bwd_weight_slabs(d, acts[src.index], dy, self._ws2[k], want_bias=not st.no_bias)
                        ev = self._event()
                        ev.record(side)
                    with torch.cuda.stream(red):
                        red.wait_event(ev)
                        ops.conv_bwd_weight_finish(d, self._ws2[k], dwv, dbv, accumulate=acc)
                        done = self._event()
                        done.record(red)
                        self._ws_free[k] = done
                elif side is not None:
                    side.wait_stream(main)
                    dy.record_stream(side)
                    with torch.cuda.stream(side):
                        wgrad(d, acts[src.index], dy, dw=dwv, dbias=dbv, want_bias=not st.no_bias, accumulate=acc)
                else:
                    wgrad(d, acts[src.index], dy, dw=dwv, dbias=dbv, want_bias=not st.no_bias, accumulate=acc)
                written.add(wname)
                if st.residual is not None:
                    r = st.residual.index
                    if r in gr:
                        raise NotImplementedError("two gradient contributions to a residual source before its own backward")
                    gr[r] = dy  # identity path: alias, consumed (read-only) by the source's other consumer
                if bf and src.needs_grad:
                    prev = gr.pop(src.index, None)
                    gr[src.index] = ops.convb_bwd_data(d, dy, st.wdb, add=prev)
                elif getattr(st, "wino_dgrad", False) and src.needs_grad:
                    if not self._u_dgrad_ready:  # forward(train=False) followed by backward is not a sequence this plan runs
                        raise RuntimeError("backward() needs forward(train=True)")
                    prev = gr.pop(src.index, None)
                    gr[src.index] = ops.wino_bwd_data(d, dy, st.u_dgrad, add=prev)
                elif src.needs_grad or (src.op in ("input", "rowpack") and need_input_grad):
                    wd = self._wd_scratch[: d.dn_pad16 * d.dk_pad]
                    ops.conv_make_dgrad_weights(d, v[wname], out=wd)
                    prev = gr.pop(src.index, None)
                    gr[src.index] = ops.conv_bwd_data(d, dy, wd, add=prev)
                remaining[wname] -= 1
                if ready_cb is not None and remaining[wname] == 0:
                    ps = self.params[wname]
                    hi = ps.offset + ps.numel + (0 if st.no_bias else d.n_pad16)
                    cb_stream = red if (red is not None and self.dtype != "bf16") else side
                    if cb_stream is not None:
                        with torch.cuda.stream(cb_stream):  # the collective must order after the stream that finishes the gradient
                            ready_cb(ps.offset, hi)
                    else:
                        ready_cb(ps.offset, hi)
            elif st.op == "mfm":
                src = st.inputs[0]
                prev = gr.pop(src.index, None)
                gr[src.index] = (ops.mfmb_bwd if self.dtype == "bf16" else ops.mfm_bwd)(
                    acts[src.index], dy, src.shape[0], st.node.attrs["ways"], st.node.attrs["order"], add=prev)
            elif st.op == "pool":
                src = st.inputs[0]
                if src.index in gr:
                    raise NotImplementedError("pool input with a second gradient contribution")
                gr[src.index] = ops.maxpool2_bwd(acts[src.index], dy, src.shape[0])
            elif st.op == "l2norm":
                src = st.inputs[0]
                c = st.shape[0]
                cp = pad4(c)
                if src.index in gr:
                    raise NotImplementedError("l2norm input with a second gradient contribution")
                dx = torch.zeros((self.batch, 1, 1, cp), dtype=torch.float32, device=self.device) if cp != c else \
                    torch.empty((self.batch, 1, 1, cp), dtype=torch.float32, device=self.device)
                ops.check(ops._lib.load().efm_l2norm_bwd(ops._p(acts[st.index]), ops._p(aux[st.index]), ops._p(dy), ops._p(dx),
                                                         self.batch, c, cp, cp, cp, 0, ops._stream()), "efm_l2norm_bwd")
                gr[src.index] = dx
            elif st.op == "rowpack":
                raise NotImplementedError("input gradient through the row-packed first convolution (build the plan with EFM_ROWPACK=0)")
            elif st.op == "input":
                dx_input = dy
        if side is not None:
            main.wait_stream(side)
        if red is not None:
            main.wait_stream(red)
        return dx_input

    def autotune(self, iters=5, verbose=False):
        """Time the tiling candidates (64/128-pixel tiles x 1..3 channel blocks) of every plain convolution forward and
        data gradient at this plan's shapes and store the winners in the descriptors (results are bit-identical for every
        choice).  For the 3x3 / pad 1 layers the Winograd F(2x2,3x3) kernels (forward, data gradient, weight gradient) are timed
        against the winner and taken where at least 3 % faster — those layers then differ from the direct kernels by fp32
        rounding (~1e-6), not bitwise; EFM_WINO=0 keeps the direct kernels everywhere.  ~1 s at B = 256; idempotent."""
        if self.device.type != "cuda":
            return {}
        chosen = {}

        def best(run, d, field, cands=None):
            cands = cands or ([0] + [mt | (ns << 4) for mt in (1, 2) for ns in (1, 2, 3)])
            times = {}
            for c in cands:
                setattr(d, field, c)
                run()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters):
                    run()
                e1.record()
                e1.synchronize()
                times[c] = e0.elapsed_time(e1)
            win = min(times, key=times.get)
            if times[win] > 0.97 * times[0]:
                win = 0  # keep the heuristic unless a candidate is clearly better
            setattr(d, field, win)
            return win, times

        def timed(run):
            run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                run()
            e1.record()
            e1.synchronize()
            return e0.elapsed_time(e1)

        wino = self.dtype == "f32" and os.environ.get("EFM_WINO", "1") != "0"
        force = os.environ.get("EFM_WINO") == "force"  # tests: Winograd wherever it applies, whatever the clock says
        # data gradients run next to the weight-gradient kernels of the side stream: which Winograd variants to time for them
        dgrad_vars = tuple(int(v) for v in os.environ.get("EFM_WINO_DGRAD_VARIANTS", "1,2").split(","))
        seen = {}
        for st in self.steps:
            if st.op != "conv":
                continue
            d = st.desc
            key = (d.hin, d.win, d.cin, d.cout, d.kh, d.pad_h, st.epi is not None, st.inputs[0].needs_grad)
            if key in seen:
                d.tune_fwd, d.tune_dgrad, st.wino_fwd, st.wino_dgrad, d.tune_wgrad = seen[key]
                continue
            st.wino_fwd = st.wino_dgrad = False
            x = torch.rand((d.batch, d.hin, d.win, d.cin_p), device=self.device)
            dy = torch.rand((d.batch, d.hout, d.wout, d.cout_p), device=self.device)
            if st.epi is None:
                w = torch.rand((d.n_pad16, d.k_pad), device=self.device)
                y = torch.empty_like(dy)
                win, times = best(lambda: ops.conv_fwd(d, x, w, None, out=y), d, "tune_fwd")
                chosen[st.pname + ":fwd"] = win
                if wino and ops.wino_supported(d):
                    tw = {}
                    for var in (1, 2):  # 8-wave / 4-wave kernel (bits 9:8 of the tune field; U's layout follows it)
                        d.tune_fwd = var << 8
                        u = ops.wino_make_u(d, w)
                        tw[var] = timed(lambda: ops.wino_fwd(d, x, u, None, out=y))
                    var = min(tw, key=tw.get)
                    if tw[var] < 0.97 * times[win] or force:
                        st.wino_fwd = True
                        d.tune_fwd = var << 8
                        chosen[st.pname + ":fwd"] = "winograd/%d" % (8 if var == 1 else 4)
                    else:
                        d.tune_fwd = win
            else:  # fused epilogue: number of channel blocks
                w = torch.rand((d.n_pad16, d.k_pad), device=self.device)
                e = st.epi
                win, times = best(lambda: ops.conv_mfm_fwd(d, x, w, None, e["ways"], e["order"], e["pool"]), d,
                                  "tune_fwd", [0, 1 << 4, 2 << 4, 3 << 4, 2, 2 | (1 << 4), 2 | (2 << 4)])
                chosen[st.pname + ":fused"] = win
                if wino and ops.wino_supported(d):
                    tw = {}
                    for var in (1, 2):
                        d.tune_fwd = var << 8
                        u = ops.wino_mfm_make_u(d, w, e["ways"])
                        tw[var] = timed(lambda: ops.wino_mfm_fwd(d, x, u, None, e["ways"], e["order"], e["pool"]))
                    var = min(tw, key=tw.get)
                    if tw[var] < 0.97 * times[win] or force:
                        st.wino_fwd = True
                        d.tune_fwd = var << 8
                        chosen[st.pname + ":fused"] = "winograd/%d" % (8 if var == 1 else 4)
                    else:
                        d.tune_fwd = win
            if st.inputs[0].needs_grad:
                wd = torch.rand((d.dn_pad16, d.dk_pad), device=self.device)
                dx = torch.empty_like(x)
                win, times = best(lambda: ops.conv_bwd_data(d, dy, wd, out=dx), d, "tune_dgrad")
                chosen[st.pname + ":dgrad"] = win
                if wino and ops.wino_supported(d):
                    tw = {}
                    wf = torch.rand((d.n_pad16, d.k_pad), device=self.device)
                    for var in dgrad_vars:
                        d.tune_dgrad = var << 8
                        u = ops.wino_make_u(d, wf, dgrad=True)
                        tw[var] = timed(lambda: ops.wino_bwd_data(d, dy, u, out=dx))
                    var = min(tw, key=tw.get)
                    if tw[var] < 0.97 * times[win] or force:
                        st.wino_dgrad = True
                        d.tune_dgrad = var << 8
                        chosen[st.pname + ":dgrad"] = "winograd/%d" % (8 if var == 1 else 4)
                    else:
                        d.tune_dgrad = win
            # weight gradient: 64 / 128 weight columns per block x the number of pixel splits (summation order differs, nothing else)
            dwt = torch.empty((d.n_pad16, d.k_pad), device=self.device)
            dbt = torch.empty((d.n_pad16,), device=self.device)
            cw = [0] + [kpw | ((blocks // 64) << 4) for kpw in (1, 2) for blocks in (768, 1024, 1536, 2560, 3840)]
            if wino and ops.wino_supported(d) and os.environ.get("EFM_WINO_WGRAD", "1") != "0":
                # bit 12: the Winograd form (transforms on the way from LDS to the matrix cores); bits 9:4 = blocks / 64
                # bits 3:0 = 1 + channel-tile shape of the block (0: the one that pads this layer least)
                cw += [0x1000 | (4 << 4), 0x1000 | (8 << 4)] + [0x1000 | (4 << 4) | sh for sh in range(1, 10)]
            chosen[st.pname + ":wgrad"] = best(lambda: ops.conv_bwd_weight(d, x, dy, dw=dwt, dbias=dbt), d, "tune_wgrad", cw)[0]
            seen[key] = (d.tune_fwd, d.tune_dgrad, st.wino_fwd, st.wino_dgrad, d.tune_wgrad)
        if verbose:
            print("[efm autotune]", {k: v for k, v in chosen.items() if v})
        self.chosen = chosen
        return chosen

    # The kernel selection as data: what autotune() decided can be written out, committed and applied again, so that the
    # selection a benchmark times is exactly the selection the parity tests exercise (timing-based choices differ run to run).
    def tuning_table(self):
        """{layer name: {tune_fwd, tune_dgrad, tune_wgrad, wino_fwd, wino_dgrad}} of every convolution step (JSON-able)."""
        table = collections.OrderedDict()
        for st in self.steps:
            if st.op == "conv":
                d = st.desc
                table[st.pname] = {"tune_fwd": int(d.tune_fwd), "tune_dgrad": int(d.tune_dgrad), "tune_wgrad": int(d.tune_wgrad),
                                   "wino_fwd": bool(getattr(st, "wino_fwd", False)), "wino_dgrad": bool(getattr(st, "wino_dgrad", False))}
        return table

    def apply_tuning(self, table):
        """Install a table written by tuning_table() (layers it does not name keep the heuristics).  Every choice is a valid
        kernel for every batch size; the Winograd flags are honoured only where the Winograd kernels apply."""
        n = 0
        for st in self.steps:
            if st.op != "conv" or st.pname not in table:
                continue
            t, d = table[st.pname], st.desc
            if isinstance(t, (list, tuple)):  # the compact row form bench.py prints
                t = dict(zip(("tune_fwd", "tune_dgrad", "tune_wgrad", "wino_fwd", "wino_dgrad"), t))
            wino_ok = self.dtype == "f32" and ops.wino_supported(d)
            st.wino_fwd = bool(t.get("wino_fwd")) and wino_ok
            st.wino_dgrad = bool(t.get("wino_dgrad")) and wino_ok
            d.tune_fwd = int(t.get("tune_fwd", 0)) if (st.wino_fwd or not t.get("wino_fwd")) else 0
            d.tune_dgrad = int(t.get("tune_dgrad", 0)) if (st.wino_dgrad or not t.get("wino_dgrad")) else 0
            d.tune_wgrad = int(t.get("tune_wgrad", 0))
            for attr in ("u_fwd", "u_dgrad"):  # U's layout follows the Winograd variant
                if hasattr(st, attr):
                    delattr(st, attr)
            n += 1
        return n

    def _side_stream(self):
        if self._side is None:
            self._side = _shared_stream(self.device, "side")
        return self._side

    def _reduce_stream(self):
        if self._red is None:
            self._red = _shared_stream(self.device, "reduce")
        return self._red

    def _event(self):
        """Events from a per-plan pool (two per convolution and backward: creating them anew costs host time every step)."""
        if self._ev_next == len(self._ev_pool):
            self._ev_pool.append(torch.cuda.Event())
        ev = self._ev_pool[self._ev_next]
        self._ev_next += 1
        return ev

    def routing_inputs(self):
        """{MFM / pooling node name -> its INPUT activation as an NCHW torch tensor} of the last forward(train=True);
        lets a higher-precision checker follow the same arg-max routes (tests only)."""
        if self.fused:
            raise RuntimeError("routing_inputs() needs an unfused plan (Plan(..., fuse=False)): fused epilogues never "
                               "materialise the MFM inputs")
        out = {}
        for st in self.steps:
            if st.op in ("mfm", "pool"):
                src = st.inputs[0]
                t = self._acts[src.index]
                c, h, w = src.shape
                out[st.node.name] = ops.nhwc_to_nchw(t.view(self.batch, h, w, pad4(c)), c)
        return out

    def release(self):
        self._acts = self._aux = None
