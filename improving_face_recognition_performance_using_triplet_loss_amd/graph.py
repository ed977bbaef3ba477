"""Symbolic network description — the host-side mirror of the reference's Symbol API.

The reference builds its EFM network as an MXNet symbol graph (`mx.symbol.Convolution`, `SliceChannel`,
`maximum`, `minimum`, `Concat`, `Pooling`, `FullyConnected`: efm_symbol.py:22-110).  This module is the
equivalent description layer of this framework: a tiny DAG of `Sym` nodes that `plan.compile()` lowers onto
the HIP kernels.  It is coarser than MXNet's on purpose — the 6-operator SliceChannel/max/min/Concat idiom is
ONE node (`MFM`), because that is the unit the kernels fuse.
"""
import itertools

_ids = itertools.count()

ORDER_GROUP = 0  # maximum(maximum(s0,s1), s2)   ref: efm_symbol.py:70-73
ORDER_RES = 1    # maximum(s2, maximum(s0,s1))   ref: efm_symbol.py:26-29


class Sym:
    """One node of the network DAG."""

    def __init__(self, op, inputs=(), name=None, **attrs):
        self.id = next(_ids)
        self.op = op
        self.inputs = list(inputs)
        self.name = name or "%s%d" % (op, self.id)
        self.attrs = attrs

    def __add__(self, other):
        # ref: efm_symbol.py:42 `conv_r0 = data + conv_r1`
        return Sym("add", [self, other])

    def __repr__(self):
        return "Sym(%s, %s, in=%s)" % (self.op, self.name, [i.name for i in self.inputs])

    def list_arguments(self):
        """Parameter names in topological order (mirrors mx.sym.Symbol.list_arguments)."""
        out = []
        for n in topo_sort([self]):
            if n.op == "var":
                out.append(n.name)
            elif n.op in ("conv", "fc"):
                out.append(n.name + "_weight")
                if not n.attrs.get("no_bias"):
                    out.append(n.name + "_bias")
        return list(dict.fromkeys(out))


def Variable(name):
    return Sym("var", name=name)


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def Convolution(data, num_filter, kernel, name, pad=(0, 0), stride=(1, 1), no_bias=False):
    """ref: mx.symbol.Convolution (efm_symbol.py:32,41,54,62,65,67). Stride must be 1 (every call site's is)."""
    if _pair(stride) != (1, 1):
        raise NotImplementedError("only stride 1 convolutions exist on the reference hot path")
    return Sym("conv", [data], name=name, num_filter=int(num_filter), kernel=_pair(kernel), pad=_pair(pad),
               no_bias=bool(no_bias))


def MFM(data, ways=3, order=ORDER_GROUP, name=None):
    """SliceChannel(ways) + maximum/minimum + Concat (ref: efm_symbol.py:25-30,63-64,69-77)."""
    return Sym("mfm", [data], name=name, ways=int(ways), order=int(order))


def Pooling(data, name=None):
    """max, 2x2, stride 2, 'valid' (ref: efm_symbol.py:78)."""
    return Sym("pool", [data], name=name)


def FullyConnected(data, num_hidden, name, no_bias=False):
    """Flatten + FullyConnected (ref: efm_symbol.py:93-94,104). Flatten order is NCHW, as MXNet's."""
    return Sym("fc", [data], name=name, num_hidden=int(num_hidden), no_bias=bool(no_bias))


def L2Normalization(data, name=None):
    """Per-row unit vectors (ref: final_efm.py:240-243), differentiable."""
    return Sym("l2norm", [data], name=name)


def topo_sort(outputs):
    seen, order = set(), []

    def visit(n):
        if n.id in seen:
            return
        seen.add(n.id)
        for i in n.inputs:
            visit(i)
        order.append(n)

    for o in outputs:
        visit(o)
    return order
