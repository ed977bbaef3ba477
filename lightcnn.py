"""Drop-in for the reference's lightcnn.py: `efm`, `res_block`, `LightCNN_29(num_classes)` with `net(x) -> (out, fc1_out)`.

The Gluon blocks of the reference (lightcnn.py:6-133) become builders over `graph.Sym`; LightCNN_29 is a torch module whose
convolutional trunk is ONE compiled plan on the HIP kernels.  Faithful to the Gluon variant: the two convolutions of a
res_block are created once and RE-APPLIED `num_blocks` times (weight sharing, :47-48,52-69), fc1 is Dense(1026) -> 684-d,
`fc1_out = BatchNorm(feature)`, `out = Dense(classes)(Dropout(.7)(feature))` (:113-118,130-133).
BatchNorm / Dropout / the id-head GEMM are plain torch ops (not on the north-star kernel list).
"""
import torch

from improving_face_recognition_performance_using_triplet_loss_amd import graph as G
from improving_face_recognition_performance_using_triplet_loss_amd.nn import SymbolNet

_uid = [0]


def _name(prefix):
    _uid[0] += 1
    return "%s%d" % (prefix, _uid[0])


class efm:
    """[conv1x1(num_filter) -> EFM ->] conv kxk(num_filter1) -> EFM   (ref: lightcnn.py:6-39)."""

    def __init__(self, num_filter, num_filter1, kernel_size, stride, padding, efm_type, prefix=None):
        self.num_filter, self.num_filter1 = num_filter, num_filter1
        self.kernel_size, self.stride, self.padding, self.efm_type = kernel_size, stride, padding, efm_type
        self.prefix = prefix or _name("efm")

    def __call__(self, x):
        if self.efm_type == 1:
            x = G.Convolution(x, self.num_filter, (1, 1), name=self.prefix + "_conv0")
            x = G.MFM(x, 3, G.ORDER_GROUP, name=self.prefix + "_efm0")
        x = G.Convolution(x, self.num_filter1, self.kernel_size, name=self.prefix + "_conv1", pad=self.padding, stride=self.stride)
        return G.MFM(x, 3, G.ORDER_GROUP, name=self.prefix + ("_efm1" if self.efm_type == 1 else "_efm"))


class res_block:
    """num_blocks x [EFM -> conv3x3(num_filter) -> EFM -> conv3x3(2/3 num_filter) -> + input], the SAME two convolutions
    every time (ref: lightcnn.py:41-71)."""

    def __init__(self, num_blocks, num_filter, prefix=None):
        self.num_blocks, self.num_filter = num_blocks, num_filter
        self.num_filter1 = int(num_filter * (2. / 3.))
        self.prefix = prefix or _name("res")

    def __call__(self, x):
        for i in range(self.num_blocks):
            e = G.MFM(x, 3, G.ORDER_GROUP, name="%s%d_efm_in" % (self.prefix, i))
            c1 = G.Convolution(e, self.num_filter, (3, 3), name=self.prefix + "_conv0", pad=(1, 1))
            e = G.MFM(c1, 3, G.ORDER_GROUP, name="%s%d_efm" % (self.prefix, i))
            c2 = G.Convolution(e, self.num_filter1, (3, 3), name=self.prefix + "_conv1", pad=(1, 1))
            x = c2 + x
        return x


def lightcnn29_feature(fc_units=1026):
    data = G.Variable("data")
    num_blocks = [1, 2, 3, 4]
    x = G.Pooling(efm(0, 99, (5, 5), (1, 1), (2, 2), 0, "g1")(data), name="g1_pool")
    for i, (nb, nf, nf1) in enumerate(zip(num_blocks, (99, 198, 387, 261), (198, 387, 261, 261))):
        x = res_block(nb, nf, "g%d_res" % (i + 2))(x)
        x = G.Pooling(efm(nf, nf1, (3, 3), (1, 1), (1, 1), 1, "g%d" % (i + 2))(x), name="g%d_pool" % (i + 2))
    fc1 = G.FullyConnected(x, fc_units, name="fc1")
    return G.MFM(fc1, 3, G.ORDER_GROUP, name="efm_fc1")


def gluon_param_names():
    """this build's parameter name -> the key Gluon's `net.save_parameters` writes for the reference's LightCNN_29: the attribute
    path of the parameter, Sequential children by index (ref: lightcnn.py:79-118 fixes the indices: 0 efm, 1 pool, 2 res_block,
    3 efm, 4 pool, ... 14 Flatten, 15 Dense(1026); fc1 = [BatchNorm], fc2 = [Dropout, Dense])."""
    m = {"g1_conv1": "conv_net.0.conv_op_2", "fc1": "conv_net.15"}
    for gi in range(4):
        g, base = gi + 2, 2 + 3 * gi
        m["g%d_res_conv0" % g], m["g%d_res_conv1" % g] = "conv_net.%d.conv_op_1" % base, "conv_net.%d.conv_op_2" % base
        m["g%d_conv0" % g], m["g%d_conv1" % g] = "conv_net.%d.conv_op_1" % (base + 1), "conv_net.%d.conv_op_2" % (base + 1)
    out = {}
    for k, v in m.items():
        out[k + "_weight"], out[k + "_bias"] = v + ".weight", v + ".bias"
    for k in ("gamma", "beta", "running_mean", "running_var"):
        out["batchnorm0_" + k] = "fc1.0." + k
    out["dense1_weight"], out["dense1_bias"] = "fc2.1.weight", "fc2.1.bias"
    return out


class LightCNN_29(torch.nn.Module):
    def __init__(self, num_classes, in_channels=1, image=128, device="cuda", seed=42, dropout=0.7, fuse=None, autotune=None):
        super().__init__()
        from improving_face_recognition_performance_using_triplet_loss_amd.nn import BatchNorm
        self.conv_net = SymbolNet([lightcnn29_feature()], in_channels, image, device=device, seed=seed, fuse=fuse, autotune=autotune)
        self.fc1 = torch.nn.Sequential(BatchNorm(684)).to(device)
        self.fc2 = torch.nn.Sequential(torch.nn.Dropout(dropout), torch.nn.Linear(684, num_classes)).to(device)
        torch.nn.init.xavier_uniform_(self.fc2[1].weight)
        torch.nn.init.zeros_(self.fc2[1].bias)

    def forward(self, x):
        (feat,) = self.conv_net(x)
        return self.fc2(feat), self.fc1(feat)

    def named_params_mx(self):
        """every parameter / BatchNorm statistic in MXNet layout under this build's names."""
        p = {k: v for k, v in self.conv_net.export_params().items()}
        bn = self.fc1[0]
        p["batchnorm0_gamma"], p["batchnorm0_beta"] = bn.gamma.detach(), bn.beta.detach()
        p["batchnorm0_running_mean"], p["batchnorm0_running_var"] = bn.running_mean, bn.running_var
        p["dense1_weight"], p["dense1_bias"] = self.fc2[1].weight.detach(), self.fc2[1].bias.detach()
        return p

    def save_parameters(self, path):
        """MXNet NDArray-list file, MXNet layouts ((cout,cin,kh,kw), Dense (units,in_units)), keyed the way Gluon's
        `net.save_parameters` keys them (ref: train_efm.py:289-290)."""
        from improving_face_recognition_performance_using_triplet_loss_amd import mxio
        names = gluon_param_names()
        mxio.save_params(path, {names[k]: v.cpu().numpy() for k, v in self.named_params_mx().items()})

    def set_params_mx(self, p):
        self.conv_net.load_params({k: v for k, v in p.items() if not k.startswith(("batchnorm0_", "dense1_"))})
        bn = self.fc1[0]
        with torch.no_grad():
            for dst, k in ((bn.gamma, "batchnorm0_gamma"), (bn.beta, "batchnorm0_beta"),
                           (bn.running_mean, "batchnorm0_running_mean"), (bn.running_var, "batchnorm0_running_var"),
                           (self.fc2[1].weight, "dense1_weight"), (self.fc2[1].bias, "dense1_bias")):
                if k in p:
                    dst.copy_(torch.as_tensor(p[k], dtype=torch.float32))

    def load_parameters(self, path):
        from improving_face_recognition_performance_using_triplet_loss_amd import mxio
        inv = {v: k for k, v in gluon_param_names().items()}
        self.set_params_mx({inv.get(k, k): v for k, v in mxio.load_params(path).items()})
