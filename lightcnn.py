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
            x = G.MFM(x, 3, G.ORDER_GROUP)
        x = G.Convolution(x, self.num_filter1, self.kernel_size, name=self.prefix + "_conv1", pad=self.padding, stride=self.stride)
        return G.MFM(x, 3, G.ORDER_GROUP)


class res_block:
    """num_blocks x [EFM -> conv3x3(num_filter) -> EFM -> conv3x3(2/3 num_filter) -> + input], the SAME two convolutions
    every time (ref: lightcnn.py:41-71)."""

    def __init__(self, num_blocks, num_filter, prefix=None):
        self.num_blocks, self.num_filter = num_blocks, num_filter
        self.num_filter1 = int(num_filter * (2. / 3.))
        self.prefix = prefix or _name("res")

    def __call__(self, x):
        for _ in range(self.num_blocks):
            e = G.MFM(x, 3, G.ORDER_GROUP)
            c1 = G.Convolution(e, self.num_filter, (3, 3), name=self.prefix + "_conv0", pad=(1, 1))
            e = G.MFM(c1, 3, G.ORDER_GROUP)
            c2 = G.Convolution(e, self.num_filter1, (3, 3), name=self.prefix + "_conv1", pad=(1, 1))
            x = c2 + x
        return x


def lightcnn29_feature(fc_units=1026):
    data = G.Variable("data")
    num_blocks = [1, 2, 3, 4]
    x = G.Pooling(efm(0, 99, (5, 5), (1, 1), (2, 2), 0, "g1")(data))
    for i, (nb, nf, nf1) in enumerate(zip(num_blocks, (99, 198, 387, 261), (198, 387, 261, 261))):
        x = res_block(nb, nf, "g%d_res" % (i + 2))(x)
        x = G.Pooling(efm(nf, nf1, (3, 3), (1, 1), (1, 1), 1, "g%d" % (i + 2))(x))
    fc1 = G.FullyConnected(x, fc_units, name="fc1")
    return G.MFM(fc1, 3, G.ORDER_GROUP, name="efm_fc1")


class LightCNN_29(torch.nn.Module):
    def __init__(self, num_classes, in_channels=1, image=128, device="cuda", seed=42):
        super().__init__()
        self.conv_net = SymbolNet([lightcnn29_feature()], in_channels, image, device=device, seed=seed)
        self.fc1 = torch.nn.BatchNorm1d(684, eps=1e-5, momentum=0.1).to(device)  # Gluon momentum .9 == torch .1
        self.fc2 = torch.nn.Sequential(torch.nn.Dropout(0.7), torch.nn.Linear(684, num_classes)).to(device)
        torch.nn.init.xavier_uniform_(self.fc2[1].weight)
        torch.nn.init.zeros_(self.fc2[1].bias)

    def forward(self, x):
        (feat,) = self.conv_net(x)
        return self.fc2(feat), self.fc1(feat)

    def save_parameters(self, path):
        """MXNet NDArray-list file, MXNet layouts ((cout,cin,kh,kw), Dense (units,in_units)), Gluon-style names."""
        from improving_face_recognition_performance_using_triplet_loss_amd import mxio
        p = {k: v.cpu().numpy() for k, v in self.conv_net.export_params().items()}
        p["batchnorm0_gamma"], p["batchnorm0_beta"] = self.fc1.weight.detach().cpu().numpy(), self.fc1.bias.detach().cpu().numpy()
        p["batchnorm0_running_mean"], p["batchnorm0_running_var"] = self.fc1.running_mean.cpu().numpy(), self.fc1.running_var.cpu().numpy()
        p["dense1_weight"], p["dense1_bias"] = self.fc2[1].weight.detach().cpu().numpy(), self.fc2[1].bias.detach().cpu().numpy()
        mxio.save_params(path, p)

    def load_parameters(self, path):
        from improving_face_recognition_performance_using_triplet_loss_amd import mxio
        p = mxio.load_params(path)
        self.conv_net.load_params({k: v for k, v in p.items() if not k.startswith(("batchnorm0_", "dense1_"))})
        with torch.no_grad():
            for dst, k in ((self.fc1.weight, "batchnorm0_gamma"), (self.fc1.bias, "batchnorm0_beta"),
                           (self.fc1.running_mean, "batchnorm0_running_mean"), (self.fc1.running_var, "batchnorm0_running_var"),
                           (self.fc2[1].weight, "dense1_weight"), (self.fc2[1].bias, "dense1_bias")):
                dst.copy_(torch.as_tensor(p[k]))
