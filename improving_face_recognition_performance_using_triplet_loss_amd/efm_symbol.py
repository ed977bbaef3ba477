"""EFM-29 network definition with the reference's builder names and signatures.

Drop-in for the net-builder half of the reference's efm_symbol.py (res_block :22-44, group :46-79,
multi_gpu :81-110, get_net :112-123) — same call signatures, same parameter names
(`conv1_weight`, `conv2_res_weight`, `conv31_res_r_bias`, `fc1_weight`, ...), but the nodes are
`graph.Sym`s that `plan.Plan` lowers onto HIP kernels instead of MXNet symbols.
"""
from . import graph as G


def _mfm(data, channels, order, name):
    # 3-way EFM when the channel count divides by 3, LightCNN's 2-way MFM otherwise (ref: efm_symbol.py:48,68)
    return G.MFM(data, 3 if channels % 3 == 0 else 2, order, name=name)


def res_block(data, num_r, layer):
    """data + conv3x3_{->2num_r/3}(EFM(conv3x3_{->num_r}(EFM(data))))  (ref: efm_symbol.py:22-44)."""
    num_r1 = int(num_r * (2. / 3.))
    e = G.MFM(data, 3, G.ORDER_RES, name="efm%s_res_in" % layer)
    conv_r = G.Convolution(e, num_r, (3, 3), name="conv%s_res" % layer, pad=(1, 1))
    e = G.MFM(conv_r, 3, G.ORDER_RES, name="efm%s_res" % layer)
    conv_r1 = G.Convolution(e, num_r1, (3, 3), name="conv%s_res_r" % layer, pad=(1, 1))
    return data + conv_r1


def group(data, num_r, num, kernel, stride, pad, layer, tar_num=0):
    """[res_block x tar_num -> conv1x1(num_r) -> MFM] -> conv kxk(num) -> MFM -> maxpool 2x2 (ref: efm_symbol.py:46-79)."""
    if num_r > 0:
        res = data
        if num_r % 3 == 0:
            for x in range(tar_num):
                res = res_block(res, num_r, layer if x == 0 else layer + str(x))
        conv_r = G.Convolution(res, num_r, (1, 1), name="conv%s_r" % layer)
        data = _mfm(conv_r, num_r, G.ORDER_RES, "efm%s_r" % layer)
    conv = G.Convolution(data, num, kernel, name="conv%s" % layer, pad=pad, stride=stride)
    mfm = _mfm(conv, num, G.ORDER_GROUP, "efm%s" % layer)
    return G.Pooling(mfm, name="pool%s" % layer)


def efm_feature(data, fc_hidden=513):
    """Five groups -> fc1 -> EFM: returns (342-d feature, raw fc1) (ref: efm_symbol.py:84-101)."""
    pool1 = group(data, 0, 99, (5, 5), (1, 1), (2, 2), str(1))
    pool2 = group(pool1, 99, 198, (3, 3), (1, 1), (1, 1), str(2), 1)
    pool3 = group(pool2, 198, 387, (3, 3), (1, 1), (1, 1), str(3), 2)
    pool4 = group(pool3, 387, 261, (3, 3), (1, 1), (1, 1), str(4), 3)
    pool5 = group(pool4, 261, 261, (3, 3), (1, 1), (1, 1), str(5), 4)
    fc1 = G.FullyConnected(pool5, fc_hidden, name="fc1")
    feat = G.MFM(fc1, 3, G.ORDER_RES, name="concat29")  # the node final_efm.py:208 pulls out as 'concat29_output'
    return feat, fc1


def multi_gpu(data, classes):
    """Returns (id-head output, fc1) like the reference (efm_symbol.py:81-110); the id head is
    Dropout(0.7) -> FullyConnected(classes) -> softmax and is only materialised when it is asked for."""
    feat, fc1 = efm_feature(data)
    fc1.feature = feat
    fc2 = G.FullyConnected(feat, classes, name="fc2")  # dropout is applied by the id-head wrapper in training
    return fc2, fc1


def get_net(classes, margin=0.2):
    """Two-output network [id logits, 342-d feature] (ref: efm_symbol.py:112-123, intended behaviour:
    the reference's MakeLoss(<class>) line is unusable and is not reproduced)."""
    data = G.Variable("data")
    logits, fc1 = multi_gpu(data, classes)
    return [logits, fc1.feature]


def embedding_net(embed_dim=128, normalize=True, fc_hidden=513):
    """BASELINE config 2: EFM-29 -> per-row L2 norm -> Dense(embed_dim, use_bias=False)
    (ref: final_efm.py:240-243 + pre-trained_efm_v3.py:180-181).  Returns [embedding, feature]."""
    data = G.Variable("data")
    feat, _ = efm_feature(data, fc_hidden)
    x = G.L2Normalization(feat, name="l2norm") if normalize else feat
    emb = G.FullyConnected(x, embed_dim, name="head", no_bias=True)
    return [emb, feat]


LIGHTCNN9_PLAN = [  # (name, 1x1 channels or 0, kxk channels, kernel, pad, pool) — BASELINE configs[2], SURVEY.md §8d
    ("1", 0, 96, 5, 2, True), ("2", 96, 192, 3, 1, True), ("3", 192, 384, 3, 1, True), ("4", 384, 256, 3, 1, False),
    ("5", 256, 256, 3, 1, True)]


def lightcnn9_feature(data, fc_hidden=512):
    """LightCNN-9 (build-defined: the reference only carries its 2-way MFM branch, efm_symbol.py:62-64,76-77):
    conv5x5(96) MFM2 pool, then [conv1x1 -> MFM2 -> conv3x3 -> MFM2 (-> pool)] x 4, fc -> MFM2 -> 256-d."""
    x = data
    for layer, num_r, num, k, pad, pool in LIGHTCNN9_PLAN:
        if num_r:
            x = G.MFM(G.Convolution(x, num_r, (1, 1), name="conv%s_r" % layer), 2, G.ORDER_GROUP, name="mfm%s_r" % layer)
        x = G.MFM(G.Convolution(x, num, (k, k), name="conv%s" % layer, pad=(pad, pad)), 2, G.ORDER_GROUP, name="mfm%s" % layer)
        if pool:
            x = G.Pooling(x, name="pool%s" % layer)
    fc1 = G.FullyConnected(x, fc_hidden, name="fc1")
    return G.MFM(fc1, 2, G.ORDER_GROUP, name="mfm_fc1")


def lightcnn9_embedding_net(normalize=True):
    """[256-d embedding (per-row L2 normalised), raw 256-d feature]."""
    data = G.Variable("data")
    feat = lightcnn9_feature(data)
    emb = G.L2Normalization(feat, name="l2norm") if normalize else feat
    return [emb, feat]


# BASELINE configs[4]: "DeepFace-style deeper CNN, 512-d embedding" — named by the reference's README (README.md:9,17-19) without
# any code; build-defined here as the LightCNN recipe carried deeper and wider (13 convolutions + fc, 2-way MFM after every
# convolution, four 2x2 poolings).  Flat list of (name, output channels before MFM, kernel, pad, pool after).
DEEPCNN_LAYERS = [
    ("conv1", 96, 5, 2, True),
    ("conv2_r", 96, 1, 0, False), ("conv2a", 192, 3, 1, False), ("conv2b", 192, 3, 1, True),
    ("conv3_r", 192, 1, 0, False), ("conv3a", 384, 3, 1, False), ("conv3b", 384, 3, 1, True),
    ("conv4_r", 384, 1, 0, False), ("conv4a", 512, 3, 1, False), ("conv4b", 512, 3, 1, False),
    ("conv5_r", 512, 1, 0, False), ("conv5a", 512, 3, 1, False), ("conv5b", 512, 3, 1, True)]


def mfm2_stack_feature(data, layers, fc_hidden):
    """[conv -> MFM2 (-> pool)] per entry of `layers`, then fc -> MFM2: every convolution's MFM / pooling folds into its epilogue,
    so the plan runs in fp32 or bf16."""
    x = data
    for name, num, k, pad, pool in layers:
        x = G.MFM(G.Convolution(x, num, (k, k), name=name, pad=(pad, pad)), 2, G.ORDER_GROUP, name="mfm_" + name)
        if pool:
            x = G.Pooling(x, name="pool_" + name)
    return G.MFM(G.FullyConnected(x, fc_hidden, name="fc1"), 2, G.ORDER_GROUP, name="mfm_fc1")


def deepcnn_embedding_net(normalize=True):
    """[512-d embedding (per-row L2 normalised), raw 512-d feature] of the build-defined deeper CNN (BASELINE configs[4])."""
    data = G.Variable("data")
    feat = mfm2_stack_feature(data, DEEPCNN_LAYERS, 1024)
    emb = G.L2Normalization(feat, name="l2norm") if normalize else feat
    return [emb, feat]
